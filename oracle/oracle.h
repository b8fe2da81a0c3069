/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's algorithm for the separable-resize
 * hot path (avaneev/avir v3.1): the AVIR planner + scanline executor and the
 * LANCIR resizer. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call this; the product path (libavirhip.so) never does.
 *
 * Parity status: PINNED -- every function is checked bit-for-bit against the
 * real reference compiled from /root/reference (oracle/_ref/libavir_ref.so,
 * built by oracle/Makefile) and against the golden FNV-1a hashes recorded in
 * SURVEY.md 8(c) (tests/test_oracle.py, tests/golden/).
 */
#ifndef AVIR_ORACLE_H
#define AVIR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Element type codes (same values as avirhip_dtype). */
#define ORACLE_U8 0
#define ORACLE_U16 1
#define ORACLE_F32 2
#define ORACLE_F64 3
#define ORACLE_U32 4 /* CLancIR only: "treated as uint16_t", lancir.h:376-377 */

/* avir::CImageResizer<fpclass_def<float>>::resizeImage<Tin,Tout>()
 * (avir.h:4680-5092). `params` = 8 doubles {CorrFltAlpha, CorrFltLen,
 * IntFltAlpha, IntFltCutoff, IntFltLen, LPFltAlpha, LPFltBaseLen,
 * LPFltCutoffMult} or NULL for CImageResizerParamsDef. Returns 0, or <0 on an
 * unsupported request. `mode_h`/`mode_v`, if not NULL, receive the build modes
 * selected per axis. */
int oracle_avir_resize( const void* src, int sw, int sh, int sstride,
	void* dst, int nw, int nh, int ch, double k, int in_type, int out_type,
	int resbits, int srcbits, const double* params, double ox, double oy,
	int build_mode, int* mode_h, int* mode_v );

/* CImageResizerVars::UseSRGBGamma / AlphaIndex (avir.h:2526-2527) of the
 * following oracle_avir_resize() calls; default off / -1. */
void oracle_set_gamma( int use, int alpha_index );
/* 0 = CImageResizerDithererDefINL, 1 = CImageResizerDithererErrdINL
 * (avir.h:4351-4537) for the following oracle_avir_resize() calls. */
void oracle_set_dither( int kind );

/* The regenerated uint8_t linearisation table (avir.h:229-291). */
void oracle_srgb_u8_table( float* tbl );

/* avir::CLancIR::resizeImage<Tin,Tout>() (lancir.h:386-713), x86-64 SSE2
 * summation order. Returns NewHeight, or 0 on a parameter error. */
int oracle_lancir_resize( const void* src, int sw, int sh, void* dst, int nw,
	int nh, int ch, int in_type, int out_type, int src_ssize, int new_ssize,
	double kx, double ky, double ox, double oy, double la );

/* FNV-1a 64-bit hash over raw bytes (offset basis 1469598103934665603, prime
 * 1099511628211) -- the hash SURVEY.md 8(c) records its golden vectors with. */
uint64_t oracle_fnv1a64( const void* p, size_t n );

/* SURVEY.md 8(c) input generators. u8: s=seed; per byte s=s*1664525+1013904223
 * (uint32), v=s>>24. f32: same LCG, x=(s>>8)*(1/16777216). */
void oracle_fill_u8( uint8_t* p, size_t n, uint32_t seed );
void oracle_fill_f32( float* p, size_t n, uint32_t seed );

#ifdef __cplusplus
}
#endif

#endif /* AVIR_ORACLE_H */
