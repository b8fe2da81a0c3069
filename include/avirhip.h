/*
 * avirhip.h -- C ABI of libavirhip: the MI355X (gfx950) executor for the
 * separable-resize hot path of avaneev/avir.
 *
 * The reference (header-only C++, /root/reference) has no FFI of its own; its
 * extension point for this path is the `fpclass` template parameter
 * (avir.h:4569-4592), a per-scanline interface. This header is the boundary a
 * drop-in replacement binds instead: the reference's front-end
 * (avir::CImageResizer<>::resizeImage, avir.h:4680-4684; avir::CLancIR::
 * resizeImage, lancir.h:386-390) keeps its signature, runs the planner on the
 * host, and hands the immutable plan + the two image buffers across this ABI.
 * See INTEGRATION.md for the reference-side stub.
 *
 * Conventions: plain C, plain pointers and sizes. Every function returning
 * `int` returns 0 on success and a negative AVIRHIP_E* code on error;
 * avirhip_last_error() returns a thread-local message. No C++ exception
 * crosses the boundary: every entry point is a function-try-block, a failed
 * host allocation comes back as AVIRHIP_ENOMEM (a failed device allocation as
 * AVIRHIP_EHIP / AVIRHIP_ENOMEM), anything else as AVIRHIP_EINTERNAL, and the
 * object the call worked on stays usable. Image geometries whose element or
 * byte counts do not fit the reference's own `int` / `size_t` arithmetic
 * (avir.h:4786-4794, lancir.h:409-430) are refused with AVIRHIP_EINVAL before
 * anything is allocated. The library copies every table it is given; the
 * caller keeps ownership of its arrays.
 */
#ifndef AVIRHIP_H
#define AVIRHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVIRHIP_OK 0
#define AVIRHIP_EINVAL (-1)      /* bad argument / malformed plan */
#define AVIRHIP_ENODEV (-2)      /* no usable gfx950 device */
#define AVIRHIP_EHIP (-3)        /* HIP runtime error (see last_error) */
#define AVIRHIP_EUNSUPPORTED (-4)/* valid request outside the built scope */
#define AVIRHIP_ENOMEM (-5)       /* host or device memory exhausted */
#define AVIRHIP_EINTERNAL (-6)    /* an unexpected C++ exception was caught at the
                                     boundary (never crosses it; see last_error) */

/* Element types of the source / destination image (Tin/Tout of
 * resizeImage<Tin,Tout>, avir.h:4670-4677; lancir.h:373-381). */
typedef enum avirhip_dtype {
	AVIRHIP_U8 = 0,
	AVIRHIP_U16 = 1,
	AVIRHIP_F32 = 2,
	AVIRHIP_F64 = 3,
	/* CLancIR only: 32-bit unsigned elements whose value range is treated as
	 * uint16_t's, 0..65535 (lancir.h:373-377; Clamp / OutMul derive from
	 * sizeof() == 1 or not, lancir.h:531-533). */
	AVIRHIP_U32 = 4
} avirhip_dtype;

/* Where a buffer passed to an execute call lives. */
#define AVIRHIP_MEM_HOST 0
#define AVIRHIP_MEM_DEVICE 1
#define AVIRHIP_MEM_AUTO 2 /* ask the HIP runtime (hipPointerGetAttributes) */

/* ---------------------------------------------------------------------
 * AVIR plan description: a faithful, flattened mirror of the reference's
 * per-axis filtering-step list (CImageResizerFilterStep, avir.h:2568-2728)
 * after buildFilterSteps / updateFilterStepBuffers / updateBufLenAndRPosPtrs
 * (avir.h:5616-6127) have run.
 * ------------------------------------------------------------------- */

typedef enum avirhip_step_kind {
	AVIRHIP_STEP_FIR = 0,          /* doFilter, avir.h:3748-3866 */
	AVIRHIP_STEP_UP_ZEROSTUFF = 1, /* doUpsample, FltOrig kept, 3260-3402 */
	AVIRHIP_STEP_UP_FILTERED = 2,  /* doUpsample with filter + DC tails, 3404-3733 */
	AVIRHIP_STEP_RESIZE = 3,       /* doResize, avir.h:3884-4096 */
	AVIRHIP_STEP_RESIZE2 = 4       /* doResize2, avir.h:4114-4328 */
} avirhip_step_kind;

/* One entry of the resizing-position table (CResizePos, avir.h:2640-2648;
 * filled by fillRPosBuf 5782-5808 and updateBufLenAndRPosPtrs 6063-6126). */
typedef struct avirhip_rpos {
	int32_t src_pos_int; /* floor(o + k*j) */
	int32_t fti;         /* fractional-delay filter index */
	int32_t phase;       /* index of `fti` in the step's compact phase list */
	float x;             /* order-1 interpolation coefficient (float) */
	int32_t src_offs_px; /* SrcOffs / ElCount: first source pixel read */
	int32_t fl;          /* taps walked (RESIZE2: FilterLen - fo, step 2) */
	int32_t ftp_off;     /* fo: offset of the first tap inside the phase */
	double x64;          /* double pipeline: the coefficient in double (x unused) */
} avirhip_rpos;

typedef struct avirhip_step {
	int32_t kind;            /* avirhip_step_kind */
	int32_t resample_factor; /* ResampleFactor (0 for RESIZE/RESIZE2) */
	int32_t in_len, in_prefix, in_suffix;
	int32_t out_len, out_prefix, out_suffix;
	int32_t edge_pixel_count;

	/* FIR / UP_FILTERED: the step's filter (Flt) and latency. Unused (len 0)
	 * for UP_ZEROSTUFF and RESIZE*. */
	int32_t flt_len;
	int32_t flt_latency;
	const float* flt;
	/* UP_FILTERED only: PrefixDC / SuffixDC tails (avir.h:5307-5353). */
	int32_t prefix_dc_len;
	const float* prefix_dc;
	int32_t suffix_dc_len;
	const float* suffix_dc;

	/* RESIZE / RESIZE2: the fractional-delay bank, compacted to the phases
	 * the position table references. Phase p holds `bank_filter_len`
	 * order-0 taps followed (order 1 only) by `bank_filter_len` difference
	 * taps (avir.h:1838-1841). */
	int32_t bank_filter_len;
	int32_t bank_order;      /* 0 or 1 */
	int32_t bank_frac_count; /* FracCount of the full bank */
	int32_t n_phases;
	const int32_t* phase_fti; /* [n_phases], ascending */
	const float* phase_taps;  /* [n_phases][filter_len * (order + 1)] */
	const avirhip_rpos* rpos; /* [out_len] */

	/* Double pipeline (avirhip_plan_desc::work_f64, the reference's
	 * fpclass_def<double>): the same tables in double precision -- the
	 * float fields above are null then, and rpos[].x64 replaces rpos[].x. */
	const double* flt64;
	const double* prefix_dc64;
	const double* suffix_dc64;
	const double* phase_taps64;
} avirhip_step;

typedef struct avirhip_axis {
	int32_t n_steps;
	const avirhip_step* steps;
	int32_t build_mode; /* informational: the mode the planner selected */
} avirhip_axis;

typedef struct avirhip_plan_desc {
	int32_t src_w, src_h;
	int32_t src_stride_elems; /* SrcScanlineSize (elements); <1 => w*ch */
	int32_t new_w, new_h;
	int32_t channels;         /* ElCountIO, 1..4 */
	int32_t in_type, out_type;/* avirhip_dtype */
	avirhip_axis h;           /* horizontal pass: src rows -> FltBuf */
	avirhip_axis v;           /* vertical pass: FltBuf columns -> result */
	/* Integer-output epilogue (avir.h:5029-5045, 4392-4419). Ignored for
	 * float/double output. */
	double tr_mul;            /* TrMul (1.0: no bit-depth truncation) */
	double pk_out;            /* PkOut: 255 or 65535 */
	/* sRGB gamma (CImageResizerVars::UseSRGBGamma / AlphaIndex, avir.h:2526-2527):
	 * the source is linearised while it is packed (packScanline gamma branch,
	 * avir.h:2841-2930, convertSRGB2Lin 206-291) and the result de-linearised
	 * before the dither/unpack stage (applySRGBGamma, avir.h:2982-3068,
	 * convertLin2SRGB 301-312); the alpha channel (index 0 or 3 of 4-channel
	 * pixels, -1 = none) is only scaled. As in the reference, float32 output
	 * is the vertical pass' in-place result and stays linear (avir.h:4956-4979). */
	int32_t use_srgb_gamma;
	int32_t alpha_index;
	/* Ditherer of the integer-output stage -- the reference selects it with
	 * the third template parameter of fpclass_def (avir.h:4569-4592):
	 * AVIRHIP_DITHER_DEF = CImageResizerDithererDefINL (round + clamp,
	 * avir.h:4351-4427), AVIRHIP_DITHER_ERRD = CImageResizerDithererErrdINL
	 * (error diffusion along and across rows, avir.h:4441-4537; "recursive":
	 * rows depend on each other, so such a plan cannot execute row bands). */
	int32_t dither;
	/* 1: the double pipeline (the reference's fpclass_def<double>,
	 * avir.h:4553-4560): tables in the steps' ...64 fields, every pass computes
	 * in double. 0: float (fpclass_def<float>, fpclass_float4). */
	int32_t work_f64;
} avirhip_plan_desc;

#define AVIRHIP_DITHER_DEF 0
#define AVIRHIP_DITHER_ERRD 1
/* CImageResizerDithererDefINL over the SSE value type of fpclass_float4: its
 * round() is cvtps_epi32 under nearest-even (avir_float4_sse.h:303-313)
 * instead of avir::round's (int)(d + 0.5) (avir.h:130-135). Such a plan also
 * has no in-place float output (avir.h:4956: sizeof(fptype) != sizeof(Tout)),
 * i.e. float / double results pass through the output stage, and with sRGB
 * gamma they ARE de-linearised. */
#define AVIRHIP_DITHER_DEF_RNE 2

/* ---------------------------------------------------------------------
 * LANCIR plan description (CLancIR internals, lancir.h:840-1381).
 * ------------------------------------------------------------------- */

typedef struct avirhip_lancir_pos {
	int32_t flt_index; /* index into the axis' `filters` array */
	int32_t so;        /* first source pixel read, relative to the padded
	                      scanline start (CResizePos::so, lancir.h:1231) */
} avirhip_lancir_pos;

typedef struct avirhip_lancir_axis {
	int32_t kernel_len;       /* KernelLen, even */
	int32_t padl, padr;       /* CResizeScanline::padl/padr */
	int32_t src_len, dst_len;
	int32_t n_filters;
	const int32_t* filter_frac; /* [n_filters] phase index 0..1000 */
	const float* filters;       /* [n_filters][kernel_len], un-replicated */
	const avirhip_lancir_pos* pos; /* [dst_len] */
} avirhip_lancir_axis;

typedef struct avirhip_lancir_desc {
	int32_t src_w, src_h, src_stride_elems;
	int32_t new_w, new_h, new_stride_elems;
	int32_t channels;
	int32_t in_type, out_type;
	avirhip_lancir_axis v; /* vertical pass runs first (lancir.h:601-646) */
	avirhip_lancir_axis h;
	float out_mul;         /* OutMul, lancir.h:532-533 */
	float clamp;           /* Clamp, lancir.h:531 */
	int32_t is_unity_mul;  /* IsUnityMul, lancir.h:528-529 */
} avirhip_lancir_desc;

/* ---------------------------------------------------------------------
 * Device / plan / execute.
 * ------------------------------------------------------------------- */

typedef struct avirhip_plan avirhip_plan;

/* Number of usable gfx950 devices (0 when there is none). */
int avirhip_device_count(void);
/* Binds the calling thread to `device` (hipSetDevice). */
int avirhip_init(int device);
/* Thread-local message of the calling thread's most recent avirhip_* call that
 * returns a status: every such entry point empties it when it begins, so the
 * text never outlives the call it describes ("" after a call that succeeded). */
const char* avirhip_last_error(void);
/* Library version string. */
const char* avirhip_version(void);

/* Uploads an AVIR plan to the current device. The plan is immutable and may
 * be shared by threads: executions that need the plan's scratch buffers are
 * serialised inside the library (a mutex on the host side, an event chain on
 * the device side); device-resident float RGBA calls of the single-launch 2x
 * kernels need no scratch and run concurrently. */
int avirhip_plan_create(const avirhip_plan_desc* desc, avirhip_plan** out);
int avirhip_lancir_plan_create(const avirhip_lancir_desc* desc,
	avirhip_plan** out);
void avirhip_plan_destroy(avirhip_plan* plan);

/* Plan tuning: force an execution path, for testing and A/B benchmarking.
 * 0 = automatic (fastest applicable), 1 = generic per-step kernels (every
 * plan), 2 = LDS-tiled two-pass kernels, 3 = LDS-tiled fused single launch,
 * 4 = specialised exact-2x RGBA marching kernel (single launch), 5 = the
 * general-ratio pass kernels (one launch per axis, autonomous wavefronts).
 * Returns AVIRHIP_EUNSUPPORTED if the path cannot run this plan. */
int avirhip_plan_set_path(avirhip_plan* plan, int path);
/* The path the next execution will take (1..5). */
int avirhip_plan_get_path(const avirhip_plan* plan);

/* Kernel variants inside a path, for differential tests and A/B timing (every
 * variant is bit-identical). `variant` is a bit mask, 0 = the automatic choice:
 * AVIRHIP_VARIANT_UP2_PLAIN_V   path 4 runs the 13-row-window form of the
 *     vertical phase (the form a plan whose 2x bank phase is not bit-symmetric
 *     takes) instead of the transposed, product-sharing one;
 * AVIRHIP_VARIANT_DN_TWO_PASS   integer-ratio downsizing (k = 2, 3) runs its
 *     two pass kernels through FltBuf instead of the single fused launch;
 * AVIRHIP_VARIANT_SACC_LADDER   the streaming-accumulation kernel of
 *     downsizing axes runs its general form (one branch per alive output, the
 *     form float sources take) where the branch-free form for integer sources
 *     would run. */
#define AVIRHIP_VARIANT_UP2_PLAIN_V 1
#define AVIRHIP_VARIANT_DN_TWO_PASS 2
#define AVIRHIP_VARIANT_SACC_LADDER 4
/* upsizing on both axes (path 5): the horizontal and the vertical pass kernel
 * through FltBuf instead of the single fused launch of gfuse.hip */
#define AVIRHIP_VARIANT_UPG_TWO_PASS 8
/* ... and the fused launch whatever the frame size (automatically: frames of
 * 7 Mpixels of output and more; CLancIR plans with integer results: up to
 * about 3x horizontally) */
#define AVIRHIP_VARIANT_UPG_FUSED 16
/* float RGBA sources on downsizing axes: the branch-free accumulation kernels
 * run optimistically (exact kernels behind an alarm) as they do for float
 * pixels of 1-3 channels; automatically such sources take the exact kernels
 * alone, which are as fast there and need no alarm launches */
#define AVIRHIP_VARIANT_SACC_OPTIMISTIC 32
int avirhip_plan_set_variant(avirhip_plan* plan, int variant);

/* Device memory the plan holds right now, in bytes: tables, scratch buffers
 * (allocated when a path first needs them), its inner plan, spares and
 * replicas -- what the front ends' plan caches count against their bound. */
unsigned long long avirhip_plan_device_bytes(avirhip_plan* plan);

/* Executes H pass + V pass (+ integer epilogue) of `plan`:
 * src -> dst. `src_mem`/`dst_mem` are AVIRHIP_MEM_HOST or _DEVICE; host
 * buffers are staged through device memory inside the call. `stream` is a
 * hipStream_t (NULL = default stream). With device buffers the call is
 * asynchronous with respect to the host. */
int avirhip_resize(avirhip_plan* plan, const void* src, int src_mem,
	void* dst, int dst_mem, void* stream);

/* Row-band execution: computes only output rows [row0, row1) of the plan's
 * destination image into `dst_band` (tightly packed band of row1-row0 rows).
 * The source is the WHOLE source image (device or host). Used for multi-GPU
 * sharding: bands computed on different devices are bit-identical to the
 * corresponding rows of a full avirhip_resize(). */
int avirhip_resize_band(avirhip_plan* plan, const void* src, int src_mem,
	void* dst_band, int dst_mem, int row0, int row1, void* stream);

/* The source rows [*first, *last] (inclusive) that output rows [row0, row1)
 * read -- the reference's vertical windows with their replicated edges folded
 * in (avir.h:6560-6619 resizeScanlineV over FltBuf columns; lancir.h:560-640).
 * No other row of `src` influences the band (tools/fuzz_values.py poisons them
 * with NaNs): a caller that streams the source -- row bands of a frame arriving
 * over a link, tiles of a larger image -- only has to have these rows in place
 * when it makes the call. `src` must still address a whole, mapped frame: the
 * marching kernels start their chunks at chunk-aligned rows and may LOAD rows
 * outside the range (their results are discarded, never stored). A caller that
 * holds only the window passes avirhip_resize_window() its rows instead. */
int avirhip_band_source_rows(const avirhip_plan* plan, int row0, int row1,
	int* first, int* last);

/* Row-band execution from a WINDOW of the source: `src_rows` holds only the
 * source rows [first_row, first_row + n_rows) (same row pitch as the plan's
 * source), which must cover avirhip_band_source_rows(row0, row1) -- otherwise
 * AVIRHIP_EINVAL. This is the multi-GPU distribution of SURVEY.md 8(e): rank g
 * receives its band's rows plus the halo and nothing else (the reference's
 * analogue: scanlines dealt to worker threads, avir.h:4885-4890, 4967-4971);
 * image borders are the frame's, never the window's, so the band is
 * bit-identical to the same rows of avirhip_resize(). No other source byte
 * moves, and for the plans the sharded configurations run -- exact 2x, float
 * RGBA in and out: the marching kernels k_up2 / k_lanc2 -- the plan holds
 * O(window) device memory: a DEVICE window is read where it lies (no staging,
 * no copy; nothing outside [src_rows, src_rows + n_rows rows) is touched -- the
 * kernels clamp their row indices to the window, which for every access that
 * reaches a result is the frame's own edge replication), a HOST window is
 * uploaded into a window-sized staging buffer. Other plans copy the window to
 * its place in a frame-sized staging buffer of the plan (their kernels index
 * the frame; what they load outside the window never reaches a result). */
int avirhip_resize_window(avirhip_plan* plan, const void* src_rows,
	int src_mem, int first_row, int n_rows, void* dst_band, int dst_mem,
	int row0, int row1, void* stream);

/* One frame sharded over several devices of THIS process:
 * the destination rows are split into `n_gpus` contiguous bands, band g is
 * computed on device `devices[g]` (a replica of the plan is created there on
 * first use; `devices` == NULL or repeated entries run the bands one after
 * another on the same device) and the bands are stitched into `dst` -- peer
 * copies over xGMI when `dst` is device memory, device-to-host copies
 * otherwise. `src`/`dst` are whole images, host or device memory (detected).
 * The stitched frame is bit-identical to avirhip_resize(). `gather_root` is
 * the index of the band whose device should hold `dst` when the caller
 * allocates it there (informational; `dst` is used where it is). The two
 * optional outputs receive the wall time of the compute and gather phases.
 * The multi-PROCESS form of the same sharding (one process per GPU, RCCL) is
 * avirhip_resize_band + avir_amd/sharding.py. */
int avirhip_resize_sharded(avirhip_plan* plan, int n_gpus, const int* devices,
	const void* src, void* dst, int gather_root, double* t_compute_ms,
	double* t_gather_ms);

/* Timing aid for bench.py: runs `iters` executions on device buffers and
 * returns the average duration in milliseconds measured with HIP events on
 * `stream` (the stream the kernels are launched on). */
int avirhip_time_resize(avirhip_plan* plan, const void* src, void* dst,
	int iters, void* stream, double* avg_ms);

/* Synthetic-workload aids for bench.py / smoke (host only): the SURVEY.md
 * 8(c) input generators -- u8: s = seed; per byte s = s*1664525 + 1013904223
 * (uint32), v = s >> 24; f32: the same LCG, x = (s >> 8) * 2^-24 -- and the
 * FNV-1a-64 hash the golden vectors are recorded with (offset basis
 * 1469598103934665603, prime 1099511628211). */
void avirhip_fill_lcg_u8(uint8_t* p, size_t n, uint32_t seed);
void avirhip_fill_lcg_f32(float* p, size_t n, uint32_t seed);
uint64_t avirhip_fnv1a64(const void* p, size_t n);

/* ---------------------------------------------------------------------
 * Front-end mirror: the reference's host-side planner re-implemented inside
 * the library (double precision, -ffp-contract=off, same libm), so that a
 * caller gets "resizeImage" semantics in one call. Names and argument
 * meaning follow avir::CImageResizer (avir.h:4630-4684) and avir::CLancIR
 * (lancir.h:386-390).
 * ------------------------------------------------------------------- */

/* CImageResizerParams (avir.h:2262-2317). */
typedef struct avirhip_params {
	double CorrFltAlpha, CorrFltLen, IntFltAlpha, IntFltCutoff, IntFltLen;
	double LPFltAlpha, LPFltBaseLen, LPFltCutoffMult;
	double HBFltAlpha, HBFltCutoff, HBFltLen;
} avirhip_params;

/* Presets, avir.h:2328-2464. */
typedef enum avirhip_preset {
	AVIRHIP_PARAMS_DEF = 0,
	AVIRHIP_PARAMS_ULR = 1,
	AVIRHIP_PARAMS_LR = 2,
	AVIRHIP_PARAMS_LOW = 3,
	AVIRHIP_PARAMS_HIGH = 4,
	AVIRHIP_PARAMS_ULTRA = 5
} avirhip_preset;

void avirhip_params_preset(int preset, avirhip_params* out);

/* The caller-settable part of CImageResizerVars (avir.h:2516-2547). */
typedef struct avirhip_vars {
	double ox, oy;
	int32_t UseSRGBGamma; /* non-zero: sRGB gamma linearisation (avir.h:2526) */
	int32_t AlphaIndex;
	int32_t BuildMode;    /* -1 = automatic */
	int32_t RndSeed;
} avirhip_vars;

void avirhip_vars_default(avirhip_vars* out);

typedef struct avirhip_resizer avirhip_resizer;

/* CImageResizer(aResBitDepth, aSrcBitDepth, aParams), avir.h:4630-4639.
 * `params` may be NULL (CImageResizerParamsDef). */
int avirhip_resizer_create(int res_bit_depth, int src_bit_depth,
	const avirhip_params* params, avirhip_resizer** out);
void avirhip_resizer_destroy(avirhip_resizer* r);
/* Selects the ditherer class (fpclass::CDitherer, avir.h:4591) of every
 * following resize of this object: AVIRHIP_DITHER_DEF (default) or
 * AVIRHIP_DITHER_ERRD. */
int avirhip_resizer_set_ditherer(avirhip_resizer* r, int dither);
/* Selects the `fptype` packing of the mirrored fpclass (avir.h:4569-4592):
 * 1 = fpclass_def<float> (default), 4 = fpclass_float4 of
 * avir_float4_sse.h (fpclass_def<float4, float>). The arithmetic per channel
 * is the same; what changes is what the reference derives from fppack:
 * ElCount = ceil(ElCountIO / fppack) in the build-mode cost model
 * (avir.h:4786, 6206-6270 -- the automatic BuildMode can differ), the
 * ditherer's rounding and the float-output path (AVIRHIP_DITHER_DEF_RNE),
 * and the CImageResizerVarsBase write-back. The error-diffusion ditherer is
 * only built for fppack 1.
 * AVIRHIP_FPCLASS_DOUBLE = fpclass_def<double> (avir.h:4553-4560): the double
 * pipeline -- tables designed and stored in double, every pass computes in
 * double (the generic kernels; default ditherer only), double output is the
 * vertical pass' in-place result. */
#define AVIRHIP_FPCLASS_DOUBLE 64
int avirhip_resizer_set_fpclass(avirhip_resizer* r, int fppack);

/* Host-only: runs the planner exactly as resizeImage() would for this call
 * and returns the plan description (owned by the library; release with
 * avirhip_plan_desc_free). Needs no GPU. */
int avirhip_resizer_build_desc(avirhip_resizer* r, int src_w, int src_h,
	int src_scanline_size, int new_w, int new_h, int el_count_io, double k,
	const avirhip_vars* vars, int in_type, int out_type,
	avirhip_plan_desc** out);
void avirhip_plan_desc_free(avirhip_plan_desc* desc);

/* The fields of CImageResizerVarsBase (avir.h:2473-2506) as resizeImage()
 * leaves them in the caller's CImageResizerVars (avir.h:4663-4669): the
 * vertical axis is planned last, so k, o, BufLen, BufOffs, ResizeStep and
 * IsResize2 describe its step list. InGammaMult / OutGammaMult are only
 * written when UseSRGBGamma is set (`gamma_valid`). */
typedef struct avirhip_vars_base {
	int32_t ElCount, ElCountIO, fppack, fpalign, elalign, packmode;
	int32_t BufLen[2], BufOffs[2];
	double k, o;
	int32_t ResizeStep, IsResize2;
	int32_t gamma_valid;
	double InGammaMult, OutGammaMult;
} avirhip_vars_base;

/* Host-only (no GPU needed): what resizeImage() with these arguments writes
 * back into *aVars. */
int avirhip_resizer_vars_base(avirhip_resizer* r, int src_w, int src_h,
	int src_scanline_size, int new_w, int new_h, int el_count_io, double k,
	const avirhip_vars* vars, int in_type, int out_type,
	avirhip_vars_base* out);

/* resizeImage<Tin,Tout>(), avir.h:4680-4684. Plans are cached per geometry
 * inside the resizer. Zero-sized source => destination zero-filled
 * (NewWidth*NewHeight elements, avir.h:4686-4692); zero-sized destination
 * => no-op. */
int avirhip_resizer_resize(avirhip_resizer* r, const void* src, int src_mem,
	int src_w, int src_h, int src_scanline_size, void* dst, int dst_mem,
	int new_w, int new_h, int el_count_io, double k,
	const avirhip_vars* vars, int in_type, int out_type, void* stream);

/* Host-only (no GPU needed): avirhip_band_source_rows() for a geometry, from
 * the planner alone -- what a host that deals source rows to the ranks of a
 * sharded frame needs before any device plan exists. */
int avirhip_resizer_band_source_rows(avirhip_resizer* r, int src_w, int src_h,
	int src_scanline_size, int new_w, int new_h, int el_count_io, double k,
	const avirhip_vars* vars, int in_type, int out_type, int row0, int row1,
	int* first, int* last);

/* Returns (and caches) the device plan resizeImage() would use for this
 * geometry; owned by the resizer. */
int avirhip_resizer_get_plan(avirhip_resizer* r, int src_w, int src_h,
	int src_scanline_size, int new_w, int new_h, int el_count_io, double k,
	const avirhip_vars* vars, int in_type, int out_type,
	avirhip_plan** out);

/* CLancIRParams, lancir.h:260-307. */
typedef struct avirhip_lancir_params {
	int32_t SrcSSize, NewSSize;
	double kx, ky, ox, oy, la;
} avirhip_lancir_params;

void avirhip_lancir_params_default(avirhip_lancir_params* out);

typedef struct avirhip_lancir avirhip_lancir;

int avirhip_lancir_create(avirhip_lancir** out);
void avirhip_lancir_destroy(avirhip_lancir* l);

int avirhip_lancir_build_desc(avirhip_lancir* l, int src_w, int src_h,
	int new_w, int new_h, int el_count, const avirhip_lancir_params* params,
	int in_type, int out_type, avirhip_lancir_desc** out);
void avirhip_lancir_desc_free(avirhip_lancir_desc* desc);

/* CLancIR::resizeImage<Tin,Tout>(), lancir.h:386-390. Returns NewHeight on
 * success, 0 on a parameter error (lancir.h:392-407), <0 on a device error. */
int avirhip_lancir_resize(avirhip_lancir* l, const void* src, int src_mem,
	int src_w, int src_h, void* dst, int dst_mem, int new_w, int new_h,
	int el_count, const avirhip_lancir_params* params, int in_type,
	int out_type, void* stream);

/* Host-only: avirhip_band_source_rows() for a CLancIR geometry. */
int avirhip_lancir_band_source_rows(avirhip_lancir* l, int src_w, int src_h,
	int new_w, int new_h, int el_count, const avirhip_lancir_params* params,
	int in_type, int out_type, int row0, int row1, int* first, int* last);

int avirhip_lancir_get_plan(avirhip_lancir* l, int src_w, int src_h,
	int new_w, int new_h, int el_count, const avirhip_lancir_params* params,
	int in_type, int out_type, avirhip_plan** out);

/* Test hook: the `nth` host allocation (operator new) the CALLING THREAD makes
 * inside the library from now on throws std::bad_alloc (1 = the next one;
 * 0 = off). tests/test_abi.py sweeps it over every allocation of the planning
 * and plan-creation entry points: each must return AVIRHIP_ENOMEM with a
 * message -- never abort the process. Returns the allocations counted on this
 * thread since the previous call. */
long avirhip_debug_fail_alloc(long nth);

/* Test hook: the uint8 sRGB-gamma output stage (applySRGBGamma + default
 * ditherer + unpackScanline, avir.h:2982-3068, 4392-4419, 3155-3215) applied to
 * `npx` float RGBA pixels in device memory, once in the form plans use (the
 * 2 x 256 threshold table, a monotone step function found by bisection on the
 * host) and once as the direct expressions: tests/test_gpu_parity.py sweeps
 * every float bit pattern through both. `tr_mul` >= 1 (1: no bit-depth
 * truncation), `alpha_index` 0..3 or -1. */
int avirhip_debug_gamma_u8(const float* dev_rgba, long npx, double tr_mul,
	int alpha_index, unsigned char* dev_out_table,
	unsigned char* dev_out_direct, void* stream);

#ifdef __cplusplus
}
#endif

#endif /* AVIRHIP_H */
