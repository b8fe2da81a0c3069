// up2.hip -- the headline kernel: exact 2x RGBA upsizing (cfg2/cfg3 of
// BASELINE.json), H pass + V pass fused in ONE launch.
//
// What the plan looks like (default parameters, mode 1 or 3), per
// axis with source length W:
//   C[m]  = f0*S[m-3] + sum_{i=1..3} f_i*(S[m-3+i] + S[m-3-i])     (7-tap FIR,
//           EdgePixelCount 3; doFilter avir.h:3748)
//   up    = zero-stuffed 2x view of C                   (doUpsample, 3260-3402)
//   o[2q]   = 0 + sum_{t<12} Fe[t]*C[q-3+t]             (doResize2, 4114-4328:
//   o[2q+1] = 0 + sum_{t<12} Fo[t]*C[q-2+t]              one bank phase, x = 0)
// with every index clamped to the producing array's range. Because edge
// replication makes Sext[u] = S[clamp(u)] constant outside the image, the
// clamped C equals FIR(Sext[m-6..m]) for EVERY m -- so no border special
// cases exist anywhere below, only clamped source loads.
//
// Structure ("marching strip"): a workgroup (2 waves) owns a strip of 64 output
// columns and a chunk of output rows, and marches down the source U2_RB rows
// at a time, two barriers per step:
//   S   U2_RB source rows x 50 px (halo 9+9): HBM -> LDS by LDS-DMA
//       (buffer_load_dwordx4 ... lds), no staging registers
//   H1  C rows  = FIR7 along x, whole pixels                     LDS -> LDS
//   H2  T rows  = 12-tap even/odd interpolation along x (one source pixel
//       = two whole output pixels per thread, one 13-pixel window of 128-bit
//       LDS reads)                                               LDS -> LDS
//   V   each thread owns ONE HALF-PIXEL output column (8 bytes) and keeps the
//       vertical FIR window (7) and the vertical interpolation window (13) in
//       REGISTERS, rotating through statically-indexed ring slots (the 16-row
//       ring period is fully unrolled), consuming one T row per step from LDS
//       and streaming two output rows per step straight to HBM (nt stores,
//       512 contiguous bytes per wave-instruction).
// Pipeline: interval B = {DMA of the next S tile in flight, H2 of this step};
// interval A = {V of this step with H1 of the NEXT step woven into it}.
// The horizontally resized intermediate (the reference's FltBuf) never exists
// in memory, vertical work is never recomputed inside a chunk, and all 56
// filter coefficients are wave-uniform SGPR pairs.
//
// Arithmetic contract: -ffp-contract=off, mul then add in the reference's tap
// order, interpolation sums started from +0.0f => bit-identical output.

#include "plan.h"
#include <algorithm>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <type_traits>

namespace avirhip {

typedef float f4 __attribute__(( ext_vector_type( 4 )));
typedef float f2 __attribute__(( ext_vector_type( 2 )));
typedef unsigned u2 __attribute__(( ext_vector_type( 2 )));

#ifndef U2_TW
#define U2_TW 64 // output columns per strip (32: one autonomous wave per strip)
#endif
#define U2_NT ( U2_TW * 2 ) // threads per workgroup: one per half pixel
#ifndef U2_RB
#define U2_RB 8   // source rows per marching step (8 or 16)
#endif
#define U2_SW ( U2_TW / 2 + 18 ) // S tile width (px)
#define U2_CW ( U2_TW / 2 + 12 ) // C tile width (px)

struct Up2Params
{
	const float* src; long src_ss; int sw, sh;
	int rmin, rmax; // source rows that exist behind `src` (a window: plan.h)
	float* dst; long dst_ss; int dst_row0; int nw, nh;
	int srow_lo, srow_hi;
	int nstrips, chunk0, nchunks, cq; // cq: source rows (output row pairs) per chunk
	// IO != 0: the caller's image instead of dst -- uint8 / uint16 (the output
	// stage of dither(), avir.h:4392-4419, without bit-depth truncation) or
	// float pixels of 1-3 channels; ibase = the band's first row
	void* ibase; int istride_b; int ich;
#ifdef U2_DBG
	int dbg; // timing ablations (debug build only)
	unsigned long long* clk; // [items][4]: shader cycles, start, end (100 MHz ticks), hw id
#endif
	const float* coef; // device: 2 axes x 64 floats, every tap stored twice:
		// [f 4x2 | fe 12x2 | fo 12x2 | pad 8]; horizontal axis first
};

// The 28 wave-uniform filter taps of one axis, each held as an aligned SGPR
// PAIR {c, c} so that v_pk_mul_f32 takes it directly as its 64-bit scalar
// operand (no s_mov to build pairs, no op_sel games). They are loaded per phase
// with a volatile scalar load: letting the compiler hoist both axes' taps out
// of the marching loop overflowed the SGPR file and spilled into VGPR lanes.
typedef float f8 __attribute__(( ext_vector_type( 8 )));
typedef float f16 __attribute__(( ext_vector_type( 16 )));

struct Taps
{
	f16 a, b, c; // f0..3 fe0..3 | fe4..11 | fo0..7   (as pairs)
	f8 d;        // fo8..11
	__device__ __forceinline__ f2 pr( const f16& v, int i ) const
		{ f2 r; r.x = v[ 2 * i ]; r.y = v[ 2 * i + 1 ]; return( r ); }
	__device__ __forceinline__ f2 f( int i ) const { return( pr( a, i )); }
	__device__ __forceinline__ f2 fe( int i ) const
		{ return( i < 4 ? pr( a, 4 + i ) : pr( b, i - 4 )); }
	__device__ __forceinline__ f2 fo( int i ) const
	{
		if( i < 8 ) return( pr( c, i ));
		f2 r; r.x = d[ 2 * ( i - 8 )]; r.y = d[ 2 * ( i - 8 ) + 1 ];
		return( r );
	}
};

// (forces a wave-uniform pointer into an SGPR pair: under register pressure
// the compiler may otherwise hand the asm a VGPR pair)
__device__ __forceinline__ const float* sgpr_ptr( const float* p )
{
	const unsigned long long a = (unsigned long long) p;
	const unsigned lo = __builtin_amdgcn_readfirstlane( (unsigned) a );
	const unsigned hi = __builtin_amdgcn_readfirstlane( (unsigned) ( a >> 32 ));
	return( (const float*) (( (unsigned long long) hi << 32 ) | lo ));
}

__device__ __forceinline__ Taps load_taps( const float* p0 )
{
	const float* const p = sgpr_ptr( p0 );
	Taps t;
	asm volatile( "s_load_dwordx16 %0, %4, 0x0\n\t"
		"s_load_dwordx16 %1, %4, 0x40\n\t"
		"s_load_dwordx16 %2, %4, 0x80\n\t"
		"s_load_dwordx8 %3, %4, 0xc0\n\t"
		AVIRHIP_WAITCNT_LGKM( 0 )
		: "=&s"( t.a ), "=&s"( t.b ), "=&s"( t.c ), "=&s"( t.d )
		: "s"( p ) : "memory" );
	return( t );
}

// f0..3 and fe0..11 only (the transposed vertical phase needs no fo)
struct TapsE
{
	f16 a, b;
	__device__ __forceinline__ f2 pr( const f16& v, int i ) const
		{ f2 r; r.x = v[ 2 * i ]; r.y = v[ 2 * i + 1 ]; return( r ); }
	__device__ __forceinline__ f2 f( int i ) const { return( pr( a, i )); }
	__device__ __forceinline__ f2 fe( int i ) const
		{ return( i < 4 ? pr( a, 4 + i ) : pr( b, i - 4 )); }
};

__device__ __forceinline__ TapsE load_taps_e( const float* p0 )
{
	const float* const p = sgpr_ptr( p0 );
	TapsE t;
	asm volatile( "s_load_dwordx16 %0, %2, 0x0\n\t"
		"s_load_dwordx16 %1, %2, 0x40\n\t"
		AVIRHIP_WAITCNT_LGKM( 0 )
		: "=&s"( t.a ), "=&s"( t.b ) : "s"( p ) : "memory" );
	return( t );
}

__device__ __forceinline__ f8 load_f8( const float* p0 )
{
	const float* const p = sgpr_ptr( p0 );
	f8 t;
	asm volatile( "s_load_dwordx8 %0, %1, 0x0\n\t" AVIRHIP_WAITCNT_LGKM( 0 )
		: "=&s"( t ) : "s"( p ) : "memory" );
	return( t );
}

// (wave-uniform tap pair) * (whole pixel) = two v_pk_mul_f32
__device__ __forceinline__ f4 tmul( const f2 t, const f4 v )
{
	const f2 lo = t * v.lo;
	const f2 hi = t * v.hi;
	return( __builtin_shufflevector( lo, hi, 0, 1, 2, 3 ));
}

// The thread index, recomputed from the lane id: per-lane addresses derived
// from it inside a phase need no registers while the other phases run (the
// compiler cannot see through the asm, so it cannot keep them alive either).
__device__ __forceinline__ int fresh_tid( const int wave )
{
	int l;
	asm volatile( "v_mbcnt_lo_u32_b32 %0, -1, 0\n\t"
		"v_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"( l ));
	return( l + wave * 64 );
}

__device__ __forceinline__ int clampi( int i, int hi )
{
	return( i < 0 ? 0 : ( i > hi ? hi : i ));
}

// With one wave per workgroup no s_barrier is needed: LDS operations of a wave
// execute in program order, only the compiler must not reorder them.
#if U2_TW == 32
#define U2_SYNC() __builtin_amdgcn_fence( __ATOMIC_RELEASE, "wavefront" ); \
	__builtin_amdgcn_wave_barrier(); \
	__builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "wavefront" )
#else
// (the S tile's LDS-DMA is a compiler builtin: the compiler's own wait-count
// insertion drains it before the barrier; `make paranoid` drains by hand too)
#define U2_SYNC() do { AVIRHIP_BARRIER_DRAIN(); __syncthreads(); } while( 0 )

// Wave priority by interval (round 6, NOTEBOOK 10.11). The vertical interval
// -- 430 packed multiply-adds per lane and step, the result stores -- runs one
// step above the horizontal one (LDS window reads, the next tile's DMA): the
// four waves of a SIMD belong to workgroups in different intervals, and the
// arbiter's oldest-first choice otherwise lets a wave that is about to wait for
// LDS anyway take issue slots from the one whose arithmetic paces the step.
// Same-box, same-process: cfg3 -4.5 %, cfg2 -1.0 ... -1.5 %; any "vertical above
// horizontal" pair measures the same, the reverse +0.4 ... -1.2 %. Results do
// not depend on it.
#define U2_PRIO_H 0
#define U2_PRIO_V 1

// s_setprio takes an immediate
__device__ __forceinline__ void u2_setprio( const int p )
{
	if( p == 0 ) __builtin_amdgcn_s_setprio( 0 );
	else if( p == 1 ) __builtin_amdgcn_s_setprio( 1 );
	else if( p == 2 ) __builtin_amdgcn_s_setprio( 2 );
	else __builtin_amdgcn_s_setprio( 3 );
}
#endif

#ifdef U2_DBG
#define U2_ON( bit ) ( !( P.dbg & ( bit )))
// Upper-bound ablations of the two levers VERDICT r5 #5 asks to be MEASURED
// (timing only: the results are wrong): bit 16 -- the third H1 iteration runs in
// wave 0 only, 2.5 iterations per wave and step instead of 3 (a 128-column
// strip's C tile is 608 px over 256 threads = 2.375: less H1 halo, and none of
// that build's costs -- 4-wave barriers, half the workgroups per CU); bit 32 --
// no horizontal phase at all in a chunk's last two marching steps (the T rows a
// chunk shares with the one below would come from an L2-resident scratch block
// instead of being recomputed: 16 of its 18 run-in rows, without the cost of
// writing and re-reading them).
#define U2_ABL_H1( k ) ( !( P.dbg & 16 ) || ( k ) < 2 || wave == 0 )
#define U2_ABL_TAIL() ( !( P.dbg & 32 ) || abl_tail == 0 )
#else
#define U2_ON( bit ) true
#define U2_ABL_H1( k ) true
#define U2_ABL_TAIL() true
#endif

#ifndef U2_STAUX
#define U2_STAUX 2 // destination stores are streaming: nt (measured -10 %)
#endif

#ifndef U2_LDAUX
#define U2_LDAUX 0 // cache policy of the source loads
#endif

#define U2_TPL ( U2_RB * ( U2_TW / 2 ) + 8 ) // px per T plane, incl. the 128-B skew

// VT ("vertical transposed"): when the bank phase is bit-symmetric (fo[t] ==
// fe[11-t], true for every parameter preset), the product fe[t]*C2[m] belongs to
// BOTH even(m+3-t) and odd(m-9+t). The vertical phase then keeps 24 running
// sums per thread instead of a 13-row window: each new C2 row costs 12
// multiplies + 24 adds instead of 24 + 24, every sum still receives its
// products in ascending tap order from a +0 start => same bits.
// IO (VT only): 0 the float RGBA result; 1 uint8, 2 uint16, 3 float pixels of
// P.ich channels stored by the vertical phase itself (a lane holds two
// channels of a pixel: one store per channel, lanes without one carry an
// out-of-range offset) -- no float result, no epilogue pass over it.
// SRC (VT only): 0 float RGBA pixels by 16-byte LDS-DMA; 10 * element size +
// channels (13, 14, 23, 24: RGB / RGBA uint8 / uint16) the caller's integer
// image as it lies -- packScanline's (float) cast and zero padding
// (avir.h:2777-2840) without a float copy of the source: a lane DMAs the two
// aligned dwords that hold its pixel into the first bytes of the pixel's own
// region of the S tile, and -- the same lane, after its wave's DMA has landed,
// before the interval's closing barrier -- reads them back, converts and writes
// the float pixel in place. P.src then points at bytes, P.src_ss counts bytes.
template< bool VT, int IO = 0, int SRC = 0 >
__global__ void __launch_bounds__( U2_NT )
__attribute__(( amdgpu_waves_per_eu( 4, 4 ))) k_up2( const Up2Params P )
{
	static_assert( IO == 0 || VT, "the fused output stage lives in the VT form" );
	static_assert( SRC == 0 || VT, "raw sources live in the VT form" );
	constexpr int SESZ = SRC / 10;       // bytes per source element (raw)
	constexpr int SCH = SRC % 10;        // channels of a source pixel (raw)
	constexpr int PXB = ( SRC != 0 ? SESZ * SCH : 16 ); // bytes per source pixel
	// (IO 4 / 5: uint8 / uint16 results of an INTEGER source -- finite and far
	// inside the int range, so the stage is add, convert, integer clamp)
	constexpr int ESZ = ( IO == 1 || IO == 4 ? 1 : ( IO == 2 || IO == 5 ? 2 : 4 ));
	// LDS tiles, whole pixels (16 B). The horizontal phases work on whole
	// pixels so that every LDS access is a conflict-free 128-bit one; the
	// vertical phase reads half pixels (8 B) of T.
	//   sS  U2_RB rows x U2_SW px, flat: exactly the lane order of the DMA
	//   sC  U2_RB rows x U2_CW px, flat
	//   sT  two planes (even / odd output column), U2_RB rows x U2_TW/2 px
	//       each; the planes sit 128 B (mod 256 B) apart so that the vertical
	//       phase's alternating-plane reads fall into disjoint banks
	__shared__ __attribute__(( aligned( 16 ))) f4 sS[ U2_RB * U2_SW ];
	__shared__ __attribute__(( aligned( 16 ))) f4 sC[ U2_RB * U2_CW ];
	__shared__ __attribute__(( aligned( 16 ))) f4 sT[ U2_TPL * 2 - 8 ];

#ifdef U2_DBG
	const unsigned long long dbg_c0 = clock64();
	const unsigned long long dbg_w0 = wall_clock64();

	// (bits 2048 / 4096: the workgroups of a CU start apart -- by half a
	// marching step in two groups, by eighths in eight -- instead of together)
	if( P.dbg & ( 2048 | 4096 ))
	{
		const int slot = ( P.dbg & 2048 ? (( blockIdx.x >> 8 ) & 1 ) * 4 :
			( blockIdx.x >> 8 ) & 7 );

		for( int i = 0; i < slot; i++ )
		{
			__builtin_amdgcn_s_sleep( 25 ); // 1,600 cycles: an eighth of a step
		}
	}
#endif
	// XCD-aware dealing of (strip, chunk) work items (see fused.hip)
	const int nwg = gridDim.x;
	const int b = blockIdx.x;
	const int xcd = b & 7;
	const int qd = nwg >> 3;
	const int rm = nwg & 7;
	const int item = ( xcd < rm ? xcd * ( qd + 1 ) :
		rm * ( qd + 1 ) + ( xcd - rm ) * qd ) + ( b >> 3 );

	// strip-major: an XCD's contiguous item range holds whole strips, i.e.
	// the same mix of full and short (last) chunks as every other XCD
	const int strip = item / P.nchunks;
	const int chunk = P.chunk0 + item - strip * P.nchunks;
	const int tid = threadIdx.x;
	const int hf = tid & 1;   // which half of the pixel (channels 0-1 / 2-3)
	const int col = tid >> 1; // column within the strip (vertical phase)

	const int qx0 = strip * ( U2_TW / 2 );
	const int qy0 = chunk * P.cq;
	const int qy1 = min( qy0 + P.cq, P.nh >> 1 );
	const int u0 = qy0 - 9;               // first Text row index consumed
	const int nsteps = qy1 - qy0 + 18;    // 6 preload + 12 warm-up + rows
	const int xo = qx0 * 2 + col;         // this thread's output column
	// rows this work item may store: its chunk, clipped to the band (uniform)
	const int ylo = max( qy0 * 2, P.srow_lo );
	const int yhi = min( qy1 * 2, P.srow_hi );
	const bool xok = ( xo < P.nw );
	const bool strip_full = ( qx0 * 2 + U2_TW <= P.nw );
	const int dlane_b = ( xo * 4 + hf * 2 ) * 4; // byte offset within a row
	const int drow_b = ( IO != 0 ? P.istride_b :
		(int) P.dst_ss * 4 );                // bytes per destination row
	const int srow_b = (int) P.src_ss * ( SRC != 0 ? 1 : 4 ); // bytes per source row

	// Buffer addressing (uniform 128-bit resource + 32-bit lane offset + scalar
	// row offset): a load/store moves ONE address dword per lane instead of a
	// 64-bit flat address -- store issue cost is proportional to the dwords a
	// wave sends to the memory pipeline. Both resources are based at this work
	// item's first row so every offset fits 31 bits for any image size.
	const int ubase = max( u0, P.rmin );
	// (raw sources: a dword-aligned base -- its two low bits go into the lanes'
	// offsets -- and the range the image's bytes end in, rounded up to a dword:
	// a pixel's second dword beyond it reads 0 without touching memory)
	const char* const sbase = (const char*) P.src + (long) ubase * srow_b;
	const int rsh = ( SRC != 0 ? (int) ( (unsigned long long) sbase & 3 ) : 0 );
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
		(void*) ( sbase - rsh ), 0, ( SRC != 0 ?
		( rsh + ( P.rmax - ubase ) * srow_b + P.sw * PXB + 3 ) & ~3 : 0x7fffffff ),
		0x00020000 );
	const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc(
		(void*) ( P.dst + (long) ( qy0 * 2 - P.dst_row0 ) * P.dst_ss ), 0,
		0x7fffffff, 0x00020000 );

	f2 tw[ 8 ];  // ring of the last T values of this column   (slot u & 7)
	f2 cw[ VT ? 1 : 16 ]; // ring of the last C2 values of this column (slot u & 15)
	f2 ea[ VT ? 12 : 1 ]; // VT: running sums of the even output rows (slot q % 12)
	f2 oa[ VT ? 12 : 1 ]; // VT: running sums of the odd output rows  (slot q % 12)

#pragma unroll
	for( int i = 0; i < 8; i++ ) tw[ i ] = (f2) 0.0f;
#pragma unroll
	for( int i = 0; i < ( VT ? 1 : 16 ); i++ ) cw[ i ] = (f2) 0.0f;
#pragma unroll
	for( int i = 0; i < ( VT ? 12 : 1 ); i++ ) ea[ i ] = oa[ i ] = (f2) 0.0f;

	// ---- VT: per-lane addresses, computed ONCE per work item and carried
	// through all phases in 6 registers (LDS byte addresses packed two to a
	// register, unpacked by one volatile instruction where a phase needs one:
	// opaque to the compiler, which would otherwise keep every derived address
	// alive next to the 64 registers of vertical state -- or, as before this
	// change, had to rebuild them from the lane id in every phase: 13 % of the
	// vector instructions of a marching step).
	constexpr int RPI0 = U2_NT / U2_SW;
	constexpr int CN0 = U2_RB * U2_CW;
	unsigned pkA = 0; // lo: H2 window start in sC; hi: this half pixel in sT
	unsigned pkB = 0; // H1 iterations 0 / 1: first source pixel in sS
	unsigned pkC = 0; // H1 iteration 2: source in sS / destination in sC
	unsigned t16 = 0; // thread index * 16 (flat pixel slot in sC / sT)
	int pvoff = -1;   // source byte offset of this lane's DMA pixel (< 0: none)
	int dlv = 0;      // destination byte offset within a row
	int dlv2 = 0;     // IO != 0: ... of the lane's second channel

	if constexpr( VT )
	{
		typedef __attribute__(( address_space( 3 ))) void* lp;
		auto la = []( const void* q ) -> unsigned
			{ return( (unsigned) (unsigned long long) (lp) q ); };

		{
			const int r = tid / U2_SW;
			const int sx = clampi( qx0 - 9 + ( tid - r * U2_SW ), P.sw - 1 );
			pvoff = ( tid < RPI0 * U2_SW ? r * srow_b + sx * PXB + rsh : -1 );
		}

		if constexpr( IO != 0 )
		{
			dlv = ( xo < P.nw && hf * 2 < P.ich ?
				( xo * P.ich + hf * 2 ) * ESZ : (int) 0x80000000u );
			dlv2 = ( xo < P.nw && hf * 2 + 1 < P.ich ?
				( xo * P.ich + hf * 2 + 1 ) * ESZ : (int) 0x80000000u );
		}
		else
		{
			dlv = ( xo < P.nw ? ( xo * 4 + hf * 2 ) * 4 : (int) 0x80000000u );
		}

		t16 = (unsigned) tid * 16;
		pkA = la( &sC[ ( tid / ( U2_TW / 2 )) * U2_CW +
			( tid & ( U2_TW / 2 - 1 ))]) |
			( la( (const f2*) sT + ((( col & 1 ) * U2_TPL + ( col >> 1 )) * 2 +
			hf )) << 16 );

		unsigned ha[ 3 ], hd = 0;
#pragma unroll
		for( int k = 0; k < 3; k++ )
		{
			const int i = min( tid + k * U2_NT, CN0 - 1 );
			const int r = (int) ((unsigned) i / U2_CW );
			ha[ k ] = la( &sS[ i + r * ( U2_SW - U2_CW )]);
			hd = la( &sC[ i ]);
		}

		pkB = ha[ 0 ] | ( ha[ 1 ] << 16 );
		pkC = ha[ 2 ] | ( hd << 16 );
		asm volatile( "" : "+v"( pkA ), "+v"( pkB ), "+v"( pkC ), "+v"( t16 ),
			"+v"( pvoff ), "+v"( dlv ));

		if constexpr( IO != 0 )
		{
			asm volatile( "" : "+v"( dlv2 ));
		}
	}

	typedef const __attribute__(( address_space( 3 ))) f4* lds_cf4;
	typedef __attribute__(( address_space( 3 ))) f4* lds_f4;
	typedef const __attribute__(( address_space( 3 ))) f2* lds_cf2;
	auto lo16 = []( const unsigned v ) -> unsigned
	{
		unsigned r;
		asm volatile( "v_and_b32 %0, 0xffff, %1" : "=v"( r ) : "v"( v ));
		return( r );
	};
	auto hi16 = []( const unsigned v ) -> unsigned
	{
		unsigned r;
		asm volatile( "v_lshrrev_b32 %0, 16, %1" : "=v"( r ) : "v"( v ));
		return( r );
	};

	// ---- source tiles: LDS-DMA (buffer_load_dwordx4 ... lds), HBM -> LDS
	// without passing through registers. One instruction moves whole pixels
	// of U2_RPI consecutive tile rows: lane f takes pixel f % U2_SW of row
	// f / U2_SW, and the hardware writes lane l of a wave to M0 + l * 16, i.e.
	// exactly the flat (row, px) order of sS. The per-lane source offset is
	// loop-invariant; the row group is selected by the scalar offset.
	constexpr int RPI = U2_NT / U2_SW;   // tile rows per DMA instruction
	constexpr int NDMA = U2_RB / RPI;
	static_assert( RPI >= 1 && U2_RB % RPI == 0, "S tile / DMA geometry" );

	typedef __attribute__(( address_space( 3 ))) void* ldsptr;
	const int wave = __builtin_amdgcn_readfirstlane( tid >> 6 );
	const int wave_b = wave * 1024;

	// raw sources: the byte offset (from rsrc's base) of this lane's pixel of
	// DMA instruction k of the tile that starts at row ub; < 0: the lane has none
	auto raw_off = [&]( const int k, const int ub ) -> int
	{
		if( ub >= P.rmin && ub + U2_RB <= P.rmax + 1 )
		{
			return( pvoff >= 0 ? pvoff + ( ub - ubase + k * RPI ) * srow_b : -1 );
		}

		// image top / bottom: rows clamp individually (rare path)
		const int tid = fresh_tid( wave );
		const int pr_r = tid / U2_SW;
		const int pr_sx = clampi( qx0 - 9 + ( tid - pr_r * U2_SW ), P.sw - 1 );
		const int sy = min( max( ub + k * RPI + pr_r, P.rmin ), P.rmax );

		return( tid < RPI * U2_SW ? ( sy - ubase ) * srow_b + pr_sx * PXB + rsh :
			-1 );
	};

	// raw sources: the tile's pixels, DMAed as pairs of dwords, become floats
	// (called by every thread after vmcnt( 0 ), before the closing barrier)
	auto land_raw = [&]( const int ub )
	{
		typedef const __attribute__(( address_space( 3 ))) unsigned* lds_cu;
		unsigned d0[ NDMA ], d1[ NDMA ];
		int off[ NDMA ];
		const int lw = ( threadIdx.x & 63 ) * 4;
#pragma unroll
		for( int k = 0; k < NDMA; k++ )
		{
			off[ k ] = raw_off( k, ub );
			const char* const st = (const char*) sS + k * RPI * U2_SW * 16 +
				wave_b + lw;
			d0[ k ] = *(lds_cu) (const void*) st;
			d1[ k ] = *(lds_cu) (const void*) ( st + 256 );
		}

		// (every read above before any write below: the float pixels go where
		// the wave's own dwords lay)
		asm volatile( "s_waitcnt lgkmcnt(0)" ::: "memory" );
#pragma unroll
		for( int k = 0; k < NDMA; k++ )
		{
			if( off[ k ] >= 0 )
			{
				const unsigned sh = (unsigned) off[ k ] & 3u;
				const unsigned lo = __builtin_amdgcn_alignbyte( d1[ k ], d0[ k ], sh );
				f4 px;

				if( SESZ == 1 )
				{
					px.x = (float) ( lo & 0xffu );
					px.y = (float) (( lo >> 8 ) & 0xffu );
					px.z = (float) (( lo >> 16 ) & 0xffu );
					px.w = ( SCH == 4 ? (float) ( lo >> 24 ) : 0.0f );
				}
				else
				{
					const unsigned hi = d1[ k ] >> ( sh * 8 );
					px.x = (float) ( lo & 0xffffu );
					px.y = (float) ( lo >> 16 );
					px.z = (float) ( hi & 0xffffu );
					px.w = ( SCH == 4 ? (float) ( hi >> 16 ) : 0.0f );
				}

				*(f4*) ((char*) sS + k * RPI * U2_SW * 16 + wave_b + lw * 4 ) = px;
			}
		}
	};

	auto prefetch = [&]( const int ub )
	{
		if constexpr( SRC != 0 )
		{
#pragma unroll
			for( int k = 0; k < NDMA; k++ )
			{
				const int o = raw_off( k, ub );

				if( o >= 0 )
				{
					const ldsptr q = (ldsptr) ((char*) sS + k * RPI * U2_SW * 16 +
						wave_b );
					__builtin_amdgcn_raw_ptr_buffer_load_lds( rsrc, q, 4, o & ~3,
						0, 0, U2_LDAUX );
					__builtin_amdgcn_raw_ptr_buffer_load_lds( rsrc,
						(ldsptr) ((char*) q + 256 ), 4, ( o & ~3 ) + 4, 0, 0,
						U2_LDAUX );
				}
			}

			return;
		}

		if( VT && ub >= P.rmin && ub + U2_RB <= P.rmax + 1 )
		{
			// (interior rows: the lane offset is the persistent one)
			if( pvoff >= 0 )
			{
#pragma unroll
				for( int k = 0; k < NDMA; k++ )
				{
					__builtin_amdgcn_raw_ptr_buffer_load_lds( rsrc,
						(ldsptr) ((char*) sS + k * RPI * U2_SW * 16 + wave_b ),
						16, pvoff, ( ub - ubase + k * RPI ) * srow_b, 0,
						U2_LDAUX );
				}
			}

			return;
		}

		const int tid = ( VT ? fresh_tid( wave ) : (int) threadIdx.x );
		const int pr_r = tid / U2_SW;
		const int pr_sx = clampi( qx0 - 9 + ( tid - pr_r * U2_SW ), P.sw - 1 );
		const bool pr_on = ( tid < RPI * U2_SW );
		const int pr_voff = pr_r * srow_b + pr_sx * 16;

		if( !pr_on )
		{
			return;
		}

		if( ub >= P.rmin && ub + U2_RB <= P.rmax + 1 )
		{
#pragma unroll
			for( int k = 0; k < NDMA; k++ )
			{
				__builtin_amdgcn_raw_ptr_buffer_load_lds( rsrc,
					(ldsptr) ((char*) sS + k * RPI * U2_SW * 16 + wave_b ), 16,
					pr_voff, ( ub - ubase + k * RPI ) * srow_b, 0, U2_LDAUX );
			}
		}
		else
		{
#pragma unroll
			for( int k = 0; k < NDMA; k++ )
			{
				// image top / bottom: rows clamp individually (rare path)
				// (the frame's first / last row -- or a window's: rows outside it
				// only feed results nobody stores, avirhip_band_source_rows)
				const int sy = min( max( ub + k * RPI + pr_r, P.rmin ), P.rmax );
				__builtin_amdgcn_raw_ptr_buffer_load_lds( rsrc,
					(ldsptr) ((char*) sS + k * RPI * U2_SW * 16 + wave_b ), 16,
					( sy - ubase ) * srow_b + pr_sx * 16, 0, 0, U2_LDAUX );
			}
		}
	};

	// ---- H1: C = FIR7 along x, one whole pixel per thread and iteration,
	// flat over the C tile (7 ds_read_b128 + 1 ds_write_b128 per pixel)
	constexpr int CN = U2_RB * U2_CW;
	constexpr int NH1 = ( CN + U2_NT - 1 ) / U2_NT;
	constexpr int H1EVERY = U2_RB / NH1; // woven into every H1EVERY-th V step
	static_assert( NH1 <= U2_RB, "H1 iterations must fit into the marching steps" );
	static_assert( !VT || ( NH1 == 3 && CN == CN0 && RPI == RPI0 ), "VT: packed H1 addresses" );

	auto h1_iter = [&]( const int k, const f2 ( &h )[ 4 ])
	{
		const int i = tid + k * U2_NT;

		if(( k + 1 ) * U2_NT <= CN || i < CN )
		{
			const int r = (int) ((unsigned) i / U2_CW );
			const f4* s = &sS[ i + r * ( U2_SW - U2_CW )];
			const f4 s0 = s[ 0 ], s1 = s[ 1 ], s2 = s[ 2 ], s3 = s[ 3 ];
			const f4 s4 = s[ 4 ], s5 = s[ 5 ], s6 = s[ 6 ];
			f4 a = tmul( h[ 0 ], s3 );
			a = a + tmul( h[ 1 ], s4 + s2 );
			a = a + tmul( h[ 2 ], s5 + s1 );
			a = a + tmul( h[ 3 ], s6 + s0 );
			sC[ i ] = a;
		}
	};

	// the VT form, without a lane condition (no control flow inside a vertical
	// step: basic-block boundaries let the compiler sink accumulation chains
	// across them; lanes past the tile's end redo its last pixel), addresses
	// from the persistent packed registers
	auto h1_iter_p = [&]( const int k, const f2 ( &h )[ 4 ])
	{
		const unsigned sa = ( k == 0 ? lo16( pkB ) : k == 1 ? hi16( pkB ) :
			lo16( pkC ));
		const unsigned da = ( k == 2 ? hi16( pkC ) : t16 + k * U2_NT * 16 +
			(unsigned) (unsigned long long) (ldsptr) sC );
		const lds_cf4 s = (lds_cf4) (unsigned long long) sa;
		const f4 s0 = s[ 0 ], s1 = s[ 1 ], s2 = s[ 2 ], s3 = s[ 3 ];
		const f4 s4 = s[ 4 ], s5 = s[ 5 ], s6 = s[ 6 ];
		f4 a = tmul( h[ 0 ], s3 );
		a = a + tmul( h[ 1 ], s4 + s2 );
		a = a + tmul( h[ 2 ], s5 + s1 );
		a = a + tmul( h[ 3 ], s6 + s0 );
		*(lds_f4) (unsigned long long) da = a;
	};

	// ---- H2: T[2q], T[2q+1] from the 13-px window C[qq .. qq+12], one source
	// pixel (two whole output pixels) per thread and iteration
	auto h2_phase = [&]()
	{
		if( VT && U2_ON( 2 ))
		{
			const Taps H = load_taps( P.coef );
			const lds_cf4 c0 = (lds_cf4) (unsigned long long) lo16( pkA );
			const lds_f4 t0 = (lds_f4) (unsigned long long)
				( t16 + (unsigned) (unsigned long long) (ldsptr) sT );
			constexpr int RS = U2_NT / ( U2_TW / 2 ); // rows per iteration

#pragma unroll
			for( int it = 0; it < U2_RB / RS; it++ )
			{
				const lds_cf4 c = c0 + it * RS * U2_CW;
				f4 e = (f4) 0.0f;
				f4 o = (f4) 0.0f;
				f4 w[ 13 ];

				// the window in three parts of 5 + 4 + 4 pixels: the
				// transposed vertical phase keeps 64 registers live
				// across this phase
#pragma unroll
				for( int t = 0; t < 5; t++ ) w[ t ] = c[ t ];
#pragma unroll
				for( int t = 0; t < 4; t++ )
				{
					e = e + tmul( H.fe( t ), w[ t ]);
					o = o + tmul( H.fo( t ), w[ t + 1 ]);
				}

				__builtin_amdgcn_sched_barrier( 0 );
#pragma unroll
				for( int t = 5; t < 9; t++ ) w[ t ] = c[ t ];
#pragma unroll
				for( int t = 4; t < 8; t++ )
				{
					e = e + tmul( H.fe( t ), w[ t ]);
					o = o + tmul( H.fo( t ), w[ t + 1 ]);
				}

				__builtin_amdgcn_sched_barrier( 0 );
#pragma unroll
				for( int t = 9; t < 13; t++ ) w[ t ] = c[ t ];
#pragma unroll
				for( int t = 8; t < 12; t++ )
				{
					e = e + tmul( H.fe( t ), w[ t ]);
					o = o + tmul( H.fo( t ), w[ t + 1 ]);
				}

				t0[ it * U2_NT ] = e;
				t0[ U2_TPL + it * U2_NT ] = o;
				__builtin_amdgcn_sched_barrier( 0 );
			}
		}
		else
		if( U2_ON( 2 ))
		{
			const Taps H = load_taps( P.coef );
			const int tid = ( VT ? fresh_tid( wave ) : (int) threadIdx.x );
			const int qq = tid & ( U2_TW / 2 - 1 );

			for( int r = tid / ( U2_TW / 2 ); r < U2_RB;
				r += U2_NT / ( U2_TW / 2 ))
			{
				const f4* c = &sC[ r * U2_CW + qq ];
				f4 e = (f4) 0.0f;
				f4 o = (f4) 0.0f;
				f4 w[ 13 ];

				if( VT )
				{
					// the window in three parts of 5 + 4 + 4 pixels: the
					// transposed vertical phase keeps 64 registers live
					// across this phase
#pragma unroll
					for( int t = 0; t < 5; t++ ) w[ t ] = c[ t ];
#pragma unroll
					for( int t = 0; t < 4; t++ )
					{
						e = e + tmul( H.fe( t ), w[ t ]);
						o = o + tmul( H.fo( t ), w[ t + 1 ]);
					}

					__builtin_amdgcn_sched_barrier( 0 );
#pragma unroll
					for( int t = 5; t < 9; t++ ) w[ t ] = c[ t ];
#pragma unroll
					for( int t = 4; t < 8; t++ )
					{
						e = e + tmul( H.fe( t ), w[ t ]);
						o = o + tmul( H.fo( t ), w[ t + 1 ]);
					}

					__builtin_amdgcn_sched_barrier( 0 );
#pragma unroll
					for( int t = 9; t < 13; t++ ) w[ t ] = c[ t ];
#pragma unroll
					for( int t = 8; t < 12; t++ )
					{
						e = e + tmul( H.fe( t ), w[ t ]);
						o = o + tmul( H.fo( t ), w[ t + 1 ]);
					}
				}
				else
				{
#pragma unroll
				for( int t = 0; t < 7; t++ ) w[ t ] = c[ t ];
#pragma unroll
				for( int t = 0; t < 6; t++ )
				{
					e = e + tmul( H.fe( t ), w[ t ]);
					o = o + tmul( H.fo( t ), w[ t + 1 ]);
				}

				// second half of the window only now: 13 pixels at once
				// would cost 52 VGPRs
				__builtin_amdgcn_sched_barrier( 0 );
#pragma unroll
				for( int t = 7; t < 13; t++ ) w[ t ] = c[ t ];
#pragma unroll
				for( int t = 6; t < 12; t++ )
				{
					e = e + tmul( H.fe( t ), w[ t ]);
					o = o + tmul( H.fo( t ), w[ t + 1 ]);
				}
				}

				sT[ r * ( U2_TW / 2 ) + qq ] = e;
				sT[ U2_TPL + r * ( U2_TW / 2 ) + qq ] = o;
			}
		}

	};

	// One marching step of U2_RB source rows. R0 = ring phase of its first
	// row (compile-time), so every ring access below is a fixed register.
	auto march = [&]( auto R0C, const int ub, const bool more,
		const bool more2 )
	{
		constexpr int R0 = decltype( R0C )::value;

		// ======== interval B: H2 of this step, while the S tile of the next
		// step lands in sS (nobody reads sS in this interval: H1 of this step
		// ran inside the previous vertical phase). The DMA is complete
		// (vmcnt) before the closing barrier.
		if( more && U2_ON( 8 ))
		{
			prefetch( ub + U2_RB );
		}

		u2_setprio( U2_PRIO_H );
		h2_phase();

		U2_SYNC();
		u2_setprio( U2_PRIO_V );

		// ======== interval A: V of this step, with H1 of the NEXT step woven
		// into it (its LDS latency hides behind the vertical VALU work)
		// ---- V: one half-column per thread, statically unrolled marching.
		// FAST: every row of this step is stored by every lane (interior of
		// the chunk, full strip): no per-row conditions at all.
		const Taps V = load_taps( P.coef + 64 );
		// horizontal FIR taps (4 pairs), parked in VGPRs: the vertical taps
		// already fill the SGPR file
		f2 hv[ 4 ];
		{
			const f8 HF = load_f8( P.coef );
#pragma unroll
			for( int t = 0; t < 4; t++ )
			{
				asm volatile( "v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"
					: "=&v"( hv[ t ].x ), "=&v"( hv[ t ].y )
					: "s"( HF[ t * 2 ]), "s"( HF[ t * 2 + 1 ]));
			}
		}
		const int yb = ( ub - 9 ) * 2; // output row pair of the step's first row
		int so = ( yb - qy0 * 2 ) * drow_b; // scalar byte offset of row yb

		// this thread's half pixel of T: plane = column parity
		const f2* sT2 = (const f2*) sT +
			((( col & 1 ) * U2_TPL + ( col >> 1 )) * 2 + hf );

		auto vsteps = [&]( auto FASTC )
		{
			constexpr bool FAST = decltype( FASTC )::value;
			tw[ R0 & 7 ] = sT2[ 0 ];
#pragma unroll
			for( int rr = 0; rr < U2_RB; rr++ )
			{
				const int g = R0 + rr; // ring phase of this row

				if( rr + 1 < U2_RB )
				{
					// ring slot (g+1)&7 holds T[u-7], outside the window
					tw[ ( g + 1 ) & 7 ] = sT2[ ( rr + 1 ) * U2_TW ];
				}

				f2 c2 = V.f( 0 ) * tw[ ( g - 3 ) & 7 ];
				c2 = c2 + V.f( 1 ) * ( tw[ ( g - 2 ) & 7 ] +
					tw[ ( g - 4 ) & 7 ]);
				c2 = c2 + V.f( 2 ) * ( tw[ ( g - 1 ) & 7 ] +
					tw[ ( g - 5 ) & 7 ]);
				c2 = c2 + V.f( 3 ) * ( tw[ g & 7 ] + tw[ ( g - 6 ) & 7 ]);
				cw[ g & 15 ] = c2;

				const int y0 = yb + rr * 2;

				if(( FAST || ( y0 + 1 >= ylo && y0 < yhi )) && U2_ON( 4 ))
				{
					f2 e = (f2) 0.0f;
					f2 o = (f2) 0.0f;
#pragma unroll
					for( int t = 0; t < 12; t++ )
					{
						// mul, mul, add, add: a dependent v_pk op needs one
						// issue slot of distance; pin the order so that slot
						// is the other accumulator's op and not an s_nop
						const f2 pe = V.fe( t ) * cw[ ( g - 12 + t ) & 15 ];
						const f2 po = V.fo( t ) * cw[ ( g - 11 + t ) & 15 ];
						__builtin_amdgcn_sched_barrier( 0 );
						e = e + pe;
						o = o + po;
						__builtin_amdgcn_sched_barrier( 0 );
					}

					// uniform row pointer + 32-bit lane byte offset
					if(( FAST || ( xok && y0 >= ylo )) &&
						( U2_ON( 1 ) || e.x == 1234.5f ))
					{
						__builtin_amdgcn_raw_buffer_store_b64(
							__builtin_bit_cast( u2, e ), rdst, dlane_b, so, U2_STAUX );
					}

					if(( FAST || ( xok && y0 + 1 < yhi )) &&
						( U2_ON( 1 ) || o.y == 1234.5f ))
					{
						__builtin_amdgcn_raw_buffer_store_b64(
							__builtin_bit_cast( u2, o ), rdst, dlane_b,
							so + drow_b, U2_STAUX );
					}
				}

				so += 2 * drow_b;

				// ---- H1 of the next step, one iteration
				if( rr % H1EVERY == 0 && rr / H1EVERY < NH1 && more &&
					U2_ON( 2 ))
				{
					h1_iter( rr / H1EVERY, hv );
				}
			}
		};

		if( strip_full && yb >= ylo && yb + 2 * U2_RB <= yhi )
		{
			vsteps( std::true_type() );
		}
		else
		{
			vsteps( std::false_type() );
		}

		U2_SYNC();
	};

	prefetch( u0 );

	if constexpr( SRC != 0 )
	{
		AVIRHIP_WAIT_VM( 0 );
		land_raw( u0 );
	}

	U2_SYNC();

	{
		// H1 of the first tile (later tiles: woven into the vertical phase)
		const f8 HF = load_f8( P.coef );
		f2 hp[ 4 ];
#pragma unroll
		for( int t = 0; t < 4; t++ )
		{
			hp[ t ].x = HF[ t * 2 ];
			hp[ t ].y = HF[ t * 2 + 1 ];
		}

#pragma unroll
		for( int k = 0; k < NH1; k++ )
		{
			h1_iter( k, hp );
		}
	}

	U2_SYNC();

	if constexpr( VT )
	{
		// ---- transposed vertical phase. Ring phase g of a row: tw slot g & 7,
		// running sums even( q ) / odd( q ) in slot q % 12 with q counted so that
		// the row of phase g carries C2 index m = g (mod 12): the product
		// fe[t]*C2[m] goes to even( m+3-t ) and odd( m-9+t ); even( m+3 ) and
		// odd( m+2 ) start on this row (0 + product), even( m-8 ) and
		// odd( m-9 ) are complete after it: output rows y0 + 2 and y0 + 1.
		// MODE 2: first 24 rows of the chunk (row j = g of the chunk: sums of
		// rows before the chunk are neither updated nor stored). One code
		// variant per ring phase: whether a row is stored is one scalar bit
		// test, and lanes right of the image carry an out-of-range buffer
		// offset (the hardware drops their stores).
		static_assert( !VT || U2_RB == 8, "VT marching period is 3 x 8 rows" );
#ifdef U2_DBG
		int abl_tail = 0; // (ablation bit 32: this step's woven H1 feeds a skipped H2)
#endif
		void* const dbase = ( IO != 0 ? (void*) ( (char*) P.ibase +
			(long) ( qy0 * 2 - P.dst_row0 ) * P.istride_b ) :
			(void*) ( P.dst + (long) ( qy0 * 2 - P.dst_row0 ) * P.dst_ss ));

		// IO != 0: a lane's two channels through the output stage (round
		// half up by truncation, clamp; the x86 build's cast beyond the int
		// range: plan.h) into the caller's image
		auto store_io = [&]( const f2 v, const __amdgpu_buffer_rsrc_t rs,
			const int soff )
		{
			if constexpr( IO == 3 )
			{
				// (element by element: __builtin_bit_cast of `v.y` stored v.x)
				const float v0 = v.x, v1 = v.y;
				__builtin_amdgcn_raw_buffer_store_b32( __float_as_uint( v0 ), rs,
					dlv, soff, U2_STAUX );
				__builtin_amdgcn_raw_buffer_store_b32( __float_as_uint( v1 ), rs,
					dlv2, soff, U2_STAUX );
			}
			else
			if constexpr( IO == 4 || IO == 5 )
			{
				// (float) (int) ( v + 0.5f ), clamped to [0, PK], cast: the same
				// number as the integer ( v + 0.5f ) truncates to, clamped --
				// for every |v| < 2^31, and an integer image's results are
				// within a few hundred of [0, PK]
				constexpr int PKI = ( IO == 4 ? 255 : 65535 );
				const int q0 = min( max( (int) ( v.x + 0.5f ), 0 ), PKI );
				const int q1 = min( max( (int) ( v.y + 0.5f ), 0 ), PKI );

				if constexpr( IO == 4 )
				{
					__builtin_amdgcn_raw_buffer_store_b8( (unsigned char) q0, rs,
						dlv, soff, U2_STAUX );
					__builtin_amdgcn_raw_buffer_store_b8( (unsigned char) q1, rs,
						dlv2, soff, U2_STAUX );
				}
				else
				{
					__builtin_amdgcn_raw_buffer_store_b16( (unsigned short) q0, rs,
						dlv, soff, U2_STAUX );
					__builtin_amdgcn_raw_buffer_store_b16( (unsigned short) q1, rs,
						dlv2, soff, U2_STAUX );
				}
			}
			else
			{
				constexpr float PK = ( IO == 1 ? 255.0f : 65535.0f );
				float t0 = (float) (int) ( v.x + 0.5f );
				float t1 = (float) (int) ( v.y + 0.5f );
				t0 = fminf( fmaxf( t0, 0.0f ), PK );
				t1 = fminf( fmaxf( t1, 0.0f ), PK );
				t0 = avirhip_x86_round_fix( v.x, t0, PK );
				t1 = avirhip_x86_round_fix( v.y, t1, PK );

				if constexpr( IO == 1 )
				{
					__builtin_amdgcn_raw_buffer_store_b8(
						(unsigned char) (unsigned) t0, rs, dlv, soff, U2_STAUX );
					__builtin_amdgcn_raw_buffer_store_b8(
						(unsigned char) (unsigned) t1, rs, dlv2, soff, U2_STAUX );
				}
				else
				{
					__builtin_amdgcn_raw_buffer_store_b16(
						(unsigned short) (unsigned) t0, rs, dlv, soff, U2_STAUX );
					__builtin_amdgcn_raw_buffer_store_b16(
						(unsigned short) (unsigned) t1, rs, dlv2, soff, U2_STAUX );
				}
			}
		};

		auto vstep_t = [&]( auto R0C, auto MODEC, const int ub,
			const bool more )
		{
			constexpr int R0 = decltype( R0C )::value;
			constexpr int MODE = decltype( MODEC )::value;
			const TapsE V = load_taps_e( P.coef + 64 );
			// the horizontal FIR taps stay in SGPRs too (16 + 4 pairs in all)
			f2 hv[ 4 ];
			{
				const f8 HF = load_f8( P.coef );
#pragma unroll
				for( int t = 0; t < 4; t++ )
				{
					hv[ t ].x = HF[ t * 2 ];
					hv[ t ].y = HF[ t * 2 + 1 ];
				}
			}

			const int yb = ( ub - 9 ) * 2;
			int so = ( yb - qy0 * 2 ) * drow_b; // scalar byte offset of row yb
			const int dlane_v = dlv;
			const lds_cf2 sT2 = (lds_cf2) (unsigned long long) hi16( pkA );
			// rows yb + 1 .. yb + 16 that lie in [ylo, yhi)
			const int rlo = min( max( ylo - ( yb + 1 ), 0 ), 2 * U2_RB );
			const int rhi = min( max( yhi - ( yb + 1 ), 0 ), 2 * U2_RB );
			const unsigned rmask = __builtin_amdgcn_readfirstlane(
				(( 1u << rhi ) - 1u ) & ~(( 1u << rlo ) - 1u ));

			tw[ R0 & 7 ] = sT2[ 0 ];
#pragma unroll
			for( int rr = 0; rr < U2_RB; rr++ )
			{
				const int g = R0 + rr; // ring phase of this row (0..23)

				if( rr + 1 < U2_RB )
				{
					tw[ ( g + 1 ) & 7 ] = sT2[ ( rr + 1 ) * U2_TW ];
				}

				f2 c2 = V.f( 0 ) * tw[ ( g - 3 ) & 7 ];
				c2 = c2 + V.f( 1 ) * ( tw[ ( g - 2 ) & 7 ] +
					tw[ ( g - 4 ) & 7 ]);
				c2 = c2 + V.f( 2 ) * ( tw[ ( g - 1 ) & 7 ] +
					tw[ ( g - 5 ) & 7 ]);
				c2 = c2 + V.f( 3 ) * ( tw[ g & 7 ] + tw[ ( g - 6 ) & 7 ]);

				if( U2_ON( 4 ))
				{
#pragma unroll
					for( int t = 0; t < 12; t += 2 )
					{
						// in the chunk's first rows (MODE 2, row j = g): even
						// sums wanted for t <= j - 6, odd sums for t >= 18 - j
						const bool we0 = ( MODE != 2 || t <= g - 6 );
						const bool wo0 = ( MODE != 2 || t >= 18 - g );
						const bool we1 = ( MODE != 2 || t + 1 <= g - 6 );
						const bool wo1 = ( MODE != 2 || t + 1 >= 18 - g );
						f2 p0, p1;

						if( we0 || wo0 ) p0 = V.fe( t ) * c2;
						if( we1 || wo1 ) p1 = V.fe( t + 1 ) * c2;

						__builtin_amdgcn_sched_barrier( 0 );

						if( we0 )
						{
							f2& a = ea[ ( g + 3 - t + 24 ) % 12 ];
							a = ( t == 0 ? (f2) 0.0f : a ) + p0;
						}

						if( wo0 )
						{
							f2& a = oa[ ( g + 3 + t ) % 12 ];
							a = a + p0;
						}

						if( we1 )
						{
							f2& a = ea[ ( g + 2 - t + 24 ) % 12 ];
							a = a + p1;
						}

						if( wo1 )
						{
							f2& a = oa[ ( g + 4 + t ) % 12 ];
							a = ( t + 1 == 11 ? (f2) 0.0f : a ) + p1;
						}

						__builtin_amdgcn_sched_barrier( 0 );
					}

					// pin the 24 sums here: the instruction-selection DAG is
					// free to postpone an add until its sum is next used (a
					// row later), keeping the product alive instead -- that
					// spilled ~300 registers
					asm volatile( "" : "+v"( ea[ 0 ]), "+v"( ea[ 1 ]),
						"+v"( ea[ 2 ]), "+v"( ea[ 3 ]), "+v"( ea[ 4 ]),
						"+v"( ea[ 5 ]), "+v"( ea[ 6 ]), "+v"( ea[ 7 ]),
						"+v"( ea[ 8 ]), "+v"( ea[ 9 ]), "+v"( ea[ 10 ]),
						"+v"( ea[ 11 ]));
					asm volatile( "" : "+v"( oa[ 0 ]), "+v"( oa[ 1 ]),
						"+v"( oa[ 2 ]), "+v"( oa[ 3 ]), "+v"( oa[ 4 ]),
						"+v"( oa[ 5 ]), "+v"( oa[ 6 ]), "+v"( oa[ 7 ]),
						"+v"( oa[ 8 ]), "+v"( oa[ 9 ]), "+v"( oa[ 10 ]),
						"+v"( oa[ 11 ]));

					const f2 od = oa[ ( g + 3 ) % 12 ]; // odd( m-9 ): row y0 + 1
					const f2 ed = ea[ ( g + 4 ) % 12 ]; // even( m-8 ): row y0 + 2

					if( MODE != 2 || g >= 17 )
					{
						// bit k of rmask: row yb + 1 + k is stored by this item.
						// No branches (they would let the compiler sink whole
						// accumulation chains into them): an unwanted row goes
						// through a zero-length view of the destination, whose
						// range check drops the store.
						if constexpr( IO != 0 )
						{
							store_io( od, __builtin_amdgcn_make_buffer_rsrc(
								dbase, 0, ( rmask & ( 1u << ( 2 * rr )) ?
								0x7fffffff : 0 ), 0x00020000 ), so + drow_b );
							store_io( ed, __builtin_amdgcn_make_buffer_rsrc(
								dbase, 0, ( rmask & ( 2u << ( 2 * rr )) ?
								0x7fffffff : 0 ), 0x00020000 ), so + 2 * drow_b );
						}
						else
						{
						if( U2_ON( 1 ) || od.x == 1234.5f )
						{
							__builtin_amdgcn_raw_buffer_store_b64(
								__builtin_bit_cast( u2, od ),
								__builtin_amdgcn_make_buffer_rsrc( dbase, 0,
								( rmask & ( 1u << ( 2 * rr )) ? 0x7fffffff : 0 ),
								0x00020000 ), dlane_v, so + drow_b, U2_STAUX );
						}

						if( U2_ON( 1 ) || ed.y == 1234.5f )
						{
							__builtin_amdgcn_raw_buffer_store_b64(
								__builtin_bit_cast( u2, ed ),
								__builtin_amdgcn_make_buffer_rsrc( dbase, 0,
								( rmask & ( 2u << ( 2 * rr )) ? 0x7fffffff : 0 ),
								0x00020000 ), dlane_v, so + 2 * drow_b, U2_STAUX );
						}
						}
					}
				}

				so += 2 * drow_b;

				// ---- H1 of the next step, one iteration
				// (unconditional: after the chunk's last step it works on
				// stale tiles and nobody reads its result)
				if( rr % H1EVERY == 0 && rr / H1EVERY < NH1 && U2_ON( 2 ) &&
					U2_ABL_H1( rr / H1EVERY ) && U2_ABL_TAIL())
				{
					h1_iter_p( rr / H1EVERY, hv );
				}
			}
		};

		typedef std::integral_constant< int, 0 > I0;
		typedef std::integral_constant< int, 1 > I1;
		typedef std::integral_constant< int, 2 > I2;
		typedef std::integral_constant< int, 8 > I8;
		typedef std::integral_constant< int, 16 > I16;

		// one marching step: interval B = {DMA of the next S tile, H2 of this
		// step}, interval A = {V of this step, H1 of the next} (see march)
		auto march_t = [&]( auto R0C, auto RAMPC, const int sb )
		{
			const int ub = u0 + sb;
			const bool more = ( sb + U2_RB < nsteps );
#ifdef U2_DBG
			// (bit 32: H2 of the chunk's last two steps and the H1 that feeds
			// them -- woven into the step before -- are skipped)
			const bool tail_h2 = (( P.dbg & 32 ) && sb + 2 * U2_RB >= nsteps &&
				nsteps > 6 * U2_RB );
			abl_tail = (( P.dbg & 32 ) && sb + 3 * U2_RB >= nsteps &&
				sb + U2_RB < nsteps && nsteps > 6 * U2_RB ? 1 : 0 );
#endif

			if( more && U2_ON( 8 ))
			{
				prefetch( ub + U2_RB );
			}

#ifdef U2_DBG
			// (bit 1024: wave priority by interval from the bits -- timing
			// experiments, results unchanged: bits 6-7 the priority of the
			// horizontal interval, bits 8-9 that of the vertical one)
			u2_setprio( P.dbg & 1024 ? ( P.dbg >> 6 ) & 3 : U2_PRIO_H );

			if( !tail_h2 )
#else
			u2_setprio( U2_PRIO_H );
#endif
			h2_phase();

			if constexpr( SRC != 0 )
			{
				if( more )
				{
					AVIRHIP_WAIT_VM( 0 );
					land_raw( ub + U2_RB );
				}
			}

			U2_SYNC();
#ifdef U2_DBG
			u2_setprio( P.dbg & 1024 ? ( P.dbg >> 8 ) & 3 : U2_PRIO_V );
#else
			u2_setprio( U2_PRIO_V );
#endif

			if( decltype( RAMPC )::value )
			{
				vstep_t( R0C, I2(), ub, more );
			}
			else
			{
				vstep_t( R0C, I1(), ub, more );
			}

			U2_SYNC();
		};

		// the chunk's first 24 rows (nsteps >= 24: cq >= 6), then the period
		// of 24 rows = 3 steps per trip
		march_t( I0(), std::true_type(), 0 );
		march_t( I8(), std::true_type(), 8 );
		march_t( I16(), std::true_type(), 16 );

		for( int sb = 24; sb < nsteps; sb += 24 )
		{
			march_t( I0(), std::false_type(), sb );

			if( sb + 8 >= nsteps )
			{
				break;
			}

			march_t( I8(), std::false_type(), sb + 8 );

			if( sb + 16 >= nsteps )
			{
				break;
			}

			march_t( I16(), std::false_type(), sb + 16 );
		}
	}
	else
	{
		// the ring period is 16 rows: 16 / U2_RB marching steps per trip
		for( int sb = 0; sb < nsteps; sb += 16 )
		{
#if U2_RB == 16
			march( std::integral_constant< int, 0 >(), u0 + sb, sb + 16 < nsteps,
				sb + 32 < nsteps );
#else
			march( std::integral_constant< int, 0 >(), u0 + sb, sb + 8 < nsteps,
				sb + 16 < nsteps );

			if( sb + 8 >= nsteps )
			{
				break;
			}

			march( std::integral_constant< int, 8 >(), u0 + sb + 8,
				sb + 16 < nsteps, sb + 24 < nsteps );
#endif
		}
	}

#ifdef U2_DBG
	if( tid == 0 && P.clk != nullptr )
	{
		P.clk[ blockIdx.x * 4 ] = clock64() - dbg_c0;
		P.clk[ blockIdx.x * 4 + 1 ] = dbg_w0;
		P.clk[ blockIdx.x * 4 + 2 ] = wall_clock64();
		P.clk[ blockIdx.x * 4 + 3 ] =
			(unsigned) __builtin_amdgcn_s_getreg(( 31 << 11 ) | 4 ) |
			((unsigned long long) __builtin_amdgcn_s_getreg(( 31 << 11 ) | 20 ) << 32 );
	}
#endif
}

// ---------------------------------------------------------------------
// host side: recognise the plan shape and extract the 28 taps per axis
// ---------------------------------------------------------------------

struct Up2Axis
{
	float f[ 4 ], fe[ 12 ], fo[ 12 ];
};

static bool match_axis( const LAxis& ax, Up2Axis& A )
{
	if( ax.ops.size() != 2 || ax.dst_len != ax.src_len * 2 )
	{
		return( false );
	}

	const LOp& a = ax.ops[ 0 ];
	const LOp& g = ax.ops[ 1 ];

	if( a.type != OP_FIR || a.view != VIEW_CLAMP || a.rf != 1 || a.lat != 3 ||
		a.e != 3 || a.in_len != ax.src_len || a.out_len != ax.src_len + 6 ||
		g.type != OP_GATHER || g.view != VIEW_ZS || g.maxtaps != 12 ||
		g.in_len != a.out_len || g.out_len != ax.dst_len )
	{
		return( false );
	}

	for( int i = 0; i < 4; i++ )
	{
		A.f[ i ] = a.h_flt[ i ];
	}

	for( int t = 0; t < 12; t++ )
	{
		A.fe[ t ] = g.h_coef[ t ];
		A.fo[ t ] = g.h_coef[ 12 + t ];
	}

	for( int j = 0; j < g.out_len; j++ )
	{
		const int q = j >> 1;
		const int st = ( j & 1 ? q - 2 : q - 3 );
		const float* cf = &g.h_coef[ (size_t) j * 12 ];
		const float* rf = ( j & 1 ? A.fo : A.fe );

		if( g.h_start[ j ] != st || g.h_ntaps[ j ] != 12 ||
			st + 11 > g.zs_mmax || memcmp( cf, rf, 12 * sizeof( float )) != 0 )
		{
			return( false );
		}
	}

	return( true );
}

struct Up2Data
{
	Up2Axis h, v;
	float* d_coef;
	bool vsym; // vertical bank phase bit-symmetric: fo[t] == fe[11-t]
};

int up2_prepare( avirhip_plan* p )
{
	p -> up2 = nullptr;

	if( p -> is_lancir || p -> ch != 4 )
	{
		return( AVIRHIP_OK );
	}

	Up2Data* D = new Up2Data();

	if( match_axis( p -> h, D -> h ) && match_axis( p -> v, D -> v ))
	{
		float hc[ 128 ] = { 0 };
		const Up2Axis* ax[ 2 ] = { &D -> h, &D -> v };

		for( int a = 0; a < 2; a++ )
		{
			float* o = hc + a * 64;

			for( int i = 0; i < 4; i++ )
			{
				o[ 2 * i ] = o[ 2 * i + 1 ] = ax[ a ] -> f[ i ];
			}

			for( int t = 0; t < 12; t++ )
			{
				o[ 8 + 2 * t ] = o[ 9 + 2 * t ] = ax[ a ] -> fe[ t ];
				o[ 32 + 2 * t ] = o[ 33 + 2 * t ] = ax[ a ] -> fo[ t ];
			}
		}

		void* q = nullptr;
		AVIRHIP_HIPCHECK( hipMalloc( &q, sizeof( hc )));
		p -> allocs.push_back( q );
		AVIRHIP_HIPCHECK( hipMemcpy( q, hc, sizeof( hc ),
			hipMemcpyHostToDevice ));
		D -> d_coef = (float*) q;
		D -> vsym = true;

		for( int t = 0; t < 12; t++ )
		{
			D -> vsym &= ( memcmp( &D -> v.fo[ t ], &D -> v.fe[ 11 - t ],
				sizeof( float )) == 0 );
		}

		p -> up2 = D;
		p -> fused_ok |= 4;
		p -> auto_path = 4;
	}
	else
	{
		delete D;
	}

	return( AVIRHIP_OK );
}

void up2_release( avirhip_plan* p )
{
	delete (Up2Data*) p -> up2;
	p -> up2 = nullptr;
}

// Whether the plan's marching kernel can store the caller's pixels itself
// (up2_run's `iout`): the transposed vertical phase, default ditherer, no
// gamma, and uint8 / uint16 without bit-depth truncation or float pixels of
// 1-3 channels.
bool up2_stores_io( const avirhip_plan* p )
{
	const Up2Data* D = (const Up2Data*) p -> up2;
	static const bool novt = ( getenv( "AVIRHIP_UP2_NOVT" ) != nullptr );

	return( D != nullptr && U2_RB == 8 && D -> vsym && !novt &&
		!( p -> variant & AVIRHIP_VARIANT_UP2_PLAIN_V ) && !p -> gamma &&
		p -> dither == AVIRHIP_DITHER_DEF &&
		(( p -> out_type == AVIRHIP_U8 && p -> tr_mul == 1.0 &&
		p -> pk_out == 255.0 ) ||
		( p -> out_type == AVIRHIP_U16 && p -> tr_mul == 1.0 &&
		p -> pk_out == 65535.0 ) ||
		( p -> out_type == AVIRHIP_F32 && p -> io_ch < 4 )));
}

// The float RGBA call up2_run cannot refuse (its own checks, io == 0): a source
// window may then be handed over as a virtual frame base.
bool up2_takes_window( const avirhip_plan* p, const void* src, const void* dst )
{
	return( p -> up2 != nullptr && !p -> is_lancir && p -> ch == 4 &&
		p -> io_ch == 4 && p -> in_type == AVIRHIP_F32 &&
		p -> out_type == AVIRHIP_F32 && !p -> gamma &&
		( (uintptr_t) src & 15 ) == 0 && ( (uintptr_t) dst & 7 ) == 0 &&
		( p -> src_stride & 3 ) == 0 &&
		(long) p -> src_stride * 4 < ( 1L << 22 ) &&
		(long) p -> new_w * 16 < ( 1L << 22 ));
}

// `iout` != nullptr: the caller's image (the band's first row) -- uint8 /
// uint16 without bit-depth truncation, or float pixels of 1-3 channels: the
// vertical phase converts and stores there (no float result, no epilogue
// pass); returns 1 if this plan's form of the kernel cannot (the caller then
// runs with a float result).
// `raw` != nullptr (with `iout`): the caller's integer image (uint8 / uint16,
// 3 or 4 channels, `raw_stride` elements per row) instead of `src`; returns 1
// if this call cannot take it (the caller then packs and calls again).
int up2_run( avirhip_plan* p, const float* src, long src_stride, float* dst,
	int row0, int row1, hipStream_t st, void* iout, const void* raw,
	long raw_stride )
{
	const Up2Data* D = (const Up2Data*) p -> up2;
	static const bool novt = ( getenv( "AVIRHIP_UP2_NOVT" ) != nullptr );
	int io = 0;

	if( iout != nullptr )
	{
		const bool vt = ( D != nullptr && U2_RB == 8 && D -> vsym && !novt &&
			!( p -> variant & AVIRHIP_VARIANT_UP2_PLAIN_V ));

		io = ( p -> out_type == AVIRHIP_U8 && p -> tr_mul == 1.0 &&
			p -> pk_out == 255.0 ? 1 :
			( p -> out_type == AVIRHIP_U16 && p -> tr_mul == 1.0 &&
			p -> pk_out == 65535.0 && ( (uintptr_t) iout & 1 ) == 0 ? 2 :
			( p -> out_type == AVIRHIP_F32 && p -> io_ch < 4 &&
			( (uintptr_t) iout & 3 ) == 0 ? 3 : 0 )));

		if( !vt || io == 0 || p -> dither != AVIRHIP_DITHER_DEF || p -> gamma )
		{
			return( 1 );
		}

		// an integer image in (the pack pass' float copy of it): every sum
		// is finite and small
		if(( io == 1 || io == 2 ) && ( p -> in_type == AVIRHIP_U8 ||
			p -> in_type == AVIRHIP_U16 ))
		{
			io += 3;
		}

	}

	// the integer image as it lies (k_up2< true, IO, SRC >): results of
	// integer sources only (io 4 / 5), whole frames behind the pointer, every
	// pixel inside two aligned dwords (uint16 RGBA: dword-aligned rows)
	int srck = 0;

	if( raw != nullptr )
	{
		const int esz = ( p -> in_type == AVIRHIP_U8 ? 1 : 2 );
		const long sb = raw_stride * esz;

		if(( io != 4 && io != 5 ) || ( p -> io_ch != 3 && p -> io_ch != 4 ) ||
			p -> win_rows > 0 || ( (uintptr_t) raw & ( esz - 1 )) ||
			( esz == 2 && p -> io_ch == 4 && (( (uintptr_t) raw | sb ) & 3 )) ||
			sb >= ( 1L << 22 ) || getenv( "AVIRHIP_UP2_NO_RAW" ) != nullptr )
		{
			return( 1 );
		}

		srck = esz * 10 + p -> io_ch;
		src = (const float*) raw;
		src_stride = sb;
	}

	// source pixels travel as 16-byte LDS-DMA pieces, destination half-pixels
	// as 8-byte stores
	if( D == nullptr || ( srck == 0 && (( (uintptr_t) src & 15 ) ||
		( src_stride & 3 ))) || ( io == 0 && ( (uintptr_t) dst & 7 )))
	{
		return( 1 );
	}

	if( row1 <= row0 )
	{
		return( AVIRHIP_OK );
	}

	// the kernel addresses rows with 32-bit float offsets
	if(( srck == 0 && (long) src_stride * 4 >= ( 1L << 22 )) ||
		(long) p -> new_w * 16 >= ( 1L << 22 ))
	{
		return( 1 );
	}

	Up2Params P;
	P.src = src; P.src_ss = src_stride; P.sw = p -> src_w; P.sh = p -> src_h;
	P.rmin = ( p -> win_rows > 0 ? p -> win_first : 0 );
	P.rmax = ( p -> win_rows > 0 ? p -> win_first + p -> win_rows : p -> src_h ) - 1;
	P.dst = dst; P.dst_ss = (long) p -> new_w * 4; P.dst_row0 = row0;
	P.nw = p -> new_w; P.nh = p -> new_h;
	P.srow_lo = row0; P.srow_hi = row1;
	P.nstrips = ( p -> new_w + U2_TW - 1 ) / U2_TW;
	P.ibase = iout; P.ich = p -> io_ch;
	P.istride_b = p -> new_w * p -> io_ch * ( io == 1 || io == 4 ? 1 :
		( io == 2 || io == 5 ? 2 : 4 ));

	// Chunk height. A chunk of cq source rows costs (cq + 18) / U2_RB marching
	// steps (6 rows of preload + 12 of warm-up), so cq = U2_RB*k - 18 wastes
	// nothing. The kernel is VALU-bound and a CU shares its VALUs among its
	// resident workgroups (up to 8), so a launch whose items are all resident
	// at once takes about
	//     m * k / thr( m ),   m = ceil( items / 256 CUs )
	// step-times -- the most loaded CU finishes last, and thr is the measured
	// VALU throughput of a CU running m workgroups (cq sweeps of cfg2 and cfg3,
	// profiles/r01_up2_chunk_sweep.txt). Few tall chunks waste little warm-up
	// work, many short ones balance and fill the CUs; launches of more than
	// 8 x 256 items run in partial rounds and measured ~6 % worse than their
	// step count.
	int cq = 0;
	{
		static const double thr[ 9 ] = { 1.0, 0.3, 0.57, 0.8, 0.9, 0.9, 0.9,
			0.97, 1.0 };
		const int rows = ( row1 - row0 + 1 ) / 2 + 1; // source rows of the band
		double best = -1.0;

		for( int k = 10; k <= 64; k++ )
		{
			const int c = U2_RB * k - 18;
			const long nch = ( rows + c - 1 ) / c;
			const long items = nch * P.nstrips;
			const long m = ( items + 255 ) / 256;
			const double cost = ( m <= 8 ? (double) m * k / thr[ m ] :
				(double) items * k / 256.0 * 1.06 );

			if( best < 0.0 || cost < best )
			{
				best = cost;
				cq = c;
			}

			if( nch == 1 )
			{
				break;
			}
		}
	}

	const char* ecq = getenv( "AVIRHIP_UP2_CQ" );

	if( ecq != nullptr && atoi( ecq ) >= 6 )
	{
		cq = (( atoi( ecq ) + 18 + U2_RB - 1 ) / U2_RB ) * U2_RB - 18;
	}

	P.cq = cq;
	const int cr = P.cq * 2;
	P.chunk0 = row0 / cr;
	const int chunk1 = ( row1 - 1 ) / cr;

	P.coef = D -> d_coef;
#ifdef U2_DBG
	P.dbg = ( getenv( "AVIRHIP_UP2_DBG" ) != nullptr ?
		atoi( getenv( "AVIRHIP_UP2_DBG" )) : 0 );
	static unsigned long long* dclk = nullptr;
	static int dclk_calls = 0;

	if( dclk == nullptr )
	{
		(void) hipMalloc( &dclk, 8192 * 32 );
	}

	P.clk = dclk;
#endif

	P.nchunks = chunk1 - P.chunk0 + 1;
	const int items = P.nstrips * P.nchunks;

	if( getenv( "AVIRHIP_VERBOSE" ) != nullptr )
	{
		int nb = -1;
		(void) hipOccupancyMaxActiveBlocksPerMultiprocessor( &nb,
			(const void*) k_up2< false, 0 >, U2_NT, 0 );
		fprintf( stderr, "k_up2: %d items (strips %d, cq %d), occupancy API "
			"%d workgroups/CU\n", items, P.nstrips, P.cq, nb );
	}

	const char* pad = getenv( "AVIRHIP_UP2_LDSPAD" ); // residency experiments
	const int ldspad = ( pad != nullptr ? atoi( pad ) : 0 );
	if( srck != 0 )
	{
#define U2_RAW( IOK, SK ) hipLaunchKernelGGL(( k_up2< U2_RB == 8, \
	U2_RB == 8 ? IOK : 0, U2_RB == 8 ? SK : 0 > ), dim3( items ), \
	dim3( U2_NT ), ldspad, st, P )
		if( io == 4 )
		{
			switch( srck )
			{
				case 13: U2_RAW( 4, 13 ); break;
				case 14: U2_RAW( 4, 14 ); break;
				case 23: U2_RAW( 4, 23 ); break;
				default: U2_RAW( 4, 24 ); break;
			}
		}
		else
		{
			switch( srck )
			{
				case 13: U2_RAW( 5, 13 ); break;
				case 14: U2_RAW( 5, 14 ); break;
				case 23: U2_RAW( 5, 23 ); break;
				default: U2_RAW( 5, 24 ); break;
			}
		}
#undef U2_RAW
	}
	else
	if( io != 0 )
	{
		switch( io )
		{
			case 1: hipLaunchKernelGGL(( k_up2< U2_RB == 8, U2_RB == 8 ? 1 : 0 > ),
				dim3( items ), dim3( U2_NT ), ldspad, st, P ); break;
			case 2: hipLaunchKernelGGL(( k_up2< U2_RB == 8, U2_RB == 8 ? 2 : 0 > ),
				dim3( items ), dim3( U2_NT ), ldspad, st, P ); break;
			case 3: hipLaunchKernelGGL(( k_up2< U2_RB == 8, U2_RB == 8 ? 3 : 0 > ),
				dim3( items ), dim3( U2_NT ), ldspad, st, P ); break;
			case 4: hipLaunchKernelGGL(( k_up2< U2_RB == 8, U2_RB == 8 ? 4 : 0 > ),
				dim3( items ), dim3( U2_NT ), ldspad, st, P ); break;
			default: hipLaunchKernelGGL(( k_up2< U2_RB == 8, U2_RB == 8 ? 5 : 0 > ),
				dim3( items ), dim3( U2_NT ), ldspad, st, P ); break;
		}
	}
	else
	if( U2_RB == 8 && D -> vsym && !novt &&
		!( p -> variant & AVIRHIP_VARIANT_UP2_PLAIN_V ))
	{
		hipLaunchKernelGGL(( k_up2< U2_RB == 8, 0 > ), dim3( items ), dim3( U2_NT ),
			ldspad, st, P );
	}
	else
	{
		hipLaunchKernelGGL(( k_up2< false, 0 > ), dim3( items ), dim3( U2_NT ),
			ldspad, st, P );
	}
	AVIRHIP_HIPCHECK( hipGetLastError() );
#ifdef U2_DBG
	if( ++dclk_calls == 60 && items <= 8192 )
	{
		// in-kernel clocks of a launch in the middle of a timing loop
		std::vector< unsigned long long > h( items * 4 );
		(void) hipStreamSynchronize( st );
		(void) hipMemcpy( h.data(), dclk, items * 32, hipMemcpyDeviceToHost );
		double sc = 0.0, sw = 0.0, mx = 0.0;
		unsigned long long t0 = ~0ULL, t1 = 0;

		for( int i = 0; i < items; i++ )
		{
			const double w = (double) ( h[ i * 4 + 2 ] - h[ i * 4 + 1 ]);
			sc += (double) h[ i * 4 ];
			sw += w;
			mx = std::max( mx, w );
			t0 = std::min( t0, h[ i * 4 + 1 ]);
			t1 = std::max( t1, h[ i * 4 + 2 ]);
		}

		fprintf( stderr, "k_up2 clocks: mean item %.0f shader cycles, %.2f us "
			"(max %.2f us, first start to last end %.2f us) => %.3f GHz\n",
			sc / items, sw / items / 100.0, mx / 100.0,
			(double) ( t1 - t0 ) / 100.0, sc / sw * 0.1 );

		const char* dump = getenv( "AVIRHIP_UP2_CLKDUMP" );

		if( dump != nullptr )
		{
			FILE* f = fopen( dump, "w" );

			for( int i = 0; f != nullptr && i < items; i++ )
			{
				fprintf( f, "%d %llu %llu %llu %llx\n", i, h[ i * 4 ],
					h[ i * 4 + 1 ] - t0, h[ i * 4 + 2 ] - t0, h[ i * 4 + 3 ]);
			}

			if( f != nullptr ) fclose( f );
		}
	}
#endif
	return( AVIRHIP_OK );
}

} // namespace avirhip
