// gpass.hip -- pass kernels for general (non-2x) ratios: ONE AXIS PER LAUNCH,
// autonomous wavefronts, every dynamically indexed window in LDS rings.
//
// A lowered axis (plan.h) of the common plans is one of
//   PRE   FIR7 then GATHER     upsizing: correction filter, then the 12-tap
//                              interpolation over the zero-stuffed view
//                              (doFilter 3748 + doUpsample 3260 + doResize2 4114)
//   POST  GATHER then FIR7     downsizing: NT-tap resize, then the correction
//                              filter (doResize 3884 / doResize2 + doFilter)
//   NONE  GATHER               LANCIR's N-tap dot products (lancir.h:2466-2544)
// with a position table (first input index per output) and one coefficient row
// per output (AVIR) or per fractional phase (LANCIR). The tiled interpreter of
// fused.hip runs any chain but keeps whole 2-D tiles in LDS: two workgroups
// per CU, phases separated by barriers, half of its wave-cycles parked
// (profiles/r02_tile). Here a wavefront owns a strip and never synchronises
// with another wave:
//
//   k_gv  vertical pass. Lane = whole pixel of a 64-pixel strip (half pixel /
//         32-pixel strips for images narrower than 49 pixels): a row is one
//         contiguous 1-KiB segment. Source rows arrive by LDS-DMA in a landing
//         queue, read a step ahead of their use; the FIR7 window and -- for
//         the tap counts that occur -- the gather window live in registers
//         (the step loop is unrolled to the window length, ring slots are
//         compile-time); coefficient rows are staged in LDS per chunk and read
//         as broadcasts one output ahead.
//   k_gh  horizontal pass. Lane = whole output pixel (16 B) of a strip of up
//         to 64 outputs; its coefficient row stays in registers while the wave
//         walks down its rows; the row's source segment arrives by LDS-DMA
//         three rows ahead.
//   k_sacc (sacc.hip) takes the downsizing axes (k >= 2) of integer / narrow
//         sources: streaming accumulation instead of a gather window.
//
// Arithmetic contract (bit-exact): mul and add separate (-ffp-contract=off),
// AVIR sums start from +0.0f and take their taps in ascending order, the FIR
// is f0*c + sum f_i*(right_i + left_i); LANCIR accumulates even and odd taps
// separately from bare first products and adds the two sums at the end.

#include "gpass_dev.h"
#include <algorithm>
#include <math.h>
#include <type_traits>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

namespace avirhip {

struct GVParams
{
	const float* src; long src_ss; int width; // px per row, floats per row
	float* dst; long dst_ss; int dst_row0;
	GPAxis ax;
	int row_lo, row_hi;  // output rows to produce
	int dbg;             // timing ablations (GP_DBG builds only)
	int nstrips, chunk;  // 32-px strips, output rows per work item
	int blk;             // output rows per block
	int rs, rc;          // ring sizes (rows, powers of two): source, C / G
	GPOut out;           // on != 0 (whole-pixel lanes): integer output stage
	const void* raw;     // != nullptr (whole-pixel lanes): the caller's image
	long raw_ss;         // (integer / narrower pixels), elements per row,
	int raw_ch, raw_kind; // channels, 1 uint8 / 2 uint16 / 3 float
	long raw_elems;      // elements of that image (the loader's bound)
};

struct GHParams
{
	const float* src; long src_ss; int src_w;
	float* dst; long dst_ss; int dst_row0; int dst_w;
	GPAxis ax;
	int row_lo, row_hi;  // rows (of both images) to process
	int nstrips, ow, chunk; // outputs per strip, rows per work item
	const int* seg;      // [nstrips][4]: source lo, count; mid lo, count
	int nseg;            // DMA instructions per row (uniform)
	int sbuf, mbuf;      // LDS pixels per source row buffer / mid buffer
	const void* raw;     // SRC != 0: the caller's image (integer / narrower
	long raw_ss;         // pixels), elements per row, channels
	int raw_ch;
	long raw_elems;      // elements of that image (the loader's bound)
	GPLOut lout;         // on != 0: LANCIR's output stage in the store
};

// One gather output: tap t reads ld( t ); cf4( b ) returns the coefficients
// of taps 4b .. 4b + 3 (rows are padded to a multiple of four; padded taps are
// never multiplied: x * 0 is not a no-op for non-finite x). NTC > 0: the tap
// count is a compile-time constant and everything unrolls.
//
// LANCIR's summation order depends on the image's channel count (the lane
// structure of resize1..4, lancir.h:2102-2544; generic.hip spells it out):
// `lord` = 4: even and odd taps; 1, 2, 3: four sums over taps 4g + j, folded
// (s0 + s2) + (s1 + s3) for 1 and 2 channels, (s0 + s1) + (s2 + s3) for 3, with
// the two tail taps of a length 4n + 2 entering where that code adds them --
// for RGB differently in channel 0 (`ch0`: this lane's .x is channel 0).
template< bool LANC, typename T, int NTC, typename LD, typename CF >
__device__ __forceinline__ T gp_dot( const int nt_, LD ld, CF cf4,
	const int lord = 4, const bool ch0 = true )
{
	const int nt = ( NTC > 0 ? NTC : nt_ );
	const int nb = nt >> 2;

	if( LANC && lord != 4 )
	{
		T a0, a1, a2, a3;
		{
			const f4 c = cf4( 0 );
			a0 = c.x * ld( 0 );
			a1 = c.y * ld( 1 );
			a2 = c.z * ld( 2 );
			a3 = c.w * ld( 3 );
		}

		if( NTC > 0 )
		{
#pragma unroll
			for( int b = 1; b < ( NTC > 0 ? NTC / 4 : 1 ); b++ )
			{
				const f4 c = cf4( b );
				a0 = a0 + c.x * ld( 4 * b );
				a1 = a1 + c.y * ld( 4 * b + 1 );
				a2 = a2 + c.z * ld( 4 * b + 2 );
				a3 = a3 + c.w * ld( 4 * b + 3 );
			}
		}
		else
		{
			for( int b = 1; b < nb; b++ )
			{
				const f4 c = cf4( b );
				a0 = a0 + c.x * ld( 4 * b );
				a1 = a1 + c.y * ld( 4 * b + 1 );
				a2 = a2 + c.z * ld( 4 * b + 2 );
				a3 = a3 + c.w * ld( 4 * b + 3 );
			}
		}

		if( nt & 2 )
		{
			const f4 c = cf4( nb );
			const T t0 = c.x * ld( 4 * nb );
			const T t1 = c.y * ld( 4 * nb + 1 );

			if( lord == 3 )
			{
				a0 = a0 + t0;
				T r = (( a0 + a1 ) + ( a2 + a3 )) + t1;     // channels 1, 2
				const T r0 = ( a0 + ( a1 + t1 )) + ( a2 + a3 ); // channel 0

				if( ch0 )
				{
					r.x = r0.x;
				}

				return( r );
			}

			return((( a0 + a2 ) + t0 ) + (( a1 + a3 ) + t1 ));
		}

		if( lord == 3 )
		{
			return(( a0 + a1 ) + ( a2 + a3 ));
		}

		return(( a0 + a2 ) + ( a1 + a3 ));
	}

	if( LANC )
	{
		// (t0*p0 + t2*p2 + ...) + (t1*p1 + t3*p3 + ...), bare first products
		// (lancir.h:2482-2515); nt is even
		T ea, oa;
		int b = 0;

		if( nt >= 4 )
		{
			const f4 c = cf4( 0 );
			ea = c.x * ld( 0 );
			oa = c.y * ld( 1 );
			ea = ea + c.z * ld( 2 );
			oa = oa + c.w * ld( 3 );
			b = 1;
		}
		else
		{
			const f4 c = cf4( 0 );
			ea = c.x * ld( 0 );
			oa = c.y * ld( 1 );
			return( ea + oa );
		}
		if( NTC > 0 )
		{
#pragma unroll
			for( int bb = 1; bb < ( NTC > 0 ? NTC / 4 : 1 ); bb++ )
			{
				const f4 c = cf4( bb );
				ea = ea + c.x * ld( 4 * bb );
				oa = oa + c.y * ld( 4 * bb + 1 );
				ea = ea + c.z * ld( 4 * bb + 2 );
				oa = oa + c.w * ld( 4 * bb + 3 );
			}
		}
		else
		{
#pragma unroll 2
			for( ; b < nb; b++ )
			{
				const f4 c = cf4( b );
				ea = ea + c.x * ld( 4 * b );
				oa = oa + c.y * ld( 4 * b + 1 );
				ea = ea + c.z * ld( 4 * b + 2 );
				oa = oa + c.w * ld( 4 * b + 3 );
			}
		}

		if( nt & 2 )
		{
			const f4 c = cf4( nb );
			ea = ea + c.x * ld( 4 * nb );
			oa = oa + c.y * ld( 4 * nb + 1 );
		}

		return( ea + oa );
	}

	T acc = (T) 0.0f;

	if( NTC > 0 )
	{
#pragma unroll
		for( int b = 0; b < ( NTC > 0 ? NTC / 4 : 1 ); b++ )
		{
			const f4 c = cf4( b );
			acc = acc + c.x * ld( 4 * b );
			acc = acc + c.y * ld( 4 * b + 1 );
			acc = acc + c.z * ld( 4 * b + 2 );
			acc = acc + c.w * ld( 4 * b + 3 );
		}
	}
	else
	{
#pragma unroll 2
		for( int b = 0; b < nb; b++ )
		{
			const f4 c = cf4( b );
			acc = acc + c.x * ld( 4 * b );
			acc = acc + c.y * ld( 4 * b + 1 );
			acc = acc + c.z * ld( 4 * b + 2 );
			acc = acc + c.w * ld( 4 * b + 3 );
		}
	}

	if( nt & 3 )
	{
		const f4 c = cf4( nb );
		acc = acc + c.x * ld( 4 * nb );

		if(( nt & 3 ) > 1 )
		{
			acc = acc + c.y * ld( 4 * nb + 1 );
		}

		if(( nt & 3 ) > 2 )
		{
			acc = acc + c.z * ld( 4 * nb + 2 );
		}
	}

	return( acc );
}

// One raw pixel (1 uint8, 2 uint16, 3 float elements; `ch` of them) as float
// RGBA with zero padding: ONE load of four elements at the pixel's (possibly
// unaligned) address -- what it reads past the pixel is dropped; only where
// those four elements would cross the end of the image (its last pixel) are
// the channels loaded one by one. Per-element loads everywhere cost three or
// four vector-memory instructions per pixel and as much time as the pack pass
// they replace. (Buffer loads force dword alignment: global loads it is.)
__device__ __forceinline__ f4 gp_load_raw( const void* const base,
	const int kind, const int ch, const long eoff, const long total )
{
	float v[ 4 ] = { 0.0f, 0.0f, 0.0f, 0.0f };
	const bool fast = ( eoff + 4 <= total );
	// integer elements, branch-free: the load starts at most at total - 4 and
	// the value is shifted down by the elements it started early (non-zero
	// for the image's last pixel only; the image holds >= 4 elements)
	const long ea = min( eoff, total - 4 );
	const int sh = (int) ( eoff - ea );

	if( kind == 1 )
	{
		typedef unsigned int u32u __attribute__(( aligned( 1 )));
		const unsigned int w = *(const u32u*) ( (const unsigned char*) base +
			ea ) >> ( 8 * sh );
#pragma unroll
		for( int c = 0; c < 4; c++ )
		{
			v[ c ] = ( c < ch ? (float) (( w >> ( 8 * c )) & 0xffu ) : 0.0f );
		}
	}
	else
	if( kind == 2 )
	{
		typedef unsigned long long u64u __attribute__(( aligned( 2 )));
		const unsigned long long w = *(const u64u*) (
			(const unsigned short*) base + ea ) >> ( 16 * sh );
#pragma unroll
		for( int c = 0; c < 4; c++ )
		{
			v[ c ] = ( c < ch ? (float) (unsigned int) (( w >> ( 16 * c )) &
				0xffffull ) : 0.0f );
		}
	}
	else
	{
		const float* const p = (const float*) base + eoff;

		if( fast )
		{
			typedef float f4u __attribute__(( ext_vector_type( 4 ),
				aligned( 4 )));
			const f4u w = *(const f4u*) p;
			const float ww[ 4 ] = { w.x, w.y, w.z, w.w };
#pragma unroll
			for( int c = 0; c < 4; c++ )
			{
				v[ c ] = ( c < ch ? ww[ c ] : 0.0f );
			}
		}
		else
		{
			for( int c = 0; c < ch; c++ )
			{
				v[ c ] = p[ c ];
			}
		}
	}

	f4 o; o.x = v[ 0 ]; o.y = v[ 1 ]; o.z = v[ 2 ]; o.w = v[ 3 ];
	return( o );
}

// ---------------------------------------------------------------------
// vertical pass
// ---------------------------------------------------------------------

// One source row per step: the row's pixels travel global memory -> register
// queue (GV_PF rows in flight, statically indexed: the step loop is unrolled
// GV_PF times) -> whatever the mode keeps of them:
//   PRE   a 7-row FIR window in registers (statically indexed too); each
//         step yields one row of C, stored in the LDS ring the gathers read. The clamped source view makes FIR( Sext )
//         the clamped view of C for EVERY index (rows before the first / after
//         the last are constant), so there are no edge cases, only clamped
//         row loads; rows of C above the zero-stuffed view's end are zero.
//   NONE  the row goes into the LDS source ring.
//   POST  the same, and gather outputs go through a second LDS ring that the
//         correction FIR reads.
// After every step the outputs whose last input has arrived are produced (a
// wave-uniform while loop). LDS holds only what is addressed dynamically:
// rs rows >= NT + 1 of source (NONE / POST), rc rows of C (>= NT + 1) or of
// gather outputs (8).
#define GV_PF 5 // DMA groups in flight (one of them is read a step ahead)
#define GV_QB (( GV_PF + 2 ) * 1024 ) // landing queue: GV_PF + 1 slots, a dump

// RW > 0 (needs NTC > 0): the gather window lives in REGISTERS. An output is
// produced in the very step its last input row arrives, so its window is
// always "the last NTC rows": with the step loop unrolled RW times (RW a power
// of two >= NTC) the window's ring slots are compile-time constants -- no LDS
// reads for the taps at all, LDS only holds the landing queue and the tables.
// WP: lane width. 1 = half pixel (8 B): a wave row is 32 pixels, one DMA
// instruction lands two rows. 2 = whole pixel (16 B): 64 pixels per wave row,
// one row per DMA instruction -- twice the independent arithmetic per
// instruction stream: the kernel is bound by the serial latency of a wave
// (queue read -> FIR chain -> 12-deep accumulation chain), not by issue slots,
// so the wide form is the default wherever the image is wide enough.
template< int WP > struct GVLane;
template<> struct GVLane< 1 > { typedef f2 T; };
template<> struct GVLane< 2 > { typedef f4 T; };

template< int MODE, bool LANC, int NTC, int RW, int WP >
__global__ void __launch_bounds__( 64 ) k_gv( const GVParams P )
{
	typedef typename GVLane< WP > :: T VT;
	constexpr int ROWB = WP * 512;   // LDS bytes of one wave row
	constexpr int RPG = 2 / WP;      // rows per DMA group (instruction)
	constexpr int PXW = 32 * WP;     // pixels per wave row
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	const GPAxis& A = P.ax;
	VT* const sring = (VT*) smem;                       // [rs][64 lanes]
	VT* const mring = sring + (size_t) P.rs * 64;       // [rc][64 lanes]
	// behind the rings and the 6 KiB landing queue: this chunk's coefficient
	// rows (padded to four taps) and positions -- in LDS, not behind scalar
	// loads: a scalar load in the tap loop costs its whole latency per use
	// (every wait on lgkmcnt with one outstanding is a full drain)
	float* const ctab = (float*) ( smem + ( P.rs + P.rc ) * ROWB + GV_QB );

	const int lane = threadIdx.x;
	const int strip = blockIdx.x % P.nstrips;
	const int chunk = blockIdx.x / P.nstrips;
	const int x0 = strip * PXW;
	const int y0 = P.row_lo + chunk * P.chunk;
	const int y1 = min( y0 + P.chunk, P.row_hi );

	if( y0 >= y1 )
	{
		return;
	}

	gp_cint* const a_start = (gp_cint*) (unsigned long) A.start;
	const int smask = P.rs - 1;
	const int mmask = P.rc - 1;
	const int slast = A.in_len - 1;
	const int nt = ( NTC > 0 ? NTC : A.nt );
	const int ntp = ( nt + 3 ) & ~3;

	// this lane's (half) pixel
	const int hx = x0 + ( WP == 2 ? lane : lane >> 1 );
	const bool xok = ( hx < P.width );
	const long dcol = (long) hx * 4 + ( WP == 2 ? 0 : ( lane & 1 ) * 2 );
	// (LANCIR, RGB: channel 0 sums its tail taps differently -- gp_dot)
	const bool ach = ( WP == 2 || ( lane & 1 ) == 0 );

	auto mread = [&]( const int m ) -> VT
	{
		return( mring[ ( m & mmask ) * 64 + lane ]);
	};

	auto sread = [&]( const int u ) -> VT
	{
		return( sring[ ( u & smask ) * 64 + lane ]);
	};

	const VT f0 = (VT) A.f[ 0 ], f1 = (VT) A.f[ 1 ], f2_ = (VT) A.f[ 2 ],
		f3 = (VT) A.f[ 3 ];

	// gather outputs this chunk evaluates: [ga, gb]
	const int glast = A.g_len - 1;
	const int ga = ( MODE == GP_POST ? gp_clamp( y0 - A.e - 3, glast ) : y0 );
	const int gb = ( MODE == GP_POST ? gp_clamp( y1 - 1 - A.e + 3, glast ) :
		y1 - 1 );

	// source rows fed: u_a .. u_b (PRE: six rows of FIR warm-up first)
	const int u_a = a_start[ ga ] - ( MODE == GP_PRE ? 6 : 0 );
	const int u_b = a_start[ gb ] + nt - 1;

	// stage the tables (plain loads: they retire before the first DMA goes
	// out). All of a lane's loads are issued before any is written to LDS --
	// a load-store loop pays the memory latency once per 64 elements, which
	// measured as a third of this kernel's time.
	int* const stab = (int*) ( ctab + ( gb - ga + 1 ) * ntp );
	{
		const int total = ( gb - ga + 1 ) * ntp;
		constexpr int GV_TE = 24; // 6 KiB of table / 4 bytes / 64 lanes
		float tmp[ GV_TE ];
#pragma unroll
		for( int k = 0; k < GV_TE; k++ )
		{
			const int i = lane + 64 * k;
			tmp[ k ] = 0.0f;

#ifdef GP_DBG
			if( P.dbg & 16 ) continue;
#endif
			if( i < total )
			{
				const int gi = i / ntp;
				const int t = i - gi * ntp;
				const long row = ( LANC ? A.fidx[ ga + gi ] : ga + gi );

				if( t < nt )
				{
					tmp[ k ] = A.coef[ row * nt + t ];
				}
			}
		}
#pragma unroll
		for( int k = 0; k < GV_TE; k++ )
		{
			const int i = lane + 64 * k;

			if( i < total )
			{
				ctab[ i ] = tmp[ k ];
			}
		}

		for( int i = lane; i <= gb - ga; i += 64 )
		{
			stab[ i ] = A.start[ ga + i ] + nt - 1;
		}
	}

	int g = ga;                 // next gather output
	int g_need = a_start[ g ] + nt - 1; // its last input row
	int n = y0;                 // POST: next final output

	// output rows leave strictly in order, from y0 on: a running pointer
	float* dp = P.dst + (long) ( y0 - P.dst_row0 ) * P.dst_ss + dcol;
	long orow = y0 - P.dst_row0; // (integer output stage: row of P.out.base)

	auto store_row = [&]( const VT o )
	{
#ifdef GP_DBG
		if(( P.dbg & 1 ) && o.x != 1234.5f ) return;
#endif
		if( WP == 2 && P.out.on != 0 )
		{
			if( xok )
			{
				f4 o4;
				o4.x = o.x; o4.y = o.y;
				o4.z = ( WP == 2 ? o[ WP == 2 ? 2 : 0 ] : 0.0f );
				o4.w = ( WP == 2 ? o[ WP == 2 ? 3 : 0 ] : 0.0f );
				gp_store_int( P.out, orow, hx, o4 );
			}

			orow++;
			return;
		}

		if( xok )
		{
			*(VT*) dp = o;
		}

		dp += P.dst_ss;
	};

	// the next output's coefficient row and the position after it are read
	// from LDS an output ahead, so that their latency hides behind the
	// arithmetic in between (a wave runs alone on most of its issue slots:
	// exposed LDS round trips were a third of the step time)
	constexpr int CNB = ( NTC > 0 && NTC <= 16 ? ( NTC + 3 ) / 4 : 0 );
	f4 cpre[ CNB > 0 ? CNB : 1 ];
	int need1_v; // g_need1, still in its vector register

	auto prefetch_tabs = [&]()
	{
		// (rows past gb are read -- clamped -- and never used)
		const int gi = min( g + 1, gb ) - ga;
		need1_v = ( g + 1 <= gb ? stab[ gi ] : 0x7fffffff );

		if( CNB > 0 )
		{
			const float* const cr = ctab + ( min( g, gb ) - ga ) * ntp;
#pragma unroll
			for( int b = 0; b < CNB; b++ )
			{
				cpre[ b ] = *(const f4*) ( cr + 4 * b );
			}
		}
	};

	VT win[ RW > 0 ? RW : 1 ]; // RW > 0: the last RW rows the gathers read
#pragma unroll
	for( int i = 0; i < ( RW > 0 ? RW : 1 ); i++ )
	{
		win[ i ] = (VT) 0.0f;
	}

	// gather output g is done: store it, or (POST) run the correction FIR
	auto finish = [&]( const VT o )
	{
		if( MODE == GP_POST )
		{
			mring[ ( g & mmask ) * 64 + lane ] = o;

			// final outputs whose FIR window [c - 3, c + 3] (clamped) is
			// complete: c + 3 <= g, or g is the last gather output
			while( n < y1 && ( gp_clamp( n - A.e + 3, glast ) <= g ))
			{
				const int c = n - A.e;
				VT r = f0 * mread( gp_clamp( c, glast ));
				r = r + f1 * ( mread( gp_clamp( c + 1, glast )) +
					mread( gp_clamp( c - 1, glast )));
				r = r + f2_ * ( mread( gp_clamp( c + 2, glast )) +
					mread( gp_clamp( c - 2, glast )));
				r = r + f3 * ( mread( gp_clamp( c + 3, glast )) +
					mread( gp_clamp( c - 3, glast )));
				store_row( r );
				n++;
			}
		}
		else
		{
			store_row( o );
		}

		g++;
		g_need = __builtin_amdgcn_readfirstlane( need1_v );
		prefetch_tabs();
	};

	// everything whose inputs are complete once row `u` is in its ring.
	// IC: the step's phase within the unrolled loop (window slot of row u).
	auto drain = [&]( const int u, auto IC )
	{
		constexpr int I = decltype( IC )::value;

		while( g <= gb && g_need <= u )
		{
			const int s0 = g_need - nt + 1;
			const float* const cr = ctab + ( g - ga ) * ntp;
			VT o;

#ifdef GP_DBG
			if( P.dbg & 2 )
			{
				o = win[ 0 ];
			}
			else
#endif
			if( RW > 0 )
			{
				// g_need == u: outputs are drained the step they complete
				o = gp_dot< LANC, VT, NTC >( nt,
					[&]( const int t ) { return( win[ ( I - ( NTC - 1 ) + t +
						4 * ( RW > 0 ? RW : 1 )) & (( RW > 0 ? RW : 1 ) - 1 )]); },
					[&]( const int b ) { return( CNB > 0 ? cpre[ CNB > 0 ? b : 0 ] :
						*(const f4*) ( cr + 4 * b )); }, A.lanc, ach );
			}
			else
			if( MODE == GP_PRE )
			{
				o = gp_dot< LANC, VT, NTC >( nt,
					[&]( const int t ) { return( mread( s0 + t )); },
					[&]( const int b ) { return( CNB > 0 ? cpre[ CNB > 0 ? b : 0 ] :
						*(const f4*) ( cr + 4 * b )); }, A.lanc, ach );
			}
			else
			{
				o = gp_dot< LANC, VT, NTC >( nt,
					[&]( const int t ) { return( sread( s0 + t )); },
					[&]( const int b ) { return( CNB > 0 ? cpre[ CNB > 0 ? b : 0 ] :
						*(const f4*) ( cr + 4 * b )); }, A.lanc, ach );
			}

			finish( o );
		}
	};

	prefetch_tabs();

	// ---- the feed: source rows by LDS-DMA, one group per instruction (WP 1:
	// a row pair, lanes 0-31 row 2p, lanes 32-63 row 2p + 1; WP 2: one row),
	// GV_PF groups in flight. Waiting for "at most GV_PF - 1 vector-memory
	// operations outstanding" right after issuing group p + GV_PF guarantees
	// group p + 1 has landed (loads retire in order; stores in between only
	// make the wait stricter): rows are read from the queue a step ahead.
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
		(void*) P.src, 0, 0x7fffffff, 0x00020000 );
	const unsigned lds0 = (unsigned) (unsigned long) (lds_char*) smem;
	const int dpx = gp_clamp( x0 + ( lane & ( PXW - 1 )), P.width - 1 );
	const int srow_b = (int) P.src_ss * 4;
	// PRE: pairs land in a queue of GV_PF + 1 slots behind the C ring;
	// NONE / POST: straight in the source ring (rs >= nt + 2 * GV_PF + 3)
	const unsigned qbase = lds0 + (unsigned) ( P.rs + P.rc ) * ROWB;
	const int pa = ( RPG == 2 ? u_a >> 1 : u_a ); // first group (floor)
	const int pb = ( RPG == 2 ? u_b >> 1 : u_b );

	int qi = 0; // landing-queue slot of the next pair issued ...
	int qr = 0; // ... and of the pair being read (both count modulo GV_PF + 1)

	auto issue_pair = [&]( const int p )
	{
		// past the last group: the count of operations per step stays
		// uniform -- a load nobody reads, into the dump slot (no branch)
		const bool live = ( p <= pb );
		const int r = gp_clamp( min( p, pb ) * RPG +
			( RPG == 2 ? lane >> 5 : 0 ), slast );

		const unsigned slot = ( !live ? qbase + (unsigned) ( GV_PF + 1 ) * 1024 :
			( MODE == GP_PRE || RW > 0 ? qbase + (unsigned) qi * 1024 :
			lds0 + (unsigned) (( p * RPG ) & smask ) * ROWB ));

		qi = ( !live ? qi : ( qi == GV_PF ? 0 : qi + 1 ));

		gp_dma( rsrc, slot, r * srow_b + dpx * 16 );
	};

	// Raw sources (integer / narrower pixels, whole-pixel lanes): a lane loads
	// its pixel of row u + 2 during step u, converts and writes it into the
	// queue slot (or ring row) of row u + 1 at the start of the next step --
	// the (float) cast and zero padding of the pack pass, in the loader.
	const bool typed = ( WP == 2 && P.raw != nullptr );

	f4 trv = (f4) 0.0f;

	auto t_load = [&]( const int u )
	{
		trv = gp_load_raw( P.raw, P.raw_kind, P.raw_ch, (long) gp_clamp( u,
			slast ) * P.raw_ss + (long) dpx * P.raw_ch, P.raw_elems );
	};

	auto t_store = [&]( const int u ) // row u: queue slot qr, or its ring row
	{
		const f4 o = trv;
		const int off = ( MODE == GP_PRE || RW > 0 ?
			( P.rs + P.rc ) * ROWB + qr * 1024 : ( u & smask ) * ROWB );

		((f4*) ( smem + off ))[ lane ] = o;
	};

	if( typed )
	{
		t_load( pa );
		t_store( pa );
		t_load( pa + 1 );
	}
	else
	{
		for( int k = 0; k < GV_PF; k++ )
		{
			issue_pair( pa + k );
		}
	}

	VT qnext = (VT) 0.0f; // the next step's row of the queue

	GP_WAIT_VM( 4 ); // the first group has landed

	if( MODE == GP_PRE || RW > 0 )
	{
		qnext = *(const VT*) ( smem + ( P.rs + P.rc ) * ROWB +
			lane * sizeof( VT ));
	}

	VT tw[ 8 ]; // PRE: the FIR window (rows u - 6 .. u in slots ( u - u_a ) & 7)
#pragma unroll
	for( int i = 0; i < 8; i++ )
	{
		tw[ i ] = (VT) 0.0f;
	}

	// steps run over whole pairs; rows before u_a / after u_b of the first /
	// last pair are fed like any other (clamped loads), outputs never need
	// them
	const int u0 = pa * RPG;

	constexpr int UN = ( RW > 8 ? RW : 8 );

	auto step = [&]( const int u, auto IC )
	{
		constexpr int i = decltype( IC )::value;

		if( u > ( pb * RPG + RPG - 1 ))
		{
			return;
		}

		if( typed )
		{
			qr = ( qr == GV_PF ? 0 : qr + 1 );
			t_store( u + 1 );
			t_load( u + 2 );
		}
		else
		{
			if( RPG == 1 || ( i & 1 ) == 0 )
			{
#ifdef GP_DBG
				if( !( P.dbg & 8 ))
#endif
				issue_pair(( RPG == 2 ? u >> 1 : u ) + GV_PF );
			}

			if( RPG == 1 || ( i & 1 ))
			{
				// row u + 1 opens a group: it is the oldest one in flight
#ifdef GP_DBG
				if( !( P.dbg & 4 ))
#endif
				GP_WAIT_VM( 4 ); // == GV_PF - 1

				qr = ( qr == GV_PF ? 0 : qr + 1 );
			}
		}

		// this step's row was read from the queue during the last one
		const VT qcur = qnext;

		if( MODE == GP_PRE || RW > 0 )
		{
			qnext = *(const VT*) ( smem + ( P.rs + P.rc ) * ROWB + qr * 1024 +
				( RPG == 2 ? (( i + 1 ) & 1 ) * 512 : 0 ) + lane * sizeof( VT ));
		}

		if( MODE == GP_PRE )
		{
			tw[ i & 7 ] = qcur;
			VT c = f0 * tw[ ( i - 3 ) & 7 ];
			c = c + f1 * ( tw[ ( i - 2 ) & 7 ] + tw[ ( i - 4 ) & 7 ]);
			c = c + f2_ * ( tw[ ( i - 1 ) & 7 ] + tw[ ( i - 5 ) & 7 ]);
			c = c + f3 * ( tw[ i & 7 ] + tw[ ( i - 6 ) & 7 ]);

			if( u > A.view_mmax )
			{
				c = (VT) 0.0f;
			}

			if( RW > 0 )
			{
				win[ i & (( RW > 0 ? RW : 1 ) - 1 )] = c;
			}
			else
			{
				mring[ ( u & mmask ) * 64 + lane ] = c;
			}

			if( u - u0 >= 6 )
			{
				drain( u, IC );
			}
		}
		else
		{
			if( RW > 0 )
			{
				VT v = qcur;

				if( u > A.view_mmax )
				{
					v = (VT) 0.0f;
				}

				win[ i & (( RW > 0 ? RW : 1 ) - 1 )] = v;
			}
			else
			if( u > A.view_mmax )
			{
				sring[ ( u & smask ) * 64 + lane ] = (VT) 0.0f;
			}

			drain( u, IC );
		}
	};

#ifdef GP_DBG
	if( P.dbg & 32 ) return;
#endif
	for( int ub = u0; ub <= u_b; ub += UN )
	{
		step( ub + 0, std::integral_constant< int, 0 >() );
		step( ub + 1, std::integral_constant< int, 1 >() );
		step( ub + 2, std::integral_constant< int, 2 >() );
		step( ub + 3, std::integral_constant< int, 3 >() );
		step( ub + 4, std::integral_constant< int, 4 >() );
		step( ub + 5, std::integral_constant< int, 5 >() );
		step( ub + 6, std::integral_constant< int, 6 >() );
		step( ub + 7, std::integral_constant< int, 7 >() );

		if( UN > 8 )
		{
			step( ub + 8, std::integral_constant< int, 8 >() );
			step( ub + 9, std::integral_constant< int, 9 >() );
			step( ub + 10, std::integral_constant< int, 10 >() );
			step( ub + 11, std::integral_constant< int, 11 >() );
			step( ub + 12, std::integral_constant< int, 12 >() );
			step( ub + 13, std::integral_constant< int, 13 >() );
			step( ub + 14, std::integral_constant< int, 14 >() );
			step( ub + 15, std::integral_constant< int, 15 >() );
		}

		if( UN > 16 )
		{
			step( ub + 16, std::integral_constant< int, 16 >() );
			step( ub + 17, std::integral_constant< int, 17 >() );
			step( ub + 18, std::integral_constant< int, 18 >() );
			step( ub + 19, std::integral_constant< int, 19 >() );
			step( ub + 20, std::integral_constant< int, 20 >() );
			step( ub + 21, std::integral_constant< int, 21 >() );
			step( ub + 22, std::integral_constant< int, 22 >() );
			step( ub + 23, std::integral_constant< int, 23 >() );
			step( ub + 24, std::integral_constant< int, 24 >() );
			step( ub + 25, std::integral_constant< int, 25 >() );
			step( ub + 26, std::integral_constant< int, 26 >() );
			step( ub + 27, std::integral_constant< int, 27 >() );
			step( ub + 28, std::integral_constant< int, 28 >() );
			step( ub + 29, std::integral_constant< int, 29 >() );
			step( ub + 30, std::integral_constant< int, 30 >() );
			step( ub + 31, std::integral_constant< int, 31 >() );
		}
	}

	GP_WAIT_VM( 0 );
}

// ---------------------------------------------------------------------
// horizontal pass
// ---------------------------------------------------------------------

#define GH_NB 4 // source row buffers: rows r .. r + 3 are in LDS or in flight

// SRC: 0 float RGBA by LDS-DMA; 1 uint8, 2 uint16, 3 float pixels of raw_ch
// channels, read as they are (packScanline's cast and zero padding,
// avir.h:2777-2840, in the loader: no float copy of the source): a lane loads
// its pixel of each 64-pixel piece a row ahead, converts and writes it into
// the row buffer after the row in hand has been computed.
template< int SRC > struct GHElem { typedef float T; };
template<> struct GHElem< 1 > { typedef unsigned char T; };
template<> struct GHElem< 2 > { typedef unsigned short T; };

template< int MODE, bool LANC, int NTC, int SRC >
__global__ void __launch_bounds__( 64 ) k_gh( const GHParams P )
{
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	const GPAxis& A = P.ax;
	f4* const sbuf = (f4*) smem;                    // [GH_NB][sbuf px]
	f4* const mbuf = sbuf + GH_NB * (size_t) P.sbuf; // [mbuf px]
	float* const cfl = (float*) ( mbuf + P.mbuf );  // NTC == 0: [nt][64 lanes]
	const unsigned sbuf_lds = (unsigned) (unsigned long) (lds_char*) smem;

	const int lane = threadIdx.x;
	const int strip = blockIdx.x % P.nstrips;
	const int chunk = blockIdx.x / P.nstrips;
	const int r0 = P.row_lo + chunk * P.chunk;
	const int r1 = min( r0 + P.chunk, P.row_hi );

	if( r0 >= r1 )
	{
		return;
	}

	const int slo = P.seg[ strip * 4 ], scnt = P.seg[ strip * 4 + 1 ];
	const int mlo = P.seg[ strip * 4 + 2 ], mcnt = P.seg[ strip * 4 + 3 ];
	const int j0 = strip * P.ow;
	const int j = j0 + lane;                        // this lane's output
	const bool jok = ( lane < P.ow && j < A.out_len );
	const int nt = ( NTC > 0 ? NTC : A.nt );
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
		(void*) P.src, 0, 0x7fffffff, 0x00020000 );
	const int srow_b = (int) P.src_ss * 4;

	// per-lane source pixels of the DMA pieces (clamped: the extended view)
	const int slast = A.in_len - 1;

	auto issue_row = [&]( const int r, const int buf )
	{
		for( int k = 0; k < P.nseg; k++ )
		{
			const int x = gp_clamp( slo + k * 64 + lane, slast );
			gp_dma( rsrc, sbuf_lds + ( buf * P.sbuf + k * 64 ) * 16,
				r * srow_b + x * 16 );
		}
	};

	// The gather this lane evaluates, its first input (relative to the
	// buffer it reads) and its coefficient row -- fixed for every row.
	//   PRE / NONE: gather output == final output j
	//   POST: lane l evaluates G[ mlo + l ] (the strip's outputs +- 3)
	const int g = ( MODE == GP_POST ? gp_clamp( mlo + lane, A.g_len - 1 ) :
		gp_clamp( j, A.g_len - 1 ));

	const bool gok = ( MODE == GP_POST ? lane < mcnt : jok );
	const int gs = A.start[ g ] - ( MODE == GP_PRE ? mlo : slo );
	// a compile-time tap count keeps the row in registers; any other count
	// parks it in LDS, tap-major (lane-consecutive: conflict-free reads)
	float cf[ NTC > 0 ? NTC : 1 ];
	{
		const float* cr = A.coef + (long) ( LANC ? A.fidx[ g ] : g ) * nt;

		if( NTC > 0 )
		{
#pragma unroll
			for( int t = 0; t < ( NTC > 0 ? NTC : 1 ); t++ )
			{
				cf[ t ] = cr[ t ];
			}
		}
		else
		{
			for( int t = 0; t < nt; t++ )
			{
				cfl[ t * 64 + lane ] = cr[ t ];
			}
		}
	}

	// typed sources: the row in flight (up to four pieces, four channels)
	f4 rw[ SRC != 0 ? 4 : 1 ];

	auto t_load = [&]( const int r )
	{
#pragma unroll
		for( int k = 0; k < ( SRC != 0 ? 4 : 1 ); k++ )
		{
			if( k < P.nseg )
			{
				rw[ k ] = gp_load_raw( P.raw, SRC, P.raw_ch, (long) r * P.raw_ss +
					(long) gp_clamp( slo + k * 64 + lane, slast ) * P.raw_ch,
					P.raw_elems );
			}
		}
	};

	auto t_store = [&]( const int buf )
	{
#pragma unroll
		for( int k = 0; k < ( SRC != 0 ? 4 : 1 ); k++ )
		{
			if( k < P.nseg )
			{
				sbuf[ buf * P.sbuf + k * 64 + lane ] = rw[ k ];
			}
		}
	};

	if( SRC != 0 )
	{
		t_load( r0 );
		t_store( 0 );

		if( r0 + 1 < r1 )
		{
			t_load( r0 + 1 );
		}
	}
	else
	{
		for( int k = 0; k < GH_NB - 1 && r0 + k < r1; k++ )
		{
			issue_row( r0 + k, k );
		}
	}

	for( int r = r0; r < r1; r++ )
	{
		const int buf = ( r - r0 ) & ( GH_NB - 1 );
		const f4* const s = sbuf + buf * P.sbuf;

		if( SRC == 0 )
		{
			// row r + 3 goes out (into the buffer row r - 1 has left), then
			// wait for row r: only the pieces of the rows after it may be in
			// flight
			if( r + GH_NB - 1 < r1 )
			{
				issue_row( r + GH_NB - 1, ( buf + GH_NB - 1 ) & ( GH_NB - 1 ));
			}

			gp_wait_vm( min( GH_NB - 1, r1 - 1 - r ) * P.nseg );
		}

		if( MODE != GP_PRE && slo + scnt - 1 > A.view_mmax )
		{
			// zero-stuffed view of the source: its tail reads 0
			for( int x = lane; x < scnt; x += 64 )
			{
				if( slo + x > A.view_mmax )
				{
					((f4*) s )[ x ] = (f4) 0.0f;
				}
			}
		}

		if( MODE == GP_PRE )
		{
			// C[ mlo .. mlo + mcnt ) = FIR7 along x of the segment
			for( int i = lane; i < mcnt; i += 64 )
			{
				const int m = mlo + i;
				f4 c;

				if( m > A.view_mmax )
				{
					c = (f4) 0.0f;
				}
				else
				{
					const f4* q = s + ( gp_clamp( m, A.c_len - 1 ) - A.e - slo );
					c = A.f[ 0 ] * q[ 0 ];
					c = c + A.f[ 1 ] * ( q[ 1 ] + q[ -1 ]);
					c = c + A.f[ 2 ] * ( q[ 2 ] + q[ -2 ]);
					c = c + A.f[ 3 ] * ( q[ 3 ] + q[ -3 ]);
				}

				mbuf[ i ] = c;
			}
		}

		f4 o = (f4) 0.0f;

		if( gok )
		{
			const f4* const in = ( MODE == GP_PRE ? mbuf : s ) + gs;
			// a compile-time tap count: the whole window is read first (all
			// its ds_read_b128 in flight at once), then the arithmetic runs
			// -- left alone the compiler keeps two reads in flight and pays
			// the LDS latency six times per pixel
			f4 w[ NTC > 0 ? NTC : 1 ];

			if( NTC > 0 )
			{
#pragma unroll
				for( int t = 0; t < ( NTC > 0 ? NTC : 1 ); t++ )
				{
					w[ t ] = in[ t ];
				}

				__builtin_amdgcn_sched_barrier( 0 );
			}

			o = gp_dot< LANC, f4, NTC >( nt,
				[&]( const int t ) { return( NTC > 0 ?
					w[ NTC > 0 ? t % ( NTC > 0 ? NTC : 1 ) : 0 ] : in[ t ]); },
				[&]( const int b ) -> f4
				{
					f4 c;

					if( NTC > 0 )
					{
						c.x = cf[ ( 4 * b ) % ( NTC > 0 ? NTC : 1 )];
						c.y = cf[ ( 4 * b + 1 ) % ( NTC > 0 ? NTC : 1 )];
						c.z = cf[ ( 4 * b + 2 ) % ( NTC > 0 ? NTC : 1 )];
						c.w = cf[ ( 4 * b + 3 ) % ( NTC > 0 ? NTC : 1 )];
					}
					else
					{
						// (rows beyond nt hold whatever: never multiplied)
						c.x = cfl[ ( 4 * b ) * 64 + lane ];
						c.y = cfl[ ( 4 * b + 1 ) * 64 + lane ];
						c.z = cfl[ ( 4 * b + 2 ) * 64 + lane ];
						c.w = cfl[ ( 4 * b + 3 ) * 64 + lane ];
					}

					return( c );
				}, A.lanc, true );
		}

		if( MODE == GP_POST )
		{
			if( gok )
			{
				mbuf[ lane ] = o;
			}

			if( jok )
			{
				// out[j] = FIR7 over G[ clamp( j - e +- i ) ]
				const int glast = A.g_len - 1;
				const int c = j - A.e;
				o = A.f[ 0 ] * mbuf[ gp_clamp( c, glast ) - mlo ];
				o = o + A.f[ 1 ] * ( mbuf[ gp_clamp( c + 1, glast ) - mlo ] +
					mbuf[ gp_clamp( c - 1, glast ) - mlo ]);
				o = o + A.f[ 2 ] * ( mbuf[ gp_clamp( c + 2, glast ) - mlo ] +
					mbuf[ gp_clamp( c - 2, glast ) - mlo ]);
				o = o + A.f[ 3 ] * ( mbuf[ gp_clamp( c + 3, glast ) - mlo ] +
					mbuf[ gp_clamp( c - 3, glast ) - mlo ]);
			}
		}

		if( LANC && P.lout.on != 0 )
		{
			if( jok )
			{
				gp_store_lancir( P.lout, r - P.dst_row0, j, o );
			}
		}
		else
		if( jok )
		{
			*(f4*) ( P.dst + (long) ( r - P.dst_row0 ) * P.dst_ss +
				(long) j * 4 ) = o;
		}

		if( SRC != 0 && r + 1 < r1 )
		{
			// row r + 1 (loaded a row ago) into its buffer, row r + 2 out
			t_store(( buf + 1 ) & ( GH_NB - 1 ));

			if( r + 2 < r1 )
			{
				t_load( r + 2 );
			}
		}
	}
}

// ---------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------

// sacc.hip: the streaming-accumulation kernel of downsizing axes
struct SAData;
int sacc_prepare_axis( const GPAxisHost& H, std::vector< void* >& allocs,
	SAData** out );
void sacc_release_axis( SAData* D );
int sacc_run_axis( const SAData* D, bool rows, const void* src, int src_type,
	int src_ch, int live_ch, long s_lane, long s_step, float* dst, long d_lane,
	long d_step, int lane_lo, int lane_hi, int out_lo, int out_hi,
	hipStream_t st, const GPOut* out = nullptr );

struct GPData
{
	GPAxisHost h, v;     // AVIR: H pass first; LANCIR: V pass first
	SAData* sa_h;        // downsizing axes: by-sample tables (sacc.hip), or
	SAData* sa_v;        // nullptr where the gather kernels run the axis
	bool h_geom, v_geom; // the gather kernels' launch geometry exists
	int* d_hseg;
	int h_nstrips, h_ow, h_nseg, h_sbuf, h_mbuf;
	int v_blk, v_rs, v_rc;
	float* mid;          // intermediate image between the passes
	size_t mid_bytes;
	std::vector< void* > allocs;
};

static int next_pow2( int v )
{
	int p = 2;
	while( p < v ) p <<= 1;
	return( p );
}

// Recognises PRE / POST / NONE in a lowered AVIR axis.
static bool match_avir_axis( const LAxis& ax, GPAxisHost& H )
{
	H.ok = false;
	const LOp* fir = nullptr;
	const LOp* gat = nullptr;
	int mode;

	if( ax.ops.size() == 2 && ax.ops[ 0 ].type == OP_FIR &&
		ax.ops[ 1 ].type == OP_GATHER )
	{
		mode = GP_PRE; fir = &ax.ops[ 0 ]; gat = &ax.ops[ 1 ];
	}
	else
	if( ax.ops.size() == 2 && ax.ops[ 0 ].type == OP_GATHER &&
		ax.ops[ 1 ].type == OP_FIR )
	{
		mode = GP_POST; gat = &ax.ops[ 0 ]; fir = &ax.ops[ 1 ];
	}
	else
	if( ax.ops.size() == 1 && ax.ops[ 0 ].type == OP_GATHER )
	{
		mode = GP_NONE; gat = &ax.ops[ 0 ];
	}
	else
	{
		return( false );
	}

	if( fir != nullptr && ( fir -> rf != 1 || fir -> lat != 3 ||
		fir -> view != VIEW_CLAMP || fir -> e < 0 || fir -> e > 3 ))
	{
		return( false );
	}

	if( gat -> view == VIEW_RAW || gat -> maxtaps < 2 || gat -> maxtaps > 64 )
	{
		return( false );
	}

	for( int j = 0; j < gat -> out_len; j++ )
	{
		if( gat -> h_ntaps[ j ] != gat -> maxtaps )
		{
			return( false ); // (a shorter row would multiply padded taps)
		}

		if( j > 0 && gat -> h_start[ j ] < gat -> h_start[ j - 1 ])
		{
			return( false );
		}
	}

	GPAxis& A = H.a;
	memset( &A, 0, sizeof( A ));
	A.mode = mode; A.lanc = 0; A.nt = gat -> maxtaps;
	A.in_len = ax.src_len;
	A.view_mmax = ( gat -> view == VIEW_ZS ? gat -> zs_mmax : 0x7fffffff );
	A.g_len = gat -> out_len;
	A.out_len = ax.dst_len;
	A.e = ( fir != nullptr ? fir -> e : 0 );
	A.c_len = 0;

	if( mode == GP_PRE )
	{
		// (the window algebra below is written for e == lat == 3: C[m] reads
		// S[clamp(m) - 6 .. clamp(m)])
		if( fir -> e != 3 || fir -> in_len != ax.src_len ||
			gat -> in_len != fir -> out_len ||
			fir -> out_len != ax.src_len + 2 * fir -> e )
		{
			return( false );
		}

		A.c_len = fir -> out_len;
	}
	else
	{
		if( gat -> in_len != ax.src_len )
		{
			return( false );
		}

		if( mode == GP_POST && ( fir -> in_len != gat -> out_len ||
			fir -> out_len != ax.dst_len ))
		{
			return( false );
		}
	}

	if( fir != nullptr )
	{
		for( int i = 0; i < 4; i++ )
		{
			A.f[ i ] = fir -> h_flt[ i ];
		}
	}

	A.start = gat -> d_start;
	A.coef = gat -> d_coef;
	A.fidx = nullptr;
	H.start = gat -> h_start;
	H.h_coef = gat -> h_coef.data();
	H.h_fidx = nullptr;
	H.ok = true;
	return( true );
}

static void match_lancir_axis( const LancirAxisDev& L, int order,
	GPAxisHost& H )
{
	GPAxis& A = H.a;
	memset( &A, 0, sizeof( A ));
	H.ok = ( L.kernel_len >= 2 && L.kernel_len <= 64 &&
		( L.kernel_len & 1 ) == 0 );

	A.mode = GP_NONE; A.lanc = ( order >= 1 && order <= 3 ? order : 4 );
	A.nt = L.kernel_len;
	A.in_len = L.src_len; A.view_mmax = 0x7fffffff;
	A.g_len = L.dst_len; A.out_len = L.dst_len; A.e = 0;
	A.start = L.d_start; A.coef = L.d_flt; A.fidx = L.d_fidx;
	H.start = L.h_start;
	H.h_coef = L.h_flt.data();
	H.h_fidx = L.h_fidx.data();

	for( size_t j = 1; j < H.start.size(); j++ )
	{
		H.ok &= ( H.start[ j ] >= H.start[ j - 1 ]);
	}
}

bool fused_dn_both( const avirhip_plan* p ); // fused.hip

int gpass_prepare( avirhip_plan* p )
{
	p -> gpass = nullptr;

	if( p -> ch != 4 || getenv( "AVIRHIP_NO_GPASS" ) != nullptr )
	{
		return( AVIRHIP_OK );
	}

	GPData* D = new GPData();
	D -> mid = nullptr; D -> mid_bytes = 0; D -> d_hseg = nullptr;
	D -> sa_h = nullptr; D -> sa_v = nullptr;
	D -> h_geom = false; D -> v_geom = false;

	if( p -> is_lancir )
	{
		// (l_order: the channel count whose summation order applies -- the
		// padded RGBA inner plan of a 1-3 channel image keeps the image's)
		match_lancir_axis( p -> lv, p -> l_order, D -> v );
		match_lancir_axis( p -> lh, p -> l_order, D -> h );

		if( p -> l_order != 4 && ( p -> lv.kernel_len < 4 ||
			p -> lh.kernel_len < 4 ))
		{
			D -> h.ok = false;
		}

		// the float pipeline only: integer I/O and the OutMul / clamp output
		// stage stay with the generic kernels
		if( p -> in_type != AVIRHIP_F32 || p -> out_type != AVIRHIP_F32 ||
			!p -> l_unity )
		{
			D -> h.ok = false;
		}
	}
	else
	{
		match_avir_axis( p -> h, D -> h );
		match_avir_axis( p -> v, D -> v );
	}

	if( !D -> h.ok || !D -> v.ok )
	{
		delete D;
		return( AVIRHIP_OK );
	}

	{
		int rc = sacc_prepare_axis( D -> h, D -> allocs, &D -> sa_h );
		if( rc == 0 ) rc = sacc_prepare_axis( D -> v, D -> allocs, &D -> sa_v );

		if( rc != 0 )
		{
			p -> gpass = D;
			gpass_release( p );
			return( rc );
		}
	}

	// ---- horizontal pass geometry: strips of `ow` outputs whose gather
	// inputs (and, PRE, their FIR inputs) fit 64-lane pieces
	{
		const GPAxis& A = D -> h.a;
		const std::vector< int >& st = D -> h.start;
		// Strip width: every candidate that fits is priced by the pixels a
		// wave moves and the lanes it keeps busy per output -- LDS-DMA pieces
		// are 64 pixels whatever the segment needs (a 67-pixel segment costs
		// two; the H pass of 1920 -> 2500 moved 88 MB for a 33 MB source with
		// 64-output strips), and a strip narrower than the wave idles lanes.
		int ow = 0;
		std::vector< int > seg;
		int nstrips = 0, maxs = 0, maxm = 0;
		double best_cost = 0.0;

		for( int cw = ( A.mode == GP_POST ? 58 : 64 ); cw >= 8; cw-- )
		{
			const int cn = ( A.out_len + cw - 1 ) / cw;
			std::vector< int > cseg( (size_t) cn * 4, 0 );
			int cs = 0, cm = 0;

			for( int s = 0; s < cn; s++ )
			{
				const int ja = s * cw;
				const int jb = std::min( A.out_len, ja + cw ) - 1;
				int ga = ja, gb = jb;

				if( A.mode == GP_POST )
				{
					ga = std::max( 0, std::min( ja - A.e - 3, A.g_len - 1 ));
					gb = std::max( 0, std::min( jb - A.e + 3, A.g_len - 1 ));
				}

				// gather input range
				int ia = st[ ga ], ib = st[ gb ] + A.nt - 1;
				int sa = ia, sb = ib, ma = ga, mb = gb;

				if( A.mode == GP_PRE )
				{
					// C[m] reads S[clamp(m) - 6 .. clamp(m)]
					ma = ia; mb = ib;
					sa = std::max( 0, std::min( ia, A.c_len - 1 )) - 2 * A.e;
					sb = std::max( 0, std::min( ib, A.c_len - 1 ));
				}

				cseg[ s * 4 ] = sa; cseg[ s * 4 + 1 ] = sb - sa + 1;
				cseg[ s * 4 + 2 ] = ma; cseg[ s * 4 + 3 ] = mb - ma + 1;
				cs = std::max( cs, sb - sa + 1 );
				cm = std::max( cm, mb - ma + 1 );
			}

			// the FIR and POST stages run one buffer entry per lane and pass
			if( cs > 256 || ( A.mode != GP_NONE && cm > 64 ))
			{
				continue;
			}

			const double cost = ((( cs + 63 ) / 64 ) * 64 + 64 ) / (double) cw;

			if( ow == 0 || cost < best_cost )
			{
				best_cost = cost;
				ow = cw; nstrips = cn; maxs = cs; maxm = cm;
				seg.swap( cseg );
			}
		}

		if( ow == 0 && D -> sa_h == nullptr )
		{
			p -> gpass = D;
			gpass_release( p );
			return( AVIRHIP_OK );
		}

		D -> h_geom = ( ow != 0 );
		D -> h_ow = ow;
		D -> h_nstrips = nstrips;
		D -> h_nseg = ( maxs + 63 ) / 64;
		D -> h_sbuf = D -> h_nseg * 64;
		D -> h_mbuf = std::max( 64, (( maxm + 63 ) / 64 ) * 64 );
		void* q = nullptr;
		if( ow != 0 )
		{
		AVIRHIP_HIPCHECK( hipMalloc( &q, seg.size() * sizeof( int )));
		D -> allocs.push_back( q );
		AVIRHIP_HIPCHECK( hipMemcpy( q, seg.data(), seg.size() * sizeof( int ),
			hipMemcpyHostToDevice ));
		D -> d_hseg = (int*) q;
		}
	}

	// ---- vertical pass geometry: LDS rings hold what the gathers address
	// dynamically, one window plus the row being written
	{
		const GPAxis& A = D -> v.a;
		int rs = 2, rc = 2;

		// tap counts the register-window kernels exist for (launch_gv): no
		// window in LDS at all
		bool rw;

		if( A.lanc )
		{
			rw = ( A.nt >= 6 && A.nt <= 24 && ( A.nt & 1 ) == 0 );
		}
		else
		if( A.mode == GP_PRE )
		{
			rw = ( A.nt == 12 );
		}
		else
		{
			rw = ( A.mode == GP_POST && A.nt >= 13 && A.nt <= 25 );
		}

		if( rw )
		{
			rc = ( A.mode == GP_POST ? 8 : 2 );
		}
		else
		if( A.mode == GP_PRE )
		{
			rc = next_pow2( A.nt + 1 );
		}
		else
		{
			// the window, the row pair being consumed and GV_PF pairs in flight
			rs = next_pow2( A.nt + 2 * GV_PF + 3 );
			rc = ( A.mode == GP_POST ? 8 : 2 );
		}

		// (+ the PRE landing queue: 5 pair slots and one dump slot)
		D -> v_geom = (( rs + rc ) * 512 + GV_QB + 6 * 1024 <= 64 * 1024 );

		if( !D -> v_geom && D -> sa_v == nullptr )
		{
			p -> gpass = D;
			gpass_release( p );
			return( AVIRHIP_OK );
		}

		D -> v_blk = 1; D -> v_rs = rs; D -> v_rc = rc;
	}

	// row offsets travel as 32-bit byte offsets
	const long mid_w = ( p -> is_lancir ? p -> src_w : p -> new_w );
	const long mid_h = ( p -> is_lancir ? p -> new_h : p -> src_h );

	if( (long) p -> src_h * p -> src_stride * 4 >= ( 1L << 31 ) ||
		(long) p -> src_h * p -> src_w * 16 >= ( 1L << 31 ) ||
		mid_h * mid_w * 16 >= ( 1L << 31 ))
	{
		p -> gpass = D;
		gpass_release( p );
		return( AVIRHIP_OK );
	}

	D -> mid_bytes = (size_t) mid_w * mid_h * 16;
	p -> gpass = D;
	return( AVIRHIP_OK );
}

void gpass_release( avirhip_plan* p )
{
	GPData* D = (GPData*) p -> gpass;

	if( D != nullptr )
	{
		for( size_t i = 0; i < D -> allocs.size(); i++ )
		{
			(void) hipFree( D -> allocs[ i ]);
		}

		sacc_release_axis( D -> sa_h );
		sacc_release_axis( D -> sa_v );
		delete D;
	}

	p -> gpass = nullptr;
}

bool gpass_ok( const avirhip_plan* p )
{
	return( p -> gpass != nullptr );
}

// Whether the pass kernels are the fastest path for this plan (measured on
// MI355X, profiles/r02_gpass): LANCIR always; AVIR when both axes upsize (FIR7
// then 12-tap gather) from a float RGBA source -- integer sources keep the
// tiled kernels, whose loader converts them in place of a pack pass, and the
// long-tap downsizing chains are still faster there too.
bool gpass_preferred( const avirhip_plan* p )
{
	const GPData* D = (const GPData*) p -> gpass;

	if( D == nullptr )
	{
		return( false );
	}

	if( p -> is_lancir )
	{
		return( true );
	}

	// both axes downsize by 2 or more: the streaming-accumulation kernel
	// (any source type: its first pass converts)
	// (measured, profiles/r02_sacc: it wins where its first pass replaces the
	// pack pass or a long-tap tile loader -- integer or narrower sources; for
	// float RGBA the gather / whole-ratio kernels stay ahead)
	if( D -> sa_h != nullptr && D -> sa_v != nullptr && !p -> gamma &&
		( p -> in_type == AVIRHIP_U8 || p -> in_type == AVIRHIP_U16 ||
		( p -> in_type == AVIRHIP_F32 && p -> io_ch != 4 )))
	{
		return( true );
	}

	// both axes upsize: float RGBA sources always; sources that need the pack
	// pass (integer, 1-3 channels, gamma) once the frame is large enough for
	// two more launches not to matter -- 1920x1080 -> 2500x1400 RGB u8: 0.068
	// against 0.094 ms on the tiles (whose loader converts in place), but
	// 640x480 -> 1024x768: 0.036 against 0.032
	// both axes downsize by 1 < k < 2 (zero-stuffed 13 .. 25-tap gathers, then
	// the correction FIR): ahead of the tiles on large frames (3840x2160 ->
	// 2560x1440 RGBA float 0.137 against 0.179 ms), even with them at 1080p
	if( D -> h.a.mode == GP_POST && D -> v.a.mode == GP_POST &&
		D -> sa_h == nullptr && D -> sa_v == nullptr )
	{
		return( (long) p -> new_w * p -> new_h >= 2000000L );
	}

	if( D -> h.a.mode != GP_PRE || D -> v.a.mode != GP_PRE )
	{
		return( false );
	}

	if( p -> in_type == AVIRHIP_F32 && p -> io_ch == 4 && !p -> gamma )
	{
		return( true );
	}

	// integer / narrow pixels on both sides without gamma: the two pass
	// kernels read and write the caller's images themselves -- two launches,
	// ahead of the tiles at every size (640x480 -> 1024x768 RGB u8: 0.027
	// against 0.032 ms)
	const bool raw_in = ( !p -> gamma && ( p -> in_type == AVIRHIP_U8 ||
		p -> in_type == AVIRHIP_U16 || p -> in_type == AVIRHIP_F32 ) &&
		D -> h_geom && D -> h_nseg <= 4 );

	const bool fused_out = ( !p -> gamma && (( p -> dither == AVIRHIP_DITHER_DEF &&
		( p -> out_type == AVIRHIP_U8 || p -> out_type == AVIRHIP_U16 )) ||
		p -> out_type == AVIRHIP_F32 ));

	return(( raw_in && fused_out ) ||
		(long) p -> new_w * p -> new_h >= 2000000L );
}

// Rows per work item. The vertical pass kernel is bound by vector-ALU issue,
// not by memory (measured: removing the DMA, its waits or the stores changes a
// launch by < 10 %): a launch takes as long as its busiest SIMD, that is
// ceil( waves / 1024 SIMDs ) waves of ( chunk + warm ) rows each, where `warm`
// = the rows of work a chunk spends before its first output. A single wave
// per SIMD has nobody to hide its latencies behind (x 1.26, measured), and
// more waves than the chip holds (`wpc` per CU) run in residency rounds.
// Fits the k_gv sweep profiles/r02_gpass/chunk_sweep.txt within 10 %.
static int balanced_chunk( int rows, int nstrips, int min_chunk, int max_chunk,
	int warm, int wpc, bool valu_bound )
{
	const double cap = 256.0 * wpc;
	double best = -1.0;
	int best_chunk = std::max( 1, std::min( rows, max_chunk ));
	min_chunk = std::max( 1, std::min( min_chunk, best_chunk ));

	for( int nch = 1; nch <= rows; nch++ )
	{
		const int chunk = ( rows + nch - 1 ) / nch;

		if( chunk > max_chunk )
		{
			continue;
		}

		if( chunk < min_chunk )
		{
			break;
		}

		const int n = ( rows + chunk - 1 ) / chunk;
		const double items = (double) n * nstrips;
		const double res = ceil( items / cap ); // residency rounds
		const double per = ceil( items / res / 1024.0 ); // waves per SIMD
		// (k_gh is not ALU-bound: there whole rounds of resident waves is
		// the model that fits -- 4661 waves on 4608 slots ran as long as 9216)
		const double cost = ( valu_bound ?
			res * per * ( chunk + (double) warm ) * ( per < 2.0 ? 1.26 : 1.0 ) :
			res * cap / items * ( chunk + (double) warm ) / chunk );

		if( best < 0.0 || cost < best )
		{
			best = cost;
			best_chunk = chunk;
		}
	}

	return( best_chunk );
}

// Register-window variants exist for the tap counts that occur: 12 for AVIR
// upsizing (FIR7 then gather), 18-24 for AVIR 1 < k < 2 downsizing (gather
// then FIR7), the even counts 6-24 for LANCIR; the LDS-window kernel takes
// everything else.
#define GV_LAUNCH( MODE, LANC, NT, RW ) do { if( wp == 2 ) \
	hipLaunchKernelGGL(( k_gv< MODE, LANC, NT, RW, 2 > ), dim3( items ), \
	dim3( 64 ), lds, st, P ); else \
	hipLaunchKernelGGL(( k_gv< MODE, LANC, NT, RW, 1 > ), dim3( items ), \
	dim3( 64 ), lds, st, P ); } while( 0 )

static void launch_gv( const GVParams& P, int wp, int items, size_t lds,
	hipStream_t st )
{
	const int nt = P.ax.nt;

	if( P.ax.lanc )
	{
		switch( nt )
		{
			case 6: GV_LAUNCH( GP_NONE, true, 6, 8 ); break;
			case 8: GV_LAUNCH( GP_NONE, true, 8, 8 ); break;
			case 10: GV_LAUNCH( GP_NONE, true, 10, 16 ); break;
			case 12: GV_LAUNCH( GP_NONE, true, 12, 16 ); break;
			case 14: GV_LAUNCH( GP_NONE, true, 14, 16 ); break;
			case 16: GV_LAUNCH( GP_NONE, true, 16, 16 ); break;
			case 18: GV_LAUNCH( GP_NONE, true, 18, 32 ); break;
			case 20: GV_LAUNCH( GP_NONE, true, 20, 32 ); break;
			case 22: GV_LAUNCH( GP_NONE, true, 22, 32 ); break;
			case 24: GV_LAUNCH( GP_NONE, true, 24, 32 ); break;
			default: GV_LAUNCH( GP_NONE, true, 0, 0 ); break;
		}
	}
	else
	if( P.ax.mode == GP_PRE )
	{
		if( nt == 12 ) GV_LAUNCH( GP_PRE, false, 12, 16 );
		else GV_LAUNCH( GP_PRE, false, 0, 0 );
	}
	else
	if( P.ax.mode == GP_POST )
	{
		switch( nt )
		{
			// (13 .. 25 taps: the 1 < k < 2 plans, avir.h bank lengths 26 .. 50
			// over the zero-stuffed view)
			case 13: GV_LAUNCH( GP_POST, false, 13, 16 ); break;
			case 14: GV_LAUNCH( GP_POST, false, 14, 16 ); break;
			case 15: GV_LAUNCH( GP_POST, false, 15, 16 ); break;
			case 16: GV_LAUNCH( GP_POST, false, 16, 16 ); break;
			case 17: GV_LAUNCH( GP_POST, false, 17, 32 ); break;
			case 18: GV_LAUNCH( GP_POST, false, 18, 32 ); break;
			case 19: GV_LAUNCH( GP_POST, false, 19, 32 ); break;
			case 20: GV_LAUNCH( GP_POST, false, 20, 32 ); break;
			case 21: GV_LAUNCH( GP_POST, false, 21, 32 ); break;
			case 22: GV_LAUNCH( GP_POST, false, 22, 32 ); break;
			case 23: GV_LAUNCH( GP_POST, false, 23, 32 ); break;
			case 24: GV_LAUNCH( GP_POST, false, 24, 32 ); break;
			case 25: GV_LAUNCH( GP_POST, false, 25, 32 ); break;
			default: GV_LAUNCH( GP_POST, false, 0, 0 ); break;
		}
	}
	else
	{
		GV_LAUNCH( GP_NONE, false, 0, 0 );
	}
}

#undef GV_LAUNCH

template< int MODE, bool LANC, int SRC >
static void launch_gh2( const GHParams& P, int items, size_t lds,
	hipStream_t st )
{
	const int nt = P.ax.nt;

	// (register-resident coefficient rows for the tap counts that occur: 6
	// LANCIR, 12 AVIR upsizing; raw sources only come with AVIR plans)
	if constexpr( SRC == 0 )
	{
		if( nt == 6 )
		{
			hipLaunchKernelGGL(( k_gh< MODE, LANC, 6, 0 > ), dim3( items ),
				dim3( 64 ), lds, st, P );
			return;
		}
	}

	if constexpr( SRC == 0 || MODE == GP_PRE )
	{
		if( nt == 12 )
		{
			hipLaunchKernelGGL(( k_gh< MODE, LANC, 12, SRC > ), dim3( items ),
				dim3( 64 ), lds, st, P );
			return;
		}
	}

	hipLaunchKernelGGL(( k_gh< MODE, LANC, 0, SRC > ), dim3( items ),
		dim3( 64 ), lds, st, P );
}

// src_kind: 0 float RGBA (LDS-DMA); 1 uint8, 2 uint16, 3 float pixels read raw
template< int MODE, bool LANC >
static void launch_gh( const GHParams& P, int src_kind, int items, size_t lds,
	hipStream_t st )
{
	switch( LANC ? 0 : src_kind )
	{
		case 1: launch_gh2< MODE, LANC, ( LANC ? 0 : 1 ) >( P, items, lds, st );
			break;
		case 2: launch_gh2< MODE, LANC, ( LANC ? 0 : 2 ) >( P, items, lds, st );
			break;
		case 3: launch_gh2< MODE, LANC, ( LANC ? 0 : 3 ) >( P, items, lds, st );
			break;
		default: launch_gh2< MODE, LANC, 0 >( P, items, lds, st ); break;
	}
}

static int run_v( const GPData* D, const float* src, long src_ss, int width,
	float* dst, long dst_ss, int dst_row0, int row_lo, int row_hi,
	hipStream_t st, const GPOut* out = nullptr, const void* raw = nullptr,
	int raw_type = 0, int raw_ch = 0, long raw_stride = 0 )
{
	GVParams P;
	memset( &P.out, 0, sizeof( P.out ));
	P.raw = raw; P.raw_ss = raw_stride; P.raw_ch = raw_ch;
	P.raw_kind = ( raw_type == AVIRHIP_U8 ? 1 : ( raw_type == AVIRHIP_U16 ? 2 : 3 ));
	// (rows are raw_stride elements apart; the last one ends with its pixels)
	P.raw_elems = (long) ( D -> v.a.in_len - 1 ) * raw_stride +
		(long) width * raw_ch;

	if( raw != nullptr && P.raw_elems < 4 )
	{
		return( 1 );
	}

	if( out != nullptr )
	{
		P.out = *out;
	}

	P.src = src; P.src_ss = src_ss; P.width = width;
	P.dst = dst; P.dst_ss = dst_ss; P.dst_row0 = dst_row0;
	P.ax = D -> v.a;
	P.row_lo = row_lo; P.row_hi = row_hi;
	P.dbg = ( getenv( "AVIRHIP_GP_DBG" ) != nullptr ? atoi( getenv( "AVIRHIP_GP_DBG" )) : 0 );
	P.blk = D -> v_blk; P.rs = D -> v_rs; P.rc = D -> v_rc;

	// whole-pixel lanes (64-pixel strips) wherever the image is wide enough
	// to fill them and the rings stay within 64 KiB
	int wp = ( width > 48 && ( P.rs + P.rc ) * 1024 + GV_QB + 6 * 1024 <= 64 * 1024 ?
		2 : 1 );

	if( getenv( "AVIRHIP_GV_WP" ) != nullptr ) // tuning aid
	{
		wp = ( atoi( getenv( "AVIRHIP_GV_WP" )) == 2 &&
			( P.rs + P.rc ) * 1024 + GV_QB + 6 * 1024 <= 64 * 1024 ? 2 : 1 );
	}

	if( out != nullptr || raw != nullptr )
	{
		// the integer output stage and the raw-source loader handle whole
		// pixels
		if(( P.rs + P.rc ) * 1024 + GV_QB + 6 * 1024 > 64 * 1024 )
		{
			return( 1 );
		}

		wp = 2;
	}

	const int rowb = wp * 512;
	P.nstrips = ( width + 32 * wp - 1 ) / ( 32 * wp );

	// chunk height: a chunk's first window (NT rows, + 6 of FIR warm-up) is
	// fed before its first output; its coefficient rows and positions live
	// in LDS (<= 6 KiB)
	const int rows = row_hi - row_lo;
	const int ntp = ( P.ax.nt + 3 ) & ~3;
	const int gextra = ( P.ax.mode == GP_POST ? 7 : 0 );
	const int maxg = std::max( gextra + 1, 6144 / ( ntp * 4 + 4 ));
	// warm-up in units of output rows (a fed row costs about a third of one)
	const int warm_div = ( getenv( "AVIRHIP_GV_WARMDIV" ) != nullptr ?
		std::max( 1, atoi( getenv( "AVIRHIP_GV_WARMDIV" ))) : 3 );

	const int warm = std::max( 1, (int) ( (long) ( P.ax.nt + 6 ) *
		std::max( 1, P.ax.out_len ) / std::max( 1, P.ax.in_len ) / warm_div ));

	const size_t lds_guess = (size_t) ( P.rs + P.rc ) * rowb + GV_QB + 3072;
	const int wpc = std::max( 1, std::min( 16, (int) ( 160 * 1024 /
		( lds_guess + 512 ))));

	P.chunk = balanced_chunk( rows, P.nstrips, 8, maxg - gextra, warm, wpc,
		true );

	if( getenv( "AVIRHIP_GV_CHUNK" ) != nullptr ) // tuning aid
	{
		P.chunk = std::max( 1, std::min( atoi( getenv( "AVIRHIP_GV_CHUNK" )),
			maxg - gextra ));
	}

	const int nch = ( rows + P.chunk - 1 ) / P.chunk;
	const int items = P.nstrips * nch;
	const size_t lds = (size_t) ( P.rs + P.rc ) * rowb + GV_QB +
		(size_t) ( P.chunk + gextra ) * ( ntp * 4 + 4 ) + 64;

#ifdef GP_DBG
	{
		static int once = 0;
		if( !once++ ) fprintf( stderr, "k_gv: wp %d strips %d chunk %d items %d "
			"lds %zu nt %d mode %d\n", wp, P.nstrips, P.chunk, items, lds,
			P.ax.nt, P.ax.mode );
	}
#endif
	launch_gv( P, wp, items, lds, st );

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

static int run_h( const GPData* D, const float* src, long src_ss, int src_w,
	float* dst, long dst_ss, int dst_row0, int dst_w, int row_lo, int row_hi,
	hipStream_t st, const void* raw = nullptr, int raw_type = 0,
	int raw_ch = 0, long raw_stride = 0, const GPLOut* lout = nullptr )
{
	GHParams P;
	memset( &P.lout, 0, sizeof( P.lout ));

	if( lout != nullptr )
	{
		P.lout = *lout;
	}

	P.raw = raw; P.raw_ss = raw_stride; P.raw_ch = raw_ch;
	const int src_kind = ( raw == nullptr ? 0 : ( raw_type == AVIRHIP_U8 ? 1 :
		( raw_type == AVIRHIP_U16 ? 2 : 3 )));

	if( raw != nullptr && ( D -> h_nseg > 4 || D -> h.a.lanc ))
	{
		return( 1 );
	}

	P.raw_elems = (long) ( row_hi - 1 ) * raw_stride + (long) src_w * raw_ch;

	if( raw != nullptr && P.raw_elems < 4 )
	{
		return( 1 );
	}

	P.src = src; P.src_ss = src_ss; P.src_w = src_w;
	P.dst = dst; P.dst_ss = dst_ss; P.dst_row0 = dst_row0; P.dst_w = dst_w;
	P.ax = D -> h.a;
	P.row_lo = row_lo; P.row_hi = row_hi;
	P.nstrips = D -> h_nstrips; P.ow = D -> h_ow;
	P.seg = D -> d_hseg; P.nseg = D -> h_nseg;
	P.sbuf = D -> h_sbuf; P.mbuf = D -> h_mbuf;

	const int rows = row_hi - row_lo;
	// (tap counts 6 and 12 keep the coefficient row in registers: no LDS table)
	const bool regcoef = ( P.ax.nt == 6 || P.ax.nt == 12 );
	const size_t lds_h = (size_t) ( GH_NB * P.sbuf + P.mbuf ) * 16 +
		( regcoef ? 0 : (size_t) (( P.ax.nt + 3 ) & ~3 ) * 64 * 4 );

	const int wpc = std::max( 1, std::min( 16, (int) ( 160 * 1024 /
		( lds_h + 512 ))));

	// (a chunk's fixed cost: its coefficient row load and three rows of DMA
	// lead, about two rows of work)
	P.chunk = balanced_chunk( rows, P.nstrips, 8, rows, 2, wpc, false );
	const int nch = ( rows + P.chunk - 1 ) / P.chunk;
	const int items = P.nstrips * nch;
	const size_t lds = lds_h;
	const GPAxis& A = P.ax;

	if( A.lanc )
	{
		launch_gh< GP_NONE, true >( P, 0, items, lds, st );
	}
	else
	if( A.mode == GP_PRE )
	{
		launch_gh< GP_PRE, false >( P, src_kind, items, lds, st );
	}
	else
	if( A.mode == GP_POST )
	{
		launch_gh< GP_POST, false >( P, src_kind, items, lds, st );
	}
	else
	{
		launch_gh< GP_NONE, false >( P, src_kind, items, lds, st );
	}

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

// Source rows (AVIR: of the H pass' output; LANCIR: unused) the vertical pass
// reads for output rows [row0, row1).
static void v_source_rows( const GPData* D, int row0, int row1, int& a, int& b )
{
	const GPAxis& A = D -> v.a;
	const std::vector< int >& st = D -> v.start;
	int ga = row0, gb = row1 - 1;

	if( A.mode == GP_POST )
	{
		ga = std::max( 0, std::min( row0 - A.e - 3, A.g_len - 1 ));
		gb = std::max( 0, std::min( row1 - 1 - A.e + 3, A.g_len - 1 ));
	}

	a = st[ ga ]; b = st[ gb ] + A.nt - 1;

	if( A.mode == GP_PRE )
	{
		a = std::max( 0, std::min( a, A.c_len - 1 )) - 2 * A.e;
		b = std::max( 0, std::min( b, A.c_len - 1 ));
	}

	a = std::max( 0, std::min( a, A.in_len - 1 ));
	b = std::max( 0, std::min( b, A.in_len - 1 ));
}

// Runs both passes for output rows [row0, row1) of a float RGBA plan. Returns
// 1 when the call cannot take this path (alignment), so that the caller falls
// back.
// LANCIR (an inner plan): whether its vertical pass -- the first one -- can
// read the owner's integer / narrower image itself (whole-pixel lanes).
bool gpass_lancir_takes_raw( const avirhip_plan* p )
{
	const GPData* D = (const GPData*) p -> gpass;
	const int lpath = ( p -> path != 0 ? p -> path : p -> auto_path );

	return( D != nullptr && p -> is_lancir && lpath == 5 && D -> v_geom &&
		D -> h_geom && p -> src_w > 48 &&
		( D -> v_rs + D -> v_rc ) * 1024 + GV_QB + 6 * 1024 <= 64 * 1024 );
}

// Whether the first pass reads the caller's image as it is (integer types,
// 1-4 channels): AVIR plans whose horizontal axis streams (sacc.hip).
bool gpass_takes_raw( const avirhip_plan* p )
{
	const GPData* D = (const GPData*) p -> gpass;
	return( D != nullptr && !p -> is_lancir && ( D -> sa_h != nullptr ||
		( D -> h_geom && D -> h_nseg <= 4 )));
}

// `iout` != nullptr: the caller's integer image (the band's first row) -- the
// last pass converts and stores there (no float result, no epilogue pass);
// returns 1 if this plan's last pass cannot (the caller then runs unfused).
int gpass_run( avirhip_plan* p, const float* src, long src_stride, float* dst,
	int row0, int row1, hipStream_t st, const void* raw, int raw_type,
	int raw_ch, long raw_stride, void* iout )
{
	GPOut O;
	memset( &O, 0, sizeof( O ));

	if( iout != nullptr )
	{
		if( p -> is_lancir || p -> gamma ||
			( p -> out_type != AVIRHIP_F32 &&
			p -> dither != AVIRHIP_DITHER_DEF ) ||
			( p -> out_type != AVIRHIP_U8 && p -> out_type != AVIRHIP_U16 &&
			!( p -> out_type == AVIRHIP_F32 && p -> io_ch < 4 )) ||
			( p -> out_type == AVIRHIP_U8 && p -> io_ch == 4 &&
			(( (uintptr_t) iout & 3 ) != 0 )))
		{
			return( 1 );
		}

		O.on = 1; O.type = p -> out_type; O.ch = p -> io_ch;
		O.use_tr = ( p -> tr_mul != 1.0 );
		O.tr_mul = (float) p -> tr_mul;
		O.tr_muli = (float) ( 1.0 / p -> tr_mul );
		O.pk_out = (float) p -> pk_out;
		O.base = iout;
		O.stride = (long) p -> new_w * p -> io_ch;
		dst = (float*) iout; // (alignment checks below: not written as floats)
	}

	const GPOut* const out = ( iout != nullptr ? &O : nullptr );

	GPData* D = (GPData*) p -> gpass;

	if( D == nullptr || ( iout == nullptr && ( (uintptr_t) dst & 15 )) ||
		( p -> is_lancir && ( p -> new_stride & 3 )) ||
		( raw == nullptr && (( (uintptr_t) src & 15 ) || ( src_stride & 3 ))) ||
		( raw != nullptr && !gpass_takes_raw( p )))
	{
		return( 1 );
	}

	if( row1 <= row0 )
	{
		return( AVIRHIP_OK );
	}

	if( D -> mid == nullptr )
	{
		void* q = nullptr;
		AVIRHIP_HIPCHECK( hipMalloc( &q, D -> mid_bytes ));
		D -> allocs.push_back( q );
		p -> alloc_bytes += D -> mid_bytes;
		D -> mid = (float*) q;
	}

	const long dst_ss = ( p -> is_lancir ? p -> new_stride : (long) p -> new_w * 4 );
	int rc;

	if( p -> is_lancir )
	{
		if( !D -> h_geom || !D -> v_geom )
		{
			return( 1 );
		}

		// vertical first (lancir.h:601-646): mid = [new_h][src_w], only the
		// rows of the band
		const long mid_ss = (long) p -> src_w * 4;

		if( p -> lraw.on )
		{
			// (an inner plan: the owner's image, read as it is)
			rc = run_v( D, src, src_stride, p -> src_w, D -> mid, mid_ss, 0,
				row0, row1, st, nullptr, p -> lraw.ptr, p -> lraw.type,
				p -> lraw.ch, p -> lraw.stride );

			if( rc != 0 ) return( rc );
		}
		else
		if(( rc = run_v( D, src, src_stride, p -> src_w, D -> mid, mid_ss, 0,
			row0, row1, st )) != 0 ) return( rc );

		if( p -> lfuse.on )
		{
			// (an inner plan: the outer plan's output stage goes into the
			// store of this pass, its result rows are never written)
			GPLOut L;
			memset( &L, 0, sizeof( L ));
			L.on = 1; L.type = p -> lfuse.type; L.ch = p -> lfuse.ch;
			L.unity = p -> lfuse.unity; L.out_mul = p -> lfuse.out_mul;
			L.clampv = p -> lfuse.clampv;
			L.l4 = ( p -> new_w * p -> lfuse.ch ) & ~3;
			L.base = p -> lfuse.dst; L.stride = p -> lfuse.stride;
			p -> lfuse.done = 1;
			return( run_h( D, D -> mid, mid_ss, p -> src_w, dst, dst_ss, row0,
				p -> new_w, row0, row1, st, nullptr, 0, 0, 0, &L ));
		}

		return( run_h( D, D -> mid, mid_ss, p -> src_w, dst, dst_ss, row0,
			p -> new_w, row0, row1, st ));
	}

	// AVIR: horizontal first into mid = FltBuf [src_h][new_w], only the source
	// rows the band's vertical windows read
	const long mid_ss = (long) p -> new_w * 4;
	int a, b;
	v_source_rows( D, row0, row1, a, b );

	rc = 1;

	if( D -> sa_h != nullptr )
	{
		if( raw != nullptr )
		{
			const long es = ( raw_type == AVIRHIP_U8 ? 1 :
				( raw_type == AVIRHIP_U16 ? 2 : 4 ));

			rc = sacc_run_axis( D -> sa_h, true, raw, raw_type, raw_ch, raw_ch,
				raw_stride * es, raw_ch * es, D -> mid, mid_ss, 4, a, b + 1, 0,
				p -> new_w, st );
		}
		else
		{
			rc = sacc_run_axis( D -> sa_h, true, src, AVIRHIP_F32, 4, p -> io_ch,
				src_stride * 4, 16, D -> mid, mid_ss, 4, a, b + 1, 0,
				p -> new_w, st );
		}
	}

	if( rc == 1 )
	{
		if( !D -> h_geom )
		{
			return( 1 );
		}

		rc = run_h( D, src, src_stride, p -> src_w, D -> mid, mid_ss, 0,
			p -> new_w, a, b + 1, st, raw, raw_type, raw_ch, raw_stride );
	}

	if( rc != 0 ) return( rc );

	rc = 1;

	if( D -> sa_v != nullptr )
	{
		rc = sacc_run_axis( D -> sa_v, false, D -> mid, AVIRHIP_F32, 4,
			p -> io_ch, 16, mid_ss * 4, dst - (long) row0 * dst_ss, 4, dst_ss,
			0, p -> new_w, row0, row1, st, out );
	}

	if( rc == 1 )
	{
		if( !D -> v_geom )
		{
			return( 1 );
		}

		rc = run_v( D, D -> mid, mid_ss, p -> new_w, dst, dst_ss, row0, row0,
			row1, st, out );
	}

	return( rc );
}

} // namespace avirhip
