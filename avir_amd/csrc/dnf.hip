// dnf.hip -- integer-ratio downsizing (cfg4 of BASELINE.json: 3x; also 2x), both
// axes in ONE marching launch: the horizontally resized intermediate (the
// reference's FltBuf, avir.h:4880-4979) never exists in memory.
//
// What the plan looks like (see dn.hip), per axis, (K, NT) = (2, 24) or (3, 38):
//   T[j] = 0 + sum_{t<NT} c[t] * S[clamp(K*j + c0 + t)]      (doResize,
//          avir.h:3884-4096, one bank phase)
//   O[j] = f0*T[j] + sum_{i=1..3} f_i*(T[clamp(j+i)] + T[clamp(j-i)])
//          (correction filter, doFilter avir.h:3748-3866)
// horizontal axis first (every source row -> one FltBuf row F), then the same
// along y over the F rows.
//
// Work item = strip of 42 output columns x chunk of output rows, one workgroup
// of 12 waves with two ROLES marching down the chunk's source rows 16 at a time:
//
//   8 "row" waves     each owns 2 source rows of the step. Its row segments
//       arrive by LDS-DMA one step ahead into wave-private buffers (loads
//       retire in order: vmcnt counts them across the barriers). Lane =
//       (row, triple of resize outputs, half pixel): R = 3 consecutive T
//       outputs accumulate in registers from ONE pass over NT + 2K inputs
//       (ds_read_b64 at compile-time offsets, lane stride 3K pixels: bank-
//       conflict free for K = 3; for K = 2 the DMA lands even and odd pixels
//       in two planes, which makes the stride 3 slots again), then the FIR7
//       along x -> one F row per source row into an LDS ring of 2 x 16 rows.
//   4 "column" waves  lane = ONE channel of one pixel column. Streaming
//       accumulation along y: a lane multiplies each new F value into the
//       ceil(NT/K) output rows alive at it (16 accumulator slots, static:
//       the row loop is unrolled over the 16-output period), every sum starts
//       from +0.0f and receives its taps in ascending order; a finished T row
//       enters an 8-slot register window and the FIR7 along y produces one
//       output row, stored straight to the destination (or through the
//       integer output stage). F rows of step s are consumed during step
//       s + 1: one s_barrier per step, nobody waits inside a step.
//
// Each SIMD hosts two row waves (packed multiplies / adds with the tap as an
// SGPR-pair half, op_sel) and one column wave (scalar v_mul / v_add with SGPR
// taps): equal vector-ALU time per step by construction (K = 3: 2 x 258 packed
// + 490 scalar instructions).
//
// Cost of the chunking: a chunk's first output row needs NT + 5K source rows
// before it, its strip 6 extra T columns -- (1 + 53/270) x (48/42) on the
// horizontal arithmetic for cfg4 with one workgroup per compute unit.
//
// Arithmetic contract: -ffp-contract=off, mul then add in the reference's tap
// order => bit-identical to the two-pass kernels and to the reference.

#include "gpass_dev.h"
#include "dn_dev.h"
#include <algorithm>
#include <string.h>

namespace avirhip {

#define DF_W 42     // output columns per strip
#define DF_NT1 48   // resize (T) outputs per row: 16 triples
#define DF_NHW 8    // row waves
#define DF_NVW 4    // column waves
#define DF_RS 16    // source rows per step: 2 per row wave
#define DF_LPV 42   // lanes (elements) per column wave
#define DF_NTHR (( DF_NHW + DF_NVW ) * 64 )

struct DnFParams
{
	DnAxisP H, V;
	const float* src; long src_ss;  // floats per source row
	float* dst; long dst_ss;        // float RGBA rows, row dst_row0 at dst
	int dst_row0;
	int row_lo, row_hi;             // output rows to store
	int nstrips, nchunks, crows;    // work items, strip fastest; rows per chunk
	GPOut out;                      // on != 0: integer output stage in the store
};

// one element (channel c of pixel x) through the integer output stage
// (gp_store_int's arithmetic per channel)
__device__ __forceinline__ void df_store_elem( const GPOut& O, const long row,
	const int x, const int c, const float v )
{
	if( c >= O.ch )
	{
		return;
	}

	const long e = row * O.stride + (long) x * O.ch + c;

	if( O.type == AVIRHIP_F32 )
	{
		((float*) O.base )[ e ] = v;
		return;
	}

	float t = v;

	if( O.use_tr )
	{
		t = (float) (int) ( t * O.tr_muli + 0.5f ) * O.tr_mul;
	}
	else
	{
		t = (float) (int) ( t + 0.5f );
	}

	t = fminf( fmaxf( t, 0.0f ), O.pk_out );
	const unsigned int q = (unsigned int) t;

	if( O.type == AVIRHIP_U8 )
	{
		((unsigned char*) O.base )[ e ] = (unsigned char) q;
	}
	else
	{
		((unsigned short*) O.base )[ e ] = (unsigned short) q;
	}
}

// LDS-DMA of 64 x 16 bytes with a scalar row offset (gp_dma + soffset)
__device__ __forceinline__ void df_dma( const __amdgpu_buffer_rsrc_t rs,
	const unsigned lds, const int voff, const unsigned soff )
{
	asm volatile( "s_mov_b32 m0, %0\n\t"
		"s_nop 0\n\t"
		"buffer_load_dwordx4 %1, %2, %3 offen lds"
		:: "s"( lds ), "v"( voff ), "s"( rs ), "s"( soff ) : "memory" );
}

// slot (16-byte unit) of relative input pixel i inside a row buffer of NP * 64
// slots: K = 3 as it comes; K = 2 even pixels first, then the odd ones
template< int K, int NP >
__device__ __forceinline__ constexpr int df_slot( const int i )
{
	return( K == 2 ? ( i & 1 ) * ( NP * 32 ) + ( i >> 1 ) : i );
}

// Packed multiply-adds of a row wave: one input half pixel (two channels)
// times the taps it has in 1-3 of the lane's outputs, then the adds -- one asm
// statement per input (hipcc pads every statement boundary with a wait state
// and never moves a load across one; products first, sums second: no
// instruction waits for its predecessor). A tap is one half of an SGPR pair
// {c[2m], c[2m+1]}: op_sel picks it for both result halves.
#define DF_SEL0 " op_sel_hi:[1,0]"
#define DF_SEL1 " op_sel:[0,1] op_sel_hi:[1,1]"
#define DF_PK1_BODY( S0 ) asm volatile( \
	"v_pk_mul_f32 %1, %2, %3" S0 "\n\t" \
	"v_pk_add_f32 %0, %0, %1" \
	: "+v"( a0 ), "=&v"( p0 ) : "v"( v ), "s"( c0 ))
#define DF_PK2_BODY( S0, S1 ) asm volatile( \
	"v_pk_mul_f32 %2, %4, %5" S0 "\n\t" \
	"v_pk_mul_f32 %3, %4, %6" S1 "\n\t" \
	"v_pk_add_f32 %0, %0, %2\n\t" \
	"v_pk_add_f32 %1, %1, %3" \
	: "+v"( a0 ), "+v"( a1 ), "=&v"( p0 ), "=&v"( p1 ) \
	: "v"( v ), "s"( c0 ), "s"( c1 ))
#define DF_PK3_BODY( S0, S1, S2 ) asm volatile( \
	"v_pk_mul_f32 %3, %6, %7" S0 "\n\t" \
	"v_pk_mul_f32 %4, %6, %8" S1 "\n\t" \
	"v_pk_mul_f32 %5, %6, %9" S2 "\n\t" \
	"v_pk_add_f32 %0, %0, %3\n\t" \
	"v_pk_add_f32 %1, %1, %4\n\t" \
	"v_pk_add_f32 %2, %2, %5" \
	: "+v"( a0 ), "+v"( a1 ), "+v"( a2 ), "=&v"( p0 ), "=&v"( p1 ), \
	"=&v"( p2 ) : "v"( v ), "s"( c0 ), "s"( c1 ), "s"( c2 ))

// PAR: bit k = parity of the tap of accumulator k
template< int PAR >
__device__ __forceinline__ void df_pk1( f2& a0, const f2 v, const f2 c0 )
{
	f2 p0;
	if constexpr( PAR == 0 ) DF_PK1_BODY( DF_SEL0 );
	else DF_PK1_BODY( DF_SEL1 );
}

template< int PAR >
__device__ __forceinline__ void df_pk2( f2& a0, f2& a1, const f2 v,
	const f2 c0, const f2 c1 )
{
	f2 p0, p1;
	if constexpr( PAR == 0 ) DF_PK2_BODY( DF_SEL0, DF_SEL0 );
	else if constexpr( PAR == 1 ) DF_PK2_BODY( DF_SEL1, DF_SEL0 );
	else if constexpr( PAR == 2 ) DF_PK2_BODY( DF_SEL0, DF_SEL1 );
	else DF_PK2_BODY( DF_SEL1, DF_SEL1 );
}

template< int PAR >
__device__ __forceinline__ void df_pk3( f2& a0, f2& a1, f2& a2, const f2 v,
	const f2 c0, const f2 c1, const f2 c2 )
{
	f2 p0, p1, p2;
	if constexpr( PAR == 0 ) DF_PK3_BODY( DF_SEL0, DF_SEL0, DF_SEL0 );
	else if constexpr( PAR == 1 ) DF_PK3_BODY( DF_SEL1, DF_SEL0, DF_SEL0 );
	else if constexpr( PAR == 2 ) DF_PK3_BODY( DF_SEL0, DF_SEL1, DF_SEL0 );
	else if constexpr( PAR == 3 ) DF_PK3_BODY( DF_SEL1, DF_SEL1, DF_SEL0 );
	else if constexpr( PAR == 4 ) DF_PK3_BODY( DF_SEL0, DF_SEL0, DF_SEL1 );
	else if constexpr( PAR == 5 ) DF_PK3_BODY( DF_SEL1, DF_SEL0, DF_SEL1 );
	else if constexpr( PAR == 6 ) DF_PK3_BODY( DF_SEL0, DF_SEL1, DF_SEL1 );
	else DF_PK3_BODY( DF_SEL1, DF_SEL1, DF_SEL1 );
}

// input i of a lane's three outputs: tap i - K * r of output r
template< int K, int NT, int I >
__device__ __forceinline__ void df_feed( f2 ( &acc )[ 3 ], const f2 v,
	const f2 ( &cp )[ NT / 2 ])
{
	constexpr int t0 = I, t1 = I - K, t2 = I - 2 * K;
	constexpr bool v0 = ( t0 >= 0 && t0 < NT );
	constexpr bool v1 = ( t1 >= 0 && t1 < NT );
	constexpr bool v2 = ( t2 >= 0 && t2 < NT );

	if constexpr( v0 && v1 && v2 )
	{
		df_pk3< ( t0 & 1 ) | (( t1 & 1 ) << 1 ) | (( t2 & 1 ) << 2 ) >(
			acc[ 0 ], acc[ 1 ], acc[ 2 ], v, cp[ t0 >> 1 ], cp[ t1 >> 1 ],
			cp[ t2 >> 1 ]);
	}
	else if constexpr( v0 && v1 )
	{
		df_pk2< ( t0 & 1 ) | (( t1 & 1 ) << 1 ) >( acc[ 0 ], acc[ 1 ], v,
			cp[ t0 >> 1 ], cp[ t1 >> 1 ]);
	}
	else if constexpr( v1 && v2 )
	{
		df_pk2< ( t1 & 1 ) | (( t2 & 1 ) << 1 ) >( acc[ 1 ], acc[ 2 ], v,
			cp[ t1 >> 1 ], cp[ t2 >> 1 ]);
	}
	else if constexpr( v0 )
	{
		df_pk1< ( t0 & 1 ) >( acc[ 0 ], v, cp[ t0 >> 1 ]);
	}
	else if constexpr( v2 )
	{
		df_pk1< ( t2 & 1 ) >( acc[ 2 ], v, cp[ t2 >> 1 ]);
	}
	else
	{
		static_assert( !v1, "NT >= 2K: a lone middle tap cannot happen" );
	}
}

// Scalar multiply-adds of a column wave: one F value times the taps of N of
// the outputs alive at its row (FIRST: a0's sum starts here, from +0.0f).
template< bool FIRST >
__device__ __forceinline__ void df_v1( float& a0,
	const float v, const float c0 )
{
	float p0;
	if constexpr( FIRST )
	{
		asm volatile( "v_mul_f32 %1, %3, %2\n\t"
			"v_add_f32 %0, 0, %1\n\t"
			: "=&v"( a0 ),
			"=&v"( p0 )
			: "v"( v ), "s"( c0 ) );
	}
	else
	{
		asm volatile( "v_mul_f32 %1, %3, %2\n\t"
			"v_add_f32 %0, %0, %1"
			: "+v"( a0 ),
			"=&v"( p0 )
			: "v"( v ), "s"( c0 ) );
	}
}

template< bool FIRST >
__device__ __forceinline__ void df_v5( float& a0, float& a1, float& a2, float& a3, float& a4,
	const float v, const float c0, const float c1, const float c2, const float c3, const float c4 )
{
	float p0, p1, p2, p3, p4;
	if constexpr( FIRST )
	{
		asm volatile( "v_mul_f32 %5, %11, %10\n\t"
			"v_mul_f32 %6, %12, %10\n\t"
			"v_mul_f32 %7, %13, %10\n\t"
			"v_mul_f32 %8, %14, %10\n\t"
			"v_mul_f32 %9, %15, %10\n\t"
			"v_add_f32 %0, 0, %5\n\t"
			"v_add_f32 %1, %1, %6\n\t"
			"v_add_f32 %2, %2, %7\n\t"
			"v_add_f32 %3, %3, %8\n\t"
			"v_add_f32 %4, %4, %9"
			: "=&v"( a0 ), "+v"( a1 ), "+v"( a2 ), "+v"( a3 ), "+v"( a4 ),
			"=&v"( p0 ), "=&v"( p1 ), "=&v"( p2 ), "=&v"( p3 ), "=&v"( p4 )
			: "v"( v ), "s"( c0 ), "s"( c1 ), "s"( c2 ), "s"( c3 ), "s"( c4 ) );
	}
	else
	{
		asm volatile( "v_mul_f32 %5, %11, %10\n\t"
			"v_mul_f32 %6, %12, %10\n\t"
			"v_mul_f32 %7, %13, %10\n\t"
			"v_mul_f32 %8, %14, %10\n\t"
			"v_mul_f32 %9, %15, %10\n\t"
			"v_add_f32 %0, %0, %5\n\t"
			"v_add_f32 %1, %1, %6\n\t"
			"v_add_f32 %2, %2, %7\n\t"
			"v_add_f32 %3, %3, %8\n\t"
			"v_add_f32 %4, %4, %9"
			: "+v"( a0 ), "+v"( a1 ), "+v"( a2 ), "+v"( a3 ), "+v"( a4 ),
			"=&v"( p0 ), "=&v"( p1 ), "=&v"( p2 ), "=&v"( p3 ), "=&v"( p4 )
			: "v"( v ), "s"( c0 ), "s"( c1 ), "s"( c2 ), "s"( c3 ), "s"( c4 ) );
	}
}

template< bool FIRST >
__device__ __forceinline__ void df_v6( float& a0, float& a1, float& a2, float& a3, float& a4, float& a5,
	const float v, const float c0, const float c1, const float c2, const float c3, const float c4, const float c5 )
{
	float p0, p1, p2, p3, p4, p5;
	if constexpr( FIRST )
	{
		asm volatile( "v_mul_f32 %6, %13, %12\n\t"
			"v_mul_f32 %7, %14, %12\n\t"
			"v_mul_f32 %8, %15, %12\n\t"
			"v_mul_f32 %9, %16, %12\n\t"
			"v_mul_f32 %10, %17, %12\n\t"
			"v_mul_f32 %11, %18, %12\n\t"
			"v_add_f32 %0, 0, %6\n\t"
			"v_add_f32 %1, %1, %7\n\t"
			"v_add_f32 %2, %2, %8\n\t"
			"v_add_f32 %3, %3, %9\n\t"
			"v_add_f32 %4, %4, %10\n\t"
			"v_add_f32 %5, %5, %11"
			: "=&v"( a0 ), "+v"( a1 ), "+v"( a2 ), "+v"( a3 ), "+v"( a4 ), "+v"( a5 ),
			"=&v"( p0 ), "=&v"( p1 ), "=&v"( p2 ), "=&v"( p3 ), "=&v"( p4 ), "=&v"( p5 )
			: "v"( v ), "s"( c0 ), "s"( c1 ), "s"( c2 ), "s"( c3 ), "s"( c4 ), "s"( c5 ) );
	}
	else
	{
		asm volatile( "v_mul_f32 %6, %13, %12\n\t"
			"v_mul_f32 %7, %14, %12\n\t"
			"v_mul_f32 %8, %15, %12\n\t"
			"v_mul_f32 %9, %16, %12\n\t"
			"v_mul_f32 %10, %17, %12\n\t"
			"v_mul_f32 %11, %18, %12\n\t"
			"v_add_f32 %0, %0, %6\n\t"
			"v_add_f32 %1, %1, %7\n\t"
			"v_add_f32 %2, %2, %8\n\t"
			"v_add_f32 %3, %3, %9\n\t"
			"v_add_f32 %4, %4, %10\n\t"
			"v_add_f32 %5, %5, %11"
			: "+v"( a0 ), "+v"( a1 ), "+v"( a2 ), "+v"( a3 ), "+v"( a4 ), "+v"( a5 ),
			"=&v"( p0 ), "=&v"( p1 ), "=&v"( p2 ), "=&v"( p3 ), "=&v"( p4 ), "=&v"( p5 )
			: "v"( v ), "s"( c0 ), "s"( c1 ), "s"( c2 ), "s"( c3 ), "s"( c4 ), "s"( c5 ) );
	}
}

template< bool FIRST >
__device__ __forceinline__ void df_v7( float& a0, float& a1, float& a2, float& a3, float& a4, float& a5, float& a6,
	const float v, const float c0, const float c1, const float c2, const float c3, const float c4, const float c5, const float c6 )
{
	float p0, p1, p2, p3, p4, p5, p6;
	if constexpr( FIRST )
	{
		asm volatile( "v_mul_f32 %7, %15, %14\n\t"
			"v_mul_f32 %8, %16, %14\n\t"
			"v_mul_f32 %9, %17, %14\n\t"
			"v_mul_f32 %10, %18, %14\n\t"
			"v_mul_f32 %11, %19, %14\n\t"
			"v_mul_f32 %12, %20, %14\n\t"
			"v_mul_f32 %13, %21, %14\n\t"
			"v_add_f32 %0, 0, %7\n\t"
			"v_add_f32 %1, %1, %8\n\t"
			"v_add_f32 %2, %2, %9\n\t"
			"v_add_f32 %3, %3, %10\n\t"
			"v_add_f32 %4, %4, %11\n\t"
			"v_add_f32 %5, %5, %12\n\t"
			"v_add_f32 %6, %6, %13"
			: "=&v"( a0 ), "+v"( a1 ), "+v"( a2 ), "+v"( a3 ), "+v"( a4 ), "+v"( a5 ), "+v"( a6 ),
			"=&v"( p0 ), "=&v"( p1 ), "=&v"( p2 ), "=&v"( p3 ), "=&v"( p4 ), "=&v"( p5 ), "=&v"( p6 )
			: "v"( v ), "s"( c0 ), "s"( c1 ), "s"( c2 ), "s"( c3 ), "s"( c4 ), "s"( c5 ), "s"( c6 ) );
	}
	else
	{
		asm volatile( "v_mul_f32 %7, %15, %14\n\t"
			"v_mul_f32 %8, %16, %14\n\t"
			"v_mul_f32 %9, %17, %14\n\t"
			"v_mul_f32 %10, %18, %14\n\t"
			"v_mul_f32 %11, %19, %14\n\t"
			"v_mul_f32 %12, %20, %14\n\t"
			"v_mul_f32 %13, %21, %14\n\t"
			"v_add_f32 %0, %0, %7\n\t"
			"v_add_f32 %1, %1, %8\n\t"
			"v_add_f32 %2, %2, %9\n\t"
			"v_add_f32 %3, %3, %10\n\t"
			"v_add_f32 %4, %4, %11\n\t"
			"v_add_f32 %5, %5, %12\n\t"
			"v_add_f32 %6, %6, %13"
			: "+v"( a0 ), "+v"( a1 ), "+v"( a2 ), "+v"( a3 ), "+v"( a4 ), "+v"( a5 ), "+v"( a6 ),
			"=&v"( p0 ), "=&v"( p1 ), "=&v"( p2 ), "=&v"( p3 ), "=&v"( p4 ), "=&v"( p5 ), "=&v"( p6 )
			: "v"( v ), "s"( c0 ), "s"( c1 ), "s"( c2 ), "s"( c3 ), "s"( c4 ), "s"( c5 ), "s"( c6 ) );
	}
}

// outputs jp0, jp0 - 1, ... jp0 - (N - 1) at row u of the period (taps
// u - K * jp): one statement
template< int K, int NT, int U, int JP0, int N >
__device__ __forceinline__ void df_vrow( float ( &acc )[ 16 ], const float v,
	const float ( &c )[ NT ])
{
#define DF_A( k ) acc[ (((( JP0 - ( k )) % 16 ) + 16 ) % 16 ) ]
#define DF_C( k ) c[ U - K * ( JP0 - ( k )) ]
	constexpr bool first = ( U - K * JP0 == 0 );

	if constexpr( N == 7 )
	{
		df_v7< first >( DF_A( 0 ), DF_A( 1 ), DF_A( 2 ), DF_A( 3 ), DF_A( 4 ),
			DF_A( 5 ), DF_A( 6 ), v, DF_C( 0 ), DF_C( 1 ), DF_C( 2 ),
			DF_C( 3 ), DF_C( 4 ), DF_C( 5 ), DF_C( 6 ));
	}
	else if constexpr( N == 6 )
	{
		df_v6< first >( DF_A( 0 ), DF_A( 1 ), DF_A( 2 ), DF_A( 3 ), DF_A( 4 ),
			DF_A( 5 ), v, DF_C( 0 ), DF_C( 1 ), DF_C( 2 ), DF_C( 3 ),
			DF_C( 4 ), DF_C( 5 ));
	}
	else if constexpr( N == 5 )
	{
		df_v5< first >( DF_A( 0 ), DF_A( 1 ), DF_A( 2 ), DF_A( 3 ), DF_A( 4 ),
			v, DF_C( 0 ), DF_C( 1 ), DF_C( 2 ), DF_C( 3 ), DF_C( 4 ));
	}
	else if constexpr( N > 0 )
	{
		df_v1< first >( DF_A( 0 ), v, DF_C( 0 ));
		df_vrow< K, NT, U, JP0 - 1, N - 1 >( acc, v, c );
	}
#undef DF_A
#undef DF_C
}

template< int KH, int NTH, int KV, int NTV >
__global__ void __launch_bounds__( DF_NTHR ) k_dnf( const DnFParams P )
{
	constexpr int NPH = ( KH * ( DF_NT1 - 1 ) + NTH + 63 ) / 64; // DMA pieces
	constexpr int ROWPX = NPH * 64;
	static_assert( KH != 2 || NPH == 2, "plane layout assumes two pieces" );
	static_assert(( NTH & 1 ) == 0, "tap pairs" );

	__shared__ __attribute__(( aligned( 16 ))) f4
		sS[ DF_NHW ][ 2 ][ 2 ][ ROWPX ];
	__shared__ __attribute__(( aligned( 16 ))) f4 sT[ DF_NHW ][ 2 ][ DF_NT1 ];
	__shared__ __attribute__(( aligned( 16 ))) float
		sF[ 2 ][ DF_RS ][ DF_W * 4 ];

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane( tid >> 6 );

	// XCD-aware dealing (workgroup b runs on XCD b % 8): an XCD's contiguous
	// item range holds whole chunk-rows of strips, so x-neighbours, which
	// share 53 of their source columns, share an L2
	const int nwg = gridDim.x;
	const int b = blockIdx.x;
	const int xcd = b & 7;
	const int qd = nwg >> 3;
	const int rm = nwg & 7;
	const int item = ( xcd < rm ? xcd * ( qd + 1 ) :
		rm * ( qd + 1 ) + ( xcd - rm ) * qd ) + ( b >> 3 );

	const int chunk = item / P.nstrips;
	const int strip = item - chunk * P.nstrips;

	// output rows of this chunk, the T rows (j) they need, the source rows
	const int oy0 = P.row_lo + chunk * P.crows;
	const int oy1 = min( oy0 + P.crows, P.row_hi );
	const int vlast = P.V.out_len - 1;
	const bool top = ( oy0 - 3 < 0 );
	const int jstart = max( oy0 - 3, 0 );
	const int n_u = oy1 + 3 - jstart;
	const int n_rows = KV * ( n_u - 1 ) + NTV;
	const int S = ( n_rows + DF_RS - 1 ) / DF_RS;
	const int r_first = KV * jstart + P.V.c0;

	if( wave < DF_NHW )
	{
		// ------------------------------------------------------------
		// row waves
		// ------------------------------------------------------------
		const int h = wave;
		// (constant address space: scalar loads, the taps are SGPR operands)
		gp_cfloat* const hc = (gp_cfloat*) P.H.coef;
		f2 cp[ NTH / 2 ];
#pragma unroll
		for( int m = 0; m < NTH / 2; m++ )
		{
			cp[ m ].x = hc[ 2 * m ];
			cp[ m ].y = hc[ 2 * m + 1 ];
		}

		const float f0 = hc[ NTH ], f1 = hc[ NTH + 1 ], f2_ = hc[ NTH + 2 ],
			f3 = hc[ NTH + 3 ];

		const int rho = lane >> 5;        // which of the wave's two rows
		const int tri = ( lane & 31 ) >> 1;
		const int hf = lane & 1;
		const int X0 = strip * DF_W - 3;  // first T output of the strip
		const int in0 = KH * X0 + P.H.c0; // its first input pixel
		const int olast = P.H.out_len - 1;
		const bool edge = ( X0 < 0 || X0 + DF_NT1 - 1 > olast );

		// per-lane source columns of the DMA pieces (clamped: the extended
		// view), the same for every row
		int voff[ NPH ];
#pragma unroll
		for( int q = 0; q < NPH; q++ )
		{
			const int px = ( KH == 2 ? lane * 2 + q : q * 64 + lane );
			voff[ q ] = dn_clampi( in0 + px, P.H.in_len - 1 ) * 16;
		}

		const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
			(void*) P.src, 0, 0x7fffffff, 0x00020000 );

		const unsigned row_b = (unsigned) P.src_ss * 4u;
		const unsigned lds_s = (unsigned) (unsigned long) (lds_char*)
			&sS[ h ][ 0 ][ 0 ][ 0 ];

		const int vin_last = P.V.in_len - 1;

		auto issue = [&]( const int s )
		{
			const int bf = s & 1;
#pragma unroll
			for( int rr = 0; rr < 2; rr++ )
			{
				const int row = dn_clampi( r_first + s * DF_RS + 2 * h + rr,
					vin_last );
				const unsigned soff = (unsigned) row * row_b;
#pragma unroll
				for( int q = 0; q < NPH; q++ )
				{
					df_dma( rsrc, lds_s + (( bf * 2 + rr ) * ROWPX + q * 64 ) *
						16, voff[ q ], soff );
				}
			}
		};

		issue( 0 );

		// FIR tasks of this lane: (row, pixel, half) of passes 0..2
		for( int s = 0; s < S; s++ )
		{
			if( s + 1 < S )
			{
				issue( s + 1 );

				if( NPH == 3 )
				{
					GP_WAIT_VM( 6 );
				}
				else
				{
					GP_WAIT_VM( 4 );
				}
			}
			else
			{
				GP_WAIT_VM( 0 );
			}

			const int bf = s & 1;

			// ---- resize: 3 consecutive outputs per lane
			{
				const f2* sp = (const f2*) &sS[ h ][ bf ][ rho ][ 0 ] +
					( KH == 2 ? 3 : 9 ) * 2 * tri + hf;

				// all inputs first (hipcc never moves a load across an asm
				// statement: read where they are used, every ds_read would be
				// waited for on the spot)
				f2 in[ NTH + 2 * KH ];
#pragma unroll
				for( int i = 0; i < NTH + 2 * KH; i++ )
				{
					in[ i ] = sp[ 2 * df_slot< KH, NPH >( i )];
				}

				f2 acc[ 3 ];
				acc[ 0 ] = (f2) 0.0f; acc[ 1 ] = (f2) 0.0f; acc[ 2 ] = (f2) 0.0f;

				dn_static_for< NTH + 2 * KH >( [&]( auto ic )
				{
					constexpr int i = decltype( ic )::value;
					df_feed< KH, NTH, i >( acc, in[ i ], cp );
				} );

				f2* tp = (f2*) &sT[ h ][ rho ][ 3 * tri ] + hf;
				tp[ 0 ] = acc[ 0 ];
				tp[ 2 ] = acc[ 1 ];
				tp[ 4 ] = acc[ 2 ];
			}

			// ---- T values outside the array are copies of the edge values
			// (the reference replicates the resize OUTPUT, avir.h:3227-3239)
			if( edge )
			{
				if( lane < DF_NT1 )
				{
					const int x = X0 + lane;

					if( x < 0 || x > olast )
					{
						const int xs = dn_clampi( x, olast ) - X0;
						const f4 v0 = sT[ h ][ 0 ][ xs ];
						const f4 v1 = sT[ h ][ 1 ][ xs ];
						sT[ h ][ 0 ][ lane ] = v0;
						sT[ h ][ 1 ][ lane ] = v1;
					}
				}
			}

			// ---- correction FIR along x: 2 rows x 42 pixels x 2 halves
#pragma unroll
			for( int q = 0; q < 3; q++ )
			{
				const int task = q * 64 + lane;

				if( task < 2 * DF_W * 2 )
				{
					const int r2 = ( task >= DF_W * 2 ? 1 : 0 );
					const int pp = task - r2 * DF_W * 2;
					const f2* t = (const f2*) &sT[ h ][ r2 ][ 0 ] + pp;
					f2 a = f0 * t[ 6 ];
					a = a + f1 * ( t[ 8 ] + t[ 4 ]);
					a = a + f2_ * ( t[ 10 ] + t[ 2 ]);
					a = a + f3 * ( t[ 12 ] + t[ 0 ]);
					*( (f2*) &sF[ bf ][ 2 * h + r2 ][ 0 ] + pp ) = a;
				}
			}

			__syncthreads();
		}
	}
	else
	{
		// ------------------------------------------------------------
		// column waves
		// ------------------------------------------------------------
		const int vi = wave - DF_NHW;
		const bool lane_ok = ( lane < DF_LPV );
		const int e = DF_LPV * vi + ( lane_ok ? lane : DF_LPV - 1 );
		const int x = strip * DF_W + ( e >> 2 );
		const int ch = e & 3;
		const bool st_ok = ( lane_ok && x < P.H.out_len );

		gp_cfloat* const vc = (gp_cfloat*) P.V.coef;
		float c[ NTV ];
#pragma unroll
		for( int t = 0; t < NTV; t++ )
		{
			c[ t ] = vc[ t ];
		}

		const float f0 = vc[ NTV ], f1 = vc[ NTV + 1 ], f2_ = vc[ NTV + 2 ],
			f3 = vc[ NTV + 3 ];

		float acc[ 16 ];
		float wr[ 8 ];
#pragma unroll
		for( int i = 0; i < 16; i++ )
		{
			acc[ i ] = 0.0f;
		}
#pragma unroll
		for( int i = 0; i < 8; i++ )
		{
			wr[ i ] = 0.0f;
		}

		int s = 0;
		int jl_base = 0; // local index of the output in slot 0 of this period

		while( s < S )
		{
			dn_static_for< KV >( [&]( auto phc )
			{
				constexpr int ph = decltype( phc )::value;

				if( s < S )
				{
					__syncthreads();
					float v[ DF_RS ];
#pragma unroll
					for( int i = 0; i < DF_RS; i++ )
					{
						v[ i ] = sF[ s & 1 ][ i ][ e ];
					}

					dn_static_for< DF_RS >( [&]( auto ic )
					{
						constexpr int i = decltype( ic )::value;
						constexpr int u = ph * DF_RS + i; // row of the period
						constexpr int jhi = u / KV;
						// outputs alive at this row (tap t = u - KV * j): j from
						// u / KV down while t < NTV -- two statements per row
						constexpr int na = ( NTV - 1 - ( u % KV )) / KV + 1;
						constexpr int n1 = ( na > 7 ? 7 : na );
						df_vrow< KV, NTV, u, jhi, n1 >( acc, v[ i ], c );
						df_vrow< KV, NTV, u, jhi - n1, na - n1 >( acc, v[ i ], c );

						// a T row completes with its last tap
						if constexpr(( u - ( NTV - 1 )) % KV == 0 )
						{
							constexpr int jp = ( u - ( NTV - 1 )) / KV;
							constexpr int slot = (( jp % 16 ) + 16 ) % 16;
							constexpr int ws = (( jp % 8 ) + 8 ) % 8;
							const int jl = jl_base + jp;

							if( jl >= 0 )
							{
								const int j = jstart + jl;
								float U = acc[ slot ];

								if( j > vlast )
								{
									U = wr[ ( ws + 7 ) & 7 ];
								}

								if( jl == 0 && top )
								{
#pragma unroll
									for( int k = 0; k < 8; k++ )
									{
										wr[ k ] = U;
									}
								}

								wr[ ws ] = U;
								const int y = j - 3;

								if( y >= oy0 && y < oy1 )
								{
									float a = f0 * wr[ ( ws + 5 ) & 7 ];
									a = a + f1 * ( wr[ ( ws + 6 ) & 7 ] +
										wr[ ( ws + 4 ) & 7 ]);
									a = a + f2_ * ( wr[ ( ws + 7 ) & 7 ] +
										wr[ ( ws + 3 ) & 7 ]);
									a = a + f3 * ( wr[ ws ] +
										wr[ ( ws + 2 ) & 7 ]);

									if( st_ok )
									{
										if( P.out.on != 0 )
										{
											df_store_elem( P.out,
												y - P.dst_row0, x, ch, a );
										}
										else
										{
											P.dst[ (long) ( y - P.dst_row0 ) *
												P.dst_ss + (long) x * 4 + ch ] = a;
										}
									}
								}
							}
						}
					} );

					s++;
				}
			} );

			jl_base += 16;
		}
	}
}

// ---------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------

static DnAxisP dnf_axis( const DnAxis& A )
{
	DnAxisP P;
	P.K = A.K; P.NT = A.NT; P.c0 = A.c0;
	P.in_len = A.in_len; P.out_len = A.out_len; P.coef = A.d_coef;
	return( P );
}

// Both axes in one launch: float RGBA source rows -> destination rows
// [row0, row1) (row row0 at `dst`, or through the integer output stage `out`).
// Returns 1 when this call cannot take the fused kernel.
int dn_run_hv( void* d, const float* src, long src_ss, float* dst, int row0,
	int row1, hipStream_t st, const GPOut* out )
{
	const DnData* D = (const DnData*) d;

	if( D == nullptr || !D -> hok || !D -> vok )
	{
		return( 1 );
	}

	if( row1 <= row0 )
	{
		return( AVIRHIP_OK );
	}

	// (row offsets of the DMA are 32-bit)
	if( (long) D -> v.in_len * src_ss * 4 >= 0x7fffffffL )
	{
		return( 1 );
	}

	DnFParams P;
	P.H = dnf_axis( D -> h );
	P.V = dnf_axis( D -> v );
	P.src = src; P.src_ss = src_ss;
	P.dst = dst; P.dst_ss = (long) D -> h.out_len * 4; P.dst_row0 = row0;
	P.row_lo = row0; P.row_hi = row1;
	memset( &P.out, 0, sizeof( P.out ));

	if( out != nullptr )
	{
		P.out = *out;
	}

	// one workgroup per compute unit: a chunk's warm-up (NT + 5K source rows)
	// is paid per work item, so as few, as tall chunks as fill the chip
	P.nstrips = ( D -> h.out_len + DF_W - 1 ) / DF_W;
	const int rows = row1 - row0;
	const int want = std::max( 1, D -> ncu / P.nstrips );
	P.crows = std::max( 8, ( rows + want - 1 ) / want );
	P.nchunks = ( rows + P.crows - 1 ) / P.crows;
	const dim3 grid( P.nstrips * P.nchunks );
	const int kk = D -> h.K * 10 + D -> v.K;

	switch( kk )
	{
		case 33:
			hipLaunchKernelGGL(( k_dnf< 3, 38, 3, 38 > ), grid, dim3( DF_NTHR ),
				0, st, P );
			break;
		case 22:
			hipLaunchKernelGGL(( k_dnf< 2, 24, 2, 24 > ), grid, dim3( DF_NTHR ),
				0, st, P );
			break;
		case 32:
			hipLaunchKernelGGL(( k_dnf< 3, 38, 2, 24 > ), grid, dim3( DF_NTHR ),
				0, st, P );
			break;
		case 23:
			hipLaunchKernelGGL(( k_dnf< 2, 24, 3, 38 > ), grid, dim3( DF_NTHR ),
				0, st, P );
			break;
		default:
			return( 1 );
	}

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

} // namespace avirhip
