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
#include "dnf_dev.h"
#include <algorithm>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <vector>

namespace avirhip {

#define DF_W 42     // output columns per strip
#define DF_NT1 48   // resize (T) outputs per row: 16 triples
// 12 waves per workgroup: NHW row waves + the column waves of the strip's 168
// elements. 8 + 4 (42 elements each): 16 rows per step, every SIMD hosts two
// row waves and a column wave. 9 + 3 (56 each): 18 rows per step, one SIMD
// hosts three row waves instead -- the column waves' share of the vector ALU
// shrinks with their idle lanes (K = 3: 209 -> 194 SIMD cycles per row), at
// the price of a longer static unroll of the column waves (the step must
// divide a whole number of 16-output periods: 8 phases instead of 3).
#define DF_NTHR 768
#define DF_MAXRS 18

struct DnFParams
{
	DnAxisP H, V;
	const float* src; long src_ss;  // floats per source row
	float* dst; long dst_ss;        // float RGBA rows, row dst_row0 at dst
	int dst_row0;
	int row_lo, row_hi;             // output rows to store
	int nstrips, nchunks, crows;    // work items, strip fastest; rows per chunk
	GPOut out;                      // on != 0: integer output stage in the store
	unsigned long long* clk;        // dbg & 16: per wave {total, dma wait, barrier
	                                // wait, 0} shader clocks
	int dbg;                        // timing ablations (AVIRHIP_DNF_DBG): 1 no
	                                // column arithmetic, 2 no row arithmetic,
	                                // 4 no source loads, 8 no FIR in the row waves
};

// one element (a lane's channel of its pixel, byte offset `voff` inside the
// row) through the integer output stage (gp_store_int's arithmetic per
// channel); row `row` of the caller's band
__device__ __forceinline__ void df_store_elem( const GPOut& O,
	const __amdgpu_buffer_rsrc_t rs, const int voff, const int row,
	const float v )
{
	if( O.type == AVIRHIP_F32 )
	{
		__builtin_amdgcn_raw_buffer_store_b32( __builtin_bit_cast( unsigned,
			v ), rs, voff, row * (int) O.stride * 4, 0 );
		return;
	}

	float t = v;
	float a = v;

	if( O.use_tr )
	{
		a = t * O.tr_muli;
		t = (float) (int) ( a + 0.5f ) * O.tr_mul;
	}
	else
	{
		t = (float) (int) ( t + 0.5f );
	}

	t = fminf( fmaxf( t, 0.0f ), O.pk_out );
	t = avirhip_x86_round_fix( a, t, O.pk_out );
	const unsigned int q = (unsigned int) t;

	if( O.type == AVIRHIP_U8 )
	{
		__builtin_amdgcn_raw_buffer_store_b8( (unsigned char) q, rs, voff,
			row * (int) O.stride, 0 );
	}
	else
	{
		__builtin_amdgcn_raw_buffer_store_b16( (unsigned short) q, rs, voff,
			row * (int) O.stride * 2, 0 );
	}
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

// ---- column waves: one asm statement per F row.
// A lane keeps the 16 accumulator slots (output j lives in slot j % 16) as 8
// register PAIRS {slot 4q + b, slot 4q + b + 2}: the two outputs of a pair,
// j - 2 and j, are at taps t + 2K and t of the same row, so ONE packed
// multiply (the F value broadcast, times the register pair V2[t + 2K] =
// {c[t + 2K], c[t]}) and one packed add serve both. V2 is c[] laid out for
// that, with zeros where a pair's other output is not alive (the table has
// NT + 2K entries): a dead half receives +-0 products, and every slot is
// zeroed on the row its next output starts at. A wave issues an instruction
// every ~4 cycles whatever it is, and the column waves are the workgroup's
// longest instruction streams: 13 taps = 26 scalar instructions become 14-16.
// The F values arrive as pairs of rows {row 2m, row 2m + 1} (ds_read2_b32):
// RSEL picks the half.
#define DF_RSEL0 "op_sel:[0,0] op_sel_hi:[0,1]"
#define DF_RSEL1 "op_sel:[1,0] op_sel_hi:[1,1]"
#define DF_VP_MULR0( i ) \
	"v_pk_mul_f32 %[q" #i "], %[vp], %[k" #i "] " DF_RSEL0 "\n\t"
#define DF_VP_MULR1( i ) \
	"v_pk_mul_f32 %[q" #i "], %[vp], %[k" #i "] " DF_RSEL1 "\n\t"
#define DF_VP_ADD( i ) "v_pk_add_f32 %[a" #i "], %[a" #i "], %[q" #i "]\n\t"
#define DF_VP_OUT( i ) [a##i] "+v"( a##i ), [q##i] "=&v"( q##i ),
#define DF_VP_IN( i ) [k##i] "v"( k##i ),
// pairs 1 .. N-1 (pair 0 closes the operand lists: no trailing comma)
#define DF_PR6( M ) M( 1 ) M( 2 ) M( 3 ) M( 4 ) M( 5 )
#define DF_PR7( M ) DF_PR6( M ) M( 6 )
#define DF_PR8( M ) DF_PR7( M ) M( 7 )
#define DF_VSTMT( PR, MULR ) asm volatile( \
	MULR( 0 ) PR( MULR ) DF_VP_ADD( 0 ) PR( DF_VP_ADD ) \
	: PR( DF_VP_OUT ) [a0] "+v"( a0 ), [q0] "=&v"( q0 ) \
	: PR( DF_VP_IN ) [k0] "v"( k0 ), [vp] "v"( vp ))

template< int N, int R >
__device__ __forceinline__ void df_vstmt( f2& a0, f2& a1, f2& a2, f2& a3,
	f2& a4, f2& a5, f2& a6, f2& a7, const f2 vp, const f2 k0, const f2 k1,
	const f2 k2, const f2 k3, const f2 k4, const f2 k5, const f2 k6,
	const f2 k7 )
{
	f2 q0, q1, q2, q3, q4, q5, q6, q7;

	if constexpr( N == 8 && R == 0 ) DF_VSTMT( DF_PR8, DF_VP_MULR0 );
	else if constexpr( N == 8 && R == 1 ) DF_VSTMT( DF_PR8, DF_VP_MULR1 );
	else if constexpr( N == 7 && R == 0 ) DF_VSTMT( DF_PR7, DF_VP_MULR0 );
	else if constexpr( N == 7 && R == 1 ) DF_VSTMT( DF_PR7, DF_VP_MULR1 );
	else if constexpr( N == 6 && R == 0 ) DF_VSTMT( DF_PR6, DF_VP_MULR0 );
	else if constexpr( N == 6 && R == 1 ) DF_VSTMT( DF_PR6, DF_VP_MULR1 );
	else
	{
		static_assert( N < 0, "6-8 pair registers are alive at a row" );
	}
}

// Row U of the 16-output period: the pair registers with an output alive and
// their V2 entries; the slot (if any) whose output starts at this row.
template< int K, int NT, int U >
struct DfRow
{
	int n;
	int pi[ 8 ], vi[ 8 ];
	int zslot; // -1: none

	constexpr DfRow() : n( 0 ), pi(), vi(), zslot( -1 )
	{
		for( int r = 0; r < 8; r++ )
		{
			const int slo = ( r >> 1 ) * 4 + ( r & 1 ), shi = slo + 2;
			int tlo = -1, thi = -1;

			// the alive outputs are j = U / K, U / K - 1, ... (tap U - K * j)
			for( int d = 0; d < 16; d++ )
			{
				const int j = U / K - d;
				const int t = U - K * j;

				if( t >= 0 && t < NT )
				{
					const int sl = (( j % 16 ) + 16 ) % 16;
					if( sl == slo ) tlo = t;
					if( sl == shi ) thi = t;
					if( t == 0 ) zslot = sl;
				}
			}

			if( tlo >= 0 || thi >= 0 )
			{
				pi[ n ] = r;
				vi[ n ] = ( thi >= 0 ? thi + 2 * K : tlo );
				n++;
			}
		}
	}
};

template< int K, int NT, int B, int I0 >
__device__ __forceinline__ void df_feed_batch( f2 ( &acc )[ 3 ],
	const DfBatch< B >& d, const f2 ( &cp )[ NT / 2 ])
{
	dn_static_for< B >( [&]( auto jc )
	{
		constexpr int j = decltype( jc )::value;
		df_feed< K, NT, I0 + j >( acc, d.v[ j ], cp );
	} );
}

// (instrumented barrier / wait: the clocks only tick when dbg & 16)
#define DF_TIMED( acc, stmt ) do { if( P.dbg & 16 ) { \
	const unsigned long long t0_ = __builtin_readcyclecounter(); stmt; \
	acc += __builtin_readcyclecounter() - t0_; } else { stmt; } } while( 0 )

template< int KH, int NTH, int KV, int NTV, int DF_NHW >
__global__ void __launch_bounds__( DF_NTHR ) k_dnf( const DnFParams P )
{
	unsigned long long tk_total = ( P.dbg & 16 ? __builtin_readcyclecounter() :
		0 ), tk_dma = 0, tk_bar = 0;
	constexpr int DF_NVW = 12 - DF_NHW;         // column waves
	constexpr int DF_RS = 2 * DF_NHW;           // source rows per step
	constexpr int DF_LPV = DF_W * 4 / DF_NVW;   // elements per column wave
	// phases of the column waves' unroll: steps per whole number of periods
	constexpr int DF_NPH = ( DF_RS == 16 ? KV : ( KV == 3 ? 8 : 16 ));
	static_assert( DF_NVW * DF_LPV == DF_W * 4 && ( DF_NPH * DF_RS ) %
		( 16 * KV ) == 0, "column wave geometry" );
	constexpr int NPH = ( KH * ( DF_NT1 - 1 ) + NTH + 63 ) / 64; // DMA pieces
	constexpr int ROWPX = NPH * 64;
	constexpr int NIN = NTH + 2 * KH; // inputs of a lane's three outputs
	constexpr int BI = NIN / 4;       // ... in four batches
	static_assert( KH != 2 || NPH == 2, "plane layout assumes two pieces" );
	static_assert(( NTH & 1 ) == 0 && BI * 4 == NIN, "tap pairs, batches" );

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

		// LDS byte address of this lane's first input in buffer 0
		const unsigned lds_in = lds_s + ( rho * ROWPX +
			( KH == 2 ? 3 : 9 ) * tri ) * 16 + hf * 8;

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

		for( int s = 0; s < S; s++ )
		{
			if( s + 1 < S )
			{
				if( !( P.dbg & 4 ))
				issue( s + 1 );

				if( NPH == 3 )
				{
					DF_TIMED( tk_dma, GP_WAIT_VM( 6 ));
				}
				else
				{
					DF_TIMED( tk_dma, GP_WAIT_VM( 4 ));
				}
			}
			else
			{
				DF_TIMED( tk_dma, GP_WAIT_VM( 0 ));
			}

			const int bf = s & 1;

			// ---- resize: 3 consecutive outputs per lane
			if( !( P.dbg & 2 ))
			{
				const unsigned ad = lds_in + bf * ( 2 * ROWPX * 16 );
				DfBatch< BI > d0, d1, d2, d3;
				f2 acc[ 3 ];
				acc[ 0 ] = (f2) 0.0f; acc[ 1 ] = (f2) 0.0f; acc[ 2 ] = (f2) 0.0f;

				df_load_at< KH, NPH, BI, 0 >( d0, ad );
				df_load_at< KH, NPH, BI, BI >( d1, ad );
				df_wait< BI, BI >( d0 );
				df_feed_batch< KH, NTH, BI, 0 >( acc, d0, cp );
				df_load_at< KH, NPH, BI, 2 * BI >( d2, ad );
				df_wait< BI, BI >( d1 );
				df_feed_batch< KH, NTH, BI, BI >( acc, d1, cp );
				df_load_at< KH, NPH, BI, 3 * BI >( d3, ad );
				df_wait< BI, BI >( d2 );
				df_feed_batch< KH, NTH, BI, 2 * BI >( acc, d2, cp );
				df_wait< BI, 0 >( d3 );
				df_feed_batch< KH, NTH, BI, 3 * BI >( acc, d3, cp );

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
			if( !( P.dbg & 8 ))
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

			DF_TIMED( tk_bar, __syncthreads());
		}
	}
	else
	{
		// ------------------------------------------------------------
		// column waves
		// ------------------------------------------------------------
		// (Raising the column waves' priority, s_setprio 3, moves the barrier
		// wait from the row waves -- 28 % of their time at equal priority,
		// in-kernel clocks of AVIRHIP_DNF_DBG=16 -- to the column waves, 47 %,
		// and the launch takes as long: the step is bound by the SIMDs' total
		// issue, not by one role. Skewing the row waves of a SIMD against each
		// other with s_sleep costs 3-5 %.)
		// K = 2 along y: the column waves are the step's longest instruction
		// streams (12 taps and a correction-filter row every second source
		// row) and the row waves spend 40 % of their time in the barrier;
		// raised, a column wave issues whenever it can (3840x2160 ->
		// 1920x1080: 0.0636 -> 0.0615 ms)
		if( KV == 2 )
		{
			__builtin_amdgcn_s_setprio( 3 );
		}

		const int vi = wave - DF_NHW;
		const bool lane_ok = ( lane < DF_LPV );
		const int e = DF_LPV * vi + ( lane_ok ? lane : DF_LPV - 1 );
		const int x = strip * DF_W + ( e >> 2 );
		const int ch = e & 3;
		const bool st_ok = ( lane_ok && x < P.H.out_len );

		// the tap pairs (see DF_VSTMT): one copy per lane, in registers
		f2 k2[ NTV + 2 * KV ];
		{
			const f2* const v2 = (const f2*) ( P.V.coef + NTV + 4 );
#pragma unroll
			for( int i = 0; i < NTV + 2 * KV; i++ )
			{
				k2[ i ] = v2[ i ];
			}
#pragma unroll
			for( int i = 0; i < NTV + 2 * KV; i++ )
			{
				asm volatile( "" : "+v"( k2[ i ])); // (no re-loading later)
			}
		}

		gp_cfloat* const vc = (gp_cfloat*) P.V.coef;
		const float f0 = vc[ NTV ], f1 = vc[ NTV + 1 ], f2_ = vc[ NTV + 2 ],
			f3 = vc[ NTV + 3 ];

		// destination: lane offset fixed, row offset scalar; lanes that must
		// not store get an offset the buffer range check drops
		int st_voff;
		__amdgpu_buffer_rsrc_t drs;

		if( P.out.on != 0 )
		{
			const int es = ( P.out.type == AVIRHIP_U8 ? 1 :
				( P.out.type == AVIRHIP_U16 ? 2 : 4 ));

			drs = __builtin_amdgcn_make_buffer_rsrc( P.out.base, 0,
				0x7fffffff, 0x00020000 );
			st_voff = ( st_ok && ch < P.out.ch ?
				( x * P.out.ch + ch ) * es : (int) 0x80000000 );
		}
		else
		{
			drs = __builtin_amdgcn_make_buffer_rsrc( (void*) P.dst, 0,
				0x7fffffff, 0x00020000 );
			st_voff = ( st_ok ? ( x * 4 + ch ) * 4 : (int) 0x80000000 );
		}

		f2 accp[ 8 ];
		float wr[ 8 ];
#pragma unroll
		for( int i = 0; i < 8; i++ )
		{
			accp[ i ] = (f2) 0.0f;
			wr[ i ] = 0.0f;
		}

		int s = 0;
		int jl_base = 0; // local index of the output in slot 0 of this period

		while( s < S )
		{
			dn_static_for< DF_NPH >( [&]( auto phc )
			{
				constexpr int ph = decltype( phc )::value;

				if( s < S )
				{
					DF_TIMED( tk_bar, __syncthreads());
					f2 vp[ DF_RS / 2 ];
#pragma unroll
					for( int m = 0; m < DF_RS / 2; m++ )
					{
						vp[ m ].x = sF[ s & 1 ][ 2 * m ][ e ];
						vp[ m ].y = sF[ s & 1 ][ 2 * m + 1 ][ e ];
					}

					if( !( P.dbg & 1 ))
					dn_static_for< DF_RS >( [&]( auto ic )
					{
						constexpr int i = decltype( ic )::value;
						constexpr int u = ph * DF_RS + i; // row of the period
						constexpr DfRow< KV, NTV, u > R;

						// the sum of an output starts from +0.0f
						if constexpr( R.zslot >= 0 )
						{
							constexpr int zp = ( R.zslot >> 2 ) * 2 +
								( R.zslot & 1 );

							if constexpr((( R.zslot >> 1 ) & 1 ) == 0 )
							{
								accp[ zp ].x = 0.0f;
							}
							else
							{
								accp[ zp ].y = 0.0f;
							}
						}

#define DF_PI( k ) accp[ R.pi[ ( k ) < R.n ? ( k ) : 0 ]]
#define DF_KI( k ) k2[ R.vi[ ( k ) < R.n ? ( k ) : 0 ]]
						df_vstmt< R.n, ( i & 1 ) >( DF_PI( 0 ), DF_PI( 1 ),
							DF_PI( 2 ), DF_PI( 3 ), DF_PI( 4 ), DF_PI( 5 ),
							DF_PI( 6 ), DF_PI( 7 ), vp[ i >> 1 ], DF_KI( 0 ),
							DF_KI( 1 ), DF_KI( 2 ), DF_KI( 3 ), DF_KI( 4 ),
							DF_KI( 5 ), DF_KI( 6 ), DF_KI( 7 ));
#undef DF_PI
#undef DF_KI

						// a T row completes with its last tap
						if constexpr(( u - ( NTV - 1 )) % KV == 0 )
						{
							constexpr int jp = ( u - ( NTV - 1 )) / KV;
							constexpr int slot = (( jp % 16 ) + 16 ) % 16;
							constexpr int sp = ( slot >> 2 ) * 2 + ( slot & 1 );
							constexpr int ws = (( jp % 8 ) + 8 ) % 8;
							const int jl = jl_base + jp;

							if( jl >= 0 )
							{
								const int j = jstart + jl;
								float U = ((( slot >> 1 ) & 1 ) == 0 ?
									accp[ sp ].x : accp[ sp ].y );

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

									const int yr = y - P.dst_row0;

									if( P.out.on == 0 )
									{
										__builtin_amdgcn_raw_buffer_store_b32(
											__builtin_bit_cast( unsigned, a ),
											drs, st_voff,
											yr * (int) P.dst_ss * 4, 0 );
									}
									else
									{
										df_store_elem( P.out, drs, st_voff, yr,
											a );
									}
								}
							}
						}
					} );

					s++;
				}
			} );

			jl_base += DF_NPH * DF_RS / KV;
		}
	}

	if(( P.dbg & 16 ) && lane == 0 && P.clk != nullptr )
	{
		unsigned long long* const q = P.clk + ( (size_t) item * 12 + wave ) * 4;
		q[ 0 ] = __builtin_readcyclecounter() - tk_total;
		q[ 1 ] = tk_dma; q[ 2 ] = tk_bar; q[ 3 ] = (unsigned long long) S;
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

	static const int dbg = []() { const char* e = getenv( "AVIRHIP_DNF_DBG" );
		return( e != nullptr ? atoi( e ) : 0 ); }();
	P.dbg = dbg;
	P.clk = nullptr;
	static unsigned long long* dclk = nullptr;
	static int dcalls = 0;


	// one workgroup per compute unit: a chunk's warm-up (NT + 5K source rows)
	// is paid per work item, so as few, as tall chunks as fill the chip
	P.nstrips = ( D -> h.out_len + DF_W - 1 ) / DF_W;
	const int rows = row1 - row0;
	const int want = std::max( 1, D -> ncu / P.nstrips );
	P.crows = std::max( 8, ( rows + want - 1 ) / want );
	P.nchunks = ( rows + P.crows - 1 ) / P.crows;
	const dim3 grid( P.nstrips * P.nchunks );

	// (debugging aid, single-threaded use: per-wave clocks of up to 4096 work
	// items; larger launches and a failed allocation run without them)
	if(( dbg & 16 ) && grid.x <= 4096 )
	{
		if( dclk == nullptr && hipMalloc( &dclk, 4096 * 12 * 32 ) != hipSuccess )
		{
			(void) hipGetLastError();
			dclk = nullptr;
		}

		P.clk = dclk;
	}
	const int kk = D -> h.K * 10 + D -> v.K;
	// K = 3 along y: 9 row waves + 3 column waves (AVIRHIP_DNF_NHW=8: the 8 + 4
	// split, for A/B timing)
	static const int nhw3 = []() { const char* e = getenv( "AVIRHIP_DNF_NHW" );
		return( e != nullptr && atoi( e ) == 8 ? 8 : 9 ); }();

#define DF_LAUNCH( ... ) hipLaunchKernelGGL(( k_dnf< __VA_ARGS__ > ), grid, \
	dim3( DF_NTHR ), 0, st, P )

	switch( kk )
	{
		case 33:
			if( nhw3 == 9 ) DF_LAUNCH( 3, 38, 3, 38, 9 );
			else DF_LAUNCH( 3, 38, 3, 38, 8 );
			break;
		case 22:
			DF_LAUNCH( 2, 24, 2, 24, 8 );
			break;
		case 32:
			DF_LAUNCH( 3, 38, 2, 24, 8 );
			break;
		case 23:
			if( nhw3 == 9 ) DF_LAUNCH( 2, 24, 3, 38, 9 );
			else DF_LAUNCH( 2, 24, 3, 38, 8 );
			break;
		default:
			return( 1 );
	}

#undef DF_LAUNCH

	if(( dbg & 16 ) && P.clk != nullptr && ++dcalls == 100 )
	{
		// in-kernel clocks of a launch in the middle of a timing loop
		std::vector< unsigned long long > hc( (size_t) grid.x * 12 * 4 );
		(void) hipStreamSynchronize( st );
		(void) hipMemcpy( hc.data(), dclk, hc.size() * 8, hipMemcpyDeviceToHost );
		const int nh = ( D -> v.K == 3 ? nhw3 : 8 );
		double tot[ 2 ] = { 0, 0 }, dma[ 2 ] = { 0, 0 }, bar[ 2 ] = { 0, 0 };
		double mx = 0.0;

		for( unsigned i = 0; i < grid.x; i++ )
		{
			for( int wv = 0; wv < 12; wv++ )
			{
				const unsigned long long* q = &hc[ ( (size_t) i * 12 + wv ) * 4 ];
				const int r = ( wv < nh ? 0 : 1 );
				tot[ r ] += (double) q[ 0 ]; dma[ r ] += (double) q[ 1 ];
				bar[ r ] += (double) q[ 2 ];
				mx = std::max( mx, (double) q[ 0 ]);
			}
		}

		const double nr = (double) grid.x * nh, nc = (double) grid.x * ( 12 - nh );
		fprintf( stderr, "k_dnf clocks (%u items, %d row waves): row waves total "
			"%.0f, DMA wait %.0f, barrier %.0f | column waves total %.0f, "
			"barrier %.0f | longest wave %.0f cycles, %llu steps\n", grid.x, nh,
			tot[ 0 ] / nr, dma[ 0 ] / nr, bar[ 0 ] / nr, tot[ 1 ] / nc,
			bar[ 1 ] / nc, mx, hc[ 3 ]);
	}
	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

} // namespace avirhip
