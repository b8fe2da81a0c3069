// gpass_h2.hip -- k_gh2: the horizontal pass of AVIR axes that downsize by
// 1 < k < 2 (a 13 .. 25-tap gather over the zero-stuffed view of the source,
// then the 7-tap correction filter: doUpsample without a filter avir.h:3260-3402
// -> doResize2 4114-4328 -> doFilter 3748-3866), float RGBA sources, TWO adjacent
// gather outputs per lane (round 6).
//
// Why: k_gh<POST> evaluates one output per lane -- NT ds_read_b128 per output,
// lanes k = 1.5 .. 1.9 pixels apart (24 .. 30 bytes: LDS bank conflicts). Its
// counters on 3840x2160 -> 2560x1440 (profiles/r06_gh_post/): the LDS busy two
// thirds of the launch, 40 % of those cycles bank conflicts, the vector ALU a
// third. The windows of two adjacent outputs overlap in all but one or two
// pixels, so a lane that evaluates BOTH reads NT + 2 pixels instead of 2 NT, and
// its neighbours sit 2k pixels away (48 bytes at k = 1.5: conflict-free
// ds_read_b128).
//
// Same bits as k_gh / the reference: output g0 sums its NT products from +0.0f
// in ascending tap order; output g0 + 1 starts d = start[g0 + 1] - start[g0]
// (1 or 2) pixels later, its coefficient row is loaded SHIFTED by d into the
// registers of window positions 1 .. NT + 1, and the one position of that range
// its window does not cover (position 1 when d = 2, position NT + 1 when d = 1)
// contributes (+0.0f) * (+0.0f): the coefficient is zero AND the pixel is
// replaced by zero (a select, so that an Inf / NaN there cannot leak in) -- a
// leading +0 leaves the sum's start at +0, a trailing +0 leaves any sum as it is
// (a sum that started from +0 is never -0).
//
// Structure as k_gh: one wavefront per (strip of `ow` <= 122 outputs) x (chunk
// of rows), the rows' source segments by LDS-DMA three rows ahead (GH_NB row
// buffers, loads-only wait counts: waits.h), gather -> LDS -> FIR7 along x ->
// two adjacent pixels (32 bytes) stored per lane.

#include "gpass_dev.h"

namespace avirhip {

#define GH2_NB 4 // source row buffers (as GH_NB)
#ifndef GH2_SPLIT
#define GH2_SPLIT 21 // windows of more pixels are read in two halves
#endif

// pixel (four floats) times ONE coefficient: half HI (a constant once the tap
// loop is unrolled) of the register pair `cp`,
// broadcast to both lanes of the packed multiply (VOP3P op_sel / op_sel_hi on
// the second source; the first, the pixel half, is taken as it is)
__device__ __forceinline__ f4 gh2_mul( const f4 v, const f2 cp, const bool HI )
{
	f2 lo, hi;
	const f2 vl = v.lo, vh = v.hi;

	if( HI )
	{
		asm( "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]"
			: "=v"( lo ) : "v"( vl ), "v"( cp ));
		asm( "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]"
			: "=v"( hi ) : "v"( vh ), "v"( cp ));
	}
	else
	{
		asm( "v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]"
			: "=v"( lo ) : "v"( vl ), "v"( cp ));
		asm( "v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]"
			: "=v"( hi ) : "v"( vh ), "v"( cp ));
	}

	return( __builtin_shufflevector( lo, hi, 0, 1, 2, 3 ));
}

template< int NTC >
__global__ void __launch_bounds__( 64 ) k_gh2( const GHParams P )
{
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	const GPAxis& A = P.ax;
	f4* const sbuf = (f4*) smem;                      // [GH2_NB][sbuf px]
	// gather outputs in two planes -- even / odd index relative to mlo: a lane's
	// two sums go to the same slot of either plane (16 bytes between lanes:
	// conflict-free), and so do the correction filter's reads
	f4* const mE = sbuf + GH2_NB * (size_t) P.sbuf;   // [64]
	f4* const mO = mE + 64;                           // [64]
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
	const int j0 = strip * P.ow + 2 * lane;          // this lane's outputs: j0, j0 + 1
	const bool jok0 = ( 2 * lane < P.ow && j0 < A.out_len );
	const bool jok1 = ( 2 * lane + 1 < P.ow && j0 + 1 < A.out_len );
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
		(void*) P.src, 0, 0x7fffffff, 0x00020000 );
	const int srow_b = (int) P.src_ss * 4;
	const int slast = A.in_len - 1;
	const int glast = A.g_len - 1;

	// per-lane source pixels of the DMA pieces (clamped: the extended view)
	int xb[ 4 ];
#pragma unroll
	for( int k = 0; k < 4; k++ )
	{
		xb[ k ] = gp_clamp( slo + k * 64 + lane, slast ) * 16;
	}

	auto issue_row = [&]( const int r, const int buf )
	{
		const int rb = r * srow_b;
		const unsigned lb = sbuf_lds + ( buf * P.sbuf ) * 16;
#pragma unroll
		for( int k = 0; k < 4; k++ )
		{
			if( k < P.nseg )
			{
				gp_dma( rsrc, lb + k * 1024, rb + xb[ k ]);
			}
		}
	};

	// the two gathers this lane evaluates: G[ mlo + 2 lane ], G[ mlo + 2 lane + 1 ]
	const int g0 = gp_clamp( mlo + 2 * lane, glast );
	const int g1 = gp_clamp( mlo + 2 * lane + 1, glast );
	const bool gok0 = ( 2 * lane < mcnt );
	const bool gok1 = ( 2 * lane + 1 < mcnt );
	const int st0 = A.start[ g0 ];
	// (1 or 2 for outputs that exist -- the host checked; 0 for the clamped
	// copy past the last output, whose sum nobody reads)
	const int d = A.start[ g1 ] - st0;
	const int gs = st0 - slo;

	// The coefficient rows, two taps to a register PAIR: a tap multiplies a
	// pixel as two v_pk_mul_f32 whose second operand is one half of a pair,
	// broadcast by op_sel (gh2_mul) -- one register per coefficient. (As plain
	// `float * f4` the compiler keeps every coefficient as a {c, c} pair: 12
	// registers per tap, 256 at 18 taps, one wave per SIMD.)
	constexpr int NP0 = ( NTC + 1 ) / 2;     // pairs of g0's row: taps 0 .. NT - 1
	constexpr int NP1 = ( NTC + 2 ) / 2;     // ... of g1's, window positions 1 .. NT + 1
	f2 c0[ NP0 ];
	f2 c1[ NP1 ];
	{
		const float* const cr0 = A.coef + (long) g0 * NTC;
		const float* const cr1 = A.coef + (long) g1 * NTC;
#pragma unroll
		for( int t = 0; t < 2 * NP0; t++ )
		{
			const float v = ( t < NTC ? cr0[ t < NTC ? t : 0 ] : 0.0f );
			if( t & 1 ) c0[ t >> 1 ].y = v; else c0[ t >> 1 ].x = v;
		}
#pragma unroll
		for( int i = 1; i <= 2 * NP1; i++ )
		{
			const int t = i - d;
			const float v = ( i <= NTC + 1 && t >= 0 && t < NTC ?
				cr1[ gp_clamp( t, NTC - 1 )] : 0.0f );
			if(( i - 1 ) & 1 ) c1[ ( i - 1 ) >> 1 ].y = v; else c1[ ( i - 1 ) >> 1 ].x = v;
		}
	}

	const bool d1 = ( d == 1 ); // position 1 belongs to g1's window
	const bool d2 = ( d == 2 ); // position NT + 1 does

	// The correction filter of this lane's two outputs j0, j0 + 1 reads the eight
	// gather outputs G[ clamp( j0 - e - 3 .. j0 - e + 4 ) ]: their LDS addresses
	// once per work item (as 14 clamped index computations per row they were a
	// fifth of the row's vector instructions)
	const f4* fa[ 8 ];
#pragma unroll
	for( int i = 0; i < 8; i++ )
	{
		const int m = min( max( gp_clamp( j0 - A.e - 3 + i, glast ) - mlo, 0 ), 127 );
		fa[ i ] = (( m & 1 ) ? mO : mE ) + ( m >> 1 );
	}

	for( int k = 0; k < GH2_NB - 1 && r0 + k < r1; k++ )
	{
		issue_row( r0 + k, k );
	}

	for( int r = r0; r < r1; r++ )
	{
		const int buf = ( r - r0 ) & ( GH2_NB - 1 );
		f4* const s = sbuf + buf * P.sbuf;

		// row r + 3 goes out (into the buffer row r - 1 has left: its window
		// was consumed by that row's arithmetic), then wait for row r: only the
		// pieces of the rows after it may be in flight (LOADS only: waits.h)
		if( r + GH2_NB - 1 < r1 )
		{
			issue_row( r + GH2_NB - 1, ( buf + GH2_NB - 1 ) & ( GH2_NB - 1 ));
		}

		if( r1 - 1 - r >= GH2_NB - 1 )
		{
			if( P.nseg <= 2 )
			{
				if( P.nseg == 1 ) gp_wait_vm(( GH2_NB - 1 ) * 1 );
				else gp_wait_vm(( GH2_NB - 1 ) * 2 );
			}
			else
			{
				if( P.nseg == 3 ) gp_wait_vm(( GH2_NB - 1 ) * 3 );
				else gp_wait_vm(( GH2_NB - 1 ) * 4 );
			}
		}
		else
		{
			gp_wait_vm( min( GH2_NB - 1, r1 - 1 - r ) * P.nseg );
		}

		if( slo + scnt - 1 > A.view_mmax )
		{
			// zero-stuffed view of the source: its tail reads 0
			for( int x = lane; x < scnt; x += 64 )
			{
				if( slo + x > A.view_mmax )
				{
					s[ x ] = (f4) 0.0f;
				}
			}
		}

		// ---- the two gathers: the window's NT + 2 pixels -- all reads in flight
		// at once up to 21 of them, in two halves beyond (the window, the two
		// coefficient rows and the FIR stage's reads have to fit 256 registers
		// for two waves per SIMD) --, then the arithmetic
		{
			const f4* const in = s + gs;
			constexpr int NW = NTC + 2;
			constexpr int HA = ( NW > GH2_SPLIT ? ( NW + 1 ) / 2 : NW );
			constexpr int HB = NW - HA;
			const f4 z = (f4) 0.0f;
			f4 a0 = (f4) 0.0f;
			f4 a1 = (f4) 0.0f;

			// window position i (pixel `wi`): tap i of g0 (i < NT), tap i - d of
			// g1 (i >= 1). A pixel outside g1's window must not reach its sum,
			// whatever it holds: selected to +0, times the +0 coefficient.
			// Products and sums alternate, pinned: left to itself the compiler
			// issues every multiply first and keeps 2 NT products alive.
			auto tap = [&]( const int i, const f4 wi )
			{
				f4 p0, p1;

				if( i < NTC )
				{
					p0 = gh2_mul( wi, c0[ ( i < NTC ? i : 0 ) >> 1 ], ( i & 1 ) != 0 );
				}

				if( i >= 1 )
				{
					const f4 x = ( i == 1 ? ( d1 ? wi : z ) :
						( i == NTC + 1 ? ( d2 ? wi : z ) : wi ));
					p1 = gh2_mul( x, c1[ ( i >= 1 ? i - 1 : 0 ) >> 1 ],
						(( i - 1 ) & 1 ) != 0 );
				}

				__builtin_amdgcn_sched_barrier( 0 );

				if( i < NTC ) a0 = a0 + p0;
				if( i >= 1 ) a1 = a1 + p1;

				// (pinned: instruction selection is free to postpone an add
				// until its sum is next used and keep the product alive instead
				// -- k_up2's lesson, NOTEBOOK 4.1)
				asm volatile( "" : "+v"( a0 ), "+v"( a1 ));
			};

			{
				f4 w[ HA ];
#pragma unroll
				for( int i = 0; i < HA; i++ )
				{
					w[ i ] = in[ i ];
				}

				__builtin_amdgcn_sched_barrier( 0 );
#pragma unroll
				for( int i = 0; i < HA; i++ )
				{
					tap( i, w[ i ]);
				}
			}

			if constexpr( HB > 0 )
			{
				f4 w[ HB > 0 ? HB : 1 ];
#pragma unroll
				for( int i = 0; i < HB; i++ )
				{
					w[ i ] = in[ HA + i ];
				}

				__builtin_amdgcn_sched_barrier( 0 );
#pragma unroll
				for( int i = 0; i < HB; i++ )
				{
					tap( HA + i, w[ i ]);
				}
			}

			if( gok0 ) mE[ lane ] = a0;
			if( gok1 ) mO[ lane ] = a1;
		}

		// ---- out[j] = f0 G[c] + f1 ( G[c+1] + G[c-1] ) + ... , c = clamp( j - e +- i )
		f4 o0, o1;
		{
			f4 v[ 8 ];
#pragma unroll
			for( int i = 0; i < 8; i++ )
			{
				v[ i ] = *fa[ i ];
			}

			o0 = A.f[ 0 ] * v[ 3 ];
			o0 = o0 + A.f[ 1 ] * ( v[ 4 ] + v[ 2 ]);
			o0 = o0 + A.f[ 2 ] * ( v[ 5 ] + v[ 1 ]);
			o0 = o0 + A.f[ 3 ] * ( v[ 6 ] + v[ 0 ]);
			o1 = A.f[ 0 ] * v[ 4 ];
			o1 = o1 + A.f[ 1 ] * ( v[ 5 ] + v[ 3 ]);
			o1 = o1 + A.f[ 2 ] * ( v[ 6 ] + v[ 2 ]);
			o1 = o1 + A.f[ 3 ] * ( v[ 7 ] + v[ 1 ]);
		}

		float* const dp = P.dst + (long) ( r - P.dst_row0 ) * P.dst_ss +
			(long) j0 * 4;

		if( jok0 )
		{
#ifdef GP_DBG
			if( !( P.dbg & 1 ) || o0.x == 1234.5f )
#endif
			*(f4*) dp = o0;
		}

		if( jok1 )
		{
#ifdef GP_DBG
			if( !( P.dbg & 1 ) || o1.x == 1234.5f )
#endif
			*(f4*) ( dp + 4 ) = o1;
		}
	}
}

// 0, or 1 when there is no variant for the tap count (nothing launched)
int launch_gh2( const GHParams& P, int items, size_t lds, hipStream_t st )
{
#define GH2_L( N ) case N: hipLaunchKernelGGL(( k_gh2< N > ), dim3( items ), \
	dim3( 64 ), lds, st, P ); return( 0 )
	switch( P.ax.nt )
	{
		GH2_L( 13 ); GH2_L( 14 ); GH2_L( 15 ); GH2_L( 16 ); GH2_L( 17 );
		GH2_L( 18 ); GH2_L( 19 ); GH2_L( 20 ); GH2_L( 21 ); GH2_L( 22 );
		GH2_L( 23 ); GH2_L( 24 ); GH2_L( 25 );
		default: break;
	}
#undef GH2_L
	return( 1 );
}

} // namespace avirhip
