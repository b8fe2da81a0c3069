// gfuse.hip -- k_gf: BOTH passes of a general-ratio UPSIZING plan in one launch
// (doFilter -> doUpsample -> doResize2 with non-zero positions on both axes,
// avir.h:3748-3866, 3249-3400, 4114-4328; resizeScanlineH then resizeScanlineV,
// avir.h:6522-6619), no FltBuf.
//
// The two pass kernels of gpass.hip share their lane mapping: k_gh's lane is
// an output pixel of a strip of <= 64 outputs, k_gv's lane a pixel column of a
// 64-pixel strip. So the horizontal result of a source row IS, lane for lane,
// the next input of the vertical stage -- it never has to leave the registers:
//
//   one wavefront per (strip of `ow` output columns) x (chunk of output rows),
//   marching down the source rows its vertical windows read. Per source row:
//     H  the row's source segment (LDS-DMA, three rows ahead, k_gh's segment
//        table) -> FIR7 along x -> LDS -> the lane's 12-tap gather with its
//        coefficient row in registers  => h = FltBuf[row][column] of the
//        reference, bit for bit;
//     V  h enters the 7-row FIR window (registers), the FIR's output the
//        12-row gather window (registers, compile-time slots: the step loop is
//        unrolled 16 times); the output rows whose last input this row was
//        are produced with wave-uniform coefficient rows (LDS, one output
//        ahead) and stored -- k_gv's vertical stage.
//
// What it buys: FltBuf's round trip (1920x1080 -> 5760x3240 writes and re-reads
// 99.5 MB of it against 331 MB of images), one launch, and every per-row
// overhead of the second kernel's input side (DMA issue, landing queue, waits).
// What it costs: a chunk's vertical run-in (18 source rows before its first
// output) now repeats the horizontal arithmetic too -- chunks are as tall as
// the chip's occupancy allows (balanced_chunk with that run-in).
//
// Arithmetic contract as in gpass.hip: mul and add apart, sums from +0.0f in
// ascending tap order, FIR as f0*c + sum f_i*(right_i + left_i) => 0 ULP.

#include "gpass_dev.h"
#include <algorithm>
#include <type_traits>

namespace avirhip {

#define GF_NB 8   // source row buffers: rows u .. u + 7 are in LDS or in flight
#define GF_TE 32  // table staging: 8 KiB of vertical coefficient rows per chunk

__global__ void __launch_bounds__( 64 ) k_gf( const GFParams P )
{
	constexpr int NT = 12;  // taps of both gathers
	constexpr int RW = 16;  // vertical gather window (ring slots, >= NT)
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	const GPAxis& HA = P.hx;
	const GPAxis& VA = P.vx;
	f4* const sbuf = (f4*) smem;                      // [GF_NB][sbuf px]
	f4* const mbuf = sbuf + GF_NB * (size_t) P.sbuf;  // [mbuf px]
	float* const ctab = (float*) ( mbuf + P.mbuf );   // [chunk rows][NT]
	const unsigned sbuf_lds = (unsigned) (unsigned long) (lds_char*) smem;

	const int lane = threadIdx.x;
	const int strip = blockIdx.x % P.nstrips;
	const int chunk = blockIdx.x / P.nstrips;
	const int y0 = P.row_lo + chunk * P.chunk;
	const int y1 = min( y0 + P.chunk, P.row_hi );

	if( y0 >= y1 )
	{
		return;
	}

	// ---- horizontal side (k_gh, PRE): this lane's output column
	const int slo = P.seg[ strip * 4 ];
	const int mlo = P.seg[ strip * 4 + 2 ], mcnt = P.seg[ strip * 4 + 3 ];
	const int j = strip * P.ow + lane;
	const bool jok = ( lane < P.ow && j < HA.out_len );
	const int gh = gp_clamp( j, HA.g_len - 1 );
	const int gs = HA.start[ gh ] - mlo;
	const int hslast = HA.in_len - 1;
	float cf[ NT ];
	{
		const float* const cr = HA.coef + (long) gh * NT;
#pragma unroll
		for( int t = 0; t < NT; t++ )
		{
			cf[ t ] = cr[ t ];
		}
	}

	// the FIR output this lane computes per row: C[ mlo + lane ] (mcnt <= 64)
	const int hm = mlo + lane;
	const bool hfir = ( lane < mcnt );
	const bool hzero = ( hm > HA.view_mmax );
	const int hq = gp_clamp( hm, HA.c_len - 1 ) - HA.e - slo;
	const float hf0 = HA.f[ 0 ], hf1 = HA.f[ 1 ], hf2 = HA.f[ 2 ], hf3 = HA.f[ 3 ];

	// ---- vertical side (k_gv, PRE, register window)
	gp_cint* const a_start = (gp_cint*) (unsigned long) VA.start;
	const int vslast = VA.in_len - 1;
	const f4 f0 = (f4) VA.f[ 0 ], f1 = (f4) VA.f[ 1 ], f2_ = (f4) VA.f[ 2 ],
		f3 = (f4) VA.f[ 3 ];
	const int ga = y0, gb = y1 - 1;
	const int u_a = a_start[ ga ] - 6; // (six rows of FIR run-in first)
	const int u_b = a_start[ gb ] + NT - 1;

	int* const stab = (int*) ( ctab + ( gb - ga + 1 ) * NT );
	{
		const int total = ( gb - ga + 1 ) * NT;
		float tmp[ GF_TE ];
#pragma unroll
		for( int k = 0; k < GF_TE; k++ )
		{
			const int i = lane + 64 * k;
			tmp[ k ] = ( i < total ? VA.coef[ (long) ga * NT + i ] : 0.0f );
		}
#pragma unroll
		for( int k = 0; k < GF_TE; k++ )
		{
			const int i = lane + 64 * k;

			if( i < total )
			{
				ctab[ i ] = tmp[ k ];
			}
		}

		for( int i = lane; i <= gb - ga; i += 64 )
		{
			stab[ i ] = VA.start[ ga + i ] + NT - 1;
		}
	}

	int g = ga;                              // next output row
	int g_need = a_start[ g ] + NT - 1;      // its last input row
	float* dp = P.dst + (long) ( y0 - P.dst_row0 ) * P.dst_ss + (long) j * 4;
	long orow = y0 - P.dst_row0;

	auto store_row = [&]( const f4 o )
	{
		if( P.out.on != 0 )
		{
			gp_store_int_row( P.out, orow, j, o, jok );

			orow++;
			return;
		}

		if( jok )
		{
			*(f4*) dp = o;
		}

		dp += P.dst_ss;
	};

	f4 cpre[ 3 ];
	int need1_v;

	auto prefetch_tabs = [&]()
	{
		const int gi = min( g + 1, gb ) - ga;
		need1_v = ( g + 1 <= gb ? stab[ gi ] : 0x7fffffff );
		const float* const cr = ctab + ( min( g, gb ) - ga ) * NT;
#pragma unroll
		for( int b = 0; b < 3; b++ )
		{
			cpre[ b ] = *(const f4*) ( cr + 4 * b );
		}
	};

	f4 win[ RW ];
#pragma unroll
	for( int i = 0; i < RW; i++ )
	{
		win[ i ] = (f4) 0.0f;
	}

	f4 tw[ 8 ];
#pragma unroll
	for( int i = 0; i < 8; i++ )
	{
		tw[ i ] = (f4) 0.0f;
	}

	auto drain = [&]( const int u, auto IC )
	{
		constexpr int I = decltype( IC )::value;

		// (after the chunk's last output g_need is prefetch_tabs' 0x7fffffff)
		while( g_need <= u )
		{
			const f4 o = gp_dot< false, f4, NT >( NT,
				[&]( const int t ) { return( win[ ( I - ( NT - 1 ) + t + 4 * RW ) &
					( RW - 1 )]); },
				[&]( const int b ) { return( cpre[ b ]); });

			store_row( o );
			g++;
			g_need = __builtin_amdgcn_readfirstlane( need1_v );
			prefetch_tabs();
		}
	};

	prefetch_tabs();

	// ---- the feed: source row clamp( u ) of step u, its segment by LDS-DMA
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
		(void*) P.src, 0, 0x7fffffff, 0x00020000 );
	const int srow_b = (int) P.src_ss * 4;

	// (the clamped columns are recomputed per row: four registers for them
	// take the kernel from 254 to 256 VGPRs -- one wave per SIMD instead of two,
	// 1920x1080 -> 5760x3240 0.105 -> 0.159 ms)
	auto issue_row = [&]( const int u, const int buf )
	{
		const int r = gp_clamp( u, vslast );

		for( int k = 0; k < P.nseg; k++ )
		{
			const int x = gp_clamp( slo + k * 64 + lane, hslast );
			gp_dma( rsrc, sbuf_lds + ( buf * P.sbuf + k * 64 ) * 16,
				r * srow_b + x * 16 );
		}
	};

	for( int k = 0; k < GF_NB - 1 && u_a + k <= u_b; k++ )
	{
		issue_row( u_a + k, k );
	}

	auto step = [&]( const int u, auto IC )
	{
		constexpr int i = decltype( IC )::value;

		if( u > u_b )
		{
			return;
		}

		const int buf = i & ( GF_NB - 1 );
		const f4* const s = sbuf + buf * P.sbuf;
		GP_PRIO_LOW();

		if( u + GF_NB - 1 <= u_b )
		{
			issue_row( u + GF_NB - 1, ( buf + GF_NB - 1 ) & ( GF_NB - 1 ));
		}

		// Row u's pieces have landed when at most the pieces of the GF_NB - 1
		// rows after it are outstanding. The stores in between count on vmcnt
		// too: the fixed count also waits for some of them and for rows nearer
		// than u + 7, which the deep queue can afford -- an exact count (pieces +
		// rows stored since) is a branch ladder per step that cost more than it
		// saved (lfuse.hip: 16.5 -> 12.1 us of skeleton).
		if( P.nseg == 1 && u_b - u >= GF_NB - 1 )
		{
			GP_WAIT_VM( 7 );
		}
		else
		{
			// (steady state: the count per segment count spelled out -- the
			// general count walks gp_wait_vm's 25-way ladder, ten branches a row)
			if( u_b - u >= GF_NB - 1 && P.nseg <= 4 )
			{
				if( P.nseg <= 2 )
				{
					if( P.nseg == 1 ) gp_wait_vm(( GF_NB - 1 ) * 1 );
					else gp_wait_vm(( GF_NB - 1 ) * 2 );
				}
				else
				{
					if( P.nseg == 3 ) gp_wait_vm(( GF_NB - 1 ) * 3 );
					else gp_wait_vm(( GF_NB - 1 ) * 4 );
				}
			}
			else
			{
				gp_wait_vm( min( GF_NB - 1, u_b - u ) * P.nseg );
			}
		}

		// H: FIR7 along x of the segment (clamped source view; C above the
		// zero-stuffed view's end is zero) ...
		if( hfir )
		{
			const f4* const q = s + hq;
			f4 c = hf0 * q[ 0 ];
			c = c + hf1 * ( q[ 1 ] + q[ -1 ]);
			c = c + hf2 * ( q[ 2 ] + q[ -2 ]);
			c = c + hf3 * ( q[ 3 ] + q[ -3 ]);
			mbuf[ lane ] = ( hzero ? (f4) 0.0f : c );
		}

		// ... then this lane's gather: FltBuf[ clamp( u ) ][ j ]
		f4 h = (f4) 0.0f;

		if( jok )
		{
			const f4* const in = mbuf + gs;
			f4 w[ NT ];
#pragma unroll
			for( int t = 0; t < NT; t++ )
			{
				w[ t ] = in[ t ];
			}

			__builtin_amdgcn_sched_barrier( 0 );
			GP_PRIO_HIGH();

			h = gp_dot< false, f4, NT >( NT,
				[&]( const int t ) { return( w[ t % NT ]); },
				[&]( const int b ) -> f4
				{
					f4 c;
					c.x = cf[ ( 4 * b ) % NT ]; c.y = cf[ ( 4 * b + 1 ) % NT ];
					c.z = cf[ ( 4 * b + 2 ) % NT ]; c.w = cf[ ( 4 * b + 3 ) % NT ];
					return( c );
				});
		}

		// V: the FIR window, then the gather window
		tw[ i & 7 ] = h;
		f4 c = f0 * tw[ ( i - 3 ) & 7 ];
		c = c + f1 * ( tw[ ( i - 2 ) & 7 ] + tw[ ( i - 4 ) & 7 ]);
		c = c + f2_ * ( tw[ ( i - 1 ) & 7 ] + tw[ ( i - 5 ) & 7 ]);
		c = c + f3 * ( tw[ i & 7 ] + tw[ ( i - 6 ) & 7 ]);

		if( u > VA.view_mmax )
		{
			c = (f4) 0.0f;
		}

		win[ i & ( RW - 1 )] = c;

		if( u - u_a >= 6 )
		{
			drain( u, IC );
		}

	};

	for( int ub = u_a; ub <= u_b; ub += 16 )
	{
		step( ub + 0, std::integral_constant< int, 0 >() );
		step( ub + 1, std::integral_constant< int, 1 >() );
		step( ub + 2, std::integral_constant< int, 2 >() );
		step( ub + 3, std::integral_constant< int, 3 >() );
		step( ub + 4, std::integral_constant< int, 4 >() );
		step( ub + 5, std::integral_constant< int, 5 >() );
		step( ub + 6, std::integral_constant< int, 6 >() );
		step( ub + 7, std::integral_constant< int, 7 >() );
		step( ub + 8, std::integral_constant< int, 8 >() );
		step( ub + 9, std::integral_constant< int, 9 >() );
		step( ub + 10, std::integral_constant< int, 10 >() );
		step( ub + 11, std::integral_constant< int, 11 >() );
		step( ub + 12, std::integral_constant< int, 12 >() );
		step( ub + 13, std::integral_constant< int, 13 >() );
		step( ub + 14, std::integral_constant< int, 14 >() );
		step( ub + 15, std::integral_constant< int, 15 >() );
	}

	GP_WAIT_VM( 0 ); // no LDS-DMA may outlive the workgroup's LDS
}

// Output rows [row_lo, row_hi) of the plan; P comes filled in by gpass.hip but
// for the launch geometry. Returns 1 when the chunk tables would not fit.
int gfuse_launch( GFParams& P, double k_v, hipStream_t st )
{
	const int rows = P.row_hi - P.row_lo;
	const size_t lds_fix = (size_t) ( GF_NB * P.sbuf + P.mbuf ) * 16;
	// a chunk's coefficient rows and positions: 8 KiB of table at most
	const int max_chunk = ( GF_TE * 64 ) / 12;
	const size_t lds_max = lds_fix + (size_t) max_chunk * ( 12 + 1 ) * 4 + 64;
	const int wpc = std::max( 1, std::min( 16, (int) ( 160 * 1024 /
		( lds_max + 512 ))));

	// the run-in of a chunk, in output rows: 18 source rows of horizontal work
	// and window filling (~140 instructions each) before the first output,
	// against ~140 per source row and ~60 per output row afterwards
	const int warm = (int) ( 18.0 * 140.0 / ( 140.0 * k_v + 60.0 ) + 0.5 );
	// (read per call: tools/stress_geom.py varies it case by case)
	const int force = ( getenv( "AVIRHIP_GF_CHUNK" ) != nullptr ?
		atoi( getenv( "AVIRHIP_GF_CHUNK" )) : 0 );

	P.chunk = ( force > 0 ? std::min( force, max_chunk ) :
		balanced_chunk( rows, P.nstrips, 4, max_chunk, warm, wpc, true ));

	const int nch = ( rows + P.chunk - 1 ) / P.chunk;
	const size_t lds = lds_fix + (size_t) P.chunk * ( 12 + 1 ) * 4 + 64;

	AVIRHIP_HIPCHECK( AVIRHIP_DYN_LDS( k_gf, lds ));

	hipLaunchKernelGGL( k_gf, dim3( P.nstrips * nch ), dim3( 64 ), lds, st, P );
	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

} // namespace avirhip
