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
//   k_gv  vertical pass. Lane = half pixel (8 B) of a 32-pixel strip: a row
//         is one contiguous 512-byte segment. Source rows arrive by LDS-DMA
//         (two rows per instruction) in an LDS ring, one block of output rows
//         ahead of the arithmetic; FIR7 rows and gather windows are read from
//         the rings with wave-uniform row addresses; coefficients are
//         wave-uniform scalar loads.
//   k_gh  horizontal pass. Lane = whole output pixel (16 B) of a strip of up
//         to 64 outputs; its coefficient row stays in registers while the wave
//         walks down its rows; the row's source segment arrives by LDS-DMA one
//         row ahead.
//
// Arithmetic contract (bit-exact): mul and add separate (-ffp-contract=off),
// AVIR sums start from +0.0f and take their taps in ascending order, the FIR
// is f0*c + sum f_i*(right_i + left_i); LANCIR accumulates even and odd taps
// separately from bare first products and adds the two sums at the end.

#include "plan.h"
#include <algorithm>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

namespace avirhip {

typedef float f4 __attribute__(( ext_vector_type( 4 )));
typedef float f2 __attribute__(( ext_vector_type( 2 )));
typedef __attribute__(( address_space( 3 ))) char lds_char;
// Wave-uniform table reads (positions, coefficient rows) go through the
// constant address space: they become scalar loads (lgkmcnt), so they never
// wait on the vector-memory counter the LDS-DMA prefetch keeps busy.
typedef const __attribute__(( address_space( 4 ))) float gp_cfloat;
typedef const __attribute__(( address_space( 4 ))) int gp_cint;

enum { GP_NONE = 0, GP_PRE = 1, GP_POST = 2 };

struct GPAxis
{
	int mode;          // GP_*
	int lanc;          // LANCIR summation order
	int nt;            // taps of every gather output
	int in_len;        // source length along the axis (clamp range)
	int c_len;         // PRE: length of the FIR's output array
	int view_mmax;     // gather input index above which the view reads 0
	int g_len;         // outputs of the gather
	int out_len;       // outputs of the axis
	int e;             // FIR edge pixel count (3)
	float f[ 4 ];      // FIR taps f0..f3
	const int* start;  // [g_len] first gather input index (unclamped)
	const float* coef; // AVIR: [g_len][nt]; LANCIR: [phases][nt]
	const int* fidx;   // LANCIR: [g_len] phase of each output
};

struct GVParams
{
	const float* src; long src_ss; int width; // px per row, floats per row
	float* dst; long dst_ss; int dst_row0;
	GPAxis ax;
	int row_lo, row_hi;  // output rows to produce
	int nstrips, chunk;  // 32-px strips, output rows per work item
	int blk;             // output rows per block
	int rs, rc;          // ring sizes (rows, powers of two): source, C / G
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
};

__device__ __forceinline__ int gp_clamp( int i, int hi )
{
	return( i < 0 ? 0 : ( i > hi ? hi : i ));
}

// LDS-DMA of 64 x 16 bytes: lane l's 16 bytes land at lds + l * 16. Inline
// assembly on purpose: the compiler must not know that these loads write LDS,
// or it would drain them (vmcnt 0) before every LDS read that follows --
// completion is awaited explicitly, one block / row later.
__device__ __forceinline__ void gp_dma( const __amdgpu_buffer_rsrc_t rs,
	const unsigned lds, const int voff )
{
	asm volatile( "s_mov_b32 m0, %0\n\t"
		"buffer_load_dwordx4 %1, %2, 0 offen lds"
		:: "s"( __builtin_amdgcn_readfirstlane( lds )), "v"( voff ), "s"( rs )
		: "memory" );
}

#define GP_WAIT_VM( n ) asm volatile( "s_waitcnt vmcnt(" #n ")" ::: "memory" )

// One gather output from a ring / buffer of T pixels: tap t reads in[ t ].
// `ld( t )` returns the input of tap t, `cf( t )` its (uniform) coefficient.
template< bool LANC, typename T, typename LD, typename CF >
__device__ __forceinline__ T gp_dot( const int nt, LD ld, CF cf )
{
	if( LANC )
	{
		// (t0*p0 + t2*p2 + ...) + (t1*p1 + t3*p3 + ...), bare first products
		// (lancir.h:2482-2515); nt is even
		T ea = cf( 0 ) * ld( 0 );
		T oa = cf( 1 ) * ld( 1 );
#pragma unroll 4
		for( int t = 2; t < nt; t += 2 )
		{
			ea = ea + cf( t ) * ld( t );
			oa = oa + cf( t + 1 ) * ld( t + 1 );
		}

		return( ea + oa );
	}

	T acc = (T) 0.0f;
	int t = 0;

	for( ; t + 4 <= nt; t += 4 )
	{
		acc = acc + cf( t ) * ld( t );
		acc = acc + cf( t + 1 ) * ld( t + 1 );
		acc = acc + cf( t + 2 ) * ld( t + 2 );
		acc = acc + cf( t + 3 ) * ld( t + 3 );
	}

	for( ; t < nt; t++ )
	{
		acc = acc + cf( t ) * ld( t );
	}

	return( acc );
}

// ---------------------------------------------------------------------
// vertical pass
// ---------------------------------------------------------------------

template< int MODE, bool LANC >
__global__ void __launch_bounds__( 64 ) k_gv( const GVParams P )
{
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	const GPAxis& A = P.ax;
	f2* const sring = (f2*) smem;                       // [rs][32 px][2 halves]
	f2* const mring = sring + (size_t) P.rs * 64;       // [rc][32 px][2]
	const unsigned sring_lds = (unsigned) (unsigned long) (lds_char*) smem;

	const int lane = threadIdx.x;
	const int strip = blockIdx.x % P.nstrips;
	const int chunk = blockIdx.x / P.nstrips;
	const int x0 = strip * 32;
	const int y0 = P.row_lo + chunk * P.chunk;
	const int y1 = min( y0 + P.chunk, P.row_hi );

	if( y0 >= y1 )
	{
		return;
	}

	gp_cint* const a_start = (gp_cint*) (unsigned long) A.start;
	gp_cint* const a_fidx = (gp_cint*) (unsigned long) A.fidx;
	gp_cfloat* const a_coef = (gp_cfloat*) (unsigned long) A.coef;
	const int smask = P.rs - 1;
	const int mmask = P.rc - 1;
	const int slast = A.in_len - 1;
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
		(void*) P.src, 0, 0x7fffffff, 0x00020000 );

	// DMA lane -> (row parity, pixel): lanes 0-31 take row u, 32-63 row u + 1
	const int dpx = gp_clamp( x0 + ( lane & 31 ), P.width - 1 );
	const int drow = lane >> 5;
	const int srow_b = (int) P.src_ss * 4;

	// arithmetic lane -> half pixel
	const int hx = x0 + ( lane >> 1 );
	const bool xok = ( hx < P.width );
	const long dcol = (long) hx * 4 + ( lane & 1 ) * 2;

	// source rows [ua, ub] -> ring (aligned pairs; clamped row indices)
	auto issue_rows = [&]( int ua, const int ub )
	{
		for( int u = ua & ~1; u <= ub; u += 2 )
		{
			const int r = gp_clamp( u + drow, slast );
			gp_dma( rsrc, sring_lds + ( u & smask ) * 512,
				r * srow_b + dpx * 16 );
		}
	};

	// rows above the view's last replicated sample read 0 (zero-stuffed view,
	// avir.h:3260-3402): overwrite them once their DMA has landed
	auto zero_rows = [&]( const int ua, const int ub )
	{
		if( MODE != GP_PRE && ub > A.view_mmax )
		{
			for( int u = max( ua & ~1, A.view_mmax + 1 ); u <= ub + 1; u++ )
			{
				sring[ ( u & smask ) * 64 + lane ] = (f2) 0.0f;
			}
		}
	};

	auto sread = [&]( const int u ) -> f2
	{
		return( sring[ ( u & smask ) * 64 + lane ]);
	};

	auto mread = [&]( const int m ) -> f2
	{
		return( mring[ ( m & mmask ) * 64 + lane ]);
	};

	const f2 f0 = (f2) A.f[ 0 ], f1 = (f2) A.f[ 1 ], f2_ = (f2) A.f[ 2 ],
		f3 = (f2) A.f[ 3 ];

	// source row range a block of gather outputs [ga, gb] needs
	auto need_lo = [&]( const int ga ) -> int
	{
		if( MODE == GP_PRE )
		{
			// C rows from start[ga] on; C[m] reads S[clamp(m) - 6 .. clamp(m)]
			return( gp_clamp( a_start[ ga ], A.c_len - 1 ) - 2 * A.e );
		}

		return( a_start[ ga ]);
	};

	auto need_hi = [&]( const int gb ) -> int
	{
		const int h = a_start[ gb ] + A.nt - 1;

		if( MODE == GP_PRE )
		{
			return( gp_clamp( h, A.c_len - 1 ));
		}

		return( h );
	};

	// gather outputs a block of final outputs [n0, n1) needs
	auto g_lo = [&]( const int n0 ) -> int
	{
		return( MODE == GP_POST ? gp_clamp( n0 - A.e - 3, A.g_len - 1 ) : n0 );
	};

	auto g_hi = [&]( const int n1 ) -> int
	{
		return( MODE == GP_POST ? gp_clamp( n1 - 1 - A.e + 3, A.g_len - 1 ) :
			n1 - 1 );
	};

	int s_next;            // first source row not yet issued
	int m_next;            // PRE: first C row not yet computed; POST: first G row
	int n0 = y0;

	// prologue: rows of the first block
	{
		const int n1 = min( n0 + P.blk, y1 );
		const int ga = g_lo( n0 );
		const int gb = g_hi( n1 );
		const int ua = need_lo( ga );
		const int ub = need_hi( gb );
		issue_rows( ua, ub );
		s_next = ( ub | 1 ) + 1;
		m_next = ( MODE == GP_PRE ? a_start[ ga ] : ga );
	}

	while( n0 < y1 )
	{
		const int n1 = min( n0 + P.blk, y1 );
		const int ga = g_lo( n0 );
		const int gb = g_hi( n1 );

		// this block's rows have landed ...
		GP_WAIT_VM( 0 );
		zero_rows( need_lo( ga ), need_hi( gb ));

		// ... and the next block's go out before the arithmetic starts
		if( n1 < y1 )
		{
			const int nn1 = min( n1 + P.blk, y1 );
			const int ub = need_hi( g_hi( nn1 ));

			if( ub >= s_next )
			{
				issue_rows( s_next, ub );
				s_next = ( ub | 1 ) + 1;
			}
		}

		if( MODE == GP_PRE )
		{
			// C rows up to the last one this block reads
			const int mh = a_start[ gb ] + A.nt - 1;

			for( ; m_next <= mh; m_next++ )
			{
				f2 c;

				if( m_next > A.view_mmax )
				{
					c = (f2) 0.0f;
				}
				else
				{
					const int cc = gp_clamp( m_next, A.c_len - 1 ) - A.e;
					c = f0 * sread( cc );
					c = c + f1 * ( sread( cc + 1 ) + sread( cc - 1 ));
					c = c + f2_ * ( sread( cc + 2 ) + sread( cc - 2 ));
					c = c + f3 * ( sread( cc + 3 ) + sread( cc - 3 ));
				}

				mring[ ( m_next & mmask ) * 64 + lane ] = c;
			}
		}

		if( MODE == GP_POST )
		{
			// G rows up to the last one this block's FIR reads
			for( ; m_next <= gb; m_next++ )
			{
				const int s0 = a_start[ m_next ];
				gp_cfloat* cr = a_coef + (long) ( LANC ? a_fidx[ m_next ] :
					m_next ) * A.nt;

				const f2 g = gp_dot< LANC, f2 >( A.nt,
					[&]( const int t ) { return( sread( s0 + t )); },
					[&]( const int t ) { return( (f2) cr[ t ]); } );

				mring[ ( m_next & mmask ) * 64 + lane ] = g;
			}
		}

		for( int n = n0; n < n1; n++ )
		{
			f2 o;

			if( MODE == GP_POST )
			{
				const int glast = A.g_len - 1;
				const int c = n - A.e;
				o = f0 * mread( gp_clamp( c, glast ));
				o = o + f1 * ( mread( gp_clamp( c + 1, glast )) +
					mread( gp_clamp( c - 1, glast )));
				o = o + f2_ * ( mread( gp_clamp( c + 2, glast )) +
					mread( gp_clamp( c - 2, glast )));
				o = o + f3 * ( mread( gp_clamp( c + 3, glast )) +
					mread( gp_clamp( c - 3, glast )));
			}
			else
			{
				const int s0 = a_start[ n ];
				gp_cfloat* cr = a_coef + (long) ( LANC ? a_fidx[ n ] : n ) *
					A.nt;

				if( MODE == GP_PRE )
				{
					o = gp_dot< LANC, f2 >( A.nt,
						[&]( const int t ) { return( mread( s0 + t )); },
						[&]( const int t ) { return( (f2) cr[ t ]); } );
				}
				else
				{
					o = gp_dot< LANC, f2 >( A.nt,
						[&]( const int t ) { return( sread( s0 + t )); },
						[&]( const int t ) { return( (f2) cr[ t ]); } );
				}
			}

			if( xok )
			{
				*(f2*) ( P.dst + (long) ( n - P.dst_row0 ) * P.dst_ss + dcol ) =
					o;
			}
		}

		n0 = n1;
	}
}

// ---------------------------------------------------------------------
// horizontal pass
// ---------------------------------------------------------------------

template< int MODE, bool LANC, int NTC >
__global__ void __launch_bounds__( 64 ) k_gh( const GHParams P )
{
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	const GPAxis& A = P.ax;
	f4* const sbuf = (f4*) smem;                    // [2][sbuf px]
	f4* const mbuf = sbuf + 2 * (size_t) P.sbuf;    // [mbuf px]
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

	issue_row( r0, 0 );

	for( int r = r0; r < r1; r++ )
	{
		const int buf = ( r - r0 ) & 1;
		const f4* const s = sbuf + buf * P.sbuf;

		// next row's segment goes out, then wait for this row's: at most the
		// P.nseg pieces just issued may still be in flight (loads return in
		// order; the previous row's store may or may not have retired)
		if( r + 1 < r1 )
		{
			issue_row( r + 1, buf ^ 1 );

			switch( P.nseg )
			{
				case 1: GP_WAIT_VM( 1 ); break;
				case 2: GP_WAIT_VM( 2 ); break;
				case 3: GP_WAIT_VM( 3 ); break;
				case 4: GP_WAIT_VM( 4 ); break;
				default: GP_WAIT_VM( 0 ); break;
			}
		}
		else
		{
			GP_WAIT_VM( 0 );
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

			o = gp_dot< LANC, f4 >( nt,
				[&]( const int t ) { return( in[ t ]); },
				[&]( const int t ) { return( NTC > 0 ? cf[ NTC > 0 ? t : 0 ] :
					cfl[ t * 64 + lane ]); } );
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

		if( jok )
		{
			*(f4*) ( P.dst + (long) ( r - P.dst_row0 ) * P.dst_ss +
				(long) j * 4 ) = o;
		}
	}
}

// ---------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------

struct GPAxisHost
{
	bool ok;
	GPAxis a;       // device pointers filled in
	double kmax;    // largest source advance per output (ring sizing)
	std::vector< int > start; // host copy of the positions
};

struct GPData
{
	GPAxisHost h, v;     // AVIR: H pass first; LANCIR: V pass first
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
		fir -> view != VIEW_CLAMP || fir -> e != 3 ))
	{
		return( false );
	}

	if( gat -> view == VIEW_RAW || gat -> maxtaps < 2 || gat -> maxtaps > 40 )
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
	A.e = 3;
	A.c_len = 0;

	if( mode == GP_PRE )
	{
		if( fir -> in_len != ax.src_len || gat -> in_len != fir -> out_len ||
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
	H.ok = true;
	return( true );
}

static void match_lancir_axis( const LancirAxisDev& L, GPAxisHost& H )
{
	GPAxis& A = H.a;
	memset( &A, 0, sizeof( A ));
	H.ok = ( L.kernel_len >= 2 && L.kernel_len <= 40 &&
		( L.kernel_len & 1 ) == 0 );

	A.mode = GP_NONE; A.lanc = 1; A.nt = L.kernel_len;
	A.in_len = L.src_len; A.view_mmax = 0x7fffffff;
	A.g_len = L.dst_len; A.out_len = L.dst_len; A.e = 0;
	A.start = L.d_start; A.coef = L.d_flt; A.fidx = L.d_fidx;
	H.start = L.h_start;

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

	if( p -> is_lancir )
	{
		match_lancir_axis( p -> lv, D -> v );
		match_lancir_axis( p -> lh, D -> h );

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

	// ---- horizontal pass geometry: strips of `ow` outputs whose gather
	// inputs (and, PRE, their FIR inputs) fit 64-lane pieces
	{
		const GPAxis& A = D -> h.a;
		const std::vector< int >& st = D -> h.start;
		int ow = ( A.mode == GP_POST ? 58 : 64 );
		std::vector< int > seg;
		int nstrips = 0, maxs = 0, maxm = 0;

		for( ; ow >= 8; ow -= ( ow > 16 ? 6 : 2 ))
		{
			nstrips = ( A.out_len + ow - 1 ) / ow;
			seg.assign( (size_t) nstrips * 4, 0 );
			maxs = 0; maxm = 0;

			for( int s = 0; s < nstrips; s++ )
			{
				const int ja = s * ow;
				const int jb = std::min( A.out_len, ja + ow ) - 1;
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

				seg[ s * 4 ] = sa; seg[ s * 4 + 1 ] = sb - sa + 1;
				seg[ s * 4 + 2 ] = ma; seg[ s * 4 + 3 ] = mb - ma + 1;
				maxs = std::max( maxs, sb - sa + 1 );
				maxm = std::max( maxm, mb - ma + 1 );
			}

			// the FIR and POST stages run one buffer entry per lane and pass
			const bool fits = ( maxs <= 256 && ( A.mode == GP_NONE ||
				maxm <= ( A.mode == GP_POST ? 64 : 128 )));

			if( fits )
			{
				break;
			}
		}

		if( ow < 8 )
		{
			delete D;
			return( AVIRHIP_OK );
		}

		D -> h_ow = ow;
		D -> h_nstrips = nstrips;
		D -> h_nseg = ( maxs + 63 ) / 64;
		D -> h_sbuf = D -> h_nseg * 64;
		D -> h_mbuf = std::max( 64, (( maxm + 63 ) / 64 ) * 64 );
		void* q = nullptr;
		AVIRHIP_HIPCHECK( hipMalloc( &q, seg.size() * sizeof( int )));
		D -> allocs.push_back( q );
		AVIRHIP_HIPCHECK( hipMemcpy( q, seg.data(), seg.size() * sizeof( int ),
			hipMemcpyHostToDevice ));
		D -> d_hseg = (int*) q;
	}

	// ---- vertical pass geometry: block height and ring sizes from the
	// largest spans any block has
	{
		const GPAxis& A = D -> v.a;
		const std::vector< int >& st = D -> v.start;
		int blk = 8;

		for( ; blk >= 1; blk >>= 1 )
		{
			int span_s = 0, span_m = 0;

			for( int n0 = 0; n0 < A.out_len; n0++ )
			{
				// any block start (chunks begin at arbitrary rows in bands)
				const int n1 = std::min( A.out_len, n0 + blk );
				int ga = n0, gb = n1 - 1;

				if( A.mode == GP_POST )
				{
					ga = std::max( 0, std::min( n0 - A.e - 3, A.g_len - 1 ));
					gb = std::max( 0, std::min( n1 - 1 - A.e + 3, A.g_len - 1 ));
				}

				// two consecutive blocks live in the source ring at once
				const int n2 = std::min( A.out_len, n1 + blk );
				int gb2 = n2 - 1;

				if( A.mode == GP_POST )
				{
					gb2 = std::max( 0, std::min( n2 - 1 - A.e + 3, A.g_len - 1 ));
				}

				int ua = st[ ga ], ub = st[ gb2 ] + A.nt - 1;

				if( A.mode == GP_PRE )
				{
					ua = std::max( 0, std::min( ua, A.c_len - 1 )) - 2 * A.e;
					ub = std::max( 0, std::min( ub, A.c_len - 1 ));
					span_m = std::max( span_m, st[ gb ] + A.nt - st[ ga ]);
				}
				else
				if( A.mode == GP_POST )
				{
					span_m = std::max( span_m, gb - ga + 1 + 7 );
				}

				span_s = std::max( span_s, ub - ua + 1 + 4 );
			}

			const int rs = next_pow2( span_s );
			const int rc = ( A.mode == GP_NONE ? 2 : next_pow2( span_m + 1 ));

			if(( rs + rc ) * 512 <= 48 * 1024 || blk == 1 )
			{
				if(( rs + rc ) * 512 > 64 * 1024 )
				{
					delete D;
					return( AVIRHIP_OK );
				}

				D -> v_blk = blk; D -> v_rs = rs; D -> v_rc = rc;
				break;
			}
		}
	}

	// row offsets travel as 32-bit byte offsets
	const long mid_w = ( p -> is_lancir ? p -> src_w : p -> new_w );
	const long mid_h = ( p -> is_lancir ? p -> new_h : p -> src_h );

	if( (long) p -> src_h * p -> src_stride * 4 >= ( 1L << 31 ) ||
		mid_h * mid_w * 16 >= ( 1L << 31 ))
	{
		delete D;
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

		delete D;
	}

	p -> gpass = nullptr;
}

bool gpass_ok( const avirhip_plan* p )
{
	return( p -> gpass != nullptr );
}

template< int MODE, bool LANC >
static void launch_gv( const GVParams& P, int items, size_t lds,
	hipStream_t st )
{
	hipLaunchKernelGGL(( k_gv< MODE, LANC > ), dim3( items ), dim3( 64 ), lds,
		st, P );
}

template< int MODE, bool LANC >
static void launch_gh( const GHParams& P, int items, size_t lds,
	hipStream_t st )
{
	switch( P.ax.nt )
	{
		case 6: hipLaunchKernelGGL(( k_gh< MODE, LANC, 6 > ), dim3( items ),
			dim3( 64 ), lds, st, P ); break;
		case 12: hipLaunchKernelGGL(( k_gh< MODE, LANC, 12 > ), dim3( items ),
			dim3( 64 ), lds, st, P ); break;
		default: hipLaunchKernelGGL(( k_gh< MODE, LANC, 0 > ), dim3( items ),
			dim3( 64 ), lds, st, P ); break;
	}
}

static int run_v( const GPData* D, const float* src, long src_ss, int width,
	float* dst, long dst_ss, int dst_row0, int row_lo, int row_hi,
	hipStream_t st )
{
	GVParams P;
	P.src = src; P.src_ss = src_ss; P.width = width;
	P.dst = dst; P.dst_ss = dst_ss; P.dst_row0 = dst_row0;
	P.ax = D -> v.a;
	P.row_lo = row_lo; P.row_hi = row_hi;
	P.nstrips = ( width + 31 ) / 32;
	P.blk = D -> v_blk; P.rs = D -> v_rs; P.rc = D -> v_rc;

	// chunk height: enough work items to fill the CUs a few times over, few
	// enough that a chunk's warm-up rows (its first window) stay a small part
	const int rows = row_hi - row_lo;
	int nch = std::max( 1, std::min( rows / ( 4 * P.blk ),
		( 4096 + P.nstrips - 1 ) / P.nstrips ));

	P.chunk = (( rows + nch - 1 ) / nch + P.blk - 1 ) / P.blk * P.blk;
	nch = ( rows + P.chunk - 1 ) / P.chunk;
	const int items = P.nstrips * nch;
	const size_t lds = (size_t) ( P.rs + P.rc ) * 512;
	const GPAxis& A = P.ax;

	if( A.lanc )
	{
		launch_gv< GP_NONE, true >( P, items, lds, st );
	}
	else
	if( A.mode == GP_PRE )
	{
		launch_gv< GP_PRE, false >( P, items, lds, st );
	}
	else
	if( A.mode == GP_POST )
	{
		launch_gv< GP_POST, false >( P, items, lds, st );
	}
	else
	{
		launch_gv< GP_NONE, false >( P, items, lds, st );
	}

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

static int run_h( const GPData* D, const float* src, long src_ss, int src_w,
	float* dst, long dst_ss, int dst_row0, int dst_w, int row_lo, int row_hi,
	hipStream_t st )
{
	GHParams P;
	P.src = src; P.src_ss = src_ss; P.src_w = src_w;
	P.dst = dst; P.dst_ss = dst_ss; P.dst_row0 = dst_row0; P.dst_w = dst_w;
	P.ax = D -> h.a;
	P.row_lo = row_lo; P.row_hi = row_hi;
	P.nstrips = D -> h_nstrips; P.ow = D -> h_ow;
	P.seg = D -> d_hseg; P.nseg = D -> h_nseg;
	P.sbuf = D -> h_sbuf; P.mbuf = D -> h_mbuf;

	const int rows = row_hi - row_lo;
	int nch = std::max( 1, std::min( rows / 8,
		( 8192 + P.nstrips - 1 ) / P.nstrips ));

	P.chunk = ( rows + nch - 1 ) / nch;
	nch = ( rows + P.chunk - 1 ) / P.chunk;
	const int items = P.nstrips * nch;
	const size_t lds = (size_t) ( 2 * P.sbuf + P.mbuf ) * 16 +
		(size_t) P.ax.nt * 64 * 4;
	const GPAxis& A = P.ax;

	if( A.lanc )
	{
		launch_gh< GP_NONE, true >( P, items, lds, st );
	}
	else
	if( A.mode == GP_PRE )
	{
		launch_gh< GP_PRE, false >( P, items, lds, st );
	}
	else
	if( A.mode == GP_POST )
	{
		launch_gh< GP_POST, false >( P, items, lds, st );
	}
	else
	{
		launch_gh< GP_NONE, false >( P, items, lds, st );
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
int gpass_run( avirhip_plan* p, const float* src, long src_stride, float* dst,
	int row0, int row1, hipStream_t st )
{
	GPData* D = (GPData*) p -> gpass;

	if( D == nullptr || ( (uintptr_t) src & 15 ) || ( (uintptr_t) dst & 15 ) ||
		( src_stride & 3 ) || ( p -> new_stride & 3 ))
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
		// vertical first (lancir.h:601-646): mid = [new_h][src_w], only the
		// rows of the band
		const long mid_ss = (long) p -> src_w * 4;

		if(( rc = run_v( D, src, src_stride, p -> src_w, D -> mid, mid_ss, 0,
			row0, row1, st )) != 0 ) return( rc );

		return( run_h( D, D -> mid, mid_ss, p -> src_w, dst, dst_ss, row0,
			p -> new_w, row0, row1, st ));
	}

	// AVIR: horizontal first into mid = FltBuf [src_h][new_w], only the source
	// rows the band's vertical windows read
	const long mid_ss = (long) p -> new_w * 4;
	int a, b;
	v_source_rows( D, row0, row1, a, b );

	if(( rc = run_h( D, src, src_stride, p -> src_w, D -> mid, mid_ss, 0,
		p -> new_w, a, b + 1, st )) != 0 ) return( rc );

	return( run_v( D, D -> mid, mid_ss, p -> new_w, dst, dst_ss, row0, row0,
		row1, st ));
}

} // namespace avirhip
