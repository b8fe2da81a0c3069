// sacc.hip -- the downsizing pass kernel: STREAMING ACCUMULATION.
//
// A downsizing axis (AVIR k >= 2: doResize's NT-tap dot products, then the
// correction FIR, avir.h:3884-4110 + 3748-3866; LANCIR k > 1, lancir.h:
// 2466-2544) has long filters -- NT grows with k, 24 .. 64 taps -- but few
// outputs alive at a time: an input sample feeds about NT / k = 12 .. 14
// outputs whatever k is. The gather kernels of gpass.hip keep an NT-deep
// window per lane (LDS rings, 33 ds_read_b128 per output, one or two waves
// per SIMD); here the window does not exist at all:
//
//   * a lane owns one line of samples ALONG the axis (vertical pass: a pixel
//     column of a 64-pixel strip; horizontal pass: an image ROW of a 64-row
//     strip) and walks it sample by sample;
//   * sample u is multiplied into the accumulators of the outputs alive at u.
//     Samples arrive in ascending order, so every accumulator sees its taps in
//     the reference's order, starting from +0.0f: bit-exact;
//   * the coefficients of the (outputs alive) x (sample) pairs are wave-uniform:
//     the host lays them out by SAMPLE (16 floats per sample, first alive output
//     first), the kernel reads a sample's row with scalar loads one sample
//     ahead -- no LDS, no vector registers for coefficients;
//   * accumulator slots are compile-time: the loop over outputs is unrolled 16
//     times (output g lives in slot g % 16), the loop over the samples that
//     complete an output is dynamic.
//
// Both passes are the same kernel: only the lane and step strides differ.
// Samples reach the lanes through a small LDS FIFO (8 slots of 64 x 16 B):
// float RGBA sources by LDS-DMA (per-lane addresses: a row of a strip, or 64
// rows of a column), integer / narrower sources (the caller's image, first
// pass only) by per-lane wide loads of whole 4-, 8-, 12- or 16-byte groups
// converted in registers -- packScanline's (float) cast and zero padding
// (avir.h:2777-2840) without a float copy of the source.

#include "gpass_dev.h"
#include <algorithm>
#include <type_traits>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>

namespace avirhip {

#define SA_NS 16 // accumulator slots: the most outputs alive at one sample
#define SA_PF 5  // DMA samples in flight
#define SA_Q 8   // FIFO slots (+ 1 dump slot)

struct SAAxis
{
	int nt, in_len, g_len, out_len, e, lanc;
	float f[ 4 ];
	int u_min;          // view index of table row 0
	const int* start;   // [g_len] first input of each output (unclamped)
	const int* meta;    // [rows][2] first, last output alive at the sample
	const float* tab;   // [rows][16] their coefficients for the sample
};

struct SAParams
{
	const void* src; long s_lane, s_step; // bytes between lanes / samples
	float* dst; long d_lane, d_step;      // floats between lanes / outputs
	int lane_lo, lane_hi;                 // lanes [lo, hi): rows or pixels
	int out_lo, out_hi;                   // outputs [lo, hi)
	int nstrips, chunk;
	SAAxis ax;
};

typedef unsigned int sa_u32u __attribute__(( aligned( 1 )));

template< int SRC > struct SAElem;
template<> struct SAElem< 1 > { typedef uint8_t T; };
template<> struct SAElem< 2 > { typedef uint16_t T; };
template<> struct SAElem< 3 > { typedef float T; };
template<> struct SAElem< 0 > { typedef float T; };

// SRC: 0 float RGBA by LDS-DMA; 1 uint8, 2 uint16, 3 float of CH channels.
template< int SRC, int CH, bool POST >
__global__ void __launch_bounds__( 64 ) k_sacc( const SAParams P )
{
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	const SAAxis& A = P.ax;
	f4* const fifo = (f4*) smem;                  // [SA_Q + 1][64 lanes]
	f4* const mring = fifo + ( SA_Q + 1 ) * 64;   // POST: [8][64 lanes]

	const int lane = threadIdx.x;
	const int strip = blockIdx.x % P.nstrips;
	const int chunk = blockIdx.x / P.nstrips;
	const int li = P.lane_lo + strip * 64 + lane;
	const bool lok = ( li < P.lane_hi );
	const int lc = min( li, P.lane_hi - 1 );      // lanes past the end reload
	const int n0 = P.out_lo + chunk * P.chunk;
	const int n1 = min( n0 + P.chunk, P.out_hi );

	if( n0 >= n1 )
	{
		return;
	}

	gp_cint* const a_start = (gp_cint*) (unsigned long) A.start;
	gp_cint* const a_meta = (gp_cint*) (unsigned long) A.meta;
	gp_cfloat* const a_tab = (gp_cfloat*) (unsigned long) A.tab;
	const int nt = A.nt;
	const int glast = A.g_len - 1;
	const int slast = A.in_len - 1;

	// accumulated outputs this chunk needs: [ga, gb]
	const int ga = ( POST ? gp_clamp( n0 - A.e - 3, glast ) : n0 );
	const int gb = ( POST ? gp_clamp( n1 - 1 - A.e + 3, glast ) : n1 - 1 );
	const int u_a = a_start[ ga ];
	const int u_b = a_start[ gb ] + nt - 1;

	// ---- the feed
	constexpr int G = ( SRC == 1 ? 4 : ( SRC == 2 ? 2 : 1 )); // samples / group
	typedef typename SAElem< SRC > :: T Tin;
	// the FIFO slot of sample u is ( u - ub0 ) & 7; groups are aligned in
	// absolute sample coordinates
	const int ub0 = ( SRC == 0 ? u_a : ( u_a >= 0 ? u_a / G : -(( -u_a + G - 1 ) / G )) * G );
	const unsigned lds0 = (unsigned) (unsigned long) (lds_char*) smem;

	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
		(void*) P.src, 0, 0x7fffffff, 0x00020000 );
	const int lofs = lc * (int) P.s_lane;
	const int sstep = (int) P.s_step;

	auto issue = [&]( const int u ) // SRC == 0
	{
		const bool live = ( u <= u_b );
		const int x = gp_clamp( min( u, u_b ), slast );
		const unsigned slot = ( live ? (unsigned) (( u - ub0 ) & ( SA_Q - 1 )) :
			(unsigned) SA_Q );

		gp_dma( rsrc, lds0 + slot * 1024, lofs + x * sstep );
	};

	const unsigned char* const lrow = (const unsigned char*) P.src +
		(long) lc * P.s_lane;

	unsigned int gw[ CH ]; // the group in flight (typed sources)
#pragma unroll
	for( int k = 0; k < CH; k++ )
	{
		gw[ k ] = 0;
	}

	auto g_interior = [&]( const int q ) -> bool
	{
		return( q * G >= 0 && q * G + G - 1 <= slast );
	};

	auto g_load = [&]( const int q ) // typed: whole group, one wide load
	{
		if( g_interior( q ) && q * G <= u_b )
		{
			const sa_u32u* const gp = (const sa_u32u*) ( lrow + (long) q * G *
				CH * sizeof( Tin ));
#pragma unroll
			for( int k = 0; k < CH; k++ )
			{
				gw[ k ] = gp[ k ];
			}
		}
	};

	auto g_store = [&]( const int q ) // typed: convert, write the FIFO slots
	{
		const int sl0 = ( q * G - ub0 ) & ( SA_Q - 1 );

		if( g_interior( q ))
		{
#pragma unroll
			for( int j = 0; j < G; j++ )
			{
				float v[ 4 ] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
				for( int c = 0; c < CH; c++ )
				{
					const int ei = j * CH + c;

					if( SRC == 1 )
					{
						v[ c ] = (float) (( gw[ ei >> 2 ] >> (( ei & 3 ) * 8 )) &
							0xffu );
					}
					else
					if( SRC == 2 )
					{
						v[ c ] = (float) (( gw[ ei >> 1 ] >> (( ei & 1 ) * 16 )) &
							0xffffu );
					}
					else
					{
						v[ c ] = __uint_as_float( gw[ ei ]);
					}
				}

				f4 o; o.x = v[ 0 ]; o.y = v[ 1 ]; o.z = v[ 2 ]; o.w = v[ 3 ];
				fifo[ ( sl0 + j ) * 64 + lane ] = o;
			}
		}
		else
		{
			// a group on the image's edge: the view repeats the edge sample
			for( int j = 0; j < G; j++ )
			{
				const int x = gp_clamp( q * G + j, slast );
				const Tin* const px = (const Tin*) lrow + (long) x * CH;
				float v[ 4 ] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
				for( int c = 0; c < CH; c++ )
				{
					v[ c ] = (float) px[ c ];
				}

				f4 o; o.x = v[ 0 ]; o.y = v[ 1 ]; o.z = v[ 2 ]; o.w = v[ 3 ];
				fifo[ ( sl0 + j ) * 64 + lane ] = o;
			}
		}
	};

	const int q_a = ( ub0 >= 0 ? ub0 / G : -(( -ub0 ) / G )); // ub0 is aligned

	if( SRC == 0 )
	{
		for( int k = 0; k < SA_PF; k++ )
		{
			issue( u_a + k );
		}

		GP_WAIT_VM( 4 ); // == SA_PF - 1: the first sample has landed
	}
	else
	{
		// (as if group q_a had just been entered: the next one is in the FIFO
		// too, the one after it in flight)
		g_load( q_a );
		g_store( q_a );
		g_load( q_a + 1 );
		g_store( q_a + 1 );
		g_load( q_a + 2 );
	}

	f4 qnext = fifo[ (( u_a - ub0 ) & ( SA_Q - 1 )) * 64 + lane ];

	// ---- the accumulators
	f4 acc[ SA_NS ];
#pragma unroll
	for( int i = 0; i < SA_NS; i++ )
	{
		acc[ i ] = (f4) 0.0f;
	}

	// ---- scalar pipeline: sample u's coefficient row and last alive output
	// are loaded during sample u - 1, its first alive output during u - 2
	auto meta_at = [&]( const int u, const int k ) -> int
	{
		return( a_meta[ ( u - A.u_min ) * 2 + k ]);
	};

	float cc[ 16 ];
	int lastc, f1;
	{
		const int lo = max( 0, ga - meta_at( u_a, 0 ));
		gp_cfloat* const tr = a_tab + ( u_a - A.u_min ) * 16 + lo;
#pragma unroll
		for( int i = 0; i < 16; i++ )
		{
			cc[ i ] = tr[ i ];
		}

		lastc = meta_at( u_a, 1 );
		f1 = meta_at( u_a + 1, 0 );
	}

	int g = ga;                    // the output being completed
	int sg = u_a;                  // its first input
	int sgn = a_start[ min( g + 1, gb )];
	int u = u_a;                   // the next sample
	int n = n0;                    // POST: the next final output

	const f4 f0 = (f4) A.f[ 0 ], f1c = (f4) A.f[ 1 ], f2c = (f4) A.f[ 2 ],
		f3c = (f4) A.f[ 3 ];

	float* dp = P.dst + (long) li * P.d_lane + (long) n0 * P.d_step;

	auto store_out = [&]( const f4 o )
	{
		if( lok )
		{
			*(f4*) dp = o;
		}

		dp += P.d_step;
	};

	auto mread = [&]( const int m ) -> f4
	{
		return( mring[ ( m & 7 ) * 64 + lane ]);
	};

	auto emit = [&]( const f4 o )
	{
		if( POST )
		{
			mring[ ( g & 7 ) * 64 + lane ] = o;

			// final outputs whose FIR window [c - 3, c + 3] (clamped) is
			// complete: c + 3 <= g, or g is the last accumulated output
			while( n < n1 && gp_clamp( n - A.e + 3, glast ) <= g )
			{
				const int c = n - A.e;
				f4 r = f0 * mread( gp_clamp( c, glast ));
				r = r + f1c * ( mread( gp_clamp( c + 1, glast )) +
					mread( gp_clamp( c - 1, glast )));
				r = r + f2c * ( mread( gp_clamp( c + 2, glast )) +
					mread( gp_clamp( c - 2, glast )));
				r = r + f3c * ( mread( gp_clamp( c + 3, glast )) +
					mread( gp_clamp( c - 3, glast )));
				store_out( r );
				n++;
			}
		}
		else
		{
			store_out( o );
		}
	};

	// One output: the samples up to its last input, then the accumulator
	// leaves. SC: its slot (compile-time). Returns true after the last one.
	auto body = [&]( auto SC ) -> bool
	{
		constexpr int S = decltype( SC )::value;
		const int end = sg + nt - 1;

		while( u <= end )
		{
			// -- the feed: sample u + 1 becomes readable
			if( SRC == 0 )
			{
				issue( u + SA_PF );
				GP_WAIT_VM( 4 ); // == SA_PF - 1
			}
			else
			if((( u - ub0 ) & ( G - 1 )) == 0 && u != u_a )
			{
				const int q = q_a + (( u - ub0 ) / G );
				g_store( q + 1 );
				g_load( q + 2 );
			}

			const f4 x = qnext;
			qnext = fifo[ (( u + 1 - ub0 ) & ( SA_Q - 1 )) * 64 + lane ];

			// -- the scalar pipeline: sample u + 1's row
			float cn[ 16 ];
			const int lo1 = max( 0, ga - f1 );
			gp_cfloat* const tr = a_tab + ( u + 1 - A.u_min ) * 16 + lo1;
#pragma unroll
			for( int i = 0; i < 16; i++ )
			{
				cn[ i ] = tr[ i ];
			}

			const int lastn = meta_at( u + 1, 1 );
			const int f2 = meta_at( u + 2, 0 );

			// -- sample u into the outputs alive: g .. min( last, gb )
			const int nl = min( lastc, gb ) - g + 1;

			// One wave-uniform branch per alive output, nested: the first
			// output that is not alive leaves the chain. (A switch with
			// fall-through entries was structurised into a decision tree with
			// thousands of register copies; un-nested ifs were if-converted
			// into all 16 products and four selects each -- the empty asm
			// statement keeps every block a real branch.)
#define SA_T( i, inner ) if( i < nl ) { asm volatile( "" ::: "memory" ); \
	acc[ ( S + i ) & ( SA_NS - 1 )] = acc[ ( S + i ) & ( SA_NS - 1 )] + \
	cc[ i ] * x; inner }

			SA_T( 0, SA_T( 1, SA_T( 2, SA_T( 3, SA_T( 4, SA_T( 5, SA_T( 6,
			SA_T( 7, SA_T( 8, SA_T( 9, SA_T( 10, SA_T( 11, SA_T( 12,
			SA_T( 13, SA_T( 14, SA_T( 15, ; ))))))))))))))))

#undef SA_T

#pragma unroll
			for( int i = 0; i < 16; i++ )
			{
				cc[ i ] = cn[ i ];
			}

			lastc = lastn;
			f1 = f2;
			u++;
		}

		emit( acc[ S ]);
		acc[ S ] = (f4) 0.0f;
		g++;
		sg = sgn;
		sgn = a_start[ min( g + 1, gb )];
		return( g > gb );
	};

	for( ; ; )
	{
		if( body( std::integral_constant< int, 0 >() )) break;
		if( body( std::integral_constant< int, 1 >() )) break;
		if( body( std::integral_constant< int, 2 >() )) break;
		if( body( std::integral_constant< int, 3 >() )) break;
		if( body( std::integral_constant< int, 4 >() )) break;
		if( body( std::integral_constant< int, 5 >() )) break;
		if( body( std::integral_constant< int, 6 >() )) break;
		if( body( std::integral_constant< int, 7 >() )) break;
		if( body( std::integral_constant< int, 8 >() )) break;
		if( body( std::integral_constant< int, 9 >() )) break;
		if( body( std::integral_constant< int, 10 >() )) break;
		if( body( std::integral_constant< int, 11 >() )) break;
		if( body( std::integral_constant< int, 12 >() )) break;
		if( body( std::integral_constant< int, 13 >() )) break;
		if( body( std::integral_constant< int, 14 >() )) break;
		if( body( std::integral_constant< int, 15 >() )) break;
	}

	if( SRC == 0 )
	{
		GP_WAIT_VM( 0 ); // no LDS-DMA may outlive the workgroup's LDS
	}
}

// ---------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------

struct SAData
{
	SAAxis a;
	double k; // average input advance per output
};

static int sa_upload( std::vector< void* >& allocs, const void* h, size_t bytes,
	void** out )
{
	void* q = nullptr;
	AVIRHIP_HIPCHECK( hipMalloc( &q, bytes ));
	allocs.push_back( q );
	AVIRHIP_HIPCHECK( hipMemcpy( q, h, bytes, hipMemcpyHostToDevice ));
	*out = q;
	return( AVIRHIP_OK );
}

// Builds the by-sample tables of a recognised axis; *out stays nullptr if the
// axis is not a streaming-accumulation case (more than 16 outputs alive at a
// sample -- upsizing --, zero-stuffed views, LANCIR's split sums).
int sacc_prepare_axis( const GPAxisHost& H, std::vector< void* >& allocs,
	SAData** out )
{
	*out = nullptr;
	const GPAxis& A = H.a;

	if( !H.ok || A.mode == GP_PRE || A.lanc || A.view_mmax != 0x7fffffff ||
		A.g_len < 1 || getenv( "AVIRHIP_NO_SACC" ) != nullptr )
	{
		return( AVIRHIP_OK );
	}

	const std::vector< int >& st = H.start;
	const int nt = A.nt;
	const int u_min = st[ 0 ];
	const int u_max = st[ A.g_len - 1 ] + nt - 1;
	const int rows = u_max - u_min + 1;
	const int pad = 4;
	std::vector< int > meta( (size_t) ( rows + pad ) * 2, 0 );
	std::vector< float > tab( (size_t) ( rows + pad ) * 16, 0.0f );
	int first = 0, last = 0;

	for( int r = 0; r < rows; r++ )
	{
		const int u = u_min + r;

		while( first < A.g_len - 1 && st[ first ] + nt - 1 < u )
		{
			first++;
		}

		while( last < A.g_len - 1 && st[ last + 1 ] <= u )
		{
			last++;
		}

		if( last - first + 1 > SA_NS || last < first )
		{
			return( AVIRHIP_OK );
		}

		meta[ (size_t) r * 2 ] = first;
		meta[ (size_t) r * 2 + 1 ] = last;

		for( int g = first; g <= last; g++ )
		{
			const float* const cr = H.h_coef + (size_t) ( H.h_fidx != nullptr ?
				H.h_fidx[ g ] : g ) * nt;

			tab[ (size_t) r * 16 + ( g - first )] = cr[ u - st[ g ]];
		}
	}

	for( int r = rows; r < rows + pad; r++ )
	{
		meta[ (size_t) r * 2 ] = A.g_len - 1;
		meta[ (size_t) r * 2 + 1 ] = A.g_len - 1;
	}

	SAData* D = new SAData();
	memset( &D -> a, 0, sizeof( D -> a ));
	D -> a.nt = nt; D -> a.in_len = A.in_len; D -> a.g_len = A.g_len;
	D -> a.out_len = A.out_len; D -> a.e = A.e; D -> a.lanc = 0;
	memcpy( D -> a.f, A.f, sizeof( A.f ));
	D -> a.u_min = u_min;
	D -> a.start = A.start;
	D -> k = (double) rows / A.g_len;

	void* q = nullptr;
	int rc = sa_upload( allocs, meta.data(), meta.size() * sizeof( int ), &q );
	D -> a.meta = (const int*) q;

	if( rc == 0 )
	{
		rc = sa_upload( allocs, tab.data(), tab.size() * sizeof( float ), &q );
		D -> a.tab = (const float*) q;
	}

	if( rc != 0 )
	{
		delete D;
		return( rc );
	}

	*out = D;
	return( AVIRHIP_OK );
}

void sacc_release_axis( SAData* D )
{
	delete D;
}

template< int SRC, int CH >
static void sa_launch2( const SAParams& P, bool post, int items, size_t lds,
	hipStream_t st )
{
	if( post )
	{
		hipLaunchKernelGGL(( k_sacc< SRC, CH, true > ), dim3( items ),
			dim3( 64 ), lds, st, P );
	}
	else
	{
		hipLaunchKernelGGL(( k_sacc< SRC, CH, false > ), dim3( items ),
			dim3( 64 ), lds, st, P );
	}
}

template< int SRC >
static void sa_launch1( const SAParams& P, int ch, bool post, int items,
	size_t lds, hipStream_t st )
{
	switch( ch )
	{
		case 1: sa_launch2< SRC, 1 >( P, post, items, lds, st ); break;
		case 2: sa_launch2< SRC, 2 >( P, post, items, lds, st ); break;
		case 3: sa_launch2< SRC, 3 >( P, post, items, lds, st ); break;
		default: sa_launch2< SRC, 4 >( P, post, items, lds, st ); break;
	}
}

// One pass. src_type / src_ch: AVIRHIP_F32 with 4 channels goes by LDS-DMA
// (16-byte aligned base and strides required: returns 1 otherwise); U8, U16
// and narrower floats are read as they are. Strides in bytes (source) and
// floats (destination); `mode` is the axis' GP_POST / GP_NONE.
int sacc_run_axis( const SAData* D, int mode, const void* src, int src_type,
	int src_ch, long s_lane, long s_step, float* dst, long d_lane, long d_step,
	int lane_lo, int lane_hi, int out_lo, int out_hi, hipStream_t st )
{
	if( lane_hi <= lane_lo || out_hi <= out_lo )
	{
		return( AVIRHIP_OK );
	}

	SAParams P;
	P.src = src; P.s_lane = s_lane; P.s_step = s_step;
	P.dst = dst; P.d_lane = d_lane; P.d_step = d_step;
	P.lane_lo = lane_lo; P.lane_hi = lane_hi;
	P.out_lo = out_lo; P.out_hi = out_hi;
	P.ax = D -> a;

	const bool dma = ( src_type == AVIRHIP_F32 && src_ch == 4 );

	if( dma && (( (uintptr_t) src & 15 ) || ( s_lane & 15 ) || ( s_step & 15 )))
	{
		return( 1 );
	}

	if( (uintptr_t) dst & 15 || ( d_lane & 3 ) || ( d_step & 3 ))
	{
		return( 1 );
	}

	const int lanes = lane_hi - lane_lo;
	const int nout = out_hi - out_lo;
	P.nstrips = ( lanes + 63 ) / 64;

	// Outputs per work item: a chunk's only overhead is feeding the NT
	// samples before its first output completes (they are multiplied into
	// this chunk's outputs only: no arithmetic is repeated). The kernel is
	// ALU-bound: whole rounds of the 1024 SIMDs, two or more waves on each.
	const size_t lds = (size_t) ( SA_Q + 1 ) * 1024 +
		( mode == GP_POST ? 8 * 1024 : 0 );

	const double cap = 256.0 * std::max( 1, std::min( 16,
		(int) ( 160 * 1024 / ( lds + 512 ))));

	const double warm = std::max( 1.0, P.ax.nt / std::max( 1.0, D -> k ) *
		0.35 ) + ( mode == GP_POST ? 2.0 : 0.0 );

	double best = -1.0;
	int best_chunk = nout;

	for( int nch = 1; nch <= nout; nch++ )
	{
		const int chunk = ( nout + nch - 1 ) / nch;

		if( chunk < 4 && nch > 1 )
		{
			break;
		}

		const double items = (double) (( nout + chunk - 1 ) / chunk ) *
			P.nstrips;

		const double res = ceil( items / cap );
		const double per = ceil( items / res / 1024.0 );
		const double cost = res * per * ( chunk + warm ) *
			( per < 2.0 ? 1.26 : 1.0 );

		if( best < 0.0 || cost < best )
		{
			best = cost;
			best_chunk = chunk;
		}
	}

	P.chunk = best_chunk;

	if( getenv( "AVIRHIP_SA_CHUNK" ) != nullptr ) // tuning aid
	{
		P.chunk = std::max( 1, atoi( getenv( "AVIRHIP_SA_CHUNK" )));
	}

	const int items = P.nstrips * (( nout + P.chunk - 1 ) / P.chunk );
	const bool post = ( mode == GP_POST );

	if( dma )
	{
		sa_launch2< 0, 4 >( P, post, items, lds, st );
	}
	else
	if( src_type == AVIRHIP_U8 )
	{
		sa_launch1< 1 >( P, src_ch, post, items, lds, st );
	}
	else
	if( src_type == AVIRHIP_U16 )
	{
		sa_launch1< 2 >( P, src_ch, post, items, lds, st );
	}
	else
	if( src_type == AVIRHIP_F32 )
	{
		sa_launch1< 3 >( P, src_ch, post, items, lds, st );
	}
	else
	{
		return( 1 );
	}

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

} // namespace avirhip
