// dn.hip -- integer-ratio downsizing passes (cfg4 of BASELINE.json: 3x; also 2x)
// for the two-pass tiled path (path 2), RGBA float.
//
// What the plan looks like (build mode 1, k = 2 or 3), per axis:
//   T[j]  = 0 + sum_{t<NT} c[t] * S[clamp(K*j + c0 + t)]    (doResize,
//           avir.h:3884-4096: ONE bank phase, so every output has the same NT
//           coefficients, and the source offset advances by exactly K)
//   O[j]  = f0*T[j] + sum_{i=1..3} f_i*(T[clamp(j+i)] + T[clamp(j-i)])
//           (correction filter, doFilter avir.h:3748-3866)
// with (K, NT) = (2, 24) or (3, 38) for the default parameters.
//
// Because the coefficients are uniform and the offsets regular, a thread can
// produce R consecutive outputs from ONE pass over NT + (R-1)*K inputs with
// every index known at compile time: each input value is read once (from LDS
// in the H pass, from L2 in the V pass) and feeds up to ceil(NT/K) multiply-
// adds in registers, instead of one LDS read per multiply-add as in k_tile.
// Each accumulator still receives its taps in ascending order starting from
// +0.0f, mul then add, so the result is bit-identical to the reference.
//
//   k_dnh   one wave per source row segment: row -> LDS by LDS-DMA (clamped
//           columns), R = 3 resize outputs per lane (lane stride R*K pixels is
//           bank-conflict free for K = 3), T -> LDS, FIR7 along x, coalesced
//           store into FltBuf. Waves are autonomous: no barrier in the loop.
//   k_dnv   one lane per pixel column, 8 final rows per lane: the 14 resize
//           rows they need accumulate in registers from FltBuf rows streamed
//           as coalesced 1-KiB wave loads (L2/MALL resident: it is the small
//           image), then the FIR7 along y runs over the registers; no LDS.

#include "gpass_dev.h"
#include "dn_dev.h"
#include <algorithm>
#include <string.h>
#include <type_traits>
#include <utility>

namespace avirhip {

typedef float f4 __attribute__(( ext_vector_type( 4 )));
typedef unsigned u4 __attribute__(( ext_vector_type( 4 )));
typedef __attribute__(( address_space( 3 ))) void* ldsptr;

struct DnHParams
{
	DnAxisP A;
	const void* src; long src_ss;  // elements per source row
	int src_type, src_ch;          // AVIRHIP_F32 / 4: float RGBA rows (LDS-DMA);
	                               // AVIRHIP_U8 / _U16, 1-4 channels: the
	                               // caller's image, converted by the loader
	float* dst; long dst_ss;       // FltBuf, floats per row
	int row_lo, row_hi;            // source rows to process
	long src_elems;                // integer images: elements up to the end of
	                               // row row_hi - 1 (the loader's bound)
};

struct DnVParams
{
	DnAxisP A;
	const float* src; long src_ss; // FltBuf
	float* dst; long dst_ss;       // destination band, row dst_row0 at dst
	int width;                     // pixels per row
	int g_lo, g_hi;                // output groups (of RO rows) to process
	int row_lo, row_hi, dst_row0;  // rows to store
	GPOut out;                     // on != 0: integer output stage in the store
};

// input i feeds tap i - K*r of output r
template< int K, int NT, int R, int I >
__device__ __forceinline__ void dn_feed( const float ( &c )[ NT ],
	f4 ( &acc )[ R ], const f4 v )
{
	dn_static_for< R >( [&]( auto rc )
	{
		constexpr int r = decltype( rc )::value;
		constexpr int t = I - K * r;

		if constexpr( t >= 0 && t < NT )
		{
			// coefficient from an SGPR, plain v_mul_f32 / v_add_f32 (the packed
			// form would need {c, c} register pairs: hipcc builds them with
			// v_mov and runs out of VGPRs; the scalar form has the same flop
			// rate and full hardware interlocks)
			// (the adds are in the statement too: left to hipcc they sink to the
			// end of the kernel and every product is spilled meanwhile)
			float p0, p1, p2, p3;
			float a0 = acc[ r ].x, a1 = acc[ r ].y, a2 = acc[ r ].z,
				a3 = acc[ r ].w;
			asm volatile( "v_mul_f32 %4, %8, %9\n\t"
				"v_mul_f32 %5, %8, %10\n\t"
				"v_mul_f32 %6, %8, %11\n\t"
				"v_mul_f32 %7, %8, %12\n\t"
				"v_add_f32 %0, %0, %4\n\t"
				"v_add_f32 %1, %1, %5\n\t"
				"v_add_f32 %2, %2, %6\n\t"
				"v_add_f32 %3, %3, %7"
				: "+v"( a0 ), "+v"( a1 ), "+v"( a2 ), "+v"( a3 ),
				"=&v"( p0 ), "=&v"( p1 ), "=&v"( p2 ), "=&v"( p3 )
				: "s"( c[ t ]), "v"( v.x ), "v"( v.y ), "v"( v.z ), "v"( v.w ));
			acc[ r ].x = a0; acc[ r ].y = a1; acc[ r ].z = a2; acc[ r ].w = a3;
		}
	} );
}

// R outputs from one pass over the inputs; `ld( i )` yields input i of the
// first output (input i - K*r of output r).
template< int K, int NT, int R, class LD >
__device__ __forceinline__ void dn_accumulate( const float ( &c )[ NT ],
	f4 ( &acc )[ R ], LD&& ld )
{
#pragma unroll
	for( int r = 0; r < R; r++ )
	{
		acc[ r ] = (f4) 0.0f;
	}

	dn_static_for< NT + ( R - 1 ) * K >( [&]( auto ic )
	{
		constexpr int i = decltype( ic )::value;
		dn_feed< K, NT, R, i >( c, acc, ld( i ));
	} );
}

// ---------------------------------------------------------------------
// H pass
// ---------------------------------------------------------------------

// one row segment of an integer image -> float RGBA pixels in LDS (the cast
// and the channel padding of packScanline, avir.h:2777-2840); all loads of
// the segment are issued before the first conversion. A pixel is ONE
// unaligned load of four elements, shifted down where it had to start early
// (the image's last pixel) -- three or four element loads per pixel made the
// integer form of this pass 20 us slower than the float one.
template< typename Tin, int NQ >
__device__ __forceinline__ void dn_load_raw( const Tin* base, const long row_e,
	const long total_e, const int ch, const int in0, const int in_len,
	const int nin, const int lane, f4* dst )
{
	unsigned long long v[ NQ ];
#pragma unroll
	for( int q = 0; q < NQ; q++ )
	{
		const long e = row_e + (long) dn_clampi( in0 + q * 64 + lane,
			in_len - 1 ) * ch;
		const long ea = ( e < total_e - 4 ? e : total_e - 4 );
		const int sh = (int) ( e - ea );

		if( sizeof( Tin ) == 1 )
		{
			typedef unsigned int u32u __attribute__(( aligned( 1 )));
			v[ q ] = *(const u32u*) ( base + ea ) >> ( 8 * sh );
		}
		else
		{
			typedef unsigned long long u64u __attribute__(( aligned( 2 )));
			v[ q ] = *(const u64u*) ( base + ea ) >> ( 16 * sh );
		}
	}
#pragma unroll
	for( int q = 0; q < NQ; q++ )
	{
		if( q * 64 + lane < nin )
		{
			float w[ 4 ];
#pragma unroll
			for( int c = 0; c < 4; c++ )
			{
				const unsigned int u = (unsigned int) ( sizeof( Tin ) == 1 ?
					( v[ q ] >> ( 8 * c )) & 0xffull :
					( v[ q ] >> ( 16 * c )) & 0xffffull );

				w[ c ] = ( c < ch ? (float) u : 0.0f );
			}

			f4 o; o.x = w[ 0 ]; o.y = w[ 1 ]; o.z = w[ 2 ]; o.w = w[ 3 ];
			dst[ q * 64 + lane ] = o;
		}
	}
}

template< int K, int NT, int R >
__global__ void __launch_bounds__( 256 ) k_dnh( const DnHParams P )
{
	constexpr int NT1 = 64 * R;                // resize outputs per wave
	constexpr int NIN = K * ( NT1 - 1 ) + NT;  // input pixels per wave
	constexpr int NDMA = ( NIN + 63 ) / 64;
	constexpr int SEG = NT1 - 6;               // FIR outputs per wave

	__shared__ __attribute__(( aligned( 16 ))) f4 sS[ 4 ][ NDMA * 64 ];
	__shared__ __attribute__(( aligned( 16 ))) f4 sT[ 4 ][ NT1 ];
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane( tid >> 6 );

	// the (uniform) coefficients: scalar loads, SGPR operands of the multiplies
	float c[ NT ];
#pragma unroll
	for( int t = 0; t < NT; t++ )
	{
		c[ t ] = P.A.coef[ t ];
	}

	const float f0 = P.A.coef[ NT ], f1 = P.A.coef[ NT + 1 ],
		f2 = P.A.coef[ NT + 2 ], f3 = P.A.coef[ NT + 3 ];

	const int y = P.row_lo + blockIdx.y * 4 + wave;

	if( y >= P.row_hi )
	{
		return;
	}

	const int X0 = blockIdx.x * SEG - 3;  // first resize output of this wave
	const int in0 = K * X0 + P.A.c0;      // its first input pixel

	// ---- source row segment -> LDS (clamped columns): lane-linear DMA for
	// float RGBA rows; integer rows are converted (and padded to 4 channels)
	// here, so no float copy of the source image is ever made
	if( P.src_type == AVIRHIP_F32 )
	{
		const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
			(void*) ( (const float*) P.src + (long) y * P.src_ss ), 0,
			0x7fffffff, 0x00020000 );

#pragma unroll
		for( int q = 0; q < NDMA; q++ )
		{
			const int i = q * 64 + lane;

			if(( q + 1 ) * 64 <= NIN || i < NIN )
			{
				const int gx = dn_clampi( in0 + i, P.A.in_len - 1 );
				__builtin_amdgcn_raw_ptr_buffer_load_lds( rsrc,
					(ldsptr) ( &sS[ wave ][ q * 64 ]), 16, gx * 16, 0, 0, 0 );
			}
		}
	}
	else
	if( P.src_type == AVIRHIP_U8 )
	{
		dn_load_raw< uint8_t, NDMA >( (const uint8_t*) P.src, (long) y *
			P.src_ss, P.src_elems, P.src_ch, in0, P.A.in_len, NIN, lane,
			sS[ wave ]);
	}
	else
	{
		dn_load_raw< uint16_t, NDMA >( (const uint16_t*) P.src, (long) y *
			P.src_ss, P.src_elems, P.src_ch, in0, P.A.in_len, NIN, lane,
			sS[ wave ]);
	}

	AVIRHIP_WAIT_VM( 0 );

	// ---- resize: R consecutive outputs per lane
	{
		const f4* base = &sS[ wave ][ K * R * lane ];
		f4 acc[ R ];
		dn_accumulate< K, NT, R >( c, acc,
			[&]( const int i ) { return( base[ i ]); } );

#pragma unroll
		for( int r = 0; r < R; r++ )
		{
			sT[ wave ][ R * lane + r ] = acc[ r ];
		}
	}

	// ---- correction FIR along x, coalesced stores
	const int olast = P.A.out_len - 1;
	float* drow = P.dst + (long) y * P.dst_ss;

#pragma unroll
	for( int q = 0; q < R; q++ )
	{
		const int n = q * 64 + lane;
		const int x = X0 + n;

		if( n >= 3 && n < NT1 - 3 && x <= olast )
		{
			const f4* t = &sT[ wave ][ -X0 ];
			f4 a = f0 * t[ x ];
			a = a + f1 * ( t[ dn_clampi( x + 1, olast )] +
				t[ dn_clampi( x - 1, olast )]);
			a = a + f2 * ( t[ dn_clampi( x + 2, olast )] +
				t[ dn_clampi( x - 2, olast )]);
			a = a + f3 * ( t[ dn_clampi( x + 3, olast )] +
				t[ dn_clampi( x - 3, olast )]);
			*(f4*) ( drow + (long) x * 4 ) = a;
		}
	}
}

// ---------------------------------------------------------------------
// V pass
// ---------------------------------------------------------------------

// RO final rows per lane: RO + 6 resize rows T[j0 .. j0+RO+5], j0 = g*RO - 3,
// accumulate in registers from one pass over NT + (RO+5)*K source rows; the
// correction filter then runs over them in registers too. T rows outside the
// array are copies of the edge rows (the reference replicates the resize
// OUTPUT there, avir.h:3227-3239), not resizes of replicated input.
template< int K, int NT, int RO >
__global__ void __launch_bounds__( 256 )
__attribute__(( amdgpu_waves_per_eu( 4, 4 ))) k_dnv( const DnVParams P )
{
	constexpr int R = RO + 6;
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane( tid >> 6 );

	float c[ NT ];
#pragma unroll
	for( int t = 0; t < NT; t++ )
	{
		c[ t ] = P.A.coef[ t ];
	}

	const float f0 = P.A.coef[ NT ], f1 = P.A.coef[ NT + 1 ],
		f2 = P.A.coef[ NT + 2 ], f3 = P.A.coef[ NT + 3 ];

	const int g = P.g_lo + blockIdx.y * 4 + wave; // group of RO output rows
	const int x = blockIdx.x * 64 + lane;

	if( g >= P.g_hi )
	{
		return;
	}

	const int j0 = g * RO - 3;      // first resize row held
	const int r0 = K * j0 + P.A.c0; // its first input row
	const int row_b = (int) P.src_ss * 4;
	const int xb = dn_clampi( x, P.width - 1 ) * 16;

	// rows are addressed relative to the first one needed so that the scalar
	// offsets stay small for any image size
	const int rbase = dn_clampi( r0, P.A.in_len - 1 );
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
		(void*) ( P.src + (long) rbase * P.src_ss ), 0, 0x7fffffff,
		0x00020000 );

	auto ld = [&]( const int i )
	{
		const int row = dn_clampi( r0 + i, P.A.in_len - 1 ) - rbase;
		return( __builtin_bit_cast( f4, __builtin_amdgcn_raw_buffer_load_b128(
			rsrc, xb, row * row_b, 0 )));
	};

	// explicit two-stage register pipeline: batch b+1 is in flight while
	// batch b is consumed (left alone, hipcc hoists all the loads and spills)
	constexpr int NI = NT + ( R - 1 ) * K;
	constexpr int BS = 6;
	constexpr int NB = ( NI + BS - 1 ) / BS;
	f4 buf[ 2 ][ BS ];
	f4 acc[ R ];
#pragma unroll
	for( int r = 0; r < R; r++ )
	{
		acc[ r ] = (f4) 0.0f;
	}

	dn_static_for< BS >( [&]( auto jc )
	{
		constexpr int j = decltype( jc )::value;
		buf[ 0 ][ j ] = ld( j );
	} );

	dn_static_for< NB >( [&]( auto bc )
	{
		constexpr int b = decltype( bc )::value;

		if constexpr( b + 1 < NB )
		{
			dn_static_for< BS >( [&]( auto jc )
			{
				constexpr int j = decltype( jc )::value;

				if constexpr(( b + 1 ) * BS + j < NI )
				{
					buf[ ( b + 1 ) & 1 ][ j ] = ld(( b + 1 ) * BS + j );
				}
			} );
		}

		// the memory clobbers pin the loads of batch b+1 between the compute
		// of batch b-1 and of batch b (hipcc otherwise issues all the loads
		// up front and spills them)
		asm volatile( "" ::: "memory" );
		__builtin_amdgcn_sched_barrier( 0 );
		dn_static_for< BS >( [&]( auto jc )
		{
			constexpr int j = decltype( jc )::value;
			constexpr int i = b * BS + j;

			if constexpr( i < NI )
			{
				dn_feed< K, NT, R, i >( c, acc, buf[ b & 1 ][ j ]);
				// keep the products of one input together (the scheduler
				// otherwise hoists every multiply of the batch)
				__builtin_amdgcn_sched_barrier( 0 );
			}
		} );
		asm volatile( "" ::: "memory" );
	} );

	// resize rows outside [0, out_len) are copies of the edge rows
	const int olast = P.A.out_len - 1;

	dn_static_for< R - 1 >( [&]( auto rc )
	{
		constexpr int r = R - 2 - decltype( rc )::value; // R-2 .. 0

		if( j0 + r < 0 )
		{
			acc[ r ] = acc[ r + 1 ];
		}
	} );

	dn_static_for< R - 1 >( [&]( auto rc )
	{
		constexpr int r = 1 + decltype( rc )::value; // 1 .. R-1

		if( j0 + r > olast )
		{
			acc[ r ] = acc[ r - 1 ];
		}
	} );

	// correction FIR along y (doFilter, avir.h:3748-3866) and store
	if( x < P.width )
	{
		dn_static_for< RO >( [&]( auto qc )
		{
			constexpr int q = decltype( qc )::value;
			const int y = j0 + 3 + q;

			if( y >= P.row_lo && y < P.row_hi )
			{
				f4 a = f0 * acc[ q + 3 ];
				a = a + f1 * ( acc[ q + 4 ] + acc[ q + 2 ]);
				a = a + f2 * ( acc[ q + 5 ] + acc[ q + 1 ]);
				a = a + f3 * ( acc[ q + 6 ] + acc[ q ]);
				if( P.out.on != 0 )
				{
					gp_store_int( P.out, y - P.dst_row0, x, a );
				}
				else
				{
					*(f4*) ( P.dst + (long) ( y - P.dst_row0 ) * P.dst_ss +
						(long) x * 4 ) = a;
				}
			}
		} );
	}
}

// ---------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------

static bool dn_supported( int K, int NT )
{
	return(( K == 2 && NT == 24 ) || ( K == 3 && NT == 38 ));
}

// Recognises "regular resize + 7-tap correction filter" and uploads its taps.
static int dn_match( const LAxis& ax, DnData* D, DnAxis& A, bool& ok )
{
	ok = false;

	if( ax.ops.size() != 2 )
	{
		return( AVIRHIP_OK );
	}

	const LOp& g = ax.ops[ 0 ];
	const LOp& f = ax.ops[ 1 ];

	if( g.type != OP_GATHER || g.view != VIEW_CLAMP || g.out_len < 8 ||
		f.type != OP_FIR || f.view != VIEW_CLAMP || f.rf != 1 || f.e != 0 ||
		f.lat != 3 || f.in_len != g.out_len || f.out_len != g.out_len ||
		(int) f.h_flt.size() < 4 )
	{
		return( AVIRHIP_OK );
	}

	const int K = g.h_start[ 1 ] - g.h_start[ 0 ];
	const int NT = g.maxtaps;

	if( !dn_supported( K, NT ))
	{
		return( AVIRHIP_OK );
	}

	for( int j = 0; j < g.out_len; j++ )
	{
		if( g.h_ntaps[ j ] != NT ||
			g.h_start[ j ] != g.h_start[ 0 ] + K * j ||
			memcmp( &g.h_coef[ (size_t) j * NT ], &g.h_coef[ 0 ],
			NT * sizeof( float )) != 0 )
		{
			return( AVIRHIP_OK );
		}
	}

	// [NT resize taps][4 FIR taps][NT + 2K tap pairs V2 of dnf.hip's column
	// waves: V2[i] = { i < NT ? c[i] : 0, i >= 2K ? c[i - 2K] : 0 }]
	float hc[ DN_MAXNT + 4 + 2 * ( DN_MAXNT + 6 )];
	memset( hc, 0, sizeof( hc ));
	memcpy( hc, &g.h_coef[ 0 ], NT * sizeof( float ));
	memcpy( hc + NT, &f.h_flt[ 0 ], 4 * sizeof( float ));

	for( int i = 0; i < NT + 2 * K; i++ )
	{
		hc[ NT + 4 + 2 * i ] = ( i < NT ? g.h_coef[ i ] : 0.0f );
		hc[ NT + 4 + 2 * i + 1 ] = ( i >= 2 * K ? g.h_coef[ i - 2 * K ] : 0.0f );
	}

	void* q = nullptr;
	AVIRHIP_HIPCHECK( hipMalloc( &q, sizeof( hc )));
	D -> allocs.push_back( q );
	AVIRHIP_HIPCHECK( hipMemcpy( q, hc, sizeof( hc ), hipMemcpyHostToDevice ));
	A.K = K; A.NT = NT; A.c0 = g.h_start[ 0 ];
	A.in_len = g.in_len; A.out_len = g.out_len;
	A.d_coef = (float*) q;
	ok = true;
	return( AVIRHIP_OK );
}

int dn_prepare( avirhip_plan* p, void** out )
{
	*out = nullptr;

	if( p -> is_lancir || p -> ch != 4 )
	{
		return( AVIRHIP_OK );
	}

	DnData* D = new DnData();
	D -> hok = D -> vok = false;
	int rc = dn_match( p -> h, D, D -> h, D -> hok );

	if( rc == 0 )
	{
		rc = dn_match( p -> v, D, D -> v, D -> vok );
	}

	if( rc != 0 || ( !D -> hok && !D -> vok ))
	{
		dn_release( D );
		return( rc );
	}

	hipDeviceProp_t prop;
	D -> ncu = ( hipGetDeviceProperties( &prop, p -> device ) == hipSuccess ?
		prop.multiProcessorCount : 256 );
	*out = D;
	return( AVIRHIP_OK );
}

void dn_release( void* d )
{
	DnData* D = (DnData*) d;

	if( D == nullptr )
	{
		return;
	}

	for( size_t i = 0; i < D -> allocs.size(); i++ )
	{
		(void) hipFree( D -> allocs[ i ]);
	}

	delete D;
}

bool dn_has_h( const void* d ) { return( d != nullptr && ((const DnData*) d ) -> hok ); }
bool dn_has_v( const void* d ) { return( d != nullptr && ((const DnData*) d ) -> vok ); }

static DnAxisP axis_params( const DnAxis& A )
{
	DnAxisP P;
	P.K = A.K; P.NT = A.NT; P.c0 = A.c0;
	P.in_len = A.in_len; P.out_len = A.out_len; P.coef = A.d_coef;
	return( P );
}

// source rows [a, b) -> FltBuf rows [a, b)
int dn_run_h( void* d, const void* src, int src_type, int src_ch, long src_ss,
	float* flt, long flt_ss, int a, int b, hipStream_t st )
{
	const DnData* D = (const DnData*) d;

	if( b <= a )
	{
		return( AVIRHIP_OK );
	}

	DnHParams P;
	P.A = axis_params( D -> h );
	P.src = src; P.src_ss = src_ss; P.dst = flt; P.dst_ss = flt_ss;
	P.src_type = src_type; P.src_ch = src_ch;
	P.row_lo = a; P.row_hi = b;
	P.src_elems = (long) ( b - 1 ) * src_ss + (long) D -> h.in_len * src_ch;

	if( src_type != AVIRHIP_F32 && P.src_elems < 4 )
	{
		return( 1 );
	}

	const int R = 3;
	const int seg = 64 * R - 6;
	const dim3 grid(( D -> h.out_len + seg - 1 ) / seg, ( b - a + 3 ) / 4 );

	if( D -> h.K == 3 )
	{
		hipLaunchKernelGGL(( k_dnh< 3, 38, 3 > ), grid, dim3( 256 ), 0, st, P );
	}
	else
	{
		hipLaunchKernelGGL(( k_dnh< 2, 24, 3 > ), grid, dim3( 256 ), 0, st, P );
	}

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

// FltBuf [in_len][width] -> destination rows [row0, row1)
int dn_run_v( void* d, const float* flt, long flt_ss, int width, float* dst,
	int row0, int row1, hipStream_t st, const GPOut* out )
{
	DnData* D = (DnData*) d;

	if( row1 <= row0 )
	{
		return( AVIRHIP_OK );
	}

	const int RO = 8;
	DnVParams P;
	P.A = axis_params( D -> v );
	P.src = flt; P.src_ss = flt_ss;
	P.dst = dst; P.dst_ss = (long) width * 4;
	P.width = width;
	P.g_lo = row0 / RO;
	P.g_hi = ( row1 - 1 ) / RO + 1;
	P.row_lo = row0; P.row_hi = row1; P.dst_row0 = row0;
	memset( &P.out, 0, sizeof( P.out ));

	if( out != nullptr )
	{
		P.out = *out;
	}

	const dim3 grid(( width + 63 ) / 64, ( P.g_hi - P.g_lo + 3 ) / 4 );

	if( D -> v.K == 3 )
	{
		hipLaunchKernelGGL(( k_dnv< 3, 38, 8 > ), grid, dim3( 256 ), 0, st, P );
	}
	else
	{
		hipLaunchKernelGGL(( k_dnv< 2, 24, 8 > ), grid, dim3( 256 ), 0, st, P );
	}

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

} // namespace avirhip
