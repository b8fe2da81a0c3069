// generic.hip -- the "always correct" executor: one kernel launch per lowered
// op, intermediates in HBM. It runs every plan the reference planner can
// produce (all build modes, any ratio, 1-4 channels, every I/O type) and is the
// in-library cross-check for the fused fast paths (fused.hip). Written for
// gfx950: 64-wide wavefronts run along the image's x axis in both passes so
// every global access of a wave is one contiguous segment; no transposition
// (the reference's convertVtoH, avir.h:3085-3136) is ever needed.
//
// Arithmetic contract (compiled -ffp-contract=off): separate v_mul_f32 /
// v_add_f32 in the reference's order, sums started from +0.0f -- the result
// is bit-identical to the reference's scalar float path.

#include "plan.h"
#include <atomic>
#include <string.h>
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>

namespace avirhip {

struct OpArgs
{
	const float* in; long in_ss, in_is; int in_prefix;
	float* out; long out_ss, out_is; int out_prefix;
	int view, in_len, zs_mmax;
	int ch;
	int scan0, nscan, idx0, nidx;
	int rf, lat, e; const float* flt;
	const int* start; const int* ntaps; const float* coef; int maxtaps;
	int flen, up_inprefix, up_R, sdc_len, pdc_len, pdc_d0;
	const float* sdc; const float* pdc;
};

__device__ __forceinline__ float ldv( const OpArgs& a, const long so, int i,
	const int c )
{
	if( a.view == VIEW_RAW )
	{
		return( a.in[ so + (long) ( i + a.in_prefix ) * a.in_is + c ]);
	}

	if( a.view == VIEW_ZS && i > a.zs_mmax )
	{
		return( 0.0f );
	}

	i = ( i < 0 ? 0 : ( i >= a.in_len ? a.in_len - 1 : i ));
	return( a.in[ so + (long) i * a.in_is + c ]);
}

// Thread (x, y): XIDX ? (idx, scan) : (scan, idx). x is always along image x.
template< bool XIDX >
__device__ __forceinline__ bool locate( const OpArgs& a, int& scan, int& idx )
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	const int y = blockIdx.y * blockDim.y + threadIdx.y;
	const int si = ( XIDX ? y : x );
	const int ii = ( XIDX ? x : y );

	if( si >= a.nscan || ii >= a.nidx )
	{
		return( false );
	}

	scan = a.scan0 + si;
	idx = a.idx0 + ii;
	return( true );
}

// doFilter, avir.h:3748-3866.
template< bool XIDX >
__global__ void __launch_bounds__( 256 ) k_fir( const OpArgs a )
{
	int scan, n;

	if( !locate< XIDX >( a, scan, n ))
	{
		return;
	}

	const long so = (long) scan * a.in_ss;
	const long oo = (long) scan * a.out_ss + (long) n * a.out_is;
	const int cp = a.rf * ( n - a.e );

	for( int c = 0; c < a.ch; c++ )
	{
		float s = a.flt[ 0 ] * ldv( a, so, cp, c );

		for( int i = 1; i <= a.lat; i++ )
		{
			s += a.flt[ i ] * ( ldv( a, so, cp + i, c ) +
				ldv( a, so, cp - i, c ));
		}

		a.out[ oo + c ] = s;
	}
}

// doResize / doResize2, avir.h:3884-4328 (coefficients pre-expanded).
template< bool XIDX >
__global__ void __launch_bounds__( 256 ) k_gather( const OpArgs a )
{
	int scan, j;

	if( !locate< XIDX >( a, scan, j ))
	{
		return;
	}

	const long so = (long) scan * a.in_ss;
	const long oo = (long) scan * a.out_ss + (long) j * a.out_is;
	const int st = a.start[ j ];
	const int nt = a.ntaps[ j ];
	const float* cf = a.coef + (long) j * a.maxtaps;

	for( int c = 0; c < a.ch; c++ )
	{
		float sum = 0.0f;

		for( int t = 0; t < nt; t++ )
		{
			sum += cf[ t ] * ldv( a, so, st + t, c );
		}

		a.out[ oo + c ] = sum;
	}
}

// doUpsample with filtering, avir.h:3404-3733, as a gather over the
// destination slot d = idx + OutPrefix. Contributions arrive in ascending
// input order, then the SuffixDC tail, then the PrefixDC tail -- the order in
// which the reference's scatter-adds touch the slot.
template< bool XIDX >
__global__ void __launch_bounds__( 256 ) k_upf( const OpArgs a )
{
	int scan, d;

	if( !locate< XIDX >( a, scan, d )) // idx0 = 0: d runs over out_total
	{
		return;
	}

	const long so = (long) scan * a.in_ss;
	const long oo = (long) scan * a.out_ss + (long) d * a.out_is;
	int rlo = d - a.flen + 1;
	rlo = ( rlo <= 0 ? 0 : ( rlo + a.rf - 1 ) / a.rf );
	int rhi = d / a.rf;

	if( rhi > a.up_R - 1 )
	{
		rhi = a.up_R - 1;
	}

	const int ts = d - a.up_R * a.rf;
	const int tp = d - a.pdc_d0;

	for( int c = 0; c < a.ch; c++ )
	{
		float s = 0.0f;

		for( int r = rlo; r <= rhi; r++ )
		{
			int m = r - a.up_inprefix;
			m = ( m < 0 ? 0 : ( m >= a.in_len ? a.in_len - 1 : m ));
			s += a.flt[ d - r * a.rf ] * a.in[ so + (long) m * a.in_is + c ];
		}

		if( ts >= 0 && ts < a.sdc_len )
		{
			s += a.in[ so + (long) ( a.in_len - 1 ) * a.in_is + c ] *
				a.sdc[ ts ];
		}

		if( tp >= 0 && tp < a.pdc_len )
		{
			s += a.in[ so + c ] * a.pdc[ tp ];
		}

		a.out[ oo + c ] = s;
	}
}

int launch_op( const LOp& op, int ch, bool x_is_idx, const Surf& in,
	const Surf& out, int scan0, int scan1, int idx0, int idx1,
	hipStream_t st )
{
	OpArgs a;
	a.in = in.base; a.in_ss = in.scan_stride; a.in_is = in.idx_stride;
	a.in_prefix = in.prefix;
	a.out = out.base + (long) out.prefix * out.idx_stride;
	a.out_ss = out.scan_stride; a.out_is = out.idx_stride;
	a.out_prefix = out.prefix;
	a.view = op.view; a.in_len = op.in_len; a.zs_mmax = op.zs_mmax;
	a.ch = ch;
	a.scan0 = scan0; a.nscan = scan1 - scan0;
	a.idx0 = idx0; a.nidx = idx1 - idx0;
	a.rf = op.rf; a.lat = op.lat; a.e = op.e; a.flt = op.d_flt;
	a.start = op.d_start; a.ntaps = op.d_ntaps; a.coef = op.d_coef;
	a.maxtaps = op.maxtaps;
	a.flen = op.flen; a.up_inprefix = op.up_inprefix; a.up_R = op.up_R;
	a.sdc_len = op.sdc_len; a.pdc_len = op.pdc_len; a.pdc_d0 = op.pdc_d0;
	a.sdc = op.d_sdc; a.pdc = op.d_pdc;

	if( op.type == OP_UPF )
	{
		// Materialise the whole buffer incl. prefix/suffix: d = 0..out_total.
		a.out = out.base;
		a.idx0 = 0;
		a.nidx = op.out_total;
	}

	if( a.nscan <= 0 || a.nidx <= 0 )
	{
		return( AVIRHIP_OK );
	}

	const dim3 blk( 64, 4 );
	const int nx = ( x_is_idx ? a.nidx : a.nscan );
	const int ny = ( x_is_idx ? a.nscan : a.nidx );
	const dim3 grd(( nx + 63 ) / 64, ( ny + 3 ) / 4 );

#define LAUNCH( K ) if( x_is_idx ) hipLaunchKernelGGL( K< true >, grd, blk, 0, \
		st, a ); else hipLaunchKernelGGL( K< false >, grd, blk, 0, st, a )

	if( op.type == OP_FIR ) { LAUNCH( k_fir ); }
	else if( op.type == OP_GATHER ) { LAUNCH( k_gather ); }
	else { LAUNCH( k_upf ); }

#undef LAUNCH

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

// packScanline, avir.h:2777-2840 (no-gamma branch): (float) cast per element.
// Pixels of `ch` channels are written with `ech` >= ch channels, the extra
// ones zero (the reference pads its SIMD lanes the same way, avir.h:2932-2971).
template< typename Tin >
__global__ void __launch_bounds__( 256 ) k_pack( const Tin* src, float* dst,
	int row_elems, int h, long src_stride, int ch, int ech )
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	const int y = blockIdx.y;

	if( x < row_elems && y < h )
	{
		const int px = x / ech;
		const int c = x - px * ech;

		dst[ (long) y * row_elems + x ] = ( c < ch ?
			(float) src[ (long) y * src_stride + px * ch + c ] : 0.0f );
	}
}

// ech == 4: one thread per pixel, one 16-byte store (the element-wise kernel
// above spends an integer division and a 4-byte store per float)
template< typename Tin, int CH >
__global__ void __launch_bounds__( 256 ) k_pack_px( const Tin* src,
	float* dst, int w, int h, long src_stride )
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	const int y = blockIdx.y;

	if( x >= w || y >= h )
	{
		return;
	}

	float v[ 4 ] = { 0.0f, 0.0f, 0.0f, 0.0f };

	if( sizeof( Tin ) <= 2 && std::is_integral< Tin > :: value )
	{
		// integer pixels: ONE unaligned load of four elements, shifted down
		// where it had to start early (the image's last pixel)
		const long total = (long) ( h - 1 ) * src_stride + (long) w * CH;
		const long e = (long) y * src_stride + (long) x * CH;
		const long ea = ( total >= 4 && e > total - 4 ? total - 4 : e );
		const int sh = (int) ( e - ea );
		unsigned long long q;

		if( total < 4 )
		{
			q = 0;

			for( int c = 0; c < CH; c++ )
			{
				q |= (unsigned long long) src[ e + c ] << ( 8 * sizeof( Tin ) * c );
			}
		}
		else
		if( sizeof( Tin ) == 1 )
		{
			typedef unsigned int u32u __attribute__(( aligned( 1 )));
			q = *(const u32u*) ( src + ea ) >> ( 8 * sh );
		}
		else
		{
			typedef unsigned long long u64u __attribute__(( aligned( 2 )));
			q = *(const u64u*) ( src + ea ) >> ( 16 * sh );
		}
#pragma unroll
		for( int c = 0; c < CH; c++ )
		{
			v[ c ] = (float) (unsigned int) ( sizeof( Tin ) == 1 ?
				( q >> ( 8 * c )) & 0xffull : ( q >> ( 16 * c )) & 0xffffull );
		}
	}
	else
	{
		const Tin* s = src + (long) y * src_stride + (long) x * CH;
#pragma unroll
		for( int c = 0; c < CH; c++ )
		{
			v[ c ] = (float) s[ c ];
		}
	}

	typedef float f4v __attribute__(( ext_vector_type( 4 )));
	f4v o; o.x = v[ 0 ]; o.y = v[ 1 ]; o.z = v[ 2 ]; o.w = v[ 3 ];
	*(f4v*) ( dst + ( (long) y * w + x ) * 4 ) = o;
}

template< typename Tin >
static void launch_pack_px( const void* src, float* dst, int w, int h, int ch,
	long src_stride, hipStream_t st )
{
	const dim3 grd(( w + 255 ) / 256, h );
	const Tin* s = (const Tin*) src;

	switch( ch )
	{
		case 1: hipLaunchKernelGGL(( k_pack_px< Tin, 1 > ), grd, dim3( 256 ),
			0, st, s, dst, w, h, src_stride ); break;
		case 2: hipLaunchKernelGGL(( k_pack_px< Tin, 2 > ), grd, dim3( 256 ),
			0, st, s, dst, w, h, src_stride ); break;
		case 3: hipLaunchKernelGGL(( k_pack_px< Tin, 3 > ), grd, dim3( 256 ),
			0, st, s, dst, w, h, src_stride ); break;
		default: hipLaunchKernelGGL(( k_pack_px< Tin, 4 > ), grd, dim3( 256 ),
			0, st, s, dst, w, h, src_stride ); break;
	}
}

int launch_pack( const void* src, int in_type, float* dst, int w, int h,
	int ch, int ech, long src_stride, hipStream_t st )
{
	if( ech == 4 )
	{
		if( in_type == AVIRHIP_U8 )
			launch_pack_px< uint8_t >( src, dst, w, h, ch, src_stride, st );
		else if( in_type == AVIRHIP_U16 )
			launch_pack_px< uint16_t >( src, dst, w, h, ch, src_stride, st );
		else if( in_type == AVIRHIP_F64 )
			launch_pack_px< double >( src, dst, w, h, ch, src_stride, st );
		else if( in_type == AVIRHIP_U32 )
			launch_pack_px< uint32_t >( src, dst, w, h, ch, src_stride, st );
		else
			launch_pack_px< float >( src, dst, w, h, ch, src_stride, st );

		AVIRHIP_HIPCHECK( hipGetLastError() );
		return( AVIRHIP_OK );
	}


	const int re = w * ech;
	const dim3 grd(( re + 255 ) / 256, h );

	if( in_type == AVIRHIP_U8 )
		hipLaunchKernelGGL( k_pack< uint8_t >, grd, dim3( 256 ), 0, st,
			(const uint8_t*) src, dst, re, h, src_stride, ch, ech );
	else if( in_type == AVIRHIP_U16 )
		hipLaunchKernelGGL( k_pack< uint16_t >, grd, dim3( 256 ), 0, st,
			(const uint16_t*) src, dst, re, h, src_stride, ch, ech );
	else if( in_type == AVIRHIP_F64 )
		hipLaunchKernelGGL( k_pack< double >, grd, dim3( 256 ), 0, st,
			(const double*) src, dst, re, h, src_stride, ch, ech );
	else if( in_type == AVIRHIP_U32 )
		hipLaunchKernelGGL( k_pack< uint32_t >, grd, dim3( 256 ), 0, st,
			(const uint32_t*) src, dst, re, h, src_stride, ch, ech );
	else
		hipLaunchKernelGGL( k_pack< float >, grd, dim3( 256 ), 0, st,
			(const float*) src, dst, re, h, src_stride, ch, ech );

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

// ---- sRGB gamma (UseSRGBGamma). The reference approximates the 2.4 and 1/2.4
// powers with a rational polynomial / nested square roots evaluated in DOUBLE
// from a float argument (pow24_sRGB avir.h:162-175, pow24i_sRGB 187-198); IEEE
// double +, *, / and sqrt are correctly rounded on gfx950 as on the CPU, and
// -ffp-contract=off keeps the operation order, so the stages are bit-exact.

__host__ __device__ __forceinline__ double srgb_pow24( const double x )
{
	const double x2 = x * x;
	const double x3 = x2 * x;
	const double x4 = x2 * x2;

	return( 0.0985766365536824 + 0.839474952656502 * x2 +
		0.363287814061725 * x3 - 0.0125559718896615 /
		( 0.12758338921578 + 0.290283465468235 * x ) -
		0.231757513261358 * x - 0.0395365717969074 * x4 );
}

__host__ __device__ __forceinline__ double srgb_pow24i( const double x )
{
	const double sx = sqrt( x );
	const double ssx = sqrt( sx );
	const double sssx = sqrt( ssx );

	return( 0.000213364515060263 + 0.0149409239419218 * x +
		0.433973412731747 * sx + ssx * ( 0.659628181609715 * sssx -
		0.0380957908841466 - 0.0706476137208521 * sx ));
}

// convertSRGB2Lin< float, Tin >, avir.h:206-219 (float arithmetic around the
// double polynomial)
__device__ __forceinline__ float srgb_to_lin( const float s0, const float m )
{
	const float s = s0 * m;
	const float a = 0.055f;

	if( s <= 0.04045f )
	{
		return( s / 12.92f );
	}

	return( (float) srgb_pow24( (double) (( s + a ) / ( 1.0f + a ))));
}

// convertLin2SRGB< float >, avir.h:301-312
__host__ __device__ __forceinline__ float lin_to_srgb( const float s )
{
	const float a = 0.055f;

	if( s <= 0.0031308f )
	{
		return( 12.92f * s );
	}

	return(( 1.0f + a ) * (float) srgb_pow24i( (double) s ) - a );
}

// The uint8_t specialisation of convertSRGB2Lin (avir.h:229-291) is a table of
// 256 literals. They are the double-precision evaluation of the same
// approximation, written with 7 significant digits: regenerate them that way
// (tests/test_oracle.py checks all 256 against the reference).
void srgb_u8_table( float* tbl )
{
	for( int i = 0; i < 256; i++ )
	{
		const double s = (double) i * ( 1.0 / 255.0 );
		const double a = 0.055;
		const double v = ( s <= 0.04045 ? s / 12.92 :
			srgb_pow24(( s + a ) / ( 1.0 + a )));

		char buf[ 64 ];
		snprintf( buf, sizeof( buf ), "%.7g", v );
		tbl[ i ] = strtof( buf, nullptr );
	}
}

// packScanline with gamma, avir.h:2841-2930: colour channels are linearised,
// the alpha channel (4-channel pixels, AlphaIndex 0 or 3) is only scaled by
// InGammaMult.
template< typename Tin >
__global__ void __launch_bounds__( 256 ) k_pack_gamma( const Tin* src,
	float* dst, int row_elems, int h, long src_stride, int ch, int ech,
	int alpha_index, float gm, const float* tbl )
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	const int y = blockIdx.y;

	if( x >= row_elems || y >= h )
	{
		return;
	}

	const int px = x / ech;
	const int c = x - px * ech;
	float r = 0.0f;

	if( c < ch )
	{
		const Tin v = src[ (long) y * src_stride + px * ch + c ];

		if( c == alpha_index )
		{
			r = (float) v * gm;
		}
		else
		if( sizeof( Tin ) == 1 )
		{
			r = tbl[ (int) v ];
		}
		else
		{
			r = srgb_to_lin( (float) v, gm );
		}
	}

	dst[ (long) y * row_elems + x ] = r;
}

// ech == 4: one thread per pixel, one 16-byte store (the element-wise kernel
// above spends an integer division and a 4-byte store per float)
template< typename Tin, int CH >
__global__ void __launch_bounds__( 256 ) k_pack_gamma_px( const Tin* src,
	float* dst, int w, int h, long src_stride, int alpha_index, float gm,
	const float* tbl )
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	const int y = blockIdx.y;

	if( x >= w || y >= h )
	{
		return;
	}

	const Tin* s = src + (long) y * src_stride + (long) x * CH;
	float v[ 4 ] = { 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
	for( int c = 0; c < CH; c++ )
	{
		const Tin e = s[ c ];

		if( c == alpha_index )
		{
			v[ c ] = (float) e * gm;
		}
		else
		if( sizeof( Tin ) == 1 )
		{
			v[ c ] = tbl[ (int) e ];
		}
		else
		{
			v[ c ] = srgb_to_lin( (float) e, gm );
		}
	}

	typedef float f4v __attribute__(( ext_vector_type( 4 )));
	f4v o; o.x = v[ 0 ]; o.y = v[ 1 ]; o.z = v[ 2 ]; o.w = v[ 3 ];
	*(f4v*) ( dst + ( (long) y * w + x ) * 4 ) = o;
}

template< typename Tin >
static void launch_pack_gamma_px( const void* src, float* dst, int w, int h,
	int ch, long src_stride, int alpha_index, float gm, const float* tbl,
	hipStream_t st )
{
	const dim3 grd(( w + 255 ) / 256, h );
	const Tin* s = (const Tin*) src;

	switch( ch )
	{
		case 1: hipLaunchKernelGGL(( k_pack_gamma_px< Tin, 1 > ), grd,
			dim3( 256 ), 0, st, s, dst, w, h, src_stride, alpha_index, gm, tbl );
			break;
		case 2: hipLaunchKernelGGL(( k_pack_gamma_px< Tin, 2 > ), grd,
			dim3( 256 ), 0, st, s, dst, w, h, src_stride, alpha_index, gm, tbl );
			break;
		case 3: hipLaunchKernelGGL(( k_pack_gamma_px< Tin, 3 > ), grd,
			dim3( 256 ), 0, st, s, dst, w, h, src_stride, alpha_index, gm, tbl );
			break;
		default: hipLaunchKernelGGL(( k_pack_gamma_px< Tin, 4 > ), grd,
			dim3( 256 ), 0, st, s, dst, w, h, src_stride, alpha_index, gm, tbl );
			break;
	}
}

int launch_pack_gamma( const void* src, int in_type, float* dst, int w, int h,
	int ch, int ech, long src_stride, int alpha_index, const float* tbl,
	hipStream_t st )
{
	const int re = w * ech;
	const dim3 grd(( re + 255 ) / 256, h );
	// InGammaMult, avir.h:4744-4754
	const float gm = (float) ( in_type == AVIRHIP_U8 ? 1.0 / 255.0 :
		( in_type == AVIRHIP_U16 ? 1.0 / 65535.0 : 1.0 ));

	if( ech == 4 && in_type != AVIRHIP_F64 )
	{
		if( in_type == AVIRHIP_U8 )
			launch_pack_gamma_px< uint8_t >( src, dst, w, h, ch, src_stride,
				alpha_index, gm, tbl, st );
		else if( in_type == AVIRHIP_U16 )
			launch_pack_gamma_px< uint16_t >( src, dst, w, h, ch, src_stride,
				alpha_index, gm, tbl, st );
		else
			launch_pack_gamma_px< float >( src, dst, w, h, ch, src_stride,
				alpha_index, gm, tbl, st );

		AVIRHIP_HIPCHECK( hipGetLastError() );
		return( AVIRHIP_OK );
	}

	if( in_type == AVIRHIP_U8 )
		hipLaunchKernelGGL( k_pack_gamma< uint8_t >, grd, dim3( 256 ), 0, st,
			(const uint8_t*) src, dst, re, h, src_stride, ch, ech,
			alpha_index, gm, tbl );
	else if( in_type == AVIRHIP_U16 )
		hipLaunchKernelGGL( k_pack_gamma< uint16_t >, grd, dim3( 256 ), 0,
			st, (const uint16_t*) src, dst, re, h, src_stride, ch, ech,
			alpha_index, gm, tbl );
	else if( in_type == AVIRHIP_F64 )
		hipLaunchKernelGGL( k_pack_gamma< double >, grd, dim3( 256 ), 0, st,
			(const double*) src, dst, re, h, src_stride, ch, ech,
			alpha_index, gm, tbl );
	else
		hipLaunchKernelGGL( k_pack_gamma< float >, grd, dim3( 256 ), 0, st,
			(const float*) src, dst, re, h, src_stride, ch, ech,
			alpha_index, gm, tbl );

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

// avir::round, avir.h:130-135 (biased, via int cast).
__host__ __device__ __forceinline__ float avir_round( const float d )
{
	return( d < 0.0f ? -(float) avirhip_x86_cvtt( 0.5f - d ) :
		(float) avirhip_x86_cvtt( d + 0.5f ));
}

// round() of avir_float4_sse.h:303-313 (the ditherer of fpclass_float4):
// _mm_cvtps_epi32 under round-to-nearest-even, back to float; values the
// 32-bit conversion cannot hold become the "integer indefinite" -2^31.
__device__ __forceinline__ float avir_round_rne( const float d )
{
	const float r = rintf( d );
	return( fabsf( r ) < 2147483648.0f ? r : -2147483648.0f );
}

// CImageResizerDithererDefINL::dither (avir.h:4392-4419) + unpackScanline
// (avir.h:3155-3215): round, clamp, optional bit-depth truncation, C cast.
// With gamma, applySRGBGamma (avir.h:2982-3068) runs first: colour channels are
// de-linearised and scaled by OutGammaMult, the alpha channel only scaled.
template< typename Tout, bool IsInt >
__global__ void __launch_bounds__( 256 ) k_epilogue( const float* res,
	Tout* dst, long n, int use_tr, float tr_mul, float tr_muli, float pk_out,
	int gamma, int ch, int ech, int alpha_index, float ogm, int rne )
{
	const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;

	if( i >= n )
	{
		return;
	}

	// `res` holds ech >= ch channels per pixel (padding dropped here)
	const long px = i / ch;
	const int c = (int) ( i - px * ch );
	float v = res[ px * ech + c ];

	if( gamma )
	{
		if( c == alpha_index )
		{
			v = v * ogm;
		}
		else
		{
			v = lin_to_srgb( v ) * ogm;
		}
	}

	if( IsInt )
	{
		if( rne )
		{
			// fpclass_float4: nearest-even, and clamp() is min(max(v, 0), pk)
			// of SSE (avir_float4_sse.h:325-329) -- the same values
			v = ( use_tr ? avir_round_rne( v * tr_muli ) * tr_mul :
				avir_round_rne( v ));
		}
		else
		if( use_tr )
		{
			v = avir_round( v * tr_muli ) * tr_mul;
		}
		else
		{
			v = avir_round( v );
		}

		v = ( v < 0.0f ? 0.0f : ( v > pk_out ? pk_out : v ));
	}

	dst[ i ] = (Tout) v;
}

// ech == 4, no gamma: one thread per pixel, one 16-byte load; 4-channel uint8
// pixels leave as one 32-bit store
template< typename Tout, bool IsInt, int CH >
__global__ void __launch_bounds__( 256 ) k_epilogue_px( const float* res,
	Tout* dst, long npx, int use_tr, float tr_mul, float tr_muli,
	float pk_out )
{
	const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;

	if( i >= npx )
	{
		return;
	}

	typedef float f4v __attribute__(( ext_vector_type( 4 )));
	const f4v r = *(const f4v*) ( res + i * 4 );
	float v[ 4 ] = { r.x, r.y, r.z, r.w };
	Tout o[ 4 ];
#pragma unroll
	for( int c = 0; c < CH; c++ )
	{
		float t = v[ c ];

		if( IsInt )
		{
			t = ( use_tr ? avir_round( t * tr_muli ) * tr_mul :
				avir_round( t ));

			t = ( t < 0.0f ? 0.0f : ( t > pk_out ? pk_out : t ));
		}

		o[ c ] = (Tout) t;
	}

	if( CH == 4 && sizeof( Tout ) == 1 )
	{
		*(uint32_t*) ( dst + i * 4 ) = (uint32_t) o[ 0 ] |
			( (uint32_t) o[ 1 ] << 8 ) | ( (uint32_t) o[ 2 ] << 16 ) |
			( (uint32_t) o[ 3 ] << 24 );
	}
	else
	{
#pragma unroll
		for( int c = 0; c < CH; c++ )
		{
			dst[ i * CH + c ] = o[ c ];
		}
	}
}

template< typename Tout, bool IsInt >
static void launch_epilogue_px( const float* res, void* dst, long npx, int ch,
	int use_tr, float trm, float trmi, float pk, hipStream_t st )
{
	const dim3 grd( (unsigned) (( npx + 255 ) / 256 ));
	Tout* d = (Tout*) dst;

	switch( ch )
	{
		case 1: hipLaunchKernelGGL(( k_epilogue_px< Tout, IsInt, 1 > ), grd,
			dim3( 256 ), 0, st, res, d, npx, use_tr, trm, trmi, pk ); break;
		case 2: hipLaunchKernelGGL(( k_epilogue_px< Tout, IsInt, 2 > ), grd,
			dim3( 256 ), 0, st, res, d, npx, use_tr, trm, trmi, pk ); break;
		case 3: hipLaunchKernelGGL(( k_epilogue_px< Tout, IsInt, 3 > ), grd,
			dim3( 256 ), 0, st, res, d, npx, use_tr, trm, trmi, pk ); break;
		default: hipLaunchKernelGGL(( k_epilogue_px< Tout, IsInt, 4 > ), grd,
			dim3( 256 ), 0, st, res, d, npx, use_tr, trm, trmi, pk ); break;
	}
}

// The gamma output stage for uint8 as a step function. applySRGBGamma + the
// default ditherer + unpackScanline map a float v to a byte through operations
// that are each monotone non-decreasing in v (a double-precision root
// polynomial, float multiplies by positive constants, round, clamp, cast), so
// the whole stage is a monotone step function of v: the byte is the number of
// thresholds thr[ k ] <= v, thr[ k ] = the smallest float whose byte is >= k.
// The host finds the 255 thresholds by bisection over float bit patterns with
// the SAME expressions (correctly rounded IEEE double sqrt on both sides), so
// the table form is bit-exact -- eight comparisons instead of three
// double-precision square roots per channel.
static float gamma_stage( const float v0, const bool alpha, const float ogm,
	const int use_tr, const float trm, const float trmi, const float pk )
{
	float v = ( alpha ? v0 * ogm : lin_to_srgb( v0 ) * ogm );
	v = ( use_tr ? avir_round( v * trmi ) * trm : avir_round( v ));
	return( v < 0.0f ? 0.0f : ( v > pk ? pk : v ));
}

// thr[ 0 .. 255 ] colour channels, thr[ 256 .. 511 ] the alpha channel;
// entry 0 of each is unused (-inf), unreachable bytes hold +inf
bool gamma_u8_thresholds( const float ogm, const int use_tr, const float trm,
	const float trmi, const float pk, float* thr )
{
	for( int a = 0; a < 2; a++ )
	{
		float* const t = thr + a * 256;
		t[ 0 ] = -INFINITY;

		for( int k = 1; k < 256; k++ )
		{
			// non-negative floats order like their bit patterns
			unsigned int lo = 0, hi; // F( lo ) < k <= F( hi )
			if( gamma_stage( 0.0f, a == 1, ogm, use_tr, trm, trmi, pk ) >= k )
			{
				t[ k ] = 0.0f;
				continue;
			}

			// (the search stops at 16.0: far above anything that still changes
			// the byte, far below the values whose rounding leaves the int range
			// -- where the host's cast and the device's differ)
			if( !( gamma_stage( 16.0f, a == 1, ogm, use_tr, trm, trmi,
				pk ) >= k ))
			{
				t[ k ] = INFINITY;
				continue;
			}

			hi = 0x41800000u; // 16.0f

			while( hi - lo > 1 )
			{
				const unsigned int mid = lo + ( hi - lo ) / 2;
				union { unsigned int u; float f; } cv; cv.u = mid;
				const float fm = cv.f;

				if( gamma_stage( fm, a == 1, ogm, use_tr, trm, trmi, pk ) >= k )
				{
					hi = mid;
				}
				else
				{
					lo = mid;
				}
			}

			{ union { unsigned int u; float f; } cv; cv.u = hi; t[ k ] = cv.f; }
		}

		// the stage must be monotone where it was sampled
		for( int k = 2; k < 256; k++ )
		{
			if( t[ k ] < t[ k - 1 ])
			{
				return( false );
			}
		}
	}

	return( true );
}

template< int CH >
__global__ void __launch_bounds__( 256 ) k_epilogue_gamma_thr( const float* res,
	unsigned char* dst, long npx, int alpha_index, const float* thr,
	int use_tr, float tr_mul, float tr_muli, float pk_out, float ogm )
{
	__shared__ float sthr[ 512 ];
	sthr[ threadIdx.x ] = thr[ threadIdx.x ];
	sthr[ threadIdx.x + 256 ] = thr[ threadIdx.x + 256 ];
	__syncthreads();

	const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;

	if( i >= npx )
	{
		return;
	}

	typedef float f4v __attribute__(( ext_vector_type( 4 )));
	const f4v r = *(const f4v*) ( res + i * 4 );
	const float vv[ 4 ] = { r.x, r.y, r.z, r.w };
#pragma unroll
	for( int c = 0; c < CH; c++ )
	{
		const float v = vv[ c ];
		const float* const t = sthr + ( c == alpha_index ? 256 : 0 );
		int lo = 0;
#pragma unroll
		for( int s = 128; s >= 1; s >>= 1 )
		{
			lo += ( v >= t[ lo + s ] ? s : 0 );
		}

		// outside the range the thresholds were searched in (16.0: nothing an
		// image holds) the direct expressions decide -- +inf, for one, is NOT
		// the largest byte: the power approximation turns it into a NaN
		// (found by the all-floats sweep of tests/test_gpu_parity.py), and
		// avir::round leaves the int range at both ends: the x86 build turns
		// a large positive value into byte 0, a large NEGATIVE one into 255
		// (plan.h avirhip_x86_cvtt; tools/fuzz_values.py found the second)
		if( !( fabsf( v ) <= 16.0f ))
		{
			float d = ( c == alpha_index ? v * ogm : lin_to_srgb( v ) * ogm );
			d = ( use_tr ? avir_round( d * tr_muli ) * tr_mul : avir_round( d ));
			d = ( d < 0.0f ? 0.0f : ( d > pk_out ? pk_out : d ));
			lo = (int) (unsigned char) d;
		}

		dst[ i * CH + c ] = (unsigned char) lo;
	}
}

// ech == 4 WITH gamma, integer output: one thread per pixel (applySRGBGamma +
// the default ditherer + unpackScanline, as k_epilogue spells them out)
template< typename Tout, int CH >
__global__ void __launch_bounds__( 256 ) k_epilogue_gamma_px( const float* res,
	Tout* dst, long npx, int use_tr, float tr_mul, float tr_muli,
	float pk_out, int alpha_index, float ogm )
{
	const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;

	if( i >= npx )
	{
		return;
	}

	typedef float f4v __attribute__(( ext_vector_type( 4 )));
	const f4v r = *(const f4v*) ( res + i * 4 );
	const float vv[ 4 ] = { r.x, r.y, r.z, r.w };
#pragma unroll
	for( int c = 0; c < CH; c++ )
	{
		float v = vv[ c ];
		v = ( c == alpha_index ? v * ogm : lin_to_srgb( v ) * ogm );
		v = ( use_tr ? avir_round( v * tr_muli ) * tr_mul : avir_round( v ));
		v = ( v < 0.0f ? 0.0f : ( v > pk_out ? pk_out : v ));
		dst[ i * CH + c ] = (Tout) v;
	}
}

template< typename Tout >
static void launch_epilogue_gamma_px( const float* res, void* dst, long npx,
	int ch, int use_tr, float trm, float trmi, float pk, int alpha_index,
	float ogm, hipStream_t st )
{
	const dim3 grd( (unsigned) (( npx + 255 ) / 256 ));
	Tout* d = (Tout*) dst;

#define EG( CH ) hipLaunchKernelGGL(( k_epilogue_gamma_px< Tout, CH > ), grd, \
		dim3( 256 ), 0, st, res, d, npx, use_tr, trm, trmi, pk, alpha_index, ogm )

	switch( ch )
	{
		case 1: EG( 1 ); break;
		case 2: EG( 2 ); break;
		case 3: EG( 3 ); break;
		default: EG( 4 ); break;
	}

#undef EG
}

int launch_epilogue( const float* res, void* dst, int out_type, long n,
	double tr_mul, double pk_out, int gamma, int ch, int ech,
	int alpha_index, hipStream_t st, const float* gthr, bool rne )
{
	const dim3 grd( (unsigned) (( n + 255 ) / 256 ));
	const int use_tr = ( tr_mul != 1.0 );
	const float trm = (float) tr_mul;
	const float trmi = (float) ( 1.0 / tr_mul );
	const float pk = (float) pk_out;
	// OutGammaMult, avir.h:4756-4763
	const float ogm = (float) ( out_type == AVIRHIP_U8 ? 255.0 :
		( out_type == AVIRHIP_U16 ? 65535.0 : 1.0 ));

	if( !rne && ech == 4 && gamma && out_type == AVIRHIP_U8 && gthr != nullptr )
	{
		const long npx = n / ch;
		const dim3 grd( (unsigned) (( npx + 255 ) / 256 ));

#define ET( CH ) hipLaunchKernelGGL(( k_epilogue_gamma_thr< CH > ), grd, \
		dim3( 256 ), 0, st, res, (unsigned char*) dst, npx, alpha_index, gthr, \
		use_tr, trm, trmi, pk, ogm )

		switch( ch )
		{
			case 1: ET( 1 ); break;
			case 2: ET( 2 ); break;
			case 3: ET( 3 ); break;
			default: ET( 4 ); break;
		}

#undef ET
		AVIRHIP_HIPCHECK( hipGetLastError() );
		return( AVIRHIP_OK );
	}

	if( !rne && ech == 4 && gamma && ( out_type == AVIRHIP_U8 ||
		out_type == AVIRHIP_U16 ))
	{
		if( out_type == AVIRHIP_U8 )
			launch_epilogue_gamma_px< uint8_t >( res, dst, n / ch, ch, use_tr,
				trm, trmi, pk, alpha_index, ogm, st );
		else
			launch_epilogue_gamma_px< uint16_t >( res, dst, n / ch, ch, use_tr,
				trm, trmi, pk, alpha_index, ogm, st );

		AVIRHIP_HIPCHECK( hipGetLastError() );
		return( AVIRHIP_OK );
	}

	if( !rne && ech == 4 && !gamma && ( (uintptr_t) dst & 3 ) == 0 )
	{
		const long npx = n / ch;

		if( out_type == AVIRHIP_U8 )
			launch_epilogue_px< uint8_t, true >( res, dst, npx, ch, use_tr,
				trm, trmi, pk, st );
		else if( out_type == AVIRHIP_U16 )
			launch_epilogue_px< uint16_t, true >( res, dst, npx, ch, use_tr,
				trm, trmi, pk, st );
		else if( out_type == AVIRHIP_F64 )
			launch_epilogue_px< double, false >( res, dst, npx, ch, use_tr,
				trm, trmi, pk, st );
		else
			launch_epilogue_px< float, false >( res, dst, npx, ch, use_tr,
				trm, trmi, pk, st );

		AVIRHIP_HIPCHECK( hipGetLastError() );
		return( AVIRHIP_OK );
	}

	if( out_type == AVIRHIP_U8 )
		hipLaunchKernelGGL(( k_epilogue< uint8_t, true > ), grd, dim3( 256 ),
			0, st, res, (uint8_t*) dst, n, use_tr, trm, trmi, pk, gamma, ch,
			ech, alpha_index, ogm, ( rne ? 1 : 0 ));
	else if( out_type == AVIRHIP_U16 )
		hipLaunchKernelGGL(( k_epilogue< uint16_t, true > ), grd, dim3( 256 ),
			0, st, res, (uint16_t*) dst, n, use_tr, trm, trmi, pk, gamma, ch,
			ech, alpha_index, ogm, ( rne ? 1 : 0 ));
	else if( out_type == AVIRHIP_F64 )
		hipLaunchKernelGGL(( k_epilogue< double, false > ), grd, dim3( 256 ),
			0, st, res, (double*) dst, n, use_tr, trm, trmi, pk, gamma, ch,
			ech, alpha_index, ogm, ( rne ? 1 : 0 ));
	else
		hipLaunchKernelGGL(( k_epilogue< float, false > ), grd, dim3( 256 ),
			0, st, res, (float*) dst, n, use_tr, trm, trmi, pk, gamma, ch,
			ech, alpha_index, ogm, ( rne ? 1 : 0 ));

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

// ---------------------------------------------------------------------
// CImageResizerDithererErrdINL::dither (avir.h:4485-4525) + unpackScanline:
// error diffusion, "recursive" in the reference's terms -- pixel p of row r
// needs the quantisation noise of pixel p-1 of its own row and of pixels p-1,
// p, p+1 of the row above. That is a wavefront with a lag of two pixels per
// row: row r works on pixel t - 2r at step t. One workgroup, one thread per
// row (up to 1024 rows in flight, passes of 1024 rows for taller frames), one
// barrier per step; the diffusion value D[p-1] of a row is complete the
// moment the row has processed pixel p and is handed to the row below through
// a lane shuffle (LDS between waves, a global row between passes).
//
// Order of the float operations, per channel (all in the reference's order):
//   v      = ( R[p] + D_above[p] ) + NoiseM1[p-1]          (no third term at p = 0)
//   z0     = round( v * TrMulI ) * TrMul;  Noise = v - z0;  out = clamp( z0 )
//   D[p-1] = (( 0 + Noise[p-2]*c3 ) + Noise[p-1]*c1 ) + Noise[p]*c2
// with c1 = 0.364842, c2 = 0.207305, c3 = 0.063011 (float).
// ---------------------------------------------------------------------

#define ED_PF 8 // pixels a lane reads ahead of its step

template< typename Tout >
__global__ void __launch_bounds__( 1024 ) k_errd( const float* res, Tout* dst,
	const int w, const int h, const int ch, const int ech, const float tr_mul,
	const float tr_muli, const float pk_out, const int gamma,
	const int alpha_index, const float ogm, float* line )
{
	__shared__ float xch[ 2 ][ 16 ][ 4 ];
	const int rr = threadIdx.x;
	const int lane = rr & 63;
	const int wave = rr >> 6;
	const int nrows = blockDim.x;
	const float c1 = (float) 0.364842;
	const float c2 = (float) 0.207305;
	const float c3 = (float) 0.063011;
	int pass = 0;

	for( int row0 = 0; row0 < h; row0 += nrows, pass++ )
	{
		const int row = row0 + rr;
		const bool rowok = ( row < h );
		// rows of this pass; its last row feeds the next pass through `line`
		const int prow = min( nrows, h - row0 );
		const bool feeds = ( rr == prow - 1 && row0 + nrows < h );
		const float* const lin = line + (size_t) ( pass & 1 ) * w * 4;
		float* const lout = line + (size_t) (( pass + 1 ) & 1 ) * w * 4;
		const float* const rp = res + (size_t) row * w * ech;
		Tout* const dp = dst + (size_t) row * w * ch;
		float A[ 4 ], B[ 4 ], nm1[ 4 ], dout[ 4 ];
#pragma unroll
		for( int c = 0; c < 4; c++ )
		{
			A[ c ] = B[ c ] = nm1[ c ] = dout[ c ] = 0.0f;
		}

		const int nsteps = w + 2 * ( prow - 1 ) + 2;

		// A lane's pixels are read ED_PF steps ahead into a register ring (the
		// step loop is unrolled over it): a step is a barrier plus a few dozen
		// flops, and without the prefetch every one of the w + 2 * rows steps
		// exposed the latency of a load that nothing else could hide (one
		// workgroup on the chip) -- 45.7 -> ms per 4K RGB frame, see
		// profiles/r03_errd. The diffusion values handed over from the
		// previous pass (lane 0 of wave 0) travel the same way.
		float pf[ ED_PF ][ 4 ], lf[ ED_PF ][ 4 ];

		auto fetch = [&]( const int pn, float ( &o )[ 4 ], float ( &lo )[ 4 ])
		{
#pragma unroll
			for( int c = 0; c < 4; c++ )
			{
				o[ c ] = 0.0f;
				lo[ c ] = 0.0f;
			}

			if( pn >= 0 && pn < w )
			{
				if( rowok )
				{
					if( ech == 4 )
					{
						typedef float f4v __attribute__(( ext_vector_type( 4 )));
						const f4v v = *(const f4v*) ( rp + (size_t) pn * 4 );
						o[ 0 ] = v.x; o[ 1 ] = v.y; o[ 2 ] = v.z; o[ 3 ] = v.w;
					}
					else
					{
#pragma unroll
						for( int c = 0; c < 4; c++ )
						{
							if( c < ch )
							{
								o[ c ] = rp[ (size_t) pn * ech + c ];
							}
						}
					}
				}

				if( rr == 0 && pass > 0 )
				{
					// written by another wave of this workgroup in the
					// previous pass: read past the L1
#pragma unroll
					for( int c = 0; c < 4; c++ )
					{
						lo[ c ] = __hip_atomic_load( &lin[ pn * 4 + c ],
							__ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
					}
				}
			}
		};

#pragma unroll
		for( int k = 0; k < ED_PF; k++ )
		{
			fetch( k - 2 * rr, pf[ k ], lf[ k ]);
		}

		for( int t0 = 0; t0 < nsteps; t0 += ED_PF )
		{
#pragma unroll
		for( int k = 0; k < ED_PF; k++ )
		{
			const int t = t0 + k;

			if( t >= nsteps )
			{
				break;
			}

			const int p = t - 2 * rr;
			float cur[ 4 ], lcur[ 4 ];
#pragma unroll
			for( int c = 0; c < 4; c++ )
			{
				cur[ c ] = pf[ k ][ c ];
				lcur[ c ] = lf[ k ][ c ];
			}

			fetch( p + ED_PF, pf[ k ], lf[ k ]);

			float din[ 4 ];
#pragma unroll
			for( int c = 0; c < 4; c++ )
			{
				din[ c ] = __shfl_up( dout[ c ], 1 );
			}

			if( lane == 0 )
			{
#pragma unroll
				for( int c = 0; c < 4; c++ )
				{
					if( wave > 0 )
					{
						din[ c ] = xch[( t + 1 ) & 1 ][ wave - 1 ][ c ];
					}
					else
					{
						din[ c ] = lcur[ c ]; // 0 in the first pass
					}
				}
			}

			if( rowok && p >= 0 && p < w )
			{
#pragma unroll
				for( int c = 0; c < 4; c++ )
				{
					if( c < ch )
					{
						float v = cur[ c ];

						if( gamma )
						{
							v = ( c == alpha_index ? v * ogm :
								lin_to_srgb( v ) * ogm );
						}

						v = v + din[ c ];

						if( p > 0 )
						{
							v = v + nm1[ c ];
						}

						const float z0 = avir_round( v * tr_muli ) * tr_mul;
						const float noise = v - z0;
						const float o = ( z0 < 0.0f ? 0.0f :
							( z0 > pk_out ? pk_out : z0 ));

						dp[ (size_t) p * ch + c ] = (Tout) o;
						const float m1 = noise * c1;
						dout[ c ] = A[ c ] + noise * c2; // D[p-1] complete
						A[ c ] = B[ c ] + m1;            // D[p]: two of three
						B[ c ] = 0.0f + noise * c3;      // D[p+1]: first term
						nm1[ c ] = m1;
					}
				}
			}
			else
			if( rowok && p == w )
			{
				// the last pixel's D has no right-hand neighbour term
#pragma unroll
				for( int c = 0; c < 4; c++ )
				{
					dout[ c ] = A[ c ];
				}
			}

			if( feeds && p >= 1 && p <= w )
			{
#pragma unroll
				for( int c = 0; c < 4; c++ )
				{
					lout[ ( p - 1 ) * 4 + c ] = dout[ c ];
				}
			}

			if( lane == 63 )
			{
#pragma unroll
				for( int c = 0; c < 4; c++ )
				{
					xch[ t & 1 ][ wave ][ c ] = dout[ c ];
				}
			}

			__syncthreads();
		}
		}

		__threadfence();
		__syncthreads();
	}
}

// A lane's output bytes collect in a 16-byte block (the aligned block of the
// destination they belong to) and leave as ONE 16-byte store when the lane's
// write position moves on: element-wide stores from 896 lanes that each walk
// their own image row are 2688 one-byte write requests per step, and those
// requests, not the arithmetic or the loads, set the step time (31.7 ms per
// 4K RGB frame before). Blocks a lane filled only in part -- the two ends of a
// row -- leave as byte stores of exactly the bytes it owns.
struct ErrdOut
{
	unsigned long long lo, hi;
	unsigned long long blk; // address >> 4 of the block being filled, 0: none
	unsigned mask;          // bytes of the block written so far
};

__device__ __forceinline__ void eo_flush( ErrdOut& o )
{
	if( o.mask == 0 )
	{
		return;
	}

	unsigned char* const q = (unsigned char*) ( o.blk << 4 );

	if( o.mask == 0xffffu )
	{
		typedef unsigned long long ull2 __attribute__(( ext_vector_type( 2 )));
		ull2 v; v.x = o.lo; v.y = o.hi;
		*(ull2*) q = v;
	}
	else
	{
		for( int i = 0; i < 16; i++ )
		{
			if(( o.mask >> i ) & 1 )
			{
				q[ i ] = (unsigned char) (( i < 8 ? o.lo >> ( 8 * i ) :
					o.hi >> ( 8 * ( i - 8 ))) & 0xff );
			}
		}
	}

	o.mask = 0;
}

// `n` bytes (one pixel: 1-4 elements, <= 8 bytes) at address `a`; they may
// run over the end of the block being filled
__device__ __forceinline__ void eo_put( ErrdOut& o, const unsigned long long a,
	const unsigned long long v, const int n )
{
	const unsigned long long b = a >> 4;

	if( b != o.blk )
	{
		eo_flush( o );
		o.blk = b; o.lo = 0; o.hi = 0;
	}

	const int k = (int) ( a & 15 );
	const int n1 = min( n, 16 - k ); // bytes that fit into this block

	if( k < 8 )
	{
		o.lo |= v << ( 8 * k );

		if( k > 0 && k + n > 8 )
		{
			o.hi |= v >> ( 8 * ( 8 - k ));
		}
	}
	else
	{
		o.hi |= v << ( 8 * ( k - 8 ));
	}

	o.mask |= (( 1u << n1 ) - 1u ) << k;

	if( n1 < n )
	{
		eo_flush( o );
		o.blk = b + 1; o.hi = 0;
		o.lo = v >> ( 8 * n1 );
		o.mask = ( 1u << ( n - n1 )) - 1u;
	}
}

// ---------------------------------------------------------------------
// The same ditherer for padded RGBA float results (ech == 4), frames of any
// height, on several compute units: the frame is cut into PASSES of 448 rows,
// one workgroup each, chained through the diffusion values of a pass' last
// row -- pass k + 1 starts as soon as pass k has published the first few of
// them (a flag per pass in global memory, write-through stores + drained flag
// on the producing side, L1-bypassing loads on the consuming side), so the
// whole frame costs about width + 2 * height steps instead of the sum over
// the passes. A pass only ever waits for the pass before it, i.e. for a
// workgroup with a lower index.
//
// Inside a workgroup 7 waves compute (lane = row) and an eighth does nothing
// but LDS-DMA: a lane's pixel t - 2 rr sits in ITS row, 448 cache lines apart
// from its neighbours', and per-lane loads (in the compute lanes, or by DMA
// one pixel per lane) made the texture path look up ~900 lines per step --
// that, not the arithmetic, set the step time (profiles/r03_errd). The I/O
// wave instead moves whole 128-byte lines: one instruction = 8 rows x 8 pixels,
// the 8 rows chosen 4 apart so that their skewed pixel positions share a
// phase, 15 instructions every second step, landing 8 steps before use in a
// two-slot LDS ring (vmcnt counts the bursts in: loads retire in order).
// ---------------------------------------------------------------------
#define EJ_CW 7                 // compute waves
#define EJ_ROWS ( EJ_CW * 64 )  // rows per pass
#define EJ_NG ( EJ_ROWS / 8 )   // row groups (one DMA instruction each)
#define EJ_LDS ( 2 * EJ_NG * 1024 + 2 * 8 * 16 + 2 * 8 * 4 * 4 )

__device__ __forceinline__ void ej_dma( const __amdgpu_buffer_rsrc_t rs,
	const unsigned lds, const int voff )
{
	asm volatile( AVIRHIP_HANDOVER "s_mov_b32 m0, %0\n\t"
		"s_nop 0\n\t"
		"buffer_load_dwordx4 %1, %2, 0 offen lds"
		:: "s"( lds ), "v"( voff ), "s"( rs ) : "memory" );
}

// (past the L1 and the L2: another compute unit has just written the line)
__device__ __forceinline__ void ej_dma_sc( const __amdgpu_buffer_rsrc_t rs,
	const unsigned lds, const int voff )
{
	asm volatile( AVIRHIP_HANDOVER "s_mov_b32 m0, %0\n\t"
		"s_nop 0\n\t"
		"buffer_load_dwordx4 %1, %2, 0 offen sc0 sc1 lds"
		:: "s"( lds ), "v"( voff ), "s"( rs ) : "memory" );
}

template< typename Tout >
__global__ void __launch_bounds__( 512 ) k_errd_mp( const float* res,
	Tout* dst, const int w, const int h, const int ch, const float tr_mul,
	const float tr_muli, const float pk_out, const int gamma,
	const int alpha_index, const float ogm, float* line, unsigned* flags,
	const unsigned epoch, const int pass0 )
{
	typedef float f4v __attribute__(( ext_vector_type( 4 )));
	typedef __attribute__(( address_space( 3 ))) char lchar;
	extern __shared__ __attribute__(( aligned( 16 ))) char smem[];
	f4v* const sIn = (f4v*) smem;                        // [2][EJ_NG][64]
	f4v* const sL = (f4v*) ( smem + 2 * EJ_NG * 1024 );  // [2][8]
	float ( * const xch )[ 8 ][ 4 ] = (float (*)[ 8 ][ 4 ])
		( smem + 2 * EJ_NG * 1024 + 256 );               // [2][8][4]
	const unsigned lds0 = (unsigned) (unsigned long) (lchar*) smem;
	const unsigned ldsL = lds0 + 2 * EJ_NG * 1024;

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane( tid >> 6 );
	const int pass = pass0 + blockIdx.x;
	const int row0 = pass * EJ_ROWS;
	const int prow = min( EJ_ROWS, h - row0 );
	const int nsteps = w + 2 * ( prow - 1 ) + 2;
	const bool last_pass = ( row0 + EJ_ROWS >= h );
	// the diffusion values a pass hands down: row `pass` of `line`
	float* const lout = line + (size_t) pass * w * 4;
	const float c1 = (float) 0.364842;
	const float c2 = (float) 0.207305;
	const float c3 = (float) 0.063011;

	if( wave == EJ_CW )
	{
		// ---- the I/O wave
		const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
			(void*) ( res + (size_t) row0 * w * 4 ), 0, 0x7fffffff, 0x00020000 );
		const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(
			(void*) ( pass > 0 ? line + (size_t) ( pass - 1 ) * w * 4 : line ),
			0, 0x7fffffff, 0x00020000 );
		const int li = lane >> 3; // row of the group
		const int lk = lane & 7;  // pixel of the segment
		int have = ( pass > 0 ? 0 : 0x7fffffff ); // values the pass above published

		// burst of step T (even): the 14 groups whose rows start a segment
		// at step T + 8, and the segment of values from the pass above
		auto burst = [&]( const int T )
		{
			// (the ring slot goes by the step the segments are first read at,
			// T + 8, the same for every row of the burst)
			const int r0 = ( T >> 1 ) & 3;
			const int slot = (( T + 8 ) >> 3 ) & 1;
#pragma unroll 1
			for( int a = 0; a < EJ_NG / 4; a++ )
			{
				const int g = a * 4 + r0;
				const int rr = 32 * a + 4 * li + r0;
				const int px = max( 0, min( T + 8 - 2 * rr + lk, w - 1 ));
				ej_dma( rs, lds0 + ( slot * EJ_NG + g ) * 1024,
					min( rr, prow - 1 ) * w * 16 + px * 16 );
			}

			// (an instruction at every burst: the wait counts are constants)
			const int ps = ( T + 8 ) & ~7;

			if( pass > 0 && ( T & 7 ) == 0 && ps < w )
			{
				const int need = min( ps + 8, w );

				while( have < need )
				{
					const unsigned f = __hip_atomic_load( &flags[ pass - 1 ],
						__ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );

					have = ( ( f >> 20 ) == epoch ? (int) ( f & 0xfffffu ) : 0 );

					if( have < need )
					{
						__builtin_amdgcn_s_sleep( 8 );
					}
				}
			}

			if( lane < 8 )
			{
				const int px = max( 0, min( ps + lane, w - 1 ));

				if( pass > 0 )
				{
					ej_dma_sc( rl, ldsL + slot * 128, px * 16 );
				}
				else
				{
					ej_dma( rs, ldsL + slot * 128, 0 );
				}
			}
		};

		for( int T = -8; T < 0; T += 2 )
		{
			burst( T );
		}

		__syncthreads(); // (the compute waves' set-up)

		for( int t = 0; t < nsteps; t++ )
		{
			if(( t & 1 ) == 0 )
			{
				burst( t );
				// bursts t, t - 2, t - 4 may be in flight: t - 6 (first read
				// at step t + 2) has landed
				AVIRHIP_WAIT_VM( 45 );
			}

			__syncthreads();
		}

		AVIRHIP_WAIT_VM( 0 );
		return;
	}

	// ---- compute waves: one row per lane
	const int rr = tid;
	const int row = row0 + rr;
	const bool rowok = ( row < h );
	const bool feeds = ( rr == prow - 1 && !last_pass );
	Tout* const dp = dst + (size_t) row * w * ch;
	// where this row's pixels sit in the ring: group, row of the group
	const int grp = ( rr >> 5 ) * 4 + ( rr & 3 );
	const int gi = ( rr & 31 ) >> 2;
	float A[ 4 ], B[ 4 ], nm1[ 4 ], dout[ 4 ];
#pragma unroll
	for( int c = 0; c < 4; c++ )
	{
		A[ c ] = B[ c ] = nm1[ c ] = dout[ c ] = 0.0f;
	}

	ErrdOut eo;
	eo.lo = 0; eo.hi = 0; eo.blk = 0; eo.mask = 0;

	__syncthreads();

	for( int t = 0; t < nsteps; t++ )
	{
		const int p = t - 2 * rr;
		const f4v px = sIn[ (((( rr >> 2 ) + ( p >> 3 )) & 1 ) * EJ_NG + grp ) *
			64 + gi * 8 + ( p & 7 )];
		const float cur[ 4 ] = { px.x, px.y, px.z, px.w };
		float din[ 4 ];
#pragma unroll
		for( int c = 0; c < 4; c++ )
		{
			din[ c ] = __shfl_up( dout[ c ], 1 );
		}

		if( lane == 0 )
		{
			if( wave > 0 )
			{
#pragma unroll
				for( int c = 0; c < 4; c++ )
				{
					din[ c ] = xch[( t + 1 ) & 1 ][ wave - 1 ][ c ];
				}
			}
			else
			if( pass > 0 && p < w )
			{
				const f4v l = sL[ (( p >> 3 ) & 1 ) * 8 + ( p & 7 )];
				din[ 0 ] = l.x; din[ 1 ] = l.y; din[ 2 ] = l.z; din[ 3 ] = l.w;
			}
			else
			{
#pragma unroll
				for( int c = 0; c < 4; c++ )
				{
					din[ c ] = 0.0f;
				}
			}
		}

		if( rowok && p >= 0 && p < w )
		{
			unsigned long long opx = 0; // the pixel's elements, packed
#pragma unroll
			for( int c = 0; c < 4; c++ )
			{
				if( c < ch )
				{
					float v = cur[ c ];

					if( gamma )
					{
						v = ( c == alpha_index ? v * ogm :
							lin_to_srgb( v ) * ogm );
					}

					v = v + din[ c ];

					if( p > 0 )
					{
						v = v + nm1[ c ];
					}

					const float z0 = avir_round( v * tr_muli ) * tr_mul;
					const float noise = v - z0;
					const float o = ( z0 < 0.0f ? 0.0f :
						( z0 > pk_out ? pk_out : z0 ));

					opx |= (unsigned long long) (Tout) o <<
						( 8 * sizeof( Tout ) * c );
					const float m1 = noise * c1;
					dout[ c ] = A[ c ] + noise * c2; // D[p-1] complete
					A[ c ] = B[ c ] + m1;            // D[p]: two of three
					B[ c ] = 0.0f + noise * c3;      // D[p+1]: first term
					nm1[ c ] = m1;
				}
			}

			eo_put( eo, (unsigned long long) ( dp + (size_t) p * ch ), opx,
				ch * (int) sizeof( Tout ));
		}
		else
		if( rowok && p == w )
		{
			// the last pixel's D has no right-hand neighbour term
#pragma unroll
			for( int c = 0; c < 4; c++ )
			{
				dout[ c ] = A[ c ];
			}

			eo_flush( eo ); // the row's last, partly filled block
		}

		if( feeds && p >= 1 && p <= w )
		{
			// write-through: the pass below reads them on another compute unit
#pragma unroll
			for( int c = 0; c < 4; c++ )
			{
				__hip_atomic_store( &lout[ ( p - 1 ) * 4 + c ], dout[ c ],
					__ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
			}

			if(( p & 7 ) == 0 || p == w )
			{
				AVIRHIP_WAIT_VM( 0 );
				__hip_atomic_store( &flags[ pass ], ( epoch << 20 ) |
					(unsigned) p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
			}
		}

		if( lane == 63 )
		{
#pragma unroll
			for( int c = 0; c < 4; c++ )
			{
				xch[ t & 1 ][ wave ][ c ] = dout[ c ];
			}
		}

		__syncthreads();
	}
}

int launch_errd( const float* res, void* dst, int out_type, int w, int h,
	int ch, int ech, double tr_mul, double pk_out, int gamma, int alpha_index,
	float* line, hipStream_t st )
{
	const float trm = (float) tr_mul;
	const float trmi = (float) ( 1.0 / tr_mul );
	const float pk = (float) pk_out;
	const float ogm = (float) ( out_type == AVIRHIP_U8 ? 255.0 : 65535.0 );
	const int nt = std::min( 1024, (( h + 63 ) / 64 ) * 64 );

	// padded RGBA results of frames worth the set-up: one workgroup per pass
	// of 448 rows, chained (k_errd_mp)
	if( ech == 4 && ( (uintptr_t) res & 15 ) == 0 && (long) w * h >= 65536 &&
		w < ( 1 << 20 ) && (long) w * 16 * EJ_ROWS < 0x7fffffffL &&
		( out_type == AVIRHIP_U8 || out_type == AVIRHIP_U16 ) &&
		getenv( "AVIRHIP_ERRD_PLAIN" ) == nullptr )
	{
		static std::atomic< unsigned > calls( 0 );
		// (never 0: the flags are zeroed below)
		const unsigned epoch = ( ++calls ) % 0xfffu + 1;
		const int npass = ( h + EJ_ROWS - 1 ) / EJ_ROWS;
		// (launch_errd's caller sized `line` for npass rows of values + flags)
		unsigned* const flags = (unsigned*) ( line + (size_t) npass * w * 4 );
		// The flags outlive the launch in the plan's buffer, and the epoch is
		// 12 bits of a process-wide counter: a plan that runs again a multiple of
		// the period later (or fresh memory that holds another plan's old flags)
		// would find "valid" flags. Zeroed in stream order before every frame.
		AVIRHIP_HIPCHECK( hipMemsetAsync( flags, 0, (size_t) npass *
			sizeof( unsigned ), st ));
		// AVIRHIP_ERRD_SERIAL: one launch per pass (stream order instead of
		// the flags; debugging aid)
		const bool serial = ( getenv( "AVIRHIP_ERRD_SERIAL" ) != nullptr );

		for( int p0 = 0; p0 < npass; p0 += ( serial ? 1 : npass ))
		{
			const dim3 grid( serial ? 1 : npass );

			if( out_type == AVIRHIP_U8 )
			{
				AVIRHIP_HIPCHECK( AVIRHIP_DYN_LDS(( k_errd_mp< uint8_t > ),
					EJ_LDS ));
				hipLaunchKernelGGL(( k_errd_mp< uint8_t > ), grid, dim3( 512 ),
					EJ_LDS, st, res, (uint8_t*) dst, w, h, ch, trm, trmi, pk,
					gamma, alpha_index, ogm, line, flags, epoch, p0 );
			}
			else
			{
				AVIRHIP_HIPCHECK( AVIRHIP_DYN_LDS(( k_errd_mp< uint16_t > ),
					EJ_LDS ));
				hipLaunchKernelGGL(( k_errd_mp< uint16_t > ), grid, dim3( 512 ),
					EJ_LDS, st, res, (uint16_t*) dst, w, h, ch, trm, trmi, pk,
					gamma, alpha_index, ogm, line, flags, epoch, p0 );
			}
		}

		AVIRHIP_HIPCHECK( hipGetLastError() );
		return( AVIRHIP_OK );
	}

	if( out_type == AVIRHIP_U8 )
	{
		hipLaunchKernelGGL(( k_errd< uint8_t > ), dim3( 1 ), dim3( nt ), 0, st,
			res, (uint8_t*) dst, w, h, ch, ech, trm, trmi, pk, gamma,
			alpha_index, ogm, line );
	}
	else
	if( out_type == AVIRHIP_U16 )
	{
		hipLaunchKernelGGL(( k_errd< uint16_t > ), dim3( 1 ), dim3( nt ), 0, st,
			res, (uint16_t*) dst, w, h, ch, ech, trm, trmi, pk, gamma,
			alpha_index, ogm, line );
	}
	else
	{
		set_error( "error-diffusion dither: integer output types only" );
		return( AVIRHIP_EINVAL );
	}

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

// ---------------------------------------------------------------------
// LANCIR generic kernels (lancir.h:541-710): vertical pass first into a float
// intermediate [rows][SrcWidth], then the horizontal pass, then the output
// stage. The dot product of one output pixel reproduces the lane structure of
// the reference's x86 SSE2 code (resize1..4, lancir.h:2102-2544), written out
// as scalar expressions -- the summation ORDER is the parity contract:
//   4 ch: even taps and odd taps accumulate separately, added at the end
//   1 ch: four lane sums over taps 4g+L, then (s0+s2)+(s1+s3); a kernel length
//         with kl%4 == 2 adds its last two taps to the two partial sums first
//   2 ch: lanes {4g, 4g+1} and {4g+2, 4g+3} per channel, folded A+B, optional
//         tail taps, then lane0 + lane2
//   3 ch: the 12 products of 4 taps x 3 channels fill three 4-lane vectors
// In every lane the first term is the bare product (no +0).
// ---------------------------------------------------------------------

// P( t, c ): input sample of tap t, channel c (edge-clamped by the caller).
template< int CH, typename PX >
__device__ __forceinline__ void lancir_dot( const float* __restrict__ f,
	const int kl, PX P, float* out )
{
	if( CH == 4 )
	{
		float ea[ 4 ], oa[ 4 ];

		for( int c = 0; c < 4; c++ )
		{
			ea[ c ] = f[ 0 ] * P( 0, c );
			oa[ c ] = f[ 1 ] * P( 1, c );
		}

		for( int t = 2; t < kl; t += 2 )
		{
			for( int c = 0; c < 4; c++ )
			{
				ea[ c ] += f[ t ] * P( t, c );
				oa[ c ] += f[ t + 1 ] * P( t + 1, c );
			}
		}

		for( int c = 0; c < 4; c++ )
		{
			out[ c ] = ea[ c ] + oa[ c ];
		}
	}
	else
	if( CH == 1 )
	{
		const int ci = kl >> 2;
		float s0 = f[ 0 ] * P( 0, 0 ), s1 = f[ 1 ] * P( 1, 0 );
		float s2 = f[ 2 ] * P( 2, 0 ), s3 = f[ 3 ] * P( 3, 0 );

		for( int g = 1; g < ci; g++ )
		{
			const int q = g * 4;
			s0 += f[ q ] * P( q, 0 );
			s1 += f[ q + 1 ] * P( q + 1, 0 );
			s2 += f[ q + 2 ] * P( q + 2, 0 );
			s3 += f[ q + 3 ] * P( q + 3, 0 );
		}

		if(( kl & 3 ) == 0 )
		{
			out[ 0 ] = ( s0 + s2 ) + ( s1 + s3 );
		}
		else
		{
			const float e0 = f[ kl - 2 ] * P( kl - 2, 0 );
			const float e1 = f[ kl - 1 ] * P( kl - 1, 0 );
			out[ 0 ] = (( s0 + s2 ) + e0 ) + (( s1 + s3 ) + e1 );
		}
	}
	else
	if( CH == 2 )
	{
		const int ci = kl >> 2;

		for( int c = 0; c < 2; c++ )
		{
			float a0 = f[ 0 ] * P( 0, c ), a2 = f[ 1 ] * P( 1, c );
			float b0 = f[ 2 ] * P( 2, c ), b2 = f[ 3 ] * P( 3, c );

			for( int g = 1; g < ci; g++ )
			{
				const int q = g * 4;
				a0 += f[ q ] * P( q, c );
				a2 += f[ q + 1 ] * P( q + 1, c );
				b0 += f[ q + 2 ] * P( q + 2, c );
				b2 += f[ q + 3 ] * P( q + 3, c );
			}

			a0 = a0 + b0;
			a2 = a2 + b2;

			if(( kl & 3 ) == 2 )
			{
				a0 += f[ kl - 2 ] * P( kl - 2, c );
				a2 += f[ kl - 1 ] * P( kl - 1, c );
			}

			out[ c ] = a0 + a2;
		}
	}
	else
	{
		// products e[i], i = 3*tap + channel; vector lanes hold i mod 12
		const int ci = kl >> 2;
		float r[ 12 ];

		for( int i = 0; i < 12; i++ )
		{
			r[ i ] = f[ i / 3 ] * P( i / 3, i % 3 );
		}

		for( int g = 1; g < ci; g++ )
		{
			for( int i = 0; i < 12; i++ )
			{
				const int t = g * 4 + i / 3;
				r[ i ] += f[ t ] * P( t, i % 3 );
			}
		}

		if(( kl & 3 ) == 2 )
		{
			const int t = ci * 4;
			r[ 0 ] += f[ t ] * P( t, 0 );
			r[ 1 ] += f[ t ] * P( t, 1 );
			r[ 2 ] += f[ t ] * P( t, 2 );
			r[ 3 ] += f[ t + 1 ] * P( t + 1, 0 );
		}

		float o0 = r[ 0 ] + r[ 3 ];
		float o1 = r[ 1 ] + r[ 4 ];
		float o2 = r[ 2 ] + r[ 5 ];
		o0 += r[ 6 ] + r[ 9 ];
		o1 += r[ 7 ] + r[ 10 ];
		o2 += r[ 8 ] + r[ 11 ];

		if(( kl & 3 ) == 2 )
		{
			const int t = ci * 4 + 1;
			o1 += f[ t ] * P( t, 1 );
			o2 += f[ t ] * P( t, 2 );
		}

		out[ 0 ] = o0;
		out[ 1 ] = o1;
		out[ 2 ] = o2;
	}
}

template< typename Tin, int CH >
__global__ void __launch_bounds__( 256 ) k_lancir_v( const Tin* src,
	float* tmp, int sw, int sh, long src_stride, int row0, int nrows,
	const int* start, const int* fidx, const float* flt, int kl )
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	const int ry = blockIdx.y;

	if( x >= sw || ry >= nrows )
	{
		return;
	}

	const int r = row0 + ry;
	const int st = start[ r ];
	const float* f = flt + (long) fidx[ r ] * kl;
	const Tin* col = src + (long) x * CH;
	float o[ CH ];

	lancir_dot< CH >( f, kl, [&]( const int t, const int c ) -> float
		{
			int y = st + t;
			y = ( y < 0 ? 0 : ( y >= sh ? sh - 1 : y ));
			return( (float) col[ (long) y * src_stride + c ]);
		}, o );

	float* op = tmp + ( (long) ry * sw + x ) * CH;

	for( int c = 0; c < CH; c++ )
	{
		op[ c ] = o[ c ];
	}
}

template< int CH >
__global__ void __launch_bounds__( 256 ) k_lancir_h( const float* tmp,
	float* res, int sw, int nw, long res_stride, int nrows, const int* start,
	const int* fidx, const float* flt, int kl )
{
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	const int ry = blockIdx.y;

	if( j >= nw || ry >= nrows )
	{
		return;
	}

	const int st = start[ j ];
	const float* f = flt + (long) fidx[ j ] * kl;
	const float* row = tmp + (long) ry * sw * CH;
	float o[ CH ];

	lancir_dot< CH >( f, kl, [&]( const int t, const int c ) -> float
		{
			int x = st + t;
			x = ( x < 0 ? 0 : ( x >= sw ? sw - 1 : x ));
			return( row[ x * CH + c ]);
		}, o );

	float* op = res + (long) ry * res_stride + (long) j * CH;

	for( int c = 0; c < CH; c++ )
	{
		op[ c ] = o[ c ];
	}
}

// outputScanline, lancir.h:1772-2056 (x86 SSE2 build): float output is a
// copy or a multiply; integer output clamps, then the elements of each full
// group of four round to nearest-even (cvtps) while the 1-3 element tail of
// the scanline uses roundclamp's +0.5 truncation (lancir.h:1746-1750).
template< typename Tout >
__global__ void __launch_bounds__( 256 ) k_lancir_out( const float* res,
	Tout* dst, int l, long res_stride, long dst_stride, int nrows, int unity,
	float out_mul, float clampv, int is_float )
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	const int ry = blockIdx.y;

	if( i >= l || ry >= nrows )
	{
		return;
	}

	float v = res[ (long) ry * res_stride + i ];

	if( !unity )
	{
		v = v * out_mul;
	}

	Tout* op = dst + (long) ry * dst_stride + i;

	if( is_float )
	{
		*op = (Tout) v;
		return;
	}

	if( i < ( l & ~3 ))
	{
		v = ( v < clampv ? v : clampv );
		v = ( v > 0.0f ? v : 0.0f );
		*op = (Tout) (int) rintf( v );
	}
	else
	{
		*op = (Tout) (int) (( v > clampv ? clampv :
			( v < 0.0f ? 0.0f : v )) + 0.5f );
	}
}

template< typename Tin >
static void launch_lancir_v( const avirhip_plan* p, const void* src,
	float* tmp, int row0, int nrows, hipStream_t st )
{
	const LancirAxisDev& V = p -> lv;
	const dim3 gv(( p -> src_w + 255 ) / 256, nrows );

#define LV( CH ) hipLaunchKernelGGL(( k_lancir_v< Tin, CH > ), gv, dim3( 256 ), \
		0, st, (const Tin*) src, tmp, p -> src_w, p -> src_h, \
		(long) p -> src_stride, row0, nrows, V.d_start, V.d_fidx, V.d_flt, \
		V.kernel_len )

	switch( p -> ch )
	{
		case 1: LV( 1 ); break;
		case 2: LV( 2 ); break;
		case 3: LV( 3 ); break;
		default: LV( 4 ); break;
	}

#undef LV
}

int launch_lancir_generic( const avirhip_plan* p, const void* src, void* dst,
	float* tmp, int row0, int row1, hipStream_t st )
{
	const int nrows = row1 - row0;

	if( nrows <= 0 )
	{
		return( AVIRHIP_OK );
	}

	const LancirAxisDev& H = p -> lh;

	if( p -> in_type == AVIRHIP_U8 )
		launch_lancir_v< uint8_t >( p, src, tmp, row0, nrows, st );
	else if( p -> in_type == AVIRHIP_U16 )
		launch_lancir_v< uint16_t >( p, src, tmp, row0, nrows, st );
	else if( p -> in_type == AVIRHIP_F64 )
		launch_lancir_v< double >( p, src, tmp, row0, nrows, st ); // (float) ip[ c ]
	else if( p -> in_type == AVIRHIP_U32 )
		launch_lancir_v< uint32_t >( p, src, tmp, row0, nrows, st );
	else
		launch_lancir_v< float >( p, src, tmp, row0, nrows, st );

	AVIRHIP_HIPCHECK( hipGetLastError() );

	// float -> float with unity gain writes straight into the destination
	const bool direct = ( p -> out_type == AVIRHIP_F32 && p -> l_unity );
	float* res = ( direct ? (float*) dst : p -> lres );
	const long rstride = ( direct ? (long) p -> new_stride :
		(long) p -> new_w * p -> ch );

	const dim3 gh(( p -> new_w + 255 ) / 256, nrows );

#define LH( CH ) hipLaunchKernelGGL( k_lancir_h< CH >, gh, dim3( 256 ), 0, st, \
		tmp, res, p -> src_w, p -> new_w, rstride, nrows, H.d_start, H.d_fidx, \
		H.d_flt, H.kernel_len )

	switch( p -> ch )
	{
		case 1: LH( 1 ); break;
		case 2: LH( 2 ); break;
		case 3: LH( 3 ); break;
		default: LH( 4 ); break;
	}

#undef LH

	AVIRHIP_HIPCHECK( hipGetLastError() );

	if( !direct )
	{
		return( launch_lancir_out( p, res, rstride, dst, nrows, st ));
	}

	return( AVIRHIP_OK );
}

// RGBA rows: a scanline is whole groups of four elements (no tail), one thread
// converts a pixel -- a 16-byte load, the clamps, four nearest-even roundings
// and one 4- or 8-byte store.
template< typename Tout >
__global__ void __launch_bounds__( 256 ) k_lancir_out_px4( const float* res,
	Tout* dst, int w, long res_stride, long dst_stride, int nrows, int unity,
	float out_mul, float clampv )
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	const int ry = blockIdx.y;

	if( x >= w || ry >= nrows )
	{
		return;
	}

	typedef float f4v __attribute__(( ext_vector_type( 4 )));
	const f4v v4 = *(const f4v*) ( res + (long) ry * res_stride + (long) x * 4 );
	float v[ 4 ] = { v4.x, v4.y, v4.z, v4.w };
	unsigned int q[ 4 ];
#pragma unroll
	for( int c = 0; c < 4; c++ )
	{
		float t = v[ c ];

		if( !unity )
		{
			t = t * out_mul;
		}

		t = ( t < clampv ? t : clampv );
		t = ( t > 0.0f ? t : 0.0f );
		q[ c ] = (unsigned int) (int) rintf( t );
	}

	Tout* const op = dst + (long) ry * dst_stride + (long) x * 4;

	if( sizeof( Tout ) == 1 )
	{
		*(unsigned int*) op = q[ 0 ] | ( q[ 1 ] << 8 ) | ( q[ 2 ] << 16 ) |
			( q[ 3 ] << 24 );
	}
	else
	{
		typedef unsigned int u2v __attribute__(( ext_vector_type( 2 )));
		u2v o; o.x = q[ 0 ] | ( q[ 1 ] << 16 ); o.y = q[ 2 ] | ( q[ 3 ] << 16 );
		*(u2v*) op = o;
	}
}

// 1-3 channel images computed zero-padded to RGBA: the result rows hold four
// floats per pixel, the scanline new_w * CH elements -- element i = x * CH + c
// rounds by ITS position (full groups of four nearest-even, the tail + 0.5).
template< typename Tout, int CH >
__global__ void __launch_bounds__( 256 ) k_lancir_out_pad( const float* res,
	Tout* dst, int w, long dst_stride, int nrows, int unity, float out_mul,
	float clampv, int is_float )
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	const int ry = blockIdx.y;

	if( x >= w || ry >= nrows )
	{
		return;
	}

	typedef float f4v __attribute__(( ext_vector_type( 4 )));
	const f4v v4 = *(const f4v*) ( res + ( (long) ry * w + x ) * 4 );
	const float vv[ 4 ] = { v4.x, v4.y, v4.z, v4.w };
	const int l = w * CH;
	Tout* const op = dst + (long) ry * dst_stride + (long) x * CH;
#pragma unroll
	for( int c = 0; c < CH; c++ )
	{
		float v = vv[ c ];

		if( !unity )
		{
			v = v * out_mul;
		}

		if( is_float )
		{
			op[ c ] = (Tout) v;
		}
		else
		if( x * CH + c < ( l & ~3 ))
		{
			v = ( v < clampv ? v : clampv );
			v = ( v > 0.0f ? v : 0.0f );
			op[ c ] = (Tout) (int) rintf( v );
		}
		else
		{
			op[ c ] = (Tout) (int) (( v > clampv ? clampv :
				( v < 0.0f ? 0.0f : v )) + 0.5f );
		}
	}
}

template< typename Tout >
static void launch_out_pad( const avirhip_plan* p, const float* res, void* dst,
	int nrows, int is_float, hipStream_t st )
{
	const dim3 gp(( p -> new_w + 255 ) / 256, nrows );

#define LP( CH ) hipLaunchKernelGGL(( k_lancir_out_pad< Tout, CH > ), gp, \
		dim3( 256 ), 0, st, res, (Tout*) dst, p -> new_w, \
		(long) p -> new_stride, nrows, p -> l_unity, p -> l_out_mul, \
		p -> l_clamp, is_float )

	switch( p -> io_ch )
	{
		case 1: LP( 1 ); break;
		case 2: LP( 2 ); break;
		default: LP( 3 ); break;
	}

#undef LP
}

int launch_lancir_out_pad( const avirhip_plan* p, const float* res, void* dst,
	int nrows, hipStream_t st )
{
	if( p -> out_type == AVIRHIP_U8 )
		launch_out_pad< uint8_t >( p, res, dst, nrows, 0, st );
	else if( p -> out_type == AVIRHIP_U16 )
		launch_out_pad< uint16_t >( p, res, dst, nrows, 0, st );
	else if( p -> out_type == AVIRHIP_U32 )
		launch_out_pad< uint32_t >( p, res, dst, nrows, 0, st );
	else if( p -> out_type == AVIRHIP_F64 )
		launch_out_pad< double >( p, res, dst, nrows, 1, st );
	else
		launch_out_pad< float >( p, res, dst, nrows, 1, st );

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

// The output stage alone: `nrows` result rows of new_w * ch floats (row pitch
// rstride) into the caller's rows.
int launch_lancir_out( const avirhip_plan* p, const float* res, long rstride,
	void* dst, int nrows, hipStream_t st )
{
	const size_t oes = ( p -> out_type == AVIRHIP_U8 ? 1 : 2 );

	if( p -> ch == 4 && ( p -> out_type == AVIRHIP_U8 ||
		p -> out_type == AVIRHIP_U16 ) && ( rstride & 3 ) == 0 &&
		( (uintptr_t) res & 15 ) == 0 && ( (uintptr_t) dst & ( 4 * oes - 1 )) == 0 &&
		( p -> new_stride & 3 ) == 0 )
	{
		const dim3 gp(( p -> new_w + 255 ) / 256, nrows );

		if( p -> out_type == AVIRHIP_U8 )
		{
			hipLaunchKernelGGL( k_lancir_out_px4< uint8_t >, gp, dim3( 256 ), 0,
				st, res, (uint8_t*) dst, p -> new_w, rstride,
				(long) p -> new_stride, nrows, p -> l_unity, p -> l_out_mul,
				p -> l_clamp );
		}
		else
		{
			hipLaunchKernelGGL( k_lancir_out_px4< uint16_t >, gp, dim3( 256 ), 0,
				st, res, (uint16_t*) dst, p -> new_w, rstride,
				(long) p -> new_stride, nrows, p -> l_unity, p -> l_out_mul,
				p -> l_clamp );
		}

		AVIRHIP_HIPCHECK( hipGetLastError() );
		return( AVIRHIP_OK );
	}

	const int l = p -> new_w * p -> ch;
	const dim3 go(( l + 255 ) / 256, nrows );

#define LO( T, isf ) hipLaunchKernelGGL( k_lancir_out< T >, go, dim3( 256 ), 0, \
		st, res, (T*) dst, l, rstride, (long) p -> new_stride, nrows, \
		p -> l_unity, p -> l_out_mul, p -> l_clamp, isf )

	if( p -> out_type == AVIRHIP_U8 ) { LO( uint8_t, 0 ); }
	else if( p -> out_type == AVIRHIP_U16 ) { LO( uint16_t, 0 ); }
	// (uint32_t: the 4-byte integer stores of lancir.h:1869-1887, range 65535)
	else if( p -> out_type == AVIRHIP_U32 ) { LO( uint32_t, 0 ); }
	// (double: (T) ip[ i ] or (T) ( ip[ i ] * OutMul ), the product in float,
	// lancir.h:1786-1856)
	else if( p -> out_type == AVIRHIP_F64 ) { LO( double, 1 ); }
	else { LO( float, 1 ); }

#undef LO

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

} // namespace avirhip

// Test hook (include/avirhip.h): the uint8 gamma output stage over caller-chosen
// float values, once as the threshold table and once as the direct expressions.
extern "C" int avirhip_debug_gamma_u8( const float* dev_rgba, long npx,
	double tr_mul, int alpha_index, unsigned char* dev_out_table,
	unsigned char* dev_out_direct, void* stream )
try
{
	using namespace avirhip;

	if( dev_rgba == nullptr || dev_out_table == nullptr ||
		dev_out_direct == nullptr || npx < 1 || !( tr_mul >= 1.0 ))
	{
		set_error( "debug_gamma_u8: bad arguments" );
		return( AVIRHIP_EINVAL );
	}

	const int use_tr = ( tr_mul != 1.0 );
	const float trm = (float) tr_mul;
	const float trmi = (float) ( 1.0 / tr_mul );
	float thr[ 512 ];

	if( !gamma_u8_thresholds( 255.0f, use_tr, trm, trmi, 255.0f, thr ))
	{
		set_error( "debug_gamma_u8: the stage is not monotone for these "
			"parameters (plans then keep the direct kernel)" );
		return( AVIRHIP_EUNSUPPORTED );
	}

	float* dthr = nullptr;
	AVIRHIP_HIPCHECK( hipMalloc( (void**) &dthr, sizeof( thr )));
	hipStream_t st = (hipStream_t) stream;
	hipError_t e = hipMemcpyAsync( dthr, thr, sizeof( thr ),
		hipMemcpyHostToDevice, st );
	const dim3 grd( (unsigned) (( npx + 255 ) / 256 ));

	if( e == hipSuccess )
	{
		hipLaunchKernelGGL(( k_epilogue_gamma_thr< 4 > ), grd, dim3( 256 ), 0,
			st, dev_rgba, dev_out_table, npx, alpha_index, dthr, use_tr, trm,
			trmi, 255.0f, 255.0f );
		launch_epilogue_gamma_px< uint8_t >( dev_rgba, dev_out_direct, npx, 4,
			use_tr, trm, trmi, 255.0f, alpha_index, 255.0f, st );
		e = hipGetLastError();
	}

	if( e == hipSuccess )
	{
		e = hipStreamSynchronize( st );
	}

	(void) hipFree( dthr );

	if( e != hipSuccess )
	{
		set_error( "debug_gamma_u8: %s", hipGetErrorString( e ));
		return( AVIRHIP_EHIP );
	}

	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_debug_gamma_u8 )
