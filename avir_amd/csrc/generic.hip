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
template< typename Tin >
__global__ void __launch_bounds__( 256 ) k_pack( const Tin* src, float* dst,
	int row_elems, int h, long src_stride )
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	const int y = blockIdx.y;

	if( x < row_elems && y < h )
	{
		dst[ (long) y * row_elems + x ] =
			(float) src[ (long) y * src_stride + x ];
	}
}

int launch_pack( const void* src, int in_type, float* dst, int w, int h,
	int ch, long src_stride, hipStream_t st )
{
	const int re = w * ch;
	const dim3 grd(( re + 255 ) / 256, h );

	if( in_type == AVIRHIP_U8 )
		hipLaunchKernelGGL( k_pack< uint8_t >, grd, dim3( 256 ), 0, st,
			(const uint8_t*) src, dst, re, h, src_stride );
	else if( in_type == AVIRHIP_U16 )
		hipLaunchKernelGGL( k_pack< uint16_t >, grd, dim3( 256 ), 0, st,
			(const uint16_t*) src, dst, re, h, src_stride );
	else if( in_type == AVIRHIP_F64 )
		hipLaunchKernelGGL( k_pack< double >, grd, dim3( 256 ), 0, st,
			(const double*) src, dst, re, h, src_stride );
	else
		hipLaunchKernelGGL( k_pack< float >, grd, dim3( 256 ), 0, st,
			(const float*) src, dst, re, h, src_stride );

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

// avir::round, avir.h:130-135 (biased, via int cast).
__device__ __forceinline__ float avir_round( const float d )
{
	return( d < 0.0f ? -(float) (int) ( 0.5f - d ) :
		(float) (int) ( d + 0.5f ));
}

// CImageResizerDithererDefINL::dither (avir.h:4392-4419) + unpackScanline
// (avir.h:3155-3215): round, clamp, optional bit-depth truncation, C cast.
template< typename Tout, bool IsInt >
__global__ void __launch_bounds__( 256 ) k_epilogue( const float* res,
	Tout* dst, long n, int use_tr, float tr_mul, float tr_muli, float pk_out )
{
	const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;

	if( i >= n )
	{
		return;
	}

	float v = res[ i ];

	if( IsInt )
	{
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

int launch_epilogue( const float* res, void* dst, int out_type, long n,
	double tr_mul, double pk_out, hipStream_t st )
{
	const dim3 grd( (unsigned) (( n + 255 ) / 256 ));
	const int use_tr = ( tr_mul != 1.0 );
	const float trm = (float) tr_mul;
	const float trmi = (float) ( 1.0 / tr_mul );
	const float pk = (float) pk_out;

	if( out_type == AVIRHIP_U8 )
		hipLaunchKernelGGL(( k_epilogue< uint8_t, true > ), grd, dim3( 256 ),
			0, st, res, (uint8_t*) dst, n, use_tr, trm, trmi, pk );
	else if( out_type == AVIRHIP_U16 )
		hipLaunchKernelGGL(( k_epilogue< uint16_t, true > ), grd, dim3( 256 ),
			0, st, res, (uint16_t*) dst, n, use_tr, trm, trmi, pk );
	else if( out_type == AVIRHIP_F64 )
		hipLaunchKernelGGL(( k_epilogue< double, false > ), grd, dim3( 256 ),
			0, st, res, (double*) dst, n, use_tr, trm, trmi, pk );
	else
		hipLaunchKernelGGL(( k_epilogue< float, false > ), grd, dim3( 256 ),
			0, st, res, (float*) dst, n, use_tr, trm, trmi, pk );

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

// ---------------------------------------------------------------------
// LANCIR generic kernels (lancir.h:541-710): vertical pass first into a float
// intermediate [rows][SrcWidth], then the horizontal pass. 4-channel x86
// summation order (resize4, lancir.h:2466-2544): even taps and odd taps are
// accumulated separately (first term is the bare product) and added last.
// ---------------------------------------------------------------------

template< typename Tin >
__global__ void __launch_bounds__( 256 ) k_lancir_v4( const Tin* src,
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
	float ea[ 4 ], oa[ 4 ];

	for( int t = 0; t < kl; t += 2 )
	{
		int y0 = st + t;
		int y1 = st + t + 1;
		y0 = ( y0 < 0 ? 0 : ( y0 >= sh ? sh - 1 : y0 ));
		y1 = ( y1 < 0 ? 0 : ( y1 >= sh ? sh - 1 : y1 ));
		const Tin* p0 = src + (long) y0 * src_stride + (long) x * 4;
		const Tin* p1 = src + (long) y1 * src_stride + (long) x * 4;

		for( int c = 0; c < 4; c++ )
		{
			const float e = f[ t ] * (float) p0[ c ];
			const float o = f[ t + 1 ] * (float) p1[ c ];

			if( t == 0 )
			{
				ea[ c ] = e;
				oa[ c ] = o;
			}
			else
			{
				ea[ c ] += e;
				oa[ c ] += o;
			}
		}
	}

	float* op = tmp + ( (long) ry * sw + x ) * 4;

	for( int c = 0; c < 4; c++ )
	{
		op[ c ] = ea[ c ] + oa[ c ];
	}
}

__global__ void __launch_bounds__( 256 ) k_lancir_h4( const float* tmp,
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
	const float* row = tmp + (long) ry * sw * 4;
	float ea[ 4 ], oa[ 4 ];

	for( int t = 0; t < kl; t += 2 )
	{
		int x0 = st + t;
		int x1 = st + t + 1;
		x0 = ( x0 < 0 ? 0 : ( x0 >= sw ? sw - 1 : x0 ));
		x1 = ( x1 < 0 ? 0 : ( x1 >= sw ? sw - 1 : x1 ));

		for( int c = 0; c < 4; c++ )
		{
			const float e = f[ t ] * row[ x0 * 4 + c ];
			const float o = f[ t + 1 ] * row[ x1 * 4 + c ];

			if( t == 0 )
			{
				ea[ c ] = e;
				oa[ c ] = o;
			}
			else
			{
				ea[ c ] += e;
				oa[ c ] += o;
			}
		}
	}

	float* op = res + (long) ry * res_stride + (long) j * 4;

	for( int c = 0; c < 4; c++ )
	{
		op[ c ] = ea[ c ] + oa[ c ];
	}
}

int launch_lancir_generic( const avirhip_plan* p, const void* src, void* dst,
	float* tmp, int row0, int row1, hipStream_t st )
{
	const int nrows = row1 - row0;

	if( nrows <= 0 )
	{
		return( AVIRHIP_OK );
	}

	const LancirAxisDev& V = p -> lv;
	const LancirAxisDev& H = p -> lh;
	const dim3 gv(( p -> src_w + 255 ) / 256, nrows );

	if( p -> in_type == AVIRHIP_U8 )
		hipLaunchKernelGGL( k_lancir_v4< uint8_t >, gv, dim3( 256 ), 0, st,
			(const uint8_t*) src, tmp, p -> src_w, p -> src_h,
			(long) p -> src_stride, row0, nrows, V.d_start, V.d_fidx, V.d_flt,
			V.kernel_len );
	else if( p -> in_type == AVIRHIP_U16 )
		hipLaunchKernelGGL( k_lancir_v4< uint16_t >, gv, dim3( 256 ), 0, st,
			(const uint16_t*) src, tmp, p -> src_w, p -> src_h,
			(long) p -> src_stride, row0, nrows, V.d_start, V.d_fidx, V.d_flt,
			V.kernel_len );
	else
		hipLaunchKernelGGL( k_lancir_v4< float >, gv, dim3( 256 ), 0, st,
			(const float*) src, tmp, p -> src_w, p -> src_h,
			(long) p -> src_stride, row0, nrows, V.d_start, V.d_fidx, V.d_flt,
			V.kernel_len );

	AVIRHIP_HIPCHECK( hipGetLastError() );

	const dim3 gh(( p -> new_w + 255 ) / 256, nrows );
	hipLaunchKernelGGL( k_lancir_h4, gh, dim3( 256 ), 0, st, tmp,
		(float*) dst, p -> src_w, p -> new_w, (long) p -> new_stride, nrows,
		H.d_start, H.d_fidx, H.d_flt, H.kernel_len );

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

} // namespace avirhip
