// generic64.hip -- the double pipeline: avir::CImageResizer< fpclass_def< double > >
// (avir.h:4553-4560). Every table of such a plan is built in double by the
// planner's double instantiation (planner_avir.inl: the reference stores,
// accumulates and reads back its filters in `fptype`, so the double tables are
// not the float ones widened), and every pass computes in double: one kernel
// launch per lowered op with the intermediates in HBM, as generic.hip does for
// float -- the reference's own example of this class is an accuracy option,
// not a speed one, and no fast path exists for it here either.
//
// Arithmetic contract (compiled -ffp-contract=off): separate v_mul_f64 /
// v_add_f64 in the reference's order, sums started from +0.0 -- bit-identical
// to the reference built the same way (tests: oracle/_ref, variant 4).

#include "plan.h"
#include <string.h>
#include <stdint.h>
#include <type_traits>

namespace avirhip {

namespace {

struct OpArgs64
{
	const double* in; long in_ss, in_is; int in_prefix;
	double* out; long out_ss, out_is; int out_prefix;
	int view, in_len, zs_mmax;
	int ch;
	int scan0, nscan, idx0, nidx;
	int rf, lat, e; const double* flt;
	const int* start; const int* ntaps; const double* coef; int maxtaps;
	int flen, up_inprefix, up_R, sdc_len, pdc_len, pdc_d0;
	const double* sdc; const double* pdc;
};

struct Surf64
{
	double* base;
	long scan_stride;
	long idx_stride;
	int prefix;
};

__device__ __forceinline__ double ldv( const OpArgs64& a, const long so, int i,
	const int c )
{
	if( a.view == VIEW_RAW )
	{
		return( a.in[ so + (long) ( i + a.in_prefix ) * a.in_is + c ]);
	}

	if( a.view == VIEW_ZS && i > a.zs_mmax )
	{
		return( 0.0 );
	}

	i = ( i < 0 ? 0 : ( i >= a.in_len ? a.in_len - 1 : i ));
	return( a.in[ so + (long) i * a.in_is + c ]);
}

// Thread (x, y): XIDX ? (idx, scan) : (scan, idx). x is always along image x.
template< bool XIDX >
__device__ __forceinline__ bool locate( const OpArgs64& a, int& scan, int& idx )
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
__global__ void __launch_bounds__( 256 ) k_fir64( const OpArgs64 a )
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
		double s = a.flt[ 0 ] * ldv( a, so, cp, c );

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
__global__ void __launch_bounds__( 256 ) k_gather64( const OpArgs64 a )
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
	const double* cf = a.coef + (long) j * a.maxtaps;

	for( int c = 0; c < a.ch; c++ )
	{
		double sum = 0.0;

		for( int t = 0; t < nt; t++ )
		{
			sum += cf[ t ] * ldv( a, so, st + t, c );
		}

		a.out[ oo + c ] = sum;
	}
}

// doUpsample with filtering, avir.h:3404-3733 (see generic.hip: k_upf).
template< bool XIDX >
__global__ void __launch_bounds__( 256 ) k_upf64( const OpArgs64 a )
{
	int scan, d;

	if( !locate< XIDX >( a, scan, d ))
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
		double s = 0.0;

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

int launch_op64( const LOp& op, int ch, bool x_is_idx, const Surf64& in,
	const Surf64& out, int scan0, int scan1, int idx0, int idx1,
	hipStream_t st )
{
	OpArgs64 a;
	a.in = in.base; a.in_ss = in.scan_stride; a.in_is = in.idx_stride;
	a.in_prefix = in.prefix;
	a.out = out.base + (long) out.prefix * out.idx_stride;
	a.out_ss = out.scan_stride; a.out_is = out.idx_stride;
	a.out_prefix = out.prefix;
	a.view = op.view; a.in_len = op.in_len; a.zs_mmax = op.zs_mmax;
	a.ch = ch;
	a.scan0 = scan0; a.nscan = scan1 - scan0;
	a.idx0 = idx0; a.nidx = idx1 - idx0;
	a.rf = op.rf; a.lat = op.lat; a.e = op.e; a.flt = op.d_flt64;
	a.start = op.d_start; a.ntaps = op.d_ntaps; a.coef = op.d_coef64;
	a.maxtaps = op.maxtaps;
	a.flen = op.flen; a.up_inprefix = op.up_inprefix; a.up_R = op.up_R;
	a.sdc_len = op.sdc_len; a.pdc_len = op.pdc_len; a.pdc_d0 = op.pdc_d0;
	a.sdc = op.d_sdc64; a.pdc = op.d_pdc64;

	if( op.type == OP_UPF )
	{
		// materialise the whole buffer incl. prefix / suffix
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

	if( op.type == OP_FIR ) { LAUNCH( k_fir64 ); }
	else if( op.type == OP_GATHER ) { LAUNCH( k_gather64 ); }
	else { LAUNCH( k_upf64 ); }

#undef LAUNCH

	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

// pow24_sRGB / pow24i_sRGB, avir.h:161-196: evaluated in double whatever T is
__device__ __forceinline__ double pow24_64( const double x )
{
	const double x2 = x * x;
	const double x3 = x2 * x;
	const double x4 = x2 * x2;

	return( 0.0985766365536824 + 0.839474952656502 * x2 +
		0.363287814061725 * x3 - 0.0125559718896615 /
		( 0.12758338921578 + 0.290283465468235 * x ) -
		0.231757513261358 * x - 0.0395365717969074 * x4 );
}

__device__ __forceinline__ double pow24i_64( const double x )
{
	const double sx = sqrt( x );
	const double ssx = sqrt( sx );
	const double sssx = sqrt( ssx );

	return( 0.000213364515060263 + 0.0149409239419218 * x +
		0.433973412731747 * sx + ssx * ( 0.659628181609715 * sssx -
		0.0380957908841466 - 0.0706476137208521 * sx ));
}

// packScanline, avir.h:2777-2930: the (fptype) cast per element, zero padding
// up to `ech` channels; with gamma the colour channels are linearised
// (convertSRGB2Lin< double, Tin >, avir.h:207-291: uint8 through the table of
// float literals), the alpha channel only scaled by InGammaMult.
template< typename Tin >
__global__ void __launch_bounds__( 256 ) k_pack64( const Tin* src, double* dst,
	int row_elems, int h, long src_stride, int ch, int ech, int gamma,
	int alpha_index, double gm, const float* tbl )
{
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	const int y = blockIdx.y;

	if( x >= row_elems || y >= h )
	{
		return;
	}

	const int px = x / ech;
	const int c = x - px * ech;
	double r = 0.0;

	if( c < ch )
	{
		const Tin v = src[ (long) y * src_stride + px * ch + c ];

		if( !gamma )
		{
			r = (double) v;
		}
		else
		if( c == alpha_index )
		{
			r = (double) v * gm;
		}
		else
		if( sizeof( Tin ) == 1 )
		{
			r = (double) tbl[ (int) v ];
		}
		else
		{
			const double s = (double) v * gm;
			const double a = 0.055;
			r = ( s <= 0.04045 ? s / 12.92 : pow24_64(( s + a ) / ( 1.0 + a )));
		}
	}

	dst[ (long) y * row_elems + x ] = r;
}

// avir::round< double >, avir.h:130-135
__device__ __forceinline__ double round64( const double d )
{
	// (the x86-64 build's cvttsd2si: INT_MIN for what the int cannot hold,
	// NaN included -- plan.h avirhip_x86_cvtt)
	const double a = ( d < 0.0 ? 0.5 - d : d + 0.5 );
	const double r = (double) ( fabs( a ) < 2147483648.0 ? (int) a :
		( -2147483647 - 1 ));

	return( d < 0.0 ? -r : r );
}

// applySRGBGamma (avir.h:2982-3068), CImageResizerDithererDefINL< double >::
// dither (avir.h:4392-4419) for integer outputs, unpackScanline's (Tout) cast
// (avir.h:3155-3215); `res` holds ech >= ch channels per pixel.
template< typename Tout, bool IsInt >
__global__ void __launch_bounds__( 256 ) k_epilogue64( const double* res,
	Tout* dst, long n, int use_tr, double tr_mul, double tr_muli, double pk_out,
	int gamma, int ch, int ech, int alpha_index, double ogm )
{
	const long i = (long) blockIdx.x * blockDim.x + threadIdx.x;

	if( i >= n )
	{
		return;
	}

	const long px = i / ch;
	const int c = (int) ( i - px * ch );
	double v = res[ px * ech + c ];

	if( gamma )
	{
		if( c == alpha_index )
		{
			v = v * ogm;
		}
		else
		{
			const double a = 0.055;
			v = ( v <= 0.0031308 ? 12.92 * v : ( 1.0 + a ) * pow24i_64( v ) - a ) *
				ogm;
		}
	}

	if( IsInt )
	{
		v = ( use_tr ? round64( v * tr_muli ) * tr_mul : round64( v ));
		v = ( v < 0.0 ? 0.0 : ( v > pk_out ? pk_out : v ));
	}

	dst[ i ] = (Tout) v;
}

int dalloc( avirhip_plan* p, const size_t bytes, double** out )
{
	void* q = nullptr;
	AVIRHIP_HIPCHECK( hipMalloc( &q, bytes ));
	p -> allocs.push_back( q );
	p -> alloc_bytes += bytes;
	*out = (double*) q;
	return( AVIRHIP_OK );
}

} // namespace

// Output rows [row0, row1) of a double-pipeline plan: pack (unless the source
// is the double image the first pass can read as it is), the horizontal ops
// over the source rows the band's vertical windows read, the vertical ops,
// then the output stage -- or, for double output, the vertical pass' in-place
// result (avir.h:4956-4979: IsOutFloat && sizeof( fptype ) == sizeof( Tout );
// float output goes through unpackScanline and is de-linearised with gamma).
int exec_f64( avirhip_plan* p, const void* src, void* dst, int row0, int row1,
	hipStream_t st )
{
	if( row1 <= row0 )
	{
		return( AVIRHIP_OK );
	}

	int rc;
	const int ch = p -> ch; // (== io_ch: no RGBA padding here)
	const int nv = (int) p -> v.ops.size();
	const int nh = (int) p -> h.ops.size();
	// the tiled two-pass kernels (tile64.hip) run every plan without a filtered
	// upsample unless path 1 is forced: their H pass reads the caller's image
	// itself and their V pass stores through the output stage, so without gamma
	// neither the double copy of the source nor the double result exists
	const int xpath = ( p -> path != 0 ? p -> path : p -> auto_path );
	const bool tiled = ( xpath != 1 && tile64_ok( p ));
	const bool need_pack = ( tiled ? p -> gamma != 0 :
		( p -> gamma || p -> in_type != AVIRHIP_F64 ));
	const bool direct = ( p -> out_type == AVIRHIP_F64 ||
		( tiled && !p -> gamma ));

	if( need_pack && p -> packed64 == nullptr &&
		( rc = dalloc( p, (size_t) p -> src_w * p -> src_h * ch *
		sizeof( double ), &p -> packed64 )) != 0 ) return( rc );

	if( !direct && p -> resbuf64 == nullptr &&
		( rc = dalloc( p, (size_t) p -> new_w * p -> new_h * ch *
		sizeof( double ), &p -> resbuf64 )) != 0 ) return( rc );

	if( !tiled && p -> hbuf64.empty() )
	{
		for( int i = 0; i < nh; i++ )
		{
			double* q;

			if(( rc = dalloc( p, (size_t) p -> h.ops[ i ].out_total * p -> src_h *
				ch * sizeof( double ), &q )) != 0 ) return( rc );

			p -> hbuf64.push_back( q );
		}

		for( int i = 0; i + 1 < nv; i++ )
		{
			double* q;

			if(( rc = dalloc( p, (size_t) p -> v.ops[ i ].out_total * p -> new_w *
				ch * sizeof( double ), &q )) != 0 ) return( rc );

			p -> vbuf64.push_back( q );
		}
	}

	// backward range propagation through the vertical chain
	std::vector< int > va( nv ), vb( nv );
	int a = row0, b = row1 - 1;

	for( int i = nv - 1; i >= 0; i-- )
	{
		const LOp& op = p -> v.ops[ i ];
		va[ i ] = a;
		vb[ i ] = b;

		if( op.type == OP_UPF )
		{
			va[ i ] = -op.out_prefix;
			vb[ i ] = op.out_total - op.out_prefix - 1;
		}

		int ia, ib;
		need_range( op, va[ i ], vb[ i ], ia, ib );
		a = ia;
		b = ib;
	}

	const int ya = a, yb = b; // FltBuf rows needed

	const double* fsrc = (const double*) src;
	long sstride = p -> src_stride;

	if( need_pack )
	{
		const int re = p -> src_w * ch;
		const int rows = yb - ya + 1;
		const dim3 grd(( re + 255 ) / 256, rows );
		// InGammaMult, avir.h:4744-4754
		const double gm = ( p -> in_type == AVIRHIP_U8 ? 1.0 / 255.0 :
			( p -> in_type == AVIRHIP_U16 ? 1.0 / 65535.0 : 1.0 ));

		double* const pd = p -> packed64 + (size_t) ya * re;

#define PK( T ) hipLaunchKernelGGL( k_pack64< T >, grd, dim3( 256 ), 0, st, \
		(const T*) src + (size_t) ya * p -> src_stride, pd, re, rows, \
		(long) p -> src_stride, ch, ch, p -> gamma, p -> alpha_index, gm, \
		p -> d_srgb_tbl )

		switch( p -> in_type )
		{
			case AVIRHIP_U8: PK( uint8_t ); break;
			case AVIRHIP_U16: PK( uint16_t ); break;
			case AVIRHIP_F32: PK( float ); break;
			default: PK( double ); break;
		}

#undef PK
		AVIRHIP_HIPCHECK( hipGetLastError() );
		fsrc = p -> packed64;
		sstride = re;
	}

	if( tiled )
	{
		void* const tdst = ( direct ? dst : (void*) p -> resbuf64 );

		rc = tile64_run( p, fsrc, ( need_pack ? AVIRHIP_F64 : p -> in_type ),
			sstride, tdst, ( p -> gamma && p -> out_type != AVIRHIP_F64 ?
			AVIRHIP_F64 : p -> out_type ), (long) p -> new_w * ch, row0, row1,
			ya, yb, st );

		if( rc != 0 || direct )
		{
			return( rc );
		}
	}

	Surf64 in;
	in.base = (double*) fsrc; in.scan_stride = sstride; in.idx_stride = ch;
	in.prefix = 0;

	for( int i = 0; !tiled && i < nh; i++ )
	{
		const LOp& op = p -> h.ops[ i ];
		Surf64 out;
		out.base = p -> hbuf64[ i ];
		out.scan_stride = (long) op.out_total * ch;
		out.idx_stride = ch;
		out.prefix = op.out_prefix;

		if(( rc = launch_op64( op, ch, true, in, out, ya, yb + 1, 0,
			op.out_len, st )) != 0 ) return( rc );

		in = out;
	}

	// vertical pass: scanlines are the NewWidth columns of FltBuf
	double* const fdst = ( direct ? (double*) dst : p -> resbuf64 );
	in.base = ( tiled ? nullptr : p -> hbuf64[ nh - 1 ]);
	in.scan_stride = ch;
	in.idx_stride = (long) p -> new_w * ch;
	in.prefix = 0;

	for( int i = 0; !tiled && i < nv; i++ )
	{
		const LOp& op = p -> v.ops[ i ];
		Surf64 out;
		out.scan_stride = ch;
		out.idx_stride = (long) p -> new_w * ch;
		out.prefix = op.out_prefix;
		out.base = ( i == nv - 1 ? fdst - (long) row0 * out.idx_stride :
			p -> vbuf64[ i ]);

		if(( rc = launch_op64( op, ch, false, in, out, 0, p -> new_w, va[ i ],
			vb[ i ] + 1, st )) != 0 ) return( rc );

		in = out;
	}

	if( direct )
	{
		return( AVIRHIP_OK );
	}

	const long n = (long) ( row1 - row0 ) * p -> new_w * ch;
	const dim3 grd( (unsigned) (( n + 255 ) / 256 ));
	const int use_tr = ( p -> tr_mul != 1.0 );
	// OutGammaMult, avir.h:4756-4763
	const double ogm = ( p -> out_type == AVIRHIP_U8 ? 255.0 :
		( p -> out_type == AVIRHIP_U16 ? 65535.0 : 1.0 ));

#define EP( T, INT ) hipLaunchKernelGGL(( k_epilogue64< T, INT > ), grd, \
	dim3( 256 ), 0, st, (const double*) fdst, (T*) dst, n, use_tr, p -> tr_mul, \
	1.0 / p -> tr_mul, p -> pk_out, p -> gamma, ch, ch, p -> alpha_index, ogm )

	switch( p -> out_type )
	{
		case AVIRHIP_U8: EP( uint8_t, true ); break;
		case AVIRHIP_U16: EP( uint16_t, true ); break;
		default: EP( float, false ); break;
	}

#undef EP
	AVIRHIP_HIPCHECK( hipGetLastError() );
	return( AVIRHIP_OK );
}

} // namespace avirhip
