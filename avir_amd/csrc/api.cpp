// api.cpp -- the C ABI of libavirhip (include/avirhip.h): plan validation and
// lowering, device upload, scratch management and pass orchestration. The
// arithmetic lives in generic.hip / fused.hip; the host-side planner mirror
// (avirhip_resizer_*, avirhip_lancir_*) lives in planner.cpp.

#include "plan.h"
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <atomic>
#include <thread>
#include <system_error>
#include <new>
#include <stdexcept>

namespace avirhip {

static thread_local char g_err[ 512 ] = "";

void set_error( const char* fmt, ... )
{
	va_list ap;
	va_start( ap, fmt );
	vsnprintf( g_err, sizeof( g_err ), fmt, ap );
	va_end( ap );
}

// (every extern "C" entry point begins with this: the message belongs to the
// calling thread's most recent call, never to an older failure)
void clear_error()
{
	g_err[ 0 ] = 0;
}

size_t dtype_size( int t )
{
	return( t == AVIRHIP_U8 ? 1 : t == AVIRHIP_U16 ? 2 : t == AVIRHIP_F32 ? 4 :
		t == AVIRHIP_U32 ? 4 : 8 );
}

int guard_fail( const char* fn ) noexcept
{
	// (called from a catch( ... ) handler: rethrow to see what is in flight)
	try
	{
		throw;
	}
	catch( const std::bad_alloc& )
	{
		set_error( "%s: out of host memory", fn );
		return( AVIRHIP_ENOMEM );
	}
	catch( const std::length_error& e )
	{
		// (a container asked for more than max_size(): the same condition)
		set_error( "%s: out of host memory (%s)", fn, e.what() );
		return( AVIRHIP_ENOMEM );
	}
	catch( const std::exception& e )
	{
		set_error( "%s: internal error: %s", fn, e.what() );
	}
	catch( ... )
	{
		set_error( "%s: internal error (unknown exception)", fn );
	}

	return( AVIRHIP_EINTERNAL );
}

bool mul_fits( size_t a, size_t b, size_t c, size_t d, size_t* out )
{
	size_t r = 0;

	if( __builtin_mul_overflow( a, b, &r ) || __builtin_mul_overflow( r, c, &r ) ||
		__builtin_mul_overflow( r, d, &r ))
	{
		return( false );
	}

	if( out != nullptr )
	{
		*out = r;
	}

	return( true );
}

bool geometry_ok( const char* fn, int src_w, int src_h, long src_stride,
	int new_w, int new_h, long new_stride, int ch, int in_type, int out_type )
{
	if( src_w < 0 || src_h < 0 || new_w < 0 || new_h < 0 || ch < 1 || ch > 4 )
	{
		set_error( "%s: bad image geometry", fn );
		return( false );
	}

	const long INTMAX = 0x7fffffffL;
	const long se = (long) src_w * ch, ne = (long) new_w * ch;

	if( src_stride < 1 ) src_stride = se;
	if( new_stride < 1 ) new_stride = ne;

	// a float copy of either image (16 bytes per RGBA pixel) is the largest
	// buffer a plan may allocate per pixel
	if( se > INTMAX || ne > INTMAX || src_stride > INTMAX || new_stride > INTMAX ||
		!mul_fits( (size_t) src_stride, (size_t) src_h, 16, 1 ) ||
		!mul_fits( (size_t) new_stride, (size_t) new_h, 16, 1 ) ||
		!mul_fits( (size_t) src_w, (size_t) new_h, 16 * ch, 1 ) ||
		!mul_fits( (size_t) new_w, (size_t) src_h, 16 * ch, 1 ) ||
		dtype_size( in_type ) == 0 || dtype_size( out_type ) == 0 )
	{
		set_error( "%s: image of %d x %d x %d -> %d x %d elements does not fit "
			"int rows / size_t bytes", fn, src_w, src_h, ch, new_w, new_h );
		return( false );
	}

	return( true );
}

PlanHold :: ~PlanHold()
{
	if( p != nullptr )
	{
		avirhip_plan_destroy( p );
	}
}

template< typename T >
static int upload( avirhip_plan* p, const std::vector< T >& h, T** d )
{
	*d = nullptr;

	if( h.empty() )
	{
		return( AVIRHIP_OK );
	}

	void* q = nullptr;
	AVIRHIP_HIPCHECK( hipMalloc( &q, h.size() * sizeof( T )));
	p -> allocs.push_back( q );
	p -> alloc_bytes += h.size() * sizeof( T );
	AVIRHIP_HIPCHECK( hipMemcpy( q, h.data(), h.size() * sizeof( T ),
		hipMemcpyHostToDevice ));

	*d = (T*) q;
	return( AVIRHIP_OK );
}

static int dev_alloc( avirhip_plan* p, size_t bytes, void** out )
{
	void* q = nullptr;
	AVIRHIP_HIPCHECK( hipMalloc( &q, bytes ));
	p -> allocs.push_back( q );
	p -> alloc_bytes += bytes;
	*out = q;
	return( AVIRHIP_OK );
}

// ---- lowering of one AVIR axis (see plan.h for the view semantics) ----

static int lower_axis( const avirhip_axis& ax, int src_len, int dst_len,
	LAxis& L, const bool f64 = false )
{
	L.ops.clear();
	L.src_len = src_len;
	L.dst_len = dst_len;

	if( ax.n_steps < 1 || ax.steps == nullptr )
	{
		set_error( "axis has no steps" );
		return( AVIRHIP_EINVAL );
	}

	int cur_len = src_len;
	bool zs = false;      // a zero-stuff view is pending for the next step
	int zs_in_len = 0, zs_mmax = 0, zs_prefix = 0;
	bool prev_upf = false;
	int prev_prefix = 0, prev_total = 0;

	for( int si = 0; si < ax.n_steps; si++ )
	{
		const avirhip_step& s = ax.steps[ si ];

		if( s.in_len != cur_len )
		{
			set_error( "step %d: in_len %d != previous out_len %d", si,
				s.in_len, cur_len );
			return( AVIRHIP_EINVAL );
		}

		if( zs && s.kind != AVIRHIP_STEP_RESIZE2 )
		{
			set_error( "step %d: zero-stuff upsample must be followed by "
				"RESIZE2", si );
			return( AVIRHIP_EUNSUPPORTED );
		}

		LOp op;
		op.type = 0; op.view = ( prev_upf ? VIEW_RAW : VIEW_CLAMP );
		op.in_len = s.in_len; op.in_prefix = ( prev_upf ? prev_prefix : 0 );
		op.zs_mmax = 0x7fffffff;
		op.out_len = s.out_len; op.out_prefix = 0; op.out_total = s.out_len;
		op.rf = 1; op.lat = 0; op.e = 0; op.d_flt = nullptr;
		op.d_flt64 = nullptr; op.d_coef64 = nullptr; op.d_sdc64 = nullptr;
		op.d_pdc64 = nullptr;
		op.maxtaps = 0; op.d_start = nullptr; op.d_ntaps = nullptr;
		op.d_coef = nullptr;
		op.flen = 0; op.up_inprefix = 0; op.up_R = 0; op.sdc_len = 0;
		op.pdc_len = 0; op.pdc_d0 = 0; op.d_sdc = nullptr; op.d_pdc = nullptr;

		if( s.kind == AVIRHIP_STEP_FIR )
		{
			if( s.resample_factor < 1 ||
				( f64 ? (const void*) s.flt64 : (const void*) s.flt ) == nullptr ||
				s.flt_latency < 0 || s.flt_len != 2 * s.flt_latency + 1 )
			{
				set_error( "step %d: malformed FIR (len %d, latency %d)", si,
					s.flt_len, s.flt_latency );
				return( AVIRHIP_EINVAL );
			}

			op.type = OP_FIR;
			op.rf = s.resample_factor;
			op.lat = s.flt_latency;
			op.e = s.edge_pixel_count;

			if( f64 )
			{
				op.h_flt64.assign( s.flt64 + s.flt_latency, s.flt64 + s.flt_len );
			}
			else
			{
				op.h_flt.assign( s.flt + s.flt_latency, s.flt + s.flt_len );
			}

			if( prev_upf )
			{
				// reads in[rf*(n-e) +- lat] of the raw upsample buffer
				const int lo = -op.e * op.rf - op.lat + prev_prefix;
				const int hi = op.rf * ( s.out_len - 1 - op.e ) + op.lat +
					prev_prefix;

				if( lo < 0 || hi >= prev_total )
				{
					set_error( "step %d: FIR reads outside upsample buffer",
						si );
					return( AVIRHIP_EINVAL );
				}
			}

			L.ops.push_back( op );
			prev_upf = false;
		}
		else
		if( s.kind == AVIRHIP_STEP_UP_ZEROSTUFF )
		{
			if( s.resample_factor != 2 || prev_upf )
			{
				set_error( "step %d: zero-stuff upsample supports factor 2 "
					"only", si );
				return( AVIRHIP_EUNSUPPORTED );
			}

			zs = true;
			zs_in_len = s.in_len;
			zs_mmax = s.in_len - 1 + s.out_suffix / 2;
			zs_prefix = s.out_prefix;
		}
		else
		if( s.kind == AVIRHIP_STEP_UP_FILTERED )
		{
			if( s.resample_factor < 2 ||
				( f64 ? (const void*) s.flt64 : (const void*) s.flt ) == nullptr ||
				s.flt_len < 1 || prev_upf )
			{
				set_error( "step %d: malformed filtered upsample", si );
				return( AVIRHIP_EINVAL );
			}

			op.type = OP_UPF;
			op.view = VIEW_CLAMP;
			op.rf = s.resample_factor;
			op.flen = s.flt_len;

			if( f64 )
			{
				op.h_flt64.assign( s.flt64, s.flt64 + s.flt_len );
			}
			else
			{
				op.h_flt.assign( s.flt, s.flt + s.flt_len );
			}

			op.up_inprefix = s.in_prefix;
			op.up_R = s.in_prefix + s.in_len + s.in_suffix;
			op.sdc_len = s.suffix_dc_len;
			op.pdc_len = s.prefix_dc_len;
			op.pdc_d0 = s.out_prefix - s.in_prefix * s.resample_factor;
			op.out_prefix = s.out_prefix;
			op.out_total = s.out_prefix + s.out_len + s.out_suffix;

			if( op.pdc_d0 < 0 ||
				op.pdc_d0 + op.pdc_len > op.out_total ||
				op.up_R * op.rf + op.sdc_len > op.out_total ||
				( op.up_R - 1 ) * op.rf + op.flen > op.out_total )
			{
				set_error( "step %d: upsample tails exceed buffer", si );
				return( AVIRHIP_EINVAL );
			}

			// stash DC tails behind the filter taps: [flt | sdc | pdc]
			if( f64 )
			{
				if( s.suffix_dc_len > 0 )
					op.h_flt64.insert( op.h_flt64.end(), s.suffix_dc64,
						s.suffix_dc64 + s.suffix_dc_len );
				if( s.prefix_dc_len > 0 )
					op.h_flt64.insert( op.h_flt64.end(), s.prefix_dc64,
						s.prefix_dc64 + s.prefix_dc_len );
			}
			else
			{
				if( s.suffix_dc_len > 0 )
					op.h_flt.insert( op.h_flt.end(), s.suffix_dc,
						s.suffix_dc + s.suffix_dc_len );
				if( s.prefix_dc_len > 0 )
					op.h_flt.insert( op.h_flt.end(), s.prefix_dc,
						s.prefix_dc + s.prefix_dc_len );
			}

			L.ops.push_back( op );
			prev_upf = true;
			prev_prefix = op.out_prefix;
			prev_total = op.out_total;
		}
		else
		if( s.kind == AVIRHIP_STEP_RESIZE || s.kind == AVIRHIP_STEP_RESIZE2 )
		{
			const bool r2 = ( s.kind == AVIRHIP_STEP_RESIZE2 );

			if( r2 != zs )
			{
				set_error( "step %d: RESIZE2 needs a preceding zero-stuff "
					"upsample (and RESIZE must not have one)", si );
				return( AVIRHIP_EUNSUPPORTED );
			}

			if( s.rpos == nullptr || ( f64 ? (const void*) s.phase_taps64 :
				(const void*) s.phase_taps ) == nullptr ||
				s.bank_filter_len < 2 || s.n_phases < 1 ||
				( s.bank_order != 0 && s.bank_order != 1 ))
			{
				set_error( "step %d: malformed resize step", si );
				return( AVIRHIP_EINVAL );
			}

			const int fl = s.bank_filter_len;
			const int fsz = fl * ( s.bank_order + 1 );
			op.type = OP_GATHER;
			op.maxtaps = ( r2 ? ( fl + 1 ) / 2 : fl );
			op.h_start.resize( s.out_len );
			op.h_ntaps.resize( s.out_len );

			if( f64 )
			{
				op.h_coef64.assign( (size_t) s.out_len * op.maxtaps, 0.0 );
			}
			else
			{
				op.h_coef.assign( (size_t) s.out_len * op.maxtaps, 0.0f );
			}

			if( r2 )
			{
				op.view = VIEW_ZS;
				op.in_len = zs_in_len;
				op.zs_mmax = zs_mmax;
			}

			for( int j = 0; j < s.out_len; j++ )
			{
				const avirhip_rpos& rp = s.rpos[ j ];

				if( rp.phase < 0 || rp.phase >= s.n_phases || rp.ftp_off < 0 ||
					rp.fl < 1 || rp.ftp_off + rp.fl > fl )
				{
					set_error( "step %d: bad rpos[%d]", si, j );
					return( AVIRHIP_EINVAL );
				}

				const int stp = ( r2 ? 2 : 1 );
				int nt = 0;

				if( f64 )
				{
					// (the same in double: xx = ftp[i] + ftp2[i]*x)
					const double* ftp = s.phase_taps64 + (size_t) rp.phase * fsz +
						rp.ftp_off;
					const double* ftp2 = ftp + fl;
					double* cf = &op.h_coef64[ (size_t) j * op.maxtaps ];

					for( int i = 0; i < rp.fl; i += stp )
					{
						if( s.bank_order == 1 )
						{
							const double t = ftp2[ i ] * rp.x64;
							cf[ nt ] = ftp[ i ] + t;
						}
						else
						{
							cf[ nt ] = ftp[ i ];
						}

						nt++;
					}
				}

				const float* ftp = ( f64 ? nullptr : s.phase_taps +
					(size_t) rp.phase * fsz + rp.ftp_off );
				const float* ftp2 = ( f64 ? nullptr : ftp + fl );
				float* cf = ( f64 ? nullptr :
					&op.h_coef[ (size_t) j * op.maxtaps ]);

				for( int i = 0; !f64 && i < rp.fl; i += stp )
				{
					// xx = ftp[i] + ftp2[i]*x in float (avir.h:3945, 4177);
					// this TU is compiled -ffp-contract=off.
					if( s.bank_order == 1 )
					{
						const float t = ftp2[ i ] * rp.x;
						cf[ nt ] = ftp[ i ] + t;
					}
					else
					{
						cf[ nt ] = ftp[ i ];
					}

					nt++;
				}

				op.h_ntaps[ j ] = nt;
				int lo, hi;

				if( r2 )
				{
					if( rp.src_offs_px & 1 )
					{
						set_error( "step %d: odd RESIZE2 offset at %d", si,
							j );
						return( AVIRHIP_EINVAL );
					}

					op.h_start[ j ] = rp.src_offs_px / 2;
					lo = op.h_start[ j ];
					hi = lo + nt - 1;

					if( 2 * lo < -zs_prefix )
					{
						set_error( "step %d: RESIZE2 reads before the "
							"upsample buffer at %d", si, j );
						return( AVIRHIP_EINVAL );
					}
				}
				else
				{
					op.h_start[ j ] = rp.src_offs_px;
					lo = rp.src_offs_px;
					hi = lo + nt - 1;

					if( prev_upf &&
						( lo + prev_prefix < 0 || hi + prev_prefix >= prev_total ))
					{
						set_error( "step %d: RESIZE reads outside the "
							"upsample buffer at %d", si, j );
						return( AVIRHIP_EINVAL );
					}
				}
			}

			L.ops.push_back( op );
			prev_upf = false;
			zs = false;
		}
		else
		{
			set_error( "step %d: unknown kind %d", si, s.kind );
			return( AVIRHIP_EINVAL );
		}

		cur_len = s.out_len;
	}

	if( zs || prev_upf || L.ops.empty() || cur_len != dst_len )
	{
		set_error( "axis does not end in a filtering/resizing step of length "
			"%d (got %d)", dst_len, cur_len );
		return( AVIRHIP_EINVAL );
	}

	return( AVIRHIP_OK );
}

static int upload_axis( avirhip_plan* p, LAxis& L )
{
	for( size_t i = 0; i < L.ops.size(); i++ )
	{
		LOp& op = L.ops[ i ];
		int rc;

		if(( rc = upload( p, op.h_flt, &op.d_flt )) != 0 ) return( rc );
		if(( rc = upload( p, op.h_start, &op.d_start )) != 0 ) return( rc );
		if(( rc = upload( p, op.h_ntaps, &op.d_ntaps )) != 0 ) return( rc );
		if(( rc = upload( p, op.h_coef, &op.d_coef )) != 0 ) return( rc );
		if(( rc = upload( p, op.h_flt64, &op.d_flt64 )) != 0 ) return( rc );
		if(( rc = upload( p, op.h_coef64, &op.d_coef64 )) != 0 ) return( rc );

		if( op.type == OP_UPF )
		{
			op.d_sdc = op.d_flt + op.flen;
			op.d_pdc = op.d_sdc + op.sdc_len;
			op.d_sdc64 = op.d_flt64 + op.flen;
			op.d_pdc64 = op.d_sdc64 + op.sdc_len;
		}
	}

	return( AVIRHIP_OK );
}

// Needed logical input range of `op` for logical outputs [a, b].
void need_range( const LOp& op, int a, int b, int& ia, int& ib )
{
	if( op.type == OP_FIR )
	{
		ia = op.rf * ( a - op.e ) - op.lat;
		ib = op.rf * ( b - op.e ) + op.lat;
	}
	else
	if( op.type == OP_GATHER )
	{
		ia = 0x7fffffff;
		ib = -0x7fffffff;

		for( int j = a; j <= b; j++ )
		{
			ia = std::min( ia, op.h_start[ j ]);
			ib = std::max( ib, op.h_start[ j ] + op.h_ntaps[ j ] - 1 );
		}
	}
	else
	{
		ia = 0;
		ib = op.in_len - 1;
		return;
	}

	if( op.view == VIEW_RAW )
	{
		return; // logical indices of the upsample buffer (may be negative)
	}

	ia = std::max( 0, std::min( ia, op.in_len - 1 ));
	ib = std::max( 0, std::min( ib, op.in_len - 1 ));
}

static int ensure_scratch( avirhip_plan* p )
{
	if( p -> is_lancir )
	{
		if( p -> resbuf == nullptr )
		{
			void* q;
			int rc = dev_alloc( p, (size_t) p -> new_h * p -> src_w *
				p -> ch * sizeof( float ), &q );
			if( rc != 0 ) return( rc );
			p -> resbuf = (float*) q;
		}

		if( p -> lres == nullptr &&
			!( p -> out_type == AVIRHIP_F32 && p -> l_unity ))
		{
			void* q;
			int rc = dev_alloc( p, (size_t) p -> new_h * p -> new_w *
				p -> ch * sizeof( float ), &q );
			if( rc != 0 ) return( rc );
			p -> lres = (float*) q;
		}

		return( AVIRHIP_OK );
	}

	int rc;
	void* q;

	if(( p -> in_type != AVIRHIP_F32 || p -> gamma || p -> ch != p -> io_ch ) &&
		p -> packed == nullptr )
	{
		if(( rc = dev_alloc( p, (size_t) p -> src_w * p -> src_h * p -> ch *
			sizeof( float ), &q )) != 0 ) return( rc );
		p -> packed = (float*) q;
	}

	if(( p -> out_type != AVIRHIP_F32 || p -> ch != p -> io_ch ||
		( p -> fp4 && p -> gamma )) && p -> resbuf == nullptr )
	{
		if(( rc = dev_alloc( p, (size_t) p -> new_w * p -> new_h * p -> ch *
			sizeof( float ), &q )) != 0 ) return( rc );
		p -> resbuf = (float*) q;
	}

	return( AVIRHIP_OK );
}

static int ensure_generic_scratch( avirhip_plan* p )
{
	int rc;
	void* q;

	if( p -> hbuf.empty() )
	{
		for( size_t i = 0; i < p -> h.ops.size(); i++ )
		{
			if(( rc = dev_alloc( p, (size_t) p -> h.ops[ i ].out_total *
				p -> src_h * p -> ch * sizeof( float ), &q )) != 0 )
				return( rc );
			p -> hbuf.push_back( (float*) q );
		}

		for( size_t i = 0; i + 1 < p -> v.ops.size(); i++ )
		{
			if(( rc = dev_alloc( p, (size_t) p -> v.ops[ i ].out_total *
				p -> new_w * p -> ch * sizeof( float ), &q )) != 0 )
				return( rc );
			p -> vbuf.push_back( (float*) q );
		}
	}

	return( AVIRHIP_OK );
}

// Runs the generic chain for output rows [row0, row1) into `dst` (a float
// surface whose row `row0` is at dst).
static int run_generic( avirhip_plan* p, const float* src, long src_stride,
	float* dst, int row0, int row1, hipStream_t st )
{
	int rc = ensure_generic_scratch( p );
	if( rc != 0 ) return( rc );

	const int ch = p -> ch;
	const int nv = (int) p -> v.ops.size();
	const int nh = (int) p -> h.ops.size();

	// Backward range propagation through the vertical chain.
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

	// Horizontal pass over source rows [ya, yb].
	Surf in;
	in.base = (float*) src; in.scan_stride = src_stride; in.idx_stride = ch;
	in.prefix = 0;

	for( int i = 0; i < nh; i++ )
	{
		const LOp& op = p -> h.ops[ i ];
		Surf out;
		out.base = p -> hbuf[ i ];
		out.scan_stride = (long) op.out_total * ch;
		out.idx_stride = ch;
		out.prefix = op.out_prefix;

		if(( rc = launch_op( op, ch, true, in, out, ya, yb + 1, 0,
			op.out_len, st )) != 0 ) return( rc );

		in = out;
	}

	// Vertical pass: scanlines are the NewWidth columns of FltBuf.
	in.base = p -> hbuf[ nh - 1 ];
	in.scan_stride = ch;
	in.idx_stride = (long) p -> new_w * ch;
	in.prefix = 0;

	for( int i = 0; i < nv; i++ )
	{
		const LOp& op = p -> v.ops[ i ];
		Surf out;
		out.scan_stride = ch;
		out.idx_stride = (long) p -> new_w * ch;
		out.prefix = op.out_prefix;

		if( i == nv - 1 )
		{
			out.base = dst - (long) row0 * out.idx_stride;
		}
		else
		{
			out.base = p -> vbuf[ i ];
		}

		if(( rc = launch_op( op, ch, false, in, out, 0, p -> new_w, va[ i ],
			vb[ i ] + 1, st )) != 0 ) return( rc );

		in = out;
	}

	return( AVIRHIP_OK );
}

static void band_src_rows( const avirhip_plan* p, int row0, int row1,
	int* first, int* last );

static int exec_device( avirhip_plan* p, const void* src, void* dst,
	int row0, int row1, hipStream_t st )
{
	int rc;

	if( p -> f64 )
	{
		return( exec_f64( p, src, dst, row0, row1, st ));
	}

	// LANCIR: the exact-2x kernel needs no scratch; the generic kernels'
	// intermediate is only allocated when they actually run
	if( !p -> is_lancir && ( rc = ensure_scratch( p )) != 0 ) return( rc );

	if( p -> is_lancir && p -> inner != nullptr && p -> path != 1 )
	{
		avirhip_plan* const q = p -> inner;
		const LancirAxisDev& V = p -> lv;
		void* m;

		if( row1 <= row0 )
		{
			return( AVIRHIP_OK );
		}

		// The float RGBA copy of the source and the float result are only
		// allocated when a pass needs them: with the raw loader and the fused
		// output stage (the automatic path of uint8 / uint16 images) neither
		// is touched -- 16 bytes per source and per destination pixel, per
		// plan, spare and replica, that also count against the plan cache.
		auto need_lres = [&]() -> int
		{
			if( p -> lres == nullptr )
			{
				const int r = dev_alloc( p, (size_t) p -> new_w * p -> new_h * 4 *
					sizeof( float ), &m );

				if( r != 0 ) return( r );
				p -> lres = (float*) m;
			}

			return( AVIRHIP_OK );
		};

		// the source rows this band's vertical windows read
		const int sa = std::max( 0, std::min( V.h_start[ row0 ],
			p -> src_h - 1 ));
		const int sb = std::max( 0, std::min( V.h_start[ row1 - 1 ] +
			V.kernel_len - 1, p -> src_h - 1 ));
		const size_t es = dtype_size( p -> in_type );
		// (double and uint32 elements, lancir.h:373-377: the pack pass and the
		// output stage convert them; the fast kernels' own loaders and fused
		// stores know uint8, uint16 and float)
		const bool in_fast = ( p -> in_type <= AVIRHIP_F32 );
		const bool out_fast = ( p -> out_type <= AVIRHIP_F32 );

		q -> path = p -> path;

		// the inner plan's first pass reads this image itself where it can;
		// otherwise the pack pass makes its float RGBA copy
		q -> lraw.on = ( getenv( "AVIRHIP_NO_FUSED_OUT" ) == nullptr && in_fast &&
			( gpass_lancir_takes_raw( q, src, p -> in_type, p -> io_ch,
			p -> src_stride ) || ( lanc2_takes_raw( q, src,
			p -> in_type, p -> io_ch, p -> src_stride ) &&
			// (RGBA: its integer stage stores a lane's two elements at once)
			( p -> io_ch == 3 || ( p -> new_stride & 1 ) == 0 ) &&
			( p -> out_type != AVIRHIP_U8 || p -> io_ch == 3 ||
			( (uintptr_t) dst & 1 ) == 0 ) &&
			( p -> out_type != AVIRHIP_U16 ||
			( (uintptr_t) dst & ( p -> io_ch == 3 ? 1 : 3 )) == 0 ) &&
			( p -> out_type != AVIRHIP_F32 || ( (uintptr_t) dst & 3 ) == 0 ))));

		q -> lraw.ptr = src; q -> lraw.type = p -> in_type;
		q -> lraw.ch = p -> io_ch; q -> lraw.stride = p -> src_stride;

		auto do_pack = [&]() -> int
		{
			if( p -> packed == nullptr )
			{
				const int r = dev_alloc( p, (size_t) p -> src_w * p -> src_h * 4 *
					sizeof( float ), &m );

				if( r != 0 ) return( r );
				p -> packed = (float*) m;
			}

			return( launch_pack( (const char*) src +
				(size_t) sa * p -> src_stride * es, p -> in_type, p -> packed +
				(size_t) sa * p -> src_w * 4, p -> src_w, sb - sa + 1,
				p -> io_ch, 4, p -> src_stride, st ));
		};

		if( !q -> lraw.on && ( rc = do_pack() ) != 0 ) return( rc );

		q -> lfuse.on = ( getenv( "AVIRHIP_NO_FUSED_OUT" ) == nullptr &&
			out_fast );
		q -> lfuse.done = 0;
		q -> lfuse.type = p -> out_type; q -> lfuse.ch = p -> io_ch;
		q -> lfuse.unity = p -> l_unity; q -> lfuse.out_mul = p -> l_out_mul;
		q -> lfuse.clampv = p -> l_clamp;
		q -> lfuse.dst = dst; q -> lfuse.stride = p -> new_stride;

		// (a null result buffer: only the fused store may run; a pass that
		// would write the float result returns AVIRHIP_NEED_DST first)
		if( !q -> lfuse.on && ( rc = need_lres() ) != 0 ) return( rc );

		rc = exec_device( q, p -> packed, p -> lres, row0, row1, st );

		// (at most one round of each: the kernel that was to read the owner's
		// image refused the call -> float copy; a pass would have written the
		// float result -> result buffer)
		for( int round = 0; round < 2 && ( rc == AVIRHIP_NEED_SRC ||
			rc == AVIRHIP_NEED_DST ); round++ )
		{
			if( rc == AVIRHIP_NEED_SRC )
			{
				q -> lraw.on = 0;
				if(( rc = do_pack() ) != 0 ) break;
			}
			else
			if(( rc = need_lres() ) != 0 ) break;

			q -> lfuse.done = 0;
			rc = exec_device( q, p -> packed, p -> lres, row0, row1, st );
		}

		q -> lfuse.on = 0;
		q -> lraw.on = 0;

		if( rc == AVIRHIP_NEED_SRC || rc == AVIRHIP_NEED_DST )
		{
			set_error( "LANCIR: the inner plan kept asking for buffers (%d)", rc );
			return( AVIRHIP_EINTERNAL );
		}

		if( rc != 0 || q -> lfuse.done )
		{
			return( rc );
		}

		if( p -> io_ch == 4 )
		{
			return( launch_lancir_out( p, p -> lres, (long) p -> new_w * 4, dst,
				row1 - row0, st ));
		}

		return( launch_lancir_out_pad( p, p -> lres, dst, row1 - row0, st ));
	}

	if( p -> is_lancir )
	{
		const int lpath = ( p -> path != 0 ? p -> path : p -> auto_path );

		if( lpath == 4 && dst == nullptr && !p -> lfuse.on )
		{
			return( AVIRHIP_NEED_DST );
		}

		if( lpath == 4 )
		{
			rc = lanc2_run( p, (const float*) src, (float*) dst, row0, row1,
				st );

			if( rc != 1 )
			{
				return( rc );
			}

			if( p -> lraw.on )
			{
				// (the kernels below would read the float copy the owner skipped)
				return( AVIRHIP_NEED_SRC );
			}

			if( p -> path != 0 )
			{
				set_error( "path 4 cannot run this call (unaligned buffers?)" );
				return( AVIRHIP_EUNSUPPORTED );
			}
		}

		if( lpath == 5 || ( lpath == 4 && p -> path == 0 && gpass_ok( p )))
		{
			// general-ratio pass kernels (vertical pass first)
			rc = gpass_run( p, (const float*) src, p -> src_stride,
				(float*) dst, row0, row1, st );

			if( rc != 1 )
			{
				return( rc );
			}

			if( p -> path != 0 )
			{
				set_error( "path 5 cannot run this call (unaligned buffers?)" );
				return( AVIRHIP_EUNSUPPORTED );
			}
		}

		if( dst == nullptr )
		{
			return( AVIRHIP_NEED_DST );
		}

		if(( rc = ensure_scratch( p )) != 0 ) return( rc );

		return( launch_lancir_generic( p, src, dst, p -> resbuf, row0, row1,
			st ));
	}

	const float* fsrc = (const float*) src;
	long sstride = p -> src_stride;

	const int path = ( p -> path != 0 ? p -> path : p -> auto_path );

	// integer sources on the tiled paths: the tile loader converts and pads
	// them itself, the float copy of the source (pack pass) is skipped
	const bool raw = ( !p -> gamma && ( path == 2 || path == 3 ) &&
		( p -> in_type == AVIRHIP_U8 || p -> in_type == AVIRHIP_U16 ) &&
		p -> ch == 4 && fused_takes_raw( p, path ));

	bool packed_done = false;
	auto do_pack = [&]() -> int
	{
		if( packed_done )
		{
			return( AVIRHIP_OK );
		}

		packed_done = true;

		// a band converts the source rows its windows read, nothing else (the
		// pipelined host-pointer call is 16 bands: 16 whole-frame packs otherwise)
		int pa, pb;
		band_src_rows( p, row0, row1, &pa, &pb );

		if( pb < pa )
		{
			return( AVIRHIP_OK );
		}

		const char* const ps = (const char*) src + (size_t) pa *
			p -> src_stride * dtype_size( p -> in_type );
		float* const pd = p -> packed + (size_t) pa * p -> src_w * p -> ch;

		if( p -> gamma )
		{
			return( launch_pack_gamma( ps, p -> in_type, pd, p -> src_w,
				pb - pa + 1, p -> io_ch, p -> ch, p -> src_stride,
				p -> alpha_index, p -> d_srgb_tbl, st ));
		}

		return( launch_pack( ps, p -> in_type, pd, p -> src_w, pb - pa + 1,
			p -> io_ch, p -> ch, p -> src_stride, st ));
	};

	const bool need_pack = ( p -> gamma || p -> in_type != AVIRHIP_F32 ||
		p -> ch != p -> io_ch );

	// the streaming pass kernel reads integer / narrower sources as they are
	const bool graw = ( need_pack && !p -> gamma && path == 5 &&
		( p -> in_type == AVIRHIP_U8 || p -> in_type == AVIRHIP_U16 ||
		p -> in_type == AVIRHIP_F32 ) && p -> ch == 4 && gpass_takes_raw( p ));

	// the exact-2x marching kernel reads RGB / RGBA uint8 / uint16 images as
	// they lie when it also stores the caller's integer pixels (it may still
	// refuse the call -- alignment, a source window: the pack pass runs then)
	const bool uraw = ( need_pack && !p -> gamma && path == 4 && !p -> is_lancir &&
		( p -> in_type == AVIRHIP_U8 || p -> in_type == AVIRHIP_U16 ) &&
		p -> ch == 4 && ( p -> io_ch == 3 || p -> io_ch == 4 ) &&
		( p -> out_type == AVIRHIP_U8 || p -> out_type == AVIRHIP_U16 ) &&
		p -> dither == AVIRHIP_DITHER_DEF && up2_stores_io( p ) &&
		getenv( "AVIRHIP_NO_FUSED_OUT" ) == nullptr );

	if( need_pack && !raw && !graw && !uraw )
	{
		if(( rc = do_pack()) != 0 ) return( rc );
	}

	if( need_pack )
	{
		fsrc = p -> packed;
		sstride = (long) p -> src_w * p -> ch;
	}

	// float output is the vertical pass' in-place result (avir.h:4956-4979):
	// with gamma it stays linear, only the other output types are
	// de-linearised in the epilogue
	// (fpclass_float4 has no in-place output: its float results pass through
	// the output stage like every other type, which matters with gamma)
	const bool direct = ( p -> out_type == AVIRHIP_F32 &&
		p -> ch == p -> io_ch && !( p -> fp4 && p -> gamma ));

	float* fdst = ( direct ? (float*) dst : p -> resbuf );

	rc = 1;

	if( path >= 2 && path <= 5 )
	{
		if( path == 5 )
		{
			// integer output without gamma / error diffusion: the last pass
			// converts and stores into the caller's image itself
			void* iout = ( !direct && !p -> gamma &&
				(( p -> dither == AVIRHIP_DITHER_DEF &&
				( p -> out_type == AVIRHIP_U8 || p -> out_type == AVIRHIP_U16 )) ||
				p -> out_type == AVIRHIP_F32 ) &&
				getenv( "AVIRHIP_NO_FUSED_OUT" ) == nullptr ? dst : nullptr );

			if( !graw && need_pack && ( rc = do_pack()) != 0 ) return( rc );

			for( int pass = 0; pass < 2; pass++ )
			{
				if( graw )
				{
					rc = gpass_run( p, nullptr, 0, fdst, row0, row1, st, src,
						p -> in_type, p -> io_ch, p -> src_stride, iout );
				}
				else
				{
					rc = gpass_run( p, fsrc, sstride, fdst, row0, row1, st,
						nullptr, 0, 0, 0, iout );
				}

				if( rc == 0 && iout != nullptr )
				{
					return( AVIRHIP_OK ); // (no epilogue)
				}

				if( rc != 1 || iout == nullptr )
				{
					break;
				}

				iout = nullptr; // this plan's last pass cannot: float result
			}
		}
		else
		if( path == 4 )
		{
			// integer / narrow output: the marching kernel's vertical phase
			// converts and stores into the caller's image itself
			void* iout4 = ( !direct && !p -> gamma &&
				(( p -> dither == AVIRHIP_DITHER_DEF &&
				( p -> out_type == AVIRHIP_U8 || p -> out_type == AVIRHIP_U16 )) ||
				p -> out_type == AVIRHIP_F32 ) &&
				getenv( "AVIRHIP_NO_FUSED_OUT" ) == nullptr ? dst : nullptr );

			rc = 1;

			if( iout4 != nullptr && uraw )
			{
				rc = up2_run( p, nullptr, 0, nullptr, row0, row1, st, iout4, src,
					p -> src_stride );

				if( rc == 0 )
				{
					return( AVIRHIP_OK ); // (no pack pass, no epilogue)
				}

				if( rc != 1 ) return( rc );
			}

			if( uraw && ( rc = do_pack()) != 0 ) return( rc );

			rc = 1;

			if( iout4 != nullptr )
			{
				rc = up2_run( p, fsrc, sstride, nullptr, row0, row1, st, iout4 );

				if( rc == 0 )
				{
					return( AVIRHIP_OK ); // (no epilogue)
				}
			}

			if( rc == 1 )
			{
				rc = up2_run( p, fsrc, sstride, fdst, row0, row1, st );
			}
		}
		else
		{
			// integer / narrow output through the whole-ratio vertical kernel:
			// it converts and stores into the caller's image itself
			void* iout2 = ( !direct && !p -> gamma &&
				(( p -> dither == AVIRHIP_DITHER_DEF &&
				( p -> out_type == AVIRHIP_U8 || p -> out_type == AVIRHIP_U16 )) ||
				p -> out_type == AVIRHIP_F32 ) &&
				!( p -> out_type == AVIRHIP_U8 && p -> io_ch == 4 &&
				( (uintptr_t) dst & 3 ) != 0 ) &&
				fused_stores_int( p, path ) &&
				getenv( "AVIRHIP_NO_FUSED_OUT" ) == nullptr ? dst : nullptr );

			if( raw )
			{
				rc = fused_run( p, path, src, p -> in_type, p -> io_ch,
					p -> src_stride, fdst, row0, row1, st, iout2 );
			}
			else
			{
				rc = fused_run( p, path, fsrc, AVIRHIP_F32, 4, sstride, fdst,
					row0, row1, st, iout2 );
			}

			if( rc == 0 && iout2 != nullptr )
			{
				return( AVIRHIP_OK ); // (no epilogue)
			}
		}

		if( rc == 1 && p -> path != 0 )
		{
			set_error( "path %d cannot run this call (unaligned buffers?)",
				path );
			return( AVIRHIP_EUNSUPPORTED );
		}
	}

	if( rc == 1 && need_pack )
	{
		if(( rc = do_pack()) != 0 ) return( rc );
		rc = 1;
	}

	if( rc == 1 )
	{
		rc = run_generic( p, fsrc, sstride, fdst, row0, row1, st );
	}

	if( rc != 0 ) return( rc );

	if( !direct && p -> dither == AVIRHIP_DITHER_ERRD )
	{
		// recursive ditherer: whole frames only (exec_any refuses bands)
		if( p -> errd_line == nullptr )
		{
			void* q;
			// one row of diffusion values per pass of 448 rows (generic.hip:
			// k_errd_mp; the single-workgroup kernels use two), then the
			// passes' progress flags
			const size_t np = (size_t) ( p -> new_h + 447 ) / 448 + 2;

			if(( rc = dev_alloc( p, np * p -> new_w * 4 * sizeof( float ) +
				np * sizeof( unsigned ) + 64, &q )) != 0 ) return( rc );
			p -> errd_line = (float*) q;
		}

		rc = launch_errd( fdst, dst, p -> out_type, p -> new_w, p -> new_h,
			p -> io_ch, p -> ch, p -> tr_mul, p -> pk_out, p -> gamma,
			p -> alpha_index, p -> errd_line, st );
	}
	else
	if( !direct )
	{
		rc = launch_epilogue( fdst, dst, p -> out_type,
			(long) ( row1 - row0 ) * p -> new_w * p -> io_ch, p -> tr_mul,
			p -> pk_out, ( p -> gamma && ( p -> out_type != AVIRHIP_F32 ||
			p -> fp4 )), p -> io_ch, p -> ch, p -> alpha_index, st,
			p -> d_gthr, ( p -> dither == AVIRHIP_DITHER_DEF_RNE ));
	}

	return( rc );
}

} // namespace avirhip

// AVIRHIP_MEM_AUTO: device memory if the HIP runtime knows the pointer as such.
extern "C" int avirhip_resolve_mem( const void* ptr, int mem )
{
	if( mem != AVIRHIP_MEM_AUTO )
	{
		return( mem );
	}

	hipPointerAttribute_t at;

	if( ptr != nullptr && hipPointerGetAttributes( &at, ptr ) == hipSuccess &&
		( at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged ))
	{
		return( AVIRHIP_MEM_DEVICE );
	}

	(void) hipGetLastError(); // unregistered host memory reports an error
	return( AVIRHIP_MEM_HOST );
}

namespace avirhip {

static int clone_plan( const avirhip_plan* s, int device, avirhip_plan** out );

// The source rows [*first, *last] (inclusive) the output rows [row0, row1)
// read: the vertical axis' op chain walked backwards (AVIR), the vertical
// filter positions (LANCIR). Rows outside the range are never touched by a band
// call -- not even by a tap that multiplies them by zero (a stale or foreign
// NaN there must not reach the result; tools/fuzz_values.py poisons them).
void axis_src_range( const LAxis& ax, int row0, int row1, int src_len,
	int* first, int* last )
{
	int a = row0, b = row1 - 1;

	for( int i = (int) ax.ops.size() - 1; i >= 0; i-- )
	{
		int ia, ib;
		const LOp& op = ax.ops[ i ];
		need_range( op, std::max( 0, std::min( a, op.out_len - 1 )),
			std::max( 0, std::min( b, op.out_len - 1 )), ia, ib );
		a = ia;
		b = ib;
	}

	*first = std::max( 0, std::min( a, src_len - 1 ));
	*last = std::max( 0, std::min( b, src_len - 1 ));
}

// (host only: the planner front end answers avirhip_*_band_source_rows for a
// geometry without creating a device plan)
int desc_band_src_rows( const avirhip_plan_desc* d, int row0, int row1,
	int* first, int* last )
{
	LAxis v;
	const int rc = lower_axis( d -> v, d -> src_h, d -> new_h, v,
		d -> work_f64 != 0 );

	if( rc != 0 )
	{
		return( rc );
	}

	axis_src_range( v, row0, row1, d -> src_h, first, last );
	return( AVIRHIP_OK );
}

static void band_src_rows( const avirhip_plan* p, int row0, int row1,
	int* first, int* last )
{
	if( row1 <= row0 )
	{
		*first = 0;
		*last = -1;
		return;
	}

	if( p -> is_lancir )
	{
		*first = std::max( 0, std::min( p -> lv.h_start[ row0 ],
			p -> src_h - 1 ));
		*last = std::max( 0, std::min( p -> lv.h_start[ row1 - 1 ] +
			p -> lv.kernel_len - 1, p -> src_h - 1 ));
		return;
	}

	axis_src_range( p -> v, row0, row1, p -> src_h, first, last );
}

static int band_last_src_row( const avirhip_plan* p, int row0, int row1 )
{
	int a, b;
	band_src_rows( p, row0, row1, &a, &b );
	return( b );
}

// Host-pointer call, pipelined (the call every existing caller of the
// reference makes, avir.h:4680-4684): the frame in row bands, the source rows
// of band b + 1 travelling host -> device (an uploader thread: a copy from
// pageable memory holds its thread until the data is staged) while band b is
// computed and band b - 1 travels device -> host. PCIe is full duplex, so the
// call costs the longer direction instead of the sum (cfg3: 531 MB down,
// 133 MB up). Returns 1 when the call should take the serial path.
static int exec_device( avirhip_plan* p, const void* src, void* dst,
	int row0, int row1, hipStream_t st );

static int exec_host_pipelined( avirhip_plan* p, const void* src, void* dst,
	size_t src_bytes, size_t dst_bytes, size_t row_bytes, hipStream_t st )
{
	static const bool off = ( getenv( "AVIRHIP_NO_HOST_PIPELINE" ) != nullptr );
	enum { NBMAX = 32 };
	static const int NB = []() { const char* e = getenv( "AVIRHIP_HOST_BANDS" );
		const int n = ( e != nullptr ? atoi( e ) : 16 );
		return( n < 2 ? 2 : ( n > NBMAX ? (int) NBMAX : n )); }();

	// Every band is one exec_device call. Stages that are not limited to the
	// band's rows would be repeated NB times: the double pipeline and the
	// generic kernels run whole axes through a filtered upsample -- such calls
	// stay serial. (The pack pass converts only the rows a band reads.)
	const int xp = ( p -> path != 0 ? p -> path : p -> auto_path );

	if( p -> f64 || ( !p -> is_lancir && xp == 1 ))
	{
		return( 1 );
	}

	if( off || p -> new_h < 4 * NB || src_bytes + dst_bytes < ( 16u << 20 ) ||
		( !p -> is_lancir && p -> dither == AVIRHIP_DITHER_ERRD &&
		( p -> out_type == AVIRHIP_U8 || p -> out_type == AVIRHIP_U16 )))
	{
		return( 1 );
	}

	if( p -> pipe_in == nullptr )
	{
		// built aside and committed to the plan only when complete: a plan
		// must never hold half a set (the next call would index past pipe_ev)
		hipStream_t a = nullptr, b = nullptr;
		std::vector< hipEvent_t > evs;
		bool ok = ( hipStreamCreateWithFlags( &a, hipStreamNonBlocking ) ==
			hipSuccess );

		ok = ok && ( hipStreamCreateWithFlags( &b, hipStreamNonBlocking ) ==
			hipSuccess );

		for( int i = 0; ok && i < 2 * NB + 1; i++ )
		{
			hipEvent_t e;
			ok = ( hipEventCreateWithFlags( &e, hipEventDisableTiming ) ==
				hipSuccess );

			if( ok ) evs.push_back( e );
		}

		if( !ok )
		{
			(void) hipGetLastError();

			for( size_t i = 0; i < evs.size(); i++ )
			{
				(void) hipEventDestroy( evs[ i ]);
			}

			if( a != nullptr ) (void) hipStreamDestroy( a );
			if( b != nullptr ) (void) hipStreamDestroy( b );

			return( 1 ); // (the serial path needs none of this)
		}

		p -> pipe_in = a; p -> pipe_out = b;
		p -> pipe_ev.swap( evs );
	}

	hipStream_t s_in = (hipStream_t) p -> pipe_in;
	hipStream_t s_out = (hipStream_t) p -> pipe_out;
	static const bool do_reg = ( getenv( "AVIRHIP_HOST_REGISTER" ) != nullptr );
	bool reg_s = false, reg_d = false;

	if( do_reg )
	{
		reg_s = ( hipHostRegister( (void*) src, src_bytes,
			hipHostRegisterDefault ) == hipSuccess );
		reg_d = ( hipHostRegister( dst, dst_bytes, hipHostRegisterDefault ) ==
			hipSuccess );
		(void) hipGetLastError();
	}

	const size_t es = dtype_size( p -> in_type );
	const size_t srow = (size_t) p -> src_stride * es;
	int cut[ NBMAX + 1 ], last[ NBMAX ];

	for( int b = 0; b <= NB; b++ )
	{
		cut[ b ] = (int) ( (long) p -> new_h * b / NB );
	}

	for( int b = 0; b < NB; b++ )
	{
		last[ b ] = band_last_src_row( p, cut[ b ], cut[ b + 1 ]);

		if( b > 0 )
		{
			last[ b ] = std::max( last[ b ], last[ b - 1 ]);
		}
	}

	last[ NB - 1 ] = p -> src_h - 1;

	// the copy streams start behind whatever the caller's stream holds
	hipEvent_t e0 = p -> pipe_ev[ 2 * NB ];
	AVIRHIP_HIPCHECK( hipEventRecord( e0, st ));
	AVIRHIP_HIPCHECK( hipStreamWaitEvent( s_in, e0, 0 ));
	AVIRHIP_HIPCHECK( hipStreamWaitEvent( s_out, e0, 0 ));

	std::atomic< int > ready( -1 );
	std::atomic< int > up_err( 0 );
	const int dev = p -> device;
	char* const dsrc = (char*) p -> stage_src;

	auto upload_rows = [&]()
	{
		hipError_t he = hipSetDevice( dev );

		if( he != hipSuccess )
		{
			up_err.store( (int) he );
			ready.store( NB );
			return;
		}

		int done = 0; // source rows on their way

		for( int b = 0; b < NB; b++ )
		{
			const int upto = last[ b ] + 1;

			if( upto > done )
			{
				const size_t o0 = (size_t) done * srow;
				const size_t o1 = std::min( src_bytes, (size_t) upto * srow );

				if( o1 > o0 && ( he = hipMemcpyAsync( dsrc + o0,
					(const char*) src + o0, o1 - o0, hipMemcpyHostToDevice,
					s_in )) != hipSuccess )
				{
					up_err.store( (int) he );
				}

				done = upto;
			}

			if(( he = hipEventRecord( p -> pipe_ev[ b ], s_in )) != hipSuccess )
			{
				up_err.store( (int) he );
			}

			ready.store( b );
		}
	};

	// (a thread that cannot be started must not throw through extern "C":
	// the rows then go up from this thread, band after band, before the loop)
	// (... and a throw below -- a host allocation failing inside a band's launch
	// code -- must not unwind past a joinable thread: that is std::terminate)
	struct Joiner
	{
		std::thread t;
		~Joiner() { if( t.joinable() ) t.join(); }
	} up;
	std::thread& uploader = up.t;

	try
	{
		uploader = std::thread( upload_rows );
	}
	catch( ... )
	{
		upload_rows();
	}

	int rc = AVIRHIP_OK;

	// band b's copy down holds this thread (pageable memory), so band b + 1's
	// kernels are enqueued BEFORE it: the device never waits for the host
	auto copy_down = [&]( const int b )
	{
		const size_t off_b = (size_t) cut[ b ] * row_bytes;
		const size_t len_b = std::min( dst_bytes,
			(size_t) cut[ b + 1 ] * row_bytes ) - off_b;

		if( hipStreamWaitEvent( s_out, p -> pipe_ev[ NB + b ], 0 ) != hipSuccess ||
			hipMemcpyAsync( (char*) dst + off_b, (char*) p -> stage_dst + off_b,
			len_b, hipMemcpyDeviceToHost, s_out ) != hipSuccess )
		{
			rc = AVIRHIP_EHIP;
		}
	};

	int enq = 0; // bands whose kernels are enqueued

	for( int b = 0; b < NB && rc == AVIRHIP_OK; b++ )
	{
		while( ready.load() < b )
		{
			std::this_thread::yield();
		}

		if( up_err.load() != 0 )
		{
			break;
		}

		if( hipStreamWaitEvent( st, p -> pipe_ev[ b ], 0 ) != hipSuccess )
		{
			rc = AVIRHIP_EHIP;
			break;
		}

		rc = exec_device( p, dsrc, (char*) p -> stage_dst +
			(size_t) cut[ b ] * row_bytes, cut[ b ], cut[ b + 1 ], st );

		if( rc != AVIRHIP_OK )
		{
			break;
		}

		if( hipEventRecord( p -> pipe_ev[ NB + b ], st ) != hipSuccess )
		{
			rc = AVIRHIP_EHIP;
			break;
		}

		enq = b + 1;

		if( b >= 1 )
		{
			copy_down( b - 1 );
		}
	}

	if( rc == AVIRHIP_OK && enq == NB )
	{
		copy_down( NB - 1 );
	}

	if( uploader.joinable() ) uploader.join();
	(void) hipStreamSynchronize( s_in );
	(void) hipStreamSynchronize( s_out );
	(void) hipStreamSynchronize( st );

	if( reg_s ) (void) hipHostUnregister( (void*) src );
	if( reg_d ) (void) hipHostUnregister( dst );

	if( up_err.load() != 0 && rc == AVIRHIP_OK )
	{
		set_error( "host-pointer call: the source upload failed: %s",
			hipGetErrorString( (hipError_t) up_err.load() ));
		rc = AVIRHIP_EHIP;
	}

	return( rc );
}


// `win_rows` > 0: `src` holds only the source rows [win_first, win_first +
// win_rows) of the frame (avirhip_resize_window); otherwise the whole frame.
static int exec_any( avirhip_plan* p, const void* src, int src_mem, void* dst,
	int dst_mem, int row0, int row1, void* stream, bool have_lock = false,
	int win_first = 0, int win_rows = 0 )
{
	src_mem = avirhip_resolve_mem( src, src_mem );
	dst_mem = avirhip_resolve_mem( dst, dst_mem );

	if( p == nullptr || src == nullptr || dst == nullptr || row0 < 0 ||
		row1 > p -> new_h || row0 > row1 )
	{
		set_error( "bad execute arguments" );
		return( AVIRHIP_EINVAL );
	}

	const bool windowed = ( win_rows > 0 );

	if( windowed )
	{
		int need_a = 0, need_b = -1;
		band_src_rows( p, row0, row1, &need_a, &need_b );

		// (compared in long: win_first + win_rows of a hostile caller must not
		// wrap; the window has to lie inside the frame whatever the band reads)
		if( win_first < 0 || (long) win_first + win_rows > (long) p -> src_h ||
			( need_b >= need_a && ( need_a < win_first ||
			(long) need_b >= (long) win_first + win_rows )))
		{
			set_error( "resize_window: output rows [%d, %d) read source rows "
				"[%d, %d], the window holds [%d, %d)", row0, row1, need_a,
				need_b, win_first, win_first + win_rows );
			return( AVIRHIP_EINVAL );
		}

		// an empty band reads nothing and stores nothing: done before any copy
		// of the window is made
		if( row1 <= row0 )
		{
			return( AVIRHIP_OK );
		}
	}

	if( !p -> is_lancir && p -> dither == AVIRHIP_DITHER_ERRD &&
		( row0 != 0 || row1 != p -> new_h ))
	{
		set_error( "the error-diffusion ditherer is recursive over rows "
			"(avir.h:4473-4480): row bands cannot be executed" );
		return( AVIRHIP_EUNSUPPORTED );
	}

	hipStream_t st = (hipStream_t) stream;

	// A plan's tables and scratch live on the device it was created on: run
	// there, whatever device the calling thread has current, and put the
	// caller's device back on every way out.
	struct DeviceGuard
	{
		int keep;
		bool on;
		DeviceGuard( int want ) : keep( 0 ), on( false )
		{
			if( hipGetDevice( &keep ) == hipSuccess && keep != want )
			{
				on = ( hipSetDevice( want ) == hipSuccess );
			}
		}
		~DeviceGuard() { if( on ) (void) hipSetDevice( keep ); }
	} devguard( p -> device );

	// Calls on one plan share its scratch buffers: serialise them (the
	// reference allows concurrent resizeImage() calls on one object). The
	// only scratch-free case, device-resident float RGBA through the
	// single-launch 2x kernels, skips this.
	const size_t src_bytes = ( (size_t) ( p -> src_h - 1 ) * p -> src_stride +
		(size_t) p -> src_w * p -> io_ch ) * dtype_size( p -> in_type );
	// (what `src` addresses: the frame, or the rows of a window)
	const size_t win_bytes = ( !windowed ? src_bytes :
		( (size_t) ( win_rows - 1 ) * p -> src_stride +
		(size_t) p -> src_w * p -> io_ch ) * dtype_size( p -> in_type ));
	const size_t row_bytes = (size_t) p -> new_stride *
		dtype_size( p -> out_type );
	const size_t dst_bytes = ( row1 > row0 ?
		(size_t) ( row1 - row0 - 1 ) * row_bytes + (size_t) p -> new_w *
		p -> io_ch * dtype_size( p -> out_type ) : 0 );
	const bool overlap = ( !windowed && src_mem == AVIRHIP_MEM_DEVICE &&
		dst_mem == AVIRHIP_MEM_DEVICE &&
		(const char*) src < (const char*) dst + dst_bytes &&
		(const char*) dst < (const char*) src + src_bytes );

	const int xpath = ( p -> path != 0 ? p -> path : p -> auto_path );
	// (CLancIR is not thread-safe in the reference either, lancir.h:319-349:
	// its plans always take the lock, and may allocate scratch under it)
	const bool scratch_free = ( !p -> is_lancir && !overlap && !windowed &&
		xpath == 4 &&
		p -> in_type == AVIRHIP_F32 &&
		p -> out_type == AVIRHIP_F32 && p -> ch == p -> io_ch &&
		!p -> gamma && src_mem == AVIRHIP_MEM_DEVICE &&
		dst_mem == AVIRHIP_MEM_DEVICE &&
		// ... and only when the 2x kernel cannot refuse the call and fall
		// back to the generic kernels (which use scratch)
		(( (uintptr_t) src | (uintptr_t) dst ) & 15 ) == 0 &&
		( p -> src_stride & 3 ) == 0 &&
		(long) p -> src_stride * 4 < ( 1L << 22 ) &&
		(long) p -> new_w * 16 < ( 1L << 22 ));

	// Scratch users take the plan's lock. When another thread holds it (the
	// reference allows concurrent resizeImage() calls on one object,
	// README.md:83-85), the call runs on a spare replica of the plan instead
	// -- same tables, its own scratch -- so that the two calls overlap on
	// their streams rather than queue behind one buffer set. Up to three
	// spares; beyond that callers wait.
	std::unique_lock< std::mutex > guard( p -> exec_mtx, std::defer_lock );

	if( !scratch_free && !have_lock && !guard.try_lock())
	{
		avirhip_plan* spare = nullptr;
		{
			// (its own mutex: avirhip_resize_sharded holds shard_mtx for the
			// whole call and reaches this point for the band on the plan's
			// own device)
			std::lock_guard< std::mutex > sl( p -> spare_mtx );

			for( size_t i = 0; i < p -> spares.size() && spare == nullptr; i++ )
			{
				if( p -> spares[ i ] -> exec_mtx.try_lock())
				{
					spare = p -> spares[ i ];
				}
			}

			if( spare == nullptr && p -> spares.size() < 3 && !p -> is_spare )
			{
				if( clone_plan( p, p -> device, &spare ) == AVIRHIP_OK )
				{
					spare -> is_spare = 1;
					spare -> path = p -> path;
					spare -> variant = p -> variant;
					spare -> exec_mtx.lock();
					p -> spares.push_back( spare );
				}
				else
				{
					spare = nullptr;
				}
			}
		}

		if( spare != nullptr )
		{
			// (exec_any on the spare: its mutex is held by this thread, and
			// released on every way out -- a throw included)
			std::unique_lock< std::mutex > sg( spare -> exec_mtx,
				std::adopt_lock );

			return( exec_any( spare, src, src_mem, dst, dst_mem, row0, row1,
				stream, true, win_first, win_rows ));
		}

		guard.lock();
	}

	// hipStreamPerThread is ONE handle that names a different stream in every
	// thread: such a call records the plan's event itself, when it ends (below).
	struct PerThreadRecord
	{
		avirhip_plan* p;
		hipStream_t st;
		bool on;
		~PerThreadRecord()
		{
			if( on && ( p -> last_done == nullptr ||
				hipEventRecord( p -> last_done, st ) != hipSuccess ))
			{
				(void) hipGetLastError();
				(void) hipStreamSynchronize( st );
			}
		}
	} ptrec = { p, st, false };

	if( !scratch_free )
	{
		// The plan's scratch buffers are reused by the next call. Calls on one
		// stream are ordered by the stream itself; only when the stream
		// CHANGES does the new one have to wait for what the old one still
		// holds -- the event is recorded on the old stream then, not after
		// every call: a record between two frames is a barrier packet with a
		// release fence, 6-8 us of idle GPU per frame (the kernel trace of
		// 640x480 -> 1024x768: H -> V inside a call 0.6 us apart, V -> the next
		// call's H 7.7 us). "The same stream" is the same handle AND, for
		// hipStreamPerThread, the same thread. (hipStreamGetId would also tell a
		// destroyed stream's recycled handle value apart, but it is a hip_7.1
		// symbol: the runtime PyTorch ships is 7.0 and could not load the
		// library. A caller that destroys a stream with calls of this plan
		// still in flight has to synchronise it first.)
		const bool per_thread = ( st == hipStreamPerThread );
		const std::thread::id tid = std::this_thread::get_id();

		if( p -> last_used )
		{
			if( p -> last_done == nullptr )
			{
				AVIRHIP_HIPCHECK( hipEventCreateWithFlags( &p -> last_done,
					hipEventDisableTiming ));
			}

			if( p -> last_recorded )
			{
				// (the last call was on a per-thread stream and recorded the
				// event when it ended)
				if( !( per_thread && p -> last_tid == tid ))
				{
					AVIRHIP_HIPCHECK( hipStreamWaitEvent( st, p -> last_done, 0 ));
				}
			}
			else
			if( p -> last_stream != (void*) st )
			{
				if( hipEventRecord( p -> last_done,
					(hipStream_t) p -> last_stream ) != hipSuccess ||
					hipStreamWaitEvent( st, p -> last_done, 0 ) != hipSuccess )
				{
					// (the old stream is gone: what it held may still run)
					(void) hipGetLastError();
					AVIRHIP_HIPCHECK( hipDeviceSynchronize() );
				}
			}
		}
		else
		if( per_thread && p -> last_done == nullptr )
		{
			AVIRHIP_HIPCHECK( hipEventCreateWithFlags( &p -> last_done,
				hipEventDisableTiming ));
		}

		p -> last_stream = (void*) st;
		p -> last_tid = tid;
		p -> last_recorded = per_thread;
		p -> last_used = true;
		ptrec.on = per_thread;
	}

	const void* dsrc = src;
	void* ddst = dst;

	// A window on the marching kernels (exact-2x float RGBA plans: the sharded
	// configurations of BASELINE.json): the kernels take a VIRTUAL frame base,
	// window - win_first rows, and clamp their row indices to the window -- for
	// every access that reaches a result "clamp to the window" IS "clamp to the
	// frame" (a needed row outside the frame is the frame's first / last row,
	// which avirhip_band_source_rows then names, so the window holds it). No
	// frame-sized buffer: a device window is read where it lies (zero copy), a
	// host window is uploaded into a window-sized staging buffer.
	const size_t win_off = (size_t) win_first * p -> src_stride *
		dtype_size( p -> in_type );
	const bool dst_direct = ( dst_mem == AVIRHIP_MEM_DEVICE );
	bool win_native = false;

	if( windowed && !overlap && getenv( "AVIRHIP_NO_NATIVE_WINDOW" ) == nullptr )
	{
		// (alignment of the pointers the kernel will see: the staging buffers
		// are 256-byte aligned, win_off is a multiple of the row pitch)
		const void* const vsrc = (const char*) ( src_mem == AVIRHIP_MEM_DEVICE ?
			src : (const void*) (uintptr_t) 256 ) - ( win_off & 255 );
		const void* const vdst = ( dst_direct ? dst : (void*) (uintptr_t) 256 );

		win_native = ( xpath == 4 && ( p -> is_lancir ?
			lanc2_takes_window( p, vsrc, vdst ) :
			up2_takes_window( p, vsrc, vdst )));
	}

	// (the window fields are the plan's for this call only: the lock is held)
	struct WindowScope
	{
		avirhip_plan* p;
		~WindowScope() { p -> win_first = 0; p -> win_rows = 0; }
	} winscope = { p };

	if( src_mem == AVIRHIP_MEM_HOST || ( windowed && !win_native ))
	{
		const size_t need = ( win_native ? win_bytes : src_bytes );

		if( p -> stage_src_bytes < need )
		{
			void* q;
			int rc = dev_alloc( p, need, &q );
			if( rc != 0 ) return( rc );
			p -> stage_src = q;
			p -> stage_src_bytes = need;
		}

		dsrc = p -> stage_src;
	}

	if( windowed && win_native )
	{
		if( src_mem == AVIRHIP_MEM_HOST )
		{
			AVIRHIP_HIPCHECK( hipMemcpyAsync( p -> stage_src, src, win_bytes,
				hipMemcpyHostToDevice, st ));
		}

		// (pointer arithmetic only: nothing below row win_first is ever read)
		dsrc = (const void*) ( (uintptr_t) dsrc - win_off );
		p -> win_first = win_first;
		p -> win_rows = win_rows;
	}
	else
	if( windowed )
	{
		// Other plans: the window's rows go to their place in a frame-sized
		// staging buffer of the plan -- the only source bytes that move (SURVEY.md
		// 8e: "GPU g receives source rows [r0 - halo, r1 + halo]"); the kernels
		// then run as on a whole frame, and whatever they load outside the window
		// never reaches a result (avirhip_band_source_rows).
		AVIRHIP_HIPCHECK( hipMemcpyAsync( (char*) p -> stage_src + win_off,
			src, win_bytes, ( src_mem == AVIRHIP_MEM_HOST ?
			hipMemcpyHostToDevice : hipMemcpyDeviceToDevice ), st ));
	}

	// NewBuf may alias SrcBuf (avir.h:4650-4652: allowed when the result is
	// not larger). The reference survives that because it has consumed the
	// source into FltBuf before it writes; the kernels here read the source
	// while they write, so an overlapping device source is copied aside first.
	if( overlap )
	{
		if( p -> stage_src_bytes < src_bytes )
		{
			void* q;
			int rc = dev_alloc( p, src_bytes, &q );
			if( rc != 0 ) return( rc );
			p -> stage_src = q;
			p -> stage_src_bytes = src_bytes;
		}

		AVIRHIP_HIPCHECK( hipMemcpyAsync( p -> stage_src, src, src_bytes,
			hipMemcpyDeviceToDevice, st ));
		dsrc = p -> stage_src;
	}

	if( dst_mem == AVIRHIP_MEM_HOST )
	{
		if( p -> stage_dst_bytes < dst_bytes )
		{
			void* q;
			int rc = dev_alloc( p, dst_bytes, &q );
			if( rc != 0 ) return( rc );
			p -> stage_dst = q;
			p -> stage_dst_bytes = dst_bytes;
		}

		ddst = p -> stage_dst;
	}

	int rc = 1;

	// (NewBuf may alias SrcBuf, avir.h:4650-4652: result rows copied down while
	// source rows still travel up would overwrite them -- such calls keep the
	// serial order, whole source up first)
	const bool host_alias = ( (const char*) src < (const char*) dst + dst_bytes &&
		(const char*) dst < (const char*) src + src_bytes );

	if( src_mem == AVIRHIP_MEM_HOST && dst_mem == AVIRHIP_MEM_HOST &&
		!windowed && !host_alias && row0 == 0 && row1 == p -> new_h &&
		(size_t) p -> new_w * p -> io_ch * dtype_size( p -> out_type ) ==
		row_bytes )
	{
		rc = exec_host_pipelined( p, src, dst, src_bytes, dst_bytes, row_bytes,
			st );

		if( rc != 1 )
		{
			return( rc );
		}
	}

	if( src_mem == AVIRHIP_MEM_HOST && !windowed )
	{
		AVIRHIP_HIPCHECK( hipMemcpyAsync( p -> stage_src, src, src_bytes,
			hipMemcpyHostToDevice, st ));
	}

	rc = exec_device( p, dsrc, ddst, row0, row1, st );
	if( rc != 0 ) return( rc );

	if( dst_mem == AVIRHIP_MEM_HOST )
	{
		const size_t payload = (size_t) p -> new_w * p -> io_ch *
			dtype_size( p -> out_type );

		if( payload == row_bytes || row1 - row0 < 2 )
		{
			AVIRHIP_HIPCHECK( hipMemcpyAsync( dst, ddst, dst_bytes,
				hipMemcpyDeviceToHost, st ));
		}
		else
		{
			// padded destination rows (CLancIRParams::NewSSize): the bytes
			// between rows belong to the caller
			AVIRHIP_HIPCHECK( hipMemcpy2DAsync( dst, row_bytes, ddst,
				row_bytes, payload, (size_t) ( row1 - row0 ),
				hipMemcpyDeviceToHost, st ));
		}

		AVIRHIP_HIPCHECK( hipStreamSynchronize( st ));
	}
	else
	if( src_mem == AVIRHIP_MEM_HOST )
	{
		AVIRHIP_HIPCHECK( hipStreamSynchronize( st ));
	}

	return( AVIRHIP_OK );
}

size_t plan_device_bytes( avirhip_plan* p )
{
	if( p == nullptr )
	{
		return( 0 );
	}

	size_t b = p -> alloc_bytes + plan_device_bytes( p -> inner );

	{
		std::lock_guard< std::mutex > sl( p -> spare_mtx );

		for( size_t i = 0; i < p -> spares.size(); i++ )
		{
			b += plan_device_bytes( p -> spares[ i ]);
		}
	}

	// (replicas are only touched under shard_mtx, which a sharded call holds
	// for its whole duration: do not wait for it, count what can be seen)
	if( p -> shard_mtx.try_lock())
	{
		for( size_t i = 0; i < p -> replicas.size(); i++ )
		{
			b += plan_device_bytes( p -> replicas[ i ]);
		}

		p -> shard_mtx.unlock();
	}

	return( b );
}

static avirhip_plan* new_plan()
{
	avirhip_plan* p = new avirhip_plan();
	p -> is_lancir = 0; p -> device = 0; p -> alloc_bytes = 0;
	p -> path = 0; p -> variant = 0; p -> fused_ok = 0; p -> auto_path = 1; p -> fused = nullptr; p -> tile64 = nullptr; p -> up2 = nullptr; p -> lanc2 = nullptr; p -> gpass = nullptr;
	p -> packed = nullptr; p -> resbuf = nullptr; p -> lres = nullptr;
	p -> stage_src = nullptr; p -> stage_dst = nullptr;
	p -> stage_src_bytes = 0; p -> stage_dst_bytes = 0;
	p -> win_first = 0; p -> win_rows = 0;
	p -> shard_band = nullptr; p -> shard_band_bytes = 0; p -> shard_ldev = -1;
	p -> last_done = nullptr; p -> last_stream = nullptr; p -> last_used = false;
	p -> last_recorded = false;
	p -> pipe_in = nullptr; p -> pipe_out = nullptr;
	p -> shard_src = nullptr; p -> shard_src_bytes = 0;
	p -> tr_mul = 1.0; p -> pk_out = 0.0;
	p -> gamma = 0; p -> alpha_index = -1; p -> d_srgb_tbl = nullptr;
	p -> d_gthr = nullptr;
	p -> dither = AVIRHIP_DITHER_DEF; p -> fp4 = 0; p -> errd_line = nullptr;
	p -> is_spare = 0;
	p -> ch = 0; p -> io_ch = 0;
	p -> l_out_mul = 1.0f; p -> l_clamp = 0.0f; p -> l_unity = 1;
	p -> inner = nullptr; p -> l_order = 4;
	memset( &p -> lfuse, 0, sizeof( p -> lfuse ));
	memset( &p -> lraw, 0, sizeof( p -> lraw ));
	(void) hipGetDevice( &p -> device );
	return( p );
}

} // namespace avirhip

using namespace avirhip;

extern "C" {

int avirhip_device_count( void )
{
	int n = 0;

	if( hipGetDeviceCount( &n ) != hipSuccess )
	{
		return( 0 );
	}

	int good = 0;

	for( int i = 0; i < n; i++ )
	{
		hipDeviceProp_t pr;

		if( hipGetDeviceProperties( &pr, i ) == hipSuccess &&
			strncmp( pr.gcnArchName, "gfx950", 6 ) == 0 )
		{
			good++;
		}
	}

	return( good );
}

int avirhip_init( int device )
try
{
	avirhip::clear_error();
	if( avirhip_device_count() < 1 )
	{
		set_error( "no gfx950 device visible" );
		return( AVIRHIP_ENODEV );
	}

	AVIRHIP_HIPCHECK( hipSetDevice( device ));
	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_init )

const char* avirhip_last_error( void )
{
	return( g_err );
}

const char* avirhip_version( void )
{
	return( "avirhip 0.1 (gfx950; avir v3.1 hot path)" );
}

int avirhip_plan_create( const avirhip_plan_desc* d, avirhip_plan** out )
try
{
	avirhip::clear_error();
	if( d == nullptr || out == nullptr )
	{
		set_error( "null argument" );
		return( AVIRHIP_EINVAL );
	}

	*out = nullptr;

	if( d -> src_w < 1 || d -> src_h < 1 || d -> new_w < 1 || d -> new_h < 1 ||
		d -> channels < 1 || d -> channels > 4 || d -> in_type < 0 ||
		d -> in_type > 3 || d -> out_type < 0 || d -> out_type > 3 )
	{
		set_error( "bad image geometry / types" );
		return( AVIRHIP_EINVAL );
	}

	if( !geometry_ok( "plan_create", d -> src_w, d -> src_h,
		d -> src_stride_elems, d -> new_w, d -> new_h, 0, d -> channels,
		d -> in_type, d -> out_type ))
	{
		return( AVIRHIP_EINVAL );
	}

	PlanHold hold( new_plan() );
	avirhip_plan* const p = hold.p;
	p -> src_w = d -> src_w; p -> src_h = d -> src_h;
	p -> src_stride = ( d -> src_stride_elems < 1 ?
		d -> src_w * d -> channels : d -> src_stride_elems );
	p -> new_w = d -> new_w; p -> new_h = d -> new_h;
	p -> new_stride = d -> new_w * d -> channels;
	p -> ch = d -> channels;
	p -> io_ch = d -> channels;
	p -> in_type = d -> in_type; p -> out_type = d -> out_type;
	p -> tr_mul = d -> tr_mul; p -> pk_out = d -> pk_out;
	p -> gamma = ( d -> use_srgb_gamma ? 1 : 0 );
	// the reference treats only AlphaIndex 0 and 3 of 4-channel pixels as an
	// alpha channel (avir.h:2859/2874, 3002/3016); any other value gamma-
	// converts all four channels
	p -> alpha_index = ( d -> channels == 4 && ( d -> alpha_index == 0 ||
		d -> alpha_index == 3 ) ? d -> alpha_index : -1 );

	if( d -> dither != AVIRHIP_DITHER_DEF && d -> dither != AVIRHIP_DITHER_ERRD &&
		d -> dither != AVIRHIP_DITHER_DEF_RNE )
	{
		set_error( "unknown ditherer %d", d -> dither );
		return( AVIRHIP_EINVAL );
	}

	// float / double output skips the dither stage (avir.h:5002-5025)
	p -> dither = ( d -> out_type == AVIRHIP_U8 || d -> out_type == AVIRHIP_U16 ?
		d -> dither : AVIRHIP_DITHER_DEF );
	// fpclass_float4 (AVIRHIP_DITHER_DEF_RNE): no in-place float output
	p -> fp4 = ( d -> dither == AVIRHIP_DITHER_DEF_RNE ? 1 : 0 );

	p -> f64 = ( d -> work_f64 ? 1 : 0 );

	if( p -> f64 && ( p -> dither != AVIRHIP_DITHER_DEF || p -> fp4 ))
	{
		set_error( "the double pipeline runs the default ditherer only" );
		return( AVIRHIP_EUNSUPPORTED );
	}

	int rc = lower_axis( d -> h, d -> src_w, d -> new_w, p -> h, p -> f64 != 0 );

	if( rc == 0 ) rc = lower_axis( d -> v, d -> src_h, d -> new_h, p -> v,
		p -> f64 != 0 );

	if( rc == 0 ) rc = finalize_avir_plan( p );

	if( rc != 0 )
	{
		return( rc );
	}

	*out = hold.release();
	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_plan_create )

} // extern "C"

namespace avirhip {

// Device side of an AVIR plan whose axes are lowered (host vectors filled):
// uploads, channel padding decision, fast-path preparation -- on the current
// device. Shared by plan creation and by the per-device replicas of
// avirhip_resize_sharded.
int finalize_avir_plan( avirhip_plan* p )
{
	int rc = upload_axis( p, p -> h );
	if( rc == 0 ) rc = upload_axis( p, p -> v );

	if( rc == 0 && p -> gamma )
	{
		std::vector< float > tbl( 256 );
		srgb_u8_table( tbl.data() );
		rc = upload( p, tbl, &p -> d_srgb_tbl );

		// uint8 output: the de-linearising output stage as a threshold table
		// (generic.hip; the default ditherer only)
		if( rc == 0 && p -> out_type == AVIRHIP_U8 &&
			p -> dither == AVIRHIP_DITHER_DEF )
		{
			std::vector< float > thr( 512 );

			if( gamma_u8_thresholds( 255.0f, p -> tr_mul != 1.0,
				(float) p -> tr_mul, (float) ( 1.0 / p -> tr_mul ),
				(float) p -> pk_out, thr.data() ))
			{
				rc = upload( p, thr, &p -> d_gthr );
			}
		}
	}

	if( rc == 0 && p -> f64 )
	{
		// the double pipeline: LDS-tiled two-pass kernels in double (tile64.hip;
		// reported as path 2), or -- plans with a filtered upsample, and path 1
		// on request -- one launch per op (generic64.hip)
		rc = tile64_prepare( p );
		p -> auto_path = ( rc == 0 && tile64_ok( p ) ? 2 : 1 );
		p -> fused_ok = ( rc == 0 && tile64_ok( p ) ? 1 : 0 );
		return( rc );
	}

	// 1-3 channel pixels: channels are independent, so executing them as
	// RGBA with zero padding (written by the pack stage, dropped by the
	// epilogue) is bit-identical and opens the RGBA fast paths -- 4/3 of the
	// arithmetic on kernels several times faster than the generic ones.
	if( rc == 0 && p -> io_ch < 4 )
	{
		p -> ch = 4;
	}

	if( rc == 0 ) rc = fused_prepare( p );
	if( rc == 0 ) rc = up2_prepare( p );
	if( rc == 0 ) rc = gpass_prepare( p );

	if( rc == 0 && gpass_ok( p ))
	{
		p -> fused_ok |= 8;

		// exact 2x keeps its marching kernel; otherwise the pass kernels
		// where they measured faster than the tiles (gpass_preferred)
		if( p -> auto_path != 4 && !fused_dn_both( p ) &&
			gpass_preferred( p ))
		{
			p -> auto_path = 5;
		}

		// exact 2x with integer pixels on both sides: the pass kernels read
		// and write the caller's images themselves, the marching kernel
		// needs a pack pass before and an epilogue pass behind it
		// (1920x1080 -> 3840x2160 RGB u8: 0.084 against 0.089 ms)
		// ... in general: whenever the marching kernel's result needs the
		// epilogue pass (integer or 1-3 channel pixels out; 1080p -> 4K:
		// RGB float 0.071 against 0.101 ms, RGB u8 -> float 0.070 / 0.102,
		// RGBA float -> u8 0.082 / 0.087); RGBA float output keeps it (a
		// uint8 RGBA source costs it one pack pass: 0.064 against 0.068)
		// Round 4: the marching kernel's vertical phase stores uint8 / uint16
		// / narrow float pixels itself (up2_stores_io): no epilogue, and from
		// 1080p sources on it is ahead of the pass kernels again (tools/
		// up2_io_sweep.py: 1920x1080 RGB u8 0.063 against 0.080 ms, u16 0.062 /
		// 0.102, RGBA float -> u8 0.053 / 0.084; 3840x2160 RGB u8 0.154 / 0.314;
		// 1280x720 RGB u8 0.046 / 0.044: the pass kernels keep small frames)
		const bool up2_io = ( p -> auto_path == 4 && up2_stores_io( p ) &&
			(long) p -> src_w * p -> src_h >= 1500000L );

		if( p -> auto_path == 4 && gpass_preferred( p ) && !p -> gamma &&
			!up2_io && p -> dither == AVIRHIP_DITHER_DEF &&
			!( p -> out_type == AVIRHIP_F32 && p -> io_ch == 4 ) &&
			( p -> out_type == AVIRHIP_U8 || p -> out_type == AVIRHIP_U16 ||
			p -> out_type == AVIRHIP_F32 ))
		{
			p -> auto_path = 5;
		}
	}

	if( rc == 0 && p -> ch != p -> io_ch && p -> auto_path == 1 )
	{
		// no fast path for this plan: run the generic kernels unpadded
		fused_release( p );
		tile64_release( p );
		up2_release( p );
		gpass_release( p );
		p -> ch = p -> io_ch;
		p -> fused_ok = 0;
	}

	return( rc );
}

static int finalize_lancir_plan( avirhip_plan* p )
{
	int rc = AVIRHIP_OK;
	LancirAxisDev* ax[ 2 ] = { &p -> lv, &p -> lh };

	for( int i = 0; i < 2 && rc == 0; i++ )
	{
		LancirAxisDev& L = *ax[ i ];
		if( rc == 0 ) rc = upload( p, L.h_flt, &L.d_flt );
		if( rc == 0 ) rc = upload( p, L.h_start, &L.d_start );
		if( rc == 0 ) rc = upload( p, L.h_fidx, &L.d_fidx );
	}

	if( rc == 0 ) rc = fused_prepare( p );
	if( rc == 0 ) rc = lanc2_prepare( p );
	if( rc == 0 ) rc = gpass_prepare( p );

	if( rc == 0 && gpass_ok( p ))
	{
		p -> fused_ok |= 8;

		// exact 2x: the marching kernel from 1080p sources on; smaller frames
		// take the pass kernels / the fused k_lf (tools/lanc2_io_sweep.py:
		// 640x480 float RGBA 0.012 against 0.023 ms, 1280x720 0.022 / 0.027,
		// 1920x1080 0.042 / 0.035; RGB uint8 0.043 / 0.050 at 720p, 0.072 /
		// 0.062 at 1080p, 0.255 / 0.140 at 2160p)
		if( p -> auto_path != 4 ||
			(long) p -> src_w * p -> src_h < 1500000L )
		{
			p -> auto_path = 5;
		}
	}

	// RGBA with integer elements or an output gain: the fast kernels compute
	// in float RGBA with unity gain -- run them in an inner plan of that
	// kind, between the pack pass and the output stage (both of which are
	// what the reference does around its float core, lancir.h:541-710)
	if( rc == 0 && p -> auto_path == 1 && p -> inner == nullptr &&
		!( p -> io_ch == 4 && p -> in_type == AVIRHIP_F32 &&
		p -> out_type == AVIRHIP_F32 && p -> l_unity ) &&
		getenv( "AVIRHIP_NO_INNER" ) == nullptr )
	{
		PlanHold qhold( new_plan() );
		avirhip_plan* const q = qhold.p;
		q -> device = p -> device;
		q -> is_lancir = 1;
		q -> src_w = p -> src_w; q -> src_h = p -> src_h;
		q -> src_stride = p -> src_w * 4;
		q -> new_w = p -> new_w; q -> new_h = p -> new_h;
		q -> new_stride = p -> new_w * 4;
		q -> ch = 4; q -> io_ch = 4;
		q -> in_type = AVIRHIP_F32; q -> out_type = AVIRHIP_F32;
		q -> lv = p -> lv; q -> lh = p -> lh; // (device pointers re-uploaded)
		// 1-3 channels run zero-padded to RGBA, in THEIR summation order
		q -> l_order = p -> io_ch;
		rc = finalize_lancir_plan( q );

		if( rc == 0 && q -> auto_path != 1 )
		{
			p -> inner = qhold.release();
			p -> auto_path = q -> auto_path;
			p -> fused_ok |= ( q -> fused_ok & ( 4 | 8 ));
		}
	}

	return( rc );
}

// A replica of plan `s` on device `device` (same tables, its own uploads,
// scratch and fast-path state).
static int clone_plan( const avirhip_plan* s, int device, avirhip_plan** out )
{
	AVIRHIP_HIPCHECK( hipSetDevice( device ));
	PlanHold hold( new_plan() );
	avirhip_plan* const q = hold.p;
	q -> is_lancir = s -> is_lancir;
	q -> src_w = s -> src_w; q -> src_h = s -> src_h;
	q -> src_stride = s -> src_stride;
	q -> new_w = s -> new_w; q -> new_h = s -> new_h;
	q -> new_stride = s -> new_stride;
	q -> ch = s -> io_ch; q -> io_ch = s -> io_ch;
	q -> in_type = s -> in_type; q -> out_type = s -> out_type;
	q -> tr_mul = s -> tr_mul; q -> pk_out = s -> pk_out;
	q -> gamma = s -> gamma; q -> alpha_index = s -> alpha_index;
	q -> dither = s -> dither;
	q -> fp4 = s -> fp4;
	q -> f64 = s -> f64;
	q -> l_out_mul = s -> l_out_mul; q -> l_clamp = s -> l_clamp;
	q -> l_unity = s -> l_unity;
	q -> l_order = s -> l_order;
	q -> h = s -> h; q -> v = s -> v;     // host vectors; device pointers are
	q -> lv = s -> lv; q -> lh = s -> lh; // replaced by the uploads below
	q -> path = s -> path;
	q -> variant = s -> variant;

	const int rc = ( s -> is_lancir ? finalize_lancir_plan( q ) :
		finalize_avir_plan( q ));

	if( rc != 0 )
	{
		return( rc );
	}

	*out = hold.release();
	return( AVIRHIP_OK );
}

static int grow( avirhip_plan* p, void** buf, size_t* have, size_t need )
{
	if( *have < need )
	{
		void* q;
		const int rc = dev_alloc( p, need, &q );
		if( rc != 0 ) return( rc );
		*buf = q;
		*have = need;
	}

	return( AVIRHIP_OK );
}

} // namespace avirhip

extern "C" {

/* one frame, contiguous output-row bands, one band per
 * entry of `devices`, stitched into `dst`. Bands are computed with global
 * indices, so the stitched frame is bit-identical to avirhip_resize(). */
int avirhip_resize_sharded( avirhip_plan* p, int n_gpus, const int* devices,
	const void* src, void* dst, int gather_root, double* t_compute_ms,
	double* t_gather_ms )
try
{
	avirhip::clear_error();
	if( p == nullptr || n_gpus < 1 || src == nullptr || dst == nullptr ||
		( devices == nullptr && n_gpus > 1 ) || gather_root < 0 ||
		gather_root >= n_gpus )
	{
		set_error( "resize_sharded: bad arguments" );
		return( AVIRHIP_EINVAL );
	}

	int ndev = 0;
	AVIRHIP_HIPCHECK( hipGetDeviceCount( &ndev ));
	const int src_mem = avirhip_resolve_mem( src, AVIRHIP_MEM_AUTO );
	const int dst_mem = avirhip_resolve_mem( dst, AVIRHIP_MEM_AUTO );
	int src_dev = -1;

	if( src_mem == AVIRHIP_MEM_DEVICE )
	{
		hipPointerAttribute_t at;
		AVIRHIP_HIPCHECK( hipPointerGetAttributes( &at, src ));
		src_dev = at.device;
	}

	int dst_dev = -1;

	if( dst_mem == AVIRHIP_MEM_DEVICE )
	{
		hipPointerAttribute_t at;
		AVIRHIP_HIPCHECK( hipPointerGetAttributes( &at, dst ));
		dst_dev = at.device;
	}

	const bool force_replica =
		( getenv( "AVIRHIP_SHARDED_FORCE_REPLICA" ) != nullptr );
	// test aid: AVIRHIP_SHARDED_DEVMAP="0,0,0" maps the LOGICAL devices of
	// `devices[]` (0, 1, 2) onto physical ones, so that the multi-device
	// branches -- one replica and one set of buffers per logical device, the
	// source copied to each, stores / copies into a peer's memory at the
	// band's offset -- execute on a box with a single GPU
	std::vector< int > devmap;
	{
		const char* dm = getenv( "AVIRHIP_SHARDED_DEVMAP" );

		while( dm != nullptr && *dm != 0 )
		{
			devmap.push_back( atoi( dm ));
			dm = strchr( dm, ',' );
			if( dm != nullptr ) dm++;
		}
	}

	const bool mocked = !devmap.empty();
	// test aid: always go through the band buffers + peer copies
	const bool force_staged =
		( getenv( "AVIRHIP_SHARDED_STAGED" ) != nullptr );

	const size_t esz_in = dtype_size( p -> in_type );
	const size_t esz_out = dtype_size( p -> out_type );
	const size_t src_bytes = ( (size_t) ( p -> src_h - 1 ) * p -> src_stride +
		(size_t) p -> src_w * p -> io_ch ) * esz_in;
	const size_t row_bytes = (size_t) p -> new_stride * esz_out;

	// the replica list and the per-replica band / source buffers are shared
	// by every sharded call on this plan; the caller's current device comes
	// back on every way out
	std::lock_guard< std::mutex > shard_lock( p -> shard_mtx );
	struct DeviceKeep
	{
		int keep;
		DeviceKeep() : keep( 0 ) { (void) hipGetDevice( &keep ); }
		~DeviceKeep() { (void) hipSetDevice( keep ); }
	} devkeep;
	std::vector< avirhip_plan* > pl( n_gpus );
	std::vector< int > r0( n_gpus ), r1( n_gpus );
	std::vector< char* > band( n_gpus );
	int rc = AVIRHIP_OK;
	const auto t0 = std::chrono::steady_clock::now();

	// ---- compute: every band on its device's default stream
	for( int g = 0; g < n_gpus && rc == 0; g++ )
	{
		// ld: the device as the caller names it; d: where it really is
		const int ld = ( devices != nullptr ? devices[ g ] : p -> device );
		const int d = ( mocked && ld >= 0 && ld < (int) devmap.size() ?
			devmap[ ld ] : ld );

		if( d < 0 || d >= ndev || ld < 0 )
		{
			set_error( "resize_sharded: device %d does not exist", ld );
			rc = AVIRHIP_EINVAL;
			break;
		}

		avirhip_plan* q = nullptr;

		if( ld == p -> device && !force_replica )
		{
			q = p;
		}
		else
		{
			for( size_t i = 0; i < p -> replicas.size(); i++ )
			{
				if( p -> replicas[ i ] -> shard_ldev == ld )
				{
					q = p -> replicas[ i ];
				}
			}

			if( q == nullptr )
			{
				if(( rc = clone_plan( p, d, &q )) != 0 ) break;
				q -> shard_ldev = ld;
				p -> replicas.push_back( q );
			}
		}

		pl[ g ] = q;
		r0[ g ] = (int) ( (long) p -> new_h * g / n_gpus );
		r1[ g ] = (int) ( (long) p -> new_h * ( g + 1 ) / n_gpus );
		AVIRHIP_HIPCHECK( hipSetDevice( d ));

		// the source where this device can read it
		const void* s = src;
		int smem = src_mem;

		// (mocked: the buffers live on logical device 0)
		if( src_mem == AVIRHIP_MEM_DEVICE && ( mocked ? ld != 0 : src_dev != d ))
		{
			if(( rc = grow( q, &q -> shard_src, &q -> shard_src_bytes,
				src_bytes )) != 0 ) break;

			AVIRHIP_HIPCHECK( hipMemcpyPeerAsync( q -> shard_src, d, src,
				src_dev, src_bytes, nullptr ));
			s = q -> shard_src;
		}

		// A device destination the band's device can address (its own memory,
		// or a peer's over xGMI): the kernels store the band straight into
		// `dst` -- the transfer rides under the compute, there is no gather
		// phase and no second pass over the band.
		bool direct = false;

		if( dst_mem == AVIRHIP_MEM_DEVICE && !force_staged )
		{
			if( mocked ? ld == 0 : d == dst_dev )
			{
				direct = true;
			}
			else
			if( mocked )
			{
				direct = true; // "peer" memory on the same physical device
			}
			else
			{
				int can = 0;

				if( hipDeviceCanAccessPeer( &can, d, dst_dev ) == hipSuccess &&
					can )
				{
					const hipError_t pe = hipDeviceEnablePeerAccess( dst_dev, 0 );
					direct = ( pe == hipSuccess ||
						pe == hipErrorPeerAccessAlreadyEnabled );
				}

				(void) hipGetLastError();
			}
		}

		if( direct )
		{
			band[ g ] = nullptr; // nothing to gather
			rc = exec_any( q, s, smem, (char*) dst + (size_t) r0[ g ] *
				row_bytes, AVIRHIP_MEM_DEVICE, r0[ g ], r1[ g ], nullptr );

			continue;
		}

		// bands that share a device (and plan) lie one after another in that
		// plan's band buffer: room for the whole frame covers every split
		if(( rc = grow( q, &q -> shard_band, &q -> shard_band_bytes,
			(size_t) p -> new_h * row_bytes + 16 )) != 0 ) break;

		band[ g ] = (char*) q -> shard_band + (size_t) r0[ g ] * row_bytes;
		rc = exec_any( q, s, smem, band[ g ], AVIRHIP_MEM_DEVICE, r0[ g ],
			r1[ g ], nullptr );
	}

	for( int g = 0; g < n_gpus && rc == 0; g++ )
	{
		AVIRHIP_HIPCHECK( hipSetDevice( pl[ g ] -> device ));
		AVIRHIP_HIPCHECK( hipStreamSynchronize( nullptr ));
	}

	const auto t1 = std::chrono::steady_clock::now();

	// ---- gather: band buffers -> dst (peer copies over xGMI, or D2H) for
	// the bands that could not be stored directly

	for( int g = 0; g < n_gpus && rc == 0; g++ )
	{
		const size_t bb = (size_t) ( r1[ g ] - r0[ g ]) * row_bytes;
		char* to = (char*) dst + (size_t) r0[ g ] * row_bytes;
		AVIRHIP_HIPCHECK( hipSetDevice( pl[ g ] -> device ));

		if( bb == 0 || band[ g ] == nullptr )
		{
			continue;
		}

		if( dst_mem == AVIRHIP_MEM_DEVICE )
		{
			AVIRHIP_HIPCHECK( hipMemcpyPeerAsync( to, dst_dev, band[ g ],
				pl[ g ] -> device, bb, nullptr ));
		}
		else
		{
			AVIRHIP_HIPCHECK( hipMemcpyAsync( to, band[ g ], bb,
				hipMemcpyDeviceToHost, nullptr ));
		}
	}

	for( int g = 0; g < n_gpus && rc == 0; g++ )
	{
		AVIRHIP_HIPCHECK( hipSetDevice( pl[ g ] -> device ));
		AVIRHIP_HIPCHECK( hipStreamSynchronize( nullptr ));
	}

	const auto t2 = std::chrono::steady_clock::now();

	if( t_compute_ms != nullptr )
	{
		*t_compute_ms = std::chrono::duration< double, std::milli >(
			t1 - t0 ).count();
	}

	if( t_gather_ms != nullptr )
	{
		*t_gather_ms = std::chrono::duration< double, std::milli >(
			t2 - t1 ).count();
	}

	return( rc );
}
AVIRHIP_CATCH( avirhip_resize_sharded )

static int lower_lancir_axis( avirhip_plan* p, const avirhip_lancir_axis& a,
	LancirAxisDev& L )
{
	if( a.kernel_len < 2 || ( a.kernel_len & 1 ) || a.n_filters < 1 ||
		a.filters == nullptr || a.pos == nullptr || a.dst_len < 1 ||
		a.src_len < 1 )
	{
		set_error( "malformed LANCIR axis" );
		return( AVIRHIP_EINVAL );
	}

	L.kernel_len = a.kernel_len; L.src_len = a.src_len;
	L.dst_len = a.dst_len; L.n_filters = a.n_filters;
	L.h_flt.assign( a.filters, a.filters + (size_t) a.n_filters *
		a.kernel_len );
	L.h_start.resize( a.dst_len );
	L.h_fidx.resize( a.dst_len );

	for( int j = 0; j < a.dst_len; j++ )
	{
		if( a.pos[ j ].flt_index < 0 || a.pos[ j ].flt_index >= a.n_filters )
		{
			set_error( "bad LANCIR pos[%d]", j );
			return( AVIRHIP_EINVAL );
		}

		// `so` is relative to the padded scanline; edge replication of the
		// padding (lancir.h:1541-1594, 1698-1734) becomes an index clamp.
		L.h_start[ j ] = a.pos[ j ].so - a.padl;
		L.h_fidx[ j ] = a.pos[ j ].flt_index;
	}

	return( AVIRHIP_OK );
}

int avirhip_lancir_plan_create( const avirhip_lancir_desc* d,
	avirhip_plan** out )
try
{
	avirhip::clear_error();
	if( d == nullptr || out == nullptr )
	{
		set_error( "null argument" );
		return( AVIRHIP_EINVAL );
	}

	*out = nullptr;

	if( d -> src_w < 1 || d -> src_h < 1 || d -> new_w < 1 || d -> new_h < 1 ||
		d -> channels < 1 || d -> channels > 4 )
	{
		set_error( "bad image geometry" );
		return( AVIRHIP_EINVAL );
	}

	if( d -> in_type < 0 || d -> in_type > AVIRHIP_U32 || d -> out_type < 0 ||
		d -> out_type > AVIRHIP_U32 )
	{
		set_error( "LANCIR: element types are uint8, uint16, uint32 (as "
			"uint16), float or double (lancir.h:373-381)" );
		return( AVIRHIP_EINVAL );
	}

	if( !geometry_ok( "lancir_plan_create", d -> src_w, d -> src_h,
		d -> src_stride_elems, d -> new_w, d -> new_h, d -> new_stride_elems,
		d -> channels, d -> in_type, d -> out_type ))
	{
		return( AVIRHIP_EINVAL );
	}

	PlanHold hold( new_plan() );
	avirhip_plan* const p = hold.p;
	p -> is_lancir = 1;
	p -> src_w = d -> src_w; p -> src_h = d -> src_h;
	p -> src_stride = ( d -> src_stride_elems < 1 ?
		d -> src_w * d -> channels : d -> src_stride_elems );
	p -> new_w = d -> new_w; p -> new_h = d -> new_h;
	p -> new_stride = ( d -> new_stride_elems < 1 ?
		d -> new_w * d -> channels : d -> new_stride_elems );
	p -> ch = d -> channels;
	p -> io_ch = d -> channels;
	p -> in_type = d -> in_type; p -> out_type = d -> out_type;
	p -> l_out_mul = d -> out_mul; p -> l_clamp = d -> clamp;
	p -> l_unity = d -> is_unity_mul;

	int rc = lower_lancir_axis( p, d -> v, p -> lv );
	if( rc == 0 ) rc = lower_lancir_axis( p, d -> h, p -> lh );
	if( rc == 0 ) rc = finalize_lancir_plan( p );

	if( rc != 0 )
	{
		return( rc );
	}

	*out = hold.release();
	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_lancir_plan_create )

void avirhip_plan_destroy( avirhip_plan* p )
{
	if( p == nullptr )
	{
		return;
	}

	for( size_t i = 0; i < p -> replicas.size(); i++ )
	{
		avirhip_plan_destroy( p -> replicas[ i ]);
	}

	p -> replicas.clear();

	for( size_t i = 0; i < p -> spares.size(); i++ )
	{
		avirhip_plan_destroy( p -> spares[ i ]);
	}

	p -> spares.clear();

	if( p -> last_done != nullptr )
	{
		(void) hipEventDestroy( p -> last_done );
	}

	for( size_t i = 0; i < p -> pipe_ev.size(); i++ )
	{
		(void) hipEventDestroy( p -> pipe_ev[ i ]);
	}

	p -> pipe_ev.clear();

	if( p -> pipe_in != nullptr )
	{
		(void) hipStreamDestroy( (hipStream_t) p -> pipe_in );
		(void) hipStreamDestroy( (hipStream_t) p -> pipe_out );
	}

	avirhip_plan_destroy( p -> inner );
	p -> inner = nullptr;
	fused_release( p );
	tile64_release( p );
	up2_release( p );
	lanc2_release( p );
	gpass_release( p );

	for( size_t i = 0; i < p -> allocs.size(); i++ )
	{
		(void) hipFree( p -> allocs[ i ]);
	}

	delete p;
}

int avirhip_plan_set_path( avirhip_plan* p, int path )
try
{
	avirhip::clear_error();
	if( p == nullptr || path < 0 || path > 5 )
	{
		set_error( "bad path" );
		return( AVIRHIP_EINVAL );
	}

	if(( path == 2 && !( p -> fused_ok & 1 )) ||
		( path == 3 && !( p -> fused_ok & 2 )) ||
		( path == 4 && !( p -> fused_ok & 4 )) ||
		( path == 5 && !( p -> fused_ok & 8 )) ||
		( p -> is_lancir && ( path == 2 || path == 3 )))
	{
		set_error( "path %d cannot run this plan", path );
		return( AVIRHIP_EUNSUPPORTED );
	}

	p -> path = path;

	// the plan's same-device spares and other-device replicas run what it runs
	{
		std::lock_guard< std::mutex > sl( p -> spare_mtx );

		for( size_t i = 0; i < p -> spares.size(); i++ )
		{
			p -> spares[ i ] -> path = path;
		}
	}
	{
		std::lock_guard< std::mutex > sl( p -> shard_mtx );

		for( size_t i = 0; i < p -> replicas.size(); i++ )
		{
			if( p -> replicas[ i ] != nullptr )
			{
				p -> replicas[ i ] -> path = path;
			}
		}
	}

	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_plan_set_path )

unsigned long long avirhip_plan_device_bytes( avirhip_plan* p )
{
	return( (unsigned long long) plan_device_bytes( p ));
}

int avirhip_plan_set_variant( avirhip_plan* p, int variant )
try
{
	avirhip::clear_error();
	if( p == nullptr || variant < 0 || variant > 63 )
	{
		set_error( "bad variant" );
		return( AVIRHIP_EINVAL );
	}

	p -> variant = variant;

	if( p -> inner != nullptr )
	{
		p -> inner -> variant = variant;
	}

	{
		std::lock_guard< std::mutex > sl( p -> spare_mtx );

		for( size_t i = 0; i < p -> spares.size(); i++ )
		{
			p -> spares[ i ] -> variant = variant;
		}
	}
	{
		std::lock_guard< std::mutex > sl( p -> shard_mtx );

		for( size_t i = 0; i < p -> replicas.size(); i++ )
		{
			if( p -> replicas[ i ] != nullptr )
			{
				p -> replicas[ i ] -> variant = variant;
			}
		}
	}

	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_plan_set_variant )

int avirhip_band_source_rows( const avirhip_plan* p, int row0, int row1,
	int* first, int* last )
try
{
	avirhip::clear_error();
	if( p == nullptr || first == nullptr || last == nullptr || row0 < 0 ||
		row1 > p -> new_h || row1 <= row0 )
	{
		set_error( "band_source_rows: bad arguments" );
		return( AVIRHIP_EINVAL );
	}

	band_src_rows( p, row0, row1, first, last );
	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_band_source_rows )

int avirhip_plan_get_path( const avirhip_plan* p )
try
{
	avirhip::clear_error();
	if( p == nullptr )
	{
		return( AVIRHIP_EINVAL );
	}

	return( p -> path != 0 ? p -> path : p -> auto_path );
}
AVIRHIP_CATCH( avirhip_plan_get_path )

int avirhip_resize( avirhip_plan* p, const void* src, int src_mem, void* dst,
	int dst_mem, void* stream )
try
{
	avirhip::clear_error();
	if( p == nullptr )
	{
		set_error( "null plan" );
		return( AVIRHIP_EINVAL );
	}

	return( exec_any( p, src, src_mem, dst, dst_mem, 0, p -> new_h, stream ));
}
AVIRHIP_CATCH( avirhip_resize )

int avirhip_resize_band( avirhip_plan* p, const void* src, int src_mem,
	void* dst_band, int dst_mem, int row0, int row1, void* stream )
try
{
	avirhip::clear_error();
	return( exec_any( p, src, src_mem, dst_band, dst_mem, row0, row1,
		stream ));
}
AVIRHIP_CATCH( avirhip_resize_band )

int avirhip_resize_window( avirhip_plan* p, const void* src_rows, int src_mem,
	int first_row, int n_rows, void* dst_band, int dst_mem, int row0, int row1,
	void* stream )
try
{
	avirhip::clear_error();
	if( p == nullptr || n_rows < 1 )
	{
		set_error( "resize_window: bad arguments" );
		return( AVIRHIP_EINVAL );
	}

	return( exec_any( p, src_rows, src_mem, dst_band, dst_mem, row0, row1,
		stream, false, first_row, n_rows ));
}
AVIRHIP_CATCH( avirhip_resize_window )

int avirhip_time_resize( avirhip_plan* p, const void* src, void* dst,
	int iters, void* stream, double* avg_ms )
try
{
	avirhip::clear_error();
	if( p == nullptr || iters < 1 || avg_ms == nullptr )
	{
		set_error( "bad arguments" );
		return( AVIRHIP_EINVAL );
	}

	hipStream_t st = (hipStream_t) stream;
	hipEvent_t e0, e1;
	AVIRHIP_HIPCHECK( hipEventCreate( &e0 ));
	AVIRHIP_HIPCHECK( hipEventCreate( &e1 ));
	AVIRHIP_HIPCHECK( hipEventRecord( e0, st ));

	for( int i = 0; i < iters; i++ )
	{
		int rc = exec_any( p, src, AVIRHIP_MEM_DEVICE, dst,
			AVIRHIP_MEM_DEVICE, 0, p -> new_h, stream );

		if( rc != 0 )
		{
			return( rc );
		}
	}

	AVIRHIP_HIPCHECK( hipEventRecord( e1, st ));
	AVIRHIP_HIPCHECK( hipEventSynchronize( e1 ));
	float ms = 0.0f;
	AVIRHIP_HIPCHECK( hipEventElapsedTime( &ms, e0, e1 ));
	(void) hipEventDestroy( e0 );
	(void) hipEventDestroy( e1 );
	*avg_ms = (double) ms / iters;
	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_time_resize )

} // extern "C"
