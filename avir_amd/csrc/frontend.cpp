// frontend.cpp -- C ABI of the front-end mirror (include/avirhip.h, second
// half): avirhip_resizer_* mirrors avir::CImageResizer<> (avir.h:4609-5092),
// avirhip_lancir_* mirrors avir::CLancIR (lancir.h:327-755). Each call runs the
// host planner (planner.cpp), caches the uploaded plan per geometry and
// executes it on the device. There is no CPU execution path.

#include "plan.h"
#include "planner.h"
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>

using namespace avirhip;

extern "C" int avirhip_resolve_mem( const void* ptr, int mem );

// Plans cached per geometry (and device), least recently used first out. A
// thumbnailer resizes many geometries through one object: without a bound
// every plan would keep its tables and full-frame scratch until the object
// dies. Bounds: AVIRHIP_CACHE_PLANS plans (default 64) and AVIRHIP_CACHE_BYTES
// of device memory (default 16 GiB). A plan is never dropped while a resize is
// using it, nor after its handle was handed out by *_get_plan().
template< class Key >
struct PlanCache
{
	struct Entry
	{
		avirhip_plan* p;
		unsigned long long tick;
		int busy;    // resizes in flight on the host side
		bool pinned; // handle given to the caller: owned until destruction
	};

	std::map< Key, Entry > m;
	unsigned long long tick = 0;

	Entry* find( const Key& k )
	{
		auto it = m.find( k );

		if( it == m.end() )
		{
			return( nullptr );
		}

		it -> second.tick = ++tick;
		return( &it -> second );
	}

	Entry* insert( const Key& k, avirhip_plan* p )
	{
		Entry e = { p, ++tick, 0, false };
		Entry* r = &( m[ k ] = e );
		trim( r );
		return( r );
	}

	static size_t limit( const char* name, size_t def )
	{
		const char* v = getenv( name );
		return( v != nullptr && atoll( v ) > 0 ? (size_t) atoll( v ) : def );
	}

	void trim( const Entry* keep )
	{
		static const size_t maxn = limit( "AVIRHIP_CACHE_PLANS", 64 );
		static const size_t maxb = limit( "AVIRHIP_CACHE_BYTES",
			(size_t) 16 << 30 );

		while( true )
		{
			size_t bytes = 0;
			auto victim = m.end();

			for( auto it = m.begin(); it != m.end(); ++it )
			{
				bytes += avirhip::plan_device_bytes( it -> second.p );

				if( &it -> second != keep && it -> second.busy == 0 &&
					!it -> second.pinned && ( victim == m.end() ||
					it -> second.tick < victim -> second.tick ))
				{
					victim = it;
				}
			}

			if(( m.size() <= maxn && bytes <= maxb ) || victim == m.end() )
			{
				return;
			}

			// (hipFree inside waits for the plan's queued work)
			avirhip_plan_destroy( victim -> second.p );
			m.erase( victim );
		}
	}

	void clear()
	{
		for( auto& c : m )
		{
			avirhip_plan_destroy( c.second.p );
		}

		m.clear();
	}
};

struct avirhip_resizer
{
	AvirPlanner* planner;
	int dither; // AVIRHIP_DITHER_*: fpclass::CDitherer of the mirrored object
	int fppack; // fpclass::fppack of the mirrored object (1 or 4)
	int f64;    // fpclass_def< double >: the double pipeline
	std::mutex mtx;
	typedef std::tuple< int, int, int, int, int, int, double, double, double,
		int, int, int, int, int, int, int > Key; // ..., ditherer + fppack, device
	PlanCache< Key > cache;
	std::map< Key, avirhip_vars_base > vbcache;
};

struct avirhip_lancir
{
	std::mutex mtx;
	typedef std::tuple< int, int, int, int, int, int, int, double, double,
		double, double, double, int, int, int > Key; // ..., device
	PlanCache< Key > cache;
};

static const avirhip_vars g_defvars = { 0.0, 0.0, 0, -1, -1, 0 };

extern "C" {

void avirhip_fill_lcg_u8( uint8_t* p, size_t n, uint32_t seed )
{
	uint32_t s = seed;

	for( size_t i = 0; i < n; i++ )
	{
		s = s * 1664525u + 1013904223u;
		p[ i ] = (uint8_t) ( s >> 24 );
	}
}

void avirhip_fill_lcg_f32( float* p, size_t n, uint32_t seed )
{
	uint32_t s = seed;

	for( size_t i = 0; i < n; i++ )
	{
		s = s * 1664525u + 1013904223u;
		p[ i ] = (float) ( s >> 8 ) * ( 1.0f / 16777216.0f );
	}
}

uint64_t avirhip_fnv1a64( const void* p, size_t n )
{
	const uint8_t* b = (const uint8_t*) p;
	uint64_t h = 1469598103934665603ULL;

	for( size_t i = 0; i < n; i++ )
	{
		h ^= b[ i ];
		h *= 1099511628211ULL;
	}

	return( h );
}

void avirhip_params_preset( int preset, avirhip_params* p )
{
	if( p == nullptr )
	{
		return;
	}

	// Literals of CImageResizerParams* (avir.h:2300-2464).
	static const double t[ 6 ][ 8 ] = {
		{ 0.97946, 6.4262, 6.41341, 0.7372, 18, 4.76449, 7.55999999999998,
			0.79285 },
		{ 0.95521, 5.70774, 1.00766, 0.74202, 18, 1.6801, 6.62, 0.67821 },
		{ 1, 5.865, 1.79529, 0.74325, 18, 1.87597, 6.89999999999999,
			0.69326 },
		{ 0.99739, 6.20326, 4.6836, 0.73879, 18, 7.86565, 6.91999999999999,
			0.78379 },
		{ 0.97433, 6.87893, 7.74731, 0.73844, 18, 4.8149, 8.07999999999996,
			0.79335 },
		{ 0.99705, 7.42695, 1.71985, 0.7571, 18, 6.71313, 8.27999999999996,
			0.78413 } };

	if( preset < 0 || preset > 5 )
	{
		preset = 0;
	}

	p -> CorrFltAlpha = t[ preset ][ 0 ];
	p -> CorrFltLen = t[ preset ][ 1 ];
	p -> IntFltAlpha = t[ preset ][ 2 ];
	p -> IntFltCutoff = t[ preset ][ 3 ];
	p -> IntFltLen = t[ preset ][ 4 ];
	p -> LPFltAlpha = t[ preset ][ 5 ];
	p -> LPFltBaseLen = t[ preset ][ 6 ];
	p -> LPFltCutoffMult = t[ preset ][ 7 ];
	p -> HBFltAlpha = 1.94609;
	p -> HBFltCutoff = 0.46437;
	p -> HBFltLen = 24;
}

void avirhip_vars_default( avirhip_vars* v )
{
	if( v != nullptr )
	{
		*v = g_defvars;
	}
}

int avirhip_resizer_create( int res_bit_depth, int src_bit_depth,
	const avirhip_params* params, avirhip_resizer** out )
try
{
	avirhip::clear_error();
	if( out == nullptr || res_bit_depth < 1 || res_bit_depth > 16 ||
		src_bit_depth < 0 || src_bit_depth > 16 )
	{
		set_error( "resizer_create: bad arguments" );
		return( AVIRHIP_EINVAL );
	}

	avirhip_params P;

	if( params == nullptr )
	{
		avirhip_params_preset( AVIRHIP_PARAMS_DEF, &P );
	}
	else
	{
		P = *params;
	}

	std::unique_ptr< AvirPlanner > pl( new AvirPlanner( res_bit_depth,
		src_bit_depth, P ));
	avirhip_resizer* r = new avirhip_resizer();
	r -> planner = pl.release();
	r -> dither = AVIRHIP_DITHER_DEF;
	r -> fppack = 1;
	r -> f64 = 0;
	*out = r;
	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_resizer_create )

int avirhip_resizer_set_ditherer( avirhip_resizer* r, int dither )
try
{
	avirhip::clear_error();
	if( r == nullptr || ( dither != AVIRHIP_DITHER_DEF &&
		dither != AVIRHIP_DITHER_ERRD ))
	{
		set_error( "set_ditherer: bad arguments" );
		return( AVIRHIP_EINVAL );
	}

	std::lock_guard< std::mutex > lock( r -> mtx );
	r -> dither = dither;
	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_resizer_set_ditherer )

int avirhip_resizer_set_fpclass( avirhip_resizer* r, int fppack )
try
{
	avirhip::clear_error();
	if( r == nullptr || ( fppack != 1 && fppack != 4 &&
		fppack != AVIRHIP_FPCLASS_DOUBLE ))
	{
		set_error( "set_fpclass: fppack 1 (float), 4 (float4) or "
			"AVIRHIP_FPCLASS_DOUBLE only" );
		return( AVIRHIP_EINVAL );
	}

	std::lock_guard< std::mutex > lock( r -> mtx );
	r -> f64 = ( fppack == AVIRHIP_FPCLASS_DOUBLE ? 1 : 0 );
	r -> fppack = ( r -> f64 ? 1 : fppack );
	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_resizer_set_fpclass )

void avirhip_resizer_destroy( avirhip_resizer* r )
{
	if( r == nullptr )
	{
		return;
	}

	r -> cache.clear();
	delete r -> planner;
	delete r;
}

int avirhip_resizer_build_desc( avirhip_resizer* r, int src_w, int src_h,
	int src_scanline_size, int new_w, int new_h, int el_count_io, double k,
	const avirhip_vars* vars, int in_type, int out_type,
	avirhip_plan_desc** out )
try
{
	avirhip::clear_error();
	if( r == nullptr || out == nullptr )
	{
		set_error( "build_desc: null argument" );
		return( AVIRHIP_EINVAL );
	}

	if( !geometry_ok( "build_desc", src_w, src_h, src_scanline_size, new_w,
		new_h, 0, el_count_io, in_type, out_type ))
	{
		return( AVIRHIP_EINVAL );
	}

	const avirhip_vars& V = ( vars == nullptr ? g_defvars : *vars );
	DescStore* S = r -> planner -> build( src_w, src_h, src_scanline_size,
		new_w, new_h, el_count_io, k, V, in_type, out_type, r -> fppack,
		r -> f64 != 0 );

	if( S == nullptr )
	{
		return( AVIRHIP_EINVAL );
	}

	if(( r -> fppack == 4 || r -> f64 ) && r -> dither != AVIRHIP_DITHER_DEF )
	{
		delete S;
		set_error( "fpclass_float4 / fpclass_def<double>: only the default "
			"ditherer is built" );
		return( AVIRHIP_EUNSUPPORTED );
	}

	S -> d.dither = ( r -> fppack == 4 ? AVIRHIP_DITHER_DEF_RNE :
		r -> dither );
	*out = &S -> d;
	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_resizer_build_desc )

void avirhip_plan_desc_free( avirhip_plan_desc* d )
{
	if( d != nullptr )
	{
		delete (DescStore*) ( (char*) d - offsetof( DescStore, d ));
	}
}

int avirhip_resizer_band_source_rows( avirhip_resizer* r, int src_w, int src_h,
	int src_scanline_size, int new_w, int new_h, int el_count_io, double k,
	const avirhip_vars* vars, int in_type, int out_type, int row0, int row1,
	int* first, int* last )
try
{
	avirhip::clear_error();
	if( r == nullptr || first == nullptr || last == nullptr || row0 < 0 ||
		row1 > new_h || row1 <= row0 )
	{
		set_error( "resizer_band_source_rows: bad arguments" );
		return( AVIRHIP_EINVAL );
	}

	avirhip_plan_desc* d = nullptr;
	int rc = avirhip_resizer_build_desc( r, src_w, src_h, src_scanline_size,
		new_w, new_h, el_count_io, k, vars, in_type, out_type, &d );

	if( rc != 0 )
	{
		return( rc );
	}

	struct Free { avirhip_plan_desc* d; ~Free() { avirhip_plan_desc_free( d ); } }
		fr = { d };

	return( desc_band_src_rows( d, row0, row1, first, last ));
}
AVIRHIP_CATCH( avirhip_resizer_band_source_rows )

// Finds or builds the plan of a call. `pin`: the handle leaves the library
// (never evicted); otherwise the entry is marked busy until release().
static int resizer_acquire( avirhip_resizer* r, int src_w, int src_h,
	int src_scanline_size, int new_w, int new_h, int el_count_io, double k,
	const avirhip_vars* vars, int in_type, int out_type, bool pin,
	avirhip_plan** out )
{
	const avirhip_vars& V = ( vars == nullptr ? g_defvars : *vars );

	if( src_scanline_size < 1 )
	{
		src_scanline_size = src_w * el_count_io;
	}

	int dev = 0;
	(void) hipGetDevice( &dev ); // a plan lives on the device it was made on

	std::lock_guard< std::mutex > lock( r -> mtx );
	const avirhip_resizer::Key key( src_w, src_h, src_scanline_size, new_w,
		new_h, el_count_io, k, V.ox, V.oy, V.BuildMode, in_type, out_type,
		( V.UseSRGBGamma ? 1 : 0 ), V.AlphaIndex,
		r -> dither + 16 * r -> fppack + 1024 * r -> f64, dev );

	auto* e = r -> cache.find( key );

	if( e == nullptr )
	{
		avirhip_plan_desc* d = nullptr;
		int rc = avirhip_resizer_build_desc( r, src_w, src_h,
			src_scanline_size, new_w, new_h, el_count_io, k, &V, in_type,
			out_type, &d );

		if( rc != 0 )
		{
			return( rc );
		}

		avirhip_plan* p = nullptr;
		rc = avirhip_plan_create( d, &p );
		avirhip_plan_desc_free( d );

		if( rc != 0 )
		{
			return( rc );
		}

		PlanHold hold( p ); // (the cache's map node may fail to allocate)
		e = r -> cache.insert( key, p );
		hold.release();
	}

	if( pin )
	{
		e -> pinned = true;
	}
	else
	{
		e -> busy++;
	}

	*out = e -> p;
	return( AVIRHIP_OK );
}

static void resizer_release( avirhip_resizer* r, avirhip_plan* p )
{
	std::lock_guard< std::mutex > lock( r -> mtx );

	for( auto& c : r -> cache.m )
	{
		if( c.second.p == p )
		{
			c.second.busy--;
		}
	}

	r -> cache.trim( nullptr ); // scratch may have grown during the call
}

int avirhip_resizer_get_plan( avirhip_resizer* r, int src_w, int src_h,
	int src_scanline_size, int new_w, int new_h, int el_count_io, double k,
	const avirhip_vars* vars, int in_type, int out_type, avirhip_plan** out )
try
{
	avirhip::clear_error();
	if( r == nullptr || out == nullptr )
	{
		set_error( "get_plan: null argument" );
		return( AVIRHIP_EINVAL );
	}

	return( resizer_acquire( r, src_w, src_h, src_scanline_size, new_w, new_h,
		el_count_io, k, vars, in_type, out_type, true, out ));
}
AVIRHIP_CATCH( avirhip_resizer_get_plan )

// CImageResizerVarsBase after resizeImage() (avir.h:2473-2506), from the
// vertical axis of the plan description. The coordinate pair (k, o) walks the
// step list (an upsampling step scales both, a filter step divides them and
// shifts o by its edge pixels); the two flip-flop scanline buffers must hold
// the largest prefix and the largest length + suffix of every step that reads
// or writes them (steps alternate between the buffers, the last one writes
// the destination).
static void fill_vars_base( const avirhip_plan_desc& d, double k0,
	const avirhip_vars& V, avirhip_vars_base* out )
{
	avirhip_vars_base& B = *out;
	memset( &B, 0, sizeof( B ));
	B.ElCount = d.channels; B.ElCountIO = d.channels;
	B.fppack = 1; B.fpalign = 4; B.elalign = 1; B.packmode = 0;

	double k, o = V.oy;

	if( k0 == 0.0 )
	{
		k = (double) d.src_h / d.new_h;
		o += ( k - 1.0 ) * 0.5;
	}
	else
	if( k0 > 0.0 )
	{
		k = k0;
		o += ( k0 - 1.0 ) * 0.5;
	}
	else
	{
		k = -k0;
	}

	int maxpre[ 2 ] = { 0, 0 }, maxlen[ 2 ] = { 0, 0 };
	const int n = d.v.n_steps;

	for( int i = 0; i < n; i++ )
	{
		const avirhip_step& s = d.v.steps[ i ];
		const int ib = i & 1;
		const bool up = ( s.kind == AVIRHIP_STEP_UP_ZEROSTUFF ||
			s.kind == AVIRHIP_STEP_UP_FILTERED );

		if( up )
		{
			k *= s.resample_factor;
			o *= s.resample_factor;
		}
		else
		if( s.kind == AVIRHIP_STEP_FIR )
		{
			k /= s.resample_factor;
			o /= s.resample_factor;
			o += s.edge_pixel_count;
		}
		else
		{
			B.ResizeStep = i;
			B.IsResize2 = ( s.kind == AVIRHIP_STEP_RESIZE2 ? 1 : 0 );
		}

		maxpre[ ib ] = std::max( maxpre[ ib ], s.in_prefix );
		maxlen[ ib ] = std::max( maxlen[ ib ], s.in_len + s.in_suffix );

		if( i + 1 < n )
		{
			const int ob = ib ^ 1;

			if( up )
			{
				maxpre[ ob ] = std::max( maxpre[ ob ], s.out_prefix );
				maxlen[ ob ] = std::max( maxlen[ ob ],
					s.out_len + s.out_suffix );
			}
			else
			{
				maxlen[ ob ] = std::max( maxlen[ ob ], s.out_len );
			}
		}
	}

	for( int i = 0; i < 2; i++ )
	{
		B.BufLen[ i ] = ( maxpre[ i ] + maxlen[ i ]) * d.channels;
		B.BufOffs[ i ] = maxpre[ i ] * d.channels;
	}

	B.k = k; B.o = o;
	B.gamma_valid = ( V.UseSRGBGamma ? 1 : 0 );

	if( V.UseSRGBGamma )
	{
		// avir.h:4744-4763
		B.InGammaMult = ( d.in_type == AVIRHIP_U8 ? 1.0 / 255.0 :
			( d.in_type == AVIRHIP_U16 ? 1.0 / 65535.0 : 1.0 ));

		B.OutGammaMult = ( d.out_type == AVIRHIP_U8 ? 255.0 :
			( d.out_type == AVIRHIP_U16 ? 65535.0 : 1.0 ));
	}
}

int avirhip_resizer_vars_base( avirhip_resizer* r, int src_w, int src_h,
	int src_scanline_size, int new_w, int new_h, int el_count_io, double k,
	const avirhip_vars* vars, int in_type, int out_type,
	avirhip_vars_base* out )
try
{
	avirhip::clear_error();
	if( r == nullptr || out == nullptr )
	{
		set_error( "vars_base: null argument" );
		return( AVIRHIP_EINVAL );
	}

	const avirhip_vars& V = ( vars == nullptr ? g_defvars : *vars );

	if( src_scanline_size < 1 )
	{
		src_scanline_size = src_w * el_count_io;
	}

	const avirhip_resizer::Key key( src_w, src_h, src_scanline_size, new_w,
		new_h, el_count_io, k, V.ox, V.oy, V.BuildMode, in_type, out_type,
		( V.UseSRGBGamma ? 1 : 0 ), V.AlphaIndex,
		16 * r -> fppack + 1024 * r -> f64, 0 );

	{
		std::lock_guard< std::mutex > lock( r -> mtx );
		auto it = r -> vbcache.find( key );

		if( it != r -> vbcache.end() )
		{
			*out = it -> second;
			return( AVIRHIP_OK );
		}
	}

	avirhip_plan_desc* d = nullptr;
	int rc = avirhip_resizer_build_desc( r, src_w, src_h, src_scanline_size,
		new_w, new_h, el_count_io, k, &V, in_type, out_type, &d );

	if( rc != 0 )
	{
		return( rc );
	}

	fill_vars_base( *d, k, V, out );

	if( d -> work_f64 )
	{
		out -> fpalign = 8; // sizeof( double ), avir.h:4578
	}

	if( d -> dither == AVIRHIP_DITHER_DEF_RNE )
	{
		// fpclass_float4: a pixel is one 16-byte fptype (avir.h:4576-4579,
		// 4786-4787); buffer lengths count fptype elements
		const int ch = d -> channels;
		out -> ElCount = ( ch + 3 ) / 4;
		out -> fppack = 4; out -> fpalign = 16;

		for( int i = 0; i < 2; i++ )
		{
			out -> BufLen[ i ] = out -> BufLen[ i ] / ch * out -> ElCount;
			out -> BufOffs[ i ] = out -> BufOffs[ i ] / ch * out -> ElCount;
		}
	}

	avirhip_plan_desc_free( d );

	std::lock_guard< std::mutex > lock( r -> mtx );

	if( r -> vbcache.size() >= 256 )
	{
		r -> vbcache.clear();
	}

	r -> vbcache[ key ] = *out;
	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_resizer_vars_base )

int avirhip_resizer_resize( avirhip_resizer* r, const void* src, int src_mem,
	int src_w, int src_h, int src_scanline_size, void* dst, int dst_mem,
	int new_w, int new_h, int el_count_io, double k, const avirhip_vars* vars,
	int in_type, int out_type, void* stream )
try
{
	avirhip::clear_error();
	if( r == nullptr || out_type < 0 || out_type > 3 )
	{
		set_error( "resize: bad arguments" );
		return( AVIRHIP_EINVAL );
	}

	// avir.h:4686-4697: zero-sized source zero-fills NewWidth*NewHeight
	// ELEMENTS (sic, not x channels); zero-sized destination is a no-op.
	if( src_w == 0 || src_h == 0 )
	{
		if( dst == nullptr || new_w < 0 || new_h < 0 )
		{
			set_error( "resize: bad destination" );
			return( AVIRHIP_EINVAL );
		}

		const size_t n = (size_t) new_w * (size_t) new_h *
			dtype_size( out_type );

		dst_mem = avirhip_resolve_mem( dst, dst_mem );

		if( dst_mem == AVIRHIP_MEM_HOST )
		{
			memset( dst, 0, n );
		}
		else
		{
			AVIRHIP_HIPCHECK( hipMemsetAsync( dst, 0, n,
				(hipStream_t) stream ));
		}

		return( AVIRHIP_OK );
	}
	else
	if( new_w == 0 || new_h == 0 )
	{
		return( AVIRHIP_OK );
	}

	avirhip_plan* p = nullptr;
	int rc = resizer_acquire( r, src_w, src_h, src_scanline_size, new_w,
		new_h, el_count_io, k, vars, in_type, out_type, false, &p );

	if( rc != 0 )
	{
		return( rc );
	}

	rc = avirhip_resize( p, src, src_mem, dst, dst_mem, stream );
	resizer_release( r, p );
	return( rc );
}
AVIRHIP_CATCH( avirhip_resizer_resize )

void avirhip_lancir_params_default( avirhip_lancir_params* p )
{
	if( p != nullptr )
	{
		p -> SrcSSize = 0; p -> NewSSize = 0;
		p -> kx = 0.0; p -> ky = 0.0; p -> ox = 0.0; p -> oy = 0.0;
		p -> la = 3.0;
	}
}

int avirhip_lancir_create( avirhip_lancir** out )
try
{
	avirhip::clear_error();
	if( out == nullptr )
	{
		set_error( "lancir_create: null argument" );
		return( AVIRHIP_EINVAL );
	}

	*out = new avirhip_lancir();
	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_lancir_create )

void avirhip_lancir_destroy( avirhip_lancir* l )
{
	if( l == nullptr )
	{
		return;
	}

	l -> cache.clear();
	delete l;
}

int avirhip_lancir_build_desc( avirhip_lancir* l, int src_w, int src_h,
	int new_w, int new_h, int el_count, const avirhip_lancir_params* params,
	int in_type, int out_type, avirhip_lancir_desc** out )
try
{
	avirhip::clear_error();
	if( out == nullptr )
	{
		set_error( "lancir_build_desc: null argument" );
		return( AVIRHIP_EINVAL );
	}

	avirhip_lancir_params P;

	if( params == nullptr )
	{
		avirhip_lancir_params_default( &P );
	}
	else
	{
		P = *params;
	}

	if( !geometry_ok( "lancir_build_desc", src_w, src_h, P.SrcSSize, new_w,
		new_h, P.NewSSize, el_count, in_type, out_type ))
	{
		return( AVIRHIP_EINVAL );
	}

	DescStore* S = lancir_build( src_w, src_h, new_w, new_h, el_count, P,
		in_type, out_type );

	if( S == nullptr )
	{
		return( AVIRHIP_EINVAL );
	}

	*out = &S -> ld;
	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_lancir_build_desc )

void avirhip_lancir_desc_free( avirhip_lancir_desc* d )
{
	if( d != nullptr )
	{
		delete (DescStore*) ( (char*) d - offsetof( DescStore, ld ));
	}
}

int avirhip_lancir_band_source_rows( avirhip_lancir* l, int src_w, int src_h,
	int new_w, int new_h, int el_count, const avirhip_lancir_params* params,
	int in_type, int out_type, int row0, int row1, int* first, int* last )
try
{
	avirhip::clear_error();
	if( first == nullptr || last == nullptr || row0 < 0 || row1 > new_h ||
		row1 <= row0 )
	{
		set_error( "lancir_band_source_rows: bad arguments" );
		return( AVIRHIP_EINVAL );
	}

	avirhip_lancir_desc* d = nullptr;
	const int rc = avirhip_lancir_build_desc( l, src_w, src_h, new_w, new_h,
		el_count, params, in_type, out_type, &d );

	if( rc != 0 )
	{
		return( rc );
	}

	// the vertical windows in un-padded coordinates, clamped to the image
	// (edge replication of the padding, lancir.h:1541-1594)
	const avirhip_lancir_axis& v = d -> v;
	const int a = v.pos[ row0 ].so - v.padl;
	const int b = v.pos[ row1 - 1 ].so - v.padl + v.kernel_len - 1;
	*first = std::max( 0, std::min( a, src_h - 1 ));
	*last = std::max( 0, std::min( b, src_h - 1 ));
	avirhip_lancir_desc_free( d );
	return( AVIRHIP_OK );
}
AVIRHIP_CATCH( avirhip_lancir_band_source_rows )

static int lancir_acquire( avirhip_lancir* l, int src_w, int src_h,
	int new_w, int new_h, int el_count, const avirhip_lancir_params* params,
	int in_type, int out_type, bool pin, avirhip_plan** out )
{
	avirhip_lancir_params P;

	if( params == nullptr )
	{
		avirhip_lancir_params_default( &P );
	}
	else
	{
		P = *params;
	}

	int dev = 0;
	(void) hipGetDevice( &dev );

	const avirhip_lancir::Key key( src_w, src_h, new_w, new_h, el_count,
		P.SrcSSize, P.NewSSize, P.kx, P.ky, P.ox, P.oy, P.la, in_type,
		out_type, dev );

	std::lock_guard< std::mutex > lock( l -> mtx );
	auto* e = l -> cache.find( key );

	if( e == nullptr )
	{
		avirhip_lancir_desc* d = nullptr;
		int rc = avirhip_lancir_build_desc( l, src_w, src_h, new_w, new_h,
			el_count, &P, in_type, out_type, &d );

		if( rc != 0 )
		{
			return( rc );
		}

		avirhip_plan* p = nullptr;
		rc = avirhip_lancir_plan_create( d, &p );
		avirhip_lancir_desc_free( d );

		if( rc != 0 )
		{
			return( rc );
		}

		PlanHold hold( p );
		e = l -> cache.insert( key, p );
		hold.release();
	}

	if( pin )
	{
		e -> pinned = true;
	}
	else
	{
		e -> busy++;
	}

	*out = e -> p;
	return( AVIRHIP_OK );
}

static void lancir_release( avirhip_lancir* l, avirhip_plan* p )
{
	std::lock_guard< std::mutex > lock( l -> mtx );

	for( auto& c : l -> cache.m )
	{
		if( c.second.p == p )
		{
			c.second.busy--;
		}
	}

	l -> cache.trim( nullptr );
}

int avirhip_lancir_get_plan( avirhip_lancir* l, int src_w, int src_h,
	int new_w, int new_h, int el_count, const avirhip_lancir_params* params,
	int in_type, int out_type, avirhip_plan** out )
try
{
	avirhip::clear_error();
	if( l == nullptr || out == nullptr )
	{
		set_error( "lancir_get_plan: null argument" );
		return( AVIRHIP_EINVAL );
	}

	return( lancir_acquire( l, src_w, src_h, new_w, new_h, el_count, params,
		in_type, out_type, true, out ));
}
AVIRHIP_CATCH( avirhip_lancir_get_plan )

int avirhip_lancir_resize( avirhip_lancir* l, const void* src, int src_mem,
	int src_w, int src_h, void* dst, int dst_mem, int new_w, int new_h,
	int el_count, const avirhip_lancir_params* params, int in_type,
	int out_type, void* stream )
try
{
	avirhip::clear_error();
	// Parameter errors return 0, lancir.h:392-407.
	if( l == nullptr || src_w < 0 || src_h < 0 || new_w <= 0 || new_h <= 0 ||
		src == nullptr || dst == nullptr || src == dst ||
		( params != nullptr && params -> la < 2.0 ))
	{
		return( 0 );
	}

	if( src_w == 0 || src_h == 0 )
	{
		// Zero-filled rows honouring NewSSize, lancir.h:413-425.
		const size_t osl = (size_t) new_w * el_count;
		const size_t nss = ( params != nullptr && params -> NewSSize >= 1 ?
			(size_t) params -> NewSSize : osl );
		const size_t es = dtype_size( out_type );
		dst_mem = avirhip_resolve_mem( dst, dst_mem );

		for( int i = 0; i < new_h; i++ )
		{
			char* op = (char*) dst + (size_t) i * nss * es;

			if( dst_mem == AVIRHIP_MEM_HOST )
			{
				memset( op, 0, osl * es );
			}
			else
			{
				AVIRHIP_HIPCHECK( hipMemsetAsync( op, 0, osl * es,
					(hipStream_t) stream ));
			}
		}

		return( new_h );
	}

	avirhip_plan* p = nullptr;
	int rc = lancir_acquire( l, src_w, src_h, new_w, new_h, el_count, params,
		in_type, out_type, false, &p );

	if( rc != 0 )
	{
		return( rc );
	}

	rc = avirhip_resize( p, src, src_mem, dst, dst_mem, stream );
	lancir_release( l, p );
	return( rc != 0 ? rc : new_h );
}
AVIRHIP_CATCH( avirhip_lancir_resize )

} // extern "C"
