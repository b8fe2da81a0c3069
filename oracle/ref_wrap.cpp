/*
 * oracle/ref_wrap.cpp -- TEST INFRASTRUCTURE ONLY (never linked into, imported
 * or called by the product path).
 *
 * Thin C wrapper that compiles the UNMODIFIED reference headers from where
 * they lie (/root/reference/avir.h, lancir.h, avir_float4_sse.h) into
 * oracle/_ref/libavir_ref.so. Nothing is copied from the reference; the
 * headers are only #included at build time (see oracle/Makefile). The built
 * .so travels to the GPU box with the repo snapshot; /root/reference itself
 * does not exist there.
 *
 * It serves three purposes:
 *   1. the bit-exact parity target (avir::CImageResizer<fpclass_def<float>>
 *      and avir::CLancIR, built -ffp-contract=off),
 *   2. a dump of the reference planner's tables in the product's plan format
 *      (include/avirhip.h) so that plan parity can be checked table by table,
 *   3. the "reference" CPU baseline timed by bench.py (incl. a std::thread
 *      implementation of the abstract avir::CImageResizerThreadPool,
 *      avir.h:2119-2225, which the reference does not ship).
 *
 * The `#define private public` trick exposes the
 * planner internals; libstdc++ headers are included first so only the
 * reference's own classes are affected.
 */

#include <cstring>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <algorithm>
#include <map>
#include <emmintrin.h>
#if defined( __AVX__ )
#include <immintrin.h>
#endif

#define private public
#define protected public
#include "avir.h"
#include "avir_float4_sse.h"
#if defined( REF_SIMD ) && defined( __AVX__ )
#include "avir_float8_avx.h"
#endif
#include "lancir.h"
#undef private
#undef protected

#include "../include/avirhip.h"

namespace {

/* Implementation of the reference's abstract thread pool (avir.h:2119-2225)
 * on PERSISTENT worker threads: they are created once per pool and parked on a
 * condition variable between the passes of resizeImage() (H pass, V pass,
 * output pass each call startAllWorkloads() once), so a pass costs a wake-up,
 * not a thread creation per workload. */
class CRefThreadPool : public avir::CImageResizerThreadPool
{
public:
	explicit CRefThreadPool( int n )
		: Count( n ), Gen( 0 ), Pending( 0 ), Quit( false )
	{
		for( int i = 0; i + 1 < n; i++ )
		{
			Threads.emplace_back( [this, i]() { worker( i ); } );
		}
	}

	~CRefThreadPool()
	{
		{
			std::unique_lock< std::mutex > l( M );
			Quit = true;
		}

		Wake.notify_all();
		for( auto& t : Threads ) t.join();
	}

	virtual int getSuggestedWorkloadCount() const { return( Count ); }
	virtual void addWorkload( CWorkload* const w ) { Workloads.push_back( w ); }

	virtual void startAllWorkloads()
	{
		std::unique_lock< std::mutex > l( M );
		Pending = (int) Workloads.size();
		Gen++;
		Wake.notify_all();
	}

	virtual void waitAllWorkloadsToFinish()
	{
		std::unique_lock< std::mutex > l( M );
		Done.wait( l, [this]() { return( Pending == 0 ); } );
	}

	virtual void removeAllWorkloads() { Workloads.clear(); }

private:
	void worker( const int idx )
	{
		unsigned long seen = 0;

		while( true )
		{
			CWorkload* w = nullptr;
			{
				std::unique_lock< std::mutex > l( M );
				Wake.wait( l, [&]() { return( Quit || Gen != seen ); } );

				if( Quit )
				{
					return;
				}

				seen = Gen;

				if( idx < (int) Workloads.size() )
				{
					w = Workloads[ idx ];
				}
			}

			if( w != nullptr )
			{
				w -> process();
				std::unique_lock< std::mutex > l( M );

				if( --Pending == 0 )
				{
					Done.notify_all();
				}
			}
		}
	}

	int Count;
	std::vector< CWorkload* > Workloads;
	std::vector< std::thread > Threads;
	std::mutex M;
	std::condition_variable Wake, Done;
	unsigned long Gen;
	int Pending;
	bool Quit;
};

void fillParams( avir::CImageResizerParams& P, const double* p )
{
	avir::CImageResizerParamsDef D;
	P = D;
	if( p != nullptr )
	{
		P.CorrFltAlpha = p[ 0 ]; P.CorrFltLen = p[ 1 ];
		P.IntFltAlpha = p[ 2 ]; P.IntFltCutoff = p[ 3 ];
		P.IntFltLen = p[ 4 ]; P.LPFltAlpha = p[ 5 ];
		P.LPFltBaseLen = p[ 6 ]; P.LPFltCutoffMult = p[ 7 ];
	}
}

/* CImageResizerVars::UseSRGBGamma / AlphaIndex of the following
 * ref_avir_resize calls (set by ref_set_gamma; test harness only). */
int g_use_gamma = 0;
int g_alpha_index = -1;

template< class R, typename Tin, typename Tout >
void doResizeT( const void* src, int sw, int sh, int sstride, void* dst,
	int nw, int nh, int ch, double k, int resbits, int srcbits,
	const double* params, double ox, double oy, int build_mode, int threads )
{
	avir::CImageResizerParams P;
	fillParams( P, params );
	R r( resbits, srcbits, P );
	avir::CImageResizerVars V;
	V.ox = ox; V.oy = oy; V.BuildMode = build_mode;
	V.UseSRGBGamma = ( g_use_gamma != 0 );
	V.AlphaIndex = g_alpha_index;
	// one pool per thread count, kept between calls (timing harness)
	static std::mutex poolm;
	static std::map< int, CRefThreadPool* > pools;
	std::unique_lock< std::mutex > plock( poolm );

	if( threads > 1 )
	{
		CRefThreadPool*& pp = pools[ threads ];

		if( pp == nullptr )
		{
			pp = new CRefThreadPool( threads );
		}

		V.ThreadPool = pp;
	}

	r.resizeImage( (const Tin*) src, sw, sh, sstride, (Tout*) dst, nw, nh,
		ch, k, &V );
}

template< class R >
int doResize( const void* src, int sw, int sh, int sstride, void* dst,
	int nw, int nh, int ch, double k, int in_type, int out_type, int resbits,
	int srcbits, const double* params, double ox, double oy, int build_mode,
	int threads )
{
#define CASE( ti, to, Tin, Tout ) \
	if( in_type == ti && out_type == to ) { \
		doResizeT< R, Tin, Tout >( src, sw, sh, sstride, dst, nw, nh, ch, k, \
			resbits, srcbits, params, ox, oy, build_mode, threads ); \
		return( 0 ); }
	CASE( AVIRHIP_U8, AVIRHIP_U8, uint8_t, uint8_t )
	CASE( AVIRHIP_U8, AVIRHIP_U16, uint8_t, uint16_t )
	CASE( AVIRHIP_U8, AVIRHIP_F32, uint8_t, float )
	CASE( AVIRHIP_U8, AVIRHIP_F64, uint8_t, double )
	CASE( AVIRHIP_U16, AVIRHIP_U8, uint16_t, uint8_t )
	CASE( AVIRHIP_U16, AVIRHIP_U16, uint16_t, uint16_t )
	CASE( AVIRHIP_U16, AVIRHIP_F32, uint16_t, float )
	CASE( AVIRHIP_U16, AVIRHIP_F64, uint16_t, double )
	CASE( AVIRHIP_F32, AVIRHIP_U8, float, uint8_t )
	CASE( AVIRHIP_F32, AVIRHIP_U16, float, uint16_t )
	CASE( AVIRHIP_F32, AVIRHIP_F32, float, float )
	CASE( AVIRHIP_F32, AVIRHIP_F64, float, double )
	CASE( AVIRHIP_F64, AVIRHIP_U8, double, uint8_t )
	CASE( AVIRHIP_F64, AVIRHIP_U16, double, uint16_t )
	CASE( AVIRHIP_F64, AVIRHIP_F32, double, float )
	CASE( AVIRHIP_F64, AVIRHIP_F64, double, double )
#undef CASE
	return( -1 );
}

typedef avir::CImageResizer< avir::fpclass_def< float > > RDef;
typedef avir::CImageResizer< avir::fpclass_float4 > RF4;
/* avir.h:4553-4560: the double pipeline */
typedef avir::CImageResizer< avir::fpclass_def< double > > RDbl;
/* README.md:135-142: the error-diffusion ditherer */
typedef avir::CImageResizer< avir::fpclass_def< float, float,
	avir::CImageResizerDithererErrdINL< float > > > RErrd;
#if defined( REF_SIMD ) && defined( __AVX__ )
typedef avir::CImageResizer< avir::fpclass_float8_dil > RF8;
#endif

/* ---- plan dump ---- */

struct DumpStore
{
	/* Owns every array referenced by a dumped description. */
	std::vector< std::vector< float > > f;
	std::vector< std::vector< int32_t > > i;
	std::vector< std::vector< avirhip_rpos > > r;
	std::vector< std::vector< avirhip_step > > s;
	std::vector< std::vector< avirhip_lancir_pos > > lp;
	avirhip_plan_desc d;
	avirhip_lancir_desc ld;
};

const float* keepF( DumpStore& S, const float* p, int n )
{
	S.f.emplace_back( p, p + n );
	return( S.f.back().data() );
}

void dumpAxis( DumpStore& S, RDef::CFilterSteps& Steps,
	const avir::CImageResizerVars& V, int mode, avirhip_axis& ax )
{
	const int n = Steps.getItemCount();
	S.s.emplace_back( (size_t) n );
	std::vector< avirhip_step >& out = S.s.back();
	int i;

	for( i = 0; i < n; i++ )
	{
		RDef::CFilterStep& fs = Steps[ i ];
		avirhip_step& o = out[ i ];
		memset( &o, 0, sizeof( o ));
		o.resample_factor = fs.ResampleFactor;
		o.in_len = fs.InLen; o.in_prefix = fs.InPrefix;
		o.in_suffix = fs.InSuffix; o.out_len = fs.OutLen;

		if( fs.ResampleFactor == 0 )
		{
			o.kind = ( V.IsResize2 ? AVIRHIP_STEP_RESIZE2 :
				AVIRHIP_STEP_RESIZE );

			const avir::CDSPFracFilterBankLin< float >* B = fs.FltBank;
			const int fl = B -> getFilterLen();
			const int fsz = fl * ( B -> getOrder() + 1 );
			o.bank_filter_len = fl;
			o.bank_order = B -> getOrder();
			o.bank_frac_count = B -> getFracCount();

			std::map< int, int > used;
			int j;
			for( j = 0; j < fs.OutLen; j++ )
			{
				used[ (*fs.RPosBuf)[ j ].fti ] = 0;
			}

			S.i.emplace_back();
			std::vector< int32_t >& pf = S.i.back();
			S.f.emplace_back();
			std::vector< float >& pt = S.f.back();
			int pi = 0;
			for( auto& u : used )
			{
				u.second = pi++;
				pf.push_back( u.first );
				const float* t = B -> getFilterConst( u.first );
				pt.insert( pt.end(), t, t + fsz );
			}

			o.n_phases = pi;
			o.phase_fti = pf.data();
			o.phase_taps = pt.data();

			S.r.emplace_back( (size_t) fs.OutLen );
			std::vector< avirhip_rpos >& rp = S.r.back();
			for( j = 0; j < fs.OutLen; j++ )
			{
				const RDef::CFilterStep::CResizePos& p = (*fs.RPosBuf)[ j ];
				rp[ j ].src_pos_int = p.SrcPosInt;
				rp[ j ].fti = p.fti;
				rp[ j ].phase = used[ p.fti ];
				rp[ j ].x = p.x;
				rp[ j ].src_offs_px = p.SrcOffs / V.ElCount;
				rp[ j ].fl = ( V.IsResize2 ? p.fl : fl );
				rp[ j ].ftp_off = (int) ( p.ftp -
					B -> getFilterConst( p.fti ));
			}
			o.rpos = rp.data();
		}
		else
		if( fs.IsUpsample )
		{
			o.out_prefix = fs.OutPrefix; o.out_suffix = fs.OutSuffix;

			if( fs.FltOrig.getCapacity() > 0 )
			{
				o.kind = AVIRHIP_STEP_UP_ZEROSTUFF;
			}
			else
			{
				o.kind = AVIRHIP_STEP_UP_FILTERED;
				o.flt_len = fs.Flt.getCapacity();
				o.flt_latency = fs.FltLatency;
				o.flt = keepF( S, &fs.Flt[ 0 ], o.flt_len );
				o.prefix_dc_len = fs.PrefixDC.getCapacity();
				o.prefix_dc = keepF( S, &fs.PrefixDC[ 0 ], o.prefix_dc_len );
				o.suffix_dc_len = fs.SuffixDC.getCapacity();
				o.suffix_dc = keepF( S, &fs.SuffixDC[ 0 ], o.suffix_dc_len );
			}
		}
		else
		{
			o.kind = AVIRHIP_STEP_FIR;
			o.edge_pixel_count = fs.EdgePixelCount;
			o.flt_len = fs.Flt.getCapacity();
			o.flt_latency = fs.FltLatency;
			o.flt = keepF( S, &fs.Flt[ 0 ], o.flt_len );
		}
	}

	ax.n_steps = n;
	ax.steps = out.data();
	ax.build_mode = mode;
}

} // namespace

extern "C" {

/* avir::CImageResizer<fpclass>::resizeImage<Tin,Tout>() on the CPU.
 * variant 0 = fpclass_def<float> (the parity oracle), 1 = fpclass_float4
 * (SSE), 2 = fpclass_float8_dil (AVX, de-interleaved; only in the REF_SIMD
 * speed build, which is a CPU-baseline timing aid and not a parity oracle),
 * 3 = fpclass_def<float,float,CImageResizerDithererErrdINL<float>>,
 * 4 = fpclass_def<double>. */
/* UseSRGBGamma / AlphaIndex for subsequent ref_avir_resize() calls. */
void ref_set_gamma( int use, int alpha_index )
{
	g_use_gamma = use;
	g_alpha_index = alpha_index;
}

/* The uint8_t specialisation of avir::convertSRGB2Lin (the 256-entry literal
 * table, avir.h:229-291), for checking the regenerated tables against it. */
void ref_srgb_u8_table( float* out )
{
	for( int i = 0; i < 256; i++ )
	{
		out[ i ] = avir::convertSRGB2Lin< float >( (unsigned char) i, 1.0f );
	}
}

/* The CImageResizerVarsBase fields resizeImage<uint8_t|float, same>() leaves
 * in *aVars (avir.h:4663-4669), for checking the product's write-back:
 * out[0..13] = ElCount, ElCountIO, fppack, fpalign, elalign, packmode,
 * BufLen[0..1], BufOffs[0..1], k, o, ResizeStep, IsResize2; out[14..15] =
 * InGammaMult, OutGammaMult (only meaningful with gamma). */
int ref_avir_vars_base( int sw, int sh, int nw, int nh, int ch, double k,
	int is_u8, int resbits, double ox, double oy, int build_mode, int gamma,
	double* out )
{
	RDef r( resbits );
	avir::CImageResizerVars V;
	V.ox = ox; V.oy = oy; V.BuildMode = build_mode;
	V.UseSRGBGamma = ( gamma != 0 );
	V.InGammaMult = -1.0; V.OutGammaMult = -1.0;

	if( is_u8 )
	{
		std::vector< uint8_t > a( (size_t) sw * sh * ch, 7 ),
			b( (size_t) nw * nh * ch );
		r.resizeImage( a.data(), sw, sh, 0, b.data(), nw, nh, ch, k, &V );
	}
	else
	{
		std::vector< float > a( (size_t) sw * sh * ch, 0.5f ),
			b( (size_t) nw * nh * ch );
		r.resizeImage( a.data(), sw, sh, 0, b.data(), nw, nh, ch, k, &V );
	}

	out[ 0 ] = V.ElCount; out[ 1 ] = V.ElCountIO; out[ 2 ] = V.fppack;
	out[ 3 ] = V.fpalign; out[ 4 ] = V.elalign; out[ 5 ] = V.packmode;
	out[ 6 ] = V.BufLen[ 0 ]; out[ 7 ] = V.BufLen[ 1 ];
	out[ 8 ] = V.BufOffs[ 0 ]; out[ 9 ] = V.BufOffs[ 1 ];
	out[ 10 ] = V.k; out[ 11 ] = V.o; out[ 12 ] = V.ResizeStep;
	out[ 13 ] = V.IsResize2; out[ 14 ] = V.InGammaMult;
	out[ 15 ] = V.OutGammaMult;
	return( 0 );
}

/* The same for avir::CImageResizer<fpclass_float4> (uint8 or float images). */
int ref_avir_vars_base_f4( int sw, int sh, int nw, int nh, int ch, double k,
	int is_u8, int resbits, double* out )
{
	RF4 r( resbits );
	avir::CImageResizerVars V;

	if( is_u8 )
	{
		std::vector< uint8_t > a( (size_t) sw * sh * ch, 7 ),
			b( (size_t) nw * nh * ch );
		r.resizeImage( a.data(), sw, sh, 0, b.data(), nw, nh, ch, k, &V );
	}
	else
	{
		std::vector< float > a( (size_t) sw * sh * ch, 0.5f ),
			b( (size_t) nw * nh * ch );
		r.resizeImage( a.data(), sw, sh, 0, b.data(), nw, nh, ch, k, &V );
	}

	out[ 0 ] = V.ElCount; out[ 1 ] = V.ElCountIO; out[ 2 ] = V.fppack;
	out[ 3 ] = V.fpalign; out[ 4 ] = V.elalign; out[ 5 ] = V.packmode;
	out[ 6 ] = V.BufLen[ 0 ]; out[ 7 ] = V.BufLen[ 1 ];
	out[ 8 ] = V.BufOffs[ 0 ]; out[ 9 ] = V.BufOffs[ 1 ];
	out[ 10 ] = V.k; out[ 11 ] = V.o; out[ 12 ] = V.ResizeStep;
	out[ 13 ] = V.IsResize2;
	return( 0 );
}

/* The same for avir::CImageResizer<fpclass_def<double>>, with an explicit
 * build mode (uint8 or double images). */
int ref_avir_vars_base_f64( int sw, int sh, int nw, int nh, int ch, double k,
	int is_u8, int resbits, int build_mode, double* out )
{
	RDbl r( resbits );
	avir::CImageResizerVars V;
	V.BuildMode = build_mode;

	if( is_u8 )
	{
		std::vector< uint8_t > a( (size_t) sw * sh * ch, 7 ),
			b( (size_t) nw * nh * ch );
		r.resizeImage( a.data(), sw, sh, 0, b.data(), nw, nh, ch, k, &V );
	}
	else
	{
		std::vector< double > a( (size_t) sw * sh * ch, 0.5 ),
			b( (size_t) nw * nh * ch );
		r.resizeImage( a.data(), sw, sh, 0, b.data(), nw, nh, ch, k, &V );
	}

	out[ 0 ] = V.ElCount; out[ 1 ] = V.ElCountIO; out[ 2 ] = V.fppack;
	out[ 3 ] = V.fpalign; out[ 4 ] = V.elalign; out[ 5 ] = V.packmode;
	out[ 6 ] = V.BufLen[ 0 ]; out[ 7 ] = V.BufLen[ 1 ];
	out[ 8 ] = V.BufOffs[ 0 ]; out[ 9 ] = V.BufOffs[ 1 ];
	out[ 10 ] = V.k; out[ 11 ] = V.o; out[ 12 ] = V.ResizeStep;
	out[ 13 ] = V.IsResize2;
	return( 0 );
}

int ref_avir_resize( const void* src, int sw, int sh, int sstride, void* dst,
	int nw, int nh, int ch, double k, int in_type, int out_type, int resbits,
	int srcbits, const double* params, double ox, double oy, int build_mode,
	int threads, int variant )
{
#if defined( REF_SIMD ) && defined( __AVX__ )
	if( variant == 2 )
	{
		return( doResize< RF8 >( src, sw, sh, sstride, dst, nw, nh, ch, k,
			in_type, out_type, resbits, srcbits, params, ox, oy, build_mode,
			threads ));
	}
#endif

	if( variant == 3 )
	{
		return( doResize< RErrd >( src, sw, sh, sstride, dst, nw, nh, ch, k,
			in_type, out_type, resbits, srcbits, params, ox, oy, build_mode,
			threads ));
	}

	if( variant == 4 )
	{
		return( doResize< RDbl >( src, sw, sh, sstride, dst, nw, nh, ch, k,
			in_type, out_type, resbits, srcbits, params, ox, oy, build_mode,
			threads ));
	}

	if( variant == 1 )
	{
		return( doResize< RF4 >( src, sw, sh, sstride, dst, nw, nh, ch, k,
			in_type, out_type, resbits, srcbits, params, ox, oy, build_mode,
			threads ));
	}

	return( doResize< RDef >( src, sw, sh, sstride, dst, nw, nh, ch, k,
		in_type, out_type, resbits, srcbits, params, ox, oy, build_mode,
		threads ));
}

/* avir::CLancIR::resizeImage<Tin,Tout>() on the CPU (x86-64 SSE2/AVX
 * summation order -- the canonical LANCIR oracle). */
int ref_lancir_resize( const void* src, int sw, int sh, void* dst, int nw,
	int nh, int ch, int in_type, int out_type, int src_ssize, int new_ssize,
	double kx, double ky, double ox, double oy, double la )
{
	avir::CLancIR L;
	avir::CLancIRParams P( src_ssize, new_ssize, kx, ky, ox, oy );
	P.la = la;
#define CASE( ti, to, Tin, Tout ) \
	if( in_type == ti && out_type == to ) \
		return( L.resizeImage( (const Tin*) src, sw, sh, (Tout*) dst, nw, nh, \
			ch, &P ));
	CASE( AVIRHIP_U8, AVIRHIP_U8, uint8_t, uint8_t )
	CASE( AVIRHIP_U8, AVIRHIP_U16, uint8_t, uint16_t )
	CASE( AVIRHIP_U8, AVIRHIP_F32, uint8_t, float )
	CASE( AVIRHIP_U16, AVIRHIP_U8, uint16_t, uint8_t )
	CASE( AVIRHIP_U16, AVIRHIP_U16, uint16_t, uint16_t )
	CASE( AVIRHIP_U16, AVIRHIP_F32, uint16_t, float )
	CASE( AVIRHIP_F32, AVIRHIP_U8, float, uint8_t )
	CASE( AVIRHIP_F32, AVIRHIP_U16, float, uint16_t )
	CASE( AVIRHIP_F32, AVIRHIP_F32, float, float )
	/* double, and uint32_t "treated as uint16_t" (lancir.h:373-381) */
	CASE( AVIRHIP_F64, AVIRHIP_F64, double, double )
	CASE( AVIRHIP_F64, AVIRHIP_F32, double, float )
	CASE( AVIRHIP_F64, AVIRHIP_U8, double, uint8_t )
	CASE( AVIRHIP_F64, AVIRHIP_U16, double, uint16_t )
	CASE( AVIRHIP_F64, AVIRHIP_U32, double, uint32_t )
	CASE( AVIRHIP_F32, AVIRHIP_F64, float, double )
	CASE( AVIRHIP_U8, AVIRHIP_F64, uint8_t, double )
	CASE( AVIRHIP_U16, AVIRHIP_F64, uint16_t, double )
	CASE( AVIRHIP_U32, AVIRHIP_F64, uint32_t, double )
	CASE( AVIRHIP_U32, AVIRHIP_U32, uint32_t, uint32_t )
	CASE( AVIRHIP_U32, AVIRHIP_U16, uint32_t, uint16_t )
	CASE( AVIRHIP_U32, AVIRHIP_U8, uint32_t, uint8_t )
	CASE( AVIRHIP_U32, AVIRHIP_F32, uint32_t, float )
	CASE( AVIRHIP_U8, AVIRHIP_U32, uint8_t, uint32_t )
	CASE( AVIRHIP_U16, AVIRHIP_U32, uint16_t, uint32_t )
	CASE( AVIRHIP_F32, AVIRHIP_U32, float, uint32_t )
#undef CASE
	return( -1 );
}

/* Dumps the plan avir::CImageResizer<fpclass_def<float>>::resizeImage()
 * builds for this call, by driving the reference's own (private) planner
 * functions in the order resizeImage() does (avir.h:4699-4954). */
avirhip_plan_desc* ref_avir_plan_dump( int sw, int sh, int sstride, int nw,
	int nh, int ch, double k, int in_type, int out_type, int resbits,
	int srcbits, const double* params, double ox0, double oy0,
	int build_mode )
{
	avir::CImageResizerParams P;
	fillParams( P, params );
	RDef r( resbits, srcbits, P );
	avir::CImageResizerVars Vars;
	Vars.ox = ox0; Vars.oy = oy0; Vars.BuildMode = build_mode;

	double kx, ky, ox = Vars.ox, oy = Vars.oy;
	if( k == 0.0 )
	{
		kx = (double) sw / nw; ox += ( kx - 1.0 ) * 0.5;
		ky = (double) sh / nh; oy += ( ky - 1.0 ) * 0.5;
	}
	else
	if( k > 0.0 )
	{
		kx = k; ky = k;
		const double ko = ( k - 1.0 ) * 0.5;
		ox += ko; oy += ko;
	}
	else
	{
		kx = -k; ky = -k;
	}

	const bool IsInFloat = ( in_type >= AVIRHIP_F32 );
	const bool IsOutFloat = ( out_type >= AVIRHIP_F32 );
	double OutMul;
	if( IsOutFloat ) OutMul = 1.0;
	else OutMul = ( out_type == AVIRHIP_U8 ? 255.0 : 65535.0 );
	if( !IsInFloat ) OutMul /= ( in_type == AVIRHIP_U8 ? 255.0 : 65535.0 );

	const int ElCount = ch;
	if( sstride < 1 ) sstride = sw * ch;
	Vars.ElCount = ElCount; Vars.ElCountIO = ch;
	Vars.fppack = 1; Vars.fpalign = 4; Vars.elalign = 1; Vars.packmode = 0;

	avir::CDSPFracFilterBankLin< float > FltBank;
	RDef::CFilterSteps FltSteps;
	RDef::CFilterStep::CRPosBufArray RPosBufArray;
	avir::CBuffer< char > UsedFracMap;

	int UseBuildMode = 1;
	const int BuildModeCount = ( r.FixedFilterBank.getOrder() == 0 ? 4 : 2 );
	int m;

	if( Vars.BuildMode >= 0 )
	{
		UseBuildMode = Vars.BuildMode;
	}
	else
	{
		int BestScore = 0x7FFFFFFF;
		for( m = 0; m < BuildModeCount; m++ )
		{
			avir::CDSPFracFilterBankLin< float > TmpBank;
			RDef::CFilterSteps TmpSteps;
			Vars.k = kx; Vars.o = ox;
			r.buildFilterSteps( TmpSteps, Vars, TmpBank, OutMul, m, true );
			RDef::updateFilterStepBuffers( TmpSteps, Vars, RPosBufArray,
				sw, nw );
			RDef::fillUsedFracMap( TmpSteps[ Vars.ResizeStep ], UsedFracMap );
			const int c = RDef::calcComplexity( TmpSteps, Vars, UsedFracMap,
				sh );
			if( c < BestScore ) { UseBuildMode = m; BestScore = c; }
		}
	}

	Vars.k = kx; Vars.o = ox;
	r.buildFilterSteps( FltSteps, Vars, FltBank, OutMul, UseBuildMode, false );
	RDef::updateFilterStepBuffers( FltSteps, Vars, RPosBufArray, sw, nw );
	RDef::updateBufLenAndRPosPtrs( FltSteps, Vars, nw );

	DumpStore* S = new DumpStore();
	memset( &S -> d, 0, sizeof( S -> d ));
	dumpAxis( *S, FltSteps, Vars, UseBuildMode, S -> d.h );

	const int PrevUseBuildMode = UseBuildMode;

	if( Vars.BuildMode >= 0 )
	{
		UseBuildMode = Vars.BuildMode;
	}
	else
	{
		avir::CImageResizerVars TmpVars( Vars );
		int BestScore = 0x7FFFFFFF;
		for( m = 0; m < BuildModeCount; m++ )
		{
			avir::CDSPFracFilterBankLin< float > TmpBank;
			TmpBank.copyInitParams( FltBank );
			RDef::CFilterSteps TmpSteps;
			TmpVars.k = ky; TmpVars.o = oy;
			r.buildFilterSteps( TmpSteps, TmpVars, TmpBank, 1.0, m, true );
			RDef::updateFilterStepBuffers( TmpSteps, TmpVars, RPosBufArray,
				sh, nh );
			RDef::fillUsedFracMap( TmpSteps[ TmpVars.ResizeStep ],
				UsedFracMap );
			const int c = RDef::calcComplexity( TmpSteps, TmpVars,
				UsedFracMap, nw );
			if( c < BestScore ) { UseBuildMode = m; BestScore = c; }
		}
	}

	Vars.k = ky; Vars.o = oy;

	if( UseBuildMode == PrevUseBuildMode && ky == kx )
	{
		if( OutMul != 1.0 ) r.modifyCorrFilterDCGain( FltSteps, 1.0 / OutMul );
	}
	else
	{
		r.buildFilterSteps( FltSteps, Vars, FltBank, 1.0, UseBuildMode,
			false );
	}

	RDef::updateFilterStepBuffers( FltSteps, Vars, RPosBufArray, sh, nh );
	RDef::updateBufLenAndRPosPtrs( FltSteps, Vars, nw );
	dumpAxis( *S, FltSteps, Vars, UseBuildMode, S -> d.v );

	avirhip_plan_desc& d = S -> d;
	d.src_w = sw; d.src_h = sh; d.src_stride_elems = sstride;
	d.new_w = nw; d.new_h = nh; d.channels = ch;
	d.in_type = in_type; d.out_type = out_type;
	d.tr_mul = 1.0; d.pk_out = 0.0;

	if( !IsOutFloat )
	{
		int TruncBits, OutRange;
		if( out_type == AVIRHIP_U8 ) { TruncBits = 8 - resbits; OutRange = 255; }
		else { TruncBits = 16 - resbits; OutRange = 65535; }
		d.pk_out = OutRange;
		d.tr_mul = ( TruncBits > 0 ?
			d.pk_out / ( OutRange >> TruncBits ) : 1.0 );
	}

	return( &S -> d );
}

void ref_avir_plan_free( avirhip_plan_desc* d )
{
	if( d == nullptr ) return;
	DumpStore* S = (DumpStore*) ( (char*) d - offsetof( DumpStore, d ));
	delete S;
}

/* Dumps CLancIR's filters / positions for this call (lancir.h:430-483). */
avirhip_lancir_desc* ref_lancir_plan_dump( int sw, int sh, int nw, int nh,
	int ch, int in_type, int out_type, int src_ssize, int new_ssize,
	double pkx, double pky, double pox, double poy, double la )
{
	if( sw <= 0 || sh <= 0 || nw <= 0 || nh <= 0 || la < 2.0 ) return( nullptr );

	avir::CLancIR L;
	double ox = pox, oy = poy, kx, ky;
	if( pkx >= 0.0 )
	{
		kx = ( pkx == 0.0 ? (double) sw / nw : pkx );
		ox += ( kx - 1.0 ) * 0.5;
	}
	else kx = -pkx;
	if( pky >= 0.0 )
	{
		ky = ( pky == 0.0 ? (double) sh / nh : pky );
		oy += ( ky - 1.0 ) * 0.5;
	}
	else ky = -pky;

	L.rfv.update( la, ky, ch );
	avir::CLancIR::CResizeFilters* rfh = &L.rfv;
	if( kx != ky ) { rfh = &L.rfh0; L.rfh0.update( la, kx, ch ); }
	L.rsv.update( sh, nh, oy, L.rfv, nullptr );
	L.rsh.update( sw, nw, ox, *rfh );

	DumpStore* S = new DumpStore();
	avirhip_lancir_desc& d = S -> ld;
	memset( &d, 0, sizeof( d ));
	d.src_w = sw; d.src_h = sh;
	d.src_stride_elems = ( src_ssize < 1 ? sw * ch : src_ssize );
	d.new_w = nw; d.new_h = nh;
	d.new_stride_elems = ( new_ssize < 1 ? nw * ch : new_ssize );
	d.channels = ch; d.in_type = in_type; d.out_type = out_type;

	const bool IsInFloat = ( in_type >= AVIRHIP_F32 );
	const bool IsOutFloat = ( out_type >= AVIRHIP_F32 );
	const size_t szin = ( in_type == AVIRHIP_U8 ? 1 : in_type == AVIRHIP_U16 ? 2 : 4 );
	const size_t szout = ( out_type == AVIRHIP_U8 ? 1 : out_type == AVIRHIP_U16 ? 2 : 4 );
	d.is_unity_mul = (( IsInFloat && IsOutFloat ) ||
		( IsInFloat == IsOutFloat && szin == szout )) ? 1 : 0;
	d.clamp = ( szout == 1 ? 255.0f : 65535.0f );
	d.out_mul = ( IsOutFloat ? 1.0f : d.clamp ) /
		( IsInFloat ? 1.0f : ( szin == 1 ? 255.0f : 65535.0f ));

	for( int a = 0; a < 2; a++ )
	{
		avir::CLancIR::CResizeFilters& rf = ( a == 0 ? L.rfv : *rfh );
		avir::CLancIR::CResizeScanline& rs = ( a == 0 ? L.rsv : L.rsh );
		avirhip_lancir_axis& ax = ( a == 0 ? d.v : d.h );
		ax.kernel_len = rf.KernelLen;
		ax.padl = rs.padl; ax.padr = rs.padr;
		ax.src_len = ( a == 0 ? sh : sw );
		ax.dst_len = ( a == 0 ? nh : nw );

		std::map< int, int > used; /* frac -> compact index */
		S -> i.emplace_back();
		std::vector< int32_t >& ff = S -> i.back();
		S -> f.emplace_back();
		std::vector< float >& ft = S -> f.back();
		S -> lp.emplace_back( (size_t) ax.dst_len );
		std::vector< avirhip_lancir_pos >& pp = S -> lp.back();

		/* Map filter pointers back to phase indices. */
		std::map< const float*, int > ptr2frac;
		for( int fr = 0; fr <= rf.FracCount; fr++ )
		{
			if( rf.Filters[ fr ] != nullptr ) ptr2frac[ rf.Filters[ fr ]] = fr;
		}

		for( int j = 0; j < ax.dst_len; j++ )
		{
			used[ ptr2frac[ rs.pos[ j ].flt ]] = 0;
		}

		int ci = 0;
		for( auto& u : used )
		{
			u.second = ci++;
			ff.push_back( u.first );
			const float* t = rf.Filters[ u.first ];
			for( int q = 0; q < rf.KernelLen; q++ )
			{
				ft.push_back( t[ q * rf.ElRepl ]);
			}
		}

		for( int j = 0; j < ax.dst_len; j++ )
		{
			pp[ j ].flt_index = used[ ptr2frac[ rs.pos[ j ].flt ]];
			pp[ j ].so = (int32_t) rs.pos[ j ].so;
		}

		ax.n_filters = ci;
		ax.filter_frac = ff.data();
		ax.filters = ft.data();
		ax.pos = pp.data();
	}

	return( &d );
}

void ref_lancir_plan_free( avirhip_lancir_desc* d )
{
	if( d == nullptr ) return;
	DumpStore* S = (DumpStore*) ( (char*) d - offsetof( DumpStore, ld ));
	delete S;
}

/* Which SIMD path the LANCIR oracle was built with (the x86 SSE2 and AVX
 * builds give identical results). */
const char* ref_build_info( void )
{
#if defined( LANCIR_AVX )
	return( "avir v" AVIR_VERSION " lancir=AVX" );
#elif defined( LANCIR_SSE2 )
	return( "avir v" AVIR_VERSION " lancir=SSE2" );
#else
	return( "avir v" AVIR_VERSION " lancir=scalar" );
#endif
}

} // extern "C"
