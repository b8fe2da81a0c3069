/*
 * avir_hip/avir.h -- drop-in front end: the public API of the reference's
 * avir.h (avaneev/avir v3.1) re-implemented as a thin shell over the C ABI of
 * libavirhip (include/avirhip.h). A program written against the reference
 * compiles against this header unchanged:
 *
 *     #include "avir.h"          // -I<repo>/include/avir_hip, link -lavirhip
 *     avir :: CImageResizer<> ImageResizer( 8 );
 *     ImageResizer.resizeImage( InBuf, 640, 480, 0, OutBuf, 1024, 768, 3, 0 );
 *
 * Same names, argument order and meaning as the reference (file:line cited at
 * each declaration): avir::CImageResizer<fpclass>(aResBitDepth, aSrcBitDepth,
 * aParams) and resizeImage<Tin,Tout>(SrcBuf, SrcWidth, SrcHeight,
 * SrcScanlineSize, NewBuf, NewWidth, NewHeight, ElCountIO, k, aVars), the
 * CImageResizerParams* presets, CImageResizerVars and the (unused on a GPU)
 * CImageResizerThreadPool hook. The planner runs on the host inside the
 * library; H pass, V pass and the integer epilogue run as HIP kernels on
 * gfx950. SrcBuf/NewBuf may be host pointers (staged) or device pointers
 * (detected; zero-copy).
 *
 * Differences by design: there is NO CPU execution path -- if the library
 * cannot run the call (no gfx950 device, or a request outside the built
 * scope) resizeImage() throws std::runtime_error instead of silently computing
 * elsewhere. UseSRGBGamma / AlphaIndex are honoured (bit-exact gamma stages).
 * Pipelines provided: fpclass_def<float> (the reference's default, every fast
 * kernel), fpclass_def<double> (avir.h:4553-4560: tables and arithmetic in
 * double, bit-identical to the reference's double class) and, through
 * avir_float4_sse.h, fpclass_float4; each with the default or the
 * error-diffusion ditherer as the third template parameter.
 */
#ifndef AVIR_HIP_AVIR_H
#define AVIR_HIP_AVIR_H

#include <stdint.h>
#include <new>
#include <stdexcept>
#include <string>
#include "../avirhip.h"

#define AVIR_VERSION "3.1" /* API level mirrored (reference avir.h:99) */

namespace avir {

/* avir.h:2119-2225. Kept so that callers which install a pool still compile;
 * the GPU executor parallelises across workgroups instead. */
class CImageResizerThreadPool
{
public:
	CImageResizerThreadPool() { }
	virtual ~CImageResizerThreadPool() { }

	class CWorkload
	{
	public:
		virtual ~CWorkload() { }
		virtual void process() = 0;
	};

	virtual int getSuggestedWorkloadCount() const { return( 1 ); }
	virtual void addWorkload( CWorkload* const ) { }
	virtual void startAllWorkloads() { }
	virtual void waitAllWorkloadsToFinish() { }
	virtual void removeAllWorkloads() { }
};

/* avir.h:2262-2317. */
struct CImageResizerParams
{
	double CorrFltAlpha;
	double CorrFltLen;
	double IntFltAlpha;
	double IntFltCutoff;
	double IntFltLen;
	double LPFltAlpha;
	double LPFltBaseLen;
	double LPFltCutoffMult;

	CImageResizerParams()
		: HBFltAlpha( 1.94609 )
		, HBFltCutoff( 0.46437 )
		, HBFltLen( 24 )
	{
	}

	double HBFltAlpha;
	double HBFltCutoff;
	double HBFltLen;

protected:
	void preset( const int i )
	{
		avirhip_params p;
		avirhip_params_preset( i, &p );
		CorrFltAlpha = p.CorrFltAlpha; CorrFltLen = p.CorrFltLen;
		IntFltAlpha = p.IntFltAlpha; IntFltCutoff = p.IntFltCutoff;
		IntFltLen = p.IntFltLen; LPFltAlpha = p.LPFltAlpha;
		LPFltBaseLen = p.LPFltBaseLen; LPFltCutoffMult = p.LPFltCutoffMult;
	}
};

/* Presets, avir.h:2328-2464. */
struct CImageResizerParamsDef : public CImageResizerParams
	{ CImageResizerParamsDef() { preset( AVIRHIP_PARAMS_DEF ); } };
struct CImageResizerParamsULR : public CImageResizerParams
	{ CImageResizerParamsULR() { preset( AVIRHIP_PARAMS_ULR ); } };
struct CImageResizerParamsLR : public CImageResizerParams
	{ CImageResizerParamsLR() { preset( AVIRHIP_PARAMS_LR ); } };
struct CImageResizerParamsLow : public CImageResizerParams
	{ CImageResizerParamsLow() { preset( AVIRHIP_PARAMS_LOW ); } };
struct CImageResizerParamsHigh : public CImageResizerParams
	{ CImageResizerParamsHigh() { preset( AVIRHIP_PARAMS_HIGH ); } };
struct CImageResizerParamsUltra : public CImageResizerParams
	{ CImageResizerParamsUltra() { preset( AVIRHIP_PARAMS_ULTRA ); } };

/* avir.h:2473-2506: overwritten by resizeImage(), as in the reference. */
class CImageResizerVarsBase
{
public:
	int ElCount;
	int ElCountIO;
	int fppack;
	int fpalign;
	int elalign;
	int packmode;
	int BufLen[ 2 ];
	int BufOffs[ 2 ];
	double k;
	double o;
	int ResizeStep;
	bool IsResize2;
	double InGammaMult;
	double OutGammaMult;
};

/* avir.h:2516-2547. */
class CImageResizerVars : public CImageResizerVarsBase
{
public:
	double ox;
	double oy;
	CImageResizerThreadPool* ThreadPool;
	bool UseSRGBGamma;
	int AlphaIndex;
	int BuildMode;
	int RndSeed;

	CImageResizerVars()
		: ox( 0.0 )
		, oy( 0.0 )
		, ThreadPool( nullptr )
		, UseSRGBGamma( false )
		, AlphaIndex( -1 )
		, BuildMode( -1 )
		, RndSeed( 0 )
	{
	}
};

/* avir.h:4351-4427: the default ditherer (round + clamp, no dithering). Here
 * a tag type: the rounding itself runs on the device (AVIRHIP_DITHER_DEF). */
template< typename fptype >
class CImageResizerDithererDefINL
{
public:
	static bool isRecursive() { return( false ); }
	static const int avirhip_dither = AVIRHIP_DITHER_DEF;
};

/* avir.h:4441-4537: error-diffusion dithering (AVIRHIP_DITHER_ERRD). */
template< typename fptype >
class CImageResizerDithererErrdINL :
	public CImageResizerDithererDefINL< fptype >
{
public:
	static bool isRecursive() { return( true ); }
	static const int avirhip_dither = AVIRHIP_DITHER_ERRD;
};

/* avir.h:4569-4592. The device pipeline is the interleaved float one, i.e.
 * fpclass_def< float >; `adith` selects the ditherer of integer outputs, as in
 * the reference (README.md:135-142). */
template< typename afptype, typename afptypeatom = afptype,
	class adith = CImageResizerDithererDefINL< afptype > >
class fpclass_def
{
public:
	typedef afptype fptype;
	typedef afptypeatom fptypeatom;
	static const int fppack = sizeof( fptype ) / sizeof( fptypeatom );
	static const int fpalign = sizeof( fptype );
	static const int elalign = 1;
	static const int packmode = 0;
	typedef adith CDitherer;
};

namespace hipdetail {

template< typename T > struct dtype_of;
template<> struct dtype_of< uint8_t > { static const int v = AVIRHIP_U8; };
template<> struct dtype_of< uint16_t > { static const int v = AVIRHIP_U16; };
template<> struct dtype_of< float > { static const int v = AVIRHIP_F32; };
template<> struct dtype_of< double > { static const int v = AVIRHIP_F64; };

/* Memory exhaustion comes back from the library as a code (no exception
 * crosses its C boundary) and leaves here as what the reference itself throws,
 * std::bad_alloc (avir.h:564-827 CBuffer :: alloc). */
inline void check( const int rc, const char* const what )
{
	if( rc == AVIRHIP_ENOMEM )
	{
		throw std::bad_alloc();
	}

	if( rc < 0 )
	{
		throw std::runtime_error( std::string( what ) + ": " +
			avirhip_last_error() );
	}
}

} // namespace hipdetail

/* avir.h:4609-5092. */
template< class fpclass = fpclass_def< float > >
class CImageResizer
{
	CImageResizer( const CImageResizer& );
	CImageResizer& operator = ( const CImageResizer& );

public:
	/* avir.h:4630-4639. */
	CImageResizer( const int aResBitDepth = 8, const int aSrcBitDepth = 0,
		const CImageResizerParams& aParams = CImageResizerParamsDef() )
		: h( nullptr )
	{
		static_assert((( sizeof( typename fpclass :: fptypeatom ) ==
			sizeof( float ) && ( fpclass :: fppack == 1 ||
			fpclass :: fppack == 4 )) ||
			( sizeof( typename fpclass :: fptypeatom ) == sizeof( double ) &&
			fpclass :: fppack == 1 )) && fpclass :: packmode == 0,
			"avir_hip: fpclass_def<float>, fpclass_float4 "
			"(avir_float4_sse.h) or fpclass_def<double>" );

		avirhip_params p;
		p.CorrFltAlpha = aParams.CorrFltAlpha; p.CorrFltLen = aParams.CorrFltLen;
		p.IntFltAlpha = aParams.IntFltAlpha;
		p.IntFltCutoff = aParams.IntFltCutoff;
		p.IntFltLen = aParams.IntFltLen; p.LPFltAlpha = aParams.LPFltAlpha;
		p.LPFltBaseLen = aParams.LPFltBaseLen;
		p.LPFltCutoffMult = aParams.LPFltCutoffMult;
		p.HBFltAlpha = aParams.HBFltAlpha; p.HBFltCutoff = aParams.HBFltCutoff;
		p.HBFltLen = aParams.HBFltLen;

		hipdetail :: check( avirhip_resizer_create( aResBitDepth,
			aSrcBitDepth, &p, &h ), "avirhip_resizer_create" );

		hipdetail :: check( avirhip_resizer_set_ditherer( h,
			fpclass :: CDitherer :: avirhip_dither ),
			"avirhip_resizer_set_ditherer" );

		if( sizeof( typename fpclass :: fptypeatom ) == sizeof( double ))
		{
			hipdetail :: check( avirhip_resizer_set_fpclass( h,
				AVIRHIP_FPCLASS_DOUBLE ), "avirhip_resizer_set_fpclass" );
		}
		else
		if( fpclass :: fppack != 1 )
		{
			hipdetail :: check( avirhip_resizer_set_fpclass( h,
				fpclass :: fppack ), "avirhip_resizer_set_fpclass" );
		}
	}

	~CImageResizer()
	{
		avirhip_resizer_destroy( h );
	}

	/* avir.h:4680-4684. Thread-safe for concurrent calls with distinct
	 * aVars, like the reference (plans are cached under a mutex). */
	template< typename Tin, typename Tout >
	void resizeImage( const Tin* const SrcBuf, const int SrcWidth,
		const int SrcHeight, int SrcScanlineSize, Tout* const NewBuf,
		const int NewWidth, const int NewHeight, const int ElCountIO,
		const double k, CImageResizerVars* const aVars = nullptr ) const
	{
		avirhip_vars v;
		avirhip_vars_default( &v );

		if( aVars != nullptr )
		{
			v.ox = aVars -> ox; v.oy = aVars -> oy;
			v.UseSRGBGamma = ( aVars -> UseSRGBGamma ? 1 : 0 );
			v.AlphaIndex = aVars -> AlphaIndex;
			v.BuildMode = aVars -> BuildMode;
			v.RndSeed = aVars -> RndSeed;
			aVars -> ElCount = ( ElCountIO + fpclass :: fppack - 1 ) /
				fpclass :: fppack;
			aVars -> ElCountIO = ElCountIO;
			aVars -> fppack = fpclass :: fppack;
			aVars -> fpalign = fpclass :: fpalign;
			aVars -> elalign = fpclass :: elalign;
			aVars -> packmode = fpclass :: packmode;
		}

		hipdetail :: check( avirhip_resizer_resize( h, SrcBuf,
			AVIRHIP_MEM_AUTO, SrcWidth, SrcHeight, SrcScanlineSize, NewBuf,
			AVIRHIP_MEM_AUTO, NewWidth, NewHeight, ElCountIO, k, &v,
			hipdetail :: dtype_of< Tin > :: v,
			hipdetail :: dtype_of< Tout > :: v, nullptr ),
			"avir::CImageResizer::resizeImage" );

		if( aVars != nullptr && SrcWidth > 0 && SrcHeight > 0 &&
			NewWidth > 0 && NewHeight > 0 )
		{
			// avir.h:4663-4669: the CImageResizerVarsBase part of *aVars is
			// overwritten by the call
			avirhip_vars_base b;

			hipdetail :: check( avirhip_resizer_vars_base( h, SrcWidth,
				SrcHeight, SrcScanlineSize, NewWidth, NewHeight, ElCountIO,
				k, &v, hipdetail :: dtype_of< Tin > :: v,
				hipdetail :: dtype_of< Tout > :: v, &b ),
				"avirhip_resizer_vars_base" );

			aVars -> BufLen[ 0 ] = b.BufLen[ 0 ];
			aVars -> BufLen[ 1 ] = b.BufLen[ 1 ];
			aVars -> BufOffs[ 0 ] = b.BufOffs[ 0 ];
			aVars -> BufOffs[ 1 ] = b.BufOffs[ 1 ];
			aVars -> k = b.k;
			aVars -> o = b.o;
			aVars -> ResizeStep = b.ResizeStep;
			aVars -> IsResize2 = ( b.IsResize2 != 0 );

			if( b.gamma_valid )
			{
				aVars -> InGammaMult = b.InGammaMult;
				aVars -> OutGammaMult = b.OutGammaMult;
			}
		}
	}

private:
	avirhip_resizer* h;
};

} // namespace avir

#endif // AVIR_HIP_AVIR_H
