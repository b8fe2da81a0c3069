/*
 * avir_hip/lancir.h -- drop-in front end for avir::CLancIR (reference
 * lancir.h, avaneev/avir v3.1) over the C ABI of libavirhip. Same class,
 * method and parameter names as the reference:
 *
 *     avir :: CLancIR ImageResizer;
 *     ImageResizer.resizeImage( InBuf, 640, 480, OutBuf, 1024, 768, 4 );
 *
 * resizeImage() returns NewHeight, or 0 on the parameter errors the
 * reference rejects (lancir.h:392-407). Filters and positions are planned on
 * the host inside the library; the vertical and horizontal Lanczos passes run
 * as HIP kernels on gfx950 with the x86 SSE2/AVX summation order of the
 * reference (bit-identical results). Device or host pointers are accepted.
 * There is no CPU execution path: a request the device library cannot run
 * throws std::runtime_error. Memory exhaustion (AVIRHIP_ENOMEM from the
 * library, which itself lets no exception out) is rethrown here as
 * std::bad_alloc -- what the reference's own `new[]` would have thrown
 * (lancir.h:769-815 reallocBuf).
 */
#ifndef AVIR_HIP_LANCIR_H
#define AVIR_HIP_LANCIR_H

#include <stdint.h>
#include <new>
#include <stdexcept>
#include <string>
#include "../avirhip.h"

#define LANCIR_VERSION "3.1" /* API level mirrored */

namespace avir {

/* lancir.h:260-307. */
class CLancIRParams
{
public:
	int SrcSSize;
	int NewSSize;
	double kx;
	double ky;
	double ox;
	double oy;
	double la;

	CLancIRParams( const int aSrcSSize = 0, const int aNewSSize = 0,
		const double akx = 0.0, const double aky = 0.0,
		const double aox = 0.0, const double aoy = 0.0 )
		: SrcSSize( aSrcSSize )
		, NewSSize( aNewSSize )
		, kx( akx )
		, ky( aky )
		, ox( aox )
		, oy( aoy )
		, la( 3.0 )
	{
	}
};

/* lancir.h:327-755. Not thread-safe, like the reference (lancir.h:319-324). */
class CLancIR
{
	CLancIR( const CLancIR& );
	CLancIR& operator = ( const CLancIR& );

	template< typename T > struct dt;

public:
	CLancIR()
		: h( nullptr )
	{
		const int rc = avirhip_lancir_create( &h );

		if( rc == AVIRHIP_ENOMEM )
		{
			throw std::bad_alloc();
		}

		if( rc < 0 )
		{
			throw std::runtime_error( avirhip_last_error() );
		}
	}

	~CLancIR()
	{
		avirhip_lancir_destroy( h );
	}

	/* lancir.h:386-390. */
	template< typename Tin, typename Tout >
	int resizeImage( const Tin* const SrcBuf, const int SrcWidth,
		const int SrcHeight, Tout* const NewBuf, const int NewWidth,
		const int NewHeight, const int ElCount,
		const CLancIRParams* const aParams = nullptr )
	{
		avirhip_lancir_params p;
		avirhip_lancir_params_default( &p );

		if( aParams != nullptr )
		{
			p.SrcSSize = aParams -> SrcSSize; p.NewSSize = aParams -> NewSSize;
			p.kx = aParams -> kx; p.ky = aParams -> ky;
			p.ox = aParams -> ox; p.oy = aParams -> oy;
			p.la = aParams -> la;
		}

		const int rc = avirhip_lancir_resize( h, SrcBuf, AVIRHIP_MEM_AUTO,
			SrcWidth, SrcHeight, NewBuf, AVIRHIP_MEM_AUTO, NewWidth,
			NewHeight, ElCount, &p, dt< Tin > :: v, dt< Tout > :: v, nullptr );

		if( rc == AVIRHIP_ENOMEM )
		{
			throw std::bad_alloc();
		}

		if( rc < 0 )
		{
			throw std::runtime_error( std::string(
				"avir::CLancIR::resizeImage: " ) + avirhip_last_error() );
		}

		return( rc );
	}

	/* Legacy overload, lancir.h:744-755. */
	template< typename Tin, typename Tout >
	int resizeImage( const Tin* const SrcBuf, const int SrcWidth,
		const int SrcHeight, const int SrcSSize, Tout* const NewBuf,
		const int NewWidth, const int NewHeight, const int NewSSize,
		const int ElCount, const double kx0 = 0.0, const double ky0 = 0.0,
		double ox = 0.0, double oy = 0.0 )
	{
		const CLancIRParams Params( SrcSSize, NewSSize, kx0, ky0, ox, oy );

		return( resizeImage( SrcBuf, SrcWidth, SrcHeight, NewBuf, NewWidth,
			NewHeight, ElCount, &Params ));
	}

private:
	avirhip_lancir* h;
};

template<> struct CLancIR :: dt< uint8_t > { static const int v = AVIRHIP_U8; };
template<> struct CLancIR :: dt< uint16_t > { static const int v = AVIRHIP_U16; };
template<> struct CLancIR :: dt< float > { static const int v = AVIRHIP_F32; };
template<> struct CLancIR :: dt< double > { static const int v = AVIRHIP_F64; };
/* "`uint32_t` type is treated as `uint16_t`" (lancir.h:376-377): 32-bit
 * elements, 0..65535 value range. */
template<> struct CLancIR :: dt< uint32_t > { static const int v = AVIRHIP_U32; };

} // namespace avir

#endif // AVIR_HIP_LANCIR_H
