// f64_dev.h -- what the executors of the double pipeline share (tile64.hip: the
// LDS-tiled interpreter; up64.hip: the marching kernels of upsizing chains):
// the output stage of fpclass_def< double > plans.
#ifndef AVIRHIP_F64_DEV_H
#define AVIRHIP_F64_DEV_H

#include <hip/hip_runtime.h>
#include <type_traits>

namespace avirhip {

// avir::round< double > (avir.h:130-135) with the x86-64 build's cvttsd2si
// (plan.h: a value the int cannot hold converts to INT_MIN)
__device__ __forceinline__ double round64( const double d )
{
	const double a = ( d < 0.0 ? 0.5 - d : d + 0.5 );
	const double r = (double) ( fabs( a ) < 2147483648.0 ? (int) a :
		( -2147483647 - 1 ));

	return( d < 0.0 ? -r : r );
}

// unpackScanline's cast, or the default ditherer's round / clamp for integer
// pixels (avir.h:4392-4419, 3155-3215); double results pass as they are
template< typename Tout >
__device__ __forceinline__ Tout out_stage64( const int use_tr,
	const double tr_mul, const double tr_muli, const double pk_out, double v )
{
	if( std::is_same< Tout, double >::value )
	{
		return( (Tout) v );
	}

	if( std::is_integral< Tout >::value )
	{
		v = ( use_tr ? round64( v * tr_muli ) * tr_mul : round64( v ));
		v = ( v < 0.0 ? 0.0 : ( v > pk_out ? pk_out : v ));
	}

	return( (Tout) v );
}

} // namespace avirhip

#endif
