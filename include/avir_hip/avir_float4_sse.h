/*
 * avir_hip/avir_float4_sse.h -- drop-in for the reference's avir_float4_sse.h
 * (/root/reference/avir_float4_sse.h:37-331): `avir::float4` and
 * `avir::fpclass_float4`, so that the README's SIMD snippet
 * (README.md:174-186)
 *
 *     #include "avir_float4_sse.h"
 *     avir :: CImageResizer< avir :: fpclass_float4 > ImageResizer( 8 );
 *
 * compiles against include/avir_hip/ and returns what the reference's SSE
 * class returns. On the MI355X the value type is only a tag -- the kernels
 * compute every channel in float, as the reference's SSE lanes do -- but the
 * class changes three things the reference derives from `fppack == 4`, and
 * the library reproduces them (avirhip_resizer_set_fpclass, avirhip.h):
 *   - ElCount = ceil(ElCountIO / 4) in the build-mode cost model, so the
 *     automatic BuildMode can differ from the scalar class (avir.h:4786,
 *     6206-6270);
 *   - the ditherer rounds to nearest-even (avir_float4_sse.h:303-313) where
 *     avir::round adds 0.5 and truncates (avir.h:130-135);
 *   - float output is not the vertical pass' in-place result (avir.h:4956),
 *     so with sRGB gamma it is de-linearised like every other output type.
 */
#ifndef AVIR_HIP_AVIR_FLOAT4_SSE_H
#define AVIR_HIP_AVIR_FLOAT4_SSE_H

#include "avir.h"

namespace avir {

/* avir_float4_sse.h:37-295: four packed floats (one RGBA pixel per value).
 * Host code only ever needs its size and alignment here. */
class float4
{
public:
	float4() {}
	float4( const float v ) { value[ 0 ] = value[ 1 ] = value[ 2 ] =
		value[ 3 ] = v; }

	alignas( 16 ) float value[ 4 ];
};

/* avir_float4_sse.h:331. */
typedef fpclass_def< avir :: float4, float > fpclass_float4;

} // namespace avir

#endif // AVIR_HIP_AVIR_FLOAT4_SSE_H
