// fused.hip -- fused single-pass kernels (placeholder until the generic path is parity-green).
#include "plan.h"
namespace avirhip {
int fused_prepare( avirhip_plan* p ) { p -> fused_ok = 0; return( AVIRHIP_OK ); }
void fused_release( avirhip_plan* p ) { }
int fused_run( avirhip_plan* p, const float* src, float* dst, int row0, int row1, hipStream_t st )
{ set_error( "fused path not available" ); return( AVIRHIP_EUNSUPPORTED ); }
}
