// gpassv_lanc.hip -- k_gv variants of LANCIR plans (even tap counts 6 .. 24
// in registers, anything else through the LDS window).
#include "gpassv_kernel.h"

namespace avirhip {

void launch_gv_lanc( const GVParams& P, int wp, int items, size_t lds,
	hipStream_t st )
{
	const int nt = P.ax.nt;

	switch( nt )
	{
		case 6: GV_LAUNCH( GP_NONE, true, 6, 8 ); break;
		case 8: GV_LAUNCH( GP_NONE, true, 8, 8 ); break;
		case 10: GV_LAUNCH( GP_NONE, true, 10, 16 ); break;
		case 12: GV_LAUNCH( GP_NONE, true, 12, 16 ); break;
		case 14: GV_LAUNCH( GP_NONE, true, 14, 16 ); break;
		case 16: GV_LAUNCH( GP_NONE, true, 16, 16 ); break;
		case 18: GV_LAUNCH( GP_NONE, true, 18, 32 ); break;
		case 20: GV_LAUNCH( GP_NONE, true, 20, 32 ); break;
		case 22: GV_LAUNCH( GP_NONE, true, 22, 32 ); break;
		case 24: GV_LAUNCH( GP_NONE, true, 24, 32 ); break;
		default: GV_LAUNCH( GP_NONE, true, 0, 0 ); break;
	}
}

} // namespace avirhip
