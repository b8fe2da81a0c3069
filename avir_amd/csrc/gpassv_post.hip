// gpassv_post.hip -- k_gv variants of the gather-then-FIR7 axes (AVIR 1 < k < 2:
// 13 .. 25 taps in registers; longer gathers through the LDS window).
#include "gpassv_kernel.h"

namespace avirhip {

void launch_gv_post( const GVParams& P, int wp, int items, size_t lds,
	hipStream_t st )
{
	const int nt = P.ax.nt;

	switch( nt )
	{
		// (13 .. 25 taps: the 1 < k < 2 plans, avir.h bank lengths 26 .. 50
		// over the zero-stuffed view)
		case 13: GV_LAUNCH( GP_POST, false, 13, 16 ); break;
		case 14: GV_LAUNCH( GP_POST, false, 14, 16 ); break;
		case 15: GV_LAUNCH( GP_POST, false, 15, 16 ); break;
		case 16: GV_LAUNCH( GP_POST, false, 16, 16 ); break;
		case 17: GV_LAUNCH( GP_POST, false, 17, 32 ); break;
		case 18: GV_LAUNCH( GP_POST, false, 18, 32 ); break;
		case 19: GV_LAUNCH( GP_POST, false, 19, 32 ); break;
		case 20: GV_LAUNCH( GP_POST, false, 20, 32 ); break;
		case 21: GV_LAUNCH( GP_POST, false, 21, 32 ); break;
		case 22: GV_LAUNCH( GP_POST, false, 22, 32 ); break;
		case 23: GV_LAUNCH( GP_POST, false, 23, 32 ); break;
		case 24: GV_LAUNCH( GP_POST, false, 24, 32 ); break;
		case 25: GV_LAUNCH( GP_POST, false, 25, 32 ); break;
		default: GV_LAUNCH( GP_POST, false, 0, 0 ); break;
	}
}

} // namespace avirhip
