// gpassv_lanc.hip -- k_gv variants of LANCIR plans (even tap counts 6 .. 24
// in registers, anything else through the LDS window).
#include "gpassv_kernel.h"

namespace avirhip {

int launch_gv_lanc( const GVParams& P, int wp, int items, size_t lds,
	hipStream_t st )
{
	const int nt = P.ax.nt;

	if( wp == 2 && P.raw != nullptr )
	{
		return( P.raw_kind == 1 ? launch_gv_lanc_raw8( P, items, lds, st ) :
			launch_gv_lanc_raw( P, items, lds, st ));
	}

	switch( nt )
	{
		case 6: GV_LAUNCH_L( 6, 8 ); break;
		case 8: GV_LAUNCH_L( 8, 8 ); break;
		case 10: GV_LAUNCH_L( 10, 16 ); break;
		case 12: GV_LAUNCH_L( 12, 16 ); break;
		case 14: GV_LAUNCH_L( 14, 16 ); break;
		case 16: GV_LAUNCH_L( 16, 16 ); break;
		case 18: GV_LAUNCH_L( 18, 32 ); break;
		case 20: GV_LAUNCH_L( 20, 32 ); break;
		case 22: GV_LAUNCH_L( 22, 32 ); break;
		case 24: GV_LAUNCH_L( 24, 32 ); break;
		default: GV_LAUNCH_L( 0, 0 ); break;
	}

	return( 0 );
}

} // namespace avirhip
