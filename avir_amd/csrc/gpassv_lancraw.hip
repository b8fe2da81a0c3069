// gpassv_lancraw.hip -- the same variants with the raw-source loader (inner
// plans of integer / narrow LANCIR images; whole-pixel lanes).
#include "gpassv_kernel.h"

namespace avirhip {

// returns 1 when no variant exists for the plan's tap count (nothing launched)
int launch_gv_lanc_raw( const GVParams& P, int items, size_t lds,
	hipStream_t st )
{
	const int nt = P.ax.nt;

	switch( nt )
	{
		case 6: GV_LAUNCH_LR( 6, 8 ); break;
		case 8: GV_LAUNCH_LR( 8, 8 ); break;
		case 10: GV_LAUNCH_LR( 10, 16 ); break;
		case 12: GV_LAUNCH_LR( 12, 16 ); break;
		case 14: GV_LAUNCH_LR( 14, 16 ); break;
		case 16: GV_LAUNCH_LR( 16, 16 ); break;
		case 18: GV_LAUNCH_LR( 18, 32 ); break;
		case 20: GV_LAUNCH_LR( 20, 32 ); break;
		case 22: GV_LAUNCH_LR( 22, 32 ); break;
		case 24: GV_LAUNCH_LR( 24, 32 ); break;
		// (other tap counts keep their window in an LDS ring of pixels, which
		// raw bytes cannot feed: gpass_lancir_takes_raw refuses them -- a caller
		// that comes here all the same must not take silence for success)
		default: return( 1 );
	}

	return( 0 );
}

} // namespace avirhip
