// gpassv_lancraw8.hip -- the raw-source variants for uint8 pixels (the element
// type at compile time; inner plans of CLancIR's 8-bit images).
#include "gpassv_kernel.h"

namespace avirhip {

// returns 1 when no variant exists for the plan's tap count (nothing launched)
int launch_gv_lanc_raw8( const GVParams& P, int items, size_t lds,
	hipStream_t st )
{
	const int nt = P.ax.nt;

	switch( nt )
	{
		case 6: GV_LAUNCH_LR8( 6, 8 ); break;
		case 8: GV_LAUNCH_LR8( 8, 8 ); break;
		case 10: GV_LAUNCH_LR8( 10, 16 ); break;
		case 12: GV_LAUNCH_LR8( 12, 16 ); break;
		case 14: GV_LAUNCH_LR8( 14, 16 ); break;
		case 16: GV_LAUNCH_LR8( 16, 16 ); break;
		case 18: GV_LAUNCH_LR8( 18, 32 ); break;
		case 20: GV_LAUNCH_LR8( 20, 32 ); break;
		case 22: GV_LAUNCH_LR8( 22, 32 ); break;
		case 24: GV_LAUNCH_LR8( 24, 32 ); break;
		// (other tap counts keep their window in an LDS ring of pixels, which
		// raw bytes cannot feed: gpass_lancir_takes_raw refuses them -- a caller
		// that comes here all the same must not take silence for success)
		default: return( 1 );
	}

	return( 0 );
}

} // namespace avirhip
