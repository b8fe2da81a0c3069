/* placeholder -- replaced by the restatement */
#include "oracle.h"
int oracle_avir_resize( const void* src, int sw, int sh, int sstride,
	void* dst, int nw, int nh, int ch, double k, int in_type, int out_type,
	int resbits, int srcbits, const double* params, double ox, double oy,
	int build_mode, int* mode_h, int* mode_v ) { return( -100 ); }
