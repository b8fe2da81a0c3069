/* placeholder -- replaced by the restatement */
#include "oracle.h"
int oracle_lancir_resize( const void* src, int sw, int sh, void* dst, int nw,
	int nh, int ch, int in_type, int out_type, int src_ssize, int new_ssize,
	double kx, double ky, double ox, double oy, double la ) { return( -100 ); }
