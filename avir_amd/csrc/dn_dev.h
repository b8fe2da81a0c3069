// dn_dev.h -- what the integer-ratio downsizing kernels share (dn.hip: the
// two passes; dnf.hip: both axes fused in one marching launch).
#ifndef AVIRHIP_DN_DEV_H
#define AVIRHIP_DN_DEV_H

#include "gpass_dev.h"
#include <type_traits>
#include <utility>

namespace avirhip {

#define DN_MAXNT 40

// One axis: T[j] = 0 + sum_{t<NT} c[t] * S[clamp(K*j + c0 + t)] (doResize,
// avir.h:3884-4096, one bank phase), then the 7-tap correction filter
// (doFilter, avir.h:3748-3866) over T with replicated edge values.
struct DnAxisP
{
	int K, NT, c0;       // offset of output j: K*j + c0
	int in_len, out_len;
	const float* coef;   // device: NT resize coefficients, then 4 FIR taps
};

// host side
struct DnAxis
{
	int K, NT, c0, in_len, out_len;
	float* d_coef;
};

struct DnData
{
	bool hok, vok;
	DnAxis h, v;
	std::vector< void* > allocs;
	int ncu; // compute units of the plan's device
};

__device__ __forceinline__ int dn_clampi( int i, int hi )
{
	return( i < 0 ? 0 : ( i > hi ? hi : i ));
}

// compile-time loop: every index below must be a constant so that the arrays
// stay in registers (a rolled loop would index them dynamically -> scratch)
template< int... I, class F >
__device__ __forceinline__ void dn_static_for_impl(
	std::integer_sequence< int, I... >, F&& f )
{
	( f( std::integral_constant< int, I >()), ... );
}

template< int N, class F >
__device__ __forceinline__ void dn_static_for( F&& f )
{
	dn_static_for_impl( std::make_integer_sequence< int, N >(), f );
}

} // namespace avirhip

#endif
