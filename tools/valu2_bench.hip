// tools/valu2_bench.hip -- does a v_pk_add_f32 / v_add_f32 with TWO VGPR sources run at
// the same rate as one with an SGPR source? (design probe for up2.hip)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <functional>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float f2 __attribute__((ext_vector_type(2)));
// mul by SGPR pair, add VGPR (two-VGPR-source add), N independent chains
template<int N> __global__ void __launch_bounds__(256) k_pk_vv(float* out, float c0, int iters)
{
	f2 a[N], b[N];
	for (int i = 0; i < N; i++) { a[i].x = threadIdx.x * 1e-3f + i; a[i].y = a[i].x + 0.5f; b[i].x = 0.25f + threadIdx.x; b[i].y = 0.125f * i + threadIdx.x; }
	f2 k0 = {c0, c0};
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < N; i++) { f2 p = a[i] * k0; a[i] = p + b[i]; }
	}
	float s = 0; for (int i = 0; i < N; i++) s += a[i].x + a[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// like the kernel: acc += coef(SGPR) * w[i] (VGPR): mul(s,v) then add(v,v) into ONE or TWO accumulators
template<int NACC> __global__ void __launch_bounds__(256) k_dot(float* out, float c0, int iters)
{
	f2 w[12]; f2 acc[NACC];
	for (int i = 0; i < 12; i++) { w[i].x = threadIdx.x * 1e-3f + i; w[i].y = w[i].x + 0.5f; }
	for (int j = 0; j < NACC; j++) acc[j] = (f2) 0.0f;
	f2 k0 = {c0, c0};
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 12; i++) {
#pragma unroll
			for (int j = 0; j < NACC; j++) acc[j] = acc[j] + k0 * w[(i + j) % 12];
		}
	}
	float s = 0; for (int j = 0; j < NACC; j++) s += acc[j].x + acc[j].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template<int N> __global__ void __launch_bounds__(256) k_sc_vv(float* out, float c0, int iters)
{
	float a[N], b[N];
	for (int i = 0; i < N; i++) { a[i] = threadIdx.x * 1e-3f + i; b[i] = 0.25f * i + threadIdx.x; }
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < N; i++) { float p = a[i] * c0; a[i] = p + b[i]; }
	}
	float s = 0; for (int i = 0; i < N; i++) s += a[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static hipEvent_t e0, e1;
static int timeit(const char* label, std::function<void()> launch, double ops)
{
	float ms; launch(); CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	CK(hipEventElapsedTime(&ms, e0, e1));
	printf("%-44s %8.3f ms  %8.2f Tflop/s\n", label, ms, ops / ms * 1e-9); return 0;
}
int main()
{
	float* out; CK(hipMalloc(&out, 256 * 2048 * 8 * sizeof(float)));
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const int iters = 4096, blocks = 256 * 8; const double th = (double) blocks * 256 * iters;
	timeit("pk mul(s,v)+add(v,v) N=8", [&]{ k_pk_vv<8><<<blocks,256>>>(out, 1.0001f, iters); }, th * 8 * 4);
	timeit("pk mul(s,v)+add(v,v) N=16", [&]{ k_pk_vv<16><<<blocks,256>>>(out, 1.0001f, iters); }, th * 16 * 4);
	timeit("scalar mul(s,v)+add(v,v) N=16", [&]{ k_sc_vv<16><<<blocks,256>>>(out, 1.0001f, iters); }, th * 16 * 2);
	timeit("pk dot12, 1 accumulator", [&]{ k_dot<1><<<blocks,256>>>(out, 1.0001f, iters / 4); }, th / 4 * 12 * 4);
	timeit("pk dot12, 2 accumulators", [&]{ k_dot<2><<<blocks,256>>>(out, 1.0001f, iters / 4); }, th / 4 * 24 * 4);
	timeit("pk dot12, 4 accumulators", [&]{ k_dot<4><<<blocks,256>>>(out, 1.0001f, iters / 4); }, th / 4 * 48 * 4);
	return 0;
}
