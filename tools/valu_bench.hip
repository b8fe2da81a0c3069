// tools/valu_bench.hip -- microbenchmarks that steer the fused-kernel design
// (not part of the product): non-FMA f32 VALU rate (v_mul+v_add vs packed
// v_pk_mul/v_pk_add), ds_read_b128 rate, and float4 copy bandwidth.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

template<int N> __global__ void __launch_bounds__(256) k_scalar(float* out, float c0, float c1, int iters)
{
	float a[N];
	for (int i = 0; i < N; i++) a[i] = threadIdx.x * 1e-3f + i;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < N; i++) { float p = a[i] * c0; a[i] = p + c1; }
	}
	float s = 0; for (int i = 0; i < N; i++) s += a[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template<int N> __global__ void __launch_bounds__(256) k_packed(float* out, float c0, float c1, int iters)
{
	float2v a[N];
	for (int i = 0; i < N; i++) { a[i].x = threadIdx.x * 1e-3f + i; a[i].y = a[i].x + 0.5f; }
	float2v k0 = {c0, c0}, k1 = {c1, c1};
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < N; i++) { float2v p = a[i] * k0; a[i] = p + k1; }
	}
	float s = 0; for (int i = 0; i < N; i++) s += a[i].x + a[i].y;
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_lds(float* out, int iters)
{
	__shared__ float4v buf[2048];
	for (int i = threadIdx.x; i < 2048; i += 256) buf[i] = float4v{(float)i, 1.f, 2.f, 3.f};
	__syncthreads();
	float4v acc = {0, 0, 0, 0};
	int idx = threadIdx.x;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int j = 0; j < 16; j++) { acc += buf[(idx + j * 67) & 2047]; }
		idx = (idx + 1) & 2047;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

__global__ void __launch_bounds__(256) k_copy(const float4v* __restrict__ in, float4v* __restrict__ out, size_t n)
{
	size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
	size_t stride = (size_t) gridDim.x * blockDim.x;
	for (; i < n; i += stride) out[i] = in[i];
}


#include <functional>
static hipEvent_t e0, e1;
static int timeit(const char* label, std::function<void()> launch, double ops)
{
	float ms;
	launch(); CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	CK(hipEventElapsedTime(&ms, e0, e1));
	printf("%-36s %8.3f ms  %8.2f T/s\n", label, ms, ops / ms * 1e-9);
	return 0;
}

int main()
{
	float* out; CK(hipMalloc(&out, 256 * 2048 * 8 * sizeof(float)));
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const int iters = 4096;
	const int blocks = 256 * 8;
	const double th = (double) blocks * 256 * iters;
	timeit("scalar mul+add N=8 (Tflop/s)", [&]{ k_scalar<8><<<blocks,256>>>(out, 1.0001f, 0.5f, iters); }, th * 8 * 2);
	timeit("scalar mul+add N=16 (Tflop/s)", [&]{ k_scalar<16><<<blocks,256>>>(out, 1.0001f, 0.5f, iters); }, th * 16 * 2);
	timeit("packed mul+add N=8 (Tflop/s)", [&]{ k_packed<8><<<blocks,256>>>(out, 1.0001f, 0.5f, iters); }, th * 8 * 4);
	timeit("packed mul+add N=16 (Tflop/s)", [&]{ k_packed<16><<<blocks,256>>>(out, 1.0001f, 0.5f, iters); }, th * 16 * 4);
	const int li = 2048;
	timeit("ds_read_b128 (TB/s)", [&]{ k_lds<<<blocks,256>>>(out, li); }, (double) blocks * 256 * li * 16 * 16);
	size_t n = (size_t) 1 << 26; // 64M float4 = 1 GiB
	float4v *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMemset(a, 1, n * 16));
	for (int g = 2048; g <= 32768; g *= 2) {
		char lbl[64]; snprintf(lbl, 64, "copy float4 1GiB grid %d (TB/s)", g);
		timeit(lbl, [&]{ k_copy<<<g,256>>>(a, b, n); }, (double) n * 32);
	}
	return 0;
}
