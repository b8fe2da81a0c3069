// tools/store_bench.hip -- write-only / read-only bandwidth vs access width (design probe)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <functional>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template<typename T> __global__ void __launch_bounds__(256) k_wr(T* out, size_t n, float v)
{
	size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x, st = (size_t) gridDim.x * blockDim.x;
	T val; for (int c = 0; c < (int)(sizeof(T)/4); c++) val[c] = v + c;
	for (; i < n; i += st) out[i] = val;
}
template<typename T> __global__ void __launch_bounds__(256) k_rd(const T* in, float* out, size_t n)
{
	size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x, st = (size_t) gridDim.x * blockDim.x;
	T acc = (T) 0.0f;
	for (; i < n; i += st) acc += in[i];
	if (acc[0] == 1234.567f) out[0] = acc[0];
}
// two-row pattern like up2: each wave writes 512 B (f2) to row y and 512 B to row y+1
__global__ void __launch_bounds__(256) k_wr_rows_f2(float* out, int nw, int nh)
{
	const int strip = blockIdx.x % (nw / 128), chunk = blockIdx.x / (nw / 128);
	const int col = threadIdx.x >> 1, hf = threadIdx.x & 1;
	f2 v = {1.f, 2.f};
	for (int y = chunk * 64; y < chunk * 64 + 64 && y < nh; y++)
		*(f2*)(out + ((size_t) y * nw + strip * 128 + col) * 4 + hf * 2) = v;
}
__global__ void __launch_bounds__(128) k_wr_rows_f4(float* out, int nw, int nh)
{
	const int strip = blockIdx.x % (nw / 128), chunk = blockIdx.x / (nw / 128);
	const int col = threadIdx.x;
	f4 v = {1.f, 2.f, 3.f, 4.f};
	for (int y = chunk * 64; y < chunk * 64 + 64 && y < nh; y++)
		*(f4*)(out + ((size_t) y * nw + strip * 128 + col) * 4) = v;
}
static hipEvent_t e0, e1;
static int timeit(const char* label, std::function<void()> launch, double bytes)
{
	float ms; launch(); CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0)); for (int i = 0; i < 5; i++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
	CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
	printf("%-44s %8.3f ms  %7.2f TB/s\n", label, ms, bytes / ms * 1e-9); return 0;
}
int main()
{
	const size_t bytes = (size_t) 7680 * 4320 * 16; // the cfg3 destination
	float* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
	float* o; CK(hipMalloc(&o, 64));
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int g = 1024; g <= 16384; g *= 4) {
		char l[80];
		snprintf(l, 80, "write f2 grid %d", g); timeit(l, [&]{ k_wr<f2><<<g,256>>>((f2*) buf, bytes / 8, 1.f); }, bytes);
		snprintf(l, 80, "write f4 grid %d", g); timeit(l, [&]{ k_wr<f4><<<g,256>>>((f4*) buf, bytes / 16, 1.f); }, bytes);
		snprintf(l, 80, "read  f2 grid %d", g); timeit(l, [&]{ k_rd<f2><<<g,256>>>((f2*) buf, o, bytes / 8); }, bytes);
		snprintf(l, 80, "read  f4 grid %d", g); timeit(l, [&]{ k_rd<f4><<<g,256>>>((f4*) buf, o, bytes / 16); }, bytes);
	}
	timeit("write rows f2 (up2 pattern, 4050 blocks)", [&]{ k_wr_rows_f2<<<60 * 68,256>>>(buf, 7680, 4320); }, bytes);
	timeit("write rows f4 (128 thr, 4050 blocks)", [&]{ k_wr_rows_f4<<<60 * 68,128>>>(buf, 7680, 4320); }, bytes);
	return 0;
}
