// tools/pk_bench.hip -- issue rate of the packed / scalar fp32 instructions the accumulation kernels are made of,
// per SIMD: nanoseconds per wave-instruction with 1, 2 and 4 waves on a SIMD (256 CUs x 4 SIMDs busy).
//   hipcc --offload-arch=gfx950 -O3 -o tools/pk_bench tools/pk_bench.hip && tools/pk_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: the quad of k_sacc2 (RGB): 6 v_pk_mul_f32 with an SGPR pair + op_sel, 6 v_pk_add_f32   (12 instr)
// MODE 1: 12 v_pk_add_f32 v,v,v      MODE 2: 12 v_pk_mul_f32 v,v,s      MODE 3: 12 v_mul_f32 v,s,v
// MODE 4: 12 v_add_f32 v,v,v         MODE 5: 6 v_mul_f32 + 6 v_add_f32 (scalar form of the quad, half the work)
template<int MODE> __global__ void __launch_bounds__(64) k(float* out, f2 c0, f2 c1, int iters)
{
	f2 a[6], p[6], x0 = {1.0f + threadIdx.x, 2.0f}, x1 = {0.5f, 0.25f};
	for (int i = 0; i < 6; i++) { a[i] = (f2) (float) i; p[i] = (f2) 0.0f; }
	for (int it = 0; it < iters; it++) {
		if (MODE == 0) asm volatile(
			"v_pk_mul_f32 %6, %12, %14 op_sel_hi:[1,0]\n v_pk_mul_f32 %7, %12, %14 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %8, %14, %13 op_sel_hi:[1,0]\n"
			"v_pk_mul_f32 %9, %12, %15 op_sel_hi:[1,0]\n v_pk_mul_f32 %10, %12, %15 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %11, %15, %13 op_sel_hi:[1,0]\n"
			"v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %7\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n v_pk_add_f32 %4, %4, %10\n v_pk_add_f32 %5, %5, %11"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5])
			: "v"(x0), "v"(x1), "s"(c0), "s"(c1));
		if (MODE == 1) asm volatile(
			"v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %6\n v_pk_add_f32 %2, %2, %6\n v_pk_add_f32 %3, %3, %6\n v_pk_add_f32 %4, %4, %6\n v_pk_add_f32 %5, %5, %6\n"
			"v_pk_add_f32 %0, %0, %7\n v_pk_add_f32 %1, %1, %7\n v_pk_add_f32 %2, %2, %7\n v_pk_add_f32 %3, %3, %7\n v_pk_add_f32 %4, %4, %7\n v_pk_add_f32 %5, %5, %7"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]) : "v"(x0), "v"(x1));
		if (MODE == 2) asm volatile(
			"v_pk_mul_f32 %0, %0, %6\n v_pk_mul_f32 %1, %1, %6\n v_pk_mul_f32 %2, %2, %6\n v_pk_mul_f32 %3, %3, %6\n v_pk_mul_f32 %4, %4, %6\n v_pk_mul_f32 %5, %5, %6\n"
			"v_pk_mul_f32 %0, %0, %7\n v_pk_mul_f32 %1, %1, %7\n v_pk_mul_f32 %2, %2, %7\n v_pk_mul_f32 %3, %3, %7\n v_pk_mul_f32 %4, %4, %7\n v_pk_mul_f32 %5, %5, %7"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]) : "s"(c0), "s"(c1));
		if (MODE == 3) asm volatile(
			"v_mul_f32 %0, %6, %0\n v_mul_f32 %1, %6, %1\n v_mul_f32 %2, %6, %2\n v_mul_f32 %3, %6, %3\n v_mul_f32 %4, %6, %4\n v_mul_f32 %5, %6, %5\n"
			"v_mul_f32 %0, %7, %0\n v_mul_f32 %1, %7, %1\n v_mul_f32 %2, %7, %2\n v_mul_f32 %3, %7, %3\n v_mul_f32 %4, %7, %4\n v_mul_f32 %5, %7, %5"
			: "+v"(a[0].x), "+v"(a[1].x), "+v"(a[2].x), "+v"(a[3].x), "+v"(a[4].x), "+v"(a[5].x) : "s"(c0.x), "s"(c1.x));
		if (MODE == 4) asm volatile(
			"v_add_f32 %0, %6, %0\n v_add_f32 %1, %6, %1\n v_add_f32 %2, %6, %2\n v_add_f32 %3, %6, %3\n v_add_f32 %4, %6, %4\n v_add_f32 %5, %6, %5\n"
			"v_add_f32 %0, %7, %0\n v_add_f32 %1, %7, %1\n v_add_f32 %2, %7, %2\n v_add_f32 %3, %7, %3\n v_add_f32 %4, %7, %4\n v_add_f32 %5, %7, %5"
			: "+v"(a[0].x), "+v"(a[1].x), "+v"(a[2].x), "+v"(a[3].x), "+v"(a[4].x), "+v"(a[5].x) : "v"(x0.x), "v"(x1.x));
		if (MODE == 5) asm volatile(
			"v_mul_f32 %6, %12, %14\n v_mul_f32 %7, %13, %14\n v_mul_f32 %8, %12, %15\n v_mul_f32 %9, %13, %15\n v_mul_f32 %10, %12, %14\n v_mul_f32 %11, %13, %15\n"
			"v_add_f32 %0, %0, %6\n v_add_f32 %1, %1, %7\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n v_add_f32 %4, %4, %10\n v_add_f32 %5, %5, %11"
			: "+v"(a[0].x), "+v"(a[1].x), "+v"(a[2].x), "+v"(a[3].x), "+v"(a[4].x), "+v"(a[5].x), "=&v"(p[0].x), "=&v"(p[1].x), "=&v"(p[2].x), "=&v"(p[3].x), "=&v"(p[4].x), "=&v"(p[5].x)
			: "s"(c0.x), "s"(c0.y), "v"(x0.x), "v"(x1.x));
	}
	float s = 0; for (int i = 0; i < 6; i++) s += a[i].x + a[i].y;
	out[blockIdx.x * 64 + threadIdx.x] = s;
}
template<int MODE> static int run(const char* label, float* out)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const int iters = 100000;
	for (int wps = 1; wps <= 4; wps *= 2) {
		const int grid = 1024 * wps; f2 c0 = {1.0001f, 0.9999f}, c1 = {1.00001f, 0.99999f};
		hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, c0, c1, iters); CK(hipDeviceSynchronize());
		CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, c0, c1, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		printf("%-46s %d wave(s)/SIMD: %7.3f ms  %6.3f ns per wave-instruction per SIMD\n", label, wps, ms, ms * 1e6 / (12.0 * iters * wps));
	}
	return 0;
}
int main()
{
	float* out; CK(hipMalloc(&out, 4096 * 64 * sizeof(float)));
	run<0>("quad: 6 pk_mul(v,s op_sel) + 6 pk_add", out);
	run<1>("12 v_pk_add_f32 v,v,v", out);
	run<2>("12 v_pk_mul_f32 v,v,s", out);
	run<3>("12 v_mul_f32 v,s,v", out);
	run<4>("12 v_add_f32 v,v,v", out);
	run<5>("6 v_mul_f32 + 6 v_add_f32", out);
	return 0;
}
