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
// MODE 6-9: the same operations with the other kind of coefficient operand (VGPR <-> SGPR); MODE 10: the quad, coefficients in VGPRs
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
		if (MODE == 11) asm volatile( // the quad with the scalar pair as src1 everywhere
			"v_pk_mul_f32 %6, %12, %14 op_sel_hi:[1,0]\n v_pk_mul_f32 %7, %12, %14 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %8, %13, %14 op_sel_hi:[0,1]\n"
			"v_pk_mul_f32 %9, %12, %15 op_sel_hi:[1,0]\n v_pk_mul_f32 %10, %12, %15 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %11, %13, %15 op_sel_hi:[0,1]\n"
			"v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %7\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n v_pk_add_f32 %4, %4, %10\n v_pk_add_f32 %5, %5, %11"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5])
			: "v"(x0), "v"(x1), "s"(c0), "s"(c1));
		if (MODE == 6) asm volatile(
			"v_pk_mul_f32 %0, %0, %6\n v_pk_mul_f32 %1, %1, %6\n v_pk_mul_f32 %2, %2, %6\n v_pk_mul_f32 %3, %3, %6\n v_pk_mul_f32 %4, %4, %6\n v_pk_mul_f32 %5, %5, %6\n"
			"v_pk_mul_f32 %0, %0, %7\n v_pk_mul_f32 %1, %1, %7\n v_pk_mul_f32 %2, %2, %7\n v_pk_mul_f32 %3, %3, %7\n v_pk_mul_f32 %4, %4, %7\n v_pk_mul_f32 %5, %5, %7"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]) : "v"(c0), "v"(c1));
		if (MODE == 7) asm volatile(
			"v_mul_f32 %0, %6, %0\n v_mul_f32 %1, %6, %1\n v_mul_f32 %2, %6, %2\n v_mul_f32 %3, %6, %3\n v_mul_f32 %4, %6, %4\n v_mul_f32 %5, %6, %5\n"
			"v_mul_f32 %0, %7, %0\n v_mul_f32 %1, %7, %1\n v_mul_f32 %2, %7, %2\n v_mul_f32 %3, %7, %3\n v_mul_f32 %4, %7, %4\n v_mul_f32 %5, %7, %5"
			: "+v"(a[0].x), "+v"(a[1].x), "+v"(a[2].x), "+v"(a[3].x), "+v"(a[4].x), "+v"(a[5].x) : "v"(c0.x), "v"(c1.x));
		if (MODE == 8) asm volatile(
			"v_add_f32 %0, %6, %0\n v_add_f32 %1, %6, %1\n v_add_f32 %2, %6, %2\n v_add_f32 %3, %6, %3\n v_add_f32 %4, %6, %4\n v_add_f32 %5, %6, %5\n"
			"v_add_f32 %0, %7, %0\n v_add_f32 %1, %7, %1\n v_add_f32 %2, %7, %2\n v_add_f32 %3, %7, %3\n v_add_f32 %4, %7, %4\n v_add_f32 %5, %7, %5"
			: "+v"(a[0].x), "+v"(a[1].x), "+v"(a[2].x), "+v"(a[3].x), "+v"(a[4].x), "+v"(a[5].x) : "s"(c0.x), "s"(c1.x));
		if (MODE == 9) asm volatile(
			"v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %6\n v_pk_add_f32 %2, %2, %6\n v_pk_add_f32 %3, %3, %6\n v_pk_add_f32 %4, %4, %6\n v_pk_add_f32 %5, %5, %6\n"
			"v_pk_add_f32 %0, %0, %7\n v_pk_add_f32 %1, %1, %7\n v_pk_add_f32 %2, %2, %7\n v_pk_add_f32 %3, %3, %7\n v_pk_add_f32 %4, %4, %7\n v_pk_add_f32 %5, %5, %7"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]) : "s"(c0), "s"(c1));
		if (MODE == 10) asm volatile( // the quad with the coefficient pairs in VGPRs
			"v_pk_mul_f32 %6, %12, %14 op_sel_hi:[1,0]\n v_pk_mul_f32 %7, %12, %14 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %8, %14, %13 op_sel_hi:[1,0]\n"
			"v_pk_mul_f32 %9, %12, %15 op_sel_hi:[1,0]\n v_pk_mul_f32 %10, %12, %15 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %11, %15, %13 op_sel_hi:[1,0]\n"
			"v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %7\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n v_pk_add_f32 %4, %4, %10\n v_pk_add_f32 %5, %5, %11"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5])
			: "v"(x0), "v"(x1), "v"(c0), "v"(c1));
	}
	float s = 0; for (int i = 0; i < 6; i++) s += a[i].x + a[i].y;
	out[blockIdx.x * 64 + threadIdx.x] = s;
}
// the same arithmetic on 12 accumulators: D == 0: two quads back to back (a product is added 6 instructions after
// its multiply); D == 1: 12 multiplies, then 12 adds (distance 12); D == 2: software-pipelined -- the multiplies of
// the next 6 accumulators go out before the adds of the previous 6 (distance 12, 6 temporaries more)
template<int D> __global__ void __launch_bounds__(64) k12(float* out, f2 c0, f2 c1, int iters)
{
	f2 a[12], p[12], x0 = {1.0f + threadIdx.x, 2.0f};
	for (int i = 0; i < 12; i++) { a[i] = (f2) (float) i; p[i] = (f2) 0.0f; }
	for (int it = 0; it < iters; it++) {
		if (D == 0) {
#define Q6(o) asm volatile( \
			"v_pk_mul_f32 %6, %12, %13 op_sel_hi:[1,0]\n v_pk_mul_f32 %7, %12, %13 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %8, %12, %14 op_sel_hi:[1,0]\n" \
			"v_pk_mul_f32 %9, %12, %14 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %10, %12, %13 op_sel_hi:[1,0]\n v_pk_mul_f32 %11, %12, %14 op_sel_hi:[1,0]\n" \
			"v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %7\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n v_pk_add_f32 %4, %4, %10\n v_pk_add_f32 %5, %5, %11" \
			: "+v"(a[o]), "+v"(a[o+1]), "+v"(a[o+2]), "+v"(a[o+3]), "+v"(a[o+4]), "+v"(a[o+5]), "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]) \
			: "v"(x0), "s"(c0), "s"(c1))
			Q6(0); Q6(6);
		}
		if (D == 1) {
			asm volatile(
			"v_pk_mul_f32 %0, %12, %13 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %12, %13 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %2, %12, %14 op_sel_hi:[1,0]\n"
			"v_pk_mul_f32 %3, %12, %14 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %4, %12, %13 op_sel_hi:[1,0]\n v_pk_mul_f32 %5, %12, %14 op_sel_hi:[1,0]\n"
			"v_pk_mul_f32 %6, %12, %13 op_sel_hi:[1,0]\n v_pk_mul_f32 %7, %12, %13 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %8, %12, %14 op_sel_hi:[1,0]\n"
			"v_pk_mul_f32 %9, %12, %14 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %10, %12, %13 op_sel_hi:[1,0]\n v_pk_mul_f32 %11, %12, %14 op_sel_hi:[1,0]"
			: "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]), "=&v"(p[8]), "=&v"(p[9]), "=&v"(p[10]), "=&v"(p[11])
			: "v"(x0), "s"(c0), "s"(c1));
			asm volatile(
			"v_pk_add_f32 %0, %0, %12\n v_pk_add_f32 %1, %1, %13\n v_pk_add_f32 %2, %2, %14\n v_pk_add_f32 %3, %3, %15\n v_pk_add_f32 %4, %4, %16\n v_pk_add_f32 %5, %5, %17\n"
			"v_pk_add_f32 %6, %6, %18\n v_pk_add_f32 %7, %7, %19\n v_pk_add_f32 %8, %8, %20\n v_pk_add_f32 %9, %9, %21\n v_pk_add_f32 %10, %10, %22\n v_pk_add_f32 %11, %11, %23"
			: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11])
			: "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7]), "v"(p[8]), "v"(p[9]), "v"(p[10]), "v"(p[11]));
		}
	}
	float s = 0; for (int i = 0; i < 12; i++) s += a[i].x + a[i].y;
	out[blockIdx.x * 64 + threadIdx.x] = s;
}
// the row-wave statement of dnf.hip: one input, NA taps: NA v_pk_mul_f32 (SGPR pair, op_sel) then NA v_pk_add_f32
// into NA accumulators; the statement repeats for the next input on the SAME accumulators
template<int NA> __global__ void __launch_bounds__(64) kna(float* out, f2 c0, f2 c1, int iters)
{
	f2 a[6], p[6], x0 = {1.0f + threadIdx.x, 2.0f};
	for (int i = 0; i < 6; i++) { a[i] = (f2) (float) i; p[i] = (f2) 0.0f; }
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < 12 / NA; r++) {
			if (NA == 2) asm volatile("v_pk_mul_f32 %2, %4, %5 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %4, %6 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_add_f32 %0, %0, %2\n v_pk_add_f32 %1, %1, %3"
				: "+v"(a[0]), "+v"(a[1]), "=&v"(p[0]), "=&v"(p[1]) : "v"(x0), "s"(c0), "s"(c1));
			if (NA == 3) asm volatile("v_pk_mul_f32 %3, %6, %7 op_sel_hi:[1,0]\n v_pk_mul_f32 %4, %6, %8 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %5, %6, %8 op_sel_hi:[1,0]\n"
				"v_pk_add_f32 %0, %0, %3\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %5"
				: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]) : "v"(x0), "s"(c0), "s"(c1));
			if (NA == 6) asm volatile("v_pk_mul_f32 %6, %12, %13 op_sel_hi:[1,0]\n v_pk_mul_f32 %7, %12, %14 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %8, %12, %14 op_sel_hi:[1,0]\n"
				"v_pk_mul_f32 %9, %12, %13 op_sel:[0,1] op_sel_hi:[1,1]\n v_pk_mul_f32 %10, %12, %14 op_sel_hi:[1,0]\n v_pk_mul_f32 %11, %12, %13 op_sel_hi:[1,0]\n"
				"v_pk_add_f32 %0, %0, %6\n v_pk_add_f32 %1, %1, %7\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %9\n v_pk_add_f32 %4, %4, %10\n v_pk_add_f32 %5, %5, %11"
				: "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]) : "v"(x0), "s"(c0), "s"(c1));
		}
	}
	float s = 0; for (int i = 0; i < 6; i++) s += a[i].x + a[i].y;
	out[blockIdx.x * 64 + threadIdx.x] = s;
}
template<int NA> static int runna(const char* label, float* out)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const int iters = 50000;
	for (int wps = 1; wps <= 4; wps++) {
		const int grid = 1024 * wps; f2 c0 = {1.0001f, 0.9999f}, c1 = {1.00001f, 0.99999f};
		hipLaunchKernelGGL(kna<NA>, dim3(grid), dim3(64), 0, 0, out, c0, c1, iters); CK(hipDeviceSynchronize());
		CK(hipEventRecord(e0)); hipLaunchKernelGGL(kna<NA>, dim3(grid), dim3(64), 0, 0, out, c0, c1, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		printf("%-46s %d wave(s)/SIMD: %7.3f ms  %6.3f ns per wave-instruction per SIMD\n", label, wps, ms, ms * 1e6 / (24.0 * iters * wps));
	}
	return 0;
}
template<int D> static int run12(const char* label, float* out)
{
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	const int iters = 50000;
	for (int wps = 1; wps <= 4; wps *= 2) {
		const int grid = 1024 * wps; f2 c0 = {1.0001f, 0.9999f}, c1 = {1.00001f, 0.99999f};
		hipLaunchKernelGGL(k12<D>, dim3(grid), dim3(64), 0, 0, out, c0, c1, iters); CK(hipDeviceSynchronize());
		CK(hipEventRecord(e0)); hipLaunchKernelGGL(k12<D>, dim3(grid), dim3(64), 0, 0, out, c0, c1, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		printf("%-46s %d wave(s)/SIMD: %7.3f ms  %6.3f ns per wave-instruction per SIMD\n", label, wps, ms, ms * 1e6 / (24.0 * iters * wps));
	}
	return 0;
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
	run<6>("12 v_pk_mul_f32 v,v,v", out);
	run<7>("12 v_mul_f32 v,v,v", out);
	run<8>("12 v_add_f32 v,s,v", out);
	run<9>("12 v_pk_add_f32 v,v,s", out);
	run<10>("quad with the coefficients in VGPRs", out);
	run<11>("quad, scalar pair as src1 in all six multiplies", out);
	runna<2>("statement of 2 taps on 2 accumulators", out);
	runna<3>("statement of 3 taps on 3 accumulators", out);
	runna<6>("statement of 6 taps on 6 accumulators", out);
	run12<0>("2 x (6 pk_mul + 6 pk_add): add 6 after its mul", out);
	run12<1>("12 pk_mul, then 12 pk_add: add 12 after its mul", out);
	return 0;
}
