// tools/pcie_bench.hip -- what the host-pointer call can hope for on this box: pageable and
// registered (pinned in place) copies of cfg3's buffers, alone and in both directions at once.
// hipcc --offload-arch=gfx950 -O2 tools/pcie_bench.hip -o tools/pcie_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main()
{
	const size_t SB = 3840ull * 2160 * 16, DB = 7680ull * 4320 * 16;
	char* hs = (char*) aligned_alloc(4096, SB); char* hd = (char*) aligned_alloc(4096, DB);
	memset(hs, 1, SB); memset(hd, 2, DB);
	void *ds, *dd; CK(hipMalloc(&ds, SB)); CK(hipMalloc(&dd, DB));
	hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
	for (int rep = 0; rep < 3; rep++) {
		double t0 = now(); CK(hipMemcpy(ds, hs, SB, hipMemcpyHostToDevice)); double t1 = now();
		CK(hipMemcpy(hd, dd, DB, hipMemcpyDeviceToHost)); double t2 = now();
		printf("pageable: H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, SB / (t1 - t0) / 1e9, (t2 - t1) * 1e3, DB / (t2 - t1) / 1e9);
	}
	for (int rep = 0; rep < 3; rep++) {
		double t0 = now(); CK(hipHostRegister(hs, SB, hipHostRegisterDefault)); double t1 = now();
		CK(hipHostRegister(hd, DB, hipHostRegisterDefault)); double t2 = now();
		CK(hipMemcpyAsync(ds, hs, SB, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); double t3 = now();
		CK(hipMemcpyAsync(hd, dd, DB, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); double t4 = now();
		CK(hipMemcpyAsync(ds, hs, SB, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(hd, dd, DB, hipMemcpyDeviceToHost, s2));
		CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); double t5 = now();
		CK(hipHostUnregister(hs)); CK(hipHostUnregister(hd)); double t6 = now();
		printf("register src %.2f ms dst %.2f ms | pinned H2D %.2f ms (%.1f GB/s) D2H %.2f ms (%.1f GB/s) | both at once %.2f ms | unregister %.2f ms\n",
			(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, SB / (t3 - t2) / 1e9, (t4 - t3) * 1e3, DB / (t4 - t3) / 1e9, (t5 - t4) * 1e3, (t6 - t5) * 1e3);
	}
	// banded: 8 bands, H2D(b+1) || D2H(b-1)
	CK(hipHostRegister(hs, SB, hipHostRegisterDefault)); CK(hipHostRegister(hd, DB, hipHostRegisterDefault));
	for (int rep = 0; rep < 3; rep++) {
		double t0 = now();
		for (int b = 0; b < 8; b++) {
			CK(hipMemcpyAsync((char*) ds + SB / 8 * b, hs + SB / 8 * b, SB / 8, hipMemcpyHostToDevice, s1));
			CK(hipMemcpyAsync(hd + DB / 8 * b, (char*) dd + DB / 8 * b, DB / 8, hipMemcpyDeviceToHost, s2));
		}
		CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
		printf("8 bands both directions (registered): %.2f ms\n", (now() - t0) * 1e3);
	}
	return 0;
}
