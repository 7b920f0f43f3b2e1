// scratch probe: cost of hipMalloc / hipFree vs the virtual-memory API (reserve + create + map) on this box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void touch(char *p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i * 4096 < n) p[i * 4096] = 1; }
int main()
{
	CK(hipSetDevice(0));
	void *w; CK(hipMalloc(&w, 1 << 20)); CK(hipFree(w));
	for (size_t mb : {16, 64, 256, 1024, 4096}) {
		size_t n = mb << 20; void *p;
		double t0 = now(); CK(hipMalloc(&p, n)); double t1 = now();
		touch<<<(n / 4096 + 255) / 256, 256>>>((char*)p, n); CK(hipDeviceSynchronize()); double t2 = now();
		CK(hipFree(p)); double t3 = now();
		printf("hipMalloc %5zu MB: %.3f ms, first touch %.3f ms, hipFree %.3f ms\n", mb, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3);
	}
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
	size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
	size_t gmin = 0; CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
	printf("granularity: recommended %zu, minimum %zu\n", gran, gmin);
	size_t va = (size_t)64 << 30; void *base;
	double t0 = now(); CK(hipMemAddressReserve(&base, va, 0, nullptr, 0)); printf("reserve 64 GB of addresses: %.3f ms\n", (now() - t0) * 1e3);
	hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
	size_t off = 0; hipMemGenericAllocationHandle_t hs[64]; int nh = 0;
	for (size_t mb : {2, 16, 64, 256, 1024, 4096}) {
		size_t n = mb << 20;
		double a = now(); CK(hipMemCreate(&hs[nh], n, &prop, 0)); double b = now();
		CK(hipMemMap((char*)base + off, n, 0, hs[nh], 0)); double c = now();
		CK(hipMemSetAccess((char*)base + off, n, &acc, 1)); double d = now();
		touch<<<(n / 4096 + 255) / 256, 256>>>((char*)base + off, n); CK(hipDeviceSynchronize()); double e = now();
		printf("vmm %5zu MB at +%zu MB: create %.3f, map %.3f, access %.3f, touch %.3f ms\n", mb, off >> 20, (b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3, (e - d) * 1e3);
		off += n; ++nh;
	}
	// whole range usable as one buffer?
	touch<<<(off / 4096 + 255) / 256, 256>>>((char*)base, off); CK(hipDeviceSynchronize());
	CK(hipMemset(base, 0, off)); CK(hipDeviceSynchronize());
	// does mapping while a kernel runs block?
	{
		hipStream_t st; CK(hipStreamCreate(&st));
		for (int i = 0; i < 50; ++i) touch<<<(off / 4096 + 255) / 256, 256, 0, st>>>((char*)base, off);
		size_t n = (size_t)256 << 20; double a = now();
		CK(hipMemCreate(&hs[nh], n, &prop, 0)); CK(hipMemMap((char*)base + off, n, 0, hs[nh], 0)); CK(hipMemSetAccess((char*)base + off, n, &acc, 1));
		double b = now(); CK(hipStreamSynchronize(st)); double c = now();
		printf("map 256 MB while kernels run: %.3f ms (stream drained %.3f ms later)\n", (b - a) * 1e3, (c - b) * 1e3);
		off += n; ++nh;
	}
	double u0 = now();
	CK(hipMemUnmap(base, off)); for (int i = 0; i < nh; ++i) CK(hipMemRelease(hs[i])); CK(hipMemAddressFree(base, va));
	printf("unmap + release + free: %.3f ms\n", (now() - u0) * 1e3);
	return 0;
}
