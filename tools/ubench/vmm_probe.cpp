// GPU box: which calls of the virtual-memory API does growing a mapped range take on this ROCm?  (rb3gpu.hip, vm_ensure: hipMemSetAccess on a chunk mapped BEHIND
// another one came back "invalid argument" in most cases at the scale of configs[4]; this probe tries the variants)   hipcc -O2 -o vmm_probe vmm_probe.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
static const char *es(hipError_t e) { return hipGetErrorString(e); }
int main(int argc, char **argv)
{
	int dev = 0;
	hipSetDevice(dev);
	hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
	prop.type = hipMemAllocationTypePinned, prop.location.type = hipMemLocationTypeDevice, prop.location.id = dev;
	size_t gmin = 0, grec = 0;
	hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
	hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
	printf("granularity: minimum %zu, recommended %zu\n", gmin, grec);
	const size_t G = 2u << 20;
	hipMemAccessDesc ad; memset(&ad, 0, sizeof(ad));
	ad.location.type = hipMemLocationTypeDevice, ad.location.id = dev, ad.flags = hipMemAccessFlagsProtReadWrite;
	// sizes in units of 2 MiB: first chunk, second chunk
	const size_t cases[][2] = { {339, 168}, {339, 138}, {164, 343}, {339, 306}, {507, 324}, {512, 256}, {512, 168}, {339, 512}, {100, 100}, {1024, 1024}, {645, 186}, {32, 32}, {33, 32}, {32, 33} };
	for (int mode = 0; mode < 3; ++mode) {
		if (argc > 2 && atoi(argv[2]) != mode) continue;
		printf("mode %d: %s\n", mode, mode == 0 ? "access set on the new chunk alone" : mode == 1 ? "access set on the whole mapped range from its base" : "reservation aligned to 1 GiB, access on the new chunk alone");
		for (auto &c : cases) {
			void *va = nullptr;
			const size_t a = c[0] * G, b = c[1] * G, res = (size_t)16 << 30;
			hipError_t e = hipMemAddressReserve(&va, res, mode == 2 ? ((size_t)1 << 30) : G, nullptr, 0);
			if (e != hipSuccess) { printf("  reserve: %s\n", es(e)); continue; }
			hipMemGenericAllocationHandle_t h1, h2;
			hipError_t e1 = hipMemCreate(&h1, a, &prop, 0), e2 = hipMemCreate(&h2, b, &prop, 0);
			hipError_t m1 = hipMemMap(va, a, 0, h1, 0), s1 = hipMemSetAccess(va, a, &ad, 1);
			hipError_t m2 = hipMemMap((char*)va + a, b, 0, h2, 0);
			hipError_t s2 = mode == 1 ? hipMemSetAccess(va, a + b, &ad, 1) : hipMemSetAccess((char*)va + a, b, &ad, 1);
			hipError_t w = hipSuccess;
			// (no write through the range here: in the first version of this probe a write behind a "successful" second hipMemSetAccess raised a GPU memory access fault)
			// `vmm_probe write MODE`: a kernel writes the whole range and reads it back (a "successful" hipMemSetAccess on a second chunk alone was followed by a GPU memory access fault
			// in the first version of this probe: run this with mode 1 only, or expect the process to die)
			if (argc > 2 && atoi(argv[2]) == mode && s2 == hipSuccess) {
				w = hipMemset(va, 0x5A, a + b);
				if (w == hipSuccess) w = hipDeviceSynchronize();
				unsigned char probe[3] = {0, 0, 0};
				if (w == hipSuccess) w = hipMemcpy(&probe[0], va, 1, hipMemcpyDeviceToHost);
				if (w == hipSuccess) w = hipMemcpy(&probe[1], (char*)va + a, 1, hipMemcpyDeviceToHost);
				if (w == hipSuccess) w = hipMemcpy(&probe[2], (char*)va + a + b - 1, 1, hipMemcpyDeviceToHost);
				if (w == hipSuccess && (probe[0] != 0x5A || probe[1] != 0x5A || probe[2] != 0x5A)) w = hipErrorUnknown;
			}
			printf("  va %p  %4zu + %4zu x 2 MiB: create %s/%s map1 %s access1 %s map2 %s access2 %s write %s\n", va, c[0], c[1], es(e1), es(e2), es(m1), es(s1), es(m2), es(s2), es(w)); fflush(stdout);
			(void)hipGetLastError();
			hipMemUnmap(va, a); hipMemUnmap((char*)va + a, b); hipMemRelease(h1); hipMemRelease(h2); hipMemAddressFree(va, res);
		}
	}
	return 0;
}
