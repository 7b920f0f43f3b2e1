// Pointer-chase latency microbenchmark (scratch tool, not part of the product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <algorithm>
#include <random>
__global__ void k_init(uint64_t *a, const uint64_t *perm, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) a[i] = perm[i]; }
template<int MODE> __global__ void k_chase(const uint64_t *a, size_t steps, uint64_t *out, uint64_t *cyc)
{
	uint64_t p = 0;
	uint64_t t0 = __builtin_readcyclecounter();
	for (size_t i = 0; i < steps; ++i) {
		if (MODE == 0) p = a[p];
		else if (MODE == 1) p = __hip_atomic_load(&a[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		else if (MODE == 2) p = __builtin_nontemporal_load(&a[p]);
	}
	uint64_t t1 = __builtin_readcyclecounter();
	if (threadIdx.x == 0) { out[0] = p; cyc[0] = t1 - t0; }
}
int main() {
	for (size_t mb : {size_t(0), size_t(1), size_t(4), size_t(16), size_t(64), size_t(512), size_t(2048)}) {
		size_t n = mb ? (mb << 20) / 8 : 4096;  // elements of 8 bytes; each element on its own 64B? no: dense
		size_t stride = 16;                     // one element per 128-B line
		size_t nl = n / stride;
		std::vector<uint64_t> idx(nl), perm(n, 0);
		for (size_t i = 0; i < nl; ++i) idx[i] = i;
		std::mt19937_64 rng(1); std::shuffle(idx.begin() + 1, idx.end(), rng);
		for (size_t i = 0; i < nl; ++i) perm[idx[i] * stride] = idx[(i + 1) % nl] * stride;
		uint64_t *d, *dp, *out, *cyc; hipMalloc(&d, n * 8); hipMalloc(&dp, n * 8); hipMalloc(&out, 8); hipMalloc(&cyc, 8);
		hipMemcpy(dp, perm.data(), n * 8, hipMemcpyHostToDevice);
		k_init<<<(n + 255) / 256, 256>>>(d, dp, n); hipDeviceSynchronize();
		size_t steps = 20000;
		uint64_t c[3];
		k_chase<0><<<1, 64>>>(d, steps, out, cyc); hipDeviceSynchronize();
		k_chase<0><<<1, 64>>>(d, steps, out, cyc); hipDeviceSynchronize(); hipMemcpy(&c[0], cyc, 8, hipMemcpyDeviceToHost);
		k_chase<1><<<1, 64>>>(d, steps, out, cyc); hipDeviceSynchronize(); hipMemcpy(&c[1], cyc, 8, hipMemcpyDeviceToHost);
		k_chase<2><<<1, 64>>>(d, steps, out, cyc); hipDeviceSynchronize(); hipMemcpy(&c[2], cyc, 8, hipMemcpyDeviceToHost);
		printf("array %6zu MB (lines %8zu): plain %7.1f  sc1(agent atomic) %7.1f  nontemporal %7.1f cycles/load\n", mb, nl, (double)c[0] / steps, (double)c[1] / steps, (double)c[2] / steps);
		hipFree(d); hipFree(dp); hipFree(out); hipFree(cyc);
	}
	return 0;
}
