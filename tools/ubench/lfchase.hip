// What bounds a dependent chain of random slot reads on this chip?  (scratch tool, not part of the product)
// Every OCTET (8 lanes) walks its own chain: per step it reads one random 128-byte line (16 B per lane), optionally after a
// dependent directory read (64-byte entries: two 8-byte words of one entry, or one 8-byte word of a compact array), optionally
// with a written-through 8-byte store to a random row.  The next position depends on what was read (a sum over the octet).
//   lfchase <slots MB> <dir entries> <mode> <walkers> <steps> [store]      mode 0: slot only, 1: 64-B dir entry (2 loads) + slot, 2: compact dir (1 load) + slot
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#ifndef EXTRA_OPS
#define EXTRA_OPS 0
#endif
__device__ __forceinline__ uint32_t dpp_mov141(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_movB1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_mov4E(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t oct_sum(uint32_t v) { v += dpp_mov141(v); v += dpp_movB1(v); v += dpp_mov4E(v); return v; }
__global__ void k_fill(uint32_t *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7); }
template<int MODE, bool STORE, int EXTRA>
__global__ void __launch_bounds__(256) k_walk(const uint4 *slots, uint64_t nslots, const uint64_t *dir, uint64_t ngrp, const uint64_t *cdir, uint64_t *rows, uint64_t nrows, int steps, unsigned long long *sink)
{
	const int j = threadIdx.x & 7;
	uint64_t k = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / 8 * 0x9E3779B97F4A7C15ull + 12345;
	uint32_t acc = 0;
	for (int s = 0; s < steps; ++s) {
		uint32_t si = __umulhi((uint32_t)(k >> 20), (uint32_t)nslots);
		if (MODE == 1) {
			const uint32_t g = __umulhi((uint32_t)(k >> 28), (uint32_t)ngrp);
			const uint64_t gc = dir[(uint64_t)g * 8 + (k & 3)], sm = dir[(uint64_t)g * 8 + 6];
			si = __umulhi((si + (uint32_t)sm + (uint32_t)(gc >> 40)) * 2654435761u, (uint32_t)nslots);
			acc += (uint32_t)gc;
		} else if (MODE == 2) {
			const uint32_t g = __umulhi((uint32_t)(k >> 28), (uint32_t)ngrp);
			const uint64_t sm = cdir[g];
			si = __umulhi((si + (uint32_t)sm) * 2654435761u, (uint32_t)nslots);
		}
		const uint4 v = slots[(uint64_t)si * 8 + j];
		uint32_t x = v.x + (v.y ^ v.z) + v.w;
#pragma unroll
		for (int e = 0; e < EXTRA; ++e) x = x * 3u + (x >> 7);
		const uint32_t t = oct_sum(x);
		acc += t;
		if (STORE && j == (s & 7)) __hip_atomic_store(&rows[__umulhi((uint32_t)(k >> 24), (uint32_t)nrows)], (uint64_t)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		k = k * 6364136223846793005ull + t;
	}
	if (acc == 0x1234567u) atomicAdd(sink, 1ull);
}
int main(int argc, char **argv)
{
	const size_t mb = argc > 1 ? atol(argv[1]) : 64;
	const uint64_t ngrp = argc > 2 ? atol(argv[2]) : 162000;
	const int mode = argc > 3 ? atoi(argv[3]) : 0;
	const long walkers = argc > 4 ? atol(argv[4]) : 22912;
	const int steps = argc > 5 ? atoi(argv[5]) : 400;
	const int store = argc > 6 ? atoi(argv[6]) : 0;
	const uint64_t nslots = (mb << 20) / 128, nrows = 8800000;
	uint4 *slots; uint64_t *dir, *cdir, *rows; unsigned long long *sink;
	hipMalloc(&slots, nslots * 128); hipMalloc(&dir, ngrp * 64); hipMalloc(&cdir, ngrp * 8); hipMalloc(&rows, nrows * 8); hipMalloc(&sink, 8);
	k_fill<<<(nslots * 32 + 255) / 256, 256>>>((uint32_t*)slots, nslots * 32);
	k_fill<<<(ngrp * 16 + 255) / 256, 256>>>((uint32_t*)dir, ngrp * 16);
	k_fill<<<(ngrp * 2 + 255) / 256, 256>>>((uint32_t*)cdir, ngrp * 2);
	hipDeviceSynchronize();
	const int blocks = (int)((walkers * 8 + 255) / 256);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	float best = 1e30f;
	for (int rep = 0; rep < 4; ++rep) {
		hipEventRecord(e0);
#define LAUNCH(M, S) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_walk<M, S, EXTRA_OPS>), dim3(blocks), dim3(256), 0, 0, slots, nslots, dir, ngrp, cdir, rows, nrows, steps, sink)
		if (mode == 0) { if (store) LAUNCH(0, true); else LAUNCH(0, false); }
		else if (mode == 1) { if (store) LAUNCH(1, true); else LAUNCH(1, false); }
		else { if (store) LAUNCH(2, true); else LAUNCH(2, false); }
		hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) best = ms;
	}
	printf("slots %5zu MB  dir %8llu entries (%5.1f MB; compact %4.1f MB)  mode %d  store %d  walkers %7ld (%.1f waves/SIMD)  steps %d: %8.3f ms  %6.2f G steps/s  %.0f ns/step/walker\n", mb, (unsigned long long)ngrp, ngrp * 64 / 1e6, ngrp * 8 / 1e6, mode, store,
			walkers, walkers / 8.0 / 1024.0, steps, best, (double)walkers * steps / best / 1e6, best * 1e6 / steps);
	return 0;
}
