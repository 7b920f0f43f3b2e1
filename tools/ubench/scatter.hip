// Random-access throughput microbenchmark (scratch tool): what the memory system sustains for the
// chain kernel's access pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ uint64_t rng(uint64_t &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
template<int MODE> __global__ void k(uint64_t *a, uint64_t n, int iters, uint64_t *sink)
{
	uint64_t s = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
	uint64_t acc = 0;
	const int lane = threadIdx.x & 63, j = lane & 7;
	for (int i = 0; i < iters; ++i) {
		uint64_t r = rng(s);
		if (MODE == 0) a[r % n] = r;                                                     // random 8-B plain store
		else if (MODE == 1) __hip_atomic_store(&a[r % n], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // sc1 store
		else if (MODE == 2) atomicMin((unsigned long long*)&a[r % n], (unsigned long long)r);  // no-return atomic
		else if (MODE == 3) acc += a[r % n];                                             // random 8-B load
		else if (MODE == 4) { // octet reads one 128-B line: 16 B per lane, same line for 8 lanes
			uint64_t rr = __shfl(r, lane & ~7);
			const uint4 *p = (const uint4*)a + ((rr % (n / 16)) * 8 + j);
			uint4 v = *p; acc += v.x + v.w;
		} else if (MODE == 5) { // 1 store per octet (lane 0 only), like the chain kernel's records
			if (j == 0) __hip_atomic_store(&a[r % n], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		} else if (MODE == 6) { if (j == 0) atomicMin((unsigned long long*)&a[r % n], (unsigned long long)r); }
		else if (MODE == 7) { if (j == 0) acc += a[r % n]; }
	}
	if (acc == 0x1234567) sink[0] = acc;
}
template<int MODE> void run(const char *name, uint64_t *a, uint64_t n, uint64_t *sink, double per_thread_frac)
{
	int iters = 200, blocks = 256 * 8, thr = 256;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	k<MODE><<<blocks, thr>>>(a, n, 20, sink);
	hipEventRecord(e0); k<MODE><<<blocks, thr>>>(a, n, iters, sink); hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	double ops = (double)blocks * thr * iters * per_thread_frac;
	printf("%-44s %8.2f G ops/s\n", name, ops / ms / 1e6);
}
int main()
{
	for (uint64_t mb : {64ull, 1024ull, 8192ull}) {
		uint64_t n = mb << 17; uint64_t *a, *sink; hipMalloc(&a, n * 8); hipMalloc(&sink, 8); hipMemset(a, 0xff, n * 8);
		printf("--- array %llu MB\n", (unsigned long long)mb);
		run<0>("random 8-B plain store, every lane", a, n, sink, 1);
		run<1>("random 8-B sc1 store, every lane", a, n, sink, 1);
		run<2>("random 8-B atomicMin (no return), every lane", a, n, sink, 1);
		run<3>("random 8-B load, every lane", a, n, sink, 1);
		run<4>("random 128-B line per octet (16 B/lane)", a, n, sink, 1.0 / 8);
		run<5>("random 8-B sc1 store, one lane per octet", a, n, sink, 1.0 / 8);
		run<6>("random 8-B atomicMin, one lane per octet", a, n, sink, 1.0 / 8);
		run<7>("random 8-B load, one lane per octet", a, n, sink, 1.0 / 8);
		hipFree(a); hipFree(sink);
	}
	return 0;
}
