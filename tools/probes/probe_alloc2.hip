// scratch probe: does hipFree wait for other streams?  cost of the stream-ordered allocator (hipMallocAsync / hipFreeAsync)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(long long cycles, int *out) { long long t0 = clock64(); while (clock64() - t0 < cycles) ; if (out) *out = 1; }
__global__ void touch(char *p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i * 4096 < n) p[i * 4096] = 1; }
int main()
{
	CK(hipSetDevice(0));
	hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
	void *p, *q; CK(hipMalloc(&p, 64 << 20)); CK(hipMalloc(&q, 64 << 20));
	spin<<<1, 64, 0, b>>>(100000000LL, nullptr); // ~50 ms at 2 GHz... (clock64 is 100 MHz on some parts)
	double t0 = now(); CK(hipFree(p)); double t1 = now(); CK(hipStreamSynchronize(b)); double t2 = now();
	printf("hipFree while another stream is busy: %.3f ms; that stream finished %.3f ms later\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
	CK(hipFree(q));
	hipMemPool_t pool; CK(hipDeviceGetDefaultMemPool(&pool, 0));
	uint64_t thr = ~0ull; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr));
	for (int rep = 0; rep < 2; ++rep)
	for (size_t mb : {16, 256, 1024, 4096}) {
		size_t n = mb << 20; void *r;
		spin<<<1, 64, 0, b>>>(20000000LL, nullptr);
		double s0 = now(); CK(hipMallocAsync(&r, n, a)); double s1 = now();
		touch<<<(n / 4096 + 255) / 256, 256, 0, a>>>((char*)r, n); CK(hipStreamSynchronize(a)); double s2 = now();
		CK(hipFreeAsync(r, a)); double s3 = now(); CK(hipStreamSynchronize(a)); double s4 = now(); CK(hipStreamSynchronize(b)); double s5 = now();
		printf("rep %d, %5zu MB: hipMallocAsync %.3f ms, touch+sync %.3f, hipFreeAsync %.3f, sync a %.3f, other stream done %.3f ms later\n", rep, mb, (s1 - s0) * 1e3, (s2 - s1) * 1e3, (s3 - s2) * 1e3, (s4 - s3) * 1e3, (s5 - s4) * 1e3);
	}
	// growth pattern: free 1 GB, allocate 1.5 GB, ...
	{
		void *r = nullptr; size_t n = (size_t)256 << 20; double g0 = now();
		CK(hipMallocAsync(&r, n, a));
		for (int i = 0; i < 6; ++i) { CK(hipFreeAsync(r, a)); n += n >> 1; CK(hipMallocAsync(&r, n, a)); touch<<<(n / 4096 + 255) / 256, 256, 0, a>>>((char*)r, n); }
		CK(hipStreamSynchronize(a));
		uint64_t resv = 0, used = 0; CK(hipMemPoolGetAttribute(pool, hipMemPoolAttrReservedMemCurrent, &resv)); CK(hipMemPoolGetAttribute(pool, hipMemPoolAttrUsedMemCurrent, &used));
		printf("six growth steps to %.0f MB: %.3f ms; pool holds %.0f MB, %.0f MB in use\n", n / 1048576.0, (now() - g0) * 1e3, resv / 1048576.0, used / 1048576.0);
		double tt = now(); CK(hipMemPoolTrimTo(pool, 0)); printf("trim: %.3f ms\n", (now() - tt) * 1e3);
		CK(hipMemPoolGetAttribute(pool, hipMemPoolAttrReservedMemCurrent, &resv)); printf("pool holds %.0f MB after the trim\n", resv / 1048576.0);
		CK(hipFreeAsync(r, a)); CK(hipStreamSynchronize(a));
	}
	return 0;
}
