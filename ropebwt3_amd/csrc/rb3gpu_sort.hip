/*
 * rb3gpu_sort.hip -- partial BWT of one batch on the GPU (replaces the host call
 * rb3_build_sais, sais-ss.c:10-56, for batches that fit).  gfx950 only.
 *
 * What has to come out is fixed by the reference (libsais in GSA mode, sais-ss.c:16-21, 36-41): the
 * suffixes of the batch text sorted with the i-th sentinel smaller than the (i+1)-th and every sentinel
 * smaller than A, bwt[i] = text[SA[i] - 1].  How is free: prefix doubling (Larsson-Sadakane) with
 * device radix sorts.
 *
 *   round 0   key = the first 20 symbols (3 bits each), cut after the first sentinel; sort (key, position);
 *             rank[p] = index of the first element of p's group of equal keys
 *   round r   depth h = 20 * 2^(r-1).  Only positions whose group still has several members take part
 *             (compacted list): key = rank[p] << nb | sec (nb = bits of n, so that the radix sort only has to
 *             look at 2 nb bits), where sec = rank[p + h], or the number of the
 *             string if p's sentinel lies within the first h symbols (then all members of the group are
 *             identical up to their sentinels and the sentinel order decides -- never look past it).
 *             After the sort an element's final slot is its old rank + its index inside its old group;
 *             its new rank is the old rank + the index of the first element with the same full key.
 *
 * rank[] ends up as the inverse suffix array, so the sampled inverse suffix array the merge wants for
 * its walker list is a gather.
 *
 * The sorts, scans and compactions are rocPRIM's (AMD's native device primitives, not a portability
 * layer); the kernels around them are below.  Not on the merge path: nothing in rb3gpu_kernels.h
 * depends on this file.
 */
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include <stdio.h>

#define RB3S_H0 20 /* symbols in the round-0 key: batches of short strings (reads: many equal strings, fewer doubling rounds) */
#define RB3S_H0_LONG 16 /* ... of long strings (genomes: 48 key bits are six radix passes instead of eight, and 16 symbols tell nearly all suffixes apart) */

struct rb3sort_ws {
	void *p[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	size_t cap[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
	int64_t bytes = 0;
};

static int ws_ensure(rb3sort_ws *ws, int i, size_t bytes)
{
	if (ws->cap[i] >= bytes && ws->p[i]) return 0;
	if (ws->p[i]) { (void)hipFree(ws->p[i]); ws->bytes -= (int64_t)ws->cap[i]; }
	ws->p[i] = nullptr, ws->cap[i] = 0;
	const size_t want = bytes + (bytes >> 3) + 256;
	if (hipMalloc(&ws->p[i], want) != hipSuccess) { (void)hipGetLastError(); return -1; }
	ws->cap[i] = want, ws->bytes += (int64_t)want;
	return 0;
}

rb3sort_ws *rb3sort_create(void) { return new rb3sort_ws; }

void rb3sort_destroy(rb3sort_ws *ws)
{
	if (!ws) return;
	for (int i = 0; i < 12; ++i) if (ws->p[i]) (void)hipFree(ws->p[i]);
	delete ws;
}

int64_t rb3sort_bytes(const rb3sort_ws *ws) { return ws ? ws->bytes : 0; }

/* ---- kernels ---- */

__global__ void __launch_bounds__(256) k_s_flag(const uint8_t *text, int64_t n, uint32_t *flag, unsigned long long *bad)
{
	const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n) return;
	const uint8_t c = text[p];
	flag[p] = c == 0 ? 1u : 0u;
	if (c > 5) atomicAdd(bad, 1ull);
}

/* sentpos[k] = position of the k-th sentinel (sid[p] = number of sentinels before p = string of p) */
__global__ void __launch_bounds__(256) k_s_sentpos(const uint8_t *text, int64_t n, const uint32_t *sid, uint32_t *sentpos)
{
	const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p < n && text[p] == 0) sentpos[sid[p]] = (uint32_t)p;
}

__global__ void __launch_bounds__(256) k_s_key0(const uint8_t *text, int64_t n, const uint32_t *sid, const uint32_t *sentpos, uint64_t *keys, uint32_t *vals, int h0)
{
	const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n) return;
	const int64_t d = (int64_t)sentpos[sid[p]] - p; // symbols before p's sentinel
	const int w = d + 1 < h0 ? (int)(d + 1) : h0;
	uint64_t k = 0;
	for (int t = 0; t < w; ++t) k |= (uint64_t)text[p + t] << (3 * (h0 - 1 - t));
	keys[p] = k, vals[p] = (uint32_t)p;
}

/* head[i] = i if element i starts a group of equal keys, else 0 (an inclusive max-scan makes it the index
 * of the group's first element); HI: compare the upper 32 bits only */
template<bool HI>
__global__ void __launch_bounds__(256) k_s_heads(const uint64_t *keys, int64_t n, int nb, uint32_t *head)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	bool h = i == 0;
	if (!h) h = HI ? (keys[i] >> nb) != (keys[i - 1] >> nb) : keys[i] != keys[i - 1];
	head[i] = h ? (uint32_t)i : 0u;
}

__global__ void __launch_bounds__(256) k_s_rank0(const uint32_t *vals, const uint32_t *hidx, int64_t n, uint32_t *rank, uint32_t *sa, uint8_t *unres)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t p = vals[i];
	rank[p] = hidx[i], sa[i] = p;
	const bool single = hidx[i] == (uint32_t)i && (i + 1 == n || hidx[i + 1] == (uint32_t)(i + 1));
	unres[i] = single ? 0 : 1;
}

__global__ void __launch_bounds__(256) k_s_key(const uint32_t *list, int64_t nu, const uint32_t *rank, const uint32_t *sid, const uint32_t *sentpos, int64_t h, int nb,
		uint64_t *keys, uint32_t *vals)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nu) return;
	const uint32_t p = list[i];
	const uint32_t s = sid[p];
	const int64_t d = (int64_t)sentpos[s] - (int64_t)p;
	const uint32_t sec = d < h ? s : rank[p + h];
	keys[i] = (uint64_t)rank[p] << nb | sec, vals[i] = p;
}

/* gh / sh: index (in the sorted list) of the first element of the old group / of the new sub-group */
__global__ void __launch_bounds__(256) k_s_update(const uint64_t *keys, const uint32_t *vals, const uint32_t *gh, const uint32_t *sh, int64_t nu, int nb,
		uint32_t *rank, uint32_t *sa, uint8_t *unres)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nu) return;
	const uint32_t r0 = (uint32_t)(keys[i] >> nb), p = vals[i];
	sa[r0 + ((uint32_t)i - gh[i])] = p;
	rank[p] = r0 + (sh[i] - gh[i]);
	const bool single = sh[i] == (uint32_t)i && (i + 1 == nu || sh[i + 1] == (uint32_t)(i + 1));
	unres[i] = single ? 0 : 1;
}

__global__ void __launch_bounds__(256) k_s_bwt(const uint32_t *sa, const uint8_t *text, int64_t n, uint8_t *bwt)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const uint32_t p = sa[i];
	bwt[i] = p ? text[p - 1] : text[n - 1];
}

__global__ void __launch_bounds__(256) k_s_ckrow(const uint32_t *rank, int64_t n, int64_t step, int64_t nck, int64_t *ckrow)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nck) ckrow[i] = rank[i * step];
}

/* text-order words for the merge's walkers (k_chain, TEXT): row of the suffix at p << 3 | the symbol before it */
__global__ void __launch_bounds__(256) k_s_tw(const uint32_t *rank, const uint8_t *text, int64_t n, uint64_t *tw)
{
	const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p < n) tw[p] = (uint64_t)rank[p] << 3 | (p ? text[p - 1] : 0);
}

/* ---- driver ---- */

#define S_HIP(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); return -2; } } while (0)
#define S_GRID(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, st

enum { W_KEYA, W_KEYB, W_VALA, W_VALB, W_RANK, W_SID, W_SA, W_SENT, W_T0, W_T1, W_FLAG, W_TMP };

/* d_text: n symbols (0..5, last one 0) in device memory; d_bwt: n bytes out; d_ckrow: ceil(n/step) rows out or NULL;
 * d_tw: n text-order words out or NULL.
 * Returns 0, -1 (out of memory), -2 (HIP error), -3 (bad text), and the number of doubling rounds in *rounds. */
int rb3sort_bwt(rb3sort_ws *ws, hipStream_t st, int64_t n, const uint8_t *d_text, uint8_t *d_bwt, int64_t step, int64_t *d_ckrow, int *rounds, uint64_t *d_tw)
{
	if (!ws || n <= 0 || n >= (1LL << 31)) return -3;
	if (ws_ensure(ws, W_KEYA, (size_t)n * 8) || ws_ensure(ws, W_KEYB, (size_t)n * 8) || ws_ensure(ws, W_VALA, (size_t)n * 4) || ws_ensure(ws, W_VALB, (size_t)n * 4) ||
		ws_ensure(ws, W_RANK, (size_t)n * 4) || ws_ensure(ws, W_SID, (size_t)n * 4) || ws_ensure(ws, W_SA, (size_t)n * 4) || ws_ensure(ws, W_SENT, (size_t)n * 4) ||
		ws_ensure(ws, W_T0, (size_t)n * 4) || ws_ensure(ws, W_T1, (size_t)n * 4) || ws_ensure(ws, W_FLAG, (size_t)n + 64)) return -1;
	uint64_t *keyA = (uint64_t*)ws->p[W_KEYA], *keyB = (uint64_t*)ws->p[W_KEYB];
	uint32_t *valA = (uint32_t*)ws->p[W_VALA], *valB = (uint32_t*)ws->p[W_VALB], *rank = (uint32_t*)ws->p[W_RANK], *sid = (uint32_t*)ws->p[W_SID];
	uint32_t *sa = (uint32_t*)ws->p[W_SA], *sentpos = (uint32_t*)ws->p[W_SENT], *t0 = (uint32_t*)ws->p[W_T0], *t1 = (uint32_t*)ws->p[W_T1];
	uint8_t *unres = (uint8_t*)ws->p[W_FLAG];
	unsigned long long *dcnt = (unsigned long long*)(unres + ((n + 15) & ~15LL)); // two counters behind the flags
	// temporary storage of the library calls: the largest request at this n
	size_t tb = 0, b;
	S_HIP(rocprim::radix_sort_pairs(nullptr, b, keyA, keyB, valA, valB, (size_t)n, 0, 64, st)); tb = std::max(tb, b);
	S_HIP(rocprim::exclusive_scan(nullptr, b, t0, sid, 0u, (size_t)n, rocprim::plus<uint32_t>(), st)); tb = std::max(tb, b);
	S_HIP(rocprim::inclusive_scan(nullptr, b, t0, t1, (size_t)n, rocprim::maximum<uint32_t>(), st)); tb = std::max(tb, b);
	S_HIP(rocprim::select(nullptr, b, valA, unres, valB, (size_t*)dcnt, (size_t)n, st)); tb = std::max(tb, b);
	if (ws_ensure(ws, W_TMP, tb + 256)) return -1;
	void *tmp = ws->p[W_TMP];
	size_t tmp_bytes = ws->cap[W_TMP];

	// strings: sid[p] = number of sentinels before p, sentpos[k] = position of the k-th sentinel
	S_HIP(hipMemsetAsync(dcnt, 0, 16, st));
	hipLaunchKernelGGL(k_s_flag, S_GRID(n), d_text, n, t0, dcnt + 1);
	b = tmp_bytes; S_HIP(rocprim::exclusive_scan(tmp, b, t0, sid, 0u, (size_t)n, rocprim::plus<uint32_t>(), st));
	hipLaunchKernelGGL(k_s_sentpos, S_GRID(n), d_text, n, (const uint32_t*)sid, sentpos);
	uint32_t nsent = 0; // sentinels before the last symbol
	{
		unsigned long long bad = 0;
		uint8_t last = 1;
		S_HIP(hipMemcpyAsync(&nsent, sid + n - 1, 4, hipMemcpyDeviceToHost, st));
		S_HIP(hipMemcpyAsync(&bad, dcnt + 1, 8, hipMemcpyDeviceToHost, st));
		S_HIP(hipMemcpyAsync(&last, d_text + n - 1, 1, hipMemcpyDeviceToHost, st));
		S_HIP(hipStreamSynchronize(st));
		if (bad != 0 || last != 0) return -3;
	}
	// round 0
	const int h0 = n / ((int64_t)nsent + 1) > 4096 ? RB3S_H0_LONG : RB3S_H0; // (the last symbol is a sentinel: nsent + 1 strings)
	hipLaunchKernelGGL(k_s_key0, S_GRID(n), d_text, n, (const uint32_t*)sid, (const uint32_t*)sentpos, keyA, valA, h0);
	b = tmp_bytes; S_HIP(rocprim::radix_sort_pairs(tmp, b, keyA, keyB, valA, valB, (size_t)n, 0, 3 * h0, st));
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s_heads<false>), S_GRID(n), (const uint64_t*)keyB, n, 0, t0);
	b = tmp_bytes; S_HIP(rocprim::inclusive_scan(tmp, b, t0, t1, (size_t)n, rocprim::maximum<uint32_t>(), st));
	hipLaunchKernelGGL(k_s_rank0, S_GRID(n), (const uint32_t*)valB, (const uint32_t*)t1, n, rank, sa, unres);
	b = tmp_bytes; S_HIP(rocprim::select(tmp, b, valB, unres, valA, (size_t*)dcnt, (size_t)n, st));
	size_t nu = 0;
	S_HIP(hipMemcpyAsync(&nu, dcnt, 8, hipMemcpyDeviceToHost, st));
	S_HIP(hipStreamSynchronize(st));
	int nr = 0, nb = 1;
	while ((1LL << nb) < n) ++nb; // ranks and string numbers are below n: 2 nb key bits
	uint32_t *list = valA, *other = valB; // the compacted positions live in one of the two value buffers
	// (rocprim::segmented_radix_sort_pairs over the groups of the list -- every group sorted by the second rank alone -- was
	// measured instead of the device-wide sort of (rank, second rank) pairs: 2.2x slower on reads (millions of groups of ~30),
	// 1.1x slower on genomes; a wave-per-group bitonic sort of its own would be the thing to write)
	for (int64_t h = h0; nu > 0; h <<= 1) {
		if (++nr > 40) return -3; // depth 20 * 2^40: cannot happen for a text that ends with a sentinel
		hipLaunchKernelGGL(k_s_key, S_GRID(nu), (const uint32_t*)list, (int64_t)nu, (const uint32_t*)rank, (const uint32_t*)sid, (const uint32_t*)sentpos, h, nb, keyA, other);
		// sorted (keys, positions) -> keyB, list (the old list is free now: its positions were copied into `other`)
		b = tmp_bytes; S_HIP(rocprim::radix_sort_pairs(tmp, b, keyA, keyB, other, list, nu, 0, 2 * nb, st));
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s_heads<true>), S_GRID(nu), (const uint64_t*)keyB, (int64_t)nu, nb, t0);
		b = tmp_bytes; S_HIP(rocprim::inclusive_scan(tmp, b, t0, t1, nu, rocprim::maximum<uint32_t>(), st));
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_s_heads<false>), S_GRID(nu), (const uint64_t*)keyB, (int64_t)nu, nb, t0);
		uint32_t *sh = (uint32_t*)keyA; // the unsorted keys are not needed any more: sub-group heads go there
		b = tmp_bytes; S_HIP(rocprim::inclusive_scan(tmp, b, t0, sh, nu, rocprim::maximum<uint32_t>(), st));
		hipLaunchKernelGGL(k_s_update, S_GRID(nu), (const uint64_t*)keyB, (const uint32_t*)list, (const uint32_t*)t1, (const uint32_t*)sh, (int64_t)nu, nb, rank, sa, unres);
		b = tmp_bytes; S_HIP(rocprim::select(tmp, b, list, unres, other, (size_t*)dcnt, nu, st));
		S_HIP(hipMemcpyAsync(&nu, dcnt, 8, hipMemcpyDeviceToHost, st));
		S_HIP(hipStreamSynchronize(st));
		std::swap(list, other);
	}
	hipLaunchKernelGGL(k_s_bwt, S_GRID(n), (const uint32_t*)sa, d_text, n, d_bwt);
	if (d_ckrow && step > 0) {
		const int64_t nck = (n + step - 1) / step;
		hipLaunchKernelGGL(k_s_ckrow, S_GRID(nck), (const uint32_t*)rank, n, step, nck, d_ckrow);
	}
	if (d_tw) hipLaunchKernelGGL(k_s_tw, S_GRID(n), (const uint32_t*)rank, d_text, n, d_tw);
	S_HIP(hipStreamSynchronize(st));
	if (rounds) *rounds = nr;
	return 0;
}
