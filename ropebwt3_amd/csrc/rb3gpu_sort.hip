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
	void *p[18] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
	size_t cap[18] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
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
	for (int i = 0; i < 18; ++i) if (ws->p[i]) (void)hipFree(ws->p[i]);
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

/* ---- doubling rounds, group by group ----
 * The unresolved suffixes leave every round in rank order, i.e. grouped by the rank they share, and a round only has to order
 * every group by the rank of the suffix h positions further on.  Round 1 of this engine sorted (rank, second rank) pairs with
 * one device-wide radix sort -- eight passes over all unresolved suffixes per round, the bulk of the sorting time for reads
 * (coverage 30: every group has ~30 members until the reads end) and for batches of similar genomes.  Groups are small, so:
 * wave c takes the groups that START in positions [64c, 64c + 64) of the list; together they span fewer than 128 positions
 * unless the last one runs past the next 64 (then it is a LARGE group: left to the device-wide sort, below), and the wave
 * sorts the span with a bitonic network, two elements per lane, on keys  start of the group (7 bits) | second rank | slot
 * (the slot makes the keys distinct and brings the payload back with one gather).  No LDS, no barrier. */
__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int m)
{
	return (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)v, m) | (uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), m) << 32;
}

__global__ void __launch_bounds__(256) k_s_key2(const uint32_t *list, int64_t nu, const uint32_t *rank, const uint32_t *sid, const uint32_t *sentpos, int64_t h,
		uint32_t *sec, uint32_t *r0, uint32_t *vals, uint32_t *headidx)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nu) return;
	const uint32_t p = list[i];
	const uint32_t s = sid[p], r = rank[p];
	const int64_t d = (int64_t)sentpos[s] - (int64_t)p;
	sec[i] = d < h ? s : rank[p + h];
	r0[i] = r, vals[i] = p;
	headidx[i] = (i == 0 || rank[list[i - 1]] != r) ? (uint32_t)i : 0u;
}

__global__ void __launch_bounds__(256) k_s_wsort(const uint32_t *sec_in, const uint32_t *vals_in, const uint32_t *gh, int64_t nu, uint32_t *sec_out, uint32_t *vals_out, uint8_t *lflag)
{
	const int lane = threadIdx.x & 63;
	const int64_t base = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
	if (base >= nu) return;
	const int64_t i0 = base + lane, i1 = base + 64 + lane;
	const bool real0 = i0 < nu && gh[i0] == (uint32_t)i0;          // a group starts here
	const bool head1 = i1 >= nu || gh[i1] == (uint32_t)i1;         // ... or the list ends (a virtual head)
	const uint64_t Rc = __ballot(real0), Vc = __ballot(i0 >= nu), Hn = __ballot(head1);
	if (Rc == 0) return; // the whole chunk continues a group that started further up
	const int f = __ffsll((unsigned long long)Rc) - 1, sl = 63 - __clzll((long long)Rc);
	int64_t E; // end of the span: end of the last group that starts in this chunk
	if (Vc) E = nu;
	else if (Hn) E = base + 64 + (__ffsll((unsigned long long)Hn) - 1);
	else { // no head in the next 64 positions either: a large group
		E = base + sl;
		if (lane == 0) lflag[base + sl] = 1;
	}
	const int64_t S = base + f;
	const int L = (int)(E - S);
	if (L <= 0) return;
	uint64_t k[2];
	uint32_t pv[2];
#pragma unroll
	for (int r = 0; r < 2; ++r) {
		const int e = r * 64 + lane;
		k[r] = ~0ull, pv[r] = 0;
		if (e < L) {
			const int64_t pos = S + e;
			k[r] = (uint64_t)(gh[pos] - (uint32_t)S) << 39 | (uint64_t)sec_in[pos] << 7 | (uint64_t)e;
			pv[r] = vals_in[pos];
		}
	}
	if (L > 1) {
#pragma unroll
		for (int kk = 2; kk <= 128; kk <<= 1) {
#pragma unroll
			for (int j = kk >> 1; j > 0; j >>= 1) {
				if (j == 64) { // partner in the other register (kk == 128: ascending)
					if (k[0] > k[1]) { const uint64_t t = k[0]; k[0] = k[1], k[1] = t; }
				} else {
#pragma unroll
					for (int r = 0; r < 2; ++r) {
						const int e = r * 64 + lane;
						const uint64_t o = shfl_xor64(k[r], j);
						const bool asc = (e & kk) == 0, lower = (lane & j) == 0;
						const bool take_min = asc == lower;
						k[r] = take_min ? (k[r] < o ? k[r] : o) : (k[r] > o ? k[r] : o);
					}
				}
			}
		}
	}
#pragma unroll
	for (int r = 0; r < 2; ++r) {
		const int e = r * 64 + lane;
		const int src = (int)(k[r] & 127u); // (meaningless for padding keys; not used then)
		const uint32_t a = (uint32_t)__shfl((int)pv[0], src & 63), b = (uint32_t)__shfl((int)pv[1], src & 63);
		if (e < L) {
			const int64_t pos = S + e;
			sec_out[pos] = (uint32_t)(k[r] >> 7), vals_out[pos] = (src & 64) ? b : a;
		}
	}
}

/* members of the large groups: flagged for the device-wide sort */
__global__ void __launch_bounds__(256) k_s_mark(const uint32_t *gh, const uint8_t *lflag, int64_t nu, uint8_t *flag)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nu) flag[i] = lflag[gh[i]];
}

__global__ void __launch_bounds__(256) k_s_lgather(const uint32_t *lpos, int64_t nl, const uint32_t *r0, const uint32_t *sec, const uint32_t *vals, int nb, uint64_t *keys, uint32_t *ov)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nl) return;
	const uint32_t pos = lpos[i];
	keys[i] = (uint64_t)r0[pos] << nb | sec[pos], ov[i] = vals[pos];
}

__global__ void __launch_bounds__(256) k_s_lscatter(const uint32_t *lpos, int64_t nl, const uint64_t *keys, const uint32_t *iv, int nb, uint32_t *sec_out, uint32_t *vals_out)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nl) return;
	const uint32_t pos = lpos[i]; // (positions ascending, keys ascending by (rank, second rank): the groups come back in place)
	sec_out[pos] = (uint32_t)(keys[i] & ((1ull << nb) - 1ull)), vals_out[pos] = iv[i];
}

/* sub-group heads after the sort inside the groups: a group starts, or the second rank changes */
__global__ void __launch_bounds__(256) k_s_heads2(const uint32_t *sec, const uint32_t *gh, int64_t nu, uint32_t *head)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nu) return;
	const bool h = gh[i] == (uint32_t)i || sec[i] != sec[i - 1];
	head[i] = h ? (uint32_t)i : 0u;
}

__global__ void __launch_bounds__(256) k_s_update2(const uint32_t *r0s, const uint32_t *vals, const uint32_t *gh, const uint32_t *sh, int64_t nu,
		uint32_t *rank, uint32_t *sa, uint8_t *unres)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nu) return;
	const uint32_t r0 = r0s[i], p = vals[i]; // (r0 is the same for the whole group: still in place after the sort inside the group)
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

enum { W_KEYA, W_KEYB, W_VALA, W_VALB, W_RANK, W_SID, W_SA, W_SENT, W_T0, W_T1, W_FLAG, W_TMP, W_LFLAG, W_LKA, W_LKB, W_LVA, W_LVB, W_SPARE };

/* d_text: n symbols (0..5, last one 0) in device memory; d_bwt: n bytes out; d_ckrow: ceil(n/step) rows out or NULL;
 * d_tw: n text-order words out or NULL; d_sa: the suffix array (n x u32: text position of the suffix of every row) out or NULL.
 * Returns 0, -1 (out of memory), -2 (HIP error), -3 (bad text), and the number of doubling rounds in *rounds. */
int rb3sort_bwt(rb3sort_ws *ws, hipStream_t st, int64_t n, const uint8_t *d_text, uint8_t *d_bwt, int64_t step, int64_t *d_ckrow, int *rounds, uint64_t *d_tw, uint32_t *d_sa)
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
	// (rocprim::segmented_radix_sort_pairs over the groups was measured too: 2.2x slower than the device-wide sort on reads;
	// carrying the ranks by list position through the compaction -- a zip-iterator select instead of two gathers of rank[] per
	// element -- was 7 % slower on reads and 12 % on genomes)
	if (ws_ensure(ws, W_LFLAG, (size_t)n + 64)) return -1;
	uint8_t *lflag = (uint8_t*)ws->p[W_LFLAG];
	uint32_t *sec_in = (uint32_t*)keyA, *sec_out = (uint32_t*)keyA + n, *r0s = (uint32_t*)keyB, *lpos = (uint32_t*)keyB + n; // (the 64-bit key buffers, as halves)
	for (int64_t h = h0; nu > 0; h <<= 1) {
		if (++nr > 40) return -3; // depth 20 * 2^40: cannot happen for a text that ends with a sentinel
		hipLaunchKernelGGL(k_s_key2, S_GRID(nu), (const uint32_t*)list, (int64_t)nu, (const uint32_t*)rank, (const uint32_t*)sid, (const uint32_t*)sentpos, h, sec_in, r0s, other, t0);
		b = tmp_bytes; S_HIP(rocprim::inclusive_scan(tmp, b, t0, t1, nu, rocprim::maximum<uint32_t>(), st)); // t1 = gh: first index of the element's group
		S_HIP(hipMemsetAsync(lflag, 0, nu, st));
		// every group sorted by the second rank: (sec_in, other) -> (sec_out, list)
		hipLaunchKernelGGL(k_s_wsort, dim3((unsigned)(((nu + 63) / 64 + 3) / 4)), dim3(256), 0, st, (const uint32_t*)sec_in, (const uint32_t*)other, (const uint32_t*)t1, (int64_t)nu, sec_out, list, lflag);
		hipLaunchKernelGGL(k_s_mark, S_GRID(nu), (const uint32_t*)t1, (const uint8_t*)lflag, (int64_t)nu, unres);
		b = tmp_bytes; S_HIP(rocprim::select(tmp, b, rocprim::counting_iterator<uint32_t>(0), unres, lpos, (size_t*)dcnt, nu, st));
		size_t nl = 0;
		S_HIP(hipMemcpyAsync(&nl, dcnt, 8, hipMemcpyDeviceToHost, st));
		S_HIP(hipStreamSynchronize(st));
		if (nl > 0) { // the members of the large groups: one device-wide sort of (rank, second rank), back into their places
			if (ws_ensure(ws, W_LKA, nl * 8) || ws_ensure(ws, W_LKB, nl * 8) || ws_ensure(ws, W_LVA, nl * 4) || ws_ensure(ws, W_LVB, nl * 4)) return -1;
			uint64_t *lka = (uint64_t*)ws->p[W_LKA], *lkb = (uint64_t*)ws->p[W_LKB];
			uint32_t *lva = (uint32_t*)ws->p[W_LVA], *lvb = (uint32_t*)ws->p[W_LVB];
			S_HIP(rocprim::radix_sort_pairs(nullptr, b, lka, lkb, lva, lvb, nl, 0, 2 * nb, st));
			if (b > tmp_bytes) { if (ws_ensure(ws, W_TMP, b + 256)) return -1; tmp = ws->p[W_TMP], tmp_bytes = ws->cap[W_TMP]; }
			hipLaunchKernelGGL(k_s_lgather, S_GRID(nl), (const uint32_t*)lpos, (int64_t)nl, (const uint32_t*)r0s, (const uint32_t*)sec_in, (const uint32_t*)other, nb, lka, lva);
			b = tmp_bytes; S_HIP(rocprim::radix_sort_pairs(tmp, b, lka, lkb, lva, lvb, nl, 0, 2 * nb, st));
			hipLaunchKernelGGL(k_s_lscatter, S_GRID(nl), (const uint32_t*)lpos, (int64_t)nl, (const uint64_t*)lkb, (const uint32_t*)lvb, nb, sec_out, list);
		}
		hipLaunchKernelGGL(k_s_heads2, S_GRID(nu), (const uint32_t*)sec_out, (const uint32_t*)t1, (int64_t)nu, t0);
		uint32_t *sh = sec_in; // the unsorted second ranks are not needed any more: sub-group heads go there
		b = tmp_bytes; S_HIP(rocprim::inclusive_scan(tmp, b, t0, sh, nu, rocprim::maximum<uint32_t>(), st));
		hipLaunchKernelGGL(k_s_update2, S_GRID(nu), (const uint32_t*)r0s, (const uint32_t*)list, (const uint32_t*)t1, (const uint32_t*)sh, (int64_t)nu, rank, sa, unres);
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
	if (d_sa) S_HIP(hipMemcpyAsync(d_sa, sa, (size_t)n * 4, hipMemcpyDeviceToDevice, st)); // the suffix array itself (row -> text position)
	S_HIP(hipStreamSynchronize(st));
	if (rounds) *rounds = nr;
	return 0;
}
