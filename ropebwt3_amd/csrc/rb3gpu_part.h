/*
 * rb3gpu_part.h -- records in text order -> pos[] in row order by a two-pass partition (an index that lives in HBM).
 *
 * With records in text order (k_chain, trec) the validation pass read rec[sa[i]] for every row i: one random 8-byte read per
 * row, each a 64-byte line from HBM -- 302 M rows = 7.3 ms, as long as the walk itself (round 3).  The same permutation as a
 * PARTITION: the inverse suffix array is at hand (tw[t] >> 3 = the row of text position t), the rows are a permutation, so the
 * rows [b << K, (b + 1) << K) form bucket b of exactly 2^K entries, and
 *   k_part_scatter  streams rec[] and tw[] in text order and appends (row & (2^K - 1)) << (64 - K) | final position to the stream of
 *                   the row's bucket -- a tile of 8192 entries is counting-sorted by bucket in LDS first, so that a bucket
 *                   receives one contiguous chunk per tile (one atomicAdd per tile and bucket, coalesced stores);
 *   k_part_place    reads a bucket's stream and writes pos[] inside the bucket's window of 2^K rows (4 MB: it stays in the L2
 *                   of the XCD that works on the bucket -- the blocks of a bucket are dispatched to one XCD);
 * then k_pos_finalize_check_rows validates pos[] and counts the rows per window as always, now streaming.
 * Traffic ~48 B per row, all of it streams or cache-resident scatter, instead of a 64-byte line per row.
 * MEASURED (round 4, 302 M rows into a 1 G-symbol index): scatter 2.74 ms + place 3.66 ms (the windows do not stay in the L2: 1.3 TB/s)
 * + the streaming validation -- no faster than the 7.3 ms gather it was meant to replace, so it is OFF by default (rb3gpu_tune "part" 1/2
 * turns it on; tests/test_gpu_engine.py keeps it bit-exact).
 * (The fm-index.c:166-168 record rb[kb] = ka + kb itself is unchanged: this is only where it is written.)
 */
#ifndef RB3GPU_PART_H
#define RB3GPU_PART_H

#define RB3_PART_TILE 8192
#define RB3_PART_THREADS 1024
#define RB3_PART_MAXB 1024          /* buckets at most (LDS tables) */
/* a stream entry: (row inside the bucket) << (64 - K) | merged position -- K bits of row, 64 - K bits of position (K <= 22 for a
 * batch of < 2^32 rows in <= 1024 buckets: positions below 2^42 at least); the all-ones position marks a row without a final value */
#define RB3_PART_VBITS(K) (64 - (K))
#define RB3_PART_INVALID(K) ((1ull << RB3_PART_VBITS(K)) - 1ull)

/* cursor[b] = b << K before the launch.  TENT: the records may be tentative (settled through sfin here). */
template<bool TENT>
__global__ void __launch_bounds__(RB3_PART_THREADS) k_part_scatter(const int64_t *rec, const uint64_t *tw, int64_t n2, int K, int nb, const int32_t *sfin, unsigned long long *bad,
		unsigned int *cursor, uint64_t *out)
{
	__shared__ uint32_t hist[RB3_PART_MAXB], lstart[RB3_PART_MAXB], gbase[RB3_PART_MAXB];
	__shared__ uint64_t stage[RB3_PART_TILE];
	__shared__ uint16_t sb[RB3_PART_TILE];
	__shared__ uint32_t wsum[RB3_PART_THREADS / 64];
	static_assert(RB3_PART_THREADS >= RB3_PART_MAXB, "one histogram entry per thread");
	const int tid = threadIdx.x;
	constexpr int PER = RB3_PART_TILE / RB3_PART_THREADS;
	if (TENT && *(volatile unsigned long long*)&bad[2] != 0) return; // (the settle pass already knows it is incomplete)
	for (int64_t tile = (int64_t)blockIdx.x * RB3_PART_TILE; tile < n2; tile += (int64_t)gridDim.x * RB3_PART_TILE) {
		for (int b = tid; b < nb; b += RB3_PART_THREADS) hist[b] = 0u;
		__syncthreads();
		uint64_t e[PER];
		uint32_t bk[PER], rk[PER];
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const int64_t t = tile + k * RB3_PART_THREADS + tid;
			bk[k] = 0xFFFFFFFFu;
			if (t < n2) {
				const uint64_t row = tw[t] >> 3;
				int64_t v = rec[t];
				v = TENT ? pos_final(v, sfin, bad) : (v < 0 ? RB3_UNSET : v);
				const uint64_t val = (v < 0 || (uint64_t)v >= RB3_PART_INVALID(K)) ? RB3_PART_INVALID(K) : (uint64_t)v;
				const uint32_t b = (uint32_t)(row >> K);
				if (b < (uint32_t)nb) {
					bk[k] = b;
					e[k] = (row & ((1ull << K) - 1ull)) << RB3_PART_VBITS(K) | val;
					rk[k] = atomicAdd(&hist[b], 1u);
				}
			}
		}
		__syncthreads();
		{ // exclusive scan of the histogram (an entry per thread); one atomic per bucket that received something
			const uint32_t h0 = tid < nb ? hist[tid] : 0u;
			const uint32_t inc = wave_incl_scan(h0);
			if ((tid & 63) == 63) wsum[tid >> 6] = inc;
			__syncthreads();
			uint32_t base = 0;
			for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
			if (tid < nb) { lstart[tid] = base + inc - h0; if (h0) gbase[tid] = atomicAdd(&cursor[tid], h0); }
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < PER; ++k)
			if (bk[k] != 0xFFFFFFFFu) {
				const uint32_t at = lstart[bk[k]] + rk[k];
				stage[at] = e[k], sb[at] = (uint16_t)bk[k];
			}
		__syncthreads();
		const int64_t rem = n2 - tile;
		const int nt = rem < RB3_PART_TILE ? (int)rem : RB3_PART_TILE; // (every entry of the tile has a bucket: rows are < n2 = nb << K at most)
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const int i = k * RB3_PART_THREADS + tid;
			if (i < nt) {
				const uint32_t b = sb[i];
				const uint64_t o = (uint64_t)gbase[b] + (uint32_t)(i - (int)lstart[b]);
				if (o < (uint64_t)n2) out[o] = stage[i]; // (only a tw[] that is not a permutation could overflow a bucket)
			}
		}
		__syncthreads();
	}
}

/* the stream of bucket b (out[b << K .. ) into pos[]: S blocks per bucket, and all blocks of a bucket on one XCD (blocks go to the XCDs round robin) */
__global__ void __launch_bounds__(256) k_part_place(const uint64_t *out, int64_t n2, int K, int nb, int S, int64_t *pos, const unsigned long long *bad)
{
	if (*(volatile const unsigned long long*)&bad[2] != 0) return;
	const int64_t bid = blockIdx.x;
	const int64_t b = (bid / (8 * S)) * 8 + (bid & 7);
	const int s = (int)((bid >> 3) % S);
	if (b >= nb) return;
	const int64_t r0 = b << K, r1 = r0 + (1ll << K) < n2 ? r0 + (1ll << K) : n2;
	const int64_t per = ((r1 - r0) + S - 1) / S, a0 = r0 + per * s, a1 = a0 + per < r1 ? a0 + per : r1;
	for (int64_t i = a0 + threadIdx.x; i < a1; i += blockDim.x) {
		const uint64_t e = out[i];
		const uint64_t v = e & RB3_PART_INVALID(K);
		const int64_t row = r0 + (int64_t)(e >> RB3_PART_VBITS(K));
		if (row < r1) pos[row] = v == RB3_PART_INVALID(K) ? RB3_UNSET : (int64_t)v;
	}
}

__global__ void __launch_bounds__(256) k_part_init(unsigned int *cursor, int nb, int K)
{
	const int b = blockIdx.x * blockDim.x + threadIdx.x;
	if (b < nb) cursor[b] = (unsigned int)((unsigned long long)b << K);
}

#endif
