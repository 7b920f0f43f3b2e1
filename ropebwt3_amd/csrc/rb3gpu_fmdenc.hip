/*
 * rb3gpu_fmdenc.hip -- the FMD word stream (rld0.c:107-216) packed on the GPU.  gfx950 only.
 *
 * Input: the maximal runs of the BWT as start << 3 | sym, resident in HBM (k_export_runs).  Output: the
 * 64-bit words rld_enc/rld_enc_finish would have produced (blocks of 8 words: a header with the symbol
 * counts of the block before, then Elias-delta codes, rld0.c:45-51, 137-151), ready for the rank index and
 * rld_dump on the host.
 *
 * The reference packs greedily and sequentially: a code goes into the current block unless it would reach
 * the end of the block's last usable word (rld0.c:142).  In terms of the prefix sums P of the code widths
 * that is: the block starting at run i holds the runs j with P[j+1] - P[i] < C (C = payload bits of the
 * block), so where a block ends only depends on where it starts -- a chain i -> next(i), at most 97 runs
 * ahead.  Chains started at different runs do NOT fall into step quickly (their offset in bits only drifts
 * by a few bits per block), so the chain is found exactly: the runs are cut into chunks of 4096; the first
 * block start in a chunk is one of its first 128 runs, and for each of those 128 possible entries a thread
 * follows the chain through the chunk and notes where it leaves (offset into the next chunk) and how many
 * blocks it passed.  The host then walks these small tables chunk by chunk -- one lookup each -- to get
 * every chunk's true entry and first block index, and a second kernel writes the block starts.  The last
 * block of a 2^20-block superblock has one payload word less (rld0.h:81): the walk stops there, that one
 * block is ended on the device, and the walk resumes behind it (the tables do not depend on it).
 * Then one thread per block packs its runs.
 *
 * Block headers are 16-bit (fewer than 0x4000 symbols in the block before, rld0.c:116-128: 2 header words, 384 payload
 * bits) or 32-bit (fewer than 2^30: 4 header words, 256 payload bits) -- the second kind is what a compressible index such
 * as many genomes of one species consists of.  The width of a block's header follows from the symbols of the block before,
 * so the state of the chain is (first run, header type) and the chunk tables have an entry per (offset, type).  A block
 * whose predecessor holds 2^30 symbols or more (64-bit header, one payload word) returns 1 and the caller packs on the host.
 */
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <time.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>
#include <stdio.h>

#define FE_C0        384          /* payload bits of a block with a 16-bit header: 6 words */
#define FE_C1        256          /* ... with a 32-bit header: 4 words */
#define FE_CBITS(t)  ((t) ? FE_C1 : FE_C0)
#define FE_WIDE      0x40000000ull /* symbols in a block from which the next header would be 64-bit: not produced here */
#define FE_SB_BLOCKS (1LL << 20)  /* blocks per superblock (2^23 words) */
#define FE_CHUNK     4096         /* runs per speculation chunk */

#define FE_HIP(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); ret = -2; goto done; } } while (0)
#define FE_GRID(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, st /* (a thread per item: for fewer than 2^32 items) */
#define FE_GRID_LOOP(n) dim3((unsigned)((((n) + 255) / 256) < (1 << 22) ? (((n) + 255) / 256) : (1 << 22))), dim3(256), 0, st /* kernels with a grid-stride loop */

__device__ __forceinline__ int fe_ilog2(uint64_t v) { return 63 - __clzll((long long)v); }

struct fe_widen { __device__ __host__ uint64_t operator()(uint8_t v) const { return (uint64_t)v; } }; // the scan accumulates in 64 bits

/* width of the code of every run (rld0.c:45-51, 140-141); flag[0] |= 1 if a code would not fit 63 bits */
__global__ void __launch_bounds__(256) k_fe_width(const uint64_t *words, int64_t nr, int64_t n_sym, uint8_t *width, unsigned int *flag)
{
	// (a grid-stride loop: a launch holds fewer than 2^32 threads, and an index of 24.8 G symbols in long contigs has 4.99 G runs -- with a thread per run
	// the launch came out 2^32 threads short, the widths behind run 696 M stayed what the fresh buffer held, and the chain ran away: round 6)
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= nr; i += (int64_t)gridDim.x * blockDim.x) {
		if (i == nr) { width[i] = 0; continue; }
		const int64_t s = (int64_t)(words[i] >> 3), e = i + 1 < nr ? (int64_t)(words[i + 1] >> 3) : n_sym;
		const uint64_t l = (uint64_t)(e - s);
		const int y = fe_ilog2(l), zz = fe_ilog2((uint64_t)y + 1);
		const int w = (zz << 1) + 1 + y + 3;
		if (w >= 64 || e <= s) atomicOr(flag, 1u);
		width[i] = (uint8_t)w;
	}
}

/* first run of the block after the one that starts at run i (payload of C bits) */
__device__ __forceinline__ int64_t fe_next(const uint64_t *P, int64_t nr, int64_t i, int C)
{
	const uint64_t lim = P[i] + (uint64_t)C;
	int64_t lo = i + 1, hi = i + 1 + C / 4 < nr ? i + 1 + C / 4 : nr; // a code has at least 4 bits
	if (P[hi] < lim) return nr; // everything that is left fits
	while (lo < hi) { // smallest t in (i, nr] with P[t] >= lim
		const int64_t mid = (lo + hi) >> 1;
		if (P[mid] >= lim) hi = mid; else lo = mid + 1;
	}
	return lo - 1;
}

#define FE_ENTRIES 128            /* possible entry offsets into a chunk (a block spans at most 97 runs) */

/* start of run i (S[nr] = n_sym) */
__device__ __forceinline__ uint64_t fe_start(const uint64_t *words, int64_t nr, int64_t n_sym, int64_t i)
{
	return i < nr ? words[i] >> 3 : (uint64_t)n_sym;
}

/* one block of the chain: (i, t) -> (first run of the next block, its header type); flag |= 8 if that header would be 64-bit */
__device__ __forceinline__ int64_t fe_step(const uint64_t *P, const uint64_t *words, int64_t nr, int64_t n_sym, int64_t i, int *t, int cut, unsigned int *flag)
{
	const int64_t nx = fe_next(P, nr, i, FE_CBITS(*t) - cut);
	const uint64_t tot = fe_start(words, nr, n_sym, nx) - fe_start(words, nr, n_sym, i);
	if (tot >= FE_WIDE) atomicOr(flag, 8u);
	*t = tot >= 0x4000 ? 1 : 0;
	return nx;
}

/* for chunk k, entry offset d and entry header type t: follow the chain from run k R + d to the first block start at or
 * behind the end of the chunk; E = that start's offset into the next chunk | its type << 7, Cn = blocks passed */
__global__ void __launch_bounds__(256) k_fe_table(int64_t K, int64_t nr, int64_t n_sym, const uint64_t *P, const uint64_t *words, uint8_t *E, uint16_t *Cn, unsigned int *flag)
{
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= K * FE_ENTRIES * 2) return;
	const int64_t k = (t >> 1) / FE_ENTRIES, end = (k + 1) * FE_CHUNK < nr ? (k + 1) * FE_CHUNK : nr;
	int64_t i = k * FE_CHUNK + ((t >> 1) % FE_ENTRIES), c = 0;
	int ty = (int)(t & 1);
	unsigned int dummy = 0; // a speculative entry that is never taken must not raise the flag: the emit pass raises it for real
	(void)flag;
	while (i < end) i = fe_step(P, words, nr, n_sym, i, &ty, 0, &dummy), ++c;
	E[t] = (uint8_t)((i - end) | ty << 7), Cn[t] = (uint16_t)c;
}

/* the same from one given state (the first chunk of a superblock is entered anywhere): out = { exit, blocks, exit type } */
__global__ void k_fe_chain1(int64_t i, int ty, int64_t end, int64_t nr, int64_t n_sym, const uint64_t *P, const uint64_t *words, int64_t *out)
{
	int64_t c = 0;
	unsigned int dummy = 0;
	while (i < end) i = fe_step(P, words, nr, n_sym, i, &ty, 0, &dummy), ++c;
	out[0] = i, out[1] = c, out[2] = ty;
}

/* block starts of the listed chunks: chunk ids[j] is entered at run entry[j] >> 1 with header type entry[j] & 1, whose block
 * has the global index base[j]; only indices up to `last` are written */
__global__ void __launch_bounds__(256) k_fe_emit(int64_t nlist, const int64_t *ids, const int64_t *entry, const int64_t *base, int64_t nr, int64_t n_sym, const uint64_t *P, const uint64_t *words,
		int64_t last, int64_t *bs, unsigned int *flag)
{
	const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= nlist) return;
	const int64_t k = ids[j], end = (k + 1) * FE_CHUNK < nr ? (k + 1) * FE_CHUNK : nr;
	int64_t i = entry[j] >> 1, b = base[j];
	int ty = (int)(entry[j] & 1);
	while (i < end && b <= last) {
		bs[b] = i, ++b;
		if (b > last) break; // the superblock's last block is ended by k_fe_special (one payload word less)
		i = fe_step(P, words, nr, n_sym, i, &ty, 0, flag);
	}
}

/* the superblock's last block: it starts at bs[gb] and has one payload word less; out[0] = first run behind it, out[1] =
 * the header type of the block that starts there */
__global__ void k_fe_special(const uint64_t *P, const uint64_t *words, int64_t nr, int64_t n_sym, const int64_t *bs, int64_t gb, int64_t *out, unsigned int *flag)
{
	const uint64_t before = fe_start(words, nr, n_sym, bs[gb]) - fe_start(words, nr, n_sym, bs[gb - 1]); // gb >= 1: it is the LAST block of a superblock
	int ty = before >= 0x4000 ? 1 : 0;
	out[0] = fe_step(P, words, nr, n_sym, bs[gb], &ty, 64, flag);
	out[1] = ty;
}

/* one thread per block: thread t < nfull packs block b0 + t (its header from the runs of the block before it: bs[b - 1] .. bs[b], none for the very
 * first block of the stream -- two equal starts), thread nfull -- if `trailing` -- writes the header-only block behind the data (rld0.c:206-216).
 * Block numbers are those of the PIECE at hand (rb3fmd_enc_piece): out[8 t ..] is the t-th block written. */
__global__ void __launch_bounds__(256) k_fe_pack(const uint64_t *words, int64_t nr, int64_t n_sym, const int64_t *bs, int64_t b0, int64_t nfull, int trailing, uint64_t *out, unsigned int *flag, unsigned int *tail)
{
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t > nfull || (t == nfull && !trailing)) return;
	const int64_t b = b0 + t;
	uint64_t z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	int type = 0;
	{ // header: what the block before contained (rld0.c:116-128), 7 x uint16 or 7 x uint32
		uint64_t c[7] = {0, 0, 0, 0, 0, 0, 0};
		for (int64_t i = bs[b - 1]; i < bs[b]; ++i) {
			const int64_t s = (int64_t)(words[i] >> 3), e = i + 1 < nr ? (int64_t)(words[i + 1] >> 3) : n_sym;
			const uint64_t l = (uint64_t)(e - s);
			c[0] += l, c[1 + (int)(words[i] & 7)] += l;
		}
		if (c[0] >= FE_WIDE) { atomicOr(flag, 2u); return; } // needs a 64-bit header: not produced here
		if (c[0] < 0x4000) {
			z[0] = c[0] | c[1] << 16 | c[2] << 32 | c[3] << 48;
			z[1] = c[4] | c[5] << 16 | c[6] << 32;
		} else {
			type = 1;
			z[0] = c[0] | c[1] << 32, z[1] = c[2] | c[3] << 32, z[2] = c[4] | c[5] << 32, z[3] = c[6];
			z[0] |= 1ull << 62;
		}
	}
	if (t == nfull) { // the trailing header-only block (rld0.c:206-216): its header words
		for (int q = 0; q < (type ? 4 : 2); ++q) out[8 * t + q] = z[q];
		if (type) tail[0] = 4;
		return;
	}
	int p = type ? 4 : 2, r = 64;
	for (int64_t i = bs[b]; i < bs[b + 1]; ++i) { // rld_enc1, rld0.c:137-151, without the block switch
		const int64_t s = (int64_t)(words[i] >> 3), e = i + 1 < nr ? (int64_t)(words[i + 1] >> 3) : n_sym;
		const uint64_t l = (uint64_t)(e - s);
		const int y = fe_ilog2(l), zz = fe_ilog2((uint64_t)y + 1);
		int w = (zz << 1) + 1 + y + 3;
		const uint64_t delta = (l ^ (uint64_t)1 << y) | (uint64_t)(y + 1) << y;
		const uint64_t x = delta << 3 | (words[i] & 7);
		if (w > r) { // straddles two words of the block
			w -= r;
			if (p >= 7) { atomicOr(flag, 4u); break; } // cannot happen if the chain is right
			z[p++] |= x >> w;
			r = 64 - w;
			z[p] = x << r;
		} else {
			r -= w;
			z[p] |= x << r;
		}
	}
#pragma unroll
	for (int q = 0; q < 8; ++q) out[8 * t + q] = z[q];
}

/* d_words: nr words start << 3 | sym of the maximal runs of a BWT of n_sym symbols (device).  On success (0) *z_out is
 * a malloc'ed host array of *n_words words: the FMD data section incl. the trailing header.  1: this index needs block
 * headers wider than 16 bits somewhere (pack on the host); < 0: -1 out of memory, -2 HIP error, -3 internal. */
/* RB3GPU_FMD_DEBUG=1: say why the packer declined an index (read once) */
static bool fmd_debug(void)
{
	static const bool on = getenv("RB3GPU_FMD_DEBUG") != nullptr;
	return on;
}

/* (RB3GPU_FMD_DEBUG) are the bit offsets the widths' running sum?  bad[0] = mismatches, bad[1] = the first one */
__global__ void __launch_bounds__(256) k_fe_check_scan(const uint8_t *width, const uint64_t *P, int64_t n, unsigned long long *bad)
{
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n; i += (int64_t)gridDim.x * blockDim.x)
		if (P[i + 1] - P[i] != (uint64_t)width[i]) { atomicAdd(&bad[0], 1ull); atomicMin(&bad[1], (unsigned long long)i); }
}

/* exclusive scan of n code widths into bit offsets, a piece of at most 2^30 items at a time (the device scan takes a size_t, but an index of 24.8 G symbols
 * in contigs of 40-135 Mbp -- more than 2^32 runs -- came back with offsets that made the chain below wander for ten minutes and give up: round 6) */
#define FE_SCAN_PIECE ((int64_t)1 << 30)
static int fe_scan_widths(hipStream_t st, void *tmp, size_t tb, const uint8_t *width, uint64_t *P, int64_t n)
{
	uint64_t base = 0;
	for (int64_t o = 0; o < n; o += FE_SCAN_PIECE) {
		const int64_t m = n - o < FE_SCAN_PIECE ? n - o : FE_SCAN_PIECE;
		size_t b = tb;
		if (rocprim::exclusive_scan(tmp, b, rocprim::make_transform_iterator(width + o, fe_widen()), P + o, base, (size_t)m, rocprim::plus<uint64_t>(), st) != hipSuccess) return -2;
		if (o + m < n) { // the piece's total: its last offset + its last width
			uint64_t lastP = 0;
			uint8_t lastw = 0;
			if (hipMemcpyAsync(&lastP, P + o + m - 1, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipMemcpyAsync(&lastw, width + o + m - 1, 1, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -2;
			base = lastP + (uint64_t)lastw;
		}
	}
	return 0;
}

static double fe_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

/* THE PACKER AS A STREAM (round 6, last session).  rb3fmd_encode held the run starts of the WHOLE index (8 bytes per run) and ~26 bytes per run of tables beside
 * them: 4.99 G runs of four human haplotypes = 40 + 130 GB around an index of 12.6 GB, and a build at the full size of BASELINE configs[3] / [4] would have gone to
 * the host's encoder (ten minutes per 8 GB on one core).  Where a block ends only depends on where it starts and on the runs behind that point, so the runs can
 * come a PIECE at a time: the caller writes the next runs behind the ones the packer has carried over (rb3fmd_enc_buffer), rb3fmd_enc_piece packs every block that
 * is complete inside the piece and keeps what is left -- the last complete block (the header of the next one counts its symbols), the block that has begun, the
 * run whose end is not known yet: fewer than 2 FE_CHUNK runs -- for the next piece.  Inside a piece the run numbers and the block numbers are the piece's own
 * (block 1 is the first block to pack, block 0 the one before it); only (header type, global block number) of the block at hand travel, the second for the
 * superblocks' short last blocks.  A piece that is not the last looks to the kernels like a complete BWT that ends where its last run starts.
 * Device memory: ~26 bytes per run of a PIECE, whatever the index holds.  rb3fmd_encode is one final piece. */
#define FE_CARRY_MAX (2 * FE_CHUNK + 512)
#define FE_PIECE_MIN (2 * FE_CHUNK + 256) /* runs of a piece that is not the last: the chain must get through a chunk (the block at hand begins in the first hundred runs) */

struct rb3fmd_enc {
	hipStream_t st;
	int64_t n_sym, cap;             // symbols of the BWT; runs the buffers hold (carried + new)
	uint64_t *w, *carry;            // run words of the piece (cap + 2); the ones carried over, on their way to the front
	uint8_t *width, *E, *hE;
	uint16_t *Cn, *hCn;
	uint64_t *P, *out;
	int64_t *bs, *lists, *scal, *hlists, Kcap, out_cap;
	unsigned int *flag;
	void *tmp;
	size_t tb;
	uint64_t *host;                 // the stream so far (malloc)
	int64_t host_n, host_cap;
	int64_t nc, o, gb;              // carried runs in front of w; the block at hand starts at run o (of w) and has global number gb ...
	int oty, finished;              // ... and header type oty
	int64_t pieces;
};

static void fe_enc_free(rb3fmd_enc *e)
{
	if (!e) return;
	free(e->host); free(e->hE); free(e->hCn); free(e->hlists);
	void *all[] = { e->w, e->carry, e->width, e->P, e->out, e->E, e->Cn, e->lists, e->bs, e->scal, e->flag, e->tmp };
	for (void *p : all) if (p) (void)hipFree(p);
	delete e;
}

/* cap_runs: the most runs a piece will hold (what the caller adds at a time + FE_CARRY_MAX).  0 / -1 (out of memory) / -2 / -3 */
int rb3fmd_enc_begin(hipStream_t st, int64_t n_sym, int64_t cap_runs, rb3fmd_enc **out_e)
{
	*out_e = nullptr;
	if (n_sym <= 0 || cap_runs <= 0) return -3;
	rb3fmd_enc *e = new rb3fmd_enc;
	memset(e, 0, sizeof(*e));
	e->st = st, e->n_sym = n_sym, e->cap = cap_runs;
	const int64_t K = (cap_runs + FE_CHUNK - 1) / FE_CHUNK + 1;
	e->Kcap = K;
	int ret = 0;
	if (hipMalloc(&e->w, (size_t)(cap_runs + 2) * 8) != hipSuccess || hipMalloc(&e->carry, (size_t)FE_CARRY_MAX * 8) != hipSuccess || hipMalloc(&e->width, (size_t)cap_runs + 16) != hipSuccess ||
		hipMalloc(&e->P, (size_t)(cap_runs + 2) * 8) != hipSuccess || hipMalloc(&e->bs, (size_t)(cap_runs + 4) * 8) != hipSuccess || hipMalloc(&e->E, (size_t)K * FE_ENTRIES * 2) != hipSuccess ||
		hipMalloc(&e->Cn, (size_t)K * FE_ENTRIES * 4) != hipSuccess || hipMalloc(&e->lists, (size_t)(K + 1) * 24) != hipSuccess || hipMalloc(&e->scal, 64) != hipSuccess || hipMalloc(&e->flag, 16) != hipSuccess) { (void)hipGetLastError(); ret = -1; goto fail; }
	e->hE = (uint8_t*)malloc((size_t)K * FE_ENTRIES * 2), e->hCn = (uint16_t*)malloc((size_t)K * FE_ENTRIES * 4), e->hlists = (int64_t*)malloc((size_t)(K + 1) * 24);
	if (!e->hE || !e->hCn || !e->hlists) { ret = -1; goto fail; }
	if (rocprim::exclusive_scan(nullptr, e->tb, rocprim::make_transform_iterator(e->width, fe_widen()), e->P, (uint64_t)0, (size_t)(cap_runs + 1 < FE_SCAN_PIECE ? cap_runs + 1 : FE_SCAN_PIECE), rocprim::plus<uint64_t>(), st) != hipSuccess) { (void)hipGetLastError(); ret = -2; goto fail; }
	e->tb += 256;
	if (hipMalloc(&e->tmp, e->tb) != hipSuccess) { (void)hipGetLastError(); ret = -1; goto fail; }
	*out_e = e;
	return 0;
fail:
	fe_enc_free(e);
	return ret;
}

/* where the caller writes the next runs (device): behind the carried ones; *room = how many fit */
uint64_t *rb3fmd_enc_buffer(rb3fmd_enc *e, int64_t *room)
{
	if (room) *room = e->cap - e->nc;
	return e->w + e->nc;
}

void rb3fmd_enc_abort(rb3fmd_enc *e) { fe_enc_free(e); }

/* the words so far become the caller's (malloc); everything else is freed.  Only after a final piece. */
int rb3fmd_enc_end(rb3fmd_enc *e, uint64_t **z_out, int64_t *n_words)
{
	int ret = e->finished ? 0 : -3;
	if (ret == 0) *z_out = e->host, *n_words = e->host_n, e->host = nullptr;
	fe_enc_free(e);
	return ret;
}

/* n_new runs have been written behind the carried ones; final: they are the last of the BWT.  0, 1 (a block needs a 64-bit header or a code 64 bits: the
 * host's encoder writes such an index), < 0: -1 out of memory, -2 HIP error, -3 internal */
int rb3fmd_enc_piece(rb3fmd_enc *e, int64_t n_new, int final)
{
	int ret = 0;
	hipStream_t st = e->st;
	const int64_t n = e->nc + n_new;
	unsigned int hflag[4] = {0, 0, 0, 0};
	int64_t o = e->o, gb0 = e->gb, cur = 0, gbp = 0, Bl = 0;
	int oty = e->oty, cty = 0;
	const int64_t gb_base = e->gb - 1; // the block that starts at run o is block 1 of the piece, the one before it (it starts at run 0 of the piece) block 0
	bool ended = false;
	const double fe_t0 = fe_now();
	if (e->finished || n_new < 0 || n > e->cap || n <= 0 || (!final && n < FE_PIECE_MIN)) return -3;
	const int64_t nr = final ? n : n - 1; // (the end of the last run of a piece that is not the last is not known yet: the run waits for the next piece)
	int64_t n_sym = e->n_sym;
	if (!final) {
		uint64_t lastw = 0;
		if (hipMemcpyAsync(&lastw, e->w + n - 1, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); return -2; }
		n_sym = (int64_t)(lastw >> 3);
	}
	const int64_t K = (nr + FE_CHUNK - 1) / FE_CHUNK;
	const int64_t k_stop = final ? K : (nr - FE_ENTRIES) / FE_CHUNK; // chunks 0 .. k_stop-1 end at least FE_ENTRIES runs in front of the piece's end: the chain that leaves them is still inside
	const uint64_t *d_words = e->w, *P = e->P;
	int64_t *bs = e->bs, *lists = e->lists, *hlists = e->hlists, *scal = e->scal;
	unsigned int *flag = e->flag;
	if (K > e->Kcap || (!final && k_stop < 1)) return -3;
	++e->pieces;
	FE_HIP(hipMemsetAsync(flag, 0, 16, st));
	FE_HIP(hipMemsetAsync(bs, 0, 8, st)); // block 0 starts at run 0
	hipLaunchKernelGGL(k_fe_width, FE_GRID_LOOP(nr + 1), d_words, nr, n_sym, e->width, flag);
	if (fe_scan_widths(st, e->tmp, e->tb, e->width, e->P, nr + 1) < 0) { (void)hipGetLastError(); ret = -2; goto done; }
	hipLaunchKernelGGL(k_fe_table, FE_GRID(K * FE_ENTRIES * 2), K, nr, n_sym, P, d_words, e->E, e->Cn, flag);
	FE_HIP(hipMemcpyAsync(e->hE, e->E, (size_t)K * FE_ENTRIES * 2, hipMemcpyDeviceToHost, st));
	FE_HIP(hipMemcpyAsync(e->hCn, e->Cn, (size_t)K * FE_ENTRIES * 4, hipMemcpyDeviceToHost, st));
	FE_HIP(hipMemcpyAsync(hflag, flag, 4, hipMemcpyDeviceToHost, st));
	FE_HIP(hipStreamSynchronize(st));
	if (hflag[0] & 1u) { if (fmd_debug()) fprintf(stderr, "[fmdenc] a code of 64 bits or more\n"); ret = 1; goto done; }
	// the chain, one superblock at a time: the host walks the chunk tables
	for (;;) {
		const int64_t s = gb0 | (FE_SB_BLOCKS - 1); // global number of this superblock's last block
		const int64_t k0 = o / FE_CHUNK;
		if (!final && k0 >= k_stop) { cur = o, cty = oty, gbp = gb0; break; } // (the block at hand begins too close to the piece's end: it waits)
		int64_t *ids = hlists, *ent = hlists + (e->Kcap + 1), *bas = hlists + 2 * (e->Kcap + 1), nl = 0, c1[3];
		hipLaunchKernelGGL(k_fe_chain1, dim3(1), dim3(1), 0, st, o, oty, (k0 + 1) * FE_CHUNK < nr ? (k0 + 1) * FE_CHUNK : nr, nr, n_sym, P, d_words, scal);
		FE_HIP(hipMemcpyAsync(c1, scal, 24, hipMemcpyDeviceToHost, st));
		FE_HIP(hipStreamSynchronize(st));
		ids[nl] = k0, ent[nl] = o << 1 | oty, bas[nl] = gb0 - gb_base, ++nl;
		int64_t gb = gb0 + c1[1]; // the block that starts at run cur has global number gb
		cur = c1[0], cty = (int)c1[2];
		for (int64_t k = k0 + 1; cur < nr && gb <= s && (final || k < k_stop); ++k) {
			const int64_t d = cur - k * FE_CHUNK;
			if (d < 0 || d >= FE_ENTRIES) { if (fmd_debug()) fprintf(stderr, "[fmdenc] the chain left its chunk: run %lld in chunk %lld of %lld (block %lld)\n", (long long)cur, (long long)k, (long long)K, (long long)gb); ret = -3; goto done; }
			ids[nl] = k, ent[nl] = cur << 1 | cty, bas[nl] = gb - gb_base, ++nl;
			const int64_t te = (k * FE_ENTRIES + d) * 2 + cty;
			gb += e->hCn[te];
			cur = ((k + 1) * FE_CHUNK < nr ? (k + 1) * FE_CHUNK : nr) + (e->hE[te] & 0x7F);
			cty = e->hE[te] >> 7;
		}
		FE_HIP(hipMemcpyAsync(lists, ids, (size_t)nl * 8, hipMemcpyHostToDevice, st));
		FE_HIP(hipMemcpyAsync(lists + (e->Kcap + 1), ent, (size_t)nl * 8, hipMemcpyHostToDevice, st));
		FE_HIP(hipMemcpyAsync(lists + 2 * (e->Kcap + 1), bas, (size_t)nl * 8, hipMemcpyHostToDevice, st));
		hipLaunchKernelGGL(k_fe_emit, FE_GRID(nl), nl, (const int64_t*)lists, (const int64_t*)(lists + (e->Kcap + 1)), (const int64_t*)(lists + 2 * (e->Kcap + 1)), nr, n_sym, P, d_words, s - gb_base, bs, flag);
		FE_HIP(hipStreamSynchronize(st)); // (the host's lists are rewritten by the next superblock)
		if (gb <= s) { // the superblock goes on behind this piece -- or the data end in it
			gbp = gb;
			ended = final && cur >= nr;
			if (!ended && final) { ret = -3; goto done; }
			break;
		}
		hipLaunchKernelGGL(k_fe_special, dim3(1), dim3(1), 0, st, P, d_words, nr, n_sym, (const int64_t*)bs, s - gb_base, scal, flag);
		{
			int64_t sp[2];
			FE_HIP(hipMemcpyAsync(sp, scal, 16, hipMemcpyDeviceToHost, st));
			FE_HIP(hipStreamSynchronize(st));
			o = sp[0], oty = (int)sp[1];
		}
		gb0 = s + 1;
		if (o >= nr) {
			if (!final) { ret = -3; goto done; } // (cannot be: the last block of a superblock began in a chunk that ends FE_ENTRIES runs in front of the piece's end)
			gbp = gb0, cur = nr, ended = true;
			break;
		}
	}
	Bl = gbp - gb_base; // the piece's number of the block that starts at `cur`: behind the data (final), or the one that waits for the next piece
	if (Bl < 1 || (!ended && Bl < 2)) { if (fmd_debug()) fprintf(stderr, "[fmdenc] piece %lld of %lld runs: no block completed\n", (long long)e->pieces, (long long)n); ret = -3; goto done; }
	{
		const int64_t endrun = ended ? nr : cur;
		FE_HIP(hipMemcpyAsync(bs + Bl, &endrun, 8, hipMemcpyHostToDevice, st));
		FE_HIP(hipStreamSynchronize(st));
	}
	{
		const int64_t nfull = Bl - 1; // blocks 1 .. Bl-1 are complete
		if (8 * (nfull + 1) + 8 > e->out_cap) {
			if (e->out) (void)hipFree(e->out);
			e->out = nullptr, e->out_cap = 8 * (nfull + 1) + 8 + (nfull >> 2) * 8;
			if (hipMalloc(&e->out, (size_t)e->out_cap * 8) != hipSuccess) { (void)hipGetLastError(); e->out_cap = 0; ret = -1; goto done; }
		}
		hipLaunchKernelGGL(k_fe_pack, FE_GRID(nfull + 1), d_words, nr, n_sym, (const int64_t*)bs, (int64_t)1, nfull, ended ? 1 : 0, e->out, flag, flag + 1);
		FE_HIP(hipMemcpyAsync(hflag, flag, 8, hipMemcpyDeviceToHost, st));
		FE_HIP(hipStreamSynchronize(st));
		if (hflag[0] & 11u) { if (fmd_debug()) fprintf(stderr, "[fmdenc] flags %u: a block needs a 64-bit header\n", hflag[0]); ret = 1; goto done; } // (not produced here: the host's encoder writes such an index)
		if (hflag[0] & 4u) { if (fmd_debug()) fprintf(stderr, "[fmdenc] flags %u: the packer met a block it cannot write\n", hflag[0]); ret = -3; goto done; }
		const int64_t tailw = !ended ? 0 : hflag[1] == 4u ? 4 : 2; // header words of the trailing header-only block (rld0.c:211)
		const int64_t nw = 8 * nfull + tailw;
		if (e->host_n + nw > e->host_cap) {
			const int64_t want = ended ? e->host_n + nw : (e->host_n + nw) + ((e->host_n + nw) >> 1) + 1024;
			uint64_t *nh = (uint64_t*)realloc(e->host, (size_t)want * 8);
			if (!nh) { ret = -1; goto done; }
			e->host = nh, e->host_cap = want;
		}
		if (nw > 0) FE_HIP(hipMemcpy(e->host + e->host_n, e->out, (size_t)nw * 8, hipMemcpyDeviceToHost));
		e->host_n += nw;
	}
	if (ended) e->finished = 1;
	else { // what is left of the piece, from the start of its last complete block on, moves to the front
		int64_t cs = 0;
		FE_HIP(hipMemcpy(&cs, bs + Bl - 1, 8, hipMemcpyDeviceToHost));
		const int64_t keep = n - cs;
		if (cs < 0 || cs > cur || keep > FE_CARRY_MAX) { if (fmd_debug()) fprintf(stderr, "[fmdenc] piece %lld: %lld runs to carry over (from run %lld of %lld)\n", (long long)e->pieces, (long long)keep, (long long)cs, (long long)n); ret = -3; goto done; }
		FE_HIP(hipMemcpyAsync(e->carry, e->w + cs, (size_t)keep * 8, hipMemcpyDeviceToDevice, st));
		FE_HIP(hipMemcpyAsync(e->w, e->carry, (size_t)keep * 8, hipMemcpyDeviceToDevice, st));
		FE_HIP(hipStreamSynchronize(st));
		e->nc = keep, e->o = cur - cs, e->oty = cty, e->gb = gbp;
	}
	if (fmd_debug()) fprintf(stderr, "[fmdenc] piece %lld: %lld runs (%lld carried in), %lld blocks%s, %.3f s\n", (long long)e->pieces, (long long)n, (long long)(n - n_new), (long long)(Bl - 1), ended ? ", the last" : "", fe_now() - fe_t0);
done:
	return ret;
}

/* d_words: nr words start << 3 | sym of the maximal runs of a BWT of n_sym symbols (device), all at once.  On success (0) *z_out is a malloc'ed host array of
 * *n_words words: the FMD data section incl. the trailing header.  1: this index needs block headers wider than 32 bits somewhere (pack on the host);
 * < 0: -1 out of memory, -2 HIP error, -3 internal. */
int rb3fmd_encode(hipStream_t st, int64_t n_sym, int64_t nr, const uint64_t *d_words, uint64_t **z_out, int64_t *n_words)
{
	rb3fmd_enc *e = nullptr;
	*z_out = nullptr, *n_words = 0;
	if (nr <= 0 || n_sym <= 0) return -3;
	int r = rb3fmd_enc_begin(st, n_sym, nr + 16, &e);
	if (r < 0) return r;
	if (hipMemcpyAsync(rb3fmd_enc_buffer(e, nullptr), d_words, (size_t)nr * 8, hipMemcpyDeviceToDevice, st) != hipSuccess) { (void)hipGetLastError(); rb3fmd_enc_abort(e); return -2; }
	r = rb3fmd_enc_piece(e, nr, 1);
	if (r != 0) { rb3fmd_enc_abort(e); return r; }
	return rb3fmd_enc_end(e, z_out, n_words);
}

/* ------------------------------------------------------------------------------------------ */
/* FMD decoding on the device (rld_dec0 / rld_dec, rld0.h:85-122; fm-index.c:56-85)            */
/* ------------------------------------------------------------------------------------------ */

/* A 64-byte block is self-contained: header (2, 4 or 7 words by type, rld0.c:71-73), then Elias-delta coded runs
 * delta(len) << 3 | sym, MSB first, ended by six zero bits or the block's last usable word (the last word of a
 * superblock of 2^23 words is never used).  One thread per block: a first pass adds up the symbols of every block,
 * an exclusive scan places them, a second pass writes the symbols, one byte each, where rb3gpu_from_plain_dev wants
 * them; runs of 64 k symbols and more are left to whole workgroups (k_fd_long). */
#define FD_LBITS 23
#define FD_LONG  65536
#define FD_LONG_CAP (1 << 20)

template<class F>
__device__ __forceinline__ bool fd_block(const uint64_t *z, int64_t head, F f)
{
	const int type = (int)(z[head] >> 62);
	if (type > 2) return false;
	const int64_t stail = head + 8 - (((head + 8) & ((1LL << FD_LBITS) - 1)) == 0 ? 2 : 1);
	int64_t p = head + (type == 0 ? 2 : type == 1 ? 4 : 7);
	int r = 64;
	while (p <= stail) {
		uint64_t x = z[p] << (64 - r);
		if (r != 64 && p != stail) x |= z[p + 1] >> r;
		if (x == 0) break;
		const int lz = __clzll((long long)x);
		if (lz >= 6) break; // no delta code starts with six zeros: end of block
		int wd = 2 * lz + 1;
		const int y = (int)(x >> (64 - wd)) - 1;
		int64_t l = (int64_t)1 << y;
		if (y > 0) l |= (int64_t)(x << wd >> (64 - y));
		wd += y;
		const int c = (int)(x << wd >> 61);
		wd += 3;
		if (r > wd) r -= wd; else ++p, r = 64 + r - wd;
		if (c > 5) return false;
		f(c, l);
	}
	return true;
}

__global__ void __launch_bounds__(256) k_fd_count(const uint64_t *z, int64_t nblk, uint64_t *btot, unsigned int *flag)
{
	const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= nblk) return;
	uint64_t tot = 0;
	if (!fd_block(z, b * 8, [&](int, int64_t l) { tot += (uint64_t)l; })) atomicOr(flag, 1u);
	btot[b] = tot;
}

__device__ __forceinline__ void fd_fill(uint8_t *q, int64_t l, int c)
{
	int64_t i = 0;
	for (; i < l && ((uintptr_t)(q + i) & 7); ++i) q[i] = (uint8_t)c; // up to an 8-byte boundary
	const uint64_t w = 0x0101010101010101ull * (uint64_t)c;
	for (; i + 8 <= l; i += 8) *(uint64_t*)(q + i) = w;
	for (; i < l; ++i) q[i] = (uint8_t)c;
}

struct fd_long_t { int64_t off, len; int64_t sym; };

__global__ void __launch_bounds__(256) k_fd_fill(const uint64_t *z, int64_t nblk, const uint64_t *boff, uint8_t *out, fd_long_t *lng, unsigned int *nlong)
{
	const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= nblk) return;
	int64_t off = (int64_t)boff[b];
	fd_block(z, b * 8, [&](int c, int64_t l) {
		bool done = false;
		if (l >= FD_LONG) {
			const unsigned int k = atomicAdd(nlong, 1u);
			if (k < (unsigned int)FD_LONG_CAP) { fd_long_t e; e.off = off, e.len = l, e.sym = c; lng[k] = e; done = true; }
		}
		if (!done) fd_fill(out + off, l, c);
		off += l;
	});
}

__global__ void __launch_bounds__(256) k_fd_long(const fd_long_t *lng, const unsigned int *nlong, uint8_t *out)
{
	const unsigned int n = *nlong < (unsigned int)FD_LONG_CAP ? *nlong : (unsigned int)FD_LONG_CAP;
	for (unsigned int k = blockIdx.y; k < n; k += gridDim.y) {
		const fd_long_t e = lng[k];
		// this workgroup's share of the run: 64 KB pieces, gridDim.x of them in flight per run
		for (int64_t s = (int64_t)blockIdx.x * FD_LONG; s < e.len; s += (int64_t)gridDim.x * FD_LONG) {
			const int64_t m = e.len - s < FD_LONG ? e.len - s : FD_LONG;
			uint8_t *q = out + e.off + s;
			for (int64_t i = threadIdx.x; i < m; i += blockDim.x) q[i] = (uint8_t)e.sym;
		}
	}
}

/* the symbols of positions [p0, p1) only, written to out[0 .. p1 - p0): a large index is decoded piece by piece */
__global__ void __launch_bounds__(256) k_fd_fill_range(const uint64_t *z, int64_t nblk, const uint64_t *boff, int64_t p0, int64_t p1, uint8_t *out, fd_long_t *lng, unsigned int *nlong)
{
	const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (b >= nblk) return;
	int64_t off = (int64_t)boff[b];
	if (off >= p1 || (int64_t)boff[b + 1] <= p0) return;
	fd_block(z, b * 8, [&](int c, int64_t l) {
		const int64_t a = off > p0 ? off : p0, e = off + l < p1 ? off + l : p1;
		if (a < e) {
			bool done = false;
			if (e - a >= FD_LONG) {
				const unsigned int k = atomicAdd(nlong, 1u);
				if (k < (unsigned int)FD_LONG_CAP) { fd_long_t x; x.off = a - p0, x.len = e - a, x.sym = c; lng[k] = x; done = true; }
			}
			if (!done) fd_fill(out + (a - p0), e - a, c);
		}
		off += l;
	});
}

struct rb3fmd_dec {
	hipStream_t st;
	int64_t nblk;
	const uint64_t *z;
	uint64_t *btot, *boff; // nblk + 1 each: symbols per block, their exclusive prefix sum
	void *tmp;
	fd_long_t *lng;
	unsigned int *flag;  // [0] bad block, [1] number of long runs
};

/* d_z: the word stream in device memory, followed by two zero words.  Returns 0 and the number of symbols, -1 (out of
 * memory), -2 (HIP error), -3 (not a valid stream). */
int rb3fmd_decode_begin(hipStream_t st, int64_t n_words, const uint64_t *d_z, rb3fmd_dec **ctx, int64_t *n_sym)
{
	int ret = 0;
	*ctx = nullptr, *n_sym = 0;
	const int64_t nblk = n_words >> 3; // (a trailing header-only block, if any, holds no runs: rld0.c:190-216)
	if (nblk <= 0) return -3;
	rb3fmd_dec *c = new rb3fmd_dec;
	c->st = st, c->nblk = nblk, c->z = d_z, c->btot = nullptr, c->boff = nullptr, c->tmp = nullptr, c->lng = nullptr, c->flag = nullptr;
	size_t tb = 0;
	unsigned int hflag = 0;
	uint64_t last[2] = {0, 0};
	if (hipMalloc(&c->btot, (size_t)(nblk + 1) * 8) != hipSuccess || hipMalloc(&c->boff, (size_t)(nblk + 1) * 8) != hipSuccess || hipMalloc(&c->lng, (size_t)FD_LONG_CAP * sizeof(fd_long_t)) != hipSuccess ||
		hipMalloc(&c->flag, 16) != hipSuccess) { (void)hipGetLastError(); ret = -1; goto done; }
	FE_HIP(rocprim::exclusive_scan(nullptr, tb, c->btot, c->boff, (uint64_t)0, (size_t)(nblk + 1), rocprim::plus<uint64_t>(), st));
	if (hipMalloc(&c->tmp, tb + 256) != hipSuccess) { (void)hipGetLastError(); ret = -1; goto done; }
	FE_HIP(hipMemsetAsync(c->flag, 0, 16, st));
	FE_HIP(hipMemsetAsync(c->btot + nblk, 0, 8, st));
	hipLaunchKernelGGL(k_fd_count, FE_GRID(nblk), d_z, nblk, c->btot, c->flag);
	{ size_t b = tb; FE_HIP(rocprim::exclusive_scan(c->tmp, b, c->btot, c->boff, (uint64_t)0, (size_t)(nblk + 1), rocprim::plus<uint64_t>(), st)); }
	FE_HIP(hipMemcpyAsync(&hflag, c->flag, 4, hipMemcpyDeviceToHost, st));
	FE_HIP(hipMemcpyAsync(last, c->boff + nblk, 8, hipMemcpyDeviceToHost, st));
	FE_HIP(hipStreamSynchronize(st));
	if (hflag != 0 || last[0] == 0 || last[0] >= (1ull << 62)) { ret = -3; goto done; }
	*n_sym = (int64_t)last[0];
	*ctx = c;
	return 0;
done:
	if (c->btot) (void)hipFree(c->btot);
	if (c->boff) (void)hipFree(c->boff);
	if (c->tmp) (void)hipFree(c->tmp);
	if (c->lng) (void)hipFree(c->lng);
	if (c->flag) (void)hipFree(c->flag);
	delete c;
	return ret;
}

/* the symbols into d_plain (n_sym bytes); frees the context */
int rb3fmd_decode_fill(rb3fmd_dec *c, uint8_t *d_plain)
{
	int ret = 0;
	hipStream_t st = c->st;
	if (d_plain == nullptr) goto done; // (the caller gives up: just free the context)
	hipLaunchKernelGGL(k_fd_fill, FE_GRID(c->nblk), c->z, c->nblk, (const uint64_t*)c->boff, d_plain, c->lng, c->flag + 1);
	hipLaunchKernelGGL(k_fd_long, dim3(64, 256), dim3(256), 0, st, (const fd_long_t*)c->lng, (const unsigned int*)(c->flag + 1), d_plain);
	FE_HIP(hipStreamSynchronize(st));
done:
	(void)hipFree(c->btot); (void)hipFree(c->boff); (void)hipFree(c->tmp); (void)hipFree(c->lng); (void)hipFree(c->flag);
	delete c;
	return ret;
}

/* the symbols of [p0, p1) of a stream opened with rb3fmd_decode_begin into d_out (p1 - p0 bytes); the context stays open */
int rb3fmd_decode_range(rb3fmd_dec *c, int64_t p0, int64_t p1, uint8_t *d_out)
{
	int ret = 0;
	hipStream_t st = c->st;
	if (p1 <= p0) return 0;
	FE_HIP(hipMemsetAsync(c->flag + 1, 0, 4, st));
	hipLaunchKernelGGL(k_fd_fill_range, FE_GRID(c->nblk), c->z, c->nblk, (const uint64_t*)c->boff, p0, p1, d_out, c->lng, c->flag + 1);
	hipLaunchKernelGGL(k_fd_long, dim3(64, 256), dim3(256), 0, st, (const fd_long_t*)c->lng, (const unsigned int*)(c->flag + 1), d_out);
done:
	return ret;
}

void rb3fmd_decode_end(rb3fmd_dec *c)
{
	if (!c) return;
	(void)hipStreamSynchronize(c->st);
	(void)rb3fmd_decode_fill(c, nullptr);
}

int64_t rb3fmd_decode_bytes(const rb3fmd_dec *c) { return c ? (int64_t)(c->nblk + 1) * 16 + (int64_t)FD_LONG_CAP * (int64_t)sizeof(fd_long_t) : 0; }
