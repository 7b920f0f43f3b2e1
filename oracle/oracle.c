/*
 * oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, CPU-only restatement of the algorithm on the hot path of
 * `ropebwt3 build` (the incremental FM-index merge).  It exists so that the
 * HIP engine can be checked bit-for-bit on seeded inputs; it is never linked
 * into, imported by or called from the product (ropebwt3_amd/ and
 * librb3gpu.so).  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load liboracle.so.
 *
 * Parity status: PINNED.  tests/test_cpu_oracle_pin.py checks every function here
 * against (a) the known-answer vectors K1-K4 of SURVEY.md section 8(c), (b) the golden
 * fixtures under tests/golden/ that were produced by the unmodified reference
 * binary (tools/make_golden.py), and (c) -- whenever oracle/_ref/ exists -- the
 * reference shared object built from /root/reference by oracle/Makefile, on
 * fresh random inputs.
 *
 * What is restated (all citations are file:line in the reference tree):
 *   orc_bwt()            sais-ss.c:10-56   text of 0-terminated strings -> BWT in
 *                                          generalised-suffix-array order (the i-th
 *                                          sentinel sorts before the (i+1)-th).  The
 *                                          reference delegates to libsais; here a
 *                                          textbook prefix-doubling sort is used, the
 *                                          BWT being uniquely defined by the order.
 *   orc_mg_rank_plain()  fm-index.c:160-175, 202-225   LF array of B2 + one LF chain
 *                                          per sentinel, rank over B1.
 *   orc_merge_plain()    fm-index.c:237-249, 279-303   merged[ka[kb]+kb] = B2[kb].
 *   orc_runs()           fm-index.c:12-29  maximal-run list of a plain BWT (what
 *                                          rld_enc sees after coalescing, rld0.c:153-161).
 *
 * The reference keeps B1 in a B+-tree of run-length blocks (mrope/rope/rle).
 * Rank over that tree is, by definition, occ(c, k) = #{i < k : B1[i] = c}
 * (mrope.h:60-68), so the restatement keeps B1 as a flat byte array with
 * sampled occurrence counts; insertion into the tree is, by definition,
 * order-preserving insertion of B2[kb] at merged position ka[kb]+kb
 * (fm-index.c:247), so the restatement scatters into a flat array.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_ASIZE 6 /* $ACGTN, fm-index.h:15 */

/* ------------------------------------------------------------------------- */
/* orc_bwt: multi-string BWT by prefix doubling (restates sais-ss.c:23-26)    */
/* ------------------------------------------------------------------------- */

typedef struct { int64_t r0, r1; int64_t i; } orc_trip_t;

static int orc_trip_cmp(const void *a, const void *b)
{
	const orc_trip_t *x = (const orc_trip_t*)a, *y = (const orc_trip_t*)b;
	if (x->r0 != y->r0) return x->r0 < y->r0 ? -1 : 1;
	if (x->r1 != y->r1) return x->r1 < y->r1 ? -1 : 1;
	return 0;
}

/* text: `len` bytes in 0..5, every string terminated by a 0, last byte 0.
 * On return bwt[i] = text[SA[i]-1] (or text[len-1] when SA[i]==0) where SA is
 * the suffix array with the j-th 0 smaller than the (j+1)-th 0 and every 0
 * smaller than 1..5.  Returns the number of strings, or -1 on bad input. */
int64_t orc_bwt(int64_t len, const uint8_t *text, uint8_t *bwt)
{
	int64_t i, n_seq = 0, h, *rank, *tmp;
	orc_trip_t *t;
	if (len <= 0 || text[len-1] != 0) return -1;
	for (i = 0; i < len; ++i) {
		if (text[i] >= ORC_ASIZE) return -1;
		if (text[i] == 0) ++n_seq;
	}
	rank = (int64_t*)malloc(len * sizeof(int64_t));
	tmp = (int64_t*)malloc(len * sizeof(int64_t));
	t = (orc_trip_t*)malloc(len * sizeof(orc_trip_t));
	/* initial rank: the j-th sentinel gets j; symbol c>0 gets n_seq + c - 1 */
	{
		int64_t j = 0;
		for (i = 0; i < len; ++i)
			rank[i] = text[i] == 0 ? j++ : n_seq + text[i] - 1;
	}
	for (h = 1;; h <<= 1) {
		int64_t n_distinct;
		for (i = 0; i < len; ++i) {
			t[i].r0 = rank[i];
			/* a sentinel is unique: nothing after it is ever compared */
			t[i].r1 = (rank[i] < n_seq || i + h >= len) ? -1 : rank[i + h];
			t[i].i = i;
		}
		qsort(t, len, sizeof(orc_trip_t), orc_trip_cmp);
		n_distinct = 0;
		for (i = 0; i < len; ++i) {
			if (i > 0 && orc_trip_cmp(&t[i-1], &t[i]) != 0) ++n_distinct;
			tmp[t[i].i] = n_distinct;
		}
		memcpy(rank, tmp, len * sizeof(int64_t));
		if (n_distinct == len - 1) break;
	}
	for (i = 0; i < len; ++i) { /* rank[] is now the inverse suffix array */
		int64_t p = i;
		bwt[rank[p]] = p == 0 ? text[len-1] : text[p-1];
	}
	free(t); free(tmp); free(rank);
	return n_seq;
}

/* ------------------------------------------------------------------------- */
/* flat rank index over a plain BWT (stands in for mrope rank, mrope.c:71-121) */
/* ------------------------------------------------------------------------- */

#define ORC_OCC_SHIFT 6

typedef struct {
	int64_t n;
	const uint8_t *b;
	int64_t *occ;             /* occ[(i>>6)*6 + c] = #{j < (i>>6<<6) : b[j]==c} */
	int64_t acc[ORC_ASIZE+1]; /* C array, fm-index.c:544-550 */
} orc_fmi_t;

static void orc_fmi_init(orc_fmi_t *f, int64_t n, const uint8_t *b)
{
	int64_t i, c[ORC_ASIZE], nb = (n >> ORC_OCC_SHIFT) + 1;
	int a;
	f->n = n, f->b = b;
	f->occ = (int64_t*)malloc(nb * ORC_ASIZE * sizeof(int64_t));
	memset(c, 0, sizeof(c));
	for (i = 0; i < n; ++i) {
		if ((i & ((1 << ORC_OCC_SHIFT) - 1)) == 0)
			memcpy(&f->occ[(i >> ORC_OCC_SHIFT) * ORC_ASIZE], c, sizeof(c));
		++c[b[i]];
	}
	if ((n & ((1 << ORC_OCC_SHIFT) - 1)) == 0)
		memcpy(&f->occ[(n >> ORC_OCC_SHIFT) * ORC_ASIZE], c, sizeof(c));
	for (f->acc[0] = 0, a = 0; a < ORC_ASIZE; ++a) f->acc[a+1] = f->acc[a] + c[a];
}

/* ok[c] = #{i < k : B[i] = c}; k >= n returns the totals (mrope.c:89-93) */
static inline void orc_rank1a(const orc_fmi_t *f, int64_t k, int64_t ok[ORC_ASIZE])
{
	int64_t i, k0;
	if (k > f->n) k = f->n;
	k0 = k >> ORC_OCC_SHIFT << ORC_OCC_SHIFT;
	memcpy(ok, &f->occ[(k >> ORC_OCC_SHIFT) * ORC_ASIZE], ORC_ASIZE * sizeof(int64_t));
	for (i = k0; i < k; ++i) ++ok[f->b[i]];
}

/* ------------------------------------------------------------------------- */
/* orc_mg_rank_plain: restates fm-index.c:160-175 and 202-225                 */
/* ------------------------------------------------------------------------- */

/* one LF chain (fm-index.c:160-175) */
static void orc_mg_rank1_plain(const orc_fmi_t *fa, int64_t *rb, int64_t p)
{
	int64_t ka = fa->acc[1], kb = p;
	int c, last_c = 0;
	for (;;) {
		int64_t oa[ORC_ASIZE], r = rb[kb] >> 3;
		c = rb[kb] & 7;
		rb[kb] = (ka + kb) << 6 | c << 3 | last_c;
		last_c = c;
		if (c == 0) break;
		kb = r;
		orc_rank1a(fa, ka, oa);
		ka = fa->acc[c] + oa[c];
	}
}

/* rb must hold n2 int64.  On return rb[kb] = (ka[kb]+kb)<<6 | B2[kb]<<3 | first
 * symbol of row kb's suffix, exactly the reference's rb[] after
 * rb3_mg_rank_plain().  acc2[7] receives the C array of B2.  Returns 0, or -1
 * if a symbol is out of range. */
int orc_mg_rank_plain(int64_t n1, const uint8_t *b1, int64_t n2, const uint8_t *b2, int64_t *rb, int64_t acc2[ORC_ASIZE+1], int n_threads)
{
	orc_fmi_t fa;
	int64_t i, k, c[ORC_ASIZE];
	int a;
	for (i = 0; i < n1; ++i) if (b1[i] >= ORC_ASIZE) return -1;
	for (i = 0; i < n2; ++i) if (b2[i] >= ORC_ASIZE) return -1;
	orc_fmi_init(&fa, n1, b1);
	memset(c, 0, sizeof(c));
	for (i = 0; i < n2; ++i) ++c[b2[i]];                      /* fm-index.c:206-208 */
	for (acc2[0] = 0, a = 0; a < ORC_ASIZE; ++a) acc2[a+1] = acc2[a] + c[a];
	memset(c, 0, sizeof(c));
	for (i = 0; i < n2; ++i) {                                /* fm-index.c:211-216 */
		a = b2[i];
		rb[i] = (acc2[a] + c[a]) << 3 | a;
		++c[a];
	}
	if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
	for (k = 0; k < acc2[1]; ++k)                             /* fm-index.c:217-224 */
		orc_mg_rank1_plain(&fa, rb, k);
	free(fa.occ);
	return 0;
}

/* ------------------------------------------------------------------------- */
/* orc_merge_plain: restates fm-index.c:279-303 (+237-249)                     */
/* ------------------------------------------------------------------------- */

/* out must hold n1+n2 bytes.  Returns 0; -2 if the bucket invariant asserted at
 * fm-index.c:246 is violated; -3 if two rows land on one merged position. */
int orc_merge_plain(int64_t n1, const uint8_t *b1, int64_t n2, const uint8_t *b2, uint8_t *out, int n_threads)
{
	int64_t *rb, acc2[ORC_ASIZE+1], i, j, n = n1 + n2;
	int ret = 0, a;
	rb = (int64_t*)malloc((n2 > 0 ? n2 : 1) * sizeof(int64_t));
	if (orc_mg_rank_plain(n1, b1, n2, b2, rb, acc2, n_threads) < 0) { free(rb); return -1; }
	memset(out, 0xff, n);
	for (a = 0; a < ORC_ASIZE; ++a)                           /* worker_mgins, one bucket each */
		for (i = acc2[a]; i < acc2[a+1]; ++i) {
			int64_t x = rb[i], pos = x >> 6;
			if ((x & 7) != a) ret = -2;                       /* fm-index.c:246 */
			if (pos < 0 || pos >= n || out[pos] != 0xff) { ret = -3; continue; }
			out[pos] = x >> 3 & 7;                            /* fm-index.c:247 */
		}
	for (i = j = 0; i < n; ++i)                               /* B1 keeps its relative order */
		if (out[i] == 0xff) out[i] = b1[j++];
	if (j != n1 && ret == 0) ret = -3;
	free(rb);
	return ret;
}

/* ------------------------------------------------------------------------- */
/* orc_runs: maximal runs of a plain BWT (fm-index.c:12-29; rld0.c:153-161)    */
/* ------------------------------------------------------------------------- */

/* runs[i] = len<<3 | sym.  Pass runs==NULL to only count.  Returns #runs. */
int64_t orc_runs(int64_t n, const uint8_t *b, int64_t *runs)
{
	int64_t i, i0, k = 0;
	for (i0 = 0, i = 1; i <= n; ++i)
		if (i == n || b[i0] != b[i]) {
			if (runs) runs[k] = (i - i0) << 3 | b[i0];
			++k, i0 = i;
		}
	return n > 0 ? k : 0;
}

/* ------------------------------------------------------------------------- */
/* orc_ssa_gen: sampled suffix array, restates ssa.c:17-40 (ssa_gen1) and      */
/* ssa.c:54-81 (rb3_ssa_gen) sequentially over a plain BWT                     */
/* ------------------------------------------------------------------------- */

/* sizes as rb3_ssa_gen computes them (ssa.c:62-64) */
void orc_ssa_dims(int64_t n, const uint8_t *b, int ss, int64_t *m, int64_t *n_ssa, int *ms)
{
	int64_t i, c0 = 0;
	int x;
	for (i = 0; i < n; ++i) c0 += b[i] == 0;
	for (x = 1; 1LL << x < c0; ++x) {}
	*m = c0, *ms = x, *n_ssa = (n - c0 + (1LL << ss) - 1LL) >> ss;
}

/* r2i: m words, ssa: n_ssa words, both zero-filled here like RB3_CALLOC (ssa.c:65-66). Returns 0. */
int orc_ssa_gen(int64_t n, const uint8_t *b, int ss, uint64_t *r2i, uint64_t *ssa)
{
	orc_fmi_t f;
	int64_t m, n_ssa, k0, mask = (1LL << ss) - 1, nb = 0, mb = 0, *buf = 0, i;
	int ms;
	orc_ssa_dims(n, b, ss, &m, &n_ssa, &ms);
	orc_fmi_init(&f, n, b);
	memset(r2i, 0, m * 8);
	memset(ssa, 0, n_ssa * 8);
	for (k0 = 0; k0 < m; ++k0) { /* ssa_gen1 (ssa.c:17-40) for string k0 */
		int64_t ok[ORC_ASIZE], k = k0, l = 0;
		int c;
		nb = 0;
		do {
			++l;
			c = k < f.n ? f.b[k] : 0; /* what rb3_fmi_rank1a returns: the symbol at k */
			orc_rank1a(&f, k, ok);
			k = f.acc[c] + ok[c];
			if (c) {
				if (((k - f.acc[1]) & mask) == 0) {
					int64_t x = (k - f.acc[1]) >> ss;
					ssa[x] = l;
					if (nb == mb) { mb = mb ? mb << 1 : 256; buf = (int64_t*)realloc(buf, mb * sizeof(int64_t)); }
					buf[nb++] = x;
				}
			} else r2i[k] = k0;
		} while (c);
		for (i = 0; i < nb; ++i)
			ssa[buf[i]] = (uint64_t)(l - 1 - (int64_t)ssa[buf[i]]) << ms | (uint64_t)k0;
	}
	free(buf); free(f.occ);
	return 0;
}

/* nt6 encoding + both-strand text assembly: restates io.c:12-40, 84-102.
 * seqs are given as one buffer of ASCII sequences separated by '\n'.
 * out must hold 2*(in_len+1) bytes.  Returns the text length. */
int64_t orc_text_from_lines(int64_t in_len, const char *in, int is_for, int is_rev, uint8_t *out)
{
	static const uint8_t tab[128] = { /* io.c:12-21 */
		0,1,2,3,4,5,5,5,5,5,5,5,5,5,5,5, 5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,
		5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5, 5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,5,
		5,1,5,2,5,5,5,3,5,5,5,5,5,5,5,5, 5,5,5,5,4,5,5,5,5,5,5,5,5,5,5,5,
		5,1,5,2,5,5,5,3,5,5,5,5,5,5,5,5, 5,5,5,5,4,5,5,5,5,5,5,5,5,5,5,5 };
	int64_t i, st = 0, l = 0;
	for (i = 0; i <= in_len; ++i) {
		if (i == in_len || in[i] == '\n') {
			int64_t j, sl = i - st;
			if (sl > 0 || i < in_len) {
				if (is_for) {
					for (j = 0; j < sl; ++j) { uint8_t ch = (uint8_t)in[st+j]; out[l++] = ch < 128 ? tab[ch] : 5; }
					out[l++] = 0;
				}
				if (is_rev) { /* io.c:30-40 */
					for (j = sl - 1; j >= 0; --j) {
						uint8_t ch = (uint8_t)in[st+j]; int c = ch < 128 ? tab[ch] : 5;
						out[l++] = (c >= 1 && c <= 4) ? 5 - c : c;
					}
					out[l++] = 0;
				}
			}
			st = i + 1;
		}
	}
	return l;
}
