/*
 * psort_core.h -- PARALLEL host suffix sorter: prefix doubling (Manber & Myers 1990, with the group bookkeeping of Larsson &
 * Sadakane 1999) over an integer alphabet, OpenMP.  Included twice by psort.c with PIDX = uint32_t / uint64_t.
 *
 * What it is for ("a parallel host suffix sorter"): the reference sorts a batch with libsais + OpenMP
 * (sais-ss.c:15-22, 35-42); this engine sorts batches on the GPU, and the host sorter is what is left for a record the GPU sorter
 * does not take (>= 2^31 symbols), for `--host-sort`, and when the device has no room.  The sequential SA-IS of sais_core.h stays
 * the sorter of one thread; with -t N > 1 this one takes over.  Same contract as sais_bwt: T[0..n) in [0, K), T[n-1] = 0 unique.
 *
 * Round h sorts every group of suffixes that agree in their first h symbols by the rank of the suffix h positions on
 * (rank = start of its group in SA, read-only during the round), then splits the groups (second phase, after a barrier: no rank is
 * read while another thread rewrites it).  Groups of one suffix are never touched again.  Small groups are sorted by one thread
 * each (dynamic schedule), a group too large for that by all threads together (LSD radix sort, 11-bit digits).
 * Every string ends with a symbol of its own (sais_bwt maps sentinel j to 1 + j), so no comparison runs past a sentinel and the
 * number of rounds is log2 of the longest repeat inside the batch.  A batch that is one long repeat (copies of a genome in one
 * batch) would take many rounds over everything: the caller gives up after PS_MAX_ROUNDS with more than 1/16 unsorted and runs
 * SA-IS, which does not care.
 */
#define PCAT_(a, b) a##b
#define PCAT(a, b) PCAT_(a, b)
#define PFN(name) PCAT(name, PSUF)

typedef struct { PIDX key, idx; } PFN(pel_t);

static int PFN(ps_cmp)(const void *x, const void *y)
{
	const PFN(pel_t) *a = (const PFN(pel_t)*)x, *b = (const PFN(pel_t)*)y;
	return a->key < b->key ? -1 : a->key > b->key ? 1 : 0; /* (ties: the split below only asks which keys are equal) */
}

/* stable LSD radix sort of a[0..n) by key (< maxkey) with all threads of the enclosing team-less context; tmp holds n elements */
static void PFN(ps_radix)(PFN(pel_t) *a, PFN(pel_t) *tmp, size_t n, uint64_t maxkey, int nt)
{
	const int BITS = 11, NB = 1 << BITS;
	int shift, pass = 0;
	size_t *hist = (size_t*)malloc((size_t)nt * NB * sizeof(size_t));
	PFN(pel_t) *src = a, *dst = tmp;
	if (hist == 0) { ps_qsort_fallback(a, n, sizeof(PFN(pel_t)), PFN(ps_cmp)); return; } /* (no memory for the histograms: one thread, comparison sort) */
	for (shift = 0; shift < 64 && (maxkey >> shift) != 0; shift += BITS, ++pass) {
#pragma omp parallel num_threads(nt)
		{
			const int t = omp_get_thread_num(), T = omp_get_num_threads();
			const size_t lo = n * (size_t)t / (size_t)T, hi = n * (size_t)(t + 1) / (size_t)T;
			size_t *h = hist + (size_t)t * NB, i;
			int b, u;
			memset(h, 0, NB * sizeof(size_t));
			for (i = lo; i < hi; ++i) ++h[((uint64_t)src[i].key >> shift) & (NB - 1)];
#pragma omp barrier
#pragma omp single
			{
				size_t sum = 0;
				for (b = 0; b < NB; ++b)
					for (u = 0; u < T; ++u) { const size_t c = hist[(size_t)u * NB + b]; hist[(size_t)u * NB + b] = sum; sum += c; }
			}
			for (i = lo; i < hi; ++i) dst[h[((uint64_t)src[i].key >> shift) & (NB - 1)]++] = src[i];
		}
		{ PFN(pel_t) *x = src; src = dst; dst = x; }
	}
	if (src != a) memcpy(a, src, n * sizeof(PFN(pel_t)));
	free(hist);
	(void)pass;
}

static void PFN(ps_small)(PFN(pel_t) *e, size_t m)
{
	if (m <= 24) { /* insertion sort */
		size_t i, j;
		for (i = 1; i < m; ++i) {
			const PFN(pel_t) x = e[i];
			for (j = i; j > 0 && e[j - 1].key > x.key; --j) e[j] = e[j - 1];
			e[j] = x;
		}
	} else qsort(e, m, sizeof(PFN(pel_t)), PFN(ps_cmp));
}

#define PS_W 8                     /* symbols of the initial key (3 bits each) */
/* the packed first PS_W symbols of suffix i, cut behind its sentinel; *tie = T of that sentinel (1 + its number; 0: none in the window, or the
 * end of the text) */
static inline PIDX PFN(ps_key)(const PIDX *T, size_t n, uint64_t n_seq, size_t i, PIDX *tie)
{
	PIDX key = 0;
	int j;
	*tie = 0;
	for (j = 0; j < PS_W; ++j) {
		const PIDX t = i + (size_t)j < n ? T[i + (size_t)j] : 0;
		if ((uint64_t)t <= n_seq) { *tie = t; break; } /* a sentinel (or the end): code 0, the rest of the key stays 0 */
		key |= (PIDX)((uint64_t)t - n_seq) << (3 * (PS_W - 1 - j));
	}
	return key;
}

#define PS_BIG ((size_t)1 << 16)   /* groups of this many suffixes or more are sorted by all threads together */
#define PS_MAX_ROUNDS 10           /* h = 2^12 symbols compared: give up if more than 1/16 of the suffixes are still unsorted */

/* SA[0..n) of T[0..n) (T[n-1] = 0, unique).  Returns 0, -1 (memory), or 1: gave up (a batch of long repeats), SA undefined. */
static int PFN(ps_main)(const PIDX *T, PIDX *SA, size_t n, uint64_t K, int nt)
{
	PIDX *rank = (PIDX*)malloc(n * sizeof(PIDX)), *gs = 0, *ge = 0, *gs2 = 0, *ge2 = 0;
	PFN(pel_t) *eb = (PFN(pel_t)*)malloc(n * sizeof(PFN(pel_t))), *et = (PFN(pel_t)*)malloc(n * sizeof(PFN(pel_t))), *small = 0;
	uint8_t *bnd = (uint8_t*)malloc(n);
	size_t ng = 0, i, h;
	int ret = 0, round = 0;
	if (!rank || !eb || !et || !bnd) { ret = -1; goto done; }
	/* round 0: by the first PS_W symbols at once (three rounds of doubling saved, each a random read of rank[] per suffix): the key packs
	 * 3 bits per symbol, a sentinel as 0 and nothing behind it; suffixes that reach their sentinel inside the window agree in the key only if
	 * the sentinel sits at the same place, and are then told apart by WHICH sentinel it is (T = 1 + its number): two stable sorts, by that
	 * number first and by the packed key second */
#pragma omp parallel for num_threads(nt) schedule(static)
	for (i = 0; i < n; ++i) { PIDX tie; (void)PFN(ps_key)(T, n, K - 6, i, &tie); eb[i].key = tie, eb[i].idx = (PIDX)i; }
	PFN(ps_radix)(eb, et, n, K - 6 + 2, nt);
#pragma omp parallel for num_threads(nt) schedule(static)
	for (i = 0; i < n; ++i) { PIDX tie; eb[i].key = PFN(ps_key)(T, n, K - 6, eb[i].idx, &tie); }
	PFN(ps_radix)(eb, et, n, (uint64_t)1 << (3 * PS_W), nt);
#pragma omp parallel for num_threads(nt) schedule(static)
	for (i = 0; i < n; ++i) {
		PIDX t0 = 0, t1 = 0;
		SA[i] = eb[i].idx;
		if (i > 0 && eb[i].key == eb[i - 1].key) (void)PFN(ps_key)(T, n, K - 6, eb[i].idx, &t0), (void)PFN(ps_key)(T, n, K - 6, eb[i - 1].idx, &t1);
		bnd[i] = (i == 0 || eb[i].key != eb[i - 1].key || t0 != t1) ? 1 : 0;
	}
	/* groups from the boundary flags: rank = start of the group; the unsorted ones (more than one suffix) are listed */
	gs = (PIDX*)malloc((n / 2 + 1) * sizeof(PIDX)), ge = (PIDX*)malloc((n / 2 + 1) * sizeof(PIDX));
	gs2 = (PIDX*)malloc((n / 2 + 1) * sizeof(PIDX)), ge2 = (PIDX*)malloc((n / 2 + 1) * sizeof(PIDX));
	small = (PFN(pel_t)*)malloc((size_t)nt * PS_BIG * sizeof(PFN(pel_t)));
	if (!gs || !ge || !gs2 || !ge2 || !small) { ret = -1; goto done; }
	{
		size_t a = 0;
		for (i = 1; i <= n; ++i)
			if (i == n || bnd[i]) {
				size_t j;
				for (j = a; j < i; ++j) rank[SA[j]] = (PIDX)a;
				if (i - a > 1) gs[ng] = (PIDX)a, ge[ng] = (PIDX)i, ++ng;
				a = i;
			}
	}
	for (h = PS_W; ng > 0; h <<= 1) {
		size_t unsorted = 0, g, ng2 = 0;
		if (++round > PS_MAX_ROUNDS) {
			for (g = 0; g < ng; ++g) unsorted += (size_t)(ge[g] - gs[g]);
			if (unsorted > n / 16) { ret = 1; goto done; }
		}
		/* phase A: sort every group by the rank h positions on (ranks are read-only here) */
		for (g = 0; g < ng; ++g) { /* the big ones, one after the other, all threads on each */
			const size_t a = gs[g], b = ge[g], m = b - a;
			if (m < PS_BIG) continue;
#pragma omp parallel for num_threads(nt) schedule(static)
			for (i = 0; i < m; ++i) eb[i].idx = SA[a + i], eb[i].key = rank[(size_t)SA[a + i] + h];
			PFN(ps_radix)(eb, et, m, (uint64_t)n, nt);
#pragma omp parallel for num_threads(nt) schedule(static)
			for (i = 0; i < m; ++i) SA[a + i] = eb[i].idx, bnd[a + i] = (i == 0 || eb[i].key != eb[i - 1].key) ? 1 : 0;
		}
#pragma omp parallel for num_threads(nt) schedule(dynamic, 256)
		for (g = 0; g < ng; ++g) {
			const size_t a = gs[g], b = ge[g], m = b - a;
			PFN(pel_t) *e = small + (size_t)omp_get_thread_num() * PS_BIG;
			size_t k;
			if (m >= PS_BIG) continue;
			for (k = 0; k < m; ++k) e[k].idx = SA[a + k], e[k].key = rank[(size_t)SA[a + k] + h];
			PFN(ps_small)(e, m);
			for (k = 0; k < m; ++k) SA[a + k] = e[k].idx, bnd[a + k] = (k == 0 || e[k].key != e[k - 1].key) ? 1 : 0;
		}
		/* phase B: split -- new ranks, and the groups that are still unsorted (per-thread lists, concatenated) */
#pragma omp parallel num_threads(nt)
		{
			size_t cap = 1024, cnt = 0, q;
			PIDX *ls = (PIDX*)malloc(cap * sizeof(PIDX)), *le = (PIDX*)malloc(cap * sizeof(PIDX));
#pragma omp for schedule(dynamic, 256) nowait
			for (g = 0; g < ng; ++g) {
				const size_t a = gs[g], b = ge[g];
				size_t s = a, j;
				for (j = a + 1; j <= b; ++j)
					if (j == b || bnd[j]) {
						size_t k;
						for (k = s; k < j; ++k) rank[SA[k]] = (PIDX)s;
						if (j - s > 1 && ls && le) {
							if (cnt == cap) {
								PIDX *t1 = (PIDX*)realloc(ls, cap * 2 * sizeof(PIDX)), *t2 = t1 ? (PIDX*)realloc(le, cap * 2 * sizeof(PIDX)) : 0;
								if (t1) ls = t1;
								if (t2) le = t2, cap *= 2;
							}
							if (cnt < cap) ls[cnt] = (PIDX)s, le[cnt] = (PIDX)j, ++cnt;
							else {
#pragma omp atomic write
								ret = -1;
							}
						} else if (j - s > 1) {
#pragma omp atomic write
							ret = -1;
						}
						s = j;
					}
			}
#pragma omp critical
			{
				for (q = 0; q < cnt; ++q) gs2[ng2 + q] = ls[q], ge2[ng2 + q] = le[q];
				ng2 += cnt;
			}
			free(ls); free(le);
		}
		if (ret < 0) goto done;
		{ PIDX *x = gs; gs = gs2; gs2 = x; x = ge; ge = ge2; ge2 = x; }
		ng = ng2;
	}
done:
	free(rank); free(eb); free(et); free(bnd); free(gs); free(ge); free(gs2); free(ge2); free(small);
	return ret;
}

/* the counterpart of sais_bwt (sais_core.h): same arguments, same result; 1 = gave up, seq untouched */
static int PFN(ps_bwt)(int64_t n_seq, int64_t len, uint8_t *seq, int64_t ck_step, int64_t *ckrow, int nt)
{
	const size_t n = (size_t)len + 1;
	const uint64_t K = (uint64_t)n_seq + 6;
	PIDX *T = (PIDX*)malloc(n * sizeof(PIDX)), *SA = (PIDX*)malloc(n * sizeof(PIDX));
	size_t i;
	int r;
	if (!T || !SA) { free(T); free(SA); return -1; }
	{ /* sentinel j -> 1 + j (its own symbol), other symbols above all sentinels (sais-ss.c:16-21: the j-th sentinel sorts before the (j+1)-th) */
		PIDX k = 0;
		for (i = 0; i + 1 < n; ++i) T[i] = seq[i] == 0 ? (PIDX)(1 + k++) : (PIDX)((uint64_t)n_seq + seq[i]);
		T[n - 1] = 0;
	}
	r = PFN(ps_main)(T, SA, n, K, nt);
	free(T);
	if (r != 0) { free(SA); return r; }
	if (ck_step > 0)
		for (i = 1; i < n; ++i)
			if ((int64_t)SA[i] % ck_step == 0) ckrow[(int64_t)SA[i] / ck_step] = (int64_t)i - 1;
	/* row i of the BWT is the symbol before suffix SA[i + 1] (SA[0] is the virtual sentinel): into SA's own bytes first, seq is still read */
#pragma omp parallel for num_threads(nt) schedule(static)
	for (i = 1; i < n; ++i) {
		const PIDX p = SA[i];
		SA[i] = p == 0 ? seq[len - 1] : seq[(size_t)p - 1];
	}
#pragma omp parallel for num_threads(nt) schedule(static)
	for (i = 1; i < n; ++i) seq[i - 1] = (uint8_t)SA[i];
	free(SA);
	return 0;
}

#undef PFN
#undef PCAT
#undef PCAT_
#undef PS_BIG
#undef PS_W
#undef PS_MAX_ROUNDS
