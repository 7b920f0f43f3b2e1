/*
 * sais_core.h -- textbook SA-IS (Nong, Zhang & Chan 2009: induced sorting of LMS
 * substrings, recursion on the reduced string) over an integer alphabet, written from the
 * published algorithm.  Included twice by sais.c with SIDX = int32_t / int64_t.
 *
 * T[0..n) holds symbols in [0, K); T[n-1] must be 0 and occur nowhere else (explicit unique
 * smallest sentinel).  SA must hold n entries.
 */
#define SCAT_(a, b) a##b
#define SCAT(a, b) SCAT_(a, b)
#define SFN(name) SCAT(name, SSUF)

#define S_EMPTY ((SIDX)-1)
#define T_ISS(i) (tp[(i) >> 3] >> ((i) & 7) & 1)
#define T_SET(i, b) (tp[(i) >> 3] = (b) ? (tp[(i) >> 3] | (uint8_t)(1 << ((i) & 7))) : (tp[(i) >> 3] & (uint8_t)~(1 << ((i) & 7))))
#define T_ISLMS(i) ((i) > 0 && T_ISS(i) && !T_ISS((i) - 1))

static void SFN(sais_buckets)(const SIDX *C, SIDX *B, SIDX K, int end)
{
	SIDX c, sum = 0;
	if (end) for (c = 0; c < K; ++c) sum += C[c], B[c] = sum;
	else for (c = 0; c < K; ++c) B[c] = sum, sum += C[c];
}

static void SFN(sais_induce)(const SIDX *T, SIDX *SA, const SIDX *C, SIDX *B, const uint8_t *tp, SIDX n, SIDX K)
{
	SIDX i, j;
	SFN(sais_buckets)(C, B, K, 0);
	for (i = 0; i < n; ++i) { /* L-type: left to right, fill bucket heads */
		j = SA[i];
		if (j > 0 && !T_ISS(j - 1)) SA[B[T[j - 1]]++] = j - 1;
	}
	SFN(sais_buckets)(C, B, K, 1);
	for (i = n - 1; i >= 0; --i) { /* S-type: right to left, fill bucket tails */
		j = SA[i];
		if (j > 0 && T_ISS(j - 1)) SA[--B[T[j - 1]]] = j - 1;
	}
}

static int SFN(sais_main)(const SIDX *T, SIDX *SA, SIDX n, SIDX K)
{
	SIDX i, j, m, n_names, *C, *B;
	uint8_t *tp;
	if (n == 1) { SA[0] = 0; return 0; }
	C = (SIDX*)calloc((size_t)K, sizeof(SIDX));
	B = (SIDX*)malloc((size_t)K * sizeof(SIDX));
	tp = (uint8_t*)calloc((size_t)(n >> 3) + 1, 1);
	if (!C || !B || !tp) { free(C); free(B); free(tp); return -1; }
	for (i = 0; i < n; ++i) ++C[T[i]];
	/* classify: S-type (1) or L-type (0) */
	T_SET(n - 1, 1);
	for (i = n - 2; i >= 0; --i)
		T_SET(i, (T[i] < T[i + 1] || (T[i] == T[i + 1] && T_ISS(i + 1))) ? 1 : 0);
	/* stage 1: sort the LMS substrings */
	SFN(sais_buckets)(C, B, K, 1);
	for (i = 0; i < n; ++i) SA[i] = S_EMPTY;
	for (i = 1; i < n; ++i)
		if (T_ISLMS(i)) SA[--B[T[i]]] = i;
	SFN(sais_induce)(T, SA, C, B, tp, n, K);
	for (i = 0, m = 0; i < n; ++i)
		if (T_ISLMS(SA[i])) SA[m++] = SA[i];
	for (i = m; i < n; ++i) SA[i] = S_EMPTY;
	{ /* name them */
		SIDX prev = -1;
		n_names = 0;
		for (i = 0; i < m; ++i) {
			SIDX pos = SA[i], d;
			int diff = 0;
			if (prev < 0) diff = 1;
			else for (d = 0;; ++d) {
				if (T[pos + d] != T[prev + d] || T_ISS(pos + d) != T_ISS(prev + d)) { diff = 1; break; }
				if (d > 0 && (T_ISLMS(pos + d) || T_ISLMS(prev + d))) break; /* both end here: equal */
			}
			if (diff) ++n_names, prev = pos;
			SA[m + (pos >> 1)] = n_names - 1;
		}
		for (i = n - 1, j = n - 1; i >= m; --i)
			if (SA[i] != S_EMPTY) SA[j--] = SA[i];
	}
	{ /* stage 2: order of the LMS suffixes from the reduced string */
		SIDX *SA1 = SA, *T1 = SA + n - m;
		if (n_names < m) {
			free(B); free(C); /* give the memory back during the recursion */
			if (SFN(sais_main)(T1, SA1, m, n_names) < 0) { free(tp); return -1; }
			C = (SIDX*)calloc((size_t)K, sizeof(SIDX));
			B = (SIDX*)malloc((size_t)K * sizeof(SIDX));
			if (!C || !B) { free(C); free(B); free(tp); return -1; }
			for (i = 0; i < n; ++i) ++C[T[i]];
		} else for (i = 0; i < m; ++i) SA1[T1[i]] = i;
		/* stage 3: induce the full suffix array */
		for (i = 1, j = 0; i < n; ++i)
			if (T_ISLMS(i)) T1[j++] = i; /* T1 now lists LMS positions in text order */
		for (i = 0; i < m; ++i) SA1[i] = T1[SA1[i]];
		for (i = m; i < n; ++i) SA[i] = S_EMPTY;
		SFN(sais_buckets)(C, B, K, 1);
		for (i = m - 1; i >= 0; --i) {
			j = SA[i], SA[i] = S_EMPTY;
			SA[--B[T[j]]] = j;
		}
		SFN(sais_induce)(T, SA, C, B, tp, n, K);
	}
	free(C); free(B); free(tp);
	return 0;
}

/* seq[0..len): symbols 0..5, strings 0-terminated, seq[len-1] == 0.  Replaced in place by the
 * BWT in generalised-suffix-array order (the j-th 0 sorts before the (j+1)-th).  If ck_step > 0,
 * ckrow[p / ck_step] receives the BWT row of the suffix starting at every text position p that is
 * a multiple of ck_step (a sampled inverse suffix array; ckrow must hold len / ck_step + 1). */
static int SFN(sais_bwt)(int64_t n_seq, int64_t len, uint8_t *seq, int64_t ck_step, int64_t *ckrow)
{
	SIDX n = (SIDX)len + 1, K = (SIDX)n_seq + 6, i, k = 0;
	SIDX *T = (SIDX*)malloc((size_t)n * sizeof(SIDX));
	SIDX *SA = (SIDX*)malloc((size_t)n * sizeof(SIDX));
	if (!T || !SA) { free(T); free(SA); return -1; }
	for (i = 0; i < n - 1; ++i)
		T[i] = seq[i] == 0 ? 1 + k++ : (SIDX)n_seq + seq[i];
	T[n - 1] = 0;
	if (SFN(sais_main)(T, SA, n, K) < 0) { free(T); free(SA); return -1; }
	free(T);
	/* SA[0] = n-1 is the virtual sentinel; row i of the BWT is SA[i+1] (sais-ss.c:23-26) */
	if (ck_step > 0)
		for (i = 1; i < n; ++i)
			if (SA[i] % ck_step == 0) ckrow[SA[i] / ck_step] = (int64_t)i - 1;
	for (i = 1; i < n; ++i) {
		SIDX p = SA[i];
		SA[i] = p == 0 ? seq[len - 1] : seq[p - 1];
	}
	for (i = 1; i < n; ++i) seq[i - 1] = (uint8_t)SA[i];
	free(SA);
	return 0;
}

#undef SFN
#undef SCAT
#undef SCAT_
#undef S_EMPTY
#undef T_ISS
#undef T_SET
#undef T_ISLMS
