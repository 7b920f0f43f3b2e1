/*
 * sais.c -- host suffix sorter for one batch: text of 0-terminated nt6 strings -> BWT.
 *
 * Replaces rb3_build_sais (sais-ss.c:50-56), which calls the vendored third-party libsais
 * in GSA mode.  The multi-string BWT is uniquely defined by the order "sentinel j < sentinel
 * j+1 < A < C < G < T < N", so any correct suffix sorter yields the same bytes; this one is a
 * from-scratch SA-IS (see sais_core.h).  The partial BWT stays on the host by design
 * (north_star); it is not part of the timed merge path.
 *
 * SA-IS is SEQUENTIAL.  With n_threads >= 4 (rb3_build_sais(n_seq, len, seq, n_threads): libsais + OpenMP, sais-ss.c:15-22) the batch
 * goes to the parallel prefix-doubling sorter of psort.c first; SA-IS takes what that one declines (small batches, a batch of long
 * repeats, no room for its ~37 bytes per symbol, ~73 with 64-bit positions: checked against the available memory up front).  The default path does not come here at all: batches are cut to fit the GPU
 * sorter (main.c); this is for `--host-sort`, for a record of 2^31 symbols or more, and for a device without room.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "rb3host.h"

#define SIDX int32_t
#define SSUF _32
#include "sais_core.h"
#undef SIDX
#undef SSUF

#define SIDX int64_t
#define SSUF _64
#include "sais_core.h"
#undef SIDX
#undef SSUF

static int check_text(int64_t *n_seq, int64_t len, const uint8_t *seq)
{
	int64_t i, k = 0;
	if (len <= 0 || seq[len - 1] != 0) return -1;
	for (i = 0; i < len; ++i) {
		if (seq[i] > 5) return -1;
		k += (seq[i] == 0);
	}
	if (*n_seq <= 0) *n_seq = k;
	if (k != *n_seq) return -1;
	for (i = 1; i < len; ++i) /* empty strings are out of contract */
		if (seq[i] == 0 && seq[i - 1] == 0) return -2;
	if (seq[0] == 0) return -2;
	return 0;
}

int rb3h_psort_bwt(int64_t n_seq, int64_t len, uint8_t *seq, int n_threads, int64_t ck_step, int64_t *ckrow); /* psort.c */

int rb3h_build_bwt(int64_t n_seq, int64_t len, uint8_t *seq, int n_threads)
{
	int r;
	if ((r = check_text(&n_seq, len, seq)) < 0) return r;
	if ((r = rb3h_psort_bwt(n_seq, len, seq, n_threads, 0, 0)) <= 0) return r; /* (1: declined, the text is untouched) */
	if (len + n_seq + 16 < INT32_MAX) return sais_bwt_32(n_seq, len, seq, 0, 0); /* sais-ss.c:52 picks 32/64 bit the same way */
	return sais_bwt_64(n_seq, len, seq, 0, 0);
}

#define RB3H_MIN_SEG 128
#define RB3H_PROBE 64 /* a walker looks that many positions to the right of its start row for a record of its right neighbour.  (24 until the end
                       * of round 3: a walker that starts late runs its first iterations slower than the neighbour that follows it -- cold caches,
                       * a crowded GPU --, so the neighbour, 10-15 rows behind, reached the walker's first rows before their records, parked for up to
                       * 8 of the WALKER's iterations, had been written: it recorded over them and settled the walker's second stretch instead of the
                       * first, which the per-walker settle pass does not resolve -- a merge in ten thousand was redone.  With 64 a late walker only
                       * starts if the neighbour is at least ~55 rows behind.) */
#ifndef RB3H_PREROLL
#define RB3H_PREROLL 16 /* = RB3_TENT_MIN_AGE of the engine (rb3gpu_kernels.h) */
#endif

/* The walker list of a batch from its sampled inverse suffix array: ckrow[i] = row of the suffix starting at text
 * position i * step (from the host sorter below, or from rb3gpu_bwt_from_text).  text is the batch BEFORE it is
 * turned into a BWT.  Same list as rb3h_build_bwt_walkers.  ckrow == NULL: the walkers are given by TEXT POSITION instead of
 * row (a sentinel's walker: the position of the sentinel), for rb3gpu_merge_text_dev. */
int rb3h_walkers_from_ckrow(int64_t len, const uint8_t *text, int64_t step, const int64_t *ckrow, int64_t *n_walkers, rb3h_walker_t **walkers)
{
	int64_t b, j, nw = 0, cap;
	rb3h_walker_t *w;
	*n_walkers = 0, *walkers = 0;
	if (step < 2 || len <= 0 || text[len - 1] != 0) return -3;
	/* (one pass over the text, at memchr speed: the list grows with the strings found -- counting the sentinels first was a second pass
	 * over 8.8 MB per genome, a third of the 0.6 ms a list took) */
	cap = len / step + 1024;
	w = (rb3h_walker_t*)malloc((size_t)cap * sizeof(rb3h_walker_t));
	if (!w) return -1;
	for (j = 0, b = 0; b < len; ++j) {
		const int64_t e = (const uint8_t*)memchr(text + b, 0, (size_t)(len - b)) - text; /* string j occupies [b, e), sentinel at e */
		int64_t prev = -1, p;
		if (nw + (e - b) / step + 2 > cap) {
			rb3h_walker_t *w2;
			cap = cap * 2 + (e - b) / step + 2;
			w2 = (rb3h_walker_t*)realloc(w, (size_t)cap * sizeof(rb3h_walker_t));
			if (!w2) { free(w); return -1; }
			w = w2;
		}
		for (p = (b / step + 1) * step; p < e; p += step) { /* multiples of step strictly inside the string */
			/* By text position (no ckrow) a walker starts RB3H_PREROLL positions to the right of its segment: it cannot record before
			 * it is that many steps old anyway (k_chain, RB3_TENT_MIN_AGE), so it spends its youth on rows its right neighbour owns
			 * and reaches its own first row old enough to record it.  Otherwise the left neighbour's first rows stay unrecorded and
			 * this walker has to go on behind the end of its segment until it meets a record -- the same number of steps, but through
			 * the kernel's general step, and a few more until the record it runs into has become visible. */
			int64_t pre = ckrow ? 0 : e - 1 - p < RB3H_PREROLL ? e - 1 - p : RB3H_PREROLL;
			if (e - p < RB3H_MIN_SEG && e - p < step) continue; /* keep the sentinel walker's own segment long (see k_chain) */
			w[nw].row = ckrow ? ckrow[p / step] : p + pre, w[nw].ka0 = -1, w[nw].flags = pre << 8; /* (flags >> 8 & 255: the part of nsteps outside the segment, for whoever thins the list) */
			if (!ckrow && pre == RB3H_PREROLL && e - 1 - (p + pre) >= RB3H_PROBE) w[nw].flags |= (int64_t)RB3H_PROBE << 16; /* (flags >> 16: see k_chain, "a walker that starts late") */
			w[nw].nsteps = prev < 0 ? INT64_MAX / 2 : p - prev + pre;
			prev = p, ++nw;
		}
		w[nw].row = ckrow ? j : e, w[nw].ka0 = -2 /* sentinel row: exact, = acc[1] of the index */, w[nw].flags = 0;
		w[nw].nsteps = prev < 0 ? INT64_MAX / 2 : e - prev;
		++nw, b = e + 1;
	}
	*n_walkers = nw, *walkers = w;
	return 0;
}

int rb3h_walkers_text(int64_t len, const uint8_t *text, int64_t step, int64_t *n_walkers, rb3h_walker_t **walkers)
{
	return rb3h_walkers_from_ckrow(len, text, step, 0, n_walkers, walkers);
}

/* BWT plus the list of LF walkers for the GPU merge: one per string (its sentinel row) and one
 * at every text position that is a multiple of `step` strictly inside a string, in text order. */
int rb3h_build_bwt_walkers(int64_t n_seq, int64_t len, uint8_t *seq, int n_threads, int64_t step, int64_t *n_walkers, rb3h_walker_t **walkers)
{
	int r;
	int64_t *ckrow, i, b, j, nw = 0, mw;
	rb3h_walker_t *w;
	uint8_t *isend; /* 1 bit per checkpoint slot: position is a sentinel or the first symbol of a string */
	*n_walkers = 0, *walkers = 0;
	if (step < 2) return -3;
	if ((r = check_text(&n_seq, len, seq)) < 0) return r;
	ckrow = (int64_t*)malloc((size_t)(len / step + 2) * 8);
	isend = (uint8_t*)calloc((size_t)(len / step + 2), 1);
	mw = n_seq + len / step + 2;
	w = (rb3h_walker_t*)malloc((size_t)mw * sizeof(rb3h_walker_t));
	if (!ckrow || !isend || !w) { free(ckrow); free(isend); free(w); return -1; }
	for (i = 0; i < len; i += step)
		if (seq[i] == 0 || i == 0 || seq[i - 1] == 0) isend[i / step] = 1;
	/* string boundaries must be read before the text is overwritten */
	{
		int64_t *ends = (int64_t*)malloc((size_t)n_seq * 8);
		if (!ends) { free(ckrow); free(isend); free(w); return -1; }
		for (i = 0, j = 0; i < len; ++i) if (seq[i] == 0) ends[j++] = i;
		if ((r = rb3h_psort_bwt(n_seq, len, seq, n_threads, step, ckrow)) > 0) /* (declined: sequential) */
			r = len + n_seq + 16 < INT32_MAX ? sais_bwt_32(n_seq, len, seq, step, ckrow) : sais_bwt_64(n_seq, len, seq, step, ckrow);
		if (r < 0) { free(ends); free(ckrow); free(isend); free(w); return r; }
		for (j = 0, b = 0; j < n_seq; ++j) { /* string j occupies [b, e), sentinel at e */
			const int64_t e = ends[j];
			int64_t prev = -1, p = (b / step + 1) * step; /* first multiple of step > b */
			for (; p < e; p += step) {
				if (isend[p / step]) continue;
				if (e - p < RB3H_MIN_SEG && e - p < step) continue; /* keep the sentinel walker's own segment long (see k_chain) */
				w[nw].row = ckrow[p / step], w[nw].ka0 = -1, w[nw].flags = 0;
				w[nw].nsteps = prev < 0 ? INT64_MAX / 2 : p - prev;
				prev = p, ++nw;
			}
			w[nw].row = j, w[nw].ka0 = -2 /* sentinel row: exact, = acc[1] of the index */, w[nw].flags = 0;
			w[nw].nsteps = prev < 0 ? INT64_MAX / 2 : e - prev;
			++nw;
			b = e + 1;
		}
		free(ends);
	}
	free(ckrow); free(isend);
	*n_walkers = nw, *walkers = w;
	return 0;
}
