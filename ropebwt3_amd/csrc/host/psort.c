/*
 * psort.c -- the parallel host suffix sorter (prefix doubling over OpenMP, psort_core.h) behind rb3h_build_bwt(..., n_threads > 1):
 * what rb3_build_sais gets from libsais + OpenMP (sais-ss.c:15-22, 35-42), written from the published algorithms.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#include <stdio.h>
#include <unistd.h>
#include "rb3host.h"

static void ps_qsort_fallback(void *a, size_t n, size_t sz, int (*cmp)(const void*, const void*)) { qsort(a, n, sz, cmp); }

#define PIDX uint32_t
#define PSUF _32
#include "psort_core.h"
#undef PIDX
#undef PSUF

#define PIDX uint64_t
#define PSUF _64
#include "psort_core.h"
#undef PIDX
#undef PSUF

/* 0: seq holds the BWT (and ckrow the sampled inverse suffix array if ck_step > 0); 1: not sorted (too few threads or symbols, a batch
 * of long repeats, or no memory for what this sorter takes -- text, suffix array, ranks, two boundary bitmaps-as-bytes, a run list and four half-length
 * group lists: ~37 bytes per symbol with 32-bit positions, ~73 with 64-bit ones; checked against the machine's AVAILABLE memory before anything
 * is allocated, because with overcommit malloc succeeds and the process is killed later): the caller runs the sequential SA-IS; < 0: error */
static int64_t ps_mem_available(void)
{
	FILE *fp = fopen("/proc/meminfo", "r");
	char line[256];
	int64_t kb = -1;
	if (fp) {
		while (fgets(line, sizeof(line), fp))
			if (strncmp(line, "MemAvailable:", 13) == 0) { kb = atoll(line + 13); break; }
		fclose(fp);
	}
	if (kb < 0) { /* no /proc: physical memory as an upper bound */
		const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGE_SIZE);
		return pages > 0 && psz > 0 ? (int64_t)pages * psz : INT64_MAX;
	}
	return kb * 1024;
}

int rb3h_psort_bwt(int64_t n_seq, int64_t len, uint8_t *seq, int n_threads, int64_t ck_step, int64_t *ckrow)
{
	int r;
	/* Prefix doubling does more work than SA-IS (a random read of rank[] per unsorted suffix and round), so it needs a few threads to win:
	 * measured on the 8 virtual CPUs of the build container, 8 M symbols (a random genome, both strands) take 0.95 s with SA-IS on one
	 * thread, 1.30 s here with 2 threads, 0.40 s with 4, 0.27 s with 8 (with the initial key of 8 symbols; starting from single symbols
	 * it was 1.0 s with 8).  It takes over from 4 threads up (RB3H_PSORT_MIN_THREADS overrides: tests). */
	const char *e = getenv("RB3H_PSORT_MIN_THREADS");
	const int min_threads = e && atoi(e) > 1 ? atoi(e) : 4;
	if (n_threads > omp_get_num_procs()) n_threads = omp_get_num_procs();
	if (n_threads < min_threads || len < (1 << 16)) return 1;
	/* Batches of SHORT strings stay with SA-IS: overlapping reads agree over up to a read length, every round of doubling then works on most
	 * of the batch in groups of a few suffixes (measured: 600 k reads of 150 bp, 90.6 M symbols: 28 s with 8 threads against 22.8 s for
	 * SA-IS) -- and such batches never come here in the first place (the GPU sorter takes them, cut to size).  What does come here is a
	 * record too long for the GPU sorter: long strings. */
	if (n_seq > 0 && len / n_seq < 1024 && !getenv("RB3H_PSORT_MIN_THREADS")) return 1;
	{ /* decline rather than be killed: what the sort allocates against what the machine can give (RB3H_PSORT_MEM_LIMIT, bytes: tests) */
		const int wide = !((uint64_t)len + 16 < 0xFFFFFFFFull) || getenv("RB3H_PSORT_FORCE64") != 0;
		const double need = (double)len * (wide ? 73.0 : 37.0);
		const char *lim = getenv("RB3H_PSORT_MEM_LIMIT");
		const double have = lim && atoll(lim) > 0 ? (double)atoll(lim) : (double)ps_mem_available();
		if (need > 0.9 * have) {
			if (rb3h_verbose >= 2) fprintf(stderr, "[W::%s] the parallel host sorter would take %.1f GB for %ld symbols, %.1f GB are available: sequential SA-IS instead\n", __func__, need / 1e9, (long)len, have / 1e9);
			return 1;
		}
	}
	r = (uint64_t)len + 16 < 0xFFFFFFFFull && !getenv("RB3H_PSORT_FORCE64") /* (tests: the 64-bit instantiation on a small batch) */ ? ps_bwt_32(n_seq, len, seq, ck_step, ckrow, n_threads) : ps_bwt_64(n_seq, len, seq, ck_step, ckrow, n_threads);
	return r == 0 ? 0 : 1; /* (no memory: SA-IS takes a third of it) */
}
