/*
 * sais.c -- host suffix sorter for one batch: text of 0-terminated nt6 strings -> BWT.
 *
 * Replaces rb3_build_sais (sais-ss.c:50-56), which calls the vendored third-party libsais
 * in GSA mode.  The multi-string BWT is uniquely defined by the order "sentinel j < sentinel
 * j+1 < A < C < G < T < N", so any correct suffix sorter yields the same bytes; this one is a
 * from-scratch SA-IS (see sais_core.h).  The partial BWT stays on the host by design
 * (north_star); it is not part of the timed merge path.
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

int rb3h_build_bwt(int64_t n_seq, int64_t len, uint8_t *seq, int n_threads)
{
	int64_t i, k = 0;
	(void)n_threads;
	if (len <= 0 || seq[len - 1] != 0) return -1;
	for (i = 0; i < len; ++i) {
		if (seq[i] > 5) return -1;
		k += (seq[i] == 0);
	}
	if (n_seq <= 0) n_seq = k;
	if (k != n_seq) return -1;
	for (i = 1; i < len; ++i) /* empty strings are out of contract */
		if (seq[i] == 0 && seq[i - 1] == 0) return -2;
	if (seq[0] == 0) return -2;
	if (len + n_seq + 16 < INT32_MAX) return sais_bwt_32(n_seq, len, seq); /* sais-ss.c:52 picks 32/64 bit the same way */
	return sais_bwt_64(n_seq, len, seq);
}
