/*
 * fmd.c -- writer and reader of the FMD ("RLD\3") on-disk format, the bit-exactness contract
 * of `ropebwt3 build -d`.  Written from the format; the functions it
 * has to agree with byte for byte are rld_enc / rld_enc1 / enc_next_block / rld_enc_finish /
 * rld_rank_index / rld_dump (rld0.c:107-243) and rld_dec0 / rld_dec (rld0.h:85-122).
 *
 * Layout: 64-bit words grouped in 64-byte blocks (8 words); a block starts with the counts
 * {total, $, A, C, G, T, N} of the PREVIOUS block as 7 x u16 / u32 / u64 (block type 0/1/2 in the
 * top two bits of the first word), followed by Elias-delta coded runs packed MSB first:
 * delta(len) << 3 | sym.  A code never straddles two blocks.  Blocks live in superblocks of
 * 2^23 words; the last word of a superblock is never used.  Here the words are kept in one
 * flat growing array; only the logical word stream matters.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "rb3host.h"

#define FMD_ASIZE   6
#define FMD_ASIZE1  7
#define FMD_ABITS   3            /* bits per symbol: ilog2(6) + 1 */
#define FMD_SBITS   3            /* log2(words per block) */
#define FMD_SSIZE   8
#define FMD_LBITS   23           /* log2(words per superblock), rld0.h:11 */
#define FMD_LSIZE   (1LL << FMD_LBITS)
#define FMD_IBITS_PLUS 4

static const int fmd_offset0[3] = { 2, 4, 7 }; /* header words per block type, rld0.c:71-73 */

static inline int fmd_ilog2(uint64_t v) /* floor(log2 v); -1 for 0 like the reference's table */
{
	return v == 0 ? -1 : 63 - __builtin_clzll(v);
}

struct rb3h_fmdw_s {
	uint64_t *z;                 /* the word stream */
	int64_t m;                   /* allocated words */
	int64_t head;                /* first word of the current block */
	int64_t p;                   /* current word */
	int r;                       /* unused bits left in word p */
	int pc;                      /* pending symbol, -1 if none */
	int64_t pl;                  /* pending run length */
	uint64_t cnt[FMD_ASIZE1], mcnt[FMD_ASIZE1];
	uint64_t n_bytes, n_frames, *frame;
	int finished;
};

static inline int64_t fmd_stail(int64_t head) /* last usable word of the block at `head` */
{
	return head + FMD_SSIZE - (((head + FMD_SSIZE) & (FMD_LSIZE - 1)) == 0 ? 2 : 1);
}

static int fmdw_reserve(rb3h_fmdw_t *w, int64_t upto)
{
	if (upto < w->m) return 0;
	int64_t m = w->m;
	while (m <= upto) m += m < (1LL << 27) ? m : (1LL << 27);
	uint64_t *z = (uint64_t*)realloc(w->z, (size_t)m * 8);
	if (z == 0) return -1;
	memset(z + w->m, 0, (size_t)(m - w->m) * 8);
	w->z = z, w->m = m;
	return 0;
}

rb3h_fmdw_t *rb3h_fmdw_init(void)
{
	rb3h_fmdw_t *w = (rb3h_fmdw_t*)calloc(1, sizeof(*w));
	if (w == 0) return 0;
	w->m = 1 << 16;
	w->z = (uint64_t*)calloc((size_t)w->m, 8);
	if (w->z == 0) { free(w); return 0; }
	w->head = 0, w->p = fmd_offset0[0], w->r = 64, w->pc = -1, w->pl = 0;
	return w;
}

void rb3h_fmdw_destroy(rb3h_fmdw_t *w)
{
	if (w == 0) return;
	free(w->z); free(w->frame); free(w);
}

/* start the next block: its header records what the block just finished contained */
static int fmdw_next_block(rb3h_fmdw_t *w)
{
	int i, type;
	uint64_t d0 = w->cnt[0] - w->mcnt[0];
	w->head += FMD_SSIZE;
	if (fmdw_reserve(w, w->head + FMD_SSIZE) < 0) return -1;
	if (d0 < 0x4000) {
		uint16_t *q = (uint16_t*)(w->z + w->head);
		for (i = 0; i < FMD_ASIZE1; ++i) q[i] = (uint16_t)(w->cnt[i] - w->mcnt[i]);
		type = 0;
	} else if (d0 < 0x40000000) {
		uint32_t *q = (uint32_t*)(w->z + w->head);
		for (i = 0; i < FMD_ASIZE1; ++i) q[i] = (uint32_t)(w->cnt[i] - w->mcnt[i]);
		type = 1;
	} else {
		uint64_t *q = w->z + w->head;
		for (i = 0; i < FMD_ASIZE1; ++i) q[i] = w->cnt[i] - w->mcnt[i];
		type = 2;
	}
	w->z[w->head] |= (uint64_t)type << 62;
	w->p = w->head + fmd_offset0[type];
	w->r = 64;
	memcpy(w->mcnt, w->cnt, sizeof(w->cnt));
	return 0;
}

/* emit one maximal run */
static int fmdw_enc1(rb3h_fmdw_t *w, int64_t l, int c)
{
	const int y = fmd_ilog2((uint64_t)l), zz = fmd_ilog2((uint64_t)y + 1);
	int width = (zz << 1) + 1 + y + FMD_ABITS;
	const uint64_t delta = ((uint64_t)l ^ (uint64_t)1 << y) | (uint64_t)(y + 1) << y;
	const uint64_t x = delta << FMD_ABITS | (uint64_t)c;
	if (width >= w->r && w->p == fmd_stail(w->head))
		if (fmdw_next_block(w) < 0) return -1;
	if (width > w->r) { /* straddles two words of the same block */
		width -= w->r;
		w->z[w->p++] |= width < 64 ? x >> width : 0;
		w->r = 64 - width;
		w->z[w->p] = x << w->r;
	} else {
		w->r -= width;
		w->z[w->p] |= x << w->r;
	}
	w->cnt[0] += (uint64_t)l, w->cnt[c + 1] += (uint64_t)l;
	return 0;
}

int rb3h_fmdw_enc(rb3h_fmdw_t *w, int64_t l, int c)
{
	if (l == 0) return 0;
	if (l < 0 || c < 0 || c >= FMD_ASIZE || w->finished) return -1;
	if (w->pc != c) {
		if (w->pl && fmdw_enc1(w, w->pl, w->pc) < 0) return -1;
		w->pl = l, w->pc = c;
	} else w->pl += l;
	return 0;
}

/* bulk form: words[i] = start << 3 | sym of maximal runs in order (rb3gpu_export_run_words); end >= 0 closes the
 * last run.  Not to be mixed with rb3h_fmdw_enc on the same writer. */
int rb3h_fmdw_enc_words(rb3h_fmdw_t *w, int64_t n, const uint64_t *words, int64_t end)
{
	int64_t i;
	if (w->finished) return -1;
	for (i = 0; i < n; ++i) {
		const int64_t s = (int64_t)(words[i] >> 3);
		const int c = (int)(words[i] & 7);
		if (c >= FMD_ASIZE) return -1;
		if (w->pc >= 0) { /* pl holds the START of the pending run here */
			if (s <= w->pl || fmdw_enc1(w, s - w->pl, w->pc) < 0) return -1;
		}
		w->pl = s, w->pc = c;
	}
	if (end >= 0 && w->pc >= 0) {
		if (end <= w->pl || fmdw_enc1(w, end - w->pl, w->pc) < 0) return -1;
		w->pc = -1, w->pl = 0;
	}
	return 0;
}

static int fmdw_rank_index(rb3h_fmdw_t *w)
{
	const uint64_t n_blks = w->n_bytes * 8 / 64 / FMD_SSIZE + 1;
	const int64_t last = (int64_t)(w->n_bytes >> 3 >> FMD_SBITS << FMD_SBITS);
	const int ibits = fmd_ilog2(w->mcnt[0] / n_blks) + FMD_IBITS_PLUS;
	uint64_t k, cnt[FMD_ASIZE];
	int64_t i;
	int j;
	w->n_frames = ((w->mcnt[0] + (1ULL << ibits) - 1) >> ibits) + 1;
	w->frame = (uint64_t*)calloc((size_t)w->n_frames * FMD_ASIZE1, 8);
	if (w->frame == 0) return -1;
	memset(cnt, 0, sizeof(cnt));
	for (i = FMD_SSIZE, k = 1; i <= last; i += FMD_SSIZE) {
		const uint64_t *p = w->z + i;
		const int type = (int)(*p >> 62);
		uint64_t sum;
		if (type == 0) {
			const uint16_t *q = (const uint16_t*)p;
			for (j = 1; j <= FMD_ASIZE; ++j) cnt[j-1] += q[j];
		} else if (type == 1) {
			const uint32_t *q = (const uint32_t*)p;
			for (j = 1; j <= FMD_ASIZE; ++j) cnt[j-1] += q[j] & 0x3fffffff;
		} else {
			for (j = 1; j <= FMD_ASIZE; ++j) cnt[j-1] += p[j];
		}
		for (j = 0, sum = 0; j < FMD_ASIZE; ++j) sum += cnt[j];
		while (sum >= k << ibits) ++k;
		if (k < w->n_frames) {
			uint64_t *f = w->frame + k * FMD_ASIZE1;
			f[0] = (uint64_t)i;
			for (j = 0; j < FMD_ASIZE; ++j) f[j + 1] = cnt[j];
		}
	}
	for (k = 1; k < w->n_frames; ++k) { /* frames nobody wrote inherit their predecessor */
		uint64_t *f = w->frame + k * FMD_ASIZE1;
		if (f[0] == 0) memcpy(f, f - FMD_ASIZE1, FMD_ASIZE1 * 8);
	}
	return 0;
}

int rb3h_fmdw_finish(rb3h_fmdw_t *w)
{
	int i;
	if (w->finished) return 0;
	if (w->pl && fmdw_enc1(w, w->pl, w->pc) < 0) return -1;
	w->pl = 0, w->pc = -1;
	if (fmdw_next_block(w) < 0) return -1; /* trailing header-only block */
	w->n_bytes = (uint64_t)w->p * 8;
	for (w->cnt[0] = 0, i = 1; i <= FMD_ASIZE; ++i) w->cnt[i] += w->cnt[i - 1];
	if (fmdw_rank_index(w) < 0) return -1;
	w->finished = 1;
	return 0;
}

/* take over a data section that was packed elsewhere (rb3gpu_export_fmd_words): `words` is malloc'ed, n_words words incl.
 * the trailing header-only block; acc[] is the C array of the BWT (acc[6] = its length).  Builds the rank index; the
 * writer is finished afterwards.  On failure the array still belongs to the caller. */
int rb3h_fmdw_adopt(rb3h_fmdw_t *w, uint64_t *words, int64_t n_words, const int64_t acc[7])
{
	int i;
	if (w->finished || words == 0 || n_words < 2) return -1;
	free(w->z);
	w->z = words, w->m = n_words;
	w->n_bytes = (uint64_t)n_words * 8;
	w->mcnt[0] = (uint64_t)acc[6];
	for (i = 1; i <= FMD_ASIZE; ++i) w->mcnt[i] = (uint64_t)(acc[i] - acc[i - 1]);
	memcpy(w->cnt, w->mcnt, sizeof(w->cnt));
	for (w->cnt[0] = 0, i = 1; i <= FMD_ASIZE; ++i) w->cnt[i] += w->cnt[i - 1];
	if (fmdw_rank_index(w) < 0) { w->z = 0, w->m = 0; return -1; } /* not taken over: on failure `words` stays with the caller */
	w->finished = 1;
	return 0;
}

int64_t rb3h_fmdw_nbytes(const rb3h_fmdw_t *w) { return (int64_t)w->n_bytes; }

int rb3h_fmdw_dump(const rb3h_fmdw_t *w, FILE *fp)
{
	uint64_t k = 0;
	const uint32_t a = FMD_ASIZE << 16 | FMD_SBITS;
	if (!w->finished) return -1;
	if (fwrite("RLD\3", 1, 4, fp) != 4) return -1;
	fwrite(&a, 4, 1, fp);
	fwrite(&k, 8, 1, fp);
	fwrite(&w->n_bytes, 8, 1, fp);
	fwrite(&w->n_frames, 8, 1, fp);
	fwrite(w->mcnt + 1, 8, FMD_ASIZE, fp);
	if (fwrite(w->z, 8, (size_t)(w->n_bytes / 8), fp) != (size_t)(w->n_bytes / 8)) return -1;
	if (fwrite(w->frame, 8 * FMD_ASIZE1, (size_t)w->n_frames, fp) != (size_t)w->n_frames) return -1;
	return fflush(fp) == 0 ? 0 : -1;
}

int rb3h_fmdw_dump_file(const rb3h_fmdw_t *w, const char *fn) /* rld_dump(e, fn) */
{
	FILE *fp = fopen(fn, "wb");
	int r;
	if (fp == 0) return -1;
	r = rb3h_fmdw_dump(w, fp);
	if (fclose(fp) != 0) r = -1;
	return r;
}

/* ------------------------------------------------------------------------------------- */
/* reader                                                                                */
/* ------------------------------------------------------------------------------------- */

/* the word stream of an FMD file, undecoded (for rb3gpu_from_fmd_words): 0 = ok (*z: malloc'ed, n_words words followed by two
 * zero words), 1 = not an FMD file of the DNA flavour, < 0 = error */
int rb3h_fmd_read_words(const char *fn, uint64_t **z_out, int64_t *n_words_out, int64_t mcnt_out[6])
{
	FILE *fp;
	char magic[4];
	uint32_t a;
	uint64_t hdr[3], mc[FMD_ASIZE], *z;
	int64_t n_words;
	int i;
	*z_out = 0, *n_words_out = 0;
	if (strcmp(fn, "-") == 0 || (fp = fopen(fn, "rb")) == 0) return 1; /* (a stream cannot be read twice: leave it to the run reader) */
	if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, "RLD\3", 4) != 0 || fread(&a, 4, 1, fp) != 1 || (int)(a >> 16) != FMD_ASIZE || (int)(a & 0xffff) != FMD_SBITS) { fclose(fp); return 1; }
	if (fread(hdr, 8, 3, fp) != 3 || fread(mc, 8, FMD_ASIZE, fp) != FMD_ASIZE) { fclose(fp); return -1; }
	n_words = (int64_t)(hdr[1] / 8);
	if (n_words < FMD_SSIZE || (z = (uint64_t*)malloc((size_t)(n_words + 2) * 8)) == 0) { fclose(fp); return -1; }
	if (fread(z, 8, (size_t)n_words, fp) != (size_t)n_words) { free(z); fclose(fp); return -1; }
	fclose(fp);
	z[n_words] = z[n_words + 1] = 0;
	for (i = 0; i < FMD_ASIZE; ++i) mcnt_out[i] = (int64_t)mc[i];
	*z_out = z, *n_words_out = n_words;
	return 0;
}

int rb3h_fmd_read_runs(FILE *fp, rb3h_run_f emit, void *data, int64_t mcnt_out[6])
{
	uint32_t a;
	uint64_t hdr[3], mc[FMD_ASIZE], *z;
	int64_t n_words, head, last;
	int asize, sbits, ssize, pc = -1, ret = 0;
	int64_t pl = 0;
	/* the 4-byte magic has been consumed by the caller */
	if (fread(&a, 4, 1, fp) != 1) return -1;
	asize = a >> 16, sbits = a & 0xffff;
	if (asize != FMD_ASIZE || sbits != FMD_SBITS) return -2; /* only the DNA flavour ropebwt3 writes */
	ssize = 1 << sbits;
	if (fread(hdr, 8, 3, fp) != 3) return -1;
	if (fread(mc, 8, FMD_ASIZE, fp) != FMD_ASIZE) return -1;
	n_words = (int64_t)(hdr[1] / 8);
	z = (uint64_t*)malloc((size_t)(n_words + 2) * 8);
	if (z == 0) return -1;
	if (fread(z, 8, (size_t)n_words, fp) != (size_t)n_words) { free(z); return -1; }
	z[n_words] = z[n_words + 1] = 0;
	if (mcnt_out) { int i; for (i = 0; i < FMD_ASIZE; ++i) mcnt_out[i] = (int64_t)mc[i]; }
	last = n_words >> sbits << sbits; /* the trailing header-only block */
	for (head = 0; head < last && ret == 0; head += ssize) {
		const int type = (int)(z[head] >> 62);
		const int64_t stail = fmd_stail(head);
		int64_t p = head + fmd_offset0[type];
		int r = 64;
		while (p <= stail) {
			/* 64-bit window, MSB first, zero-filled past the block's last usable word */
			uint64_t x = z[p] << (64 - r);
			if (r != 64 && p != stail) x |= z[p + 1] >> r;
			int lz, wd, y, c;
			int64_t l;
			if (x == 0) break;
			lz = __builtin_clzll(x);
			if (lz >= 6) break; /* no delta code starts with six zeros: end of block */
			wd = 2 * lz + 1;
			y = (int)(x >> (64 - wd)) - 1;
			l = (int64_t)1 << y;
			if (y > 0) l |= (int64_t)(x << wd >> (64 - y));
			wd += y;
			c = (int)(x << wd >> (64 - FMD_ABITS));
			wd += FMD_ABITS;
			if (r > wd) r -= wd; else ++p, r = 64 + r - wd;
			if (c >= FMD_ASIZE) { ret = -3; break; }
			if (c == pc) pl += l;
			else {
				if (pl > 0 && emit(data, pc, pl) != 0) { ret = -4; break; }
				pc = c, pl = l;
			}
		}
	}
	if (ret == 0 && pl > 0 && emit(data, pc, pl) != 0) ret = -4;
	free(z);
	return ret;
}
