/*
 * fmr.c -- writer and reader of the FMR ("RB\2") format: six B+-trees ("ropes"), rope a
 * holding the BWT rows whose suffix starts with symbol a (mrope.h:10-14).  The format is
 * defined by mr_dump / rope_dump / rope_dump_node (mrope.c:152-159, rope.c:265-287) and the
 * leaf byte codec rle_enc1 / rle_dec1 (rle.h:39-75); it is not canonical (tree shape is free)
 * but must load in the reference (rope_restore, rope.c:289-330): n <= max_nodes per node,
 * leaf bytes + 2 <= block_len, and room left in every leaf for a later insertion
 * (RLE_MIN_SPACE, rle.h:36; rope.c:143).
 *
 * The engine has no tree: the writer takes the ordered run stream exported from the GPU,
 * cuts it at the six bucket boundaries, packs runs into leaves and builds the internal
 * levels bottom-up.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "rb3host.h"

#define FMR_MIN_SPACE 18 /* RLE_MIN_SPACE */

typedef struct {
	int64_t c[6];
	uint16_t nbytes;
	int64_t off; /* offset of the codes in the rope's byte pool */
} fmr_leaf_t;

typedef struct {
	fmr_leaf_t *leaf;
	int64_t n_leaf, m_leaf;
	uint8_t *pool;
	int64_t n_pool, m_pool;
	int pc;       /* pending run */
	int64_t pl;
	int64_t remaining; /* symbols still to come for this rope */
} fmr_rope_t;

struct rb3h_fmrw_s {
	int max_nodes, block_len, cur;
	int64_t acc[7];
	fmr_rope_t r[6];
};

/* leaf run codec, rle.h:53-75 */
static int fmr_enc1(uint8_t *p, int c, int64_t l)
{
	if (l < 1LL << 4) { *p = (uint8_t)(l << 3 | c); return 1; }
	else if (l < 1LL << 8) { p[0] = (uint8_t)(0xC0 | l >> 6 << 3 | c); p[1] = (uint8_t)(0x80 | (l & 0x3f)); return 2; }
	else if (l < 1LL << 19) {
		p[0] = (uint8_t)(0xE0 | l >> 18 << 3 | c);
		p[1] = (uint8_t)(0x80 | (l >> 12 & 0x3f)), p[2] = (uint8_t)(0x80 | (l >> 6 & 0x3f)), p[3] = (uint8_t)(0x80 | (l & 0x3f));
		return 4;
	} else {
		int i, shift = 36;
		p[0] = (uint8_t)(0xF0 | l >> 42 << 3 | c);
		for (i = 1; i < 8; ++i, shift -= 6) p[i] = (uint8_t)(0x80 | (l >> shift & 0x3f));
		return 8;
	}
}

/* rle.h:39-51 */
static const uint8_t *fmr_dec1(const uint8_t *p, int *c, int64_t *l)
{
	*c = *p & 7;
	if ((*p & 0x80) == 0) { *l = *p++ >> 3; }
	else if (*p >> 5 == 6) { *l = ((int64_t)(*p & 0x18) << 3) | (p[1] & 0x3f); p += 2; }
	else {
		int n = ((*p & 0x10) >> 2) + 4;
		*l = *p++ >> 3 & 1;
		while (--n) *l = (*l << 6) | (*p++ & 0x3f);
	}
	return p;
}

rb3h_fmrw_t *rb3h_fmrw_init(const int64_t acc[7], int max_nodes, int block_len)
{
	int a;
	rb3h_fmrw_t *w = (rb3h_fmrw_t*)calloc(1, sizeof(*w));
	if (w == 0) return 0;
	if (max_nodes <= 0) max_nodes = 64;   /* ROPE_DEF_MAX_NODES */
	if (block_len <= 0) block_len = 512;  /* ROPE_DEF_BLOCK_LEN */
	if (block_len < 32) block_len = 32;   /* rope_init, rope.c:59-62 */
	if (max_nodes < 4) max_nodes = 4;
	w->max_nodes = (max_nodes + 1) >> 1 << 1;
	w->block_len = (block_len + 7) >> 3 << 3;
	memcpy(w->acc, acc, sizeof(w->acc));
	for (a = 0; a < 6; ++a) w->r[a].pc = -1, w->r[a].remaining = acc[a + 1] - acc[a];
	w->cur = 0;
	return w;
}

void rb3h_fmrw_destroy(rb3h_fmrw_t *w)
{
	int a;
	if (w == 0) return;
	for (a = 0; a < 6; ++a) free(w->r[a].leaf), free(w->r[a].pool);
	free(w);
}

static int fmr_new_leaf(fmr_rope_t *r)
{
	if (r->n_leaf == r->m_leaf) {
		r->m_leaf = r->m_leaf ? r->m_leaf * 2 : 1024;
		r->leaf = (fmr_leaf_t*)realloc(r->leaf, (size_t)r->m_leaf * sizeof(fmr_leaf_t));
		if (r->leaf == 0) return -1;
	}
	memset(&r->leaf[r->n_leaf], 0, sizeof(fmr_leaf_t));
	r->leaf[r->n_leaf].off = r->n_pool;
	++r->n_leaf;
	return 0;
}

static int fmr_flush_run(rb3h_fmrw_t *w, fmr_rope_t *r)
{
	uint8_t code[8];
	int k;
	fmr_leaf_t *lf;
	if (r->pl == 0) return 0;
	k = fmr_enc1(code, r->pc, r->pl);
	if (r->n_leaf == 0 && fmr_new_leaf(r) < 0) return -1;
	lf = &r->leaf[r->n_leaf - 1];
	if (lf->nbytes + k + 2 + FMR_MIN_SPACE > w->block_len) { /* leave room for one later insertion */
		if (fmr_new_leaf(r) < 0) return -1;
		lf = &r->leaf[r->n_leaf - 1];
	}
	if (r->n_pool + k > r->m_pool) {
		r->m_pool = r->m_pool ? r->m_pool * 2 : 1 << 16;
		r->pool = (uint8_t*)realloc(r->pool, (size_t)r->m_pool);
		if (r->pool == 0) return -1;
	}
	memcpy(r->pool + r->n_pool, code, k);
	r->n_pool += k, lf->nbytes += k, lf->c[r->pc] += r->pl;
	r->pl = 0, r->pc = -1;
	return 0;
}

int rb3h_fmrw_enc(rb3h_fmrw_t *w, int64_t l, int c)
{
	if (l < 0 || c < 0 || c > 5) return -1;
	while (l > 0) {
		fmr_rope_t *r;
		int64_t t;
		while (w->cur < 6 && w->r[w->cur].remaining == 0) {
			if (fmr_flush_run(w, &w->r[w->cur]) < 0) return -1;
			++w->cur;
		}
		if (w->cur >= 6) return -2; /* more symbols than the counts announced */
		r = &w->r[w->cur];
		t = l < r->remaining ? l : r->remaining;
		if (r->pc == c) r->pl += t;
		else {
			if (fmr_flush_run(w, r) < 0) return -1;
			r->pc = c, r->pl = t;
		}
		r->remaining -= t, l -= t;
	}
	return 0;
}

/* write the subtree covering leaves [beg, end) with `span` leaves per child at this level */
static void fmr_dump_level(const rb3h_fmrw_t *w, const fmr_rope_t *r, int64_t beg, int64_t end, int64_t span, FILE *fp)
{
	if (span == 1) { /* bottom node: children are leaves */
		uint8_t is_bottom = 1;
		int16_t n = (int16_t)(end - beg);
		int64_t i;
		fwrite(&is_bottom, 1, 1, fp);
		fwrite(&n, 2, 1, fp);
		for (i = beg; i < end; ++i) {
			fwrite(r->leaf[i].c, 8, 6, fp);
			fwrite(&r->leaf[i].nbytes, 2, 1, fp);
			fwrite(r->pool + r->leaf[i].off, 1, r->leaf[i].nbytes, fp);
		}
	} else {
		const int64_t fan = w->max_nodes / 2 > 2 ? w->max_nodes / 2 : 2;
		uint8_t is_bottom = 0;
		int16_t n = (int16_t)((end - beg + span - 1) / span);
		int64_t i;
		fwrite(&is_bottom, 1, 1, fp);
		fwrite(&n, 2, 1, fp);
		for (i = beg; i < end; i += span)
			fmr_dump_level(w, r, i, i + span < end ? i + span : end, span / fan, fp);
	}
}

int rb3h_fmrw_dump(rb3h_fmrw_t *w, FILE *fp)
{
	int a;
	const uint8_t so = 0; /* MR_SO_IO */
	for (a = 0; a < 6; ++a) {
		if (w->r[a].remaining != 0) return -2; /* fewer symbols than announced */
		if (fmr_flush_run(w, &w->r[a]) < 0) return -1;
	}
	fwrite("RB\2", 1, 3, fp);
	fwrite(&so, 1, 1, fp);
	for (a = 0; a < 6; ++a) {
		fmr_rope_t *r = &w->r[a];
		const int32_t mn = w->max_nodes, bl = w->block_len;
		const int64_t fan = w->max_nodes / 2 > 2 ? w->max_nodes / 2 : 2; /* half-full nodes, like a freshly split tree */
		int64_t span = 1;
		if (r->n_leaf == 0 && fmr_new_leaf(r) < 0) return -1; /* an empty rope still has one empty leaf, rope.c:64-68 */
		while (span * fan < r->n_leaf) span *= fan;
		fwrite(&mn, 4, 1, fp);
		fwrite(&bl, 4, 1, fp);
		fmr_dump_level(w, r, 0, r->n_leaf, r->n_leaf <= fan ? 1 : span, fp);
	}
	return fflush(fp) == 0 ? 0 : -1;
}

/* ------------------------------------------------------------------------------------- */
/* reader                                                                                */
/* ------------------------------------------------------------------------------------- */

typedef struct { rb3h_run_f emit; void *data; int pc; int64_t pl; int err; } fmr_rd_t;

static void fmr_rd_run(fmr_rd_t *d, int c, int64_t l)
{
	if (l == 0 || d->err) return;
	if (c == d->pc) d->pl += l;
	else {
		if (d->pl > 0 && d->emit(d->data, d->pc, d->pl) != 0) d->err = -4;
		d->pc = c, d->pl = l;
	}
}

static int fmr_read_node(FILE *fp, int block_len, fmr_rd_t *d, int depth)
{
	uint8_t is_bottom;
	int16_t n, i;
	if (depth > 80) return -3; /* ROPE_MAX_DEPTH */
	if (fread(&is_bottom, 1, 1, fp) != 1 || fread(&n, 2, 1, fp) != 1 || n < 0) return -1;
	if (is_bottom) {
		uint8_t *buf = (uint8_t*)malloc((size_t)block_len + 8);
		if (buf == 0) return -1;
		for (i = 0; i < n; ++i) {
			int64_t c6[6];
			uint16_t nb;
			const uint8_t *q, *end;
			if (fread(c6, 8, 6, fp) != 6 || fread(&nb, 2, 1, fp) != 1 || nb + 2 > block_len || fread(buf, 1, nb, fp) != nb) { free(buf); return -1; }
			for (q = buf, end = buf + nb; q < end;) {
				int c;
				int64_t l;
				q = fmr_dec1(q, &c, &l);
				if (c > 5) { free(buf); return -3; }
				fmr_rd_run(d, c, l);
			}
		}
		free(buf);
	} else {
		for (i = 0; i < n; ++i) {
			int r = fmr_read_node(fp, block_len, d, depth + 1);
			if (r < 0) return r;
		}
	}
	return d->err;
}

/* the 3-byte magic "RB\2" has been consumed by the caller together with 1 byte that is the
 * sorting order when the file is an FMR */
int rb3h_fmr_read_runs(FILE *fp, rb3h_run_f emit, void *data)
{
	int a;
	fmr_rd_t d;
	d.emit = emit, d.data = data, d.pc = -1, d.pl = 0, d.err = 0;
	for (a = 0; a < 6; ++a) {
		int32_t mn, bl;
		int r;
		if (fread(&mn, 4, 1, fp) != 1 || fread(&bl, 4, 1, fp) != 1 || bl < 8 || bl > (1 << 16) + 8) return -1;
		if ((r = fmr_read_node(fp, bl, &d, 0)) < 0) return r;
	}
	if (d.pl > 0 && emit(data, d.pc, d.pl) != 0) return -4;
	return d.err;
}

int rb3h_index_read_runs(const char *fn, rb3h_run_f emit, void *data)
{
	FILE *fp = strcmp(fn, "-") == 0 ? stdin : fopen(fn, "rb");
	char magic[4];
	int ret;
	if (fp == 0) return -1;
	if (fread(magic, 1, 4, fp) != 4) ret = -1;
	else if (memcmp(magic, "RLD\3", 4) == 0) ret = rb3h_fmd_read_runs(fp, emit, data, 0);
	else if (memcmp(magic, "RB\2", 3) == 0) ret = rb3h_fmr_read_runs(fp, emit, data);
	else ret = -2;
	if (fp != stdin) fclose(fp);
	return ret;
}
