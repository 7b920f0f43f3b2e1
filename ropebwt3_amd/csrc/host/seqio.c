/*
 * seqio.c -- sequence input for `build`: FASTA/FASTQ or one-sequence-per-line, gzip
 * transparently, nt6 encoding, both strands, batching.  Restates io.c:12-125 (and the parts of
 * the kseq.h FASTX grammar that io.c relies on) with its own buffered reader.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include "rb3host.h"

#define SIO_BUF 0x10000

struct rb3h_seqio_s {
	gzFile fp;
	int is_line, is_eof, last_char;
	int err;          /* FASTX parsing error met (code < -1); nothing more is read from the file */
	int beg, end;
	uint8_t *buf;
	rb3h_buf_t rec, qual;
};

/* A/C/G/T -> 1..4 (either case), 0..4 stay, everything else -> 5 (io.c:12-28) */
static uint8_t sio_nt6[256];
static int sio_nt6_ready = 0;

static void sio_init_table(void)
{
	int i;
	if (sio_nt6_ready) return;
	for (i = 0; i < 256; ++i) sio_nt6[i] = i < 5 ? i : 5;
	sio_nt6['A'] = sio_nt6['a'] = 1, sio_nt6['C'] = sio_nt6['c'] = 2;
	sio_nt6['G'] = sio_nt6['g'] = 3, sio_nt6['T'] = sio_nt6['t'] = 4;
	sio_nt6_ready = 1;
}

void rb3h_char2nt6(int64_t l, uint8_t *s)
{
	int64_t i;
	sio_init_table();
	for (i = 0; i < l; ++i) s[i] = sio_nt6[s[i]];
}

void rb3h_revcomp6(int64_t l, uint8_t *s) /* in place; 1<->4, 2<->3, 0 and 5 unchanged */
{
	int64_t i, j;
	for (i = 0, j = l - 1; i < j; ++i, --j) {
		uint8_t a = s[i], b = s[j];
		s[i] = (b >= 1 && b <= 4) ? 5 - b : b;
		s[j] = (a >= 1 && a <= 4) ? 5 - a : a;
	}
	if (i == j) s[i] = (s[i] >= 1 && s[i] <= 4) ? 5 - s[i] : s[i];
}

static int buf_grow(rb3h_buf_t *b, int64_t need)
{
	if (need <= b->m) return 0;
	int64_t m = need + (need >> 1) + 16;
	uint8_t *s = (uint8_t*)realloc(b->s, (size_t)m);
	if (s == 0) return -1;
	b->s = s, b->m = m;
	return 0;
}

rb3h_seqio_t *rb3h_seq_open(const char *fn, int is_line)
{
	gzFile f = fn && strcmp(fn, "-") ? gzopen(fn, "r") : gzdopen(0, "r");
	rb3h_seqio_t *fp;
	if (f == 0) return 0;
	sio_init_table();
	fp = (rb3h_seqio_t*)calloc(1, sizeof(*fp));
	fp->fp = f, fp->is_line = !!is_line;
	fp->buf = (uint8_t*)malloc(SIO_BUF);
	return fp;
}

void rb3h_seq_close(rb3h_seqio_t *fp)
{
	if (fp == 0) return;
	gzclose(fp->fp);
	free(fp->buf); free(fp->rec.s); free(fp->qual.s); free(fp);
}

static int sio_fill(rb3h_seqio_t *fp)
{
	if (fp->is_eof) return 0;
	fp->beg = 0;
	fp->end = gzread(fp->fp, fp->buf, SIO_BUF);
	if (fp->end < SIO_BUF) fp->is_eof = 1;
	if (fp->end <= 0) { fp->end = 0; return 0; }
	return 1;
}

static inline int sio_getc(rb3h_seqio_t *fp)
{
	if (fp->beg >= fp->end && !sio_fill(fp)) return -1;
	return fp->buf[fp->beg++];
}

/* append the rest of the current line to b (newline consumed, one trailing '\r' dropped when
 * the accumulated string is longer than 1, like kseq.h:146); returns -1 at EOF with nothing
 * read, else the accumulated length */
static int64_t sio_getline(rb3h_seqio_t *fp, rb3h_buf_t *b, int append)
{
	int got = 0;
	if (!append) b->l = 0;
	if (fp->beg >= fp->end && fp->is_eof) return -1;
	for (;;) {
		int i;
		const uint8_t *nl;
		if (fp->beg >= fp->end && !sio_fill(fp)) break;
		nl = (const uint8_t*)memchr(fp->buf + fp->beg, '\n', (size_t)(fp->end - fp->beg));
		i = nl ? (int)(nl - fp->buf) : fp->end;
		if (buf_grow(b, b->l + (i - fp->beg) + 2) < 0) return -2;
		memcpy(b->s + b->l, fp->buf + fp->beg, i - fp->beg);
		b->l += i - fp->beg;
		got = 1;
		if (i < fp->end) { fp->beg = i + 1; break; }
		fp->beg = fp->end;
	}
	(void)got;
	if (b->s == 0 && buf_grow(b, 2) < 0) return -2;
	if (b->l > 1 && b->s[b->l - 1] == '\r') --b->l;
	return b->l;
}

/* one FASTA/FASTQ record into fp->rec; >=0 length, -1 EOF, -2 truncated quality */
static int64_t sio_read_fastx(rb3h_seqio_t *fp)
{
	int c;
	if (fp->last_char == 0) { /* jump to the next header line */
		while ((c = sio_getc(fp)) != -1 && c != '>' && c != '@');
		if (c == -1) return -1;
		fp->last_char = c;
	}
	fp->rec.l = fp->qual.l = 0;
	/* header line (name + comment) is not needed for the BWT */
	if (fp->beg >= fp->end && fp->is_eof) return -1;
	{
		rb3h_buf_t *h = &fp->qual; /* scratch */
		if (sio_getline(fp, h, 0) < 0) return -1;
		h->l = 0;
	}
	if (buf_grow(&fp->rec, 256) < 0) return -2;
	while ((c = sio_getc(fp)) != -1 && c != '>' && c != '+' && c != '@') {
		if (c == '\n') continue; /* empty line */
		if (buf_grow(&fp->rec, fp->rec.l + 2) < 0) return -2;
		fp->rec.s[fp->rec.l++] = (uint8_t)c;
		sio_getline(fp, &fp->rec, 1);
	}
	if (c == '>' || c == '@') fp->last_char = c;
	if (c != '+') return fp->rec.l;
	while ((c = sio_getc(fp)) != -1 && c != '\n'); /* rest of the '+' line */
	if (c == -1) return -2;
	while (sio_getline(fp, &fp->qual, 1) >= 0 && fp->qual.l < fp->rec.l);
	fp->last_char = 0;
	if (fp->qual.l != fp->rec.l) return -2;
	return fp->rec.l;
}

static int64_t sio_add(rb3h_buf_t *seq, int is_for, int is_rev, int64_t l, uint8_t *s)
{ /* io.c:84-102 */
	int64_t n = 0;
	rb3h_char2nt6(l, s);
	if (buf_grow(seq, seq->l + 2 * (l + 1) + 1) < 0) return -1;
	if (is_for) {
		memcpy(seq->s + seq->l, s, l);
		seq->s[seq->l + l] = 0;
		seq->l += l + 1, ++n;
	}
	if (is_rev) { /* reverse complement straight into the batch: 1<->4, 2<->3, 0 and 5 unchanged (io.c:30-40) */
		static const uint8_t comp[8] = { 0, 4, 3, 2, 1, 5, 6, 7 };
		uint8_t *d = seq->s + seq->l;
		int64_t i;
		for (i = 0; i < l; ++i) d[i] = comp[s[l - 1 - i] & 7];
		d[l] = 0;
		seq->l += l + 1, ++n;
	}
	return n;
}

int64_t rb3h_seq_read(rb3h_seqio_t *fp, rb3h_buf_t *seq, int64_t max_len, int is_for, int is_rev, int64_t *n_empty)
{
	int64_t n_seq = 0, ret;
	if (!is_for && !is_rev) return -3;
	if (fp->err) return 0; /* like the end of the file (the reference's loop ends there too, build.c:212) */
	for (;;) {
		ret = fp->is_line ? sio_getline(fp, &fp->rec, 0) : sio_read_fastx(fp);
		if (ret < 0) break;
		if (ret == 0) { /* an empty record would put two adjacent sentinels into the text, which the
		                   reference mis-sorts or crashes on; skip it and tell the caller */
			if (n_empty) ++*n_empty;
			continue;
		}
		ret = sio_add(seq, is_for, is_rev, fp->rec.l, fp->rec.s);
		if (ret < 0) return -4;
		n_seq += ret;
		if (max_len > 0 && seq->l > max_len) break; /* io.c:114,119 */
	}
	if (!fp->is_line && ret < -1) fp->err = (int)ret; /* FASTX parsing error: the records read before it still count (io.c:121-124) */
	return n_seq;
}

int rb3h_seq_error(const rb3h_seqio_t *fp) { return fp->err; }
