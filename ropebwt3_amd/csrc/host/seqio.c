/*
 * seqio.c -- sequence input for `build`: FASTA/FASTQ or one-sequence-per-line, gzip
 * transparently, nt6 encoding, both strands, batching.  Restates io.c:12-125 (and the parts of
 * the kseq.h FASTX grammar that io.c relies on) with its own buffered reader.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <unistd.h>
#include <fcntl.h>
#include <errno.h>
#include <sys/stat.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include "rb3host.h"

#define SIO_BUF 0x100000

struct rb3h_seqio_s {
	gzFile fp;        /* NULL: the file is not gzip-compressed and is read with read(2) (no pass through zlib's buffer) */
	int fd;
	int is_line, is_eof, last_char;
	int err;          /* FASTX parsing error met (code < -1); nothing more is read from the file */
	int io_err;       /* read(2) / gzread failed in the middle of the file: what was read so far must not be taken for the whole input */
	int beg, end;
	uint8_t *buf;
	rb3h_buf_t rec, qual;
	int64_t buf_off;   /* file offset of buf[0] (plain files only: a byte range of a file, rb3h_seq_open_range) */
	int64_t next_off;  /* file offset of the next fill */
	int64_t range_end; /* > 0: records that start at or behind this file offset belong to somebody else */
};

/* A/C/G/T -> 1..4 (either case), 0..4 stay, everything else -> 5 (io.c:12-28) */
static uint8_t sio_nt6[256];
static uint8_t sio_nt6c[256]; /* the complement of the nt6 code: 1<->4, 2<->3, 0 and 5 unchanged (io.c:30-40) */
static int sio_nt6_ready = 0;

static void sio_init_table(void)
{
	int i;
	if (sio_nt6_ready) return;
	for (i = 0; i < 256; ++i) sio_nt6[i] = i < 5 ? i : 5;
	sio_nt6['A'] = sio_nt6['a'] = 1, sio_nt6['C'] = sio_nt6['c'] = 2;
	sio_nt6['G'] = sio_nt6['g'] = 3, sio_nt6['T'] = sio_nt6['t'] = 4;
	for (i = 0; i < 256; ++i) sio_nt6c[i] = (sio_nt6[i] >= 1 && sio_nt6[i] <= 4) ? 5 - sio_nt6[i] : sio_nt6[i];
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

/* l characters -> nt6 codes at dfor (if not NULL) and the reverse complement at drev (if not NULL): io.c:12-40, 84-102 in one
 * pass over the input, 16 characters at a time where SSE2 is there (a table lookup per character ran at ~1 GB/s and was the
 * slowest stage of a build from reads).  Chunks that hold raw codes 0..4 (binary input) go through the tables. */
static void sio_convert(const uint8_t *src, int64_t l, uint8_t *dfor, uint8_t *drev)
{
	int64_t i = 0;
#if defined(__SSE2__)
	const __m128i m_up = _mm_set1_epi8((char)0xDF), cA = _mm_set1_epi8('A'), cC = _mm_set1_epi8('C'), cG = _mm_set1_epi8('G'), cT = _mm_set1_epi8('T');
	const __m128i k1 = _mm_set1_epi8(1), k2 = _mm_set1_epi8(2), k3 = _mm_set1_epi8(3), k4 = _mm_set1_epi8(4), k5 = _mm_set1_epi8(5);
	for (; i + 16 <= l; i += 16) {
		const __m128i c = _mm_loadu_si128((const __m128i*)(src + i));
		if (_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_min_epu8(c, k4), c)) != 0) { /* a raw code among them: the slow way */
			int64_t q;
			if (dfor) for (q = i; q < i + 16; ++q) dfor[q] = sio_nt6[src[q]];
			if (drev) for (q = i; q < i + 16; ++q) drev[l - 1 - q] = sio_nt6c[src[q]];
			continue;
		}
		{
			const __m128i up = _mm_and_si128(c, m_up);
			const __m128i a = _mm_cmpeq_epi8(up, cA), cc = _mm_cmpeq_epi8(up, cC), g = _mm_cmpeq_epi8(up, cG), t = _mm_cmpeq_epi8(up, cT);
			if (dfor) {
				const __m128i v = _mm_or_si128(_mm_or_si128(_mm_and_si128(a, k4), _mm_and_si128(cc, k3)), _mm_or_si128(_mm_and_si128(g, k2), _mm_and_si128(t, k1)));
				_mm_storeu_si128((__m128i*)(dfor + i), _mm_sub_epi8(k5, v)); /* A 1, C 2, G 3, T 4, the rest 5 */
			}
			if (drev) {
				const __m128i v = _mm_or_si128(_mm_or_si128(_mm_and_si128(a, k1), _mm_and_si128(cc, k2)), _mm_or_si128(_mm_and_si128(g, k3), _mm_and_si128(t, k4)));
				__m128i x = _mm_sub_epi8(k5, v); /* A 4, C 3, G 2, T 1, the rest 5 */
				x = _mm_or_si128(_mm_slli_epi16(x, 8), _mm_srli_epi16(x, 8)); /* reverse the 16 bytes */
				x = _mm_shufflelo_epi16(x, _MM_SHUFFLE(0, 1, 2, 3));
				x = _mm_shufflehi_epi16(x, _MM_SHUFFLE(0, 1, 2, 3));
				x = _mm_shuffle_epi32(x, _MM_SHUFFLE(1, 0, 3, 2));
				_mm_storeu_si128((__m128i*)(drev + (l - 16 - i)), x);
			}
		}
	}
#endif
	for (; i < l; ++i) {
		if (dfor) dfor[i] = sio_nt6[src[i]];
		if (drev) drev[l - 1 - i] = sio_nt6c[src[i]];
	}
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

/* the batch buffer may live in memory from another allocator (page-locked: see rb3h_seq_set_batch_allocator) */
static rb3h_alloc_f sio_batch_alloc = 0;
static rb3h_free_f sio_batch_release = 0;

void rb3h_seq_set_batch_allocator(rb3h_alloc_f alloc, rb3h_free_f release)
{
	sio_batch_alloc = alloc && release ? alloc : 0, sio_batch_release = alloc && release ? release : 0;
}

void rb3h_batch_free(void *p)
{
	if (p == 0) return;
	if (sio_batch_release) sio_batch_release(p); else free(p);
}

static int batch_grow(rb3h_buf_t *b, int64_t need)
{
	if (need <= b->m) return 0;
	if (sio_batch_alloc == 0) return buf_grow(b, need);
	{
		int64_t cap = 0, want = need * 2 + 16; /* (page-locked memory is slow to obtain: grow in few steps) */
		uint8_t *s = (uint8_t*)sio_batch_alloc(want, &cap);
		if (s == 0 || cap < need) return -1;
		if (b->l > 0) memcpy(s, b->s, (size_t)b->l);
		if (b->s) sio_batch_release(b->s);
		b->s = s, b->m = cap;
	}
	return 0;
}

/* read(2) until n bytes are there or the file ends; an interrupted call is repeated, any other error is reported in *io_err
 * (the bytes read before it are still returned) */
static int sio_read_full(int fd, uint8_t *buf, int n, int *io_err)
{
	int got = 0;
	while (got < n) {
		const ssize_t r = read(fd, buf + got, (size_t)(n - got));
		if (r < 0) {
			if (errno == EINTR) continue;
			*io_err = errno ? errno : EIO;
			break;
		}
		if (r == 0) break;
		got += (int)r;
	}
	return got;
}

rb3h_seqio_t *rb3h_seq_open(const char *fn, int is_line)
{
	const int fd = fn && strcmp(fn, "-") ? open(fn, O_RDONLY) : dup(0);
	rb3h_seqio_t *fp;
	off_t at;
	if (fd < 0) return 0;
	sio_init_table();
	fp = (rb3h_seqio_t*)calloc(1, sizeof(*fp));
	fp->fd = fd, fp->is_line = !!is_line;
	fp->buf = (uint8_t*)malloc(SIO_BUF);
	/* A seekable file is sniffed with pread(2), which consumes nothing: gzip magic -> through zlib, anything else -> plain
	 * read(2) (no pass through zlib's buffer).  A pipe, FIFO or terminal cannot be given bytes back, so it always goes through
	 * zlib, which handles gzip and plain input alike (gzdopen as in io.c:64): nothing zlib needs is ever read here. */
	at = lseek(fd, 0, SEEK_CUR);
	if (at == (off_t)-1) fp->fp = gzdopen(fd, "r");
	else {
		uint8_t magic[2];
		const ssize_t k = pread(fd, magic, 2, at);
		if (k == 2 && magic[0] == 0x1f && magic[1] == 0x8b) fp->fp = gzdopen(fd, "r");
		else return fp;
	}
	if (fp->fp == 0) { close(fd); free(fp->buf); free(fp); return 0; }
	return fp;
}

/* can a file be cut into byte ranges that are read independently?  A regular file that is not gzip-compressed; *size = its length */
int rb3h_seq_splittable(const char *fn, int64_t *size)
{
	struct stat st;
	uint8_t magic[2];
	int fd, ok = 0;
	if (fn == 0 || !strcmp(fn, "-") || (fd = open(fn, O_RDONLY)) < 0) return 0;
	if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode)) {
		const ssize_t k = pread(fd, magic, 2, 0);
		ok = !(k == 2 && magic[0] == 0x1f && magic[1] == 0x8b);
		if (size) *size = (int64_t)st.st_size;
	}
	close(fd);
	return ok;
}

/* first record start at or behind file offset `from` (the file's length if there is none).  A record starts where a line starts
 * -- with -L every line; FASTA (the first non-empty line of the file begins with '>'): a line that begins with '>'; FASTQ (it begins
 * with '@'): a line L0 that begins with '@' whose second successor L2 begins with '+', with L1 and L3 of one length and L4 (if there
 * is one) beginning with '@' again -- a quality line may begin with '@' too, but then the line after it is a header and the one after
 * that a sequence.  Anything else (multi-line FASTQ, a file that begins with neither) has no start this function vouches for: the
 * answer is the file's length, i.e. such a file is not cut.  What is found IS a record start for the sequential reader as well, and
 * rb3h_seq_open_range applies the function to BOTH ends of a range, so ranges that share an end tile the file whatever it holds. */
static int64_t sio_record_start(int fd, int64_t from, int64_t size, int is_line)
{
	uint8_t first = 0, *buf;
	int64_t ls[5] = {-1, -1, -1, -1, -1}, off, found = -1; /* the last five line starts and their first bytes */
	uint8_t lc[5] = {0, 0, 0, 0, 0};
	int at_start; /* the next byte begins a line */
	if (from <= 0) return 0;
	if (from >= size) return size;
	buf = (uint8_t*)malloc(1 << 20);
	if (buf == 0) return -1;
	if (!is_line) { /* the first byte of the first non-empty line says what the file is */
		for (off = 0; off < size && first == 0;) {
			const ssize_t k = pread(fd, buf, 1 << 16, off);
			ssize_t i;
			if (k <= 0) break;
			for (i = 0; i < k && first == 0; ++i) if (buf[i] != '\n' && buf[i] != '\r') first = buf[i];
			off += k;
		}
		if (first != '>' && first != '@') { free(buf); return size; }
		if (first == '@') { /* FASTQ is only cut if its first records are four lines each: @..., sequence, +..., quality of the sequence's length
		                       (in a multi-line FASTQ file quality lines that begin with '@' and '+' can imitate any local pattern) */
			const ssize_t k = pread(fd, buf, 1 << 20, 0);
			ssize_t i = 0, b0;
			int nrec = 0, line = 0, ok = 1;
			int64_t len1 = 0;
			while (i < k && (buf[i] == '\n' || buf[i] == '\r')) ++i;
			for (b0 = i; i < k && ok && nrec < 64; ++i) {
				if (buf[i] != '\n') continue;
				if (line == 0) ok = buf[b0] == '@';
				else if (line == 1) len1 = i - b0;
				else if (line == 2) ok = buf[b0] == '+';
				else ok = i - b0 == len1, ++nrec;
				line = (line + 1) & 3, b0 = i + 1;
			}
			if (!ok) { free(buf); return size; }
		}
	}
	off = from - 1, at_start = 0; /* (the byte before `from` says whether `from` itself begins a line) */
	while (found < 0 && off < size) {
		const ssize_t k = pread(fd, buf, 1 << 20, off);
		ssize_t i;
		if (k <= 0) break;
		for (i = 0; i < k && found < 0; ++i) {
			if (at_start) {
				const int64_t o = off + i;
				at_start = 0;
				if (is_line) found = o;
				else if (first == '>') { if (buf[i] == '>') found = o; }
				else {
					memmove(ls, ls + 1, 4 * sizeof(ls[0])), memmove(lc, lc + 1, 4), ls[4] = o, lc[4] = buf[i];
					if (ls[0] >= 0 && lc[0] == '@' && lc[2] == '+' && lc[4] == '@' && ls[2] - ls[1] == ls[4] - ls[3]) found = ls[0];
				}
			}
			if (buf[i] == '\n') at_start = 1;
		}
		off += k;
	}
	if (found < 0 && !is_line && first == '@' && ls[1] >= 0 && lc[1] == '@' && lc[3] == '+') { /* the last record of the file: L1..L4 with nothing behind */
		uint8_t last = 0;
		const int64_t l4 = (pread(fd, &last, 1, size - 1) == 1 && last == '\n' ? size : size + 1) - ls[4];
		if (ls[3] - ls[2] == l4) found = ls[1];
	}
	free(buf);
	return found >= 0 ? found : size;
}

/* the cut sio_record_start makes at file offset `off` (what rb3h_seq_open_range does with either end of its range) */
int64_t rb3h_seq_record_start(const char *fn, int is_line, int64_t off)
{
	int64_t size = 0, at;
	int fd;
	if (!rb3h_seq_splittable(fn, &size) || (fd = open(fn, O_RDONLY)) < 0) return -1;
	at = sio_record_start(fd, off, size, is_line);
	close(fd);
	return at;
}

/* the records of a plain file that START in the byte range [beg, end) (end <= 0: to the end of the file), both ends moved to the
 * first record start at or behind them: ranges that share their ends tile a file -- every record goes to exactly one reader, in
 * file order -- because the reader of [a, b) stops at the very offset the reader of [b, c) starts at */
rb3h_seqio_t *rb3h_seq_open_range(const char *fn, int is_line, int64_t beg, int64_t end)
{
	int64_t size = 0, at, stop = 0;
	rb3h_seqio_t *fp;
	if (beg <= 0 && end <= 0) return rb3h_seq_open(fn, is_line);
	if (!rb3h_seq_splittable(fn, &size)) return 0;
	fp = rb3h_seq_open(fn, is_line);
	if (fp == 0 || fp->fp != 0) { rb3h_seq_close(fp); return 0; }
	at = sio_record_start(fp->fd, beg, size, is_line);
	if (end > 0 && end < size) stop = end <= beg ? at : sio_record_start(fp->fd, end, size, is_line);
	if (at < 0 || stop < 0 || lseek(fp->fd, (off_t)at, SEEK_SET) == (off_t)-1) { rb3h_seq_close(fp); return 0; }
	fp->next_off = at, fp->buf_off = at;
	fp->range_end = stop > 0 && stop < size ? stop : 0; /* (no record start behind `end`: this reader takes the rest of the file) */
	if (fp->range_end > 0 && at >= fp->range_end) fp->is_eof = 1; /* (no record starts in this range) */
	if (end > 0 && end < size && stop >= size && at >= size) fp->is_eof = 1;
	return fp;
}

void rb3h_seq_close(rb3h_seqio_t *fp)
{
	if (fp == 0) return;
	if (fp->fp) gzclose(fp->fp); /* (closes the descriptor it was opened on) */
	else if (fp->fd >= 0) close(fp->fd);
	free(fp->buf); free(fp->rec.s); free(fp->qual.s); free(fp);
}

static int sio_fill(rb3h_seqio_t *fp)
{
	if (fp->is_eof) return 0;
	fp->beg = 0;
	if (fp->fp) {
		fp->end = gzread(fp->fp, fp->buf, SIO_BUF);
		if (fp->end < SIO_BUF) { /* the end of the stream, or of what could be read of it: a truncated or corrupt gzip file is an error */
			int zerr = Z_OK;
			(void)gzerror(fp->fp, &zerr);
			if (fp->end < 0 || (zerr != Z_OK && zerr != Z_STREAM_END)) fp->io_err = EIO;
		}
	} else {
		fp->end = sio_read_full(fp->fd, fp->buf, SIO_BUF, &fp->io_err);
		fp->buf_off = fp->next_off;
		if (fp->end > 0) fp->next_off += fp->end;
	}
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
	/* a byte range of a file: the record whose header starts at or behind the end of the range is the next reader's first */
	if (fp->range_end > 0 && fp->buf_off + fp->beg - 1 >= fp->range_end) return -1;
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
	uint8_t *df, *dr;
	if (batch_grow(seq, seq->l + 2 * (l + 1) + 1) < 0) return -1;
	df = is_for ? seq->s + seq->l : 0, dr = is_rev ? seq->s + seq->l + (is_for ? l + 1 : 0) : 0;
	sio_convert(s, l, df, dr); /* both strands straight into the batch */
	if (df) df[l] = 0, seq->l += l + 1, ++n;
	if (dr) dr[l] = 0, seq->l += l + 1, ++n;
	return n;
}

int64_t rb3h_seq_read(rb3h_seqio_t *fp, rb3h_buf_t *seq, int64_t max_len, int is_for, int is_rev, int64_t *n_empty)
{
	int64_t n_seq = 0, ret;
	if (!is_for && !is_rev) return -3;
	if (fp->err) return 0; /* like the end of the file (the reference's loop ends there too, build.c:212) */
	for (;;) {
		if (fp->is_line && fp->range_end > 0) { /* a byte range of a file: the line that starts at or behind its end is the next reader's first */
			if (fp->beg >= fp->end && !fp->is_eof) sio_fill(fp);
			if (fp->beg < fp->end && fp->buf_off + fp->beg >= fp->range_end) { ret = -1; break; }
		}
		/* one-sequence-per-line input whose next line lies whole in the I/O buffer (all but one line per megabyte): converted
		 * straight from there into the batch, both strands, without the detour through the record buffer */
		if (fp->is_line && fp->beg < fp->end) {
			const uint8_t *src = fp->buf + fp->beg;
			const uint8_t *nl = (const uint8_t*)memchr(src, '\n', (size_t)(fp->end - fp->beg));
			if (nl) {
				int64_t l = nl - src;
				fp->beg += (int)l + 1;
				if (l > 1 && src[l - 1] == '\r') --l; /* (as sio_getline) */
				if (l == 0) { if (n_empty) ++*n_empty; continue; }
				if (batch_grow(seq, seq->l + 2 * (l + 1) + 1) < 0) return -4;
				{
					uint8_t *df = is_for ? seq->s + seq->l : 0, *dr = is_rev ? seq->s + seq->l + (is_for ? l + 1 : 0) : 0;
					sio_convert(src, l, df, dr);
					if (df) df[l] = 0, seq->l += l + 1, ++n_seq;
					if (dr) dr[l] = 0, seq->l += l + 1, ++n_seq;
				}
				ret = 0;
				if (max_len > 0 && seq->l > max_len) break; /* io.c:114,119 */
				continue;
			}
		}
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
	if (fp->io_err) return -5; /* the file could not be read to its end: not an end of file (the caller must not index a prefix) */
	if (!fp->is_line && ret < -1) fp->err = (int)ret; /* FASTX parsing error: the records read before it still count (io.c:121-124) */
	return n_seq;
}

int rb3h_seq_error(const rb3h_seqio_t *fp) { return fp->err; }

/* the records of a batch that was read with BOTH strands (rb3h_seq_read with is_for and is_rev: per record l symbols, 0, the reverse
 * complement, 0): pair_start[i] = offset of record i, for rb3gpu_sorter_upload_fwd.  Returns the number of records, or 0 if the
 * batch has more than max_pairs records or not that layout (n_seq is the number of strings: two per record). */
int64_t rb3h_strand_pairs(int64_t len, const uint8_t *text, int64_t n_seq, int64_t max_pairs, int64_t *pair_start)
{
	int64_t i, p = 0;
	if (n_seq <= 0 || (n_seq & 1) || n_seq / 2 > max_pairs || len < 4) return 0;
	for (i = 0; i < n_seq / 2; ++i) {
		const uint8_t *e = p < len ? (const uint8_t*)memchr(text + p, 0, (size_t)(len - p)) : 0;
		int64_t l;
		if (e == 0) return 0;
		l = (e - text) - p;
		if (l < 1 || p + 2 * (l + 1) > len || text[p + 2 * l + 1] != 0) return 0;
		pair_start[i] = p;
		p += 2 * (l + 1);
	}
	return p == len ? n_seq / 2 : 0;
}
