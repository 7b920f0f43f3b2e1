/*
 * rb3host.h -- host-side C of the MI355X build: everything around the HIP merge engine that
 * `ropebwt3 build` needs (sequence input, per-batch suffix sorting, FMD / FMR codecs).
 * Citations are file:line in the reference tree.
 */
#ifndef RB3HOST_H
#define RB3HOST_H

#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RB3H_VERSION "3.10-r281-mi355x-r4"

extern int rb3h_verbose;

/* ---- misc (misc.c:7-16, 116-150) ---- */
int64_t rb3h_parse_num(const char *str);
double rb3h_realtime(void);
double rb3h_cputime(void);
double rb3h_percent_cpu(void);
long rb3h_peakrss(void);
void rb3h_init(void);

/* ---- suffix sorting of one batch (sais-ss.c:50-56) ----
 * n_threads is there for signature parity with rb3_build_sais and is IGNORED: this from-scratch SA-IS is sequential
 * (parallelism on the host side is `-p N`: N batches at once).  It is the fallback sorter only. */
int rb3h_build_bwt(int64_t n_seq, int64_t len, uint8_t *seq, int n_threads);
/* same, plus the LF-walker list for rb3gpu_merge_plain_walkers (layout = rb3gpu_walker_t): the
 * sampled inverse suffix array is free here because the suffix array is still in memory */
typedef struct { int64_t row, ka0, nsteps, flags; } rb3h_walker_t;
int rb3h_build_bwt_walkers(int64_t n_seq, int64_t len, uint8_t *seq, int n_threads, int64_t step, int64_t *n_walkers, rb3h_walker_t **walkers);
int rb3h_walkers_from_ckrow(int64_t len, const uint8_t *text, int64_t step, const int64_t *ckrow, int64_t *n_walkers, rb3h_walker_t **walkers);
int rb3h_walkers_text(int64_t len, const uint8_t *text, int64_t step, int64_t *n_walkers, rb3h_walker_t **walkers); /* by text position (rb3gpu_merge_text_dev) */

/* ---- sequence input (io.c) ---- */
typedef struct { int64_t l, m; uint8_t *s; } rb3h_buf_t;
struct rb3h_seqio_s;
typedef struct rb3h_seqio_s rb3h_seqio_t;
rb3h_seqio_t *rb3h_seq_open(const char *fn, int is_line);                  /* io.c:60-72 */
void rb3h_seq_close(rb3h_seqio_t *fp);                                     /* io.c:74-82 */
/* a byte range of a plain file (`build --gpus N` on one file): the records that START in [beg, end); end <= 0: to the end of the file */
int rb3h_seq_splittable(const char *fn, int64_t *size);                    /* a regular file that is not gzip-compressed */
rb3h_seqio_t *rb3h_seq_open_range(const char *fn, int is_line, int64_t beg, int64_t end);
int64_t rb3h_seq_record_start(const char *fn, int is_line, int64_t off);            /* the cut rb3h_seq_open_range makes at an offset (-1: not a plain file) */
/* io.c:104-125: fill `seq` with nt6(forward)+0 and nt6(revcomp)+0 per record until
 * seq->l > max_len; returns the number of strings appended (0 at EOF) or <0 on a parse error;
 * *n_empty counts records of length 0, which are skipped (out of contract in the reference) */
int64_t rb3h_seq_read(rb3h_seqio_t *fp, rb3h_buf_t *seq, int64_t max_len, int is_for, int is_rev, int64_t *n_empty);
/* where the batch buffer (`seq` of rb3h_seq_read) gets its memory: alloc returns at least min_bytes and says how much it
 * gave in *cap (a pool may hand out a larger buffer); NULL, NULL = realloc/free (default).  The CLI sets page-locked memory
 * here (rb3gpu_pinned_alloc) so that a batch goes to HBM with one DMA; a batch buffer is then freed with rb3h_batch_free. */
typedef void *(*rb3h_alloc_f)(int64_t min_bytes, int64_t *cap);
typedef void (*rb3h_free_f)(void *p);
void rb3h_seq_set_batch_allocator(rb3h_alloc_f alloc, rb3h_free_f release);
void rb3h_batch_free(void *p);
int rb3h_seq_error(const rb3h_seqio_t *fp); /* != 0: a FASTX parsing error ended the file early (code as in kseq: -2 truncated quality, ...) */
int64_t rb3h_strand_pairs(int64_t len, const uint8_t *text, int64_t n_seq, int64_t max_pairs, int64_t *pair_start); /* record offsets of a both-strand batch */
void rb3h_char2nt6(int64_t l, uint8_t *s);                                 /* io.c:23-28 */
void rb3h_revcomp6(int64_t l, uint8_t *s);                                 /* io.c:30-40 */

/* ---- FMD (rld0) writer / reader ---- */
struct rb3h_fmdw_s;
typedef struct rb3h_fmdw_s rb3h_fmdw_t;
rb3h_fmdw_t *rb3h_fmdw_init(void);                                         /* rld_init(6, 3), rld0.c:57-75 */
int rb3h_fmdw_enc(rb3h_fmdw_t *w, int64_t l, int c);                        /* rld_enc, rld0.c:153-161 */
int rb3h_fmdw_enc_words(rb3h_fmdw_t *w, int64_t n, const uint64_t *words, int64_t end); /* bulk: start << 3 | sym of maximal runs */
int rb3h_fmdw_adopt(rb3h_fmdw_t *w, uint64_t *words, int64_t n_words, const int64_t acc[7]); /* data section packed elsewhere */
int rb3h_fmdw_finish(rb3h_fmdw_t *w);                                      /* rld_enc_finish, rld0.c:206-216 */
int rb3h_fmdw_dump_file(const rb3h_fmdw_t *w, const char *fn);
int rb3h_fmdw_dump(const rb3h_fmdw_t *w, FILE *fp);                         /* rld_dump, rld0.c:222-243 */
void rb3h_fmdw_destroy(rb3h_fmdw_t *w);
int64_t rb3h_fmdw_nbytes(const rb3h_fmdw_t *w);

typedef int (*rb3h_run_f)(void *data, int c, int64_t l);
/* decode every run of an FMD file in order (rld_restore + rld_dec, rld0.c:267-320, rld0.h:85-122) */
int rb3h_fmd_read_runs(FILE *fp, rb3h_run_f emit, void *data, int64_t mcnt[6]);
int rb3h_fmd_read_words(const char *fn, uint64_t **z, int64_t *n_words, int64_t mcnt[6]); /* undecoded, for rb3gpu_from_fmd_words */

/* ---- FMR (mrope) writer / reader ---- */
struct rb3h_fmrw_s;
typedef struct rb3h_fmrw_s rb3h_fmrw_t;
/* stream runs in BWT order, then dump the six ropes (mr_dump, mrope.c:152-159; rope_dump,
 * rope.c:265-287).  cnt[6] = symbol counts of the whole BWT (rope a holds rows C[a]..C[a+1]). */
rb3h_fmrw_t *rb3h_fmrw_init(const int64_t acc[7], int max_nodes, int block_len);
int rb3h_fmrw_enc(rb3h_fmrw_t *w, int64_t l, int c);
int rb3h_fmrw_dump(rb3h_fmrw_t *w, FILE *fp);
void rb3h_fmrw_destroy(rb3h_fmrw_t *w);
/* decode an FMR file (mr_restore, mrope.c:161-177; rope_restore, rope.c:289-330) */
int rb3h_fmr_read_runs(FILE *fp, rb3h_run_f emit, void *data);

/* open an index file of either kind and stream its runs; returns 0, or <0 on error */
int rb3h_index_read_runs(const char *fn, rb3h_run_f emit, void *data);

#ifdef __cplusplus
}
#endif

#endif
