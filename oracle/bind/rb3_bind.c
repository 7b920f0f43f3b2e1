/*
 * rb3_bind.c -- the binding of INTEGRATION.md, compiled and tested: the UNMODIFIED reference `ropebwt3` (its own main.c,
 * build.c, io.c, sais-ss.c, rld0.c ... objects, built by oracle/Makefile straight from /root/reference) with the calls its
 * build.c makes on an mrope_t* for the merge path redirected to the MI355X engine at LINK time (GNU ld --wrap):
 *
 *   rb3_enc_plain2fmr   (fm-index.c:114, called at build.c:77,223)  -> rb3gpu_create + rb3gpu_from_plain
 *   rb3_fmi_merge_plain (fm-index.c:279, called at build.c:78,226)  -> rb3gpu_merge_plain
 *   rb3_enc_fmr2fmd     (fm-index.c:31,  called at build.c:250)     -> rb3gpu_export_runs feeding the reference's own rld_enc
 *   mr_print_bwt        (mrope.c:201,    called at build.c:254)     -> rb3gpu_export_runs
 *   mr_destroy          (mrope.c:28,     called at build.c:261)     -> rb3gpu_destroy
 *
 * The "mrope_t*" that travels through build.c is a tagged box around the engine handle; a real mrope_t (ropebwt2 insertion,
 * an index loaded with -i, the other sub-commands) still reaches the reference's own functions through __real_*.
 * Result: oracle/_ref/ropebwt3-bound, the reference CLI with the merge path on the GPU (tests/test_gpu_cli.py runs it on the
 * golden fixtures).  Test infrastructure / integration demo: nothing in the product links or executes this file.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "rb3gpu.h"
#include "mrope.h"   /* the reference's own headers, from -I/root/reference: types only */
#include "rld0.h"

#define RB3_BOX_MAGIC 0x3130555047334252ull /* "RB3GPU01" */

typedef struct { uint64_t magic; rb3gpu_t *g; } box_t;

static box_t *as_box(const void *r) { return r && ((const box_t*)r)->magic == RB3_BOX_MAGIC ? (box_t*)r : 0; }

static void die(const char *what, int code)
{
	fprintf(stderr, "ERROR: %s: %s\n", what, rb3gpu_strerror(code));
	exit(1);
}

mrope_t *__real_rb3_enc_plain2fmr(int64_t len, const uint8_t *bwt, int max_nodes, int block_len, int32_t n_threads);
void __real_rb3_fmi_merge_plain(mrope_t *r, int64_t len, const uint8_t *seq, int n_threads);
rld_t *__real_rb3_enc_fmr2fmd(mrope_t *r, int cbits, int is_free);
void __real_mr_print_bwt(const mrope_t *mr, FILE *fp);
void __real_mr_destroy(mrope_t *r);

mrope_t *__wrap_rb3_enc_plain2fmr(int64_t len, const uint8_t *bwt, int max_nodes, int block_len, int32_t n_threads)
{
	rb3gpu_opt_t opt;
	box_t *b = (box_t*)calloc(1, sizeof(box_t));
	int r;
	(void)max_nodes; (void)block_len; (void)n_threads;
	rb3gpu_opt_init(&opt);
	opt.verbose = 1;
	b->magic = RB3_BOX_MAGIC, b->g = rb3gpu_create(&opt);
	if (b->g == 0) { fprintf(stderr, "ERROR: no usable MI355X; this build of ropebwt3 has its merge path on the GPU\n"); exit(1); }
	if ((r = rb3gpu_from_plain(b->g, len, bwt)) < 0) die("rb3gpu_from_plain", r);
	return (mrope_t*)b;
}

void __wrap_rb3_fmi_merge_plain(mrope_t *r, int64_t len, const uint8_t *seq, int n_threads)
{
	box_t *b = as_box(r);
	int ret;
	if (b == 0) { __real_rb3_fmi_merge_plain(r, len, seq, n_threads); return; } /* a real rope (e.g. loaded with -i): the reference's own merge */
	if ((ret = rb3gpu_merge_plain(b->g, len, seq)) < 0) die("rb3gpu_merge_plain", ret);
}

static int emit_rld(void *d, int c, int64_t l)
{
	void **a = (void**)d;
	return rld_enc((rld_t*)a[0], (rlditr_t*)a[1], l, (uint8_t)c); /* coalesces equal neighbours, rld0.c:153-161 */
}

rld_t *__wrap_rb3_enc_fmr2fmd(mrope_t *r, int cbits, int is_free)
{
	box_t *b = as_box(r);
	rld_t *e;
	rlditr_t ei;
	void *a[2];
	int ret;
	if (b == 0) return __real_rb3_enc_fmr2fmd(r, cbits, is_free);
	e = rld_init(6, cbits > 0 ? cbits : 3);
	rld_itr_init(e, &ei, 0);
	a[0] = e, a[1] = &ei;
	if ((ret = rb3gpu_export_runs(b->g, emit_rld, a)) < 0) die("rb3gpu_export_runs", ret);
	rld_enc_finish(e, &ei);
	if (is_free) { rb3gpu_destroy(b->g); b->g = 0, b->magic = 0; free(b); }
	return e;
}

static int emit_plain(void *d, int c, int64_t l)
{
	int64_t j;
	for (j = 0; j < l; ++j) fputc("$ACGTN"[c], (FILE*)d);
	return 0;
}

void __wrap_mr_print_bwt(const mrope_t *mr, FILE *fp)
{
	box_t *b = as_box(mr);
	int ret;
	if (b == 0) { __real_mr_print_bwt(mr, fp); return; }
	if ((ret = rb3gpu_export_runs(b->g, emit_plain, fp)) < 0) die("rb3gpu_export_runs", ret);
	fputc('\n', fp);
}

void __wrap_mr_destroy(mrope_t *r)
{
	box_t *b = as_box(r);
	if (b == 0) { __real_mr_destroy(r); return; }
	rb3gpu_destroy(b->g);
	b->magic = 0;
	free(b);
}
