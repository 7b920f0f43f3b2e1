/*
 * main.c -- `ropebwt3-amd`: the `build` command of ropebwt3 with the merge path running on an
 * MI355X through the C ABI of include/rb3gpu.h.  Same options, same input handling and the
 * same output bytes as the reference's build.c:136-263 (FMD and plain output are byte-identical;
 * FMR is loadable by the reference but, as in the reference, not canonical).
 *
 * Host side stays C: sequence parsing (seqio.c), per-batch suffix sorting (sais.c), FMD/FMR
 * codecs (fmd.c, fmr.c).  What build.c does on an mrope_t* is done on an rb3gpu_t*:
 *   rb3_enc_plain2fmr  (build.c:77,223)  -> rb3gpu_from_plain
 *   rb3_fmi_merge_plain (build.c:78,226) -> rb3gpu_merge_plain
 *   rb3_enc_fmd2fmr / mr_restore (180-181) -> rb3h_index_read_runs + rb3gpu_from_runs
 *   rb3_enc_fmr2fmd + rld_dump (249-252), mr_dump (247), mr_print_bwt (254) -> rb3gpu_export_runs
 *     feeding rb3h_fmdw_* / rb3h_fmrw_* / the plain printer
 * There is no CPU merge path in this program: without a HIP device it exits with an error.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <getopt.h>
#include <pthread.h>
#include "rb3host.h"
#include "rb3gpu.h"

#define BF_NO_FOR 0x1
#define BF_NO_REV 0x2
#define BF_LINE   0x4

enum { FMT_PLAIN = 0, FMT_FMD, FMT_FMR };

typedef struct {
	int64_t flag, batch_size;
	int fmt, n_threads, sais_threads, block_len, max_nodes;
	int device, n_gpus, split_log2, rebatch, gpu_sort, host_fmd, interval;
	int64_t gpu_batch;      /* with GPU suffix sorting a batch (-m) is cut into sub-batches of at most this many symbols, at record
	                           boundaries: the .fmd does not depend on the batching, and the GPU sorter takes < 2^31 */
	int64_t gpu_sort_limit; /* batches of this many symbols or more go to the host sorter (one record longer than a sub-batch) */
} bopt_t;

/* how the batches of this run were sorted (for the closing statistics line) */
static struct { int64_t n_gpu, n_host, sym_gpu, sym_host; double ms_upload, ms_sort; } g_sorted;
static pthread_mutex_t g_sorted_mtx = PTHREAD_MUTEX_INITIALIZER; /* (the slices of a multi-GPU build end at different times) */
/* what the handles of the other slices of a multi-GPU build did, added up before they are destroyed (the closing statistics) */
static struct { double ms_path; int64_t n_sym; } g_other_slices;

static void bopt_init(bopt_t *o) /* build.c:31-41 */
{
	memset(o, 0, sizeof(*o));
	o->n_threads = 4, o->sais_threads = -1, o->fmt = FMT_PLAIN;
	o->block_len = 512, o->max_nodes = 64, o->batch_size = 7000000000LL;
	o->device = 0, o->n_gpus = 1, o->split_log2 = 0, o->rebatch = 0, o->gpu_sort = 1, o->host_fmd = 0;
	o->gpu_batch = 1LL << 29, o->gpu_sort_limit = (int64_t)INT32_MAX - 16;
}

/* the size at which the reader cuts batches: -m, or the GPU sub-batch size if that is smaller */
static int64_t batch_cut(const bopt_t *o)
{
	if (o->gpu_sort && o->gpu_batch > 0 && (o->batch_size <= 0 || o->gpu_batch < o->batch_size)) return o->gpu_batch;
	return o->batch_size;
}

static int usage_build(FILE *fp, const bopt_t *opt)
{
	fprintf(fp, "Usage: ropebwt3-amd build [options] <in.fa> [...]\n");
	fprintf(fp, "Options:\n");
	fprintf(fp, "  Algorithm:\n");
	fprintf(fp, "    -m NUM      batch size [7G]\n");
	fprintf(fp, "    -t INT      total number of threads [%d]\n", opt->n_threads);
	fprintf(fp, "    -p INT      sort INT batches at once ahead of the GPU merge: sorter threads with a GPU sorter each (at most 3;\n");
	fprintf(fp, "                default 1) or, with --host-sort, host sorter threads (default 0: sort and merge in turn)\n");
	fprintf(fp, "    -l INT      leaf block size in B+-tree (FMR output only) [%d]\n", opt->block_len);
	fprintf(fp, "    -n INT      max number children per internal node (FMR output only) [%d]\n", opt->max_nodes);
	fprintf(fp, "    --gpu INT   HIP device ordinal [%d]\n", opt->device);
	fprintf(fp, "    --gpus INT  build on INT GPUs: the input files are cut into INT contiguous slices, every GPU indexes one (devices --gpu,\n");
	fprintf(fp, "                --gpu + 1, ...), and the indexes are merged pairwise in input order (same output) [1]\n");
	fprintf(fp, "    --interval  with --gpus INT: ONE index cut into INT intervals of positions, one per GPU; every batch after the first is\n");
	fprintf(fp, "                merged by all GPUs in lock step, its strings' LF chains hopping between the GPUs that own their insertion\n");
	fprintf(fp, "                points (for batches of short strings: one round per symbol of the longest one; same output)\n");
	fprintf(fp, "    --split INT start extra LF walkers every 2^INT rows (0=auto, -1=never) [%d]\n", opt->split_log2);
	fprintf(fp, "    --rebatch   let a batch span input files (same output, fewer merge rounds)\n");
	fprintf(fp, "    --host-sort suffix-sort the batches on the host (default: on the GPU; same output; -p then sets the number\n");
	fprintf(fp, "                of host sorter threads)\n");
	fprintf(fp, "    --gpu-batch NUM  with GPU sorting, cut the batches of -m into sub-batches of at most NUM symbols at record\n");
	fprintf(fp, "                boundaries (the output does not depend on the batching; the GPU sorter takes < 2^31) [512M]\n");
	fprintf(fp, "    --host-fmd  pack/unpack FMD files on the host even where the GPU could\n");
	fprintf(fp, "  Input:\n");
	fprintf(fp, "    -i FILE     read existing index from FILE []\n");
	fprintf(fp, "    -L          one sequence per line in the input\n");
	fprintf(fp, "    -F          no forward strand\n");
	fprintf(fp, "    -R          no reverse strand\n");
	fprintf(fp, "  Output:\n");
	fprintf(fp, "    -o FILE     output to FILE [stdout]\n");
	fprintf(fp, "    -d          dump in the fermi-delta format (FMD)\n");
	fprintf(fp, "    -b          dump in the ropebwt format (FMR)\n");
	fprintf(fp, "    -S FILE     save the current index to FILE after each input file []\n");
	fprintf(fp, "  Not available in this build (ropebwt2 insertion and debugging formats): -2 -s -r -T -e\n");
	return fp == stdout ? 0 : 1;
}

/* ---- batch buffers in page-locked memory ------------------------------------------------------
 * The reader writes every batch straight into memory from rb3gpu_pinned_alloc, so its way to HBM is one DMA at PCIe speed
 * (through pageable memory a single host thread has to stage it first, at half that or less: 88 of the 437 ms of the merge
 * path of a 152-genome build).  Obtaining page-locked memory is slow (it is mapped for the device), so the buffers of
 * finished batches go on a free list and the reader takes them from there: a build allocates a handful, at its start. */
#define PINPOOL_MAX 8
#define PINPOOL_SLOTS 64
static struct { pthread_mutex_t mtx; void *p[PINPOOL_SLOTS]; int64_t cap[PINPOOL_SLOTS]; int busy[PINPOOL_SLOTS]; int n; } g_pin = { PTHREAD_MUTEX_INITIALIZER, {0}, {0}, {0}, 0 };
static int64_t g_pin_limit = 0; /* RB3_PINNED_LIMIT (tests): requests above it are answered as if the runtime had none */
static volatile int g_pin_off = 0; /* page-locked memory could not be had: pageable buffers from then on */

static void *pin_alloc(int64_t min_bytes, int64_t *cap)
{
	int i, best = -1, n_free = 0;
	void *p = 0;
	pthread_mutex_lock(&g_pin.mtx);
	for (i = 0; i < g_pin.n; ++i)
		if (!g_pin.busy[i]) {
			++n_free;
			if (g_pin.cap[i] >= min_bytes && (best < 0 || g_pin.cap[i] < g_pin.cap[best])) best = i;
		}
	if (best >= 0) g_pin.busy[best] = 1, p = g_pin.p[best], *cap = g_pin.cap[best];
	else if (n_free > 0) { /* nothing large enough: the small ones only hold on to page-locked memory */
		for (i = 0; i < g_pin.n; ++i)
			if (!g_pin.busy[i]) {
				rb3gpu_pinned_free(g_pin.p[i]);
				g_pin.p[i] = g_pin.p[g_pin.n - 1], g_pin.cap[i] = g_pin.cap[g_pin.n - 1], g_pin.busy[i] = g_pin.busy[g_pin.n - 1];
				--g_pin.n, --i;
			}
	}
	pthread_mutex_unlock(&g_pin.mtx);
	if (p) return p;
	if (min_bytes < (16 << 20)) min_bytes = 16 << 20; /* (a batch of one short line still gets a buffer worth keeping) */
	/* Page-locked memory is an optimisation (one DMA per batch), not a requirement: where the runtime cannot supply it (a multi-GB
	 * batch with --host-sort) or the table is full (many --gpus slices), the batch lives in pageable memory, which the engine
	 * stages (rb3gpu.h: "NULL: use malloc").  pin_release() tells the two apart by the table. */
	p = g_pin_off || (g_pin_limit > 0 && min_bytes > g_pin_limit) ? 0 : rb3gpu_pinned_alloc(min_bytes);
	if (p != 0) {
		pthread_mutex_lock(&g_pin.mtx);
		if (g_pin.n < PINPOOL_SLOTS) g_pin.p[g_pin.n] = p, g_pin.cap[g_pin.n] = min_bytes, g_pin.busy[g_pin.n] = 1, ++g_pin.n;
		else { rb3gpu_pinned_free(p); p = 0; }
		pthread_mutex_unlock(&g_pin.mtx);
	} else if (!g_pin_off && g_pin_limit <= 0) {
		g_pin_off = 1; /* (asking again for every batch would cost a failed system call each time) */
		if (rb3h_verbose >= 2) fprintf(stderr, "WARNING: no page-locked memory for a batch of %ld bytes; batches are staged from pageable memory from here on\n", (long)min_bytes);
	}
	if (p == 0 && (p = malloc((size_t)min_bytes)) == 0) return 0;
	*cap = min_bytes;
	return p;
}

static void pin_release(void *p)
{
	int i, n_free = 0, pooled = 0;
	pthread_mutex_lock(&g_pin.mtx);
	for (i = 0; i < g_pin.n; ++i) {
		if (g_pin.p[i] == p) g_pin.busy[i] = 0, pooled = 1;
		n_free += !g_pin.busy[i];
	}
	if (!pooled) { /* a pageable stand-in (pin_alloc): a busy page-locked buffer is always in the table */
		pthread_mutex_unlock(&g_pin.mtx);
		free(p);
		return;
	}
	if (n_free > PINPOOL_MAX) { /* give the smallest idle one back */
		int k = -1;
		for (i = 0; i < g_pin.n; ++i)
			if (!g_pin.busy[i] && (k < 0 || g_pin.cap[i] < g_pin.cap[k])) k = i;
		if (k >= 0) {
			rb3gpu_pinned_free(g_pin.p[k]);
			g_pin.p[k] = g_pin.p[g_pin.n - 1], g_pin.cap[k] = g_pin.cap[g_pin.n - 1], g_pin.busy[k] = g_pin.busy[g_pin.n - 1];
			--g_pin.n;
		}
	}
	pthread_mutex_unlock(&g_pin.mtx);
}

static void pin_drain(void)
{
	int i;
	pthread_mutex_lock(&g_pin.mtx);
	for (i = 0; i < g_pin.n; ++i) rb3gpu_pinned_free(g_pin.p[i]);
	g_pin.n = 0;
	pthread_mutex_unlock(&g_pin.mtx);
}

/* ---- run sinks --------------------------------------------------------------------------- */

typedef struct { uint64_t *a; int64_t n, m; } runvec_t;

static int sink_runvec(void *data, int c, int64_t l)
{
	runvec_t *v = (runvec_t*)data;
	if (v->n == v->m) {
		v->m = v->m ? v->m * 2 : 1 << 16;
		v->a = (uint64_t*)realloc(v->a, (size_t)v->m * 8);
		if (v->a == 0) return -1;
	}
	v->a[v->n++] = (uint64_t)l << 3 | (uint64_t)c;
	return 0;
}

static int sink_fmd_words(void *data, int64_t n, const uint64_t *words, int64_t end) { return rb3h_fmdw_enc_words((rb3h_fmdw_t*)data, n, words, end); }

/* the .fmd of the index (rb3_enc_fmr2fmd + rld_dump, build.c:248-252): the data section is packed on the GPU (16-bit and
 * 32-bit block headers); if a block needs a 64-bit header or the device has no room for the packer, the GPU finds the runs
 * and the host packs them; the rank index is built here */
static int g_host_fmd = 0; /* --host-fmd */

static int write_fmd(rb3gpu_t *h, FILE *fp)
{
	rb3h_fmdw_t *w = rb3h_fmdw_init();
	uint64_t *words = 0;
	int64_t n_words = 0, acc[7];
	int ret;
	if (w == 0) return -1;
	ret = g_host_fmd ? RB3GPU_EUNSUP : rb3gpu_export_fmd_words(h, &words, &n_words);
	if (ret == 0) {
		rb3gpu_get_acc(h, acc);
		ret = rb3h_fmdw_adopt(w, words, n_words, acc); /* takes the array over */
		if (ret < 0) rb3gpu_host_free(words);
		else if (rb3h_verbose >= 3) fprintf(stderr, "[M::%s::%.3f*%.2f] packed the FMD on the GPU\n", __func__, rb3h_realtime(), rb3h_percent_cpu());
	} else if (ret != RB3GPU_ESTATE && ret != RB3GPU_EINVAL) { /* wider block headers needed, no room to pack the whole index at once -- or the packer gave up: the host's encoder takes the runs chunk by chunk */
		if (ret != RB3GPU_EUNSUP && ret != RB3GPU_ENOMEM && rb3h_verbose >= 2) fprintf(stderr, "[W::%s] the GPU's FMD packer failed (%s); encoding on the host\n", __func__, rb3gpu_strerror(ret));
		ret = rb3gpu_export_run_words(h, sink_fmd_words, w);
		if (ret == 0) ret = rb3h_fmdw_finish(w);
	}
	if (ret == 0) ret = rb3h_fmdw_dump(w, fp);
	rb3h_fmdw_destroy(w);
	return ret;
}
static int sink_fmr(void *data, int c, int64_t l) { return rb3h_fmrw_enc((rb3h_fmrw_t*)data, l, c); }

static int sink_plain(void *data, int c, int64_t l) /* mr_print_bwt, mrope.c:201-214 */
{
	FILE *fp = (FILE*)data;
	char buf[4096];
	memset(buf, "$ACGTN"[c], l < 4096 ? (size_t)l : 4096);
	while (l > 0) {
		size_t t = l < 4096 ? (size_t)l : 4096;
		if (fwrite(buf, 1, t, fp) != t) return -1;
		l -= (int64_t)t;
	}
	return 0;
}

static int iv_live(void);
static int iv_get_acc(int64_t acc[7]);
static int iv_export_runs(rb3gpu_emit_f sink, void *data);

/* the index as an FMR file -- from the handle, or, once --interval has cut it, from the intervals in rank order where they are
 * (-S after each input file, build.c:232-238, and the final -r output) */
static int dump_fmr(rb3gpu_t *h, const bopt_t *opt, FILE *fp)
{
	int64_t acc[7];
	rb3h_fmrw_t *w;
	int ret;
	if (iv_live()) iv_get_acc(acc);
	else rb3gpu_get_acc(h, acc);
	w = rb3h_fmrw_init(acc, opt->max_nodes, opt->block_len);
	if (w == 0) return -1;
	ret = iv_live() ? iv_export_runs(sink_fmr, w) : rb3gpu_export_runs(h, sink_fmr, w);
	if (ret == 0) ret = rb3h_fmrw_dump(w, fp);
	rb3h_fmrw_destroy(w);
	return ret;
}

/* an existing index into HBM (build.c:172-184, rb3_fmi_restore): an FMD file is decoded on the device
 * (rb3gpu_from_fmd_words); an FMR file, a stream, or an FMD the device declined goes through the host decoder */
static int load_index(rb3gpu_t *h, const char *fn)
{
	runvec_t rv = {0, 0, 0};
	uint64_t *z = 0;
	int64_t nw = 0, mc[6];
	int r = g_host_fmd ? 1 : rb3h_fmd_read_words(fn, &z, &nw, mc);
	if (r < 0) return -1;
	if (r == 0) {
		r = rb3gpu_from_fmd_words(h, nw, z, mc);
		free(z);
		if (r == 0) return 0;
		if (rb3h_verbose >= 2) fprintf(stderr, "[W::%s] the GPU did not decode '%s' (%s); decoding it on the host\n", __func__, fn, rb3gpu_strerror(r));
	}
	if (rb3h_index_read_runs(fn, sink_runvec, &rv) < 0 || rv.n == 0) { free(rv.a); return -1; }
	r = rb3gpu_from_runs(h, rv.n, rv.a);
	free(rv.a);
	return r < 0 ? -2 : 0;
}

/* ---- batches ----------------------------------------------------------------------------- */

typedef struct {
	int64_t n_seq, len, n_walkers, step;
	uint8_t *bwt;             /* host: the BWT, or the text if raw */
	rb3h_walker_t *walkers;
	int walkers_pinned;       /* the list lives in a buffer of the page-locked pool (pin_release, not free) */
	int ret, raw;             /* raw: not sorted yet, the consumer's GPU handle sorts it */
	void *d_bwt;              /* device: the BWT from a sorter thread's own GPU sorter (gs), to be released after the merge */
	void *d_tw;               /* device: its text-order words (long strings: the walkers are then given by text position) */
	void *d_sa;               /* device: its suffix array, behind them (the engine leaves its records in text order where that pays) */
	rb3gpu_sorter_t *gs;
	int64_t *sent, n_sent;    /* --interval: text positions of the batch's sentinels (where its LF chains start) */
} batch_t;

static int g_pin_on = 0; /* batch buffers (and walker lists) in page-locked memory */

/* The walker list goes to the device inside the merge call, between the LF kernels and the walkers: out of page-locked memory
 * that is one DMA; out of malloc'd memory the engine first copies it into its staging buffer while the device waits. */
static void walkers_pin(batch_t *b)
{
	int64_t cap = 0;
	const int64_t bytes = b->n_walkers * (int64_t)sizeof(rb3h_walker_t);
	void *p;
	if (!g_pin_on || b->walkers == 0 || b->walkers_pinned || bytes <= 0 || bytes > (16 << 20)) return;
	if ((p = pin_alloc(bytes, &cap)) == 0) return;
	memcpy(p, b->walkers, (size_t)bytes);
	free(b->walkers);
	b->walkers = (rb3h_walker_t*)p, b->walkers_pinned = 1;
}

static void walkers_free(batch_t *b)
{
	if (b->walkers_pinned) pin_release(b->walkers);
	else free(b->walkers);
	b->walkers = 0, b->walkers_pinned = 0;
}

/* --gpus N --interval: the index lives in N intervals on N devices from the second batch on (rb3gpu_shard_*, include/rb3gpu.h) */
static struct { int n, devices[RB3GPU_SH_MAXIV]; rb3gpu_shard_t *s; rb3gpu_opt_t gopt; double t_walk; int64_t rounds, batches; } g_iv;

static int64_t *sentinels_of(const uint8_t *text, int64_t len, int64_t n_hint, int64_t *n_out)
{
	int64_t n = 0, m = n_hint > 16 ? n_hint : 16, *a = (int64_t*)malloc((size_t)m * 8);
	const uint8_t *p = text, *end = text + len;
	while (a && p < end && (p = (const uint8_t*)memchr(p, 0, (size_t)(end - p))) != 0) {
		if (n == m) { int64_t *t = (int64_t*)realloc(a, (size_t)(m *= 2) * 8); if (t == 0) { free(a); a = 0; break; } a = t; }
		a[n++] = p - text, ++p;
	}
	*n_out = a ? n : 0;
	return a;
}

static int iv_live(void) { return g_iv.s != 0; }
static int iv_get_acc(int64_t acc[7]) { return rb3gpu_shard_get_acc(g_iv.s, acc); }
static int iv_export_runs(rb3gpu_emit_f sink, void *data) { return rb3gpu_shard_export_runs(g_iv.s, sink, data); }

/* one batch into the sharded index; the first call cuts the index the handle holds into its intervals */
static int interval_merge(rb3gpu_t *h, int64_t len, const void *d_bwt, const void *d_tw, int64_t n_sent, const int64_t *sent)
{
	int64_t rounds = 0;
	int r;
	const double t0 = rb3h_realtime();
	if (sent == 0 || n_sent <= 0 || d_tw == 0) return RB3GPU_EINVAL;
	if (rb3h_verbose >= 2) { /* one lock-step round per symbol of the longest string: say so before a batch of long strings takes minutes */
		int64_t i, longest = sent[0] + 1;
		for (i = 1; i < n_sent; ++i) if (sent[i] - sent[i - 1] > longest) longest = sent[i] - sent[i - 1];
		if (longest > 100000) fprintf(stderr, "WARNING: --interval walks a batch in lock step, one round (~20 us or more) per symbol of its longest string: %ld rounds for this batch; it is meant for reads (without --interval the strings are cut among many walkers)\n", (long)longest);
	}
	if (g_iv.s == 0) {
		if (rb3gpu_get_tot(h) < g_iv.n) return rb3gpu_merge_text_dev(h, len, (const uint8_t*)d_bwt, (const uint64_t*)d_tw, n_sent, 0, 1); /* (fewer symbols than intervals: not cut yet) */
		if ((g_iv.s = rb3gpu_shard_split(h, g_iv.n, g_iv.devices, &g_iv.gopt)) == 0) return RB3GPU_ENODEV;
		if (rb3h_verbose >= 3) fprintf(stderr, "[M::%s::%.3f*%.2f] index cut into %d intervals\n", "main_build", rb3h_realtime(), rb3h_percent_cpu(), g_iv.n);
	}
	r = rb3gpu_shard_merge(g_iv.s, len, (const uint8_t*)d_bwt, (const uint64_t*)d_tw, n_sent, sent, &rounds);
	g_iv.t_walk += rb3h_realtime() - t0, g_iv.rounds += rounds, ++g_iv.batches;
	return r;
}

/* --gpu-sort: the batch arrives as text; suffix sorting, BWT and inverse suffix array on the GPU
 * (rb3gpu_sort_text / rb3gpu_bwt_from_text instead of rb3_build_sais, build.c:220), the BWT never leaves HBM */
static int process_raw_batch(rb3gpu_t *h, batch_t *b, int *has_index)
{
	void *d_bwt = 0, *d_tw = 0;
	const int text_walk = *has_index; /* text-order words; long strings: walkers by text position, short ones: one walker per string */
	int ret;
	/* only "no room on the device" sends the batch to the host sorter (return 1); any other failure is an error of the build */
	if ((ret = rb3gpu_dev_alloc(h, b->len + 16, &d_bwt)) < 0) return ret == RB3GPU_ENOMEM ? 1 : ret;
	if (text_walk && (ret = rb3gpu_dev_alloc(h, b->len * 8, &d_tw)) < 0) { rb3gpu_dev_free(h, d_bwt); return ret == RB3GPU_ENOMEM ? 1 : ret; }
	if (text_walk) ret = rb3gpu_sort_text(h, b->len, b->bwt, (uint8_t*)d_bwt, (uint64_t*)d_tw);
	else ret = rb3gpu_bwt_from_text(h, b->len, b->bwt, (uint8_t*)d_bwt, 0, 0);
	if (ret == RB3GPU_ENOMEM) { /* the sorter's scratch did not fit: the caller sorts this batch on the host */
		if (d_tw) rb3gpu_dev_free(h, d_tw);
		rb3gpu_dev_free(h, d_bwt);
		return 1;
	}
	if (ret == 0) __sync_fetch_and_add(&g_sorted.n_gpu, 1), __sync_fetch_and_add(&g_sorted.sym_gpu, b->len);
	if (ret == 0 && rb3h_verbose >= 3)
		fprintf(stderr, "[M::%s::%.3f*%.2f] constructed partial BWT for %ld symbols on the GPU\n", "main_build", rb3h_realtime(), rb3h_percent_cpu(), (long)b->len);
	if (ret == 0 && !*has_index) ret = rb3gpu_from_plain_dev(h, b->len, (const uint8_t*)d_bwt);
	else if (ret == 0 && text_walk && g_iv.n > 1) {
		int64_t n_sent = 0, *sent = sentinels_of(b->bwt, b->len, b->n_seq, &n_sent);
		ret = sent ? interval_merge(h, b->len, d_bwt, d_tw, n_sent, sent) : RB3GPU_ENOMEM;
		free(sent);
	} else if (ret == 0 && text_walk) {
		if (b->step > 0) ret = rb3gpu_merge_text_step_dev(h, b->len, (const uint8_t*)d_bwt, (const uint64_t*)d_tw, 0, b->n_seq, b->step, 1); /* long strings: the walker list (one per string, one every `step` positions) is made on the device */
		else ret = rb3gpu_merge_text_dev(h, b->len, (const uint8_t*)d_bwt, (const uint64_t*)d_tw, b->n_seq, 0, 1); /* short strings: one walker per string */
	} else if (ret == 0) ret = rb3gpu_merge_plain_dev(h, b->len, (const uint8_t*)d_bwt, 1);
	if (d_tw) rb3gpu_dev_free(h, d_tw);
	rb3gpu_dev_free(h, d_bwt);
	return ret;
}

static int process_batch(rb3gpu_t *h, batch_t *b, int *has_index)
{
	int ret;
	if (b->d_bwt) { /* sorted on the GPU by a sorter thread while the batch before was being merged */
		const int first = !*has_index;
		if (first) ret = rb3gpu_from_plain_dev(h, b->len, (const uint8_t*)b->d_bwt);
		else if (g_iv.n > 1) ret = interval_merge(h, b->len, b->d_bwt, b->d_tw, b->n_sent, b->sent);
		else if (b->walkers && b->d_tw) ret = rb3gpu_merge_text_sa_dev(h, b->len, (const uint8_t*)b->d_bwt, (const uint64_t*)b->d_tw, (const uint32_t*)b->d_sa, b->n_walkers, (const rb3gpu_walker_t*)b->walkers, 1); /* (RB3_HOST_WALKERS: the list of rounds 2-4, made by the sorter thread) */
		else if (b->d_tw && b->step > 0 && b->n_seq > 0) ret = rb3gpu_merge_text_step_dev(h, b->len, (const uint8_t*)b->d_bwt, (const uint64_t*)b->d_tw, (const uint32_t*)b->d_sa, b->n_seq, b->step, 1); /* long strings: walker list made on the device */
		else if (b->d_tw && b->step == 0 && b->n_seq > 0) ret = rb3gpu_merge_text_sa_dev(h, b->len, (const uint8_t*)b->d_bwt, (const uint64_t*)b->d_tw, (const uint32_t*)b->d_sa, b->n_seq, 0, 1); /* short strings: one walker per string */
		else ret = rb3gpu_merge_plain_dev(h, b->len, (const uint8_t*)b->d_bwt, 1);
		rb3gpu_sorter_release(b->gs, b->d_bwt);
		b->d_bwt = b->d_tw = b->d_sa = 0;
		if (ret == 0 && rb3h_verbose >= 3)
			fprintf(stderr, "[M::%s::%.3f*%.2f] %s the partial BWT for %ld symbols\n", "main_build", rb3h_realtime(), rb3h_percent_cpu(), first ? "encoded" : "merged", (long)b->len);
	} else if (b->raw) {
		const int first = !*has_index;
		ret = process_raw_batch(h, b, has_index);
		if (ret == 1) { /* host sorter instead */
			if (rb3h_verbose >= 2) fprintf(stderr, "[W::%s] the GPU suffix sorter could not take a batch of %ld symbols; sorting it on the host\n", "main_build", (long)b->len);
			ret = b->step > 0 ? rb3h_build_bwt_walkers(b->n_seq, b->len, b->bwt, 1, b->step, &b->n_walkers, &b->walkers) : rb3h_build_bwt(b->n_seq, b->len, b->bwt, 1);
			if (ret < 0) { fprintf(stderr, "ERROR: failed to construct the partial BWT (code %d)\n", ret); return -1; }
			b->raw = 0;
			return process_batch(h, b, has_index);
		}
		if (ret == 0 && rb3h_verbose >= 3)
			fprintf(stderr, "[M::%s::%.3f*%.2f] %s the partial BWT for %ld symbols\n", "main_build", rb3h_realtime(), rb3h_percent_cpu(), first ? "encoded" : "merged", (long)b->len);
	} else if (!*has_index) {
		ret = rb3gpu_from_plain(h, b->len, b->bwt);
		if (ret == 0 && rb3h_verbose >= 3)
			fprintf(stderr, "[M::%s::%.3f*%.2f] encoded the partial BWT for %ld symbols\n", "main_build", rb3h_realtime(), rb3h_percent_cpu(), (long)b->len);
	} else {
		/* long strings: hand over the sampled inverse suffix array as LF walkers (same result, text-regular
		 * parallelism); short strings (reads): one walker per string is what the engine does by itself */
		if (g_iv.s) { /* --interval and a batch the HOST had to sort (a record beyond the GPU sorter's limit, no device room): there are no text-order
		                 words to walk the intervals with, so the intervals go back into one handle, this batch is merged the ordinary way, and the
		                 next GPU-sorted batch cuts the index again (a long build must not die of one such batch) */
			const int r = rb3gpu_shard_gather(g_iv.s);
			g_iv.s = 0;
			if (r < 0) { fprintf(stderr, "ERROR: the GPU engine failed to put the intervals together for a host-sorted batch: %s\n", rb3gpu_strerror(r)); return -1; }
			if (rb3h_verbose >= 2) fprintf(stderr, "[W::%s] --interval: a host-sorted batch of %ld symbols is merged on one device; the index is cut again afterwards\n", "main_build", (long)b->len);
		}
		if (b->walkers) ret = rb3gpu_merge_plain_walkers(h, b->len, b->bwt, b->n_walkers, (const rb3gpu_walker_t*)b->walkers);
		else ret = rb3gpu_merge_plain(h, b->len, b->bwt);
		if (ret == 0 && rb3h_verbose >= 3)
			fprintf(stderr, "[M::%s::%.3f*%.2f] merged the partial BWT for %ld symbols\n", "main_build", rb3h_realtime(), rb3h_percent_cpu(), (long)b->len);
	}
	if (ret < 0) {
		fprintf(stderr, "ERROR: the GPU engine failed on a batch of %ld symbols: %s\n", (long)b->len, rb3gpu_strerror(ret));
		return ret;
	}
	*has_index = 1;
	return 0;
}

/* Pipeline (build.c:55-83, 186-201, generalised).  The reference overlaps the suffix sorting of batch
 * i+1 with the merge of batch i.  Here the merge takes milliseconds and the suffix sorter is the slow
 * stage, so `-p N` runs N sorter threads: one reader thread cuts the batches in input order, the
 * sorters work on N batches at once, and the calling thread feeds the GPU strictly in input order. */
typedef struct {
	rb3h_buf_t seq;      /* raw text of the batch */
	int64_t n_seq;
	batch_t *out;        /* sorted batch */
	int state;           /* 0 free, 1 raw, 2 being sorted, 3 ready (or an end-of-file marker), */
	int end_of_file, err;
} job_t;

typedef struct {
	pthread_mutex_t mtx;
	pthread_cond_t cv;
	job_t *ring;
	int cap;
	int64_t head, tail, next_sort; /* next to consume / to fill / to sort */
	int reader_done;
	const bopt_t *opt;
	int device;
} pool_t;

static int sort_batch(const bopt_t *opt, int device, rb3h_buf_t *seq, int64_t n_seq, int n_threads, batch_t **out, rb3gpu_sorter_t *gs)
{
	batch_t *b;
	int64_t n_walkers = 0;
	rb3h_walker_t *walkers = 0;
	/* walkers inside long strings: as many as the walker kernel keeps resident on the GPU (rb3gpu_walker_step), or every 2^k positions (-k) */
	int64_t step = opt->split_log2 > 0 ? 1LL << opt->split_log2 : rb3gpu_walker_step(device, seq->l, n_seq); /* (the device of this slice: --gpus N) */
	int r;
	if (step < 192 && opt->split_log2 <= 0) step = 384; /* (no such device: the merge will say so) */
	if (step < (seq->l >> 20)) step = seq->l >> 20; /* at most ~2^20 walkers per batch: the engine's stretch table is finite */
	if (opt->gpu_sort && seq->l < opt->gpu_sort_limit) { /* the GPU sorts (its sorter handles < 2^31 symbols; batches are cut to fit, see batch_cut) */
		b = (batch_t*)calloc(1, sizeof(batch_t)); /* (counted in g_sorted where the sort really happens: below, or in process_batch) */
		b->n_seq = n_seq, b->len = seq->l, b->bwt = seq->s, b->raw = 1;
		b->step = (opt->split_log2 >= 0 && n_seq > 0 && seq->l / n_seq > 4 * step && seq->l / step + n_seq < (1 << 22)) ? step : 0;
		seq->s = 0, seq->l = seq->m = 0;
		if (gs) { /* this thread has a GPU sorter of its own: sort now, while the consumer merges the batch before */
			/* long strings: also the text-order words, and LF walkers by text position (same result, text-regular parallelism);
			 * short strings (reads): one walker per string is what the engine does by itself */
			/* a few long records on both strands: only the forward strands cross PCIe, the reverse complements are made on the device */
			int64_t pair_start[32], n_pairs = 0;
			int r2 = -1;
			if (!(opt->flag & (BF_NO_FOR | BF_NO_REV))) n_pairs = rb3h_strand_pairs(b->len, b->bwt, n_seq, 32, pair_start);
			if (n_pairs > 0 && (r2 = rb3gpu_sorter_upload_fwd(gs, b->len, b->bwt, n_pairs, pair_start)) == 0)
				r2 = rb3gpu_sorter_sort_uploaded_sa(gs, b->len, &b->d_bwt, &b->d_tw, &b->d_sa);
			if (n_pairs <= 0 || r2 == RB3GPU_EINVAL) r2 = rb3gpu_sorter_sort_sa(gs, b->len, b->bwt, &b->d_bwt, &b->d_tw, &b->d_sa);
			if (r2 == 0) {
				if (rb3h_verbose >= 3)
					fprintf(stderr, "[M::%s::%.3f*%.2f] constructed partial BWT for %ld symbols on the GPU\n", "main_build", rb3h_realtime(), rb3h_percent_cpu(), (long)b->len);
				if (b->step > 0 && getenv("RB3_HOST_WALKERS") && rb3h_walkers_text(b->len, b->bwt, b->step, &b->n_walkers, &b->walkers) < 0) b->walkers = 0, b->n_walkers = 0; /* (experiments: the host's list) */
				walkers_pin(b);
				b->gs = gs, b->raw = 0; /* (the walker list of a batch of long strings is made on the device, inside the merge call: b->n_seq and b->step say how) */
				if (opt->interval) b->sent = sentinels_of(b->bwt, b->len, n_seq, &b->n_sent);
				__sync_fetch_and_add(&g_sorted.n_gpu, 1), __sync_fetch_and_add(&g_sorted.sym_gpu, b->len);
				rb3h_batch_free(b->bwt); b->bwt = 0; /* the text is not needed any more */
			} else b->d_bwt = b->d_tw = b->d_sa = 0; /* leave it to the consumer (its handle's sorter, then the host sorter) */
		}
		*out = b;
		return 0;
	}
	__sync_fetch_and_add(&g_sorted.n_host, 1), __sync_fetch_and_add(&g_sorted.sym_host, seq->l);
	if (opt->split_log2 >= 0 && n_seq > 0 && seq->l / n_seq > 4 * step && seq->l / step + n_seq < (1 << 22))
		r = rb3h_build_bwt_walkers(n_seq, seq->l, seq->s, n_threads, step, &n_walkers, &walkers);
	else r = rb3h_build_bwt(n_seq, seq->l, seq->s, n_threads);
	if (r < 0) {
		fprintf(stderr, "ERROR: failed to construct the partial BWT (code %d)\n", r);
		return -1;
	}
	if (rb3h_verbose >= 3)
		fprintf(stderr, "[M::%s::%.3f*%.2f] constructed partial BWT for %ld symbols\n", "main_build", rb3h_realtime(), rb3h_percent_cpu(), (long)seq->l);
	b = (batch_t*)calloc(1, sizeof(batch_t));
	b->n_seq = n_seq, b->len = seq->l, b->bwt = seq->s, b->n_walkers = n_walkers, b->walkers = walkers;
	seq->s = 0, seq->l = seq->m = 0; /* ownership moves to the batch */
	*out = b;
	return 0;
}

/* what to do with a freshly cut batch (raw text in *seq, ownership passes) or an end-of-file event */
typedef int (*submit_f)(void *data, rb3h_buf_t *seq, int64_t n_seq, int end_of_file);

/* read every input file, cutting batches as io.c:104-125 does */
/* rbeg / rend (NULL: whole files): the byte range of file i whose records this reader takes (`build --gpus N`: slices cut inside files) */
static int for_each_batch(const bopt_t *opt, int n_files, char **files, const int64_t *rbeg, const int64_t *rend, submit_f submit, void *data, int64_t *n_empty)
{
	rb3h_buf_t seq = {0, 0, 0};
	int64_t n_seq_acc = 0;
	int i, ret = 0;
	for (i = 0; i < n_files && ret == 0; ++i) {
		rb3h_seqio_t *fp = rbeg ? rb3h_seq_open_range(files[i], !!(opt->flag & BF_LINE), rbeg[i], rend[i]) : rb3h_seq_open(files[i], !!(opt->flag & BF_LINE));
		int64_t n_seq;
		if (fp == 0) {
			if (rb3h_verbose >= 1) fprintf(stderr, "ERROR: failed to open file '%s'\n", files[i]);
			continue; /* build.c:208-211 */
		}
		for (;;) {
			const int64_t l0 = seq.l;
			n_seq = rb3h_seq_read(fp, &seq, batch_cut(opt), !(opt->flag & BF_NO_FOR), !(opt->flag & BF_NO_REV), n_empty);
			if (n_seq < 0) {
				if (rb3h_verbose >= 1) fprintf(stderr, "ERROR: failed to read sequences from '%s' (code %ld)\n", files[i], (long)n_seq);
				ret = -1; /* (an I/O error in mid-file or no memory: indexing the prefix read so far would be a silently wrong index) */
				break;
			}
			if (rb3h_seq_error(fp) && seq.l != l0 && rb3h_verbose >= 1) /* the records before the error are indexed, as in io.c:121-124 */
				fprintf(stderr, "ERROR: FASTX parsing error (code %d)\n", rb3h_seq_error(fp));
			if (n_seq == 0 && seq.l == l0) break; /* EOF */
			n_seq_acc += n_seq;
			if (rb3h_verbose >= 3)
				fprintf(stderr, "[M::%s::%.3f*%.2f] read %ld symbols from file '%s'\n", "main_build", rb3h_realtime(), rb3h_percent_cpu(), (long)(seq.l - l0), files[i]);
			if (opt->rebatch && !(batch_cut(opt) > 0 && seq.l > batch_cut(opt))) break; /* keep filling from the next file */
			if ((ret = submit(data, &seq, n_seq_acc, 0)) != 0) break;
			n_seq_acc = 0;
		}
		rb3h_seq_close(fp);
		if (ret == 0 && !(opt->rebatch && seq.l > 0)) ret = submit(data, 0, 0, 1); /* end of file i */
	}
	if (ret == 0 && seq.l > 0) { /* the last, partly filled re-batched batch */
		if ((ret = submit(data, &seq, n_seq_acc, 0)) == 0) ret = submit(data, 0, 0, 1);
	}
	rb3h_batch_free(seq.s);
	return ret;
}

typedef struct {
	rb3gpu_t *h;
	const bopt_t *opt;
	int has_index;
	const char *fn_tmp;
	int device;
} consumer_t;

static int consume(consumer_t *c, batch_t *b, int end_of_file)
{
	if (b) {
		int r = process_batch(c->h, b, &c->has_index);
		rb3h_batch_free(b->bwt); walkers_free(b); free(b->sent); free(b);
		if (r < 0) return r;
	}
	if (end_of_file && c->fn_tmp && c->has_index) { /* build.c:232-238 */
		FILE *fp = fopen(c->fn_tmp, "wb");
		if (fp != 0) {
			dump_fmr(c->h, c->opt, fp);
			fclose(fp);
			if (rb3h_verbose >= 3) fprintf(stderr, "[M::%s::%.3f*%.2f] saved the current index to '%s'\n", "main_build", rb3h_realtime(), rb3h_percent_cpu(), c->fn_tmp);
		}
	}
	return 0;
}

/* serial mode: sort and merge on the calling thread */
static int submit_serial(void *data, rb3h_buf_t *seq, int64_t n_seq, int end_of_file)
{
	consumer_t *c = (consumer_t*)data;
	batch_t *b = 0;
	if (seq && sort_batch(c->opt, c->device, seq, n_seq, c->opt->n_threads, &b, 0) < 0) return -1;
	return consume(c, b, end_of_file);
}

/* pipelined mode: the reader thread queues raw batches */
static int submit_pool(void *data, rb3h_buf_t *seq, int64_t n_seq, int end_of_file)
{
	pool_t *q = (pool_t*)data;
	job_t *j;
	pthread_mutex_lock(&q->mtx);
	while (q->tail - q->head >= q->cap) pthread_cond_wait(&q->cv, &q->mtx);
	j = &q->ring[q->tail % q->cap];
	memset(j, 0, sizeof(*j));
	if (seq) j->seq = *seq, j->n_seq = n_seq, j->state = 1, seq->s = 0, seq->l = seq->m = 0;
	else j->state = 3, j->end_of_file = 1;
	(void)end_of_file;
	++q->tail;
	pthread_cond_broadcast(&q->cv);
	pthread_mutex_unlock(&q->mtx);
	return 0;
}

typedef struct { pool_t *q; int n_files; char **files; const int64_t *rbeg, *rend; int64_t n_empty; int err; } reader_t;

static void *reader_main(void *arg)
{
	reader_t *r = (reader_t*)arg;
	r->err = for_each_batch(r->q->opt, r->n_files, r->files, r->rbeg, r->rend, submit_pool, r->q, &r->n_empty);
	pthread_mutex_lock(&r->q->mtx);
	r->q->reader_done = 1;
	pthread_cond_broadcast(&r->q->cv);
	pthread_mutex_unlock(&r->q->mtx);
	return 0;
}

typedef struct { pool_t *q; rb3gpu_sorter_t *gs; } sorter_arg_t;

static void *sorter_main(void *arg)
{
	pool_t *q = ((sorter_arg_t*)arg)->q;
	rb3gpu_sorter_t *gs = ((sorter_arg_t*)arg)->gs;
	for (;;) {
		job_t *j = 0;
		pthread_mutex_lock(&q->mtx);
		for (;;) {
			while (q->next_sort < q->tail && q->ring[q->next_sort % q->cap].state != 1) ++q->next_sort; /* markers */
			if (q->next_sort < q->tail) { j = &q->ring[q->next_sort % q->cap]; j->state = 2; ++q->next_sort; break; }
			if (q->reader_done) break;
			pthread_cond_wait(&q->cv, &q->mtx);
		}
		pthread_mutex_unlock(&q->mtx);
		if (j == 0) return 0;
		{
			batch_t *b = 0;
			int err = sort_batch(q->opt, q->device, &j->seq, j->n_seq, 1, &b, gs);
			pthread_mutex_lock(&q->mtx);
			j->out = b, j->err = err, j->state = 3;
			pthread_cond_broadcast(&q->cv);
			pthread_mutex_unlock(&q->mtx);
		}
	}
}

/* one slice of the input on one GPU: reader thread -> sorter thread(s) -> merges, in input order (build.c:186-239).  A build on
 * one GPU is one slice; `--gpus N` runs N of them side by side, one per device, and merges their indexes afterwards. */
typedef struct {
	rb3gpu_t *h;
	const bopt_t *opt;
	int device, n_files, has_index, ret;
	char **files;
	const int64_t *rbeg, *rend; /* NULL: whole files; else the byte range of every file whose records belong to this slice */
	const char *fn_tmp;
	int64_t n_empty;
	rb3gpu_sorter_t *old_sorters[4];
	int n_old_sorters;
} slice_t;

static void *run_slice(void *arg)
{
	slice_t *sl = (slice_t*)arg;
	const bopt_t *opt = sl->opt;
	rb3gpu_t *h = sl->h;
	int ret = 0, has_index = sl->has_index;
	int64_t n_empty = 0;
	const char *fn_tmp = sl->fn_tmp;
	rb3gpu_sorter_t **old_sorters = sl->old_sorters;
	int n_old_sorters = 0;
	const int argc = sl->n_files, optind = 0;
	char **argv = sl->files;
	if (opt->sais_threads > 0 && argc - optind >= 1) { /* N suffix sorters ahead of the GPU merge */
		pool_t q;
		reader_t rd;
		pthread_t rt, *st;
		consumer_t cs = { h, opt, has_index, fn_tmp, sl->device };
		int k, n_sort = opt->sais_threads;
		sorter_arg_t *sa;
		if (opt->gpu_sort && n_sort > 3) n_sort = 3; /* GPU sorters: more than a few at once only compete for the same GPU */
		memset(&q, 0, sizeof(q));
		pthread_mutex_init(&q.mtx, 0);
		pthread_cond_init(&q.cv, 0);
		q.cap = n_sort + 2, q.ring = (job_t*)calloc((size_t)q.cap, sizeof(job_t)), q.opt = opt, q.device = sl->device;
		memset(&rd, 0, sizeof(rd));
		rd.q = &q, rd.n_files = argc - optind, rd.files = argv + optind, rd.rbeg = sl->rbeg, rd.rend = sl->rend;
		st = (pthread_t*)calloc((size_t)n_sort, sizeof(pthread_t));
		sa = (sorter_arg_t*)calloc((size_t)n_sort, sizeof(sorter_arg_t));
		pthread_create(&rt, 0, reader_main, &rd);
		for (k = 0; k < n_sort; ++k) {
			sa[k].q = &q, sa[k].gs = opt->gpu_sort ? rb3gpu_sorter_create(sl->device) : 0; /* NULL: the consumer's handle sorts */
			pthread_create(&st[k], 0, sorter_main, &sa[k]);
		}
		for (;;) {
			job_t j;
			pthread_mutex_lock(&q.mtx);
			while (!(q.head < q.tail && q.ring[q.head % q.cap].state == 3) && !(q.reader_done && q.head == q.tail))
				pthread_cond_wait(&q.cv, &q.mtx);
			if (q.head == q.tail) { pthread_mutex_unlock(&q.mtx); break; }
			j = q.ring[q.head % q.cap];
			q.ring[q.head % q.cap].state = 0;
			++q.head;
			pthread_cond_broadcast(&q.cv);
			pthread_mutex_unlock(&q.mtx);
			if (j.err != 0) ret = -1;
			if (ret == 0) ret = consume(&cs, j.out, j.end_of_file);
			else if (j.out) {
				if (j.out->d_bwt) rb3gpu_sorter_release(j.out->gs, j.out->d_bwt);
				rb3h_batch_free(j.out->bwt); walkers_free(j.out); free(j.out->sent); free(j.out);
			}
		}
		pthread_join(rt, 0);
		for (k = 0; k < n_sort; ++k) pthread_join(st[k], 0);
		for (k = 0; k < n_sort; ++k) {
			double up = 0, so = 0;
			if (sa[k].gs && rb3gpu_sorter_stats(sa[k].gs, &up, &so, 0, 0) == 0) {
				pthread_mutex_lock(&g_sorted_mtx);
				g_sorted.ms_upload += up, g_sorted.ms_sort += so;
				pthread_mutex_unlock(&g_sorted_mtx);
			}
			/* the sorters' scratch (tens of bytes per symbol of a batch) is given back AFTER the index has been written: freeing
			 * gigabytes here made the next device allocation -- the run list of the FMD export -- wait for up to 1.4 s */
			if (sa[k].gs && n_old_sorters < 4) old_sorters[n_old_sorters++] = sa[k].gs;
			else rb3gpu_sorter_destroy(sa[k].gs);
		}
		free(st); free(sa); free(q.ring);
		if (rd.err != 0) ret = -1;
		n_empty = rd.n_empty, has_index = cs.has_index;
	} else if (argc - optind >= 1) {
		consumer_t cs = { h, opt, has_index, fn_tmp, sl->device };
		ret = for_each_batch(opt, argc - optind, argv + optind, sl->rbeg, sl->rend, submit_serial, &cs, &n_empty);
		has_index = cs.has_index;
	}
	sl->ret = ret, sl->has_index = has_index, sl->n_empty = n_empty, sl->n_old_sorters = n_old_sorters;
	return 0;
}

static const struct option long_opts[] = {
	{ "gpu", required_argument, 0, 301 },
	{ "split", required_argument, 0, 302 },
	{ "rebatch", no_argument, 0, 303 },
	{ "gpu-sort", no_argument, 0, 304 },
	{ "host-sort", no_argument, 0, 305 },
	{ "gpu-batch", required_argument, 0, 306 },
	{ "gpu-sort-limit", required_argument, 0, 307 }, /* (tests: pretend the GPU sorter takes less than it does) */
	{ "host-fmd", no_argument, 0, 308 },
	{ "gpus", required_argument, 0, 309 },
	{ "interval", no_argument, 0, 310 },
	{ 0, 0, 0, 0 }
};

int main_build(int argc, char *argv[])
{
	bopt_t opt;
	int c, ret = 0, has_index = 0;
	char *fn_in = 0, *fn_tmp = 0;
	rb3gpu_t *h;
	rb3gpu_opt_t gopt;
	int64_t n_empty = 0;
	rb3gpu_sorter_t *old_sorters[4];
	int n_old_sorters = 0;

	bopt_init(&opt);
	optind = 1;
	while ((c = getopt_long(argc, argv, "l:n:m:t:2sri:LFRo:dbTS:p:e", long_opts, 0)) >= 0) {
		if (c == 'm') opt.batch_size = rb3h_parse_num(optarg);
		else if (c == 't') opt.n_threads = atoi(optarg);
		else if (c == 'p') opt.sais_threads = atoi(optarg);
		else if (c == 'l') opt.block_len = atoi(optarg);
		else if (c == 'n') opt.max_nodes = atoi(optarg);
		else if (c == '2' || c == 's' || c == 'r') {
			fprintf(stderr, "ERROR: -%c selects the ropebwt2 insertion algorithm, which this build does not include; the default (suffix sorting + merge) gives the same BWT for -2\n", c);
			return 1;
		} else if (c == 'T' || c == 'e') {
			fprintf(stderr, "ERROR: output format -%c is not available in this build; use -d (FMD), -b (FMR) or the default plain text\n", c);
			return 1;
		} else if (c == 'i') fn_in = optarg;
		else if (c == 'L') opt.flag |= BF_LINE;
		else if (c == 'F') opt.flag |= BF_NO_FOR;
		else if (c == 'R') opt.flag |= BF_NO_REV;
		else if (c == 'o') { if (freopen(optarg, "wb", stdout) == 0) { fprintf(stderr, "ERROR: failed to write to '%s'\n", optarg); return 1; } }
		else if (c == 'd') opt.fmt = FMT_FMD;
		else if (c == 'b') opt.fmt = FMT_FMR;
		else if (c == 'S') fn_tmp = optarg;
		else if (c == 301) opt.device = atoi(optarg);
		else if (c == 302) opt.split_log2 = atoi(optarg);
		else if (c == 303) opt.rebatch = 1;
		else if (c == 304) opt.gpu_sort = 1;
		else if (c == 305) opt.gpu_sort = 0;
		else if (c == 306) opt.gpu_batch = rb3h_parse_num(optarg);
		else if (c == 307) opt.gpu_sort_limit = rb3h_parse_num(optarg);
		else if (c == 308) opt.host_fmd = g_host_fmd = 1;
		else if (c == 309) opt.n_gpus = atoi(optarg);
		else if (c == 310) opt.interval = 1;
		else if (c == '?') return 1;
	}
	if (opt.gpu_sort_limit > (int64_t)INT32_MAX - 16) opt.gpu_sort_limit = (int64_t)INT32_MAX - 16;
	if (opt.gpu_batch <= 0 || opt.gpu_batch > opt.gpu_sort_limit - 1) opt.gpu_batch = opt.gpu_sort_limit - 1;
	if (argc == optind && fn_in == 0) return usage_build(stderr, &opt);
	if ((opt.flag & BF_NO_FOR) && (opt.flag & BF_NO_REV)) {
		fprintf(stderr, "ERROR: -F and -R together leave nothing to index\n");
		return 1;
	}

	rb3gpu_opt_init(&gopt);
	gopt.device = opt.device, gopt.split_log2 = opt.split_log2, gopt.verbose = rb3h_verbose;
	h = rb3gpu_create(&gopt);
	if (h == 0) {
		fprintf(stderr, "ERROR: no usable MI355X/HIP device (device %d); the merge path has no CPU fallback\n", opt.device);
		return 1;
	}

	if (getenv("RB3_PINNED_LIMIT")) g_pin_limit = atoll(getenv("RB3_PINNED_LIMIT"));
	if (!getenv("RB3_NO_PINNED")) g_pin_on = 1, rb3h_seq_set_batch_allocator(pin_alloc, pin_release); /* batch buffers in page-locked memory (one DMA per batch) */

	if (fn_in) { /* build.c:172-184 */
		const int r = load_index(h, fn_in);
		if (r == -1) {
			if (rb3h_verbose >= 1) fprintf(stderr, "ERROR: failed to open index file '%s'\n", fn_in);
			rb3gpu_destroy(h);
			return 1;
		}
		if (r < 0) {
			fprintf(stderr, "ERROR: failed to load the index into HBM\n");
			rb3gpu_destroy(h);
			return 1;
		}
		has_index = 1;
		if (rb3h_verbose >= 3) {
			rb3gpu_stats_t st;
			rb3gpu_stats(h, &st);
			fprintf(stderr, "[M::%s::%.3f*%.2f] loaded the index from file '%s' (index %.1f MB in HBM, peak device memory while loading %.1f MB)\n", __func__, rb3h_realtime(), rb3h_percent_cpu(), fn_in, st.bytes_index / 1e6, st.bytes_peak / 1e6);
		}
	}

	if (opt.sais_threads < 0) opt.sais_threads = opt.gpu_sort ? 1 : 0; /* one batch sorted on the GPU while the one before is merged */
	if (opt.n_gpus < 1) opt.n_gpus = 1;
	/* Slices: contiguous pieces of the input in input order, equal in bytes, cut INSIDE files at record boundaries where every file
	 * can be read from an offset (regular files that are not gzip-compressed: a reads file of config 4 shards over all GPUs, each
	 * with its own reader and sorter, as kt_for spreads the chains of a batch, fm-index.c:217-224); with a pipe or a .gz among the
	 * inputs the slices are whole files. */
	int64_t *fsize = 0, total_bytes = 0;
	memset(&g_iv, 0, sizeof(g_iv));
	if (opt.interval && opt.n_gpus > 1) { /* ONE reader / sorter / consumer; the index is what is spread over the GPUs (north_star's split) */
		const int ndev = rb3gpu_device_count();
		if (!opt.gpu_sort || opt.n_gpus > RB3GPU_SH_MAXIV) { /* (-S works: the FMR writer takes the intervals where they are, see dump_fmr) */
			fprintf(stderr, "ERROR: --interval needs GPU suffix sorting and at most %d GPUs\n", RB3GPU_SH_MAXIV);
			rb3gpu_destroy(h);
			return 1;
		}
		g_iv.n = opt.n_gpus, g_iv.gopt = gopt;
		for (c = 0; c < g_iv.n; ++c) g_iv.devices[c] = (opt.device + c) % (ndev > 0 ? ndev : 1);
		opt.n_gpus = 1;
	}
	int by_bytes = opt.n_gpus > 1 && argc - optind >= 1;
	if (by_bytes) {
		fsize = (int64_t*)calloc((size_t)(argc - optind), sizeof(int64_t));
		for (c = 0; c < argc - optind && by_bytes; ++c) {
			if (!rb3h_seq_splittable(argv[optind + c], &fsize[c])) by_bytes = 0;
			else total_bytes += fsize[c];
		}
		if (total_bytes < (int64_t)opt.n_gpus * 64) by_bytes = 0; /* (nothing to cut) */
	}
	if (!by_bytes && opt.n_gpus > argc - optind) opt.n_gpus = argc - optind > 0 ? argc - optind : 1; /* slices are cut at file boundaries */
	if (opt.n_gpus > 1 && fn_tmp) { fprintf(stderr, "ERROR: -S (save after each file) is not available with --gpus\n"); rb3gpu_destroy(h); return 1; }
	if (opt.n_gpus == 1) {
		slice_t sl;
		memset(&sl, 0, sizeof(sl));
		sl.h = h, sl.opt = &opt, sl.device = opt.device, sl.rbeg = sl.rend = 0, sl.n_files = argc - optind, sl.files = argv + optind, sl.has_index = has_index, sl.fn_tmp = fn_tmp;
		run_slice(&sl);
		ret = sl.ret, has_index = sl.has_index, n_empty = sl.n_empty;
		for (c = 0; c < sl.n_old_sorters && n_old_sorters < 4; ++c) old_sorters[n_old_sorters++] = sl.old_sorters[c];
	} else { /* partitioned build: slice k of the files on device (--gpu + k) mod #devices, then a binary tree of whole-index merges */
		const int N = opt.n_gpus, ndev = rb3gpu_device_count(), nf = argc - optind;
		slice_t *sl = (slice_t*)calloc((size_t)N, sizeof(slice_t));
		pthread_t *th = (pthread_t*)calloc((size_t)N, sizeof(pthread_t));
		int k, stride;
		double t_tree;
		int64_t *rb = by_bytes ? (int64_t*)calloc((size_t)N * nf * 2, sizeof(int64_t)) : 0; /* per slice: begin and end offset in every file */
		for (k = 0; k < N && ret == 0; ++k) {
			int f0 = (int)((int64_t)nf * k / N), f1 = (int)((int64_t)nf * (k + 1) / N);
			if (by_bytes) { /* global byte range [lo, hi) of the concatenated inputs -> the files it touches and the range inside each */
				const int64_t lo = total_bytes / N * k, hi = k + 1 == N ? total_bytes : total_bytes / N * (k + 1);
				int64_t at = 0, *b = rb + (size_t)k * nf * 2, *e = b + nf;
				f0 = nf, f1 = 0;
				for (c = 0; c < nf; at += fsize[c], ++c) {
					const int64_t x0 = lo > at ? lo - at : 0, x1 = hi - at < fsize[c] ? hi - at : fsize[c];
					if (x1 <= x0 && !(fsize[c] == 0 && at >= lo && at < hi)) continue; /* (file c has nothing for this slice) */
					b[c] = x0, e[c] = x1 >= fsize[c] ? 0 : x1; /* (0: to the end of the file) */
					if (c < f0) f0 = c;
					f1 = c + 1;
				}
				if (f0 >= f1) f0 = f1 = 0;
				sl[k].rbeg = b + f0, sl[k].rend = e + f0;
			}
			sl[k].opt = &opt, sl[k].device = (opt.device + k) % (ndev > 0 ? ndev : 1), sl[k].n_files = f1 - f0, sl[k].files = argv + optind + f0;
			if (k == 0) sl[k].h = h, sl[k].has_index = has_index; /* (an index given with -i is the start of slice 0) */
			else {
				gopt.device = sl[k].device;
				sl[k].h = rb3gpu_create(&gopt);
				if (sl[k].h == 0) { fprintf(stderr, "ERROR: no usable HIP device %d for slice %d\n", sl[k].device, k); ret = -1; }
			}
		}
		if (ret == 0) {
			for (k = 0; k < N; ++k) pthread_create(&th[k], 0, run_slice, &sl[k]);
			for (k = 0; k < N; ++k) {
				pthread_join(th[k], 0);
				if (sl[k].ret != 0) ret = -1; /* (a slice whose files held nothing has no index: skipped below, as the reference skips such files, build.c:208-211) */
				n_empty += sl[k].n_empty;
				for (c = 0; c < sl[k].n_old_sorters; ++c) rb3gpu_sorter_destroy(sl[k].old_sorters[c]); /* (their scratch must not sit on the devices during the tree merge) */
			}
		}
		if (ret == 0 && rb3h_verbose >= 3)
			fprintf(stderr, "[M::%s::%.3f*%.2f] %d slices indexed on %d GPUs\n", __func__, rb3h_realtime(), rb3h_percent_cpu(), N, ndev < N ? ndev : N);
		t_tree = rb3h_realtime();
		{ /* the slices that hold an index, in input order (one whose files held nothing is skipped, as the single-GPU build and the reference skip such files) */
			int *live = (int*)calloc((size_t)N, sizeof(int)), nl = 0;
			for (k = 0; k < N; ++k) if (sl[k].has_index) live[nl++] = k;
			for (stride = 1; stride < nl && ret == 0; stride *= 2) /* merge(A, B) ranks the sentinels of B after those of A (fm-index.c:147): adjacent slices, left to right */
				for (k = 0; k + stride < nl && ret == 0; k += 2 * stride) { /* (the merges of one level are independent; they are short next to the slices and run in turn) */
					const int a = live[k], b = live[k + stride];
					const int r = rb3gpu_merge_index(sl[a].h, sl[b].h);
					if (r < 0) { fprintf(stderr, "ERROR: the GPU engine failed to merge the index of slice %d into slice %d: %s\n", b, a, rb3gpu_strerror(r)); ret = -1; }
					else {
						rb3gpu_stats_t so;
						if (rb3gpu_stats(sl[b].h, &so) == 0) g_other_slices.ms_path += so.ms_h2d + so.ms_lf + so.ms_rank + so.ms_build, g_other_slices.n_sym += so.n_symbols_merged;
						rb3gpu_destroy(sl[b].h), sl[b].h = 0;
						if (rb3h_verbose >= 3) fprintf(stderr, "[M::%s::%.3f*%.2f] merged the index of slice %d into slice %d\n", __func__, rb3h_realtime(), rb3h_percent_cpu(), b, a);
					}
				}
			if (ret == 0 && nl > 0 && live[0] != 0) { /* slice 0 held nothing: the result lives in another slice's handle */
				rb3gpu_destroy(h);
				h = sl[live[0]].h, sl[live[0]].h = 0, sl[0].h = h;
			}
			has_index = ret == 0 && nl > 0;
			free(live);
		}
		if (ret == 0 && rb3h_verbose >= 3) fprintf(stderr, "[M::%s] tree merge of %d slices: %.3f s\n", __func__, N, rb3h_realtime() - t_tree);
		for (k = 1; k < N; ++k) if (sl[k].h) rb3gpu_destroy(sl[k].h);
		free(sl); free(th); free(rb);
	}
	free(fsize);
	if (g_iv.batches > 0 && rb3h_verbose >= 3)
		fprintf(stderr, "[M::%s::%.3f*%.2f] %ld batches merged into %d intervals: %ld lock-step rounds, %.3f s\n", __func__, rb3h_realtime(), rb3h_percent_cpu(), (long)g_iv.batches, g_iv.n, (long)g_iv.rounds, g_iv.t_walk);
	if (n_empty > 0 && rb3h_verbose >= 2)
		fprintf(stderr, "WARNING: skipped %ld empty sequence(s)\n", (long)n_empty);

	if (ret != 0 || !has_index) {
		if (g_iv.s) rb3gpu_shard_destroy(g_iv.s), g_iv.s = 0;
		while (n_old_sorters > 0) rb3gpu_sorter_destroy(old_sorters[--n_old_sorters]);
		pin_drain();
		rb3gpu_destroy(h);
		return 1;
	}

	if (g_iv.s) { /* --interval: the writers take the intervals in rank order, where they are (the reference writes its ropes one after the other,
	                 fm-index.c:31-54, and rld_enc joins the runs that meet: rld0.c:153-161) -- the index is never put together on one device */
		int64_t acc[7];
		rb3gpu_shard_get_acc(g_iv.s, acc);
		if (opt.fmt == FMT_FMR) {
			ret = dump_fmr(h, &opt, stdout);
		} else if (opt.fmt == FMT_FMD) {
			rb3h_fmdw_t *w = rb3h_fmdw_init();
			ret = w ? rb3gpu_shard_export_run_words(g_iv.s, sink_fmd_words, w) : -1;
			if (ret == 0) ret = rb3h_fmdw_finish(w);
			if (ret == 0) ret = rb3h_fmdw_dump(w, stdout);
			if (w) rb3h_fmdw_destroy(w);
		} else {
			ret = rb3gpu_shard_export_runs(g_iv.s, sink_plain, stdout);
			fputc('\n', stdout);
		}
		if (rb3h_verbose >= 3) { /* what each device held */
			int64_t bnd[RB3GPU_SH_MAXIV + 1];
			const int n = rb3gpu_shard_bounds(g_iv.s, bnd);
			for (c = 0; c < n; ++c) {
				rb3gpu_stats_t st;
				if (rb3gpu_stats(rb3gpu_shard_handle(g_iv.s, c), &st) == 0)
					fprintf(stderr, "[M::%s] interval %d on device %d: %ld symbols, index %.1f MB, peak device memory %.1f MB; merge path: rank %.3f + rebuild %.3f ms\n", __func__, c, g_iv.devices[c], (long)(bnd[c + 1] - bnd[c]), st.bytes_index / 1e6, st.bytes_peak / 1e6, st.ms_rank, st.ms_build);
			}
		}
		/* the writers are done: the rank threads, the handles of intervals 1 .. N-1 and the replicated buffers go BEFORE the caller's handle (interval 0),
		   which the object points at */
		rb3gpu_shard_destroy(g_iv.s), g_iv.s = 0;
	} else if (opt.fmt == FMT_FMR) { /* build.c:245-260 */
		ret = dump_fmr(h, &opt, stdout);
	} else if (opt.fmt == FMT_FMD) {
		ret = write_fmd(h, stdout);
	} else {
		ret = rb3gpu_export_runs(h, sink_plain, stdout);
		fputc('\n', stdout);
	}
	fflush(stdout);
	if (rb3h_verbose >= 3) {
		rb3gpu_stats_t st;
		rb3gpu_stats(h, &st);
		fprintf(stderr, "[M::%s] GPU merge path: %ld symbols merged in %.3f ms (H2D %.3f + LF %.3f + rank %.3f + rebuild %.3f); index %.1f MB in HBM\n", __func__,
				(long)st.n_symbols_merged, st.ms_h2d + st.ms_lf + st.ms_rank + st.ms_build, st.ms_h2d, st.ms_lf, st.ms_rank, st.ms_build, st.bytes_index / 1e6);
		if (g_other_slices.n_sym > 0)
			fprintf(stderr, "[M::%s] the other slices of this multi-GPU build (their own GPUs, side by side): %ld symbols merged in %.3f ms of merge path, summed over the slices\n", __func__, (long)g_other_slices.n_sym, g_other_slices.ms_path);
		if (st.ms_sort > 0)
			fprintf(stderr, "[M::%s] GPU suffix sorting: %.3f ms in all (%ld doubling rounds), text upload included\n", __func__, st.ms_sort, (long)st.n_sort_rounds);
		if (g_sorted.ms_sort > 0)
			fprintf(stderr, "[M::%s] GPU sorter threads: text upload %.3f ms, suffix sorting %.3f ms (overlapped with the merges)\n", __func__, g_sorted.ms_upload, g_sorted.ms_sort);
		fprintf(stderr, "[M::%s] rebuild: %.3f ms for %ld algorithmic bytes (9 B x rows + old + new block array per round) = %.1f GB/s; LF walkers: k_chain %.3f ms in %ld launches, %ld steps\n", __func__,
				st.ms_build, (long)st.bytes_rebuild, st.ms_build > 0 ? st.bytes_rebuild / st.ms_build / 1e6 : 0.0, st.ms_chain, (long)st.n_rank_launches, (long)st.n_lf_steps);
		fprintf(stderr, "[M::%s] run-space rebuild: %ld groups, %ld of them handed on to the window kernels; %ld merges redone without tentative records, %ld needed the long settle pass; %ld rows LF-checked; %.1f ms in %ld device allocations\n", __func__,
				(long)st.n_reb_groups, (long)st.n_reb_groups_window, (long)st.n_fallbacks, (long)st.n_long_settles, (long)st.n_lf_checked, st.ms_alloc, (long)st.n_allocs);
		if (st.tent_mask_bits > 256)
			fprintf(stderr, "[M::%s] tentative records: drop-out masks of %ld bits (walkers met intervals of more than %ld matching suffixes)\n", __func__, (long)st.tent_mask_bits, (long)st.tent_mask_bits / 2 - 1);
		fprintf(stderr, "[M::%s] device memory of the index handle: peak %.1f MB, index %.1f MB\n", __func__, st.bytes_peak / 1e6, st.bytes_index / 1e6);
		fprintf(stderr, "[M::%s] batches: %ld (%ld symbols) suffix-sorted on the GPU, %ld (%ld symbols) on the host; -m %ld%s\n", __func__,
				(long)g_sorted.n_gpu, (long)g_sorted.sym_gpu, (long)g_sorted.n_host, (long)g_sorted.sym_host, (long)opt.batch_size,
				batch_cut(&opt) != opt.batch_size ? " cut into GPU sub-batches (--gpu-batch)" : "");
	}
	while (n_old_sorters > 0) rb3gpu_sorter_destroy(old_sorters[--n_old_sorters]);
	pin_drain();
	rb3gpu_destroy(h);
	if (ret != 0) { fprintf(stderr, "ERROR: failed to write the index (code %d)\n", ret); return 1; }
	return 0;
}

/* merge, main.c:84-133 (rb3_fmi_merge, fm-index.c:251-277): merge whole indexes into the first one.
 * The plain BWT of an index is itself a valid partial BWT whose sentinels all sort after those of
 * the base (fm-index.c:147), so every further index is decoded to plain symbols on the host and goes
 * through the same rb3gpu_merge_plain path as a batch. */
typedef struct { uint8_t *s; int64_t l, m; } plainvec_t;

static int sink_plainvec(void *data, int c, int64_t l)
{
	plainvec_t *v = (plainvec_t*)data;
	if (v->l + l > v->m) {
		v->m = (v->l + l) + ((v->l + l) >> 1) + 1024;
		v->s = (uint8_t*)realloc(v->s, (size_t)v->m);
		if (v->s == 0) return -1;
	}
	memset(v->s + v->l, c, (size_t)l);
	v->l += l;
	return 0;
}

int main_merge(int argc, char *argv[])
{
	int c, i, ret = 0, fmt = FMT_FMR, device = 0;
	char *fn_tmp = 0;
	rb3gpu_t *h;
	rb3gpu_opt_t gopt;
	bopt_t opt;
	bopt_init(&opt);
	optind = 1;
	while ((c = getopt_long(argc, argv, "t:o:S:db", long_opts, 0)) >= 0) {
		if (c == 't') opt.n_threads = atoi(optarg);
		else if (c == 'o') { if (freopen(optarg, "wb", stdout) == 0) return 1; }
		else if (c == 'S') fn_tmp = optarg;
		else if (c == 'd') fmt = FMT_FMD;
		else if (c == 'b') fmt = FMT_FMR;
		else if (c == 301) device = atoi(optarg);
		else if (c == 308) g_host_fmd = 1;
		else if (c == '?') return 1;
	}
	if (argc - optind < 2) {
		fprintf(stdout, "Usage: ropebwt3-amd merge [options] <base.fmr|fmd> <other1.fmr|fmd> [...]\n");
		fprintf(stdout, "Options:\n");
		fprintf(stdout, "  -o FILE    output to FILE [stdout]\n");
		fprintf(stdout, "  -b / -d    output FMR (default, as the reference) / FMD\n");
		fprintf(stdout, "  -S FILE    save the current index to FILE after each input file []\n");
		fprintf(stdout, "  --gpu INT  HIP device ordinal [0]\n");
		return 1;
	}
	rb3gpu_opt_init(&gopt);
	gopt.device = device, gopt.verbose = rb3h_verbose;
	h = rb3gpu_create(&gopt);
	if (h == 0) { fprintf(stderr, "ERROR: no usable MI355X/HIP device; the merge path has no CPU fallback\n"); return 1; }
	if (load_index(h, argv[optind]) < 0) {
		fprintf(stderr, "ERROR: failed to load FMR/FMD file '%s'\n", argv[optind]);
		rb3gpu_destroy(h);
		return 1;
	}
	for (i = optind + 1; i < argc && ret == 0; ++i) {
		plainvec_t pv = {0, 0, 0};
		{ /* an FMD file: decoded on the device and merged as one batch */
			uint64_t *z = 0;
			int64_t nw = 0, mc[6];
			if (!g_host_fmd && rb3h_fmd_read_words(argv[i], &z, &nw, mc) == 0) {
				ret = rb3gpu_merge_fmd_words(h, nw, z, mc);
				free(z);
				if (ret == 0) {
					if (rb3h_verbose >= 3) fprintf(stderr, "[M::%s::%.3f*%.2f] merged '%s'\n", __func__, rb3h_realtime(), rb3h_percent_cpu(), argv[i]);
					if (fn_tmp) {
						FILE *fp = fopen(fn_tmp, "wb");
						if (fp) { dump_fmr(h, &opt, fp); fclose(fp); }
					}
					continue;
				}
				if (ret != RB3GPU_ESYMBOL && ret != RB3GPU_ENOMEM) { fprintf(stderr, "ERROR: the GPU engine failed to merge '%s': %s\n", argv[i], rb3gpu_strerror(ret)); break; }
				ret = 0; /* the device declined the stream: host decoder */
			}
		}
		if (rb3h_index_read_runs(argv[i], sink_plainvec, &pv) < 0 || pv.l == 0) {
			fprintf(stderr, "ERROR: failed to load FMR/FMD file '%s'\n", argv[i]);
			free(pv.s); ret = 1;
			break;
		}
		ret = rb3gpu_merge_plain(h, pv.l, pv.s);
		free(pv.s);
		if (ret < 0) { fprintf(stderr, "ERROR: the GPU engine failed to merge '%s': %s\n", argv[i], rb3gpu_strerror(ret)); break; }
		if (rb3h_verbose >= 3) fprintf(stderr, "[M::%s::%.3f*%.2f] merged '%s'\n", __func__, rb3h_realtime(), rb3h_percent_cpu(), argv[i]);
		if (fn_tmp) {
			FILE *fp = fopen(fn_tmp, "wb");
			if (fp) { dump_fmr(h, &opt, fp); fclose(fp); }
		}
	}
	if (ret == 0) {
		if (fmt == FMT_FMR) ret = dump_fmr(h, &opt, stdout);
		else {
			ret = write_fmd(h, stdout);
		}
		fflush(stdout);
	}
	rb3gpu_destroy(h);
	return ret == 0 ? 0 : 1;
}

/* `ssa`, ssa.c:246-279: sampled suffix array of an index, written as rb3_ssa_dump does (ssa.c:198-213) */
int main_ssa(int argc, char *argv[])
{
	int c, ret, ssa_shift = 8, device = 0, ms = 0;
	int64_t m = 0, n_ssa = 0;
	uint64_t *r2i = 0, *ssa = 0;
	char *fn = 0;
	FILE *fp;
	rb3gpu_t *h;
	rb3gpu_opt_t gopt;
	optind = 1;
	while ((c = getopt_long(argc, argv, "t:s:o:", long_opts, 0)) >= 0) {
		if (c == 't') {} /* threads of the reference's kt_for: nothing to size here */
		else if (c == 's') ssa_shift = atoi(optarg);
		else if (c == 'o') fn = optarg;
		else if (c == 301) device = atoi(optarg);
		else if (c == 308) g_host_fmd = 1;
		else if (c == '?') return 1;
	}
	if (argc == optind) {
		fprintf(stderr, "Usage: ropebwt3-amd ssa [options] <in.fmd|fmr>\n");
		fprintf(stderr, "Options:\n");
		fprintf(stderr, "  -s INT     sample rate one SA per 2**INT bases [%d]\n", ssa_shift);
		fprintf(stderr, "  -o FILE    output to file [stdout]\n");
		fprintf(stderr, "  --gpu INT  HIP device ordinal [0]\n");
		return 1;
	}
	rb3gpu_opt_init(&gopt);
	gopt.device = device, gopt.verbose = rb3h_verbose;
	h = rb3gpu_create(&gopt);
	if (h == 0) { fprintf(stderr, "ERROR: no usable MI355X/HIP device; there is no CPU fallback\n"); return 1; }
	if (load_index(h, argv[optind]) < 0) {
		fprintf(stderr, "[E::%s] failed to load the FM-index\n", __func__);
		rb3gpu_destroy(h);
		return 1;
	}
	if (rb3h_verbose >= 3) fprintf(stderr, "[M::%s::%.3f*%.2f] loaded the index\n", __func__, rb3h_realtime(), rb3h_percent_cpu());
	ret = rb3gpu_ssa_dims(h, ssa_shift, &m, &n_ssa, &ms);
	if (ret == 0) {
		r2i = (uint64_t*)calloc((size_t)(m > 0 ? m : 1), 8);
		ssa = (uint64_t*)calloc((size_t)(n_ssa > 0 ? n_ssa : 1), 8);
		ret = r2i && ssa ? rb3gpu_ssa_gen(h, ssa_shift, r2i, ssa) : -1;
	}
	if (ret < 0) fprintf(stderr, "ERROR: the GPU engine failed to generate the sampled suffix array: %s\n", rb3gpu_strerror(ret));
	else {
		if (rb3h_verbose >= 3) {
			rb3gpu_stats_t st;
			rb3gpu_stats(h, &st);
			fprintf(stderr, "[M::%s::%.3f*%.2f] %lld samples of %lld strings in %.3f ms on the GPU (LF walk %.3f ms)\n", __func__, rb3h_realtime(), rb3h_percent_cpu(),
					(long long)n_ssa, (long long)m, st.ms_ssa, st.ms_ssa_walk);
		}
		fp = fn && strcmp(fn, "-") ? fopen(fn, "wb") : stdout;
		if (fp == 0) ret = -1;
		else {
			uint32_t y;
			fwrite("SSA\1", 1, 4, fp);
			y = (uint32_t)ssa_shift; fwrite(&y, 4, 1, fp);
			y = (uint32_t)ms; fwrite(&y, 4, 1, fp);
			fwrite(&m, 8, 1, fp);
			fwrite(&n_ssa, 8, 1, fp);
			fwrite(r2i, 8, (size_t)m, fp);
			fwrite(ssa, 8, (size_t)n_ssa, fp);
			if (fp != stdout) fclose(fp); else fflush(fp);
		}
	}
	free(r2i); free(ssa);
	rb3gpu_destroy(h);
	return ret == 0 ? 0 : 1;
}

/* recode: decode an FMD/FMR file on the host and write it back as plain text (default), FMD (-d)
 * or FMR (-b).  Host-only utility; also the CPU-side test bench of the two codecs. */
typedef struct { int64_t cnt[6]; runvec_t rv; } recode_t;

static int sink_recode(void *data, int c, int64_t l)
{
	recode_t *r = (recode_t*)data;
	r->cnt[c] += l;
	return sink_runvec(&r->rv, c, l);
}

static int main_recode(int argc, char *argv[])
{
	int c, fmt = FMT_PLAIN, ret = 0, block_len = 0, max_nodes = 0;
	int64_t i;
	recode_t rc;
	optind = 1;
	while ((c = getopt(argc, argv, "dbo:l:n:")) >= 0) {
		if (c == 'd') fmt = FMT_FMD;
		else if (c == 'b') fmt = FMT_FMR;
		else if (c == 'l') block_len = atoi(optarg);
		else if (c == 'n') max_nodes = atoi(optarg);
		else if (c == 'o' && freopen(optarg, "wb", stdout) == 0) return 1;
	}
	if (argc - optind < 1) { fprintf(stderr, "Usage: ropebwt3-amd recode [-d|-b] [-o out] <in.fmd|in.fmr>\n"); return 1; }
	memset(&rc, 0, sizeof(rc));
	if (rb3h_index_read_runs(argv[optind], sink_recode, &rc) < 0) { fprintf(stderr, "ERROR: failed to read '%s'\n", argv[optind]); free(rc.rv.a); return 1; }
	if (fmt == FMT_FMD) {
		rb3h_fmdw_t *w = rb3h_fmdw_init();
		for (i = 0; i < rc.rv.n && ret == 0; ++i) ret = rb3h_fmdw_enc(w, (int64_t)(rc.rv.a[i] >> 3), (int)(rc.rv.a[i] & 7));
		if (ret == 0) ret = rb3h_fmdw_finish(w);
		if (ret == 0) ret = rb3h_fmdw_dump(w, stdout);
		rb3h_fmdw_destroy(w);
	} else if (fmt == FMT_FMR) {
		int64_t acc[7];
		rb3h_fmrw_t *w;
		for (acc[0] = 0, c = 0; c < 6; ++c) acc[c + 1] = acc[c] + rc.cnt[c];
		w = rb3h_fmrw_init(acc, max_nodes, block_len);
		for (i = 0; i < rc.rv.n && ret == 0; ++i) ret = rb3h_fmrw_enc(w, (int64_t)(rc.rv.a[i] >> 3), (int)(rc.rv.a[i] & 7));
		if (ret == 0) ret = rb3h_fmrw_dump(w, stdout);
		rb3h_fmrw_destroy(w);
	} else {
		for (i = 0; i < rc.rv.n && ret == 0; ++i) ret = sink_plain(stdout, (int)(rc.rv.a[i] & 7), (int64_t)(rc.rv.a[i] >> 3));
		fputc('\n', stdout);
	}
	free(rc.rv.a);
	return ret == 0 ? 0 : 1;
}

/* plain2fmd (the reference's main.c:299-331): every byte of the input is one BWT symbol, '\n' and '$' are sentinels; host only.
 * The symbols go to the FMD writer a run at a time, not one by one. */
static int main_plain2fmd(int argc, char *argv[])
{
	int c, j, ret = 0, cur = -1;
	int64_t run = 0;
	rb3h_fmdw_t *w;
	uint8_t *buf, tab[256];
	optind = 1;
	while ((c = getopt(argc, argv, "o:")) >= 0)
		if (c == 'o' && freopen(optarg, "wb", stdout) == 0) { fprintf(stderr, "ERROR: failed to open '%s' for writing\n", optarg); return 1; }
	if (argc - optind < 1) {
		fprintf(stdout, "Usage: ropebwt3-amd plain2fmd [-o output.fmd] <in.txt>\n");
		return 0;
	}
	buf = (uint8_t*)malloc(1 << 20);
	for (c = 0; c < 256; ++c) tab[c] = (uint8_t)c;
	rb3h_char2nt6(256, tab);
	tab['\n'] = tab['$'] = 0;
	w = rb3h_fmdw_init();
	for (j = optind; j < argc && ret == 0; ++j) {
		FILE *fp = strcmp(argv[j], "-") == 0 ? stdin : fopen(argv[j], "r");
		size_t i, len;
		if (fp == 0) { fprintf(stderr, "ERROR: failed to open '%s'\n", argv[j]); ret = -1; break; }
		while (ret == 0 && (len = fread(buf, 1, 1 << 20, fp)) > 0) {
			for (i = 0; i < len && ret == 0; ++i) {
				const int x = tab[buf[i]];
				if (x == cur) { ++run; continue; }
				if (run > 0) ret = rb3h_fmdw_enc(w, run, cur);
				cur = x, run = 1;
			}
		}
		if (fp != stdin) fclose(fp);
	}
	if (ret == 0 && run > 0) ret = rb3h_fmdw_enc(w, run, cur);
	if (ret == 0) ret = rb3h_fmdw_finish(w);
	if (ret == 0) ret = rb3h_fmdw_dump(w, stdout);
	rb3h_fmdw_destroy(w);
	free(buf);
	return ret == 0 ? 0 : 1;
}

static int usage(FILE *fp)
{
	fprintf(fp, "Usage: ropebwt3-amd <command> <arguments>\n");
	fprintf(fp, "Commands:\n");
	fprintf(fp, "    build      construct a BWT (merge path on an MI355X)\n");
	fprintf(fp, "    merge      merge BWTs (on an MI355X)\n");
	fprintf(fp, "    ssa        generate sampled suffix array (on an MI355X)\n");
	fprintf(fp, "    recode     convert an FMD/FMR file to plain text, FMD (-d) or FMR (-b) (host only)\n");
	fprintf(fp, "    plain2fmd  convert BWT in plain text to FMD (host only)\n");
	fprintf(fp, "    version    print the version number\n");
	return fp == stdout ? 0 : 1;
}

int main(int argc, char *argv[])
{
	int ret = 0;
	rb3h_init();
	if (getenv("RB3_VERBOSE")) rb3h_verbose = atoi(getenv("RB3_VERBOSE")); /* (diagnostics: 4 makes the engine print every merge's phases from its HIP events; the reference's level is 3) */
	if (argc == 1) return usage(stdout);
	else if (strcmp(argv[1], "build") == 0) ret = main_build(argc - 1, argv + 1);
	else if (strcmp(argv[1], "merge") == 0) ret = main_merge(argc - 1, argv + 1);
	else if (strcmp(argv[1], "ssa") == 0) ret = main_ssa(argc - 1, argv + 1);
	else if (strcmp(argv[1], "recode") == 0) ret = main_recode(argc - 1, argv + 1);
	else if (strcmp(argv[1], "plain2fmd") == 0) ret = main_plain2fmd(argc - 1, argv + 1);
	else if (strcmp(argv[1], "version") == 0) { printf("%s\n", RB3H_VERSION); return 0; }
	else { fprintf(stderr, "ERROR: unknown command '%s'\n", argv[1]); return 1; }
	if (rb3h_verbose >= 3 && argc > 2 && ret == 0) { /* main.c:73-80 */
		int i;
		fprintf(stderr, "[M::%s] Version: %s\n", __func__, RB3H_VERSION);
		fprintf(stderr, "[M::%s] CMD:", __func__);
		for (i = 0; i < argc; ++i) fprintf(stderr, " %s", argv[i]);
		fprintf(stderr, "\n[M::%s] Real time: %.3f sec; CPU: %.3f sec; Peak RSS: %.3f GB\n", __func__, rb3h_realtime(), rb3h_cputime(), rb3h_peakrss() / 1024.0 / 1024.0 / 1024.0);
	}
	return ret;
}
