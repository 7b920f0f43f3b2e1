/*
 * rb3gpu.h -- C ABI of the MI355X-native BWT-merge engine (librb3gpu.so).
 *
 * This is the drop-in boundary for the hot path of `ropebwt3 build`: every entry
 * point replaces one call the reference's build.c makes on an `mrope_t*` -- and,
 * further down, the calls on either side of it: the suffix sorting of a batch, the
 * run export / FMD packing, the sampled suffix array (citations are file:line in
 * the reference tree).  The opaque `rb3gpu_t`
 * stands where `mrope_t*` stood; it owns a flat run/bit-plane block array in
 * HBM instead of the reference's B+-tree of RLE leaves.  Plain C types only:
 * no HIP, torch or C++ types cross this boundary.  All functions return 0 on
 * success or a negative RB3GPU_E* code (the reference asserts/aborts instead,
 * fm-index.c:125,246); nothing here ever falls back to a CPU implementation.
 */
#ifndef RB3GPU_H
#define RB3GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RB3GPU_ASIZE 6           /* $ACGTN = 0..5, fm-index.h:15 (RB3_ASIZE) */

#define RB3GPU_OK         0
#define RB3GPU_ENODEV    -1      /* no usable HIP device / HIP runtime error */
#define RB3GPU_ENOMEM    -2      /* device or host allocation failed */
#define RB3GPU_EINVAL    -3      /* bad argument (NULL, negative length, ...) */
#define RB3GPU_ESYMBOL   -4      /* a BWT byte is outside 0..5 (fm-index.c:124-125) */
#define RB3GPU_ESTATE    -5      /* call not valid in this state (e.g. merge into an empty index) */
#define RB3GPU_EINTERNAL -6      /* invariant violated on the device (fm-index.c:246 analogue) */
#define RB3GPU_EUNSUP    -7      /* this call cannot serve this index; use the alternative named in its description */

typedef struct rb3gpu_s rb3gpu_t;

typedef struct {
	int32_t device;              /* HIP device ordinal */
	int32_t split_log2;          /* long-chain splitting: start an extra LF walker at every row whose
	                                hashed id is 0 mod 2^split_log2; 0 = automatic, <0 = never split */
	int32_t verbose;             /* >=3: per-phase "[M::...]" lines on stderr like the reference */
	int32_t reserved;
} rb3gpu_opt_t;

/* per-handle counters, all cumulative since rb3gpu_create()/rb3gpu_stats_reset() */
typedef struct {
	double  ms_h2d;              /* host->device copies of partial BWTs */
	double  ms_lf;               /* LF-array construction of B2 (fm-index.c:206-216) */
	double  ms_rank;             /* LF-chain / rank kernels (fm-index.c:160-175, 217-224) */
	double  ms_build;            /* interleave + block-array rebuild (fm-index.c:237-249, 294-299) */
	double  ms_export;           /* run / symbol export kernels + D2H */
	double  ms_chain;            /* k_chain alone (HIP events on the launch stream): the dominant kernel */
	int64_t n_rank_launches;     /* launches of the chain kernel */
	int64_t n_lf_steps;          /* LF steps executed by the chain kernel (incl. speculative ones) */
	int64_t n_symbols_merged;    /* sum of `len` over merge calls */
	int64_t n_rounds;            /* chain-kernel launches incl. fallbacks */
	int64_t n_fallbacks;         /* merges whose optimistic (tentative-record) rank phase had to be redone */
	int64_t bytes_index;         /* current size of the block array + group directory in HBM */
	int64_t bytes_peak;          /* high-water mark of device memory owned by the handle */
	double  ms_ssa;              /* rb3gpu_ssa_gen: kernels + copy-back */
	double  ms_ssa_walk;         /* k_ssa_walk alone (one LF step per row of the index) */
	double  ms_sort;             /* rb3gpu_bwt_from_text: upload + suffix sorting + BWT */
	int64_t n_sort_rounds;       /* prefix-doubling rounds of those calls */
	int64_t n_reb_groups;        /* groups (8192 symbols) of the merges whose rebuild went through the run-space kernel ... */
	int64_t n_reb_groups_window; /* ... and how many of them it handed on to the per-window kernels (single-sync merges) */
	int64_t n_lf_checked;        /* batch rows whose LF relation was verified against the index after the rank phase (approximate) */
	int64_t n_long_settles;      /* merges whose tentative records needed the pointer-jumping settle pass (paths over > 64 walkers) */
	double  ms_alloc;            /* host time inside hipMalloc / hipFree for the handle's buffers (they grow with the index; replaced ones are freed in bulk) */
	int64_t n_allocs;
	int64_t n_reb_again;         /* rebuilds done twice because a buffer sized by an estimate did not take the result */
	int64_t bytes_rebuild;       /* algorithmic bytes of the rebuilds: per merge 9 B x rows + old block array + new block array */
	int64_t n_thinned;           /* merges done again with fewer, longer walkers because the table of tentative stretches was full */
	int64_t tent_mask_bits;      /* width of the drop-out masks the last single-sync merge settled its tentative records with: 256, or 512 / 1024 /
	                                2048 once walkers have met intervals of more matching suffixes than that (an index of > 255 relatives) */
	int64_t n_junctions_checked; /* junctions of the speculative walk (a walker meeting somebody's record; a drop-out event) whose LF relation was
	                                verified after the rank phase: all of them, on every merge with text-order words */
	int64_t n_peer_rounds;       /* lock-step rounds of the interval-sharded merge that ran as PEER ROUNDS (one kernel per rank, states written straight
	                                into the owner's receive buffer; rb3gpu_comm_t.stream_barrier): 0 where the host drove the rounds */
} rb3gpu_stats_t;

void rb3gpu_opt_init(rb3gpu_opt_t *opt);
const char *rb3gpu_strerror(int err);

/* mr_init (mrope.c:15-26) / mr_destroy (mrope.c:28-34) */
rb3gpu_t *rb3gpu_create(const rb3gpu_opt_t *opt);
void rb3gpu_destroy(rb3gpu_t *h);

/* rb3_enc_plain2fmr(len, bwt, max_nodes, block_len, n_threads), fm-index.c:114-137,
 * called at build.c:77,223 -- index the first batch.  `bwt` is caller-owned host memory,
 * read-only during the call.  Any previous content of the handle is dropped. */
int rb3gpu_from_plain(rb3gpu_t *h, int64_t len, const uint8_t *bwt);

/* rb3_fmi_merge_plain(r, len, seq, n_threads), fm-index.c:279-303, called at
 * build.c:78,226 -- merge the partial BWT of a later batch in place. */
int rb3gpu_merge_plain(rb3gpu_t *h, int64_t len, const uint8_t *bwt);

/* Same two operations with the partial BWT already resident in HBM on the handle's
 * device (`d_bwt` is a device pointer).  Used by the pipelined build and by bench.py,
 * whose timed region starts with inputs in HBM.  commit=0 computes the merged block
 * array and then discards it, leaving the index unchanged (repeatable benchmark step). */
int rb3gpu_from_plain_dev(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt);
int rb3gpu_merge_plain_dev(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt, int commit);

/* An LF WALKER follows one string of the batch right to left (the loop of rb3_mg_rank1_plain,
 * fm-index.c:160-175).  The reference starts one per string; this engine can also start walkers in
 * the middle of long strings (see rb3gpu_kernels.h).  Callers that still hold the suffix array of
 * the batch (sais-ss.c:23-26 has it when the BWT is written) can say where: */
typedef struct {
	int64_t row;                 /* row of B2 (rank of the suffix in the batch) where the walker starts */
	int64_t ka0;                 /* exact insertion point of that suffix in B1 if known (sentinel rows:
	                                acc[1] of the index, fm-index.c:164), else -1 */
	int64_t nsteps;              /* LF steps to the end of the walker's segment: the text distance to the walker on its left, plus the steps the
	                                walker starts outside its segment (flags >> 8) */
	int64_t flags;               /* RB3GPU_WK_* in the low byte; flags >> 8 & 255: a walker given by text position may start that many positions
	                                to the RIGHT of its segment (rb3h_walkers_text does: 32, the age from which a walker records), on rows its
	                                right neighbour owns and records; flags >> 16 & 255: if not 0, the walker does not start at all when the
	                                row that many positions further right already carries a record (it comes too late: rb3h_walkers_text, RB3H_PROBE = 64) */
} rb3gpu_walker_t;
#define RB3GPU_KA_SENTINEL (-2)  /* ka0 of a sentinel row: the engine substitutes acc[1] of the index */
#define RB3GPU_WK_CHECK 2        /* the rows ahead may already be recorded: check each before recording */

/* rb3gpu_merge_plain with an explicit walker list (host memory); same result, more parallelism
 * and no dependence on how the suffix array scatters the strings.  An entry that is not a walker of this batch (row outside it, no
 * steps) makes the call return RB3GPU_EINVAL with nothing installed; the list is checked on the device, by the kernel that walks it. */
int rb3gpu_merge_plain_walkers(rb3gpu_t *h, int64_t len, const uint8_t *bwt, int64_t n_walkers, const rb3gpu_walker_t *walkers);
/* (device BWT) walkers == NULL: one walker per string, made on the device; n_walkers = the number of strings of the batch */
int rb3gpu_merge_plain_dev_walkers(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt, int64_t n_walkers, const rb3gpu_walker_t *walkers, int commit);

/* The same merge in three stages, for a multi-GPU build with the index replicated and the batch's
 * walkers sharded by text range: every rank calls begin (LF array of the whole batch), walk on ITS
 * walkers (several calls allowed: hand-off values arriving from the neighbour rank start fix-up
 * walkers), then the ranks combine pos[] (RCCL all-reduce MAX over the device buffer returned by
 * rb3gpu_mg_pos_ptr: rows not yet recorded hold negative words, identical on every rank) and every rank calls finish.  d_pos_ext, if not NULL, is a
 * caller-owned device buffer of len int64 to use for pos[].  stop_row >= 0 names the row where the
 * territory of the next rank begins (the start row of its top walker): any walker reaching it stops,
 * and *arrive (host, optional) receives the exact value a walker arrived there with, or -1 if none did. */
int rb3gpu_mg_begin(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt, void *d_pos_ext, int64_t acc2[RB3GPU_ASIZE+1]);
int rb3gpu_mg_walk(rb3gpu_t *h, int64_t n_walkers, const rb3gpu_walker_t *walkers, int64_t stop_row, int64_t *arrive);
int rb3gpu_mg_pos_ptr(rb3gpu_t *h, void **d_pos, int64_t *len);
int rb3gpu_mg_finish(rb3gpu_t *h, int commit);

/* rb3_mg_rank_plain(fa, len, seq, rb, acc, n_threads), fm-index.c:202-225 -- the rank phase
 * alone, for parity tests against the reference's rb[] array: on return pos[kb] = ka[kb]+kb,
 * the merged position of row kb of B2 (= rb[kb]>>6 in the reference), and acc2[7] the C array
 * of B2.  The index is not modified.  `pos` is host memory of `len` int64. */
int rb3gpu_mg_rank_plain(rb3gpu_t *h, int64_t len, const uint8_t *bwt, int64_t *pos, int64_t acc2[RB3GPU_ASIZE+1]);

/* the same through the walker-list entry point (tests: pos[] of the single-synchronisation path) */
int rb3gpu_mg_rank_plain_walkers(rb3gpu_t *h, int64_t len, const uint8_t *bwt, int64_t n_walkers, const rb3gpu_walker_t *walkers, int64_t *pos, int64_t acc2[RB3GPU_ASIZE+1]);

/* rb3_fmi_rank1a(f, k, ok), fm-index.h:109-112 -> mr_rank2a, mrope.c:71-121: for each of the
 * n query offsets k[i] in [0, total], ok[6*i+c] = #{j < k[i] : B[j] = c}.  Host arrays. */
int rb3gpu_rank1a_batch(rb3gpu_t *h, int64_t n, const int64_t *k, int64_t *ok);

/* rb3_fmi_get_acc (fm-index.c:544-550) -> mr_get_ac (mrope.h:113-120): acc[a] = #symbols < a */
int rb3gpu_get_acc(const rb3gpu_t *h, int64_t acc[RB3GPU_ASIZE+1]);
int64_t rb3gpu_get_tot(const rb3gpu_t *h);          /* mr_get_tot, mrope.h:122-130 */

/* Ordered export of the index as runs, replacing the leaf-block iteration of
 * rb3_enc_fmr2fmd (fm-index.c:31-54) / mr_print_bwt (mrope.c:201-214): emit(c, l, data) is
 * called on the host for consecutive runs, in BWT order; adjacent calls may carry the same
 * symbol (the FMD writer coalesces, rld0.c:153-161).  A non-zero return from emit aborts. */
typedef int (*rb3gpu_emit_f)(void *data, int c, int64_t l);
int rb3gpu_export_runs(rb3gpu_t *h, rb3gpu_emit_f emit, void *data);

/* The same runs in bulk, for writers that want a tight loop instead of a call per run: emit receives arrays of
 * words start << 3 | sym, one per MAXIMAL run in BWT order (adjacent runs differ in symbol); run i ends where
 * run i+1 starts -- possibly in the next call -- and the last call is (n = 0, words = NULL, end = total length),
 * which closes the last run.  In the other calls end is -1. */
typedef int (*rb3gpu_emit_words_f)(void *data, int64_t n, const uint64_t *words, int64_t end);
int rb3gpu_export_run_words(rb3gpu_t *h, rb3gpu_emit_words_f emit, void *data);

/* The data section of the .fmd packed on the GPU (rb3_enc_fmr2fmd + rld_enc + rld_enc_finish, fm-index.c:31-52,
 * rld0.c:107-216): *words receives a malloc'ed array (free it with rb3gpu_host_free) of *n_words 64-bit words, the
 * blocks incl. the trailing header-only block, exactly what rld_dump writes between the file header and the rank
 * index (rld0.c:237-239; n_bytes = 8 * *n_words).  Blocks with 16-bit and with 32-bit headers (rld0.c:116-128: the
 * block before holds fewer than 0x4000 / 0x40000000 symbols) are packed on the device; an index with a block of 2^30
 * symbols or more (64-bit header) returns RB3GPU_EUNSUP and the caller packs the runs of rb3gpu_export_run_words on
 * the host. */
int rb3gpu_export_fmd_words(rb3gpu_t *h, uint64_t **words, int64_t *n_words);
void rb3gpu_host_free(void *p);

/* Page-locked host memory (hipHostMalloc) for batch buffers: a text or BWT that the caller builds in such a buffer is copied
 * to HBM by one DMA at PCIe speed (rb3gpu_from_plain, rb3gpu_merge_plain*, rb3gpu_sort_text, rb3gpu_sorter_*); any other host
 * pointer is staged through pinned buffers chunk by chunk, which a single host thread feeds at memcpy speed.  The reference's
 * counterpart is the malloc of the batch buffer (build.c:205, io.c:92).  NULL: no such memory to be had (use malloc). */
void *rb3gpu_pinned_alloc(int64_t n_bytes);
void rb3gpu_pinned_free(void *p);

/* The text distance between the LF walkers of a batch of n_strings long strings, `len` symbols in all (the `step` of
 * rb3h_walkers_text / rb3h_build_bwt_walkers; the reference has one chain per string, fm-index.c:217-224): as many walkers as the
 * walker kernel keeps resident on `device` -- one per group of eight lanes, compute units x 160 -- because a wave lasts as long as
 * its longest walker and walkers beyond the resident ones only start when others have finished; never closer than 192 positions
 * (a walker must be 16 steps old before it records -- RB3_TENT_MIN_AGE --, and segments shorter than 128 are not split).  < 0: no such device. */
int64_t rb3gpu_walker_step(int device, int64_t len, int64_t n_strings);

/* The whole BWT as one symbol per byte (0..5) into host memory of rb3gpu_get_tot() bytes;
 * small indexes / tests only. */
int rb3gpu_export_plain(rb3gpu_t *h, uint8_t *out);

/* same into device memory of the handle's GPU (rb3gpu_get_tot() bytes): the plain BWT of an index
 * is itself a valid partial BWT, so a whole index can be merged into another one with
 * rb3gpu_merge_plain_dev -- the tree-shaped multi-GPU build (rb3_fmi_merge, fm-index.c:251-277) */
int rb3gpu_export_plain_dev(rb3gpu_t *h, uint8_t *d_out);
int rb3gpu_export_plain_range_dev(rb3gpu_t *h, int64_t beg, int64_t end, uint8_t *d_out); /* symbols [beg, end) only */
/* bounds[0..n] of n intervals of positions that hold about equal BYTES of the block array (cut at group boundaries) */
int rb3gpu_balanced_bounds(rb3gpu_t *h, int n, int64_t *bounds);

/* Sampled suffix array of the index, `ropebwt3 ssa` (rb3_ssa_gen ssa.c:54-81; the words are those of
 * rb3_ssa_t fm-index.h:28-36 as written by rb3_ssa_dump ssa.c:198-213):
 *   r2i[k], k in [0, m): the string whose first suffix is reached from sentinel row k (ssa.c:36);
 *   ssa[x], x in [0, n_ssa): for row m + (x << ssa_shift), (offset of its suffix in its string) << ms | string.
 * rb3gpu_ssa_dims gives the sizes (m = acc[1], n_ssa = ceil((n - m) / 2^ssa_shift), ms = bits of m as in
 * ssa.c:62-64); rb3gpu_ssa_gen fills caller-owned host arrays of m and n_ssa words. */
int rb3gpu_ssa_dims(const rb3gpu_t *h, int ssa_shift, int64_t *m, int64_t *n_ssa, int *ms);
int rb3gpu_ssa_gen(rb3gpu_t *h, int ssa_shift, uint64_t *r2i, uint64_t *ssa);

/* The merge of a batch that comes with its TEXT-ORDER WORDS: d_tw[t] = row of the suffix at text position t << 3 |
 * the symbol before it (0 at the start of a string), i.e. the inverse suffix array of the batch next to its BWT
 * (rb3gpu_sort_text / rb3gpu_sorter_sort produce both; a host sorter has the suffix array and inverts it).
 * walkers[i].row is then the TEXT POSITION a walker starts at (a sentinel's walker: the position of the sentinel),
 * everything else as in rb3gpu_merge_plain_dev_walkers.  walkers == NULL: one walker per string, made on the device
 * (n_walkers = the number of strings of the batch; short strings, i.e. reads).  Same result as rb3_fmi_merge_plain (fm-index.c:279-303);
 * the LF walkers read their batch-side state as a stream instead of one random row word per step.
 * d_bwt: len bytes, d_tw: len 64-bit words, both device memory.  rb3gpu_mg_rank_text_dev: the rank phase alone
 * (pos[] as in rb3gpu_mg_rank_plain), for parity tests. */
int rb3gpu_merge_text_dev(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt, const uint64_t *d_tw, int64_t n_walkers, const rb3gpu_walker_t *walkers, int commit);
/* the same for a caller that also has the batch's suffix array (d_sa[i] = text position of the suffix of row i, len x u32 of device
 * memory: the GPU sorter's rb3gpu_sorter_sort[_uploaded]_sa / rb3gpu_sort_text_sa leave it behind the text-order words).  The
 * walkers may then leave their records in text order -- eight consecutive words per store instead of eight random rows -- and the
 * validation pass gathers them into row order through d_sa: what an index that lives in HBM wants (a batch of reads into a large
 * index: the record stores are two thirds of the walk).  d_sa == NULL: rb3gpu_merge_text_dev.  Same result either way. */
int rb3gpu_merge_text_sa_dev(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt, const uint64_t *d_tw, const uint32_t *d_sa, int64_t n_walkers, const rb3gpu_walker_t *walkers, int commit);
/* The same with the walker list made ON THE DEVICE: the caller passes the number of strings of the batch and a spacing (rb3gpu_walker_step), the
 * engine puts a walker at every sentinel and at every multiple of `step` strictly inside a string -- the list rb3h_walkers_text makes on the host
 * (70 ms of one core per 152-genome build), by two small kernels in front of the walk.  d_sa may be NULL.  A wrong string count: RB3GPU_EINVAL. */
int rb3gpu_merge_text_step_dev(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt, const uint64_t *d_tw, const uint32_t *d_sa, int64_t n_strings, int64_t step, int commit);
/* that list on the host, without its empty slots (tests): *walkers is malloc'ed, free it with rb3gpu_host_free */
int rb3gpu_walkers_step_dev(rb3gpu_t *h, int64_t len, const uint64_t *d_tw, int64_t n_strings, int64_t step, int64_t *n_walkers, rb3gpu_walker_t **walkers);
int rb3gpu_mg_rank_text_dev(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt, const uint64_t *d_tw, int64_t n_walkers, const rb3gpu_walker_t *walkers, int64_t *pos, int64_t acc2[RB3GPU_ASIZE+1]);

/* Partial BWT of one batch on the GPU, instead of rb3_build_sais on the host (sais-ss.c:10-56; libsais in GSA
 * mode: the i-th sentinel sorts before the (i+1)-th, sais-ss.c:16-21).  text: len symbols 0..5 in host memory,
 * every string terminated by 0 (so text[len-1] == 0), exactly what rb3_seq_read leaves in seq->s (io.c:104-125);
 * it is not modified.  d_bwt: len bytes of device memory (rb3gpu_dev_alloc) that receive the BWT, ready for
 * rb3gpu_from_plain_dev / rb3gpu_merge_plain_dev[_walkers].  If step > 0 and ckrow != NULL, ckrow[i] (host,
 * ceil(len / step) entries) receives the row of the suffix that starts at text position i * step -- the sampled
 * inverse suffix array a walker list is made from.  len < 2^31. */
int rb3gpu_bwt_from_text(rb3gpu_t *h, int64_t len, const uint8_t *text, uint8_t *d_bwt, int64_t step, int64_t *ckrow);
/* the same, with the text-order words of the batch (len 64-bit words of device memory) for rb3gpu_merge_text_dev */
int rb3gpu_sort_text(rb3gpu_t *h, int64_t len, const uint8_t *text, uint8_t *d_bwt, uint64_t *d_tw);
/* ... and the suffix array (len x u32 of device memory) for rb3gpu_merge_text_sa_dev */
int rb3gpu_sort_text_sa(rb3gpu_t *h, int64_t len, const uint8_t *text, uint8_t *d_bwt, uint64_t *d_tw, uint32_t *d_sa);

/* The same sorter as an object of its own (own HIP stream and scratch, independent of any index handle), so that a
 * host thread can sort batch i+1 while another merges batch i -- the pipeline of build.c:55-83, 186-201 with both
 * stages on the GPU.  rb3gpu_sorter_bwt blocks until the BWT is complete and returns it in one of the sorter's two
 * output buffers (*d_bwt, device memory, len bytes, valid until rb3gpu_sorter_release; it waits while both are out).
 * A sorter is used by one thread at a time; release may come from any thread. */
typedef struct rb3gpu_sorter_s rb3gpu_sorter_t;
rb3gpu_sorter_t *rb3gpu_sorter_create(int device);
void rb3gpu_sorter_destroy(rb3gpu_sorter_t *s);
int rb3gpu_sorter_bwt(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text, void **d_bwt, int64_t step, int64_t *ckrow);
/* BWT + text-order words (*d_tw: len 64-bit words in the same output buffer; released together with *d_bwt) */
int rb3gpu_sorter_sort(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text, void **d_bwt, void **d_tw);
int rb3gpu_sorter_release(rb3gpu_sorter_t *s, void *d_bwt);
/* rb3gpu_sorter_sort in its two stages, for callers that time or overlap them: the upload of the text (the H2D copy of the
 * merge path, SURVEY 8(d): one DMA if `text` lies in memory from rb3gpu_pinned_alloc, else through pinned staging buffers),
 * then the suffix sorting of the text uploaded last (len must be the same). */
int rb3gpu_sorter_upload(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text);
/* the same for a batch of at most 32 records on both strands (rb3_seq_read with is_for and is_rev: per record l symbols, 0, the l
 * symbols of the reverse complement, 0; io.c:84-102): only the forward strands are copied, the reverse complements (io.c:30-40) are
 * written on the device.  pair_start[i] = offset of record i in `text` (pair_start[0] = 0; record n_pairs - 1 ends at len).
 * RB3GPU_EINVAL if the text does not have that layout (use rb3gpu_sorter_upload). */
int rb3gpu_sorter_upload_fwd(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text, int64_t n_pairs, const int64_t *pair_start);
/* rb3gpu_sorter_upload_fwd in two halves, so that the copy of the NEXT batch runs beside the merge of the current one on the copy
 * engine (what the CLI gets from its sorter thread; build.c:203-239 reads the next batch while the current one is inserted):
 * _begin queues the copies and the strand kernel on the sorter's stream and returns at once if `text` is page-locked
 * (rb3gpu_pinned_alloc; otherwise it is rb3gpu_sorter_upload_fwd), and `text` must stay untouched until _end -- or the sort, which
 * the stream orders behind the copies anyway -- has returned.  _end waits for whatever the sorter's stream still holds. */
int rb3gpu_sorter_upload_fwd_begin(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text, int64_t n_pairs, const int64_t *pair_start);
int rb3gpu_sorter_upload_end(rb3gpu_sorter_t *s);
int rb3gpu_sorter_sort_uploaded(rb3gpu_sorter_t *s, int64_t len, void **d_bwt, void **d_tw);
/* the same with the suffix array (*d_sa: len x u32 behind the text-order words in the same output buffer, released with *d_bwt) */
int rb3gpu_sorter_sort_uploaded_sa(rb3gpu_sorter_t *s, int64_t len, void **d_bwt, void **d_tw, void **d_sa);
int rb3gpu_sorter_sort_sa(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text, void **d_bwt, void **d_tw, void **d_sa);
/* cumulative times of a sorter: text upload (host -> HBM, through its pinned staging buffer) and suffix sorting proper */
int rb3gpu_sorter_stats(const rb3gpu_sorter_t *s, double *ms_upload, double *ms_sort, int64_t *n_batches, int64_t *n_symbols);

/* Import for `build -i` (rb3_enc_fmd2fmr fm-index.c:56-85, mr_restore mrope.c:161-177):
 * runs[i] = len<<3 | sym in BWT order (host memory). */
int rb3gpu_from_runs(rb3gpu_t *h, int64_t n_runs, const uint64_t *runs);

/* rb3_enc_fmd2fmr (fm-index.c:56-85: rld_dec over the whole file + rope inserts) with the decoding on the device:
 * `words` (host) is the word stream of an FMD file -- what follows the 80-byte header, n_words = n_bytes / 8
 * (rld0.c:218-243) --, mcnt[6] the symbol counts of the header (NULL: not checked).  One thread decodes one 64-byte
 * block (blocks are self-contained, rld0.h:85-122); the symbols are built into the block array as by
 * rb3gpu_from_plain_dev.  RB3GPU_ESYMBOL: not a valid stream. */
int rb3gpu_from_fmd_words(rb3gpu_t *h, int64_t n_words, const uint64_t *words, const int64_t mcnt[RB3GPU_ASIZE]);
/* the same stream decoded on the device and merged into the index the handle holds as one batch, like rb3gpu_merge_plain
 * (`ropebwt3 merge`, main.c:84-133, with the right-hand index taken as its BWT) */
int rb3gpu_merge_fmd_words(rb3gpu_t *h, int64_t n_words, const uint64_t *words, const int64_t mcnt[RB3GPU_ASIZE]);

/* rb3_fmi_merge(fa, fb), fm-index.c:251-277 (`ropebwt3 merge`, main.c:84-133) between two handles: every string of the index in
 * `src` is merged into `h` as one batch (its sentinels rank after those of `h`, fm-index.c:147).  The two may sit on different
 * GPUs: the plain BWT of `src` goes device to device (xGMI).  This is the tree step of a partitioned multi-GPU build
 * (`ropebwt3-amd build --gpus N`): slices of the input indexed on N devices, then merged pairwise in input order. */
int rb3gpu_merge_index(rb3gpu_t *h, rb3gpu_t *src);

/* INTERVAL-SHARDED INDEX over several GPUs (one process and one handle per GPU; no reference analogue beyond the walk over
 * the per-rope totals in mr_rank2a, mrope.c:76-88, and the kt_for over strings, fm-index.c:217-224).  The accumulated BWT is
 * cut into n_iv contiguous intervals of positions, iv_bounds[i] .. iv_bounds[i+1]; the handle of rank i holds the block array
 * of interval i only (build it with rb3gpu_from_plain[_dev] from that slice of the BWT).  A chain STATE is the text position
 * of the current suffix of a string of the batch and its insertion point ka in the whole BWT; it lives on the rank whose
 * interval contains ka.  The batch (its text-order words d_tw, rb3gpu_sort_text) is replicated on every rank.
 *   rb3gpu_sh_step   one LF step (fm-index.c:166-173) for the n_states states resident here: ka is recorded for the suffix's
 *                    row in d_ka (len of the batch int64, -1 = not recorded), the next states are written to d_send GROUPED
 *                    BY THE INTERVAL THAT OWNS THEM, counts[d] of them for interval d (the split sizes of the all-to-all),
 *                    counts[n_iv] = chains that reached the start of their string.  adj[c] = C[c] of the whole BWT + symbols c
 *                    in the intervals before this one - C[c] of this interval (rb3gpu_get_acc of every rank, all-gathered).
 *   rb3gpu_sh_finish the rows jlo .. jlo + n_rows - 1 of the batch are the ones whose insertion points lie in this interval
 *                    (ka is monotone in the row number, so they are contiguous, and this rank recorded every one of them):
 *                    interleave them into the interval (worker_mgins, fm-index.c:237-249) and rebuild its block array. */
#define RB3GPU_SH_MAXIV 64
typedef struct { int64_t tp, ka; } rb3gpu_state_t;
int rb3gpu_sh_step(rb3gpu_t *h, int64_t n_states, const rb3gpu_state_t *d_in, const uint64_t *d_tw, int64_t *d_ka, const int64_t adj[RB3GPU_ASIZE],
		int n_iv, const int64_t *iv_bounds, int my_iv, rb3gpu_state_t *d_send, int64_t *counts);
int rb3gpu_sh_finish(rb3gpu_t *h, int64_t jlo, int64_t n_rows, const uint8_t *d_bwt, const int64_t *d_ka, int64_t iv_start, int commit);

/* The whole merge of one batch into the interval-sharded index, DRIVEN FROM THE LIBRARY (rb3_mg_rank_plain + worker_mgins,
 * fm-index.c:202-249, with the kt_for over strings of 217-224 cut by interval instead of by thread): every rank calls
 * rb3gpu_sh_merge with the same batch and bounds; per lock-step round there is ONE kernel (k_sh_round: LF step, record, next
 * states written straight into per-destination send regions), one read-back of the split sizes, one all-gather of them and one
 * all-to-all of 16-byte states -- or, over a communicator with stream_barrier (below), ONE kernel per round and nothing else.  The rows that land in an interval are kept as (row, insertion point) pairs -- 16 bytes per
 * landed row, nothing of the size of the whole batch -- and placed when the walk is over; then the interval is rebuilt.
 * What connects the ranks is a COMMUNICATOR of two collectives (host vectors of int64; device buffers of states):
 *   all_gather  every rank contributes n int64, recv gets world * n of them in rank order
 *   all_to_all  region d of this rank's send buffer (d_send + d * stride states, send_cnt[d] states) goes to rank d; d_recv
 *               receives recv_cnt[s] states from every rank s, packed in rank order.  `stream` is the HIP stream the engine's
 *               kernels run on: the send regions are complete on it, and d_recv must be usable on it when the call returns.
 * Both return 0 or a negative RB3GPU_E* code; a rank that fails must make the others fail too (abort), not leave them waiting.
 * Three communicators ship with the library: ranks as THREADS of one process with one device each (rb3gpu_group_*: barriers +
 * peer copies over xGMI -- what `ropebwt3-amd build --gpus N --interval` uses), one PROCESS per GPU over RCCL (rb3gpu_rccl_*:
 * grouped ncclSend/ncclRecv on the engine's stream, librccl loaded at run time), and any pair of callbacks (tests: gloo).
 *   iv_bounds  world + 1 positions, the same on every rank; updated to the bounds after the merge when commit != 0
 *   chain_tp   text positions of the batch's sentinels (host, n_chains of them, the same on every rank): where the chains start
 *   n_rounds   (may be NULL) lock-step rounds this merge took = longest string + 1 */
typedef struct rb3gpu_comm_s {
	void *ctx;
	int rank, world;
	int (*all_gather)(void *ctx, const int64_t *send, int n, int64_t *recv);
	int (*all_to_all)(void *ctx, const rb3gpu_state_t *d_send, int64_t stride, const int64_t *send_cnt, rb3gpu_state_t *d_recv, const int64_t *recv_cnt, void *stream);
	void (*abort)(void *ctx);   /* may be NULL: called by a rank whose merge failed locally, so that the others do not wait for it */
	/* may be NULL.  Not NULL says two things: (1) the ranks can address each other's device memory (one process, peer access: a pointer that
	 * all_gather carried from another rank can be given to a kernel here), and (2) stream_barrier(ctx, stream) makes everything this rank queues
	 * on `stream` AFTER the call wait for everything EVERY rank queued on its stream BEFORE its call -- without waiting for the devices (events the
	 * streams wait for, not a host synchronisation).  With it the lock-step rounds of rb3gpu_sh_merge[_text] run as PEER ROUNDS: one kernel per
	 * rank and round that writes the next states straight into the owner's receive buffer over xGMI, no read-back, no all-gather, no all-to-all. */
	int (*stream_barrier)(void *ctx, void *stream);
	/* may be NULL (then a device pointer means the same on every rank: threads of one process).  Ranks that are PROCESSES name a device buffer to each other
	 * by a handle of 8 words (peer_export: 0 or a negative code) which all_gather carries and the other side turns into a pointer of its own address space
	 * (peer_import: NULL if it cannot; with handle == NULL: the rank named is about to replace its buffers -- whatever is mapped of them here is given up, NULL is
	 * returned) -- HIP IPC memory handles in rb3gpu_ipc_peer_enable below. */
	int (*peer_export)(void *ctx, void *d_ptr, int64_t handle[8]);
	void *(*peer_import)(void *ctx, int rank, const int64_t handle[8]);
} rb3gpu_comm_t;
int rb3gpu_sh_merge(rb3gpu_t *h, const rb3gpu_comm_t *comm, int64_t *iv_bounds, int64_t len, const uint8_t *d_bwt, const uint64_t *d_tw,
		int64_t n_chains, const int64_t *chain_tp, int commit, int64_t *n_rounds);
/* The same with the BATCH sharded as well (what rb3gpu_shard_merge runs): every rank holds d_tprev -- the symbol before every text position,
 * one byte each, the only thing a step needs of the batch (rb3gpu_tprev_from_tw makes it from the text-order words) -- and d_tw_slice, the
 * text-order words of ITS text range [len * rank / world, len * (rank + 1) / world) only.  A record is (text position, insertion point); when
 * the chains have ended, one all-to-all takes the records to the owners of their text positions and one brings row << 3 | symbol back (the same
 * 16-byte collective as the rounds'): replicated 1 byte per batch symbol instead of 9, and 16 bytes per symbol cross the links once instead of
 * 8 (world - 1) out of one device. */
int rb3gpu_sh_merge_text(rb3gpu_t *h, const rb3gpu_comm_t *comm, int64_t *iv_bounds, int64_t len, const uint8_t *d_tprev, const uint64_t *d_tw_slice,
		int64_t n_chains, const int64_t *chain_tp, int commit, int64_t *n_rounds);
int rb3gpu_tprev_from_tw(rb3gpu_t *h, int64_t len, const uint64_t *d_tw, uint8_t *d_out);

/* ranks = threads of ONE process, one handle (device) each.  rb3gpu_group_create(world) once, rb3gpu_group_comm(g, rank, h, &comm)
 * by each rank's thread; rb3gpu_group_abort wakes every rank waiting in a collective (they return RB3GPU_ESTATE). */
typedef struct rb3gpu_group_s rb3gpu_group_t;
rb3gpu_group_t *rb3gpu_group_create(int world);
int rb3gpu_group_comm(rb3gpu_group_t *g, int rank, rb3gpu_t *h, rb3gpu_comm_t *comm);
void rb3gpu_group_abort(rb3gpu_group_t *g);
void rb3gpu_group_destroy(rb3gpu_group_t *g);

/* one PROCESS per GPU over RCCL.  Rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the others by whatever means the
 * launcher has (bench.py: torch.distributed broadcast; MPI; a file); every rank then creates its communicator on its handle's
 * device.  RB3GPU_EUNSUP: librccl.so could not be loaded. */
#define RB3GPU_RCCL_ID_BYTES 128
int rb3gpu_rccl_unique_id(char id[RB3GPU_RCCL_ID_BYTES]);
int rb3gpu_rccl_comm_create(rb3gpu_t *h, int rank, int world, const char id[RB3GPU_RCCL_ID_BYTES], rb3gpu_comm_t *comm);
void rb3gpu_rccl_comm_destroy(rb3gpu_comm_t *comm);

/* PEER ROUNDS for ranks that are PROCESSES of one node (bench.py --gpus N, torchrun: one process per GPU): wraps a communicator of world > 1 -- any of the above --
 * so that it offers stream_barrier / peer_export / peer_import: the receive buffers and counter tables are shared through HIP IPC memory handles
 * (hipIpcGetMemHandle / hipIpcOpenMemHandle, opened once per buffer); between two rounds every rank waits for its own stream and the processes meet at a spin
 * barrier in POSIX shared memory (streams of different processes cannot wait for each other on this runtime: hipStreamWaitEvent refuses events that came through
 * hipIpcOpenEventHandle) -- one host synchronisation per lock-step round, but no collective, no read-back, no copy.  COLLECTIVE: every rank calls it with its
 * handle and its communicator; it returns 0 on every rank (enabled everywhere) or RB3GPU_EUNSUP on every rank (some rank could not: the communicator is
 * left as it was).  rb3gpu_ipc_peer_disable undoes it (before the wrapped communicator is destroyed). */
int rb3gpu_ipc_peer_enable(rb3gpu_t *h, rb3gpu_comm_t *comm);
void rb3gpu_ipc_peer_disable(rb3gpu_comm_t *comm);

/* The interval-sharded index as ONE object for a single-process host program (`ropebwt3-amd build --gpus N --interval`): N handles,
 * one per device, a thread per handle (they live as long as the object), the thread-group communicator above between them.  Nothing of
 * the index, and of a batch nothing but one byte per symbol, is ever whole on one device:
 *   rb3gpu_shard_split   h0 holds a whole index (the first batch): it is cut into n intervals that hold about equal BYTES of the block array
 *                        (rb3gpu_balanced_bounds), an interval at a time (its symbols, 1 byte each, device to device); h0 keeps the first,
 *                        n - 1 new handles (options *opt, devices[1..n-1]; devices[0] must be h0's) get the others.  NULL: fewer symbols
 *                        than intervals, or a device / memory error.
 *   rb3gpu_shard_merge   one batch (text-order words on h0's device -- d_bwt is not read --, sentinel positions on the host): every rank
 *                        pulls the symbol-before array (1 byte per symbol) and ITS text range of the text-order words (8 bytes per symbol / n)
 *                        over its own xGMI link, all at once; then n threads run rb3gpu_sh_merge_text.
 *   rb3gpu_shard_export_runs / _run_words   the runs of the intervals in rank order, joined where they meet at a seam (the reference writes
 *                        its ropes one after the other and rld_enc joins them, fm-index.c:31-54, rld0.c:153-161): what the FMD / FMR /
 *                        plain writers take -- no gather.  rb3gpu_shard_get_acc: the cumulative symbol counts of the whole index.
 *   rb3gpu_shard_destroy the other handles destroyed, the object freed; h0 still holds interval 0.
 *   rb3gpu_shard_gather  the intervals put back together in h0 (plain symbols, the WHOLE index on h0's device: only for a caller that must
 *                        merge a batch the ordinary way), the other handles destroyed and the object freed.
 *   rb3gpu_shard_handle  the handle of interval i (statistics, tests); rb3gpu_shard_bounds copies the n + 1 bounds, returns n. */
typedef struct rb3gpu_shard_s rb3gpu_shard_t;
rb3gpu_shard_t *rb3gpu_shard_split(rb3gpu_t *h0, int n, const int *devices, const rb3gpu_opt_t *opt);
int rb3gpu_shard_merge(rb3gpu_shard_t *s, int64_t len, const uint8_t *d_bwt, const uint64_t *d_tw, int64_t n_chains, const int64_t *chain_tp, int64_t *n_rounds);
int rb3gpu_shard_gather(rb3gpu_shard_t *s);
int rb3gpu_shard_get_acc(const rb3gpu_shard_t *s, int64_t acc[RB3GPU_ASIZE + 1]);
int rb3gpu_shard_export_runs(rb3gpu_shard_t *s, rb3gpu_emit_f emit, void *data);
int rb3gpu_shard_export_run_words(rb3gpu_shard_t *s, rb3gpu_emit_words_f emit, void *data);
void rb3gpu_shard_destroy(rb3gpu_shard_t *s);
/* bounds moved to where equal shares of the block array's BYTES lie, if the largest interval holds more than pct per cent more than the mean (pct < 0:
 * always), every interval rebuilt from its new range on its own device; called by rb3gpu_shard_merge after every batch with pct = 25 (SURVEY 8(e);
 * RB3GPU_SHARD_REBALANCE_PCT overrides, -1 = never).  1: rebalanced, 0: not needed, < 0: error */
int rb3gpu_shard_rebalance(rb3gpu_shard_t *s, int pct);
rb3gpu_t *rb3gpu_shard_handle(rb3gpu_shard_t *s, int i);
int rb3gpu_shard_bounds(const rb3gpu_shard_t *s, int64_t *bounds);

/* the HIP device and stream of a handle (for communicators implemented outside the library) */
int rb3gpu_device_of(const rb3gpu_t *h);
void *rb3gpu_stream_of(const rb3gpu_t *h);
/* wait for a stream handed to a communicator's all_to_all (a communicator that moves the send regions with anything that is not ordered behind that stream -- the
 * runtime's synchronous copies, a library on another stream -- calls this first: the engine's streams are non-blocking) */
int rb3gpu_stream_sync(void *stream);

/* Diagnostic switches of a handle (no reference analogue; none is needed in normal use).  Each key is also read ONCE from
 * the environment variable RB3GPU_<KEY> when the handle is created; the merge path itself never calls getenv().
 *   "tent" 0/1, "staged" 0/1, "group_rebuild" 0/1, "window_rebuild" 0/1, "reb_force" 0/1, "octs" 1..8, "blkmul", "blkcap", "ssa_split" 4..20,
 *   "lf_check" n (verify the LF relation of every n-th batch row against the index after each merge; default 4096, 0 = off)
 *   "tent_q" 1/2/4/8 (width of the drop-out masks in units of 256 bits; 0 = follows what the walkers report), "trec" 0/1/-1 (records of a
 *   text-order walk in text order: never / always / where the index does not fit the caches), "copy_walkers" 0/1, "b2_split" S (splitter
 *   spacing 2^S of the walker list the engine makes for the BWT-only entry points; -1 = by the size of the batch), "abs_limit" N (indexes
 *   of fewer than N symbols carry the LF base in their slot headers; at most 2^32, only before an index exists: RB3GPU_ESTATE after);
 *   the full table with defaults is in docs/LAB_NOTEBOOK.md section 8c
 * Test hooks "force_fallback", "tent_limit", "text_mode" exist only in the test build of the library (compiled with
 * -DRB3GPU_TEST_HOOKS, librb3gpu_hooks.so); the release library answers RB3GPU_EUNSUP.  Unknown key: RB3GPU_EINVAL. */
int rb3gpu_tune(rb3gpu_t *h, const char *key, int64_t value);

int rb3gpu_stats(const rb3gpu_t *h, rb3gpu_stats_t *st);
/* what the handle holds right now, buffer by buffer (diagnostics: which scratch makes up bytes_peak): i = 0, 1, ... until RB3GPU_EINVAL; *name is a static
 * string.  (No counterpart in the reference: its memory is the rope's, mrope.c:15-34.) */
int rb3gpu_buffer_bytes(const rb3gpu_t *h, int i, const char **name, int64_t *bytes);
void rb3gpu_stats_reset(rb3gpu_t *h);

/* device scratch helpers so that callers without a HIP binding (C host code, ctypes) can
 * keep a partial BWT resident in HBM: plain hipMalloc/hipMemcpy/hipFree on the handle's device */
int rb3gpu_dev_alloc(rb3gpu_t *h, int64_t n_bytes, void **d_ptr);
int rb3gpu_dev_upload(rb3gpu_t *h, void *d_dst, const void *src, int64_t n_bytes);
int rb3gpu_dev_download(rb3gpu_t *h, void *dst, const void *d_src, int64_t n_bytes);
int rb3gpu_dev_free(rb3gpu_t *h, void *d_ptr);
int rb3gpu_dev_copy(rb3gpu_t *h, void *d_dst, const void *d_src, int64_t n_bytes);   /* device to device */
int rb3gpu_dev_memset(rb3gpu_t *h, void *d_dst, int byte, int64_t n_bytes);
int rb3gpu_sync(rb3gpu_t *h);

/* number of HIP devices visible, or a negative error */
int rb3gpu_device_count(void);

#ifdef __cplusplus
}
#endif

#endif
