/*
 * rb3gpu.hip -- host side of librb3gpu.so: the C ABI declared in include/rb3gpu.h on top of
 * the gfx950 kernels in rb3gpu_kernels.h.  One handle = one HIP device + one stream.
 *
 * The call sequence of a merge mirrors rb3_fmi_merge_plain (fm-index.c:279-303):
 *   C array of B1 (286-287)            -> kept in the handle / folded into the group directory
 *   rb3_mg_rank_plain (288-289)        -> k_tile_hist + scan + k_lf2, then k_chain
 *   kt_for(worker_mgins) (294-299)     -> k_pos_finalize_check_rows + k_pass1w + k_decide + scan + k_pass2w
 *                                         (k_group_rows + k_pass1 + scan + k_pass2 when the window scratch would be too big)
 * Suffix sorting of a batch (rb3gpu_sort.hip) and FMD packing (rb3gpu_fmdenc.hip) are separate translation units
 * because they use rocPRIM's sorts and scans; nothing on the merge path depends on them.
 */
#include <hip/hip_runtime.h>
#include <vector>
#include <map>
#include <algorithm>
#include <utility>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <new>
#include "rb3gpu.h"
#include "rb3gpu_kernels.h"
#include "rb3gpu_planes.h"
#include "rb3gpu_part.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
		if (h && h->opt.verbose >= 1) fprintf(stderr, "[E::rb3gpu] %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
		return e_ == hipErrorOutOfMemory ? RB3GPU_ENOMEM : RB3GPU_ENODEV; } } while (0)

#define RB3_STAGE_BYTES ((size_t)32 << 20) // pinned host staging buffers

/* the suffix sorter lives in rb3gpu_sort.hip (it pulls in rocPRIM; kept out of this translation unit) */
struct rb3sort_ws;
rb3sort_ws *rb3sort_create(void);
void rb3sort_destroy(rb3sort_ws *ws);
int64_t rb3sort_bytes(const rb3sort_ws *ws);
int rb3sort_bwt(rb3sort_ws *ws, hipStream_t st, int64_t n, const uint8_t *d_text, uint8_t *d_bwt, int64_t step, int64_t *d_ckrow, int *rounds, uint64_t *d_tw, uint32_t *d_sa);
/* the FMD packer lives in rb3gpu_fmdenc.hip */
int rb3fmd_encode(hipStream_t st, int64_t n_sym, int64_t nr, const uint64_t *d_words, uint64_t **z_out, int64_t *n_words);
struct rb3fmd_enc;
int rb3fmd_enc_begin(hipStream_t st, int64_t n_sym, int64_t cap_runs, rb3fmd_enc **e);
uint64_t *rb3fmd_enc_buffer(rb3fmd_enc *e, int64_t *room);
int rb3fmd_enc_piece(rb3fmd_enc *e, int64_t n_new, int final);
int rb3fmd_enc_end(rb3fmd_enc *e, uint64_t **z_out, int64_t *n_words);
void rb3fmd_enc_abort(rb3fmd_enc *e);
struct rb3fmd_dec;
int rb3fmd_decode_begin(hipStream_t st, int64_t n_words, const uint64_t *d_z, rb3fmd_dec **ctx, int64_t *n_sym);
int rb3fmd_decode_fill(rb3fmd_dec *ctx, uint8_t *d_plain);
int rb3fmd_decode_range(rb3fmd_dec *ctx, int64_t p0, int64_t p1, uint8_t *d_out);
void rb3fmd_decode_end(rb3fmd_dec *ctx);
int64_t rb3fmd_decode_bytes(const rb3fmd_dec *ctx);

struct Buf {
	void *p = nullptr;
	size_t cap = 0;
};

/* Page-locked host memory handed out by rb3gpu_pinned_alloc: a batch that the reader wrote straight into such a buffer goes to
 * HBM with ONE DMA at PCIe speed; anything else is copied into pinned staging buffers first, chunk by chunk (a single host
 * thread copies at 10-25 GB/s, i.e. slower than the link).  The registry says which is which. */
#include <pthread.h>
static pthread_mutex_t g_pin_mtx = PTHREAD_MUTEX_INITIALIZER;
static std::vector<std::pair<const uint8_t*, size_t> > g_pinned;

static bool is_pinned(const void *p, size_t bytes)
{
	bool hit = false;
	pthread_mutex_lock(&g_pin_mtx);
	for (const auto &r : g_pinned)
		if ((const uint8_t*)p >= r.first && (const uint8_t*)p + bytes <= r.first + r.second) { hit = true; break; }
	pthread_mutex_unlock(&g_pin_mtx);
	return hit;
}

/* host -> device copy of n bytes on stream st: direct from page-locked memory, else through the two pinned staging buffers
 * (the CPU fills one while the DMA engine empties the other).  Returns with the copy complete. */
static hipError_t h2d_copy(void *d_dst, const uint8_t *src, size_t n, hipStream_t st, uint8_t *stage[2], hipEvent_t done[2])
{
	hipError_t e;
	if (n == 0) return hipSuccess;
	if (is_pinned(src, n) || !stage || !stage[0] || !stage[1] || n <= (size_t)(256 << 10)) {
		if ((e = hipMemcpyAsync(d_dst, src, n, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
		return hipStreamSynchronize(st);
	}
	// chunks of 4 MB: the first DMA starts after 4 MB have been staged, not after 32
	const size_t chunk = (size_t)4 << 20;
	bool used[2] = { false, false };
	int i = 0;
	for (size_t off = 0; off < n; off += chunk, i ^= 1) {
		const size_t k = n - off < chunk ? n - off : chunk;
		if (used[i] && (e = hipEventSynchronize(done[i])) != hipSuccess) return e; // the DMA that last read this staging buffer
		memcpy(stage[i], src + off, k);
		if ((e = hipMemcpyAsync((uint8_t*)d_dst + off, stage[i], k, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
		if ((e = hipEventRecord(done[i], st)) != hipSuccess) return e;
		used[i] = true;
	}
	return hipStreamSynchronize(st);
}

/* Diagnostic switches of a handle.  They are read from the environment ONCE, in rb3gpu_create (RB3GPU_<KEY>), and can be
 * changed on a live handle with rb3gpu_tune(); nothing on the merge path calls getenv().  The keys under
 * RB3GPU_TEST_HOOKS only exist in the test build of the library (librb3gpu_hooks.so): the release build has no code
 * that pretends a failure. */
struct Tune {
	int tent = 1;            // tentative records (0: every inexact walker runs until it is exact)
	int staged = 0;          // the three-stage merge (several host syncs) instead of the single-sync one
	int group_rebuild = 0;   // group-sequential rebuild kernels instead of the window-parallel ones
	int window_rebuild = 0;  // the per-window rebuild (k_pass1w) instead of the run-space rebuild per group
	int scan_place = 1;      // the scan over the groups and the placement of their slots in one kernel (k_scan_place); 0: three scan kernels + k_place / k_place_pg + k_grp_compact
	int part = 0;            // 1: records in text order reach pos[] by a two-pass partition (k_part_*, rb3gpu_part.h) where the batch is large; 2: always; 0 (default): by the gather
	                         // through the suffix array -- measured on 302 M rows: scatter 2.7 ms + place 3.7 ms + streaming validation against 7.3 ms of gather: no gain, so off
	int reb_t1_rows = 96;    // the small tier of the run-space rebuild runs first where a group receives at most this many batch rows on average
	int plane_rebuild = 1;   // symbols are rebuilt in plane space, a lane per 32 symbols (k_plane_group); 0: a wave per window (k_pass1w / k_decide / k_pass2w, rounds 1-3)
	int resolve_v1 = 0;      // settle the tentative stretches with k_resolve (one hop per stretch) instead of k_cum / k_resolve_w / k_sfin
	int reb_force = 0;       // the run-space rebuild whatever the row density and the old index look like (tests: the hand-over paths)
	int octs = 8;            // octets per wave of k_chain
	int lpw = 8;             // (lanes per walker of k_chain: an octet.  The key is still accepted; the quad variant of rounds 2-5 is gone)
	int blkmul = 1;          // launch width multiplier of k_chain
	int64_t blkcap = 2048;   // block cap of k_chain
	int trec = -1;           // records of a text-order walk in text order (needs the batch's suffix array): 1 always, 0 never, -1: where the index does not fit the caches
	int64_t abs_limit = INT64_MAX; // indexes of fewer symbols carry the LF base in their slot headers -- whole below 2^32 symbols, its low half + the table IdxView.sb from there on (round 6;
	                               // rounds 3-5: 2^32, beyond which the headers counted from the group start and a rank read the 64-byte directory entry as well); set before an index exists
	int abs_table = 0;             // (tests) the table and the low-half arithmetic below 2^32 symbols too
	int tent_q = 0;          // width of the drop-out masks of the tentative stretches in units of 256 bits: 1, 2, 4, 8; 0: follows what the walkers report
	int copy_walkers = 0;    // a walker list in page-locked memory is copied to the device all the same (instead of being read in place)
	int lf_after = 1;        // the batch's histogram kernels (side stream) wait for the walkers to finish instead of running beside their first steps (measured: 186.3 -> 184.3 ms per build; 0: beside the walkers)
	int chain_bs = 256;      // threads per block of k_chain in the single-sync merge (64, 128 or 256: the kernel has no block-level state; smaller blocks spread the waves more evenly over the CUs)
	int ssa_split = 8;       // splitter spacing 2^S of the sampled-suffix-array walk
	int b2_split = -1;       // splitter spacing 2^S of the batch's own LF walk (walkers for the BWT-only entry point); 0: SA-regular walkers (staged path); -1: by the size of the batch (merge_core)
	int log_alloc = 0;       // print every device allocation and the time it took
	int poison = 0;          // fill every new device buffer with 0xA5 bytes (debugging: nothing may rely on what fresh memory holds)
	int guard = 0;           // (debugging) 4 KB of fill pattern behind every buffer of the handle, verified after every merge
	int vmm = 1;             // (values above 1: the threshold in KB instead of 64 MB -- tests make every buffer a growable range) buffers of 64 MB and more are ranges of reserved device address space that grow IN PLACE, a few physical chunks at a time (vm_ensure; round 6); 0: hipMalloc + reallocation
	int b2_tw = 1;           // a batch that comes as its BWT only (the reference's signature) gets its text-order words from its own sparse LF walk and is merged like one that came with them (round 6); 0: walkers over row words (rounds 2-5)
	int vmm_reserve = 0;     // (tests) MB of address space a new range reserves instead of 32 x its size (at least 16 GB)
	int defer_free = 1;      // keep replaced buffers on a list and hipFree them in bulk (0: at once; hipFree waits for every stream of the device)
	int64_t fmd_piece = 0;   // runs the FMD packer takes at a time (0: 64 M; rb3gpu_export_fmd_words)
	int sh_host_rounds = 0;  // rb3gpu_sh_merge with ONE interval: the host reads the split sizes back after every round, as with several (0: the rounds run back to back on the device)
	int sh_block = 0;        // threads per block of k_sh_round at eight states per octet: 256 or 1024; 0: 1024 below 3 M chains
	int sh_states = 0;       // states per octet of k_sh_round (1, 2, 4, 8); 0: by the number of chains
	int lf_check = 4096;     // sampled LF-consistency check of pos[] after every merge: every n-th row (0: off)
	int junction_check = 1;  // (round 6, last session: 1 = the events of EVERY stretch -- 16 before: every 16th; beside the rebuild the full check costs the 152-genome build 2.3 of 167 ms, every 2nd 0.4, every 4th nothing) ... and the LF relation at the junctions of the speculative walk (k_junction_check): wherever a walker met somebody's record -- all
	                         // of them, always -- and at the drop-out events of every n-th stretch id (1: every event, ~10 ms per 152-genome build; 0: off)
	int ev_blocks = 2048, cum_blocks = 2048, resw_blocks = 512, sfin_blocks = 2048; // launch widths of the settle kernels (k_events, k_cum, k_resolve_w, k_sfin); round 5: k_cum 1024 -> 2048 (-1 ms per 152-genome build), the others make no difference (profiles/r5_ab_settle_widths.txt)
	int64_t load_chunk = 16384; // groups (of 8192 symbols) an FMD stream is decoded and built by at a time when it holds more than that (rb3gpu_from_fmd_words)
#ifdef RB3GPU_TEST_HOOKS
	int hide_first = 0;      // k_chain: exact walkers do not see the tentative records of first stretches (the late-walker race of DESIGN.md, made deterministic)
	int force_fallback = 0;  // pretend the tentative pass left unsettled records
	int64_t tent_limit = -1; // shrink the stretch table
	int text_mode = 0;       // force how the text-order words are fetched (1: per lane, 2: 64 bytes per octet)
	int64_t reb_lcap = 0;    // > 0: entries of the hand-over list the window scratch takes (a longer list: rebuild done again)
	int64_t reb_slot_cap = 0;// > 0: pretend the slot array of a single-sync rebuild holds this many slots
	int corrupt_pos = 0;     // move a range of rows of pos[] by one after the walk (still monotone): the LF check must notice
	int64_t corrupt_sfin = -1; // >= 0: give the n-th settled event stretch of the merge a wrong unknown (k_test_corrupt_sfin): the junction check must notice
	int64_t pos_limit = 0;   // > 0: pretend merged positions must stay below this instead of 2^38 (beyond: staged path, no tentative records)
	int64_t win_scratch = 0; // > 0: pretend the window kernels' scratch may take this many bytes instead of 8 GB (beyond: group-sequential rebuild)
	int64_t slot_bytes = 0;  // > 0: pretend the upper bound of the slot array may take this many bytes instead of 16 GB (beyond: staged path)
#endif
};

/* A buffer that GROWS IN PLACE (round 6).  The buffers of an index that grows round by round -- the two slot arrays, the rows' positions, the scratch of
 * the rebuild -- used to be given up and obtained again, an eighth larger, every few merges: device memory costs ~37 us per MB to obtain, a 10 GB slot
 * array 0.4 s, and 60 M reads (18 G symbols) spent 4.0 of their 14 s in 108 hipMalloc calls with 69 GB of device memory around a 9.2 GB index (the slack,
 * and the replaced buffers waiting for their hipFree).  Now such a buffer is a RANGE OF RESERVED ADDRESS SPACE (hipMemAddressReserve: the device has
 * 2^48 bytes of it) into which physical chunks are mapped as it grows (hipMemCreate + hipMemMap): growing costs the new bytes only, nothing is copied,
 * replaced or left over, the contents stay, and the address never changes. */
struct VmRange {
	void *va = nullptr;
	size_t va_size = 0, mapped = 0;
	std::vector<std::pair<hipMemGenericAllocationHandle_t, size_t> > hs; // physical chunks in address order
};

struct rb3gpu_s {
	int dev = 0;
	Tune tn;
	hipStream_t st = nullptr;
	hipStream_t st2 = nullptr;  // side stream: the sampled LF check of pos[] runs beside the rebuild
	hipEvent_t evx[3];
	unsigned long long *hm_pin = nullptr; // page-locked landing place of the counters a merge reads back
	rb3gpu_opt_t opt;
	rb3gpu_stats_t stt;
	// the index: grp/slots point into ib[cur]; a merge builds into ib[1-cur] and swaps on commit
	int64_t n = 0, ngrp = 0, nslots = 0;
	int64_t acc[7] = {0, 0, 0, 0, 0, 0, 0};
	rb3_grp_t *grp = nullptr;
	rb3_slot_t *slots = nullptr;
	struct { rb3_grp_t *grp; size_t grp_cap; rb3_slot_t *slots; size_t slots_cap; uint64_t *sbt; size_t sbt_cap; } ib[2] = {{nullptr, 0, nullptr, 0, nullptr, 0}, {nullptr, 0, nullptr, 0, nullptr, 0}};
	// (sbt: the table of LF bases every 2^31 symbols, IdxView.sb, of an index of 2^32 symbols and more -- sbt_cap entries of 8 words)
	// (behind the grp_cap directory entries of a buffer sit grp_cap 8-byte words: the compact copy of the entries' slot words, IdxView.gsm)
	int cur = 0;
	// scratch, grown on demand and kept between calls
	Buf b2, pos, post, tcnt, tpre, ctot, ctot2, gstat, gpre, jg, misc, xbuf, wl, wls, dl, dlx, wstat, wplane, wruns, gslots, glist, pslots, lbst; // (wls: the sentinels' text positions of a device-made walker list -- a buffer of its own, NOT dlx, which tent_masks() hands out as the wide drop-out masks and may reallocate: ADVICE r5)
	Buf shc, shn, shs, shr, shk;
	Buf shp0, shp1;        // peer rounds: the two receive buffers the OTHER ranks write into -- buffers of their own, never reused by the merge's last phase: a rank that is
	                       // another process has them mapped (HIP IPC), and a mapping of a buffer its owner has replaced meanwhile is what a merge must never meet
	Buf twb; // the text-order words of a batch that came as its BWT only (merge_plain_via_tw) // interval-sharded merge (rb3gpu_sh_merge): states of this and of the next round, send regions, landed (row, insertion point) pairs, counters
	// a merge in progress (rb3gpu_mg_begin .. rb3gpu_mg_finish)
	int mg_active = 0;
	int64_t mg_len = 0, mg_acc2[7] = {0, 0, 0, 0, 0, 0, 0};
	const uint8_t *mg_b2 = nullptr;
	int64_t *mg_pos = nullptr;
	hipEvent_t ev[8];
	int64_t bytes_owned = 0;
	size_t dev_mem = 0;         // the device's memory (what the scratch limits of the rebuild are fractions of)
	std::vector<VmRange*> vmr;  // the growable ranges of this handle (vm_ensure)
	int vm_state = 0;           // 0: not asked yet, 1: the device maps memory into reserved address space, -1: it does not (or tune vmm = 0): hipMalloc
	size_t vm_gran = 0, vm_total = 0; // granularity of a physical chunk, the device's memory (the most a range ever reserves)
	std::vector<std::pair<void*, size_t> > garbage; // replaced buffers not yet given back (dev_free)
	int64_t bytes_garbage = 0;
	uint64_t reb_slot_cap = 0; // capacity of the slot array the last rebuild emitted into (build_index)
	bool lf_wait = false;      // the main stream has not waited yet for the batch's histogram on the side stream (merge_core -> build_index)
	bool reb_prepared = false; // merge_core has cleared the counters of the run-space rebuild together with everything else
	double t0 = 0;
	uint8_t *stage[2] = {nullptr, nullptr}; // pinned staging buffers for host->device copies
	rb3sort_ws *sorter = nullptr;           // scratch of rb3gpu_bwt_from_text, created on first use
	unsigned long long lb_epoch = 0; // launches of k_scan_place so far (its look-back state is tagged with it instead of being cleared)
	bool reb_pp_all = false; // the last rebuilds handed most groups on to the symbol path: skip the run-space tiers until most groups qualify (merge_core keeps it up to date)
	int64_t reb_last[2] = {-1, -1}; // groups the first / the last tier of the run-space rebuild handed on in the merge before (-1: unknown)
	int64_t mg_step = 0;             // > 0 (rb3gpu_merge_text_step_dev): no list, a count of strings: the walker list is made on the device, a walker every mg_step text positions
	const uint32_t *mg_sa = nullptr; // the suffix array of the batch being merged, if its caller has it (rb3gpu_merge_text_sa_dev): records in text order
	int tent_q = 1;          // masks of 256 * tent_q bits (merge_core doubles it when walkers report intervals wider than that)
	int64_t sid_dirty[2] = {RB3_TENT_HALF, RB3_TENT_HALF}; // entries of the two halves of the stretch tables (dl) that may be non-zero
};

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + ts.tv_nsec * 1e-9;
}

#define RB3_GUARD 4096 /* tune "guard": bytes behind every buffer of the handle that must keep their fill pattern */

/* give the replaced buffers back.  hipFree waits for EVERY stream of the device -- the sorter thread's kernels included, tens of
 * milliseconds per call while a batch is being sorted -- and the buffers of an index that grows round by round are replaced all
 * the time (408 of the 536 ms of the merge path of a 152-genome build were spent there).  So a replaced buffer only goes on a
 * list, and the list is emptied when it has grown to half of what the handle holds (or 4 GB), when memory runs out, and when the
 * handle is destroyed.  (The stream-ordered allocator, hipMallocAsync, was tried instead: on ROCm 7.2 freshly obtained pool
 * memory was seen zeroed AFTER the first stream-ordered writes to it, which corrupted the index.) */
static void garbage_collect(rb3gpu_t *h, bool force)
{
	if (h->garbage.empty()) return;
	const int64_t live = h->bytes_owned - h->bytes_garbage, limit = live / 2 > ((int64_t)4 << 30) ? live / 2 : ((int64_t)4 << 30);
	if (!force && h->bytes_garbage <= limit) return;
	const double t0 = now_s();
	for (auto &g : h->garbage) (void)hipFree(g.first);
	h->stt.ms_alloc += (now_s() - t0) * 1e3;
	h->bytes_owned -= h->bytes_garbage, h->bytes_garbage = 0;
	h->garbage.clear();
}

static int dev_malloc(rb3gpu_t *h, void **p, size_t bytes)
{
	*p = nullptr;
	if (bytes == 0) bytes = 256;
	const size_t payload = bytes;
	if (h->tn.guard) bytes += RB3_GUARD;
	const double t0 = now_s();
	hipError_t e = hipMalloc(p, bytes);
	if (e == hipErrorOutOfMemory && !h->garbage.empty()) {
		(void)hipGetLastError();
		garbage_collect(h, true);
		e = hipMalloc(p, bytes);
	}
	h->stt.ms_alloc += (now_s() - t0) * 1e3, h->stt.n_allocs += 1;
	if (h->tn.log_alloc) fprintf(stderr, "[M::rb3gpu] hipMalloc of %.1f MB took %.3f ms -> %p\n", (double)bytes / 1e6, (now_s() - t0) * 1e3, *p);
	HIPCHK(e);
	if (h->tn.poison) HIPCHK(hipMemsetAsync(*p, 0xA5, bytes, h->st));
	if (h->tn.guard) HIPCHK(hipMemsetAsync((uint8_t*)*p + payload, 0xA5, RB3_GUARD, h->st));
	h->bytes_owned += (int64_t)bytes;
	if (h->bytes_owned > h->stt.bytes_peak) h->stt.bytes_peak = h->bytes_owned;
	return 0;
}

static VmRange *vm_find(rb3gpu_t *h, const void *p);
static void vm_release(rb3gpu_t *h, VmRange *r);

static void dev_free(rb3gpu_t *h, void *p, size_t bytes)
{
	if (p == nullptr) return;
	if (VmRange *r = vm_find(h, p)) { // (rare: a growable range is only given up with its handle, or when an index is dropped)
		const double t0 = now_s();
		(void)hipStreamSynchronize(h->st), (void)hipStreamSynchronize(h->st2);
		vm_release(h, r);
		h->stt.ms_alloc += (now_s() - t0) * 1e3;
		return;
	}
	if (bytes == 0) bytes = 256;
	if (h->tn.guard) bytes += RB3_GUARD;
	if (h->tn.defer_free) { // (every user of the buffer is on h->st or joined to it, and the list is only emptied by hipFree, which waits for the device)
		h->garbage.push_back(std::make_pair(p, bytes)), h->bytes_garbage += (int64_t)bytes;
		return;
	}
	const double t0 = now_s();
	(void)hipFree(p);
	h->stt.ms_alloc += (now_s() - t0) * 1e3;
	h->bytes_owned -= (int64_t)bytes;
}

#define RB3_VM_MIN ((size_t)64 << 20) /* smaller buffers: hipMalloc */

static bool vm_usable(rb3gpu_t *h, size_t bytes)
{
	if (h->vm_state == 0) {
		int ok = 0;
		h->vm_state = -1;
		if (h->tn.vmm && !h->tn.guard && hipDeviceGetAttribute(&ok, hipDeviceAttributeVirtualMemoryManagementSupported, h->dev) == hipSuccess && ok) {
			hipMemAllocationProp prop;
			memset(&prop, 0, sizeof(prop));
			prop.type = hipMemAllocationTypePinned, prop.location.type = hipMemLocationTypeDevice, prop.location.id = h->dev;
			size_t gran = 0, fr = 0, tot = 0;
			if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran > 0 && hipMemGetInfo(&fr, &tot) == hipSuccess && tot > 0)
				h->vm_gran = gran < ((size_t)2 << 20) ? ((size_t)2 << 20) : gran, h->vm_total = (tot + h->vm_gran - 1) / h->vm_gran * h->vm_gran, h->vm_state = 1;
		}
		(void)hipGetLastError();
	}
	return h->vm_state == 1 && bytes >= (h->tn.vmm > 1 ? (size_t)h->tn.vmm << 10 : RB3_VM_MIN);
}

static VmRange *vm_find(rb3gpu_t *h, const void *p)
{
	if (p) for (VmRange *r : h->vmr) if (r->va == p) return r;
	return nullptr;
}

static int vm_map_chunk(rb3gpu_t *h, VmRange *r, size_t off, hipMemGenericAllocationHandle_t hd, size_t size)
{
	{ const hipError_t e = hipMemMap((char*)r->va + off, size, 0, hd, 0);
	  if (e != hipSuccess) { if (h->tn.log_alloc) fprintf(stderr, "[M::rb3gpu] hipMemMap(%p + %zu, %zu): %s\n", r->va, off, size, hipGetErrorString(e)); (void)hipGetLastError(); return RB3GPU_ENOMEM; } }
	// (this device only: the ranges are buffers that nothing but this handle's kernels touch -- what peers pull over xGMI are ordinary allocations)
	hipMemAccessDesc ad;
	memset(&ad, 0, sizeof(ad));
	ad.location.type = hipMemLocationTypeDevice, ad.location.id = h->dev, ad.flags = hipMemAccessFlagsProtReadWrite;
	{ // Access for the new chunk alone -- or, where this runtime answers "invalid argument" to that (ROCm 7.2: for about half of all (address, size) pairs behind a first
	  // chunk -- tools/ubench/vmm_probe.cpp; at the scale of BASELINE configs[4] most growths of the slot arrays and of the rebuild's scratch came back like that and fell onto
	  // hipMalloc / hipMemCreate of the WHOLE buffer, 0.5-1.8 s apiece: 7.5 of a build's 19 s), for everything mapped so far from the range's base, which it always takes.
		hipError_t e = hipMemSetAccess((char*)r->va + off, size, &ad, 1);
		if (e != hipSuccess && off > 0) { (void)hipGetLastError(); e = hipMemSetAccess(r->va, off + size, &ad, 1); }
		if (e != hipSuccess) { if (h->tn.log_alloc) fprintf(stderr, "[M::rb3gpu] hipMemSetAccess(%p + %zu, %zu): %s\n", r->va, off, size, hipGetErrorString(e)); (void)hipGetLastError(); (void)hipMemUnmap((char*)r->va + off, size); return RB3GPU_ENOMEM; }
	}
	return 0;
}

/* *p (a range of this handle, or NULL) holds at least `bytes` mapped bytes afterwards; what it held stays; *cap = the mapped bytes.  < 0: nothing changed */
static int vm_ensure(rb3gpu_t *h, void **p, size_t *cap, size_t bytes)
{
	const size_t G = h->vm_gran;
	VmRange *r = vm_find(h, *p);
	const double t0 = now_s();
	if (r == nullptr) {
		r = new (std::nothrow) VmRange();
		if (!r) return RB3GPU_ENOMEM;
		size_t want = bytes * 32 < ((size_t)16 << 30) ? ((size_t)16 << 30) : bytes * 32; // address space is free: room for the buffer to grow thirty-two-fold where it stands
		if (h->tn.vmm_reserve > 0) want = (size_t)h->tn.vmm_reserve << 20; // (tests: a small reservation, so that ranges move to larger ones)
		want = (want + G - 1) / G * G;
		if (want > h->vm_total) want = h->vm_total;
		if (want < (bytes + G - 1) / G * G) want = (bytes + G - 1) / G * G;
		if (hipMemAddressReserve(&r->va, want, G, nullptr, 0) != hipSuccess || r->va == nullptr) {
			if (h->tn.log_alloc) fprintf(stderr, "[M::rb3gpu] vm_ensure: no address range of %.1f MB (%s)\n", (double)want / 1e6, hipGetErrorString(hipGetLastError()));
			(void)hipGetLastError(); delete r; return RB3GPU_ENOMEM;
		}
		r->va_size = want;
		h->vmr.push_back(r);
	}
	if (bytes > r->va_size && h->tn.log_alloc) fprintf(stderr, "[M::rb3gpu] vm_ensure: %.1f MB asked of a range of %.1f MB: it moves\n", (double)bytes / 1e6, (double)r->va_size / 1e6);
	if (bytes > r->va_size) { // the reserved range is used up (the buffer grew more than eightfold): a larger one, the same physical chunks mapped into it -- nothing is copied
		size_t want = r->va_size * 4 > bytes * 2 ? r->va_size * 4 : bytes * 2;
		want = (want + G - 1) / G * G;
		void *nva = nullptr;
		HIPCHK(hipStreamSynchronize(h->st));  // (kernels in flight use the old addresses)
		HIPCHK(hipStreamSynchronize(h->st2));
		if (hipMemAddressReserve(&nva, want, G, nullptr, 0) != hipSuccess || nva == nullptr) { (void)hipGetLastError(); return RB3GPU_ENOMEM; }
		VmRange nr;
		nr.va = nva, nr.va_size = want;
		size_t off = 0;
		for (auto &c : r->hs) { (void)hipMemUnmap((char*)r->va + off, c.second); off += c.second; }
		off = 0;
		size_t moved = 0;
		for (auto &c : r->hs) { if (vm_map_chunk(h, &nr, off, c.first, c.second) < 0) break; off += c.second, ++moved; }
		(void)hipMemAddressFree(r->va, r->va_size);
		r->va = nva, r->va_size = want;
		if (moved < r->hs.size()) { // (cannot be: the chunks were mapped a moment ago.  The range keeps what did move; the caller's buffer is gone)
			for (size_t i = moved; i < r->hs.size(); ++i) { (void)hipMemRelease(r->hs[i].first); h->bytes_owned -= (int64_t)r->hs[i].second; }
			r->hs.resize(moved), r->mapped = off;
			*p = r->va, *cap = r->mapped;
			return RB3GPU_EINTERNAL;
		}
	}
	if (bytes > r->mapped) {
		size_t inc = (bytes - r->mapped + G - 1) / G * G;
		hipMemAllocationProp prop;
		memset(&prop, 0, sizeof(prop));
		prop.type = hipMemAllocationTypePinned, prop.location.type = hipMemLocationTypeDevice, prop.location.id = h->dev;
		hipMemGenericAllocationHandle_t hd;
		hipError_t e = hipMemCreate(&hd, inc, &prop, 0);
		if (e == hipErrorOutOfMemory && !h->garbage.empty()) { (void)hipGetLastError(); garbage_collect(h, true); e = hipMemCreate(&hd, inc, &prop, 0); }
		if (e != hipSuccess && h->tn.log_alloc) fprintf(stderr, "[M::rb3gpu] vm_ensure: hipMemCreate of %.1f MB failed (%s)\n", (double)inc / 1e6, hipGetErrorString(e));
		if (e != hipSuccess) { (void)hipGetLastError(); if (r->mapped == 0) { (void)hipMemAddressFree(r->va, r->va_size); h->vmr.erase(std::find(h->vmr.begin(), h->vmr.end(), r)); delete r; } return RB3GPU_ENOMEM; }
		if (vm_map_chunk(h, r, r->mapped, hd, inc) < 0) { if (h->tn.log_alloc) fprintf(stderr, "[M::rb3gpu] vm_ensure: %.1f MB could not be mapped behind %.1f MB of a range of %.1f MB\n", (double)inc / 1e6, (double)r->mapped / 1e6, (double)r->va_size / 1e6); (void)hipMemRelease(hd); if (r->mapped == 0) { (void)hipMemAddressFree(r->va, r->va_size); h->vmr.erase(std::find(h->vmr.begin(), h->vmr.end(), r)); delete r; } return RB3GPU_ENOMEM; }
		r->hs.push_back(std::make_pair(hd, inc));
		r->mapped += inc;
		h->bytes_owned += (int64_t)inc;
		if (h->bytes_owned > h->stt.bytes_peak) h->stt.bytes_peak = h->bytes_owned;
		h->stt.n_allocs += 1;
		if (h->tn.log_alloc) fprintf(stderr, "[M::rb3gpu] %.1f MB mapped behind %.1f MB at %p in %.3f ms\n", (double)inc / 1e6, (double)(r->mapped - inc) / 1e6, r->va, (now_s() - t0) * 1e3);
		if (h->tn.poison) HIPCHK(hipMemsetAsync((char*)r->va + r->mapped - inc, 0xA5, inc, h->st));
	}
	h->stt.ms_alloc += (now_s() - t0) * 1e3;
	*p = r->va, *cap = r->mapped;
	return 0;
}

/* give a range back (the device must be done with it: callers synchronise) */
static void vm_release(rb3gpu_t *h, VmRange *r)
{
	size_t off = 0;
	for (auto &c : r->hs) { (void)hipMemUnmap((char*)r->va + off, c.second); (void)hipMemRelease(c.first); off += c.second; }
	(void)hipMemAddressFree(r->va, r->va_size);
	h->bytes_owned -= (int64_t)r->mapped;
	h->vmr.erase(std::find(h->vmr.begin(), h->vmr.end(), r));
	delete r;
}

static size_t grow_slack(size_t n) { return n < ((size_t)256 << 20) ? n >> 1 : n < ((size_t)2 << 30) ? n >> 2 : n >> 3; }

/* growable: the buffer may be a range that grows in place (vm_ensure) -- ONLY buffers that nothing but kernels touch: the runtime's copies and fills look an
 * address up as ONE allocation, and a range is several (a copy across a chunk boundary of the records of the interval-sharded merge lost rows, round 6) */
static int buf_ensure(rb3gpu_t *h, Buf &b, size_t bytes, bool exact = false, bool growable = false)
{
	if (b.cap >= bytes && b.p) return 0;
	if (h->tn.log_alloc) { // which buffer: its place among the handle's
		static const char *names[] = { "b2", "pos", "post", "tcnt", "tpre", "ctot", "ctot2", "gstat", "gpre", "jg", "misc", "xbuf", "wl", "wls", "dl", "dlx", "wstat", "wplane", "wruns", "gslots", "glist", "pslots", "lbst", "shc", "shn", "shs", "shr", "shk" };
		const ptrdiff_t k = &b - &h->b2;
		fprintf(stderr, "[M::rb3gpu] buffer '%s' grows from %.1f to %.1f MB%s\n", k >= 0 && k < 28 ? names[k] : "(local)", (double)b.cap / 1e6, (double)bytes / 1e6, growable ? " (in place)" : "");
	}
	if (growable && vm_usable(h, bytes)) { // grows where it stands: a sixteenth of slack (at most 256 MB) so that not every merge maps a chunk
		const size_t sl = exact ? 0 : (bytes >> 4) < ((size_t)256 << 20) ? (bytes >> 4) : ((size_t)256 << 20);
		if (b.p && !vm_find(h, b.p)) dev_free(h, b.p, b.cap), b.p = nullptr, b.cap = 0; // (it was small so far)
		if (vm_ensure(h, &b.p, &b.cap, bytes + sl + 256) == 0 || vm_ensure(h, &b.p, &b.cap, bytes + 256) == 0) return 0;
		if (b.p && b.cap >= bytes) return 0;
	}
	if (b.p) dev_free(h, b.p, b.cap);
	b.p = nullptr, b.cap = 0;
	// geometric growth: an index that grows round by round must not realloc every round (exact: a one-off, e.g. loading an index, or a table of a fixed size).
	// Half as much again while a buffer is small, a quarter from 256 MB, an eighth from 2 GB: at 360 M rows per batch the halves were 10 GB of nothing.
	size_t want = exact ? bytes + 256 : bytes + grow_slack(bytes) + 256;
	int r = dev_malloc(h, &b.p, want);
	if (r == RB3GPU_ENOMEM && want != bytes) r = dev_malloc(h, &b.p, want = bytes);
	if (r < 0) return r;
	b.cap = want;
	return 0;
}

static void buf_release(rb3gpu_t *h, Buf &b)
{
	if (b.p) dev_free(h, b.p, b.cap);
	b.p = nullptr, b.cap = 0;
}

static float ev_ms(hipEvent_t a, hipEvent_t b)
{
	float ms = 0;
	if (hipEventElapsedTime(&ms, a, b) != hipSuccess) ms = 0;
	return ms;
}

/* table of the tentative stretches (k_chain): one 64-byte record each, followed by the compact array
 * of settled unknowns (sfin, int32 per stretch) that k_resolve fills for k_pos_finalize_check */
static void fill_add(FillJobs *jb, void *p, size_t bytes, uint32_t val)
{
	if (bytes == 0 || jb->n >= 8) return;
	jb->p[jb->n] = p, jb->n16[jb->n] = (bytes + 15) / 16, jb->val[jb->n] = val, ++jb->n; // (every buffer of the handle has slack behind it: rounding up is safe)
}

static void fill_launch(rb3gpu_t *h, const FillJobs &jb)
{
	unsigned long long tot = 0;
	for (int i = 0; i < jb.n; ++i) tot += jb.n16[i];
	if (tot == 0) return;
	unsigned long long nblk = (tot + 1023) / 1024; // four units per thread
	if (nblk > 8192) nblk = 8192;
	hipLaunchKernelGGL(k_fill_regions, dim3((unsigned)nblk), dim3(256), 0, h->st, jb);
}

/* jb != NULL: the regions to clear are added to a job list (one launch with everything else the merge clears) instead of being
 * cleared by memsets of their own */
static int tent_prepare(rb3gpu_t *h, rb3_stretch_t **tab, int32_t **sfin, FillJobs *jb = nullptr)
{
	const size_t bytes = (size_t)RB3_TENT_IDS * (sizeof(rb3_stretch_t) + 4);
	const bool fresh = !(h->dl.p && h->dl.cap >= bytes);
	int r;
	if ((r = buf_ensure(h, h->dl, bytes, true)) < 0) return r; // (a table of a fixed size)
	rb3_stretch_t *t = *tab = (rb3_stretch_t*)h->dl.p;
	int32_t *f = *sfin = (int32_t*)(t + RB3_TENT_IDS);
	// zero what the previous merge used: the blocks at the bottom of the table, the single ids from the middle up
	const int64_t da = fresh ? RB3_TENT_HALF : h->sid_dirty[0], db = fresh ? RB3_TENT_HALF : h->sid_dirty[1];
	if (jb) {
		if (da > 0) fill_add(jb, t, (size_t)da * sizeof(rb3_stretch_t), 0u), fill_add(jb, f, (size_t)da * 4, 0u);
		if (db > 0) fill_add(jb, t + RB3_TENT_HALF, (size_t)db * sizeof(rb3_stretch_t), 0u), fill_add(jb, f + RB3_TENT_HALF, (size_t)db * 4, 0u);
	} else {
		if (da > 0) HIPCHK(hipMemsetAsync(t, 0, (size_t)da * sizeof(rb3_stretch_t), h->st));
		if (da > 0) HIPCHK(hipMemsetAsync(f, 0, (size_t)da * 4, h->st));
		if (db > 0) HIPCHK(hipMemsetAsync(t + RB3_TENT_HALF, 0, (size_t)db * sizeof(rb3_stretch_t), h->st));
		if (db > 0) HIPCHK(hipMemsetAsync(f + RB3_TENT_HALF, 0, (size_t)db * 4, h->st));
	}
	h->sid_dirty[0] = h->sid_dirty[1] = RB3_TENT_HALF; // until the number of stretches this merge opens has been read back
	return 0;
}

static void tent_used(rb3gpu_t *h, unsigned long long sidctr) // the two 32-bit counters of misc[5]
{
	const unsigned long long a = sidctr & 0xFFFFFFFFull, b = sidctr >> 32;
	h->sid_dirty[0] = a < (unsigned long long)RB3_TENT_HALF ? (int64_t)a : RB3_TENT_HALF;
	h->sid_dirty[1] = b < (unsigned long long)RB3_TENT_HALF ? (int64_t)b : RB3_TENT_HALF;
}

/* masks of 256 q bits for the stretches of the lower half of the table (q > 1: more than 255 matching suffixes per interval) */
static int tent_masks(rb3gpu_t *h, int q, uint32_t **mx)
{
	*mx = nullptr;
	if (q <= 1) return 0;
	int r;
	if ((r = buf_ensure(h, h->dlx, (size_t)RB3_TENT_HALF * 32 * (size_t)q)) < 0) return r;
	*mx = (uint32_t*)h->dlx.p;
	return 0;
}

/* the settle kernels behind k_chain: the rows that dropped at every event, per walker the cumulative mask, the paths over first
 * stretches (up to maxhops hops), all other stretches */
static void launch_settle(rb3gpu_t *h, const IdxView &iv, rb3_stretch_t *tab, uint32_t *mx, const uint32_t *sidctr, int32_t *sfin, unsigned long long *bad, int maxhops, int q, const uint32_t *mctr = nullptr)
{
	// mctr: the extent of the stretch ids in use has not been worked out yet (k_tent_extent); the first kernel of the narrow-mask settle does it on its way
	const bool fused_extent = mctr != nullptr && !(q >= 2 && mx) && !h->tn.resolve_v1;
	if (mctr != nullptr && !fused_extent) hipLaunchKernelGGL(k_tent_extent, dim3(1), dim3(64), 0, h->st, mctr, (uint32_t*)sidctr);
#define RB3_SETTLE_X(Q) do { \
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_events_x<Q>), dim3(2048), dim3(256), 0, h->st, iv, (const rb3_stretch_t*)tab, mx, sidctr); \
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cum_x<Q>), dim3(4096), dim3(64), 0, h->st, tab, mx, sidctr); \
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_resolve_w_x<Q>), dim3(512), dim3(256), 0, h->st, tab, (const uint32_t*)mx, sidctr, sfin, bad, maxhops); \
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sfin_x<Q>), dim3(2048), dim3(256), 0, h->st, (const rb3_stretch_t*)tab, (const uint32_t*)mx, sidctr, sfin); \
	} while (0)
	if (q >= 8 && mx) RB3_SETTLE_X(8);
	else if (q >= 4 && mx) RB3_SETTLE_X(4);
	else if (q >= 2 && mx) RB3_SETTLE_X(2);
	else {
		hipLaunchKernelGGL(k_events, dim3((unsigned)h->tn.ev_blocks), dim3(256), 0, h->st, iv, tab, sidctr, fused_extent ? mctr : (const uint32_t*)nullptr, (uint32_t*)sidctr);
		if (h->tn.resolve_v1) hipLaunchKernelGGL(k_resolve, dim3(2048), dim3(256), 0, h->st, tab, sidctr, sfin);
		else {
			hipLaunchKernelGGL(k_cum, dim3((unsigned)h->tn.cum_blocks), dim3(256), 0, h->st, tab, sidctr);
			hipLaunchKernelGGL(k_resolve_w, dim3((unsigned)h->tn.resw_blocks), dim3(256), 0, h->st, tab, sidctr, sfin, bad, maxhops);
			hipLaunchKernelGGL(k_sfin, dim3((unsigned)h->tn.sfin_blocks), dim3(256), 0, h->st, (const rb3_stretch_t*)tab, sidctr, sfin);
		}
	}
#undef RB3_SETTLE_X
}

#define RB3_GRP_ALLOC (sizeof(rb3_grp_t) + 8) /* bytes per directory entry of an index buffer: the entry + its word of the compact copy */

static IdxView view_of(const rb3gpu_t *h)
{
	IdxView v;
	v.grp64 = (const uint64_t*)h->grp, v.slot16 = (const uint4*)h->slots, v.n = h->n, v.m = h->acc[1];
	v.gsm = h->grp ? (const uint64_t*)(h->ib[h->cur].grp + h->ib[h->cur].grp_cap) : nullptr;
	const int64_t nwin = (h->n >> RB3_WIN_BITS) + 1;
	v.abs = RB3_ABS_HEADERS(h->n, h->tn.abs_limit) ? (h->n >= RB3_ABS_LIMIT || h->tn.abs_table ? 2 : 1) : 0;
	v.sb = h->ib[h->cur].sbt;
	v.dense = h->nslots == nwin ? (v.abs ? 2 : 1) : 0;
	return v;
}

extern "C" {

void rb3gpu_opt_init(rb3gpu_opt_t *opt)
{
	memset(opt, 0, sizeof(*opt));
	opt->device = 0, opt->split_log2 = 0, opt->verbose = 1;
}

const char *rb3gpu_strerror(int err)
{
	switch (err) {
	case RB3GPU_OK: return "success";
	case RB3GPU_ENODEV: return "HIP device or runtime error";
	case RB3GPU_ENOMEM: return "out of memory";
	case RB3GPU_EINVAL: return "invalid argument";
	case RB3GPU_ESYMBOL: return "BWT symbol outside 0..5";
	case RB3GPU_ESTATE: return "operation not valid in this state";
	case RB3GPU_EINTERNAL: return "device-side invariant violated";
	case RB3GPU_EUNSUP: return "not supported for this index by this call";
	default: return "unknown error";
	}
}

int rb3gpu_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return RB3GPU_ENODEV;
	return n;
}


static int tune_set(rb3gpu_t *h, const char *key, int64_t v)
{
	Tune &t = h->tn;
	if (!strcmp(key, "tent")) t.tent = v != 0;
	else if (!strcmp(key, "staged")) t.staged = v != 0;
	else if (!strcmp(key, "group_rebuild")) t.group_rebuild = v != 0;
	else if (!strcmp(key, "window_rebuild")) t.window_rebuild = v != 0;
	else if (!strcmp(key, "plane_rebuild")) t.plane_rebuild = v != 0;
	else if (!strcmp(key, "reb_t1_rows")) t.reb_t1_rows = (int)v;
	else if (!strcmp(key, "part")) t.part = (int)v;
	else if (!strcmp(key, "scan_place")) t.scan_place = v != 0;
	else if (!strcmp(key, "reb_force")) t.reb_force = v != 0;
	else if (!strcmp(key, "resolve_v1")) t.resolve_v1 = v != 0;
	else if (!strcmp(key, "octs")) t.octs = v < 1 ? 1 : v > 8 ? 8 : (int)v;
	else if (!strcmp(key, "lpw")) t.lpw = 8;
	else if (!strcmp(key, "blkmul")) t.blkmul = v < 1 ? 1 : (int)v;
	else if (!strcmp(key, "blkcap")) t.blkcap = v < 1 ? 1 : v;
	else if (!strcmp(key, "chain_bs")) t.chain_bs = v == 64 ? 64 : v == 128 ? 128 : 256;
	else if (!strcmp(key, "lf_after")) t.lf_after = v != 0;
	else if (!strcmp(key, "vmm")) t.vmm = v < 0 ? 0 : (int)v;
	else if (!strcmp(key, "abs_table")) t.abs_table = v != 0;
	else if (!strcmp(key, "b2_tw")) t.b2_tw = v != 0;
	else if (!strcmp(key, "vmm_reserve")) t.vmm_reserve = v < 0 ? 0 : (int)v;
	else if (!strcmp(key, "copy_walkers")) t.copy_walkers = v != 0;
	else if (!strcmp(key, "trec")) t.trec = v < 0 ? -1 : v != 0;
	else if (!strcmp(key, "abs_limit")) {
		if (h->grp) return RB3GPU_ESTATE; // the headers of the index in place were written under the old limit
		t.abs_limit = v < 0 ? INT64_MAX : v;
	} else if (!strcmp(key, "tent_q")) t.tent_q = v >= 8 ? 8 : v >= 4 ? 4 : v >= 2 ? 2 : v >= 1 ? 1 : 0;
	else if (!strcmp(key, "ssa_split")) t.ssa_split = v < 4 ? 4 : v > 20 ? 20 : (int)v;
	else if (!strcmp(key, "b2_split")) t.b2_split = v < 0 ? -1 : v > 12 ? 12 : (int)v;
	else if (!strcmp(key, "log_alloc")) t.log_alloc = v != 0;
	else if (!strcmp(key, "defer_free")) t.defer_free = v != 0;
	else if (!strcmp(key, "poison")) t.poison = v != 0;
	else if (!strcmp(key, "guard")) t.guard = v != 0;
	else if (!strcmp(key, "load_chunk")) t.load_chunk = v < 1 ? 1 : v;
	else if (!strcmp(key, "lf_check")) t.lf_check = v < 0 ? 0 : v > (1 << 30) ? (1 << 30) : (int)v;
	else if (!strcmp(key, "fmd_piece")) t.fmd_piece = v < 0 ? 0 : v;
	else if (!strcmp(key, "sh_host_rounds")) t.sh_host_rounds = v < 0 ? -1 : v != 0; // (-1: rounds on the device whatever the number of chains)
	else if (!strcmp(key, "ev_blocks")) t.ev_blocks = v < 1 ? 1 : v > 65536 ? 65536 : (int)v;
	else if (!strcmp(key, "cum_blocks")) t.cum_blocks = v < 1 ? 1 : v > 65536 ? 65536 : (int)v;
	else if (!strcmp(key, "resw_blocks")) t.resw_blocks = v < 1 ? 1 : v > 65536 ? 65536 : (int)v;
	else if (!strcmp(key, "sfin_blocks")) t.sfin_blocks = v < 1 ? 1 : v > 65536 ? 65536 : (int)v;
	else if (!strcmp(key, "sh_block")) t.sh_block = v >= 1024 ? 1024 : v > 0 ? 256 : 0;
	else if (!strcmp(key, "sh_states")) t.sh_states = v >= 8 ? 8 : v >= 4 ? 4 : v >= 2 ? 2 : v >= 1 ? 1 : 0;
	else if (!strcmp(key, "junction_check")) t.junction_check = v < 0 ? 0 : v > 4096 ? 4096 : (int)v;
	else if (!strcmp(key, "corrupt_sfin") || !strcmp(key, "force_fallback") || !strcmp(key, "hide_first") || !strcmp(key, "tent_limit") || !strcmp(key, "text_mode") || !strcmp(key, "corrupt_pos") || !strcmp(key, "reb_lcap") || !strcmp(key, "reb_slot_cap") ||
			!strcmp(key, "pos_limit") || !strcmp(key, "win_scratch") || !strcmp(key, "slot_bytes")) {
#ifdef RB3GPU_TEST_HOOKS
		if (!strcmp(key, "pos_limit")) t.pos_limit = v;
		else if (!strcmp(key, "win_scratch")) t.win_scratch = v;
		else if (!strcmp(key, "slot_bytes")) t.slot_bytes = v;
		else if (!strcmp(key, "corrupt_pos")) t.corrupt_pos = v != 0;
		else if (!strcmp(key, "corrupt_sfin")) t.corrupt_sfin = v;
		else if (!strcmp(key, "reb_lcap")) t.reb_lcap = v;
		else if (!strcmp(key, "reb_slot_cap")) t.reb_slot_cap = v;
		else if (!strcmp(key, "force_fallback")) t.force_fallback = v != 0;
		else if (!strcmp(key, "hide_first")) t.hide_first = v != 0;
		else if (!strcmp(key, "tent_limit")) t.tent_limit = v;
		else t.text_mode = v == 1 || v == 2 ? (int)v : 0;
#else
		return RB3GPU_EUNSUP; // test hooks are compiled out of the release library
#endif
	} else return RB3GPU_EINVAL;
	return 0;
}

int rb3gpu_tune(rb3gpu_t *h, const char *key, int64_t value)
{
	if (!h || !key) return RB3GPU_EINVAL;
	return tune_set(h, key, value);
}

static void tune_from_env(rb3gpu_t *h) // once per handle
{
	static const char *keys[] = { "tent", "staged", "group_rebuild", "window_rebuild", "plane_rebuild", "reb_t1_rows", "part", "scan_place", "reb_force", "resolve_v1", "octs", "lpw", "blkmul", "blkcap", "chain_bs", "lf_after", "copy_walkers", "tent_q", "trec", "abs_limit", "abs_table", "ssa_split", "b2_split", "lf_check", "junction_check", "sh_block", "sh_states", "ev_blocks", "cum_blocks", "resw_blocks", "sfin_blocks", "sh_host_rounds", "fmd_piece", "load_chunk", "log_alloc", "defer_free", "vmm", "vmm_reserve", "b2_tw", "poison", "guard",
		"force_fallback", "hide_first", "tent_limit", "text_mode", "corrupt_pos", "corrupt_sfin", "reb_lcap", "reb_slot_cap", "pos_limit", "win_scratch", "slot_bytes", nullptr };
	for (int i = 0; keys[i]; ++i) {
		char name[64] = "RB3GPU_";
		size_t l = strlen(name);
		for (const char *p = keys[i]; *p && l + 1 < sizeof(name); ++p) name[l++] = (char)(*p >= 'a' && *p <= 'z' ? *p - 32 : *p);
		name[l] = 0;
		const char *v = getenv(name);
		if (v && *v) (void)tune_set(h, keys[i], atoll(v));
	}
}

rb3gpu_t *rb3gpu_create(const rb3gpu_opt_t *opt)
{
	rb3gpu_opt_t o;
	int ndev = 0;
	if (opt) o = *opt; else rb3gpu_opt_init(&o);
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || o.device < 0 || o.device >= ndev) {
		if (o.verbose >= 1) fprintf(stderr, "[E::rb3gpu_create] no usable HIP device (count=%d, requested=%d); this engine has no CPU fallback\n", ndev, o.device);
		return nullptr;
	}
	if (hipSetDevice(o.device) != hipSuccess) return nullptr;
	rb3gpu_t *h = new (std::nothrow) rb3gpu_s();
	if (!h) return nullptr;
	h->dev = o.device, h->opt = o;
	{ size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) == hipSuccess) h->dev_mem = tot; else (void)hipGetLastError(); }
	tune_from_env(h);
	memset(&h->stt, 0, sizeof(h->stt));
	if (hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) != hipSuccess) { delete h; return nullptr; }
	if (hipStreamCreateWithFlags(&h->st2, hipStreamNonBlocking) != hipSuccess) { delete h; return nullptr; }
	for (int i = 0; i < 8; ++i)
		if (hipEventCreate(&h->ev[i]) != hipSuccess) { delete h; return nullptr; }
	for (int i = 0; i < 3; ++i)
		if (hipEventCreateWithFlags(&h->evx[i], hipEventDisableTiming) != hipSuccess) { delete h; return nullptr; }
	if (hipHostMalloc((void**)&h->hm_pin, 128 * 8, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); h->hm_pin = nullptr; }
	h->t0 = now_s();
	return h;
}

static void index_drop(rb3gpu_t *h)
{
	h->grp = nullptr, h->slots = nullptr, h->n = h->ngrp = h->nslots = 0;
	memset(h->acc, 0, sizeof(h->acc));
	h->stt.bytes_index = 0;
}

static void ib_release(rb3gpu_t *h, int i)
{
	dev_free(h, h->ib[i].grp, h->ib[i].grp_cap * RB3_GRP_ALLOC);
	dev_free(h, h->ib[i].slots, h->ib[i].slots_cap * sizeof(rb3_slot_t));
	dev_free(h, h->ib[i].sbt, h->ib[i].sbt_cap * 64);
	h->ib[i].grp = nullptr, h->ib[i].slots = nullptr, h->ib[i].sbt = nullptr, h->ib[i].grp_cap = h->ib[i].slots_cap = h->ib[i].sbt_cap = 0;
}

/* make ib[i] hold at least ngrp directory entries and nslots slots (contents are not preserved) */
static int ib_ensure(rb3gpu_t *h, int i, int64_t ngrp, int64_t nslots, bool exact = false)
{
	int r;
	{ // the table of LF bases every 2^31 symbols (IdxView.sb; a few hundred bytes): room for it beside every index, used from 2^32 symbols on
		const size_t nsb = ((size_t)ngrp >> (RB3_SB_BITS - RB3_GRP_BITS)) + 4;
		if (h->ib[i].sbt_cap < nsb) {
			dev_free(h, h->ib[i].sbt, h->ib[i].sbt_cap * 64);
			h->ib[i].sbt = nullptr, h->ib[i].sbt_cap = 0;
			if ((r = dev_malloc(h, (void**)&h->ib[i].sbt, (nsb + 12) * 64)) < 0) return r;
			h->ib[i].sbt_cap = nsb + 12;
		}
	}
	// (the slot array grows in place; the directory -- 72 bytes per 8192 symbols, read back by the host in places -- stays an ordinary allocation)
	if (h->ib[i].slots_cap < (size_t)nslots && vm_usable(h, (size_t)nslots * sizeof(rb3_slot_t))) {
		void *p = h->ib[i].slots;
		size_t cap = 0;
		if (p && !vm_find(h, p)) dev_free(h, p, h->ib[i].slots_cap * sizeof(rb3_slot_t)), p = nullptr, h->ib[i].slots = nullptr, h->ib[i].slots_cap = 0;
		const size_t slk = exact ? 0 : ((size_t)nslots >> 4) < ((size_t)2 << 20) ? ((size_t)nslots >> 4) : ((size_t)2 << 20); // (a sixteenth, at most 256 MB)
		if (vm_ensure(h, &p, &cap, ((size_t)nslots + slk + 64) * sizeof(rb3_slot_t)) == 0) h->ib[i].slots = (rb3_slot_t*)p, h->ib[i].slots_cap = cap / sizeof(rb3_slot_t);
	}
	if (h->ib[i].grp_cap < (size_t)ngrp) {
		dev_free(h, h->ib[i].grp, h->ib[i].grp_cap * RB3_GRP_ALLOC);
		h->ib[i].grp = nullptr, h->ib[i].grp_cap = 0;
		size_t want = exact ? (size_t)ngrp + 16 : (size_t)ngrp + (size_t)(ngrp >> 1) + 16;
		if ((r = dev_malloc(h, (void**)&h->ib[i].grp, want * RB3_GRP_ALLOC)) < 0) {
			want = (size_t)ngrp + 4; // (spare entries: k_chain asks for the slot words of a group and of the one behind it)
			if ((r = dev_malloc(h, (void**)&h->ib[i].grp, want * RB3_GRP_ALLOC)) < 0) return r;
		}
		h->ib[i].grp_cap = want;
	}
	if (h->ib[i].slots_cap < (size_t)nslots) {
		dev_free(h, h->ib[i].slots, h->ib[i].slots_cap * sizeof(rb3_slot_t));
		h->ib[i].slots = nullptr, h->ib[i].slots_cap = 0;
		size_t want = exact ? (size_t)nslots + 64 : (size_t)nslots + grow_slack((size_t)nslots * sizeof(rb3_slot_t)) / sizeof(rb3_slot_t) + 64;
		if ((r = dev_malloc(h, (void**)&h->ib[i].slots, want * sizeof(rb3_slot_t))) < 0) {
			want = (size_t)nslots;
			if ((r = dev_malloc(h, (void**)&h->ib[i].slots, want * sizeof(rb3_slot_t))) < 0) return r;
		}
		h->ib[i].slots_cap = want;
	}
	return 0;
}

/* tune "guard": have the kernels written behind any buffer of the handle?  (call with the stream idle) */
static void guard_check(rb3gpu_t *h, const char *where)
{
	if (!h->tn.guard) return;
	static const char *names[] = { "b2", "pos", "post", "tcnt", "tpre", "ctot", "ctot2", "gstat", "gpre", "jg", "misc", "xbuf", "wl", "dl", "dlx", "wstat", "wplane", "wruns", "gslots", "glist" };
	Buf *all[] = { &h->b2, &h->pos, &h->post, &h->tcnt, &h->tpre, &h->ctot, &h->ctot2, &h->gstat, &h->gpre, &h->jg, &h->misc, &h->xbuf, &h->wl, &h->wls, &h->dl, &h->dlx, &h->wstat, &h->wplane, &h->wruns, &h->gslots, &h->glist, &h->pslots, &h->lbst, &h->shc, &h->shn, &h->shs, &h->shr, &h->shk };
	uint8_t g[RB3_GUARD];
	for (int i = 0; i < 20 + 4; ++i) {
		const uint8_t *p = nullptr; size_t cap = 0; const char *name = "";
		if (i < 20) p = (const uint8_t*)all[i]->p, cap = all[i]->cap, name = names[i];
		else if (i < 22) p = (const uint8_t*)h->ib[i - 20].grp, cap = h->ib[i - 20].grp_cap * RB3_GRP_ALLOC, name = "index directory";
		else p = (const uint8_t*)h->ib[i - 22].slots, cap = h->ib[i - 22].slots_cap * sizeof(rb3_slot_t), name = "index slots";
		if (!p) continue;
		if (cap == 0) cap = 256;
		if (hipMemcpy(g, p + cap, RB3_GUARD, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); continue; }
		for (int k = 0; k < RB3_GUARD; ++k)
			if (g[k] != 0xA5) {
				int nbad = 0, last = k;
				for (int q = k; q < RB3_GUARD; ++q) if (g[q] != 0xA5) ++nbad, last = q;
				fprintf(stderr, "[E::rb3gpu] %s: buffer '%s' (%zu bytes at %p) was written behind its end: %d bytes differ, offsets %d..%d:", where, name, cap, (const void*)p, nbad, k, last);
				for (int q = k; q < k + 32 && q < RB3_GUARD; ++q) fprintf(stderr, " %02x", g[q]);
				fprintf(stderr, "\n");
				break;
			}
	}
}

void rb3gpu_destroy(rb3gpu_t *h)
{
	if (!h) return;
	(void)hipSetDevice(h->dev);
	(void)hipStreamSynchronize(h->st);
#ifdef RB3_DEBUG_CUM
	{
		unsigned long long c[8];
		if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_cum_dbg), sizeof(c)) == hipSuccess)
			fprintf(stderr, "[debug cum] walkers with events %llu, of them first stretch not settled by a walker %llu, a later stretch settled %llu (unambiguous %llu, ambiguous %llu)\n", c[0], c[1], c[2], c[3], c[4]);
	}
#endif
#ifdef RB3_PROF_REB
	{
		unsigned long long p[16];
		if (hipMemcpyFromSymbol(p, HIP_SYMBOL(g_reb_prof), sizeof(p)) == hipSuccess && p[8])
			fprintf(stderr, "[prof] k_reb_group: %llu groups done (rows %.1f, runs %.1f, slots %.2f per group); cycles per group: setup %.0f, runs %.0f, rows-load %.0f, rows %.0f, run items %.0f, heads %.0f, partition %.0f, slots %.0f\n",
					p[8], (double)p[9] / p[8], (double)p[10] / p[8], (double)p[11] / p[8], (double)p[0] / p[8], (double)p[1] / p[8], (double)p[2] / p[8], (double)p[3] / p[8], (double)p[4] / p[8], (double)p[5] / p[8], (double)p[6] / p[8], (double)p[7] / p[8]);
		unsigned long long w[8];
		if (hipMemcpyFromSymbol(w, HIP_SYMBOL(g_reb_why), sizeof(w)) == hipSuccess)
			fprintf(stderr, "[prof] k_reb_group tiers handed on (counted per tier): too many rows %llu, old slots %llu, bit-plane slot %llu, old runs %llu, row runs %llu, new slots %llu, last group %llu\n", w[0], w[1], w[2], w[3], w[4], w[5], w[6]);
	}
#endif
	index_drop(h);
	ib_release(h, 0), ib_release(h, 1);
	Buf *all[] = { &h->b2, &h->pos, &h->post, &h->tcnt, &h->tpre, &h->ctot, &h->ctot2, &h->gstat, &h->gpre, &h->jg, &h->misc, &h->xbuf, &h->wl, &h->wls, &h->dl, &h->dlx, &h->wstat, &h->wplane, &h->wruns, &h->gslots, &h->glist, &h->pslots, &h->lbst, &h->shc, &h->shn, &h->shs, &h->shr, &h->shk, &h->twb, &h->shp0, &h->shp1 };
	for (Buf *b : all) buf_release(h, *b);
	garbage_collect(h, true);
	for (int i = 0; i < 8; ++i) (void)hipEventDestroy(h->ev[i]);
	for (int i = 0; i < 3; ++i) (void)hipEventDestroy(h->evx[i]);
	if (h->hm_pin) (void)hipHostFree(h->hm_pin);
	(void)hipStreamDestroy(h->st2);
	for (int i = 0; i < 2; ++i) if (h->stage[i]) (void)hipHostFree(h->stage[i]);
	rb3sort_destroy(h->sorter);
	(void)hipStreamDestroy(h->st);
	delete h;
}

/* exclusive scan of nrec records of 8 x u32 -> 8 x u64 (7 columns used).  The 8 totals are left in
 * device memory at `dtot_keep`; if `total` is not NULL they are also copied to the host (one sync). */
/* side: on the handle's side stream, with chunk totals of its own (the batch's scan beside the walkers, see merge_core) */
static int scan_records(rb3gpu_t *h, const uint32_t *in, int64_t nrec, uint64_t *out, uint64_t *dtot_keep, uint64_t total[8], bool side = false)
{
	const int64_t nchunk = (nrec + RB3_SCAN_CHUNK - 1) / RB3_SCAN_CHUNK;
	Buf &cb = side ? h->ctot2 : h->ctot;
	hipStream_t st = side ? h->st2 : h->st;
	int r;
	if ((r = buf_ensure(h, cb, (size_t)(nchunk + 1) * 64)) < 0) return r;
	uint64_t *ctot = (uint64_t*)cb.p;
	hipLaunchKernelGGL(k_scan_chunk_totals, dim3((unsigned)nchunk), dim3(256), 0, st, in, nrec, ctot);
	hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(256), 0, st, ctot, nchunk, dtot_keep);
	hipLaunchKernelGGL(k_scan_records, dim3((unsigned)nchunk), dim3(256), 0, st, in, nrec, (const uint64_t*)ctot, out);
	if (total) {
		HIPCHK(hipMemcpyAsync(total, dtot_keep, 64, hipMemcpyDeviceToHost, st));
		HIPCHK(hipStreamSynchronize(st));
	}
	return 0;
}

/* layout of the 64 x u64 words of h->misc (zeroed per merge up to MISC_KEEP):
 *   [0] walker queue head  [1] LF steps  [2] rows unset  [3] rows out of order  [4] tentative unsettled
 *   [5] tentative stretches opened (u32)
 *   [16..23] totals of the batch scan (symbol counts of B2, [22] = bad bytes)
 *   [24..31] totals of the rebuild scan (symbol counts of the merged BWT, [30] = slots)
 *   [14] (two u32) lengths of the two lists of groups the run-space rebuild hands on: after the small tier, after the large tier */
#define MISC_WORDS   (64 + 32 * RB3_TENT_NCTR / 2) /* the counters and totals of a merge, then (MISC_MCTR) the id counters of the tentative stretches, one per 128 bytes */
#define MISC_MCTR    64
#define MISC_LF_TOT  16
#define MISC_IX_TOT  24
#define MISC_RG_LISTS 14
#define MISC_LF_CHK   6    /* [6] rows whose LF relation was verified, [7] rows that failed it */
#define MISC_B2_MODE  15   /* what the device-made walker list is (k_b2_mode) */
#define MISC_WIDE     39   /* k_chain: steps of walkers that could not record tentatively because their interval is wider than the masks */
#define MISC_RG_OVER  32   /* set by k_decide<LISTED> / k_plane_group<LISTED>: the hand-over list of the run-space rebuild is longer than the scratch of the symbol path */
#define MISC_PP_OK    33   /* k_plane_group: groups whose old range holds no bit-plane slot (the run-space rebuild could take them) */
#define MISC_BAD_WALKERS RB3_MISC_BADW /* k_chain: entries of the caller's walker list that are not walkers of this batch (row outside it, no steps) */

/* build a block array for ntot symbols into ib[1-cur]; FROM_PLAIN: symbols are d_b2[0..ntot);
 * otherwise the interleave of the current index with d_b2 at merged positions pos[].
 * nosync: size the slot array by its upper bound (one slot per window) and do not wait for the
 * scan totals; the caller reads them from misc[MISC_IX_TOT..] after its own sync. */
/* does the run-space rebuild (k_reb_group) run for a merge of n2 rows into the current index?  (the statistics ask too) */
static bool runspace_applies(const rb3gpu_t *h, int64_t n2, int64_t ntot)
{
	const int64_t nwin_old = (h->n >> RB3_WIN_BITS) + 1;
	if (h->tn.window_rebuild || h->nslots == nwin_old) return false; // (a fully bit-plane index has no run slot to start from)
	if (h->tn.reb_force) return true;
	if (h->reb_pp_all && h->tn.plane_rebuild) return false; // (most groups still hold a bit-plane slot: every one of them would be handed on)
	return h->nslots * 2 < nwin_old && (double)n2 * RB3_GRP <= 4096.0 * (double)ntot; // (run-coded index; not where half of every group is new)
}

/* the size limits of the single-synchronisation merge; the test build of the library can shrink them so that a test crosses them */
static int64_t lim_pos(const rb3gpu_t *h)
{
#ifdef RB3GPU_TEST_HOOKS
	if (h->tn.pos_limit > 0 && h->tn.pos_limit < (1LL << RB3_TENT_PBITS)) return h->tn.pos_limit;
#endif
	(void)h;
	return 1LL << RB3_TENT_PBITS;
}
static size_t lim_win_scratch(const rb3gpu_t *h)
{
#ifdef RB3GPU_TEST_HOOKS
	if (h->tn.win_scratch > 0) return (size_t)h->tn.win_scratch;
#endif
	// (8 GB until round 5: an index of 9.5 G symbols fell off it onto the group-sequential kernels -- 12 -> 58 ms per rebuild, 120 ms at 24 G symbols,
	// tools/r6/scale_hap.sh.  A quarter of the device's memory: 72 GB of 288)
	return h->dev_mem / 4 > ((size_t)8 << 30) ? h->dev_mem / 4 : (size_t)8 << 30;
}
static size_t lim_slot_bytes(const rb3gpu_t *h)
{
#ifdef RB3GPU_TEST_HOOKS
	if (h->tn.slot_bytes > 0) return (size_t)h->tn.slot_bytes;
#endif
	return h->dev_mem / 3 > ((size_t)16 << 30) ? h->dev_mem / 3 : (size_t)16 << 30; // (16 GB until round 5; a third of the device's memory)
}

static bool use_winpar(const rb3gpu_t *h, int64_t nwin)
{
	// (scratch per window: 128 B of slots in plane space, 216 B for the window kernels of rounds 1-3)
	return (size_t)nwin * (h->tn.plane_rebuild && !h->tn.window_rebuild ? 136 : 216) <= lim_win_scratch(h) && !h->tn.group_rebuild;
}

/* slots to make room for before a single-sync merge of n2 rows: one per window is the upper bound; where the index is
 * run-coded (a fifth of that or less in a pangenome) the old count plus a margin stands in, and the rebuild is emitted again
 * in the rare case that it does not fit (see build_index) */
static int64_t slot_estimate(const rb3gpu_t *h, int64_t n2, int64_t ntot)
{
	const int64_t nwin = (ntot >> RB3_WIN_BITS) + 1;
	if (!use_winpar(h, nwin) || h->nslots * 2 >= (h->n >> RB3_WIN_BITS) + 1) return nwin;
	// (round 5: a quarter of the old count + a sixteenth of the rows until the index holds 2^22 slots = 512 MB; beyond, an eighth + a thirty-second --
	// the 24-haplotype build adds n2 / 750 slots per round, and the two slot arrays of its 1.5 GB index had grown to 6.5 GB each)
	const int64_t est = h->nslots < (1 << 22) ? h->nslots + h->nslots / 4 + n2 / 16 + 4096 : h->nslots + h->nslots / 8 + n2 / 32 + 4096;
	return est < nwin ? est : nwin;
}

extern "C++" {
/* rows_done: jg[] (rows before every window) has been filled by k_pos_finalize_check_rows already */
template<bool FROM_PLAIN>
static int build_index(rb3gpu_t *h, int64_t n2, const uint8_t *d_b2, const int64_t *d_pos, int64_t ntot, bool nosync,
		int64_t *ongrp, int64_t *onslots, int64_t oacc[7], bool rows_done = false, bool full = false)
{
	const int64_t ngrp = (ntot >> RB3_GRP_BITS) + 1, nwin = (ntot >> RB3_WIN_BITS) + 1;
	const int dst = 1 - h->cur;
	const bool prepared = h->reb_prepared; // (only the first rebuild of the single-sync merge that set it)
	h->reb_prepared = false;
	int r;
	if (ngrp > 0x7fffffffLL) return RB3GPU_EINVAL;
	// single-sync merge: the rank phase's validation counters (misc[2..4]) are still on the device; the kernels look
	const unsigned long long *skip = (nosync && !FROM_PLAIN && h->misc.p) ? (const unsigned long long*)h->misc.p + 2 : nullptr;
	if ((r = buf_ensure(h, h->gstat, (size_t)ngrp * 32)) < 0) return r;
	if ((r = buf_ensure(h, h->gpre, (size_t)ngrp * 64)) < 0) return r;
	if ((r = buf_ensure(h, h->misc, MISC_WORDS * 8)) < 0) return r;
	int64_t *jg = nullptr;
	if (!FROM_PLAIN) {
		if ((r = buf_ensure(h, h->jg, (size_t)(ngrp + 1) * 8)) < 0) return r;
		jg = (int64_t*)h->jg.p;
	}
	// window-parallel kernels (one wave per 256-symbol window, planes cached between the passes) unless
	// their scratch (216 B per window) would be unreasonably large; then one wave per 8192-symbol group
	const bool winpar = use_winpar(h, nwin);
	const double rows_per_group = (double)n2 * RB3_GRP / (double)(ntot > 0 ? ntot : 1);
	const bool runspace = runspace_applies(h, n2, ntot) && winpar && !FROM_PLAIN;
	// Scratch of the window kernels (208 B per window).  Behind the run-space rebuild they only see the groups it hands on, and
	// index the scratch by the place in that list: room for an eighth of the groups (all of them while the index is small) --
	// device memory costs ~37 us per MB to obtain, and the buffers of a growing index are obtained over and over.  A longer
	// list is noticed on the device (k_decide) and the rebuild is done again with full = true.
	int64_t lcap = ngrp;
	if (runspace && nosync && !full) {
		lcap = ngrp < 32768 ? ngrp : 32768;
		if (lcap < ngrp / 8) lcap = ngrp / 8;
		// (and room for one and a half times what the merge before handed on: a build whose batches put hundreds of rows into every
		// group -- haplotypes of 360 M symbols into an index of a few G -- hands on a fifth of its groups round after round)
		if (h->reb_last[1] > 0 && lcap < h->reb_last[1] + h->reb_last[1] / 2 + 1024) lcap = h->reb_last[1] + h->reb_last[1] / 2 + 1024;
		if (lcap > ngrp) lcap = ngrp;
#ifdef RB3GPU_TEST_HOOKS
		if (h->tn.reb_lcap > 0 && h->tn.reb_lcap < lcap) lcap = h->tn.reb_lcap;
#endif
	}
	// symbols in plane space (k_plane_group: a block per group, a lane per 32 symbols) instead of a wave per window; the first
	// batch (FROM_PLAIN) keeps the window kernels
	const bool planes = winpar && !FROM_PLAIN && h->tn.plane_rebuild && !h->tn.window_rebuild;
	if (winpar) {
		const int64_t nws = runspace ? lcap * RB3_GRP_WINS : nwin;
		if (planes) {
			if ((r = buf_ensure(h, h->pslots, (size_t)(runspace ? lcap : ngrp) * RB3_GRP_WINS * sizeof(rb3_slot_t), false, true)) < 0) return r;
		} else {
			if ((r = buf_ensure(h, h->wstat, (size_t)nws * 16, false, true)) < 0) return r;
			if ((r = buf_ensure(h, h->wplane, (size_t)nws * 96, false, true)) < 0) return r;
			if ((r = buf_ensure(h, h->wruns, (size_t)nws * RB3_RLE_CODES * 2, false, true)) < 0) return r;
		}
		if (!FROM_PLAIN && (r = buf_ensure(h, h->jg, (size_t)(nwin + 1) * 8)) < 0) return r;
		jg = (int64_t*)h->jg.p;
	}
	// run-space rebuild per group (k_reb_group) wherever the old index is run-coded; what it leaves over goes through the
	// window kernels.  A fully bit-plane index (old.dense) has nothing for it.
	// (not where a group receives more rows than the tables of the run-space kernel take, and not where the old index is
	// mostly bit planes -- few of its groups would qualify and every one that does not costs a hand-over)
	uint32_t *glist[2] = {nullptr, nullptr}, *nglist = nullptr, *gpos = nullptr;
	uint8_t *gkind = nullptr;
	if (runspace) {
		if ((r = buf_ensure(h, h->gslots, (size_t)ngrp * RB3_RG_MAXSLOTS * sizeof(rb3_slot_t), false, true)) < 0) return r;
		if ((r = buf_ensure(h, h->glist, (size_t)ngrp * 13 + 64)) < 0) return r;
		glist[0] = (uint32_t*)h->glist.p, glist[1] = glist[0] + ngrp, gpos = glist[1] + ngrp, gkind = (uint8_t*)(gpos + ngrp);
		nglist = (uint32_t*)((uint64_t*)h->misc.p + MISC_RG_LISTS);
	}
	// scan + placement + compact directory copy in ONE kernel (k_scan_place) where the slots come from the scratch arrays of the group kernels
	const bool fused = planes && nosync && h->tn.scan_place && h->misc.p != nullptr;
	const int64_t sp_nblk = (ngrp + RB3_SP_GROUPS - 1) / RB3_SP_GROUPS;
	if (fused) {
		const size_t need = (size_t)sp_nblk * 16 * 8 + 64;
		if (h->lbst.cap < need || !h->lbst.p) {
			if ((r = buf_ensure(h, h->lbst, need)) < 0) return r;
			HIPCHK(hipMemsetAsync(h->lbst.p, 0, h->lbst.cap, h->st)); // (state words: no launch has the epoch 0)
		}
	}
	if (nosync) { // before any launch: hipMalloc may synchronise
		// the number of slots is only known after the scan: one per window is the upper bound; where the index is run-coded an
		// estimate from the old count stands in (the emitting kernels compare the scan total with the capacity and leave the
		// array alone if it does not fit; the caller then emits again after its sync)
		const int64_t est = (FROM_PLAIN || full) ? nwin : slot_estimate(h, n2, ntot);
		if ((r = ib_ensure(h, dst, ngrp, est)) < 0) return r;
	}
	uint64_t slot_cap = nosync ? (uint64_t)h->ib[dst].slots_cap : ~0ull;
#ifdef RB3GPU_TEST_HOOKS
	if (nosync && !full && h->tn.reb_slot_cap > 0 && (uint64_t)h->tn.reb_slot_cap < slot_cap) slot_cap = (uint64_t)h->tn.reb_slot_cap;
#endif
	h->reb_slot_cap = slot_cap;
	IdxView old = view_of(h);
	uint32_t *gstat = (uint32_t*)h->gstat.p;
	uint64_t *gpre = (uint64_t*)h->gpre.p, *dtot = (uint64_t*)h->misc.p + MISC_IX_TOT;
	if (winpar) {
		if (!FROM_PLAIN && !rows_done) {
			const int64_t nt = n2 + 1;
			HIPCHK(hipMemsetAsync(jg, 0, (size_t)(nwin + 1) * 8, h->st)); // defined even if pos[] turns out invalid
			hipLaunchKernelGGL(k_win_rows, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, h->st, d_pos, n2, jg, nwin, skip);
		}
		if (runspace) {
			// Tiers by the average number of batch rows per group: the small-table kernel where most groups have few rows, then the
			// medium one on what it left (or on all groups where the small one would mostly fail), then the window kernels on the
			// list the last tier leaves (its length stays on the device: fixed grids, grid-stride loops).
			if (!prepared) { // (the first rebuild of a single-sync merge finds them cleared with everything else)
				HIPCHK(hipMemsetAsync(nglist, 0, 8, h->st));
				HIPCHK(hipMemsetAsync((uint64_t*)h->misc.p + MISC_RG_OVER, 0, 8, h->st));
			}
			const int64_t gw4 = (ngrp + RB3_RG_WAVES - 1) / RB3_RG_WAVES;
			const unsigned grs = (unsigned)(gw4 < 3072 ? gw4 : 3072), grm = (unsigned)(gw4 < 2048 ? gw4 : 2048);
			if (rows_per_group <= (double)h->tn.reb_t1_rows) {
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_reb_group<384, 128, false>), dim3(grs), dim3(64 * RB3_RG_WAVES), 0, h->st, old, d_pos, d_b2, ntot, (const int64_t*)jg, ngrp,
						gstat, (uint4*)h->gslots.p, gkind, (const uint32_t*)nullptr, (const uint32_t*)nullptr, glist[0], nglist, skip);
				// (what the small tables left: a few hundred groups in a grown pangenome, most of them in its first rounds; the grid for
				// one and a half times what the merge before left -- the kernel loops over its list whatever the grid)
				const int64_t g2 = h->reb_last[0] < 0 ? gw4 : (h->reb_last[0] + h->reb_last[0] / 2) / RB3_RG_WAVES + 16;
				const unsigned gr2 = (unsigned)(g2 < grm ? g2 : grm);
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_reb_group<624, 320, true>), dim3(gr2), dim3(64 * RB3_RG_WAVES), 0, h->st, old, d_pos, d_b2, ntot, (const int64_t*)jg, ngrp,
						gstat, (uint4*)h->gslots.p, gkind, (const uint32_t*)glist[0], (const uint32_t*)nglist, glist[1], nglist + 1, skip);
			} else
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_reb_group<624, 320, false>), dim3(grm), dim3(64 * RB3_RG_WAVES), 0, h->st, old, d_pos, d_b2, ntot, (const int64_t*)jg, ngrp,
						gstat, (uint4*)h->gslots.p, gkind, (const uint32_t*)nullptr, (const uint32_t*)nullptr, glist[1], nglist + 1, skip);
			// the window kernels behind the tiers loop over the last tier's list: grids for one and a half times the list of the merge
			// before (a launch of thousands of blocks that find nothing to do costs ~5-10 us, three times per round)
			const int64_t lw = h->reb_last[1] < 0 ? ngrp : h->reb_last[1] + h->reb_last[1] / 2 + 8; // groups
			const int64_t gwn = lw * RB3_GRP_WINS / RB3_REB_WAVES + 1;
			const unsigned gw = (unsigned)(gwn < 8192 ? gwn : 8192);
			const unsigned gl = (unsigned)(lw < 4096 ? lw : 4096);
			if (planes)
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_plane_group<true>), dim3((unsigned)(lw < 8192 ? lw : 8192)), dim3(256), 0, h->st, old, d_pos, d_b2, n2, ntot, (const int64_t*)jg, nwin, ngrp,
						gstat, (uint4*)h->pslots.p, skip, (const uint32_t*)glist[1], (const uint32_t*)(nglist + 1), (uint32_t)lcap, (unsigned long long*)h->misc.p + MISC_RG_OVER, (unsigned long long*)nullptr, gpos);
			else {
			if (n2 * RB3_WIN > 3 * ntot)
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass1w<FROM_PLAIN, 7, true>), dim3(gw), dim3(64 * RB3_REB_WAVES), 0, h->st, old, d_pos, d_b2, n2, ntot, (const int64_t*)jg,
						(uint4*)h->wstat.p, (uint32_t*)h->wplane.p, (uint16_t*)h->wruns.p, nwin, skip, (const uint32_t*)glist[1], (const uint32_t*)(nglist + 1), (uint32_t)lcap);
			else
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass1w<FROM_PLAIN, 3, true>), dim3(gw), dim3(64 * RB3_REB_WAVES), 0, h->st, old, d_pos, d_b2, n2, ntot, (const int64_t*)jg,
						(uint4*)h->wstat.p, (uint32_t*)h->wplane.p, (uint16_t*)h->wruns.p, nwin, skip, (const uint32_t*)glist[1], (const uint32_t*)(nglist + 1), (uint32_t)lcap);
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_decide<true>), dim3(gl), dim3(64), 0, h->st, (const uint4*)h->wstat.p, ntot, gstat, ngrp, skip,
					(const uint32_t*)glist[1], (const uint32_t*)(nglist + 1), (uint32_t)lcap, (unsigned long long*)h->misc.p + MISC_RG_OVER);
			}
		} else if (planes) {
			if (!prepared) HIPCHK(hipMemsetAsync((uint64_t*)h->misc.p + MISC_PP_OK, 0, 8, h->st));
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_plane_group<false>), dim3((unsigned)(ngrp < 16384 ? ngrp : 16384)), dim3(256), 0, h->st, old, d_pos, d_b2, n2, ntot, (const int64_t*)jg, nwin, ngrp,
					gstat, (uint4*)h->pslots.p, skip, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, (unsigned long long*)nullptr, (unsigned long long*)h->misc.p + MISC_PP_OK);
		} else {
			const dim3 g1w((unsigned)((nwin + RB3_REB_WAVES * RB3_REB_WPW - 1) / (RB3_REB_WAVES * RB3_REB_WPW))), b1w(64 * RB3_REB_WAVES);
			// the run-space short cut takes windows with up to 3 batch rows, or up to 7 where a window receives more than 3 on average
			if (!FROM_PLAIN && n2 * RB3_WIN > 3 * ntot)
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass1w<FROM_PLAIN, 7>), g1w, b1w, 0, h->st, old, d_pos, d_b2, n2, ntot, (const int64_t*)jg,
						(uint4*)h->wstat.p, (uint32_t*)h->wplane.p, (uint16_t*)h->wruns.p, nwin, skip, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
			else
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass1w<FROM_PLAIN, 3>), g1w, b1w, 0, h->st, old, d_pos, d_b2, n2, ntot, (const int64_t*)jg,
						(uint4*)h->wstat.p, (uint32_t*)h->wplane.p, (uint16_t*)h->wruns.p, nwin, skip, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_decide<false>), dim3((unsigned)ngrp), dim3(64), 0, h->st, (const uint4*)h->wstat.p, ntot, gstat, ngrp, skip,
					(const uint32_t*)nullptr, (const uint32_t*)nullptr);
		}
	} else {
		if (!FROM_PLAIN) {
			const int64_t nt = n2 + 1;
			HIPCHK(hipMemsetAsync(jg, 0, (size_t)(ngrp + 1) * 8, h->st)); // defined even if pos[] turns out invalid
			hipLaunchKernelGGL(k_group_rows, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, h->st, d_pos, n2, jg, ngrp, skip);
		}
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass1<FROM_PLAIN>), dim3((unsigned)ngrp), dim3(64), 0, h->st, old, d_pos, d_b2, n2, ntot, (const int64_t*)jg, gstat, ngrp, skip);
	}
	if (fused) {
		Acc7 ao;
		for (int a = 0; a < 7; ++a) ao.a[a] = h->acc[a];
		// k_scan_place is the first kernel of a merge that reads the batch's symbol totals (side stream: merge_core): the wait for them sits here,
		// behind the group kernels, where its ~15 us in the command processor pass beside their 250 (in front of them the chip idled)
		if (h->lf_wait) { HIPCHK(hipStreamWaitEvent(h->st, h->evx[2], 0)); h->lf_wait = false; }
		uint8_t *lbp = (uint8_t*)h->lbst.p;
		// the look-back words are told apart by a 20-bit tag of the launch, never cleared -- except where the tag comes round again: words that a
		// launch 2^20 merges ago left in blocks no launch has written since (the index was re-made smaller on this handle) would pass for this one's
		if (((h->lb_epoch + 1) & 0xFFFFFull) == 0) HIPCHK(hipMemsetAsync(lbp + 64, 0, h->lbst.cap > 64 ? h->lbst.cap - 64 : 0, h->st));
		hipLaunchKernelGGL(k_scan_place, dim3((unsigned)sp_nblk), dim3(RB3_SP_THREADS), 0, h->st, (const uint32_t*)gstat, ngrp, ntot, (const uint8_t*)(runspace ? gkind : nullptr), (const uint4*)h->gslots.p, (const uint4*)h->pslots.p,
				(const uint32_t*)(runspace ? gpos : nullptr), h->ib[dst].grp, (uint64_t*)(h->ib[dst].grp + h->ib[dst].grp_cap), (uint4*)h->ib[dst].slots, (unsigned long long*)dtot,
				(unsigned long long*)(lbp + 64), (unsigned int*)lbp /* the block counter sits in FRONT of the look-back state: a fixed address, zero between launches */, (++h->lb_epoch, (h->lb_epoch & 0xFFFFFull) ? h->lb_epoch : ++h->lb_epoch), ao, (const uint64_t*)((uint64_t*)h->misc.p + MISC_LF_TOT), skip,
				runspace ? (const uint32_t*)(nglist + 1) : (const uint32_t*)nullptr, (uint32_t)lcap, slot_cap, h->tn.abs_limit);
		*ongrp = ngrp;
		return 0;
	}
	uint64_t total[8];
	if ((r = scan_records(h, gstat, ngrp, gpre, dtot, nosync ? nullptr : total)) < 0) return r;
	if (!nosync) {
		oacc[0] = 0;
		for (int a = 0; a < 6; ++a) oacc[a + 1] = oacc[a] + (int64_t)total[a];
		if (oacc[6] != ntot) {
			if (h->opt.verbose >= 1) fprintf(stderr, "[E::rb3gpu] symbol counts do not add up: %lld vs %lld\n", (long long)oacc[6], (long long)ntot);
			return RB3GPU_EINTERNAL;
		}
		*onslots = (int64_t)total[6];
		if ((r = ib_ensure(h, dst, ngrp, *onslots)) < 0) return r;
	}
	if (winpar && runspace) {
		const int64_t lw2 = h->reb_last[1] < 0 ? ngrp : h->reb_last[1] + h->reb_last[1] / 2 + 8; // (as above: the grid for the list the merge before left)
		if (planes)
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_place_pg<true>), dim3((unsigned)(lw2 < 16384 ? (lw2 + 3) / 4 : 4096)), dim3(256), 0, h->st, (const uint32_t*)gstat, (const uint64_t*)gpre, (const uint64_t*)dtot, (const uint4*)h->pslots.p,
					h->ib[dst].grp, (uint4*)h->ib[dst].slots, ngrp, ntot, skip, (const uint32_t*)glist[1], (const uint32_t*)(nglist + 1), (uint32_t)lcap, slot_cap, h->tn.abs_limit);
		else
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2w<true>), dim3(lw2 < 4096 ? (unsigned)lw2 : 4096u), dim3(64 * RB3_REB_WAVES), 0, h->st, (const uint4*)h->wstat.p, (const uint32_t*)h->wplane.p, (const uint16_t*)h->wruns.p, ntot,
				(const uint32_t*)gstat, (const uint64_t*)gpre, (const uint64_t*)dtot, h->ib[dst].grp, (uint4*)h->ib[dst].slots, nwin, skip, (const uint32_t*)glist[1], (const uint32_t*)(nglist + 1), (uint32_t)lcap, slot_cap, (int64_t)-1, (int64_t)-1, h->tn.abs_limit);
		hipLaunchKernelGGL(k_place, dim3((unsigned)((ngrp + 3) / 4)), dim3(256), 0, h->st, (const uint8_t*)gkind, (const uint32_t*)gstat, (const uint64_t*)gpre, (const uint64_t*)dtot,
				(const uint4*)h->gslots.p, h->ib[dst].grp, (uint4*)h->ib[dst].slots, ngrp, nwin, ntot, skip, (const uint32_t*)(nglist + 1), (uint32_t)lcap, slot_cap, h->tn.abs_limit);
	} else if (planes)
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_place_pg<false>), dim3((unsigned)(ngrp < 65536 ? (ngrp + 3) / 4 : 16384)), dim3(256), 0, h->st, (const uint32_t*)gstat, (const uint64_t*)gpre, (const uint64_t*)dtot, (const uint4*)h->pslots.p,
				h->ib[dst].grp, (uint4*)h->ib[dst].slots, ngrp, ntot, skip, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, slot_cap, h->tn.abs_limit);
	else if (winpar)
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2w<false>), dim3((unsigned)ngrp), dim3(64 * RB3_REB_WAVES), 0, h->st, (const uint4*)h->wstat.p, (const uint32_t*)h->wplane.p, (const uint16_t*)h->wruns.p, ntot,
				(const uint32_t*)gstat, (const uint64_t*)gpre, (const uint64_t*)dtot, h->ib[dst].grp, (uint4*)h->ib[dst].slots, nwin, skip, (const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, slot_cap, (int64_t)-1, (int64_t)-1, h->tn.abs_limit);
	else
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<FROM_PLAIN>), dim3((unsigned)ngrp), dim3(64), 0, h->st, old, d_pos, d_b2, n2, ntot, (const int64_t*)jg,
				(const uint32_t*)gstat, (const uint64_t*)gpre, (const uint64_t*)dtot, h->ib[dst].grp, (uint4*)h->ib[dst].slots, ngrp, skip, h->tn.abs_limit);
	// the compact copy of the slot words of the new directory (IdxView.gsm), behind its entries
	hipLaunchKernelGGL(k_grp_compact, dim3((unsigned)((ngrp + 255) / 256)), dim3(256), 0, h->st, (const uint64_t*)h->ib[dst].grp, ngrp, (uint64_t*)(h->ib[dst].grp + h->ib[dst].grp_cap), skip);
	*ongrp = ngrp;
	return 0;
}
} // extern "C++"

/* make ib[1-cur] (just built) the index */
static void index_install(rb3gpu_t *h, int64_t ngrp, int64_t nslots, int64_t ntot, const int64_t acc[7])
{
	h->cur = 1 - h->cur;
	if (RB3_ABS_HEADERS(ntot, h->tn.abs_limit) && (ntot >= RB3_ABS_LIMIT || h->tn.abs_table)) { // the LF bases every 2^31 symbols, from the directory the rebuild has just written (IdxView.sb)
		auto &b = h->ib[h->cur]; // (ib_ensure made room for the table)
		if (b.sbt && b.sbt_cap >= (size_t)(ntot >> RB3_SB_BITS) + 1) hipLaunchKernelGGL(k_sb_table, dim3((unsigned)((((ntot >> RB3_SB_BITS) + 1) * 8 + 63) / 64)), dim3(64), 0, h->st, (const uint64_t*)b.grp, (ntot >> RB3_SB_BITS) + 1, b.sbt);
	}
	h->grp = h->ib[h->cur].grp, h->slots = h->ib[h->cur].slots, h->ngrp = ngrp, h->nslots = nslots, h->n = ntot;
	memcpy(h->acc, acc, sizeof(h->acc));
	h->stt.bytes_index = ngrp * (int64_t)RB3_GRP_ALLOC + nslots * (int64_t)sizeof(rb3_slot_t);
	// do not sit on a large spare buffer: the next merge re-allocates it (it is sized for a bigger index anyway)
	const int o = 1 - h->cur;
	if (h->ib[o].slots_cap * sizeof(rb3_slot_t) > ((size_t)4 << 30) && !vm_find(h, h->ib[o].slots)) ib_release(h, o); // (a range that grows in place stays: the next rebuild maps what it lacks)
}

/* histogram + row words of B2 (LF word of every row, fm-index.c:206-216) into d_row.  Totals stay on the device (misc[MISC_LF_TOT..]); acc2 != NULL also
 * brings the C array of B2 to the host (one sync) and checks the symbols (fm-index.c:124-125). */
static int lf_build(rb3gpu_t *h, int64_t len, const uint8_t *d_b2, int64_t *d_row, int64_t *acc2, bool words = true, bool rows_filled = false, bool side = false)
{
	hipStream_t st = side ? h->st2 : h->st; // (side: only with words = false and rows_filled = true, i.e. nothing but the histogram and its scan)
	const int64_t ntile = (len + RB3_TILE - 1) / RB3_TILE;
	int r;
	if ((r = buf_ensure(h, h->tcnt, (size_t)ntile * 32)) < 0) return r;
	if ((r = buf_ensure(h, h->tpre, (size_t)ntile * 64)) < 0) return r;
	if ((r = buf_ensure(h, h->misc, MISC_WORDS * 8)) < 0) return r;
	uint64_t *dtot = (uint64_t*)h->misc.p + MISC_LF_TOT;
	hipLaunchKernelGGL(k_tile_hist, dim3((unsigned)ntile), dim3(256), 0, st, d_b2, len, (uint32_t*)h->tcnt.p);
	uint64_t total[8];
	if ((r = scan_records(h, (const uint32_t*)h->tcnt.p, ntile, (uint64_t*)h->tpre.p, dtot, acc2 ? total : nullptr, side)) < 0) return r;
	if (acc2) {
		if (total[6] != 0) return RB3GPU_ESYMBOL;
		acc2[0] = 0;
		for (int a = 0; a < 6; ++a) acc2[a + 1] = acc2[a] + (int64_t)total[a];
	}
	if (words) hipLaunchKernelGGL(k_lf2, dim3((unsigned)ntile), dim3(256), 0, h->st, d_b2, len, (const uint64_t*)h->tpre.p, (const uint64_t*)dtot, (uint64_t*)d_row);
	else if (!rows_filled) HIPCHK(hipMemsetAsync(d_row, 0xff, (size_t)len * 8, h->st)); // text-order walk: the rows only hold records, all unvisited
	return 0;
}

/* the sampled LF-consistency check of pos[] (k_lf_check), against the index as it is BEFORE the merge is installed; counts into
 * misc[4] (with the unsettled tentative records: a failure first makes the merge redo its rank phase without speculation) */
struct JuncArgs { const uint64_t *tw; const rb3_stretch_t *tab; const uint32_t *sidctr; const int64_t *jmet; int64_t nwalk; const unsigned long long *nwalk_dev; };
#define MISC_JUNC 41 /* junctions k_junction_check looked at */
#ifndef RB3_JUNC_BLOCKS
#define RB3_JUNC_BLOCKS 512 /* blocks of k_junction_check (side stream, beside the rebuild) */
#endif
static bool launch_lf_check(rb3gpu_t *h, const int64_t *dpos, const uint8_t *d_b2, int64_t len, bool side, hipEvent_t here = nullptr, const JuncArgs *ja = nullptr) // here: an event the caller has just recorded on the main stream (saves recording another one: every record is a packet the command processor works through, ~3-5 us)
{
	const int64_t stride = h->tn.lf_check;
	if (stride <= 0 || h->tpre.p == nullptr) return false;
	unsigned long long *misc = (unsigned long long*)h->misc.p;
	const int64_t ns = (len + stride - 1) / stride;
	hipStream_t s = h->st;
	if (side) { // beside the rebuild: pos[] is read-only from here on, and a rebuild from a wrong-but-monotone pos[] is harmless (never installed)
		if ((here == nullptr && hipEventRecord(h->evx[0], h->st) != hipSuccess) || hipStreamWaitEvent(h->st2, here ? here : h->evx[0], 0) != hipSuccess) side = false;
		else s = h->st2;
	}
	if (!side && h->lf_wait) { (void)hipStreamWaitEvent(h->st, h->evx[2], 0); h->lf_wait = false; } // (on the main stream after all: the histogram it reads runs on the side stream)
	hipLaunchKernelGGL(k_lf_check, dim3((unsigned)((ns * 8 + 255) / 256)), dim3(256), 0, s, view_of(h), dpos, d_b2, len, (const uint64_t*)h->tpre.p,
			(const uint64_t*)(misc + MISC_LF_TOT), stride, misc + 2, misc + MISC_LF_CHK);
	if (ja != nullptr && h->tn.junction_check) // every junction of the speculative walk, deterministically (k_junction_check)
		hipLaunchKernelGGL(k_junction_check, dim3(RB3_JUNC_BLOCKS), dim3(256), 0, s, view_of(h), dpos, ja->tw, len, ja->tab, ja->sidctr, ja->jmet, ja->nwalk, ja->nwalk_dev, misc + 2, misc + MISC_LF_CHK, misc + MISC_JUNC,
				h->tn.junction_check, (int)(h->stt.n_rounds % h->tn.junction_check)); // (a different residue of the stretch ids every merge)
	if (side) (void)hipEventRecord(h->evx[1], h->st2);
	return side; // true: the caller makes its stream wait for evx[1] before it reads the counters
}

static int pick_split(const rb3gpu_t *h, int64_t len, int64_t m2)
{
	if (h->opt.split_log2 < 0) return 0;
	if (h->opt.split_log2 > 0) return h->opt.split_log2 > 40 ? 40 : h->opt.split_log2;
	// automatic: aim at ~64k walkers; never split strings that are already short
	int lg = 8;
	while ((len >> lg) > 65536 && lg < 30) ++lg;
	if (m2 > 0 && len / m2 <= (2LL << lg)) return 0;
	return lg;
}

/* ---- the merge in three stages (the collective of a multi-GPU build goes between 2 and 3) ---- */

int rb3gpu_mg_begin(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt, void *d_pos_ext, int64_t acc2_out[RB3GPU_ASIZE+1])
{
	if (!h || len <= 0 || !d_bwt) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->n <= 0 || h->grp == nullptr) return RB3GPU_ESTATE;
	int r;
	h->mg_active = 0;
	if (d_pos_ext) h->mg_pos = (int64_t*)d_pos_ext;
	else {
		if ((r = buf_ensure(h, h->pos, (size_t)len * 8)) < 0) return r;
		h->mg_pos = (int64_t*)h->pos.p;
	}
	if ((r = buf_ensure(h, h->misc, MISC_WORDS * 8)) < 0) return r;
	HIPCHK(hipEventRecord(h->ev[0], h->st));
	if ((r = lf_build(h, len, d_bwt, h->mg_pos, h->mg_acc2)) < 0) return r;
	if (h->mg_acc2[1] <= 0) return RB3GPU_EINVAL; // a batch always ends with a sentinel
	HIPCHK(hipEventRecord(h->ev[1], h->st));
	HIPCHK(hipMemsetAsync(h->misc.p, 0, 128, h->st)); // words 0..15; the scan totals behind them stay
	HIPCHK(hipEventRecord(h->ev[2], h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	h->stt.ms_lf += ev_ms(h->ev[0], h->ev[1]);
	h->stt.ms_rank += ev_ms(h->ev[1], h->ev[2]);
	h->mg_len = len, h->mg_b2 = d_bwt, h->mg_active = 1;
	if (acc2_out) memcpy(acc2_out, h->mg_acc2, sizeof(h->mg_acc2));
	return 0;
}

/* run LF walkers: walkers == NULL -> one per sentinel row plus automatic SA-order splitting.
 * tent: allow tentative records (single-call merges only; see k_chain) */
static int mg_walk_impl(rb3gpu_t *h, int64_t n_walkers, const rb3gpu_walker_t *walkers, int64_t stop_row, int64_t *arrive, int tent)
{
	if (!h || (walkers && n_walkers <= 0)) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (!h->mg_active) return RB3GPU_ESTATE;
	const int64_t len = h->mg_len, m2 = h->mg_acc2[1];
	unsigned long long *qhead = (unsigned long long*)h->misc.p, *nsteps = qhead + 1;
	int r, logM = 0;
	int64_t nwalk;
	Walker *dwl = nullptr;
	int64_t *darr = nullptr;
	if (walkers) {
		for (int64_t i = 0; i < n_walkers; ++i)
			if (walkers[i].row < 0 || walkers[i].row >= len || walkers[i].nsteps <= 0) return RB3GPU_EINVAL;
		nwalk = n_walkers;
		if ((r = buf_ensure(h, h->wl, (size_t)n_walkers * 40)) < 0) return r;
		dwl = (Walker*)h->wl.p, darr = (int64_t*)((char*)h->wl.p + (size_t)n_walkers * 32);
		if (stop_row >= len) return RB3GPU_EINVAL;
		{ // ka0 == RB3GPU_KA_SENTINEL: a sentinel row, whose insertion point is acc[1] of the index (fm-index.c:164)
			rb3gpu_walker_t *tmp = (rb3gpu_walker_t*)malloc((size_t)n_walkers * sizeof(rb3gpu_walker_t));
			if (!tmp) return RB3GPU_ENOMEM;
			memcpy(tmp, walkers, (size_t)n_walkers * sizeof(rb3gpu_walker_t));
			for (int64_t i = 0; i < n_walkers; ++i)
				if (tmp[i].ka0 == RB3GPU_KA_SENTINEL) tmp[i].ka0 = h->acc[1];
			hipError_t e = hipMemcpy(dwl, tmp, (size_t)n_walkers * 32, hipMemcpyHostToDevice);
			free(tmp);
			HIPCHK(e);
		}
		HIPCHK(hipMemsetAsync(darr, 0xff, 8, h->st));
	} else {
		logM = pick_split(h, len, m2);
		nwalk = m2;
		if (logM > 0) {
			const int64_t M = 1LL << logM, first = (m2 + M - 1) >> logM << logM;
			if (first < len) nwalk += (len - first + M - 1) >> logM;
		}
	}
	// tentative records need merged positions < 2^38
	if (!h->tn.tent) tent = 0;
	if ((walkers ? n_walkers : nwalk - m2) <= 0 || h->n + len >= lim_pos(h) || stop_row >= 0) tent = 0;
	rb3_stretch_t *tab = nullptr;
	int32_t *sfin = nullptr;
#ifdef RB3GPU_TEST_HOOKS
	const uint32_t sid_limit = h->tn.tent_limit >= 0 ? (uint32_t)h->tn.tent_limit : 0xFFFFFFFFu; // test hook: a tiny stretch table
#else
	const uint32_t sid_limit = 0xFFFFFFFFu;
#endif
	uint32_t *sidctr = (uint32_t*)(qhead + 5);
	if (tent && (r = tent_prepare(h, &tab, &sfin)) < 0) return r;
	HIPCHK(hipMemsetAsync(qhead, 0, 8, h->st));
	HIPCHK(hipMemsetAsync(sidctr, 0, 8, h->st));
	// octets per wave: all 8 when there are enough walkers to fill the chip (256 CUs x 32 waves), fewer
	// when the launch is latency-bound anyway
	const int octs = h->tn.octs; // 8 by default; measured: fewer octets per wave (more waves) is slower even for few walkers
	int64_t nblk = (nwalk + 4 * octs - 1) / (4 * octs);
	if (nblk > 256 * 8) nblk = 256 * 8;
	if (nblk < 1) nblk = 1;
	HIPCHK(hipEventRecord(h->ev[6], h->st));
	{
		const IdxView iv = view_of(h);
		const int64_t sr = stop_row < 0 ? -1 : stop_row;
		const dim3 grid((unsigned)nblk), blk(256);
#define RB3_LAUNCH_CHAIN(L, D, T) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain<L, D, T, 0>), grid, blk, 0, h->st, iv, h->mg_pos, len, m2, \
			walkers ? 0 : logM, (const Walker*)dwl, nwalk, walkers ? sr : (int64_t)-1, darr, qhead, nsteps, octs, tab, sidctr, sid_limit, (const uint64_t*)nullptr)
		const int sel = (walkers ? 4 : 0) | (iv.dense == 2 ? 2 : 0) | (tent ? 1 : 0);
		switch (sel) {
		case 0: RB3_LAUNCH_CHAIN(false, false, false); break;
		case 1: RB3_LAUNCH_CHAIN(false, false, true); break;
		case 2: RB3_LAUNCH_CHAIN(false, true, false); break;
		case 3: RB3_LAUNCH_CHAIN(false, true, true); break;
		case 4: RB3_LAUNCH_CHAIN(true, false, false); break;
		case 5: RB3_LAUNCH_CHAIN(true, false, true); break;
		case 6: RB3_LAUNCH_CHAIN(true, true, false); break;
		default: RB3_LAUNCH_CHAIN(true, true, true); break;
		}
#undef RB3_LAUNCH_CHAIN
		HIPCHK(hipEventRecord(h->ev[7], h->st));
		if (tent) {
			launch_settle(h, iv, tab, nullptr, (const uint32_t*)sidctr, sfin, qhead + 2, (int)RB3_TENT_IDS, 1); // (no second chance on this path: follow every path to its end)
			hipLaunchKernelGGL(k_pos_finalize, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, h->st, h->mg_pos, len, (const int32_t*)sfin, qhead + 2);
		}
	}
	HIPCHK(hipEventRecord(h->ev[5], h->st));
	if (walkers && arrive) HIPCHK(hipMemcpyAsync(arrive, darr, 8, hipMemcpyDeviceToHost, h->st));
	unsigned long long nsid = 0;
	if (tent) HIPCHK(hipMemcpyAsync(&nsid, sidctr, 8, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	if (tent) tent_used(h, nsid); // (a full table poisons the records it could not serve: they count as unsettled)
	h->stt.ms_chain += ev_ms(h->ev[6], h->ev[7]), h->stt.ms_rank += ev_ms(h->ev[6], h->ev[5]);
	h->stt.n_rank_launches += 1, h->stt.n_rounds += 1;
	guard_check(h, __func__);
	return 0;
}

int rb3gpu_mg_walk(rb3gpu_t *h, int64_t n_walkers, const rb3gpu_walker_t *walkers, int64_t stop_row, int64_t *arrive)
{
	return mg_walk_impl(h, n_walkers, walkers, stop_row, arrive, 0);
}

int rb3gpu_mg_pos_ptr(rb3gpu_t *h, void **d_pos, int64_t *len)
{
	if (!h || !d_pos) return RB3GPU_EINVAL;
	if (!h->mg_active) return RB3GPU_ESTATE;
	*d_pos = h->mg_pos;
	if (len) *len = h->mg_len;
	return 0;
}

/* stage 3: validate pos[], interleave and rebuild; rank_only skips the rebuild */
static int mg_finish(rb3gpu_t *h, int commit, int64_t *host_pos, int rank_only)
{
	if (!h->mg_active) return RB3GPU_ESTATE;
	const int64_t len = h->mg_len, ntot = h->n + len;
	int r;
	h->mg_active = 0;
	unsigned long long *misc = (unsigned long long*)h->misc.p;
	HIPCHK(hipEventRecord(h->ev[2], h->st));
	hipLaunchKernelGGL(k_pos_check, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, h->st, (const int64_t*)h->mg_pos, len, ntot, misc + 2);
	(void)launch_lf_check(h, (const int64_t*)h->mg_pos, h->mg_b2, len, false);
	unsigned long long hm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	HIPCHK(hipMemcpyAsync(hm, misc, sizeof(hm), hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	h->stt.n_lf_steps += (int64_t)hm[1];
	h->stt.n_lf_checked += (int64_t)hm[MISC_LF_CHK];
	if (hm[2] != 0 || hm[3] != 0 || hm[4] != 0 || hm[MISC_LF_CHK + 1] != 0) {
		if (h->opt.verbose >= 1) fprintf(stderr, "[E::rb3gpu] rank phase left %llu rows unset, %llu out of order, %llu tentative records unsettled, %llu sampled rows that fail the LF relation\n", hm[2], hm[3], hm[4], hm[MISC_LF_CHK + 1]);
		return RB3GPU_EINTERNAL;
	}
	int64_t ngrp = 0, nslots = 0, acc[7];
	if (!rank_only) {
		if ((r = build_index<false>(h, len, h->mg_b2, (const int64_t*)h->mg_pos, ntot, false, &ngrp, &nslots, acc)) < 0) return r;
	}
	HIPCHK(hipEventRecord(h->ev[3], h->st));
	if (host_pos) HIPCHK(hipMemcpyAsync(host_pos, h->mg_pos, (size_t)len * 8, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	h->stt.ms_build += ev_ms(h->ev[2], h->ev[3]);
	h->stt.n_symbols_merged += len;
	if (!rank_only) {
		for (int a = 0; a <= 6; ++a) if (acc[a] != h->acc[a] + h->mg_acc2[a]) return RB3GPU_EINTERNAL;
		h->stt.bytes_rebuild += 9 * len + h->stt.bytes_index + ngrp * (int64_t)sizeof(rb3_grp_t) + nslots * (int64_t)sizeof(rb3_slot_t);
		if (commit) index_install(h, ngrp, nslots, ntot, acc);
	}
	if (h->opt.verbose >= 3)
		fprintf(stderr, "[M::%s::%.3f] merged %lld symbols (%lld strings): %llu LF steps, rebuild %.3f ms\n", __func__,
				now_s() - h->t0, (long long)len, (long long)h->mg_acc2[1], hm[1], ev_ms(h->ev[2], h->ev[3]));
	return 0;
}

int rb3gpu_mg_finish(rb3gpu_t *h, int commit)
{
	if (!h) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	return mg_finish(h, commit, nullptr, 0);
}

/* the staged implementation (several host round trips); also the fallback of merge_fast */
static int merge_staged(rb3gpu_t *h, int64_t len, const uint8_t *d_b2, int commit, int64_t *host_pos, int64_t *host_acc2, int rank_only,
		int64_t n_walkers, const rb3gpu_walker_t *walkers, int tent)
{
	int r;
	if ((r = rb3gpu_mg_begin(h, len, d_b2, nullptr, host_acc2)) < 0) return r;
	if ((r = mg_walk_impl(h, n_walkers, walkers, -1, nullptr, tent)) < 0) return r;
	if (tent) { // tentative records are optimistic: if any is left unsettled, redo the rank phase without them
		unsigned long long unsettled = 0;
		HIPCHK(hipMemcpy(&unsettled, (unsigned long long*)h->misc.p + 4, 8, hipMemcpyDeviceToHost));
		if (unsettled != 0) {
			h->stt.n_fallbacks += 1;
			if ((r = lf_build(h, len, d_b2, h->mg_pos, nullptr)) < 0) return r; // fresh row words
			HIPCHK(hipMemsetAsync(h->misc.p, 0, 128, h->st));
			if ((r = mg_walk_impl(h, n_walkers, walkers, -1, nullptr, 0)) < 0) return r;
		}
	}
	return mg_finish(h, commit, host_pos, rank_only);
}

/* One merge with a single host synchronisation at the end: everything the host would have to read
 * in between (symbol totals, slot count, validation counters) stays in device memory, the new index
 * is built into a pooled buffer sized by its upper bound (one slot per window), and all checks are
 * made after the final copy-back.  Needs an explicit walker list (the automatic SA-order split needs
 * the number of strings on the host to size its launch) and an index small enough for the upper
 * bound to be affordable; anything else goes through merge_staged. */
/* walkers given by text position (text-order walk) -> the same list by row, for the paths that walk row words */
__global__ void __launch_bounds__(256) k_walker_rows(Walker *wl, int64_t n, const uint64_t *tw)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) wl[i].row = (int64_t)(tw[wl[i].row] >> 3);
}

/* one walker per string, made on the device: the suffixes with rows 0..m2-1 are the sentinels */
__global__ void __launch_bounds__(256) k_walkers_per_string(Walker *wl, int64_t m2, const uint64_t *tw, int64_t n)
{
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	const int64_t r = (int64_t)(tw[t] >> 3);
	const bool sentinel = t + 1 == n || (tw[t + 1] & 7u) == 0; // the word after it says which symbol sits at t
	if (r < m2 && sentinel) { Walker w; w.row = t, w.ka0 = -2, w.nsteps = INT64_MAX / 2, w.flags = 0; wl[r] = w; }
}

/* The walker list of a batch of long strings, made on the device (VERDICT r4 "what's missing" 4: the host spent 70 ms of one core per 152-genome
 * build on it, outside the timed region): one walker per string -- at its sentinel -- and one at every text position that is a multiple of `step`
 * strictly inside a string, exactly the list of rb3h_walkers_text (host/sais.c:74-113: same pre-roll, same probe distance, the same rule for the
 * multiple next to a sentinel).  Slot k - 1 of the list belongs to the multiple k * step (k = 1 .. K), slot K + j to the sentinel of string j; a
 * multiple that gets no walker leaves its slot empty (row -1: k_chain passes over it).  The slots are in TEXT ORDER, as the host's list is -- the multiple
 * p of a string behind j sentinels at slot p / step - 1 + j, the sentinel at e of string j at slot e / step + j --: octet i takes walker i first, so the
 * walkers of a wave are neighbours in the text and run in lock step; with the sentinels' walkers at the end of the list instead, a build of 320 relatives
 * with the masks pinned to 256 bits redid 5 of its merges where the host's list makes it redo 1 (same list, another order: tools/gpu_r5_exp8.sh).
 * Pass 1: sent[j] = text position of the sentinel of string j (its row IS j: the suffixes that start at a sentinel sort first, in string order). */
#define RB3_WL_PREROLL ((int64_t)RB3_TENT_MIN_AGE) /* = RB3H_PREROLL: a walker starts as many positions to the right of its segment as it must be old to record */
#define RB3_WL_PROBE   64  /* = RB3H_PROBE */
#define RB3_WL_MIN_SEG 128 /* = RB3H_MIN_SEG */
__global__ void __launch_bounds__(256) k_wl_sentinels(const uint64_t *tw, int64_t n, int64_t m2, int64_t *sent)
{
	const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n) return;
	const bool sentinel = t + 1 == n || (tw[t + 1] & 7u) == 0; // the word after it says which symbol sits at t
	const int64_t r = (int64_t)(tw[t] >> 3);
	if (sentinel && r < m2) sent[r] = t;
}

/* Pass 2: the list.  sent[] came filled with -1: a string count that is not the batch's leaves entries there, and the walkers of those strings
 * are not made (their rows stay unset: the merge reports EINVAL, as for a wrong count of the per-string list) */
/* sa != NULL: the sentinels' positions straight from the batch's suffix array (sent[j] = sa[j], what k_wl_sentinels_sa would have copied: one launch less in front of the walkers,
 * ~10 us of a merge with the chip idle) */
__global__ void __launch_bounds__(256) k_wl_make(const int64_t *sent_, const uint32_t *sa, int64_t m2, int64_t n, int64_t step, int64_t K, Walker *wl)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= K + m2) return;
	auto sent = [&](int64_t j) -> int64_t { if (sa == nullptr) return sent_[j]; const uint32_t v = sa[j]; return v < (uint32_t)n ? (int64_t)v : -1; };
	const int64_t INF = INT64_MAX / 2;
	Walker w;
	int64_t slot = -1;
	w.row = -1, w.ka0 = -1, w.nsteps = 0, w.flags = 0;
	if (i < K) { // the multiple p = (i + 1) * step: string [b, e) with sentinel e = the first sentinel at or behind p
		const int64_t p = (i + 1) * step;
		int64_t lo = 0, hi = m2; // first j with sent(j) >= p
		while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sent(mid) >= p) hi = mid; else lo = mid + 1; }
		slot = i + lo;
		if (lo < m2 && p < n) {
			const int64_t e = sent(lo), b = lo > 0 ? sent(lo - 1) + 1 : 0;
			const bool ok = e >= 0 && (lo == 0 || sent(lo - 1) >= 0) && p > b && p < e && !(e - p < RB3_WL_MIN_SEG && e - p < step);
			if (ok) {
				const int64_t pre = e - 1 - p < RB3_WL_PREROLL ? e - 1 - p : RB3_WL_PREROLL;
				w.row = p + pre, w.ka0 = -1, w.flags = pre << 8;
				if (pre == RB3_WL_PREROLL && e - 1 - (p + pre) >= RB3_WL_PROBE) w.flags |= (int64_t)RB3_WL_PROBE << 16;
				w.nsteps = p - step > b ? step + pre : INF; // (the multiple before it, if it lies inside the string, has a walker: it is further from the sentinel)
			}
		}
	} else { // the sentinel of string j
		const int64_t j = i - K, e = sent(j), b = j > 0 ? sent(j - 1) + 1 : 0;
		if (e >= 0) slot = e / step + j;
		if (e >= 0 && (j == 0 || sent(j - 1) >= 0)) {
			int64_t q = e > 0 ? (e - 1) / step * step : 0; // the last multiple below e ... that has a walker
			if (q > b && q < e && (e - q < RB3_WL_MIN_SEG && e - q < step)) q -= step;
			w.row = e, w.ka0 = -2, w.flags = 0;
			w.nsteps = q > b && q >= step ? e - q : INF;
		}
	}
	if (slot >= 0 && slot < K + m2) wl[slot] = w; // (the list came filled with empty slots)
}

__global__ void __launch_bounds__(256) k_walkers_sentinel_rows(Walker *wl, int64_t m2)
{
	const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r < m2) { Walker w; w.row = r, w.ka0 = -2, w.nsteps = INT64_MAX / 2, w.flags = 0; wl[r] = w; }
}

static int walkers_text_to_rows(rb3gpu_t *h, int64_t n_walkers, const rb3gpu_walker_t *walkers, const uint64_t *d_tw, rb3gpu_walker_t **out)
{
	int r;
	*out = nullptr;
	if ((r = buf_ensure(h, h->wl, (size_t)n_walkers * 40)) < 0) return r;
	rb3gpu_walker_t *w = (rb3gpu_walker_t*)malloc((size_t)n_walkers * sizeof(rb3gpu_walker_t));
	if (!w) return RB3GPU_ENOMEM;
	hipError_t e = hipMemcpyAsync(h->wl.p, walkers, (size_t)n_walkers * 32, hipMemcpyHostToDevice, h->st);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_walker_rows, dim3((unsigned)((n_walkers + 255) / 256)), dim3(256), 0, h->st, (Walker*)h->wl.p, n_walkers, d_tw);
		e = hipMemcpyAsync(w, h->wl.p, (size_t)n_walkers * 32, hipMemcpyDeviceToHost, h->st);
	}
	if (e == hipSuccess) e = hipStreamSynchronize(h->st);
	if (e != hipSuccess) { (void)hipGetLastError(); free(w); return RB3GPU_ENODEV; }
	*out = w;
	return 0;
}

static int merge_core(rb3gpu_t *h, int64_t len, const uint8_t *d_b2, int commit, int64_t *host_pos, int64_t *host_acc2, int rank_only,
		int64_t n_walkers, const rb3gpu_walker_t *walkers, const uint64_t *d_tw = nullptr, int thin = 1);

/* the paths that walk row words, for a batch that came with text-order words */
static int merge_staged_text(rb3gpu_t *h, int64_t len, const uint8_t *d_b2, int commit, int64_t *host_pos, int64_t *host_acc2, int rank_only,
		int64_t n_walkers, const rb3gpu_walker_t *walkers, const uint64_t *d_tw, int tent)
{
	rb3gpu_walker_t *wr = nullptr;
	int r;
	if ((r = walkers_text_to_rows(h, n_walkers, walkers, d_tw, &wr)) < 0) return r;
	r = merge_staged(h, len, d_b2, commit, host_pos, host_acc2, rank_only, n_walkers, wr, tent);
	free(wr);
	return r;
}

/* every thin-th walker of a list in text order (rb3h_walkers_text / rb3h_build_bwt_walkers: per string its inner walkers from left to
 * right, each with the distance to the one before, then the walker of its sentinel): the segments of the dropped ones go to their
 * right-hand neighbours */
static rb3gpu_walker_t *thin_walkers(int64_t n, const rb3gpu_walker_t *w, int thin, int64_t *n_out)
{
	rb3gpu_walker_t *o = (rb3gpu_walker_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(rb3gpu_walker_t));
	if (!o) return nullptr;
	const int64_t INF = INT64_MAX / 2;
	int64_t k = 0, idx = 0, acc = 0;
	for (int64_t i = 0; i < n; ++i) {
		const int64_t pre = w[i].flags >> 8 & 0xFF; // (rb3h_walkers_text: the walker starts that many positions outside its segment, and nsteps counts them)
		acc = (acc >= INF || w[i].nsteps >= INF) ? INF : acc + w[i].nsteps - pre;
		const bool sentinel = w[i].ka0 == RB3GPU_KA_SENTINEL || w[i].ka0 >= 0; // (a walker that knows its insertion point ends a string's group, or is somebody's hand-off: always kept)
		if (sentinel || idx % thin == thin - 1) { o[k] = w[i], o[k].nsteps = acc >= INF ? INF : acc + pre, ++k, acc = 0; }
		idx = sentinel ? 0 : idx + 1;
	}
	*n_out = k;
	return o;
}

/* the list of k_wl_sentinels / k_wl_make into wl (K + n_strings slots), queued on the handle's stream; scratch: n_strings words */
static void step_list_launch(rb3gpu_t *h, int64_t len, const uint64_t *d_tw, int64_t n_strings, int64_t step, int64_t *d_sent, Walker *wl, hipStream_t st = nullptr, bool wl_cleared = false)
{
	if (st == nullptr) st = h->st;
	const int64_t K = len / step;
	if (!wl_cleared) (void)hipMemsetAsync(wl, 0xff, (size_t)(K + n_strings) * 32, st); // every slot empty (row -1) (wl_cleared: the merge's fill kernel has done it)
	if (h->mg_sa != nullptr && len < (1LL << 32)) { // (a wrong string count: sa[j] of a j that is no sentinel's row is a position whose next word does not start a string -- k_wl_make's
		// binary search then meets unsorted positions; the merge's own count of the sentinels decides)
		hipLaunchKernelGGL(k_wl_make, dim3((unsigned)((K + n_strings + 255) / 256)), dim3(256), 0, st, (const int64_t*)nullptr, h->mg_sa, n_strings, len, step, K, wl);
		return;
	}
	(void)hipMemsetAsync(d_sent, 0xff, (size_t)n_strings * 8, st);
	hipLaunchKernelGGL(k_wl_sentinels, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, st, d_tw, len, n_strings, d_sent);
	hipLaunchKernelGGL(k_wl_make, dim3((unsigned)((K + n_strings + 255) / 256)), dim3(256), 0, st, (const int64_t*)d_sent, (const uint32_t*)nullptr, n_strings, len, step, K, wl);
}

/* ... and on the host, without its empty slots (the paths that walk row words, a merge that is redone: rare) */
static int step_list_host(rb3gpu_t *h, int64_t len, const uint64_t *d_tw, int64_t n_strings, int64_t step, int64_t *nw, rb3gpu_walker_t **out)
{
	const int64_t K = len / step, n = K + n_strings;
	int r;
	*out = nullptr, *nw = 0;
	if ((r = buf_ensure(h, h->wl, (size_t)n * 40)) < 0 || (r = buf_ensure(h, h->xbuf, (size_t)n_strings * 8)) < 0) return r;
	step_list_launch(h, len, d_tw, n_strings, step, (int64_t*)h->xbuf.p, (Walker*)h->wl.p);
	rb3gpu_walker_t *w = (rb3gpu_walker_t*)malloc((size_t)n * sizeof(rb3gpu_walker_t));
	if (!w) return RB3GPU_ENOMEM;
	if (hipMemcpyAsync(w, h->wl.p, (size_t)n * 32, hipMemcpyDeviceToHost, h->st) != hipSuccess || hipStreamSynchronize(h->st) != hipSuccess) { free(w); return RB3GPU_ENODEV; }
	int64_t m = 0;
	for (int64_t i = 0; i < n; ++i) if (w[i].row >= 0) w[m++] = w[i];
	*nw = m, *out = w;
	return 0;
}

static int merge_core(rb3gpu_t *h, int64_t len, const uint8_t *d_b2, int commit, int64_t *host_pos, int64_t *host_acc2, int rank_only,
		int64_t n_walkers, const rb3gpu_walker_t *walkers, const uint64_t *d_tw, int thin)
{
	const int64_t ntot = h->n + len, nwin = (ntot >> RB3_WIN_BITS) + 1;
	int tent = h->tn.tent;
	if (h->n <= 0 || h->grp == nullptr) return RB3GPU_ESTATE;
	garbage_collect(h, false); // (nothing of this merge is queued yet)
	// no list but a count: one walker per string (n_walkers = number of strings), made on the device
	// ... or a count and a spacing (rb3gpu_merge_text_step_dev): one walker per string and one every `wstep` text positions inside the strings, made on the device
	const int64_t wstep = !walkers && d_tw && n_walkers > 0 && h->mg_step >= 2 ? h->mg_step : 0, n_strings = n_walkers;
	const bool step_list = wstep > 0;
	const bool per_string = !walkers && n_walkers > 0 && !step_list;
	if (d_tw && !walkers && !per_string && !step_list) return RB3GPU_EINVAL;
	if (step_list) n_walkers = len / wstep + n_strings; // slots of the list (some stay empty)
	// neither (the reference's signature: BWT only): the list is made on the device from a sparse LF walk of the batch itself
	const bool auto_list = !walkers && !per_string && !d_tw && h->tn.b2_split != 0 && h->opt.split_log2 == 0 && len >= 4096;
	// Splitters every 2^S rows.  The ranking of the splitters is pointer jumping over all of them (13 rounds for 21 M): a small batch
	// is bound by the latency of the walks between splitters (S = 4: ~210 dependent steps at most), a whole index merged into another
	// by the volume of the jumping rounds (334 M symbols: list 18.4 ms of a 45 ms merge at S = 4, 10 ms at S = 6; one genome of 8.8 M:
	// 1.29 ms per merge at S = 4, 1.47 at S = 5 -- fewer splitters leave more windows without a pick, and a wave of walkers lasts as
	// long as its longest one).
	const int b2S = h->tn.b2_split > 0 ? h->tn.b2_split : len < (32LL << 20) ? 4 : len < (128LL << 20) ? 5 : 6;
	// the device-made list: a walker every b2W text positions -- 384, or more where the batch is so large that the events of that
	// many walkers would not fit the stretch table (and `thin` times that after a merge whose table did overflow)
	// How many stretches will the walkers open?  A walker opens one per indexed relative that drops out of its interval, i.e. per
	// relative whose next private variant lies within the walker's segment (mean distance ~1000 symbols between relatives worth
	// the name), plus its first block of ids: walkers x (K (1 - e^(-W/1000)) + 8), K = relatives ~ mean run length of the index,
	// estimated from how many symbols a slot holds.  Where that exceeds the table the walkers are spaced further apart from the
	// start (a whole-index merge at the top of a multi-GPU tree: 670 M rows into 76 relatives) -- a full table is only found
	// out after the walk, and then costs the walk.
	const double kest = h->nslots > 0 ? (double)h->n / ((double)h->nslots * 32.0) : 1.0;
	auto events_at = [&](double W) { return (double)len / W * ((kest < 1.0 ? 1.0 : kest) * (1.0 - exp(-W / 1000.0)) + (double)RB3_TENT_CHUNK); };
	const double tent_room = 0.6 * (double)RB3_TENT_HALF;
	int64_t b2W = RB3_B2_W; // (rb3gpu_walker_step's spacing -- one walker per resident octet -- was measured on this path too: k_chain 0.48 -> 0.68 ms for one genome into one; row words are a random read per step more, and more walkers only add to them)
	while (len / b2W > (1 << 18) || (h->tn.tent && thin == 1 && events_at((double)b2W) > tent_room && b2W < len / 256)) b2W *= 2;
	b2W *= thin;
	const int64_t b2_nbk = len / b2W + 1, b2_m2cap = len / 64 + 1, b2_nspmax = (len >> b2S) + b2_m2cap + 2;
	if (auto_list) n_walkers = b2_nbk + b2_m2cap; // capacity of the list; how many are in use stays on the device
	const int tent_auto = tent;
	if (per_string) tent = 0; // every walker is exact
	// (the entries of a caller's list are checked by k_chain itself, by the octet that takes a walker: a host loop over 40 k entries took ~40 us with the chip idle)
	if (step_list && tent && thin == 1 && n_walkers > 4096) { // (the same estimate for the list made on the device: a wider spacing from the start)
		int t = 1;
		while (t < 64 && events_at((double)wstep * t) > tent_room) t *= 2;
		if (t > 1) {
			h->stt.n_thinned += 1, h->mg_step = wstep * t;
			const int r2 = merge_core(h, len, d_b2, commit, host_pos, host_acc2, rank_only, n_strings, nullptr, d_tw, t);
			h->mg_step = wstep;
			return r2;
		}
	}
	if (walkers && tent && thin == 1 && n_walkers > 4096) { // (the same estimate for a list the caller made: every t-th walker from the start)
		int t = 1;
		while (t < 64 && events_at((double)len / (double)n_walkers * t) > tent_room) t *= 2;
		if (t > 1) {
			int64_t nw2 = 0;
			rb3gpu_walker_t *w2 = thin_walkers(n_walkers, walkers, t, &nw2);
			if (!w2) return RB3GPU_ENOMEM;
			h->stt.n_thinned += 1;
			const int r2 = merge_core(h, len, d_b2, commit, host_pos, host_acc2, rank_only, nw2, w2, d_tw, t);
			free(w2);
			return r2;
		}
	}
	if ((!walkers && !per_string && !auto_list && !step_list) || n_walkers > (1 << 24) || n_walkers > len || ntot >= lim_pos(h) || (size_t)nwin * sizeof(rb3_slot_t) > lim_slot_bytes(h) || h->tn.staged) {
		if (per_string) return merge_staged(h, len, d_b2, commit, host_pos, host_acc2, rank_only, 0, nullptr, tent_auto);
		if (step_list) { // (the staged paths take a list from the host)
			rb3gpu_walker_t *hw = nullptr;
			int64_t nhw = 0;
			int r0 = step_list_host(h, len, d_tw, n_strings, wstep, &nhw, &hw);
			if (r0 == 0) r0 = merge_staged_text(h, len, d_b2, commit, host_pos, host_acc2, rank_only, nhw, hw, d_tw, tent);
			free(hw);
			return r0;
		}
		if (d_tw) return merge_staged_text(h, len, d_b2, commit, host_pos, host_acc2, rank_only, n_walkers, walkers, d_tw, tent);
		return merge_staged(h, len, d_b2, commit, host_pos, host_acc2, rank_only, n_walkers, walkers, tent);
	}
	int r;
	// every allocation first: hipMalloc may synchronise
	const int64_t ngrp_new = (ntot >> RB3_GRP_BITS) + 1;
	if ((r = buf_ensure(h, h->pos, (size_t)len * 8)) < 0) return r;
	if ((r = buf_ensure(h, h->wl, (size_t)n_walkers * 40)) < 0) return r;
	if ((r = buf_ensure(h, h->misc, MISC_WORDS * 8)) < 0) return r;
	if (auto_list && (r = buf_ensure(h, h->xbuf, ((size_t)b2_nspmax * 4 + (size_t)b2_m2cap + 2 + (size_t)b2_nbk) * 8)) < 0) return r;
	if (step_list && (r = buf_ensure(h, h->wls, (size_t)n_strings * 8)) < 0) return r; // the sentinels' text positions (k_wl_sentinels)
	rb3_stretch_t *tab = nullptr;
	int32_t *sfin = nullptr;
#ifdef RB3GPU_TEST_HOOKS
	const uint32_t sid_limit = h->tn.tent_limit >= 0 ? (uint32_t)h->tn.tent_limit : 0xFFFFFFFFu; // test hook: a tiny stretch table
#else
	const uint32_t sid_limit = 0xFFFFFFFFu;
#endif
	FillJobs jb;
	memset(&jb, 0, sizeof(jb));
	if (tent && (r = tent_prepare(h, &tab, &sfin, &jb)) < 0) return r;
	// width of the drop-out masks: 256 bits inside the stretch records, or 512 / 1024 / 2048 in an array of their own once the
	// walkers of an earlier merge have met intervals wider than the masks (see below); if that array cannot be had, 256 it is
	int tq = !tent ? 1 : h->tn.tent_q > 0 ? h->tn.tent_q : h->tent_q;
	uint32_t *mx = nullptr;
	if (tq > 1 && tent_masks(h, tq, &mx) < 0) tq = 1, mx = nullptr;
	if (!rank_only && (r = ib_ensure(h, 1 - h->cur, ngrp_new, slot_estimate(h, len, ntot))) < 0) return r;
	const bool rows_fused = !rank_only && use_winpar(h, nwin);
	if (rows_fused && (r = buf_ensure(h, h->jg, (size_t)(nwin + 1) * 8)) < 0) return r;
	// Records in text order (k_chain, trec): needs the batch's suffix array for the gather back into row order, which the validation
	// pass does on its way (one random 8-byte read per row instead of one random 8-byte write per step: reads are what the memory
	// system is good at).  By default where the index does not sit in the caches (slots > 192 MB): there the record stores are
	// two thirds of the walk; on a cache-resident index the gather costs what the stores cost.
	const uint32_t *d_sa = h->mg_sa;
	const bool trec = d_sa != nullptr && d_tw != nullptr && rows_fused && !auto_list && len < (1LL << 32) &&
		(h->tn.trec > 0 || (h->tn.trec < 0 && (size_t)h->nslots * sizeof(rb3_slot_t) > ((size_t)192 << 20)));
	if (trec && (r = buf_ensure(h, h->post, (size_t)len * 8)) < 0) return r;
	// ... and from there to pos[] by a partition instead of a gather where the batch is large (rb3gpu_part.h): buckets of 2^K rows
	int partK = 19;
	while ((len >> partK) + 1 > RB3_PART_MAXB) ++partK;
	const int part_nb = (int)((len + (1LL << partK) - 1) >> partK);
	const bool part = trec && h->tn.part != 0 && (h->tn.part > 1 || len >= (1LL << 24)) && ntot < (int64_t)RB3_PART_INVALID(partK);
	if (part && (r = buf_ensure(h, h->xbuf, (size_t)len * 8 + (size_t)RB3_PART_MAXB * 4 + 64)) < 0) return r;
	unsigned long long *misc = (unsigned long long*)h->misc.p;
	Walker *dwl = (Walker*)h->wl.p;
	uint32_t *sidctr = (uint32_t*)(misc + 5);
	uint32_t *mctr = (uint32_t*)(misc + MISC_MCTR); // (cleared with the rest below)
	h->mg_active = 0;
	HIPCHK(hipEventRecord(h->ev[0], h->st));
	// one launch clears what this merge accumulates into: the counters (the scan totals behind them are written later), the
	// stretches the merge before opened, the rows-per-window table of the rebuild, and -- text-order walk -- the row records
	fill_add(&jb, misc, 128, 0u);
	fill_add(&jb, misc + MISC_RG_OVER, tent ? (size_t)(MISC_WORDS - MISC_RG_OVER) * 8 : (size_t)(MISC_BAD_WALKERS + 1 - MISC_RG_OVER) * 8, 0u); // (... MISC_WIDE, MISC_BAD_WALKERS, and the id counters behind them)
	// (the rows-per-window table is not cleared: k_pos_finalize_check_rows writes every entry when pos[] validates, and when it does not the
	// validation counters make every rebuild kernel return before it reads the table -- 42 MB of fill per round of a 1.3 G-symbol build)
	const bool rows_filled = d_tw != nullptr && jb.n < 8;
	if (rows_filled) fill_add(&jb, trec ? h->post.p : h->pos.p, (size_t)len * 8, 0xFFFFFFFFu);
	else if (trec) HIPCHK(hipMemsetAsync(h->post.p, 0xff, (size_t)len * 8, h->st));
	const bool wl_cleared = step_list && jb.n < 8; // the empty slots of the list made on the device: with this launch too (a memset of its own was 8 us in front of the walkers)
	if (wl_cleared) fill_add(&jb, h->wl.p, (size_t)n_walkers * 32, 0xFFFFFFFFu);
	fill_launch(h, jb);
	h->reb_prepared = true; // (build_index: the counters of the run-space rebuild are clear)
	// The histogram of the batch and its scan (the C array of B2 and the rows before every tile, fm-index.c:206-216): a text-order walk
	// does not read them -- the validation behind it and the host do --, so they run on the side stream, beside the walkers.
	const bool lf_beside = rows_filled && !auto_list;
	if (lf_beside) { // (every allocation before the walkers are launched: hipMalloc may synchronise)
		const int64_t ntile = (len + RB3_TILE - 1) / RB3_TILE;
		if ((r = buf_ensure(h, h->ctot2, (size_t)((ntile + RB3_SCAN_CHUNK - 1) / RB3_SCAN_CHUNK + 1) * 64)) < 0) return r;
		if ((r = buf_ensure(h, h->tcnt, (size_t)ntile * 32)) < 0) return r;
		if ((r = buf_ensure(h, h->tpre, (size_t)ntile * 64)) < 0) return r;
	} else if ((r = lf_build(h, len, d_b2, (int64_t*)h->pos.p, nullptr, d_tw == nullptr, rows_filled, false)) < 0) return r;
	// (lf_beside: the histogram kernels are queued BEHIND the walkers' launch, further down -- the host needs ~30 us for their five calls, and with
	// them and the check of the caller's walker list, now done by k_chain itself, in front of it the walkers started ~85 us after the fill kernel)
	if (!lf_beside) HIPCHK(hipEventRecord(h->ev[1], h->st)); // (lf_beside: the rank phase starts where k_chain's own event, ev[6], is recorded)
#ifdef RB3_PROF_STEP
	HIPCHK(hipMemsetAsync(misc + 34, 0, 40, h->st));
#endif
	unsigned long long *b2_nwalk = nullptr; // device-side length of a device-made list
	if (auto_list) {
		// scratch: two link tables (2 words per splitter), string lengths / offsets, one word per window
		uint64_t *lnk[2] = { (uint64_t*)h->xbuf.p, (uint64_t*)h->xbuf.p + 2 * b2_nspmax };
		uint64_t *slen = lnk[1] + 2 * b2_nspmax;
		unsigned long long *bucket = (unsigned long long*)(slen + b2_m2cap + 2);
		const uint64_t *tot2 = (const uint64_t*)(misc + MISC_LF_TOT);
		unsigned long long *mode = misc + MISC_B2_MODE;
		b2_nwalk = misc + 13;
		HIPCHK(hipMemsetAsync(bucket, 0xff, (size_t)b2_nbk * 8, h->st));
		HIPCHK(hipMemsetAsync(slen, 0, (size_t)(b2_m2cap + 2) * 8, h->st));
		hipLaunchKernelGGL(k_b2_mode, dim3(1), dim3(64), 0, h->st, tot2, len, b2_m2cap, mode, b2W);
		hipLaunchKernelGGL(k_b2_walk, dim3((unsigned)((b2_nspmax + 255) / 256)), dim3(256), 0, h->st, (const int64_t*)h->pos.p, len, tot2, b2S, (const unsigned long long*)mode, lnk[0]);
		int cur = 0;
		for (int64_t reach = 1; reach < b2_nspmax; reach <<= 2, cur ^= 1) // (on the upper bound of the splitter count; entries behind the real count are never read)
			hipLaunchKernelGGL(k_b2_jump4, dim3((unsigned)((b2_nspmax + 255) / 256)), dim3(256), 0, h->st, b2_nspmax, (const uint64_t*)lnk[cur], lnk[cur ^ 1]);
		hipLaunchKernelGGL(k_b2_strings, dim3(64), dim3(256), 0, h->st, tot2, (const unsigned long long*)mode, (const uint64_t*)lnk[cur], slen);
		hipLaunchKernelGGL(k_b2_scan, dim3(1), dim3(1024), 0, h->st, tot2, (const unsigned long long*)mode, slen);
		hipLaunchKernelGGL(k_b2_pick, dim3((unsigned)((b2_nspmax + 255) / 256)), dim3(256), 0, h->st, len, tot2, b2S, (const unsigned long long*)mode, (const uint64_t*)lnk[cur], (const uint64_t*)slen, bucket, b2W);
		hipLaunchKernelGGL(k_b2_list, dim3((unsigned)((n_walkers + 255) / 256 < 2048 ? (n_walkers + 255) / 256 : 2048)), dim3(256), 0, h->st, len, tot2, b2S, (const unsigned long long*)mode,
				(const uint64_t*)lnk[cur], (const uint64_t*)slen, (const unsigned long long*)bucket, b2_nbk, (Walker*)h->wl.p, b2_nwalk, (const int64_t*)h->pos.p, b2W);
	} else if (step_list) { // one walker per string and one every wstep text positions, made here (two small kernels in front of the walkers; on the side stream,
		// beside the fill, they were measured in round 5: the wait for the event costs more than they take -- fill-to-walkers 6.0 -> 7.0 ms per 152-genome build)
		step_list_launch(h, len, d_tw, n_strings, wstep, (int64_t*)h->wls.p, (Walker*)h->wl.p, nullptr, wl_cleared);
	} else if (per_string && d_tw) {
		HIPCHK(hipMemsetAsync(h->wl.p, 0xff, (size_t)n_walkers * 32, h->st)); // a walker that nobody fills in starts at row -1: caught below
		hipLaunchKernelGGL(k_walkers_per_string, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, h->st, (Walker*)h->wl.p, n_walkers, d_tw, len);
	} else if (per_string) { // row words: the sentinels are rows 0 .. n_walkers-1 (the count is checked against the batch below)
		hipLaunchKernelGGL(k_walkers_sentinel_rows, dim3((unsigned)((n_walkers + 255) / 256)), dim3(256), 0, h->st, (Walker*)h->wl.p, n_walkers);
	} else { // walker list: through the pinned staging buffer when it fits (a pageable source is staged by the runtime, slowly)
		const size_t wb = (size_t)n_walkers * 32;
		if (h->stage[0] == nullptr)
			for (int i = 0; i < 2; ++i)
				if (hipHostMalloc((void**)&h->stage[i], RB3_STAGE_BYTES, hipHostMallocDefault) != hipSuccess) { h->stage[i] = nullptr; break; }
		// The list goes over the side stream, beside the LF kernels queued above (on the main stream its DMA sat between them and
		// the walkers: ~45 us per merge with the chip idle); the host copies it into pinned memory while those kernels run.
		const void *src = walkers;
		if (is_pinned(walkers, wb) && !h->tn.copy_walkers) {
			// A list in page-locked memory (rb3gpu_pinned_alloc) is read where it lies: every entry is fetched once, by the octet
			// that takes the walker (~2 us over PCIe, once or twice per wave), and the ~40 us the walkers waited for the copy engine
			// to deliver the list -- with the chip idle -- are gone.
			dwl = (Walker*)walkers;
		} else {
			if (!is_pinned(walkers, wb) && h->stage[0] && wb <= RB3_STAGE_BYTES) {
				memcpy(h->stage[0], walkers, wb); // safe to reuse: every earlier copy out of it was synchronised
				src = h->stage[0];
			}
			HIPCHK(hipMemcpyAsync(dwl, src, wb, hipMemcpyHostToDevice, h->st2));
			HIPCHK(hipEventRecord(h->evx[1], h->st2));
			HIPCHK(hipStreamWaitEvent(h->st, h->evx[1], 0));
		}
	}
	int64_t *dpos = (int64_t*)h->pos.p;
	int64_t *drec = trec ? (int64_t*)h->post.p : dpos; // where the walkers leave their records
	// one word per walker behind the list's buffer: where it met somebody's record (k_chain: jmet; read by k_junction_check)
	int64_t *jmet = tent && d_tw != nullptr && !auto_list ? (int64_t*)((char*)h->wl.p + (size_t)n_walkers * 32) : nullptr;
	{
		const IdxView iv = view_of(h);
		// lanes per walker: an octet.  (A QUAD per walker -- 16 walkers per wave, every lane two slices of a slot -- existed in rounds 2-5 and was slower in every
		// regime, 147 against 86 ms of k_chain per 152-genome build when last measured (round 6); it went when the run codes became cumulative ends.)
		const int octs = h->tn.octs; // octets per wave that take walkers
		const int64_t n_expect = auto_list ? b2_nbk + 64 : n_walkers; // (a device-made list: capacity >> walkers; size the launch for the walkers)
		int64_t nblk = (n_expect + 4 * octs - 1) / (4 * octs) * h->tn.blkmul;
		// persistent waves: 2048 blocks x 4 waves fill the chip once (256 CUs x 32); more blocks only queue behind them (measured: 10 % slower at 4096)
		{ const int64_t cap = h->tn.blkcap; nblk = nblk > cap ? cap : nblk < 1 ? 1 : nblk; }
#ifdef RB3_PROF
		fprintf(stderr, "[prof] launching %lld blocks x 256 threads, %d octets per wave, %lld walkers\n", (long long)nblk, octs, (long long)n_walkers);
#endif
		const dim3 grid((unsigned)(nblk * (256 / h->tn.chain_bs))), blk((unsigned)h->tn.chain_bs);
		HIPCHK(hipEventRecord(h->ev[6], h->st)); // (measured, round 6: without this event a merge is ~5 us shorter -- k_chain's own time is worth that)
#ifdef RB3GPU_TEST_HOOKS
#define RB3_TREC_ARG ((trec ? 1 : 0) | (h->tn.hide_first ? 2 : 0))
#else
#define RB3_TREC_ARG (trec ? 1 : 0)
#endif
#define RB3_LAUNCH_FAST1(D, T, X, W) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain<true, D, T, X, W>), grid, blk, 0, h->st, iv, drec, len, (int64_t)0, per_string ? -1 : 0, \
			(const Walker*)dwl, n_walkers, (int64_t)-1, (int64_t*)nullptr, misc, misc + 1, octs, tab, sidctr, sid_limit, d_tw, (const unsigned long long*)b2_nwalk, 256 * tq - 1, mctr, RB3_TREC_ARG, jmet)
#define RB3_LAUNCH_FAST(D, T, X) RB3_LAUNCH_FAST1(D, T, X, 8)
		// text-order words: per-lane loads while the L1 can hold a line per walker (256 CUs x 32 waves x 8 octets), else 64-byte fetches
		int text_mode = !d_tw ? 0 : n_walkers <= 65536 ? 1 : 2;
#ifdef RB3GPU_TEST_HOOKS
		if (d_tw && h->tn.text_mode) text_mode = h->tn.text_mode;
#endif
		switch ((iv.dense == 2 ? 12 : 0) + (tent ? 6 : 0) + text_mode) {
		case 14: RB3_LAUNCH_FAST(true, false, 2); break;
		case 13: RB3_LAUNCH_FAST(true, false, 1); break;
		case 12: RB3_LAUNCH_FAST(true, false, 0); break;
		case 20: RB3_LAUNCH_FAST(true, true, 2); break;
		case 19: RB3_LAUNCH_FAST(true, true, 1); break;
		case 18: RB3_LAUNCH_FAST(true, true, 0); break;
		case 2: RB3_LAUNCH_FAST(false, false, 2); break;
		case 1: RB3_LAUNCH_FAST(false, false, 1); break;
		case 0: RB3_LAUNCH_FAST(false, false, 0); break;
		case 8: RB3_LAUNCH_FAST(false, true, 2); break;
		case 7: // (the headline's kernel: 32-bit positions in the common step where index and batch allow it)
			if (iv.abs == 1 && iv.n < (1LL << 32) - (1LL << 20) && len < (1LL << 29))
#ifdef RB3_PROF_STEP

#endif
#ifdef RB3_PROF_STEP /* kernel experiment: RB3_EXP_DYNLDS bytes of unused LDS per block bound the waves per SIMD (how long is a step of a wave that has its SIMD to itself?) */
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain<true, false, true, 1, 8, true>), grid, blk, getenv("RB3_EXP_DYNLDS") ? (unsigned)atoi(getenv("RB3_EXP_DYNLDS")) : 0u, h->st, iv, drec, len, (int64_t)0, per_string ? -1 : 0,
#else
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_chain<true, false, true, 1, 8, true>), grid, blk, 0, h->st, iv, drec, len, (int64_t)0, per_string ? -1 : 0,
#endif
					(const Walker*)dwl, n_walkers, (int64_t)-1, (int64_t*)nullptr, misc, misc + 1, octs, tab, sidctr, sid_limit, d_tw, (const unsigned long long*)b2_nwalk, 256 * tq - 1, mctr, RB3_TREC_ARG, jmet);
			else RB3_LAUNCH_FAST(false, true, 1);
			break;
		default: RB3_LAUNCH_FAST(false, true, 0); break;
		}
#undef RB3_LAUNCH_FAST
#undef RB3_LAUNCH_FAST1
		HIPCHK(hipEventRecord(h->ev[7], h->st));
		if (lf_beside) {
			if (h->tn.lf_after) HIPCHK(hipStreamWaitEvent(h->st2, h->ev[7], 0)); // beside the settle kernels, not beside the walkers' first steps
			if ((r = lf_build(h, len, d_b2, (int64_t*)h->pos.p, nullptr, false, true, true)) < 0) return r;
			HIPCHK(hipEventRecord(h->evx[2], h->st2));
		}
		if (tent) {
			launch_settle(h, iv, tab, mx, (const uint32_t*)sidctr, sfin, misc + 2, (int)RB3_RESW_MAXHOPS, tq, (const uint32_t*)mctr), h->stt.tent_mask_bits = 256 * tq;
#ifdef RB3GPU_TEST_HOOKS
			if (h->tn.corrupt_sfin >= 0) hipLaunchKernelGGL(k_test_corrupt_sfin, dim3(1), dim3(64), 0, h->st, (const rb3_stretch_t*)tab, (const uint32_t*)sidctr, sfin, h->tn.corrupt_sfin, (long long*)(misc + 42));
#endif
		}
		if (rows_fused) { // validation and the rows-per-window table of the rebuild in one pass over pos[]
			const dim3 g1((unsigned)((len + 1 + 255) / 256));
			const int64_t *frec = trec ? (const int64_t*)drec : nullptr;
			const uint32_t *fsa = trec ? d_sa : nullptr;
			if (part) { // the permutation as two streaming passes; the validation below then reads pos[] in place
				uint64_t *pout = (uint64_t*)h->xbuf.p;
				unsigned int *pcur = (unsigned int*)(pout + len);
				const int S = 256; // (blocks per bucket: enough to fill an XCD, so that an XCD works on ONE window of 2^K rows at a time and its L2 holds it)
				const int64_t ntile = (len + RB3_PART_TILE - 1) / RB3_PART_TILE;
				hipLaunchKernelGGL(k_part_init, dim3((unsigned)((part_nb + 255) / 256)), dim3(256), 0, h->st, pcur, part_nb, partK);
				if (tent) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_part_scatter<true>), dim3((unsigned)(ntile < 4096 ? ntile : 4096)), dim3(RB3_PART_THREADS), 0, h->st, (const int64_t*)drec, d_tw, len, partK, part_nb, (const int32_t*)sfin, misc + 2, pcur, pout);
				else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_part_scatter<false>), dim3((unsigned)(ntile < 4096 ? ntile : 4096)), dim3(RB3_PART_THREADS), 0, h->st, (const int64_t*)drec, d_tw, len, partK, part_nb, (const int32_t*)nullptr, misc + 2, pcur, pout);
				hipLaunchKernelGGL(k_part_place, dim3((unsigned)((part_nb + 7) / 8 * 8 * S)), dim3(256), 0, h->st, (const uint64_t*)pout, len, partK, part_nb, S, dpos, (const unsigned long long*)(misc + 2));
				frec = nullptr, fsa = nullptr;
			}
			const dim3 g2((unsigned)((len / RB3_FIN_ROWS + 1 + 255) / 256)); // (records in row order: several rows per thread)
			if (frec == nullptr && tent) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pos_finalize_check_rowsN<true, RB3_FIN_ROWS>), g2, dim3(256), 0, h->st, dpos, len, ntot, (const int32_t*)sfin, misc + 2, (int64_t*)h->jg.p, nwin);
			else if (frec == nullptr) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pos_finalize_check_rowsN<false, RB3_FIN_ROWS>), g2, dim3(256), 0, h->st, dpos, len, ntot, (const int32_t*)nullptr, misc + 2, (int64_t*)h->jg.p, nwin);
			else if (tent) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pos_finalize_check_rows<true>), g1, dim3(256), 0, h->st, dpos, len, ntot, (const int32_t*)sfin, misc + 2, (int64_t*)h->jg.p, nwin, frec, fsa);
			else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pos_finalize_check_rows<false>), g1, dim3(256), 0, h->st, dpos, len, ntot, (const int32_t*)nullptr, misc + 2, (int64_t*)h->jg.p, nwin, frec, fsa);
		} else if (tent)
			hipLaunchKernelGGL(k_pos_finalize_check, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, h->st, dpos, len, ntot, (const int32_t*)sfin, misc + 2);
		else
			hipLaunchKernelGGL(k_pos_check, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, h->st, (const int64_t*)dpos, len, ntot, misc + 2);
	}
#ifdef RB3GPU_TEST_HOOKS
	if (h->tn.corrupt_pos) hipLaunchKernelGGL(k_test_corrupt, dim3((unsigned)(len / 6 / 256 + 1)), dim3(256), 0, h->st, dpos, len);
#endif
	HIPCHK(hipEventRecord(h->ev[2], h->st));
	h->lf_wait = lf_beside; // (the batch's totals, side stream: awaited where they are first read -- k_scan_place, or below before the counters go to the host)
	const JuncArgs ja = { d_tw, tab, (const uint32_t*)sidctr, jmet, n_walkers, (const unsigned long long*)nullptr };
	const bool lf_side = launch_lf_check(h, (const int64_t*)dpos, d_b2, len, true, h->ev[2], jmet ? &ja : nullptr);
	int64_t ngrp = 0, nslots = 0, acc[7];
	if (!rank_only && (r = build_index<false>(h, len, d_b2, (const int64_t*)dpos, ntot, true, &ngrp, &nslots, acc, rows_fused)) < 0) { h->lf_wait = false; return r; }
	if (h->lf_wait) { HIPCHK(hipStreamWaitEvent(h->st, h->evx[2], 0)); h->lf_wait = false; }
	HIPCHK(hipEventRecord(h->ev[3], h->st));
	if (lf_side) HIPCHK(hipStreamWaitEvent(h->st, h->evx[1], 0));
	unsigned long long hm[48];
	// The only synchronisation of the merge.  The counters land in page-locked memory: a copy into pageable memory goes through a
	// staging buffer of the runtime, ~10 us more per merge (151 merges per build).
	unsigned long long *hml = h->hm_pin ? h->hm_pin : hm;
	HIPCHK(hipMemcpyAsync(hml, misc, sizeof(hm), hipMemcpyDeviceToHost, h->st));
	if (host_pos) HIPCHK(hipMemcpyAsync(host_pos, dpos, (size_t)len * 8, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	if (hml != hm) memcpy(hm, hml, sizeof(hm));
#ifdef RB3_DEBUG_PART /* kernel experiment: the partition's pos[] against the gather it replaces */
	if (part && len < (1LL << 26)) {
		std::vector<int64_t> hp((size_t)len), hr((size_t)len);
		std::vector<uint32_t> hs((size_t)len);
		std::vector<uint64_t> ht((size_t)len);
		(void)hipMemcpy(hp.data(), dpos, (size_t)len * 8, hipMemcpyDeviceToHost);
		(void)hipMemcpy(hr.data(), drec, (size_t)len * 8, hipMemcpyDeviceToHost);
		(void)hipMemcpy(hs.data(), d_sa, (size_t)len * 4, hipMemcpyDeviceToHost);
		(void)hipMemcpy(ht.data(), d_tw, (size_t)len * 8, hipMemcpyDeviceToHost);
		int64_t nbad = 0, nperm = 0;
		for (int64_t i = 0; i < len; ++i) {
			if ((int64_t)(ht[hs[i]] >> 3) != i) ++nperm;
			const int64_t want = hr[hs[i]];
			if (!(want & RB3_TENT) && want >= 0 && hp[i] != want && nbad++ < 10)
				fprintf(stderr, "[debug part] row %lld: pos %lld, rec[sa] %lld (text position %u, tw row %lld)\n", (long long)i, (long long)hp[i], (long long)want, hs[i], (long long)(ht[hs[i]] >> 3));
		}
		fprintf(stderr, "[debug part] len %lld K %d nb %d: %lld rows differ from the gather, %lld rows with isa[sa[i]] != i\n", (long long)len, partK, part_nb, (long long)nbad, (long long)nperm);
	}
#endif
	hipEvent_t ev_rank0 = lf_beside ? h->ev[6] : h->ev[1];
	h->stt.ms_lf += ev_ms(h->ev[0], ev_rank0);
	h->stt.ms_chain += ev_ms(h->ev[6], h->ev[7]);
	h->stt.ms_rank += ev_ms(ev_rank0, h->ev[2]);
	h->stt.ms_build += ev_ms(h->ev[2], h->ev[3]);
	h->stt.n_rank_launches += 1, h->stt.n_rounds += 1;
	guard_check(h, __func__);
	h->stt.n_lf_steps += (int64_t)hm[1];
	h->stt.n_junctions_checked += jmet ? (int64_t)hm[MISC_JUNC] : 0;
	h->stt.n_lf_checked += (int64_t)hm[MISC_LF_CHK];
#ifdef RB3_PROF_STEP
	if (hm[37]) fprintf(stderr, "[prof] common step x %llu (lane 0 of every wave): directory %.0f cycles, slot %.0f, decode+rest %.0f, between steps %.0f; %.1f %% of these steps ran the two-decode side for some walker; directory wait of the steps behind a flush of records: %.0f cycles\n", hm[37], (double)hm[34] / hm[37], (double)hm[35] / hm[37], (double)hm[36] / hm[37], (double)hm[38] / hm[37], 100.0 * hm[9] / hm[37], 8.0 * hm[10] / hm[37]);
	if (hm[37] && (hm[11] >> 40)) fprintf(stderr, "[prof] steps behind a flush of records x %llu: directory %.0f cycles, slot %.0f\n", hm[11] >> 40, (double)(hm[11] & ((1ull << 40) - 1)) / (double)(hm[11] >> 40), (double)hm[12] / (double)(hm[11] >> 40));
	if (hm[37]) fprintf(stderr, "[prof] general steps x %llu (lane 0 of every wave): %.0f ticks each; common steps %.0f ticks each (s_memtime ticks, whole step incl. the time between steps): general steps are %.1f %% of the iterations and %.1f %% of the stepping time\n",
		hm[10] >> 32, (double)(hm[10] & 0xFFFFFFFFull) / (double)((hm[10] >> 32) ? (hm[10] >> 32) : 1), (double)(hm[34] + hm[35] + hm[36] + hm[38]) / hm[37],
		100.0 * (double)(hm[10] >> 32) / (double)((hm[10] >> 32) + hm[37]), 100.0 * (double)(hm[10] & 0xFFFFFFFFull) / (double)((hm[10] & 0xFFFFFFFFull) + hm[34] + hm[35] + hm[36] + hm[38]));
#endif
#ifdef RB3_PROF_WAVES /* kernel experiment: the waves of this launch, by where they ran (merge number RB3_PROF_WAVES_AT of the handle, default 140) */
	{
		static int at = getenv("RB3_PROF_WAVES_AT") ? atoi(getenv("RB3_PROF_WAVES_AT")) : 140;
		unsigned long long nw = 0, zero = 0;
		(void)hipMemcpyFromSymbol(&nw, HIP_SYMBOL(g_prof_wave_n), 8);
		static int printed = 0;
		if ((int)(h->stt.n_rounds % 151) == at % 151 && nw > 0 && printed++ < 2) {
			if (nw > RB3_PROF_WAVES_MAX) nw = RB3_PROF_WAVES_MAX;
			std::vector<unsigned long long> w((size_t)nw * 4);
			(void)hipMemcpyFromSymbol(w.data(), HIP_SYMBOL(g_prof_wave), (size_t)nw * 32);
			unsigned long long t0 = ~0ull, t1 = 0;
			for (size_t i = 0; i < nw; ++i) { if (w[4 * i] < t0) t0 = w[4 * i]; if (w[4 * i + 1] > t1) t1 = w[4 * i + 1]; }
			// where: xcc (8) x se (4?) x cu (16) x simd (4)
			std::map<unsigned, std::vector<size_t>> by_cu, by_simd;
			for (size_t i = 0; i < nw; ++i) {
				const unsigned hw = (unsigned)w[4 * i + 2], xcc = (unsigned)(w[4 * i + 2] >> 32) & 15u;
				const unsigned simd = hw >> 4 & 3u, cu = hw >> 8 & 15u, sh = hw >> 12 & 1u, se = hw >> 13 & 7u;
				const unsigned cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu;
				by_cu[cuid].push_back(i), by_simd[cuid * 4 + simd].push_back(i);
			}
			double dsum[16] = {0}; long cnt[16] = {0};
			for (auto &kv : by_simd) {
				const size_t k = kv.second.size() < 15 ? kv.second.size() : 15;
				for (size_t i : kv.second) dsum[k] += (double)(w[4 * i + 1] - w[4 * i]), ++cnt[k];
			}
			(void)t1; // (s_memtime counts from another base on every XCD: only a wave's own duration means something)
			fprintf(stderr, "[prof waves] merge %d: %llu waves on %zu compute units / %zu SIMDs (durations in ticks of s_memtime)\n", at, nw, by_cu.size(), by_simd.size());
			for (int k = 1; k < 16; ++k) if (cnt[k]) fprintf(stderr, "[prof waves]   SIMDs with %d waves: %ld waves, last %.0f ticks on average\n", k, cnt[k], dsum[k] / cnt[k]);
			double cd[64] = {0}; long cc[64] = {0};
			for (auto &kv : by_cu) {
				const size_t k = kv.second.size() < 63 ? kv.second.size() : 63;
				for (size_t i : kv.second) cd[k] += (double)(w[4 * i + 1] - w[4 * i]), ++cc[k];
			}
			for (int k = 1; k < 64; ++k) if (cc[k]) fprintf(stderr, "[prof waves]   compute units with %d waves: %ld waves, last %.0f ticks on average\n", k, cc[k], cd[k] / cc[k]);
			double xd[16] = {0}; long xc[16] = {0};
			for (size_t i = 0; i < nw; ++i) { const unsigned x = (unsigned)(w[4 * i + 2] >> 32) & 15u; xd[x] += (double)(w[4 * i + 1] - w[4 * i]), ++xc[x]; }
			for (int x = 0; x < 16; ++x) if (xc[x]) fprintf(stderr, "[prof waves]   XCD %d: %ld waves, last %.0f ticks on average\n", x, xc[x], xd[x] / xc[x]);
			std::vector<double> durs;
			for (size_t i = 0; i < nw; ++i) durs.push_back((double)(w[4 * i + 1] - w[4 * i]));
			std::sort(durs.begin(), durs.end());
			fprintf(stderr, "[prof waves]   durations: 1 %% %.0f, 10 %% %.0f, 50 %% %.0f, 90 %% %.0f, 99 %% %.0f, longest %.0f; iterations of the first wave: %llu\n", durs[nw / 100], durs[nw / 10], durs[nw / 2], durs[nw * 9 / 10], durs[nw * 99 / 100], durs[nw - 1], w[3] & 0xFFFFFFFFull);
		}
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof_wave_n), &zero, 8);
	}
#endif
#ifdef RB3_PROF
	fprintf(stderr, "[prof] k_chain waves %llu: max %.0f cycles, mean %.0f cycles, mean iterations %.1f, %.1f %% of them with the two-decode path -> %.1f cycles/iteration\n", hm[11], (double)hm[8], (double)hm[9] / hm[11], (double)hm[10] / hm[11], 100.0 * (double)hm[12] / (double)hm[10], (double)hm[9] / hm[10]);
#endif
	if (hm[MISC_LF_TOT + 6] != 0) return RB3GPU_ESYMBOL; // fm-index.c:124-125
	int64_t acc2[7];
	acc2[0] = 0;
	for (int a = 0; a < 6; ++a) acc2[a + 1] = acc2[a] + (int64_t)hm[MISC_LF_TOT + a];
	if (acc2[1] <= 0) return RB3GPU_EINVAL; // a batch always ends with a sentinel
	if (per_string && acc2[1] != n_walkers) return RB3GPU_EINVAL; // not the number of strings of this batch (nothing was installed)
	if (step_list && acc2[1] != n_strings) return RB3GPU_EINVAL;
	if (hm[MISC_BAD_WALKERS] != 0) return RB3GPU_EINVAL; // not a walker list for this batch (k_chain skipped those entries; nothing was installed)
#ifdef RB3_DEBUG_B2
	if (auto_list) { // kernel experiment: what the device-made list looks like
		const int64_t nw = (int64_t)hm[13];
		rb3gpu_walker_t *w = (rb3gpu_walker_t*)malloc((size_t)nw * 32);
		(void)hipMemcpy(w, h->wl.p, (size_t)nw * 32, hipMemcpyDeviceToHost);
		int64_t nv = 0, ninf = 0, mx = 0, mn = INT64_MAX, sum = 0, nbig = 0;
		for (int64_t i = 0; i < nw; ++i) {
			if (w[i].row < 0) continue;
			++nv;
			if (w[i].nsteps > (1LL << 60)) { ++ninf; continue; }
			sum += w[i].nsteps, mx = w[i].nsteps > mx ? w[i].nsteps : mx, mn = w[i].nsteps < mn ? w[i].nsteps : mn, nbig += w[i].nsteps > 600;
		}
		fprintf(stderr, "[debug] device-made list: %lld slots, %lld walkers, %lld with nsteps = inf, gaps min %lld mean %.1f max %lld, %lld above 600; mode %llu\n", (long long)nw, (long long)nv, (long long)ninf,
				(long long)mn, (double)sum / (double)(nv - ninf > 0 ? nv - ninf : 1), (long long)mx, (long long)nbig, hm[MISC_B2_MODE]);
		free(w);
	}
#endif
	if (auto_list && hm[MISC_B2_MODE] == 2) // more strings than the device-made list takes (strings shorter than 64 symbols on average): nothing was walked
		return merge_staged(h, len, d_b2, commit, host_pos, host_acc2, rank_only, 0, nullptr, tent);
	if (host_acc2) memcpy(host_acc2, acc2, sizeof(acc2));
	if (tent) {
		tent_used(h, hm[5]);
		if (h->opt.verbose >= 4) fprintf(stderr, "[M::rb3gpu] merge of %lld rows into %lld: %lld list slots, %llu LF steps, stretch ids %llu (blocks of a walker with relatives) + %llu (single); fill-to-walkers %.3f ms, k_chain %.3f, settle + validation %.3f, rebuild %.3f\n",
				(long long)len, (long long)h->n, (long long)n_walkers, hm[1], hm[5] & 0xFFFFFFFFull, hm[5] >> 32, ev_ms(h->ev[0], ev_rank0), ev_ms(h->ev[6], h->ev[7]), ev_ms(h->ev[7], h->ev[2]), ev_ms(h->ev[2], h->ev[3]));
		// Walkers that were old enough to record tentatively but sat on an interval wider than the masks (more relatives in the index
		// than mask bits) walked without recording: where that was more than a few percent of all steps the next merges use masks of
		// twice the width (the index only gains relatives).  Nothing is redone: this merge is complete, only slower than it could be.
		if (h->tn.tent_q == 0 && tq < RB3_TENT_QMAX && hm[MISC_WIDE] * 32 > hm[1] && h->tent_q <= tq) {
			h->tent_q = tq * 2;
			if (h->opt.verbose >= 3) fprintf(stderr, "[M::rb3gpu] %.1f %% of the LF steps were walked by walkers whose interval is wider than %d rows; the next merges track up to %d\n",
					100.0 * (double)hm[MISC_WIDE] / (double)(hm[1] ? hm[1] : 1), 256 * tq - 1, 512 * tq - 1);
		}
#ifdef RB3GPU_TEST_HOOKS
		if (h->tn.force_fallback) hm[4] = 1; // test hook: exercise the redo path
#endif
	}
	if (tent && hm[4] != 0 && hm[MISC_LF_CHK + 1] == 0 && !h->tn.resolve_v1
#ifdef RB3GPU_TEST_HOOKS
			&& !h->tn.force_fallback
#endif
			) {
		// Second chance: dependency paths longer than k_resolve_w follows (a string that repeats indexed text).  Nothing was installed
		// and the unsettled records are still in pos[]: settle them by pointer jumping over the walkers, then validate and rebuild again.
		const int64_t na = h->sid_dirty[0], nb = h->sid_dirty[1], nblk = (na + RB3_TENT_BLOCK - 1) / RB3_TENT_BLOCK, nn = nblk + nb;
		const size_t wj_bytes = tq > 1 ? 32 * (size_t)tq + 16 : sizeof(WjNode);
		if (nn > 0 && (r = buf_ensure(h, h->xbuf, (size_t)nn * 2 * wj_bytes)) < 0) return r;
		if (nn > 0) {
			HIPCHK(hipEventRecord(h->ev[4], h->st));
			const dim3 gj((unsigned)((nn + 255) / 256)), bj(256);
#define RB3_WJ_X(Q) do { \
				WjNodeX<Q> *nd[2] = { (WjNodeX<Q>*)h->xbuf.p, (WjNodeX<Q>*)h->xbuf.p + nn }; \
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wj_init_x<Q>), gj, bj, 0, h->st, (const rb3_stretch_t*)tab, (const uint32_t*)mx, (const int32_t*)sfin, nblk, nb, nd[0]); \
				int cur = 0; \
				for (int64_t reach = 1; reach <= nn; reach <<= 1, cur ^= 1) \
					hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wj_round_x<Q>), gj, bj, 0, h->st, nn, (const WjNodeX<Q>*)nd[cur], nd[cur ^ 1]); \
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_wj_apply_x<Q>), gj, bj, 0, h->st, nblk, nb, (const WjNodeX<Q>*)nd[cur], sfin); \
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sfin_x<Q>), dim3(2048), dim3(256), 0, h->st, (const rb3_stretch_t*)tab, (const uint32_t*)mx, (const uint32_t*)sidctr, sfin); \
			} while (0)
			if (tq >= 8) RB3_WJ_X(8);
			else if (tq >= 4) RB3_WJ_X(4);
			else if (tq >= 2) RB3_WJ_X(2);
			else {
			WjNode *nd[2] = { (WjNode*)h->xbuf.p, (WjNode*)h->xbuf.p + nn };
			hipLaunchKernelGGL(k_wj_init, gj, bj, 0, h->st, (const rb3_stretch_t*)tab, (const int32_t*)sfin, nblk, nb, nd[0]);
			int cur = 0;
			for (int64_t reach = 1; reach <= nn; reach <<= 1, cur ^= 1)
				hipLaunchKernelGGL(k_wj_round, gj, bj, 0, h->st, nn, (const WjNode*)nd[cur], nd[cur ^ 1]);
			hipLaunchKernelGGL(k_wj_apply, gj, bj, 0, h->st, nblk, nb, (const WjNode*)nd[cur], sfin);
			hipLaunchKernelGGL(k_sfin, dim3(2048), dim3(256), 0, h->st, (const rb3_stretch_t*)tab, (const uint32_t*)sidctr, sfin);
			}
#undef RB3_WJ_X
			HIPCHK(hipMemsetAsync(misc + 2, 0, 24, h->st));
			if (rows_fused) {
				HIPCHK(hipMemsetAsync(h->jg.p, 0, (size_t)(nwin + 1) * 8, h->st));
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pos_finalize_check_rows<true>), dim3((unsigned)((len + 1 + 255) / 256)), dim3(256), 0, h->st, dpos, len, ntot, (const int32_t*)sfin, misc + 2, (int64_t*)h->jg.p, nwin,
						trec ? (const int64_t*)drec : (const int64_t*)nullptr, trec ? d_sa : (const uint32_t*)nullptr);
			} else
				hipLaunchKernelGGL(k_pos_finalize_check, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, h->st, dpos, len, ntot, (const int32_t*)sfin, misc + 2);
			HIPCHK(hipEventRecord(h->ev[5], h->st));
			(void)launch_lf_check(h, (const int64_t*)dpos, d_b2, len, false, nullptr, jmet ? &ja : nullptr);
			if (!rank_only && (r = build_index<false>(h, len, d_b2, (const int64_t*)dpos, ntot, true, &ngrp, &nslots, acc, rows_fused)) < 0) return r;
			HIPCHK(hipEventRecord(h->ev[3], h->st));
			HIPCHK(hipMemcpyAsync(hm, misc, sizeof(hm), hipMemcpyDeviceToHost, h->st));
			if (host_pos) HIPCHK(hipMemcpyAsync(host_pos, dpos, (size_t)len * 8, hipMemcpyDeviceToHost, h->st));
			HIPCHK(hipStreamSynchronize(h->st));
			h->stt.ms_rank += ev_ms(h->ev[4], h->ev[5]);
			h->stt.ms_build += ev_ms(h->ev[5], h->ev[3]);
			h->stt.n_long_settles += 1;
		}
	}
	const unsigned long long tent_cap_a = sid_limit < (uint32_t)RB3_TENT_HALF ? sid_limit : (uint32_t)RB3_TENT_HALF, tent_cap_b = sid_limit < (uint32_t)(RB3_TENT_POISON - RB3_TENT_HALF) ? sid_limit : (uint32_t)(RB3_TENT_POISON - RB3_TENT_HALF);
	if (tent && hm[4] != 0 && thin < 64 && !per_string && ((hm[5] & 0xFFFFFFFFull) + RB3_TENT_CHUNK > tent_cap_a || (hm[5] >> 32) >= tent_cap_b)) {
		// The stretch table was full (a huge batch whose walkers pass the variants of many indexed relatives: events ~ walkers x
		// relatives): once more with every eighth walker -- the events of a walker stop growing once it has seen every relative
		// drop out, so fewer, longer walkers need fewer stretches -- before giving up on tentative records altogether.
		if (h->opt.verbose >= 2) fprintf(stderr, "[W::rb3gpu] the table of tentative stretches is full (%llu + %llu ids); once more with %d times fewer walkers\n", hm[5] & 0xFFFFFFFFull, hm[5] >> 32, 8);
		h->stt.n_thinned += 1;
		if (walkers) {
			int64_t nw2 = 0;
			rb3gpu_walker_t *w2 = thin_walkers(n_walkers, walkers, 8, &nw2);
			if (!w2) return RB3GPU_ENOMEM;
			r = merge_core(h, len, d_b2, commit, host_pos, host_acc2, rank_only, nw2, w2, d_tw, thin * 8);
			free(w2);
			return r;
		}
		if (step_list) { // the list made on the device: eight times the spacing
			h->mg_step = wstep * 8;
			r = merge_core(h, len, d_b2, commit, host_pos, host_acc2, rank_only, n_strings, nullptr, d_tw, thin * 8);
			h->mg_step = wstep;
			return r;
		}
		return merge_core(h, len, d_b2, commit, host_pos, host_acc2, rank_only, 0, nullptr, d_tw, thin * 8);
	}
	// (the paths below walk row words and take their list from the host: the one made on the device is fetched)
	rb3gpu_walker_t *hw_step = nullptr;
	struct FreeHw { rb3gpu_walker_t **p; ~FreeHw() { free(*p); } } free_hw = { &hw_step };
	if (step_list && tent && (hm[4] != 0 || hm[2] != 0 || hm[3] != 0 || hm[MISC_LF_CHK + 1] != 0)) {
		int64_t nhw = 0;
		if ((r = step_list_host(h, len, d_tw, n_strings, wstep, &nhw, &hw_step)) < 0) return r;
		walkers = hw_step, n_walkers = nhw;
	}
	if (tent && (hm[4] != 0 || hm[2] != 0 || hm[3] != 0)) { // the optimistic pass did not validate (records unsettled, or rows nobody reached): nothing was installed, redo without tentative records
		h->stt.n_fallbacks += 1;
		if (h->opt.verbose >= 2) fprintf(stderr, "[W::rb3gpu] %llu tentative records unsettled, %llu rows unset, %llu out of order; redoing the merge without tentative records\n", hm[4], hm[2], hm[3]);
#ifdef RB3_DEBUG_UNSETTLED /* kernel experiment: what do the records that stayed unsettled look like?  (rows, their stretches and the chains these hang on) */
		if (!trec && len < (1LL << 28)) {
			std::vector<int64_t> hp((size_t)len);
			(void)hipMemcpy(hp.data(), dpos, (size_t)len * 8, hipMemcpyDeviceToHost);
			std::vector<uint64_t> htw;
			if (d_tw) { htw.resize((size_t)len); (void)hipMemcpy(htw.data(), d_tw, (size_t)len * 8, hipMemcpyDeviceToHost); }
			std::vector<int64_t> row2tp;
			if (d_tw) { row2tp.assign((size_t)len, -1); for (int64_t t = 0; t < len; ++t) { const int64_t r = (int64_t)(htw[t] >> 3); if (r >= 0 && r < len) row2tp[r] = t; } }
			auto show = [&](int sid, const char *what) {
				if (sid < 0 || sid >= RB3_TENT_IDS) return;
				rb3_stretch_t rec; int32_t sf = 0;
				(void)hipMemcpy(&rec, tab + sid, sizeof(rec), hipMemcpyDeviceToHost);
				(void)hipMemcpy(&sf, sfin + sid, 4, hipMemcpyDeviceToHost);
				fprintf(stderr, "      %s sid %d: type %llu prev %d lo %lld w1 %llx del %d child %d pad %x %x sfin %d mask %08x %08x\n", what, sid, (unsigned long long)(rec.w0 >> 62), RB3_DEP_PREV(rec.w0),
						(long long)(rec.w0 & (uint64_t)RB3_TENT_MASK), (unsigned long long)rec.w1, rec.del, rec.child, rec.pad[0], rec.pad[1], sf, rec.mask[0], rec.mask[1]);
			};
			int shown = 0, last_sid = -1;
			for (int64_t i = 0; i < len && shown < 12; ++i) {
				const int64_t v = hp[(size_t)i];
				if (v >= 0 && (v & RB3_TENT)) {
					const int sid = (int)(v >> RB3_TENT_PBITS) & (RB3_TENT_IDS - 1);
					if (sid == last_sid) continue;
					last_sid = sid, ++shown;
					fprintf(stderr, "   row %lld (text position %lld): tentative value %lld, stretch %d; rows around: %lld %lld | %lld %lld\n", (long long)i, d_tw ? (long long)row2tp[(size_t)i] : -1LL, (long long)(v & RB3_TENT_MASK), sid,
							i > 1 ? (long long)hp[(size_t)i - 2] : -9LL, i > 0 ? (long long)hp[(size_t)i - 1] : -9LL, i + 1 < len ? (long long)hp[(size_t)i + 1] : -9LL, i + 2 < len ? (long long)hp[(size_t)i + 2] : -9LL);
					show(sid, "this");
					rb3_stretch_t rec; (void)hipMemcpy(&rec, tab + sid, sizeof(rec), hipMemcpyDeviceToHost);
					int first = (int)(rec.pad[0] & 0x7FFFFFFFu) - 1;
					if ((rec.w0 >> 62) == RB3_DEP_EVENT) show(first, "walker's first");
					int cur = sid;
					for (int hop = 0; hop < 6; ++hop) { // what it hangs on
						rb3_stretch_t rc2; (void)hipMemcpy(&rc2, tab + cur, sizeof(rc2), hipMemcpyDeviceToHost);
						if ((rc2.w0 >> 62) == 0) break;
						cur = RB3_DEP_PREV(rc2.w0);
						show(cur, (rc2.w0 >> 62) == RB3_DEP_EVENT ? "  before the event" : "  linked to");
					}
					// the text neighbourhood: records of the rows of the text positions to the right (where the settling walker comes from)
					if (d_tw && row2tp[(size_t)i] >= 0) {
						const int64_t t0 = row2tp[(size_t)i];
						fprintf(stderr, "      records at text positions t0-4 .. t0+40 (t0 = %lld):", (long long)t0);
						for (int64_t t = t0 - 4; t <= t0 + 40; ++t) if (t >= 0 && t < len) { const int64_t vv = hp[(size_t)(htw[t] >> 3)]; fprintf(stderr, " %s%d", vv < 0 ? "U" : (vv & RB3_TENT) ? "T" : "F", vv >= 0 && (vv & RB3_TENT) ? (int)((vv >> RB3_TENT_PBITS) & (RB3_TENT_IDS - 1)) : 0); }
						fprintf(stderr, "\n");
						if (shown == 1) { // the wider neighbourhood, run-length coded (kind, stretch, count), and the walkers that start there
							fprintf(stderr, "      records at text positions t0-60 .. t0+700:");
							int64_t run = 0; long long lastk = -2;
							for (int64_t t = t0 - 60; t <= t0 + 700; ++t) if (t >= 0 && t < len) {
								const int64_t vv = hp[(size_t)(htw[t] >> 3)];
								const long long k = vv < 0 ? -1 : (vv & RB3_TENT) ? (long long)((vv >> RB3_TENT_PBITS) & (RB3_TENT_IDS - 1)) : -3;
								if (k != lastk) { if (run) fprintf(stderr, " %s%lldx%lld", lastk == -1 ? "U" : lastk == -3 ? "F" : "T", lastk < 0 ? 0LL : lastk, (long long)run); lastk = k, run = 0; }
								++run;
							}
							fprintf(stderr, " %s%lldx%lld\n", lastk == -1 ? "U" : lastk == -3 ? "F" : "T", lastk < 0 ? 0LL : lastk, (long long)run);
							for (int64_t wi = 0; walkers && wi < n_walkers; ++wi)
								if (walkers[wi].row >= t0 - 400 && walkers[wi].row <= t0 + 1000)
									fprintf(stderr, "      walker %lld: starts at text position %lld (t0%+lld), nsteps %lld, flags %llx, ka0 %lld\n", (long long)wi, (long long)walkers[wi].row, (long long)(walkers[wi].row - t0), (long long)walkers[wi].nsteps, (unsigned long long)walkers[wi].flags, (long long)walkers[wi].ka0);
						}
					}
				}
			}
			if (walkers) fprintf(stderr, "   walkers %lld; first: text position %lld nsteps %lld flags %llx; second: %lld %lld\n", (long long)n_walkers, (long long)walkers[0].row, (long long)walkers[0].nsteps, (unsigned long long)walkers[0].flags, n_walkers > 1 ? (long long)walkers[1].row : -1LL, n_walkers > 1 ? (long long)walkers[1].nsteps : -1LL);
		}
#endif
		if (d_tw) return merge_staged_text(h, len, d_b2, commit, host_pos, host_acc2, rank_only, n_walkers, walkers, d_tw, 0);
		return merge_staged(h, len, d_b2, commit, host_pos, host_acc2, rank_only, n_walkers, walkers, 0);
	}
	if (tent && hm[MISC_LF_CHK + 1] != 0) { // sampled rows fail the LF relation: nothing was installed; once more without speculative records
		h->stt.n_fallbacks += 1;
		if (h->opt.verbose >= 2) fprintf(stderr, "[W::rb3gpu] %llu junctions or sampled rows fail the LF relation; redoing the merge without tentative records\n", hm[MISC_LF_CHK + 1]);
		if (d_tw) return merge_staged_text(h, len, d_b2, commit, host_pos, host_acc2, rank_only, n_walkers, walkers, d_tw, 0);
		return merge_staged(h, len, d_b2, commit, host_pos, host_acc2, rank_only, n_walkers, walkers, 0);
	}
	if (hm[2] != 0 || hm[3] != 0 || hm[4] != 0 || hm[MISC_LF_CHK + 1] != 0) {
		if (h->opt.verbose >= 1) fprintf(stderr, "[E::rb3gpu] rank phase left %llu rows unset, %llu out of order, %llu sampled rows that fail the LF relation\n", hm[2], hm[3], hm[MISC_LF_CHK + 1]);
		return RB3GPU_EINTERNAL;
	}
	h->stt.n_symbols_merged += len;
	if (!rank_only) {
		const bool ran_runspace = runspace_applies(h, len, ntot) && use_winpar(h, nwin);
		acc[0] = 0;
		for (int a = 0; a < 6; ++a) acc[a + 1] = acc[a] + (int64_t)hm[MISC_IX_TOT + a];
		nslots = (int64_t)hm[MISC_IX_TOT + 6];
		if ((ran_runspace && hm[MISC_RG_OVER] != 0) || (uint64_t)nslots > h->reb_slot_cap) {
			// the scratch of the window kernels or the slot array, both sized by estimates, did not take this rebuild: nothing was
			// emitted.  Once more with full sizes (pos[] and the rows per window are still on the device).
			if (h->opt.verbose >= 3) fprintf(stderr, "[M::%s] rebuild emitted again: %llu groups handed on, %lld slots vs a capacity of %llu\n", __func__,
					hm[MISC_RG_LISTS] >> 32, (long long)nslots, (unsigned long long)h->reb_slot_cap);
			HIPCHK(hipEventRecord(h->ev[4], h->st));
			if ((r = build_index<false>(h, len, d_b2, (const int64_t*)dpos, ntot, false, &ngrp, &nslots, acc, true, true)) < 0) return r;
			HIPCHK(hipEventRecord(h->ev[5], h->st));
			HIPCHK(hipMemcpyAsync(hm + MISC_RG_LISTS, misc + MISC_RG_LISTS, 8, hipMemcpyDeviceToHost, h->st));
			HIPCHK(hipStreamSynchronize(h->st));
			h->stt.ms_build += ev_ms(h->ev[4], h->ev[5]);
			h->stt.n_reb_again += 1;
		}
		if (ran_runspace) { // the run-space rebuild ran: how many groups it handed to the window kernels
			h->stt.n_reb_groups += ngrp, h->stt.n_reb_groups_window += (int64_t)(hm[MISC_RG_LISTS] >> 32);
			h->reb_last[0] = (int64_t)(hm[MISC_RG_LISTS] & 0xFFFFFFFFull), h->reb_last[1] = (int64_t)(hm[MISC_RG_LISTS] >> 32); // (the next rebuild sizes its grids by them)
		} else h->reb_last[0] = h->reb_last[1] = -1;
		// Which rebuild the next merge starts with: the run-space tiers hand on every group whose old range holds a bit-plane slot
		// (rounds ~7-30 of a pangenome build: most of them, and each costs the attempt on top of the symbol path), so where that was
		// most groups the next merges go straight to the plane kernel, which reports how many groups would have qualified.
		if (h->tn.plane_rebuild && !h->tn.window_rebuild && use_winpar(h, nwin)) {
			if (ran_runspace) { if ((int64_t)(hm[MISC_RG_LISTS] >> 32) * 10 > ngrp * 7) h->reb_pp_all = true; }
			else if (h->reb_pp_all && (int64_t)hm[MISC_PP_OK] * 10 >= ngrp * 3) h->reb_pp_all = false;
		}
		for (int a = 0; a <= 6; ++a)
			if (acc[a] != h->acc[a] + acc2[a]) {
				if (h->opt.verbose >= 1) fprintf(stderr, "[E::rb3gpu] the rebuilt index counts %lld symbols below %d, expected %lld\n", (long long)acc[a], a, (long long)(h->acc[a] + acc2[a]));
				return RB3GPU_EINTERNAL;
			}
		if (nslots > nwin) {
			if (h->opt.verbose >= 1) fprintf(stderr, "[E::rb3gpu] the rebuilt index has %lld slots for %lld windows\n", (long long)nslots, (long long)nwin);
			return RB3GPU_EINTERNAL;
		}
		// algorithmic bytes of this rebuild: 9 B per batch row + the old block array read + the new one written
		h->stt.bytes_rebuild += 9 * len + h->stt.bytes_index + ngrp * (int64_t)sizeof(rb3_grp_t) + nslots * (int64_t)sizeof(rb3_slot_t);
		if (commit) index_install(h, ngrp, nslots, ntot, acc);
	}
	if (h->opt.verbose >= 3)
		fprintf(stderr, "[M::%s::%.3f] merged %lld symbols (%lld strings): lf %.3f ms, rank %.3f ms (%llu LF steps), rebuild %.3f ms\n", __func__,
				now_s() - h->t0, (long long)len, (long long)acc2[1], ev_ms(h->ev[0], ev_rank0), ev_ms(ev_rank0, h->ev[2]), hm[1], ev_ms(h->ev[2], h->ev[3]));
	return 0;
}

/* host -> HBM copy of a partial BWT through two pinned staging buffers: the CPU copies chunk i+1
 * into pinned memory while the DMA engine moves chunk i (a pageable hipMemcpy is staged by the
 * runtime anyway, single-buffered and several times slower) */
static int upload_b2(rb3gpu_t *h, int64_t len, const uint8_t *bwt)
{
	int r;
	if ((r = buf_ensure(h, h->b2, (size_t)len + 16)) < 0) return r;
	if (h->stage[0] == nullptr) {
		for (int i = 0; i < 2; ++i)
			if (hipHostMalloc((void**)&h->stage[i], RB3_STAGE_BYTES, hipHostMallocDefault) != hipSuccess) { h->stage[i] = nullptr; break; }
	}
	const double t0 = now_s();
	hipEvent_t done[2] = { h->ev[4], h->ev[5] };
	HIPCHK(h2d_copy(h->b2.p, bwt, (size_t)len, h->st, h->stage, done));
	h->stt.ms_h2d += (now_s() - t0) * 1e3;
	return 0;
}

int rb3gpu_from_plain_dev(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt)
{
	if (!h || len <= 0 || !d_bwt) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	int r;
	int64_t ngrp = 0, nslots = 0, acc[7];
	// validate symbols with the tile histogram (fm-index.c:122-125)
	const int64_t ntile = (len + RB3_TILE - 1) / RB3_TILE;
	if ((r = buf_ensure(h, h->tcnt, (size_t)ntile * 32)) < 0) return r;
	if ((r = buf_ensure(h, h->tpre, (size_t)ntile * 64)) < 0) return r;
	if ((r = buf_ensure(h, h->misc, MISC_WORDS * 8)) < 0) return r;
	HIPCHK(hipEventRecord(h->ev[0], h->st));
	hipLaunchKernelGGL(k_tile_hist, dim3((unsigned)ntile), dim3(256), 0, h->st, d_bwt, len, (uint32_t*)h->tcnt.p);
	uint64_t total[8];
	if ((r = scan_records(h, (const uint32_t*)h->tcnt.p, ntile, (uint64_t*)h->tpre.p, (uint64_t*)h->misc.p + MISC_LF_TOT, total)) < 0) return r;
	if (total[6] != 0) return RB3GPU_ESYMBOL;
	index_drop(h);
	if ((r = build_index<true>(h, len, d_bwt, nullptr, len, false, &ngrp, &nslots, acc)) < 0) return r;
	HIPCHK(hipEventRecord(h->ev[1], h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	h->stt.ms_build += ev_ms(h->ev[0], h->ev[1]);
	index_install(h, ngrp, nslots, len, acc);
	if (h->opt.verbose >= 3)
		fprintf(stderr, "[M::%s::%.3f] encoded %lld symbols into %lld slots (%.3f ms)\n", __func__, now_s() - h->t0, (long long)len, (long long)nslots, ev_ms(h->ev[0], h->ev[1]));
	return 0;
}

int rb3gpu_from_plain(rb3gpu_t *h, int64_t len, const uint8_t *bwt)
{
	if (!h || len <= 0 || !bwt) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	int r;
	if ((r = upload_b2(h, len, bwt)) < 0) return r;
	return rb3gpu_from_plain_dev(h, len, (const uint8_t*)h->b2.p);
}

/* The reference's signature (the partial BWT only, fm-index.c:279) for a batch of long strings: row words (lf_build), the sparse LF walk and its jumping
 * rounds (as for the device-made walker list of rounds 3-5), then every splitter writes the text-order words of its stretch (k_b2_tw) and the batch goes
 * through the text path -- streamed words, the common step of k_chain, the walker list of rb3gpu_merge_text_step_dev -- instead of through walkers that
 * chase row words (139 against 80 ms of k_chain per 152-genome build).  One host synchronisation more (the number of strings).  *done = false: not a batch
 * for this path (short strings, a split forced by the caller, an empty index): the caller goes on as before. */
static int merge_plain_via_tw(rb3gpu_t *h, int64_t len, const uint8_t *d_b2, int commit, bool *done)
{
	*done = false;
	if (!h->tn.b2_tw || h->tn.b2_split == 0 || h->opt.split_log2 != 0 || len < 4096 || h->n <= 0 || h->grp == nullptr || h->tn.staged || len >= (1LL << 40)) return 0;
	const int b2S = h->tn.b2_split > 0 ? h->tn.b2_split : len < (32LL << 20) ? 4 : len < (128LL << 20) ? 5 : 6;
	const int64_t m2cap = len / 64 + 1, nspmax = (len >> b2S) + m2cap + 2;
	int r;
	if ((r = buf_ensure(h, h->post, (size_t)len * 8)) < 0) return r;
	if ((r = buf_ensure(h, h->twb, (size_t)len * 8 + 64)) < 0) return r;
	if ((r = buf_ensure(h, h->xbuf, ((size_t)nspmax * 4 + 2 * (size_t)(m2cap + 2)) * 8)) < 0) return r;
	if ((r = buf_ensure(h, h->misc, MISC_WORDS * 8)) < 0) return r;
	int64_t acc2[7];
	HIPCHK(hipEventRecord(h->ev[0], h->st));
	if ((r = lf_build(h, len, d_b2, (int64_t*)h->post.p, acc2)) < 0) return r; // row words + the batch's symbol counts on the host (one synchronisation)
	const int64_t m2 = acc2[1], step = m2 > 0 ? rb3gpu_walker_step(h->dev, len, m2) : 0;
	if (m2 <= 0 || m2 > m2cap || len / m2 <= 4 * (int64_t)RB3_B2_W || step < 2) return 0; // short strings: one walker per string over the row words, as before
	unsigned long long *misc = (unsigned long long*)h->misc.p;
	const uint64_t *tot2 = (const uint64_t*)(misc + MISC_LF_TOT);
	unsigned long long *mode = misc + MISC_B2_MODE, *bad = misc + 12;
	uint64_t *lnk[2] = { (uint64_t*)h->xbuf.p, (uint64_t*)h->xbuf.p + 2 * nspmax };
	uint64_t *slen2 = lnk[1] + 2 * nspmax, *sidx = slen2 + m2cap + 2;
	HIPCHK(hipMemsetAsync(slen2, 0, (size_t)(m2cap + 2) * 8, h->st));
	HIPCHK(hipMemsetAsync(sidx, 0xff, (size_t)(m2cap + 2) * 8, h->st));
	HIPCHK(hipMemsetAsync(bad, 0, 8, h->st));
	hipLaunchKernelGGL(k_b2_mode, dim3(1), dim3(64), 0, h->st, tot2, len, m2cap, mode, (int64_t)RB3_B2_W);
	hipLaunchKernelGGL(k_b2_walk, dim3((unsigned)((nspmax + 255) / 256)), dim3(256), 0, h->st, (const int64_t*)h->post.p, len, tot2, b2S, (const unsigned long long*)mode, lnk[0]);
	int cur = 0;
	for (int64_t reach = 1; reach < nspmax; reach <<= 2, cur ^= 1)
		hipLaunchKernelGGL(k_b2_jump4, dim3((unsigned)((nspmax + 255) / 256)), dim3(256), 0, h->st, nspmax, (const uint64_t*)lnk[cur], lnk[cur ^ 1]);
	hipLaunchKernelGGL(k_b2_strings2, dim3(64), dim3(256), 0, h->st, tot2, (const unsigned long long*)mode, (const uint64_t*)lnk[cur], slen2, sidx);
	hipLaunchKernelGGL(k_b2_scan, dim3(1), dim3(1024), 0, h->st, tot2, (const unsigned long long*)mode, slen2);
	hipLaunchKernelGGL(k_b2_tw, dim3((unsigned)((nspmax + 255) / 256)), dim3(256), 0, h->st, (const int64_t*)h->post.p, len, tot2, b2S, (const unsigned long long*)mode, (const uint64_t*)lnk[cur],
			(const uint64_t*)slen2, (const uint64_t*)sidx, (uint64_t*)h->twb.p, bad);
	unsigned long long hb[2] = {0, 0};
	HIPCHK(hipMemcpyAsync(&hb[0], bad, 8, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipMemcpyAsync(&hb[1], slen2 + m2, 8, hipMemcpyDeviceToHost, h->st)); // (the total behind the last string: every row has its place)
	HIPCHK(hipEventRecord(h->ev[1], h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	h->stt.ms_lf += ev_ms(h->ev[0], h->ev[1]);
	if (hb[0] != 0 || (int64_t)hb[1] != len) { // (a BWT that is not one of complete strings: the walk over row words finds out what is wrong with it)
		if (h->opt.verbose >= 2) fprintf(stderr, "[W::rb3gpu] text-order words from the batch's own LF walk: %llu splitters without a string, %llu of %lld rows placed; walking row words instead\n", hb[0], hb[1], (long long)len);
		return 0;
	}
	h->mg_sa = nullptr, h->mg_step = step;
	r = merge_core(h, len, d_b2, commit, nullptr, nullptr, 0, m2, nullptr, (const uint64_t*)h->twb.p);
	h->mg_step = 0;
	if (r == RB3GPU_EINVAL || r == RB3GPU_EINTERNAL) { // (a failed merge installs nothing: the walk over row words decides what is wrong with the batch, or merges it)
		if (h->opt.verbose >= 2) fprintf(stderr, "[W::rb3gpu] the merge through device-made text-order words failed (%s); walking row words instead\n", rb3gpu_strerror(r));
		return 0;
	}
	*done = true;
	return r;
}

int rb3gpu_merge_plain_dev(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt, int commit)
{
	if (!h || len <= 0 || !d_bwt) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	bool done = false;
	const int r = merge_plain_via_tw(h, len, d_bwt, commit, &done);
	if (r < 0 || done) return r;
	return merge_core(h, len, d_bwt, commit, nullptr, nullptr, 0, 0, nullptr);
}

int rb3gpu_merge_plain(rb3gpu_t *h, int64_t len, const uint8_t *bwt)
{
	if (!h || len <= 0 || !bwt) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->n <= 0) return RB3GPU_ESTATE;
	int r;
	if ((r = upload_b2(h, len, bwt)) < 0) return r;
	bool done = false;
	r = merge_plain_via_tw(h, len, (const uint8_t*)h->b2.p, 1, &done);
	if (r < 0 || done) return r;
	return merge_core(h, len, (const uint8_t*)h->b2.p, 1, nullptr, nullptr, 0, 0, nullptr);
}

int rb3gpu_merge_plain_walkers(rb3gpu_t *h, int64_t len, const uint8_t *bwt, int64_t n_walkers, const rb3gpu_walker_t *walkers)
{
	if (!h || len <= 0 || !bwt || n_walkers <= 0 || !walkers) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->n <= 0) return RB3GPU_ESTATE;
	int r;
	if ((r = upload_b2(h, len, bwt)) < 0) return r;
	return merge_core(h, len, (const uint8_t*)h->b2.p, 1, nullptr, nullptr, 0, n_walkers, walkers);
}

int rb3gpu_merge_plain_dev_walkers(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt, int64_t n_walkers, const rb3gpu_walker_t *walkers, int commit)
{
	if (!h || len <= 0 || !d_bwt || n_walkers <= 0) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	return merge_core(h, len, d_bwt, commit, nullptr, nullptr, 0, n_walkers, walkers);
}

int rb3gpu_merge_text_dev(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt, const uint64_t *d_tw, int64_t n_walkers, const rb3gpu_walker_t *walkers, int commit)
{
	if (!h || len <= 0 || !d_bwt || !d_tw || n_walkers <= 0) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	return merge_core(h, len, d_bwt, commit, nullptr, nullptr, 0, n_walkers, walkers, d_tw);
}

/* ... with the walker list made on the device: n_strings strings, a walker at every sentinel and at every multiple of `step` inside a string (the list of
 * rb3h_walkers_text, host/sais.c:74-113; step from rb3gpu_walker_step).  d_sa: NULL, or the suffix array as for rb3gpu_merge_text_sa_dev. */
int rb3gpu_merge_text_step_dev(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt, const uint64_t *d_tw, const uint32_t *d_sa, int64_t n_strings, int64_t step, int commit)
{
	if (!h || len <= 0 || !d_bwt || !d_tw || n_strings <= 0 || n_strings > len || step < 2) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	h->mg_sa = d_sa, h->mg_step = step;
	const int r = merge_core(h, len, d_bwt, commit, nullptr, nullptr, 0, n_strings, nullptr, d_tw);
	h->mg_sa = nullptr, h->mg_step = 0;
	return r;
}

/* the list rb3gpu_merge_text_step_dev walks, copied to the host without its empty slots (tests; *walkers is malloc'ed: rb3gpu_host_free) */
int rb3gpu_walkers_step_dev(rb3gpu_t *h, int64_t len, const uint64_t *d_tw, int64_t n_strings, int64_t step, int64_t *n_walkers, rb3gpu_walker_t **walkers)
{
	if (!h || len <= 0 || !d_tw || n_strings <= 0 || n_strings > len || step < 2 || !n_walkers || !walkers) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	return step_list_host(h, len, d_tw, n_strings, step, n_walkers, walkers);
}

int rb3gpu_merge_text_sa_dev(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt, const uint64_t *d_tw, const uint32_t *d_sa, int64_t n_walkers, const rb3gpu_walker_t *walkers, int commit)
{
	if (!h || len <= 0 || !d_bwt || !d_tw || n_walkers <= 0) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	h->mg_sa = d_sa; // (NULL: rb3gpu_merge_text_dev)
	const int r = merge_core(h, len, d_bwt, commit, nullptr, nullptr, 0, n_walkers, walkers, d_tw);
	h->mg_sa = nullptr;
	return r;
}

int rb3gpu_mg_rank_text_dev(rb3gpu_t *h, int64_t len, const uint8_t *d_bwt, const uint64_t *d_tw, int64_t n_walkers, const rb3gpu_walker_t *walkers, int64_t *pos, int64_t acc2[RB3GPU_ASIZE+1])
{
	if (!h || len <= 0 || !d_bwt || !d_tw || n_walkers <= 0 || !pos) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->n <= 0) return RB3GPU_ESTATE;
	return merge_core(h, len, d_bwt, 0, pos, acc2, 1, n_walkers, walkers, d_tw);
}

int rb3gpu_mg_rank_plain(rb3gpu_t *h, int64_t len, const uint8_t *bwt, int64_t *pos, int64_t acc2[RB3GPU_ASIZE+1])
{
	if (!h || len <= 0 || !bwt || !pos) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->n <= 0) return RB3GPU_ESTATE;
	int r;
	if ((r = upload_b2(h, len, bwt)) < 0) return r;
	return merge_core(h, len, (const uint8_t*)h->b2.p, 0, pos, acc2, 1, 0, nullptr);
}

int rb3gpu_mg_rank_plain_walkers(rb3gpu_t *h, int64_t len, const uint8_t *bwt, int64_t n_walkers, const rb3gpu_walker_t *walkers, int64_t *pos, int64_t acc2[RB3GPU_ASIZE+1])
{
	if (!h || len <= 0 || !bwt || !pos || n_walkers <= 0 || !walkers) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->n <= 0) return RB3GPU_ESTATE;
	int r;
	if ((r = upload_b2(h, len, bwt)) < 0) return r;
	return merge_core(h, len, (const uint8_t*)h->b2.p, 0, pos, acc2, 1, n_walkers, walkers);
}

int rb3gpu_rank1a_batch(rb3gpu_t *h, int64_t n, const int64_t *k, int64_t *ok)
{
	if (!h || n < 0 || (n > 0 && (!k || !ok))) return RB3GPU_EINVAL;
	if (n == 0) return 0;
	HIPCHK(hipSetDevice(h->dev));
	if (h->grp == nullptr) return RB3GPU_ESTATE;
	int r;
	if ((r = buf_ensure(h, h->xbuf, (size_t)n * 56)) < 0) return r;
	int64_t *dk = (int64_t*)h->xbuf.p, *dok = dk + n;
	HIPCHK(hipMemcpyAsync(dk, k, (size_t)n * 8, hipMemcpyHostToDevice, h->st));
	Acc7 acc;
	memcpy(acc.a, h->acc, sizeof(acc.a));
	int64_t nblk = (n * 8 + 255) / 256;
	if (nblk > 4096) nblk = 4096;
	hipLaunchKernelGGL(k_rank_batch, dim3((unsigned)nblk), dim3(256), 0, h->st, view_of(h), acc, n, (const int64_t*)dk, dok);
	HIPCHK(hipMemcpyAsync(ok, dok, (size_t)n * 48, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	return 0;
}

int rb3gpu_get_acc(const rb3gpu_t *h, int64_t acc[RB3GPU_ASIZE+1])
{
	if (!h || !acc) return RB3GPU_EINVAL;
	memcpy(acc, h->acc, sizeof(h->acc));
	return 0;
}

int64_t rb3gpu_get_tot(const rb3gpu_t *h)
{
	return h ? h->n : RB3GPU_EINVAL;
}

#define RB3_XCHUNK (64LL << 20)

static int export_chunk(rb3gpu_t *h, int64_t beg, int64_t end, uint8_t *host)
{
	int r;
	if ((r = buf_ensure(h, h->xbuf, (size_t)RB3_XCHUNK)) < 0) return r;
	int64_t nblk = (end - beg + 255) / 256;
	if (nblk > 65536) nblk = 65536;
	hipLaunchKernelGGL(k_export_plain, dim3((unsigned)nblk), dim3(256), 0, h->st, view_of(h), beg, end, (uint8_t*)h->xbuf.p);
	HIPCHK(hipMemcpyAsync(host, h->xbuf.p, (size_t)(end - beg), hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	return 0;
}

int rb3gpu_export_plain(rb3gpu_t *h, uint8_t *out)
{
	if (!h || !out) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->grp == nullptr) return RB3GPU_ESTATE;
	double t = now_s();
	for (int64_t beg = 0; beg < h->n; beg += RB3_XCHUNK) {
		int64_t end = beg + RB3_XCHUNK < h->n ? beg + RB3_XCHUNK : h->n;
		int r = export_chunk(h, beg, end, out + beg);
		if (r < 0) return r;
	}
	h->stt.ms_export += (now_s() - t) * 1e3;
	return 0;
}

int rb3gpu_export_plain_dev(rb3gpu_t *h, uint8_t *d_out)
{
	if (!h || !d_out) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->grp == nullptr) return RB3GPU_ESTATE;
	int64_t nblk = (h->n + 255) / 256;
	if (nblk > 65536) nblk = 65536;
	hipLaunchKernelGGL(k_export_plain, dim3((unsigned)nblk), dim3(256), 0, h->st, view_of(h), (int64_t)0, h->n, d_out);
	HIPCHK(hipStreamSynchronize(h->st));
	return 0;
}

/* the symbols [beg, end) of the index, one byte each, into device memory (what rb3gpu_shard_split cuts the index with: an interval at a time) */
int rb3gpu_export_plain_range_dev(rb3gpu_t *h, int64_t beg, int64_t end, uint8_t *d_out)
{
	if (!h || !d_out || beg < 0 || end < beg) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->grp == nullptr || end > h->n) return RB3GPU_ESTATE;
	if (end == beg) return 0;
	int64_t nblk = (end - beg + 255) / 256;
	if (nblk > 65536) nblk = 65536;
	hipLaunchKernelGGL(k_export_plain, dim3((unsigned)nblk), dim3(256), 0, h->st, view_of(h), beg, end, d_out);
	HIPCHK(hipStreamSynchronize(h->st));
	return 0;
}

/* n intervals of positions that hold about the same number of BYTES of the block array each ("balanced by bytes of runs";
 * a run-coded stretch of the index costs a fraction of what a stretch of bit planes costs per symbol): bounds[0] = 0 <= ... <= bounds[n] =
 * the index's symbols, cut at group boundaries (8192 symbols) from the slot numbers in the group directory.  Fewer groups than intervals:
 * equal numbers of symbols. */
int rb3gpu_balanced_bounds(rb3gpu_t *h, int n, int64_t *bounds)
{
	if (!h || !bounds || n < 1) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->grp == nullptr) return RB3GPU_ESTATE;
	const int64_t tot = h->n, ngrp = h->ngrp;
	for (int i = 0; i <= n; ++i) bounds[i] = tot / n * i + (tot % n) * i / n;
	bounds[n] = tot;
	if (ngrp < 4 * (int64_t)n || h->nslots <= 0) return 0;
	std::vector<uint64_t> sm((size_t)ngrp);
	HIPCHK(hipMemcpy(sm.data(), view_of(h).gsm, (size_t)ngrp * 8, hipMemcpyDeviceToHost));
	// bytes before group g: 72 per group + 128 per slot before it
	int64_t g = 0;
	const double total = (double)ngrp * (double)RB3_GRP_ALLOC + (double)h->nslots * (double)sizeof(rb3_slot_t);
	for (int i = 1; i < n; ++i) {
		const double want = total * i / n;
		while (g < ngrp && (double)g * (double)RB3_GRP_ALLOC + (double)(uint32_t)sm[(size_t)g] * (double)sizeof(rb3_slot_t) < want) ++g;
		int64_t b = g << RB3_GRP_BITS;
		if (b <= bounds[i - 1]) b = bounds[i - 1] + 1; // (every interval holds a symbol)
		if (b > tot - (n - i)) b = tot - (n - i);
		bounds[i] = b;
	}
	return 0;
}

static int sort_text_impl(rb3gpu_t *h, int64_t len, const uint8_t *text, uint8_t *d_bwt, int64_t step, int64_t *ckrow, uint64_t *d_tw, uint32_t *d_sa = nullptr)
{
	if (!h || !text || !d_bwt || len <= 0 || len >= (1LL << 31) || step < 0) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	const double t0 = now_s();
	int r, rounds = 0;
	if (h->sorter == nullptr && (h->sorter = rb3sort_create()) == nullptr) return RB3GPU_ENOMEM;
	if ((r = upload_b2(h, len, text)) < 0) return r; // the text, into the batch buffer
	int64_t *d_ck = nullptr;
	const int64_t nck = step > 0 && ckrow ? (len + step - 1) / step : 0;
	if (nck > 0) {
		if ((r = buf_ensure(h, h->xbuf, (size_t)nck * 8)) < 0) return r;
		d_ck = (int64_t*)h->xbuf.p;
	}
	r = rb3sort_bwt(h->sorter, h->st, len, (const uint8_t*)h->b2.p, d_bwt, step, d_ck, &rounds, d_tw, d_sa);
	if (r == -1 && !h->garbage.empty()) { // the sorter's scratch did not fit while replaced buffers of the handle are still held: give them back, once more
		(void)hipGetLastError();
		garbage_collect(h, true);
		r = rb3sort_bwt(h->sorter, h->st, len, (const uint8_t*)h->b2.p, d_bwt, step, d_ck, &rounds, d_tw, d_sa);
	}
	if (r < 0) return r == -1 ? RB3GPU_ENOMEM : r == -3 ? RB3GPU_ESYMBOL : RB3GPU_ENODEV;
	if (nck > 0) HIPCHK(hipMemcpy(ckrow, d_ck, (size_t)nck * 8, hipMemcpyDeviceToHost));
	h->stt.ms_sort += (now_s() - t0) * 1e3, h->stt.n_sort_rounds += rounds;
	if (h->bytes_owned + rb3sort_bytes(h->sorter) > h->stt.bytes_peak) h->stt.bytes_peak = h->bytes_owned + rb3sort_bytes(h->sorter);
	if (h->opt.verbose >= 3)
		fprintf(stderr, "[M::%s::%.3f] suffix-sorted %lld symbols on the GPU in %.3f ms (%d doubling rounds)\n", __func__, now_s() - h->t0, (long long)len, (now_s() - t0) * 1e3, rounds);
	return 0;
}

int rb3gpu_bwt_from_text(rb3gpu_t *h, int64_t len, const uint8_t *text, uint8_t *d_bwt, int64_t step, int64_t *ckrow)
{
	return sort_text_impl(h, len, text, d_bwt, step, ckrow, nullptr);
}

int rb3gpu_sort_text(rb3gpu_t *h, int64_t len, const uint8_t *text, uint8_t *d_bwt, uint64_t *d_tw)
{
	if (!d_tw) return RB3GPU_EINVAL;
	return sort_text_impl(h, len, text, d_bwt, 0, nullptr, d_tw);
}

int rb3gpu_sort_text_sa(rb3gpu_t *h, int64_t len, const uint8_t *text, uint8_t *d_bwt, uint64_t *d_tw, uint32_t *d_sa)
{
	if (!d_tw || !d_sa) return RB3GPU_EINVAL;
	return sort_text_impl(h, len, text, d_bwt, 0, nullptr, d_tw, d_sa);
}

/* ---- a sorter of its own: stream, scratch, two output buffers handed out in turn ---- */

#define SCHK(x) do { if ((x) != hipSuccess) { (void)hipGetLastError(); return RB3GPU_ENODEV; } } while (0)

struct rb3gpu_sorter_s {
	int dev = 0;
	hipStream_t st = nullptr;
	rb3sort_ws *ws = nullptr;
	void *text = nullptr, *ck = nullptr, *out[2] = {nullptr, nullptr};
	size_t text_cap = 0, ck_cap = 0, out_cap[2] = {0, 0};
	int busy[2] = {0, 0};
	uint8_t *stage[2] = {nullptr, nullptr}; // pinned, for the upload of a text that is not in page-locked memory itself
	hipEvent_t done[2];
	int64_t text_len = 0;     // symbols of the text that rb3gpu_sorter_upload left in `text`
	double ms_upload = 0, ms_sort = 0; // cumulative: text host -> HBM; suffix sorting + BWT + text-order words
	int64_t n_sorted = 0, n_symbols = 0;
	pthread_mutex_t mtx;
	pthread_cond_t cv;
};

static int sorter_grow(void **p, size_t *cap, size_t bytes)
{
	if (*p && *cap >= bytes) return 0;
	if (*p) (void)hipFree(*p);
	*p = nullptr, *cap = 0;
	const size_t want = bytes + (bytes >> 3) + 256;
	if (hipMalloc(p, want) != hipSuccess) { (void)hipGetLastError(); return RB3GPU_ENOMEM; }
	*cap = want;
	return 0;
}

rb3gpu_sorter_t *rb3gpu_sorter_create(int device)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n || hipSetDevice(device) != hipSuccess) return nullptr;
	rb3gpu_sorter_t *s = new rb3gpu_sorter_s;
	s->dev = device;
	if (hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking) != hipSuccess || (s->ws = rb3sort_create()) == nullptr) { delete s; return nullptr; }
	for (int i = 0; i < 2; ++i) {
		if (hipHostMalloc((void**)&s->stage[i], (size_t)4 << 20, hipHostMallocDefault) != hipSuccess) s->stage[i] = nullptr;
		if (hipEventCreateWithFlags(&s->done[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); if (s->stage[i]) (void)hipHostFree(s->stage[i]); s->stage[i] = nullptr; }
	}
	pthread_mutex_init(&s->mtx, nullptr);
	pthread_cond_init(&s->cv, nullptr);
	return s;
}

void rb3gpu_sorter_destroy(rb3gpu_sorter_t *s)
{
	if (!s) return;
	(void)hipSetDevice(s->dev);
	(void)hipStreamSynchronize(s->st);
	rb3sort_destroy(s->ws);
	void *all[] = { s->text, s->ck, s->out[0], s->out[1] };
	for (void *p : all) if (p) (void)hipFree(p);
	for (int i = 0; i < 2; ++i) if (s->stage[i]) { (void)hipHostFree(s->stage[i]); (void)hipEventDestroy(s->done[i]); }
	(void)hipStreamDestroy(s->st);
	pthread_mutex_destroy(&s->mtx);
	pthread_cond_destroy(&s->cv);
	delete s;
}

static void sorter_give_back(rb3gpu_sorter_t *s, int slot)
{
	pthread_mutex_lock(&s->mtx);
	s->busy[slot] = 0;
	pthread_cond_broadcast(&s->cv);
	pthread_mutex_unlock(&s->mtx);
}

/* stage 1: the text of a batch into the sorter's text buffer in HBM (the H2D of the merge path) */
static int sorter_upload(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text)
{
	if (!s || !text || len <= 0 || len >= (1LL << 31)) return RB3GPU_EINVAL;
	SCHK(hipSetDevice(s->dev));
	int r;
	s->text_len = 0;
	if ((r = sorter_grow(&s->text, &s->text_cap, (size_t)len + 16)) < 0) return r;
	const double t_up = now_s();
	if (h2d_copy(s->text, text, (size_t)len, s->st, s->stage, s->done) != hipSuccess) { (void)hipGetLastError(); return RB3GPU_ENODEV; }
	s->ms_upload += (now_s() - t_up) * 1e3;
	s->text_len = len;
	return 0;
}

/* Stage 1 for a batch of a few long records on both strands (genomes, assemblies): only the FORWARD strands cross PCIe -- each with
 * its sentinel, straight to its place in the text -- and the reverse complements (io.c:30-40, 84-102) are written on the device:
 * half the bytes of the H2D copy.  pair_start[i] .. pair_start[i+1] is record i as rb3_seq_read lays it out: l symbols, 0, the l
 * symbols of the reverse complement, 0. */
#define RB3_FWD_MAXPAIRS 32
struct FwdPairs { int64_t start[RB3_FWD_MAXPAIRS + 1]; int n; };

__global__ void __launch_bounds__(256) k_revcomp_fill(uint8_t *text, FwdPairs pp)
{
	const int64_t tot = pp.start[pp.n] - pp.start[0];
	for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < tot / 2; t += (int64_t)gridDim.x * blockDim.x) {
		// t counts the positions of the second halves (reverse strand + its sentinel) of all pairs
		int i = 0;
		int64_t u = t;
		while (i + 1 < pp.n && u >= (pp.start[i + 1] - pp.start[i]) / 2) u -= (pp.start[i + 1] - pp.start[i]) / 2, ++i;
		const int64_t ps = pp.start[i], l = (pp.start[i + 1] - ps) / 2 - 1;
		uint8_t c = 0;
		if (u < l) { c = text[ps + l - 1 - u]; c = (c >= 1 && c <= 4) ? (uint8_t)(5 - c) : c; }
		text[ps + l + 1 + u] = c;
	}
}

static int sorter_upload_fwd(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text, int64_t n_pairs, const int64_t *pair_start, bool wait = true)
{
	if (!s || !text || !pair_start || len <= 0 || len >= (1LL << 31) || n_pairs < 1 || n_pairs > RB3_FWD_MAXPAIRS) return RB3GPU_EINVAL;
	FwdPairs pp;
	pp.n = (int)n_pairs;
	for (int64_t i = 0; i <= n_pairs; ++i) pp.start[i] = i < n_pairs ? pair_start[i] : len;
	if (pp.start[0] != 0) return RB3GPU_EINVAL;
	for (int64_t i = 0; i < n_pairs; ++i) {
		const int64_t sz = pp.start[i + 1] - pp.start[i];
		if (sz < 4 || (sz & 1) || text[pp.start[i] + sz / 2 - 1] != 0 || text[pp.start[i + 1] - 1] != 0) return RB3GPU_EINVAL; // not the layout of two strands per record
	}
	SCHK(hipSetDevice(s->dev));
	int r;
	s->text_len = 0;
	if ((r = sorter_grow(&s->text, &s->text_cap, (size_t)len + 16)) < 0) return r;
	const double t_up = now_s();
	const bool pinned = is_pinned(text, (size_t)len);
	for (int64_t i = 0; i < n_pairs; ++i) {
		const int64_t ps = pp.start[i], half = (pp.start[i + 1] - ps) / 2;
		if (pinned) { if (hipMemcpyAsync((uint8_t*)s->text + ps, text + ps, (size_t)half, hipMemcpyHostToDevice, s->st) != hipSuccess) { (void)hipGetLastError(); return RB3GPU_ENODEV; } }
		else if (h2d_copy((uint8_t*)s->text + ps, text + ps, (size_t)half, s->st, s->stage, s->done) != hipSuccess) { (void)hipGetLastError(); return RB3GPU_ENODEV; }
	}
	int64_t nblk = (len / 2 + 1023) / 1024;
	hipLaunchKernelGGL(k_revcomp_fill, dim3((unsigned)(nblk > 4096 ? 4096 : nblk)), dim3(256), 0, s->st, (uint8_t*)s->text, pp);
	// (begin/end form: the copies of a page-locked text are queued and the caller goes on -- e.g. merges the batch before; the sorter's
	// stream orders them before the sort whether or not anybody waits)
	if ((wait || !pinned) && hipStreamSynchronize(s->st) != hipSuccess) { (void)hipGetLastError(); return RB3GPU_ENODEV; }
	s->ms_upload += (now_s() - t_up) * 1e3;
	s->text_len = len;
	return 0;
}

/* stage 2: suffix-sort the uploaded text into an output buffer that is not with the merger */
static int sorter_sort_uploaded(rb3gpu_sorter_t *s, int64_t len, void **d_bwt, int64_t step, int64_t *ckrow, void **d_tw, void **d_sa = nullptr)
{
	if (!s || !d_bwt || len <= 0 || len != s->text_len || step < 0) return RB3GPU_EINVAL;
	SCHK(hipSetDevice(s->dev));
	int r, slot, rounds = 0;
	*d_bwt = nullptr;
	pthread_mutex_lock(&s->mtx);
	while (s->busy[0] && s->busy[1]) pthread_cond_wait(&s->cv, &s->mtx);
	slot = s->busy[0] ? 1 : 0;
	s->busy[slot] = 1;
	pthread_mutex_unlock(&s->mtx);
	const int64_t nck = step > 0 && ckrow ? (len + step - 1) / step : 0;
	const size_t tw_off = ((size_t)len + 16 + 255) & ~(size_t)255; // the text-order words sit behind the BWT in the same buffer
	if (d_tw) *d_tw = nullptr;
	if (d_sa) *d_sa = nullptr;
	if (d_sa && !d_tw) { sorter_give_back(s, slot); return RB3GPU_EINVAL; }
	if ((r = sorter_grow(&s->out[slot], &s->out_cap[slot], d_tw ? tw_off + (size_t)len * (d_sa ? 12 : 8) : (size_t)len + 16)) < 0 ||
		(nck > 0 && (r = sorter_grow(&s->ck, &s->ck_cap, (size_t)nck * 8)) < 0)) {
		sorter_give_back(s, slot); // (or the next two calls would wait for it for ever)
		return r;
	}
	const double t_so = now_s();
	r = rb3sort_bwt(s->ws, s->st, len, (const uint8_t*)s->text, (uint8_t*)s->out[slot], step, nck > 0 ? (int64_t*)s->ck : nullptr, &rounds,
			d_tw ? (uint64_t*)((uint8_t*)s->out[slot] + tw_off) : nullptr, d_sa ? (uint32_t*)((uint8_t*)s->out[slot] + tw_off + (size_t)len * 8) : nullptr);
	s->ms_sort += (now_s() - t_so) * 1e3, s->n_sorted += 1, s->n_symbols += len;
	if (r == 0 && nck > 0 && (hipMemcpyAsync(ckrow, s->ck, (size_t)nck * 8, hipMemcpyDeviceToHost, s->st) != hipSuccess || hipStreamSynchronize(s->st) != hipSuccess)) r = -2;
	if (r < 0) {
		sorter_give_back(s, slot);
		return r == -1 ? RB3GPU_ENOMEM : r == -3 ? RB3GPU_ESYMBOL : RB3GPU_ENODEV;
	}
	*d_bwt = s->out[slot];
	if (d_tw) *d_tw = (uint8_t*)s->out[slot] + tw_off;
	if (d_sa) *d_sa = (uint8_t*)s->out[slot] + tw_off + (size_t)len * 8; // (behind the text-order words, released with the BWT)
	return 0;
}

static int sorter_impl(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text, void **d_bwt, int64_t step, int64_t *ckrow, void **d_tw)
{
	if (!s || !text || !d_bwt || len <= 0 || len >= (1LL << 31) || step < 0) return RB3GPU_EINVAL;
	int r;
	if ((r = sorter_upload(s, len, text)) < 0) return r;
	return sorter_sort_uploaded(s, len, d_bwt, step, ckrow, d_tw);
}

int rb3gpu_sorter_upload(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text)
{
	return sorter_upload(s, len, text);
}

int rb3gpu_sorter_upload_fwd(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text, int64_t n_pairs, const int64_t *pair_start)
{
	return sorter_upload_fwd(s, len, text, n_pairs, pair_start);
}

int64_t rb3gpu_walker_step(int device, int64_t len, int64_t n_strings)
{
	static int cus[64]; // compute units per device x resident walkers per compute unit / 160 (0: not asked yet)
	if (device < 0 || device >= 64 || len <= 0) return RB3GPU_EINVAL;
	int cu = __atomic_load_n(&cus[device], __ATOMIC_RELAXED);
	if (cu == 0) {
		int dev0 = 0, nb = 0;
		if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cu <= 0) { (void)hipGetLastError(); return RB3GPU_ENODEV; }
		// blocks of 256 threads of the walker kernel (the variant the text-order walk of a pangenome build runs) a compute unit keeps
		// resident: 5 with its 88 registers; asked of the runtime so that a compiler that allocates differently does not silently
		// put the walkers beyond the resident ones
		if (hipGetDevice(&dev0) == hipSuccess && hipSetDevice(device) == hipSuccess) {
			if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_chain<true, false, true, 1, 8>, 256, 0) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
			(void)hipSetDevice(dev0);
		}
		if (nb >= 1 && nb <= 8) cu = cu * nb / 5; // (room below counts 160 walkers = 5 blocks x 4 waves x 8 per unit)
		__atomic_store_n(&cus[device], cu, __ATOMIC_RELAXED);
	}
	// k_chain: 88 registers -> 5 waves per SIMD, 4 SIMDs per compute unit, 8 walkers per wave.  One sixteenth is left free: the
	// kernels that run beside the walkers (the batch's histogram on the side stream, the strand kernel of the next batch's upload)
	// take wave slots too, and a walker whose wave starts late is not only slow -- whoever runs into its rows before it has
	// recorded them walks them as well, and a merge now and then (1 in ~1000 at 63/64) had to be redone without tentative records
	int64_t room = (int64_t)cu * 160 * 15 / 16 - (n_strings > 0 ? n_strings : 0);
	if (room < 1024) room = 1024;
	const int64_t step = (len + room - 1) / room;
	return step < 192 ? 192 : step;
}

int rb3gpu_sorter_upload_fwd_begin(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text, int64_t n_pairs, const int64_t *pair_start)
{
	return sorter_upload_fwd(s, len, text, n_pairs, pair_start, false);
}

int rb3gpu_sorter_upload_end(rb3gpu_sorter_t *s)
{
	if (!s) return RB3GPU_EINVAL;
	SCHK(hipSetDevice(s->dev));
	const double t = now_s();
	if (hipStreamSynchronize(s->st) != hipSuccess) { (void)hipGetLastError(); return RB3GPU_ENODEV; }
	s->ms_upload += (now_s() - t) * 1e3;
	return 0;
}

int rb3gpu_sorter_sort_uploaded(rb3gpu_sorter_t *s, int64_t len, void **d_bwt, void **d_tw)
{
	if (!d_tw) return RB3GPU_EINVAL;
	return sorter_sort_uploaded(s, len, d_bwt, 0, nullptr, d_tw);
}

int rb3gpu_sorter_sort_uploaded_sa(rb3gpu_sorter_t *s, int64_t len, void **d_bwt, void **d_tw, void **d_sa)
{
	if (!d_tw || !d_sa) return RB3GPU_EINVAL;
	return sorter_sort_uploaded(s, len, d_bwt, 0, nullptr, d_tw, d_sa);
}

int rb3gpu_sorter_bwt(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text, void **d_bwt, int64_t step, int64_t *ckrow)
{
	return sorter_impl(s, len, text, d_bwt, step, ckrow, nullptr);
}

int rb3gpu_sorter_sort_sa(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text, void **d_bwt, void **d_tw, void **d_sa)
{
	if (!s || !text || !d_bwt || !d_tw || !d_sa || len <= 0 || len >= (1LL << 31)) return RB3GPU_EINVAL;
	int r;
	if ((r = sorter_upload(s, len, text)) < 0) return r;
	return sorter_sort_uploaded(s, len, d_bwt, 0, nullptr, d_tw, d_sa);
}

int rb3gpu_sorter_sort(rb3gpu_sorter_t *s, int64_t len, const uint8_t *text, void **d_bwt, void **d_tw)
{
	if (!d_tw) return RB3GPU_EINVAL;
	return sorter_impl(s, len, text, d_bwt, 0, nullptr, d_tw);
}

int rb3gpu_sorter_stats(const rb3gpu_sorter_t *s, double *ms_upload, double *ms_sort, int64_t *n_batches, int64_t *n_symbols)
{
	if (!s) return RB3GPU_EINVAL;
	if (ms_upload) *ms_upload = s->ms_upload;
	if (ms_sort) *ms_sort = s->ms_sort;
	if (n_batches) *n_batches = s->n_sorted;
	if (n_symbols) *n_symbols = s->n_symbols;
	return 0;
}

int rb3gpu_sorter_release(rb3gpu_sorter_t *s, void *d_bwt)
{
	if (!s || !d_bwt) return RB3GPU_EINVAL;
	int found = 0;
	pthread_mutex_lock(&s->mtx);
	for (int i = 0; i < 2; ++i)
		if (s->out[i] == d_bwt && s->busy[i]) s->busy[i] = 0, found = 1;
	pthread_cond_broadcast(&s->cv);
	pthread_mutex_unlock(&s->mtx);
	return found ? 0 : RB3GPU_EINVAL;
}

int rb3gpu_ssa_dims(const rb3gpu_t *h, int ssa_shift, int64_t *m, int64_t *n_ssa, int *ms)
{
	if (!h || ssa_shift < 0 || ssa_shift > 40) return RB3GPU_EINVAL;
	if (h->grp == nullptr) return RB3GPU_ESTATE;
	int b;
	for (b = 1; (1LL << b) < h->acc[1]; ++b) {} // ssa.c:63
	if (m) *m = h->acc[1];
	if (n_ssa) *n_ssa = (h->n - h->acc[1] + (1LL << ssa_shift) - 1) >> ssa_shift; // ssa.c:64
	if (ms) *ms = b;
	return 0;
}

int rb3gpu_ssa_gen(rb3gpu_t *h, int ssa_shift, uint64_t *r2i, uint64_t *ssa)
{
	int64_t m, n_ssa;
	int ms, r;
	if (!h || !r2i || (r = rb3gpu_ssa_dims(h, ssa_shift, &m, &n_ssa, &ms)) < 0) return h && r2i ? r : RB3GPU_EINVAL;
	if (n_ssa > 0 && !ssa) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	// splitter spacing: the walk's longest sublist is about 2^S ln(number of splitters) steps, and every splitter
	// costs a queue pull; the linking is pointer jumping, whose cost hardly depends on S
	// (141 M rows, 32 strings: walk 14.5 / 9.6 / 11.7 / 18.6 ms for S = 7 / 8 / 9 / 10)
	int S = h->tn.ssa_split;
	if (S < 4) S = 4;
	if (S > 20) S = 20;
	const int64_t nsp = m + ((h->n - m + (1LL << S) - 1) >> S);
	if (nsp >= (1LL << (64 - RB3_SSA_LBITS)) || ms + 1 > 63) return RB3GPU_EINVAL;
	// scratch: two link tables (2 words per splitter), r2i, ssa, all u64
	const size_t words = (size_t)nsp * 4 + (size_t)m + (size_t)n_ssa + 8;
	if ((r = buf_ensure(h, h->xbuf, words * 8)) < 0) return r;
	if ((r = buf_ensure(h, h->misc, MISC_WORDS * 8)) < 0) return r;
	uint64_t *lnk[2] = { (uint64_t*)h->xbuf.p, (uint64_t*)h->xbuf.p + 2 * nsp };
	uint64_t *d_r2i = lnk[1] + 2 * nsp, *d_ssa = d_r2i + m;
	unsigned long long *misc = (unsigned long long*)h->misc.p;
	const double t0 = now_s();
	HIPCHK(hipMemsetAsync(misc, 0, 128, h->st));
	HIPCHK(hipMemsetAsync(d_r2i, 0, (size_t)(m + n_ssa) * 8, h->st)); // RB3_CALLOC in ssa.c:65-66
	HIPCHK(hipEventRecord(h->ev[0], h->st));
	{
		int64_t nblk = (nsp + 31) / 32;
		nblk = nblk > 256 * 8 ? 256 * 8 : nblk < 1 ? 1 : nblk;
		hipLaunchKernelGGL(k_ssa_walk, dim3((unsigned)nblk), dim3(256), 0, h->st, view_of(h), S, ssa_shift, nsp, lnk[0], d_ssa, misc, misc + 2);
	}
	HIPCHK(hipEventRecord(h->ev[1], h->st));
	int cur = 0;
	for (int64_t reach = 1; reach < nsp; reach <<= 1, cur ^= 1) // after this round every link spans 2 * reach splitters
		hipLaunchKernelGGL(k_ssa_jump, dim3((unsigned)((nsp + 255) / 256)), dim3(256), 0, h->st, nsp, (const uint64_t*)lnk[cur], lnk[cur ^ 1]);
	hipLaunchKernelGGL(k_ssa_heads, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, h->st, m, (const uint64_t*)lnk[cur], d_r2i, misc + 2);
	if (n_ssa > 0)
		hipLaunchKernelGGL(k_ssa_final, dim3((unsigned)((n_ssa + 255) / 256)), dim3(256), 0, h->st, n_ssa, ms, m, (const uint64_t*)lnk[cur], (const uint64_t*)d_r2i, d_ssa, misc + 2);
	HIPCHK(hipEventRecord(h->ev[2], h->st));
	unsigned long long hm[4];
	HIPCHK(hipMemcpyAsync(hm, misc, sizeof(hm), hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipMemcpyAsync(r2i, d_r2i, (size_t)m * 8, hipMemcpyDeviceToHost, h->st));
	if (n_ssa > 0) HIPCHK(hipMemcpyAsync(ssa, d_ssa, (size_t)n_ssa * 8, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	h->stt.ms_ssa += (now_s() - t0) * 1e3;
	h->stt.ms_ssa_walk += ev_ms(h->ev[0], h->ev[1]);
	if (hm[2] != 0 || hm[3] != 0) {
		if (h->opt.verbose >= 1) fprintf(stderr, "[E::rb3gpu] ssa: %llu sublists too long, %llu broken links\n", hm[2], hm[3]);
		return RB3GPU_EINTERNAL;
	}
	if (h->opt.verbose >= 3)
		fprintf(stderr, "[M::%s::%.3f] sampled suffix array: %lld strings, %lld samples, walk %.3f ms, link+final %.3f ms\n", __func__, now_s() - h->t0,
				(long long)m, (long long)n_ssa, ev_ms(h->ev[0], h->ev[1]), ev_ms(h->ev[1], h->ev[2]));
	return 0;
}

/* Runs are found on the device (k_export_runs_g: count, scan, emit), chunk by chunk; only start << 3 | sym of
 * every run crosses PCIe, and the host turns consecutive starts into lengths. */
#define RB3_RCHUNK_WINS (1LL << 16) /* windows (16 M symbols) per chunk: at most 128 MB of run words */
/* chunk by chunk: run starts on the device, copied through the pinned staging buffers when they fit */
static int export_run_words(rb3gpu_t *h, rb3gpu_emit_words_f emit, void *data)
{
	const int64_t nwin = (h->n + RB3_WIN - 1) >> RB3_WIN_BITS;
	const IdxView iv = view_of(h);
	int ret = 0;
	uint64_t *host = nullptr;
	size_t host_cap = 0;
	if ((ret = buf_ensure(h, h->misc, MISC_WORDS * 8)) < 0) return ret;
	if (h->stage[0] == nullptr)
		for (int i = 0; i < 2; ++i)
			if (hipHostMalloc((void**)&h->stage[i], RB3_STAGE_BYTES, hipHostMallocDefault) != hipSuccess) { h->stage[i] = nullptr; break; }
	const int64_t ngrp = (h->n + RB3_GRP - 1) >> RB3_GRP_BITS, gchunk = RB3_RCHUNK_WINS / RB3_GRP_WINS;
	(void)nwin;
	for (int64_t g0 = 0; g0 < ngrp && ret == 0; g0 += gchunk) {
		const int64_t ng = g0 + gchunk < ngrp ? gchunk : ngrp - g0;
		if ((ret = buf_ensure(h, h->gstat, (size_t)ng * 32)) < 0) break;
		if ((ret = buf_ensure(h, h->gpre, (size_t)ng * 64)) < 0) break;
		uint32_t *cnt8 = (uint32_t*)h->gstat.p;
		uint64_t *off8 = (uint64_t*)h->gpre.p, total[8];
		const dim3 grid((unsigned)((ng + 3) / 4)), blk(256);
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_export_runs_g<false>), grid, blk, 0, h->st, iv, g0, ng, cnt8, (const uint64_t*)nullptr, (uint64_t*)nullptr);
		if ((ret = scan_records(h, cnt8, ng, off8, (uint64_t*)h->misc.p + MISC_IX_TOT, total)) < 0) break;
		const int64_t nr = (int64_t)total[0];
		if (nr == 0) continue;
		if ((ret = buf_ensure(h, h->xbuf, (size_t)nr * 8)) < 0) break;
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_export_runs_g<true>), grid, blk, 0, h->st, iv, g0, ng, cnt8, (const uint64_t*)off8, (uint64_t*)h->xbuf.p);
		uint64_t *dst;
		if (h->stage[0] && (size_t)nr * 8 <= RB3_STAGE_BYTES) dst = (uint64_t*)h->stage[0];
		else {
			if ((size_t)nr > host_cap) {
				free(host);
				host_cap = (size_t)nr + ((size_t)nr >> 2) + 1024;
				if ((host = (uint64_t*)malloc(host_cap * 8)) == nullptr) { ret = RB3GPU_ENOMEM; break; }
			}
			dst = host;
		}
		HIPCHK(hipMemcpyAsync(dst, h->xbuf.p, (size_t)nr * 8, hipMemcpyDeviceToHost, h->st));
		HIPCHK(hipStreamSynchronize(h->st));
		if (emit(data, nr, dst, -1) != 0) ret = RB3GPU_EINVAL;
	}
	if (ret == 0 && emit(data, 0, nullptr, h->n) != 0) ret = RB3GPU_EINVAL;
	free(host);
	return ret;
}

int rb3gpu_export_run_words(rb3gpu_t *h, rb3gpu_emit_words_f emit, void *data)
{
	if (!h || !emit) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->grp == nullptr) return RB3GPU_ESTATE;
	const double t = now_s();
	const int ret = export_run_words(h, emit, data);
	h->stt.ms_export += (now_s() - t) * 1e3;
	return ret;
}

/* column 0 of the scan's 8-word records (the runs before every group), one word per group, and the total behind them */
__global__ void __launch_bounds__(256) k_runs_before(const uint64_t *off8, int64_t ngrp, uint64_t total, uint64_t *out)
{
	const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g <= ngrp) out[g] = g < ngrp ? off8[g * 8] : total;
}

/* The FMD word stream of the index, packed on the device A PIECE OF THE RUNS AT A TIME (rb3gpu_fmdenc.hip, rb3fmd_enc_*; round 6): the runs of as many groups as
 * make `piece` runs are written behind what the packer carried over from the piece before, packed, and their words go to the host.  Device memory: 34 bytes per
 * run of a PIECE (64 M runs by default: 2.2 GB; rb3gpu_tune "fmd_piece"), not of the index -- 4.99 G runs of four human haplotypes held 40 GB of run starts and
 * ~130 GB of tables in one piece. */
int rb3gpu_export_fmd_words(rb3gpu_t *h, uint64_t **words, int64_t *n_words)
{
	if (!h || !words || !n_words) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->grp == nullptr) return RB3GPU_ESTATE;
	const double t = now_s();
	const IdxView iv = view_of(h);
	int r;
	*words = nullptr, *n_words = 0;
	// the run starts of every group: count, scan
	if ((r = buf_ensure(h, h->misc, MISC_WORDS * 8)) < 0) return r;
	const int64_t ngrp = (h->n + RB3_GRP - 1) >> RB3_GRP_BITS;
	if ((r = buf_ensure(h, h->gstat, (size_t)ngrp * 32)) < 0) return r;
	if ((r = buf_ensure(h, h->gpre, (size_t)ngrp * 64)) < 0) return r;
	uint32_t *cnt8 = (uint32_t*)h->gstat.p;
	uint64_t *off8 = (uint64_t*)h->gpre.p, total[8];
	const dim3 grid((unsigned)((ngrp + 3) / 4)), blk(256);
	hipLaunchKernelGGL(HIP_KERNEL_NAME(k_export_runs_g<false>), grid, blk, 0, h->st, iv, (int64_t)0, ngrp, cnt8, (const uint64_t*)nullptr, (uint64_t*)nullptr);
	if ((r = scan_records(h, cnt8, ngrp, off8, (uint64_t*)h->misc.p + MISC_IX_TOT, total)) < 0) return r;
	const int64_t nr = (int64_t)total[0];
	if (nr <= 0) return RB3GPU_EINTERNAL;
	int64_t piece = h->tn.fmd_piece > 0 ? h->tn.fmd_piece : ((int64_t)64 << 20);
	if (piece < 32768) piece = 32768; // (a piece must hold a few chunks of the packer's speculation -- rb3fmd_enc_piece --, and it ends with a whole group: up to 8192 runs short)
	const bool one = nr <= piece;
	const int64_t cap = one ? nr + 16 : piece + RB3_GRP + 8192 + 1024; // (a piece ends with a whole group: at most 8192 runs more; + what the packer carries over)
	{ // the packer holds ~26 bytes per run of a piece beside the run starts (8): where that does not fit beside the merge scratch of the handle, the scratch goes (the next merge obtains it again)
		size_t fr = 0, tot = 0;
		if (hipMemGetInfo(&fr, &tot) == hipSuccess && (size_t)cap * 34 + ((size_t)64 << 20) > fr && !h->mg_active) { // (not inside a two-phase merge: its uncommitted index lives in the spare buffers)
			HIPCHK(hipStreamSynchronize(h->st));
			Buf *scratch[] = { &h->b2, &h->pos, &h->post, &h->tcnt, &h->tpre, &h->jg, &h->wl, &h->wls, &h->dl, &h->dlx, &h->wstat, &h->wplane, &h->wruns, &h->gslots, &h->glist, &h->pslots, &h->shc, &h->shn, &h->shs, &h->shr };
			for (Buf *b : scratch) buf_release(h, *b);
			ib_release(h, 1 - h->cur);
			garbage_collect(h, true);
			h->sid_dirty[0] = h->sid_dirty[1] = RB3_TENT_HALF; // (the stretch table is gone: a fresh one is cleared whole)
			if (h->opt.verbose >= 3) fprintf(stderr, "[M::%s::%.3f] %lld runs: the merge scratch of the handle released for the packer's tables\n", __func__, now_s() - h->t0, (long long)nr);
		} else (void)hipGetLastError();
	}
	rb3fmd_enc *enc = nullptr;
	int fr = rb3fmd_enc_begin(h->st, h->n, cap, &enc);
	if (fr == -1 && !h->garbage.empty()) { // no room for the packer's tables while replaced buffers are still held
		(void)hipGetLastError();
		garbage_collect(h, true);
		fr = rb3fmd_enc_begin(h->st, h->n, cap, &enc);
	}
	int64_t npieces = 0;
	if (fr == 0 && one) {
		hipLaunchKernelGGL(HIP_KERNEL_NAME(k_export_runs_g<true>), grid, blk, 0, h->st, iv, (int64_t)0, ngrp, cnt8, (const uint64_t*)off8, rb3fmd_enc_buffer(enc, nullptr));
		fr = rb3fmd_enc_piece(enc, nr, 1), npieces = 1;
	} else if (fr == 0) {
		std::vector<uint64_t> before((size_t)ngrp + 1);
		if ((r = buf_ensure(h, h->xbuf, (size_t)(ngrp + 1) * 8)) < 0) { rb3fmd_enc_abort(enc); return r; }
		hipLaunchKernelGGL(k_runs_before, dim3((unsigned)((ngrp + 1 + 255) / 256)), dim3(256), 0, h->st, (const uint64_t*)off8, ngrp, (uint64_t)nr, (uint64_t*)h->xbuf.p);
		if (hipMemcpyAsync(before.data(), h->xbuf.p, (size_t)(ngrp + 1) * 8, hipMemcpyDeviceToHost, h->st) != hipSuccess || hipStreamSynchronize(h->st) != hipSuccess) { (void)hipGetLastError(); rb3fmd_enc_abort(enc); return RB3GPU_ENODEV; }
		for (int64_t ga = 0; ga < ngrp && fr == 0; ++npieces) {
			int64_t room = 0;
			uint64_t *dst = rb3fmd_enc_buffer(enc, &room);
			if (room > piece + RB3_GRP) room = piece + RB3_GRP;
			// the groups [ga, gb) hold at most `room` runs and, if they can, at least `piece` of them
			int64_t lo = ga + 1, hi = ngrp;
			while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if ((int64_t)(before[(size_t)mid] - before[(size_t)ga]) <= room - RB3_GRP) lo = mid; else hi = mid - 1; }
			const int64_t gb = lo, nn = (int64_t)(before[(size_t)gb] - before[(size_t)ga]);
			if (nn > room) { fr = -3; break; }
			if (nn > 0)
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_export_runs_g<true>), dim3((unsigned)((gb - ga + 3) / 4)), blk, 0, h->st, iv, ga, gb - ga, cnt8 + ga * 8, (const uint64_t*)(off8 + ga * 8), dst - before[(size_t)ga]);
			fr = rb3fmd_enc_piece(enc, nn, gb == ngrp ? 1 : 0);
			ga = gb;
		}
	}
	if (fr == 0) fr = rb3fmd_enc_end(enc, words, n_words), enc = nullptr;
	if (enc) rb3fmd_enc_abort(enc);
	(void)hipGetLastError();
	h->stt.ms_export += (now_s() - t) * 1e3;
	if (fr == 1) return RB3GPU_EUNSUP;
	if (fr < 0) return fr == -1 ? RB3GPU_ENOMEM : fr == -2 ? RB3GPU_ENODEV : RB3GPU_EINTERNAL;
	if (h->opt.verbose >= 3)
		fprintf(stderr, "[M::%s::%.3f] packed %lld runs into %lld FMD words on the GPU in %.3f ms (%lld piece%s)\n", __func__, now_s() - h->t0, (long long)nr, (long long)*n_words, (now_s() - t) * 1e3, (long long)npieces, npieces == 1 ? "" : "s");
	return 0;
}

void rb3gpu_host_free(void *p) { free(p); }

void *rb3gpu_pinned_alloc(int64_t n_bytes)
{
	void *p = nullptr;
	if (n_bytes <= 0) return nullptr;
	if (hipHostMalloc(&p, (size_t)n_bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; } // (page-locked for every device: a multi-GPU build uploads from any of them)
	pthread_mutex_lock(&g_pin_mtx);
	g_pinned.push_back(std::make_pair((const uint8_t*)p, (size_t)n_bytes));
	pthread_mutex_unlock(&g_pin_mtx);
	return p;
}

void rb3gpu_pinned_free(void *p)
{
	if (!p) return;
	pthread_mutex_lock(&g_pin_mtx);
	for (size_t i = 0; i < g_pinned.size(); ++i)
		if (g_pinned[i].first == (const uint8_t*)p) { g_pinned[i] = g_pinned.back(); g_pinned.pop_back(); break; }
	pthread_mutex_unlock(&g_pin_mtx);
	(void)hipHostFree(p);
}

/* one call per run on top of the bulk export (the host turns consecutive starts into lengths) */
struct RunAdapter { rb3gpu_emit_f emit; void *data; int c; int64_t start; };

static int run_adapter(void *data, int64_t n, const uint64_t *words, int64_t end)
{
	RunAdapter *a = (RunAdapter*)data;
	for (int64_t i = 0; i < n; ++i) {
		const int64_t s = (int64_t)(words[i] >> 3);
		if (a->c >= 0 && a->emit(a->data, a->c, s - a->start) != 0) return -1;
		a->c = (int)(words[i] & 7), a->start = s;
	}
	if (end >= 0 && a->c >= 0 && a->emit(a->data, a->c, end - a->start) != 0) return -1;
	return 0;
}

int rb3gpu_export_runs(rb3gpu_t *h, rb3gpu_emit_f emit, void *data)
{
	if (!h || !emit) return RB3GPU_EINVAL;
	RunAdapter a = { emit, data, -1, 0 };
	return rb3gpu_export_run_words(h, run_adapter, &a);
}

/* fm-index.c:56-85 (rb3_enc_fmd2fmr) with the decoding on the device: the word stream of an FMD file -> symbols in HBM
 * (one thread per 64-byte block) -> rb3gpu_from_plain_dev */
static int fmd_words_to_b2(rb3gpu_t *h, int64_t n_words, const uint64_t *words, const int64_t mcnt[RB3GPU_ASIZE], int64_t *n_sym_out)
{
	const double t = now_s();
	int r;
	if ((r = buf_ensure(h, h->xbuf, (size_t)(n_words + 2) * 8)) < 0) return r;
	HIPCHK(hipMemsetAsync((uint64_t*)h->xbuf.p + n_words, 0, 16, h->st));
	HIPCHK(hipMemcpyAsync(h->xbuf.p, words, (size_t)n_words * 8, hipMemcpyHostToDevice, h->st));
	rb3fmd_dec *ctx = nullptr;
	int64_t n_sym = 0;
	r = rb3fmd_decode_begin(h->st, n_words, (const uint64_t*)h->xbuf.p, &ctx, &n_sym);
	if (r < 0) return r == -1 ? RB3GPU_ENOMEM : r == -2 ? RB3GPU_ENODEV : RB3GPU_ESYMBOL;
	if (mcnt) { // the header of the file says how many symbols there are
		int64_t tot = 0;
		for (int a = 0; a < RB3GPU_ASIZE; ++a) tot += mcnt[a];
		if (tot != n_sym) { (void)rb3fmd_decode_fill(ctx, nullptr); return RB3GPU_ESYMBOL; }
	}
	if ((r = buf_ensure(h, h->b2, (size_t)n_sym + 16)) < 0) { (void)rb3fmd_decode_fill(ctx, nullptr); return r; }
	r = rb3fmd_decode_fill(ctx, (uint8_t*)h->b2.p);
	if (r < 0) return RB3GPU_ENODEV;
	if (h->opt.verbose >= 3)
		fprintf(stderr, "[M::%s::%.3f] decoded %lld FMD words into %lld symbols on the GPU in %.3f ms\n", __func__, now_s() - h->t0, (long long)n_words, (long long)n_sym, (now_s() - t) * 1e3);
	*n_sym_out = n_sym;
	return 0;
}

/* A large FMD stream becomes a block array WITHOUT passing through one byte per symbol (rb3_enc_fmd2fmr streams the runs too,
 * fm-index.c:56-85): the stream is decoded chunk by chunk (load_chunk groups of 8192 symbols; a 64-byte FMD block is
 * self-contained, so any range of positions can be decoded on its own), twice -- the first pass only counts the slots and symbols
 * of every group (k_pass1w + k_decide on the chunk), one scan over all groups then gives every slot its place and the directory
 * its counts, the second pass writes the slots of each chunk there (k_pass1w + k_pass2w).  Device memory beyond the stream and the
 * index: one chunk of symbols and its window scratch (~0.23 GB for the default chunk of 134 M symbols). */
static int from_fmd_chunked(rb3gpu_t *h, rb3fmd_dec *ctx, int64_t n, const int64_t mcnt[RB3GPU_ASIZE])
{
	const int64_t ngrp = (n >> RB3_GRP_BITS) + 1, nwin = (n >> RB3_WIN_BITS) + 1, CG = h->tn.load_chunk;
	const int64_t csym = CG << RB3_GRP_BITS, cwin = CG * RB3_GRP_WINS;
	const int dst = 1 - h->cur;
	int r;
	if ((r = buf_ensure(h, h->b2, (size_t)csym + 16, true)) < 0) return r;
	if ((r = buf_ensure(h, h->wstat, (size_t)(cwin + 1) * 16, true, true)) < 0) return r;
	if ((r = buf_ensure(h, h->wplane, (size_t)(cwin + 1) * 96, true, true)) < 0) return r;
	if ((r = buf_ensure(h, h->wruns, (size_t)(cwin + 1) * RB3_RLE_CODES * 2, true, true)) < 0) return r;
	if ((r = buf_ensure(h, h->gstat, (size_t)ngrp * 32, true)) < 0) return r;
	if ((r = buf_ensure(h, h->gpre, (size_t)ngrp * 64, true)) < 0) return r;
	if ((r = buf_ensure(h, h->misc, MISC_WORDS * 8)) < 0) return r;
	uint32_t *gstat = (uint32_t*)h->gstat.p;
	uint64_t *gpre = (uint64_t*)h->gpre.p, *dtot = (uint64_t*)h->misc.p + MISC_IX_TOT, total[8];
	const IdxView none = view_of(h); // (not read when building from symbols)
	int64_t onslots = 0;
	HIPCHK(hipEventRecord(h->ev[0], h->st));
	for (int pass = 0; pass < 2; ++pass) {
		for (int64_t g0 = 0; g0 < ngrp; g0 += CG) {
			const int64_t g1 = g0 + CG < ngrp ? g0 + CG : ngrp, p0 = g0 << RB3_GRP_BITS, p1 = (g1 << RB3_GRP_BITS) < n ? (g1 << RB3_GRP_BITS) : n;
			const int64_t rem = n - p0;                                   // symbols from the chunk's start to the end of the index
			const int64_t nw = g1 == ngrp ? (rem >> RB3_WIN_BITS) + 1 : (g1 - g0) * RB3_GRP_WINS; // (the last chunk ends with the window of position n)
			if (rb3fmd_decode_range(ctx, p0, p1, (uint8_t*)h->b2.p) < 0) return RB3GPU_ENODEV;
			const dim3 g1w((unsigned)((nw + RB3_REB_WAVES * RB3_REB_WPW - 1) / (RB3_REB_WAVES * RB3_REB_WPW))), b1w(64 * RB3_REB_WAVES);
			hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass1w<true, 3>), g1w, b1w, 0, h->st, none, (const int64_t*)nullptr, (const uint8_t*)h->b2.p, rem, rem, (const int64_t*)nullptr,
					(uint4*)h->wstat.p, (uint32_t*)h->wplane.p, (uint16_t*)h->wruns.p, nw, (const unsigned long long*)nullptr, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
			if (pass == 0)
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_decide<false>), dim3((unsigned)(g1 - g0)), dim3(64), 0, h->st, (const uint4*)h->wstat.p, rem, gstat + g0 * 8, g1 - g0, (const unsigned long long*)nullptr,
						(const uint32_t*)nullptr, (const uint32_t*)nullptr);
			else
				hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2w<false>), dim3((unsigned)(g1 - g0)), dim3(64 * RB3_REB_WAVES), 0, h->st, (const uint4*)h->wstat.p, (const uint32_t*)h->wplane.p, (const uint16_t*)h->wruns.p, rem,
						(const uint32_t*)(gstat + g0 * 8), (const uint64_t*)(gpre + g0 * 8), (const uint64_t*)dtot, h->ib[dst].grp + g0, (uint4*)h->ib[dst].slots, nw, (const unsigned long long*)nullptr,
						(const uint32_t*)nullptr, (const uint32_t*)nullptr, 0u, ~0ull, nwin, n, h->tn.abs_limit);
		}
		if (pass == 0) {
			if ((r = scan_records(h, gstat, ngrp, gpre, dtot, total)) < 0) return r; // (one synchronisation: the slot count sizes the index)
			int64_t tot = 0;
			for (int a = 0; a < 6; ++a) {
				tot += (int64_t)total[a];
				if (mcnt && (int64_t)total[a] != mcnt[a]) return RB3GPU_ESYMBOL;
			}
			if (tot != n) return RB3GPU_ESYMBOL;
			onslots = (int64_t)total[6];
			index_drop(h);
			if ((r = ib_ensure(h, dst, ngrp, onslots, true)) < 0) return r; // (no head room: the first merge into a loaded index sizes its own buffers)
		}
	}
	hipLaunchKernelGGL(k_grp_compact, dim3((unsigned)((ngrp + 255) / 256)), dim3(256), 0, h->st, (const uint64_t*)h->ib[dst].grp, ngrp, (uint64_t*)(h->ib[dst].grp + h->ib[dst].grp_cap), (const unsigned long long*)nullptr);
	HIPCHK(hipEventRecord(h->ev[1], h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	h->stt.ms_build += ev_ms(h->ev[0], h->ev[1]);
	int64_t acc[7];
	acc[0] = 0;
	for (int a = 0; a < 6; ++a) acc[a + 1] = acc[a] + (int64_t)total[a];
	index_install(h, ngrp, onslots, n, acc);
	if (h->bytes_owned + rb3fmd_decode_bytes(ctx) > h->stt.bytes_peak) h->stt.bytes_peak = h->bytes_owned + rb3fmd_decode_bytes(ctx);
	if (h->opt.verbose >= 3)
		fprintf(stderr, "[M::%s::%.3f] built %lld symbols into %lld slots from the FMD stream in chunks of %lld symbols (%.3f ms)\n", __func__, now_s() - h->t0, (long long)n, (long long)onslots, (long long)csym, ev_ms(h->ev[0], h->ev[1]));
	return 0;
}

int rb3gpu_from_fmd_words(rb3gpu_t *h, int64_t n_words, const uint64_t *words, const int64_t mcnt[RB3GPU_ASIZE])
{
	if (!h || n_words < 8 || !words) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	int64_t n_sym = 0;
	int r;
	{ // an index of more than one chunk is never expanded to one byte per symbol
		if ((r = buf_ensure(h, h->xbuf, (size_t)(n_words + 2) * 8, true)) < 0) return r;
		HIPCHK(hipMemsetAsync((uint64_t*)h->xbuf.p + n_words, 0, 16, h->st));
		HIPCHK(hipMemcpyAsync(h->xbuf.p, words, (size_t)n_words * 8, hipMemcpyHostToDevice, h->st));
		rb3fmd_dec *ctx = nullptr;
		r = rb3fmd_decode_begin(h->st, n_words, (const uint64_t*)h->xbuf.p, &ctx, &n_sym);
		if (r < 0) return r == -1 ? RB3GPU_ENOMEM : r == -2 ? RB3GPU_ENODEV : RB3GPU_ESYMBOL;
		if (mcnt) {
			int64_t tot = 0;
			for (int a = 0; a < RB3GPU_ASIZE; ++a) tot += mcnt[a];
			if (tot != n_sym) { rb3fmd_decode_end(ctx); return RB3GPU_ESYMBOL; }
		}
		if (n_sym > (h->tn.load_chunk << RB3_GRP_BITS)) {
			r = from_fmd_chunked(h, ctx, n_sym, mcnt);
			rb3fmd_decode_end(ctx);
			if (r < 0) index_drop(h);
			return r;
		}
		rb3fmd_decode_end(ctx); // (small: the one-pass path below decodes it again, once)
	}
	if ((r = fmd_words_to_b2(h, n_words, words, mcnt, &n_sym)) < 0) return r;
	if ((r = rb3gpu_from_plain_dev(h, n_sym, (const uint8_t*)h->b2.p)) < 0) return r;
	if (mcnt)
		for (int a = 0; a < RB3GPU_ASIZE; ++a)
			if (h->acc[a + 1] - h->acc[a] != mcnt[a]) { index_drop(h); return RB3GPU_ESYMBOL; }
	return 0;
}

int rb3gpu_merge_fmd_words(rb3gpu_t *h, int64_t n_words, const uint64_t *words, const int64_t mcnt[RB3GPU_ASIZE])
{
	if (!h || n_words < 8 || !words) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->n <= 0) return RB3GPU_ESTATE;
	int64_t n_sym = 0;
	int r;
	if ((r = fmd_words_to_b2(h, n_words, words, mcnt, &n_sym)) < 0) return r; // (checks the total against the header)
	return merge_core(h, n_sym, (const uint8_t*)h->b2.p, 1, nullptr, nullptr, 0, 0, nullptr);
}

/* rb3_fmi_merge (fm-index.c:251-277) between two handles, which may sit on different GPUs of the node: the index of `src` is
 * decoded to its plain BWT on its own device, copied device to device (xGMI where the two have peer access) and merged into `h`
 * as one batch through the reference's signature (the walker list is made on the device).  `src` is left as it is. */
int rb3gpu_merge_index(rb3gpu_t *h, rb3gpu_t *src)
{
	if (!h || !src || h == src) return RB3GPU_EINVAL;
	if (h->n <= 0 || h->grp == nullptr || src->n <= 0 || src->grp == nullptr) return RB3GPU_ESTATE;
	const int64_t n = src->n;
	int r;
	void *d_src = nullptr;
	{ rb3gpu_t *h = src; HIPCHK(hipSetDevice(src->dev)); } // (HIPCHK reports through `h`)
	if (hipMalloc(&d_src, (size_t)n + 16) != hipSuccess) { (void)hipGetLastError(); return RB3GPU_ENOMEM; }
	if ((r = rb3gpu_export_plain_dev(src, (uint8_t*)d_src)) < 0) { (void)hipFree(d_src); return r; }
	hipError_t e = hipSetDevice(h->dev);
	if (e == hipSuccess && (r = buf_ensure(h, h->b2, (size_t)n + 16)) < 0) { (void)hipSetDevice(src->dev); (void)hipFree(d_src); return r; }
	if (e == hipSuccess && src->dev != h->dev) {
		int can = 0;
		if (hipDeviceCanAccessPeer(&can, h->dev, src->dev) == hipSuccess && can) { (void)hipDeviceEnablePeerAccess(src->dev, 0); (void)hipGetLastError(); } // (already enabled: fine)
		e = hipMemcpyPeerAsync(h->b2.p, h->dev, d_src, src->dev, (size_t)n, h->st);
	} else if (e == hipSuccess) e = hipMemcpyAsync(h->b2.p, d_src, (size_t)n, hipMemcpyDeviceToDevice, h->st);
	if (e == hipSuccess) e = hipStreamSynchronize(h->st);
	(void)hipSetDevice(src->dev);
	(void)hipFree(d_src);
	HIPCHK(e);
	HIPCHK(hipSetDevice(h->dev));
	return merge_core(h, n, (const uint8_t*)h->b2.p, 1, nullptr, nullptr, 0, 0, nullptr);
}

int rb3gpu_from_runs(rb3gpu_t *h, int64_t n_runs, const uint64_t *runs)
{
	if (!h || n_runs <= 0 || !runs) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	int64_t tot = 0;
	for (int64_t i = 0; i < n_runs; ++i) {
		if ((runs[i] & 7) > 5) return RB3GPU_ESYMBOL;
		tot += (int64_t)(runs[i] >> 3);
	}
	if (tot <= 0) return RB3GPU_EINVAL;
	int r;
	if ((r = buf_ensure(h, h->b2, (size_t)tot + 16)) < 0) return r;
	const int64_t cap = RB3_XCHUNK;
	uint8_t *buf = (uint8_t*)malloc((size_t)cap);
	if (!buf) return RB3GPU_ENOMEM;
	int64_t off = 0, fill = 0;
	for (int64_t i = 0; i < n_runs; ++i) {
		int64_t l = (int64_t)(runs[i] >> 3);
		const int c = (int)(runs[i] & 7);
		while (l > 0) {
			int64_t t = l < cap - fill ? l : cap - fill;
			memset(buf + fill, c, (size_t)t);
			fill += t, l -= t;
			if (fill == cap) {
				if (hipMemcpy((uint8_t*)h->b2.p + off, buf, (size_t)fill, hipMemcpyHostToDevice) != hipSuccess) { free(buf); return RB3GPU_ENODEV; }
				off += fill, fill = 0;
			}
		}
	}
	if (fill > 0 && hipMemcpy((uint8_t*)h->b2.p + off, buf, (size_t)fill, hipMemcpyHostToDevice) != hipSuccess) { free(buf); return RB3GPU_ENODEV; }
	free(buf);
	return rb3gpu_from_plain_dev(h, tot, (const uint8_t*)h->b2.p);
}

/* ---- interval-sharded index (north_star): see k_sh_step ---- */

int rb3gpu_sh_step(rb3gpu_t *h, int64_t n_states, const rb3gpu_state_t *d_in, const uint64_t *d_tw, int64_t *d_ka, const int64_t adj[RB3GPU_ASIZE],
		int n_iv, const int64_t *iv_bounds, int my_iv, rb3gpu_state_t *d_send, int64_t *counts)
{
	if (!h || n_states < 0 || !d_tw || !d_ka || !adj || !iv_bounds || !counts || n_iv < 1 || n_iv > RB3_SH_MAXIV || my_iv < 0 || my_iv >= n_iv) return RB3GPU_EINVAL;
	if (n_states > 0 && (!d_in || !d_send)) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->grp == nullptr) return RB3GPU_ESTATE;
	for (int i = 0; i <= n_iv; ++i) counts[i] = 0;
	if (n_states == 0) return 0;
	int r;
	// scratch: new states + destinations, counters (n_iv + 1), cursors (n_iv + 1), bad
	const size_t cnt_off = ((size_t)n_states * 20 + 255) & ~(size_t)255;
	if ((r = buf_ensure(h, h->xbuf, cnt_off + (size_t)(2 * RB3_SH_MAXIV + 4) * 8)) < 0) return r;
	ShState *d_out = (ShState*)h->xbuf.p;
	int32_t *d_dest = (int32_t*)(d_out + n_states);
	unsigned long long *d_cnt = (unsigned long long*)((char*)h->xbuf.p + cnt_off), *d_cur = d_cnt + RB3_SH_MAXIV + 1, *d_bad = d_cur + RB3_SH_MAXIV + 1;
	ShArgs a;
	memset(&a, 0, sizeof(a));
	for (int c = 0; c < 6; ++c) a.adj[c] = adj[c];
	for (int i = 0; i <= n_iv; ++i) a.bounds[i] = iv_bounds[i];
	a.iv_start = iv_bounds[my_iv], a.n_iv = n_iv;
	HIPCHK(hipMemsetAsync(d_cnt, 0, (size_t)(2 * RB3_SH_MAXIV + 4) * 8, h->st));
	HIPCHK(hipEventRecord(h->ev[0], h->st));
	int64_t nblk = (n_states * 8 + 255) / 256;
	if (nblk > 256 * 16) nblk = 256 * 16;
	hipLaunchKernelGGL(k_sh_step, dim3((unsigned)nblk), dim3(256), 0, h->st, view_of(h), a, n_states, (const ShState*)d_in, d_tw, d_ka, d_out, d_dest, d_cnt, d_bad);
	unsigned long long hc[RB3_SH_MAXIV + 1], hbad = 0;
	HIPCHK(hipMemcpyAsync(hc, d_cnt, (size_t)(n_iv + 1) * 8, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipMemcpyAsync(&hbad, d_bad, 8, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	if (hbad != 0) return RB3GPU_EINTERNAL;
	ShOffsets o;
	memset(&o, 0, sizeof(o));
	o.n_iv = n_iv;
	int64_t tot = 0;
	for (int i = 0; i <= n_iv; ++i) {
		counts[i] = (int64_t)hc[i];
		o.off[i] = tot;
		if (i < n_iv) tot += counts[i];
	}
	if (tot + counts[n_iv] != n_states) return RB3GPU_EINTERNAL;
	if (tot > 0)
		hipLaunchKernelGGL(k_sh_scatter, dim3((unsigned)((n_states + 255) / 256)), dim3(256), 0, h->st, o, n_states, (const ShState*)d_out, (const int32_t*)d_dest, (ShState*)d_send, d_cur);
	HIPCHK(hipEventRecord(h->ev[1], h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	h->stt.ms_rank += ev_ms(h->ev[0], h->ev[1]);
	h->stt.n_lf_steps += n_states, h->stt.n_rank_launches += 1;
	return 0;
}

/* (No sampled LF check here, unlike every other merge path: k_lf_check guards against a wrong-but-monotone pos[] out of SPECULATIVE
 * records, and the sharded walk has none -- every ka is the exact result of its chain's previous step -- while the relation it tests,
 * ka[LF2(kb)] = C1[c] + rank_B1(c, ka[kb]), needs the ranks of ALL intervals, which no single GPU holds.  Completeness and order of
 * the interval's rows are checked (k_sh_localpos, k_pos_check), and the symbol totals after the rebuild.) */
int rb3gpu_sh_finish(rb3gpu_t *h, int64_t jlo, int64_t n_rows, const uint8_t *d_bwt, const int64_t *d_ka, int64_t iv_start, int commit)
{
	if (!h || jlo < 0 || n_rows < 0 || !d_bwt || !d_ka) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (h->grp == nullptr) return RB3GPU_ESTATE;
	if (n_rows == 0) return 0; // nothing landed in this interval
	const int64_t ntot = h->n + n_rows;
	int r;
	if ((r = buf_ensure(h, h->pos, (size_t)n_rows * 8)) < 0) return r;
	if ((r = buf_ensure(h, h->misc, MISC_WORDS * 8)) < 0) return r;
	unsigned long long *misc = (unsigned long long*)h->misc.p;
	int64_t *dpos = (int64_t*)h->pos.p;
	HIPCHK(hipMemsetAsync(misc, 0, 128, h->st));
	HIPCHK(hipEventRecord(h->ev[2], h->st));
	hipLaunchKernelGGL(k_sh_localpos, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, h->st, jlo, n_rows, d_ka, iv_start, dpos, misc + 2);
	hipLaunchKernelGGL(k_pos_check, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, h->st, (const int64_t*)dpos, n_rows, ntot, misc + 2);
	unsigned long long hm[5] = {0, 0, 0, 0, 0};
	HIPCHK(hipMemcpyAsync(hm, misc, 40, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	if (hm[2] != 0 || hm[3] != 0) {
		if (h->opt.verbose >= 1) fprintf(stderr, "[E::rb3gpu] sharded merge: %llu rows of the interval unset or misrouted, %llu out of order\n", hm[2], hm[3]);
		return RB3GPU_EINTERNAL;
	}
	int64_t ngrp = 0, nslots = 0, acc[7];
	if ((r = build_index<false>(h, n_rows, d_bwt + jlo, (const int64_t*)dpos, ntot, false, &ngrp, &nslots, acc)) < 0) return r;
	HIPCHK(hipEventRecord(h->ev[3], h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	h->stt.ms_build += ev_ms(h->ev[2], h->ev[3]);
	h->stt.n_symbols_merged += n_rows;
	h->stt.bytes_rebuild += 9 * n_rows + h->stt.bytes_index + ngrp * (int64_t)sizeof(rb3_grp_t) + nslots * (int64_t)sizeof(rb3_slot_t);
	if (commit) index_install(h, ngrp, nslots, ntot, acc);
	return 0;
}

/* like buf_ensure, but the first `used` bytes of the buffer survive a reallocation */
static int buf_grow_keep(rb3gpu_t *h, Buf &b, size_t used, size_t bytes)
{
	if (b.cap >= bytes && b.p) return 0;
	if (vm_find(h, b.p)) return buf_ensure(h, b, bytes); // (grows where it stands, contents and all)
	Buf nb;
	int r = buf_ensure(h, nb, bytes);
	if (r < 0) return r;
	if (b.p && used) HIPCHK(hipMemcpyAsync(nb.p, b.p, used, hipMemcpyDeviceToDevice, h->st));
	if (b.p) dev_free(h, b.p, b.cap); // (given back by a hipFree, which waits for the copy)
	b = nb;
	return 0;
}

/* the rows [jlo, jlo + n_rows) of the batch with their positions inside this interval: checked, interleaved, index rebuilt */
static int sh_rebuild(rb3gpu_t *h, int64_t n_rows, const uint8_t *d_b2rows, int64_t *dpos, unsigned long long *misc, int commit)
{
	const int64_t ntot = h->n + n_rows;
	hipLaunchKernelGGL(k_pos_check, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, h->st, (const int64_t*)dpos, n_rows, ntot, misc + 2);
	unsigned long long hm[5] = {0, 0, 0, 0, 0};
	HIPCHK(hipMemcpyAsync(hm, misc, 40, hipMemcpyDeviceToHost, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	if (hm[2] != 0 || hm[3] != 0) {
		if (h->opt.verbose >= 1) fprintf(stderr, "[E::rb3gpu] sharded merge: %llu rows of the interval unset or misrouted, %llu out of order\n", hm[2], hm[3]);
		return RB3GPU_EINTERNAL;
	}
	int r;
	int64_t ngrp = 0, nslots = 0, acc[7];
	if ((r = build_index<false>(h, n_rows, d_b2rows, (const int64_t*)dpos, ntot, false, &ngrp, &nslots, acc)) < 0) return r;
	HIPCHK(hipEventRecord(h->ev[3], h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	h->stt.ms_build += ev_ms(h->ev[2], h->ev[3]);
	h->stt.n_symbols_merged += n_rows;
	h->stt.bytes_rebuild += 9 * n_rows + h->stt.bytes_index + ngrp * (int64_t)sizeof(rb3_grp_t) + nslots * (int64_t)sizeof(rb3_slot_t);
	if (commit) index_install(h, ngrp, nslots, ntot, acc);
	return 0;
}

/* d_tprev != NULL: the batch is SHARDED -- this rank holds the symbol before every text position (d_tprev, one byte each, the only thing a
 * step needs of the batch) and the text-order words of its own text range [t_lo, t_hi) only (d_tw = the slice; text ranges: len * q / world);
 * the records are (text position, insertion point) and the rows come from the owners of the text positions at the end: one all-to-all of
 * the records to the owners, one back with row << 3 | symbol in place of the text position (d_bwt is not read). */
static int sh_merge_impl(rb3gpu_t *h, const rb3gpu_comm_t *comm, int64_t *iv_bounds, int64_t len, const uint8_t *d_bwt, const uint64_t *d_tw,
		int64_t n_chains, const int64_t *chain_tp, int commit, int64_t *n_rounds, const uint8_t *d_tprev = nullptr, int64_t t_lo = 0, int64_t t_hi = 0)
{
	const int world = comm->world, rank = comm->rank;
	HIPCHK(hipSetDevice(h->dev));
	if (h->grp == nullptr) return RB3GPU_ESTATE;
	if (iv_bounds[rank + 1] - iv_bounds[rank] != h->n) return RB3GPU_EINVAL; // not the interval this handle holds
	int r;
	// symbol totals of every interval -> C array of the whole BWT and this interval's additive offsets (mrope.c:76-88)
	std::vector<int64_t> mine(6), allc((size_t)world * 6), M((size_t)world * world), cnt_h(world), recv_cnt(world), rows_all(world);
	for (int c = 0; c < 6; ++c) mine[c] = h->acc[c + 1] - h->acc[c];
	if ((r = comm->all_gather(comm->ctx, mine.data(), 6, allc.data())) < 0) return r;
	ShArgs a;
	memset(&a, 0, sizeof(a));
	int64_t tot[6], C = 0;
	for (int c = 0; c < 6; ++c) {
		int64_t pre = 0;
		tot[c] = 0;
		for (int q = 0; q < world; ++q) { if (q < rank) pre += allc[(size_t)q * 6 + c]; tot[c] += allc[(size_t)q * 6 + c]; }
		a.adj[c] = C + pre - h->acc[c];
		C += tot[c];
	}
	for (int i = 0; i <= world; ++i) a.bounds[i] = iv_bounds[i];
	a.iv_start = iv_bounds[rank], a.n_iv = world;
	const int64_t m1 = tot[0]; // every chain starts at ka = #sentinels of the index (fm-index.c:164)
	int owner0 = 0;
	for (int i = 1; i < world; ++i) owner0 += iv_bounds[i] <= m1 ? 1 : 0;
	// two state buffers for the whole merge (a rank never holds more states than there are strings), counters of both parities + bad
	if ((r = buf_ensure(h, h->shc, (size_t)n_chains * 16)) < 0) return r;
	if ((r = buf_ensure(h, h->shn, (size_t)n_chains * 16)) < 0) return r;
	// three sets of counters (two for the rounds the host drives, three for the rounds that run back to back) + bad, 1 KB apart: a round that reads its
	// number of states from the set before while every block adds to its own set must not find the two in one cache line (with the sets 72 bytes apart every
	// block's first load queued behind the atomics of its own launch: 247 us per round of 2 M chains instead of 156)
#define RB3_SH_CNT_STRIDE 128
#define RB3_SH_CNT_WORDS (4 * RB3_SH_CNT_STRIDE)
	static_assert(RB3_SH_CNT_STRIDE >= RB3_SH_MAXIV + 1, "a set of counters per KB");
	if ((r = buf_ensure(h, h->shk, (size_t)RB3_SH_CNT_WORDS * 8)) < 0) return r;
	unsigned long long *d_cnt[3] = { (unsigned long long*)h->shk.p, (unsigned long long*)h->shk.p + RB3_SH_CNT_STRIDE, (unsigned long long*)h->shk.p + 2 * RB3_SH_CNT_STRIDE }, *d_bad = (unsigned long long*)h->shk.p + 3 * RB3_SH_CNT_STRIDE;
	HIPCHK(hipMemsetAsync(h->shk.p, 0, (size_t)RB3_SH_CNT_WORDS * 8, h->st));
	ShState *cur = (ShState*)h->shc.p, *nxt = (ShState*)h->shn.p;
	int64_t n_cur = 0;
	for (int64_t i = 0; i < n_chains; ++i)
		if (chain_tp[i] < 0 || chain_tp[i] >= len) return RB3GPU_EINVAL;
	ShState *pr0 = nullptr, *pr1 = nullptr; // peer rounds: this rank's two receive buffers
	auto first_states = [&](ShState *dst) -> int { // every chain starts on the rank that owns ka = m1
		n_cur = 0;
		if (rank != owner0) return 0;
		std::vector<ShState> st((size_t)n_chains);
		for (int64_t i = 0; i < n_chains; ++i) st[(size_t)i].tp = chain_tp[i], st[(size_t)i].ka = m1;
		HIPCHK(hipMemcpyAsync(dst, st.data(), (size_t)n_chains * 16, hipMemcpyHostToDevice, h->st));
		HIPCHK(hipStreamSynchronize(h->st));
		n_cur = n_chains;
		return 0;
	};
	auto longest_string = [&]() -> int64_t { // rounds = symbols of the longest string, its sentinel included
		int64_t longest = 0, prev = -1;
		bool sorted = true; // (the usual caller hands the sentinels over in text order: one pass -- sorting 2 M of them took longer than fifty rounds of the walk, with the device idle)
		for (int64_t i = 0; i < n_chains && sorted; ++i) { sorted = chain_tp[i] > prev; if (chain_tp[i] - prev > longest) longest = chain_tp[i] - prev; prev = chain_tp[i]; }
		if (sorted) return longest;
		std::vector<int64_t> tp(chain_tp, chain_tp + n_chains);
		std::sort(tp.begin(), tp.end());
		longest = 0, prev = -1;
		for (int64_t i = 0; i < n_chains; ++i) { if (tp[(size_t)i] - prev > longest) longest = tp[(size_t)i] - prev; prev = tp[(size_t)i]; }
		return longest;
	};
	// PEER ROUNDS (several intervals whose ranks address each other's memory -- rb3gpu_comm_t.stream_barrier --, up to RB3_SH_MAXPEER of them): a round is
	// ONE kernel per rank that writes the next states straight into the owners' receive buffers, into the region each buffer keeps for this source,
	// at places from this rank's own cursors, and leaves its totals in the owners' tables of incoming counts (k_sh_round, ShPeers); every rank
	// queues all its rounds, the streams wait for each other's events between them, the host looks at the result once.  What the host used to do
	// per round -- a read-back of the split sizes with a synchronisation, an all-gather, an all-to-all of peer copies with two more
	// synchronisations and three barriers -- is gone, and nothing that returns a value crosses a link (VERDICT r5 "next" 8).  The ranks agree on
	// the path first; the receive buffers then hold a region of n_chains states per source.
	bool peer = false;
	static const bool ipc_dbg = getenv("RB3GPU_IPC_DEBUG") != nullptr; // (read once per process: where does a sharded merge between processes stand?)
#define RB3_IPC_DBG(...) do { if (ipc_dbg) { fprintf(stderr, "[ipc debug] rank %d: ", rank); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)
	std::vector<int64_t> all4((size_t)world * 4);
	if (world > 1) {
		int64_t can = comm->stream_barrier != nullptr && world <= RB3_SH_MAXPEER && h->tn.sh_host_rounds <= 0 ? 1 : 0;
		// (a rank that must replace its receive buffers -- more chains than any batch before -- says so first: between processes the others give their mappings of them up
		// before they go; a mapping of a buffer that its owner has freed is what a later merge must never meet)
		const size_t pneed = (size_t)world * (size_t)n_chains * 16;
		int64_t cg[2] = { can, can && (h->shp0.p == nullptr || h->shp0.cap < pneed || h->shp1.p == nullptr || h->shp1.cap < pneed) ? 1 : 0 };
		std::vector<int64_t> can_all((size_t)world * 2);
		RB3_IPC_DBG("merge of %lld symbols, %lld chains: can %lld, new buffers %lld", (long long)len, (long long)n_chains, (long long)can, (long long)cg[1]);
		if ((r = comm->all_gather(comm->ctx, cg, 2, can_all.data())) < 0) return r;
		peer = true;
		for (int q = 0; q < world; ++q) peer = peer && can_all[(size_t)q * 2] != 0;
		if (peer && comm->peer_import) {
			bool any = false;
			for (int q = 0; q < world; ++q)
				if (can_all[(size_t)q * 2 + 1] != 0) { any = true; if (q != rank) (void)comm->peer_import(comm->ctx, q, nullptr); }
			if (any && (r = comm->stream_barrier(comm->ctx, (void*)h->st)) < 0) return r; // (everybody has let go)
		}
		if (peer) {
			if (buf_ensure(h, h->shp0, (size_t)world * (size_t)n_chains * 16) < 0 || buf_ensure(h, h->shp1, (size_t)world * (size_t)n_chains * 16) < 0) can = 0; // (no room: the others must hear of it)
			pr0 = (ShState*)h->shp0.p, pr1 = (ShState*)h->shp1.p;
			RB3_IPC_DBG("agreed (%d), buffers %p %p %p", (int)peer, h->shp0.p, h->shp1.p, h->shk.p);
			// the three buffers the other ranks write into: as pointers (threads of one process), or as the handles the communicator makes of them (processes: HIP IPC)
			int64_t mine25[25];
			memset(mine25, 0, sizeof(mine25));
			void *mybuf[3] = { h->shp0.p, h->shp1.p, h->shk.p };
			for (int i = 0; i < 3 && can; ++i) {
				if (comm->peer_export) { if (mybuf[i] == nullptr || comm->peer_export(comm->ctx, mybuf[i], mine25 + 1 + 8 * i) < 0) can = 0; }
				else mine25[1 + 8 * i] = (int64_t)(intptr_t)mybuf[i];
			}
			mine25[0] = can;
			std::vector<int64_t> all25((size_t)world * 25);
			RB3_IPC_DBG("exported (can %lld)", (long long)can);
			if ((r = comm->all_gather(comm->ctx, mine25, 25, all25.data())) < 0) return r;
			RB3_IPC_DBG("handles gathered");
			for (int q = 0; q < world; ++q) peer = peer && all25[(size_t)q * 25] != 0;
			int64_t ok = 1;
			for (int q = 0; q < world && peer; ++q) { // (processes: the buffers of ONE rank at a time are mapped by the others while that rank waits -- two processes opening each other's handles at the same moment did not come back)
				for (int i = 0; i < 3; ++i) {
					void *ptr = q == rank ? mybuf[i] : comm->peer_import ? comm->peer_import(comm->ctx, q, all25.data() + (size_t)q * 25 + 1 + 8 * i) : (void*)(intptr_t)all25[(size_t)q * 25 + 1 + 8 * i];
					if (ptr == nullptr) ok = 0;
					all4[(size_t)q * 4 + 1 + i] = (int64_t)(intptr_t)ptr;
				}
				if (comm->peer_import && (r = comm->stream_barrier(comm->ctx, (void*)h->st)) < 0) return r;
			}
			RB3_IPC_DBG("imported (ok %lld)", (long long)ok);
			if (peer && comm->peer_import) { // (a rank that could not map a buffer of another one: everybody hears of it)
				std::vector<int64_t> oks((size_t)world);
				if ((r = comm->all_gather(comm->ctx, &ok, 1, oks.data())) < 0) return r;
				for (int q = 0; q < world; ++q) peer = peer && oks[(size_t)q] != 0;
			}
			if (peer && (!pr0 || !pr1)) return RB3GPU_ENOMEM;
			RB3_IPC_DBG("buffers exchanged: peer %d", (int)peer);
		}
	}
	if ((r = first_states(peer ? pr0 : cur)) < 0) return r;
	RB3_IPC_DBG("first states up");
	unsigned long long hc_stack[RB3_SH_MAXIV + 2], *hc = h->hm_pin ? h->hm_pin : hc_stack;
	const IdxView iv = view_of(h);
	int64_t rows = 0, rounds = 0;
	int par = 0;
	bool dirty[2] = { false, false };
	HIPCHK(hipEventRecord(h->ev[0], h->st));
	bool walked = false;
	if (peer) {
		const int64_t longest = longest_string();
		int64_t rec_cap = 3 * (len / world) + (1 << 16); // (nobody knows beforehand how many rows land in an interval: three times its share; a round that would overrun it says so)
		if (rec_cap > len) rec_cap = len;
		if ((r = buf_ensure(h, h->shr, (size_t)rec_cap * 16)) < 0) return r;
		if ((r = buf_ensure(h, h->xbuf, (size_t)(longest + 3) * 8)) < 0) return r;
		unsigned long long *rb = (unsigned long long*)h->xbuf.p;
		// behind the three sets of cursors and the two words of d_bad: the count of finished blocks, the incoming counts of both parities
#define RB3_SH_DONE_WORD (3 * RB3_SH_CNT_STRIDE + 8)
#define RB3_SH_CIN_WORD(par) (3 * RB3_SH_CNT_STRIDE + 16 + 16 * (par))
		static_assert(RB3_SH_CIN_WORD(1) + RB3_SH_MAXPEER <= RB3_SH_CNT_WORDS, "the counters of the peer rounds fit the buffer");
		unsigned long long *shk = (unsigned long long*)h->shk.p;
		HIPCHK(hipMemsetAsync(rb, 0, 8, h->st));
		{ const unsigned long long n0 = (unsigned long long)n_cur; HIPCHK(hipMemcpyAsync(shk + RB3_SH_CIN_WORD(1), &n0, 8, hipMemcpyHostToDevice, h->st)); HIPCHK(hipStreamSynchronize(h->st)); } // round 0 finds the first states in region 0, "from rank 0"
		RB3_IPC_DBG("%lld rounds, room for %lld records", (long long)longest, (long long)rec_cap);
		if ((r = comm->stream_barrier(comm->ctx, (void*)h->st)) < 0) return r; // (nobody writes into a table that its owner has not cleared yet)
		const int S = h->tn.sh_states ? h->tn.sh_states : n_chains >= ((int64_t)1 << 15) ? 8 : 1;
		const int BS = S == 8 && (h->tn.sh_block ? h->tn.sh_block == 1024 : n_chains < ((int64_t)3 << 20)) ? 1024 : 256;
		const unsigned nblk = (unsigned)((n_chains + (BS / 8) * S - 1) / ((BS / 8) * S));
		for (int64_t k = 0; k < longest; ++k) {
			ShPeers pe;
			memset(&pe, 0, sizeof(pe));
			pe.on = 1, pe.rank = rank, pe.rec_cap = rec_cap, pe.stride = n_chains;
			pe.cin_mine = shk + RB3_SH_CIN_WORD((k + 1) & 1), pe.done = shk + RB3_SH_DONE_WORD;
			for (int q = 0; q < world; ++q) { // round k reads the buffers k & 1 (0: shc) and fills the other ones, whose counts are the tables k & 1
				pe.dst[q] = (ShState*)(intptr_t)all4[(size_t)q * 4 + ((k & 1) ? 1 : 2)];
				pe.cin[q] = (unsigned long long*)(intptr_t)all4[(size_t)q * 4 + 3] + RB3_SH_CIN_WORD(k & 1);
			}
			const ShState *in = (k & 1) ? pr1 : pr0;
			unsigned long long *c_add = d_cnt[k % 3], *c_clr = d_cnt[(k + 1) % 3];
#define RB3_SH_ROUNDP(SS) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sh_round<SS>), dim3(nblk), dim3(256), 0, h->st, iv, a, n_chains, in, d_tw, (ShRec*)h->shr.p, (ShState*)nullptr, (int64_t)0, c_add, c_clr, d_bad, d_tprev, (const unsigned long long*)nullptr, rb + k, pe)
			if (BS == 1024) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sh_round<8, 1024>), dim3(nblk), dim3(1024), 0, h->st, iv, a, n_chains, in, d_tw, (ShRec*)h->shr.p, (ShState*)nullptr, (int64_t)0, c_add, c_clr, d_bad, d_tprev, (const unsigned long long*)nullptr, rb + k, pe);
			else if (S == 8) RB3_SH_ROUNDP(8); else if (S == 4) RB3_SH_ROUNDP(4); else if (S == 2) RB3_SH_ROUNDP(2); else RB3_SH_ROUNDP(1);
#undef RB3_SH_ROUNDP
			if ((r = comm->stream_barrier(comm->ctx, (void*)h->st)) < 0) return r;
		}
		RB3_IPC_DBG("rounds queued");
		unsigned long long fin[4 + RB3_SH_MAXPEER];
		memset(fin, 0, sizeof(fin));
		HIPCHK(hipMemcpyAsync(&fin[0], rb + longest, 8, hipMemcpyDeviceToHost, h->st));
		HIPCHK(hipMemcpyAsync(&fin[2], d_bad, 16, hipMemcpyDeviceToHost, h->st));
		HIPCHK(hipMemcpyAsync(&fin[4], shk + RB3_SH_CIN_WORD((longest + 1) & 1), (size_t)world * 8, hipMemcpyDeviceToHost, h->st)); // states that arrived behind the last round: none
		HIPCHK(hipStreamSynchronize(h->st));
		for (int q = 0; q < world; ++q) fin[1] += fin[4 + q];
		int64_t ok2[2] = { fin[1] == 0 && fin[2] == 0 && fin[3] == 0 ? 1 : 0, (int64_t)fin[0] }, tot_rows = 0;
		std::vector<int64_t> ok_all((size_t)world * 2);
		if ((r = comm->all_gather(comm->ctx, ok2, 2, ok_all.data())) < 0) return r;
		bool ok = true;
		for (int q = 0; q < world; ++q) ok = ok && ok_all[(size_t)q * 2] != 0, tot_rows += ok_all[(size_t)q * 2 + 1];
		RB3_IPC_DBG("rounds done: %llu rows here, ok %d, %lld rows in all", fin[0], (int)ok, (long long)tot_rows);
		if (ok && tot_rows == len) {
			walked = true, rows = (int64_t)fin[0], rounds = longest;
			h->stt.n_lf_steps += rows, h->stt.n_rank_launches += longest, h->stt.n_peer_rounds += longest;
		} else { // (an interval that takes more than three times its share of the rows, or something wrong: the rounds driven by the host find out which)
			if (h->opt.verbose >= 2) fprintf(stderr, "[W::rb3gpu] sharded merge, rank %d: peer rounds gave up (%llu rows recorded here of %lld in all, room for %lld; %llu states left, %llu misrouted, %llu rounds without room); once more with the rounds driven by the host\n",
					rank, fin[0], (long long)len, (long long)rec_cap, fin[1], fin[2], fin[3]);
			HIPCHK(hipMemsetAsync(h->shk.p, 0, (size_t)RB3_SH_CNT_WORDS * 8, h->st));
			if ((r = first_states(cur)) < 0) return r;
			if ((r = comm->stream_barrier(comm->ctx, (void*)h->st)) < 0) return r; // (every rank is done with the peer rounds' buffers)
		}
	}
	if (walked) {
	} else if (world == 1 && h->tn.sh_host_rounds <= 0 && (n_chains <= ((int64_t)1 << 19) || h->tn.sh_host_rounds < 0)) {
		const int64_t longest = longest_string();
		if ((r = buf_ensure(h, h->shr, (size_t)len * 16)) < 0) return r;
		if ((r = buf_ensure(h, h->xbuf, (size_t)(longest + 3) * 8)) < 0) return r;
		unsigned long long *rb = (unsigned long long*)h->xbuf.p;
		HIPCHK(hipMemsetAsync(rb, 0, 8, h->st));
		// three sets of counters in turn: round k adds to set k % 3, takes its number of states from set (k - 1) % 3 (what the round before counted for
		// interval 0) and clears set (k + 1) % 3 for the round behind it -- three different sets, so no block reads a word another block of the same launch clears
		{ const unsigned long long n0 = (unsigned long long)n_chains; HIPCHK(hipMemcpyAsync(d_cnt[2], &n0, 8, hipMemcpyHostToDevice, h->st)); HIPCHK(hipStreamSynchronize(h->st)); } // "the round before" of round 0 (n0 lives on this stack)
		// (states per octet: eight from 2^15 chains on -- round 5, profiles/r5_sh_states.txt: 200 k chains 38.7 us per round with four, 32.7 with eight; sixteen are no faster; one for few chains)
		const int S = h->tn.sh_states ? h->tn.sh_states : n_chains >= ((int64_t)1 << 15) ? 8 : 1;
		const int BS = S == 8 && (h->tn.sh_block ? h->tn.sh_block == 1024 : n_chains < ((int64_t)3 << 20)) ? 1024 : 256; // (round 5, profiles/r5_sh_states.txt: 200 k chains 32.6 -> 27.2 us per round, 2 M 166 -> 159, 4 M 278 -> 297)
		const unsigned nblk = (unsigned)((n_chains + (BS / 8) * S - 1) / ((BS / 8) * S));
		ShState *sa = cur, *sb = nxt;
		for (int64_t k = 0; k < longest; ++k) {
			unsigned long long *c_add = d_cnt[k % 3], *c_n = d_cnt[(k + 2) % 3], *c_clr = d_cnt[(k + 1) % 3];
#define RB3_SH_ROUND1(SS) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sh_round<SS>), dim3(nblk), dim3(256), 0, h->st, iv, a, n_chains, (const ShState*)sa, d_tw, (ShRec*)h->shr.p, sb, n_chains, c_add, c_clr, d_bad, d_tprev, (const unsigned long long*)c_n, rb + k)
			if (BS == 1024) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sh_round<8, 1024>), dim3(nblk), dim3(1024), 0, h->st, iv, a, n_chains, (const ShState*)sa, d_tw, (ShRec*)h->shr.p, sb, n_chains, c_add, c_clr, d_bad, d_tprev, (const unsigned long long*)c_n, rb + k);
			else if (S == 8) RB3_SH_ROUND1(8); else if (S == 4) RB3_SH_ROUND1(4); else if (S == 2) RB3_SH_ROUND1(2); else RB3_SH_ROUND1(1);
#undef RB3_SH_ROUND1
			ShState *t = sa; sa = sb, sb = t;
		}
		unsigned long long fin[3] = {0, 0, 0};
		HIPCHK(hipMemcpyAsync(&fin[0], rb + longest, 8, hipMemcpyDeviceToHost, h->st));
		HIPCHK(hipMemcpyAsync(&fin[1], d_cnt[(longest + 2) % 3], 8, hipMemcpyDeviceToHost, h->st)); // states left after the last round: none
		HIPCHK(hipMemcpyAsync(&fin[2], d_bad, 8, hipMemcpyDeviceToHost, h->st));
		HIPCHK(hipStreamSynchronize(h->st));
		if (fin[2] != 0 || fin[1] != 0 || (int64_t)fin[0] != len) {
			if (h->opt.verbose >= 1) fprintf(stderr, "[E::rb3gpu] sharded merge (one interval, %lld rounds on the device): %llu of %lld rows recorded, %llu states left, %llu misrouted\n", (long long)longest, fin[0], (long long)len, fin[1], fin[2]);
			return RB3GPU_EINTERNAL;
		}
		rows = len, rounds = longest;
		h->stt.n_lf_steps += len, h->stt.n_rank_launches += longest;
	} else
	for (;;) {
		for (int i = 0; i < world; ++i) cnt_h[i] = 0;
		if (n_cur > 0) {
			if ((r = buf_grow_keep(h, h->shr, (size_t)rows * 16, (size_t)(rows + n_cur) * 16)) < 0) return r;
			ShState *send = nxt; // one interval: the next states are the next round's states
			if (world > 1) {
				if ((r = buf_ensure(h, h->shs, (size_t)world * (size_t)n_cur * 16)) < 0) return r;
				send = (ShState*)h->shs.p;
			}
			// states per octet: enough blocks to fill the chip first, then as many states per cursor atomic as the registers take
			const int S = h->tn.sh_states ? h->tn.sh_states : n_cur >= ((int64_t)1 << 15) ? 8 : 1;
			const int BS = S == 8 && (h->tn.sh_block ? h->tn.sh_block == 1024 : n_cur < ((int64_t)3 << 20)) ? 1024 : 256;
			const unsigned nblk = (unsigned)((n_cur + (BS / 8) * S - 1) / ((BS / 8) * S));
#define RB3_SH_ROUND(SS) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sh_round<SS>), dim3(nblk), dim3(256), 0, h->st, iv, a, n_cur, (const ShState*)cur, d_tw, (ShRec*)h->shr.p + rows, send, n_cur, d_cnt[par], d_cnt[1 - par], d_bad, d_tprev)
			if (BS == 1024) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_sh_round<8, 1024>), dim3(nblk), dim3(1024), 0, h->st, iv, a, n_cur, (const ShState*)cur, d_tw, (ShRec*)h->shr.p + rows, send, n_cur, d_cnt[par], d_cnt[1 - par], d_bad, d_tprev);
			else if (S == 8) RB3_SH_ROUND(8); else if (S == 4) RB3_SH_ROUND(4); else if (S == 2) RB3_SH_ROUND(2); else RB3_SH_ROUND(1);
#undef RB3_SH_ROUND
			HIPCHK(hipMemcpyAsync(hc, d_cnt[par], (size_t)(world + 1) * 8, hipMemcpyDeviceToHost, h->st));
			HIPCHK(hipMemcpyAsync(hc + world + 1, d_bad, 8, hipMemcpyDeviceToHost, h->st));
			HIPCHK(hipStreamSynchronize(h->st));
			if (hc[world + 1] != 0) return RB3GPU_EINTERNAL;
			int64_t sum = 0;
			for (int i = 0; i <= world; ++i) sum += (int64_t)hc[i];
			if (sum != n_cur) return RB3GPU_EINTERNAL;
			for (int i = 0; i < world; ++i) cnt_h[i] = (int64_t)hc[i];
			dirty[par] = true, dirty[1 - par] = false;
			h->stt.n_lf_steps += n_cur, h->stt.n_rank_launches += 1;
		} else if (dirty[1 - par]) { // nothing here this round: the counters the next round adds to still hold an earlier round's sizes
			HIPCHK(hipMemsetAsync(d_cnt[1 - par], 0, (size_t)(RB3_SH_MAXIV + 1) * 8, h->st));
			dirty[1 - par] = false;
		}
		rows += n_cur;
		++rounds;
		int64_t moving = 0, n_in = 0;
		if (world == 1) {
			moving = n_in = cnt_h[0];
		} else {
			if ((r = comm->all_gather(comm->ctx, cnt_h.data(), world, M.data())) < 0) return r; // M[s][d]: states rank s sends to rank d
			for (size_t i = 0; i < M.size(); ++i) moving += M[i];
			for (int q = 0; q < world; ++q) recv_cnt[q] = M[(size_t)q * world + rank], n_in += recv_cnt[q];
		}
		if (moving == 0) break;
		if (n_in > n_chains) return RB3GPU_EINTERNAL;
		if (world == 1) {
			ShState *t = cur; cur = nxt, nxt = t; // (k_sh_round wrote region 0 = nxt)
		} else {
			if ((r = comm->all_to_all(comm->ctx, (const rb3gpu_state_t*)h->shs.p, n_cur, cnt_h.data(), (rb3gpu_state_t*)nxt, recv_cnt.data(), (void*)h->st)) < 0) return r;
			ShState *t = cur; cur = nxt, nxt = t;
		}
		n_cur = n_in, par ^= 1;
	}
	HIPCHK(hipEventRecord(h->ev[1], h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	h->stt.ms_rank += ev_ms(h->ev[0], h->ev[1]);
	if (n_rounds) *n_rounds = rounds;
	// which rows landed here: rows are contiguous per interval, in interval order
	if ((r = comm->all_gather(comm->ctx, &rows, 1, rows_all.data())) < 0) return r;
	int64_t jlo = 0, rsum = 0;
	for (int q = 0; q < world; ++q) { if (q < rank) jlo += rows_all[q]; rsum += rows_all[q]; }
	if (rsum != len) {
		if (h->opt.verbose >= 1) fprintf(stderr, "[E::rb3gpu] sharded merge recorded %lld of %lld rows\n", (long long)rsum, (long long)len);
		return RB3GPU_EINTERNAL;
	}
	if (d_tprev != nullptr) { // the rows of the records, from the owners of their text positions (every rank takes part, with or without records of its own)
		ShTextBounds tb;
		memset(&tb, 0, sizeof(tb));
		tb.n = world;
		for (int q = 0; q <= world; ++q) tb.b[q] = len / world * q + (len % world) * q / world;
		tb.b[world] = len;
		if (tb.b[rank] != t_lo || tb.b[rank + 1] != t_hi) return RB3GPU_EINVAL;
		if ((r = buf_ensure(h, h->misc, MISC_WORDS * 8)) < 0) return r;
		unsigned long long *misc = (unsigned long long*)h->misc.p;
		HIPCHK(hipMemsetAsync(misc, 0, 128, h->st));
		unsigned long long *d_oc = (unsigned long long*)h->shk.p, *d_cur = d_oc + RB3_SH_CNT_STRIDE; // (the counters of the rounds: done with)
		HIPCHK(hipMemsetAsync(h->shk.p, 0, (size_t)RB3_SH_CNT_WORDS * 8, h->st));
		std::vector<int64_t> scnt(world, 0), rcnt(world, 0), M2((size_t)world * world);
		if (rows > 0) {
			hipLaunchKernelGGL(k_sh_owner_count, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, h->st, rows, (const ShRec*)h->shr.p, tb, d_oc, misc + 2);
			unsigned long long hc2[RB3_SH_MAXIV + 1];
			HIPCHK(hipMemcpyAsync(hc2, d_oc, (size_t)world * 8, hipMemcpyDeviceToHost, h->st));
			HIPCHK(hipStreamSynchronize(h->st));
			int64_t sum = 0;
			for (int q = 0; q < world; ++q) scnt[q] = (int64_t)hc2[q], sum += scnt[q];
			if (sum != rows) return RB3GPU_EINTERNAL;
		}
		if ((r = comm->all_gather(comm->ctx, scnt.data(), world, M2.data())) < 0) return r; // M2[s][d]: records rank s asks rank d about
		int64_t stride1 = 1, n_in = 0, stride2 = 1;
		for (int q = 0; q < world; ++q) { stride1 = scnt[q] > stride1 ? scnt[q] : stride1; rcnt[q] = M2[(size_t)q * world + rank]; n_in += rcnt[q]; stride2 = rcnt[q] > stride2 ? rcnt[q] : stride2; }
		// requests out (regions by owner), requests in (packed by source), answers out (regions by source), answers in (packed by owner: over the records)
		if ((r = buf_ensure(h, h->shs, (size_t)world * (size_t)stride1 * 16)) < 0) return r;
		if ((r = buf_ensure(h, h->shc, (size_t)(n_in > 0 ? n_in : 1) * 16)) < 0) return r;
		if ((r = buf_ensure(h, h->shn, (size_t)world * (size_t)stride2 * 16)) < 0) return r;
		if (rows > 0) hipLaunchKernelGGL(k_sh_owner_scatter, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, h->st, rows, (const ShRec*)h->shr.p, tb, stride1, d_cur, (ShRec*)h->shs.p);
		if (world > 1) {
			if ((r = comm->all_to_all(comm->ctx, (const rb3gpu_state_t*)h->shs.p, stride1, scnt.data(), (rb3gpu_state_t*)h->shc.p, rcnt.data(), (void*)h->st)) < 0) return r;
		} else if (rows > 0) HIPCHK(hipMemcpyAsync(h->shc.p, h->shs.p, (size_t)rows * 16, hipMemcpyDeviceToDevice, h->st));
		if (n_in > 0) {
			ShTextBounds off;
			memset(&off, 0, sizeof(off));
			off.n = world;
			for (int q = 0; q < world; ++q) off.b[q + 1] = off.b[q] + rcnt[q];
			hipLaunchKernelGGL(k_sh_lookup, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, h->st, n_in, (const ShRec*)h->shc.p, off, stride2, d_tw, t_lo, t_hi, (ShRec*)h->shn.p, misc + 2);
		}
		if ((r = buf_ensure(h, h->shr, (size_t)(rows > 0 ? rows : 1) * 16)) < 0) return r;
		if (world > 1) {
			if ((r = comm->all_to_all(comm->ctx, (const rb3gpu_state_t*)h->shn.p, stride2, rcnt.data(), (rb3gpu_state_t*)h->shr.p, scnt.data(), (void*)h->st)) < 0) return r;
		} else if (rows > 0) HIPCHK(hipMemcpyAsync(h->shr.p, h->shn.p, (size_t)rows * 16, hipMemcpyDeviceToDevice, h->st));
		if (rows > 0) {
			if ((r = buf_ensure(h, h->pos, (size_t)rows * 8)) < 0) return r;
			if ((r = buf_ensure(h, h->b2, (size_t)rows + 64)) < 0) return r;
			int64_t *dpos = (int64_t*)h->pos.p;
			HIPCHK(hipMemsetAsync(dpos, 0xff, (size_t)rows * 8, h->st)); // RB3_UNSET
			HIPCHK(hipEventRecord(h->ev[2], h->st));
			hipLaunchKernelGGL(k_sh_place_text, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, h->st, rows, (const ShRec*)h->shr.p, jlo, iv_bounds[rank], dpos, (uint8_t*)h->b2.p, misc + 2);
			if ((r = sh_rebuild(h, rows, (const uint8_t*)h->b2.p, dpos, misc, commit)) < 0) return r;
		}
	} else if (rows > 0) {
		if ((r = buf_ensure(h, h->pos, (size_t)rows * 8)) < 0) return r;
		if ((r = buf_ensure(h, h->misc, MISC_WORDS * 8)) < 0) return r;
		unsigned long long *misc = (unsigned long long*)h->misc.p;
		int64_t *dpos = (int64_t*)h->pos.p;
		HIPCHK(hipMemsetAsync(misc, 0, 128, h->st));
		HIPCHK(hipMemsetAsync(dpos, 0xff, (size_t)rows * 8, h->st)); // RB3_UNSET
		HIPCHK(hipEventRecord(h->ev[2], h->st));
		hipLaunchKernelGGL(k_sh_place, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, h->st, rows, (const ShRec*)h->shr.p, jlo, iv_bounds[rank], dpos, misc + 2);
		if ((r = sh_rebuild(h, rows, d_bwt + jlo, dpos, misc, commit)) < 0) return r;
	}
	if (commit) {
		int64_t grow = 0;
		for (int q = 0; q < world; ++q) grow += rows_all[q], iv_bounds[q + 1] += grow;
	}
	return 0;
}

int rb3gpu_sh_merge(rb3gpu_t *h, const rb3gpu_comm_t *comm, int64_t *iv_bounds, int64_t len, const uint8_t *d_bwt, const uint64_t *d_tw,
		int64_t n_chains, const int64_t *chain_tp, int commit, int64_t *n_rounds)
{
	if (!h || !comm || !iv_bounds || len <= 0 || !d_bwt || !d_tw || n_chains <= 0 || n_chains > len || !chain_tp) return RB3GPU_EINVAL;
	if (comm->world < 1 || comm->world > RB3_SH_MAXIV || comm->rank < 0 || comm->rank >= comm->world || !comm->all_gather || (comm->world > 1 && !comm->all_to_all)) return RB3GPU_EINVAL;
	const int r = sh_merge_impl(h, comm, iv_bounds, len, d_bwt, d_tw, n_chains, chain_tp, commit, n_rounds);
	if (r < 0 && comm->abort) comm->abort(comm->ctx); // the other ranks are (or will be) waiting in a collective
	return r;
}

/* the same with the batch SHARDED over the ranks (see sh_merge_impl): d_tprev = the symbol before every text position, one byte each, all of
 * it on this device; d_tw_slice = the text-order words of this rank's text range [len * rank / world, len * (rank + 1) / world) only */
int rb3gpu_sh_merge_text(rb3gpu_t *h, const rb3gpu_comm_t *comm, int64_t *iv_bounds, int64_t len, const uint8_t *d_tprev, const uint64_t *d_tw_slice,
		int64_t n_chains, const int64_t *chain_tp, int commit, int64_t *n_rounds)
{
	if (!h || !comm || !iv_bounds || len <= 0 || !d_tprev || !d_tw_slice || n_chains <= 0 || n_chains > len || !chain_tp) return RB3GPU_EINVAL;
	if (comm->world < 1 || comm->world > RB3_SH_MAXIV || comm->rank < 0 || comm->rank >= comm->world || !comm->all_gather || (comm->world > 1 && !comm->all_to_all)) return RB3GPU_EINVAL;
	const int64_t w = comm->world, q = comm->rank;
	const int64_t t_lo = len / w * q + (len % w) * q / w, t_hi = q + 1 == w ? len : len / w * (q + 1) + (len % w) * (q + 1) / w;
	const int r = sh_merge_impl(h, comm, iv_bounds, len, nullptr, d_tw_slice, n_chains, chain_tp, commit, n_rounds, d_tprev, t_lo, t_hi);
	if (r < 0 && comm->abort) comm->abort(comm->ctx);
	return r;
}

/* the symbol before every text position, one byte each, from the text-order words (what rb3gpu_sh_merge_text walks on) */
int rb3gpu_tprev_from_tw(rb3gpu_t *h, int64_t len, const uint64_t *d_tw, uint8_t *d_out)
{
	if (!h || len <= 0 || !d_tw || !d_out) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	hipLaunchKernelGGL(k_tprev_from_tw, dim3((unsigned)((len / 8 + 1 + 255) / 256)), dim3(256), 0, h->st, d_tw, len, d_out);
	HIPCHK(hipStreamSynchronize(h->st));
	return 0;
}

int rb3gpu_device_of(const rb3gpu_t *h) { return h ? h->dev : -1; }
void *rb3gpu_stream_of(const rb3gpu_t *h) { return h ? (void*)h->st : nullptr; }

int rb3gpu_dev_copy(rb3gpu_t *h, void *d_dst, const void *d_src, int64_t n_bytes)
{
	if (!h || n_bytes < 0 || (n_bytes > 0 && (!d_dst || !d_src))) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (n_bytes > 0) HIPCHK(hipMemcpyAsync(d_dst, d_src, (size_t)n_bytes, hipMemcpyDeviceToDevice, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	return 0;
}

int rb3gpu_dev_memset(rb3gpu_t *h, void *d_dst, int byte, int64_t n_bytes)
{
	if (!h || n_bytes < 0 || (n_bytes > 0 && !d_dst)) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	if (n_bytes > 0) HIPCHK(hipMemsetAsync(d_dst, byte, (size_t)n_bytes, h->st));
	HIPCHK(hipStreamSynchronize(h->st));
	return 0;
}

int rb3gpu_stats(const rb3gpu_t *h, rb3gpu_stats_t *st)
{
	if (!h || !st) return RB3GPU_EINVAL;
	*st = h->stt;
	return 0;
}

int rb3gpu_buffer_bytes(const rb3gpu_t *h, int i, const char **name, int64_t *bytes)
{
	static const char *names[] = { "b2 (batch symbols)", "pos (row records)", "post (records in text order)", "tcnt", "tpre", "ctot", "ctot2", "gstat", "gpre", "jg (rows per window)", "misc", "xbuf (scratch)", "wl (walker list)", "wls (sentinel positions of a device-made list)",
		"dl (stretch table)", "dlx (wide masks)", "wstat", "wplane", "wruns", "gslots (run-space rebuild)", "glist", "pslots (plane-space rebuild)", "lbst", "shc", "shn", "shs", "shr", "shk" };
	if (!h || !name || !bytes || i < 0) return RB3GPU_EINVAL;
	rb3gpu_t *m = const_cast<rb3gpu_t*>(h);
	Buf *all[] = { &m->b2, &m->pos, &m->post, &m->tcnt, &m->tpre, &m->ctot, &m->ctot2, &m->gstat, &m->gpre, &m->jg, &m->misc, &m->xbuf, &m->wl, &m->wls, &m->dl, &m->dlx, &m->wstat, &m->wplane, &m->wruns, &m->gslots, &m->glist, &m->pslots, &m->lbst,
		&m->shc, &m->shn, &m->shs, &m->shr, &m->shk };
	const int nb = (int)(sizeof(all) / sizeof(all[0]));
	static_assert(sizeof(all) / sizeof(all[0]) == sizeof(names) / sizeof(names[0]), "a name per buffer");
	if (i < nb) { *name = names[i], *bytes = all[i]->p ? (int64_t)all[i]->cap : 0; return 0; }
	if (i < nb + 2) { *name = i == nb + h->cur ? "index directory (current)" : "index directory (being built)", *bytes = h->ib[i - nb].grp ? (int64_t)(h->ib[i - nb].grp_cap * RB3_GRP_ALLOC) : 0; return 0; }
	if (i < nb + 4) { *name = i == nb + 2 + h->cur ? "index slots (current)" : "index slots (being built)", *bytes = h->ib[i - nb - 2].slots ? (int64_t)(h->ib[i - nb - 2].slots_cap * sizeof(rb3_slot_t)) : 0; return 0; }
	return RB3GPU_EINVAL;
}

void rb3gpu_stats_reset(rb3gpu_t *h)
{
	if (!h) return;
	const int64_t bi = h->stt.bytes_index;
	memset(&h->stt, 0, sizeof(h->stt));
	h->stt.bytes_index = bi, h->stt.bytes_peak = h->bytes_owned;
}

int rb3gpu_dev_alloc(rb3gpu_t *h, int64_t n_bytes, void **d_ptr)
{
	if (!h || n_bytes < 0 || !d_ptr) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	hipError_t e = hipMalloc(d_ptr, (size_t)(n_bytes > 0 ? n_bytes : 1));
	if (e == hipErrorOutOfMemory && !h->garbage.empty()) { // replaced buffers of the handle are still held (defer_free): give them back first
		(void)hipGetLastError();
		garbage_collect(h, true);
		e = hipMalloc(d_ptr, (size_t)(n_bytes > 0 ? n_bytes : 1));
	}
	HIPCHK(e);
	return 0;
}

int rb3gpu_dev_upload(rb3gpu_t *h, void *d_dst, const void *src, int64_t n_bytes)
{
	if (!h || !d_dst || !src || n_bytes < 0) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	HIPCHK(hipMemcpy(d_dst, src, (size_t)n_bytes, hipMemcpyHostToDevice));
	return 0;
}

int rb3gpu_dev_download(rb3gpu_t *h, void *dst, const void *d_src, int64_t n_bytes)
{
	if (!h || !dst || !d_src || n_bytes < 0) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	HIPCHK(hipMemcpy(dst, d_src, (size_t)n_bytes, hipMemcpyDeviceToHost));
	return 0;
}

int rb3gpu_stream_sync(void *stream)
{
	if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { (void)hipGetLastError(); return RB3GPU_ENODEV; }
	return 0;
}

int rb3gpu_dev_free(rb3gpu_t *h, void *d_ptr)
{
	if (!h) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	HIPCHK(hipFree(d_ptr));
	return 0;
}

int rb3gpu_sync(rb3gpu_t *h)
{
	if (!h) return RB3GPU_EINVAL;
	HIPCHK(hipSetDevice(h->dev));
	HIPCHK(hipStreamSynchronize(h->st));
	return 0;
}

} // extern "C"
