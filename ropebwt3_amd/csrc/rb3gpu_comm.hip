/*
 * rb3gpu_comm.hip -- the two communicators that ship with librb3gpu.so for the interval-sharded merge (rb3gpu_sh_merge,
 * include/rb3gpu.h): what carries the split sizes and the 16-byte chain states between the GPUs of one node.  No kernels here.
 *
 *   rb3gpu_group_*   ranks are THREADS of one process, one device each (the CLI's `build --gpus N --interval`): the collectives are
 *                    barriers between the threads plus hipMemcpyPeerAsync device to device over xGMI -- every rank PULLS its
 *                    share out of the other ranks' send regions, all links busy at once, nothing staged through the host.
 *   rb3gpu_rccl_*    one PROCESS per GPU (bench.py --gpus N, torchrun): grouped ncclSend / ncclRecv on the engine's stream --
 *                    RCCL's all-to-all over the point-to-point xGMI links -- and ncclAllGather for the split sizes.  librccl is
 *                    loaded at run time (dlopen), so the library itself has no link-time dependency on it.
 *
 * The reference has no counterpart: its ropes are shared by the threads of kt_for (fm-index.c:217-224, kthread.c:40-52).
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include "rb3gpu.h"

/* ---- ranks as threads of one process ---- */

struct GroupMember { rb3gpu_group_s *g; int rank, dev; };

struct rb3gpu_group_s {
	int world = 0;
	pthread_mutex_t mtx;
	pthread_cond_t cv;
	int waiting = 0, aborted = 0;
	unsigned long gen = 0;
	int64_t slots[RB3GPU_SH_MAXIV * RB3GPU_SH_MAXIV];           // all_gather: n <= RB3GPU_SH_MAXIV values per rank
	struct { const rb3gpu_state_t *send; int64_t stride; int64_t cnt[RB3GPU_SH_MAXIV]; int dev; } pub[RB3GPU_SH_MAXIV];
	GroupMember mem[RB3GPU_SH_MAXIV];
};

static int group_barrier(rb3gpu_group_s *g)
{
	int r = 0;
	pthread_mutex_lock(&g->mtx);
	if (!g->aborted) {
		const unsigned long my = g->gen;
		if (++g->waiting == g->world) {
			g->waiting = 0, ++g->gen;
			pthread_cond_broadcast(&g->cv);
		} else {
			while (g->gen == my && !g->aborted) pthread_cond_wait(&g->cv, &g->mtx);
		}
	}
	if (g->aborted) r = RB3GPU_ESTATE;
	pthread_mutex_unlock(&g->mtx);
	return r;
}

static int group_all_gather(void *ctx, const int64_t *send, int n, int64_t *recv)
{
	GroupMember *m = (GroupMember*)ctx;
	rb3gpu_group_s *g = m->g;
	if (n < 0 || n > RB3GPU_SH_MAXIV) return RB3GPU_EINVAL;
	memcpy(g->slots + (size_t)m->rank * n, send, (size_t)n * 8);
	int r;
	if ((r = group_barrier(g)) < 0) return r;
	memcpy(recv, g->slots, (size_t)g->world * n * 8);
	return group_barrier(g); // (nobody overwrites the slots before everybody has read them)
}

static int group_all_to_all(void *ctx, const rb3gpu_state_t *d_send, int64_t stride, const int64_t *send_cnt, rb3gpu_state_t *d_recv, const int64_t *recv_cnt, void *stream)
{
	GroupMember *m = (GroupMember*)ctx;
	rb3gpu_group_s *g = m->g;
	hipStream_t st = (hipStream_t)stream;
	int r;
	if (hipSetDevice(m->dev) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rb3gpu_group_abort(g); return RB3GPU_ENODEV; } // the send regions are complete
	g->pub[m->rank].send = d_send, g->pub[m->rank].stride = stride, g->pub[m->rank].dev = m->dev;
	memcpy(g->pub[m->rank].cnt, send_cnt, (size_t)g->world * 8);
	if ((r = group_barrier(g)) < 0) return r;
	int64_t at = 0;
	for (int src = 0; src < g->world; ++src) { // (receive order is rank order: the states land packed by source)
		const int64_t n = g->pub[src].cnt[m->rank];
		if (n != recv_cnt[src]) { rb3gpu_group_abort(g); return RB3GPU_EINTERNAL; }
		if (n > 0) {
			const rb3gpu_state_t *from = g->pub[src].send + (int64_t)m->rank * g->pub[src].stride;
			const hipError_t e = g->pub[src].dev == m->dev ? hipMemcpyAsync(d_recv + at, from, (size_t)n * 16, hipMemcpyDeviceToDevice, st)
			                                               : hipMemcpyPeerAsync(d_recv + at, m->dev, from, g->pub[src].dev, (size_t)n * 16, st);
			if (e != hipSuccess) { rb3gpu_group_abort(g); return RB3GPU_ENODEV; }
		}
		at += n;
	}
	if (hipStreamSynchronize(st) != hipSuccess) { rb3gpu_group_abort(g); return RB3GPU_ENODEV; }
	return group_barrier(g); // (a send region is not rewritten before every rank has pulled its share)
}

static void group_abort_cb(void *ctx) { rb3gpu_group_abort(((GroupMember*)ctx)->g); }

extern "C" {

rb3gpu_group_t *rb3gpu_group_create(int world)
{
	if (world < 1 || world > RB3GPU_SH_MAXIV) return nullptr;
	rb3gpu_group_s *g = new (std::nothrow) rb3gpu_group_s;
	if (!g) return nullptr;
	g->world = world;
	pthread_mutex_init(&g->mtx, nullptr);
	pthread_cond_init(&g->cv, nullptr);
	memset(g->pub, 0, sizeof(g->pub));
	return g;
}

int rb3gpu_group_comm(rb3gpu_group_t *g, int rank, rb3gpu_t *h, rb3gpu_comm_t *comm)
{
	if (!g || !h || !comm || rank < 0 || rank >= g->world) return RB3GPU_EINVAL;
	GroupMember *m = &g->mem[rank];
	m->g = g, m->rank = rank, m->dev = rb3gpu_device_of(h);
	// direct peer access where the devices differ (an error here only means "already enabled" or "copies get staged": both fine)
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) == hipSuccess && hipSetDevice(m->dev) == hipSuccess)
		for (int d = 0; d < ndev; ++d)
			if (d != m->dev) { int can = 0; if (hipDeviceCanAccessPeer(&can, m->dev, d) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(d, 0); (void)hipGetLastError(); }
	comm->ctx = m, comm->rank = rank, comm->world = g->world;
	comm->all_gather = group_all_gather, comm->all_to_all = group_all_to_all, comm->abort = group_abort_cb;
	return 0;
}

void rb3gpu_group_abort(rb3gpu_group_t *g)
{
	if (!g) return;
	pthread_mutex_lock(&g->mtx);
	g->aborted = 1;
	pthread_cond_broadcast(&g->cv);
	pthread_mutex_unlock(&g->mtx);
}

void rb3gpu_group_destroy(rb3gpu_group_t *g)
{
	if (!g) return;
	pthread_cond_destroy(&g->cv);
	pthread_mutex_destroy(&g->mtx);
	delete g;
}

} // extern "C"

/* ---- one process per GPU over RCCL ---- */

struct RcclApi {
	void *lib = nullptr;
	decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
	decltype(&ncclCommInitRank) CommInitRank = nullptr;
	decltype(&ncclCommDestroy) CommDestroy = nullptr;
	decltype(&ncclCommAbort) CommAbort = nullptr;
	decltype(&ncclAllGather) AllGather = nullptr;
	decltype(&ncclSend) Send = nullptr;
	decltype(&ncclRecv) Recv = nullptr;
	decltype(&ncclGroupStart) GroupStart = nullptr;
	decltype(&ncclGroupEnd) GroupEnd = nullptr;
	decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
static RcclApi g_rccl;
static pthread_mutex_t g_rccl_mtx = PTHREAD_MUTEX_INITIALIZER;

static int rccl_load(void)
{
	pthread_mutex_lock(&g_rccl_mtx);
	if (!g_rccl.lib) {
		// The RCCL that sits NEXT TO the HIP runtime this process runs on comes first: a process can hold two ROCm trees (PyTorch wheels
		// bundle their own libamdhip64 / libhsa-runtime64 / librccl under torch/lib, all with the system's sonames), the HIP runtime is
		// whichever was loaded first, and an RCCL of the other tree brings up a second, uninitialised HSA runtime ("no ROCm-capable device").
		char beside[2][4096] = { "", "" };
		Dl_info di;
		if (dladdr((const void*)&hipGetDeviceCount, &di) && di.dli_fname) {
			const char *slash = strrchr(di.dli_fname, '/');
			if (slash && (size_t)(slash - di.dli_fname) < sizeof(beside[0]) - 32) {
				const int dl = (int)(slash - di.dli_fname);
				snprintf(beside[0], sizeof(beside[0]), "%.*s/librccl.so.1", dl, di.dli_fname);
				snprintf(beside[1], sizeof(beside[1]), "%.*s/librccl.so", dl, di.dli_fname);
			}
		}
		const char *names[] = { getenv("RB3GPU_RCCL_LIB"), beside[0], beside[1], "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
		void *lib = nullptr;
		for (const char *n : names)
			if (n && *n && (lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
		if (lib) {
			RcclApi a;
			a.lib = lib;
#define RB3_SYM(f) a.f = (decltype(a.f))dlsym(lib, "nccl" #f)
			RB3_SYM(GetUniqueId); RB3_SYM(CommInitRank); RB3_SYM(CommDestroy); RB3_SYM(CommAbort); RB3_SYM(AllGather);
			RB3_SYM(Send); RB3_SYM(Recv); RB3_SYM(GroupStart); RB3_SYM(GroupEnd); RB3_SYM(GetErrorString);
#undef RB3_SYM
			if (a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.CommAbort && a.AllGather && a.Send && a.Recv && a.GroupStart && a.GroupEnd) g_rccl = a;
			else dlclose(lib);
		}
	}
	const int ok = g_rccl.lib != nullptr;
	pthread_mutex_unlock(&g_rccl_mtx);
	return ok ? 0 : RB3GPU_EUNSUP;
}

struct RcclCtx {
	ncclComm_t comm = nullptr;
	int rank = 0, world = 1, dev = 0;
	hipStream_t st = nullptr;       // the engine's stream: every collective is ordered behind its kernels
	int64_t *d_s = nullptr, *d_r = nullptr, *h_r = nullptr; // all_gather staging: RB3GPU_SH_MAXIV values out, world times that in (page-locked landing place)
};

#define RB3_NCCL(x) do { const ncclResult_t e_ = (x); if (e_ != ncclSuccess) { \
		fprintf(stderr, "[E::rb3gpu] RCCL: %s (%s:%d)\n", g_rccl.GetErrorString ? g_rccl.GetErrorString(e_) : "error", __FILE__, __LINE__); return RB3GPU_ENODEV; } } while (0)

static int rccl_all_gather(void *ctx, const int64_t *send, int n, int64_t *recv)
{
	RcclCtx *c = (RcclCtx*)ctx;
	if (n < 0 || n > RB3GPU_SH_MAXIV) return RB3GPU_EINVAL;
	if (hipSetDevice(c->dev) != hipSuccess) return RB3GPU_ENODEV;
	(void)hipGetLastError();
	if (hipMemcpyAsync(c->d_s, send, (size_t)n * 8, hipMemcpyHostToDevice, c->st) != hipSuccess) return RB3GPU_ENODEV;
	RB3_NCCL(g_rccl.AllGather(c->d_s, c->d_r, (size_t)n, ncclInt64, c->comm, c->st));
	if (hipMemcpyAsync(c->h_r, c->d_r, (size_t)c->world * n * 8, hipMemcpyDeviceToHost, c->st) != hipSuccess) return RB3GPU_ENODEV;
	if (hipStreamSynchronize(c->st) != hipSuccess) return RB3GPU_ENODEV;
	memcpy(recv, c->h_r, (size_t)c->world * n * 8);
	return 0;
}

static int rccl_all_to_all(void *ctx, const rb3gpu_state_t *d_send, int64_t stride, const int64_t *send_cnt, rb3gpu_state_t *d_recv, const int64_t *recv_cnt, void *stream)
{
	RcclCtx *c = (RcclCtx*)ctx;
	hipStream_t st = (hipStream_t)stream;
	if (hipSetDevice(c->dev) != hipSuccess) return RB3GPU_ENODEV;
	(void)hipGetLastError();
	// every pair inside one group: RCCL schedules the sends and receives of a rank concurrently over its xGMI links (a state = 2 x int64)
	RB3_NCCL(g_rccl.GroupStart());
	int64_t at = 0;
	for (int p = 0; p < c->world; ++p) {
		if (send_cnt[p] > 0) RB3_NCCL(g_rccl.Send(d_send + (int64_t)p * stride, (size_t)send_cnt[p] * 2, ncclInt64, p, c->comm, st));
		if (recv_cnt[p] > 0) RB3_NCCL(g_rccl.Recv(d_recv + at, (size_t)recv_cnt[p] * 2, ncclInt64, p, c->comm, st));
		at += recv_cnt[p];
	}
	RB3_NCCL(g_rccl.GroupEnd());
	return 0; // (stream-ordered: the next round's kernel waits for the receives, and rewrites the send regions only behind the sends)
}

static void rccl_abort_cb(void *ctx)
{
	RcclCtx *c = (RcclCtx*)ctx;
	if (c && c->comm) { (void)g_rccl.CommAbort(c->comm); c->comm = nullptr; }
}

extern "C" {

int rb3gpu_rccl_unique_id(char id[RB3GPU_RCCL_ID_BYTES])
{
	if (!id) return RB3GPU_EINVAL;
	int r;
	if ((r = rccl_load()) < 0) return r;
	ncclUniqueId u;
	static_assert(sizeof(u) == RB3GPU_RCCL_ID_BYTES, "RB3GPU_RCCL_ID_BYTES must be NCCL_UNIQUE_ID_BYTES");
	RB3_NCCL(g_rccl.GetUniqueId(&u));
	memcpy(id, &u, sizeof(u));
	return 0;
}

int rb3gpu_rccl_comm_create(rb3gpu_t *h, int rank, int world, const char id[RB3GPU_RCCL_ID_BYTES], rb3gpu_comm_t *comm)
{
	if (!h || !id || !comm || world < 1 || world > RB3GPU_SH_MAXIV || rank < 0 || rank >= world) return RB3GPU_EINVAL;
	int r;
	if ((r = rccl_load()) < 0) return r;
	RcclCtx *c = new (std::nothrow) RcclCtx;
	if (!c) return RB3GPU_ENOMEM;
	c->rank = rank, c->world = world, c->dev = rb3gpu_device_of(h), c->st = (hipStream_t)rb3gpu_stream_of(h);
	ncclUniqueId u;
	memcpy(&u, id, sizeof(u));
	const char *what = nullptr;
	ncclResult_t ne = ncclSuccess;
	if (hipSetDevice(c->dev) != hipSuccess) what = "hipSetDevice";
	else if (hipMalloc((void**)&c->d_s, RB3GPU_SH_MAXIV * 8) != hipSuccess || hipMalloc((void**)&c->d_r, (size_t)RB3GPU_SH_MAXIV * RB3GPU_SH_MAXIV * 8) != hipSuccess) what = "hipMalloc";
	else if (hipHostMalloc((void**)&c->h_r, (size_t)RB3GPU_SH_MAXIV * RB3GPU_SH_MAXIV * 8, hipHostMallocDefault) != hipSuccess) what = "hipHostMalloc";
	else if ((void)hipGetLastError(), (ne = g_rccl.CommInitRank(&c->comm, world, u, rank)) != ncclSuccess) what = "ncclCommInitRank"; // (RCCL takes a stale
	// "last error" of this thread -- a tolerated hipErrorNotReady of an event query, say -- for one of its own: cleared first)
	if (what) {
		fprintf(stderr, "[E::rb3gpu] RCCL communicator of rank %d/%d on device %d: %s failed%s%s\n", rank, world, c->dev, what, ne != ncclSuccess && g_rccl.GetErrorString ? ": " : "", ne != ncclSuccess && g_rccl.GetErrorString ? g_rccl.GetErrorString(ne) : "");
		(void)hipGetLastError();
		if (c->d_s) (void)hipFree(c->d_s);
		if (c->d_r) (void)hipFree(c->d_r);
		if (c->h_r) (void)hipHostFree(c->h_r);
		delete c;
		return RB3GPU_ENODEV;
	}
	comm->ctx = c, comm->rank = rank, comm->world = world;
	comm->all_gather = rccl_all_gather, comm->all_to_all = rccl_all_to_all, comm->abort = rccl_abort_cb;
	return 0;
}

void rb3gpu_rccl_comm_destroy(rb3gpu_comm_t *comm)
{
	if (!comm || !comm->ctx) return;
	RcclCtx *c = (RcclCtx*)comm->ctx;
	(void)hipSetDevice(c->dev);
	(void)hipStreamSynchronize(c->st);
	if (c->comm) (void)g_rccl.CommDestroy(c->comm);
	(void)hipFree(c->d_s);
	(void)hipFree(c->d_r);
	(void)hipHostFree(c->h_r);
	delete c;
	comm->ctx = nullptr;
}

} // extern "C"

/* ---- the interval-sharded index as one object (a single-process host program: the CLI) ---- */

struct rb3gpu_shard_s {
	int n = 0;
	rb3gpu_t *h[RB3GPU_SH_MAXIV];        // h[0] is the caller's
	int dev[RB3GPU_SH_MAXIV];
	int64_t bounds[RB3GPU_SH_MAXIV + 1];
	rb3gpu_group_t *grp = nullptr;
	void *rep_bwt[RB3GPU_SH_MAXIV], *rep_tw[RB3GPU_SH_MAXIV]; // replicas of the batch on the devices of intervals 1.. (kept between merges)
	int64_t rep_cap[RB3GPU_SH_MAXIV];
};

static int copy_across(void *dst, int ddev, const void *src, int sdev, size_t n)
{
	if (n == 0) return 0;
	// A device-to-device hipMemcpy returns when the copy is QUEUED on the null stream, not when it is done, and the handles' streams do not
	// wait for the null stream: without the synchronisation the interval handles were built from (or the batch read out of) buffers the
	// copy had not reached yet -- only when something else kept the device busy (the CLI's sorter thread), and then silently:
	// a wrong-but-valid BWT (found by the 10 M-read test of the interval build; tools/probe_iv_scale.py reproduces it).
	hipError_t e = ddev == sdev ? hipMemcpy(dst, src, n, hipMemcpyDeviceToDevice) : hipMemcpyPeer(dst, ddev, src, sdev, n);
	int cur = 0;
	if (e == hipSuccess) e = hipGetDevice(&cur);
	if (e == hipSuccess && (e = hipSetDevice(sdev)) == hipSuccess) e = hipDeviceSynchronize();
	if (e == hipSuccess && ddev != sdev && (e = hipSetDevice(ddev)) == hipSuccess) e = hipDeviceSynchronize();
	if (e == hipSuccess) e = hipSetDevice(cur);
	if (e != hipSuccess) { (void)hipGetLastError(); return RB3GPU_ENODEV; }
	return 0;
}

struct ShardJob { rb3gpu_shard_s *s; int rank; int64_t len; const uint8_t *d_bwt; const uint64_t *d_tw; int64_t n_chains; const int64_t *chain_tp; int64_t bounds[RB3GPU_SH_MAXIV + 1]; int64_t rounds; int ret; };

static void *shard_thread(void *arg)
{
	ShardJob *j = (ShardJob*)arg;
	rb3gpu_comm_t comm;
	j->ret = rb3gpu_group_comm(j->s->grp, j->rank, j->s->h[j->rank], &comm);
	if (j->ret < 0) { rb3gpu_group_abort(j->s->grp); return nullptr; }
	j->ret = rb3gpu_sh_merge(j->s->h[j->rank], &comm, j->bounds, j->len, j->d_bwt, j->d_tw, j->n_chains, j->chain_tp, 1, &j->rounds);
	return nullptr;
}

extern "C" {

static void shard_free(rb3gpu_shard_s *s, bool handles)
{
	for (int i = 1; i < s->n; ++i) {
		if (s->h[i]) {
			if (s->rep_bwt[i]) (void)rb3gpu_dev_free(s->h[i], s->rep_bwt[i]);
			if (s->rep_tw[i]) (void)rb3gpu_dev_free(s->h[i], s->rep_tw[i]);
			if (handles) rb3gpu_destroy(s->h[i]);
		}
	}
	rb3gpu_group_destroy(s->grp);
	delete s;
}

rb3gpu_shard_t *rb3gpu_shard_split(rb3gpu_t *h0, int n, const int *devices, const rb3gpu_opt_t *opt)
{
	if (!h0 || !devices || !opt || n < 1 || n > RB3GPU_SH_MAXIV || devices[0] != rb3gpu_device_of(h0)) return nullptr;
	const int64_t tot = rb3gpu_get_tot(h0);
	if (tot < n) return nullptr;
	rb3gpu_shard_s *s = new (std::nothrow) rb3gpu_shard_s;
	if (!s) return nullptr;
	s->n = n;
	for (int i = 0; i < RB3GPU_SH_MAXIV; ++i) s->h[i] = nullptr, s->rep_bwt[i] = s->rep_tw[i] = nullptr, s->rep_cap[i] = 0, s->dev[i] = 0;
	s->h[0] = h0;
	for (int i = 0; i <= n; ++i) s->bounds[i] = tot / n * i + (tot % n) * i / n;
	s->bounds[n] = tot;
	if ((s->grp = rb3gpu_group_create(n)) == nullptr) { delete s; return nullptr; }
	void *plain = nullptr;
	int64_t acc0[RB3GPU_ASIZE + 1], sum[RB3GPU_ASIZE] = {0, 0, 0, 0, 0, 0};
	int r = rb3gpu_get_acc(h0, acc0);
	if (r == 0 && n > 1) r = rb3gpu_dev_alloc(h0, tot, &plain);
	if (r == 0 && n > 1) r = rb3gpu_export_plain_dev(h0, (uint8_t*)plain);
	for (int i = 0; i < n && r == 0; ++i) {
		s->dev[i] = devices[i];
		const int64_t len = s->bounds[i + 1] - s->bounds[i];
		if (i == 0) { if (n > 1) r = rb3gpu_from_plain_dev(h0, len, (const uint8_t*)plain); continue; } // (h0 rebuilds itself from the copy: its first interval)
		rb3gpu_opt_t o = *opt;
		o.device = devices[i];
		if ((s->h[i] = rb3gpu_create(&o)) == nullptr) { r = RB3GPU_ENODEV; break; }
		void *part = nullptr;
		if ((r = rb3gpu_dev_alloc(s->h[i], len, &part)) < 0) break;
		r = copy_across(part, s->dev[i], (const uint8_t*)plain + s->bounds[i], s->dev[0], (size_t)len);
		if (r == 0) r = rb3gpu_from_plain_dev(s->h[i], len, (const uint8_t*)part);
		(void)rb3gpu_dev_free(s->h[i], part);
	}
	if (plain) (void)rb3gpu_dev_free(h0, plain);
	for (int i = 0; i < n && r == 0; ++i) { // the intervals together hold the symbols the index held (a copy that went wrong would show here, not in a wrong BWT later)
		int64_t acc[RB3GPU_ASIZE + 1];
		if ((r = rb3gpu_get_acc(s->h[i], acc)) == 0)
			for (int c = 0; c < RB3GPU_ASIZE; ++c) sum[c] += acc[c + 1] - acc[c];
	}
	for (int c = 0; c < RB3GPU_ASIZE && r == 0; ++c) if (sum[c] != acc0[c + 1] - acc0[c]) r = RB3GPU_EINTERNAL;
	if (r < 0) { shard_free(s, true); return nullptr; } // (h0 may hold its first interval only: the caller gives the build up)
	return s;
}

int rb3gpu_shard_merge(rb3gpu_shard_t *s, int64_t len, const uint8_t *d_bwt, const uint64_t *d_tw, int64_t n_chains, const int64_t *chain_tp, int64_t *n_rounds)
{
	if (!s || len <= 0 || !d_bwt || !d_tw || n_chains <= 0 || !chain_tp) return RB3GPU_EINVAL;
	int r;
	// the batch on every device (1 + 8 bytes per symbol, device to device)
	for (int i = 1; i < s->n; ++i) {
		if (s->dev[i] == s->dev[0]) continue; // (several intervals on one device: they read the same copy)
		if (s->rep_cap[i] < len) {
			if (s->rep_bwt[i]) (void)rb3gpu_dev_free(s->h[i], s->rep_bwt[i]);
			if (s->rep_tw[i]) (void)rb3gpu_dev_free(s->h[i], s->rep_tw[i]);
			s->rep_bwt[i] = s->rep_tw[i] = nullptr, s->rep_cap[i] = 0;
			const int64_t cap = len + (len >> 2);
			if ((r = rb3gpu_dev_alloc(s->h[i], cap, &s->rep_bwt[i])) < 0 || (r = rb3gpu_dev_alloc(s->h[i], cap * 8, &s->rep_tw[i])) < 0) return r;
			s->rep_cap[i] = cap;
		}
		if ((r = copy_across(s->rep_bwt[i], s->dev[i], d_bwt, s->dev[0], (size_t)len)) < 0) return r;
		if ((r = copy_across(s->rep_tw[i], s->dev[i], d_tw, s->dev[0], (size_t)len * 8)) < 0) return r;
	}
	ShardJob *jobs = new (std::nothrow) ShardJob[s->n];
	pthread_t *th = new (std::nothrow) pthread_t[s->n];
	if (!jobs || !th) { delete[] jobs; delete[] th; return RB3GPU_ENOMEM; }
	for (int i = 0; i < s->n; ++i) {
		const bool here = i == 0 || s->dev[i] == s->dev[0];
		jobs[i].s = s, jobs[i].rank = i, jobs[i].len = len, jobs[i].n_chains = n_chains, jobs[i].chain_tp = chain_tp, jobs[i].rounds = 0, jobs[i].ret = 0;
		jobs[i].d_bwt = here ? d_bwt : (const uint8_t*)s->rep_bwt[i], jobs[i].d_tw = here ? d_tw : (const uint64_t*)s->rep_tw[i];
		memcpy(jobs[i].bounds, s->bounds, sizeof(s->bounds));
	}
	int started = 0;
	for (int i = 1; i < s->n; ++i, ++started)
		if (pthread_create(&th[i], nullptr, shard_thread, &jobs[i]) != 0) { rb3gpu_group_abort(s->grp); jobs[i].ret = RB3GPU_ENOMEM; break; }
	if (started == s->n - 1) shard_thread(&jobs[0]); // interval 0 on the calling thread
	else jobs[0].ret = RB3GPU_ENOMEM;
	for (int i = 1; i <= started; ++i) pthread_join(th[i], nullptr);
	r = 0;
	for (int i = 0; i < s->n; ++i) if (jobs[i].ret < 0 && (r == 0 || r == RB3GPU_ESTATE)) r = jobs[i].ret; // (ESTATE: a rank that was only woken up by another one's failure)
	if (r == 0) {
		memcpy(s->bounds, jobs[0].bounds, sizeof(s->bounds));
		if (n_rounds) *n_rounds = jobs[0].rounds;
	}
	delete[] jobs;
	delete[] th;
	return r;
}

int rb3gpu_shard_gather(rb3gpu_shard_t *s)
{
	if (!s) return RB3GPU_EINVAL;
	int r = 0;
	if (s->n > 1) {
		const int64_t tot = s->bounds[s->n];
		void *plain = nullptr;
		if ((r = rb3gpu_dev_alloc(s->h[0], tot, &plain)) == 0) {
			for (int i = 0; i < s->n && r == 0; ++i) {
				const int64_t len = s->bounds[i + 1] - s->bounds[i];
				if (rb3gpu_get_tot(s->h[i]) != len) { r = RB3GPU_EINTERNAL; break; }
				if (s->dev[i] == s->dev[0]) { r = rb3gpu_export_plain_dev(s->h[i], (uint8_t*)plain + s->bounds[i]); continue; }
				void *part = nullptr;
				if ((r = rb3gpu_dev_alloc(s->h[i], len, &part)) < 0) break;
				if ((r = rb3gpu_export_plain_dev(s->h[i], (uint8_t*)part)) == 0) r = copy_across((uint8_t*)plain + s->bounds[i], s->dev[0], part, s->dev[i], (size_t)len);
				(void)rb3gpu_dev_free(s->h[i], part);
			}
			int64_t sum[RB3GPU_ASIZE] = {0, 0, 0, 0, 0, 0}, acc[RB3GPU_ASIZE + 1];
			for (int i = 0; i < s->n && r == 0; ++i)
				if ((r = rb3gpu_get_acc(s->h[i], acc)) == 0)
					for (int c = 0; c < RB3GPU_ASIZE; ++c) sum[c] += acc[c + 1] - acc[c];
			if (r == 0) r = rb3gpu_from_plain_dev(s->h[0], tot, (const uint8_t*)plain);
			if (r == 0 && (r = rb3gpu_get_acc(s->h[0], acc)) == 0)
				for (int c = 0; c < RB3GPU_ASIZE; ++c) if (acc[c + 1] - acc[c] != sum[c]) r = RB3GPU_EINTERNAL; // (the symbols of the intervals, no more and no fewer)
			(void)rb3gpu_dev_free(s->h[0], plain);
		}
	}
	shard_free(s, true);
	return r;
}

rb3gpu_t *rb3gpu_shard_handle(rb3gpu_shard_t *s, int i) { return s && i >= 0 && i < s->n ? s->h[i] : nullptr; }

int rb3gpu_shard_bounds(const rb3gpu_shard_t *s, int64_t *bounds)
{
	if (!s || !bounds) return RB3GPU_EINVAL;
	memcpy(bounds, s->bounds, (size_t)(s->n + 1) * 8);
	return s->n;
}

} // extern "C"
