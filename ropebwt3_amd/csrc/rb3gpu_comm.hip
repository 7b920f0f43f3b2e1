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
