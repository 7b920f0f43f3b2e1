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
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <sched.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <atomic>
#include <vector>
#include "rb3gpu.h"

/* ---- ranks as threads of one process ---- */

struct GroupMember { rb3gpu_group_s *g; int rank, dev; hipEvent_t ev[2]; int par; };

struct rb3gpu_group_s {
	int world = 0;
	pthread_mutex_t mtx;
	pthread_cond_t cv;
	std::atomic<int> waiting{0}, aborted{0};
	std::atomic<unsigned long> gen{0};
	int64_t slots[RB3GPU_SH_MAXIV * RB3GPU_SH_MAXIV];           // all_gather: n <= RB3GPU_SH_MAXIV values per rank
	struct { const rb3gpu_state_t *send; int64_t stride; int64_t cnt[RB3GPU_SH_MAXIV]; int dev; } pub[RB3GPU_SH_MAXIV];
	GroupMember mem[RB3GPU_SH_MAXIV];
};

/* A rank that arrives spins for a few tens of microseconds before it goes to sleep on the condition variable: the peer rounds of the sharded merge pass
 * one barrier per lock-step round (group_stream_barrier), the ranks arrive within microseconds of each other, and a wake-up through the kernel costs
 * more than the round's launch.  The generation changes under the mutex, so a rank that gives up spinning cannot miss it. */
static int group_barrier(rb3gpu_group_s *g)
{
	if (g->aborted.load(std::memory_order_acquire)) return RB3GPU_ESTATE;
	const unsigned long my = g->gen.load(std::memory_order_acquire);
	if (g->waiting.fetch_add(1, std::memory_order_acq_rel) + 1 == g->world) {
		g->waiting.store(0, std::memory_order_relaxed); // (nobody adds to it again before the generation has changed)
		pthread_mutex_lock(&g->mtx);
		g->gen.store(my + 1, std::memory_order_release);
		pthread_cond_broadcast(&g->cv);
		pthread_mutex_unlock(&g->mtx);
	} else {
		for (int spins = 0; spins < 4000 && g->gen.load(std::memory_order_acquire) == my && !g->aborted.load(std::memory_order_relaxed); ++spins) __builtin_ia32_pause();
		if (g->gen.load(std::memory_order_acquire) == my && !g->aborted.load(std::memory_order_acquire)) {
			pthread_mutex_lock(&g->mtx);
			while (g->gen.load(std::memory_order_acquire) == my && !g->aborted.load(std::memory_order_acquire)) pthread_cond_wait(&g->cv, &g->mtx);
			pthread_mutex_unlock(&g->mtx);
		}
	}
	return g->aborted.load(std::memory_order_acquire) ? RB3GPU_ESTATE : 0;
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

/* rb3gpu_comm_t.stream_barrier: every rank records an event behind what it has queued, the threads meet (the host waits for the other THREADS, not
 * for any device), and every rank's stream then waits for the other ranks' events.  Two events per rank, used in turn: a rank records event p again
 * two barriers later, and it cannot get there before every other rank has queued its wait for the earlier record (they all pass the barrier in between). */
static int group_stream_barrier(void *ctx, void *stream)
{
	GroupMember *m = (GroupMember*)ctx;
	rb3gpu_group_s *g = m->g;
	hipStream_t st = (hipStream_t)stream;
	if (hipSetDevice(m->dev) != hipSuccess) { rb3gpu_group_abort(g); return RB3GPU_ENODEV; }
	for (int i = 0; i < 2; ++i)
		if (!m->ev[i] && hipEventCreateWithFlags(&m->ev[i], hipEventDisableTiming) != hipSuccess) { m->ev[i] = nullptr; rb3gpu_group_abort(g); return RB3GPU_ENODEV; }
	const int p = m->par;
	m->par ^= 1;
	if (hipEventRecord(m->ev[p], st) != hipSuccess) { rb3gpu_group_abort(g); return RB3GPU_ENODEV; }
	int r;
	if ((r = group_barrier(g)) < 0) return r;
	for (int q = 0; q < g->world; ++q)
		if (q != m->rank && hipStreamWaitEvent(st, g->mem[q].ev[p], 0) != hipSuccess) { rb3gpu_group_abort(g); return RB3GPU_ENODEV; }
	return 0;
}

static void group_abort_cb(void *ctx) { rb3gpu_group_abort(((GroupMember*)ctx)->g); }

/* ---- peer rounds for ranks that are PROCESSES of one node: HIP IPC (rb3gpu_ipc_peer_enable) ---- */

struct IpcShm { std::atomic<uint32_t> count, gen, aborted; };

struct IpcPeer {
	rb3gpu_comm_t inner;            // the communicator underneath: its collectives are passed through
	int rank = 0, world = 1, dev = 0;
	IpcShm *shm = nullptr;
	struct Open { int64_t key[8]; void *ptr; int rank; unsigned long used; };
	std::vector<Open> open;         // buffers of other ranks mapped into this process, by handle
	unsigned long tick = 0;
};

static int ipc_all_gather(void *ctx, const int64_t *send, int n, int64_t *recv) { IpcPeer *p = (IpcPeer*)ctx; return p->inner.all_gather(p->inner.ctx, send, n, recv); }
static int ipc_all_to_all(void *ctx, const rb3gpu_state_t *d_send, int64_t stride, const int64_t *send_cnt, rb3gpu_state_t *d_recv, const int64_t *recv_cnt, void *stream)
{
	IpcPeer *p = (IpcPeer*)ctx;
	return p->inner.all_to_all(p->inner.ctx, d_send, stride, send_cnt, d_recv, recv_cnt, stream);
}
static void ipc_abort(void *ctx)
{
	IpcPeer *p = (IpcPeer*)ctx;
	if (p->shm) p->shm->aborted.store(1u, std::memory_order_release);
	if (p->inner.abort) p->inner.abort(p->inner.ctx);
}

/* the processes meet (nobody waits for a device): a spin barrier in shared memory; a rank that has given up makes the others return */
static int ipc_barrier(IpcPeer *p)
{
	IpcShm *b = p->shm;
	if (b->aborted.load(std::memory_order_acquire)) return RB3GPU_ESTATE;
	const uint32_t my = b->gen.load(std::memory_order_acquire);
	if (b->count.fetch_add(1u, std::memory_order_acq_rel) + 1u == (uint32_t)p->world) {
		b->count.store(0u, std::memory_order_relaxed);
		b->gen.store(my + 1u, std::memory_order_release);
	} else {
		unsigned long spins = 0;
		while (b->gen.load(std::memory_order_acquire) == my && !b->aborted.load(std::memory_order_relaxed)) {
			if (++spins < 4000) __builtin_ia32_pause(); else sched_yield();
			if (spins > 4000ul + 60000000ul) { b->aborted.store(1u, std::memory_order_release); break; } // (a rank that died: minutes of yields, then everybody gives up)
		}
	}
	return b->aborted.load(std::memory_order_acquire) ? RB3GPU_ESTATE : 0;
}

/* Between processes the streams cannot wait for each other: on this runtime (ROCm 7.2) hipStreamWaitEvent answers "invalid argument" to an event that came through
 * hipIpcOpenEventHandle.  So a rank waits for ITS OWN stream -- its round's kernel and with it every store into the other ranks' buffers is done --, then the
 * processes meet at the barrier in shared memory: one host synchronisation per round, but still no collective, no read-back and no copy. */
static int ipc_stream_barrier(void *ctx, void *stream)
{
	IpcPeer *p = (IpcPeer*)ctx;
	hipStream_t st = (hipStream_t)stream;
	hipError_t e = hipSetDevice(p->dev);
	if (e == hipSuccess) e = hipStreamSynchronize(st);
	if (e != hipSuccess) { fprintf(stderr, "[E::rb3gpu] peer rounds between processes, rank %d: %s\n", p->rank, hipGetErrorString(e)); (void)hipGetLastError(); ipc_abort(p); return RB3GPU_ENODEV; }
	return ipc_barrier(p);
}

static int ipc_peer_export(void *ctx, void *d_ptr, int64_t handle[8])
{
	(void)ctx;
	hipIpcMemHandle_t hd;
	static_assert(sizeof(hd) == 64, "an IPC memory handle is 8 words");
	{ const hipError_t e = hipIpcGetMemHandle(&hd, d_ptr); if (e != hipSuccess) { fprintf(stderr, "[E::rb3gpu] peer rounds between processes: hipIpcGetMemHandle(%p): %s\n", d_ptr, hipGetErrorString(e)); (void)hipGetLastError(); return RB3GPU_ENODEV; } }
	memcpy(handle, &hd, 64);
	return 0;
}

static void *ipc_peer_import(void *ctx, int rank, const int64_t handle[8])
{
	IpcPeer *p = (IpcPeer*)ctx;
	if (handle == nullptr) { // that rank is about to replace its buffers: what is mapped of them here goes first
		(void)hipSetDevice(p->dev);
		for (size_t i = 0; i < p->open.size(); )
			if (p->open[i].rank == rank) { (void)hipIpcCloseMemHandle(p->open[i].ptr); (void)hipGetLastError(); p->open.erase(p->open.begin() + (long)i); } else ++i;
		return nullptr;
	}
	++p->tick;
	for (auto &o : p->open)
		if (o.rank == rank && memcmp(o.key, handle, 64) == 0) { o.used = p->tick; return o.ptr; }
	// a rank shows three buffers at a time (two receive buffers, its counters): more than six mappings of one rank are buffers it has replaced since
	int n_of = 0;
	size_t oldest = 0;
	for (size_t i = 0; i < p->open.size(); ++i)
		if (p->open[i].rank == rank) { if (n_of == 0 || p->open[i].used < p->open[oldest].used) oldest = i; ++n_of; }
	if (n_of >= 6) { (void)hipIpcCloseMemHandle(p->open[oldest].ptr); (void)hipGetLastError(); p->open.erase(p->open.begin() + (long)oldest); }
	hipIpcMemHandle_t hd;
	memcpy(&hd, handle, 64);
	void *ptr = nullptr;
	{ hipError_t e = hipSetDevice(p->dev); if (e == hipSuccess) e = hipIpcOpenMemHandle(&ptr, hd, hipIpcMemLazyEnablePeerAccess);
	  if (e != hipSuccess || ptr == nullptr) { fprintf(stderr, "[E::rb3gpu] peer rounds between processes, rank %d: hipIpcOpenMemHandle of a buffer of rank %d: %s\n", p->rank, rank, hipGetErrorString(e)); (void)hipGetLastError(); return nullptr; } }
	IpcPeer::Open o;
	memcpy(o.key, handle, 64), o.ptr = ptr, o.rank = rank, o.used = p->tick;
	p->open.push_back(o);
	return ptr;
}

static void ipc_free(IpcPeer *p)
{
	if (!p) return;
	(void)hipSetDevice(p->dev);
	for (auto &o : p->open) (void)hipIpcCloseMemHandle(o.ptr);
	if (p->shm) munmap(p->shm, 4096);
	(void)hipGetLastError();
	delete p;
}

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
	memset(g->mem, 0, sizeof(g->mem));
	return g;
}

int rb3gpu_group_comm(rb3gpu_group_t *g, int rank, rb3gpu_t *h, rb3gpu_comm_t *comm)
{
	if (!g || !h || !comm || rank < 0 || rank >= g->world) return RB3GPU_EINVAL;
	GroupMember *m = &g->mem[rank];
	m->g = g, m->rank = rank, m->dev = rb3gpu_device_of(h); // (m->par stays: the ranks pass their stream barriers together, whichever communicator structs they go through)
	// direct peer access where the devices differ (an error here only means "already enabled" or "copies get staged": both fine)
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) == hipSuccess && hipSetDevice(m->dev) == hipSuccess)
		for (int d = 0; d < ndev; ++d)
			if (d != m->dev) { int can = 0; if (hipDeviceCanAccessPeer(&can, m->dev, d) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(d, 0); (void)hipGetLastError(); }
	comm->ctx = m, comm->rank = rank, comm->world = g->world;
	comm->all_gather = group_all_gather, comm->all_to_all = group_all_to_all, comm->abort = group_abort_cb;
	// peer rounds: this rank must reach the memory of every device of the node that another rank may sit on (the ranks agree on the path
	// among themselves before they take it: rb3gpu.hip, sh_merge_impl)
	bool reach = true;
	for (int d = 0; d < ndev; ++d)
		if (d != m->dev) { int can = 0; if (hipDeviceCanAccessPeer(&can, m->dev, d) != hipSuccess || !can) reach = false; }
	(void)hipGetLastError();
	comm->stream_barrier = reach && !getenv("RB3GPU_NO_PEER_ROUNDS") ? group_stream_barrier : nullptr;
	comm->peer_export = nullptr, comm->peer_import = nullptr; // (threads of one process: a device pointer is what it is)
	return 0;
}

void rb3gpu_group_abort(rb3gpu_group_t *g)
{
	if (!g) return;
	pthread_mutex_lock(&g->mtx);
	g->aborted.store(1, std::memory_order_release);
	pthread_cond_broadcast(&g->cv);
	pthread_mutex_unlock(&g->mtx);
}

int rb3gpu_ipc_peer_enable(rb3gpu_t *h, rb3gpu_comm_t *comm)
{
	if (!h || !comm || !comm->all_gather || comm->world < 2 || comm->world > RB3GPU_SH_MAXIV || comm->stream_barrier) return RB3GPU_EINVAL;
	IpcPeer *p = new (std::nothrow) IpcPeer;
	if (!p) return RB3GPU_ENOMEM;
	p->inner = *comm, p->rank = comm->rank, p->world = comm->world, p->dev = rb3gpu_device_of(h);
	const int W = p->world;
	int64_t ok = 1;
	// 1. the barrier: rank 0 makes the shared-memory object, everybody maps it, rank 0 takes the name away again
	int64_t name8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	std::vector<int64_t> all((size_t)W * 16);
	int fd = -1;
	if (p->rank == 0) {
		struct timespec ts;
		clock_gettime(CLOCK_MONOTONIC, &ts);
		snprintf((char*)name8, 63, "/rb3gpu_%d_%ld%09ld", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec);
		fd = shm_open((const char*)name8, O_CREAT | O_EXCL | O_RDWR, 0600);
		if (fd < 0 || ftruncate(fd, 4096) != 0) ok = 0;
	}
	if (comm->all_gather(comm->ctx, name8, 8, all.data()) < 0) { if (fd >= 0) { close(fd); shm_unlink((const char*)name8); } delete p; return RB3GPU_ENODEV; }
	char name[64];
	memcpy(name, all.data(), 64), name[63] = 0; // (rank 0's)
	if (p->rank != 0 && name[0]) fd = shm_open(name, O_RDWR, 0600);
	if (fd >= 0) {
		void *m = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
		close(fd);
		if (m != MAP_FAILED) p->shm = (IpcShm*)m; // (a fresh object is all zero: count, generation and the abort flag start at 0)
	}
	if (!p->shm) ok = 0;
	// 2. (a second all-gather: everybody has mapped the object -- or has failed to, and says so below -- before rank 0 takes its name away)
	if (hipSetDevice(p->dev) != hipSuccess) ok = 0;
	{
		int64_t dummy = 0;
		if (comm->all_gather(comm->ctx, &dummy, 1, all.data()) < 0) { if (p->rank == 0 && name[0]) shm_unlink(name); ipc_free(p); return RB3GPU_ENODEV; }
	}
	if (p->rank == 0 && name[0]) shm_unlink(name);
	// 3. everywhere or nowhere
	std::vector<int64_t> oks((size_t)W);
	if (comm->all_gather(comm->ctx, &ok, 1, oks.data()) < 0) { ipc_free(p); return RB3GPU_ENODEV; }
	for (int o = 0; o < W; ++o) ok = ok && oks[(size_t)o] != 0;
	if (!ok) { ipc_free(p); return RB3GPU_EUNSUP; }
	comm->ctx = p;
	comm->all_gather = ipc_all_gather, comm->all_to_all = p->inner.all_to_all ? ipc_all_to_all : nullptr, comm->abort = ipc_abort;
	comm->stream_barrier = ipc_stream_barrier, comm->peer_export = ipc_peer_export, comm->peer_import = ipc_peer_import;
	return 0;
}

void rb3gpu_ipc_peer_disable(rb3gpu_comm_t *comm)
{
	if (!comm || comm->stream_barrier != ipc_stream_barrier || !comm->ctx) return;
	IpcPeer *p = (IpcPeer*)comm->ctx;
	*comm = p->inner;
	ipc_free(p);
}

void rb3gpu_group_destroy(rb3gpu_group_t *g)
{
	if (!g) return;
	for (int q = 0; q < g->world; ++q)
		for (int i = 0; i < 2; ++i)
			if (g->mem[q].ev[i]) { (void)hipSetDevice(g->mem[q].dev); (void)hipEventDestroy(g->mem[q].ev[i]); }
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
	comm->stream_barrier = nullptr, comm->peer_export = nullptr, comm->peer_import = nullptr; // (processes: a pointer of another rank means nothing here -- rb3gpu_ipc_peer_enable wraps this communicator for peer rounds)
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

/* Nothing of the index and nothing of a batch but one byte per symbol is ever whole on one device ("what's missing" 1 and 2):
 *   split    an interval at a time: the symbols of interval i are exported by h0 (1 byte each, len_i of them), copied across, built into
 *            handle i; h0 rebuilds itself from its own interval last.  Bounds by BYTES of the block array (rb3gpu_balanced_bounds).
 *   merge    device 0 (where the sorter left the batch) makes the byte array "symbol before text position t" from the text-order words;
 *            every rank PULLS that array (len bytes) and the text-order words of ITS text range (8 len / n bytes) at the same time, each on
 *            its own stream over its own xGMI link, then the ranks walk (rb3gpu_sh_merge_text): states hop between the intervals, records
 *            are (text position, insertion point), and at the end one all-to-all asks the owners of the text ranges for the rows.
 *            The rank threads live as long as the object; peer access is set up once.
 *   export   the writers take the intervals in rank order (rb3gpu_shard_export_runs / _run_words): runs that meet at a seam are joined, as
 *            rld_enc joins what the ropes of the reference hand it one after the other (fm-index.c:31-54, rld0.c:153-161).  No gather.
 *   gather   kept for the one caller that needs the whole index on one device (a batch the host had to sort: merged the ordinary way). */
struct ShardJob { int kind; int64_t len; const uint8_t *d_tprev0; const uint64_t *d_tw0; int64_t n_chains; const int64_t *chain_tp; int64_t bounds[RB3GPU_SH_MAXIV + 1]; int64_t rounds; int ret; };

struct rb3gpu_shard_s {
	int n = 0;
	rb3gpu_t *h[RB3GPU_SH_MAXIV];        // h[0] is the caller's
	int dev[RB3GPU_SH_MAXIV];
	int64_t bounds[RB3GPU_SH_MAXIV + 1];
	rb3gpu_group_t *grp = nullptr;
	rb3gpu_comm_t comm[RB3GPU_SH_MAXIV];
	void *rep_tp[RB3GPU_SH_MAXIV], *rep_tw[RB3GPU_SH_MAXIV]; // per rank on another device than 0: the symbol-before array (whole) and the rank's slice of the text-order words
	int64_t rep_tp_cap[RB3GPU_SH_MAXIV], rep_tw_cap[RB3GPU_SH_MAXIV];
	void *d_tprev0 = nullptr; int64_t tprev0_cap = 0;        // on device 0
	// the rank threads (ranks 1 .. n-1; rank 0 runs on the caller's thread)
	pthread_t th[RB3GPU_SH_MAXIV];
	pthread_mutex_t mtx;
	pthread_cond_t cv;
	unsigned long gen = 0;
	int done = 0, quit = 0, n_threads = 0;
	int n_rebalanced = 0;
	int n_rebalance_skipped = 0; // rebalances given up before anything was touched (no memory for the new ranges): the intervals stayed as they were
	int verbose = 1;
	int rebalance_pct = 25;   // RB3GPU_SHARD_REBALANCE_PCT, read once when the object is made (25: SURVEY 8(e); -1: never; 0: whenever the shares differ at all -- tests)
	ShardJob job[RB3GPU_SH_MAXIV];
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

/* rank `rank` pulls what it needs of the batch from device 0 (on the stream of its own handle: all ranks at once, a link each) and walks */
static void shard_rank_merge(rb3gpu_shard_s *s, int rank)
{
	ShardJob *j = &s->job[rank];
	const int64_t len = j->len, w = s->n;
	const int64_t t_lo = len / w * rank + (len % w) * rank / w, t_hi = rank + 1 == w ? len : len / w * (rank + 1) + (len % w) * (rank + 1) / w;
	const uint8_t *tprev = j->d_tprev0;
	const uint64_t *tws = j->d_tw0 + t_lo;
	int r = 0;
	if (s->dev[rank] != s->dev[0]) {
		const int64_t nsl = t_hi - t_lo;
		if (s->rep_tp_cap[rank] < len) {
			if (s->rep_tp[rank]) (void)rb3gpu_dev_free(s->h[rank], s->rep_tp[rank]);
			s->rep_tp[rank] = nullptr, s->rep_tp_cap[rank] = 0;
			if ((r = rb3gpu_dev_alloc(s->h[rank], len + (len >> 2) + 64, &s->rep_tp[rank])) == 0) s->rep_tp_cap[rank] = len + (len >> 2);
		}
		if (r == 0 && s->rep_tw_cap[rank] < nsl) {
			if (s->rep_tw[rank]) (void)rb3gpu_dev_free(s->h[rank], s->rep_tw[rank]);
			s->rep_tw[rank] = nullptr, s->rep_tw_cap[rank] = 0;
			if ((r = rb3gpu_dev_alloc(s->h[rank], (nsl + (nsl >> 2) + 8) * 8, &s->rep_tw[rank])) == 0) s->rep_tw_cap[rank] = nsl + (nsl >> 2);
		}
		if (r == 0) {
			hipStream_t st = (hipStream_t)rb3gpu_stream_of(s->h[rank]);
			hipError_t e = hipSetDevice(s->dev[rank]);
			if (e == hipSuccess) e = hipMemcpyPeerAsync(s->rep_tp[rank], s->dev[rank], j->d_tprev0, s->dev[0], (size_t)len, st);
			if (e == hipSuccess && nsl > 0) e = hipMemcpyPeerAsync(s->rep_tw[rank], s->dev[rank], j->d_tw0 + t_lo, s->dev[0], (size_t)nsl * 8, st);
			if (e == hipSuccess) e = hipStreamSynchronize(st);
			if (e != hipSuccess) { (void)hipGetLastError(); r = RB3GPU_ENODEV; }
		}
		tprev = (const uint8_t*)s->rep_tp[rank], tws = (const uint64_t*)s->rep_tw[rank];
	}
	if (r < 0) { rb3gpu_group_abort(s->grp); j->ret = r; return; }
	j->ret = rb3gpu_sh_merge_text(s->h[rank], &s->comm[rank], j->bounds, len, tprev, tws, j->n_chains, j->chain_tp, 1, &j->rounds);
}

struct ShardThreadArg { rb3gpu_shard_s *s; int rank; };

static void *shard_thread(void *arg)
{
	ShardThreadArg *a = (ShardThreadArg*)arg;
	rb3gpu_shard_s *s = a->s;
	const int rank = a->rank;
	delete a;
	unsigned long seen = 0;
	for (;;) {
		pthread_mutex_lock(&s->mtx);
		while (s->gen == seen && !s->quit) pthread_cond_wait(&s->cv, &s->mtx);
		const int quit = s->quit;
		seen = s->gen;
		pthread_mutex_unlock(&s->mtx);
		if (quit) return nullptr;
		shard_rank_merge(s, rank);
		pthread_mutex_lock(&s->mtx);
		++s->done;
		pthread_cond_broadcast(&s->cv);
		pthread_mutex_unlock(&s->mtx);
	}
}

extern "C" {

static void shard_free(rb3gpu_shard_s *s, bool handles)
{
	if (s->n_threads > 0) {
		pthread_mutex_lock(&s->mtx);
		s->quit = 1;
		pthread_cond_broadcast(&s->cv);
		pthread_mutex_unlock(&s->mtx);
		for (int i = 1; i <= s->n_threads; ++i) pthread_join(s->th[i], nullptr);
	}
	if (s->d_tprev0 && s->h[0]) (void)rb3gpu_dev_free(s->h[0], s->d_tprev0);
	for (int i = 1; i < s->n; ++i) {
		if (s->h[i]) {
			if (s->rep_tp[i]) (void)rb3gpu_dev_free(s->h[i], s->rep_tp[i]);
			if (s->rep_tw[i]) (void)rb3gpu_dev_free(s->h[i], s->rep_tw[i]);
			if (handles) rb3gpu_destroy(s->h[i]);
		}
	}
	rb3gpu_group_destroy(s->grp);
	pthread_mutex_destroy(&s->mtx);
	pthread_cond_destroy(&s->cv);
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
	s->verbose = opt->verbose;
	{ const char *e = getenv("RB3GPU_SHARD_REBALANCE_PCT"); if (e && *e) s->rebalance_pct = atoi(e); }
	pthread_mutex_init(&s->mtx, nullptr);
	pthread_cond_init(&s->cv, nullptr);
	for (int i = 0; i < RB3GPU_SH_MAXIV; ++i) s->h[i] = nullptr, s->rep_tp[i] = s->rep_tw[i] = nullptr, s->rep_tp_cap[i] = s->rep_tw_cap[i] = 0, s->dev[i] = 0;
	s->h[0] = h0;
	int r = rb3gpu_balanced_bounds(h0, n, s->bounds); // about equal BYTES of runs per interval
	if (r < 0 || (s->grp = rb3gpu_group_create(n)) == nullptr) { pthread_mutex_destroy(&s->mtx); pthread_cond_destroy(&s->cv); delete s; return nullptr; }
	int64_t acc0[RB3GPU_ASIZE + 1], sum[RB3GPU_ASIZE] = {0, 0, 0, 0, 0, 0};
	r = rb3gpu_get_acc(h0, acc0);
	// an interval at a time, the last one first and h0's own last of all (it rebuilds itself from its interval: until then it holds the whole index):
	// never more than one interval's symbols (1 byte each) beside the index
	for (int i = n - 1; i >= 0 && r == 0; --i) {
		s->dev[i] = devices[i];
		const int64_t len = s->bounds[i + 1] - s->bounds[i];
		void *part0 = nullptr, *part = nullptr;
		if (n == 1) break;
		if ((r = rb3gpu_dev_alloc(h0, len, &part0)) < 0) break;
		r = rb3gpu_export_plain_range_dev(h0, s->bounds[i], s->bounds[i + 1], (uint8_t*)part0);
		if (r == 0 && i == 0) r = rb3gpu_from_plain_dev(h0, len, (const uint8_t*)part0);
		else if (r == 0) {
			rb3gpu_opt_t o = *opt;
			o.device = devices[i];
			if ((s->h[i] = rb3gpu_create(&o)) == nullptr) r = RB3GPU_ENODEV;
			if (r == 0 && s->dev[i] == s->dev[0]) r = rb3gpu_from_plain_dev(s->h[i], len, (const uint8_t*)part0);
			else if (r == 0 && (r = rb3gpu_dev_alloc(s->h[i], len, &part)) == 0) {
				r = copy_across(part, s->dev[i], part0, s->dev[0], (size_t)len);
				if (r == 0) r = rb3gpu_from_plain_dev(s->h[i], len, (const uint8_t*)part);
				(void)rb3gpu_dev_free(s->h[i], part);
			}
		}
		(void)rb3gpu_dev_free(h0, part0);
	}
	for (int i = 0; i < n && r == 0; ++i) { // the intervals together hold the symbols the index held (a copy that went wrong would show here, not in a wrong BWT later)
		int64_t acc[RB3GPU_ASIZE + 1];
		if ((r = rb3gpu_get_acc(s->h[i], acc)) == 0)
			for (int c = 0; c < RB3GPU_ASIZE; ++c) sum[c] += acc[c + 1] - acc[c];
	}
	for (int c = 0; c < RB3GPU_ASIZE && r == 0; ++c) if (sum[c] != acc0[c + 1] - acc0[c]) r = RB3GPU_EINTERNAL;
	// the communicators (peer access between the devices is switched on here, once) and the rank threads
	for (int i = 0; i < n && r == 0; ++i) r = rb3gpu_group_comm(s->grp, i, s->h[i], &s->comm[i]);
	for (int i = 1; i < n && r == 0; ++i) {
		ShardThreadArg *a = new (std::nothrow) ShardThreadArg;
		if (!a) { r = RB3GPU_ENOMEM; break; }
		a->s = s, a->rank = i;
		if (pthread_create(&s->th[i], nullptr, shard_thread, a) != 0) { delete a; r = RB3GPU_ENOMEM; break; }
		s->n_threads = i;
	}
	if (r < 0) { shard_free(s, true); return nullptr; } // (h0 may hold its first interval only: the caller gives the build up)
	return s;
}

int rb3gpu_shard_merge(rb3gpu_shard_t *s, int64_t len, const uint8_t *d_bwt, const uint64_t *d_tw, int64_t n_chains, const int64_t *chain_tp, int64_t *n_rounds)
{
	if (!s || len <= 0 || !d_tw || n_chains <= 0 || !chain_tp) return RB3GPU_EINVAL;
	(void)d_bwt; // (the symbol of a row comes back with its number: rb3gpu_sh_merge_text)
	int r;
	// the symbol before every text position, one byte each: all a walking rank needs of the batch
	if (s->tprev0_cap < len) {
		if (s->d_tprev0) (void)rb3gpu_dev_free(s->h[0], s->d_tprev0);
		s->d_tprev0 = nullptr, s->tprev0_cap = 0;
		if ((r = rb3gpu_dev_alloc(s->h[0], len + (len >> 2) + 64, &s->d_tprev0)) < 0) return r;
		s->tprev0_cap = len + (len >> 2);
	}
	if ((r = rb3gpu_tprev_from_tw(s->h[0], len, d_tw, (uint8_t*)s->d_tprev0)) < 0) return r;
	for (int i = 0; i < s->n; ++i) {
		ShardJob *j = &s->job[i];
		j->kind = 1, j->len = len, j->d_tprev0 = (const uint8_t*)s->d_tprev0, j->d_tw0 = d_tw, j->n_chains = n_chains, j->chain_tp = chain_tp, j->rounds = 0, j->ret = 0;
		memcpy(j->bounds, s->bounds, sizeof(s->bounds));
	}
	pthread_mutex_lock(&s->mtx);
	s->done = 0, ++s->gen;
	pthread_cond_broadcast(&s->cv);
	pthread_mutex_unlock(&s->mtx);
	shard_rank_merge(s, 0); // interval 0 on the calling thread
	pthread_mutex_lock(&s->mtx);
	while (s->done < s->n_threads) pthread_cond_wait(&s->cv, &s->mtx);
	pthread_mutex_unlock(&s->mtx);
	r = 0;
	for (int i = 0; i < s->n; ++i) if (s->job[i].ret < 0 && (r == 0 || r == RB3GPU_ESTATE)) r = s->job[i].ret; // (ESTATE: a rank that was only woken up by another one's failure)
	if (r == 0) {
		memcpy(s->bounds, s->job[0].bounds, sizeof(s->bounds));
		if (n_rounds) *n_rounds = s->job[0].rounds;
		const int pct = s->rebalance_pct;
		if (pct >= 0 && s->n > 1) {
			const int rr = rb3gpu_shard_rebalance(s, pct);
			if (rr < 0) r = rr;
			else s->n_rebalanced += rr;
		}
	}
	return r;
}

/* The intervals grow unevenly (a batch lands where its strings sort): when the largest holds more than `pct` per cent more BYTES of the block array than
 * the mean (25), the bounds are moved to where equal shares of the bytes lie -- found from 64 byte-quantiles of every interval -- and every
 * interval is rebuilt from the symbols of its new range: each device gets its range as plain symbols (1 byte each, its own share only) from the
 * neighbours that held them, device to device, and rebuilds its handle in place.  Nothing is ever whole on one device.  1: rebalanced, 0: not needed. */
int rb3gpu_shard_rebalance(rb3gpu_shard_t *s, int pct)
{
	if (!s) return RB3GPU_EINVAL;
	const int n = s->n;
	if (n < 2) return 0;
	enum { Q = 64 };
	double bytes[RB3GPU_SH_MAXIV], total = 0, mx = 0;
	for (int i = 0; i < n; ++i) {
		rb3gpu_stats_t st;
		int r = rb3gpu_stats(s->h[i], &st);
		if (r < 0) return r;
		bytes[i] = (double)st.bytes_index, total += bytes[i], mx = bytes[i] > mx ? bytes[i] : mx;
	}
	if (pct >= 0 && mx * n <= total * (1.0 + pct / 100.0)) return 0;
	// the global profile: (position, bytes before it) at Q + 1 points per interval
	std::vector<double> px((size_t)n * Q + 1), py((size_t)n * Q + 1);
	double before = 0;
	int r = 0;
	for (int i = 0; i < n && r == 0; ++i) {
		int64_t qb[Q + 1];
		if (s->bounds[i + 1] - s->bounds[i] >= Q) r = rb3gpu_balanced_bounds(s->h[i], Q, qb);
		else for (int k = 0; k <= Q; ++k) qb[k] = (s->bounds[i + 1] - s->bounds[i]) * k / Q;
		for (int k = 0; k < Q; ++k) px[(size_t)i * Q + k] = (double)(s->bounds[i] + qb[k]), py[(size_t)i * Q + k] = before + bytes[i] * k / Q;
		before += bytes[i];
	}
	if (r < 0) return r;
	px[(size_t)n * Q] = (double)s->bounds[n], py[(size_t)n * Q] = total;
	int64_t nb[RB3GPU_SH_MAXIV + 1];
	nb[0] = 0, nb[n] = s->bounds[n];
	size_t at = 0;
	for (int j = 1; j < n; ++j) {
		const double want = total * j / n;
		while (at + 1 < px.size() - 1 && py[at + 1] < want) ++at;
		const double dy = py[at + 1] - py[at], f = dy > 0 ? (want - py[at]) / dy : 0.0;
		int64_t b = (int64_t)(px[at] + f * (px[at + 1] - px[at]));
		if (b <= nb[j - 1]) b = nb[j - 1] + 1;
		if (b > nb[n] - (n - j)) b = nb[n] - (n - j);
		nb[j] = b;
	}
	int64_t acc0[RB3GPU_ASIZE + 1], acc1[RB3GPU_ASIZE + 1];
	if ((r = rb3gpu_shard_get_acc(s, acc0)) < 0) return r;
	// phase 1: every device collects the symbols of its new range (the old handles stay as they are until all have)
	void *plain[RB3GPU_SH_MAXIV];
	for (int j = 0; j < n; ++j) plain[j] = nullptr;
	for (int j = 0; j < n && r == 0; ++j) {
		const int64_t len = nb[j + 1] - nb[j];
		if ((r = rb3gpu_dev_alloc(s->h[j], len + 64, &plain[j])) < 0) break;
		for (int i = 0; i < n && r == 0; ++i) {
			const int64_t a = nb[j] > s->bounds[i] ? nb[j] : s->bounds[i], b = nb[j + 1] < s->bounds[i + 1] ? nb[j + 1] : s->bounds[i + 1];
			if (a >= b) continue;
			uint8_t *dst = (uint8_t*)plain[j] + (a - nb[j]);
			if (s->dev[i] == s->dev[j]) { r = rb3gpu_export_plain_range_dev(s->h[i], a - s->bounds[i], b - s->bounds[i], dst); continue; }
			void *tmp = nullptr;
			if ((r = rb3gpu_dev_alloc(s->h[i], b - a + 64, &tmp)) < 0) break;
			if ((r = rb3gpu_export_plain_range_dev(s->h[i], a - s->bounds[i], b - s->bounds[i], (uint8_t*)tmp)) == 0) r = copy_across(dst, s->dev[j], tmp, s->dev[i], (size_t)(b - a));
			(void)rb3gpu_dev_free(s->h[i], tmp);
		}
	}
	if (r < 0) {
		// Nothing has been touched yet: the old handles hold the whole index with the old bounds.  The rebalance is an optimisation of a build whose
		// merge has already succeeded, so running out of memory HERE -- exactly when one device is fuller than the rest -- means "not rebalanced",
		// not "give the build up" (ADVICE r5): free what was collected, keep the bounds, say so at verbose >= 2.
		for (int j = 0; j < n; ++j) if (plain[j]) (void)rb3gpu_dev_free(s->h[j], plain[j]);
		if (s->verbose >= 2) fprintf(stderr, "[W::rb3gpu_shard_rebalance] could not collect the new ranges (error %d): the intervals stay as they are\n", r);
		++s->n_rebalance_skipped;
		return 0;
	}
	// phase 2: every handle rebuilt from its new range
	for (int j = 0; j < n && r == 0; ++j) r = rb3gpu_from_plain_dev(s->h[j], nb[j + 1] - nb[j], (const uint8_t*)plain[j]);
	for (int j = 0; j < n; ++j) if (plain[j]) (void)rb3gpu_dev_free(s->h[j], plain[j]);
	if (r < 0) return r; // (a handle has been rebuilt and another could not be: the index is lost, the caller gives the build up)
	memcpy(s->bounds, nb, (size_t)(n + 1) * 8);
	if ((r = rb3gpu_shard_get_acc(s, acc1)) < 0) return r;
	for (int c = 0; c <= RB3GPU_ASIZE; ++c) if (acc0[c] != acc1[c]) return RB3GPU_EINTERNAL; // (the symbols the intervals held, no more and no fewer)
	return 1;
}

int rb3gpu_shard_get_acc(const rb3gpu_shard_t *s, int64_t acc[RB3GPU_ASIZE + 1])
{
	if (!s || !acc) return RB3GPU_EINVAL;
	int64_t sum[RB3GPU_ASIZE] = {0, 0, 0, 0, 0, 0}, a[RB3GPU_ASIZE + 1];
	for (int i = 0; i < s->n; ++i) {
		const int r = rb3gpu_get_acc(s->h[i], a);
		if (r < 0) return r;
		for (int c = 0; c < RB3GPU_ASIZE; ++c) sum[c] += a[c + 1] - a[c];
	}
	acc[0] = 0;
	for (int c = 0; c < RB3GPU_ASIZE; ++c) acc[c + 1] = acc[c] + sum[c];
	return 0;
}

/* runs of the intervals in rank order, the ones that meet at a seam joined */
struct SeamRuns { rb3gpu_emit_f emit; void *data; int c; int64_t l; };
static int seam_emit(void *data, int c, int64_t l)
{
	SeamRuns *q = (SeamRuns*)data;
	if (l <= 0) return 0;
	if (c == q->c) { q->l += l; return 0; }
	if (q->l > 0) { const int r = q->emit(q->data, q->c, q->l); if (r != 0) return r; }
	q->c = c, q->l = l;
	return 0;
}

int rb3gpu_shard_export_runs(rb3gpu_shard_t *s, rb3gpu_emit_f emit, void *data)
{
	if (!s || !emit) return RB3GPU_EINVAL;
	SeamRuns q = { emit, data, -1, 0 };
	for (int i = 0; i < s->n; ++i) {
		if (rb3gpu_get_tot(s->h[i]) != s->bounds[i + 1] - s->bounds[i]) return RB3GPU_EINTERNAL;
		const int r = rb3gpu_export_runs(s->h[i], seam_emit, &q);
		if (r != 0) return r;
	}
	if (q.l > 0) return emit(data, q.c, q.l);
	return 0;
}

/* the same in bulk (words start << 3 | sym of maximal runs, starts counted from the beginning of the WHOLE index) */
struct SeamWords { rb3gpu_emit_words_f emit; void *data; int64_t base; int last_c; uint64_t *buf; int64_t cap; };
static int seam_words(void *data, int64_t n, const uint64_t *words, int64_t end)
{
	SeamWords *q = (SeamWords*)data;
	if (n <= 0) return 0; // (the closing call of an interval: the run goes on into the next interval, or is closed by the caller at the very end)
	(void)end;
	int64_t i0 = 0;
	if ((int)(words[0] & 7) == q->last_c) i0 = 1; // the first run of this call continues the last one of the call (or interval) before
	if (n - i0 > q->cap) {
		uint64_t *nb = (uint64_t*)realloc(q->buf, (size_t)(n - i0) * 8);
		if (!nb) return RB3GPU_ENOMEM;
		q->buf = nb, q->cap = n - i0;
	}
	for (int64_t i = i0; i < n; ++i) q->buf[i - i0] = (((words[i] >> 3) + (uint64_t)q->base) << 3) | (words[i] & 7);
	q->last_c = (int)(words[n - 1] & 7);
	return n > i0 ? q->emit(q->data, n - i0, q->buf, -1) : 0;
}

int rb3gpu_shard_export_run_words(rb3gpu_shard_t *s, rb3gpu_emit_words_f emit, void *data)
{
	if (!s || !emit) return RB3GPU_EINVAL;
	SeamWords q = { emit, data, 0, -1, nullptr, 0 };
	int r = 0;
	for (int i = 0; i < s->n && r == 0; ++i) {
		if (rb3gpu_get_tot(s->h[i]) != s->bounds[i + 1] - s->bounds[i]) { r = RB3GPU_EINTERNAL; break; }
		q.base = s->bounds[i];
		r = rb3gpu_export_run_words(s->h[i], seam_words, &q);
	}
	free(q.buf);
	if (r == 0) r = emit(data, 0, nullptr, s->bounds[s->n]); // closes the last run
	return r;
}

void rb3gpu_shard_destroy(rb3gpu_shard_t *s)
{
	if (s) shard_free(s, true); // (h0, the caller's handle, stays: it holds interval 0)
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
