// fi_comm.hip -- slab-to-slab exchange over RCCL (xGMI).
//
// The reference is single-process (no collectives of any kind); this file exists
// because the lattice is domain-decomposed along its slowest axis, one slab per GPU / process:
//   exchange_halo   the `halo` ghost planes of the CG search direction go to the two neighbouring
//                   slabs: grouped ncclSend/ncclRecv on the solver stream, contiguous planes, no packing
//                   (a slab only talks to 2 of the 7 xGMI peers);
//   allreduce_sum   the 1-2 fp64 dot-product scalars of a CG iteration: ncclAllReduce in place on the
//                   device-resident CgScalars, no host round trip.
// RCCL is loaded lazily with dlopen so that a single-GPU run never pays for it (and so that the library
// binds to whichever librccl.so.1 the process already holds, e.g. the one torch.distributed loaded).

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>
#include <rccl/rccl.h>

#include "fi_internal.h"

namespace fi {

// Test transport (fi_comm_init_host): the same two operations carried through a POSIX shared-memory segment by
// host copies, for ranks that share ONE GPU (RCCL refuses two ranks on one device).  It lets the real per-process
// orchestration -- rank bootstrap, point filtering, per-rank assembly, rank-set solvers, bench.py --gpus N -- run on a
// single-GPU machine; only the RCCL wire itself stays untested there.  Not a production path: every exchange
// synchronises the stream and both ranks' hosts.
struct HostShm {
	uint32_t              magic;
	uint32_t              nranks;
	uint64_t              slot_bytes;
	std::atomic<uint32_t> arrived;
	std::atomic<uint32_t> generation;
	std::atomic<uint32_t> failed;  // a rank that gives up (timeout, HIP error) releases everybody else
};
struct HostComm {
	std::string name;
	int         fd = -1;
	size_t      bytes = 0;
	char*       base = nullptr;
	bool        owner = false;
	HostShm*    hdr() const { return reinterpret_cast<HostShm*>(base); }
	char*       slot(int r) const { return base + 4096 + static_cast<size_t>(r) * hdr()->slot_bytes; }
};

struct Comm {
	ncclComm_t comm = nullptr;
	HostComm*  host = nullptr;
	long       n_allreduce = 0;  // all-reduces issued through this communicator (scalars and vectors): fi_stats.reductions
};

const char* rccl_error_string(ncclResult_t r);

namespace {

struct Rccl {
	void* handle = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char* (*GetErrorString)(ncclResult_t) = nullptr;
	// diagnostics (fi_comm_info): optional -- a library without them reports -1
	ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
	ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
	ncclResult_t (*CommCuDevice)(const ncclComm_t, int*) = nullptr;
};

Rccl& rccl()
{
	static Rccl r;
	if (r.handle) { return r; }
	// resolved into a local table first: a missing symbol must not leave a half-filled table behind a set handle
	Rccl t;
	const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	for (const char* n : names) {
		t.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
		if (t.handle) { break; }
	}
	FI_REQUIRE(t.handle != nullptr, FI_ERR_COMM, "cannot load librccl: %s", dlerror());
	auto sym = [&](const char* name) {
		void* p = dlsym(t.handle, name);
		FI_REQUIRE(p != nullptr, FI_ERR_COMM, "librccl lacks %s", name);
		return p;
	};
	t.GetUniqueId    = reinterpret_cast<decltype(t.GetUniqueId)>(sym("ncclGetUniqueId"));
	t.CommInitRank   = reinterpret_cast<decltype(t.CommInitRank)>(sym("ncclCommInitRank"));
	t.CommDestroy    = reinterpret_cast<decltype(t.CommDestroy)>(sym("ncclCommDestroy"));
	t.AllReduce      = reinterpret_cast<decltype(t.AllReduce)>(sym("ncclAllReduce"));
	t.Send           = reinterpret_cast<decltype(t.Send)>(sym("ncclSend"));
	t.Recv           = reinterpret_cast<decltype(t.Recv)>(sym("ncclRecv"));
	t.GroupStart     = reinterpret_cast<decltype(t.GroupStart)>(sym("ncclGroupStart"));
	t.GroupEnd       = reinterpret_cast<decltype(t.GroupEnd)>(sym("ncclGroupEnd"));
	t.GetErrorString = reinterpret_cast<decltype(t.GetErrorString)>(sym("ncclGetErrorString"));
	t.CommCount      = reinterpret_cast<decltype(t.CommCount)>(dlsym(t.handle, "ncclCommCount"));
	t.CommUserRank   = reinterpret_cast<decltype(t.CommUserRank)>(dlsym(t.handle, "ncclCommUserRank"));
	t.CommCuDevice   = reinterpret_cast<decltype(t.CommCuDevice)>(dlsym(t.handle, "ncclCommCuDevice"));
	r = t;
	return r;
}

}  // namespace

const char* rccl_error_string(ncclResult_t r) { return rccl().GetErrorString(r); }

namespace {

#define FI_NCCL_TRY(expr)                                                                       \
	do {                                                                                        \
		ncclResult_t r_ = (expr);                                                               \
		if (r_ != ncclSuccess) {                                                                \
			::fi::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, ::fi::rccl_error_string(r_)); \
			throw ::fi::Fail{FI_ERR_COMM};                                                      \
		}                                                                                       \
	} while (0)

}  // namespace

namespace {

#ifdef FI_TEST_TRANSPORT  // libfi_hip_test.so only: the release library carries RCCL alone
constexpr uint32_t kHostMagic = 0x46494853u;  // "FIHS"

void host_close(HostComm* h)
{
	if (!h) { return; }
	if (h->base) { (void)munmap(h->base, h->bytes); }
	if (h->fd >= 0) { (void)close(h->fd); }
	if (h->owner) { (void)shm_unlink(h->name.c_str()); }
	delete h;
}

// all ranks meet here; a rank that has given up releases the others with an error instead of a hang
void host_barrier(HostComm* h)
{
	HostShm* s = h->hdr();
	const uint32_t gen = s->generation.load(std::memory_order_acquire);
	if (s->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == s->nranks) {
		s->arrived.store(0, std::memory_order_relaxed);
		s->generation.fetch_add(1, std::memory_order_release);
		return;
	}
	const auto t0 = std::chrono::steady_clock::now();
	while (s->generation.load(std::memory_order_acquire) == gen) {
		if (s->failed.load(std::memory_order_acquire)) {
			set_error("host-staged exchange: another rank failed");
			throw Fail{FI_ERR_COMM};
		}
		if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
			s->failed.store(1, std::memory_order_release);
			set_error("host-staged exchange: no partner within 120 s");
			throw Fail{FI_ERR_COMM};
		}
		usleep(20);
	}
}

void host_allreduce(fi_ctx* c, double* dev, int count)
{
	HostComm* h = c->comm->host;
	FI_REQUIRE(count > 0 && count <= 8, FI_ERR_INVALID, "host-staged all-reduce of %d values", count);
	double* mine = reinterpret_cast<double*>(h->slot(c->rank));
	FI_HIP_TRY(hipMemcpyAsync(mine, dev, sizeof(double) * count, hipMemcpyDeviceToHost, c->stream));
	FI_HIP_TRY(hipStreamSynchronize(c->stream));
	host_barrier(h);
	double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	for (int r = 0; r < c->nranks; ++r) {  // rank order: every rank forms the same bits
		const double* v = reinterpret_cast<const double*>(h->slot(r));
		for (int k = 0; k < count; ++k) { sum[k] += v[k]; }
	}
	host_barrier(h);  // everybody has read: the slots may be rewritten
	FI_HIP_TRY(hipMemcpyAsync(dev, sum, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
	FI_HIP_TRY(hipStreamSynchronize(c->stream));
}

// a whole (small) vector: the right-hand side of a replicated coarse level (build_levels), in chunks of the slot size
void host_allreduce_vec(fi_ctx* c, void* dev, int64_t count, bool f64)
{
	HostComm* h = c->comm->host;
	const size_t es = f64 ? sizeof(double) : sizeof(float);
	const int64_t chunk = static_cast<int64_t>((h->hdr()->slot_bytes - 64) / es);
	std::vector<double> sum;
	for (int64_t o = 0; o < count; o += chunk) {
		const int64_t n = count - o < chunk ? count - o : chunk;
		char* mine = h->slot(c->rank) + 64;
		FI_HIP_TRY(hipMemcpyAsync(mine, static_cast<char*>(dev) + o * es, es * n, hipMemcpyDeviceToHost, c->stream));
		FI_HIP_TRY(hipStreamSynchronize(c->stream));
		host_barrier(h);
		sum.assign(static_cast<size_t>(n), 0.0);
		for (int r = 0; r < c->nranks; ++r) {  // rank order: every rank forms the same bits
			const char* v = h->slot(r) + 64;
			for (int64_t k = 0; k < n; ++k) {
				sum[k] += f64 ? reinterpret_cast<const double*>(v)[k] : static_cast<double>(reinterpret_cast<const float*>(v)[k]);
			}
		}
		host_barrier(h);  // everybody has read: the slots may be rewritten
		for (int64_t k = 0; k < n; ++k) {
			if (f64) { reinterpret_cast<double*>(mine)[k] = sum[k]; } else { reinterpret_cast<float*>(mine)[k] = static_cast<float>(sum[k]); }
		}
		FI_HIP_TRY(hipMemcpyAsync(static_cast<char*>(dev) + o * es, mine, es * n, hipMemcpyHostToDevice, c->stream));
		FI_HIP_TRY(hipStreamSynchronize(c->stream));
		host_barrier(h);  // (my slot is rewritten by the next chunk only after everybody is done with this one)
	}
}

void host_exchange(fi_ctx* c, void* v, hipStream_t stream, int width)
{
	HostComm* h = c->comm->host;
	const Geom&  g     = c->g;
	const int    L     = g.ndim - 1;
	const int    H     = width;   // planes exchanged: they land in the ghost planes next to the slab
	const size_t es    = elem_size(c);
	const size_t plane = static_cast<size_t>(g.stride[L]);
	const size_t bytes = es * plane * H;
	FI_REQUIRE(64 + 2 * bytes <= h->hdr()->slot_bytes, FI_ERR_COMM, "host-staged exchange: halo larger than the slot");
	char* base = static_cast<char*>(v);
	// slot layout: [64 B of sums][planes for the lower neighbour][planes for the upper neighbour]
	char* mine = h->slot(c->rank) + 64;
	if (c->rank > 0) { FI_HIP_TRY(hipMemcpyAsync(mine, base + es * plane * g.own_lo[L], bytes, hipMemcpyDeviceToHost, stream)); }
	if (c->rank + 1 < c->nranks) {
		FI_HIP_TRY(hipMemcpyAsync(mine + bytes, base + es * plane * (g.own_hi[L] - H), bytes, hipMemcpyDeviceToHost, stream));
	}
	FI_HIP_TRY(hipStreamSynchronize(stream));
	host_barrier(h);
	if (c->rank > 0) {  // the lower neighbour's planes for its upper neighbour -> my lower ghost planes
		FI_HIP_TRY(hipMemcpyAsync(base + es * plane * (g.own_lo[L] - H), h->slot(c->rank - 1) + 64 + bytes, bytes, hipMemcpyHostToDevice, stream));
	}
	if (c->rank + 1 < c->nranks) {
		FI_HIP_TRY(hipMemcpyAsync(base + es * plane * g.own_hi[L], h->slot(c->rank + 1) + 64, bytes, hipMemcpyHostToDevice, stream));
	}
	FI_HIP_TRY(hipStreamSynchronize(stream));
	host_barrier(h);
}

#else
void host_close(HostComm*) {}
#endif

}  // namespace

void comm_destroy(Comm* cm)
{
	if (!cm) { return; }
	if (cm->comm) { (void)rccl().CommDestroy(cm->comm); }
	host_close(cm->host);
	delete cm;
}

long comm_allreduces(const fi_ctx* c) { return c->comm ? c->comm->n_allreduce : 0; }

void allreduce_sum(fi_ctx* c, double* dev, int count)
{
	FI_REQUIRE(c->comm && (c->comm->comm || c->comm->host), FI_ERR_STATE, "slab context without fi_comm_init");
	++c->comm->n_allreduce;
#ifdef FI_TEST_TRANSPORT
	if (c->comm->host) {
		if (count <= 8) {
			host_allreduce(c, dev, count);
		} else {  // (the field rule's sums + maxima of up to 16 slabs: the vector form)
			host_allreduce_vec(c, dev, count, true);
		}
		return;
	}
#endif
	FI_NCCL_TRY(rccl().AllReduce(dev, dev, static_cast<size_t>(count), ncclFloat64, ncclSum, c->comm->comm, c->stream));
}

// sum of a device vector over the ranks, in place (fp32 or fp64): the right-hand side of a replicated coarse level
void allreduce_sum_vec(fi_ctx* c, void* dev, int64_t count, bool f64)
{
	FI_REQUIRE(c->comm && (c->comm->comm || c->comm->host), FI_ERR_STATE, "slab context without fi_comm_init");
	++c->comm->n_allreduce;
#ifdef FI_TEST_TRANSPORT
	if (c->comm->host) {
		host_allreduce_vec(c, dev, count, f64);
		return;
	}
#endif
	FI_NCCL_TRY(rccl().AllReduce(dev, dev, static_cast<size_t>(count), f64 ? ncclFloat64 : ncclFloat32, ncclSum, c->comm->comm, c->stream));
}

bool comm_ready(const fi_ctx* c) { return c->comm && (c->comm->comm || c->comm->host); }

void exchange_halo(fi_ctx* c, void* v, int width) { exchange_halo_on(c, v, c->stream, width); }

void exchange_halo_on(fi_ctx* c, void* v, hipStream_t stream, int width)
{
	if (c->nranks <= 1) { return; }
	FI_REQUIRE(c->comm && (c->comm->comm || c->comm->host), FI_ERR_STATE, "slab context without fi_comm_init");
	if (width <= 0) { width = c->reach; }
	FI_REQUIRE(width <= c->halo, FI_ERR_INVALID, "exchange of %d planes into %d ghost planes", width, c->halo);
#ifdef FI_TEST_TRANSPORT
	if (c->comm->host) {
		host_exchange(c, v, stream, width);
		return;
	}
#endif
	const Geom&  g     = c->g;
	const int    L     = g.ndim - 1;
	const int    H     = width;  // planes exchanged: they land in the ghost planes next to the slab
	const size_t es    = elem_size(c);
	const size_t plane = static_cast<size_t>(g.stride[L]);
	const size_t count = plane * H;
	const ncclDataType_t dt = c->dtype == FI_F64 ? ncclFloat64 : ncclFloat32;
	char* base = static_cast<char*>(v);
	char* lower_ghost = base + es * plane * (g.own_lo[L] - H);
	char* first_owned = base + es * plane * g.own_lo[L];
	char* last_owned  = base + es * plane * (g.own_hi[L] - H);
	char* upper_ghost = base + es * plane * g.own_hi[L];
	Rccl& r = rccl();
	FI_NCCL_TRY(r.GroupStart());
	if (c->rank > 0) {
		FI_NCCL_TRY(r.Send(first_owned, count, dt, c->rank - 1, c->comm->comm, stream));
		FI_NCCL_TRY(r.Recv(lower_ghost, count, dt, c->rank - 1, c->comm->comm, stream));
	}
	if (c->rank + 1 < c->nranks) {
		FI_NCCL_TRY(r.Send(last_owned, count, dt, c->rank + 1, c->comm->comm, stream));
		FI_NCCL_TRY(r.Recv(upper_ghost, count, dt, c->rank + 1, c->comm->comm, stream));
	}
	FI_NCCL_TRY(r.GroupEnd());
}

}  // namespace fi

extern "C" {

int fi_comm_unique_id(void* out128)
{
	try {
		FI_REQUIRE(out128 != nullptr, FI_ERR_INVALID, "null output");
		static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
		ncclUniqueId id;
		FI_NCCL_TRY(fi::rccl().GetUniqueId(&id));
		memcpy(out128, &id, sizeof(id));
	} catch (const fi::Fail& f) {
		return f.code;
	}
	return FI_OK;
}

int fi_comm_self_test(int device, long count)
{
	// The exact RCCL call pattern of exchange_halo / allreduce_sum on a one-rank communicator: a grouped
	// ncclSend + ncclRecv (to and from rank 0 = self) of `count` floats on a stream, then an in-place fp64 all-reduce.
	try {
		FI_REQUIRE(count > 0, FI_ERR_INVALID, "count must be positive");
		FI_HIP_TRY(hipSetDevice(device));
		fi::Rccl& r = fi::rccl();
		ncclUniqueId id;
		FI_NCCL_TRY(r.GetUniqueId(&id));
		struct Guard {  // the communicator and the stream go away on every path out of here
			fi::Rccl&   r;
			ncclComm_t  comm = nullptr;
			hipStream_t st = nullptr;
			~Guard()
			{
				if (comm) { (void)r.CommDestroy(comm); }
				if (st) { (void)hipStreamDestroy(st); }
			}
		} guard{r};
		ncclComm_t& comm = guard.comm;
		FI_NCCL_TRY(r.CommInitRank(&comm, 1, id, 0));
		hipStream_t& st = guard.st;
		FI_HIP_TRY(hipStreamCreate(&st));
		fi::DevBuf src, dst, sums;
		src.alloc(sizeof(float) * count);
		dst.alloc(sizeof(float) * count);
		sums.alloc(sizeof(double) * 4);
		std::vector<float> h(count), back(count);
		for (long i = 0; i < count; ++i) { h[i] = static_cast<float>((i * 2654435761u) % 1000003u) * 1e-3f; }
		const double hs[4] = {1.5, -2.25, 3.0e10, 7.0e-10};
		double       hb[4] = {0, 0, 0, 0};
		FI_HIP_TRY(hipMemcpyAsync(src.p, h.data(), sizeof(float) * count, hipMemcpyHostToDevice, st));
		FI_HIP_TRY(hipMemsetAsync(dst.p, 0, sizeof(float) * count, st));
		FI_HIP_TRY(hipMemcpyAsync(sums.p, hs, sizeof(hs), hipMemcpyHostToDevice, st));
		FI_NCCL_TRY(r.GroupStart());
		FI_NCCL_TRY(r.Send(src.p, static_cast<size_t>(count), ncclFloat32, 0, comm, st));
		FI_NCCL_TRY(r.Recv(dst.p, static_cast<size_t>(count), ncclFloat32, 0, comm, st));
		FI_NCCL_TRY(r.GroupEnd());
		FI_NCCL_TRY(r.AllReduce(sums.p, sums.p, 4, ncclFloat64, ncclSum, comm, st));
		FI_HIP_TRY(hipMemcpyAsync(back.data(), dst.p, sizeof(float) * count, hipMemcpyDeviceToHost, st));
		FI_HIP_TRY(hipMemcpyAsync(hb, sums.p, sizeof(hb), hipMemcpyDeviceToHost, st));
		FI_HIP_TRY(hipStreamSynchronize(st));
		for (long i = 0; i < count; ++i) {
			FI_REQUIRE(back[i] == h[i], FI_ERR_COMM, "self send/recv: element %ld differs", i);
		}
		for (int k = 0; k < 4; ++k) { FI_REQUIRE(hb[k] == hs[k], FI_ERR_COMM, "one-rank all-reduce changed element %d", k); }
	} catch (const fi::Fail& f) {
		return f.code;
	}
	return FI_OK;
}

int fi_comm_init(fi_ctx* c, const void* unique_id128)
{
	try {
		FI_REQUIRE(c != nullptr && unique_id128 != nullptr, FI_ERR_INVALID, "null argument");
		FI_REQUIRE(c->nranks > 1, FI_ERR_STATE, "fi_comm_init on a single-rank context");
		FI_HIP_TRY(hipSetDevice(c->device));
		ncclUniqueId id;
		memcpy(&id, unique_id128, sizeof(id));
		if (!c->comm) { c->comm = new fi::Comm(); }
		if (c->comm->comm) {  // a second call replaces the communicator
			(void)fi::rccl().CommDestroy(c->comm->comm);
			c->comm->comm = nullptr;
		}
		FI_NCCL_TRY(fi::rccl().CommInitRank(&c->comm->comm, c->nranks, id, c->rank));
	} catch (const fi::Fail& f) {
		return f.code;
	}
	return FI_OK;
}

int fi_comm_info(const fi_ctx* c, long out[7])
{
	try {
		FI_REQUIRE(c != nullptr && out != nullptr, FI_ERR_INVALID, "null argument");
		for (int i = 0; i < 7; ++i) { out[i] = 0; }
		out[4] = c->nranks > 1 ? c->reach : 0;
		out[5] = c->nranks > 1 ? c->halo : 0;
		int64_t plane = 1;
		for (int d = 0; d + 1 < c->g.ndim; ++d) { plane *= c->g.gn[d]; }
		out[6] = static_cast<long>(plane * (c->dtype == FI_F64 ? 8 : 4));
		if (!c->comm) { return FI_OK; }
		if (c->comm->comm) {
			fi::Rccl& r = fi::rccl();
			int v = -1;
			out[0] = (r.CommCount && r.CommCount(c->comm->comm, &v) == ncclSuccess) ? v : -1;
			out[1] = (r.CommUserRank && r.CommUserRank(c->comm->comm, &v) == ncclSuccess) ? v : -1;
			out[2] = (r.CommCuDevice && r.CommCuDevice(c->comm->comm, &v) == ncclSuccess) ? v : -1;
			out[3] = 1;
		} else if (c->comm->host) {
			out[0] = c->nranks;
			out[1] = c->rank;
			out[2] = c->device;
			out[3] = 2;
		}
	} catch (const fi::Fail& f) {
		return f.code;
	}
	return FI_OK;
}

int fi_comm_init_host(fi_ctx* c, const char* name, int create)
{
#ifndef FI_TEST_TRANSPORT
	(void)c; (void)name; (void)create;
	fi::set_error("fi_comm_init_host: the host-staged TEST transport is built into libfi_hip_test.so only (-DFI_TEST_TRANSPORT)");
	return FI_ERR_UNSUPPORTED;
#else
	fi::HostComm* h = nullptr;
	try {
		FI_REQUIRE(c != nullptr && name != nullptr && name[0] == '/', FI_ERR_INVALID, "bad argument (the name starts with '/')");
		FI_REQUIRE(c->nranks > 1, FI_ERR_STATE, "fi_comm_init_host on a single-rank context");
		// slot: 64 bytes of sums + the ghost planes for both neighbours at the finest level in fp64 with the widest exchange
		// (8 planes: model_4 needs 4, the polynomial's deep exchange 2 (d - 1); the model may be set after the
		// communicator) -- coarser levels and fp32 replicas are smaller
		const fi::Geom& g = c->g;
		const size_t plane = static_cast<size_t>(g.stride[g.ndim - 1]);
		const size_t slot = ((64 + 2 * plane * 8 * sizeof(double)) + 4095) / 4096 * 4096;
		h = new fi::HostComm();
		h->name  = name;
		h->bytes = 4096 + slot * static_cast<size_t>(c->nranks);
		h->owner = create != 0;
		h->fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
		FI_REQUIRE(h->fd >= 0, FI_ERR_COMM, "shm_open(%s) failed", name);
		if (create) { FI_REQUIRE(ftruncate(h->fd, static_cast<off_t>(h->bytes)) == 0, FI_ERR_COMM, "ftruncate(%s) failed", name); }
		void* m = mmap(nullptr, h->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, h->fd, 0);
		FI_REQUIRE(m != MAP_FAILED, FI_ERR_COMM, "mmap(%s) failed", name);
		h->base = static_cast<char*>(m);
		fi::HostShm* s = h->hdr();
		if (create) {
			s->nranks     = static_cast<uint32_t>(c->nranks);
			s->slot_bytes = slot;
			s->arrived.store(0);
			s->generation.store(0);
			s->failed.store(0);
			std::atomic_thread_fence(std::memory_order_release);
			s->magic = fi::kHostMagic;
		} else {
			FI_REQUIRE(s->magic == fi::kHostMagic && s->nranks == static_cast<uint32_t>(c->nranks) && s->slot_bytes == slot,
			           FI_ERR_COMM, "host segment %s does not match this decomposition", name);
		}
		if (!c->comm) { c->comm = new fi::Comm(); }
		fi::host_close(c->comm->host);
		c->comm->host = h;
	} catch (const fi::Fail& f) {
		fi::host_close(h);
		return f.code;
	}
	return FI_OK;
#endif
}

}  // extern "C"
