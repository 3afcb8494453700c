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
#include <cstring>
#include <vector>
#include <rccl/rccl.h>

#include "fi_internal.h"

namespace fi {

struct Comm {
	ncclComm_t comm = nullptr;
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
};

Rccl& rccl()
{
	static Rccl r;
	if (r.handle) { return r; }
	const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
	for (const char* n : names) {
		r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
		if (r.handle) { break; }
	}
	FI_REQUIRE(r.handle != nullptr, FI_ERR_COMM, "cannot load librccl: %s", dlerror());
	auto sym = [&](const char* name) {
		void* p = dlsym(r.handle, name);
		FI_REQUIRE(p != nullptr, FI_ERR_COMM, "librccl lacks %s", name);
		return p;
	};
	r.GetUniqueId    = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
	r.CommInitRank   = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
	r.CommDestroy    = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
	r.AllReduce      = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
	r.Send           = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
	r.Recv           = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
	r.GroupStart     = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
	r.GroupEnd       = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
	r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
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

void comm_destroy(Comm* cm)
{
	if (!cm) { return; }
	if (cm->comm) { (void)rccl().CommDestroy(cm->comm); }
	delete cm;
}

void allreduce_sum(fi_ctx* c, double* dev, int count)
{
	FI_REQUIRE(c->comm && c->comm->comm, FI_ERR_STATE, "slab context without fi_comm_init");
	FI_NCCL_TRY(rccl().AllReduce(dev, dev, static_cast<size_t>(count), ncclFloat64, ncclSum, c->comm->comm, c->stream));
}

void exchange_halo(fi_ctx* c, void* v)
{
	if (c->nranks <= 1) { return; }
	FI_REQUIRE(c->comm && c->comm->comm, FI_ERR_STATE, "slab context without fi_comm_init");
	const Geom&  g     = c->g;
	const int    L     = g.ndim - 1;
	const int    H     = c->halo;
	const size_t es    = elem_size(c);
	const size_t plane = static_cast<size_t>(g.stride[L]);
	const size_t count = plane * H;
	const ncclDataType_t dt = c->dtype == FI_F64 ? ncclFloat64 : ncclFloat32;
	char* base = static_cast<char*>(v);
	char* lower_ghost = base;
	char* first_owned = base + es * plane * g.own_lo[L];
	char* last_owned  = base + es * plane * (g.own_hi[L] - H);
	char* upper_ghost = base + es * plane * g.own_hi[L];
	Rccl& r = rccl();
	FI_NCCL_TRY(r.GroupStart());
	if (c->rank > 0) {
		FI_NCCL_TRY(r.Send(first_owned, count, dt, c->rank - 1, c->comm->comm, c->stream));
		FI_NCCL_TRY(r.Recv(lower_ghost, count, dt, c->rank - 1, c->comm->comm, c->stream));
	}
	if (c->rank + 1 < c->nranks) {
		FI_NCCL_TRY(r.Send(last_owned, count, dt, c->rank + 1, c->comm->comm, c->stream));
		FI_NCCL_TRY(r.Recv(upper_ghost, count, dt, c->rank + 1, c->comm->comm, c->stream));
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
		ncclComm_t comm = nullptr;
		FI_NCCL_TRY(r.CommInitRank(&comm, 1, id, 0));
		hipStream_t st = nullptr;
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
		(void)r.CommDestroy(comm);
		(void)hipStreamDestroy(st);
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
		FI_NCCL_TRY(fi::rccl().CommInitRank(&c->comm->comm, c->nranks, id, c->rank));
	} catch (const fi::Fail& f) {
		return f.code;
	}
	return FI_OK;
}

}  // extern "C"
