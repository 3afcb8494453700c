// fi_capi.hip -- the C ABI of include/fi_hip.h (every entry point cites the reference interface it replaces there): contexts,
// model, points, assemble, the solver entry points and their options, statistics.  Nothing throws or aborts across it.
#include "fi_solver_internal.h"
#include "fi_workers.h"

namespace fi {

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_error;

void set_error(const char* fmt, ...)
{
	char    buf[1024];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_error = buf;
}


// upscale_field (field_interpolation.cpp:431-485): one thread per point of the large lattice.
namespace {
struct UpscaleArgs {
	int ndim;
	int ssz[3], lsz[3];
};
__global__ __launch_bounds__(kThreads) void k_upscale(UpscaleArgs a, int64_t nlarge, const float* __restrict__ small,
                                                       float* __restrict__ out)
{
	const int64_t li = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (li >= nlarge) { return; }
	int     base[3];
	float   t[3];
	int64_t rest = li;
	for (int d = 0; d < a.ndim; ++d) {
		const int c = static_cast<int>(rest % a.lsz[d]);
		rest /= a.lsz[d];
		const float sp = static_cast<float>(c) * (static_cast<float>(a.ssz[d]) - 1.0f) /
		                 (static_cast<float>(a.lsz[d]) - 1.0f);
		const float fl = floorf(sp);
		base[d] = static_cast<int>(fl);
		t[d]    = sp - static_cast<float>(base[d]);
	}
	float wsum = 0.0f, fsum = 0.0f;
	for (int q = 0; q < (1 << a.ndim); ++q) {
		int64_t idx = 0, stride = 1;
		float   w  = 1.0f;
		bool    in = true;
		for (int d = 0; d < a.ndim; ++d) {
			const int up = (q >> d) & 1;
			const int cc = base[d] + up;
			idx += stride * cc;
			stride *= a.ssz[d];
			w *= up ? t[d] : 1.0f - t[d];
			in = in && (0 <= cc) && (cc < a.ssz[d]);
		}
		if (in) {
			wsum += w;
			fsum += w * small[idx];
		}
	}
	out[li] = (wsum == 0.0f) ? 0.0f : fsum / wsum;
}


}  // namespace

void check_ctx(const fi_ctx* c) { FI_REQUIRE(c != nullptr, FI_ERR_INVALID, "null context"); }

void check_assembled(const fi_ctx* c)
{
	check_ctx(c);
	FI_REQUIRE(c->assembled, FI_ERR_STATE, "fi_assemble has not been called");
}

void bind_device(const fi_ctx* c) { FI_HIP_TRY(hipSetDevice(c->device)); }

fi_ctx* create_ctx(int ndim, const int* sizes, int dtype, int rank, int nranks)
{
	FI_REQUIRE(1 <= ndim && ndim <= FI_MAX_DIM, FI_ERR_INVALID, "ndim must be 1..%d (got %d)", FI_MAX_DIM, ndim);
	FI_REQUIRE(sizes != nullptr, FI_ERR_INVALID, "sizes is null");
	FI_REQUIRE(dtype == FI_F32 || dtype == FI_F64, FI_ERR_INVALID, "unknown dtype %d", dtype);
	FI_REQUIRE(nranks >= 1 && 0 <= rank && rank < nranks, FI_ERR_INVALID, "bad rank %d of %d", rank, nranks);
	int64_t n = 1;
	for (int d = 0; d < ndim; ++d) {
		FI_REQUIRE(sizes[d] >= 1, FI_ERR_INVALID, "sizes[%d] = %d", d, sizes[d]);
		n *= sizes[d];
		FI_REQUIRE(static_cast<int64_t>(sizes[d]) + 1 < (1 << 20), FI_ERR_INVALID, "sizes[%d] too large", d);
	}
	FI_REQUIRE(n < (1LL << 31), FI_ERR_UNSUPPORTED, "lattice has %lld unknowns; the reference indexes with int",
	           static_cast<long long>(n));
	FI_REQUIRE(nranks == 1 || sizes[ndim - 1] >= nranks, FI_ERR_INVALID, "fewer planes (%d) than ranks (%d)",
	           sizes[ndim - 1], nranks);
	auto* c = new fi_ctx();
	try {
		c->dtype  = dtype;
		c->rank   = rank;
		c->nranks = nranks;
		FI_HIP_TRY(hipGetDevice(&c->device));
		c->stream = stream_take();
		c->halo = 1;
		compute_geom(c, ndim, sizes);
		c->scal.alloc(3 * sizeof(CgScalars));  // [0]: the state every kernel and the host look at, [1]: mid-iteration copy,
		                                       // [2]: landing place of the dot products summed over slabs (rank sets)
		FI_HIP_TRY(hipMemset(c->scal.p, 0, 2 * sizeof(CgScalars)));
		c->scal_host = static_cast<CgScalars*>(pinned_take(sizeof(CgScalars), &c->scal_host_cap));
		// default Weights (field_interpolation.hpp:75-95)
		c->w = fi_weights{1.0f, 1.0f, 0.0f, 0.0f, 0.5f, 0.0f, 0.0f, 0.0f, FI_VALUE_LINEAR_INTERPOLATION,
		                  FI_GRADIENT_CELL_EDGES};
	} catch (...) {
		fi_ctx_destroy(c);
		throw;
	}
	return c;
}


}  // namespace fi

// ===================================================================================================
// C ABI

extern "C" {

const char* fi_last_error(void) { return fi::g_error.c_str(); }

int fi_device_count(int* count)
{
	FI_API_BEGIN
	FI_REQUIRE(count != nullptr, FI_ERR_INVALID, "count is null");
	int n = 0;
	const hipError_t e = hipGetDeviceCount(&n);
	*count = (e == hipSuccess) ? n : 0;
	FI_API_END
}

int fi_ctx_create(fi_ctx** out, int ndim, const int* sizes, int dtype)
{
	FI_API_BEGIN
	FI_REQUIRE(out != nullptr, FI_ERR_INVALID, "out is null");
	*out = fi::create_ctx(ndim, sizes, dtype, 0, 1);
	FI_API_END
}

int fi_ctx_create_slab(fi_ctx** out, int ndim, const int* sizes, int dtype, int rank, int nranks)
{
	FI_API_BEGIN
	FI_REQUIRE(out != nullptr, FI_ERR_INVALID, "out is null");
	*out = fi::create_ctx(ndim, sizes, dtype, rank, nranks);
	FI_API_END
}

int fi_ctx_destroy(fi_ctx* c)
{
	if (!c) { return FI_OK; }
	(void)hipSetDevice(c->device);
	// The context's blocks go to the pool of fi_pool.hip: nothing in flight may still touch them.  Everything that works on
	// a context's blocks is enqueued on one of ITS streams (the solver stream -- a group's members share member 0's --, the
	// helper streams of the assembly, the communication stream): those are drained, level by level as the recursion below
	// reaches them; the rest of the device (other contexts, torch, other threads) is not stalled.
	bool drained = true;
	for (hipStream_t st : {c->stream, c->level_stream, c->level_stream2, c->build_stream, c->comm_stream}) {
		if (st && hipStreamSynchronize(st) != hipSuccess) { drained = false; }
	}
	struct Quiescent {
		bool was;
		explicit Quiescent(bool ok) : was(fi::pool_quiescent) { fi::pool_quiescent = ok; }
		~Quiescent() { fi::pool_quiescent = was; }
	} quiescent(drained);
	for (auto* pb : c->pending) { delete pb; }
	for (auto* pb : c->pending_pool) { delete pb; }
	for (auto* b : c->batches) { delete b; }
	for (auto* b : c->batches_pool) { delete b; }
	if (c->coarse) { fi_ctx_destroy(c->coarse); }
	if (c->twin) { fi_ctx_destroy(c->twin); }
	c->pending.clear();
	c->pending_pool.clear();
	for (auto e : c->ev) { (void)hipEventDestroy(e); }
	for (auto e : c->ev_prec) { (void)hipEventDestroy(e); }
	fi::stream_give(c->level_stream, drained);
	if (c->ev_level) { (void)hipEventDestroy(c->ev_level); }
	fi::stream_give(c->level_stream2, drained);
	if (c->build_stream) {
		fi::stream_give(c->build_stream, drained);
		(void)hipEventDestroy(c->ev_build);
	}
	if (c->ev_level2) { (void)hipEventDestroy(c->ev_level2); }
	if (c->comm_stream) {
		fi::stream_give(c->comm_stream, drained);
		(void)hipEventDestroy(c->ev_ready);
		(void)hipEventDestroy(c->ev_halo);
	}
	if (c->comm && c->owns_comm) { fi::comm_destroy(c->comm); }
	// pinned blocks go back to the pool only when nothing can still be copying into or out of them (the stop-flag copies of an
	// unwatched level, point staging): a context whose streams did not drain frees them instead -- hipHostFree waits
	if (c->scal_host) {
		if (drained) { fi::pinned_give(c->scal_host, c->scal_host_cap); } else { (void)hipHostFree(c->scal_host); }
	}
	for (int s = 0; s < 3; ++s) {
		if (!c->pin[s]) { continue; }
		if (drained) { fi::pinned_give(c->pin[s], c->pin_bytes[s]); } else { (void)hipHostFree(c->pin[s]); }
	}
	if (c->ev_unwatched) { (void)hipEventDestroy(c->ev_unwatched); }
	if (c->ev_asm0) {
		(void)hipEventDestroy(c->ev_asm0);
		(void)hipEventDestroy(c->ev_asm1);
	}
	if (c->stream && c->owns_stream) { fi::stream_give(c->stream, drained); }
	delete c;
	return FI_OK;
}

int fi_memory_pool(long long keep_bytes, long long* cached_bytes)
{
	FI_API_BEGIN
	const size_t keep = keep_bytes < 0 ? ~size_t(0) : static_cast<size_t>(keep_bytes);
	const size_t left = fi::pool_trim(keep);
	if (keep == 0) {  // "nothing left over from earlier contexts": neither their blocks nor what they learnt
		fi::forget_iterations();
		fi::forget_lambdas();
	}
	if (cached_bytes) { *cached_bytes = static_cast<long long>(left); }
	FI_API_END
}

int fi_slab_partition(int planes, int rank, int nranks, int* lo, int* hi)
{
	FI_API_BEGIN
	FI_REQUIRE(nranks >= 1 && 0 <= rank && rank < nranks && planes >= 0, FI_ERR_INVALID, "bad slab request");
	if (lo) { *lo = static_cast<int>(static_cast<int64_t>(rank) * planes / nranks); }
	if (hi) { *hi = static_cast<int>(static_cast<int64_t>(rank + 1) * planes / nranks); }
	FI_API_END
}

int fi_halo_width(const fi_weights* w, int* width)
{
	FI_API_BEGIN
	FI_REQUIRE(w && width, FI_ERR_INVALID, "null argument");
	const int reach = fi::model_reach(*w);
	*width = reach > 1 ? reach : 1;
	FI_API_END
}

int fi_slab_range(const fi_ctx* c, int* lo, int* hi)
{
	FI_API_BEGIN
	fi::check_ctx(c);
	if (lo) { *lo = c->slab_lo; }
	if (hi) { *hi = c->slab_hi; }
	FI_API_END
}

int fi_slab_point_range(const fi_ctx* c, float* lo, float* hi)
{
	FI_API_BEGIN
	fi::check_ctx(c);
	// Level l halves the lattice l times (coarse plane k sits on fine plane k * 2^l; this rank keeps coarse planes
	// ceil(slab_lo / 2^l) .. ceil(slab_hi / 2^l) - 1), and a rank needs every cell that touches an owned plane of
	// that level plus one cell of margin for the nearest-neighbour kernels: 2 cells of 2^l fine planes below, 1 above --
	// 2 above as well, because a level halved cell-centred (fi_ctx::cc) sees a point up to half a coarse cell further down
	// (position / 2^l - (1 - 2^-l) / 2).
	// A hierarchy whose deeper levels are replicated (whole lattices on every rank once the slabs would be thinner than 4
	// planes: build_levels) is assembled from ALL the points: the range is then everything.
	const int L = c->levels_wanted > 0 ? c->levels_wanted : 0;
	int first_tail = 0;
	const int nlevels = fi::plan_levels(c, &first_tail);
	if (c->nranks > 1 && first_tail <= nlevels) {
		if (lo) { *lo = -std::numeric_limits<float>::max(); }
		if (hi) { *hi = std::numeric_limits<float>::max(); }
		return FI_OK;
	}
	const float cell = static_cast<float>(1 << (L < 20 ? L : 20));
	if (lo) { *lo = static_cast<float>(c->slab_lo) - 2.0f * cell; }
	if (hi) { *hi = static_cast<float>(c->slab_hi) + (L > 0 ? 2.0f : 1.0f) * cell; }
	FI_API_END
}

int fi_set_model(fi_ctx* c, const fi_weights* w)
{
	FI_API_BEGIN
	fi::check_ctx(c);
	FI_REQUIRE(w != nullptr, FI_ERR_INVALID, "weights is null");
	c->w         = *w;
	c->model_set = true;
	c->assembled = false;
	for (fi_ctx* l = c; l; l = l->coarse) { l->poly_lambda = 0; }  // the polynomial preconditioner's bound belongs to the model
	if (c->twin) {
		for (fi_ctx* l = c->twin; l; l = l->coarse) { l->poly_lambda = 0; }
	}
	FI_API_END
}

int fi_add_points(fi_ctx* c, long n, const float* positions, const float* normals, const float* point_weights,
                  const float* values, float value_weight, int value_kernel, float gradient_weight, int gradient_kernel,
                  int memory)
{
	FI_API_BEGIN
	fi::check_ctx(c);
	fi::bind_device(c);
	FI_REQUIRE(n >= 0, FI_ERR_INVALID, "negative point count");
	if (n == 0) { return FI_OK; }
	FI_REQUIRE(positions != nullptr, FI_ERR_INVALID, "positions is null");  // CHECK_NOTNULL_F, cpp:382
	FI_REQUIRE(value_kernel == FI_VALUE_NEAREST_NEIGHBOR || value_kernel == FI_VALUE_LINEAR_INTERPOLATION,
	           FI_ERR_INVALID, "Unknown value kernel: %d", value_kernel);
	FI_REQUIRE(!(value_kernel == FI_VALUE_NEAREST_NEIGHBOR && normals == nullptr), FI_ERR_INVALID,
	           "nearest-neighbour value kernel needs normals (field_interpolation.cpp:361)");
	if (normals) {
		FI_REQUIRE(gradient_kernel >= 0 && gradient_kernel <= 2, FI_ERR_INVALID, "Unknown gradient kernel: %d",
		           gradient_kernel);  // ABORT_F, cpp:238
// (GradientKernel::kLinearInterpolation over slabs: its rows are kept as triplets with local columns, fi_generic.hip)
	}
	FI_REQUIRE(memory == FI_HOST || memory == FI_DEVICE, FI_ERR_INVALID, "bad memory kind %d", memory);
	const int D = c->g.ndim;
	fi::DevBuf dpos, dnrm, dpw, dval;
	const float *p = positions, *g = normals, *w = point_weights, *v = values;
	if (memory == FI_HOST) {
		auto up = [&](fi::DevBuf& b, const float* src, size_t count) -> const float* {
			if (!src) { return nullptr; }
			b.alloc(sizeof(float) * count);
			FI_HIP_TRY(hipMemcpyAsync(b.p, src, sizeof(float) * count, hipMemcpyHostToDevice, c->stream));
			return b.as<float>();
		};
		p = up(dpos, positions, static_cast<size_t>(n) * D);
		g = up(dnrm, normals, static_cast<size_t>(n) * D);
		w = up(dpw, point_weights, static_cast<size_t>(n));
		v = up(dval, values, static_cast<size_t>(n));
	}
	fi::add_points_device(c, n, p, g, w, v, value_weight, value_kernel, gradient_weight, gradient_kernel);
	c->assembled = false;
	FI_API_END
}

}  // extern "C"

namespace fi {
// positions / normals / weights / values already on the device
void add_points_device(fi_ctx* c, long n, const float* p, const float* g, const float* w, const float* v, float value_weight,
                       int value_kernel, float gradient_weight, int gradient_kernel)
{
	const int D = c->g.ndim;
	{   // keep the points on the device: coarser levels of a multilevel solve are assembled from them
		fi::PointBatch* b = nullptr;
		if (!c->batches_pool.empty()) {
			b = c->batches_pool.back();
			c->batches_pool.pop_back();
		} else {
			b = new fi::PointBatch();
		}
		c->batches.push_back(b);
		auto keep = [&](fi::DevBuf& dst, const float* src, size_t count) {
			if (!src) { return false; }
			dst.alloc(sizeof(float) * count);
			FI_HIP_TRY(hipMemcpyAsync(dst.p, src, sizeof(float) * count, hipMemcpyDeviceToDevice, c->stream));
			return true;
		};
		b->n = n;
		b->prior = false;
		keep(b->pos, p, static_cast<size_t>(n) * D);
		b->has_nrm = keep(b->nrm, g, static_cast<size_t>(n) * D);
		b->has_pw  = keep(b->pw, w, static_cast<size_t>(n));
		b->has_val = keep(b->val, v, static_cast<size_t>(n));
		b->vw = value_weight;
		b->gw = gradient_weight;
		b->vk = value_kernel;
		b->gk = gradient_kernel;
	}
	const bool lin = g && gradient_kernel == FI_GRADIENT_LINEAR_INTERPOLATION;
	// cell-local rows (value rows; gradient rows of the nearest-neighbour / cell-edge kernels) ...
	fi::emit_point_rows(c, n, p, g, w, v, value_weight, value_kernel, lin ? 0.0f : gradient_weight,
	                    lin ? FI_GRADIENT_CELL_EDGES : gradient_kernel);
	// ... and the 3-point-wide rows of GradientKernel::kLinearInterpolation as generic sparse rows
	if (lin && gradient_weight != 0.0f) { fi::generic_add_gradient_linear(c, n, p, g, w, gradient_weight); }
	FI_HIP_TRY(hipStreamSynchronize(c->stream));
}
}  // namespace fi

extern "C" {

int fi_add_border_prior(fi_ctx* c, float weight)
{
	FI_API_BEGIN
	fi::check_ctx(c);
	fi::bind_device(c);
	if (weight == 0.0f) { return FI_OK; }  // add_equation skips zero weights (sparse_linear.cpp:36)
	// The distance is to the nearest point of the WHOLE cloud (sdf_field.cpp:218-246); a slab context holds only the points
	// of fi_slab_point_range, so the prior of a decomposed lattice would be silently wrong (or infinite on a rank without
	// points): not supported -- add the prior's rows with fi_add_points(FI_VALUE_NEAREST_NEIGHBOR) from the caller's side.
	FI_REQUIRE(c->nranks == 1, FI_ERR_UNSUPPORTED, "fi_add_border_prior on a slab context: a rank sees only its own points");
	bool any = false;
	for (const fi::PointBatch* b : c->batches) { any = any || (b->n > 0 && !b->prior); }
	FI_REQUIRE(any, FI_ERR_STATE, "fi_add_border_prior needs the data points: call it after fi_add_points");
	fi::DevBuf pos, val, zero;
	const int64_t nb = fi::border_prior_points(c, pos, val);
	if (nb > 0) {
		zero.alloc(sizeof(float) * nb * c->g.ndim);
		FI_HIP_TRY(hipMemsetAsync(zero.p, 0, sizeof(float) * nb * c->g.ndim, c->stream));
		// the row [1] * w, rhs d * w at the lattice point itself: a nearest-neighbour value constraint with a zero gradient
		fi::add_points_device(c, static_cast<long>(nb), pos.as<float>(), zero.as<float>(), nullptr, val.as<float>(), weight,
		                      FI_VALUE_NEAREST_NEIGHBOR, 0.0f, FI_GRADIENT_NEAREST_NEIGHBOR);
		c->batches.back()->prior = true;
	}
	c->assembled = false;
	FI_API_END
}

int fi_add_rows_coo(fi_ctx* c, long nrows, long ntriplets, const fi_triplet* triplets, const float* rhs, int memory)
{
	FI_API_BEGIN
	fi::check_ctx(c);
	fi::bind_device(c);
	FI_REQUIRE(nrows >= 0 && ntriplets >= 0, FI_ERR_INVALID, "negative count");
	FI_REQUIRE(memory == FI_HOST || memory == FI_DEVICE, FI_ERR_INVALID, "bad memory kind %d", memory);
	FI_REQUIRE(c->nranks == 1, FI_ERR_UNSUPPORTED, "generic rows need an undivided lattice");
	FI_REQUIRE((ntriplets == 0 || triplets) && (nrows == 0 || rhs), FI_ERR_INVALID, "null buffer");
	fi::generic_add_coo(c, nrows, ntriplets, triplets, rhs, memory);
	c->assembled = false;
	FI_API_END
}

int fi_clear_points(fi_ctx* c)
{
	FI_API_BEGIN
	fi::check_ctx(c);
	for (auto* pb : c->pending) { c->pending_pool.push_back(pb); }  // keep the HBM buffers for the next batch
	c->pending.clear();
	for (auto* b : c->batches) { c->batches_pool.push_back(b); }
	c->batches.clear();
	fi::generic_clear(c);
	c->assembled = false;
	FI_API_END
}

namespace fi {
// the time of the last fi_assemble, once its second event has passed (waits for it: the events are on the context's stream)
void finish_assemble_timing(fi_ctx* c)
{
	if (!c->asm_time_pending) { return; }
	c->asm_time_pending = false;
	if (hipEventSynchronize(c->ev_asm1) != hipSuccess) { return; }
	float ms = 0;
	if (hipEventElapsedTime(&ms, c->ev_asm0, c->ev_asm1) == hipSuccess) { c->stats.assemble_ms = ms; }
}
}  // namespace fi

int fi_assemble(fi_ctx* c)
{
	FI_API_BEGIN
	fi::check_ctx(c);
	fi::bind_device(c);
	// The assembly is timed between two events of the context and nobody waits for the second one here: the caller's next
	// call -- normally the solve -- queues behind the assembly's last kernels instead of finding an idle GPU after a host
	// round trip (~0.1 ms of a config-4 step).  fi_get_stats (and the next fi_assemble) read the time.
	fi::finish_assemble_timing(c);
	if (!c->ev_asm0) {
		FI_HIP_TRY(hipEventCreate(&c->ev_asm0));
		FI_HIP_TRY(hipEventCreate(&c->ev_asm1));
	}
	const hipEvent_t e0 = c->ev_asm0, e1 = c->ev_asm1;
	fi::chain_mark(nullptr);
	FI_HIP_TRY(hipEventRecord(e0, c->stream));
	// Ghost planes along the decomposed axis.  reach: the widest model stencil, at least the cell reach (1) -- the width of
	// an exchange.  halo (planes stored): the reach, or the polynomial preconditioner's DEEP exchange: 2 (d - 1) planes of
	// r travel once per polynomial and the steps run redundantly on the shrinking ghost zone (cg_run_poly) instead of one
	// exchange per step -- 3-D lattices, 3 to 5 terms set before the assemble, slabs at least that thick on every rank.
	// Data facts every rank must see alike (fi_ctx::any_trip): one all-reduce in front of everything they decide
	{
		bool trip = c->generic.ntrip != 0, grad = !fi::holds_value_rows_only(c);
		if (c->facts_forced) {  // a loop-back group has looked at all its members
			trip = c->forced_trip;
			grad = c->forced_grad;
		} else if (c->nranks > 1 && fi::comm_ready(c)) {
			double* slot = (c->scal.as<fi::CgScalars>() + 2)->sums;
			double  v[2] = {trip ? 1.0 : 0.0, grad ? 1.0 : 0.0};
			FI_HIP_TRY(hipMemcpyAsync(slot, v, sizeof(v), hipMemcpyHostToDevice, c->stream));
			fi::allreduce_sum(c, slot, 2);
			FI_HIP_TRY(hipMemcpyAsync(v, slot, sizeof(v), hipMemcpyDeviceToHost, c->stream));
			FI_HIP_TRY(hipStreamSynchronize(c->stream));
			trip = v[0] > 0.0;
			grad = v[1] > 0.0;
		}
		c->any_trip        = trip;
		c->value_rows_only = !grad && !trip;
		c->facts_forced    = false;  // (the group's verdict is for THIS assemble: a later fi_assemble of a member looks again)
	}
	const int reach = fi::model_reach(c->w);
	const int want_reach = reach > 1 ? reach : 1;
	int want_halo = want_reach;
	if (c->nranks > 1 && c->g.ndim == 3 && c->poly_terms >= 3 && c->poly_terms <= 5 && !c->any_trip &&
	    !fi::test_switch("FI_NO_DEEP_HALO")) {
		const int deep = 2 * (c->poly_terms - 1);
		const int thinnest = c->g.gn[2] / c->nranks;  // (the equal split: floor(G / n) is the thinnest slab)
		if (deep > want_halo && thinnest >= deep) { want_halo = deep; }
	}
	// ... and the V-cycle's polynomial smoother (fp32 levels of value rows, 3 to 5 terms: poly_chain, fi_multigrid.hip)
	if (c->nranks > 1 && c->g.ndim == 3 && c->mg_mode == 1 && c->mg_smoother == 1 && c->levels_wanted > 0 && c->value_rows_only &&
	    !c->any_trip && (c->dtype == FI_F32 || c->mixed) && c->mg_terms >= 3 && c->mg_terms <= 5 &&
	    !(c->w.model_3 > 0 || c->w.model_4 > 0 || c->w.gradient_smoothness > 0) && !fi::test_switch("FI_NO_DEEP_HALO") &&
	    !fi::test_switch("FI_MG_FULL_SMOOTHER")) {
		const int deep = 2 * (c->mg_terms - 1);
		const int thinnest = c->g.gn[2] / c->nranks;
		if (deep > want_halo && thinnest >= deep) { want_halo = deep; }
	}
	c->min_slab = c->nranks > 1 ? c->g.gn[c->g.ndim - 1] / c->nranks : c->g.gn[c->g.ndim - 1];
	if (c->nranks > 1 && (want_halo != c->halo || want_reach != c->reach)) {
		c->halo  = want_halo;
		c->reach = want_reach;
		int sizes[3] = {c->g.gn[0], c->g.gn[1], c->g.gn[2]};
		fi::compute_geom(c, c->g.ndim, sizes);
		c->vectors_ready = false;
		c->vectors_stale = true;
	}
	if (c->nranks > 1) {
		FI_REQUIRE(c->slab_hi - c->slab_lo >= c->reach, FI_ERR_UNSUPPORTED,
		           "slab of %d planes is thinner than the stencil reach %d", c->slab_hi - c->slab_lo, c->reach);
	}
	// The coarser levels are problems of their own, assembled from the same point batches: a helper thread builds them on
	// a second stream while this one assembles the finest level (both are chains of small launches with host round trips
	// for list sizes; 256^3 with one coarser level: 2.05 -> 1.6 ms).  Contexts without triplet rows; the
	// helper's failure is re-raised here.  The helper does no communication: over slabs the levels' exchange of the
	// diagonal's ghost planes is done below, by this thread.
	// Mixed precision: the fp32 replica and ITS levels are the helper's work (the fp64 context keeps no levels of its own).
	const bool mixed64 = c->mixed && c->dtype == FI_F64;
	c->want_lump = mixed64 && fi::lumped_twin_wanted(c);  // (the assembly forms the replica's diagonal beside A^T b)
	const bool beside = (c->levels_wanted > 0 || mixed64) && !c->any_trip && !fi::test_switch("FI_SERIAL_LEVELS");
	if (beside && mixed64) {  // (levels an earlier, unmixed assemble may have left on this context)
		const int keep = c->levels_wanted;
		c->levels_wanted = 0;
		fi::build_levels(c);
		c->levels_wanted = keep;
	}
	if (beside) {
		if (!c->level_stream) {
			c->level_stream = fi::stream_take();
			FI_HIP_TRY(hipEventCreateWithFlags(&c->ev_level, hipEventDisableTiming));
		}
		FI_HIP_TRY(hipEventRecord(c->ev_level, c->stream));  // the point batches were written on the solver stream
		FI_HIP_TRY(hipStreamWaitEvent(c->level_stream, c->ev_level, 0));
		int         helper_code = FI_OK;
		std::string helper_msg;
		// mixed precision: the replica's finest level on `level_stream`, its coarser levels on `level_stream2` -- two more
		// chains of small launches beside this thread's (256^3, 3 coarser levels: 4.9 ms one after the other, 4.3 with one
		// helper, 3 with two)
		if (mixed64) {
			fi::twin_prepare(c);
			if (!c->level_stream2) {
				c->level_stream2 = fi::stream_take();
				FI_HIP_TRY(hipEventCreateWithFlags(&c->ev_level2, hipEventDisableTiming));
			}
			FI_HIP_TRY(hipStreamWaitEvent(c->level_stream2, c->ev_level, 0));
		}
		auto guarded = [&](auto&& work, int* code, std::string* msg) {
			try {
				FI_HIP_TRY(hipSetDevice(c->device));
				fi::AllocStream none(nullptr);  // (a helper's buffers: each level names its own stream, fi_levels.hip)
				work();
			} catch (const fi::Fail& f) {
				*code = f.code;
				*msg  = fi_last_error();  // thread-local: carried over to the caller's thread below
			} catch (...) {
				*code = FI_ERR_HIP;
				*msg  = "unexpected exception while building the coarser levels";
			}
		};
		const bool lumped = mixed64 && c->twin && c->twin->lumped;
		auto build = [&]() {
			guarded([&]() {
				if (lumped) {
					// (the replica's finest level needs the assembled fp64 operator: built below, by this thread)
				} else if (mixed64) {
					fi::twin_assemble(c, c->level_stream);
				} else {
					fi::build_levels(c, nullptr, c->level_stream);
				}
			}, &helper_code, &helper_msg);
		};
		int         helper2_code = FI_OK;
		std::string helper2_msg;
		auto build2 = [&]() {
			guarded([&]() { fi::build_levels(c->twin, c, c->level_stream2); }, &helper2_code, &helper2_msg);
		};
		// (persistent threads: fi_workers.h)
		fi::Worker *helper = nullptr, *helper2 = nullptr;
		try {
			if (!lumped) {
				helper = fi::worker_pool().acquire();
				helper->run(build);
			}
			if (mixed64) {
				helper2 = fi::worker_pool().acquire();
				helper2->run(build2);
			}
		} catch (...) {  // no thread to be had: the levels are built below, after the finest level, on their stream
		}
		int main_code = FI_OK;
		c->defer_scaling_exchange = true;  // slabs: the one exchange of the assembly comes after the ranks have agreed (below)
		try {
			fi::AllocStream alloc_on(c->stream);  // (this chain's buffers are first used on the solver stream: fi_internal.h)
			fi::chain_mark("threads started");
			fi::assemble(c);
			fi::chain_mark("finest: rows assembled");
			fi::generic_assemble(c);
			fi::stencil_prepare(c);
			fi::chain_mark("finest: stencil prepared");
			fi::operator_prepare(c);
			fi::chain_mark("finest: operator prepared");
		} catch (const fi::Fail& f) {
			main_code = f.code;
		} catch (...) {  // never leave the helper unjoined
			main_code = FI_ERR_HIP;
			fi::set_error("unexpected exception while assembling the finest level");
		}
		// The lumped replica needs nothing but the finest level this thread has just assembled: built here, on the solver
		// stream, while the helpers are still busy with the coarser levels (0.3 ms of a 256^3 assemble).  No communication
		// (its share of the assembly's one exchange comes with operator_finish_ghosts below).
		if (lumped && main_code == FI_OK) {
			try {
				fi::AllocStream alloc_on(c->stream);
				c->twin->defer_scaling_exchange = true;
				fi::twin_assemble_lumped(c);
				fi::chain_mark("finest: lumped replica");
			} catch (const fi::Fail& f) {
				main_code = f.code;
			} catch (...) {
				main_code = FI_ERR_HIP;
				fi::set_error("unexpected exception while building the lumped replica");
			}
		}
		if (helper) {
			helper->wait();
			fi::worker_pool().release(helper);
		} else if (main_code == FI_OK && !lumped) {
			build();
		}
		if (mixed64) {
			if (helper2) {
				helper2->wait();
				fi::worker_pool().release(helper2);
			} else if (main_code == FI_OK) {
				build2();
			}
			if (helper_code == FI_OK && helper2_code != FI_OK) {
				helper_code = helper2_code;
				helper_msg  = helper2_msg;
			}
		}
		fi::chain_mark("helpers joined");
		fi_ctx* const first_built = mixed64 ? c->twin : c->coarse;  // the replica, then its levels / the levels
		for (fi_ctx* l = first_built; l; l = l->coarse) { l->stream = c->stream; }
		const bool mine_ok = main_code == FI_OK && helper_code == FI_OK;
		bool peers_ok = true;
		try {
			peers_ok = fi::all_ranks_ok(c, mine_ok);  // (every rank gets here: nothing above is collective)
		} catch (const fi::Fail&) {
			peers_ok = false;
		}
		if (!mine_ok || !peers_ok) {
			(void)hipStreamSynchronize(c->level_stream);  // nothing of the helpers' work stays in flight behind the error
			if (c->level_stream2) { (void)hipStreamSynchronize(c->level_stream2); }
			c->defer_scaling_exchange = false;
			if (main_code != FI_OK) { throw fi::Fail{main_code}; }
			if (helper_code != FI_OK) {
				fi::set_error("%s", helper_msg.c_str());
				throw fi::Fail{helper_code};
			}
			fi::set_error("fi_assemble: another rank failed while assembling its slab");
			throw fi::Fail{FI_ERR_COMM};
		}
		if (fi::tuning_switch("FI_ASM_CHAIN_TIMES")) {  // (diagnostic: where each chain of the assembly ends, from the start event)
			hipEvent_t a = nullptr, b = nullptr, d = nullptr;
			FI_HIP_TRY(hipEventCreate(&a));
			FI_HIP_TRY(hipEventCreate(&b));
			FI_HIP_TRY(hipEventCreate(&d));
			FI_HIP_TRY(hipEventRecord(a, c->stream));
			FI_HIP_TRY(hipEventRecord(b, c->level_stream));
			if (mixed64) { FI_HIP_TRY(hipEventRecord(d, c->level_stream2)); }
			FI_HIP_TRY(hipEventSynchronize(a));
			FI_HIP_TRY(hipEventSynchronize(b));
			if (mixed64) { FI_HIP_TRY(hipEventSynchronize(d)); }
			float ta = 0, tb = 0, td = 0;
			(void)hipEventElapsedTime(&ta, e0, a);
			(void)hipEventElapsedTime(&tb, e0, b);
			if (mixed64) { (void)hipEventElapsedTime(&td, e0, d); }
			std::fprintf(stderr, "fi_assemble chains: solver stream %.3f ms, level stream %.3f ms, second level stream %.3f ms\n", ta, tb, td);
			(void)hipEventDestroy(a);
			(void)hipEventDestroy(b);
			(void)hipEventDestroy(d);
		}
		FI_HIP_TRY(hipEventRecord(c->ev_level, c->level_stream));
		FI_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_level, 0));
		if (mixed64) {
			FI_HIP_TRY(hipEventRecord(c->ev_level2, c->level_stream2));
			FI_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_level2, 0));
			fi::twin_finish(c);
		}
		// slabs: the levels' share of the assembly's one exchange (the diagonal's ghost planes), in level order on every rank
		fi::operator_finish_ghosts(c);
		for (fi_ctx* l = first_built; l; l = l->coarse) { fi::operator_finish_ghosts(l); }
	} else {
	fi::AllocStream alloc_on(c->stream);
	fi::assemble(c);
	fi::generic_assemble(c);
	fi::stencil_prepare(c);
	fi::operator_prepare(c);
	}
	if (beside) {
		if (!mixed64) { fi::build_twin(c); }  // (drops a replica left by an earlier, mixed assemble)
	} else if (mixed64) {  // the fp32 replica carries the levels
		const int keep = c->levels_wanted;
		c->levels_wanted = 0;
		fi::build_levels(c);
		c->levels_wanted = keep;
		fi::build_twin(c);
	} else {
		fi::build_levels(c);
		fi::build_twin(c);
	}
	FI_HIP_TRY(hipEventRecord(e1, c->stream));
	fi::chain_mark("fi_assemble returns");
	c->asm_time_pending   = true;
	c->stats.assemble_ms  = 0.0;
	c->stats.num_levels   = 1;
	for (fi_ctx* l = c->coarse; l; l = l->coarse) { c->stats.num_levels += 1; }
	if (c->twin) { c->stats.num_levels = c->twin->stats.num_levels; }
	c->stats.num_unknowns = c->g.nown;
	c->stats.spmv_bytes   = fi::apply_algorithmic_bytes(c);
	c->assembled          = true;
	c->vectors_ready      = false;
	FI_API_END
}

int fi_solve_cg(fi_ctx* c, const float* guess, int max_iterations, float tol, float* out, int* iterations,
                float* rel_residual, int memory)
{
	FI_API_BEGIN
	fi::check_assembled(c);
	fi::bind_device(c);
	FI_REQUIRE(memory == FI_HOST || memory == FI_DEVICE, FI_ERR_INVALID, "bad memory kind %d", memory);
	fi::AllocStream alloc_on(c->stream);  // (every level of a hierarchy solves on the finest level's stream)
	if (c->dtype == FI_F64) {
		fi::solve_cg_t<double>(c, guess, max_iterations, tol, out, iterations, rel_residual, memory);
	} else {
		fi::solve_cg_t<float>(c, guess, max_iterations, tol, out, iterations, rel_residual, memory);
	}
	FI_API_END
}

int fi_set_option(fi_ctx* c, int option, double value)
{
	FI_API_BEGIN
	fi::check_ctx(c);
	switch (option) {
	case FI_OPT_VERIFY_RESIDUAL: c->verify_residual = value != 0.0; break;
	case FI_OPT_LEVELS:
		c->levels_wanted = value > 0 ? static_cast<int>(value) : 0;
		c->assembled = false;
		break;
	case FI_OPT_COARSE_TOLERANCE: c->coarse_tol = value > 0 ? value : 1e-3; break;
	case FI_OPT_MULTIGRID: c->mg_mode = value != 0.0 ? 1 : 0; break;
	case FI_OPT_MIXED_PRECISION:
		FI_REQUIRE(value == 0.0 || c->dtype == FI_F64, FI_ERR_INVALID, "FI_OPT_MIXED_PRECISION needs an FI_F64 context");
		c->mixed = value != 0.0 ? 1 : 0;
		c->assembled = false;
		break;
	case FI_OPT_POLY_TERMS:
		FI_REQUIRE(value >= 0 && value <= 32, FI_ERR_INVALID, "FI_OPT_POLY_TERMS must be 0..32");
		c->poly_terms = static_cast<int>(value);
		break;
	case FI_OPT_POLY_RATIO:
		FI_REQUIRE(value > 1.0 && value <= 1000.0, FI_ERR_INVALID, "FI_OPT_POLY_RATIO must be in (1, 1000]");
		c->poly_ratio = value;
		break;
	case FI_OPT_MG_SMOOTHER:
		c->mg_smoother = value != 0.0 ? 1 : 0;
		c->assembled = false;  // the levels take the setting when they are built
		break;
	case FI_OPT_MG_SAFE_FACTOR:
		FI_REQUIRE(value >= 1.0 && value <= 64.0, FI_ERR_INVALID, "FI_OPT_MG_SAFE_FACTOR must be 1..64");
		c->mg_safe = value;
		c->assembled = false;
		break;
	case FI_OPT_MG_TERMS:
		FI_REQUIRE(value >= 2 && value <= 16, FI_ERR_INVALID, "FI_OPT_MG_TERMS must be 2..16");
		c->mg_terms = static_cast<int>(value);
		c->assembled = false;  // the levels take the setting when they are built
		break;
	case FI_OPT_MG_RATIO:
		FI_REQUIRE(value > 1.0 && value <= 1000.0, FI_ERR_INVALID, "FI_OPT_MG_RATIO must be in (1, 1000]");
		c->mg_pratio = value;
		c->assembled = false;
		break;
	case FI_OPT_MG_CHEB_DEGREE:
		FI_REQUIRE(value == 0 || (value >= 2 && value <= 16), FI_ERR_INVALID, "FI_OPT_MG_CHEB_DEGREE must be 0 or 2..16");
		c->mg_cheb_degree = static_cast<int>(value);
		c->assembled = false;
		break;
	case FI_OPT_MG_CHEB_RATIO:
		FI_REQUIRE(value == 0 || (value > 1.0 && value <= 1000.0), FI_ERR_INVALID, "FI_OPT_MG_CHEB_RATIO must be 0 or in (1, 1000]");
		c->mg_cheb_ratio = value;
		c->assembled = false;
		break;
	case FI_OPT_MG_KCYCLE:
		FI_REQUIRE(value >= 0 && value <= 16, FI_ERR_INVALID, "FI_OPT_MG_KCYCLE must be 0..16");
		c->mg_kcycle = static_cast<int>(value);
		c->assembled = false;  // the levels take the setting when they are built
		break;
	case FI_OPT_FIELD_TOLERANCE:
		FI_REQUIRE(value >= 0.0 && value < 1.0, FI_ERR_INVALID, "FI_OPT_FIELD_TOLERANCE must be in [0, 1)");
		c->field_tol = value;
		break;
	default: FI_REQUIRE(false, FI_ERR_INVALID, "unknown option %d", option);
	}
	FI_API_END
}

int fi_jacobi(fi_ctx* c, const float* guess, int num_iterations, float weight, float* out, int memory)
{
	FI_API_BEGIN
	fi::check_assembled(c);
	fi::bind_device(c);
	FI_REQUIRE(memory == FI_HOST || memory == FI_DEVICE, FI_ERR_INVALID, "bad memory kind %d", memory);
	if (num_iterations < 0) { num_iterations = 0; }  // sparse_linear.cpp:220: returns the guess
	if (c->dtype == FI_F64) {
		fi::jacobi_t<double>(c, guess, num_iterations, weight, out, memory);
	} else {
		fi::jacobi_t<float>(c, guess, num_iterations, weight, out, memory);
	}
	FI_API_END
}

int fi_error_map(fi_ctx* c, const float* solution, float* out, int memory)
{
	FI_API_BEGIN
	fi::check_assembled(c);
	fi::bind_device(c);
	FI_REQUIRE(memory == FI_HOST || memory == FI_DEVICE, FI_ERR_INVALID, "bad memory kind %d", memory);
	FI_REQUIRE(solution && out, FI_ERR_INVALID, "fi_error_map needs a solution and an output buffer");
	fi::ensure_vectors(c);
	fi::RankSet R{c};
	if (c->dtype == FI_F64) {
		fi::load_owned<double>(c, c->x, solution, memory);
		fi::halo_exchange(R, &fi_ctx::x);
		fi::error_map(c, c->x.p, c->q.p);
		fi::store_owned<double>(c, c->q, out, memory);
	} else {
		fi::load_owned<float>(c, c->x, solution, memory);
		fi::halo_exchange(R, &fi_ctx::x);
		fi::error_map(c, c->x.p, c->q.p);
		fi::store_owned<float>(c, c->q, out, memory);
	}
	FI_API_END
}

int fi_tile_pass(fi_ctx* c, const float* guess, int tile_size, float* out, int memory)
{
	FI_API_BEGIN
	fi::check_assembled(c);
	fi::bind_device(c);
	FI_REQUIRE(memory == FI_HOST || memory == FI_DEVICE, FI_ERR_INVALID, "bad memory kind %d", memory);
	FI_REQUIRE(tile_size >= 2, FI_ERR_INVALID, "tile_size %d < 2 (sparse_linear.cpp:254)", tile_size);
	FI_REQUIRE(guess && out, FI_ERR_INVALID, "fi_tile_pass needs a guess and an output buffer");
	FI_REQUIRE(c->nranks == 1 || c->generic.ntrip == 0, FI_ERR_UNSUPPORTED, "the tile pre-solver over triplet rows needs an undivided lattice");
	if (c->dtype == FI_F64) {
		fi::tile_pass_t<double>(c, guess, tile_size, out, memory);
	} else {
		fi::tile_pass_t<float>(c, guess, tile_size, out, memory);
	}
	FI_API_END
}

int fi_get_solution_f64(fi_ctx* c, double* out)
{
	FI_API_BEGIN
	fi::check_assembled(c);
	fi::bind_device(c);
	FI_REQUIRE(c->vectors_ready && out, FI_ERR_STATE, "no solution yet");
	c->dtype == FI_F64 ? fi::get_vec_f64_t<double>(c, c->x, out) : fi::get_vec_f64_t<float>(c, c->x, out);
	FI_API_END
}

int fi_true_residual(fi_ctx* c, double* rel)
{
	FI_API_BEGIN
	fi::check_assembled(c);
	fi::bind_device(c);
	FI_REQUIRE(rel != nullptr, FI_ERR_INVALID, "null output");
	*rel = c->dtype == FI_F64 ? fi::true_residual_t<double>(c) : fi::true_residual_t<float>(c);
	FI_API_END
}

int fi_apply_AtA_f64(fi_ctx* c, const double* x, double* y)
{
	FI_API_BEGIN
	fi::check_assembled(c);
	fi::bind_device(c);
	FI_REQUIRE(x && y, FI_ERR_INVALID, "null vector");
	c->dtype == FI_F64 ? fi::apply_f64_t<double>(c, x, y) : fi::apply_f64_t<float>(c, x, y);
	FI_API_END
}

int fi_get_Atb_f64(fi_ctx* c, double* out)
{
	FI_API_BEGIN
	fi::check_assembled(c);
	fi::bind_device(c);
	c->dtype == FI_F64 ? fi::get_vec_f64_t<double>(c, c->atb, out) : fi::get_vec_f64_t<float>(c, c->atb, out);
	FI_API_END
}

int fi_get_diag_f64(fi_ctx* c, double* out)
{
	FI_API_BEGIN
	fi::check_assembled(c);
	fi::bind_device(c);
	c->dtype == FI_F64 ? fi::get_vec_f64_t<double>(c, c->diag, out) : fi::get_vec_f64_t<float>(c, c->diag, out);
	FI_API_END
}

int fi_get_stats(const fi_ctx* c, fi_stats* out)
{
	FI_API_BEGIN
	fi::check_ctx(c);
	FI_REQUIRE(out != nullptr, FI_ERR_INVALID, "null output");
	fi::finish_assemble_timing(const_cast<fi_ctx*>(c));
	*out = c->stats;
	// Levels of the coarse-to-fine start that were solved without a look at their stop flag (cg_run, `unwatched`) reported
	// the PREDICTED iteration count; the flag's copy has reached pinned memory since (the finest level's solve ended with a
	// synchronisation behind it): the caller gets what the levels really did, and `coarse_unconverged` says how many of
	// them had not met their tolerance after the predicted steps (their start guess was that much poorer, nothing else).
	out->coarse_unconverged = 0;
	const fi_ctx* top = c->twin && c->twin->coarse ? c->twin : c;
	for (const fi_ctx* l = top->coarse; l; l = l->coarse) {
		if (!l->unwatched_pending || !l->ev_unwatched || !l->pin[2]) { continue; }
		if (hipEventSynchronize(l->ev_unwatched) != hipSuccess) { continue; }
		const fi::CgScalars* was = static_cast<const fi::CgScalars*>(l->pin[2]);
		const bool met = was->done == 1 || was->done == 5 || was->done == 4;
		if (!met) { out->coarse_unconverged += 1; }
		out->coarse_iterations += (met ? was->iter : l->unwatched_expected) - l->unwatched_expected;
	}
	FI_API_END
}

int fi_time_apply(fi_ctx* c, int reps, double* ms_per_launch)
{
	FI_API_BEGIN
	fi::check_assembled(c);
	fi::bind_device(c);
	FI_REQUIRE(reps > 0 && ms_per_launch, FI_ERR_INVALID, "bad arguments");
	fi::ensure_vectors(c);
	fi::CgScalars init{};
	FI_HIP_TRY(hipMemcpyAsync(c->scal.p, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
	fi::EventPair timer;  // (destroyed on every way out: a coarse level's breakdown, a timeout)
	const hipEvent_t e0 = timer.e0, e1 = timer.e1;
	// timing builds: FI_TIME_STEP = 1 / 2 / 3 times that step of the polynomial preconditioner instead (operands: the
	// solver's vectors as they are -- isolated launches, the numbers of profiles/r2_ablation.md)
	const char* which = fi::tuning_switch("FI_TIME_STEP");
	const int   step = which ? atoi(which) : 0;
	auto launch = [&]() {
		if (step >= 1 && fi::stencil_cheb_available(c)) {
			c->dtype == FI_F64 ? fi::ensure_poly_vectors<double>(c) : fi::ensure_poly_vectors<float>(c);
			if (step == 1) {
				fi::stencil_cheb_step(c, c->r.p, nullptr, c->r.p, c->mg_d.p, 0.3, 0.2, c->partial.as<double>(), 0, 0.0, 0.5);
			} else if (step == 2) {
				fi::stencil_cheb_step(c, c->mg_d.p, c->mg_x.p, c->r.p, c->mg_x.p, 0.3, 0.2, c->partial.as<double>(), 0, 0.5);
			} else {
				fi::stencil_cheb_step(c, c->mg_x.p, c->mg_d.p, c->r.p, c->mg_d.p, 0.3, 0.2, c->partial.as<double>());
			}
		} else {
			fi::apply_AtA(c, c->p.p, c->q.p, c->partial.as<double>());
		}
	};
	launch();  // warm-up
	FI_HIP_TRY(hipEventRecord(e0, c->stream));
	for (int k = 0; k < reps; ++k) { launch(); }
	FI_HIP_TRY(hipEventRecord(e1, c->stream));
	FI_HIP_TRY(hipEventSynchronize(e1));
	float ms = 0;
	FI_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	*ms_per_launch = ms / reps;
	FI_API_END
}

int fi_upscale_field(const float* small_field, int ndim, const int* small_sizes, const int* large_sizes, float* out,
                     int memory)
{
	FI_API_BEGIN
	FI_REQUIRE(small_field && small_sizes && large_sizes && out, FI_ERR_INVALID, "null argument");
	FI_REQUIRE(1 <= ndim && ndim <= FI_MAX_DIM, FI_ERR_INVALID, "ndim must be 1..3");
	FI_REQUIRE(memory == FI_HOST || memory == FI_DEVICE, FI_ERR_INVALID, "bad memory kind %d", memory);
	fi::UpscaleArgs a{};
	a.ndim = ndim;
	int64_t ns = 1, nl = 1;
	for (int d = 0; d < ndim; ++d) {
		FI_REQUIRE(small_sizes[d] >= 1 && large_sizes[d] >= 1, FI_ERR_INVALID, "bad size");
		a.ssz[d] = small_sizes[d];
		a.lsz[d] = large_sizes[d];
		ns *= small_sizes[d];
		nl *= large_sizes[d];
	}
	fi::DevBuf ds, dl;
	const float* s = small_field;
	float*       o = out;
	if (memory == FI_HOST) {
		ds.alloc(sizeof(float) * ns);
		dl.alloc(sizeof(float) * nl);
		FI_HIP_TRY(hipMemcpy(ds.p, small_field, sizeof(float) * ns, hipMemcpyHostToDevice));
		s = ds.as<float>();
		o = dl.as<float>();
	}
	hipLaunchKernelGGL(fi::k_upscale, dim3(fi::blocks_for(nl)), dim3(fi::kThreads), 0, nullptr, a, nl, s, o);
	FI_HIP_TRY(hipGetLastError());
	FI_HIP_TRY(hipDeviceSynchronize());
	if (memory == FI_HOST) { FI_HIP_TRY(hipMemcpy(out, dl.p, sizeof(float) * nl, hipMemcpyDeviceToHost)); }
	FI_API_END
}



}  // extern "C"
