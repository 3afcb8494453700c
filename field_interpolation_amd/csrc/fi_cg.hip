// fi_cg.hip -- rank sets (one slab per process over RCCL, or the loop-back group), Jacobi-preconditioned CG with the same
// stop rule as Eigen::BiCGSTAB in sparse_linear.cpp:199-206 / :429-436, weighted Jacobi (sparse_linear.cpp:214-241), the tile
// pre-solver's driver (sparse_linear.cpp:246-390), true residual and the fp64 test hooks.
#include "fi_solver_internal.h"

namespace fi {

// ---- iteration counts across contexts (see fi_solver_internal.h) ------------------------------------------------------
namespace {
struct PredKey {
	int    kind, dtype, ndim, gn[3], level, below, mixed, rows_only;
	float  w[8];
	double tol;
	bool operator<(const PredKey& o) const { return std::memcmp(this, &o, sizeof(PredKey)) < 0; }
};
PredKey pred_key(const fi_ctx* c, int kind, double tol)
{
	PredKey k;
	std::memset(&k, 0, sizeof(k));  // (padding bytes take part in the comparison)
	k.kind  = kind;
	k.dtype = c->dtype;
	k.ndim  = c->g.ndim;
	for (int d = 0; d < 3; ++d) { k.gn[d] = c->g.gn[d]; }
	k.level = c->level;
	for (const fi_ctx* l = (c->twin && c->twin->coarse ? c->twin->coarse : c->coarse); l; l = l->coarse) { ++k.below; }
	k.mixed     = c->mixed;
	k.rows_only = c->value_rows_only ? 1 : 0;
	const float w[8] = {c->w.data_pos, c->w.data_gradient, c->w.model_0, c->w.model_1, c->w.model_2, c->w.model_3, c->w.model_4,
	                    c->w.gradient_smoothness};
	std::memcpy(k.w, w, sizeof(w));
	k.tol = tol;
	return k;
}
std::mutex g_pred_mutex;
std::map<PredKey, int> g_pred;
}  // namespace

void remember_iterations(const fi_ctx* c, int kind, double tol, int iterations)
{
	if (c->nranks != 1 || iterations <= 0 || test_switch("FI_NO_LAMBDA_CACHE")) { return; }
	std::lock_guard<std::mutex> lock(g_pred_mutex);
	g_pred[pred_key(c, kind, tol)] = iterations;
}

// once per context and kind: afterwards the context's own history decides (a level whose prediction failed must be
// able to watch its flag again)
int recall_iterations(fi_ctx* c, int kind, double tol)
{
	if (c->pred_recalled[kind]) { return 0; }
	c->pred_recalled[kind] = true;
	if (c->nranks != 1 || test_switch("FI_NO_LAMBDA_CACHE") || test_switch("FI_LOOK_ALWAYS")) { return 0; }
	std::lock_guard<std::mutex> lock(g_pred_mutex);
	auto it = g_pred.find(pred_key(c, kind, tol));
	return it == g_pred.end() ? 0 : it->second;
}

// fi_memory_pool(0): a process that wants nothing left over from earlier contexts (bench.py's cold step) forgets the counts too
void forget_iterations()
{
	std::lock_guard<std::mutex> lock(g_pred_mutex);
	g_pred.clear();
}

// ---- vector kernels (owned range is contiguous: the slowest axis is the decomposed one) ---------

// r = b - q; p = Dinv r; partials: r.(Dinv r), r.r, b.b
template <typename T>
__global__ __launch_bounds__(kThreads) void k_cg_init(int64_t n, const T* __restrict__ b, const T* __restrict__ q,
                                                       const T* __restrict__ dinv, T* __restrict__ r,
                                                       T* __restrict__ p, double* __restrict__ partial, int nblk)
{
	double acc[3] = {0, 0, 0};
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		const T bi = b[i];
		const T ri = bi - q[i];
		const T zi = dinv[i] * ri;
		r[i] = ri;
		p[i] = zi;
		acc[0] += static_cast<double>(ri) * static_cast<double>(zi);
		acc[1] += static_cast<double>(ri) * static_cast<double>(ri);
		acc[2] += static_cast<double>(bi) * static_cast<double>(bi);
	}
	double out[3];
	block_sum<3>(acc, out);
	if (threadIdx.x == 0) {
		partial[blockIdx.x]            = out[0];
		partial[nblk + blockIdx.x]     = out[1];
		partial[2 * nblk + blockIdx.x] = out[2];
	}
}

// 16-byte vector view of T for the streaming kernels
template <typename T, bool VEC>
__global__ __launch_bounds__(kThreads) void k_cg_resid(int64_t n, const CgScalars* __restrict__ sc,
                                                        const T* __restrict__ q, const T* __restrict__ dinv,
                                                        T* __restrict__ r, double* __restrict__ partial, int nblk)
{
	if (sc->done) { return; }
	using V = typename Vec16<T>::V;
	constexpr int N = VEC ? Vec16<T>::N : 1;
	const T alpha = static_cast<T>(sc->alpha);
	double acc[2] = {0, 0};
	const int64_t nv = n / N;
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < nv;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		T rv[N], qv[N], dv[N];
		if (VEC) {
			*reinterpret_cast<V*>(rv) = reinterpret_cast<const V*>(r)[i];
			ld16_nt(qv, q, i);
			ld16_nt(dv, dinv, i);
		} else {
			rv[0] = r[i]; qv[0] = q[i]; dv[0] = dinv[i];
		}
		T s0 = T(0), s1 = T(0);
#pragma unroll
		for (int j = 0; j < N; ++j) {
			rv[j] -= alpha * qv[j];
			s0 += rv[j] * (dv[j] * rv[j]);
			s1 += rv[j] * rv[j];
		}
		if (VEC) { reinterpret_cast<V*>(r)[i] = *reinterpret_cast<V*>(rv); } else { r[i] = rv[0]; }
		acc[0] += static_cast<double>(s0);
		acc[1] += static_cast<double>(s1);
	}
	double out[2];
	block_sum<2>(acc, out);
	if (threadIdx.x == 0) {
		partial[blockIdx.x]        = out[0];
		partial[nblk + blockIdx.x] = out[1];
	}
}

// CG step, second half: x += alpha p (p still the direction the step was taken along), then
// p = Dinv r + beta p                                                      (reads x, p, r, Dinv; writes x, p)
// `iteration` is the 1-based number of the CG step these launches belong to: the x update is applied
// exactly once, by the step that actually ran (sc->iter == iteration), also when that step converged.
template <typename T, bool VEC>
__global__ __launch_bounds__(kThreads) void k_cg_xp(int64_t n, const CgScalars* __restrict__ sc, int iteration,
                                                     const T* __restrict__ r, const T* __restrict__ dinv,
                                                     T* __restrict__ x, T* __restrict__ p)
{
	if (sc->iter != iteration || sc->done == 2) { return; }
	using V = typename Vec16<T>::V;
	constexpr int N = VEC ? Vec16<T>::N : 1;
	const T    alpha = static_cast<T>(sc->alpha);
	const T    beta  = static_cast<T>(sc->beta);
	const bool go_on = sc->done == 0;
	const int64_t nv = n / N;
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < nv;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		T xv[N], pv[N], rv[N], dv[N];
		if (VEC) {
			ld16_nt(xv, x, i);
			*reinterpret_cast<V*>(pv) = reinterpret_cast<const V*>(p)[i];
		} else {
			xv[0] = x[i]; pv[0] = p[i];
		}
#pragma unroll
		for (int j = 0; j < N; ++j) { xv[j] += alpha * pv[j]; }
		if (VEC) { st16_nt(x, i, xv); } else { x[i] = xv[0]; }
		if (go_on) {
			if (VEC) {
				ld16_nt(rv, r, i);
				ld16_nt(dv, dinv, i);
			} else {
				rv[0] = r[i]; dv[0] = dinv[i];
			}
#pragma unroll
			for (int j = 0; j < N; ++j) { pv[j] = dv[j] * rv[j] + beta * pv[j]; }
			if (VEC) { reinterpret_cast<V*>(p)[i] = *reinterpret_cast<V*>(pv); } else { p[i] = pv[0]; }
		}
	}
}

template <typename T, bool VEC>
__global__ __launch_bounds__(kThreads) void k_cg_resid_f(int64_t n, const CgScalars* __restrict__ in,
                                                          CgScalars* __restrict__ mid, int tag,
                                                          const double* __restrict__ pq_partial, int pq_count,
                                                          const T* __restrict__ q, const T* __restrict__ dinv,
                                                          T* __restrict__ r, double* __restrict__ partial, int nblk)
{
	// the scalar record is read by ONE thread per workgroup and handed on through LDS: thousands of waves reading the
	// same cache lines queue on one L2 channel (profiles/r2_ablation.md section 6)
	__shared__ double sh_rz;
	__shared__ int    sh_done;
	if (threadIdx.x == 0) {
		sh_done = in->done;
		sh_rz   = in->rz;
	}
	__syncthreads();
	if (sh_done) { return; }
	const double pq    = sum_partials(pq_partial, pq_count);
	const double alpha_d = sh_rz / pq;
	const bool   bad   = !(pq > 0.0) || !isfinite(pq);
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		CgScalars s = *in;
		s.sums[0] = pq;
		s.pq      = pq;
		s.alpha   = alpha_d;
		s.tag     = tag;
		if (bad) { s.done = 2; }  // breakdown; the second half publishes it
		*mid = s;
	}
	if (bad) { return; }
	using V = typename Vec16<T>::V;
	constexpr int N = VEC ? Vec16<T>::N : 1;
	const T alpha = static_cast<T>(alpha_d);
	double acc[2] = {0, 0};
	const int64_t nv = n / N;
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < nv;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		T rv[N], qv[N], dv[N];
		if (VEC) {
			*reinterpret_cast<V*>(rv) = reinterpret_cast<const V*>(r)[i];
			ld16_nt(qv, q, i);
			ld16_nt(dv, dinv, i);
		} else {
			rv[0] = r[i]; qv[0] = q[i]; dv[0] = dinv[i];
		}
		T s0 = T(0), s1 = T(0);
#pragma unroll
		for (int j = 0; j < N; ++j) {
			rv[j] -= alpha * qv[j];
			s0 += rv[j] * (dv[j] * rv[j]);
			s1 += rv[j] * rv[j];
		}
		if (VEC) { reinterpret_cast<V*>(r)[i] = *reinterpret_cast<V*>(rv); } else { r[i] = rv[0]; }
		acc[0] += static_cast<double>(s0);
		acc[1] += static_cast<double>(s1);
	}
	double out[2];
	block_sum<2>(acc, out);
	if (threadIdx.x == 0) {
		partial[blockIdx.x]        = out[0];
		partial[nblk + blockIdx.x] = out[1];
	}
}

// second half: beta and the stop test from the partials of the first half, x += alpha p, p = Dinv r + beta p
template <typename T, bool VEC>
__global__ __launch_bounds__(kThreads) void k_cg_xp_f(int64_t n, const CgScalars* __restrict__ mid,
                                                       CgScalars* __restrict__ out_sc, int tag,
                                                       const double* __restrict__ partial, int nblk,
                                                       const T* __restrict__ r, const T* __restrict__ dinv,
                                                       T* __restrict__ x, T* __restrict__ p)
{
	__shared__ CgScalars sh;  // read once per workgroup (see k_cg_resid_f)
	if (threadIdx.x == 0) { sh = *mid; }
	__syncthreads();
	if (sh.tag != tag) { return; }  // the first half of this iteration did not run: the solve had finished
	if (sh.done == 2) {
		if (blockIdx.x == 0 && threadIdx.x == 0) { *out_sc = sh; }
		return;
	}
	const double rz_new = sum_partials(partial, nblk);
	const double rr     = sum_partials(partial + nblk, nblk);
	const double beta_d = rz_new / sh.rz;
	const int    iter   = sh.iter + 1;
	const int    done   = !isfinite(rr) ? 2 : (!(rr > sh.tol2) ? 1 : (iter >= sh.max_iter ? 3 : 0));
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		CgScalars s = sh;
		s.sums[0] = rz_new;
		s.sums[1] = rr;
		s.rz_new  = rz_new;
		s.rr      = rr;
		s.beta    = beta_d;
		s.rz      = rz_new;
		s.iter    = iter;
		s.done    = done;
		*out_sc   = s;
	}
	if (done == 2) { return; }
	using V = typename Vec16<T>::V;
	constexpr int N = VEC ? Vec16<T>::N : 1;
	const T    alpha = static_cast<T>(sh.alpha);
	const T    beta  = static_cast<T>(beta_d);
	const bool go_on = done == 0;
	const int64_t nv = n / N;
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < nv;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		T xv[N], pv[N], rv[N], dv[N];
		if (VEC) {
			ld16_nt(xv, x, i);
			*reinterpret_cast<V*>(pv) = reinterpret_cast<const V*>(p)[i];
		} else {
			xv[0] = x[i]; pv[0] = p[i];
		}
#pragma unroll
		for (int j = 0; j < N; ++j) { xv[j] += alpha * pv[j]; }
		if (VEC) { st16_nt(x, i, xv); } else { x[i] = xv[0]; }
		if (go_on) {
			if (VEC) {
				ld16_nt(rv, r, i);
				ld16_nt(dv, dinv, i);
			} else {
				rv[0] = r[i]; dv[0] = dinv[i];
			}
#pragma unroll
			for (int j = 0; j < N; ++j) { pv[j] = dv[j] * rv[j] + beta * pv[j]; }
			if (VEC) { reinterpret_cast<V*>(p)[i] = *reinterpret_cast<V*>(pv); } else { p[i] = pv[0]; }
		}
	}
}

// x <- x + w * (b - q) * Dinv     (jacobi_iterations, sparse_linear.cpp:233-239, algebraically identical)
template <typename T>
__global__ __launch_bounds__(kThreads) void k_jacobi_update(int64_t n, T w, const T* __restrict__ b,
                                                             const T* __restrict__ q, const T* __restrict__ dinv,
                                                             T* __restrict__ x)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		x[i] += w * (b[i] - q[i]) * dinv[i];
	}
}

// r = b - q; partials r.r, b.b
template <typename T>
__global__ __launch_bounds__(kThreads) void k_residual_norm(int64_t n, const T* __restrict__ b, const T* __restrict__ q,
                                                             double* __restrict__ partial, int nblk)
{
	double acc[2] = {0, 0};
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		const double bi = static_cast<double>(b[i]);
		const double ri = bi - static_cast<double>(q[i]);
		acc[0] += ri * ri;
		acc[1] += bi * bi;
	}
	double out[2];
	block_sum<2>(acc, out);
	if (threadIdx.x == 0) {
		partial[blockIdx.x]        = out[0];
		partial[nblk + blockIdx.x] = out[1];
	}
}

// ---- host side -------------------------------------------------------------------------------------

void compute_geom(fi_ctx* c, int ndim, const int* sizes)
{
	Geom& g = c->g;
	g = Geom{};
	g.ndim = ndim;
	c->vectors_stale = true;  // (ghost planes may have moved)
	const int L = ndim - 1;
	for (int d = 0; d < 3; ++d) {
		g.gn[d]     = d < ndim ? sizes[d] : 1;
		g.n[d]      = g.gn[d];
		g.off[d]    = 0;
		g.own_lo[d] = 0;
		g.own_hi[d] = g.gn[d];
		g.cn[d]     = d < ndim ? g.gn[d] + 1 : 1;
		g.coff[d]   = d < ndim ? -1 : 0;
		g.pshift[d] = c->pos_shift[d];
	}
	const int G = g.gn[L];
	if (!c->slab_fixed) {
		c->slab_lo = static_cast<int>(static_cast<int64_t>(c->rank) * G / c->nranks);
		c->slab_hi = static_cast<int>(static_cast<int64_t>(c->rank + 1) * G / c->nranks);
	}
	const int H = c->nranks > 1 ? c->halo : 0;
	g.n[L]      = (c->slab_hi - c->slab_lo) + 2 * H;
	g.off[L]    = c->slab_lo - H;
	g.own_lo[L] = H;
	g.own_hi[L] = H + (c->slab_hi - c->slab_lo);
	if (c->nranks > 1) {
		g.cn[L]   = (c->slab_hi - c->slab_lo) + 1;
		g.coff[L] = c->slab_lo - 1;
	}
	int64_t s = 1;
	for (int d = 0; d < 3; ++d) {
		g.stride[d] = s;
		s *= g.n[d];
	}
	g.nloc = s;
	g.nown = 1;
	for (int d = 0; d < 3; ++d) { g.nown *= (g.own_hi[d] - g.own_lo[d]); }
	g.own_first = static_cast<int64_t>(g.own_lo[L]) * g.stride[L];
}

int model_reach(const fi_weights& w)
{
	int k = 0;
	if (w.model_1 > 0) { k = 1; }
	if (w.model_2 > 0) { k = 2; }
	if (w.model_3 > 0) { k = 3; }
	if (w.model_4 > 0) { k = 4; }
	return k;
}

void ensure_vectors(fi_ctx* c)
{
	if (c->vectors_ready) { return; }
	const size_t es = elem_size(c);
	const Geom&  g  = c->g;
	// zeroed when they are new or the local geometry has changed (ghost planes outside the lattice are never written and
	// must stay finite); a re-assembled problem of the same shape keeps them (every solve writes all it reads: 4 fills of
	// 134 MB = 80 us per step of a 256^3 fp64 context)
	const size_t need = es * g.nloc;
	const bool fresh = c->vectors_stale || c->x.bytes < need || c->r.bytes < need || c->p.bytes < need || c->q.bytes < need;
	c->x.alloc(need);
	c->r.alloc(need);
	c->p.alloc(need);
	c->q.alloc(need);
	if (fresh) {
		FI_HIP_TRY(hipMemsetAsync(c->x.p, 0, need, c->stream));
		FI_HIP_TRY(hipMemsetAsync(c->r.p, 0, need, c->stream));
		FI_HIP_TRY(hipMemsetAsync(c->p.p, 0, need, c->stream));
		FI_HIP_TRY(hipMemsetAsync(c->q.p, 0, need, c->stream));
	}
	c->vectors_stale = false;
	int nb = apply_num_partials(c);
	if (nb < 4096) { nb = 4096; }  // also covers the plain kernels of the tile operator (fi_tile_pass)
	if (stencil_cheb_available(c) && nb < stencil_cheb_partials_max(c)) { nb = stencil_cheb_partials_max(c); }
	c->max_blocks = nb;
	c->partial.alloc(sizeof(double) * 4 * nb);
	c->vectors_ready = true;
}

template <typename T>
void load_owned(fi_ctx* c, DevBuf& v, const float* src, int memory)
{
	const Geom& g = c->g;
	if (!src) {
		FI_HIP_TRY(hipMemsetAsync(v.p, 0, sizeof(T) * g.nloc, c->stream));
		return;
	}
	DevBuf tmp;
	const float* dsrc = src;
	if (memory == FI_HOST) {
		tmp.alloc(sizeof(float) * g.nown);
		FI_HIP_TRY(hipMemcpyAsync(tmp.p, src, sizeof(float) * g.nown, hipMemcpyHostToDevice, c->stream));
		dsrc = tmp.as<float>();
	}
	hipLaunchKernelGGL((k_from_float<T>), dim3(blocks_for(g.nown)), dim3(kThreads), 0, c->stream, g.nown, dsrc,
	                   owned<T>(c, v));
	FI_HIP_TRY(hipGetLastError());
	FI_HIP_TRY(hipStreamSynchronize(c->stream));
}

template <typename T>
void store_owned(fi_ctx* c, const DevBuf& v, float* dst, int memory)
{
	const Geom& g = c->g;
	if (!dst) { return; }
	if (memory == FI_DEVICE) {
		hipLaunchKernelGGL((k_to_float<T>), dim3(blocks_for(g.nown)), dim3(kThreads), 0, c->stream, g.nown,
		                   owned<T>(c, v), dst);
		FI_HIP_TRY(hipGetLastError());
		FI_HIP_TRY(hipStreamSynchronize(c->stream));
		return;
	}
	DevBuf tmp;
	tmp.alloc(sizeof(float) * g.nown);
	hipLaunchKernelGGL((k_to_float<T>), dim3(blocks_for(g.nown)), dim3(kThreads), 0, c->stream, g.nown, owned<T>(c, v),
	                   tmp.as<float>());
	FI_HIP_TRY(hipGetLastError());
	FI_HIP_TRY(hipMemcpyAsync(dst, tmp.p, sizeof(float) * g.nown, hipMemcpyDeviceToHost, c->stream));
	FI_HIP_TRY(hipStreamSynchronize(c->stream));
}

void halo_exchange(RankSet& R, DevBuf fi_ctx::*vec, int width)  // width planes next to the slabs (0: the stencil's reach)
{
	if (R[0]->nranks == 1) { return; }  // whole lattices (a loop-back group's copies of the replicated tail included)
	++R[0]->n_halo_exchanges;
	if (R.size() == 1) {
		exchange_halo(R[0], (R[0]->*vec).p, width);
		return;
	}
	for (size_t i = 0; i + 1 < R.size(); ++i) {
		fi_ctx* lo = R[i];
		fi_ctx* hi = R[i + 1];
		const Geom& gl = lo->g;
		const Geom& gh = hi->g;
		const int    L     = gl.ndim - 1;
		const int    H     = width > 0 ? width : lo->reach;
		const size_t es    = elem_size(lo);
		const size_t plane = static_cast<size_t>(gl.stride[L]);
		const size_t bytes = es * plane * H;
		char* lo_base = static_cast<char*>((lo->*vec).p);
		char* hi_base = static_cast<char*>((hi->*vec).p);
		// lo's last H owned planes -> hi's lower ghost planes (the ones next to its slab)
		FI_HIP_TRY(hipMemcpyAsync(hi_base + es * plane * (gh.own_lo[L] - H), lo_base + es * plane * (gl.own_hi[L] - H), bytes,
		                          hipMemcpyDeviceToDevice, lo->stream));
		// hi's first H owned planes -> lo's upper ghost planes
		FI_HIP_TRY(hipMemcpyAsync(lo_base + es * plane * gl.own_hi[L], hi_base + es * plane * gh.own_lo[L], bytes,
		                          hipMemcpyDeviceToDevice, lo->stream));
	}
}

// ---- one slab per process: the exchange of the ghost planes overlaps the interior of the apply -------------------
// The ghost planes of `v` travel on the context's communication stream (RCCL grouped send / recv) while the workgroups
// of the marching kernel that read none of them run on the solver stream; the first and last z-chunk follow when
// the planes have arrived.  Returns false (nothing launched) where the apply is not one marching launch over all
// workgroups -- the caller then exchanges first and applies in one go.
bool overlap_possible(const fi_ctx* c)
{
	return c->nranks > 1 && c->march.valid && !c->march.wide && c->march.n_inner > 0 && !c->any_trip && c->generic.ntrip == 0 && c->tile_ts == 0 &&
	       (c->cells.ncell == 0 || cells_fused(c)) && !test_switch("FI_NO_OVERLAP");
}
void exchange_begin(fi_ctx* c, void* v)
{
	if (!c->comm_stream) {
		c->comm_stream = stream_take();
		FI_HIP_TRY(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
		FI_HIP_TRY(hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
	}
	FI_HIP_TRY(hipEventRecord(c->ev_ready, c->stream));            // v is complete behind everything enqueued so far
	FI_HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_ready, 0));
	exchange_halo_on(c, v, c->comm_stream);
	FI_HIP_TRY(hipEventRecord(c->ev_halo, c->comm_stream));
}
void exchange_wait(fi_ctx* c) { FI_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_halo, 0)); }

// y = AtA x with the halo exchange of x, for every member of a rank set
void apply_exchanged(RankSet& R, DevBuf fi_ctx::*in, DevBuf fi_ctx::*out, double* (*partials_of)(fi_ctx*))
{
	fi_ctx* c0 = R[0];
	if (R.size() == 1 && overlap_possible(c0)) {
		fi_ctx* c = c0;
		double* part = partials_of ? partials_of(c) : nullptr;
		exchange_begin(c, (c->*in).p);
		if (stencil_apply_part(c, (c->*in).p, (c->*out).p, part, 1)) {
			exchange_wait(c);
			stencil_apply_part(c, (c->*in).p, (c->*out).p, part, 2);
			FI_HIP_TRY(hipGetLastError());
			return;
		}
		exchange_wait(c);
		apply_AtA(c, (c->*in).p, (c->*out).p, part);
		return;
	}
	halo_exchange(R, in);
	for (fi_ctx* c : R) { apply_AtA(c, (c->*in).p, (c->*out).p, partials_of ? partials_of(c) : nullptr); }
}

// partial sums -> sums[] on every rank, summed over ranks, then the scalar recurrences of `phase`
// (phase < 0: no recurrences).  `count_of(c)` partials per vector, vectors `stride_of(c)` apart.
void reset_scalars(RankSet& R, const CgScalars& init)
{
	for (fi_ctx* c : R) {
		FI_HIP_TRY(hipMemcpyAsync(c->scal.p, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
	}
}

// The wall-clock guard of a solve reads every rank's OWN clock.  With one slab per process the ranks must leave the
// loop in the same poll round -- a rank that went on alone would enqueue halo exchanges and all-reduces that have no
// partner and hang in them -- so the flag is all-reduced at every look (slot 2, sums[3]: no reduction of the drivers
// uses more than three sums).  Every other stop condition comes from all-reduced device scalars and is agreed anyway.
bool timed_out_anywhere(RankSet& R, bool mine)
{
	fi_ctx* c0 = R[0];
	if (!(R.size() == 1 && c0->nranks > 1)) { return mine; }
	double* flag = (c0->scal.as<CgScalars>() + 2)->sums + 3;
	double  v = mine ? 1.0 : 0.0;
	FI_HIP_TRY(hipMemcpyAsync(flag, &v, sizeof(double), hipMemcpyHostToDevice, c0->stream));
	allreduce_sum(c0, flag, 1);
	FI_HIP_TRY(hipMemcpyAsync(&v, flag, sizeof(double), hipMemcpyDeviceToHost, c0->stream));
	FI_HIP_TRY(hipStreamSynchronize(c0->stream));
	return v > 0.0;
}

// One slab per process: do all ranks report success?  (An all-reduce of a flag, like timed_out_anywhere.)  A rank that
// failed before a collective would leave its peers waiting in it forever -- RCCL has no timeout.
bool all_ranks_ok(fi_ctx* c, bool mine)
{
	if (!(c->nranks > 1 && comm_ready(c))) { return mine; }
	double* flag = (c->scal.as<CgScalars>() + 2)->sums + 3;
	double  v = mine ? 0.0 : 1.0;
	FI_HIP_TRY(hipMemcpyAsync(flag, &v, sizeof(double), hipMemcpyHostToDevice, c->stream));
	allreduce_sum(c, flag, 1);
	FI_HIP_TRY(hipMemcpyAsync(&v, flag, sizeof(double), hipMemcpyDeviceToHost, c->stream));
	FI_HIP_TRY(hipStreamSynchronize(c->stream));
	return v == 0.0;
}

// Jacobi-PCG over a rank set; x of every member holds the guess on entry and the solution on return.
template <typename T>
void cg_run(RankSet& R, int max_iterations, float tol)
{
	fi_ctx* c0 = R[0];
	hipStream_t st = c0->stream;
	if (max_iterations <= 0) {
		const int64_t dflt = 2 * static_cast<int64_t>(c0->g.gn[0]) * c0->g.gn[1] * c0->g.gn[2];  // Eigen: 2 * cols
		max_iterations = dflt > std::numeric_limits<int>::max() ? std::numeric_limits<int>::max() : static_cast<int>(dflt);
	}
	const double tolerance = tol > 0 ? static_cast<double>(tol) : static_cast<double>(std::numeric_limits<float>::epsilon());

	// A level of a coarse-to-fine start that has just been solved WITHOUT a look at its stop flag (below): what the flag
	// said comes in now -- the copy went to pinned memory behind that solve's last kernel, long since done.
	if (c0->unwatched_pending) {
		c0->unwatched_pending = false;
		FI_HIP_TRY(hipEventSynchronize(c0->ev_unwatched));
		const CgScalars* was = static_cast<const CgScalars*>(pinned(c0, 2, sizeof(CgScalars)));
		const bool as_expected = (was->done == 1 || was->done == 5) && was->iter > 0 && was->iter <= c0->unwatched_expected;
		c0->last_cg_iterations = as_expected ? was->iter : 0;  // (0: this solve watches its flag again and learns the new count)
		c0->pred_recalled[0]   = true;
		c0->cg_count_recalled  = false;
	}
	// Such a solve: a coarser level of an undivided lattice whose previous solve ended after n iterations with the same
	// tolerance gets n iterations and no look at all -- the level below it is waiting for its result, and a look is a host
	// round trip of ~40 us per level.  Should n not have been enough this time, the start guess is that much worse and the
	// next solve of this level watches again.
	// A count recalled from ANOTHER context (same lattice, model, level and tolerance -- the key knows nothing of the data)
	// only schedules this solve's first look at the flag: the solve is watched, so what a fresh context computes never
	// depends on what the process solved before.  "No look at all" is for a context's own history.
	if (c0->level > 0 && R.size() == 1 && c0->last_cg_iterations == 0) {  // a fresh context: what the one before it learnt
		const int n = recall_iterations(c0, 0, tolerance);
		if (n > 0) {
			c0->last_cg_iterations = n;
			c0->last_cg_tol        = tolerance;
			c0->cg_count_recalled  = true;
		}
	}
	const bool unwatched = c0->level > 0 && R.size() == 1 && c0->nranks == 1 && !c0->verify_residual && c0->last_cg_iterations > 0 && !c0->cg_count_recalled &&
	                       c0->last_cg_iterations <= max_iterations && c0->last_cg_iterations <= 64 && c0->last_cg_tol == tolerance && !test_switch("FI_LOOK_ALWAYS");
	c0->last_cg_tol = tolerance;

	EventPair timer;  // (destroyed on every way out: a coarse level's breakdown, a timeout)
	const hipEvent_t e0 = timer.e0, e1 = timer.e1;
	if (!unwatched) { FI_HIP_TRY(hipEventRecord(e0, st)); }

	CgScalars init{};
	init.tol2     = tolerance * tolerance;
	init.max_iter = max_iterations;
	reset_scalars(R, init);

	auto nbv       = [](fi_ctx* c) { return stream_blocks(c->g.nown); };
	auto nb_apply  = [](fi_ctx* c) { return apply_num_partials(c); };
	auto zero      = [](fi_ctx*) { return 0; };

	// r0 = b - A x0
	halo_exchange(R, &fi_ctx::x);
	for (fi_ctx* c : R) { apply_AtA(c, c->x.p, c->q.p, nullptr); }
	for (fi_ctx* c : R) {
		const int64_t o = c->g.own_first;
		hipLaunchKernelGGL((k_cg_init<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, c->atb.as<T>() + o,
		                   c->q.as<T>() + o, c->dinv.as<T>() + o, c->r.as<T>() + o, c->p.as<T>() + o,
		                   c->partial.as<double>(), nbv(c));
	}
	reduce_phase(R, 3, nbv, nbv, kPhaseInit);

	int samples = 0;
	while (static_cast<int>(c0->ev.size()) < 2 * kMaxSamples) {
		hipEvent_t e;
		FI_HIP_TRY(hipEventCreate(&e));
		c0->ev.push_back(e);
	}
	// wall-clock guard: a solve that cannot reach its tolerance (fp32 stagnation with the default 2N
	// iteration cap) must not hold the GPU for hours.  FI_SOLVE_TIMEOUT_S overrides the 600 s default.
	double limit_s = 600.0;
	if (const char* env = getenv("FI_SOLVE_TIMEOUT_S")) { limit_s = atof(env); }
	const auto wall0 = std::chrono::steady_clock::now();
	bool timed_out = false;
	CgScalars* sc0 = c0->scal.as<CgScalars>();
	auto vec_ok = [](fi_ctx* c) {
		constexpr int N = Vec16<T>::N;
		return (c->g.own_first % N == 0) && (c->g.nown % N == 0);
	};
	// one context, one process: the dot-product reductions are folded into the vector kernels (3 launches per step)
	const bool folded = R.size() == 1 && c0->nranks == 1 && !tuning_switch("FI_NO_FOLD");
	const bool folded_set = !folded && !tuning_switch("FI_NO_FOLD");  // slabs: the same kernels behind a reduction over the rank set
	int issued = 0;         // CG steps enqueued so far (the device runs step k only while it is not done)
	int restarts_left = c0->verify_residual ? 3 : 0;
	// Coarser levels of a coarse-to-fine start (launches of 2-25 us: the host issues them no faster than the GPU retires
	// them): the first look at the stop flag comes when this level's previous solve had finished -- a re-assembled problem
	// changes little -- and the look right behind the initialisation is skipped (kernels of a finished solve exit at once).
	// Config 4's three levels: 48 iterations issued for 22 needed -> 25, four looks less per solve.
	int looks = 0;
	for (;;) {
		const bool skip_look = looks == 0 && c0->level > 0 && c0->last_cg_iterations > 0;
		if (unwatched && looks == 1) {  // the predicted iterations are on the stream: the flag travels to pinned memory, nobody waits
			if (!c0->ev_unwatched) { FI_HIP_TRY(hipEventCreateWithFlags(&c0->ev_unwatched, hipEventDisableTiming)); }
			FI_HIP_TRY(hipMemcpyAsync(pinned(c0, 2, sizeof(CgScalars)), sc0, sizeof(CgScalars), hipMemcpyDeviceToHost, st));
			FI_HIP_TRY(hipEventRecord(c0->ev_unwatched, st));
			c0->unwatched_pending  = true;
			c0->unwatched_expected = c0->last_cg_iterations;
			for (fi_ctx* c : R) {  // (what the previous solve reported: the caller reads the iteration count and `converged`)
				c->stats.iterations = c0->last_cg_iterations;
				c->stats.converged  = 1;
				c->stats.operator_applies = c0->last_cg_iterations + 1;
				c->stats.restarts   = 0;
				c->stats.solve_ms   = 0.0;
				c->stats.spmv_samples = 0;
			}
			return;
		}
		++looks;
		if (!skip_look) {
			FI_HIP_TRY(hipMemcpyAsync(c0->scal_host, sc0, sizeof(CgScalars), hipMemcpyDeviceToHost, st));
			FI_HIP_TRY(hipStreamSynchronize(st));
		} else {
			c0->scal_host->done = 0;
		}
		if (c0->scal_host->done) {
			// The recurrence residual met the tolerance.  In fp32 it drifts away from b - A x over hundreds of
			// steps, so the true residual is evaluated once; if it misses the tolerance CG restarts from it
			// (residual replacement).  Steps enqueued past the stop did nothing: resynchronise the numbering.
			if (c0->scal_host->done != 1 || restarts_left <= 0) { break; }
			--restarts_left;
			issued = c0->scal_host->iter;
			for (fi_ctx* c : R) {  // the apply kernels exit at once while the flag is up
				hipLaunchKernelGGL(k_set_done, dim3(1), dim3(1), 0, c->stream, c->scal.as<CgScalars>(), 0);
			}
			halo_exchange(R, &fi_ctx::x);
			for (fi_ctx* c : R) { apply_AtA(c, c->x.p, c->q.p, nullptr); }
			for (fi_ctx* c : R) {
				const int64_t o = c->g.own_first;
				hipLaunchKernelGGL((k_cg_init<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, c->atb.as<T>() + o,
				                   c->q.as<T>() + o, c->dinv.as<T>() + o, c->r.as<T>() + o, c->p.as<T>() + o,
				                   c->partial.as<double>(), nbv(c));
			}
			reduce_phase(R, 2, nbv, nbv, kPhaseRestart);
			continue;
		}
		if (timed_out_anywhere(R, std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() > limit_s)) {
			timed_out = true;
			break;
		}
		// (a coarser level of a coarse-to-fine start is done within a few steps: shorter bursts between two looks at the flag
		// leave fewer launches behind that only find the flag up -- 31 of 48 iterations of config 4's three levels)
		int burst = kCheckEvery;
		if (c0->level > 0) {
			burst = kCheckEvery / 2;
			if (c0->last_cg_iterations > 0) { burst = looks == 1 ? (c0->last_cg_iterations < 64 ? c0->last_cg_iterations : 64) : 2; }
		}
		for (int k = 0; k < burst; ++k) {
			++issued;
			// every 4th apply is timed: an event record is a barrier packet of its own in the queue
			const bool sample = c0->level == 0 && samples < 8 && (issued & 3) == 1;  // event pairs cost the stream ~11 us each
			if (sample) { FI_HIP_TRY(hipEventRecord(c0->ev[2 * samples], st)); }
			apply_exchanged(R, &fi_ctx::p, &fi_ctx::q, +[](fi_ctx* c) -> double* { return c->partial.as<double>(); });
			if (sample) {
				FI_HIP_TRY(hipEventRecord(c0->ev[2 * samples + 1], st));
				++samples;
			}
			if (folded) {
				fi_ctx* c = c0;
				const int64_t o   = c->g.own_first;
				const int     nbf = nbv(c) > 1024 ? 1024 : nbv(c);  // every block re-reads all partials: keep them few
				CgScalars*    sc  = c->scal.as<CgScalars>();
				double*       pp  = c->partial.as<double>();
				double*       pr  = pp + c->max_blocks;  // the apply partials are still being read: separate region
				if (vec_ok(c)) {
					hipLaunchKernelGGL((k_cg_resid_f<T, true>), dim3(nbf), dim3(kThreads), 0, st, c->g.nown, sc, sc + 1, issued, pp,
					                   nb_apply(c), c->q.as<T>() + o, c->dinv.as<T>() + o, c->r.as<T>() + o, pr, nbf);
					hipLaunchKernelGGL((k_cg_xp_f<T, true>), dim3(nbf), dim3(kThreads), 0, st, c->g.nown, sc + 1, sc, issued, pr, nbf,
					                   c->r.as<T>() + o, c->dinv.as<T>() + o, c->x.as<T>() + o, c->p.as<T>() + o);
				} else {
					hipLaunchKernelGGL((k_cg_resid_f<T, false>), dim3(nbf), dim3(kThreads), 0, st, c->g.nown, sc, sc + 1, issued, pp,
					                   nb_apply(c), c->q.as<T>() + o, c->dinv.as<T>() + o, c->r.as<T>() + o, pr, nbf);
					hipLaunchKernelGGL((k_cg_xp_f<T, false>), dim3(nbf), dim3(kThreads), 0, st, c->g.nown, sc + 1, sc, issued, pr, nbf,
					                   c->r.as<T>() + o, c->dinv.as<T>() + o, c->x.as<T>() + o, c->p.as<T>() + o);
				}
				continue;
			}
			if (folded_set) {
				// rank sets: the folded kernels on every member, fed with the dot products summed over the slabs
				auto nbf_of = [&](fi_ctx* c) { return nbv(c) > 1024 ? 1024 : nbv(c); };
				reduce_to_slot2(R, 1, nb_apply, zero, +[](fi_ctx* c) -> const double* { return c->partial.as<double>(); });
				for (fi_ctx* c : R) {
					const int64_t o   = c->g.own_first;
					const int     nbf = nbf_of(c);
					CgScalars*    sc  = c->scal.as<CgScalars>();
					double*       pr  = c->partial.as<double>() + c->max_blocks;
					if (vec_ok(c)) {
						hipLaunchKernelGGL((k_cg_resid_f<T, true>), dim3(nbf), dim3(kThreads), 0, c->stream, c->g.nown, sc, sc + 1, issued,
						                   (sc + 2)->sums, 1, c->q.as<T>() + o, c->dinv.as<T>() + o, c->r.as<T>() + o, pr, nbf);
					} else {
						hipLaunchKernelGGL((k_cg_resid_f<T, false>), dim3(nbf), dim3(kThreads), 0, c->stream, c->g.nown, sc, sc + 1, issued,
						                   (sc + 2)->sums, 1, c->q.as<T>() + o, c->dinv.as<T>() + o, c->r.as<T>() + o, pr, nbf);
					}
				}
				reduce_to_slot2(R, 2, nbf_of, nbf_of,
				                +[](fi_ctx* c) -> const double* { return c->partial.as<double>() + c->max_blocks; });
				for (fi_ctx* c : R) {
					const int64_t o   = c->g.own_first;
					const int     nbf = nbf_of(c);
					CgScalars*    sc  = c->scal.as<CgScalars>();
					if (vec_ok(c)) {
						hipLaunchKernelGGL((k_cg_xp_f<T, true>), dim3(nbf), dim3(kThreads), 0, c->stream, c->g.nown, sc + 1, sc, issued,
						                   (sc + 2)->sums, 1, c->r.as<T>() + o, c->dinv.as<T>() + o, c->x.as<T>() + o, c->p.as<T>() + o);
					} else {
						hipLaunchKernelGGL((k_cg_xp_f<T, false>), dim3(nbf), dim3(kThreads), 0, c->stream, c->g.nown, sc + 1, sc, issued,
						                   (sc + 2)->sums, 1, c->r.as<T>() + o, c->dinv.as<T>() + o, c->x.as<T>() + o, c->p.as<T>() + o);
					}
				}
				continue;
			}
			reduce_phase(R, 1, nb_apply, zero, kPhaseSpmv);
			for (fi_ctx* c : R) {
				const int64_t o = c->g.own_first;
				if (vec_ok(c)) {
					hipLaunchKernelGGL((k_cg_resid<T, true>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown,
					                   c->scal.as<CgScalars>(), c->q.as<T>() + o, c->dinv.as<T>() + o, c->r.as<T>() + o,
					                   c->partial.as<double>(), nbv(c));
				} else {
					hipLaunchKernelGGL((k_cg_resid<T, false>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown,
					                   c->scal.as<CgScalars>(), c->q.as<T>() + o, c->dinv.as<T>() + o, c->r.as<T>() + o,
					                   c->partial.as<double>(), nbv(c));
				}
			}
			reduce_phase(R, 2, nbv, nbv, kPhaseUpdate);
			for (fi_ctx* c : R) {
				const int64_t o = c->g.own_first;
				if (vec_ok(c)) {
					hipLaunchKernelGGL((k_cg_xp<T, true>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown,
					                   c->scal.as<CgScalars>(), issued, c->r.as<T>() + o, c->dinv.as<T>() + o,
					                   c->x.as<T>() + o, c->p.as<T>() + o);
				} else {
					hipLaunchKernelGGL((k_cg_xp<T, false>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown,
					                   c->scal.as<CgScalars>(), issued, c->r.as<T>() + o, c->dinv.as<T>() + o,
					                   c->x.as<T>() + o, c->p.as<T>() + o);
				}
			}
		}
		FI_HIP_TRY(hipGetLastError());
	}
	FI_HIP_TRY(hipEventRecord(e1, st));
	FI_HIP_TRY(hipEventSynchronize(e1));
	float ms = 0;
	FI_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));

	const CgScalars h = *c0->scal_host;
	int used = samples;  // samples of iterations that actually ran (kernels of later iterations exited on the flag)
	if ((h.iter + 3) / 4 < used) { used = (h.iter + 3) / 4; }  // sample k belongs to iteration 4k + 1
	double sum_ms = 0;
	for (int k = 0; k < used; ++k) {
		float t = 0;
		FI_HIP_TRY(hipEventElapsedTime(&t, c0->ev[2 * k], c0->ev[2 * k + 1]));
		sum_ms += t;
	}
	for (fi_ctx* c : R) {
		c->stats.spmv_samples = used;
		c->stats.spmv_ms_avg  = used ? sum_ms / used : 0.0;
		c->stats.spmv_bytes   = apply_algorithmic_bytes(c);
		c->stats.prec_samples = 0;
		c->stats.prec_ms_avg  = 0.0;
		c->stats.prec_bytes   = 0.0;
		c->stats.operator_applies = h.iter + 1 + h.restarts;
		c->stats.solve_ms     = ms;
		c->stats.iterations   = h.iter;
		c->last_cg_iterations = timed_out ? 0 : h.iter;  // the same on every rank: the scalars are sums over all of them
		c->cg_count_recalled  = false;                   // (this context's own history from here on)
		if (R.size() == 1 && c->level > 0 && !timed_out && (h.done == 1 || h.done == 5)) { remember_iterations(c, 0, tolerance, h.iter); }
		// with the verified stop on, "converged" means b - A x itself met the tolerance (done == 5); a recurrence
		// that converged while the true residual stagnated above it (fp32 on an ill-conditioned system) is not
		c->stats.converged    = (!timed_out && (h.done == 4 || h.done == 5 || (h.done == 1 && !c0->verify_residual))) ? 1 : 0;
		c->stats.rel_residual = h.bb > 0 ? std::sqrt(h.rr / h.bb) : 0.0;
		c->stats.restarts     = h.restarts;
		c->stats.verified_residual = (h.restarts > 0 && h.bb > 0) ? std::sqrt(h.true_rr / h.bb) : -1.0;
		if (h.done == 4) {  // rhs == 0  ->  x = 0 (Eigen's early return)
			FI_HIP_TRY(hipMemsetAsync(c->x.p, 0, sizeof(T) * c->g.nloc, c->stream));
		}
	}
	FI_REQUIRE(h.done != 2, FI_ERR_BREAKDOWN, "CG breakdown: non-finite or non-positive curvature (p.AtA p = %g)", h.pq);
	FI_REQUIRE(!timed_out, FI_ERR_TIMEOUT, "solve stopped by the wall-clock guard (FI_SOLVE_TIMEOUT_S = %g s) after %d iterations, "
	           "relative residual %g", limit_s, h.iter, h.bb > 0 ? std::sqrt(h.rr / h.bb) : 0.0);
}

template <typename T>
void solve_cg_t(fi_ctx* c, const float* guess, int max_iterations, float tol, float* out, int* iterations,
                float* rel_residual, int memory)
{
	ensure_vectors(c);
	c->stats.coarse_iterations = 0;
	RankSet R{c};
	struct Report {
		fi_ctx* c; int* it; float* rel;
		~Report() { if (it) { *it = c->stats.iterations; } if (rel) { *rel = static_cast<float>(c->stats.rel_residual); } }
	} report{c, iterations, rel_residual};
	if (!guess && c->twin && c->twin->coarse) {
		twin_cascade_guess(R);  // coarse-to-fine start on the fp32 replica, widened
	} else if (!guess && c->coarse) {
		cascade_guess<T>(R);
	} else {
		load_owned<T>(c, c->x, guess, memory);
	}
	if (test_switch("FI_START_ONLY")) {  // (tests: the start guess itself, no iteration on the finest level)
		c->stats.iterations = 0;
		store_owned<T>(c, c->x, out, memory);
		return;
	}
	if (c->mg_mode == 1 && (c->coarse || (c->twin && c->twin->coarse))) {
		c->predictable_start = !guess;
		cg_run_mg<T>(R, max_iterations, tol);
	} else if (poly_ok(c)) {
		cg_run_poly_or_jacobi<T>(R, max_iterations, tol);
	} else {
		cg_run<T>(R, max_iterations, tol);
	}
	store_owned<T>(c, c->x, out, memory);
}

// rhs of the tile systems: b - 2 (AtA g - B g), B = same-tile entries of AtA.  `bg` holds (B + 1e-6 I) g.
// The factor 2 is the reference's: tile_solver_square visits every stored off-tile entry of the symmetric
// matrix -- (i, j) and (j, i) -- and moves it to BOTH rows' right-hand sides (sparse_linear.cpp:327-334).
template <typename T>
__global__ __launch_bounds__(kThreads) void k_tile_rhs(int64_t n, const T* __restrict__ b, const T* __restrict__ ag,
                                                        const T* __restrict__ bg, const T* __restrict__ g,
                                                        T* __restrict__ out)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		out[i] = b[i] - T(2) * (ag[i] - (bg[i] - T(1e-6f) * g[i]));
	}
}

// tile_solver_square (sparse_linear.cpp:246-390) as one CG solve of the block-diagonal tile operator: every
// tile is an independent SPD system ((AtA restricted to the tile) + 1e-6 I) x_t = rhs_t, so CG on the whole
// lattice solves all of them at once; the couplings to other tiles enter through the guess, as in the reference.
template <typename T>
void tile_pass_run(RankSet& R, int tile_size)  // x of every member: the guess on entry, the tile solutions on return
{
	CgScalars init{};
	reset_scalars(R, init);
	halo_exchange(R, &fi_ctx::x);
	struct Restore {
		RankSet& R;
		std::vector<bool> swapped;
		~Restore()
		{
			for (size_t i = 0; i < R.size(); ++i) {
				fi_ctx* c = R[i];
				c->tile_ts = 0;
				if (swapped[i]) {
					c->atb.swap(c->scratch[21]);
				}
			}
		}
	} restore{R, std::vector<bool>(R.size(), false)};
	for (size_t i = 0; i < R.size(); ++i) {
		fi_ctx* c = R[i];
		DevBuf& rhs = c->scratch[21];
		rhs.alloc(elem_size(c) * c->g.nloc);
		FI_HIP_TRY(hipMemsetAsync(rhs.p, 0, elem_size(c) * c->g.nloc, c->stream));
		apply_AtA(c, c->x.p, c->q.p, nullptr);
		c->tile_ts = tile_size;
		apply_AtA(c, c->x.p, c->r.p, nullptr);
		const int64_t o = c->g.own_first;
		hipLaunchKernelGGL((k_tile_rhs<T>), dim3(stream_blocks(c->g.nown)), dim3(kThreads), 0, c->stream, c->g.nown,
		                   c->atb.as<T>() + o, c->q.as<T>() + o, c->r.as<T>() + o, c->x.as<T>() + o, rhs.as<T>() + o);
		FI_HIP_TRY(hipGetLastError());
		c->atb.swap(rhs);
		restore.swapped[i] = true;
	}
	// contexts of materialised rows only (the drop-in's solve_tiled_with_guess): tiles without any entry keep the guess
	std::vector<DevBuf*> kept(R.size(), nullptr);
	for (size_t i = 0; i < R.size(); ++i) {
		fi_ctx* c = R[i];
		const fi_weights& w = c->w;
		const bool rows_only = c->generic.nnz > 0 && c->cells.ncell == 0 && c->nranks == 1 && !(w.model_0 > 0) && !(w.model_1 > 0) &&
		                       !(w.model_2 > 0) && !(w.model_3 > 0) && !(w.model_4 > 0) && !(w.gradient_smoothness > 0);
		if (rows_only) {
			DevBuf& g0 = c->scratch[22];
			g0.alloc(elem_size(c) * c->g.nloc);
			FI_HIP_TRY(hipMemcpyAsync(g0.p, c->x.p, elem_size(c) * c->g.nloc, hipMemcpyDeviceToDevice, c->stream));
			kept[i] = &g0;
		}
	}
	cg_run<T>(R, 4000, sizeof(T) == 8 ? 1e-12f : 1e-6f);  // the reference factorises: iterate to the precision's floor
	for (size_t i = 0; i < R.size(); ++i) {
		if (kept[i]) { generic_keep_guess_in_empty_tiles(R[i], tile_size, kept[i]->p, R[i]->x.p); }
	}
}

template <typename T>
void tile_pass_t(fi_ctx* c, const float* guess, int tile_size, float* out, int memory)
{
	ensure_vectors(c);
	load_owned<T>(c, c->x, guess, memory);
	RankSet R{c};
	tile_pass_run<T>(R, tile_size);
	store_owned<T>(c, c->x, out, memory);
}

template <typename T>
void jacobi_run(RankSet& R, int sweeps, float weight)
{
	CgScalars init{};
	reset_scalars(R, init);
	for (int s = 0; s < sweeps; ++s) {
		halo_exchange(R, &fi_ctx::x);
		for (fi_ctx* c : R) { apply_AtA(c, c->x.p, c->q.p, nullptr); }
		for (fi_ctx* c : R) {
			const int64_t o = c->g.own_first;
			hipLaunchKernelGGL((k_jacobi_update<T>), dim3(stream_blocks(c->g.nown)), dim3(kThreads), 0, c->stream, c->g.nown,
			                   static_cast<T>(weight), c->atb.as<T>() + o, c->q.as<T>() + o, c->dinv.as<T>() + o,
			                   c->x.as<T>() + o);
		}
	}
	FI_HIP_TRY(hipGetLastError());
}

template <typename T>
void jacobi_t(fi_ctx* c, const float* guess, int sweeps, float weight, float* out, int memory)
{
	ensure_vectors(c);
	load_owned<T>(c, c->x, guess, memory);
	RankSet R{c};
	jacobi_run<T>(R, sweeps, weight);
	store_owned<T>(c, c->x, out, memory);
}

// ||Atb - AtA x|| / ||Atb|| of the current x, evaluated on the device
template <typename T>
double true_residual_run(RankSet& R)
{
	CgScalars init{};
	reset_scalars(R, init);
	halo_exchange(R, &fi_ctx::x);
	for (fi_ctx* c : R) { apply_AtA(c, c->x.p, c->q.p, nullptr); }
	auto nbv = [](fi_ctx* c) { return stream_blocks(c->g.nown); };
	for (fi_ctx* c : R) {
		const int64_t o = c->g.own_first;
		hipLaunchKernelGGL((k_residual_norm<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, c->atb.as<T>() + o,
		                   c->q.as<T>() + o, c->partial.as<double>(), nbv(c));
	}
	reduce_phase(R, 2, nbv, nbv, -1);
	fi_ctx* c0 = R[0];
	FI_HIP_TRY(hipMemcpyAsync(c0->scal_host, c0->scal.p, sizeof(CgScalars), hipMemcpyDeviceToHost, c0->stream));
	FI_HIP_TRY(hipStreamSynchronize(c0->stream));
	const double rr = c0->scal_host->sums[0], bb = c0->scal_host->sums[1];
	return bb > 0 ? std::sqrt(rr / bb) : std::sqrt(rr);
}

template <typename T>
double true_residual_t(fi_ctx* c)
{
	ensure_vectors(c);
	RankSet R{c};
	return true_residual_run<T>(R);
}

// y = AtA x with fp64 host vectors holding, rank after rank, the owned unknowns of every member
template <typename T>
void apply_f64_run(RankSet& R, const double* xin, double* yout)
{
	CgScalars init{};
	reset_scalars(R, init);
	std::vector<DevBuf> tmp(R.size());
	int64_t at = 0;
	for (size_t i = 0; i < R.size(); ++i) {
		fi_ctx* c = R[i];
		const Geom& g = c->g;
		tmp[i].alloc(sizeof(double) * g.nown);
		FI_HIP_TRY(hipMemcpyAsync(tmp[i].p, xin + at, sizeof(double) * g.nown, hipMemcpyHostToDevice, c->stream));
		FI_HIP_TRY(hipMemsetAsync(c->p.p, 0, sizeof(T) * g.nloc, c->stream));
		hipLaunchKernelGGL((k_convert<double, T>), dim3(blocks_for(g.nown)), dim3(kThreads), 0, c->stream, g.nown,
		                   tmp[i].as<double>(), owned<T>(c, c->p));
		at += g.nown;
	}
	halo_exchange(R, &fi_ctx::p);
	for (fi_ctx* c : R) { apply_AtA(c, c->p.p, c->q.p, nullptr); }
	at = 0;
	for (size_t i = 0; i < R.size(); ++i) {
		fi_ctx* c = R[i];
		const Geom& g = c->g;
		hipLaunchKernelGGL((k_convert<T, double>), dim3(blocks_for(g.nown)), dim3(kThreads), 0, c->stream, g.nown,
		                   owned<T>(c, c->q), tmp[i].as<double>());
		FI_HIP_TRY(hipMemcpyAsync(yout + at, tmp[i].p, sizeof(double) * g.nown, hipMemcpyDeviceToHost, c->stream));
		at += g.nown;
	}
	FI_HIP_TRY(hipGetLastError());
	FI_HIP_TRY(hipStreamSynchronize(R[0]->stream));
}

template <typename T>
void apply_f64_t(fi_ctx* c, const double* xin, double* yout)
{
	ensure_vectors(c);
	RankSet R{c};
	apply_f64_run<T>(R, xin, yout);
}

template <typename T>
void get_vec_f64_t(fi_ctx* c, const DevBuf& v, double* out)
{
	const Geom& g = c->g;
	DevBuf tmp;
	tmp.alloc(sizeof(double) * g.nown);
	hipLaunchKernelGGL((k_convert<T, double>), dim3(blocks_for(g.nown)), dim3(kThreads), 0, c->stream, g.nown,
	                   owned<T>(c, v), tmp.as<double>());
	FI_HIP_TRY(hipGetLastError());
	FI_HIP_TRY(hipMemcpyAsync(out, tmp.p, sizeof(double) * g.nown, hipMemcpyDeviceToHost, c->stream));
	FI_HIP_TRY(hipStreamSynchronize(c->stream));
}


// ---- explicit instantiations (declared in fi_solver_internal.h) ----
template void load_owned<float>(fi_ctx*, DevBuf&, const float*, int);
template void load_owned<double>(fi_ctx*, DevBuf&, const float*, int);
template void store_owned<float>(fi_ctx*, const DevBuf&, float*, int);
template void store_owned<double>(fi_ctx*, const DevBuf&, float*, int);
template void cg_run<float>(RankSet&, int, float);
template void cg_run<double>(RankSet&, int, float);
template void solve_cg_t<float>(fi_ctx*, const float*, int, float, float*, int*, float*, int);
template void solve_cg_t<double>(fi_ctx*, const float*, int, float, float*, int*, float*, int);
template void tile_pass_run<float>(RankSet&, int);
template void tile_pass_run<double>(RankSet&, int);
template void tile_pass_t<float>(fi_ctx*, const float*, int, float*, int);
template void tile_pass_t<double>(fi_ctx*, const float*, int, float*, int);
template void jacobi_run<float>(RankSet&, int, float);
template void jacobi_run<double>(RankSet&, int, float);
template void jacobi_t<float>(fi_ctx*, const float*, int, float, float*, int);
template void jacobi_t<double>(fi_ctx*, const float*, int, float, float*, int);
template double true_residual_run<float>(RankSet&);
template double true_residual_run<double>(RankSet&);
template double true_residual_t<float>(fi_ctx*);
template double true_residual_t<double>(fi_ctx*);
template void apply_f64_run<float>(RankSet&, const double*, double*);
template void apply_f64_run<double>(RankSet&, const double*, double*);
template void apply_f64_t<float>(fi_ctx*, const double*, double*);
template void apply_f64_t<double>(fi_ctx*, const double*, double*);
template void get_vec_f64_t<float>(fi_ctx*, const DevBuf&, double*);
template void get_vec_f64_t<double>(fi_ctx*, const DevBuf&, double*);

}  // namespace fi
