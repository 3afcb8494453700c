// fi_solver.hip -- Jacobi-preconditioned conjugate gradients and weighted Jacobi on A^T A x = A^T b,
// plus the C ABI (include/fi_hip.h).
//
// Reference path replaced: Eigen::BiCGSTAB<SparseMatrix<float>> with its default diagonal
// preconditioner, reached from solve_sparse_linear_with_guess (sparse_linear.cpp:186-212) and
// solve_tiled_with_guess (:427-440), and the hand-written loop of jacobi_iterations (:214-241).
// A^T A is symmetric positive semi-definite, so CG applies ("Planned CDNA4
// counterpart"); the stop rule is Eigen's: ||r||_2 <= tol * ||A^T b||_2.
//
// One CG iteration on one GPU = 3 launches on one stream:
//   apply         q = AtA p, per-block partials of p.q                      (fi_operator / fi_stencil / fi_stencil2d)
//   k_cg_resid_f  every block sums those partials (fixed order: bit-identical alpha in all blocks), r -= alpha q,
//                 partials of r.(Dinv r) and r.r
//   k_cg_xp_f     every block sums those, beta and the stop test; x += alpha p; p = Dinv r + beta p
// Block 0 publishes the scalars into the other of two scalar slots (`CgScalars[2]`): no block reads the slot its
// kernel writes.  Over slabs (loop-back group or one process per GPU) the sums cross ranks, so the reductions are
// launches of their own: k_reduce -> device sum / ncclAllReduce -> k_cg_logic, around k_cg_resid / k_cg_xp.
// Dot products: fp64 per-thread products, wave64 __shfl_down tree, LDS across the 4 waves, one partial per block,
// summed in a fixed order => bitwise reproducible run to run.  No host synchronisation inside an iteration:
// alpha/beta/flags stay in HBM; the host looks at the flag every `kCheckEvery` iterations; kernels of a finished
// solve exit at once.
//
// Also here: levels (coarser replicas, cascade start), the V-cycle preconditioner over rank sets, the mixed-
// precision solve (fp64 CG, fp32 V-cycle on a replica), the tile pre-solver, and the C ABI entry points.

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstdarg>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <thread>

#include "fi_internal.h"
#include "fi_transfer.h"
#include "fi_tail.h"

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

namespace {

constexpr int kThreads    = 256;
constexpr int kCheckEvery = 16;
constexpr int kMaxSamples = 128;
constexpr int kPolySamples = 3;  // per solve and kind, in the polynomial PCG (see there)

__device__ inline double wave_sum(double v)
{
	for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o, 64); }
	return v;
}

template <int NV>
__device__ inline void block_sum(double* v, double* out)  // out[] valid in thread 0
{
	__shared__ double s[NV][kThreads / 64];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int k = 0; k < NV; ++k) {
		const double w = wave_sum(v[k]);
		if (lane == 0) { s[k][wave] = w; }
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int k = 0; k < NV; ++k) {
			double r = 0;
			for (int w = 0; w < kThreads / 64; ++w) { r += s[k][w]; }
			out[k] = r;
		}
	}
	__syncthreads();
}

// two events around a timed region, destroyed on every way out of it
struct EventPair {
	hipEvent_t e0 = nullptr, e1 = nullptr;
	EventPair()
	{
		FI_HIP_TRY(hipEventCreate(&e0));
		if (hipEventCreate(&e1) != hipSuccess) {
			(void)hipEventDestroy(e0);
			e0 = nullptr;
			FI_HIP_TRY(hipErrorOutOfMemory);
		}
	}
	EventPair(const EventPair&) = delete;
	EventPair& operator=(const EventPair&) = delete;
	~EventPair()
	{
		if (e0) { (void)hipEventDestroy(e0); }
		if (e1) { (void)hipEventDestroy(e1); }
	}
};

inline int blocks_for(int64_t n) { return static_cast<int>((n + kThreads - 1) / kThreads); }

// grid-stride launch width for the streaming vector kernels: 256 CUs x 8 blocks
inline int stream_blocks(int64_t n)
{
	const int64_t b = (n + kThreads * 4 - 1) / (kThreads * 4);
	return static_cast<int>(b < 1 ? 1 : (b > 2048 ? 2048 : b));
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
template <typename T>
struct Vec16;
template <>
struct Vec16<float> {
	using V = float4;
	static constexpr int N = 4;
};
template <>
struct Vec16<double> {
	using V = double2;
	static constexpr int N = 2;
};

// 16-byte accesses with the non-temporal hint, for the streams of the folded CG kernels that nobody reads again
// soon (x, Dinv, q, and r where it is read last): the search direction p written by k_cg_xp_f then survives in the
// caches until the apply reads it -- config 4 at 256^3: apply inside CG 63.6 -> 55.1 us, the vector kernels +1.5 us.
// (The same hint on the apply's record loads costs 7 us: the four waves of a workgroup share those lines.)
template <typename T>
__device__ inline void ld16_nt(T* dst, const T* base, int64_t i)
{
	typedef T NV __attribute__((ext_vector_type(16 / sizeof(T))));
	*reinterpret_cast<NV*>(dst) = __builtin_nontemporal_load(reinterpret_cast<const NV*>(base) + i);
}
template <typename T>
__device__ inline void st16_nt(T* base, int64_t i, const T* src)
{
	typedef T NV __attribute__((ext_vector_type(16 / sizeof(T))));
	__builtin_nontemporal_store(*reinterpret_cast<const NV*>(src), reinterpret_cast<NV*>(base) + i);
}

// CG step, first half: r -= alpha q; partials of r.(Dinv r) and r.r          (reads r, q, Dinv; writes r)
// VEC: pointers 16-byte aligned and n a multiple of the vector width -> one 16-byte access per array.
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

// ---- single-rank CG with the reductions folded into the consumers -------------------------------------------
// The separate one-block reduce launches (4.8 us each plus two kernel boundaries per iteration) disappear:
// every block of the consumer kernel sums the producer's partials itself, in the same fixed order, so all
// blocks hold bit-identical scalars.  Block 0 publishes them into the OTHER scalar slot (no block reads the slot
// its kernel writes): k_cg_resid_f reads slot 0 and writes slot 1, k_cg_xp_f reads slot 1 and writes slot 0.
__device__ inline double block_sum_all(double v)  // the sum, in every thread
{
	__shared__ double s[kThreads / 64];
	const double w = wave_sum(v);
	if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6] = w; }
	__syncthreads();
	double r = 0;
	for (int k = 0; k < kThreads / 64; ++k) { r += s[k]; }
	__syncthreads();
	return r;
}

__device__ inline double sum_partials(const double* __restrict__ partial, int count)
{
	double acc = 0;
	for (int i = threadIdx.x; i < count; i += kThreads) { acc += partial[i]; }
	return block_sum_all(acc);
}

// first half: alpha from the p.q partials of the apply, r -= alpha q, partials of r.(Dinv r) and r.r
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

// ---- scalar kernels: one block --------------------------------------------------------------------
// Sums `nvec` partial arrays (each `stride` apart, `count` entries) in a fixed order into sc->sums[].
__global__ __launch_bounds__(kThreads) void k_reduce(CgScalars* sc, const double* __restrict__ partial, int nvec,
                                                      int stride, int count, int respect_done)
{
	if (respect_done && sc->done) { return; }
	for (int v = 0; v < nvec; ++v) {
		double acc[1] = {0};
		for (int i = threadIdx.x; i < count; i += kThreads) { acc[0] += partial[v * stride + i]; }
		double out[1];
		block_sum<1>(acc, out);
		if (threadIdx.x == 0) { sc->sums[v] = out[0]; }
	}
}

enum Phase { kPhaseInit = 0, kPhaseSpmv = 1, kPhaseUpdate = 2, kPhaseRestart = 3 };

__device__ inline void cg_logic(CgScalars* sc, int phase)
{
	if (phase == kPhaseInit) {
		sc->rz = sc->sums[0];
		sc->rr = sc->sums[1];
		sc->bb = sc->sums[2];
		sc->tol2 *= sc->bb;  // tol^2 * ||Atb||^2
		sc->iter = 0;
		sc->done = 0;
		if (sc->bb == 0.0) {
			sc->done = 4;  // rhs == 0: Eigen returns x = 0
		} else if (!(sc->rr > sc->tol2)) {
			sc->done = 1;
		} else if (sc->max_iter <= 0) {
			sc->done = 3;
		}
		return;
	}
	if (sc->done && phase != kPhaseRestart) { return; }
	if (phase == kPhaseSpmv) {
		sc->pq    = sc->sums[0];
		sc->alpha = sc->rz / sc->pq;
		if (!(sc->pq > 0.0) || !isfinite(sc->pq)) { sc->done = 2; }  // breakdown
		return;
	}
	if (phase == kPhaseRestart) {
		// r has been replaced by the true residual b - A x, p by Dinv r: accept if it meets the tolerance,
		// otherwise CG restarts from here (bb, tol2, iter and max_iter stay)
		sc->rz = sc->sums[0];
		sc->rr = sc->sums[1];
		sc->restarts += 1;
		sc->true_rr = sc->rr;
		sc->done = 0;
		if (!isfinite(sc->rr)) {
			sc->done = 2;
		} else if (!(sc->rr > sc->tol2)) {
			sc->done = 5;  // converged, and verified against b - A x
		} else if (sc->iter >= sc->max_iter) {
			sc->done = 3;
		}
		return;
	}
	// after the residual update
	sc->rz_new = sc->sums[0];
	sc->rr     = sc->sums[1];
	sc->beta   = sc->rz_new / sc->rz;
	sc->rz     = sc->rz_new;
	sc->iter += 1;
	if (!isfinite(sc->rr)) {
		sc->done = 2;
	} else if (!(sc->rr > sc->tol2)) {
		sc->done = 1;
	} else if (sc->iter >= sc->max_iter) {
		sc->done = 3;
	}
}

__global__ void k_set_done(CgScalars* sc, int value)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) { sc->done = value; }
}

__global__ void k_set_sum2(CgScalars* sc)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) { sc->sums[2] = sc->sums[0]; }
}
__global__ void k_bump_restarts(CgScalars* sc)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) { sc->restarts += 1; sc->done = 0; }
}

__global__ void k_cg_logic(CgScalars* sc, int phase)
{
	if (threadIdx.x == 0 && blockIdx.x == 0) { cg_logic(sc, phase); }
}

// single-rank form: fixed-order sum of the partials and the scalar recurrences in one launch
__global__ __launch_bounds__(kThreads) void k_reduce_logic(CgScalars* sc, const double* __restrict__ partial, int nvec,
                                                            int stride, int count, int phase)
{
	if (phase != kPhaseInit && phase != kPhaseRestart && sc->done) { return; }
	for (int v = 0; v < nvec; ++v) {
		double acc[1] = {0};
		for (int i = threadIdx.x; i < count; i += kThreads) { acc[0] += partial[v * stride + i]; }
		double out[1];
		block_sum<1>(acc, out);
		if (threadIdx.x == 0) { sc->sums[v] = out[0]; }
	}
	if (threadIdx.x == 0) { cg_logic(sc, phase); }
}

// ---- layout conversion between caller fp32 buffers (owned unknowns) and solver vectors ------------
template <typename T>
__global__ __launch_bounds__(kThreads) void k_from_float(int64_t n, const float* __restrict__ src, T* __restrict__ dst)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (i < n) { dst[i] = static_cast<T>(src[i]); }
}
template <typename T>
__global__ __launch_bounds__(kThreads) void k_to_float(int64_t n, const T* __restrict__ src, float* __restrict__ dst)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (i < n) { dst[i] = static_cast<float>(src[i]); }
}
template <typename T, typename U>
__global__ __launch_bounds__(kThreads) void k_convert(int64_t n, const T* __restrict__ src, U* __restrict__ dst)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (i < n) { dst[i] = static_cast<U>(src[i]); }
}

// upscale_field (field_interpolation.cpp:431-485): one thread per point of the large lattice.
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

// Coarse -> fine interpolation between two levels of the multilevel hierarchy, per axis (fi_ctx::cc):
//   vertex-centred (odd fine extent): fine point 2i coincides with coarse point i, odd fine points take the mean of their
//     two coarse neighbours (an even extent halved this way: the last fine point copies its only neighbour);
//   cell-centred (even fine extent): coarse point j sits between fine 2j and 2j+1; fine 2j takes 3/4 of coarse j and
//     1/4 of j-1, fine 2j+1 takes 3/4 of j and 1/4 of j+1; the first and the last fine point extrapolate (5/4, -1/4).
// One thread per fine point; mode 0: fine = P coarse, mode 1: fine += P coarse.
LevelPair level_pair(const fi_ctx* fine, const fi_ctx* coarse)
{
	LevelPair L{};
	L.ndim = fine->g.ndim;
	for (int d = 0; d < 3; ++d) {
		L.nf[d] = fine->g.gn[d];
		L.nc[d] = coarse->g.gn[d];
		L.cc[d] = coarse->cc[d];
	}
	const int a = L.ndim - 1;
	L.f_z0     = fine->g.off[a] + fine->g.own_lo[a];
	L.f_planes = fine->g.own_hi[a] - fine->g.own_lo[a];
	L.f_base   = fine->g.off[a];
	L.c_z0     = coarse->g.off[a] + coarse->g.own_lo[a];
	L.c_planes = coarse->g.own_hi[a] - coarse->g.own_lo[a];
	L.c_base   = coarse->g.off[a];
	if (coarse->replicated && !fine->replicated && fine->nranks > 1) {
		// a slab level above the replicated tail: the coarse lattice is whole on every rank; a rank RESTRICTS into the
		// coarse planes whose fine plane 2k it owns (the slab rule of build_levels; the parts are summed over the ranks) and
		// INTERPOLATES from any plane it needs
		const int lo = L.f_z0, hi = L.f_z0 + L.f_planes;
		L.c_z0     = (lo + 1) / 2;
		L.c_planes = (hi + 1) / 2 - L.c_z0;
		if (L.c_z0 + L.c_planes > L.nc[a]) { L.c_planes = L.nc[a] - L.c_z0; }
		if (L.c_planes < 0) { L.c_planes = 0; }
	}
	return L;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_prolong(LevelPair L, const T* __restrict__ coarse, T* __restrict__ fine,
                                                       int mode)
{
	// grid: (x blocks, y, z) over the OWNED fine points; `fine` / `coarse` are the local arrays (ghosts included)
	const int a = L.ndim - 1;
	int f[3] = {static_cast<int>(blockIdx.x * kThreads + threadIdx.x), static_cast<int>(blockIdx.y),
	            static_cast<int>(blockIdx.z)};
	if (f[0] >= (a == 0 ? L.f_planes : L.nf[0])) { return; }
	f[a] += L.f_z0;  // global coordinate along the decomposed axis
	int c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
	T   w0[3] = {T(1), T(1), T(1)}, w1[3] = {T(0), T(0), T(0)};
	for (int d = 0; d < L.ndim; ++d) { prolong_taps<T>(f[d], L.nc[d], L.cc[d], &c0[d], &c1[d], &w0[d], &w1[d]); }
	c0[a] -= L.c_base;  // local plane indices of the coarse slab (ghost planes hold the neighbours' values)
	c1[a] -= L.c_base;
	f[a] -= L.f_base;
	const int64_t csy = L.nc[0], csz = (L.ndim > 2 ? static_cast<int64_t>(L.nc[0]) * L.nc[1] : 0);
	T acc = T(0);
	for (int q = 0; q < (1 << L.ndim); ++q) {
		const int ux = q & 1, uy = (q >> 1) & 1, uz = (q >> 2) & 1;
		T w = ux ? w1[0] : w0[0];
		int64_t idx = ux ? c1[0] : c0[0];
		if (L.ndim > 1) { w *= uy ? w1[1] : w0[1]; idx += csy * (uy ? c1[1] : c0[1]); }
		if (L.ndim > 2) { w *= uz ? w1[2] : w0[2]; idx += csz * (uz ? c1[2] : c0[2]); }
		if (w != T(0)) { acc += w * coarse[idx]; }
	}
	const int64_t i = (L.ndim > 2 ? static_cast<int64_t>(f[2]) * L.nf[1] * L.nf[0] : 0) +
	                  (L.ndim > 1 ? static_cast<int64_t>(f[1]) * L.nf[0] : 0) + f[0];
	fine[i] = mode ? fine[i] + acc : acc;
}

// The same for 3-D lattices with everything resolved at compile time (the generic kernel indexes its coordinate
// arrays by the runtime axis: they live in scratch memory -- 1.15 ms per call at 512^3 against 0.3 ms here).  A thread
// owns a 2 x 2 x 2 block of fine points (2j, 2j+1 along every axis): between them they draw on the coarse points
// j-1 .. j+1, a window of 3 x 3 rows that is interpolated along x once (three loads, both x parities) and then spread over
// the four (y, z) parities -- and every thread of a workgroup has work (one thread per PAIR of points and a row per
// workgroup left half of the threads idle at 256^3: 78 us for 142 MB).
template <typename T>
__device__ inline void linear_window(int f, int nc, int cc, int u0, bool live, T* W)
{
	int i0, i1;
	T   w0, w1;
	prolong_taps<T>(f, nc, cc, &i0, &i1, &w0, &w1);
#pragma unroll
	for (int s = 0; s < 3; ++s) { W[s] = live ? ((i0 - u0 == s ? w0 : T(0)) + (i1 - u0 == s ? w1 : T(0))) : T(0); }
}
template <typename T>
__global__ __launch_bounds__(kThreads) void k_prolong3(LevelPair L, const T* __restrict__ coarse, T* __restrict__ fine,
                                                        int mode)
{
	const int px = (L.nf[0] + 1) / 2, py = (L.nf[1] + 1) / 2;
	const int jz0 = L.f_z0 >> 1, jz1 = (L.f_z0 + L.f_planes - 1) >> 1;
	int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (t >= static_cast<int64_t>(px) * py * (jz1 - jz0 + 1)) { return; }
	const int jx = static_cast<int>(t % px);
	t /= px;
	const int jy = static_cast<int>(t % py), jz = jz0 + static_cast<int>(t / py);
	const int ux = jx - 1, uy = jy - 1, uz = jz - 1;
	T Wx[2][3], Wy[2][3], Wz[2][3];
	bool live[3][2];
#pragma unroll
	for (int p = 0; p < 2; ++p) {
		live[0][p] = 2 * jx + p < L.nf[0];
		live[1][p] = 2 * jy + p < L.nf[1];
		live[2][p] = 2 * jz + p >= L.f_z0 && 2 * jz + p < L.f_z0 + L.f_planes;  // (slabs: the owned planes only)
		linear_window<T>(live[0][p] ? 2 * jx + p : 2 * jx, L.nc[0], L.cc[0], ux, live[0][p], Wx[p]);
		linear_window<T>(live[1][p] ? 2 * jy + p : 2 * jy, L.nc[1], L.cc[1], uy, live[1][p], Wy[p]);
		linear_window<T>(live[2][p] ? 2 * jz + p : 2 * jz + 1 - p, L.nc[2], L.cc[2], uz, live[2][p], Wz[p]);
	}
	int xi[3];
#pragma unroll
	for (int s = 0; s < 3; ++s) {
		const int v = ux + s;
		xi[s] = v < 0 ? 0 : (v > L.nc[0] - 1 ? L.nc[0] - 1 : v);
	}
	const int64_t csy = L.nc[0], csz = static_cast<int64_t>(L.nc[0]) * L.nc[1];
	T acc[2][2][2];
#pragma unroll
	for (int q = 0; q < 8; ++q) { acc[q >> 2][(q >> 1) & 1][q & 1] = T(0); }
	// every load unconditional; a plane without weight (it may lie beyond the slab's ghost plane) is replaced by coarse
	// plane jz, which every live parity draws on
	const int safe_z = (jz > L.nc[2] - 1 ? L.nc[2] - 1 : jz) - L.c_base;
#pragma unroll
	for (int sz = 0; sz < 3; ++sz) {
		const int vz = uz + sz;
		const int cz = (Wz[0][sz] == T(0) && Wz[1][sz] == T(0)) ? safe_z
		                                                        : (vz < 0 ? 0 : (vz > L.nc[2] - 1 ? L.nc[2] - 1 : vz)) - L.c_base;
#pragma unroll
		for (int sy = 0; sy < 3; ++sy) {
			const int vy = uy + sy;
			const T* row = coarse + csz * cz + csy * (vy < 0 ? 0 : (vy > L.nc[1] - 1 ? L.nc[1] - 1 : vy));
			const T v0 = row[xi[0]], v1 = row[xi[1]], v2 = row[xi[2]];
			const T e = Wx[0][0] * v0 + Wx[0][1] * v1 + Wx[0][2] * v2;
			const T o = Wx[1][0] * v0 + Wx[1][1] * v1 + Wx[1][2] * v2;
#pragma unroll
			for (int pz = 0; pz < 2; ++pz) {
#pragma unroll
				for (int pyb = 0; pyb < 2; ++pyb) {
					const T w = Wy[pyb][sy] * Wz[pz][sz];
					acc[pz][pyb][0] += w * e;
					acc[pz][pyb][1] += w * o;
				}
			}
		}
	}
#pragma unroll
	for (int pz = 0; pz < 2; ++pz) {
#pragma unroll
		for (int pyb = 0; pyb < 2; ++pyb) {
			if (!live[2][pz] || !live[1][pyb]) { continue; }
			const int64_t i = (static_cast<int64_t>(2 * jz + pz - L.f_base) * L.nf[1] + (2 * jy + pyb)) * L.nf[0] + 2 * jx;
			fine[i] = mode ? fine[i] + acc[pz][pyb][0] : acc[pz][pyb][0];
			if (live[0][1]) { fine[i + 1] = mode ? fine[i + 1] + acc[pz][pyb][1] : acc[pz][pyb][1]; }
		}
	}
}
// 2-D form (the generic kernel indexes its coordinate arrays by the runtime axis -- scratch memory: 127 us per call at
// 4096^2 against the 25 us two lattice passes take).  A thread owns the fine points 2t and 2t+1 of a row; y is the
// decomposed axis.
template <typename T>
__global__ __launch_bounds__(kThreads) void k_prolong2(LevelPair L, const T* __restrict__ coarse, T* __restrict__ fine, int mode)
{
	const int t  = static_cast<int>(blockIdx.x * kThreads + threadIdx.x);
	const int fx = 2 * t;
	if (fx >= L.nf[0]) { return; }
	const int fy = static_cast<int>(blockIdx.y) + L.f_z0;  // global row
	int xe0, xe1, xo0, xo1, cy[2];
	T   we0, we1, wo0, wo1, wy[2];
	prolong_taps<T>(fx, L.nc[0], L.cc[0], &xe0, &xe1, &we0, &we1);
	prolong_taps<T>(fx + 1 < L.nf[0] ? fx + 1 : fx, L.nc[0], L.cc[0], &xo0, &xo1, &wo0, &wo1);
	prolong_taps<T>(fy, L.nc[1], L.cc[1], &cy[0], &cy[1], &wy[0], &wy[1]);
	T even = T(0), odd = T(0);
#pragma unroll
	for (int uy = 0; uy < 2; ++uy) {
		if (wy[uy] == T(0)) { continue; }
		const T* row = coarse + static_cast<int64_t>(cy[uy] - L.c_base) * L.nc[0];
		even += wy[uy] * (we0 * row[xe0] + we1 * row[xe1]);
		odd += wy[uy] * (wo0 * row[xo0] + wo1 * row[xo1]);
	}
	const int64_t i = static_cast<int64_t>(fy - L.f_base) * L.nf[0] + fx;
	fine[i] = mode ? fine[i] + even : even;
	if (fx + 1 < L.nf[0]) { fine[i + 1] = mode ? fine[i + 1] + odd : odd; }
}

// Cubic interpolation for the coarse-to-fine START (not the V-cycle: its P must stay the transpose of R).  Vertex-centred
// axis: a fine point between two coarse points takes (-1, 9, 9, -1) / 16 of the four nearest (next to the lattice's ends,
// where they do not fit, the mean of its two neighbours), a coincident one the coarse value.  Cell-centred axis: a fine point sits a quarter of a coarse cell
// from its coarse point j -- the cubic through j-1 .. j+2 at +1/4 (mirrored at -1/4); where the four taps do not fit
// (first and last two fine points) the linear taps of k_prolong.  3-D lattices; slabs need two ghost planes of the
// coarse solution.
template <typename T>
__device__ inline void cubic_taps(int f, int nc, int cc, int* idx, T* w)
{
	const int j = f >> 1;
	if (!cc) {
		if ((f & 1) && (j < 1 || j + 2 > nc - 1)) {
			// next to an end the four taps do not fit: the mean of the two neighbours, like k_prolong (with clamped indices the
			// weights (-1, 9, 9, -1) / 16 put 7/16 where a linear function needs 1/2 -- rounds 2 and early 3)
			prolong_taps<T>(f, nc, 0, &idx[0], &idx[1], &w[0], &w[1]);
			idx[2] = idx[3] = idx[0];
			w[2] = w[3] = T(0);
			return;
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int v = j - 1 + k;
			idx[k] = v < 0 ? 0 : (v > nc - 1 ? nc - 1 : v);
		}
		if (f & 1) {
			w[0] = T(-1.0 / 16.0); w[1] = T(9.0 / 16.0); w[2] = T(9.0 / 16.0); w[3] = T(-1.0 / 16.0);
		} else {
			w[0] = T(0); w[1] = T(1); w[2] = T(0); w[3] = T(0);
		}
		return;
	}
	const int base = (f & 1) ? j - 1 : j - 2;
	if (base >= 0 && base + 3 < nc) {
#pragma unroll
		for (int k = 0; k < 4; ++k) { idx[k] = base + k; }
		// Lagrange weights of the nodes -1, 0, 1, 2 at t = 1/4
		const T a = T(-0.0546875), b = T(0.8203125), c = T(0.2734375), d = T(-0.0390625);
		if (f & 1) { w[0] = a; w[1] = b; w[2] = c; w[3] = d; } else { w[0] = d; w[1] = c; w[2] = b; w[3] = a; }
	} else {
		prolong_taps<T>(f, nc, 1, &idx[0], &idx[1], &w[0], &w[1]);
		idx[2] = idx[3] = idx[0];
		w[2] = w[3] = T(0);
	}
}
// The weights of fine point f on the WINDOW of five coarse points u0 .. u0+4 that the pair of fine points (2j, 2j+1)
// draws on between them (u0 = j-2 cell-centred, j-1 vertex-centred): the four taps of cubic_taps, dropped into their slots
// (a clamped index that occurs twice adds up).  Static indices only: everything stays in registers.
template <typename T>
__device__ inline void cubic_window(int f, int nc, int cc, int u0, bool live, T* W)
{
	int idx[4];
	T   w[4];
	cubic_taps<T>(f, nc, cc, idx, w);
#pragma unroll
	for (int s = 0; s < 5; ++s) {
		T acc = T(0);
#pragma unroll
		for (int k = 0; k < 4; ++k) { acc += (idx[k] - u0 == s) ? w[k] : T(0); }
		W[s] = live ? acc : T(0);
	}
}
// A thread owns a 2 x 2 x 2 block of fine points: the 5 x 5 coarse rows around it are interpolated along x ONCE each (five
// loads, both x parities) and then spread over the four (y, z) parities -- 125 loads for eight fine points instead of the
// 64 per point of one thread per pair of points (256^3 from 128^3: 177 -> ... us).
template <typename T>
__global__ __launch_bounds__(kThreads) void k_prolong3_cubic(LevelPair L, const T* __restrict__ coarse, T* __restrict__ fine)
{
	const int px = (L.nf[0] + 1) / 2, py = (L.nf[1] + 1) / 2;
	const int jz0 = L.f_z0 >> 1, jz1 = (L.f_z0 + L.f_planes - 1) >> 1;
	int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (t >= static_cast<int64_t>(px) * py * (jz1 - jz0 + 1)) { return; }
	const int jx = static_cast<int>(t % px);
	t /= px;
	const int jy = static_cast<int>(t % py), jz = jz0 + static_cast<int>(t / py);
	const int ux = jx - (L.cc[0] ? 2 : 1), uy = jy - (L.cc[1] ? 2 : 1), uz = jz - (L.cc[2] ? 2 : 1);
	T Wx[2][5], Wy[2][5], Wz[2][5];
	bool live[3][2];
#pragma unroll
	for (int p = 0; p < 2; ++p) {
		live[0][p] = 2 * jx + p < L.nf[0];
		live[1][p] = 2 * jy + p < L.nf[1];
		live[2][p] = 2 * jz + p >= L.f_z0 && 2 * jz + p < L.f_z0 + L.f_planes;  // (slabs: the owned planes only)
		cubic_window<T>(live[0][p] ? 2 * jx + p : 2 * jx, L.nc[0], L.cc[0], ux, live[0][p], Wx[p]);
		cubic_window<T>(live[1][p] ? 2 * jy + p : 2 * jy, L.nc[1], L.cc[1], uy, live[1][p], Wy[p]);
		cubic_window<T>(live[2][p] ? 2 * jz + p : 2 * jz + 1 - p, L.nc[2], L.cc[2], uz, live[2][p], Wz[p]);
	}
	int xi[5];
#pragma unroll
	for (int s = 0; s < 5; ++s) {
		const int v = ux + s;
		xi[s] = v < 0 ? 0 : (v > L.nc[0] - 1 ? L.nc[0] - 1 : v);
	}
	const int64_t csy = L.nc[0], csz = static_cast<int64_t>(L.nc[0]) * L.nc[1];
	T acc[2][2][2];
#pragma unroll
	for (int q = 0; q < 8; ++q) { acc[q >> 2][(q >> 1) & 1][q & 1] = T(0); }
	// Every load is unconditional (25 dependent round trips otherwise: 103 us at 256^3): a plane without weight -- it may
	// lie beyond the slab's ghost planes -- is replaced by coarse plane jz, which every live parity draws on.
	const int safe_z = (jz > L.nc[2] - 1 ? L.nc[2] - 1 : jz) - L.c_base;
#pragma unroll
	for (int sz = 0; sz < 5; ++sz) {
		const int vz = uz + sz;
		const int cz = (Wz[0][sz] == T(0) && Wz[1][sz] == T(0)) ? safe_z
		                                                        : (vz < 0 ? 0 : (vz > L.nc[2] - 1 ? L.nc[2] - 1 : vz)) - L.c_base;
#pragma unroll
		for (int sy = 0; sy < 5; ++sy) {
			const int vy = uy + sy;
			const T* row = coarse + csz * cz + csy * (vy < 0 ? 0 : (vy > L.nc[1] - 1 ? L.nc[1] - 1 : vy));
			const T v0 = row[xi[0]], v1 = row[xi[1]], v2 = row[xi[2]], v3 = row[xi[3]], v4 = row[xi[4]];
			const T e = Wx[0][0] * v0 + Wx[0][1] * v1 + Wx[0][2] * v2 + Wx[0][3] * v3 + Wx[0][4] * v4;
			const T o = Wx[1][0] * v0 + Wx[1][1] * v1 + Wx[1][2] * v2 + Wx[1][3] * v3 + Wx[1][4] * v4;
#pragma unroll
			for (int pz = 0; pz < 2; ++pz) {
#pragma unroll
				for (int pyb = 0; pyb < 2; ++pyb) {
					const T w = Wy[pyb][sy] * Wz[pz][sz];
					acc[pz][pyb][0] += w * e;
					acc[pz][pyb][1] += w * o;
				}
			}
		}
	}
#pragma unroll
	for (int pz = 0; pz < 2; ++pz) {
#pragma unroll
		for (int pyb = 0; pyb < 2; ++pyb) {
			if (!live[2][pz] || !live[1][pyb]) { continue; }
			const int64_t i = (static_cast<int64_t>(2 * jz + pz - L.f_base) * L.nf[1] + (2 * jy + pyb)) * L.nf[0] + 2 * jx;
			fine[i] = acc[pz][pyb][0];
			if (live[0][1]) { fine[i + 1] = acc[pz][pyb][1]; }
		}
	}
}

template <typename T>
void launch_prolong(const LevelPair& L, const T* coarse, T* fine, int mode, hipStream_t st);

// grid over the owned points of a level: x in blocks of 256, then y, z (the decomposed axis counts planes)
inline dim3 owned_grid(const int* n, int ndim, int planes)
{
	int e[3] = {n[0], n[1], n[2]};
	e[ndim - 1] = planes;
	return dim3((e[0] + kThreads - 1) / kThreads, e[1], e[2]);
}

template <typename T>
void launch_prolong(const LevelPair& L, const T* coarse, T* fine, int mode, hipStream_t st)
{
	if (L.ndim == 3) {
		const int64_t blocks8 = static_cast<int64_t>((L.nf[0] + 1) / 2) * ((L.nf[1] + 1) / 2) *
		                        (((L.f_z0 + L.f_planes - 1) >> 1) - (L.f_z0 >> 1) + 1);
		if (L.f_planes > 0) {
			hipLaunchKernelGGL((k_prolong3<T>), dim3(static_cast<unsigned>((blocks8 + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, L,
			                   coarse, fine, mode);
		}
	} else if (L.ndim == 2) {
		const int pairs = (L.nf[0] + 1) / 2;
		hipLaunchKernelGGL((k_prolong2<T>), dim3((pairs + kThreads - 1) / kThreads, L.f_planes), dim3(kThreads), 0, st, L, coarse, fine,
		                   mode);
	} else {
		hipLaunchKernelGGL((k_prolong<T>), owned_grid(L.nf, L.ndim, L.f_planes), dim3(kThreads), 0, st, L, coarse, fine, mode);
	}
}

// ---- host side -------------------------------------------------------------------------------------

void compute_geom(fi_ctx* c, int ndim, const int* sizes)
{
	Geom& g = c->g;
	g = Geom{};
	g.ndim = ndim;
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

template <typename T>
T* owned(const fi_ctx* c, const DevBuf& b)
{
	return b.as<T>() + c->g.own_first;
}

void ensure_vectors(fi_ctx* c)
{
	if (c->vectors_ready) { return; }
	const size_t es = elem_size(c);
	const Geom&  g  = c->g;
	c->x.alloc(es * g.nloc);
	c->r.alloc(es * g.nloc);
	c->p.alloc(es * g.nloc);
	c->q.alloc(es * g.nloc);
	FI_HIP_TRY(hipMemsetAsync(c->x.p, 0, es * g.nloc, c->stream));
	FI_HIP_TRY(hipMemsetAsync(c->r.p, 0, es * g.nloc, c->stream));
	FI_HIP_TRY(hipMemsetAsync(c->p.p, 0, es * g.nloc, c->stream));
	FI_HIP_TRY(hipMemsetAsync(c->q.p, 0, es * g.nloc, c->stream));
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

// ---- rank sets ---------------------------------------------------------------------------------------
// The solver drivers run over a set of slab contexts in lockstep.  In production the set has ONE member
// (this process's slab; neighbours are reached through RCCL, fi_comm.hip).  A loop-back group
// (fi_group_create) puts ALL slabs of a decomposition into one process on one device and one stream: halo
// planes move with device-to-device copies and the dot products are summed by a tiny kernel.  It exists so
// that the slab geometry, halo widths, global-coordinate boundary masks and cell ownership rules can be
// tested on a single GPU against the undivided solve; it runs the same kernels as the RCCL path.
using RankSet = std::vector<fi_ctx*>;

__global__ void k_group_sum(CgScalars* const* sc, int nranks, int nvec, int slot)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) { return; }
	for (int v = 0; v < nvec; ++v) {
		double s = 0;
		for (int r = 0; r < nranks; ++r) { s += sc[r][slot].sums[v]; }  // fixed order
		for (int r = 0; r < nranks; ++r) { sc[r][slot].sums[v] = s; }
	}
}

void halo_exchange(RankSet& R, DevBuf fi_ctx::*vec, int width = 0)  // width planes next to the slabs (0: the stencil's reach)
{
	if (R[0]->nranks == 1) { return; }  // whole lattices (a loop-back group's copies of the replicated tail included)
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
	return c->nranks > 1 && c->march.valid && c->march.n_inner > 0 && !c->any_trip && c->generic.ntrip == 0 && c->tile_ts == 0 &&
	       (c->cells.ncell == 0 || cells_fused(c)) && !test_switch("FI_NO_OVERLAP");
}
void exchange_begin(fi_ctx* c, void* v)
{
	if (!c->comm_stream) {
		FI_HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
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
template <typename CountFn, typename StrideFn>
void reduce_phase(RankSet& R, int nvec, CountFn count_of, StrideFn stride_of, int phase)
{
	if (R.size() == 1 && R[0]->nranks == 1 && phase >= 0) {
		fi_ctx* c = R[0];
		hipLaunchKernelGGL(k_reduce_logic, dim3(1), dim3(kThreads), 0, c->stream, c->scal.as<CgScalars>(),
		                   c->partial.as<double>(), nvec, stride_of(c), count_of(c), phase);
		return;
	}
	for (fi_ctx* c : R) {
		hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kThreads), 0, c->stream, c->scal.as<CgScalars>(),
		                   c->partial.as<double>(), nvec, stride_of(c), count_of(c),
		                   (phase == kPhaseInit || phase == kPhaseRestart || phase < 0) ? 0 : 1);
	}
	if (R.size() > 1) {
		fi_ctx* c0 = R[0];
		hipLaunchKernelGGL(k_group_sum, dim3(1), dim3(1), 0, c0->stream, c0->group_scal.as<CgScalars*>(),
		                   static_cast<int>(R.size()), nvec, 0);
	} else if (R[0]->nranks > 1) {
		allreduce_sum(R[0], R[0]->scal.as<CgScalars>()->sums, nvec);
	}
	if (phase >= 0) {
		for (fi_ctx* c : R) { hipLaunchKernelGGL(k_cg_logic, dim3(1), dim3(1), 0, c->stream, c->scal.as<CgScalars>(), phase); }
	}
}

// The same sums for the folded CG kernels of a rank set: partials -> one value per vector in scalar slot 2 of every
// member, summed over slabs (loop-back group: a summing kernel; one slab per process: RCCL all-reduce in place).  The
// folded kernels then take slot 2's sums as a partial list of length one and do the scalar recurrences themselves --
// no k_cg_logic launch.  Slot 2 is written only here, between the kernels that read it.
template <typename CountFn, typename StrideFn>
void reduce_to_slot2(RankSet& R, int nvec, CountFn count_of, StrideFn stride_of, const double* (*partials_of)(fi_ctx*))
{
	for (fi_ctx* c : R) {
		hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kThreads), 0, c->stream, c->scal.as<CgScalars>() + 2, partials_of(c), nvec,
		                   stride_of(c), count_of(c), 0);
	}
	if (R.size() > 1) {
		fi_ctx* c0 = R[0];
		hipLaunchKernelGGL(k_group_sum, dim3(1), dim3(1), 0, c0->stream, c0->group_scal.as<CgScalars*>(),
		                   static_cast<int>(R.size()), nvec, 2);
	} else if (R[0]->nranks > 1) {
		allreduce_sum(R[0], (R[0]->scal.as<CgScalars>() + 2)->sums, nvec);
	}
}

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

	EventPair timer;  // (destroyed on every way out: a coarse level's breakdown, a timeout)
	const hipEvent_t e0 = timer.e0, e1 = timer.e1;
	FI_HIP_TRY(hipEventRecord(e0, st));

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
	for (;;) {
		FI_HIP_TRY(hipMemcpyAsync(c0->scal_host, sc0, sizeof(CgScalars), hipMemcpyDeviceToHost, st));
		FI_HIP_TRY(hipStreamSynchronize(st));
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
		for (int k = 0; k < kCheckEvery; ++k) {
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

bool poly_ok(const fi_ctx* c);
void remember_lambda(const fi_ctx* c);
template <typename T>
void cg_run_poly(RankSet& R, int max_iterations, float tol);
template <typename T>
void estimate_poly_lambda(RankSet& R);
template <typename T>
void cg_run_mg(RankSet& R, int max_iterations, float tol);

// Coarse-to-fine start (the reference's own remedy for large lattices: solve a coarser lattice, upscale, use
// as the guess -- src/sdf_field.cpp:272-288, README.md "My resolution is huge"): every coarser level is
// solved from the interpolated solution of the level below it, to a loose tolerance; x of `c` receives the
// interpolated guess.  All on the device.
template <typename T>
void cascade_guess(RankSet& R)
{
	// chains[l] = the level-l contexts of all members
	std::vector<RankSet> chains;
	{
		RankSet cur = R;
		for (;;) {
			chains.push_back(cur);
			RankSet next;
			for (fi_ctx* c : cur) {
				if (c->coarse) { next.push_back(c->coarse); }
			}
			if (next.size() != cur.size()) { break; }
			cur = next;
		}
	}
	for (auto& lev : chains) {
		for (fi_ctx* c : lev) { ensure_vectors(c); }
	}
	for (fi_ctx* c : chains.back()) { FI_HIP_TRY(hipMemsetAsync(c->x.p, 0, sizeof(T) * c->g.nloc, c->stream)); }
	fi_ctx* root = R[0];
	for (size_t k = chains.size(); k-- > 1;) {
		RankSet& lc = chains[k];
		RankSet& lf = chains[k - 1];
		try {
			for (fi_ctx* l : lc) {  // the levels solve the way the finest level does
				l->poly_terms = root->poly_terms;
				l->poly_ratio = root->poly_ratio;
				if (const char* e = tuning_switch("FI_COARSE_TERMS")) { l->poly_terms = atoi(e); }
				if (const char* e = tuning_switch("FI_COARSE_RATIO")) { l->poly_ratio = atof(e); }
			}
			// with the V-cycle preconditioner on, a level that has coarser levels below it is solved with it too (a full
			// multigrid start): Jacobi-PCG needs thousands of iterations on the coarse levels of an SDF (config 3: 4 338,
			// most of the solve's wall time)
			const bool mg = root->mg_mode == 1 && lc[0]->coarse && !test_switch("FI_CASCADE_NO_MG");
			for (fi_ctx* l : lc) { l->mg_mode = root->mg_mode; }
			for_each_copy(lc, [&](RankSet& lc) {  // (the copies of a replicated level: each on its own)
			if (poly_ok(lc[0])) {
				cg_run_poly<T>(lc, 0, static_cast<float>(root->coarse_tol));
			} else if (mg) {
				// Data-rich levels (config 4) are done after a few dozen cheap Jacobi-PCG steps; where that is not enough the
				// V-cycle takes over from the iterate.  A start guess is worth a bounded effort, and fp32 levels cannot go
				// below ~1e-5 anyway (the recurrence stalls there: 29 000 iterations without reaching 1e-6 at 48^3).
				const double floor_tol = sizeof(T) == 4 ? 1e-5 : 0.0;
				const float  ltol = static_cast<float>(root->coarse_tol > floor_tol ? root->coarse_tol : floor_tol);
				// (measured, tools/r3_sweep_cascade.sh: config 3 wants its levels converged -- 13 fine iterations at 4096^2
				// instead of 19 / 26 with 8 / 4 cycles per level --, config 5's 256^3 level is not worth more than 8 cycles:
				// 439 -> 396 ms per step)
				int64_t n_level = 1;
				for (int d = 0; d < 3; ++d) { n_level *= lc[0]->g.gn[d]; }
				const int cap_mg = n_level <= (1LL << 22) ? 40 : 8;
				// (oriented points: the 48 cheap steps never finish a level -- 144 launches per level for nothing, a tenth of
				// config 3's step -- so those levels go straight to the V-cycle)
				int coarse_it = 0;
				bool finished = false;
				if (lc[0]->value_rows_only) {
					cg_run<T>(lc, 48, ltol);
					coarse_it = lc[0]->stats.iterations;
					finished  = lc[0]->stats.converged != 0;
				}
				if (!finished) {
					cg_run_mg<T>(lc, cap_mg, ltol);
					coarse_it += lc[0]->stats.iterations;
				}
				for (fi_ctx* l : lc) { l->stats.iterations = coarse_it; }
			} else {
				cg_run<T>(lc, 0, static_cast<float>(root->coarse_tol));
			}
			});
		} catch (const Fail& f) {
			if (f.code != FI_ERR_BREAKDOWN) { throw; }  // a coarse level without data: keep what it has
		}
		root->stats.coarse_iterations += lc[0]->stats.iterations;
		halo_exchange(lc, &fi_ctx::x);  // interpolation reads one coarse plane beyond the slab
		for (size_t i = 0; i < lf.size(); ++i) {
			const LevelPair L = level_pair(lf[i], lc[i]);
			// cubic where the coarse vector stays in cache (64 taps per pair of fine points): 256^3 from 128^3 20 -> 19
			// outer iterations; at 512^3 the kernel would cost more than the start it improves
			const bool cubic = L.ndim == 3 && (lf[i]->nranks == 1 || lc[i]->reach >= 2) && !test_switch("FI_LINEAR_START") &&
			                   sizeof(T) * static_cast<size_t>(lc[i]->g.nloc) <= (32u << 20);
			if (cubic) {
				const int64_t blocks8 = static_cast<int64_t>((L.nf[0] + 1) / 2) * ((L.nf[1] + 1) / 2) *
				                        (((L.f_z0 + L.f_planes - 1) >> 1) - (L.f_z0 >> 1) + 1);
				if (L.f_planes > 0) {
					hipLaunchKernelGGL((k_prolong3_cubic<T>), dim3(static_cast<unsigned>((blocks8 + kThreads - 1) / kThreads)), dim3(kThreads),
					                   0, lf[i]->stream, L, lc[i]->x.as<T>(), lf[i]->x.as<T>());
				}
			} else {
				launch_prolong<T>(L, lc[i]->x.as<T>(), lf[i]->x.as<T>(), 0, lf[i]->stream);
			}
		}
		FI_HIP_TRY(hipGetLastError());
	}
}

// ---- multigrid V-cycle preconditioner ----------------------------------------------------------------
// With FI_OPT_MULTIGRID the coarser replicas (build_levels) precondition CG on the finest level:
//   z = V(r):  pre-smooth from zero, restrict the residual (R = P^T), recurse, interpolate and add, post-smooth.
// Smoother: a degree-k Chebyshev polynomial in Dinv*AtA on [lambda_max/ratio, 1.1 lambda_max] -- only operator
// applies and axpys, no dot products (nothing to all-reduce), and the same polynomial before and after the
// coarse correction makes V symmetric positive definite, as CG needs.  lambda_max comes from 10 steps of the
// power method per level at assemble time (one host read per level).

// restriction = transpose of k_prolong.  Vertex-centred axis: coarse point c gathers fine 2c (weight 1) and 2c-1, 2c+1
// (weight 1/2; the last coarse point also takes the full weight of a fine point beyond it).  Cell-centred axis: fine
// 2c-1 .. 2c+2 with (1/4, 3/4, 3/4, 1/4); the end points' extrapolation puts 5/4 of fine 0 on coarse 0 and -1/4 of it on
// coarse 1 (mirrored at the other end): five taps.  Indices are relative to `base` and clamped where the weight is 0.
template <typename T>
__global__ __launch_bounds__(kThreads) void k_restrict(LevelPair L, const T* __restrict__ fine, T* __restrict__ coarse)
{
	// grid: (x blocks, y, z) over the OWNED coarse points; local arrays, ghost planes of `fine` up to date
	const int a = L.ndim - 1;
	int c[3] = {static_cast<int>(blockIdx.x * kThreads + threadIdx.x), static_cast<int>(blockIdx.y),
	            static_cast<int>(blockIdx.z)};
	if (c[0] >= (a == 0 ? L.c_planes : L.nc[0])) { return; }
	c[a] += L.c_z0;
	int f[3][kRTaps];
	T   w[3][kRTaps];
	for (int d = 0; d < 3; ++d) {
		for (int k = 0; k < kRTaps; ++k) { f[d][k] = 0; w[d][k] = (k == 0) ? T(1) : T(0); }
	}
	for (int d = 0; d < L.ndim; ++d) { restrict_taps<T>(c[d], L.nf[d], L.nc[d], L.cc[d], d == a ? L.f_base : 0, f[d], w[d]); }
	c[a] -= L.c_base;
	const int64_t sy = L.nf[0];
	const int64_t sz = static_cast<int64_t>(L.nf[0]) * L.nf[1];
	T acc = T(0);
	const int n1 = L.ndim > 1 ? kRTaps : 1, n2 = L.ndim > 2 ? kRTaps : 1;
	for (int k2 = 0; k2 < n2; ++k2) {
		for (int k1 = 0; k1 < n1; ++k1) {
			const T w12 = w[1][k1] * w[2][k2];
			if (w12 == T(0)) { continue; }
			const int64_t base = (L.ndim > 1 ? sy * f[1][k1] : 0) + (L.ndim > 2 ? sz * f[2][k2] : 0);
			for (int k0 = 0; k0 < kRTaps; ++k0) {
				if (w[0][k0] != T(0)) { acc += w[0][k0] * w12 * fine[base + f[0][k0]]; }
			}
		}
	}
	const int64_t i = (L.ndim > 2 ? static_cast<int64_t>(c[2]) * L.nc[1] * L.nc[0] : 0) +
	                  (L.ndim > 1 ? static_cast<int64_t>(c[1]) * L.nc[0] : 0) + c[0];
	coarse[i] = acc;
}

// 3-D form of k_restrict with compile-time loops (same weights, same order of summation)
template <typename T>
__global__ __launch_bounds__(kThreads) void k_restrict3(LevelPair L, const T* __restrict__ fine, T* __restrict__ coarse)
{
	const int cx = static_cast<int>(blockIdx.x * kThreads + threadIdx.x);
	if (cx >= L.nc[0]) { return; }
	const int cy = static_cast<int>(blockIdx.y);
	const int cz = static_cast<int>(blockIdx.z) + L.c_z0;  // global plane
	int fx[kRTaps], fy[kRTaps], fz[kRTaps];
	T   wx[kRTaps], wy[kRTaps], wz[kRTaps];
	restrict_taps<T>(cx, L.nf[0], L.nc[0], L.cc[0], 0, fx, wx);
	restrict_taps<T>(cy, L.nf[1], L.nc[1], L.cc[1], 0, fy, wy);
	restrict_taps<T>(cz, L.nf[2], L.nc[2], L.cc[2], L.f_base, fz, wz);
	const int64_t sy = L.nf[0], sz = static_cast<int64_t>(L.nf[0]) * L.nf[1];
	T acc = T(0);
#pragma unroll
	for (int k2 = 0; k2 < kRTaps; ++k2) {
		if (wz[k2] == T(0)) { continue; }
#pragma unroll
		for (int k1 = 0; k1 < kRTaps; ++k1) {
			const T w12 = wy[k1] * wz[k2];
			if (w12 == T(0)) { continue; }
			const T* row = fine + sy * fy[k1] + sz * fz[k2];
#pragma unroll
			for (int k0 = 0; k0 < kRTaps; ++k0) {
				if (wx[k0] != T(0)) { acc += wx[k0] * w12 * row[fx[k0]]; }
			}
		}
	}
	coarse[(static_cast<int64_t>(cz - L.c_base) * L.nc[1] + cy) * L.nc[0] + cx] = acc;
}
// 2-D form of k_restrict with compile-time loops
template <typename T>
__global__ __launch_bounds__(kThreads) void k_restrict2(LevelPair L, const T* __restrict__ fine, T* __restrict__ coarse)
{
	const int cx = static_cast<int>(blockIdx.x * kThreads + threadIdx.x);
	if (cx >= L.nc[0]) { return; }
	const int cy = static_cast<int>(blockIdx.y) + L.c_z0;  // global row
	int fx[kRTaps], fy[kRTaps];
	T   wx[kRTaps], wy[kRTaps];
	restrict_taps<T>(cx, L.nf[0], L.nc[0], L.cc[0], 0, fx, wx);
	restrict_taps<T>(cy, L.nf[1], L.nc[1], L.cc[1], L.f_base, fy, wy);
	T acc = T(0);
#pragma unroll
	for (int k1 = 0; k1 < kRTaps; ++k1) {
		if (wy[k1] == T(0)) { continue; }
		const T* row = fine + static_cast<int64_t>(fy[k1]) * L.nf[0];
		T r = T(0);
#pragma unroll
		for (int k0 = 0; k0 < kRTaps; ++k0) {
			if (wx[k0] != T(0)) { r += wx[k0] * row[fx[k0]]; }
		}
		acc += wy[k1] * r;
	}
	coarse[static_cast<int64_t>(cy - L.c_base) * L.nc[0] + cx] = acc;
}

// The same restriction in two passes (P is a tensor product): first along x and y inside every LOCAL fine plane (ghost
// planes included) into tmp[fine plane][cy][cx], then along z.  Cell-centred axes have 4 - 5 taps: the one-pass kernel
// gathers up to 125 fine values per coarse point (171 us from 256^3 to 128^3), the two passes 16 + 4 (about 45 us).
template <typename T>
__global__ __launch_bounds__(kThreads) void k_restrict3_xy(LevelPair L, int planes, const T* __restrict__ fine, T* __restrict__ tmp)
{
	// one thread per (cx, cy, fine plane), the index flat (a row of 128 coarse points per 256-thread workgroup left half of
	// the threads idle); all 25 loads unconditional -- an index without weight is clamped into the row by restrict_taps
	int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (t >= static_cast<int64_t>(L.nc[0]) * L.nc[1] * planes) { return; }
	const int cx = static_cast<int>(t % L.nc[0]);
	t /= L.nc[0];
	const int cy = static_cast<int>(t % L.nc[1]), fz = static_cast<int>(t / L.nc[1]);  // fz: local plane
	int fx[kRTaps], fy[kRTaps];
	T   wx[kRTaps], wy[kRTaps];
	restrict_taps<T>(cx, L.nf[0], L.nc[0], L.cc[0], 0, fx, wx);
	restrict_taps<T>(cy, L.nf[1], L.nc[1], L.cc[1], 0, fy, wy);
	const T* plane = fine + static_cast<int64_t>(fz) * L.nf[0] * L.nf[1];
	T acc = T(0);
#pragma unroll
	for (int k1 = 0; k1 < kRTaps; ++k1) {
		const T* row = plane + static_cast<int64_t>(fy[k1]) * L.nf[0];
		T r = T(0);
#pragma unroll
		for (int k0 = 0; k0 < kRTaps; ++k0) { r += wx[k0] * row[fx[k0]]; }
		acc += wy[k1] * r;
	}
	tmp[(static_cast<int64_t>(fz) * L.nc[1] + cy) * L.nc[0] + cx] = acc;
}
// The same pass through LDS: a workgroup owns 64 x 8 coarse points of one fine plane and stages the 132 x 20 fine values
// they gather from with coalesced loads (the flat kernel's 25 loads per thread have a stride of two fine points between
// neighbouring lanes: 61 us from 256^3 to 128^3 for 84 MB, against 25 us here).  Same taps, same order of summation:
// the same bits.
template <typename T>
__global__ __launch_bounds__(kThreads) void k_restrict3_xy_tiled(LevelPair L, int planes, const T* __restrict__ fine, T* __restrict__ tmp)
{
	constexpr int CX = 64, CY = 8, FW = 2 * CX + 4, FH = 2 * CY + 4, PW = FW + 1;
	__shared__ T tile[FH][PW];
	const int cx0 = static_cast<int>(blockIdx.x) * CX, cy0 = static_cast<int>(blockIdx.y) * CY, fz = static_cast<int>(blockIdx.z);
	const int fx0 = 2 * cx0 - 2, fy0 = 2 * cy0 - 2;
	const T* plane = fine + static_cast<int64_t>(fz) * L.nf[0] * L.nf[1];
	for (int i = threadIdx.x; i < FW * FH; i += kThreads) {
		const int row = i / FW, col = i - row * FW;
		int gx = fx0 + col, gy = fy0 + row;
		gx = gx < 0 ? 0 : (gx >= L.nf[0] ? L.nf[0] - 1 : gx);  // (clamped values only ever meet taps without weight)
		gy = gy < 0 ? 0 : (gy >= L.nf[1] ? L.nf[1] - 1 : gy);
		tile[row][col] = plane[static_cast<int64_t>(gy) * L.nf[0] + gx];
	}
	__syncthreads();
	const int tx = threadIdx.x % CX, ty = threadIdx.x / CX;  // ty: 0 .. 3, two coarse rows per thread
	const int cx = cx0 + tx;
	if (cx >= L.nc[0]) { return; }
	int fx[kRTaps];
	T   wx[kRTaps];
	restrict_taps<T>(cx, L.nf[0], L.nc[0], L.cc[0], fx0, fx, wx);
#pragma unroll
	for (int h = 0; h < 2; ++h) {
		const int cy = cy0 + ty + 4 * h;
		if (cy >= L.nc[1]) { continue; }
		int fy[kRTaps];
		T   wy[kRTaps];
		restrict_taps<T>(cy, L.nf[1], L.nc[1], L.cc[1], fy0, fy, wy);
		T acc = T(0);
#pragma unroll
		for (int k1 = 0; k1 < kRTaps; ++k1) {
			T r = T(0);
#pragma unroll
			for (int k0 = 0; k0 < kRTaps; ++k0) { r += wx[k0] * tile[fy[k1]][fx[k0]]; }
			acc += wy[k1] * r;
		}
		tmp[(static_cast<int64_t>(fz) * L.nc[1] + cy) * L.nc[0] + cx] = acc;
	}
}
template <typename T>
__global__ __launch_bounds__(kThreads) void k_restrict3_z(LevelPair L, const T* __restrict__ tmp, T* __restrict__ coarse)
{
	const int64_t cplane = static_cast<int64_t>(L.nc[0]) * L.nc[1];
	const int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (t >= cplane * L.c_planes) { return; }
	const int64_t o  = t % cplane;
	const int     cz = static_cast<int>(t / cplane) + L.c_z0;  // global plane
	int fz[kRTaps];
	T   wz[kRTaps];
	restrict_taps<T>(cz, L.nf[2], L.nc[2], L.cc[2], L.f_base, fz, wz);
	T acc = T(0);
#pragma unroll
	for (int k = 0; k < kRTaps; ++k) {
		if (wz[k] != T(0)) { acc += wz[k] * tmp[fz[k] * cplane + o]; }  // (a plane without weight may lie outside the slab)
	}
	coarse[(static_cast<int64_t>(cz - L.c_base)) * cplane + o] = acc;
}

// tmp (3-D, optional): a work array of the fine level with room for (local fine planes) x nc[1] x nc[0] values -- any of
// the fine level's lattice vectors will do -- selects the two-pass form; f_local_planes = the fine level's local planes
template <typename T>
void launch_restrict(const LevelPair& L, const T* fine, T* coarse, hipStream_t st, T* tmp = nullptr, int f_local_planes = 0)
{
	if (L.ndim == 3 && tmp && f_local_planes > 0 && !test_switch("FI_ONE_PASS_RESTRICT")) {
		const int64_t n_xy = static_cast<int64_t>(L.nc[0]) * L.nc[1] * f_local_planes, n_z = static_cast<int64_t>(L.nc[0]) * L.nc[1] * L.c_planes;
		if (L.nc[0] >= 32 && f_local_planes <= 65535 && !test_switch("FI_FLAT_RESTRICT")) {
			hipLaunchKernelGGL((k_restrict3_xy_tiled<T>), dim3((L.nc[0] + 63) / 64, (L.nc[1] + 7) / 8, f_local_planes), dim3(kThreads), 0, st,
			                   L, f_local_planes, fine, tmp);
		} else {
			hipLaunchKernelGGL((k_restrict3_xy<T>), dim3(static_cast<unsigned>((n_xy + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, L,
			                   f_local_planes, fine, tmp);
		}
		if (n_z > 0) {
			hipLaunchKernelGGL((k_restrict3_z<T>), dim3(static_cast<unsigned>((n_z + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, L, tmp,
			                   coarse);
		}
		return;
	}
	if (L.ndim == 3) {
		hipLaunchKernelGGL((k_restrict3<T>), owned_grid(L.nc, L.ndim, L.c_planes), dim3(kThreads), 0, st, L, fine, coarse);
	} else if (L.ndim == 2) {
		hipLaunchKernelGGL((k_restrict2<T>), dim3((L.nc[0] + kThreads - 1) / kThreads, L.c_planes), dim3(kThreads), 0, st, L, fine, coarse);
	} else {
		hipLaunchKernelGGL((k_restrict<T>), owned_grid(L.nc, L.ndim, L.c_planes), dim3(kThreads), 0, st, L, fine, coarse);
	}
}

// r = b - q (q may be null: r = b);  d = alpha * Dinv r;  x = zero_x ? d : x + d
template <typename T>
__global__ __launch_bounds__(kThreads) void k_cheb_init(int64_t n, const T* __restrict__ b, const T* __restrict__ q,
                                                         const T* __restrict__ dinv, T* __restrict__ r,
                                                         T* __restrict__ d, T* __restrict__ x, T alpha, int zero_x)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		const T ri = q ? b[i] - q[i] : b[i];
		const T di = alpha * dinv[i] * ri;
		r[i] = ri;
		d[i] = di;
		x[i] = zero_x ? di : x[i] + di;
	}
}

// r = r_in - q;  d = c1 d + c2 Dinv r;  x = x_in + d.  Passes that nobody would read are skipped: r_in is the
// right-hand side itself on the first step of a smoother that started from zero (then x_in is d: x == d so far),
// and the last step of a polynomial stores neither r nor d.
template <typename T>
__global__ __launch_bounds__(kThreads) void k_cheb_iter(int64_t n, const T* __restrict__ q, const T* __restrict__ dinv,
                                                         const T* r_in, T* r, T* d, const T* x_in, T* x, T c1, T c2,
                                                         int store_rd)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		const T ri = r_in[i] - q[i];
		const T d0 = d[i];
		const T di = c1 * d0 + c2 * dinv[i] * ri;
		const T xi = (x_in == d ? d0 : x_in[i]) + di;
		if (store_rd) {
			r[i] = ri;
			d[i] = di;
		}
		x[i] = xi;
	}
}

// start of a smoother from zero: d = alpha Dinv b (x == d and r == b are not stored; a polynomial of degree 1
// stores x instead)
template <typename T>
__global__ __launch_bounds__(kThreads) void k_cheb_first(int64_t n, const T* __restrict__ b, const T* __restrict__ dinv,
                                                          T* __restrict__ d, T alpha)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		d[i] = alpha * dinv[i] * b[i];
	}
}

// the same with the bfloat16 scaling of the fused smoother (one polynomial: every step scales by the same diagonal)
template <typename T>
__global__ __launch_bounds__(kThreads) void k_cheb_first16(int64_t n, const T* __restrict__ b,
                                                            const unsigned short* __restrict__ dinv16, T* __restrict__ d, T alpha)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		d[i] = alpha * static_cast<T>(__uint_as_float(static_cast<unsigned int>(dinv16[i]) << 16)) * b[i];
	}
}

// r = b - q
template <typename T>
__global__ __launch_bounds__(kThreads) void k_sub(int64_t n, const T* __restrict__ b, const T* __restrict__ q,
                                                   T* __restrict__ r)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		r[i] = b[i] - q[i];
	}
}

// power method: v = Dinv q, partial of v.v
template <typename T>
__global__ __launch_bounds__(kThreads) void k_power_step(int64_t n, const T* __restrict__ q, const T* __restrict__ dinv,
                                                          T* __restrict__ v, double* __restrict__ partial)
{
	double acc[1] = {0};
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		const T w = dinv[i] * q[i];
		v[i] = w;
		acc[0] += static_cast<double>(w) * static_cast<double>(w);
	}
	double out[1];
	block_sum<1>(acc, out);
	if (threadIdx.x == 0) { partial[blockIdx.x] = out[0]; }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_seed(int64_t n, int64_t first, T* __restrict__ v)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		uint32_t h = static_cast<uint32_t>(first + i) * 2654435761u;
		h ^= h >> 15;
		h *= 2246822519u;
		h ^= h >> 13;
		v[i] = static_cast<T>(static_cast<double>(h & 0xFFFFu) / 65536.0 - 0.5);
	}
}

// mixed precision: r32 = r / s and z = s * z32 with s = ||r|| / ||b|| from the device-resident scalars (the V-cycle
// is linear, the scaling only keeps its fp32 operands near the size of b while r shrinks by ten decades).  The scale in
// use is CgScalars::tscale: the residual norm of the PREVIOUS step inside the loop (k_mg_step_mixed writes r32 before
// the new norm exists), of the current one after a restart.
__device__ inline double mixed_scale(const CgScalars* sc)
{
	return (sc->rr > 0.0 && sc->bb > 0.0) ? sqrt(sc->rr / sc->bb) : 1.0;
}
__device__ inline double twin_scale(const CgScalars* sc) { return sc->tscale > 0.0 ? sc->tscale : 1.0; }

// CG with a preconditioner: r -= alpha q, x += alpha p (p is still the direction of this step), partial r.r
template <typename T>
__global__ __launch_bounds__(kThreads) void k_mg_step(int64_t n, const CgScalars* __restrict__ sc, const T* __restrict__ p,
                                                       const T* __restrict__ q, T* __restrict__ x, T* __restrict__ r,
                                                       double* __restrict__ partial)
{
	if (sc->done) { return; }
	const T alpha = static_cast<T>(sc->alpha);
	double acc[1] = {0};
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		x[i] += alpha * p[i];
		const T ri = r[i] - alpha * q[i];
		r[i] = ri;
		acc[0] += static_cast<double>(ri) * static_cast<double>(ri);
	}
	double out[1];
	block_sum<1>(acc, out);
	if (threadIdx.x == 0) { partial[blockIdx.x] = out[0]; }
}

// partial of a.b
template <typename T>
__global__ __launch_bounds__(kThreads) void k_dot(int64_t n, const T* __restrict__ a, const T* __restrict__ b,
                                                   double* __restrict__ partial)
{
	double acc[1] = {0};
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		acc[0] += static_cast<double>(a[i]) * static_cast<double>(b[i]);
	}
	double out[1];
	block_sum<1>(acc, out);
	if (threadIdx.x == 0) { partial[blockIdx.x] = out[0]; }
}

// p = z + beta p
template <typename T>
__global__ __launch_bounds__(kThreads) void k_mg_direction(int64_t n, const CgScalars* __restrict__ sc,
                                                            const T* __restrict__ z, T* __restrict__ p, int first)
{
	const T beta = first ? T(0) : static_cast<T>(sc->beta);
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		p[i] = z[i] + beta * p[i];
	}
}

// Mixed precision (CG in fp64, V-cycle on the fp32 replica): the fp32 copy of the residual leaves k_mg_step with the
// update itself, and the fp64 copy of z = V(r) is never formed -- r.z and the new direction read the fp32 result.
// Per step 3 fp64 lattice passes less than k_mg_step + k_to_twin + k_from_twin + k_dot + k_mg_direction.
__global__ __launch_bounds__(kThreads) void k_mg_step_mixed(int64_t n, const CgScalars* __restrict__ sc,
                                                             const double* __restrict__ p, const double* __restrict__ q,
                                                             double* __restrict__ x, double* __restrict__ r,
                                                             float* __restrict__ r32, double* __restrict__ partial)
{
	if (sc->done) { return; }
	const double alpha = sc->alpha;
	const double inv = 1.0 / mixed_scale(sc);  // sc->rr is still the previous norm: k_mg_logic(kMgResid) records this scale
	double acc[1] = {0};
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		x[i] += alpha * p[i];
		const double ri = r[i] - alpha * q[i];
		r[i]   = ri;
		r32[i] = static_cast<float>(ri * inv);
		acc[0] += ri * ri;
	}
	double out[1];
	block_sum<1>(acc, out);
	if (threadIdx.x == 0) { partial[blockIdx.x] = out[0]; }
}
// partial of r . (s z32)
__global__ __launch_bounds__(kThreads) void k_dot_mixed(int64_t n, const CgScalars* __restrict__ sc, const double* __restrict__ r,
                                                         const float* __restrict__ z32, double* __restrict__ partial)
{
	const double s = twin_scale(sc);
	double acc[1] = {0};
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		acc[0] += r[i] * (s * static_cast<double>(z32[i]));
	}
	double out[1];
	block_sum<1>(acc, out);
	if (threadIdx.x == 0) { partial[blockIdx.x] = out[0]; }
}
// p = s z32 + beta p
__global__ __launch_bounds__(kThreads) void k_mg_direction_mixed(int64_t n, const CgScalars* __restrict__ sc,
                                                                  const float* __restrict__ z32, double* __restrict__ p, int first)
{
	const double beta = first ? 0.0 : sc->beta;
	const double s = twin_scale(sc);
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		p[i] = s * static_cast<double>(z32[i]) + beta * p[i];
	}
}

// scalar steps of the preconditioned recurrence (single block, thread 0)
enum MgPhase { kMgInitRz = 10, kMgAlpha = 11, kMgResid = 12, kMgBeta = 13, kMgInitRr = 14 };
__global__ __launch_bounds__(kThreads) void k_mg_logic(CgScalars* sc, const double* __restrict__ partial, int count,
                                                        int phase)
{
	if (sc->done && phase != kMgInitRr && phase != kMgInitRz) { return; }
	double acc[1] = {0};
	for (int i = threadIdx.x; i < count; i += kThreads) { acc[0] += partial[i]; }
	double out[1];
	block_sum<1>(acc, out);
	if (threadIdx.x != 0) { return; }
	const double s = partial ? out[0] : sc->sums[0];  // no partials: the sum over blocks and ranks is in sums[0]
	switch (phase) {
	case kMgInitRr:  // sums: r.r (partial 0) -- b.b was stored by the caller in sums[2]
		sc->rr = s;
		sc->true_rr = s;
		if (sc->bb == 0.0) { sc->bb = sc->sums[2]; sc->tol2 *= sc->bb; }
		sc->tscale = mixed_scale(sc);
		sc->done = 0;
		if (sc->bb == 0.0) {
			sc->done = 4;
		} else if (!(sc->rr > sc->tol2)) {
			sc->done = sc->restarts > 0 ? 5 : 1;
		} else if (sc->iter >= sc->max_iter) {
			sc->done = 3;
		}
		break;
	case kMgInitRz:
		sc->rz = s;
		if (!(s > 0.0) && sc->rr > sc->tol2) { sc->done = 2; }  // r.V(r) <= 0: the preconditioner is not positive definite
		break;
	case kMgAlpha:
		sc->pq    = s;
		sc->alpha = sc->rz / s;
		if (!(s > 0.0) || !isfinite(s)) { sc->done = 2; }
		break;
	case kMgResid:
		sc->tscale = mixed_scale(sc);  // of the norm k_mg_step_mixed scaled its fp32 residual by
		sc->rr = s;
		sc->iter += 1;
		if (!isfinite(s)) {
			sc->done = 2;
		} else if (!(s > sc->tol2)) {
			sc->done = 1;
		} else if (sc->iter >= sc->max_iter) {
			sc->done = 3;
		}
		break;
	case kMgBeta:
		sc->beta = s / sc->rz;
		sc->rz   = s;
		if (!(s > 0.0) || !isfinite(s)) { sc->done = 2; }  // (an indefinite or diverging V-cycle: see cg_run_mg)
		break;
	}
}

// r = b - q with the partials of r.r and b.b in the same pass (the start and the verification of V-cycle PCG on an
// undivided lattice: k_sub + two k_dot + their one-block sums were five launches and three more lattice passes)
template <typename T>
__global__ __launch_bounds__(kThreads) void k_resid_norms(int64_t n, const T* __restrict__ b, const T* __restrict__ q, T* __restrict__ r,
                                                           double* __restrict__ partial_rr, double* __restrict__ partial_bb)
{
	double acc[2] = {0, 0};
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		const T bi = b[i];
		const T ri = bi - q[i];
		r[i] = ri;
		acc[0] += static_cast<double>(ri) * static_cast<double>(ri);
		acc[1] += static_cast<double>(bi) * static_cast<double>(bi);
	}
	double out[2];
	block_sum<2>(acc, out);
	if (threadIdx.x == 0) {
		partial_rr[blockIdx.x] = out[0];
		partial_bb[blockIdx.x] = out[1];
	}
}
// b.b into sums[2] (where kMgInitRr expects it), in front of k_mg_logic(kMgInitRr) on the same stream
__global__ __launch_bounds__(kThreads) void k_sum_to_slot2(CgScalars* sc, const double* __restrict__ partial, int count)
{
	double acc[1] = {0};
	for (int i = threadIdx.x; i < count; i += kThreads) { acc[0] += partial[i]; }
	double out[1];
	block_sum<1>(acc, out);
	if (threadIdx.x == 0) { sc->sums[2] = out[0]; }
}

int mg_degree() { const char* e = tuning_switch("FI_MG_DEGREE"); return e && atoi(e) > 0 ? atoi(e) : 4; }  // config 3 / 5: degree 2 -> 4 halves the solve time
double mg_ratio() { const char* e = tuning_switch("FI_MG_RATIO"); return e && atof(e) > 1 ? atof(e) : 10.0; }

template <typename T>
void mg_alloc(fi_ctx* c)
{
	ensure_vectors(c);
	const size_t bytes = sizeof(T) * c->g.nloc;
	const bool fresh = c->mg_b.bytes < bytes;
	c->mg_b.alloc(bytes);
	c->mg_x.alloc(bytes);
	c->mg_r.alloc(bytes);
	c->mg_d.alloc(bytes);
	if (fresh) {  // ghost planes outside the lattice are never written: keep them finite
		FI_HIP_TRY(hipMemsetAsync(c->mg_b.p, 0, bytes, c->stream));
		FI_HIP_TRY(hipMemsetAsync(c->mg_x.p, 0, bytes, c->stream));
		FI_HIP_TRY(hipMemsetAsync(c->mg_r.p, 0, bytes, c->stream));
		FI_HIP_TRY(hipMemsetAsync(c->mg_d.p, 0, bytes, c->stream));
	}
}

// The multigrid drivers work on a rank set like cg_run: one member = an undivided lattice or this process's
// slab (halo planes and sums over RCCL), several members = the loop-back group.  A vector is named by its
// fi_ctx member so that every member's copy can be addressed: base pointer for the operator / transfer
// kernels (ghost planes included), owned part for the elementwise ones.
using Vec = DevBuf fi_ctx::*;
template <typename T>
T* vbase(fi_ctx* c, Vec v) { return (c->*v).template as<T>(); }
template <typename T>
T* vown(fi_ctx* c, Vec v) { return (c->*v).template as<T>() + c->g.own_first; }

template <typename T>
__global__ __launch_bounds__(kThreads) void k_add_vec(int64_t n, const T* __restrict__ d, T* __restrict__ x)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		x[i] += d[i];
	}
}

// A loop-back group's copies of a replicated level: every member holds the WHOLE lattice and does everything on its own,
// like the ranks of a real decomposition do (R.size() == 1 there, and nranks == 1 switches every collective off).
bool replicated_copies(const RankSet& R) { return R.size() > 1 && R[0]->nranks == 1; }
template <typename Fn>
void for_each_copy(RankSet& R, Fn fn)
{
	if (replicated_copies(R)) {
		for (fi_ctx* c : R) {
			RankSet one{c};
			fn(one);
		}
	} else {
		fn(R);
	}
}

// Sum of vector `v` (whole lattices: the replicated level below a slab level) over the ranks, in place: every rank has
// restricted its slab's residual into its own coarse planes, the others are zero.
template <typename T>
void sum_over_ranks(RankSet& Rfine, RankSet& Rc, DevBuf fi_ctx::*v)
{
	const int64_t n = Rc[0]->g.nloc;
	if (Rc.size() > 1) {  // loop-back group: add up into member 0 in member order, copy back
		fi_ctx* c0 = Rc[0];
		for (size_t m = 1; m < Rc.size(); ++m) {
			hipLaunchKernelGGL((k_add_vec<T>), dim3(stream_blocks(n)), dim3(kThreads), 0, c0->stream, n, (Rc[m]->*v).as<T>(), (c0->*v).as<T>());
		}
		for (size_t m = 1; m < Rc.size(); ++m) {
			FI_HIP_TRY(hipMemcpyAsync((Rc[m]->*v).p, (c0->*v).p, sizeof(T) * n, hipMemcpyDeviceToDevice, c0->stream));
		}
	} else if (Rfine[0]->nranks > 1) {
		allreduce_sum_vec(Rfine[0], (Rc[0]->*v).p, n, sizeof(T) == 8);
	}
}

RankSet coarse_of(const RankSet& R)
{
	RankSet r;
	for (fi_ctx* c : R) { r.push_back(c->coarse); }
	return r;
}

void apply_all(RankSet& R, Vec in, Vec out, bool partials)
{
	halo_exchange(R, in);
	for (fi_ctx* c : R) { apply_AtA(c, (c->*in).p, (c->*out).p, partials ? c->partial.as<double>() : nullptr); }
}

// global index of the first owned unknown (seeds of the power method must not depend on the decomposition)
int64_t global_first(const fi_ctx* c)
{
	const Geom& g = c->g;
	const int a = g.ndim - 1;
	int64_t plane = 1;
	for (int d = 0; d < a; ++d) { plane *= g.gn[d]; }
	return plane * (g.off[a] + g.own_lo[a]);
}

// largest eigenvalue of Dinv*AtA by the power method (10 steps, unnormalised: growth <= 8^10, fine in fp32)
template <typename T>
void estimate_lambda(RankSet& R)
{
	for (fi_ctx* c : R) { mg_alloc<T>(c); }
	CgScalars init{};
	reset_scalars(R, init);
	auto nbv = [](fi_ctx* c) { return stream_blocks(c->g.nown); };
	for (fi_ctx* c : R) {
		hipLaunchKernelGGL((k_seed<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, global_first(c),
		                   vown<T>(c, &fi_ctx::mg_d));
	}
	const int steps = 10;
	double sums[2] = {0, 0};
	for (int k = 0; k < steps; ++k) {
		apply_all(R, &fi_ctx::mg_d, &fi_ctx::q, false);
		for (fi_ctx* c : R) {
			hipLaunchKernelGGL((k_power_step<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, vown<T>(c, &fi_ctx::q),
			                   vown<T>(c, &fi_ctx::dinv), vown<T>(c, &fi_ctx::mg_d), c->partial.as<double>());
		}
		if (k >= steps - 2) {  // |v|^2 after the last two steps, summed over all slabs
			reduce_phase(R, 1, nbv, nbv, -1);
			fi_ctx* c0 = R[0];
			FI_HIP_TRY(hipMemcpyAsync(c0->scal_host, c0->scal.p, sizeof(CgScalars), hipMemcpyDeviceToHost, c0->stream));
			FI_HIP_TRY(hipStreamSynchronize(c0->stream));
			sums[k - (steps - 2)] = c0->scal_host->sums[0];
		}
	}
	const double lambda = (sums[0] > 0 && sums[1] > 0) ? std::sqrt(sums[1] / sums[0]) : 2.0;
	for (fi_ctx* c : R) { c->lambda_max = lambda; }
}

// The smoother through the marching kernel's epilogue (fi_stencil.hip, ChebEpi mode 2): the same polynomial written as
// a three-term recurrence in the iterates themselves,
//     x_{k+1} = (1 + c1) x_k - c1 x_{k-1} + c2 Dinv (b - A x_k),   c1 = rho_k rho_{k-1}, c2 = 2 rho_k / delta
// (d_k = x_{k+1} - x_k of the form below), so that a step is ONE launch that reads x_k, x_{k-1}, b, Dinv and the cell
// records and writes x_{k+1}: 5 lattice passes instead of the 10 of apply + k_cheb_iter.  The iterates rotate through
// x, mg_d and mg_r; the buffer that ends up holding the result is swapped into x.
bool smooth_fused_ok(const RankSet& R)  // (stencil_full_epi_available knows which precisions a context's kernel covers)
{
	if (test_switch("FI_NO_FUSED_SMOOTHER")) { return false; }  // tests compare the two forms of the smoother
	for (const fi_ctx* c : R) {
		if (!stencil_full_epi_available(c)) { return false; }
	}
	return true;
}
void swap_vectors(RankSet& R, Vec a, Vec b)
{
	if (a == b) { return; }
	for (fi_ctx* c : R) {
		(c->*a).swap(c->*b);
	}
}
template <typename T>
void cheb_smooth_fused(RankSet& R, Vec b, Vec x, int degree, double ratio, bool from_zero)
{
	const double hi = 1.1 * R[0]->lambda_max, lo = hi / ratio;
	const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
	const Vec ring[3] = {x, &fi_ctx::mg_d, &fi_ctx::mg_r};
	int cur = 0, prev = -1;  // ring positions of x_k and x_{k-1} (-1: x_{k-1} = 0, or the first step of a smoother that starts at x)
	auto step = [&](double a, double c1, double c2) {
		const int next = prev < 0 ? (cur + 1) % 3 : 3 - cur - prev;
		halo_exchange(R, ring[cur]);
		for (fi_ctx* c : R) {
			stencil_full_step(c, (c->*ring[cur]).p, prev < 0 ? nullptr : (c->*ring[prev]).p, (c->*b).p, false,
			                  (c->*ring[next]).p, a, c1, c2);
		}
		prev = cur;
		cur  = next;
	};
	bool have_prev = false;  // x_{k-1} is a vector (not the zero start)
	if (from_zero) {
		for (fi_ctx* c : R) {  // x_1 = Dinv b / theta
			hipLaunchKernelGGL((k_cheb_first16<T>), dim3(stream_blocks(c->g.nown)), dim3(kThreads), 0, c->stream, c->g.nown,
			                   vown<T>(c, b), c->dinv16.as<unsigned short>() + c->g.own_first, vown<T>(c, x),
			                   static_cast<T>(1.0 / theta));
		}
	} else {
		step(1.0, 0.0, 1.0 / theta);  // x_1 = x_0 + Dinv (b - A x_0) / theta
		have_prev = true;
	}
	double rho = 1.0 / sigma;
	for (int k = 1; k < degree; ++k) {
		const double rho_new = 1.0 / (2.0 * sigma - rho);
		const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
		if (!have_prev) { prev = -1; }  // x_0 = 0: its term drops out, a = 1 + c1 stays
		step(1.0 + c1, c1, c2);
		have_prev = true;
		rho = rho_new;
	}
	swap_vectors(R, x, ring[cur]);
}

// degree-k Chebyshev smoothing of AtA x = b on [lmax/ratio, 1.1 lmax]; from_zero: x starts at 0
template <typename T>
void cheb_smooth(RankSet& R, Vec b, Vec x, int degree, double ratio, bool from_zero)
{
	if (smooth_fused_ok(R)) {
		cheb_smooth_fused<T>(R, b, x, degree, ratio, from_zero);
		return;
	}
	const double hi = 1.1 * R[0]->lambda_max, lo = hi / ratio;
	const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
	auto nbv = [](fi_ctx* c) { return stream_blocks(c->g.nown); };
	if (from_zero) {
		// x == d and r == b until the first update: only d is written (straight into x when the polynomial ends here)
		for (fi_ctx* c : R) {
			hipLaunchKernelGGL((k_cheb_first<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, vown<T>(c, b),
			                   vown<T>(c, &fi_ctx::dinv), degree > 1 ? vown<T>(c, &fi_ctx::mg_d) : vown<T>(c, x),
			                   static_cast<T>(1.0 / theta));
		}
	} else {
		apply_all(R, x, &fi_ctx::q, false);
		for (fi_ctx* c : R) {
			hipLaunchKernelGGL((k_cheb_init<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, vown<T>(c, b),
			                   vown<T>(c, &fi_ctx::q), vown<T>(c, &fi_ctx::dinv), vown<T>(c, &fi_ctx::mg_r),
			                   vown<T>(c, &fi_ctx::mg_d), vown<T>(c, x), static_cast<T>(1.0 / theta), 0);
		}
	}
	double rho = 1.0 / sigma;
	for (int k = 1; k < degree; ++k) {
		const double rho_new = 1.0 / (2.0 * sigma - rho);
		apply_all(R, &fi_ctx::mg_d, &fi_ctx::q, false);
		const bool first = from_zero && k == 1, last = k == degree - 1;
		for (fi_ctx* c : R) {
			T* d = vown<T>(c, &fi_ctx::mg_d);
			T* r = vown<T>(c, &fi_ctx::mg_r);
			hipLaunchKernelGGL((k_cheb_iter<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, vown<T>(c, &fi_ctx::q),
			                   vown<T>(c, &fi_ctx::dinv), first ? static_cast<const T*>(vown<T>(c, b)) : static_cast<const T*>(r), r,
			                   d, first ? static_cast<const T*>(d) : static_cast<const T*>(vown<T>(c, x)), vown<T>(c, x),
			                   static_cast<T>(rho_new * rho), static_cast<T>(2.0 * rho_new / delta), last ? 0 : 1);
		}
		rho = rho_new;
	}
}

// ---- the V-cycle's polynomial smoother ----------------------------------------------------------------------------
// M = p_d(D^-1 A^) D^-1 with A^ = A_model + f diag(A_data), D = diag(A^): the polynomial of cg_run_poly's preconditioner
// (same kernel, same recurrence: fi_stencil.hip ChebEpi mode 0) over an operator that BOUNDS the full one -- a cell's
// block is sum a a^T <= 2^D diag(a_i^2), so with f = 2^D, A <= A^ and the spectrum of M A stays below 1 + eps < 2: a
// convergent smoother whatever the data; f = mg_safe = 4 is the measured optimum (tools/proto_cc.py: 11 / 11 / 13
// iterations for f = 2 / 4 / 8 on config 4, f = 1 diverges on dense data).  A smoothing pass is
//     pre  (from zero):  x = M b                                   d - 1 plain launches of (2.5 .. 4.5) lattice passes
//     post:              x += M (b - A x)                          one full apply with the residual epilogue + the same
// against 2 d launches of the fused (data-cell) kernel with the epilogue for the Chebyshev smoother in A itself.
int    mg_poly_terms(const fi_ctx* c)
{
	if (c->level > 0) {
		const char* ec = tuning_switch("FI_MG_COARSE_TERMS");
		if (ec && atoi(ec) > 1) { return atoi(ec); }
	}
	const char* e = tuning_switch("FI_MG_TERMS");
	return e && atoi(e) > 0 ? atoi(e) : c->mg_terms;
}
double mg_poly_ratio(const fi_ctx* c) { const char* e = tuning_switch("FI_MG_RATIO"); return e && atof(e) > 1 ? atof(e) : c->mg_pratio; }
template <typename T>
bool poly_smoother_ok(const RankSet& R)
{
	if (sizeof(T) != 4 || test_switch("FI_MG_FULL_SMOOTHER")) { return false; }  // tests compare the two smoothers
	for (const fi_ctx* c : R) {
		// 3-D levels only: the 2-D tile kernel applies its cells in the same single launch, so the Chebyshev smoother in
		// the full operator costs no more per step there and is the better smoother (config 3: 13 iterations against 38)
		if (c->mg_smoother != 1 || !c->value_rows_only || !c->march.valid || !stencil_full_epi_available(c) || c->generic.ntrip != 0 || c->any_trip) {
			return false;
		}
	}
	return true;
}

// z = M r through the work vectors za / zb; returns the one that holds the result (ghost planes not exchanged)
template <typename T>
Vec poly_chain(RankSet& R, Vec r, Vec za, Vec zb, double* chain_bytes = nullptr, int* chain_launches = nullptr)
{
	fi_ctx* c0 = R[0];
	const int    terms = mg_poly_terms(c0);
	const double lam = c0->poly_lambda > 1.0 ? c0->poly_lambda : 1.0;
	const double hi = 1.1 * lam, lo = hi / mg_poly_ratio(c0);
	const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
	const bool single = R.size() == 1 && c0->nranks == 1;
	bool ghosts_scaled = true;
	for (const fi_ctx* c : R) { ghosts_scaled = ghosts_scaled && c->scaling_ghosts; }
	const bool pro = (single || ghosts_scaled) && terms > 2 && c0->march.valid && !test_switch("FI_NO_Z0_ON_LOAD");
	auto region2 = [](fi_ctx* c) { return c->partial.as<double>() + 2 * static_cast<size_t>(c->max_blocks); };
	if (!pro) {
		for (fi_ctx* c : R) {  // z_0 = Dinv r / theta
			hipLaunchKernelGGL((k_cheb_first16<T>), dim3(stream_blocks(c->g.nown)), dim3(kThreads), 0, c->stream, c->g.nown,
			                   vown<T>(c, r), c->dinv16s.as<unsigned short>() + c->g.own_first, vown<T>(c, za),
			                   static_cast<T>(1.0 / theta));
		}
	}
	if (chain_bytes) {  // z, z_prev, r in, z_new out + the bfloat16 scaling; the first step has no z_prev (formed on load: no z
		*chain_bytes = 0;   // either), the second recomputes its z_prev from r: 2.5 / 3.5 / 4.5 lattice passes in fp32
		for (int j = 1; j < terms; ++j) {
			const double vecs = j == 1 ? (pro ? 2.0 : 3.0) : (j == 2 ? 3.0 : 4.0);
			*chain_bytes += (static_cast<double>(sizeof(T)) * vecs + 2.0) * static_cast<double>(c0->g.nown);
		}
	}
	if (chain_launches) { *chain_launches = terms - 1; }
	Vec zin = za, zout = zb;
	double rho = 1.0 / sigma;
	for (int k = 1; k < terms; ++k) {
		const double rho_new = 1.0 / (2.0 * sigma - rho);
		const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
		const bool   first_on_load = pro && k == 1;
		halo_exchange(R, first_on_load ? r : zin);
		for (fi_ctx* c : R) {
			const unsigned short* sc = c->dinv16s.as<unsigned short>();
			if (first_on_load) {
				stencil_cheb_step(c, (c->*r).p, nullptr, (c->*r).p, (c->*zout).p, c1, c2, region2(c), 0, 0.0, 1.0 / theta, sc);
			} else {
				// the second step's z_prev is z_0 = Dinv r / theta, recomputed from r and the scaling
				stencil_cheb_step(c, (c->*zin).p, k == 1 ? nullptr : (c->*zout).p, (c->*r).p, (c->*zout).p, c1, c2, region2(c), 0,
				                  k == 2 ? 1.0 / theta : 0.0, 0.0, sc);
			}
		}
		std::swap(zin, zout);
		rho = rho_new;
	}
	return zin;
}

// ---- the V-cycle of the hierarchy's tail as a program of the small-level engine (fi_tail.h) ----------------------------
// The same cycle as vcycle() below -- the same smoothers with the same constants, the same transfers -- written out as the
// stages of ONE cooperative launch.  Vectors: rhs b and result x as the caller names them on the top level, mg_b / mg_x
// below; work vectors mg_r (residual), mg_d and q (the polynomials' iterates: x itself is only ever the final target of a
// stage, never an intermediate, so the program's pointers stay valid from cycle to cycle).
template <typename T>
bool poly_smoother_ok(const RankSet& R);

struct TailProgram {
	std::vector<TailOp> ops;
	std::vector<fi_ctx*> chain;
	bool ok = true;

	void op(int kind, int level, const float* a, const float* b, const float* c, float* out, float* acc, const unsigned short* scale,
	        double s0 = 0, double s1 = 0, double s2 = 0)
	{
		ops.push_back(TailOp{kind, level, a, b, c, out, acc, scale, static_cast<float>(s0), static_cast<float>(s1), static_cast<float>(s2), 0});
	}
	// target (+)= M r: the polynomial in A_model + f diag(A_data) (poly_chain); out: target of the last step, acc: accumulate
	void poly_ops(int l, const float* r, float* out, float* acc)
	{
		fi_ctx* c = chain[l];
		const int    terms = mg_poly_terms(c);
		const double lam = c->poly_lambda > 1.0 ? c->poly_lambda : 1.0;
		const double hi = 1.1 * lam, lo = hi / mg_poly_ratio(c);
		const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
		const unsigned short* sc = c->dinv16s.as<unsigned short>();
		float* W[2] = {c->mg_d.as<float>(), c->q.as<float>()};
		if (terms < 2) { ok = false; return; }
		op(kTailScale, l, r, nullptr, nullptr, W[0], nullptr, sc, 1.0 / theta);
		int cur = 0;
		double rho = 1.0 / sigma;
		for (int k = 1; k < terms; ++k) {
			const double rho_new = 1.0 / (2.0 * sigma - rho);
			const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
			const bool   last = k == terms - 1;
			op(kTailPolyStep, l, W[cur], k == 1 ? nullptr : W[1 - cur], r, last ? out : W[1 - cur], last ? acc : nullptr, sc, 1.0 + c1,
			   k == 1 ? 0.0 : c1, c2);
			cur = 1 - cur;
			rho = rho_new;
		}
	}
	// degree-k Chebyshev smoothing in the full operator (cheb_smooth_fused): x starts at zero / at its present value
	void cheb_ops(int l, const float* b, float* x, int degree, double ratio, bool from_zero)
	{
		fi_ctx* c = chain[l];
		const double hi = 1.1 * c->lambda_max, lo = hi / ratio;
		const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
		const unsigned short* sc = c->dinv16.as<unsigned short>();
		float* W[2] = {c->mg_d.as<float>(), c->mg_r.as<float>()};
		if (degree < 2 || !(c->lambda_max > 0)) { ok = false; return; }
		const float* zk = nullptr;    // x_k
		const float* zp = nullptr;    // x_{k-1} (null: zero, or no such term)
		int steps = degree - 1;       // recurrence steps after the first term
		if (from_zero) {
			op(kTailScale, l, b, nullptr, nullptr, W[0], nullptr, sc, 1.0 / theta);   // x_1 = Dinv b / theta
			zk = W[0];
		} else {
			op(kTailChebStep, l, x, nullptr, b, W[0], nullptr, sc, 1.0, 0.0, 1.0 / theta);  // x_1 = x_0 + Dinv (b - A x_0) / theta
			zk = W[0];
			zp = x;
		}
		double rho = 1.0 / sigma;
		for (int k = 1; k <= steps; ++k) {
			const double rho_new = 1.0 / (2.0 * sigma - rho);
			const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
			const bool   last = k == steps;
			float* out = last ? x : (zk == W[0] ? W[1] : W[0]);   // (x_{k-1}'s buffer may be overwritten: it is read at the point itself only)
			op(kTailChebStep, l, zk, zp, b, out, nullptr, sc, 1.0 + c1, zp ? c1 : 0.0, c2);
			zp = zk;
			zk = out;
			rho = rho_new;
		}
	}
	void residual(int l, const float* x, const float* b, float* out) { op(kTailResidual, l, x, nullptr, b, out, nullptr, nullptr); }

	void cycle(int l, const float* b, float* x)
	{
		fi_ctx* c = chain[l];
		RankSet one{c};
		const bool last = l + 1 == static_cast<int>(chain.size());
		const bool poly = poly_smoother_ok<float>(one);
		const int    deg = mg_degree();
		const double ratio = mg_ratio();
		float* r = c->mg_r.as<float>();
		if (c->lumped) { ok = false; return; }
		if (last) {
			if (poly && c->dinv16s_valid && c->data_pinned && !test_switch("FI_MG_COARSEST_CHEB")) {
				poly_ops(l, b, x, nullptr);
				residual(l, x, b, r);
				poly_ops(l, r, nullptr, x);
			} else {
				cheb_ops(l, b, x, 4 * deg + 4, 10.0 * ratio, true);
			}
			return;
		}
		fi_ctx* co = chain[l + 1];
		if (poly) {
			poly_ops(l, b, x, nullptr);
		} else {
			cheb_ops(l, b, x, deg, ratio, true);
		}
		residual(l, x, b, r);
		op(kTailRestrict, l, r, nullptr, nullptr, co->mg_b.as<float>(), nullptr, nullptr);
		cycle(l + 1, co->mg_b.as<float>(), co->mg_x.as<float>());
		op(kTailProlongAdd, l, co->mg_x.as<float>(), nullptr, nullptr, x, nullptr, nullptr);
		if (poly) {
			residual(l, x, b, r);
			poly_ops(l, r, nullptr, x);
		} else {
			cheb_ops(l, b, x, deg, ratio, false);
		}
	}
};

// x = V(b) on the tail that starts at R's level, in one launch; false: the level is not the engine's (the caller recurses)
template <typename T>
bool tail_vcycle(RankSet& R, Vec b, Vec x)
{
	if (sizeof(T) != 4 || R.size() != 1) { return false; }
	fi_ctx* c = R[0];
	if (c->level == 0 || !c->tail_ok || c->nranks != 1) { return false; }
	const void* bp = (c->*b).p;
	const void* xp = (c->*x).p;
	if (!c->tail_prog_valid || c->tail_prog_b != bp || c->tail_prog_x != xp) {
		TailProgram P;
		for (fi_ctx* l = c; l; l = l->coarse) { P.chain.push_back(l); }
		P.cycle(0, static_cast<const float*>(bp), static_cast<float*>(const_cast<void*>(xp)));
		if (!P.ok || P.ops.empty()) {
			c->tail_ok = false;  // (a setting the engine does not cover: the tiled kernels run this hierarchy)
			return false;
		}
		std::vector<unsigned char> blob(sizeof(TailLevel) * kTailMaxLevels + sizeof(TailOp) * P.ops.size());
		TailLevel* lv = reinterpret_cast<TailLevel*>(blob.data());
		int64_t widest = 0;
		for (size_t k = 0; k < P.chain.size(); ++k) {
			lv[k] = tail_level_of(P.chain[k]);
			if (k + 1 < P.chain.size()) { lv[k].to_coarse = level_pair(P.chain[k], P.chain[k + 1]); }
			widest = lv[k].nn > widest ? lv[k].nn : widest;
		}
		std::memcpy(blob.data() + sizeof(TailLevel) * kTailMaxLevels, P.ops.data(), sizeof(TailOp) * P.ops.size());
		c->tail_prog.alloc(blob.size());
		FI_HIP_TRY(hipMemcpyAsync(c->tail_prog.p, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream));
		FI_HIP_TRY(hipStreamSynchronize(c->stream));  // (the host buffer dies here)
		c->tail_nlev       = static_cast<int>(P.chain.size());
		c->tail_nops       = static_cast<int>(P.ops.size());
		c->tail_widest     = widest;
		c->tail_prog_b     = bp;
		c->tail_prog_x     = xp;
		c->tail_prog_valid = true;
	}
	tail_run(c, c->tail_prog.p, c->tail_nlev, c->tail_nops, c->tail_widest);
	return true;
}

// x = V(b) on the level of R.  Over slabs every level is a slab decomposition of its own (coarse plane k lives
// with fine plane 2k): restriction reads one ghost plane of the fine residual, interpolation one of the coarse
// correction.
template <typename T>
void vcycle(RankSet& R, Vec b, Vec x)
{
	if (replicated_copies(R)) {  // the replicated tail in a loop-back group: every member's copy on its own
		for_each_copy(R, [&](RankSet& one) { vcycle<T>(one, b, x); });
		return;
	}
	if (tail_vcycle<T>(R, b, x)) { return; }  // the small-level engine: this level and all below it in one launch
	const int deg = mg_degree();
	const double ratio = mg_ratio();
	if (tuning_switch("FI_MG_POLY")) {  // experiment: the polynomial alone as the preconditioner, no coarse correction
		cheb_smooth<T>(R, b, x, deg, ratio, true);
		return;
	}
	const bool poly = poly_smoother_ok<T>(R);
	for (const fi_ctx* c : R) {
		FI_REQUIRE(!c->lumped || poly, FI_ERR_UNSUPPORTED, "a lumped replica smooths with the polynomial only");
	}
	auto residual = [&]() {  // mg_r = b - A x (a lumped replica: its own operator, A_model + diag(dlump))
		halo_exchange(R, x);
		for (fi_ctx* c : R) {
			if (c->lumped) {
				stencil_lumped_residual(c, (c->*x).p, (c->*b).p, c->mg_r.p);
			} else {
				stencil_full_step(c, (c->*x).p, nullptr, (c->*b).p, true, c->mg_r.p, 0.0, 0.0, 0.0);
			}
		}
	};
	auto post_smooth = [&]() {  // x += M (b - A x)
		residual();
		const Vec d = poly_chain<T>(R, &fi_ctx::mg_r, &fi_ctx::mg_d, &fi_ctx::q);
		for (fi_ctx* c : R) {
			hipLaunchKernelGGL((k_add_vec<T>), dim3(stream_blocks(c->g.nown)), dim3(kThreads), 0, c->stream, c->g.nown, vown<T>(c, d),
			                   vown<T>(c, x));
		}
	};
	if (!R[0]->coarse) {
		// coarsest level: two sweeps of the polynomial smoother where the data pin every point (fi_ctx::data_pinned: config
		// 4 -- as good as an exact solve there, tools/proto_cc.py, and 7 launches instead of 20); else a longer polynomial
		// over a wider band, in the full operator: an SDF's coarsest level still has weakly held global modes (97 instead of
		// 37 iterations on the 40 x 32 x 48 test problem with the sweeps).
		if (poly && R[0]->dinv16s_valid && R[0]->data_pinned && !test_switch("FI_MG_COARSEST_CHEB")) {
			swap_vectors(R, x, poly_chain<T>(R, b, x, &fi_ctx::mg_d));
			post_smooth();
			return;
		}
		cheb_smooth<T>(R, b, x, 4 * deg + 4, 10.0 * ratio, true);
		return;
	}
	RankSet Rc = coarse_of(R);
	// a slab level above the replicated tail: every rank restricts into its own coarse planes of the WHOLE coarse lattice,
	// the parts are summed over the ranks (the one collective of the tail), the interpolation needs no exchange
	const bool junction = Rc[0]->replicated && !R[0]->replicated && R[0]->nranks > 1;
	if (poly) {
		// x = M b.  The finest level's chain is timed for the first cycles of a solve (cg_run_mg sets the budget): all its
		// launches between one pair of event records, like the polynomial PCG's samples
		fi_ctx* c0 = R[0];
		const bool sample = c0->level == 0 && c0->prec_budget > 0 && static_cast<int>(c0->ev_prec.size()) >= 2 * (c0->prec_taken + 1);
		if (sample) { FI_HIP_TRY(hipEventRecord(c0->ev_prec[2 * c0->prec_taken], c0->stream)); }
		swap_vectors(R, x, poly_chain<T>(R, b, x, &fi_ctx::mg_d, sample ? &c0->prec_chain_bytes : nullptr,
		                                 sample ? &c0->prec_chain_launches : nullptr));
		if (sample) {
			FI_HIP_TRY(hipEventRecord(c0->ev_prec[2 * c0->prec_taken + 1], c0->stream));
			++c0->prec_taken;
			--c0->prec_budget;
		}
		residual();
		halo_exchange(R, &fi_ctx::mg_r);
		for (size_t i = 0; i < R.size(); ++i) {
			const LevelPair L = level_pair(R[i], Rc[i]);
			if (junction) { FI_HIP_TRY(hipMemsetAsync(Rc[i]->mg_b.p, 0, sizeof(T) * Rc[i]->g.nloc, R[i]->stream)); }
			launch_restrict<T>(L, vbase<T>(R[i], &fi_ctx::mg_r), vbase<T>(Rc[i], &fi_ctx::mg_b), R[i]->stream, vbase<T>(R[i], &fi_ctx::q),
			                   R[i]->g.n[2]);
		}
		if (junction) { sum_over_ranks<T>(R, Rc, &fi_ctx::mg_b); }
		vcycle<T>(Rc, &fi_ctx::mg_b, &fi_ctx::mg_x);
		halo_exchange(Rc, &fi_ctx::mg_x);
		for (size_t i = 0; i < R.size(); ++i) {
			const LevelPair L = level_pair(R[i], Rc[i]);
			launch_prolong<T>(L, vbase<T>(Rc[i], &fi_ctx::mg_x), vbase<T>(R[i], x), 1, R[i]->stream);
		}
		post_smooth();
		return;
	}
	cheb_smooth<T>(R, b, x, deg, ratio, true);
	if (smooth_fused_ok(R)) {  // mg_r = b - A x in one launch
		halo_exchange(R, x);
		for (fi_ctx* c : R) { stencil_full_step(c, (c->*x).p, nullptr, (c->*b).p, true, c->mg_r.p, 0.0, 0.0, 0.0); }
	} else {
		apply_all(R, x, &fi_ctx::q, false);
		for (fi_ctx* c : R) {
			hipLaunchKernelGGL((k_sub<T>), dim3(stream_blocks(c->g.nown)), dim3(kThreads), 0, c->stream, c->g.nown, vown<T>(c, b),
			                   vown<T>(c, &fi_ctx::q), vown<T>(c, &fi_ctx::mg_r));
		}
	}
	halo_exchange(R, &fi_ctx::mg_r);
	for (size_t i = 0; i < R.size(); ++i) {
		const LevelPair L = level_pair(R[i], Rc[i]);
		if (junction) { FI_HIP_TRY(hipMemsetAsync(Rc[i]->mg_b.p, 0, sizeof(T) * Rc[i]->g.nloc, R[i]->stream)); }
		launch_restrict<T>(L, vbase<T>(R[i], &fi_ctx::mg_r), vbase<T>(Rc[i], &fi_ctx::mg_b), R[i]->stream, vbase<T>(R[i], &fi_ctx::q),
		                   R[i]->g.n[2]);
	}
	if (junction) { sum_over_ranks<T>(R, Rc, &fi_ctx::mg_b); }
	vcycle<T>(Rc, &fi_ctx::mg_b, &fi_ctx::mg_x);
	halo_exchange(Rc, &fi_ctx::mg_x);
	for (size_t i = 0; i < R.size(); ++i) {
		const LevelPair L = level_pair(R[i], Rc[i]);
		launch_prolong<T>(L, vbase<T>(Rc[i], &fi_ctx::mg_x), vbase<T>(R[i], x), 1, R[i]->stream);
	}
	cheb_smooth<T>(R, b, x, deg, ratio, false);
}

// one dot-product reduction of the preconditioned recurrence: partials -> sum over blocks and ranks -> `phase`
template <typename CountFn>
void mg_reduce(RankSet& R, CountFn count_of, int phase)
{
	if (R.size() == 1 && R[0]->nranks == 1) {
		fi_ctx* c = R[0];
		hipLaunchKernelGGL(k_mg_logic, dim3(1), dim3(kThreads), 0, c->stream, c->scal.as<CgScalars>(), c->partial.as<double>(),
		                   count_of(c), phase);
		return;
	}
	reduce_phase(R, 1, count_of, count_of, -1);  // sums[0] on every member, summed over all of them
	for (fi_ctx* c : R) {
		hipLaunchKernelGGL(k_mg_logic, dim3(1), dim3(kThreads), 0, c->stream, c->scal.as<CgScalars>(),
		                   static_cast<const double*>(nullptr), 0, phase);
	}
}

// work vectors, cleared stop flags and smoother bounds for every level below (and including) R
template <typename T>
void mg_prepare(RankSet& R, bool clear_finest)
{
	RankSet lev = R;
	bool finest = true;
	while (!lev.empty() && lev[0]) {
		for (fi_ctx* l : lev) {
			mg_alloc<T>(l);
			if (!finest || clear_finest) {  // the operator kernels of a level exit early while ITS stop flag is up
				FI_HIP_TRY(hipMemsetAsync(l->scal.p, 0, sizeof(CgScalars), l->stream));  // (no host source that could go out of scope)
			}
		}
		finest = false;
		if (!lev[0]->coarse) { break; }
		lev = coarse_of(lev);
	}
	// smoother bounds.  Levels that smooth with the polynomial in A_model + f diag(A_data) (poly_smoother_ok): the bound of
	// the model operator (a number of the lattice and the weights: kept across assembles) and the scaling array.  The
	// others (Chebyshev in the full operator; the coarsest level always): power method on Dinv A, once per assemble.
	if (!R[0]->coarse) { return; }
	std::vector<RankSet> chain;
	for (RankSet l = R;; l = coarse_of(l)) {
		chain.push_back(l);
		if (!l[0]->coarse) { break; }
	}
	for (size_t k = chain.size(); k-- > 0;) {
		RankSet& l = chain[k];
		const bool coarsest = k + 1 == chain.size();
		bool poly_level = poly_smoother_ok<T>(l);
		if (poly_level) {
			for (fi_ctx* c : l) {
				if (!c->dinv16s_valid) { prepare_safe_scaling(c); }  // (also says whether the data pin a small level)
			}
			// the coarsest level: the polynomial only where the data hold every point at least as firmly as the model
			// couples it -- an SDF's coarsest level keeps weakly held global modes and needs the long Chebyshev polynomial in A
			if (coarsest) {
				for (const fi_ctx* c : l) { poly_level = poly_level && c->data_pinned && !test_switch("FI_MG_COARSEST_CHEB"); }
				if (!poly_level) {
					for (fi_ctx* c : l) { c->data_pinned = false; }
				}
			}
		}
		if (poly_level) {
			if (!(l[0]->poly_lambda > 0)) { for_each_copy(l, [&](RankSet& s) { estimate_poly_lambda<T>(s); }); }
		} else if (!(l[0]->lambda_max > 0)) {
			// Every level estimates its own bound.  (Rounds 1-2 let the finest level -- 8x the work -- take the estimate of
			// the level below it: the coarse replicas weigh data against model differently, and a stress case with
			// nearest-neighbour gradient rows, tests/stress_solve.py seed 5039, stagnated at 2e-3 for 60 000 iterations under
			// a smoother whose interval was too short.  Levels that smooth with the polynomial need no estimate at all.)
			for_each_copy(l, [&](RankSet& s) { estimate_lambda<T>(s); });
		}
	}
}

__global__ __launch_bounds__(kThreads) void k_to_twin(int64_t n, const CgScalars* __restrict__ sc, const double* __restrict__ r,
                                                       float* __restrict__ r32)
{
	const double inv = 1.0 / twin_scale(sc);
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		r32[i] = static_cast<float>(r[i] * inv);
	}
}

// z = V(r).  Mixed precision (Tw: the fp32 replicas): the V-cycle reads the replica's r and leaves z in the replica's
// mg_x, scaled by CgScalars::tscale -- `have_r32`: k_mg_step_mixed has already written the fp32 residual; the fp64 copy
// of z is not formed (k_dot_mixed / k_mg_direction_mixed read the fp32 one).
template <typename T>
void precondition(RankSet& R, RankSet& Tw, Vec r, Vec z, bool have_r32)
{
	vcycle<T>(R, r, z);
}
template <>
void precondition<double>(RankSet& R, RankSet& Tw, Vec r, Vec z, bool have_r32)
{
	if (Tw.empty()) {
		vcycle<double>(R, r, z);
		return;
	}
	if (!have_r32) {
		for (size_t i = 0; i < R.size(); ++i) {
			fi_ctx* c = R[i];
			fi_ctx* t = Tw[i];
			hipLaunchKernelGGL(k_to_twin, dim3(stream_blocks(c->g.nown)), dim3(kThreads), 0, c->stream, c->g.nown,
			                   c->scal.as<CgScalars>(), vown<double>(c, r), vown<float>(t, &fi_ctx::r));
		}
	}
	vcycle<float>(Tw, &fi_ctx::r, &fi_ctx::mg_x);
}

// coarse-to-fine start on the fp32 replicas of FI_F64 contexts (mixed precision), widened into x
__global__ __launch_bounds__(kThreads) void k_widen(int64_t n, const float* __restrict__ src, double* __restrict__ dst)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		dst[i] = static_cast<double>(src[i]);
	}
}
void twin_cascade_guess(RankSet& R)
{
	RankSet Tw;
	for (fi_ctx* c : R) {
		ensure_vectors(c->twin);
		c->twin->stats.coarse_iterations = 0;
		Tw.push_back(c->twin);
	}
	cascade_guess<float>(Tw);
	for (fi_ctx* c : R) {
		fi_ctx* t = c->twin;
		FI_HIP_TRY(hipMemsetAsync(c->x.p, 0, sizeof(double) * c->g.nloc, c->stream));
		hipLaunchKernelGGL(k_widen, dim3(stream_blocks(c->g.nown)), dim3(kThreads), 0, c->stream, c->g.nown,
		                   t->x.as<float>() + t->g.own_first, c->x.as<double>() + c->g.own_first);
		c->stats.coarse_iterations = t->stats.coarse_iterations;
	}
	FI_HIP_TRY(hipGetLastError());
}

// V-cycle preconditioned CG on the finest level; x of every member holds the guess on entry
template <typename T>
void cg_run_mg(RankSet& R, int max_iterations, float tol)
{
	fi_ctx* c0 = R[0];
	hipStream_t st = c0->stream;
	// mixed precision: the V-cycle runs on the fp32 replicas (RankSet Tw), CG itself stays in T = double
	RankSet Tw;
	if (sizeof(T) == 8 && c0->twin && c0->twin->coarse) {
		for (fi_ctx* c : R) { Tw.push_back(c->twin); }
		mg_prepare<float>(Tw, true);
	} else {
		mg_prepare<T>(R, false);
	}
	if (max_iterations <= 0) {
		const int64_t dflt = 2 * static_cast<int64_t>(c0->g.gn[0]) * c0->g.gn[1] * c0->g.gn[2];
		max_iterations = dflt > std::numeric_limits<int>::max() ? std::numeric_limits<int>::max() : static_cast<int>(dflt);
	}
	const double tolerance = tol > 0 ? static_cast<double>(tol) : static_cast<double>(std::numeric_limits<float>::epsilon());
	EventPair timer;  // (destroyed on every way out: a coarse level's breakdown, a timeout)
	const hipEvent_t e0 = timer.e0, e1 = timer.e1;
	FI_HIP_TRY(hipEventRecord(e0, st));
	CgScalars init{};
	init.tol2     = tolerance * tolerance;
	init.max_iter = max_iterations;
	reset_scalars(R, init);
	CgScalars* sc0 = c0->scal.as<CgScalars>();
	auto nbv      = [](fi_ctx* c) { return stream_blocks(c->g.nown); };
	auto nb_apply = [](fi_ctx* c) { return apply_num_partials(c); };
	const Vec X = &fi_ctx::x, Rv = &fi_ctx::r, P = &fi_ctx::p, Q = &fi_ctx::q, Z = &fi_ctx::mg_x, B = &fi_ctx::atb;
	while (static_cast<int>(c0->ev.size()) < 2 * kMaxSamples) {
		hipEvent_t e;
		FI_HIP_TRY(hipEventCreate(&e));
		c0->ev.push_back(e);
	}
	int samples = 0;
	const bool mixed = !Tw.empty();
	fi_ctx* const prec_ctx = mixed ? Tw[0] : c0;  // the context whose finest-level smoother chains are timed (vcycle)
	if (prec_ctx->level == 0 && !(mixed ? replicated_copies(Tw) : replicated_copies(R))) {
		while (static_cast<int>(prec_ctx->ev_prec.size()) < 2 * kPolySamples) {
			hipEvent_t e;
			FI_HIP_TRY(hipEventCreate(&e));
			prec_ctx->ev_prec.push_back(e);
		}
		prec_ctx->prec_budget = kPolySamples;
		prec_ctx->prec_taken  = 0;
	}
	auto dot = [&](Vec a, Vec b) {
		for (fi_ctx* c : R) {
			hipLaunchKernelGGL((k_dot<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, vown<T>(c, a), vown<T>(c, b),
			                   c->partial.as<double>());
		}
	};
	auto dot_rz = [&]() {  // partials of r . z
		if constexpr (std::is_same<T, double>::value) {
			if (mixed) {
				for (size_t i = 0; i < R.size(); ++i) {
					fi_ctx* c = R[i];
					hipLaunchKernelGGL(k_dot_mixed, dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, c->scal.as<CgScalars>(),
					                   vown<double>(c, Rv), vown<float>(Tw[i], &fi_ctx::mg_x), c->partial.as<double>());
				}
				return;
			}
		}
		dot(Rv, Z);
	};
	auto direction = [&](int first) {
		if constexpr (std::is_same<T, double>::value) {
			if (mixed) {
				for (size_t i = 0; i < R.size(); ++i) {
					fi_ctx* c = R[i];
					hipLaunchKernelGGL(k_mg_direction_mixed, dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown,
					                   c->scal.as<CgScalars>(), vown<float>(Tw[i], &fi_ctx::mg_x), vown<double>(c, P), first);
				}
				return;
			}
		}
		for (fi_ctx* c : R) {
			hipLaunchKernelGGL((k_mg_direction<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown,
			                   c->scal.as<CgScalars>(), vown<T>(c, Z), vown<T>(c, P), first);
		}
	};

	auto read_flag = [&]() {
		FI_HIP_TRY(hipMemcpyAsync(c0->scal_host, sc0, sizeof(CgScalars), hipMemcpyDeviceToHost, st));
		FI_HIP_TRY(hipStreamSynchronize(st));
		return c0->scal_host->done;
	};
	// r = b - A x, rr (b.b on the first call) and the stop test on it; unless that ends the solve: z = V(r), p = z, rz.
	// Returns the stop flag.  (The flag is read BEFORE the V-cycle is spent: a verification that confirms the recurrence's
	// residual -- the usual outcome -- costs one operator application, not a cycle.)
	auto restart = [&]() -> int {
		apply_all(R, X, Q, false);
		if (R.size() == 1 && c0->nranks == 1) {  // undivided lattice: one pass for r, r.r and b.b
			double* prr = c0->partial.as<double>();
			double* pbb = prr + static_cast<size_t>(c0->max_blocks);
			hipLaunchKernelGGL((k_resid_norms<T>), dim3(nbv(c0)), dim3(kThreads), 0, st, c0->g.nown, vown<T>(c0, B), vown<T>(c0, Q),
			                   vown<T>(c0, Rv), prr, pbb);
			hipLaunchKernelGGL(k_sum_to_slot2, dim3(1), dim3(kThreads), 0, st, sc0, pbb, nbv(c0));
			mg_reduce(R, nbv, kMgInitRr);
		} else {
		for (fi_ctx* c : R) {
			hipLaunchKernelGGL((k_sub<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, vown<T>(c, B), vown<T>(c, Q),
			                   vown<T>(c, Rv));
		}
		dot(B, B);
		reduce_phase(R, 1, nbv, nbv, -1);  // -> sums[0]
		for (fi_ctx* c : R) { hipLaunchKernelGGL(k_set_sum2, dim3(1), dim3(1), 0, c->stream, c->scal.as<CgScalars>()); }
		dot(Rv, Rv);
		mg_reduce(R, nbv, kMgInitRr);
		}
		const int flag = read_flag();
		if (flag) { return flag; }
		precondition<T>(R, Tw, Rv, Z, false);
		dot_rz();
		mg_reduce(R, nbv, kMgInitRz);
		direction(1);
		return 0;
	};
	int done = restart();

	double limit_s = 600.0;
	if (const char* env = getenv("FI_SOLVE_TIMEOUT_S")) { limit_s = atof(env); }
	const auto wall0 = std::chrono::steady_clock::now();
	bool timed_out = false;
	int restarts_left = c0->verify_residual ? 3 : 0;
	int widenings_left = 2;
	// One look at the stop flag per iteration, right behind the residual update: the V-cycle of an iteration that has just
	// converged is not launched.
	for (;;) {
		if (done == 2 && widenings_left > 0 && std::isfinite(c0->scal_host->rr) && std::isfinite(c0->scal_host->pq)) {
			// r.V(r) <= 0 with finite numbers: a smoother's interval is too short for its level (the bounds are power-method
			// estimates: lower bounds with 10 % headroom) and the V-cycle is not positive definite.  Widen every level's
			// interval and go on from the last iterate.
			--widenings_left;
			RankSet top = mixed ? Tw : R;
			for (fi_ctx* c : top) {
				for (fi_ctx* l = c; l; l = l->coarse) {
					l->tail_prog_valid = false;  // (the small-level engine's program carries the smoothers' constants)
					if (l->lambda_max > 0) { l->lambda_max *= 1.5; }
					if (l->poly_lambda > 0) {
						l->poly_lambda = (l->poly_lambda > 1.0 ? l->poly_lambda : 1.0) * 1.25;
						remember_lambda(l);
					}
				}
			}
			for (fi_ctx* c : R) { hipLaunchKernelGGL(k_set_done, dim3(1), dim3(1), 0, c->stream, c->scal.as<CgScalars>(), 0); }
			done = restart();
			continue;
		}
		if (done) {
			if (done != 1 || restarts_left <= 0) { break; }
			--restarts_left;  // recurrence converged: check b - A x, continue from it if it misses the tolerance
			for (fi_ctx* c : R) { hipLaunchKernelGGL(k_bump_restarts, dim3(1), dim3(1), 0, c->stream, c->scal.as<CgScalars>()); }
			done = restart();
			continue;
		}
		if (timed_out_anywhere(R, std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() > limit_s)) {
			timed_out = true;
			break;
		}
		const bool sample = samples < kMaxSamples;
		halo_exchange(R, P);
		if (sample) { FI_HIP_TRY(hipEventRecord(c0->ev[2 * samples], st)); }
		for (fi_ctx* c : R) { apply_AtA(c, c->p.p, c->q.p, c->partial.as<double>()); }
		if (sample) {
			FI_HIP_TRY(hipEventRecord(c0->ev[2 * samples + 1], st));
			++samples;
		}
		mg_reduce(R, nb_apply, kMgAlpha);
		bool stepped = false;
		if constexpr (std::is_same<T, double>::value) {
			if (mixed) {
				for (size_t i = 0; i < R.size(); ++i) {
					fi_ctx* c = R[i];
					hipLaunchKernelGGL(k_mg_step_mixed, dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, c->scal.as<CgScalars>(),
					                   vown<double>(c, P), vown<double>(c, Q), vown<double>(c, X), vown<double>(c, Rv),
					                   vown<float>(Tw[i], &fi_ctx::r), c->partial.as<double>());
				}
				stepped = true;
			}
		}
		if (!stepped) {
			for (fi_ctx* c : R) {
				hipLaunchKernelGGL((k_mg_step<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, c->scal.as<CgScalars>(),
				                   vown<T>(c, P), vown<T>(c, Q), vown<T>(c, X), vown<T>(c, Rv), c->partial.as<double>());
			}
		}
		mg_reduce(R, nbv, kMgResid);
		done = read_flag();
		if (done) { continue; }
		precondition<T>(R, Tw, Rv, Z, stepped);
		dot_rz();
		mg_reduce(R, nbv, kMgBeta);
		direction(0);
		FI_HIP_TRY(hipGetLastError());
	}
	FI_HIP_TRY(hipEventRecord(e1, st));
	FI_HIP_TRY(hipEventSynchronize(e1));
	float ms = 0;
	FI_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	const CgScalars h = *c0->scal_host;
	int used = samples < h.iter ? samples : h.iter;
	double sum_ms = 0;
	for (int k = 0; k < used; ++k) {
		float t = 0;
		FI_HIP_TRY(hipEventElapsedTime(&t, c0->ev[2 * k], c0->ev[2 * k + 1]));
		sum_ms += t;
	}
	// the timed smoother chains of cycles that ran (cycle k belongs to iteration k: the first one to the start)
	int pused = 0, plaunch = 0;
	double psum = 0;
	prec_ctx->prec_budget = 0;
	if (prec_ctx->level == 0) {
		const int ran = prec_ctx->prec_taken < h.iter + 1 ? prec_ctx->prec_taken : h.iter + 1;
		for (int k = 0; k < ran; ++k) {
			float t = 0;
			FI_HIP_TRY(hipEventElapsedTime(&t, prec_ctx->ev_prec[2 * k], prec_ctx->ev_prec[2 * k + 1]));
			psum += t;
			++pused;
		}
		plaunch = prec_ctx->prec_chain_launches;
	}
	for (fi_ctx* c : R) {
		c->stats.spmv_samples = used;
		c->stats.spmv_ms_avg  = used ? sum_ms / used : 0.0;
		c->stats.spmv_bytes   = apply_algorithmic_bytes(c);
		// per LAUNCH, like the polynomial PCG's figures: a sample holds the launches of one chain
		c->stats.prec_samples = pused * plaunch;
		c->stats.prec_ms_avg  = pused && plaunch ? psum / (pused * plaunch) : 0.0;
		c->stats.prec_bytes   = pused && plaunch ? prec_ctx->prec_chain_bytes / plaunch : 0.0;
		c->stats.operator_applies = h.iter + 1 + h.restarts;
		c->stats.solve_ms     = ms;
		c->stats.iterations   = h.iter;
		c->stats.converged    = (!timed_out && (h.done == 4 || h.done == 5 || (h.done == 1 && !c0->verify_residual))) ? 1 : 0;
		c->stats.rel_residual = h.bb > 0 ? std::sqrt(h.rr / h.bb) : 0.0;
		c->stats.restarts     = h.restarts;
		c->stats.verified_residual = (h.restarts > 0 && h.bb > 0) ? std::sqrt(h.true_rr / h.bb) : -1.0;
		if (h.done == 4) { FI_HIP_TRY(hipMemsetAsync(c->x.p, 0, sizeof(T) * c->g.nloc, c->stream)); }
	}
	FI_REQUIRE(h.done != 2, FI_ERR_BREAKDOWN, "CG breakdown: non-finite or non-positive curvature (p.AtA p = %g)", h.pq);
	FI_REQUIRE(!timed_out, FI_ERR_TIMEOUT, "solve stopped by the wall-clock guard (FI_SOLVE_TIMEOUT_S = %g s) after %d iterations, "
	           "relative residual %g", limit_s, h.iter, h.bb > 0 ? std::sqrt(h.rr / h.bb) : 0.0);
}


// the polynomial preconditioner runs through the 3-D marching kernel; contexts it does not cover (1-D / 2-D lattices,
// model_3 / model_4 / gradient_smoothness, rows given as triplets) keep the Jacobi diagonal
bool poly_ok(const fi_ctx* c) { return c->poly_terms > 1 && stencil_cheb_available(c) && c->generic.ntrip == 0 && !c->any_trip && c->tile_ts == 0; }

// ---- CG preconditioned by a Chebyshev polynomial in Dinv (A_model + diag(A_data)) ---------------------------------
// z = M r,  M = p_d(Dinv A~) Dinv  with  A~ = the model rows + the DIAGONAL of the data rows: symmetric positive definite
// whenever the interval's upper end bounds the spectrum of Dinv A~, which is at most max(lambda_max(m^-1 A_model), 1)
// (m = diag(A_model); the data diagonal only lowers the Rayleigh quotients) -- a number of the lattice and the model
// weights alone, found once by the power method (poly_lambda) and kept across assembles.  Iteration counts equal
// those of the polynomial in the full operator (profiles/r2_ablation.md), but a step of the polynomial never reads a
// cell record and is ONE launch of the plain marching kernel with the recurrence in its epilogue: 5 lattice passes.
// An outer iteration of d terms = 1 full apply (2 passes + records) + k_pcg_resid (5) + (d - 1) steps (4-5 each) +
// k_pcg_xp (5): 7 passes per operator application at d = 4 against the 12 of a Jacobi-PCG iteration, and TWO
// reductions (p.q; r.r with r.z) per outer iteration instead of two per application.
//
// The scalar recurrences are folded into the vector kernels like in cg_run: k_pcg_resid reads slot 0, sums the apply's
// p.q partials (or takes the all-reduced value from slot 2) and publishes alpha in slot 1; k_pcg_xp reads slot 1, sums
// r.r and r.z, publishes beta, the iteration count and the stop flag in slot 0.

// first half: alpha, r -= alpha q, z1 = Dinv r / theta, partials r.r and r.z1  (reads r, q, Dinv; writes r, z1)
// phase 0 / 2 (start / restart from b - A x): r = b - q instead, and b.b on the start
// streams of the polynomial-PCG vector kernels: plain or non-temporal 16-byte accesses (FI_PCG_NT; measured, see
// profiles/r2_ablation.md)
#ifndef FI_PCG_NT
#define FI_PCG_NT 1
#endif
template <typename T>
__device__ inline void pld16(T* dst, const T* base, int64_t i)
{
	if (FI_PCG_NT) { ld16_nt(dst, base, i); } else { *reinterpret_cast<typename Vec16<T>::V*>(dst) = reinterpret_cast<const typename Vec16<T>::V*>(base)[i]; }
}
template <typename T>
__device__ inline void pst16(T* base, int64_t i, const T* src)
{
	if (FI_PCG_NT) { st16_nt(base, i, src); } else { reinterpret_cast<typename Vec16<T>::V*>(base)[i] = *reinterpret_cast<const typename Vec16<T>::V*>(src); }
}

template <typename T, bool VEC>
__global__ __launch_bounds__(kThreads) void k_pcg_resid(int64_t n, const CgScalars* __restrict__ in, CgScalars* __restrict__ mid,
                                                         int tag, int phase, const double* __restrict__ pq_partial, int pq_count,
                                                         const T* __restrict__ b, const T* __restrict__ q,
                                                         const unsigned short* __restrict__ dinv, T* __restrict__ r, T* __restrict__ z1,
                                                         T inv_theta, double* __restrict__ prr, double* __restrict__ prz,
                                                         double* __restrict__ pbb)
{
	// one thread per workgroup reads the scalar record and the sum (see k_pcg_xp), the others take alpha from LDS
	__shared__ double sh_alpha;
	__shared__ int    sh_quit;
	double pq_all = 0.0;
	if (phase == 1 && pq_count > 1) { pq_all = sum_partials(pq_partial, pq_count); }  // folded form (undivided lattice)
	if (threadIdx.x == 0) {
		double a = 0.0;
		int    quit = 0;
		if (phase == 1) {
			CgScalars s = *in;
			if (s.done) {
				quit = 1;
			} else {
				const double pq  = pq_count > 1 ? pq_all : pq_partial[0];
				const bool   bad = !(pq > 0.0) || !isfinite(pq);
				a = s.rz / pq;
				if (blockIdx.x == 0) {
					s.pq    = pq;
					s.alpha = a;
					s.tag   = tag;
					if (bad) { s.done = 2; }
					*mid = s;
				}
				if (bad) { quit = 1; }
			}
		} else if (blockIdx.x == 0) {
			CgScalars s = *in;
			s.alpha = 0.0;
			s.tag   = tag;
			s.done  = 0;
			*mid = s;
		}
		sh_alpha = a;
		sh_quit  = quit;
	}
	__syncthreads();
	if (sh_quit) { return; }
	const double alpha_d = sh_alpha;
	using V = typename Vec16<T>::V;
	constexpr int N = VEC ? Vec16<T>::N : 1;
	const T alpha = static_cast<T>(alpha_d);
	double acc[3] = {0, 0, 0};
	const int64_t nv = n / N;
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < nv;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		T rv[N], qv[N], dv[N], zv[N], bv[N];
		typedef unsigned short D16 __attribute__((ext_vector_type(N)));  // the bfloat16 scaling the polynomial's steps use
		D16 d16 = D16{};
		const bool want_z = z1 != nullptr;  // null: the first step of the polynomial forms z1 itself while it loads r
		if (VEC) {
			pld16(qv, q, i);
			if (want_z) { d16 = reinterpret_cast<const D16*>(dinv)[i]; }
			if (phase == 1) { *reinterpret_cast<V*>(rv) = reinterpret_cast<const V*>(r)[i]; } else { pld16(bv, b, i); }
		} else {
			qv[0] = q[i];
			if (want_z) { d16[0] = dinv[i]; }
			if (phase == 1) { rv[0] = r[i]; } else { bv[0] = b[i]; }
		}
#pragma unroll
		for (int j = 0; j < N; ++j) { dv[j] = static_cast<T>(__uint_as_float(static_cast<unsigned int>(d16[j]) << 16)); }
		T s0 = T(0), s1 = T(0), s2 = T(0);
#pragma unroll
		for (int j = 0; j < N; ++j) {
			if (phase == 1) { rv[j] -= alpha * qv[j]; } else { rv[j] = bv[j] - qv[j]; s2 += bv[j] * bv[j]; }
			zv[j] = inv_theta * dv[j] * rv[j];
			s0 += rv[j] * rv[j];
			s1 += rv[j] * zv[j];
		}
		if (VEC) {
			reinterpret_cast<V*>(r)[i] = *reinterpret_cast<V*>(rv);
			if (want_z) { reinterpret_cast<V*>(z1)[i] = *reinterpret_cast<V*>(zv); }
		} else {
			r[i] = rv[0];
			if (want_z) { z1[i] = zv[0]; }
		}
		acc[0] += static_cast<double>(s0);
		acc[1] += static_cast<double>(s1);
		acc[2] += static_cast<double>(s2);
	}
	double out[3];
	block_sum<3>(acc, out);
	if (threadIdx.x == 0) {
		prr[blockIdx.x] = out[0];
		prz[blockIdx.x] = out[1];
		if (phase == 0) { pbb[blockIdx.x] = out[2]; }
	}
}

// second half: beta and the stop test, x += alpha p, p = z + beta p            (reads x, p, z; writes x, p)
// phase 0: start (b.b, tolerance, p = z); phase 2: restart from the true residual (p = z, verified stop)
template <typename T, bool VEC>
__global__ __launch_bounds__(kThreads) void k_pcg_xp(int64_t n, const CgScalars* __restrict__ mid, CgScalars* __restrict__ out_sc,
                                                      int tag, int phase, const double* __restrict__ prr, int nrr,
                                                      const double* __restrict__ prz, int nrz, const double* __restrict__ pbb,
                                                      int nbb, const T* __restrict__ z, T* __restrict__ x, T* __restrict__ p)
{
	// The scalar record and the sums sit in two or three cache lines that EVERY wave of the launch would read: ~20 k
	// requests to one or two L2 channels, which cost the launch 10-20 us (profiles/r2_ablation.md section 6).  One thread
	// per workgroup reads them and forms the new record; the others take it from LDS.
	__shared__ CgScalars sh;
	__shared__ int       sh_quit;
	double rr_all = 0.0, rz_all = 0.0, bb_all = 0.0;
	const bool lists = nrr > 1 || nrz > 1 || nbb > 1;  // folded form (undivided lattice): fixed-order sums by the whole workgroup
	if (lists) {
		rr_all = sum_partials(prr, nrr);
		rz_all = sum_partials(prz, nrz);
		if (phase == 0) { bb_all = sum_partials(pbb, nbb); }
	}
	if (threadIdx.x == 0) {
		CgScalars s = *mid;
		int quit = 0;
		if (s.tag != tag) {
			quit = 1;  // the first half of this iteration did not run: the solve had finished
		} else if (s.done == 2) {
			if (blockIdx.x == 0) { *out_sc = s; }
			quit = 1;
		} else {
			const double rr = lists ? rr_all : prr[0];
			const double rz = lists ? rz_all : prz[0];
			double beta_d = 0.0;
			if (phase == 1) {
				beta_d = rz / s.rz;
				s.iter += 1;
				// r.z <= 0 with a residual above the tolerance: the preconditioner is not positive definite (cg_run_poly widens
				// the polynomial's interval and goes on)
				s.done = !isfinite(rr) || !isfinite(rz) ? 2 : (!(rr > s.tol2) ? 1 : (!(rz > 0.0) ? 2 : (s.iter >= s.max_iter ? 3 : 0)));
			} else {
				if (phase == 0) {
					s.bb   = lists ? bb_all : pbb[0];
					s.tol2 = s.tol2 * s.bb;
					s.iter = 0;
				} else if (phase == 2) {
					s.restarts += 1;
					s.true_rr = rr;
				}
				s.done = !isfinite(rr) ? 2 : (s.bb == 0.0 ? 4 : (!(rr > s.tol2) ? (phase >= 2 ? 5 : 1) : (!(rz > 0.0) ? 2 : (s.iter >= s.max_iter ? 3 : 0))));
			}
			s.rz_new = rz;
			s.rr     = rr;
			s.beta   = beta_d;
			s.rz     = rz;
			if (blockIdx.x == 0) { *out_sc = s; }
			if (s.done == 2) { quit = 1; }
		}
		sh      = s;
		sh_quit = quit;
	}
	__syncthreads();
	if (sh_quit) { return; }
	using V = typename Vec16<T>::V;
	constexpr int N = VEC ? Vec16<T>::N : 1;
	const T    alpha = static_cast<T>(sh.alpha);
	const T    beta  = static_cast<T>(sh.beta);
	const bool go_on = sh.done == 0;
	const int64_t nv = n / N;
	// all three streams are loaded before anything is stored (a load placed behind the store of x costs a second
	// memory round trip per sweep: the kernel ran latency-bound, 69 instead of 53 us at 256^3)
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < nv;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		T xv[N], pv[N], zv[N];
		if (VEC) {
			pld16(zv, z, i);
			if (phase == 1) {
				pld16(xv, x, i);
				*reinterpret_cast<V*>(pv) = reinterpret_cast<const V*>(p)[i];
			}
		} else {
			zv[0] = z[i];
			if (phase == 1) { xv[0] = x[i]; pv[0] = p[i]; }
		}
		if (phase == 1) {
#pragma unroll
			for (int j = 0; j < N; ++j) {
				xv[j] += alpha * pv[j];
				pv[j] = zv[j] + beta * pv[j];
			}
			if (VEC) { pst16(x, i, xv); } else { x[i] = xv[0]; }
		} else {
#pragma unroll
			for (int j = 0; j < N; ++j) { pv[j] = zv[j]; }
		}
		if (go_on) {
			if (VEC) { reinterpret_cast<V*>(p)[i] = *reinterpret_cast<V*>(pv); } else { p[i] = pv[0]; }
		}
	}
}

// verified stop: r = b - A x has been formed by k_pcg_resid (phase 2); decide whether it meets the tolerance before
// the polynomial is spent on it (the usual outcome: it does, and the solve ends here)
__global__ __launch_bounds__(kThreads) void k_pcg_verify(CgScalars* __restrict__ mid, CgScalars* __restrict__ out_sc,
                                                          const double* __restrict__ prr, int nrr)
{
	const double rr = sum_partials(prr, nrr);
	if (threadIdx.x != 0) { return; }
	CgScalars s = *mid;
	s.restarts += 1;
	s.true_rr = rr;
	s.rr      = rr;
	s.done    = !isfinite(rr) ? 2 : (!(rr > s.tol2) ? 5 : (s.iter >= s.max_iter ? 3 : 0));
	*out_sc = s;
	s.done  = 0;
	*mid    = s;  // the restart that may follow (k_pcg_xp, phase 3) continues from this record
}

// sums of up to three partial lists of different lengths into sums[0..2] of a scalar slot (rank sets: the values
// then cross the slabs by k_group_sum / the all-reduce)
__global__ __launch_bounds__(kThreads) void k_reduce3(CgScalars* sc, const double* __restrict__ pa, int na,
                                                       const double* __restrict__ pb, int nb, const double* __restrict__ pc,
                                                       int nc)
{
	const double a = sum_partials(pa, na);
	const double b = pb ? sum_partials(pb, nb) : 0.0;
	const double c = pc ? sum_partials(pc, nc) : 0.0;
	if (threadIdx.x == 0) {
		sc->sums[0] = a;
		sc->sums[1] = b;
		sc->sums[2] = c;
	}
}

template <typename T>
void ensure_poly_vectors(fi_ctx* c)
{
	ensure_vectors(c);
	const size_t bytes = sizeof(T) * c->g.nloc;
	const bool fresh = c->mg_x.bytes < bytes || c->mg_d.bytes < bytes;
	c->mg_x.alloc(bytes);
	c->mg_d.alloc(bytes);
	if (fresh) {  // ghost planes outside the lattice are never written: keep them finite
		FI_HIP_TRY(hipMemsetAsync(c->mg_x.p, 0, bytes, c->stream));
		FI_HIP_TRY(hipMemsetAsync(c->mg_d.p, 0, bytes, c->stream));
	}
}

// largest eigenvalue of diag(A_model)^-1 A_model by the power method (16 steps through the marching kernel's epilogue,
// unnormalised: growth <= 4^16): a property of the lattice and the model weights, kept until fi_set_model
// ... and beyond the context: the estimate is a function of the lattice's extents, the model weights and the precision
// alone, so a process keeps the ones it has computed (a context that lives for one solve paid 16 launches and two host
// round trips per level for a number the context before it had already found: 0.8 ms of a 256^3 cold step).
struct LambdaKey {
	int   dtype, ndim, gn[3];
	float w[3];
	bool operator<(const LambdaKey& o) const { return std::memcmp(this, &o, sizeof(LambdaKey)) < 0; }
};
LambdaKey lambda_key(const fi_ctx* c)
{
	LambdaKey k;
	std::memset(&k, 0, sizeof(k));  // (padding bytes take part in the comparison)
	k.dtype = c->dtype;
	k.ndim  = c->g.ndim;
	for (int d = 0; d < 3; ++d) { k.gn[d] = c->g.gn[d]; }
	k.w[0] = c->w.model_0;
	k.w[1] = c->w.model_1;
	k.w[2] = c->w.model_2;
	return k;
}
std::mutex g_lambda_mutex;
std::map<LambdaKey, double> g_lambda_cache;
// a solve that found the bound too small has widened it (done == 2): the process-wide entry follows, so that the next
// context of this lattice and model does not repeat the breakdown and the restart
void remember_lambda(const fi_ctx* c)
{
	if (test_switch("FI_NO_LAMBDA_CACHE") || test_switch("FI_POLY_LAMBDA_SCALE") || !(c->poly_lambda > 0)) { return; }
	std::lock_guard<std::mutex> lock(g_lambda_mutex);
	double& v = g_lambda_cache[lambda_key(c)];
	if (c->poly_lambda > v) { v = c->poly_lambda; }
}

template <typename T>
void estimate_poly_lambda(RankSet& R)
{
	if (!test_switch("FI_NO_LAMBDA_CACHE")) {
		std::lock_guard<std::mutex> lock(g_lambda_mutex);
		auto it = g_lambda_cache.find(lambda_key(R[0]));
		if (it != g_lambda_cache.end()) {
			for (fi_ctx* c : R) { c->poly_lambda = it->second; }
			return;
		}
	}
	for (fi_ctx* c : R) { ensure_poly_vectors<T>(c); }
	auto nbv = [](fi_ctx* c) { return stream_blocks(c->g.nown); };
	for (fi_ctx* c : R) {
		FI_HIP_TRY(hipMemsetAsync(c->scal.p, 0, sizeof(CgScalars), c->stream));  // (no host source that could go out of scope)
		hipLaunchKernelGGL((k_seed<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, global_first(c),
		                   c->mg_x.as<T>() + c->g.own_first);
	}
	const int steps = 16;
	double sums[2] = {0, 0};
	DevBuf fi_ctx::*cur = &fi_ctx::mg_x, fi_ctx::*nxt = &fi_ctx::mg_d;
	for (int k = 0; k < steps; ++k) {
		halo_exchange(R, cur);
		for (fi_ctx* c : R) { stencil_power_step(c, (c->*cur).p, (c->*nxt).p, c->partial.as<double>()); }
		if (k >= steps - 2) {
			for (fi_ctx* c : R) {
				hipLaunchKernelGGL(k_reduce3, dim3(1), dim3(kThreads), 0, c->stream, c->scal.as<CgScalars>() + 2, c->partial.as<double>(),
				                   stencil_cheb_partials(c), static_cast<const double*>(nullptr), 0, static_cast<const double*>(nullptr), 0);
			}
			if (R.size() > 1) {
				hipLaunchKernelGGL(k_group_sum, dim3(1), dim3(1), 0, R[0]->stream, R[0]->group_scal.as<CgScalars*>(),
				                   static_cast<int>(R.size()), 1, 2);
			} else if (R[0]->nranks > 1) {
				allreduce_sum(R[0], (R[0]->scal.as<CgScalars>() + 2)->sums, 1);
			}
			fi_ctx* c0 = R[0];
			FI_HIP_TRY(hipMemcpyAsync(c0->scal_host, c0->scal.as<CgScalars>() + 2, sizeof(CgScalars), hipMemcpyDeviceToHost, c0->stream));
			FI_HIP_TRY(hipStreamSynchronize(c0->stream));
			sums[k - (steps - 2)] = c0->scal_host->sums[0];
		}
		std::swap(cur, nxt);
	}
	const double lambda = (sums[0] > 0 && sums[1] > 0 && std::isfinite(sums[1])) ? std::sqrt(sums[1] / sums[0]) : 4.0;
	for (fi_ctx* c : R) { c->poly_lambda = lambda; }
	if (sums[0] > 0 && sums[1] > 0 && std::isfinite(sums[1])) {
		std::lock_guard<std::mutex> lock(g_lambda_mutex);
		if (g_lambda_cache.size() > 4096) { g_lambda_cache.clear(); }
		g_lambda_cache[lambda_key(R[0])] = lambda;
	}
}

template <typename T>
void cg_run_poly(RankSet& R, int max_iterations, float tol)
{
	fi_ctx* c0 = R[0];
	hipStream_t st = c0->stream;
	const int terms = c0->poly_terms;
	if (max_iterations <= 0) {
		const int64_t dflt = 2 * static_cast<int64_t>(c0->g.gn[0]) * c0->g.gn[1] * c0->g.gn[2];
		max_iterations = dflt > std::numeric_limits<int>::max() ? std::numeric_limits<int>::max() : static_cast<int>(dflt);
	}
	const double tolerance = tol > 0 ? static_cast<double>(tol) : static_cast<double>(std::numeric_limits<float>::epsilon());
	for (fi_ctx* c : R) { ensure_poly_vectors<T>(c); }
	if (!(c0->poly_lambda > 0)) { estimate_poly_lambda<T>(R); }

	EventPair timer;  // (destroyed on every way out: a coarse level's breakdown, a timeout)
	const hipEvent_t e0 = timer.e0, e1 = timer.e1;
	FI_HIP_TRY(hipEventRecord(e0, st));

	// Chebyshev interval and recurrence constants (the same polynomial as cheb_smooth).  The preconditioner is positive
	// definite while the spectrum of Dinv A~ stays below hi + lo; poly_lambda is a power-method estimate -- a LOWER bound of
	// the largest eigenvalue -- with 10 % headroom, so a lattice it underestimates by more shows up as non-positive
	// curvature (done == 2).  The solve then widens the interval (x 1.25, twice) and goes on from its last iterate, and
	// after that falls back to the Jacobi diagonal (cg_run): see the end of the loop.  FI_POLY_LAMBDA_SCALE (tests):
	// scales the estimate, to drive that path.
	double lam_scale = 1.0;
	if (const char* env = test_switch("FI_POLY_LAMBDA_SCALE")) { lam_scale = atof(env) > 0 ? atof(env) : 1.0; }
	double theta = 1.0, delta = 1.0;
	std::vector<double> c1s, c2s;
	auto set_interval = [&]() {
		const double lam = (c0->poly_lambda > 1.0 ? c0->poly_lambda : 1.0) * lam_scale;
		const double hi = 1.1 * lam, lo = hi / (c0->poly_ratio > 1.0 ? c0->poly_ratio : 10.0);
		theta = 0.5 * (hi + lo);
		delta = 0.5 * (hi - lo);
		const double sigma = theta / delta;
		c1s.clear();
		c2s.clear();
		double rho = 1.0 / sigma;
		for (int k = 1; k < terms; ++k) {
			const double rho_new = 1.0 / (2.0 * sigma - rho);
			c1s.push_back(rho_new * rho);
			c2s.push_back(2.0 * rho_new / delta);
			rho = rho_new;
		}
	};
	set_interval();

	CgScalars init{};
	init.tol2     = tolerance * tolerance;
	init.max_iter = max_iterations;
	init.rz       = 1.0;
	reset_scalars(R, init);

	auto nbf_of   = [](fi_ctx* c) { const int b = stream_blocks(c->g.nown); return b > 1024 ? 1024 : b; };
	auto vec_ok = [](fi_ctx* c) {
		constexpr int N = Vec16<T>::N;
		return (c->g.own_first % N == 0) && (c->g.nown % N == 0);
	};
	const bool single = R.size() == 1 && c0->nranks == 1;
	// 3 terms or more: the first step of the polynomial reads r and the scaling and forms z_0 as it loads them
	// (fi_stencil.hip, PRO) -- k_pcg_resid then stores no z_0, and step 2 recomputes it as its z_prev.  One slab per
	// process: the ghost planes of r are exchanged instead of z_0's, those of the scaling came with the assembly
	// (operator_prepare); the loop-back group (no transport at assembly time) keeps the stored z_0.
	bool ghosts_scaled = true;  // slabs: the scaling on the ghost planes is the neighbour's (exchanged with the assembly)
	for (const fi_ctx* c : R) { ghosts_scaled = ghosts_scaled && c->scaling_ghosts; }
	const bool z0_on_load = (single || ghosts_scaled) && terms > 2 && c0->march.valid &&
	                        !test_switch("FI_NO_Z0_ON_LOAD");  // (3-D: the marching kernel; 2-D lattices store z_0)
	// Deep exchange (slabs; fi_assemble has given the vectors 2 (d - 1) ghost planes): the ghost planes of r travel ONCE per
	// polynomial; step k then also computes its 2 (d - 1 - k) nearest ghost planes -- the values the neighbour computes
	// for its own planes, bit for bit -- so that no step waits for an exchange: 2 exchanges per outer iteration (p for the
	// apply, r for the polynomial) instead of d.  FI_NO_DEEP_HALO: one exchange per step (tests: identical results).
	const int  deep_width = 2 * (terms - 1);
	const bool deep = z0_on_load && c0->nranks > 1 && c0->halo >= deep_width && c0->min_slab >= deep_width &&
	                  !test_switch("FI_NO_DEEP_HALO");
	// Undivided lattice: the sums of the per-workgroup partials (p.q; r.r, r.z) are folded into their consumers (every
	// workgroup sums the 1-4 k partials in the same fixed order); rank sets form them once, by a one-block kernel in
	// front of the all-reduce.  Measured at 256^3 with both forms (profiles/r2_ablation.md section 6): folded 10.65 ms
	// per bench step, one-block kernels 10.82 (two 4.6 us launches per outer iteration).
	const bool folded = single && !tuning_switch("FI_POLY_UNFOLDED");
	// partial regions of every member: [0] apply p.q, [1] r.r, [2] r.z, [3] b.b
	auto region = [](fi_ctx* c, int k) { return c->partial.as<double>() + static_cast<size_t>(k) * c->max_blocks; };
	auto slot2 = [](fi_ctx* c) { return (c->scal.as<CgScalars>() + 2)->sums; };
	int n_exchanges = 0, n_reductions = 0;  // (statistics: what an outer iteration costs over slabs)
	auto cross = [&](int nvec) {  // sums[0..nvec) of slot 2 over the slabs
		if (R.size() > 1 || c0->nranks > 1) { ++n_reductions; }
		if (R.size() > 1) {
			hipLaunchKernelGGL(k_group_sum, dim3(1), dim3(1), 0, st, c0->group_scal.as<CgScalars*>(), static_cast<int>(R.size()), nvec, 2);
		} else if (c0->nranks > 1) {
			allreduce_sum(c0, slot2(c0), nvec);
		}
	};
	const Vec ZA = &fi_ctx::mg_x, ZB = &fi_ctx::mg_d;

	int tag = 0;
	int psamples = 0;  // timed Chebyshev steps
	std::vector<int> ptags;
	std::vector<double> pbytes;  // algorithmic bytes of the sampled steps
	std::vector<hipEvent_t>& pev = c0->ev_prec;
	while (static_cast<int>(pev.size()) < 2 * kMaxSamples) {
		hipEvent_t e;
		FI_HIP_TRY(hipEventCreate(&e));
		pev.push_back(e);
	}
	// one pass of the recurrence: phase 1 = a CG step (the apply of p has been launched), 0 / 2 = start / restart (the
	// apply of x has been launched)
	auto first_half = [&](int phase) {  // alpha, r, z1 (phase 0 / 2 / 3: r = b - A x)
		++tag;
		if (!folded && phase == 1) {
			for (fi_ctx* c : R) {
				hipLaunchKernelGGL(k_reduce3, dim3(1), dim3(kThreads), 0, c->stream, c->scal.as<CgScalars>() + 2, region(c, 0),
				                   apply_num_partials(c), static_cast<const double*>(nullptr), 0, static_cast<const double*>(nullptr), 0);
			}
			cross(1);
		}
		for (fi_ctx* c : R) {
			const int64_t o = c->g.own_first;
			const int     nbf = nbf_of(c);
			CgScalars*    sc = c->scal.as<CgScalars>();
			const double* pq = folded ? region(c, 0) : slot2(c);
			const int     npq = folded ? apply_num_partials(c) : 1;
			auto go = [&](auto kernel) {
				hipLaunchKernelGGL(kernel, dim3(nbf), dim3(kThreads), 0, c->stream, c->g.nown, sc, sc + 1, tag, phase, pq, npq,
				                   c->atb.as<T>() + o, c->q.as<T>() + o, c->dinv16.as<unsigned short>() + o, c->r.as<T>() + o,
				                   z0_on_load ? static_cast<T*>(nullptr) : vown<T>(c, ZA),
				                   static_cast<T>(1.0 / theta), region(c, 1), region(c, 2), region(c, 3));
			};
			if (vec_ok(c)) { go(k_pcg_resid<T, true>); } else { go(k_pcg_resid<T, false>); }
		}
	};
	auto second_half = [&](int phase) {  // the polynomial, beta and the stop test, x and p
		// the polynomial: z_{k+1} from z_k (ZA / ZB alternate; the result ends in `zfin`)
		Vec zin = ZA, zout = ZB;
		for (int k = 1; k < terms; ++k) {
			// every 4th pass is timed: ALL its steps between one pair of event records (first step: 2.5 lattice passes,
			// second: 3.5, the others 4.5) -- a pair of records idles the stream for a few microseconds, which a single
			// 45 us launch between them shows as 5 % (0.58 against the trace's 0.61), three launches as 2 %.  The roofline
			// figure is bytes over time of all sampled launches; the statistics report the mean per launch.
			const bool sample_pass = phase == 1 && c0->level == 0 && psamples < kPolySamples && (tag & 3) == 3 &&
			                         !tuning_switch("FI_NO_SAMPLES");
			const bool sample = sample_pass && k == 1, sample_end = sample_pass && k == terms - 1;
			const bool overlap = !deep && R.size() == 1 && overlap_possible(c0) && c0->march.np_inner > 0;
			const bool pro = z0_on_load && k == 1;
			const Vec  zsrc = pro ? static_cast<Vec>(&fi_ctx::r) : zin;  // the vector whose ghost planes the step reads
			const int  ext = deep ? 2 * (terms - 1 - k) : 0;            // ghost planes this step computes for the next one
			if (deep) {
				if (k == 1) {
					halo_exchange(R, &fi_ctx::r, deep_width);
					++n_exchanges;
				}
			} else {
				if (overlap) { exchange_begin(c0, (c0->*zsrc).p); } else { halo_exchange(R, zsrc); }
				if (c0->nranks > 1) { ++n_exchanges; }
			}
			if (sample) {
				FI_HIP_TRY(hipEventRecord(pev[2 * psamples], st));
				ptags.push_back(tag);
				// z, z_prev, r in, z_new out + the bfloat16 scaling; the first step has no z_prev and (formed on load) reads r as
				// its z; the second step's z_prev is recomputed from r
				double bytes = 0;
				for (int j = 1; j < terms; ++j) {
					const double vecs = j == 1 ? (z0_on_load ? 2.0 : 3.0) : (j == 2 ? 3.0 : 4.0);
					bytes += (static_cast<double>(sizeof(T)) * vecs + 2.0) * static_cast<double>(c0->g.nown);
				}
				pbytes.push_back(bytes);
			}
			for (fi_ctx* c : R) {
				const void* zp = k == 1 ? nullptr : (c->*zout).p;
				// the second step's z_prev is z_0 = Dinv r / theta: recomputed from r and Dinv, which the step reads anyway
				const double zs = k == 2 ? 1.0 / theta : 0.0;
				if (pro) {
					if (overlap) {
						stencil_cheb_step(c, c->r.p, nullptr, c->r.p, (c->*zout).p, c1s[0], c2s[0], region(c, 2), 1, 0.0, 1.0 / theta);
						exchange_wait(c);
						stencil_cheb_step(c, c->r.p, nullptr, c->r.p, (c->*zout).p, c1s[0], c2s[0], region(c, 2), 2, 0.0, 1.0 / theta);
					} else {
						stencil_cheb_step(c, c->r.p, nullptr, c->r.p, (c->*zout).p, c1s[0], c2s[0], region(c, 2), 0, 0.0, 1.0 / theta, nullptr, ext);
					}
					continue;
				}
				if (overlap) {  // the workgroups that read no ghost plane, then the first and last z-chunk
					stencil_cheb_step(c, (c->*zin).p, zp, c->r.p, (c->*zout).p, c1s[k - 1], c2s[k - 1], region(c, 2), 1, zs);
					exchange_wait(c);
					stencil_cheb_step(c, (c->*zin).p, zp, c->r.p, (c->*zout).p, c1s[k - 1], c2s[k - 1], region(c, 2), 2, zs);
				} else {
					stencil_cheb_step(c, (c->*zin).p, zp, c->r.p, (c->*zout).p, c1s[k - 1], c2s[k - 1], region(c, 2), 0, zs, 0.0, nullptr, ext);
				}
			}
			if (sample_end) {
				FI_HIP_TRY(hipEventRecord(pev[2 * psamples + 1], st));
				++psamples;
			}
			std::swap(zin, zout);
		}
		const Vec zfin = zin;
		if (!folded) {
			for (fi_ctx* c : R) {
				hipLaunchKernelGGL(k_reduce3, dim3(1), dim3(kThreads), 0, c->stream, c->scal.as<CgScalars>() + 2, region(c, 1), nbf_of(c),
				                   region(c, 2), terms > 1 ? stencil_cheb_partials(c) : nbf_of(c),
				                   phase == 0 ? region(c, 3) : static_cast<const double*>(nullptr), nbf_of(c));
			}
			cross(phase == 0 ? 3 : 2);
		}
		for (fi_ctx* c : R) {
			const int64_t o = c->g.own_first;
			const int     nbf = nbf_of(c);
			CgScalars*    sc = c->scal.as<CgScalars>();
			const int     nrz = terms > 1 ? stencil_cheb_partials(c) : nbf;
			auto go = [&](auto kernel) {
				if (folded) {
					hipLaunchKernelGGL(kernel, dim3(nbf), dim3(kThreads), 0, c->stream, c->g.nown, sc + 1, sc, tag, phase, region(c, 1), nbf,
					                   region(c, 2), nrz, region(c, 3), nbf, vown<T>(c, zfin), c->x.as<T>() + o, c->p.as<T>() + o);
				} else {
					hipLaunchKernelGGL(kernel, dim3(nbf), dim3(kThreads), 0, c->stream, c->g.nown, sc + 1, sc, tag, phase, slot2(c), 1,
					                   slot2(c) + 1, 1, slot2(c) + 2, 1, vown<T>(c, zfin), c->x.as<T>() + o, c->p.as<T>() + o);
				}
			};
			if (vec_ok(c)) { go(k_pcg_xp<T, true>); } else { go(k_pcg_xp<T, false>); }
		}
	};
	auto half_steps = [&](int phase) {
		first_half(phase);
		second_half(phase);
	};
	auto start = [&](int phase) {  // r = b - A x, z = M r, p = z
		apply_exchanged(R, &fi_ctx::x, &fi_ctx::q, nullptr);
		half_steps(phase);
	};
	start(0);

	int samples = 0;
	while (static_cast<int>(c0->ev.size()) < 2 * kMaxSamples) {
		hipEvent_t e;
		FI_HIP_TRY(hipEventCreate(&e));
		c0->ev.push_back(e);
	}
	double limit_s = 600.0;
	if (const char* env = getenv("FI_SOLVE_TIMEOUT_S")) { limit_s = atof(env); }
	const auto wall0 = std::chrono::steady_clock::now();
	bool timed_out = false;
	CgScalars* sc0 = c0->scal.as<CgScalars>();
	int restarts_left = c0->verify_residual ? 3 : 0;
	int widenings_left = 2, iter_base = 0;
	const int burst = terms >= 4 ? 4 : 8;  // outer iterations between two looks at the stop flag
	// the first look comes when the context's previous solve had finished (the per-frame / re-assembled problem of a
	// caller changes little): every look is a host round trip of ~35 us
	int next_burst = c0->last_outer_iterations > 0 ? (c0->last_outer_iterations < 64 ? c0->last_outer_iterations : 64) : burst;
	int issued = 0;
	for (;;) {
		FI_HIP_TRY(hipMemcpyAsync(c0->scal_host, sc0, sizeof(CgScalars), hipMemcpyDeviceToHost, st));
		FI_HIP_TRY(hipStreamSynchronize(st));
		if (c0->scal_host->done == 2 && widenings_left > 0 && std::isfinite(c0->scal_host->pq) && std::isfinite(c0->scal_host->rr) &&
		    std::isfinite(c0->scal_host->rz)) {
			// non-positive curvature with finite numbers: the polynomial's interval is too narrow for this lattice (see
			// set_interval).  Widen it -- for the context's later solves too -- and go on from the last iterate: x has
			// not been touched by the step that broke down.
			--widenings_left;
			iter_base += c0->scal_host->iter;
			for (fi_ctx* c : R) { c->poly_lambda = (c->poly_lambda > 1.0 ? c->poly_lambda : 1.0) * 1.25; }
			remember_lambda(c0);  // (the next context of this lattice and model starts from the widened bound)
			set_interval();
			init.max_iter = max_iterations > iter_base ? max_iterations - iter_base : 1;
			reset_scalars(R, init);
			FI_HIP_TRY(hipStreamSynchronize(st));  // (reset_scalars copies from `init`)
			start(0);
			continue;
		}
		if (c0->scal_host->done) {
			if (c0->scal_host->done != 1 || restarts_left <= 0) { break; }
			--restarts_left;  // the recurrence met the tolerance: check b - A x, go on from it if it misses
			for (fi_ctx* c : R) { hipLaunchKernelGGL(k_set_done, dim3(1), dim3(1), 0, c->stream, c->scal.as<CgScalars>(), 0); }
			apply_exchanged(R, &fi_ctx::x, &fi_ctx::q, nullptr);
			first_half(2);  // r = b - A x, z1, partials of r.r
			if (!folded) {
				for (fi_ctx* c : R) {
					hipLaunchKernelGGL(k_reduce3, dim3(1), dim3(kThreads), 0, c->stream, c->scal.as<CgScalars>() + 2, region(c, 1), nbf_of(c),
					                   static_cast<const double*>(nullptr), 0, static_cast<const double*>(nullptr), 0);
				}
				cross(1);
			}
			for (fi_ctx* c : R) {
				CgScalars* sc = c->scal.as<CgScalars>();
				hipLaunchKernelGGL(k_pcg_verify, dim3(1), dim3(kThreads), 0, c->stream, sc + 1, sc, folded ? region(c, 1) : slot2(c),
				                   folded ? nbf_of(c) : 1);
			}
			FI_HIP_TRY(hipMemcpyAsync(c0->scal_host, sc0, sizeof(CgScalars), hipMemcpyDeviceToHost, st));
			FI_HIP_TRY(hipStreamSynchronize(st));
			if (c0->scal_host->done) { break; }   // verified (5), out of iterations (3) or not finite (2)
			second_half(3);  // the true residual misses the tolerance: CG goes on from it (z = M r, p = z)
			continue;
		}
		if (timed_out_anywhere(R, std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() > limit_s)) {
			timed_out = true;
			break;
		}
		const int nb = next_burst;
		next_burst = burst > 2 ? 2 : burst;  // after the first look the solve is close to its end
		for (int k = 0; k < nb; ++k) {
			++issued;
			// timed launches: a pair of event records costs the stream ~11 us of idle time (two 5.6 us gaps around the
			// launch), so only the finest level is sampled, three applies and three Chebyshev steps per solve
			const bool sample = c0->level == 0 && samples < kPolySamples && (issued & 3) == 1 && !tuning_switch("FI_NO_SAMPLES");
			if (sample) { FI_HIP_TRY(hipEventRecord(c0->ev[2 * samples], st)); }
			apply_exchanged(R, &fi_ctx::p, &fi_ctx::q, +[](fi_ctx* c) -> double* { return c->partial.as<double>(); });
			if (sample) {
				FI_HIP_TRY(hipEventRecord(c0->ev[2 * samples + 1], st));
				++samples;
			}
			half_steps(1);
		}
		FI_HIP_TRY(hipGetLastError());
	}
	FI_HIP_TRY(hipEventRecord(e1, st));
	FI_HIP_TRY(hipEventSynchronize(e1));
	float ms = 0;
	FI_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));

	const CgScalars h = *c0->scal_host;
	int used = samples;
	if ((h.iter + 3) / 4 < used) { used = (h.iter + 3) / 4; }  // sample k belongs to outer iteration 4k + 1
	double sum_ms = 0;
	for (int k = 0; k < used; ++k) {
		float t = 0;
		FI_HIP_TRY(hipEventElapsedTime(&t, c0->ev[2 * k], c0->ev[2 * k + 1]));
		sum_ms += t;
	}
	// the timed steps of passes that ran (pass t is outer iteration t - 1; passes past the stop exited at once)
	int pused = 0;
	double psum = 0, pbsum = 0;
	for (int k = 0; k < psamples; ++k) {
		if (ptags[k] - 1 > h.iter) { break; }  // pass `tag` is CG step tag - 1 (or earlier, after restarts): it ran
		float t = 0;
		FI_HIP_TRY(hipEventElapsedTime(&t, pev[2 * k], pev[2 * k + 1]));
		psum += t;
		pbsum += pbytes[k];
		++pused;
	}
	for (fi_ctx* c : R) {
		c->stats.spmv_samples = used;
		c->stats.spmv_ms_avg  = used ? sum_ms / used : 0.0;
		c->stats.spmv_bytes   = apply_algorithmic_bytes(c);
		// per LAUNCH: a sample holds the terms - 1 steps of one polynomial; bytes / time is their byte-weighted rate
		c->stats.prec_samples = pused * (terms - 1);
		c->stats.prec_ms_avg  = pused ? psum / (pused * (terms - 1)) : 0.0;
		c->stats.prec_bytes   = pused ? pbsum / (pused * (terms - 1)) : 0.0;
		c->stats.operator_applies = (iter_base + h.iter + 1) * terms + h.restarts;
		// (+ one exchange of p per full apply: every outer iteration, the start and each verification)
		c->stats.halo_exchanges = c0->nranks > 1 ? n_exchanges + issued + 1 + h.restarts : 0;
		c->stats.reductions     = n_reductions;
		c->last_outer_iterations = h.iter;
		c->stats.solve_ms     = ms;
		c->stats.iterations   = iter_base + h.iter;
		c->stats.converged    = (!timed_out && (h.done == 4 || h.done == 5 || (h.done == 1 && !c0->verify_residual))) ? 1 : 0;
		c->stats.rel_residual = h.bb > 0 ? std::sqrt(h.rr / h.bb) : 0.0;
		c->stats.restarts     = h.restarts;
		c->stats.verified_residual = (h.restarts > 0 && h.bb > 0) ? std::sqrt(h.true_rr / h.bb) : -1.0;
		if (h.done == 4) { FI_HIP_TRY(hipMemsetAsync(c->x.p, 0, sizeof(T) * c->g.nloc, c->stream)); }
	}
	FI_REQUIRE(h.done != 2, FI_ERR_BREAKDOWN, "CG breakdown: non-finite or non-positive curvature (p.AtA p = %g)", h.pq);
	FI_REQUIRE(!timed_out, FI_ERR_TIMEOUT, "solve stopped by the wall-clock guard (FI_SOLVE_TIMEOUT_S = %g s) after %d iterations, "
	           "relative residual %g", limit_s, h.iter, h.bb > 0 ? std::sqrt(h.rr / h.bb) : 0.0);
}

// ---- the same solve over slabs with ONE reduction per outer iteration ---------------------------------------------------
// Chronopoulos-Gear form of preconditioned CG: with z = M r and w = A z
//     gamma = r.z, delta = z.w (and r.r for the stop test) -- ONE all-reduce of three numbers --
//     beta = gamma / gamma_old,  alpha = gamma / (delta - beta gamma / alpha_old),
//     p = z + beta p,  s = w + beta s  (s = A p by recurrence),  x += alpha p,  r -= alpha s.
// Against cg_run_poly an outer iteration trades its second all-reduce for a fourth vector recurrence (22.5 instead of 20.5
// lattice passes): over slabs of a strong split, where an iteration is a few dozen microseconds of kernels between
// latency-bound collectives, that is the better trade -- 2 exchanges (p's role is taken by z; r's deep exchange) + 1
// all-reduce per outer iteration.  r.z comes out of the polynomial's last step, z.w out of the apply, r.r out of the
// previous update: no extra pass for the dot products.  Same stop rule (on the residual the iteration STARTS from: x and r
// are left consistent), same verified stop; a breakdown hands over to cg_run_poly (which widens the polynomial's
// interval) from the current iterate.  Undivided lattices keep the two-reduction form (folded sums, fewer passes).

// r = b - q, partials r.r and b.b (start / verification)
template <typename T>
__global__ __launch_bounds__(kThreads) void k_sr_residual(int64_t n, const T* __restrict__ b, const T* __restrict__ q, T* __restrict__ r,
                                                           double* __restrict__ prr, double* __restrict__ pbb)
{
	double acc[2] = {0, 0};
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		const T bv = b[i], rv = bv - q[i];
		r[i] = rv;
		acc[0] += static_cast<double>(rv) * static_cast<double>(rv);
		acc[1] += static_cast<double>(bv) * static_cast<double>(bv);
	}
	double out[2];
	block_sum<2>(acc, out);
	if (threadIdx.x == 0) {
		prr[blockIdx.x] = out[0];
		pbb[blockIdx.x] = out[1];
	}
}

// the scalar record after the start (phase 0: b.b, the tolerance) or a verification (phase 2): sums = {r.r, b.b}
__global__ void k_sr_setup(CgScalars* state, const CgScalars* sums_slot, int phase)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) { return; }
	CgScalars s = *state;
	const double rr = sums_slot->sums[0];
	if (phase == 0) {
		s.bb   = sums_slot->sums[1];
		s.tol2 = s.tol2 * s.bb;
		s.iter = 0;
	} else {
		s.restarts += 1;
		s.true_rr = rr;
	}
	s.rr   = rr;
	s.done = !isfinite(rr) ? 2 : (s.bb == 0.0 ? 4 : (!(rr > s.tol2) ? (phase == 2 ? 5 : 1) : (s.iter >= s.max_iter ? 3 : 0)));
	*state = s;
}

// scalars of one step from the all-reduced sums {r.z, z.w, r.r}, then the four recurrences; partials of the new r.r
template <typename T>
__global__ __launch_bounds__(kThreads) void k_sr_update(int64_t n, const CgScalars* __restrict__ in, CgScalars* __restrict__ out,
                                                         const CgScalars* __restrict__ sums_slot, int first,
                                                         const T* __restrict__ z, const T* __restrict__ w, T* __restrict__ p,
                                                         T* __restrict__ s, T* __restrict__ x, T* __restrict__ r,
                                                         double* __restrict__ prr)
{
	__shared__ double sh_alpha, sh_beta;
	__shared__ int    sh_quit;
	if (threadIdx.x == 0) {
		CgScalars st = *in;
		int quit = 0;
		double alpha = 0.0, beta = 0.0;
		if (st.done) {
			quit = 1;
		} else {
			const double gamma = sums_slot->sums[0], delta = sums_slot->sums[1], rr = sums_slot->sums[2];
			st.rr = rr;
			if (!isfinite(rr) || !isfinite(gamma) || !isfinite(delta)) {
				st.done = 2;
			} else if (!(rr > st.tol2)) {
				st.done = 1;
			} else if (st.iter >= st.max_iter) {
				st.done = 3;
			} else {
				beta = first ? 0.0 : gamma / st.rz;
				const double denom = first ? delta : delta - beta * gamma / st.alpha;
				alpha = gamma / denom;
				if (!(gamma > 0.0) || !(denom > 0.0) || !isfinite(alpha)) { st.done = 2; }
				st.pq = denom;
			}
			if (st.done) {
				quit = 1;
			} else {
				st.rz    = gamma;
				st.alpha = alpha;
				st.beta  = beta;
				st.iter += 1;
			}
		}
		if (blockIdx.x == 0) { *out = st; }
		sh_alpha = alpha;
		sh_beta  = beta;
		sh_quit  = quit;
	}
	__syncthreads();
	if (sh_quit) { return; }
	const T alpha = static_cast<T>(sh_alpha), beta = static_cast<T>(sh_beta);
	double acc = 0.0;
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		const T pv = first ? z[i] : z[i] + beta * p[i];
		const T sv = first ? w[i] : w[i] + beta * s[i];
		p[i] = pv;
		s[i] = sv;
		x[i] += alpha * pv;
		const T rv = r[i] - alpha * sv;
		r[i] = rv;
		acc += static_cast<double>(rv) * static_cast<double>(rv);
	}
	double accv[1] = {acc}, sum[1];
	block_sum<1>(accv, sum);
	if (threadIdx.x == 0) { prr[blockIdx.x] = sum[0]; }
}

template <typename T>
void cg_run_poly_sr(RankSet& R, int max_iterations, float tol)
{
	fi_ctx* c0 = R[0];
	hipStream_t st = c0->stream;
	const int terms = c0->poly_terms;
	if (max_iterations <= 0) {
		const int64_t dflt = 2 * static_cast<int64_t>(c0->g.gn[0]) * c0->g.gn[1] * c0->g.gn[2];
		max_iterations = dflt > std::numeric_limits<int>::max() ? std::numeric_limits<int>::max() : static_cast<int>(dflt);
	}
	const double tolerance = tol > 0 ? static_cast<double>(tol) : static_cast<double>(std::numeric_limits<float>::epsilon());
	for (fi_ctx* c : R) {
		ensure_poly_vectors<T>(c);
		const size_t bytes = sizeof(T) * c->g.nloc;
		if (c->mg_r.bytes < bytes) {
			c->mg_r.alloc(bytes);
			FI_HIP_TRY(hipMemsetAsync(c->mg_r.p, 0, bytes, c->stream));
		}
	}
	if (!(c0->poly_lambda > 0)) { estimate_poly_lambda<T>(R); }
	EventPair timer;  // (destroyed on every way out: a coarse level's breakdown, a timeout)
	const hipEvent_t e0 = timer.e0, e1 = timer.e1;
	FI_HIP_TRY(hipEventRecord(e0, st));

	double lam_scale = 1.0;
	if (const char* env = test_switch("FI_POLY_LAMBDA_SCALE")) { lam_scale = atof(env) > 0 ? atof(env) : 1.0; }
	const double lam = (c0->poly_lambda > 1.0 ? c0->poly_lambda : 1.0) * lam_scale;
	const double hi = 1.1 * lam, lo = hi / (c0->poly_ratio > 1.0 ? c0->poly_ratio : 10.0);
	const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
	std::vector<double> c1s, c2s;
	{
		double rho = 1.0 / sigma;
		for (int k = 1; k < terms; ++k) {
			const double rho_new = 1.0 / (2.0 * sigma - rho);
			c1s.push_back(rho_new * rho);
			c2s.push_back(2.0 * rho_new / delta);
			rho = rho_new;
		}
	}
	CgScalars init{};
	init.tol2     = tolerance * tolerance;
	init.max_iter = max_iterations;
	init.rz       = 1.0;
	init.alpha    = 1.0;
	reset_scalars(R, init);

	auto nbf_of = [](fi_ctx* c) { const int b = stream_blocks(c->g.nown); return b > 1024 ? 1024 : b; };
	auto region = [](fi_ctx* c, int k) { return c->partial.as<double>() + static_cast<size_t>(k) * c->max_blocks; };
	auto slot   = [](fi_ctx* c, int k) { return c->scal.as<CgScalars>() + k; };
	int n_exchanges = 0, n_reductions = 0;
	auto cross = [&](int nvec) {
		++n_reductions;
		if (R.size() > 1) {
			hipLaunchKernelGGL(k_group_sum, dim3(1), dim3(1), 0, st, c0->group_scal.as<CgScalars*>(), static_cast<int>(R.size()), nvec, 2);
		} else if (c0->nranks > 1) {
			allreduce_sum(c0, slot(c0, 2)->sums, nvec);
		}
	};
	bool ghosts_scaled = true;
	for (const fi_ctx* c : R) { ghosts_scaled = ghosts_scaled && c->scaling_ghosts; }
	const bool z0_on_load = ghosts_scaled && terms > 2 && c0->march.valid && !test_switch("FI_NO_Z0_ON_LOAD");
	const int  deep_width = 2 * (terms - 1);
	const bool deep = z0_on_load && c0->halo >= deep_width && c0->min_slab >= deep_width && !test_switch("FI_NO_DEEP_HALO");
	const Vec ZA = &fi_ctx::mg_x, ZB = &fi_ctx::mg_d, W = &fi_ctx::mg_r, S = &fi_ctx::q;

	// z = M r (the polynomial of cg_run_poly, without its sampling and overlap); returns the buffer holding z; the partials
	// of r . z are in region 2
	auto polynomial = [&]() -> Vec {
		if (!z0_on_load) {
			for (fi_ctx* c : R) {
				hipLaunchKernelGGL((k_cheb_first16<T>), dim3(stream_blocks(c->g.nown)), dim3(kThreads), 0, c->stream, c->g.nown,
				                   vown<T>(c, &fi_ctx::r), c->dinv16.as<unsigned short>() + c->g.own_first, vown<T>(c, ZA),
				                   static_cast<T>(1.0 / theta));
			}
		}
		Vec zin = ZA, zout = ZB;
		for (int k = 1; k < terms; ++k) {
			const bool pro = z0_on_load && k == 1;
			const int  ext = deep ? 2 * (terms - 1 - k) : 0;
			if (deep) {
				if (k == 1) {
					halo_exchange(R, &fi_ctx::r, deep_width);
					++n_exchanges;
				}
			} else {
				halo_exchange(R, pro ? static_cast<Vec>(&fi_ctx::r) : zin);
				++n_exchanges;
			}
			for (fi_ctx* c : R) {
				if (pro) {
					stencil_cheb_step(c, c->r.p, nullptr, c->r.p, (c->*zout).p, c1s[0], c2s[0], region(c, 2), 0, 0.0, 1.0 / theta, nullptr, ext);
				} else {
					stencil_cheb_step(c, (c->*zin).p, k == 1 ? nullptr : (c->*zout).p, c->r.p, (c->*zout).p, c1s[k - 1], c2s[k - 1],
					                  region(c, 2), 0, k == 2 ? 1.0 / theta : 0.0, 0.0, nullptr, ext);
				}
			}
			std::swap(zin, zout);
		}
		return zin;
	};
	// r = b - A x with its norm (and b's): start (phase 0) and verification (phase 2)
	auto true_residual = [&](int phase) {
		apply_exchanged(R, &fi_ctx::x, S, nullptr);
		++n_exchanges;
		for (fi_ctx* c : R) {
			const int64_t o = c->g.own_first;
			hipLaunchKernelGGL((k_sr_residual<T>), dim3(nbf_of(c)), dim3(kThreads), 0, c->stream, c->g.nown, c->atb.as<T>() + o,
			                   c->q.as<T>() + o, c->r.as<T>() + o, region(c, 1), region(c, 3));
			hipLaunchKernelGGL(k_reduce3, dim3(1), dim3(kThreads), 0, c->stream, slot(c, 2), region(c, 1), nbf_of(c), region(c, 3), nbf_of(c),
			                   static_cast<const double*>(nullptr), 0);
		}
		cross(2);
		for (fi_ctx* c : R) { hipLaunchKernelGGL(k_sr_setup, dim3(1), dim3(1), 0, c->stream, slot(c, 0), slot(c, 2), phase); }
	};
	int  replace_every = 16, since_replace = 0;
	if (const char* env = test_switch("FI_SR_REPLACE")) { replace_every = atoi(env); }
	int  state = 0;      // the slot holding the current scalar record (0 / 1 alternate: no block reads the slot its kernel writes)
	bool first = true;   // the next update starts the recurrences (p = z, s = w)
	auto iterate = [&]() {
		const Vec zfin = polynomial();
		apply_exchanged(R, zfin, W, +[](fi_ctx* c) -> double* { return c->partial.as<double>(); });
		++n_exchanges;
		for (fi_ctx* c : R) {
			hipLaunchKernelGGL(k_reduce3, dim3(1), dim3(kThreads), 0, c->stream, slot(c, 2), region(c, 2), stencil_cheb_partials(c),
			                   region(c, 0), apply_num_partials(c), region(c, 1), nbf_of(c));
		}
		cross(3);
		for (fi_ctx* c : R) {
			const int64_t o = c->g.own_first;
			hipLaunchKernelGGL((k_sr_update<T>), dim3(nbf_of(c)), dim3(kThreads), 0, c->stream, c->g.nown, slot(c, state), slot(c, state ^ 1),
			                   slot(c, 2), first ? 1 : 0, vown<T>(c, zfin), vown<T>(c, W), c->p.as<T>() + o, c->q.as<T>() + o,
			                   c->x.as<T>() + o, c->r.as<T>() + o, region(c, 1));
		}
		state ^= 1;
		first = false;
		// The recurrence s = w + beta s drifts from A p in fp32 (a 2-D system of 150 outer iterations took 168 over three
		// slabs): every 16th step s is recomputed as A p -- one more apply and exchange per 16 outer iterations.  In fp64 the
		// recurrence tracks the two-reduction form to rounding for hundreds of steps (563 = 563).  FI_SR_REPLACE: tests.
		if (replace_every > 0 && ++since_replace >= replace_every) {
			since_replace = 0;
			apply_exchanged(R, &fi_ctx::p, S, nullptr);
			++n_exchanges;
		}
	};
	auto read_state = [&]() -> const CgScalars& {
		FI_HIP_TRY(hipMemcpyAsync(c0->scal_host, slot(c0, state), sizeof(CgScalars), hipMemcpyDeviceToHost, st));
		FI_HIP_TRY(hipStreamSynchronize(st));
		return *c0->scal_host;
	};

	true_residual(0);
	double limit_s = 600.0;
	if (const char* env = getenv("FI_SOLVE_TIMEOUT_S")) { limit_s = atof(env); }
	const auto wall0 = std::chrono::steady_clock::now();
	bool timed_out = false;
	int  restarts_left = c0->verify_residual ? 3 : 0;
	int  next_burst = c0->last_outer_iterations > 0 ? (c0->last_outer_iterations < 64 ? c0->last_outer_iterations + 1 : 64) : 4;
	for (;;) {
		const CgScalars& h = read_state();
		if (h.done) {
			if (h.done != 1 || restarts_left <= 0) { break; }
			--restarts_left;  // the recurrence's residual met the tolerance: check b - A x, go on from it if it misses
			if (state != 0) {  // (k_sr_setup works on slot 0)
				for (fi_ctx* c : R) { FI_HIP_TRY(hipMemcpyAsync(slot(c, 0), slot(c, 1), sizeof(CgScalars), hipMemcpyDeviceToDevice, c->stream)); }
				state = 0;
			}
			// (the operator kernels exit at once while the stop flag of slot 0 is up)
			for (fi_ctx* c : R) { hipLaunchKernelGGL(k_set_done, dim3(1), dim3(1), 0, c->stream, slot(c, 0), 0); }
			true_residual(2);
			first = true;
			continue;
		}
		if (timed_out_anywhere(R, std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() > limit_s)) {
			timed_out = true;
			break;
		}
		for (int k = 0; k < next_burst; ++k) { iterate(); }
		next_burst = 2;
		FI_HIP_TRY(hipGetLastError());
	}
	FI_HIP_TRY(hipEventRecord(e1, st));
	FI_HIP_TRY(hipEventSynchronize(e1));
	float ms = 0;
	FI_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	const CgScalars h = *c0->scal_host;
	for (fi_ctx* c : R) {
		c->stats.spmv_samples = 0;
		c->stats.spmv_ms_avg  = 0.0;
		c->stats.spmv_bytes   = apply_algorithmic_bytes(c);
		c->stats.prec_samples = 0;
		c->stats.prec_ms_avg  = 0.0;
		c->stats.prec_bytes   = 0.0;
		c->stats.operator_applies = h.iter * terms + 1 + h.restarts;
		c->stats.halo_exchanges = n_exchanges;
		c->stats.reductions     = n_reductions;
		c->last_outer_iterations = h.iter;
		c->stats.solve_ms     = ms;
		c->stats.iterations   = h.iter;
		c->stats.converged    = (!timed_out && (h.done == 4 || h.done == 5 || (h.done == 1 && !c0->verify_residual))) ? 1 : 0;
		c->stats.rel_residual = h.bb > 0 ? std::sqrt(h.rr / h.bb) : 0.0;
		c->stats.restarts     = h.restarts;
		c->stats.verified_residual = (h.restarts > 0 && h.bb > 0) ? std::sqrt(h.true_rr / h.bb) : -1.0;
		if (h.done == 4) { FI_HIP_TRY(hipMemsetAsync(c->x.p, 0, sizeof(T) * c->g.nloc, c->stream)); }
	}
	FI_REQUIRE(h.done != 2, FI_ERR_BREAKDOWN, "CG breakdown in the single-reduction recurrence (r.M r or p.A p not positive)");
	FI_REQUIRE(!timed_out, FI_ERR_TIMEOUT, "solve stopped by the wall-clock guard (FI_SOLVE_TIMEOUT_S = %g s) after %d iterations, "
	           "relative residual %g", limit_s, h.iter, h.bb > 0 ? std::sqrt(h.rr / h.bb) : 0.0);
}

// polynomial PCG; if its preconditioner stays indefinite after two widenings of the interval: the Jacobi diagonal, from
// the last iterate (x is finite: the step that breaks down does not touch it).  Over slabs the single-reduction form
// runs first (cg_run_poly_sr); a breakdown there hands over to the two-reduction form, which knows how to widen.
template <typename T>
void cg_run_poly_or_jacobi(RankSet& R, int max_iterations, float tol)
{
	if ((R.size() > 1 || R[0]->nranks > 1 || test_switch("FI_FORCE_SINGLE_REDUCTION")) && !test_switch("FI_NO_SINGLE_REDUCTION")) {
		try {
			cg_run_poly_sr<T>(R, max_iterations, tol);
			return;
		} catch (const Fail& f) {
			if (f.code != FI_ERR_BREAKDOWN) { throw; }
		}
	}
	try {
		cg_run_poly<T>(R, max_iterations, tol);
	} catch (const Fail& f) {
		if (f.code != FI_ERR_BREAKDOWN) { throw; }
		cg_run<T>(R, max_iterations, tol);
	}
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

fi_ctx* create_ctx(int ndim, const int* sizes, int dtype, int rank, int nranks);

// Coarser replicas of the assembled problem, each with the lattice halved (fine point 2i <-> coarse point i):
//   * model weights rescaled so that the smoothness energy stays the same functional of the field: a k-th
//     difference on the coarse lattice is 2^k times the fine one and there are 2^D times fewer rows, hence
//     w_k,coarse^2 = w_k^2 * 2^D / 4^k  (gradient_smoothness like k = 2);
//   * the same data points, positions halved; gradients double in coarse lattice units and their rows get
//     half the weight (value rows keep theirs).
// Levels stop when an axis would drop below 8 points.  Hand-built rows (fi_add_rows_coo) have no geometry to
// coarsen: contexts holding them stay single-level.
// build_stream: the stream the levels are ASSEMBLED on (fi_assemble runs this function on a helper thread beside the
// assembly of the finest level); the levels then go back to the solver stream of `c`.
// Pure arithmetic every rank agrees on: how many of the wanted levels exist (extents >= 8) and from which level on the
// slabs would be thinner than max(halo, 4) planes -- the REPLICATED TAIL: those levels are whole lattices on every rank.
// first_tail = levels + 1 when there is none.
int plan_levels(const fi_ctx* c, int* first_tail)
{
	const int D = c->g.ndim;
	int n[3] = {c->g.gn[0], c->g.gn[1], c->g.gn[2]};
	std::vector<int> lo(c->nranks), hi(c->nranks);
	for (int r = 0; r < c->nranks; ++r) {
		lo[r] = static_cast<int>(static_cast<int64_t>(r) * n[D - 1] / c->nranks);
		hi[r] = static_cast<int>(static_cast<int64_t>(r + 1) * n[D - 1] / c->nranks);
	}
	int levels = 0, tail = 0;
	const bool allow_tail = c->nranks > 1 && !test_switch("FI_NO_REPLICATED_TAIL");
	for (int l = 1; l <= c->levels_wanted; ++l) {
		bool ok = true;
		for (int d = 0; d < D; ++d) {
			n[d] = (n[d] + 1) / 2;
			ok = ok && n[d] >= 8;
		}
		if (!ok) { break; }
		if (c->nranks > 1 && !tail) {
			bool thick = true;
			for (int r = 0; r < c->nranks; ++r) {
				lo[r] = (lo[r] + 1) / 2;
				hi[r] = (hi[r] + 1) / 2;
				thick = thick && (hi[r] - lo[r]) >= (c->reach > 4 ? c->reach : 4);
			}
			if (!thick) {
				if (!allow_tail) { break; }
				tail = l;
			}
		}
		levels = l;
	}
	if (first_tail) { *first_tail = tail ? tail : levels + 1; }
	return levels;
}

bool holds_value_rows_only(const fi_ctx* src)
{
	for (const PointBatch* b : src->batches) {
		if (b->n > 0 && b->has_nrm && b->gw != 0.0f) { return false; }
	}
	return src->generic.ntrip == 0;
}

void build_levels(fi_ctx* c, fi_ctx* src = nullptr, hipStream_t build_stream = nullptr)  // src: the context holding the point batches (default: c)
{
	if (c->level != 0) { return; }
	if (!src) { src = c; }
	bool wanted = c->levels_wanted > 0;
	if (wanted && c->generic.ntrip > 0) {
		// generic rows that came from points (gradient kLinearInterpolation) can be re-emitted; hand-built ones cannot
		long from_points = 0;
		for (auto* b : src->batches) {
			if (b->has_nrm && b->gk == FI_GRADIENT_LINEAR_INTERPOLATION && b->gw != 0.0f) { from_points += b->n * c->g.ndim; }
		}
		wanted = from_points == c->generic.nrows;
	}
	if (!wanted) {
		if (c->coarse) {
			fi_ctx_destroy(c->coarse);
			c->coarse = nullptr;
		}
		return;
	}
	fi_ctx* fine = c;
	const int D = c->g.ndim;
	// slab ranges of EVERY rank on the current level (pure arithmetic: all ranks agree on where the levels stop)
	std::vector<int> lo(c->nranks), hi(c->nranks);
	for (int r = 0; r < c->nranks; ++r) {
		lo[r] = static_cast<int>(static_cast<int64_t>(r) * c->g.gn[D - 1] / c->nranks);
		hi[r] = static_cast<int>(static_cast<int64_t>(r + 1) * c->g.gn[D - 1] / c->nranks);
	}
	int first_tail = 0;
	const int nlevels = plan_levels(c, &first_tail);
	std::vector<fi_ctx*> built;
	for (int l = 1; l <= nlevels; ++l) {
		// from first_tail on the levels are whole lattices that every rank assembles -- from ALL the data points, which
		// fi_slab_point_range asks the caller for in that case -- and solves in full (fi_ctx::replicated)
		const bool tail = l >= first_tail;
		int sizes[3] = {1, 1, 1}, cc[3] = {0, 0, 0};
		float shift[3] = {0, 0, 0};
		for (int d = 0; d < D; ++d) {
			sizes[d] = (fine->g.gn[d] + 1) / 2;
			// even extents are halved cell-centred (fi_ctx::cc); along the decomposed axis the transfers then reach two
			// planes beyond the slab, which the ghost planes of model_2 and wider stencils cover
			cc[d] = fine->g.gn[d] % 2 == 0 && (d != D - 1 || c->nranks == 1 || c->reach >= 2 || (tail && fine->replicated)) &&
			        !test_switch("FI_VERTEX_LEVELS");
			shift[d] = 0.5f * (fine->pos_shift[d] - (cc[d] ? 0.5f : 0.0f));
		}
		// coarse plane k sits on fine plane 2k: a rank keeps the coarse planes whose fine plane it owns
		for (int r = 0; r < c->nranks; ++r) {
			lo[r] = (lo[r] + 1) / 2;
			hi[r] = (hi[r] + 1) / 2;
		}
		int min_slab = sizes[D - 1];
		if (!tail) {
			for (int r = 0; r < c->nranks; ++r) { min_slab = hi[r] - lo[r] < min_slab ? hi[r] - lo[r] : min_slab; }
		}
		const int co_nranks = tail ? 1 : c->nranks;
		fi_ctx* co = fine->coarse;
		if (co && (co->g.gn[0] != sizes[0] || co->g.gn[1] != sizes[1] || co->g.gn[2] != sizes[2] || co->dtype != c->dtype ||
		           co->halo != c->halo || co->cc[0] != cc[0] || co->cc[1] != cc[1] || co->cc[2] != cc[2] || co->nranks != co_nranks)) {
			fi_ctx_destroy(co);
			co = nullptr;
		}
		if (co) {  // same shape as last time: keep its HBM, drop its rows
			for (auto* pb : co->pending) { co->pending_pool.push_back(pb); }
			co->pending.clear();
			generic_clear(co);
		} else {
			co = create_ctx(D, sizes, c->dtype, tail ? 0 : c->rank, co_nranks);
			(void)hipStreamDestroy(co->stream);
			co->stream      = c->stream;
			co->owns_stream = false;
			co->owns_comm   = false;
			co->level       = l;
			co->finer       = fine;
			co->verify_residual = 0;
			co->replicated  = tail;
			if (!tail) {
				co->slab_fixed  = true;
				co->slab_lo     = lo[c->rank];
				co->slab_hi     = hi[c->rank];
			}
			co->halo        = c->halo;
			co->reach       = c->reach;
			for (int d = 0; d < 3; ++d) {
				co->cc[d]        = cc[d];
				co->pos_shift[d] = shift[d];
			}
			compute_geom(co, D, sizes);
			fine->coarse    = co;
		}
		co->min_slab = min_slab;
		co->comm = tail ? nullptr : c->comm;
		co->mg_smoother = c->mg_smoother;
		co->mg_safe     = c->mg_safe;
		co->mg_terms    = c->mg_terms;
		co->mg_pratio   = c->mg_pratio;
		co->value_rows_only = src->value_rows_only;  // (agreed over the ranks: fi_assemble)
		co->any_trip        = src->any_trip;
		co->stream = build_stream ? build_stream : c->stream;
		co->defer_scaling_exchange = build_stream != nullptr;  // a helper thread never talks to the neighbours
		const float vol = static_cast<float>(1 << D);
		fi_weights w = fine->w;
		w.model_0 = fine->w.model_0 * std::sqrt(vol);
		w.model_1 = fine->w.model_1 * std::sqrt(vol / 4.0f);
		w.model_2 = fine->w.model_2 * std::sqrt(vol / 16.0f);
		w.model_3 = fine->w.model_3 * std::sqrt(vol / 64.0f);
		w.model_4 = fine->w.model_4 * std::sqrt(vol / 256.0f);
		w.gradient_smoothness = fine->w.gradient_smoothness * std::sqrt(vol / 16.0f);
		co->w = w;
		built.push_back(co);
		fine = co;
	}
	if (fine->coarse) {  // deeper levels left over from an earlier, larger request
		fi_ctx_destroy(fine->coarse);
		fine->coarse = nullptr;
	}
	// The levels are problems of their own, each a chain of small launches with host round trips for its list sizes:
	// rows from the point batches, cells, lists, diagonal.  On a helper's stream (fi_assemble) the levels beyond the first
	// get a thread and a stream each (config 3's assembly 3.7 -> 2.2 ms, config 5's 24.3 -> 21.1 ms, the accurate leg of
	// config 4 16.2 -> 15.9 ms per step: there the fp64 finest level is the longest chain).  FI_SERIAL_LEVEL_CHAINS: one
	// after the other (tests: the same bits).
	auto assemble_level = [src](fi_ctx* co) {
		const int   l  = co->level;
		const float ps = 1.0f / static_cast<float>(1 << l), ns = static_cast<float>(1 << l);
		for (auto* b : src->batches) {
			const float* nrm = b->has_nrm ? b->nrm.as<float>() : nullptr;
			const float* pw  = b->has_pw ? b->pw.as<float>() : nullptr;
			const float* val = b->has_val ? b->val.as<float>() : nullptr;
			const bool   lin = nrm && b->gk == FI_GRADIENT_LINEAR_INTERPOLATION;
			emit_point_rows(co, b->n, b->pos.as<float>(), nrm, pw, val, b->vw, b->vk, lin ? 0.0f : b->gw * ps,
			                lin ? FI_GRADIENT_CELL_EDGES : b->gk, ps, ns);
			if (lin && b->gw != 0.0f) {
				generic_add_gradient_linear(co, b->n, b->pos.as<float>(), nrm, pw, b->gw * ps, ps, ns);
			}
		}
		assemble(co);
		generic_assemble(co);
		stencil_prepare(co);
		operator_prepare(co);
		// the polynomial smoother's scaling (and whether the data pin a small level) with the level's assembly, on its chain's
		// stream, instead of at the head of the first solve (undivided levels: over slabs the ghost planes' diagonal comes later)
		if (co->dtype == FI_F32 && co->g.ndim == 3 && co->mg_smoother == 1 && co->value_rows_only && !co->any_trip &&
		    co->march.valid && co->nranks == 1 && !test_switch("FI_MG_FULL_SMOOTHER")) {
			prepare_safe_scaling(co);
		}
		co->tail_prog_valid = false;
		if (tail_level_supported(co)) { tail_build_operator(co); }  // the small-level engine's view of the data rows
		co->assembled = true;
		co->vectors_ready = co->vectors_ready && co->max_blocks >= apply_num_partials(co);
		co->stats.num_unknowns = co->g.nown;
	};
	const bool chains = build_stream != nullptr && built.size() > 1 && !test_switch("FI_SERIAL_LEVEL_CHAINS");
	if (!chains) {
		for (fi_ctx* co : built) { assemble_level(co); }
	} else {
		struct Go {
			hipEvent_t e = nullptr;
			~Go() { if (e) { (void)hipEventDestroy(e); } }
		} go_holder;
		FI_HIP_TRY(hipEventCreateWithFlags(&go_holder.e, hipEventDisableTiming));
		const hipEvent_t go = go_holder.e;
		FI_HIP_TRY(hipEventRecord(go, build_stream));  // (behind the caller's wait for the point batches)
		std::vector<std::thread> workers;
		std::vector<int>         codes(built.size(), FI_OK);
		std::vector<std::string> msgs(built.size());
		for (size_t i = 1; i < built.size(); ++i) {
			fi_ctx* co = built[i];
			if (!co->build_stream) {
				FI_HIP_TRY(hipStreamCreateWithFlags(&co->build_stream, hipStreamNonBlocking));
				FI_HIP_TRY(hipEventCreateWithFlags(&co->ev_build, hipEventDisableTiming));
			}
			FI_HIP_TRY(hipStreamWaitEvent(co->build_stream, go, 0));
			co->stream = co->build_stream;
		}
		auto guarded = [&](size_t i) {
			try {
				FI_HIP_TRY(hipSetDevice(c->device));
				assemble_level(built[i]);
			} catch (const Fail& f) {
				codes[i] = f.code;
				msgs[i]  = fi_last_error();
			} catch (...) {
				codes[i] = FI_ERR_HIP;
				msgs[i]  = "unexpected exception while assembling a coarser level";
			}
		};
		for (size_t i = 1; i < built.size(); ++i) {
			try {
				workers.emplace_back(guarded, i);
			} catch (...) {  // no thread to be had: this one on the caller's thread, behind the first level
				workers.emplace_back();
			}
		}
		guarded(0);
		for (size_t i = 1; i < built.size(); ++i) {
			std::thread& t = workers[i - 1];
			if (t.joinable()) { t.join(); } else { guarded(i); }
		}
		// the caller orders `build_stream` against the solver stream: the other chains end in it
		for (size_t i = 1; i < built.size(); ++i) {
			(void)hipEventRecord(built[i]->ev_build, built[i]->build_stream);
			(void)hipStreamWaitEvent(build_stream, built[i]->ev_build, 0);
		}
		for (size_t i = 0; i < built.size(); ++i) {
			if (codes[i] != FI_OK) {
				for (size_t k = 1; k < built.size(); ++k) { (void)hipStreamSynchronize(built[k]->build_stream); }
				set_error("%s", msgs[i].c_str());
				throw Fail{codes[i]};
			}
		}
	}
	for (fi_ctx* l = c->coarse; l; l = l->coarse) { l->stream = c->stream; }  // the caller orders the two streams
	// the tail of the hierarchy the small-level engine runs (fi_tail.h): from the coarsest level up while the levels qualify
	{
		std::vector<fi_ctx*> chain;
		for (fi_ctx* l = c->coarse; l; l = l->coarse) { chain.push_back(l); }
		bool ok = true;
		for (size_t k = chain.size(); k-- > 0;) {
			ok = ok && tail_level_supported(chain[k]) && static_cast<int>(chain.size() - k) <= kTailMaxLevels;
			chain[k]->tail_ok = ok;
			chain[k]->tail_prog_valid = false;
		}
	}
	// smoother bounds of the V-cycle (a global power method over all slabs) are estimated by the next multigrid solve
	for (fi_ctx* l = c; l; l = l->coarse) { l->lambda_max = 0; }
}

// fp32 replica of an FI_F64 context for the mixed-precision solve: same lattice, same slab, same weights, the
// same data points (re-emitted from the batches kept in HBM), with the levels and solver options of `c`.
// In three parts, so that fi_assemble can run the replica's finest level and the replica's coarser levels on two helper
// threads beside the fp64 finest level: twin_prepare (the context; cheap, on the caller's thread), twin_assemble (rows +
// finest level on `stream`), build_levels(c->twin, c, stream) and twin_finish.
// The replica's finest level on the LUMPED operator (fi_ctx::lumped): value rows only, a 3-D lattice the marching kernel
// covers, the V-cycle with the polynomial smoother.  FI_NO_LUMPED_TWIN: the replica assembles its own cells (tests).
bool lumped_twin_wanted(const fi_ctx* c)
{
	if (test_switch("FI_NO_LUMPED_TWIN") || test_switch("FI_MG_FULL_SMOOTHER") || test_switch("FI_NO_MARCH")) { return false; }
	const fi_weights& w = c->w;
	return c->g.ndim == 3 && c->mg_mode == 1 && c->mg_smoother == 1 && c->levels_wanted > 0 && c->value_rows_only &&
	       c->generic.ntrip == 0 && !c->any_trip && c->g.gn[0] >= 4 && !(w.model_3 > 0 || w.model_4 > 0 || w.gradient_smoothness > 0) &&
	       (w.model_1 > 0 || w.model_2 > 0);
}

__global__ __launch_bounds__(kThreads) void k_fill_f64(int64_t n, double v, double* __restrict__ out)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kThreads) {
		out[i] = v;
	}
}
// dlump = max(A 1 - D w0^2, 0) (the model_0 rows [w0] are diagonal already and stay with the model part); the replica's
// `diag` starts as dlump, k_model_diag adds the model diagonal
__global__ __launch_bounds__(kThreads) void k_lumped_diag(int64_t n, const double* __restrict__ a1, double model0,
                                                           float* __restrict__ dlump, float* __restrict__ diag)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kThreads) {
		const double v = a1[i] - model0;
		const float  f = v > 0.0 ? static_cast<float>(v) : 0.0f;
		dlump[i] = f;
		diag[i]  = f;
	}
}

// The lumped replica of an ASSEMBLED fp64 context, on the context's stream: row sums of the data term by one apply of the
// fp64 operator to the vector of ones (every model row of order >= 1 sums to zero), then the replica's diagonal and scalings.
void twin_assemble_lumped(fi_ctx* c)
{
	fi_ctx* t = c->twin;
	t->stream = c->stream;
	for (auto* pb : t->pending) { t->pending_pool.push_back(pb); }
	t->pending.clear();
	generic_clear(t);
	assemble(t);  // no rows: atb and diag zeroed, no cells
	const Geom& g = c->g;
	t->dlump.alloc(sizeof(float) * g.nloc);
	FI_HIP_TRY(hipMemsetAsync(t->dlump.p, 0, sizeof(float) * g.nloc, c->stream));
	ensure_vectors(c);
	FI_HIP_TRY(hipMemsetAsync(c->scal.p, 0, sizeof(CgScalars), c->stream));  // (the operator kernels exit at once while the stop flag of the last solve is up)
	hipLaunchKernelGGL(k_fill_f64, dim3(stream_blocks(g.nloc)), dim3(kThreads), 0, c->stream, g.nloc, 1.0, c->p.as<double>());
	apply_AtA(c, c->p.p, c->q.p, nullptr);
	const double w0 = c->w.model_0 > 0 ? static_cast<double>(c->w.model_0) : 0.0;
	hipLaunchKernelGGL(k_lumped_diag, dim3(stream_blocks(g.nown)), dim3(kThreads), 0, c->stream, g.nown,
	                   c->q.as<double>() + g.own_first, g.ndim * w0 * w0, t->dlump.as<float>() + g.own_first,
	                   t->diag.as<float>() + g.own_first);
	FI_HIP_TRY(hipGetLastError());
	generic_assemble(t);
	stencil_prepare(t);
	operator_prepare(t);
}

fi_ctx* twin_prepare(fi_ctx* c)
{
	if (c->level != 0) { return nullptr; }
	if (!(c->mixed && c->dtype == FI_F64)) {
		if (c->twin) {
			fi_ctx_destroy(c->twin);
			c->twin = nullptr;
		}
		return nullptr;
	}
	const int D = c->g.ndim;
	{
		long from_points = 0;
		for (auto* b : c->batches) {
			if (b->has_nrm && b->gk == FI_GRADIENT_LINEAR_INTERPOLATION && b->gw != 0.0f) { from_points += b->n * D; }
		}
		FI_REQUIRE(from_points == c->generic.nrows, FI_ERR_UNSUPPORTED,
		           "mixed precision needs rows that came from points; fi_add_rows_coo rows cannot be replicated");
	}
	int sizes[3] = {c->g.gn[0], c->g.gn[1], c->g.gn[2]};
	fi_ctx* t = c->twin;
	if (t && t->halo != c->halo) {
		fi_ctx_destroy(t);
		t = nullptr;
	}
	if (t) {
		for (auto* pb : t->pending) { t->pending_pool.push_back(pb); }
		t->pending.clear();
		generic_clear(t);
	} else {
		t = create_ctx(D, sizes, FI_F32, c->rank, c->nranks);
		(void)hipStreamDestroy(t->stream);
		t->stream      = c->stream;
		t->owns_stream = false;
		t->owns_comm   = false;
		t->slab_fixed  = true;
		t->slab_lo     = c->slab_lo;
		t->slab_hi     = c->slab_hi;
		t->halo        = c->halo;
		t->reach       = c->reach;
		compute_geom(t, D, sizes);
		c->twin = t;
	}
	t->comm            = c->comm;
	t->stream          = c->stream;
	t->defer_scaling_exchange = false;
	t->w               = c->w;
	t->model_set       = true;
	t->verify_residual = 0;
	t->levels_wanted   = c->levels_wanted;
	t->coarse_tol      = c->coarse_tol;
	t->mg_mode         = c->mg_mode;
	t->mg_smoother     = c->mg_smoother;
	t->mg_safe         = c->mg_safe;
	t->mg_terms        = c->mg_terms;
	t->mg_pratio       = c->mg_pratio;
	t->min_slab        = c->min_slab;
	t->poly_terms      = c->poly_terms;   // (the coarse-to-fine start on the replica solves its levels with them)
	t->poly_ratio      = c->poly_ratio;
	t->value_rows_only = c->value_rows_only;  // (agreed over the ranks: fi_assemble)
	t->any_trip        = c->any_trip;
	t->lumped          = lumped_twin_wanted(c);
	return t;
}

void twin_assemble(fi_ctx* c, hipStream_t build_stream)
{
	fi_ctx* t = c->twin;
	if (t->lumped) {  // (fi_assemble's helper threads never get here: the lumped form needs the assembled fp64 operator)
		FI_REQUIRE(build_stream == nullptr, FI_ERR_STATE, "the lumped replica is built behind the fp64 level");
		twin_assemble_lumped(c);
		return;
	}
	if (build_stream) {
		t->stream = build_stream;
		t->defer_scaling_exchange = true;  // a helper thread never talks to the neighbours
	}
	for (auto* b : c->batches) {
		const float* nrm = b->has_nrm ? b->nrm.as<float>() : nullptr;
		const float* pw  = b->has_pw ? b->pw.as<float>() : nullptr;
		const float* val = b->has_val ? b->val.as<float>() : nullptr;
		const bool   lin = nrm && b->gk == FI_GRADIENT_LINEAR_INTERPOLATION;
		emit_point_rows(t, b->n, b->pos.as<float>(), nrm, pw, val, b->vw, b->vk, lin ? 0.0f : b->gw,
		                lin ? FI_GRADIENT_CELL_EDGES : b->gk, 1.0f, 1.0f);
		if (lin && b->gw != 0.0f) { generic_add_gradient_linear(t, b->n, b->pos.as<float>(), nrm, pw, b->gw, 1.0f, 1.0f); }
	}
	assemble(t);
	generic_assemble(t);
	stencil_prepare(t);
	operator_prepare(t);
}

void twin_finish(fi_ctx* c)
{
	fi_ctx* t = c->twin;
	t->assembled = true;
	t->vectors_ready = t->vectors_ready && t->max_blocks >= apply_num_partials(t);
	t->stats.num_unknowns = t->g.nown;
	t->stats.num_levels = 1;
	for (fi_ctx* l = t->coarse; l; l = l->coarse) { t->stats.num_levels += 1; }
}

void build_twin(fi_ctx* c)  // the three parts one after the other, on the context's stream
{
	if (!twin_prepare(c)) { return; }
	twin_assemble(c, nullptr);
	build_levels(c->twin, c);
	twin_finish(c);
}

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
		FI_HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
		c->halo = 1;
		compute_geom(c, ndim, sizes);
		c->scal.alloc(3 * sizeof(CgScalars));  // [0]: the state every kernel and the host look at, [1]: mid-iteration copy,
		                                       // [2]: landing place of the dot products summed over slabs (rank sets)
		FI_HIP_TRY(hipMemset(c->scal.p, 0, 2 * sizeof(CgScalars)));
		FI_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->scal_host), sizeof(CgScalars), hipHostMallocDefault));
		// default Weights (field_interpolation.hpp:75-95)
		c->w = fi_weights{1.0f, 1.0f, 0.0f, 0.0f, 0.5f, 0.0f, 0.0f, 0.0f, FI_VALUE_LINEAR_INTERPOLATION,
		                  FI_GRADIENT_CELL_EDGES};
	} catch (...) {
		fi_ctx_destroy(c);
		throw;
	}
	return c;
}

}  // namespace
}  // namespace fi

// ===================================================================================================
// C ABI

#define FI_API_BEGIN try {
#define FI_API_END                                                   \
	}                                                                \
	catch (const fi::Fail& f) { return f.code; }                     \
	catch (const std::exception& e)                                  \
	{                                                                \
		fi::set_error("exception: %s", e.what());                    \
		return FI_ERR_INVALID;                                       \
	}                                                                \
	catch (...)                                                      \
	{                                                                \
		fi::set_error("unknown exception");                          \
		return FI_ERR_INVALID;                                       \
	}                                                                \
	return FI_OK;

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
	if (c->level_stream) { (void)hipStreamDestroy(c->level_stream); }
	if (c->ev_level) { (void)hipEventDestroy(c->ev_level); }
	if (c->level_stream2) { (void)hipStreamDestroy(c->level_stream2); }
	if (c->build_stream) {
		(void)hipStreamDestroy(c->build_stream);
		(void)hipEventDestroy(c->ev_build);
	}
	if (c->ev_level2) { (void)hipEventDestroy(c->ev_level2); }
	if (c->comm_stream) {
		(void)hipStreamDestroy(c->comm_stream);
		(void)hipEventDestroy(c->ev_ready);
		(void)hipEventDestroy(c->ev_halo);
	}
	if (c->comm && c->owns_comm) { fi::comm_destroy(c->comm); }
	if (c->scal_host) { (void)hipHostFree(c->scal_host); }
	if (c->stream && c->owns_stream) { (void)hipStreamDestroy(c->stream); }
	delete c;
	return FI_OK;
}

int fi_memory_pool(long long keep_bytes, long long* cached_bytes)
{
	FI_API_BEGIN
	const size_t keep = keep_bytes < 0 ? ~size_t(0) : static_cast<size_t>(keep_bytes);
	const size_t left = fi::pool_trim(keep);
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

int fi_assemble(fi_ctx* c)
{
	FI_API_BEGIN
	fi::check_ctx(c);
	fi::bind_device(c);
	fi::EventPair timer;  // (destroyed on every way out: a coarse level's breakdown, a timeout)
	const hipEvent_t e0 = timer.e0, e1 = timer.e1;
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
	c->min_slab = c->nranks > 1 ? c->g.gn[c->g.ndim - 1] / c->nranks : c->g.gn[c->g.ndim - 1];
	if (c->nranks > 1 && (want_halo != c->halo || want_reach != c->reach)) {
		c->halo  = want_halo;
		c->reach = want_reach;
		int sizes[3] = {c->g.gn[0], c->g.gn[1], c->g.gn[2]};
		fi::compute_geom(c, c->g.ndim, sizes);
		c->vectors_ready = false;
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
	const bool beside = (c->levels_wanted > 0 || mixed64) && !c->any_trip && !fi::test_switch("FI_SERIAL_LEVELS");
	if (beside && mixed64) {  // (levels an earlier, unmixed assemble may have left on this context)
		const int keep = c->levels_wanted;
		c->levels_wanted = 0;
		fi::build_levels(c);
		c->levels_wanted = keep;
	}
	if (beside) {
		if (!c->level_stream) {
			FI_HIP_TRY(hipStreamCreateWithFlags(&c->level_stream, hipStreamNonBlocking));
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
				FI_HIP_TRY(hipStreamCreateWithFlags(&c->level_stream2, hipStreamNonBlocking));
				FI_HIP_TRY(hipEventCreateWithFlags(&c->ev_level2, hipEventDisableTiming));
			}
			FI_HIP_TRY(hipStreamWaitEvent(c->level_stream2, c->ev_level, 0));
		}
		auto guarded = [&](auto&& work, int* code, std::string* msg) {
			try {
				FI_HIP_TRY(hipSetDevice(c->device));
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
		std::thread helper, helper2;
		try {
			if (!lumped) { helper = std::thread(build); }
			if (mixed64) { helper2 = std::thread(build2); }
		} catch (...) {  // no thread to be had: the levels are built below, after the finest level, on their stream
		}
		int main_code = FI_OK;
		c->defer_scaling_exchange = true;  // slabs: the one exchange of the assembly comes after the ranks have agreed (below)
		try {
			fi::assemble(c);
			fi::generic_assemble(c);
			fi::stencil_prepare(c);
			fi::operator_prepare(c);
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
				c->twin->defer_scaling_exchange = true;
				fi::twin_assemble_lumped(c);
			} catch (const fi::Fail& f) {
				main_code = f.code;
			} catch (...) {
				main_code = FI_ERR_HIP;
				fi::set_error("unexpected exception while building the lumped replica");
			}
		}
		if (helper.joinable()) { helper.join(); } else if (main_code == FI_OK && !lumped) { build(); }
		if (mixed64) {
			if (helper2.joinable()) { helper2.join(); } else if (main_code == FI_OK) { build2(); }
			if (helper_code == FI_OK && helper2_code != FI_OK) {
				helper_code = helper2_code;
				helper_msg  = helper2_msg;
			}
		}
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
	FI_HIP_TRY(hipEventSynchronize(e1));
	float ms = 0;
	FI_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	c->stats.assemble_ms  = ms;
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
	*out = c->stats;
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


// ---- loop-back group: all slabs of a decomposition in one process, on one device ------------------
struct fi_group {
	std::vector<fi_ctx*> members;
	int dtype = FI_F32;
};

int fi_group_create(fi_group** out, int ndim, const int* sizes, int dtype, int nranks)
{
	FI_API_BEGIN
	FI_REQUIRE(out != nullptr, FI_ERR_INVALID, "out is null");
	FI_REQUIRE(nranks >= 2, FI_ERR_INVALID, "a group needs at least two slabs");
	auto* g = new fi_group();
	g->dtype = dtype;
	try {
		for (int r = 0; r < nranks; ++r) { g->members.push_back(fi::create_ctx(ndim, sizes, dtype, r, nranks)); }
		fi_ctx* c0 = g->members[0];
		std::vector<fi::CgScalars*> ptrs;
		for (fi_ctx* c : g->members) {
			ptrs.push_back(c->scal.as<fi::CgScalars>());
			if (c != c0) {  // one stream for the whole group: program order is the synchronisation
				(void)hipStreamDestroy(c->stream);
				c->stream      = c0->stream;
				c->owns_stream = false;
			}
		}
		c0->group_scal.alloc(sizeof(fi::CgScalars*) * ptrs.size());
		FI_HIP_TRY(hipMemcpy(c0->group_scal.p, ptrs.data(), sizeof(fi::CgScalars*) * ptrs.size(), hipMemcpyHostToDevice));
	} catch (...) {
		for (fi_ctx* c : g->members) { fi_ctx_destroy(c); }
		delete g;
		throw;
	}
	*out = g;
	FI_API_END
}

int fi_group_destroy(fi_group* g)
{
	if (!g) { return FI_OK; }
	for (size_t i = g->members.size(); i-- > 0;) { fi_ctx_destroy(g->members[i]); }  // member 0 owns the stream
	delete g;
	return FI_OK;
}

int fi_group_size(const fi_group* g) { return g ? static_cast<int>(g->members.size()) : 0; }

fi_ctx* fi_group_rank(fi_group* g, int rank)
{
	if (!g || rank < 0 || rank >= static_cast<int>(g->members.size())) { return nullptr; }
	return g->members[rank];
}

int fi_group_assemble(fi_group* g)
{
	FI_API_BEGIN
	FI_REQUIRE(g != nullptr, FI_ERR_INVALID, "null group");
	{  // the data facts a real decomposition agrees on by an all-reduce (fi_ctx::any_trip)
		bool trip = false, grad = false;
		for (fi_ctx* c : g->members) {
			trip = trip || c->generic.ntrip != 0;
			grad = grad || !fi::holds_value_rows_only(c);
		}
		for (fi_ctx* c : g->members) {
			c->facts_forced = true;
			c->forced_trip  = trip;
			c->forced_grad  = grad;
		}
	}
	for (fi_ctx* c : g->members) {
		const int rc = fi_assemble(c);
		if (rc != FI_OK) { return rc; }
	}
	// coarser levels (and the fp32 replicas of mixed precision with theirs): the loop-back dot-product sum needs
	// the scalar blocks of every member of a level
	// The diagonal's ghost planes, like a process per slab gets them through its transport at assembly time: the scaling
	// on the ghost planes is then the neighbour's, the polynomial's first step forms its operand on load and the deep
	// exchange has the scaling of the whole ghost zone.
	auto ghosts = [&](std::vector<fi_ctx*>& lev) {
		if (lev[0]->nranks <= 1 || lev[0]->g.nown == lev[0]->g.nloc) { return; }
		fi::RankSet R(lev.begin(), lev.end());
		fi::halo_exchange(R, &fi_ctx::diag, lev[0]->min_slab >= lev[0]->halo ? lev[0]->halo : lev[0]->reach);
		for (fi_ctx* c : lev) { fi::operator_rescale_with_ghosts(c); }
	};
	ghosts(g->members);
	auto link_chain = [&](std::vector<fi_ctx*> lev) {
		while (lev[0]) {
			std::vector<fi::CgScalars*> ptrs;
			for (fi_ctx*& c : lev) {
				FI_REQUIRE(c != nullptr, FI_ERR_STATE, "members disagree on the number of levels");
				ptrs.push_back(c->scal.as<fi::CgScalars>());
			}
			ghosts(lev);
			lev[0]->group_scal.alloc(sizeof(fi::CgScalars*) * ptrs.size());
			FI_HIP_TRY(hipMemcpy(lev[0]->group_scal.p, ptrs.data(), sizeof(fi::CgScalars*) * ptrs.size(), hipMemcpyHostToDevice));
			for (fi_ctx*& c : lev) { c = c->coarse; }
		}
	};
	std::vector<fi_ctx*> lev, twins;
	for (fi_ctx* c : g->members) {
		lev.push_back(c->coarse);
		twins.push_back(c->twin);
	}
	link_chain(lev);
	link_chain(twins);
	FI_API_END
}

static void group_ready(fi_group* g)
{
	FI_REQUIRE(g != nullptr, FI_ERR_INVALID, "null group");
	for (fi_ctx* c : g->members) {
		fi::check_assembled(c);
		fi::ensure_vectors(c);
	}
	FI_HIP_TRY(hipSetDevice(g->members[0]->device));
}

int fi_group_solve_cg(fi_group* g, const float* guess, int max_iterations, float tol, float* out, int* iterations,
                      float* rel_residual)
{
	FI_API_BEGIN
	group_ready(g);
	int64_t at = 0;
	fi_ctx* c0 = g->members[0];
	c0->stats.coarse_iterations = 0;
	if (!guess && c0->twin && c0->twin->coarse) {
		fi::twin_cascade_guess(g->members);
	} else if (!guess && c0->coarse) {
		g->dtype == FI_F64 ? fi::cascade_guess<double>(g->members) : fi::cascade_guess<float>(g->members);
	} else {
		for (fi_ctx* c : g->members) {
			if (g->dtype == FI_F64) {
				fi::load_owned<double>(c, c->x, guess ? guess + at : nullptr, FI_HOST);
			} else {
				fi::load_owned<float>(c, c->x, guess ? guess + at : nullptr, FI_HOST);
			}
			at += c->g.nown;
		}
	}
	struct Report {
		fi_ctx* c; int* it; float* rel;
		~Report() { if (it) { *it = c->stats.iterations; } if (rel) { *rel = static_cast<float>(c->stats.rel_residual); } }
	} report{c0, iterations, rel_residual};
	if (c0->mg_mode == 1 && (c0->coarse || (c0->twin && c0->twin->coarse))) {
		g->dtype == FI_F64 ? fi::cg_run_mg<double>(g->members, max_iterations, tol)
		                   : fi::cg_run_mg<float>(g->members, max_iterations, tol);
	} else if (fi::poly_ok(c0)) {
		g->dtype == FI_F64 ? fi::cg_run_poly_or_jacobi<double>(g->members, max_iterations, tol)
		                   : fi::cg_run_poly_or_jacobi<float>(g->members, max_iterations, tol);
	} else {
		g->dtype == FI_F64 ? fi::cg_run<double>(g->members, max_iterations, tol) : fi::cg_run<float>(g->members, max_iterations, tol);
	}
	at = 0;
	for (fi_ctx* c : g->members) {
		if (g->dtype == FI_F64) {
			fi::store_owned<double>(c, c->x, out ? out + at : nullptr, FI_HOST);
		} else {
			fi::store_owned<float>(c, c->x, out ? out + at : nullptr, FI_HOST);
		}
		at += c->g.nown;
	}
	FI_API_END
}

int fi_group_apply_AtA_f64(fi_group* g, const double* x, double* y)
{
	FI_API_BEGIN
	group_ready(g);
	FI_REQUIRE(x && y, FI_ERR_INVALID, "null vector");
	g->dtype == FI_F64 ? fi::apply_f64_run<double>(g->members, x, y) : fi::apply_f64_run<float>(g->members, x, y);
	FI_API_END
}

int fi_group_true_residual(fi_group* g, double* rel)
{
	FI_API_BEGIN
	group_ready(g);
	FI_REQUIRE(rel != nullptr, FI_ERR_INVALID, "null output");
	*rel = g->dtype == FI_F64 ? fi::true_residual_run<double>(g->members) : fi::true_residual_run<float>(g->members);
	FI_API_END
}

int fi_group_tile_pass(fi_group* g, const float* guess, int tile_size, float* out)
{
	FI_API_BEGIN
	group_ready(g);
	FI_REQUIRE(tile_size >= 2 && guess && out, FI_ERR_INVALID, "fi_group_tile_pass: tile_size >= 2, guess and out required");
	FI_REQUIRE(g->members[0]->generic.ntrip == 0, FI_ERR_UNSUPPORTED, "the tile pre-solver over triplet rows needs an undivided lattice");
	int64_t at = 0;
	for (fi_ctx* c : g->members) {
		g->dtype == FI_F64 ? fi::load_owned<double>(c, c->x, guess + at, FI_HOST) : fi::load_owned<float>(c, c->x, guess + at, FI_HOST);
		at += c->g.nown;
	}
	g->dtype == FI_F64 ? fi::tile_pass_run<double>(g->members, tile_size) : fi::tile_pass_run<float>(g->members, tile_size);
	at = 0;
	for (fi_ctx* c : g->members) {
		g->dtype == FI_F64 ? fi::store_owned<double>(c, c->x, out + at, FI_HOST) : fi::store_owned<float>(c, c->x, out + at, FI_HOST);
		at += c->g.nown;
	}
	FI_API_END
}

int fi_group_error_map(fi_group* g, const float* solution, float* out)
{
	FI_API_BEGIN
	group_ready(g);
	FI_REQUIRE(solution && out, FI_ERR_INVALID, "fi_group_error_map needs a solution and an output buffer");
	int64_t at = 0;
	for (fi_ctx* c : g->members) {
		g->dtype == FI_F64 ? fi::load_owned<double>(c, c->x, solution + at, FI_HOST)
		                   : fi::load_owned<float>(c, c->x, solution + at, FI_HOST);
		at += c->g.nown;
	}
	fi::halo_exchange(g->members, &fi_ctx::x);
	at = 0;
	for (fi_ctx* c : g->members) {
		fi::error_map(c, c->x.p, c->q.p);
		g->dtype == FI_F64 ? fi::store_owned<double>(c, c->q, out + at, FI_HOST) : fi::store_owned<float>(c, c->q, out + at, FI_HOST);
		at += c->g.nown;
	}
	FI_API_END
}

int fi_group_get_solution_f64(fi_group* g, double* out)
{
	FI_API_BEGIN
	group_ready(g);
	int64_t at = 0;
	for (fi_ctx* c : g->members) {
		g->dtype == FI_F64 ? fi::get_vec_f64_t<double>(c, c->x, out + at) : fi::get_vec_f64_t<float>(c, c->x, out + at);
		at += c->g.nown;
	}
	FI_API_END
}

}  // extern "C"
