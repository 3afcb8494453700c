// fi_solver_internal.h -- what the solver's translation units share (not installed): fi_cg.hip (rank sets, Jacobi-PCG, the
// plain drivers), fi_poly.hip (polynomial PCG), fi_transfer.hip (level transfers), fi_multigrid.hip (smoothers, V-cycle, V-cycle
// PCG, coarse-to-fine start), fi_levels.hip (coarser levels, replicas), fi_capi.hip (the C ABI), fi_group.hip (loop-back group).
// Small kernels and helpers every unit uses live here in an unnamed namespace (a private copy per unit); everything else is
// declared here and defined once.
#pragma once

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


#include <type_traits>
#include <vector>

#include "fi_internal.h"
#include "fi_transfer.h"
#include "fi_tail.h"

namespace fi {

using RankSet = std::vector<fi_ctx*>;
using Vec = DevBuf fi_ctx::*;

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


/* ======== */
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

/* ======== */
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

/* ======== */
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


/* ======== */
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


/* ======== */
template <typename T>
T* owned(const fi_ctx* c, const DevBuf& b)
{
	return b.as<T>() + c->g.own_first;
}


/* ======== */
// ---- rank sets ---------------------------------------------------------------------------------------
// The solver drivers run over a set of slab contexts in lockstep.  In production the set has ONE member
// (this process's slab; neighbours are reached through RCCL, fi_comm.hip).  A loop-back group
// (fi_group_create) puts ALL slabs of a decomposition into one process on one device and one stream: halo
// planes move with device-to-device copies and the dot products are summed by a tiny kernel.  It exists so
// that the slab geometry, halo widths, global-coordinate boundary masks and cell ownership rules can be
// tested on a single GPU against the undivided solve; it runs the same kernels as the RCCL path.

__global__ void k_group_sum(CgScalars* const* sc, int nranks, int nvec, int slot)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) { return; }
	for (int v = 0; v < nvec; ++v) {
		double s = 0;
		for (int r = 0; r < nranks; ++r) { s += sc[r][slot].sums[v]; }  // fixed order
		for (int r = 0; r < nranks; ++r) { sc[r][slot].sums[v] = s; }
	}
}


/* ======== */
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


// The multigrid drivers work on a rank set like cg_run: one member = an undivided lattice or this process's
// slab (halo planes and sums over RCCL), several members = the loop-back group.  A vector is named by its
// fi_ctx member so that every member's copy can be addressed: base pointer for the operator / transfer
// kernels (ghost planes included), owned part for the elementwise ones.
template <typename T>
T* vbase(fi_ctx* c, Vec v) { return (c->*v).template as<T>(); }
template <typename T>
T* vown(fi_ctx* c, Vec v) { return (c->*v).template as<T>() + c->g.own_first; }


// mixed precision: r32 = r / s and z = s * z32 with s = ||r|| / ||b|| from the device-resident scalars (the V-cycle
// is linear, the scaling only keeps its fp32 operands near the size of b while r shrinks by ten decades).  The scale in
// use is CgScalars::tscale: the residual norm of the PREVIOUS step inside the loop (k_mg_step_mixed writes r32 before
// the new norm exists), of the current one after a restart.
__device__ inline double mixed_scale(const CgScalars* sc)
{
	return (sc->rr > 0.0 && sc->bb > 0.0) ? sqrt(sc->rr / sc->bb) : 1.0;
}
__device__ inline double twin_scale(const CgScalars* sc) { return sc->tscale > 0.0 ? sc->tscale : 1.0; }


// scalar steps of the preconditioned recurrence (single block, thread 0)
enum MgPhase { kMgInitRz = 10, kMgAlpha = 11, kMgResid = 12, kMgBeta = 13, kMgInitRr = 14, kMgFlex = 15 };
constexpr double kFieldMargin = 2.0;  // FI_OPT_FIELD_TOLERANCE: see k_mg_logic(kMgResid)
constexpr int    kFieldMinIter = 3;   // ... no stop before the third iteration: CG's first steps remove the rough part of the error,
                                      // the residual falls and the steps are small while the smooth part has not moved yet (an fp32 2-D
                                      // case of tests/stress_field_rule.py stopped after ONE iteration, 83 % off)
constexpr int    kFieldMinIterGuess = 8;  // ... behind a CALLER'S guess: its error may be smooth -- an old solution after a small change of the
                                      // data -- and a smooth error shows neither in the residual nor in the first steps (the coarsest levels
                                      // are solved loosely); the coarse-to-fine start's error is spread over all modes
constexpr double kFieldFast   = 0.3;  // ... every window gains more than this factor per iteration: the last step and its own ratio are used
constexpr int    kFieldCarry  = 8;    // ... steps carried forward at the rate (CG on an ill-conditioned system converges in stairs: a lull of
                                      // three to seven iterations with tiny steps and a falling residual, the error unchanged, then the next stair)
__global__ __launch_bounds__(kThreads) void k_mg_logic(CgScalars* sc, const double* __restrict__ partial, int count,
                                                        int phase, int twin_sum = 0)
{
	if (sc->done && phase != kMgInitRr && phase != kMgInitRz) { return; }
	// (eight loads in flight per thread: a 4096^2 lattice hands over 16 384 partials, and one load per trip of the loop made
	// this single workgroup a chain of 64 cache round trips -- 37 us per reduction, profiles/r4_by_grid_c3.md; the order of the
	// additions is fixed all the same: run to run the same bits)
	double acc[1] = {0};
	int i = threadIdx.x;
	for (; i + 7 * kThreads < count; i += 8 * kThreads) {
		double v[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) { v[k] = partial[i + k * kThreads]; }
#pragma unroll
		for (int k = 0; k < 8; ++k) { acc[0] += v[k]; }
	}
	for (; i < count; i += kThreads) { acc[0] += partial[i]; }
	double out[1];
	block_sum<1>(acc, out);
	if (threadIdx.x != 0) { return; }
	double s = partial ? out[0] : sc->sums[0];  // no partials: the sum over blocks and ranks is in sums[0]
	// twin_sum: the partials are the fp32 replica's b . x (ChebEpi::dotv) with b = r / t and x = z / t, t = CgScalars::tscale
	if (twin_sum) { s *= twin_scale(sc) * twin_scale(sc); }
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
	case kMgResid: {
		sc->tscale = mixed_scale(sc);  // of the norm k_mg_step_mixed scaled its fp32 residual by
		const double prev = sc->rr;
		sc->rr = s;
		sc->iter += 1;
		bool field_met = false;
		if (sc->field_tol > 0.0) {
			// FI_OPT_FIELD_TOLERANCE: x* - x_k is the sum of the steps still to come.  If the steps shrink by sigma per
			// iteration, ||e_k|| <= ||x_k - x_(k-1)|| sigma / (1 - sigma).  sigma: the SLOWEST mean decay of the residual norm
			// over the last 1, 2, 4 .. 16 iterations and over the whole solve (one step's ratio ||r_k|| / ||r_(k-1)|| alone reads a lucky drop of a
			// slowly converging solve as its rate: tests/stress_field_rule.py, errors up to 16 x the tolerance); the step: the
			// largest of the last 1 + kFieldCarry, each carried forward at that rate.  A margin (kFieldMargin) for the smooth modes,
			// which converge last; no estimate while the residual falls by less than 5 % per iteration.
			double dmax = __longlong_as_double(static_cast<long long>(sc->dmax_bits));  // (k_field_max, just before)
			double xmax = __longlong_as_double(static_cast<long long>(sc->xmax_bits));
			if (sc->field_ranks > 0) {  // over slabs: every slab's pair, summed into place with r.r (k_field_max_slot)
				dmax = 0.0;
				xmax = 0.0;
				for (int r = 0; r < sc->field_ranks; ++r) {
					dmax = fmax(dmax, sc->rank_max[2 * r]);
					xmax = fmax(xmax, sc->rank_max[2 * r + 1]);
				}
			}
			sc->field_est = -1.0;
			const int k = sc->iter;  // (restarts are off under this rule: the iterations count from the solve's start)
			if (sc->bb > 0.0 && xmax > 0.0 && prev > 0.0 && isfinite(s)) {
				const double ra = sqrt(prev / sc->bb), rb = sqrt(s / sc->bb);
				if (k == 1) {
					sc->hist_r[0] = ra;
					sc->hist_r0   = ra;
				}
				sc->hist_r[k % kFieldHist] = rb;
				sc->hist_s[k % kFieldHist] = dmax / xmax;
				sc->hist_t[k % kFieldHist] = sc->alpha * sc->rz;  // (rz: still this step's, kMgBeta replaces it)
				double sigma = rb / ra;
				for (int lag = 2; lag <= k && lag < kFieldHist; lag *= 2) {
					const double r0 = sc->hist_r[(k - lag) % kFieldHist];
					if (r0 > 0.0) {
						const double rho = pow(rb / r0, 1.0 / lag);
						sigma = rho > sigma ? rho : sigma;
					}
				}
				if (k > 2 && sc->hist_r0 > 0.0) {  // ... and over the whole solve (CG's faster late phases are not the tail's rate)
					const double rho = pow(rb / sc->hist_r0, 1.0 / k);
					sigma = rho > sigma ? rho : sigma;
				}
				if (sigma < 0.95 && rb > 0.0) {
					double step = dmax / xmax, f = sigma;
					if (sigma < kFieldFast) {
						// every window gains more than a factor 2 per iteration (a healthy V-cycle): the last step and its own
						// ratio predict the next ones best, and the margin covers the rest (the goldens of configs 2 and 4:
						// estimates 3 to 30 times the true error)
						sigma = rb / ra;
					} else {
						for (int j = 1; j <= kFieldCarry && j < k; ++j) {
							const double sj = sc->hist_s[(k - j) % kFieldHist] * f;
							step = sj > step ? sj : step;
							f *= sigma;
						}
					}
					sc->field_est   = kFieldMargin * step * sigma / (1.0 - sigma);
					sc->field_kappa = sc->field_est / (kFieldMargin * rb);
					field_met = sc->field_est <= sc->field_tol && k >= (sc->field_min_iter > 0 ? sc->field_min_iter : kFieldMinIter);
				}
			}
		}
		if (!isfinite(s)) {
			sc->done = 2;
		} else if (field_met || !(s > sc->tol2)) {
			sc->done = 1;
		} else if (sc->iter >= sc->max_iter) {
			sc->done = 3;
		}
		break;
	}
	case kMgBeta:
		sc->beta = s / sc->rz;
		sc->rz   = s;
		if (!(s > 0.0) || !isfinite(s)) { sc->done = 2; }  // (an indefinite or diverging V-cycle: see cg_run_mg)
		break;
	case kMgFlex: {
		// A preconditioner that depends on its argument (FI_OPT_MG_KCYCLE): the flexible beta, z_(k+1) . (r_(k+1) - r_k) / (z_k . r_k)
		// = -alpha z_(k+1) . q_k / rz_old (s: z_(k+1) . q_k).  kMgBeta has run: beta = rz_new / rz_old, rz = rz_new.
		const double rz_old = sc->beta > 0.0 ? sc->rz / sc->beta : 0.0;
		const double flex = rz_old > 0.0 ? -sc->alpha * s / rz_old : sc->beta;
		if (isfinite(flex)) { sc->beta = flex > 0.0 ? flex : 0.0; }  // (a negative one: restart the directions from z)
		break;
	}
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

// global index of the first owned unknown (seeds of the power method must not depend on the decomposition)
inline int64_t global_first(const fi_ctx* c)
{
	const Geom& g = c->g;
	const int a = g.ndim - 1;
	int64_t plane = 1;
	for (int d = 0; d < a; ++d) { plane *= g.gn[d]; }
	return plane * (g.off[a] + g.own_lo[a]);
}

}  // namespace

// ---- defined once --------------------------------------------------------------------------------------------------
// fi_cg.hip
void compute_geom(fi_ctx* c, int ndim, const int* sizes);
int  model_reach(const fi_weights& w);
void ensure_vectors(fi_ctx* c);
template <typename T> void load_owned(fi_ctx* c, DevBuf& v, const float* src, int memory);
template <typename T> void store_owned(fi_ctx* c, const DevBuf& v, float* dst, int memory);
void halo_exchange(RankSet& R, DevBuf fi_ctx::*vec, int width = 0);
bool overlap_possible(const fi_ctx* c);
void exchange_begin(fi_ctx* c, void* v);
void exchange_wait(fi_ctx* c);
void apply_exchanged(RankSet& R, DevBuf fi_ctx::*in, DevBuf fi_ctx::*out, double* (*partials_of)(fi_ctx*));
void reset_scalars(RankSet& R, const CgScalars& init);
bool timed_out_anywhere(RankSet& R, bool mine);
bool all_ranks_ok(fi_ctx* c, bool mine);
template <typename T> void cg_run(RankSet& R, int max_iterations, float tol);
// iteration counts of finished solves, kept per process by lattice shape, model, level and tolerance: a context that lives
// for one solve (the reference's stateless callers, sparse_linear.cpp:194-196) starts with the predictions the context before
// it had learnt -- they decide when a solve first LOOKS at its stop flag, never what it computes: a solve scheduled by a
// recalled count is watched like any other, only a context's OWN previous solve lets a coarse level run without a look
// (cg_run, `unwatched`).  fi_memory_pool(0) clears the record (and the cached eigenvalue bounds of fi_poly.hip).
void remember_iterations(const fi_ctx* c, int kind, double tol, int iterations);
int  recall_iterations(fi_ctx* c, int kind, double tol);
void forget_iterations();
void forget_lambdas();
template <typename T> void solve_cg_t(fi_ctx* c, const float* guess, int max_iterations, float tol, float* out, int* iterations,
                                      float* rel_residual, int memory);
template <typename T> void tile_pass_run(RankSet& R, int tile_size);
template <typename T> void tile_pass_t(fi_ctx* c, const float* guess, int tile_size, float* out, int memory);
template <typename T> void jacobi_run(RankSet& R, int sweeps, float weight);
template <typename T> void jacobi_t(fi_ctx* c, const float* guess, int sweeps, float weight, float* out, int memory);
template <typename T> double true_residual_run(RankSet& R);
template <typename T> double true_residual_t(fi_ctx* c);
template <typename T> void apply_f64_run(RankSet& R, const double* xin, double* yout);
template <typename T> void apply_f64_t(fi_ctx* c, const double* xin, double* yout);
template <typename T> void get_vec_f64_t(fi_ctx* c, const DevBuf& v, double* out);
// fi_poly.hip
bool poly_ok(const fi_ctx* c);
void remember_lambda(const fi_ctx* c);
template <typename T> void ensure_poly_vectors(fi_ctx* c);
template <typename T> void estimate_poly_lambda(RankSet& R);
template <typename T> void cg_run_poly(RankSet& R, int max_iterations, float tol);
template <typename T> void cg_run_poly_sr(RankSet& R, int max_iterations, float tol);
template <typename T> void cg_run_poly_or_jacobi(RankSet& R, int max_iterations, float tol);
// fi_transfer.hip
LevelPair level_pair(const fi_ctx* fine, const fi_ctx* coarse);
template <typename T> void launch_prolong(const LevelPair& L, const T* coarse, T* fine, int mode, hipStream_t st);
template <typename T, typename TO> void launch_prolong_cubic(const LevelPair& L, const T* coarse, TO* fine, hipStream_t st);
template <typename T> void launch_restrict(const LevelPair& L, const T* fine, T* coarse, hipStream_t st, T* tmp = nullptr, int f_local_planes = 0);
// fi_multigrid.hip
// coarse-to-fine start into x of R; `wide`: the fp64 contexts whose fp32 replicas R are -- the last interpolation then writes
// THEIR x where it can (returns true), and the caller has nothing to widen
template <typename T> bool cascade_guess(RankSet& R, RankSet* wide = nullptr);
void twin_cascade_guess(RankSet& R);
template <typename T> void cg_run_mg(RankSet& R, int max_iterations, float tol);
template <typename T> void mg_alloc(fi_ctx* c);
// fi_levels.hip
int  plan_levels(const fi_ctx* c, int* first_tail);
bool holds_value_rows_only(const fi_ctx* src);
void build_levels(fi_ctx* c, fi_ctx* src = nullptr, hipStream_t build_stream = nullptr);
fi_ctx* twin_prepare(fi_ctx* c);
void twin_assemble(fi_ctx* c, hipStream_t build_stream);
void twin_assemble_lumped(fi_ctx* c);
bool lumped_twin_wanted(const fi_ctx* c);
void twin_finish(fi_ctx* c);
void build_twin(fi_ctx* c);
// fi_capi.hip
fi_ctx* create_ctx(int ndim, const int* sizes, int dtype, int rank, int nranks);
void check_ctx(const fi_ctx* c);
void check_assembled(const fi_ctx* c);
void bind_device(const fi_ctx* c);

namespace {
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


}  // namespace

}  // namespace fi

// nothing throws or aborts across the C ABI
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
