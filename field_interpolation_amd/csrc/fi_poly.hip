// fi_poly.hip -- CG preconditioned by a Chebyshev polynomial in Dinv (A_model + diag(A_data)), its single-reduction form
// over slabs, and the power-method bound of the model operator.  Replaces the diagonal preconditioner of
// Eigen::BiCGSTAB (sparse_linear.cpp:199-206) where FI_OPT_POLY_TERMS asks for it.
#include "fi_solver_internal.h"

namespace fi {

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

void forget_lambdas()  // fi_memory_pool(0): the next context runs its power method again
{
	std::lock_guard<std::mutex> lock(g_lambda_mutex);
	g_lambda_cache.clear();
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


// ---- explicit instantiations (declared in fi_solver_internal.h) ----
template void ensure_poly_vectors<float>(fi_ctx*);
template void ensure_poly_vectors<double>(fi_ctx*);
template void estimate_poly_lambda<float>(RankSet&);
template void estimate_poly_lambda<double>(RankSet&);
template void cg_run_poly<float>(RankSet&, int, float);
template void cg_run_poly<double>(RankSet&, int, float);
template void cg_run_poly_sr<float>(RankSet&, int, float);
template void cg_run_poly_sr<double>(RankSet&, int, float);
template void cg_run_poly_or_jacobi<float>(RankSet&, int, float);
template void cg_run_poly_or_jacobi<double>(RankSet&, int, float);

}  // namespace fi
