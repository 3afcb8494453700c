// fi_multigrid.hip -- the coarse-to-fine start (src/sdf_field.cpp:272-288 generalised to several levels), the smoothers,
// the V-cycle and V-cycle preconditioned CG (fp64 CG around an fp32 V-cycle in mixed precision).
#include "fi_solver_internal.h"

namespace fi {

// Coarse-to-fine start (the reference's own remedy for large lattices: solve a coarser lattice, upscale, use
// as the guess -- src/sdf_field.cpp:272-288, README.md "My resolution is huge"): every coarser level is
// solved from the interpolated solution of the level below it, to a loose tolerance; x of `c` receives the
// interpolated guess.  All on the device.
template <typename T>
bool cascade_guess(RankSet& R, RankSet* wide)
{
	bool wrote_wide = false;
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
					for (fi_ctx* l : lc) { l->predictable_start = true; }  // (from the level below or the 48 steps above)
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
			if (cubic && k == 1 && wide && sizeof(T) == 4) {
				// (same lattice, same slab, same local geometry: the replica's index is the fp64 context's)
				launch_prolong_cubic<T, double>(L, lc[i]->x.as<T>(), (*wide)[i]->x.as<double>(), lf[i]->stream);
				wrote_wide = true;
			} else if (cubic) {
				launch_prolong_cubic<T, T>(L, lc[i]->x.as<T>(), lf[i]->x.as<T>(), lf[i]->stream);
			} else {
				launch_prolong<T>(L, lc[i]->x.as<T>(), lf[i]->x.as<T>(), 0, lf[i]->stream);
			}
		}
		FI_HIP_TRY(hipGetLastError());
	}
	return wrote_wide;
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


// FI_OPT_FIELD_TOLERANCE: a workgroup's max |x_k - x_(k-1)| and max |x_k| (the step kernels, for free in the pass that
// updates x) into the partial array behind the r.r partials -- [grid] + [grid] values; k_field_max reduces them into the
// scalars in front of k_mg_logic(kMgResid).  (First form: an atomicMax per wave on the two words -- 32 k atomics on one
// cache line per launch serialise at ~10 ns each: k_mg_step_mixed 144 -> 253 us.)
__device__ inline void field_maxes(double* __restrict__ field_part, double md, double mx)
{
	if (!field_part) { return; }
	__shared__ double s_md[kThreads / 64], s_mx[kThreads / 64];
	for (int o = 32; o > 0; o >>= 1) {
		md = fmax(md, __shfl_down(md, o, 64));
		mx = fmax(mx, __shfl_down(mx, o, 64));
	}
	if ((threadIdx.x & 63) == 0) {
		s_md[threadIdx.x >> 6] = md;
		s_mx[threadIdx.x >> 6] = mx;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < kThreads / 64; ++w) {
			md = fmax(md, s_md[w]);
			mx = fmax(mx, s_mx[w]);
		}
		field_part[blockIdx.x]             = md;
		field_part[gridDim.x + blockIdx.x] = mx;
	}
}
__global__ __launch_bounds__(kThreads) void k_field_max(CgScalars* sc, const double* __restrict__ field_part, int count)
{
	if (sc->done) { return; }
	double md = 0.0, mx = 0.0;
	for (int i = threadIdx.x; i < count; i += kThreads) {
		md = fmax(md, field_part[i]);
		mx = fmax(mx, field_part[count + i]);
	}
	__shared__ double s_md[kThreads / 64], s_mx[kThreads / 64];
	for (int o = 32; o > 0; o >>= 1) {
		md = fmax(md, __shfl_down(md, o, 64));
		mx = fmax(mx, __shfl_down(mx, o, 64));
	}
	if ((threadIdx.x & 63) == 0) {
		s_md[threadIdx.x >> 6] = md;
		s_mx[threadIdx.x >> 6] = mx;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < kThreads / 64; ++w) {
			md = fmax(md, s_md[w]);
			mx = fmax(mx, s_mx[w]);
		}
		sc->dmax_bits = static_cast<unsigned long long>(__double_as_longlong(md));
		sc->xmax_bits = static_cast<unsigned long long>(__double_as_longlong(mx));
	}
}

// The same over slabs: the slab's maxima into its two entries of CgScalars::rank_max, zeros in everybody else's (and in the
// sums the all-reduce carries along unused) -- summed over the slabs with the r.r partial sums in ONE collective.
__global__ __launch_bounds__(kThreads) void k_field_max_slot(CgScalars* sc, const double* __restrict__ field_part, int count, int slot,
                                                              int nslots)
{
	if (sc->done) { return; }
	double md = 0.0, mx = 0.0;
	for (int i = threadIdx.x; i < count; i += kThreads) {
		md = fmax(md, field_part[i]);
		mx = fmax(mx, field_part[count + i]);
	}
	__shared__ double s_md[kThreads / 64], s_mx[kThreads / 64];
	for (int o = 32; o > 0; o >>= 1) {
		md = fmax(md, __shfl_down(md, o, 64));
		mx = fmax(mx, __shfl_down(mx, o, 64));
	}
	if ((threadIdx.x & 63) == 0) {
		s_md[threadIdx.x >> 6] = md;
		s_mx[threadIdx.x >> 6] = mx;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int w = 1; w < kThreads / 64; ++w) {
			md = fmax(md, s_md[w]);
			mx = fmax(mx, s_mx[w]);
		}
		for (int i = 0; i < 2 * nslots; ++i) { sc->rank_max[i] = 0.0; }
		sc->rank_max[2 * slot]     = md;
		sc->rank_max[2 * slot + 1] = mx;
		sc->sums[1] = sc->sums[2] = sc->sums[3] = 0.0;
	}
}
// loop-back group: the members' maxima summed entry by entry like k_group_sum sums sums[] (every entry has one non-zero term)
__global__ void k_group_sum_field(CgScalars* const* sc, int nranks)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) { return; }
	for (int v = 0; v < 2 * nranks; ++v) {
		double s = 0;
		for (int r = 0; r < nranks; ++r) { s += sc[r]->rank_max[v]; }
		for (int r = 0; r < nranks; ++r) { sc[r]->rank_max[v] = s; }
	}
}

// CG with a preconditioner: r -= alpha q, x += alpha p (p is still the direction of this step), partial r.r
template <typename T>
__global__ __launch_bounds__(kThreads) void k_mg_step(int64_t n, const CgScalars* __restrict__ sc, const T* __restrict__ p,
                                                       const T* __restrict__ q, T* __restrict__ x, T* __restrict__ r,
                                                       double* __restrict__ partial, double* __restrict__ field_part)
{
	if (sc->done) { return; }
	const T alpha = static_cast<T>(sc->alpha);
	double acc[1] = {0};
	double mp = 0.0, mx = 0.0;  // FI_OPT_FIELD_TOLERANCE: max |p|, max |x_new|
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		const T pi = p[i];
		const T xi = x[i] + alpha * pi;
		x[i] = xi;
		mp = fmax(mp, fabs(static_cast<double>(pi)));
		mx = fmax(mx, fabs(static_cast<double>(xi)));
		const T ri = r[i] - alpha * q[i];
		r[i] = ri;
		acc[0] += static_cast<double>(ri) * static_cast<double>(ri);
	}
	field_maxes(field_part, fabs(static_cast<double>(alpha)) * mp, mx);
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

// Streaming accesses for vectors that pass through ONCE and do not fit the L2 (8 x 4 MB): nontemporal loads / stores keep them
// out of its working set -- the kernels themselves run at 6.3 instead of 5.7 TB/s and the stencil kernels behind them find
// their halo lines still cached (profiles/r5_ablation.md section 26).  Vectors of fewer than kStreamMin points stay cached:
// a 1024^2 lattice's fp64 vectors (8 MB each) are served from the L2 iteration after iteration.
constexpr int64_t kStreamMin = int64_t(1) << 22;
template <bool NT, typename U>
__device__ inline U ldv(const U* p)
{
	if constexpr (NT) { return __builtin_nontemporal_load(p); } else { return *p; }
}
template <bool NT, typename U>
__device__ inline void stv(U v, U* p)
{
	if constexpr (NT) { __builtin_nontemporal_store(v, p); } else { *p = v; }
}

// Mixed precision (CG in fp64, V-cycle on the fp32 replica): the fp32 copy of the residual leaves k_mg_step with the
// update itself, and the fp64 copy of z = V(r) is never formed -- r.z and the new direction read the fp32 result.
// Per step 3 fp64 lattice passes less than k_mg_step + k_to_twin + k_from_twin + k_dot + k_mg_direction.
template <bool NT>
__global__ __launch_bounds__(kThreads) void k_mg_step_mixed(int64_t n, const CgScalars* __restrict__ sc,
                                                             const double* __restrict__ p, const double* __restrict__ q,
                                                             double* __restrict__ x, double* __restrict__ r,
                                                             float* __restrict__ r32, double* __restrict__ partial,
                                                             double* __restrict__ field_part)
{
	if (sc->done) { return; }
	const double alpha = sc->alpha;
	const double inv = 1.0 / mixed_scale(sc);  // sc->rr is still the previous norm: k_mg_logic(kMgResid) records this scale
	double acc[1] = {0};
	// a workgroup walks ONE contiguous piece of the seven streams (512^3: 1.68 -> 1.48 ms against a grid-stride loop whose
	// workgroups touch 2 048 places of every stream at once; 256^3: the same 150 us)
	const int64_t piece = ((n + gridDim.x - 1) / gridDim.x + kThreads - 1) / kThreads * kThreads;
	const int64_t i0 = static_cast<int64_t>(blockIdx.x) * piece, i1 = i0 + piece < n ? i0 + piece : n;
	double mp = 0.0, mx = 0.0;  // FI_OPT_FIELD_TOLERANCE: max |p|, max |x_new|
	for (int64_t i = i0 + threadIdx.x; i < i1; i += kThreads) {
		const double pi = ldv<NT>(p + i);
		const double xi = ldv<NT>(x + i) + alpha * pi;
		stv<NT>(xi, x + i);
		mp = fmax(mp, fabs(pi));
		mx = fmax(mx, fabs(xi));
		const double ri = ldv<NT>(r + i) - alpha * ldv<NT>(q + i);
		stv<NT>(ri, r + i);
		stv<NT>(static_cast<float>(ri * inv), r32 + i);
		acc[0] += ri * ri;
	}
	field_maxes(field_part, fabs(alpha) * mp, mx);
	double out[1];
	block_sum<1>(acc, out);
	if (threadIdx.x == 0) { partial[blockIdx.x] = out[0]; }
}
// partial of r . (s z32)
template <bool NT>
__global__ __launch_bounds__(kThreads) void k_dot_mixed(int64_t n, const CgScalars* __restrict__ sc, const double* __restrict__ r,
                                                         const float* __restrict__ z32, double* __restrict__ partial)
{
	const double s = twin_scale(sc);
	double acc[1] = {0};
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		acc[0] += ldv<NT>(r + i) * (s * static_cast<double>(ldv<NT>(z32 + i)));
	}
	double out[1];
	block_sum<1>(acc, out);
	if (threadIdx.x == 0) { partial[blockIdx.x] = out[0]; }
}
// p = s z32 + beta p
template <bool NT>
__global__ __launch_bounds__(kThreads) void k_mg_direction_mixed(int64_t n, const CgScalars* __restrict__ sc,
                                                                  const float* __restrict__ z32, double* __restrict__ p, int first)
{
	const double beta = first ? 0.0 : sc->beta;
	const double s = twin_scale(sc);
	const int64_t piece = ((n + gridDim.x - 1) / gridDim.x + kThreads - 1) / kThreads * kThreads;  // (contiguous pieces: k_mg_step_mixed)
	const int64_t i0 = static_cast<int64_t>(blockIdx.x) * piece, i1 = i0 + piece < n ? i0 + piece : n;
	for (int64_t i = i0 + threadIdx.x; i < i1; i += kThreads) {
		stv<NT>(s * static_cast<double>(ldv<NT>(z32 + i)) + beta * ldv<NT>(p + i), p + i);
	}
}

// r = b - q with the partials of r.r and b.b in the same pass (the start and the verification of V-cycle PCG on an
// undivided lattice: k_sub + two k_dot + their one-block sums were five launches and three more lattice passes)
template <typename T, bool NT>
__global__ __launch_bounds__(kThreads) void k_resid_norms(int64_t n, const T* __restrict__ b, const T* __restrict__ q, T* __restrict__ r,
                                                           double* __restrict__ partial_rr, double* __restrict__ partial_bb)
{
	double acc[2] = {0, 0};
	const int64_t piece = ((n + gridDim.x - 1) / gridDim.x + kThreads - 1) / kThreads * kThreads;  // (contiguous pieces: k_mg_step_mixed)
	const int64_t i0 = static_cast<int64_t>(blockIdx.x) * piece, i1 = i0 + piece < n ? i0 + piece : n;
	for (int64_t i = i0 + threadIdx.x; i < i1; i += kThreads) {
		const T bi = ldv<NT>(b + i);
		const T ri = bi - ldv<NT>(q + i);
		stv<NT>(ri, r + i);
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

// The Chebyshev smoother in the full operator (every level that does not run the polynomial smoother: 2-D lattices, oriented
// points, fp64 levels): degree and interval [lambda / ratio, 1.1 lambda].  Round 2: degree 2 -> 4 halved the solve time of
// configs 3 / 5.  Round 6, swept under the field stop rule (tools/r6_sweep_sdf_smoother.sh, profiles/r6_sweep_smoother.txt):
// a wider interval pays -- rediscretised coarse levels correct an SDF's intermediate modes poorly, the smoother has to reach
// further down -- config 5 (4, 10) 588 ms / 56 iterations, (5, 40) 508 / 40, (6, 40) 513 / 36, (5, 80) 838 / 68 (past 40 the
// damping at the upper end gives out); config 3 (4, 10) 52.8 ms / 34, (4, 40) 49.1 / 29, (5, 40) 52.7 / 28; config 2 5.1 -> 4.8.
// Under the K-cycle (FI_OPT_MG_KCYCLE) the coarse correction is strong and the smoother need not reach down: config 5 with four
// K-levels (4, 10) 259 ms / 14 iterations, (5, 40) 347 / 16, (3, 20) 256 / 16, (6, 20) 252 / 10 -- but NOT on a shallow hierarchy (96 x 80 x 64
// with two levels: 123 iterations against the V-cycle's 96 with (5, 40)): FI_OPT_MG_CHEB_DEGREE / _RATIO, set by bench_settings for config 5.
int mg_degree(const fi_ctx* c)
{
	const char* e = tuning_switch("FI_MG_DEGREE");
	if (e && atoi(e) > 0) { return atoi(e); }
	return c->mg_cheb_degree > 0 ? c->mg_cheb_degree : (c->g.ndim == 3 ? 5 : 4);
}
// the coarsest level's solve: a longer polynomial over a wider band (FI_MG_COARSEST_STEPS / FI_MG_COARSEST_RATIO: timing builds)
int mg_coarsest_steps(const fi_ctx* c)
{
	const char* e = tuning_switch("FI_MG_COARSEST_STEPS");
	return e && atoi(e) > 1 ? atoi(e) : 4 * mg_degree(c) + 4;
}
double mg_ratio(const fi_ctx* c);
double mg_coarsest_ratio(const fi_ctx* c)
{
	const char* e = tuning_switch("FI_MG_COARSEST_RATIO");
	return e && atof(e) > 1 ? atof(e) : 10.0 * mg_ratio(c);
}
double mg_ratio(const fi_ctx* c)
{
	const char* e = tuning_switch("FI_MG_RATIO");
	if (e && atof(e) > 1) { return atof(e); }
	return c->mg_cheb_ratio > 1.0 ? c->mg_cheb_ratio : (c->g.ndim == 3 ? 40.0 : 10.0);
}

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

template <typename T>
__global__ __launch_bounds__(kThreads) void k_add_vec(int64_t n, const T* __restrict__ d, T* __restrict__ x)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		x[i] += d[i];
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
		// (FI_POLY_SMOOTHER_ANY_DATA, timing builds: the experiment of profiles/r6_ablation.md section 9 -- oriented points on
		// this smoother)
		const bool rows_ok = c->value_rows_only || tuning_switch("FI_POLY_SMOOTHER_ANY_DATA");
		if (c->mg_smoother != 1 || !rows_ok || !c->march.valid || !stencil_full_epi_available(c) || c->generic.ntrip != 0 || c->any_trip) {
			return false;
		}
	}
	return true;
}

// z = M r through the work vectors za / zb; returns the one that holds the result (ghost planes not exchanged).
// onto: the vector the result is ADDED to by the polynomial's last step itself (x += M r: the post-smoothing; returns onto) --
// one launch and three lattice passes less than a sum of its own; taken when the last step is not the one that forms its
// operand on load, else the result comes back in a work vector as without it.
template <typename T>
Vec poly_chain(RankSet& R, Vec r, Vec za, Vec zb, double* chain_bytes = nullptr, int* chain_launches = nullptr, Vec onto = nullptr,
               Vec dotv = nullptr)
{
	fi_ctx* c0 = R[0];
	const int    terms = mg_poly_terms(c0);
	if (onto && (terms < 3 || test_switch("FI_NO_STEP_ONTO"))) { onto = nullptr; }
	// dotv: the last step's partials are those of dotv . (onto after the sum) -- asked for by the top of a replica's cycle
	// (vcycle), delivered when that step runs in the marching kernel on one undivided context
	if (dotv && !(onto && R.size() == 1 && c0->nranks == 1 && c0->march.valid && !stencil_cheb_direct(c0))) { dotv = nullptr; }
	c0->bx_dot_done = dotv != nullptr;
	const double lam = c0->poly_lambda > 1.0 ? c0->poly_lambda : 1.0;
	const double hi = 1.1 * lam, lo = hi / mg_poly_ratio(c0);
	const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
	const bool single = R.size() == 1 && c0->nranks == 1;
	bool ghosts_scaled = true;
	for (const fi_ctx* c : R) { ghosts_scaled = ghosts_scaled && c->scaling_ghosts; }
	const bool pro = (single || ghosts_scaled) && terms > 2 && c0->march.valid && !test_switch("FI_NO_Z0_ON_LOAD");
	// (the V-cycle's smoother takes no dot products: a small level then runs its steps as direct launches, fi_stencil.hip)
	auto region2 = [](fi_ctx* c) { return stencil_cheb_direct(c) ? nullptr : c->partial.as<double>() + 2 * static_cast<size_t>(c->max_blocks); };
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
	// The iterates between the first and the last step live as bfloat16 on an undivided fp32 level of some size (no ghost
	// planes to exchange, whole 16-byte groups per row): they are the operands of a PRECONDITIONER, and a step then moves
	// 8 / 10 / 12 / 14 bytes per point instead of 10 / 14 / 18 / 18 (FI_NO_Z16: fp32 iterates, tests).
	const bool z16 = single && pro && sizeof(T) == 4 && c0->g.ndim == 3 && c0->g.gn[0] % 4 == 0 && c0->g.nloc >= (1 << 21) &&
	                 !test_switch("FI_NO_Z16");
	if (chain_bytes && z16) {
		*chain_bytes = 0;
		for (int j = 1; j < terms; ++j) {
			const bool last = j == terms - 1;
			const double in = j == 1 ? 4.0 : 2.0, prev = j <= 2 ? 0.0 : 2.0, rr = j == 1 ? 0.0 : 4.0, out = last ? 4.0 : 2.0;
			*chain_bytes += (in + prev + rr + out + 2.0) * static_cast<double>(c0->g.nown);
		}
	}
	if (z16) {
		// Where the iterates live: z_j (j = 1 .. n - 1, bfloat16) in one of the four half-buffers of za / zb, chosen so that
		// a step never writes where its own operands are -- a step that reads bfloat16 and writes fp32 (the last one) or the
		// other way round is NOT in place -- and the last step finds zb free: z_(n-1), z_(n-2) in the halves of za, z_(n-3),
		// z_(n-4) in those of zb, and so on (a half is reused four steps later, two after its last reader).
		fi_ctx* c = c0;
		const int n = terms - 1;
		auto where = [&](int j) -> void* {
			const int d = n - 1 - j;
			char* base = static_cast<char*>((d & 2) ? (c->*zb).p : (c->*za).p);
			return base + ((d & 1) ? 2 * static_cast<size_t>(c->g.nloc) : 0);
		};
		const unsigned short* sc = c->dinv16s.as<unsigned short>();
		double rho = 1.0 / sigma;
		for (int k = 1; k <= n; ++k) {
			const double rho_new = 1.0 / (2.0 * sigma - rho);
			const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
			void* out = k == n ? (onto ? (c->*onto).p : (c->*zb).p) : where(k);
			const int fmt = (k > 1 ? 1 : 0) | (k > 2 ? 2 : 0) | (k < n ? 4 : 0);
			if (k == 1) {
				stencil_cheb_step(c, (c->*r).p, nullptr, (c->*r).p, out, c1, c2, region2(c), 0, 0.0, 1.0 / theta, sc, 0, fmt);
			} else {
				stencil_cheb_step(c, where(k - 1), k == 2 ? (c->*r).p : where(k - 2), (c->*r).p, out, c1, c2, region2(c), 0,
				                  k == 2 ? 1.0 / theta : 0.0, 0.0, sc, 0, fmt, k == n && onto ? (c->*onto).p : nullptr,
				                  k == n && dotv ? (c->*dotv).p : nullptr);
			}
			rho = rho_new;
		}
		return onto ? onto : zb;
	}
	// Deep exchange over slabs (the polynomial PCG's, fi_poly.hip, for the V-cycle's smoother): fi_assemble has given the
	// vectors 2 (d - 1) ghost planes; r's travel ONCE per polynomial, step k then also computes its 2 (d - 1 - k) nearest ghost
	// planes -- what the neighbour computes for its own planes, bit for bit -- and no step waits for an exchange: one
	// exchange per polynomial instead of d - 1 (a level's cycle: 6 instead of 12).  Levels whose slabs are thinner than the
	// deep width keep one exchange per step; FI_NO_DEEP_HALO: every level does (tests: the same bits).
	const int  deep_width = 2 * (terms - 1);
	const bool deep = pro && c0->nranks > 1 && c0->halo >= deep_width && c0->min_slab >= deep_width && !test_switch("FI_NO_DEEP_HALO");
	Vec zin = za, zout = zb;
	double rho = 1.0 / sigma;
	for (int k = 1; k < terms; ++k) {
		const double rho_new = 1.0 / (2.0 * sigma - rho);
		const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
		const bool   first_on_load = pro && k == 1;
		const int    ext = deep ? 2 * (terms - 1 - k) : 0;  // ghost planes this step computes for the next one
		if (deep) {
			if (k == 1) { halo_exchange(R, r, deep_width); }
		} else {
			halo_exchange(R, first_on_load ? r : zin);
		}
		for (fi_ctx* c : R) {
			const unsigned short* sc = c->dinv16s.as<unsigned short>();
			if (first_on_load) {
				stencil_cheb_step(c, (c->*r).p, nullptr, (c->*r).p, (c->*zout).p, c1, c2, region2(c), 0, 0.0, 1.0 / theta, sc, ext);
			} else {
				// the second step's z_prev is z_0 = Dinv r / theta, recomputed from r and the scaling
				const bool last_onto = onto && k == terms - 1;  // (z_prev is read from zout, the result goes onto `onto`)
				stencil_cheb_step(c, (c->*zin).p, k == 1 ? nullptr : (c->*zout).p, (c->*r).p, last_onto ? (c->*onto).p : (c->*zout).p, c1, c2,
				                  region2(c), 0, k == 2 ? 1.0 / theta : 0.0, 0.0, sc, ext, 0, last_onto ? (c->*onto).p : nullptr,
				                  last_onto && dotv ? (c->*dotv).p : nullptr);
			}
		}
		std::swap(zin, zout);
		rho = rho_new;
	}
	return onto ? onto : zin;
}

// ---- the V-cycle of the hierarchy's tail as a program of the small-level engine (fi_tail.h) ----------------------------
// The same cycle as vcycle() below -- the same smoothers with the same constants, the same transfers -- written out as the
// stages of ONE launch of one workgroup.  Every tail level has five vectors in LDS (fi_tail.h): B (right-hand side), X
// (result), R (residual), W0, W1 (the polynomials' iterates); the top level's B is filled from the caller's vector when the
// kernel starts, its X goes back to the caller's when it ends -- the program holds no pointer to a vector, so it stays valid
// from cycle to cycle and from solve to solve (until the levels are re-assembled or a smoother's interval is widened).
template <typename T>
bool poly_smoother_ok(const RankSet& R);

struct TailProgram {
	enum { B = 0, X = 1, Rr = 2, W0 = 3, W1 = 4 };
	std::vector<TailOp> ops;
	std::vector<fi_ctx*> chain;
	std::vector<int> base, nn, vstride, guard;
	bool ok = true;

	int vec(int l, int v) const { return base[l] + guard[l] + v * vstride[l]; }  // point 0 of vector v (fi_tail.h: TailLevel)
	void op(int kind, int level, int a, int b, int c, int out, int acc, const unsigned short* scale, double s0 = 0, double s1 = 0, double s2 = 0)
	{
		ops.push_back(TailOp{kind, level, a, b, c, out, acc, 0, scale, static_cast<float>(s0), static_cast<float>(s1), static_cast<float>(s2), 0});
	}
	// target (+)= M r: the polynomial in A_model + f diag(A_data) (poly_chain); r: vector id of the level; out: target of the last
	// step (-1: none), acc: accumulate into (-1: none)
	void poly_ops(int l, int r, int out, int acc)
	{
		fi_ctx* c = chain[l];
		const int    terms = mg_poly_terms(c);
		const double lam = c->poly_lambda > 1.0 ? c->poly_lambda : 1.0;
		const double hi = 1.1 * lam, lo = hi / mg_poly_ratio(c);
		const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
		const unsigned short* sc = c->dinv16s.as<unsigned short>();
		const int W[2] = {vec(l, W0), vec(l, W1)};
		if (terms < 2 || !c->dinv16s_valid) { ok = false; return; }
		op(kTailScale, l, vec(l, r), -1, -1, W[0], -1, sc, 1.0 / theta);
		int cur = 0;
		double rho = 1.0 / sigma;
		for (int k = 1; k < terms; ++k) {
			const double rho_new = 1.0 / (2.0 * sigma - rho);
			const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
			const bool   last = k == terms - 1;
			op(kTailPolyStep, l, W[cur], k == 1 ? -1 : W[1 - cur], vec(l, r), last ? (out >= 0 ? vec(l, out) : -1) : W[1 - cur],
			   last && acc >= 0 ? vec(l, acc) : -1, sc, 1.0 + c1, k == 1 ? 0.0 : c1, c2);
			cur = 1 - cur;
			rho = rho_new;
		}
	}
	// degree-k Chebyshev smoothing in the full operator (cheb_smooth_fused): X starts at zero / at its present value
	void cheb_ops(int l, int degree, double ratio, bool from_zero)
	{
		fi_ctx* c = chain[l];
		const double hi = 1.1 * c->lambda_max, lo = hi / ratio;
		const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
		const unsigned short* sc = c->dinv16.as<unsigned short>();
		const int b = vec(l, B), x = vec(l, X);
		const int W[3] = {vec(l, W0), vec(l, W1), vec(l, Rr)};   // (the residual vector is free while a smoother runs)
		if (degree < 2 || !(c->lambda_max > 0)) { ok = false; return; }
		int zk = -1;    // x_k
		int zp = -1;    // x_{k-1} (-1: zero, or no such term)
		const int steps = degree - 1;  // recurrence steps after the first term
		if (from_zero) {
			op(kTailScale, l, b, -1, -1, W[0], -1, sc, 1.0 / theta);   // x_1 = Dinv b / theta
			zk = W[0];
		} else {
			op(kTailChebStep, l, x, -1, b, W[0], -1, sc, 1.0, 0.0, 1.0 / theta);  // x_1 = x_0 + Dinv (b - A x_0) / theta
			zk = W[0];
			zp = x;
		}
		double rho = 1.0 / sigma;
		for (int k = 1; k <= steps; ++k) {
			const double rho_new = 1.0 / (2.0 * sigma - rho);
			const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
			const bool   last = k == steps;
			// a stage reads its input's NEIGHBOURS: the output is a vector that is neither x_k nor (read at the point only, but
			// by other threads' stages not at all) x_{k-1}'s: three work vectors rotate, the last step lands in X
			int out = x;
			if (!last || zk == x) {
				for (int w = 0; w < 3; ++w) {
					if (W[w] != zk && W[w] != zp) { out = W[w]; break; }
				}
			}
			op(kTailChebStep, l, zk, zp, b, out, -1, sc, 1.0 + c1, zp >= 0 ? c1 : 0.0, c2);
			zp = zk;
			zk = out;
			rho = rho_new;
		}
		if (zk != x) { ok = false; }  // (cannot happen: the last step writes X unless X is its own input, and it never is)
	}
	void residual(int l) { op(kTailResidual, l, vec(l, X), -1, vec(l, B), vec(l, Rr), -1, nullptr); }

	void cycle(int l)
	{
		fi_ctx* c = chain[l];
		RankSet one{c};
		const bool last = l + 1 == static_cast<int>(chain.size());
		const bool poly = poly_smoother_ok<float>(one);
		const int    deg = mg_degree(c);
		const double ratio = mg_ratio(c);
		if (c->lumped) { ok = false; return; }
		if (last) {
			if (poly && c->dinv16s_valid && c->data_pinned && !test_switch("FI_MG_COARSEST_CHEB")) {
				poly_ops(l, B, X, -1);
				residual(l);
				poly_ops(l, Rr, -1, X);
			} else {
				cheb_ops(l, mg_coarsest_steps(c), mg_coarsest_ratio(c), true);
			}
			return;
		}
		if (poly) {
			poly_ops(l, B, X, -1);
		} else {
			cheb_ops(l, deg, ratio, true);
		}
		residual(l);
		op(kTailRestrict, l, vec(l, Rr), -1, -1, vec(l + 1, B), -1, nullptr);
		cycle(l + 1);
		op(kTailProlongAdd, l, vec(l + 1, X), -1, -1, vec(l, X), -1, nullptr);
		if (poly) {
			residual(l);
			poly_ops(l, Rr, -1, X);
		} else {
			cheb_ops(l, deg, ratio, false);
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
	if (!c->tail_prog_valid) {
		TailProgram P;
		int floats = 0;
		for (fi_ctx* l = c; l; l = l->coarse) {
			const int nn = static_cast<int>(l->g.nloc);
			const int gd = 2 * (l->g.ndim > 2 ? l->g.gn[0] * l->g.gn[1] : l->g.gn[0]);
			P.chain.push_back(l);
			P.base.push_back(floats);
			P.nn.push_back(nn);
			P.guard.push_back(gd);
			P.vstride.push_back(nn + 2 * gd);
			floats += tail_level_floats(l->g.ndim, l->g.gn);
		}
		P.cycle(0);
		if (!P.ok || P.ops.empty() || static_cast<int>(P.chain.size()) > kTailMaxLevels || floats * sizeof(float) > 160u * 1024u) {
			c->tail_ok = false;  // (a setting the engine does not cover: the tiled kernels run this hierarchy)
			return false;
		}
		std::vector<unsigned char> blob(sizeof(TailLevel) * kTailMaxLevels + sizeof(TailOp) * P.ops.size());
		TailLevel* lv = reinterpret_cast<TailLevel*>(blob.data());
		for (size_t k = 0; k < P.chain.size(); ++k) {
			lv[k] = tail_level_of(P.chain[k]);
			lv[k].base    = P.base[k] + P.guard[k];
			lv[k].vstride = P.vstride[k];
			lv[k].guard   = P.guard[k];
			lv[k].ctab    = P.base[k] + kTailVectors * P.vstride[k];
			lv[k].ktab    = lv[k].ctab + P.nn[k];
			if (k + 1 < P.chain.size()) { lv[k].to_coarse = level_pair(P.chain[k], P.chain[k + 1]); }
		}
		std::memcpy(blob.data() + sizeof(TailLevel) * kTailMaxLevels, P.ops.data(), sizeof(TailOp) * P.ops.size());
		c->tail_prog.alloc(blob.size());
		FI_HIP_TRY(hipMemcpyAsync(c->tail_prog.p, blob.data(), blob.size(), hipMemcpyHostToDevice, c->stream));
		FI_HIP_TRY(hipStreamSynchronize(c->stream));  // (the host buffer dies here)
		c->tail_nlev       = static_cast<int>(P.chain.size());
		c->tail_nops       = static_cast<int>(P.ops.size());
		c->tail_lds_floats = floats;
		c->tail_prog_valid = true;
	}
	tail_run(c, c->tail_prog.p, c->tail_nlev, c->tail_nops, c->tail_lds_floats, (c->*b).as<float>(), (c->*x).as<float>());
	return true;
}

template <typename T>
void vcycle(RankSet& R, Vec b, Vec x);
template <typename T>
__global__ __launch_bounds__(kThreads) void k_add_scaled(int64_t n, const T* __restrict__ d, T* __restrict__ x, T w)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kThreads) {
		x[i] += w * d[i];
	}
}

// ---- K-cycle (FI_OPT_MG_KCYCLE) ----------------------------------------------------------------------------------------
// The correction of a coarse level by two steps of flexible CG on ITS system A e = b, each preconditioned by the level's own
// cycle K(.):  c1 = K(b), then c2 = K(b - s1 A c1), and e = the combination of c1 and c2 that minimises the energy norm of the
// error over their span.  Everything is written with RESIDUALS w = rhs - A c (the launch a cycle makes anyway, ChebEpi mode
// 3) instead of products: v1 = A c1 = b - w1, v2 = A c2 = r1 - w2.  Five vectors of the level beside the cycle's own:
// c1 = mg_x, w1 = r, r1 = x, c2 = p, w2 = q (the level's CG vectors: idle inside a cycle).
struct KcScalars {
	double a1, rho1;      // c1 . b, c1 . A c1
	double s1;            // a1 / rho1
	double coef1, coef2;  // e = coef1 c1 + coef2 c2
};
// step 1: partials of c1 . b and c1 . w1 -> s1;  step 2: of c2 . b, c2 . w1, c2 . r1, c2 . w2 -> coef1, coef2
__global__ __launch_bounds__(kThreads) void k_kc_coef(KcScalars* kc, const double* __restrict__ partial, int stride, int count, int step)
{
	double acc[4] = {0, 0, 0, 0};
	const int nv = step == 1 ? 2 : 4;
	for (int v = 0; v < nv; ++v) {
		for (int i = threadIdx.x; i < count; i += kThreads) { acc[v] += partial[static_cast<size_t>(v) * stride + i]; }
	}
	double out[4];
	block_sum<4>(acc, out);
	if (threadIdx.x != 0) { return; }
	if (step == 1) {
		kc->a1   = out[0];
		kc->rho1 = out[0] - out[1];  // c1 . (b - w1)
		const bool ok = kc->rho1 > 0.0 && isfinite(kc->rho1) && isfinite(kc->a1);
		kc->s1    = ok ? kc->a1 / kc->rho1 : 1.0;  // (a cycle that is not positive on b: its plain result, like a V-cycle)
		kc->coef1 = kc->s1;
		kc->coef2 = 0.0;
		return;
	}
	// r1 = b - s1 v1 = (1 - s1) b + s1 w1
	const double s1 = kc->s1, rho1 = kc->rho1;
	const double gam  = out[0] - out[1];   // c2 . v1
	const double beta = out[2] - out[3];   // c2 . v2 = c2 . (r1 - w2)
	const double a2   = out[2];            // c2 . r1
	const double rho2 = beta - (rho1 > 0.0 ? gam * gam / rho1 : 0.0);
	if (rho1 > 0.0 && rho2 > 1e-30 * fabs(beta) && isfinite(rho2) && isfinite(a2) && isfinite(gam)) {
		kc->coef1 = s1 - gam * a2 / (rho1 * rho2);
		kc->coef2 = a2 / rho2;
	} else {  // the second direction adds nothing (or the numbers are not usable): the first step stands
		kc->coef1 = s1;
		kc->coef2 = 0.0;
	}
}
// the dot products of one step in ONE pass over the vectors: a . b0, a . b1 (step 1), ... a . b3 (step 2) -> partial[v * stride + block]
template <typename T, int NV>
__global__ __launch_bounds__(kThreads) void k_kc_dots(int64_t n, const T* __restrict__ a, const T* __restrict__ b0, const T* __restrict__ b1,
                                                       const T* __restrict__ b2, const T* __restrict__ b3, double* __restrict__ partial, int stride)
{
	double acc[NV];
#pragma unroll
	for (int v = 0; v < NV; ++v) { acc[v] = 0.0; }
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kThreads) {
		const double av = static_cast<double>(a[i]);
		acc[0] += av * static_cast<double>(b0[i]);
		acc[1] += av * static_cast<double>(b1[i]);
		if (NV > 2) {
			acc[2] += av * static_cast<double>(b2[i]);
			acc[3] += av * static_cast<double>(b3[i]);
		}
	}
	double out[NV];
	block_sum<NV>(acc, out);
	if (threadIdx.x == 0) {
#pragma unroll
		for (int v = 0; v < NV; ++v) { partial[static_cast<size_t>(v) * stride + blockIdx.x] = out[v]; }
	}
}
template <typename T>
__global__ __launch_bounds__(kThreads) void k_kc_r1(int64_t n, const KcScalars* __restrict__ kc, const T* __restrict__ b, const T* __restrict__ w1,
                                                     T* __restrict__ r1)
{
	const T s1 = static_cast<T>(kc->s1);
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kThreads) {
		r1[i] = (T(1) - s1) * b[i] + s1 * w1[i];
	}
}
template <typename T>
__global__ __launch_bounds__(kThreads) void k_kc_combine(int64_t n, const KcScalars* __restrict__ kc, T* __restrict__ c1, const T* __restrict__ c2)
{
	const T a = static_cast<T>(kc->coef1), b = static_cast<T>(kc->coef2);
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n; i += static_cast<int64_t>(gridDim.x) * kThreads) {
		c1[i] = a * c1[i] + b * c2[i];
	}
}
bool kcycle_level(const RankSet& Rc)
{
	const fi_ctx* c = Rc[0];
	return Rc.size() == 1 && c->nranks == 1 && c->mg_kcycle > 0 && c->level >= 1 && c->level <= c->mg_kcycle && c->coarse != nullptr && !c->lumped &&
	       !c->tail_ok && !c->replicated && smooth_fused_ok(Rc) && !test_switch("FI_NO_KCYCLE");
}
template <typename T>
void kcycle_correction(RankSet& Rc)
{
	fi_ctx* c = Rc[0];
	ensure_vectors(c);
	c->kc.alloc(sizeof(KcScalars));
	KcScalars* kc = c->kc.as<KcScalars>();
	const int64_t n = c->g.nown;
	const int nb = stream_blocks(n);
	const int stride = c->max_blocks;
	double* part = c->partial.as<double>();
	const Vec B = &fi_ctx::mg_b, C1 = &fi_ctx::mg_x, W1 = &fi_ctx::r, R1 = &fi_ctx::x, C2 = &fi_ctx::p, W2 = &fi_ctx::q;
	vcycle<T>(Rc, B, C1);
	stencil_full_step(c, (c->*C1).p, nullptr, (c->*B).p, true, (c->*W1).p, 0.0, 0.0, 0.0);   // w1 = b - A c1
	hipLaunchKernelGGL((k_kc_dots<T, 2>), dim3(nb), dim3(kThreads), 0, c->stream, n, vown<T>(c, C1), vown<T>(c, B), vown<T>(c, W1),
	                   static_cast<const T*>(nullptr), static_cast<const T*>(nullptr), part, stride);
	hipLaunchKernelGGL(k_kc_coef, dim3(1), dim3(kThreads), 0, c->stream, kc, part, stride, nb, 1);
	if (const char* e = tuning_switch("FI_KC_STEPS")) {  // (timing builds: ONE step -- the cycle's result scaled by its line search)
		if (atoi(e) == 1) {
			hipLaunchKernelGGL((k_kc_combine<T>), dim3(nb), dim3(kThreads), 0, c->stream, n, kc, vown<T>(c, C1), vown<T>(c, C1));
			return;
		}
	}
	hipLaunchKernelGGL((k_kc_r1<T>), dim3(nb), dim3(kThreads), 0, c->stream, n, kc, vown<T>(c, B), vown<T>(c, W1), vown<T>(c, R1));
	vcycle<T>(Rc, R1, C2);
	stencil_full_step(c, (c->*C2).p, nullptr, (c->*R1).p, true, (c->*W2).p, 0.0, 0.0, 0.0);  // w2 = r1 - A c2
	hipLaunchKernelGGL((k_kc_dots<T, 4>), dim3(nb), dim3(kThreads), 0, c->stream, n, vown<T>(c, C2), vown<T>(c, B), vown<T>(c, W1), vown<T>(c, R1),
	                   vown<T>(c, W2), part, stride);
	hipLaunchKernelGGL(k_kc_coef, dim3(1), dim3(kThreads), 0, c->stream, kc, part, stride, nb, 2);
	hipLaunchKernelGGL((k_kc_combine<T>), dim3(nb), dim3(kThreads), 0, c->stream, n, kc, vown<T>(c, C1), vown<T>(c, C2));
	FI_HIP_TRY(hipGetLastError());
}

// The coarse correction of a level: mg_x = V(mg_b) on the coarser level Rc -- and, in timing builds with FI_MG_GAMMA=2 (the
// experiment of profiles/r6_ablation.md section 11: a W-cycle), once more on what that left: mg_x += omega V(mg_b - A mg_x), through
// the level's CG vectors r and p (idle inside a cycle); FI_MG_GAMMA_FROM: the first level visited twice; FI_MG_OMEGA: the damping.
// Levels the small-level engine runs in one launch keep their V-cycle.  NOT shipped: it cuts the iterations of oriented-point
// problems by a third to two thirds where it works and is INDEFINITE on config 5's hierarchy (the rediscretised coarse levels
// overcorrect some modes more than threefold -- harmless under CG, fatal once a level iterates on its own correction).
template <typename T>
void coarse_correction(RankSet& Rc)
{
	if (kcycle_level(Rc)) {
		kcycle_correction<T>(Rc);
		return;
	}
	vcycle<T>(Rc, &fi_ctx::mg_b, &fi_ctx::mg_x);
	const char* g = tuning_switch("FI_MG_GAMMA");
	const char* from = tuning_switch("FI_MG_GAMMA_FROM");  // the first level (1 = the one below the finest) that is visited twice
	if (!(g && atoi(g) == 2) || !Rc[0]->coarse || !smooth_fused_ok(Rc) || replicated_copies(Rc)) { return; }
	if (from && Rc[0]->level < atoi(from)) { return; }
	if (const char* to = tuning_switch("FI_MG_GAMMA_TO")) {  // ... and the last one
		if (Rc[0]->level > atoi(to)) { return; }
	}
	for (fi_ctx* c : Rc) {
		if (c->lumped || c->tail_ok) { return; }
	}
	for (fi_ctx* c : Rc) { ensure_vectors(c); }
	halo_exchange(Rc, &fi_ctx::mg_x);
	for (fi_ctx* c : Rc) { stencil_full_step(c, c->mg_x.p, nullptr, c->mg_b.p, true, c->r.p, 0.0, 0.0, 0.0); }  // r = mg_b - A mg_x
	vcycle<T>(Rc, &fi_ctx::r, &fi_ctx::p);
	const char* om = tuning_switch("FI_MG_OMEGA");  // damping of the second visit's correction (1: the plain W-cycle)
	const T omega = om ? static_cast<T>(atof(om)) : T(1);
	for (fi_ctx* c : Rc) {
		hipLaunchKernelGGL((k_add_scaled<T>), dim3(stream_blocks(c->g.nown)), dim3(kThreads), 0, c->stream, c->g.nown, vown<T>(c, &fi_ctx::p),
		                   vown<T>(c, &fi_ctx::mg_x), omega);
	}
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
	const int deg = mg_degree(R[0]);
	const double ratio = mg_ratio(R[0]);
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
		// (the top of a replica's cycle: b . x, the fp64 CG's r . z, leaves the last step as its partials when asked for)
		const bool want_bx = R[0]->level == 0 && R[0]->bx_dot_wanted && R[0]->coarse != nullptr;
		const Vec d = poly_chain<T>(R, &fi_ctx::mg_r, &fi_ctx::mg_d, &fi_ctx::q, nullptr, nullptr, x, want_bx ? b : nullptr);
		if (d == x) { return; }  // (the last step has added its result onto x)
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
		cheb_smooth<T>(R, b, x, mg_coarsest_steps(R[0]), mg_coarsest_ratio(R[0]), true);
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
		coarse_correction<T>(Rc);
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
	coarse_correction<T>(Rc);
	halo_exchange(Rc, &fi_ctx::mg_x);
	for (size_t i = 0; i < R.size(); ++i) {
		const LevelPair L = level_pair(R[i], Rc[i]);
		launch_prolong<T>(L, vbase<T>(Rc[i], &fi_ctx::mg_x), vbase<T>(R[i], x), 1, R[i]->stream);
	}
	cheb_smooth<T>(R, b, x, deg, ratio, false);
}

struct ScalarRecords {
	CgScalars* p[16];
	int        n = 0;
};
__global__ __launch_bounds__(64) void k_clear_scalars(ScalarRecords recs)
{
	uint32_t* w = reinterpret_cast<uint32_t*>(recs.p[blockIdx.x]);
	for (unsigned int i = threadIdx.x; i < sizeof(CgScalars) / 4; i += 64) { w[i] = 0u; }
}

// work vectors, cleared stop flags and smoother bounds for every level below (and including) R
template <typename T>
void mg_prepare(RankSet& R, bool clear_finest)
{
	RankSet lev = R;
	bool finest = true;
	ScalarRecords recs;
	hipStream_t recs_stream = nullptr;
	auto flush = [&]() {  // one launch for the records of up to 16 levels on one stream (was: a fill per level)
		if (recs.n > 0) {
			hipLaunchKernelGGL(k_clear_scalars, dim3(recs.n), dim3(64), 0, recs_stream, recs);
			FI_HIP_TRY(hipGetLastError());
		}
		recs.n = 0;
	};
	while (!lev.empty() && lev[0]) {
		for (fi_ctx* l : lev) {
			mg_alloc<T>(l);
			if (!finest || clear_finest) {  // the operator kernels of a level exit early while ITS stop flag is up
				if (recs.n == 16 || (recs.n > 0 && l->stream != recs_stream)) { flush(); }
				recs_stream = l->stream;
				recs.p[recs.n++] = l->scal.as<CgScalars>();
			}
		}
		finest = false;
		if (!lev[0]->coarse) { break; }
		lev = coarse_of(lev);
	}
	flush();
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

template <bool NT>
__global__ __launch_bounds__(kThreads) void k_to_twin(int64_t n, const CgScalars* __restrict__ sc, const double* __restrict__ r,
                                                       float* __restrict__ r32)
{
	const double inv = 1.0 / twin_scale(sc);
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < n;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		stv<NT>(static_cast<float>(ldv<NT>(r + i) * inv), r32 + i);
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
			hipLaunchKernelGGL(c->g.nown >= kStreamMin ? k_to_twin<true> : k_to_twin<false>, dim3(stream_blocks(c->g.nown)), dim3(kThreads), 0, c->stream, c->g.nown,
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
	for (fi_ctx* c : R) {  // (ghost planes outside the lattice; an undivided lattice is written whole below)
		if (c->g.nown != c->g.nloc) { FI_HIP_TRY(hipMemsetAsync(c->x.p, 0, sizeof(double) * c->g.nloc, c->stream)); }
	}
	const bool widened = cascade_guess<float>(Tw, &R);
	for (fi_ctx* c : R) {
		fi_ctx* t = c->twin;
		if (!widened) {
			hipLaunchKernelGGL(k_widen, dim3(stream_blocks(c->g.nown)), dim3(kThreads), 0, c->stream, c->g.nown,
			                   t->x.as<float>() + t->g.own_first, c->x.as<double>() + c->g.own_first);
		}
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
	// FI_OPT_FIELD_TOLERANCE (an undivided lattice): the solve stops by the field (k_mg_logic, kMgResid); the residual rule
	// stays as the floor of what the precision's recurrence can still tell apart
	// Over slabs (a loop-back group's members or one slab per process, at most kFieldRanks of them) every slab's maxima
	// travel with the r.r sum (CgScalars::rank_max): no collective of their own, and every rank decides on the same numbers.
	const int  field_slabs = R.size() > 1 ? static_cast<int>(R.size()) : (c0->nranks > 1 ? c0->nranks : 0);
	const bool by_field = c0->field_tol > 0 && field_slabs <= kFieldRanks && !replicated_copies(R);
	const double tolerance = by_field ? (sizeof(T) == 8 ? 1e-13 : 2e-7)
	                                  : (tol > 0 ? static_cast<double>(tol) : static_cast<double>(std::numeric_limits<float>::epsilon()));
	EventPair timer;  // (destroyed on every way out: a coarse level's breakdown, a timeout)
	const hipEvent_t e0 = timer.e0, e1 = timer.e1;
	FI_HIP_TRY(hipEventRecord(e0, st));
	CgScalars init{};
	init.tol2      = tolerance * tolerance;
	init.max_iter  = max_iterations;
	init.field_tol = by_field ? c0->field_tol : 0.0;
	init.field_est = -1.0;
	init.field_ranks = by_field ? field_slabs : 0;
	init.field_min_iter = c0->predictable_start ? kFieldMinIter : kFieldMinIterGuess;
	reset_scalars(R, init);
	CgScalars* sc0 = c0->scal.as<CgScalars>();
	auto nbv      = [](fi_ctx* c) { return stream_blocks(c->g.nown); };
	auto nb_apply = [](fi_ctx* c) { return apply_num_partials(c); };
	// (the step kernels' workgroup maxima: behind the r.r partials -- the array holds 4 x max_blocks doubles)
	auto field_part = [&](fi_ctx* c) -> double* { return by_field ? c->partial.as<double>() + static_cast<size_t>(c->max_blocks) : nullptr; };
	const Vec X = &fi_ctx::x, Rv = &fi_ctx::r, P = &fi_ctx::p, Q = &fi_ctx::q, Z = &fi_ctx::mg_x, B = &fi_ctx::atb;
	while (static_cast<int>(c0->ev.size()) < 2 * kMaxSamples) {
		hipEvent_t e;
		FI_HIP_TRY(hipEventCreate(&e));
		c0->ev.push_back(e);
	}
	int samples = 0;
	const bool mixed = !Tw.empty();
	// FI_OPT_MG_KCYCLE: the cycle then depends on its argument and CG takes the flexible beta (one undivided context)
	const bool flexible = (mixed ? Tw[0] : c0)->mg_kcycle > 0 && (mixed ? Tw[0] : c0)->coarse != nullptr && R.size() == 1 && c0->nranks == 1 &&
	                      !test_switch("FI_NO_KCYCLE");
	fi_ctx* const prec_ctx = mixed ? Tw[0] : c0;  // the context whose finest-level smoother chains are timed (vcycle)
	if (prec_ctx->level == 0 && !(mixed ? replicated_copies(Tw) : replicated_copies(R))) {
		while (static_cast<int>(prec_ctx->ev_prec.size()) < 2 * kPolySamples) {
			hipEvent_t e;
			FI_HIP_TRY(hipEventCreate(&e));
			prec_ctx->ev_prec.push_back(e);
		}
		prec_ctx->prec_budget = 1;  // (one chain per solve: every record is a marker the stream stops at, ~6 us each side)
		prec_ctx->prec_taken  = 0;
	}
	auto dot = [&](Vec a, Vec b) {
		for (fi_ctx* c : R) {
			hipLaunchKernelGGL((k_dot<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, vown<T>(c, a), vown<T>(c, b),
			                   c->partial.as<double>());
		}
	};
	// mixed precision on one undivided context: r . z = t^2 (b . x of the replica's cycle), summed by the cycle's last launch
	// (ChebEpi::dotv) where that launch is the marching kernel's -- no pass of its own over r and z (k_dot_mixed)
	if (mixed) {
		for (fi_ctx* t : Tw) { t->bx_dot_wanted = R.size() == 1 && c0->nranks == 1 && !test_switch("FI_NO_TWIN_DOT"); }
	}
	auto twin_dot = [&]() { return mixed && Tw[0]->bx_dot_wanted && Tw[0]->bx_dot_done; };
	auto reduce_rz = [&](int phase) {
		if (twin_dot()) {
			fi_ctx* t = Tw[0];
			hipLaunchKernelGGL(k_mg_logic, dim3(1), dim3(kThreads), 0, c0->stream, sc0,
			                   t->partial.as<double>() + 2 * static_cast<size_t>(t->max_blocks), stencil_cheb_partials(t), phase, 1);
			return;
		}
		mg_reduce(R, nbv, phase);
	};
	auto dot_rz = [&]() {  // partials of r . z
		if (twin_dot()) { return; }
		if constexpr (std::is_same<T, double>::value) {
			if (mixed) {
				for (size_t i = 0; i < R.size(); ++i) {
					fi_ctx* c = R[i];
					hipLaunchKernelGGL(c->g.nown >= kStreamMin ? k_dot_mixed<true> : k_dot_mixed<false>, dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, c->scal.as<CgScalars>(),
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
					hipLaunchKernelGGL(c->g.nown >= kStreamMin ? k_mg_direction_mixed<true> : k_mg_direction_mixed<false>, dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown,
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
	// A solve that starts like the previous one of this context (coarse-to-fine or zero start, the same tolerance) most
	// likely takes as many iterations: no look at the stop flag before that count -- a look is a copy and a host round
	// trip of 25-40 us in front of every V-cycle.  Should the solve be over earlier, the fp64 kernels exit on the flag and
	// the cycles in between are wasted, nothing else; from the predicted count on every iteration looks again.
	if (c0->last_mg_iterations == 0 && R.size() == 1 && c0->predictable_start) {  // a fresh context: what the one before it learnt
		const int n = recall_iterations(c0, 1, tolerance);
		if (n > 0) {
			c0->last_mg_iterations = n;
			c0->last_mg_tol        = tolerance;
		}
	}
	// (one iteration EARLIER than the previous solve ended: on changed data the count moves by one either way -- 512^3 with three
	// data sets in turn took 6, 7, 6 iterations -- and a V-cycle launched behind an unseen stop is a whole cycle wasted (10 ms
	// there) where a look costs 40 us)
	const int predicted = (c0->predictable_start && c0->last_mg_tol == tolerance && !test_switch("FI_LOOK_ALWAYS") && c0->last_mg_iterations > 1)
	                          ? c0->last_mg_iterations - 1 : 0;
	{  // (slabs: the exchanges of every level of the solve, counted by halo_exchange)
		RankSet& top = mixed ? Tw : R;
		for (fi_ctx* l = top[0]; l; l = l->coarse) { l->n_halo_exchanges = 0; }
		c0->n_halo_exchanges = 0;
	}
	// (slabs: the all-reduces of the solve -- dot products of this level, of the start's and the cycles' coarser levels, the
	// replicated tail's vector sums; every level of a context shares one communicator)
	auto reduces_now = [&]() {
		long n = comm_allreduces(c0);
		if (mixed) { n += comm_allreduces(Tw[0]) ; }
		return n;
	};
	const bool same_comm = mixed && Tw[0]->comm == c0->comm;
	const long reduces0 = reduces_now();
	int  steps = 0;
	bool first_restart = true;
	auto restart = [&]() -> int {
		apply_all(R, X, Q, false);
		if (R.size() == 1 && c0->nranks == 1) {  // undivided lattice: one pass for r, r.r and b.b
			double* prr = c0->partial.as<double>();
			double* pbb = prr + static_cast<size_t>(c0->max_blocks);
			hipLaunchKernelGGL((c0->g.nown >= kStreamMin ? k_resid_norms<T, true> : k_resid_norms<T, false>), dim3(nbv(c0)), dim3(kThreads), 0, st, c0->g.nown, vown<T>(c0, B), vown<T>(c0, Q),
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
		const bool look = !(first_restart && predicted > 0);
		first_restart = false;
		const int flag = look ? read_flag() : 0;
		if (flag) { return flag; }
		if (mixed) { Tw[0]->bx_dot_done = false; }
		precondition<T>(R, Tw, Rv, Z, false);
		dot_rz();
		reduce_rz(kMgInitRz);
		direction(1);
		return 0;
	};
	int done = restart();

	double limit_s = 600.0;
	if (const char* env = getenv("FI_SOLVE_TIMEOUT_S")) { limit_s = atof(env); }
	const auto wall0 = std::chrono::steady_clock::now();
	bool timed_out = false;
	int restarts_left = (c0->verify_residual && !by_field) ? 3 : 0;  // (by the field: the recurrence of an fp64 CG is the residual)
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
		// (over slabs the guard is an all-reduce and a host round trip: every eighth iteration -- the same ones on every rank --
		// is often enough for a limit of minutes)
		if ((c0->nranks == 1 || (steps & 7) == 7) &&
		    timed_out_anywhere(R, std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count() > limit_s)) {
			timed_out = true;
			break;
		}
		// the apply of the first two iterations is timed (fi_stats::spmv_ms_avg): an event record is a marker the stream stops
		// at -- ~6 us on either side of the launch, 11 us per timed iteration: timing all five of config 4's cost the solve 1 %
		const bool sample = samples < 2;
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
					hipLaunchKernelGGL(c->g.nown >= kStreamMin ? k_mg_step_mixed<true> : k_mg_step_mixed<false>, dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, c->scal.as<CgScalars>(),
					                   vown<double>(c, P), vown<double>(c, Q), vown<double>(c, X), vown<double>(c, Rv),
					                   vown<float>(Tw[i], &fi_ctx::r), c->partial.as<double>(), field_part(c));
				}
				stepped = true;
			}
		}
		if (!stepped) {
			for (fi_ctx* c : R) {
				hipLaunchKernelGGL((k_mg_step<T>), dim3(nbv(c)), dim3(kThreads), 0, c->stream, c->g.nown, c->scal.as<CgScalars>(),
				                   vown<T>(c, P), vown<T>(c, Q), vown<T>(c, X), vown<T>(c, Rv), c->partial.as<double>(), field_part(c));
			}
		}
		if (by_field && field_slabs == 0) {
			hipLaunchKernelGGL(k_field_max, dim3(1), dim3(kThreads), 0, st, sc0, field_part(c0), nbv(c0));
			mg_reduce(R, nbv, kMgResid);
		} else if (by_field) {
			// mg_reduce with the maxima on board: local sums and maxima, ONE sum over the slabs of sums[0 .. 3] + rank_max[], the logic
			for (size_t i = 0; i < R.size(); ++i) {
				fi_ctx* c = R[i];
				hipLaunchKernelGGL(k_reduce, dim3(1), dim3(kThreads), 0, c->stream, c->scal.as<CgScalars>(), c->partial.as<double>(), 1, nbv(c),
				                   nbv(c), 1);
				hipLaunchKernelGGL(k_field_max_slot, dim3(1), dim3(kThreads), 0, c->stream, c->scal.as<CgScalars>(), field_part(c), nbv(c),
				                   R.size() > 1 ? static_cast<int>(i) : c->rank, field_slabs);
			}
			if (R.size() > 1) {
				hipLaunchKernelGGL(k_group_sum, dim3(1), dim3(1), 0, st, c0->group_scal.as<CgScalars*>(), static_cast<int>(R.size()), 1, 0);
				hipLaunchKernelGGL(k_group_sum_field, dim3(1), dim3(1), 0, st, c0->group_scal.as<CgScalars*>(), static_cast<int>(R.size()));
			} else {
				allreduce_sum(c0, sc0->sums, 4 + 2 * field_slabs);
			}
			for (fi_ctx* c : R) {
				hipLaunchKernelGGL(k_mg_logic, dim3(1), dim3(kThreads), 0, c->stream, c->scal.as<CgScalars>(), static_cast<const double*>(nullptr), 0,
				                   kMgResid);
			}
		} else {
			mg_reduce(R, nbv, kMgResid);
		}
		++steps;
		done = steps < predicted ? 0 : read_flag();
		if (by_field && tuning_switch("FI_FIELD_TRACE")) {  // (timing builds: the rule's history, iteration by iteration -- tools/scratch/r6_field_trace.py)
			if (steps < predicted) { (void)read_flag(); }
			const CgScalars& h = *c0->scal_host;
			std::fprintf(stderr, "field trace %d r %.6e s %.6e t2 %.6e est %.6e done %d\n", h.iter, h.bb > 0 ? std::sqrt(h.rr / h.bb) : 0.0,
			             h.hist_s[h.iter % kFieldHist], h.hist_t[h.iter % kFieldHist], h.field_est, h.done);
		}
		if (done) { continue; }
		if (mixed) { Tw[0]->bx_dot_done = false; }
		precondition<T>(R, Tw, Rv, Z, stepped);
		dot_rz();
		reduce_rz(kMgBeta);
		if (flexible) {  // z_(k+1) . q_k for the flexible beta (q: still A p_k)
			if constexpr (std::is_same<T, double>::value) {
				if (mixed) {
					hipLaunchKernelGGL(c0->g.nown >= kStreamMin ? k_dot_mixed<true> : k_dot_mixed<false>, dim3(nbv(c0)), dim3(kThreads), 0, st, c0->g.nown, sc0,
					                   vown<double>(c0, Q), vown<float>(Tw[0], &fi_ctx::mg_x), c0->partial.as<double>());
				} else {
					dot(Q, Z);
				}
			} else {
				dot(Q, Z);
			}
			mg_reduce(R, nbv, kMgFlex);
		}
		direction(0);
		FI_HIP_TRY(hipGetLastError());
	}
	FI_HIP_TRY(hipEventRecord(e1, st));
	FI_HIP_TRY(hipEventSynchronize(e1));
	float ms = 0;
	FI_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	if (tuning_switch("FI_MG_LOOP_DBG")) { std::fprintf(stderr, "cg_run_mg: %d host steps, predicted %d, device iterations %d\n", steps, predicted, c0->scal_host->iter); }
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
		{
			int ex = c0->n_halo_exchanges;
			for (fi_ctx* l = (mixed ? Tw[0] : c0->coarse); l; l = l->coarse) { ex += l->n_halo_exchanges; }
			c->stats.halo_exchanges = ex;
		}
		c->stats.solve_ms     = ms;
		c->stats.iterations   = h.iter;
		c->stats.reductions   = static_cast<int>((reduces_now() - reduces0) / (same_comm ? 2 : 1));
		c->last_mg_iterations = timed_out || h.done == 2 ? 0 : h.iter;  // (the same on every rank: the scalars are sums over all)
		c->last_mg_tol        = tolerance;
		if (R.size() == 1 && c0->predictable_start && !timed_out && h.done != 2) { remember_iterations(c, 1, tolerance, h.iter); }
		// by the field: met when the estimate says so -- or, in fp64, when the recurrence has reached the precision's floor (the
		// field is then as converged as the arithmetic allows).  An fp32 solve that ends at ITS floor (2e-7) with the estimate
		// above the tolerance has not delivered what was asked for: converged = 0, fi_stats::field_estimate says how far off.
		const bool field_ok = by_field && h.done == 1 && ((h.field_est >= 0.0 && h.field_est <= h.field_tol) || sizeof(T) == 8);
		c->stats.converged    = (!timed_out && (h.done == 4 || h.done == 5 || (h.done == 1 && (by_field ? field_ok : !c0->verify_residual)))) ? 1 : 0;
		c->stats.rel_residual = h.bb > 0 ? std::sqrt(h.rr / h.bb) : 0.0;
		c->stats.restarts     = h.restarts;
		c->stats.verified_residual = (h.restarts > 0 && h.bb > 0) ? std::sqrt(h.true_rr / h.bb) : -1.0;
		c->stats.field_estimate     = by_field ? h.field_est : -1.0;
		c->stats.field_per_residual = by_field ? h.field_kappa : 0.0;
		c->stats.stop_residual      = c->stats.rel_residual;
		c->stats.field_rounds       = by_field ? 1 : 0;
		if (h.done == 4) { FI_HIP_TRY(hipMemsetAsync(c->x.p, 0, sizeof(T) * c->g.nloc, c->stream)); }
	}
	FI_REQUIRE(h.done != 2, FI_ERR_BREAKDOWN, "CG breakdown: non-finite or non-positive curvature (p.AtA p = %g)", h.pq);
	FI_REQUIRE(!timed_out, FI_ERR_TIMEOUT, "solve stopped by the wall-clock guard (FI_SOLVE_TIMEOUT_S = %g s) after %d iterations, "
	           "relative residual %g", limit_s, h.iter, h.bb > 0 ? std::sqrt(h.rr / h.bb) : 0.0);
}



// ---- explicit instantiations (declared in fi_solver_internal.h) ----
template bool cascade_guess<float>(RankSet&, RankSet*);
template bool cascade_guess<double>(RankSet&, RankSet*);
template void cg_run_mg<float>(RankSet&, int, float);
template void cg_run_mg<double>(RankSet&, int, float);
template void mg_alloc<float>(fi_ctx*);
template void mg_alloc<double>(fi_ctx*);

}  // namespace fi
