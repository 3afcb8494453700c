// fi_operator.hip -- the matrix-free normal-equation operator  y = (A^T A) x.
//
// Reference path replaced: make_square (sparse_linear.cpp:105-113) builds A^T A explicitly with a
// sparse*sparse product and every solver iteration multiplies by it (Eigen CSC SpMV, :199-206, :235).
// Here A^T A is never formed:
//
//   model rows  (add_model_constraint, field_interpolation.cpp:243-316) are forward-anchored finite
//               differences along one axis; y += S^T (S x) is evaluated per lattice point from the
//               2k+1 neighbours on that axis, with the reference's existence rule for boundary rows
//               (row anchored at a exists iff 0 <= a and a + k < size, :265,273,282,292) applied through
//               GLOBAL coordinates, so slabs and tiles reproduce the 1,5,6,...,6,5,1 boundary diagonal.
//   data rows   live as one symmetric 2^D x 2^D block per occupied cell (fi_assembly.hip).
//
// Kernels in this file
//   k_apply_generic   any D, any model order, gradient_smoothness; one thread per owned point, direct
//                     (L1/L2-served) neighbour loads.  1-D lattices, model_3/4/gradient_smoothness, and the
//                     tile operator of fi_tile_pass (row members outside the point's tile dropped).
//   k_apply_cells     one thread per occupied cell: y[corners] += B x[corners] (atomics).
//   k_model_diag      analytic diag of the model part; k_invert_diag: Jacobi scaling.
//   k_error_model / k_error_rows   generate_error_map.
// The LDS-tiled kernels for 3-D / 2-D lattices with model_0/1/2 are in fi_stencil.hip / fi_stencil2d.hip.

#include "fi_internal.h"

namespace fi {

namespace {

constexpr int kThreads = 256;

__device__ inline double wave_sum(double v)
{
	for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o, 64); }
	return v;
}

// Sum over the 256 threads of a block; result valid in thread 0.
__device__ inline double block_sum(double v)
{
	__shared__ double s[kThreads / 64];
	v = wave_sum(v);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (lane == 0) { s[wave] = v; }
	__syncthreads();
	double r = 0;
	if (threadIdx.x == 0) {
		for (int w = 0; w < kThreads / 64; ++w) { r += s[w]; }
	}
	__syncthreads();
	return r;
}

__host__ __device__ inline int packed_index(int i, int j, int nc)  // i <= j
{
	return i * nc - (i * (i - 1)) / 2 + (j - i);
}

template <int D>
__device__ inline int64_t owned_to_local(const Geom& g, int64_t o, int* li)
{
	int64_t idx = 0;
	for (int d = 0; d < D; ++d) {
		const int ext = g.own_hi[d] - g.own_lo[d];
		li[d] = g.own_lo[d] + static_cast<int>(o % ext);
		o /= ext;
		idx += static_cast<int64_t>(li[d]) * g.stride[d];
	}
	return idx;
}

template <int D, typename T>
__global__ __launch_bounds__(kThreads) void k_apply_generic(Geom g, ModelCoef<T> mc, const T* __restrict__ x,
                                                             T* __restrict__ y, double* __restrict__ partial,
                                                             const int* __restrict__ done, int ts)
{
	// ts > 0: the tile operator of tile_solver_square (sparse_linear.cpp:246-390): entries (i, j) of AtA are kept
	// only when i and j lie in the same ts^D tile, plus 1e-6 on the diagonal (:296-300).  Row by row that is
	// "every row split into its per-tile pieces": a row member j contributes to point c iff tile(j) == tile(c).
	if (done && *done) { return; }
	double contrib = 0.0;
	for (int64_t o = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; o < g.nown;
	     o += static_cast<int64_t>(gridDim.x) * kThreads) {
		int li[3] = {0, 0, 0};
		const int64_t idx = owned_to_local<D>(g, o, li);
		const T xi = x[idx];
		T acc = 0;
		for (int d = 0; d < D; ++d) {
			const int     c = li[d] + g.off[d];
			const int     n = g.gn[d];
			const int64_t s = g.stride[d];
			if (mc.on[0]) { acc += mc.w0sq * xi; }
			if (mc.maxk > 0) {
				T win[9];
				for (int m = -4; m <= 4; ++m) {
					const int gc = c + m;
					const bool in = (m >= -mc.maxk) && (m <= mc.maxk) && (gc >= 0) && (gc < n);
					const bool same = ts <= 0 || (in && gc / ts == c / ts);
					win[m + 4] = (in && same) ? x[idx + m * s] : T(0);
				}
				for (int k = 1; k <= 4; ++k) {
					if (!mc.on[k]) { continue; }
					for (int m = 0; m <= k; ++m) {
						const int a = c - m;  // anchor of a row that touches this point with coefficient c[k][m]
						if (a >= 0 && a + k < n) {
							T t = 0;
							for (int j = 0; j <= k; ++j) { t += mc.c[k][j] * win[4 - m + j]; }
							acc += mc.c[k][m] * t;
						}
					}
				}
			}
		}
		if (mc.on[5]) {
			// field_interpolation.cpp:303-315: rows [-1,+1,+1,-1]*gs on {0, s_d, s_o, s_o+s_d}; every
			// unordered axis pair is emitted twice (once from each axis).
			for (int d = 0; d < D; ++d) {
				for (int e = d + 1; e < D; ++e) {
					const int cd = li[d] + g.off[d], ce = li[e] + g.off[e];
					const int64_t sd = g.stride[d], se = g.stride[e];
					for (int bd = 0; bd < 2; ++bd) {
						for (int be = 0; be < 2; ++be) {
							const int ad = cd - bd, ae = ce - be;
							if (ad >= 0 && ad + 1 < g.gn[d] && ae >= 0 && ae + 1 < g.gn[e]) {
								const int64_t a = idx - bd * sd - be * se;
								T m00 = T(1), m10 = T(1), m01 = T(1), m11 = T(1);  // members in the tile of this point
								if (ts > 0) {
									const bool d0 = ad / ts == cd / ts, d1 = (ad + 1) / ts == cd / ts;
									const bool e0 = ae / ts == ce / ts, e1 = (ae + 1) / ts == ce / ts;
									m00 = (d0 && e0) ? T(1) : T(0);
									m10 = (d1 && e0) ? T(1) : T(0);
									m01 = (d0 && e1) ? T(1) : T(0);
									m11 = (d1 && e1) ? T(1) : T(0);
								}
								const T row = mc.gs * (-m00 * x[a] + m10 * x[a + sd] + m01 * x[a + se] - m11 * x[a + sd + se]);
								const T sign = (bd ^ be) ? T(1) : T(-1);
								acc += T(2) * (sign * mc.gs) * row;
							}
						}
					}
				}
			}
		}
		if (ts > 0) { acc += T(1e-6f) * xi; }
		y[idx] = acc;
		contrib += static_cast<double>(xi) * static_cast<double>(acc);
	}
	if (partial) {
		const double s = block_sum(contrib);
		if (threadIdx.x == 0) { partial[blockIdx.x] = s; }
	}
}

// ---- the wide model rows of a 3-D lattice the marching kernel covers (MarchState::wide) -----------------------------------
// y += (S3^T S3 + S4^T S4 + the gradient_smoothness rows) x: model_3 [+1,-3,+3,-1], model_4 [+1,-4,+6,-4,+1] along every
// axis, gradient_smoothness [-1,+1,+1,-1] on every axis pair (field_interpolation.cpp:282-315), rows that exist by the
// reference's rule in GLOBAL coordinates, coefficients fp32(stencil * weight) as in A.  The marching kernel has stored
// (A_model_0/1/2 + A_data) x in y; this kernel adds onto it -- every point by its own thread, a plain read-add-write: the
// same bits on every run.  A workgroup owns 64 x 4 points of one plane and walks over the tiles of its share of the lattice;
// a point's neighbours (+-4 per axis, the 3 x 3 patches of the axis pairs) come through L1 / L2: the two rows and the
// eight planes around a tile are another workgroup's tiles.  Rounds 1-4 ran these contexts through k_apply_generic + 2^D
// colour launches of k_apply_cells: 0.7-1.5 ms per apply at 256^3 (0.03-0.06 of the HBM peak, profiles/r4_wide_stencils.txt).
constexpr int kWideTX = 64, kWideTY = 4, kWideTZ = 4;
// the 2R + 1 coefficients of (S3^T S3 + S4^T S4) along an axis where every row exists (R points or more from both ends)
template <typename T>
struct WideTaps {
	T cf[9];  // cf[4 + t], t = -4 .. 4
};
template <typename T, bool K3, bool K4, int R>
__device__ inline T wide_axis(const ModelCoef<T>& mc, const WideTaps<T>& tp, const T* win, int c, int n)  // win[R] = the point itself
{
	if (c >= R && c + R < n) {  // interior: one fixed stencil
		T acc = 0;
#pragma unroll
		for (int t = -R; t <= R; ++t) { acc += tp.cf[4 + t] * win[R + t]; }
		return acc;
	}
	T acc = 0;
	if (K3) {
#pragma unroll
		for (int m = 0; m <= 3; ++m) {
			const int a = c - m;  // anchor of a row that touches this point with coefficient c[3][m]
			T tsum = 0;
#pragma unroll
			for (int j = 0; j <= 3; ++j) { tsum += mc.c[3][j] * win[R - m + j]; }
			acc += (a >= 0 && a + 3 < n) ? mc.c[3][m] * tsum : T(0);
		}
	}
	if (K4) {
#pragma unroll
		for (int m = 0; m <= 4; ++m) {
			const int a = c - m;
			T tsum = 0;
#pragma unroll
			for (int j = 0; j <= 4; ++j) { tsum += mc.c[4][j] * win[R - m + j]; }
			acc += (a >= 0 && a + 4 < n) ? mc.c[4][m] * tsum : T(0);
		}
	}
	return acc;
}

template <typename T, bool K3, bool K4, bool GS>
__global__ __launch_bounds__(kThreads) void k_add_wide3(Geom g, ModelCoef<T> mc, WideTaps<T> tp, const T* __restrict__ x, T* __restrict__ y,
                                                         double* __restrict__ partial, const int* __restrict__ done, int tiles_x,
                                                         int tiles_y, int ntiles)
{
	constexpr int R = K4 ? 4 : 3;
	constexpr bool STAR = K3 || K4;
	const int stop = done ? *done : 0;
	if (stop) { return; }
	const int tx = threadIdx.x % kWideTX, ty = threadIdx.x / kWideTX;
	const int ext0 = g.own_hi[0] - g.own_lo[0], ext1 = g.own_hi[1] - g.own_lo[1], ext2 = g.own_hi[2] - g.own_lo[2];
	const int64_t s1 = g.stride[1], s2 = g.stride[2];
	double contrib = 0.0;
	for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
		const int tz = t / (tiles_x * tiles_y), r = t - tz * (tiles_x * tiles_y);
		const int oy = (r / tiles_x) * kWideTY + ty, ox = (r % tiles_x) * kWideTX + tx;
		if (ox >= ext0 || oy >= ext1) { continue; }
		const int lx = g.own_lo[0] + ox, ly = g.own_lo[1] + oy, lz0 = g.own_lo[2] + tz * kWideTZ;
		const int cx = lx + g.off[0], cy = ly + g.off[1];
		const int64_t col = lx + ly * s1;
		// the thread's column of kWideTZ points: its z window is loaded once (R planes below, R above)
		T wz[STAR ? kWideTZ + 2 * R : 1];
		if (STAR) {
#pragma unroll
			for (int j = 0; j < kWideTZ + 2 * R; ++j) {
				const int lz = lz0 + j - R, gz = lz + g.off[2];
				wz[j] = (gz >= 0 && gz < g.gn[2] && lz >= 0 && lz < g.n[2]) ? x[col + lz * s2] : T(0);
			}
		}
#pragma unroll
		for (int k = 0; k < kWideTZ; ++k) {
			if (tz * kWideTZ + k >= ext2) { break; }
			const int lz = lz0 + k, cz = lz + g.off[2];
			const int64_t idx = col + lz * s2;
			const T xi = STAR ? wz[k + R] : x[idx];
			T acc = 0;
			if (STAR) {
				T win[2 * R + 1];
#pragma unroll
				for (int m = -R; m <= R; ++m) { win[m + R] = (cx + m >= 0 && cx + m < g.gn[0]) ? x[idx + m] : T(0); }
				acc += wide_axis<T, K3, K4, R>(mc, tp, win, cx, g.gn[0]);
#pragma unroll
				for (int m = -R; m <= R; ++m) { win[m + R] = (cy + m >= 0 && cy + m < g.gn[1]) ? x[idx + m * s1] : T(0); }
				acc += wide_axis<T, K3, K4, R>(mc, tp, win, cy, g.gn[1]);
				acc += wide_axis<T, K3, K4, R>(mc, tp, &wz[k], cz, g.gn[2]);
			}
			if (GS) {
				const int li[3] = {lx, ly, lz};
#pragma unroll
				for (int d = 0; d < 3; ++d) {
#pragma unroll
					for (int e = d + 1; e < 3; ++e) {
						const int cd = li[d] + g.off[d], ce = li[e] + g.off[e];
						const int64_t sd = d == 0 ? 1 : (d == 1 ? s1 : s2), se = e == 1 ? s1 : s2;
#pragma unroll
						for (int bd = 0; bd < 2; ++bd) {
#pragma unroll
							for (int be = 0; be < 2; ++be) {
								const int ad = cd - bd, ae = ce - be;
								if (ad >= 0 && ad + 1 < g.gn[d] && ae >= 0 && ae + 1 < g.gn[e]) {
									const int64_t a = idx - bd * sd - be * se;
									const T row  = mc.gs * (-x[a] + x[a + sd] + x[a + se] - x[a + sd + se]);
									const T sign = (bd ^ be) ? T(1) : T(-1);
									acc += T(2) * (sign * mc.gs) * row;   // (every axis pair is emitted twice: cpp:303-315)
								}
							}
						}
					}
				}
			}
			y[idx] += acc;
			contrib += static_cast<double>(xi) * static_cast<double>(acc);
		}
	}
	if (partial) {
		const double sum = block_sum(contrib);
		if (threadIdx.x == 0) { partial[blockIdx.x] = sum; }
	}
}

struct WideLaunch {
	int tiles_x, tiles_y, ntiles, blocks;
};
inline WideLaunch wide_launch(const fi_ctx* c)
{
	const Geom& g = c->g;
	WideLaunch w;
	w.tiles_x = (g.own_hi[0] - g.own_lo[0] + kWideTX - 1) / kWideTX;
	w.tiles_y = (g.own_hi[1] - g.own_lo[1] + kWideTY - 1) / kWideTY;
	const int64_t nt = static_cast<int64_t>(w.tiles_x) * w.tiles_y * ((g.own_hi[2] - g.own_lo[2] + kWideTZ - 1) / kWideTZ);
	w.ntiles = static_cast<int>(nt);
	w.blocks = static_cast<int>(nt < 2048 ? (nt < 1 ? 1 : nt) : 2048);  // (its partials are summed by single-workgroup kernels)
	return w;
}

template <typename T>
__device__ inline void atomic_add(T* p, T v)
{
	unsafeAtomicAdd(p, v);
}

// y[corners] += B x[corners] for every occupied cell.  Inputs may sit on ghost planes; outputs go to
// owned points only.  fp32/fp64 hardware atomics (global_atomic_add_f32 / _f64, no CAS loop).
// colour >= 0: only the cells whose origin has that parity pattern (bit d = parity along axis d), with plain
// read-add-write instead of atomics: two cells of one colour are at least two apart along some axis and share no corner, so a
// launch per colour, 2^D launches in colour order, forms every sum in the same order on every run -- bitwise reproducible,
// which the atomic form (colour < 0: FI_CELLS_ATOMIC, tests) is not.  The path of the contexts the tiled kernels do not
// cover (1-D lattices, model_3 / model_4, gradient_smoothness: field_interpolation.cpp:282-315) and of the tile operator.
template <int D, typename T>
__global__ __launch_bounds__(kThreads) void k_apply_cells(Geom g, int64_t ncell, const uint32_t* __restrict__ cell_id,
                                                           const T* __restrict__ blk, const T* __restrict__ x,
                                                           T* __restrict__ y, double* __restrict__ partial,
                                                           const int* __restrict__ done, int ts, int colour)
{
	if (done && *done) { return; }
	constexpr int NC = 1 << D;
	double contrib = 0.0;
	for (int64_t c = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; c < ncell;
	     c += static_cast<int64_t>(gridDim.x) * kThreads) {
		uint32_t id = cell_id[c];
		int l[3] = {0, 0, 0};
		int col = 0;
		for (int d = 0; d < D; ++d) {
			l[d] = static_cast<int>(id % static_cast<uint32_t>(g.cn[d]));
			id /= static_cast<uint32_t>(g.cn[d]);
			col |= (l[d] & 1) << d;
		}
		if (colour >= 0 && col != colour) { continue; }
		int64_t idx[NC];
		bool    in[NC], own[NC];
		T       xv[NC];
		int     tid[NC];  // tile of the corner (tile operator only)
		for (int q = 0; q < NC; ++q) {
			int64_t ix = 0;
			bool ok = true, ow = true;
			tid[q] = 0;
			for (int d = 0; d < D; ++d) {
				const int gq = l[d] + g.coff[d] + ((q >> d) & 1);
				if (ts > 0) { tid[q] = tid[q] * 2 + ((gq >= 0 ? gq : 0) / ts - (l[d] + g.coff[d] >= 0 ? l[d] + g.coff[d] : 0) / ts); }
				const int li = gq - g.off[d];
				ok = ok && (0 <= gq) && (gq < g.gn[d]);
				ow = ow && (g.own_lo[d] <= li) && (li < g.own_hi[d]);
				ix += static_cast<int64_t>(li) * g.stride[d];
			}
			idx[q] = ix;
			in[q]  = ok;
			own[q] = ok && ow;
			xv[q]  = ok ? x[ix] : T(0);
		}
		for (int i = 0; i < NC; ++i) {
			if (!own[i]) { continue; }
			T s = 0;
			for (int j = 0; j < NC; ++j) {
				const int e = i <= j ? packed_index(i, j, NC) : packed_index(j, i, NC);
				if (ts > 0 && tid[i] != tid[j]) { continue; }
				s += blk[c * (NC * (NC + 1) / 2) + e] * xv[j];
			}
			if (colour >= 0) { y[idx[i]] += s; } else { atomic_add(&y[idx[i]], s); }
			contrib += static_cast<double>(xv[i]) * static_cast<double>(s);
		}
		(void)in;
	}
	if (partial) {
		const double s = block_sum(contrib);
		if (threadIdx.x == 0) { partial[blockIdx.x] = colour > 0 ? partial[blockIdx.x] + s : s; }  // (the colours' launches follow each other on the stream)
	}
}

// diag(A^T A) of the model rows, added onto `diag` (which already holds the data part).
template <int D, typename T>
__global__ __launch_bounds__(kThreads) void k_model_diag(Geom g, ModelCoef<T> mc, T* __restrict__ diag, T* __restrict__ dinv,
                                                         unsigned short* __restrict__ d16)
{
	const int64_t o = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (o >= g.nown) { return; }
	int li[3] = {0, 0, 0};
	const int64_t idx = owned_to_local<D>(g, o, li);
	T acc = 0;
	for (int d = 0; d < D; ++d) {
		const int c = li[d] + g.off[d], n = g.gn[d];
		if (mc.on[0]) { acc += mc.w0sq; }
		for (int k = 1; k <= 4; ++k) {
			if (!mc.on[k]) { continue; }
			for (int m = 0; m <= k; ++m) {
				const int a = c - m;
				if (a >= 0 && a + k < n) { acc += mc.c[k][m] * mc.c[k][m]; }
			}
		}
	}
	if (mc.on[5]) {
		for (int d = 0; d < D; ++d) {
			for (int e = d + 1; e < D; ++e) {
				const int cd = li[d] + g.off[d], ce = li[e] + g.off[e];
				for (int bd = 0; bd < 2; ++bd) {
					for (int be = 0; be < 2; ++be) {
						const int ad = cd - bd, ae = ce - be;
						if (ad >= 0 && ad + 1 < g.gn[d] && ae >= 0 && ae + 1 < g.gn[e]) {
							acc += T(2) * mc.gs * mc.gs;
						}
					}
				}
			}
		}
	}
	const T d = diag[idx] + acc;
	diag[idx] = d;
	if (dinv) {  // undivided lattice: the Jacobi scaling in the same pass (k_invert_diag otherwise: ghost planes too)
		const T v = (d != T(0)) ? T(1) / d : T(1);
		dinv[idx] = v;
		d16[idx]  = static_cast<unsigned short>(__float_as_uint(static_cast<float>(v)) >> 16);
	}
}

// Eigen::DiagonalPreconditioner semantics: 1/diag, or 1 where diag == 0.
// d16: the same scaling cut to bfloat16 (truncated: never above 1/diag), read by the recurrences that run in the
// marching kernel's epilogue -- any fixed positive diagonal is a valid scaling there, and this one is half (a quarter in
// fp64) of a lattice pass per step.
template <typename T>
__global__ __launch_bounds__(kThreads) void k_invert_diag(int64_t n, const T* __restrict__ diag, T* __restrict__ dinv,
                                                           unsigned short* __restrict__ d16)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (i < n) {
		const T d = diag[i];
		const T v = (d != T(0)) ? T(1) / d : T(1);
		dinv[i]   = v;
		d16[i]    = static_cast<unsigned short>(__float_as_uint(static_cast<float>(v)) >> 16);
	}
}

inline int blocks_for(int64_t n) { return static_cast<int>((n + kThreads - 1) / kThreads); }

// The scaling of the V-cycle's polynomial smoother (fi_ctx::dinv16s): bfloat16 of 1 / (m + f (diag - m)), m = the model
// diagonal from the global coordinates (as k_model_diag), over ALL local points: the ghost planes of `diag` hold the
// neighbours' values where the scaling is read there (fi_ctx::scaling_ghosts).  Truncated like dinv16: never above.
template <int D, typename T>
__global__ __launch_bounds__(kThreads) void k_safe_scaling(Geom g, ModelCoef<T> mc, const T* __restrict__ diag, T factor,
                                                           unsigned short* __restrict__ d16, unsigned int* __restrict__ weak)
{
	const int64_t idx = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (idx >= g.nloc) { return; }
	int64_t o = idx;
	T m = 0;
	for (int d = 0; d < D; ++d) {
		const int c = static_cast<int>(o % g.n[d]) + g.off[d], n = g.gn[d];
		o /= g.n[d];
		if (mc.on[0]) { m += mc.w0sq; }
		for (int k = 1; k <= 4; ++k) {
			if (!mc.on[k]) { continue; }
			for (int j = 0; j <= k; ++j) {
				const int a = c - j;
				if (a >= 0 && a + k < n) { m += mc.c[k][j] * mc.c[k][j]; }
			}
		}
	}
	const T dg = diag[idx];
	const T dd = dg > m ? dg - m : T(0);
	// (small levels only: how many points the data hold less firmly than the model couples them -- fi_ctx::data_pinned)
	if (weak && dd < m) { atomicAdd(weak, 1u); }
	const T s  = m + factor * dd;
	const T v  = (s > T(0)) ? T(1) / s : T(1);
	d16[idx] = static_cast<unsigned short>(__float_as_uint(static_cast<float>(v)) >> 16);
}

// grid-stride launch width: enough blocks to fill 256 CUs x 8, few enough that the fixed-order sum of
// the per-block dot-product partials stays a ~microsecond single-block kernel
inline int capped_blocks(int64_t n)
{
	const int64_t b = (n + kThreads - 1) / kThreads;
	return static_cast<int>(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

template <typename T>
ModelCoef<T> make_coef(const fi_weights& w)
{
	// A holds fp32(stencil * weight) (sparse_linear.cpp:43); square in T afterwards.
	static const float pascal[5][5] = {
	    {1, 0, 0, 0, 0}, {-1, +1, 0, 0, 0}, {+1, -2, +1, 0, 0}, {+1, -3, +3, -1, 0}, {+1, -4, +6, -4, +1}};
	const float wk[5] = {w.model_0, w.model_1, w.model_2, w.model_3, w.model_4};
	ModelCoef<T> mc{};
	mc.maxk = 0;
	for (int k = 0; k <= 4; ++k) {
		mc.on[k] = wk[k] > 0.0f;  // field_interpolation.cpp:257,265,273,282,292: `weights.model_k > 0`
		for (int m = 0; m <= 4; ++m) {
			volatile float prod = pascal[k][m] * wk[k];
			mc.c[k][m] = mc.on[k] ? static_cast<T>(prod) : T(0);
		}
		if (k >= 1 && mc.on[k]) { mc.maxk = k; }
	}
	{
		volatile float w0 = 1.0f * w.model_0;
		mc.w0sq = mc.on[0] ? static_cast<T>(w0) * static_cast<T>(w0) : T(0);
	}
	mc.on[5] = w.gradient_smoothness > 0.0f;
	mc.gs    = mc.on[5] ? static_cast<T>(w.gradient_smoothness) : T(0);
	return mc;
}

// The same for 3-D lattices without gradient_smoothness, a row of x per workgroup row: the y and z parts of the model
// diagonal are uniform over the workgroup, no 64-bit divisions per point (256^3 fp64: 203 -> ... us, the kernel was
// bound by its index arithmetic, not by the 436 MB it moves).  Same sums in the same order: the same bits.
constexpr int kDiagRows = 4;  // rows of x per workgroup (256^3: 65 536 workgroups of one row each were bound by their dispatch)
template <typename T>
__global__ __launch_bounds__(kThreads) void k_model_diag3(Geom g, ModelCoef<T> mc, T* __restrict__ diag, T* __restrict__ dinv,
                                                          unsigned short* __restrict__ d16, unsigned short* __restrict__ d16s, T factor,
                                                          const float* __restrict__ lump_in, float* __restrict__ dlump_out)
{
	const int ext0 = g.own_hi[0] - g.own_lo[0], ext1 = g.own_hi[1] - g.own_lo[1];
	const int x = static_cast<int>(blockIdx.x) * kThreads + threadIdx.x;
	if (x >= ext0) { return; }
	const int lx = g.own_lo[0] + x, lz = g.own_lo[2] + static_cast<int>(blockIdx.z);
	const int y0 = static_cast<int>(blockIdx.y) * kDiagRows;
	T in[kDiagRows];
#pragma unroll
	for (int j = 0; j < kDiagRows; ++j) {
		const int64_t idx = lx * g.stride[0] + (g.own_lo[1] + y0 + j) * g.stride[1] + lz * g.stride[2];
		if (lump_in) {  // a lumped replica (fi_levels.hip): its data diagonal IS the clamped row sums the fp64 level's assembly formed
			const float v = y0 + j < ext1 ? lump_in[idx] : 0.0f;
			in[j] = static_cast<T>(v > 0.0f ? v : 0.0f);
			if (y0 + j < ext1) { dlump_out[idx] = static_cast<float>(in[j]); }
		} else {
			in[j] = y0 + j < ext1 ? diag[idx] : T(0);
		}
	}
#pragma unroll
	for (int j = 0; j < kDiagRows; ++j) {
		if (y0 + j >= ext1) { break; }
		const int ly = g.own_lo[1] + y0 + j;
		const int64_t idx = lx * g.stride[0] + ly * g.stride[1] + lz * g.stride[2];
		// (sum of the terms in the order d = 0, 1, 2, term by term as before: a running sum, not a sum of three parts)
		T acc = 0;
#pragma unroll
		for (int d = 0; d < 3; ++d) {
			const int li = d == 0 ? lx : (d == 1 ? ly : lz);
			const int c = li + g.off[d], n = g.gn[d];
			if (mc.on[0]) { acc += mc.w0sq; }
			for (int k = 1; k <= 4; ++k) {
				if (!mc.on[k]) { continue; }
				for (int m = 0; m <= k; ++m) {
					const int a = c - m;
					if (a >= 0 && a + k < n) { acc += mc.c[k][m] * mc.c[k][m]; }
				}
			}
		}
		const T d = in[j] + acc;
		__builtin_nontemporal_store(d, diag + idx);  // (streaming stores: see k_tile_sums3)
		if (dinv) {
			const T v = (d != T(0)) ? T(1) / d : T(1);
			__builtin_nontemporal_store(v, dinv + idx);
			__builtin_nontemporal_store(static_cast<unsigned short>(__float_as_uint(static_cast<float>(v)) >> 16), d16 + idx);
		}
		if (d16s) {  // the polynomial smoother's scaling in the same pass (k_safe_scaling: the same m, the same sums)
			const T dd = d > acc ? d - acc : T(0);
			const T sc = acc + factor * dd;
			const T v  = (sc > T(0)) ? T(1) / sc : T(1);
			__builtin_nontemporal_store(static_cast<unsigned short>(__float_as_uint(static_cast<float>(v)) >> 16), d16s + idx);
		}
	}
}

template <int D, typename T>
void prepare_dim(fi_ctx* c, bool with_scaling, const float* lump_in)
{
	const Geom& g = c->g;
	const ModelCoef<T> mc = make_coef<T>(c->w);
	c->dinv.alloc(sizeof(T) * g.nloc);
	c->dinv16.alloc(sizeof(unsigned short) * g.nloc);
	const bool whole = g.nown == g.nloc;  // no ghost planes: every local point is an owned one
	const int ext1 = g.own_hi[1] - g.own_lo[1], ext2 = g.own_hi[2] - g.own_lo[2];
	// with_scaling: the polynomial smoother's scaling of an undivided level in the same pass (levels of up to 2^16 points
	// also want the count of weakly held points: the separate kernel); lump_in: a lumped replica's diagonal from the row sums
	const bool fuse_scaling = with_scaling && whole && c->nranks == 1 && g.nloc > (1 << 16);
	FI_REQUIRE(!lump_in || (D == 3 && !mc.on[5] && ext1 <= 65535 && ext2 <= 65535 && whole && sizeof(T) == 4), FI_ERR_STATE,
	           "lumped replica: unexpected lattice");
	if (D == 3 && !mc.on[5] && ext1 <= 65535 && ext2 <= 65535) {
		if (fuse_scaling) { c->dinv16s.alloc(sizeof(unsigned short) * g.nloc); }
		if (lump_in) { c->dlump.alloc(sizeof(float) * g.nloc); }
		hipLaunchKernelGGL((k_model_diag3<T>), dim3((g.own_hi[0] - g.own_lo[0] + kThreads - 1) / kThreads, (ext1 + kDiagRows - 1) / kDiagRows, ext2),
		                   dim3(kThreads), 0, c->stream, g, mc, c->diag.as<T>(), whole ? c->dinv.as<T>() : static_cast<T*>(nullptr),
		                   whole ? c->dinv16.as<unsigned short>() : static_cast<unsigned short*>(nullptr),
		                   fuse_scaling ? c->dinv16s.as<unsigned short>() : static_cast<unsigned short*>(nullptr),
		                   static_cast<T>(c->lumped ? 1.0 : c->mg_safe), lump_in, lump_in ? c->dlump.as<float>() : static_cast<float*>(nullptr));
		if (fuse_scaling) {
			c->data_pinned   = false;
			c->dinv16s_valid = true;
		}
	} else
	hipLaunchKernelGGL((k_model_diag<D, T>), dim3(blocks_for(g.nown)), dim3(kThreads), 0, c->stream, g, mc,
	                   c->diag.as<T>(), whole ? c->dinv.as<T>() : static_cast<T*>(nullptr),
	                   whole ? c->dinv16.as<unsigned short>() : static_cast<unsigned short*>(nullptr));
	c->scaling_ghosts = false;
	if (!whole) {
		// one slab per process: the neighbours' diagonal on the ghost planes, so that the scaling there is theirs (the
		// polynomial's first step forms its operand from r and the scaling, ghost planes included).  A collective: every
		// rank assembles the same levels in the same order.
		if (c->nranks > 1 && comm_ready(c) && !c->defer_scaling_exchange) {
			exchange_halo(c, c->diag.p, c->min_slab >= c->halo ? c->halo : c->reach);  // (deep: the scaling of the whole ghost zone)
			c->scaling_ghosts = true;
		}
		hipLaunchKernelGGL((k_invert_diag<T>), dim3(blocks_for(g.nloc)), dim3(kThreads), 0, c->stream, g.nloc,
		                   c->diag.as<T>(), c->dinv.as<T>(), c->dinv16.as<unsigned short>());
	}
	FI_HIP_TRY(hipGetLastError());
}

template <typename T>
void finish_ghosts_t(fi_ctx* c)
{
	const Geom& g = c->g;
	exchange_halo(c, c->diag.p, c->min_slab >= c->halo ? c->halo : c->reach);
	hipLaunchKernelGGL((k_invert_diag<T>), dim3(blocks_for(g.nloc)), dim3(kThreads), 0, c->stream, g.nloc, c->diag.as<T>(),
	                   c->dinv.as<T>(), c->dinv16.as<unsigned short>());
	FI_HIP_TRY(hipGetLastError());
	c->scaling_ghosts = true;
}

template <int D, typename T>
void safe_scaling_dim(fi_ctx* c)
{
	const Geom& g = c->g;
	const ModelCoef<T> mc = make_coef<T>(c->w);
	c->dinv16s.alloc(sizeof(unsigned short) * g.nloc);
	// levels of up to 2^16 points (a coarsest level): count the points whose data diagonal is below their model diagonal
	unsigned int* weak = nullptr;
	DevBuf cnt;
	if (g.nloc <= (1 << 16) && c->nranks == 1) {
		cnt.alloc(sizeof(unsigned int));
		FI_HIP_TRY(hipMemsetAsync(cnt.p, 0, sizeof(unsigned int), c->stream));
		weak = cnt.as<unsigned int>();
	}
	hipLaunchKernelGGL((k_safe_scaling<D, T>), dim3(blocks_for(g.nloc)), dim3(kThreads), 0, c->stream, g, mc, c->diag.as<T>(),
	                   static_cast<T>(c->lumped ? 1.0 : c->mg_safe), c->dinv16s.as<unsigned short>(), weak);  // (lumped: diag - m IS the bound)
	FI_HIP_TRY(hipGetLastError());
	c->data_pinned = false;
	if (weak) {
		unsigned int* h = static_cast<unsigned int*>(pinned(c, 0, sizeof(unsigned int)));
		*h = 1;
		FI_HIP_TRY(hipMemcpyAsync(h, weak, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
		FI_HIP_TRY(hipStreamSynchronize(c->stream));
		c->data_pinned = *h == 0;
	}
	c->dinv16s_valid = true;
}

int cells_partials(const fi_ctx* c) { return capped_blocks(c->cells.ncell); }  // partials of the untiled path's cell kernel

template <int D, typename T>
void apply_dim(fi_ctx* c, const T* x, T* y, double* partial)
{
	const Geom& g = c->g;
	const ModelCoef<T> mc = make_coef<T>(c->w);
	const int* done = c->scal.p ? &c->scal.as<CgScalars>()->done : nullptr;
	const int ts = c->tile_ts;  // > 0: the tile operator (fi_tile_pass) through the plain kernels
	int nb_model = ts > 0 ? 0 : stencil_partials(c);
	int nb_cells = 0;
	if (nb_model > 0) {
		stencil_apply(c, x, y, partial);
		if (D == 3 && c->march.valid && c->march.wide) {  // the wide model rows on top of what the marching kernel stored
			const WideLaunch wl = wide_launch(c);
			double* pw = partial ? partial + nb_model : nullptr;
			const bool k3 = mc.on[3], k4 = mc.on[4], gs = mc.on[5];
			WideTaps<T> tp{};
			for (int t = -4; t <= 4; ++t) {  // (S3^T S3 + S4^T S4)(c, c + t) where every row exists
				T v = 0;
				for (int k = 3; k <= 4; ++k) {
					if (!mc.on[k]) { continue; }
					for (int m = 0; m <= k; ++m) {
						const int j = m + t;
						if (j >= 0 && j <= k) { v += mc.c[k][m] * mc.c[k][j]; }
					}
				}
				tp.cf[4 + t] = v;
			}
			auto launch = [&](auto kernel) {
				hipLaunchKernelGGL(kernel, dim3(wl.blocks), dim3(kThreads), 0, c->stream, g, mc, tp, x, y, pw, done, wl.tiles_x, wl.tiles_y, wl.ntiles);
			};
			if (gs) {
				k4 ? (k3 ? launch(k_add_wide3<T, true, true, true>) : launch(k_add_wide3<T, false, true, true>))
				   : (k3 ? launch(k_add_wide3<T, true, false, true>) : launch(k_add_wide3<T, false, false, true>));
			} else {
				k4 ? (k3 ? launch(k_add_wide3<T, true, true, false>) : launch(k_add_wide3<T, false, true, false>))
				   : launch(k_add_wide3<T, true, false, false>);
			}
			nb_model += wl.blocks;
		}
	} else {
		nb_model = capped_blocks(g.nown);
		hipLaunchKernelGGL((k_apply_generic<D, T>), dim3(nb_model), dim3(kThreads), 0, c->stream, g, mc, x, y, partial,
		                   done, ts);
	}
	if (c->cells.ncell > 0 && (ts > 0 || !cells_fused(c))) {
		nb_cells = capped_blocks(c->cells.ncell);
		ensure_cell_blocks(c);
		const int first = test_switch("FI_CELLS_ATOMIC") ? -1 : 0, last = first < 0 ? -1 : (1 << D) - 1;  // (tests compare the two forms)
		for (int colour = first; colour <= last; ++colour) {
			hipLaunchKernelGGL((k_apply_cells<D, T>), dim3(nb_cells), dim3(kThreads), 0, c->stream, g,
			                   c->cells.ncell, c->cells.cell_id.as<uint32_t>(), c->cells.blk.as<T>(), x, y,
			                   partial ? partial + nb_model : nullptr, done, ts, colour);
		}
	}
	if (ts > 0) {
		generic_apply_tile(c, x, y, partial ? partial + nb_model + nb_cells : nullptr, ts);
		FI_HIP_TRY(hipGetLastError());
		return;
	}
	generic_apply(c, x, y, partial ? partial + nb_model + nb_cells : nullptr);
	FI_HIP_TRY(hipGetLastError());
}

}  // namespace

size_t elem_size(const fi_ctx* c) { return c->dtype == FI_F64 ? sizeof(double) : sizeof(float); }

int apply_num_partials(const fi_ctx* c)
{
	if (c->tile_ts > 0) {
		return capped_blocks(c->g.nown) + (c->cells.ncell > 0 ? cells_partials(c) : 0) + generic_num_partials(c);
	}
	int nb_model = stencil_partials(c);
	if (nb_model <= 0) { nb_model = capped_blocks(c->g.nown); }
	if (c->march.valid && c->march.wide) { nb_model += wide_launch(c).blocks; }  // k_add_wide3's share of x . A x
	int n = nb_model + generic_num_partials(c);
	if (!cells_fused(c) && c->cells.ncell > 0) { n += cells_partials(c); }
	return n;
}

double apply_algorithmic_bytes(const fi_ctx* c)
{
	// B_spmv = 2*s*N + C_occ*(4 + s*2^D(2^D+1)/2): read x once, write y once, read every
	// occupied cell's record once.  The fused 3-D kernel keeps a cell that holds a single data row as that
	// row (2^D coefficients) instead of the packed block, so its record is counted at its real, smaller size.
	const double s = static_cast<double>(elem_size(c));
	const double lattice = 2.0 * s * static_cast<double>(c->g.nown) +
	                       2.0 * static_cast<double>(c->generic.nnz) * (4.0 + s);    // generic rows: CSR + CSC pass
	if (c->march.valid && c->march.fused) {
		return lattice + static_cast<double>(c->march.cells_row) * (4.0 + s * 8.0) +
		       static_cast<double>(c->march.cells_blk) * (4.0 + s * 36.0);  // a multi-row cell at its packed-block size
	}
	if (c->tile2.valid && c->tile2.fused) { return lattice + static_cast<double>(c->cells.ncell) * (4.0 + s * 16.0); }
	return lattice + static_cast<double>(c->cells.ncell) * (4.0 + s * static_cast<double>(c->cells.nb));
}

// ---- generate_error_map (field_interpolation.cpp:402-429) --------------------------------------------------
// Every row r = (a, b) blames its unknowns for its squared residual: out[j] += a_j^2 / |a|^2 * (b - a.x)^2.
// Model rows are enumerated per lattice point exactly like k_apply_generic does (row anchored at c-m touches c
// with coefficient c[k][m]); data rows come from the row tables the assembly was built from.
template <int D, typename T>
__global__ __launch_bounds__(kThreads) void k_error_model(Geom g, ModelCoef<T> mc, const T* __restrict__ x,
                                                           T* __restrict__ out)
{
	for (int64_t o = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; o < g.nown;
	     o += static_cast<int64_t>(gridDim.x) * kThreads) {
		int li[3] = {0, 0, 0};
		const int64_t idx = owned_to_local<D>(g, o, li);
		const T xi = x[idx];
		T acc = 0;
		for (int d = 0; d < D; ++d) {
			const int     c = li[d] + g.off[d];
			const int     n = g.gn[d];
			const int64_t s = g.stride[d];
			if (mc.on[0]) { acc += mc.w0sq * xi * xi; }  // row [w0], alone in its row: all the blame
			for (int k = 1; k <= 4; ++k) {
				if (!mc.on[k]) { continue; }
				T sq = 0;
				for (int j = 0; j <= k; ++j) { sq += mc.c[k][j] * mc.c[k][j]; }
				for (int m = 0; m <= k; ++m) {
					const int a = c - m;
					if (a >= 0 && a + k < n) {
						T t = 0;
						for (int j = 0; j <= k; ++j) { t += mc.c[k][j] * x[idx + (j - m) * s]; }
						acc += (mc.c[k][m] * mc.c[k][m] / sq) * t * t;
					}
				}
			}
		}
		if (mc.on[5]) {  // rows [-1,+1,+1,-1]*gs, each unordered axis pair emitted twice (cpp:303-315): blame 1/4 each
			for (int d = 0; d < D; ++d) {
				for (int e = d + 1; e < D; ++e) {
					const int cd = li[d] + g.off[d], ce = li[e] + g.off[e];
					const int64_t sd = g.stride[d], se = g.stride[e];
					for (int bd = 0; bd < 2; ++bd) {
						for (int be = 0; be < 2; ++be) {
							const int ad = cd - bd, ae = ce - be;
							if (ad >= 0 && ad + 1 < g.gn[d] && ae >= 0 && ae + 1 < g.gn[e]) {
								const int64_t a = idx - bd * sd - be * se;
								const T row = mc.gs * (-x[a] + x[a + sd] + x[a + se] - x[a + sd + se]);
								acc += T(2) * T(0.25) * row * row;
							}
						}
					}
				}
			}
		}
		out[idx] = acc;
	}
}

template <int D, typename T>
__global__ __launch_bounds__(kThreads) void k_error_rows(Geom g, int64_t nrows, uint32_t invalid,
                                                          const uint32_t* __restrict__ key,
                                                          const float* __restrict__ coef, const float* __restrict__ rhs,
                                                          const T* __restrict__ x, T* __restrict__ out)
{
	constexpr int NC = 1 << D;
	for (int64_t r = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; r < nrows;
	     r += static_cast<int64_t>(gridDim.x) * kThreads) {
		uint32_t id = key[r];
		if (id >= invalid) { continue; }  // slots of rows that were not emitted carry the number of extended cells
		int l[3] = {0, 0, 0};
		for (int d = 0; d < D; ++d) {
			l[d] = static_cast<int>(id % static_cast<uint32_t>(g.cn[d]));
			id /= static_cast<uint32_t>(g.cn[d]);
		}
		int64_t idx[NC];
		bool    own[NC];
		T       a[NC];
		T       res = static_cast<T>(rhs[r]), sq = 0;
		for (int q = 0; q < NC; ++q) {
			int64_t ix = 0;
			bool ok = true, ow = true;
			for (int d = 0; d < D; ++d) {
				const int gq = l[d] + g.coff[d] + ((q >> d) & 1);
				const int lq = gq - g.off[d];
				ok = ok && (0 <= gq) && (gq < g.gn[d]);
				ow = ow && (g.own_lo[d] <= lq) && (lq < g.own_hi[d]);
				ix += static_cast<int64_t>(lq) * g.stride[d];
			}
			idx[q] = ix;
			own[q] = ok && ow;
			a[q]   = ok ? static_cast<T>(coef[r * NC + q]) : T(0);
			if (ok) { res -= a[q] * x[ix]; }
			sq += a[q] * a[q];
		}
		if (!(sq > 0)) { continue; }
		const T e = res * res / sq;
		for (int q = 0; q < NC; ++q) {
			if (own[q] && a[q] != T(0)) { atomic_add(&out[idx[q]], a[q] * a[q] * e); }
		}
	}
}

template <int D, typename T>
void error_map_dim(fi_ctx* c, const T* x, T* out)
{
	const Geom& g = c->g;
	const ModelCoef<T> mc = make_coef<T>(c->w);
	hipLaunchKernelGGL((k_error_model<D, T>), dim3(capped_blocks(g.nown)), dim3(kThreads), 0, c->stream, g, mc, x, out);
	for (const Pending* pb : c->pending) {
		if (pb->nrows == 0) { continue; }
		hipLaunchKernelGGL((k_error_rows<D, T>), dim3(capped_blocks(pb->nrows)), dim3(kThreads), 0, c->stream, g,
		                   static_cast<int64_t>(pb->nrows),
		                   static_cast<uint32_t>(static_cast<int64_t>(g.cn[0]) * g.cn[1] * g.cn[2]), pb->key.as<uint32_t>(),
		                   pb->coef.as<float>(),
		                   pb->rhs.as<float>(), x, out);
	}
	FI_HIP_TRY(hipGetLastError());
	generic_error_map(c, x, out);
}

void error_map(fi_ctx* c, const void* x, void* out)
{
	const bool f64 = c->dtype == FI_F64;
	switch (c->g.ndim) {
	case 1:
		f64 ? error_map_dim<1, double>(c, static_cast<const double*>(x), static_cast<double*>(out))
		    : error_map_dim<1, float>(c, static_cast<const float*>(x), static_cast<float*>(out));
		break;
	case 2:
		f64 ? error_map_dim<2, double>(c, static_cast<const double*>(x), static_cast<double*>(out))
		    : error_map_dim<2, float>(c, static_cast<const float*>(x), static_cast<float*>(out));
		break;
	default:
		f64 ? error_map_dim<3, double>(c, static_cast<const double*>(x), static_cast<double*>(out))
		    : error_map_dim<3, float>(c, static_cast<const float*>(x), static_cast<float*>(out));
		break;
	}
}

void operator_finish_ghosts(fi_ctx* c)
{
	c->defer_scaling_exchange = false;
	if (c->nranks <= 1 || !comm_ready(c) || c->g.nown == c->g.nloc) { return; }
	c->dtype == FI_F64 ? finish_ghosts_t<double>(c) : finish_ghosts_t<float>(c);
	c->dinv16s_valid = false;
}

// The scaling over ALL local planes from `diag` as it stands (the loop-back group has filled the ghost planes by device
// copies: fi_group_assemble)
void operator_rescale_with_ghosts(fi_ctx* c)
{
	const Geom& g = c->g;
	if (c->dtype == FI_F64) {
		hipLaunchKernelGGL((k_invert_diag<double>), dim3(blocks_for(g.nloc)), dim3(kThreads), 0, c->stream, g.nloc, c->diag.as<double>(),
		                   c->dinv.as<double>(), c->dinv16.as<unsigned short>());
	} else {
		hipLaunchKernelGGL((k_invert_diag<float>), dim3(blocks_for(g.nloc)), dim3(kThreads), 0, c->stream, g.nloc, c->diag.as<float>(),
		                   c->dinv.as<float>(), c->dinv16.as<unsigned short>());
	}
	FI_HIP_TRY(hipGetLastError());
	c->scaling_ghosts = true;
	c->dinv16s_valid  = false;
}

void prepare_safe_scaling(fi_ctx* c)
{
	const bool f64 = c->dtype == FI_F64;
	switch (c->g.ndim) {
	case 1: f64 ? safe_scaling_dim<1, double>(c) : safe_scaling_dim<1, float>(c); break;
	case 2: f64 ? safe_scaling_dim<2, double>(c) : safe_scaling_dim<2, float>(c); break;
	default: f64 ? safe_scaling_dim<3, double>(c) : safe_scaling_dim<3, float>(c); break;
	}
}

void operator_prepare(fi_ctx* c, bool with_scaling, const float* lump_in)
{
	const bool f64 = c->dtype == FI_F64;
	c->dinv16s_valid = false;
	switch (c->g.ndim) {
	case 1: f64 ? prepare_dim<1, double>(c, with_scaling, lump_in) : prepare_dim<1, float>(c, with_scaling, lump_in); break;
	case 2: f64 ? prepare_dim<2, double>(c, with_scaling, lump_in) : prepare_dim<2, float>(c, with_scaling, lump_in); break;
	default: f64 ? prepare_dim<3, double>(c, with_scaling, lump_in) : prepare_dim<3, float>(c, with_scaling, lump_in); break;
	}
	if (with_scaling && !c->dinv16s_valid) { prepare_safe_scaling(c); }
}

void apply_AtA(fi_ctx* c, const void* x, void* y, double* partial)
{
	const bool f64 = c->dtype == FI_F64;
	switch (c->g.ndim) {
	case 1:
		f64 ? apply_dim<1, double>(c, static_cast<const double*>(x), static_cast<double*>(y), partial)
		    : apply_dim<1, float>(c, static_cast<const float*>(x), static_cast<float*>(y), partial);
		break;
	case 2:
		f64 ? apply_dim<2, double>(c, static_cast<const double*>(x), static_cast<double*>(y), partial)
		    : apply_dim<2, float>(c, static_cast<const float*>(x), static_cast<float*>(y), partial);
		break;
	default:
		f64 ? apply_dim<3, double>(c, static_cast<const double*>(x), static_cast<double*>(y), partial)
		    : apply_dim<3, float>(c, static_cast<const float*>(x), static_cast<float*>(y), partial);
		break;
	}
}

}  // namespace fi
