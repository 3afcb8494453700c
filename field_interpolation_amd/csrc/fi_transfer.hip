// fi_transfer.hip -- interpolation and restriction between the levels of a hierarchy (R = P^T; vertex- and cell-centred
// axes, slabs): the device side of upscale_field's role in the coarse-to-fine start (field_interpolation.cpp:431-485,
// src/sdf_field.cpp:272-288) and of the V-cycle's transfers.
#include "fi_solver_internal.h"

namespace fi {

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
	// Rows of whole pairs (even extent along x): a thread's two points of a row travel as ONE access, and in mode 1 the four
	// pairs it adds onto are requested HERE, before the 27 coarse loads, not after the sums (two 4-byte accesses per row with a
	// stride of two points between lanes touched every line twice, and the old values were waited for at the very end:
	// 256^3 from 128^3 43 -> ... us)
	struct alignas(2 * sizeof(T)) Pair { T a, b; };
	const bool pairs = (L.nf[0] & 1) == 0;
	Pair old4[2][2];
#pragma unroll
	for (int pz = 0; pz < 2; ++pz) {
#pragma unroll
		for (int pyb = 0; pyb < 2; ++pyb) {
			old4[pz][pyb] = Pair{T(0), T(0)};
			if (pairs && mode && live[2][pz] && live[1][pyb]) {
				const int64_t i = (static_cast<int64_t>(2 * jz + pz - L.f_base) * L.nf[1] + (2 * jy + pyb)) * L.nf[0] + 2 * jx;
				old4[pz][pyb] = *reinterpret_cast<const Pair*>(fine + i);
			}
		}
	}
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
			if (pairs) {  // (the same sums: old + acc)
				*reinterpret_cast<Pair*>(fine + i) = Pair{old4[pz][pyb].a + acc[pz][pyb][0], old4[pz][pyb].b + acc[pz][pyb][1]};
				continue;
			}
			fine[i] = mode ? fine[i] + acc[pz][pyb][0] : acc[pz][pyb][0];
			if (live[0][1]) { fine[i + 1] = mode ? fine[i + 1] + acc[pz][pyb][1] : acc[pz][pyb][1]; }
		}
	}
}
// The same interpolation on undivided lattices halved cell-centred along all three axes (even extents: the levels of
// configs 4 / 5), a thread per TWO coarse columns and R coarse rows of one coarse plane: every coarse row of the three
// planes it draws on is interpolated along x ONCE into the four fine columns of a 16-byte store, combined along z for the
// two fine planes, and a window of three such rows gives the fine rows 2 cy, 2 cy + 1 -- 9 (R + 2) loads for 16 R fine points
// where a thread per 2 x 2 x 2 block issued 27 for 8.  The weights are prolong_taps' own (the end points' 5/4, -1/4).
#ifndef FI_PRO_ROWS
#define FI_PRO_ROWS 4
#endif
constexpr int kProRows = FI_PRO_ROWS;
template <typename T>
__device__ inline void prolong_window(int f, int nc, int u0, int nslots, T* W)  // cell-centred taps of fine index f on slots u0 ..
{
	int i0, i1;
	T   w0, w1;
	prolong_taps<T>(f, nc, 1, &i0, &i1, &w0, &w1);
#pragma unroll
	for (int s = 0; s < 4; ++s) {
		if (s < nslots) { W[s] = (i0 - u0 == s ? w0 : T(0)) + (i1 - u0 == s ? w1 : T(0)); }
	}
}
template <typename T>
__global__ __launch_bounds__(kThreads) void k_prolong3_rows(LevelPair L, const T* __restrict__ coarse, T* __restrict__ fine, int mode)
{
#pragma clang fp contract(fast)  // (this file is compiled without contraction; these sums need no particular rounding)
	constexpr int R = kProRows;
	typedef T V4 __attribute__((ext_vector_type(4)));
	typedef T V2 __attribute__((ext_vector_type(2)));
	const int g   = static_cast<int>(blockIdx.x) * 64 + (threadIdx.x & 63);                     // coarse columns 2g, 2g+1
	const int cyb = (static_cast<int>(blockIdx.y) * (kThreads / 64) + (threadIdx.x >> 6)) * R;  // first coarse row (wave-uniform)
	if (2 * g >= L.nc[0] || cyb >= L.nc[1]) { return; }
	const int jz = static_cast<int>(blockIdx.z);
	T Wx[4][4];
#pragma unroll
	for (int q = 0; q < 4; ++q) { prolong_window<T>(4 * g + q, L.nc[0], 2 * g - 1, 4, Wx[q]); }
	T Wz[2][3];
#pragma unroll
	for (int pz = 0; pz < 2; ++pz) { prolong_window<T>(2 * jz + pz, L.nc[2], jz - 1, 3, Wz[pz]); }
	const int xl = 2 * g - 1 < 0 ? 0 : 2 * g - 1, xr = 2 * g + 2 > L.nc[0] - 1 ? L.nc[0] - 1 : 2 * g + 2;
	const int64_t csy = L.nc[0], csz = static_cast<int64_t>(L.nc[0]) * L.nc[1];
	const T* cpl[3];
#pragma unroll
	for (int p = 0; p < 3; ++p) {
		const int cz = jz - 1 + p;
		cpl[p] = coarse + csz * (cz < 0 ? 0 : (cz > L.nc[2] - 1 ? L.nc[2] - 1 : cz));  // (a clamped plane meets a weight of zero)
	}
	T zc[R + 2][2][4];  // rows cyb-1 .. cyb+R, interpolated along x and combined along z for the two fine planes
#pragma unroll
	for (int k = 0; k < R + 2; ++k) {
		int cy = cyb - 1 + k;
		cy = cy < 0 ? 0 : (cy > L.nc[1] - 1 ? L.nc[1] - 1 : cy);
#pragma unroll
		for (int pz = 0; pz < 2; ++pz) {
#pragma unroll
			for (int q = 0; q < 4; ++q) { zc[k][pz][q] = T(0); }
		}
#pragma unroll
		for (int p = 0; p < 3; ++p) {
			const T* row = cpl[p] + csy * cy;
			const V2 v = *reinterpret_cast<const V2*>(row + 2 * g);
			const T  a = row[xl], b = row[xr];
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const T xi = Wx[q][0] * a + Wx[q][1] * v[0] + Wx[q][2] * v[1] + Wx[q][3] * b;
				zc[k][0][q] += Wz[0][p] * xi;
				zc[k][1][q] += Wz[1][p] * xi;
			}
		}
		if (k >= 2) {  // coarse row cyb + k - 2 has its three rows: its two fine rows of both fine planes
			const int cyo = cyb + k - 2;
			if (cyo < L.nc[1]) {
#pragma unroll
				for (int py = 0; py < 2; ++py) {
					T Wy[3];
					prolong_window<T>(2 * cyo + py, L.nc[1], cyo - 1, 3, Wy);
#pragma unroll
					for (int pz = 0; pz < 2; ++pz) {
						V4 out;
#pragma unroll
						for (int q = 0; q < 4; ++q) { out[q] = Wy[0] * zc[k - 2][pz][q] + Wy[1] * zc[k - 1][pz][q] + Wy[2] * zc[k][pz][q]; }
						T* dst = fine + (static_cast<int64_t>(2 * jz + pz) * L.nf[1] + (2 * cyo + py)) * L.nf[0] + 4 * g;
						if (mode) {
							const V4 old = *reinterpret_cast<const V4*>(dst);
							out += old;
						}
						*reinterpret_cast<V4*>(dst) = out;
					}
				}
			}
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
	if ((L.nf[0] & 1) == 0) {  // rows of whole pairs: the thread's two points as one access (a pair is aligned: fx is even)
		struct alignas(2 * sizeof(T)) Pair { T a, b; };
		Pair* const dst = reinterpret_cast<Pair*>(fine + i);
		Pair v{even, odd};
		if (mode) {
			const Pair old = *dst;
			v.a += old.a;
			v.b += old.b;
		}
		*dst = v;
		return;
	}
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
template <typename T, typename TO>
__global__ __launch_bounds__(kThreads) void k_prolong3_cubic(LevelPair L, const T* __restrict__ coarse, TO* __restrict__ fine)
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
			fine[i] = static_cast<TO>(acc[pz][pyb][0]);
			if (live[0][1]) { fine[i + 1] = static_cast<TO>(acc[pz][pyb][1]); }
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
		const bool halves = L.cc[0] && L.cc[1] && L.cc[2] && L.nf[0] == 2 * L.nc[0] && L.nf[1] == 2 * L.nc[1] && L.nf[2] == 2 * L.nc[2] &&
		                    L.nc[0] % 2 == 0 && L.nc[0] >= 128 && L.nc[1] >= 64 && L.nc[2] >= 64 && L.f_base == 0 && L.c_base == 0 && L.f_z0 == 0 &&
		                    L.f_planes == L.nf[2] && L.nc[2] <= 65535 && !test_switch("FI_BLOCK_PROLONG");
		if (halves) {
			constexpr int rows_per_wg = (kThreads / 64) * kProRows;
			hipLaunchKernelGGL((k_prolong3_rows<T>), dim3((L.nc[0] / 2 + 63) / 64, (L.nc[1] + rows_per_wg - 1) / rows_per_wg, L.nc[2]), dim3(kThreads), 0,
			                   st, L, coarse, fine, mode);
		} else if (L.f_planes > 0) {
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
// The cubic interpolation of the start in the row form of k_prolong3_rows (undivided lattices halved cell-centred along all
// axes): a thread owns two coarse columns (four fine ones) and R coarse rows of one coarse plane; windows of 6 (x) / 5 (y, z)
// coarse points carry cubic_taps' weights, the ends' linear taps included.  15 (R + 4) eight-byte loads for 16 R fine points
// where a thread per 2 x 2 x 2 block issued 125 for 8 (256^3 from 128^3, fp64 result: 91 -> ... us).
template <typename T, typename TO>
__global__ __launch_bounds__(kThreads) void k_prolong3_cubic_rows(LevelPair L, const T* __restrict__ coarse, TO* __restrict__ fine)
{
#pragma clang fp contract(fast)
	constexpr int R = kProRows;
	typedef T  V2 __attribute__((ext_vector_type(2)));
	typedef TO O2 __attribute__((ext_vector_type(2)));
	const int g   = static_cast<int>(blockIdx.x) * 64 + (threadIdx.x & 63);                     // coarse columns 2g, 2g+1
	const int cyb = (static_cast<int>(blockIdx.y) * (kThreads / 64) + (threadIdx.x >> 6)) * R;  // first coarse row (wave-uniform)
	if (2 * g >= L.nc[0] || cyb >= L.nc[1]) { return; }
	const int jz = static_cast<int>(blockIdx.z);
	T Wx[4][6];  // fine column 4g+q on the coarse slots 2g-2 .. 2g+3
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		T w5[5];
		cubic_window<T>(4 * g + q, L.nc[0], 1, 2 * g + (q >> 1) - 2, true, w5);
#pragma unroll
		for (int s = 0; s < 6; ++s) { Wx[q][s] = (s - (q >> 1) >= 0 && s - (q >> 1) < 5) ? w5[s - (q >> 1) < 0 ? 0 : (s - (q >> 1) > 4 ? 4 : s - (q >> 1))] : T(0); }
	}
	T Wz[2][5];
#pragma unroll
	for (int pz = 0; pz < 2; ++pz) { cubic_window<T>(2 * jz + pz, L.nc[2], 1, jz - 2, true, Wz[pz]); }
	const int64_t csy = L.nc[0], csz = static_cast<int64_t>(L.nc[0]) * L.nc[1];
	// the three pairs of a row: whole pairs are inside the lattice or outside it (even extent); an outside pair is read at a
	// clamped place and meets weights of zero
	const int p0 = 2 * g - 2 < 0 ? 0 : 2 * g - 2, p2 = 2 * g + 2 > L.nc[0] - 2 ? L.nc[0] - 2 : 2 * g + 2;
	const T* cpl[5];
#pragma unroll
	for (int p = 0; p < 5; ++p) {
		const int cz = jz - 2 + p;
		cpl[p] = coarse + csz * (cz < 0 ? 0 : (cz > L.nc[2] - 1 ? L.nc[2] - 1 : cz));
	}
	T zc[R + 4][2][4];  // rows cyb-2 .. cyb+R+1, interpolated along x and combined along z for the two fine planes
#pragma unroll
	for (int k = 0; k < R + 4; ++k) {
		int cy = cyb - 2 + k;
		cy = cy < 0 ? 0 : (cy > L.nc[1] - 1 ? L.nc[1] - 1 : cy);
#pragma unroll
		for (int pz = 0; pz < 2; ++pz) {
#pragma unroll
			for (int q = 0; q < 4; ++q) { zc[k][pz][q] = T(0); }
		}
#pragma unroll
		for (int p = 0; p < 5; ++p) {
			const T* row = cpl[p] + csy * cy;
			const V2 a = *reinterpret_cast<const V2*>(row + p0), v = *reinterpret_cast<const V2*>(row + 2 * g), b = *reinterpret_cast<const V2*>(row + p2);
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const T xi = Wx[q][0] * a[0] + Wx[q][1] * a[1] + Wx[q][2] * v[0] + Wx[q][3] * v[1] + Wx[q][4] * b[0] + Wx[q][5] * b[1];
				zc[k][0][q] += Wz[0][p] * xi;
				zc[k][1][q] += Wz[1][p] * xi;
			}
		}
		if (k >= 4) {  // coarse row cyb + k - 4 has its five rows
			const int cyo = cyb + k - 4;
			if (cyo < L.nc[1]) {
#pragma unroll
				for (int py = 0; py < 2; ++py) {
					T Wy[5];
					cubic_window<T>(2 * cyo + py, L.nc[1], 1, cyo - 2, true, Wy);
#pragma unroll
					for (int pz = 0; pz < 2; ++pz) {
						T out[4];
#pragma unroll
						for (int q = 0; q < 4; ++q) {
							out[q] = Wy[0] * zc[k - 4][pz][q] + Wy[1] * zc[k - 3][pz][q] + Wy[2] * zc[k - 2][pz][q] + Wy[3] * zc[k - 1][pz][q] + Wy[4] * zc[k][pz][q];
						}
						TO* dst = fine + (static_cast<int64_t>(2 * jz + pz) * L.nf[1] + (2 * cyo + py)) * L.nf[0] + 4 * g;
						*reinterpret_cast<O2*>(dst)     = O2{static_cast<TO>(out[0]), static_cast<TO>(out[1])};
						*reinterpret_cast<O2*>(dst + 2) = O2{static_cast<TO>(out[2]), static_cast<TO>(out[3])};
					}
				}
			}
		}
	}
}

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
// The x / y pass on lattices halved cell-centred along x and y (even extents: 256^3 .. 8^3 of configs 4 / 5), without LDS
// and without a barrier: a thread owns TWO coarse columns -- the four fine points of one 16-byte load, plus its two
// neighbours' edge points through the cache -- and walks 2 R + 4 fine rows for R coarse rows of them, every row restricted
// along x once and spread over the (at most three) coarse rows it belongs to.  The weights are restrict_taps' own (the end
// rows / columns with their 5/4 and -1/4), dropped into windows of six; the sums run in another order than the tiled
// kernel's (same weights).  25 LDS reads, their index arithmetic and the tile's 11 loads per thread made that one
// instruction-bound: 42 us for 84 MB at 256^3 -> 128^3.
constexpr int kRowsR = 8;
template <typename T>
__device__ inline void restrict_window6(int c, int nf, int nc, int base, T* W)
{
	int f[kRTaps];
	T   w[kRTaps];
	restrict_taps<T>(c, nf, nc, 1, base, f, w);
#pragma unroll
	for (int s = 0; s < 6; ++s) {
		T acc = T(0);
#pragma unroll
		for (int k = 0; k < kRTaps; ++k) { acc += (f[k] == s) ? w[k] : T(0); }
		W[s] = acc;
	}
}
template <typename T>
__global__ __launch_bounds__(kThreads) void k_restrict3_xy_rows(LevelPair L, const T* __restrict__ fine, T* __restrict__ tmp)
{
#pragma clang fp contract(fast)  // (this file is compiled without contraction; these sums need no particular rounding)
	constexpr int R = kRowsR;
	typedef T V4 __attribute__((ext_vector_type(4)));
	typedef T V2 __attribute__((ext_vector_type(2)));
	const int g   = static_cast<int>(blockIdx.x) * 64 + (threadIdx.x & 63);             // fine columns 4g .. 4g+3
	const int cyb = (static_cast<int>(blockIdx.y) * (kThreads / 64) + (threadIdx.x >> 6)) * R;  // first coarse row (wave-uniform)
	if (4 * g >= L.nf[0] || cyb >= L.nc[1]) { return; }
	const int fz = static_cast<int>(blockIdx.z);
	T We[6], Wo[6];
	restrict_window6<T>(2 * g, L.nf[0], L.nc[0], 4 * g - 1, We);
	restrict_window6<T>(2 * g + 1, L.nf[0], L.nc[0], 4 * g - 1, Wo);
	const int xl = 4 * g - 1 < 0 ? 0 : 4 * g - 1, xr = 4 * g + 4 > L.nf[0] - 1 ? L.nf[0] - 1 : 4 * g + 4;
	const T* plane = fine + static_cast<int64_t>(fz) * L.nf[0] * L.nf[1];
	T oe[R], oo[R];
#pragma unroll
	for (int j = 0; j < R; ++j) { oe[j] = T(0); oo[j] = T(0); }
	T Wy[R][6];
#pragma unroll
	for (int j = 0; j < R; ++j) {
		const int cy = cyb + j < L.nc[1] ? cyb + j : L.nc[1] - 1;
		restrict_window6<T>(cy, L.nf[1], L.nc[1], 2 * cy - 2, Wy[j]);
	}
#pragma unroll
	for (int k = 0; k < 2 * R + 4; ++k) {
		int fy = 2 * cyb - 2 + k;
		fy = fy < 0 ? 0 : (fy > L.nf[1] - 1 ? L.nf[1] - 1 : fy);  // (a clamped row meets weights of zero only)
		const T* row = plane + static_cast<int64_t>(fy) * L.nf[0];
		const V4 v = *reinterpret_cast<const V4*>(row + 4 * g);
		const T  a = row[xl], b = row[xr];
		const T e = We[0] * a + We[1] * v[0] + We[2] * v[1] + We[3] * v[2] + We[4] * v[3] + We[5] * b;
		const T o = Wo[0] * a + Wo[1] * v[0] + Wo[2] * v[1] + Wo[3] * v[2] + Wo[4] * v[3] + Wo[5] * b;
#pragma unroll
		for (int j = 0; j < R; ++j) {
			if (k - 2 * j >= 0 && k - 2 * j < 6) {  // (static)
				oe[j] += Wy[j][k - 2 * j] * e;
				oo[j] += Wy[j][k - 2 * j] * o;
			}
		}
	}
#pragma unroll
	for (int j = 0; j < R; ++j) {
		if (cyb + j < L.nc[1]) {
			*reinterpret_cast<V2*>(tmp + (static_cast<int64_t>(fz) * L.nc[1] + (cyb + j)) * L.nc[0] + 2 * g) = V2{oe[j], oo[j]};
		}
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

// the z pass with four coarse points of a row per thread (rows of whole 16-byte groups): the same sums in the same order
template <typename T>
__global__ __launch_bounds__(kThreads) void k_restrict3_z4(LevelPair L, const T* __restrict__ tmp, T* __restrict__ coarse)
{
	typedef T V4 __attribute__((ext_vector_type(4)));
	const int64_t cplane = static_cast<int64_t>(L.nc[0]) * L.nc[1], q4 = cplane / 4;
	const int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (t >= q4 * L.c_planes) { return; }
	const int64_t o  = (t % q4) * 4;
	const int     cz = static_cast<int>(t / q4) + L.c_z0;  // global plane
	int fz[kRTaps];
	T   wz[kRTaps];
	restrict_taps<T>(cz, L.nf[2], L.nc[2], L.cc[2], L.f_base, fz, wz);
	V4 acc = V4{T(0), T(0), T(0), T(0)};
#pragma unroll
	for (int k = 0; k < kRTaps; ++k) {
		if (wz[k] != T(0)) {  // (a plane without weight may lie outside the slab)
			const V4 v = *reinterpret_cast<const V4*>(tmp + fz[k] * cplane + o);
#pragma unroll
			for (int j = 0; j < 4; ++j) { acc[j] += wz[k] * v[j]; }
		}
	}
	*reinterpret_cast<V4*>(coarse + (static_cast<int64_t>(cz - L.c_base)) * cplane + o) = acc;
}

// tmp (3-D, optional): a work array of the fine level with room for (local fine planes) x nc[1] x nc[0] values -- any of
// the fine level's lattice vectors will do -- selects the two-pass form; f_local_planes = the fine level's local planes
template <typename T>
void launch_restrict(const LevelPair& L, const T* fine, T* coarse, hipStream_t st, T* tmp, int f_local_planes)
{
	if (L.ndim == 3 && tmp && f_local_planes > 0 && !test_switch("FI_ONE_PASS_RESTRICT")) {
		const int64_t n_xy = static_cast<int64_t>(L.nc[0]) * L.nc[1] * f_local_planes, n_z = static_cast<int64_t>(L.nc[0]) * L.nc[1] * L.c_planes;
		const bool halves = L.cc[0] && L.cc[1] && L.nf[0] == 2 * L.nc[0] && L.nf[1] == 2 * L.nc[1] && L.nf[0] % 4 == 0 && L.nc[0] >= 8 && L.nc[1] >= 8;
		if (halves && L.nc[0] >= 32 && f_local_planes <= 65535 && !test_switch("FI_FLAT_RESTRICT") && !test_switch("FI_TILED_RESTRICT")) {
			constexpr int rows_per_wg = (kThreads / 64) * kRowsR;
			hipLaunchKernelGGL((k_restrict3_xy_rows<T>), dim3((L.nf[0] / 4 + 63) / 64, (L.nc[1] + rows_per_wg - 1) / rows_per_wg, f_local_planes),
			                   dim3(kThreads), 0, st, L, fine, tmp);
		} else if (L.nc[0] >= 32 && f_local_planes <= 65535 && !test_switch("FI_FLAT_RESTRICT")) {
			hipLaunchKernelGGL((k_restrict3_xy_tiled<T>), dim3((L.nc[0] + 63) / 64, (L.nc[1] + 7) / 8, f_local_planes), dim3(kThreads), 0, st,
			                   L, f_local_planes, fine, tmp);
		} else {
			hipLaunchKernelGGL((k_restrict3_xy<T>), dim3(static_cast<unsigned>((n_xy + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, L,
			                   f_local_planes, fine, tmp);
		}
		if (n_z > 0 && L.nc[0] % 4 == 0 && n_z >= (1 << 18)) {
			hipLaunchKernelGGL((k_restrict3_z4<T>), dim3(static_cast<unsigned>((n_z / 4 + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, L, tmp,
			                   coarse);
		} else if (n_z > 0) {
			hipLaunchKernelGGL((k_restrict3_z<T>), dim3(static_cast<unsigned>((n_z + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, L, tmp,
			                   coarse);
		}
		return;
	}
	if (L.ndim == 3) {
		hipLaunchKernelGGL((k_restrict3<T>), owned_grid(L.nc, L.ndim, L.c_planes), dim3(kThreads), 0, st, L, fine, coarse);
	} else if (L.ndim == 2 && L.f_base == 0 && L.c_base == 0 && L.c_z0 == 0 && L.c_planes == L.nc[1] && L.f_planes == L.nf[1] &&
	           L.nc[0] >= 32 && !test_switch("FI_FLAT_RESTRICT")) {
		// an undivided 2-D lattice is ONE plane of the tiled x / y pass: coalesced loads through LDS instead of 25 loads per
		// thread with a stride of two fine points between lanes (4096^2 -> 2048^2: 61 -> ... us for 84 MB); the same taps in the
		// same order
		hipLaunchKernelGGL((k_restrict3_xy_tiled<T>), dim3((L.nc[0] + 63) / 64, (L.nc[1] + 7) / 8, 1), dim3(kThreads), 0, st, L, 1, fine, coarse);
	} else if (L.ndim == 2) {
		hipLaunchKernelGGL((k_restrict2<T>), dim3((L.nc[0] + kThreads - 1) / kThreads, L.c_planes), dim3(kThreads), 0, st, L, fine, coarse);
	} else {
		hipLaunchKernelGGL((k_restrict<T>), owned_grid(L.nc, L.ndim, L.c_planes), dim3(kThreads), 0, st, L, fine, coarse);
	}
}


// cubic interpolation of the coarse-to-fine start (k_prolong3_cubic): a thread per 2 x 2 x 2 fine points
template <typename T, typename TO>
void launch_prolong_cubic(const LevelPair& L, const T* coarse, TO* fine, hipStream_t st)
{
	const int64_t blocks8 = static_cast<int64_t>((L.nf[0] + 1) / 2) * ((L.nf[1] + 1) / 2) *
	                        (((L.f_z0 + L.f_planes - 1) >> 1) - (L.f_z0 >> 1) + 1);
	const bool halves = L.cc[0] && L.cc[1] && L.cc[2] && L.nf[0] == 2 * L.nc[0] && L.nf[1] == 2 * L.nc[1] && L.nf[2] == 2 * L.nc[2] &&
	                    L.nc[0] % 2 == 0 && L.nc[0] >= 128 && L.nc[1] >= 64 && L.nc[2] >= 64 && L.f_base == 0 && L.c_base == 0 && L.f_z0 == 0 &&
	                    L.f_planes == L.nf[2] && L.nc[2] <= 65535 && !test_switch("FI_BLOCK_PROLONG");
	if (halves) {
		constexpr int rows_per_wg = (kThreads / 64) * kProRows;
		hipLaunchKernelGGL((k_prolong3_cubic_rows<T, TO>), dim3((L.nc[0] / 2 + 63) / 64, (L.nc[1] + rows_per_wg - 1) / rows_per_wg, L.nc[2]),
		                   dim3(kThreads), 0, st, L, coarse, fine);
	} else if (L.f_planes > 0) {
		hipLaunchKernelGGL((k_prolong3_cubic<T, TO>), dim3(static_cast<unsigned>((blocks8 + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, L,
		                   coarse, fine);
	}
}

// ---- explicit instantiations (declared in fi_solver_internal.h) ----
template void launch_prolong<float>(const LevelPair&, const float*, float*, int, hipStream_t);
template void launch_prolong<double>(const LevelPair&, const double*, double*, int, hipStream_t);
template void launch_prolong_cubic<float, float>(const LevelPair&, const float*, float*, hipStream_t);
template void launch_prolong_cubic<double, double>(const LevelPair&, const double*, double*, hipStream_t);
template void launch_prolong_cubic<float, double>(const LevelPair&, const float*, double*, hipStream_t);  // the replica's start, widened on the way
template void launch_restrict<float>(const LevelPair&, const float*, float*, hipStream_t, float*, int);
template void launch_restrict<double>(const LevelPair&, const double*, double*, hipStream_t, double*, int);

}  // namespace fi
