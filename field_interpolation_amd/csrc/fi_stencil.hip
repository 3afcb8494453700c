// fi_stencil.hip -- LDS-tiled, z-marching AtA apply for 3-D lattices (the CG SpMV hot kernel).
//
// Reference path replaced: the Eigen CSC SpMV with the explicit AtA inside BiCGSTAB
// (sparse_linear.cpp:199-206 / :429-436; ~112 B per lattice point and iteration in 3-D).  Here the model
// part of AtA (rows of add_model_constraint, field_interpolation.cpp:265-280: model_1 [-1,+1] and
// model_2 [+1,-2,+1] along every axis, plus the model_0 diagonal :257-263) is applied as S^T(S x):
//     u_a = x_a - 2 x_{a+1} + x_{a+2}      (row anchored at a, exists iff 0 <= a and a+2 < size)
//     y_c += u_{c-2} - 2 u_{c-1} + u_c     (rows that touch c)
// with non-existing rows masked to zero through GLOBAL coordinates, which reproduces the reference's
// boundary rows (diag 1,5,6,...,6,5,1) on any tile / slab.  The data rows (value / gradient constraints,
// field_interpolation.cpp:57-187) enter as one symmetric 8x8 block per occupied cell (fi_assembly.hip)
// and are applied inside the same kernel.  Algorithmic traffic: read x once, write y once, read every
// block once = 2*sizeof(T) B per lattice point + (8 + 36*sizeof(T)) B per occupied cell.
//
// Work decomposition (CDNA4): one workgroup = 256 threads = a TX x 16 tile of (x, y) marching over ZC
// planes of z; a thread owns VX consecutive x (one 16-byte global load/store per plane: float4/double2).
//   * z neighbours live in registers: x(z), x(z+1), x(z+2) plus the two carried row values u(z-1), u(z-2)
//     -- each plane is read from HBM once;
//   * x/y neighbours come from an LDS copy of the plane (tile + halo ring), 3-deep ring => one barrier per
//     plane; own columns are 16-byte aligned in LDS (ds_read_b128 for the y rows);
//   * boundary masks for x/y are per-thread constants hoisted out of the march; z masks are wave-uniform;
//   * cell blocks of layer z (corners on planes z and z+1, both in the LDS ring): one thread per cell
//     multiplies its block with the 8 corner values and stores the 8 products into 8 LDS planes indexed by
//     corner -- for a fixed corner index two cells never hit the same lattice point, so there are no
//     atomics and no ordering: bitwise reproducible.  After one extra barrier the owner of a lattice point
//     adds the 4 "lower" products to its output and carries the 4 "upper" ones to the next plane in
//     registers.  Layers without data (the common case for surface point clouds) skip all of it.
//   * p.q partials: fp32 products per plane, fp64 per-thread accumulation, wave64 shuffle tree, one
//     partial per workgroup;
//   * blockIdx -> tile map is XCD-aware: blocks b, b+8, b+16.. (same XCD, same L2) get adjacent tiles.

#include <hipcub/hipcub.hpp>

#include "fi_internal.h"

namespace fi {

namespace {

constexpr int kThreads = 256;
constexpr int kTY      = 16;
constexpr int kTXT     = 16;  // threads along x
constexpr int kR       = 2;   // halo rows/cols kept in LDS

template <typename T>
struct VecOf;
template <>
struct VecOf<float> {
	using V = float4;
	static constexpr int VX = 4;
};
template <>
struct VecOf<double> {
	using V = double2;
	static constexpr int VX = 2;
};

template <typename T>
struct MarchCoef {
	T w0x3;  // 3 * model_0^2
	T w1sq;  // model_1^2
	T w2sq;  // model_2^2
};

struct CellLists {
	const uint32_t* lay_off;  // [nwg*(zc+1)+1]
	const uint2*    rec;      // {cell index, (tcx+1) | (tcy+1)<<16}
	const void*     blk;      // T[ncell][36]
};

__device__ inline double wave_sum(double v)
{
	for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o, 64); }
	return v;
}

__host__ __device__ constexpr int tri(int i, int j)  // packed upper-triangle index of an 8x8 block, i <= j
{
	return i * 8 - (i * (i - 1)) / 2 + (j - i);
}

template <typename T, bool HAS1, bool HAS2, bool CELLS>
__global__ __launch_bounds__(kThreads) void k_apply_march3d(MarchParams P, MarchCoef<T> C, CellLists L,
                                                             const T* __restrict__ x, T* __restrict__ y,
                                                             double* __restrict__ partial,
                                                             const int* __restrict__ done)
{
	using V = typename VecOf<T>::V;
	constexpr int VX   = VecOf<T>::VX;
	constexpr int TX   = kTXT * VX;
	constexpr int PADX = VX;             // own columns start 16-byte aligned
	constexpr int W    = TX + 2 * PADX;  // LDS row length
	constexpr int ROWS = kTY + 2 * kR;
	constexpr int R    = HAS2 ? 2 : 1;
	constexpr int NHALO = 2 * R * (TX + 2 * R) + 2 * R * kTY;
	constexpr int NH    = (NHALO + kThreads - 1) / kThreads;
	constexpr int NYB   = CELLS ? 8 : 1;

	__shared__ __attribute__((aligned(16))) T xs[3][ROWS][W];
	__shared__ __attribute__((aligned(16))) T yb[NYB][CELLS ? kTY : 1][CELLS ? TX : VX];
	__shared__ double red[kThreads / 64];

	if (done && *done) { return; }

	// XCD-aware tile order: consecutive tiles on one XCD.
	const int per = (P.nwg + 7) / 8;
	const int wg  = (blockIdx.x % 8) * per + blockIdx.x / 8;
	if (wg >= P.nwg) { return; }
	const int tiles_xy = P.tiles_x * P.tiles_y;
	const int chunk    = wg / tiles_xy;
	const int txy      = wg % tiles_xy;
	const int tile_y   = txy / P.tiles_x;
	const int tile_x   = txy % P.tiles_x;

	const int tx = threadIdx.x % kTXT, ty = threadIdx.x / kTXT;
	const int x0 = tile_x * TX, y0 = tile_y * kTY;
	const int gx = x0 + VX * tx, gy = y0 + ty;
	const bool active = (gx < P.nx) && (gy < P.ny);  // nx % VX == 0: a VX group is all in or all out
	const int lx = PADX + VX * tx, ly = kR + ty;

	const int z_begin = P.own_z0 + chunk * P.zc;
	int       z_end   = z_begin + P.zc;
	if (z_end > P.own_z1) { z_end = P.own_z1; }

	const int64_t col = static_cast<int64_t>(gy) * P.nx + gx;  // offset inside a plane

	// ---- per-thread constants: halo slots and x/y boundary masks --------------------------------
	int  h_lds[NH];
	int  h_glb[NH];
	bool h_ok[NH];
#pragma unroll
	for (int s = 0; s < NH; ++s) {
		const int h = threadIdx.x + s * kThreads;
		int hlx = 0, hly = 0;
		bool in = h < NHALO;
		if (h < 2 * R * (TX + 2 * R)) {
			const int r = h / (TX + 2 * R), c = h % (TX + 2 * R);
			hly = r < R ? (kR - R + r) : (kR + kTY + (r - R));
			hlx = PADX - R + c;
		} else {
			const int hh = h - 2 * R * (TX + 2 * R);
			const int r = hh / (2 * R), k = hh % (2 * R);
			hly = kR + r;
			hlx = k < R ? (PADX - R + k) : (PADX + TX + (k - R));
		}
		const int hgx = x0 + hlx - PADX, hgy = y0 + hly - kR;
		h_lds[s] = hly * W + hlx;
		h_glb[s] = hgy * P.nx + hgx;
		h_ok[s]  = in && (0 <= hgx) && (hgx < P.nx) && (0 <= hgy) && (hgy < P.ny);
		if (!in) { h_lds[s] = -1; }
	}

	// model_2 rows along x anchored at gx-2 .. gx+VX-1; along y anchored at gy-2, gy-1, gy
	T m2x[VX + 2];
	T c2y[3];
	T m1x[VX + 1];
	T c1y[2];
	if (HAS2) {
#pragma unroll
		for (int k = 0; k < VX + 2; ++k) {
			const int a = gx - 2 + k;
			m2x[k] = (a >= 0 && a + 2 < P.nx) ? T(1) : T(0);
		}
		c2y[0] = (gy - 2 >= 0 && gy < P.ny) ? T(1) : T(0);
		c2y[1] = (gy - 1 >= 0 && gy + 1 < P.ny) ? T(-2) : T(0);
		c2y[2] = (gy + 2 < P.ny) ? T(1) : T(0);
	}
	if (HAS1) {
#pragma unroll
		for (int k = 0; k < VX + 1; ++k) {
			const int a = gx - 1 + k;  // rows [-1,+1] anchored at a: x_{a+1} - x_a
			m1x[k] = (a >= 0 && a + 1 < P.nx) ? T(1) : T(0);
		}
		c1y[0] = (gy - 1 >= 0 && gy < P.ny) ? T(1) : T(0);   // row anchored at gy-1 touches gy with +1
		c1y[1] = (gy + 1 < P.ny) ? T(-1) : T(0);             // row anchored at gy touches gy with -1
	}

	auto load_own = [&](int lz) -> V {
		V v;
		T* pv = reinterpret_cast<T*>(&v);
#pragma unroll
		for (int j = 0; j < VX; ++j) { pv[j] = T(0); }
		const int gzc = lz + P.zoff;
		if (active && lz >= 0 && lz < P.nzl && gzc >= 0 && gzc < P.gz) {
			v = *reinterpret_cast<const V*>(x + static_cast<int64_t>(lz) * P.plane + col);
		}
		return v;
	};
	auto load_halo = [&](int lz, T* hv) {
		const int  gzc = lz + P.zoff;
		const bool pz  = lz >= 0 && lz < P.nzl && gzc >= 0 && gzc < P.gz;
#pragma unroll
		for (int s = 0; s < NH; ++s) {
			hv[s] = (pz && h_ok[s]) ? x[static_cast<int64_t>(lz) * P.plane + h_glb[s]] : T(0);
		}
	};
	auto write_plane = [&](int buf, const V& own, const T* hv) {
		*reinterpret_cast<V*>(&xs[buf][ly][lx]) = own;
#pragma unroll
		for (int s = 0; s < NH; ++s) {
			if (h_lds[s] >= 0) { (&xs[buf][0][0])[h_lds[s]] = hv[s]; }
		}
	};

	// ---- cell blocks of one layer: products into the 8 corner planes -----------------------------------
	const uint32_t* lay = CELLS ? L.lay_off + static_cast<int64_t>(wg) * (P.zc + 1) : nullptr;
	auto cells_scatter = [&](uint32_t rs, uint32_t re, int buf_lo, int buf_hi) {
		const T* blk = static_cast<const T*>(L.blk);
		for (uint32_t r = rs + threadIdx.x; r < re; r += kThreads) {
			const uint2 rc  = L.rec[r];
			const int   tcx = static_cast<int>(rc.y & 0xFFFFu) - 1;
			const int   tcy = static_cast<int>(rc.y >> 16) - 1;
			T b[36];
			const V* bp = reinterpret_cast<const V*>(blk + static_cast<int64_t>(rc.x) * 36);
#pragma unroll
			for (int k = 0; k < 36 / VX; ++k) {
				const V    v  = bp[k];
				const T*   pv = reinterpret_cast<const T*>(&v);
#pragma unroll
				for (int j = 0; j < VX; ++j) { b[k * VX + j] = pv[j]; }
			}
			T xv[8];
#pragma unroll
			for (int q = 0; q < 8; ++q) {
				const int bx = q & 1, by = (q >> 1) & 1, bz = q >> 2;
				xv[q] = xs[bz ? buf_hi : buf_lo][kR + tcy + by][PADX + tcx + bx];
			}
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				T s = T(0);
#pragma unroll
				for (int j = 0; j < 8; ++j) { s += b[i <= j ? tri(i, j) : tri(j, i)] * xv[j]; }
				const int px = tcx + (i & 1), py = tcy + ((i >> 1) & 1);
				if (px >= 0 && px < TX && py >= 0 && py < kTY) { yb[i][py][px] = s; }
			}
		}
	};
	// owner side: lower 4 planes -> this plane, upper 4 planes -> carry; planes are zeroed for the next layer
	auto cells_gather = [&](T* lower, T* upper) {
		const V zero = V{};
#pragma unroll
		for (int j = 0; j < VX; ++j) { lower[j] = T(0); upper[j] = T(0); }
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			V* slot = reinterpret_cast<V*>(&yb[q][ty][VX * tx]);
			const V  v  = *slot;
			const T* pv = reinterpret_cast<const T*>(&v);
#pragma unroll
			for (int j = 0; j < VX; ++j) {
				if (q < 4) { lower[j] += pv[j]; } else { upper[j] += pv[j]; }
			}
			*slot = zero;
		}
	};

	// ---- prologue ---------------------------------------------------------------------------------
	// registers at the top of step z: xc = x(z), xp1 = x(z+1), xp2 = x(z+2), xnext = x(z+3) (in flight),
	// u1 = masked u(z-1), u2 = masked u(z-2), d1 = masked (x(z) - x(z-1)), carry = upper cell products.
	V xc = load_own(z_begin), xp1 = load_own(z_begin + 1), xp2 = load_own(z_begin + 2);
	V xnext = load_own(z_begin + 3);
	T hcur[NH], hnext[NH];
	load_halo(z_begin, hcur);
	load_halo(z_begin + 1, hnext);
	T u1[VX], u2[VX], d1[VX], carry[VX];
#pragma unroll
	for (int j = 0; j < VX; ++j) { carry[j] = T(0); }
	{
		const V xa = load_own(z_begin - 2), xb = load_own(z_begin - 1);
		const T* pa = reinterpret_cast<const T*>(&xa);
		const T* pb = reinterpret_cast<const T*>(&xb);
		const T* pc = reinterpret_cast<const T*>(&xc);
		const T* pd = reinterpret_cast<const T*>(&xp1);
		const int g2 = z_begin - 2 + P.zoff, g1 = z_begin - 1 + P.zoff;
		const T mz2 = (g2 >= 0 && g2 + 2 < P.gz) ? T(1) : T(0);
		const T mz1 = (g1 >= 0 && g1 + 2 < P.gz) ? T(1) : T(0);
		const T md1 = (g1 >= 0 && g1 + 1 < P.gz) ? T(1) : T(0);
#pragma unroll
		for (int j = 0; j < VX; ++j) {
			u2[j] = mz2 * (pa[j] - T(2) * pb[j] + pc[j]);
			u1[j] = mz1 * (pb[j] - T(2) * pc[j] + pd[j]);
			d1[j] = md1 * (pc[j] - pb[j]);
		}
		if (CELLS) {
			// layer z_begin-1: its upper corners sit on plane z_begin
			const V zero = V{};
#pragma unroll
			for (int q = 0; q < 8; ++q) { *reinterpret_cast<V*>(&yb[q][ty][VX * tx]) = zero; }
			const uint32_t rs = lay[0], re = lay[1];
			if (re > rs) {
				T hprev[NH];
				load_halo(z_begin - 1, hprev);
				write_plane((z_begin + 2) % 3, xb, hprev);  // plane z_begin-1
				write_plane(z_begin % 3, xc, hcur);
				__syncthreads();
				cells_scatter(rs, re, (z_begin + 2) % 3, z_begin % 3);
				__syncthreads();
				T lower[VX];
				cells_gather(lower, carry);
				__syncthreads();  // plane z_begin-1's buffer is rewritten as plane z_begin+2 two steps on
			}
		}
	}
	write_plane(z_begin % 3, xc, hcur);

	double dot_acc = 0.0;

	for (int z = z_begin; z < z_end; ++z) {
		// stage plane z+1 into the LDS ring (needed by the cell blocks of layer z) and prefetch ahead
		write_plane((z + 1) % 3, xp1, hnext);
		load_halo(z + 2, hnext);
		const V xfar = load_own(z + 4);
		__syncthreads();

		const int gzc = z + P.zoff;
		const T (*pl)[W] = xs[z % 3];
		const T* pc  = reinterpret_cast<const T*>(&xc);
		const T* pp1 = reinterpret_cast<const T*>(&xp1);
		const T* pp2 = reinterpret_cast<const T*>(&xp2);

		uint32_t rs = 0, re = 0;
		if (CELLS) {
			rs = lay[z - z_begin + 1];
			re = lay[z - z_begin + 2];
			if (re > rs) { cells_scatter(rs, re, z % 3, (z + 1) % 3); }
		}

		T acc2[VX], acc1[VX];
#pragma unroll
		for (int j = 0; j < VX; ++j) { acc2[j] = T(0); acc1[j] = T(0); }

		// ---- x axis: window x[gx-2 .. gx+VX+1] = 2 left (LDS) + own (registers) + 2 right (LDS)
		{
			T w[VX + 4];
			w[0] = pl[ly][lx - 2];
			w[1] = pl[ly][lx - 1];
#pragma unroll
			for (int j = 0; j < VX; ++j) { w[2 + j] = pc[j]; }
			w[VX + 2] = pl[ly][lx + VX];
			w[VX + 3] = pl[ly][lx + VX + 1];
			if (HAS2) {
				T u[VX + 2];
#pragma unroll
				for (int k = 0; k < VX + 2; ++k) { u[k] = m2x[k] * (w[k] - T(2) * w[k + 1] + w[k + 2]); }
#pragma unroll
				for (int j = 0; j < VX; ++j) { acc2[j] += u[j] - T(2) * u[j + 1] + u[j + 2]; }
			}
			if (HAS1) {
				T d[VX + 1];
#pragma unroll
				for (int k = 0; k < VX + 1; ++k) { d[k] = m1x[k] * (w[k + 2] - w[k + 1]); }  // anchor gx-1+k
#pragma unroll
				for (int j = 0; j < VX; ++j) { acc1[j] += d[j] - d[j + 1]; }
			}
		}
		// ---- y axis: rows ly-2 .. ly+2 at the own columns (aligned 16-byte LDS reads)
		{
			const V r1v = *reinterpret_cast<const V*>(&pl[ly - 1][lx]);
			const V r3v = *reinterpret_cast<const V*>(&pl[ly + 1][lx]);
			const T* r1 = reinterpret_cast<const T*>(&r1v);
			const T* r3 = reinterpret_cast<const T*>(&r3v);
			if (HAS2) {
				const V r0v = *reinterpret_cast<const V*>(&pl[ly - 2][lx]);
				const V r4v = *reinterpret_cast<const V*>(&pl[ly + 2][lx]);
				const T* r0 = reinterpret_cast<const T*>(&r0v);
				const T* r4 = reinterpret_cast<const T*>(&r4v);
#pragma unroll
				for (int j = 0; j < VX; ++j) {
					const T ua = r0[j] - T(2) * r1[j] + pc[j];
					const T ub = r1[j] - T(2) * pc[j] + r3[j];
					const T uc = pc[j] - T(2) * r3[j] + r4[j];
					acc2[j] += c2y[0] * ua + c2y[1] * ub + c2y[2] * uc;
				}
			}
			if (HAS1) {
#pragma unroll
				for (int j = 0; j < VX; ++j) { acc1[j] += c1y[0] * (pc[j] - r1[j]) + c1y[1] * (r3[j] - pc[j]); }
			}
		}
		// ---- z axis: carried row values
		{
			if (HAS2) {
				const T mz = (gzc >= 0 && gzc + 2 < P.gz) ? T(1) : T(0);
#pragma unroll
				for (int j = 0; j < VX; ++j) {
					const T u0 = mz * (pc[j] - T(2) * pp1[j] + pp2[j]);
					acc2[j] += u2[j] - T(2) * u1[j] + u0;
					u2[j] = u1[j];
					u1[j] = u0;
				}
			}
			if (HAS1) {
				const T mz = (gzc + 1 < P.gz) ? T(1) : T(0);
#pragma unroll
				for (int j = 0; j < VX; ++j) {
					const T d0 = mz * (pp1[j] - pc[j]);
					acc1[j] += d1[j] - d0;
					d1[j] = d0;
				}
			}
		}

		// ---- data term: lower products of this layer + upper products carried from the layer below
		T data[VX];
#pragma unroll
		for (int j = 0; j < VX; ++j) { data[j] = carry[j]; carry[j] = T(0); }
		if (CELLS && re > rs) {
			__syncthreads();
			T lower[VX];
			cells_gather(lower, carry);
#pragma unroll
			for (int j = 0; j < VX; ++j) { data[j] += lower[j]; }
		}

		V out;
		T* po = reinterpret_cast<T*>(&out);
		T  dsum = T(0);
#pragma unroll
		for (int j = 0; j < VX; ++j) {
			T v = C.w0x3 * pc[j] + data[j];
			if (HAS2) { v += C.w2sq * acc2[j]; }
			if (HAS1) { v += C.w1sq * acc1[j]; }
			po[j] = v;
			dsum += pc[j] * v;
		}
		if (active) {
			*reinterpret_cast<V*>(y + static_cast<int64_t>(z) * P.plane + col) = out;
			dot_acc += static_cast<double>(dsum);
		}
		xc    = xp1;
		xp1   = xp2;
		xp2   = xnext;
		xnext = xfar;
	}

	if (partial) {
		const double wsum = wave_sum(dot_acc);
		if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = wsum; }
		__syncthreads();
		if (threadIdx.x == 0) { partial[wg] = red[0] + red[1] + red[2] + red[3]; }
	}
}

// ---- per-workgroup cell lists -------------------------------------------------------------------------
// A cell with global origin (cx, cy, cz) touches the tile columns {cx/TX, and (cx+1)/TX when cx+1 is a
// tile start}, likewise rows, and along z the chunk holding plane cz plus the next chunk when plane cz+1
// starts it (layer 0 of that chunk).  mode 0: count per (workgroup, layer); mode 1: fill.
__global__ __launch_bounds__(kThreads) void k_cell_lists(MarchParams P, Geom g, int64_t ncell,
                                                          const uint32_t* __restrict__ cell_id,
                                                          uint32_t* __restrict__ count,
                                                          const uint32_t* __restrict__ off,
                                                          uint32_t* __restrict__ cursor, uint2* __restrict__ rec,
                                                          int mode)
{
	const int64_t c = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (c >= ncell) { return; }
	uint32_t id = cell_id[c];
	const int cx = static_cast<int>(id % static_cast<uint32_t>(g.cn[0])) + g.coff[0];
	id /= static_cast<uint32_t>(g.cn[0]);
	const int cy = static_cast<int>(id % static_cast<uint32_t>(g.cn[1])) + g.coff[1];
	id /= static_cast<uint32_t>(g.cn[1]);
	const int cz = static_cast<int>(id) + g.coff[2];      // global
	const int zz = cz - P.zoff - P.own_z0;                // plane index relative to the first owned plane
	const int nz_own = P.own_z1 - P.own_z0;

	int tix[2], tcx[2], nx_ = 0;
	if (cx >= 0 && cx / P.tx < P.tiles_x) { tix[nx_] = cx / P.tx; tcx[nx_] = cx % P.tx; ++nx_; }
	if ((cx + 1) % P.tx == 0 && (cx + 1) / P.tx < P.tiles_x) { tix[nx_] = (cx + 1) / P.tx; tcx[nx_] = -1; ++nx_; }
	int tiy[2], tcy[2], ny_ = 0;
	if (cy >= 0 && cy / kTY < P.tiles_y) { tiy[ny_] = cy / kTY; tcy[ny_] = cy % kTY; ++ny_; }
	if ((cy + 1) % kTY == 0 && (cy + 1) / kTY < P.tiles_y) { tiy[ny_] = (cy + 1) / kTY; tcy[ny_] = -1; ++ny_; }
	int tk[2], tl[2], nz_ = 0;
	if (zz >= 0 && zz < nz_own) { tk[nz_] = zz / P.zc; tl[nz_] = zz % P.zc + 1; ++nz_; }
	if (zz + 1 >= 0 && zz + 1 < nz_own && (zz + 1) % P.zc == 0) { tk[nz_] = (zz + 1) / P.zc; tl[nz_] = 0; ++nz_; }

	for (int a = 0; a < nz_; ++a) {
		for (int b = 0; b < ny_; ++b) {
			for (int d = 0; d < nx_; ++d) {
				const int      wg     = (tk[a] * P.tiles_y + tiy[b]) * P.tiles_x + tix[d];
				const uint32_t bucket = static_cast<uint32_t>(wg) * (P.zc + 1) + tl[a];
				if (mode == 0) {
					atomicAdd(&count[bucket], 1u);
				} else {
					const uint32_t pos = off[bucket] + atomicAdd(&cursor[bucket], 1u);
					rec[pos] = make_uint2(static_cast<uint32_t>(c),
					                      static_cast<uint32_t>(tcx[d] + 1) | (static_cast<uint32_t>(tcy[b] + 1) << 16));
				}
			}
		}
	}
}

int pick_chunk(int tiles_xy, int nz_own)
{
	if (const char* env = getenv("FI_ZC")) {
		const int v = atoi(env);
		if (v > 0) { return v; }
	}
	// aim for >= 2048 workgroups (8 per CU), planes per chunk between 8 and 64
	int zc = 64;
	while (zc > 8 && static_cast<int64_t>(tiles_xy) * ((nz_own + zc - 1) / zc) < 2048) { zc /= 2; }
	return zc;
}

template <typename T>
bool march_setup(const fi_ctx* c, MarchParams* P)
{
	const Geom& g = c->g;
	constexpr int VX = VecOf<T>::VX;
	constexpr int TX = kTXT * VX;
	if (getenv("FI_NO_MARCH")) { return false; }
	if (g.ndim != 3) { return false; }
	if (g.gn[0] % VX != 0) { return false; }
	const fi_weights& w = c->w;
	if (w.model_3 > 0 || w.model_4 > 0 || w.gradient_smoothness > 0) { return false; }
	if (!(w.model_1 > 0) && !(w.model_2 > 0)) { return false; }
	P->nx = g.gn[0];
	P->ny = g.gn[1];
	P->nzl = g.n[2];
	P->gz = g.gn[2];
	P->zoff = g.off[2];
	P->own_z0 = g.own_lo[2];
	P->own_z1 = g.own_hi[2];
	P->tx      = TX;
	P->tiles_x = (P->nx + TX - 1) / TX;
	P->tiles_y = (P->ny + kTY - 1) / kTY;
	const int nz_own = P->own_z1 - P->own_z0;
	P->zc     = pick_chunk(P->tiles_x * P->tiles_y, nz_own);
	P->chunks = (nz_own + P->zc - 1) / P->zc;
	P->nwg    = P->tiles_x * P->tiles_y * P->chunks;
	P->plane  = static_cast<int64_t>(P->nx) * P->ny;
	return true;
}

template <typename T>
MarchCoef<T> march_coef(const fi_weights& w)
{
	MarchCoef<T> C;
	const T w0 = w.model_0 > 0 ? static_cast<T>(w.model_0) : T(0);
	const T w1 = w.model_1 > 0 ? static_cast<T>(w.model_1) : T(0);
	const T w2 = w.model_2 > 0 ? static_cast<T>(w.model_2) : T(0);
	C.w0x3 = T(3) * w0 * w0;
	C.w1sq = w1 * w1;
	C.w2sq = w2 * w2;
	return C;
}

template <typename T, bool CELLS>
void march_launch_cells(fi_ctx* c, const T* x, T* y, double* partial)
{
	const MarchParams& P = c->march.P;
	const MarchCoef<T> C = march_coef<T>(c->w);
	CellLists L{c->march.lay_off.as<uint32_t>(), c->march.rec.as<uint2>(), c->cells.blk.p};
	const int* done = c->scal.p ? &c->scal.as<CgScalars>()->done : nullptr;
	const int  grid = ((P.nwg + 7) / 8) * 8;
	const bool h1 = c->w.model_1 > 0, h2 = c->w.model_2 > 0;
	if (h1 && h2) {
		hipLaunchKernelGGL((k_apply_march3d<T, true, true, CELLS>), dim3(grid), dim3(kThreads), 0, c->stream, P, C, L, x,
		                   y, partial, done);
	} else if (h2) {
		hipLaunchKernelGGL((k_apply_march3d<T, false, true, CELLS>), dim3(grid), dim3(kThreads), 0, c->stream, P, C, L, x,
		                   y, partial, done);
	} else {
		hipLaunchKernelGGL((k_apply_march3d<T, true, false, CELLS>), dim3(grid), dim3(kThreads), 0, c->stream, P, C, L, x,
		                   y, partial, done);
	}
	FI_HIP_TRY(hipGetLastError());
}

void build_cell_lists(fi_ctx* c)
{
	MarchState& m = c->march;
	const MarchParams& P = m.P;
	const int64_t ncell = c->cells.ncell;
	const int64_t nbuckets = static_cast<int64_t>(P.nwg) * (P.zc + 1);
	hipStream_t st = c->stream;
	DevBuf count, cursor, tmp;
	count.alloc(sizeof(uint32_t) * (nbuckets + 1));
	cursor.alloc(sizeof(uint32_t) * (nbuckets + 1));
	m.lay_off.alloc(sizeof(uint32_t) * (nbuckets + 1));
	FI_HIP_TRY(hipMemsetAsync(count.p, 0, sizeof(uint32_t) * (nbuckets + 1), st));
	FI_HIP_TRY(hipMemsetAsync(cursor.p, 0, sizeof(uint32_t) * (nbuckets + 1), st));
	const int nb = static_cast<int>((ncell + kThreads - 1) / kThreads);
	hipLaunchKernelGGL(k_cell_lists, dim3(nb), dim3(kThreads), 0, st, P, c->g, ncell, c->cells.cell_id.as<uint32_t>(),
	                   count.as<uint32_t>(), nullptr, nullptr, nullptr, 0);
	size_t tb = 0;
	FI_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, count.as<uint32_t>(), m.lay_off.as<uint32_t>(),
	                                            static_cast<int>(nbuckets + 1), st));
	tmp.alloc(tb);
	FI_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, tb, count.as<uint32_t>(), m.lay_off.as<uint32_t>(),
	                                            static_cast<int>(nbuckets + 1), st));
	uint32_t total = 0;
	FI_HIP_TRY(hipMemcpyAsync(&total, m.lay_off.as<uint32_t>() + nbuckets, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	FI_HIP_TRY(hipStreamSynchronize(st));
	m.nrec = total;
	m.rec.alloc(sizeof(uint2) * (total ? total : 1));
	hipLaunchKernelGGL(k_cell_lists, dim3(nb), dim3(kThreads), 0, st, P, c->g, ncell, c->cells.cell_id.as<uint32_t>(),
	                   nullptr, m.lay_off.as<uint32_t>(), cursor.as<uint32_t>(), m.rec.as<uint2>(), 1);
	FI_HIP_TRY(hipGetLastError());
	FI_HIP_TRY(hipStreamSynchronize(st));
}

}  // namespace

void stencil_prepare(fi_ctx* c)
{
	MarchState& m = c->march;
	m.valid = c->dtype == FI_F64 ? march_setup<double>(c, &m.P) : march_setup<float>(c, &m.P);
	m.fused = false;
	m.nrec  = 0;
	if (!m.valid) { return; }
	if (c->cells.ncell > 0 && !getenv("FI_NO_FUSE")) {
		build_cell_lists(c);
		m.fused = true;
	}
}

// Number of p.q partials the stencil kernel writes, or 0 when the generic kernel must run.
int stencil_partials(const fi_ctx* c) { return c->march.valid ? c->march.P.nwg : 0; }

bool stencil_apply(fi_ctx* c, const void* x, void* y, double* partial)
{
	if (!c->march.valid) { return false; }
	if (c->dtype == FI_F64) {
		c->march.fused ? march_launch_cells<double, true>(c, static_cast<const double*>(x), static_cast<double*>(y), partial)
		               : march_launch_cells<double, false>(c, static_cast<const double*>(x), static_cast<double*>(y), partial);
	} else {
		c->march.fused ? march_launch_cells<float, true>(c, static_cast<const float*>(x), static_cast<float*>(y), partial)
		               : march_launch_cells<float, false>(c, static_cast<const float*>(x), static_cast<float*>(y), partial);
	}
	return true;
}

}  // namespace fi
