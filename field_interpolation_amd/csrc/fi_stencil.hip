// fi_stencil.hip -- LDS-tiled, z-marching AtA apply for 3-D lattices (the CG SpMV hot kernel).
//
// Reference path replaced: the Eigen CSC SpMV with the explicit AtA inside BiCGSTAB
// (sparse_linear.cpp:199-206 / :429-436; ~112 B per lattice point and iteration in 3-D).  Here the model
// part of AtA (rows of add_model_constraint, field_interpolation.cpp:265-280: model_1 [-1,+1] and
// model_2 [+1,-2,+1] along every axis, plus the model_0 diagonal :257-263) is applied as S^T(S x):
//     u_a = x_a - 2 x_{a+1} + x_{a+2}      (row anchored at a, exists iff 0 <= a and a+2 < size)
//     y_c += u_{c-2} - 2 u_{c-1} + u_c     (rows that touch c)
// with non-existing rows masked to zero through GLOBAL coordinates, which reproduces the reference's
// boundary rows (diag 1,5,6,...,6,5,1) on any tile / slab.  Algorithmic traffic: read x once, write y
// once = 2*sizeof(T) bytes per lattice point.
//
// Work decomposition (CDNA4): one workgroup = 256 threads = a TX x 16 tile of (x, y) marching over ZC
// planes of z; a thread owns VX consecutive x (one 16-byte global load/store per plane: float4/double2).
//   * z neighbours live in registers: x(z), x(z+1), x(z+2) plus the two carried row values u(z-1), u(z-2)
//     -- each plane is read from HBM once;
//   * x/y neighbours come from an LDS copy of the plane (tile + halo ring), 3-deep ring => one barrier per
//     plane; own columns are 16-byte aligned in LDS (ds_read_b128 for the y rows);
//   * boundary masks for x/y are per-thread constants hoisted out of the march; z masks are wave-uniform;
//   * p.q partials: fp32 products per plane, fp64 per-thread accumulation, wave64 shuffle tree, one
//     partial per workgroup;
//   * blockIdx -> tile map is XCD-aware: blocks b, b+8, b+16.. (same XCD, same L2) get adjacent tiles.
// Data term (per-cell blocks): see k_apply_cells in fi_operator.hip (fused variant: section "cells").

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

struct MarchParams {
	int     nx, ny;          // lattice extent in x and y
	int     nzl;             // local planes (incl. ghost planes)
	int     gz;              // global extent of z
	int     zoff;            // global z of local plane 0
	int     own_z0, own_z1;  // owned local planes [z0, z1)
	int     tiles_x, tiles_y, chunks, zc;
	int     nwg;
	int64_t plane;  // nx * ny
};

template <typename T>
struct MarchCoef {
	T w0x3;  // 3 * model_0^2
	T w1sq;  // model_1^2
	T w2sq;  // model_2^2
};

__device__ inline double wave_sum(double v)
{
	for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o, 64); }
	return v;
}

template <typename T, bool HAS1, bool HAS2>
__global__ __launch_bounds__(kThreads) void k_apply_march3d(MarchParams P, MarchCoef<T> C, const T* __restrict__ x,
                                                             T* __restrict__ y, double* __restrict__ partial,
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

	__shared__ __attribute__((aligned(16))) T xs[3][ROWS][W];
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

	// ---- prologue ---------------------------------------------------------------------------------
	// registers at the top of step z: xm1 = x(z-1) [HAS1 only], xc = x(z), xp1 = x(z+1), xp2 = x(z+2),
	// u1 = masked u(z-1), u2 = masked u(z-2), d1 = masked (x(z) - x(z-1)).
	V xc = load_own(z_begin), xp1 = load_own(z_begin + 1), xp2 = load_own(z_begin + 2);
	V xnext = load_own(z_begin + 3);
	T hcur[NH], hnext[NH];
	load_halo(z_begin, hcur);
	load_halo(z_begin + 1, hnext);
	T u1[VX], u2[VX], d1[VX];
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

		V out;
		T* po = reinterpret_cast<T*>(&out);
		T  dsum = T(0);
#pragma unroll
		for (int j = 0; j < VX; ++j) {
			T v = C.w0x3 * pc[j];
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
bool march_setup(const fi_ctx* c, MarchParams* P, MarchCoef<T>* C)
{
	const Geom& g = c->g;
	constexpr int VX = VecOf<T>::VX;
	constexpr int TX = kTXT * VX;
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
	P->tiles_x = (P->nx + TX - 1) / TX;
	P->tiles_y = (P->ny + kTY - 1) / kTY;
	const int nz_own = P->own_z1 - P->own_z0;
	P->zc     = pick_chunk(P->tiles_x * P->tiles_y, nz_own);
	P->chunks = (nz_own + P->zc - 1) / P->zc;
	P->nwg    = P->tiles_x * P->tiles_y * P->chunks;
	P->plane  = static_cast<int64_t>(P->nx) * P->ny;
	const T w0 = w.model_0 > 0 ? static_cast<T>(w.model_0) : T(0);
	const T w1 = w.model_1 > 0 ? static_cast<T>(w.model_1) : T(0);
	const T w2 = w.model_2 > 0 ? static_cast<T>(w.model_2) : T(0);
	C->w0x3 = T(3) * w0 * w0;
	C->w1sq = w1 * w1;
	C->w2sq = w2 * w2;
	return true;
}

template <typename T>
bool march_launch(fi_ctx* c, const T* x, T* y, double* partial, int* nwg_out)
{
	MarchParams  P;
	MarchCoef<T> C;
	if (!march_setup<T>(c, &P, &C)) { return false; }
	if (nwg_out) { *nwg_out = P.nwg; }
	if (!x) { return true; }
	const int* done = c->scal.p ? &c->scal.as<CgScalars>()->done : nullptr;
	const int  grid = ((P.nwg + 7) / 8) * 8;
	const bool h1 = c->w.model_1 > 0, h2 = c->w.model_2 > 0;
	if (h1 && h2) {
		hipLaunchKernelGGL((k_apply_march3d<T, true, true>), dim3(grid), dim3(kThreads), 0, c->stream, P, C, x, y,
		                   partial, done);
	} else if (h2) {
		hipLaunchKernelGGL((k_apply_march3d<T, false, true>), dim3(grid), dim3(kThreads), 0, c->stream, P, C, x, y,
		                   partial, done);
	} else {
		hipLaunchKernelGGL((k_apply_march3d<T, true, false>), dim3(grid), dim3(kThreads), 0, c->stream, P, C, x, y,
		                   partial, done);
	}
	FI_HIP_TRY(hipGetLastError());
	return true;
}

}  // namespace

// Returns the number of p.q partials the stencil kernel writes, or 0 when the generic kernel must run.
int stencil_partials(const fi_ctx* c)
{
	if (getenv("FI_NO_MARCH")) { return 0; }
	int n = 0;
	const bool ok = c->dtype == FI_F64 ? march_launch<double>(const_cast<fi_ctx*>(c), nullptr, nullptr, nullptr, &n)
	                                   : march_launch<float>(const_cast<fi_ctx*>(c), nullptr, nullptr, nullptr, &n);
	return ok ? n : 0;
}

bool stencil_apply(fi_ctx* c, const void* x, void* y, double* partial)
{
	if (getenv("FI_NO_MARCH")) { return false; }
	return c->dtype == FI_F64
	           ? march_launch<double>(c, static_cast<const double*>(x), static_cast<double*>(y), partial, nullptr)
	           : march_launch<float>(c, static_cast<const float*>(x), static_cast<float*>(y), partial, nullptr);
}

}  // namespace fi
