// fi_stencil2d.hip -- LDS-tiled AtA apply for 2-D lattices (BASELINE configs 2 and 3).
//
// Reference path replaced: the Eigen CSC SpMV with the explicit AtA (sparse_linear.cpp:199-206, :429-436),
// for lattices built by add_field_constraints (field_interpolation.cpp:265-280: model_1 [-1,+1], model_2
// [+1,-2,+1] along both axes, model_0 diagonal :257-263) plus cell-local data rows (:57-187).
// Same formulation as the 3-D kernel (fi_stencil.hip): rows applied as S^T(S x) with non-existing rows masked
// from GLOBAL coordinates; data cells as one symmetric 4x4 block per occupied cell.
//
// One workgroup = 256 threads = a TX x 16 tile (TX = 64 fp32 / 32 fp64 points; a thread owns 4 / 2 consecutive
// x: one 16-byte load and store).  The tile plus a 2-wide halo ring goes to LDS once (halo rows as 16-byte
// loads, halo columns as scalars; clamped addresses, no branches), x/y neighbours are read back from LDS.
// Data cells of the tile (origin in [x0-1, x0+TX) x [y0-1, y0+16)): one thread per cell multiplies the 4x4
// block with the 4 corner values and stores the 4 products into 4 LDS planes indexed by corner -- two cells
// never write the same slot of a plane, so there are no atomics and the result is bitwise reproducible; the
// owner of a lattice point adds its 4 slots.  x.y partials: fp64 per thread, wave64 shuffle tree, one per
// workgroup.  Algorithmic traffic: 2*sizeof(T) B per lattice point + (4 + 16*sizeof(T)) B per occupied cell.

#include "fi_prim.h"
#include "fi_sort.h"

#include "fi_internal.h"

namespace fi {

namespace {

constexpr int kThreads = 256;
constexpr int kTXT = 16;  // threads along x
constexpr int kTY  = 16;  // tile rows
constexpr int kR   = 2;

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
struct Coef2 {
	T w0x2;  // 2 * model_0^2
	T w1sq, w2sq;
};

struct CellList2 {
	const uint32_t* off;   // [ntiles + 1]
	const uint32_t* pos;   // (tcx+1) | (tcy+1) << 16
	const void*     blk;   // T[n][16] full symmetric 4x4 block
};

__device__ inline double wave_sum(double v)
{
	for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o, 64); }
	return v;
}

// Epilogue of the V-cycle's smoother (fi_solver.hip cheb_smooth_fused; the 3-D kernel's ChebEpi modes 2 and 3): instead of
// storing q = A z the kernel stores  z_new = a z - c1 z_prev + c2 Dinv (r - q)  (residual: z_new = r - q), Dinv the
// context's bfloat16 scaling.  z_prev may be z_new's own buffer (read and written by the point's owner only).
// residual = 2 (plain launches; the polynomial preconditioner / smoother of fi_solver.hip, the 3-D kernel's ChebEpi mode 0):
//     s = Dinv (q - m z) + z = Dinv (A_model + diag(A_data)) z,  m = the model diagonal from the boundary masks,
//     z_new = a z - c1 z_prev + c2 (Dinv r - s),  partials r . z_new;   zp_scale != 0: z_prev = zp_scale Dinv (vector zprev)
// residual = 3: a power-method step on the model operator, z_new = q / m, partials z_new . z_new.
template <typename T>
struct Epi2 {
	const T* zprev;
	const T* r;
	const unsigned short* dinv;
	T*       znew;
	T        a, c1, c2;
	int      residual;
	T        zp_scale;
};

template <typename T, bool HAS1, bool HAS2, bool CELLS, bool EPI = false>
__global__ __launch_bounds__(kThreads) void k_apply_tile2d(Tile2Params P, Coef2<T> C, CellList2 L,
                                                            const T* __restrict__ x, T* __restrict__ y,
                                                            double* __restrict__ partial, const int* __restrict__ done,
                                                            Epi2<T> E = Epi2<T>{})
{
	using V = typename VecOf<T>::V;
	constexpr int VX   = VecOf<T>::VX;
	constexpr int TX   = kTXT * VX;
	constexpr int PADX = VX;
	constexpr int W    = TX + 2 * PADX;
	constexpr int ROWS = kTY + 2 * kR;
	constexpr int R    = HAS2 ? 2 : 1;
	constexpr int NVEC = 2 * R * kTXT;
	constexpr int NSC  = 2 * R * (kTY + 2 * R);

	__shared__ __attribute__((aligned(16))) T xs[ROWS][W];
	__shared__ __attribute__((aligned(16))) T yb[CELLS ? 4 : 1][CELLS ? kTY : 1][CELLS ? TX : VX];
	__shared__ T ydump[CELLS ? 64 : 1];
	__shared__ double red[kThreads / 64];

	// The stop flag is REQUESTED here and looked at below, when every load of the prologue is on its way: as the first
	// statement it was a cache round trip of its own in front of everything else, and the levels of a V-cycle's lower half
	// are launches of 4-8 us that consist of three or four such trips (profiles/r5_ablation.md).
	const int stop = done ? *done : 0;
	const int per  = (P.ntiles + 7) / 8;
	const int tile = (blockIdx.x % 8) * per + blockIdx.x / 8;  // XCD-aware: neighbouring tiles share an L2
	if (tile >= P.ntiles) { return; }
	const int tile_y = tile / P.tiles_x, tile_x = tile % P.tiles_x;
	const int tx = threadIdx.x % kTXT, ty = threadIdx.x / kTXT;
	const int x0 = tile_x * TX;
	const int ly0 = P.own_y0 + tile_y * kTY;  // local row of the tile's first row
	const int gx = x0 + VX * tx;
	const int lyr = ly0 + ty;                 // local row of this thread
	const int gy = lyr + P.yoff;              // global row
	// all VX points inside (active), the partly filled last group of a row (nvalid of them, stored one by one), or outside
	const int  nvalid = lyr < P.own_y1 ? (P.nx - gx < 0 ? 0 : (P.nx - gx > VX ? VX : P.nx - gx)) : 0;
	const bool active = nvalid == VX;
	const int lx = PADX + VX * tx, ly = kR + ty;

	// clamped addresses: a wrong value is only ever multiplied by a zero mask / zero block coefficient
	const int lr_lo = P.yoff < 0 ? -P.yoff : 0;
	const int lr_hi = (P.nyl < P.gy - P.yoff ? P.nyl : P.gy - P.yoff) - 1;
	auto clamp_row = [&](int r) { return r < lr_lo ? lr_lo : (r > lr_hi ? lr_hi : r); };
	// a group that straddles the row end reads on into the next row (finite values under zero masks; 64 zeroed bytes
	// of slack follow every buffer); groups entirely outside are moved inside
	auto clamp_x = [&](int c, int width) { return c < 0 ? 0 : (c >= P.nx ? (P.nx > width ? P.nx - width : 0) : c); };

	// tiles without a single cell (most of them for surface-type data) skip the data path altogether
	uint32_t rs = 0, re = 0;
	if (CELLS) {
		rs = L.off[tile];
		re = L.off[tile + 1];
	}
	const bool has_cells = CELLS && re > rs;  // workgroup-uniform
	// operands of the epilogue (EPI), requested with the tile itself instead of after the stencil (one more round trip)
	T e_r[EPI ? VX : 1], e_zp[EPI ? VX : 1];
	unsigned short e_dv[EPI ? VX : 1];
	// rows of whole 16-byte groups (every thread's group is inside the row or outside it): the operands come and the result
	// goes as ONE 16-byte access per thread -- point by point they were four instructions that each touched a quarter of
	// every cache line of the row (the smoother's steps on a 4096^2 level: 84 -> ... us)
	const bool rowvec = EPI && P.nx % VX == 0;
	if (EPI && rowvec) {
		const int64_t i = static_cast<int64_t>(lyr < P.own_y1 ? lyr : P.own_y1 - 1) * P.nx + (gx + VX <= P.nx ? gx : P.nx - VX);
		const bool full = E.residual == 0 || E.residual == 2;
		V vr = V{}, vz = V{};
		if (E.residual != 3) { vr = *reinterpret_cast<const V*>(E.r + i); }
		if (full) { vz = *reinterpret_cast<const V*>(E.zprev + i); }
		const T* pr = reinterpret_cast<const T*>(&vr);
		const T* pz = reinterpret_cast<const T*>(&vz);
		if (VX == 4) {
			uint2 d = uint2{0u, 0u};
			if (full) { d = *reinterpret_cast<const uint2*>(E.dinv + i); }
			e_dv[0] = static_cast<unsigned short>(d.x & 0xFFFFu);
			e_dv[1] = static_cast<unsigned short>(d.x >> 16);
			e_dv[VX > 2 ? 2 : 0] = static_cast<unsigned short>(d.y & 0xFFFFu);
			e_dv[VX > 2 ? 3 : 1] = static_cast<unsigned short>(d.y >> 16);
		} else {
			unsigned int d = 0u;
			if (full) { d = *reinterpret_cast<const unsigned int*>(E.dinv + i); }
			e_dv[0] = static_cast<unsigned short>(d & 0xFFFFu);
			e_dv[1] = static_cast<unsigned short>(d >> 16);
		}
#pragma unroll
		for (int j = 0; j < VX; ++j) {
			e_r[j]  = pr[j];
			e_zp[j] = pz[j];
		}
	} else if (EPI) {
#pragma unroll
		for (int j = 0; j < VX; ++j) {
			const int64_t i = static_cast<int64_t>(lyr < P.own_y1 ? lyr : P.own_y1 - 1) * P.nx + (gx + j < P.nx ? gx + j : P.nx - 1);
			e_r[j]  = E.residual != 3 ? E.r[i] : T(0);
			e_zp[j] = (E.residual == 0 || E.residual == 2) ? E.zprev[i] : T(0);
			e_dv[j] = (E.residual == 0 || E.residual == 2) ? E.dinv[i] : static_cast<unsigned short>(0);
		}
	}

	const V own = *reinterpret_cast<const V*>(x + static_cast<int64_t>(clamp_row(lyr)) * P.nx + clamp_x(gx, VX));
	*reinterpret_cast<V*>(&xs[ly][lx]) = own;
	if (threadIdx.x < NVEC) {
		const int hrow = threadIdx.x / kTXT, vx = threadIdx.x % kTXT;
		const int hly  = hrow < R ? (kR - R + hrow) : (kR + kTY + (hrow - R));
		const V v = *reinterpret_cast<const V*>(x + static_cast<int64_t>(clamp_row(ly0 + hly - kR)) * P.nx +
		                                         clamp_x(x0 + VX * vx, VX));
		*reinterpret_cast<V*>(&xs[hly][PADX + VX * vx]) = v;
	} else if (threadIdx.x < NVEC + NSC) {
		const int u = threadIdx.x - NVEC;
		const int row = u / (2 * R), k = u % (2 * R);
		const int hly = kR - R + row;
		const int hlx = k < R ? (PADX - R + k) : (PADX + TX + (k - R));
		xs[hly][hlx] = x[static_cast<int64_t>(clamp_row(ly0 + hly - kR)) * P.nx + clamp_x(x0 + hlx - PADX, 1)];
	}
	if (has_cells) {
		const V zero = V{};
#pragma unroll
		for (int q = 0; q < 4; ++q) { *reinterpret_cast<V*>(&yb[q][ty][VX * tx]) = zero; }
	}
	if (stop) { return; }  // (workgroup-uniform; nothing has been stored to memory yet)
	__syncthreads();

	// ---- data cells of this tile ---------------------------------------------------------------------
	if (has_cells) {
		const T* blk = static_cast<const T*>(L.blk);
		for (uint32_t r = rs + threadIdx.x; r < re; r += kThreads) {
			const uint32_t pos = L.pos[r];
			const int tcx = static_cast<int>(pos & 0xFFFFu) - 1, tcy = static_cast<int>(pos >> 16) - 1;
			T b[16];
			const V* bp = reinterpret_cast<const V*>(blk + static_cast<int64_t>(r) * 16);
#pragma unroll
			for (int k = 0; k < 16 / VX; ++k) {
				const V  v  = bp[k];
				const T* pv = reinterpret_cast<const T*>(&v);
#pragma unroll
				for (int j = 0; j < VX; ++j) { b[k * VX + j] = pv[j]; }
			}
			T xv[4];
#pragma unroll
			for (int q = 0; q < 4; ++q) { xv[q] = xs[kR + tcy + (q >> 1)][PADX + tcx + (q & 1)]; }
			T* const dump = &ydump[threadIdx.x & 63];
			const bool vx0 = tcx >= 0, vx1 = tcx + 1 < TX, vy0 = tcy >= 0, vy1 = tcy + 1 < kTY;
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				T s = T(0);
#pragma unroll
				for (int j = 0; j < 4; ++j) { s += b[i * 4 + j] * xv[j]; }
				const bool ok = ((i & 1) ? vx1 : vx0) && ((i & 2) ? vy1 : vy0);
				T* dst = &yb[0][0][0] + i * (kTY * TX) + (tcy + (i >> 1)) * TX + tcx + (i & 1);
				dst = ok ? dst : dump;
				*dst = s;
			}
		}
		__syncthreads();
	}

	// ---- stencil --------------------------------------------------------------------------------------
	const T* pc = reinterpret_cast<const T*>(&own);
	T acc2[VX], acc1[VX];
#pragma unroll
	for (int j = 0; j < VX; ++j) { acc2[j] = T(0); acc1[j] = T(0); }
	{
		T w[VX + 4];
		w[0] = xs[ly][lx - 2];
		w[1] = xs[ly][lx - 1];
#pragma unroll
		for (int j = 0; j < VX; ++j) { w[2 + j] = pc[j]; }
		w[VX + 2] = xs[ly][lx + VX];
		w[VX + 3] = xs[ly][lx + VX + 1];
		if (HAS2) {
			T u[VX + 2];
#pragma unroll
			for (int k = 0; k < VX + 2; ++k) {
				const int a = gx - 2 + k;
				const T m = (a >= 0 && a + 2 < P.nx) ? T(1) : T(0);
				u[k] = m * (w[k] - T(2) * w[k + 1] + w[k + 2]);
			}
#pragma unroll
			for (int j = 0; j < VX; ++j) { acc2[j] += u[j] - T(2) * u[j + 1] + u[j + 2]; }
		}
		if (HAS1) {
			T d[VX + 1];
#pragma unroll
			for (int k = 0; k < VX + 1; ++k) {
				const int a = gx - 1 + k;
				const T m = (a >= 0 && a + 1 < P.nx) ? T(1) : T(0);
				d[k] = m * (w[k + 2] - w[k + 1]);
			}
#pragma unroll
			for (int j = 0; j < VX; ++j) { acc1[j] += d[j] - d[j + 1]; }
		}
	}
	{
		const V r1v = *reinterpret_cast<const V*>(&xs[ly - 1][lx]);
		const V r3v = *reinterpret_cast<const V*>(&xs[ly + 1][lx]);
		const T* r1 = reinterpret_cast<const T*>(&r1v);
		const T* r3 = reinterpret_cast<const T*>(&r3v);
		if (HAS2) {
			const V r0v = *reinterpret_cast<const V*>(&xs[ly - 2][lx]);
			const V r4v = *reinterpret_cast<const V*>(&xs[ly + 2][lx]);
			const T* r0 = reinterpret_cast<const T*>(&r0v);
			const T* r4 = reinterpret_cast<const T*>(&r4v);
			const T c0 = (gy - 2 >= 0 && gy < P.gy) ? T(1) : T(0);
			const T c1 = (gy - 1 >= 0 && gy + 1 < P.gy) ? T(-2) : T(0);
			const T c2 = (gy + 2 < P.gy) ? T(1) : T(0);
#pragma unroll
			for (int j = 0; j < VX; ++j) {
				const T ua = r0[j] - T(2) * r1[j] + pc[j];
				const T ub = r1[j] - T(2) * pc[j] + r3[j];
				const T uc = pc[j] - T(2) * r3[j] + r4[j];
				acc2[j] += c0 * ua + c1 * ub + c2 * uc;
			}
		}
		if (HAS1) {
			const T c0 = (gy - 1 >= 0 && gy < P.gy) ? T(1) : T(0);
			const T c1 = (gy + 1 < P.gy) ? T(-1) : T(0);
#pragma unroll
			for (int j = 0; j < VX; ++j) { acc1[j] += c0 * (pc[j] - r1[j]) + c1 * (r3[j] - pc[j]); }
		}
	}
	V out;
	T* po = reinterpret_cast<T*>(&out);
	T  dsum = T(0);
#pragma unroll
	for (int j = 0; j < VX; ++j) {
		T v = C.w0x2 * pc[j];
		if (HAS2) { v += C.w2sq * acc2[j]; }
		if (HAS1) { v += C.w1sq * acc1[j]; }
		po[j] = v;
	}
	if (has_cells) {
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const V  v  = *reinterpret_cast<const V*>(&yb[q][ty][VX * tx]);
			const T* pv = reinterpret_cast<const T*>(&v);
#pragma unroll
			for (int j = 0; j < VX; ++j) { po[j] += pv[j]; }
		}
	}
	if (EPI) {
		T* const dst = E.znew + static_cast<int64_t>(lyr) * P.nx + gx;
		// the model diagonal along y (the same for the thread's VX points), for the polynomial's modes
		T my = T(0);
		if (E.residual >= 2) {
			if (HAS2) {
				my += C.w2sq * (((gy - 2 >= 0 && gy < P.gy) ? T(1) : T(0)) + ((gy - 1 >= 0 && gy + 1 < P.gy) ? T(4) : T(0)) +
				                ((gy >= 0 && gy + 2 < P.gy) ? T(1) : T(0)));
			}
			if (HAS1) { my += C.w1sq * (((gy - 1 >= 0 && gy < P.gy) ? T(1) : T(0)) + ((gy >= 0 && gy + 1 < P.gy) ? T(1) : T(0))); }
			my += C.w0x2;
		}
		T part = T(0);
		V znv = V{};
		T* pzn = reinterpret_cast<T*>(&znv);
		for (int j = 0; j < VX; ++j) {  // (point by point where rows are not 16-byte multiples)
			if (j < nvalid) {
				T zn;
				if (E.residual == 1) {
					zn = e_r[j] - po[j];
				} else if (E.residual == 0) {
					const T dv = static_cast<T>(__uint_as_float(static_cast<unsigned int>(e_dv[j]) << 16));
					zn = E.a * pc[j] - E.c1 * e_zp[j] + E.c2 * (dv * (e_r[j] - po[j]));
				} else {
					const int g = gx + j;
					T m = my;
					if (HAS2) {
						m += C.w2sq * (((g - 2 >= 0 && g < P.nx) ? T(1) : T(0)) + ((g - 1 >= 0 && g + 1 < P.nx) ? T(4) : T(0)) +
						               ((g + 2 < P.nx) ? T(1) : T(0)));
					}
					if (HAS1) { m += C.w1sq * (((g - 1 >= 0) ? T(1) : T(0)) + ((g + 1 < P.nx) ? T(1) : T(0))); }
					if (E.residual == 3) {
						zn = po[j] / m;
						part += zn * zn;
					} else {
						const T dv = static_cast<T>(__uint_as_float(static_cast<unsigned int>(e_dv[j]) << 16));
						const T rv = e_r[j];
						const T sv = dv * (po[j] - m * pc[j]) + pc[j];
						const T zq = E.zp_scale != T(0) ? E.zp_scale * dv * e_zp[j] : e_zp[j];
						zn = E.a * pc[j] - E.c1 * zq + E.c2 * (dv * rv - sv);
						part += rv * zn;
					}
				}
				if (rowvec) { pzn[j] = zn; } else { dst[j] = zn; }
			}
		}
		if (rowvec && nvalid == VX) { *reinterpret_cast<V*>(dst) = znv; }
		if (E.residual >= 2 && partial) {
			const double wsum = wave_sum(static_cast<double>(part));
			if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = wsum; }
			__syncthreads();
			if (threadIdx.x == 0) { partial[tile] = red[0] + red[1] + red[2] + red[3]; }
		}
		return;  // (modes 0 and 1 write no partials: the smoother takes no dot products)
	}
#pragma unroll
	for (int j = 0; j < VX; ++j) { dsum += pc[j] * po[j]; }
	double dot = 0.0;
	if (active) {
		*reinterpret_cast<V*>(y + static_cast<int64_t>(lyr) * P.nx + gx) = out;
		dot = static_cast<double>(dsum);
	} else if (nvalid > 0) {
		T part = T(0);
#pragma unroll
		for (int j = 0; j < VX - 1; ++j) {
			if (j < nvalid) {
				y[static_cast<int64_t>(lyr) * P.nx + gx + j] = po[j];
				part += pc[j] * po[j];
			}
		}
		dot = static_cast<double>(part);
	}
	if (partial) {
		const double wsum = wave_sum(dot);
		if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = wsum; }
		__syncthreads();
		if (threadIdx.x == 0) { partial[tile] = red[0] + red[1] + red[2] + red[3]; }
	}
}

// ---- per-tile cell lists: membership slots, radix sort by tile, records with the full 4x4 block ------------
constexpr uint32_t kNoKey = 0xFFFFFFFFu;

__global__ __launch_bounds__(kThreads) void k_cell_members2(Tile2Params P, Geom g, int64_t ncell,
                                                             const uint32_t* __restrict__ cell_id,
                                                             uint32_t* __restrict__ key, uint32_t* __restrict__ pos,
                                                             uint32_t* __restrict__ count)
{
	const int64_t c = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (c >= ncell) { return; }
	uint32_t id = cell_id[c];
	const int cx = static_cast<int>(id % static_cast<uint32_t>(g.cn[0])) + g.coff[0];
	id /= static_cast<uint32_t>(g.cn[0]);
	const int cy = static_cast<int>(id) + g.coff[1];      // global row of the cell origin
	const int yy = cy - P.yoff - P.own_y0;                // relative to the first owned row
	const int ny_own = P.own_y1 - P.own_y0;
	const bool x_ok[2] = {cx >= 0 && cx / P.tx < P.tiles_x, (cx + 1) % P.tx == 0 && (cx + 1) / P.tx < P.tiles_x};
	const int  x_ti[2] = {cx >= 0 ? cx / P.tx : 0, (cx + 1) / P.tx};
	const int  x_tc[2] = {cx >= 0 ? cx % P.tx : 0, -1};
	const bool y_ok[2] = {yy >= 0 && yy < ny_own, yy + 1 >= 0 && yy + 1 < ny_own && (yy + 1) % kTY == 0};
	const int  y_ti[2] = {yy >= 0 ? yy / kTY : 0, (yy + 1) / kTY};
	const int  y_tc[2] = {yy >= 0 ? yy % kTY : 0, -1};
#pragma unroll
	for (int m = 0; m < 4; ++m) {
		const int b = m >> 1, d = m & 1;
		uint32_t k = kNoKey, pp = 0;
		if (y_ok[b] && x_ok[d]) {
			k  = static_cast<uint32_t>(y_ti[b] * P.tiles_x + x_ti[d]);
			pp = static_cast<uint32_t>(x_tc[d] + 1) | (static_cast<uint32_t>(y_tc[b] + 1) << 16);
			atomicAdd(&count[k], 1u);
		}
		key[c * 4 + m] = k;
		pos[c * 4 + m] = pp;
	}
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_cell_records2(int64_t n, const uint32_t* __restrict__ slot_sorted,
                                                             const uint32_t* __restrict__ pos,
                                                             const T* __restrict__ blk10, uint32_t* __restrict__ pos_out,
                                                             T* __restrict__ blk16)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (i >= n) { return; }
	const uint32_t slot = slot_sorted[i];
	const int64_t  c    = slot >> 2;
	pos_out[i] = pos[slot];
	// packed upper triangle (10) -> full symmetric 4x4
	for (int a = 0; a < 4; ++a) {
		for (int b = 0; b < 4; ++b) {
			const int lo = a < b ? a : b, hi = a < b ? b : a;
			blk16[i * 16 + a * 4 + b] = blk10[c * 10 + (lo * 4 - (lo * (lo - 1)) / 2 + (hi - lo))];
		}
	}
}

__global__ void k_iota2(uint32_t* v, int64_t n)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i < n) { v[i] = static_cast<uint32_t>(i); }
}

template <typename T>
bool tile2_setup(const fi_ctx* c, Tile2Params* P)
{
	const Geom& g = c->g;
	constexpr int VX = VecOf<T>::VX;
	constexpr int TX = kTXT * VX;
	if (test_switch("FI_NO_TILE2D")) { return false; }
	if (g.ndim != 2 || g.gn[0] < VX) { return false; }
	const fi_weights& w = c->w;
	if (w.model_3 > 0 || w.model_4 > 0 || w.gradient_smoothness > 0) { return false; }
	if (!(w.model_1 > 0) && !(w.model_2 > 0)) { return false; }
	P->nx = g.gn[0];
	P->nyl = g.n[1];
	P->gy = g.gn[1];
	P->yoff = g.off[1];
	P->own_y0 = g.own_lo[1];
	P->own_y1 = g.own_hi[1];
	P->tx = TX;
	P->tiles_x = (P->nx + TX - 1) / TX;
	P->tiles_y = (P->own_y1 - P->own_y0 + kTY - 1) / kTY;
	P->ntiles = P->tiles_x * P->tiles_y;
	return true;
}

template <typename T>
void build_lists2(fi_ctx* c)
{
	Tile2State& m = c->tile2;
	const Tile2Params& P = m.P;
	const int64_t ncell = c->cells.ncell, nslots = ncell * 4;
	hipStream_t st = c->stream;
	DevBuf &count = c->scratch[14], &key = c->scratch[15], &pos = c->scratch[16], &slot_in = c->scratch[17],
	       &key_sorted = c->scratch[18], &slot_sorted = c->scratch[19], &tmp = c->scratch[20];
	count.alloc(sizeof(uint32_t) * (P.ntiles + 2));
	key.alloc(sizeof(uint32_t) * nslots);
	pos.alloc(sizeof(uint32_t) * nslots);
	slot_in.alloc(sizeof(uint32_t) * nslots);
	key_sorted.alloc(sizeof(uint32_t) * nslots);
	slot_sorted.alloc(sizeof(uint32_t) * nslots);
	m.off.alloc(sizeof(uint32_t) * (P.ntiles + 2));
	FI_HIP_TRY(hipMemsetAsync(count.p, 0, sizeof(uint32_t) * (P.ntiles + 2), st));
	const int nb = static_cast<int>((ncell + kThreads - 1) / kThreads);
	hipLaunchKernelGGL(k_cell_members2, dim3(nb), dim3(kThreads), 0, st, P, c->g, ncell, c->cells.cell_id.as<uint32_t>(),
	                   key.as<uint32_t>(), pos.as<uint32_t>(), count.as<uint32_t>());
	hipLaunchKernelGGL(k_iota2, dim3(static_cast<int>((nslots + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
	                   slot_in.as<uint32_t>(), nslots);
	size_t tb = 0, tb2 = 0;
	const int key_bits = 32;  // (slots outside every tile carry the key 0xFFFFFFFF: they sort to the end)
	FI_HIP_TRY(sort_pairs_u32(nullptr, tb, key.as<uint32_t>(), key_sorted.as<uint32_t>(), slot_in.as<uint32_t>(), slot_sorted.as<uint32_t>(),
	                          static_cast<unsigned int>(nslots), 0, key_bits, st));
	FI_HIP_TRY(prim::exclusive_sum(nullptr, tb2, count.as<uint32_t>(), m.off.as<uint32_t>(), static_cast<size_t>(P.ntiles + 1), st));
	tmp.alloc(tb > tb2 ? tb : tb2);
	FI_HIP_TRY(sort_pairs_u32(tmp.p, tb, key.as<uint32_t>(), key_sorted.as<uint32_t>(), slot_in.as<uint32_t>(), slot_sorted.as<uint32_t>(),
	                          static_cast<unsigned int>(nslots), 0, key_bits, st));
	FI_HIP_TRY(prim::exclusive_sum(tmp.p, tb2, count.as<uint32_t>(), m.off.as<uint32_t>(), static_cast<size_t>(P.ntiles + 1), st));
	uint32_t total = 0;
	FI_HIP_TRY(hipMemcpyAsync(&total, m.off.as<uint32_t>() + P.ntiles, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	FI_HIP_TRY(hipStreamSynchronize(st));
	m.nrec = total;
	m.pos.alloc(sizeof(uint32_t) * (total + 1));
	m.blk.alloc(sizeof(T) * 16 * (total + 1));
	if (total > 0) {
		hipLaunchKernelGGL((k_cell_records2<T>), dim3(static_cast<int>((total + kThreads - 1) / kThreads)), dim3(kThreads), 0,
		                   st, static_cast<int64_t>(total), slot_sorted.as<uint32_t>(), pos.as<uint32_t>(),
		                   c->cells.blk.as<T>(), m.pos.as<uint32_t>(), m.blk.as<T>());
	}
	FI_HIP_TRY(hipGetLastError());
}

template <typename T, bool CELLS>
void tile2_launch(fi_ctx* c, const T* x, T* y, double* partial, const Epi2<T>* epi = nullptr)
{
	const Tile2State& m = c->tile2;
	Coef2<T> C;
	const fi_weights& w = c->w;
	const T w0 = w.model_0 > 0 ? static_cast<T>(w.model_0) : T(0);
	const T w1 = w.model_1 > 0 ? static_cast<T>(w.model_1) : T(0);
	const T w2 = w.model_2 > 0 ? static_cast<T>(w.model_2) : T(0);
	C.w0x2 = T(2) * w0 * w0;
	C.w1sq = w1 * w1;
	C.w2sq = w2 * w2;
	CellList2 L{m.off.as<uint32_t>(), m.pos.as<uint32_t>(), m.blk.p};
	const int* done = c->scal.p ? &c->scal.as<CgScalars>()->done : nullptr;
	const int  grid = ((m.P.ntiles + 7) / 8) * 8;
	const bool h1 = w.model_1 > 0, h2 = w.model_2 > 0;
	if (epi) {
		if (h1 && h2) {
			hipLaunchKernelGGL((k_apply_tile2d<T, true, true, CELLS, true>), dim3(grid), dim3(kThreads), 0, c->stream, m.P, C, L, x, y,
			                   partial, done, *epi);
		} else if (h2) {
			hipLaunchKernelGGL((k_apply_tile2d<T, false, true, CELLS, true>), dim3(grid), dim3(kThreads), 0, c->stream, m.P, C, L, x, y,
			                   partial, done, *epi);
		} else {
			hipLaunchKernelGGL((k_apply_tile2d<T, true, false, CELLS, true>), dim3(grid), dim3(kThreads), 0, c->stream, m.P, C, L, x, y,
			                   partial, done, *epi);
		}
	} else if (h1 && h2) {
		hipLaunchKernelGGL((k_apply_tile2d<T, true, true, CELLS>), dim3(grid), dim3(kThreads), 0, c->stream, m.P, C, L, x, y,
		                   partial, done, Epi2<T>{});
	} else if (h2) {
		hipLaunchKernelGGL((k_apply_tile2d<T, false, true, CELLS>), dim3(grid), dim3(kThreads), 0, c->stream, m.P, C, L, x, y,
		                   partial, done, Epi2<T>{});
	} else {
		hipLaunchKernelGGL((k_apply_tile2d<T, true, false, CELLS>), dim3(grid), dim3(kThreads), 0, c->stream, m.P, C, L, x, y,
		                   partial, done, Epi2<T>{});
	}
	FI_HIP_TRY(hipGetLastError());
}

}  // namespace

void tile2d_prepare(fi_ctx* c)
{
	Tile2State& m = c->tile2;
	m.valid = c->dtype == FI_F64 ? tile2_setup<double>(c, &m.P) : tile2_setup<float>(c, &m.P);
	m.fused = false;
	m.nrec  = 0;
	if (!m.valid) { return; }
	if (c->cells.ncell > 0 && !test_switch("FI_NO_FUSE")) {
		c->dtype == FI_F64 ? build_lists2<double>(c) : build_lists2<float>(c);
		m.fused = true;
	}
}

int tile2d_partials(const fi_ctx* c) { return c->tile2.valid ? c->tile2.P.ntiles : 0; }

// z_new = a z - c1 z_prev + c2 Dinv (r - A z)  (residual: z_new = r - A z) in one launch; z with valid ghost rows
bool tile2d_full_epi_available(const fi_ctx* c) { return c->tile2.valid && (c->cells.ncell == 0 || c->tile2.fused); }
template <typename T>
static void tile2_full_step_t(fi_ctx* c, const void* z, const void* zprev, const void* r, bool residual, void* znew, double a,
                              double c1, double c2)
{
	Epi2<T> E{static_cast<const T*>(zprev ? zprev : z), static_cast<const T*>(r), c->dinv16.as<unsigned short>(),
	          static_cast<T*>(znew), static_cast<T>(a), static_cast<T>(zprev ? c1 : 0.0), static_cast<T>(c2), residual ? 1 : 0, T(0)};
	c->tile2.fused ? tile2_launch<T, true>(c, static_cast<const T*>(z), nullptr, nullptr, &E)
	               : tile2_launch<T, false>(c, static_cast<const T*>(z), nullptr, nullptr, &E);
}
void tile2d_full_step(fi_ctx* c, const void* z, const void* zprev, const void* r, bool residual, void* znew, double a, double c1,
                      double c2)
{
	FI_REQUIRE(tile2d_full_epi_available(c), FI_ERR_UNSUPPORTED, "no fused recurrence step for this context");
	c->dtype == FI_F64 ? tile2_full_step_t<double>(c, z, zprev, r, residual, znew, a, c1, c2)
	                   : tile2_full_step_t<float>(c, z, zprev, r, residual, znew, a, c1, c2);
}

// One step of the Chebyshev polynomial in Dinv (A_model + diag(A_data)) / of the power method on the model operator:
// plain launches of the tile kernel (no cell records), one partial per tile.  The 2-D form of stencil_cheb_step.
template <typename T>
static void tile2_cheb_step_t(fi_ctx* c, const void* z, const void* zprev, const void* r, void* znew, double c1, double c2,
                              double* partial, double zprev_scale, const unsigned short* scaling, int mode)
{
	const void* zp = zprev_scale != 0.0 ? r : (zprev ? zprev : z);
	const bool  has_prev = zprev_scale != 0.0 || zprev;
	Epi2<T> E{static_cast<const T*>(zp), static_cast<const T*>(r), scaling ? scaling : c->dinv16.as<unsigned short>(),
	          static_cast<T*>(znew), static_cast<T>(1.0 + c1), static_cast<T>(has_prev ? c1 : 0.0), static_cast<T>(c2), mode,
	          static_cast<T>(zprev_scale)};
	tile2_launch<T, false>(c, static_cast<const T*>(z), nullptr, partial, &E);
}
void tile2d_cheb_step(fi_ctx* c, const void* z, const void* zprev, const void* r, void* znew, double c1, double c2, double* partial,
                      double zprev_scale, const unsigned short* scaling)
{
	c->dtype == FI_F64 ? tile2_cheb_step_t<double>(c, z, zprev, r, znew, c1, c2, partial, zprev_scale, scaling, 2)
	                   : tile2_cheb_step_t<float>(c, z, zprev, r, znew, c1, c2, partial, zprev_scale, scaling, 2);
}
void tile2d_power_step(fi_ctx* c, const void* v, void* vnew, double* partial)
{
	c->dtype == FI_F64 ? tile2_cheb_step_t<double>(c, v, nullptr, v, vnew, 0, 0, partial, 0.0, nullptr, 3)
	                   : tile2_cheb_step_t<float>(c, v, nullptr, v, vnew, 0, 0, partial, 0.0, nullptr, 3);
}

bool tile2d_apply(fi_ctx* c, const void* x, void* y, double* partial)
{
	if (!c->tile2.valid) { return false; }
	if (c->dtype == FI_F64) {
		c->tile2.fused ? tile2_launch<double, true>(c, static_cast<const double*>(x), static_cast<double*>(y), partial)
		               : tile2_launch<double, false>(c, static_cast<const double*>(x), static_cast<double*>(y), partial);
	} else {
		c->tile2.fused ? tile2_launch<float, true>(c, static_cast<const float*>(x), static_cast<float*>(y), partial)
		               : tile2_launch<float, false>(c, static_cast<const float*>(x), static_cast<float*>(y), partial);
	}
	return true;
}

}  // namespace fi
