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
// Work decomposition (CDNA4): one workgroup = 256 threads = a TX x TY tile of (x, y) (128 x 8 in fp32, 64 x 8 in
// fp64; 64 x 16 / 32 x 16 where the wide tile would overhang the lattice more) marching over ZC planes of z; a
// thread owns VX consecutive x (one 16-byte global load/store per plane: float4/double2).
//   * z neighbours live in registers: x(z), x(z+1), x(z+2) plus the two carried row values u(z-1), u(z-2)
//     -- each plane is read from HBM once, five planes ahead of its use;
//   * x/y neighbours come from an LDS copy of the plane (tile + halo ring), 3-deep ring => one barrier per
//     plane, with or without data; own columns are 16-byte aligned in LDS (ds_read_b128 for the y rows);
//   * every load that crosses a step is unconditional (clamped addresses, dummy halo slots): conditional ones make
//     the compiler's s_waitcnt bookkeeping drain the whole pipeline (vmcnt(0)) at the first use, and so does any
//     register spill -- the kernels are held to zero spills (tests/test_kernel_resources.py);
//   * boundary masks for x/y are per-thread lane masks (SGPR pairs) hoisted out of the march; z masks are
//     wave-uniform; rows need not be a multiple of the 16-byte group (the last group of a row reads on into the
//     next row under zero masks and stores its valid points one by one);
//   * data cells of layer z (corners on planes z and z+1, both in the LDS ring): one lane per cell.  A
//     cell holding a single data row a is kept as that row (y += a (a.x), 32 B in fp32); a cell holding
//     more rows as up to 8 factor rows (fp64: the packed symmetric block).  The 8 corner products are
//     ADDED into LDS accumulation planes, one pair per lattice plane: [corner y-bit][TY][TX], ring of 3
//     planes.  Ordering without barriers or races: the cells of a layer are split into 4 bands by the
//     y-row of their origin, one band per wave; a lattice point of the by=0 plane only receives products
//     from cells whose origin row is the point's row (one band, one wave), of the by=1 plane only from the
//     row below (one band, one wave).  Inside a wave the add is a plain LDS read-add-write in two phases by
//     the corner's x-bit: inside a phase the 4 corners of a cell go to 4 different planes and the cells of one
//     instruction are distinct, and the LDS instructions of a wave execute in order -- so every sum is formed
//     in the same order on every run: bitwise reproducible.  (LDS float atomics would also be correct here, but
//     cost ~3 cycles per lane: +31 us per launch.)  Layers with at most 64 cells are scattered by one wave.
//     The owner of a lattice point collects plane z-1 one step late (after the barrier of step z, when
//     layers z-2 and z-1 are complete), so the data path needs no barrier of its own; the stencil result
//     of plane z-1 waits in registers for that one step.  Row records are prefetched one plane ahead by
//     every wave (dense layers: each wave its band).  The scatter code is branch-free (a per-thread dump slot
//     for corners outside the tile).  Nothing on the plane step's critical path may wait for a load it has just
//     issued: a cell with TWO rows is listed as two row records in neighbouring lanes (scattered in two passes)
//     so that it travels with the prefetched stream; contexts of mostly multi-row cells keep cells of >= 3 rows as
//     packed blocks (36 coefficients in independent batches, out = B x) and run the PACK instantiation;
//   * workgroups without any data cell run the plain variant (two launches over disjoint lists), the plain ones
//     over runs of consecutive empty chunks of a tile;
//   * p.q partials: fp32 products per plane, fp64 per-thread accumulation, wave64 shuffle tree, one
//     partial per workgroup;
//   * blockIdx -> tile map is XCD-aware: blocks b, b+8, b+16.. (same XCD, same L2) get adjacent tiles; the
//     chunk length ZC is chosen so that the grid covers the CUs in whole rounds (pick_chunk).

#include <hip/hip_bf16.h>

#include <algorithm>
#include <cstring>
#include <cmath>
#include <type_traits>
#include <vector>

#include <hip/hip_fp16.h>

#include "fi_internal.h"
#include "fi_stencil_common.h"

namespace fi {

namespace {

constexpr int kThreads = 256;
// Depth of the register rings of the loads with a short lead: a ring of R sets gives a lead of R steps (the set a step
// has consumed is refilled at once with the data of R steps on).  Measured (profiles/r2_ablation.md): rings of 2 or 3
// sets, with 3 or 2 workgroups per CU, are no faster than single sets -- the step is a chain of dependent LDS round
// trips and memory-issue stalls, not a wait for one late load.
#ifndef FI_HALO_RING
#define FI_HALO_RING 1
#endif
#ifndef FI_ROW_RING
#define FI_ROW_RING 1
#endif
#ifndef FI_CELL_WAVES
#define FI_CELL_WAVES 3  // waves per SIMD the fused (data cell) variant is register-allocated for (one less with both
                         // model_1 and model_2 on: that variant would spill, and a spill reload drains the load pipeline)
#endif
// Tile shapes (threads along x; a thread owns VX points): 32 -> 128 x 8 tiles in fp32 (512-byte runs per row: 7 %
// faster than 64 x 16 at 512^3, equal at 256^3), 16 -> 64 x 16 for lattices a wide tile would mostly overhang
// (64^3 levels of the cascade: half of every 128-wide tile would be idle).  Chosen per context by march_setup.
// A layer with more records than MarchParams::dense_min is scattered by all four waves (a band each), a sparser one by ONE
// wave while the others skip the code.  Swept in round 4 on the isolated apply of config 4 (60-67 records per layer and
// tile: right at the threshold; profiles/r4_ablation.md section 6), 256^3 fp32 / fp64: 8: 52.2 / 140.3 us, 16: 52.2 / 140.8,
// 32: 52.8 / 132.2, 64: 51.4-52.2 / 103.8-105.9, 128: 58.5 / 106.4, 256: 58.3 / 105.3 (512^3 fp32: 381 / 381 / 381 / 374-381 /
// 436 / 437 us).  64 it stays, for row records and for packed blocks (an SDF: 64 is best there too).
#ifndef FI_DENSE_MIN_ROWS
#define FI_DENSE_MIN_ROWS 64
#endif
#ifndef FI_DENSE_MIN_PACK
#define FI_DENSE_MIN_PACK 64
#endif
constexpr int kR       = 2;   // halo rows/cols kept in LDS

#ifdef FI_STAMPS  // timing builds: in-kernel time stamps of one workgroup's march (tools/exp_stamps.py)
__device__ unsigned long long g_stamp[64 * 8 * 4];
#endif

template <typename T>
struct MarchCoef {
	T w0x3;  // 3 * model_0^2
	T w1sq;  // model_1^2
	T w2sq;  // model_2^2
};

// Epilogue for one step of a Chebyshev recurrence: instead of storing q = A z the kernel forms, for its own points,
//     mode 0 (plain variant; polynomial preconditioner of fi_solver.hip, cg_run_poly):
//         s     = Dinv (q - m z) + z                   = Dinv (A_model + diag(A_data)) z,  m = the model diagonal
//         z_new = a z - c1 z_prev + c2 (Dinv r - s)    (three-term recurrence, a = 1 + c1; the step from z_prev = 0
//                                                       passes z itself with c1 = 0)
//       and the partials of r . z_new.  The data rows enter that preconditioner through their diagonal only -- as good
//       a preconditioner as the polynomial in the full operator (profiles/r2_ablation.md: equal iteration counts) -- so
//       a step never reads a cell record: 5 lattice passes (z, z_prev, r, Dinv in; z_new out) in one launch;
//     mode 1 (plain variant): one step of the power method for that polynomial's bound, on the MODEL operator alone
//       (data only lowers the Rayleigh quotients of Dinv (A_model + diag)): z_new = (A_model z) / m, partials of
//       z_new . z_new;
//     mode 2 (every variant; smoother of the V-cycle, fi_solver.hip cheb_smooth): the same recurrence on the FULL
//       operator, q = A z with the data cells: z_new = a z - c1 z_prev + c2 Dinv (r - q);
//     mode 3 (every variant): the residual z_new = r - q.
//     mode 4 (plain variant): the residual of the LUMPED operator A~ = A_model + diag(l), l = the row sums of the data term
//       (fi_ctx::dlump, passed as z_prev): z_new = r - (A_model z + l z).  The fp32 replica of a mixed-precision context
//       whose data are value rows runs its finest level on A~ (fi_solver.hip, twin_assemble_lumped): no cell records.
// The fused (data cell) variant completes a plane one step late and runs its epilogue there; with the operands of the
// epilogue it is register-allocated for two workgroups per CU (at three it would spill ~190 VGPRs).
template <typename T>
struct ChebEpi {
	const T* zprev;
	const T* r;
	const unsigned short* dinv;  // bfloat16 (fi_ctx::dinv16)
	T*       znew;
	T        a, c1, c2;
	int      mode;
	T        pro_scale;  // PRO variant (the polynomial's FIRST step): the kernel's input vector is r and the operand of the
	                     // stencil is formed on load, z_0 = pro_scale * Dinv * r -- z_0 is never stored (z_prev = 0 there)
	T        zp_scale;  // mode 0, non-zero: z_prev = zp_scale * Dinv * (the vector passed as zprev) -- the polynomial's second
	                    // step passes r here: its z_prev is z_0 = Dinv r / theta, which then needs no lattice pass of its own
	int      fmt;       // bfloat16 STORAGE of the polynomial's iterates (plain fp32 variant with the epilogue, mode 0): bit 0 the
	                    // kernel's input vector, bit 1 z_prev, bit 2 z_new -- the pointers then address unsigned shorts.  The
	                    // iterates of a preconditioner need no more (section 9 of profiles/r4_ablation.md: the same iteration
	                    // counts to 1e-9), and a step moves 8-14 bytes per point instead of 10-18.
	int      round16;   // timing builds (FI_Z16): the step's result rounded to bfloat16 (1) / half (2) precision -- what 16-bit
	                    // storage of the polynomial's iterates would do to the preconditioner (profiles/r4_ablation.md section 9)
	const T* acc;       // mode 0 without the operand formed on load, non-null: z_new = acc + (the step's result) -- the LAST step of
	                    // a smoother's polynomial adds the correction onto x itself (acc == znew: x += M (b - A x) with no pass of
	                    // its own for the sum; fp32 / fp64 storage, never bfloat16)
	const T* dotv;      // with acc, non-null: the launch's partials are those of dotv . z_new (z_new after the sum) instead of r . z_new --
	                    // the V-cycle's b . x, which is the fp64 CG's r . z, summed by the cycle's last launch (marching kernel only)
};

struct CellLists {
	const uint32_t* lay_row;   // [nwg*(zc+1)*4+1]
	const uint32_t* lay_blk;   // [nwg*(zc+1)*4+1]
	const uint32_t* pos_row;   // (tcx+1) | (tcy+1)<<16
	const uint32_t* pos_blk;
	const void*     coef_row;  // T[n_row][8]
	const void*     coef_blk;  // T[n_blk][8][8]: up to 8 factor rows per multi-row cell
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

// row / column of packed index e (inverse of tri)
__host__ __device__ constexpr int tri_row(int e)
{
	int i = 0;
	while (tri(i + 1, i + 1) <= e && i < 7) { ++i; }
	return i;
}
__host__ __device__ constexpr int tri_col(int e) { return tri_row(e) + (e - tri(tri_row(e), tri_row(e))); }

// workgroups per CU the fused variant is register-allocated for: one less with both model_1 and model_2 on, and for
// model_1 alone in the fp32 variant for packed blocks (those variants would spill, and a spill reload drains the
// load pipeline)
template <typename T>
__host__ __device__ constexpr int fused_waves(bool has1, bool has2, bool pack)
{
	// fp64: the bound of 2 leaves the allocator room; it still lands at 166-168 VGPRs = 3 workgroups per CU, without the
	// 4 spilled registers (20 bytes of scratch per lane) the bound of 3 cost the model_2 variant
#ifndef FI_F64_FUSED_WAVES
#define FI_F64_FUSED_WAVES (FI_CELL_WAVES - 1)
#endif
	if (sizeof(T) == 8) { return FI_F64_FUSED_WAVES; }
	return (has1 && (has2 || (pack && sizeof(T) == 4))) ? FI_CELL_WAVES - 1 : FI_CELL_WAVES;
}

// PACK (fused variants): the context keeps its cells of >= 3 rows as packed blocks (CellData::pack: an SDF, the
// coarse levels of a cascade); the other variant carries the factor-row loop only (and, in fp64, the packed block
// of cells beyond 8 rows).  Two variants because the unrolled block product is code the row-dominated contexts
// would only pay for: config 4's finest level ran 2-4 us per launch slower with it compiled in.
template <typename T, bool HAS1, bool HAS2, bool CELLS, int TXT, bool PACK, bool EPI = false, bool PRO = false>
__global__ __launch_bounds__(kThreads, CELLS ? (EPI ? 2 : fused_waves<T>(HAS1, HAS2, PACK)) : (EPI && HAS1 && HAS2 && sizeof(T) == 4 ? FI_BASE_WAVES - 1 : FI_BASE_WAVES)) void k_apply_march3d(MarchParams P, MarchCoef<T> C, CellLists L,
                                                             const T* __restrict__ x, T* __restrict__ y,
                                                             double* __restrict__ partial,
                                                             const int* __restrict__ done,
                                                             const uint32_t* __restrict__ wg_list, int nlist,
                                                             const uint32_t* __restrict__ wg_runs,
                                                             ChebEpi<T> E = ChebEpi<T>{})
{
	static_assert(!(EPI && CELLS && sizeof(T) == 8), "the fused variant carries the epilogue in fp32 only");
	static_assert(!PRO || (EPI && !CELLS), "the input formed on load belongs to the plain variant with the epilogue");
	using V = typename VecOf<T>::V;
	constexpr int kTXT = TXT;             // threads along x
	constexpr int kTY  = kThreads / TXT;  // tile rows (= threads along y)
	constexpr int VX   = VecOf<T>::VX;
	constexpr int TX   = kTXT * VX;
	constexpr int PADX = VX;             // own columns start 16-byte aligned
	constexpr int W    = TX + 2 * PADX;  // LDS row length
	constexpr int ROWS = kTY + 2 * kR;
	constexpr int R    = HAS2 ? 2 : 1;
	// halo ring of the plane tile: the 2R rows above/below the tile over the own columns go as 16-byte loads
	// (threads 0 .. NVEC-1, one each); the 2R columns left/right of the tile, corners included, as scalars
	// (threads NVEC .. NVEC+NSC-1).  At most one halo load and one LDS store per thread and plane.
	constexpr int NVEC = 2 * R * kTXT;
	constexpr int NSC  = 2 * R * (kTY + 2 * R);
	static_assert(NVEC + NSC <= kThreads, "halo slots exceed the workgroup");
	constexpr int NLAY  = 4 * 66 + 1;  // (zc + 2) layers x 4 bands of record bounds, zc <= 64
	constexpr bool kDensePF = sizeof(T) == 4 && HAS2 && !HAS1 && (TXT == 32 || !PACK);  // band prefetch in dense layers (prefetch_rows): the variants with registers to spare

	__shared__ __attribute__((aligned(16))) T xs[3][ROWS][W];
	// accumulation planes of the data term: [plane ring of 3][corner y-bit][TY][TX]
	__shared__ __attribute__((aligned(16))) T yb[CELLS ? 3 : 1][CELLS ? 2 : 1][CELLS ? kTY : 1][CELLS ? TX : VX];
	__shared__ T ydump[CELLS ? kThreads : 1];  // write-only target of corner products that fall outside the tile
	__shared__ uint32_t s_lay[2][CELLS ? NLAY : 1];  // record bounds of this workgroup's (layer, band) lists
	__shared__ double red[kThreads / 64];
#ifdef FI_STAMPS
	__shared__ unsigned long long s_stamp[4][64 * 8];
#define FI_STAMP(stepno, k)                                                                                  \
	do {                                                                                                     \
		if ((threadIdx.x & 63) == 0 && (stepno) < 64) {                                                      \
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                               \
			s_stamp[threadIdx.x >> 6][(stepno) * 8 + (k)] = __builtin_amdgcn_s_memtime();                     \
		}                                                                                                    \
	} while (0)
#else
#define FI_STAMP(stepno, k) do { } while (0)
#endif

	if (done && *done) { return; }
#ifdef FI_STAMPS
	for (int i = threadIdx.x; i < 4 * 64 * 8; i += kThreads) { s_stamp[i / 512][i % 512] = 0ull; }
	__syncthreads();
#endif

	// XCD-aware tile order: consecutive tiles on one XCD.  With a list, the launch covers only the listed
	// workgroups (the ones with / without data cells: march_launch).
	const int nrun = wg_list ? nlist : P.nwg;
	const int per  = (nrun + 7) / 8;
	const int slot = (blockIdx.x % 8) * per + blockIdx.x / 8;
	if (slot >= nrun) { return; }
	const int wg = wg_list ? static_cast<int>(wg_list[slot]) : slot;
	// plain launches over surface-type data: a workgroup marches over `nrun` consecutive chunks of its tile (the
	// chunks of a split launch are short for the sake of the data workgroups)
	const int nchunk = (!CELLS && wg_runs) ? static_cast<int>(wg_runs[slot]) : 1;
	const int tiles_xy = P.tiles_x * P.tiles_y;
	const int chunk    = wg / tiles_xy;
	const int txy      = wg % tiles_xy;
	const int tile_y   = txy / P.tiles_x;
	const int tile_x   = txy % P.tiles_x;

	const int tx = threadIdx.x % kTXT, ty = threadIdx.x / kTXT;
	const int x0 = tile_x * TX, y0 = tile_y * kTY;
	const int gx = x0 + VX * tx, gy = y0 + ty;
	// a thread's VX points are all inside (active), partly inside (the last group of a row whose length is not a
	// multiple of VX: `tail` points are stored one by one) or outside the lattice
	const int  nvalid = gy < P.ny ? (P.nx - gx < 0 ? 0 : (P.nx - gx > VX ? VX : P.nx - gx)) : 0;
	const bool active = nvalid == VX;
	const bool tail   = nvalid > 0 && nvalid < VX;
	const int lx = PADX + VX * tx, ly = kR + ty;

	const int z_begin = P.own_z0 + chunk * P.zc;
	int       z_end   = z_begin + P.zc * nchunk;
	if (z_end > P.own_z1) { z_end = P.own_z1; }

	// offsets inside a plane stay 32-bit and unsigned: the plane base is wave-uniform (SGPR pair), so every
	// global access is `saddr + 32-bit voffset` and no 64-bit address lives in VGPRs
	const uint32_t col = static_cast<uint32_t>(gy) * static_cast<uint32_t>(P.nx) + static_cast<uint32_t>(gx);

	// ---- per-thread constants: halo slots and x/y boundary masks --------------------------------
	typedef T NV __attribute__((ext_vector_type(VX)));
	typedef unsigned short DV16 __attribute__((ext_vector_type(VX)));  // VX bfloat16 values: 8 / 4 bytes
	auto dinv_of = [](const DV16& d, int j) -> T {
		return static_cast<T>(__uint_as_float(static_cast<unsigned int>(d[j]) << 16));
	};
	struct HaloRegs {
		NV vec;
		T  sc;
		DV16           dvec;  // PRO: the scaling of the same points
		unsigned short dsc;
	};
	const bool hv_on = threadIdx.x < NVEC;
	const bool hs_on = threadIdx.x >= NVEC && threadIdx.x < NVEC + NSC;
	int hv_lds = 0, hs_lds = 0;   // LDS element offsets inside a plane buffer
	int hv_glb = 0, hs_glb = 0;   // element offsets inside a lattice plane (clamped into the lattice)
	{
		const int t    = hv_on ? static_cast<int>(threadIdx.x) : 0;
		const int hrow = t / kTXT, vx = t % kTXT;
		const int hly  = hrow < R ? (kR - R + hrow) : (kR + kTY + (hrow - R));
		int hgy = y0 + hly - kR, hgx = x0 + VX * vx;
		hgy = hgy < 0 ? 0 : (hgy >= P.ny ? P.ny - 1 : hgy);
		hgx = hgx >= P.nx ? (P.nx > VX ? P.nx - VX : 0) : hgx;  // a group that straddles the row end reads on into the
		hv_lds = hly * W + PADX + VX * vx;                      // next row (finite values under zero masks)
		hv_glb = hgy * P.nx + hgx;
	}
	{
		const int u   = hs_on ? static_cast<int>(threadIdx.x) - NVEC : 0;
		const int row = u / (2 * R), k = u % (2 * R);
		const int hly = kR - R + row;
		const int hlx = k < R ? (PADX - R + k) : (PADX + TX + (k - R));
		int hgy = y0 + hly - kR, hgx = x0 + hlx - PADX;
		hgy = hgy < 0 ? 0 : (hgy >= P.ny ? P.ny - 1 : hgy);
		hgx = hgx < 0 ? 0 : (hgx >= P.nx ? P.nx - 1 : hgx);
		hs_lds = hly * W + hlx;
		hs_glb = hgy * P.nx + hgx;
	}

	const int      h_lds = hv_on ? hv_lds : hs_lds;
	const uint32_t hvg = static_cast<uint32_t>(hv_on ? hv_glb : 0), hsg = static_cast<uint32_t>(hs_on ? hs_glb : 0);

	// model_2 rows along x anchored at gx-2 .. gx+VX-1; along y anchored at gy-2, gy-1, gy
	bool m2x[VX + 2];  // lane masks (SGPR pairs), applied by select
	bool c2y[3];  // lane masks like the x masks (SGPR pairs)
	bool m1x[VX + 1];
	bool c1y[2];
	if (HAS2) {
#pragma unroll
		for (int k = 0; k < VX + 2; ++k) {
			const int a = gx - 2 + k;
			m2x[k] = (a >= 0 && a + 2 < P.nx);
		}
		c2y[0] = (gy - 2 >= 0 && gy < P.ny);
		c2y[1] = (gy - 1 >= 0 && gy + 1 < P.ny);
		c2y[2] = (gy + 2 < P.ny);
	}
	if (HAS1) {
#pragma unroll
		for (int k = 0; k < VX + 1; ++k) {
			const int a = gx - 1 + k;  // rows [-1,+1] anchored at a: x_{a+1} - x_a
			m1x[k] = (a >= 0 && a + 1 < P.nx);
		}
		c1y[0] = (gy - 1 >= 0 && gy < P.ny);   // row anchored at gy-1 touches gy with +1
		c1y[1] = (gy + 1 < P.ny);              // row anchored at gy touches gy with -1
	}

	// Loads never branch: every address is clamped into the lattice.  A clamped (wrong) value is only ever
	// multiplied by a zero mask / a zero block coefficient, so it just has to be finite.
	const int lz_lo = P.zoff < 0 ? -P.zoff : 0;
	const int lz_hi = (P.nzl < P.gz - P.zoff ? P.nzl : P.gz - P.zoff) - 1;
	const int gxc = gx < P.nx ? gx : (P.nx > VX ? P.nx - VX : 0);
	const int gyc = gy < P.ny ? gy : P.ny - 1;
	const uint32_t xoff = static_cast<uint32_t>(gyc) * static_cast<uint32_t>(P.nx) + static_cast<uint32_t>(gxc);
	auto clamp_plane = [&](int lz) { return lz < lz_lo ? lz_lo : (lz > lz_hi ? lz_hi : lz); };
	constexpr bool Z16 = EPI && !CELLS && sizeof(T) == 4;  // the variant whose vectors may be stored as bfloat16 (ChebEpi::fmt)
	auto from16 = [](const DV16& d) -> V {
		V v;
		T* p = reinterpret_cast<T*>(&v);
#pragma unroll
		for (int j = 0; j < VX; ++j) { p[j] = static_cast<T>(__uint_as_float(static_cast<unsigned int>(d[j]) << 16)); }
		return v;
	};
	auto to16 = [](const V& v) -> DV16 {  // round to nearest even: one v_cvt_pk_bf16_f32 per pair on gfx950
		DV16 d;
		const T* p = reinterpret_cast<const T*>(&v);
#pragma unroll
		for (int j = 0; j + 1 < VX; j += 2) {
			const __hip_bfloat162 b = __float22bfloat162_rn(float2{static_cast<float>(p[j]), static_cast<float>(p[j + 1])});
			d[j]     = __bfloat16_as_ushort(b.x);
			d[j + 1] = __bfloat16_as_ushort(b.y);
		}
		return d;
	};
	// bfloat16 vectors (ChebEpi::fmt): a load leaves the RAW bits in the low half of its registers, and they become floats
	// where the values are first used (decode_own / write_plane / the epilogue) -- NOT behind the load: a conversion there is
	// a use, and the wave waited (s_waitcnt vmcnt(0): for every load and store in flight) for the plane it had just
	// requested 5 steps ahead, for the halo and for z_prev, three exposed memory latencies per plane step (round 4's bfloat16
	// storage took 27 % of the bytes and 16 % of the time off a step for this reason; profiles/r5_ablation.md section 15)
	auto raw16 = [](const DV16& d) -> V {
		V v;
		*reinterpret_cast<DV16*>(&v) = d;
		return v;
	};
	auto load_own = [&](int lz) -> V {
		if (Z16 && (E.fmt & 1)) {
			const unsigned short* xp = reinterpret_cast<const unsigned short*>(x) + static_cast<int64_t>(clamp_plane(lz)) * P.plane;
			return raw16(*reinterpret_cast<const DV16*>(xp + xoff));
		}
		const T* xp = x + static_cast<int64_t>(clamp_plane(lz)) * P.plane;
		return *reinterpret_cast<const V*>(xp + xoff);
	};
	auto decode_own = [&](V& v) {
		if (Z16 && (E.fmt & 1)) { v = from16(*reinterpret_cast<const DV16*>(&v)); }
	};
	auto load_halo = [&](int lz, HaloRegs& h) {
		const T* xp = x + static_cast<int64_t>(clamp_plane(lz)) * P.plane;
#ifdef FI_TIMING_BUILD  // timing builds only (tools/build_variant.sh -DFI_TIMING_BUILD): results wrong by construction
		if (P.dbg & 1) { return; }
#endif
		if (Z16 && (E.fmt & 1)) {
			const unsigned short* xq = reinterpret_cast<const unsigned short*>(x) + static_cast<int64_t>(clamp_plane(lz)) * P.plane;
			const V hv = raw16(*reinterpret_cast<const DV16*>(xq + hvg));  // (raw bits: decoded by write_plane)
			h.vec = *reinterpret_cast<const NV*>(&hv);
			h.sc  = static_cast<T>(__uint_as_float(static_cast<unsigned int>(xq[hsg])));
		} else {
			h.vec = *reinterpret_cast<const NV*>(xp + hvg);
			h.sc  = xp[hsg];
		}
		if (PRO) {
#ifdef FI_TIMING_BUILD
			if (P.dbg & 64) { return; }  // no halo loads of the scaling
#endif
			const unsigned short* dp = E.dinv + static_cast<int64_t>(clamp_plane(lz)) * P.plane;
			h.dvec = *reinterpret_cast<const DV16*>(dp + hvg);
			h.dsc  = dp[hsg];
		}
	};
	// PRO: the scaling of the thread's own points of plane lz, and the operand formed from a loaded pair
	auto load_own_d = [&](int lz) -> DV16 {
#ifdef FI_TIMING_BUILD
		if (P.dbg & 128) { return DV16{}; }  // no own loads of the scaling
#endif
		return *reinterpret_cast<const DV16*>(E.dinv + static_cast<int64_t>(clamp_plane(lz)) * P.plane + xoff);
	};
	auto form_own = [&](V& v, const DV16& d) {
		T* pv = reinterpret_cast<T*>(&v);
#pragma unroll
		for (int j = 0; j < VX; ++j) { pv[j] = E.pro_scale * dinv_of(d, j) * pv[j]; }
	};
	auto write_plane = [&](int buf, const V& own, const HaloRegs& hraw) {
		T* base = &xs[buf][0][0];
		*reinterpret_cast<V*>(&xs[buf][ly][lx]) = own;
		HaloRegs h = hraw;
		if (Z16 && (E.fmt & 1)) {  // the halo's raw bfloat16 bits (load_halo)
			const V hv = from16(*reinterpret_cast<const DV16*>(&hraw.vec));
			h.vec = *reinterpret_cast<const NV*>(&hv);
			h.sc  = static_cast<T>(__uint_as_float(__float_as_uint(static_cast<float>(hraw.sc)) << 16));
		}
		if (PRO) {
			NV hv;
#pragma unroll
			for (int j = 0; j < VX; ++j) { hv[j] = E.pro_scale * dinv_of(h.dvec, j) * h.vec[j]; }
			if (hv_on) { *reinterpret_cast<NV*>(base + h_lds) = hv; }
			if (hs_on) { base[h_lds] = E.pro_scale * static_cast<T>(__uint_as_float(static_cast<unsigned int>(h.dsc) << 16)) * h.sc; }
		} else {
			if (hv_on) { *reinterpret_cast<NV*>(base + h_lds) = h.vec; }
			if (hs_on) { base[h_lds] = h.sc; }
		}
	};

	// ---- data cells of one layer: corner products into the 8 corner planes ------------------------------
	// the zc+2 range bounds of this workgroup are staged in LDS once, in the prologue (a scalar global load
	// per plane would put its latency on every step of the march)
	const uint32_t* layR = s_lay[0];
	const uint32_t* layB = s_lay[1];
	// cells: one band of origin rows per wave (wave-uniform values are kept in SGPRs)
	const int band = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)), lane = threadIdx.x & 63;
	auto uni = [](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); };
	// add the 8 corner products of a cell into the accumulation planes of lattice planes slot_lo (corner z-bit
	// 0) and slot_hi (z-bit 1); corners outside the tile -- and the lower ones when lo_ok is false -- go to
	// the thread's dump slot (no branches)
	auto put8 = [&](int tcx, int tcy, const T* out, int slot_lo, int slot_hi, bool lo_ok, bool on = true) {
		T* const dump = &ydump[threadIdx.x];
		const bool vx0 = on && tcx >= 0, vx1 = on && tcx + 1 < TX, vy0 = tcy >= 0, vy1 = tcy + 1 < kTY;
		T* const base = &yb[0][0][0][0] + tcy * TX + tcx;
		// Plain read-add-write (LDS float atomics cost ~3 cycles per lane here).  Two phases by the corner's
		// x-bit: inside a phase the 4 corners of a cell go to 4 different planes and the cells of a wave
		// instruction are distinct, so no two lanes touch one address; across phases (and across the batches
		// of a loop) the LDS instructions of the wave execute in program order -- the clobbers keep the compiler
		// from moving a phase's reads above the previous phase's writes.
#pragma unroll
		for (int bx = 0; bx < 2; ++bx) {
			asm volatile("" ::: "memory");
			T* dst[4];
			T  cur[4];
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int by = k & 1, bz = k >> 1;
				const bool ok = (bx ? vx1 : vx0) && (by ? vy1 : vy0) && (bz ? true : lo_ok);
				T* d = base + ((bz ? slot_hi : slot_lo) * 2 + by) * (kTY * TX) + by * TX + bx;
				dst[k] = ok ? d : dump;
			}
#pragma unroll
			for (int k = 0; k < 4; ++k) { cur[k] = *dst[k]; }
#pragma unroll
			for (int k = 0; k < 4; ++k) { *dst[k] = cur[k] + out[bx + 2 * (k & 1) + 4 * (k >> 1)]; }
		}
		asm volatile("" ::: "memory");
	};
	auto corners = [&](int tcx, int tcy, int buf_lo, int buf_hi, T* xv) {
#pragma unroll
		for (int q = 0; q < 8; ++q) {
			const int bx = q & 1, by = (q >> 1) & 1, bz = q >> 2;
			xv[q] = xs[bz ? buf_hi : buf_lo][kR + tcy + by][PADX + tcx + bx];
		}
	};
	auto row_apply = [&](uint32_t pos, const T* a, int buf_lo, int buf_hi, int slot_lo, int slot_hi, bool lo_ok) {
		const int tcx = static_cast<int>(pos & 0xFFu) - 1, tcy = static_cast<int>(pos >> 16) - 1;
		T xv[8];
		corners(tcx, tcy, buf_lo, buf_hi, xv);
		T t = T(0);
#pragma unroll
		for (int q = 0; q < 8; ++q) { t += a[q] * xv[q]; }
		T out[8];
#pragma unroll
		for (int i = 0; i < 8; ++i) { out[i] = a[i] * t; }
		// The second row of a two-row cell sits in the lane next to the first and goes to the same addresses: the first row's
		// lane takes its products over a DPP shift (VALU: no LDS round trip) and adds both at once.  A pass of its own for
		// the second rows -- round 2-4 -- cost the scattering wave two more dependent LDS round trips in every layer that
		// holds ONE such cell (60 % of config 4's layers), on the path the step's barrier waits for.  (A second row in lane 0
		// has its partner in the previous batch of 64: it adds by itself.)
		const bool second = ((pos >> 8) & 0xFFu) != 0u;
		if (__ballot(second) != 0ull) {
			const int nxt = __builtin_amdgcn_update_dpp(0, second ? 1 : 0, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				T o;
				if constexpr (sizeof(T) == 8) {
					const long long b = __double_as_longlong(static_cast<double>(out[i]));
					const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(b), 0x130, 0xF, 0xF, false);
					const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(b >> 32), 0x130, 0xF, 0xF, false);
					o = static_cast<T>(__longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo)));
				} else {
					o = static_cast<T>(__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(static_cast<float>(out[i])), 0x130, 0xF, 0xF, false)));
				}
				if (nxt) { out[i] += o; }
			}
		}
		put8(tcx, tcy, out, slot_lo, slot_hi, lo_ok, !(second && lane > 0));
	};
	auto load_row = [&](uint32_t r, uint32_t* pos, T* a) {
		*pos = L.pos_row[r];
		const V* ap = reinterpret_cast<const V*>(static_cast<const T*>(L.coef_row) + static_cast<int64_t>(r) * 8);
#pragma unroll
		for (int k = 0; k < 8 / VX; ++k) {
			const V  v  = ap[k];
			const T* pv = reinterpret_cast<const T*>(&v);
#pragma unroll
			for (int j = 0; j < VX; ++j) { a[k * VX + j] = pv[j]; }
		}
	};
	// this wave's band of layer `layer`: rows from record rsR on (the first 64 may have been prefetched) and
	// all block records
	auto cells_scatter = [&](uint32_t rsR, uint32_t reR, uint32_t rsB, uint32_t reB, int buf_lo, int buf_hi, int slot_lo,
	                         int slot_hi, bool lo_ok) {
		for (uint32_t r = rsR + lane; r < reR; r += 64) {
			uint32_t pos;
			T a[8];
			load_row(r, &pos, a);
			row_apply(pos, a, buf_lo, buf_hi, slot_lo, slot_hi, lo_ok);
		}
		const T* multi = static_cast<const T*>(L.coef_blk);
		for (uint32_t r = rsB + lane; r < reB; r += 64) {
			const uint32_t pos = L.pos_blk[r];
			const int tcx = static_cast<int>(pos & 0xFFu) - 1, tcy = static_cast<int>(pos >> 16) - 1;
			const int nrows = static_cast<int>((pos >> 8) & 0xFFu);
			T xv[8], out[8];
			corners(tcx, tcy, buf_lo, buf_hi, xv);
#pragma unroll
			for (int i = 0; i < 8; ++i) { out[i] = T(0); }
			const uint32_t ro = r * 64u;  // 32-bit element offset against the uniform base: one address register
			const V* ap = reinterpret_cast<const V*>(multi + static_cast<int64_t>(r) * 64);
			if ((PACK || sizeof(T) == 8) && nrows == 0xFF) {  // the packed symmetric block, out = B x
				if (PACK) {
					// batches of 8 coefficients -- the register footprint of one factor row; all 36 at once do not fit beside
					// the march's register rings.  The record is 144 B in fp32: the first batch brings in its lines, the others
					// hit them in L1.
	#pragma unroll
					for (int batch = 0; batch < 5; ++batch) {
						constexpr int NV8 = 8 / VX;
						V w[NV8];
	#pragma unroll
						for (int v = 0; v < NV8; ++v) {
							if (batch * 8 + v * VX < 36) { w[v] = *reinterpret_cast<const V*>(multi + (ro + static_cast<uint32_t>(batch * 8 + v * VX))); }
						}
	#pragma unroll
						for (int e = 0; e < 8; ++e) {
							if (batch * 8 + e < 36) {
								const int i = tri_row(batch * 8 + e), j = tri_col(batch * 8 + e);  // constants once unrolled
								const T bv = reinterpret_cast<const T*>(&w[e / VX])[e % VX];
								out[i] += bv * xv[j];
								if (i != j) { out[j] += bv * xv[i]; }
							}
						}
						asm volatile("" ::: "memory");
					}
				} else {  // fp64, cells beyond 8 rows
					T b[36];
#pragma unroll
					for (int k = 0; k < 36 / VX; ++k) {
						const V  w  = ap[k];
						const T* pw = reinterpret_cast<const T*>(&w);
#pragma unroll
						for (int j = 0; j < VX; ++j) { b[k * VX + j] = pw[j]; }
					}
#pragma unroll
					for (int i = 0; i < 8; ++i) {
#pragma unroll
						for (int j = 0; j < 8; ++j) { out[i] += b[i <= j ? tri(i, j) : tri(j, i)] * xv[j]; }
					}
				}
				put8(tcx, tcy, out, slot_lo, slot_hi, lo_ok);
				continue;
			}
			for (int k = 0; k < nrows; ++k) {  // the cell's factor rows, one after another: out += a (a.x)
				T a[8];
#pragma unroll
				for (int v = 0; v < 8 / VX; ++v) {
					const V  w  = ap[k * (8 / VX) + v];
					const T* pw = reinterpret_cast<const T*>(&w);
#pragma unroll
					for (int j = 0; j < VX; ++j) { a[v * VX + j] = pw[j]; }
				}
				T t = T(0);
#pragma unroll
				for (int q = 0; q < 8; ++q) { t += a[q] * xv[q]; }
#pragma unroll
				for (int i = 0; i < 8; ++i) { out[i] += a[i] * t; }
			}
			put8(tcx, tcy, out, slot_lo, slot_hi, lo_ok);
		}
	};
	// owner side: the finished sums of one lattice plane; the slots are zeroed for the plane 3 steps on
	auto cells_gather = [&](int slot, T* data) {
		const V zero = V{};
		V* s0 = reinterpret_cast<V*>(&yb[slot][0][ty][VX * tx]);
		V* s1 = reinterpret_cast<V*>(&yb[slot][1][ty][VX * tx]);
		const V  v0 = *s0, v1 = *s1;
		const T* p0 = reinterpret_cast<const T*>(&v0);
		const T* p1 = reinterpret_cast<const T*>(&v1);
#pragma unroll
		for (int j = 0; j < VX; ++j) { data[j] = p0[j] + p1[j]; }
		*s0 = zero;
		*s1 = zero;
	};

	// ---- Chebyshev epilogue (EPI): the x/y part of the model diagonal of the thread's points, and the operand loads
	T mxy[EPI ? VX : 1];
	if (EPI && !CELLS) {
		T ay = T(0), by = T(0);
		if (HAS2) { ay = (c2y[0] ? T(1) : T(0)) + (c2y[1] ? T(4) : T(0)) + (c2y[2] ? T(1) : T(0)); }
		if (HAS1) { by = (c1y[0] ? T(1) : T(0)) + (c1y[1] ? T(1) : T(0)); }
#pragma unroll
		for (int j = 0; j < VX; ++j) {
			T m = C.w0x3;
			if (HAS2) { m += C.w2sq * ((m2x[j] ? T(1) : T(0)) + (m2x[j + 1] ? T(4) : T(0)) + (m2x[j + 2] ? T(1) : T(0)) + ay); }
			if (HAS1) { m += C.w1sq * ((m1x[j] ? T(1) : T(0)) + (m1x[j + 1] ? T(1) : T(0)) + by); }
			mxy[j] = m;
		}
	}
	struct EpiRegs {
		V    zp, rv;
		DV16 dv;
		V    av;  // ChebEpi::acc
		V    bv;  // ChebEpi::dotv
	};
	// operands of plane lz (clamped like every load that crosses a step): issued right behind the epilogue that
	// consumed the previous set, used one step later
	auto load_epi = [&](int lz, EpiRegs& e) {
		const int64_t o = static_cast<int64_t>(clamp_plane(lz)) * P.plane + xoff;
		if (Z16 && (E.fmt & 2)) {
			e.zp = raw16(*reinterpret_cast<const DV16*>(reinterpret_cast<const unsigned short*>(E.zprev) + o));  // (decoded by epi_zp)
		} else {
			e.zp = *reinterpret_cast<const V*>(E.zprev + o);  // never null: the host passes z itself with c1 = 0 on the first step
		}
		e.rv = *reinterpret_cast<const V*>(E.r + o);
		e.dv = *reinterpret_cast<const DV16*>(E.dinv + o);
		if (!CELLS && E.acc) {
			e.av = *reinterpret_cast<const V*>(E.acc + o);
			if (E.dotv) { e.bv = *reinterpret_cast<const V*>(E.dotv + o); }
		}
	};

	auto epi_zp = [&](const EpiRegs& e) -> V {  // z_prev of the epilogue's operand set as floats
		if (Z16 && (E.fmt & 2)) { return from16(*reinterpret_cast<const DV16*>(&e.zp)); }
		return e.zp;
	};
	// modes 2 / 3 for one completed plane: zc = the plane's z values, q = A z (full operator)
	auto epi_full = [&](const EpiRegs& e, const T* zc, const T* q, T* pz) -> T {
		const V  zpv = epi_zp(e);
		const T* zp = reinterpret_cast<const T*>(&zpv);
		const T* rv = reinterpret_cast<const T*>(&e.rv);
		T dv[VX];
#pragma unroll
		for (int j = 0; j < VX; ++j) { dv[j] = dinv_of(e.dv, j); }
		T rz = T(0);
		if (E.mode == 3) {
#pragma unroll
			for (int j = 0; j < VX; ++j) { pz[j] = rv[j] - q[j]; }
		} else if (E.mode == 4) {  // the residual of the LUMPED operator: z_prev carries the lumped data diagonal
#pragma unroll
			for (int j = 0; j < VX; ++j) { pz[j] = rv[j] - (q[j] + zp[j] * zc[j]); }
		} else {
#pragma unroll
			for (int j = 0; j < VX; ++j) { pz[j] = E.a * zc[j] - E.c1 * zp[j] + E.c2 * (dv[j] * (rv[j] - q[j])); }
		}
#pragma unroll
		for (int j = 0; j < VX; ++j) { rz += rv[j] * pz[j]; }
		return rz;
	};

	// ---- prologue ---------------------------------------------------------------------------------
	// Software pipeline, all ring indices relative to the step number s = z - z_begin:
	//   own x values: plane s lives in register slot X[s % 6]; step s reads slots s, s+1, s+2 and issues the
	//                 load of plane s+5 into the slot plane s-1 left (3 steps of lead);
	//   halo values : plane s in H[s % 2]; step s stages plane s+1 and loads plane s+2 (1 step of lead);
	//   row records : layer s in PF[s % 3]; consumed at step s, refilled with layer s+3;
	//   carried     : u1 = masked u(z-1), u2 = masked u(z-2), d1 = masked (x(z) - x(z-1)); with data cells
	//                 also the stencil result of plane z-1, which is completed (data sums added, stored)
	//                 at step s after the barrier.
	// The loop body is instantiated six times so that no ring ever needs a register move (a move would
	// wait for the load it copies and cut the lead to less than one step).
	struct RowPF {
		uint32_t pos;
		T        a[8];
		bool     ok;
	};
	V X0 = load_own(z_begin), X1 = load_own(z_begin + 1), X2 = load_own(z_begin + 2);
	V X3 = load_own(z_begin + 3), X4 = load_own(z_begin + 4), X5 = V{};
	// PRO: the scaling of the same planes, slot for slot; a slot is turned into the operand (form_own) two steps before
	// it becomes the centre plane -- when it is first needed, as x(z+2) of the z stencil -- slots 0 and 1 here
	DV16 Q0{}, Q1{}, Q2{}, Q3{}, Q4{}, Q5{};
	decode_own(X0);  // (slots 2.. are decoded by the step that first uses them, as x(z+2))
	decode_own(X1);
	if (PRO) {
		Q0 = load_own_d(z_begin); Q1 = load_own_d(z_begin + 1); Q2 = load_own_d(z_begin + 2);
		Q3 = load_own_d(z_begin + 3); Q4 = load_own_d(z_begin + 4);
		form_own(X0, Q0);
		form_own(X1, Q1);
	}
	// ring of 2: plane s in H[s % 2]; step s stages plane s+1 and loads plane s+2.  The fused variant is short of
	// registers: there H1 alone carries every plane from z_begin+1 on (the load follows the LDS store of the same step)
	// (round 5: the polynomial's steps that do not form their operand on load have the registers for a halo ring of two --
	// 127 VGPRs, four workgroups per CU as before -- and run 2.5 % faster with it; the fused variants would lose a workgroup per CU)
	constexpr int HR = (EPI && !CELLS && !PRO && sizeof(T) == 4 && FI_HALO_RING == 1) ? 2 : FI_HALO_RING, PR = CELLS ? FI_ROW_RING : 1;
	static_assert(6 % HR == 0 && 6 % PR == 0, "ring depths must divide the 6 instantiations of the step");
	// halo sets: step k (mod 6) consumes set (k + 1) % HR -- plane z_begin + k + 1 -- and refills it with plane + HR
	HaloRegs H0{}, Hr[HR];
	load_halo(z_begin, H0);
#pragma unroll
	for (int k = 0; k < HR; ++k) { load_halo(z_begin + k + 1, Hr[(k + 1) % HR]); }
	if (CELLS) {  // issued after the plane loads so that the two latencies overlap
		for (int i = threadIdx.x; i < 4 * (P.zc + 1) + 1; i += kThreads) {
			const int64_t o = static_cast<int64_t>(wg) * (P.zc + 1) * 4 + i;
			s_lay[0][i] = L.lay_row[o];
			s_lay[1][i] = L.lay_blk[o];
		}
		__syncthreads();
	}
	T U0[VX], U1[VX], U2[VX], D0[VX], D1[VX], held[VX];  // U: u(z) ring of 3, D: d(z) ring of 2
#pragma unroll
	for (int j = 0; j < VX; ++j) { held[j] = T(0); U0[j] = T(0); D0[j] = T(0); }
	const int nsteps = z_end - z_begin;
	EpiRegs EP{};
	if (EPI && !PRO) { load_epi(z_begin, EP); }
	RowPF PFr[PR];  // record sets: step k consumes set (k + 1) % PR -- layer k + 1 -- and refills it with layer + PR
#pragma unroll
	for (int k = 0; k < PR; ++k) { PFr[k].ok = false; }
	// A layer with at most 64 records is scattered by ONE wave (wave s % 4 at step s, all 4 bands, one lane per
	// cell): the other waves skip the code, which matters because the kernel is instruction-issue bound.  That
	// wave prefetches the layer's row records 4 steps ahead.  Denser layers: every wave takes its band, direct loads.
	auto layer_dense = [&](int layer) {
		return (uni(layR[layer * 4 + 4]) - uni(layR[layer * 4])) + (uni(layB[layer * 4 + 4]) - uni(layB[layer * 4])) > static_cast<uint32_t>(P.dense_min);
	};
	// Every wave issues the record loads of the next layer in every step, unconditionally (clamped index):
	// loads that cross a step must sit in straight-line code, or the compiler's s_waitcnt bookkeeping gives up
	// and drains the whole pipeline (vmcnt(0)) at the first use.  The three extra waves hit the same lines in L1.
	auto prefetch_rows = [&](int layer, RowPF& pf) {  // layer index l: cell plane z_begin - 1 + l
#ifdef FI_TIMING_BUILD
		if (P.dbg & 16) { pf.ok = false; return; }
#endif
		const int      lc = layer <= nsteps ? layer : nsteps;
		// dense layers: the first 64 records of this wave's band (fp32 only: the fp64 variant has no registers left
		// for a prefetch that is live across the band loop -- it would spill, and a spill reload drains the pipeline)
		const bool     dense = layer_dense(lc);
		const bool     mine  = kDensePF && dense;
		const uint32_t r0 = uni(layR[lc * 4 + (mine ? band : 0)]), r1 = uni(layR[lc * 4 + (mine ? band + 1 : 4)]);
		const uint32_t r  = r0 + lane;
		pf.ok = layer <= nsteps && (kDensePF || !dense) && r < r1;
#ifdef FI_TIMING_BUILD
		if ((P.dbg & 32) && band != 0) { load_row(r0, &pf.pos, pf.a); return; }
#endif
		load_row(r < r1 ? r : r0, &pf.pos, pf.a);  // r0 <= n_row, and the arrays hold n_row + 1 records
	};
	{
		V xa = load_own(z_begin - 2), xb = load_own(z_begin - 1);
		decode_own(xa);
		decode_own(xb);
		if (PRO) {
			form_own(xa, load_own_d(z_begin - 2));
			form_own(xb, load_own_d(z_begin - 1));
		}
		const T* pa = reinterpret_cast<const T*>(&xa);
		const T* pb = reinterpret_cast<const T*>(&xb);
		const T* pc = reinterpret_cast<const T*>(&X0);
		const T* pd = reinterpret_cast<const T*>(&X1);
		const int g2 = z_begin - 2 + P.zoff, g1 = z_begin - 1 + P.zoff;
		const T mz2 = (g2 >= 0 && g2 + 2 < P.gz) ? T(1) : T(0);
		const T mz1 = (g1 >= 0 && g1 + 2 < P.gz) ? T(1) : T(0);
		const T md1 = (g1 >= 0 && g1 + 1 < P.gz) ? T(1) : T(0);
#pragma unroll
		for (int j = 0; j < VX; ++j) {
			U1[j] = mz2 * (pa[j] - T(2) * pb[j] + pc[j]);  // u(z_begin-2)
			U2[j] = mz1 * (pb[j] - T(2) * pc[j] + pd[j]);  // u(z_begin-1)
			D1[j] = md1 * (pc[j] - pb[j]);                 // d(z_begin-1)
		}
		if (CELLS) {
			// layer z_begin-1 (l = 0): its upper corners sit on plane z_begin (accumulation slot 0); the lower
			// ones belong to the workgroup below and go to the dump slots
			{
				const V zero = V{};
#pragma unroll
				for (int q = 0; q < 6; ++q) { *reinterpret_cast<V*>(&yb[q >> 1][q & 1][ty][VX * tx]) = zero; }
			}
#pragma unroll
			for (int k = 0; k < PR; ++k) { prefetch_rows(k + 1, PFr[(k + 1) % PR]); }
			if (layR[4] > layR[0] || layB[4] > layB[0]) {  // workgroup-uniform
				HaloRegs hprev{};
				load_halo(z_begin - 1, hprev);
				write_plane(2, xb, hprev);  // plane z_begin-1 borrows ring slot 2 (rewritten at step 1)
				write_plane(0, X0, H0);
				__syncthreads();
				cells_scatter(uni(layR[band]), uni(layR[band + 1]), uni(layB[band]), uni(layB[band + 1]), 2, 0, 2, 0, false);
			}
		}
	}
	write_plane(0, X0, H0);

	double dot_acc = 0.0;
	// the last, partly filled group of a row: its points one by one (rows whose length is not a multiple of VX)
	auto store_tail = [&](T* yplane, const T* xv, const T* ov) {
		T part = T(0);
#pragma unroll
		for (int j = 0; j < VX - 1; ++j) {
			if (j < nvalid) {
				yplane[col + j] = ov[j];
				part += xv[j] * ov[j];
			}
		}
		dot_acc += static_cast<double>(part);
	};

	auto step = [&](int s, const V& xc, const V& xp1, V& xp2, V& xload, HaloRegs& h,
	                RowPF& pf,
	                const T* uA, const T* uB, T* uC, const T* dA, T* dC,
	                const DV16& qc, const DV16& q2, DV16& qload) {
		const int z = z_begin + s;
		decode_own(xp2);
		if (PRO) { form_own(xp2, q2); }
		FI_STAMP(s, 0);
		// stage plane z+1 into the LDS ring (needed by the cells of layer z) and refill its halo set HR planes ahead
		write_plane((s + 1) % 3, xp1, h);
		FI_STAMP(s, 1);
		load_halo(z + 1 + HR, h);
		if (!CELLS) { xload = load_own(z + 5); }
		if (PRO) { qload = load_own_d(z + 5); }
		__syncthreads();
		FI_STAMP(s, 2);

		const int gzc = z + P.zoff;
		const int b0 = s % 3, b1 = (s + 1) % 3;
		const T (*pl)[W] = xs[b0];
		const T* pc  = reinterpret_cast<const T*>(&xc);
		const T* pp1 = reinterpret_cast<const T*>(&xp1);
		const T* pp2 = reinterpret_cast<const T*>(&xp2);

		if (CELLS) {
			// plane z-1 is complete: its layers z-2 and z-1 were scattered before this step's barrier
			if (s > 0) {
				T data[VX];
				cells_gather((s + 2) % 3, data);
				const T* pm = reinterpret_cast<const T*>(&xload);  // still x(z-1): reloaded below
				V out;
				T* po = reinterpret_cast<T*>(&out);
				T  dsum = T(0);
#pragma unroll
				for (int j = 0; j < VX; ++j) {
					po[j] = held[j] + data[j];
					dsum += pm[j] * po[j];
				}
				if (EPI) {
					V zn;
					T* pz = reinterpret_cast<T*>(&zn);
					const T rz = epi_full(EP, pm, po, pz);
					if (active) {
						*reinterpret_cast<V*>((E.znew + static_cast<int64_t>(z - 1) * P.plane) + col) = zn;
						dot_acc += static_cast<double>(rz);
					} else if (tail) {
						store_tail(E.znew + static_cast<int64_t>(z - 1) * P.plane, reinterpret_cast<const T*>(&EP.rv), pz);
					}
					load_epi(z, EP);
				} else if (active) {
					*reinterpret_cast<V*>((y + static_cast<int64_t>(z - 1) * P.plane) + col) = out;
					dot_acc += static_cast<double>(dsum);
				} else if (tail) {
					store_tail(y + static_cast<int64_t>(z - 1) * P.plane, pm, po);
				}
			}
			xload = load_own(z + 5);
			FI_STAMP(s, 3);
			// layer z into the accumulation planes of z and z+1
			// (the waves that scatter are the ones the step's barrier waits for: they issue ahead of the other workgroups' waves
			// on their SIMD -- 256^3 fp32 52.3 -> 50.0 us, fp64 105.3 -> 102.8 us per apply)
			__builtin_amdgcn_s_setprio(1);
			if (layer_dense(s + 1)) {
				const int o = (s + 1) * 4 + band;
				if (kDensePF) {
					if (pf.ok) { row_apply(pf.pos, pf.a, b0, b1, b0, b1, true); }
					const uint32_t re = uni(layR[o + 1]), rs = uni(layR[o]) + 64u;  // beyond the prefetched 64: loaded here
					cells_scatter(rs < re ? rs : re, re, uni(layB[o]), uni(layB[o + 1]), b0, b1, b0, b1, true);
				} else {
					cells_scatter(uni(layR[o]), uni(layR[o + 1]), uni(layB[o]), uni(layB[o + 1]), b0, b1, b0, b1, true);
				}
			} else if (band == (s & 3)) {
				const int o = (s + 1) * 4;
				if (pf.ok) { row_apply(pf.pos, pf.a, b0, b1, b0, b1, true); }
				// (row records beyond the 64 prefetched ones -- dense_min may lie above 64 -- are loaded here)
				const uint32_t re = uni(layR[o + 4]), rs = uni(layR[o]) + 64u;
				cells_scatter(rs < re ? rs : re, re, uni(layB[o]), uni(layB[o + 4]), b0, b1, b0, b1, true);
			}
			__builtin_amdgcn_s_setprio(0);
			FI_STAMP(s, 4);
			prefetch_rows(s + 1 + PR, pf);
		}

		FI_STAMP(s, 5);
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
				for (int k = 0; k < VX + 2; ++k) { u[k] = m2x[k] ? (w[k] - T(2) * w[k + 1] + w[k + 2]) : T(0); }
#pragma unroll
				for (int j = 0; j < VX; ++j) { acc2[j] += u[j] - T(2) * u[j + 1] + u[j + 2]; }
			}
			if (HAS1) {
				T d[VX + 1];
#pragma unroll
				for (int k = 0; k < VX + 1; ++k) { d[k] = m1x[k] ? (w[k + 2] - w[k + 1]) : T(0); }  // anchor gx-1+k
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
					acc2[j] += (c2y[0] ? ua : T(0)) - T(2) * (c2y[1] ? ub : T(0)) + (c2y[2] ? uc : T(0));
				}
			}
			if (HAS1) {
#pragma unroll
				for (int j = 0; j < VX; ++j) { acc1[j] += (c1y[0] ? pc[j] - r1[j] : T(0)) - (c1y[1] ? r3[j] - pc[j] : T(0)); }
			}
		}
		// ---- z axis: carried row values
		{
			if (HAS2) {
				const T mz = (gzc >= 0 && gzc + 2 < P.gz) ? T(1) : T(0);
#pragma unroll
				for (int j = 0; j < VX; ++j) {
					const T u0 = mz * (pc[j] - T(2) * pp1[j] + pp2[j]);
					acc2[j] += uA[j] - T(2) * uB[j] + u0;
					uC[j] = u0;
				}
			}
			if (HAS1) {
				const T mz = (gzc + 1 < P.gz) ? T(1) : T(0);
#pragma unroll
				for (int j = 0; j < VX; ++j) {
					const T d0 = mz * (pp1[j] - pc[j]);
					acc1[j] += dA[j] - d0;
					dC[j] = d0;
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
		FI_STAMP(s, 6);
		if (CELLS) {  // completed at the next step, when the data sums of this plane are final
#pragma unroll
			for (int j = 0; j < VX; ++j) { held[j] = po[j]; }
		} else if (EPI) {
			// model diagonal along z (wave-uniform) on top of the thread's x/y part
			T mz = T(0);
			if (HAS2) {
				const int g = gzc;
				mz += C.w2sq * (((g - 2 >= 0 && g < P.gz) ? T(1) : T(0)) + ((g - 1 >= 0 && g + 1 < P.gz) ? T(4) : T(0)) +
				                ((g + 2 < P.gz) ? T(1) : T(0)));
			}
			if (HAS1) { mz += C.w1sq * (((gzc - 1 >= 0) ? T(1) : T(0)) + ((gzc + 1 < P.gz) ? T(1) : T(0))); }
			const V  zpv = epi_zp(EP);
			const T* zp = reinterpret_cast<const T*>(&zpv);
			const T* rv = reinterpret_cast<const T*>(&EP.rv);
			T dv[VX];
#pragma unroll
			for (int j = 0; j < VX; ++j) { dv[j] = dinv_of(PRO ? qc : EP.dv, j); }
			V zn;
			T* pz = reinterpret_cast<T*>(&zn);
			T  rz = T(0);
			if (PRO) {  // z_prev = 0 and Dinv r = z / pro_scale: z_new = a z + c2 (z / pro_scale - s); its r . z_new is not used
				const T inv = T(1) / E.pro_scale;
#pragma unroll
				for (int j = 0; j < VX; ++j) {
					const T sv = dv[j] * (po[j] - (mxy[j] + mz) * pc[j]) + pc[j];
					pz[j] = E.a * pc[j] + E.c2 * (pc[j] * inv - sv);
				}
			} else if (E.mode >= 2) {  // a workgroup without data cells: the model rows are the full operator here
				rz = epi_full(EP, pc, po, pz);
			} else if (E.mode == 1) {
#pragma unroll
				for (int j = 0; j < VX; ++j) {
					pz[j] = po[j] / (mxy[j] + mz);
					rz += pz[j] * pz[j];
				}
			} else {
#pragma unroll
				for (int j = 0; j < VX; ++j) {
					const T sv = dv[j] * (po[j] - (mxy[j] + mz) * pc[j]) + pc[j];
					const T zq = E.zp_scale != T(0) ? E.zp_scale * dv[j] * zp[j] : zp[j];
					pz[j] = E.a * pc[j] - E.c1 * zq + E.c2 * (dv[j] * rv[j] - sv);
					rz += rv[j] * pz[j];
				}
				if (E.acc) {
					const T* av = reinterpret_cast<const T*>(&EP.av);
#pragma unroll
					for (int j = 0; j < VX; ++j) { pz[j] += av[j]; }
					if (E.dotv) {
						const T* bv = reinterpret_cast<const T*>(&EP.bv);
						rz = T(0);
#pragma unroll
						for (int j = 0; j < VX; ++j) { rz += bv[j] * pz[j]; }
					}
				}
			}
#ifdef FI_TIMING_BUILD
			if (E.round16 && E.mode == 0 && sizeof(T) == 4) {
#pragma unroll
				for (int j = 0; j < VX; ++j) {
					if (E.round16 == 1) {
						uint32_t u = __float_as_uint(static_cast<float>(pz[j]));
						u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
						pz[j] = static_cast<T>(__uint_as_float(u));
					} else {
						pz[j] = static_cast<T>(__half2float(__float2half(static_cast<float>(pz[j]))));
					}
				}
			}
#endif
			if (active) {
				if (Z16 && (E.fmt & 4)) {
					*reinterpret_cast<DV16*>((reinterpret_cast<unsigned short*>(E.znew) + static_cast<int64_t>(z) * P.plane) + col) = to16(zn);
				} else {
					*reinterpret_cast<V*>((E.znew + static_cast<int64_t>(z) * P.plane) + col) = zn;
				}
				dot_acc += static_cast<double>(rz);
			} else if (tail) {
				store_tail(E.znew + static_cast<int64_t>(z) * P.plane,
				           (PRO || E.mode == 1) ? pz : ((E.acc && E.dotv) ? reinterpret_cast<const T*>(&EP.bv) : rv), pz);
			}
			if (!PRO) { load_epi(z + 1, EP); }
		} else if (active) {
#ifdef FI_TIMING_BUILD
			if (!(P.dbg & 2))
#endif
			{ *reinterpret_cast<V*>((y + static_cast<int64_t>(z) * P.plane) + col) = out; }
			dot_acc += static_cast<double>(dsum);
		} else if (tail) {
			store_tail(y + static_cast<int64_t>(z) * P.plane, pc, po);
		}
	};

	for (int s0 = 0; s0 < nsteps; s0 += 6) {
		step(s0, X0, X1, X2, X5, Hr[1 % HR], PFr[1 % PR], U1, U2, U0, D1, D0, Q0, Q2, Q5);
		if (s0 + 1 >= nsteps) { break; }
		step(s0 + 1, X1, X2, X3, X0, Hr[2 % HR], PFr[2 % PR], U2, U0, U1, D0, D1, Q1, Q3, Q0);
		if (s0 + 2 >= nsteps) { break; }
		step(s0 + 2, X2, X3, X4, X1, Hr[3 % HR], PFr[3 % PR], U0, U1, U2, D1, D0, Q2, Q4, Q1);
		if (s0 + 3 >= nsteps) { break; }
		step(s0 + 3, X3, X4, X5, X2, Hr[4 % HR], PFr[4 % PR], U1, U2, U0, D0, D1, Q3, Q5, Q2);
		if (s0 + 4 >= nsteps) { break; }
		step(s0 + 4, X4, X5, X0, X3, Hr[5 % HR], PFr[5 % PR], U2, U0, U1, D1, D0, Q4, Q0, Q3);
		if (s0 + 5 >= nsteps) { break; }
		step(s0 + 5, X5, X0, X1, X4, Hr[6 % HR], PFr[6 % PR], U0, U1, U2, D0, D1, Q5, Q1, Q4);
	}
	if (CELLS) {  // the last plane of the chunk: x(z_end-1) sits in ring slot (nsteps-1) % 6
		__syncthreads();
		T data[VX];
		cells_gather((nsteps - 1) % 3, data);
		V xl;
		switch ((nsteps - 1) % 6) {
		case 0: xl = X0; break;
		case 1: xl = X1; break;
		case 2: xl = X2; break;
		case 3: xl = X3; break;
		case 4: xl = X4; break;
		default: xl = X5; break;
		}
		const T* pm = reinterpret_cast<const T*>(&xl);
		V out;
		T* po = reinterpret_cast<T*>(&out);
		T  dsum = T(0);
#pragma unroll
		for (int j = 0; j < VX; ++j) {
			po[j] = held[j] + data[j];
			dsum += pm[j] * po[j];
		}
		if (EPI) {
			V zn;
			T* pz = reinterpret_cast<T*>(&zn);
			const T rz = epi_full(EP, pm, po, pz);
			if (active) {
				*reinterpret_cast<V*>((E.znew + static_cast<int64_t>(z_end - 1) * P.plane) + col) = zn;
				dot_acc += static_cast<double>(rz);
			} else if (tail) {
				store_tail(E.znew + static_cast<int64_t>(z_end - 1) * P.plane, reinterpret_cast<const T*>(&EP.rv), pz);
			}
		} else if (active) {
			*reinterpret_cast<V*>((y + static_cast<int64_t>(z_end - 1) * P.plane) + col) = out;
			dot_acc += static_cast<double>(dsum);
		} else if (tail) {
			store_tail(y + static_cast<int64_t>(z_end - 1) * P.plane, pm, po);
		}
	}

#ifdef FI_STAMPS
	__syncthreads();
#ifndef FI_STAMPS_SEL
#define FI_STAMPS_SEL true  // (which launches report: an expression over the kernel's template flags and arguments)
#endif
	if (wg == (P.dbg >> 8) && (FI_STAMPS_SEL)) {
		for (int i = threadIdx.x; i < 4 * 64 * 8; i += kThreads) { g_stamp[i] = s_stamp[i / 512][i % 512]; }
	}
#endif
	if (partial) {
		const double wsum = wave_sum(dot_acc);
		if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = wsum; }
		__syncthreads();
		if (threadIdx.x == 0) { partial[wg] = red[0] + red[1] + red[2] + red[3]; }
		// the slots of the other chunks of a run: the consumers sum all P.nwg partials
		if (static_cast<int>(threadIdx.x) > 0 && static_cast<int>(threadIdx.x) < nchunk) {
			partial[wg + static_cast<int>(threadIdx.x) * tiles_xy] = 0.0;
		}
	}
}

// Planes per workgroup.  A workgroup pays ~5 planes of pipeline fill and re-reads 4 halo planes of its
// neighbours, so long chunks are cheaper per plane; but the grid should cover the CUs in whole rounds --
// `slots` workgroups run at a time (256 CUs x 4 / 3 / 2 resident workgroups, see the launch bounds).
// Cost model: rounds(zc) * (zc + 5); a fractional last round counts in full while the grid is only a few rounds.
int pick_chunk(int tiles_xy, int nz_own, int slots, int forced, int zc_max)
{
	if (forced > 0) { return forced; }
	if (const char* env = test_switch("FI_ZC")) {
		const int v = atoi(env);
		if (v > 0) { return v > zc_max ? zc_max : v; }
	}
	// small lattices (the coarse levels of a cascade) do not fill the CUs with 4-plane chunks: down to single planes
	// (64^3 with 1 M points: 44 us at 4 planes, 29 us at 2)
	int    best = 1;
	double best_cost = 1e300;
	for (int zc = 1; zc <= zc_max; ++zc) {
		if (zc > nz_own && zc > 1) { break; }
		const int64_t nwg = static_cast<int64_t>(tiles_xy) * ((nz_own + zc - 1) / zc);
		const double  r   = static_cast<double>(nwg) / slots;
		const double  rounds = r < 6.0 ? static_cast<double>((nwg + slots - 1) / slots) : r;
		const double  cost = rounds * (zc + 5);
		if (cost < best_cost * 0.999) {
			best_cost = cost;
			best = zc;
		}
	}
	return best;
}

template <typename T>
bool march_setup(const fi_ctx* c, MarchParams* P, int forced_zc = 0, bool plain = false)
{
	const Geom& g = c->g;
	constexpr int VX = VecOf<T>::VX;
	if (test_switch("FI_NO_MARCH")) { return false; }
	if (g.ndim != 3) { return false; }
	if (g.gn[0] < VX) { return false; }  // rows shorter than one 16-byte group: the plain kernel
	const fi_weights& w = c->w;
	// model_3 / model_4 / gradient_smoothness (field_interpolation.cpp:282-315): this kernel applies the model_0/1/2 rows and
	// the data cells of such a context, k_add_wide3 (fi_operator.hip) adds the wide rows onto its result (MarchState::wide)
	const bool wide = w.model_3 > 0 || w.model_4 > 0 || w.gradient_smoothness > 0;
	if (wide && test_switch("FI_NO_WIDE_MARCH")) { return false; }  // tests: the untiled path (k_apply_generic + colour launches)
	if (!(w.model_1 > 0) && !(w.model_2 > 0) && !wide) { return false; }
	P->nx = g.gn[0];
	P->ny = g.gn[1];
	P->nzl = g.n[2];
	P->gz = g.gn[2];
	P->zoff = g.off[2];
	P->own_z0 = g.own_lo[2];
	P->own_z1 = g.own_hi[2];
	// tile shape: the one whose tiles overhang the lattice least; the wide one on a tie
	{
		double best = 1e300;
		for (int txt : {32, 16}) {
			const int tx = txt * VX, ty = kThreads / txt;
			const double padded = static_cast<double>((P->nx + tx - 1) / tx) * tx * ((P->ny + ty - 1) / ty) * ty;
			if (padded < best * 0.999) {
				best   = padded;
				P->txt = txt;
			}
		}
		if (const char* env = tuning_switch("FI_TXT")) {
			if (atoi(env) == 16 || atoi(env) == 32) { P->txt = atoi(env); }
		}
	}
	const int TX = P->txt * VX;
	P->tx      = TX;
	P->ty      = kThreads / P->txt;
	P->tiles_x = (P->nx + TX - 1) / TX;
	P->tiles_y = (P->ny + P->ty - 1) / P->ty;
	const int nz_own = P->own_z1 - P->own_z0;
	int cus = 256;
	(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
	const bool fused = !plain && c->cells.ncell > 0 && !test_switch("FI_NO_FUSE");
	// resident workgroups per CU.  fp32: what the variant is register-allocated for.  fp64: every fused variant lands at
	// 162-168 VGPRs and 46-48 KB of LDS under its bound of 2 (see fused_waves) -- THREE fit a CU, and a grid sized for two
	// left a third of the wave slots empty: 256^3 fp64 with config 4's data 64 planes x 512 workgroups 107 us, 43 planes x
	// 768 workgroups 92.7 us (profiles/r5_ablation.md section 14; tests/test_kernel_resources.py pins the occupancy)
	// fp32 levels of a V-cycle: their cells are applied by the fused variant WITH the smoother's epilogue (residuals, the
	// full-operator smoother's steps: 2-10 launches per cycle against one plain apply per CG iteration), which is
	// register-allocated for two workgroups per CU -- a grid sized for three ran its last third as a second round
	// (config 4's 128^3 level: 688 workgroups on 512 slots, 48 us per residual)
	const bool epi_level  = fused && sizeof(T) == 4 && c->mg_mode == 1 && !test_switch("FI_NO_EPI_SIZING");
	const int  wgs_per_cu = !fused ? FI_BASE_WAVES
	                               : (sizeof(T) == 8 ? 3 : (epi_level ? 2 : fused_waves<T>(w.model_1 > 0, w.model_2 > 0, c->cells.pack)));
	// the fused variant stages the list bounds of at most 64 + 2 layers in LDS (s_lay); without data cells the chunk
	// may be as long as one round of workgroups allows (512^3: 128 planes, 1024 workgroups)
	P->zc     = pick_chunk(P->tiles_x * P->tiles_y, nz_own, (cus > 0 ? cus : 256) * wgs_per_cu, forced_zc, fused ? 64 : 256);
	P->chunks = (nz_own + P->zc - 1) / P->zc;
	P->nwg    = P->tiles_x * P->tiles_y * P->chunks;
	P->plane  = static_cast<int64_t>(P->nx) * P->ny;
	P->dense_min = c->cells.pack ? FI_DENSE_MIN_PACK : FI_DENSE_MIN_ROWS;
	P->dbg    = 0;
#if defined(FI_TIMING_BUILD) || defined(FI_STAMPS)
	P->dbg    = tuning_switch("FI_DBG") ? atoi(tuning_switch("FI_DBG")) : 0;
#endif
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
void march_launch_cells(fi_ctx* c, const T* x, T* y, double* partial, const uint32_t* wg_list = nullptr, int nlist = 0,
                        const uint32_t* wg_runs = nullptr, const ChebEpi<T>* epi = nullptr)
{
	const MarchParams& P = c->march.P;
	const MarchCoef<T> C = march_coef<T>(c->w);
	const MarchState& m = c->march;
	FI_REQUIRE(!(CELLS && m.no_lists), FI_ERR_STATE, "the marching kernel was asked for the cells of a level that keeps diagonals only");
	FI_REQUIRE(!(CELLS && m.strip_lists), FI_ERR_STATE, "the marching kernel was asked for the cells of a context whose lists are the strip kernel's");
	CellLists L{m.lay_row.as<uint32_t>(), m.lay_blk.as<uint32_t>(), m.pos_row.as<uint32_t>(), m.pos_blk.as<uint32_t>(),
	            m.coef_row.p, m.coef_blk.p};
	const int* done = c->scal.p ? &c->scal.as<CgScalars>()->done : nullptr;
	const int  nrun = wg_list ? nlist : P.nwg;
	if (nrun <= 0) { return; }
	const int  grid = ((nrun + 7) / 8) * 8;
	const bool h1 = c->w.model_1 > 0, h2 = c->w.model_2 > 0;
	auto launch = [&](auto kernel) {
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), 0, c->stream, P, C, L, x, y, partial, done, wg_list, nlist,
		                   wg_runs, epi ? *epi : ChebEpi<T>{});
	};
	const bool pack = CELLS && c->cells.pack;  // the variant that matches the context's block records
	auto pick = [&](auto txt, auto pk, auto ep) {
		constexpr int  TXT = decltype(txt)::value;
		constexpr bool PK  = decltype(pk)::value;
		constexpr bool EP  = decltype(ep)::value;
		if constexpr (EP && CELLS && sizeof(T) == 8) {
			FI_REQUIRE(false, FI_ERR_UNSUPPORTED, "the fused epilogue exists in fp32 only");  // callers check stencil_full_epi_available
		} else {
			if (h1 && h2) {
				launch(k_apply_march3d<T, true, true, CELLS, TXT, PK, EP>);
			} else if (h2) {
				launch(k_apply_march3d<T, false, true, CELLS, TXT, PK, EP>);
			} else {
				launch(k_apply_march3d<T, true, false, CELLS, TXT, PK, EP>);
			}
		}
	};
	using std::integral_constant;
	auto pick_shape = [&](auto ep) {
		if (P.txt == 32) {
			if (CELLS && pack) { pick(integral_constant<int, 32>{}, integral_constant<bool, CELLS>{}, ep); } else { pick(integral_constant<int, 32>{}, integral_constant<bool, false>{}, ep); }
		} else {
			if (CELLS && pack) { pick(integral_constant<int, 16>{}, integral_constant<bool, CELLS>{}, ep); } else { pick(integral_constant<int, 16>{}, integral_constant<bool, false>{}, ep); }
		}
	};
	if (epi) { pick_shape(integral_constant<bool, true>{}); } else { pick_shape(integral_constant<bool, false>{}); }
	FI_HIP_TRY(hipGetLastError());
}

// One Chebyshev step of the polynomial preconditioner: the plain variant with the epilogue, over the whole lattice
// (its own chunking: MarchState::Pplain).  partial: r . z_new per workgroup.
// extend > 0 (slabs, the polynomial's deep exchange): the launch also covers `extend` ghost planes below and above the
// slab -- as far as the lattice goes -- so that the next step finds its operand there without an exchange; the values are
// the neighbour's own, bit for bit (a point's result does not depend on the workgroup that computes it).
MarchParams extended_params(const MarchParams& P0, int extend)
{
	MarchParams P = P0;
	if (extend <= 0) { return P; }
	const int lo_room = P.own_z0 + P.zoff;                 // lattice planes below the slab
	const int hi_room = P.gz - (P.own_z1 + P.zoff);        // ... above it
	const int e_lo = extend < lo_room ? extend : lo_room, e_hi = extend < hi_room ? extend : hi_room;
	P.own_z0 -= e_lo < P.own_z0 ? e_lo : P.own_z0;
	P.own_z1 += e_hi < P.nzl - P.own_z1 ? e_hi : P.nzl - P.own_z1;
	P.chunks = (P.own_z1 - P.own_z0 + P.zc - 1) / P.zc;
	P.nwg    = P.tiles_x * P.tiles_y * P.chunks;
	return P;
}

template <typename T>
void march_launch_epi(fi_ctx* c, const T* z, const ChebEpi<T>& E, double* partial, int part = 0, bool pro = false, int extend = 0)
{
	const MarchParams P = extended_params(c->march.Pplain, extend);
	const MarchCoef<T> C = march_coef<T>(c->w);
	CellLists L{};
	const int* done = c->scal.p ? &c->scal.as<CgScalars>()->done : nullptr;
	// part 1 / 2: the workgroups that read no ghost plane / the first and last z-chunk (slabs: the exchange of the ghost
	// planes overlaps the interior launch)
	const uint32_t* list = part == 1 ? c->march.wgp_inner.as<uint32_t>() : (part == 2 ? c->march.wgp_edge.as<uint32_t>() : nullptr);
	const int nlist = part == 1 ? c->march.np_inner : (part == 2 ? c->march.np_edge : 0);
	if (part != 0 && nlist <= 0) { return; }
	const int  grid = (((part ? nlist : P.nwg) + 7) / 8) * 8;
	const bool h1 = c->w.model_1 > 0, h2 = c->w.model_2 > 0;
	auto launch = [&](auto kernel) {
		hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), 0, c->stream, P, C, L, z, static_cast<T*>(nullptr), partial, done,
		                   list, nlist, static_cast<const uint32_t*>(nullptr), E);
	};
	auto pick = [&](auto txt, auto pr) {
		constexpr int  TXT = decltype(txt)::value;
		constexpr bool PR  = decltype(pr)::value;
		if (h1 && h2) {
			launch(k_apply_march3d<T, true, true, false, TXT, false, true, PR>);
		} else if (h2) {
			launch(k_apply_march3d<T, false, true, false, TXT, false, true, PR>);
		} else {
			launch(k_apply_march3d<T, true, false, false, TXT, false, true, PR>);
		}
	};
	using std::integral_constant;
	if (pro) {
		if (P.txt == 32) { pick(integral_constant<int, 32>{}, integral_constant<bool, true>{}); } else { pick(integral_constant<int, 16>{}, integral_constant<bool, true>{}); }
	} else {
		if (P.txt == 32) { pick(integral_constant<int, 32>{}, integral_constant<bool, false>{}); } else { pick(integral_constant<int, 16>{}, integral_constant<bool, false>{}); }
	}
	FI_HIP_TRY(hipGetLastError());
}

// Surface-type data (an SDF from oriented points) leaves most workgroups without a single cell: those run the
// plain variant (4 workgroups per CU, none of the data path's per-plane work), the others the fused one.
template <typename T>
void march_launch(fi_ctx* c, const T* x, T* y, double* partial, const ChebEpi<T>* epi = nullptr)
{
	const MarchState& m = c->march;
	if (!m.fused) {
		march_launch_cells<T, false>(c, x, y, partial, nullptr, 0, nullptr, epi);
	} else if ((m.P.nwg - m.n_wg_cells) * 16 < m.P.nwg || tuning_switch("FI_NO_SPLIT")) {  // (nearly) every workgroup holds data
		march_launch_cells<T, true>(c, x, y, partial, nullptr, 0, nullptr, epi);
	} else {
		// two launches over disjoint workgroups, back to back (running the data workgroups on a side stream next to
		// the plain ones was tried: 543 instead of 482 us at 512^3 -- the long data columns starve the plain launch)
		march_launch_cells<T, true>(c, x, y, partial, m.wg_cells.as<uint32_t>(), m.n_wg_cells, nullptr, epi);
		march_launch_cells<T, false>(c, x, y, partial, m.wg_plain.as<uint32_t>(), m.n_wg_plain, m.wg_runs.as<uint32_t>(), epi);
	}
}

}  // namespace

void stencil_prepare_slab_lists(fi_ctx* c);

bool stencil_will_fuse(const fi_ctx* c)
{
	if (test_switch("FI_NO_FUSE")) { return false; }
	MarchParams P{};
	return c->dtype == FI_F64 ? march_setup<double>(c, &P) : march_setup<float>(c, &P);
}

void stencil_prepare(fi_ctx* c)
{
	MarchState& m = c->march;
	m.valid = c->dtype == FI_F64 ? march_setup<double>(c, &m.P) : march_setup<float>(c, &m.P);
	m.fused = false;
	m.wide  = m.valid && (c->w.model_3 > 0 || c->w.model_4 > 0 || c->w.gradient_smoothness > 0);
	m.n_row = m.n_blk = 0;
	if (m.valid) {  // chunking of whole-lattice launches of the plain variant (polynomial preconditioner)
		c->dtype == FI_F64 ? march_setup<double>(c, &m.Pplain, 0, true) : march_setup<float>(c, &m.Pplain, 0, true);
	}
	m.n_edge = m.n_inner = m.np_edge = m.np_inner = 0;
	c->tile2.valid = c->tile2.fused = false;
	if (c->g.ndim == 2) { tile2d_prepare(c); }
	if (!m.valid) { return; }
	m.no_lists = false;
	m.strip_lists = false;
	strip_setup(c);
	if (c->strip.valid) {
		// the context's apply runs as wave-private strips (fi_strip.hip): the cell lists are built for THAT decomposition, the
		// marching kernel keeps none (its plain variant still serves the epilogue launches of this context)
		if (c->cells.ncell > 0 && !test_switch("FI_NO_FUSE")) {
			build_cell_lists<double>(c, c->strip);
			c->strip.fused = true;
			m.fused        = true;
			m.strip_lists  = true;
			m.n_row = c->strip.n_row;
			m.n_blk = c->strip.n_blk;
			m.cells_row  = c->strip.cells_row;
			m.cells_blk  = c->strip.cells_blk;
			m.n_wg_cells = m.P.nwg;
			m.n_wg_plain = 0;
		}
		stencil_prepare_slab_lists(c);
		return;
	}
	if (c->cells.ncell > 0 && !test_switch("FI_NO_FUSE") && c->level > 0 && stencil_cheb_direct(c) && stencil_full_direct_wanted(c) &&
	    !test_switch("FI_KEEP_CELL_LISTS")) {
		// A small level of a V-cycle hierarchy: every launch that applies its cells -- residuals, the full-operator smoother,
		// the start's products -- runs as k_full_direct3 on the diagonals fi_levels.hip builds right after this; the marching
		// kernel's per-workgroup cell lists (six kernels, two scans, a host round trip and the copy of every record: 180-200 us
		// of a 64^3 / 32^3 level's assembly chain in config 4) would never be read.  From here on the level's operator is the
		// diagonals', whatever the switches say at solve time (full_direct_now).
		m.fused      = true;
		m.no_lists   = true;
		m.n_wg_cells = m.P.nwg;
		m.n_wg_plain = 0;
		m.cells_row  = 0;
		m.cells_blk  = c->cells.ncell;
	} else if (c->cells.ncell > 0 && !test_switch("FI_NO_FUSE")) {
		c->dtype == FI_F64 ? build_cell_lists<double>(c, m) : build_cell_lists<float>(c, m);
		m.fused = true;
		// Surface-type data: fewer than half of the workgroups hold cells, and those are long latency-bound columns
		// (march_launch runs them in a launch of their own).  Short chunks turn them into 4-8 times as many
		// workgroups: 512^3 SDF data 476 -> 398 us, 256^3 98 -> 62 us.  The lists are rebuilt for the new chunking.
		if (m.n_wg_cells * 2 < m.P.nwg && m.P.zc > 8 && !test_switch("FI_ZC") && !tuning_switch("FI_NO_SPLIT")) {
			c->dtype == FI_F64 ? march_setup<double>(c, &m.P, 8) : march_setup<float>(c, &m.P, 8);
			c->dtype == FI_F64 ? build_cell_lists<double>(c, m) : build_cell_lists<float>(c, m);
		}
	}
	stencil_prepare_slab_lists(c);
}

// Slabs: the workgroups of the first and last z-chunk read ghost planes, the others do not.
void build_edge_lists(fi_ctx* c, const MarchParams& P, DevBuf& edge, DevBuf& inner, int* n_edge, int* n_inner)
{
	const int tiles_xy = P.tiles_x * P.tiles_y;
	const int nz_own = P.own_z1 - P.own_z0;
	const int reach = c->reach > 1 ? c->reach : 1;  // planes a chunk reads beyond its own (short chunks: several chunks deep)
	std::vector<uint32_t> e, in;
	for (int ch = 0; ch < P.chunks; ++ch) {
		const int z0 = ch * P.zc, z1 = (ch + 1) * P.zc < nz_own ? (ch + 1) * P.zc : nz_own;
		const bool ghost = z0 < reach || nz_own - z1 < reach;
		std::vector<uint32_t>& dst = ghost ? e : in;
		for (int t = 0; t < tiles_xy; ++t) { dst.push_back(static_cast<uint32_t>(ch * tiles_xy + t)); }
	}
	*n_edge  = static_cast<int>(e.size());
	*n_inner = static_cast<int>(in.size());
	edge.alloc(sizeof(uint32_t) * (e.size() + 1));
	inner.alloc(sizeof(uint32_t) * (in.size() + 1));
	if (!e.empty()) { FI_HIP_TRY(hipMemcpyAsync(edge.p, e.data(), sizeof(uint32_t) * e.size(), hipMemcpyHostToDevice, c->stream)); }
	if (!in.empty()) { FI_HIP_TRY(hipMemcpyAsync(inner.p, in.data(), sizeof(uint32_t) * in.size(), hipMemcpyHostToDevice, c->stream)); }
	FI_HIP_TRY(hipStreamSynchronize(c->stream));  // the host vectors die here
}

void stencil_prepare_slab_lists(fi_ctx* c)
{
	MarchState& m = c->march;
	if (!m.valid || c->nranks <= 1) { return; }
	build_edge_lists(c, m.P, m.wg_edge, m.wg_inner, &m.n_edge, &m.n_inner);
	build_edge_lists(c, m.Pplain, m.wgp_edge, m.wgp_inner, &m.np_edge, &m.np_inner);
}

// Number of p.q partials the stencil kernel writes, or 0 when the generic kernel must run.
namespace {
bool full_direct_now(const fi_ctx* c);  // (the level's full operator runs as k_full_direct3: below)
}
int stencil_partials(const fi_ctx* c)
{
	if (full_direct_now(c)) { return static_cast<int>((c->g.nloc + kThreads - 1) / kThreads); }  // (k_full_direct3: one per workgroup)
	if (c->march.valid && c->strip.valid) { return c->strip.P.nwg; }  // (fi_strip.hip: one per wave)
	return c->march.valid ? c->march.P.nwg : tile2d_partials(c);
}

bool cells_fused(const fi_ctx* c) { return (c->march.valid && c->march.fused) || (c->tile2.valid && c->tile2.fused); }

#ifdef FI_STAMPS
}  // namespace fi
extern "C" int fi_debug_stamps(unsigned long long* out)
{
	return hipMemcpyFromSymbol(out, HIP_SYMBOL(fi::g_stamp), sizeof(unsigned long long) * 64 * 8 * 4) == hipSuccess ? 0 : 1;
}
namespace fi {
#endif

// ---- the polynomial's step on SMALL levels: one thread per point, neighbours straight from the caches ------------------
// The marching kernel walks a workgroup through zc + 5 plane steps, a barrier each: on a 64^3 level that is 8-9 us per launch
// for 3 MB of traffic, on 32^3 7 us -- the latency of the walk, not work (profiles/r4_step_timeline_c4.txt: the coarse levels
// of a V-cycle cost 405 us of 1.2 ms).  A level of <= 2^19 points runs the same step (ChebEpi mode 0, with or without the
// operand formed on load) as ONE round of loads per point: 13 neighbours through L1 / L2, the same arithmetic (u = S x
// masked from the global coordinates, S^T u, the model diagonal from the same masks).  Undivided fp32 levels, whole lattice,
// fp32 iterates, no dot-product partials (the V-cycle's smoother takes none).
#ifndef FI_DIRECT_MAX_LOG2
#define FI_DIRECT_MAX_LOG2 19
#endif
constexpr int64_t kDirectMaxPoints = int64_t(1) << FI_DIRECT_MAX_LOG2;
__device__ inline float bf16_of(unsigned short v) { return __uint_as_float(static_cast<unsigned int>(v) << 16); }

template <bool HAS1, bool HAS2, bool PRO>
__global__ __launch_bounds__(kThreads) void k_cheb_direct3(int nx, int ny, int nz, MarchCoef<float> C, const float* __restrict__ z,
                                                            ChebEpi<float> E, const int* __restrict__ done)
{
	const int n = nx * ny * nz;  // (<= 2^19: 32-bit index arithmetic)
	const int i = static_cast<int>(blockIdx.x) * kThreads + threadIdx.x;
	if (i >= n) { return; }
	const int cx = i % nx;
	const int t  = i / nx;
	const int cy = t % ny, cz = t / ny;
	const int cc[3] = {cx, cy, cz}, nn[3] = {nx, ny, nz};
	const int st[3] = {1, nx, nx * ny};
	// every load of the step is issued before anything is waited for (the stop flag included: a finished solve stores nothing)
	const int stop = done ? *done : 0;
	float v[3][5];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
#pragma unroll
		for (int k = -2; k <= 2; ++k) {
			if (k == 0 && d > 0) {
				v[d][2] = 0.0f;
				continue;
			}
			if (!HAS2 && (k == -2 || k == 2)) {
				v[d][k + 2] = 0.0f;
				continue;
			}
			const int g = cc[d] + k;
			const int j = i + (g < 0 ? -cc[d] : (g >= nn[d] ? nn[d] - 1 - cc[d] : k)) * st[d];  // clamped: met by a zero mask only
			v[d][k + 2] = PRO ? E.pro_scale * bf16_of(E.dinv[j]) * z[j] : z[j];
		}
	}
	const float dv = bf16_of(E.dinv[i]);
	const float rv = PRO ? 0.0f : E.r[i];
	const float zp = PRO ? 0.0f : E.zprev[i];
	const float av = (!PRO && E.acc) ? E.acc[i] : 0.0f;
	if (stop) { return; }
	const float pc = v[0][2];
	float acc2 = 0.0f, acc1 = 0.0f, m2 = 0.0f, m1 = 0.0f;
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		const int c = cc[d], nd = nn[d];
		const float w[5] = {v[d][0], v[d][1], pc, v[d][3], v[d][4]};
		if (HAS2) {
			// rows anchored at a = c - 2, c - 1, c exist iff 0 <= a and a + 2 < n (field_interpolation.cpp:273)
			const float e0 = (c - 2 >= 0 && c < nd) ? 1.0f : 0.0f, e1 = (c - 1 >= 0 && c + 1 < nd) ? 1.0f : 0.0f, e2 = (c + 2 < nd) ? 1.0f : 0.0f;
			const float u0 = e0 * (w[0] - 2.0f * w[1] + w[2]), u1 = e1 * (w[1] - 2.0f * w[2] + w[3]), u2 = e2 * (w[2] - 2.0f * w[3] + w[4]);
			acc2 += u0 - 2.0f * u1 + u2;
			m2 += e0 + 4.0f * e1 + e2;
		}
		if (HAS1) {
			const float f0 = (c - 1 >= 0) ? 1.0f : 0.0f, f1 = (c + 1 < nd) ? 1.0f : 0.0f;   // rows [-1, +1] anchored at c - 1, c (:265)
			const float d0 = f0 * (w[2] - w[1]), d1 = f1 * (w[3] - w[2]);
			acc1 += d0 - d1;
			m1 += f0 + f1;
		}
	}
	float po = C.w0x3 * pc, m = C.w0x3;
	if (HAS2) {
		po += C.w2sq * acc2;
		m += C.w2sq * m2;
	}
	if (HAS1) {
		po += C.w1sq * acc1;
		m += C.w1sq * m1;
	}
	const float sv = dv * (po - m * pc) + pc;
	float zn;
	if (PRO) {
		zn = E.a * pc + E.c2 * (pc * (1.0f / E.pro_scale) - sv);
	} else {
		const float zq = E.zp_scale != 0.0f ? E.zp_scale * dv * zp : zp;
		zn = E.a * pc - E.c1 * zq + E.c2 * (dv * rv - sv) + av;
	}
	E.znew[i] = zn;
}

bool stencil_cheb_direct(const fi_ctx* c)
{
	return c->march.valid && !c->march.wide && c->dtype == FI_F32 && c->nranks == 1 && c->g.nown == c->g.nloc && c->g.nloc <= kDirectMaxPoints &&
	       !test_switch("FI_NO_DIRECT_STEP");
}

// ---- the FULL operator on small levels: one thread per point, the data rows as 27 diagonals -------------------------------
// A 64^3 level of a cascade holds 4 points per cell, a 32^3 level 30: every cell is a packed 8 x 8 block, and the marching
// kernel walks its workgroups through single planes (zc = 1: 1 + 5 plane steps, every layer's blocks multiplied by the two
// workgroups that share it) -- 35 / 24 / 19 us per residual in config 4's cycle for 36 MB / 5 MB of blocks.  The same
// level's operator as A_model (the 13-point star from masks, as k_cheb_direct3) + 27 diagonals of the data term
// (fi_tail.hip: tail_build_operator, summed in a fixed order when the level is assembled) is one round of coalesced loads
// per point.  ChebEpi modes 2 (the full-operator smoother's step) and 3 (residual); fp32 undivided levels of <= 2^19
// points whose cells are assembled as a level of a hierarchy (fi_levels.hip).  Not the bits of the marching kernel (another
// order of the data term's sums): tests compare iteration counts and solutions (FI_NO_DIRECT_FULL).
template <bool HAS1, bool HAS2>
__global__ __launch_bounds__(kThreads) void k_full_direct3(int nx, int ny, int nz, MarchCoef<float> C, const float* __restrict__ z,
                                                            const float* __restrict__ dia, ChebEpi<float> E, const int* __restrict__ done,
                                                            double* __restrict__ partial)
{
	const int n = nx * ny * nz;
	const int i_raw = static_cast<int>(blockIdx.x) * kThreads + threadIdx.x;
	const bool live = i_raw < n;
	const int i = live ? i_raw : n - 1;  // (every thread stays for the workgroup's sum; the last point stands in for loads)
	const int cx = i % nx;
	const int t  = i / nx;
	const int cy = t % ny, cz = t / ny;
	const int cc[3] = {cx, cy, cz}, nn[3] = {nx, ny, nz};
	const int st[3] = {1, nx, nx * ny};
	const int stop = done ? *done : 0;
	// clamped steps to the neighbours at -1 / +1 along every axis (a clamped value meets a zero diagonal / a zero mask)
	int dm[3], dp[3];
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		dm[d] = cc[d] > 0 ? -st[d] : 0;
		dp[d] = cc[d] + 1 < nn[d] ? st[d] : 0;
	}
	float xb[3][3][3], dg[27];
#pragma unroll
	for (int s = 0; s < 27; ++s) { dg[s] = dia[static_cast<int64_t>(s) * n + i]; }
#pragma unroll
	for (int a = 0; a < 3; ++a) {
#pragma unroll
		for (int b = 0; b < 3; ++b) {
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				const int j = i + (a == 0 ? dm[2] : (a == 2 ? dp[2] : 0)) + (b == 0 ? dm[1] : (b == 2 ? dp[1] : 0)) + (c == 0 ? dm[0] : (c == 2 ? dp[0] : 0));
				xb[a][b][c] = z[j];
			}
		}
	}
	float far[3][2];  // the neighbours at -2 / +2 (model_2)
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		far[d][0] = HAS2 ? z[i + (cc[d] >= 2 ? -2 * st[d] : -cc[d] * st[d])] : 0.0f;
		far[d][1] = HAS2 ? z[i + (cc[d] + 2 < nn[d] ? 2 * st[d] : (nn[d] - 1 - cc[d]) * st[d])] : 0.0f;
	}
	const float dv = E.mode == 2 ? bf16_of(E.dinv[i]) : 0.0f;
	const float rv = E.mode == 5 ? 0.0f : E.r[i];
	const float zp = E.mode == 2 ? E.zprev[i] : 0.0f;
	if (stop) { return; }  // (workgroup-uniform)
	const float pc = xb[1][1][1];
	float acc2 = 0.0f, acc1 = 0.0f;
#pragma unroll
	for (int d = 0; d < 3; ++d) {
		const int c = cc[d], nd = nn[d];
		const float wm = d == 0 ? xb[1][1][0] : (d == 1 ? xb[1][0][1] : xb[0][1][1]);
		const float wp = d == 0 ? xb[1][1][2] : (d == 1 ? xb[1][2][1] : xb[2][1][1]);
		const float w[5] = {far[d][0], wm, pc, wp, far[d][1]};
		if (HAS2) {
			const float e0 = (c - 2 >= 0 && c < nd) ? 1.0f : 0.0f, e1 = (c - 1 >= 0 && c + 1 < nd) ? 1.0f : 0.0f, e2 = (c + 2 < nd) ? 1.0f : 0.0f;
			const float u0 = e0 * (w[0] - 2.0f * w[1] + w[2]), u1 = e1 * (w[1] - 2.0f * w[2] + w[3]), u2 = e2 * (w[2] - 2.0f * w[3] + w[4]);
			acc2 += u0 - 2.0f * u1 + u2;
		}
		if (HAS1) {
			const float f0 = (c - 1 >= 0) ? 1.0f : 0.0f, f1 = (c + 1 < nd) ? 1.0f : 0.0f;
			const float d0 = f0 * (w[2] - w[1]), d1 = f1 * (w[3] - w[2]);
			acc1 += d0 - d1;
		}
	}
	float q = C.w0x3 * pc;
	if (HAS2) { q += C.w2sq * acc2; }
	if (HAS1) { q += C.w1sq * acc1; }
	float data = 0.0f;
#pragma unroll
	for (int s = 0; s < 27; ++s) { data += dg[s] * xb[s / 9][(s / 3) % 3][s % 3]; }
	q += data;
	float zn;
	if (E.mode == 5) {  // the plain product y = A x, with the workgroup's share of x . A x
		zn = q;
	} else if (E.mode == 3) {
		zn = rv - q;
	} else {
		zn = E.a * pc - E.c1 * zp + E.c2 * (dv * (rv - q));
	}
	if (live) { E.znew[i] = zn; }
	if (partial) {
		__shared__ double red[kThreads / 64];
		const double w = wave_sum(live ? static_cast<double>(pc) * static_cast<double>(q) : 0.0);
		if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = w; }
		__syncthreads();
		if (threadIdx.x == 0) { partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3]; }
	}
}

namespace {
bool full_direct_now(const fi_ctx* c)
{
	if (c->march.valid && c->march.no_lists) {  // (decided at assembly: the level has nothing else to apply its cells with)
		FI_REQUIRE(c->dia_valid && c->tail_dia.p, FI_ERR_STATE, "a level without cell lists has lost its diagonals");
		return true;
	}
	// (march.fused: with FI_NO_FUSE the cells go through the kernel of their own behind the model rows, not through the diagonals too)
	return c->dia_valid && c->tail_dia.p && c->march.fused && stencil_cheb_direct(c) && stencil_full_direct_wanted(c);
}
void full_direct_launch(fi_ctx* c, const float* x, const ChebEpi<float>& E, double* partial)
{
	const MarchCoef<float> C = march_coef<float>(c->w);
	const int* done = c->scal.p ? &c->scal.as<CgScalars>()->done : nullptr;
	const int  nx = c->g.n[0], ny = c->g.n[1], nz = c->g.n[2];
	const dim3 grid(static_cast<unsigned>((c->g.nloc + kThreads - 1) / kThreads));
	const bool h1 = c->w.model_1 > 0, h2 = c->w.model_2 > 0;
	auto launch = [&](auto kernel) {
		hipLaunchKernelGGL(kernel, grid, dim3(kThreads), 0, c->stream, nx, ny, nz, C, x, c->tail_dia.as<float>(), E, done, partial);
	};
	if (h1 && h2) {
		launch(k_full_direct3<true, true>);
	} else if (h2) {
		launch(k_full_direct3<false, true>);
	} else {
		launch(k_full_direct3<true, false>);
	}
	FI_HIP_TRY(hipGetLastError());
}
}  // namespace

bool stencil_full_direct_wanted(const fi_ctx* c)
{
	return c->dtype == FI_F32 && c->g.ndim == 3 && c->nranks == 1 && c->g.nown == c->g.nloc && c->g.nloc <= kDirectMaxPoints && c->mg_mode == 1 &&
	       c->cells.ncell > 0 && c->generic.ntrip == 0 && !(c->w.model_3 > 0 || c->w.model_4 > 0 || c->w.gradient_smoothness > 0) &&
	       (c->w.model_1 > 0 || c->w.model_2 > 0) && !test_switch("FI_NO_DIRECT_FULL");
}

// Chebyshev step through the marching kernel (see ChebEpi); false when the kernel does not apply to this context.
bool stencil_cheb_available(const fi_ctx* c) { return (c->march.valid && !c->march.wide) || c->tile2.valid; }  // (2-D: the tile kernel)
int  stencil_cheb_partials(const fi_ctx* c) { return c->march.valid ? c->march.Pplain.nwg : tile2d_partials(c); }
int  stencil_cheb_partials_max(const fi_ctx* c)
{
	if (!c->march.valid) { return tile2d_partials(c); }
	return extended_params(c->march.Pplain, c->nranks > 1 ? c->halo : 0).nwg;
}
void stencil_cheb_step(fi_ctx* c, const void* z, const void* zprev, const void* r, void* znew, double c1, double c2,
                       double* partial, int part, double zprev_scale, double pro_scale, const unsigned short* scaling, int extend, int fmt,
                       const void* acc, const void* dotv)
{
	FI_REQUIRE(!acc || (!c->tile2.valid && pro_scale == 0.0 && !(fmt & 4)), FI_ERR_STATE,
	           "polynomial step onto a vector: 3-D levels, not the step that forms its operand on load, fp32 / fp64 result");
	FI_REQUIRE(!dotv || (acc && partial && !stencil_cheb_direct(c)), FI_ERR_STATE, "b . x partials: the marching kernel's last step onto x");
	FI_REQUIRE(fmt == 0 || (!c->tile2.valid && c->dtype == FI_F32 && c->g.gn[0] % 4 == 0), FI_ERR_STATE,
	           "bfloat16 iterates: fp32 3-D levels with rows of whole 16-byte groups");
	// pro_scale != 0 (the first step, z_prev = 0): `z` is r and the kernel forms z_0 = pro_scale * Dinv * r on load
	// zprev == nullptr: the step from z_prev = 0 (z itself stands in under a zero coefficient);
	// zprev_scale != 0: z_prev = zprev_scale * Dinv r, read through r's own cache lines
	if (c->tile2.valid) {  // 2-D lattices: no operand formed on load, no partial launches
		FI_REQUIRE(pro_scale == 0.0 && part == 0 && extend == 0, FI_ERR_UNSUPPORTED, "2-D polynomial step: stored z_0, whole lattice");
		tile2d_cheb_step(c, z, zprev, r, znew, c1, c2, partial, zprev_scale, scaling);
		return;
	}
	const unsigned short* d16 = scaling ? scaling : c->dinv16.as<unsigned short>();
	const void* zp = zprev_scale != 0.0 ? r : (zprev ? zprev : z);
	const bool  has_prev = zprev_scale != 0.0 || zprev;
	if (!partial && part == 0 && extend == 0 && fmt == 0 && stencil_cheb_direct(c)) {  // small levels: one round of loads per point
		ChebEpi<float> E{static_cast<const float*>(zp), static_cast<const float*>(r), d16, static_cast<float*>(znew),
		                 static_cast<float>(1.0 + c1), static_cast<float>(has_prev ? c1 : 0.0), static_cast<float>(c2), 0,
		                 static_cast<float>(pro_scale), static_cast<float>(zprev_scale), 0, 0, static_cast<const float*>(acc)};
		const MarchCoef<float> C = march_coef<float>(c->w);
		const int* done = c->scal.p ? &c->scal.as<CgScalars>()->done : nullptr;
		const int  nx = c->g.n[0], ny = c->g.n[1], nz = c->g.n[2];
		const dim3 grid(static_cast<unsigned>((c->g.nloc + kThreads - 1) / kThreads));
		const bool h1 = c->w.model_1 > 0, h2 = c->w.model_2 > 0, pro = pro_scale != 0.0;
		const float* zin = static_cast<const float*>(z);
		auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(kThreads), 0, c->stream, nx, ny, nz, C, zin, E, done); };
		if (h1 && h2) {
			pro ? launch(k_cheb_direct3<true, true, true>) : launch(k_cheb_direct3<true, true, false>);
		} else if (h2) {
			pro ? launch(k_cheb_direct3<false, true, true>) : launch(k_cheb_direct3<false, true, false>);
		} else {
			pro ? launch(k_cheb_direct3<true, false, true>) : launch(k_cheb_direct3<true, false, false>);
		}
		FI_HIP_TRY(hipGetLastError());
		return;
	}
	if (c->dtype == FI_F64) {
		ChebEpi<double> E{static_cast<const double*>(zp), static_cast<const double*>(r), d16, static_cast<double*>(znew), 1.0 + c1,
		                  has_prev ? c1 : 0.0, c2, 0, pro_scale, zprev_scale, 0, 0, static_cast<const double*>(acc), static_cast<const double*>(dotv)};
		march_launch_epi<double>(c, static_cast<const double*>(z), E, partial, part, pro_scale != 0.0, extend);
	} else {
		ChebEpi<float> E{static_cast<const float*>(zp), static_cast<const float*>(r), d16, static_cast<float*>(znew),
		                 static_cast<float>(1.0 + c1), static_cast<float>(has_prev ? c1 : 0.0), static_cast<float>(c2), 0,
		                 static_cast<float>(pro_scale), static_cast<float>(zprev_scale), fmt, 0, static_cast<const float*>(acc),
		                 static_cast<const float*>(dotv)};
		if (const char* e = tuning_switch("FI_Z16")) { E.round16 = c->level == 0 ? atoi(e) : 0; }
		march_launch_epi<float>(c, static_cast<const float*>(z), E, partial, part, pro_scale != 0.0, extend);
	}
}
// v_new = (A_model v) / diag(A_model), partials of v_new . v_new (power method on the model operator)
void stencil_power_step(fi_ctx* c, const void* v, void* vnew, double* partial)
{
	if (c->tile2.valid) {
		tile2d_power_step(c, v, vnew, partial);
		return;
	}
	const unsigned short* d16 = c->dinv16.as<unsigned short>();  // loaded, not used
	if (c->dtype == FI_F64) {
		ChebEpi<double> E{static_cast<const double*>(v), static_cast<const double*>(v), d16, static_cast<double*>(vnew), 0, 0, 0, 1, 0, 0};
		march_launch_epi<double>(c, static_cast<const double*>(v), E, partial);
	} else {
		ChebEpi<float> E{static_cast<const float*>(v), static_cast<const float*>(v), d16, static_cast<float*>(vnew), 0, 0, 0, 1, 0, 0};
		march_launch_epi<float>(c, static_cast<const float*>(v), E, partial);
	}
}

// The recurrence step / residual on the FULL operator in one pass over the lattice (ChebEpi modes 2 and 3): the launches
// of stencil_apply with the epilogue.  3-D: fp32 contexts whose data cells (if any) are fused into the marching kernel;
// 2-D: the tile kernel (fi_stencil2d.hip), both precisions.
bool stencil_full_epi_available(const fi_ctx* c)
{
	// rows kept as triplets (GradientKernel::kLinearInterpolation, fi_add_rows_coo) are applied by fi_generic.hip, not by
	// the tiled kernels: a recurrence step in THEIR epilogue would smooth with an operator that lacks those rows
	if (c->generic.ntrip != 0) { return false; }
	if (c->tile2.valid) { return tile2d_full_epi_available(c); }  // 2-D: the tile kernel, both precisions
	return c->march.valid && !c->march.wide && c->dtype == FI_F32 && (c->cells.ncell == 0 || c->march.fused);
}
void stencil_full_step(fi_ctx* c, const void* z, const void* zprev, const void* r, bool residual, void* znew, double a,
                       double c1, double c2)
{
	FI_REQUIRE(stencil_full_epi_available(c), FI_ERR_UNSUPPORTED, "no fused recurrence step for this context");
	if (c->tile2.valid) {
		tile2d_full_step(c, z, zprev, r, residual, znew, a, c1, c2);
		return;
	}
	const float* zf = static_cast<const float*>(z);
	// a null z_prev is never used with its coefficient: z stands in
	ChebEpi<float> E{static_cast<const float*>(zprev ? zprev : z), static_cast<const float*>(r), c->dinv16.as<unsigned short>(),
	                 static_cast<float*>(znew), static_cast<float>(a), static_cast<float>(zprev ? c1 : 0.0),
	                 static_cast<float>(c2), residual ? 3 : 2, 0.0f, 0.0f};
	if (full_direct_now(c)) {  // small levels: the data rows as diagonals
		full_direct_launch(c, zf, E, nullptr);
		return;
	}
	march_launch<float>(c, zf, nullptr, nullptr, &E);
}

// out = b - (A_model x + dlump x): the residual of the lumped operator of a replica without cell records (ChebEpi mode 4)
void stencil_lumped_residual(fi_ctx* c, const void* x, const void* b, void* out)
{
	FI_REQUIRE(c->lumped && c->march.valid && c->dtype == FI_F32 && c->cells.ncell == 0, FI_ERR_UNSUPPORTED,
	           "no lumped operator on this context");
	ChebEpi<float> E{c->dlump.as<float>(), static_cast<const float*>(b), c->dinv16.as<unsigned short>(), static_cast<float*>(out),
	                 0.0f, 0.0f, 0.0f, 4, 0.0f, 0.0f};
	march_launch_epi<float>(c, static_cast<const float*>(x), E, nullptr);
}

bool stencil_apply_part(fi_ctx* c, const void* x, void* y, double* partial, int part)
{
	const MarchState& m = c->march;
	if (c->tile2.valid || !m.valid || c->nranks <= 1 || m.n_inner <= 0) { return false; }
	const bool all_fused = m.fused && ((m.P.nwg - m.n_wg_cells) * 16 < m.P.nwg || tuning_switch("FI_NO_SPLIT"));
	if (m.fused && !all_fused) { return false; }  // surface data: two launches over cell / plain lists already
	const uint32_t* list = part == 1 ? m.wg_inner.as<uint32_t>() : m.wg_edge.as<uint32_t>();
	const int nlist = part == 1 ? m.n_inner : m.n_edge;
	if (c->dtype == FI_F64) {
		if (m.fused) {
			march_launch_cells<double, true>(c, static_cast<const double*>(x), static_cast<double*>(y), partial, list, nlist);
		} else {
			march_launch_cells<double, false>(c, static_cast<const double*>(x), static_cast<double*>(y), partial, list, nlist);
		}
	} else {
		if (m.fused) {
			march_launch_cells<float, true>(c, static_cast<const float*>(x), static_cast<float*>(y), partial, list, nlist);
		} else {
			march_launch_cells<float, false>(c, static_cast<const float*>(x), static_cast<float*>(y), partial, list, nlist);
		}
	}
	return true;
}

bool stencil_apply(fi_ctx* c, const void* x, void* y, double* partial)
{
	if (c->tile2.valid) { return tile2d_apply(c, x, y, partial); }
	if (!c->march.valid) { return false; }
	if (full_direct_now(c)) {
		ChebEpi<float> E{static_cast<const float*>(x), static_cast<const float*>(x), c->dinv16.as<unsigned short>(), static_cast<float*>(y), 0.0f,
		                 0.0f, 0.0f, 5, 0.0f, 0.0f};
		full_direct_launch(c, static_cast<const float*>(x), E, partial);
		return true;
	}
	if (c->strip.valid) {
		strip_apply(c, x, y, partial);
		return true;
	}
	if (c->dtype == FI_F64) {
		march_launch<double>(c, static_cast<const double*>(x), static_cast<double*>(y), partial);
	} else {
		march_launch<float>(c, static_cast<const float*>(x), static_cast<float*>(y), partial);
	}
	return true;
}

}  // namespace fi
