// fi_strip.hip -- the AtA apply of undivided 3-D fp64 lattices as WAVE-PRIVATE STRIPS (round 6): no LDS copy of the lattice
// planes, no barrier.
//
// Reference path replaced: the Eigen CSC SpMV with the explicit AtA inside BiCGSTAB (sparse_linear.cpp:199-206 / :429-436).
// Same operator as fi_stencil.hip (model rows of add_model_constraint, field_interpolation.cpp:257-280, as S^T(S x) with the
// rows that do not exist masked from GLOBAL coordinates; the data rows of field_interpolation.cpp:57-187 as per-cell records),
// another decomposition.  fi_stencil.hip stages every plane in an LDS ring shared by four waves and pays one barrier per
// plane; measured in round 6 (tools/micro/stencil_probe.hip, profiles/r6_ablation.md): at 512^3 that structure stops at
// 0.57 of 8 TB/s in both precisions, a grid-stride COPY reaches 0.61-0.63, and the form below 0.62-0.68:
//   * one WAVE (a 64-thread workgroup) owns a strip of 64 * VX points along x by RY rows of y and marches along z;
//   * z neighbours: a register ring of own planes (3 live + 1 in flight); the row values u(z-1), u(z-2) are carried;
//   * y neighbours: the lane's own RY rows + 4 halo rows (two above, two below) loaded straight from the lines the
//     neighbouring strips read as their own (L1 / L2 hits), a set of halo registers two plane steps ahead;
//   * x neighbours: the neighbouring LANE's values by DPP wave shifts (v_mov_b32 wave_shr:1 / wave_shl:1 with the end
//     lane's `old` operand = a 16-byte halo load of lanes 0 / 63) -- no LDS round trip;
//   * one wave per SIMD (up to 512 registers per lane: the compiler parks ring slots in AccVGPRs): what hides the memory
//     latency is the 8-16 KB every wave keeps in flight, not other waves;
//   * data cells: per (strip, layer) record lists from the same builder as the marching kernel's (fi_stencil_lists.hip, the
//     strip standing in for the workgroup tile: cells on a strip's border are listed by both strips).  One lane per cell:
//     the 8 corner values of x come from global memory (lines this wave or its neighbours have just read), the 8 corner
//     products are ADDED into wave-private LDS accumulation planes [plane ring of 2][corner y-bit][RY][TX] by plain
//     read-add-write in two phases by the corner's x-bit (inside a phase a cell's 4 corners go to 4 different planes and the
//     cells of one instruction are distinct; the LDS instructions of a wave execute in order: every sum is formed in the
//     same order on every run -- bitwise reproducible, no atomics).  The owner collects a plane's sums in the same step,
//     behind the scatter of its second layer: wave-private, so nothing waits for another wave.  One launch whatever the
//     share of strips that hold cells (the marching kernel runs surface-type data as two launches);
//   * x . y partials: fp64 per lane, wave64 shuffle tree, one partial per wave.
// Applies to: 3-D, fp64, one rank (undivided lattice), model_0/1/2 (+ the wide rows through k_add_wide3 as before), rows of
// whole 16-byte groups and at least one full strip wide.  Everything else stays with fi_stencil.hip (FI_NO_STRIP: all of it).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "fi_internal.h"
#include "fi_stencil_common.h"

namespace fi {

namespace {

constexpr int kWave = 64;
constexpr int kOwn  = 4;  // ring of own planes: 3 live + 1 step of lead (tools/micro/stencil_probe.hip: deeper rings are slower)
#ifndef FI_STRIP_HAL
#define FI_STRIP_HAL 2
#endif
#ifndef FI_STRIP_WPS_CELLS
#define FI_STRIP_WPS_CELLS 1  // waves per SIMD the variant with data cells is register-allocated for
#endif
constexpr int kHal  = FI_STRIP_HAL;  // sets of halo registers = steps of lead
constexpr int kMaxZc = 128;  // longest chunk (planes per wave)
constexpr int kRing  = 4;    // accumulation planes of the data term in LDS: the cell wave runs up to kRing - 1 layers ahead

template <typename T>
struct StripCoef {
	T w0x3, w1sq, w2sq;
};

struct StripLists {
	const uint32_t* lay_row;
	const uint32_t* lay_blk;
	const uint32_t* pos_row;
	const uint32_t* pos_blk;
	const void*     coef_row;
	const void*     coef_blk;
};

__device__ inline double strip_wave_sum(double v)
{
	for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o, 64); }
	return v;
}

__host__ __device__ constexpr int stri(int i, int j) { return i * 8 - (i * (i - 1)) / 2 + (j - i); }
__host__ __device__ constexpr int stri_row(int e)
{
	int i = 0;
	while (stri(i + 1, i + 1) <= e && i < 7) { ++i; }
	return i;
}
__host__ __device__ constexpr int stri_col(int e) { return stri_row(e) + (e - stri(stri_row(e), stri_row(e))); }

// lane i takes lane i - 1's (shr) / lane i + 1's (shl) value; the end lane keeps `edge` (bound_ctrl off: `old` survives)
template <typename T, bool SHR>
__device__ inline T lane_shift(T edge, T v)
{
	constexpr int ctrl = SHR ? 0x138 : 0x130;  // wave_shr:1 / wave_shl:1
	if constexpr (sizeof(T) == 8) {
		const long long o = __double_as_longlong(static_cast<double>(edge)), s = __double_as_longlong(static_cast<double>(v));
		const int lo = __builtin_amdgcn_update_dpp(static_cast<int>(o), static_cast<int>(s), ctrl, 0xF, 0xF, false);
		const int hi = __builtin_amdgcn_update_dpp(static_cast<int>(o >> 32), static_cast<int>(s >> 32), ctrl, 0xF, 0xF, false);
		return static_cast<T>(__longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo)));
	} else {
		return static_cast<T>(__int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(static_cast<float>(edge)), __float_as_int(static_cast<float>(v)), ctrl, 0xF, 0xF, false)));
	}
}

// CELLS: a workgroup is TWO waves with fixed roles.  Wave 0 marches the strip (everything above); wave 1 is the strip's CELL
// wave: it walks the strip's record lists layer by layer, up to kRing - 1 layers ahead of the march, and adds the corner
// products into the LDS accumulation ring; the march collects a plane when the cell wave has published its second layer and
// hands the slot back.  Two words in LDS carry the progress (layers done / planes collected); each is written by one wave and
// polled by the other -- no barrier after the prologue, no atomics on the words.  Why two waves: with one wave per SIMD the
// cell path's round trips (a record is an HBM miss, its corner values an L2 round trip, the LDS adds) had nothing to hide
// behind -- the single-wave form cost 0.6 us of every 2.3 us step at 512^3 (profiles/r6_ablation.md); a wave of its own runs
// them ahead of time, beside the march instead of inside it.  Both roles stay under 256 registers: two waves per SIMD.
// PACK: the context keeps its cells of >= 3 rows as packed blocks (CellData::pack: an SDF, the coarse levels of a cascade) --
// the cell wave's pipeline then carries the first 64 BLOCK records of a layer (and fetches row records where it uses them);
// otherwise the first 64 row records (and the rare block records where it uses them).
template <typename T, bool HAS1, bool HAS2, bool CELLS, int RY, bool PACK = false>
__global__ __launch_bounds__(CELLS ? 2 * kWave : kWave, CELLS ? 2 : 1) void k_apply_strip3d(MarchParams P, StripCoef<T> C, StripLists L, const T* __restrict__ x,
                                                            T* __restrict__ y, double* __restrict__ partial, const int* __restrict__ done)
{
	using V = typename VecOf<T>::V;
	constexpr int VX = VecOf<T>::VX;
	constexpr int TX = kWave * VX;
	constexpr int U  = (kOwn % kHal == 0) ? kOwn : kOwn * kHal;  // instantiations of the step: every ring index a constant
	typedef T PairU __attribute__((ext_vector_type(2), aligned(8)));  // two neighbouring points at any even / odd column

	// accumulation planes of the data term: a ring of kRing lattice planes [RY][TX], and the write-only target of corner
	// products that fall outside the strip
	// (with a ring of halo points -- rows -1 .. RY, columns -1 .. TX at column index + 2, so that the strip's own points sit at
	// 16-byte offsets: EVERY corner of every listed cell has a slot, the cell wave adds without a bounds test, and the march
	// collects the strip's own points only; what gathers in the halo belongs to the neighbouring strips' own lists)
	constexpr int PW = TX + 4, PS = (RY + 2) * PW;  // row pitch and plane stride of the ring, in points
	__shared__ __attribute__((aligned(16))) T yb[CELLS ? kRing * PS : VX];
	__shared__ T ydump[CELLS ? PW + 8 : 1];
	// this strip's record bounds, one pair per layer (a strip takes all 4 bands of a layer as one list): staged once
	__shared__ uint32_t s_lay[2][CELLS ? kMaxZc + 3 : 1];
	// (relaxed workgroup-scope atomics on plain LDS words, NOT volatile: the address-space inference leaves volatile accesses
	// generic, and a FLAT load / store of the word drained every global load in flight -- s_waitcnt vmcnt(0) -- twice per
	// plane step in both waves: the first form of this kernel lost 40 % to that)
	__shared__ int s_flag[2];  // [0]: layers the cell wave has finished; [1]: planes the march has collected
	__shared__ int s_role;              // the role of wave 0 (0: the march)

	if (done && *done) { return; }
	// XCD-aware order: blocks b, b + 8, ... (one XCD, one L2) take neighbouring strips
	const int per  = (P.nwg + 7) / 8;
	const int slot = (blockIdx.x % 8) * per + blockIdx.x / 8;
	if (slot >= P.nwg) { return; }
	const int lane = threadIdx.x & (kWave - 1);
	const int wave = CELLS ? __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6) : 0;
	const int strips_xy = P.tiles_x * P.tiles_y;
	const int chunk = slot / strips_xy, sxy = slot % strips_xy;
	const int x0 = (sxy % P.tiles_x) * TX, y0 = (sxy / P.tiles_x) * RY;
	const int gx = x0 + VX * lane;
	const bool lane_ok = gx + VX <= P.nx;  // (rows are whole 16-byte groups: a lane's points are all inside or all outside)
	const int z_begin = P.own_z0 + chunk * P.zc;
	const int z_end   = z_begin + P.zc < P.own_z1 ? z_begin + P.zc : P.own_z1;
	const int nsteps  = z_end - z_begin;
	auto uni = [](uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); };
	auto row_lo = [&](int l) { return uni(s_lay[0][l]); };
	auto blk_lo = [&](int l) { return uni(s_lay[1][l]); };
	auto wait_ge = [&](int which, int v) {  // until the other wave's progress word has reached v
		while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_flag[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < v) {
			__builtin_amdgcn_s_sleep(2);
		}
		asm volatile("" ::: "memory");  // (nothing that follows is read or added ahead of the word)
	};
	bool any_cells = false;
	if (CELLS) {
		const int64_t lay0 = static_cast<int64_t>(slot) * (P.zc + 1) * 4;  // this strip's (layer, band) bounds
		for (int i = threadIdx.x; i <= P.zc + 2; i += 2 * kWave) {  // (entry zc + 2: the last bound once more)
			const int k = i <= P.zc + 1 ? i : P.zc + 1;
			s_lay[0][i] = L.lay_row[lay0 + 4 * k];
			s_lay[1][i] = L.lay_blk[lay0 + 4 * k];
		}
		const V zero = V{};
		for (int i = threadIdx.x; i < kRing * PS / VX; i += 2 * kWave) { *reinterpret_cast<V*>(&yb[0] + VX * i) = zero; }
		if (threadIdx.x == 0) {
			s_flag[0] = 0;
			s_flag[1] = 0;
			// Which wave marches: the hardware hands a workgroup's two waves to neighbouring SIMDs, the same pair for every
			// other workgroup of the CU -- with fixed roles the four marches of a CU (the instruction-heavy role) would share
			// two SIMDs and the four cell waves idle on the other two.  Wave 0 looks at where it sits (SIMD id and wave slot
			// from HW_ID) and takes the march when their parities agree; wave 1 takes the other role, whatever IT would see.
#ifdef FI_STRIP_ROLES_BY_SIMD
			const int simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);   // HW_REG_HW_ID[5:4]
			const int wslot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID[3:0]
			s_role = (simd ^ wslot) & 1;
#else
			s_role = 0;
#endif
		}
		__syncthreads();
		any_cells = row_lo(nsteps + 1) > row_lo(0) || blk_lo(nsteps + 1) > blk_lo(0);
	}
	const int role = CELLS ? (wave == 0 ? s_role : 1 - s_role) : 0;  // 0: the march, 1: the cell wave

	if (CELLS && role == 1) {
		// ================================ the cell wave ================================================================
		if (!any_cells) { return; }
#if defined(FI_STRIP_DBG) && (FI_STRIP_DBG & 32)  // timing builds: no cell wave at all
		return;
#endif
		// the 8 corner values of x of the cell with strip-relative origin (tcx, tcy) on local planes lz, lz + 1: four pairs.  A
		// cell on the lattice's far faces (the one-point rows of nearest-neighbour constraints, field_interpolation.cpp:82-107)
		// or at extended origin -1 has corners OUTSIDE the lattice under zero coefficients: every address is clamped into the
		// lattice (a clamped value only has to be finite)
		// LOADS only (raw pairs, clamped addresses, straight-line code): the selection of the pair's elements happens where the
		// values are used (corners_fix) -- a select right behind the load is a use, and the wave would wait for the load it has
		// just issued and, in order, for every older one
		auto corners_issue = [&](uint32_t pos, int lz, T* raw) {
#if defined(FI_STRIP_DBG) && (FI_STRIP_DBG & 512)  // timing builds: no corner loads
			for (int q = 0; q < 8; ++q) { raw[q] = T(lane + q + lz); }
			return;
#endif
			const int tcx = static_cast<int>(pos & 0xFFu) - 1, tcy = static_cast<int>((pos >> 16) & 0xFFu) - 1;
			const int cx = x0 + tcx, cy = y0 + tcy;
			const int bx = cx < 0 ? 0 : (cx > P.nx - 2 ? P.nx - 2 : cx);  // the pair's first column
#pragma unroll
			for (int bz = 0; bz < 2; ++bz) {
				const int pz = lz + bz < 0 ? 0 : (lz + bz > P.nzl - 1 ? P.nzl - 1 : lz + bz);
				const T* p = x + static_cast<int64_t>(pz) * P.plane;
#pragma unroll
				for (int by = 0; by < 2; ++by) {
					const int ry = cy + by < 0 ? 0 : (cy + by > P.ny - 1 ? P.ny - 1 : cy + by);
					const PairU v = *reinterpret_cast<const PairU*>(p + (static_cast<uint32_t>(ry) * P.nx + static_cast<uint32_t>(bx)));
					raw[4 * bz + 2 * by]     = v[0];
					raw[4 * bz + 2 * by + 1] = v[1];
				}
			}
		};
		auto corners_fix = [&](uint32_t pos, const T* raw, T* xv) {
			const int cx = x0 + static_cast<int>(pos & 0xFFu) - 1;
			const bool left = cx < 0, right = cx > P.nx - 2;  // the pair was read one column to the right / to the left
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				xv[2 * k]     = right ? raw[2 * k + 1] : raw[2 * k];
				xv[2 * k + 1] = left ? raw[2 * k] : raw[2 * k + 1];
			}
		};
		auto corners = [&](uint32_t pos, int lz, T* xv) {
			T raw[8];
			corners_issue(pos, lz, raw);
			corners_fix(pos, raw, xv);
		};
		// add the 8 corner products of a cell of layer l into the ring slots of planes l - 1 (corner z-bit 0) and l (z-bit 1);
		// every corner has a slot (the ring's halo); the corners beyond the chunk's ends go to a dump area.  Plain LDS read-add-write in four
		// phases by the corner's (x-bit, y-bit): inside a phase a cell's two corners go to two different planes and the cells
		// of one instruction are distinct -- no two lanes meet; across phases the LDS instructions of a wave execute in
		// program order.  Every sum is formed in the same order on every run: bitwise reproducible.  (LDS hardware adds,
		// ds_add_f64, were measured first: 8 x 64 lane-operations per batch and cell wave go through the CU's atomic unit
		// one after the other -- 290 us of a 512^3 apply, profiles/r6_ablation.md.  The four round trips of this form are
		// latency of the cell wave alone, which runs ahead of the march.)
		auto put8 = [&](uint32_t pos, const T* out, int l, bool on) {
			const int tcx = static_cast<int>(pos & 0xFFu) - 1, tcy = static_cast<int>((pos >> 16) & 0xFFu) - 1;
#if defined(FI_STRIP_DBG) && (FI_STRIP_DBG & 2)  // timing builds: no adds
			ydump[lane & 7] = out[0] + out[1] + out[2] + out[3] + out[4] + out[5] + out[6] + out[7];
			return;
#endif
			// the slots of the cell's lower and upper corners (corner y-bit and x-bit are immediate offsets from them); a lane
			// that is off -- the second row of a pair, merged into its neighbour -- and the corners beyond the chunk's ends go
			// to the dump area (any number of lanes may meet there)
#if defined(FI_STRIP_DBG) && (FI_STRIP_DBG & 256)  // timing builds: the adds at lane-linear addresses (no bank conflicts)
			T* const at = &yb[0] + PW + 2 + 2 * lane + 0 * (tcx + tcy);
#else
			T* const at = &yb[0] + (tcy + 1) * PW + (tcx + 2);
#endif
			T* const lo = (on && l > 0) ? at + ((l + kRing - 1) % kRing) * PS : &ydump[0];
			T* const hi = (on && l < nsteps) ? at + (l % kRing) * PS : &ydump[0];
#pragma unroll
			for (int ph = 0; ph < 4; ++ph) {
				const int bx = ph & 1, by = ph >> 1;
				asm volatile("" ::: "memory");
				T* d0 = lo + by * PW + bx;
				T* d1 = hi + by * PW + bx;
				const T c0 = *d0, c1 = *d1;
				*d0 = c0 + out[bx + 2 * by];
				*d1 = c1 + out[bx + 2 * by + 4];
			}
			asm volatile("" ::: "memory");
		};
		auto row_finish = [&](uint32_t pos, const T* a, const T* xv, int l) {
			T t = T(0);
#pragma unroll
			for (int q = 0; q < 8; ++q) { t += a[q] * xv[q]; }
			T out[8];
#pragma unroll
			for (int i = 0; i < 8; ++i) { out[i] = a[i] * t; }
			// the second row of a two-row cell sits in the lane next to the first and goes to the same addresses: the first row's
			// lane takes its products over a DPP shift and adds both at once (a second row in lane 0 has its partner in the
			// previous batch of 64: it adds by itself)
			const bool second = ((pos >> 8) & 0xFFu) != 0u;
			if (__ballot(second) != 0ull) {
				const int nxt = __builtin_amdgcn_update_dpp(0, second ? 1 : 0, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					const T o = lane_shift<T, false>(T(0), out[i]);
					if (nxt) { out[i] += o; }
				}
			}
			put8(pos, out, l, !(second && lane > 0));
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
		struct RowRec {
			uint32_t pos;
			T        a[8];
			bool     ok;
		};
		// the first 64 row records of layer l (clamped index: the loads of the pipeline are unconditional)
		auto fetch_rows = [&](int l, RowRec& rec) {
			const int      lc = l <= nsteps ? l : nsteps;
			const uint32_t r0 = row_lo(lc), r1 = row_lo(lc + 1);
			const uint32_t r  = r0 + lane;
			rec.ok = l <= nsteps && r < r1;
			load_row(r < r1 ? r : r0, &rec.pos, rec.a);  // (r0 <= n_row, and the arrays hold n_row + 1 records)
		};
		auto fetch_blk_pos = [&](int l, uint32_t* pos, bool* ok) {
			const int      lc = l <= nsteps ? l : nsteps;
			const uint32_t b0 = blk_lo(lc), b1 = blk_lo(lc + 1);
			const uint32_t r  = b0 + lane;
			*ok  = l <= nsteps && r < b1;
			*pos = L.pos_blk[r < b1 ? r : b0];
		};
		const T* multi = static_cast<const T*>(L.coef_blk);
		// out = (the cell's block) x for block record r: the packed symmetric block (36 coefficients, all asked for at once) or
		// the cell's factor rows one after another
		auto block_apply = [&](uint32_t r, uint32_t pos, const T* xv, int l, const T* first8) {
			const int nrows = static_cast<int>((pos >> 8) & 0xFFu);
			T out[8];
#pragma unroll
			for (int i = 0; i < 8; ++i) { out[i] = T(0); }
			const uint32_t ro = r * 64u;
			auto mul8 = [&](int e0, const T* w, int n) {  // packed coefficients e0 .. e0 + n - 1 (constants once unrolled)
#pragma unroll
				for (int e = 0; e < 12; ++e) {
					if (e < n) {
						const int i = stri_row(e0 + e), j = stri_col(e0 + e);
						out[i] += w[e] * xv[j];
						if (i != j) { out[j] += w[e] * xv[i]; }
					}
				}
			};
			if (nrows == 0xFF) {
				// the packed symmetric block, out = B x: 36 coefficients in batches (all at once do not fit beside the pipeline's
				// register sets); the first 8 may have travelled with the pipeline, their lines' neighbours touched
				T w[12];
				if (first8) {
					mul8(0, first8, 8);
				} else {
#pragma unroll
					for (int v = 0; v < 8 / VX; ++v) { *reinterpret_cast<V*>(&w[v * VX]) = *reinterpret_cast<const V*>(multi + (ro + static_cast<uint32_t>(v * VX))); }
					mul8(0, w, 8);
				}
#pragma unroll
				for (int v = 0; v < 12 / VX; ++v) { *reinterpret_cast<V*>(&w[v * VX]) = *reinterpret_cast<const V*>(multi + (ro + static_cast<uint32_t>(8 + v * VX))); }
				mul8(8, w, 12);
#pragma unroll
				for (int v = 0; v < 12 / VX; ++v) { *reinterpret_cast<V*>(&w[v * VX]) = *reinterpret_cast<const V*>(multi + (ro + static_cast<uint32_t>(20 + v * VX))); }
				mul8(20, w, 12);
#pragma unroll
				for (int v = 0; v < 4 / VX; ++v) { *reinterpret_cast<V*>(&w[v * VX]) = *reinterpret_cast<const V*>(multi + (ro + static_cast<uint32_t>(32 + v * VX))); }
				mul8(32, w, 4);
			} else {
				const V* ap = reinterpret_cast<const V*>(multi + static_cast<int64_t>(r) * 64);
				for (int k = 0; k < nrows; ++k) {  // out += a (a . x)
					T a[8];
#pragma unroll
					for (int v = 0; v < 8 / VX; ++v) {
						const V  w2 = ap[k * (8 / VX) + v];
						const T* pw = reinterpret_cast<const T*>(&w2);
#pragma unroll
						for (int j = 0; j < VX; ++j) { a[v * VX + j] = pw[j]; }
					}
					T t = T(0);
#pragma unroll
					for (int q = 0; q < 8; ++q) { t += a[q] * xv[q]; }
#pragma unroll
					for (int i = 0; i < 8; ++i) { out[i] += a[i] * t; }
				}
			}
			put8(pos, out, l, true);
		};
		// Pipeline over LAYERS (the march consumes one per step; the ring lets this wave ADD kRing - 1 layers ahead, and ask
		// for data as far ahead as it likes).  A record's origin word decides the addresses of its corner values, and a wave
		// that waits for a load waits for every older one: with one layer between the origins' load and the corners' load the
		// wave advanced one layer per memory round trip and the march waited for it (profiles/r6_ablation.md).  So: the
		// origins of the first 64 row records and of the first 64 block records of layer l + 4, then the coefficients and
		// the corner values of layer l + 2, then layer l is multiplied and added.  Six origin sets and three data sets take
		// the roles in turn -- the loop body is instantiated six times: a register MOVE of a set in flight would wait for
		// its loads.  Records beyond the first 64 of a layer (dense layers: the polar caps of an SDF, coarse lattices) are
		// fetched where they are used.
		struct Org {
			uint32_t pos;
			bool     ok;
		};
		struct Dat {
			T a[8], xv[8];
			uint32_t touch[2];
		};
		Org O0{}, O1{}, O2{}, O3{}, O4{}, O5{};
		Dat D0{}, D1{}, D2{};
		// the piped kind's record of this lane in layer l: index (clamped: the pipeline's loads are unconditional) and presence
		auto piped = [&](int l, uint32_t* r, bool* ok) {
			const int      lc = l <= nsteps ? l : nsteps;
			const uint32_t r0 = PACK ? blk_lo(lc) : row_lo(lc), r1 = PACK ? blk_lo(lc + 1) : row_lo(lc + 1);
			*ok = l <= nsteps && r0 + lane < r1;
			*r  = r0 + lane < r1 ? r0 + lane : r0;  // (r0 <= the number of records, and the arrays hold one more)
		};
		auto fetch_org = [&](int l, Org& o) {
			uint32_t r;
			piped(l, &r, &o.ok);
			o.pos = PACK ? L.pos_blk[r] : L.pos_row[r];
		};
		auto fetch_dat = [&](int l, const Org& o, Dat& d) {
			uint32_t r;
			bool     ok;
			piped(l, &r, &ok);
#if defined(FI_STRIP_DBG) && (FI_STRIP_DBG & 1024)  // timing builds: every lane reads the layer's first record
			r = r - lane * 0 - (r - r) ;
			const T* rec = PACK ? multi + static_cast<int64_t>(__builtin_amdgcn_readfirstlane(r)) * 64 : static_cast<const T*>(L.coef_row) + static_cast<int64_t>(__builtin_amdgcn_readfirstlane(r)) * 8;
#else
			const T* rec = PACK ? multi + static_cast<int64_t>(r) * 64 : static_cast<const T*>(L.coef_row) + static_cast<int64_t>(r) * 8;
#endif
			const V* ap = reinterpret_cast<const V*>(rec);
#pragma unroll
			for (int k = 0; k < 8 / VX; ++k) {
				const V  v  = ap[k];
				const T* pv = reinterpret_cast<const T*>(&v);
#pragma unroll
				for (int j = 0; j < VX; ++j) { d.a[k * VX + j] = pv[j]; }
			}
			if (PACK) {  // the block's other lines on their way to the L2 (288 bytes = the first line and up to two more)
				d.touch[0] = *reinterpret_cast<const uint32_t*>(rec + 128 / sizeof(T));
				d.touch[1] = *reinterpret_cast<const uint32_t*>(rec + 256 / sizeof(T));
			}
			corners_issue(o.pos, z_begin - 1 + l, d.xv);
		};
		auto stage = [&](int l, Org& oNew, const Org& oMid, Dat& dMid, const Org& o, const Dat& d) {
			const int lz = z_begin - 1 + l;  // local plane of the layer's origins
			const int lq = l <= nsteps ? l : nsteps + 1;  // (a layer beyond the chunk's last: the empty range at the lists' end)
			fetch_org(l + 4, oNew);
			fetch_dat(l + 2, oMid, dMid);
			// the slot of plane l must have been collected (plane l - kRing): the march is at most kRing - 1 planes behind
#if !(defined(FI_STRIP_DBG) && (FI_STRIP_DBG & 64))  // timing builds: the cell wave does not wait for the march
			wait_ge(1, (l <= nsteps ? l : nsteps) - kRing + 1);
#endif
			T xvf[8];
			corners_fix(o.pos, d.xv, xvf);
			if (!PACK && o.ok) { row_finish(o.pos, d.a, xvf, l); }
#if !(defined(FI_STRIP_DBG) && (FI_STRIP_DBG & 128))  // timing builds: no records beyond the piped ones
			{
				const uint32_t rs = row_lo(lq) + (PACK ? 0u : 64u), re = row_lo(lq + 1);
				for (uint32_t r = rs + lane; r < re; r += 64) {
					RowRec rr;
					load_row(r, &rr.pos, rr.a);
					T xv[8];
					corners(rr.pos, lz, xv);
					row_finish(rr.pos, rr.a, xv, l);
				}
			}
			const uint32_t bs = blk_lo(lq), be = blk_lo(lq + 1);
			if (PACK && o.ok) {
				asm volatile("" ::"v"(d.touch[0]), "v"(d.touch[1]));  // (the touches are loads somebody waits for)
				block_apply(bs + lane, o.pos, xvf, l, d.a);
			}
			for (uint32_t r = bs + (PACK ? 64u : 0u) + lane; r < be; r += 64) {
				const uint32_t pos = L.pos_blk[r];
				T xv[8];
				corners(pos, lz, xv);
				block_apply(r, pos, xv, l, nullptr);
			}
#else
			const uint32_t bs = blk_lo(lq);
			if (PACK && o.ok) { block_apply(bs + lane, o.pos, xvf, l, d.a); }
#endif
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the layer's adds have been performed
			if (l <= nsteps) { __hip_atomic_store(&s_flag[0], l + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
		};
		fetch_org(0, O0);
		fetch_org(1, O1);
		fetch_org(2, O2);
		fetch_org(3, O3);
		fetch_dat(0, O0, D0);
		fetch_dat(1, O1, D1);
		// (whole rounds of six: layers beyond the chunk's last find no records and wait for nothing -- a loop body without
		// exits keeps the compiler's count of the loads in flight exact)
		for (int l0 = 0; l0 <= nsteps; l0 += 6) {
			stage(l0, O4, O2, D2, O0, D0);
			stage(l0 + 1, O5, O3, D0, O1, D1);
			stage(l0 + 2, O0, O4, D1, O2, D2);
			stage(l0 + 3, O1, O5, D2, O3, D0);
			stage(l0 + 4, O2, O0, D0, O4, D1);
			stage(l0 + 5, O3, O1, D1, O5, D2);
		}
		return;
	}

	// ==================================== the march =========================================================================
	// ---- masks: rows that do not exist (global coordinates) -----------------------------------------------------------
	bool m2x[VX + 2], m1x[VX + 1];
#pragma unroll
	for (int k = 0; k < VX + 2; ++k) {
		const int a = gx - 2 + k;
		m2x[k] = HAS2 && a >= 0 && a + 2 < P.nx;
	}
#pragma unroll
	for (int k = 0; k < VX + 1; ++k) {
		const int a = gx - 1 + k;
		m1x[k] = HAS1 && a >= 0 && a + 1 < P.nx;
	}
	T c2y[RY][3], c1y[RY][2];  // wave-uniform
	bool row_ok[RY];
#pragma unroll
	for (int j = 0; j < RY; ++j) {
		const int gy = y0 + j;
		row_ok[j] = gy < P.ny;
		c2y[j][0] = (HAS2 && gy - 2 >= 0 && gy < P.ny) ? T(1) : T(0);
		c2y[j][1] = (HAS2 && gy - 1 >= 0 && gy + 1 < P.ny) ? T(1) : T(0);
		c2y[j][2] = (HAS2 && gy + 2 < P.ny) ? T(1) : T(0);
		c1y[j][0] = (HAS1 && gy - 1 >= 0 && gy < P.ny) ? T(1) : T(0);
		c1y[j][1] = (HAS1 && gy + 1 < P.ny) ? T(1) : T(0);
	}
	// ---- element offsets inside a plane (32-bit), clamped into the lattice: a clamped value only ever meets a zero mask ----
	uint32_t own_off[RY], hy_off[4], hx_off[RY];
	{
		const int gxc = lane_ok ? gx : (P.nx - VX);
#pragma unroll
		for (int j = 0; j < RY; ++j) {
			const int gy = y0 + j < P.ny ? y0 + j : P.ny - 1;
			own_off[j] = static_cast<uint32_t>(gy) * P.nx + gxc;
			int hx = lane == kWave - 1 ? x0 + TX : x0 - 2;  // (lanes 1 .. 62 read lane 0's address: the same line, no branch)
			hx = hx < 0 ? 0 : (hx > P.nx - 2 ? P.nx - 2 : hx);
			hx_off[j] = static_cast<uint32_t>(gy) * P.nx + hx;
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			int hy = k < 2 ? y0 - 2 + k : y0 + RY + (k - 2);
			hy = hy < 0 ? 0 : (hy >= P.ny ? P.ny - 1 : hy);
			hy_off[k] = static_cast<uint32_t>(hy) * P.nx + gxc;
		}
	}
	const int lz_lo = P.zoff < 0 ? -P.zoff : 0;
	const int lz_hi = (P.nzl < P.gz - P.zoff ? P.nzl : P.gz - P.zoff) - 1;
	auto plane_of = [&](int lz) { return x + static_cast<int64_t>(lz < lz_lo ? lz_lo : (lz > lz_hi ? lz_hi : lz)) * P.plane; };

	struct Halo {
		V     hy[4];
		PairU hx[RY];
	};
	V    X[kOwn][RY];
	Halo H[kHal];
	auto load_own = [&](int lz, V* dst) {
		const T* p = plane_of(lz);
#pragma unroll
		for (int j = 0; j < RY; ++j) { dst[j] = *reinterpret_cast<const V*>(p + own_off[j]); }
	};
	auto load_halo = [&](int lz, Halo& h) {
		const T* p = plane_of(lz);
#pragma unroll
		for (int k = 0; k < 4; ++k) { h.hy[k] = *reinterpret_cast<const V*>(p + hy_off[k]); }
#pragma unroll
		for (int j = 0; j < RY; ++j) { h.hx[j] = *reinterpret_cast<const PairU*>(p + hx_off[j]); }
	};

	// ---- prologue -----------------------------------------------------------------------------------------------------
	T U1[RY][VX], U2[RY][VX], D1[RY][VX];  // masked u(z-2), u(z-1); masked d(z-1) = x(z) - x(z-1)
	{
		V a[RY], b[RY];
		load_own(z_begin - 2, a);
		load_own(z_begin - 1, b);
#pragma unroll
		for (int k = 0; k < kOwn - 1; ++k) { load_own(z_begin + k, X[k]); }
#pragma unroll
		for (int k = 0; k < kHal; ++k) { load_halo(z_begin + k, H[k]); }
		const int g2 = z_begin - 2 + P.zoff, g1 = z_begin - 1 + P.zoff;
		const T mz2 = (HAS2 && g2 >= 0 && g2 + 2 < P.gz) ? T(1) : T(0);
		const T mz1 = (HAS2 && g1 >= 0 && g1 + 2 < P.gz) ? T(1) : T(0);
		const T md1 = (HAS1 && g1 >= 0 && g1 + 1 < P.gz) ? T(1) : T(0);
#pragma unroll
		for (int j = 0; j < RY; ++j) {
			const T* pa = reinterpret_cast<const T*>(&a[j]);
			const T* pb = reinterpret_cast<const T*>(&b[j]);
			const T* pc = reinterpret_cast<const T*>(&X[0][j]);
			const T* pd = reinterpret_cast<const T*>(&X[1][j]);
#pragma unroll
			for (int e = 0; e < VX; ++e) {
				U1[j][e] = mz2 * (pa[e] - T(2) * pb[e] + pc[e]);
				U2[j][e] = mz1 * (pb[e] - T(2) * pc[e] + pd[e]);
				D1[j][e] = md1 * (pc[e] - pb[e]);
			}
		}
	}

	double dot_acc = 0.0;
	for (int s0 = 0; s0 < nsteps; s0 += U) {
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int s = s0 + u;
			if (s >= nsteps) { break; }
			const int z = z_begin + s;
			const V* xc  = X[u % kOwn];
			const V* xp1 = X[(u + 1) % kOwn];
			const V* xp2 = X[(u + 2) % kOwn];
			Halo& h = H[u % kHal];
			load_own(z + kOwn - 1, X[(u + kOwn - 1) % kOwn]);
			const int gzc = z + P.zoff;
			const T mz  = (HAS2 && gzc >= 0 && gzc + 2 < P.gz) ? T(1) : T(0);
			const T mzd = (HAS1 && gzc + 1 < P.gz) ? T(1) : T(0);
			T* yp = y + static_cast<int64_t>(z) * P.plane;
			V outs[RY];
#pragma unroll
			for (int j = 0; j < RY; ++j) {
				const T* pc  = reinterpret_cast<const T*>(&xc[j]);
				const T* pp1 = reinterpret_cast<const T*>(&xp1[j]);
				const T* pp2 = reinterpret_cast<const T*>(&xp2[j]);
				T acc2[VX], acc1[VX];
#pragma unroll
				for (int e = 0; e < VX; ++e) { acc2[e] = T(0); acc1[e] = T(0); }
				// x: window x[gx-2 .. gx+VX+1]: two values from each neighbouring lane
				{
					T w[VX + 4];
					w[0] = lane_shift<T, true>(h.hx[j][0], pc[VX - 2]);
					w[1] = lane_shift<T, true>(h.hx[j][1], pc[VX - 1]);
#pragma unroll
					for (int e = 0; e < VX; ++e) { w[2 + e] = pc[e]; }
					w[VX + 2] = lane_shift<T, false>(h.hx[j][0], pc[0]);
					w[VX + 3] = lane_shift<T, false>(h.hx[j][1], pc[1]);
					if (HAS2) {
						T ux[VX + 2];
#pragma unroll
						for (int k = 0; k < VX + 2; ++k) { ux[k] = m2x[k] ? (w[k] - T(2) * w[k + 1] + w[k + 2]) : T(0); }
#pragma unroll
						for (int e = 0; e < VX; ++e) { acc2[e] += ux[e] - T(2) * ux[e + 1] + ux[e + 2]; }
					}
					if (HAS1) {
						T dx[VX + 1];
#pragma unroll
						for (int k = 0; k < VX + 1; ++k) { dx[k] = m1x[k] ? (w[k + 2] - w[k + 1]) : T(0); }
#pragma unroll
						for (int e = 0; e < VX; ++e) { acc1[e] += dx[e] - dx[e + 1]; }
					}
				}
				// y: rows j-2 .. j+2 out of the halo rows and the own rows (registers)
				{
					auto row = [&](int r) -> const T* {
						return r < 0 ? reinterpret_cast<const T*>(&h.hy[2 + r])
						             : (r >= RY ? reinterpret_cast<const T*>(&h.hy[2 + (r - RY)]) : reinterpret_cast<const T*>(&xc[r]));
					};
					const T *r1 = row(j - 1), *r3 = row(j + 1);
					if (HAS2) {
						const T *r0 = row(j - 2), *r4 = row(j + 2);
#pragma unroll
						for (int e = 0; e < VX; ++e) {
							const T ua = r0[e] - T(2) * r1[e] + pc[e];
							const T ub = r1[e] - T(2) * pc[e] + r3[e];
							const T uc = pc[e] - T(2) * r3[e] + r4[e];
							acc2[e] += c2y[j][0] * ua - T(2) * (c2y[j][1] * ub) + c2y[j][2] * uc;
						}
					}
					if (HAS1) {
#pragma unroll
						for (int e = 0; e < VX; ++e) { acc1[e] += c1y[j][0] * (pc[e] - r1[e]) - c1y[j][1] * (r3[e] - pc[e]); }
					}
				}
				// z: carried row values
				if (HAS2) {
#pragma unroll
					for (int e = 0; e < VX; ++e) {
						const T u0 = mz * (pc[e] - T(2) * pp1[e] + pp2[e]);
						acc2[e] += U1[j][e] - T(2) * U2[j][e] + u0;
						U1[j][e] = U2[j][e];
						U2[j][e] = u0;
					}
				}
				if (HAS1) {
#pragma unroll
					for (int e = 0; e < VX; ++e) {
						const T d0 = mzd * (pp1[e] - pc[e]);
						acc1[e] += D1[j][e] - d0;
						D1[j][e] = d0;
					}
				}
				T* po = reinterpret_cast<T*>(&outs[j]);
#pragma unroll
				for (int e = 0; e < VX; ++e) {
					T v = C.w0x3 * pc[e];
					if (HAS2) { v += C.w2sq * acc2[e]; }
					if (HAS1) { v += C.w1sq * acc1[e]; }
					po[e] = v;
				}
			}
			if (CELLS && any_cells) {
				// plane s is complete when the cell wave has finished layers s and s + 1; its slot goes back zeroed
#if !(defined(FI_STRIP_DBG) && (FI_STRIP_DBG & 16))  // timing builds: the march does not wait for the cell wave
				wait_ge(0, s + 2);
#endif
				T* slotp = &yb[0] + (s % kRing) * PS + PW + 2 + VX * lane;
#pragma unroll
				for (int j = 0; j < RY; ++j) {
					V* sp = reinterpret_cast<V*>(slotp + j * PW);
					const V  v0 = *sp;
					const T* p0 = reinterpret_cast<const T*>(&v0);
					T* po = reinterpret_cast<T*>(&outs[j]);
#pragma unroll
					for (int e = 0; e < VX; ++e) { po[e] += p0[e]; }
					*sp = V{};
				}
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the zeros are in before the slot is handed back
				__hip_atomic_store(&s_flag[1], s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			}
#pragma unroll
			for (int j = 0; j < RY; ++j) {
				if (lane_ok && row_ok[j]) {
					const T* pc = reinterpret_cast<const T*>(&xc[j]);
					const T* po = reinterpret_cast<const T*>(&outs[j]);
					*reinterpret_cast<V*>(yp + own_off[j]) = outs[j];
					T dsum = T(0);
#pragma unroll
					for (int e = 0; e < VX; ++e) { dsum += pc[e] * po[e]; }
					dot_acc += static_cast<double>(dsum);
				}
			}
			load_halo(z + kHal, h);
		}
	}
	if (partial) {
		const double wsum = strip_wave_sum(dot_acc);
		if (lane == 0) { partial[slot] = wsum; }
	}
}

#ifndef FI_STRIP_RY
#define FI_STRIP_RY 4
#endif
template <typename T>
constexpr int strip_rows() { return FI_STRIP_RY; }

// FI_STRIP_CHECK (tests, debugging): every list bound and record the kernel would follow, checked without following it
__global__ void k_strip_check(MarchParams P, StripLists L, int64_t n_row, int64_t n_blk, int ry, int tx, unsigned int* bad)
{
	const int slot = blockIdx.x;
	if (slot >= P.nwg) { return; }
	const int strips_xy = P.tiles_x * P.tiles_y;
	const int chunk = slot / strips_xy, sxy = slot % strips_xy;
	const int x0 = (sxy % P.tiles_x) * tx, y0 = (sxy / P.tiles_x) * ry;
	const int z_begin = P.own_z0 + chunk * P.zc;
	const int64_t lay0 = static_cast<int64_t>(slot) * (P.zc + 1) * 4;
	for (int l = 0; l <= P.zc; ++l) {
		const uint32_t r0 = L.lay_row[lay0 + l * 4], r1 = L.lay_row[lay0 + l * 4 + 4];
		const uint32_t b0 = L.lay_blk[lay0 + l * 4], b1 = L.lay_blk[lay0 + l * 4 + 4];
		if (r0 > r1 || r1 > n_row) { atomicAdd(&bad[0], 1u); continue; }
		if (b0 > b1 || b1 > n_blk) { atomicAdd(&bad[1], 1u); continue; }
		const int lz = z_begin - 1 + l;
		for (int kind = 0; kind < 2; ++kind) {
			const uint32_t a = kind ? b0 : r0, e = kind ? b1 : r1;
			for (uint32_t r = a + threadIdx.x; r < e; r += blockDim.x) {
				const uint32_t pos = kind ? L.pos_blk[r] : L.pos_row[r];
				const int tcx = static_cast<int>(pos & 0xFFu) - 1, tcy = static_cast<int>(pos >> 16) - 1;
				if (tcx < -1 || tcx >= tx || tcy < -1 || tcy >= ry) { atomicAdd(&bad[2], 1u); continue; }
				const int cx = x0 + tcx, cy = y0 + tcy;
				if (cx < -1 || cx >= P.nx || cy < -1 || cy >= P.ny || lz < -1 || lz >= P.nzl) { atomicAdd(&bad[3], 1u); }
			}
		}
	}
}

}  // namespace

// The strip kernel takes the context's apply: see the head of this file for the conditions.
bool strip_wanted(const fi_ctx* c)
{
	// NOT the default (round 6's measurements, profiles/r6_ablation.md): without data cells the strips run at 0.60-0.66 of
	// 8 TB/s where the marching kernel stops at 0.55-0.57, but every data-carrying context of this library pays for the
	// cell wave's scattered reads of corner values and records through the CU's one vector-memory pipeline -- 0.46-0.55
	// against the marching kernel's 0.56 (value data) / 0.47 (oriented points) at 512^3.  FI_STRIP=1 takes this path (tests do).
	if (!test_switch("FI_STRIP") || test_switch("FI_NO_STRIP") || test_switch("FI_NO_MARCH")) { return false; }
	const Geom& g = c->g;
	if (g.ndim != 3 || c->dtype != FI_F64 || c->nranks != 1 || g.nown != g.nloc) { return false; }
	constexpr int VX = VecOf<double>::VX;
	if (g.gn[0] % VX != 0 || g.gn[0] < kWave * VX) { return false; }
	if (!(c->w.model_1 > 0) && !(c->w.model_2 > 0)) { return false; }
	return c->march.valid;  // (the marching kernel's own conditions: 3-D, model_0/1/2 or the wide rows on top)
}

// strips and chunks: whole rounds of one wave per SIMD (4 per CU), chunks as long as that allows (a chunk re-reads 4 planes)
void strip_setup(fi_ctx* c)
{
	MarchState& m = c->strip;
	m.valid = m.fused = false;  // (the buffers of an earlier assemble are kept for the next)
	m.n_row = m.n_blk = m.cells_row = m.cells_blk = 0;
	if (!strip_wanted(c)) { return; }
	const Geom& g = c->g;
	constexpr int VX = VecOf<double>::VX;
	constexpr int RY = strip_rows<double>();
	MarchParams& P = m.P;
	P = c->march.P;  // nx, ny, nzl, gz, zoff, own_z0, own_z1, plane
	P.tx = kWave * VX;
	P.ty = RY;
	P.txt = kWave;
	P.tiles_x = (P.nx + P.tx - 1) / P.tx;
	P.tiles_y = (P.ny + RY - 1) / RY;
	const int nz_own = P.own_z1 - P.own_z0;
	int cus = 256;
	(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
	const int slots = (cus > 0 ? cus : 256) * 4;
	const int strips = P.tiles_x * P.tiles_y;
	// cost model of pick_chunk: rounds x (planes + fill); a chunk reads 4 planes of its neighbours
	int    best = 1;
	double best_cost = 1e300;
	for (int zc = 1; zc <= 128 && zc <= nz_own; ++zc) {
		const int64_t nw = static_cast<int64_t>(strips) * ((nz_own + zc - 1) / zc);
		const double  r  = static_cast<double>(nw) / slots;
		const double  rounds = r < 6.0 ? static_cast<double>((nw + slots - 1) / slots) : r;
		const double  cost = rounds * (zc + 5);
		if (cost < best_cost * 0.999) {
			best_cost = cost;
			best = zc;
		}
	}
	if (const char* env = test_switch("FI_STRIP_ZC")) {
		if (atoi(env) > 0) { best = atoi(env) < nz_own ? atoi(env) : nz_own; }
	}
	P.zc     = best;
	P.chunks = (nz_own + P.zc - 1) / P.zc;
	P.nwg    = strips * P.chunks;
	P.dense_min = 0;
	P.dbg    = 0;
	m.valid  = true;
	(void)g;
}

template <typename T>
void strip_launch(fi_ctx* c, const T* x, T* y, double* partial)
{
	const MarchState& m = c->strip;
	const MarchParams& P = m.P;
	StripCoef<T> C;
	{
		const fi_weights& w = c->w;
		const T w0 = w.model_0 > 0 ? static_cast<T>(w.model_0) : T(0);
		const T w1 = w.model_1 > 0 ? static_cast<T>(w.model_1) : T(0);
		const T w2 = w.model_2 > 0 ? static_cast<T>(w.model_2) : T(0);
		C.w0x3 = T(3) * w0 * w0;
		C.w1sq = w1 * w1;
		C.w2sq = w2 * w2;
	}
	StripLists L{m.lay_row.as<uint32_t>(), m.lay_blk.as<uint32_t>(), m.pos_row.as<uint32_t>(), m.pos_blk.as<uint32_t>(), m.coef_row.p,
	             m.coef_blk.p};
	const int* done = c->scal.p ? &c->scal.as<CgScalars>()->done : nullptr;
	const int  grid = ((P.nwg + 7) / 8) * 8;
	const bool h1 = c->w.model_1 > 0, h2 = c->w.model_2 > 0;
	constexpr int RY = strip_rows<T>();
	const int  threads = m.fused ? 2 * kWave : kWave;  // (with data cells: the march and the strip's cell wave)
	// with data cells: FOUR workgroups per CU -- four marches, as without cells; the cell waves of strips without a record
	// leave at once, and the freed wave slots would let a fifth to eighth march in (measured slower: 539 against 465 us at
	// 512^3 without data).  37 KB of LDS per workgroup (27 static + this pad) keep the fifth out.
	size_t pad = 0;
	if (m.fused) {
		if (const char* e = test_switch("FI_STRIP_LDS_PAD")) { pad = static_cast<size_t>(atoi(e)); } else { pad = 10 * 1024; }
	}
	auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), pad, c->stream, P, C, L, x, y, partial, done); };
	auto pick = [&](auto cells, auto pack) {
		constexpr bool CL = decltype(cells)::value;
		constexpr bool PK = decltype(pack)::value;
		if (h1 && h2) {
			launch(k_apply_strip3d<T, true, true, CL, RY, PK>);
		} else if (h2) {
			launch(k_apply_strip3d<T, false, true, CL, RY, PK>);
		} else {
			launch(k_apply_strip3d<T, true, false, CL, RY, PK>);
		}
	};
	if (m.fused && test_switch("FI_STRIP_CHECK")) {
		DevBuf bad;
		bad.alloc(4 * sizeof(unsigned int));
		FI_HIP_TRY(hipMemsetAsync(bad.p, 0, 4 * sizeof(unsigned int), c->stream));
		hipLaunchKernelGGL(k_strip_check, dim3(P.nwg), dim3(kWave), 0, c->stream, P, L, m.n_row, m.n_blk, RY, kWave * VecOf<T>::VX, bad.as<unsigned int>());
		unsigned int h[4] = {0, 0, 0, 0};
		FI_HIP_TRY(hipMemcpyAsync(h, bad.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
		FI_HIP_TRY(hipStreamSynchronize(c->stream));
		fprintf(stderr, "FI_STRIP_CHECK: nwg %d zc %d tiles %d x %d chunks %d n_row %lld n_blk %lld: bad row bounds %u, bad block bounds %u, bad origins %u, corners outside %u\n",
		        P.nwg, P.zc, P.tiles_x, P.tiles_y, P.chunks, static_cast<long long>(m.n_row), static_cast<long long>(m.n_blk), h[0], h[1], h[2], h[3]);
		FI_REQUIRE(h[0] + h[1] + h[2] + h[3] == 0, FI_ERR_STATE, "strip lists failed their check");
		if (!strcmp(test_switch("FI_STRIP_CHECK"), "only")) { return; }
	}
	using std::integral_constant;
	if (!m.fused) {
		pick(integral_constant<bool, false>{}, integral_constant<bool, false>{});
	} else if (c->cells.pack) {
		pick(integral_constant<bool, true>{}, integral_constant<bool, true>{});
	} else {
		pick(integral_constant<bool, true>{}, integral_constant<bool, false>{});
	}
	FI_HIP_TRY(hipGetLastError());
}

void strip_apply(fi_ctx* c, const void* x, void* y, double* partial)
{
	FI_REQUIRE(c->strip.valid && c->dtype == FI_F64, FI_ERR_STATE, "the strip kernel does not apply to this context");
	strip_launch<double>(c, static_cast<const double*>(x), static_cast<double*>(y), partial);
}

}  // namespace fi
