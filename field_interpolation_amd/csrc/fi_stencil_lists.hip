// fi_stencil_lists.hip -- the per-workgroup, per-layer record lists the fused marching kernel (fi_stencil.hip) applies its
// data cells from: membership count and write, radix sort by (workgroup, layer, band), list bounds, self-contained records,
// classification of the workgroups with and without cells.  Part of the assembly (sparse_linear.cpp:59-70, 105-113 replaced).
#include "fi_prim.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "fi_internal.h"
#include "fi_sort.h"
#include "fi_stencil_common.h"

namespace fi {

namespace {

constexpr int kThreads = 256;
// distinct cells per kind (statistics): 64 counter pairs, one 128-byte line each -- atomics on one line serialise at
// ~10 ns each whatever the address within it (64 pairs side by side on 4 lines: the counting pass of 1 M cells took 76 us,
// the writing pass 19)
constexpr int kCountStride = 32;
constexpr int kCountWords  = 64 * kCountStride;

// ---- per-workgroup cell lists -------------------------------------------------------------------------
// A cell with global origin (cx, cy, cz) touches the tile columns {cx/TX, and (cx+1)/TX when cx+1 is a
// tile start}, likewise rows, and along z the chunk holding plane cz plus the next chunk when plane cz+1
// starts it (layer 0 of that chunk): up to 8 (workgroup, layer) lists.  Built without ordering hazards:
//   k_cell_members  pass 1: every cell counts its memberships (most have one; a two-row cell two per membership);
//                   exclusive scan of the counts = the cell's first slot;  pass 2: it writes them there: key =
//                   kind*nbuckets + bucket (kind 0: row records, 1: block record), the tile-relative origin, its
//                   own index.  (Round 1 wrote 8 fixed slots per cell and sorted all of them, 7 of 8 empty: the sort
//                   of 7.8 M pairs was the longest item of the assembly.)
//   radix sort      slots by key (stable: the lists come out in cell order, run to run identical);
//   k_list_bounds   binary searches in the sorted keys = list bounds;
//   k_cell_records  one thread per sorted slot copies the row / packed block into the self-contained record.

template <bool WRITE>
__global__ __launch_bounds__(kThreads) void k_cell_members(MarchParams P, Geom g, int64_t ncell, int64_t nbuckets,
                                                            const uint32_t* __restrict__ cell_id,
                                                            const uint32_t* __restrict__ nrow,
                                                            uint32_t* __restrict__ nslot, const uint32_t* __restrict__ first,
                                                            uint32_t* __restrict__ key, uint32_t* __restrict__ pos,
                                                            uint32_t* __restrict__ cell_of, uint32_t* __restrict__ count)
{
	const int64_t c = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	const bool live = c < ncell;
	const uint32_t rows_c = live ? nrow[c] : 0u;
	const int  kind = (rows_c == 1u) ? 0 : 1;             // 0: single-row cell, 1: multi-row cell (statistics)
	if (!WRITE) {
		// distinct cells of each kind: one atomic per wave (a single hot address serialises in L2)
		// distinct cells of each kind (statistics): one atomic per wave, spread over 64 counter pairs -- atomics on
		// a single hot address serialise at ~10 ns each, 15 k waves would cost 0.15 ms
		const unsigned long long rows = __ballot(live && kind == 0), blks = __ballot(live && kind == 1);
		if ((threadIdx.x & 63) == 0) {
			const int w = (blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) & 63;
			if (rows) { atomicAdd(&count[kCountStride * w], static_cast<uint32_t>(__popcll(rows))); }
			if (blks) { atomicAdd(&count[kCountStride * w + 1], static_cast<uint32_t>(__popcll(blks))); }
		}
	}
	if (!live) { return; }
	uint32_t id = cell_id[c];
	const int cx = static_cast<int>(id % static_cast<uint32_t>(g.cn[0])) + g.coff[0];
	id /= static_cast<uint32_t>(g.cn[0]);
	const int cy = static_cast<int>(id % static_cast<uint32_t>(g.cn[1])) + g.coff[1];
	id /= static_cast<uint32_t>(g.cn[1]);
	const int cz = static_cast<int>(id) + g.coff[2];      // global
	const int zz = cz - P.zoff - P.own_z0;                // plane index relative to the first owned plane
	const int nz_own = P.own_z1 - P.own_z0;

	// candidate a = 0: the tile/chunk holding the origin; a = 1: the next one, which sees the cell at -1
	const bool x_ok[2] = {cx >= 0 && cx / P.tx < P.tiles_x, (cx + 1) % P.tx == 0 && (cx + 1) / P.tx < P.tiles_x};
	const int  x_ti[2] = {cx >= 0 ? cx / P.tx : 0, (cx + 1) / P.tx};
	const int  x_tc[2] = {cx >= 0 ? cx % P.tx : 0, -1};
	const bool y_ok[2] = {cy >= 0 && cy / P.ty < P.tiles_y, (cy + 1) % P.ty == 0 && (cy + 1) / P.ty < P.tiles_y};
	const int  y_ti[2] = {cy >= 0 ? cy / P.ty : 0, (cy + 1) / P.ty};
	const int  y_tc[2] = {cy >= 0 ? cy % P.ty : 0, -1};
	const bool z_ok[2] = {zz >= 0 && zz < nz_own, zz + 1 >= 0 && zz + 1 < nz_own && (zz + 1) % P.zc == 0};
	const int  z_tk[2] = {zz >= 0 ? zz / P.zc : 0, (zz + 1) / P.zc};
	const int  z_tl[2] = {zz >= 0 ? zz % P.zc + 1 : 0, 0};
	// A cell with exactly two data rows joins the ROW lists with two records (row index in bits 8..15 of pos, the
	// pair adjacent and in row order: the sort is stable and the pair's slots are neighbours), so that its rows are
	// prefetched like every other row record; the kernel scatters the two rows of a cell in two passes.  That takes
	// two of the cell's 8 slots per membership, so a two-row cell with more than 4 memberships (the corner of a tile
	// AND of a chunk) stays a block record.  Block records are loaded where they are used, on the critical path of
	// the plane step: config 4 (3 % two-row cells) 66 -> 47 us per launch.
	int nmemb = 0;
#pragma unroll
	for (int m = 0; m < 8; ++m) { nmemb += (z_ok[m >> 2] && y_ok[(m >> 1) & 1] && x_ok[m & 1]) ? 1 : 0; }
	const bool pair     = rows_c == 2u && nmemb <= 4;
	const int  listkind = (rows_c == 1u || pair) ? 0 : 1;  // 0: row records, 1: block record
	if (!WRITE) {
		nslot[c] = static_cast<uint32_t>(pair ? 2 * nmemb : nmemb);
		return;
	}
	uint32_t kk[8], pv[8];
#pragma unroll
	for (int m = 0; m < 8; ++m) { kk[m] = static_cast<uint32_t>(2 * nbuckets); pv[m] = 0; }  // unused slot: sorts last
	int j = 0;
#pragma unroll
	for (int m = 0; m < 8; ++m) {
		const int a = m >> 2, b = (m >> 1) & 1, d = m & 1;
		if (z_ok[a] && y_ok[b] && x_ok[d]) {
			const int     wg     = (z_tk[a] * P.tiles_y + y_ti[b]) * P.tiles_x + x_ti[d];
			const int     band   = (y_tc[b] + 1) * 4 / (P.ty + 1);  // 4 bands of consecutive origin rows -1 .. ty-1
			const int64_t bucket = (static_cast<int64_t>(wg) * (P.zc + 1) + z_tl[a]) * 4 + band;
			const uint32_t k  = static_cast<uint32_t>(listkind * nbuckets + bucket);
			const uint32_t pp = static_cast<uint32_t>(x_tc[d] + 1) | (static_cast<uint32_t>(y_tc[b] + 1) << 16);
			if (pair) {
#pragma unroll
				for (int q = 0; q < 8; ++q) {  // slots 2j, 2j+1 (static indexing: the arrays stay in registers)
					if (q == 2 * j) { kk[q] = k; pv[q] = pp; }
					if (q == 2 * j + 1) { kk[q] = k; pv[q] = pp | (1u << 8); }
				}
				++j;
			} else {
				kk[m] = k;
				pv[m] = pp;
			}
		}
	}
	// the used slots, in slot order (the order the fixed-slot form sorted them in)
	uint32_t at = first[c];
#pragma unroll
	for (int m = 0; m < 8; ++m) {
		if (kk[m] != static_cast<uint32_t>(2 * nbuckets)) {
			key[at]     = kk[m];
			pos[at]     = pv[m];
			cell_of[at] = static_cast<uint32_t>(c);
			++at;
		}
	}
}

// ---- the same lists WITHOUT a sort (round 4) -----------------------------------------------------------------------------
// The cells are sorted by extended id -- z, y, x -- and a list is (workgroup, layer, band of origin rows): for each of the
// band's <= 5 origin rows the cells with x origin tile_x * tx - 1 .. tile_x * tx + tx - 1, a contiguous RANGE of the sorted
// cells; the ranges one after the other are the list in the order the stable sort by (key, slot) produced.  With
//   kinds[c]   records cell c contributes to a list it is a member of: (row records | block records << 32)
//   P[c]       exclusive prefix sums of kinds over the sorted cells
//   seg        where the x range of every tile starts and ends in every (y, z) row of cells (k_seg_bounds)
// a list's record counts are differences of P at its ranges' ends (k_list_count, a thread per list), the lists' bounds an
// exclusive scan of the counts, and a cell's record goes to bound + records of the band's earlier rows + P[c] - P[range
// start] (k_list_fill, a thread per cell as in k_cell_members).  Replaces: membership count + scan + round trip +
// membership write + iota + radix sort of 1.15 M pairs (3 passes, 7 fills) + bounds by binary search + record copy --
// 27 calls and two host round trips per level -> 13 and one.
__device__ inline bool pair_cell(const MarchParams& P, int cx, int cy, int zz)
{
	const int nz_own = P.own_z1 - P.own_z0;
	const int mx = ((cx >= 0 && cx / P.tx < P.tiles_x) ? 1 : 0) + (((cx + 1) % P.tx == 0 && (cx + 1) / P.tx < P.tiles_x) ? 1 : 0);
	const int my = ((cy >= 0 && cy / P.ty < P.tiles_y) ? 1 : 0) + (((cy + 1) % P.ty == 0 && (cy + 1) / P.ty < P.tiles_y) ? 1 : 0);
	const int mz = ((zz >= 0 && zz < nz_own) ? 1 : 0) + ((zz + 1 >= 0 && zz + 1 < nz_own && (zz + 1) % P.zc == 0) ? 1 : 0);
	return mx * my * mz <= 4;  // a two-row cell with more memberships stays a block record (k_cell_members)
}

__global__ __launch_bounds__(kThreads) void k_cell_kinds(MarchParams P, Geom g, int64_t ncell, const uint32_t* __restrict__ cell_id,
                                                          const uint32_t* __restrict__ nrow, unsigned long long* __restrict__ kinds,
                                                          uint32_t* __restrict__ uniq)
{
	const int64_t c = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	const bool live = c < ncell;
	const uint32_t rows_c = live ? nrow[c] : 0u;
	// distinct cells of each kind (statistics): one atomic per wave, spread over 64 counter pairs on lines of their own
	const unsigned long long rows = __ballot(live && rows_c == 1u), blks = __ballot(live && rows_c != 1u);
	if ((threadIdx.x & 63) == 0) {
		const int w = (blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6)) & 63;
		if (rows) { atomicAdd(&uniq[kCountStride * w], static_cast<uint32_t>(__popcll(rows))); }
		if (blks) { atomicAdd(&uniq[kCountStride * w + 1], static_cast<uint32_t>(__popcll(blks))); }
	}
	if (c > ncell) { return; }
	if (c == ncell) {  // (the scan runs over ncell + 1 entries: P[ncell] = all records)
		kinds[c] = 0ull;
		return;
	}
	uint32_t id = cell_id[c];
	const int cx = static_cast<int>(id % static_cast<uint32_t>(g.cn[0])) + g.coff[0];
	id /= static_cast<uint32_t>(g.cn[0]);
	const int cy = static_cast<int>(id % static_cast<uint32_t>(g.cn[1])) + g.coff[1];
	id /= static_cast<uint32_t>(g.cn[1]);
	const int zz = static_cast<int>(id) + g.coff[2] - P.zoff - P.own_z0;
	const bool pair = rows_c == 2u && pair_cell(P, cx, cy, zz);
	kinds[c] = (rows_c == 1u || pair) ? (pair ? 2ull : 1ull) : (1ull << 32);
}

// seg[(row * (tiles_x + 1) + t) * 2 + {0, 1}]: first cell of row `row` with x origin >= t * tx - 1 / >= t * tx (extended-local
// rows: row = lz * cn1 + ly).  Tile t's cells in the row: [seg[.. t ..][0], seg[.. t + 1 ..][1]).
__global__ __launch_bounds__(kThreads) void k_seg_bounds(MarchParams P, Geom g, int64_t nrows, int64_t ncell,
                                                          const uint32_t* __restrict__ cell_id, uint32_t* __restrict__ seg)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	const int per = P.tiles_x + 1;
	if (i >= nrows * per) { return; }
	const int64_t row = i / per;
	const int     t   = static_cast<int>(i % per);
	// (binary searches over ALL the sorted cells, ~20 steps: a table of the rows' bounds first would be a launch more)
#pragma unroll
	for (int which = 0; which < 2; ++which) {
		int lx = t * P.tx - 1 + which - g.coff[0];  // first extended-local x of interest
		if (lx < 0) { lx = 0; }
		if (lx > g.cn[0]) { lx = g.cn[0]; }    // (= the first cell of the next row)
		const uint64_t key = static_cast<uint64_t>(row) * static_cast<uint64_t>(g.cn[0]) + static_cast<uint64_t>(lx);
		int64_t lo = 0, hi = ncell;
		while (lo < hi) {
			const int64_t mid = (lo + hi) >> 1;
			if (static_cast<uint64_t>(cell_id[mid]) < key) { lo = mid + 1; } else { hi = mid; }
		}
		seg[i * 2 + which] = static_cast<uint32_t>(lo);
	}
}

// the sorted-cell range of (tile_x, extended-local row ly of plane lz), empty when the row does not exist
__device__ inline void tile_row_range(const MarchParams& P, const Geom& g, const uint32_t* __restrict__ seg, int tile_x, int ly, int lz,
                                      uint32_t* s, uint32_t* e)
{
	*s = 0;
	*e = 0;
	if (ly < 0 || ly >= g.cn[1] || lz < 0 || lz >= g.cn[2]) { return; }
	const int64_t row = static_cast<int64_t>(lz) * g.cn[1] + ly;
	const int64_t i   = row * (P.tiles_x + 1) + tile_x;
	*s = seg[i * 2];
	*e = seg[(i + 1) * 2 + 1];
	if (*e < *s) { *e = *s; }
}

// counts[b] = row records of list b, counts[nbuckets + b] = block records
__global__ __launch_bounds__(kThreads) void k_list_count(MarchParams P, Geom g, int64_t nbuckets, const uint32_t* __restrict__ seg,
                                                          const unsigned long long* __restrict__ pre, uint32_t* __restrict__ counts)
{
	const int64_t b = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (b >= nbuckets) { return; }
	const int per_wg = (P.zc + 1) * 4;
	const int wg = static_cast<int>(b / per_wg), rem = static_cast<int>(b % per_wg);
	const int layer = rem / 4, band = rem % 4;
	const int tiles_xy = P.tiles_x * P.tiles_y;
	const int chunk = wg / tiles_xy, txy = wg % tiles_xy;
	const int tile_y = txy / P.tiles_x, tile_x = txy % P.tiles_x;
	const int nz_own = P.own_z1 - P.own_z0;
	const int zz = chunk * P.zc + layer - 1;  // plane of the origins, relative to the first owned plane (layer 0: the one below the chunk)
	unsigned long long n = 0;
	if (layer >= 1 ? (zz >= 0 && zz < nz_own) : (zz + 1 >= 0 && zz + 1 < nz_own)) {
		const int lz = zz + P.zoff + P.own_z0 - g.coff[2];
		for (int r = 0; r <= P.ty; ++r) {
			if (r * 4 / (P.ty + 1) != band) { continue; }
			uint32_t s, e;
			tile_row_range(P, g, seg, tile_x, tile_y * P.ty + r - 1 - g.coff[1], lz, &s, &e);
			n += pre[e] - pre[s];
		}
	}
	counts[b]            = static_cast<uint32_t>(n & 0xFFFFFFFFull);
	counts[nbuckets + b] = static_cast<uint32_t>(n >> 32);
}

// the lists' bounds in the kernel's form from the scan of the counts: row records from 0, block records from 0
__global__ __launch_bounds__(kThreads) void k_list_layout(int64_t nbuckets, const uint32_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ first, uint32_t* __restrict__ lay_row,
                                                           uint32_t* __restrict__ lay_blk)
{
	const int64_t b = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (b > nbuckets) { return; }
	const uint32_t rows_all = first[nbuckets];  // = all row records: the block records' scan starts there
	if (b < nbuckets) {
		lay_row[b] = first[b];
		lay_blk[b] = first[nbuckets + b] - rows_all;
	} else {
		lay_row[b] = rows_all;
		lay_blk[b] = first[2 * nbuckets - 1] + counts[2 * nbuckets - 1] - rows_all;
	}
}

// one thread per cell: its records into every list it is a member of (the memberships of k_cell_members)
template <typename T>
__global__ __launch_bounds__(kThreads) void k_list_fill(MarchParams P, Geom g, int64_t ncell, const uint32_t* __restrict__ seg,
                                                         const unsigned long long* __restrict__ pre, const uint32_t* __restrict__ cell_id,
                                                         const uint32_t* __restrict__ nrow, const uint32_t* __restrict__ lay_row,
                                                         const uint32_t* __restrict__ lay_blk, const T* __restrict__ row1,
                                                         const T* __restrict__ mrow, const uint32_t* __restrict__ nfac,
                                                         uint32_t* __restrict__ pos_row, uint32_t* __restrict__ pos_blk,
                                                         T* __restrict__ coef_row, T* __restrict__ coef_blk)
{
	using V = typename VecOf<T>::V;
	constexpr int VX = VecOf<T>::VX;
	const int64_t c = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (c >= ncell) { return; }
	uint32_t id = cell_id[c];
	const int lx = static_cast<int>(id % static_cast<uint32_t>(g.cn[0]));
	id /= static_cast<uint32_t>(g.cn[0]);
	const int ly = static_cast<int>(id % static_cast<uint32_t>(g.cn[1]));
	id /= static_cast<uint32_t>(g.cn[1]);
	const int lz = static_cast<int>(id);
	const int cx = lx + g.coff[0], cy = ly + g.coff[1];
	const int zz = lz + g.coff[2] - P.zoff - P.own_z0;
	const int nz_own = P.own_z1 - P.own_z0;
	const unsigned long long mine = pre[c + 1] - pre[c];
	const bool     as_rows = (mine & 0xFFFFFFFFull) != 0;
	const uint32_t nrec    = as_rows ? static_cast<uint32_t>(mine) : 1u;  // 1, or 2 for a pair
	const unsigned long long pc = pre[c];

	const bool x_ok[2] = {cx >= 0 && cx / P.tx < P.tiles_x, (cx + 1) % P.tx == 0 && (cx + 1) / P.tx < P.tiles_x};
	const int  x_ti[2] = {cx >= 0 ? cx / P.tx : 0, (cx + 1) / P.tx};
	const int  x_tc[2] = {cx >= 0 ? cx % P.tx : 0, -1};
	const bool y_ok[2] = {cy >= 0 && cy / P.ty < P.tiles_y, (cy + 1) % P.ty == 0 && (cy + 1) / P.ty < P.tiles_y};
	const int  y_ti[2] = {cy >= 0 ? cy / P.ty : 0, (cy + 1) / P.ty};
	const int  y_tc[2] = {cy >= 0 ? cy % P.ty : 0, -1};
	const bool z_ok[2] = {zz >= 0 && zz < nz_own, zz + 1 >= 0 && zz + 1 < nz_own && (zz + 1) % P.zc == 0};
	const int  z_tk[2] = {zz >= 0 ? zz / P.zc : 0, (zz + 1) / P.zc};
	const int  z_tl[2] = {zz >= 0 ? zz % P.zc + 1 : 0, 0};
	const uint32_t kf = as_rows ? 0u : nfac[c];
	for (int m = 0; m < 8; ++m) {
		const int a = m >> 2, b = (m >> 1) & 1, d = m & 1;
		if (!(z_ok[a] && y_ok[b] && x_ok[d])) { continue; }
		const int     wg     = (z_tk[a] * P.tiles_y + y_ti[b]) * P.tiles_x + x_ti[d];
		const int     r      = y_tc[b] + 1;
		const int     band   = r * 4 / (P.ty + 1);
		const int64_t bucket = (static_cast<int64_t>(wg) * (P.zc + 1) + z_tl[a]) * 4 + band;
		// records of the list in front of this cell's: the band's earlier rows, then the cells before it in its own row's range
		unsigned long long before = 0;
		for (int rr = r - 1; rr >= 0 && rr * 4 / (P.ty + 1) == band; --rr) {
			uint32_t s, e;
			tile_row_range(P, g, seg, x_ti[d], ly - (r - rr), lz, &s, &e);
			before += pre[e] - pre[s];
		}
		{
			uint32_t s, e;
			tile_row_range(P, g, seg, x_ti[d], ly, lz, &s, &e);
			before += pc - pre[s];
		}
		const uint32_t pp = static_cast<uint32_t>(x_tc[d] + 1) | (static_cast<uint32_t>(r) << 16);
		if (as_rows) {
			int64_t at = static_cast<int64_t>(lay_row[bucket]) + static_cast<int64_t>(before & 0xFFFFFFFFull);
			for (uint32_t ridx = 0; ridx < nrec; ++ridx, ++at) {
				pos_row[at] = pp | (ridx << 8);
				const V* src = reinterpret_cast<const V*>(ridx ? mrow + c * 64 + ridx * 8 : row1 + c * 8);
				V*       dst = reinterpret_cast<V*>(coef_row + at * 8);
#pragma unroll
				for (int v = 0; v < 8 / VX; ++v) { dst[v] = src[v]; }
			}
		} else {
			const int64_t at = static_cast<int64_t>(lay_blk[bucket]) + static_cast<int64_t>(before >> 32);
			pos_blk[at] = pp | (kf << 8);
			const V* src = reinterpret_cast<const V*>(mrow + c * 64);
			V*       dst = reinterpret_cast<V*>(coef_blk + at * 64);
			const uint32_t nvec = (kf == 0xFFu ? 36u : kf * 8u) / VX;  // 255: the packed block (fp64 contexts)
			for (uint32_t v = 0; v < nvec; ++v) { dst[v] = src[v]; }
		}
	}
}

// the host's view of the lists in one buffer: [0..1] workgroups with / without cells, [2..3] row / block records,
// [4..5] distinct cells per kind (sum of the 64 counter pairs), then one flag byte per workgroup
__global__ __launch_bounds__(kThreads) void k_pack_readback(int nwg, const int* __restrict__ nsel, const uint32_t* __restrict__ n_row,
                                                             const uint32_t* __restrict__ n_blk, const uint32_t* __restrict__ uniq64,
                                                             const uint8_t* __restrict__ has, uint8_t* __restrict__ out)
{
	uint32_t* head = reinterpret_cast<uint32_t*>(out);
	if (threadIdx.x == 0) {
		uint32_t a = 0, b = 0;
		for (int w = 0; w < 64; ++w) {
			a += uniq64[kCountStride * w];
			b += uniq64[kCountStride * w + 1];
		}
		head[0] = static_cast<uint32_t>(nsel[0]);
		head[1] = static_cast<uint32_t>(nsel[1]);
		head[2] = *n_row;
		head[3] = *n_blk;
		head[4] = a;
		head[5] = b;
	}
	for (int i = threadIdx.x; i < nwg; i += kThreads) { out[24 + i] = has[i]; }
}

// total number of slots = first slot + count of the last cell
__global__ void k_slot_total(int64_t ncell, const uint32_t* __restrict__ first, const uint32_t* __restrict__ nslot,
                             uint32_t* __restrict__ total)
{
	*total = first[ncell - 1] + nslot[ncell - 1];
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_cell_records(int64_t n_row, int64_t n_all,
                                                            const uint32_t* __restrict__ slot_sorted,
                                                            const uint32_t* __restrict__ pos,
                                                            const uint32_t* __restrict__ cell_of,
                                                            const T* __restrict__ row1, const T* __restrict__ mrow,
                                                            const uint32_t* __restrict__ nfac,
                                                            uint32_t* __restrict__ pos_row, uint32_t* __restrict__ pos_blk,
                                                            T* __restrict__ coef_row, T* __restrict__ coef_blk)
{
	using V = typename VecOf<T>::V;
	constexpr int VX = VecOf<T>::VX;
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (i >= n_all) { return; }
	const uint32_t slot = slot_sorted[i];
	const int64_t  c    = cell_of[slot];
	if (i < n_row) {  // keys of kind 0 sort first
		const uint32_t pp = pos[slot];
		pos_row[i] = pp;
		// row (pp >> 8) & 0xFF of the cell: the data rows of a cell with <= 8 rows are its factor rows; row 0 is row1 (the
		// only place a single-row cell keeps it)
		const uint32_t ridx = (pp >> 8) & 0xFFu;
		const V* src = reinterpret_cast<const V*>(ridx ? mrow + c * 64 + ridx * 8 : row1 + c * 8);
		V*       dst = reinterpret_cast<V*>(coef_row + i * 8);
#pragma unroll
		for (int k = 0; k < 8 / VX; ++k) { dst[k] = src[k]; }
	} else {
		const int64_t  j = i - n_row;
		const uint32_t k = nfac[c];
		pos_blk[j] = pos[slot] | (k << 8);
		const V* src = reinterpret_cast<const V*>(mrow + c * 64);
		V*       dst = reinterpret_cast<V*>(coef_blk + j * 64);
		const uint32_t nvec = (k == 0xFFu ? 36u : k * 8u) / VX;  // 255: the packed block (fp64 contexts)
		for (uint32_t r = 0; r < nvec; ++r) { dst[r] = src[r]; }
	}
}

// List bounds straight from the sorted keys (no per-bucket counters: 10^6 scattered atomics cost 0.3 ms):
// bound[k] = first sorted slot whose key is >= k, for k = 0 .. nkeys: one binary search per key value
// (unused slots carry nkeys and sort behind every real key).
__global__ __launch_bounds__(kThreads) void k_list_bounds(int64_t nslots, int64_t nkeys, const uint32_t* __restrict__ key_sorted,
                                                           uint32_t* __restrict__ bound)
{
	const int64_t k = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (k > nkeys) { return; }
	int64_t lo = 0, hi = nslots;  // first index with key_sorted[i] >= k
	while (lo < hi) {
		const int64_t mid = (lo + hi) >> 1;
		if (static_cast<int64_t>(key_sorted[mid]) < k) { lo = mid + 1; } else { hi = mid; }
	}
	bound[k] = static_cast<uint32_t>(lo);
}

// bounds of the block records are counted from the first block record
__global__ __launch_bounds__(kThreads) void k_split_bounds(int64_t nbuckets, const uint32_t* __restrict__ bound,
                                                            uint32_t* __restrict__ lay_row, uint32_t* __restrict__ lay_blk)
{
	const int64_t b = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (b > nbuckets) { return; }
	lay_row[b] = bound[b];
	lay_blk[b] = bound[nbuckets + b] - bound[nbuckets];
}

__global__ void k_iota32(uint32_t* v, int64_t n)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i < n) { v[i] = static_cast<uint32_t>(i); }
}

// Workgroups with and without cells, both lists in ascending order, and everything the host wants to read back, in ONE
// single-block launch (round 3: a classification kernel, two stream compactions of three launches each and a packing
// kernel -- eight launches for a few thousand flags).  out: [0..1] the two counts, [2..3] row / block records, [4..5]
// distinct cells per kind (sum of the 64 counter pairs), then one flag byte per workgroup.
__global__ __launch_bounds__(1024) void k_classify_pack(int nwg, int per_wg, const uint32_t* __restrict__ lay_row,
                                                         const uint32_t* __restrict__ lay_blk, const uint32_t* __restrict__ n_row,
                                                         const uint32_t* __restrict__ n_blk, const uint32_t* __restrict__ uniq64,
                                                         uint32_t* __restrict__ wg_cells, uint32_t* __restrict__ wg_plain,
                                                         uint8_t* __restrict__ out)
{
	using Scan = rocprim::block_scan<int, 1024>;
	__shared__ typename Scan::storage_type tmp;
	__shared__ int base_with, base_without;
	if (threadIdx.x == 0) {
		base_with    = 0;
		base_without = 0;
	}
	__syncthreads();
	for (int w0 = 0; w0 < nwg; w0 += 1024) {
		const int wg = w0 + static_cast<int>(threadIdx.x);
		bool any = false;
		if (wg < nwg) {
			const int64_t a = static_cast<int64_t>(wg) * per_wg, b = a + per_wg;
			any = lay_row[b] > lay_row[a] || lay_blk[b] > lay_blk[a];
			out[24 + wg] = any ? 1 : 0;
		}
		const int flag = (wg < nwg && any) ? 1 : 0;
		int pos = 0, total = 0;
		Scan().exclusive_scan(flag, pos, 0, total, tmp, rocprim::plus<int>());
		const int live = nwg - w0 < 1024 ? nwg - w0 : 1024;  // workgroups of this chunk
		if (wg < nwg) {
			if (any) {
				wg_cells[base_with + pos] = static_cast<uint32_t>(wg);
			} else {
				wg_plain[base_without + (static_cast<int>(threadIdx.x) - pos)] = static_cast<uint32_t>(wg);
			}
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			base_with += total;
			base_without += live - total;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		uint32_t* head = reinterpret_cast<uint32_t*>(out);
		uint32_t a = 0, b = 0;
		for (int w = 0; w < 64; ++w) {
			a += uniq64[kCountStride * w];
			b += uniq64[kCountStride * w + 1];
		}
		head[0] = static_cast<uint32_t>(base_with);
		head[1] = static_cast<uint32_t>(base_without);
		head[2] = *n_row;
		head[3] = *n_blk;
		head[4] = a;
		head[5] = b;
	}
}

// 1 for a workgroup whose lists (all its layers, both record kinds) hold at least one cell
__global__ __launch_bounds__(kThreads) void k_classify_wg(int nwg, int per_wg, const uint32_t* __restrict__ lay_row,
                                                           const uint32_t* __restrict__ lay_blk, uint32_t* __restrict__ ids,
                                                           uint8_t* __restrict__ has, uint8_t* __restrict__ has_not)
{
	const int wg = blockIdx.x * kThreads + threadIdx.x;
	if (wg >= nwg) { return; }
	const int64_t a = static_cast<int64_t>(wg) * per_wg, b = a + per_wg;
	const bool any = lay_row[b] > lay_row[a] || lay_blk[b] > lay_blk[a];
	ids[wg]     = static_cast<uint32_t>(wg);
	has[wg]     = any ? 1 : 0;
	has_not[wg] = any ? 0 : 1;
}

}  // namespace

template <typename T>
void build_cell_lists(fi_ctx* c, MarchState& m)
{
	const MarchParams& P = m.P;
	const int64_t ncell = c->cells.ncell;
	const int64_t nbuckets = static_cast<int64_t>(P.nwg) * (P.zc + 1) * 4;  // (workgroup, layer, band of origin rows)
	FI_REQUIRE(ncell * 8 < (1LL << 31) && 2 * nbuckets < (1LL << 31), FI_ERR_UNSUPPORTED, "too many data cells for one context");
	hipStream_t st = c->stream;
	const int nwg = P.nwg;
	m.lay_row.alloc(sizeof(uint32_t) * (nbuckets + 1));
	m.lay_blk.alloc(sizeof(uint32_t) * (nbuckets + 1));
	m.wg_cells.alloc(sizeof(uint32_t) * nwg);
	m.wg_plain.alloc(sizeof(uint32_t) * nwg);
	// everything the host needs in ONE copy: the two selection counts, the record totals, the distinct cells per kind, and
	// the per-workgroup flags
	DevBuf& pack = c->scratch[34];
	pack.alloc(24 + static_cast<size_t>(nwg));
	const size_t h_pack_bytes = 24 + static_cast<size_t>(nwg);
	uint8_t* const h_pack = static_cast<uint8_t*>(pinned(c, 0, h_pack_bytes));
	if (!test_switch("FI_LISTS_BY_SORT")) {
		// ---- lists as ranges of the sorted cells (no sort) ----
		DevBuf &uniq = c->scratch[14], &counts_d = c->scratch[15], &first_d = c->scratch[16], &kinds = c->scratch[17],
		       &pre = c->scratch[18], &seg = c->scratch[19], &tmp = c->scratch[20];
		const int64_t nrows = static_cast<int64_t>(c->g.cn[1]) * c->g.cn[2];
		const int64_t nseg  = nrows * (P.tiles_x + 1);
		uniq.alloc(sizeof(uint32_t) * kCountWords);
		counts_d.alloc(sizeof(uint32_t) * (2 * nbuckets + 1));
		first_d.alloc(sizeof(uint32_t) * (2 * nbuckets + 1));
		kinds.alloc(sizeof(unsigned long long) * (ncell + 1));
		pre.alloc(sizeof(unsigned long long) * (ncell + 1));
		seg.alloc(sizeof(uint32_t) * 2 * (nseg + 1));
		FI_HIP_TRY(hipMemsetAsync(uniq.p, 0, sizeof(uint32_t) * kCountWords, st));
		hipLaunchKernelGGL(k_cell_kinds, dim3(static_cast<int>((ncell + 1 + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, P, c->g, ncell,
		                   c->cells.cell_id.as<uint32_t>(), c->cells.nrow.as<uint32_t>(), kinds.as<unsigned long long>(), uniq.as<uint32_t>());
		size_t tb0 = 0;
		FI_HIP_TRY(prim::exclusive_sum(nullptr, tb0, kinds.as<unsigned long long>(), pre.as<unsigned long long>(), static_cast<size_t>(ncell + 1), st));
		tmp.alloc(tb0);
		FI_HIP_TRY(prim::exclusive_sum(tmp.p, tb0, kinds.as<unsigned long long>(), pre.as<unsigned long long>(), static_cast<size_t>(ncell + 1), st));
		hipLaunchKernelGGL(k_seg_bounds, dim3(static_cast<int>((nseg + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, P, c->g, nrows,
		                   ncell, c->cells.cell_id.as<uint32_t>(), seg.as<uint32_t>());
		const int nbb = static_cast<int>((nbuckets + 1 + kThreads - 1) / kThreads);
		hipLaunchKernelGGL(k_list_count, dim3(nbb), dim3(kThreads), 0, st, P, c->g, nbuckets, seg.as<uint32_t>(),
		                   pre.as<unsigned long long>(), counts_d.as<uint32_t>());
		size_t tb1 = 0;
		FI_HIP_TRY(prim::exclusive_sum(nullptr, tb1, counts_d.as<uint32_t>(), first_d.as<uint32_t>(), static_cast<size_t>(2 * nbuckets), st));
		tmp.alloc(tb1 > tb0 ? tb1 : tb0);
		FI_HIP_TRY(prim::exclusive_sum(tmp.p, tb1, counts_d.as<uint32_t>(), first_d.as<uint32_t>(), static_cast<size_t>(2 * nbuckets), st));
		hipLaunchKernelGGL(k_list_layout, dim3(nbb), dim3(kThreads), 0, st, nbuckets, counts_d.as<uint32_t>(), first_d.as<uint32_t>(),
		                   m.lay_row.as<uint32_t>(), m.lay_blk.as<uint32_t>());
		hipLaunchKernelGGL(k_classify_pack, dim3(1), dim3(1024), 0, st, nwg, (P.zc + 1) * 4, m.lay_row.as<uint32_t>(), m.lay_blk.as<uint32_t>(),
		                   m.lay_row.as<uint32_t>() + nbuckets, m.lay_blk.as<uint32_t>() + nbuckets, uniq.as<uint32_t>(),
		                   m.wg_cells.as<uint32_t>(), m.wg_plain.as<uint32_t>(), pack.as<uint8_t>());
		FI_HIP_TRY(hipMemcpyAsync(h_pack, pack.p, h_pack_bytes, hipMemcpyDeviceToHost, st));
		FI_HIP_TRY(hipStreamSynchronize(st));
	} else {
		// ---- the same lists by a radix sort of (list key, slot) pairs: the form of rounds 1-3, kept for the comparison ----
		DevBuf &count = c->scratch[14], &key = c->scratch[15], &pos = c->scratch[16], &slot_in = c->scratch[17],
		       &key_sorted = c->scratch[18], &slot_sorted = c->scratch[19], &tmp = c->scratch[20], &nslot = c->scratch[31],
		       &first = c->scratch[32], &cell_of = c->scratch[33];
		count.alloc(sizeof(uint32_t) * (2 * nbuckets + 2 + kCountWords));  // distinct cells per kind (64 pairs), then the list bounds
		nslot.alloc(sizeof(uint32_t) * (ncell + 1));
		first.alloc(sizeof(uint32_t) * (ncell + 1));
		FI_HIP_TRY(hipMemsetAsync(count.p, 0, sizeof(uint32_t) * kCountWords, st));
		const int nb = static_cast<int>((ncell + kThreads - 1) / kThreads);
		hipLaunchKernelGGL(k_cell_members<false>, dim3(nb), dim3(kThreads), 0, st, P, c->g, ncell, nbuckets,
		                   c->cells.cell_id.as<uint32_t>(), c->cells.nrow.as<uint32_t>(), nslot.as<uint32_t>(),
		                   static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr),
		                   static_cast<uint32_t*>(nullptr), count.as<uint32_t>());
		size_t tb0 = 0;
		FI_HIP_TRY(prim::exclusive_sum(nullptr, tb0, nslot.as<uint32_t>(), first.as<uint32_t>(), static_cast<size_t>(ncell), st));
		tmp.alloc(tb0);
		FI_HIP_TRY(prim::exclusive_sum(tmp.p, tb0, nslot.as<uint32_t>(), first.as<uint32_t>(), static_cast<size_t>(ncell), st));
		hipLaunchKernelGGL(k_slot_total, dim3(1), dim3(1), 0, st, ncell, first.as<uint32_t>(), nslot.as<uint32_t>(),
		                   first.as<uint32_t>() + ncell);
		uint32_t h_slots = 0;
		FI_HIP_TRY(hipMemcpyAsync(&h_slots, first.as<uint32_t>() + ncell, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
		FI_HIP_TRY(hipStreamSynchronize(st));  // the sort below is sized by it
		const int64_t nslots = h_slots;
		key.alloc(sizeof(uint32_t) * (nslots + 1));
		pos.alloc(sizeof(uint32_t) * (nslots + 1));
		cell_of.alloc(sizeof(uint32_t) * (nslots + 1));
		slot_in.alloc(sizeof(uint32_t) * (nslots + 1));
		key_sorted.alloc(sizeof(uint32_t) * (nslots + 1));
		slot_sorted.alloc(sizeof(uint32_t) * (nslots + 1));
		hipLaunchKernelGGL(k_cell_members<true>, dim3(nb), dim3(kThreads), 0, st, P, c->g, ncell, nbuckets,
		                   c->cells.cell_id.as<uint32_t>(), c->cells.nrow.as<uint32_t>(), static_cast<uint32_t*>(nullptr),
		                   first.as<uint32_t>(), key.as<uint32_t>(), pos.as<uint32_t>(), cell_of.as<uint32_t>(), count.as<uint32_t>());
		if (nslots > 0) {
			hipLaunchKernelGGL(k_iota32, dim3(static_cast<int>((nslots + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
			                   slot_in.as<uint32_t>(), nslots);
			int key_bits = 1;  // only the bits the keys have take part
			while ((1LL << key_bits) < 2 * nbuckets) { ++key_bits; }
			size_t tb = 0;
			FI_HIP_TRY(sort_pairs_u32(nullptr, tb, key.as<uint32_t>(), key_sorted.as<uint32_t>(), slot_in.as<uint32_t>(),
			                          slot_sorted.as<uint32_t>(), static_cast<int>(nslots), 0, key_bits, st));
			tmp.alloc(tb);
			FI_HIP_TRY(sort_pairs_u32(tmp.p, tb, key.as<uint32_t>(), key_sorted.as<uint32_t>(), slot_in.as<uint32_t>(),
			                          slot_sorted.as<uint32_t>(), static_cast<int>(nslots), 0, key_bits, st));
		}
		uint32_t* bound = count.as<uint32_t>() + kCountWords;  // [2 * nbuckets + 1]
		hipLaunchKernelGGL(k_list_bounds, dim3(static_cast<int>((2 * nbuckets + 1 + kThreads - 1) / kThreads)), dim3(kThreads), 0,
		                   st, nslots, 2 * nbuckets, key_sorted.as<uint32_t>(), bound);
		hipLaunchKernelGGL(k_split_bounds, dim3(static_cast<int>((nbuckets + 1 + kThreads - 1) / kThreads)), dim3(kThreads), 0, st,
		                   nbuckets, bound, m.lay_row.as<uint32_t>(), m.lay_blk.as<uint32_t>());
		hipLaunchKernelGGL(k_classify_pack, dim3(1), dim3(1024), 0, st, nwg, (P.zc + 1) * 4, m.lay_row.as<uint32_t>(), m.lay_blk.as<uint32_t>(),
		                   m.lay_row.as<uint32_t>() + nbuckets, m.lay_blk.as<uint32_t>() + nbuckets, count.as<uint32_t>(),
		                   m.wg_cells.as<uint32_t>(), m.wg_plain.as<uint32_t>(), pack.as<uint8_t>());
		FI_HIP_TRY(hipMemcpyAsync(h_pack, pack.p, h_pack_bytes, hipMemcpyDeviceToHost, st));
		FI_HIP_TRY(hipStreamSynchronize(st));
	}
	uint32_t head[6];
	memcpy(head, h_pack, sizeof(head));
	const int counts[2] = {static_cast<int>(head[0]), static_cast<int>(head[1])};
	const uint32_t totals[2] = {head[2], head[3]}, uniq_cells[2] = {head[4], head[5]};
	const uint8_t* h_has = h_pack + 24;
	m.n_row = totals[0];
	m.n_blk = totals[1];
	m.cells_row = uniq_cells[0];
	m.cells_blk = uniq_cells[1];
	m.pos_row.alloc(sizeof(uint32_t) * (m.n_row + 1));
	m.pos_blk.alloc(sizeof(uint32_t) * (m.n_blk + 1));
	m.coef_row.alloc(sizeof(T) * 8 * (m.n_row + 1));
	m.coef_blk.alloc(sizeof(T) * 64 * (m.n_blk + 1));
	const int64_t n_all = m.n_row + m.n_blk;
	if (n_all > 0) {
		if (!test_switch("FI_LISTS_BY_SORT")) {
			hipLaunchKernelGGL((k_list_fill<T>), dim3(static_cast<int>((ncell + kThreads - 1) / kThreads)), dim3(kThreads), 0, st, P, c->g,
			                   ncell, c->scratch[19].as<uint32_t>(), c->scratch[18].as<unsigned long long>(), c->cells.cell_id.as<uint32_t>(),
			                   c->cells.nrow.as<uint32_t>(), m.lay_row.as<uint32_t>(), m.lay_blk.as<uint32_t>(), c->cells.row1.as<T>(),
			                   c->cells.mrow.as<T>(), c->cells.nfac.as<uint32_t>(), m.pos_row.as<uint32_t>(), m.pos_blk.as<uint32_t>(),
			                   m.coef_row.as<T>(), m.coef_blk.as<T>());
		} else {
			hipLaunchKernelGGL((k_cell_records<T>), dim3(static_cast<int>((n_all + kThreads - 1) / kThreads)), dim3(kThreads), 0,
			                   st, m.n_row, n_all, c->scratch[19].as<uint32_t>(), c->scratch[16].as<uint32_t>(), c->scratch[33].as<uint32_t>(),
			                   c->cells.row1.as<T>(), c->cells.mrow.as<T>(), c->cells.nfac.as<uint32_t>(), m.pos_row.as<uint32_t>(),
			                   m.pos_blk.as<uint32_t>(), m.coef_row.as<T>(), m.coef_blk.as<T>());
		}
	}
	FI_HIP_TRY(hipGetLastError());
	{
		m.n_wg_cells = counts[0];
		m.n_wg_plain = counts[1];
		// Plain workgroups as runs of consecutive empty chunks of one tile: a split launch cuts the lattice into short
		// chunks for the sake of the data workgroups, and short chunks cost the plain ones 4 overlap planes and the
		// pipeline fill per 8 planes (512^3 without data: 344 us at 8 planes, 212 us at 128).  Run starts in chunk-major
		// order (neighbours in the list are neighbours in the plane); a run covers at most 128 planes.
		{
			const int tiles_xy = P.tiles_x * P.tiles_y;
			int cus = 256;
			(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
			const double slots = static_cast<double>((cus > 0 ? cus : 256) * FI_BASE_WAVES);
			std::vector<uint32_t> first, len;
			std::vector<uint8_t>  used(static_cast<size_t>(nwg));
			auto build_runs = [&](int cap) {
				first.clear();
				len.clear();
				std::fill(used.begin(), used.end(), 0);
				for (int ch = 0; ch < P.chunks; ++ch) {
					for (int t = 0; t < tiles_xy; ++t) {
						const int w0 = ch * tiles_xy + t;
						if (h_has[w0] || used[w0]) { continue; }
						int n = 0;
						while (n < cap && ch + n < P.chunks && !h_has[w0 + n * tiles_xy]) {
							used[w0 + n * tiles_xy] = 1;
							++n;
						}
						first.push_back(static_cast<uint32_t>(w0));
						len.push_back(static_cast<uint32_t>(n));
					}
				}
			};
			// longest run: the one that minimises rounds x (planes + pipeline fill), whole rounds while there are few
			// (the cost model of pick_chunk); the grid must still cover the CUs
			int    best_cap = 1;
			double best_cost = 1e300;
			for (int cap = 1; cap * P.zc <= 128; cap *= 2) {
				build_runs(cap);
				const double r = static_cast<double>(first.size()) / slots;
				const double rounds = r < 6.0 ? std::ceil(r) : r;
				const double cost = rounds * (cap * P.zc + 5);
				if (cost < best_cost * 0.999) {
					best_cost = cost;
					best_cap  = cap;
				}
			}
			if (const char* env = tuning_switch("FI_RUN_CAP")) {  // experiments
				if (atoi(env) > 0) { best_cap = atoi(env); }
			}
			build_runs(best_cap);
			m.n_wg_plain = static_cast<int>(first.size());
			m.wg_runs.alloc(sizeof(uint32_t) * (first.size() + 1));
			if (!first.empty()) {
				// through the context's pinned upload buffer: it is next written by the next call of this function, behind that
				// call's own round trip -- no wait here
				const size_t nb = sizeof(uint32_t) * first.size();
				uint8_t* up = static_cast<uint8_t*>(pinned(c, 1, 2 * nb));
				memcpy(up, first.data(), nb);
				memcpy(up + nb, len.data(), nb);
				FI_HIP_TRY(hipMemcpyAsync(m.wg_plain.p, up, nb, hipMemcpyHostToDevice, st));
				FI_HIP_TRY(hipMemcpyAsync(m.wg_runs.p, up + nb, nb, hipMemcpyHostToDevice, st));
			}
		}
	}
}


template void build_cell_lists<float>(fi_ctx*, MarchState&);
template void build_cell_lists<double>(fi_ctx*, MarchState&);

}  // namespace fi
