// fi_assembly.hip -- data-constraint assembly on the GPU.
//
// Reference path replaced: add_points -> add_value_constraint / add_gradient_constraint
// (field_interpolation.cpp:343-371, 57-107, 123-187) followed by as_sparse_matrix_float, make_square
// and A^T*b (sparse_linear.cpp:59-70, 105-113, 120).  The reference appends one 12-byte triplet per
// coefficient and later squares the whole matrix with a sparse*sparse product.  Here:
//
//   1. k_emit_rows      one thread per data point: the point's value row and D gradient rows are written
//                       as "cell rows" -- (extended cell id, 2^D corner coefficients, rhs) -- with the
//                       same fp32 arithmetic, in the same order, as the reference (multilerp :15-55).
//                       All default kernels touch only the 2^D corners of the cell floor(pos).
//   2. radix sort       rows by cell id (stable, so rows of a cell keep their input order; hipCUB).
//   3. k_build_blocks   one thread per occupied cell: fp64 accumulation of the symmetric 2^D x 2^D block
//                       sum a a^T (packed upper triangle, one cell after another) and of sum a*rhs.
//   4. k_scatter_cells  A^T b and the data part of diag(A^T A) onto the lattice, 2^D parity colours so
//                       that no two cells of one launch share a corner: deterministic, no atomics.
//
// HBM layout: rows: key[], coef[row][2^D], rhs[]; blocks: blk[cell][2^D(2^D+1)/2] (144 B per 3-D cell in
// fp32: the apply kernels read a block as 9 consecutive 16-byte loads).

#include "fi_prim.h"

#include "fi_internal.h"
#include "fi_sort.h"

namespace fi {

namespace {

constexpr int kThreads = 256;

__host__ __device__ inline int packed_index(int i, int j, int nc)  // i <= j
{
	return i * nc - (i * (i - 1)) / 2 + (j - i);
}

// sum of row q of the symmetric block: the cell's share of (A_data 1)(corner q) -- what the lumped replica of a
// mixed-precision context puts on its diagonal (fi_levels.hip twin_assemble_lumped)
template <int NC>
__device__ inline double block_row_sum(const double* B, int q)
{
	double s = 0.0;
#pragma unroll
	for (int j = 0; j < NC; ++j) { s += B[q <= j ? packed_index(q, j, NC) : packed_index(j, q, NC)]; }
	return s;
}

struct EmitArgs {
	Geom  g;
	float vw, gw;
	int   vk, gk;
	int   has_nrm, has_pw, has_val;
	int   rows_per_point;        // 1 + D, or 1 when the batch has no gradient rows (no normals or a zero gradient weight)
	uint32_t invalid_key;
	float pos_scale, nrm_scale;  // 1 on the caller's lattice; 1/2^l and 2^l on coarser levels
};

// Extended local cell id of the cell with GLOBAL origin c[] (origins run from -1), or invalid when the
// cell does not touch this rank's slab.
template <int D>
__device__ inline uint32_t cell_key(const Geom& g, const int* c, uint32_t invalid)
{
	uint32_t key = 0;
	uint32_t mul = 1;
	for (int d = 0; d < D; ++d) {
		const int l = c[d] - g.coff[d];
		if (l < 0 || l >= g.cn[d]) { return invalid; }
		key += static_cast<uint32_t>(l) * mul;
		mul *= static_cast<uint32_t>(g.cn[d]);
	}
	return key;
}

template <int D>
__global__ __launch_bounds__(kThreads) void k_emit_rows(EmitArgs a, long n, const float* __restrict__ pos,
                                                         const float* __restrict__ nrm,
                                                         const float* __restrict__ pw,
                                                         const float* __restrict__ val,
                                                         uint32_t* __restrict__ key, float* __restrict__ coef,
                                                         float* __restrict__ rhs)
{
	constexpr int NC = 1 << D;
	const long i = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
	if (i >= n) { return; }
	const Geom& g = a.g;

	float p[D];
	bool  finite = true;
	for (int d = 0; d < D; ++d) {
		p[d]   = pos[i * D + d] * a.pos_scale;
		if (g.pshift[d] != 0.0f) { p[d] += g.pshift[d]; }  // a level halved cell-centred along d
		finite = finite && isfinite(p[d]);
	}
	const float w     = a.has_pw ? pw[i] : 1.0f;
	const float value = a.has_val ? val[i] : 0.0f;
	const long  slot0 = i * a.rows_per_point;

	// cell of the point: floor(pos) per axis (multilerp :29-32, cell_index :115)
	int   cell[D];
	float t[D];
	bool  cell_in_ext = finite;  // origin within [-1, size-1] on every axis
	bool  cell_valid  = finite;  // 0 <= origin and origin + 1 < size (cell_index :116)
	for (int d = 0; d < D; ++d) {
		const float fl = floorf(p[d]);
		if (!(fl >= -1.0f && fl <= static_cast<float>(g.gn[d] - 1))) {
			cell_in_ext = false;
			cell_valid  = false;
			cell[d]     = 0;
			t[d]        = 0.0f;
			continue;
		}
		cell[d] = static_cast<int>(fl);
		t[d]    = p[d] - static_cast<float>(cell[d]);
		if (!(0 <= cell[d] && cell[d] + 1 < g.gn[d])) { cell_valid = false; }
	}

	// ---- value row ----------------------------------------------------------------------------
	{
		const float cw = w * a.vw;
		uint32_t k = a.invalid_key;
		float    c[NC];
		float    b = 0.0f;
		for (int q = 0; q < NC; ++q) { c[q] = 0.0f; }
		if (a.vk == FI_VALUE_LINEAR_INTERPOLATION) {
			// field_interpolation.cpp:57-80: corners outside the lattice are dropped, the kept weights
			// are not renormalised; rhs = (sum of kept coefficient) * value.
			if (cw != 0.0f && cell_in_ext) {
				int   kept = 0;
				float sum  = 0.0f;
				for (int q = 0; q < NC; ++q) {
					float lw = 1.0f;
					bool  in = true;
					for (int d = 0; d < D; ++d) {
						const int up = (q >> d) & 1;
						const int cc = cell[d] + up;
						lw *= up ? t[d] : 1.0f - t[d];
						in = in && (0 <= cc) && (cc < g.gn[d]);
					}
					if (in) {
						const float s = lw * cw;
						c[q] = s;
						sum += s;
						++kept;
					}
				}
				if (kept > 0) {
					k = cell_key<D>(g, cell, a.invalid_key);
					b = sum * value;
				}
			}
		} else {
			// field_interpolation.cpp:82-107 through add_equation (sparse_linear.cpp:34-50): nearest
			// lattice point by std::round; row [1]*cw, rhs (value - (pos-nearest).gradient)*cw.
			if (cw != 0.0f && finite) {
				bool  ok    = true;
				float along = 0.0f;
				int   corner = 0;
				int   cc[D];
				for (int d = 0; d < D; ++d) {
					const float r = roundf(p[d]);
					if (!(r >= 0.0f && r <= static_cast<float>(g.gn[d] - 1))) {
						ok = false;
						cc[d] = 0;
						continue;
					}
					const int q = static_cast<int>(r);
					along += (p[d] - static_cast<float>(q)) * (nrm[i * D + d] * a.nrm_scale);
					// the nearest point is a corner of the (extended) cell floor(pos)
					int base = static_cast<int>(floorf(p[d]));
					if (base < -1) { base = -1; }
					if (base > q) { base = q; }
					if (q - base > 1) { base = q - 1; }
					cc[d] = base;
					corner |= (q - base) << d;
				}
				if (ok) {
					k = cell_key<D>(g, cc, a.invalid_key);
					c[corner] = 1.0f * cw;
					b = (value - along) * cw;
				}
			}
		}
		key[slot0] = k;
		rhs[slot0] = b;
		for (int q = 0; q < NC; ++q) { coef[slot0 * NC + q] = c[q]; }
	}

	// ---- gradient rows ------------------------------------------------------------------------
	if (a.rows_per_point == 1) { return; }
	for (int d = 0; d < D; ++d) {
		const long slot = slot0 + 1 + d;
		uint32_t k = a.invalid_key;
		float    c[NC];
		float    b = 0.0f;
		for (int q = 0; q < NC; ++q) { c[q] = 0.0f; }
		if (a.has_nrm) {
			const float cw = w * a.gw;
			const float gd = nrm[i * D + d] * a.nrm_scale;
			if (cw != 0.0f && cell_valid) {
				if (a.gk == FI_GRADIENT_NEAREST_NEIGHBOR) {
					// field_interpolation.cpp:134-149: [-1, +1]*cw on the cell edge along d.
					c[0]      = -1.0f * cw;
					c[1 << d] = +1.0f * cw;
					b         = gd * cw;
					k         = cell_key<D>(g, cell, a.invalid_key);
				} else if (a.gk == FI_GRADIENT_CELL_EDGES) {
					// field_interpolation.cpp:150-187: +-cw*2/2^D on all corners, rhs cw*g_d.
					const float term = cw * 2.0f / static_cast<float>(NC);
					for (int q = 0; q < NC; ++q) { c[q] = (((q >> d) & 1) ? +1.0f : -1.0f) * term; }
					b = cw * gd;
					k = cell_key<D>(g, cell, a.invalid_key);
				}
			}
		}
		key[slot] = k;
		rhs[slot] = b;
		for (int q = 0; q < NC; ++q) { coef[slot * NC + q] = c[q]; }
	}
}

__global__ void k_iota(uint32_t* v, long n)
{
	const long i = static_cast<long>(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i < n) { v[i] = static_cast<uint32_t>(i); }
}

constexpr int kRec = 3;  // the cell's record for the sums over the lattice points: [A^T b | diagonal | row sums] x 2^D corners
constexpr uint32_t kHeavyRows = 192;  // cells with more rows are summed by a whole workgroup (k_build_heavy)
// 3-D (eight lanes per cell, all cells side by side): a workgroup per cell only pays for the few cells of the coarsest levels.
// Config 5's 32^3 level (35 k cells of ~570 rows each) through k_build_heavy: 2.8 ms; through the lanes: see r4_ablation 12.
constexpr uint32_t kHeavyRows3 = 4096;

// everything after the sums of a cell: block, first row, rhs, factor rows
// DIRECT: block, first row and the (rhs, diagonal) record go straight to memory (k_build_heavy: one thread per
// workgroup); k_build_blocks sends them through LDS instead and calls this for the factor rows only.
template <int D, typename T, bool DIRECT = true>
__device__ inline void finish_cell(long c, long ncell, uint32_t s, uint32_t m, double* B, const double* gvec,
                                   const uint32_t* __restrict__ sorted_row, const float* __restrict__ coef,
                                   T* __restrict__ blk, T* __restrict__ cell_dr, uint32_t* __restrict__ nrow,
                                   T* __restrict__ row1, T* __restrict__ mrow, uint32_t* __restrict__ nfac,
                                   uint32_t pack_min)
{
	constexpr int NC = 1 << D;
	constexpr int NB = NC * (NC + 1) / 2;
	nrow[c] = m;
	if (DIRECT) {
		for (int e = 0; e < NB; ++e) { blk[c * NB + e] = static_cast<T>(B[e]); }
		const long row = sorted_row[s];  // first row of the cell (the only one when m == 1)
		for (int q = 0; q < NC; ++q) { row1[c * NC + q] = static_cast<T>(coef[row * NC + q]); }
	}
	// what the sums over the lattice points read (k_gather_cells*, k_scatter_cells): the cell's share of A^T b and of the
	// diagonal, rounded to T here, side by side -- one 64-byte line per 3-D cell in fp32 instead of an fp64 vector and
	// eight entries strewn over the 144-byte block (256^3, 1 M cells: the gather 372 -> ... us)
	if (DIRECT) {
		for (int q = 0; q < NC; ++q) {
			cell_dr[static_cast<long>(c) * kRec * NC + q]      = static_cast<T>(gvec[q]);
			cell_dr[static_cast<long>(c) * kRec * NC + NC + q] = static_cast<T>(B[packed_index(q, q, NC)]);
			cell_dr[static_cast<long>(c) * kRec * NC + 2 * NC + q] = static_cast<T>(block_row_sum<NC>(B, q));
		}
	}

	// Factor rows for the fused 3-D kernel: the cell's block as a sum of <= 2^D outer products a a^T.
	//   m <= 2^D rows : the data rows themselves (nothing to compute);
	//   m >  2^D rows : fp32 contexts: the rows of U from B = U^T U (Cholesky of the fp64 block; B is positive
	//                   semi-definite, a vanishing pivot means a vanishing row/column of the remaining Schur
	//                   complement and simply drops out); fp64 contexts: the packed block (marker k = 255).
	// Contexts whose cells mostly hold several rows (pack_min = 3: an SDF, the coarse levels of a cascade) keep every
	// cell of >= 3 rows as its packed block: the kernel fetches a factor row per loop trip, one after another on the
	// plane step's critical path, while the 36 coefficients of a block come in five independent batches and cost
	// 64 instead of 16 m FMAs (512^3 SDF data: fp32 403 -> 374 us, fp64 747 -> 650 us per apply).
	if (mrow) {
		uint32_t k = 0;
		if (m <= static_cast<uint32_t>(NC) && m < pack_min) {
			// (a single-row cell's factor row is row1: k_cell_records takes row 0 from there, nothing to write)
			for (uint32_t r = 0; r < m && m > 1; ++r) {
				const long row = sorted_row[s + r];
				for (int q = 0; q < NC; ++q) { mrow[(c * NC + r) * NC + q] = static_cast<T>(coef[row * NC + q]); }
			}
			k = m;
		} else if (sizeof(T) == 8 || pack_min <= static_cast<uint32_t>(NC)) {
			// fp64 contexts keep the packed block itself: a Cholesky factor of a nearly singular block is only
			// good to ~1e-9 (half the digits), which fp32 never sees but fp64 parity (1e-12) does
			for (int e = 0; e < NB; ++e) { mrow[c * NC * NC + e] = static_cast<T>(B[e]); }
			k = 0xFFu;
		} else {
			double dmax = 0;
			for (int i = 0; i < NC; ++i) { dmax = fmax(dmax, B[packed_index(i, i, NC)]); }
			const double tiny = 1e-14 * dmax;
#pragma unroll
			for (int i = 0; i < NC; ++i) {
				// row i of U, in place: U[i][j] = (B[i][j] - sum_{t<i} U[t][i] U[t][j]) / U[i][i]
				double piv = B[packed_index(i, i, NC)];
#pragma unroll
				for (int t = 0; t < i; ++t) { piv -= B[packed_index(t, i, NC)] * B[packed_index(t, i, NC)]; }
				const bool   live = piv > tiny;
				const double inv  = live ? 1.0 / sqrt(piv) : 0.0;
				B[packed_index(i, i, NC)] = live ? sqrt(piv) : 0.0;
#pragma unroll
				for (int j = i + 1; j < NC; ++j) {
					double v = B[packed_index(i, j, NC)];
#pragma unroll
					for (int t = 0; t < i; ++t) { v -= B[packed_index(t, i, NC)] * B[packed_index(t, j, NC)]; }
					B[packed_index(i, j, NC)] = v * inv;
				}
			}
#pragma unroll
			for (int i = 0; i < NC; ++i) {
				if (B[packed_index(i, i, NC)] != 0.0) {
					for (int q = 0; q < NC; ++q) {
						mrow[(c * NC + k) * NC + q] = q < i ? T(0) : static_cast<T>(B[packed_index(i, q, NC)]);
					}
					++k;
				}
			}
		}
		nfac[c] = k;
	}
}

// One thread per cell.  What every cell gets -- block, first row, (rhs, diagonal) record: 60 values -- leaves through LDS
// so that a store instruction covers consecutive addresses (a thread storing its own cell's 144-byte block puts 64
// lanes on 64 different lines: 256^3 with 1 M cells 143 -> ... us); the factor rows of the multi-row cells are stored
// directly (finish_cell).
template <int D, typename T>
__global__ __launch_bounds__(kThreads) void k_build_blocks(long ncell, const uint32_t* __restrict__ start,
                                                            const uint32_t* __restrict__ count,
                                                            const uint32_t* __restrict__ sorted_row,
                                                            const float* __restrict__ coef,
                                                            const float* __restrict__ rhs, T* __restrict__ blk,
                                                            T* __restrict__ cell_dr, uint32_t* __restrict__ nrow,
                                                            T* __restrict__ row1, T* __restrict__ mrow,
                                                            uint32_t* __restrict__ nfac, uint32_t* __restrict__ heavy,
                                                            uint32_t pack_min, const uint32_t* __restrict__ uniq,
                                                            uint32_t* __restrict__ cell_id)
{
	constexpr int NC = 1 << D;
	constexpr int NB = NC * (NC + 1) / 2;
	// the block in ROUNDS parts (fp64 3-D: two halves of 18, 64 KB of static LDS is the limit), then record + first row
	constexpr int ROUNDS = (D == 3 && sizeof(T) == 8) ? 2 : 1;
	constexpr int EPR    = NB / ROUNDS;
	constexpr int ST1 = EPR | 1, ST2 = kRec * NC + 1;  // odd strides: a column of the tile spreads over the banks
	__shared__ T tile[kThreads * (ST1 > ST2 ? ST1 : ST2)];
	const long c0 = static_cast<long>(blockIdx.x) * kThreads;
	const long c  = c0 + threadIdx.x;
	const int  nb = static_cast<int>(ncell - c0 < kThreads ? ncell - c0 : kThreads);
	uint32_t s = 0, m = 0;
	bool     mine = false;
	if (c < ncell) {
		cell_id[c] = uniq[c];
		s = start[c];
		m = count[c];
		if (m > kHeavyRows) {  // coarse levels put 10^4..10^6 rows into one cell: not a job for one thread
			heavy[1 + atomicAdd(&heavy[0], 1u)] = static_cast<uint32_t>(c);
		} else {
			mine = true;
		}
	}
	double B[NB];
	double gvec[NC];
	for (int e = 0; e < NB; ++e) { B[e] = 0.0; }
	for (int q = 0; q < NC; ++q) { gvec[q] = 0.0; }
	T first[NC];
	for (int q = 0; q < NC; ++q) { first[q] = T(0); }
	if (mine) {
		for (uint32_t r = 0; r < m; ++r) {
			const long row = sorted_row[s + r];
			double     a[NC];
			for (int q = 0; q < NC; ++q) { a[q] = static_cast<double>(coef[row * NC + q]); }
			if (r == 0) {
				for (int q = 0; q < NC; ++q) { first[q] = static_cast<T>(coef[row * NC + q]); }
			}
			const double b = static_cast<double>(rhs[row]);
			int e = 0;
			for (int i = 0; i < NC; ++i) {
				for (int j = i; j < NC; ++j) { B[e++] += a[i] * a[j]; }
				gvec[i] += a[i] * b;
			}
		}
	}
	// (the heavy cells' slots receive zeros here and their values from k_build_heavy, the next launch on the stream)
#pragma unroll
	for (int round = 0; round < ROUNDS; ++round) {
		if (round) { __syncthreads(); }
#pragma unroll
		for (int e = 0; e < EPR; ++e) { tile[threadIdx.x * ST1 + e] = static_cast<T>(B[round * EPR + e]); }
		__syncthreads();
		for (int i = threadIdx.x; i < nb * EPR; i += kThreads) {
			const int cell = i / EPR, e = i - cell * EPR;
			blk[(c0 + cell) * NB + round * EPR + e] = tile[cell * ST1 + e];
		}
	}
	__syncthreads();
#pragma unroll
	for (int q = 0; q < NC; ++q) {
		tile[threadIdx.x * ST2 + q]          = static_cast<T>(gvec[q]);
		tile[threadIdx.x * ST2 + NC + q]     = static_cast<T>(B[packed_index(q, q, NC)]);
		tile[threadIdx.x * ST2 + 2 * NC + q] = static_cast<T>(block_row_sum<NC>(B, q));
	}
	__syncthreads();
	for (int i = threadIdx.x; i < nb * kRec * NC; i += kThreads) {
		const int cell = i / (kRec * NC), e = i - cell * kRec * NC;
		cell_dr[c0 * kRec * NC + i] = tile[cell * ST2 + e];
	}
	__syncthreads();
#pragma unroll
	for (int q = 0; q < NC; ++q) { tile[threadIdx.x * (NC + 1) + q] = first[q]; }
	__syncthreads();
	for (int i = threadIdx.x; i < nb * NC; i += kThreads) {
		const int cell = i / NC, e = i - cell * NC;
		row1[c0 * NC + i] = tile[cell * (NC + 1) + e];
	}
	if (mine) {
		finish_cell<D, T, false>(c, ncell, s, m, B, gvec, sorted_row, coef, blk, cell_dr, nrow, row1, mrow, nfac, pack_min);
	}
}

// 3-D: EIGHT lanes per cell, lane j owns column j of the symmetric 8 x 8 block (8 + 1 fp64 accumulators instead of 44 in
// one thread: 142-154 VGPRs and 38-51 KB of LDS held k_build_blocks at 3 waves per SIMD, each walking a chain of dependent
// loads -- 165 us for the 970 k cells of config 4's fp64 level, 60 us for 30 k cells of ~30 rows on its 32^3 level).  The
// same products summed in the same order, the Cholesky factor of a many-row cell by the same recurrence with the columns
// spread over the lanes (values of row i of U travel by 8-wide shuffles): the same bits as k_build_blocks + finish_cell.
template <typename T>
__global__ __launch_bounds__(kThreads) void k_build_blocks3(long ncell, const uint32_t* __restrict__ start,
                                                             const uint32_t* __restrict__ count,
                                                             const uint32_t* __restrict__ sorted_row,
                                                             const float* __restrict__ coef, const float* __restrict__ rhs,
                                                             T* __restrict__ blk, T* __restrict__ cell_dr,
                                                             uint32_t* __restrict__ nrow, T* __restrict__ row1,
                                                             T* __restrict__ mrow, uint32_t* __restrict__ nfac,
                                                             uint32_t* __restrict__ heavy, uint32_t pack_min, int write_blk,
                                                             const uint32_t* __restrict__ uniq, uint32_t* __restrict__ cell_id)
{
	constexpr int NC = 8, NB = 36;
	const long t = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
	const long c = t >> 3;
	const int  j = static_cast<int>(t & 7);
	if (c >= ncell) { return; }  // (whole groups of 8 lanes: the shuffles below stay inside a live group)
	const uint32_t s = start[c], m = count[c];
	bool mine = true;
	if (m > kHeavyRows3) {  // coarse levels put 10^4..10^6 rows into one cell: a workgroup's job (k_build_heavy fills the slots)
		if (j == 0) { heavy[1 + atomicAdd(&heavy[0], 1u)] = static_cast<uint32_t>(c); }
		mine = false;
	}
	double col[NC];  // col[i] = B[i][j]
	double g = 0.0;
#pragma unroll
	for (int i = 0; i < NC; ++i) { col[i] = 0.0; }
	T first = T(0);
	if (mine) {
		for (uint32_t r = 0; r < m; ++r) {
			const long   row = sorted_row[s + r];
			const float4 lo = *reinterpret_cast<const float4*>(coef + row * NC), hi = *reinterpret_cast<const float4*>(coef + row * NC + 4);
			const float  fj = coef[row * NC + j];
			const double aj = static_cast<double>(fj);
			const double a[NC] = {static_cast<double>(lo.x), static_cast<double>(lo.y), static_cast<double>(lo.z), static_cast<double>(lo.w),
			                      static_cast<double>(hi.x), static_cast<double>(hi.y), static_cast<double>(hi.z), static_cast<double>(hi.w)};
			if (r == 0) { first = static_cast<T>(fj); }
#pragma unroll
			for (int i = 0; i < NC; ++i) { col[i] += a[i] * aj; }
			g += aj * static_cast<double>(rhs[row]);
		}
	}
	// block (upper triangle, packed), the record of the sums over the lattice points, the first row
	double dj = 0.0, sum = 0.0;
#pragma unroll
	for (int i = 0; i < NC; ++i) {
		if (i <= j && write_blk) { blk[c * NB + packed_index(i, j, NC)] = static_cast<T>(col[i]); }
		if (i == j) { dj = col[i]; }
		sum += col[i];  // row sum of the symmetric block, entries in the order 0 .. 7 (block_row_sum)
	}
	cell_dr[c * kRec * NC + j]          = static_cast<T>(g);
	cell_dr[c * kRec * NC + NC + j]     = static_cast<T>(dj);
	cell_dr[c * kRec * NC + 2 * NC + j] = static_cast<T>(sum);
	row1[c * NC + j] = first;
	if (j == 0) {
		nrow[c]    = m;
		cell_id[c] = uniq[c];
	}
	if (!mine) { return; }
	// factor rows for the fused kernel (finish_cell)
	uint32_t k = 0;
	if (m <= static_cast<uint32_t>(NC) && m < pack_min) {
		for (uint32_t r = 0; r < m && m > 1; ++r) {
			const long row = sorted_row[s + r];
			mrow[(c * NC + r) * NC + j] = static_cast<T>(coef[row * NC + j]);
		}
		k = m;
	} else if (sizeof(T) == 8 || pack_min <= static_cast<uint32_t>(NC)) {
#pragma unroll
		for (int i = 0; i < NC; ++i) {
			if (i <= j) { mrow[c * NC * NC + packed_index(i, j, NC)] = static_cast<T>(col[i]); }
		}
		k = 0xFFu;
	} else {
		double dmax = dj;
#pragma unroll
		for (int o = 1; o < NC; o <<= 1) { dmax = fmax(dmax, __shfl_xor(dmax, o, NC)); }
		const double tiny = 1e-14 * dmax;
		double u[NC];  // u[i] = U[i][j]
#pragma unroll
		for (int i = 0; i < NC; ++i) {
			// row i of U: U[i][j] = (B[i][j] - sum_{t<i} U[t][i] U[t][j]) / U[i][i]; every lane forms the pivot from lane i's values
			double piv = __shfl(col[i], i, NC);
			double v   = col[i];
#pragma unroll
			for (int tt = 0; tt < i; ++tt) {
				const double uti = __shfl(u[tt], i, NC);
				piv -= uti * uti;
				v -= uti * u[tt];
			}
			const bool   live = piv > tiny;
			const double inv  = live ? 1.0 / sqrt(piv) : 0.0;
			u[i] = j == i ? (live ? sqrt(piv) : 0.0) : (j > i ? v * inv : 0.0);
			if (live) {  // (uniform over the group)
				mrow[(c * NC + k) * NC + j] = static_cast<T>(u[i]);
				++k;
			}
		}
	}
	if (j == 0) { nfac[c] = k; }
}

// One workgroup per heavy cell: threads stride over the cell's rows, then a fixed-shape tree (wave shuffles,
// 4 wave sums added in order) -- the same bits on every run, whatever order the list was filled in.
template <int D, typename T>
__global__ __launch_bounds__(kThreads) void k_build_heavy(long ncell, const uint32_t* __restrict__ start,
                                                           const uint32_t* __restrict__ count,
                                                           const uint32_t* __restrict__ sorted_row,
                                                           const float* __restrict__ coef,
                                                           const float* __restrict__ rhs, T* __restrict__ blk,
                                                           T* __restrict__ cell_dr, uint32_t* __restrict__ nrow,
                                                           T* __restrict__ row1, T* __restrict__ mrow,
                                                           uint32_t* __restrict__ nfac, const uint32_t* __restrict__ heavy,
                                                           uint32_t pack_min)
{
	constexpr int NC = 1 << D;
	constexpr int NB = NC * (NC + 1) / 2;
	__shared__ double part[kThreads / 64][NB + NC];
	const uint32_t nheavy = heavy[0];
	for (uint32_t h = blockIdx.x; h < nheavy; h += gridDim.x) {
		const long     c = heavy[1 + h];
		const uint32_t s = start[c], m = count[c];
		double B[NB];
		double gvec[NC];
		for (int e = 0; e < NB; ++e) { B[e] = 0.0; }
		for (int q = 0; q < NC; ++q) { gvec[q] = 0.0; }
		for (uint32_t r = threadIdx.x; r < m; r += kThreads) {
			const long row = sorted_row[s + r];
			double     a[NC];
			for (int q = 0; q < NC; ++q) { a[q] = static_cast<double>(coef[row * NC + q]); }
			const double b = static_cast<double>(rhs[row]);
			int e = 0;
			for (int i = 0; i < NC; ++i) {
				for (int j = i; j < NC; ++j) { B[e++] += a[i] * a[j]; }
				gvec[i] += a[i] * b;
			}
		}
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
		for (int e = 0; e < NB + NC; ++e) {  // (unrolled: a run-time index would move the sums to scratch memory)
			double v = e < NB ? B[e] : gvec[e - NB];
			for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o, 64); }
			if (lane == 0) { part[wave][e] = v; }
		}
		__syncthreads();
		if (threadIdx.x == 0) {
#pragma unroll
			for (int e = 0; e < NB + NC; ++e) {
				double v = 0;
				for (int w = 0; w < kThreads / 64; ++w) { v += part[w][e]; }
				if (e < NB) { B[e] = v; } else { gvec[e - NB] = v; }
			}
			finish_cell<D, T>(c, ncell, s, m, B, gvec, sorted_row, coef, blk, cell_dr, nrow, row1, mrow, nfac, pack_min);
		}
		__syncthreads();
	}
}

template <int D, typename T>
__global__ __launch_bounds__(kThreads) void k_scatter_cells(Geom g, long ncell, const uint32_t* __restrict__ cell_id,
                                                             const T* __restrict__ cell_dr,
                                                             T* __restrict__ atb, T* __restrict__ diag,
                                                             float* __restrict__ lump, int colour)
{
	constexpr int NC = 1 << D;
	const long c = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
	if (c >= ncell) { return; }
	uint32_t id = cell_id[c];
	int      l[3] = {0, 0, 0};
	int      col  = 0;
	for (int d = 0; d < D; ++d) {
		l[d] = static_cast<int>(id % static_cast<uint32_t>(g.cn[d]));
		id /= static_cast<uint32_t>(g.cn[d]);
		col |= (l[d] & 1) << d;
	}
	if (col != colour) { return; }
	for (int q = 0; q < NC; ++q) {
		int64_t idx = 0;
		bool    ok  = true;
		for (int d = 0; d < D; ++d) {
			const int gq = l[d] + g.coff[d] + ((q >> d) & 1);  // global coordinate of the corner
			const int li = gq - g.off[d];
			ok = ok && (0 <= gq) && (gq < g.gn[d]) && (g.own_lo[d] <= li) && (li < g.own_hi[d]);
			idx += static_cast<int64_t>(li) * g.stride[d];
		}
		if (ok) {
			atb[idx] += cell_dr[static_cast<long>(c) * kRec * NC + q];
			diag[idx] += cell_dr[static_cast<long>(c) * kRec * NC + NC + q];
			if (lump) { lump[idx] += static_cast<float>(cell_dr[static_cast<long>(c) * kRec * NC + 2 * NC + q]); }
		}
	}
}

// The same sums by GATHER, for lattices where a sizeable share of the cells holds data: a dense map
// extended cell id -> cell index, then one thread per owned lattice point looks up its 2^D incident cells.  The
// cells are visited in the order of their parity colour, so every sum is formed in exactly the order of the 2^D
// scatter launches above (bit-identical results); one launch over N points instead of 2^D launches of scattered
// read-modify-writes (config 4, 256^3: 490 -> ~80 us).
// The same for 3-D lattices, four consecutive x points per thread: the 8 incident cells of the four points lie in 5
// columns x 2 rows x 2 planes of the map (20 look-ups instead of 32, issued as 4 runs of 5), the sums go out as one
// 16-byte store per array.  Same cells in the same (colour) order as the generic kernel: bit-identical sums.
template <typename T>
__global__ __launch_bounds__(kThreads) void k_gather_cells3(Geom g, long ncell, const uint32_t* __restrict__ map,
                                                             const T* __restrict__ cell_dr, T* __restrict__ atb,
                                                             T* __restrict__ diag, float* __restrict__ lump)
{
	constexpr int NC = 8;
	const int ext0 = g.own_hi[0] - g.own_lo[0], ext1 = g.own_hi[1] - g.own_lo[1], ext2 = g.own_hi[2] - g.own_lo[2];
	const int groups = (ext0 + 3) / 4;  // per row; the thread index runs over (group, row, plane)
	int64_t t = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (t >= static_cast<int64_t>(groups) * ext1 * ext2) { return; }
	const int x0 = 4 * static_cast<int>(t % groups);
	t /= groups;
	const int li[3] = {g.own_lo[0] + x0, g.own_lo[1] + static_cast<int>(t % ext1), g.own_lo[2] + static_cast<int>(t / ext1)};
	const int lp[3] = {li[0] + g.off[0] - g.coff[0], li[1] + g.off[1] - g.coff[1], li[2] + g.off[2] - g.coff[2]};
	const int nvalid = ext0 - x0 < 4 ? ext0 - x0 : 4;
	// cid[bz][by][k]: the cell with origin (lp0 - 1 + k, lp1 - by, lp2 - bz), k = 0 .. 4
	uint32_t cid[2][2][5];
#pragma unroll
	for (int bz = 0; bz < 2; ++bz) {
#pragma unroll
		for (int by = 0; by < 2; ++by) {
			const int  ly = lp[1] - by, lz = lp[2] - bz;
			const bool row_ok = ly >= 0 && ly < g.cn[1] && lz >= 0 && lz < g.cn[2];
			const uint32_t base = (static_cast<uint32_t>(lz < 0 ? 0 : lz) * static_cast<uint32_t>(g.cn[1]) +
			                       static_cast<uint32_t>(ly < 0 ? 0 : ly)) * static_cast<uint32_t>(g.cn[0]);
#pragma unroll
			for (int k = 0; k < 5; ++k) {
				const int lx = lp[0] - 1 + k;
				const bool ok = row_ok && lx >= 0 && lx < g.cn[0] && k <= nvalid;
				cid[bz][by][k] = ok ? map[base + static_cast<uint32_t>(lx)] : 0xFFFFFFFFu;
			}
		}
	}
	T a[4], dg[4];
	float lp4[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		a[j]  = T(0);
		dg[j] = T(0);
		lp4[j] = 0.0f;
#pragma unroll
		for (int colour = 0; colour < NC; ++colour) {  // the generic kernel's order: by the parity of the cell's origin
			const int bx = ((lp[0] + j) ^ colour) & 1, by = (lp[1] ^ (colour >> 1)) & 1, bz = (lp[2] ^ (colour >> 2)) & 1;
			// selects over static indices (a runtime index would move the table to scratch memory)
			const uint32_t c00 = bx ? cid[0][0][j] : cid[0][0][j + 1], c01 = bx ? cid[0][1][j] : cid[0][1][j + 1];
			const uint32_t c10 = bx ? cid[1][0][j] : cid[1][0][j + 1], c11 = bx ? cid[1][1][j] : cid[1][1][j + 1];
			const uint32_t c = bz ? (by ? c11 : c10) : (by ? c01 : c00);
			if (c == 0xFFFFFFFFu) { continue; }
			const int q = bx | (by << 1) | (bz << 2);
			a[j] += cell_dr[static_cast<long>(c) * kRec * NC + q];
			dg[j] += cell_dr[static_cast<long>(c) * kRec * NC + NC + q];
			if (lump) { lp4[j] += static_cast<float>(cell_dr[static_cast<long>(c) * kRec * NC + 2 * NC + q]); }
		}
	}
	const int64_t idx = li[0] * g.stride[0] + li[1] * g.stride[1] + li[2] * g.stride[2];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		if (j < nvalid) {
			atb[idx + j]  = a[j];
			diag[idx + j] = dg[j];
			if (lump) { lump[idx + j] = lp4[j]; }
		}
	}
}

// ---- the same sums for 3-D lattices, by TILES of lattice points in LDS ------------------------------------------------
// One workgroup owns 64 x 4 x 4 owned points (round 5; 64 x 4 x 2 before: a tile is a chain of a dozen dependent phases --
// range loads, record loads, eight colour rounds with a barrier each -- and the 256^3 fp64 level runs 16 rounds of such
// workgroups: half as many, twice as large, 173 -> 155 us alone; 64 x 4 x 8: 186 us).  It collects the cells that touch them (the
// cells are sorted by extended id, x fastest: k_xtile_bounds gives the range of the tile's x in every (y, z) row of cells, a tile
// looks at 5 x 5 rows), orders them by the
// parity colour of their origin, adds their records to the tile's accumulators colour after colour -- within a colour no
// two cells share a corner -- and stores the tile with full lines.  The same sums in the same order as the 2^D colour
// launches and the gather launch: the same bits.  No cell map, no zeroing of the arrays, and the time follows the occupied
// cells, not the look-ups (256^3 fp64, 6 % of the cells occupied: gather 310 us + map 45 us + zeroing 42 us -> ... us).
constexpr int kTileX = 64, kTileY = 4, kTileZ = 4;

// xt[(row * (ntx + 1) + t) * 2 + which]: first cell of row `row` whose x origin is >= first + 64 t - 1 + which, `first` the
// extended-local x of the lattice's first owned point.  The cells of the 64 points of tile t in that row -- origins one
// below the tile's first point up to its last point -- are [xt[.. t ..][0], xt[.. t + 1 ..][1]).
__global__ __launch_bounds__(kThreads) void k_xtile_bounds(int64_t nrows, int ntx, int first, int row_len, long ncell,
                                                            const uint32_t* __restrict__ cell_id, uint32_t* __restrict__ xt)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (i >= nrows * (ntx + 1) * 2) { return; }
	const int     which = static_cast<int>(i & 1);
	const int64_t e     = i >> 1;
	const int64_t row   = e / (ntx + 1);
	const int     t     = static_cast<int>(e % (ntx + 1));
	int lx = first + kTileX * t - 1 + which;
	if (lx < 0) { lx = 0; }
	if (lx > row_len) { lx = row_len; }
	const uint64_t key = static_cast<uint64_t>(row) * row_len + static_cast<uint64_t>(lx);
	long lo = 0, hi = ncell;
	while (lo < hi) {
		const long mid = (lo + hi) >> 1;
		if (static_cast<uint64_t>(cell_id[mid]) < key) { lo = mid + 1; } else { hi = mid; }
	}
	xt[i] = static_cast<uint32_t>(lo);
}
constexpr int kTileCells = (kTileX + 1) * (kTileY + 1) * (kTileZ + 1);  // 975: every cell of the tile's 5 x 3 rows

template <typename T>
__global__ __launch_bounds__(kThreads) void k_tile_sums3(Geom g, const uint32_t* __restrict__ xt, int ntx,
                                                          const uint32_t* __restrict__ cell_id, const T* __restrict__ cell_dr,
                                                          T* __restrict__ atb, T* __restrict__ diag, float* __restrict__ lump)
{
	constexpr int NC = 8;
	constexpr int NP = kTileX * kTileY * kTileZ;
	__shared__ T        acc_b[NP], acc_d[NP];
	__shared__ float    acc_l[NP];
	__shared__ uint32_t list_cell[kTileCells], list_loc[kTileCells];
	__shared__ int      ncol[NC], base[NC + 1], fill[NC];
	const int tid = threadIdx.x;
	const int ext0 = g.own_hi[0] - g.own_lo[0], ext1 = g.own_hi[1] - g.own_lo[1], ext2 = g.own_hi[2] - g.own_lo[2];
	const int x0 = static_cast<int>(blockIdx.x) * kTileX, y0 = static_cast<int>(blockIdx.y) * kTileY,
	          z0 = static_cast<int>(blockIdx.z) * kTileZ;  // first owned point of the tile (owned-relative)
	// the same point on the extended cell grid: a point at lp is corner b of the cell with origin lp - b
	const int lp0[3] = {g.own_lo[0] + x0 + g.off[0] - g.coff[0], g.own_lo[1] + y0 + g.off[1] - g.coff[1],
	                    g.own_lo[2] + z0 + g.off[2] - g.coff[2]};
	for (int i = tid; i < NP; i += kThreads) {
		acc_b[i] = T(0);
		acc_d[i] = T(0);
		acc_l[i] = 0.0f;
	}
	if (tid < NC) {
		ncol[tid] = 0;
		fill[tid] = 0;
	}
	__syncthreads();
	// the tile's cells: rows (ly, lz) = lp0 - 1 .. lp0 + tile - 1, within a row lx = lp0x - 1 .. lp0x + 63.  The rows' ranges
	// first (one round of loads), then two passes over ALL their cells at once, a cell per thread and trip: count by colour,
	// then place (the order within a colour is free: its cells touch distinct points).  Every wave walking its own rows one
	// after the other was a chain of eight dependent rounds of loads: 19 us per tile.
	constexpr int ROWS = (kTileY + 1) * (kTileZ + 1);
	__shared__ uint32_t row_first[ROWS], row_key[ROWS], row_at[ROWS + 1];
	if (tid < ROWS) {
		const int ry = tid % (kTileY + 1), rz = tid / (kTileY + 1);
		const int ly = lp0[1] - 1 + ry, lz = lp0[2] - 1 + rz;
		uint32_t s = 0, e = 0, key0 = 0;
		if (ly >= 0 && ly < g.cn[1] && lz >= 0 && lz < g.cn[2]) {
			const uint32_t row = static_cast<uint32_t>(lz) * static_cast<uint32_t>(g.cn[1]) + static_cast<uint32_t>(ly);
			key0 = row * static_cast<uint32_t>(g.cn[0]);
			const int64_t at = (static_cast<int64_t>(row) * (ntx + 1) + static_cast<int64_t>(blockIdx.x)) * 2;
			s = xt[at];
			e = xt[at + 3];  // entry (row, tile + 1), second bound
			if (e < s) { e = s; }
		}
		row_first[tid] = s;
		row_key[tid]   = key0;
		// inclusive scan of the rows' lengths over the first lanes of wave 0 (ROWS <= 64)
		uint32_t a = e - s;
#pragma unroll
		for (int o = 1; o < ROWS; o <<= 1) {
			const uint32_t up = __shfl_up(a, o, 64);
			if (tid >= o) { a += up; }
		}
		row_at[tid + 1] = a;
		if (tid == 0) { row_at[0] = 0; }
	}
	__syncthreads();
	const uint32_t in_rows = row_at[ROWS];
	if (in_rows <= kThreads) {
		// One cell per thread (the usual case on a sparsely occupied lattice): its record straight into registers, then the
		// eight colours one after the other -- no list, no counting pass.
		const bool live = tid < static_cast<int>(in_rows);
		T     rb[NC], rd[NC];
		float rl[NC];
		int   cx = 0, ry = 0, rz = 0, colour = -1;
		if (live) {
			int r = 0;
#pragma unroll
			for (int jj = 1; jj < ROWS; ++jj) { r += row_at[jj] <= static_cast<uint32_t>(tid) ? 1 : 0; }
			const long c = row_first[r] + (static_cast<uint32_t>(tid) - row_at[r]);
			ry = r % (kTileY + 1);
			rz = r / (kTileY + 1);
			const int ly = lp0[1] - 1 + ry, lz = lp0[2] - 1 + rz;
			const int lx = static_cast<int>(cell_id[c] - row_key[r]);
			cx = lx - (lp0[0] - 1);
			colour = (lx & 1) | ((ly & 1) << 1) | ((lz & 1) << 2);
#pragma unroll
			for (int q = 0; q < NC; ++q) {
				rb[q] = cell_dr[c * kRec * NC + q];
				rd[q] = cell_dr[c * kRec * NC + NC + q];
				rl[q] = lump ? static_cast<float>(cell_dr[c * kRec * NC + 2 * NC + q]) : 0.0f;
			}
		}
		for (int col = 0; col < NC; ++col) {
			if (colour == col) {
#pragma unroll
				for (int q = 0; q < NC; ++q) {
					const int px = cx - 1 + (q & 1), py = ry - 1 + ((q >> 1) & 1), pz = rz - 1 + ((q >> 2) & 1);
					if (px >= 0 && px < kTileX && py >= 0 && py < kTileY && pz >= 0 && pz < kTileZ) {
						const int p = (pz * kTileY + py) * kTileX + px;
						// (no two cells of a colour share a point: the hardware add is the plain sum, without the round
						// trip of a read-add-write per corner -- 24 dependent LDS round trips per colour)
						unsafeAtomicAdd(&acc_b[p], rb[q]);
						unsafeAtomicAdd(&acc_d[p], rd[q]);
						unsafeAtomicAdd(&acc_l[p], rl[q]);
					}
				}
			}
			__syncthreads();
		}
	} else {
#pragma unroll 1
	for (int pass = 0; pass < 2; ++pass) {
		for (uint32_t k = tid; k < in_rows; k += kThreads) {
			int r = 0;
#pragma unroll
			for (int j = 1; j < ROWS; ++j) { r += row_at[j] <= k ? 1 : 0; }
			const uint32_t i = row_first[r] + (k - row_at[r]);
			const int ry = r % (kTileY + 1), rz = r / (kTileY + 1);
			const int ly = lp0[1] - 1 + ry, lz = lp0[2] - 1 + rz;
			const int lx = static_cast<int>(cell_id[i] - row_key[r]);
			const int cx = lx - (lp0[0] - 1);  // 0 .. 64 inside the tile's reach
			if (cx < 0 || cx > kTileX) { continue; }
			const int colour = (lx & 1) | ((ly & 1) << 1) | ((lz & 1) << 2);
			if (pass == 0) {
				atomicAdd(&ncol[colour], 1);
			} else {
				const int at = base[colour] + atomicAdd(&fill[colour], 1);
				list_cell[at] = i;
				list_loc[at]  = static_cast<uint32_t>(cx) | (static_cast<uint32_t>(ry) << 8) | (static_cast<uint32_t>(rz) << 12) |
				                (static_cast<uint32_t>(colour) << 16);
			}
		}
		__syncthreads();
		if (pass == 0) {
			if (tid == 0) {
				int a = 0;
				for (int k = 0; k < NC; ++k) {
					base[k] = a;
					a += ncol[k];
				}
				base[NC] = a;
			}
			__syncthreads();
		}
	}
	const int total = base[NC];
	// chunks of one entry per thread: the record comes into registers in one round of loads, then the colours of the chunk
	// one after another (the list is in colour order, so are the chunks)
	for (int c0 = 0; c0 < total; c0 += kThreads) {
		const int  i    = c0 + tid;
		const bool live = i < total;
		T        rb[NC], rd[NC];
		float    rl[NC];
		uint32_t loc = 0;
		if (live) {
			const long c = list_cell[i];
			loc = list_loc[i];
#pragma unroll
			for (int q = 0; q < NC; ++q) {
				rb[q] = cell_dr[c * kRec * NC + q];
				rd[q] = cell_dr[c * kRec * NC + NC + q];
				rl[q] = lump ? static_cast<float>(cell_dr[c * kRec * NC + 2 * NC + q]) : 0.0f;
			}
		}
		const int last = c0 + kThreads < total ? c0 + kThreads - 1 : total - 1;
		const int col_lo = static_cast<int>(list_loc[c0] >> 16), col_hi = static_cast<int>(list_loc[last] >> 16);
		for (int colour = col_lo; colour <= col_hi; ++colour) {
			if (live && static_cast<int>(loc >> 16) == colour) {
				const int cx = static_cast<int>(loc & 0xFFu), ry = static_cast<int>((loc >> 8) & 0xFu), rz = static_cast<int>((loc >> 12) & 0xFu);
#pragma unroll
				for (int q = 0; q < NC; ++q) {
					const int px = cx - 1 + (q & 1), py = ry - 1 + ((q >> 1) & 1), pz = rz - 1 + ((q >> 2) & 1);
					if (px >= 0 && px < kTileX && py >= 0 && py < kTileY && pz >= 0 && pz < kTileZ) {
						const int p = (pz * kTileY + py) * kTileX + px;
						acc_b[p] += rb[q];
						acc_d[p] += rd[q];
						acc_l[p] += rl[q];
					}
				}
			}
			__syncthreads();
		}
	}
	}
	__syncthreads();
	for (int i = tid; i < NP; i += kThreads) {
		const int px = i % kTileX, py = (i / kTileX) % kTileY, pz = i / (kTileX * kTileY);
		if (x0 + px >= ext0 || y0 + py >= ext1 || z0 + pz >= ext2) { continue; }
		const int64_t idx = (g.own_lo[0] + x0 + px) * g.stride[0] + (g.own_lo[1] + y0 + py) * g.stride[1] +
		                    (g.own_lo[2] + z0 + pz) * g.stride[2];
		// (every owned point, written once and read much later: streaming stores keep these 335 MB out of the L2 the other
		// assembly chains' sorts and scans work in)
		__builtin_nontemporal_store(acc_b[i], atb + idx);
		__builtin_nontemporal_store(acc_d[i], diag + idx);
		if (lump) { __builtin_nontemporal_store(acc_l[i], lump + idx); }
	}
}

__global__ __launch_bounds__(kThreads) void k_cell_map(long ncell, const uint32_t* __restrict__ cell_id,
                                                        uint32_t* __restrict__ map)
{
	const long c = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
	if (c < ncell) { map[cell_id[c]] = static_cast<uint32_t>(c); }
}

template <int D, typename T>
__global__ __launch_bounds__(kThreads) void k_gather_cells(Geom g, long ncell, const uint32_t* __restrict__ map,
                                                            const T* __restrict__ cell_dr,
                                                            T* __restrict__ atb, T* __restrict__ diag, float* __restrict__ lump)
{
	constexpr int NC = 1 << D;
	int64_t o = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (o >= g.nown) { return; }
	int     lp[3] = {0, 0, 0};  // the point's coordinate on the extended cell grid
	int64_t idx = 0;
	for (int d = 0; d < D; ++d) {
		const int ext = g.own_hi[d] - g.own_lo[d];
		const int li  = g.own_lo[d] + static_cast<int>(o % ext);
		o /= ext;
		idx += static_cast<int64_t>(li) * g.stride[d];
		lp[d] = li + g.off[d] - g.coff[d];
	}
	// all 2^D map look-ups first (independent loads), then the few hits in colour order
	uint32_t cidx[NC];
	int      cq[NC];
#pragma unroll
	for (int colour = 0; colour < NC; ++colour) {
		// the incident cell whose origin has this parity: the point is its corner q
		uint32_t key = 0, mul = 1;
		int      q  = 0;
		bool     ok = true;
#pragma unroll
		for (int d = 0; d < D; ++d) {
			const int b = (lp[d] ^ (colour >> d)) & 1;
			const int l = lp[d] - b;
			ok = ok && l >= 0 && l < g.cn[d];
			key += static_cast<uint32_t>(l) * mul;
			mul *= static_cast<uint32_t>(g.cn[d]);
			q |= b << d;
		}
		cidx[colour] = ok ? map[key] : 0xFFFFFFFFu;
		cq[colour]   = q;
	}
	T a = T(0), dg = T(0);  // both arrays are zero when this kernel runs (assemble): written, not accumulated
	float lp1 = 0.0f;
#pragma unroll
	for (int colour = 0; colour < NC; ++colour) {
		const uint32_t c = cidx[colour];
		if (c == 0xFFFFFFFFu) { continue; }
		const int q = cq[colour];
		a += cell_dr[static_cast<long>(c) * kRec * NC + q];
		dg += cell_dr[static_cast<long>(c) * kRec * NC + NC + q];
		if (lump) { lp1 += static_cast<float>(cell_dr[static_cast<long>(c) * kRec * NC + 2 * NC + q]); }
	}
	atb[idx]  = a;
	diag[idx] = dg;
	if (lump) { lump[idx] = lp1; }
}

// {number of runs, key of the last run, length of the last run}
__global__ void k_rle_tail(const uint32_t* __restrict__ nruns, const uint32_t* __restrict__ uniq,
                           const uint32_t* __restrict__ counts, uint32_t* __restrict__ out, uint32_t* __restrict__ heavy_count)
{
	*heavy_count = 0u;  // (the list of many-row cells k_build_blocks* fill: one fill less per level)
	const uint32_t n = nruns[0];
	out[0] = n;
	out[1] = n ? uniq[n - 1] : 0u;
	out[2] = n ? counts[n - 1] : 0u;
}

inline int blocks_for(long n) { return static_cast<int>((n + kThreads - 1) / kThreads); }

template <int D>
void emit_rows_dim(fi_ctx* c, long n, const float* pos, const float* nrm, const float* pw, const float* val, float vw,
                   int vk, float gw, int gk, float pos_scale, float nrm_scale)
{
	constexpr int NC = 1 << D;
	Pending* pb = nullptr;
	if (!c->pending_pool.empty()) {
		pb = c->pending_pool.back();
		c->pending_pool.pop_back();
	} else {
		pb = new Pending();
	}
	c->pending.push_back(pb);
	// value rows only: one row per point (the rows the sort, the run lengths and the blocks walk: 4 M -> 1 M on config 4)
	const int rows_per_point = (nrm != nullptr && gw != 0.0f) ? 1 + D : 1;
	pb->nrows = n * rows_per_point;
	pb->key.alloc(sizeof(uint32_t) * pb->nrows);
	pb->coef.alloc(sizeof(float) * pb->nrows * NC);
	pb->rhs.alloc(sizeof(float) * pb->nrows);
	EmitArgs a;
	a.g = c->g;
	a.vw = vw;
	a.gw = gw;
	a.vk = vk;
	a.gk = gk;
	a.has_nrm = nrm != nullptr;
	a.rows_per_point = rows_per_point;
	a.has_pw  = pw != nullptr;
	a.has_val = val != nullptr;
	a.pos_scale = pos_scale;
	a.nrm_scale = nrm_scale;
	a.invalid_key = static_cast<uint32_t>(static_cast<int64_t>(c->g.cn[0]) * c->g.cn[1] * c->g.cn[2]);
	if (n > 0) {
		hipLaunchKernelGGL(k_emit_rows<D>, dim3(blocks_for(n)), dim3(kThreads), 0, c->stream, a, n, pos, nrm, pw, val,
		                   pb->key.as<uint32_t>(), pb->coef.as<float>(), pb->rhs.as<float>());
		FI_HIP_TRY(hipGetLastError());
	}
}

template <int D, typename T>
void assemble_dim(fi_ctx* c)
{
	constexpr int NC = 1 << D;
	constexpr int NB = NC * (NC + 1) / 2;
	hipStream_t st = c->stream;
	const Geom& g  = c->g;
	const uint32_t invalid = static_cast<uint32_t>(static_cast<int64_t>(g.cn[0]) * g.cn[1] * g.cn[2]);

	// operator arrays over local storage
	const bool fresh = c->atb.bytes < sizeof(T) * g.nloc || c->diag.bytes < sizeof(T) * g.nloc;
	c->atb.alloc(sizeof(T) * g.nloc);
	c->diag.alloc(sizeof(T) * g.nloc);
	// (zeroed below, once the path of the sums over the lattice points is known: the gather writes every owned point itself)
	// row sums of the data term for the lumped replica of a mixed-precision context (fi_assemble asks: want_lump)
	float* lump = nullptr;
	if (c->want_lump) {
		c->lump.alloc(sizeof(float) * g.nloc);
		lump = c->lump.as<float>();
	}
	auto zero_operator = [&]() {
		FI_HIP_TRY(hipMemsetAsync(c->atb.p, 0, sizeof(T) * g.nloc, st));
		FI_HIP_TRY(hipMemsetAsync(c->diag.p, 0, sizeof(T) * g.nloc, st));
		if (lump) { FI_HIP_TRY(hipMemsetAsync(lump, 0, sizeof(float) * g.nloc, st)); }
	};

	long total = 0;
	for (auto* pb : c->pending) { total += pb->nrows; }
	c->cells.ncell = 0;
	c->cells.nb    = NB;
	c->stats.num_data_rows = 0;
	c->stats.num_cells     = 0;
	if (total == 0) {
		// (an undivided lumped replica: its diagonal is written whole by operator_prepare, its A^T b is never read and stays
		// the zeros of its first assembly -- two fills of 67 MB per assemble at 256^3)
		if (!(c->lumped && g.nown == g.nloc && c->nranks == 1 && !fresh)) { zero_operator(); }
		return;
	}
	FI_REQUIRE(total < (1L << 31), FI_ERR_UNSUPPORTED, "more than 2^31 data rows in one context");

	// gather the batches into one row table (single batch: used in place)
	DevBuf &key_cat = c->scratch[0], &coef_cat = c->scratch[1], &rhs_cat = c->scratch[2];
	const uint32_t* key  = nullptr;
	const float*    coef = nullptr;
	const float*    rhs  = nullptr;
	if (c->pending.size() == 1) {
		key  = c->pending[0]->key.as<uint32_t>();
		coef = c->pending[0]->coef.as<float>();
		rhs  = c->pending[0]->rhs.as<float>();
	} else {
		key_cat.alloc(sizeof(uint32_t) * total);
		coef_cat.alloc(sizeof(float) * total * NC);
		rhs_cat.alloc(sizeof(float) * total);
		long at = 0;
		for (auto* pb : c->pending) {
			if (pb->nrows == 0) { continue; }
			FI_HIP_TRY(hipMemcpyAsync(key_cat.as<uint32_t>() + at, pb->key.p, sizeof(uint32_t) * pb->nrows,
			                          hipMemcpyDeviceToDevice, st));
			FI_HIP_TRY(hipMemcpyAsync(coef_cat.as<float>() + at * NC, pb->coef.p, sizeof(float) * pb->nrows * NC,
			                          hipMemcpyDeviceToDevice, st));
			FI_HIP_TRY(hipMemcpyAsync(rhs_cat.as<float>() + at, pb->rhs.p, sizeof(float) * pb->nrows,
			                          hipMemcpyDeviceToDevice, st));
			at += pb->nrows;
		}
		key  = key_cat.as<uint32_t>();
		coef = coef_cat.as<float>();
		rhs  = rhs_cat.as<float>();
	}

	// sort rows by cell
	DevBuf &row_in = c->scratch[3], &row_sorted = c->scratch[4], &key_sorted = c->scratch[5], &uniq = c->scratch[6],
	       &counts = c->scratch[7], &starts = c->scratch[8], &nruns = c->scratch[9], &tmp = c->scratch[10];
	row_in.alloc(sizeof(uint32_t) * total);
	row_sorted.alloc(sizeof(uint32_t) * total);
	key_sorted.alloc(sizeof(uint32_t) * total);
	uniq.alloc(sizeof(uint32_t) * total);
	counts.alloc(sizeof(uint32_t) * total);
	starts.alloc(sizeof(uint32_t) * total);
	nruns.alloc(sizeof(uint32_t) * 4);
	// (an index array and a pair sort: with a counting iterator as the values rocPRIM first copies keys and values into its
	// own buffers -- two transform launches instead of this one)
	hipLaunchKernelGGL(k_iota, dim3(blocks_for(total)), dim3(kThreads), 0, st, row_in.as<uint32_t>(), total);
	int end_bit = 1;
	while ((1ull << end_bit) <= invalid) { ++end_bit; }
	size_t tb = 0;
	FI_HIP_TRY(sort_pairs_u32(nullptr, tb, key, key_sorted.as<uint32_t>(), row_in.as<uint32_t>(), row_sorted.as<uint32_t>(),
	                          static_cast<unsigned int>(total), 0, end_bit, st));
	tmp.alloc(tb);
	FI_HIP_TRY(sort_pairs_u32(tmp.p, tb, key, key_sorted.as<uint32_t>(), row_in.as<uint32_t>(), row_sorted.as<uint32_t>(),
	                          static_cast<unsigned int>(total), 0, end_bit, st));
	// runs of equal keys = occupied cells (+ one run of invalid rows at the end)
	size_t tb2 = 0;
	FI_HIP_TRY(prim::run_length_encode(nullptr, tb2, key_sorted.as<uint32_t>(), uniq.as<uint32_t>(), counts.as<uint32_t>(),
	                                    nruns.as<uint32_t>(), static_cast<size_t>(total), st));
	DevBuf& tmp2 = c->scratch[11];
	tmp2.alloc(tb2);
	FI_HIP_TRY(prim::run_length_encode(tmp2.p, tb2, key_sorted.as<uint32_t>(), uniq.as<uint32_t>(), counts.as<uint32_t>(),
	                                    nruns.as<uint32_t>(), static_cast<size_t>(total), st));
	// the run count and the last run (the invalid rows, if any) in ONE host round trip
	DevBuf& tail3 = c->scratch[26];
	tail3.alloc(sizeof(uint32_t) * 3);
	DevBuf& heavy = c->scratch[28];  // cells with very many rows (coarse levels): listed by k_build_blocks*, summed by k_build_heavy
	heavy.alloc(sizeof(uint32_t) * (static_cast<size_t>(total / kHeavyRows) + 2));
	hipLaunchKernelGGL(k_rle_tail, dim3(1), dim3(1), 0, st, nruns.as<uint32_t>(), uniq.as<uint32_t>(), counts.as<uint32_t>(),
	                   tail3.as<uint32_t>(), heavy.as<uint32_t>());
	uint32_t* h_tail = static_cast<uint32_t*>(pinned(c, 0, 3 * sizeof(uint32_t)));
	FI_HIP_TRY(hipMemcpyAsync(h_tail, tail3.p, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
	FI_HIP_TRY(hipStreamSynchronize(st));
	const uint32_t h_runs = h_tail[0];
	if (h_runs == 0) {
		zero_operator();
		return;
	}
	const uint32_t h_last_key = h_tail[1], h_last_count = h_tail[2];
	long ncell   = h_runs;
	long invalid_rows = 0;
	if (h_last_key == invalid) {
		ncell -= 1;
		invalid_rows = h_last_count;
	}
	c->stats.num_data_rows = total - invalid_rows;
	c->stats.num_cells     = ncell;
	if (ncell == 0) {
		zero_operator();
		return;
	}
	size_t tb3 = 0;
	FI_HIP_TRY(prim::exclusive_sum(nullptr, tb3, counts.as<uint32_t>(), starts.as<uint32_t>(), static_cast<size_t>(h_runs), st));
	DevBuf& tmp3 = c->scratch[12];
	tmp3.alloc(tb3);
	FI_HIP_TRY(prim::exclusive_sum(tmp3.p, tb3, counts.as<uint32_t>(), starts.as<uint32_t>(), static_cast<size_t>(h_runs), st));

	c->cells.ncell = ncell;
	// two or more data rows per occupied cell on average: multi-row cells as packed blocks (finish_cell)
	c->cells.pack = (total - invalid_rows) >= 2 * ncell && !test_switch("FI_NO_PACK");
	const uint32_t pack_min = c->cells.pack ? 3u : static_cast<uint32_t>(NC) + 1u;
	c->cells.cell_id.alloc(sizeof(uint32_t) * ncell);
	c->cells.blk.alloc(sizeof(T) * NB * ncell);
	c->cells.nrow.alloc(sizeof(uint32_t) * ncell);
	c->cells.row1.alloc(sizeof(T) * NC * ncell);
	if (D == 3) {  // factor rows feed the fused 3-D kernel only
		c->cells.mrow.alloc(sizeof(T) * NC * NC * ncell);
		c->cells.nfac.alloc(sizeof(uint32_t) * ncell);
	}
	DevBuf& cell_dr = c->scratch[13];
	cell_dr.alloc(sizeof(T) * kRec * NC * ncell);
	// (the cells' ids travel from the run-length encoding to CellData::cell_id inside k_build_blocks*: no copy of its own)
	// The packed blocks themselves are what the kernels OUTSIDE the fused marching kernel read (the cell kernel of the
	// untiled path and of fi_tile_pass, the 2-D tile kernel): a 3-D context the marching kernel will cover keeps only the
	// factor rows and forms the blocks when somebody asks (ensure_cell_blocks) -- 280 MB of the 530 MB the fp64 level of
	// config 4 wrote here, at the 3.6 TB/s these stores reach (k_build_blocks3 160 -> 82 us).
	const bool keep_blocks = !(D == 3 && stencil_will_fuse(c)) || test_switch("FI_BLOCKS_PER_THREAD") || test_switch("FI_KEEP_BLOCKS");
	c->cells.blk_valid = keep_blocks;
	if (D == 3 && !test_switch("FI_BLOCKS_PER_THREAD")) {
		hipLaunchKernelGGL((k_build_blocks3<T>), dim3(blocks_for(ncell * 8)), dim3(kThreads), 0, st, ncell, starts.as<uint32_t>(),
		                   counts.as<uint32_t>(), row_sorted.as<uint32_t>(), coef, rhs, c->cells.blk.as<T>(), cell_dr.as<T>(),
		                   c->cells.nrow.as<uint32_t>(), c->cells.row1.as<T>(), c->cells.mrow.as<T>(), c->cells.nfac.as<uint32_t>(),
		                   heavy.as<uint32_t>(), pack_min, keep_blocks ? 1 : 0, uniq.as<uint32_t>(), c->cells.cell_id.as<uint32_t>());
	} else
	hipLaunchKernelGGL((k_build_blocks<D, T>), dim3(blocks_for(ncell)), dim3(kThreads), 0, st, ncell,
	                   starts.as<uint32_t>(), counts.as<uint32_t>(), row_sorted.as<uint32_t>(), coef, rhs,
	                   c->cells.blk.as<T>(), cell_dr.as<T>(), c->cells.nrow.as<uint32_t>(), c->cells.row1.as<T>(),
	                   D == 3 ? c->cells.mrow.as<T>() : static_cast<T*>(nullptr),
	                   D == 3 ? c->cells.nfac.as<uint32_t>() : static_cast<uint32_t*>(nullptr), heavy.as<uint32_t>(), pack_min,
	                   uniq.as<uint32_t>(), c->cells.cell_id.as<uint32_t>());
	{
		const long max_heavy = total / kHeavyRows + 1;
		const int  grid = static_cast<int>(max_heavy < 2048 ? max_heavy : 2048);
		hipLaunchKernelGGL((k_build_heavy<D, T>), dim3(grid), dim3(kThreads), 0, st, ncell, starts.as<uint32_t>(),
		                   counts.as<uint32_t>(), row_sorted.as<uint32_t>(), coef, rhs, c->cells.blk.as<T>(),
		                   cell_dr.as<T>(), c->cells.nrow.as<uint32_t>(), c->cells.row1.as<T>(),
		                   D == 3 ? c->cells.mrow.as<T>() : static_cast<T*>(nullptr),
		                   D == 3 ? c->cells.nfac.as<uint32_t>() : static_cast<uint32_t*>(nullptr), heavy.as<uint32_t>(), pack_min);
	}
	FI_HIP_TRY(hipGetLastError());
	int64_t ncells_ext = 1;
	for (int d = 0; d < D; ++d) { ncells_ext *= g.cn[d]; }
	// One launch over the POINTS (dense cell map, 2^D look-ups per point) on small lattices, 2^D launches over the CELLS
	// elsewhere: the gather costs by the lattice (256^3 fp64: 310 us alone on the GPU, 6 % of the cells occupied), the
	// scatter by the occupied cells (8 launches of ~8 us there); below ~64^3 the launches outweigh the look-ups.
	int64_t gather_max = 1 << 18;
	if (const char* e = tuning_switch("FI_GATHER_MAX")) { gather_max = atoll(e); }
	if (test_switch("FI_GATHER_ALWAYS")) { gather_max = 1LL << 32; }
	const bool gather = static_cast<int64_t>(ncell) * 64 >= ncells_ext && ncells_ext < (1LL << 32) && ncells_ext <= gather_max &&
	                    !test_switch("FI_NO_GATHER");
	const int ext0 = g.own_hi[0] - g.own_lo[0], ext1 = g.own_hi[1] - g.own_lo[1], ext2 = g.own_hi[2] - g.own_lo[2];
	const bool tiles = D == 3 && !test_switch("FI_NO_GATHER") && !test_switch("FI_NO_TILE_SUMS") && ncells_ext < (1LL << 32) &&
	                   (ext1 + kTileY - 1) / kTileY <= 65535 && (ext2 + kTileZ - 1) / kTileZ <= 65535;
	if (!(gather || tiles) || g.nown != g.nloc) { zero_operator(); }
	if (tiles) {
		const int     ntx   = (ext0 + kTileX - 1) / kTileX;
		const int64_t nrows = static_cast<int64_t>(g.cn[1]) * g.cn[2];
		DevBuf& rb = c->scratch[27];
		rb.alloc(sizeof(uint32_t) * (nrows * (ntx + 1) * 2 + 4));
		hipLaunchKernelGGL(k_xtile_bounds, dim3(blocks_for(nrows * (ntx + 1) * 2)), dim3(kThreads), 0, st, nrows, ntx,
		                   g.own_lo[0] + g.off[0] - g.coff[0], g.cn[0], ncell, c->cells.cell_id.as<uint32_t>(), rb.as<uint32_t>());
		hipLaunchKernelGGL((k_tile_sums3<T>), dim3((ext0 + kTileX - 1) / kTileX, (ext1 + kTileY - 1) / kTileY, (ext2 + kTileZ - 1) / kTileZ),
		                   dim3(kThreads), 0, st, g, rb.as<uint32_t>(), ntx, c->cells.cell_id.as<uint32_t>(), cell_dr.as<T>(), c->atb.as<T>(),
		                   c->diag.as<T>(), lump);
	} else if (gather) {
		DevBuf& map = c->scratch[24];
		map.alloc(sizeof(uint32_t) * ncells_ext);
		FI_HIP_TRY(hipMemsetAsync(map.p, 0xFF, sizeof(uint32_t) * ncells_ext, st));
		hipLaunchKernelGGL(k_cell_map, dim3(blocks_for(ncell)), dim3(kThreads), 0, st, ncell,
		                   c->cells.cell_id.as<uint32_t>(), map.as<uint32_t>());
		if (D == 3) {
			const int64_t groups = static_cast<int64_t>((g.own_hi[0] - g.own_lo[0] + 3) / 4) * (g.own_hi[1] - g.own_lo[1]) *
			                       (g.own_hi[2] - g.own_lo[2]);
			hipLaunchKernelGGL((k_gather_cells3<T>), dim3(blocks_for(groups)), dim3(kThreads), 0, st, g, ncell, map.as<uint32_t>(),
			                   cell_dr.as<T>(), c->atb.as<T>(), c->diag.as<T>(), lump);
		} else {
			hipLaunchKernelGGL((k_gather_cells<D, T>), dim3(blocks_for(g.nown)), dim3(kThreads), 0, st, g, ncell,
			                   map.as<uint32_t>(), cell_dr.as<T>(), c->atb.as<T>(), c->diag.as<T>(), lump);
		}
	} else {
		for (int colour = 0; colour < NC; ++colour) {
			hipLaunchKernelGGL((k_scatter_cells<D, T>), dim3(blocks_for(ncell)), dim3(kThreads), 0, st, g, ncell,
			                   c->cells.cell_id.as<uint32_t>(), cell_dr.as<T>(), c->atb.as<T>(), c->diag.as<T>(), lump, colour);
		}
	}
	FI_HIP_TRY(hipGetLastError());
	// no host round trip here: every buffer above is scratch the context owns, and whatever comes next is enqueued on
	// the same stream
}

}  // namespace

// ---- border prior (the reference application: src/sdf_field.cpp:218-246) ------------------------------------------
// Every lattice point on the border of the lattice gets the value row [1] * w = d * w, d = the distance to the nearest
// data point -- "far from the surface the field is positive and about the distance".  The reference loops over the
// border points and, for each, over all points (O(border x points), fp32: dx*dx + dy*dy, min, sqrt).  Here: the border
// points are enumerated in lattice order by a stream compaction of the index range (no lattice-sized buffer), one
// thread per border point walks the point batches in LDS tiles of 256 positions with the same fp32 arithmetic, and
// the rows enter the assembly as a batch of nearest-neighbour value constraints AT the lattice points
// (field_interpolation.cpp:82-107 with a zero gradient gives exactly the row [1] * w, rhs value * w).
namespace {

struct BorderPred {
	int n[3];
	int ndim;
	__host__ __device__ bool operator()(const uint32_t& i) const
	{
		uint32_t r = i;
		bool border = false;
		for (int d = 0; d < ndim; ++d) {
			const int cd = static_cast<int>(r % static_cast<uint32_t>(n[d]));
			r /= static_cast<uint32_t>(n[d]);
			border = border || cd == 0 || cd == n[d] - 1;
		}
		return border;
	}
};

template <int D>
__global__ __launch_bounds__(256) void k_border_min_dist(int64_t nb, const uint32_t* __restrict__ idx, BorderPred g, long npts,
                                                          const float* __restrict__ pos, float* __restrict__ d2)
{
	__shared__ float tile[256 * D];
	const int64_t b = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
	float x[D];
	{
		uint32_t r = b < nb ? idx[b] : 0u;
		for (int d = 0; d < D; ++d) {
			x[d] = static_cast<float>(r % static_cast<uint32_t>(g.n[d]));
			r /= static_cast<uint32_t>(g.n[d]);
		}
	}
	float best = b < nb ? d2[b] : 0.0f;
	for (long base = 0; base < npts; base += 256) {
		const long m = npts - base < 256 ? npts - base : 256;
		__syncthreads();
		for (long k = threadIdx.x; k < m * D; k += 256) { tile[k] = pos[base * D + k]; }
		__syncthreads();
		for (long k = 0; k < m; ++k) {
			float s = 0.0f;
#pragma unroll
			for (int d = 0; d < D; ++d) {
				const float dd = tile[k * D + d] - x[d];   // pos - lattice coordinate, as the reference
				s = s + dd * dd;
			}
			best = s < best ? s : best;
		}
	}
	if (b < nb) { d2[b] = best; }
}

template <int D>
__global__ __launch_bounds__(256) void k_border_finish(int64_t nb, const uint32_t* __restrict__ idx, BorderPred g,
                                                        const float* __restrict__ d2, float* __restrict__ pos,
                                                        float* __restrict__ val)
{
	const int64_t b = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
	if (b >= nb) { return; }
	uint32_t r = idx[b];
	for (int d = 0; d < D; ++d) {
		pos[b * D + d] = static_cast<float>(r % static_cast<uint32_t>(g.n[d]));
		r /= static_cast<uint32_t>(g.n[d]);
	}
	val[b] = sqrtf(d2[b]);
}

__global__ void k_fill_f32(int64_t n, float v, float* out)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i < n) { out[i] = v; }
}

}  // namespace

// -> number of border points; their coordinates and the nearest-point distances in pos / val (device buffers)
int64_t border_prior_points(fi_ctx* c, DevBuf& pos, DevBuf& val)
{
	const Geom& g = c->g;
	const int D = g.ndim;
	int64_t total = 1;
	for (int d = 0; d < D; ++d) { total *= g.gn[d]; }
	FI_REQUIRE(total <= 0x7fffffffLL, FI_ERR_UNSUPPORTED, "border prior: lattice too large for the 32-bit item count of the selection");
	BorderPred pred{{g.gn[0], D > 1 ? g.gn[1] : 1, D > 2 ? g.gn[2] : 1}, D};
	int64_t inner = 1;
	for (int d = 0; d < D; ++d) { inner *= g.gn[d] > 2 ? g.gn[d] - 2 : 0; }
	const int64_t nb_expected = total - inner;
	DevBuf idx, count, tmp, d2;
	idx.alloc(sizeof(uint32_t) * (nb_expected + 1));
	count.alloc(sizeof(int));
	hipStream_t st = c->stream;
	size_t tb = 0;
	FI_HIP_TRY(prim::select_indices(nullptr, tb, idx.as<uint32_t>(), count.as<int>(), static_cast<size_t>(total), pred, st));
	tmp.alloc(tb);
	FI_HIP_TRY(prim::select_indices(tmp.p, tb, idx.as<uint32_t>(), count.as<int>(), static_cast<size_t>(total), pred, st));
	int nb = 0;
	FI_HIP_TRY(hipMemcpyAsync(&nb, count.p, sizeof(int), hipMemcpyDeviceToHost, st));
	FI_HIP_TRY(hipStreamSynchronize(st));
	FI_REQUIRE(nb == nb_expected, FI_ERR_HIP, "border prior: %d border points selected, %lld expected", nb,
	           static_cast<long long>(nb_expected));
	d2.alloc(sizeof(float) * nb);
	pos.alloc(sizeof(float) * nb * D);
	val.alloc(sizeof(float) * nb);
	const int blocks = static_cast<int>((nb + 255) / 256);
	hipLaunchKernelGGL(k_fill_f32, dim3(blocks), dim3(256), 0, st, static_cast<int64_t>(nb), INFINITY, d2.as<float>());
	for (const PointBatch* b : c->batches) {
		if (b->n <= 0 || b->prior) { continue; }
		switch (D) {
		case 1: hipLaunchKernelGGL(k_border_min_dist<1>, dim3(blocks), dim3(256), 0, st, static_cast<int64_t>(nb), idx.as<uint32_t>(), pred, b->n, b->pos.as<float>(), d2.as<float>()); break;
		case 2: hipLaunchKernelGGL(k_border_min_dist<2>, dim3(blocks), dim3(256), 0, st, static_cast<int64_t>(nb), idx.as<uint32_t>(), pred, b->n, b->pos.as<float>(), d2.as<float>()); break;
		default: hipLaunchKernelGGL(k_border_min_dist<3>, dim3(blocks), dim3(256), 0, st, static_cast<int64_t>(nb), idx.as<uint32_t>(), pred, b->n, b->pos.as<float>(), d2.as<float>()); break;
		}
	}
	switch (D) {
	case 1: hipLaunchKernelGGL(k_border_finish<1>, dim3(blocks), dim3(256), 0, st, static_cast<int64_t>(nb), idx.as<uint32_t>(), pred, d2.as<float>(), pos.as<float>(), val.as<float>()); break;
	case 2: hipLaunchKernelGGL(k_border_finish<2>, dim3(blocks), dim3(256), 0, st, static_cast<int64_t>(nb), idx.as<uint32_t>(), pred, d2.as<float>(), pos.as<float>(), val.as<float>()); break;
	default: hipLaunchKernelGGL(k_border_finish<3>, dim3(blocks), dim3(256), 0, st, static_cast<int64_t>(nb), idx.as<uint32_t>(), pred, d2.as<float>(), pos.as<float>(), val.as<float>()); break;
	}
	FI_HIP_TRY(hipGetLastError());
	FI_HIP_TRY(hipStreamSynchronize(st));  // the temporaries die here
	return nb;
}

namespace {
// the packed block of a 3-D cell from what the assembly kept of it: its rows (<= 8: the sum of their outer products, the
// very sums k_build_blocks3 forms), the packed block of a packed cell, or the Cholesky factor rows U of a many-row cell
// of an fp32 context (U^T U = the block to fp32 rounding)
template <typename T>
__global__ __launch_bounds__(kThreads) void k_blocks_from_factors(long ncell, const uint32_t* __restrict__ nfac, const uint32_t* __restrict__ nrow,
                                                                   const T* __restrict__ row1, const T* __restrict__ mrow, T* __restrict__ blk)
{
	constexpr int NC = 8, NB = 36;
	const long c = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
	if (c >= ncell) { return; }
	const uint32_t k = nfac[c];
	if (k == 0xFFu) {
		for (int e = 0; e < NB; ++e) { blk[c * NB + e] = mrow[c * NC * NC + e]; }
		return;
	}
	// A cell of exactly ONE data row keeps no factor row of its own: the row itself is row1.  (nfac == 1 does not say that: the
	// Cholesky factor of a many-row cell whose block has rank one -- more than 8 nearest-neighbour rows on one corner,
	// duplicated points -- is ONE row of mrow, and row1 is then only the first of the data rows.)
	const bool single = nrow[c] == 1u;
	double B[NB];
	for (int e = 0; e < NB; ++e) { B[e] = 0.0; }
	for (uint32_t r = 0; r < k; ++r) {
		const T* f = single ? row1 + c * NC : mrow + (c * NC + r) * NC;
		double a[NC];
		for (int q = 0; q < NC; ++q) { a[q] = static_cast<double>(f[q]); }
		int e = 0;
		for (int i = 0; i < NC; ++i) {
			for (int j = i; j < NC; ++j) { B[e++] += a[i] * a[j]; }
		}
	}
	for (int e = 0; e < NB; ++e) { blk[c * NB + e] = static_cast<T>(B[e]); }
}
}  // namespace

void ensure_cell_blocks(fi_ctx* c)
{
	if (c->cells.blk_valid || c->cells.ncell == 0) { return; }
	FI_REQUIRE(c->g.ndim == 3 && c->cells.mrow.p && c->cells.nfac.p, FI_ERR_STATE, "no cell blocks and nothing to form them from");
	const long n = static_cast<long>(c->cells.ncell);
	if (c->dtype == FI_F64) {
		hipLaunchKernelGGL((k_blocks_from_factors<double>), dim3(blocks_for(n)), dim3(kThreads), 0, c->stream, n, c->cells.nfac.as<uint32_t>(),
		                   c->cells.nrow.as<uint32_t>(), c->cells.row1.as<double>(), c->cells.mrow.as<double>(), c->cells.blk.as<double>());
	} else {
		hipLaunchKernelGGL((k_blocks_from_factors<float>), dim3(blocks_for(n)), dim3(kThreads), 0, c->stream, n, c->cells.nfac.as<uint32_t>(),
		                   c->cells.nrow.as<uint32_t>(), c->cells.row1.as<float>(), c->cells.mrow.as<float>(), c->cells.blk.as<float>());
	}
	FI_HIP_TRY(hipGetLastError());
	c->cells.blk_valid = true;
}

void* pinned(fi_ctx* c, int slot, size_t bytes)
{
	if (c->pin_bytes[slot] < bytes) {
		if (c->pin[slot]) {
			// (the block may still be the source or target of a copy on the context's stream: drained before another context
			// can take it from the pool -- growth is rare, a few times in a context's life)
			(void)hipStreamSynchronize(c->stream);
			pinned_give(c->pin[slot], c->pin_bytes[slot]);
			c->pin[slot] = nullptr;
			c->pin_bytes[slot] = 0;
		}
		size_t cap = 0;
		c->pin[slot] = pinned_take(bytes, &cap);
		c->pin_bytes[slot] = cap;
	}
	return c->pin[slot];
}

void emit_point_rows(fi_ctx* c, long n, const float* pos, const float* nrm, const float* pw, const float* val, float vw,
                     int vk, float gw, int gk, float pos_scale, float nrm_scale)
{
	switch (c->g.ndim) {
	case 1: emit_rows_dim<1>(c, n, pos, nrm, pw, val, vw, vk, gw, gk, pos_scale, nrm_scale); break;
	case 2: emit_rows_dim<2>(c, n, pos, nrm, pw, val, vw, vk, gw, gk, pos_scale, nrm_scale); break;
	default: emit_rows_dim<3>(c, n, pos, nrm, pw, val, vw, vk, gw, gk, pos_scale, nrm_scale); break;
	}
}

void assemble(fi_ctx* c)
{
	const bool f64 = c->dtype == FI_F64;
	switch (c->g.ndim) {
	case 1: f64 ? assemble_dim<1, double>(c) : assemble_dim<1, float>(c); break;
	case 2: f64 ? assemble_dim<2, double>(c) : assemble_dim<2, float>(c); break;
	default: f64 ? assemble_dim<3, double>(c) : assemble_dim<3, float>(c); break;
	}
}

}  // namespace fi
