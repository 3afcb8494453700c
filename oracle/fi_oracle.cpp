// fi_oracle.cpp -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A plain, single-threaded C++ restatement of the reference hot path
// (emilk/field_interpolation: field_interpolation/field_interpolation.cpp and
// field_interpolation/sparse_linear.cpp).  Every function cites the reference
// file:line whose behaviour it restates.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library; the product
// (field_interpolation_amd/, include/) never links, imports or calls it.
//
// PARITY STATUS: **parity unpinned by reference tests** -- the reference ships no
// tests, golden vectors or fixtures, and its sources cannot be
// built in this image (they need <loguru.hpp> from an absent submodule and
// Eigen 3, which is not installed; no stand-ins are written for either).  The
// only known answers that exist are pinned in tests/test_oracle_known_answers.py:
//   * README.md:29-40  -- the 8x6 worked example (matrix A and rhs b),
//   * SURVEY.md 8(c)   -- survey-time outputs of the reference assembly for the
//                         field_1d.cpp:20-29 default input (14 rows/38 triplets and
//                         the float64 least-squares solution),
//   * SURVEY.md 8 table-- closed-form row/triplet counts for configs C1..C5.
// A second, independent numpy restatement (oracle/fi_oracle_py.py) is checked
// against this file triplet by triplet.
//
// Eigen 3 (un-vendored, un-pinned; 3.3.x era, sparse_linear.cpp:3-5) holds the
// solver arithmetic in the reference.  Its algorithms are restated here from
// their published definitions: setFromTriplets duplicate summation in input
// order, A^T*A by column accumulation, SimplicialLLT -> Cholesky (here banded),
// BiCGSTAB with DiagonalPreconditioner (van der Vorst 1992, with Eigen's restart
// rule), relative residual stop ||r|| <= tol*||rhs||.
//
// Build: see oracle/Makefile (g++ -O2 -ffp-contract=off, no -march flags, so the
// fp32 arithmetic is evaluated operation by operation like the reference build,
// build.sh:68 "-O2 -DNDEBUG").

#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <vector>

namespace fio {

static const int kMaxDim = 3;  // field_interpolation.hpp:44

// ---------------------------------------------------------------------------------------------
// Storage.  sparse_linear.hpp:8-22 (Triplet, LinearEquation).

struct Entry {
	int32_t row, col;
	float   val;
};

struct System {
	std::vector<Entry> ent;
	std::vector<float> rhs;
};

// sparse_linear.cpp:34-50  add_equation: a zero weight drops the row; zero coefficients
// are skipped; the rhs is only recorded when at least one coefficient survived.
static void push_row(System* s, float weight, float rhs, int n, const int* cols, const float* coef)
{
	if (weight == 0.0f) { return; }
	const int row = static_cast<int>(s->rhs.size());
	bool kept_any = false;
	for (int k = 0; k < n; ++k) {
		if (coef[k] != 0.0f) {
			s->ent.push_back(Entry{row, cols[k], coef[k] * weight});
			kept_any = true;
		}
	}
	if (kept_any) { s->rhs.push_back(rhs * weight); }
}

// field_interpolation.hpp:75-95  Weights (enum values: hpp:47-59).
struct Weights {
	float data_pos, data_gradient;
	float model_0, model_1, model_2, model_3, model_4;
	float gradient_smoothness;
	int   value_kernel;     // 0 nearest neighbour, 1 linear interpolation
	int   gradient_kernel;  // 0 nearest neighbour, 1 cell edges, 2 linear interpolation
};

// field_interpolation.hpp:97-114  LatticeField: x is the fastest axis.
struct Field {
	System sys;
	int    ndim;
	int    size[kMaxDim];
	int    stride[kMaxDim];
};

static void field_init(Field* f, int ndim, const int* sizes)
{
	f->ndim = ndim;
	int s = 1;
	for (int d = 0; d < kMaxDim; ++d) {
		f->size[d]   = d < ndim ? sizes[d] : 1;
		f->stride[d] = s;
		s *= f->size[d];
	}
}

// ---------------------------------------------------------------------------------------------
// field_interpolation.cpp:15-55  multilerp.  Corner i takes the upper neighbour along d when bit
// d of i is set.  Corners outside [0, size-extra) are dropped and the kept weights are NOT
// renormalised.  The weight product runs over d in ascending order, in fp32.
static int corner_weights(int* out_index, float* out_w, int ndim, const int* size, const int* stride,
                          const float* pos, int extra)
{
	int   base[kMaxDim];
	float frac[kMaxDim];
	for (int d = 0; d < ndim; ++d) {
		base[d] = static_cast<int>(std::floor(pos[d]));
		frac[d] = pos[d] - static_cast<float>(base[d]);
	}
	int kept = 0;
	for (int c = 0; c < (1 << ndim); ++c) {
		int   idx = 0;
		float w   = 1.0f;
		bool  ok  = true;
		for (int d = 0; d < ndim; ++d) {
			const int up = (c >> d) & 1;
			const int q  = base[d] + up;
			idx += stride[d] * q;
			w *= up ? frac[d] : 1.0f - frac[d];
			ok = ok && (0 <= q) && (q + extra < size[d]);
		}
		if (ok) {
			out_index[kept] = idx;
			out_w[kept]     = w;
			++kept;
		}
	}
	return kept;
}

// field_interpolation.cpp:57-80  add_value_constraint.  Coefficients are pushed even when they
// are exactly zero (no add_equation filtering on this path); rhs = (sum of kept weights)*value.
static bool value_row(Field* f, const float* pos, float value, float cw)
{
	if (cw == 0.0f) { return false; }
	int   idx[16];
	float w[16];
	const int n = corner_weights(idx, w, f->ndim, f->size, f->stride, pos, 0);
	if (n == 0) { return false; }
	const int row = static_cast<int>(f->sys.rhs.size());
	float sum = 0.0f;
	for (int k = 0; k < n; ++k) {
		const float c = w[k] * cw;
		f->sys.ent.push_back(Entry{row, idx[k], c});
		sum += c;
	}
	f->sys.rhs.push_back(sum * value);
	return true;
}

// field_interpolation.cpp:82-107  add_value_constraint_nearest_neighbor.  std::round (half away
// from zero) per axis; any axis out of range rejects the point.
static bool value_row_nearest(Field* f, const float* pos, const float* grad, float value, float cw)
{
	int   idx   = 0;
	float along = 0.0f;
	for (int d = 0; d < f->ndim; ++d) {
		const int q = static_cast<int>(std::round(pos[d]));
		if (q < 0 || f->size[d] <= q) { return false; }
		along += (pos[d] - static_cast<float>(q)) * grad[d];
		idx += q * f->stride[d];
	}
	const int   col  = idx;
	const float one  = 1.0f;
	push_row(&f->sys, cw, value - along, 1, &col, &one);
	return true;
}

// field_interpolation.cpp:110-121  cell_index: origin of the cell holding pos, or -1.
static int cell_origin(const Field* f, const float* pos)
{
	int idx = 0;
	for (int d = 0; d < f->ndim; ++d) {
		const int q = static_cast<int>(std::floor(pos[d]));
		if (!(0 <= q && q + 1 < f->size[d])) { return -1; }
		idx += q * f->stride[d];
	}
	return idx;
}

// field_interpolation.cpp:123-240  add_gradient_constraint.
// returns 1 (added), 0 (ignored), -1 (unknown kernel: the reference ABORT_Fs, cpp:238).
static int gradient_rows(Field* f, const float* pos, const float* grad, float cw, int kernel)
{
	if (cw == 0.0f) { return 0; }
	const int D = f->ndim;
	if (kernel == 0) {
		// cpp:134-149: per axis, [-1 at cell, +1 at cell+stride] = g_d, through add_equation.
		const int o = cell_origin(f, pos);
		if (o < 0) { return 0; }
		for (int d = 0; d < D; ++d) {
			const int   cols[2] = {o, o + f->stride[d]};
			const float coef[2] = {-1.0f, +1.0f};
			push_row(&f->sys, cw, grad[d], 2, cols, coef);
		}
		return 1;
	}
	if (kernel == 1) {
		// cpp:150-187: one row per axis over all 2^D corners, +-cw*2/2^D, rhs cw*g_d.
		const int o = cell_origin(f, pos);
		if (o < 0) { return 0; }
		const int nc = 1 << D;
		for (int d = 0; d < D; ++d) {
			const int   row  = static_cast<int>(f->sys.rhs.size());
			const float term = cw * 2.0f / static_cast<float>(nc);
			for (int c = 0; c < nc; ++c) {
				int col = o;
				for (int a = 0; a < D; ++a) { col += f->stride[a] * ((c >> a) % 2); }
				const float sign = ((c >> d) % 2) ? +1.0f : -1.0f;
				f->sys.ent.push_back(Entry{row, col, sign * term});
			}
			f->sys.rhs.push_back(cw * grad[d]);
		}
		return 1;
	}
	if (kernel == 2) {
		// cpp:188-236: multilerp(pos - 0.5, extra_bound = 1); per axis one row holding
		// -w at idx and +w at idx+stride for every kept sample (duplicate columns are left
		// for the matrix builder to sum); rhs = (sum w) * g_d.
		float shifted[kMaxDim] = {0, 0, 0};
		for (int d = 0; d < D; ++d) { shifted[d] = pos[d] - 0.5f; }
		int   idx[16];
		float w[16];
		const int n = corner_weights(idx, w, D, f->size, f->stride, shifted, 1);
		if (n == 0) { return 0; }
		for (int d = 0; d < D; ++d) {
			const int row = static_cast<int>(f->sys.rhs.size());
			float sum = 0.0f;
			for (int k = 0; k < n; ++k) {
				const float c = w[k] * cw;
				f->sys.ent.push_back(Entry{row, idx[k], -c});
				f->sys.ent.push_back(Entry{row, idx[k] + f->stride[d], +c});
				sum += c;
			}
			f->sys.rhs.push_back(sum * grad[d]);
		}
		return 1;
	}
	return -1;
}

// field_interpolation.cpp:243-316  add_model_constraint: forward-anchored difference rows along
// axis d at lattice point `index`; a row exists only where the whole stencil fits.
static void model_rows_at(Field* f, const Weights& w, const int* coord, int index, int d)
{
	const int size = f->size[d];
	const int st   = f->stride[d];
	const int c    = coord[d];
	System*   s    = &f->sys;

	if (w.model_0 > 0 && 0 <= c && c < size) {  // cpp:257-263 (once per axis!)
		const int cols[1] = {index};
		const float k[1] = {1.0f};
		push_row(s, w.model_0, 0.0f, 1, cols, k);
	}
	if (w.model_1 > 0 && 0 <= c && c + 1 < size) {  // cpp:265-271
		const int cols[2] = {index, index + st};
		const float k[2] = {-1.0f, +1.0f};
		push_row(s, w.model_1, 0.0f, 2, cols, k);
	}
	if (w.model_2 > 0 && 0 <= c && c + 2 < size) {  // cpp:273-280
		const int cols[3] = {index, index + st, index + 2 * st};
		const float k[3] = {+1.0f, -2.0f, +1.0f};
		push_row(s, w.model_2, 0.0f, 3, cols, k);
	}
	if (w.model_3 > 0 && 0 <= c && c + 3 < size) {  // cpp:282-290
		const int cols[4] = {index, index + st, index + 2 * st, index + 3 * st};
		const float k[4] = {+1.0f, -3.0f, +3.0f, -1.0f};
		push_row(s, w.model_3, 0.0f, 4, cols, k);
	}
	if (w.model_4 > 0 && 0 <= c && c + 4 < size) {  // cpp:292-301
		const int cols[5] = {index, index + st, index + 2 * st, index + 3 * st, index + 4 * st};
		const float k[5] = {+1.0f, -4.0f, +6.0f, -4.0f, +1.0f};
		push_row(s, w.model_4, 0.0f, 5, cols, k);
	}
	if (w.gradient_smoothness > 0 && 0 <= c && c + 1 < size) {  // cpp:303-315
		for (int o = 0; o < f->ndim; ++o) {
			if (o == d) { continue; }
			if (coord[o] + 1 >= f->size[o]) { continue; }
			const int so = f->stride[o];
			const int cols[4] = {index, index + st, index + so, index + so + st};
			const float k[4] = {-1.0f, +1.0f, +1.0f, -1.0f};
			push_row(s, w.gradient_smoothness, 0.0f, 4, cols, k);
		}
	}
}

// field_interpolation.cpp:318-341  coordinate_from_index + add_field_constraints.
static void model_rows(Field* f, const Weights& w)
{
	long n = 1;
	for (int d = 0; d < f->ndim; ++d) { n *= f->size[d]; }
	for (int index = 0; index < n; ++index) {
		int coord[kMaxDim];
		int rest = index;
		for (int d = 0; d < f->ndim; ++d) {
			coord[d] = rest % f->size[d];
			rest /= f->size[d];
		}
		for (int d = 0; d < f->ndim; ++d) { model_rows_at(f, w, coord, index, d); }
	}
}

// field_interpolation.cpp:343-371  add_points.  Value target is always 0.  Returns -1 when the
// nearest-neighbour value kernel is asked for without normals (reference CHECK_NOTNULL_F, cpp:361).
static int point_rows(Field* f, float vw, int vkernel, float gw, int gkernel, int n,
                      const float* pos, const float* normals, const float* pw)
{
	const int D = f->ndim;
	for (int i = 0; i < n; ++i) {
		const float  w = pw ? pw[i] : 1.0f;
		const float* p = pos + static_cast<size_t>(i) * D;
		const float* g = normals ? normals + static_cast<size_t>(i) * D : nullptr;
		if (vkernel == 0) {
			if (!normals) { return -1; }
			value_row_nearest(f, p, g, 0.0f, w * vw);
		} else {
			value_row(f, p, 0.0f, w * vw);
		}
		if (normals) {
			if (gradient_rows(f, p, g, w * gw, gkernel) < 0) { return -1; }
		}
	}
	return 0;
}

// ---------------------------------------------------------------------------------------------
// field_interpolation.cpp:402-429  generate_error_map.
static void blame_map(size_t nent, const Entry* ent, size_t nrows, const float* rhs, size_t ncols,
                      const float* x, float* out)
{
	std::vector<float> err(rhs, rhs + nrows);
	std::vector<float> sq(nrows, 0.0f);
	for (size_t k = 0; k < nent; ++k) {
		err[ent[k].row] -= x[ent[k].col] * ent[k].val;
		sq[ent[k].row] += ent[k].val * ent[k].val;
	}
	for (auto& e : err) { e *= e; }
	std::fill(out, out + ncols, 0.0f);
	for (size_t k = 0; k < nent; ++k) {
		if (sq[ent[k].row] != 0) {
			const float frac = (ent[k].val * ent[k].val) / sq[ent[k].row];
			out[ent[k].col] += frac * err[ent[k].row];
		}
	}
}

// field_interpolation.cpp:431-485  upscale_field.
static void upscale(const float* small, int ndim, const int* ssz, const int* lsz, float* out)
{
	int sstride[kMaxDim];
	{
		int s = 1;
		for (int d = 0; d < ndim; ++d) { sstride[d] = s; s *= ssz[d]; }
	}
	long nl = 1;
	for (int d = 0; d < ndim; ++d) { nl *= lsz[d]; }
	for (long li = 0; li < nl; ++li) {
		long  rest = li;
		float sp[kMaxDim];
		for (int d = 0; d < ndim; ++d) {
			const int c = static_cast<int>(rest % lsz[d]);
			rest /= lsz[d];
			sp[d] = static_cast<float>(c) * (static_cast<float>(ssz[d]) - 1.0f) /
			        (static_cast<float>(lsz[d]) - 1.0f);
		}
		int   idx[16];
		float w[16];
		const int n = corner_weights(idx, w, ndim, ssz, sstride, sp, 0);
		float wsum = 0, fsum = 0;
		for (int k = 0; k < n; ++k) {
			wsum += w[k];
			fsum += w[k] * small[idx[k]];
		}
		out[li] = (wsum == 0) ? 0.0f : fsum / wsum;
	}
}

// ---------------------------------------------------------------------------------------------
// Sparse matrices.  Eigen's default SparseMatrix is column-major; we keep compressed columns
// AND compressed rows of A because the product A^T*A walks both.

template <typename T>
struct Compressed {  // compressed along "outer": CSC when outer = column
	int              n_outer = 0, n_inner = 0;
	std::vector<int> ptr, idx;
	std::vector<T>   val;
};

// sparse_linear.cpp:59-70 / 72-93 + sparse_linear.hpp:43: triplets -> compressed matrix,
// duplicates summed in input order (Eigen setFromTriplets: insert in order, then collapse
// duplicates within each outer vector in stored order).  drop_zero mirrors :84.
// Returns false on an out-of-range index (reference CHECK_*_F aborts, :80-83).
template <typename T>
static bool compress(const std::vector<Entry>& ent, int nrows, int ncols, bool by_col, bool drop_zero,
                     Compressed<T>* out)
{
	const int n_outer = by_col ? ncols : nrows;
	const int n_inner = by_col ? nrows : ncols;
	std::vector<int> count(n_outer + 1, 0);
	for (const auto& e : ent) {
		if (e.row < 0 || e.col < 0 || e.row >= nrows || e.col >= ncols) { return false; }
		if (drop_zero && e.val == 0.0f) { continue; }
		count[(by_col ? e.col : e.row) + 1]++;
	}
	for (int i = 0; i < n_outer; ++i) { count[i + 1] += count[i]; }
	std::vector<int> pos(count.begin(), count.end() - 1);
	std::vector<int> inner(count[n_outer]);
	std::vector<T>   val(count[n_outer]);
	for (const auto& e : ent) {  // stable bucket by outer index
		if (drop_zero && e.val == 0.0f) { continue; }
		const int o = by_col ? e.col : e.row;
		inner[pos[o]] = by_col ? e.row : e.col;
		val[pos[o]]   = static_cast<T>(e.val);
		pos[o]++;
	}
	// within each outer vector: stable sort by inner index, then sum runs in order.
	out->n_outer = n_outer;
	out->n_inner = n_inner;
	out->ptr.assign(n_outer + 1, 0);
	out->idx.clear();
	out->val.clear();
	std::vector<int> order;
	for (int o = 0; o < n_outer; ++o) {
		const int b = count[o], e = count[o + 1];
		order.resize(e - b);
		for (int k = 0; k < e - b; ++k) { order[k] = b + k; }
		std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return inner[a] < inner[c]; });
		for (size_t k = 0; k < order.size();) {
			const int i = inner[order[k]];
			T         s = val[order[k]];
			size_t    m = k + 1;
			while (m < order.size() && inner[order[m]] == i) { s += val[order[m]]; ++m; }
			out->idx.push_back(i);
			out->val.push_back(s);
			k = m;
		}
		out->ptr[o + 1] = static_cast<int>(out->idx.size());
	}
	return true;
}

// sparse_linear.cpp:105-113  make_square: AtA = A^T * A as a column-compressed matrix.
// Column j of the product accumulates, over the rows k stored in column j of A (ascending),
// A(k,j) * (row k of A)^T -- the conservative sparse*sparse product order.
template <typename T>
static void normal_matrix(const Compressed<T>& Acsc, const Compressed<T>& Acsr, Compressed<T>* AtA)
{
	const int n = Acsc.n_outer;
	AtA->n_outer = AtA->n_inner = n;
	AtA->ptr.assign(n + 1, 0);
	AtA->idx.clear();
	AtA->val.clear();
	std::vector<T>   acc(n, T(0));
	std::vector<int> mark(n, -1);
	std::vector<int> touched;
	for (int j = 0; j < n; ++j) {
		touched.clear();
		for (int a = Acsc.ptr[j]; a < Acsc.ptr[j + 1]; ++a) {
			const int k   = Acsc.idx[a];
			const T   akj = Acsc.val[a];
			for (int b = Acsr.ptr[k]; b < Acsr.ptr[k + 1]; ++b) {
				const int i = Acsr.idx[b];
				if (mark[i] != j) { mark[i] = j; acc[i] = T(0); touched.push_back(i); }
				acc[i] += Acsr.val[b] * akj;
			}
		}
		std::sort(touched.begin(), touched.end());
		for (int i : touched) {
			AtA->idx.push_back(i);
			AtA->val.push_back(acc[i]);
		}
		AtA->ptr[j + 1] = static_cast<int>(AtA->idx.size());
	}
}

// A^T * b with A column-compressed (sparse_linear.cpp:120,159,196,225,411).
template <typename T>
static std::vector<T> transpose_times(const Compressed<T>& Acsc, const std::vector<T>& b)
{
	std::vector<T> y(Acsc.n_outer, T(0));
	for (int j = 0; j < Acsc.n_outer; ++j) {
		T s = 0;
		for (int a = Acsc.ptr[j]; a < Acsc.ptr[j + 1]; ++a) { s += Acsc.val[a] * b[Acsc.idx[a]]; }
		y[j] = s;
	}
	return y;
}

// y = M*x for a column-compressed square M (Eigen col-major sparse * dense: axpy per column).
template <typename T>
static void matvec(const Compressed<T>& M, const T* x, T* y)
{
	const int n = M.n_inner;
	for (int i = 0; i < n; ++i) { y[i] = 0; }
	for (int j = 0; j < M.n_outer; ++j) {
		const T xj = x[j];
		for (int a = M.ptr[j]; a < M.ptr[j + 1]; ++a) { y[M.idx[a]] += M.val[a] * xj; }
	}
}

template <typename T>
static std::vector<T> diagonal(const Compressed<T>& M)
{
	std::vector<T> d(M.n_outer, T(0));
	for (int j = 0; j < M.n_outer; ++j) {
		for (int a = M.ptr[j]; a < M.ptr[j + 1]; ++a) {
			if (M.idx[a] == j) { d[j] = M.val[a]; }
		}
	}
	return d;
}

template <typename T>
struct Normal {
	Compressed<T>  AtA;
	std::vector<T> Atb;
};

// The common prologue of every solver in sparse_linear.cpp (e.g. :194-196).
template <typename T>
static bool build_normal(const System& s, int ncols, bool drop_zero, Normal<T>* out)
{
	const int nrows = static_cast<int>(s.rhs.size());
	Compressed<T> Acsc, Acsr;
	if (!compress<T>(s.ent, nrows, ncols, true, drop_zero, &Acsc)) { return false; }
	if (!compress<T>(s.ent, nrows, ncols, false, drop_zero, &Acsr)) { return false; }
	normal_matrix(Acsc, Acsr, &out->AtA);
	std::vector<T> b(s.rhs.begin(), s.rhs.end());
	out->Atb = transpose_times(Acsc, b);
	return true;
}

// ---------------------------------------------------------------------------------------------
// Cholesky (stands in for Eigen::SimplicialLLT, sparse_linear.cpp:135,167,354).  Banded storage,
// no reordering: mathematically the same factorisation/solution, failure (non-positive pivot)
// is reported like solver.info() != Success.
template <typename T>
static bool cholesky_solve(const Compressed<T>& M, const std::vector<T>& b, std::vector<T>* x)
{
	const int n = M.n_outer;
	int bw = 0;
	for (int j = 0; j < n; ++j) {
		for (int a = M.ptr[j]; a < M.ptr[j + 1]; ++a) { bw = std::max(bw, std::abs(M.idx[a] - j)); }
	}
	const size_t w = static_cast<size_t>(bw) + 1;
	std::vector<T> L(static_cast<size_t>(n) * w, T(0));  // L[i][j] at L[i*w + (j-i+bw)], i-bw<=j<=i
	for (int j = 0; j < n; ++j) {
		for (int a = M.ptr[j]; a < M.ptr[j + 1]; ++a) {
			const int i = M.idx[a];
			if (i >= j) { L[static_cast<size_t>(i) * w + (j - i + bw)] = M.val[a]; }
		}
	}
	for (int i = 0; i < n; ++i) {
		T* Li = &L[static_cast<size_t>(i) * w];
		const int j0 = std::max(0, i - bw);
		for (int j = j0; j <= i; ++j) {
			const T* Lj = &L[static_cast<size_t>(j) * w];
			const int k0 = std::max(j0, j - bw);
			T s = Li[j - i + bw];
			// sum_k L[i][k]*L[j][k], k0 <= k < j
			const T* pi = Li + (k0 - i + bw);
			const T* pj = Lj + (k0 - j + bw);
			for (int k = 0; k < j - k0; ++k) { s -= pi[k] * pj[k]; }
			if (j < i) {
				Li[j - i + bw] = s / Lj[bw];
			} else {
				if (!(s > T(0))) { return false; }
				Li[bw] = std::sqrt(s);
			}
		}
	}
	std::vector<T> y(b);
	for (int i = 0; i < n; ++i) {  // forward
		const T* Li = &L[static_cast<size_t>(i) * w];
		T s = y[i];
		for (int j = std::max(0, i - bw); j < i; ++j) { s -= Li[j - i + bw] * y[j]; }
		y[i] = s / Li[bw];
	}
	for (int i = n - 1; i >= 0; --i) {  // backward with L^T
		T s = y[i];
		for (int j = i + 1; j <= std::min(n - 1, i + bw); ++j) {
			s -= L[static_cast<size_t>(j) * w + (i - j + bw)] * y[j];
		}
		y[i] = s / L[static_cast<size_t>(i) * w + bw];
	}
	for (int i = 0; i < n; ++i) {
		if (!std::isfinite(static_cast<double>(y[i]))) { return false; }
	}
	*x = y;
	return true;
}

// sparse_linear.cpp:115-152 (fast, float) and :154-184 (exact, double; zeros dropped :84).
// false <=> the reference returns {}.
static bool solve_direct(const System& s, int ncols, bool use_double, std::vector<float>* out)
{
	if (use_double) {
		Normal<double> ne;
		if (!build_normal<double>(s, ncols, true, &ne)) { return false; }
		std::vector<double> x;
		if (!cholesky_solve(ne.AtA, ne.Atb, &x)) { return false; }
		out->assign(x.begin(), x.end());
		return true;
	}
	Normal<float> ne;
	if (!build_normal<float>(s, ncols, false, &ne)) { return false; }
	std::vector<float> x;
	if (!cholesky_solve(ne.AtA, ne.Atb, &x)) { return false; }
	*out = x;
	return true;
}

// ---------------------------------------------------------------------------------------------
// Eigen::DiagonalPreconditioner: 1/diag, or 1 where the diagonal entry is zero/missing.
template <typename T>
static std::vector<T> jacobi_scaling(const Compressed<T>& M)
{
	std::vector<T> d = diagonal(M);
	for (auto& v : d) { v = (v != T(0)) ? T(1) / v : T(1); }
	return d;
}

template <typename T>
static T dot(const std::vector<T>& a, const std::vector<T>& b)
{
	T s = 0;
	for (size_t i = 0; i < a.size(); ++i) { s += a[i] * b[i]; }
	return s;
}

// Eigen::BiCGSTAB<SparseMatrix<float>> as used at sparse_linear.cpp:199-206 and :429-436:
// diagonal preconditioner, solveWithGuess, default max iterations 2*n, default tolerance
// epsilon, stop when ||r||^2 <= tol^2 * ||rhs||^2, restart when r becomes orthogonal to r0.
template <typename T>
static void bicgstab(const Compressed<T>& M, const std::vector<T>& rhs, std::vector<T>* xio, int max_it,
                     T tol, int* iters_out, T* err_out)
{
	const int n = M.n_outer;
	std::vector<T>& x = *xio;
	const std::vector<T> inv = jacobi_scaling(M);
	if (max_it <= 0) { max_it = 2 * n; }
	if (!(tol > 0)) { tol = std::numeric_limits<T>::epsilon(); }

	std::vector<T> r(n), r0(n), v(n, T(0)), p(n, T(0)), y(n), z(n), s(n), t(n), tmp(n);
	matvec(M, x.data(), tmp.data());
	for (int i = 0; i < n; ++i) { r[i] = rhs[i] - tmp[i]; }
	r0 = r;
	T r0_sq  = dot(r0, r0);
	const T rhs_sq = dot(rhs, rhs);
	if (rhs_sq == 0) {
		std::fill(x.begin(), x.end(), T(0));
		*iters_out = 0;
		*err_out   = 0;
		return;
	}
	T rho = 1, alpha = 1, w = 1;
	const T tol2 = tol * tol * rhs_sq;
	const T eps2 = std::numeric_limits<T>::epsilon() * std::numeric_limits<T>::epsilon();
	int it = 0, restarts = 0;
	while (dot(r, r) > tol2 && it < max_it) {
		const T rho_old = rho;
		rho = dot(r0, r);
		if (std::abs(rho) < eps2 * r0_sq) {
			matvec(M, x.data(), tmp.data());
			for (int i = 0; i < n; ++i) { r[i] = rhs[i] - tmp[i]; }
			r0  = r;
			rho = r0_sq = dot(r, r);
			if (restarts++ == 0) { it = 0; }
		}
		const T beta = (rho / rho_old) * (alpha / w);
		for (int i = 0; i < n; ++i) { p[i] = r[i] + beta * (p[i] - w * v[i]); }
		for (int i = 0; i < n; ++i) { y[i] = inv[i] * p[i]; }
		matvec(M, y.data(), v.data());
		alpha = rho / dot(r0, v);
		for (int i = 0; i < n; ++i) { s[i] = r[i] - alpha * v[i]; }
		for (int i = 0; i < n; ++i) { z[i] = inv[i] * s[i]; }
		matvec(M, z.data(), t.data());
		const T tt = dot(t, t);
		w = (tt > T(0)) ? dot(t, s) / tt : T(0);
		for (int i = 0; i < n; ++i) { x[i] += alpha * y[i] + w * z[i]; }
		for (int i = 0; i < n; ++i) { r[i] = s[i] - w * t[i]; }
		++it;
	}
	*iters_out = it;
	*err_out   = std::sqrt(dot(r, r) / rhs_sq);
}

// Jacobi-preconditioned conjugate gradients on AtA: the algorithm the GPU product runs (AtA is
// symmetric positive semi-definite, so CG applies; same stop rule as above).  Kept here so the
// GPU iterates can be compared step for step with a CPU run of the same recurrence.
template <typename T>
static void pcg(const Compressed<T>& M, const std::vector<T>& rhs, std::vector<T>* xio, int max_it, T tol,
                int* iters_out, T* err_out)
{
	const int n = M.n_outer;
	std::vector<T>& x = *xio;
	const std::vector<T> inv = jacobi_scaling(M);
	if (max_it <= 0) { max_it = 2 * n; }
	if (!(tol > 0)) { tol = std::numeric_limits<T>::epsilon(); }
	std::vector<T> r(n), p(n), q(n), z(n);
	matvec(M, x.data(), q.data());
	for (int i = 0; i < n; ++i) { r[i] = rhs[i] - q[i]; }
	const T rhs_sq = dot(rhs, rhs);
	if (rhs_sq == 0) {
		std::fill(x.begin(), x.end(), T(0));
		*iters_out = 0;
		*err_out   = 0;
		return;
	}
	const T tol2 = tol * tol * rhs_sq;
	for (int i = 0; i < n; ++i) { z[i] = inv[i] * r[i]; }
	p = z;
	T rz = dot(r, z);
	int it = 0;
	while (dot(r, r) > tol2 && it < max_it) {
		matvec(M, p.data(), q.data());
		const T pq = dot(p, q);
		if (!(pq > 0)) { break; }
		const T a = rz / pq;
		for (int i = 0; i < n; ++i) { x[i] += a * p[i]; r[i] -= a * q[i]; }
		for (int i = 0; i < n; ++i) { z[i] = inv[i] * r[i]; }
		const T rz_new = dot(r, z);
		const T b = rz_new / rz;
		rz = rz_new;
		for (int i = 0; i < n; ++i) { p[i] = z[i] + b * p[i]; }
		++it;
	}
	*iters_out = it;
	*err_out   = std::sqrt(dot(r, r) / rhs_sq);
}

// sparse_linear.cpp:214-241  jacobi_iterations: x_j <- w*(Atb - R x)_j / D_j + (1-w)*x_j with
// R = AtA - diag; `temp` is complete before x is touched (true Jacobi).  D_j == 0 is unguarded
// in the reference and is unguarded here.
static void jacobi_sweeps(const Compressed<float>& M, const std::vector<float>& Atb, std::vector<float>* xio,
                          int sweeps, float w)
{
	const int n = M.n_outer;
	std::vector<float>& x = *xio;
	const std::vector<float> D = diagonal(M);
	Compressed<float> R = M;
	for (int j = 0; j < n; ++j) {
		for (int a = R.ptr[j]; a < R.ptr[j + 1]; ++a) {
			if (R.idx[a] == j) { R.val[a] = 0.0f; }
		}
	}
	std::vector<float> tmp(n);
	for (int s = 0; s < sweeps; ++s) {
		matvec(R, x.data(), tmp.data());
		for (int j = 0; j < n; ++j) { tmp[j] = Atb[j] - tmp[j]; }
		for (int j = 0; j < n; ++j) { x[j] = w * tmp[j] / D[j] + (1.0f - w) * x[j]; }
	}
}

// sparse_linear.cpp:246-390  tile_solver_square: non-overlapping tile_size^D tiles, each solved
// exactly (float Cholesky) with couplings to other tiles moved to the rhs using the guess; 1e-6 on
// every tile diagonal; tiles holding only the regularisation are skipped; failed tiles keep the
// guess.  Off-tile entries of the symmetric matrix are visited once per stored entry and applied
// to BOTH ends (:332-333), exactly as written in the reference.
static std::vector<float> tile_pass(const Compressed<float>& M, const std::vector<float>& b,
                                    const std::vector<float>& guess, int ndim, const int* sizes, int ts,
                                    int* failures_out)
{
	int ntile[kMaxDim] = {1, 1, 1};
	int total = 1, per_tile = 1;
	for (int d = 0; d < ndim; ++d) {
		ntile[d] = (sizes[d] + ts - 1) / ts;
		total *= ntile[d];
		per_tile *= ts;
	}
	auto locate = [&](int full, int* tile, int* local) {
		int t = 0, l = 0, tstride = 1, lstride = 1;
		for (int d = 0; d < ndim; ++d) {
			const int c = full % sizes[d];
			full /= sizes[d];
			t += (c / ts) * tstride;
			l += (c % ts) * lstride;
			tstride *= ntile[d];
			lstride *= ts;
		}
		*tile  = t;
		*local = l;
	};
	struct Tile {
		std::vector<Entry> ent;
		std::vector<float> rhs;
	};
	std::vector<Tile> tiles(total);
	for (auto& t : tiles) {
		t.rhs.assign(per_tile, 0.0f);
		for (int i = 0; i < per_tile; ++i) { t.ent.push_back(Entry{i, i, 1e-6f}); }
	}
	for (int i = 0; i < static_cast<int>(b.size()); ++i) {
		int t, l;
		locate(i, &t, &l);
		tiles[t].rhs[l] = b[i];
	}
	for (int j = 0; j < M.n_outer; ++j) {
		for (int a = M.ptr[j]; a < M.ptr[j + 1]; ++a) {
			const int i = M.idx[a];
			int rt, rl, ct, cl;
			locate(i, &rt, &rl);
			locate(j, &ct, &cl);
			if (rt == ct) {
				tiles[rt].ent.push_back(Entry{rl, cl, M.val[a]});
			} else {
				tiles[rt].rhs[rl] -= M.val[a] * guess[j];
				tiles[ct].rhs[cl] -= M.val[a] * guess[i];
			}
		}
	}
	std::vector<float> sol(guess);
	int failures = 0;
	for (int t = 0; t < total; ++t) {
		if (static_cast<int>(tiles[t].ent.size()) == per_tile) { continue; }
		Compressed<float> Mt;
		if (!compress<float>(tiles[t].ent, per_tile, per_tile, true, false, &Mt)) { ++failures; continue; }
		std::vector<float> xt;
		if (!cholesky_solve(Mt, tiles[t].rhs, &xt)) { ++failures; continue; }
		for (int l = 0; l < per_tile; ++l) {
			int tt = t, ll = l, stride = 1, full = 0;
			bool inside = true;
			for (int d = 0; d < ndim; ++d) {
				const int c = (tt % ntile[d]) * ts + (ll % ts);
				inside = inside && (c < sizes[d]);
				full += c * stride;
				tt /= ntile[d];
				ll /= ts;
				stride *= sizes[d];
			}
			if (inside) { sol[full] = xt[l]; }
		}
	}
	*failures_out = failures;
	return sol;
}

}  // namespace fio

// =============================================================================================
// C ABI for the test harness (ctypes).  Handles are fio::Field*.

using fio::Field;

extern "C" {

struct fio_weights {  // mirrors fio::Weights / field_interpolation.hpp:75-95
	float data_pos, data_gradient, model_0, model_1, model_2, model_3, model_4, gradient_smoothness;
	int   value_kernel, gradient_kernel;
};

struct fio_solve_options {  // sparse_linear.hpp:66-73
	int   tile, tile_size, cg, max_iterations;
	float error_tolerance;
};

static fio::Weights to_weights(const fio_weights* w)
{
	return fio::Weights{w->data_pos, w->data_gradient, w->model_0, w->model_1, w->model_2, w->model_3,
	                    w->model_4, w->gradient_smoothness, w->value_kernel, w->gradient_kernel};
}

void* fio_field_new(int ndim, const int* sizes)
{
	if (ndim < 0 || ndim > fio::kMaxDim) { return nullptr; }
	Field* f = new Field();
	int one[3] = {1, 1, 1};
	fio::field_init(f, ndim, ndim ? sizes : one);
	return f;
}
void fio_field_free(void* h) { delete static_cast<Field*>(h); }
long fio_num_rows(void* h) { return static_cast<long>(static_cast<Field*>(h)->sys.rhs.size()); }
long fio_num_triplets(void* h) { return static_cast<long>(static_cast<Field*>(h)->sys.ent.size()); }

void fio_get(void* h, int* rows, int* cols, float* vals, float* rhs)
{
	const Field* f = static_cast<Field*>(h);
	for (size_t k = 0; k < f->sys.ent.size(); ++k) {
		rows[k] = f->sys.ent[k].row;
		cols[k] = f->sys.ent[k].col;
		vals[k] = f->sys.ent[k].val;
	}
	std::copy(f->sys.rhs.begin(), f->sys.rhs.end(), rhs);
}

// raw access like the apps use (bipolar_2d.cpp:251,260-261)
void fio_push_triplet(void* h, int row, int col, float v)
{
	static_cast<Field*>(h)->sys.ent.push_back(fio::Entry{row, col, v});
}
void fio_push_rhs(void* h, float v) { static_cast<Field*>(h)->sys.rhs.push_back(v); }

void fio_add_equation(void* h, float weight, float rhs, int n, const int* cols, const float* coef)
{
	fio::push_row(&static_cast<Field*>(h)->sys, weight, rhs, n, cols, coef);
}
int fio_add_value_constraint(void* h, const float* pos, float value, float w)
{
	return fio::value_row(static_cast<Field*>(h), pos, value, w) ? 1 : 0;
}
// harness convenience: add_value_constraint (cpp:57-80) for n points with per-point targets, in order.
int fio_add_value_constraints(void* h, int n, const float* pos, const float* values, const float* pw, float weight)
{
	Field* f = static_cast<Field*>(h);
	int accepted = 0;
	for (int i = 0; i < n; ++i) {
		const float w = pw ? pw[i] : 1.0f;
		accepted += fio::value_row(f, pos + static_cast<size_t>(i) * f->ndim, values[i], w * weight) ? 1 : 0;
	}
	return accepted;
}
int fio_add_value_constraint_nearest_neighbor(void* h, const float* pos, const float* grad, float value, float w)
{
	return fio::value_row_nearest(static_cast<Field*>(h), pos, grad, value, w) ? 1 : 0;
}
int fio_add_gradient_constraint(void* h, const float* pos, const float* grad, float w, int kernel)
{
	return fio::gradient_rows(static_cast<Field*>(h), pos, grad, w, kernel);
}
void fio_add_field_constraints(void* h, const fio_weights* w)
{
	fio::model_rows(static_cast<Field*>(h), to_weights(w));
}
int fio_add_points(void* h, float vw, int vk, float gw, int gk, int n, const float* pos, const float* normals,
                   const float* pw)
{
	return fio::point_rows(static_cast<Field*>(h), vw, vk, gw, gk, n, pos, normals, pw);
}
// field_interpolation.cpp:373-400  sdf_from_points: model rows first, then the point rows.
void* fio_sdf_from_points(int ndim, const int* sizes, const fio_weights* w, int n, const float* pos,
                          const float* normals, const float* pw)
{
	if (!pos) { return nullptr; }  // CHECK_NOTNULL_F, cpp:382
	Field* f = static_cast<Field*>(fio_field_new(ndim, sizes));
	if (!f) { return nullptr; }
	const fio::Weights ww = to_weights(w);
	fio::model_rows(f, ww);
	if (fio::point_rows(f, ww.data_pos, ww.value_kernel, ww.data_gradient, ww.gradient_kernel, n, pos, normals,
	                    pw) != 0) {
		delete f;
		return nullptr;
	}
	return f;
}

void fio_error_map(void* h, long ncols, const float* solution, float* out)
{
	const Field* f = static_cast<Field*>(h);
	fio::blame_map(f->sys.ent.size(), f->sys.ent.data(), f->sys.rhs.size(), f->sys.rhs.data(),
	               static_cast<size_t>(ncols), solution, out);
}
void fio_upscale_field(const float* small, int ndim, const int* small_sizes, const int* large_sizes, float* out)
{
	fio::upscale(small, ndim, small_sizes, large_sizes, out);
}

// ---- solvers: return 1 on success, 0 when the reference would return {} ----------------------

int fio_solve_exact(void* h, int ncols, float* out)
{
	std::vector<float> x;
	if (!fio::solve_direct(static_cast<Field*>(h)->sys, ncols, true, &x)) { return 0; }
	std::copy(x.begin(), x.end(), out);
	return 1;
}
int fio_solve_fast(void* h, int ncols, float* out)
{
	std::vector<float> x;
	if (!fio::solve_direct(static_cast<Field*>(h)->sys, ncols, false, &x)) { return 0; }
	std::copy(x.begin(), x.end(), out);
	return 1;
}
// float64 solution of the same normal equations, kept in double (golden-vector generation).
int fio_solve_exact_f64(void* h, int ncols, double* out)
{
	fio::Normal<double> ne;
	if (!fio::build_normal<double>(static_cast<Field*>(h)->sys, ncols, true, &ne)) { return 0; }
	std::vector<double> x;
	if (!fio::cholesky_solve(ne.AtA, ne.Atb, &x)) { return 0; }
	std::copy(x.begin(), x.end(), out);
	return 1;
}

// sparse_linear.cpp:186-212  solve_sparse_linear_with_guess
int fio_solve_with_guess(void* h, int ncols, const float* guess, int max_iterations, float tol, float* out,
                         int* iters, float* err)
{
	fio::Normal<float> ne;
	if (!fio::build_normal<float>(static_cast<Field*>(h)->sys, ncols, false, &ne)) { return 0; }
	std::vector<float> x(guess, guess + ncols);
	fio::bicgstab(ne.AtA, ne.Atb, &x, max_iterations, tol, iters, err);
	std::copy(x.begin(), x.end(), out);
	return 1;
}

// Jacobi-PCG on the explicit AtA (fp32 or fp64): same algorithm as the GPU solver.
int fio_solve_pcg(void* h, int ncols, const float* guess, int max_iterations, double tol, int use_double,
                  double* out, int* iters, double* err)
{
	if (use_double) {
		fio::Normal<double> ne;
		if (!fio::build_normal<double>(static_cast<Field*>(h)->sys, ncols, false, &ne)) { return 0; }
		std::vector<double> x(guess, guess + ncols);
		fio::pcg(ne.AtA, ne.Atb, &x, max_iterations, tol, iters, err);
		std::copy(x.begin(), x.end(), out);
	} else {
		fio::Normal<float> ne;
		if (!fio::build_normal<float>(static_cast<Field*>(h)->sys, ncols, false, &ne)) { return 0; }
		std::vector<float> x(guess, guess + ncols);
		float e = 0;
		fio::pcg(ne.AtA, ne.Atb, &x, max_iterations, static_cast<float>(tol), iters, &e);
		*err = e;
		for (int i = 0; i < ncols; ++i) { out[i] = x[i]; }
	}
	return 1;
}

// The fp64 Jacobi-PCG above on `threads` cores, for golden solutions at the benchmark's own size (tests/golden/
// make_golden_fullsize.py): the explicit AtA of the reference's rows (sparse_linear.cpp:105-113), fp64 throughout.  AtA is
// symmetric, so a stored column is also the row: y_j = sum_a val[a] x[idx[a]] is summed by ONE thread in stored order --
// the iterates do not depend on the thread count; dot products stay serial.  A guess in fp64 (a restart from a stored
// iterate) and a progress line every `report` iterations.  Test infrastructure only.
int fio_solve_pcg_f64_mt(void* h, int ncols, const double* guess, int max_iterations, double tol, int threads, int report,
                         double* out, int* iters, double* err)
{
	fio::Normal<double> ne;
	if (!fio::build_normal<double>(static_cast<Field*>(h)->sys, ncols, false, &ne)) { return 0; }
#ifdef _OPENMP
	if (threads > 0) { omp_set_num_threads(threads); }
#endif
	const fio::Compressed<double>& M = ne.AtA;
	const int n = ncols;
	auto apply = [&](const std::vector<double>& v, std::vector<double>& y) {
#pragma omp parallel for schedule(static)
		for (int j = 0; j < n; ++j) {
			double sum = 0;
			for (int a = M.ptr[j]; a < M.ptr[j + 1]; ++a) { sum += M.val[a] * v[M.idx[a]]; }
			y[j] = sum;
		}
	};
	auto dot = [&](const std::vector<double>& a, const std::vector<double>& b) {
		double sum = 0;
		for (int i = 0; i < n; ++i) { sum += a[i] * b[i]; }
		return sum;
	};
	const std::vector<double> inv = fio::jacobi_scaling(M);
	std::vector<double> x(guess, guess + n), r(n), p(n), q(n), z(n);
	if (max_iterations <= 0) { max_iterations = 2 * n; }
	apply(x, q);
	for (int i = 0; i < n; ++i) { r[i] = ne.Atb[i] - q[i]; }
	const double rhs_sq = dot(ne.Atb, ne.Atb);
	if (rhs_sq == 0) {
		std::fill(out, out + n, 0.0);
		*iters = 0;
		*err   = 0;
		return 1;
	}
	const double tol2 = tol * tol * rhs_sq;
	for (int i = 0; i < n; ++i) { z[i] = inv[i] * r[i]; }
	p = z;
	double rz = dot(r, z), rr = dot(r, r);
	int it = 0;
	while (rr > tol2 && it < max_iterations) {
		apply(p, q);
		const double pq = dot(p, q);
		if (!(pq > 0)) { break; }
		const double a = rz / pq;
#pragma omp parallel for schedule(static)
		for (int i = 0; i < n; ++i) {
			x[i] += a * p[i];
			r[i] -= a * q[i];
			z[i] = inv[i] * r[i];
		}
		const double rz_new = dot(r, z);
		const double b = rz_new / rz;
		rz = rz_new;
#pragma omp parallel for schedule(static)
		for (int i = 0; i < n; ++i) { p[i] = z[i] + b * p[i]; }
		rr = dot(r, r);
		++it;
		if (report > 0 && it % report == 0) {
			std::fprintf(stderr, "[fio_solve_pcg_f64_mt] %d iterations, relative residual %.3e\n", it, std::sqrt(rr / rhs_sq));
		}
	}
	std::copy(x.begin(), x.end(), out);
	*iters = it;
	*err   = std::sqrt(rr / rhs_sq);
	return 1;
}

// "Best-effort CPU" figure of SURVEY.md 8(d), NOT the reference's algorithm: Jacobi-PCG on the normal equations
// without ever forming AtA -- q = A^T (A p) from the compressed rows and columns of the reference's own A
// (sparse_linear.cpp:59-70 semantics: duplicates summed), fp32 storage, fp64 dot products, OpenMP over rows /
// columns / vector elements on `threads` cores.  Same stop rule as the GPU solver: ||r|| <= tol ||A^T b||.
// seconds[0] = building the compressed rows and columns, seconds[1] = the iteration.
int fio_solve_pcg_rows_omp(void* h, int ncols, const float* guess, int max_iterations, double tol, int threads,
                           float* out, int* iters, double* err, double* seconds)
{
	using clock = std::chrono::steady_clock;
	const fio::System& sys = static_cast<Field*>(h)->sys;
	const int nrows = static_cast<int>(sys.rhs.size());
	const int n = ncols;
#ifdef _OPENMP
	if (threads > 0) { omp_set_num_threads(threads); }
#endif
	const auto t0 = clock::now();
	fio::Compressed<float> Acsc, Acsr;
	if (!fio::compress<float>(sys.ent, nrows, ncols, true, false, &Acsc)) { return 0; }
	if (!fio::compress<float>(sys.ent, nrows, ncols, false, false, &Acsr)) { return 0; }
	std::vector<float> b(sys.rhs.begin(), sys.rhs.end());
	std::vector<float> atb(n), inv(n), x(guess, guess + n), r(n), p(n), q(n), t(nrows);
#pragma omp parallel for schedule(static)
	for (int j = 0; j < n; ++j) {
		double sb = 0, sd = 0;
		for (int a = Acsc.ptr[j]; a < Acsc.ptr[j + 1]; ++a) {
			sb += static_cast<double>(Acsc.val[a]) * b[Acsc.idx[a]];
			sd += static_cast<double>(Acsc.val[a]) * Acsc.val[a];
		}
		atb[j] = static_cast<float>(sb);
		inv[j] = sd != 0 ? static_cast<float>(1.0 / sd) : 1.0f;  // Eigen DiagonalPreconditioner: 1 where the diagonal is 0
	}
	const auto t1 = clock::now();
	auto apply = [&](const std::vector<float>& v, std::vector<float>& y) {
#pragma omp parallel for schedule(static)
		for (int i = 0; i < nrows; ++i) {
			float sum = 0;
			for (int a = Acsr.ptr[i]; a < Acsr.ptr[i + 1]; ++a) { sum += Acsr.val[a] * v[Acsr.idx[a]]; }
			t[i] = sum;
		}
#pragma omp parallel for schedule(static)
		for (int j = 0; j < n; ++j) {
			float sum = 0;
			for (int a = Acsc.ptr[j]; a < Acsc.ptr[j + 1]; ++a) { sum += Acsc.val[a] * t[Acsc.idx[a]]; }
			y[j] = sum;
		}
	};
	auto dot = [&](const std::vector<float>& u, const std::vector<float>& v) {
		double sum = 0;
#pragma omp parallel for reduction(+ : sum) schedule(static)
		for (int i = 0; i < n; ++i) { sum += static_cast<double>(u[i]) * v[i]; }
		return sum;
	};
	if (max_iterations <= 0) { max_iterations = 2 * n; }
	if (!(tol > 0)) { tol = std::numeric_limits<float>::epsilon(); }
	apply(x, q);
#pragma omp parallel for schedule(static)
	for (int i = 0; i < n; ++i) { r[i] = atb[i] - q[i]; }
	const double rhs_sq = dot(atb, atb);
	int it = 0;
	double rr = dot(r, r);
	if (rhs_sq == 0) {
		std::fill(x.begin(), x.end(), 0.0f);
		rr = 0;
	} else {
		const double tol2 = tol * tol * rhs_sq;
		double rz = 0;
#pragma omp parallel for reduction(+ : rz) schedule(static)
		for (int i = 0; i < n; ++i) {
			p[i] = inv[i] * r[i];
			rz += static_cast<double>(r[i]) * p[i];
		}
		while (rr > tol2 && it < max_iterations) {
			apply(p, q);
			const double pq = dot(p, q);
			if (!(pq > 0)) { break; }
			const float al = static_cast<float>(rz / pq);
			double rz_new = 0, rr_new = 0;
#pragma omp parallel for reduction(+ : rz_new, rr_new) schedule(static)
			for (int i = 0; i < n; ++i) {
				x[i] += al * p[i];
				r[i] -= al * q[i];
				rz_new += static_cast<double>(r[i]) * inv[i] * r[i];
				rr_new += static_cast<double>(r[i]) * r[i];
			}
			const float be = static_cast<float>(rz_new / rz);
			rz = rz_new;
			rr = rr_new;
#pragma omp parallel for schedule(static)
			for (int i = 0; i < n; ++i) { p[i] = inv[i] * r[i] + be * p[i]; }
			++it;
		}
	}
	const auto t2 = clock::now();
	std::copy(x.begin(), x.end(), out);
	*iters = it;
	*err   = rhs_sq > 0 ? std::sqrt(rr / rhs_sq) : 0.0;
	seconds[0] = std::chrono::duration<double>(t1 - t0).count();
	seconds[1] = std::chrono::duration<double>(t2 - t1).count();
	return 1;
}

// sparse_linear.cpp:214-241  jacobi_iterations
int fio_jacobi_iterations(void* h, int ncols, const float* guess, int sweeps, float w, float* out)
{
	if (sweeps <= 0) {
		std::copy(guess, guess + ncols, out);
		return 1;
	}
	fio::Normal<float> ne;
	if (!fio::build_normal<float>(static_cast<Field*>(h)->sys, ncols, false, &ne)) { return 0; }
	std::vector<float> x(guess, guess + ncols);
	fio::jacobi_sweeps(ne.AtA, ne.Atb, &x, sweeps, w);
	std::copy(x.begin(), x.end(), out);
	return 1;
}

// sparse_linear.cpp:392-443  solve_tiled_with_guess (guess_len != prod(sizes) -> {} :402-405)
int fio_solve_tiled_with_guess(void* h, long guess_len, const float* guess, int ndim, const int* sizes,
                               const fio_solve_options* opt, float* out, int* iters, float* err)
{
	long n = 1;
	for (int d = 0; d < ndim; ++d) { n *= sizes[d]; }
	if (guess_len != n) { return 0; }
	fio::Normal<float> ne;
	if (!fio::build_normal<float>(static_cast<Field*>(h)->sys, static_cast<int>(n), false, &ne)) { return 0; }
	std::vector<float> x(guess, guess + n);
	*iters = 0;
	*err   = 0;
	if (opt->tile) {
		if (opt->tile_size < 2) { return 0; }  // CHECK_GE_F(tile_size, 2), :254
		int failures = 0;
		x = fio::tile_pass(ne.AtA, ne.Atb, x, ndim, sizes, opt->tile_size, &failures);
	}
	if (opt->cg) { fio::bicgstab(ne.AtA, ne.Atb, &x, opt->max_iterations, opt->error_tolerance, iters, err); }
	std::copy(x.begin(), x.end(), out);
	return 1;
}

// ---- operator extraction for the operator-parity tests (float64, duplicates summed) ----------
// Two-call pattern: first with ptr/idx/val == NULL to get nnz.
long fio_normal_equations_f64(void* h, int ncols, int* colptr, int* rowidx, double* val, double* Atb,
                              double* diag)
{
	fio::Normal<double> ne;
	if (!fio::build_normal<double>(static_cast<Field*>(h)->sys, ncols, false, &ne)) { return -1; }
	const long nnz = static_cast<long>(ne.AtA.idx.size());
	if (colptr) { std::copy(ne.AtA.ptr.begin(), ne.AtA.ptr.end(), colptr); }
	if (rowidx) { std::copy(ne.AtA.idx.begin(), ne.AtA.idx.end(), rowidx); }
	if (val) { std::copy(ne.AtA.val.begin(), ne.AtA.val.end(), val); }
	if (Atb) { std::copy(ne.Atb.begin(), ne.Atb.end(), Atb); }
	if (diag) {
		const std::vector<double> d = fio::diagonal(ne.AtA);
		std::copy(d.begin(), d.end(), diag);
	}
	return nnz;
}

// y = AtA*x in float64 without forming AtA: y = A^T (A x).  Used at sizes where AtA is too big.
int fio_apply_normal_f64(void* h, int ncols, const double* x, double* y)
{
	const Field* f = static_cast<Field*>(h);
	std::vector<double> t(f->sys.rhs.size(), 0.0);
	for (const auto& e : f->sys.ent) {
		if (e.col < 0 || e.col >= ncols || e.row < 0 || e.row >= static_cast<int>(t.size())) { return 0; }
		t[e.row] += static_cast<double>(e.val) * x[e.col];
	}
	std::fill(y, y + ncols, 0.0);
	for (const auto& e : f->sys.ent) { y[e.col] += static_cast<double>(e.val) * t[e.row]; }
	return 1;
}

// y = A^T b in float64 from the rows themselves (sparse_linear.cpp:120 without the compressed matrix).
int fio_apply_transpose_rhs_f64(void* h, int ncols, double* y)
{
	const Field* f = static_cast<Field*>(h);
	const int nrows = static_cast<int>(f->sys.rhs.size());
	std::fill(y, y + ncols, 0.0);
	for (const auto& e : f->sys.ent) {
		if (e.col < 0 || e.col >= ncols || e.row < 0 || e.row >= nrows) { return 0; }
		y[e.col] += static_cast<double>(e.val) * static_cast<double>(f->sys.rhs[e.row]);
	}
	return 1;
}

}  // extern "C"
