// fi_generic.hip -- arbitrary sparse rows (the generic `LinearEquation` path) on the GPU.
//
// Reference path replaced: callers that hand a hand-built `LinearEquation` (sparse_linear.hpp:18-22) to the
// solvers -- src/bipolar_2d.cpp:177-302 (torus wrap, mip pyramid), src/line_2d.cpp:49-104 (interleaved xy
// unknowns) -- and the one lattice kernel whose rows are not cell-local:
// GradientKernel::kLinearInterpolation (field_interpolation.cpp:188-236: every row spans 3 lattice points
// along its axis).  The reference turns the triplets into an Eigen CSC matrix (duplicates summed,
// sparse_linear.hpp:43) and squares it (make_square, sparse_linear.cpp:105-113).  Here A^T A is never formed:
//
//   assemble   radix sort of the triplets by (row, col), duplicates summed in input order (stable sort;
//              fp32 like as_sparse_matrix_float), CSR of A; a second sort by (col, row) gives CSC for the
//              transposed product, so that  y = A^T (A x)  needs no atomics and is bitwise reproducible.
//              A^T b and diag(A^T A) are accumulated per column in fp64.
//   apply      k_generic_Ax   one thread per row     t = A x
//              k_generic_Aty  one thread per column  y[col] += sum_i A[i,col] t[i]   (+ x.y partials)
// Rows are short (<= 16 entries on lattice problems), so thread-per-row is adequate; traffic per apply:
// 2 * nnz * (4 + sizeof(T)) bytes + gathers.  Single-GPU contexts only (columns are global unknowns).

#include "fi_prim.h"

#include "fi_internal.h"

namespace fi {

namespace {

constexpr int kThreads = 256;

inline int blocks_for(int64_t n) { return static_cast<int>((n + kThreads - 1) / kThreads); }
inline int capped_blocks(int64_t n)
{
	const int64_t b = (n + kThreads - 1) / kThreads;
	return static_cast<int>(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

__device__ inline double wave_sum(double v)
{
	for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o, 64); }
	return v;
}
__device__ inline double block_sum(double v)
{
	__shared__ double s[kThreads / 64];
	v = wave_sum(v);
	if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6] = v; }
	__syncthreads();
	double r = 0;
	if (threadIdx.x == 0) {
		for (int w = 0; w < kThreads / 64; ++w) { r += s[w]; }
	}
	__syncthreads();
	return r;
}

// AoS fi_triplet (12 bytes, sparse_linear.hpp:8-15) -> sort key (row_offset + row) << 32 | col, value
__global__ __launch_bounds__(kThreads) void k_split_triplets(int64_t n, const fi_triplet* __restrict__ t,
                                                              uint32_t row_offset, uint64_t* __restrict__ key,
                                                              float* __restrict__ val, int64_t nrows, int64_t ncols,
                                                              int* __restrict__ bad)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (i >= n) { return; }
	const fi_triplet e = t[i];
	if (e.row < 0 || e.row >= nrows || e.col < 0 || e.col >= ncols) {  // reference CHECK_*_F, sparse_linear.cpp:80-83
		*bad = 1;
		key[i] = 0;
		val[i] = 0.0f;
		return;
	}
	key[i] = (static_cast<uint64_t>(row_offset + static_cast<uint32_t>(e.row)) << 32) | static_cast<uint32_t>(e.col);
	val[i] = e.value;
}

// GradientKernel::kLinearInterpolation (field_interpolation.cpp:188-236): multilerp(pos - 0.5, extra_bound 1);
// per axis d one row with -w at idx and +w at idx + stride_d for every kept sample, rhs (sum w) * g_d.
// A point whose samples are all dropped contributes empty rows (the reference adds none: same A^T A, A^T b).
template <int D>
__global__ __launch_bounds__(kThreads) void k_emit_gradient_linear(Geom g, long n, const float* __restrict__ pos,
                                                                    const float* __restrict__ nrm,
                                                                    const float* __restrict__ pw, float gw,
                                                                    float pos_scale, float nrm_scale,
                                                                    uint32_t row_offset, uint64_t* __restrict__ key,
                                                                    float* __restrict__ val, float* __restrict__ rhs)
{
	constexpr int NC = 1 << D;
	const long i = static_cast<long>(blockIdx.x) * kThreads + threadIdx.x;
	if (i >= n) { return; }
	const float w  = pw ? pw[i] : 1.0f;
	const float cw = w * gw;
	int   base[D];
	float t[D];
	bool  finite = true;
	for (int d = 0; d < D; ++d) {
		const float p  = (g.pshift[d] != 0.0f ? pos[i * D + d] * pos_scale + g.pshift[d] : pos[i * D + d] * pos_scale) - 0.5f;
		finite = finite && isfinite(p);
		const float fl = floorf(p);
		const bool  in = fl >= -1.0f && fl <= static_cast<float>(g.gn[d]);
		base[d] = in ? static_cast<int>(fl) : -4;  // far outside: every sample is dropped
		t[d]    = p - static_cast<float>(base[d]);
	}
	int     idx[NC];
	float   lw[NC];
	int     kept = 0;
	// Slabs: a point's rows reach the planes base .. base + 2 of the
	// slowest axis.  The rank keeps them whole if one of those planes is its own -- the others are then ghost planes (the
	// stencil's reach is at least 2) -- and not at all otherwise: every rank applies a row to its own columns only, like a
	// data cell.  Which samples exist is decided on GLOBAL coordinates, the same on every rank.
	// (The columns stay GLOBAL here -- the ghost planes, and with them the local numbering, are fixed by fi_assemble, which
	// shifts the sorted copy: generic_assemble_t.)
	constexpr int L = D - 1;
	bool mine = true;
	if (g.nown != g.nloc) {
		const int lo = g.off[L] + g.own_lo[L], hi = g.off[L] + g.own_hi[L];
		mine = base[L] + 2 >= lo && base[L] < hi;
	}
	for (int q = 0; q < NC; ++q) {
		int64_t ix = 0;
		float   ww = 1.0f;
		bool    in = finite && cw != 0.0f && mine;
		for (int d = 0; d < D; ++d) {
			const int up = (q >> d) & 1;
			const int cc = base[d] + up;
			ix += g.stride[d] * cc;
			ww *= up ? t[d] : 1.0f - t[d];
			in = in && (0 <= cc) && (cc + 1 < g.gn[d]);  // extra_bound = 1
		}
		if (in) {
			idx[kept] = static_cast<int>(ix);
			lw[kept]  = ww;
			++kept;
		}
	}
	for (int d = 0; d < D; ++d) {
		const uint32_t row = row_offset + static_cast<uint32_t>(i * D + d);
		const long     o   = (i * D + d) * (2 * NC);
		float sum = 0.0f;
		for (int k = 0; k < NC; ++k) {
			float    c   = 0.0f;
			uint32_t c0 = 0, c1 = 0;
			if (k < kept) {
				c  = lw[k] * cw;
				c0 = static_cast<uint32_t>(idx[k]);
				c1 = static_cast<uint32_t>(idx[k] + static_cast<int>(g.stride[d]));
				sum += c;
			}
			key[o + 2 * k]     = (static_cast<uint64_t>(row) << 32) | c0;
			val[o + 2 * k]     = -c;
			key[o + 2 * k + 1] = (static_cast<uint64_t>(row) << 32) | c1;
			val[o + 2 * k + 1] = c;
		}
		rhs[static_cast<long>(row)] = sum * (nrm[i * D + d] * nrm_scale);
	}
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_widen(int64_t n, const float* __restrict__ in, T* __restrict__ out)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (i < n) { out[i] = static_cast<T>(in[i]); }
}

__global__ __launch_bounds__(kThreads) void k_shift_cols(int64_t n, uint64_t* __restrict__ key, int64_t shift)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (i >= n) { return; }
	const uint64_t k = key[i];
	const int64_t  col = static_cast<int64_t>(k & 0xFFFFFFFFull) - shift;   // (empty rows carry column 0 with value 0: clamp)
	key[i] = (k & 0xFFFFFFFF00000000ull) | static_cast<uint64_t>(col < 0 ? 0 : col);
}

__global__ __launch_bounds__(kThreads) void k_unpack_csr(int64_t nnz, const uint64_t* __restrict__ key,
                                                          uint32_t* __restrict__ col, uint32_t* __restrict__ row_count)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (i >= nnz) { return; }
	col[i] = static_cast<uint32_t>(key[i] & 0xFFFFFFFFull);
	atomicAdd(&row_count[key[i] >> 32], 1u);
}

__global__ __launch_bounds__(kThreads) void k_transpose_keys(int64_t nnz, const uint64_t* __restrict__ key,
                                                              uint64_t* __restrict__ tkey)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (i >= nnz) { return; }
	tkey[i] = (key[i] << 32) | (key[i] >> 32);
}

__global__ __launch_bounds__(kThreads) void k_unpack_csc(int64_t nnz, const uint64_t* __restrict__ tkey,
                                                          uint32_t* __restrict__ row, uint32_t* __restrict__ colkey)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (i >= nnz) { return; }
	row[i]    = static_cast<uint32_t>(tkey[i] & 0xFFFFFFFFull);
	colkey[i] = static_cast<uint32_t>(tkey[i] >> 32);
}

// A^T b and diag(A^T A), one thread per column that holds entries, fp64 accumulation, added to the lattice arrays
template <typename T>
__global__ __launch_bounds__(kThreads) void k_generic_rhs_diag(int64_t ncols, const uint32_t* __restrict__ cols,
                                                                const uint32_t* __restrict__ ptr,
                                                                const uint32_t* __restrict__ row,
                                                                const T* __restrict__ val,
                                                                const float* __restrict__ rhs, T* __restrict__ atb,
                                                                T* __restrict__ diag, int64_t own_first, int64_t nown)
{
	const int64_t c = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (c >= ncols) { return; }
	if (static_cast<uint64_t>(static_cast<int64_t>(cols[c]) - own_first) >= static_cast<uint64_t>(nown)) { return; }  // a ghost column: the neighbour's
	double b = 0, d = 0;
	for (uint32_t k = ptr[c]; k < ptr[c + 1]; ++k) {
		const double a = static_cast<double>(val[k]);
		b += a * static_cast<double>(rhs[row[k]]);
		d += a * a;
	}
	atb[cols[c]] += static_cast<T>(b);
	diag[cols[c]] += static_cast<T>(d);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_generic_Ax(int64_t nrows, const uint32_t* __restrict__ ptr,
                                                          const uint32_t* __restrict__ col,
                                                          const T* __restrict__ val, const T* __restrict__ x,
                                                          T* __restrict__ t, const int* __restrict__ done)
{
	if (done && *done) { return; }
	for (int64_t r = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; r < nrows;
	     r += static_cast<int64_t>(gridDim.x) * kThreads) {
		T s = 0;
		for (uint32_t k = ptr[r]; k < ptr[r + 1]; ++k) { s += val[k] * x[col[k]]; }
		t[r] = s;
	}
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_generic_Aty(int64_t ncols, const uint32_t* __restrict__ cols,
                                                           const uint32_t* __restrict__ ptr,
                                                           const uint32_t* __restrict__ row,
                                                           const T* __restrict__ val, const T* __restrict__ t,
                                                           const T* __restrict__ x, T* __restrict__ y,
                                                           double* __restrict__ partial, const int* __restrict__ done,
                                                           int64_t own_first, int64_t nown)
{
	if (done && *done) { return; }
	double contrib = 0;
	for (int64_t c = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; c < ncols;
	     c += static_cast<int64_t>(gridDim.x) * kThreads) {
		if (static_cast<uint64_t>(static_cast<int64_t>(cols[c]) - own_first) >= static_cast<uint64_t>(nown)) { continue; }  // ghost column
		T s = 0;
		for (uint32_t k = ptr[c]; k < ptr[c + 1]; ++k) { s += val[k] * t[row[k]]; }
		const uint32_t j = cols[c];
		y[j] += s;
		contrib += static_cast<double>(x[j]) * static_cast<double>(s);
	}
	if (partial) {
		const double s = block_sum(contrib);
		if (threadIdx.x == 0) { partial[blockIdx.x] = s; }
	}
}

void grow(DevBuf& b, size_t used_bytes, size_t need_bytes, hipStream_t st)
{
	if (need_bytes <= b.bytes) { return; }
	size_t cap = b.bytes ? b.bytes : 4096;
	while (cap < need_bytes) { cap *= 2; }
	DevBuf n;
	n.alloc(cap);
	if (used_bytes) { FI_HIP_TRY(hipMemcpyAsync(n.p, b.p, used_bytes, hipMemcpyDeviceToDevice, st)); }
	FI_HIP_TRY(hipStreamSynchronize(st));
	b.swap(n);
}

}  // namespace

void generic_reserve(fi_ctx* c, int64_t more_trip, int64_t more_rows)
{
	GenericRows& G = c->generic;
	grow(G.key, sizeof(uint64_t) * G.ntrip, sizeof(uint64_t) * (G.ntrip + more_trip), c->stream);
	grow(G.val, sizeof(float) * G.ntrip, sizeof(float) * (G.ntrip + more_trip), c->stream);
	grow(G.rhs, sizeof(float) * G.nrows, sizeof(float) * (G.nrows + more_rows), c->stream);
}

void generic_add_coo(fi_ctx* c, int64_t nrows, int64_t ntrip, const fi_triplet* trip, const float* rhs, int memory)
{
	GenericRows& G = c->generic;
	FI_REQUIRE(G.nrows + nrows < (1LL << 31) && G.ntrip + ntrip < (1LL << 31), FI_ERR_UNSUPPORTED, "too many generic rows");
	generic_reserve(c, ntrip, nrows);
	hipStream_t st = c->stream;
	DevBuf dtrip, dbad;
	const fi_triplet* src = trip;
	if (memory == FI_HOST && ntrip > 0) {
		dtrip.alloc(sizeof(fi_triplet) * ntrip);
		FI_HIP_TRY(hipMemcpyAsync(dtrip.p, trip, sizeof(fi_triplet) * ntrip, hipMemcpyHostToDevice, st));
		src = dtrip.as<fi_triplet>();
	}
	dbad.alloc(sizeof(int));
	FI_HIP_TRY(hipMemsetAsync(dbad.p, 0, sizeof(int), st));
	if (ntrip > 0) {
		hipLaunchKernelGGL(k_split_triplets, dim3(blocks_for(ntrip)), dim3(kThreads), 0, st, ntrip, src,
		                   static_cast<uint32_t>(G.nrows), G.key.as<uint64_t>() + G.ntrip, G.val.as<float>() + G.ntrip,
		                   nrows, c->g.nown, dbad.as<int>());
	}
	if (nrows > 0) {
		FI_HIP_TRY(hipMemcpyAsync(G.rhs.as<float>() + G.nrows, rhs, sizeof(float) * nrows,
		                          memory == FI_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, st));
	}
	int bad = 0;
	FI_HIP_TRY(hipMemcpyAsync(&bad, dbad.p, sizeof(int), hipMemcpyDeviceToHost, st));
	FI_HIP_TRY(hipStreamSynchronize(st));
	FI_REQUIRE(bad == 0, FI_ERR_INVALID, "triplet index out of range (rows %lld, columns %lld)",
	           static_cast<long long>(nrows), static_cast<long long>(c->g.nown));
	G.ntrip += ntrip;
	G.nrows += nrows;
}

void generic_add_gradient_linear(fi_ctx* c, long n, const float* pos, const float* nrm, const float* pw, float gw,
                                 float pos_scale, float nrm_scale)
{
	GenericRows& G = c->generic;
	const int    D  = c->g.ndim;
	const int64_t rows = static_cast<int64_t>(n) * D, trips = rows * 2 * (1 << D);
	FI_REQUIRE(G.nrows + rows < (1LL << 31) && G.ntrip + trips < (1LL << 31), FI_ERR_UNSUPPORTED, "too many generic rows");
	generic_reserve(c, trips, rows);
	uint64_t* key = G.key.as<uint64_t>() + G.ntrip;
	float*    val = G.val.as<float>() + G.ntrip;
	const uint32_t off = static_cast<uint32_t>(G.nrows);
	switch (D) {
	case 1:
		hipLaunchKernelGGL(k_emit_gradient_linear<1>, dim3(blocks_for(n)), dim3(kThreads), 0, c->stream, c->g, n, pos, nrm,
		                   pw, gw, pos_scale, nrm_scale, off, key, val, G.rhs.as<float>());
		break;
	case 2:
		hipLaunchKernelGGL(k_emit_gradient_linear<2>, dim3(blocks_for(n)), dim3(kThreads), 0, c->stream, c->g, n, pos, nrm,
		                   pw, gw, pos_scale, nrm_scale, off, key, val, G.rhs.as<float>());
		break;
	default:
		hipLaunchKernelGGL(k_emit_gradient_linear<3>, dim3(blocks_for(n)), dim3(kThreads), 0, c->stream, c->g, n, pos, nrm,
		                   pw, gw, pos_scale, nrm_scale, off, key, val, G.rhs.as<float>());
		break;
	}
	FI_HIP_TRY(hipGetLastError());
	G.ntrip += trips;
	G.nrows += rows;
}

void generic_clear(fi_ctx* c)
{
	c->generic.ntrip = 0;
	c->generic.nrows = 0;
	c->generic.nnz   = 0;
	c->generic.ncols = 0;
}

template <typename T>
static void generic_assemble_t(fi_ctx* c)
{
	GenericRows& G = c->generic;
	G.nnz = G.ncols = 0;
	c->stats.num_generic_rows = G.nrows;
	if (G.ntrip == 0) { return; }
	hipStream_t st = c->stream;
	const int n = static_cast<int>(G.ntrip);
	DevBuf ksort, vsort, ukey, tmp, nruns, tkey, tkey_s, val_s, colkey, colcount;
	ksort.alloc(sizeof(uint64_t) * n);
	vsort.alloc(sizeof(float) * n);
	ukey.alloc(sizeof(uint64_t) * n);
	G.csr_val.alloc(sizeof(T) * n);
	nruns.alloc(sizeof(int) * 2);
	size_t tb = 0;
	FI_HIP_TRY(prim::sort_pairs_u64(nullptr, tb, G.key.as<uint64_t>(), ksort.as<uint64_t>(), G.val.as<float>(), vsort.as<float>(),
	                                 static_cast<size_t>(n), 0, 64, st));
	tmp.alloc(tb);
	FI_HIP_TRY(prim::sort_pairs_u64(tmp.p, tb, G.key.as<uint64_t>(), ksort.as<uint64_t>(), G.val.as<float>(), vsort.as<float>(),
	                                 static_cast<size_t>(n), 0, 64, st));
	// slabs: global -> local column numbers (the local array starts at global plane g.off[L]: a uniform shift, the order
	// stays).  Every kept row lies within the ghost planes: it touches an owned plane and reaches 2 planes (reach >= 2).
	const int64_t col_shift = static_cast<int64_t>(c->g.off[c->g.ndim - 1]) * c->g.stride[c->g.ndim - 1];
	if (col_shift != 0) {
		hipLaunchKernelGGL(k_shift_cols, dim3(blocks_for(n)), dim3(kThreads), 0, st, static_cast<int64_t>(n), ksort.as<uint64_t>(),
		                   col_shift);
	}
	// duplicates: summed in input order (stable sort) -- sparse_linear.hpp:43, Eigen setFromTriplets -- in the
	// context's precision: fp32 like as_sparse_matrix_float (:59-70), fp64 like as_sparse_matrix_double (:72-93)
	DevBuf vT;
	vT.alloc(sizeof(T) * n);
	hipLaunchKernelGGL((k_widen<T>), dim3(blocks_for(n)), dim3(kThreads), 0, st, static_cast<int64_t>(n), vsort.as<float>(),
	                   vT.as<T>());
	size_t tb2 = 0;
	FI_HIP_TRY(prim::sum_by_key(nullptr, tb2, ksort.as<uint64_t>(), ukey.as<uint64_t>(), vT.as<T>(), G.csr_val.as<T>(), nruns.as<int>(),
	                             static_cast<size_t>(n), st));
	DevBuf tmp2;
	tmp2.alloc(tb2);
	FI_HIP_TRY(prim::sum_by_key(tmp2.p, tb2, ksort.as<uint64_t>(), ukey.as<uint64_t>(), vT.as<T>(), G.csr_val.as<T>(), nruns.as<int>(),
	                             static_cast<size_t>(n), st));
	int nnz = 0;
	FI_HIP_TRY(hipMemcpyAsync(&nnz, nruns.p, sizeof(int), hipMemcpyDeviceToHost, st));
	FI_HIP_TRY(hipStreamSynchronize(st));
	G.nnz = nnz;
	// CSR
	G.csr_col.alloc(sizeof(uint32_t) * nnz);
	G.csr_ptr.alloc(sizeof(uint32_t) * (G.nrows + 2));
	DevBuf rowcount;
	rowcount.alloc(sizeof(uint32_t) * (G.nrows + 2));
	FI_HIP_TRY(hipMemsetAsync(rowcount.p, 0, sizeof(uint32_t) * (G.nrows + 2), st));
	hipLaunchKernelGGL(k_unpack_csr, dim3(blocks_for(nnz)), dim3(kThreads), 0, st, nnz, ukey.as<uint64_t>(),
	                   G.csr_col.as<uint32_t>(), rowcount.as<uint32_t>());
	size_t tb3 = 0;
	FI_HIP_TRY(prim::exclusive_sum(nullptr, tb3, rowcount.as<uint32_t>(), G.csr_ptr.as<uint32_t>(), static_cast<size_t>(G.nrows + 1), st));
	DevBuf tmp3;
	tmp3.alloc(tb3);
	FI_HIP_TRY(prim::exclusive_sum(tmp3.p, tb3, rowcount.as<uint32_t>(), G.csr_ptr.as<uint32_t>(), static_cast<size_t>(G.nrows + 1), st));
	// CSC: sort the unique entries by (col, row)
	tkey.alloc(sizeof(uint64_t) * nnz);
	tkey_s.alloc(sizeof(uint64_t) * nnz);
	G.csc_val.alloc(sizeof(T) * nnz);
	hipLaunchKernelGGL(k_transpose_keys, dim3(blocks_for(nnz)), dim3(kThreads), 0, st, nnz, ukey.as<uint64_t>(),
	                   tkey.as<uint64_t>());
	size_t tb4 = 0;
	FI_HIP_TRY(prim::sort_pairs_u64(nullptr, tb4, tkey.as<uint64_t>(), tkey_s.as<uint64_t>(), G.csr_val.as<T>(), G.csc_val.as<T>(),
	                                 static_cast<size_t>(nnz), 0, 64, st));
	DevBuf tmp4;
	tmp4.alloc(tb4);
	FI_HIP_TRY(prim::sort_pairs_u64(tmp4.p, tb4, tkey.as<uint64_t>(), tkey_s.as<uint64_t>(), G.csr_val.as<T>(), G.csc_val.as<T>(),
	                                 static_cast<size_t>(nnz), 0, 64, st));
	G.csc_row.alloc(sizeof(uint32_t) * nnz);
	colkey.alloc(sizeof(uint32_t) * nnz);
	hipLaunchKernelGGL(k_unpack_csc, dim3(blocks_for(nnz)), dim3(kThreads), 0, st, nnz, tkey_s.as<uint64_t>(),
	                   G.csc_row.as<uint32_t>(), colkey.as<uint32_t>());
	G.csc_cols.alloc(sizeof(uint32_t) * nnz);
	colcount.alloc(sizeof(uint32_t) * (nnz + 1));
	size_t tb5 = 0;
	FI_HIP_TRY(prim::run_length_encode(nullptr, tb5, colkey.as<uint32_t>(), G.csc_cols.as<uint32_t>(), colcount.as<uint32_t>(), nruns.as<int>(),
	                                    static_cast<size_t>(nnz), st));
	DevBuf tmp5;
	tmp5.alloc(tb5);
	FI_HIP_TRY(prim::run_length_encode(tmp5.p, tb5, colkey.as<uint32_t>(), G.csc_cols.as<uint32_t>(), colcount.as<uint32_t>(), nruns.as<int>(),
	                                    static_cast<size_t>(nnz), st));
	int ncols = 0;
	FI_HIP_TRY(hipMemcpyAsync(&ncols, nruns.p, sizeof(int), hipMemcpyDeviceToHost, st));
	FI_HIP_TRY(hipStreamSynchronize(st));
	G.ncols = ncols;
	G.csc_ptr.alloc(sizeof(uint32_t) * (ncols + 2));
	FI_HIP_TRY(hipMemsetAsync(colcount.as<uint32_t>() + ncols, 0, sizeof(uint32_t), st));
	size_t tb6 = 0;
	FI_HIP_TRY(prim::exclusive_sum(nullptr, tb6, colcount.as<uint32_t>(), G.csc_ptr.as<uint32_t>(), static_cast<size_t>(ncols + 1), st));
	DevBuf tmp6;
	tmp6.alloc(tb6);
	FI_HIP_TRY(prim::exclusive_sum(tmp6.p, tb6, colcount.as<uint32_t>(), G.csc_ptr.as<uint32_t>(), static_cast<size_t>(ncols + 1), st));
	G.t.alloc(elem_size(c) * (G.nrows + 1));
	hipLaunchKernelGGL((k_generic_rhs_diag<T>), dim3(blocks_for(ncols)), dim3(kThreads), 0, st, ncols,
	                   G.csc_cols.as<uint32_t>(), G.csc_ptr.as<uint32_t>(), G.csc_row.as<uint32_t>(), G.csc_val.as<T>(),
	                   G.rhs.as<float>(), c->atb.as<T>(), c->diag.as<T>(), c->g.own_first, c->g.nown);
	FI_HIP_TRY(hipGetLastError());
	FI_HIP_TRY(hipStreamSynchronize(st));
}

void generic_assemble(fi_ctx* c)
{
	// (slabs: only rows emitted from points -- GradientKernel::kLinearInterpolation, local columns -- get here:
	// fi_add_rows_coo refuses slab contexts)
	FI_REQUIRE(c->nranks == 1 || c->generic.ntrip == 0 || c->reach >= 2, FI_ERR_UNSUPPORTED,
	           "GradientKernel::kLinearInterpolation over slabs needs two ghost planes (model_2 or wider)");
	c->dtype == FI_F64 ? generic_assemble_t<double>(c) : generic_assemble_t<float>(c);
}

int generic_num_partials(const fi_ctx* c) { return c->generic.nnz > 0 ? capped_blocks(c->generic.ncols) : 0; }

// The tile operator of tile_solver_square (sparse_linear.cpp:246-390) for materialised rows: entry (i, j) of A^T A
// is kept only when unknowns i and j lie in the same ts^D tile of the lattice, i.e. every row is split into its
// per-tile pieces:  y_i += sum_r a_ri * sum_{j in row r, tile(j) == tile(i)} a_rj x_j.  One thread per non-empty
// column i walks the column (CSC) and, for every row it meets, that row's entries (CSR): rows are short.  The
// pre-solver runs once per solve, not per iteration of the main CG.
__device__ inline uint32_t tile_of(const Geom& g, uint32_t j, int ts)
{
	uint32_t t = 0, mul = 1;
	for (int d = 0; d < g.ndim; ++d) {
		const uint32_t n = static_cast<uint32_t>(g.gn[d]);
		t += ((j % n) / static_cast<uint32_t>(ts)) * mul;
		mul *= (n + static_cast<uint32_t>(ts) - 1) / static_cast<uint32_t>(ts);
		j /= n;
	}
	return t;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_generic_tile(Geom g, int ts, int64_t ncols, const uint32_t* __restrict__ cols,
                                                            const uint32_t* __restrict__ ptr,
                                                            const uint32_t* __restrict__ row,
                                                            const T* __restrict__ val,
                                                            const uint32_t* __restrict__ csr_ptr,
                                                            const uint32_t* __restrict__ csr_col,
                                                            const T* __restrict__ csr_val, const T* __restrict__ x,
                                                            T* __restrict__ y, double* __restrict__ partial,
                                                            const int* __restrict__ done)
{
	if (done && *done) { return; }
	double contrib = 0;
	for (int64_t c = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; c < ncols;
	     c += static_cast<int64_t>(gridDim.x) * kThreads) {
		const uint32_t j  = cols[c];
		const uint32_t tj = tile_of(g, j, ts);
		T s = 0;
		for (uint32_t k = ptr[c]; k < ptr[c + 1]; ++k) {
			const uint32_t r = row[k];
			T u = 0;
			for (uint32_t e = csr_ptr[r]; e < csr_ptr[r + 1]; ++e) {
				const uint32_t jj = csr_col[e];
				if (tile_of(g, jj, ts) == tj) { u += csr_val[e] * x[jj]; }
			}
			s += val[k] * u;
		}
		y[j] += s;
		contrib += static_cast<double>(x[j]) * static_cast<double>(s);
	}
	if (partial) {
		const double s = block_sum(contrib);
		if (threadIdx.x == 0) { partial[blockIdx.x] = s; }
	}
}

template <typename T>
static void generic_apply_tile_t(fi_ctx* c, const T* x, T* y, double* partial, int ts)
{
	GenericRows& G = c->generic;
	const int* done = c->scal.p ? &c->scal.as<CgScalars>()->done : nullptr;
	hipLaunchKernelGGL((k_generic_tile<T>), dim3(capped_blocks(G.ncols)), dim3(kThreads), 0, c->stream, c->g, ts, G.ncols,
	                   G.csc_cols.as<uint32_t>(), G.csc_ptr.as<uint32_t>(), G.csc_row.as<uint32_t>(), G.csc_val.as<T>(),
	                   G.csr_ptr.as<uint32_t>(), G.csr_col.as<uint32_t>(), G.csr_val.as<T>(), x, y, partial, done);
	FI_HIP_TRY(hipGetLastError());
}

template <typename T>
static void generic_apply_t(fi_ctx* c, const T* x, T* y, double* partial)
{
	GenericRows& G = c->generic;
	const int* done = c->scal.p ? &c->scal.as<CgScalars>()->done : nullptr;
	hipLaunchKernelGGL((k_generic_Ax<T>), dim3(capped_blocks(G.nrows)), dim3(kThreads), 0, c->stream, G.nrows,
	                   G.csr_ptr.as<uint32_t>(), G.csr_col.as<uint32_t>(), G.csr_val.as<T>(), x, G.t.as<T>(), done);
	hipLaunchKernelGGL((k_generic_Aty<T>), dim3(capped_blocks(G.ncols)), dim3(kThreads), 0, c->stream, G.ncols,
	                   G.csc_cols.as<uint32_t>(), G.csc_ptr.as<uint32_t>(), G.csc_row.as<uint32_t>(), G.csc_val.as<T>(),
	                   G.t.as<T>(), x, y, partial, done, c->g.own_first, c->g.nown);
	FI_HIP_TRY(hipGetLastError());
}

// generate_error_map over the raw triplets (duplicates NOT summed: the reference walks eq.triplets,
// field_interpolation.cpp:408-427).  Row residuals and |a|^2 are accumulated with fp64 atomics.
template <typename T>
__global__ __launch_bounds__(kThreads) void k_errmap_rows(int64_t ntrip, const uint64_t* __restrict__ key,
                                                           const float* __restrict__ val, const T* __restrict__ x,
                                                           double* __restrict__ res, double* __restrict__ sq, int64_t shift)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < ntrip;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		const uint32_t row = static_cast<uint32_t>(key[i] >> 32);
		const int64_t  col = static_cast<int64_t>(key[i] & 0xFFFFFFFFull) - shift;  // (slabs: the raw keys are global)
		const double a = static_cast<double>(val[i]);
		if (a == 0.0) { continue; }
		unsafeAtomicAdd(&res[row], -a * static_cast<double>(x[col]));
		unsafeAtomicAdd(&sq[row], a * a);
	}
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_errmap_blame(int64_t ntrip, const uint64_t* __restrict__ key,
                                                            const float* __restrict__ val, const double* __restrict__ res,
                                                            const double* __restrict__ sq, T* __restrict__ out, int64_t shift)
{
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x; i < ntrip;
	     i += static_cast<int64_t>(gridDim.x) * kThreads) {
		const uint32_t row = static_cast<uint32_t>(key[i] >> 32);
		const int64_t  col = static_cast<int64_t>(key[i] & 0xFFFFFFFFull) - shift;
		const double a = static_cast<double>(val[i]);
		if (a == 0.0 || !(sq[row] > 0.0)) { continue; }
		unsafeAtomicAdd(&out[col], static_cast<T>(a * a / sq[row] * res[row] * res[row]));
	}
}

__global__ __launch_bounds__(kThreads) void k_errmap_init(int64_t nrows, const float* __restrict__ rhs,
                                                           double* __restrict__ res, double* __restrict__ sq)
{
	const int64_t r = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (r < nrows) {
		res[r] = static_cast<double>(rhs[r]);
		sq[r]  = 0.0;
	}
}

template <typename T>
static void generic_error_map_t(fi_ctx* c, const T* x, T* out)
{
	GenericRows& G = c->generic;
	DevBuf &res = c->scratch[22], &sq = c->scratch[23];
	const int64_t shift = static_cast<int64_t>(c->g.off[c->g.ndim - 1]) * c->g.stride[c->g.ndim - 1];
	res.alloc(sizeof(double) * G.nrows);
	sq.alloc(sizeof(double) * G.nrows);
	hipLaunchKernelGGL(k_errmap_init, dim3(blocks_for(G.nrows)), dim3(kThreads), 0, c->stream, G.nrows, G.rhs.as<float>(),
	                   res.as<double>(), sq.as<double>());
	hipLaunchKernelGGL((k_errmap_rows<T>), dim3(capped_blocks(G.ntrip)), dim3(kThreads), 0, c->stream, G.ntrip,
	                   G.key.as<uint64_t>(), G.val.as<float>(), x, res.as<double>(), sq.as<double>(), shift);
	hipLaunchKernelGGL((k_errmap_blame<T>), dim3(capped_blocks(G.ntrip)), dim3(kThreads), 0, c->stream, G.ntrip,
	                   G.key.as<uint64_t>(), G.val.as<float>(), res.as<double>(), sq.as<double>(), out, shift);
	FI_HIP_TRY(hipGetLastError());
}

void generic_error_map(fi_ctx* c, const void* x, void* out)
{
	if (c->generic.ntrip == 0 || c->generic.nrows == 0) { return; }
	c->dtype == FI_F64 ? generic_error_map_t<double>(c, static_cast<const double*>(x), static_cast<double*>(out))
	                   : generic_error_map_t<float>(c, static_cast<const float*>(x), static_cast<float*>(out));
}

// tile_solver_square skips a tile that holds nothing but its 1e-6 diagonal (sparse_linear.cpp:343-346 in spirit:
// "ent.size() == per_tile"): its unknowns keep the guess.  With rows only (no model, no cells) a tile is empty iff
// none of its unknowns is a non-empty column.
__global__ __launch_bounds__(kThreads) void k_tile_flags(Geom g, int ts, int64_t ncols, const uint32_t* __restrict__ cols,
                                                          uint8_t* __restrict__ flag)
{
	const int64_t c = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (c < ncols) { flag[tile_of(g, cols[c], ts)] = 1; }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void k_keep_guess_in_empty_tiles(Geom g, int ts, int64_t n,
                                                                         const uint8_t* __restrict__ flag,
                                                                         const T* __restrict__ guess, T* __restrict__ x)
{
	const int64_t i = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (i < n && !flag[tile_of(g, static_cast<uint32_t>(i), ts)]) { x[i] = guess[i]; }
}

void generic_keep_guess_in_empty_tiles(fi_ctx* c, int ts, const void* guess, void* x)
{
	GenericRows& G = c->generic;
	int64_t ntiles = 1;
	for (int d = 0; d < c->g.ndim; ++d) { ntiles *= (c->g.gn[d] + ts - 1) / ts; }
	DevBuf& flag = c->scratch[25];
	flag.alloc(static_cast<size_t>(ntiles));
	FI_HIP_TRY(hipMemsetAsync(flag.p, 0, static_cast<size_t>(ntiles), c->stream));
	if (G.ncols > 0) {
		hipLaunchKernelGGL(k_tile_flags, dim3(blocks_for(G.ncols)), dim3(kThreads), 0, c->stream, c->g, ts, G.ncols,
		                   G.csc_cols.as<uint32_t>(), flag.as<uint8_t>());
	}
	const int64_t n = c->g.nloc;
	if (c->dtype == FI_F64) {
		hipLaunchKernelGGL((k_keep_guess_in_empty_tiles<double>), dim3(blocks_for(n)), dim3(kThreads), 0, c->stream, c->g, ts, n,
		                   flag.as<uint8_t>(), static_cast<const double*>(guess), static_cast<double*>(x));
	} else {
		hipLaunchKernelGGL((k_keep_guess_in_empty_tiles<float>), dim3(blocks_for(n)), dim3(kThreads), 0, c->stream, c->g, ts, n,
		                   flag.as<uint8_t>(), static_cast<const float*>(guess), static_cast<float*>(x));
	}
	FI_HIP_TRY(hipGetLastError());
}

void generic_apply_tile(fi_ctx* c, const void* x, void* y, double* partial, int ts)
{
	if (c->generic.nnz == 0) { return; }
	c->dtype == FI_F64 ? generic_apply_tile_t<double>(c, static_cast<const double*>(x), static_cast<double*>(y), partial, ts)
	                   : generic_apply_tile_t<float>(c, static_cast<const float*>(x), static_cast<float*>(y), partial, ts);
}

void generic_apply(fi_ctx* c, const void* x, void* y, double* partial)
{
	if (c->generic.nnz == 0) { return; }
	c->dtype == FI_F64 ? generic_apply_t<double>(c, static_cast<const double*>(x), static_cast<double*>(y), partial)
	                   : generic_apply_t<float>(c, static_cast<const float*>(x), static_cast<float*>(y), partial);
}

}  // namespace fi
