// fi_tail.hip -- the small-level engine (fi_tail.h): the coarse tail of a level hierarchy in ONE launch of ONE workgroup.
//
// Why: a V-cycle visits every level of a few thousand unknowns with 15-25 launches that each finish in 4-7 us -- the kernel
// boundary, not the work -- and configs 2 / 3 spend a fifth of their step in such launches (profiles/r4_by_grid_c{2,3}.md: 171
// and 773 launches per step on the coarsest level alone).  Here the stages of the cycle (smoother steps, residuals,
// restriction, interpolation) of ALL tail levels run inside one kernel of 1024 threads: five vectors per level in LDS, a
// workgroup barrier behind every stage, the restricted residual read from and the correction written to global memory once.
//
// Operator of a level: the model rows (model_0 / model_1 / model_2, field_interpolation.cpp:257-280) matrix-free from the
// global coordinates -- the same boundary rows as the tiled kernels -- and the data rows (cell blocks of fi_assembly.hip) as
// 3^D diagonals (`dia`, global memory): out-of-lattice corners of a cell carry zero coefficients, so the diagonals need no
// masks.  Reference role: the small exact solves of tile_solver_square (sparse_linear.cpp:246-390).
#include <map>
#include <mutex>

#include "fi_tail.h"

namespace fi {

namespace {

constexpr int kThreads = 256;

__host__ __device__ constexpr int ipow3(int d) { return d == 0 ? 1 : 3 * ipow3(d - 1); }

__host__ __device__ inline int packed_index(int i, int j, int nc)  // i <= j
{
	return i * nc - (i * (i - 1)) / 2 + (j - i);
}

__device__ inline float bf16(unsigned short v) { return __uint_as_float(static_cast<unsigned int>(v) << 16); }

// The five coefficients k[delta + 2], delta = -2 .. 2, of row `c` of the model operator along one axis of n points:
// w2^2 S2^T S2 + w1^2 S1^T S1 + w0^2, S2 rows [1, -2, 1] anchored at a (0 <= a, a + 2 < n), S1 rows [-1, +1] (a + 1 < n).
__device__ inline void axis_coefs(int c, int n, float w0sq, float w1sq, float w2sq, float* k)
{
	const float e0 = (c - 2 >= 0 && c < n) ? w2sq : 0.0f;          // row anchored at c - 2: c is its third point
	const float e1 = (c - 1 >= 0 && c + 1 < n) ? w2sq : 0.0f;      // ... at c - 1
	const float e2 = (c + 2 < n) ? w2sq : 0.0f;                    // ... at c
	const float f0 = (c - 1 >= 0) ? w1sq : 0.0f;                   // [-1, +1] anchored at c - 1
	const float f1 = (c + 1 < n) ? w1sq : 0.0f;                    // ... at c
	k[0] = e0;
	k[1] = -2.0f * e0 - 2.0f * e1 - f0;
	k[2] = e0 + 4.0f * e1 + e2 + f0 + f1 + w0sq;
	k[3] = -2.0f * e1 - 2.0f * e2 - f1;
	k[4] = e2;
}

__device__ inline int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// coordinates of a point, packed by the kernel's prologue into the level's sixth LDS vector: x | y << 12 | z << 24
// (extents: 2-D <= 512, 3-D <= 64 -- a level has at most 4096 points and no extent below 8)
template <int D>
__device__ inline void unpack_coords(int pc, int* c)
{
	c[0] = pc & 0xFFF;
	if (D > 1) { c[1] = (pc >> 12) & 0xFFF; }
	if (D > 2) { c[2] = (pc >> 24) & 0xFF; }
}

// The values of `v` (a vector of the level, LDS offset `off`) the operator reads around point i: the 3^D box (data rows;
// its axis neighbours serve the model rows too) and the four-D points two steps away along the axes.  Indices are clamped
// into the vector: a clamped value only ever meets a zero coefficient.
template <int D>
struct Around {
	float box[D == 2 ? 9 : 27];
	float far[2 * D];
};
// every vector of the tail: ONE dynamic LDS array, named at file scope so that every access below is an LDS access by type
// (through a `float*` parameter the address space is only inferred, and was not everywhere: flat loads, scratch)
extern __shared__ float fi_tail_lds[];
#define FI_TAIL_LDS(off, idx) fi_tail_lds[(off) + (idx)]

template <int D, bool BOX>
__device__ inline void load_around(const TailLevel& L, int off, int i, Around<D>& A)
{
	// (no clamps: two rows / planes of zeros lie on either side of every vector, and whatever a neighbour index finds beyond
	// the point's own row, plane or lattice meets a zero coefficient)
	const int n0 = L.n[0], n01 = L.n[0] * L.n[1];
	if (BOX) {
#pragma unroll
		for (int s = 0; s < (D == 2 ? 9 : 27); ++s) {
			const int dx = s % 3 - 1, dy = (s / 3) % 3 - 1, dz = D > 2 ? s / 9 - 1 : 0;
			A.box[s] = FI_TAIL_LDS(off, i + dx + dy * n0 + dz * n01);
		}
	} else {  // the axis neighbours and the centre only
		constexpr int ctr = D == 2 ? 4 : 13;
		A.box[ctr]     = FI_TAIL_LDS(off, i);
		A.box[ctr - 1] = FI_TAIL_LDS(off, i - 1);
		A.box[ctr + 1] = FI_TAIL_LDS(off, i + 1);
		A.box[ctr - 3] = FI_TAIL_LDS(off, i - n0);
		A.box[ctr + 3] = FI_TAIL_LDS(off, i + n0);
		if (D > 2) {
			A.box[ctr - 9] = FI_TAIL_LDS(off, i - n01);
			A.box[ctr + 9] = FI_TAIL_LDS(off, i + n01);
		}
	}
	A.far[0] = FI_TAIL_LDS(off, i - 2);
	A.far[1] = FI_TAIL_LDS(off, i + 2);
	A.far[2] = FI_TAIL_LDS(off, i - 2 * n0);
	A.far[3] = FI_TAIL_LDS(off, i + 2 * n0);
	if (D > 2) {
		A.far[4] = FI_TAIL_LDS(off, i - 2 * n01);
		A.far[5] = FI_TAIL_LDS(off, i + 2 * n01);
	}
}

// (A_model v)_i and the model diagonal of point i: the five coefficients of the point's row along every axis come from
// the level's table (axis_coefs of every coordinate, filled by the kernel's prologue)
template <int D>
__device__ inline float model_row(const TailLevel& L, const int* c, const Around<D>& A, float* diag)
{
	constexpr int ctr = D == 2 ? 4 : 13;
	float acc = 0.0f, m = 0.0f;
	int   at = L.ktab;
#pragma unroll
	for (int d = 0; d < D; ++d) {
		const int kk = at + 5 * c[d];
		const float k0 = FI_TAIL_LDS(kk, 0), k1 = FI_TAIL_LDS(kk, 1), k2 = FI_TAIL_LDS(kk, 2), k3 = FI_TAIL_LDS(kk, 3), k4 = FI_TAIL_LDS(kk, 4);
		const int st = d == 0 ? 1 : (d == 1 ? 3 : 9);  // the axis neighbour's place in the box
		m += k2;
		acc += k0 * A.far[2 * d] + k1 * A.box[ctr - st] + k2 * A.box[ctr] + k3 * A.box[ctr + st] + k4 * A.far[2 * d + 1];
		at += 5 * L.n[d];
	}
	*diag = m;
	return acc;
}

// (A_data v)_i: the 3^D diagonals (global memory, consecutive points in consecutive lanes)
template <int D>
__device__ inline float data_row(const TailLevel& L, int i, const Around<D>& A)
{
	float acc = 0.0f;
#pragma unroll
	for (int s = 0; s < (D == 2 ? 9 : 27); ++s) { acc += L.dia[s * L.nn + i] * A.box[s]; }
	return acc;
}

template <int D>
__device__ inline void stage(const TailLevel* __restrict__ levels, const TailOp& op, int tid)
{
	constexpr int nthreads = kTailThreads;
	const TailLevel& L = levels[op.level];
	const int ctab = L.ctab;
	const bool data = L.dia != nullptr;
	switch (op.kind) {
	case kTailScale:
		for (int i = tid; i < L.nn; i += nthreads) { FI_TAIL_LDS(op.out, i) = op.s0 * bf16(op.scale[i]) * FI_TAIL_LDS(op.a, i); }
		break;
	case kTailPolyStep:
		for (int i = tid; i < L.nn; i += nthreads) {
			int c[3];
			unpack_coords<D>(__float_as_int(FI_TAIL_LDS(ctab, i)), c);
			Around<D> A;
			load_around<D, false>(L, op.a, i, A);
			float m;
			const float q  = model_row<D>(L, c, A, &m);
			const float z  = A.box[D == 2 ? 4 : 13], dv = bf16(op.scale[i]);
			const float sv = dv * (q - m * z) + z;
			const float zp = op.b >= 0 ? FI_TAIL_LDS(op.b, i) : 0.0f;
			const float zn = op.s0 * z - op.s1 * zp + op.s2 * (dv * FI_TAIL_LDS(op.c, i) - sv);
			if (op.out >= 0) { FI_TAIL_LDS(op.out, i) = zn; }
			if (op.acc >= 0) { FI_TAIL_LDS(op.acc, i) += zn; }
		}
		break;
	case kTailChebStep:
	case kTailResidual: {
		// A thread's points (at most kTailMaxPoints / kTailThreads = 4): the diagonals of ALL of them are requested from global
		// memory before the first is used -- one point after the other the stage was a chain of cache round trips (5 us for 4096
		// points: four trips of ~1 us per wave), not work
		constexpr int PER = static_cast<int>(kTailMaxPoints) / kTailThreads;
		constexpr int NS  = D == 2 ? 9 : 27;
		float dia[PER][D == 2 ? 9 : 1];
		float scl[PER];
		if (D == 2 && data) {
#pragma unroll
			for (int k = 0; k < PER; ++k) {
				const int i = tid + k * nthreads;
				const int ii = i < L.nn ? i : L.nn - 1;
#pragma unroll
				for (int s2 = 0; s2 < NS; ++s2) { dia[k][s2 % (D == 2 ? 9 : 1)] = L.dia[s2 * L.nn + ii]; }
			}
		}
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const int i = tid + k * nthreads;
			scl[k] = (op.kind == kTailChebStep && i < L.nn) ? bf16(op.scale[i]) : 0.0f;
		}
#pragma unroll
		for (int k = 0; k < PER; ++k) {
			const int i = tid + k * nthreads;
			if (i >= L.nn) { break; }
			int c[3];
			unpack_coords<D>(__float_as_int(FI_TAIL_LDS(ctab, i)), c);
			Around<D> A;
			float m, q;
			if (data) {
				load_around<D, true>(L, op.a, i, A);
				q = model_row<D>(L, c, A, &m);
				if (D == 2) {
#pragma unroll
					for (int s2 = 0; s2 < 9; ++s2) { q += dia[k][s2 % (D == 2 ? 9 : 1)] * A.box[s2]; }
				} else {
					q += data_row<D>(L, i, A);
				}
			} else {
				load_around<D, false>(L, op.a, i, A);
				q = model_row<D>(L, c, A, &m);
			}
			const float rhs = FI_TAIL_LDS(op.c, i);
			if (op.kind == kTailResidual) {
				FI_TAIL_LDS(op.out, i) = rhs - q;
			} else {
				const float xp = op.b >= 0 ? FI_TAIL_LDS(op.b, i) : 0.0f;
				FI_TAIL_LDS(op.out, i) = op.s0 * A.box[D == 2 ? 4 : 13] - op.s1 * xp + op.s2 * (scl[k] * (rhs - q));
			}
		}
		break;
	}
	case kTailRestrict: {
		const LevelPair& P = L.to_coarse;
		const TailLevel& C = levels[op.level + 1];
		const int ctab_c = C.ctab;
		for (int j = tid; j < C.nn; j += nthreads) {
			int c[3];
			unpack_coords<D>(__float_as_int(FI_TAIL_LDS(ctab_c, j)), c);
			int   f0[kRTaps], f1[kRTaps], f2[kRTaps];
			float w0[kRTaps], w1[kRTaps], w2[kRTaps];
			restrict_taps<float>(c[0], P.nf[0], P.nc[0], P.cc[0], 0, f0, w0);
			restrict_taps<float>(c[1], P.nf[1], P.nc[1], P.cc[1], 0, f1, w1);
			if (D > 2) { restrict_taps<float>(c[2], P.nf[2], P.nc[2], P.cc[2], 0, f2, w2); }
			float acc = 0.0f;
#pragma unroll
			for (int k2 = 0; k2 < (D > 2 ? kRTaps : 1); ++k2) {
				float r2 = 0.0f;
#pragma unroll
				for (int k1 = 0; k1 < kRTaps; ++k1) {
					const int base = ((D > 2 ? f2[k2] * P.nf[1] : 0) + f1[k1]) * P.nf[0];
					float r = 0.0f;
#pragma unroll
					for (int k0 = 0; k0 < kRTaps; ++k0) { r += w0[k0] * FI_TAIL_LDS(op.a, base + f0[k0]); }
					r2 += w1[k1] * r;
				}
				acc += (D > 2 ? w2[k2] : 1.0f) * r2;
			}
			FI_TAIL_LDS(op.out, j) = acc;
		}
		break;
	}
	case kTailProlongAdd: {
		const LevelPair& P = L.to_coarse;
		for (int i = tid; i < L.nn; i += nthreads) {
			int c[3];
			unpack_coords<D>(__float_as_int(FI_TAIL_LDS(ctab, i)), c);
			int   a0, a1, b0, b1, e0 = 0, e1 = 0;
			float u0, u1, v0, v1, t0 = 1.0f, t1 = 0.0f;
			prolong_taps<float>(c[0], P.nc[0], P.cc[0], &a0, &a1, &u0, &u1);
			prolong_taps<float>(c[1], P.nc[1], P.cc[1], &b0, &b1, &v0, &v1);
			if (D > 2) { prolong_taps<float>(c[2], P.nc[2], P.cc[2], &e0, &e1, &t0, &t1); }
			float acc = 0.0f;
#pragma unroll
			for (int q = 0; q < (1 << D); ++q) {
				const float w = ((q & 1) ? u1 : u0) * ((q & 2) ? v1 : v0) * (D > 2 ? ((q & 4) ? t1 : t0) : 1.0f);
				const int idx = ((q & 1) ? a1 : a0) + P.nc[0] * (((q & 2) ? b1 : b0) + (D > 2 ? P.nc[1] * ((q & 4) ? e1 : e0) : 0));
				acc += w * FI_TAIL_LDS(op.a, idx);
			}
			FI_TAIL_LDS(op.out, i) += acc;
		}
		break;
	}
	default: break;
	}
}

// ONE workgroup.  The right-hand side of the tail's top level comes in from global memory, its result goes out; everything
// in between -- every vector of every tail level -- lives in LDS, a stage ends in a workgroup barrier.  The stage loops have
// no early exit: every wave reaches every barrier.
template <int D>
__global__ __launch_bounds__(kTailThreads) void k_tail(const TailLevel* __restrict__ levels, const TailOp* __restrict__ ops, int nlev, int nops,
                                                        int lds_floats, const float* __restrict__ b, float* __restrict__ x)
{
	float* const lds = fi_tail_lds;
	const int tid = threadIdx.x;
	const int nn0 = levels[0].nn, base0 = levels[0].base;
	// everything zero first (the guard zones stay so), then the right-hand side, the points' coordinates and the model rows'
	// coefficients per coordinate -- once per launch
	for (int i = tid; i < lds_floats; i += kTailThreads) { lds[i] = 0.0f; }
	__syncthreads();
	for (int i = tid; i < nn0; i += kTailThreads) { lds[base0 + i] = b[i]; }
	for (int l = 0; l < nlev; ++l) {
		const TailLevel& L = levels[l];
		for (int i = tid; i < L.nn; i += kTailThreads) {
			const int cx = i % L.n[0], t = i / L.n[0];
			const int cy = D > 2 ? t % L.n[1] : t, cz = D > 2 ? t / L.n[1] : 0;
			lds[L.ctab + i] = __int_as_float(cx | (cy << 12) | (cz << 24));
		}
		int at = L.ktab;
		for (int d = 0; d < D; ++d) {
			for (int cc = tid; cc < L.n[d]; cc += kTailThreads) {
				float k[5];
				axis_coefs(cc, L.n[d], L.w0sq, L.w1sq, L.w2sq, k);
#pragma unroll
				for (int j = 0; j < 5; ++j) { lds[at + 5 * cc + j] = k[j]; }
			}
			at += 5 * L.n[d];
		}
	}
	__syncthreads();
	for (int o = 0; o < nops; ++o) {
		const TailOp op = ops[o];
		stage<D>(levels, op, tid);
		__syncthreads();
	}
	for (int i = tid; i < nn0; i += kTailThreads) { x[i] = lds[base0 + levels[0].vstride + i]; }
}
#undef FI_TAIL_LDS

__global__ __launch_bounds__(kThreads) void k_tail_map(int64_t ncell, const uint32_t* __restrict__ cell_id, uint32_t* __restrict__ map)
{
	const int64_t c = static_cast<int64_t>(blockIdx.x) * kThreads + threadIdx.x;
	if (c < ncell) { map[cell_id[c]] = static_cast<uint32_t>(c); }
}

// The data rows as 3^D diagonals: point i is corner q of the cell with origin (coordinates - bits of q); that cell's block
// couples it to the cell's other corners q', i.e. to the neighbour at offset bits(q') - bits(q).  Summed in the fixed order
// (q, q'): the same bits on every run.
template <int D>
__global__ __launch_bounds__(kThreads) void k_tail_dia(Geom g, int nn, const uint32_t* __restrict__ map, const float* __restrict__ blk,
                                                        float* __restrict__ dia)
{
	constexpr int NC = 1 << D, NB = NC * (NC + 1) / 2, NS = ipow3(D);
	const int i = blockIdx.x * kThreads + threadIdx.x;
	if (i >= nn) { return; }
	int c[3] = {0, 0, 0};
	c[0] = i % g.gn[0];
	int t = i / g.gn[0];
	if (D > 1) {
		c[1] = t % g.gn[1];
		t /= g.gn[1];
	}
	if (D > 2) { c[2] = t; }
	float acc[NS];
#pragma unroll
	for (int s = 0; s < NS; ++s) { acc[s] = 0.0f; }
#pragma unroll
	for (int q = 0; q < NC; ++q) {
		uint32_t key = 0, mul = 1;
#pragma unroll
		for (int d = 0; d < D; ++d) {
			const int l = c[d] - ((q >> d) & 1) - g.coff[d];  // extended cell coordinate: 0 .. cn - 1
			key += static_cast<uint32_t>(l) * mul;
			mul *= static_cast<uint32_t>(g.cn[d]);
		}
		const uint32_t ci = map[key];
		if (ci == 0xFFFFFFFFu) { continue; }
		const float* B = blk + static_cast<int64_t>(ci) * NB;
#pragma unroll
		for (int p = 0; p < NC; ++p) {
			int slot = 0, mul3 = 1;
#pragma unroll
			for (int d = 0; d < D; ++d) {
				slot += (((p >> d) & 1) - ((q >> d) & 1) + 1) * mul3;
				mul3 *= 3;
			}
			acc[slot] += B[packed_index(q < p ? q : p, q < p ? p : q, NC)];
		}
	}
#pragma unroll
	for (int s = 0; s < NS; ++s) { dia[static_cast<int64_t>(s) * nn + i] = acc[s]; }
}

}  // namespace

bool tail_level_supported(const fi_ctx* c)
{
	const fi_weights& w = c->w;
	if (test_switch("FI_NO_TAIL")) { return false; }  // tests: the tiled kernels run every level
	if (c->dtype != FI_F32 || c->nranks != 1 || (c->g.ndim != 2 && c->g.ndim != 3)) { return false; }
	if (w.model_3 > 0 || w.model_4 > 0 || w.gradient_smoothness > 0 || !(w.model_1 > 0 || w.model_2 > 0)) { return false; }
	if (c->generic.ntrip != 0 || c->any_trip) { return false; }
	int64_t nn = 1;
	for (int d = 0; d < c->g.ndim; ++d) { nn *= c->g.gn[d]; }
	return nn <= kTailMaxPoints && c->g.nown == c->g.nloc;
}

void tail_build_operator(fi_ctx* c)
{
	const Geom& g = c->g;
	const int64_t nn = g.nloc;
	const int D = g.ndim, ns = ipow3(D);
	c->tail_prog_valid = false;
	c->dia_valid = false;
	if (c->cells.ncell <= 0) {
		c->tail_dia.release();
		return;
	}
	int64_t ncells_ext = 1;
	for (int d = 0; d < D; ++d) { ncells_ext *= g.cn[d]; }
	c->tail_map.alloc(sizeof(uint32_t) * ncells_ext);
	c->tail_dia.alloc(sizeof(float) * ns * nn);
	FI_HIP_TRY(hipMemsetAsync(c->tail_map.p, 0xFF, sizeof(uint32_t) * ncells_ext, c->stream));
	hipLaunchKernelGGL(k_tail_map, dim3(static_cast<unsigned>((c->cells.ncell + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream,
	                   c->cells.ncell, c->cells.cell_id.as<uint32_t>(), c->tail_map.as<uint32_t>());
	const dim3 grid(static_cast<unsigned>((nn + kThreads - 1) / kThreads));
	ensure_cell_blocks(c);
	if (D == 2) {
		hipLaunchKernelGGL(k_tail_dia<2>, grid, dim3(kThreads), 0, c->stream, g, static_cast<int>(nn), c->tail_map.as<uint32_t>(),
		                   c->cells.blk.as<float>(), c->tail_dia.as<float>());
	} else {
		hipLaunchKernelGGL(k_tail_dia<3>, grid, dim3(kThreads), 0, c->stream, g, static_cast<int>(nn), c->tail_map.as<uint32_t>(),
		                   c->cells.blk.as<float>(), c->tail_dia.as<float>());
	}
	FI_HIP_TRY(hipGetLastError());
	c->dia_valid = true;
}

TailLevel tail_level_of(const fi_ctx* c)
{
	TailLevel L{};
	L.ndim = c->g.ndim;
	for (int d = 0; d < 3; ++d) { L.n[d] = c->g.gn[d]; }
	L.nn = static_cast<int>(c->g.nloc);
	const float w0 = c->w.model_0 > 0 ? c->w.model_0 : 0.0f, w1 = c->w.model_1 > 0 ? c->w.model_1 : 0.0f;
	const float w2 = c->w.model_2 > 0 ? c->w.model_2 : 0.0f;
	L.w0sq = w0 * w0;
	L.w1sq = w1 * w1;
	L.w2sq = w2 * w2;
	L.dia  = (c->cells.ncell > 0 && c->tail_dia.p) ? c->tail_dia.as<float>() : nullptr;
	return L;
}

void tail_run(fi_ctx* top, const void* prog, int nlev, int nops, int lds_floats, const float* b, float* x)
{
	// the dynamic LDS limit of k_tail<2> / k_tail<3> has been raised -- per DEVICE (the attribute belongs to the function on
	// a device: a process that drives a second GPU must raise it there too) and under a lock
	static std::mutex          lds_mutex;
	static std::map<int, bool> lds_allowed[2];
	const int D = top->g.ndim;
	const size_t lds_bytes = sizeof(float) * static_cast<size_t>(lds_floats);
	FI_REQUIRE(lds_bytes <= 160u * 1024u, FI_ERR_STATE, "the small-level engine's vectors (%zu bytes) do not fit the LDS", lds_bytes);
	{
		std::lock_guard<std::mutex> lock(lds_mutex);
		if (!lds_allowed[D - 2][top->device]) {
			if (D == 2) {
				FI_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tail<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
			} else {
				FI_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tail<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
			}
			lds_allowed[D - 2][top->device] = true;
		}
	}
	const TailLevel* levels = static_cast<const TailLevel*>(prog);
	const TailOp*    ops    = reinterpret_cast<const TailOp*>(levels + kTailMaxLevels);
	if (D == 2) {
		hipLaunchKernelGGL(k_tail<2>, dim3(1), dim3(kTailThreads), lds_bytes, top->stream, levels, ops, nlev, nops, lds_floats, b, x);
	} else {
		hipLaunchKernelGGL(k_tail<3>, dim3(1), dim3(kTailThreads), lds_bytes, top->stream, levels, ops, nlev, nops, lds_floats, b, x);
	}
	FI_HIP_TRY(hipGetLastError());
}

}  // namespace fi
