// fi_tail.hip -- the small-level engine (fi_tail.h): the coarse tail of a level hierarchy in one cooperative launch.
//
// Why: a V-cycle visits every level of <= 64^3 / 512^2 unknowns with 15-25 launches that each finish in 4-7 us -- less than
// the launch itself costs -- and config 3 (2-D 4096^2, seven coarser levels) spent 58 % of its GPU time in 18 820 such
// launches per four steps (profiles/r3_kernel_stats_c3.md).  Here the stages of the cycle (smoother steps, residuals,
// restriction, interpolation) of ALL tail levels run inside one kernel, separated by grid barriers (hipLaunchCooperative-
// Kernel: every workgroup is resident, so the barrier cannot starve).
//
// Operator of a level: the model rows (model_0 / model_1 / model_2, field_interpolation.cpp:257-280) matrix-free from the
// global coordinates -- the same boundary rows as the tiled kernels -- and the data rows (cell blocks of fi_assembly.hip) as
// 3^D diagonals (`dia`): out-of-lattice corners of a cell carry zero coefficients, so the diagonals need no masks.
// Reference role: the small exact solves of tile_solver_square (sparse_linear.cpp:246-390).
#include <hip/hip_cooperative_groups.h>

#include "fi_tail.h"

namespace cg = cooperative_groups;

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

template <int D>
__device__ inline void coords_of(const TailLevel& L, int i, int* c)
{
	c[0] = i % L.n[0];
	int t = i / L.n[0];
	if (D > 1) {
		c[1] = t % L.n[1];
		t /= L.n[1];
	}
	if (D > 2) { c[2] = t; }
}

__device__ inline int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// (A_model v)_i and the model diagonal of point i
template <int D>
__device__ inline float model_row(const TailLevel& L, int i, const int* c, const float* __restrict__ v, float* diag)
{
	float acc = 0.0f, m = 0.0f;
	int stride = 1;
#pragma unroll
	for (int d = 0; d < D; ++d) {
		float k[5];
		axis_coefs(c[d], L.n[d], L.w0sq, L.w1sq, L.w2sq, k);
		m += k[2];
#pragma unroll
		for (int t = 0; t < 5; ++t) {
			// (a coefficient is non-zero only where the neighbour exists; the clamp keeps the other loads inside the array)
			acc += k[t] * v[clampi(i + (t - 2) * stride, L.nn - 1)];
		}
		stride *= L.n[d];
	}
	*diag = m;
	return acc;
}

// (A_data v)_i: the 3^D diagonals
template <int D>
__device__ inline float data_row(const TailLevel& L, int i, const float* __restrict__ v)
{
	if (!L.dia) { return 0.0f; }
	float acc = 0.0f;
	const int n0 = L.n[0], n01 = L.n[0] * (D > 1 ? L.n[1] : 1);
#pragma unroll
	for (int s = 0; s < ipow3(D); ++s) {
		const int dx = s % 3 - 1, dy = D > 1 ? (s / 3) % 3 - 1 : 0, dz = D > 2 ? s / 9 - 1 : 0;
		const int off = dx + dy * n0 + dz * n01;
		acc += L.dia[static_cast<int64_t>(s) * L.nn + i] * v[clampi(i + off, L.nn - 1)];
	}
	return acc;
}

template <int D>
__device__ inline void stage(const TailLevel* __restrict__ levels, const TailOp& op, int tid, int nthreads)
{
	const TailLevel& L = levels[op.level];
	switch (op.kind) {
	case kTailScale:
		for (int i = tid; i < L.nn; i += nthreads) { op.out[i] = op.s0 * bf16(op.scale[i]) * op.a[i]; }
		break;
	case kTailPolyStep:
		for (int i = tid; i < L.nn; i += nthreads) {
			int c[3];
			coords_of<D>(L, i, c);
			float m;
			const float q  = model_row<D>(L, i, c, op.a, &m);
			const float z  = op.a[i], dv = bf16(op.scale[i]);
			const float sv = dv * (q - m * z) + z;
			const float zp = op.b ? op.b[i] : 0.0f;
			const float zn = op.s0 * z - op.s1 * zp + op.s2 * (dv * op.c[i] - sv);
			if (op.out) { op.out[i] = zn; }
			if (op.acc) { op.acc[i] += zn; }
		}
		break;
	case kTailChebStep:
		for (int i = tid; i < L.nn; i += nthreads) {
			int c[3];
			coords_of<D>(L, i, c);
			float m;
			const float q  = model_row<D>(L, i, c, op.a, &m) + data_row<D>(L, i, op.a);
			const float xp = op.b ? op.b[i] : 0.0f;
			op.out[i] = op.s0 * op.a[i] - op.s1 * xp + op.s2 * (bf16(op.scale[i]) * (op.c[i] - q));
		}
		break;
	case kTailResidual:
		for (int i = tid; i < L.nn; i += nthreads) {
			int c[3];
			coords_of<D>(L, i, c);
			float m;
			op.out[i] = op.c[i] - (model_row<D>(L, i, c, op.a, &m) + data_row<D>(L, i, op.a));
		}
		break;
	case kTailRestrict: {
		const LevelPair& P = L.to_coarse;
		const TailLevel& C = levels[op.level + 1];
		for (int j = tid; j < C.nn; j += nthreads) {
			int c[3];
			coords_of<D>(C, j, c);
			int   f[3][kRTaps];
			float w[3][kRTaps];
#pragma unroll
			for (int d = 0; d < D; ++d) { restrict_taps<float>(c[d], P.nf[d], P.nc[d], P.cc[d], 0, f[d], w[d]); }
			float acc = 0.0f;
			if (D == 2) {
#pragma unroll
				for (int k1 = 0; k1 < kRTaps; ++k1) {
					float r = 0.0f;
#pragma unroll
					for (int k0 = 0; k0 < kRTaps; ++k0) { r += w[0][k0] * op.a[f[1][k1] * P.nf[0] + f[0][k0]]; }
					acc += w[1][k1] * r;
				}
			} else {
#pragma unroll
				for (int k2 = 0; k2 < kRTaps; ++k2) {
					if (w[2][k2] == 0.0f) { continue; }
					float r2 = 0.0f;
#pragma unroll
					for (int k1 = 0; k1 < kRTaps; ++k1) {
						if (w[1][k1] == 0.0f) { continue; }
						const int base = (f[2][k2] * P.nf[1] + f[1][k1]) * P.nf[0];
						float r = 0.0f;
#pragma unroll
						for (int k0 = 0; k0 < kRTaps; ++k0) { r += w[0][k0] * op.a[base + f[0][k0]]; }
						r2 += w[1][k1] * r;
					}
					acc += w[2][k2] * r2;
				}
			}
			op.out[j] = acc;
		}
		break;
	}
	case kTailProlongAdd: {
		const LevelPair& P = L.to_coarse;
		for (int i = tid; i < L.nn; i += nthreads) {
			int c[3];
			coords_of<D>(L, i, c);
			int   c0[3], c1[3];
			float w0[3], w1[3];
#pragma unroll
			for (int d = 0; d < D; ++d) { prolong_taps<float>(c[d], P.nc[d], P.cc[d], &c0[d], &c1[d], &w0[d], &w1[d]); }
			float acc = 0.0f;
#pragma unroll
			for (int q = 0; q < (1 << D); ++q) {
				float w   = (q & 1) ? w1[0] : w0[0];
				int   idx = (q & 1) ? c1[0] : c0[0];
				if (D > 1) {
					w *= (q & 2) ? w1[1] : w0[1];
					idx += P.nc[0] * ((q & 2) ? c1[1] : c0[1]);
				}
				if (D > 2) {
					w *= (q & 4) ? w1[2] : w0[2];
					idx += P.nc[0] * P.nc[1] * ((q & 4) ? c1[2] : c0[2]);
				}
				if (w != 0.0f) { acc += w * op.a[idx]; }
			}
			op.out[i] += acc;
		}
		break;
	}
	default: break;
	}
}

// One barrier per stage.  A monotone arrival counter (zeroed by the host before the launch): the stage with index s is
// complete when the counter has reached (s + 1) * gridDim.x.  Every workgroup is resident (cooperative launch), and every
// wave reaches every barrier (the stage loops have no early exit), so the wait always ends.
__device__ inline void grid_barrier(unsigned int* counter, unsigned int target)
{
	__syncthreads();
	if (threadIdx.x == 0) {
		__threadfence();
		atomicAdd(counter, 1u);
		while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) { __builtin_amdgcn_s_sleep(1); }
		__threadfence();
	}
	__syncthreads();
}

template <int D>
__global__ __launch_bounds__(kThreads) void k_tail(const TailLevel* __restrict__ levels, const TailOp* __restrict__ ops, int nops,
                                                    unsigned int* counter)
{
	const int tid = blockIdx.x * kThreads + threadIdx.x, nthreads = gridDim.x * kThreads;
	for (int o = 0; o < nops; ++o) {
		const TailOp op = ops[o];
		stage<D>(levels, op, tid, nthreads);
		if (o + 1 < nops) { grid_barrier(counter, static_cast<unsigned int>(o + 1) * gridDim.x); }
	}
}

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
	// MEASURED AND NOT SHIPPED (profiles/r4_ablation.md section 5): a grid barrier across the 8 XCDs of an MI355X costs as much as
	// the kernel boundary it replaces (3-25 us with 64-512 workgroups arriving on one counter, cache write-back and
	// invalidate included), so 45 stages in one launch take 0.29-1.45 ms where 45 launches take 0.2 ms.  The engine runs in
	// timing builds only (-DFI_TIMING_BUILD, FI_TAIL_ENGINE=1).
	if (!tuning_switch("FI_TAIL_ENGINE")) { return false; }
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

void tail_run(fi_ctx* top, const void* prog, int nlev, int nops, int64_t widest)
{
	(void)nlev;
	static int max_blocks[2] = {0, 0};  // resident workgroups of k_tail<2> / k_tail<3> on this device
	const int D = top->g.ndim;
	int& limit = max_blocks[D - 2];
	if (limit == 0) {
		int per_cu = 0, cus = 0;
		if (D == 2) {
			FI_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tail<2>, kThreads, 0));
		} else {
			FI_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tail<3>, kThreads, 0));
		}
		FI_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, top->device));
		limit = per_cu * cus;
		FI_REQUIRE(limit > 0, FI_ERR_HIP, "the small-level engine does not fit the device");
	}
	// two workgroups per CU at most: the barrier's cost grows with the number of arrivals, the stages' work does not need more
	int cus = 256;
	(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, top->device);
	int64_t want = (widest + kThreads - 1) / kThreads;
	if (want > 2 * cus) { want = 2 * cus; }
	if (want > limit) { want = limit; }
	if (const char* e = tuning_switch("FI_TAIL_BLOCKS")) {
		if (atoi(e) > 0 && atoi(e) <= limit) { want = atoi(e); }
	}
	if (want < 1) { want = 1; }
	top->tail_bar.alloc(sizeof(unsigned int) * 16);
	FI_HIP_TRY(hipMemsetAsync(top->tail_bar.p, 0, sizeof(unsigned int), top->stream));
	const TailLevel* levels = static_cast<const TailLevel*>(prog);
	const TailOp*    ops    = reinterpret_cast<const TailOp*>(levels + kTailMaxLevels);
	unsigned int*    bar    = top->tail_bar.as<unsigned int>();
	void* args[] = {&levels, &ops, &nops, &bar};
	if (D == 2) {
		FI_HIP_TRY(hipLaunchCooperativeKernel(reinterpret_cast<void*>(k_tail<2>), dim3(static_cast<unsigned>(want)), dim3(kThreads), args, 0,
		                                      top->stream));
	} else {
		FI_HIP_TRY(hipLaunchCooperativeKernel(reinterpret_cast<void*>(k_tail<3>), dim3(static_cast<unsigned>(want)), dim3(kThreads), args, 0,
		                                      top->stream));
	}
}

}  // namespace fi
