// Two steps of the polynomial preconditioner's Chebyshev recurrence in ONE launch (fp32, 3-D lattices, one GPU):
//     z_2 = aA z_1 - c1A z_0 + c2A (Dinv r - s(z_1)),   z_0 = zsA * Dinv * r   (the polynomial's second step)
//     z_3 = aB z_2 - c1B z_1 + c2B (Dinv r - s(z_2))                           (its third)
// with s(z) = Dinv (A_model + diag A_data) z as in ChebEpi mode 0 (fi_stencil.hip).  z_2 is never stored: a workgroup
// marches along z over a 64 x 16 tile, forms z_2 on the tile plus a ring of `reach` points from z_1 on the tile plus two
// rings, and z_3 on the tile from that.  Per lattice point the pair reads z_1, r and the bf16 scaling and writes z_3:
// 14 bytes instead of the 14 + 18 of two launches of k_apply_march3d<EPI> (z_2 out and in again, z_1, r and the scaling a
// second time).  Every point of the z_1 region -- (64 + 4 reach) x (16 + 4 reach) columns -- belongs to one thread, which
// keeps the column's last 2 reach + 1 planes of z_1 and of z_2 in registers (the z neighbours); the x / y neighbours of
// the plane in work come from LDS, one plane of z_1 and one of z_2, double-buffered: one barrier per plane.
// Boundary rows are masked from global coordinates like everywhere else (field_interpolation.cpp:273).
#include "fi_internal.h"

#include <hip/hip_runtime.h>

namespace fi {
namespace {

struct PairParams {
	int     nx, ny, nz;
	int64_t plane;
	int     tiles_x, tiles_y, chunks, zc, nwg;
	float   w0x3, w1sq, w2sq;
	float   dfull;  // x / y part of the model diagonal where every row exists
	float   star0, star1;  // interior points: A_model z = star0 z + star1 (6 nearest) + w2sq (6 second nearest)
	float   aA, c1A, c2A, zsA;
	float   aB, c1B, c2B;
};

constexpr int kPairThreads = 512;
constexpr int kPTX = 64, kPTY = 16;

__device__ inline float bf16_to_float(unsigned short v) { return __uint_as_float(static_cast<unsigned int>(v) << 16); }

// bits of a column: 0 the thread owns it, 1 z_2 is formed there, 2 z_3 is formed (and stored) there;
// 3..5 model_2 rows along x anchored at gx-2, gx-1, gx exist; 6..8 the same along y; 9, 10 model_1 rows along x anchored
// at gx-1, gx; 11, 12 along y
constexpr uint32_t kOwn = 1u, kInA = 2u, kInB = 4u;
constexpr uint32_t kFullXY = 1u << 13;  // every model row along x and y through the point exists

template <bool HAS1, bool HAS2, int NR>
__device__ inline float model_rows(const float* __restrict__ pl, int ci, int stride_y, const float* w, uint32_t bits, const float* mz2,
                                   const float* mz1, const PairParams& P, bool fast)
{
	constexpr int RCH = NR / 2;
	const float cv = w[RCH];
	const float xm1 = pl[ci - 1], xp1 = pl[ci + 1], ym1 = pl[ci - stride_y], yp1 = pl[ci + stride_y];
	if (fast) {
		// every row through the point exists (the point is at least `reach` points inside the lattice along every axis):
		// the rows add up to the constant star  c0 z + cA (sum of the 6 nearest) + cB (sum of the 6 second nearest)
		float s1 = (xm1 + xp1) + (ym1 + yp1) + (w[RCH - 1] + w[RCH + 1]);
		float v  = P.star0 * cv + P.star1 * s1;
		if (HAS2) {
			const float s2 = (pl[ci - 2] + pl[ci + 2]) + (pl[ci - 2 * stride_y] + pl[ci + 2 * stride_y]) + (w[RCH - 2] + w[RCH + 2]);
			v += P.w2sq * s2;
		}
		return v;
	}
	float v = P.w0x3 * cv;
	if (HAS2) {
		const float xm2 = pl[ci - 2], xp2 = pl[ci + 2], ym2 = pl[ci - 2 * stride_y], yp2 = pl[ci + 2 * stride_y];
		float acc = 0.0f;
		{
			const float ua = xm2 - 2.0f * xm1 + cv, ub = xm1 - 2.0f * cv + xp1, uc = cv - 2.0f * xp1 + xp2;
			acc += ((bits & (1u << 3)) ? ua : 0.0f) - 2.0f * ((bits & (1u << 4)) ? ub : 0.0f) + ((bits & (1u << 5)) ? uc : 0.0f);
		}
		{
			const float ua = ym2 - 2.0f * ym1 + cv, ub = ym1 - 2.0f * cv + yp1, uc = cv - 2.0f * yp1 + yp2;
			acc += ((bits & (1u << 6)) ? ua : 0.0f) - 2.0f * ((bits & (1u << 7)) ? ub : 0.0f) + ((bits & (1u << 8)) ? uc : 0.0f);
		}
		{
			const float ua = w[RCH - 2] - 2.0f * w[RCH - 1] + cv, ub = w[RCH - 1] - 2.0f * cv + w[RCH + 1], uc = cv - 2.0f * w[RCH + 1] + w[RCH + 2];
			acc += mz2[0] * ua - 2.0f * (mz2[1] * ub) + mz2[2] * uc;
		}
		v += P.w2sq * acc;
	}
	if (HAS1) {
		float acc = 0.0f;
		acc += ((bits & (1u << 9)) ? cv - xm1 : 0.0f) - ((bits & (1u << 10)) ? xp1 - cv : 0.0f);
		acc += ((bits & (1u << 11)) ? cv - ym1 : 0.0f) - ((bits & (1u << 12)) ? yp1 - cv : 0.0f);
		acc += mz1[0] * (cv - w[RCH - 1]) - mz1[1] * (w[RCH + 1] - cv);
		v += P.w1sq * acc;
	}
	return v;
}

// the x / y part of the model diagonal at a column, from its row bits (model_0's term included)
template <bool HAS1, bool HAS2>
__device__ inline float diag_xy(uint32_t b, const PairParams& P)
{
	float m = P.w0x3;
	if (HAS2) {
		const uint32_t ones = ((b >> 3) & 1u) + ((b >> 5) & 1u) + ((b >> 6) & 1u) + ((b >> 8) & 1u), fours = ((b >> 4) & 1u) + ((b >> 7) & 1u);
		m += P.w2sq * static_cast<float>(ones + 4u * fours);
	}
	if (HAS1) { m += P.w1sq * static_cast<float>(((b >> 9) & 1u) + ((b >> 10) & 1u) + ((b >> 11) & 1u) + ((b >> 12) & 1u)); }
	return m;
}

template <bool HAS1, bool HAS2>
__global__ __launch_bounds__(kPairThreads, 4) void k_cheb_pair(PairParams P, const float* __restrict__ z1, const float* __restrict__ r,
                                                                 const unsigned short* __restrict__ s16, float* __restrict__ z3,
                                                                 double* __restrict__ partial, const int* __restrict__ done)
{
	constexpr int RCH = HAS2 ? 2 : 1;
	constexpr int NR = 2 * RCH + 1;
	constexpr int W1 = kPTX + 4 * RCH, H1 = kPTY + 4 * RCH;
	constexpr int NCOLS = W1 * H1;
	constexpr int NK = (NCOLS + kPairThreads - 1) / kPairThreads;
	__shared__ float  la[2][NCOLS], lb[2][NCOLS];
	__shared__ double red[kPairThreads / 64];
	if (done && *done) { return; }
	// XCD-aware order: consecutive tiles on one XCD (blocks b, b + 8, ... share an L2)
	const int per  = (P.nwg + 7) / 8;
	const int slot = static_cast<int>(blockIdx.x % 8) * per + static_cast<int>(blockIdx.x / 8);
	if (slot >= P.nwg) { return; }
	const int tiles_xy = P.tiles_x * P.tiles_y;
	const int chunk = slot / tiles_xy, txy = slot % tiles_xy;
	const int tile_y = txy / P.tiles_x, tile_x = txy % P.tiles_x;
	const int x0 = tile_x * kPTX - 2 * RCH, y0 = tile_y * kPTY - 2 * RCH;  // origin of the z_1 region
	const int z_begin = chunk * P.zc;
	const int z_end   = z_begin + P.zc < P.nz ? z_begin + P.zc : P.nz;

	uint32_t off[NK], bits[NK];  // (a column's LDS index is threadIdx.x + k * kPairThreads)
#pragma unroll
	for (int k = 0; k < NK; ++k) {
		const int  c   = static_cast<int>(threadIdx.x) + k * kPairThreads;
		const bool own = c < NCOLS;
		const int  cc = own ? c : 0;
		const int  ly = cc / W1, lx = cc % W1;
		const int  gx = x0 + lx, gy = y0 + ly;
		const bool in_lattice = gx >= 0 && gx < P.nx && gy >= 0 && gy < P.ny;
		const int  cx = gx < 0 ? 0 : (gx >= P.nx ? P.nx - 1 : gx), cy = gy < 0 ? 0 : (gy >= P.ny ? P.ny - 1 : gy);
		off[k] = static_cast<uint32_t>(cy) * static_cast<uint32_t>(P.nx) + static_cast<uint32_t>(cx);
		uint32_t b = own ? kOwn : 0u;
		if (own && lx >= RCH && lx < W1 - RCH && ly >= RCH && ly < H1 - RCH) { b |= kInA; }
		if (own && lx >= 2 * RCH && lx < W1 - 2 * RCH && ly >= 2 * RCH && ly < H1 - 2 * RCH && in_lattice) { b |= kInB; }
		if (HAS2) {
#pragma unroll
			for (int i = 0; i < 3; ++i) {
				const int a = gx - 2 + i, e = gy - 2 + i;
				if (a >= 0 && a + 2 < P.nx) { b |= 1u << (3 + i); }
				if (e >= 0 && e + 2 < P.ny) { b |= 1u << (6 + i); }
			}
		}
		if (HAS1) {
#pragma unroll
			for (int i = 0; i < 2; ++i) {
				const int a = gx - 1 + i, e = gy - 1 + i;
				if (a >= 0 && a + 1 < P.nx) { b |= 1u << (9 + i); }
				if (e >= 0 && e + 1 < P.ny) { b |= 1u << (11 + i); }
			}
		}
		{
			const uint32_t need = (HAS2 ? 0x1F8u : 0u) | (HAS1 ? 0x1E00u : 0u);
			if ((b & need) == need) { b |= kFullXY; }
		}
		bits[k] = b;
	}

	auto clamp_z = [&](int p) { return p < 0 ? 0 : (p > P.nz - 1 ? P.nz - 1 : p); };
	float z1r[NK][NR], z2r[NK][NR], nxt[NK];
#pragma unroll
	for (int k = 0; k < NK; ++k) {
#pragma unroll
		for (int j = 0; j < NR; ++j) {
			z1r[k][j] = 0.0f;
			z2r[k][j] = 0.0f;
		}
	}
	const int L0 = z_begin - 2 * RCH, L1 = z_end - 1 + 2 * RCH;
	{
		const float* p1 = z1 + static_cast<int64_t>(clamp_z(L0)) * P.plane;
#pragma unroll
		for (int k = 0; k < NK; ++k) { nxt[k] = p1[off[k]]; }
	}
	double dot_acc = 0.0;
	for (int L = L0; L <= L1; ++L) {
		const int pA = L - RCH, pB = L - 2 * RCH;
		const bool doA = pA >= z_begin - RCH, doB = pB >= z_begin;
		// rotate: the newest plane of z_1 and the operands of this step's first stage arrive from the previous step's loads
#pragma unroll
		for (int k = 0; k < NK; ++k) {
#pragma unroll
			for (int j = 0; j + 1 < NR; ++j) { z1r[k][j] = z1r[k][j + 1]; }
			z1r[k][NR - 1] = nxt[k];
		}
		// Loads: unconditional, clamped.  The operands of this step's stages first, the next plane of z_1 LAST: loads return
		// in order, so the wait for the operands leaves the plane that is only needed a step later in flight.
		float rA[NK], rB[NK];
		unsigned short sA[NK], sB[NK];
		{
			const int64_t oa = static_cast<int64_t>(clamp_z(pA)) * P.plane, ob = static_cast<int64_t>(clamp_z(pB)) * P.plane;
#pragma unroll
			for (int k = 0; k < NK; ++k) {
				rA[k] = r[oa + off[k]];
				sA[k] = s16[oa + off[k]];
			}
#pragma unroll
			for (int k = 0; k < NK; ++k) {
				rB[k] = r[ob + off[k]];
				sB[k] = s16[ob + off[k]];
			}
			const float* p1 = z1 + static_cast<int64_t>(clamp_z(L + 1)) * P.plane;
#pragma unroll
			for (int k = 0; k < NK; ++k) { nxt[k] = p1[off[k]]; }
		}
		const int buf = L & 1;
#pragma unroll
		for (int k = 0; k < NK; ++k) {
			if (bits[k] & kOwn) { la[buf][threadIdx.x + k * kPairThreads] = z1r[k][RCH]; }
			if (bits[k] & kInA) { lb[buf][threadIdx.x + k * kPairThreads] = z2r[k][NR - RCH]; }
		}
		__syncthreads();
		float z2new[NK];
#pragma unroll
		for (int k = 0; k < NK; ++k) { z2new[k] = 0.0f; }
		if (doA) {
			float mz2[3], mz1[2], mzd = 0.0f;
#pragma unroll
			for (int i = 0; i < 3; ++i) {
				const int a = pA - 2 + i;
				mz2[i] = (a >= 0 && a + 2 < P.nz) ? 1.0f : 0.0f;
			}
#pragma unroll
			for (int i = 0; i < 2; ++i) {
				const int a = pA - 1 + i;
				mz1[i] = (a >= 0 && a + 1 < P.nz) ? 1.0f : 0.0f;
			}
			if (HAS2) { mzd += P.w2sq * (mz2[0] + 4.0f * mz2[1] + mz2[2]); }
			if (HAS1) { mzd += P.w1sq * (mz1[0] + mz1[1]); }
			const bool zfull = (!HAS2 || (pA - 2 >= 0 && pA + 2 < P.nz)) && (!HAS1 || (pA - 1 >= 0 && pA + 1 < P.nz));
#pragma unroll
			for (int k = 0; k < NK; ++k) {
				if (bits[k] & kInA) {
					const float cv = z1r[k][RCH];
					const float po = model_rows<HAS1, HAS2, NR>(la[buf], static_cast<int>(threadIdx.x) + k * kPairThreads, W1, z1r[k], bits[k], mz2, mz1, P, zfull && (bits[k] & kFullXY));
					const float dv = bf16_to_float(sA[k]);
					const float sv = dv * (po - (((bits[k] & kFullXY) ? P.dfull : diag_xy<HAS1, HAS2>(bits[k], P)) + mzd) * cv) + cv;
					const float zq = P.zsA * dv * rA[k];
					z2new[k] = P.aA * cv - P.c1A * zq + P.c2A * (dv * rA[k] - sv);
				}
			}
		}
		if (doB) {
			float mz2[3], mz1[2], mzd = 0.0f;
#pragma unroll
			for (int i = 0; i < 3; ++i) {
				const int a = pB - 2 + i;
				mz2[i] = (a >= 0 && a + 2 < P.nz) ? 1.0f : 0.0f;
			}
#pragma unroll
			for (int i = 0; i < 2; ++i) {
				const int a = pB - 1 + i;
				mz1[i] = (a >= 0 && a + 1 < P.nz) ? 1.0f : 0.0f;
			}
			if (HAS2) { mzd += P.w2sq * (mz2[0] + 4.0f * mz2[1] + mz2[2]); }
			if (HAS1) { mzd += P.w1sq * (mz1[0] + mz1[1]); }
			const bool zfull = (!HAS2 || (pB - 2 >= 0 && pB + 2 < P.nz)) && (!HAS1 || (pB - 1 >= 0 && pB + 1 < P.nz));
			float* const out = z3 + static_cast<int64_t>(pB) * P.plane;
			float part = 0.0f;
#pragma unroll
			for (int k = 0; k < NK; ++k) {
				if (bits[k] & kInB) {
					float w[NR];  // z_2 at planes pB - reach .. pB + reach: the ring's last 2 reach planes and the one just formed
#pragma unroll
					for (int j = 0; j + 1 < NR; ++j) { w[j] = z2r[k][j + 1]; }
					w[NR - 1] = z2new[k];
					const float cv = w[RCH];
					const float po = model_rows<HAS1, HAS2, NR>(lb[buf], static_cast<int>(threadIdx.x) + k * kPairThreads, W1, w, bits[k], mz2, mz1, P, zfull && (bits[k] & kFullXY));
					const float dv = bf16_to_float(sB[k]);
					const float sv = dv * (po - (((bits[k] & kFullXY) ? P.dfull : diag_xy<HAS1, HAS2>(bits[k], P)) + mzd) * cv) + cv;
					const float zn = P.aB * cv - P.c1B * z1r[k][0] + P.c2B * (dv * rB[k] - sv);
					out[off[k]] = zn;  // (a column of the tile inside the lattice: its offset is not clamped)
					part += rB[k] * zn;
				}
			}
			dot_acc += static_cast<double>(part);
		}
		if (doA) {
#pragma unroll
			for (int k = 0; k < NK; ++k) {
#pragma unroll
				for (int j = 0; j + 1 < NR; ++j) { z2r[k][j] = z2r[k][j + 1]; }
				z2r[k][NR - 1] = z2new[k];
			}
		}
	}
	if (partial) {
		double v = dot_acc;
		for (int o = 32; o > 0; o >>= 1) { v += __shfl_down(v, o, 64); }
		if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = v; }
		__syncthreads();
		if (threadIdx.x == 0) {
			double s = 0.0;
			for (int w = 0; w < kPairThreads / 64; ++w) { s += red[w]; }
			partial[slot] = s;
		}
	}
}

bool pair_geometry(const fi_ctx* c, PairParams* P)
{
	const Geom& g = c->g;
	P->nx = g.gn[0];
	P->ny = g.gn[1];
	P->nz = g.gn[2];
	P->plane   = static_cast<int64_t>(P->nx) * P->ny;
	P->tiles_x = (P->nx + kPTX - 1) / kPTX;
	P->tiles_y = (P->ny + kPTY - 1) / kPTY;
	const int tiles_xy = P->tiles_x * P->tiles_y;
	int cus = 256;
	(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device);
	if (cus <= 0) { cus = 256; }
	// one round of two workgroups per CU, chunks of at least 16 planes (a chunk reads 4 reach planes beyond its own)
	int chunks = (2 * cus + tiles_xy / 2) / tiles_xy;
	if (chunks < 1) { chunks = 1; }
	if (chunks > P->nz / 16) { chunks = P->nz / 16 > 0 ? P->nz / 16 : 1; }
	P->zc     = (P->nz + chunks - 1) / chunks;
	P->chunks = (P->nz + P->zc - 1) / P->zc;
	P->nwg    = tiles_xy * P->chunks;
	// smaller lattices leave CUs idle: the two launches of the marching kernel serve them (FI_PAIR_ALWAYS: tests)
	return P->nwg >= cus + cus / 2 || test_switch("FI_PAIR_ALWAYS");
}

}  // namespace

// The polynomial's steps two and three as one launch: fp32 3-D contexts on one GPU whose operator the marching kernel
// applies (model_0/1/2), large enough to fill the chip.  FI_NO_PAIR (tests): the two launches.
bool cheb_pair_available(const fi_ctx* c)
{
	if (!c->march.valid || c->dtype != FI_F32 || c->nranks != 1 || c->g.ndim != 3 || test_switch("FI_NO_PAIR")) { return false; }
	if (c->g.gn[0] < 8 || c->g.gn[1] < 8) { return false; }
	PairParams P;
	return pair_geometry(c, &P);
}
int cheb_pair_partials(const fi_ctx* c)
{
	PairParams P;
	(void)pair_geometry(c, &P);
	return P.nwg;
}
// z3 = step(step(z1)): the first step's z_prev is zprev_scale * Dinv * r (the polynomial's z_0), the second's is z1.
// z3 must not alias z1; partial: cheb_pair_partials(c) sums of r . z3.
void cheb_pair_step(fi_ctx* c, const void* z1, const void* r, void* z3, double c1A, double c2A, double zprev_scale, double c1B,
                    double c2B, double* partial, const unsigned short* scaling)
{
	FI_REQUIRE(cheb_pair_available(c) && z1 != z3, FI_ERR_UNSUPPORTED, "cheb_pair_step: not available for this context");
	PairParams P;
	(void)pair_geometry(c, &P);
	const fi_weights& w = c->w;
	const float w0 = w.model_0 > 0 ? static_cast<float>(w.model_0) : 0.0f;
	const float w1 = w.model_1 > 0 ? static_cast<float>(w.model_1) : 0.0f;
	const float w2 = w.model_2 > 0 ? static_cast<float>(w.model_2) : 0.0f;
	P.w0x3 = 3.0f * w0 * w0;
	P.w1sq = w1 * w1;
	P.w2sq = w2 * w2;
	P.dfull = P.w0x3 + 12.0f * P.w2sq + 4.0f * P.w1sq;
	P.star0 = P.w0x3 + 18.0f * P.w2sq + 6.0f * P.w1sq;
	P.star1 = -4.0f * P.w2sq - P.w1sq;
	P.aA  = static_cast<float>(1.0 + c1A);
	P.c1A = static_cast<float>(c1A);
	P.c2A = static_cast<float>(c2A);
	P.zsA = static_cast<float>(zprev_scale);
	P.aB  = static_cast<float>(1.0 + c1B);
	P.c1B = static_cast<float>(c1B);
	P.c2B = static_cast<float>(c2B);
	const unsigned short* d16 = scaling ? scaling : c->dinv16.as<unsigned short>();
	const int* done = c->scal.p ? &c->scal.as<CgScalars>()->done : nullptr;
	const dim3 grid(static_cast<unsigned>(8 * ((P.nwg + 7) / 8))), block(kPairThreads);
	const bool has1 = w.model_1 > 0, has2 = w.model_2 > 0;
	auto go = [&](auto kernel) {
		hipLaunchKernelGGL(kernel, grid, block, 0, c->stream, P, static_cast<const float*>(z1), static_cast<const float*>(r), d16,
		                   static_cast<float*>(z3), partial, done);
	};
	if (has1 && has2) {
		go(k_cheb_pair<true, true>);
	} else if (has2) {
		go(k_cheb_pair<false, true>);
	} else {
		go(k_cheb_pair<true, false>);
	}
	FI_HIP_TRY(hipGetLastError());
}

}  // namespace fi
