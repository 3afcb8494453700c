// micro benchmark (round 6): how fast can the 13-point model_2 star y = S^T S x run on gfx950 WITHOUT the LDS plane ring and
// its barrier per plane?  profiles/r6_ablation.md section 1.
//   copy   : y = x, a thread per 16 bytes, grid-stride -- the achievable streaming rate for one read + one write stream
//   naive  : a thread per VX points, 13 neighbours through L1 / L2, no reuse in registers
//   strip  : a WAVE owns a strip of 64 * VX points along x by RY rows and marches along z; z neighbours in a register ring,
//            y neighbours in the lane's own registers (RY rows + 4 halo rows loaded from the neighbouring strips' lines),
//            x neighbours from the neighbouring lanes by DPP wave shifts (lanes 0 / 63: a 16-byte halo load); no LDS, no barrier
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast tools/micro/stencil_probe.hip -o exp_libs/stencil_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(e)                                                                                  \
	do {                                                                                       \
		hipError_t e_ = (e);                                                                   \
		if (e_ != hipSuccess) {                                                                \
			fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));         \
			exit(1);                                                                           \
		}                                                                                      \
	} while (0)

template <typename T>
struct Vec;
template <>
struct Vec<double> {
	using V = double2;
	static constexpr int VX = 2;
};
template <>
struct Vec<float> {
	using V = float4;
	static constexpr int VX = 4;
};

struct Dim {
	int nx, ny, nz;
	long long plane;
};

template <typename T>
__global__ __launch_bounds__(256) void k_copy(const T* __restrict__ x, T* __restrict__ y, long long n16)
{
	using V = typename Vec<T>::V;
	const V* xs = reinterpret_cast<const V*>(x);
	V*       ys = reinterpret_cast<V*>(y);
	for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n16; i += static_cast<long long>(gridDim.x) * 256) { ys[i] = xs[i]; }
}

// ---- the reference: masks from global coordinates, a thread per point -------------------------------------------------
template <typename T>
__device__ inline T u_row(const T* p, long long i, long long s, int a, int n)  // row anchored at coordinate a along an axis of stride s
{
	return (a >= 0 && a + 2 < n) ? p[i] - T(2) * p[i + s] + p[i + 2 * s] : T(0);
}
template <typename T>
__global__ __launch_bounds__(256) void k_naive(Dim d, T w2sq, const T* __restrict__ x, T* __restrict__ y)
{
	const long long i = blockIdx.x * 256ll + threadIdx.x;
	const long long n = d.plane * d.nz;
	if (i >= n) { return; }
	const int cx = static_cast<int>(i % d.nx), cy = static_cast<int>((i / d.nx) % d.ny), cz = static_cast<int>(i / d.plane);
	T acc = T(0);
	const long long st[3] = {1, d.nx, d.plane};
	const int c[3] = {cx, cy, cz}, sz[3] = {d.nx, d.ny, d.nz};
	for (int ax = 0; ax < 3; ++ax) {
		const long long s = st[ax];
		acc += u_row(x, i - 2 * s, s, c[ax] - 2, sz[ax]) - T(2) * u_row(x, i - s, s, c[ax] - 1, sz[ax]) + u_row(x, i, s, c[ax], sz[ax]);
	}
	y[i] = w2sq * acc;
}

// ---- the strip kernel ----------------------------------------------------------------------------------------------
// RY rows per lane, OWN = ring of own planes (3 live + OWN - 3 steps of lead), HAL = sets of halo registers (lead HAL steps)
template <typename T, int RY, int OWN, int HAL, int WPS>
__global__ __launch_bounds__(256, WPS) void k_strip(Dim d, int zc, T w2sq, const T* __restrict__ x, T* __restrict__ y, int wpb)
{
	using V = typename Vec<T>::V;
	constexpr int VX = Vec<T>::VX;
	constexpr int TX = 64 * VX;
	constexpr int U  = (OWN % HAL == 0) ? OWN : OWN * HAL;  // instantiations of the step: every ring index a constant
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int tiles_x = d.nx / TX, tiles_y = d.ny / (wpb * RY);
	// XCD-aware order as in the product kernel: blocks b, b + 8, ... (one XCD) take neighbouring tiles
	const int nwg  = tiles_x * tiles_y * ((d.nz + zc - 1) / zc);
	const int per  = (nwg + 7) / 8;
	const int slot = (blockIdx.x % 8) * per + blockIdx.x / 8;
	if (slot >= nwg) { return; }
	const int txy = slot % (tiles_x * tiles_y), chunk = slot / (tiles_x * tiles_y);
	const int x0 = (txy % tiles_x) * TX, y0 = (txy / tiles_x) * (wpb * RY) + wave * RY;
	const int gx = x0 + VX * lane;
	const int z_begin = chunk * zc, z_end = (z_begin + zc < d.nz) ? z_begin + zc : d.nz;

	// x masks of the VX + 2 rows anchored at gx - 2 + k (lane masks); y masks per own row (wave-uniform)
	bool m2x[VX + 2];
#pragma unroll
	for (int k = 0; k < VX + 2; ++k) {
		const int a = gx - 2 + k;
		m2x[k] = a >= 0 && a + 2 < d.nx;
	}
	T cy[RY][3];
#pragma unroll
	for (int j = 0; j < RY; ++j) {
		const int gy = y0 + j;
		cy[j][0] = (gy - 2 >= 0 && gy < d.ny) ? T(1) : T(0);
		cy[j][1] = (gy - 1 >= 0 && gy + 1 < d.ny) ? T(-2) : T(0);
		cy[j][2] = (gy + 2 < d.ny) ? T(1) : T(0);
	}
	// element offsets inside a plane (32-bit), clamped into the lattice: a clamped value only ever meets a zero mask
	uint32_t own_off[RY], hy_off[4], hx_off[RY];
#pragma unroll
	for (int j = 0; j < RY; ++j) {
		own_off[j] = static_cast<uint32_t>(y0 + j) * d.nx + gx;
		int hx = lane == 63 ? x0 + TX : x0 - 2;   // (lanes 1..62 read lane 0's address: one line more, no branch)
		hx = hx < 0 ? 0 : (hx > d.nx - 2 ? d.nx - 2 : hx);
		hx_off[j] = static_cast<uint32_t>(y0 + j) * d.nx + hx;
	}
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		int hy = k < 2 ? y0 - 2 + k : y0 + RY + (k - 2);
		hy = hy < 0 ? 0 : (hy >= d.ny ? d.ny - 1 : hy);
		hy_off[k] = static_cast<uint32_t>(hy) * d.nx + gx;
	}
	auto plane_of = [&](int z) { return x + static_cast<long long>(z < 0 ? 0 : (z >= d.nz ? d.nz - 1 : z)) * d.plane; };
	struct Halo {
		V hy[4];
		V hx[RY];
	};
	V X[OWN][RY];
	Halo H[HAL];
	auto load_own = [&](int z, V* dst) {
		const T* p = plane_of(z);
#pragma unroll
		for (int j = 0; j < RY; ++j) { dst[j] = *reinterpret_cast<const V*>(p + own_off[j]); }
	};
	auto load_halo = [&](int z, Halo& h) {
		const T* p = plane_of(z);
#pragma unroll
		for (int k = 0; k < 4; ++k) { h.hy[k] = *reinterpret_cast<const V*>(p + hy_off[k]); }
#pragma unroll
		for (int j = 0; j < RY; ++j) {
			if constexpr (VX == 2) {
				h.hx[j] = *reinterpret_cast<const V*>(p + hx_off[j]);
			} else {  // fp32: two neighbours = 8 bytes
				const float2 v = *reinterpret_cast<const float2*>(p + hx_off[j]);
				h.hx[j] = V{v.x, v.y, 0.f, 0.f};
			}
		}
	};
	// carried row values along z
	T U1[RY][VX], U2[RY][VX];
	{
		V a[RY], b[RY];
		load_own(z_begin - 2, a);
		load_own(z_begin - 1, b);
#pragma unroll
		for (int k = 0; k < OWN - 1; ++k) { load_own(z_begin + k, X[k]); }
#pragma unroll
		for (int k = 0; k < HAL; ++k) { load_halo(z_begin + k, H[k]); }
		const int g2 = z_begin - 2, g1 = z_begin - 1;
		const T m2 = (g2 >= 0 && g2 + 2 < d.nz) ? T(1) : T(0), m1 = (g1 >= 0 && g1 + 2 < d.nz) ? T(1) : T(0);
#pragma unroll
		for (int j = 0; j < RY; ++j) {
			const T* pa = reinterpret_cast<const T*>(&a[j]);
			const T* pb = reinterpret_cast<const T*>(&b[j]);
			const T* pc = reinterpret_cast<const T*>(&X[0][j]);
			const T* pd = reinterpret_cast<const T*>(&X[1][j]);
#pragma unroll
			for (int e = 0; e < VX; ++e) {
				U1[j][e] = m2 * (pa[e] - T(2) * pb[e] + pc[e]);
				U2[j][e] = m1 * (pb[e] - T(2) * pc[e] + pd[e]);
			}
		}
	}
	auto shift = [](T own_edge, T v, int ctrl_is_shr) -> T {  // lane i takes lane i -/+ 1's v; the end lane keeps own_edge
		if constexpr (sizeof(T) == 8) {
			const long long o = __double_as_longlong(static_cast<double>(own_edge)), s = __double_as_longlong(static_cast<double>(v));
			int lo, hi;
			if (ctrl_is_shr) {
				lo = __builtin_amdgcn_update_dpp(static_cast<int>(o), static_cast<int>(s), 0x138, 0xF, 0xF, false);
				hi = __builtin_amdgcn_update_dpp(static_cast<int>(o >> 32), static_cast<int>(s >> 32), 0x138, 0xF, 0xF, false);
			} else {
				lo = __builtin_amdgcn_update_dpp(static_cast<int>(o), static_cast<int>(s), 0x130, 0xF, 0xF, false);
				hi = __builtin_amdgcn_update_dpp(static_cast<int>(o >> 32), static_cast<int>(s >> 32), 0x130, 0xF, 0xF, false);
			}
			return static_cast<T>(__longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo)));
		} else {
			const int o = __float_as_int(static_cast<float>(own_edge)), s = __float_as_int(static_cast<float>(v));
			return static_cast<T>(__int_as_float(ctrl_is_shr ? __builtin_amdgcn_update_dpp(o, s, 0x138, 0xF, 0xF, false)
			                                                 : __builtin_amdgcn_update_dpp(o, s, 0x130, 0xF, 0xF, false)));
		}
	};

	const int nsteps = z_end - z_begin;
	for (int s0 = 0; s0 < nsteps; s0 += U) {
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int s = s0 + u;
			if (s >= nsteps) { break; }
			const int z = z_begin + s;
			V* xc  = X[u % OWN];
			V* xp1 = X[(u + 1) % OWN];
			V* xp2 = X[(u + 2) % OWN];
			load_own(z + OWN - 1, X[(u + OWN - 1) % OWN]);
			Halo& h = H[u % HAL];
			const T mz = (z + 2 < d.nz) ? T(1) : T(0);
			T* yp = y + static_cast<long long>(z) * d.plane;
#pragma unroll
			for (int j = 0; j < RY; ++j) {
				const T* pc  = reinterpret_cast<const T*>(&xc[j]);
				const T* pp1 = reinterpret_cast<const T*>(&xp1[j]);
				const T* pp2 = reinterpret_cast<const T*>(&xp2[j]);
				const T* hx  = reinterpret_cast<const T*>(&h.hx[j]);
				T acc[VX];
				// x: window of VX + 4 values
				T w[VX + 4];
				w[0] = shift(hx[0], pc[VX - 2], 1);
				w[1] = shift(hx[1], pc[VX - 1], 1);
#pragma unroll
				for (int e = 0; e < VX; ++e) { w[2 + e] = pc[e]; }
				w[VX + 2] = shift(hx[0], pc[0], 0);
				w[VX + 3] = shift(hx[1], pc[1], 0);
				T ux[VX + 2];
#pragma unroll
				for (int k = 0; k < VX + 2; ++k) { ux[k] = m2x[k] ? (w[k] - T(2) * w[k + 1] + w[k + 2]) : T(0); }
#pragma unroll
				for (int e = 0; e < VX; ++e) { acc[e] = ux[e] - T(2) * ux[e + 1] + ux[e + 2]; }
				// y: rows j-2 .. j+2 out of the halo rows and the own rows
				auto row = [&](int r) -> const T* {
					return r < 0 ? reinterpret_cast<const T*>(&h.hy[2 + r]) : (r >= RY ? reinterpret_cast<const T*>(&h.hy[2 + (r - RY)]) : reinterpret_cast<const T*>(&xc[r]));
				};
				const T *r0 = row(j - 2), *r1 = row(j - 1), *r3 = row(j + 1), *r4 = row(j + 2);
#pragma unroll
				for (int e = 0; e < VX; ++e) {
					const T ua = r0[e] - T(2) * r1[e] + pc[e];
					const T ub = r1[e] - T(2) * pc[e] + r3[e];
					const T uc = pc[e] - T(2) * r3[e] + r4[e];
					acc[e] += cy[j][0] * ua + cy[j][1] * ub + cy[j][2] * uc;
				}
				// z: carried rows
				V out;
				T* po = reinterpret_cast<T*>(&out);
#pragma unroll
				for (int e = 0; e < VX; ++e) {
					const T u0 = mz * (pc[e] - T(2) * pp1[e] + pp2[e]);
					acc[e] += U1[j][e] - T(2) * U2[j][e] + u0;
					U1[j][e] = U2[j][e];
					U2[j][e] = u0;
					po[e] = w2sq * acc[e];
				}
				*reinterpret_cast<V*>(yp + own_off[j]) = out;
			}
			load_halo(z + HAL, h);
		}
	}
}

template <typename T>
static void run(int side, int reps)
{
	Dim d{side, side, side, static_cast<long long>(side) * side};
	const long long n = d.plane * d.nz;
	T *x, *y, *yr;
	CK(hipMalloc(&x, n * sizeof(T) + 256));
	CK(hipMalloc(&y, n * sizeof(T) + 256));
	CK(hipMalloc(&yr, n * sizeof(T) + 256));
	std::vector<T> hx(n);
	unsigned long long sd = 88172645463325252ull;
	for (long long i = 0; i < n; ++i) {
		sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17;
		hx[i] = static_cast<T>((sd >> 11) * (1.0 / 9007199254740992.0) - 0.5);
	}
	CK(hipMemcpy(x, hx.data(), n * sizeof(T), hipMemcpyHostToDevice));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const double bytes = 2.0 * sizeof(T) * n;
	auto time = [&](const char* name, auto launch, bool check) {
		CK(hipMemset(y, 0xFF, n * sizeof(T)));
		launch();
		CK(hipDeviceSynchronize());
		double err = -1;
		if (check) {
			std::vector<T> a(n), b(n);
			CK(hipMemcpy(a.data(), y, n * sizeof(T), hipMemcpyDeviceToHost));
			CK(hipMemcpy(b.data(), yr, n * sizeof(T), hipMemcpyDeviceToHost));
			err = 0;
			for (long long i = 0; i < n; ++i) {
				const double e = std::fabs(static_cast<double>(a[i]) - static_cast<double>(b[i]));
				if (!(e <= err)) { err = e; }
			}
		}
		float best = 1e30f;
		for (int r = 0; r < 3; ++r) {
			CK(hipEventRecord(e0));
			for (int k = 0; k < reps; ++k) { launch(); }
			CK(hipEventRecord(e1));
			CK(hipEventSynchronize(e1));
			float ms;
			CK(hipEventElapsedTime(&ms, e0, e1));
			if (ms / reps < best) { best = ms / reps; }
		}
		printf("%-34s %s side %d: %8.1f us  %.3f of 8 TB/s  maxerr %.2e\n", name, sizeof(T) == 8 ? "f64" : "f32", side, best * 1e3,
		       bytes / (best * 1e-3) / 8e12, err);
		fflush(stdout);
	};
	const T w2sq = T(0.25);
	hipLaunchKernelGGL(k_naive<T>, dim3((n + 255) / 256), dim3(256), 0, 0, d, w2sq, x, yr);
	CK(hipDeviceSynchronize());
	time("copy (grid 256*8)", [&] { hipLaunchKernelGGL(k_copy<T>, dim3(2048), dim3(256), 0, 0, x, y, n * sizeof(T) / 16); }, false);
	time("copy (grid 256*16)", [&] { hipLaunchKernelGGL(k_copy<T>, dim3(4096), dim3(256), 0, 0, x, y, n * sizeof(T) / 16); }, false);
	time("naive", [&] { hipLaunchKernelGGL(k_naive<T>, dim3((n + 255) / 256), dim3(256), 0, 0, d, w2sq, x, y); }, true);
	constexpr int VX = Vec<T>::VX;
	auto strip = [&](const char* name, auto kern, int ry, int zc, int wpb = 4) {
		if (side % (64 * VX) || side % (wpb * ry)) { return; }
		const int nwg = (side / (64 * VX)) * (side / (wpb * ry)) * ((side + zc - 1) / zc);
		char buf[96];
		snprintf(buf, sizeof(buf), "%s wpb=%d zc=%d (%d wgs)", name, wpb, zc, nwg);
		time(buf, [&] { hipLaunchKernelGGL(kern, dim3(((nwg + 7) / 8) * 8), dim3(64 * wpb), 0, 0, d, zc, w2sq, x, y, wpb); }, true);
	};
	for (int zc : {8, 16, 32}) {  // one-wave workgroups: the granularity a 256^3 fp32 level needs (a strip is a whole row)
		strip("strip RY4 OWN4 HAL2 wps1", k_strip<T, 4, 4, 2, 1>, 4, zc, 1);
		strip("strip RY8 OWN4 HAL2 wps1", k_strip<T, 8, 4, 2, 1>, 8, zc, 1);
		strip("strip RY2 OWN4 HAL2 wps1", k_strip<T, 2, 4, 2, 1>, 2, zc, 1);
	}
	for (int zc : {32, 64, 128}) {
		strip("strip RY8 OWN4 HAL2 wps1", k_strip<T, 8, 4, 2, 1>, 8, zc);
		strip("strip RY8 OWN4 HAL1 wps1", k_strip<T, 8, 4, 1, 1>, 8, zc);
		strip("strip RY8 OWN5 HAL1 wps1", k_strip<T, 8, 5, 1, 1>, 8, zc);
		strip("strip RY8 OWN6 HAL2 wps1", k_strip<T, 8, 6, 2, 1>, 8, zc);
		strip("strip RY4 OWN4 HAL2 wps1", k_strip<T, 4, 4, 2, 1>, 4, zc);
		strip("strip RY4 OWN4 HAL2 wps2", k_strip<T, 4, 4, 2, 2>, 4, zc);
		strip("strip RY4 OWN4 HAL1 wps2", k_strip<T, 4, 4, 1, 2>, 4, zc);
		strip("strip RY4 OWN6 HAL2 wps1", k_strip<T, 4, 6, 2, 1>, 4, zc);
	}
	CK(hipFree(x));
	CK(hipFree(y));
	CK(hipFree(yr));
}

int main(int argc, char** argv)
{
	const int side = argc > 1 ? atoi(argv[1]) : 512;
	const int reps = argc > 2 ? atoi(argv[2]) : 10;
	const char* dt = argc > 3 ? argv[3] : "both";
	if (strcmp(dt, "f32")) { run<double>(side, reps); }
	if (strcmp(dt, "f64")) { run<float>(side, reps); }
	return 0;
}
