// scratch: rocprim radix_sort_pairs configs for the assembly's row sort (u32 keys, u32 values)
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class Config>
int run(const char* name, unsigned n, int bits, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout)
{
	size_t tb = 0;
	CK((rocprim::radix_sort_pairs<Config>(nullptr, tb, kin, kout, vin, vout, n, 0, bits, 0)));
	void* tmp = nullptr;
	CK(hipMalloc(&tmp, tb));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int w = 0; w < 3; ++w) { CK((rocprim::radix_sort_pairs<Config>(tmp, tb, kin, kout, vin, vout, n, 0, bits, 0))); }
	CK(hipDeviceSynchronize());
	CK(hipEventRecord(e0, 0));
	const int reps = 20;
	for (int r = 0; r < reps; ++r) { CK((rocprim::radix_sort_pairs<Config>(tmp, tb, kin, kout, vin, vout, n, 0, bits, 0))); }
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<uint32_t> h(n);
	CK(hipMemcpy(h.data(), kout, 4ull * n, hipMemcpyDeviceToHost));
	bool ok = true;
	for (unsigned i = 1; i < n; ++i) { if (h[i - 1] > h[i]) { ok = false; break; } }
	printf("%-28s n %8u bits %2d: %7.1f us per sort  %s\n", name, n, bits, 1e3 * ms / reps, ok ? "sorted" : "NOT SORTED");
	CK(hipFree(tmp));
	return 0;
}

using namespace rocprim;
template <unsigned BS, unsigned IPT, unsigned RB>
using OS = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<BS, IPT>, kernel_config<BS, IPT>, RB, block_radix_rank_algorithm::match>, 16384>;

int main()
{
	for (unsigned n : {1000000u, 4000000u, 20000000u}) {
		for (int bits : {16, 19, 22, 25}) {
			std::vector<uint32_t> k(n), v(n);
			std::mt19937 rng(1);
			for (unsigned i = 0; i < n; ++i) { k[i] = rng() & ((1u << bits) - 1); v[i] = i; }
			uint32_t *kin, *kout, *vin, *vout;
			CK(hipMalloc(&kin, 4ull * n)); CK(hipMalloc(&kout, 4ull * n)); CK(hipMalloc(&vin, 4ull * n)); CK(hipMalloc(&vout, 4ull * n));
			CK(hipMemcpy(kin, k.data(), 4ull * n, hipMemcpyHostToDevice));
			CK(hipMemcpy(vin, v.data(), 4ull * n, hipMemcpyHostToDevice));
						if (run<OS<1024, 8, 9>>("1024x8, 9 bits", n, bits, kin, kout, vin, vout)) return 1;
			if (run<OS<1024, 8, 10>>("1024x8, 10 bits", n, bits, kin, kout, vin, vout)) return 1;
			if (run<OS<1024, 4, 9>>("1024x4, 9 bits", n, bits, kin, kout, vin, vout)) return 1;
			if (run<OS<1024, 4, 10>>("1024x4, 10 bits", n, bits, kin, kout, vin, vout)) return 1;
			if (run<OS<1024, 12, 9>>("1024x12, 9 bits", n, bits, kin, kout, vin, vout)) return 1;
			if (run<OS<1024, 12, 10>>("1024x12, 10 bits", n, bits, kin, kout, vin, vout)) return 1;
			CK(hipFree(kin)); CK(hipFree(kout)); CK(hipFree(vin)); CK(hipFree(vout));
		}
	}
	return 0;
}
