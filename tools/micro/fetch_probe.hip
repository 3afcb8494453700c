// FETCH_SIZE / WRITE_SIZE calibration: streams of known size read with 2, 4, 8 and 16 bytes per lane (and written with 8 / 16),
// one kernel per width, to be run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes):
// what factor turns the counter (KB) into bytes for each access width on gfx950?  (profiles/r5_ablation.md section 25)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename V>
__global__ __launch_bounds__(256) void k_read(const V* __restrict__ src, size_t n, float* __restrict__ sink)
{
	float acc = 0.0f;
	for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256) {
		const V v = src[i];
		const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
		acc += static_cast<float>(b[0]);
	}
	if (acc == 12345.678f) { sink[0] = acc; }
}
template <typename V>
__global__ __launch_bounds__(256) void k_write(V* __restrict__ dst, size_t n)
{
	V v{};
	for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * 256) { dst[i] = v; }
}

int main()
{
	const size_t bytes = size_t(1) << 28;  // 256 MiB per stream
	void* buf = nullptr;
	float* sink = nullptr;
	hipMalloc(&buf, bytes);
	hipMalloc(&sink, 4);
	hipMemset(buf, 1, bytes);
	hipDeviceSynchronize();
	for (int rep = 0; rep < 2; ++rep) {
		hipLaunchKernelGGL(k_read<unsigned short>, dim3(8192), dim3(256), 0, 0, static_cast<const unsigned short*>(buf), bytes / 2, sink);
		hipLaunchKernelGGL(k_read<uint32_t>, dim3(8192), dim3(256), 0, 0, static_cast<const uint32_t*>(buf), bytes / 4, sink);
		hipLaunchKernelGGL(k_read<uint2>, dim3(8192), dim3(256), 0, 0, static_cast<const uint2*>(buf), bytes / 8, sink);
		hipLaunchKernelGGL(k_read<uint4>, dim3(8192), dim3(256), 0, 0, static_cast<const uint4*>(buf), bytes / 16, sink);
		hipLaunchKernelGGL(k_write<uint2>, dim3(8192), dim3(256), 0, 0, static_cast<uint2*>(buf), bytes / 8);
		hipLaunchKernelGGL(k_write<uint4>, dim3(8192), dim3(256), 0, 0, static_cast<uint4*>(buf), bytes / 16);
		hipDeviceSynchronize();
	}
	printf("streams of %zu bytes\n", bytes);
	return 0;
}
