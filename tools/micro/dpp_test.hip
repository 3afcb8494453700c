// micro test: which lane does DPP wave_shl:1 / wave_shr:1 read from on gfx950, and what do inactive source lanes give
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out)
{
	const int lane = threadIdx.x;
	int v = 100 + lane;
	out[lane]       = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, true);   // wave_shl:1
	out[64 + lane]  = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, true);   // wave_shr:1
	int w = -7;
	if (lane % 3 != 1) { w = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, true); }  // sources partly inactive
	out[128 + lane] = w;
	out[192 + lane] = __shfl_down(v, 1, 64);
}
int main()
{
	int* d;
	hipMalloc(&d, 256 * 4);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
	int h[256];
	hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
	for (int r = 0; r < 4; ++r) {
		printf("row %d:", r);
		for (int i = 0; i < 64; ++i) printf(" %d", h[r * 64 + i]);
		printf("\n");
	}
	return 0;
}
