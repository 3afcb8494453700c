// tools/micro/graph_chain.hip -- what does a hipGraph replay buy over stream launches for a chain of DEPENDENT small kernels?
// (round 5, VERDICT r4 task 1a).  A chain of N kernels, each a 7-point average over an L2-resident array of `n` floats
// (ping-pong), `wgs` workgroups of 256 threads; timed per kernel for
//   (a) N stream launches back to back,
//   (b) the same chain captured once and replayed as a graph,
//   (c) ONE kernel of one workgroup doing all N steps with __syncthreads between them (the "single-workgroup tail"), n <= 4096.
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/graph_chain.hip -o tools/micro/graph_chain
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                             \
	do {                                                                                     \
		hipError_t e_ = (x);                                                                 \
		if (e_ != hipSuccess) {                                                              \
			fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
			exit(1);                                                                         \
		}                                                                                    \
	} while (0)

__global__ __launch_bounds__(256) void k_step(int n, const float* __restrict__ in, float* __restrict__ out)
{
	for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
		const int a = i >= 2 ? i - 2 : i, b = i >= 1 ? i - 1 : i, c = i + 1 < n ? i + 1 : i, d = i + 2 < n ? i + 2 : i;
		out[i] = 0.2f * (in[a] + in[b] + in[i] + in[c] + in[d]);
	}
}

// one workgroup, all steps: the vectors in LDS, a barrier per step
__global__ __launch_bounds__(1024) void k_tail(int n, int steps, const float* __restrict__ in, float* __restrict__ out)
{
	__shared__ float a[4096], b[4096];
	for (int i = threadIdx.x; i < n; i += 1024) { a[i] = in[i]; }
	__syncthreads();
	float* src = a;
	float* dst = b;
	for (int s = 0; s < steps; ++s) {
		for (int i = threadIdx.x; i < n; i += 1024) {
			const int p = i >= 2 ? i - 2 : i, q = i >= 1 ? i - 1 : i, c = i + 1 < n ? i + 1 : i, d = i + 2 < n ? i + 2 : i;
			dst[i] = 0.2f * (src[p] + src[q] + src[i] + src[c] + src[d]);
		}
		__syncthreads();
		float* t = src;
		src = dst;
		dst = t;
	}
	for (int i = threadIdx.x; i < n; i += 1024) { out[i] = src[i]; }
}

static double now_us()
{
	return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
	const int N = argc > 1 ? atoi(argv[1]) : 60;
	hipStream_t st;
	CHECK(hipStreamCreate(&st));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	const int sizes[] = {4096, 32768, 262144, 2097152};
	const int wgss[]  = {4, 32, 256, 1024};
	printf("chain of %d dependent kernels; per-kernel time in us (GPU events / host wall incl. sync)\n", N);
	for (int t = 0; t < 4; ++t) {
		const int n = sizes[t], wgs = wgss[t];
		float *a, *b;
		CHECK(hipMalloc(&a, sizeof(float) * n));
		CHECK(hipMalloc(&b, sizeof(float) * n));
		CHECK(hipMemset(a, 0, sizeof(float) * n));
		CHECK(hipMemset(b, 0, sizeof(float) * n));
		auto chain = [&]() {
			for (int k = 0; k < N; ++k) {
				hipLaunchKernelGGL(k_step, dim3(wgs), dim3(256), 0, st, n, (k & 1) ? b : a, (k & 1) ? a : b);
			}
		};
		// (a) stream launches
		for (int w = 0; w < 3; ++w) { chain(); }
		CHECK(hipStreamSynchronize(st));
		const int reps = 20;
		double t0 = now_us();
		CHECK(hipEventRecord(e0, st));
		for (int r = 0; r < reps; ++r) { chain(); }
		CHECK(hipEventRecord(e1, st));
		CHECK(hipStreamSynchronize(st));
		double t1 = now_us();
		float ms = 0;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		const double stream_gpu = 1e3 * ms / (reps * N), stream_wall = (t1 - t0) / (reps * N);
		// (b) graph replay
		hipGraph_t g;
		hipGraphExec_t ge;
		CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
		chain();
		CHECK(hipStreamEndCapture(st, &g));
		double ti0 = now_us();
		CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
		double ti1 = now_us();
		for (int w = 0; w < 3; ++w) { CHECK(hipGraphLaunch(ge, st)); }
		CHECK(hipStreamSynchronize(st));
		t0 = now_us();
		CHECK(hipEventRecord(e0, st));
		for (int r = 0; r < reps; ++r) { CHECK(hipGraphLaunch(ge, st)); }
		CHECK(hipEventRecord(e1, st));
		CHECK(hipStreamSynchronize(st));
		t1 = now_us();
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		const double graph_gpu = 1e3 * ms / (reps * N), graph_wall = (t1 - t0) / (reps * N);
		// one graph launch alone, host time of the call and time to completion
		t0 = now_us();
		CHECK(hipGraphLaunch(ge, st));
		double tc = now_us();
		CHECK(hipStreamSynchronize(st));
		t1 = now_us();
		printf("n=%8d wgs=%5d  stream %6.2f / %6.2f   graph %6.2f / %6.2f   (instantiate %.0f us; one replay: call %.0f us, done after %.0f us = %.2f per node)\n",
		       n, wgs, stream_gpu, stream_wall, graph_gpu, graph_wall, ti1 - ti0, tc - t0, t1 - t0, (t1 - t0) / N);
		if (n <= 4096) {
			for (int w = 0; w < 3; ++w) { hipLaunchKernelGGL(k_tail, dim3(1), dim3(1024), 0, st, n, N, a, b); }
			CHECK(hipStreamSynchronize(st));
			CHECK(hipEventRecord(e0, st));
			for (int r = 0; r < reps; ++r) { hipLaunchKernelGGL(k_tail, dim3(1), dim3(1024), 0, st, n, N, a, b); }
			CHECK(hipEventRecord(e1, st));
			CHECK(hipStreamSynchronize(st));
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			printf("           one workgroup, %d steps in LDS: %.2f us per launch = %.3f us per step\n", N, 1e3 * ms / reps, 1e3 * ms / (reps * N));
		}
		CHECK(hipGraphExecDestroy(ge));
		CHECK(hipGraphDestroy(g));
		CHECK(hipFree(a));
		CHECK(hipFree(b));
	}
	return 0;
}
