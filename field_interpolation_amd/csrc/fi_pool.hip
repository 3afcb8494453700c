// Pool of device blocks between contexts (see DevBuf, fi_internal.h).  Per device, best fit: the smallest pooled block
// that holds the request and is at most twice as large.  Blocks enter only from fi_ctx_destroy (quiescent by a device
// synchronisation); the pool holds at most kPoolBytes per device, beyond that a block is freed as before.
#include "fi_internal.h"

#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <vector>

namespace fi {

thread_local bool pool_quiescent = false;
thread_local hipStream_t alloc_stream = nullptr;

namespace {
constexpr size_t kPoolBytes = size_t(8) << 30;
struct DevicePool {
	struct Block { void* p; size_t used; };
	std::multimap<size_t, Block> blocks;  // capacity -> block, and the bytes its last owner used
	size_t bytes = 0;
};
std::mutex g_mutex;
std::map<int, DevicePool> g_pools;
bool pool_off() { return test_switch("FI_NO_POOL") != nullptr; }
}  // namespace

void* pool_take(size_t capacity_wanted, size_t* capacity, size_t* used)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { return nullptr; }
	std::lock_guard<std::mutex> lock(g_mutex);
	auto pit = g_pools.find(dev);
	if (pit == g_pools.end()) { return nullptr; }
	DevicePool& P = pit->second;
	auto it = P.blocks.lower_bound(capacity_wanted);
	if (it == P.blocks.end() || it->first > 2 * capacity_wanted) { return nullptr; }
	void* p = it->second.p;
	*capacity = it->first;
	*used     = it->second.used;
	P.bytes -= it->first;
	P.blocks.erase(it);
	return p;
}

bool pool_give(void* p, size_t capacity, size_t used)
{
	if (!p || capacity == 0 || pool_off()) { return false; }
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { return false; }
	std::lock_guard<std::mutex> lock(g_mutex);
	DevicePool& P = g_pools[dev];
	if (P.bytes + capacity > kPoolBytes) { return false; }
	P.blocks.emplace(capacity, DevicePool::Block{p, used});
	P.bytes += capacity;
	return true;
}

// ---- streams and pinned staging buffers of destroyed contexts -------------------------------------------------------------
// hipStreamCreateWithFlags takes 0.4-6 ms, hipStreamDestroy 0.4-1.2 ms, hipHostMalloc / hipHostFree 0.1-0.25 ms (rocprofv3
// --hip-runtime-trace of tools/r5_cold.py): a context of the headline solver -- five contexts with nine streams and a dozen
// staging buffers between them -- paid 3-4 ms of its first assemble and 5-6 ms of its destruction for them.  A caller that
// creates, solves and destroys per call (the reference's stateless solve_sparse_linear*, sparse_linear.cpp:194-196) now finds
// the streams and buffers of the context before.  Streams enter drained (fi_ctx_destroy synchronises them first).
namespace {
struct HostPool {
	std::vector<hipStream_t> streams;
	std::multimap<size_t, void*> pinned;
};
std::map<int, HostPool> g_host;
constexpr size_t kMaxPooledStreams = 64, kMaxPooledPinned = 128;
}  // namespace

hipStream_t stream_take()
{
	int dev = 0;
	if (hipGetDevice(&dev) == hipSuccess && !pool_off()) {
		std::lock_guard<std::mutex> lock(g_mutex);
		HostPool& H = g_host[dev];
		if (!H.streams.empty()) {
			hipStream_t st = H.streams.back();
			H.streams.pop_back();
			return st;
		}
	}
	// FI_DUMMY_STREAMS=k (diagnostic): k streams created, used once and kept, in front of the first one the library makes --
	// shifts which of the library's streams share a hardware queue (profiles/r6_ablation.md: the assembly's chains)
	static bool shifted = false;
	if (!shifted) {
		shifted = true;
		if (const char* v = tuning_switch("FI_DUMMY_STREAMS")) {
			for (int i = 0; i < std::atoi(v); ++i) {
				hipStream_t d = nullptr;
				FI_HIP_TRY(hipStreamCreateWithFlags(&d, hipStreamNonBlocking));
				void* p = nullptr;
				FI_HIP_TRY(hipMalloc(&p, 256));
				FI_HIP_TRY(hipMemsetAsync(p, 0, 256, d));
				FI_HIP_TRY(hipStreamSynchronize(d));
				(void)hipFree(p);
			}
		}
	}
	hipStream_t st = nullptr;
	FI_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	return st;
}

void stream_give(hipStream_t st, bool drained)
{
	if (!st) { return; }
	int dev = 0;
	if (drained && !pool_off() && hipGetDevice(&dev) == hipSuccess) {
		std::lock_guard<std::mutex> lock(g_mutex);
		HostPool& H = g_host[dev];
		if (H.streams.size() < kMaxPooledStreams) {
			H.streams.push_back(st);
			return;
		}
	}
	(void)hipStreamDestroy(st);
}

void* pinned_take(size_t bytes, size_t* capacity)
{
	int dev = 0;
	if (hipGetDevice(&dev) == hipSuccess && !pool_off()) {
		std::lock_guard<std::mutex> lock(g_mutex);
		HostPool& H = g_host[dev];
		auto it = H.pinned.lower_bound(bytes);
		if (it != H.pinned.end() && it->first <= 4 * (bytes < 4096 ? 4096 : bytes)) {
			void* p = it->second;
			*capacity = it->first;
			H.pinned.erase(it);
			return p;
		}
	}
	size_t want = 4096;
	while (want < bytes) { want *= 2; }
	void* p = nullptr;
	FI_HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
	*capacity = want;
	return p;
}

void pinned_give(void* p, size_t capacity)
{
	if (!p) { return; }
	int dev = 0;
	if (!pool_off() && hipGetDevice(&dev) == hipSuccess) {
		std::lock_guard<std::mutex> lock(g_mutex);
		HostPool& H = g_host[dev];
		if (H.pinned.size() < kMaxPooledPinned) {
			H.pinned.emplace(capacity, p);
			return;
		}
	}
	(void)hipHostFree(p);
}

size_t pool_trim(size_t keep_bytes)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { return 0; }
	std::lock_guard<std::mutex> lock(g_mutex);
	if (keep_bytes == 0) {  // "give everything back": the streams and staging buffers too
		auto hit = g_host.find(dev);
		if (hit != g_host.end()) {
			for (hipStream_t st : hit->second.streams) { (void)hipStreamDestroy(st); }
			for (auto& kv : hit->second.pinned) { (void)hipHostFree(kv.second); }
			hit->second.streams.clear();
			hit->second.pinned.clear();
		}
	}
	auto pit = g_pools.find(dev);
	if (pit == g_pools.end()) { return 0; }
	DevicePool& P = pit->second;
	while (P.bytes > keep_bytes && !P.blocks.empty()) {
		auto it = std::prev(P.blocks.end());  // the largest first
		(void)hipFree(it->second.p);
		P.bytes -= it->first;
		P.blocks.erase(it);
	}
	return P.bytes;
}

}  // namespace fi
