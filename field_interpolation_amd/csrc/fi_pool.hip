// Pool of device blocks between contexts (see DevBuf, fi_internal.h).  Per device, best fit: the smallest pooled block
// that holds the request and is at most twice as large.  Blocks enter only from fi_ctx_destroy (quiescent by a device
// synchronisation); the pool holds at most kPoolBytes per device, beyond that a block is freed as before.
#include "fi_internal.h"

#include <hip/hip_runtime.h>

#include <map>
#include <mutex>

namespace fi {

thread_local bool pool_quiescent = false;

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

size_t pool_trim(size_t keep_bytes)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { return 0; }
	std::lock_guard<std::mutex> lock(g_mutex);
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
