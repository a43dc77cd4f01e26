// alloc.cpp -- process-wide cache of large freed device blocks (common.h: why).
#include <map>
#include <mutex>

#include "common.h"
#include "hprlp_amd.h"

namespace hprlp {

namespace {

struct Cache {
    std::mutex mu;
    std::multimap<std::pair<int, size_t>, void *> blocks;  // (device, capacity) -> block
    size_t bytes = 0;
    bool off = false;
    Cache() {
        const char *e = std::getenv("HPRLP_NO_ALLOC_CACHE");
        off = e && e[0] == '1';
    }
};

Cache &cache() {
    static Cache c;
    return c;
}

}  // namespace

void *device_cache_get(size_t bytes, size_t *capacity) {
    Cache &c = cache();
    if (c.off) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(c.mu);
    // smallest cached block of this device that holds the request and wastes at most a quarter of itself
    auto it = c.blocks.lower_bound({dev, bytes});
    if (it == c.blocks.end() || it->first.first != dev || it->first.second > bytes + bytes / 3) return nullptr;
    void *p = it->second;
    *capacity = it->first.second;
    c.bytes -= it->first.second;
    c.blocks.erase(it);
    return p;
}

bool device_cache_put(void *p, size_t capacity) {
    Cache &c = cache();
    if (c.off || !p) return false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    // hipFree waits for the device; code that frees a buffer right behind the kernels that used it relies on that
    if (hipDeviceSynchronize() != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(c.mu);
    if (c.bytes + capacity > kDeviceCacheCapBytes) return false;
    c.blocks.insert({{dev, capacity}, p});
    c.bytes += capacity;
    return true;
}

void device_cache_trim() {
    Cache &c = cache();
    std::lock_guard<std::mutex> lock(c.mu);
    for (auto &kv : c.blocks) (void)hipFree(kv.second);
    c.blocks.clear();
    c.bytes = 0;
}

}  // namespace hprlp

extern "C" void hprlp_release_device_cache(void) { hprlp::device_cache_trim(); }
