// alloc.cpp -- process-wide cache of large freed device blocks (common.h: why).
//
// A block belongs to the device it was allocated on, whatever device the freeing thread has current: put() asks the
// runtime for the owner (hipPointerGetAttributes), waits for THAT device and files the block under it; get() serves the
// calling thread's current device, like hipMalloc.  The cap is per device and bounded by a third of the device's memory.
#include <map>
#include <mutex>

#include "common.h"
#include "env.h"
#include "hprlp_amd.h"

namespace hprlp {

namespace {

// The bookkeeping, free of HIP calls (hprlp_alloc_cache_selftest exercises it without a GPU).
struct BlockIndex {
    std::multimap<std::pair<int, size_t>, void *> blocks;  // (owning device, capacity) -> block
    std::map<int, size_t> bytes;                           // per device
    // smallest cached block of `dev` that holds the request and wastes at most a quarter of itself
    void *take(int dev, size_t want, size_t *capacity) {
        auto it = blocks.lower_bound({dev, want});
        if (it == blocks.end() || it->first.first != dev || it->first.second > want + want / 3) return nullptr;
        void *p = it->second;
        *capacity = it->first.second;
        bytes[dev] -= it->first.second;
        blocks.erase(it);
        return p;
    }
    bool file(int dev, size_t capacity, void *p, size_t cap_per_device) {
        if (bytes[dev] + capacity > cap_per_device) return false;
        blocks.insert({{dev, capacity}, p});
        bytes[dev] += capacity;
        return true;
    }
    size_t held(int dev) const {
        auto it = bytes.find(dev);
        return it == bytes.end() ? 0 : it->second;
    }
};

struct Cache {
    std::mutex mu;
    BlockIndex idx;
    std::map<int, size_t> cap;  // per device: min(kDeviceCacheCapBytes, total memory / 3)
    bool off = false;
    Cache() {
        const char *e = env_get("HPRLP_NO_ALLOC_CACHE");
        off = e && e[0] == '1';
    }
};

Cache &cache() {
    static Cache c;
    return c;
}

// RAII: make `dev` current, restore the caller's device on exit
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope() {
        if (switched) (void)hipSetDevice(prev);
    }
};

}  // namespace

void *device_cache_get(size_t bytes, size_t *capacity) {
    Cache &c = cache();
    if (c.off) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(c.mu);
    return c.idx.take(dev, bytes, capacity);
}

bool device_cache_put(void *p, size_t capacity) {
    Cache &c = cache();
    if (c.off || !p) return false;
    hipPointerAttribute_t attr{};
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    const int owner = attr.device;
    DeviceScope scope(owner);
    // hipFree waits for the device; code that frees a buffer right behind the kernels that used it relies on that
    if (hipDeviceSynchronize() != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(c.mu);
    auto it = c.cap.find(owner);
    if (it == c.cap.end()) {
        size_t free_b = 0, total_b = 0;
        size_t cap = kDeviceCacheCapBytes;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) cap = std::min(cap, total_b / 3);
        it = c.cap.emplace(owner, cap).first;
    }
    return c.idx.file(owner, capacity, p, it->second);
}

void device_cache_trim() {
    Cache &c = cache();
    std::lock_guard<std::mutex> lock(c.mu);
    for (auto &kv : c.idx.blocks) {
        DeviceScope scope(kv.first.first);
        (void)hipFree(kv.second);
    }
    c.idx.blocks.clear();
    c.idx.bytes.clear();
}

void *device_malloc_or_trim(size_t bytes) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipErrorOutOfMemory) {
        // the cache may be sitting on the memory this request needs (blocks of other size classes): give it back, retry once
        (void)hipGetLastError();
        device_cache_trim();
        e = hipMalloc(&p, bytes);
    }
    if (e != hipSuccess)
        throw std::runtime_error(std::string("HIP error ") + hipGetErrorString(e) + " in hipMalloc of " + std::to_string(bytes) + " bytes");
    return p;
}

}  // namespace hprlp

extern "C" void hprlp_release_device_cache(void) { hprlp::device_cache_trim(); }

// Bookkeeping self-test without a GPU (tests/test_abi.py): blocks are filed under their OWNING device and only handed to
// requests of that device; the cap is per device.  Returns 0 on success, the number of the failed check otherwise.
extern "C" int hprlp_alloc_cache_selftest(void) {
    using hprlp::BlockIndex;
    BlockIndex ix;
    char a, b, c3, d;
    const size_t MiB = size_t(1) << 20, cap = 64 * MiB;
    size_t got = 0;
    if (!ix.file(0, 8 * MiB, &a, cap)) return 1;
    if (!ix.file(1, 8 * MiB, &b, cap)) return 2;
    if (ix.take(2, 8 * MiB, &got) != nullptr) return 3;             // nothing of device 2
    if (ix.take(1, 8 * MiB, &got) != &b || got != 8 * MiB) return 4;  // device 1 gets ITS block, not device 0's
    if (ix.take(1, 8 * MiB, &got) != nullptr) return 5;             // and never device 0's
    if (ix.held(0) != 8 * MiB || ix.held(1) != 0) return 6;
    if (!ix.file(0, 40 * MiB, &c3, cap)) return 7;
    if (ix.file(0, 32 * MiB, &d, cap)) return 8;                    // over device 0's cap ...
    if (!ix.file(1, 32 * MiB, &d, cap)) return 9;                   // ... which does not count against device 1
    if (ix.take(0, 6 * MiB, &got) != &a) return 10;                 // 8 MiB block serves 6 MiB (wastes a quarter)
    if (ix.take(0, 20 * MiB, &got) != nullptr) return 11;           // the 40 MiB block would waste half of itself
    if (ix.take(0, 32 * MiB, &got) != &c3 || got != 40 * MiB) return 12;
    return 0;
}
