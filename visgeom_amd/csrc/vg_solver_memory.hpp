// vg_solver_memory.hpp -- the memory of one LM solve: one device block and one pinned block per solve, kept by the library for
// the next one (vg_release_cached_memory), typed views into them (DevBuf / PinnedBuf) and the staged upload of every index
// table in ONE copy.  Part of the solver translation unit (vg_solver_tu.hip); see vg_solver_impl.hpp for the driver.
#pragma once

#include <mutex>
#include <vector>

#include "vg_internal.hpp"

namespace {

// ---- memory of one solve -----------------------------------------------------------------------------------
// A solve needs ~40 device buffers, a dozen small uploads and a few pinned read-back buffers.  One hipMalloc /
// hipMemcpy / hipHostMalloc each made the set-up 0.45 ms of a 1.4 ms solve (10 k EUCM images), so they come out of ONE
// device allocation and ONE pinned allocation per solve: region U holds the uploads (staged at the same offsets of the
// pinned block, ONE asynchronous copy at flush()), the rest is bump-allocated scratch.  Both blocks are kept for the
// next solve of the process (vg_release_cached_memory frees them); anything that does not fit falls back to its own
// allocation.
struct SolveArena {
    char *dev = nullptr, *pin = nullptr;
    size_t dev_cap = 0, pin_cap = 0;
    size_t up_cap = 0, up_used = 0;   // [0, up_cap) of both blocks: uploads
    size_t dev_used = 0, pin_used = 0;  // bump pointers behind the upload region
    int device = -1;
    bool flushed = false;
    static size_t align(size_t n) { return (n + 255) & ~(size_t)255; }
    void *dev_alloc(size_t bytes)
    {
        const size_t o = align(dev_used);
        if (!dev || o + bytes > dev_cap) return nullptr;
        dev_used = o + bytes;
        return dev + o;
    }
    void *pin_alloc(size_t bytes)
    {
        const size_t o = align(pin_used);
        if (!pin || o + bytes > pin_cap) return nullptr;
        pin_used = o + bytes;
        return pin + o;
    }
    // device address of an upload of `bytes`, its bytes staged for flush(); NULL when the region is full / already flushed
    void *upload(const void *src, size_t bytes)
    {
        const size_t o = align(up_used);
        if (!dev || !pin || flushed || o + bytes > up_cap) return nullptr;
        std::memcpy(pin + o, src, bytes);
        up_used = o + bytes;
        return dev + o;
    }
    int flush(hipStream_t st)
    {
        if (!flushed && up_used) VG_HIP(hipMemcpyAsync(dev, pin, up_used, hipMemcpyHostToDevice, st));
        flushed = true;
        return VG_OK;
    }
};

struct ArenaCache {
    std::mutex m;
    char *dev = nullptr, *pin = nullptr;
    size_t dev_cap = 0, pin_cap = 0;
    int device = -1;
    void drop()
    {
        // (hipFree takes the pointer's own device: the calling thread's current device is not touched -- a torch caller relies
        //  on its own)
        if (dev) (void)hipFree(dev);
        if (pin) (void)hipHostFree(pin);
        dev = pin = nullptr;
        dev_cap = pin_cap = 0;
    }
};
ArenaCache g_arena_cache;
thread_local SolveArena *t_arena = nullptr;

// takes the cached blocks when they are large enough, allocates otherwise; the destructor hands the blocks back
struct ArenaScope {
    SolveArena a;
    bool discard = false;  // set when the solve's stream could not be drained: the blocks are freed, never cached
    ArenaScope(int device, size_t dev_need, size_t pin_need, size_t up_cap)
    {
        {
            std::lock_guard<std::mutex> lk(g_arena_cache.m);
            if (g_arena_cache.dev && g_arena_cache.device == device && g_arena_cache.dev_cap >= dev_need && g_arena_cache.pin_cap >= pin_need) {
                a.dev = g_arena_cache.dev;
                a.pin = g_arena_cache.pin;
                a.dev_cap = g_arena_cache.dev_cap;
                a.pin_cap = g_arena_cache.pin_cap;
                g_arena_cache.dev = g_arena_cache.pin = nullptr;
                g_arena_cache.dev_cap = g_arena_cache.pin_cap = 0;
            }
        }
        if (!a.dev) {
            if (hipMalloc(reinterpret_cast<void **>(&a.dev), dev_need) != hipSuccess) a.dev = nullptr;
            if (a.dev && hipHostMalloc(reinterpret_cast<void **>(&a.pin), pin_need, hipHostMallocCoherent) != hipSuccess) {
                (void)hipFree(a.dev);
                a.dev = a.pin = nullptr;
            }
            (void)hipGetLastError();  // a failed block only means: every buffer takes its own allocation
            a.dev_cap = a.dev ? dev_need : 0;
            a.pin_cap = a.pin ? pin_need : 0;
        }
        a.device = device;
        a.up_cap = up_cap < a.dev_cap && up_cap < a.pin_cap ? up_cap : 0;
        a.dev_used = a.pin_used = a.up_cap;
        t_arena = &a;
    }
    ~ArenaScope()
    {
        t_arena = nullptr;
        if (!a.dev) return;
        std::lock_guard<std::mutex> lk(g_arena_cache.m);
        // keep the pair that serves more solves: a hit needs BOTH capacities, so a pair replaces the cached one only when it
        // is at least as large in both (or nothing is cached)
        if (!discard && a.dev_cap >= g_arena_cache.dev_cap && a.pin_cap >= g_arena_cache.pin_cap) {
            g_arena_cache.drop();
            g_arena_cache.dev = a.dev;
            g_arena_cache.pin = a.pin;
            g_arena_cache.dev_cap = a.dev_cap;
            g_arena_cache.pin_cap = a.pin_cap;
            g_arena_cache.device = a.device;
        } else {
            (void)hipFree(a.dev);
            (void)hipHostFree(a.pin);
        }
    }
    ArenaScope(const ArenaScope &) = delete;
    ArenaScope &operator=(const ArenaScope &) = delete;
};

template <typename T>
struct DevBuf {
    T *p = nullptr;
    bool owned = false;
    ~DevBuf()
    {
        if (p && owned) (void)hipFree(p);
    }
    int alloc(size_t n)
    {
        const size_t bytes = sizeof(T) * (n ? n : 1);
        if (t_arena && (p = static_cast<T *>(t_arena->dev_alloc(bytes))) != nullptr) return VG_OK;
        VG_HIP(hipMalloc(&p, bytes));
        owned = true;
        return VG_OK;
    }
    int upload(const std::vector<T> &h)
    {
        if (t_arena && !h.empty() && (p = static_cast<T *>(t_arena->upload(h.data(), sizeof(T) * h.size()))) != nullptr) return VG_OK;
        int rc = alloc(h.size());
        if (rc != VG_OK) return rc;
        if (!h.empty()) VG_HIP(hipMemcpy(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
        return VG_OK;
    }
};

// pinned host memory: small per-iteration transfers go straight over DMA instead of through a staging copy kernel
struct PinnedBuf {
    double *p = nullptr;
    size_t n = 0;
    bool owned = false;
    ~PinnedBuf()
    {
        if (p && owned) (void)hipHostFree(p);
    }
    int alloc(size_t count)
    {
        n = count;
        const size_t bytes = sizeof(double) * (count ? count : 1);
        if (t_arena && (p = static_cast<double *>(t_arena->pin_alloc(bytes))) != nullptr) return VG_OK;
        VG_HIP(hipHostMalloc(reinterpret_cast<void **>(&p), bytes, hipHostMallocCoherent));   // the host spins on words kernels store here
        owned = true;
        return VG_OK;
    }
};

}  // namespace
