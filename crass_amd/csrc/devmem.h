// Device allocations of the library, and three debugging aids that cost one predictable branch when off:
//
//  * CRASS_GUARD_PAGES=1 — every allocation is mapped with the virtual-memory API between two UNMAPPED ranges, its end
//    flush (to 16 bytes) against the upper one: a kernel that reads or writes one element past a buffer takes a memory
//    access fault there and then, whatever else the heap holds.  (hipMalloc rounds sizes up and packs allocations, so an
//    overrun of a few bytes lands in mapped memory and shows up — if ever — as a fault that depends on the heap's history.)
//  * CRASS_POISON=1 — fresh allocations are filled with 0xA5 instead of zero (see dev_alloc).
//  * CRASS_TRACE_LAUNCHES=1 — every kernel launch prints its name to stderr first; with AMD_SERIALIZE_KERNEL=3 the last
//    line before a fault names the kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

namespace crass {

inline bool env_flag(const char *name)
{
    const char *v = getenv(name);
    return v && *v && *v != '0';
}

inline bool trace_launches()
{
    static const bool on = env_flag("CRASS_TRACE_LAUNCHES");
    return on;
}

struct GuardedAllocs {
    struct Rec { char *base; size_t reserved, mapped; hipMemGenericAllocationHandle_t h; int device; };
    std::mutex mu;
    std::unordered_map<void *, Rec> live;
    static GuardedAllocs &get() { static GuardedAllocs g; return g; }
    static bool enabled() { static const bool on = env_flag("CRASS_GUARD_PAGES"); return on; }

    hipError_t alloc(void **out, size_t bytes)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
        if (e != hipSuccess) return e;
        if (!gran) gran = 4096;
        static const size_t align = [] { const char *v = getenv("CRASS_GUARD_ALIGN"); const size_t a = v ? (size_t)atol(v) : 16; return a && !(a & (a - 1)) ? a : 16; }();
        const size_t want = (bytes + align - 1) & ~(align - 1);
        const size_t mapped = (want + gran - 1) / gran * gran;
        Rec r{nullptr, mapped + 2 * gran, mapped, {}, dev};
        void *va = nullptr;
        e = hipMemAddressReserve(&va, r.reserved, gran, nullptr, 0);
        if (e != hipSuccess) return e;
        r.base = (char *)va;
        e = hipMemCreate(&r.h, mapped, &prop, 0);
        if (e != hipSuccess) { (void)hipMemAddressFree(va, r.reserved); return e == hipErrorInvalidValue ? hipErrorOutOfMemory : e; }
        e = hipMemMap(r.base + gran, mapped, 0, r.h, 0);
        if (e == hipSuccess) {
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            e = hipMemSetAccess(r.base + gran, mapped, &acc, 1);
            if (e != hipSuccess) (void)hipMemUnmap(r.base + gran, mapped);
        }
        if (e != hipSuccess) { (void)hipMemRelease(r.h); (void)hipMemAddressFree(va, r.reserved); return e; }
        void *p = r.base + gran + (mapped - want);
        { std::lock_guard<std::mutex> lk(mu); live.emplace(p, r); }
        *out = p;
        return hipSuccess;
    }

    bool free(void *p)
    {
        Rec r;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = live.find(p);
            if (it == live.end()) return false;
            r = it->second;
            live.erase(it);
        }
        // on the ALLOCATION's device (a group frees buffers of several devices from one thread), restoring the caller's
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (cur != r.device) (void)hipSetDevice(r.device);
        (void)hipDeviceSynchronize();
        const size_t gran = (r.reserved - r.mapped) / 2;
        (void)hipMemUnmap(r.base + gran, r.mapped);
        (void)hipMemRelease(r.h);
        if (cur >= 0 && cur != r.device) (void)hipSetDevice(cur);
        // the address range stays reserved for the life of the process: no later buffer can land on it, so a kernel that
        // still holds the freed pointer faults instead of reading a stranger's data
        return true;
    }
};

// Every buffer the library allocates starts ZERO-FILLED.  hipMalloc promises nothing of the kind: a block recycled from an
// earlier hipFree keeps what it held, so code that (knowingly or not) relies on a fresh table, flag array or padding being
// zero would work in a young process and fail — or read a wild index — in an old one.  (Found with CRASS_POISON=1, which
// fills with 0xA5 instead: the diagnostic for "who depends on the zero fill".)  Allocation happens at load and when a pool
// grows, never on the per-step path, so the fill costs nothing that is measured.
inline hipError_t dev_alloc(void **p, size_t bytes)
{
    static const bool poison = env_flag("CRASS_POISON");
    const hipError_t e = GuardedAllocs::enabled() ? GuardedAllocs::get().alloc(p, bytes ? bytes : 1) : hipMalloc(p, bytes);
    if (e != hipSuccess || !bytes) return e;
    hipError_t f = hipMemsetAsync(*p, poison ? 0xA5 : 0, bytes, nullptr);
    if (f == hipSuccess) f = hipStreamSynchronize(nullptr);      // (the contexts' streams do not wait for the null stream)
    if (f != hipSuccess) { if (!(GuardedAllocs::enabled() && GuardedAllocs::get().free(*p))) (void)hipFree(*p); *p = nullptr; }
    return f;
}

inline void dev_free(void *p)
{
    if (!p) return;
    if (GuardedAllocs::enabled() && GuardedAllocs::get().free(p)) return;
    (void)hipFree(p);
}

} // namespace crass

// kernel launches of the library's .hip files go through this (same arguments as hipLaunchKernelGGL)
#define CRASS_LAUNCH(kernel, ...)                                                            \
    do {                                                                                     \
        if (crass::trace_launches()) fprintf(stderr, "[crass launch] %s\n", #kernel);        \
        hipLaunchKernelGGL(kernel, __VA_ARGS__);                                             \
    } while (0)
