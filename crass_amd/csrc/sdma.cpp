// sdma.cpp — device -> pinned-host copies on the GPU's DMA engines, through the HSA runtime HIP itself sits on.
//
// Why not hipMemcpyAsync: on this platform the HIP runtime performs a device-to-host copy with a blit KERNEL
// (__amd_rocclr_copyBuffer in every trace; HSA_ENABLE_SDMA / GPU_FORCE_BLIT_COPY_SIZE do not change that), and 16 MB of
// PCIe stores issued from shader waves — the runtime's grid or a 32-workgroup kernel of ours, on one XCD or on all — cost
// the kernels of the OTHER stream the copy's whole duration (0.2-0.3 ms per step at 100 M reads, wherever the copy is placed:
// profiles/NOTES_r03.md).  The DMA engines have their own path to the link.
//
// The library is loaded with dlopen (it is already in the process: libamdhip64 depends on it), so nothing new is linked;
// every failure makes the caller fall back to the copy kernel.
#include "engine_internal.h"

#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace crass {

namespace {

struct Hsa {
    void *lib = nullptr;
    bool ok = false;
    decltype(&hsa_init) init = nullptr;
    decltype(&hsa_signal_create) signal_create = nullptr;
    decltype(&hsa_signal_destroy) signal_destroy = nullptr;
    decltype(&hsa_signal_store_relaxed) signal_store = nullptr;
    decltype(&hsa_signal_wait_scacquire) signal_wait = nullptr;
    decltype(&hsa_amd_pointer_info) pointer_info = nullptr;
    decltype(&hsa_amd_memory_async_copy) async_copy = nullptr;
};

Hsa &hsa()
{
    static Hsa h;
    static std::once_flag once;
    std::call_once(once, [] {
        if (getenv("CRASS_NO_SDMA")) return;                       // A/B switch
        h.lib = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h.lib) { fprintf(stderr, "[crass_sdma] libhsa-runtime64.so.1 not loadable: hand-off copies use the copy kernel\n"); return; }
        // (tested with ROCm 7.2.0 / HSA runtime 1.18: a release that drops or renames one of these says so once and the copy
        // kernel takes over — INTEGRATION.md, "DMA engine")
#define SYM(field, name) h.field = reinterpret_cast<decltype(h.field)>(dlsym(h.lib, #name)); \
        if (!h.field) { fprintf(stderr, "[crass_sdma] %s missing in the HSA runtime: hand-off copies use the copy kernel\n", #name); return; }
        SYM(init, hsa_init) SYM(signal_create, hsa_signal_create) SYM(signal_destroy, hsa_signal_destroy)
        SYM(signal_store, hsa_signal_store_relaxed) SYM(signal_wait, hsa_signal_wait_scacquire)
        SYM(pointer_info, hsa_amd_pointer_info) SYM(async_copy, hsa_amd_memory_async_copy)
#undef SYM
        if (h.init() != HSA_STATUS_SUCCESS) return;                // (reference counted: HIP holds the runtime open anyway)
        h.ok = true;
    });
    return h;
}

} // namespace

struct SdmaCopy {
    hsa_signal_t sig{};
    bool have_sig = false, pending = false, broken = false;
    const void *src = nullptr; void *dst = nullptr; size_t bytes = 0;      // the copy in flight (redone by the caller's runtime if the engine fails)
};

SdmaCopy *sdma_create()
{
    Hsa &h = hsa();
    if (!h.ok) return nullptr;
    SdmaCopy *s = new SdmaCopy();
    if (h.signal_create(0, 0, nullptr, &s->sig) != HSA_STATUS_SUCCESS) { delete s; return nullptr; }
    s->have_sig = true;
    // the engine's queue is created by the first copy (~6 ms): here, at context creation, not inside the first step
    void *d = nullptr, *hp = nullptr;
    if (hipMalloc(&d, 256) == hipSuccess && hipHostMalloc(&hp, 256, hipHostMallocDefault) == hipSuccess && hipMemset(d, 0, 256) == hipSuccess &&
        hipDeviceSynchronize() == hipSuccess && sdma_start(s, d, hp, 256))
        (void)sdma_wait(s);
    if (d) (void)hipFree(d);
    if (hp) (void)hipHostFree(hp);
    return s;
}

void sdma_destroy(SdmaCopy *s)
{
    if (!s) return;
    (void)sdma_wait(s);
    if (s->have_sig) (void)hsa().signal_destroy(s->sig);
    delete s;
}

// starts the copy at once: the caller has already waited for whatever produced d_src.  false: nothing was started.
bool sdma_start(SdmaCopy *s, const void *d_src, void *h_dst, size_t bytes)
{
    Hsa &h = hsa();
    if (!s || !h.ok || s->pending || s->broken || !bytes) return false;
    static const bool dbg = getenv("CRASS_SDMA_DEBUG") != nullptr;
    hsa_amd_pointer_info_t si{}, di{};
    si.size = sizeof(si); di.size = sizeof(di);
    if (h.pointer_info(const_cast<void *>(d_src), &si, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return false;
    if (h.pointer_info(h_dst, &di, nullptr, nullptr, nullptr) != HSA_STATUS_SUCCESS) return false;
    if (dbg) fprintf(stderr, "[crass_sdma] src %p type %d agent %llx base %p size %zu | dst %p type %d agent %llx base %p size %zu | %zu bytes\n", d_src, (int)si.type,
                     (unsigned long long)si.agentOwner.handle, si.agentBaseAddress, si.sizeInBytes, h_dst, (int)di.type, (unsigned long long)di.agentOwner.handle,
                     di.agentBaseAddress, di.sizeInBytes, bytes);
    // plain runtime allocations only: a range mapped with the virtual-memory API (CRASS_GUARD_PAGES, devmem.h) comes back as a
    // reserved address, and the copy call crashes on it
    if (si.type != HSA_EXT_POINTER_TYPE_HSA || (di.type != HSA_EXT_POINTER_TYPE_HSA && di.type != HSA_EXT_POINTER_TYPE_LOCKED)) {
        static bool told = false;
        if (!told && getenv("CRASS_GUARD_PAGES") == nullptr) { told = true; fprintf(stderr, "[crass_sdma] buffers not owned by the HSA runtime (types %d / %d): copies use the copy kernel\n", (int)si.type, (int)di.type); }
        return false;
    }
    h.signal_store(s->sig, 1);
    if (h.async_copy(h_dst, di.agentOwner, d_src, si.agentOwner, bytes, 0, nullptr, s->sig) != HSA_STATUS_SUCCESS) return false;
    s->pending = true;
    s->src = d_src; s->dst = h_dst; s->bytes = bytes;
    return true;
}

// 0: the records are in host memory (or nothing was pending); -1: the engine reported an error and the runtime's copy failed too
int sdma_wait(SdmaCopy *s)
{
    if (!s || !s->pending) return 0;
    Hsa &h = hsa();
    // bounded: a DMA engine that never signals must not hang seed_scan / recruit / destroy for ever.  The timeout is in
    // units of the system timestamp counter (>= 1 MHz everywhere: 2^26 ticks are at most about a minute, usually under a second
    // per slice); ten slices, then the engine counts as broken and the runtime copies the records instead.
    hsa_signal_value_t v = 1;
    for (int slice = 0; slice < 10 && v >= 1; slice++)
        v = h.signal_wait(s->sig, HSA_SIGNAL_CONDITION_LT, 1, 1ull << 26, HSA_WAIT_STATE_BLOCKED);
    s->pending = false;
    if (v >= 1) {
        fprintf(stderr, "[crass_sdma] the DMA engine did not signal a %zu-byte copy; falling back to the runtime's copy for this context\n", s->bytes);
        v = -1;
    }
    if (v < 0) {
        // the engine reported an error: the records are copied again by the HIP runtime, and this context stays off the engines
        s->broken = true;
        return hipMemcpy(s->dst, s->src, s->bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
    }
    return 0;
}

bool sdma_pending(const SdmaCopy *s) { return s && s->pending; }

} // namespace crass
