// engine.cpp — the context behind include/crass_hip.h: device residency of the packed reads,
// kernel sequencing on one HIP stream, and the host-side sink that turns device records into
// the ordered hand-off (candidates, tokens, groups, patterns, recruits).
//
// There is no CPU fallback anywhere in this file: every search decision is made by the HIP
// kernels in kernels.hip; the host only orders, tokenises and clusters their output
// ("stays on host").
#include "../../include/crass_hip.h"
#include "engine_internal.h"
#include "merge.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <memory>
#include <utility>
#include <vector>

using namespace crass;

namespace {

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// One helper thread per context: rebuilds the host view of a device-side merge while the calling thread keeps
// the device fed (pass 2).  submit() hands over one job; wait() returns when it has finished.
class HostWorker {
public:
    ~HostWorker() { stop(); }
    void submit(std::function<void()> job)
    {
        wait();
        {
            std::lock_guard<std::mutex> lk(m_);
            if (!started_) { th_ = std::thread([this] { loop(); }); started_ = true; }
            job_ = std::move(job); busy_ = true;
        }
        cv_.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [this] { return !busy_; });
    }
    // The caller is about to wait for the job: if the helper has not even picked it up yet (its wake-up is normally
    // ~50 us, but on a busy or CPU-limited host it was seen to take 4-14 ms — every slow step of tools/step_spikes.py was
    // this wait), the caller runs the job itself instead.  Returns false when the helper has it (then: wait()).
    bool run_here_if_not_started()
    {
        std::function<void()> job;
        {
            std::lock_guard<std::mutex> lk(m_);
            if (!(busy_ && job_)) return false;
            job = std::move(job_); job_ = nullptr;          // the helper's wait predicate (busy_ && job_) stays false: it sleeps on
        }
        try { job(); } catch (...) { failed_.store(true, std::memory_order_relaxed); }      // (the job records its own status; this only keeps busy_ honest)
        { std::lock_guard<std::mutex> lk(m_); busy_ = false; }
        cv_.notify_all();
        return true;
    }
    bool take_failed() { return failed_.exchange(false, std::memory_order_relaxed); }
    void stop()
    {
        if (!started_) return;
        wait();
        { std::lock_guard<std::mutex> lk(m_); quit_ = true; }
        cv_.notify_all();
        th_.join();
        started_ = false;
    }
private:
    void loop()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return quit_ || (busy_ && job_); });
                if (quit_) return;
                job = std::move(job_); job_ = nullptr;
            }
            try { job(); } catch (...) { failed_.store(true, std::memory_order_relaxed); }
            { std::lock_guard<std::mutex> lk(m_); busy_ = false; }
            cv_.notify_all();
        }
    }
    std::atomic<bool> failed_{false};
    std::thread th_;
    std::mutex m_;
    std::condition_variable cv_;
    std::function<void()> job_;
    bool busy_ = false, quit_ = false, started_ = false;
};

// std::allocator whose resize() leaves trivially constructible elements uninitialised
template <typename T> struct NoInitAlloc : std::allocator<T> {
    template <typename U> struct rebind { typedef NoInitAlloc<U> other; };
    NoInitAlloc() = default;
    template <typename U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <typename U, typename... A> void construct(U *p, A &&...a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
        else ::new ((void *)p) U(std::forward<A>(a)...);
    }
};

// device allocations of this library, and an optional cap on them (tests: CRASS_POOL_CAP_MB makes an allocation beyond the
// cap fail with hipErrorOutOfMemory, so the error paths can be exercised without exhausting a 288 GB device)
std::atomic<uint64_t> g_dev_bytes{0};
std::atomic<uint64_t> g_dev_cap{0};

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t ensure(size_t want)
    {
        if (want <= n && p) return hipSuccess;
        release();
        if (want == 0) want = 1;
        const uint64_t cap = g_dev_cap.load(std::memory_order_relaxed);
        if (cap && g_dev_bytes.load(std::memory_order_relaxed) + want * sizeof(T) > cap) return hipErrorOutOfMemory;
        hipError_t e = crass::dev_alloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) { n = want; g_dev_bytes.fetch_add(want * sizeof(T), std::memory_order_relaxed); }
        else p = nullptr;
        return e;
    }
    void release()
    {
        if (p) { crass::dev_free(p); g_dev_bytes.fetch_sub(n * sizeof(T), std::memory_order_relaxed); }
        p = nullptr; n = 0;
    }
};

template <typename T> struct PinBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t ensure(size_t want)
    {
        if (want <= n && p) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; n = 0; }
        if (want == 0) want = 1;
        hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) n = want;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
};

} // namespace

// A/B and test switches from the environment, read ONCE per context (crass_hip_create; crass_hip_reload_env re-reads
// them for a live context) — never on the per-call path
struct EnvSwitches {
    bool no_lookback = false, no_pos_hints = false, merge_profile = false, no_lane_kernel = false;
    bool host_merge = false, no_speculation = false, exc_separate = false, dm_inject_fail = false, dm_init_late = false;
    bool no_presize = false;                         // A/B switch: no first-call bounds / pool sizing at crass_hip_load_reads
    bool dd_full_table = false;                      // A/B switch: pass 1's de-duplication table sized for the survivor slots (the round-3 size)
    bool no_device_view = false;                     // A/B switch: the host rebuilds crass_merge_view from root_of / blank (the round-3 path)
    bool force_device_view = false;                  // CRASS_DEVICE_VIEW=1: the device assembles it for a single context too (default: multi-rank only)
    bool no_dense_light = false;                     // CRASS_NO_DENSE_LIGHT: the A/B switch of the light walk over the dense path's survivors
    bool no_warm_launch = false;                     // CRASS_NO_WARM_LAUNCH: the A/B switch of first_call_bounds' empty launch
    uint32_t view_group_cap = 32768;                 // groups beyond this many members are ranked by the host (CRASS_VIEW_GROUP_CAP)
    uint32_t view_sort_max = 2048;                   // groups of 65 .. this many members are ranked by a sort in LDS (k_dmx_sort; CRASS_VIEW_SORT_MAX: tests)
    uint64_t test_bounds[4] = {0, 0, 0, 0};          // tests: CRASS_TEST_BOUNDS="survivors,distinct,flagged,gathered" replaces the
                                                     // first-call bounds (0 = computed), so that every overflow path can be forced
    uint32_t row_cap = 256, dm_group_cap = 16384, surv_debug = 0;     // row_cap: Levenshtein fallback rows of the long-read layout (strings beyond it: second launch, full rows)
    int stage_timing = -1;
    uint64_t pool_cap_bytes = 0;                     // tests: device allocations beyond this total fail with hipErrorOutOfMemory
    bool no_hint_filter = false;                     // A/B switch: k_filter_general where the hint bits would be the filter
    uint32_t wave_walk_min = 800;                    // pass 2 walks a read per wave when the longest read is beyond this (CRASS_WAVE_WALK_MIN)
    uint32_t long_min = 2048;                        // read sets whose longest read is beyond this take the long-read path (CRASS_LONG_MIN: the A/B switch)
    void read()
    {
        auto on = [](const char *n) { return getenv(n) != nullptr; };
        no_lookback = on("CRASS_NO_LOOKBACK"); no_pos_hints = on("CRASS_NO_POS_HINTS");
        merge_profile = on("CRASS_MERGE_PROFILE"); no_lane_kernel = on("CRASS_NO_LANE_KERNEL"); host_merge = on("CRASS_HOST_MERGE");
        no_speculation = on("CRASS_NO_SPECULATION"); exc_separate = on("CRASS_EXC_SEPARATE"); dm_init_late = on("CRASS_DM_INIT_LATE");
        dm_inject_fail = on("CRASS_DM_INJECT_FAIL"); no_presize = on("CRASS_NO_PRESIZE"); no_device_view = on("CRASS_NO_DEVICE_VIEW"); dd_full_table = on("CRASS_DD_FULL_TABLE"); force_device_view = on("CRASS_DEVICE_VIEW");
        view_group_cap = 32768; if (const char *e = getenv("CRASS_VIEW_GROUP_CAP")) view_group_cap = (uint32_t)std::max(1, atoi(e));
        view_sort_max = 2048; if (const char *e = getenv("CRASS_VIEW_SORT_MAX")) view_sort_max = (uint32_t)std::min(2048, std::max(64, atoi(e)));
        row_cap = 256; if (const char *e = getenv("CRASS_ROW_CAP")) row_cap = (uint32_t)std::max(1, atoi(e));
        dm_group_cap = 16384; if (const char *e = getenv("CRASS_DM_GROUP_CAP")) dm_group_cap = (uint32_t)std::max(1, atoi(e));
        surv_debug = 0; if (const char *e = getenv("CRASS_SURV_DEBUG")) surv_debug = (uint32_t)atoi(e);
        stage_timing = -1; if (const char *e = getenv("CRASS_STAGE_TIMING")) stage_timing = std::min(2, std::max(0, atoi(e)));
        for (auto &b : test_bounds) b = 0;
        if (const char *e = getenv("CRASS_TEST_BOUNDS")) { int k = 0; for (const char *q = e; *q && k < 4; k++) { test_bounds[k] = (uint64_t)atoll(q); while (*q && *q != ',') q++; if (*q == ',') q++; } }
        no_hint_filter = on("CRASS_NO_HINT_FILTER");
        wave_walk_min = 800; if (const char *e = getenv("CRASS_WAVE_WALK_MIN")) wave_walk_min = (uint32_t)std::max(0, atoi(e));
        long_min = 2048; if (const char *e = getenv("CRASS_LONG_MIN")) long_min = (uint32_t)std::max(0, atoi(e));
        no_warm_launch = getenv("CRASS_NO_WARM_LAUNCH") != nullptr;
        no_dense_light = getenv("CRASS_NO_DENSE_LIGHT") != nullptr;
        pool_cap_bytes = 0; if (const char *e = getenv("CRASS_POOL_CAP_MB")) pool_cap_bytes = (uint64_t)std::max(1ll, atoll(e)) << 20;
    }
};

// ---- the context's state, in one place -----------------------------------------------------------------------------
// Results:      have_reads -> have_pass1 -> have_merge -> have_pass2 (each API call clears everything to its right and
//               refuses with CRASS_ERR_STATE if what it needs on its left is missing).  dense.active / q_blob_active say
//               which representation holds pass 1's / pass 2's records (compact pinned blob vs host vectors).
// Speculation:  five bounds learnt from the PREVIOUS call of the same stage, each 1.5 x the count seen then, each used to
//               size and queue the next stage before the host has seen the current count.  Every one of them has the same
//               three-step protocol: (1) the kernels read the real count on the device and never touch a slot beyond the
//               bound, (2) the real count reaches the host with the stage's final synchronisation, (3) count > bound =>
//               nothing of the speculative launch is used, the bound is dropped (set to 0) and the stage is repeated with
//               the exact count.  None is ever trusted for a result.
//                 surv_cap_hint     survivors of the seed scan        -> survivor kernels queued behind the compaction
//                 dx_cap_hint       distinct DR strings (one GPU)     -> the device merge queued behind pass 1 (premerge)
//                 xchg.gx_cap_hint  distinct DR strings (all ranks)   -> the same behind the exchange's de-duplication
//                 hit_cap_hint      reads flagged by the anchor probe -> verify / finish / pack queued behind it
//                 (recruit_exact    one-shot: the next recruit call must not speculate, it repeats an overflowed one)
// Queued merge: premerge 0 none / 2 queued by the seed scan and valid (count within the bound), premerge_inflight: its
//               kernels may still be running when the seed scan returns; dm_prepared_n / dm_prepared_src: the survivor
//               kernel already cleared the merge tables for that many tokens of that buffer.  crass_hip_merge ADOPTS a
//               queued merge iff no strings were passed in and premerge == 2, otherwise launches its own.  dm_prev_local: the previous merge ran on the device for this
//               context alone (only then is the next one queued ahead of time).  dm.active: the installed pattern set is
//               the device-built one (dm.M), not the host-built automaton / anchors; dm.host_built, dm.build_pending: the
//               host view (c->merge) of a device merge is being / has been rebuilt by the helper thread.
// Copies:       bulk_pending: exact-size copies of pass 1's hand-off records are in flight on copy_stream; wait_bulk()
//               before any getter reads them (a failed copy is reported there, bulk_status).
// Failure:      a device merge that gives up sets dm.h_st->fail; whoever sees it calls host_merge_fallback (counted in
//               n_merge_fallbacks) and repeats pass 2.  An allocation failure anywhere returns CRASS_ERR_OOM with the
//               context still usable (tests/test_gpu_device_merge.py::test_out_of_memory_is_reported_and_the_context_survives).
struct crass_hip_ctx {
    crass_params prm{};
    EnvSwitches env;
    DevParams dp{};
    int device = 0;
    hipStream_t stream = nullptr;
    mutable int last_hip = 0;

    // resident reads
    DevReads R{};
    bool have_reads = false;
    uint64_t read_base = 0;
    uint32_t max_len = 0;
    bool uniform = false;
    DevBuf<uint32_t> r_packed; DevBuf<uint64_t> r_word_off; DevBuf<uint32_t> r_lengths; DevBuf<uint64_t> r_header_id;
    DevBuf<uint32_t> r_exc_mask; DevBuf<uint64_t> r_exc_read; DevBuf<uint64_t> r_exc_off; DevBuf<uint8_t> r_exc_bytes;
    std::vector<uint64_t> h_exc_read;        // host copy of the exception read indices (local)

    // scratch
    DevBuf<uint64_t> d_mask; DevBuf<uint32_t> d_word_prefix; DevBuf<uint32_t> d_block_sums;
    DevBuf<uint64_t> d_idx; DevBuf<uint32_t> d_count; DevBuf<uint8_t> d_found; DevBuf<uint32_t> d_hit_info;
    bool hints_valid = false;       // d_hit_info doubles as the pass-1 seed-hint array (pass 2 reuses it afterwards)
    DevBuf<SurvOut> d_surv; DevBuf<char> d_dr; DevBuf<uint32_t> d_ss_pool; DevBuf<uint32_t> d_ss_used;
    DevBuf<RecruitOut> d_rec; DevBuf<uint32_t> d_exc_hit; DevBuf<uint64_t> d_extra;
    PinBuf<uint32_t> h_count; PinBuf<SurvOut> h_surv; PinBuf<char> h_dr; PinBuf<uint32_t> h_ss; PinBuf<uint64_t> h_idx;
    DevBuf<SurvOut> g_surv; DevBuf<char> g_dr; DevBuf<uint32_t> g_ss;       // host-loop sink: the found records, dense
    PinBuf<RecruitOut> h_rec;
    // automaton
    DevBuf<uint16_t> a_go16; DevBuf<uint32_t> a_go32; DevBuf<uint16_t> a_out; DevBuf<uint16_t> a_go4;
    DevAutomaton A{};
    HostAutomaton H;
    bool full_automaton_uploaded = false;
    DevBuf<uint32_t> a_go4w;
    DevBuf<uint32_t> a_anchor; DevBuf<uint32_t> d_slot_info; DevBuf<uint32_t> d_slot_pid; DevBuf<uint32_t> a_out_pid; DevBuf<uint32_t> a_pat_token;
    bool have_pat_token = false;
    DevAnchors K{};
    bool have_anchors = false;
    bool have_patterns = false;
    uint32_t n_installed_patterns = 0;

    // pass-1 results (host), final hand-off layout
    bool have_pass1 = false;
    struct P1List {
        std::vector<uint64_t> read; std::vector<uint8_t> low; std::vector<uint32_t> replen, nss;
        // (the two large arrays are filled by the host pool right after resize(): no value-initialisation pass over them)
        std::vector<uint64_t> ss_off; std::vector<uint32_t, NoInitAlloc<uint32_t>> ss; std::vector<uint16_t> dr_len; std::vector<char, NoInitAlloc<char>> dr;
        void clear() { read.clear(); low.clear(); replen.clear(); nss.clear(); ss_off.clear(); ss.clear(); dr_len.clear(); dr.clear(); }
        void reserve(size_t n, uint32_t stride) { read.reserve(n); low.reserve(n); replen.reserve(n); nss.reserve(n); ss_off.reserve(n); ss.reserve(n * 6); dr_len.reserve(n); dr.reserve(n * stride); }
        size_t size() const { return read.size(); }
    } cand;
    // fast path: the found records are gathered on the device into dense arrays and land in pinned
    // host memory already in the hand-off layout (no per-record host work)
    mutable struct P1Dense {
        DevBuf<uint16_t> d_dr_len; DevBuf<char> d_dr;       // dense DR strings of the found records (input of the de-duplication)
        uint64_t n = 0;
        bool active = false;
        // everything else of the found records: the compact hand-off blob (p1_blob_layout), assembled by the gather kernel
        // in device memory; its used bytes are copied by the runtime on the copy stream once the host knows the record
        // count (beside the merge kernels; bulk_pending until wait_bulk())
        PinBuf<uint8_t> h_blob; DevBuf<uint8_t> d_blob;
        P1Blob lay{};
        uint32_t pack_ss_cap = 0;
        // the ABI's wide per-candidate arrays (crass_candidates), widened from the compact blob on first request
        bool wide_ready = false;
        // (the two large ones are filled by the host pool right after resize(): no value-initialisation pass over 85 MB)
        std::vector<uint32_t> w_replen, w_nss; std::vector<uint32_t, NoInitAlloc<uint32_t>> w_ss; std::vector<uint64_t> w_ss_off; std::vector<uint16_t> w_dr_len;
        std::vector<char, NoInitAlloc<char>> w_dr;
        PinBuf<char> h_dr_fb; PinBuf<uint16_t> h_dr_len_fb; bool dr_fallback = false;    // candidates' own strings (no distinct list)
        void release()
        {
            h_blob.release(); d_blob.release(); h_dr_fb.release(); h_dr_len_fb.release(); d_dr_len.release(); d_dr.release();
        }
    } dense;
    DevBuf<uint64_t> d_fidx;
    DevBuf<uint64_t> g_fidx;                        // host-loop sink: the found records' slots, gathered (a source of the sink's copies)
    DevBuf<uint64_t> d_pos_hint, d_pos_hint_off; uint64_t n_pos_hint_words = 0;     // long reads: per-position seed hints
    DevBuf<uint32_t> d_punt;                                                          // long reads: [0] count, [1 ..] slots the light walk handed over
    DevBuf<uint32_t> d_redo;                                                          // ... [0] count, [1 ..] slots the full kernel hands on to its full-layout launch
    SdmaCopy *dma_sink[4] = {nullptr, nullptr, nullptr, nullptr};                     // the long-read sink's four copies on DMA engines (created with the first long-read set)
    DevBuf<uint32_t> d_pos_hint_blk; bool pos_hint_blk = false;                       // ragged lengths: read of every 256th hint word
    // device-side DR de-duplication (single-GPU merge fast path)
    DevBuf<unsigned long long> dd_keys; DevBuf<uint32_t> dd_first, dd_slot, dd_rep; DevBuf<uint64_t> dd_hash;
    PinBuf<uint32_t> h_rep; PinBuf<uint64_t> h_hash;
    bool have_rep = false;
    // device-side token ranks: distinct strings (first-occurrence order) + every candidate's distinct index.
    // With these the host merge never touches the per-candidate strings.
    DevBuf<uint32_t> dd_map; DevBuf<char> dd_dx_chars; DevBuf<uint16_t> dd_dx_len; DevBuf<uint64_t> dd_dx_hash;
    PinBuf<uint32_t> h_dmap; PinBuf<char> h_dx_chars; PinBuf<uint16_t> h_dx_len; PinBuf<uint64_t> h_dx_hash;
    uint64_t n_dx = 0;
    bool have_dev_tokens = false;
    uint32_t n_cu = 0;
    bool host_view_light = false;              // crass_hip_set_host_view: a device merge's host view = the own candidates' tokens only
    HostWorker worker;
    uint64_t surv_cap_hint = 0;
    uint64_t hit_cap_hint = 0;                // speculative bound for pass 2's flagged reads (0: none yet)
    bool recruit_exact = false;               // the next recruit call must not speculate (it repeats an overflowed one)               // speculative survivor bound for the next seed scan (0: none yet)
    hipStream_t copy_stream = nullptr;
    // Long reads: k_hint_positions runs in kHintParts slices of the reads, the first on the main stream and the others on
    // hint_stream BESIDE the walking kernel, which is launched slice by slice behind the slice's hints (run_survivors): the
    // hint kernel is VALU-issue bound, the walk LDS-latency bound at two waves per SIMD.
    static constexpr int kHintParts = 4;     // (at most; the default is two — see setup_pos_hints)
    hipStream_t hint_stream = nullptr;
    hipEvent_t ev_hint[kHintParts] = {nullptr, nullptr, nullptr, nullptr}, ev_hint_go = nullptr;
    int hint_parts = 1;                         // slices of this read set (1: one launch on the main stream)
    bool hint_filter = false;                   // the hint bits are this set's seed-scan filter (no lane-per-read filter for its layout)
    bool hint_filter_any = false;               // ... in their every-position form (another window or seed lattice): computed for the filter, not kept
    uint64_t hint_read_split[kHintParts + 1] = {0, 0, 0, 0, 0}, hint_word_split[kHintParts + 1] = {0, 0, 0, 0, 0};
    DevBuf<uint16_t> g_dr_len; DevBuf<uint64_t> hl_dx_idx;      // host-loop sink: dense DR lengths, scratch of the device de-duplication
    bool hint_pending = false;                  // slices are in flight on hint_stream: the main stream has not waited for them yet
    hipEvent_t ev_gathered = nullptr;
    hipEvent_t ev_premerge = nullptr;           // recorded behind a merge queued in the seed scan: the hand-off copy waits for it
    mutable bool bulk_pending = false;          // rare paths only: D2H copies of the candidates' own strings in flight on copy_stream
    // CRASS_OK, or CRASS_ERR_HIP with last_hip set: a failed copy must not be mistaken for delivered records
    int wait_bulk() const
    {
        if (!bulk_pending) return bulk_status;
        const hipError_t e = hipStreamSynchronize(copy_stream);
        int de = sdma_wait(dma);
        for (SdmaCopy *s : dma_sink) de |= sdma_wait(s);
        bulk_pending = false;
        if (e != hipSuccess) { last_hip = (int)e; bulk_status = CRASS_ERR_HIP; }
        else if (de) { last_hip = (int)hipErrorUnknown; bulk_status = CRASS_ERR_HIP; }
        return bulk_status;
    }
    SdmaCopy *dma = nullptr;                    // the hand-off records travel on a DMA engine when the HSA runtime offers one (sdma.cpp)
    // The distinct list of pass 1 (every candidate's index in it, the strings, their lengths) reaches the host the same way:
    // the de-duplication kernel used to write it straight into pinned memory — 5 MB of PCIe stores at 100 M reads, 90 us of
    // the pass-1 tail — while its first reader (the helper thread's view build, get_candidates, the host-merge fall-backs) is
    // a millisecond away.  dx_via_dma: three engine copies started when the counts are known; wait_dx() before any read.
    SdmaCopy *dma_dx[3] = {nullptr, nullptr, nullptr};
    bool dx_via_dma = false;
    mutable std::mutex dx_mu;
    mutable bool dx_pending = false, dx_stream_copy = false;
    mutable int dx_status = CRASS_OK;
    mutable bool dx_hash_valid = true;          // h_dx_hash holds the strings' TokenTable hashes (else: computed on first use)
    int wait_dx() const
    {
        std::lock_guard<std::mutex> lk(dx_mu);
        if (!dx_pending) return dx_status;
        int bad = 0;
        for (SdmaCopy *s : dma_dx) bad |= sdma_wait(s);
        if (dx_stream_copy) { const hipError_t e = hipStreamSynchronize(copy_stream); if (e != hipSuccess) { last_hip = (int)e; bad = 1; } }
        dx_pending = false; dx_stream_copy = false;
        if (bad) { if (!last_hip) last_hip = (int)hipErrorUnknown; dx_status = CRASS_ERR_HIP; }
        return dx_status;
    }
    mutable int bulk_status = CRASS_OK;         // sticky until the next seed scan issues new copies
    // device-side merge (dmerge.hip): clustering, non-redundant set, anchor keys and the pass-2 verification
    // index are built on the device; the host view (c->merge) is rebuilt from its per-token results while
    // pass 2 runs.  dm.active: the installed pattern set lives in dm.M, not in the automaton/anchors above.
    struct DM {
        DevBuf<uint64_t> packed, tmask; DevBuf<uint32_t> codes, owner, root_of, pat_token;
        DevBuf<uint32_t> kset_u32, ent_slot, anchor_tab, anchor_fp; DevBuf<uint8_t> blank; DevBuf<uint64_t> ents, rents; DevBuf<uint32_t> rset_u32, rd_slot;
        DevBuf<unsigned long long> rset_key, bk_key;
        DevBuf<unsigned long long> kset_key; DevBuf<DevMergeState> st; DevBuf<uint32_t> hot;
        PinBuf<DevMergeState> h_st; PinBuf<uint32_t> h_root; PinBuf<uint8_t> h_blank;
        std::vector<uint32_t> gid_tmp;
        DevMerge M{};
        bool active = false, host_built = false;
        int build_status = 0;                       // result of the host-view build that runs on `worker`
        bool build_pending = false;
        // what the build found out, written by whichever thread runs it and copied into the context's own fields
        // (n_installed_patterns, cnt.*, last_hip) by the CALLING thread once the build is known to be over (apply_build_result):
        // the helper thread never writes a field the caller may read without waiting for it
        struct BuildResult { bool valid = false; uint32_t n_patterns = 0, n_keys = 0; float ms_device = 0; int hip = 0; } br;
        uint64_t n_cand = 0;
        // what the host view is rebuilt from: the distinct list (pinned host copy) and every own candidate's index in it
        const char *hx_chars = nullptr; const uint16_t *hx_len = nullptr; uint64_t n_tok = 0;
        // multi-rank: the merge runs over the de-duplicated concatenation of every rank's list
        bool global = false; uint64_t my_off = 0, n_global = 0;
        DevBuf<char> g_chars, gx_chars; DevBuf<uint16_t> g_len, gx_len; DevBuf<unsigned long long> g_keys;
        DevBuf<uint32_t> g_first, g_slot, g_rep, g_prefix, g_bsum; DevBuf<uint64_t> g_hash, g_mask, g_idx;
        PinBuf<uint32_t> h_gmap; PinBuf<char> h_gx_chars; PinBuf<uint16_t> h_gx_len; PinBuf<uint64_t> h_gx_hash;
        std::vector<uint32_t> cand_map;
        hipEvent_t ev_done = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
        uint32_t want_post = 0;                     // the stage flag value pass 2's probe stores for this merge (h_flags[2])
        // crass_merge_view assembled on the device (k_dmx_*, dmerge.hip): scratch, the dense blob, its totals; the blob
        // reaches h_view on a DMA engine (dma_view) started by the helper thread once ev_view has fired
        DevBuf<uint32_t> x_u32, x_members, x_tile; DevBuf<uint8_t> x_blob; DevBuf<DevViewTotals> x_tot;
        PinBuf<DevViewTotals> x_htot; PinBuf<uint8_t> h_view;
        hipStream_t view_stream = nullptr; hipEvent_t ev_fork = nullptr, ev_view = nullptr, ev_apply = nullptr;
        SdmaCopy *dma_view = nullptr, *dma_view2 = nullptr, *dma_view3 = nullptr;
        uint32_t want_blank = 0;                    // the stage flag value k_dm_keys stores for this merge (h_flags[3]): blank[] is final      // (the blob travels in two pieces: the token half behind k_dmx_apply, the rest behind the last kernel)
        bool hx_on_host = true;                     // the gathered distinct list has a pinned host copy (else: on the device only)
        bool view_launched = false;                 // export kernels may be running on view_stream (ev_view orders after them)
        bool apply_recorded = false;                // ev_apply was recorded behind this merge's k_dmx_apply
        bool view_ready = false;                    // h_view holds the view of the current merge (crass_hip_get_merge reads it)
        DevViewTotals view_tot{};
        void release()
        {
            x_u32.release(); x_members.release(); x_tile.release(); x_blob.release(); x_tot.release(); x_htot.release(); h_view.release();
            packed.release(); tmask.release(); bk_key.release(); codes.release(); owner.release(); root_of.release();
            pat_token.release(); kset_u32.release(); ent_slot.release(); anchor_tab.release(); anchor_fp.release();
            blank.release(); ents.release(); rents.release(); rset_u32.release(); rd_slot.release(); rset_key.release(); kset_key.release(); st.release(); hot.release(); h_st.release(); h_root.release(); h_blank.release();
            g_chars.release(); gx_chars.release(); g_len.release(); gx_len.release(); g_keys.release(); g_first.release(); g_slot.release();
            g_rep.release(); g_prefix.release(); g_bsum.release(); g_hash.release(); g_mask.release(); g_idx.release();
            h_gmap.release(); h_gx_chars.release(); h_gx_len.release(); h_gx_hash.release();
            if (ev_done) (void)hipEventDestroy(ev_done);
            if (ev_t0) (void)hipEventDestroy(ev_t0);
            if (ev_t1) (void)hipEventDestroy(ev_t1);
            ev_done = ev_t0 = ev_t1 = nullptr;
            if (view_stream) (void)hipStreamSynchronize(view_stream);
            sdma_destroy(dma_view); dma_view = nullptr;
            sdma_destroy(dma_view2); dma_view2 = nullptr;
            sdma_destroy(dma_view3); dma_view3 = nullptr;
            if (ev_apply) (void)hipEventDestroy(ev_apply);
            ev_apply = nullptr;
            if (ev_fork) (void)hipEventDestroy(ev_fork);
            if (ev_view) (void)hipEventDestroy(ev_view);
            if (view_stream) (void)hipStreamDestroy(view_stream);
            ev_fork = ev_view = nullptr; view_stream = nullptr;
            view_launched = view_ready = false;
        }
    } dm;
    // one-collective exchange (crass_hip_exchange_setup): this rank's distinct list in a fixed-size device buffer
    struct Xchg {
        bool active = false;
        uint32_t world = 1, rank = 0, slot = 0;
        uint64_t cap = 0, needed = 0;
        uint32_t gx_cap_hint = 0;                   // bound for the merge queued ahead of the counters (previous global count x 1.5)
        hipEvent_t ev_counts = nullptr;
        DevBuf<uint8_t> send; DevBuf<uint32_t> xinfo; PinBuf<uint32_t> h_xinfo;
        uint64_t send_bytes() const { return (cap + 1) * (uint64_t)slot; }
    } xchg;
    // distinct candidate strings (multi-GPU exchange)
    std::vector<char> dx_chars; std::vector<uint16_t> dx_len; std::vector<uint32_t> dx_map; bool have_distinct = false;
    // host-loop sink, one chunk without exception reads: the found records reach pinned host memory on the copy stream (bulk_pending)
    // while the merge and pass 2 are queued; the ABI's per-candidate arrays (`cand`) are filled from them when somebody asks
    mutable std::function<void()> cand_fill;
    hipEvent_t ev_sink_copies = nullptr;        // behind those copies on the copy stream: pass 2 re-uses one of their source buffers (d_fidx)
    mutable uint64_t cand_pending_n = 0;
    void materialize_cand() const { if (cand_fill) { std::function<void()> f; f.swap(cand_fill); f(); } }
    uint64_t n_cand() const { return dense.active ? dense.n : (cand_fill ? cand_pending_n : cand.size()); }
    // the distinct DR strings (token order) and every candidate's index among them came from the DEVICE: the dense sink, or the
    // host-loop sink of a long-read set (hl_tokens: its gathered records were de-duplicated on the device beside the copies)
    bool hl_tokens = false;
    bool dev_tokens() const { return have_dev_tokens && (dense.active || hl_tokens); }
    void widen_p1() const;
    const char *cand_dr() const { if (dense.active) { widen_p1(); return dense.w_dr.data(); } materialize_cand(); return cand.dr.data(); }
    const uint16_t *cand_dr_len() const { if (dense.active) { widen_p1(); return dense.w_dr_len.data(); } materialize_cand(); return cand.dr_len.data(); }
    uint32_t dr_stride = 48;
    // merge
    MergeResult merge;
    bool have_merge = false;
    // pass-2 results
    bool have_pass2 = false;
    std::vector<uint64_t> q_read; std::vector<uint8_t> q_low; std::vector<uint32_t> q_start, q_end, q_token;
    std::vector<uint16_t> q_dr_len; std::vector<char, NoInitAlloc<char>> q_dr;      // (q_dr: whoever grows it fills the new bytes)
    // ... or, when the sink ran on the device, one pinned blob (p2_blob_layout)
    PinBuf<uint8_t> h_qblob; P2Blob q_lay{}; uint64_t q_n = 0; bool q_blob_active = false, q_wide_ready = false;

    crass_counters cnt{};
    uint32_t n_merge_fallbacks = 0, last_fallback_bits = 0;
    std::atomic<uint32_t> n_view_fallbacks{0};      // device merges whose host view was built by the host after all (a group beyond the export's cap)
    uint32_t n_bound_overflows[4] = {0, 0, 0, 0};   // speculation bounds that turned out too small (stage repeated): survivors, distinct, flagged, gathered
    hipEvent_t ev[12]{};
    // stage timing (crass_hip_set_stage_timing): an event record costs ~6 us of stream time, 14 of them 8 % of a 1 ms step.
    // 0 none (the default: a crass run reads no stage times), 1 the three large kernels only (seed scan, survivors, pass-2 scan), 2 every stage
    int timing_level = 0;
    // single-pass compaction (k_mask_compact_lb): status words + ticket counter shared by every launch of the context
    DevBuf<unsigned long long> lb_status; DevBuf<uint32_t> lb_ticket; PinBuf<uint32_t> h_lb_fail;
    uint32_t lb_epoch = 0;
    bool lb_on = true;
    // nullptr: use the three-kernel form (switched off, or allocation failed)
    const Lookback *next_lookback(uint64_t n_words, Lookback *out)
    {
        const uint32_t tw = lookback_tile_words(n_words);
        return next_lookback_tiles((n_words + tw - 1) / tw, out);
    }
    // element-wise kernels (k_found_compact, k_dx_flag_rank, k_valid_compact): tiles of 4096 elements
    const Lookback *next_lookback_elems(uint64_t n, Lookback *out) { return next_lookback_tiles((n + 4095) / 4096, out); }
    const Lookback *next_lookback_tiles(uint64_t n_tiles, Lookback *out)
    {
        if (!lb_on || n_tiles == 0) return nullptr;
        if (n_tiles > (1u << 24)) return nullptr;
        if (!lb_status.p || lb_status.n < n_tiles) {
            if (lb_status.ensure(std::max<uint64_t>(n_tiles * 2, 4096)) != hipSuccess) return nullptr;
            if (hipMemsetAsync(lb_status.p, 0, lb_status.n * 8, stream) != hipSuccess) return nullptr;
        }
        if (!lb_ticket.p) {
            if (lb_ticket.ensure(4) != hipSuccess || h_lb_fail.ensure(4) != hipSuccess) return nullptr;
            if (hipMemsetAsync(lb_ticket.p, 0, 16, stream) != hipSuccess) return nullptr;
            h_lb_fail.p[0] = 0;
        }
        lb_epoch = (lb_epoch + 1) & 0x3FFFFFFFu;
        if (lb_epoch == 0) lb_epoch = 1;
        out->status = lb_status.p; out->ticket = lb_ticket.p; out->epoch = lb_epoch; out->fail = h_lb_fail.p;
        return out;
    }
    // allocation only (crass_hip_load_reads): hands out no tickets — a launch that never runs must not move the ticket base
    void prealloc_lookback(uint64_t n_tiles)
    {
        if (!lb_on || n_tiles == 0 || n_tiles > (1u << 24)) return;
        if (!lb_status.p || lb_status.n < n_tiles) {
            if (lb_status.ensure(std::max<uint64_t>(n_tiles * 2, 4096)) != hipSuccess) return;
            if (hipMemsetAsync(lb_status.p, 0, lb_status.n * 8, stream) != hipSuccess) return;
            // (a re-allocated status array starts a new life: epochs of the old one mean nothing in it)
        }
        if (!lb_ticket.p) {
            if (lb_ticket.ensure(4) != hipSuccess || h_lb_fail.ensure(4) != hipSuccess) return;
            if (hipMemsetAsync(lb_ticket.p, 0, 16, stream) != hipSuccess) return;
            h_lb_fail.p[0] = 0;
        }
    }
    // after a synchronisation: a look-back spin that gave up invalidates the stage (never expected)
    int lookback_ok()
    {
        if (!h_lb_fail.p || !h_lb_fail.p[0]) return CRASS_OK;
        h_lb_fail.p[0] = 0; lb_on = false;
        last_hip = (int)hipErrorLaunchTimeOut;
        return CRASS_ERR_HIP;
    }
    // the device merge queued by the seed scan itself, right behind pass 1 (no host round trip in between): sized by a
    // bound learnt from the previous merge, adopted by crass_hip_merge when the counts turn out to fit
    bool dd_full_table = false;                         // a de-duplication table sized for the distinct strings overflowed once
    uint32_t dx_cap_hint = 0; bool dm_prev_local = false; int premerge = 0;      // premerge: 0 none, 2 queued and valid
    // a merge sized ahead of pass 1's survivor stage, whose kernel cleared the merge's tables (device_merge_prepare)
    uint64_t dm_prepared_n = 0; const char *dm_prepared_src = nullptr;
    bool premerge_inflight = false;                 // merge kernels may still be running when the seed scan returns
    double t_p1_sync = 0;                           // CRASS_MERGE_PROFILE: host time line between pass 1 and the merge
    bool spans_p1 = false, spans_p2 = false, span_survivors = false;     // spans to evaluate at the next counters fetch
    // Deferred pass-1 tail (crass_hip_exchange_set_deferred; a rank of a multi-rank job): crass_hip_seed_scan queues pass 1 up to the
    // kernel that fills the exchange's send buffer and RETURNS — the caller queues the collective and crass_hip_merge_gathered its
    // kernels behind it on the same stream, and only then does the host wait for pass 1's counters (p1_finish).  The ~60 us the
    // device idled between pass 1's last kernel and the exchange (host wake-up, the collective's and the unpack's launches) are
    // gone.  The counters of the deferred stage land in h_count[16..24) (the exchange's kernels rewrite h_count[0..8) meanwhile).
    // A launch that turns out unusable (bound too small, no device-resident distinct list) was marked by k_xg_fill in the send
    // buffer's header from the same counters: every rank then sees an exchange that "did not fit" and the step is repeated, this
    // context's next seed scan running synchronously (force_sync).
    // Stage flags (DevMerge::flag_pre, engine_internal.h): pinned words stored by the first kernel of a stage; the host polls them
    // where it used to wait for an event recorded between two kernels.  [0] pass 1 complete (k_xg_fill, deferred pass 1),
    // [1] everything in front of the merge complete (k_dm_pack_codes), [2] the merge complete (pass 2's probe).  A poll that runs
    // out of its budget falls back to a stream synchronisation (CRASS_NO_POLL: the events of round 5, the A/B switch).
    PinBuf<uint32_t> h_flags; uint32_t flag_seq = 0, want_p1 = 0, want_pre = 0, want_post = 0; bool poll_on = false;
    mutable std::atomic<uint32_t> n_poll_timeouts{0};
    bool poll_flag(int k, uint32_t want, double budget_ms) const
    {
        const volatile uint32_t *f = h_flags.p + k;
        const double t0 = now_ms();
        for (uint32_t it = 0;; it++) {
            if ((int32_t)(*f - want) >= 0) { std::atomic_thread_fence(std::memory_order_acquire); return true; }
            __builtin_ia32_pause();
            if ((it & 255u) == 255u && now_ms() - t0 > budget_ms) { n_poll_timeouts++; return false; }
        }
    }
    struct P1Deferred {
        bool enabled = false, active = false, finishing = false, force_sync = false;
        uint64_t n_bound = 0; uint32_t ss_cap = 0, ss_elem = 1;
        bool fast = false, hint_filtered = false;
    } p1d;
    const uint32_t *p1_counts() const { return p1d.finishing ? h_count.p + 16 : h_count.p; }
    // level 1 only: which of the three large kernels are bracketed (bit 0 seed scan, 1 survivors, 2 pass-2 scan)
    unsigned timing_focus = 7;
    bool timed(int i, int level) const
    {
        if (timing_level < level) return false;
        if (timing_level >= 2 || level != 1) return true;
        const unsigned bit = (i <= 1) ? 1u : (i == 8 || i == 9) ? 2u : 4u;
        return (timing_focus & bit) != 0;
    }
    hipError_t stamp(int i, int level) { return timed(i, level) ? hipEventRecord(ev[i], stream) : hipSuccess; }
    float span(int a, int b, int level) const
    {
        float ms = 0;
        if (!timed(a, level) || !timed(b, level) || hipEventElapsedTime(&ms, ev[a], ev[b]) != hipSuccess) ms = 0;
        return ms;
    }
};

// crass_candidates' arrays from the compact hand-off blob (and the distinct-string list for the DR strings)
void crass_hip_ctx::widen_p1() const
{
    P1Dense &D = dense;
    if (D.wide_ready) return;
    (void)wait_bulk();                  // (callers have checked its status)
    const uint64_t n = D.n;
    const uint32_t ss_cap = D.pack_ss_cap, stride = dr_stride;
    const uint8_t *hb = D.h_blob.p;
    const uint16_t *b_replen = (const uint16_t *)(hb + D.lay.replen), *b_ss = (const uint16_t *)(hb + D.lay.ss);
    const uint8_t *b_nss = hb + D.lay.nss, *b_ss8 = hb + D.lay.ss;
    D.w_replen.resize(n); D.w_nss.resize(n); D.w_ss_off.resize(n); D.w_ss.resize(n * (size_t)ss_cap);
    D.w_dr_len.resize(n); D.w_dr.resize(n * (size_t)stride);
    if (D.dr_fallback) (void)hipStreamSynchronize(copy_stream);
    else (void)wait_dx();               // (the distinct list the candidates' strings are read from)
    // ~150 bytes per candidate (564 k candidates at 100 M reads: 85 MB, 4.5 ms on one thread): ranges of candidates on the host pool
    const size_t per_task = 16384;
    host_parallel_for((size_t)((n + per_task - 1) / per_task), 16, [&](size_t t) {
        const uint64_t k0 = t * per_task, k1 = std::min<uint64_t>(n, k0 + per_task);
        for (uint64_t k = k0; k < k1; k++) { D.w_replen[k] = b_replen[k]; D.w_nss[k] = b_nss[k]; D.w_ss_off[k] = k * (uint64_t)ss_cap; }
        if (D.lay.ss_elem == 1) for (uint64_t i = k0 * ss_cap; i < k1 * (uint64_t)ss_cap; i++) D.w_ss[i] = b_ss8[i];
        else for (uint64_t i = k0 * ss_cap; i < k1 * (uint64_t)ss_cap; i++) D.w_ss[i] = b_ss[i];
        if (D.dr_fallback) {
            memcpy(D.w_dr.data() + k0 * (size_t)stride, D.h_dr_fb.p + k0 * (size_t)stride, (size_t)(k1 - k0) * stride);
            memcpy(D.w_dr_len.data() + k0, D.h_dr_len_fb.p + k0, (size_t)(k1 - k0) * 2);
        } else {
            for (uint64_t k = k0; k < k1; k++) {
                const uint32_t j = h_dmap.p[k];
                memcpy(D.w_dr.data() + k * (size_t)stride, h_dx_chars.p + j * (size_t)stride, stride);
                D.w_dr_len[k] = h_dx_len.p[j];
            }
        }
    });
    D.wide_ready = true;
}

// the helper thread owns c->merge while a host-view build is in flight: wait for it before touching that state
static void quiesce_worker(crass_hip_ctx *c)
{
    if (c->dm.build_pending) {
        if (!c->worker.run_here_if_not_started()) c->worker.wait();
        c->dm.build_pending = false;
        (void)c->worker.take_failed();
        c->dm.br.valid = false;                         // (results of a build nobody asked for any more)
        if (c->dm.br.hip) { c->last_hip = c->dm.br.hip; c->dm.br.hip = 0; }
    }
}

static bool getenv_once_copy_early()                 // A/B switch: CRASS_COPY_EARLY=1 starts the hand-off copy beside the merge (the old order)
{
    static const bool on = getenv("CRASS_COPY_EARLY") != nullptr;
    return on;
}

#define HIPCHK(ctx, call)                                                       \
    do {                                                                        \
        hipError_t e__ = (call);                                                \
        if (e__ != hipSuccess) { (ctx)->last_hip = (int)e__; return e__ == hipErrorOutOfMemory ? CRASS_ERR_OOM : CRASS_ERR_HIP; } \
    } while (0)

extern "C" {

int crass_hip_abi_version(void) { return CRASS_HIP_ABI_VERSION; }

void crass_default_params(crass_params *p)
{
    p->lowDRsize = 23; p->highDRsize = 47; p->lowSpacerSize = 26; p->highSpacerSize = 50;
    p->searchWindowLength = 8; p->minNumRepeats = 2; p->kmer_clust_size = 6;
}

const char *crass_hip_strerror(int s)
{
    switch (s) {
        case CRASS_OK: return "ok";
        case CRASS_ERR_INVALID_ARG: return "invalid argument";
        case CRASS_ERR_UNSUPPORTED: return "parameter outside the device path's implementation limits";
        case CRASS_ERR_NO_DEVICE: return "no usable HIP device";
        case CRASS_ERR_HIP: return "HIP runtime call failed";
        case CRASS_ERR_OOM: return "out of memory";
        case CRASS_ERR_STATE: return "call order violated";
        case CRASS_ERR_SEARCH_FATAL: return "Fatal error in search algorithm!";
        case CRASS_ERR_OVERFLOW: return "device pool overflow";
        case CRASS_ERR_IO: return "I/O error";
        case CRASS_ERR_RCCL: return "RCCL call failed (crass_hip_group_last_error())";
        default: return "unknown status";
    }
}

int crass_hip_last_hip_error(const crass_hip_ctx *ctx) { return ctx ? ctx->last_hip : 0; }
void *crass_hip_stream(const crass_hip_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int crass_hip_create(const crass_params *p, int device, crass_hip_ctx **out)
{
    if (!p || !out) return CRASS_ERR_INVALID_ARG;
    *out = nullptr;
    // option validation as crass.cpp:264-271,316-324,351-358,366-374,388-400
    if (p->searchWindowLength < CRASS_HIP_MIN_WINDOW || p->searchWindowLength > CRASS_HIP_MAX_WINDOW) return CRASS_ERR_INVALID_ARG;
    if (p->lowDRsize < 8 || p->lowSpacerSize < 8) return CRASS_ERR_INVALID_ARG;
    if (p->lowDRsize >= p->highDRsize || p->lowSpacerSize >= p->highSpacerSize) return CRASS_ERR_INVALID_ARG;
    if (p->minNumRepeats < 2) return CRASS_ERR_INVALID_ARG;
    if (p->highDRsize > CRASS_HIP_MAX_DR) return CRASS_ERR_UNSUPPORTED;
    // lowDR < 2w-1 makes the reference's `unsigned skips = lowDR - (2w-1)` wrap to ~4e9
    // (libcrispr.cpp:281); its `j = j + skips` then walks BACKWARDS after a failed candidate and
    // need not terminate.  Ill-defined in the reference, refused here.
    if (p->lowDRsize < 2 * p->searchWindowLength - 1) return CRASS_ERR_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return CRASS_ERR_NO_DEVICE;
    crass_hip_ctx *c = new (std::nothrow) crass_hip_ctx();
    if (!c) return CRASS_ERR_OOM;
    c->prm = *p;
    c->device = device;
    c->dp.lowDR = p->lowDRsize; c->dp.highDR = p->highDRsize; c->dp.lowSp = p->lowSpacerSize; c->dp.highSp = p->highSpacerSize;
    c->dp.window = p->searchWindowLength; c->dp.minRepeats = p->minNumRepeats;
    uint32_t skips = p->lowDRsize - (2 * p->searchWindowLength - 1);     // unsigned, libcrispr.cpp:281
    if (skips < 1) skips = 1;
    c->dp.skips = skips;
    c->env.read();
    g_dev_cap.store(c->env.pool_cap_bytes, std::memory_order_relaxed);
    c->dp.debug_stop = c->env.surv_debug;
    if (c->env.stage_timing >= 0) c->timing_level = c->env.stage_timing;
    if (c->env.no_lookback) c->lb_on = false;                    // A/B switch: three-kernel compaction
    c->dr_stride = (p->highDRsize + 15u) & ~15u;
    { const hipError_t he = hipSetDevice(device);
      if (he != hipSuccess) { fprintf(stderr, "[crass_hip] hipSetDevice(%d): %s\n", device, hipGetErrorString(he)); delete c; return CRASS_ERR_NO_DEVICE; } }
    if (const char *e = getenv("CRASS_WAIT")) {           // A/B: how a host thread waits for the device (spin | yield | block)
        const unsigned f = !strcmp(e, "spin") ? hipDeviceScheduleSpin : !strcmp(e, "yield") ? hipDeviceScheduleYield : !strcmp(e, "block") ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto;
        const hipError_t fe = hipSetDeviceFlags(f);
        if (fe != hipSuccess) fprintf(stderr, "[crass_hip] hipSetDeviceFlags(%s): %s\n", e, hipGetErrorString(fe));
    }
    if (getenv("CRASS_SURV_PROF")) {                    // diagnostics: phase cycles of the wave-per-read kernel (tools/longread_phases.py)
        void *pp = nullptr;
        if (hipMalloc(&pp, (192 + 2 * 16384) * 8) == hipSuccess && hipMemset(pp, 0, (192 + 2 * 16384) * 8) == hipSuccess) {
            c->dp.prof = (unsigned long long *)pp;
            const unsigned long long big = ~0ull;
            (void)hipMemcpy(c->dp.prof + 188, &big, 8, hipMemcpyHostToDevice);
        }
    }
    for (hipStream_t *sp : {&c->stream, &c->copy_stream}) {
        const hipError_t he = hipStreamCreateWithFlags(sp, hipStreamNonBlocking);
        if (he != hipSuccess) { fprintf(stderr, "[crass_hip] hipStreamCreateWithFlags on device %d: %s\n", device, hipGetErrorString(he)); delete c; return CRASS_ERR_NO_DEVICE; }
    }
    // (hint_stream and its events: created with the first long-read set, setup_pos_hints)
    for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) { delete c; return CRASS_ERR_HIP; }
    if (c->h_flags.ensure(8) != hipSuccess) { delete c; return CRASS_ERR_OOM; }
    memset(c->h_flags.p, 0, 32);
    c->poll_on = getenv("CRASS_NO_POLL") == nullptr;
    if (hipEventCreateWithFlags(&c->ev_gathered, hipEventDisableTiming) != hipSuccess) { delete c; return CRASS_ERR_HIP; }
    if (hipEventCreateWithFlags(&c->ev_premerge, hipEventDisableTiming) != hipSuccess) { delete c; return CRASS_ERR_HIP; }
    c->dma = sdma_create();                             // (nullptr: the copy kernel is used)
    if (c->dma && !getenv("CRASS_DX_PINNED")) {         // (A/B switch: the de-duplication kernel writes the distinct list to pinned memory itself)
        for (auto &d : c->dma_dx) d = sdma_create();
        c->dx_via_dma = c->dma_dx[0] && c->dma_dx[1] && c->dma_dx[2];
    }
    (void)warm_dmerge_module();                         // (code-object load: here, not inside the first merge)
    unsigned char tab[128];
    build_comp_table(tab);
    if (upload_comp_table(tab) != hipSuccess) { delete c; return CRASS_ERR_HIP; }
    *out = c;
    return CRASS_OK;
}

int crass_hip_stream_wait_event(crass_hip_ctx *c, void *event)
{
    if (!c || !event) return CRASS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    HIPCHK(c, hipStreamWaitEvent(c->stream, (hipEvent_t)event, 0));
    return CRASS_OK;
}

int crass_hip_reload_env(crass_hip_ctx *c)
{
    if (!c) return CRASS_ERR_INVALID_ARG;
    c->env.read();
    g_dev_cap.store(c->env.pool_cap_bytes, std::memory_order_relaxed);
    c->dp.debug_stop = c->env.surv_debug;
    if (c->env.stage_timing >= 0) c->timing_level = c->env.stage_timing;
    c->lb_on = !c->env.no_lookback;
    return CRASS_OK;
}

int crass_hip_set_stage_timing(crass_hip_ctx *c, int level)
{
    if (!c || level < 0 || level > 2) return CRASS_ERR_INVALID_ARG;
    c->timing_level = level;
    return CRASS_OK;
}

int crass_hip_set_host_view(crass_hip_ctx *c, int light)
{
    if (!c || light < 0 || light > 1) return CRASS_ERR_INVALID_ARG;
    quiesce_worker(c);
    c->host_view_light = light != 0;
    // a merge prepared before this call (crass_hip_exchange_setup sizes one at load) would still export the view nobody reads:
    // a light context's merges skip the k_dmx_* kernels (the next prepare decides again)
    if (c->host_view_light) c->dm.M.x_on = 0;
    return CRASS_OK;
}

int crass_hip_set_timing_focus(crass_hip_ctx *c, unsigned kernels)
{
    if (!c || kernels > 7u) return CRASS_ERR_INVALID_ARG;
    c->timing_focus = kernels;
    return CRASS_OK;
}

void crass_hip_destroy(crass_hip_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    quiesce_worker(c);
    c->worker.stop();
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->hint_stream) (void)hipStreamSynchronize(c->hint_stream);
    (void)sdma_wait(c->dma);                            // (before any buffer goes)
    (void)c->wait_dx();
    if (getenv("CRASS_POLL_REPORT")) fprintf(stderr, "[crass_hip] stage-flag polls that ran out of their budget: %u (of %u flags)\n", c->n_poll_timeouts.load(), c->flag_seq);
    c->lb_status.release(); c->lb_ticket.release(); c->h_lb_fail.release(); c->h_flags.release();
    if (c->xchg.ev_counts) (void)hipEventDestroy(c->xchg.ev_counts);
    c->dm.release(); c->h_qblob.release(); c->xchg.send.release(); c->xchg.xinfo.release(); c->xchg.h_xinfo.release();
    c->dd_map.release(); c->dd_dx_chars.release(); c->dd_dx_len.release(); c->dd_dx_hash.release();
    c->h_dmap.release(); c->h_dx_chars.release(); c->h_dx_len.release(); c->h_dx_hash.release();
    if (c->ev_gathered) (void)hipEventDestroy(c->ev_gathered);
    if (c->ev_sink_copies) (void)hipEventDestroy(c->ev_sink_copies);
    for (int q = 0; q < crass_hip_ctx::kHintParts; q++) if (c->ev_hint[q]) (void)hipEventDestroy(c->ev_hint[q]);
    if (c->ev_hint_go) (void)hipEventDestroy(c->ev_hint_go);
    if (c->hint_stream) (void)hipStreamDestroy(c->hint_stream);
    sdma_destroy(c->dma); c->dma = nullptr;
    for (auto &d : c->dma_sink) { sdma_destroy(d); d = nullptr; }
    for (auto &d : c->dma_dx) { sdma_destroy(d); d = nullptr; }
    if (c->ev_premerge) (void)hipEventDestroy(c->ev_premerge);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    c->r_packed.release(); c->r_word_off.release(); c->r_lengths.release(); c->r_header_id.release();
    c->r_exc_mask.release(); c->r_exc_read.release(); c->r_exc_off.release(); c->r_exc_bytes.release();
    c->d_mask.release(); c->d_word_prefix.release(); c->d_block_sums.release(); c->d_idx.release(); c->d_count.release();
    c->d_found.release(); c->d_hit_info.release(); c->d_surv.release(); c->d_dr.release(); c->d_ss_pool.release();
    c->d_ss_used.release(); c->d_rec.release(); c->d_exc_hit.release(); c->d_extra.release();
    c->g_surv.release(); c->g_dr.release(); c->g_ss.release(); c->g_fidx.release(); c->h_count.release(); c->h_surv.release(); c->h_dr.release(); c->h_ss.release(); c->h_idx.release(); c->h_rec.release();
    c->a_go4w.release(); c->a_go16.release(); c->a_go32.release(); c->a_out.release(); c->a_go4.release(); c->dense.release(); c->d_fidx.release(); c->d_pos_hint.release(); c->d_pos_hint_off.release(); c->d_pos_hint_blk.release(); c->d_punt.release(); c->d_redo.release(); c->dd_keys.release(); c->dd_first.release(); c->dd_slot.release(); c->dd_rep.release(); c->dd_hash.release(); c->h_rep.release(); c->h_hash.release(); c->a_anchor.release(); c->d_slot_info.release(); c->d_slot_pid.release(); c->a_out_pid.release(); c->a_pat_token.release();
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int validate_reads(const crass_reads *r)
{
    if (!r) return CRASS_ERR_INVALID_ARG;
    if (r->n_reads && !r->packed) return CRASS_ERR_INVALID_ARG;
    if (!r->stride_words && r->n_reads && !r->word_off) return CRASS_ERR_INVALID_ARG;
    if (!r->uniform_len && r->n_reads && !r->lengths) return CRASS_ERR_INVALID_ARG;
    if (r->n_exceptions && (!r->exc_read || !r->exc_off || !r->exc_bytes)) return CRASS_ERR_INVALID_ARG;
    if (r->uniform_len > CRASS_HIP_MAX_READ_LEN) return CRASS_ERR_UNSUPPORTED;
    return CRASS_OK;
}

static int alloc_scratch(crass_hip_ctx *c)
{
    const uint64_t n = c->R.n_reads;
    // (the mask / prefix scratch also serves compactions over survivor and hit SLOTS, whose launches are sized by bounds of
    // at least 65 536 / 4 096 slots whatever the read count: never smaller than that many bits)
    const uint64_t n_words = std::max<uint64_t>((n + 63) / 64, 1024);
    HIPCHK(c, c->d_mask.ensure(n_words + 1));
    HIPCHK(c, c->d_word_prefix.ensure(n_words + 1));
    HIPCHK(c, c->d_block_sums.ensure((n_words + 255) / 256 + 2));
    HIPCHK(c, c->d_idx.ensure(n + 1));
    HIPCHK(c, c->d_count.ensure(8));
    HIPCHK(c, c->d_found.ensure(n + 1));
    HIPCHK(c, c->d_hit_info.ensure(n + 1));
    HIPCHK(c, c->d_ss_used.ensure(4));
    HIPCHK(c, c->h_count.ensure(32));      // [0..8) the stage counters, [16..24) those of a deferred pass 1 (p1d)
    HIPCHK(c, hipMemsetAsync(c->d_found.p, 0, n + 1, c->stream));
    return CRASS_OK;
}

// long reads skip the per-read filter; with the default window and DR/spacer bounds they get one seed-hint bit
// per base position instead (k_hint_positions).  lengths == nullptr: uniform length.
static int setup_pos_hints(crass_hip_ctx *c, const uint32_t *lengths, uint32_t uniform_len, uint64_t n)
{
    c->n_pos_hint_words = 0;
    c->R.pos_hint = nullptr; c->R.pos_hint_off = nullptr; c->R.wave_walk = 0; c->R.hint_all = 0;
    const DevParams &P = c->dp;
    // (skips == 8: the hints are kept per residue class mod 8 — k_hint_positions fills the lattice class, the walking wave the others)
    // (... and any shift range the hint tile's halo covers: -s / -S / -D keep the hints, launch_hint_positions)
    // (... and sets that stay on the filtered path but have no lane-per-read filter — reads of 257 .. 2 048 bases, strides that differ:
    // the hint bits are their filter, crass_hip_seed_scan; CRASS_NO_HINT_FILTER: the A/B switch, back to k_filter_general)
    const bool lane_filter = c->R.stride_words >= 4 && c->R.stride_words <= 16;
    c->hint_filter = c->max_len <= c->env.long_min && !lane_filter && !c->env.no_hint_filter;
    c->hint_filter_any = false;
    const bool shifts_ok = P.lowDR + P.lowSp >= 17 && P.highDR + P.highSp <= 127 && P.highDR + P.highSp >= P.lowDR + P.lowSp;
    const bool lattice_hints = P.window == 8 && P.skips == 8 && shifts_ok;
    // (... under another window or seed lattice those sets get the every-position form of the bits as their filter, nothing kept:
    // launch_hint_filter_any; the tiles' read index is all that is set up for it)
    const bool any_filter = c->hint_filter && !lattice_hints && shifts_ok && P.window >= 6 && P.window <= 9 && P.skips >= 1;
    if ((c->max_len <= c->env.long_min && !c->hint_filter) || n == 0 || !(lattice_hints || any_filter)) return CRASS_OK;
    if (c->env.no_pos_hints) return CRASS_OK;             // A/B switch
    std::vector<uint64_t> off(n + 1);
    uint64_t at = 0;
    for (uint64_t i = 0; i < n; i++) { off[i] = at; at += ((uint64_t)(lengths ? lengths[i] : uniform_len) + 63) / 64; }
    off[n] = at;
    HIPCHK(c, c->d_pos_hint_off.ensure(n + 1)); HIPCHK(c, c->d_pos_hint.ensure(at + 1));
    HIPCHK(c, hipMemcpy(c->d_pos_hint_off.p, off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    c->n_pos_hint_words = at;
    c->hint_filter_any = any_filter;
    c->R.pos_hint = c->d_pos_hint.p; c->R.pos_hint_off = c->d_pos_hint_off.p;
    c->R.hint_all = any_filter ? 1u : 0u;               // (every position's bit, written by the filter launch itself)
    c->R.wave_walk = c->max_len > std::min<uint32_t>(c->env.long_min, c->env.wave_walk_min) ? 1u : 0u;
    c->pos_hint_blk = false;
    // slices: read boundaries n i / K; slice i covers the hint words [roundup256(off[r_i]), roundup256(off[r_i+1])), so every word
    // of a read below r_i+1 belongs to a slice <= i (CRASS_HINT_PARTS=1: the A/B switch)
    { const char *hp = getenv("CRASS_HINT_PARTS");       // (A/B: 1 = one launch on the main stream)
      const int want = hp ? atoi(hp) : 2;
      // (two: each slice's walk is its own launch with its own ramp and tail — four slices were slower than none, 15.5 vs 14.3 ms
      // per step on configs[3], two are 14.0; an explicit CRASS_HINT_PARTS also slices small sets: the tests)
      const bool big = hp ? (n >= 64 && at >= 2048) : (n >= 4096 && at >= (1u << 20));
      c->hint_parts = (big && !c->hint_filter) ? std::min(std::max(want, 1), (int)crass_hip_ctx::kHintParts) : 1; }
    if (c->hint_parts > 1 && !c->hint_stream) {
        // lowest priority: its blocks fill what the walking kernel (two waves per SIMD: its LDS) leaves free, not the other way
        // round.  Created with the first long-read set only: a second low-priority stream in every context changed how the
        // runtime spreads streams over its hardware queues, and the merge's view export (its own low-priority stream) then
        // cost a short-read step 0.5 ms instead of 0.1
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        HIPCHK(c, hipStreamCreateWithPriority(&c->hint_stream, hipStreamNonBlocking, getenv("CRASS_HINT_SAME_PRIORITY") ? 0 : prio_lo));
        for (int q = 0; q < crass_hip_ctx::kHintParts; q++) HIPCHK(c, hipEventCreateWithFlags(&c->ev_hint[q], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_hint_go, hipEventDisableTiming));
    }
    for (int q = 0; q <= c->hint_parts; q++) {
        const uint64_t r = q == c->hint_parts ? n : n * (uint64_t)q / (uint64_t)c->hint_parts;
        c->hint_read_split[q] = r;
        c->hint_word_split[q] = q == c->hint_parts ? at : ((off[r] + 255) & ~255ull);
    }
    if (lengths && n <= 0xFFFFFFFFull) {                // ragged: the read of every block's first hint word (k_hint_positions)
        std::vector<uint32_t> blk((at + 255) / 256 + 1, 0);
        uint64_t r = 0;
        for (uint64_t b = 0; b * 256 < at; b++) {
            while (r + 1 < n && off[r + 1] <= b * 256) r++;
            blk[b] = (uint32_t)r;
        }
        blk.back() = (uint32_t)(n - 1);                  // (the bisection's upper end for the last block)
        HIPCHK(c, c->d_pos_hint_blk.ensure(blk.size()));
        HIPCHK(c, hipMemcpy(c->d_pos_hint_blk.p, blk.data(), blk.size() * 4, hipMemcpyHostToDevice));
        c->pos_hint_blk = true;
    }
    return CRASS_OK;
}

static int first_call_bounds(crass_hip_ctx *c);
static void presize_hostloop(crass_hip_ctx *c);

static void reset_results(crass_hip_ctx *c)
{
    quiesce_worker(c);
    if (c->p1d.active) { (void)hipStreamSynchronize(c->stream); c->p1d.active = false; }      // (a deferred pass 1 nobody finished)
    (void)c->wait_bulk();               // (a hand-off copy still on its DMA engine: the buffers may be re-sized next, and a
                                        // free only waits for the device's queues)
    c->have_pass1 = c->have_merge = c->have_pass2 = c->have_patterns = false;
    c->dm.active = false;
    memset(&c->cnt, 0, sizeof(c->cnt));
}

int crass_hip_load_reads(crass_hip_ctx *c, const crass_reads *h)
{
    if (!c) return CRASS_ERR_INVALID_ARG;
    int v = validate_reads(h);
    if (v) return v;
    (void)hipSetDevice(c->device);
    reset_results(c);
    c->have_reads = false;                              // (a failure below must not leave the previous reads half replaced)
    const uint64_t n = h->n_reads;
    // host-side scan for the total word count / max length
    uint64_t total_words = 0;
    uint32_t max_len = h->uniform_len;
    if (!h->uniform_len) for (uint64_t i = 0; i < n; i++) max_len = std::max(max_len, h->lengths[i]);
    if (max_len > CRASS_HIP_MAX_READ_LEN) return CRASS_ERR_UNSUPPORTED;
    if (h->stride_words) total_words = n * (uint64_t)h->stride_words;
    else if (n) {
        uint32_t lastL = h->uniform_len ? h->uniform_len : h->lengths[n - 1];
        total_words = h->word_off[n - 1] + (lastL + 15) / 16;
    }
    HIPCHK(c, c->r_packed.ensure(total_words + 4));
    HIPCHK(c, hipMemcpyAsync(c->r_packed.p, h->packed, total_words * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->r_packed.p + total_words, 0, 16, c->stream));
    DevReads R{};
    R.packed = c->r_packed.p; R.n_reads = n; R.stride_words = h->stride_words; R.uniform_len = h->uniform_len;
    if (!h->stride_words) {
        HIPCHK(c, c->r_word_off.ensure(n));
        HIPCHK(c, hipMemcpyAsync(c->r_word_off.p, h->word_off, n * 8, hipMemcpyHostToDevice, c->stream));
        R.word_off = c->r_word_off.p;
    }
    if (!h->uniform_len) {
        HIPCHK(c, c->r_lengths.ensure(n));
        HIPCHK(c, hipMemcpyAsync(c->r_lengths.p, h->lengths, n * 4, hipMemcpyHostToDevice, c->stream));
        R.lengths = c->r_lengths.p;
    }
    if (h->header_id) {
        HIPCHK(c, c->r_header_id.ensure(n));
        HIPCHK(c, hipMemcpyAsync(c->r_header_id.p, h->header_id, n * 8, hipMemcpyHostToDevice, c->stream));
        R.header_id = c->r_header_id.p;
    }
    const uint64_t mask_words = (n + 31) / 32 + 1;
    HIPCHK(c, c->r_exc_mask.ensure(mask_words));
    HIPCHK(c, hipMemsetAsync(c->r_exc_mask.p, 0, mask_words * 4, c->stream));
    R.exc_mask = c->r_exc_mask.p;
    R.n_exc = h->n_exceptions;
    c->h_exc_read.assign(h->exc_read, h->exc_read + h->n_exceptions);
    if (h->n_exceptions) {
        const uint64_t ne = h->n_exceptions;
        for (uint64_t i = 0; i < ne; i++) {
            uint64_t l = h->exc_off[i + 1] - h->exc_off[i];
            if (l > CRASS_HIP_MAX_READ_LEN) return CRASS_ERR_UNSUPPORTED;
            max_len = std::max<uint32_t>(max_len, (uint32_t)l);
        }
        HIPCHK(c, c->r_exc_read.ensure(ne));
        HIPCHK(c, c->r_exc_off.ensure(ne + 1));
        HIPCHK(c, c->r_exc_bytes.ensure(h->exc_off[ne] + 16));
        HIPCHK(c, hipMemcpyAsync(c->r_exc_read.p, h->exc_read, ne * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->r_exc_off.p, h->exc_off, (ne + 1) * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->r_exc_bytes.p, h->exc_bytes, h->exc_off[ne], hipMemcpyHostToDevice, c->stream));
        R.exc_read = c->r_exc_read.p; R.exc_off = c->r_exc_off.p; R.exc_bytes = c->r_exc_bytes.p;
        HIPCHK(c, launch_build_exc_mask(R.exc_read, ne, c->r_exc_mask.p, c->stream));
    }
    c->R = R;
    c->read_base = h->read_index_base;
    c->max_len = max_len;
    c->uniform = h->uniform_len != 0;
    c->have_reads = true;
    int s = alloc_scratch(c);
    if (s) return s;
    s = setup_pos_hints(c, h->uniform_len ? nullptr : h->lengths, h->uniform_len, n);
    if (s) return s;
    s = first_call_bounds(c);
    if (s) return s;
    presize_hostloop(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->cnt.n_reads = n; c->cnt.n_exceptions = h->n_exceptions; c->cnt.bytes_reads_device = total_words * 4;
    return CRASS_OK;
}

int crass_hip_attach_device_reads(crass_hip_ctx *c, const crass_reads *d)
{
    if (!c) return CRASS_ERR_INVALID_ARG;
    int v = validate_reads(d);
    if (v) return v;
    if (!d->uniform_len || !d->stride_words) return CRASS_ERR_UNSUPPORTED;   // attach mode: uniform shards only
    if (d->n_exceptions) return CRASS_ERR_UNSUPPORTED;
    (void)hipSetDevice(c->device);
    reset_results(c);
    c->have_reads = false;
    DevReads R{};
    R.packed = d->packed; R.n_reads = d->n_reads; R.stride_words = d->stride_words; R.uniform_len = d->uniform_len;
    R.header_id = d->header_id;
    const uint64_t mask_words = (d->n_reads + 31) / 32 + 1;
    HIPCHK(c, c->r_exc_mask.ensure(mask_words));
    HIPCHK(c, hipMemsetAsync(c->r_exc_mask.p, 0, mask_words * 4, c->stream));
    R.exc_mask = c->r_exc_mask.p;
    c->h_exc_read.clear();
    c->R = R;
    c->read_base = d->read_index_base;
    c->max_len = d->uniform_len;
    c->uniform = true;
    c->have_reads = true;
    int s = alloc_scratch(c);
    if (s) return s;
    s = setup_pos_hints(c, nullptr, d->uniform_len, d->n_reads);
    if (s) return s;
    s = first_call_bounds(c);
    if (s) return s;
    presize_hostloop(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->cnt.n_reads = d->n_reads; c->cnt.n_exceptions = 0;
    c->cnt.bytes_reads_device = d->n_reads * (uint64_t)d->stride_words * 4;
    return CRASS_OK;
}

static void ensure_distinct(crass_hip_ctx *c);

// ------------------------------------------------------------------------------------------
// pass 1
// ------------------------------------------------------------------------------------------
// runs the survivor kernel over `n_total` survivors (packed list in d_idx, or the exception
// list) in chunks and appends every found record, in order, to `L`
// the main stream waits for every slice of the position hints (whoever reads them without slicing its own launches)
static int hint_wait_all(crass_hip_ctx *c)
{
    if (!c->hint_pending) return CRASS_OK;
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_hint[c->hint_parts - 1], 0));
    c->hint_pending = false;
    return CRASS_OK;
}

static int ensure_mask_scratch(crass_hip_ctx *c, uint64_t n_bits);
static int device_merge_enqueue(crass_hip_ctx *c, const char *dx_chars, const uint16_t *dx_len, uint64_t n_tok, const uint32_t *d_ntok, bool prepared);
static bool getenv_once_hl_host() { static const bool v = getenv("CRASS_HL_HOST_DEDUPE") != nullptr; return v; }      // A/B switch: the round-3 host de-duplication + merge for long reads

// CRASS_SURV_PROF: one line per category of reads (0 skipped without a hinted seed, 1 walked / nothing found, 2 found, 3 handed
// over) with the summed cycles of the kernel's phases, then a histogram of cycles per read; the counters are cleared
static void dump_surv_prof(crass_hip_ctx *c)
{
    if (!c->dp.prof) return;
    unsigned long long v[192];
    if (hipMemcpy(v, c->dp.prof, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return;
    (void)hipMemset(c->dp.prof, 0, sizeof v);
    { const unsigned long long big = ~0ull; (void)hipMemcpy(c->dp.prof + 188, &big, 8, hipMemcpyHostToDevice); }
    if (v[191]) fprintf(stderr, "[surv_prof] %llu waves; first start .. last start %.1f us, first start .. last end %.1f us\n", v[186], (double)(v[187] - v[188]) / 100.0, (double)(v[189] - v[188]) / 100.0);
    if (v[186] && v[186] <= 16384) {
        // per-wave start / end (10 ns ticks): when did the waves of the launch start, how long did they live
        std::vector<unsigned long long> se(2 * v[186]);
        if (hipMemcpy(se.data(), c->dp.prof + 192, se.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            std::vector<double> st, life, en;
            for (size_t k = 0; k < v[186]; k++) if (se[2 * k] && se[2 * k + 1]) { st.push_back((double)(se[2 * k] - v[188]) / 100.0); en.push_back((double)(se[2 * k + 1] - v[188]) / 100.0); life.push_back((double)(se[2 * k + 1] - se[2 * k]) / 100.0); }
            auto pct = [](std::vector<double> &a, double q) { if (a.empty()) return 0.0; std::sort(a.begin(), a.end()); return a[(size_t)(q * (a.size() - 1))]; };
            fprintf(stderr, "[surv_prof] per wave (us): start p10 %.0f p50 %.0f p90 %.0f max %.0f | life p10 %.0f p50 %.0f p90 %.0f max %.0f | end p10 %.0f p50 %.0f p90 %.0f max %.0f\n",
                    pct(st, .1), pct(st, .5), pct(st, .9), pct(st, 1.), pct(life, .1), pct(life, .5), pct(life, .9), pct(life, 1.), pct(en, .1), pct(en, .5), pct(en, .9), pct(en, 1.));
        }
        (void)hipMemset(c->dp.prof + 192, 0, 2 * 16384 * 8);
    }
    if (v[185]) fprintf(stderr, "[surv_prof] slowest read: slot %llu, %llu cycles\n", v[185] & 0xFFFFFFull, v[185] >> 24);
    if (v[191]) fprintf(stderr, "[surv_prof] waves' lifetimes: %llu shader cycles in %llu ticks of 10 ns: %.3f GHz\n", v[190], v[191], (double)v[190] / (10.0 * (double)v[191]));
    static const char *nm[] = {"total", "stage", "find", "scan", "extend", "qc", "hints", "out", "n_cand", "n_scanfind", "n_switch", "n_qc", "loop", "n_iter", "max", "reads"};
    for (int cat = 0; cat < 4; cat++) {
        const unsigned long long n = v[cat * 16 + 15];
        if (!n) continue;
        fprintf(stderr, "[surv_prof] cat %d", cat);
        for (int k = 0; k < 16; k++) fprintf(stderr, " %s %llu", nm[k], v[cat * 16 + k]);
        fprintf(stderr, "\n[surv_prof] cat %d hist(log2 cycles):", cat);
        for (int b = 0; b < 24; b++) if (v[64 + cat * 24 + b]) fprintf(stderr, " %d:%llu", b + 8, v[64 + cat * 24 + b]);
        fprintf(stderr, "\n");
    }
}

static int run_survivors(crass_hip_ctx *c, bool exc, uint64_t n_total, crass_hip_ctx::P1List &L,
                         const uint64_t *surv_idx_host)
{
    if (n_total == 0) return hint_wait_all(c);
    // Long reads: the Levenshtein fallback rows are sized for 256-base strings (spacers are a few dozen bases), which
    // is what lets several waves share a CU's LDS; a read that needs longer rows comes back with err == 6 and is
    // redone by a second launch with the uncapped layout.
    const SurvLds lds_full = survivor_lds_layout(c->max_len, c->dp);
    if (lds_full.total_bytes > 160 * 1024) return CRASS_ERR_UNSUPPORTED;
    const uint32_t row_cap = c->env.row_cap;            // (tests: CRASS_ROW_CAP forces the second launch)
    // Reads beyond 2 048 bases also get an ASCII WINDOW instead of room for the whole read (the byte-wise consumers only ever
    // read around the repeats of the candidate in hand: rh_ascii) and a start/stop list of 256 entries: 11.5 instead of 19.7 KB
    // of LDS per wave at 10 kbp — three waves per SIMD (with 168 registers) instead of two.  An array longer than the window
    // (> ~4.2 kbp), a 129th repeat or a string beyond the rows: the second launch.  CRASS_LONG_FULL_LAYOUT: the A/B switch.
    // (both read per call — once per seed scan of a long-read set —, so that a test can change them inside one process)
    const bool long_full = getenv("CRASS_LONG_FULL_LAYOUT") != nullptr;
    const uint32_t win_env = getenv("CRASS_SEQ_WINDOW") ? (uint32_t)atoi(getenv("CRASS_SEQ_WINDOW")) : 0u;      // (tests: a tiny window)
    const bool windowed = !exc && (win_env || (c->max_len > c->env.long_min && !long_full));
    const SurvLds lds = windowed ? survivor_lds_layout(c->max_len, c->dp, row_cap, win_env ? win_env : 4608u, 256u)
                                 : survivor_lds_layout(c->max_len, c->dp, row_cap);
    const bool capped = lds.row_elems != lds_full.row_elems || lds.seq_window != 0 || lds.ss_cap != lds_full.ss_cap;
    const uint64_t chunk_cap = std::min<uint64_t>(n_total, 1u << 20);
    const uint64_t ss_per = std::min<uint64_t>(lds_full.ss_cap, 64);
    const uint64_t pool_cap = std::min<uint64_t>(std::max<uint64_t>(chunk_cap * ss_per, 1u << 16), 1ull << 28);
    HIPCHK(c, c->d_surv.ensure(chunk_cap));
    HIPCHK(c, c->d_dr.ensure(chunk_cap * c->dr_stride));
    HIPCHK(c, c->d_ss_pool.ensure(pool_cap));
    HIPCHK(c, c->h_surv.ensure(chunk_cap));
    HIPCHK(c, c->h_dr.ensure(chunk_cap * c->dr_stride));
    // the survivor list on the device is 0, 1, 2, ... (no filter, no exception read, one chunk): the kernel is told so and needs no
    // look-up per read
    const bool ident_list = !exc && !surv_idx_host && c->R.n_exc == 0 && n_total <= chunk_cap;
    // ... and with position hints on top the walk is split: k_long_light for every read, the full kernel for what it hands over
    // (CRASS_NO_LIGHT: the A/B switch; the cut-short walks of CRASS_SURV_DEBUG belong to the full kernel)
    // ... or without position hints, for another window or seed lattice (-w / -d): k_long_light_any computes every position's bit itself
    const uint32_t sh0 = c->dp.lowDR + c->dp.lowSp, sh1 = c->dp.highDR + c->dp.highSp;
    const bool light_any = !c->R.pos_hint && c->max_len > c->env.long_min && !(c->dp.window == 8 && c->dp.skips == 8) && c->dp.window >= 6 && c->dp.window <= 9 &&
                           sh0 >= 17 && sh1 <= 127 && sh1 >= sh0 && !c->env.no_pos_hints;
    const bool use_light = ident_list && ((c->R.pos_hint && c->dp.skips == 8 && c->dp.window == 8) || light_any) && c->dp.debug_stop == 0 && !getenv("CRASS_NO_LIGHT");
    const int grid = 256 * 64;      // waves striding over the reads: 6 resident per CU at 10 kbp; 1 536 / 8 192 / 16 384 / 65 536 blocks: 11.5 / 8.6 / 8.2 / 9.1 ms
    const uint32_t stride = c->dr_stride;
    L.reserve(L.size() + n_total / 2 + 16, stride);
    for (uint64_t off = 0; off < n_total; off += chunk_cap) {
        const uint64_t nchunk = std::min(chunk_cap, n_total - off);
        HIPCHK(c, hipMemsetAsync(c->d_ss_used.p, 0, 4, c->stream));
        DevReads R = c->R;
        if (exc) {      // shift the exception window
            R.exc_read = c->R.exc_read + off; R.exc_off = c->R.exc_off + off; R.n_exc = nchunk;
        }
        // for the non-exception path the count lives on the device; chunking uses a host-known bound
        if (use_light) {
            HIPCHK(c, c->d_punt.ensure(nchunk + 1)); HIPCHK(c, hipMemsetAsync(c->d_punt.p, 0, 4, c->stream));
            HIPCHK(c, c->d_redo.ensure(nchunk + 1)); HIPCHK(c, hipMemsetAsync(c->d_redo.p, 0, 4, c->stream));
        }
        if (!exc && off == 0) HIPCHK(c, c->stamp(8, 1));
        if (!exc && c->hint_pending && off == 0 && nchunk == n_total) {
            // the walk, slice by slice behind the slice's hints (slot boundaries: the survivors with a read below the slice's end;
            // the list is ascending, and 0, 1, 2, ... when no host copy of it was made)
            uint64_t s0 = 0;
            for (int q = 0; q < c->hint_parts; q++) {
                const uint64_t r1 = c->hint_read_split[q + 1];
                uint64_t s1 = q + 1 == c->hint_parts ? nchunk
                            : (surv_idx_host ? (uint64_t)(std::lower_bound(surv_idx_host, surv_idx_host + nchunk, r1) - surv_idx_host) : std::min<uint64_t>(r1, nchunk));
                if (q > 0) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_hint[q], 0));
                if (s1 > s0 && use_light)
                    HIPCHK(c, launch_long_light(R, c->dp, c->d_count.p + 1, s1 - s0, c->d_surv.p + s0, s0, c->max_len, c->stream, c->d_punt.p + 1, c->d_punt.p));
                else if (s1 > s0)
                    HIPCHK(c, launch_survivor(R, c->dp, false, ident_list ? nullptr : c->d_idx.p + s0, c->d_count.p + 1, s1 - s0,
                                              c->d_surv.p + s0, c->d_dr.p + s0 * (size_t)stride, stride, c->d_ss_pool.p, (uint32_t)pool_cap, c->d_ss_used.p,
                                              c->d_found.p, c->hints_valid ? c->d_hit_info.p : nullptr, lds,
                                              (int)std::min<uint64_t>(grid, s1 - s0), c->stream, 0, s0, nchunk));
                s0 = s1;
            }
            c->hint_pending = false;
        } else if (use_light) {
            { const int hw = hint_wait_all(c); if (hw) return hw; }
            HIPCHK(c, launch_long_light(R, c->dp, c->d_count.p + 1, nchunk, c->d_surv.p, 0, c->max_len, c->stream, c->d_punt.p + 1, c->d_punt.p));
        } else {
            { const int hw = hint_wait_all(c); if (hw) return hw; }
            HIPCHK(c, launch_survivor(R, c->dp, exc, (exc || ident_list) ? nullptr : c->d_idx.p + off, c->d_count.p + (exc ? 0 : 1), nchunk,
                                      c->d_surv.p, c->d_dr.p, stride, c->d_ss_pool.p, (uint32_t)pool_cap, c->d_ss_used.p,
                                      c->d_found.p, (!exc && c->hints_valid) ? c->d_hit_info.p : nullptr, lds,
                                      (int)std::min<uint64_t>(grid, nchunk), c->stream));
        }
        // the reads the light walk handed over (an array, a candidate due for the QC: one read in 17 at 10 kbp): the full kernel
        if (use_light)
            HIPCHK(c, launch_survivor(R, c->dp, false, nullptr, c->d_count.p + 1, nchunk,
                                      c->d_surv.p, c->d_dr.p, stride, c->d_ss_pool.p, (uint32_t)pool_cap, c->d_ss_used.p,
                                      c->d_found.p, nullptr, lds, (int)std::min<uint64_t>(256 * 12, nchunk), c->stream, 7, 0, nchunk, c->d_punt.p + 1, c->d_punt.p,
                                      capped ? c->d_redo.p : nullptr));
        if (use_light && c->dp.prof) { HIPCHK(c, hipStreamSynchronize(c->stream)); dump_surv_prof(c); }      // (diagnostics: this launch's counters alone)
        if (capped && use_light)                    // (only the full kernel can have met such a read: its list)
            HIPCHK(c, launch_survivor(R, c->dp, false, nullptr, c->d_count.p + 1, nchunk,
                                      c->d_surv.p, c->d_dr.p, stride, c->d_ss_pool.p, (uint32_t)pool_cap, c->d_ss_used.p,
                                      c->d_found.p, nullptr, lds_full, (int)std::min<uint64_t>(256 * 2, nchunk), c->stream, 6, 0, nchunk, c->d_redo.p + 1, c->d_redo.p));
        else if (capped)
            HIPCHK(c, launch_survivor(R, c->dp, exc, (exc || ident_list) ? nullptr : c->d_idx.p + off, c->d_count.p + (exc ? 0 : 1), nchunk,
                                      c->d_surv.p, c->d_dr.p, stride, c->d_ss_pool.p, (uint32_t)pool_cap, c->d_ss_used.p,
                                      c->d_found.p, (!exc && c->hints_valid) ? c->d_hit_info.p : nullptr, lds_full,
                                      (int)std::min<uint64_t>(grid, nchunk), c->stream, 6));
        if (!exc && off == 0) HIPCHK(c, c->stamp(9, 1));
        // only the FOUND records travel (all 1 M reads of a 10 kbp set are survivors, 5 % are found; copying every slot
        // and every start/stop slot area was 340 MB and 10 of 39 ms per step): select -> ordered index list -> gather
        // into dense arrays with the start/stops packed -> three counters -> exact-size copies
        const uint64_t n_words = (nchunk + 63) / 64;
        HIPCHK(c, c->d_fidx.ensure(nchunk)); HIPCHK(c, c->g_surv.ensure(nchunk)); HIPCHK(c, c->g_dr.ensure(nchunk * (size_t)stride + 16));
        HIPCHK(c, c->g_ss.ensure(pool_cap)); HIPCHK(c, c->g_fidx.ensure(nchunk));
        HIPCHK(c, c->d_mask.ensure(n_words + 1)); HIPCHK(c, c->d_word_prefix.ensure(n_words + 1)); HIPCHK(c, c->d_block_sums.ensure((n_words + 255) / 256 + 2));
        HIPCHK(c, hipMemsetAsync(c->d_count.p + 2, 0, 8, c->stream));            // [2] found, [3] worst error
        HIPCHK(c, hipMemsetAsync(c->d_ss_used.p, 0, 4, c->stream));              // (reused: words of packed start/stops)
        HIPCHK(c, launch_select_found(c->d_surv.p, nchunk, c->d_mask.p, c->d_count.p + 3, c->stream));
        HIPCHK(c, launch_compact(c->d_mask.p, n_words, nchunk, c->d_word_prefix.p, c->d_block_sums.p, c->d_fidx.p, nchunk, c->d_count.p + 2, c->stream));
        // One chunk of a set without exception reads: the found records' strings are de-duplicated on the device as well (the
        // kernels of the dense sink's tail, queued behind the first wait with the exact count, beside the copies and the host
        // loop below) — the distinct list in token order + every candidate's index in it are what the DEVICE merge takes, so a
        // long-read set no longer hashes 50 k strings and clusters them on the host (1.3 ms per step at 1 M x 10 kbp)
        const bool ss16 = c->max_len < 65536;           // start/stops travel as 16-bit values
        const bool hl_dedupe = !exc && off == 0 && nchunk == n_total && c->R.n_exc == 0 && !c->xchg.active && !c->env.host_merge &&
                               c->prm.lowDRsize >= (int)kDevMinDR && stride <= 64 && (stride & 15u) == 0 && nchunk < (1u << 24) && !getenv_once_hl_host();
        if (hl_dedupe) HIPCHK(c, c->g_dr_len.ensure(nchunk));
        HIPCHK(c, launch_gather_sparse(c->d_fidx.p, c->d_count.p + 2, nchunk, c->d_surv.p, c->d_dr.p, stride, c->d_ss_pool.p, c->g_surv.p,
                                       c->g_fidx.p, c->g_dr.p, c->g_ss.p, (uint32_t)pool_cap, c->d_ss_used.p, c->stream,
                                       hl_dedupe ? c->g_dr_len.p : nullptr, ss16 ? 1 : 0));
        HIPCHK(c, hipMemcpyAsync(c->h_count.p + 2, c->d_count.p + 2, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_count.p + 6, c->d_ss_used.p, 4, hipMemcpyDeviceToHost, c->stream));
        const double tq0 = now_ms();
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const double tq1 = now_ms();
        dump_surv_prof(c);
        if (c->h_count.p[3] == 2) return CRASS_ERR_SEARCH_FATAL;
        if (c->h_count.p[3]) return CRASS_ERR_OVERFLOW;
        const uint64_t nf = c->h_count.p[2];
        const uint32_t used = c->h_count.p[6];
        if (nf > nchunk || used > pool_cap) return CRASS_ERR_OVERFLOW;
        HIPCHK(c, c->h_ss.ensure(used + 1)); HIPCHK(c, c->h_idx.ensure(nf + 1));
        // the copies: on the copy stream when the records' consumers can wait (one chunk, no exception list to merge with, slots =
        // reads: `cand` is then filled on request, materialize_cand) — the main stream goes on with the de-duplication, the
        // merge and pass 2 —, else on the main stream with the host loop right behind them
        static const bool sink_eager = getenv("CRASS_SINK_EAGER") != nullptr;      // A/B switch
        const bool lazy = !sink_eager && !exc && off == 0 && nchunk == n_total && c->R.n_exc == 0 && !surv_idx_host && &L == &c->cand && L.read.empty();
        if (nf) {
            hipStream_t cs = lazy ? c->copy_stream : c->stream;      // (the main stream is idle: it was waited for above)
            // lazy: on DMA engines where the runtime offers them — as blit kernels these 12 MB of PCIe stores stretched whatever
            // ran beside them (k_block_scan of the de-duplication: 138 us instead of 5, profiles/r05_timeline_c3.txt)
            // (the records' slots are gathered into a buffer of their own, g_fidx: d_fidx is written again by pass 2, and the copies
            // below may still be reading when it is queued — crass_hip_recruit used to wait for them, 0.5 ms of a long-read step)
            const void *srcs[4] = {c->g_surv.p, c->g_dr.p, c->g_fidx.p, c->g_ss.p};
            void *dsts[4] = {c->h_surv.p, c->h_dr.p, c->h_idx.p, c->h_ss.p};
            const size_t nbs[4] = {nf * sizeof(SurvOut), nf * (size_t)stride, nf * 8, used ? (size_t)used * (ss16 ? 2 : 4) : 0};
            static const bool sink_blit = getenv("CRASS_COPY_BLIT") != nullptr;
            for (int q = 0; q < 4; q++) {
                if (!nbs[q]) continue;
                if (lazy && !sink_blit) {
                    if (!c->dma_sink[q]) c->dma_sink[q] = sdma_create();
                    if (sdma_start(c->dma_sink[q], srcs[q], dsts[q], nbs[q])) continue;
                }
                HIPCHK(c, hipMemcpyAsync(dsts[q], srcs[q], nbs[q], hipMemcpyDeviceToHost, cs));
            }
            if (lazy) {
                if (!c->ev_sink_copies) HIPCHK(c, hipEventCreateWithFlags(&c->ev_sink_copies, hipEventDisableTiming));
                HIPCHK(c, hipEventRecord(c->ev_sink_copies, c->copy_stream));
                c->bulk_pending = true;
            }
            else HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        bool hl_queued = false, hl_premerge = false;
        if (nf && hl_dedupe) {
            // (behind the copies above; the host loop below runs beside these kernels; their small outputs land in pinned memory)
            uint32_t tsize = 1024;
            while (tsize < nf * 2) tsize <<= 1;
            HIPCHK(c, c->dd_keys.ensure(tsize)); HIPCHK(c, c->dd_first.ensure(tsize)); HIPCHK(c, c->dd_slot.ensure(nf)); HIPCHK(c, c->dd_rep.ensure(nf));
            HIPCHK(c, c->dd_hash.ensure(nf)); HIPCHK(c, c->hl_dx_idx.ensure(nf));
            HIPCHK(c, c->dd_dx_chars.ensure(nf * (size_t)stride + 16)); HIPCHK(c, c->dd_dx_len.ensure(nf));
            HIPCHK(c, c->h_dmap.ensure(nf)); HIPCHK(c, c->h_dx_chars.ensure(nf * (size_t)stride + 16)); HIPCHK(c, c->h_dx_len.ensure(nf)); HIPCHK(c, c->h_dx_hash.ensure(nf));
            { const int ms = ensure_mask_scratch(c, nf); if (ms) return ms; }
            HIPCHK(c, hipMemsetAsync(c->d_count.p + 4, 0, 8, c->stream));            // [4] distinct strings, [5] mismatch flag
            HIPCHK(c, launch_dr_dedupe(c->g_dr.p, c->g_dr_len.p, stride, c->d_count.p + 2, (uint32_t)nf, c->dd_keys.p, c->dd_first.p, tsize, c->dd_hash.p,
                                       c->dd_slot.p, c->dd_rep.p, c->stream, false));
            HIPCHK(c, launch_dx_tokens(c->g_dr.p, c->g_dr_len.p, c->dd_hash.p, stride, c->d_count.p + 2, (uint32_t)nf, c->dd_rep.p, c->dd_slot.p, c->dd_first.p,
                                       c->d_mask.p, c->d_word_prefix.p, c->d_block_sums.p, c->hl_dx_idx.p, c->d_count.p + 4, c->d_count.p + 5, c->h_dmap.p,
                                       c->h_dx_chars.p, c->h_dx_len.p, c->h_dx_hash.p, c->dd_dx_chars.p, c->dd_dx_len.p, c->stream,
                                       c->d_count.p, c->h_count.p, 8, nullptr));
            hl_queued = true;
            // the merge itself right behind (its kernels read the token count from the device and are sized for nf, a bound on
            // it): it runs while the host returns from this call and enters crass_hip_merge, which adopts it
            HIPCHK(c, hipEventRecord(c->ev_gathered, c->stream));
            c->premerge = 0; c->premerge_inflight = false;
            if (!c->env.no_speculation && nf <= (1u << 18)) {
                const int ps = device_merge_enqueue(c, c->dd_dx_chars.p, c->dd_dx_len.p, nf, c->d_count.p + 4, false);
                if (ps) return ps;
                HIPCHK(c, hipEventRecord(c->ev_premerge, c->stream));
                hl_premerge = true;
            }
        }
        const double tq2 = now_ms();
        auto fill = [c, nf, ss16, stride, exc, off, surv_idx_host, &L]() {
        if (c->wait_bulk()) return;                         // (the copies; a failed one is reported by whoever asks for the records)
        const SurvOut *so = c->h_surv.p;
        const char *drs = c->h_dr.p;
        const uint32_t *pool = c->h_ss.p;
        // append the chunk's records: offsets first (one pass), then every array filled in place by the host pool
        // (50 k records with 80 start/stops each at 10 kbp: 20 MB, 4 ms on one thread)
        const size_t base = L.read.size();
        L.read.resize(base + nf); L.low.resize(base + nf); L.replen.resize(base + nf); L.nss.resize(base + nf);
        L.ss_off.resize(base + nf); L.dr_len.resize(base + nf); L.dr.resize((base + nf) * (size_t)stride);
        uint64_t at = L.ss.size();
        for (uint64_t q = 0; q < nf; q++) { L.ss_off[base + q] = at; at += so[q].n_ss; }
        L.ss.resize(at);
        const size_t per_task = 2048;
        host_parallel_for((nf + per_task - 1) / per_task, 16, [&](size_t t) {
            const uint64_t q1 = std::min<uint64_t>(nf, (t + 1) * per_task);
            for (uint64_t q = t * per_task; q < q1; q++) {
                const SurvOut &o = so[q];
                const uint64_t k = c->h_idx.p[q];                // the record's slot in the chunk
                L.read[base + q] = c->read_base + (exc ? c->h_exc_read[off + k] : (surv_idx_host ? surv_idx_host[off + k] : off + k));
                L.low[base + q] = o.low_lexi;
                L.replen[base + q] = o.repeat_len;
                L.nss[base + q] = o.n_ss;
                if (ss16) { const uint16_t *p16 = reinterpret_cast<const uint16_t *>(pool) + o.ss_off; uint32_t *d = L.ss.data() + L.ss_off[base + q]; for (uint32_t i = 0; i < o.n_ss; i++) d[i] = p16[i]; }
                else memcpy(L.ss.data() + L.ss_off[base + q], pool + o.ss_off, (size_t)o.n_ss * 4);
                L.dr_len[base + q] = o.dr_len;
                memcpy(L.dr.data() + (base + q) * (size_t)stride, drs + q * stride, stride);
            }
        });
        };
        if (lazy && nf) { c->cand_fill = fill; c->cand_pending_n = nf; }      // (L is c->cand: it outlives this call)
        else fill();
        if (hl_queued) {
            HIPCHK(c, hipEventSynchronize(c->ev_gathered));      // (the de-duplication's counters; the merge behind it may still run)
            c->premerge_inflight = hl_premerge;
            if (c->h_count.p[5] == 0 && c->h_count.p[2] == nf && c->h_count.p[4] > 0) {
                c->n_dx = c->h_count.p[4];
                c->have_dev_tokens = true; c->hl_tokens = true;
                c->dx_hash_valid = true;
                if (hl_premerge) c->premerge = 2;
            }
        }
        if (c->env.merge_profile)
            fprintf(stderr, "[crass_sink] survivors %llu: kernel+D2H wait %.3f ms, pool D2H %.3f ms, host loop %.3f ms\n",
                    (unsigned long long)nchunk, tq1 - tq0, tq2 - tq1, now_ms() - tq2);
    }
    return CRASS_OK;
}

// A long-read set (position hints: no per-read filter, every read is a survivor) takes the host-loop sink; crass scans a read
// set ONCE, so what that sink and the stages behind it allocate on their first call is allocated with the reads — best effort:
// whatever cannot be had now is had then, as before (the first step of a 1 M x 10 kbp set: 17.4 ms against 10.3 in steady
// state; the guesses: one found read in eight, 128 start/stops per found read)
static int device_merge_prepare(crass_hip_ctx *c, const char *dx_chars, const uint16_t *dx_len, uint64_t n_tok, const uint32_t *d_ntok);
static void presize_hostloop(crass_hip_ctx *c)
{
    if (!c->R.pos_hint || c->max_len <= c->env.long_min || c->env.no_presize || c->R.n_reads == 0) return;
    const SurvLds lds_full = survivor_lds_layout(c->max_len, c->dp);
    if (lds_full.total_bytes > 160 * 1024) return;
    for (auto &d : c->dma_sink) if (!d) d = sdma_create();      // (an engine's queue is created by its first copy, ~6 ms: here)
    const uint64_t chunk_cap = std::min<uint64_t>(c->R.n_reads, 1u << 20);
    const uint64_t ss_per = std::min<uint64_t>(lds_full.ss_cap, 64);
    const uint64_t pool_cap = std::min<uint64_t>(std::max<uint64_t>(chunk_cap * ss_per, 1u << 16), 1ull << 28);
    const uint32_t stride = c->dr_stride;
    const uint64_t nf = std::min<uint64_t>(chunk_cap, std::max<uint64_t>(65536, chunk_cap / 8));
    const uint64_t n_words = (chunk_cap + 63) / 64;
    uint32_t tsize = 1024;
    while (tsize < nf * 2) tsize <<= 1;
    bool ok = true;
    auto need = [&](hipError_t e) { if (e != hipSuccess) ok = false; };
    need(c->d_punt.ensure(chunk_cap + 1)); need(c->d_redo.ensure(chunk_cap + 1));
    need(c->d_surv.ensure(chunk_cap)); need(c->d_dr.ensure(chunk_cap * stride)); need(c->d_ss_pool.ensure(pool_cap));
    need(c->h_surv.ensure(chunk_cap)); need(c->h_dr.ensure(chunk_cap * stride));
    need(c->d_fidx.ensure(chunk_cap)); need(c->g_surv.ensure(chunk_cap)); need(c->g_dr.ensure(chunk_cap * (size_t)stride + 16)); need(c->g_ss.ensure(pool_cap));
    need(c->g_fidx.ensure(chunk_cap));
    need(c->d_mask.ensure(n_words + 1)); need(c->d_word_prefix.ensure(n_words + 1)); need(c->d_block_sums.ensure((n_words + 255) / 256 + 2));
    need(c->g_dr_len.ensure(chunk_cap)); need(c->h_ss.ensure(nf * 128 + 1)); need(c->h_idx.ensure(nf + 1));
    need(c->dd_keys.ensure(tsize)); need(c->dd_first.ensure(tsize)); need(c->dd_slot.ensure(nf)); need(c->dd_rep.ensure(nf));
    need(c->dd_hash.ensure(nf)); need(c->hl_dx_idx.ensure(nf)); need(c->dd_dx_chars.ensure(nf * (size_t)stride + 16)); need(c->dd_dx_len.ensure(nf));
    need(c->h_dmap.ensure(nf)); need(c->h_dx_chars.ensure(nf * (size_t)stride + 16)); need(c->h_dx_len.ensure(nf)); need(c->h_dx_hash.ensure(nf));
    if (ok && c->prm.lowDRsize >= (int)kDevMinDR && stride <= 64 && !c->env.host_merge && !c->xchg.active)
        (void)device_merge_prepare(c, c->dd_dx_chars.p, c->dd_dx_len.p, std::min<uint64_t>(nf, 1u << 18), nullptr);      // (the merge's tables: buffers only, nothing is launched)
    c->last_hip = 0;
}

// 2^26 survivors = 100 M+ read shards at the ~2 % filter pass rate; the slot pool stays below 2^31 words
static const uint64_t kDenseMaxSurvivors = 1ull << 26;

// Fast path of the pass-1 sink: one chunk, fixed start/stop slots, no exception reads.  Returns
// CRASS_ERR_STATE when it does not apply (the caller then uses the host-loop path).
static uint64_t hit_bound(uint64_t n_hits)             // the same for pass 2's flagged reads
{
    return std::max<uint64_t>(4096, (n_hits + n_hits / 2 + 4095) & ~4095ull);
}

static uint64_t survivor_bound(uint64_t n_surv)        // the speculative bound learnt from a call with n_surv survivors
{
    return std::min<uint64_t>(std::max<uint64_t>(65536, (n_surv + n_surv / 2 + 65535) & ~65535ull), 1ull << 26);
}

// n_surv: number of filter survivors, or (speculative mode: d_nsurv = the compaction's device-side count, no host
// round trip before this call) an upper bound for it.  The exact count then arrives with the final copy of
// the counters; *overflow is set when it exceeds the bound (nothing of this call is valid then).
static int device_merge_prepare(crass_hip_ctx *c, const char *dx_chars, const uint16_t *dx_len, uint64_t n_tok, const uint32_t *d_ntok);
static int device_merge_enqueue(crass_hip_ctx *c, const char *dx_chars, const uint16_t *dx_len, uint64_t n_tok, const uint32_t *d_ntok, bool prepared = false);

// every buffer the dense pass-1 sink touches for up to n_alloc survivors (crass_hip_load_reads sizes them for the first
// call's bound, so that the first seed scan of a context does not allocate; ensure() is a no-op when a buffer is large enough)
// The mask / prefix / block-sum scratch is sized for the READ count at load; the three-kernel compactions over survivor and
// hit SLOTS (CRASS_NO_LOOKBACK, or after a look-back give-up) write (bound + 63) / 64 words of it, and a slot bound is not
// bounded by the read count (70 000 reads with 50 000 survivors: bound 131 072 = 2 048 words into 1 095).  Whoever sizes
// slot buffers for a bound grows the scratch with them.  (Its contents are dead between stages: a stage's mask is consumed by
// the compaction queued right behind it, and the release inside ensure() waits for the device.)
static int ensure_mask_scratch(crass_hip_ctx *c, uint64_t n_bits)
{
    const uint64_t n_words = (n_bits + 63) / 64;
    if (c->d_mask.n >= n_words + 1 && c->d_word_prefix.n >= n_words + 1 && c->d_block_sums.n >= (n_words + 255) / 256 + 2) return CRASS_OK;
    HIPCHK(c, c->d_mask.ensure(n_words + 1));
    HIPCHK(c, c->d_word_prefix.ensure(n_words + 1));
    HIPCHK(c, c->d_block_sums.ensure((n_words + 255) / 256 + 2));
    return CRASS_OK;
}

static int ensure_dense_buffers(crass_hip_ctx *c, uint64_t n_alloc, uint64_t pool_cap, const SurvLds &lds, bool dedupe)
{
    crass_hip_ctx::P1Dense &D = c->dense;
    const uint32_t stride = c->dr_stride;
    { const int ms = ensure_mask_scratch(c, n_alloc); if (ms) return ms; }
    HIPCHK(c, c->d_surv.ensure(n_alloc));
    HIPCHK(c, c->d_dr.ensure(n_alloc * stride));
    HIPCHK(c, c->d_ss_pool.ensure(std::max<uint64_t>(pool_cap, n_alloc * (uint64_t)lds.ss_cap)));
    HIPCHK(c, c->d_fidx.ensure(n_alloc));
    HIPCHK(c, D.d_dr_len.ensure(n_alloc + 8)); HIPCHK(c, D.d_dr.ensure(n_alloc * stride + 16));
    const uint32_t ss_elem = c->max_len <= 256 ? 1u : 2u;            // read positions fit a byte
    HIPCHK(c, D.h_blob.ensure(p1_blob_layout(n_alloc, lds.ss_cap, ss_elem).total + 64));
    HIPCHK(c, D.d_blob.ensure(D.h_blob.n));
    if (dedupe) {
        uint32_t tsize_alloc = 1024;
        while (tsize_alloc < n_alloc * 2) tsize_alloc <<= 1;
        HIPCHK(c, c->dd_keys.ensure(tsize_alloc)); HIPCHK(c, c->dd_first.ensure(tsize_alloc));
        HIPCHK(c, c->dd_slot.ensure(n_alloc)); HIPCHK(c, c->dd_hash.ensure(n_alloc));
        HIPCHK(c, c->dd_rep.ensure(n_alloc)); HIPCHK(c, c->h_rep.ensure(n_alloc)); HIPCHK(c, c->h_hash.ensure(n_alloc));
        HIPCHK(c, c->h_dmap.ensure(n_alloc)); HIPCHK(c, c->h_dx_chars.ensure(n_alloc * stride + 16)); HIPCHK(c, c->h_dx_len.ensure(n_alloc));
        HIPCHK(c, c->h_dx_hash.ensure(n_alloc));
        HIPCHK(c, c->dd_dx_chars.ensure(n_alloc * stride + 16)); HIPCHK(c, c->dd_dx_len.ensure(n_alloc));
        if (c->dx_via_dma) HIPCHK(c, c->dd_map.ensure(n_alloc));
    }
    return CRASS_OK;
}

static int dense_after_sync(crass_hip_ctx *c, uint64_t n_surv, uint32_t ss_cap, uint32_t ss_elem, bool dedupe, bool premerge_queued, bool *overflow);
// defer: the launch is queued and the function returns WITHOUT waiting for it (c->p1d holds what dense_after_sync needs) — a rank
// of a multi-rank job, whose exchange and merge can be queued behind pass 1 before the host has seen a count (p1_finish)
static int run_survivors_dense(crass_hip_ctx *c, uint64_t n_surv, const uint32_t *d_nsurv, bool *overflow, bool defer = false)
{
    *overflow = false;
    { const int hw = hint_wait_all(c); if (hw) return hw; }
    const SurvLds lds = survivor_lds_layout(c->max_len, c->dp);
    if (lds.total_bytes > 160 * 1024) return CRASS_ERR_UNSUPPORTED;
    const uint32_t stride = c->dr_stride;
    const uint64_t pool_cap = std::max<uint64_t>(n_surv * (uint64_t)lds.ss_cap, 1u << 16);
    if (n_surv == 0 || n_surv > kDenseMaxSurvivors || pool_cap >= (1ull << 31) || lds.ss_cap > 128 || (stride & 15)) return CRASS_ERR_STATE;
    crass_hip_ctx::P1Dense &D = c->dense;
    // buffers are sized for the bound the NEXT call will speculate with (1.5 x this call's count, see
    // crass_hip_seed_scan), so that the second call of a context does not re-allocate everything
    const bool speculative = (d_nsurv == c->d_count.p);          // n_surv is already such a bound
    const uint64_t n_alloc = speculative ? n_surv : std::max<uint64_t>(n_surv, survivor_bound(n_surv));
    const uint32_t ss_elem = c->max_len <= 256 ? 1u : 2u;            // read positions fit a byte
    const bool dedupe = n_surv < (1u << 24);
    const bool defer_ok = defer && dedupe && speculative && c->xchg.active;
    { const int as = ensure_dense_buffers(c, n_alloc, pool_cap, lds, dedupe); if (as) return as; }
    // [2] = found count, [3] = worst error, [4] = n distinct, [5] = de-duplication mismatch flag
    if (!speculative) {                                 // (speculative launch: cleared by the filter's compaction, see seed scan)
        HIPCHK(c, hipMemsetAsync(c->d_count.p + 2, 0, 20, c->stream));      // ([6]: the lane kernel's punt count)
        HIPCHK(c, hipMemsetAsync(c->d_ss_used.p, 0, 4, c->stream));
    }
    // The merge that will follow this stage is sized here already when the previous call's merge ran on the device (its
    // kernels read the token count from the device and are sized by a bound): the survivor kernel then clears the
    // merge's tables on its way, and the merge is queued right behind pass 1's tail further down.
    c->dm_prepared_n = 0; c->dm_prepared_src = nullptr;
    const DevMerge *init_merge = nullptr;
    if (speculative && dedupe && c->prm.lowDRsize >= (int)kDevMinDR && stride <= 64 && !c->env.host_merge && !c->env.no_speculation && !c->env.dm_init_late) {
        const char *src = nullptr; const uint16_t *src_len = nullptr; uint64_t bound = 0;
        if (!c->xchg.active && c->dm_prev_local && c->dx_cap_hint) {
            HIPCHK(c, c->dd_dx_chars.ensure(n_alloc * stride + 16)); HIPCHK(c, c->dd_dx_len.ensure(n_alloc));
            src = c->dd_dx_chars.p; src_len = c->dd_dx_len.p; bound = c->dx_cap_hint;
        } else if (c->xchg.active && c->xchg.gx_cap_hint && c->xchg.gx_cap_hint <= c->xchg.world * c->xchg.cap) {
            const uint64_t n_max = c->xchg.world * c->xchg.cap;
            HIPCHK(c, c->dm.gx_chars.ensure((size_t)n_max * stride + 16)); HIPCHK(c, c->dm.gx_len.ensure(n_max));
            src = c->dm.gx_chars.p; src_len = c->dm.gx_len.p; bound = c->xchg.gx_cap_hint;
        }
        if (src) {
            const int ps = device_merge_prepare(c, src, src_len, bound, c->d_count.p + 4);
            if (ps) return ps;
            init_merge = &c->dm.M;
        }
    }
    HIPCHK(c, c->stamp(8, 1));
    // lane-per-read kernel for uniform short reads; whatever it punts (err == 4) and every other
    // layout goes through the wave-per-read kernel
    static const bool no_hints_dbg = getenv("CRASS_SURV_NO_HINTS") != nullptr;      // (timing experiments only: the lanes walk every lattice seed)
    const uint32_t *hints = (c->hints_valid && !no_hints_dbg) ? c->d_hit_info.p : nullptr;
    const bool no_lanes = c->env.no_lane_kernel;
    hipError_t le = no_lanes ? hipErrorNotSupported
                             : launch_survivor_lanes(c->R, c->dp, c->d_idx.p, d_nsurv, n_surv, c->d_surv.p, c->d_dr.p, stride,
                                                     c->d_ss_pool.p, lds.ss_cap, c->d_found.p, hints, c->stream, init_merge, c->max_len, c->d_count.p + 6);
    if (le != hipSuccess && le != hipErrorNotSupported) { c->last_hip = (int)le; return CRASS_ERR_HIP; }
    if (le == hipSuccess && init_merge) { c->dm_prepared_n = init_merge->n_tok; c->dm_prepared_src = init_merge->dx_chars; }
    // Reads the lane kernel does not take (513 .. 2 048 bases): the long reads' LIGHT walk over the survivor list first — nearly every
    // survivor of such a set is a chance seed hit that ends without a candidate (a quarter of 1 000-base reads pass a 7-base window's
    // filter) — and the full wave kernel only for what it hands over, from the seed it stopped at.  Position hints of the default
    // lattice: k_long_light on them; another window or lattice: k_long_light_any, which works the bits out as it walks.
    bool light_done = false;
    if (le == hipErrorNotSupported && !no_lanes && c->dp.debug_stop == 0 && !c->env.no_dense_light && c->max_len <= 12288 && c->dp.window >= 6 && c->dp.window <= 9) {
        DevReads RL = c->R;
        const bool lattice = c->R.pos_hint && !c->R.hint_all && c->dp.skips == 8 && c->dp.window == 8;
        if (!lattice && !c->R.hint_all) RL.pos_hint = nullptr;
        HIPCHK(c, c->d_punt.ensure(n_surv + 1)); HIPCHK(c, hipMemsetAsync(c->d_punt.p, 0, 4, c->stream));
        const hipError_t ll = launch_long_light(RL, c->dp, d_nsurv, n_surv, c->d_surv.p, 0, c->max_len, c->stream, c->d_punt.p + 1, c->d_punt.p, c->d_idx.p);
        if (ll == hipSuccess) {
            light_done = true;
            HIPCHK(c, launch_survivor(c->R, c->dp, false, c->d_idx.p, d_nsurv, n_surv, c->d_surv.p, c->d_dr.p, stride,
                                      c->d_ss_pool.p, (uint32_t)pool_cap, c->d_ss_used.p, c->d_found.p, hints, lds,
                                      (int)std::min<uint64_t>(256 * 12, n_surv), c->stream, 7, 0, 0, c->d_punt.p + 1, c->d_punt.p));
        } else if (ll != hipErrorNotSupported) { c->last_hip = (int)ll; return CRASS_ERR_HIP; }
    }
    if (!light_done)
    HIPCHK(c, launch_survivor(c->R, c->dp, false, c->d_idx.p, d_nsurv, n_surv, c->d_surv.p, c->d_dr.p, stride,
                              c->d_ss_pool.p, (uint32_t)pool_cap, c->d_ss_used.p, c->d_found.p, hints, lds,
                              (int)std::min<uint64_t>(256 * 32, n_surv), c->stream, le == hipSuccess ? 4 : 0, 0, 0, nullptr,
                              le == hipSuccess ? c->d_count.p + 6 : nullptr));      // (the lane kernel's punt count: none -> the launch leaves at once)
    if (c->R.n_exc)                                     // exception reads in the list (err == 5): raw bytes, same slots
        HIPCHK(c, launch_survivor(c->R, c->dp, true, c->d_idx.p, d_nsurv, n_surv, c->d_surv.p, c->d_dr.p, stride,
                                  c->d_ss_pool.p, (uint32_t)pool_cap, c->d_ss_used.p, c->d_found.p, nullptr, lds,
                                  (int)std::min<uint64_t>(256 * 32, n_surv), c->stream, 5));
    HIPCHK(c, c->stamp(9, 1));
    const uint64_t n_words = (n_surv + 63) / 64;
    uint32_t tsize = 1024;
    if (dedupe) while (tsize < n_surv * 2) tsize <<= 1;
    // The table holds the DISTINCT strings.  Sized for the survivor slots (the only bound that cannot fail) it is 200 MB of
    // clearing per step at 100 M reads for 42 k strings; a speculative launch sizes it from the bound on the distinct
    // strings the merge is sized with, and an insert that runs out of probes reports the miss (h_count[5] bit 2): the host
    // de-duplicates that call itself and the context goes back to the full-size table.
    if (dedupe && speculative && c->dx_cap_hint && !c->dd_full_table && !c->env.dd_full_table) {
        uint32_t small = c->env.test_bounds[1] ? 64u : 4096u;          // (tests: a table that does overflow)
        while (small < c->dx_cap_hint * 4ull && small < tsize) small <<= 1;
        tsize = std::min(tsize, small);
    }
    Lookback lbf;
    if (const Lookback *lb = c->next_lookback_elems(n_surv, &lbf)) {
        // found flags -> ranks in one pass (also clears the de-duplication table)
        HIPCHK(c, launch_found_compact(c->d_surv.p, d_nsurv, n_surv, c->d_count.p + 3, dedupe ? c->dd_keys.p : nullptr,
                                       dedupe ? c->dd_first.p : nullptr, tsize, c->d_fidx.p, c->d_count.p + 2, *lb, c->stream));
    } else {
        HIPCHK(c, launch_found_mask(c->d_surv.p, d_nsurv, n_surv, c->d_mask.p, c->d_count.p + 3, c->stream,
                                    dedupe ? c->dd_keys.p : nullptr, dedupe ? c->dd_first.p : nullptr, tsize));
        HIPCHK(c, launch_compact(c->d_mask.p, n_words, n_surv, c->d_word_prefix.p, c->d_block_sums.p, c->d_fidx.p, n_surv, c->d_count.p + 2, c->stream));
    }
    // the gather assembles the hand-off blob and inserts every candidate's DR string into the de-duplication table
    HIPCHK(c, launch_gather_found(c->d_fidx.p, c->d_count.p + 2, n_surv, c->d_surv.p, c->d_idx.p, c->read_base, c->d_dr.p, stride,
                                  c->d_ss_pool.p, lds.ss_cap, ss_elem, D.d_blob.p, D.d_dr_len.p, D.d_dr.p, c->stream,
                                  dedupe ? c->dd_keys.p : nullptr, dedupe ? c->dd_first.p : nullptr, tsize, c->dd_hash.p, c->dd_slot.p,
                                  c->d_count.p + 5));
    // de-duplication and token ranks follow without a host round trip (the found count stays on the device);
    // their small outputs — all the merge needs — are written straight into pinned host memory
    c->have_rep = false;
    c->have_dev_tokens = false;
    if (dedupe) {
        // distinct strings in first-occurrence order and every candidate's rank among them, exact
        Lookback lbd;
        const bool dxd = c->dx_via_dma;                 // the list stays on the device; DMA engines bring it over (below)
        HIPCHK(c, launch_dx_tokens(D.d_dr.p, D.d_dr_len.p, c->dd_hash.p, stride, c->d_count.p + 2, (uint32_t)n_surv, c->dd_rep.p, c->dd_slot.p, c->dd_first.p, c->d_mask.p,
                                   c->d_word_prefix.p, c->d_block_sums.p, c->d_fidx.p, c->d_count.p + 4, c->d_count.p + 5, dxd ? c->dd_map.p : c->h_dmap.p,
                                   dxd ? nullptr : c->h_dx_chars.p, dxd ? nullptr : c->h_dx_len.p, dxd ? nullptr : c->h_dx_hash.p, c->dd_dx_chars.p, c->dd_dx_len.p, c->stream,
                                   c->d_count.p, defer_ok ? c->h_count.p + 16 : c->h_count.p, 8,           // (the counters leave with the last kernel: no copy call)
                                   c->next_lookback_tiles((n_surv + 1023) / 1024, &lbd)));
        if (c->xchg.active)                             // multi-rank: the list goes straight into the collective's send buffer
            HIPCHK(c, launch_xg_fill(c->dd_dx_chars.p, c->dd_dx_len.p, c->d_count.p + 4, stride, c->xchg.cap, c->xchg.slot, c->xchg.send.p, c->stream,
                                     defer_ok ? c->d_count.p : nullptr, n_surv,
                                     (defer_ok && c->poll_on) ? c->h_flags.p : nullptr, (defer_ok && c->poll_on) ? (c->want_p1 = ++c->flag_seq) : 0u));
    }
    if (!dedupe) HIPCHK(c, hipMemcpyAsync(c->h_count.p, c->d_count.p, 32, hipMemcpyDeviceToHost, c->stream));
    bool premerge_queued = false;
    c->premerge = 0;
    D.wide_ready = false; D.dr_fallback = false;
    D.pack_ss_cap = lds.ss_cap;
    // The merge itself is queued here too when the previous call's merge ran on the device: its kernels read the token
    // count from the device (d_count[4]) and are sized by a bound; crass_hip_merge adopts the result if the counts fit.
    const bool will_premerge = speculative && dedupe && !c->xchg.active && c->dm_prev_local && c->dx_cap_hint && c->prm.lowDRsize >= (int)kDevMinDR && stride <= 64 &&
                               !c->env.host_merge && !c->env.no_speculation;
    // (what the host waits for when the merge is queued behind pass 1, or pass 1 deferred: a stage flag — else this event)
    if (!c->poll_on && (will_premerge || defer_ok)) HIPCHK(c, hipEventRecord(c->ev_gathered, c->stream));
    if (will_premerge) {
        const bool prepared = c->dm_prepared_n == c->dx_cap_hint && c->dm_prepared_src == c->dd_dx_chars.p;
        const int ps = device_merge_enqueue(c, c->dd_dx_chars.p, c->dd_dx_len.p, c->dx_cap_hint, c->d_count.p + 4, prepared);
        if (ps) return ps;
        premerge_queued = true;
        // (only the fall-back order of the hand-off copy — no DMA engine — waits for this event)
        if (!c->poll_on || !c->dma) HIPCHK(c, hipEventRecord(c->ev_premerge, c->stream));
    }
    host_pool_warm();                                   // the merge follows: wake the host workers while the device finishes
    if (defer_ok) {
        c->p1d.active = true; c->p1d.n_bound = n_surv; c->p1d.ss_cap = lds.ss_cap; c->p1d.ss_elem = ss_elem;
        return CRASS_OK;
    }
    return dense_after_sync(c, n_surv, lds.ss_cap, ss_elem, dedupe, premerge_queued, overflow);
}

// the host's half of the dense pass-1 tail: waits for the kernels queued by run_survivors_dense, reads the counters they left in
// pinned memory, starts the hand-off copies
static int dense_after_sync(crass_hip_ctx *c, uint64_t n_surv, uint32_t ss_cap, uint32_t ss_elem, bool dedupe, bool premerge_queued, bool *overflow)
{
    crass_hip_ctx::P1Dense &D = c->dense;
    const uint32_t stride = c->dr_stride;
    // (with the merge queued behind it, the host waits for pass 1 only — the event recorded above — and goes on to
    // adopt the merge and queue pass 2 while the merge kernels run)
    if (premerge_queued || c->p1d.finishing) {
        if (!c->poll_on) HIPCHK(c, hipEventSynchronize(c->ev_gathered));
        else if (!c->poll_flag(c->p1d.finishing ? 0 : 1, c->p1d.finishing ? c->want_p1 : c->want_pre, 200.0)) HIPCHK(c, hipStreamSynchronize(c->stream));
    } else HIPCHK(c, hipStreamSynchronize(c->stream));
    c->premerge_inflight = premerge_queued;
    c->t_p1_sync = now_ms();
    const uint32_t *hc = c->p1_counts();
    if (hc[0] > n_surv) { *overflow = true; return CRASS_OK; }
    const uint64_t nf = hc[2];
    const uint32_t err = hc[3];
    if (err == 1) return CRASS_ERR_SEARCH_FATAL;
    if (err) return CRASS_ERR_OVERFLOW;
    D.lay = p1_blob_layout(nf, ss_cap, ss_elem);
    if (nf) {
        // the hand-off records: the host has waited for the gather, the copy runs beside what follows.  As PCIe stores from
        // shader waves (the runtime's blit kernel or ours) it cost the kernels beside it its own 0.2-0.3 ms wherever it was
        // placed — k_dm_pack_codes 0.29 instead of 0.02 ms, or k_dm_greedy 0.27 instead of 0.06, or pass 2's filter 1.32
        // instead of 1.11 (profiles/NOTES_r03.md).  The fall-back order keeps it away from the merge, whose kernels are
        // short dependent chains of agent-scope loads and atomics.
        static const bool blit = getenv("CRASS_COPY_BLIT") != nullptr;      // A/B switch: the runtime's copy (a blit kernel over every CU)
        if (!blit && sdma_start(c->dma, D.d_blob.p, D.h_blob.p, D.lay.total)) {
            // (a DMA engine: nothing of it runs on the CUs, it starts now)
        } else {
            if (premerge_queued && !getenv_once_copy_early()) HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_premerge, 0));
            if (blit) HIPCHK(c, hipMemcpyAsync(D.h_blob.p, D.d_blob.p, D.lay.total, hipMemcpyDeviceToHost, c->copy_stream));
            else HIPCHK(c, launch_copy_to_host(D.d_blob.p, D.h_blob.p, D.lay.total, c->copy_stream));
        }
        c->bulk_pending = true;
    }
    if (nf && !dedupe) {                                // no distinct list: the candidates' own strings travel
        HIPCHK(c, D.h_dr_fb.ensure(nf * stride + 16)); HIPCHK(c, D.h_dr_len_fb.ensure(nf + 8));
        HIPCHK(c, hipMemcpyAsync(D.h_dr_fb.p, D.d_dr.p, nf * stride, hipMemcpyDeviceToHost, c->copy_stream));
        HIPCHK(c, hipMemcpyAsync(D.h_dr_len_fb.p, D.d_dr_len.p, nf * 2, hipMemcpyDeviceToHost, c->copy_stream));
        D.dr_fallback = true;
        c->bulk_pending = true;
    }
    if (nf) {
        if (dedupe) {
            if (hc[5] == 0) {
                c->n_dx = hc[4];
                c->have_dev_tokens = true;
                c->dx_hash_valid = !c->dx_via_dma;
                if (c->dx_via_dma) {
                    // (the host has waited for the kernels that wrote these: the engines start at once, beside the merge)
                    std::lock_guard<std::mutex> lk(c->dx_mu);
                    const void *src[3] = {c->dd_map.p, c->dd_dx_chars.p, c->dd_dx_len.p};
                    void *dst[3] = {c->h_dmap.p, c->h_dx_chars.p, c->h_dx_len.p};
                    const size_t bytes[3] = {(size_t)nf * 4, (size_t)c->n_dx * stride, (size_t)c->n_dx * 2};
                    for (int q = 0; q < 3; q++) {
                        if (!bytes[q] || sdma_start(c->dma_dx[q], src[q], dst[q], bytes[q])) continue;
                        HIPCHK(c, hipMemcpyAsync(dst[q], src[q], bytes[q], hipMemcpyDeviceToHost, c->copy_stream));
                        c->dx_stream_copy = true;
                    }
                    c->dx_pending = true;
                }
                if (premerge_queued && c->n_dx > 0 && c->n_dx <= c->dx_cap_hint) c->premerge = 2;
                else if (premerge_queued && c->n_dx > c->dx_cap_hint) c->n_bound_overflows[1]++;     // (crass_hip_merge launches its own)
            } else if (hc[5] & 2u) {
                // the de-duplication table was sized for fewer distinct strings than there are: nothing of it is usable.  The
                // candidates' own strings travel, the host de-duplicates them, and later calls size the table for the slots.
                c->dd_full_table = true;
                c->n_bound_overflows[1]++;
                HIPCHK(c, D.h_dr_fb.ensure(nf * stride + 16)); HIPCHK(c, D.h_dr_len_fb.ensure(nf + 8));
                HIPCHK(c, hipMemcpyAsync(D.h_dr_fb.p, D.d_dr.p, nf * stride, hipMemcpyDeviceToHost, c->copy_stream));
                HIPCHK(c, hipMemcpyAsync(D.h_dr_len_fb.p, D.d_dr_len.p, nf * 2, hipMemcpyDeviceToHost, c->copy_stream));
                D.dr_fallback = true;
                c->bulk_pending = true;
            } else {
                // (hash collision among the candidates: the host merge wants the first-occurrence map)
                HIPCHK(c, D.h_dr_fb.ensure(nf * stride + 16)); HIPCHK(c, D.h_dr_len_fb.ensure(nf + 8));
                HIPCHK(c, hipMemcpyAsync(D.h_dr_fb.p, D.d_dr.p, nf * stride, hipMemcpyDeviceToHost, c->copy_stream));
                HIPCHK(c, hipMemcpyAsync(D.h_dr_len_fb.p, D.d_dr_len.p, nf * 2, hipMemcpyDeviceToHost, c->copy_stream));
                D.dr_fallback = true;
                HIPCHK(c, hipMemcpyAsync(c->h_rep.p, c->dd_rep.p, nf * 4, hipMemcpyDeviceToHost, c->copy_stream));
                HIPCHK(c, hipMemcpyAsync(c->h_hash.p, c->dd_hash.p, nf * 8, hipMemcpyDeviceToHost, c->copy_stream));
                c->bulk_pending = true;
                c->have_rep = true;
            }
        }
    } else D.lay.total = 0;
    D.n = nf;
    D.active = true;
    return CRASS_OK;
}

// every buffer pass 2's tail touches for up to h_alloc flagged reads (+ n_exc exception reads scanned byte-wise)
static int ensure_recruit_buffers(crass_hip_ctx *c, uint64_t h_alloc, uint64_t n_exc, bool dev_sink, bool anchors)
{
    const uint64_t s_alloc = h_alloc + n_exc;
    { const int ms = ensure_mask_scratch(c, h_alloc); if (ms) return ms; }
    HIPCHK(c, c->d_rec.ensure(s_alloc + 1));
    HIPCHK(c, c->d_dr.ensure((s_alloc + 1) * c->dr_stride));
    if (!dev_sink) {                           // host sink only
        HIPCHK(c, c->h_rec.ensure(s_alloc + 1));
        HIPCHK(c, c->h_dr.ensure((s_alloc + 1) * c->dr_stride));
        HIPCHK(c, c->h_idx.ensure(h_alloc + 1));
    }
    if (anchors) {
        HIPCHK(c, c->d_slot_info.ensure(h_alloc + 1));
        HIPCHK(c, c->d_slot_pid.ensure(h_alloc + 1));
    }
    if (dev_sink) {
        HIPCHK(c, c->h_qblob.ensure(p2_blob_layout(h_alloc).total + 64));
        HIPCHK(c, c->d_fidx.ensure(h_alloc + 1));
    }
    return CRASS_OK;
}

// ------------------------------------------------------------------------------------------
// The first call of a context.  The three speculation bounds (survivors, distinct DR strings, flagged reads) are
// normally learnt from the previous call; a crass run scans each read set ONCE, so crass_hip_load_reads sets them from
// the read set itself — the expected pass rate of the seed filter on random sequence plus an allowance for real
// arrays — and sizes every pool for them.  The first seed scan / merge / recruit then neither allocates nor waits
// for a count before queueing the next stage; a bound that turns out too small repeats the stage with the exact
// count, exactly like a learnt one.  CRASS_NO_PRESIZE / CRASS_NO_SPECULATION switch this off.
// ------------------------------------------------------------------------------------------
static int first_call_bounds_impl(crass_hip_ctx *c);
static int first_call_bounds(crass_hip_ctx *c)
{
    // best effort: a pool that cannot be sized now (out of memory, a cap) just means the first call sizes things itself, as
    // every call did before — the load itself has succeeded
    const int s = first_call_bounds_impl(c);
    if (s == CRASS_ERR_OOM) {
        c->surv_cap_hint = 0; c->hit_cap_hint = 0; c->dx_cap_hint = 0; c->dm_prev_local = false; c->premerge = 0;
        c->last_hip = 0;
        c->dm.release();                                // (the merge tables sized for the bound: the first call sizes them exactly)
        c->dm_prepared_n = 0; c->dm_prepared_src = nullptr;
        return CRASS_OK;
    }
    return s;
}
static int first_call_bounds_impl(crass_hip_ctx *c)
{
    c->surv_cap_hint = 0; c->hit_cap_hint = 0; c->dx_cap_hint = 0; c->dm_prev_local = false; c->premerge = 0;
    c->recruit_exact = false;
    const uint64_t n = c->R.n_reads;
    if (c->env.no_presize || c->env.no_speculation || n == 0 || c->max_len > c->env.long_min) return CRASS_OK;
    if (c->R.n_exc && c->env.exc_separate) return CRASS_OK;
    const DevParams &P = c->dp;
    const SurvLds lds = survivor_lds_layout(c->max_len, P);
    if (lds.total_bytes > 160 * 1024 || lds.ss_cap > 128 || (c->dr_stride & 15)) return CRASS_OK;
    // seed filter on random sequence: every lattice seed has ~(highDR+highSp-lowDR-lowSp+1) candidate positions, each
    // an equal w-mer with probability 4^-w (libcrispr.cpp:295-339); + 2 % of the reads for real arrays; x 1.5
    const uint32_t min_len = P.lowDR + P.lowSp + P.window + 1;
    const double seeds = c->max_len > min_len ? (double)(c->max_len - min_len) / P.skips + 1.0 : 1.0;
    const double cand = (double)(P.highDR + P.highSp - P.lowDR - P.lowSp + 1);
    double rate = seeds * cand / (double)(1ull << (2 * P.window)) + 0.02;
    if (rate > 1.0) rate = 1.0;
    uint64_t surv = survivor_bound((uint64_t)((double)n * rate));
    if (surv > n + 65536) surv = survivor_bound(n * 2 / 3);       // (1.5 x inside: the bound never needs to exceed n)
    if (c->env.test_bounds[0]) surv = std::max<uint64_t>(64, c->env.test_bounds[0]);
    const uint64_t pool_cap = std::max<uint64_t>(surv * (uint64_t)lds.ss_cap, 1u << 16);
    if (surv > kDenseMaxSurvivors || pool_cap >= (1ull << 31)) return CRASS_OK;
    // keep the pools a small part of the device: ~(20 + 2*stride + 4*ss_cap + 100) bytes per survivor slot
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return CRASS_OK;
    const uint64_t per = 140ull + 3ull * c->dr_stride + 6ull * lds.ss_cap;
    if (surv * per > free_b / 4) return CRASS_OK;
    const bool dedupe = surv < (1u << 24);
    int s = ensure_dense_buffers(c, surv, pool_cap, lds, dedupe);
    if (s) return s;
    c->surv_cap_hint = surv;
    // look-back status words for the largest launch of a step (the masks over all reads), cleared once
    { const uint64_t nw = (n + 63) / 64; c->prealloc_lookback(std::max<uint64_t>((nw + lookback_tile_words(nw) - 1) / lookback_tile_words(nw), (surv + 1023) / 1024)); }
    // distinct DR strings: a few hundred per million reads on metagenome-like input; the device merge is queued
    // behind pass 1 for this many
    if (dedupe && c->prm.lowDRsize >= (int)kDevMinDR && c->dr_stride <= 64 && !c->env.host_merge) {
        uint64_t dx = crass_hip_exchange_rows_for(n);
        if (c->env.test_bounds[1]) dx = std::max<uint64_t>(16, c->env.test_bounds[1]);
        s = device_merge_prepare(c, c->dd_dx_chars.p, c->dd_dx_len.p, dx, c->d_count.p + 4);
        if (s) return s;
        c->dx_cap_hint = (uint32_t)dx;
        c->dm_prev_local = true;
        // the host view of that merge (token arena, per-candidate tokens, pattern list, flat groups) is built by the helper thread
        // while pass 2 runs; on a context's first call its containers grew and faulted their pages in there — 2.2 ms instead of
        // 1.1 at 100 M reads, 0.5 ms of it past the end of pass 2.  They are grown and touched here, with the reads
        // (MergeResult::clear keeps the capacity).
        try {
            auto touch = [](auto &v, size_t k) { v.resize(k); v.clear(); };
            MergeResult &m = c->merge;
            touch(m.cand_token, (size_t)(n / 128 + 4096));
            touch(m.tokens.strings.chars, (size_t)dx * 48); touch(m.tokens.strings.off, (size_t)dx + 2); m.tokens.strings.clear();
            touch(m.patterns.chars, (size_t)dx * 24); touch(m.patterns.off, (size_t)dx + 2); m.patterns.clear();
            touch(m.pat_group, (size_t)dx); touch(m.pat_token, (size_t)dx); touch(m.grp_tokens, (size_t)dx); touch(m.grp_off, (size_t)dx / 4 + 16);
        } catch (const std::bad_alloc &) { }
    }
    // reads flagged by the anchor probe: reads of arrays that pass 1 did not find (+ ~0 false positives)
    uint64_t hits = hit_bound(n / 64);
    if (c->env.test_bounds[2]) hits = std::max<uint64_t>(16, c->env.test_bounds[2]);
    s = ensure_recruit_buffers(c, hits, 0, true, true);
    if (s) return s;
    c->hit_cap_hint = hits;
    // The wave-per-read survivor kernel spills a few registers, i.e. needs scratch memory: the runtime sets a queue's scratch up
    // at the first dispatch that asks for it — the device sat idle for ~160 us in front of that launch and the launch itself
    // took 130 us instead of 20 (a fresh context's step at 100 M reads: 4.77-4.87 ms against 4.45-4.55 for its later steps,
    // profiles/r06_single_shot_timeline.txt).  An empty launch of the same shape here, with the reads, takes that off the step.
    if (!c->env.no_warm_launch) {
        HIPCHK(c, hipMemsetAsync(c->d_count.p, 0, 32, c->stream));
        HIPCHK(c, launch_survivor(c->R, c->dp, false, c->d_idx.p, c->d_count.p + 1, surv, c->d_surv.p, c->d_dr.p, c->dr_stride,
                                  c->d_ss_pool.p, (uint32_t)pool_cap, c->d_ss_used.p, c->d_found.p, nullptr, lds,
                                  (int)std::min<uint64_t>(256 * 32, surv), c->stream, 4));
    }
    host_pool_warm();
    return CRASS_OK;
}

int crass_hip_seed_scan(crass_hip_ctx *c)
{
    if (!c) return CRASS_ERR_INVALID_ARG;
    if (!c->have_reads) return CRASS_ERR_STATE;
    (void)hipSetDevice(c->device);
    (void)c->wait_bulk();               // (copies of the previous step: their records are dropped below)
    (void)c->wait_dx();                 // (... and the distinct list's: the engines read buffers this scan rewrites)
    c->bulk_status = CRASS_OK; c->dx_status = CRASS_OK;
    quiesce_worker(c);
    c->have_pass1 = c->have_merge = c->have_pass2 = false;
    c->dm.active = false;
    c->premerge_inflight = false;
    c->p1d.active = false;                  // (a deferred pass 1 nobody finished: its kernels are ahead of this scan's on the stream, its results dropped)
    const bool defer_p1 = c->p1d.enabled && !c->p1d.force_sync && c->xchg.active;
    c->p1d.force_sync = false;
    // (the survivor kernel clears the merge's words, x_* included: behind whatever export of a merge nobody adopted)
    if (c->dm.view_launched) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->dm.ev_view, 0)); c->dm.view_launched = false; }
    const uint64_t n = c->R.n_reads;
    const uint64_t n_words = (n + 63) / 64;
    bool found_cleared = false;                 // (the fixed-range filter kernel clears the found flags on its way)
    HIPCHK(c, c->stamp(0, 1));
    // step 1: filter.  Reads longer than 2 kbp almost surely contain a spurious lattice hit
    // (P ~ seeds*49/4^w), so the filter is skipped and every read goes to the survivor kernel.
    bool fast = false, hint_filtered = false;
    const bool use_filter = c->max_len <= c->env.long_min;
    // exception reads (a byte outside ACGT) join the survivor list and are evaluated byte-wise in place, so that
    // the dense pass-1 path also holds for inputs with a few N reads
    c->dp.exc_survive = (use_filter && c->R.n_exc > 0 && !c->env.exc_separate) ? 1u : 0u;
    if (use_filter) {
        hipError_t fe = hipErrorNotSupported;
        // (uniform STRIDE is what the bit-parallel kernel needs; the lengths may differ — trimmed reads padded to one stride)
        static const bool found_memset = getenv("CRASS_FOUND_MEMSET") != nullptr;      // A/B switch: the fill kernel in front of the scan
        if (c->R.stride_words >= 4 && c->R.stride_words <= 16)
            fe = launch_filter_fast(c->R, c->dp, c->d_mask.p, c->d_hit_info.p, c->stream, found_memset ? nullptr : c->d_found.p, &found_cleared);
        if (fe != hipSuccess) found_cleared = false;
        if (fe == hipSuccess) fast = true;
        else if (fe == hipErrorNotSupported && c->hint_filter && c->R.pos_hint && !c->hint_filter_any) {
            // no lane-per-read filter for this layout: one hint bit per lattice position (the long reads' kernel), which also flags the reads that have one.
            // The survivor kernel walks on the same bits
            HIPCHK(c, hipMemsetAsync(c->d_mask.p, 0, n_words * 8, c->stream));
            const hipError_t he = launch_hint_positions(c->R, c->dp, c->d_pos_hint_off.p, c->pos_hint_blk ? c->d_pos_hint_blk.p : nullptr, c->n_pos_hint_words,
                                                        c->d_pos_hint.p, c->stream, 0, c->n_pos_hint_words, c->d_mask.p);
            if (he != hipSuccess) { c->last_hip = (int)he; return CRASS_ERR_HIP; }
            c->hint_pending = false;
            hint_filtered = true;
        }
        else if (fe == hipErrorNotSupported && c->hint_filter_any && c->n_pos_hint_words) {
            HIPCHK(c, hipMemsetAsync(c->d_mask.p, 0, n_words * 8, c->stream));
            const hipError_t he = launch_hint_filter_any(c->R, c->dp, c->d_pos_hint_off.p, c->pos_hint_blk ? c->d_pos_hint_blk.p : nullptr, c->n_pos_hint_words,
                                                         c->d_mask.p, c->d_pos_hint.p, c->stream);
            c->hint_pending = false;
            if (he != hipSuccess) { c->last_hip = (int)he; return CRASS_ERR_HIP; }
            hint_filtered = true;
        }
        else if (fe == hipErrorNotSupported) { HIPCHK(c, launch_filter_general(c->R, c->dp, c->d_mask.p, c->max_len, c->stream)); }
        else { c->last_hip = (int)fe; return CRASS_ERR_HIP; }
    } else {
        // all non-exception reads survive: mask = ~exc_mask (exc_mask is 32-bit words of the same bit order)
        HIPCHK(c, hipMemsetAsync(c->d_mask.p, 0xFF, n_words * 8, c->stream));
        if (c->R.pos_hint) {
            const uint32_t *blk = c->pos_hint_blk ? c->d_pos_hint_blk.p : nullptr;
            if (c->hint_parts > 1) { HIPCHK(c, hipEventRecord(c->ev_hint_go, c->stream)); HIPCHK(c, hipStreamWaitEvent(c->hint_stream, c->ev_hint_go, 0)); }
            for (int q = 0; q < c->hint_parts; q++) {
                hipStream_t hs = q == 0 ? c->stream : c->hint_stream;
                hipError_t he = launch_hint_positions(c->R, c->dp, c->d_pos_hint_off.p, blk, c->n_pos_hint_words, c->d_pos_hint.p, hs,
                                                      c->hint_word_split[q], c->hint_word_split[q + 1]);
                if (he != hipSuccess) { c->last_hip = (int)he; return CRASS_ERR_HIP; }
                if (q > 0) HIPCHK(c, hipEventRecord(c->ev_hint[q], c->hint_stream));
            }
            c->hint_pending = c->hint_parts > 1;
        }
    }
    c->hints_valid = fast;
    if (!found_cleared) HIPCHK(c, hipMemsetAsync(c->d_found.p, 0, n + 1, c->stream));      // (in front of the survivor kernels, which set them)
    HIPCHK(c, c->stamp(1, 1));
    // step 2: ordered compaction
    // (the scan kernel also clears the survivor stage's counters: d_count[2..6) and the start/stop pool cursor)
    Lookback lbs;
    HIPCHK(c, launch_compact(c->d_mask.p, n_words, n, c->d_word_prefix.p, c->d_block_sums.p, c->d_idx.p, n, c->d_count.p, c->stream,
                             c->d_count.p + 2, 5, c->d_ss_used.p, 1, c->next_lookback(n_words, &lbs)));      // ([2..6) + [6], the lane kernel's punt count)
    HIPCHK(c, c->stamp(2, 2));
    c->dense.active = false;
    c->have_rep = false;
    c->have_dev_tokens = false; c->hl_tokens = false;
    c->have_distinct = false;
    c->cand_fill = nullptr; c->cand_pending_n = 0;      // (a fill nobody asked for: its records are dropped)
    c->cand.clear();
    const double t_sink0 = now_ms();
    uint64_t n_surv = 0;
    int s = CRASS_ERR_STATE;
    // Speculative fast path: the survivor stage is launched right behind the compaction with the survivor
    // count still on the device, sized by a bound learnt from the previous call (twice its count).  If the
    // bound turns out too small the stage is simply redone below with the exact count.
    const bool exc_ok = c->R.n_exc == 0 || c->dp.exc_survive;
    if (use_filter && exc_ok && c->surv_cap_hint && !c->env.no_speculation) {
        bool overflow = false;
        HIPCHK(c, c->stamp(3, 2));
        s = run_survivors_dense(c, c->surv_cap_hint, c->d_count.p, &overflow, defer_p1);
        if (s != CRASS_OK && s != CRASS_ERR_STATE) return s;
        if (s == CRASS_OK && c->p1d.active) {
            // queued up to the send buffer's fill kernel, not waited for: p1_finish does the rest when the caller has queued the
            // exchange and crass_hip_merge_gathered its kernels (or when anybody asks for pass 1's results)
            c->p1d.fast = fast; c->p1d.hint_filtered = hint_filtered;
            HIPCHK(c, c->stamp(4, 2));
            return CRASS_OK;
        }
        if (s == CRASS_OK) {
            n_surv = c->h_count.p[0];
            if (overflow) c->n_bound_overflows[0]++;
            if (overflow || n_surv == 0) {              // redo below with the exact count; undo the attempt's found flags
                c->dense.active = false;
                s = CRASS_ERR_STATE;
                HIPCHK(c, hipMemsetAsync(c->d_found.p, 0, n + 1, c->stream));
            }
        } else {                                        // nothing was launched (layout limits): plain path
            HIPCHK(c, hipMemcpyAsync(c->h_count.p, c->d_count.p, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            n_surv = c->h_count.p[0];
        }
    } else if (!use_filter && c->R.n_exc == 0) {
        // nothing filters and no read is an exception: every read survives and the compaction's list is 0, 1, 2, ... —
        // the survivor kernel is queued right behind it, no host round trip (0.7 ms of an idle device at 1 M x 10 kbp:
        // the wait, a host-built copy of the list and its 8 MB upload)
        n_surv = n;
    } else {
        HIPCHK(c, hipMemcpyAsync(c->h_count.p, c->d_count.p, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        n_surv = c->h_count.p[0];
    }
    const bool spec_done = (s == CRASS_OK);
    const bool try_dense = !spec_done && use_filter && exc_ok && n_surv > 0 && n_surv <= kDenseMaxSurvivors;
    std::vector<uint64_t> surv_idx;
    if (!spec_done) {
        // the survivor kernel reads its count from d_count[1] (chunk-local bound is passed separately)
        uint32_t big = 0xFFFFFFFFu;
        c->h_count.p[1] = big;
        HIPCHK(c, hipMemcpyAsync(c->d_count.p + 1, c->h_count.p + 1, 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, c->stamp(3, 2));
        bool overflow = false;
        s = try_dense ? run_survivors_dense(c, n_surv, c->d_count.p + 1, &overflow) : CRASS_ERR_STATE;
        if (s != CRASS_OK && s != CRASS_ERR_STATE) return s;
    }
    if (use_filter && exc_ok) {                         // bound for the next call: 1.5 x this call's count
        c->surv_cap_hint = survivor_bound(n_surv);
    }
    if (s == CRASS_ERR_STATE) {
        // host-loop path: label records with their read index from a host copy of the survivor list
        const bool identity = !use_filter && c->R.n_exc == 0;      // (see above: the list on the device is 0, 1, 2, ...)
        if (!identity) surv_idx.resize(n_surv);
        if (identity) {
        } else if (n_surv && use_filter) {
            HIPCHK(c, c->h_idx.ensure(n_surv));
            HIPCHK(c, hipMemcpyAsync(c->h_idx.p, c->d_idx.p, n_surv * 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            memcpy(surv_idx.data(), c->h_idx.p, n_surv * 8);
        } else if (n_surv) {
            // no filter: every non-exception read survives, in order
            size_t e = 0, w = 0;
            for (uint64_t r = 0; r < n; r++) {
                while (e < c->h_exc_read.size() && c->h_exc_read[e] < r) e++;
                if (e < c->h_exc_read.size() && c->h_exc_read[e] == r) continue;
                surv_idx[w++] = r;
            }
            surv_idx.resize(w);
            n_surv = w;
            if (n_surv) HIPCHK(c, hipMemcpyAsync(c->d_idx.p, surv_idx.data(), n_surv * 8, hipMemcpyHostToDevice, c->stream));
        }
        s = run_survivors(c, false, n_surv, c->cand, identity ? nullptr : surv_idx.data());
        if (s) return s;
    }
    if (c->R.n_exc && !c->dense.active) {               // (the dense path evaluated them in place)
        crass_hip_ctx::P1List el;
        s = run_survivors(c, true, c->R.n_exc, el, nullptr);
        if (s) return s;
        if (el.size()) {
            // merge the two ascending lists by read index (exception reads are rare)
            crass_hip_ctx::P1List &a = c->cand, m;
            m.reserve(a.size() + el.size(), c->dr_stride);
            size_t ia = 0, ib = 0;
            while (ia < a.size() || ib < el.size()) {
                const bool takeA = ib >= el.size() || (ia < a.size() && a.read[ia] < el.read[ib]);
                const crass_hip_ctx::P1List &src = takeA ? a : el;
                const size_t k = takeA ? ia++ : ib++;
                m.read.push_back(src.read[k]); m.low.push_back(src.low[k]); m.replen.push_back(src.replen[k]);
                m.nss.push_back(src.nss[k]); m.ss_off.push_back(m.ss.size());
                m.ss.insert(m.ss.end(), src.ss.begin() + src.ss_off[k], src.ss.begin() + src.ss_off[k] + src.nss[k]);
                m.dr_len.push_back(src.dr_len[k]);
                m.dr.insert(m.dr.end(), src.dr.begin() + k * c->dr_stride, src.dr.begin() + (k + 1) * c->dr_stride);
            }
            c->cand = std::move(m);
        }
    }
    HIPCHK(c, c->stamp(4, 2));
    if (!(c->premerge_inflight && (c->dense.active || c->hl_tokens))) HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t total = (size_t)c->n_cand();
    c->have_pass1 = true;
    if (c->xchg.active && !(c->dense.active && c->have_dev_tokens)) {
        // no device-resident distinct list (exception reads, no candidates, ...): the send buffer is filled from the host's
        ensure_distinct(c);
        const bool dev = false;
        (void)dev;
        const uint64_t nd = c->dx_len.size();
        std::vector<uint8_t> buf(c->xchg.send_bytes(), 0);
        reinterpret_cast<uint64_t *>(buf.data())[0] = nd;
        reinterpret_cast<uint32_t *>(buf.data())[2] = c->dr_stride;
        reinterpret_cast<uint32_t *>(buf.data())[3] = (uint32_t)c->xchg.cap;
        for (uint64_t j = 0; j < nd && j < c->xchg.cap; j++) {
            uint8_t *row = buf.data() + (j + 1) * (size_t)c->xchg.slot;
            memcpy(row, c->dx_chars.data() + j * (size_t)c->dr_stride, c->dr_stride);
            const uint32_t l = c->dx_len[j];
            memcpy(row + c->dr_stride, &l, 4);
        }
        HIPCHK(c, hipMemcpy(c->xchg.send.p, buf.data(), buf.size(), hipMemcpyHostToDevice));
    }
    c->cnt.ms_sink_host = (float)(now_ms() - t_sink0);     // includes the survivor kernel + D2H it waits for
    c->cnt.n_filter_survivors = n_surv + (c->dp.exc_survive ? 0 : c->R.n_exc);
    c->cnt.n_pass1_found = total;
    c->cnt.used_fast_filter = fast ? 1 : (hint_filtered ? 2 : 0);
    // (the event spans are evaluated when the counters are fetched: each query costs microseconds of host time
    // between two stages)
    c->spans_p1 = true; c->span_survivors = n_surv != 0;
    return c->lookback_ok();
}

// The host's half of a deferred pass 1 (p1d).  CRASS_OK: the context is where a synchronous crass_hip_seed_scan leaves it.
// kP1Redo: the speculative launch cannot be used — k_xg_fill marked the send buffer from the same counters, so every rank of the
// job sees an exchange that did not fit and repeats the step; this context's next seed scan runs synchronously.
static constexpr int kP1Redo = 1000;
static int p1_finish(crass_hip_ctx *c)
{
    if (!c->p1d.active) return CRASS_OK;
    c->p1d.active = false;
    c->p1d.finishing = true;
    bool overflow = false;
    const int s = dense_after_sync(c, c->p1d.n_bound, c->p1d.ss_cap, c->p1d.ss_elem, true, false, &overflow);
    const uint64_t n_surv = c->h_count.p[16];
    c->p1d.finishing = false;
    if (s != CRASS_OK) { c->p1d.force_sync = true; (void)hipStreamSynchronize(c->stream); return s; }
    if (overflow) c->n_bound_overflows[0]++;
    c->surv_cap_hint = survivor_bound(n_surv);
    const bool usable = !overflow && n_surv != 0 && c->dense.active && (c->dense.n == 0 || c->have_dev_tokens);
    if (!usable) {
        c->dense.active = false; c->have_dev_tokens = false;
        c->p1d.force_sync = true;
        HIPCHK(c, hipStreamSynchronize(c->stream));       // (whatever was queued behind the launch worked on nothing usable)
        return kP1Redo;
    }
    c->have_pass1 = true;
    c->cnt.ms_sink_host = 0;
    c->cnt.n_filter_survivors = n_surv + (c->dp.exc_survive ? 0 : c->R.n_exc);
    c->cnt.n_pass1_found = c->n_cand();
    c->cnt.used_fast_filter = c->p1d.fast ? 1 : (c->p1d.hint_filtered ? 2 : 0);
    c->spans_p1 = true; c->span_survivors = n_surv != 0;
    return c->lookback_ok();
}
// every entry point that reads pass 1's results, other than crass_hip_merge_gathered: a launch that cannot be used is repeated
// here, synchronously (the send buffer is refilled; a caller that had already gathered it sees the redo mark in what it gathered)
static int settle_p1(crass_hip_ctx *c)
{
    if (!c->p1d.active) return CRASS_OK;
    const int s = p1_finish(c);
    if (s == kP1Redo) return crass_hip_seed_scan(c);
    return s;
}

int crass_hip_exchange_set_deferred(crass_hip_ctx *c, int on)
{
    if (!c) return CRASS_ERR_INVALID_ARG;
    if (!on) { const int s = settle_p1(c); if (s) return s; }
    c->p1d.enabled = on != 0 && !c->env.no_speculation && getenv("CRASS_NO_DEFER_P1") == nullptr;
    return CRASS_OK;
}

int crass_hip_get_candidates(const crass_hip_ctx *c, crass_candidates *o)
{
    if (!c || !o) return CRASS_ERR_INVALID_ARG;
    if (c->p1d.active) { const int ds = settle_p1(const_cast<crass_hip_ctx *>(c)); if (ds) return ds; }
    if (!c->have_pass1) return CRASS_ERR_STATE;
    if (const int bs = c->wait_bulk()) return bs;
    if (c->dense.active) {
        const crass_hip_ctx::P1Dense &D = c->dense;
        c->widen_p1();
        const uint8_t *hb = D.h_blob.p;
        o->n = D.n; o->read_idx = (const uint64_t *)(hb + D.lay.read); o->low_lexi = hb + D.lay.low; o->repeat_len = D.w_replen.data();
        o->n_ss = D.w_nss.data(); o->ss_off = D.w_ss_off.data(); o->ss_pool = D.w_ss.data();
        o->dr_stride = c->dr_stride; o->dr_len = D.w_dr_len.data(); o->dr_chars = D.w_dr.data();
    } else {
        c->materialize_cand();
        o->n = c->cand.size();
        o->read_idx = c->cand.read.data(); o->low_lexi = c->cand.low.data(); o->repeat_len = c->cand.replen.data();
        o->n_ss = c->cand.nss.data(); o->ss_off = c->cand.ss_off.data(); o->ss_pool = c->cand.ss.data();
        o->dr_stride = c->dr_stride; o->dr_len = c->cand.dr_len.data(); o->dr_chars = c->cand.dr.data();
    }
    o->max_read_len = c->max_len;
    return CRASS_OK;
}

// ------------------------------------------------------------------------------------------
// merge + patterns
// ------------------------------------------------------------------------------------------
static int install_patterns(crass_hip_ctx *c, const StringArena &pats)
{
    c->n_installed_patterns = (uint32_t)pats.size();
    c->have_patterns = false;
    quiesce_worker(c);
    c->dm.active = false;                               // the installed set is the host-built one from here on
    c->cnt.n_patterns = (uint32_t)pats.size();
    if (pats.empty()) { c->cnt.ac_states = 0; return CRASS_OK; }
    for (size_t i = 0; i < pats.size(); i++) if (pats.len(i) == 0 || pats.len(i) > 255) return CRASS_ERR_UNSUPPORTED;
    HostAutomaton &H = c->H;
    HostAnchors HK;
    const bool prof = c->env.merge_profile;
    const double tb0 = now_ms();
    build_automaton_and_anchors(H, HK, pats);
    const double tb2 = now_ms();
    DevAutomaton A{};
    A.n_states = H.n_states; A.n_sym1 = H.n_sym1;
    A.max_pat_len = 0;
    for (size_t i = 0; i < pats.size(); i++) A.max_pat_len = std::max<uint32_t>(A.max_pat_len, (uint32_t)pats.len(i));
    memcpy(A.sym, H.sym, 256);
    (void)hipSetDevice(c->device);
    // the packed-read scans need the 4-column table only; the full byte-symbol table goes up on demand
    // (ensure_full_automaton: full scans without anchors, exception reads)
    c->full_automaton_uploaded = false;
    if (H.n_states <= 65535) {
        HIPCHK(c, c->a_go4.ensure(H.go4.size()));
        HIPCHK(c, hipMemcpyAsync(c->a_go4.p, H.go4.data(), H.go4.size() * 2, hipMemcpyHostToDevice, c->stream));
        A.go4 = c->a_go4.p; A.acgt_ok = 1;
    } else {
        HIPCHK(c, c->a_go4w.ensure(H.go4w.size()));
        HIPCHK(c, hipMemcpyAsync(c->a_go4w.p, H.go4w.data(), H.go4w.size() * 4, hipMemcpyHostToDevice, c->stream));
        A.go4w = c->a_go4w.p; A.acgt_ok = 0;
    }
    HIPCHK(c, c->a_out.ensure(H.out_len.size()));
    HIPCHK(c, hipMemcpyAsync(c->a_out.p, H.out_len.data(), H.out_len.size() * 2, hipMemcpyHostToDevice, c->stream));
    A.out_len = c->a_out.p;
    HIPCHK(c, c->a_out_pid.ensure(H.out_pid.size()));
    HIPCHK(c, hipMemcpyAsync(c->a_out_pid.p, H.out_pid.data(), H.out_pid.size() * 4, hipMemcpyHostToDevice, c->stream));
    A.out_pid = c->a_out_pid.p;
    c->have_pat_token = false;
    c->A = A;
    // anchor keys for the pass-2 fast path
    c->have_anchors = false;
    if (HK.ok) {
        HIPCHK(c, c->a_anchor.ensure(HK.table.size()));
        HIPCHK(c, hipMemcpyAsync(c->a_anchor.p, HK.table.data(), HK.table.size() * 4, hipMemcpyHostToDevice, c->stream));
        c->K.table = c->a_anchor.p; c->K.log_size = HK.log_size; c->K.mode = HK.mode; c->K.m1 = HK.m1;
        c->K.m2 = HK.m2; c->K.n_keys = HK.n_keys;
        c->have_anchors = true;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));        // HK's table is a local
    c->have_patterns = true;
    c->cnt.ac_states = H.n_states;
    if (prof) fprintf(stderr, "[crass_merge] automaton + anchors %.3f ms, uploads %.3f ms\n", tb2 - tb0, now_ms() - tb2);
    return CRASS_OK;
}

// full byte-symbol transition table: only the scans that cannot use the 4-column table need it
static int ensure_full_automaton(crass_hip_ctx *c)
{
    if (c->full_automaton_uploaded) return CRASS_OK;
    const HostAutomaton &H = c->H;
    if (H.n_states <= 65535) {
        std::vector<uint16_t> g16(H.go.size());
        for (size_t i = 0; i < H.go.size(); i++) g16[i] = (uint16_t)H.go[i];
        HIPCHK(c, c->a_go16.ensure(g16.size()));
        HIPCHK(c, hipMemcpyAsync(c->a_go16.p, g16.data(), g16.size() * 2, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));     // g16 is a local
        c->A.go16 = c->a_go16.p;
    } else {
        HIPCHK(c, c->a_go32.ensure(H.go.size()));
        HIPCHK(c, hipMemcpyAsync(c->a_go32.p, H.go.data(), H.go.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->A.go32 = c->a_go32.p;
    }
    c->full_automaton_uploaded = true;
    return CRASS_OK;
}

static int finish_merge(crass_hip_ctx *c, double t0);
static int build_host_merge(crass_hip_ctx *c);

// ---- the merge on the device (dmerge.hip) ----
static bool device_merge_applies(const crass_hip_ctx *c)
{
    if (c->env.host_merge) return false;                 // A/B switch: force the host merge (merge.cpp)
    return c->have_pass1 && c->dev_tokens() && c->prm.lowDRsize >= (int)kDevMinDR &&
           c->dr_stride <= 64 && c->n_dx <= (1u << 20);
}

// dx_*: distinct strings in token order on the device; hx_*: the same list in pinned host memory
// the kernels only (n_tok: the token count, or — with d_ntok — the bound the launch is sized for while the count is
// still on the device); the context's state is untouched until device_merge_commit
// buffers + the kernels' argument block for a merge over (up to) n_tok tokens; nothing is launched
static int device_merge_prepare(crass_hip_ctx *c, const char *dx_chars, const uint16_t *dx_len, uint64_t n_tok, const uint32_t *d_ntok)
{
    crass_hip_ctx::DM &d = c->dm;
    const uint32_t n = (uint32_t)n_tok, stride = c->dr_stride;
    if (!d.ev_done) {
        HIPCHK(c, hipEventCreateWithFlags(&d.ev_done, hipEventDisableTiming));
        HIPCHK(c, hipEventCreate(&d.ev_t0)); HIPCHK(c, hipEventCreate(&d.ev_t1));
    }
    DevMerge M{};
    M.dx_chars = dx_chars; M.dx_len = dx_len; M.stride = stride; M.n_tok = n; M.d_ntok = d_ntok;
    M.thr = (uint32_t)std::max(c->prm.kmer_clust_size, 2); M.kmax = stride - 10;
    M.min_len = (uint32_t)c->prm.lowDRsize; M.akey_shift = M.min_len >= 23u ? 3u : 2u;
    { const char *ab = getenv("CRASS_DM_ABLATE"); M.ablate = ab ? (uint32_t)strtoul(ab, nullptr, 0) : 0u; }      // (profiling aid, read per merge)
    M.kset_log = 10; while ((1ull << M.kset_log) < 32ull * n) M.kset_log++;
    M.tab_log_alloc = 16; while (M.tab_log_alloc < 24 && (1ull << M.tab_log_alloc) < 48ull * n) M.tab_log_alloc++;
    HIPCHK(c, d.packed.ensure((size_t)n * 4)); HIPCHK(c, d.codes.ensure((size_t)n * M.kmax)); HIPCHK(c, d.owner.ensure((1u << 22) + kDmBadSlots));
    HIPCHK(c, d.tmask.ensure((size_t)n * 2)); HIPCHK(c, d.bk_key.ensure(kDmBadSlots));
    HIPCHK(c, d.root_of.ensure(n)); HIPCHK(c, d.blank.ensure(n)); HIPCHK(c, d.pat_token.ensure((size_t)n * 2));
    HIPCHK(c, d.kset_key.ensure((size_t)1 << M.kset_log)); HIPCHK(c, d.kset_u32.ensure((size_t)3 << M.kset_log));
    HIPCHK(c, d.ent_slot.ensure((size_t)n * 16)); HIPCHK(c, d.ents.ensure((size_t)n * 64));
    M.rset_log = 10; while ((1ull << M.rset_log) < 4ull * n) M.rset_log++;
    HIPCHK(c, d.rset_key.ensure((size_t)1 << M.rset_log)); HIPCHK(c, d.rset_u32.ensure((size_t)3 << M.rset_log));
    HIPCHK(c, d.rd_slot.ensure(n)); HIPCHK(c, d.rents.ensure((size_t)n * 4));
    HIPCHK(c, d.anchor_tab.ensure((size_t)1 << M.tab_log_alloc)); HIPCHK(c, d.st.ensure(1)); HIPCHK(c, d.hot.ensure(kDmHotWords)); HIPCHK(c, d.anchor_fp.ensure(1u << 15));
    HIPCHK(c, d.h_st.ensure(1)); HIPCHK(c, d.h_root.ensure(n)); HIPCHK(c, d.h_blank.ensure(n));
    M.packed = d.packed.p; M.codes = d.codes.p; M.owner = d.owner.p; M.bk_key = d.bk_key.p; M.root_of = d.root_of.p;
    M.tmask = d.tmask.p; M.pat_mask = d.tmask.p; M.blank = d.blank.p; M.pat_token = d.pat_token.p;
    M.kset_key = d.kset_key.p; M.kset_cnt = d.kset_u32.p; M.kset_base = d.kset_u32.p + ((size_t)1 << M.kset_log);
    M.kset_fill = d.kset_u32.p + ((size_t)2 << M.kset_log); M.ent_slot = d.ent_slot.p; M.ents = d.ents.p; M.ent_cap = 16u * n;
    M.rset_key = d.rset_key.p; M.rset_cnt = d.rset_u32.p; M.rset_base = d.rset_u32.p + ((size_t)1 << M.rset_log);
    M.rset_fill = d.rset_u32.p + ((size_t)2 << M.rset_log); M.rd_slot = d.rd_slot.p; M.rents = d.rents.p;
    if (!c->n_cu) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || v <= 0) v = 64;
        c->n_cu = (uint32_t)v;
    }
    M.n_cu = c->n_cu;
    M.h_st = d.h_st.p; M.h_root = d.h_root.p; M.h_blank = d.h_blank.p;
    // the host view of this merge is assembled on the device too (k_dmx_*), unless this context only reports its own
    // candidates' tokens (ranks > 0 of a group)
    // — and only where the host's build would be on the critical path: a rank of a multi-rank job (its shard's pass 2 is shorter
    // than the 0.6 ms the host needs for 42 k tokens).  A single context hides the host's build behind its own pass 2 (1.5 ms of
    // device work at 100 M reads), and there the export's kernels, running beside pass 2's probe, cost the step 0.08-0.1 ms
    // (4.65 vs 4.73-4.77 ms, same box, profiles/NOTES_r04.md): CRASS_DEVICE_VIEW=1 forces the export there (tests, A/B).
    M.x_on = (!c->env.no_device_view && !c->host_view_light && n <= (1u << 20) && (c->xchg.active || c->env.force_device_view)) ? 1u : 0u;
    if (M.x_on) {
        if (!d.view_stream) {
            // lowest priority: the export fills whatever the merge's own kernels and pass 2's probe leave idle (at equal priority
            // its 1 024-thread blocks kept the probe kernel's blocks off the CUs: 144 -> 216 us for an eighth of the 100 M reads)
            int prio_lo = 0, prio_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            static const bool view_same_prio = getenv("CRASS_VIEW_SAME_PRIORITY") != nullptr;      // A/B switch
            HIPCHK(c, hipStreamCreateWithPriority(&d.view_stream, hipStreamNonBlocking, view_same_prio ? 0 : prio_lo));
            HIPCHK(c, hipEventCreateWithFlags(&d.ev_fork, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&d.ev_view, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&d.ev_apply, hipEventDisableTiming));
            d.dma_view = sdma_create();                 // (nullptr: the runtime's copy is used)
            d.dma_view2 = sdma_create();
            d.dma_view3 = sdma_create();
        }
        const uint64_t cap = view_layout(n, n, 2ull * n, (uint64_t)n * stride, 2ull * n * stride).total;
        HIPCHK(c, d.x_u32.ensure((size_t)n * 8)); HIPCHK(c, d.x_members.ensure(n)); HIPCHK(c, d.x_tile.ensure(kDmxTiles * kDmxVals + 4));
        HIPCHK(c, d.x_blob.ensure(cap + 64)); HIPCHK(c, d.x_tot.ensure(1)); HIPCHK(c, d.x_htot.ensure(2));      // (x_htot[1]: the totals as k_dmx_apply published them)
        if (!d.h_view.p) HIPCHK(c, d.h_view.ensure(std::max<uint64_t>(cap / 4, 1u << 16)));      // (grown by the build if a merge needs more)
        uint32_t *x = d.x_u32.p;
        M.x_size = x; M.x_kept = x + n; M.x_kchars = x + 2ull * n; M.x_fill = x + 3ull * n;
        M.x_gid = x + 4ull * n; M.x_goff = x + 5ull * n; M.x_pat0 = x + 6ull * n; M.x_pch0 = x + 7ull * n;
        M.x_members = d.x_members.p; M.x_tile = d.x_tile.p; M.x_blob = d.x_blob.p; M.x_tot = d.x_tot.p; M.x_htot = d.x_htot.p;
        M.x_group_cap = c->env.view_group_cap;
        M.x_sort_max = c->env.view_sort_max;
    }
    M.inject_fail = c->env.dm_inject_fail ? 1u : 0u;
    M.group_cap = c->env.dm_group_cap;
    M.anchor_tab = d.anchor_tab.p; M.anchor_fp = d.anchor_fp.p; M.m1 = 0x9E3779u; M.m2 = 0x85EBCBu; M.st = d.st.p; M.hot = d.hot.p;
    d.M = M;
    return CRASS_OK;
}

// dx_*: distinct strings in token order on the device; hx_*: the same list in pinned host memory
// the kernels only (n_tok: the token count, or — with d_ntok — the bound the launch is sized for while the count is
// still on the device); the context's state is untouched until device_merge_commit.  prepared: device_merge_prepare
// ran for exactly these arguments and an earlier kernel of the step cleared the tables (dm_init_slice)
static int device_merge_enqueue(crass_hip_ctx *c, const char *dx_chars, const uint16_t *dx_len, uint64_t n_tok, const uint32_t *d_ntok,
                                bool prepared)
{
    crass_hip_ctx::DM &d = c->dm;
    if (!prepared) { const int ps = device_merge_prepare(c, dx_chars, dx_len, n_tok, d_ntok); if (ps) return ps; }
    const double tl0 = now_ms();
    if (d.view_launched) HIPCHK(c, hipStreamWaitEvent(c->stream, d.ev_view, 0));     // (an abandoned merge's export still owns the x_* words)
    if (c->timing_level >= 2) HIPCHK(c, hipEventRecord(d.ev_t0, c->stream));
    d.M.flag_pre = c->poll_on ? c->h_flags.p + 1 : nullptr; d.M.flag_pre_val = c->want_pre = ++c->flag_seq;
    d.M.flag_post = c->poll_on ? c->h_flags.p + 2 : nullptr; d.M.flag_post_val = d.want_post = ++c->flag_seq;
    d.M.flag_blank = c->poll_on ? c->h_flags.p + 3 : nullptr; d.M.flag_blank_val = d.want_blank = ++c->flag_seq;
    static const bool view_inline = getenv("CRASS_VIEW_INLINE") != nullptr;      // A/B switch: the export on the merge's own stream
    HIPCHK(c, launch_device_merge(d.M, c->stream, prepared, d.M.x_on ? (view_inline ? c->stream : d.view_stream) : nullptr, d.ev_fork, d.ev_view,
                                  view_inline ? nullptr : d.ev_apply));
    d.apply_recorded = d.M.x_on && !view_inline;
    d.view_launched = d.M.x_on != 0;
    if (c->timing_level >= 2) HIPCHK(c, hipEventRecord(d.ev_t1, c->stream));
    if (c->env.merge_profile)
        fprintf(stderr, "[crass_dm] host: pass-1 sync -> merge launch start %.1f us, launching the merge kernels %.1f us\n",
                1e3 * (tl0 - c->t_p1_sync), 1e3 * (now_ms() - tl0));
    // the per-token results the host view is rebuilt from (a few 10 KB): the helper thread polls the stage flag that pass 2's
    // probe stores, or waits for this event
    if (!c->poll_on) HIPCHK(c, hipEventRecord(d.ev_done, c->stream));
    return CRASS_OK;
}

// the merge just queued becomes the context's merge: hx_*: the distinct list in pinned host memory
static int device_merge_commit(crass_hip_ctx *c, uint64_t n_tok, const char *hx_chars, const uint16_t *hx_len)
{
    crass_hip_ctx::DM &d = c->dm;
    d.hx_chars = hx_chars; d.hx_len = hx_len; d.n_tok = n_tok;
    d.active = true; d.host_built = false; d.view_ready = false; d.n_cand = c->n_cand();
    // the host view (tokens, groups, pattern list) is rebuilt by the helper thread as soon as the kernels are through
    // every field of the caller's side is set BEFORE the job is handed over: from submit() on, the helper thread owns
    // c->merge and dm.br until ensure_host_merge / quiesce_worker has waited for it
    c->have_merge = true; c->have_pass2 = false;
    c->have_patterns = true; c->have_anchors = false; c->have_pat_token = false;
    c->n_installed_patterns = 2;                                  // >= 1 survivor exists; the exact count arrives with h_st
    d.br = crass_hip_ctx::DM::BuildResult();
    d.build_status = CRASS_OK;
    d.build_pending = true;
    c->worker.submit([c] {
        int st;
        try { st = build_host_merge(c); } catch (const std::bad_alloc &) { st = CRASS_ERR_OOM; } catch (...) { st = CRASS_ERR_STATE; }
        c->dm.build_status = st;
    });
    return CRASS_OK;
}

// dx_*: distinct strings in token order on the device; hx_*: the same list in pinned host memory
static int device_merge(crass_hip_ctx *c, const char *dx_chars, const uint16_t *dx_len, uint64_t n_tok, const char *hx_chars,
                        const uint16_t *hx_len)
{
    const int s = device_merge_enqueue(c, dx_chars, dx_len, n_tok, nullptr);
    if (s) return s;
    return device_merge_commit(c, n_tok, hx_chars, hx_len);
}

// the distinct list on the host, complete: the engine copies have landed and the strings' hashes are there (the host-merge
// forms want them; with the list brought over by DMA they are computed here, on the rare paths that need them)
static int ensure_host_distinct(crass_hip_ctx *c, bool want_hash)
{
    if (const int ws = c->wait_dx()) return ws;
    if (want_hash && !c->dx_hash_valid) {
        for (uint64_t j = 0; j < c->n_dx; j++) c->h_dx_hash.p[j] = TokenTable::hash(c->h_dx_chars.p + j * (size_t)c->dr_stride, c->h_dx_len.p[j]);
        c->dx_hash_valid = true;
    }
    return CRASS_OK;
}

// host merge after all (the device path reported a condition it does not handle)
static int merge_global_host(crass_hip_ctx *c, const char *dr_chars, const uint16_t *dr_len, uint32_t dr_stride, uint64_t n_global,
                             uint64_t my_offset, double t0);

static int host_merge_fallback(crass_hip_ctx *c)
{
    quiesce_worker(c);
    c->n_merge_fallbacks++;                             // (reported by crass_hip_get_counters: a silent latency cliff otherwise)
    c->last_fallback_bits = c->dm.h_st.p ? c->dm.h_st.p->fail : 0u;
    c->dm.active = false;
    c->cnt.used_device_merge = 0;
    c->dm_prev_local = false;                           // (no merge is queued ahead of time after a device-side failure)
    const double t0 = now_ms();
    if (c->dm.global) {
        // the concatenated list is still on the device (engine-owned copy)
        crass_hip_ctx::DM &d = c->dm;
        std::vector<char> gc(d.n_global * (size_t)c->dr_stride);
        std::vector<uint16_t> gl(d.n_global);
        HIPCHK(c, hipMemcpy(gc.data(), d.g_chars.p, gc.size(), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(gl.data(), d.g_len.p, gl.size() * 2, hipMemcpyDeviceToHost));
        return merge_global_host(c, gc.data(), gl.data(), c->dr_stride, d.n_global, d.my_off, t0);
    }
    if (const int ds = ensure_host_distinct(c, true)) return ds;
    if (!merge_from_distinct(c->merge, c->h_dx_chars.p, c->h_dx_len.p, c->h_dx_hash.p, c->dr_stride, c->n_dx, c->h_dmap.p, c->n_cand(),
                             c->prm.kmer_clust_size))
        merge_candidates(c->merge, c->cand_dr(), c->cand_dr_len(), c->dr_stride, c->n_cand(), c->prm.kmer_clust_size);
    return finish_merge(c, t0);
}

// c->merge for a merge that ran on the device: 0 ok, CRASS_ERR_STATE = the device path failed (caller falls back)
static int build_host_merge(crass_hip_ctx *c);
static void apply_build_result(crass_hip_ctx *c);

static int ensure_host_merge(crass_hip_ctx *c)
{
    crass_hip_ctx::DM &d = c->dm;
    if (d.build_pending) {                              // started by the merge call on the helper thread
        if (!c->worker.run_here_if_not_started()) c->worker.wait();
        d.build_pending = false;
        if (c->worker.take_failed()) d.build_status = CRASS_ERR_OOM;        // (the job threw: std::bad_alloc is the one thing it can throw)
        apply_build_result(c);
        if (d.build_status == CRASS_ERR_HIP || d.build_status == CRASS_ERR_OOM) return d.build_status;
        if (d.build_status == CRASS_ERR_STATE) return CRASS_ERR_STATE;
    }
    const int s = build_host_merge(c);
    apply_build_result(c);
    return s;
}

static void apply_build_result(crass_hip_ctx *c)
{
    crass_hip_ctx::DM::BuildResult &b = c->dm.br;
    if (b.hip) { c->last_hip = b.hip; b.hip = 0; }
    if (!b.valid) return;
    b.valid = false;
    c->n_installed_patterns = b.n_patterns;
    c->cnt.n_patterns = b.n_patterns;
    c->cnt.ac_states = 0;
    c->cnt.anchor_keys = b.n_keys;
    c->cnt.ms_merge_device = b.ms_device;
}

// runs on the helper thread OR on the caller's (never both at once): touches c->merge and dm.br only
static int build_host_merge(crass_hip_ctx *c)
{
    crass_hip_ctx::DM &d = c->dm;
    if (!d.active || d.host_built) return CRASS_OK;
    {   // once per thread and device: the helper thread runs this for every step
        static thread_local int tl_device = -1;
        if (tl_device != c->device) { (void)hipSetDevice(c->device); tl_device = c->device; }
    }
    auto wait_done = [&]() -> int {
        // (the probe of pass 2 is normally queued right behind the merge; a caller that asks for the merge's view without running
        // pass 2 gets it after the poll's budget)
        if (c->poll_on && c->poll_flag(2, d.want_post, 1.0)) return CRASS_OK;
        const hipError_t e = c->poll_on ? hipStreamSynchronize(c->stream) : hipEventSynchronize(d.ev_done);
        if (e != hipSuccess) { d.br.hip = (int)e; return e == hipErrorOutOfMemory ? CRASS_ERR_OOM : CRASS_ERR_HIP; }
        return CRASS_OK;
    };
    auto publish = [&]() {
        d.br.n_patterns = d.h_st.p->n_patterns; d.br.n_keys = d.h_st.p->n_keys;
        float ms = 0;
        if (c->timing_level >= 2) (void)hipEventElapsedTime(&ms, d.ev_t0, d.ev_t1);
        d.br.ms_device = ms;
        d.br.valid = true;
    };
    // every exit waits for the view export (a few small kernels on view_stream): whoever re-uses the x_* words or the blob
    // next must find them idle
    struct DrainView {
        crass_hip_ctx::DM &d;
        ~DrainView() { if (d.view_launched && d.ev_view) (void)hipEventSynchronize(d.ev_view); }
    } drain{d};
    // local merge: the token strings and the candidates' tokens only need pass 1's outputs, which the host has
    // already waited for — that half of the host view is built while the merge kernels are still running
    // (the gathered form too: the global distinct list and every gathered row's rank in it were written to pinned memory
    // by the de-duplication kernels, which the host has waited for before it committed the merge)
    const double tb00 = now_ms();
    if (const int ws = c->wait_dx()) { d.br.hip = c->last_hip; return ws; }      // (the candidates' indices and the distinct strings: DMA copies)
    const uint32_t *cmap = c->h_dmap.p;
    if (d.global) {                                     // own candidate -> own distinct string -> its rank in the global list
        d.cand_map.resize(d.n_cand);
        for (uint64_t k = 0; k < d.n_cand; k++) d.cand_map[k] = d.h_gmap.p[d.my_off + c->h_dmap.p[k]];
        cmap = d.cand_map.data();
    }
    if (c->host_view_light) {
        // a rank of a group other than rank 0: tokens, groups and patterns are read from rank 0's view (they are identical on
        // every rank); what is this rank's own are its candidates' tokens
        c->merge.clear();
        c->merge.cand_token.resize(d.n_cand);
        for (uint64_t k = 0; k < d.n_cand; k++) {
            if (cmap[k] >= d.n_tok) { c->merge.clear(); return CRASS_ERR_STATE; }
            c->merge.cand_token[k] = cmap[k] + 2;
        }
        if (const int ws = wait_done()) return ws;
        if (d.h_st.p->fail) return CRASS_ERR_STATE;
        d.host_built = true;
        publish();
        d.br.ms_device = 0;
        return CRASS_OK;
    }
    if (d.M.x_on && d.view_launched) {
        // The view itself comes from the device (k_dmx_*): what is left for the host are its own candidates' tokens (pass 1's
        // outputs, done while the merge kernels run), one DMA copy of the blob, and pointers.
        c->merge.clear();
        c->merge.cand_token.resize(d.n_cand);
        for (uint64_t k = 0; k < d.n_cand; k++) {
            if (cmap[k] >= d.n_tok) { c->merge.clear(); return CRASS_ERR_STATE; }
            c->merge.cand_token[k] = cmap[k] + 2;
        }
        const double tv0 = now_ms();
        // The device's part of the view (k_dmx_*, forked behind k_dm_greedy): tok_off, the token strings, grp_off — complete
        // behind k_dmx_apply, copied first, beside the kernels that order the members — and grp_tokens, complete behind the
        // export's last kernel.  The PATTERN LIST (per group the survivors by length then token, then their reverse
        // complements: WorkHorse.cpp:690-697, remove_redundant's order) is put together here, from that view and blank[],
        // which k_dm_redundant has finished when k_dm_keys' stage flag shows — while the device builds pass 2's index and
        // runs pass 2.  (Until round 6 the export ran behind k_dm_redundant and ranked the patterns too: its last kernels
        // shared the CUs with pass 2's probe and verification, and the view was ready ~20 us after the step's last kernel.)
        auto hipfail = [&](hipError_t e) { d.br.hip = (int)e; return e == hipErrorOutOfMemory ? CRASS_ERR_OOM : CRASS_ERR_HIP; };
        bool usable = d.apply_recorded && d.dma_view && d.dma_view2 && d.dma_view3;
        DevViewTotals E{};
        if (usable) {
            const hipError_t e = hipEventSynchronize(d.ev_apply);
            if (e != hipSuccess) return hipfail(e);
            E = d.x_htot.p[1];
            usable = E.ok && E.n_tok == d.n_tok && E.n_groups >= 1 && E.lay.total <= d.x_blob.n && E.lay.grp_tokens <= E.lay.pat_off && E.lay.pat_off <= E.lay.total;
        }
        if (usable) {
            // room for the whole view whatever survives (every token a pattern, twice): grown before the first copy lands
            const uint64_t room = view_layout(E.n_tok, E.n_groups, 2ull * E.n_tok, E.tok_chars, 2ull * E.tok_chars).total;
            if (d.h_view.n < room) { const hipError_t e2 = d.h_view.ensure(room + room / 4); if (e2 != hipSuccess) return hipfail(e2); }
            bool ok = sdma_start(d.dma_view2, d.x_blob.p, d.h_view.p, E.lay.grp_tokens);
            struct Guard { SdmaCopy *s; bool on; ~Guard() { if (on) (void)sdma_wait(s); } } g2{d.dma_view2, ok}, g1{d.dma_view, false}, g3{d.dma_view3, false};
            // blank[]: final when k_dm_redundant is through (the flag is k_dm_keys'); a poll that runs out waits for the merge
            if (!(c->poll_on && c->poll_flag(3, d.want_blank, 2.0))) { if (const int ws = wait_done()) return ws; }
            g3.on = ok && sdma_start(d.dma_view3, d.blank.p, d.h_blank.p, d.n_tok);
            ok = ok && g3.on;
            { const hipError_t e = hipEventSynchronize(d.ev_view); if (e != hipSuccess) return hipfail(e); }
            const DevViewTotals T0 = *d.x_htot.p;
            usable = T0.ok && T0.n_tok == d.n_tok && T0.n_groups == E.n_groups && T0.lay.grp_tokens == E.lay.grp_tokens && T0.lay.pat_off == E.lay.pat_off;
            if (usable) {
                g1.on = ok && sdma_start(d.dma_view, d.x_blob.p + E.lay.grp_tokens, d.h_view.p + E.lay.grp_tokens, E.lay.pat_off - E.lay.grp_tokens);
                ok = ok && g1.on;
                if (g1.on) { g1.on = false; ok = sdma_wait(d.dma_view) == 0 && ok; }
                if (g2.on) { g2.on = false; ok = sdma_wait(d.dma_view2) == 0 && ok; }
                if (g3.on) { g3.on = false; ok = sdma_wait(d.dma_view3) == 0 && ok; }
                if (!ok) {                              // (no engine: the runtime's copies, behind the export on its stream)
                    hipError_t e = hipMemcpyAsync(d.h_view.p, d.x_blob.p, E.lay.pat_off, hipMemcpyDeviceToHost, d.view_stream);
                    if (e == hipSuccess) e = hipMemcpyAsync(d.h_blank.p, d.blank.p, d.n_tok, hipMemcpyDeviceToHost, d.view_stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(d.view_stream);
                    if (e != hipSuccess) return hipfail(e);
                }
                const double tv1 = now_ms();
                // ---- the pattern list ----
                uint8_t *hb = d.h_view.p;
                const uint64_t *tok_off = reinterpret_cast<const uint64_t *>(hb + E.lay.tok_off), *grp_off = reinterpret_cast<const uint64_t *>(hb + E.lay.grp_off);
                const char *tok_chars = reinterpret_cast<const char *>(hb + E.lay.tok_chars);
                const uint32_t *grp_tokens = reinterpret_cast<const uint32_t *>(hb + E.lay.grp_tokens);
                const uint8_t *blank = d.h_blank.p;
                // (over the host pool: a group's part of the list depends on the groups in front of it only through two offsets —
                // one thread took 1.1 ms for 13 k patterns of 42 k tokens; chunks of groups of ~equal member counts)
                const uint32_t NG = E.n_groups;
                const size_t n_chunks = std::min<size_t>(64, NG);
                std::vector<uint32_t> chunk_g0(n_chunks + 1, NG);
                {
                    size_t cidx = 0;
                    for (uint32_t g = 0; g < NG && cidx < n_chunks; g++)
                        if (grp_off[g] * n_chunks >= (uint64_t)cidx * E.n_tok) chunk_g0[cidx++] = g;
                    for (; cidx < n_chunks; cidx++) chunk_g0[cidx] = NG;
                    chunk_g0[n_chunks] = NG;
                }
                std::vector<uint64_t> ck(n_chunks + 1, 0), cc(n_chunks + 1, 0);     // kept tokens / their characters per chunk
                std::atomic<int> bad{0};
                bool sane = grp_off[NG] == E.n_tok;
                if (sane) host_parallel_for(n_chunks, 64, [&](size_t ci) {
                    uint64_t k = 0, chs = 0;
                    for (uint64_t i = grp_off[chunk_g0[ci]], e = chunk_g0[ci + 1] < NG ? grp_off[chunk_g0[ci + 1]] : E.n_tok; i < e; i++) {
                        const uint32_t t = grp_tokens[i] - 2u;
                        if (t >= E.n_tok) { bad = 1; return; }
                        const uint64_t len = tok_off[t + 1] - tok_off[t];
                        if (len > 64) { bad = 1; return; }
                        if (!blank[t]) { k++; chs += len; }
                    }
                    ck[ci + 1] = k; cc[ci + 1] = chs;
                });
                sane = sane && !bad.load();
                for (size_t ci = 0; ci < n_chunks; ci++) { ck[ci + 1] += ck[ci]; cc[ci + 1] += cc[ci]; }
                const uint64_t n_kept = ck[n_chunks], kept_chars = cc[n_chunks];
                if (sane) {
                    DevViewTotals T = T0;
                    T.n_kept = (uint32_t)n_kept; T.kept_chars = (uint32_t)kept_chars;
                    T.lay = view_layout(E.n_tok, E.n_groups, 2ull * n_kept, E.tok_chars, 2ull * kept_chars);
                    uint64_t *pat_off = reinterpret_cast<uint64_t *>(hb + T.lay.pat_off);
                    char *pat_chars = reinterpret_cast<char *>(hb + T.lay.pat_chars);
                    uint32_t *pat_group = reinterpret_cast<uint32_t *>(hb + T.lay.pat_group);
                    host_parallel_for(n_chunks, 64, [&](size_t ci) {
                        static thread_local std::vector<uint32_t> order;        // a group's survivors by length, then token
                        static const struct Comp { char t[256]; Comp() { for (int i = 0; i < 256; i++) t[i] = (char)i; t['A'] = 'T'; t['C'] = 'G'; t['G'] = 'C'; t['T'] = 'A'; } } comp;   // SeqUtils.cpp:50-59 over the device merge's alphabet
                        uint64_t p = 2ull * ck[ci], ch = 2ull * cc[ci];
                        for (uint32_t g = chunk_g0[ci]; g < chunk_g0[ci + 1]; g++) {
                            const uint64_t m0 = grp_off[g], m1 = grp_off[g + 1];
                            uint32_t cnt[66] = {0};
                            uint32_t kg = 0;
                            for (uint64_t i = m0; i < m1; i++) {
                                const uint32_t t = grp_tokens[i] - 2u;
                                if (blank[t]) continue;
                                cnt[tok_off[t + 1] - tok_off[t] + 1]++; kg++;
                            }
                            for (int l = 1; l < 66; l++) cnt[l] += cnt[l - 1];
                            order.resize(kg);
                            for (uint64_t i = m0; i < m1; i++) {             // (token order inside a length: the list is sorted by token)
                                const uint32_t t = grp_tokens[i] - 2u;
                                if (!blank[t]) order[cnt[tok_off[t + 1] - tok_off[t]]++] = t;
                            }
                            uint64_t chr = ch;
                            for (uint32_t j = 0; j < kg; j++) chr += tok_off[order[j] + 1] - tok_off[order[j]];
                            for (uint32_t j = 0; j < kg; j++) {
                                const uint32_t t = order[j];
                                const uint64_t len = tok_off[t + 1] - tok_off[t];
                                const char *src = tok_chars + tok_off[t];
                                pat_off[p + j] = ch; pat_group[p + j] = g + 1u;
                                pat_off[p + kg + j] = chr; pat_group[p + kg + j] = g + 1u;
                                memcpy(pat_chars + ch, src, len);
                                char *dst = pat_chars + chr;
                                for (uint64_t q = 0; q < len; q++) dst[q] = comp.t[(unsigned char)src[len - 1 - q]];      // reverseComplement
                                ch += len; chr += len;
                            }
                            p += 2ull * kg; ch = chr;
                        }
                    });
                    pat_off[2ull * n_kept] = 2ull * kept_chars;
                    const double tv2 = now_ms();
                    if (const int ws = wait_done()) return ws;
                    if (d.h_st.p->fail) return CRASS_ERR_STATE;
                    if (sane && 2u * T.n_kept == d.h_st.p->n_patterns) {
                        d.view_tot = T;
                        d.view_ready = true;
                        d.host_built = true;
                        if (c->env.merge_profile)
                            fprintf(stderr, "[crass_dm] helper: candidates' tokens %.1f us, the device's part of the view in host memory %.1f us later, pattern list %.1f us (done %.1f us after the pass-1 sync; the merge's state %.1f us after that)\n",
                                    1e3 * (tv0 - tb00), 1e3 * (tv1 - tv0), 1e3 * (tv2 - tv1), 1e3 * (tv2 - c->t_p1_sync), 1e3 * (now_ms() - tv2));
                        publish();
                        return CRASS_OK;
                    }
                }
            }
        }
        if (const int ws = wait_done()) return ws;
        if (d.h_st.p->fail) return CRASS_ERR_STATE;
        // (a group beyond x_group_cap, or totals that do not add up: the host builds the view from the per-token results)
        c->n_view_fallbacks++;
    }
    if (d.global && !d.hx_on_host) {                    // the gathered list stayed on the device: fetch it now
        hipError_t e = hipMemcpyAsync(d.h_gx_chars.p, d.gx_chars.p, d.n_tok * (size_t)c->dr_stride, hipMemcpyDeviceToHost, d.view_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d.h_gx_len.p, d.gx_len.p, d.n_tok * 2, hipMemcpyDeviceToHost, d.view_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(d.view_stream);
        if (e != hipSuccess) { d.br.hip = (int)e; return CRASS_ERR_HIP; }
        d.hx_on_host = true;
    }
    if (!merge_from_device_begin(c->merge, d.hx_chars, d.hx_len, c->dr_stride, d.n_tok, cmap, d.n_cand)) return CRASS_ERR_STATE;
    host_pool_warm();                                   // the second half fans out over the pool: wake it while the device is busy
    const double tb0 = now_ms();
    if (const int ws = wait_done()) return ws;
    const double tb1 = now_ms();
    if (d.h_st.p->fail) return CRASS_ERR_STATE;
    if (d.M.x_on) {                                     // (k_dm_fill_finish left the per-token results on the device)
        hipError_t e = hipMemcpyAsync(d.h_root.p, d.root_of.p, d.n_tok * 4, hipMemcpyDeviceToHost, d.view_stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d.h_blank.p, d.blank.p, d.n_tok, hipMemcpyDeviceToHost, d.view_stream);
        if (e == hipSuccess) e = hipStreamSynchronize(d.view_stream);
        if (e != hipSuccess) { d.br.hip = (int)e; return CRASS_ERR_HIP; }
    }
    if (!merge_from_device_finish_roots(c->merge, d.h_root.p, d.h_blank.p, d.gid_tmp) ||
        c->merge.patterns.size() != d.h_st.p->n_patterns)
        return CRASS_ERR_STATE;
    d.host_built = true;
    if (c->env.merge_profile)
        fprintf(stderr, "[crass_dm] helper: first half %.1f us, waited %.1f us for the merge kernels, second half %.1f us (done %.1f us after the pass-1 sync)\n",
                1e3 * (tb0 - tb00), 1e3 * (tb1 - tb0), 1e3 * (now_ms() - tb1), 1e3 * (now_ms() - c->t_p1_sync));
    publish();
    return CRASS_OK;
}

int crass_hip_merge(crass_hip_ctx *c, const char *dr_chars, const uint16_t *dr_len, uint32_t dr_stride, uint64_t n)
{
    if (!c) return CRASS_ERR_INVALID_ARG;
    if (c->p1d.active) { const int ds = settle_p1(c); if (ds) return ds; }
    const double t0 = now_ms();
    quiesce_worker(c);
    c->dm.active = false;
    c->cnt.used_device_merge = 0; c->cnt.ms_merge_device = 0;
    const bool adopt = !dr_chars && c->premerge == 2;
    c->premerge = 0;
    if (!dr_chars && device_merge_applies(c)) {
        (void)hipSetDevice(c->device);
        c->dm.global = false;
        int s = !c->n_dx ? CRASS_ERR_STATE
                : adopt  ? device_merge_commit(c, c->n_dx, c->h_dx_chars.p, c->h_dx_len.p)        // queued by the seed scan
                         : device_merge(c, c->dd_dx_chars.p, c->dd_dx_len.p, c->n_dx, c->h_dx_chars.p, c->h_dx_len.p);
        if (s == CRASS_ERR_STATE) goto host_path;
        if (s == CRASS_OK) {
            // the merge kernels are already running: the pass-1 hand-off pack (PCIe bound) goes beside them — small,
            // latency-bound kernels — rather than beside the pass-2 filter, which it would slow down
            c->cnt.used_device_merge = 1;
            c->cnt.ms_merge_host = (float)(now_ms() - t0);
            c->dm_prev_local = true;
            c->dx_cap_hint = (uint32_t)(((uint64_t)c->n_dx * 3 / 2 + 4095) & ~4095ull);
            return CRASS_OK;
        }
        return s;
    }
host_path:
    c->dm_prev_local = false;
    if (!dr_chars && c->have_pass1 && c->dev_tokens()) { if (const int ds = ensure_host_distinct(c, true)) return ds; }
    if (!dr_chars && c->have_pass1 && c->dev_tokens() &&
        merge_from_distinct(c->merge, c->h_dx_chars.p, c->h_dx_len.p, c->h_dx_hash.p, c->dr_stride, c->n_dx, c->h_dmap.p, c->n_cand(),
                            c->prm.kmer_clust_size))
        return finish_merge(c, t0);
    if (!dr_chars) {
        if (!c->have_pass1) return CRASS_ERR_STATE;
        dr_chars = c->cand_dr(); dr_len = c->cand_dr_len(); dr_stride = c->dr_stride; n = c->n_cand();
    } else if (!dr_len || !dr_stride) return CRASS_ERR_INVALID_ARG;
    const bool own = (dr_chars == c->cand_dr());
    merge_candidates(c->merge, dr_chars, dr_len, dr_stride, n, c->prm.kmer_clust_size,
                     (own && c->have_rep && c->dense.active) ? c->h_rep.p : nullptr,
                     (own && c->have_rep && c->dense.active) ? c->h_hash.p : nullptr);
    return finish_merge(c, t0);
}

// shared tail of the merge entry points: automaton/anchors/pattern tokens for pass 2
static int finish_merge(crass_hip_ctx *c, double t0)
{
    c->have_merge = true;
    c->have_pass2 = false;
    int s = install_patterns(c, c->merge.patterns);
    if (s == CRASS_OK && c->have_patterns) {
        // token of every pattern's low-lexi form, resolved once per pattern (merge.cpp) instead of once per
        // recruited read
        const std::vector<uint32_t> &pt = c->merge.pat_token;
        HIPCHK(c, c->a_pat_token.ensure(pt.size()));
        HIPCHK(c, hipMemcpyAsync(c->a_pat_token.p, pt.data(), pt.size() * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->have_pat_token = true;
    }
    if (const int bs = c->wait_bulk()) return bs;            // the per-candidate records are in host memory when the merge returns
    c->cnt.ms_merge_host = (float)(now_ms() - t0);
    return s;
}

static void ensure_distinct(crass_hip_ctx *c)
{
    if (c->have_distinct) return;
    if (c->dev_tokens()) { c->have_distinct = true; return; }   // straight from the device
    const uint64_t n = c->n_cand();
    const char *dr = c->cand_dr();
    const uint16_t *len = c->cand_dr_len();
    const uint32_t stride = c->dr_stride;
    c->dx_chars.clear(); c->dx_len.clear(); c->dx_map.assign(n, 0);
    if (c->have_rep && c->dense.active) {
        // the device already knows every candidate's first occurrence
        for (uint64_t k = 0; k < n; k++) {
            const uint32_t f = c->h_rep.p[k];
            if (f == k) {
                c->dx_map[k] = (uint32_t)c->dx_len.size();
                c->dx_len.push_back(len[k]);
                c->dx_chars.insert(c->dx_chars.end(), dr + k * (uint64_t)stride, dr + (k + 1) * (uint64_t)stride);
            } else c->dx_map[k] = c->dx_map[f];
        }
    } else {
        TokenTable t;
        for (uint64_t k = 0; k < n; k++) {
            const char *p = dr + k * (uint64_t)stride;
            uint32_t tok = t.get(p, len[k]);
            if (!tok) {
                tok = t.add(p, len[k]);
                c->dx_len.push_back(len[k]);
                c->dx_chars.insert(c->dx_chars.end(), p, p + stride);
            }
            c->dx_map[k] = tok - 2;
        }
    }
    c->have_distinct = true;
}

int crass_hip_get_distinct(crass_hip_ctx *c, crass_distinct *o)
{
    if (!c || !o) return CRASS_ERR_INVALID_ARG;
    if (c->p1d.active) { const int ds = settle_p1(c); if (ds) return ds; }
    if (!c->have_pass1) return CRASS_ERR_STATE;
    ensure_distinct(c);
    o->dr_stride = c->dr_stride;
    if (c->dev_tokens()) {
        if (const int ds = c->wait_dx()) return ds;
        o->n_distinct = c->n_dx; o->dr_len = c->h_dx_len.p; o->dr_chars = c->h_dx_chars.p;
        o->n_candidates = c->n_cand(); o->cand_distinct = c->h_dmap.p;
    } else {
        o->n_distinct = c->dx_len.size(); o->dr_len = c->dx_len.data(); o->dr_chars = c->dx_chars.data();
        o->n_candidates = c->dx_map.size(); o->cand_distinct = c->dx_map.data();
    }
    return CRASS_OK;
}

// host merge over the concatenation of every rank's distinct list
static int merge_global_host(crass_hip_ctx *c, const char *dr_chars, const uint16_t *dr_len, uint32_t dr_stride, uint64_t n_global,
                             uint64_t my_offset, double t0)
{
    c->premerge = 0; c->dm_prev_local = false;          // (multi-rank: nothing is queued ahead of the exchange)
    ensure_distinct(c);
    const bool dev = c->dev_tokens();
    const uint64_t my_nd = dev ? c->n_dx : c->dx_len.size();
    if (dev) { if (const int ds = c->wait_dx()) return ds; }
    const uint32_t *my_map = dev ? c->h_dmap.p : c->dx_map.data();
    const size_t my_n = dev ? (size_t)c->n_cand() : c->dx_map.size();
    if (my_offset + my_nd > n_global) {
        if (getenv("CRASS_GROUP_DEBUG")) fprintf(stderr, "[crass_xchg] merge_global_host: offset %llu + own %llu > global %llu (dev %d)\n",
                                                 (unsigned long long)my_offset, (unsigned long long)my_nd, (unsigned long long)n_global, (int)dev);
        return CRASS_ERR_INVALID_ARG;
    }
    merge_candidates(c->merge, dr_chars, dr_len, dr_stride, n_global, c->prm.kmer_clust_size);
    // tokens of this context's own candidates through their distinct index
    std::vector<uint32_t> own(my_n);
    for (size_t k = 0; k < own.size(); k++) own[k] = c->merge.cand_token[my_offset + my_map[k]];
    c->merge.cand_token.swap(own);
    return finish_merge(c, t0);
}

// The same on the device: de-duplicate the concatenation (first occurrence in rank order = global token order,
// the kernels of the single-GPU pass-1 tail), then the device merge over the global distinct list.  d_chars /
// d_len are device pointers owned by the engine (dm.g_chars / dm.g_len).  CRASS_ERR_STATE: not applicable,
// the caller merges on the host.
static int merge_global_device(crass_hip_ctx *c, uint64_t n_global, uint64_t my_offset)
{
    crass_hip_ctx::DM &d = c->dm;
    const uint32_t stride = c->dr_stride;
    c->premerge = 0; c->dm_prev_local = false;          // (multi-rank: nothing is queued ahead of the exchange)
    if (!device_merge_applies(c) || n_global == 0 || n_global > (1u << 22) || my_offset + c->n_dx > n_global) return CRASS_ERR_STATE;
    const uint32_t n = (uint32_t)n_global;
    uint32_t tsize = 1024;
    while (tsize < n * 2) tsize <<= 1;
    const uint64_t n_words = (n_global + 63) / 64;
    HIPCHK(c, d.g_keys.ensure(tsize)); HIPCHK(c, d.g_first.ensure(tsize)); HIPCHK(c, d.g_slot.ensure(n)); HIPCHK(c, d.g_rep.ensure(n));
    HIPCHK(c, d.g_hash.ensure(n)); HIPCHK(c, d.g_mask.ensure(n_words + 1)); HIPCHK(c, d.g_prefix.ensure(n_words + 1));
    HIPCHK(c, d.g_bsum.ensure((n_words + 255) / 256 + 2)); HIPCHK(c, d.g_idx.ensure(n));
    HIPCHK(c, d.gx_chars.ensure((size_t)n * stride + 16)); HIPCHK(c, d.gx_len.ensure(n));
    HIPCHK(c, d.h_gmap.ensure(n)); HIPCHK(c, d.h_gx_chars.ensure((size_t)n * stride + 16)); HIPCHK(c, d.h_gx_len.ensure(n)); HIPCHK(c, d.h_gx_hash.ensure(n));
    // d_count[7] = n_global, [4] = distinct count, [5] = hash-collision flag
    c->h_count.p[7] = n;
    HIPCHK(c, hipMemcpyAsync(c->d_count.p + 7, c->h_count.p + 7, 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_count.p + 4, 0, 8, c->stream));
    HIPCHK(c, launch_dr_dedupe(d.g_chars.p, d.g_len.p, stride, c->d_count.p + 7, n, d.g_keys.p, d.g_first.p, tsize, d.g_hash.p, d.g_slot.p,
                               d.g_rep.p, c->stream));
    HIPCHK(c, launch_dx_tokens(d.g_chars.p, d.g_len.p, d.g_hash.p, stride, c->d_count.p + 7, n, d.g_rep.p, d.g_slot.p, d.g_first.p, d.g_mask.p, d.g_prefix.p, d.g_bsum.p,
                               d.g_idx.p, c->d_count.p + 4, c->d_count.p + 5, d.h_gmap.p, d.h_gx_chars.p, d.h_gx_len.p, d.h_gx_hash.p,
                               d.gx_chars.p, d.gx_len.p, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_count.p + 4, c->d_count.p + 4, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->h_count.p[5] != 0 || c->h_count.p[4] == 0 || c->h_count.p[4] > (1u << 20)) return CRASS_ERR_STATE;
    d.global = true; d.my_off = my_offset; d.n_global = n_global; d.hx_on_host = true;
    return device_merge(c, d.gx_chars.p, d.gx_len.p, c->h_count.p[4], d.h_gx_chars.p, d.h_gx_len.p);
}

int crass_hip_merge_distinct(crass_hip_ctx *c, const char *dr_chars, const uint16_t *dr_len, uint32_t dr_stride,
                             uint64_t n_global, uint64_t my_offset)
{
    if (!c || (n_global && (!dr_chars || !dr_len || !dr_stride))) return CRASS_ERR_INVALID_ARG;
    if (c->p1d.active) { const int ds = settle_p1(c); if (ds) return ds; }
    if (!c->have_pass1) return CRASS_ERR_STATE;
    const double t0 = now_ms();
    quiesce_worker(c);
    c->dm.active = false;
    c->cnt.used_device_merge = 0; c->cnt.ms_merge_device = 0;
    (void)hipSetDevice(c->device);
    if (dr_stride == c->dr_stride && n_global && n_global <= (1u << 22) && device_merge_applies(c)) {
        crass_hip_ctx::DM &d = c->dm;
        HIPCHK(c, d.g_chars.ensure(n_global * (size_t)dr_stride + 16)); HIPCHK(c, d.g_len.ensure(n_global));
        HIPCHK(c, hipMemcpyAsync(d.g_chars.p, dr_chars, n_global * (size_t)dr_stride, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d.g_len.p, dr_len, n_global * 2, hipMemcpyHostToDevice, c->stream));
        const int s = merge_global_device(c, n_global, my_offset);
        if (s == CRASS_OK) { c->cnt.used_device_merge = 1; c->cnt.ms_merge_host = (float)(now_ms() - t0); return CRASS_OK; }
        if (s != CRASS_ERR_STATE) return s;
        HIPCHK(c, hipStreamSynchronize(c->stream));                 // (the uploads read the caller's buffers)
        c->dm.global = false;
    }
    return merge_global_host(c, dr_chars, dr_len, dr_stride, n_global, my_offset, t0);
}

// every buffer crass_hip_merge_gathered touches for up to n_max gathered rows
static int ensure_gathered_buffers(crass_hip_ctx *c, uint64_t n_max)
{
    crass_hip_ctx::DM &d = c->dm;
    const uint32_t stride = c->dr_stride, n = (uint32_t)n_max;
    uint32_t tsize = 1024;
    while (tsize < n * 2) tsize <<= 1;
    const uint64_t n_words = (n_max + 63) / 64;
    HIPCHK(c, d.g_chars.ensure(n_max * (size_t)stride + 16)); HIPCHK(c, d.g_len.ensure(n_max + 1));
    HIPCHK(c, d.g_keys.ensure(tsize)); HIPCHK(c, d.g_first.ensure(tsize)); HIPCHK(c, d.g_slot.ensure(n)); HIPCHK(c, d.g_rep.ensure(n));
    HIPCHK(c, d.g_hash.ensure(n)); HIPCHK(c, d.g_mask.ensure(n_words + 1)); HIPCHK(c, d.g_prefix.ensure(n_words + 1));
    HIPCHK(c, d.g_bsum.ensure((n_words + 255) / 256 + 2)); HIPCHK(c, d.g_idx.ensure(n));
    HIPCHK(c, d.gx_chars.ensure((size_t)n * stride + 16)); HIPCHK(c, d.gx_len.ensure(n));
    HIPCHK(c, d.h_gmap.ensure(n)); HIPCHK(c, d.h_gx_chars.ensure((size_t)n * stride + 16)); HIPCHK(c, d.h_gx_len.ensure(n)); HIPCHK(c, d.h_gx_hash.ensure(n));
    return CRASS_OK;
}

int crass_hip_exchange_setup(crass_hip_ctx *c, uint32_t world, uint32_t rank, uint64_t cap_rows, crass_exchange *o)
{
    if (!c || !o || world == 0 || rank >= world || cap_rows == 0 || cap_rows > (1u << 22)) return CRASS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    crass_hip_ctx::Xchg &X = c->xchg;
    static const bool dbg = getenv("CRASS_GROUP_DEBUG") != nullptr;
#define XDBG(msg) do { if (dbg) { fprintf(stderr, "[crass_xchg] rank %u: %s\n", rank, msg); fflush(stderr); } } while (0)
    XDBG("setup: quiesce");
    quiesce_worker(c);
    c->p1d.active = false;                              // (a deferred pass 1 nobody finished: dropped; the stream is waited for below)
    XDBG("setup: stream sync");
    HIPCHK(c, hipStreamSynchronize(c->stream));          // (kernels of an abandoned queued merge may still be running on the old buffers)
    X.world = world; X.rank = rank; X.cap = cap_rows; X.slot = c->dr_stride + 16; X.needed = 0;
    XDBG("setup: send buffer");
    HIPCHK(c, X.send.ensure(X.send_bytes())); HIPCHK(c, X.xinfo.ensure(8)); HIPCHK(c, X.h_xinfo.ensure(8));
    XDBG("setup: memset");
    HIPCHK(c, hipMemsetAsync(X.send.p, 0, X.send_bytes(), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    XDBG("setup: send buffer ready");
    X.active = true;
    X.gx_cap_hint = 0;
    // first call (see first_call_bounds): a bound for the GLOBAL distinct list from the job's size (every rank holds ~1/world
    // of the reads), and the buffers of crass_hip_merge_gathered sized now rather than inside the first step
    if (c->have_reads && c->surv_cap_hint && c->dx_cap_hint && !c->env.no_presize && !c->env.no_speculation) {
        const uint64_t n_max = (uint64_t)world * cap_rows;
        const uint64_t n_job = (uint64_t)world * c->R.n_reads;
        uint64_t gx = std::min<uint64_t>(std::min<uint64_t>(n_max, 1u << 20), std::max<uint64_t>(16384, (n_job / 1024 + 4095) & ~4095ull));
        if (c->env.test_bounds[3]) gx = std::min<uint64_t>(n_max, std::max<uint64_t>(16, c->env.test_bounds[3]));
        if (n_max <= (1u << 22)) {
            int s = ensure_gathered_buffers(c, n_max);
            if (s) return s;
            s = device_merge_prepare(c, c->dm.gx_chars.p, c->dm.gx_len.p, gx, c->d_count.p + 4);
            if (s) return s;
            X.gx_cap_hint = (uint32_t)gx;
        }
    }
    XDBG("setup: done");
#undef XDBG
    o->d_send = X.send.p; o->send_bytes = X.send_bytes(); o->slot_bytes = X.slot; o->cap_rows = X.cap;
    return CRASS_OK;
}

uint64_t crass_hip_exchange_needed_rows(const crass_hip_ctx *c) { return c ? c->xchg.needed : 0; }

// first-call bound for a shard's distinct DR strings (first_call_bounds_impl uses the same one for the queued merge)
static uint64_t distinct_bound_for(uint64_t n_reads)
{
    return std::min<uint64_t>(1u << 20, std::max<uint64_t>(16384, (n_reads / 1024 + 4095) & ~4095ull));
}
uint64_t crass_hip_exchange_rows_for(uint64_t n_reads) { return distinct_bound_for(n_reads); }

int crass_hip_merge_gathered(crass_hip_ctx *c, const void *d_recv)
{
    if (!c || !d_recv) return CRASS_ERR_INVALID_ARG;
    const bool deferred = c->p1d.active;                // pass 1 is still running (or its counters unread): p1_finish below
    if ((!c->have_pass1 && !deferred) || !c->xchg.active) return CRASS_ERR_STATE;
    const double t0 = now_ms();
    auto lap = [&](const char *what) { if (c->env.merge_profile) fprintf(stderr, "[crass_xg] %-28s +%.1f us\n", what, 1e3 * (now_ms() - t0)); };
    crass_hip_ctx::Xchg &X = c->xchg;
    crass_hip_ctx::DM &d = c->dm;
    quiesce_worker(c);
    lap("quiesced");
    d.active = false;
    c->cnt.used_device_merge = 0; c->cnt.ms_merge_device = 0;
    (void)hipSetDevice(c->device);
    const uint32_t stride = c->dr_stride;
    const uint64_t n_max = X.world * X.cap;
    const uint32_t n = (uint32_t)n_max;
    HIPCHK(c, d.g_chars.ensure(n_max * (size_t)stride + 16)); HIPCHK(c, d.g_len.ensure(n_max + 1));
    // (deferred: what device_merge_applies asks of pass 1's results is checked once they are known)
    bool dev = (deferred ? (!c->env.host_merge && c->prm.lowDRsize >= (int)kDevMinDR && c->dr_stride <= 64) : device_merge_applies(c)) && n_max <= (1u << 22);
    uint32_t tsize = 1024;
    while (tsize < n * 2) tsize <<= 1;
    if (dev) { const int as = ensure_gathered_buffers(c, n_max); if (as) return as; }
    lap("buffers ensured");
    HIPCHK(c, launch_xg_unpack((const uint8_t *)d_recv, X.world, X.rank, stride, X.cap, X.slot, d.g_chars.p, d.g_len.p, X.xinfo.p, c->stream,
                               X.h_xinfo.p, c->d_count.p + 4,        // (the four counters also land in pinned host memory: no copy call;
                                                                     //  d_count[4..5] = the de-duplication's counters, cleared on the way)
                               dev ? d.g_keys.p : nullptr, dev ? d.g_first.p : nullptr, tsize));      // (... and its table)
    lap("unpack queued");
    if (dev) {
        // the global count lives on the device (xinfo[0]); n_max bounds it
        HIPCHK(c, launch_dr_dedupe(d.g_chars.p, d.g_len.p, stride, X.xinfo.p, n, d.g_keys.p, d.g_first.p, tsize, d.g_hash.p, d.g_slot.p, d.g_rep.p,
                                   c->stream, true));
        Lookback lbg;
        // the global list's pinned copy is only read by the host-built view (3.7 MB of PCIe stores from the gather kernel at
        // 42 k tokens: 53 us); with the view exported by the device, or a light view, it stays on the device (fetch_host_list)
        d.hx_on_host = c->env.no_device_view && !c->host_view_light;
        HIPCHK(c, launch_dx_tokens(d.g_chars.p, d.g_len.p, d.g_hash.p, stride, X.xinfo.p, n, d.g_rep.p, d.g_slot.p, d.g_first.p, d.g_mask.p, d.g_prefix.p, d.g_bsum.p,
                                   d.g_idx.p, c->d_count.p + 4, c->d_count.p + 5, d.h_gmap.p, d.hx_on_host ? d.h_gx_chars.p : nullptr,
                                   d.hx_on_host ? d.h_gx_len.p : nullptr, d.hx_on_host ? d.h_gx_hash.p : nullptr,
                                   d.gx_chars.p, d.gx_len.p, c->stream,
                                   c->d_count.p, c->h_count.p, 8, c->next_lookback_tiles(((uint64_t)n + 1023) / 1024, &lbg)));   // counters leave with the last kernel
    }
    // As in the seed scan of the one-GPU path: when the previous step's merge ran on the device, this step's merge is
    // queued right here (token count read on the device, sized by a bound) and the host only waits for the counters.
    lap("de-duplication queued");
    bool queued = false;
    if (dev && X.gx_cap_hint && X.gx_cap_hint <= n_max && !c->env.no_speculation) {
        if (!c->poll_on) {
            if (!X.ev_counts) HIPCHK(c, hipEventCreateWithFlags(&X.ev_counts, hipEventDisableTiming));
            HIPCHK(c, hipEventRecord(X.ev_counts, c->stream));
        }
        const bool prepared = c->dm_prepared_n == X.gx_cap_hint && c->dm_prepared_src == d.gx_chars.p;
        c->dm_prepared_n = 0;                               // (a repeated exchange after an overflow starts from scratch)
        const int qs = device_merge_enqueue(c, d.gx_chars.p, d.gx_len.p, X.gx_cap_hint, c->d_count.p + 4, prepared);
        if (qs) return qs;
        queued = true;
        lap("merge queued");
    }
    if (deferred) {
        // everything of this step up to the merge is queued: now the host looks at pass 1
        const int fs = p1_finish(c);
        lap("pass 1 finished");
        if (fs == kP1Redo) { X.needed = 1; X.gx_cap_hint = 0; return CRASS_ERR_OVERFLOW; }      // (every rank finds the mark in what it gathered)
        if (fs) return fs;
        if (dev && !device_merge_applies(c)) {          // (no device-resident distinct list after all, e.g. no candidate in this shard)
            HIPCHK(c, hipStreamSynchronize(c->stream));
            dev = false; queued = false;
        }
    }
    if (queued && !c->poll_on) HIPCHK(c, hipEventSynchronize(X.ev_counts));
    else if (!(queued && c->poll_flag(1, c->want_pre, 200.0))) HIPCHK(c, hipStreamSynchronize(c->stream));      // (the merge's first kernel stores the flag)
    lap("counts on the host");
    const uint64_t n_global = X.h_xinfo.p[0], my_off = X.h_xinfo.p[1];
    if (X.h_xinfo.p[2]) { X.needed = X.h_xinfo.p[3]; X.gx_cap_hint = 0; return CRASS_ERR_OVERFLOW; }
    if (dev && n_global && c->h_count.p[5] == 0 && c->h_count.p[4] != 0 && c->h_count.p[4] <= (1u << 20) && my_off + c->n_dx <= n_global) {
        d.global = true; d.my_off = my_off; d.n_global = n_global;
        const uint32_t n_tok = c->h_count.p[4];
        if (queued && n_tok > X.gx_cap_hint) c->n_bound_overflows[3]++;
        const int s = (queued && n_tok <= X.gx_cap_hint) ? device_merge_commit(c, n_tok, d.h_gx_chars.p, d.h_gx_len.p)
                                                         : device_merge(c, d.gx_chars.p, d.gx_len.p, n_tok, d.h_gx_chars.p, d.h_gx_len.p);
        lap("committed");
        if (s == CRASS_OK) {
            c->cnt.used_device_merge = 1; c->cnt.ms_merge_host = (float)(now_ms() - t0);
            X.gx_cap_hint = (uint32_t)std::min<uint64_t>(n_max, ((uint64_t)n_tok * 3 / 2 + 4095) & ~4095ull);
            return CRASS_OK;
        }
        if (s != CRASS_ERR_STATE) return s;
    }
    X.gx_cap_hint = 0;
    // host merge
    d.global = false;
    std::vector<char> gc(n_global * (size_t)stride);
    std::vector<uint16_t> gl(n_global);
    if (n_global) {
        HIPCHK(c, hipMemcpy(gc.data(), d.g_chars.p, gc.size(), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(gl.data(), d.g_len.p, gl.size() * 2, hipMemcpyDeviceToHost));
    }
    return merge_global_host(c, gc.data(), gl.data(), stride, n_global, my_off, t0);
}

int crass_hip_get_distinct_device(crass_hip_ctx *c, crass_distinct_dev *o)
{
    if (!c || !o) return CRASS_ERR_INVALID_ARG;
    if (c->p1d.active) { const int ds = settle_p1(c); if (ds) return ds; }
    if (!c->have_pass1 || !c->dev_tokens()) return CRASS_ERR_STATE;
    o->n_distinct = c->n_dx; o->dr_stride = c->dr_stride; o->d_chars = c->dd_dx_chars.p; o->d_len = c->dd_dx_len.p;
    return CRASS_OK;
}

int crass_hip_merge_distinct_device(crass_hip_ctx *c, const char *d_chars, const uint16_t *d_len, uint32_t dr_stride,
                                    uint64_t n_global, uint64_t my_offset)
{
    if (!c || (n_global && (!d_chars || !d_len || !dr_stride))) return CRASS_ERR_INVALID_ARG;
    if (c->p1d.active) { const int ds = settle_p1(c); if (ds) return ds; }
    if (!c->have_pass1) return CRASS_ERR_STATE;
    if (dr_stride != c->dr_stride) return CRASS_ERR_INVALID_ARG;
    const double t0 = now_ms();
    quiesce_worker(c);
    c->dm.active = false;
    c->cnt.used_device_merge = 0; c->cnt.ms_merge_device = 0;
    (void)hipSetDevice(c->device);
    crass_hip_ctx::DM &d = c->dm;
    // engine-owned copy of the concatenation (the caller's buffers belong to its own stream and allocator)
    HIPCHK(c, d.g_chars.ensure(n_global * (size_t)dr_stride + 16)); HIPCHK(c, d.g_len.ensure(n_global + 1));
    if (n_global) {
        HIPCHK(c, hipMemcpyAsync(d.g_chars.p, d_chars, n_global * (size_t)dr_stride, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d.g_len.p, d_len, n_global * 2, hipMemcpyDeviceToDevice, c->stream));
    }
    int s = merge_global_device(c, n_global, my_offset);
    if (s == CRASS_OK) { c->cnt.used_device_merge = 1; c->cnt.ms_merge_host = (float)(now_ms() - t0); return CRASS_OK; }
    if (s != CRASS_ERR_STATE) return s;
    // host merge after all
    c->dm.global = false;
    std::vector<char> gc(n_global * (size_t)dr_stride);
    std::vector<uint16_t> gl(n_global);
    if (n_global) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipMemcpy(gc.data(), d.g_chars.p, gc.size(), hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(gl.data(), d.g_len.p, gl.size() * 2, hipMemcpyDeviceToHost));
    }
    return merge_global_host(c, gc.data(), gl.data(), dr_stride, n_global, my_offset, t0);
}

int crass_hip_get_merge(const crass_hip_ctx *c, crass_merge_view *o)
{
    if (!c || !o) return CRASS_ERR_INVALID_ARG;
    if (!c->have_merge) return CRASS_ERR_STATE;
    if (c->dm.active && !c->dm.host_built) {
        crass_hip_ctx *mc = const_cast<crass_hip_ctx *>(c);
        int s = ensure_host_merge(mc);
        if (s == CRASS_ERR_STATE) s = host_merge_fallback(mc);
        if (s) return s;
    }
    if (c->dm.active && c->dm.view_ready) {
        // the arrays the device assembled (k_dmx_*), where the DMA engine put them; the candidates' tokens are the host's
        const DevViewTotals &T = c->dm.view_tot;
        const uint8_t *b = c->dm.h_view.p;
        o->n_tokens = T.n_tok; o->tok_chars = (const char *)(b + T.lay.tok_chars); o->tok_off = (const uint64_t *)(b + T.lay.tok_off);
        o->n_candidates = c->merge.cand_token.size(); o->cand_token = c->merge.cand_token.data();
        o->n_groups = T.n_groups; o->grp_tokens = (const uint32_t *)(b + T.lay.grp_tokens); o->grp_off = (const uint64_t *)(b + T.lay.grp_off);
        o->n_patterns = 2u * T.n_kept; o->pat_chars = (const char *)(b + T.lay.pat_chars); o->pat_off = (const uint64_t *)(b + T.lay.pat_off);
        o->pat_group = (const uint32_t *)(b + T.lay.pat_group); o->next_free_gid = (int32_t)T.n_groups + 1;
        return CRASS_OK;
    }
    const MergeResult &m = c->merge;
    o->n_tokens = m.tokens.size(); o->tok_chars = m.tokens.strings.chars.data(); o->tok_off = m.tokens.strings.off.data();
    o->n_candidates = m.cand_token.size(); o->cand_token = m.cand_token.data();
    o->n_groups = (uint32_t)m.groups.size(); o->grp_tokens = m.grp_tokens.data(); o->grp_off = m.grp_off.data();
    o->n_patterns = (uint32_t)m.patterns.size(); o->pat_chars = m.patterns.chars.data(); o->pat_off = m.patterns.off.data();
    o->pat_group = m.pat_group.data(); o->next_free_gid = m.next_free_gid;
    return CRASS_OK;
}

int crass_hip_set_patterns(crass_hip_ctx *c, const char *const *patterns, const uint32_t *lengths, uint32_t n)
{
    if (!c || (n && (!patterns || !lengths))) return CRASS_ERR_INVALID_ARG;
    StringArena pats;
    for (uint32_t i = 0; i < n; i++) pats.push(patterns[i], lengths[i]);
    c->have_pass2 = false;
    return install_patterns(c, pats);
}

// ------------------------------------------------------------------------------------------
// pass 2
// ------------------------------------------------------------------------------------------
int crass_hip_recruit(crass_hip_ctx *c, const uint64_t *extra_found, uint64_t n_extra)
{
    if (!c) return CRASS_ERR_INVALID_ARG;
    if (!c->have_reads) return CRASS_ERR_STATE;
    (void)hipSetDevice(c->device);
    if (c->p1d.active) { const int ds = settle_p1(c); if (ds) return ds; }
    c->q_read.clear(); c->q_low.clear(); c->q_start.clear(); c->q_end.clear(); c->q_token.clear(); c->q_dr_len.clear(); c->q_dr.clear();
    c->have_pass2 = false;
    c->q_blob_active = false;
    c->cnt.n_pass2_found = 0;
    // (the host-loop sink's copies — over copy_stream or on DMA engines — may still be running: their sources, g_surv / g_dr /
    // g_fidx / g_ss, are the sink's own and nothing in this pass writes them, so nobody waits here.  Round 5 gathered the slots
    // in place in d_fidx, which this pass writes near its end, and the wait that needed was 0.5 ms of a 1 M x 10 kbp step)
    // findSingletons is only called when the non-redundant set is non-empty (WorkHorse.cpp:373)
    if (c->n_installed_patterns == 0) { c->have_pass2 = true; return CRASS_OK; }
    if (!c->have_patterns) return CRASS_ERR_STATE;
    const uint64_t n = c->R.n_reads;
    const uint64_t n_words = (n + 63) / 64;
    if (n_extra) {
        // additional found headers (local read indices, e.g. every header another input file found): one upload, one kernel
        for (uint64_t i = 0; i < n_extra; i++) if (extra_found[i] >= n) return CRASS_ERR_INVALID_ARG;
        HIPCHK(c, c->d_extra.ensure(n_extra));
        HIPCHK(c, hipMemcpyAsync(c->d_extra.p, extra_found, n_extra * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, launch_mark_found(c->d_extra.p, n_extra, c->R.header_id, c->d_found.p, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));        // (the caller's array may go away)
    }
    HIPCHK(c, c->stamp(5, 1));
    // fast path: anchor filter (exact superset) then an exact scan of the flagged reads only;
    // otherwise the automaton scans every read (LDS table when it fits).
    bool anchors = false, lds = false;
    const bool dmp = c->dm.active;                  // pattern set built on the device (dmerge.hip)
    // exception reads: with a device-built (pure ACGT) pattern set they go through the anchor filter and the exact
    // verification on their packed words like every other read (k_dm_verify checks the bytes of a candidate);
    // otherwise the byte-wise automaton scans them separately
    const uint64_t n_exc = dmp ? 0 : c->R.n_exc;
    if (dmp) {
        HIPCHK(c, launch_anchor_filter_dev(c->R, c->dm.M, c->d_found.p, c->d_mask.p, c->stream));
        anchors = true;
    } else if (c->have_anchors) {
        hipError_t ae = launch_anchor_filter(c->R, c->K, c->d_found.p, c->d_mask.p, c->stream);
        if (ae == hipSuccess) anchors = true;
        else if (ae != hipErrorNotSupported) { c->last_hip = (int)ae; return CRASS_ERR_HIP; }
    }
    if (!anchors || n_exc) { int fs = ensure_full_automaton(c); if (fs) return fs; }
    if (!anchors) {
        hipError_t re = launch_recruit_lds(c->R, c->A, c->d_found.p, c->d_mask.p, c->d_hit_info.p, c->stream);
        lds = (re == hipSuccess);
        if (re == hipErrorNotSupported) { HIPCHK(c, launch_recruit_general(c->R, c->A, c->d_found.p, c->d_mask.p, c->d_hit_info.p, c->stream)); }
        else if (re != hipSuccess) { c->last_hip = (int)re; return CRASS_ERR_HIP; }
    }
    if (n_exc) {
        HIPCHK(c, c->d_exc_hit.ensure(n_exc));
        HIPCHK(c, launch_recruit_exceptions(c->R, c->A, c->d_found.p, c->d_exc_hit.p, c->stream));
    }
    HIPCHK(c, c->stamp(6, 1));
    Lookback lbr;
    HIPCHK(c, launch_compact(c->d_mask.p, n_words, n, c->d_word_prefix.p, c->d_block_sums.p, c->d_idx.p, n, c->d_count.p, c->stream,
                             nullptr, 0, nullptr, 0, c->next_lookback(n_words, &lbr)));
    // Speculative tail (device merge path): verification, finish and the hand-off pack are launched with the hit
    // count still on the device, sized by a bound learnt from the previous call; the exact count arrives with the
    // final synchronisation, and a bound that was too small repeats the tail with the exact count.
    const bool spec = dmp && n_exc == 0 && c->hit_cap_hint && !c->env.no_speculation && !c->recruit_exact;
    c->recruit_exact = false;
    // (speculative: the count reaches the host with the last kernel of the tail, k_pack_p2_blob)
    if (!spec) HIPCHK(c, hipMemcpyAsync(c->h_count.p, c->d_count.p, 4, hipMemcpyDeviceToHost, c->stream));
    if (!spec) HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!spec && dmp && c->dm.h_st.p->fail) {       // the device merge gave up: host merge, then pass 2 again
        const int fs = host_merge_fallback(c);
        if (fs) return fs;
        return crass_hip_recruit(c, extra_found, n_extra);
    }
    const uint64_t n_hits = spec ? c->hit_cap_hint : c->h_count.p[0];
    const uint64_t n_slots = n_hits + n_exc;
    // (sized for the bound the next call will speculate with, so that it does not re-allocate)
    const uint64_t h_alloc = spec ? n_hits : std::max<uint64_t>(n_hits, hit_bound(n_hits));
    { const int as = ensure_recruit_buffers(c, h_alloc, n_exc, dmp && n_exc == 0, anchors); if (as) return as; }
    if (anchors) {
        if (dmp) HIPCHK(c, launch_dm_verify(c->R, c->dm.M, c->d_idx.p, c->d_count.p, n_hits, c->d_slot_info.p, c->d_slot_pid.p, c->stream));
        else HIPCHK(c, launch_recruit_list(c->R, c->A, c->d_idx.p, c->d_count.p, n_hits, c->d_slot_info.p, c->d_slot_pid.p, c->stream));
    }
    const bool dev_tokens = dmp || (anchors && c->have_pat_token && c->have_merge);
    HIPCHK(c, launch_recruit_finish(c->R, c->d_idx.p, c->d_count.p, n_hits, anchors ? c->d_slot_info.p : c->d_hit_info.p, anchors, false,
                                    dev_tokens ? c->d_slot_pid.p : nullptr, dev_tokens ? (dmp ? c->dm.M.pat_token : c->a_pat_token.p) : nullptr,
                                    c->d_rec.p, (dmp && n_exc == 0) ? nullptr : c->d_dr.p, c->dr_stride, c->stream, dmp ? c->dm.M.pat_mask : nullptr));
    if (n_exc)
        HIPCHK(c, launch_recruit_finish(c->R, nullptr, nullptr, n_exc, c->d_exc_hit.p, true, true, nullptr, nullptr, c->d_rec.p + n_hits,
                                        c->d_dr.p + n_hits * c->dr_stride, c->dr_stride, c->stream));
    HIPCHK(c, c->stamp(7, 2));
    // device merge path: the sink runs on the device too (drop the slots without a match, pack the records in
    // read order) and ONE copy brings the hand-off arrays to pinned host memory
    const bool dev_sink = dmp && n_exc == 0;
    c->q_blob_active = false;
    if (dev_sink) {
        static const bool blob12 = getenv("CRASS_P2_BLOB12") != nullptr;      // A/B switch: the 12-byte form of round 3
        // (the 9-byte form needs the token strings when the records are widened: a context whose host view is the light one —
        // the ranks of a group other than the first — keeps the 12-byte form)
        const uint32_t narrow = (c->max_len <= 255 && c->R.n_reads < (1ull << 32)) ? ((blob12 || c->host_view_light) ? 1u : 2u) : 0u;
        c->q_lay = p2_blob_layout(n_hits, narrow);
        c->q_wide_ready = false;
        Lookback lbq;
        if (n_hits) {
            HIPCHK(c, launch_pack_p2_blob(c->d_rec.p, c->d_idx.p, c->read_base, c->d_count.p, n_hits, c->d_mask.p,
                                          c->d_word_prefix.p, c->d_block_sums.p, c->d_fidx.p, c->d_count.p + 6, c->h_qblob.p, c->stream,
                                          spec ? c->h_count.p : nullptr, c->next_lookback_elems(n_hits, &lbq), narrow));
        } else memset(c->h_qblob.p, 0, 16);
        const double th0 = now_ms();
        const int hs = ensure_host_merge(c);            // host view of the merge, rebuilt while the device verifies
        c->cnt.ms_merge_host += (float)(now_ms() - th0);
        if (c->env.merge_profile)
            fprintf(stderr, "[crass_dm] recruit: tail queued %.1f us after the pass-1 sync, then blocked %.1f us on the host view\n",
                    1e3 * (th0 - c->t_p1_sync), 1e3 * (now_ms() - th0));
        if (hs == CRASS_ERR_STATE) {                    // inconsistent device results: never expected
            HIPCHK(c, hipStreamSynchronize(c->stream));
            const int fs = host_merge_fallback(c);
            if (fs) return fs;
            return crass_hip_recruit(c, extra_found, n_extra);
        }
        if (hs) return hs;
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->env.merge_profile)
            fprintf(stderr, "[crass_dm] recruit: final synchronisation returned %.1f us after the pass-1 sync\n", 1e3 * (now_ms() - c->t_p1_sync));
        if (spec) {
            if (c->dm.h_st.p->fail) {                   // the device merge gave up: host merge, then pass 2 again
                const int fs = host_merge_fallback(c);
                if (fs) return fs;
                return crass_hip_recruit(c, extra_found, n_extra);
            }
            if (c->h_count.p[0] > n_hits) {             // bound too small: repeat with the exact count
                c->n_bound_overflows[2]++;
                c->hit_cap_hint = 0;
                c->recruit_exact = true;
                return crass_hip_recruit(c, extra_found, n_extra);
            }
        }
        {
            c->hit_cap_hint = hit_bound(c->h_count.p[0]);
        }
        if (const int bs = c->wait_bulk()) return bs;   // the step ends with the pass-1 hand-off records in host memory too
        c->q_n = *reinterpret_cast<const uint64_t *>(c->h_qblob.p);
        c->q_blob_active = true;
        c->have_pass2 = true;
        c->cnt.n_pass2_found = c->q_n;
        c->cnt.used_lds_automaton = 2;
        c->cnt.anchor_keys = c->dm.h_st.p->n_keys;
        c->cnt.anchor_table_kind = c->dm.h_st.p->tab_mode == 3 ? 1 : c->dm.h_st.p->tab_mode;
        c->spans_p2 = true;
        return c->lookback_ok();
    }
    if (n_slots) {
        HIPCHK(c, hipMemcpyAsync(c->h_rec.p, c->d_rec.p, n_slots * sizeof(RecruitOut), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->h_dr.p, c->d_dr.p, n_slots * c->dr_stride, hipMemcpyDeviceToHost, c->stream));
    }
    if (n_hits) HIPCHK(c, hipMemcpyAsync(c->h_idx.p, c->d_idx.p, n_hits * 8, hipMemcpyDeviceToHost, c->stream));
    if (dmp) {
        // the host view of the merge (tokens, groups, pattern list) is rebuilt while the device verifies
        const double th0 = now_ms();
        const int hs = ensure_host_merge(c);
        c->cnt.ms_merge_host += (float)(now_ms() - th0);
        if (hs == CRASS_ERR_STATE) {                // inconsistent device results: never expected
            HIPCHK(c, hipStreamSynchronize(c->stream));
            const int fs = host_merge_fallback(c);
            if (fs) return fs;
            return crass_hip_recruit(c, extra_found, n_extra);
        }
        if (hs) return hs;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const double t0 = now_ms();
    // sink: merge packed hits and exception hits by read index; token = existing or new (addReadHolder)
    size_t ia = 0, ib = 0;
    const size_t nb = n_exc;
    auto exc_valid = [&](size_t b) { return c->h_rec.p[n_hits + b].dr_len != 0; };
    while (ib < nb && !exc_valid(ib)) ib++;
    auto hit_valid = [&](size_t a) { return c->h_rec.p[a].dr_len != 0; };
    while (ia < n_hits && !hit_valid(ia)) ia++;                  // anchor false positives carry no match
    c->q_dr.reserve(n_slots * c->dr_stride);
    c->q_read.reserve(n_slots); c->q_low.reserve(n_slots); c->q_start.reserve(n_slots); c->q_end.reserve(n_slots);
    c->q_dr_len.reserve(n_slots); c->q_token.reserve(n_slots);
    while (ia < n_hits || ib < nb) {
        const bool takeA = ib >= nb || (ia < n_hits && c->h_idx.p[ia] < c->h_exc_read[ib]);
        const size_t slot = takeA ? ia : n_hits + ib;
        const RecruitOut &o = c->h_rec.p[slot];
        const uint64_t r = takeA ? c->h_idx.p[ia] : c->h_exc_read[ib];
        c->q_read.push_back(c->read_base + r);
        c->q_low.push_back(o.low_lexi);
        c->q_start.push_back(o.start);
        c->q_end.push_back(o.end);
        c->q_dr_len.push_back(o.dr_len);
        const char *dr = c->h_dr.p + slot * c->dr_stride;
        const size_t at = c->q_dr.size();
        c->q_dr.resize(at + c->dr_stride);
        memset(c->q_dr.data() + at, 0, c->dr_stride);
        memcpy(c->q_dr.data() + at, dr, o.dr_len);
        uint32_t tok = o.token;
        if (!tok && c->have_merge) {
            tok = c->merge.tokens.get(dr, o.dr_len);
            if (!tok) tok = c->merge.tokens.add(dr, o.dr_len);
        }
        c->q_token.push_back(tok);
        if (takeA) { ia++; while (ia < n_hits && !hit_valid(ia)) ia++; }
        else { ib++; while (ib < nb && !exc_valid(ib)) ib++; }
    }
    c->have_pass2 = true;
    c->cnt.ms_sink_host += (float)(now_ms() - t0);
    c->cnt.n_pass2_found = c->q_read.size();
    c->cnt.used_lds_automaton = anchors ? 2 : (lds ? 1 : 0);     // 2 = anchor filter + exact list scan
    c->cnt.anchor_keys = !anchors ? 0 : (dmp ? c->dm.h_st.p->n_keys : c->K.n_keys);
    c->cnt.anchor_table_kind = !anchors ? 0 : (dmp ? (c->dm.h_st.p->tab_mode == 3 ? 1u : c->dm.h_st.p->tab_mode) : (c->K.log_size > 15 ? 2 : c->K.mode));
    c->spans_p2 = true;
    return c->lookback_ok();
}

int crass_hip_get_recruits(const crass_hip_ctx *c, crass_recruits *o)
{
    if (!c || !o) return CRASS_ERR_INVALID_ARG;
    if (!c->have_pass2) return CRASS_ERR_STATE;
    if (c->q_blob_active && !c->q_wide_ready) {
        // crass_recruits' arrays from the compact blob; the DR string of a recruit is its token's string
        crass_hip_ctx *mc = const_cast<crass_hip_ctx *>(c);
        const uint8_t *hb = c->h_qblob.p;
        const P2Blob &b = c->q_lay;
        const uint64_t n = c->q_n;
        const uint32_t *b_tok = (const uint32_t *)(hb + b.token);
        const uint16_t *b_start = (const uint16_t *)(hb + b.start), *b_end = (const uint16_t *)(hb + b.end);
        mc->q_read.resize(n); mc->q_token.resize(n); mc->q_low.resize(n);
        mc->q_start.resize(n); mc->q_end.resize(n); mc->q_dr_len.resize(n); mc->q_dr.resize(n * (size_t)c->dr_stride);
        // token strings: the device-built view (h_view) or the host-built arena
        const bool dv = c->dm.active && c->dm.view_ready;
        const uint8_t *vb = c->dm.h_view.p;
        const uint64_t *v_off = dv ? (const uint64_t *)(vb + c->dm.view_tot.lay.tok_off) : nullptr;
        const char *v_chars = dv ? (const char *)(vb + c->dm.view_tot.lay.tok_chars) : nullptr;
        const uint32_t n_tok = dv ? c->dm.view_tot.n_tok : c->merge.tokens.size();
        // ~100 bytes per recruit (436 k at 100 M reads): ranges of recruits on the host pool
        const size_t per_task = 16384;
        const size_t stride = c->dr_stride;
        host_parallel_for((size_t)((n + per_task - 1) / per_task), 16, [&](size_t task) {
            const uint64_t k0 = task * per_task, k1 = std::min<uint64_t>(n, k0 + per_task);
            memset(mc->q_dr.data() + k0 * stride, 0, (size_t)(k1 - k0) * stride);
            for (uint64_t k = k0; k < k1; k++) {
                mc->q_read[k] = b.narrow ? c->read_base + ((const uint32_t *)(hb + b.read))[k] : ((const uint64_t *)(hb + b.read))[k];
                const uint32_t t = b.narrow == 2 ? (b_tok[k] & 0x7FFFFFFFu) : b_tok[k];
                mc->q_token[k] = t;
                size_t tlen = 0;
                if (t >= 2 && t - 2 < n_tok) {
                    tlen = dv ? (size_t)(v_off[t - 1] - v_off[t - 2]) : (size_t)c->merge.tokens.strings.len(t - 2);
                    if (dv) memcpy(mc->q_dr.data() + k * stride, v_chars + v_off[t - 2], tlen);
                    else memcpy(mc->q_dr.data() + k * stride, c->merge.tokens.strings.data(t - 2), tlen);
                }
                if (b.narrow == 2) {
                    // the 9-byte form: orientation in the token's top bit; the repeat is as long as its token's string and ends
                    // at start + length - 1 in either orientation (k_recruit_finish)
                    mc->q_low[k] = (uint8_t)(b_tok[k] >> 31);
                    mc->q_start[k] = (hb + b.start)[k];
                    mc->q_dr_len[k] = (uint16_t)tlen;
                    mc->q_end[k] = mc->q_start[k] + (tlen ? (uint32_t)tlen - 1u : 0u);
                    continue;
                }
                mc->q_low[k] = (hb + b.low)[k];
                if (b.narrow) { mc->q_start[k] = (hb + b.start)[k]; mc->q_end[k] = (hb + b.end)[k]; }
                else { mc->q_start[k] = b_start[k]; mc->q_end[k] = b_end[k]; }
                mc->q_dr_len[k] = (hb + b.dr_len)[k];
            }
        });
        mc->q_wide_ready = true;
    }
    o->n = c->q_read.size(); o->read_idx = c->q_read.data(); o->low_lexi = c->q_low.data();
    o->start = c->q_start.data(); o->end = c->q_end.data(); o->dr_stride = c->dr_stride;
    o->dr_len = c->q_dr_len.data(); o->dr_chars = c->q_dr.data(); o->token = c->q_token.data();
    return CRASS_OK;
}

int crass_hip_levenshtein_batch(crass_hip_ctx *c, const char *chars, uint64_t n_chars, const uint64_t *a_off,
                                const uint32_t *a_len, const uint64_t *b_off, const uint32_t *b_len, uint64_t n_pairs,
                                int32_t *dist_out, float *sim_out)
{
    if (!c || (n_pairs && (!chars || !a_off || !a_len || !b_off || !b_len))) return CRASS_ERR_INVALID_ARG;
    if (n_pairs == 0) return CRASS_OK;
    (void)hipSetDevice(c->device);
    uint32_t max_len = 1;
    for (uint64_t k = 0; k < n_pairs; k++) {
        if (a_off[k] + a_len[k] > n_chars || b_off[k] + b_len[k] > n_chars) return CRASS_ERR_INVALID_ARG;
        max_len = std::max(max_len, std::max(a_len[k], b_len[k]));
    }
    if (max_len > 30000) return CRASS_ERR_UNSUPPORTED;
    DevBuf<uint8_t> d_chars; DevBuf<uint64_t> d_ao, d_bo; DevBuf<uint32_t> d_al, d_bl; DevBuf<int32_t> d_dist; DevBuf<float> d_sim;
    int rc = CRASS_OK;
    auto body = [&]() -> int {
        HIPCHK(c, d_chars.ensure(n_chars + 1)); HIPCHK(c, d_ao.ensure(n_pairs)); HIPCHK(c, d_bo.ensure(n_pairs));
        HIPCHK(c, d_al.ensure(n_pairs)); HIPCHK(c, d_bl.ensure(n_pairs)); HIPCHK(c, d_dist.ensure(n_pairs)); HIPCHK(c, d_sim.ensure(n_pairs));
        HIPCHK(c, hipMemcpyAsync(d_chars.p, chars, n_chars, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_ao.p, a_off, n_pairs * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_bo.p, b_off, n_pairs * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_al.p, a_len, n_pairs * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_bl.p, b_len, n_pairs * 4, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, launch_levenshtein_batch(d_chars.p, d_ao.p, d_al.p, d_bo.p, d_bl.p, n_pairs, d_dist.p, sim_out ? d_sim.p : nullptr, max_len, c->stream));
        if (dist_out) HIPCHK(c, hipMemcpyAsync(dist_out, d_dist.p, n_pairs * 4, hipMemcpyDeviceToHost, c->stream));
        if (sim_out) HIPCHK(c, hipMemcpyAsync(sim_out, d_sim.p, n_pairs * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return CRASS_OK;
    };
    rc = body();
    d_chars.release(); d_ao.release(); d_bo.release(); d_al.release(); d_bl.release(); d_dist.release(); d_sim.release();
    return rc;
}

int crass_hip_get_counters(const crass_hip_ctx *c, crass_counters *o)
{
    if (!c || !o) return CRASS_ERR_INVALID_ARG;
    crass_hip_ctx *m = const_cast<crass_hip_ctx *>(c);
    if (m->p1d.active) { const int ds = settle_p1(m); if (ds) return ds; }
    if (m->spans_p1) {
        m->spans_p1 = false;
        m->cnt.ms_filter = c->span(0, 1, 1);
        m->cnt.ms_compact = c->span(1, 2, 2);
        m->cnt.ms_survivor = c->span_survivors ? c->span(8, 9, 1) : 0.f;      // first chunk's kernel only (D2H excluded)
        m->cnt.ms_pass1_total = c->span(0, 4, 2);
    }
    if (m->spans_p2) {
        m->spans_p2 = false;
        m->cnt.ms_recruit = c->span(5, 6, 1);
        m->cnt.ms_recruit_finish = c->span(6, 7, 2);
        m->cnt.ms_pass2_total = c->span(5, 7, 2);
    }
    *o = c->cnt;
    o->n_merge_fallbacks = c->n_merge_fallbacks;
    o->last_fallback_bits = c->last_fallback_bits;
    for (int k = 0; k < 4; k++) o->n_bound_overflows[k] = c->n_bound_overflows[k];
    o->used_device_view = (c->dm.active && c->dm.view_ready) ? 1u : 0u;
    o->n_view_fallbacks = c->n_view_fallbacks.load();
    return CRASS_OK;
}

} // extern "C"
