// group.cpp — several GPUs from ONE process: crass_hip_group_* of include/crass_hip.h.
//
// A group is N contexts (engine.cpp), one per device, driven by N host threads (the caller is rank 0's thread).  The
// path shards by contiguous read ranges; the only exchange is ONE all-gather per step of every rank's distinct
// candidate DR strings between pass 1 and the merge, issued by rank 0's thread for all ranks inside
// ncclGroupStart / ncclGroupEnd on the contexts' own streams — the canonical single-process form of RCCL
// (communicators from ncclCommInitAll).  RCCL is bound at run time (dlopen of librccl.so.1: the library stays loadable
// where RCCL is absent, and a process that already holds an RCCL — torch's — shares it); a failed RCCL call is
// reported as CRASS_ERR_RCCL with its error string, never retried by other means.
// Everything here goes through the public C ABI of the contexts; nothing computes a search result.
#include "../../include/crass_hip.h"

#include <hip/hip_runtime.h>
#include "devmem.h"
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

std::mutex g_err_mu;
std::string g_last_error;
void set_error(const std::string &s) { std::lock_guard<std::mutex> lk(g_err_mu); g_last_error = s; }

// the six RCCL entry points the group uses
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // thread-safe and idempotent: groups may be created from several threads.  CRASS_RCCL_LIB names the library to bind
    // instead (tests: a name that does not exist must end in CRASS_ERR_RCCL, not in a crash)
    bool load()
    {
        std::lock_guard<std::mutex> lk(mu);
        if (lib) return true;
        const char *forced = getenv("CRASS_RCCL_LIB");
        const char *names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        std::string why;
        if (forced && *forced) {
            lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!lib) { const char *e = dlerror(); why = e ? e : "dlopen failed"; }      // (dlerror() clears the state: read it ONCE)
        } else {
            for (const char *n : names) {
                lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (lib) break;
                const char *e = dlerror();
                if (why.empty()) why = e ? e : "dlopen(librccl.so.1) failed";
            }
        }
        if (!lib) { set_error("RCCL not found: " + why); return false; }
        bool missing = false;
        auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p) { set_error(std::string("RCCL symbol missing: ") + n); missing = true; } return p; };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        CommCount = (decltype(CommCount))sym("ncclCommCount");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (missing) { dlclose(lib); lib = nullptr; return false; }
        return true;
    }
    std::mutex mu;
};
Rccl g_rccl;

// sense-reversing barrier; waiters spin briefly (the ranks of a step normally arrive within microseconds of each other), then
// sleep on a condition variable: N - 1 cores at 100 % for as long as the slowest shard takes are not ours to burn
class Barrier {
public:
    explicit Barrier(int n) : n_(n) {}
    void wait()
    {
        const int gen = gen_.load(std::memory_order_acquire);
        if (count_.fetch_add(1, std::memory_order_acq_rel) + 1 == n_) {
            count_.store(0, std::memory_order_relaxed);
            { std::lock_guard<std::mutex> lk(m_); gen_.store(gen + 1, std::memory_order_release); }
            cv_.notify_all();
            return;
        }
        for (unsigned spin = 0; spin < 20000; spin++) {
            if (gen_.load(std::memory_order_acquire) != gen) return;
            if (spin > 2000) std::this_thread::yield();
        }
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != gen; });
    }
private:
    const int n_;
    std::atomic<int> count_{0}, gen_{0};
    std::mutex m_;
    std::condition_variable cv_;
};

enum Phase : unsigned { PH_SEED = 1, PH_MERGE = 2, PH_RECRUIT = 4, PH_LOAD = 8 };

const bool g_debug = getenv("CRASS_GROUP_DEBUG") != nullptr;           // stage trace on stderr
#define GDBG(...) do { if (g_debug) { fprintf(stderr, "[crass_group] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

} // namespace

struct crass_hip_group {
    int n = 0;
    std::vector<int> devices;
    std::vector<crass_hip_ctx *> ctx;
    bool local_copies = false;
    std::vector<ncclComm_t> comms;
    int rccl_ranks = 0;
    // exchange buffers: rank r's send buffer belongs to its context (crass_hip_exchange_setup), recv[r] is ours
    std::vector<crass_exchange> xc;
    std::vector<void *> recv;
    std::vector<hipEvent_t> ev_send;            // local copies only: behind rank r's pass 1 on its stream (the seed scans return before their kernels end)
    uint64_t cap_rows = 16384;                  // rows per rank in the exchange; sized from the largest shard at load unless forced
    bool cap_rows_forced = false;               // CRASS_GROUP_CAP_ROWS (tests: the overflow path)
    // helper threads (ranks 1 .. n-1); the caller's thread is rank 0
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv, cv_done;
    uint64_t job_id = 0;
    unsigned job_phases = 0;
    bool quit = false;
    std::atomic<int> done{0};
    Barrier *bar = nullptr;
    std::vector<int> status;                    // per rank, of the current job
    std::atomic<int> failed{0};                 // some rank failed: the others skip their work but still meet at the barriers
    std::atomic<uint64_t> need_rows{0};
    // sharding
    uint64_t n_reads = 0, base = 0;                // base: the caller's read_index_base
    std::vector<uint64_t> first;                // [n+1] first global read of every shard
    std::vector<crass_reads> shard;             // per-rank views into the caller's / our rebased arrays (load only)
    std::vector<std::vector<uint64_t>> sh_word_off, sh_exc_read, sh_exc_off, sh_header_id;
    // duplicate headers across shards: global header id -> local index of the first read with it, per rank
    bool have_dups = false;
    std::vector<uint64_t> hid_global;           // [n_reads] copy of header_id (only when have_dups)
    std::vector<std::unordered_map<uint64_t, uint64_t>> dup_local;
    std::vector<std::vector<uint64_t>> extra;   // per rank: local read indices to mark found before pass 2 (other shards' pass-1 hits)
    std::vector<std::vector<uint64_t>> extra_user;   // per rank: the caller's extra_found, routed to the shards
    bool loaded = false, have_p1 = false, have_merge = false, have_p2 = false, explicit_patterns = false;
    // concatenated hand-off (group getters)
    struct {
        std::vector<uint64_t> read, ss_off; std::vector<uint8_t> low; std::vector<uint32_t> replen, nss, ss; std::vector<uint16_t> dr_len;
        std::vector<char> dr; uint32_t stride = 0, max_len = 0; bool ready = false;
    } C;
    struct { std::vector<uint32_t> cand_token; bool ready = false; } M;
    struct {
        std::vector<uint64_t> read; std::vector<uint8_t> low; std::vector<uint32_t> start, end, token; std::vector<uint16_t> dr_len;
        std::vector<char> dr; uint32_t stride = 0; bool ready = false;
    } Q;
};

namespace {

int rccl_fail(const char *what, ncclResult_t r)
{
    set_error(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error") + " (" + std::to_string((int)r) + ")");
    return CRASS_ERR_RCCL;
}

int setup_exchange(crass_hip_group *g, int r)
{
    int s = crass_hip_exchange_setup(g->ctx[r], (uint32_t)g->n, (uint32_t)r, g->cap_rows, &g->xc[r]);
    if (s) return s;
    if (hipSetDevice(g->devices[r]) != hipSuccess) return CRASS_ERR_HIP;
    if (g->recv[r]) { crass::dev_free(g->recv[r]); g->recv[r] = nullptr; }
    if (crass::dev_alloc(&g->recv[r], (size_t)g->n * g->xc[r].send_bytes) != hipSuccess) return CRASS_ERR_OOM;
    // the seed scan returns with pass 1 queued; the collective and the merge's kernels are queued behind it before the host looks
    // at a counter (include/crass_hip.h).  CRASS_GROUP_SYNC_P1: the A/B switch (the round-5 order: wait, then exchange)
    const bool sync_p1 = getenv("CRASS_GROUP_SYNC_P1") != nullptr;        // (read per set-up: the tests flip it)
    return crass_hip_exchange_set_deferred(g->ctx[r], sync_p1 ? 0 : 1);
}

// rank 0's thread, between two barriers: every rank's send buffer is complete (its seed scan has returned)
int all_gather(crass_hip_group *g)
{
    const size_t bytes = (size_t)g->xc[0].send_bytes;
    if (g->local_copies) {
        // (every rank's stream reads every other rank's send buffer: ordered behind that rank's pass 1 by an event — a collective
        // orders this by itself)
        for (int s = 0; s < g->n; s++) {
            if (hipSetDevice(g->devices[s]) != hipSuccess) return CRASS_ERR_HIP;
            if (!g->ev_send[s] && hipEventCreateWithFlags(&g->ev_send[s], hipEventDisableTiming) != hipSuccess) return CRASS_ERR_HIP;
            if (hipEventRecord(g->ev_send[s], (hipStream_t)crass_hip_stream(g->ctx[s])) != hipSuccess) return CRASS_ERR_HIP;
        }
        for (int r = 0; r < g->n; r++) {
            if (hipSetDevice(g->devices[r]) != hipSuccess) return CRASS_ERR_HIP;
            hipStream_t st = (hipStream_t)crass_hip_stream(g->ctx[r]);
            for (int s = 0; s < g->n; s++)
                if (s != r && hipStreamWaitEvent(st, g->ev_send[s], 0) != hipSuccess) return CRASS_ERR_HIP;
            for (int s = 0; s < g->n; s++)
                if (hipMemcpyAsync((char *)g->recv[r] + (size_t)s * bytes, g->xc[s].d_send, bytes, hipMemcpyDefault, st) != hipSuccess) return CRASS_ERR_HIP;
        }
        return CRASS_OK;
    }
    ncclResult_t e = g_rccl.GroupStart();
    if (e != ncclSuccess) return rccl_fail("ncclGroupStart", e);
    for (int r = 0; r < g->n; r++) {
        e = g_rccl.AllGather(g->xc[r].d_send, g->recv[r], bytes, ncclChar, g->comms[r], (hipStream_t)crass_hip_stream(g->ctx[r]));
        if (e != ncclSuccess) { (void)g_rccl.GroupEnd(); return rccl_fail("ncclAllGather", e); }
    }
    e = g_rccl.GroupEnd();
    if (e != ncclSuccess) return rccl_fail("ncclGroupEnd", e);
    return CRASS_OK;
}

void note(crass_hip_group *g, int r, int s)
{
    if (s && !g->status[r]) g->status[r] = s;
    if (s) g->failed.store(1, std::memory_order_release);
}

// found headers that also occur in other shards: marked there before pass 2 (readsFound, libcrispr.cpp:138,411)
void collect_extra(crass_hip_group *g)
{
    for (auto &e : g->extra) e.clear();
    if (!g->have_dups) return;
    for (int r = 0; r < g->n; r++) {
        crass_candidates c;
        if (crass_hip_get_candidates(g->ctx[r], &c) != CRASS_OK) continue;
        for (uint64_t k = 0; k < c.n; k++) {
            const uint64_t h = g->hid_global[c.read_idx[k] - g->base];
            for (int q = 0; q < g->n; q++) {
                if (q == r) continue;
                auto it = g->dup_local[q].find(h);
                if (it != g->dup_local[q].end()) g->extra[q].push_back(it->second);
            }
        }
    }
}

// one rank's part of a job; every rank passes the same barriers whatever happens
void run_rank(crass_hip_group *g, int r, unsigned phases)
{
    crass_hip_ctx *c = g->ctx[r];
    auto ok = [&] { return g->failed.load(std::memory_order_acquire) == 0; };
    if (phases & PH_LOAD) {
        int s = crass_hip_load_reads(c, &g->shard[r]);
        if (!s) s = setup_exchange(g, r);
        note(g, r, s);
        return;
    }
    if ((phases & PH_SEED) && ok()) note(g, r, crass_hip_seed_scan(c));
    if (phases & PH_MERGE) {
        for (int attempt = 0; attempt < 6; attempt++) {
            GDBG("rank %d attempt %d: at barrier A", r, attempt);
            g->bar->wait();                                         // every send buffer is complete
            if (r == 0) { g->need_rows.store(0); if (ok()) note(g, 0, all_gather(g)); }
            g->bar->wait();                                         // the collective is queued on every rank's stream
            GDBG("rank %d attempt %d: merge_gathered", r, attempt);
            int s = ok() ? crass_hip_merge_gathered(c, g->recv[r]) : CRASS_OK;
            GDBG("rank %d attempt %d: merge_gathered -> %d", r, attempt, s);
            if (s == CRASS_ERR_OVERFLOW) {
                uint64_t need = crass_hip_exchange_needed_rows(c), cur = g->need_rows.load();
                while (need > cur && !g->need_rows.compare_exchange_weak(cur, need)) {}
                s = CRASS_OK;
            }
            note(g, r, s);
            g->bar->wait();                                         // every rank knows whether the lists fitted
            const uint64_t need = g->need_rows.load();
            if (!need || !ok()) break;
            if (attempt == 5) { note(g, r, CRASS_ERR_OVERFLOW); break; }
            // some rank's list did not fit: larger buffers everywhere, pass 1 again (its kernel fills the send buffer)
            if (r == 0) { uint64_t cap = g->cap_rows; while (cap < need * 2) cap *= 2; g->cap_rows = cap; }
            g->bar->wait();
            GDBG("rank %d attempt %d: %llu rows needed, exchange set up again with %llu", r, attempt, (unsigned long long)need, (unsigned long long)g->cap_rows);
            int t = setup_exchange(g, r);
            GDBG("rank %d attempt %d: setup -> %d, pass 1 again", r, attempt, t);
            if (!t) t = crass_hip_seed_scan(c);
            GDBG("rank %d attempt %d: pass 1 -> %d", r, attempt, t);
            note(g, r, t);
        }
        if (g->have_dups) {
            g->bar->wait();
            if (r == 0 && ok()) { try { collect_extra(g); } catch (const std::bad_alloc &) { note(g, 0, CRASS_ERR_OOM); } }
            g->bar->wait();
        }
    }
    if ((phases & PH_RECRUIT) && ok()) {
        std::vector<uint64_t> &e = g->extra[r];
        const std::vector<uint64_t> &u = g->extra_user[r];
        // (no barrier behind this point: an allocation failure here only ends this rank's part)
        try {
            if (!u.empty()) {                                       // (this call's; extra[r] is rebuilt by the next merge)
                std::vector<uint64_t> both(e);
                both.insert(both.end(), u.begin(), u.end());
                note(g, r, crass_hip_recruit(c, both.data(), both.size()));
            } else note(g, r, crass_hip_recruit(c, e.empty() ? nullptr : e.data(), e.size()));
        } catch (const std::bad_alloc &) { note(g, r, CRASS_ERR_OOM); }
    }
}

void helper_loop(crass_hip_group *g, int r)
{
    uint64_t seen = 0;
    (void)hipSetDevice(g->devices[r]);                  // this thread's current device (every context call sets it again)
    for (;;) {
        unsigned phases;
        {
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv.wait(lk, [&] { return g->quit || g->job_id != seen; });
            if (g->quit) return;
            seen = g->job_id; phases = g->job_phases;
        }
        run_rank(g, r, phases);
        g->done.fetch_add(1, std::memory_order_release);
        { std::lock_guard<std::mutex> lk(g->mu); }      // (the waiter is either before its check or asleep: no lost wake-up)
        g->cv_done.notify_one();
    }
}

int dispatch(crass_hip_group *g, unsigned phases)
{
    std::fill(g->status.begin(), g->status.end(), 0);
    g->failed.store(0);
    g->done.store(0);
    if (g->n > 1) {
        { std::lock_guard<std::mutex> lk(g->mu); g->job_id++; g->job_phases = phases; }
        g->cv.notify_all();
    }
    run_rank(g, 0, phases);
    for (unsigned spin = 0; g->done.load(std::memory_order_acquire) != g->n - 1; spin++) {
        if (spin > 20000) {                             // (rank 0 finished well ahead of the others: sleep until the last one reports)
            std::unique_lock<std::mutex> lk(g->mu);
            g->cv_done.wait_for(lk, std::chrono::milliseconds(2), [&] { return g->done.load(std::memory_order_acquire) == g->n - 1; });
        } else if (spin > 2000) std::this_thread::yield();
    }
    for (int r = 0; r < g->n; r++) if (g->status[r]) return g->status[r];
    return CRASS_OK;
}

} // namespace

extern "C" {

const char *crass_hip_group_last_error(void)
{
    static thread_local std::string copy;
    std::lock_guard<std::mutex> lk(g_err_mu);
    copy = g_last_error;
    return copy.c_str();
}

int crass_hip_group_create(const crass_params *p, const int *devices, int n, unsigned flags, crass_hip_group **out)
{
    if (!p || !devices || !out || n <= 0 || n > 64 || (flags & ~CRASS_GROUP_LOCAL_COPIES)) return CRASS_ERR_INVALID_ARG;
    *out = nullptr;
    set_error("");
    bool dup = false;
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) if (devices[i] == devices[j]) dup = true;
    const bool local = (flags & CRASS_GROUP_LOCAL_COPIES) != 0;
    if (dup && !local) { set_error("a device is listed twice: RCCL needs one rank per device (CRASS_GROUP_LOCAL_COPIES for tests)"); return CRASS_ERR_INVALID_ARG; }
    if (!local && !g_rccl.load()) return CRASS_ERR_RCCL;            // (before any context: no RCCL, no group — text in crass_hip_group_last_error)
    if (!local && getenv("CRASS_GROUP_INJECT_RCCL_FAIL")) {         // tests: what a caller sees when ncclCommInitAll refuses (its fall-back path)
        set_error("ncclCommInitAll: injected failure (CRASS_GROUP_INJECT_RCCL_FAIL)");
        return CRASS_ERR_RCCL;
    }
    crass_hip_group *g = new (std::nothrow) crass_hip_group();
    if (!g) return CRASS_ERR_OOM;
    g->n = n; g->devices.assign(devices, devices + n); g->local_copies = local;
    g->ctx.assign(n, nullptr); g->xc.assign(n, crass_exchange{}); g->recv.assign(n, nullptr); g->status.assign(n, 0); g->ev_send.assign(n, nullptr);
    g->extra.resize(n); g->extra_user.resize(n); g->dup_local.resize(n);
    if (const char *e = getenv("CRASS_GROUP_CAP_ROWS")) { g->cap_rows = (uint64_t)std::max(1, atoi(e)); g->cap_rows_forced = true; }      // (tests: force the overflow path)
    for (int r = 0; r < n; r++) {
        const int s = crass_hip_create(p, devices[r], &g->ctx[r]);
        if (s) { crass_hip_group_destroy(g); return s; }
        if (r > 0) (void)crass_hip_set_host_view(g->ctx[r], 1);       // ONE host view for the group: rank 0's
    }
    if (!local) {
        g->comms.assign(n, nullptr);
        const ncclResult_t e = g_rccl.CommInitAll(g->comms.data(), n, devices);
        if (e != ncclSuccess) { g->comms.clear(); crass_hip_group_destroy(g); return rccl_fail("ncclCommInitAll", e); }
        int cnt = 0;
        const ncclResult_t e2 = g_rccl.CommCount(g->comms[0], &cnt);
        if (e2 != ncclSuccess) { crass_hip_group_destroy(g); return rccl_fail("ncclCommCount", e2); }
        g->rccl_ranks = cnt;
    }
    g->bar = new Barrier(n);
    for (int r = 1; r < n; r++) g->threads.emplace_back(helper_loop, g, r);
    *out = g;
    return CRASS_OK;
}

void crass_hip_group_destroy(crass_hip_group *g)
{
    if (!g) return;
    { std::lock_guard<std::mutex> lk(g->mu); g->quit = true; }
    g->cv.notify_all();
    for (auto &t : g->threads) t.join();
    for (int r = 0; r < g->n; r++) {
        if (g->recv[r]) { (void)hipSetDevice(g->devices[r]); crass::dev_free(g->recv[r]); }
        if (g->ctx[r]) {
            (void)hipSetDevice(g->devices[r]);
            (void)hipStreamSynchronize((hipStream_t)crass_hip_stream(g->ctx[r]));
        }
    }
    for (int r = 0; r < g->n; r++) if (g->ev_send[r]) { (void)hipSetDevice(g->devices[r]); (void)hipEventDestroy(g->ev_send[r]); }
    for (auto cm : g->comms) if (cm) (void)g_rccl.CommDestroy(cm);
    for (auto c : g->ctx) if (c) crass_hip_destroy(c);
    delete g->bar;
    delete g;
}

int crass_hip_group_size(const crass_hip_group *g) { return g ? g->n : 0; }
int crass_hip_group_rccl_ranks(const crass_hip_group *g) { return g ? g->rccl_ranks : 0; }
crass_hip_ctx *crass_hip_group_ctx(crass_hip_group *g, int rank) { return (g && rank >= 0 && rank < g->n) ? g->ctx[rank] : nullptr; }

int crass_hip_group_load_reads(crass_hip_group *g, const crass_reads *h)
{
    if (!g || !h) return CRASS_ERR_INVALID_ARG;
    if (h->n_reads && !h->packed) return CRASS_ERR_INVALID_ARG;
    if (!h->stride_words && h->n_reads && !h->word_off) return CRASS_ERR_INVALID_ARG;
    if (!h->uniform_len && h->n_reads && !h->lengths) return CRASS_ERR_INVALID_ARG;
    if (h->n_exceptions && (!h->exc_read || !h->exc_off || !h->exc_bytes)) return CRASS_ERR_INVALID_ARG;
    const int N = g->n;
    const uint64_t n = h->n_reads;
    g->loaded = g->have_p1 = g->have_merge = g->have_p2 = false;
    g->C.ready = g->M.ready = g->Q.ready = false;
    g->n_reads = n; g->base = h->read_index_base;
    g->first.assign(N + 1, 0);
    for (int r = 0; r <= N; r++) g->first[r] = n * (uint64_t)r / (uint64_t)N;
    g->shard.assign(N, crass_reads{});
    g->sh_word_off.assign(N, {}); g->sh_exc_read.assign(N, {}); g->sh_exc_off.assign(N, {}); g->sh_header_id.assign(N, {});
    for (auto &m : g->dup_local) m.clear();
    for (auto &e : g->extra) e.clear();
    g->have_dups = false; g->hid_global.clear();
    // header ids that occur on more than one read (header_id[i] != i marks a repeat of read header_id[i]'s header)
    std::unordered_map<uint64_t, uint8_t> is_dup;
    if (h->header_id) {
        for (uint64_t i = 0; i < n; i++) if (h->header_id[i] != i) { if (h->header_id[i] > i) return CRASS_ERR_INVALID_ARG; is_dup[h->header_id[i]] = 1; }
        if (!is_dup.empty()) { g->have_dups = true; g->hid_global.assign(h->header_id, h->header_id + n); }
    }
    for (int r = 0; r < N; r++) {
        const uint64_t lo = g->first[r], hi = g->first[r + 1], m = hi - lo;
        crass_reads &s = g->shard[r];
        s.n_reads = m; s.stride_words = h->stride_words; s.uniform_len = h->uniform_len;
        s.read_index_base = h->read_index_base + lo;
        if (h->stride_words) s.packed = h->packed + lo * (uint64_t)h->stride_words;
        else if (m) {
            const uint64_t w0 = h->word_off[lo];
            std::vector<uint64_t> &wo = g->sh_word_off[r];
            wo.resize(m);
            for (uint64_t i = 0; i < m; i++) wo[i] = h->word_off[lo + i] - w0;
            s.packed = h->packed + w0; s.word_off = wo.data();
        } else s.packed = h->packed;
        if (!h->uniform_len) s.lengths = h->lengths + lo;
        if (h->n_exceptions) {
            const uint64_t *eb = std::lower_bound(h->exc_read, h->exc_read + h->n_exceptions, lo);
            const uint64_t *ee = std::lower_bound(h->exc_read, h->exc_read + h->n_exceptions, hi);
            const uint64_t e0 = (uint64_t)(eb - h->exc_read), ne = (uint64_t)(ee - eb);
            if (ne) {
                std::vector<uint64_t> &er = g->sh_exc_read[r], &eo = g->sh_exc_off[r];
                er.resize(ne); eo.resize(ne + 1);
                const uint64_t b0 = h->exc_off[e0];
                for (uint64_t i = 0; i < ne; i++) { er[i] = h->exc_read[e0 + i] - lo; eo[i] = h->exc_off[e0 + i] - b0; }
                eo[ne] = h->exc_off[e0 + ne] - b0;
                s.n_exceptions = ne; s.exc_read = er.data(); s.exc_off = eo.data(); s.exc_bytes = h->exc_bytes + b0;
            }
        }
        if (g->have_dups) {
            // local header ids: the first read OF THIS SHARD with the same header; headers that occur more than once anywhere
            // are remembered so that other shards' pass-1 hits can be marked here (collect_extra)
            std::vector<uint64_t> &hl = g->sh_header_id[r];
            hl.resize(m);
            std::unordered_map<uint64_t, uint64_t> &loc = g->dup_local[r];
            bool any_local_dup = false;
            for (uint64_t i = 0; i < m; i++) {
                const uint64_t gh = h->header_id[lo + i];
                if (gh == lo + i && !is_dup.count(gh)) { hl[i] = i; continue; }
                auto it = loc.find(gh);
                if (it == loc.end()) { loc.emplace(gh, i); hl[i] = i; }
                else { hl[i] = it->second; any_local_dup = true; }
            }
            if (any_local_dup) s.header_id = hl.data();
        }
    }
    // the exchange's row capacity from the largest shard (every rank must use the same one): the first step then neither
    // overflows nor repeats pass 1
    if (!g->cap_rows_forced) g->cap_rows = crass_hip_exchange_rows_for((n + (uint64_t)N - 1) / (uint64_t)N);
    const int st = dispatch(g, PH_LOAD);
    // (the shard views point into the caller's arrays: they are only used inside this call)
    g->sh_word_off.clear(); g->sh_exc_read.clear(); g->sh_exc_off.clear(); g->sh_header_id.clear();
    if (st) return st;
    g->loaded = true;
    return CRASS_OK;
}

static int run(crass_hip_group *g, unsigned phases)
{
    if (!g) return CRASS_ERR_INVALID_ARG;
    if (!g->loaded) return CRASS_ERR_STATE;
    if ((phases & PH_MERGE) && !(phases & PH_SEED) && !g->have_p1) return CRASS_ERR_STATE;
    if ((phases & PH_RECRUIT) && !(phases & PH_MERGE) && !g->have_merge) return CRASS_ERR_STATE;
    g->C.ready = (phases & PH_SEED) ? false : g->C.ready;
    g->M.ready = false; g->Q.ready = false;
    if (phases & PH_SEED) g->have_p1 = g->have_merge = g->have_p2 = false;
    if (phases & PH_MERGE) { g->have_merge = g->have_p2 = false; g->C.ready = false; g->explicit_patterns = false; }      // (an overflow repeats pass 1)
    if (phases & PH_RECRUIT) g->have_p2 = false;
    const int s = dispatch(g, phases);
    if (s) return s;
    if (phases & PH_SEED) g->have_p1 = true;
    if (phases & PH_MERGE) g->have_merge = true;
    if (phases & PH_RECRUIT) g->have_p2 = true;
    return CRASS_OK;
}

// found headers named by the caller (job-level read indices) -> local read indices of every shard that holds the header
static int route_extra(crass_hip_group *g, const uint64_t *extra_found, uint64_t n_extra)
{
    for (auto &e : g->extra_user) e.clear();
    for (uint64_t k = 0; k < n_extra; k++) {
        const uint64_t i = extra_found[k];
        if (i >= g->n_reads) return CRASS_ERR_INVALID_ARG;
        const int r = (int)(std::upper_bound(g->first.begin(), g->first.end(), i) - g->first.begin()) - 1;
        bool routed = false;
        if (g->have_dups) {
            const uint64_t h = g->hid_global[i];
            for (int q = 0; q < g->n; q++) {
                auto it = g->dup_local[q].find(h);
                if (it != g->dup_local[q].end()) { g->extra_user[q].push_back(it->second); routed = routed || q == r; }
            }
        }
        if (!routed) g->extra_user[r].push_back(i - g->first[r]);
    }
    return CRASS_OK;
}

int crass_hip_group_seed_scan(crass_hip_group *g) { return run(g, PH_SEED); }
int crass_hip_group_merge(crass_hip_group *g) { return run(g, PH_MERGE); }
int crass_hip_group_recruit(crass_hip_group *g, const uint64_t *extra_found, uint64_t n_extra)
{
    if (!g || (n_extra && !extra_found)) return CRASS_ERR_INVALID_ARG;
    if (!g->loaded) return CRASS_ERR_STATE;
    const int s = route_extra(g, extra_found, n_extra);
    if (s) return s;
    const int rs = run(g, PH_RECRUIT);
    for (auto &e : g->extra_user) e.clear();
    return rs;
}
int crass_hip_group_step(crass_hip_group *g)
{
    if (g) for (auto &e : g->extra_user) e.clear();
    return run(g, PH_SEED | PH_MERGE | PH_RECRUIT);
}

// an explicit pattern list on every rank (findSingletons' argument, libcrispr.h:86-92): the seam's form of pass 2, where
// createNonRedundantSet ran on the caller's side
int crass_hip_group_set_patterns(crass_hip_group *g, const char *const *patterns, const uint32_t *lengths, uint32_t n)
{
    if (!g || (n && (!patterns || !lengths))) return CRASS_ERR_INVALID_ARG;
    if (!g->loaded) return CRASS_ERR_STATE;
    g->M.ready = g->Q.ready = false; g->have_p2 = false;
    for (int r = 0; r < g->n; r++) {
        const int s = crass_hip_set_patterns(g->ctx[r], patterns, lengths, n);
        if (s) return s;
    }
    for (auto &e : g->extra) e.clear();                 // (no device merge ran: cross-shard found headers come from the caller)
    g->have_merge = true; g->explicit_patterns = true;
    return CRASS_OK;
}

int crass_hip_group_get_candidates(crass_hip_group *g, crass_candidates *o)
{
    if (!g || !o) return CRASS_ERR_INVALID_ARG;
    if (!g->have_p1) return CRASS_ERR_STATE;
    auto &C = g->C;
    if (!C.ready) {
        C.read.clear(); C.ss_off.clear(); C.low.clear(); C.replen.clear(); C.nss.clear(); C.ss.clear(); C.dr_len.clear(); C.dr.clear();
        C.max_len = 0;
        for (int r = 0; r < g->n; r++) {
            crass_candidates c;
            const int s = crass_hip_get_candidates(g->ctx[r], &c);
            if (s) return s;
            C.stride = c.dr_stride; C.max_len = std::max(C.max_len, c.max_read_len);
            C.read.insert(C.read.end(), c.read_idx, c.read_idx + c.n);
            C.low.insert(C.low.end(), c.low_lexi, c.low_lexi + c.n);
            C.replen.insert(C.replen.end(), c.repeat_len, c.repeat_len + c.n);
            C.nss.insert(C.nss.end(), c.n_ss, c.n_ss + c.n);
            C.dr_len.insert(C.dr_len.end(), c.dr_len, c.dr_len + c.n);
            C.dr.insert(C.dr.end(), c.dr_chars, c.dr_chars + c.n * (size_t)c.dr_stride);
            for (uint64_t k = 0; k < c.n; k++) {                     // start/stops packed tightly
                C.ss_off.push_back(C.ss.size());
                C.ss.insert(C.ss.end(), c.ss_pool + c.ss_off[k], c.ss_pool + c.ss_off[k] + c.n_ss[k]);
            }
        }
        C.ready = true;
    }
    o->n = C.read.size(); o->read_idx = C.read.data(); o->low_lexi = C.low.data(); o->repeat_len = C.replen.data();
    o->n_ss = C.nss.data(); o->ss_off = C.ss_off.data(); o->ss_pool = C.ss.data(); o->dr_stride = C.stride;
    o->dr_len = C.dr_len.data(); o->dr_chars = C.dr.data(); o->max_read_len = C.max_len;
    return CRASS_OK;
}

int crass_hip_group_get_merge(crass_hip_group *g, crass_merge_view *o)
{
    if (!g || !o) return CRASS_ERR_INVALID_ARG;
    if (!g->have_merge || g->explicit_patterns) return CRASS_ERR_STATE;
    int s = crass_hip_get_merge(g->ctx[0], o);
    if (s) return s;
    if (g->n == 1) return CRASS_OK;
    if (!g->M.ready) {
        g->M.cand_token.assign(o->cand_token, o->cand_token + o->n_candidates);
        for (int r = 1; r < g->n; r++) {
            crass_merge_view v;
            s = crass_hip_get_merge(g->ctx[r], &v);
            if (s) return s;
            g->M.cand_token.insert(g->M.cand_token.end(), v.cand_token, v.cand_token + v.n_candidates);
        }
        g->M.ready = true;
    }
    o->n_candidates = g->M.cand_token.size(); o->cand_token = g->M.cand_token.data();
    return CRASS_OK;
}

int crass_hip_group_get_recruits(crass_hip_group *g, crass_recruits *o)
{
    if (!g || !o) return CRASS_ERR_INVALID_ARG;
    if (!g->have_p2) return CRASS_ERR_STATE;
    auto &Q = g->Q;
    if (!Q.ready) {
        crass_merge_view mv{};
        int s = g->explicit_patterns ? CRASS_OK : crass_hip_get_merge(g->ctx[0], &mv);     // (explicit patterns: the host sink filled the strings)
        if (s) return s;
        Q.read.clear(); Q.low.clear(); Q.start.clear(); Q.end.clear(); Q.token.clear(); Q.dr_len.clear(); Q.dr.clear();
        for (int r = 0; r < g->n; r++) {
            crass_recruits q;
            s = crass_hip_get_recruits(g->ctx[r], &q);
            if (s) return s;
            Q.stride = q.dr_stride;
            const size_t base = Q.read.size();
            Q.read.insert(Q.read.end(), q.read_idx, q.read_idx + q.n);
            Q.low.insert(Q.low.end(), q.low_lexi, q.low_lexi + q.n);
            Q.start.insert(Q.start.end(), q.start, q.start + q.n);
            Q.end.insert(Q.end.end(), q.end, q.end + q.n);
            Q.token.insert(Q.token.end(), q.token, q.token + q.n);
            Q.dr_len.insert(Q.dr_len.end(), q.dr_len, q.dr_len + q.n);
            Q.dr.insert(Q.dr.end(), q.dr_chars, q.dr_chars + q.n * (size_t)q.dr_stride);
            // a recruit's DR string is its token's string: ranks with the light host view left it empty
            for (uint64_t k = 0; k < q.n; k++) {
                const uint32_t t = q.token[k];
                if (t >= 2 && t - 2 < mv.n_tokens) {
                    const uint64_t a = mv.tok_off[t - 2], len = mv.tok_off[t - 1] - a;
                    if (len <= q.dr_stride) memcpy(Q.dr.data() + (base + k) * (size_t)q.dr_stride, mv.tok_chars + a, len);
                }
            }
        }
        Q.ready = true;
    }
    o->n = Q.read.size(); o->read_idx = Q.read.data(); o->low_lexi = Q.low.data(); o->start = Q.start.data(); o->end = Q.end.data();
    o->dr_stride = Q.stride; o->dr_len = Q.dr_len.data(); o->dr_chars = Q.dr.data(); o->token = Q.token.data();
    return CRASS_OK;
}

} // extern "C"
