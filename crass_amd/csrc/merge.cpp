// merge.cpp — the serial "merge" step between the two device passes (stays on the host:
// a few thousand DR variants, negligible next to the scans).
#include "merge.h"
#include "../../include/crass_hip.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

namespace crass {

void build_comp_table(unsigned char tab[128])
{
    // IUPAC complement pairs, U->A, everything else maps to itself; entry 96 ('`') holds 64,
    // exactly like the reference table (SeqUtils.cpp:50-59).
    for (int i = 0; i < 128; i++) tab[i] = (unsigned char)i;
    const char *a = "ACBDKRSWN", *b = "TGVHMYSWN";
    for (int i = 0; a[i]; i++) {
        tab[(int)a[i]] = (unsigned char)b[i];
        tab[(int)b[i]] = (unsigned char)a[i];
        tab[(int)a[i] + 32] = (unsigned char)(b[i] + 32);
        tab[(int)b[i] + 32] = (unsigned char)(a[i] + 32);
    }
    tab['U'] = 'A';
    tab['u'] = 'a';
    tab[96] = 64;
}

static const unsigned char *comp_table()
{
    static unsigned char tab[128];
    static bool ready = false;
    if (!ready) { build_comp_table(tab); ready = true; }
    return tab;
}

std::string reverse_complement(const std::string &s)
{
    const unsigned char *tab = comp_table();
    std::string r(s.size(), '\0');
    for (size_t i = 0; i < s.size(); i++) r[i] = (char)tab[(unsigned char)s[s.size() - 1 - i] & 127];
    return r;
}

uint64_t TokenTable::hash(const char *p, size_t n)
{
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xD6E8FEB86659FD93ull);
    while (n >= 8) { uint64_t v; memcpy(&v, p, 8); h = (h ^ v) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; p += 8; n -= 8; }
    if (n) { uint64_t v = 0; memcpy(&v, p, n); h = (h ^ v) * 0xC4CEB9FE1A85EC53ull; h ^= h >> 29; }
    return h ^ (h >> 31);
}

void TokenTable::load_strings(const char *chars, const uint16_t *len, uint32_t stride, size_t n)
{
    clear();
    size_t total = 0;
    for (size_t j = 0; j < n; j++) total += len[j];
    strings.chars.resize(total);
    strings.off.resize(n + 1);
    size_t at = 0;
    for (size_t j = 0; j < n; j++) {
        strings.off[j] = at;
        memcpy(strings.chars.data() + at, chars + j * (size_t)stride, len[j]);
        at += len[j];
    }
    strings.off[n] = at;
    index_stale = n != 0;
}

void TokenTable::ensure_index() const
{
    if (!index_stale) return;
    index_stale = false;
    size_t cap = 1024;
    while (cap < (strings.size() + 1) * 2) cap <<= 1;
    slot_token.assign(cap, 0); slot_hash.assign(cap, 0);
    for (size_t t = 0; t < strings.size(); t++) {
        const uint64_t h = hash(strings.data(t), strings.len(t));
        size_t i = (size_t)h & (cap - 1);
        while (slot_token[i]) i = (i + 1) & (cap - 1);
        slot_token[i] = (uint32_t)t + 2; slot_hash[i] = h;
    }
}

uint32_t TokenTable::get(const char *p, size_t n) const
{
    ensure_index();
    if (slot_token.empty()) return 0;
    const uint64_t h = hash(p, n);
    const size_t mask = slot_token.size() - 1;
    for (size_t i = (size_t)h & mask;; i = (i + 1) & mask) {
        const uint32_t t = slot_token[i];
        if (!t) return 0;
        if (slot_hash[i] == h && strings.len(t - 2) == n && memcmp(strings.data(t - 2), p, n) == 0) return t;
    }
}

void TokenTable::grow()
{
    ensure_index();
    const size_t cap = slot_token.empty() ? 1024 : slot_token.size() * 2;
    std::vector<uint32_t> nt(cap, 0);
    std::vector<uint64_t> nh(cap, 0);
    for (size_t i = 0; i < slot_token.size(); i++) if (slot_token[i]) {
        size_t j = (size_t)slot_hash[i] & (cap - 1);
        while (nt[j]) j = (j + 1) & (cap - 1);
        nt[j] = slot_token[i]; nh[j] = slot_hash[i];
    }
    slot_token.swap(nt); slot_hash.swap(nh);
}

uint32_t TokenTable::add(const char *p, size_t n) { return add_hashed(p, n, hash(p, n)); }

void TokenTable::reserve(size_t n_strings, size_t n_chars)
{
    ensure_index();
    size_t cap = slot_token.empty() ? 1024 : slot_token.size();
    while (cap < (n_strings + 1) * 2) cap <<= 1;
    if (cap != slot_token.size()) {
        std::vector<uint32_t> nt(cap, 0);
        std::vector<uint64_t> nh(cap, 0);
        for (size_t i = 0; i < slot_token.size(); i++) if (slot_token[i]) {
            size_t j = (size_t)slot_hash[i] & (cap - 1);
            while (nt[j]) j = (j + 1) & (cap - 1);
            nt[j] = slot_token[i]; nh[j] = slot_hash[i];
        }
        slot_token.swap(nt); slot_hash.swap(nh);
    }
    strings.chars.reserve(n_chars);
    strings.off.reserve(n_strings + 1);
}

uint32_t TokenTable::add_unique_hashed(const char *p, size_t n, uint64_t h)
{
    ensure_index();
    if ((strings.size() + 1) * 2 > slot_token.size()) grow();
    const size_t mask = slot_token.size() - 1;
    size_t i = (size_t)h & mask;
    for (; slot_token[i]; i = (i + 1) & mask) {
        const uint32_t t = slot_token[i];
        if (slot_hash[i] == h && strings.len(t - 2) == n && memcmp(strings.data(t - 2), p, n) == 0) return 0;
    }
    strings.push(p, n);
    const uint32_t t = (uint32_t)strings.size() + 1;
    slot_token[i] = t; slot_hash[i] = h;
    return t;
}

uint32_t TokenTable::add_hashed(const char *p, size_t n, uint64_t h)
{
    ensure_index();
    if ((strings.size() + 1) * 2 > slot_token.size()) grow();
    strings.push(p, n);
    const uint32_t t = (uint32_t)strings.size() + 1;
    const size_t mask = slot_token.size() - 1;
    size_t i = (size_t)h & mask;
    while (slot_token[i]) i = (i + 1) & mask;
    slot_token[i] = t; slot_hash[i] = h;
    return t;
}

void MergeResult::clear()
{
    tokens.clear(); cand_token.clear(); groups.clear(); patterns.clear(); pat_group.clear(); pat_token.clear();
    next_free_gid = 1;
    grp_tokens.clear(); grp_off.clear();
}

void MergeResult::flatten()
{
    grp_tokens.clear(); grp_off.assign(1, 0);
    for (const auto &g : groups) { grp_tokens.insert(grp_tokens.end(), g.begin(), g.end()); grp_off.push_back(grp_tokens.size()); }
}

namespace {

constexpr int kClusterKmer = 11;     // CRASS_DEF_KMER_SIZE (crassDefines.h:66)

inline int acgt_code(unsigned char c)
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

// k-mer -> GID map of WorkHorse::clusterDRReads (std::map<std::string,int> there).  ACGT-only
// 11-mers are kept as 22-bit integers (first base most significant, so integer order ==
// lexicographic order and laurenize() is a min of two integers) in a small open-addressing
// table (variants of one DR share most k-mers, so it stays cache-resident); anything else falls
// back to string keys.  Same lookups, same results, a fraction of the host time.
struct KmerGid {
    struct Slot { uint32_t key; int32_t gid; };                  // key = code + 1, 0 = empty
    std::vector<Slot> tab;
    size_t used = 0;
    std::unordered_map<std::string, int> other;
    explicit KmerGid(size_t expect = 0) : tab(1u << 14, Slot{0, 0})
    {
        size_t cap = tab.size();
        while (cap < expect * 2 && cap < (1u << 22)) cap <<= 1;    // skip the early rehashes
        if (cap != tab.size()) tab.assign(cap, Slot{0, 0});
    }
    static size_t slot(uint32_t code, size_t mask) { return (size_t)((code * 0x9E3779B1u) >> 8) & mask; }
    void prefetch(uint32_t code) const { __builtin_prefetch(&tab[slot(code, tab.size() - 1)]); }
    int32_t find(uint32_t code) const
    {
        const size_t mask = tab.size() - 1;
        for (size_t i = slot(code, mask);; i = (i + 1) & mask) {
            if (tab[i].key == 0) return 0;
            if (tab[i].key == code + 1) return tab[i].gid;
        }
    }
    void put(uint32_t code, int32_t g)                           // insert or overwrite
    {
        if ((used + 1) * 2 > tab.size()) {
            std::vector<Slot> old;
            old.swap(tab);
            tab.assign(old.size() * 2, Slot{0, 0}); used = 0;
            for (const Slot &o : old) if (o.key) put(o.key - 1, o.gid);
        }
        const size_t mask = tab.size() - 1;
        size_t i = slot(code, mask);
        while (tab[i].key != 0 && tab[i].key != code + 1) i = (i + 1) & mask;
        if (tab[i].key == 0) { tab[i].key = code + 1; used++; }
        tab[i].gid = g;
    }
};

// laurenized 11-mer codes of one DR in order (laurenize, SeqUtils.cpp:89-97: seq1 < seq2 ? seq1 : seq2);
// -1 marks a window holding a non-ACGT byte (handled through string keys).  Pure function of the
// string, so it is computed for all variants in parallel before the order-dependent greedy pass.
void kmer_codes(const char *dr, size_t dr_size, int32_t *out)
{
    const int n_mers = (int)dr_size - kClusterKmer + 1;
    uint32_t fwd = 0, rev = 0;
    int bad = 0;
    const uint32_t mask = (1u << 22) - 1;
    for (int i = 0; i < (int)dr_size; i++) {
        int c = acgt_code((unsigned char)dr[i]);
        if (c < 0) { bad = kClusterKmer; c = 0; } else if (bad > 0) bad--;
        fwd = ((fwd << 2) | (uint32_t)c) & mask;
        rev = (rev >> 2) | ((uint32_t)(3 - c) << 20);
        const int start = i - kClusterKmer + 1;
        if (start < 0 || start >= n_mers) continue;
        out[start] = (bad == 0) ? (int32_t)(fwd < rev ? fwd : rev) : -1;
    }
}

// WorkHorse::clusterDRReads (WorkHorse.cpp:1404-1637): greedy, order-dependent assignment of
// one DR variant to a group through shared laurenized 11-mers.  Returns the GID.
int cluster_one(const char *dr_p, size_t dr_n, const int32_t *codes, int n_mers, int &next_free_gid, KmerGid &kg, int min_shared,
                std::vector<uint32_t> &homeless_code, std::vector<std::pair<int, int>> &group_count)
{
    homeless_code.clear();
    group_count.clear();                                        // std::map<int,int> in the reference; tiny
    std::vector<std::string> homeless_str;
    int group = 0;
    auto seen = [&](int gid) {
        if (group != 0) return;
        for (auto &gc : group_count)
            if (gc.first == gid) { if (min_shared <= ++gc.second) group = gid; return; }
        group_count.emplace_back(gid, 1);                       // the first sighting is not tested (:1577-1580)
    };
    for (int i = 0; i < n_mers; i++) {
        if (codes[i] >= 0) {
            const int32_t gid = kg.find((uint32_t)codes[i]);
            if (gid == 0) homeless_code.push_back((uint32_t)codes[i]); else seen(gid);
        } else {
            (void)dr_n;
            std::string km(dr_p + i, kClusterKmer), rc = reverse_complement(km);
            const std::string &lau = (km < rc) ? km : rc;
            auto it = kg.other.find(lau);
            if (it == kg.other.end()) homeless_str.push_back(lau); else seen(it->second);
        }
    }
    if (group == 0) group = next_free_gid++;
    for (uint32_t k : homeless_code) kg.put(k, group);
    for (const auto &k : homeless_str) kg.other[k] = group;
    return group;
}

// A small persistent worker pool.  Spawning std::threads per call costs ~50 us each on a big host, and even
// waking a sleeping worker through a condition variable costs 100-200 us there — more than most of the
// jobs below.  Workers therefore keep polling for a couple of milliseconds after a job (one merge is a
// burst of short jobs) and only then go to sleep; warm() wakes them ahead of the burst.
class HostPool {
public:
    static HostPool &get() { static HostPool p; return p; }
    // run fn(task) for task in [0, n_tasks) on up to max_threads threads (caller included).  The caller never
    // waits for a helper to show up, only for the tasks that were actually claimed: a helper that is slow to
    // wake simply takes no part in the job.
    void run(size_t n_tasks, unsigned max_threads, const std::function<void(size_t)> &fn)
    {
        if (n_tasks == 0) return;
        const unsigned helpers = std::min<unsigned>((unsigned)workers_.size(), max_threads > 0 ? max_threads - 1 : 0);
        if (n_tasks < 2 || helpers == 0 || n_tasks > 0x7FFFFFFFull) { for (size_t t = 0; t < n_tasks; t++) fn(t); return; }
        std::lock_guard<std::mutex> job_lock(job_mutex_);      // one job at a time
        const uint64_t epoch = (ticket_.load(std::memory_order_relaxed) >> 32) + 1;
        // two descriptors, indexed by epoch parity: a helper still looking at the previous job's descriptor can
        // only claim a task while the ticket carries that job's epoch, i.e. before this one is published, and
        // the descriptor it reads is not rewritten until the job after this one
        Job &job = jobs_[epoch & 1];
        job.fn = &fn; job.n_tasks = n_tasks; job.want = helpers;
        done_.store(0, std::memory_order_relaxed);
        ticket_.store(epoch << 32);                            // seq_cst: publishes the job, ordered before the sleepers_ check
        wake_sleepers();
        size_t mine = 0;
        for (;;) {
            uint64_t tk = ticket_.load(std::memory_order_relaxed);
            if ((tk & 0xFFFFFFFFull) >= n_tasks) break;
            if (!ticket_.compare_exchange_weak(tk, tk + 1, std::memory_order_acq_rel)) continue;
            fn((size_t)(tk & 0xFFFFFFFFull));
            mine++;
        }
        done_.fetch_add(mine, std::memory_order_acq_rel);
        for (unsigned spins = 0; done_.load(std::memory_order_acquire) != n_tasks;) { cpu_relax(); if ((++spins & yield_mask_) == 0) std::this_thread::yield(); }
    }
    // wake the workers now: the next jobs, within the polling window, start without wake-up latency
    void warm()
    {
        warm_until_.store(now_us() + poll_us_, std::memory_order_relaxed);
        wake_sleepers();
    }
private:
    static int64_t now_us()
    {
        return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    void wake_sleepers()
    {
        if (sleepers_.load() == 0) return;
        { std::lock_guard<std::mutex> lk(m_); wake_seq_++; }
        cv_.notify_all();
    }
    HostPool()
    {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        unsigned n = std::min<unsigned>(hw, 16);
        // The workers poll between the jobs of a burst, i.e. they burn their cores while a search is running.  Under a CPU
        // quota (cgroup cpu.max: the GPU boxes give a 256-thread host 16 CPUs per 100 ms) a process that exceeds it is
        // frozen until the period ends: 15 pollers + the caller + the helper thread did, and every ~137th step took
        // 3-4 ms (tools/step_spikes.py).  So: at most half the quota, shared between the ranks of this node.
        {
            const double quota_cpus = host_cpu_quota();
            if (quota_cpus > 0) {
                int ranks = 1;
                if (const char *e = getenv("LOCAL_WORLD_SIZE")) ranks = std::max(1, atoi(e));
                const unsigned cap = (unsigned)std::max(2.0, quota_cpus * 0.5 / ranks);
                n = std::min(n, cap);
                if (quota_cpus / ranks < 4.0) poll_us_ = 100;          // hardly any room: the workers sleep between bursts
            }
        }
        if (const char *e = getenv("CRASS_HOST_THREADS")) n = (unsigned)std::max(1, atoi(e));    // 1 = everything on the caller
        if (const char *e = getenv("CRASS_POOL_POLL_US")) poll_us_ = atoll(e);
        // a small (possibly oversubscribed) host must not be starved by polling workers: yield often there
        yield_mask_ = hw <= 16 ? 31u : 4095u;
        for (unsigned i = 0; i + 1 < n; i++) workers_.emplace_back([this, i] { loop(i); });
    }
    ~HostPool()
    {
        stop_.store(true);
        ticket_.store(((ticket_.load() >> 32) + 1) << 32 | 0x7FFFFFFFull);
        { std::lock_guard<std::mutex> lk(m_); wake_seq_++; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void loop(unsigned id)
    {
        uint64_t seen = 0;                                      // epoch of the last job this worker looked at
        int64_t poll_until = 0;
        for (;;) {
            // poll for a while, then sleep
            unsigned spins = 0;
            while ((ticket_.load(std::memory_order_acquire) >> 32) == seen) {
                cpu_relax();
                if ((++spins & yield_mask_) == 0) std::this_thread::yield();
                if ((spins & 255u) != 0) continue;
                const int64_t now = now_us();
                if (now < poll_until || now < warm_until_.load(std::memory_order_relaxed)) continue;
                std::unique_lock<std::mutex> lk(m_);
                sleepers_.fetch_add(1);
                const uint64_t ws = wake_seq_;
                cv_.wait(lk, [&] { return wake_seq_ != ws || (ticket_.load(std::memory_order_acquire) >> 32) != seen; });
                sleepers_.fetch_sub(1);
            }
            if (stop_.load()) return;
            uint64_t tk = ticket_.load(std::memory_order_acquire);
            seen = tk >> 32;
            // the job fields were written before this epoch was published and stay valid while tasks of this
            // epoch can still be claimed (the caller waits for every claimed task)
            const Job &job = jobs_[seen & 1];
            const std::function<void(size_t)> *fn = job.fn;
            const size_t n = job.n_tasks;
            if (id < job.want) {
                size_t mine = 0;
                for (;;) {
                    if ((tk >> 32) != seen || (tk & 0xFFFFFFFFull) >= n) break;
                    if (!ticket_.compare_exchange_weak(tk, tk + 1, std::memory_order_acq_rel)) continue;   // tk reloaded
                    (*fn)((size_t)(tk & 0xFFFFFFFFull));
                    mine++;
                    tk = ticket_.load(std::memory_order_acquire);
                }
                if (mine) done_.fetch_add(mine, std::memory_order_acq_rel);
            }
            poll_until = now_us() + poll_us_;
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_, job_mutex_;
    std::condition_variable cv_;
    struct Job { const std::function<void(size_t)> *fn = nullptr; size_t n_tasks = 0; unsigned want = 0; } jobs_[2];
    std::atomic<uint64_t> ticket_{0};                          // epoch << 32 | next task index
    std::atomic<size_t> done_{0};
    std::atomic<int> sleepers_{0};
    std::atomic<int64_t> warm_until_{0};
    std::atomic<bool> stop_{false};
    uint64_t wake_seq_ = 0;                                    // guarded by m_
    int64_t poll_us_ = 2500;
    unsigned yield_mask_ = 31u;
};

template <typename F> void parallel_tasks(size_t n_tasks, unsigned max_threads, F fn)
{
    HostPool::get().run(n_tasks, max_threads, std::function<void(size_t)>(fn));
}

// addReadHolder's token assignment (libcrispr.cpp:1137-1143) for all candidates: token = 2 + rank of
// the string's FIRST occurrence.  Deterministic parallel form: candidates are partitioned by hash
// (all occurrences of a string land in one partition), each partition is scanned in candidate order
// by one thread to find every candidate's first occurrence, then first occurrences are ranked.
void assign_tokens(MergeResult &m, const char *dr_chars, const uint16_t *dr_len, uint32_t dr_stride, uint64_t n)
{
    m.cand_token.resize(n);
    if (n < 8192) {
        for (uint64_t k = 0; k < n; k++) {
            const char *dr = dr_chars + k * (uint64_t)dr_stride;
            uint32_t t = m.tokens.get(dr, dr_len[k]);
            if (t == 0) t = m.tokens.add(dr, dr_len[k]);
            m.cand_token[k] = t;
        }
        return;
    }
    constexpr unsigned P = 64;                                   // partitions
    std::vector<uint64_t> hash(n);
    const size_t chunk = 4096;
    parallel_tasks((n + chunk - 1) / chunk, 16, [&](size_t c) {
        const uint64_t lo = c * chunk, hi = std::min<uint64_t>(n, lo + chunk);
        for (uint64_t k = lo; k < hi; k++) hash[k] = TokenTable::hash(dr_chars + k * (uint64_t)dr_stride, dr_len[k]);
    });
    // stable counting sort of candidate indices by partition
    std::vector<uint32_t> start(P + 1, 0), order(n);
    for (uint64_t k = 0; k < n; k++) start[(hash[k] >> 58) + 1]++;
    for (unsigned p = 0; p < P; p++) start[p + 1] += start[p];
    {
        std::vector<uint32_t> fill(start.begin(), start.end() - 1);
        for (uint64_t k = 0; k < n; k++) order[fill[hash[k] >> 58]++] = (uint32_t)k;
    }
    std::vector<uint32_t> rep(n);                                // first occurrence (candidate index) of k's string
    parallel_tasks(P, 16, [&](size_t p) {
        const uint32_t lo = start[p], hi = start[p + 1];
        size_t cap = 64;
        while (cap < (size_t)(hi - lo) * 2) cap <<= 1;
        std::vector<uint32_t> slot(cap, 0xFFFFFFFFu);            // candidate index of a first occurrence
        for (uint32_t q = lo; q < hi; q++) {
            const uint32_t k = order[q];
            const char *dr = dr_chars + k * (uint64_t)dr_stride;
            size_t i = (size_t)(hash[k] >> 6) & (cap - 1);
            for (;; i = (i + 1) & (cap - 1)) {
                const uint32_t f = slot[i];
                if (f == 0xFFFFFFFFu) { slot[i] = k; rep[k] = k; break; }
                if (hash[f] == hash[k] && dr_len[f] == dr_len[k] && memcmp(dr_chars + f * (uint64_t)dr_stride, dr, dr_len[k]) == 0) { rep[k] = f; break; }
            }
        }
    });
    // rank first occurrences in candidate order -> tokens; register the strings
    for (uint64_t k = 0; k < n; k++)
        if (rep[k] == k) m.cand_token[k] = m.tokens.add(dr_chars + k * (uint64_t)dr_stride, dr_len[k]);
    for (uint64_t k = 0; k < n; k++) m.cand_token[k] = m.cand_token[rep[k]];
}

bool shorter_first(const std::string &a, const std::string &b) { return a.length() < b.length(); }
bool not_empty(const std::string &a) { return !a.empty(); }

inline uint64_t hash_bytes(const char *p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
    return h;
}

// 2-bit code of s[pos, pos+16) (first base least significant); false if a non-ACGT byte is inside
inline bool code16(const std::string &s, size_t pos, uint32_t &out)
{
    uint32_t v = 0;
    for (int i = 0; i < 16; i++) { int c = acgt_code((unsigned char)s[pos + i]); if (c < 0) return false; v |= (uint32_t)c << (2 * i); }
    out = v;
    return true;
}

// A member of a DR group while the non-redundant set is computed: a view of the token string.
struct Member {
    uint32_t tok;          // token - 2
    uint32_t len;
    bool blank;
};
// 2-bit packing of an ACGT-only token of <= 64 bases (base i in bits 2i..2i+1) and of its reverse complement
typedef unsigned __int128 u128;
struct PackedDr {
    u128 fwd, rev;
    bool ok;
};
bool member_shorter_first(const Member &a, const Member &b) { return a.len < b.len; }     // sortLengthAssending
bool member_kept(const Member &a) { return !a.blank; }

void pack_dr(const char *p, size_t n, PackedDr &out)
{
    static const struct Tab { int8_t t[256]; Tab() { memset(t, -1, sizeof(t)); t['A'] = 0; t['C'] = 1; t['G'] = 2; t['T'] = 3; } } tab;
    out.fwd = out.rev = 0;
    out.ok = n <= 64;
    if (!out.ok) return;
    uint64_t f[2] = {0, 0};
    u128 rev = 0;
    for (size_t i = 0; i < n; i++) {
        const int c = tab.t[(unsigned char)p[i]];
        if (c < 0) { out.ok = false; return; }
        f[i >> 5] |= (uint64_t)c << (2 * (i & 31));
        rev = (rev << 2) | (unsigned)(3 - c);
    }
    out.fwd = ((u128)f[1] << 64) | f[0];
    out.rev = rev;
}

// WorkHorse::removeRedundantRepeats (WorkHorse.cpp:612-645) with includeSubstring (:78-86).
// A string is blanked iff some strictly shorter member (or its reverse complement) occurs in it
// (equal-length members can only contain each other if identical, and tokens are distinct), so
// the O(n^2) pairwise find() is replaced by an index of every member's leading 16 bases: for
// ACGT-only groups the members are 2-bit packed, a window is one shift of a 128-bit integer, and
// the index is a small open-addressing table behind a bitmap pre-filter; byte-hash index otherwise.
// ORDER of the survivors: the reference sorts with std::sort and drops the blanks with std::partition, both of which
// leave the relative order of equivalent elements unspecified (what a given libstdc++ happens to produce is an artefact
// of its introsort; only the SET reaches pass 2, SURVEY a-14).  Every path of this library — this one, the host view of
// a device merge below, the view the device exports itself (dmerge.hip, k_dmx_*) and the oracle — uses the one
// permitted outcome that needs no sequential sort: length ascending, token order among equal lengths, blanks erased in place.
void remove_redundant(std::vector<Member> &v, const StringArena &strs, const PackedDr *packed)
{
    std::stable_sort(v.begin(), v.end(), member_shorter_first);
    if (v.size() > 1) {
        const size_t min_len = v.front().len;
        bool all_packed = min_len >= 16;
        for (size_t i = 0; i < v.size() && all_packed; i++) all_packed = packed[v[i].tok].ok;
        if (min_len == 0) {
            for (size_t j = 1; j < v.size(); j++) v[j].blank = true;    // an empty string is a substring of everything
        } else if (all_packed) {
            // Members shorter than the longest are the only possible needles.  Variants of one DR mostly share
            // their leading bases, so the needles are kept in an exact set keyed by (whole packed string,
            // length); a 65536-bit map of the needles' leading 16-mers rejects almost every window first.
            const uint32_t max_len = v.back().len;
            size_t n_needles = 0;
            while (n_needles < v.size() && v[n_needles].len < max_len) n_needles++;
            size_t cap = 64;
            while (cap < n_needles * 2 + 2) cap <<= 1;
            std::vector<int32_t> slot(cap, -1);
            std::vector<uint64_t> bitmap(1024, 0);
            std::vector<uint32_t> lens;                              // distinct needle lengths, ascending
            auto bit_of = [&](uint32_t key) { return (uint32_t)((key * 0x85EBCA6Bu) >> 16); };
            auto slot_of = [&](u128 val, uint32_t len) {
                const uint64_t h = ((uint64_t)val * 0x9E3779B97F4A7C15ull) ^ (((uint64_t)(val >> 64) + len) * 0xC2B2AE3D27D4EB4Full);
                return (size_t)((h ^ (h >> 29)) & (cap - 1));
            };
            for (uint32_t i = 0; i < n_needles; i++) {
                const u128 val = packed[v[i].tok].fwd;
                size_t sl = slot_of(val, v[i].len);
                while (slot[sl] >= 0) sl = (sl + 1) & (cap - 1);
                slot[sl] = (int32_t)i;
                const uint32_t b = bit_of((uint32_t)val);
                bitmap[b >> 6] |= 1ull << (b & 63);
                if (lens.empty() || lens.back() != v[i].len) lens.push_back(v[i].len);
            }
            for (size_t j = 0; j < v.size(); j++) {
                const uint32_t sl = v[j].len;
                if (sl == min_len) continue;                       // nothing is strictly shorter
                const PackedDr &ps = packed[v[j].tok];
                bool blank = false;
                // t or revcomp(t) occurs in s  <=>  t occurs in s or in revcomp(s)
                for (int o = 0; o < 2 && !blank; o++) {
                    u128 hay = o ? ps.rev : ps.fwd;
                    for (uint32_t p = 0; p + min_len <= sl && !blank; p++, hay >>= 2) {     // the shortest needle must still fit
                        const uint32_t b = bit_of((uint32_t)hay);
                        if (!(bitmap[b >> 6] >> (b & 63) & 1)) continue;
                        for (uint32_t tl : lens) {
                            if (tl >= sl || p + tl > sl) break;    // ascending: the rest is longer still
                            const u128 val = hay & ((((u128)1) << (2 * tl)) - 1);     // tl < sl <= 64
                            for (size_t q = slot_of(val, tl); slot[q] >= 0; q = (q + 1) & (cap - 1)) {
                                const Member &t = v[(size_t)slot[q]];
                                if (t.len == tl && packed[t.tok].fwd == val) { blank = true; break; }
                            }
                            if (blank) break;
                        }
                    }
                }
                v[j].blank = blank;
            }
        } else {
            // general strings (a byte outside ACGTN somewhere): includeSubstring(a, b) = b.find(a) || b.find(rc(a))
            // (WorkHorse.cpp:78-86) literally — BOTH needles against s.  ("a in s or in rc(s)" is only the same thing
            // when the complement table is an involution, and comp_tab is not: U -> A -> T, [96] = 64.)
            const size_t kAnchor = std::min<size_t>(16, min_len);
            std::vector<std::string> rcs(v.size());
            std::unordered_multimap<uint64_t, uint32_t> index;          // anchor hash -> (member << 1) | (1: its reverse complement)
            index.reserve(v.size() * 4);
            for (uint32_t i = 0; i < v.size(); i++) {
                rcs[i] = reverse_complement(strs[v[i].tok]);
                index.emplace(hash_bytes(strs.data(v[i].tok), kAnchor), i << 1);
                index.emplace(hash_bytes(rcs[i].data(), kAnchor), (i << 1) | 1u);
            }
            for (size_t j = 0; j < v.size(); j++) {
                const std::string s = strs[v[j].tok];
                bool blank = false;
                for (size_t p = 0; p + kAnchor <= s.size() && !blank; p++) {
                    auto range = index.equal_range(hash_bytes(s.data() + p, kAnchor));
                    for (auto it = range.first; it != range.second; ++it) {
                        const Member &t = v[it->second >> 1];
                        const char *needle = (it->second & 1u) ? rcs[it->second >> 1].data() : strs.data(t.tok);
                        if (t.len < s.size() && p + t.len <= s.size() && memcmp(s.data() + p, needle, t.len) == 0) { blank = true; break; }
                    }
                }
                v[j].blank = blank;
            }
        }
    }
    for (auto &m : v) if (m.len == 0) m.blank = true;             // empty strings never survive the partition
    v.erase(std::stable_partition(v.begin(), v.end(), member_kept), v.end());
}

// k-mer code -> smallest token index containing it, filled by several threads at once.
// One 64-bit word per slot: generation (16) | code + 1 (23) | owner (25); a slot of another generation is
// empty, so the table is reused from merge to merge without clearing.
class OwnerTable {
public:
    static constexpr uint32_t kNone = 0xFFFFFFFFu;
    void begin(size_t n_instances)
    {
        size_t cap = 1024;
        while (cap < n_instances * 2) cap <<= 1;
        if (cap > slots_.size() || ++gen_ > 0xFFFFu) {
            slots_ = std::vector<std::atomic<uint64_t>>(std::max(cap, slots_.size()));
            for (auto &x : slots_) x.store(0, std::memory_order_relaxed);
            gen_ = 1;
        }
        mask_ = slots_.size() - 1;
    }
    void insert(uint32_t code, uint32_t owner)
    {
        const uint64_t tag = ((uint64_t)gen_ << 48) | ((uint64_t)(code + 1) << 25);
        for (size_t i = slot(code);; i = (i + 1) & mask_) {
            uint64_t cur = slots_[i].load(std::memory_order_relaxed);
            for (;;) {
                if ((cur >> 48) != gen_) {                                     // empty in this generation
                    if (slots_[i].compare_exchange_weak(cur, tag | owner, std::memory_order_relaxed)) return;
                    continue;                                                  // cur reloaded
                }
                if ((cur >> 25) != (tag >> 25)) break;                         // another k-mer: next slot
                if ((uint32_t)(cur & 0x1FFFFFFu) <= owner) return;
                if (slots_[i].compare_exchange_weak(cur, tag | owner, std::memory_order_relaxed)) return;
            }
        }
    }
    uint32_t find(uint32_t code) const
    {
        const uint64_t want = (((uint64_t)gen_ << 48) | ((uint64_t)(code + 1) << 25)) >> 25;
        for (size_t i = slot(code);; i = (i + 1) & mask_) {
            const uint64_t cur = slots_[i].load(std::memory_order_relaxed);
            if ((cur >> 48) != gen_) return kNone;
            if ((cur >> 25) == want) return (uint32_t)(cur & 0x1FFFFFFu);
        }
    }
private:
    size_t slot(uint32_t code) const { return (size_t)((code * 0x9E3779B1u) >> 7) & mask_; }
    std::vector<std::atomic<uint64_t>> slots_;
    size_t mask_ = 0;
    uint64_t gen_ = 0;
};

} // namespace

// CPUs the cgroup's quota gives this process per period (cgroup v2 cpu.max, else v1 cfs_quota/cfs_period); 0 = no quota
double host_cpu_quota()
{
    static const double q = [] {
        double quota_cpus = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                           // "<quota|max> <period>"
            char qs[64] = {0}; long long per = 0;
            if (fscanf(f, "%63s %lld", qs, &per) == 2 && per > 0 && qs[0] != 'm') quota_cpus = (double)atoll(qs) / (double)per;
            fclose(f);
        } else {
            long long qq = -1, per = 0;
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &qq) != 1) qq = -1; fclose(g); }
            if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &per) != 1) per = 0; fclose(g); }
            if (qq > 0 && per > 0) quota_cpus = (double)qq / (double)per;
        }
        return quota_cpus;
    }();
    return q;
}

void host_pool_warm() { HostPool::get().warm(); }
void host_parallel_for(size_t n_tasks, unsigned max_threads, const std::function<void(size_t)> &fn) { HostPool::get().run(n_tasks, max_threads, fn); }

static double prof_now()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// token assignment from a device-computed first-occurrence map; false if the map is inconsistent
static bool assign_tokens_from_rep(MergeResult &m, const char *dr_chars, const uint16_t *dr_len, uint32_t dr_stride, uint64_t n,
                                   const uint32_t *rep, const uint64_t *hash)
{
    m.cand_token.resize(n);
    for (uint64_t k = 0; k < n; k++) {
        const uint32_t f = rep[k];
        if (f > k || rep[f] != f) return false;
        const char *dr = dr_chars + k * (uint64_t)dr_stride;
        if (f == k) {
            if (hash[k] != TokenTable::hash(dr, dr_len[k])) return false;
            // two different first occurrences must be different strings: guaranteed by the table (same
            // string => same hash => same slot); get() re-checks it exactly
            if (m.tokens.get(dr, dr_len[k]) != 0) return false;
            m.cand_token[k] = m.tokens.add_hashed(dr, dr_len[k], hash[k]);
        } else {
            if (dr_len[f] != dr_len[k] || memcmp(dr_chars + f * (uint64_t)dr_stride, dr, dr_len[k]) != 0) return false;
            m.cand_token[k] = m.cand_token[f];
        }
    }
    return true;
}

// every group contributes its survivors, then their reverse complements (WorkHorse.cpp:690-705)
static void emit_patterns(MergeResult &m, const std::vector<std::vector<Member>> &survivors)
{
    const unsigned char *ctab = comp_table();
    size_t n_pat = 0, n_pat_chars = 0;
    for (const auto &g : survivors) { n_pat += 2 * g.size(); for (const Member &x : g) n_pat_chars += 2 * (size_t)x.len; }
    m.patterns.chars.reserve(n_pat_chars); m.patterns.off.reserve(n_pat + 1);
    m.pat_group.reserve(n_pat); m.pat_token.reserve(n_pat);
    std::string rc;
    for (size_t g = 0; g < m.groups.size(); g++) {
        const std::vector<Member> &clustered = survivors[g];
        for (const Member &x : clustered) m.patterns.push(m.tokens.strings.data(x.tok), x.len);
        const size_t first_rc = m.patterns.size();
        for (const Member &x : clustered) {
            const char *p = m.tokens.strings.data(x.tok);
            rc.resize(x.len);
            for (uint32_t i = 0; i < x.len; i++) rc[i] = (char)ctab[(unsigned char)p[x.len - 1 - i] & 127];
            m.patterns.push(rc);
        }
        // token of the low-lexi form (DRLowLexi: tmp_dr < rev_comp ? tmp_dr : rev_comp) of the pattern and of
        // its reverse complement alike; a token string is normally low-lexi already
        for (size_t i = 0; i < clustered.size(); i++) {
            const Member &x = clustered[i];
            const char *p = m.tokens.strings.data(x.tok);
            const char *q = m.patterns.data(first_rc + i);
            const int cmp = memcmp(p, q, x.len);
            m.pat_token.push_back(cmp <= 0 ? x.tok + 2 : m.tokens.get(q, x.len));
        }
        for (size_t i = 0; i < clustered.size(); i++) m.pat_token.push_back(m.pat_token[m.pat_token.size() - clustered.size()]);
        m.pat_group.insert(m.pat_group.end(), 2 * clustered.size(), (uint32_t)(g + 1));
    }
}

static void cluster_and_patterns(MergeResult &m, int kmer_clust_size, double t0, double t1);

void merge_candidates(MergeResult &m, const char *dr_chars, const uint16_t *dr_len, uint32_t dr_stride,
                      uint64_t n, int kmer_clust_size, const uint32_t *rep, const uint64_t *hash)
{
    const double t0 = prof_now();
    m.clear();
    if (!(rep && hash && assign_tokens_from_rep(m, dr_chars, dr_len, dr_stride, n, rep, hash))) {
        m.clear();
        assign_tokens(m, dr_chars, dr_len, dr_stride, n);
    }
    cluster_and_patterns(m, kmer_clust_size, t0, prof_now());
}

bool merge_from_distinct(MergeResult &m, const char *dx_chars, const uint16_t *dx_len, const uint64_t *dx_hash, uint32_t dr_stride,
                         uint64_t n_distinct, const uint32_t *cand_distinct, uint64_t n, int kmer_clust_size)
{
    const double t0 = prof_now();
    m.clear();
    // the device hashed with the same function (checked here, a few threads) ...
    std::atomic<int> bad{0};
    const size_t chunk = 2048;
    parallel_tasks((n_distinct + chunk - 1) / chunk, 8, [&](size_t c) {
        const uint64_t lo = c * chunk, hi = std::min<uint64_t>(n_distinct, lo + chunk);
        for (uint64_t j = lo; j < hi; j++)
            if (dx_hash[j] != TokenTable::hash(dx_chars + j * (uint64_t)dr_stride, dx_len[j])) bad.store(1);
    });
    if (bad.load()) return false;
    size_t n_chars = 0;
    for (uint64_t j = 0; j < n_distinct; j++) n_chars += dx_len[j];
    m.tokens.reserve(n_distinct, n_chars);
    // ... and the insertion itself proves the strings are pairwise distinct
    for (uint64_t j = 0; j < n_distinct; j++)
        if (m.tokens.add_unique_hashed(dx_chars + j * (uint64_t)dr_stride, dx_len[j], dx_hash[j]) == 0) { m.clear(); return false; }
    m.cand_token.resize(n);
    uint32_t worst = 0;
    for (uint64_t k = 0; k < n; k++) {
        worst = std::max(worst, cand_distinct[k]);
        m.cand_token[k] = cand_distinct[k] + 2;
    }
    if (n && worst >= n_distinct) { m.clear(); return false; }
    cluster_and_patterns(m, kmer_clust_size, t0, prof_now());
    return true;
}

// Host view of a merge that ran on the device (dmerge.hip): token table, groups and pattern list are
// rebuilt from the device's per-token results (gid_of = GID, blank = dropped by removeRedundantRepeats) with
// the reference's own sort/partition predicates, so the hand-off matches the host merge field for field.
// first half: what only needs pass 1's outputs (token strings, every candidate's token) — the caller runs it while
// the merge kernels are still busy
bool merge_from_device_begin(MergeResult &m, const char *dx_chars, const uint16_t *dx_len, uint32_t dr_stride, uint64_t n_distinct,
                             const uint32_t *cand_distinct, uint64_t n)
{
    m.clear();
    // the device compared every candidate with its representative byte for byte, so the list is pairwise
    // distinct; the string -> token side is built lazily (pass 2 resolves its tokens on the device)
    m.tokens.load_strings(dx_chars, dx_len, dr_stride, n_distinct);
    m.cand_token.resize(n);
    for (uint64_t k = 0; k < n; k++) {
        if (cand_distinct[k] >= n_distinct) { m.clear(); return false; }
        m.cand_token[k] = cand_distinct[k] + 2;
    }
    return true;
}
// second half: groups and the pattern list from the device's per-token results.  Groups are independent, so everything
// per group runs on the host pool over flat, pre-sized arrays (41.6 k tokens / 13.3 k patterns at 100 M reads: 0.71 ms
// on one thread, which outlasted the device's pass 2 as soon as the reads were sharded over several GPUs).
bool merge_from_device_finish(MergeResult &m, const uint32_t *gid_of, const uint8_t *blank, uint32_t n_groups)
{
    const bool prof = getenv("CRASS_MERGE_PROFILE") != nullptr;
    const double p2 = prof_now();
    const uint32_t n_distinct = m.tokens.size();
    m.next_free_gid = (int)n_groups + 1;
    // members by group, token order inside a group (stable counting sort) = the sequence the reference sorts.  Chunks of
    // the token range count and scatter on the host pool: chunk c's members of group g go behind those of the chunks
    // before it, so the order inside a group stays the token order.
    const unsigned host_threads = 16;
    const size_t n_chunks = n_distinct >= 8192 && (uint64_t)n_groups * 8 <= n_distinct ? 8 : 1;
    const size_t per_chunk = (n_distinct + n_chunks - 1) / n_chunks;
    std::vector<uint32_t> hist(n_chunks * ((size_t)n_groups + 1), 0);
    std::atomic<int> bad_gid{0};
    parallel_tasks(n_chunks, host_threads, [&](size_t c) {
        uint32_t *h = hist.data() + c * ((size_t)n_groups + 1);
        const uint32_t lo = (uint32_t)(c * per_chunk), hi = (uint32_t)std::min<size_t>(n_distinct, (c + 1) * per_chunk);
        for (uint32_t t = lo; t < hi; t++) {
            const uint32_t g = gid_of[t];
            if (g == 0 || g > n_groups) { bad_gid.store(1, std::memory_order_relaxed); return; }
            h[g - 1]++;
        }
    });
    if (bad_gid.load()) { m.clear(); return false; }
    std::vector<uint32_t> goff(n_groups + 1, 0);
    {   // hist[c][g] -> first slot of chunk c's members of group g
        uint32_t at = 0;
        for (uint32_t g = 0; g < n_groups; g++) {
            goff[g] = at;
            for (size_t c = 0; c < n_chunks; c++) { uint32_t &h = hist[c * ((size_t)n_groups + 1) + g]; const uint32_t k = h; h = at; at += k; }
        }
        goff[n_groups] = at;
    }
    std::unique_ptr<Member[]> members(new Member[n_distinct ? n_distinct : 1]);
    m.grp_tokens.resize(n_distinct);
    m.grp_off.resize(n_groups + 1);
    for (uint32_t g = 0; g <= n_groups; g++) m.grp_off[g] = goff[g];
    parallel_tasks(n_chunks, host_threads, [&](size_t c) {
        uint32_t *cur = hist.data() + c * ((size_t)n_groups + 1);
        const uint32_t lo = (uint32_t)(c * per_chunk), hi = (uint32_t)std::min<size_t>(n_distinct, (c + 1) * per_chunk);
        for (uint32_t t = lo; t < hi; t++) {
            const uint32_t q = cur[gid_of[t] - 1]++;
            members[q] = Member{t, (uint32_t)m.tokens.strings.len(t), blank[t] != 0};
            m.grp_tokens[q] = t + 2;
        }
    });
    m.groups.assign(n_groups, {});
    const double p2b = prof_now();
    // per group: sort by length (token order among equal lengths), survivors in front in that order (remove_redundant)
    std::vector<uint32_t> kept(n_groups + 1, 0);
    std::vector<uint64_t> kept_chars(n_groups + 1, 0);
    const size_t gchunk = n_distinct >= 4096 ? 1 : std::max<size_t>(n_groups, 1);          // small sets: one task, no dispatch
    const size_t n_tasks = (n_groups + gchunk - 1) / gchunk;
    parallel_tasks(n_tasks, host_threads, [&](size_t c) {
        for (size_t g = c * gchunk; g < std::min<size_t>(n_groups, (c + 1) * gchunk); g++) {
            m.groups[g].assign(m.grp_tokens.begin() + goff[g], m.grp_tokens.begin() + goff[g + 1]);
            Member *lo = members.get() + goff[g], *hi = members.get() + goff[g + 1];
            std::stable_sort(lo, hi, member_shorter_first);
            Member *mid = std::stable_partition(lo, hi, member_kept);
            kept[g + 1] = (uint32_t)(mid - lo);
            uint64_t ch = 0;
            for (Member *x = lo; x < mid; x++) ch += x->len;
            kept_chars[g + 1] = ch;
        }
    });
    const double p3 = prof_now();
    // pattern list: per group the survivors, then their reverse complements (WorkHorse.cpp:690-697)
    for (uint32_t g = 0; g < n_groups; g++) { kept[g + 1] += kept[g]; kept_chars[g + 1] += kept_chars[g]; }
    const size_t n_pat = 2 * (size_t)kept[n_groups];
    m.patterns.chars.resize(2 * kept_chars[n_groups]);
    m.patterns.off.resize(n_pat + 1);
    m.patterns.off[0] = 0;
    m.pat_group.resize(n_pat);
    m.pat_token.resize(n_pat);
    const unsigned char *ctab = comp_table();
    std::atomic<int> need_lookup{0};
    parallel_tasks(n_tasks, host_threads, [&](size_t c) {
        for (size_t g = c * gchunk; g < std::min<size_t>(n_groups, (c + 1) * gchunk); g++) {
            const Member *lo = members.get() + goff[g];
            const uint32_t k = kept[g + 1] - kept[g];
            const size_t p0 = 2 * (size_t)kept[g];                     // first pattern of the group
            uint64_t at = 2 * kept_chars[g];
            char *out = m.patterns.chars.data();
            for (uint32_t i = 0; i < k; i++) {                          // forward forms
                memcpy(out + at, m.tokens.strings.data(lo[i].tok), lo[i].len);
                at += lo[i].len;
                m.patterns.off[p0 + i + 1] = at;
            }
            for (uint32_t i = 0; i < k; i++) {                          // reverse complements
                const char *src = m.tokens.strings.data(lo[i].tok);
                const uint32_t len = lo[i].len;
                char *q = out + at;
                for (uint32_t j = 0; j < len; j++) q[j] = (char)ctab[(unsigned char)src[len - 1 - j] & 127];
                at += len;
                m.patterns.off[p0 + k + i + 1] = at;
                // token of the low-lexi form (DRLowLexi: tmp_dr < rev_comp ? tmp_dr : rev_comp) of the pattern and of its
                // reverse complement alike; a token string is normally low-lexi already (else: looked up below)
                const int cmp = memcmp(src, q, len);
                const uint32_t tok = cmp <= 0 ? lo[i].tok + 2 : 0xFFFFFFFFu;
                if (cmp > 0) need_lookup.store(1, std::memory_order_relaxed);
                m.pat_token[p0 + i] = tok;
                m.pat_token[p0 + k + i] = tok;
            }
            for (uint32_t i = 0; i < 2 * k; i++) m.pat_group[p0 + i] = (uint32_t)(g + 1);
        }
    });
    if (need_lookup.load())                                            // (the lookup side of the token table is built lazily: one thread)
        for (size_t g = 0; g < n_groups; g++) {
            const uint32_t k = kept[g + 1] - kept[g];
            const size_t p0 = 2 * (size_t)kept[g];
            for (uint32_t i = 0; i < k; i++)
                if (m.pat_token[p0 + i] == 0xFFFFFFFFu) {
                    const uint32_t tok = m.tokens.get(m.patterns.data(p0 + k + i), m.patterns.len(p0 + k + i));
                    m.pat_token[p0 + i] = tok;
                    m.pat_token[p0 + k + i] = tok;
                }
        }
    if (prof)
        fprintf(stderr, "[crass_merge] host view: members by group %.1f us, sort + partition %.1f us, patterns %.1f us\n", 1e3 * (p2b - p2), 1e3 * (p3 - p2b),
                1e3 * (prof_now() - p3));
    return true;
}
// the same from the device merge's per-token ROOTS (first token of the token's group; the device keeps no dense group
// ids): GIDs number the roots in token order — nextFreeGID++ order, WorkHorse.cpp:1598
bool merge_from_device_finish_roots(MergeResult &m, const uint32_t *root_of, const uint8_t *blank, std::vector<uint32_t> &gid_tmp)
{
    const uint32_t n = m.tokens.size();
    gid_tmp.resize(n);
    uint32_t n_groups = 0;
    for (uint32_t t = 0; t < n; t++) {
        const uint32_t r = root_of[t];
        if (r > t || (r < t && root_of[r] != r)) { m.clear(); return false; }
        gid_tmp[t] = (r == t) ? ++n_groups : gid_tmp[r];
    }
    return merge_from_device_finish(m, gid_tmp.data(), blank, n_groups);
}
bool merge_from_device(MergeResult &m, const char *dx_chars, const uint16_t *dx_len, uint32_t dr_stride, uint64_t n_distinct,
                       const uint32_t *cand_distinct, uint64_t n, const uint32_t *gid_of, const uint8_t *blank, uint32_t n_groups)
{
    return merge_from_device_begin(m, dx_chars, dx_len, dr_stride, n_distinct, cand_distinct, n) &&
           merge_from_device_finish(m, gid_of, blank, n_groups);
}

static void cluster_and_patterns(MergeResult &m, int kmer_clust_size, double t0, double t1)
{
    const bool prof = getenv("CRASS_MERGE_PROFILE") != nullptr;
    // createNonRedundantSet: cluster every token in ascending token order (std::map iteration).
    // The k-mer codes are order-independent and computed up front on several threads.
    const uint32_t ntok = m.tokens.size();
    std::vector<uint32_t> code_off(ntok + 1, 0);
    for (uint32_t t = 0; t < ntok; t++) {
        const int nm = (int)m.tokens.strings.len(t) - kClusterKmer + 1;
        code_off[t + 1] = code_off[t] + (uint32_t)(nm > 0 ? nm : 0);
    }
    std::vector<int32_t> codes(code_off[ntok]);
    parallel_tasks((ntok + 255) / 256, 8, [&](size_t c) {
        for (uint32_t t = (uint32_t)c * 256; t < std::min<uint32_t>(ntok, ((uint32_t)c + 1) * 256); t++)
            if (code_off[t + 1] > code_off[t]) kmer_codes(m.tokens.strings.data(t), m.tokens.strings.len(t), codes.data() + code_off[t]);
    });
    // WorkHorse::clusterDRReads keeps a map k-mer -> GID that every token extends with its "homeless" k-mers
    // after it has been placed.  A k-mer is therefore in the map exactly when an earlier token contains it, and
    // its value is the GID of the FIRST token containing it (its owner).  The owners do not depend on the
    // greedy order, so they are found on several threads; the order-dependent pass that is left only follows
    // owner -> GID links and stops as soon as the token's group is decided.
    const size_t n_inst = codes.size();
    std::vector<uint32_t> owner_of(n_inst);
    bool any_string_kmer = false;
    if (ntok >= (1u << 25)) { fprintf(stderr, "crass merge: too many DR variants\n"); abort(); }
    {
        static thread_local OwnerTable owners_tls;               // reused from merge to merge by the calling thread
        OwnerTable &owners = owners_tls;                         // (the workers must see THIS thread's table)
        owners.begin(n_inst);
        const size_t tchunk = 256;
        parallel_tasks((ntok + tchunk - 1) / tchunk, 8, [&](size_t c) {
            for (uint32_t t = (uint32_t)(c * tchunk); t < std::min<uint32_t>(ntok, (uint32_t)((c + 1) * tchunk)); t++)
                for (uint32_t q = code_off[t]; q < code_off[t + 1]; q++)
                    if (codes[q] >= 0) owners.insert((uint32_t)codes[q], t);
        });
        std::atomic<int> neg{0};
        const size_t qchunk = 8192;
        parallel_tasks((n_inst + qchunk - 1) / qchunk, 8, [&](size_t c) {
            bool n = false;
            for (size_t q = c * qchunk; q < std::min(n_inst, (c + 1) * qchunk); q++) {
                if (codes[q] >= 0) owner_of[q] = owners.find((uint32_t)codes[q]);
                else { owner_of[q] = OwnerTable::kNone; n = true; }
            }
            if (n) neg.store(1, std::memory_order_relaxed);
        });
        any_string_kmer = neg.load() != 0;
    }
    std::unordered_map<std::string, int> other;                  // k-mers holding a non-ACGT byte: string keys, serial
    int next_gid = 1;
    std::vector<int> gid_of(ntok);
    std::vector<std::pair<int, int>> group_count;                // std::map<int,int> in the reference; tiny
    std::vector<std::string> homeless_str;
    for (uint32_t t = 0; t < ntok; t++) {
        group_count.clear();
        int group = 0;
        auto seen = [&](int gid) {
            if (group != 0) return;
            for (auto &gc : group_count)
                if (gc.first == gid) { if (kmer_clust_size <= ++gc.second) group = gid; return; }
            group_count.emplace_back(gid, 1);                    // the first sighting is not tested (:1577-1580)
        };
        bool strings_here = false;
        if (any_string_kmer)
            for (uint32_t q = code_off[t]; q < code_off[t + 1]; q++) if (codes[q] < 0) { strings_here = true; break; }
        if (!strings_here) {
            for (uint32_t q = code_off[t]; q < code_off[t + 1] && group == 0; q++) {
                const uint32_t o = owner_of[q];
                if (o != t) seen(gid_of[o]);                     // o < t: the k-mer was in the map already
            }
        } else {
            homeless_str.clear();
            const char *dr_p = m.tokens.strings.data(t);
            for (uint32_t q = code_off[t]; q < code_off[t + 1]; q++) {
                if (codes[q] >= 0) {
                    const uint32_t o = owner_of[q];
                    if (o != t) seen(gid_of[o]);
                } else {
                    std::string km(dr_p + (q - code_off[t]), kClusterKmer), rc = reverse_complement(km);
                    const std::string &lau = (km < rc) ? km : rc;      // laurenize (SeqUtils.cpp:89-97)
                    auto it = other.find(lau);
                    if (it == other.end()) homeless_str.push_back(lau); else seen(it->second);
                }
            }
        }
        if (group == 0) group = next_gid++;
        if (strings_here) for (const auto &k : homeless_str) other[k] = group;
        gid_of[t] = group;
    }
    m.next_free_gid = next_gid;
    m.groups.assign((size_t)(next_gid - 1), {});
    for (uint32_t t = 0; t < m.tokens.size(); t++) m.groups[(size_t)gid_of[t] - 1].push_back(t + 2);
    const double t2 = prof_now();
    // per-group substring de-duplication is independent across groups: a few host threads
    std::vector<PackedDr> packed(ntok);
    parallel_tasks((ntok + 511) / 512, 8, [&](size_t c) {
        for (uint32_t t = (uint32_t)c * 512; t < std::min<uint32_t>(ntok, ((uint32_t)c + 1) * 512); t++)
            pack_dr(m.tokens.strings.data(t), m.tokens.strings.len(t), packed[t]);
    });
    const double t2a = prof_now();
    std::vector<std::vector<Member>> survivors(m.groups.size());
    {
        const size_t ng = m.groups.size();
        parallel_tasks(ng, m.tokens.size() < 2000 ? 1 : 16, [&](size_t g) {
            std::vector<Member> &clustered = survivors[g];
            clustered.reserve(m.groups[g].size());
            for (uint32_t tok : m.groups[g]) clustered.push_back(Member{tok - 2, (uint32_t)m.tokens.strings.len(tok - 2), false});
            remove_redundant(clustered, m.tokens.strings, packed.data());
        });
    }
    const double t2b = prof_now();
    if (prof) fprintf(stderr, "[crass_merge]   pack %.3f ms, remove_redundant %.3f ms\n", t2a - t2, t2b - t2a);
    emit_patterns(m, survivors);
    const double t3 = prof_now();
    m.flatten();
    if (prof)
        fprintf(stderr, "[crass_merge] tokens %.3f ms, cluster %.3f ms, non-redundant %.3f ms, flatten %.3f ms\n",
                t1 - t0, t2 - t1, t3 - t2, prof_now() - t3);
}

void build_automaton(HostAutomaton &a, const StringArena &patterns)
{
    memset(a.sym, 0, sizeof(a.sym));
    uint32_t nsym = 0;
    for (char ch : patterns.chars) { const unsigned char c = (unsigned char)ch; if (!a.sym[c]) a.sym[c] = (uint8_t)++nsym; }
    const size_t total = patterns.chars.size() + 1;
    const size_t np = patterns.size();
    a.max_pat_len = 0;
    for (size_t i = 0; i < np; i++) if (patterns.len(i) > a.max_pat_len) a.max_pat_len = (uint32_t)patterns.len(i);
    const uint32_t S = nsym + 1;
    a.n_sym1 = S;
    const uint32_t NONE = 0xFFFFFFFFu;
    std::vector<uint32_t> &go = a.go;                 // built in place: [state][symbol], trimmed to the used states below
    go.assign(total * S, NONE);
    std::vector<uint16_t> &term = a.out_len;
    std::vector<uint32_t> &term_pid = a.out_pid;
    term.assign(total, 0);
    term_pid.assign(total, 0);
    uint32_t ns = 1;
    for (size_t pid = 0; pid < np; pid++) {
        const unsigned char *p = (const unsigned char *)patterns.data(pid);
        const size_t n = patterns.len(pid);
        uint32_t s = 0;
        for (size_t i = 0; i < n; i++) {
            uint32_t &g = go[(size_t)s * S + a.sym[p[i]]];
            if (g == NONE) g = ns++;
            s = g;
        }
        if (n > term[s]) { term[s] = (uint16_t)n; term_pid[s] = (uint32_t)pid; }
    }
    std::vector<uint32_t> fail(ns, 0), queue;
    queue.reserve(ns);
    go[0] = 0;
    for (uint32_t c = 1; c < S; c++) {
        const uint32_t t = go[c];
        if (t == NONE) go[c] = 0; else queue.push_back(t);
    }
    for (size_t qh = 0; qh < queue.size(); qh++) {
        const uint32_t s = queue[qh];
        const uint32_t f = fail[s];
        if (!term[s]) { term[s] = term[f]; term_pid[s] = term_pid[f]; }   // longest pattern that is a suffix of this state
        uint32_t *gs = &go[(size_t)s * S];
        const uint32_t *gf = &go[(size_t)f * S];
        gs[0] = 0;                                    // byte in no pattern: back to ROOT (acism.c:35-40)
        for (uint32_t c = 1; c < S; c++) {
            const uint32_t t = gs[c];
            if (t == NONE) gs[c] = gf[c];
            else { fail[t] = gf[c]; queue.push_back(t); }
        }
    }
    a.n_states = ns;
    go.resize((size_t)ns * S);
    term.resize(ns);
    term_pid.resize(ns);
    a.go4.clear();
    a.go4w.clear();
    const uint32_t sa = a.sym[(unsigned char)'A'], sc = a.sym[(unsigned char)'C'], sg = a.sym[(unsigned char)'G'], st = a.sym[(unsigned char)'T'];
    if (ns <= 65535) {
        a.go4.resize((size_t)ns * 4);
        for (uint32_t s = 0; s < ns; s++) {
            const uint32_t *gs = &go[(size_t)s * S];
            uint16_t *o = &a.go4[(size_t)s * 4];
            o[0] = (uint16_t)gs[sa]; o[1] = (uint16_t)gs[sc]; o[2] = (uint16_t)gs[sg]; o[3] = (uint16_t)gs[st];
        }
    } else {
        a.go4w.resize((size_t)ns * 4);
        for (uint32_t s = 0; s < ns; s++) {
            const uint32_t *gs = &go[(size_t)s * S];
            uint32_t *o = &a.go4w[(size_t)s * 4];
            o[0] = gs[sa]; o[1] = gs[sc]; o[2] = gs[sg]; o[3] = gs[st];
        }
    }
}

void build_automaton_and_anchors(HostAutomaton &a, HostAnchors &k, const StringArena &patterns)
{
    parallel_tasks(2, 2, [&](size_t t) { if (t == 0) build_automaton(a, patterns); else build_anchors(k, patterns); });
}

void build_anchors(HostAnchors &k, const StringArena &patterns)
{
    k = HostAnchors();
    std::vector<uint32_t> keys;
    keys.reserve(patterns.size() * 8);
    for (size_t pi = 0; pi < patterns.size(); pi++) {
        const unsigned char *p = (const unsigned char *)patterns.data(pi);
        const size_t n = patterns.len(pi);
        // a non-ACGT byte anywhere makes the pattern unmatchable in a packed read: it contributes no keys
        bool acgt = true;
        for (size_t i = 0; i < n; i++) if (acgt_code(p[i]) < 0) { acgt = false; break; }
        if (!acgt) continue;
        if (n < 23) return;                          // the anchor argument needs |P| >= 16 + 7
        // rolling 2-bit code of p[r, r+16) for r = 0..7
        uint32_t v = 0;
        for (int i = 0; i < 16; i++) v |= (uint32_t)acgt_code(p[i]) << (2 * i);
        keys.push_back(v);
        for (int r = 1; r < 8; r++) {
            v = (v >> 2) | ((uint32_t)acgt_code(p[(size_t)r + 15]) << 30);
            keys.push_back(v);
        }
    }
    {   // de-duplicate (variants of one DR share most 16-mers) with a throw-away open-addressing set
        size_t cap = 1024;
        while (cap < keys.size() * 2) cap <<= 1;
        std::vector<uint32_t> slot(cap, 0);
        std::vector<uint8_t> full(cap, 0);
        size_t w = 0;
        for (uint32_t key : keys) {
            size_t i = (size_t)((key * 0x9E3779B1u) >> 8) & (cap - 1);
            while (full[i] && slot[i] != key) i = (i + 1) & (cap - 1);
            if (!full[i]) { full[i] = 1; slot[i] = key; keys[w++] = key; }
        }
        keys.resize(w);
    }
    k.n_keys = (uint32_t)keys.size();
    if (keys.empty()) { k.ok = false; return; }
    // h_i(V) = ak_hash(V, m_i) >> (32 - log_size) (merge.h): 24-bit multiplies are full rate on the device; a few
    // multiplier pairs are tried until the cuckoo insertion succeeds.
    static const uint32_t MU[][2] = {{0x9E3779u, 0x85EBCBu}, {0xC2B2AFu, 0x27D4EBu}, {0x7FEB35u, 0x846CA7u}, {0xB5297Bu, 0x68E31Du},
                                     {0xD6E8FFu, 0x3C6EF3u}, {0xA24BAFu, 0x5BD1E9u}, {0xE7037Fu, 0x4CF5ADu}, {0x93D765u, 0x2127A5u}};
    constexpr uint32_t N_MU = sizeof(MU) / sizeof(MU[0]);
    // Tier 0: exact keys, one per slot, up to 2^15 slots (128 KB of LDS).
    // Tier 1: key sets beyond that (large shards / many ranks: one error variant per pattern adds up)
    //         keep the table in LDS as buckets of two 16-bit fingerprints: a superset filter with a
    //         false-positive rate of 4 * 2^-16 per probe (the flagged reads are verified exactly anyway).
    // Tier 2: exact keys again, probed in global memory (L2), for anything larger.
    auto try_exact = [&](uint32_t log_size) -> bool {
        const uint32_t size = 1u << log_size, rsh = 32 - log_size;
        for (uint32_t b = 0; b < N_MU; b++) {
            const uint32_t m1 = MU[b][0], m2 = MU[b][1];
            auto h1 = [&](uint32_t v) { return ak_hash(v, m1) >> rsh; };
            auto h2 = [&](uint32_t v) { return ak_hash(v, m2) >> rsh; };
            std::vector<uint32_t> tab(size, 0);
            std::vector<char> used(size, 0);
            bool ok = true;
            for (uint32_t key : keys) {
                uint32_t cur = key;
                uint32_t pos = h1(cur);
                int kicks = 0;
                for (;;) {
                    if (!used[pos]) { tab[pos] = cur; used[pos] = 1; break; }
                    std::swap(cur, tab[pos]);                     // evict
                    const uint32_t p1 = h1(cur), p2 = h2(cur);
                    pos = (pos == p1) ? p2 : p1;
                    if (++kicks > 500) { ok = false; break; }
                }
                if (!ok) break;
            }
            if (!ok) continue;
            for (uint32_t i = 0; i < size; i++) if (!used[i]) tab[i] = keys[0];   // unused slots hold a member key
            k.ok = true; k.mode = 0; k.log_size = log_size; k.m1 = m1; k.m2 = m2;
            k.table.swap(tab);
            return true;
        }
        return false;
    };
    auto try_buckets = [&](uint32_t log_size) -> bool {
        const uint32_t nb = 1u << log_size, rsh = 32 - log_size;
        for (uint32_t b = 0; b < N_MU; b++) {
            const uint32_t m1 = MU[b][0], m2 = MU[b][1];
            auto h1 = [&](uint32_t v) { return ak_hash(v, m1) >> rsh; };
            auto h2 = [&](uint32_t v) { return ak_hash(v, m2) >> rsh; };
            std::vector<uint32_t> slot_key((size_t)nb * 2, 0);
            std::vector<uint8_t> fill(nb, 0);                      // occupied slots of the bucket (0..2)
            uint32_t rnd = 0x2545F491u;
            bool ok = true;
            for (uint32_t key : keys) {
                uint32_t cur = key, avoid = 0xFFFFFFFFu;
                int kicks = 0;
                for (;;) {
                    const uint32_t b1 = h1(cur), b2 = h2(cur);
                    if (fill[b1] < 2) { slot_key[(size_t)b1 * 2 + fill[b1]++] = cur; break; }
                    if (fill[b2] < 2) { slot_key[(size_t)b2 * 2 + fill[b2]++] = cur; break; }
                    if (++kicks > 2000) { ok = false; break; }
                    rnd ^= rnd << 13; rnd ^= rnd >> 17; rnd ^= rnd << 5;
                    uint32_t bk = (rnd & 1) ? b1 : b2;
                    if (bk == avoid && b1 != b2) bk = (bk == b1) ? b2 : b1;      // do not bounce straight back
                    const size_t sl = (size_t)bk * 2 + ((rnd >> 1) & 1);
                    std::swap(cur, slot_key[sl]);
                    avoid = bk;
                }
                if (!ok) break;
            }
            if (!ok) continue;
            // fingerprint = high 16 bits of (h1 ^ h2) (kernels.hip, anchor_probe); empty slots keep 0 (any value
            // there is just one more false-positive source of the same 2^-16 weight)
            std::vector<uint32_t> tab(nb, 0);
            for (uint32_t bk = 0; bk < nb; bk++) {
                uint32_t w = 0;
                for (uint32_t j = 0; j < fill[bk]; j++) {
                    const uint32_t v = slot_key[(size_t)bk * 2 + j];
                    w |= ((ak_hash(v, m1) ^ ak_hash(v, m2)) >> 16) << (16 * j);
                }
                if (fill[bk] == 1) w |= w << 16;
                tab[bk] = w;
            }
            k.ok = true; k.mode = 1; k.log_size = log_size; k.m1 = m1; k.m2 = m2;
            k.table.swap(tab);
            return true;
        }
        return false;
    };
    const size_t nk = keys.size();
    for (uint32_t log_size = 10; log_size <= 15; log_size++) {
        const size_t size = (size_t)1 << log_size;
        if (nk * 3 > size && log_size != 15) continue;            // load <= 1/3: the first hash pair nearly always works
        if (nk * 2 > size) continue;                              // (<= 1/2 at the LDS limit)
        if (try_exact(log_size)) return;
    }
    for (uint32_t log_size = 14; log_size <= 15; log_size++) {
        const size_t slots = (size_t)2 << log_size;
        if (nk * 10 > slots * (log_size == 15 ? 8 : 6)) continue; // load <= 0.6 (<= 0.8 at the LDS limit)
        if (try_buckets(log_size)) return;
    }
    for (uint32_t log_size = 16; log_size <= 22; log_size++) {
        const size_t size = (size_t)1 << log_size;
        if (nk * 3 > size && log_size != 22) continue;
        if (nk * 2 > size) continue;
        if (try_exact(log_size)) return;
    }
}

} // namespace crass

// ---- context-free C entry points (include/crass_hip.h) ----
struct crass_merge_handle { crass::MergeResult m; };

extern "C" {

int crass_merge_create(const char *dr_chars, const uint16_t *dr_len, uint32_t dr_stride, uint64_t n,
                       int32_t kmer_clust_size, crass_merge_handle **out)
{
    if (!out || (n && (!dr_chars || !dr_len || !dr_stride))) return CRASS_ERR_INVALID_ARG;
    crass_merge_handle *h = new crass_merge_handle();
    crass::merge_candidates(h->m, dr_chars, dr_len, dr_stride, n, kmer_clust_size);
    *out = h;
    return CRASS_OK;
}

int crass_merge_rebuild(const char *dx_chars, const uint16_t *dx_len, uint32_t dr_stride, uint64_t n_distinct,
                        const uint32_t *cand_distinct, uint64_t n, const uint32_t *gid_of, const uint8_t *dropped,
                        uint32_t n_groups, crass_merge_handle **out)
{
    if (!out || (n_distinct && (!dx_chars || !dx_len || !dr_stride || !gid_of || !dropped)) || (n && !cand_distinct)) return CRASS_ERR_INVALID_ARG;
    crass_merge_handle *h = new crass_merge_handle();
    if (!crass::merge_from_device(h->m, dx_chars, dx_len, dr_stride, n_distinct, cand_distinct, n, gid_of, dropped, n_groups)) {
        delete h;
        return CRASS_ERR_INVALID_ARG;
    }
    *out = h;
    return CRASS_OK;
}

int crass_merge_get(const crass_merge_handle *h, crass_merge_view *o)
{
    if (!h || !o) return CRASS_ERR_INVALID_ARG;
    const crass::MergeResult &m = h->m;
    o->n_tokens = m.tokens.size(); o->tok_chars = m.tokens.strings.chars.data(); o->tok_off = m.tokens.strings.off.data();
    o->n_candidates = m.cand_token.size(); o->cand_token = m.cand_token.data();
    o->n_groups = (uint32_t)m.groups.size(); o->grp_tokens = m.grp_tokens.data(); o->grp_off = m.grp_off.data();
    o->n_patterns = (uint32_t)m.patterns.size(); o->pat_chars = m.patterns.chars.data(); o->pat_off = m.patterns.off.data();
    o->pat_group = m.pat_group.data(); o->next_free_gid = m.next_free_gid;
    return CRASS_OK;
}

void crass_merge_destroy(crass_merge_handle *h) { delete h; }

} // extern "C"
