// merge.h — host side of the boundary: token registry (StringCheck), DR-variant clustering
// and the non-redundant pattern set (WorkHorse::createNonRedundantSet), plus the pass-2
// automaton builder.  Plain C++17, no HIP.
#pragma once
#include <stdint.h>
#include <string>
#include <functional>
#include <vector>
#include <unordered_map>

// Hash of the pass-2 anchor tables (host-built: merge.cpp; device-built: dmerge.hip; probed by kernels.hip): the low 24 bits
// of the 16-mer times a 24-bit multiplier, plus the 16-mer's top byte left where it is (bits 24..31: inside the slot index
// of every table size).  On the device that is one v_and_b32 shared by both hashes of a window and one v_mad_u32_u24 each.
// The earlier form, mul24(V ^ (V >> s_i), m_i), took three instructions per hash (the probe kernel is VALU-bound) and folds
// 32 bits into 24 before multiplying: on variant-rich key sets near load 1/2 it left hundreds of keys unplaced where this
// one places all of them with a third fewer eviction chains (profiles/NOTES_r03.md).  Adding V >> 8 instead (the top byte
// in bits 16..23) is NOT good enough: tables below 2^16 slots index with bits above those, and keys that differ only in
// their last four bases then share both slots.
#if defined(__HIPCC__)
#define CRASS_HD __host__ __device__
#else
#define CRASS_HD
#endif
CRASS_HD static inline uint32_t ak_hash(uint32_t v, uint32_t m)
{
    // (operands known to fit 24 bits: the AMDGPU back end selects v_mad_u32_u24 for this)
    return (v & 0xFFFFFFu) * (m & 0xFFFFFFu) + (v & 0xFF000000u);
}

namespace crass {

void build_comp_table(unsigned char tab[128]);                 // SeqUtils.cpp:50-59
std::string reverse_complement(const std::string &s);          // SeqUtils.cpp:61-87

// StringCheck (StringCheck.h:52-71, StringCheck.cpp:46-81): first token is 2, discovery order.
// The string -> token side is an open-addressing table over (pointer, length) views so that the
// per-candidate lookup of the sink allocates nothing.
// string i -> chars[off[i] .. off[i+1]): one arena instead of one heap string per entry
// (this is exactly the layout crass_merge_view exposes, so "flattening" is free)
struct StringArena {
    std::vector<char> chars;
    std::vector<uint64_t> off{0};
    size_t size() const { return off.size() - 1; }
    bool empty() const { return size() == 0; }
    std::string operator[](size_t i) const { return std::string(chars.data() + off[i], (size_t)(off[i + 1] - off[i])); }
    const char *data(size_t i) const { return chars.data() + off[i]; }
    size_t len(size_t i) const { return (size_t)(off[i + 1] - off[i]); }
    void push(const char *p, size_t n) { chars.insert(chars.end(), p, p + n); off.push_back(chars.size()); }
    void push(const std::string &s) { push(s.data(), s.size()); }
    void clear() { chars.clear(); off.assign(1, 0); }
};

struct TokenTable {
    typedef StringArena Strings;
    Strings strings;                                           // token t -> strings[t - 2]
    mutable std::vector<uint32_t> slot_token;                  // 0 = empty
    mutable std::vector<uint64_t> slot_hash;
    mutable bool index_stale = false;                          // strings loaded in bulk, string -> token side not built yet
    // token order given (device merge): fill the arena only; the lookup side is built on the first get()/add()
    void load_strings(const char *chars, const uint16_t *len, uint32_t stride, size_t n);
    void ensure_index() const;
    static uint64_t hash(const char *p, size_t n);
    uint32_t get(const char *p, size_t n) const;
    uint32_t add(const char *p, size_t n);                     // caller checked get() == 0
    uint32_t add_hashed(const char *p, size_t n, uint64_t h);  // same, hash already known
    uint32_t add_unique_hashed(const char *p, size_t n, uint64_t h);   // 0 if the string is already present
    void reserve(size_t n_strings, size_t n_chars);
    uint32_t get(const std::string &s) const { return get(s.data(), s.size()); }
    uint32_t add(const std::string &s) { return add(s.data(), s.size()); }
    uint32_t size() const { return (uint32_t)strings.size(); }
    void clear() { strings.clear(); slot_token.clear(); slot_hash.clear(); index_stale = false; }
private:
    void grow();
};

struct MergeResult {
    TokenTable tokens;
    std::vector<uint32_t> cand_token;                          // token of every candidate fed in
    std::vector<std::vector<uint32_t>> groups;                 // mDR2GIDMap: groups[g] = tokens of GID g+1
    StringArena patterns;                                      // createNonRedundantSet's Vecstr
    std::vector<uint32_t> pat_group;
    std::vector<uint32_t> pat_token;                           // token of each pattern's low-lexi form (0: none)
    int next_free_gid = 1;
    // flat copy of the groups for the C view (tokens and patterns are arenas already)
    std::vector<uint32_t> grp_tokens; std::vector<uint64_t> grp_off;
    void flatten();
    void clear();
};

// Wake the host worker pool ahead of a merge (its workers then poll for a few milliseconds): call it while
// waiting for the device so that the wake-up latency is hidden.
void host_pool_warm();
// CPUs the cgroup quota allows this process (0: none); thread counts of the host pool and of the ingest follow it
double host_cpu_quota();
// fn(task) for task in [0, n_tasks) on the same pool (caller included), at most max_threads threads
void host_parallel_for(size_t n_tasks, unsigned max_threads, const std::function<void(size_t)> &fn);

// addReadHolder's token assignment (libcrispr.cpp:1137-1143) for every candidate DR in read
// order, then createNonRedundantSet (WorkHorse.cpp:648-709).
// rep/hash (optional, from the device de-duplication): rep[k] = candidate index of the first occurrence
// of candidate k's string, hash[k] = TokenTable::hash of it.  Verified with memcmp; on any
// inconsistency the plain host path is used.
void merge_candidates(MergeResult &m, const char *dr_chars, const uint16_t *dr_len, uint32_t dr_stride,
                      uint64_t n, int kmer_clust_size, const uint32_t *rep = nullptr, const uint64_t *hash = nullptr);

// Same, starting from the DISTINCT strings in first-occurrence order (the device computed them, with their
// TokenTable::hash values) and every candidate's index into that list.  false (m cleared) if the list turns
// out not to be pairwise distinct or a hash does not match — the caller then uses merge_candidates().
bool merge_from_distinct(MergeResult &m, const char *dx_chars, const uint16_t *dx_len, const uint64_t *dx_hash, uint32_t dr_stride,
                         uint64_t n_distinct, const uint32_t *cand_distinct, uint64_t n, int kmer_clust_size);

// Host view of a merge computed on the device (dmerge.hip): gid_of[t] = GID of token t + 2, blank[t] != 0 when
// removeRedundantRepeats dropped it.  false (m cleared) if the inputs are inconsistent.
bool merge_from_device(MergeResult &m, const char *dx_chars, const uint16_t *dx_len, uint32_t dr_stride, uint64_t n_distinct,
                       const uint32_t *cand_distinct, uint64_t n, const uint32_t *gid_of, const uint8_t *blank, uint32_t n_groups);
// the same in two halves: begin() needs pass 1's outputs only (it can run while the merge kernels are busy),
// finish() the device's per-token results
bool merge_from_device_begin(MergeResult &m, const char *dx_chars, const uint16_t *dx_len, uint32_t dr_stride, uint64_t n_distinct,
                             const uint32_t *cand_distinct, uint64_t n);
bool merge_from_device_finish(MergeResult &m, const uint32_t *gid_of, const uint8_t *blank, uint32_t n_groups);
bool merge_from_device_finish_roots(MergeResult &m, const uint32_t *root_of, const uint8_t *blank, std::vector<uint32_t> &gid_tmp);

// byte-wise Aho-Corasick with fully resolved goto; semantics of acism_create + the first
// callback of acism_scan (acism_create.c:71-392, acism.c:25-106)
struct HostAutomaton {
    uint32_t n_states = 0, n_sym1 = 1;
    uint8_t sym[256];
    std::vector<uint32_t> go;          // [n_states][n_sym1]
    std::vector<uint16_t> out_len;     // longest pattern ending at the state
    std::vector<uint32_t> out_pid;     // index (into the pattern list) of that pattern
    std::vector<uint16_t> go4;         // [n_states][4] for A,C,G,T (empty if n_states > 65535)
    std::vector<uint32_t> go4w;        // the same, 32-bit, when n_states > 65535
    uint32_t max_pat_len = 0;
};
void build_automaton(HostAutomaton &a, const StringArena &patterns);

// pass-2 anchor keys: every 16-mer starting at offset 0..7 of an ACGT-only pattern, packed like
// the reads (base i in bits 2i..2i+1), in a two-choice cuckoo table (see kernels.hip).
struct HostAnchors {
    bool ok = false;                    // false: some pattern is shorter than 23 or the table would not fit
    uint32_t log_size = 0, m1 = 0, m2 = 0, n_keys = 0;
    uint32_t mode = 0;                  // 0: exact 32-bit keys; 1: buckets of two 16-bit fingerprints (see kernels.hip)
    std::vector<uint32_t> table;
};
void build_anchors(HostAnchors &k, const StringArena &patterns);
// both, concurrently on the host pool (they are independent)
void build_automaton_and_anchors(HostAutomaton &a, HostAnchors &k, const StringArena &patterns);

} // namespace crass
