// merge.h — host side of the boundary: token registry (StringCheck), DR-variant clustering
// and the non-redundant pattern set (WorkHorse::createNonRedundantSet), plus the pass-2
// automaton builder.  Plain C++17, no HIP.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include <unordered_map>

namespace crass {

void build_comp_table(unsigned char tab[128]);                 // SeqUtils.cpp:50-59
std::string reverse_complement(const std::string &s);          // SeqUtils.cpp:61-87

// StringCheck (StringCheck.h:52-71, StringCheck.cpp:46-81): first token is 2, discovery order.
// The string -> token side is an open-addressing table over (pointer, length) views so that the
// per-candidate lookup of the sink allocates nothing.
struct TokenTable {
    // token t -> chars[off[t-2] .. off[t-1]): one arena instead of one heap string per variant
    // (this is exactly the layout crass_merge_view exposes, so "flattening" is free)
    struct Strings {
        std::vector<char> chars;
        std::vector<uint64_t> off{0};
        size_t size() const { return off.size() - 1; }
        std::string operator[](size_t i) const { return std::string(chars.data() + off[i], (size_t)(off[i + 1] - off[i])); }
        const char *data(size_t i) const { return chars.data() + off[i]; }
        size_t len(size_t i) const { return (size_t)(off[i + 1] - off[i]); }
        void push(const char *p, size_t n) { chars.insert(chars.end(), p, p + n); off.push_back(chars.size()); }
        void clear() { chars.clear(); off.assign(1, 0); }
    } strings;
    std::vector<uint32_t> slot_token;                          // 0 = empty
    std::vector<uint64_t> slot_hash;
    static uint64_t hash(const char *p, size_t n);
    uint32_t get(const char *p, size_t n) const;
    uint32_t add(const char *p, size_t n);                     // caller checked get() == 0
    uint32_t add_hashed(const char *p, size_t n, uint64_t h);  // same, hash already known
    uint32_t get(const std::string &s) const { return get(s.data(), s.size()); }
    uint32_t add(const std::string &s) { return add(s.data(), s.size()); }
    uint32_t size() const { return (uint32_t)strings.size(); }
    void clear() { strings.clear(); slot_token.clear(); slot_hash.clear(); }
private:
    void grow();
};

struct MergeResult {
    TokenTable tokens;
    std::vector<uint32_t> cand_token;                          // token of every candidate fed in
    std::vector<std::vector<uint32_t>> groups;                 // mDR2GIDMap: groups[g] = tokens of GID g+1
    std::vector<std::string> patterns;                         // createNonRedundantSet's Vecstr
    std::vector<uint32_t> pat_group;
    int next_free_gid = 1;
    // flat copies for the C views
    std::vector<char> tok_chars; std::vector<uint64_t> tok_off;
    std::vector<uint32_t> grp_tokens; std::vector<uint64_t> grp_off;
    std::vector<char> pat_chars; std::vector<uint64_t> pat_off;
    void flatten();
    void clear();
};

// addReadHolder's token assignment (libcrispr.cpp:1137-1143) for every candidate DR in read
// order, then createNonRedundantSet (WorkHorse.cpp:648-709).
// rep/hash (optional, from the device de-duplication): rep[k] = candidate index of the first occurrence
// of candidate k's string, hash[k] = TokenTable::hash of it.  Verified with memcmp; on any
// inconsistency the plain host path is used.
void merge_candidates(MergeResult &m, const char *dr_chars, const uint16_t *dr_len, uint32_t dr_stride,
                      uint64_t n, int kmer_clust_size, const uint32_t *rep = nullptr, const uint64_t *hash = nullptr);

// byte-wise Aho-Corasick with fully resolved goto; semantics of acism_create + the first
// callback of acism_scan (acism_create.c:71-392, acism.c:25-106)
struct HostAutomaton {
    uint32_t n_states = 0, n_sym1 = 1;
    uint8_t sym[256];
    std::vector<uint32_t> go;          // [n_states][n_sym1]
    std::vector<uint16_t> out_len;     // longest pattern ending at the state
    std::vector<uint32_t> out_pid;     // index (into the pattern list) of that pattern
    std::vector<uint16_t> go4;         // [n_states][4] for A,C,G,T (empty if n_states > 65535)
    uint32_t max_pat_len = 0;
};
void build_automaton(HostAutomaton &a, const std::vector<std::string> &patterns);

// pass-2 anchor keys: every 16-mer starting at offset 0..7 of an ACGT-only pattern, packed like
// the reads (base i in bits 2i..2i+1), in a two-choice cuckoo table (see kernels.hip).
struct HostAnchors {
    bool ok = false;                    // false: some pattern is shorter than 23 or the table would not fit
    uint32_t log_size = 0, s1 = 0, s2 = 0, m1 = 0, m2 = 0, n_keys = 0;
    std::vector<uint32_t> table;
};
void build_anchors(HostAnchors &k, const std::vector<std::string> &patterns);

} // namespace crass
