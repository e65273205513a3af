// merge.cpp — the serial "merge" step between the two device passes (stays on the host:
// a few thousand DR variants, negligible next to the scans).
#include "merge.h"
#include "../../include/crass_hip.h"

#include <algorithm>
#include <cstring>
#include <map>

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

void MergeResult::clear()
{
    tokens.clear(); cand_token.clear(); groups.clear(); patterns.clear(); pat_group.clear();
    next_free_gid = 1;
    tok_chars.clear(); tok_off.clear(); grp_tokens.clear(); grp_off.clear(); pat_chars.clear(); pat_off.clear();
}

void MergeResult::flatten()
{
    tok_chars.clear(); tok_off.assign(1, 0);
    for (const auto &s : tokens.strings) { tok_chars.insert(tok_chars.end(), s.begin(), s.end()); tok_off.push_back(tok_chars.size()); }
    grp_tokens.clear(); grp_off.assign(1, 0);
    for (const auto &g : groups) { grp_tokens.insert(grp_tokens.end(), g.begin(), g.end()); grp_off.push_back(grp_tokens.size()); }
    pat_chars.clear(); pat_off.assign(1, 0);
    for (const auto &s : patterns) { pat_chars.insert(pat_chars.end(), s.begin(), s.end()); pat_off.push_back(pat_chars.size()); }
}

namespace {

constexpr int kClusterKmer = 11;     // CRASS_DEF_KMER_SIZE (crassDefines.h:66)

// WorkHorse::clusterDRReads (WorkHorse.cpp:1404-1637): greedy, order-dependent assignment of
// one DR variant to a group through shared laurenized 11-mers.  Returns the GID.
int cluster_one(const std::string &dr, int &next_free_gid, std::unordered_map<std::string, int> &kmer_gid,
                int min_shared)
{
    const int n_mers = (int)dr.size() - kClusterKmer + 1;
    std::vector<std::string> homeless;
    std::map<int, int> group_count;
    int group = 0;
    for (int i = 0; i < n_mers; ++i) {
        std::string km = dr.substr((size_t)i, kClusterKmer);
        std::string rc = reverse_complement(km);
        const std::string &lau = (km < rc) ? km : rc;              // laurenize (SeqUtils.cpp:89-97)
        auto it = kmer_gid.find(lau);
        if (it == kmer_gid.end()) {
            homeless.push_back(lau);
        } else if (group == 0) {
            auto gc = group_count.find(it->second);
            if (gc == group_count.end()) group_count[it->second] = 1;    // the first sighting is not tested (:1577-1580)
            else if (min_shared <= ++gc->second) group = it->second;
        }
    }
    if (group == 0) group = next_free_gid++;
    for (const auto &k : homeless) kmer_gid[k] = group;
    return group;
}

bool shorter_first(const std::string &a, const std::string &b) { return a.length() < b.length(); }
bool not_empty(const std::string &a) { return !a.empty(); }

// WorkHorse::removeRedundantRepeats (WorkHorse.cpp:612-645) with includeSubstring (:78-86).
// Uses the same std::sort / std::partition calls so that the surviving ORDER matches the
// reference built against the same libstdc++ (only the set matters for pass 2).
void remove_redundant(std::vector<std::string> &v)
{
    std::sort(v.begin(), v.end(), shorter_first);
    for (size_t i = 0; i < v.size(); i++) {
        if (v[i].empty()) continue;
        const std::string rc = reverse_complement(v[i]);
        for (size_t j = i + 1; j < v.size(); j++) {
            if (v[j].empty()) continue;
            if (v[j].find(v[i]) != std::string::npos || v[j].find(rc) != std::string::npos) v[j].clear();
        }
    }
    v.erase(std::partition(v.begin(), v.end(), not_empty), v.end());
}

} // namespace

void merge_candidates(MergeResult &m, const char *dr_chars, const uint16_t *dr_len, uint32_t dr_stride,
                      uint64_t n, int kmer_clust_size)
{
    m.clear();
    m.cand_token.resize(n);
    for (uint64_t k = 0; k < n; k++) {
        std::string dr(dr_chars + k * (uint64_t)dr_stride, dr_len[k]);
        uint32_t t = m.tokens.get(dr);
        if (t == 0) t = m.tokens.add(dr);
        m.cand_token[k] = t;
    }
    // createNonRedundantSet: cluster every token in ascending token order (std::map iteration)
    std::unordered_map<std::string, int> kmer_gid;
    int next_gid = 1;
    std::vector<int> gid_of(m.tokens.size());
    for (uint32_t t = 0; t < m.tokens.size(); t++)
        gid_of[t] = cluster_one(m.tokens.strings[t], next_gid, kmer_gid, kmer_clust_size);
    m.next_free_gid = next_gid;
    m.groups.assign((size_t)(next_gid - 1), {});
    for (uint32_t t = 0; t < m.tokens.size(); t++) m.groups[(size_t)gid_of[t] - 1].push_back(t + 2);
    for (size_t g = 0; g < m.groups.size(); g++) {
        std::vector<std::string> clustered;
        for (uint32_t tok : m.groups[g]) clustered.push_back(m.tokens.strings[tok - 2]);
        remove_redundant(clustered);
        const size_t first = m.patterns.size();
        m.patterns.insert(m.patterns.end(), clustered.begin(), clustered.end());
        for (size_t i = 0; i < clustered.size(); i++) m.patterns.push_back(reverse_complement(m.patterns[first + i]));
        m.pat_group.insert(m.pat_group.end(), 2 * clustered.size(), (uint32_t)(g + 1));
    }
    m.flatten();
}

void build_automaton(HostAutomaton &a, const std::vector<std::string> &patterns)
{
    memset(a.sym, 0, sizeof(a.sym));
    uint32_t nsym = 0;
    size_t total = 1;
    a.max_pat_len = 0;
    for (const auto &p : patterns) {
        total += p.size();
        if (p.size() > a.max_pat_len) a.max_pat_len = (uint32_t)p.size();
        for (unsigned char c : p) if (!a.sym[c]) a.sym[c] = (uint8_t)++nsym;
    }
    const uint32_t S = nsym + 1;
    a.n_sym1 = S;
    std::vector<int32_t> go(total * S, -1);
    std::vector<uint16_t> term(total, 0);
    uint32_t ns = 1;
    for (const auto &p : patterns) {
        uint32_t s = 0;
        for (unsigned char c : p) {
            uint32_t sy = a.sym[c];
            if (go[(size_t)s * S + sy] < 0) go[(size_t)s * S + sy] = (int32_t)ns++;
            s = (uint32_t)go[(size_t)s * S + sy];
        }
        if (p.size() > term[s]) term[s] = (uint16_t)p.size();
    }
    std::vector<uint32_t> fail(ns, 0), queue;
    queue.reserve(ns);
    go[0] = 0;
    for (uint32_t c = 1; c < S; c++) {
        int32_t t = go[c];
        if (t < 0) go[c] = 0; else queue.push_back((uint32_t)t);
    }
    for (size_t qh = 0; qh < queue.size(); qh++) {
        uint32_t s = queue[qh];
        if (!term[s]) term[s] = term[fail[s]];       // longest pattern that is a suffix of this state
        go[(size_t)s * S] = 0;                        // byte in no pattern: back to ROOT (acism.c:35-40)
        for (uint32_t c = 1; c < S; c++) {
            int32_t t = go[(size_t)s * S + c];
            if (t < 0) go[(size_t)s * S + c] = go[(size_t)fail[s] * S + c];
            else { fail[(uint32_t)t] = (uint32_t)go[(size_t)fail[s] * S + c]; queue.push_back((uint32_t)t); }
        }
    }
    a.n_states = ns;
    a.go.resize((size_t)ns * S);
    for (size_t i = 0; i < (size_t)ns * S; i++) a.go[i] = (uint32_t)go[i];
    a.out_len.assign(term.begin(), term.begin() + ns);
    a.go4.clear();
    if (ns <= 65535) {
        a.go4.resize((size_t)ns * 4);
        const char acgt[4] = {'A', 'C', 'G', 'T'};
        for (uint32_t s = 0; s < ns; s++)
            for (int c = 0; c < 4; c++)
                a.go4[(size_t)s * 4 + c] = (uint16_t)a.go[(size_t)s * S + a.sym[(unsigned char)acgt[c]]];
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

int crass_merge_get(const crass_merge_handle *h, crass_merge_view *o)
{
    if (!h || !o) return CRASS_ERR_INVALID_ARG;
    const crass::MergeResult &m = h->m;
    o->n_tokens = m.tokens.size(); o->tok_chars = m.tok_chars.data(); o->tok_off = m.tok_off.data();
    o->n_candidates = m.cand_token.size(); o->cand_token = m.cand_token.data();
    o->n_groups = (uint32_t)m.groups.size(); o->grp_tokens = m.grp_tokens.data(); o->grp_off = m.grp_off.data();
    o->n_patterns = (uint32_t)m.patterns.size(); o->pat_chars = m.pat_chars.data(); o->pat_off = m.pat_off.data();
    o->pat_group = m.pat_group.data(); o->next_free_gid = m.next_free_gid;
    return CRASS_OK;
}

void crass_merge_destroy(crass_merge_handle *h) { delete h; }

} // extern "C"
