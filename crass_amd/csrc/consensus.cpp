// consensus.cpp — host side of the stage right behind the search hot path, behind the C ABI
// crass_hip_consensus (include/crass_hip.h): WorkHorse::findConsensusDRs (src/crass/WorkHorse.cpp:578-611).
//
// What is sequential in the reference stays sequential here, on the host: groups in ascending GID order, the master DR,
// the order in which reversed slaves / split forms receive their new tokens (StringCheck::addString) and GIDs
// (nextFreeGID++), the recursion into split groups (parseGroupedDRs :1135-1379, splitGroupedDR :940-1132,
// calculateDRConsensus :801-938, combineGroupsWithIdenticalDRs :416-452).  The work per read / per DR variant runs on the
// device (consensus.hip): per group ONE ksw batch (every variant and its reverse complement against the master:
// Aligner::getOffsetAgainstMaster, Aligner.cpp:263-362; a second small batch for the variants whose two scores tie),
// ONE coverage launch (Aligner::placeReadsInCoverageArray, :364-418), and at the very end ONE batch over all reads for
// ReadHolder::updateStartStops' partial-repeat search (ReadHolder.cpp:382-511: smithWaterman + the Levenshtein filter of
// SmithWaterman.cpp:283).  updateStartStops can be deferred to the end because nothing after a group's leaf reads its
// reads again (combineGroupsWithIdenticalDRs only moves tokens).
// No search decision is taken by the host: alignments, coverage counts, DP and edit distances all come from the kernels.
#include "../../include/crass_hip.h"
#include "consensus_internal.h"
#include "engine_internal.h"
#include "merge.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <climits>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

using namespace crass;

namespace {

template <typename T> struct DBuf {
    T *p = nullptr; size_t n = 0;
    hipError_t ensure(size_t want)
    {
        if (want <= n && p) return hipSuccess;
        if (p) { crass::dev_free(p); p = nullptr; n = 0; }
        if (!want) want = 1;
        want += want / 2;
        const hipError_t e = crass::dev_alloc((void **)&p, want * sizeof(T));
        if (e == hipSuccess) n = want;
        return e;
    }
    ~DBuf() { if (p) crass::dev_free(p); }
};

// pinned host staging (the per-group round trip: one H2D and one D2H instead of four pageable ones)
template <typename T> struct HBuf {
    T *p = nullptr; size_t n = 0;
    hipError_t ensure(size_t want)
    {
        if (want <= n && p) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; n = 0; }
        if (!want) want = 1;
        want += want / 2;
        const hipError_t e = hipHostMalloc((void **)&p, want * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) n = want;
        return e;
    }
    ~HBuf() { if (p) (void)hipHostFree(p); }
};

struct Rec {
    uint64_t read = 0, roff = 0;
    int L = 0;
    std::vector<uint32_t> ss;
    uint8_t rc = 0, alive = 1;
    uint8_t host_rc = 0;                         // orientation of the record's characters in the HOST mirror (flipped lazily: rseq)
    // deferred updateStartStops
    bool upd = false, upd_rev = false; int upd_front = 0; uint32_t upd_dr = 0, upd_first = 0, upd_last = 0;
};

constexpr double kConsArrayStart = 0.5, kZoneExt = 0.55, kCollapsedCons = 0.75, kCollapsedThr = 0.30, kPartialSim = 0.85, kKmerMaxAbundance = 0.23;
constexpr int kConsArrayMul = 4, kMinReadDepth = 2, kMinPartialLen = 4;

struct CMap4 { int present[4] = {0, 0, 0, 0}; int val[4] = {0, 0, 0, 0}; int size() const { return present[0] + present[1] + present[2] + present[3]; } };
inline int c4(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
inline uint8_t nt4(char c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } }

struct Aligner {
    int length = 0;
    std::vector<int> cov; std::vector<char> cons; std::vector<float> conserv;
    std::map<int, int> off;                       // AL_Offsets
    int master = -1, master_len = 0, zone_start = 0, zone_end = 0; bool zone_set = false;
    std::vector<uint32_t> plc_rec; std::vector<int32_t> plc_pos;       // pending placements (one coverage launch per group)
    std::vector<uint32_t> flips;                                        // records reverse-complemented since the last device sync
};

} // namespace

struct crass_cons {
    crass_params prm{};
    int device = 0, hip_err = 0, error = 0;
    hipStream_t st = nullptr;
    int max_read_len = 0;
    std::vector<Rec> rec;
    std::vector<char> hseq;                       // host mirror of the records' RH_Seq; a record's characters follow the device copy
                                                  // only when somebody reads them (rseq: extendSlaveDR's ties)
    HBuf<uint32_t> h_stage; HBuf<int> h_cov; HBuf<int32_t> h_ksw; HBuf<uint8_t> h_ksw_in; size_t n_pre = 0;      // pinned: a group's flips + placements going up, its coverage coming down
    DBuf<uint32_t> d_stage;
    double t_place = 0, t_flip = 0, t_sync = 0, t_cons = 0, t_ksw = 0, t_split = 0, t_fa = 0, t_fb = 0, t_fc = 0, t_fd = 0, t_pre = 0;      // CRASS_TIMING: where the group loop's time goes
    std::vector<std::string> tok;                 // token t = tok[t - 2]
    std::vector<std::unique_ptr<std::vector<int>>> reads_of;
    std::map<int, std::unique_ptr<std::vector<int>>> group;     // mDR2GIDMap (absent / nullptr = none)
    std::map<int, std::string> true_dr;           // mTrueDRs
    int next_gid = 1;
    unsigned char comp[128];
    // device
    DBuf<uint8_t> d_seq, d_comp, d_qcodes, d_target, d_dirs, d_drchars; DBuf<uint64_t> d_roff, d_a_off, d_b_off; DBuf<uint32_t> d_rlen, d_list, d_plc_rec, d_qoff, d_qlen, d_qtgt, d_toff, d_tlen, d_droff, d_drlen, d_a_len, d_b_len;
    DBuf<int32_t> d_plc_pos, d_ksw_out, d_lev; DBuf<int> d_cov; DBuf<ConsSwTask> d_tasks; DBuf<ConsSwOut> d_swout;
    ConsKswParams ksw{};
    struct Pre { size_t first = 0, count = 0; };
    std::map<int, Pre> pre; std::vector<std::array<int, 6>> pre_res;       // the original groups' alignments (prealign_original_groups)
    crass_counters_cons cnt{};
    std::vector<char> dr_tab; std::vector<uint64_t> dr_tab_end;      // the true DRs as found (NOT laurenized), back to back: updateStartStops' DR argument
    // flattened view
    std::vector<char> o_tok_chars, o_dr_chars; std::vector<uint64_t> o_tok_off, o_dr_off, o_grp_off, o_ss_off, o_tokread_off, o_tokread_idx;
    std::vector<int32_t> o_grp_gid; std::vector<uint32_t> o_grp_tokens, o_token, o_nss, o_ss; std::vector<uint8_t> o_alive, o_rc, o_has;
};

namespace {

#define HCHK(c, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { (c)->hip_err = (int)e__; return e__ == hipErrorOutOfMemory ? CRASS_ERR_OOM : CRASS_ERR_HIP; } } while (0)

int add_string(crass_cons *s, const std::string &str)
{   // StringCheck::addString (StringCheck.cpp:46-55): always a NEW token
    s->tok.push_back(str);
    s->reads_of.emplace_back(nullptr);
    return (int)s->tok.size() + 1;
}
inline const std::string &tstr(const crass_cons *s, int tok) { return s->tok[tok - 2]; }
inline std::vector<int> *rlist(crass_cons *s, int tok) { return s->reads_of[tok - 2].get(); }
inline double now_sec() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// the record's characters in its CURRENT orientation (ReadHolder::reverseComplementSeq, :593-609, applied to the host mirror
// only now: 42 k of 100 k records are flipped in a 10 M-read job and one or two of them are ever looked at)
// one character of the record in its current orientation, whatever the host mirror's
inline char rchar(const crass_cons *s, const Rec &r, int pos)
{
    const char *q = s->hseq.data() + r.roff;
    return r.host_rc == r.rc ? q[pos] : (char)s->comp[q[r.L - 1 - pos] & 127];
}
inline char *rseq(crass_cons *s, Rec &r)
{
    char *q = s->hseq.data() + r.roff;
    if (r.host_rc != r.rc) {
        for (int i = 0, j = r.L - 1; i <= j; i++, j--) { const char a = q[i], b = q[j]; q[i] = (char)s->comp[b & 127]; q[j] = (char)s->comp[a & 127]; }
        r.host_rc = r.rc;
    }
    return q;
}

void reverse_start_stops(Rec &r)
{   // ReadHolder::reverseStartStops, ReadHolder.cpp:321-380
    if (r.ss.empty()) return;
    // The reference walks the list from the back, accumulating the gaps onto L - back - 1: entry i of the new list is
    // (L - 1 - back) + (back - ss[n-1-i]) = L - 1 - ss[n-1-i] in its 32-bit unsigned arithmetic — the list mirrored, in place
    // (no second vector: 42 k records are turned in a 10 M-read job and every one of them allocated)
    const uint32_t top = (uint32_t)(r.L - 1);
    std::reverse(r.ss.begin(), r.ss.end());
    for (auto &x : r.ss) x = top - x;
}
void flip_record(crass_cons *s, Aligner *al, int k)
{   // ReadHolder::reverseComplementSeq (:593-609): the device copy with the next sync, the host mirror when it is read (rseq)
    Rec &r = s->rec[k];
    reverse_start_stops(r);
    r.rc = !r.rc;
    if (al) al->flips.push_back((uint32_t)k);
}

// the full-length repeat search every Aligner routine starts with (Aligner.cpp:373-378,434-438,465-469): index of the first
// repeat whose stop - start == len - 1; -1 when there is none (startStopsAt(): std::out_of_range in the reference)
int first_full(const Rec &r, int len)
{
    size_t a = 0, b = 1;
    while (b < r.ss.size() && ((int)r.ss[b] - (int)r.ss[a]) != (len - 1)) { a += 2; b += 2; }
    return b < r.ss.size() ? (int)a : -1;
}

void place_reads(crass_cons *s, Aligner &al, int tok)
{   // Aligner::placeReadsInCoverageArray (Aligner.cpp:364-418): the increments themselves run on the device
    std::vector<int> *l = rlist(s, tok);
    if (!l) { s->error = 4; return; }
    const int cur_len = (int)tstr(s, tok).size(), off = al.off[tok];
    for (int k : *l) {
        const Rec &r = s->rec[k];
        int a = first_full(r, cur_len);
        if (a < 0) { s->error = 5; return; }
        int b = a + 1;
        do {
            if (((int)r.ss[b] - (int)r.ss[a]) == (cur_len - 1)) {
                const int pos = off - (int)r.ss[a];
                if (pos < 0 || pos + r.L > al.length) { s->error = 6; return; }     // "MEMORY CORRUPTION" in the reference
                al.plc_rec.push_back((uint32_t)k); al.plc_pos.push_back(pos);
            }
            a += 2; b += 2;
            if (a >= (int)(r.ss.size() / 2) * 2) break;
        } while (((int)r.ss[b] - (int)r.ss[a]) == (cur_len - 1));
    }
}

void calc_zone(crass_cons *s, Aligner &al)
{   // Aligner::calculateDRZone, Aligner.cpp:454-485
    std::vector<int> *l = rlist(s, al.master);
    if (!l) { s->error = 4; return; }
    for (int k : *l) {
        const Rec &r = s->rec[k];
        const int a = first_full(r, al.master_len);
        if (a < 0) { s->error = 5; return; }
        const int pos = al.off[al.master] - (int)r.ss[a];
        al.zone_start = pos + (int)r.ss[a]; al.zone_end = pos + (int)r.ss[a + 1]; al.zone_set = true;
        break;
    }
}

// device: ksw batch — string v against master tgt[v].  res[v] = {score_f, tb_f, qb_f, score_r, tb_r, qb_r}
// defer: the results stay in pinned memory (s->h_ksw) behind the stream's work — ksw_collect picks them up after the caller's
// next wait on the stream (the original groups' batch runs beside the host's record set-up)
int ksw_collect(crass_cons *s, size_t n, std::vector<std::array<int, 6>> &res)
{
    res.assign(n, {0, -1, -1, 0, -1, -1});
    for (size_t v = 0; v < n; v++) for (int q = 0; q < 6; q++) res[v][q] = s->h_ksw.p[v * 6 + q];
    return CRASS_OK;
}
int ksw_batch(crass_cons *s, const std::vector<std::string> &strs, const std::vector<uint32_t> &tgt, const std::vector<std::string> &masters,
              std::vector<std::array<int, 6>> &res, bool defer = false)
{
    res.assign(strs.size(), {0, -1, -1, 0, -1, -1});
    if (strs.empty()) return CRASS_OK;
    // the batch's seven small arrays go up from ONE pinned staging buffer that belongs to the batch's owner: the copies are queued,
    // not waited for (defer), and what they read must outlive this function (function-local pageable vectors only worked because
    // the runtime's copy from pageable memory happens to block the host)
    size_t n_codes = 0, n_tcodes = 0;
    uint32_t max_q = 1;
    for (const std::string &q : strs) { n_codes += q.size(); max_q = std::max(max_q, (uint32_t)q.size()); }
    for (const std::string &m : masters) n_tcodes += m.size();
    auto up4 = [](size_t x) { return (x + 3) & ~(size_t)3; };
    const size_t nq = strs.size(), nm = masters.size();
    const size_t o_codes = 0, o_off = up4(std::max<size_t>(n_codes, 1)), o_len = o_off + nq * 4, o_tgt = o_len + nq * 4, o_tcodes = o_tgt + nq * 4,
                 o_toff = o_tcodes + up4(std::max<size_t>(n_tcodes, 1)), o_tlen = o_toff + nm * 4, total = o_tlen + nm * 4;
    if (total + 8 > s->h_ksw_in.n) HCHK(s, hipStreamSynchronize(s->st));       // (growing frees the old buffer: nothing may still read it)
    HCHK(s, s->h_ksw_in.ensure(total + 8));
    uint8_t *hb = s->h_ksw_in.p;
    uint8_t *codes = hb + o_codes, *tcodes = hb + o_tcodes;
    uint32_t *off = reinterpret_cast<uint32_t *>(hb + o_off), *len = reinterpret_cast<uint32_t *>(hb + o_len), *tg = reinterpret_cast<uint32_t *>(hb + o_tgt),
             *toff = reinterpret_cast<uint32_t *>(hb + o_toff), *tlen = reinterpret_cast<uint32_t *>(hb + o_tlen);
    codes[0] = 0; tcodes[0] = 0;
    {
        size_t at = 0;
        for (size_t v = 0; v < nq; v++) { off[v] = (uint32_t)at; len[v] = (uint32_t)strs[v].size(); tg[v] = tgt[v]; for (char ch : strs[v]) codes[at++] = nt4(ch); }
        at = 0;
        for (size_t m = 0; m < nm; m++) { toff[m] = (uint32_t)at; tlen[m] = (uint32_t)masters[m].size(); for (char ch : masters[m]) tcodes[at++] = nt4(ch); }
    }
    HCHK(s, s->d_qcodes.ensure(std::max<size_t>(n_codes, 1))); HCHK(s, s->d_qoff.ensure(nq)); HCHK(s, s->d_qlen.ensure(nq)); HCHK(s, s->d_qtgt.ensure(nq));
    HCHK(s, s->d_target.ensure(std::max<size_t>(n_tcodes, 1))); HCHK(s, s->d_toff.ensure(nm)); HCHK(s, s->d_tlen.ensure(nm)); HCHK(s, s->d_ksw_out.ensure(nq * 6));
    HCHK(s, hipMemcpyAsync(s->d_qcodes.p, codes, std::max<size_t>(n_codes, 1), hipMemcpyHostToDevice, s->st));
    HCHK(s, hipMemcpyAsync(s->d_qoff.p, off, nq * 4, hipMemcpyHostToDevice, s->st));
    HCHK(s, hipMemcpyAsync(s->d_qlen.p, len, nq * 4, hipMemcpyHostToDevice, s->st));
    HCHK(s, hipMemcpyAsync(s->d_qtgt.p, tg, nq * 4, hipMemcpyHostToDevice, s->st));
    HCHK(s, hipMemcpyAsync(s->d_target.p, tcodes, std::max<size_t>(n_tcodes, 1), hipMemcpyHostToDevice, s->st));
    HCHK(s, hipMemcpyAsync(s->d_toff.p, toff, nm * 4, hipMemcpyHostToDevice, s->st));
    HCHK(s, hipMemcpyAsync(s->d_tlen.p, tlen, nm * 4, hipMemcpyHostToDevice, s->st));
    HCHK(s, launch_cons_ksw(s->d_qcodes.p, s->d_qoff.p, s->d_qlen.p, s->d_qtgt.p, (uint32_t)strs.size(), max_q, s->d_target.p, s->d_toff.p, s->d_tlen.p, s->ksw,
                            s->d_ksw_out.p, s->st));
    s->cnt.n_ksw_alignments += 2 * strs.size(); s->cnt.n_ksw_launches++;
    if (defer) {
        HCHK(s, s->h_ksw.ensure(strs.size() * 6));
        HCHK(s, hipMemcpyAsync(s->h_ksw.p, s->d_ksw_out.p, strs.size() * 6 * 4, hipMemcpyDeviceToHost, s->st));
        return CRASS_OK;
    }
    std::vector<int32_t> out(strs.size() * 6);
    HCHK(s, hipMemcpyAsync(out.data(), s->d_ksw_out.p, out.size() * 4, hipMemcpyDeviceToHost, s->st));
    HCHK(s, hipStreamSynchronize(s->st));
    for (size_t v = 0; v < strs.size(); v++) for (int q = 0; q < 6; q++) res[v][q] = out[v * 6 + q];
    return CRASS_OK;
}
int ksw_batch(crass_cons *s, const std::vector<std::string> &strs, const std::string &master, std::vector<std::array<int, 6>> &res)
{
    return ksw_batch(s, strs, std::vector<uint32_t>(strs.size(), 0u), std::vector<std::string>(1, master), res);
}

// findMasterDR (WorkHorse.cpp:711-748): the longest DR of the group, the first of equals
int find_master(const crass_cons *s, const std::vector<int> &g)
{
    int master = -1; size_t longest = 0;
    for (int tok : g) if (s->tok[tok - 2].size() > longest) { master = tok; longest = s->tok[tok - 2].size(); }
    return master;
}

// Every ORIGINAL group's slave alignments in ONE launch before the group loop starts: the master of a group depends on
// its token strings only, and no group is touched before its turn (combineGroupsWithIdenticalDRs only appends to groups
// that have been through already).  Groups that come out of a split are aligned when they are parsed.
int prealign_original_groups(crass_cons *s, int n_groups)
{
    std::vector<std::string> strs, masters; std::vector<uint32_t> tgt;
    for (int gid = 1; gid <= n_groups; gid++) {
        auto it = s->group.find(gid);
        if (it == s->group.end() || !it->second) continue;
        const int master = find_master(s, *it->second);
        if (master < 0) continue;
        crass_cons::Pre &pre = s->pre[gid];
        pre.first = strs.size();
        for (int tok : *it->second) if (tok != master) { strs.push_back(s->tok[tok - 2]); tgt.push_back((uint32_t)masters.size()); }
        pre.count = strs.size() - pre.first;
        masters.push_back(s->tok[master - 2]);
    }
    s->n_pre = strs.size();
    return ksw_batch(s, strs, tgt, masters, s->pre_res, true);      // (collected by the caller behind its next wait: ksw_collect)
}

enum { F_REVERSED = 1, F_FAILED = 2, F_EQUAL = 4 };
// the decision part of Aligner::getOffsetAgainstMaster (Aligner.cpp:303-361) on the two alignments of one string
int offset_decision(const std::array<int, 6> &a, int slen, int minsc, int &flags)
{
    const int fs = a[0], rs = a[3];
    if (rs == fs) { flags |= F_EQUAL; return 0; }
    int score, tb, qb;
    if (rs > fs) { score = rs; tb = a[4]; qb = a[5]; flags |= F_REVERSED; } else { score = fs; tb = a[1]; qb = a[2]; }
    if (slen / 2 > score) { flags |= F_FAILED; return 0; }
    if (score < minsc) { flags |= F_FAILED; return 0; }
    return tb - qb;
}

std::string extend_slave(crass_cons *s, int tok, int slave_len)
{   // Aligner::extendSlaveDR, Aligner.cpp:421-450
    std::vector<int> *l = rlist(s, tok);
    if (!l) { s->error = 4; return std::string(); }
    for (int k : *l) {
        Rec &r = s->rec[k];
        const int a = first_full(r, slave_len);
        if (a < 0) { s->error = 5; return std::string(); }
        if ((int)r.ss[a] - 2 < 0 || (int)r.ss[a + 1] + 2 > r.L) continue;
        const int pos = (int)r.ss[a] - 2;
        const int n = std::min(slave_len + 4, r.L - pos);
        return std::string(rseq(s, r) + pos, (size_t)n);
    }
    return std::string();
}

int parse_grouped_drs(crass_cons *s, int GID);

int dr_has_abundant_kmers(const std::string &dr)
{   // drHasHighlyAbundantKmers, libcrispr.cpp:1077-1117
    if (dr.size() < 3) return -1;
    std::map<std::string, int> cnt;
    int total = 0;
    for (size_t i = 0; i < dr.size() - 3; i++) { cnt[dr.substr(i, 3)]++; total++; }
    int mx = 0;
    for (auto &kv : cnt) mx = std::max(mx, kv.second);
    const float f = (float)mx / (float)total;
    return (double)f > kKmerMaxAbundance ? 1 : 0;
}
bool is_low_complexity(const std::string &rep)
{   // isRepeatLowComplexity, libcrispr.cpp:1031-1069
    int ca = 0, cc = 0, cg = 0, ct = 0, cn = 0;
    for (char ch : rep) switch (ch) { case 'a': case 'A': ca++; break; case 'c': case 'C': cc++; break; case 'g': case 'G': cg++; break; case 't': case 'T': ct++; break; default: cn++; }
    const int cut = (int)((double)(int)rep.size() * 0.75);
    return ca > cut || ct > cut || cg > cut || cc > cut || cn > cut;
}

void generate_consensus(crass_cons *s, Aligner &al)
{   // Aligner::generateConsensus, Aligner.cpp:155-240
    static const char alphabet[4] = {'A', 'C', 'G', 'T'};
    int num_gt_zero = 0;
    for (int j = 0; j < al.length; j++) {
        int max_count = 0; float total = 0.0f;
        for (int i = 0; i < 4; i++) { const int c = al.cov[(size_t)i * al.length + j]; total += (float)c; if (c > max_count) { max_count = c; al.cons[j] = alphabet[i]; } }
        if (total > kMinReadDepth) { al.conserv[j] = (float)max_count / total; num_gt_zero++; } else al.conserv[j] = 0;
    }
    if (!al.zone_set) { s->error = 7; return; }
    auto at = [&](int i, bool &ok) -> double { if (i < 0 || i >= al.length) { ok = false; return 0; } return (double)al.conserv[i]; };
    bool ok = true;
    if (num_gt_zero >= kMinReadDepth) {
        while (ok && al.zone_start > 0) { if (at(al.zone_start - 1, ok) < kZoneExt && ok) al.zone_start++; else break; }
        while (ok && al.zone_end < al.length - 1) { if (at(al.zone_end + 1, ok) < kZoneExt && ok) al.zone_end--; else break; }
    }
    while (ok && al.zone_start > 0) { if (at(al.zone_start - 1, ok) >= kZoneExt && ok) al.zone_start--; else break; }
    while (ok && al.zone_end < al.length - 1) { if (at(al.zone_end + 1, ok) >= kZoneExt && ok) al.zone_end++; else break; }
    if (!ok) s->error = 8;
}

std::string calc_dr_consensus(crass_cons *s, int GID, Aligner &al, int &collapsedPos, CMap4 &opts, std::vector<uint8_t> &refined)
{   // WorkHorse::calculateDRConsensus, WorkHorse.cpp:801-938
    generate_consensus(s, al);
    std::string true_dr;
    if (s->error) return true_dr;
    for (int i = al.zone_start; i <= al.zone_end; i++) {
        if (i < 0 || i >= al.length) { s->error = 8; break; }
        collapsedPos++;
        if ((double)al.conserv[i] >= kCollapsedCons) { refined[i] = 1; true_dr += al.cons[i]; continue; }
        refined[i] = 0;
        const float total = (float)(al.cov[i] + al.cov[(size_t)al.length + i] + al.cov[(size_t)2 * al.length + i] + al.cov[(size_t)3 * al.length + i]);
        for (int k = 0; k < 4; k++) {
            const float prop = (float)((float)al.cov[(size_t)k * al.length + i] / total);
            if ((double)prop >= kCollapsedThr) { opts.present[k] = 1; opts.val[k] = opts.size() + s->next_gid; s->next_gid++; }
        }
        if (2 > opts.size()) { opts = CMap4(); true_dr += al.cons[i]; refined[i] = 1; continue; }
        refined[i] = 0;
        CMap4 opts2;
        for (int tok : *s->group[GID]) {
            if (!al.off.count(tok)) al.off[tok] = 0;                 // AL_Offsets[tok] default-inserts
            if (-1 != al.off[tok]) {
                const int p = collapsedPos + al.zone_start, len = (int)tstr(s, tok).size();
                if (p >= al.off[tok] && p - al.off[tok] < len) {
                    const int di = c4(tstr(s, tok)[al.zone_start - al.off[tok] + collapsedPos]);
                    if (di < 0) { s->error = 9; continue; }
                    if (!opts.present[di]) { opts.present[di] = 1; opts.val[di] = 0; }
                    opts2.present[di] = 1; opts2.val[di] = opts.val[di];
                }
            }
        }
        if (2 > opts2.size()) { true_dr += al.cons[i]; refined[i] = 1; opts = CMap4(); }
        else { opts = opts2; collapsedPos += al.zone_start; i = al.zone_end + 1; }
    }
    return true_dr;
}

int read_decision_char(crass_cons *s, const Rec &r, int dec_diff, unsigned want)
{
    for (size_t k = 0; k < r.ss.size(); k += 2) {
        const int pos = (int)r.ss[k] + dec_diff;
        if (pos > 0 && pos < r.L) { const int di = c4(rchar(s, r, pos)); if (di >= 0 && (want & (1u << di))) return di; }
    }
    return -1;
}
void clear_read_list(crass_cons *s, int tok)
{
    std::vector<int> *l = rlist(s, tok);
    if (!l) return;
    for (int k : *l) if (k >= 0) s->rec[k].alive = 0;
    l->clear();
}

void split_grouped_dr(crass_cons *s, const CMap4 &opts, Aligner &al, int collapsed_pos, int GID)
{   // WorkHorse::splitGroupedDR, WorkHorse.cpp:940-1132
    int char_gid[4] = {0, 0, 0, 0};
    unsigned opt_mask = 0;
    for (int k = 0; k < 4; k++) if (opts.present[k]) {
        const int g = s->next_gid++;
        s->group[g].reset(new std::vector<int>());
        char_gid[k] = g; opt_mask |= 1u << k;
    }
    const std::vector<int> members = *s->group[GID];
    for (int tok : members) {
        if (!al.off.count(tok)) al.off[tok] = 0;
        if (-1 == al.off[tok]) continue;
        const int off = al.off[tok], tlen = (int)tstr(s, tok).size();
        if (off <= collapsed_pos && collapsed_pos < off + tlen) {
            const int di = c4(tstr(s, tok)[collapsed_pos - off]);
            if (di < 0 || !char_gid[di]) { s->error = 10; continue; }
            s->group[char_gid[di]]->push_back(tok);
            continue;
        }
        const int dec_diff = collapsed_pos - off;
        std::vector<int> *l = rlist(s, tok);
        if (!l) { s->error = 4; continue; }
        unsigned forms = 0;
        for (int k : *l) { const int di = read_decision_char(s, s->rec[k], dec_diff, opt_mask); if (di >= 0) forms |= 1u << di; }
        const int n_forms = __builtin_popcount(forms);
        if (n_forms == 1) s->group[char_gid[__builtin_ctz(forms)]]->push_back(tok);
        else if (n_forms == 0) { clear_read_list(s, tok); s->reads_of[tok - 2].reset(); }
        else {
            int form_tok[4] = {0, 0, 0, 0};
            const std::string str = tstr(s, tok);
            for (int k = 0; k < 4; k++) if (forms & (1u << k)) {
                const int st = add_string(s, str);
                s->reads_of[st - 2].reset(new std::vector<int>());
                form_tok[k] = st;
                s->group[char_gid[k]]->push_back(st);
            }
            l = rlist(s, tok);
            for (int &k : *l) {
                const int di = read_decision_char(s, s->rec[k], dec_diff, forms);
                if (di >= 0) { rlist(s, form_tok[di])->push_back(k); k = -1; }
            }
            clear_read_list(s, tok);
            s->reads_of[tok - 2].reset();
        }
    }
    s->group[GID].reset();
    for (int k = 0; k < 4; k++) if (char_gid[k]) parse_grouped_drs(s, char_gid[k]);
}

// device: bring the flipped records up to date and add the pending placements to the coverage array.  One pinned staging
// buffer goes up ({flips, placement records, placement positions}), the coverage comes down into pinned memory: per group
// two copies and one wait (it was four pageable uploads — each a staging copy inside the runtime — and a pageable download).
int sync_coverage(crass_cons *s, Aligner &al)
{
    const double t0 = now_sec();
    const size_t nf = al.flips.size(), np = al.plc_rec.size();
    HCHK(s, s->h_stage.ensure(nf + 2 * np + 4)); HCHK(s, s->d_stage.ensure(nf + 2 * np + 4));
    HCHK(s, s->h_cov.ensure((size_t)al.length * 4)); HCHK(s, s->d_cov.ensure((size_t)al.length * 4));
    uint32_t *hs = s->h_stage.p;
    if (nf) memcpy(hs, al.flips.data(), nf * 4);
    if (np) { memcpy(hs + nf, al.plc_rec.data(), np * 4); memcpy(hs + nf + np, al.plc_pos.data(), np * 4); }
    if (nf + np) HCHK(s, hipMemcpyAsync(s->d_stage.p, hs, (nf + 2 * np) * 4, hipMemcpyHostToDevice, s->st));
    if (nf) {
        HCHK(s, launch_cons_flip(s->d_seq.p, s->d_roff.p, s->d_rlen.p, s->d_stage.p, (uint32_t)nf, s->d_comp.p, s->st));
        s->cnt.n_flips += nf;
    }
    HCHK(s, hipMemsetAsync(s->d_cov.p, 0, (size_t)al.length * 16, s->st));
    if (np) {
        HCHK(s, launch_cons_cover(s->d_seq.p, s->d_roff.p, s->d_rlen.p, s->d_stage.p + nf, reinterpret_cast<const int32_t *>(s->d_stage.p + nf + np), (uint32_t)np,
                                  s->d_cov.p, al.length, s->st));
        s->cnt.n_placements += np;
    }
    HCHK(s, hipMemcpyAsync(s->h_cov.p, s->d_cov.p, (size_t)al.length * 16, hipMemcpyDeviceToHost, s->st));
    HCHK(s, hipStreamSynchronize(s->st));
    memcpy(al.cov.data(), s->h_cov.p, (size_t)al.length * 16);
    al.flips.clear(); al.plc_rec.clear(); al.plc_pos.clear();
    s->cnt.n_groups_parsed++;
    s->t_sync += now_sec() - t0;
    return CRASS_OK;
}

int parse_grouped_drs(crass_cons *s, int GID)
{   // WorkHorse::parseGroupedDRs, WorkHorse.cpp:1135-1379
    if (s->error || s->hip_err) return 0;
    std::vector<int> &g = *s->group[GID];
    const int master = find_master(s, g);
    if (master < 0) { s->error = 11; return 0; }
    Aligner al;
    al.length = kConsArrayMul * s->max_read_len;
    al.cov.assign((size_t)al.length * 4, 0); al.cons.assign((size_t)al.length, 'N'); al.conserv.assign((size_t)al.length, 0.0f);
    // Aligner::setMasterDR :73-86
    al.master = master; al.off[master] = (int)(al.length * kConsArrayStart); al.master_len = (int)tstr(s, master).size();
    double tq = now_sec();
    place_reads(s, al, master);
    calc_zone(s, al);
    s->t_place += now_sec() - tq;
    if (s->error) return 0;
    // populateCoverageArray :750-798 — first every slave's two alignments in one batch ...
    tq = now_sec();
    const std::string master_str = tstr(s, master);
    std::vector<int> slave_pos; std::vector<std::string> strs;
    for (size_t q = 0; q < g.size(); q++) if (g[q] != master) { slave_pos.push_back((int)q); strs.push_back(tstr(s, g[q])); }
    std::vector<std::array<int, 6>> res, res2;
    auto pre = s->pre.find(GID);
    if (pre != s->pre.end() && pre->second.count == strs.size()) {         // aligned up-front with every other original group
        res.assign(s->pre_res.begin() + (long)pre->second.first, s->pre_res.begin() + (long)(pre->second.first + pre->second.count));
        s->pre.erase(pre);
    } else { tq = now_sec(); const int kb = ksw_batch(s, strs, master_str, res); s->t_ksw += now_sec() - tq; if (kb) return 0; }
    std::vector<int> flags(strs.size(), 0), offs(strs.size(), 0);
    std::vector<int> tie_idx; std::vector<std::string> ext;
    for (size_t v = 0; v < strs.size(); v++) {
        offs[v] = offset_decision(res[v], (int)strs[v].size(), s->ksw.minsc, flags[v]);
        if (flags[v] & F_EQUAL) { tie_idx.push_back((int)v); ext.push_back(extend_slave(s, g[slave_pos[v]], (int)strs[v].size())); }
    }
    if (s->error) return 0;
    if (!ext.empty()) {                          // ... the ties once more with two more bases on either side (alignSlave :98-115)
        tq = now_sec();
        const int kb2 = ksw_batch(s, ext, master_str, res2);
        s->t_ksw += now_sec() - tq;
        if (kb2) return 0;
        for (size_t e = 0; e < ext.size(); e++) {
            const int v = tie_idx[e];
            flags[v] = 0;
            offs[v] = offset_decision(res2[e], (int)ext[e].size(), s->ksw.minsc, flags[v]);
            if (flags[v] & F_EQUAL) flags[v] |= F_FAILED;
        }
    }
    // ... then Aligner::alignSlave's effects, in group order (new tokens are numbered in this order)
    s->t_pre += now_sec() - tq;
    tq = now_sec();
    // (the bookkeeping first; then the reads themselves — turning a reversed slave's reads, finding every read's full-length
    // repeats for the coverage array — in one pass over a flat list, so that the records can be requested ahead of their turn)
    struct Item { const std::vector<int> *l; int cur_len, off; bool flip; };
    std::vector<Item> items;
    // the reads of the slaves queued so far (run once behind the loop — or before a move below destroys a list that is queued)
    auto run_items = [&]() {
        if (s->error || items.empty()) { items.clear(); return; }
        double tb = now_sec();
        struct Ent { int k, cur_len, off; bool flip; };
        std::vector<Ent> ents;
        for (const Item &it : items) for (int k : *it.l) ents.push_back(Ent{k, it.cur_len, it.off, it.flip});
        // one pass on this thread with the records requested ahead of their turn: every record is a cache miss and its list a
        // second one (a few hundred reads per group: waking the host pool 64 times cost more than it saved — 6.5 ms of 9.7)
        s->t_fb += now_sec() - tb; tb = now_sec();
        const size_t ne = ents.size();
        for (size_t e = 0; e < ne && !s->error; e++) {
            if (e + 16 < ne) __builtin_prefetch(&s->rec[ents[e + 16].k]);
            if (e + 8 < ne) __builtin_prefetch(s->rec[ents[e + 8].k].ss.data());
            Rec &r = s->rec[ents[e].k];
            if (ents[e].flip) { reverse_start_stops(r); r.rc = !r.rc; }          // flip_record: the device copy follows with the sync
            const int cur_len = ents[e].cur_len;
            int a = first_full(r, cur_len);
            if (a < 0) { s->error = 5; break; }
            int b = a + 1;
            do {
                if (((int)r.ss[b] - (int)r.ss[a]) == (cur_len - 1)) {
                    const int pos = ents[e].off - (int)r.ss[a];
                    if (pos < 0 || pos + r.L > al.length) { s->error = 6; break; }     // "MEMORY CORRUPTION" in the reference
                    al.plc_rec.push_back((uint32_t)ents[e].k); al.plc_pos.push_back(pos);
                }
                a += 2; b += 2;
                if (a >= (int)(r.ss.size() / 2) * 2) break;
            } while (((int)r.ss[b] - (int)r.ss[a]) == (cur_len - 1));
        }
        s->t_fc += now_sec() - tb; tb = now_sec();
        for (const Ent &e : ents) if (e.flip) al.flips.push_back((uint32_t)e.k);
        s->t_fd += now_sec() - tb;
        items.clear();
    };
    for (size_t v = 0; v < strs.size() && !s->error; v++) {
        int tok = g[slave_pos[v]];
        al.off[tok] = -1;
        if (flags[v] & F_FAILED) continue;
        bool flip = false;
        if (flags[v] & F_REVERSED) {
            if (!rlist(s, tok)) { s->error = 4; break; }
            flip = true;
            const int st = add_string(s, reverse_complement(strs[v]));
            // (the reverse complement may BE a token that already has a list — a group holding X and rc(X) — and that list may be
            // queued above: its reads are placed first, as the reference's slave-by-slave order has it, before the move frees it)
            if (s->reads_of[st - 2]) run_items();
            s->reads_of[st - 2] = std::move(s->reads_of[tok - 2]);
            g[slave_pos[v]] = st;
            tok = st;
        }
        al.off[tok] = al.off[master] + offs[v];
        const std::vector<int> *l = rlist(s, tok);
        if (!l) { s->error = 4; break; }                 // (place_reads' check)
        items.push_back(Item{l, (int)tstr(s, tok).size(), al.off[tok], flip});
    }
    s->t_fa += now_sec() - tq;
    run_items();
    if (s->error) return 0;
    for (size_t q = 0; q < g.size();) {          // "kill the unfounded ones"
        const int tok = g[q];
        auto it = al.off.find(tok);
        if (it != al.off.end() && it->second == -1 && rlist(s, tok) != nullptr) { clear_read_list(s, tok); s->reads_of[tok - 2].reset(); g.erase(g.begin() + (long)q); continue; }
        q++;
    }
    s->t_flip += now_sec() - tq;
    if (sync_coverage(s, al)) return 0;
    tq = now_sec();
    int collapsed_pos = -1;
    CMap4 opts;
    std::vector<uint8_t> refined((size_t)al.length + 2, 0);
    const std::string true_DR = calc_dr_consensus(s, GID, al, collapsed_pos, opts, refined);
    s->t_cons += now_sec() - tq;
    if (s->error) return 0;
    if (true_DR.size() > (size_t)s->prm.highDRsize) { s->group[GID].reset(); return 0; }
    if (opts.size() == 0) {
        if (true_DR.size() < (size_t)s->prm.lowDRsize) { s->group[GID].reset(); return 0; }
        if (is_low_complexity(true_DR)) { s->group[GID].reset(); return 0; }
        const int ab = dr_has_abundant_kmers(true_DR);
        if (ab < 0) { s->error = 12; return 0; }
        if (ab) { s->group[GID].reset(); return 0; }
        int zs = al.zone_start, ze = al.zone_end, diffs = ze - zs + 1 - (int)true_DR.size(), guard = 0;
        while (0 < diffs) {
            const bool re = ze >= 0 && ze < al.length && refined[ze];
            if (!re) { ze--; diffs--; }
            if (0 < diffs) { const bool rs = zs >= 0 && zs < al.length && refined[zs]; if (!rs) { zs++; diffs--; } }
            if (++guard > 4 * al.length) { s->error = 13; return 0; }
        }
        al.zone_start = zs; al.zone_end = ze;
    }
    if (opts.size() > 0) { tq = now_sec(); split_grouped_dr(s, opts, al, collapsed_pos, GID); s->t_split += now_sec() - tq; return 1; }
    const std::string rcd = reverse_complement(true_DR);
    const std::string lau = true_DR < rcd ? true_DR : rcd;           // laurenize, SeqUtils.cpp:89-97
    const bool rev_comp = lau != true_DR;
    s->true_dr[GID] = lau;
    // the true DR joins the batch's DR table; every read of the group gets its deferred updateStartStops
    const uint32_t dr_id = (uint32_t)s->cnt.n_true_drs++;
    s->dr_tab.insert(s->dr_tab.end(), true_DR.begin(), true_DR.end());
    s->dr_tab_end.push_back(s->dr_tab.size());
    for (int tok : *s->group[GID]) {
        auto it = al.off.find(tok);
        if (it == al.off.end() || it->second == -1) continue;          // logError only
        std::vector<int> *l = rlist(s, tok);
        if (!l) { s->error = 4; return 0; }
        for (int k : *l) { Rec &r = s->rec[k]; r.upd = true; r.upd_front = it->second - al.zone_start; r.upd_dr = dr_id; r.upd_rev = rev_comp; }
    }
    return 1;
}

void combine_groups(crass_cons *s)
{   // WorkHorse::combineGroupsWithIdenticalDRs, WorkHorse.cpp:416-452
    std::map<std::string, int> first;
    for (auto it = s->true_dr.begin(); it != s->true_dr.end();) {
        auto p = first.find(it->second);
        if (p != first.end()) {
            auto &src = s->group[it->first]; auto &dst = s->group[p->second];
            if (!src || !dst) { s->error = 14; return; }
            dst->insert(dst->end(), src->begin(), src->end());
            s->group.erase(it->first);
            it = s->true_dr.erase(it);
        } else { first[it->second] = it->first; ++it; }
    }
}

// ReadHolder::updateStartStops for every read that reached a leaf (ReadHolder.cpp:382-511): the pair arithmetic on the
// host, the partial-repeat searches (smithWaterman + Levenshtein) as one device batch
int update_all_start_stops(crass_cons *s)
{
    const uint32_t lowSp = (uint32_t)s->prm.lowSpacerSize;
    const uint32_t n_dr = (uint32_t)s->cnt.n_true_drs;
    std::vector<uint32_t> dr_off(n_dr + 1, 0), dr_len(n_dr + 1, 0);
    for (uint32_t d = 0; d < n_dr; d++) { dr_off[d] = (uint32_t)(d ? s->dr_tab_end[d - 1] : 0); dr_len[d] = (uint32_t)(s->dr_tab_end[d] - dr_off[d]); }
    const std::vector<char> &drc = s->dr_tab;
    std::vector<ConsSwTask> tasks; std::vector<uint8_t> which;          // which: 0 front, 1 back
    uint64_t dir_total = 0;
    const bool timing = getenv("CRASS_TIMING") != nullptr;
    double t_ph = now_sec();
    auto sub = [&](const char *what) { if (timing) { const double t = now_sec(); fprintf(stderr, "[crass_timing] consensus:   updateStartStops: %-28s %.4f s\n", what, t - t_ph); t_ph = t; } };
    // pass 1 (ranges of records on the host pool: every record's list is its own allocation, i.e. a cache miss): the pair
    // arithmetic, and which of the two searches the record needs; pass 2 (in order): the task list and its scratch offsets
    const size_t n_rec = s->rec.size();
    std::vector<uint8_t> need(n_rec, 0);                 // bit 0: front search, bit 1: back search
    {
        const size_t per_task = 4096;
        host_parallel_for((n_rec + per_task - 1) / per_task, 16, [&](size_t t) {
            const size_t k1 = std::min(n_rec, (t + 1) * per_task);
            for (size_t k = t * per_task; k < k1; k++) {
                Rec &r = s->rec[k];
                if (!r.upd || !r.alive) continue;
                const int DR_length = (int)dr_len[r.upd_dr];
                for (size_t q = 0; q + 1 < r.ss.size(); q += 2) {
                    int usable = DR_length - 1;
                    if (r.upd_front >= (int)r.ss[q]) { usable = DR_length - (r.upd_front - (int)r.ss[q]) - 1; r.ss[q] = 0; }
                    else r.ss[q] -= (uint32_t)r.upd_front;
                    r.ss[q + 1] = r.ss[q] + (uint32_t)usable;
                    if (r.ss[q + 1] >= (uint32_t)r.L) r.ss[q + 1] = (uint32_t)r.L - 1;
                }
                if (r.ss.empty()) continue;
                uint8_t nd = 0;
                if (r.ss[0] > lowSp) nd |= 1;
                if ((uint32_t)r.L - r.ss.back() > lowSp) nd |= 2;
                need[k] = nd;
                r.upd_first = r.ss[0]; r.upd_last = r.ss.back();
            }
        });
    }
    for (size_t k = 0; k < n_rec; k++) {
        const uint8_t nd = need[k];
        if (!nd) continue;
        const Rec &r = s->rec[k];
        const int DR_length = (int)dr_len[r.upd_dr];
        if (nd & 1) {
            ConsSwTask t; t.rec = (uint32_t)k; t.dr = r.upd_dr; t.start = 0; t.len = (int)r.upd_first - (int)lowSp; t.dir_off = dir_total;
            dir_total += cons_sw_scratch_bytes((uint32_t)t.len, (uint32_t)DR_length);
            tasks.push_back(t); which.push_back(0);
        }
        if (nd & 2) {
            const uint32_t end_dist = (uint32_t)r.L - r.upd_last;
            ConsSwTask t; t.rec = (uint32_t)k; t.dr = r.upd_dr; t.start = (int)(r.upd_last + lowSp); t.len = (int)(end_dist - lowSp); t.dir_off = dir_total;
            dir_total += cons_sw_scratch_bytes((uint32_t)t.len, (uint32_t)DR_length);
            tasks.push_back(t); which.push_back(1);
        }
    }
    s->cnt.n_sw_tasks = tasks.size();
    sub("pair arithmetic, task list");
    std::vector<ConsSwOut> out(tasks.size());
    std::vector<int32_t> lev(tasks.size(), 0);
    if (!tasks.empty()) {
        // the DR strings follow the records in the device buffer (one character array for the Levenshtein batch)
        const uint64_t seq_bytes = s->hseq.size();
        HCHK(s, s->d_drchars.ensure(drc.size() + 1)); HCHK(s, s->d_droff.ensure(n_dr + 1)); HCHK(s, s->d_drlen.ensure(n_dr + 1));
        HCHK(s, hipMemcpyAsync(s->d_drchars.p, drc.data(), drc.size(), hipMemcpyHostToDevice, s->st));
        HCHK(s, hipMemcpyAsync(s->d_seq.p + seq_bytes, drc.data(), drc.size(), hipMemcpyHostToDevice, s->st));
        HCHK(s, hipMemcpyAsync(s->d_droff.p, dr_off.data(), (n_dr + 1) * 4, hipMemcpyHostToDevice, s->st));
        HCHK(s, hipMemcpyAsync(s->d_drlen.p, dr_len.data(), (n_dr + 1) * 4, hipMemcpyHostToDevice, s->st));
        // chunks bounded by the traceback scratch (long reads: (search length + 1) x (DR + 1) bytes per task)
        const uint64_t budget = 1ull << 30;
        size_t at = 0;
        while (at < tasks.size()) {
            size_t end = at; const uint64_t base = tasks[at].dir_off;
            while (end < tasks.size() && (end == at || tasks[end].dir_off + cons_sw_scratch_bytes((uint32_t)tasks[end].len, dr_len[tasks[end].dr]) - base <= budget)) end++;
            const size_t n = end - at;
            // (straight from / into pageable memory: pinned staging buffers would be allocated per call — measured slower, 7.0 vs 5.8 ms)
            const bool whole = at == 0 && end == tasks.size();           // one chunk (the usual case): its offsets are already relative
            std::vector<ConsSwTask> part;
            if (!whole) { part.assign(tasks.begin() + (long)at, tasks.begin() + (long)end); for (auto &t : part) t.dir_off -= base; }
            const ConsSwTask *chunk = whole ? tasks.data() : part.data();
            const uint64_t bytes = chunk[n - 1].dir_off + cons_sw_scratch_bytes((uint32_t)chunk[n - 1].len, dr_len[chunk[n - 1].dr]);
            HCHK(s, s->d_dirs.ensure(bytes + 64)); HCHK(s, s->d_tasks.ensure(n)); HCHK(s, s->d_swout.ensure(n));
            HCHK(s, hipMemcpyAsync(s->d_tasks.p, chunk, n * sizeof(ConsSwTask), hipMemcpyHostToDevice, s->st));
            HCHK(s, launch_cons_sw(s->d_seq.p, s->d_roff.p, s->d_rlen.p, s->d_tasks.p, (uint32_t)n, s->d_drchars.p, s->d_droff.p, s->d_drlen.p, s->d_dirs.p,
                                   s->d_swout.p, s->st));
            HCHK(s, hipMemcpyAsync(out.data() + at, s->d_swout.p, n * sizeof(ConsSwOut), hipMemcpyDeviceToHost, s->st));
            HCHK(s, hipStreamSynchronize(s->st));
            at = end;
        }
        sub("smithWaterman batch");
        // the Levenshtein filter of SmithWaterman.cpp:283 over (a_ret, b_ret), as one batch of the engine's kernel
        std::vector<uint64_t> a_off(tasks.size()), b_off(tasks.size()); std::vector<uint32_t> a_len(tasks.size()), b_len(tasks.size());
        uint32_t max_len = 1;
        for (size_t q = 0; q < tasks.size(); q++) {
            if (out[q].err) { s->error = 3; return CRASS_OK; }
            a_off[q] = s->rec[tasks[q].rec].roff + (uint64_t)out[q].a_off; a_len[q] = (uint32_t)out[q].a_len;
            b_off[q] = seq_bytes + dr_off[tasks[q].dr] + (uint64_t)out[q].b_off; b_len[q] = (uint32_t)out[q].b_len;
            max_len = std::max(max_len, std::max(a_len[q], b_len[q]));
        }
        const size_t n = tasks.size();
        HCHK(s, s->d_a_off.ensure(n)); HCHK(s, s->d_b_off.ensure(n)); HCHK(s, s->d_a_len.ensure(n)); HCHK(s, s->d_b_len.ensure(n)); HCHK(s, s->d_lev.ensure(n));
        HCHK(s, hipMemcpyAsync(s->d_a_off.p, a_off.data(), n * 8, hipMemcpyHostToDevice, s->st));
        HCHK(s, hipMemcpyAsync(s->d_b_off.p, b_off.data(), n * 8, hipMemcpyHostToDevice, s->st));
        HCHK(s, hipMemcpyAsync(s->d_a_len.p, a_len.data(), n * 4, hipMemcpyHostToDevice, s->st));
        HCHK(s, hipMemcpyAsync(s->d_b_len.p, b_len.data(), n * 4, hipMemcpyHostToDevice, s->st));
        HCHK(s, launch_levenshtein_batch(s->d_seq.p, s->d_a_off.p, s->d_a_len.p, s->d_b_off.p, s->d_b_len.p, n, s->d_lev.p, nullptr, max_len, s->st));
        HCHK(s, hipMemcpyAsync(lev.data(), s->d_lev.p, n * 4, hipMemcpyDeviceToHost, s->st));
        HCHK(s, hipStreamSynchronize(s->st));
    }
    sub("Levenshtein batch");
    // the decisions of updateStartStops on the alignments, front task before back task of a read (tasks are in that order);
    // ranges of tasks on the host pool, cut between records (a record's two tasks stay with one worker)
    std::vector<size_t> cuts(1, 0);
    { const size_t per_task = 8192;
      for (size_t q = per_task; q < tasks.size(); q += per_task) { size_t c = q; while (c < tasks.size() && tasks[c].rec == tasks[c - 1].rec) c++; if (c > cuts.back() && c < tasks.size()) cuts.push_back(c); }
      cuts.push_back(tasks.size()); }
    std::vector<uint64_t> added(cuts.size(), 0);
    std::vector<int> bad(cuts.size(), 0);
    host_parallel_for(cuts.size() - 1, 16, [&](size_t ti) {
    uint64_t n_added = 0;
    for (size_t q = cuts[ti]; q < cuts[ti + 1]; q++) {
        Rec &r = s->rec[tasks[q].rec];
        const ConsSwOut &o = out[q];
        const char *DR = drc.data() + dr_off[tasks[q].dr];
        const size_t DR_size = dr_len[tasks[q].dr];
        int part_s = o.a_start, part_e = o.a_end, a_len = o.a_len, b_len = o.b_len;
        const double similarity_ld = 1.0 - (lev[q] / (double)a_len);
        if (!(similarity_ld >= kPartialSim)) { part_s = 0; part_e = 0; a_len = 0; b_len = 0; }
        if (0 == part_e || part_e - part_s < kMinPartialLen) continue;
        // b_ret = DR.substr(b_off, b_len) (clamped to the string's end).  "DR.rfind(b_ret) + b_ret.size() == DR.size()": the LAST
        // occurrence ends where DR ends <=> DR ends with b_ret; "0 == DR.find(b_ret)" <=> DR starts with it (no strings built:
        // 108 k tasks each made two)
        if ((size_t)o.b_off > DR_size) { bad[ti] = 1; return; }
        const char *b_ret = DR + o.b_off;
        const size_t b_size = std::min((size_t)b_len, DR_size - (size_t)o.b_off);
        if (which[q] == 0) {
            if (0 == memcmp(DR + DR_size - b_size, b_ret, b_size) && 0 == part_s) { r.ss.insert(r.ss.begin(), (uint32_t)part_e); r.ss.insert(r.ss.begin(), 0u); n_added++; }
        } else {
            if ((r.L - 1) == part_e && 0 == memcmp(DR, b_ret, b_size)) {
                uint32_t i = (uint32_t)(part_s + std::abs(a_len - b_len)), j = (uint32_t)part_e;       // startStopsAdd :263-297
                if (j >= (uint32_t)r.L) j = (uint32_t)r.L - 1;
                r.ss.push_back(i); r.ss.push_back(j); n_added++;
            }
        }
    }
    added[ti] = n_added;
    });
    for (size_t ti = 0; ti + 1 < cuts.size(); ti++) { s->cnt.n_partials_added += added[ti]; if (bad[ti]) { s->error = 3; return CRASS_OK; } }
    sub("decisions");
    {   // (the sequence itself is not handed back)
        const size_t per_task = 8192;
        host_parallel_for((n_rec + per_task - 1) / per_task, 16, [&](size_t t) {
            const size_t k1 = std::min(n_rec, (t + 1) * per_task);
            for (size_t k = t * per_task; k < k1; k++) { Rec &r = s->rec[k]; if (r.upd && r.alive && r.upd_rev) { reverse_start_stops(r); r.rc = !r.rc; } }
        });
    }
    sub("final orientation");
    return CRASS_OK;
}

void flatten(crass_cons *s)
{
    s->o_tok_chars.clear(); s->o_tok_off.assign(1, 0);
    for (auto &t : s->tok) { s->o_tok_chars.insert(s->o_tok_chars.end(), t.begin(), t.end()); s->o_tok_off.push_back(s->o_tok_chars.size()); }
    s->o_dr_chars.clear(); s->o_dr_off.assign(1, 0); s->o_grp_gid.clear(); s->o_grp_tokens.clear(); s->o_grp_off.assign(1, 0);
    for (auto &kv : s->group) {
        if (!kv.second) continue;
        auto it = s->true_dr.find(kv.first);
        if (it == s->true_dr.end()) continue;
        s->o_grp_gid.push_back(kv.first);
        s->o_dr_chars.insert(s->o_dr_chars.end(), it->second.begin(), it->second.end()); s->o_dr_off.push_back(s->o_dr_chars.size());
        for (int t : *kv.second) s->o_grp_tokens.push_back((uint32_t)t);
        s->o_grp_off.push_back(s->o_grp_tokens.size());
    }
    const size_t nr = s->rec.size();
    s->o_alive.assign(nr, 0); s->o_rc.assign(nr, 0); s->o_token.assign(nr, 0); s->o_nss.assign(nr, 0); s->o_ss_off.assign(nr + 1, 0); s->o_ss.clear();
    {   // offsets in order, then the lists themselves in ranges on the host pool
        uint64_t at = 0;
        for (size_t k = 0; k < nr; k++) { s->o_ss_off[k] = at; at += s->rec[k].ss.size(); }
        s->o_ss_off[nr] = at;
        s->o_ss.resize((size_t)at);
        const size_t per_task = 8192;
        host_parallel_for((nr + per_task - 1) / per_task, 16, [&](size_t t) {
            const size_t k1 = std::min(nr, (t + 1) * per_task);
            for (size_t k = t * per_task; k < k1; k++) {
                const Rec &r = s->rec[k];
                s->o_alive[k] = r.alive; s->o_rc[k] = r.rc; s->o_nss[k] = (uint32_t)r.ss.size();
                if (!r.ss.empty()) memcpy(s->o_ss.data() + s->o_ss_off[k], r.ss.data(), r.ss.size() * 4);
            }
        });
    }
    s->o_tokread_off.assign(s->tok.size() + 1, 0); s->o_tokread_idx.clear(); s->o_has.assign(s->tok.size() + 1, 0);
    for (size_t t = 0; t < s->tok.size(); t++) {
        s->o_tokread_off[t] = s->o_tokread_idx.size();
        if (s->reads_of[t]) { s->o_has[t] = 1; for (int k : *s->reads_of[t]) { s->o_tokread_idx.push_back((uint64_t)k); s->o_token[(size_t)k] = (uint32_t)t + 2; } }
    }
    s->o_tokread_off[s->tok.size()] = s->o_tokread_idx.size();
    if (s->o_ss.empty()) s->o_ss.push_back(0);
    if (s->o_tokread_idx.empty()) s->o_tokread_idx.push_back(0);
    if (s->o_grp_tokens.empty()) s->o_grp_tokens.push_back(0);
    if (s->o_grp_gid.empty()) s->o_grp_gid.push_back(0);
    if (s->o_dr_chars.empty()) s->o_dr_chars.push_back(0);
    if (s->o_tok_chars.empty()) s->o_tok_chars.push_back(0);
}

} // namespace

// Streams are kept between calls, per device: creating one (and the hardware queue behind it, on its first launch) is
// milliseconds — a tenth of a 10 M-read call.  A call takes one from the pool, crass_hip_consensus_free puts it back.
namespace {
std::mutex g_stream_mu;
std::map<int, std::vector<hipStream_t>> g_stream_pool;
hipStream_t take_stream(int device)
{
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        auto &v = g_stream_pool[device];
        if (!v.empty()) { hipStream_t st = v.back(); v.pop_back(); return st; }
    }
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return st;
}
void give_stream(int device, hipStream_t st)
{
    if (!st) return;
    std::lock_guard<std::mutex> lk(g_stream_mu);
    auto &v = g_stream_pool[device];
    if (v.size() < 8) v.push_back(st); else (void)hipStreamDestroy(st);
}
} // namespace

extern "C" {

int crass_hip_consensus(const crass_params *p, int device, const crass_cons_input *in, crass_cons **out)
{
    if (!p || !in || !out) return CRASS_ERR_INVALID_ARG;
    *out = nullptr;
    if (in->n_rec && (!in->seqs || !in->seq_off || !in->rec_read || !in->rec_lowlexi || !in->rec_token || !in->rec_nss || !in->rec_ss_off || !in->ss_pool))
        return CRASS_ERR_INVALID_ARG;
    if (in->n_tokens && (!in->tok_chars || !in->tok_off)) return CRASS_ERR_INVALID_ARG;
    if (in->n_groups && (!in->grp_tokens || !in->grp_off)) return CRASS_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return CRASS_ERR_NO_DEVICE;
    std::unique_ptr<crass_cons> sp(new (std::nothrow) crass_cons());
    if (!sp) return CRASS_ERR_OOM;
    crass_cons *s = sp.get();
    s->prm = *p; s->device = device; s->max_read_len = (int)in->max_read_len;
    if (hipSetDevice(device) != hipSuccess) return CRASS_ERR_NO_DEVICE;
    s->st = take_stream(device);
    if (!s->st) return CRASS_ERR_HIP;
    build_comp_table(s->comp);
    // Aligner ctor (Aligner.h:112-136): gapo 5, gape 2, minsc 5, match 1, mismatch -3, ambiguous 0
    s->ksw.gapo = 5; s->ksw.gape = 2; s->ksw.minsc = 5;
    { int k = 0; for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) s->ksw.mat[k++] = i == j ? 1 : -3; s->ksw.mat[k++] = 0; } for (int j = 0; j < 5; ++j) s->ksw.mat[k++] = 0; }
    int rc = CRASS_OK;
    const bool timing = getenv("CRASS_TIMING") != nullptr;
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_ph = now_s();
    auto lap = [&](const char *what) { if (timing) { const double t = now_s(); fprintf(stderr, "[crass_timing] consensus: %-36s %.4f s\n", what, t - t_ph); t_ph = t; } };
    auto body = [&]() -> int {
        // ---- the hand-off: records with their RH_Seq (DRLowLexi's orientation, ReadHolder.cpp:573-590), tokens, groups ----
        for (uint32_t t = 0; t < in->n_tokens; t++) { s->tok.emplace_back(in->tok_chars + in->tok_off[t], (size_t)(in->tok_off[t + 1] - in->tok_off[t])); s->reads_of.emplace_back(nullptr); }
        s->next_gid = (int)in->n_groups + 1;
        for (uint32_t g = 0; g < in->n_groups; g++) {
            // a GID without tokens is a NULL entry of mDR2GIDMap (or a key missing from groupKmerCountsMap): the reference
            // `continue`s over it (WorkHorse.cpp:592-595), so it must not become an (empty) group here
            if (in->grp_off[g + 1] < in->grp_off[g]) return CRASS_ERR_INVALID_ARG;
            if (in->grp_off[g + 1] == in->grp_off[g]) continue;
            std::unique_ptr<std::vector<int>> v(new std::vector<int>());
            for (uint64_t q = in->grp_off[g]; q < in->grp_off[g + 1]; q++) {
                if (in->grp_tokens[q] < 2 || in->grp_tokens[q] > in->n_tokens + 1) return CRASS_ERR_INVALID_ARG;      // indexes s->tok[tok - 2]
                v->push_back((int)in->grp_tokens[q]);
            }
            s->group[(int)g + 1] = std::move(v);
        }
        // ---- findConsensusDRs (WorkHorse.cpp:578-611): the ORIGINAL groups in ascending GID order.  Their slave alignments
        // need the token strings and the groups only: the batch is launched now and runs (1.4 ms of kernel for the 10 M-read job)
        // while the host sets the records up ----
        { const int ps = prealign_original_groups(s, (int)in->n_groups); if (ps) return ps; }
        lap("groups, slave alignments launched");
        s->rec.resize((size_t)in->n_rec);
        uint64_t at = 0;
        std::vector<uint64_t> roff((size_t)in->n_rec + 1); std::vector<uint32_t> rlen((size_t)in->n_rec + 1);
        for (uint64_t k = 0; k < in->n_rec; k++) {           // offsets and the token lists: in record order
            Rec &r = s->rec[(size_t)k];
            r.read = in->rec_read[k];
            if (r.read >= in->n_reads) return CRASS_ERR_INVALID_ARG;
            r.L = (int)(in->seq_off[r.read + 1] - in->seq_off[r.read]);
            r.roff = at; at += (uint64_t)r.L;
            roff[(size_t)k] = r.roff; rlen[(size_t)k] = (uint32_t)r.L;
            r.rc = in->rec_lowlexi[k] ? 0 : 1;
            const int tok = (int)in->rec_token[k];
            if (tok < 2 || tok > (int)s->tok.size() + 1) return CRASS_ERR_INVALID_ARG;
            if (!s->reads_of[tok - 2]) s->reads_of[tok - 2].reset(new std::vector<int>());
            s->reads_of[tok - 2]->push_back((int)k);
        }
        // the records' characters and start/stop lists: ranges of records on the host pool.  The host mirror keeps the reads as
        // they came (host_rc = 0); RH_Seq's orientation (DRLowLexi's, ReadHolder.cpp:573-590) is applied on the DEVICE, by the flip
        // kernel over the records with rc set — the host's own copy of a record is only turned when it is read (rseq)
        s->hseq.resize((size_t)at);
        std::vector<uint32_t> rc_list;
        for (uint64_t k = 0; k < in->n_rec; k++) if (s->rec[(size_t)k].rc) rc_list.push_back((uint32_t)k);
        {
            const size_t per_task = 4096, n_rec = (size_t)in->n_rec;
            host_parallel_for((n_rec + per_task - 1) / per_task, 16, [&](size_t t) {
                const size_t k1 = std::min(n_rec, (t + 1) * per_task);
                for (size_t k = t * per_task; k < k1; k++) {
                    Rec &r = s->rec[k];
                    r.ss.assign(in->ss_pool + in->rec_ss_off[k], in->ss_pool + in->rec_ss_off[k] + in->rec_nss[k]);
                    memcpy(s->hseq.data() + r.roff, in->seqs + in->seq_off[r.read], (size_t)r.L);
                    r.host_rc = 0;
                }
            });
        }
        uint64_t dr_room = 0;                              // every group may end with one true DR (<= 4 x maxL, in practice <= highDR)
        for (uint32_t g = 0; g < in->n_groups; g++) dr_room += 64;
        dr_room = std::max<uint64_t>(dr_room * 8, 1u << 16);
        HCHK(s, s->d_seq.ensure((size_t)at + dr_room + 64)); HCHK(s, s->d_roff.ensure(roff.size())); HCHK(s, s->d_rlen.ensure(rlen.size())); HCHK(s, s->d_comp.ensure(128));
        HCHK(s, hipMemcpyAsync(s->d_seq.p, s->hseq.data(), (size_t)at, hipMemcpyHostToDevice, s->st));
        HCHK(s, hipMemcpyAsync(s->d_roff.p, roff.data(), roff.size() * 8, hipMemcpyHostToDevice, s->st));
        HCHK(s, hipMemcpyAsync(s->d_rlen.p, rlen.data(), rlen.size() * 4, hipMemcpyHostToDevice, s->st));
        HCHK(s, hipMemcpyAsync(s->d_comp.p, s->comp, 128, hipMemcpyHostToDevice, s->st));
        if (!rc_list.empty()) {
            HCHK(s, s->d_list.ensure(rc_list.size()));
            HCHK(s, hipMemcpyAsync(s->d_list.p, rc_list.data(), rc_list.size() * 4, hipMemcpyHostToDevice, s->st));
            HCHK(s, launch_cons_flip(s->d_seq.p, s->d_roff.p, s->d_rlen.p, s->d_list.p, (uint32_t)rc_list.size(), s->d_comp.p, s->st));
        }
        HCHK(s, hipStreamSynchronize(s->st));
        lap("records, RH_Seq, upload");
        // (the slave alignments were launched before the records were set up: their results are in pinned memory by now)
        { const int cs = ksw_collect(s, s->n_pre, s->pre_res); if (cs) return cs; }
        for (int gid = 1; gid <= (int)in->n_groups && !s->error && !s->hip_err; gid++) {
            auto it = s->group.find(gid);
            if (it == s->group.end() || !it->second) continue;
            parse_grouped_drs(s, gid);
            if (!s->error && !s->hip_err) combine_groups(s);
        }
        if (s->hip_err) return s->hip_err == (int)hipErrorOutOfMemory ? CRASS_ERR_OOM : CRASS_ERR_HIP;
        lap("groups (coverage, consensus, splits)");
        if (timing) fprintf(stderr, "[crass_timing] consensus:   of which place %.4f, slaves' effects %.4f, coverage round trip %.4f, consensus %.4f, ksw (split groups, ties) %.4f, splits (incl. their groups) %.4f s\n",
                            s->t_place, s->t_flip, s->t_sync, s->t_cons, s->t_ksw, s->t_split);
        if (timing) fprintf(stderr, "[crass_timing] consensus:   slaves: decisions %.4f; effects = bookkeeping %.4f + entries %.4f + records %.4f + gather %.4f s\n",
                            s->t_pre, s->t_fa, s->t_fb, s->t_fc, s->t_fd);
        if (!s->error) {
            if ((uint64_t)s->dr_tab.size() > dr_room) return CRASS_ERR_OVERFLOW;
            const int us = update_all_start_stops(s);
            if (us) return us;
        }
        lap("updateStartStops (SW + Levenshtein)");
        flatten(s);
        lap("flatten");
        return CRASS_OK;
    };
    rc = body();
    if (rc != CRASS_OK) { if (s->st) { (void)hipStreamSynchronize(s->st); give_stream(device, s->st); s->st = nullptr; } return rc; }
    *out = sp.release();
    return CRASS_OK;
}

int crass_hip_consensus_view(const crass_cons *s, crass_cons_view *v)
{
    if (!s || !v) return CRASS_ERR_INVALID_ARG;
    v->error = s->error; v->next_free_gid = s->next_gid; v->n_tokens = (uint32_t)s->tok.size();
    v->tok_chars = s->o_tok_chars.data(); v->tok_off = s->o_tok_off.data();
    v->n_groups = (uint32_t)(s->o_grp_off.size() - 1); v->grp_gid = s->o_grp_gid.data(); v->dr_chars = s->o_dr_chars.data(); v->dr_off = s->o_dr_off.data();
    v->grp_tokens = s->o_grp_tokens.data(); v->grp_off = s->o_grp_off.data();
    v->n_rec = s->rec.size(); v->rec_alive = s->o_alive.data(); v->rec_rc = s->o_rc.data(); v->rec_token = s->o_token.data();
    v->rec_nss = s->o_nss.data(); v->rec_ss_off = s->o_ss_off.data(); v->ss_pool = s->o_ss.data();
    v->tokread_off = s->o_tokread_off.data(); v->tokread_idx = s->o_tokread_idx.data(); v->tok_has_list = s->o_has.data();
    v->counters = s->cnt;
    return CRASS_OK;
}

void crass_hip_consensus_free(crass_cons *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->st) { (void)hipStreamSynchronize(s->st); give_stream(s->device, s->st); s->st = nullptr; }
    delete s;
}

} // extern "C"
