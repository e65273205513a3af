// crass_adapter.cpp — the reference's seam (searchFile / createNonRedundantSet / findSingletons /
// addReadHolder) implemented over the C ABI of libcrass_hip.so.  See crass_adapter.h.
#include "crass_adapter.h"
#include <chrono>
#include <functional>
#include <future>
#include <thread>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <unordered_map>
#include <vector>
#include <unistd.h>

namespace crass_hip {

namespace {

int g_device = -1;
std::vector<int> g_devices;             // setDevices(): more than one entry shards every read set over a group of contexts
bool g_local_copies = false;            // tests: several contexts on one GPU (CRASS_GROUP_LOCAL_COPIES)

// what a finished search stage leaves behind (its index, its device contexts) is freed beside the next stage
struct Reaper { std::future<void> f; void wait() { if (f.valid()) f.get(); } ~Reaper() { wait(); } } g_reaper_out, g_reaper_ctx;
struct DeadContexts {
    struct Item { crass_hip_ctx *c; crass_hip_group *g; crass_fastx_index *ix; };
    std::vector<Item> v;
    void park(crass_hip_ctx *c, crass_hip_group *g, crass_fastx_index *ix) { if (c || g || ix) v.push_back(Item{c, g, ix}); }
    static void destroy(const std::vector<Item> &d, bool device)
    {
        for (auto &p : d) crass_fastx_index_free(p.ix);
        if (device) for (auto &p : d) { if (p.g) crass_hip_group_destroy(p.g); if (p.c) crass_hip_destroy(p.c); }
    }
    bool leave = false;                                     // leaveTeardownToProcessEnd()
    void destroy_async() { if (v.empty() || leave) return; g_reaper_ctx.wait(); auto d = std::move(v); v.clear(); g_reaper_ctx.f = std::async(std::launch::async, [d] { destroy(d, true); }); }
    // (what is still parked at the process's end: the host side is freed, the device contexts go with the process — the HIP
    // runtime may be gone by now)
    ~DeadContexts() { g_reaper_ctx.wait(); destroy(v, false); }
} g_dead;

int device()
{
    if (g_device >= 0) return g_device;
    const char *e = getenv("CRASS_HIP_DEVICE");
    return e ? atoi(e) : 0;
}

const unsigned char *comp_tab()
{
    // reverseComplement table (SeqUtils.cpp:50-59): IUPAC pairs, U->A, identity otherwise, [96] = 64
    // (filled once, by whichever thread gets here first: the holders are filled and written back over the cores)
    struct Tab {
        unsigned char t[128];
        Tab()
        {
            for (int i = 0; i < 128; i++) t[i] = (unsigned char)i;
            const char *a = "ACBDKRSWN", *b = "TGVHMYSWN";
            for (int i = 0; a[i]; i++) {
                t[(int)a[i]] = (unsigned char)b[i]; t[(int)b[i]] = (unsigned char)a[i];
                t[(int)a[i] + 32] = (unsigned char)(b[i] + 32); t[(int)b[i] + 32] = (unsigned char)(a[i] + 32);
            }
            t['U'] = 'A'; t['u'] = 'a'; t[96] = 64;
        }
    };
    static const Tab tab;
    return tab.t;
}

// fn(begin, end) over [0, n) on up to 16 threads (the hand-off's holders are objects of their own: flattening them, filling them
// and writing results back into them is a walk over half a million heap objects — memory latency, which the cores overlap)
// v sorted by cmp: chunks sorted on their own threads, then merged pairwise (the merges of one round beside each other)
template <class T, class C> void par_sort(std::vector<T> &v, C cmp)
{
    const size_t n = v.size();
    unsigned nt = (unsigned)std::min<size_t>(std::max<size_t>(1, n / 16384), std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u));
    unsigned p2 = 1;
    while (p2 * 2 <= nt) p2 *= 2;
    nt = p2;
    if (nt <= 1) { std::sort(v.begin(), v.end(), cmp); return; }
    auto cut = [&](unsigned i) { return n * i / nt; };
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back([&, t] { std::sort(v.begin() + cut(t), v.begin() + cut(t + 1), cmp); });
        std::sort(v.begin(), v.begin() + cut(1), cmp);
        for (auto &x : th) x.join();
    }
    for (unsigned w = 1; w < nt; w *= 2) {
        std::vector<std::thread> th;
        for (unsigned t = 0; t + w < nt; t += 2 * w)
            th.emplace_back([&, t, w] { std::inplace_merge(v.begin() + cut(t), v.begin() + cut(t + w), v.begin() + cut(std::min(nt, t + 2 * w)), cmp); });
        for (auto &x : th) x.join();
    }
}

template <class F> void par_ranges(size_t n, size_t grain, F fn)
{
    const unsigned nt = (unsigned)std::min<size_t>(std::max<size_t>(1, n / std::max<size_t>(1, grain)), std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u));
    if (nt <= 1) { fn((size_t)0, n); return; }
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back([&fn, n, nt, t] { fn(n * t / nt, n * (t + 1) / nt); });
    fn((size_t)0, n / nt);
    for (auto &x : th) x.join();
}

std::string revcomp(const std::string &s)
{
    const unsigned char *t = comp_tab();
    std::string r(s.size(), '\0');
    for (size_t i = 0; i < s.size(); i++) r[i] = (char)t[(unsigned char)s[s.size() - 1 - i] & 127];
    return r;
}

#define CRASS_THROW(msg) throw exception(__FILE__, __LINE__, __PRETTY_FUNCTION__, (msg))

void chk(int status, const char *what)
{
    if (status == CRASS_OK) return;
    if (status == CRASS_ERR_SEARCH_FATAL) CRASS_THROW("Fatal error in search algorithm!");     // libcrispr.cpp:145-148
    CRASS_THROW(std::string(what) + ": " + crass_hip_strerror(status));
}

crass_params to_params(const options &o)
{
    crass_params p;
    p.lowDRsize = o.lowDRsize; p.highDRsize = o.highDRsize; p.lowSpacerSize = o.lowSpacerSize;
    p.highSpacerSize = o.highSpacerSize; p.searchWindowLength = o.searchWindowLength;
    p.minNumRepeats = o.minNumRepeats; p.kmer_clust_size = o.kmer_clust_size;
    return p;
}

// one input file, resident on the GPU between the two passes
struct FileState {
    crass_fastx fx{};
    crass_packed pk{};
    crass_hip_ctx *ctx = nullptr;
    crass_hip_group *grp = nullptr;      // setDevices() named several GPUs: the file's reads are sharded over them
    bool unique_headers = true;
    uint64_t n_found_own = 0;
    ~FileState()
    {
        if (grp) crass_hip_group_destroy(grp);
        if (ctx) crass_hip_destroy(ctx);
        crass_free_packed(&pk);
        crass_free_fastx(&fx);
    }
    std::string field(const uint8_t *buf, const uint64_t *off, uint64_t i) const { return std::string((const char *)buf + off[i], off[i + 1] - off[i]); }
    std::string name(uint64_t i) const { return field(fx.name, fx.name_off, i); }
    std::string seq(uint64_t i) const { return field(fx.seq, fx.seq_off, i); }
};

std::map<std::string, std::unique_ptr<FileState>> &session()
{
    static std::map<std::string, std::unique_ptr<FileState>> s;
    return s;
}

int g_read_counter_p1 = 0, g_read_counter_p2 = 0;     // `static int read_counter` (libcrispr.cpp:92,477)

FileState &open_file(const char *path, const options &opts)
{
    auto &s = session();
    auto it = s.find(path);
    if (it != s.end()) return *it->second;
    std::unique_ptr<FileState> f(new FileState());
    const bool timing = getenv("CRASS_TIMING") != nullptr;          // stage times of the ingest side
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    // the engine context (HIP start-up, code object load, stream and pinned buffers: 0.1-0.2 s) is created on its own
    // thread while this one reads, parses and packs the file
    crass_params p = to_params(opts);
    const int dev = device();
    const std::vector<int> devs = g_devices;
    const bool local = g_local_copies;
    struct Made { int rc = 0; crass_hip_ctx *c = nullptr; crass_hip_group *g = nullptr; };
    struct PendingCtx {
        std::future<Made> fut;
        Made get() { return fut.get(); }
        ~PendingCtx() { if (fut.valid()) { Made r = fut.get(); if (r.c) crass_hip_destroy(r.c); if (r.g) crass_hip_group_destroy(r.g); } }    // (an exception on the way)
    } pending;
    pending.fut = std::async(std::launch::async, [p, dev, devs, local]() {
        Made m;
        if (devs.size() > 1) m.rc = crass_hip_group_create(&p, devs.data(), (int)devs.size(), local ? CRASS_GROUP_LOCAL_COPIES : 0u, &m.g);
        else m.rc = crass_hip_create(&p, devs.size() == 1 ? devs[0] : dev, &m.c);
        return m;
    });
    int rc = crass_read_fastx(path, &f->fx);
    const double t1 = now();
    if (rc == CRASS_ERR_IO) {
        // getFileHandle prints and exit(1)s (SeqUtils.cpp:112-124); a library throws instead
        CRASS_THROW(std::string("Could not open FASTQ ") + path + " for reading.");
    }
    chk(rc, "crass_read_fastx");
    chk(crass_pack_reads(f->fx.seq, f->fx.seq_off, f->fx.n_reads, 2, &f->pk), "crass_pack_reads");
    const double t2 = now();
    for (uint64_t i = 0; i < f->fx.n_reads; i++) if (f->fx.header_id[i] != i) { f->unique_headers = false; break; }
    {
        Made made = pending.get();
        f->ctx = made.c; f->grp = made.g;
        if (made.rc == CRASS_ERR_RCCL) CRASS_THROW(std::string("crass_hip_group_create: ") + crass_hip_group_last_error());
        chk(made.rc, devs.size() > 1 ? "crass_hip_group_create" : "crass_hip_create");
    }
    crass_reads r = f->pk.reads;
    r.header_id = f->unique_headers ? nullptr : f->fx.header_id;
    if (f->grp) chk(crass_hip_group_load_reads(f->grp, &r), "crass_hip_group_load_reads");
    else chk(crass_hip_load_reads(f->ctx, &r), "crass_hip_load_reads");
    if (timing)
        fprintf(stderr, "[crass_timing] %s: %llu reads; read+parse %.3f s, 2-bit pack %.3f s, wait for the context + H2D %.3f s\n", path,
                (unsigned long long)f->fx.n_reads, t1 - t0, t2 - t1, now() - t2);
    FileState &ref = *f;
    s[path] = std::move(f);
    return ref;
}

// fill the per-read fields searchFile/on_match copy from the kseq record (libcrispr.cpp:112-131,425-435)
void fill_from_record(ReadHolder &h, const FileState &f, uint64_t i, bool low_lexi)
{
    std::string seq = f.seq(i);
    h.RH_Seq = low_lexi ? seq : revcomp(seq);          // reverseComplementSeq (ReadHolder.cpp:593-609)
    h.RH_Header = f.name(i);
    if (f.fx.has_comment[i]) h.RH_Comment = f.field(f.fx.comment, f.fx.comment_off, i);
    if (f.fx.has_qual[i]) { h.RH_Qual = f.field(f.fx.qual, f.fx.qual_off, i); h.RH_IsFasta = false; }
    h.RH_WasLowLexi = low_lexi;
}

void sink(ReadMap *mReads, StringCheck *mStringCheck, ReadHolder *candidate, const std::string &dr_lowlexi)
{
    StringToken st = mStringCheck->getToken(dr_lowlexi);
    if (0 == st) {
        st = mStringCheck->addString(dr_lowlexi);
        (*mReads)[st] = new ReadList();
    }
    (*mReads)[st]->push_back(candidate);
}

} // namespace

std::string StringCheck::getString(StringToken token) const
{
    auto it = mT2S_map.find(token);
    if (it == mT2S_map.end()) CRASS_THROW("Token not stored");
    return it->second;
}

void setDevice(int d) { g_device = d; g_devices.clear(); }
void leaveTeardownToProcessEnd(bool yes) { g_dead.leave = yes; }
void setDevices(const std::vector<int> &devices, bool local_copies) { g_devices = devices; g_local_copies = local_copies; }
void releaseDeviceReads() { session().clear(); }

void clearReadMap(ReadMap *m)
{
    for (auto &kv : *m) {
        if (!kv.second) continue;
        for (ReadHolder *h : *kv.second) delete h;
        delete kv.second;
    }
    m->clear();
}

// addReadHolder (libcrispr.cpp:1119-1162) for a holder that has NOT been oriented yet: applies
// ReadHolder::DRLowLexi (ReadHolder.cpp:513-591) on the host copy, then sinks it.  The device
// path does not use this (its records arrive oriented); it exists for interface parity.
void addReadHolder(ReadMap *mReads, StringCheck *mStringCheck, ReadHolder &tmp)
{
    ReadHolder *c = new ReadHolder(tmp);
    const StartStopList &ss = c->RH_StartStops;
    if (ss.size() < 2 || (ss.size() & 1)) { delete c; CRASS_THROW("Cannot obtain read in lowlexi form"); }
    const int n = (int)ss.size() / 2;
    unsigned pick;
    if (n == 1) pick = 0;
    else if (n == 2) {
        if (ss.front() == 0) pick = 2;
        else if (ss.back() == (unsigned)c->RH_Seq.length()) pick = 0;
        else pick = ((int)(ss[1] - ss[0]) > (int)(ss[3] - ss[2])) ? 0 : 2;
    } else pick = 2;
    std::string dr = c->repeatStringAt(pick), rc = revcomp(dr);
    if (dr < rc) c->RH_WasLowLexi = true;
    else {
        c->RH_Seq = revcomp(c->RH_Seq);
        StartStopList m(ss.size());
        const unsigned L = (unsigned)c->RH_Seq.length();
        for (size_t k = 0; k < ss.size(); k++) m[k] = L - 1 - ss[ss.size() - 1 - k];     // reverseStartStops
        c->RH_StartStops = m;
        c->RH_WasLowLexi = false;
        dr = rc;
    }
    sink(mReads, mStringCheck, c, dr);
}

int searchFile(const char *inputFastq, const options &opts, ReadMap *mReads, StringCheck *mStringCheck,
               lookupTable &patternsHash, lookupTable &readsFound, time_t &time_start)
{
    FileState &f = open_file(inputFastq, opts);
    try {
        if (f.grp) chk(crass_hip_group_seed_scan(f.grp), "crass_hip_group_seed_scan");
        else chk(crass_hip_seed_scan(f.ctx), "crass_hip_seed_scan");
    } catch (exception &e) {
        std::cerr << e.what() << std::endl;
        CRASS_THROW("Fatal error in search algorithm!");
    }
    crass_candidates c;
    if (f.grp) chk(crass_hip_group_get_candidates(f.grp, &c), "crass_hip_group_get_candidates");
    else chk(crass_hip_get_candidates(f.ctx, &c), "crass_hip_get_candidates");
    for (uint64_t k = 0; k < c.n; k++) {
        const uint64_t i = c.read_idx[k];
        ReadHolder *h = new ReadHolder();
        fill_from_record(*h, f, i, c.low_lexi[k] != 0);
        h->RH_StartStops.assign(c.ss_pool + c.ss_off[k], c.ss_pool + c.ss_off[k] + c.n_ss[k]);
        h->RH_RepeatLength = (int)c.repeat_len[k];
        sink(mReads, mStringCheck, h, std::string(c.dr_chars + k * (uint64_t)c.dr_stride, c.dr_len[k]));
        // patternsHash[tmp_holder.repeatStringAt(0)] on the UN-oriented holder (libcrispr.cpp:137)
        const StartStopList &ss = h->RH_StartStops;
        std::string first_rep;
        if (c.low_lexi[k]) first_rep = h->repeatStringAt(0);
        else first_rep = revcomp(h->RH_Seq.substr(ss[ss.size() - 2], ss[ss.size() - 1] - ss[ss.size() - 2] + 1));
        patternsHash[first_rep] = true;
        readsFound[h->RH_Header] = true;
    }
    f.n_found_own = c.n;
    g_read_counter_p1 += (int)f.fx.n_reads;
    time_t now; time(&now);
    std::cout << "\r[crass_patternFinder]: Processed " << g_read_counter_p1 << " ..." << difftime(now, time_start) << " sec" << std::flush;
    return (int)f.fx.max_len;
}

Vecstr *createNonRedundantSet(ReadMap &mReads, StringCheck &mStringCheck, DR_Cluster_Map &mDR2GIDMap,
                              std::map<int, bool> &mGroupMap, GroupKmerMap &groupKmerCountsMap, int &nextFreeGID,
                              const options &opts)
{
    // tokens in ascending order (std::map iteration of mReads, WorkHorse.cpp:660-665)
    std::vector<StringToken> toks;
    std::vector<std::string> strs;
    uint32_t stride = 16;
    for (auto &kv : mReads) {
        toks.push_back(kv.first);
        strs.push_back(mStringCheck.getString(kv.first));
        stride = std::max<uint32_t>(stride, (uint32_t)((strs.back().size() + 15) & ~15u));
    }
    std::vector<char> chars(strs.size() * (size_t)stride, 0);
    std::vector<uint16_t> lens(strs.size());
    for (size_t i = 0; i < strs.size(); i++) { memcpy(chars.data() + i * stride, strs[i].data(), strs[i].size()); lens[i] = (uint16_t)strs[i].size(); }
    crass_merge_handle *mh = nullptr;
    chk(crass_merge_create(chars.data(), lens.data(), stride, strs.size(), opts.kmer_clust_size, &mh), "crass_merge_create");
    crass_merge_view v;
    chk(crass_merge_get(mh, &v), "crass_merge_get");
    const int gid_base = nextFreeGID;                 // GIDs continue from the caller's counter (1 in parseSeqFiles)
    for (uint32_t g = 0; g < v.n_groups; g++) {
        const int gid = gid_base + (int)g;
        mGroupMap[gid] = true;
        DR_Cluster *cl = new DR_Cluster();
        std::map<std::string, int> *counts = new std::map<std::string, int>();
        for (uint64_t q = v.grp_off[g]; q < v.grp_off[g + 1]; q++) {
            const uint32_t internal = v.grp_tokens[q];                      // 2 + index into `toks`
            cl->push_back(toks[internal - 2]);
            const std::string &dr = strs[internal - 2];
            for (size_t i = 0; i + 11 <= dr.size(); i++) {                  // local_kmer_CountMap (WorkHorse.cpp:1547-1560,1620-1625)
                std::string km = dr.substr(i, 11), rc = revcomp(km);
                (*counts)[km < rc ? km : rc] += 1;
            }
        }
        mDR2GIDMap[gid] = cl;
        groupKmerCountsMap[gid] = counts;
    }
    nextFreeGID = gid_base + (int)v.n_groups;
    std::cout << '[' << "crass" << "_clusterCore]: " << mReads.size() << " variants mapped to " << mDR2GIDMap.size() << " clusters" << std::endl;
    std::cout << '[' << "crass" << "_clusterCore]: creating non-redundant set" << std::endl;
    Vecstr *out = new Vecstr();
    for (uint32_t i = 0; i < v.n_patterns; i++) out->push_back(std::string(v.pat_chars + v.pat_off[i], v.pat_off[i + 1] - v.pat_off[i]));
    crass_merge_destroy(mh);
    return out;
}

void findSingletons(const char *inputFastq, const options &opts, std::vector<std::string> *nonRedundantPatterns,
                    lookupTable &readsFound, ReadMap *mReads, StringCheck *mStringCheck, time_t &startTime)
{
    FileState &f = open_file(inputFastq, opts);
    std::vector<const char *> pp;
    std::vector<uint32_t> pl;
    for (const auto &s : *nonRedundantPatterns) { pp.push_back(s.data()); pl.push_back((uint32_t)s.size()); }
    if (f.grp) chk(crass_hip_group_set_patterns(f.grp, pp.data(), pl.data(), (uint32_t)pp.size()), "crass_hip_group_set_patterns");
    else chk(crass_hip_set_patterns(f.ctx, pp.data(), pl.data(), (uint32_t)pp.size()), "crass_hip_set_patterns");
    // readsFound is keyed by header (libcrispr.cpp:411): headers found in OTHER files must suppress recruitment here too.
    // The device already knows this file's own pass-1 hits (their count is n_found_own: with one input file, or as long
    // as every entry of readsFound is this file's, there is nothing to add); otherwise every header of readsFound is
    // resolved through the table the reader built for header_id (crass_fastx_find) — no second name index — and the
    // engine sets the flags with one upload + one kernel.
    std::vector<uint64_t> extra;
    // (a group of contexts: a shard only knows its own pass-1 hits, so every found header is named whenever headers repeat)
    if (readsFound.size() != f.n_found_own || !f.unique_headers) {
        extra.reserve(readsFound.size());
        for (const auto &kv : readsFound) {
            const uint64_t i = crass_fastx_find(&f.fx, kv.first.data(), kv.first.size());
            if (i != UINT64_MAX) extra.push_back(i);
        }
    }
    crass_recruits r;
    if (f.grp) {
        chk(crass_hip_group_recruit(f.grp, extra.empty() ? nullptr : extra.data(), extra.size()), "crass_hip_group_recruit");
        chk(crass_hip_group_get_recruits(f.grp, &r), "crass_hip_group_get_recruits");
    } else {
        chk(crass_hip_recruit(f.ctx, extra.empty() ? nullptr : extra.data(), extra.size()), "crass_hip_recruit");
        chk(crass_hip_get_recruits(f.ctx, &r), "crass_hip_get_recruits");
    }
    for (uint64_t k = 0; k < r.n; k++) {
        ReadHolder *h = new ReadHolder();
        fill_from_record(*h, f, r.read_idx[k], r.low_lexi[k] != 0);
        h->RH_StartStops.push_back(r.start[k]);
        h->RH_StartStops.push_back(r.end[k]);
        sink(mReads, mStringCheck, h, std::string(r.dr_chars + k * (uint64_t)r.dr_stride, r.dr_len[k]));
    }
    g_read_counter_p2 += (int)f.fx.n_reads;
    time_t now; time(&now);
    std::cout << "\r[crass_singletonFinder]: Processed " << g_read_counter_p2 << " ..." << difftime(now, startTime) << " sec" << std::flush;
}


// ------------------------------------------------------------------------------------------------------------------
// The first half of WorkHorse::parseSeqFiles (WorkHorse.cpp:336-398) in ONE call, on the device(s) end to end: every
// file's reads form one read set in (file, read) order, pass 1 runs over all of it, createNonRedundantSet runs ON THE
// GPU (dmerge.hip; with several devices the distinct candidate DR strings cross in one RCCL all-gather issued by the
// engine), pass 2 runs over all of it.  Same effects on the hand-off state as the three calls of the seam.
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct JobFiles {
    std::vector<crass_fastx> fx;
    std::vector<uint64_t> base;                 // first job-level read index of every file (+ total)
    std::vector<uint8_t> seq; std::vector<uint64_t> seq_off, header_id;     // concatenation (several files only)
    crass_packed pk{};
    ~JobFiles() { crass_free_packed(&pk); for (auto &f : fx) crass_free_fastx(&f); }
    void locate(uint64_t i, size_t &file, uint64_t &local) const
    {
        file = (size_t)(std::upper_bound(base.begin(), base.end(), i) - base.begin()) - 1;
        local = i - base[file];
    }
};

void fill_holder(ReadHolder &h, const crass_fastx &fx, uint64_t i, bool low_lexi)
{
    std::string seq((const char *)fx.seq + fx.seq_off[i], fx.seq_off[i + 1] - fx.seq_off[i]);
    h.RH_Seq = low_lexi ? seq : revcomp(seq);
    h.RH_Header.assign((const char *)fx.name + fx.name_off[i], fx.name_off[i + 1] - fx.name_off[i]);
    if (fx.has_comment[i]) h.RH_Comment.assign((const char *)fx.comment + fx.comment_off[i], fx.comment_off[i + 1] - fx.comment_off[i]);
    if (fx.has_qual[i]) { h.RH_Qual.assign((const char *)fx.qual + fx.qual_off[i], fx.qual_off[i + 1] - fx.qual_off[i]); h.RH_IsFasta = false; }
    h.RH_WasLowLexi = low_lexi;
}
// ---- bounded-memory ingest (the reference's memory model: one record at a time, every input read once per pass,
//      kseq.cpp:171-226, WorkHorse.cpp:336-393) ----
// Whole-file ingest keeps every record's text for the hand-off (3 GB of host memory for 5 M x 150 bp); streamed ingest reads
// the inputs in chunks (crass_fastx_stream_*), keeps the 2-bit packed reads (40 bytes per 150 bp read) + 8-12 bytes of
// bookkeeping per read, and reads the inputs a SECOND time after the search to pick up the text of the ~1 % of reads that
// are handed on — as the reference itself reads every file twice.  CRASS_INGEST=stream|whole forces one; default: stream when
// holding the inputs' records would take more than a quarter of the host's available memory (want_streamed_ingest).
struct StreamedJob {
    std::vector<uint32_t> packed;               // uniform stride (stride > 0) or tight words + word_off
    std::vector<uint64_t> word_off;
    std::vector<uint32_t> lengths;              // empty while every read has length uni_len
    std::vector<uint64_t> exc_read, exc_off{0};
    std::vector<uint8_t> exc_bytes;
    std::vector<uint64_t> hid;                  // job-level index of the first read with the same header
    bool any_dup = false;
    uint64_t n = 0;
    uint32_t stride = 0, uni_len = 0, max_len = 0, min_len = 0xFFFFFFFFu;
    std::vector<uint64_t> base{0};              // first job-level read index of every file (+ total)
    size_t chunk_bytes = 0;

    void reserve(uint64_t n_est, uint32_t words_per_read) { packed.reserve((size_t)n_est * words_per_read + 8); }
    void to_ragged()                            // a read of another length arrived: tight words + offsets from here on
    {
        if (!stride) return;
        const uint32_t w = (uni_len + 15) / 16;
        std::vector<uint32_t> tight((size_t)n * w + 4, 0);
        word_off.resize(n); lengths.assign(n, uni_len);
        for (uint64_t i = 0; i < n; i++) { word_off[i] = i * (uint64_t)w; memcpy(tight.data() + i * (uint64_t)w, packed.data() + i * (uint64_t)stride, (size_t)w * 4); }
        tight.resize((size_t)n * w);
        packed.swap(tight);
        stride = 0; uni_len = 0;
    }
    void append(const crass_fastx &fx)
    {
        crass_packed pk{};
        chk(crass_pack_reads(fx.seq, fx.seq_off, fx.n_reads, 0, &pk), "crass_pack_reads");
        const crass_reads &r = pk.reads;
        const uint64_t m = r.n_reads;
        if (n == 0 && r.uniform_len) { stride = r.stride_words; uni_len = r.uniform_len; }
        if (stride && !(r.uniform_len == uni_len && r.stride_words == stride)) to_ragged();
        if (stride) {
            packed.resize((size_t)(n + m) * stride);
            memcpy(packed.data() + (size_t)n * stride, r.packed, (size_t)m * stride * 4);
        } else {
            uint64_t at = packed.size();
            for (uint64_t i = 0; i < m; i++) {
                const uint32_t len = r.uniform_len ? r.uniform_len : r.lengths[i];
                const uint32_t w = (len + 15) / 16;
                const uint32_t *src = r.stride_words ? r.packed + i * (uint64_t)r.stride_words : r.packed + r.word_off[i];
                word_off.push_back(at); lengths.push_back(len);
                packed.insert(packed.end(), src, src + w);
                at += w;
            }
        }
        for (uint64_t e = 0; e < r.n_exceptions; e++) {
            exc_read.push_back(n + r.exc_read[e]);
            exc_bytes.insert(exc_bytes.end(), r.exc_bytes + r.exc_off[e], r.exc_bytes + r.exc_off[e + 1]);
            exc_off.push_back(exc_bytes.size());
        }
        // header ids: kept from the first duplicate on (identity before it)
        for (uint64_t i = 0; i < m; i++) {
            if (!any_dup && fx.header_id[i] != n + i) { any_dup = true; hid.resize(n + i); for (uint64_t q = 0; q < n + i; q++) hid[q] = q; }
            if (any_dup) hid.push_back(fx.header_id[i]);
        }
        for (uint64_t i = 0; i < m; i++) { const uint32_t l = (uint32_t)(fx.seq_off[i + 1] - fx.seq_off[i]); max_len = std::max(max_len, l); min_len = std::min(min_len, l); }
        n += m;
        crass_free_packed(&pk);
    }
    // crass_pack_reads' pad rule (mode 2) on the whole job: short reads of differing lengths take one stride when that costs
    // at most twice the words
    void finish(crass_reads &r)
    {
        if (!stride && n) {
            uint64_t tight = 0;
            for (uint32_t l : lengths) tight += (l + 15) / 16;
            const uint32_t w = (max_len + 15) / 16;
            if (max_len <= 256 && max_len >= 64 && n * (uint64_t)w <= 2 * tight) {
                std::vector<uint32_t> padded((size_t)n * w + 4, 0);
                for (uint64_t i = 0; i < n; i++) memcpy(padded.data() + i * (uint64_t)w, packed.data() + word_off[i], (size_t)((lengths[i] + 15) / 16) * 4);
                packed.swap(padded);
                std::vector<uint64_t>().swap(word_off);
                stride = w;
            }
        }
        packed.resize(packed.size() + 4, 0);                    // (the kernels may read a few words past the last read)
        memset(&r, 0, sizeof(r));
        r.n_reads = n; r.packed = packed.data(); r.stride_words = stride;
        r.word_off = stride ? nullptr : word_off.data();
        r.uniform_len = (n && min_len == max_len) ? max_len : 0;
        r.lengths = r.uniform_len ? nullptr : lengths.data();
        r.n_exceptions = exc_read.size(); r.exc_read = exc_read.data(); r.exc_off = exc_off.data(); r.exc_bytes = exc_bytes.data();
        r.header_id = any_dup ? hid.data() : nullptr;
        if (!any_dup) std::vector<uint64_t>().swap(hid);
    }
    void locate(uint64_t i, size_t &file, uint64_t &local) const
    {
        file = (size_t)(std::upper_bound(base.begin(), base.end(), i) - base.begin()) - 1;
        local = i - base[file];
    }
};

bool want_streamed_ingest(const Vecstr &files)
{
    if (const char *e = getenv("CRASS_INGEST")) {
        if (!strcmp(e, "stream")) return true;
        if (!strcmp(e, "whole")) return false;
    }
    uint64_t plain = 0, gz = 0;
    for (const std::string &f : files) {
        FILE *fp = fopen(f.c_str(), "rb");
        if (!fp) continue;                                      // (the reader reports it)
        unsigned char m[2] = {0, 0};
        const bool is_gz = fread(m, 1, 2, fp) == 2 && m[0] == 0x1f && m[1] == 0x8b;
        fseek(fp, 0, SEEK_END);
        const long sz = ftell(fp);
        fclose(fp);
        if (sz > 0) (is_gz ? gz : plain) += (uint64_t)sz;
    }
    // auto.  Plain-text inputs: streamed — bounded memory, and since its chunks are read ahead and assembled in blocks it is the
    // faster of the two as well (50 M reads end to end: 5.6 s against 7.3), although it reads every input twice.  With a gzip input
    // the second pass inflates again and the whole-file reader inflates through libdeflate, three times zlib's rate: whole-file
    // while its footprint — ~3.6 bytes per byte of text while the records are assembled (file image + parsed pieces + record
    // arrays) — stays within a quarter of the memory this host has available (8 GB if that cannot be read), else streamed.
    if (gz == 0 && plain > 0) return true;
    const uint64_t text = plain + 4 * gz;
    uint64_t avail = 32ull << 30;
    if (FILE *fp = fopen("/proc/meminfo", "r")) {
        char line[256];
        while (fgets(line, sizeof(line), fp)) if (!strncmp(line, "MemAvailable:", 13)) { avail = (uint64_t)atoll(line + 13) << 10; break; }
        fclose(fp);
    }
    return text * 36 / 10 > avail / 4;
}

// a hand-off record whose text (header, comment, sequence, quality) is still to come from its input file
struct Fill { uint64_t idx; ReadHolder *h; bool low; };
} // namespace

int searchAndRecruit(const Vecstr &seqFiles, const options &opts, ReadMap *mReads, StringCheck *mStringCheck,
                     DR_Cluster_Map &mDR2GIDMap, std::map<int, bool> &mGroupMap, GroupKmerMap &groupKmerCountsMap,
                     int &nextFreeGID, lookupTable &patternsHash, lookupTable &readsFound, Vecstr *nonRedundantPatterns,
                     time_t &time_start)
{
    if (mStringCheck->mNextFreeToken != 1 || !mReads->empty()) CRASS_THROW("searchAndRecruit needs a fresh StringCheck / ReadMap (it is the whole search stage)");
    const bool timing = getenv("CRASS_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    g_dead.destroy_async();                                // (an earlier call's contexts, if no stage in between took them)
    crass_params p = to_params(opts);
    std::vector<int> devs = g_devices;
    if (devs.empty()) devs.push_back(device());
    const bool local = g_local_copies;
    // contexts (HIP start-up, code objects, RCCL communicators) come up on their own thread while this one reads the files
    struct Made { int rc = 0; crass_hip_ctx *c = nullptr; crass_hip_group *g = nullptr; };
    struct Pending {
        std::future<Made> fut; Made m; bool got = false;
        Made &get() { if (!got) { m = fut.get(); got = true; } return m; }
        ~Pending() { if (fut.valid() || got) { Made &r = get(); if (r.g) crass_hip_group_destroy(r.g); if (r.c) crass_hip_destroy(r.c); } }
    } dev;
    dev.fut = std::async(std::launch::async, [p, devs, local]() {
        Made m;
        if (devs.size() > 1) m.rc = crass_hip_group_create(&p, devs.data(), (int)devs.size(), local ? CRASS_GROUP_LOCAL_COPIES : 0u, &m.g);
        else m.rc = crass_hip_create(&p, devs[0], &m.c);
        return m;
    });
    // The INDEXED reader (crass_index_fastx_files) — every input stays mapped (a gzip'd one: inflated once), every read is packed as
    // its piece is parsed and its text dropped, the records that are handed on are parsed again when the hand-off asks for them.
    // 40 + 12 bytes of host memory per 150 bp read and no second pass over the inputs; several inputs (paired-end files) are one read
    // set in (file, read) order with header ids across them.  CRASS_INGEST=index|whole|stream forces a reader (index: an input it
    // does not take is an error then); default: the index, and for what it does not take stream / whole (want_streamed_ingest).
    struct Index { crass_fastx_index *ix = nullptr; ~Index() { crass_fastx_index_free(ix); } } IX;
    bool indexed = false;
    {
        const char *e = getenv("CRASS_INGEST");
        const bool force = e && !strcmp(e, "index");
        if ((force || !e) && !seqFiles.empty()) {
            std::vector<const char *> paths;
            for (const std::string &f : seqFiles) paths.push_back(f.c_str());
            const int rc = crass_index_fastx_files(paths.data(), (uint32_t)paths.size(), &IX.ix);
            if (rc == CRASS_ERR_IO) {
                for (const std::string &f : seqFiles) if (access(f.c_str(), R_OK) != 0) CRASS_THROW(std::string("Could not open FASTQ ") + f + " for reading.");
                CRASS_THROW(std::string("Could not open FASTQ ") + seqFiles[0] + " for reading.");
            }
            if (rc == CRASS_OK) indexed = true;
            else if (rc != CRASS_ERR_UNSUPPORTED || force) chk(rc, "crass_index_fastx_files");
        }
    }
    const bool streamed = !indexed && want_streamed_ingest(seqFiles);
    JobFiles J;
    StreamedJob S;
    int max_len = 0;
    uint64_t n = 0;
    double t1 = t0;
    crass_reads r;
    if (indexed) {
        uint32_t ml = 0; int lr = 0;
        chk(crass_fastx_index_reads(IX.ix, &r, &ml, &lr), "crass_fastx_index_reads");
        n = r.n_reads; max_len = (int)ml;
        t1 = now();
    } else if (streamed) {
        // pass A over the inputs: chunk -> 2-bit pack -> append; the chunk's text is dropped
        struct Names { crass_name_table *t = crass_name_table_create(); ~Names() { crass_name_table_destroy(t); } } names;
        double t_append = 0;
        for (size_t f = 0; f < seqFiles.size(); f++) {
            crass_fastx_stream *st = nullptr;
            const int rc = crass_fastx_stream_open(seqFiles[f].c_str(), 0, names.t, S.n, &st);
            if (rc == CRASS_ERR_IO) CRASS_THROW(std::string("Could not open FASTQ ") + seqFiles[f] + " for reading.");
            chk(rc, "crass_fastx_stream_open");
            struct Close { crass_fastx_stream *s; ~Close() { crass_fastx_stream_close(s); } } closer{st};
            for (;;) {
                crass_fastx fx;
                chk(crass_fastx_stream_next(st, &fx), "crass_fastx_stream_next");
                if (fx.n_reads == 0) break;
                if (S.n == 0 && f == 0) {
                    // one allocation for the job instead of a doubling vector (whose growth holds old + new at once): the read
                    // count from the inputs' sizes and the first chunk's bytes per record (gzip: ~4 x its size in text)
                    uint64_t text = 0;
                    for (const std::string &p2 : seqFiles) {
                        FILE *fp = fopen(p2.c_str(), "rb");
                        if (!fp) continue;
                        unsigned char m2[2] = {0, 0};
                        const bool is_gz = fread(m2, 1, 2, fp) == 2 && m2[0] == 0x1f && m2[1] == 0x8b;
                        fseek(fp, 0, SEEK_END);
                        const long sz = ftell(fp);
                        fclose(fp);
                        if (sz > 0) text += (uint64_t)sz * (is_gz ? 4u : 1u);
                    }
                    const uint64_t rec_bytes = std::max<uint64_t>(1, (fx.seq_off[fx.n_reads] + fx.name_off[fx.n_reads] + fx.qual_off[fx.n_reads] + fx.comment_off[fx.n_reads]) / fx.n_reads + 3);
                    const uint64_t est = text / rec_bytes + text / rec_bytes / 16 + 1024;
                    S.reserve(est, (uint32_t)((fx.max_len + 15) / 16));
                    crass_name_table_reserve(names.t, est);
                }
                const double ta0 = now();
                S.append(fx);
                t_append += now() - ta0;
            }
            S.base.push_back(S.n);
        }
        n = S.n; max_len = (int)S.max_len;
        t1 = now();
        if (timing) fprintf(stderr, "[crass_timing] streamed ingest: packing and appending the chunks %.3f s of the pass\n", t_append);
        S.finish(r);
    } else {
    J.fx.resize(seqFiles.size());
    J.base.assign(1, 0);
    for (size_t f = 0; f < seqFiles.size(); f++) {
        const int rc = crass_read_fastx(seqFiles[f].c_str(), &J.fx[f]);
        if (rc == CRASS_ERR_IO) CRASS_THROW(std::string("Could not open FASTQ ") + seqFiles[f] + " for reading.");
        chk(rc, "crass_read_fastx");
        J.base.push_back(J.base.back() + J.fx[f].n_reads);
        max_len = std::max(max_len, (int)J.fx[f].max_len);
    }
    n = J.base.back();
    t1 = now();
    const uint8_t *seq = nullptr; const uint64_t *off = nullptr; const uint64_t *hid = nullptr;
    if (seqFiles.size() == 1) {
        seq = J.fx[0].seq; off = J.fx[0].seq_off;
        for (uint64_t i = 0; i < n; i++) if (J.fx[0].header_id[i] != i) { hid = J.fx[0].header_id; break; }
    } else {
        // one read set in (file, read) order; header ids across files: the first read of the JOB with the same name
        // (readsFound is keyed by the header string whatever file it came from, libcrispr.cpp:138,411)
        J.seq_off.assign(1, 0); J.seq_off.reserve(n + 1); J.header_id.resize(n);
        uint64_t bytes = 0;
        for (auto &f : J.fx) bytes += f.n_reads ? f.seq_off[f.n_reads] : 0;
        J.seq.resize(bytes);
        uint64_t at = 0;
        bool any = false;
        for (size_t f = 0; f < J.fx.size(); f++) {
            const crass_fastx &x = J.fx[f];
            if (x.n_reads) memcpy(J.seq.data() + at, x.seq, x.seq_off[x.n_reads]);
            for (uint64_t i = 0; i < x.n_reads; i++) {
                J.seq_off.push_back(at + x.seq_off[i + 1]);
                uint64_t g = J.base[f] + x.header_id[i];
                if (x.header_id[i] == i)                      // first of its name in this file: did an earlier file have it?
                    for (size_t e = 0; e < f; e++) {
                        const uint64_t q = crass_fastx_find(&J.fx[e], (const char *)x.name + x.name_off[i], x.name_off[i + 1] - x.name_off[i]);
                        if (q != UINT64_MAX) { g = J.base[e] + q; break; }
                    }
                else g = J.header_id[J.base[f] + x.header_id[i]];      // same as the first of its name in this file
                J.header_id[J.base[f] + i] = g;
                any = any || g != J.base[f] + i;
            }
            at += x.n_reads ? x.seq_off[x.n_reads] : 0;
        }
        seq = J.seq.data(); off = J.seq_off.data(); hid = any ? J.header_id.data() : nullptr;
    }
    chk(crass_pack_reads(seq, off, n, 2, &J.pk), "crass_pack_reads");
    r = J.pk.reads;
    r.header_id = hid;
    }
    const double t2 = now();
    auto rss_mb = [] {
        long kb = 0;
        if (FILE *f = fopen("/proc/self/status", "r")) {
            char line[256];
            while (fgets(line, sizeof(line), f)) if (!strncmp(line, "VmRSS:", 6)) { kb = atol(line + 6); break; }
            fclose(f);
        }
        return kb / 1024.0;
    };
    const double rss_ingested = timing ? rss_mb() : 0;
    Made &made = dev.get();
    const double rss_ctx = timing ? rss_mb() : 0;
    if (made.rc == CRASS_ERR_RCCL) CRASS_THROW(std::string("crass_hip_group_create: ") + crass_hip_group_last_error());
    chk(made.rc, made.g ? "crass_hip_group_create" : "crass_hip_create");
    crass_candidates c; crass_merge_view v; crass_recruits q;
    // (the reads are resident on the device(s) once loaded: the host copy of a streamed job goes at once)
    auto drop_host_reads = [&] {
        if (!streamed) return;
        std::vector<uint32_t>().swap(S.packed); std::vector<uint64_t>().swap(S.word_off); std::vector<uint32_t>().swap(S.lengths);
        std::vector<uint64_t>().swap(S.hid); std::vector<uint8_t>().swap(S.exc_bytes);
    };
    try {
        if (made.g) {
            chk(crass_hip_group_load_reads(made.g, &r), "crass_hip_group_load_reads");
            drop_host_reads();
            const int s = crass_hip_group_step(made.g);
            if (s == CRASS_ERR_RCCL) CRASS_THROW(std::string("crass_hip_group_step: ") + crass_hip_group_last_error());
            chk(s, "crass_hip_group_step");
            chk(crass_hip_group_get_candidates(made.g, &c), "crass_hip_group_get_candidates");
            chk(crass_hip_group_get_merge(made.g, &v), "crass_hip_group_get_merge");
            chk(crass_hip_group_get_recruits(made.g, &q), "crass_hip_group_get_recruits");
        } else {
            chk(crass_hip_load_reads(made.c, &r), "crass_hip_load_reads");
            drop_host_reads();
            chk(crass_hip_seed_scan(made.c), "crass_hip_seed_scan");
            chk(crass_hip_merge(made.c, nullptr, nullptr, 0, 0), "crass_hip_merge");
            chk(crass_hip_recruit(made.c, nullptr, 0), "crass_hip_recruit");
            chk(crass_hip_get_candidates(made.c, &c), "crass_hip_get_candidates");
            chk(crass_hip_get_merge(made.c, &v), "crass_hip_get_merge");
            chk(crass_hip_get_recruits(made.c, &q), "crass_hip_get_recruits");
        }
    } catch (exception &e) {
        std::cerr << e.what() << std::endl;
        CRASS_THROW("Fatal error in search algorithm!");
    }
    const double t3 = now();
    time_t tnow; time(&tnow);
    g_read_counter_p1 += (int)n;
    std::cout << "\r[crass_patternFinder]: Processed " << g_read_counter_p1 << " ..." << difftime(tnow, time_start) << " sec" << std::endl;
    // ---- the hand-off ----
    double th[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int thi = 0; double th_prev = now();
    auto hlap = [&]() { const double t = now(); if (thi < 8) th[thi++] = t - th_prev; th_prev = t; };
    // StringCheck: tokens 2.. in discovery order (the engine's token t is the reference's token t)
    std::vector<ReadList *> list_of((size_t)v.n_tokens + 2, nullptr);      // (tokens are dense: no look-up in the map per record)
    for (uint32_t t = 0; t < v.n_tokens; t++) {
        const StringToken st = mStringCheck->addString(std::string(v.tok_chars + v.tok_off[t], (size_t)(v.tok_off[t + 1] - v.tok_off[t])));
        ReadList *l = new ReadList();
        (*mReads)[st] = l;
        if ((size_t)st < list_of.size()) list_of[(size_t)st] = l;
    }
    hlap();      // 0: tokens
    if (v.n_candidates != c.n) CRASS_THROW("candidate / token count mismatch");
    // (streamed ingest: the records' text is picked up by a second pass over the inputs, below — `fills`)
    std::vector<Fill> fills;
    std::vector<ReadHolder *> cand_holders(c.n);
    for (uint64_t k = 0; k < c.n; k++) {
        ReadHolder *h = new ReadHolder();
        if (streamed || indexed) fills.push_back(Fill{c.read_idx[k], h, c.low_lexi[k] != 0});
        else {
            size_t f; uint64_t i;
            J.locate(c.read_idx[k], f, i);
            fill_holder(*h, J.fx[f], i, c.low_lexi[k] != 0);
        }
        h->RH_StartStops.assign(c.ss_pool + c.ss_off[k], c.ss_pool + c.ss_off[k] + c.n_ss[k]);
        h->RH_RepeatLength = (int)c.repeat_len[k];
        { const size_t tk = (size_t)v.cand_token[k]; ReadList *l = tk < list_of.size() ? list_of[tk] : nullptr; (l ? l : (*mReads)[(StringToken)tk])->push_back(h); }
        cand_holders[k] = h;
    }
    hlap();      // 1: candidate holders
    // createNonRedundantSet's outputs: groups (GID -> tokens), per-group k-mer counts, the pattern list
    const int gid_base = nextFreeGID;
    {
        // a group's cluster and its 11-mer counts depend on the group alone: built over the cores (a million map operations for the
        // 42 k tokens of a 50 M-read job), put into the job's maps in GID order
        std::vector<DR_Cluster *> cls(v.n_groups, nullptr);
        std::vector<std::map<std::string, int> *> cnts(v.n_groups, nullptr);
        par_ranges((size_t)v.n_groups, 1, [&](size_t a, size_t b) {
            for (size_t g = a; g < b; g++) {
                DR_Cluster *cl = new DR_Cluster();
                std::map<std::string, int> *counts = new std::map<std::string, int>();
                for (uint64_t w = v.grp_off[g]; w < v.grp_off[g + 1]; w++) {
                    const uint32_t t = v.grp_tokens[w];
                    cl->push_back((StringToken)t);
                    const std::string dr(v.tok_chars + v.tok_off[t - 2], (size_t)(v.tok_off[t - 1] - v.tok_off[t - 2]));
                    for (size_t i = 0; i + 11 <= dr.size(); i++) {                  // local_kmer_CountMap (WorkHorse.cpp:1547-1560,1620-1625)
                        std::string km = dr.substr(i, 11), rc = revcomp(km);
                        (*counts)[km < rc ? km : rc] += 1;
                    }
                }
                cls[g] = cl; cnts[g] = counts;
            }
        });
        for (uint32_t g = 0; g < v.n_groups; g++) {
            const int gid = gid_base + (int)g;
            mGroupMap[gid] = true;
            mDR2GIDMap[gid] = cls[g];
            groupKmerCountsMap[gid] = cnts[g];
        }
    }
    nextFreeGID = gid_base + (int)v.n_groups;
    std::cout << "[crass_clusterCore]: " << v.n_tokens << " variants mapped to " << mDR2GIDMap.size() << " clusters" << std::endl;
    std::cout << "[crass_clusterCore]: creating non-redundant set" << std::endl;
    if (nonRedundantPatterns) {
        nonRedundantPatterns->clear();
        for (uint32_t i = 0; i < v.n_patterns; i++) nonRedundantPatterns->push_back(std::string(v.pat_chars + v.pat_off[i], (size_t)(v.pat_off[i + 1] - v.pat_off[i])));
    }
    if (v.n_patterns) std::cout << "[crass_clusterCore]: " << v.n_patterns << " non-redundant patterns." << std::endl;
    hlap();      // 2: groups, patterns
    for (uint64_t k = 0; k < q.n; k++) {
        ReadHolder *h = new ReadHolder();
        if (streamed || indexed) fills.push_back(Fill{q.read_idx[k], h, q.low_lexi[k] != 0});
        else {
            size_t f; uint64_t i;
            J.locate(q.read_idx[k], f, i);
            fill_holder(*h, J.fx[f], i, q.low_lexi[k] != 0);
        }
        h->RH_StartStops.push_back(q.start[k]);
        h->RH_StartStops.push_back(q.end[k]);
        StringToken st = (StringToken)q.token[k];
        if (st >= 2 && (size_t)st < list_of.size() && list_of[(size_t)st]) { list_of[(size_t)st]->push_back(h); continue; }
        if (st < 2 || !mReads->count(st)) {                 // (not expected with the engine's own pattern set: addReadHolder's general form)
            const std::string dr(q.dr_chars + k * (uint64_t)q.dr_stride, q.dr_len[k]);
            st = mStringCheck->getToken(dr);
            if (0 == st) { st = mStringCheck->addString(dr); (*mReads)[st] = new ReadList(); }
        }
        (*mReads)[st]->push_back(h);
    }
    hlap();      // 3: recruit holders
    double t_fill = 0;
    if (indexed && !fills.empty()) {
        // the text of the records that are handed on, parsed from the mapping (crass_fastx_index_fetch: every core), then the
        // holders filled — also over the cores: a holder is its own object
        const double tf0 = now();
        std::vector<uint64_t> want(fills.size());
        for (size_t k = 0; k < fills.size(); k++) want[k] = fills[k].idx;
        crass_fastx fx;
        chk(crass_fastx_index_fetch(IX.ix, want.data(), want.size(), &fx), "crass_fastx_index_fetch");
        struct Free { crass_fastx *f; ~Free() { crass_free_fastx(f); } } fr{&fx};
        const size_t nf = fills.size();
        const unsigned nt = (unsigned)std::min<size_t>(std::max<size_t>(1, nf / 4096), std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u));
        std::vector<std::thread> th;
        auto run = [&](unsigned t) { for (size_t k = nf * t / nt; k < nf * (t + 1) / nt; k++) fill_holder(*fills[k].h, fx, k, fills[k].low); };
        for (unsigned t = 1; t < nt; t++) th.emplace_back(run, t);
        run(0);
        for (auto &x : th) x.join();
        t_fill = now() - tf0;
    }
    hlap();      // 4: text
    if (indexed) crass_fastx_index_drop_text(IX.ix);
    hlap();      // 5: mappings dropped       // (the inputs' mappings, over the cores and now: not by the thread that frees the rest later)
    if (streamed && !fills.empty()) {
        // pass B over the inputs (the reference reads every file a second time too, findSingletons): the text of the records
        // that are handed on, chunk by chunk
        const double tf0 = now();
        std::sort(fills.begin(), fills.end(), [](const Fill &a, const Fill &b) { return a.idx < b.idx; });
        size_t at = 0;
        for (size_t f = 0; f < seqFiles.size() && at < fills.size(); f++) {
            if (fills[at].idx >= S.base[f + 1]) continue;          // nothing wanted from this file
            crass_fastx_stream *st = nullptr;
            chk(crass_fastx_stream_open(seqFiles[f].c_str(), 0, nullptr, 0, &st), "crass_fastx_stream_open");
            struct Close { crass_fastx_stream *s; ~Close() { crass_fastx_stream_close(s); } } closer{st};
            uint64_t first = S.base[f];
            while (at < fills.size() && fills[at].idx < S.base[f + 1]) {
                crass_fastx fx;
                chk(crass_fastx_stream_next(st, &fx), "crass_fastx_stream_next");
                if (fx.n_reads == 0) CRASS_THROW("an input changed between the two passes over it");
                for (; at < fills.size() && fills[at].idx < first + fx.n_reads; at++)
                    fill_holder(*fills[at].h, fx, fills[at].idx - first, fills[at].low);
                first += fx.n_reads;
            }
        }
        if (at != fills.size()) CRASS_THROW("an input changed between the two passes over it");
        t_fill = now() - tf0;
    }
    {
        // patternsHash[tmp_holder.repeatStringAt(0)] on the UN-oriented holder and readsFound[header] (libcrispr.cpp:137-138) for
        // every pass-1 record: the keys are made over the cores, sorted, and go into the maps in order with the end as the hint
        // — half a million look-ups in a tree of strings were 0.2 s of the hand-off
        std::vector<std::string> pat((size_t)c.n);
        std::vector<const std::string *> hdr((size_t)c.n);
        par_ranges((size_t)c.n, 8192, [&](size_t a, size_t b) {
            for (size_t k = a; k < b; k++) {
                ReadHolder *h = cand_holders[k];
                const StartStopList &ss = h->RH_StartStops;
                pat[k] = c.low_lexi[k] ? h->repeatStringAt(0) : revcomp(h->RH_Seq.substr(ss[ss.size() - 2], ss[ss.size() - 1] - ss[ss.size() - 2] + 1));
                hdr[k] = &h->RH_Header;
            }
        });
        par_sort(pat, [](const std::string &x, const std::string &y) { return x < y; });
        par_sort(hdr, [](const std::string *x, const std::string *y) { return *x < *y; });
        for (size_t k = 0; k < pat.size(); k++) if (k == 0 || pat[k] != pat[k - 1]) patternsHash.insert(patternsHash.end(), std::make_pair(pat[k], true))->second = true;
        for (size_t k = 0; k < hdr.size(); k++) if (k == 0 || *hdr[k] != *hdr[k - 1]) readsFound.insert(readsFound.end(), std::make_pair(*hdr[k], true))->second = true;
    }
    hlap();      // 6: patternsHash / readsFound
    if (timing) fprintf(stderr, "[crass_timing] hand-off: tokens %.3f, candidate holders %.3f, groups + patterns %.3f, recruit holders %.3f, text %.3f, mappings dropped %.3f, (streamed pass B +) patternsHash / readsFound %.3f s\n", th[0], th[1], th[2], th[3], th[4], th[5], th[6]);
    g_read_counter_p2 += (int)n;
    time(&tnow);
    std::cout << "\r[crass_singletonFinder]: Processed " << g_read_counter_p2 << " ..." << difftime(tnow, time_start) << " sec" << std::endl;
    if (timing)
        fprintf(stderr, "[crass_timing] searchAndRecruit: %llu reads on %zu device(s)%s; read+parse%s %.3f s, pack %.3f s, device (H2D + pass 1 + merge + pass 2) %.3f s, hand-off %.3f s%s\n",
                (unsigned long long)n, devs.size(), streamed ? ", streamed ingest" : indexed ? ", indexed ingest" : "", (streamed || indexed) ? "+pack" : "", t1 - t0, t2 - t1, t3 - t2, now() - t3,
                streamed ? (" (of which the second pass over the inputs " + std::to_string(t_fill) + " s)").c_str()
                         : indexed ? (" (of which the handed-on records' text " + std::to_string(t_fill) + " s)").c_str() : "");
    if (timing)
        fprintf(stderr, "[crass_timing] searchAndRecruit: resident set after ingest %.0f MB, with the device context(s) up %.0f MB, at the end %.0f MB\n",
                rss_ingested, rss_ctx, rss_mb());
    // the search stage's state is freed on a thread of its own, beside a later stage: unmapping an 8 GB input, freeing the index
    // and the device context(s) was 0.3 s between this function's last line and its caller's next one
    {
        // (the index goes here and now, its gigabytes handed back over sixteen threads — crass_fastx_index_free, ~20 ms for the 2.3 GB
        // of a 50 M-read job — not by ONE thread at the process's end (0.2 s) or beside a later stage.  The device is idle: every
        // result has been fetched)
        const double tf0 = now();
        crass_fastx_index_free(IX.ix); IX.ix = nullptr;
        if (timing) fprintf(stderr, "[crass_timing] searchAndRecruit: index freed %.3f s\n", now() - tf0);
        crass_fastx_index *ix = nullptr;
        Made &m = dev.get();
        crass_hip_ctx *cx = m.c; crass_hip_group *gx = m.g;
        m.c = nullptr; m.g = nullptr;
        // (they wait for a stage that does not use the device — buildGraphsAndOutput, the next search, or the process's end:
        // a context destroyed, or gigabytes that had been the source of a host-to-device copy unmapped, while the consensus stage
        // runs made that stage's device work wait 0.3 s, whichever of its steps was in flight)
        g_dead.park(cx, gx, ix);
    }
    return max_len;
}

int findConsensusDRs(ReadMap &mReads, StringCheck &mStringCheck, DR_Cluster_Map &mDR2GIDMap, std::map<int, std::string> &mTrueDRs,
                     GroupKmerMap &groupKmerCountsMap, int &nextFreeGID, int mMaxReadLength, const options &opts)
{
    // ---- flatten the hand-off: holders in mReads order (std::map by token, list order inside) ----
    const bool timing = getenv("CRASS_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tc0 = now();
    std::vector<ReadHolder *> holders;
    std::vector<uint64_t> seq_off(1, 0), rec_read, rec_ss_off;
    std::vector<uint8_t> rec_low;
    std::vector<uint32_t> rec_token, rec_nss, ss_pool;
    std::string seqs;
    // (the order: one walk over the map and its lists; sizes, then text and start/stops: over the cores)
    for (auto &kv : mReads) {
        if (!kv.second) continue;
        for (ReadHolder *h : *kv.second) { holders.push_back(h); rec_token.push_back((uint32_t)kv.first); }
    }
    {
        const size_t nh = holders.size();
        rec_read.resize(nh); rec_low.assign(nh, 1);             // RH_Seq is passed as it stands: nothing to orient
        rec_nss.resize(nh); rec_ss_off.resize(nh); seq_off.resize(nh + 1);
        par_ranges(nh, 8192, [&](size_t a, size_t b) {
            for (size_t k = a; k < b; k++) { rec_read[k] = k; seq_off[k + 1] = holders[k]->RH_Seq.size(); rec_nss[k] = (uint32_t)holders[k]->RH_StartStops.size(); }
        });
        uint64_t ss_at = 0;
        for (size_t k = 0; k < nh; k++) { seq_off[k + 1] += seq_off[k]; rec_ss_off[k] = ss_at; ss_at += rec_nss[k]; }
        seqs.resize(seq_off[nh]); ss_pool.resize(ss_at);
        par_ranges(nh, 8192, [&](size_t a, size_t b) {
            for (size_t k = a; k < b; k++) {
                const ReadHolder *h = holders[k];
                memcpy(&seqs[seq_off[k]], h->RH_Seq.data(), h->RH_Seq.size());
                std::copy(h->RH_StartStops.begin(), h->RH_StartStops.end(), ss_pool.begin() + rec_ss_off[k]);
            }
        });
    }
    // tokens 2 .. mNextFreeToken must be dense for the flat token table (they are: StringCheck hands them out in order)
    std::string tok_chars; std::vector<uint64_t> tok_off(1, 0);
    const int n_tok = mStringCheck.mNextFreeToken - 1;
    for (int t = 2; t <= mStringCheck.mNextFreeToken; t++) { tok_chars += mStringCheck.getString(t); tok_off.push_back(tok_chars.size()); }
    // the ORIGINAL groups: the keys of groupKmerCountsMap (WorkHorse.cpp:587-594), GIDs 1 .. nextFreeGID - 1
    std::vector<uint32_t> grp_tokens; std::vector<uint64_t> grp_off(1, 0);
    const int n_groups = nextFreeGID - 1;
    for (int g = 1; g <= n_groups; g++) {
        auto it = mDR2GIDMap.find(g);
        if (it != mDR2GIDMap.end() && it->second && groupKmerCountsMap.count(g)) for (StringToken t : *it->second) grp_tokens.push_back((uint32_t)t);
        grp_off.push_back(grp_tokens.size());
    }
    if (ss_pool.empty()) ss_pool.push_back(0);
    if (grp_tokens.empty()) grp_tokens.push_back(0);
    crass_cons_input in{};
    in.seqs = seqs.data(); in.seq_off = seq_off.data(); in.n_reads = holders.size();
    in.n_rec = holders.size(); in.rec_read = rec_read.data(); in.rec_lowlexi = rec_low.data(); in.rec_token = rec_token.data();
    in.rec_nss = rec_nss.data(); in.rec_ss_off = rec_ss_off.data(); in.ss_pool = ss_pool.data();
    in.n_tokens = (uint32_t)n_tok; in.tok_chars = tok_chars.data(); in.tok_off = tok_off.data();
    in.n_groups = (uint32_t)n_groups; in.grp_tokens = grp_tokens.data(); in.grp_off = grp_off.data();
    in.max_read_len = (uint32_t)mMaxReadLength;
    const crass_params p = to_params(opts);
    crass_cons *h = nullptr;
    const double tc1 = now();
    chk(crass_hip_consensus(&p, device(), &in, &h), "crass_hip_consensus");
    const double tc2 = now();
    crass_cons_view v{};
    chk(crass_hip_consensus_view(h, &v), "crass_hip_consensus_view");
    if (v.error) { crass_hip_consensus_free(h); return 1; }
    struct Lap { bool on; double a, b, c0; std::function<double()> nw; ~Lap() { if (on) fprintf(stderr, "[crass_timing] consensus (adapter): hand-off flattened %.3f s, crass_hip_consensus %.3f s, results written back %.3f s\n", a - c0, b - a, nw() - b); } } lap{timing, tc1, tc2, tc0, now};
    // ---- write the results back into the hand-off state ----
    for (uint32_t t = (uint32_t)n_tok; t < v.n_tokens; t++)
        mStringCheck.addString(std::string(v.tok_chars + v.tok_off[t], (size_t)(v.tok_off[t + 1] - v.tok_off[t])));
    par_ranges((size_t)v.n_rec, 8192, [&](size_t a, size_t b) {
        for (size_t k = a; k < b; k++) {
            ReadHolder *r = holders[k];
            if (!v.rec_alive[k]) { delete r; holders[k] = nullptr; continue; }
            if (v.rec_rc[k]) { r->RH_Seq = revcomp(r->RH_Seq); r->RH_WasLowLexi = !r->RH_WasLowLexi; }     // ReadHolder::reverseComplementSeq
            r->RH_StartStops.assign(v.ss_pool + v.rec_ss_off[k], v.ss_pool + v.rec_ss_off[k] + v.rec_nss[k]);
        }
    });
    for (auto &kv : mReads) { delete kv.second; kv.second = nullptr; }         // (the lists; the holders move)
    for (uint32_t t = 0; t < v.n_tokens; t++) {
        if (!v.tok_has_list[t]) { if (mReads.count((StringToken)t + 2)) mReads[(StringToken)t + 2] = nullptr; continue; }
        ReadList *l = new ReadList();
        for (uint64_t q = v.tokread_off[t]; q < v.tokread_off[t + 1]; q++) l->push_back(holders[v.tokread_idx[q]]);
        mReads[(StringToken)t + 2] = l;
    }
    for (auto &kv : mDR2GIDMap) { delete kv.second; kv.second = nullptr; }     // killed / split / merged groups stay as NULL entries
    mTrueDRs.clear();
    for (uint32_t g = 0; g < v.n_groups; g++) {
        DR_Cluster *c = new DR_Cluster();
        for (uint64_t q = v.grp_off[g]; q < v.grp_off[g + 1]; q++) c->push_back((StringToken)v.grp_tokens[q]);
        mDR2GIDMap[v.grp_gid[g]] = c;
        mTrueDRs[v.grp_gid[g]] = std::string(v.dr_chars + v.dr_off[g], (size_t)(v.dr_off[g + 1] - v.dr_off[g]));
    }
    nextFreeGID = v.next_free_gid;
    for (auto &kv : groupKmerCountsMap) { delete kv.second; kv.second = nullptr; }      // "delete the kmer count lists" (WorkHorse.cpp:603-607)
    crass_hip_consensus_free(h);
    return 0;
}

int buildGraphsAndOutput(ReadMap &mReads, DR_Cluster_Map &mDR2GIDMap, std::map<int, std::string> &mTrueDRs, const options &opts,
                         const std::string &timeStamp, const std::string &commandLine)
{
    // buildGraph's order (WorkHorse.cpp:454-505): groups in ascending GID that have a cluster and a true DR; for every token of
    // the cluster, the token's ReadHolders in list order
    const bool timing = getenv("CRASS_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tb0 = now();
    g_dead.destroy_async();                                // (the search stage's device contexts: this stage runs on the host)
    std::vector<int32_t> gid;
    std::string dr, hdr, com, seq;
    std::vector<uint64_t> dr_off(1, 0), grp_rec_off(1, 0), hdr_off(1, 0), com_off(1, 0), seq_off(1, 0), rec_ss_off;
    std::vector<uint32_t> rec_nss, ss_pool;
    std::vector<const ReadHolder *> flat;
    for (auto &kv : mDR2GIDMap) {
        if (!kv.second) continue;
        auto td = mTrueDRs.find(kv.first);
        if (td == mTrueDRs.end()) continue;
        gid.push_back(kv.first);
        dr += td->second; dr_off.push_back(dr.size());
        for (StringToken t : *kv.second) {
            auto it = mReads.find(t);
            if (it == mReads.end() || !it->second) continue;
            for (ReadHolder *h : *it->second) if (h) flat.push_back(h);
        }
        grp_rec_off.push_back(flat.size());
    }
    {
        // (the order above: one walk over the maps and lists; sizes, then the text: over the cores)
        const size_t nh = flat.size();
        hdr_off.resize(nh + 1); com_off.resize(nh + 1); seq_off.resize(nh + 1); rec_nss.resize(nh); rec_ss_off.resize(nh);
        par_ranges(nh, 8192, [&](size_t a, size_t b) {
            for (size_t k = a; k < b; k++) {
                const ReadHolder *h = flat[k];
                hdr_off[k + 1] = h->RH_Header.size(); com_off[k + 1] = h->RH_Comment.size(); seq_off[k + 1] = h->RH_Seq.size();
                rec_nss[k] = (uint32_t)h->RH_StartStops.size();
            }
        });
        uint64_t ss_at = 0;
        for (size_t k = 0; k < nh; k++) { hdr_off[k + 1] += hdr_off[k]; com_off[k + 1] += com_off[k]; seq_off[k + 1] += seq_off[k]; rec_ss_off[k] = ss_at; ss_at += rec_nss[k]; }
        hdr.resize(hdr_off[nh]); com.resize(com_off[nh]); seq.resize(seq_off[nh]); ss_pool.resize(ss_at);
        par_ranges(nh, 8192, [&](size_t a, size_t b) {
            for (size_t k = a; k < b; k++) {
                const ReadHolder *h = flat[k];
                memcpy(&hdr[hdr_off[k]], h->RH_Header.data(), h->RH_Header.size());
                memcpy(&com[com_off[k]], h->RH_Comment.data(), h->RH_Comment.size());
                memcpy(&seq[seq_off[k]], h->RH_Seq.data(), h->RH_Seq.size());
                std::copy(h->RH_StartStops.begin(), h->RH_StartStops.end(), ss_pool.begin() + rec_ss_off[k]);
            }
        });
    }
    if (ss_pool.empty()) ss_pool.push_back(0);
    if (com.empty()) com.push_back('\0');
    if (hdr.empty()) hdr.push_back('\0');
    if (seq.empty()) seq.push_back('\0');
    if (dr.empty()) dr.push_back('\0');
    rec_nss.push_back(0); rec_ss_off.push_back(0); gid.push_back(0);      // (never empty arrays)
    crass_graph_input in{};
    in.n_groups = (uint32_t)(dr_off.size() - 1); in.gid = gid.data(); in.dr_chars = dr.data(); in.dr_off = dr_off.data();
    in.grp_rec_off = grp_rec_off.data(); in.n_rec = hdr_off.size() - 1;
    in.hdr_chars = hdr.data(); in.hdr_off = hdr_off.data(); in.com_chars = com.data(); in.com_off = com_off.data();
    in.seq_chars = seq.data(); in.seq_off = seq_off.data(); in.rec_nss = rec_nss.data(); in.rec_ss_off = rec_ss_off.data(); in.ss_pool = ss_pool.data();
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd)) cwd[0] = 0;
    crass_output_opts oo{};
    oo.out_dir = opts.output_fastq.c_str(); oo.timestamp = timeStamp.c_str(); oo.command_line = commandLine.c_str(); oo.cwd = cwd;
    oo.log_to_screen = opts.logToScreen ? 1 : 0; oo.cov_cutoff = opts.covCutoff; oo.node_kmer = opts.cNodeKmerLength;
    oo.show_singles = opts.showSingles ? 1 : 0; oo.long_description = opts.longDescription ? 1 : 0;
    crass_outputs *res = nullptr;
    const double tb1 = now();
    const int rc = crass_build_outputs(&in, &oo, &res);
    if (rc != CRASS_OK) { std::cerr << "[ERROR]: building the spacer graphs failed: " << crass_hip_strerror(rc) << std::endl; return -1; }
    const double tb2 = now();
    crass_outputs_view v{};
    (void)crass_outputs_get(res, &v);
    std::cout << v.stdout_text << std::flush;
    const int wrc = crass_outputs_write(res, opts.output_fastq.c_str());
    if (timing) fprintf(stderr, "[crass_timing] outputs (adapter): hand-off flattened %.3f s, crass_build_outputs %.3f s, files written %.3f s\n", tb1 - tb0, tb2 - tb1, now() - tb2);
    const int n = (int)v.n_groups_kept;
    g_reaper_out.wait();
    g_reaper_out.f = std::async(std::launch::async, [res] { crass_outputs_free(res); });      // (waits for the graphs' tear-down: not on this thread)
    if (wrc != CRASS_OK) { std::cerr << "[ERROR]: cannot write the output files to " << opts.output_fastq << std::endl; return -1; }
    return n;
}

} // namespace crass_hip
