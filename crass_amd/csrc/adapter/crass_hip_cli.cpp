// crass_hip_cli.cpp — `crass-hip`: crass's command line for the search stage, running on the
// MI355X engine through the adapter.  Mirrors the call sequence of WorkHorse::parseSeqFiles
// (WorkHorse.cpp:321-414) and the stdout tags crass prints; the options are the subset of
// crass's getopt string that the search path reads (-d -D -s -S -w -n -k -l -g -o; crass.cpp:198).
// Behind the search it runs crass's remaining stages over the hand-off — true DRs (findConsensusDRs), spacer graphs and
// the output files of WorkHorse::outputResults: <outdir>crass.crispr, Group_<gid>_<DR>.fa, Spacers_*_spacers.gv,
// crass.<timestamp>.keys.gv (and crass.<timestamp>.log unless -g) — so that it drops in for the crass binary.  With
// --dump-handoff it also writes the hand-off itself — every ReadHolder of mReads, the token table, the DR groups and the
// pattern list — as line-oriented text (<outdir>crass_hip_handoff.tsv / crass_hip_consensus.tsv) that the parity tests
// diff against the oracle.
#include "crass_adapter.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <getopt.h>
#include <iostream>
#include <chrono>
#include <unistd.h>
#include <dirent.h>
#include <fcntl.h>
#include <thread>
#include <sys/resource.h>
#include <algorithm>
#include <vector>
#include <cstdint>
#include <map>
#include <string>

using namespace crass_hip;

static void usage()
{
    std::cout << "crass-hip (MI355X search stage of crass 1.0.1)\n"
                 "usage: crass-hip [-d minDR] [-D maxDR] [-s minSpacer] [-S maxSpacer] [-w window] [-n minRepeats]\n"
                 "                 [-k kmerClust] [-f covCutoff] [-K nodeKmer] [-G] [-L] [-l logLevel] [-g] [-o outdir]\n"
                 "                 [--gpus N | --devices a,b,..] [--seam] [--dump-handoff]\n"
                 "                 <reads.f[aq][.gz]> ...\n"
                 "  --gpus N       shard the reads over GPUs 0..N-1 (one RCCL all-gather of the candidate DR strings per job)\n"
                 "  --devices L    the same with an explicit device list\n"
                 "  --seam         drive the engine through crass's three calls (searchFile / createNonRedundantSet /\n"
                 "                 findSingletons: the DR merge then runs on the host) instead of the one-call device path\n";
}

// CRASS_SMAPS_AT_EXIT=1: the mappings with the largest resident sets, at the end of every stage (what the kernel takes back at _exit)
static void dump_smaps(const char *tag)
{
    if (!getenv("CRASS_SMAPS_AT_EXIT")) return;
    FILE *f = fopen("/proc/self/smaps", "r");
    if (!f) return;
    struct Reg { std::string name; long rss = 0, huge = 0, size = 0; };
    std::vector<Reg> regs;
    char line[512];
    while (fgets(line, sizeof(line), f)) {
        unsigned long a, b;
        if (sscanf(line, "%lx-%lx ", &a, &b) == 2 && !strstr(line, "kB")) { regs.emplace_back(); regs.back().name = line; regs.back().name.erase(regs.back().name.find_last_not_of("\n") + 1); }
        else if (!regs.empty()) {
            if (!strncmp(line, "Rss:", 4)) regs.back().rss = atol(line + 4);
            else if (!strncmp(line, "Size:", 5)) regs.back().size = atol(line + 5);
            else if (!strncmp(line, "AnonHugePages:", 14)) regs.back().huge = atol(line + 14);
        }
    }
    fclose(f);
    std::sort(regs.begin(), regs.end(), [](const Reg &x, const Reg &y) { return x.rss > y.rss; });
    long tot = 0, huge = 0, n50 = 0, r50 = 0;
    for (auto &r : regs) { tot += r.rss; huge += r.huge; if (r.size >= (40 << 10) && r.size <= (65 << 10)) { n50++; r50 += r.rss; } }
    fprintf(stderr, "[crass_timing] smaps at %s: %zu mappings, %ld MB resident (%ld MB in huge pages); %ld mappings of 40..65 MB hold %ld MB\n", tag, regs.size(), tot >> 10, huge >> 10, n50, r50 >> 10);
    {   // (by size class: count, resident MB)
        std::map<long, std::pair<long, long>> hist;
        for (auto &r : regs) { long c = 1; while (c < (r.size >> 10)) c <<= 1; hist[c].first++; hist[c].second += r.rss; }
        std::string h;
        for (auto &kv : hist) if (kv.second.second >> 10) h += " <=" + std::to_string(kv.first) + "MB:" + std::to_string(kv.second.first) + "x/" + std::to_string(kv.second.second >> 10) + "MB";
        fprintf(stderr, "[crass_timing] smaps %s by size class (count / resident):%s\n", tag, h.c_str());
    }
    if (getenv("CRASS_SMAPS_PEEK")) {
        int shown = 0;
        for (auto &r : regs) {
            if (r.huge || r.rss < (12 << 10) || r.name.find("rw-p") == std::string::npos || shown >= 24) continue;
            unsigned long a = 0, b = 0;
            sscanf(r.name.c_str(), "%lx-%lx", &a, &b);
            const uint32_t *w = (const uint32_t *)a;
            const size_t nw = (b - a) / 4;
            std::string h;
            char buf[32];
            for (size_t off : {(size_t)0, (size_t)1024, nw / 2, nw - 1024}) { h += " |"; for (int k = 0; k < 8; k++) { snprintf(buf, sizeof(buf), " %08x", w[off + k]); h += buf; } }
            fprintf(stderr, "[crass_timing] peek %s %ld MB:%s\n", r.name.substr(0, 25).c_str(), r.size >> 10, h.c_str());
            shown++;
        }
    }
    for (size_t i = 0; i < regs.size() && i < 6; i++) fprintf(stderr, "[crass_timing] smaps %s: rss %6ld MB (huge %6ld MB) of %6ld MB  %s\n", tag, regs[i].rss >> 10, regs[i].huge >> 10, regs[i].size >> 10, regs[i].name.c_str());
}

// glibc's malloc backs its arenas with 2 MB pages only when the tunable glibc.malloc.hugetlb is set, and tunables are read when the
// process starts: half a million ReadHolders, the groups' graphs and every temporary string live in those arenas, and with them in
// huge pages a 50 M-read run takes 1.31-1.38 s instead of 1.55-1.7 (every host stage: fewer first-touch faults, fewer TLB misses;
// profiles/NOTES_r06.md 6b).  So the command line starts itself again ONCE with the tunable — as its very first act, and only from a
// process that has nothing to do with a GPU yet: no tool library preloaded (a profiler's initialises the device before main), no
// device file open.  CRASS_NO_REEXEC=1 keeps the process as it was started.
extern char **environ;
static void restart_with_huge_malloc(char **argv)
{
    const char *t = getenv("GLIBC_TUNABLES");
    if (getenv("CRASS_NO_REEXEC") || (t && strstr(t, "glibc.malloc.hugetlb"))) return;
    for (char **e = environ; e && *e; e++)
        if (!strncmp(*e, "LD_PRELOAD=", 11) || !strncmp(*e, "HSA_TOOLS_LIB=", 14) || !strncmp(*e, "ROCP", 4) || !strncmp(*e, "ROCTRACER", 9) || !strncmp(*e, "LD_AUDIT=", 9)) return;
    if (FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r")) {
        char line[128] = {0};
        const bool never = fgets(line, sizeof(line), f) && strstr(line, "[never]");
        fclose(f);
        if (never) return;
    } else return;
    if (DIR *d = opendir("/proc/self/fd")) {               // (a device file open already: this process is not ours alone — never exec from it)
        bool dev = false;
        while (struct dirent *e = readdir(d)) {
            char path[300], link[300];
            snprintf(path, sizeof(path), "/proc/self/fd/%s", e->d_name);
            const ssize_t n = readlink(path, link, sizeof(link) - 1);
            if (n > 0) { link[n] = 0; if (strstr(link, "/dev/kfd") || strstr(link, "/dev/dri")) dev = true; }
        }
        closedir(d);
        if (dev) return;
    } else return;
    const std::string v = (t && *t) ? std::string(t) + ":glibc.malloc.hugetlb=1" : std::string("glibc.malloc.hugetlb=1");
    setenv("GLIBC_TUNABLES", v.c_str(), 1);
    execv("/proc/self/exe", argv);
    // (only reached when the exec did not happen: the process goes on as it was started)
}

int main(int argc, char *argv[])
{
    restart_with_huge_malloc(argv);
    // The descriptor table is grown ONCE, now, while this process has a single thread: later every doubling of it (the HIP runtime's
    // device and event descriptors, a hundred output files) waits for an RCU grace period, 30 .. 50 ms each, because the table is
    // shared with other threads by then (fs/file.c expand_fdtable)
    { const int hi = fcntl(0, F_DUPFD, 4000); if (hi >= 0) close(hi); }
    options opts;
    static struct option long_options[] = {
        {"minDR", required_argument, nullptr, 'd'}, {"maxDR", required_argument, nullptr, 'D'},
        {"minSpacer", required_argument, nullptr, 's'}, {"maxSpacer", required_argument, nullptr, 'S'},
        {"windowLength", required_argument, nullptr, 'w'}, {"minNumRepeats", required_argument, nullptr, 'n'},
        {"kmerCount", required_argument, nullptr, 'k'}, {"logLevel", required_argument, nullptr, 'l'},
        {"logToScreen", no_argument, nullptr, 'g'}, {"outDir", required_argument, nullptr, 'o'},
        {"gpus", required_argument, nullptr, 1001}, {"devices", required_argument, nullptr, 1002}, {"seam", no_argument, nullptr, 1003},
        {"local-copies", no_argument, nullptr, 1004}, {"dump-handoff", no_argument, nullptr, 1005}, {"timestamp", required_argument, nullptr, 1006},
        {"covCutoff", required_argument, nullptr, 'f'}, {"kmerNodes", required_argument, nullptr, 'K'}, {"showSingltons", no_argument, nullptr, 'G'},
        {"longDescription", no_argument, nullptr, 'L'},
        {"help", no_argument, nullptr, 'h'}, {nullptr, 0, nullptr, 0}};
    std::vector<int> devices;
    bool seam = false, local_copies = false, dump_handoff = false;
    std::string timestamp;
    int c, idx = 0;
    while ((c = getopt_long(argc, argv, "d:D:s:S:w:n:k:l:go:hf:K:GL", long_options, &idx)) != -1) {
        switch (c) {
            case 'd':
                opts.lowDRsize = (unsigned)atoi(optarg);
                if (opts.lowDRsize < 8) {      // crass.cpp:264-271
                    std::cerr << "crass [WARNING]: The lower bound for direct repeat sizes cannot be " << opts.lowDRsize << " changing to 23" << std::endl;
                    opts.lowDRsize = 23;
                }
                break;
            case 'D': opts.highDRsize = (unsigned)atoi(optarg); break;
            case 's':
                opts.lowSpacerSize = (unsigned)atoi(optarg);
                if (opts.lowSpacerSize < 8) {  // crass.cpp:351-358
                    std::cerr << "crass [WARNING]: The lower bound for spacer sizes cannot be " << opts.lowSpacerSize << " changing to 26" << std::endl;
                    opts.lowSpacerSize = 26;
                }
                break;
            case 'S': opts.highSpacerSize = (unsigned)atoi(optarg); break;
            case 'w':
                opts.searchWindowLength = (unsigned)atoi(optarg);
                if (opts.searchWindowLength < 6 || opts.searchWindowLength > 9) {   // crass.cpp:366-374
                    std::cerr << "crass [WARNING]: Specified window length higher than max. Changing window length to 8 instead of " << opts.searchWindowLength << std::endl;
                    opts.searchWindowLength = 8;
                }
                break;
            case 'n':
                opts.minNumRepeats = (unsigned)atoi(optarg);
                if (opts.minNumRepeats < 2) {  // crass.cpp:316-324
                    std::cerr << "crass [ERROR]: The mininum number of repeats cannot be less than 2" << std::endl;
                    return 1;
                }
                break;
            case 'k':
                opts.kmer_clust_size = atoi(optarg);
                if (opts.kmer_clust_size < 4) {  // crass.cpp:294-301
                    std::cerr << "crass [WARNING]: Minimum value for kmer clustering size is: 4 changing to 6" << std::endl;
                    opts.kmer_clust_size = 6;
                }
                break;
            case 'l': opts.logLevel = atoi(optarg); break;
            case 'g': opts.logToScreen = true; break;
            case 'o':
                opts.output_fastq = optarg;
                if (opts.output_fastq[opts.output_fastq.length() - 1] != '/') opts.output_fastq += '/';
                break;
            case 1001: {
                const int ng = atoi(optarg);
                if (ng < 1 || ng > 64) { std::cerr << "crass [ERROR]: --gpus needs a device count between 1 and 64" << std::endl; return 1; }
                devices.clear();
                for (int d = 0; d < ng; d++) devices.push_back(d);
                break;
            }
            case 1002: {
                devices.clear();
                for (const char *q = optarg; *q;) { devices.push_back(atoi(q)); while (*q && *q != ',') q++; if (*q == ',') q++; }
                if (devices.empty()) { std::cerr << "crass [ERROR]: --devices needs a comma-separated device list" << std::endl; return 1; }
                break;
            }
            case 1003: seam = true; break;
            case 1004: local_copies = true; break;      // (tests: several contexts on one GPU)
            case 1005: dump_handoff = true; break;
            case 1006: timestamp = optarg; break;       // (tests: a fixed mTimeStamp)
            case 'f': opts.covCutoff = atoi(optarg); break;                // crass.cpp:280-282
            case 'K': opts.cNodeKmerLength = atoi(optarg); break;          // crass.cpp:302-304
            case 'G': opts.showSingles = true; break;
            case 'L': opts.longDescription = true; break;
            case 'h': usage(); return 0;
            default: usage(); return 1;
        }
    }
    if (opts.lowDRsize >= opts.highDRsize) {      // crass.cpp:388-393
        std::cerr << "crass [ERROR]: The lower direct repeat bound is bigger than the higher bound (" << opts.lowDRsize << " >= " << opts.highDRsize << ")" << std::endl;
        return 1;
    }
    if (opts.lowSpacerSize >= opts.highSpacerSize) {
        std::cerr << "crass [ERROR]: The lower spacer bound is bigger than the higher bound (" << opts.lowSpacerSize << " >= " << opts.highSpacerSize << ")" << std::endl;
        return 1;
    }
    if (opts.lowDRsize < 2 * opts.searchWindowLength - 1) {
        // The reference accepts this (crass.cpp:264-271 only rejects -d < 8), but its seed stride `unsigned skips = lowDR -
        // (2w - 1)` then wraps to ~4e9 (libcrispr.cpp:281-285): after a rejected candidate `j = back() - 1 + skips` walks
        // BACKWARDS and the search need not terminate.  The engine refuses such parameters (CRASS_ERR_UNSUPPORTED); the
        // command line warns in crass's style and uses the smallest well-defined bound instead.
        std::cerr << "crass [WARNING]: The lower bound for direct repeat sizes (" << opts.lowDRsize << ") is smaller than 2 * windowLength - 1 ("
                  << 2 * opts.searchWindowLength - 1 << "): the seed stride of the search is not defined for it; changing to "
                  << 2 * opts.searchWindowLength - 1 << std::endl;
        opts.lowDRsize = 2 * opts.searchWindowLength - 1;
        if (opts.lowDRsize >= opts.highDRsize) { std::cerr << "crass [ERROR]: The lower direct repeat bound is bigger than the higher bound" << std::endl; return 1; }
    }
    if (optind >= argc) { std::cerr << "crass [ERROR]: No input files were provided. Try ./crass-hip -h for help." << std::endl; return 1; }
    std::vector<std::string> seqFiles(argv + optind, argv + argc);
    if (timestamp.empty()) {                    // crass.cpp:471-479
        char buffer[80];
        time_t rawtime; time(&rawtime);
        strftime(buffer, 80, "%d_%m_%Y_%H%M%S", localtime(&rawtime));
        timestamp = buffer;
    }
    std::string cmd_line;                       // crass.cpp:508-512
    for (int i = 0; i < argc; ++i) { cmd_line += argv[i]; cmd_line += ' '; }

    // This process runs every stage ONCE: its host pool may be as wide as the box's CPU quota allows (the engine's default — half the
    // quota — is for a caller that steps in a loop with the pool's workers polling between the steps; the output stage's per-group
    // work went from 0.43 to 0.33 s at 50 M reads)
    setenv("CRASS_HOST_THREADS", std::to_string(std::min(16u, std::max(1u, std::thread::hardware_concurrency()))).c_str(), 0);
    ReadMap mReads;
    StringCheck mStringCheck;
    DR_Cluster_Map mDR2GIDMap;
    std::map<int, bool> mGroupMap;
    GroupKmerMap group_kmer_counts_map;
    lookupTable patterns_lookup, reads_found;
    int mMaxReadLength = 0;
    int rc = 0;
    const bool timing = getenv("CRASS_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_main = now();
    double t_prev = t_main;
    // (peak resident set so far, VmHWM of /proc/self/status: where the memory of a run goes, stage by stage)
    auto peak_rss_mb = [] {
        long kb = 0;
        if (FILE *f = fopen("/proc/self/status", "r")) {
            char line[256];
            while (fgets(line, sizeof(line), f)) if (!strncmp(line, "VmHWM:", 6)) { kb = atol(line + 6); break; }
            fclose(f);
        }
        return kb / 1024.0;
    };
    // (wall-clock stamps of main's first and last line: what the process spends before and after them — loader, static
    // constructors; address-space and KFD tear-down — is the difference to the parent's own clock, tools/e2e_big.py)
    auto epoch = [] { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); };
    if (timing) fprintf(stderr, "[crass_timing] cli: main entered at epoch %.6f\n", epoch());
    if (timing) fprintf(stderr, "[crass_timing] cli: GLIBC_TUNABLES=%s\n", getenv("GLIBC_TUNABLES") ? getenv("GLIBC_TUNABLES") : "(unset)");
    // (CPU seconds of the whole process, user + system: under a cgroup CPU quota — the GPU boxes give 16 CPUs per 100 ms — a stage's
    // wall time is its CPU seconds / 16 however many threads it starts)
    auto cpu_s = [](double *sys) {
        struct rusage ru;
        getrusage(RUSAGE_SELF, &ru);
        if (sys) *sys = ru.ru_stime.tv_sec + 1e-6 * ru.ru_stime.tv_usec;
        return ru.ru_utime.tv_sec + 1e-6 * ru.ru_utime.tv_usec;
    };
    // (... and how often the quota froze the cgroup: periods in which it ran dry, cpu.stat's nr_throttled)
    auto throttled = [] {
        long n = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.stat", "r")) {
            char line[128];
            while (fgets(line, sizeof(line), f)) if (!strncmp(line, "nr_throttled ", 13)) { n = atol(line + 13); break; }
            fclose(f);
        }
        return n;
    };
    double cpu_prev_u = 0, cpu_prev_s = 0;
    long thr_prev = timing ? throttled() : 0;
    auto lap = [&](const char *what) {
        if (!timing) return;
        const double t = now();
        double sy = 0; const double us = cpu_s(&sy);
        const long th = throttled();
        fprintf(stderr, "[crass_timing] cli: %-28s %.3f s (at %.3f s, peak RSS %.0f MB; CPU %.2f s user + %.2f s system, %ld quota period(s) ran dry)\n", what, t - t_prev, t - t_main, peak_rss_mb(),
                us - cpu_prev_u, sy - cpu_prev_s, th - thr_prev);
        t_prev = t; cpu_prev_u = us; cpu_prev_s = sy; thr_prev = th;
    };
    try {
        time_t start_time; time(&start_time);
        if (!devices.empty()) setDevices(devices, local_copies);
        // (this process ends behind its outputs: see the end of main.  CRASS_TEARDOWN=async: the search stage's contexts and index
        // are handed back by a thread beside the output stage — which runs on the host — instead of by the kernel at _exit)
        const char *td_mode = getenv("CRASS_TEARDOWN");
        if (!getenv("CRASS_RELEASE_AT_EXIT") && !(td_mode && !strcmp(td_mode, "async"))) leaveTeardownToProcessEnd(true);
        int next_free_GID = 1;
        Vecstr *nr = nullptr;
        if (seam) {
            for (const auto &f : seqFiles) {
                int max_len = searchFile(f.c_str(), opts, &mReads, &mStringCheck, patterns_lookup, reads_found, start_time);
                mMaxReadLength = std::max(mMaxReadLength, max_len);
            }
            std::cout << std::endl;
            lap("searchFile (ingest + pass 1)");
            nr = createNonRedundantSet(mReads, mStringCheck, mDR2GIDMap, mGroupMap, group_kmer_counts_map, next_free_GID, opts);
            if (nr->size() > 0) {
                std::cout << "[crass_clusterCore]: " << nr->size() << " non-redundant patterns." << std::endl;
                time(&start_time);
                for (const auto &f : seqFiles) findSingletons(f.c_str(), opts, nr, reads_found, &mReads, &mStringCheck, start_time);
            }
            std::cout << std::endl;
            lap("merge + findSingletons");
        } else {
            nr = new Vecstr();
            mMaxReadLength = searchAndRecruit(seqFiles, opts, &mReads, &mStringCheck, mDR2GIDMap, mGroupMap, group_kmer_counts_map, next_free_GID,
                                              patterns_lookup, reads_found, nr, start_time);
            lap("searchAndRecruit (ingest + pass 1 + merge + pass 2)");
            dump_smaps("search done");
        }
        size_t n_reads = 0;
        for (auto &kv : mReads) n_reads += kv.second->size();
        std::cout << "[crass_patternFinder]: Found " << n_reads << " reads" << std::endl;

        // ---- hand-off dump ----
        if (dump_handoff) {
        std::ofstream out((opts.output_fastq + "crass_hip_handoff.tsv").c_str());
        out << "#max_read_length\t" << mMaxReadLength << "\n#next_free_GID\t" << next_free_GID << "\n";
        for (auto &kv : mStringCheck.mT2S_map) out << "T\t" << kv.first << "\t" << kv.second << "\n";
        for (auto &kv : mDR2GIDMap) { out << "G\t" << kv.first; for (StringToken t : *kv.second) out << "\t" << t; out << "\n"; }
        for (auto &kv : group_kmer_counts_map) for (auto &kc : *kv.second) out << "K\t" << kv.first << "\t" << kc.first << "\t" << kc.second << "\n";
        for (const auto &p : *nr) out << "P\t" << p << "\n";
        for (auto &kv : mReads)
            for (ReadHolder *h : *kv.second) {
                out << "R\t" << kv.first << "\t" << h->RH_Header << "\t" << (h->RH_WasLowLexi ? 1 : 0) << "\t" << h->RH_RepeatLength << "\t" << (h->RH_IsFasta ? 1 : 0) << "\t";
                for (size_t i = 0; i < h->RH_StartStops.size(); i++) out << (i ? "," : "") << h->RH_StartStops[i];
                out << "\t" << h->RH_Seq << "\t" << h->RH_Comment << "\t" << h->RH_Qual << "\n";
            }
        out.close();
        lap("hand-off dump");
        }
        delete nr;
        // ---- the stage behind the search: true DRs + repaired start/stops (WorkHorse.cpp:403) ----
        std::map<int, std::string> mTrueDRs;
        if (findConsensusDRs(mReads, mStringCheck, mDR2GIDMap, mTrueDRs, group_kmer_counts_map, next_free_GID, mMaxReadLength, opts)) {
            std::cerr << "[ERROR]: Wierd stuff happend when trying to get the 'true' direct repeat" << std::endl;     // WorkHorse.cpp:405
            rc = 2;
        } else {
            if (dump_handoff) {
            std::ofstream con((opts.output_fastq + "crass_hip_consensus.tsv").c_str());
            con << "#next_free_GID\t" << next_free_GID << "\n";
            for (auto &kv : mStringCheck.mT2S_map) con << "T\t" << kv.first << "\t" << kv.second << "\n";
            for (auto &kv : mTrueDRs) con << "D\t" << kv.first << "\t" << kv.second << "\n";
            for (auto &kv : mDR2GIDMap) { if (!kv.second) continue; con << "G\t" << kv.first; for (StringToken t : *kv.second) con << "\t" << t; con << "\n"; }
            for (auto &kv : mReads) {
                if (!kv.second) continue;
                for (ReadHolder *h : *kv.second) {
                    con << "R\t" << kv.first << "\t" << h->RH_Header << "\t" << (h->RH_WasLowLexi ? 1 : 0) << "\t";
                    for (size_t i = 0; i < h->RH_StartStops.size(); i++) con << (i ? "," : "") << h->RH_StartStops[i];
                    con << "\t" << h->RH_Seq << "\n";
                }
            }
            }
            std::cout << "[crass_consensus]: " << mTrueDRs.size() << " true direct repeats" << std::endl;
            lap("findConsensusDRs + dump");
            dump_smaps("consensus done");
            // ---- spacer graphs + crass's output files (WorkHorse.cpp:196-316) ----
            if (!opts.logToScreen) {                // the log file crass.cpp:484-496 opens; <file type="log"> of the XML names it
                std::ofstream lg((opts.output_fastq + "crass." + timestamp + ".log").c_str());
                lg << "crass-hip (MI355X search stage) " << cmd_line << "\n" << n_reads << " reads found, " << mTrueDRs.size() << " true direct repeats\n";
            }
            if (buildGraphsAndOutput(mReads, mDR2GIDMap, mTrueDRs, opts, timestamp, cmd_line) < 0) rc = 2;
            lap("spacer graphs + outputs");
        }
    } catch (std::exception &e) {
        std::cerr << e.what() << std::endl;
        rc = 2;            // doWork -> 2 -> process exit code 2
    }
    if (!getenv("CRASS_RELEASE_AT_EXIT")) {
        // the process ends here: handing 10^5..10^7 heap objects, the resident reads and the HIP context back one by one cost
        // 0.17 s of a 0.9 s run (profiles/r02_e2e_cli.txt) for memory the kernel reclaims anyway; CRASS_RELEASE_AT_EXIT=1 keeps
        // the orderly tear-down (leak checks)
        std::cout.flush(); std::cerr.flush(); fflush(nullptr);
        lap("exit without tear-down");
        dump_smaps("exit");
        if (timing) { fprintf(stderr, "[crass_timing] cli: _exit called at epoch %.6f\n", epoch()); fflush(stderr); }
        _exit(rc);
    }
    releaseDeviceReads();
    clearReadMap(&mReads);
    for (auto &kv : mDR2GIDMap) delete kv.second;
    for (auto &kv : group_kmer_counts_map) delete kv.second;
    lap("release");
    return rc;
}
