// engine_internal.h — types shared by the HIP kernels (kernels.hip) and the host engine
// (engine.cpp).  Not part of the public ABI (include/crass_hip.h is).
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <hip/hip_runtime.h>
#include "devmem.h"
#include "merge.h"

namespace crass {

// device view of the resident reads (all pointers are device pointers)
struct DevReads {
    const uint32_t *packed;
    const uint64_t *word_off;     // nullptr when stride_words > 0
    const uint32_t *lengths;      // nullptr when uniform_len > 0
    const uint64_t *header_id;    // nullptr: unique headers
    const uint32_t *exc_mask;     // bit per read, 1 = exception read (never nullptr; may be all zero)
    uint64_t n_reads;
    uint32_t stride_words;
    uint32_t uniform_len;
    // exception reads (raw bytes)
    const uint64_t *exc_read;
    const uint64_t *exc_off;
    const uint8_t  *exc_bytes;
    uint64_t n_exc;
    // long reads (no per-read filter): one hint bit per base position, bit p clear => the seed at p provably has
    // no copy in its search window (k_hint_positions); read r's bits start at word pos_hint_off[r].  May be nullptr.
    const uint64_t *pos_hint;
    const uint64_t *pos_hint_off;
    // pass 2 walks one read per WAVE (lane = window) instead of one per lane: sets whose longest read is beyond ~800 bases
    // (a lane walking its own long read touches one word per row; below that a wave's 256 windows per round stay mostly empty)
    uint32_t wave_walk;
    // pos_hint holds a bit for EVERY position (the every-position form, another window or seed lattice: k_hint_filter_any), not
    // the default lattice's residue class 0 with the others filled in by the walking wave
    uint32_t hint_all;
};

struct DevParams {
    uint32_t lowDR, highDR, lowSp, highSp, window, minRepeats;
    uint32_t skips;               // lowDR - (2w-1) as unsigned, clamped to >=1 only when 0 (libcrispr.cpp:281-285)
    uint32_t debug_stop;          // diagnostics only (env CRASS_SURV_DEBUG): 0 = normal; 1..3 cut the survivor kernel short
    uint32_t exc_survive;         // 1: the filter passes every exception read on (they join the survivor list and are
                                  // evaluated byte-wise in place, so the dense pass-1 path also holds with N reads)
    unsigned long long *prof;     // diagnostics only (env CRASS_SURV_PROF): 192 counters of the wave kernel's phases, else nullptr
};

// per-survivor output slot of the pass-1 survivor kernel
struct SurvOut {
    uint32_t found;               // searchCore returned true
    uint32_t n_ss;
    uint32_t repeat_len;
    uint32_t ss_off;              // offset into the ss pool
    uint16_t dr_len;
    uint8_t  low_lexi;
    uint8_t  err;                 // 1: reference would throw, 2: start/stop capacity, 3: pool overflow,
                                  // 4: punted by the lane-per-read kernel to the wave-per-read kernel
};

// per-hit output of the pass-2 finish kernel
struct RecruitOut {
    uint32_t start, end;
    uint32_t token;               // StringToken of the low-lexi DR when the device knows it, else 0
    uint16_t dr_len;
    uint8_t  low_lexi;
    uint8_t  pad;
};

// pass-2 automaton (byte-wise Aho-Corasick, goto fully resolved)
struct DevAutomaton {
    const uint16_t *go16;         // [n_states][n_sym1] when n_states <= 65535 (uploaded on demand: full scans and
    const uint32_t *go32;         // otherwise                                   exception reads only)
    const uint16_t *out_len;      // [n_states] longest pattern ending at the state (0 = none)
    const uint32_t *out_pid;      // [n_states] index of that pattern in the pattern list
    const uint16_t *go4;          // [n_states][4] ACGT-only compact table (packed reads), may be nullptr
    const uint32_t *go4w;         // the same with 32-bit entries when n_states > 65535, may be nullptr
    uint32_t n_states;
    uint32_t n_sym1;              // symbols + 1 (symbol 0 = byte in no pattern)
    uint8_t  sym[256];            // byte -> symbol
    uint32_t acgt_ok;             // n_states <= 65535 (go4 valid)
    uint32_t max_pat_len;         // longest pattern (= depth of the trie): a scan that starts this many bases before a
                                  // position is in the exact state there
};

// tab_mode 2's Bloom filter (2^20 bits = 2^15 words in LDS), blocked: word = the hash's top 15 bits, two bits inside the word from
// its bits 12..16 and 7..11 (a shift by a register uses the low five bits of the register: no masking)
static __host__ __device__ __forceinline__ uint32_t ak_bloom_word(uint32_t h) { return h >> 17; }
static __host__ __device__ __forceinline__ uint32_t ak_bloom_bits(uint32_t h) { return (1u << ((h >> 12) & 31u)) | (1u << ((h >> 7) & 31u)); }

// pass-2 anchor filter: cuckoo hash set (two choices, one slot each) of every 16-mer that
// starts at offset 0..7 of a pattern, as 32-bit packed values; lives in LDS.
// h_i(V) = ak_hash(V, m_i) >> (32 - log_size)   (merge.h: one v_mad_u32_u24 per hash)
struct DevAnchors {
    const uint32_t *table;        // [1 << log_size]; mode 0: one key per slot (unused slots hold a member key);
                                  // mode 1: two 16-bit fingerprints per slot (bucketed cuckoo, superset filter)
    uint32_t log_size;
    uint32_t mode;
    uint32_t m1, m2;
    uint32_t n_keys;
    uint32_t with_exc;            // 1: exception reads are probed on their packed words too (pattern set known to be pure ACGT)
};

// ---- device-side DR merge (dmerge.hip) ----
// small device-resident result/flag words of one merge; copied to pinned host memory afterwards
struct DevMergeState {
    uint32_t n_survivors;         // non-redundant variants
    uint32_t n_patterns;          // 2 * n_survivors
    uint32_t n_keys;              // distinct anchor keys
    uint32_t log_size;            // anchor table holds 1 << log_size slots (<= 15: staged in LDS)
    uint32_t fail;                // != 0: the device path does not apply (1 token outside ACGT/23..64 or too many 11-mers with
                                  // an 'N', 2 key set too large or empty, 4 cuckoo insertion gave up, 8 a bounded spin gave up,
                                  // 16 injected (tests), 32 a needle key with more candidates than DevMerge::group_cap,
                                  // 64 more tokens than the launch was sized for) -> the host merge is used instead
    uint32_t k0;                  // a member key (fills unused table slots)
    uint32_t all_t;               // the key 0xFFFFFFFF is a member
    uint32_t unused_a, unused_b;  // (the two range cursors moved to DevMerge::hot)
    uint32_t tab_mode;            // anchor table: 0 exact keys staged in LDS (log_size <= 15), 3 16-bit fingerprints of a
                                  // 2^16-slot table staged in LDS (anchor_fp), 2 exact keys probed in global memory
    uint32_t n_badk;              // distinct 11-mers that contain an 'N'
    uint32_t pad[5];
};

// Counters that EVERY block of a merge kernel adds to (survivors, keys, smallest key).  In DevMergeState they shared one cache
// line with `fail`, which every wave of every kernel reads: 2 600-4 096 block atomics on that line were 43 of k_dm_keys' 61 us
// and ~33 of k_dm_redundant's 55 (tools/dm_ablate.py, round 4).  Each counter is now 16 words in 16 different lines, a block
// uses stripe blockIdx % 16, and the one kernel that needs the value adds the stripes up.
// (The two range cursors — one returning atomic per block — live here as well, unstriped: a line of their own each.)
static constexpr uint32_t kDmHotStripes = 16, kDmHotCounters = 5, kDmHotWords = kDmHotCounters * kDmHotStripes * 32;
enum { kHotSurvivors = 0, kHotKeys = 1, kHotK0 = 2, kHotEntCursor = 3, kHotRdCursor = 4 };
__host__ __device__ inline uint32_t dm_hot(uint32_t counter, uint32_t block) { return (counter * kDmHotStripes + (block & (kDmHotStripes - 1u))) * 32u; }

static constexpr uint32_t kDmBadSlots = 4096;       // identity table of the 11-mers with an 'N' (open addressing)
static constexpr uint32_t kDmBadKmerCap = 2048;     // ... of which at most this many distinct ones; more -> host merge

// The view's arrays in one dense blob.  Sizes are only known on the device (groups, survivors, characters): k_dmx_apply
// evaluates the layout from the totals and publishes it (DevViewTotals) for the kernels behind it and for the host.
struct ViewLayout { uint64_t tok_off, tok_chars, grp_off, grp_tokens, pat_off, pat_chars, pat_group, total; };
__host__ __device__ inline ViewLayout view_layout(uint64_t n_tok, uint64_t n_groups, uint64_t n_pat, uint64_t tok_chars, uint64_t pat_chars)
{
    ViewLayout v;
    uint64_t at = 0;
    auto sec = [&](uint64_t bytes) { const uint64_t o = at; at += (bytes + 15u) & ~15ull; return o; };
    v.tok_off = sec(8 * (n_tok + 1)); v.tok_chars = sec(tok_chars); v.grp_off = sec(8 * (n_groups + 1)); v.grp_tokens = sec(4 * n_tok);
    v.pat_off = sec(8 * (n_pat + 1)); v.pat_chars = sec(pat_chars); v.pat_group = sec(4 * n_pat);
    v.total = at;
    return v;
}
struct DevViewTotals {
    uint32_t ok;                  // 1: the blob holds the complete view
    uint32_t n_tok, n_groups, n_kept, tok_chars, kept_chars, max_group, pad;
    ViewLayout lay;
};
static constexpr uint32_t kDmxVals = 5;             // per-token scan values: is root, group size, survivors, survivors' chars (at roots), length
static constexpr uint32_t kDmxTiles = 1024;         // tiles of 1024 tokens: n_tok <= 2^20

// Pattern ids on the device: pid = 2 * token_index + orientation (0 = the variant, 1 = its reverse complement), i.e. the
// pattern arrays ARE the per-token arrays (packed[t][2o..2o+1], tmask[t][o]); ids of dropped variants are simply unused.
// (The reference's pattern ORDER — per group, survivors then reverse complements, WorkHorse.cpp:690-697 — only exists in
// the host view; pass 2 depends on the set alone, SURVEY a-14.)
// The device merge and pass 2's anchor probe take DR strings of kDevMinDR .. 64 bases.  The anchor argument: a pattern P that
// occurs at offset o of a read contains the read's aligned 16-base window at a = ceil_A(o) iff |P| >= 16 + A - 1 — every 8 bases
// for |P| >= 23 (crass's default lowDRsize), every 4 bases for |P| >= 19 (-d 19 .. 22: four keys per pattern instead of eight,
// twice the windows per read).  Shorter patterns (-d 8 .. 18) take the host merge and the byte-wise automaton kernels.
static constexpr uint32_t kDevMinDR = 19;
struct DevMerge {
    // input: distinct candidate DR strings in first-occurrence (= token) order
    const char *dx_chars; const uint16_t *dx_len;
    uint32_t stride, n_tok;
    const uint32_t *d_ntok;       // nullptr, or the device-side token count (n_tok is then the bound the launch is sized for)
    uint32_t min_len;             // lowDRsize: no token is shorter (kDevMinDR <= min_len)
    uint32_t akey_shift;          // log2 of the anchor windows' alignment: 3 (every 8 bases: min_len >= 23) or 2 (every 4: min_len >= 19)
    uint32_t thr;                 // max(kmer_clust_size, 2): sightings of a group that decide membership
    uint32_t kmax;                // k-mer slots per token (stride - 10)
    // per token
    uint64_t *packed;             // [n_tok][4] 2-bit packed string (lo, hi) and its reverse complement (lo, hi) = pattern bits [pid][2]
    uint64_t *tmask;              // [n_tok][2] bit i: base i is an 'N' (packed as 'A'); forward and reverse complement = [pid]
    uint32_t *codes;              // [n_tok][kmax] laurenized 11-mer codes; (1 << 22) + slot for an 11-mer with an 'N'
    uint32_t *owner;              // [(1 << 22) + kDmBadSlots] smallest token containing the code
    unsigned long long *bk_key;   // [kDmBadSlots] identity of the 11-mers with an 'N': laurenized 33-bit key | 1 << 40; 0 = empty
    uint32_t *root_of;            // [n_tok] first token of the token's group
    uint8_t  *blank;              // [n_tok] removed by removeRedundantRepeats
    // needle index of removeRedundantRepeats: key = ((root + 1) << 32) | first 16 bases
    unsigned long long *rset_key; // [1 << rset_log]
    uint32_t *rset_cnt, *rset_base, *rset_fill;   // [1 << rset_log]
    uint32_t rset_log;
    uint32_t *rd_slot;            // [n_tok] the member's key slot (bit 31: it claimed the slot)
    uint64_t *rents;              // [n_tok][4] {len | token << 32, bits lo, bits hi, N mask}, grouped by key
    uint32_t *pat_token;          // [2 * n_tok] StringToken of pattern pid (= (pid >> 1) + 2)
    const uint64_t *pat_mask;     // = tmask
    // anchor keys: entry e = pid * 8 + r (capacity 16 * n_tok)
    unsigned long long *kset_key; // [1 << kset_log] distinct keys (key | 1 << 32; 0 = empty)
    uint32_t *kset_cnt, *kset_base, *kset_fill;   // [1 << kset_log] the key's entries: ents[base .. base + cnt)
    uint32_t kset_log;
    uint32_t *ent_slot;           // [16 * n_tok] the entry's key slot | 0x80000000 when the entry claimed it; 0xFFFFFFFF: no entry
    uint64_t *ents;               // verification index, grouped by key: [ent_cap][2] {meta, bases 0..31}, then [ent_cap][2] {bases 32..63, N mask}
    uint32_t ent_cap;             // = 16 * n_tok
    uint32_t *anchor_tab;         // [1 << tab_log_alloc] cuckoo table of the keys (see DevAnchors)
    uint32_t *anchor_fp;          // [1 << 15] words: tab_mode 3 = 2^16 16-bit fingerprints of the slots' keys;
                                  // tab_mode 2 = a 2^20-bit Bloom filter (2 hashes) staged in LDS in front of the L2 probes
    uint32_t tab_log_alloc;
    uint32_t m1, m2;              // hash multipliers of the table (ak_hash, merge.h)
    uint32_t n_cu;                // compute units of the device (bounds the grid of the kernel whose waves wait for each other)
    DevMergeState *st;
    // pinned host memory the last kernel exports to (state words, root and dropped flag per token)
    DevMergeState *h_st; uint32_t *h_root; uint8_t *h_blank;
    // ---- the host view (crass_merge_view) assembled on the device: k_dmx_* (dmerge.hip) ----
    // per ROOT token (first token of a group): members, survivors of removeRedundantRepeats, the survivors' characters,
    // a claim cursor; then the group's dense id and its bases in the view's arrays (exclusive prefix sums in root order)
    uint32_t *x_size, *x_kept, *x_kchars, *x_fill;                     // [n_tok] cleared by dm_init_slice
    uint32_t *x_gid, *x_goff, *x_pat0, *x_pch0;                        // [n_tok] written by k_dmx_apply for roots
    uint32_t *x_members;          // [n_tok] member keys (blank << 27 | len << 20 | token), grouped by group, any order inside
    uint32_t *x_tile;             // [1024][kDmxVals] per-tile sums (tiles of 1024 tokens) + [0..3] behind them: max group size
    uint8_t  *x_blob;             // the view's arrays, dense (view_layout over the totals), device memory
    DevViewTotals *x_tot;         // device; x_htot: its pinned mirror
    DevViewTotals *x_htot;
    uint32_t x_on;                // 0: no export (the host rebuilds the view from h_root / h_blank)
    uint32_t x_group_cap;         // a group with more members than this is not ranked here (quadratic): the host builds the view
    // stage flags (engine.cpp: crass_hip_ctx::h_flags): pinned host words that the FIRST kernel of a stage stores at system scope
    // — everything queued in front of that kernel is then complete — for the host to poll instead of an event recorded
    // between two kernels (~6 us of stream time each).  flag_pre: k_dm_pack_codes; flag_post: pass 2's probe
    uint32_t *flag_pre; uint32_t flag_pre_val;
    uint32_t *flag_post; uint32_t flag_post_val;
    uint32_t *flag_blank; uint32_t flag_blank_val;  // k_dm_keys: blank[] is final (the host builds the view's pattern list from it)
    uint32_t x_sort_max;          // groups of 65 .. this many members (<= 2048) are ranked by k_dmx_sort, the others by k_dmx_rank
    uint32_t *hot;                // the counters every block adds to, striped: [kDmHotCounters][kDmHotStripes] words, 128 bytes apart (dm_hot)
    uint32_t ablate;              // profiling aid (CRASS_DM_ABLATE, tools/dm_ablate.py): bits switch parts of the merge kernels OFF — results are then garbage
    uint32_t inject_fail;         // tests only: start with the fail word set (exercises the fall-back to the host merge)
    uint32_t group_cap;           // a needle key with more candidates than this sets fail bit 32: removeRedundantRepeats here
                                  // compares a window with every shorter member that shares its first 16 bases, which is
                                  // quadratic in a group of near-identical variants; the host merge handles such inputs
};

// the words a merge polls, counts into or probes, cleared by k_dm_init — or, when the launch is sized before pass 1 has
// finished (crass_hip_seed_scan queues the merge ahead of time), by the survivor kernel on its way: a 148 us, issue-bound
// kernel next to which 20 MB of stores are free.  tid / nth: this thread's index / the number of threads that share the job.
static __device__ __forceinline__ void dm_init_slice(const DevMerge &M, uint64_t tid, uint64_t nth)
{
    uint4 ones; ones.x = ones.y = ones.z = ones.w = 0xFFFFFFFFu;
    uint4 *o4 = reinterpret_cast<uint4 *>(M.owner);
    for (uint64_t i = tid; i < ((1u << 22) + kDmBadSlots) / 4; i += nth) o4[i] = ones;
    for (uint64_t i = tid; i < kDmBadSlots; i += nth) M.bk_key[i] = 0ull;
    for (uint64_t i = tid; i < M.n_tok; i += nth) M.root_of[i] = 0xFFFFFFFFu;
    if (M.x_on) {
        for (uint64_t i = tid; i < M.n_tok; i += nth) { M.x_size[i] = 0u; M.x_kept[i] = 0u; M.x_kchars[i] = 0u; M.x_fill[i] = 0u; }
        if (tid < 4) M.x_tile[kDmxTiles * kDmxVals + tid] = 0u;
    }
    const uint64_t ks = 1ull << M.kset_log;
    for (uint64_t i = tid; i < ks; i += nth) { M.kset_key[i] = 0ull; M.kset_cnt[i] = 0u; M.kset_fill[i] = 0u; }
    const uint64_t rs = 1ull << M.rset_log;
    for (uint64_t i = tid; i < rs; i += nth) { M.rset_key[i] = 0ull; M.rset_cnt[i] = 0u; M.rset_fill[i] = 0u; }
    for (uint64_t i = tid; i < (1u << 15); i += nth) M.anchor_fp[i] = 0u;
    uint4 *t4 = reinterpret_cast<uint4 *>(M.anchor_tab);
    for (uint64_t i = tid; i < (1ull << M.tab_log_alloc) / 4; i += nth) t4[i] = ones;
    if (tid < kDmHotStripes) {
        M.hot[dm_hot(kHotSurvivors, (uint32_t)tid)] = 0u; M.hot[dm_hot(kHotKeys, (uint32_t)tid)] = 0u; M.hot[dm_hot(kHotK0, (uint32_t)tid)] = 0xFFFFFFFFu;
        M.hot[dm_hot(kHotEntCursor, (uint32_t)tid)] = 0u; M.hot[dm_hot(kHotRdCursor, (uint32_t)tid)] = 0u;
    }
    if (tid == 0) {
        DevMergeState s{};
        s.k0 = 0xFFFFFFFFu;
        s.fail = M.inject_fail ? 16u : 0u;
        *M.st = s;
    }
}

// a stage flag's store: every write of the kernels in front of this one is visible to the host that sees it
static __device__ __forceinline__ void stage_flag_store(uint32_t *flag, uint32_t val)
{
    __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// one-collective exchange (see crass_hip_exchange_setup): fill this rank's send buffer / unpack the gathered buffers
// p1_counts / surv_bound (a deferred pass 1, engine.cpp p1d): the stage's device counters — [0] survivors, [2] found records,
// [3] worst error, [5] de-duplication flags — are checked HERE, by the kernel that fills the send buffer: a launch the host will
// find unusable (more survivors than the bound, none, an error, no exact distinct list) sets bit 63 of the header's count, which
// every rank's k_xg_unpack reports as an exchange that did not fit
hipError_t launch_xg_fill(const char *dx_chars, const uint16_t *dx_len, const uint32_t *d_nd, uint32_t stride, uint64_t cap_rows,
                          uint32_t slot_bytes, uint8_t *send, hipStream_t st, const uint32_t *p1_counts = nullptr, uint64_t surv_bound = 0,
                          uint32_t *h_flag = nullptr, uint32_t flag_val = 0);      // h_flag: stage flag "pass 1 is complete" (see DevMerge::flag_pre)
// xinfo (device, 8 words): [0] n_global, [1] my_offset, [2] overflow flag, [3] largest per-rank count
hipError_t launch_xg_unpack(const uint8_t *recv, uint32_t world, uint32_t rank, uint32_t stride, uint64_t cap_rows, uint32_t slot_bytes,
                            char *g_chars, uint16_t *g_len, uint32_t *xinfo, hipStream_t st, uint32_t *h_xinfo = nullptr,   // h_xinfo: pinned mirror
                            uint32_t *zero2 = nullptr,                                                            // two words cleared on the way
                            unsigned long long *dd_keys = nullptr, uint32_t *dd_first = nullptr, uint32_t dd_size = 0);  // ... and the de-duplication table that follows
// init_done: the tables were cleared by an earlier kernel of the step (dm_init_slice)
// view_st / ev_fork / ev_view (M.x_on): the view export runs on view_st beside the merge's last three kernels, forked behind
// k_dm_redundant (ev_fork); ev_view is recorded behind its last kernel
hipError_t launch_device_merge(const DevMerge &M, hipStream_t st, bool init_done = false, hipStream_t view_st = nullptr,
                               hipEvent_t ev_fork = nullptr, hipEvent_t ev_view = nullptr,
                               hipEvent_t ev_apply = nullptr);     // ev_apply: behind k_dmx_apply (the blob's token half is complete)
// pass-2 anchor filter with table parameters read from the device (M.st); flags nothing when M.st->fail
hipError_t launch_anchor_filter_dev(const DevReads &R, const DevMerge &M, const uint8_t *found_flag, uint64_t *hitmask, hipStream_t st);
hipError_t launch_dm_verify(const DevReads &R, const DevMerge &M, const uint64_t *idx, const uint32_t *d_n, uint64_t n_max,
                            uint32_t *info_by_slot, uint32_t *pid_by_slot, hipStream_t st);
hipError_t warm_dmerge_module();      // loads dmerge.hip's code object on the current device (crass_hip_create)

// layout of the survivor kernel's dynamic LDS (bytes), computed on the host
struct SurvLds {
    uint32_t seq_bytes;           // >= maxL + 16, multiple of 16
    uint32_t ss_cap;              // entries (uint32)
    uint32_t row_elems;           // uint16 entries per Levenshtein boundary row
    uint32_t words_cap;           // uint32 entries of the packed-read copy
    uint32_t hint_words;          // uint64 entries of the read's per-position seed hints (long reads)
    uint32_t total_bytes;
    uint32_t seq_window;          // != 0: seq_bytes holds an ASCII WINDOW of a packed read (rh_ascii) — a region that does not fit, like a
                                  // start/stop list beyond ss_cap or a string beyond the rows, sends the read to the launch with the full layout (err 6)
    uint32_t ss_slot;             // entries per survivor slot of the start/stop pool in slot mode: the FULL layout's ss_cap in every launch of a set
};

// ---- launch wrappers implemented in kernels.hip (all asynchronous on `st`) ----
hipError_t launch_filter_general(const DevReads &R, const DevParams &P, uint64_t *hitmask,
                                 uint32_t max_len, hipStream_t st);
// per-position seed hints for long / ragged packed reads (default window and DR/spacer bounds only):
// hint_off[n_reads + 1] = prefix sums of ceil(L/64) (device), n_words = hint_off[n_reads] (host-known)
hipError_t launch_hint_positions(const DevReads &R, const DevParams &P, const uint64_t *hint_off, const uint32_t *blk_read, uint64_t n_words,
                                 uint64_t *hint_bits, hipStream_t st,       // blk_read[b] = read of tile 256 b (ragged lengths; else nullptr)
                                 uint64_t w_begin = 0, uint64_t w_end = ~0ull,       // the hint words [w_begin, w_end) only; w_begin a multiple of 256
                                 uint64_t *hitmask = nullptr);      // ... and the bits as the seed-scan filter (cleared by the caller): bit r = read r has a hint bit
// ... the same under another window or seed lattice: every position's bit computed on the spot, nothing kept (window 6 .. 9, shifts 17 .. 127)
hipError_t launch_hint_filter_any(const DevReads &R, const DevParams &P, const uint64_t *hint_off, const uint32_t *blk_read, uint64_t n_words,
                                  uint64_t *hitmask, uint64_t *hint_bits, hipStream_t st);      // hint_bits: every position's bit, kept for the survivors' walks
// long reads with position hints, an identity survivor list and no exception read: the walk of the reads without an array; a read
// that needs the full searchCore leaves with err == 7 for launch_survivor(..., punt_only = 7)
hipError_t launch_long_light(const DevReads &R, const DevParams &P, const uint32_t *d_n, uint64_t n_max, SurvOut *out, uint64_t slot_base,
                             uint32_t max_len, hipStream_t st, uint32_t *punt_list, uint32_t *d_punt_n, const uint64_t *surv_idx = nullptr);      // punt_list[(*d_punt_n)++] = slot of a read handed over
// clear_found / cleared: the fixed-range kernel also zeroes the n_reads + 1 found flags on its way (*cleared says whether the form
// that was launched did: the caller clears them itself otherwise)
hipError_t launch_filter_fast(const DevReads &R, const DevParams &P, uint64_t *hitmask, uint32_t *seed_hint, hipStream_t st,
                              uint8_t *clear_found = nullptr, bool *cleared = nullptr);
// ---- "last VGPR of the allocation" guard ----
// Observed on the MI355X pool (minimal reproductions: profiles/ubench/vgpr_edge2.hip and vgpr_edge3.hip, write-up in
// DESIGN.md 3.9): a wave that is NOT the first wave on its SIMD mis-executes a 64-bit shift (v_lshrrev_b64 / v_lshlrev_b64 /
// v_ashrrev_i64) whose 32-bit shift AMOUNT lives in the LAST register of its VGPR allocation (granule 8 registers on gfx950);
// 32-bit ALU reads of that register, 64-bit data pairs that end in it and v_mad_u64_u32 factors are not affected (100
// launches each).  k_recruit_finish (24 VGPRs, its 128-bit shift amount in v23) flipped DRLowLexi for about 1 recruit in 1000
// that way.  The build disassembles every kernel and refuses exactly that pattern (crass_amd/vgpr_guard.py); the remedy for a
// refused kernel is CRASS_VGPR_FLOOR(n): it marks v<n> as used (no instruction is emitted), which moves .vgpr_count to at least
// n + 1, off the granule boundary.  One kernel carries it today (k_recruit_finish<false>).
#define CRASS_VGPR_FLOOR(N) asm volatile("" ::: "v" #N)

#ifdef __HIPCC__
// ---- one atomic per BLOCK on the state words.  A device-scope atomic on ONE address retires every ~2.5-3.5 ns on this
// part however many CUs issue them, so "every claimant adds its count to a cursor" serialises: 6.9 k claimants = 24 us
// of a 26 us kernel at 10 M reads, 30 k = 150 us at 100 M.  block_reserve: exclusive offsets for the threads of a block
// from one atomicAdd (wave scan by shuffles, wave totals through LDS).  Every thread of the block must call it.
template <int THREADS>
static __device__ __forceinline__ uint32_t block_reserve(uint32_t v, uint32_t *counter)
{
    __shared__ uint32_t br_tot[THREADS / 64];
    __shared__ uint32_t br_base;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += u;
    }
    if (lane == 63) br_tot[w] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int i = 0; i < THREADS / 64; i++) { const uint32_t x = br_tot[i]; br_tot[i] = tot; tot += x; }
        br_base = tot ? atomicAdd(counter, tot) : 0u;
    }
    __syncthreads();
    return br_base + br_tot[w] + incl - v;
}
#endif

// Single-pass ordered compaction (decoupled look-back): per-tile status words, a ticket counter that hands out
// tile ids in start order (a tile only ever waits for tiles that started before it; the launch's last ticket resets it)
// and an epoch tag so that the status words need no clearing between launches.  Built by crass_hip_ctx::next_lookback().
struct Lookback {
    unsigned long long *status;   // [>= tiles] (epoch << 34) | (flag << 32) | value; flag 1 = tile total, 2 = inclusive prefix
    uint32_t *ticket;             // 0 between launches: the block that draws a launch's last ticket resets it (lb_tile_id)
    uint32_t epoch;               // 30 bits, never 0
    uint32_t *fail;               // pinned host word, set when a spin gives up (never expected; reported as an error)
};
// mask words per tile: fat tiles (1024 threads x 4 words) for the masks over all reads, small ones (256 x 1) for the short masks
static inline uint32_t lookback_tile_words(uint64_t n_words) { return n_words >= 16384 ? 4096u : 256u; }

// mask (n_words 64-bit words) -> ascending index list; *d_count receives the number of set bits.
// scratch: word_prefix[n_words] u32, block_sums[(n_words+255)/256 + 1] u32
hipError_t launch_compact(const uint64_t *mask, uint64_t n_words, uint64_t n_bits, uint32_t *word_prefix,
                          uint32_t *block_sums, uint64_t *out_idx, uint64_t out_cap, uint32_t *d_count,
                          hipStream_t st, uint32_t *zero_a = nullptr, uint32_t n_a = 0, uint32_t *zero_b = nullptr, uint32_t n_b = 0,
                          const Lookback *lb = nullptr);     // lb: one kernel instead of three
// (zero_a/zero_b: up to 1024 words each that the scan kernel clears on the way)
hipError_t launch_survivor(const DevReads &R, const DevParams &P, bool exceptions,
                           const uint64_t *surv_idx, const uint32_t *d_n_surv, uint64_t n_surv_max,
                           SurvOut *out, char *dr_chars, uint32_t dr_stride,
                           uint32_t *ss_pool, uint32_t ss_pool_cap, uint32_t *d_ss_used,
                           uint8_t *found_flag, const uint32_t *seed_hint, const SurvLds &lds, int grid, hipStream_t st,
                           int punt_only = 0,
                           uint64_t slot_base = 0, uint64_t slot_total = 0,       // a slice of a larger launch: out / dr_chars point at the slice, the start/stop pool is shared
                           const uint32_t *punt_list = nullptr, const uint32_t *d_punt_n = nullptr,       // punt_only with a list: exactly those slots, one per wave and turn
                           uint32_t *redo_list = nullptr);      // [0] count, [1 ..] slots this launch hands on to the launch with the full layout (err 6)
hipError_t launch_survivor_lanes(const DevReads &R, const DevParams &P, const uint64_t *surv_idx, const uint32_t *d_n_surv,
                                 uint64_t n_surv_max, SurvOut *out, char *dr_chars, uint32_t dr_stride, uint32_t *ss_pool,
                                 uint32_t ss_cap, uint8_t *found_flag, const uint32_t *seed_hint, hipStream_t st,
                                 const DevMerge *init_merge = nullptr, uint32_t max_len = 0, uint32_t *punt_cnt = nullptr);     // max_len: the rows' size when the strides differ   // also clears that merge's tables (dm_init_slice)
hipError_t launch_recruit_general(const DevReads &R, const DevAutomaton &A, const uint8_t *found_flag,
                                  uint64_t *hitmask, uint32_t *hit_info, hipStream_t st);
hipError_t launch_recruit_lds(const DevReads &R, const DevAutomaton &A, const uint8_t *found_flag,
                              uint64_t *hitmask, uint32_t *hit_info, hipStream_t st);
hipError_t launch_anchor_filter(const DevReads &R, const DevAnchors &K, const uint8_t *found_flag,
                                uint64_t *hitmask, hipStream_t st);
hipError_t launch_recruit_list(const DevReads &R, const DevAutomaton &A, const uint64_t *idx, const uint32_t *d_n,
                               uint64_t n_max, uint32_t *info_by_slot, uint32_t *pid_by_slot, hipStream_t st);
hipError_t launch_recruit_exceptions(const DevReads &R, const DevAutomaton &A, const uint8_t *found_flag,
                                     uint32_t *exc_hit_info /*[n_exc], 0 = none*/, hipStream_t st);
hipError_t launch_recruit_finish(const DevReads &R, const uint64_t *hit_idx, const uint32_t *d_n_hits,
                                 uint64_t n_hits_max, const uint32_t *hit_info, bool info_by_slot, bool exceptions,
                                 const uint32_t *pid_by_slot, const uint32_t *pat_token,
                                 RecruitOut *out, char *dr_chars, uint32_t dr_stride, hipStream_t st,
                                 const uint64_t *pat_mask = nullptr);   // device-built patterns: 'N' positions per pattern
hipError_t launch_levenshtein_batch(const uint8_t *chars, const uint64_t *a_off, const uint32_t *a_len,
                                    const uint64_t *b_off, const uint32_t *b_len, uint64_t n_pairs,
                                    int32_t *dist, float *sim, uint32_t max_len, hipStream_t st);
// found_flag[header_id(idx[i])] = 1 for every listed read (readsFound entries of other input files, libcrispr.cpp:411)
hipError_t launch_mark_found(const uint64_t *idx, uint64_t n, const uint64_t *header_id, uint8_t *found_flag, hipStream_t st);
hipError_t launch_copy_to_host(const void *d_src, void *h_dst, uint64_t bytes, hipStream_t st);   // pinned destination, few workgroups
// device -> pinned host on the DMA engines (sdma.cpp); nullptr / false: not available, use the copy kernel
struct SdmaCopy;
SdmaCopy *sdma_create();
void sdma_destroy(SdmaCopy *s);
bool sdma_start(SdmaCopy *s, const void *d_src, void *h_dst, size_t bytes);
int sdma_wait(SdmaCopy *s);
bool sdma_pending(const SdmaCopy *s);
hipError_t launch_build_exc_mask(const uint64_t *exc_read, uint64_t n_exc, uint32_t *exc_mask, hipStream_t st);

// k_found_mask + compaction as one decoupled-look-back kernel (tiles of 256 survivor slots); fidx[rank] = slot
hipError_t launch_found_compact(const SurvOut *out, const uint32_t *d_n, uint64_t n_max, uint32_t *d_err, unsigned long long *dd_keys,
                                uint32_t *dd_first, uint32_t dd_size, uint64_t *fidx, uint32_t *d_nf, const Lookback &lb, hipStream_t st);
// host-loop form of the pass-1 sink (long / ragged reads): the found records of a chunk and nothing else, ready for one
// set of exact-size copies.  sel_mask: bit per slot = found; d_err: 2 = the reference would throw, 1 = capacity (exception
// reads in the list, err == 5, are somebody else's business); gather: record k = slot fidx[k] with its start/stops moved to
// a dense pool (offsets from one atomicAdd per block)
hipError_t launch_select_found(const SurvOut *out, uint64_t n, uint64_t *mask, uint32_t *d_err, hipStream_t st);
hipError_t launch_gather_sparse(const uint64_t *fidx, const uint32_t *d_nf, uint64_t n_max, const SurvOut *out, const char *dr_chars, uint32_t dr_stride,
                                const uint32_t *ss_pool, SurvOut *g_out, uint64_t *g_slot, char *g_dr, uint32_t *g_ss, uint32_t g_ss_cap,
                                uint32_t *d_ss_total, hipStream_t st, uint16_t *g_dr_len = nullptr,      // g_dr_len: the records' DR lengths, dense
                                int ss16 = 0);                                               // the packed start/stops as uint16 (positions < 65 536)
hipError_t launch_found_mask(const SurvOut *out, const uint32_t *d_n, uint64_t n, uint64_t *mask, uint32_t *d_err, hipStream_t st,
                             unsigned long long *dd_keys = nullptr, uint32_t *dd_first = nullptr, uint32_t dd_size = 0);   // also clears that table
// found records -> (a) the compact hand-off blob (p1_blob_layout; device memory — the runtime copies its used bytes to
// the host — or pinned host memory); (b) dense device arrays of the DR strings for the de-duplication, which this kernel
// also starts: every string is inserted into the (cleared) table on the way
hipError_t launch_gather_found(const uint64_t *fidx, const uint32_t *d_nf, uint64_t n_max, const SurvOut *out,
                               const uint64_t *surv_idx, uint64_t read_base, const char *dr_chars, uint32_t dr_stride,
                               const uint32_t *ss_pool, uint32_t ss_cap, uint32_t ss_elem, uint8_t *blob,
                               uint16_t *g_dr_len, char *g_dr, hipStream_t st,
                               unsigned long long *dd_keys = nullptr, uint32_t *dd_first = nullptr, uint32_t dd_size = 0,
                               uint64_t *dd_hash = nullptr, uint32_t *dd_slot = nullptr, uint32_t *d_mismatch = nullptr);
// probes after which an insert into the de-duplication table gives up (a table sized for a bound on the DISTINCT strings
// that turned out too small): bit 2 of the mismatch word
static constexpr uint32_t kDdMaxProbes = 256;
// pass-1 hand-off blob: what the host needs of the nf found records, compact and packed back to back so that
// it crosses PCIe once: read index, repeat length, start/stop count and list (read positions fit 16 bits:
// CRASS_HIP_MAX_READ_LEN < 65536), orientation flag.  The DR string of candidate k is the distinct string
// cand_distinct[k], which the host already has.  The layout is a function of nf, evaluated on the device (the
// count lives there) and again on the host once it knows nf.
struct P1Blob { uint64_t read, replen, nss, low, ss, total; uint32_t ss_elem; };
// ss_elem: bytes per start/stop entry — 1 when every read position fits a byte (reads of at most 256 bases), else 2
__host__ __device__ inline P1Blob p1_blob_layout(uint64_t nf, uint32_t ss_cap, uint32_t ss_elem = 2)
{
    P1Blob b;
    uint64_t at = 0;
    auto sec = [&](uint64_t bytes) { const uint64_t o = at; at += (bytes + 15u) & ~15ull; return o; };
    b.read = sec(nf * 8); b.replen = sec(nf * 2); b.nss = sec(nf); b.low = sec(nf); b.ss = sec(nf * (uint64_t)ss_cap * ss_elem);
    b.total = at; b.ss_elem = ss_elem;
    return b;
}
// pass-2 hand-off blob: header (record count, 16 bytes) + compact arrays with `cap` slots each; the DR string of a
// recruit is its token's string
struct P2Blob { uint64_t read, token, start, end, dr_len, low, total; uint32_t narrow; };
// narrow: read sets of fewer than 2^32 reads of at most 255 bases (every short-read job) carry the LOCAL read index in four bytes
// and the two positions in one byte each — 12 instead of 18 bytes per recruit over PCIe, which bounds the pack kernel (2: see below)
__host__ __device__ inline P2Blob p2_blob_layout(uint64_t cap, uint32_t narrow = 0)
{
    P2Blob b;
    uint64_t at = 16;
    auto sec = [&](uint64_t bytes) { const uint64_t o = at; at += (bytes + 15u) & ~15ull; return o; };
    b.read = sec(cap * (narrow ? 4 : 8)); b.token = sec(cap * 4); b.start = sec(cap * (narrow ? 1 : 2));
    // narrow == 2 (the device merge's tokens are known for every recruit): 9 bytes — the orientation flag rides in bit 31 of the
    // token, the repeat's length is the token string's, and its end is start + length - 1 in either orientation (k_recruit_finish)
    if (narrow == 2) { b.end = at; b.dr_len = at; b.low = at; }
    else { b.end = sec(cap * (narrow ? 1 : 2)); b.dr_len = sec(cap); b.low = sec(cap); }
    b.total = at; b.narrow = narrow;
    return b;
}
// valid hits (dr_len != 0) of the finish kernel's slots -> compacted, read-ordered compact arrays in `blob`
hipError_t launch_pack_p2_blob(const RecruitOut *rec, const uint64_t *hit_idx, uint64_t read_base,
                               const uint32_t *d_n_hits, uint64_t n_hits_max, uint64_t *mask, uint32_t *word_prefix, uint32_t *block_sums,
                               uint64_t *vidx, uint32_t *d_nv, uint8_t *blob, hipStream_t st,
                               uint32_t *h_n_hits = nullptr,         // pinned word that receives *d_n_hits
                               const Lookback *lb = nullptr, uint32_t narrow = 0);
hipError_t launch_dx_tokens(const char *dr, const uint16_t *dr_len, const uint64_t *hash, uint32_t stride, const uint32_t *d_n, uint32_t n, uint32_t *rep,
                            uint32_t *slot_of, const uint32_t *first,       // slot_of: with lb, overwritten by the representatives' ranks
                            uint64_t *mask, uint32_t *word_prefix, uint32_t *block_sums, uint64_t *dx_idx, uint32_t *d_nd,
                            uint32_t *d_mismatch, uint32_t *dmap, char *out_chars, uint16_t *out_len, uint64_t *out_hash,
                            char *dev_chars, uint16_t *dev_len, hipStream_t st,
                            const uint32_t *cnt_src = nullptr, uint32_t *cnt_dst = nullptr, uint32_t n_cnt = 0,    // counters -> pinned host words
                            const Lookback *lb = nullptr);      // lb: look-back over tiles of 1024 candidates (next_lookback_tiles((n + 1023) / 1024))
hipError_t launch_dr_dedupe(const char *dr, const uint16_t *dr_len, uint32_t stride, const uint32_t *d_n, uint32_t n, unsigned long long *keys,
                            uint32_t *first, uint32_t table_size, uint64_t *hash_out, uint32_t *slot_tmp, uint32_t *rep, hipStream_t st,
                            bool table_cleared = false);
// row_len_cap: longest string the Levenshtein fallback rows hold in this layout (reads that need more come back
// with err == 6 and are redone with the uncapped layout)
SurvLds survivor_lds_layout(uint32_t max_len, const DevParams &P, uint32_t row_len_cap = 0xFFFFFFFFu,
                            uint32_t seq_window_bytes = 0, uint32_t ss_entries_cap = 0);   // the two caps of the long-read layout (0: none)
hipError_t upload_comp_table(const unsigned char *tab128);

} // namespace crass
