/*
 * crass_hip.h — C ABI of the MI355X-native crass search engine (libcrass_hip.so).
 *
 * This is the drop-in boundary for crass's WorkHorse search path.  The reference has no
 * FFI layer: the seam is three C++ free functions called from WorkHorse::parseSeqFiles
 * (reference paths are relative to the crass v1.0.1 tree):
 *
 *   searchFile()               src/crass/libcrispr.h:74-80   (call: WorkHorse.cpp:340-346)
 *   createNonRedundantSet()    src/crass/WorkHorse.h:111-112 (call: WorkHorse.cpp:370)
 *   findSingletons()           src/crass/libcrispr.h:86-92   (call: WorkHorse.cpp:386)
 *   sink: addReadHolder()      src/crass/libcrispr.h:123-125
 *
 * Each entry point below names the reference interface it replaces.  Plain pointers and
 * sizes only; caller-allocated/caller-freed flat buffers or views into context-owned
 * memory; no C++ objects or exceptions cross this boundary; every function returns an
 * int status (0 = CRASS_OK) and crass_hip_strerror() explains it.  One context per GPU,
 * one host thread per context (the reference is single-threaded).
 *
 * INTEGRATION.md shows the C++ adapter a crass maintainer would add on top of this.
 */
#ifndef CRASS_HIP_H
#define CRASS_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRASS_HIP_ABI_VERSION 3   /* 2: crass_counters, crass_fastx and crass_synth_spec grew (round 2); group API (round 3);
                                    3: crass_counters grew (used_device_view, n_view_fallbacks) (round 4) */

/* ---- status codes (reference: crispr::exception -> exit code) ---- */
enum {
    CRASS_OK = 0,
    CRASS_ERR_INVALID_ARG   = 1,   /* bad pointer / size / parameter combination             */
    CRASS_ERR_UNSUPPORTED   = 2,   /* parameter outside the engine's implementation limits   */
    CRASS_ERR_NO_DEVICE     = 3,   /* no HIP device / HIP runtime failure at init            */
    CRASS_ERR_HIP           = 4,   /* a HIP call failed (crass_hip_last_hip_error())         */
    CRASS_ERR_OOM           = 5,   /* host or device allocation failed                       */
    CRASS_ERR_STATE         = 6,   /* call order violated (e.g. recruit before patterns)     */
    CRASS_ERR_SEARCH_FATAL  = 7,   /* the reference would have thrown "Fatal error in search
                                      algorithm!" (libcrispr.cpp:141-149)                    */
    CRASS_ERR_OVERFLOW      = 8,   /* an internal device pool overflowed                     */
    CRASS_ERR_IO            = 9,   /* file could not be opened / parsed                      */
    CRASS_ERR_RCCL          = 10   /* an RCCL call failed (crass_hip_group_last_error())     */
};

/* implementation limits of the device path (checked by crass_hip_create) */
#define CRASS_HIP_MAX_WINDOW   9      /* CRASS_DEF_MAX_SEARCH_WINDOW_LENGTH, crassDefines.h:55 */
#define CRASS_HIP_MIN_WINDOW   6
#define CRASS_HIP_MAX_DR       240    /* highDRsize upper bound handled on device             */
#define CRASS_HIP_MAX_READ_LEN 60000  /* longest read the device path accepts                 */

/* ---- options: POD mirror of the hot-path fields of `options` (crassDefines.h:140-170) ---- */
typedef struct {
    uint32_t lowDRsize;          /* -d  default 23  (crassDefines.h:121) */
    uint32_t highDRsize;         /* -D  default 47  (:122)               */
    uint32_t lowSpacerSize;      /* -s  default 26  (:123)               */
    uint32_t highSpacerSize;     /* -S  default 50  (:124)               */
    uint32_t searchWindowLength; /* -w  default 8   (:56), 6..9          */
    uint32_t minNumRepeats;      /* -n  default 2   (:91)                */
    int32_t  kmer_clust_size;    /* -k  default 6   (:67)                */
} crass_params;

void crass_default_params(crass_params *p);       /* crass.cpp:430-460 */

typedef struct crass_hip_ctx crass_hip_ctx;

/* ---- reads: the device-resident replacement for "re-read every file twice" ---- *
 * 2-bit packed bases (A=0 C=1 G=2 T=3), 16 bases per uint32, base i of a read in bits
 * [2*(i%16), 2*(i%16)+2) of word i/16; every read starts on a word boundary.
 * A read containing any byte outside {A,C,G,T} is an *exception read*: it keeps its slot in
 * `packed` (content ignored) and is listed, as raw bytes, in the exception arrays, so that
 * the reference's raw-byte semantics (N matches N, lower case is not ACGT)
 * are kept on the device's byte-wise kernels.                                              */
typedef struct {
    uint64_t        n_reads;
    const uint32_t *packed;        /* all reads, word-aligned                                      */
    uint32_t        stride_words;  /* >0: read i starts at word i*stride_words; 0: use word_off    */
    const uint64_t *word_off;      /* [n_reads] start word of each read (stride_words == 0)        */
    uint32_t        uniform_len;   /* >0: every read has this length; 0: use lengths               */
    const uint32_t *lengths;       /* [n_reads] (uniform_len == 0)                                 */
    uint64_t        n_exceptions;
    const uint64_t *exc_read;      /* [n_exceptions] ascending read indices                        */
    const uint64_t *exc_off;       /* [n_exceptions+1] offsets into exc_bytes                      */
    const uint8_t  *exc_bytes;     /* raw sequence bytes of the exception reads                    */
    const uint64_t *header_id;     /* [n_reads] or NULL.  header_id[i] = index of the FIRST read
                                      with the same header (readsFound is keyed by header string,
                                      libcrispr.cpp:138,411); NULL = all headers unique            */
    uint64_t        read_index_base; /* global index of read 0 (multi-GPU shards)                  */
} crass_reads;

/* ---- lifecycle ---- */
/* replaces: the `options` argument of searchFile/findSingletons */
int  crass_hip_create(const crass_params *p, int device, crass_hip_ctx **out);
/* Stage timing with HIP events on the context's stream.  An event record costs ~6 us of stream time, so the
 * default (0) times nothing; 1 times the three large kernels (counters ms_filter, ms_survivor, ms_recruit); 2 adds every
 * stage (ms_compact, ms_pass1_total, ms_merge_device, ms_recruit_finish, ms_pass2_total).  (Until round 6 the default was 1:
 * six records = ~35 us of every step of every caller, 3.5 % of a rank's step at 100 M reads over 8 GPUs.)
 * Environment override at creation: CRASS_STAGE_TIMING=0|1|2.  (No reference counterpart: crass has no timers.) */
int  crass_hip_set_stage_timing(crass_hip_ctx *ctx, int level);
/* The A/B and test switches of the environment (CRASS_HOST_MERGE, CRASS_NO_SPECULATION, CRASS_DM_*, ...) are read once,
 * when the context is created — never on the per-call path.  This re-reads them for a live context (tests that flip
 * a switch between two runs of one context).  (No reference counterpart.) */
int  crass_hip_reload_env(crass_hip_ctx *ctx);
/* Level 1 only: which of the three large kernels are bracketed — bit 0 seed scan (ms_filter), bit 1 survivors
 * (ms_survivor), bit 2 pass-2 scan (ms_recruit); default 7.  A caller that reports one kernel's duration over many
 * calls (bench.py: the dominant one) times just that kernel: two event records per call instead of six. */
int  crass_hip_set_timing_focus(crass_hip_ctx *ctx, unsigned kernels);
/* Orders the context's stream behind a HIP event recorded on another stream (hipEvent_t passed as void*): lets the
 * caller's collective (RCCL all-gather on its own stream) feed crass_hip_merge_gathered without a host wait. */
int  crass_hip_stream_wait_event(crass_hip_ctx *ctx, void *event);
void crass_hip_destroy(crass_hip_ctx *ctx);
const char *crass_hip_strerror(int status);
int  crass_hip_last_hip_error(const crass_hip_ctx *ctx);

/* replaces: getFileHandle/kseq_read loops of searchFile+findSingletons (libcrispr.cpp:84-96,
 * 471-487): host buffers are copied to HBM once and stay resident for both passes. */
int crass_hip_load_reads(crass_hip_ctx *ctx, const crass_reads *host_reads);
/* same, but every pointer in `dev_reads` is a DEVICE pointer the caller owns (e.g. torch
 * tensors) and keeps alive until the context is destroyed or new reads are set. */
int crass_hip_attach_device_reads(crass_hip_ctx *ctx, const crass_reads *dev_reads);

/* ---- pass 1 : searchFile / searchCore (libcrispr.cpp:68-166, 265-395) ---- *
 * Runs the seed-scan filter, ordered compaction, and the survivor kernel (scanRight,
 * extendPreRepeat, qcFoundRepeats incl. Levenshtein, DRLowLexi) on the device.  Results are
 * kept in the context in read order.                                                       */
int crass_hip_seed_scan(crass_hip_ctx *ctx);

/* candidates found by pass 1, in read order (views into context memory, valid until the
 * next seed_scan/load).  One entry per read for which searchCore returned true.           */
typedef struct {
    uint64_t        n;
    const uint64_t *read_idx;     /* global read index (read_index_base + local)                  */
    const uint8_t  *low_lexi;     /* RH_WasLowLexi after DRLowLexi (ReadHolder.cpp:513-591)       */
    const uint32_t *repeat_len;   /* RH_RepeatLength                                              */
    const uint32_t *n_ss;         /* RH_StartStops.size()                                         */
    const uint64_t *ss_off;       /* offset of this read's start/stops in ss_pool                 */
    const uint32_t *ss_pool;      /* start/stop pairs AFTER DRLowLexi (mirrored if flipped)       */
    uint32_t        dr_stride;    /* bytes per DR slot                                            */
    const uint16_t *dr_len;
    const char     *dr_chars;     /* low-lexi representative DR of read k at dr_chars+k*dr_stride */
    uint32_t        max_read_len; /* searchFile's return value (libcrispr.cpp:98,165)             */
} crass_candidates;
int crass_hip_get_candidates(const crass_hip_ctx *ctx, crass_candidates *out);

/* ---- merge : addReadHolder token order + createNonRedundantSet (libcrispr.cpp:1119-1162,
 * StringCheck.cpp:46-81, WorkHorse.cpp:612-709, 1404-1637) ---- *
 * Input: the representative DR strings of ALL pass-1 candidates in global read order (for
 * one GPU: the context's own; for several GPUs: the all-gathered slots of every rank, rank
 * order == read order).  dr_chars == NULL means "use this context's candidates".
 * Deterministic: every rank that feeds the same list builds the same tokens, groups and
 * pattern list.  Also installs the pattern list for crass_hip_recruit().                   */
int crass_hip_merge(crass_hip_ctx *ctx, const char *dr_chars, const uint16_t *dr_len,
                    uint32_t dr_stride, uint64_t n_candidates);

/* Multi-GPU exchange in its compact form: only the DISTINCT representative DR strings of a rank's
 * candidates travel (first-occurrence order), not one string per candidate.  Scanning the ranks'
 * lists in rank order reproduces the global first-occurrence order, so tokens come out exactly
 * as if one process had seen every read (libcrispr.cpp:1137-1143).                            */
typedef struct {
    uint64_t        n_distinct;
    uint32_t        dr_stride;
    const uint16_t *dr_len;        /* [n_distinct]                                                */
    const char     *dr_chars;      /* string d at dr_chars + d*dr_stride                          */
    uint64_t        n_candidates;
    const uint32_t *cand_distinct; /* [n_candidates] index of each candidate's string in the list */
} crass_distinct;
int crass_hip_get_distinct(crass_hip_ctx *ctx, crass_distinct *out);
/* merge from the concatenation (rank order) of every rank's distinct list; `my_offset` is where this
 * context's own list starts in it.  crass_merge_view.cand_token then covers this context's candidates. */
int crass_hip_merge_distinct(crass_hip_ctx *ctx, const char *dr_chars, const uint16_t *dr_len, uint32_t dr_stride,
                             uint64_t n_global, uint64_t my_offset);

/* The same exchange with the lists left on the device (RCCL all-gather between device buffers): the device
 * addresses of this context's distinct list (CRASS_ERR_STATE when pass 1 did not produce it on the device,
 * use crass_hip_get_distinct then), and the merge from a device-resident concatenation.  The merge copies
 * the list, so the caller's buffers may be released as soon as the call returns.                          */
typedef struct {
    uint64_t        n_distinct;
    uint32_t        dr_stride;
    const char     *d_chars;       /* device pointer, n_distinct * dr_stride bytes                */
    const uint16_t *d_len;         /* device pointer                                              */
} crass_distinct_dev;
int crass_hip_get_distinct_device(crass_hip_ctx *ctx, crass_distinct_dev *out);
int crass_hip_merge_distinct_device(crass_hip_ctx *ctx, const char *d_chars, const uint16_t *d_len, uint32_t dr_stride,
                                    uint64_t n_global, uint64_t my_offset);

/* One-collective form of the exchange.  After crass_hip_exchange_setup every seed scan leaves this rank's
 * distinct list in a fixed-size device buffer: row 0 = header {uint64 n_distinct, uint32 stride, uint32
 * rows}, then `cap_rows` slots of `slot_bytes` (DR bytes, zero padded, then the uint16 length).  The caller
 * all-gathers that buffer (ncclAllGather / torch all_gather_into_tensor: world * send_bytes) and hands the
 * result to crass_hip_merge_gathered, which compacts, de-duplicates and merges on the device.
 * CRASS_ERR_OVERFLOW: some rank had more than cap_rows strings (every rank sees that in the headers) —
 * set up again with crass_hip_exchange_needed_rows() rows, repeat the seed scan and the collective.       */
typedef struct {
    void    *d_send;              /* device pointer, send_bytes bytes, owned by the context              */
    uint64_t send_bytes;          /* (cap_rows + 1) * slot_bytes                                          */
    uint32_t slot_bytes;
    uint64_t cap_rows;
} crass_exchange;
int crass_hip_exchange_setup(crass_hip_ctx *ctx, uint32_t world, uint32_t rank, uint64_t cap_rows, crass_exchange *out);
int crass_hip_merge_gathered(crass_hip_ctx *ctx, const void *d_recv);
uint64_t crass_hip_exchange_needed_rows(const crass_hip_ctx *ctx);
/* on != 0: from the next call on, crass_hip_seed_scan of a context with an exchange set up QUEUES pass 1 up to the kernel that
 * fills the send buffer and returns without waiting for it; the caller queues its collective on the context's stream
 * (crass_hip_stream) — or on a stream ordered behind it — and calls crass_hip_merge_gathered, which queues its own kernels behind
 * the collective and only then waits for pass 1's counters.  The device no longer idles between pass 1 and the exchange while
 * the host wakes up, launches the collective and the unpack (~60 us of a rank's ~1 ms step at 100 M reads over 8 GPUs).  Any
 * other call that reads pass 1's results settles the pending scan first.  A scan whose speculative launch turns out unusable
 * (more survivors than its bound, no device-resident distinct list) marks its send buffer on the device: every rank's
 * crass_hip_merge_gathered then returns CRASS_ERR_OVERFLOW exactly as for a list that did not fit (needed rows <= the capacity
 * in use), and the repeated seed scan of the marked rank runs synchronously.  Off by default; crass_hip_group_* and
 * crass_amd.distributed.GatheredExchange switch it on.  (No reference counterpart: crass is single-threaded, TODO.md:18.) */
int crass_hip_exchange_set_deferred(crass_hip_ctx *ctx, int on);
/* A row capacity for the first exchange of a job whose LARGEST shard holds n_reads reads: the bound the engine itself
 * speculates with for a shard's distinct DR strings (a few hundred per million reads on metagenome-like input, >= 16 384).
 * Every rank must use the same capacity; with this one the first step of a group neither overflows nor repeats pass 1
 * (16 384 rows, the round-3 default, did at 50 M reads per rank: 18.4 ms for the first step against 4.9 in steady state). */
uint64_t crass_hip_exchange_rows_for(uint64_t n_reads_of_largest_shard);

typedef struct {
    uint32_t        n_tokens;     /* StringCheck size; tokens are 2 .. n_tokens+1                 */
    const char     *tok_chars;    /* token t string = tok_chars[tok_off[t-2] .. tok_off[t-1])     */
    const uint64_t *tok_off;      /* [n_tokens+1]                                                 */
    uint64_t        n_candidates; /* length of cand_token                                         */
    const uint32_t *cand_token;   /* token of every candidate fed to merge, in the same order     */
    uint32_t        n_groups;     /* mDR2GIDMap size; GIDs are 1 .. n_groups                      */
    const uint32_t *grp_tokens;   /* group g (GID g+1) = grp_tokens[grp_off[g] .. grp_off[g+1])   */
    const uint64_t *grp_off;      /* [n_groups+1]                                                 */
    uint32_t        n_patterns;   /* Vecstr* returned by createNonRedundantSet                    */
    const char     *pat_chars;
    const uint64_t *pat_off;      /* [n_patterns+1]                                               */
    const uint32_t *pat_group;    /* GID of each pattern                                          */
    int32_t         next_free_gid;/* nextFreeGID after clustering (WorkHorse.cpp:369-370)         */
} crass_merge_view;
int crass_hip_get_merge(const crass_hip_ctx *ctx, crass_merge_view *out);

/* The same merge as a context-free host function (no GPU needed): used by the multi-rank
 * driver's CPU tests and by callers that only want createNonRedundantSet's result.        */
typedef struct crass_merge_handle crass_merge_handle;
int  crass_merge_create(const char *dr_chars, const uint16_t *dr_len, uint32_t dr_stride,
                        uint64_t n_candidates, int32_t kmer_clust_size, crass_merge_handle **out);
/* the host view of a merge whose per-token results (GID, dropped by removeRedundantRepeats) were computed
 * elsewhere — by the device merge in the engine; exported so that the rebuild can be tested without a GPU.
 * dx_*: the distinct DR strings in token order; cand_distinct[k] = index of candidate k's string.          */
int  crass_merge_rebuild(const char *dx_chars, const uint16_t *dx_len, uint32_t dr_stride, uint64_t n_distinct,
                         const uint32_t *cand_distinct, uint64_t n_candidates, const uint32_t *gid_of,
                         const uint8_t *dropped, uint32_t n_groups, crass_merge_handle **out);
int  crass_merge_get(const crass_merge_handle *h, crass_merge_view *out);
void crass_merge_destroy(crass_merge_handle *h);

/* ---- pass 2 : findSingletons / on_match (libcrispr.cpp:399-518) ---- */
/* replaces: refsplit + acism_create (libcrispr.cpp:452-469).  Optional: merge() already
 * installed the non-redundant set; use this to recruit with an explicit pattern list.     */
int crass_hip_set_patterns(crass_hip_ctx *ctx, const char *const *patterns,
                           const uint32_t *lengths, uint32_t n);
/* replaces: the acism_scan loop.  Reads whose header was found in pass 1 (readsFound,
 * libcrispr.cpp:411) are skipped; `extra_found` (may be NULL) lists further header ids,
 * as global read indices, found by OTHER ranks' pass 1 (duplicate headers across shards).  */
int crass_hip_recruit(crass_hip_ctx *ctx, const uint64_t *extra_found, uint64_t n_extra);

typedef struct {
    uint64_t        n;
    const uint64_t *read_idx;     /* global read index, ascending                                 */
    const uint8_t  *low_lexi;     /* RH_WasLowLexi                                                */
    const uint32_t *start;        /* the single repeat [start, end] AFTER DRLowLexi               */
    const uint32_t *end;
    uint32_t        dr_stride;
    const uint16_t *dr_len;
    const char     *dr_chars;     /* low-lexi DR (== a stored token string)                       */
    const uint32_t *token;        /* StringToken (existing, or newly added like addReadHolder)    */
} crass_recruits;
int crass_hip_get_recruits(const crass_hip_ctx *ctx, crass_recruits *out);

/* ---- Levenshtein / similarity batch (PatternMatcher.cpp:111-204) on the device ---- *
 * pair k compares chars[a_off[k] .. a_off[k]+a_len[k]) with chars[b_off[k] ..)              */
int crass_hip_levenshtein_batch(crass_hip_ctx *ctx, const char *chars, uint64_t n_chars,
                                const uint64_t *a_off, const uint32_t *a_len,
                                const uint64_t *b_off, const uint32_t *b_len,
                                uint64_t n_pairs, int32_t *dist_out, float *sim_out);

/* ---- observability (the reference only has a wall-clock progress line) ---- */
typedef struct {
    uint64_t n_reads, n_exceptions;
    uint64_t n_filter_survivors;    /* reads with a lattice seed hit (superset allowed)           */
    uint64_t n_pass1_found, n_pass2_found;
    uint32_t n_patterns, ac_states;
    uint32_t used_fast_filter;      /* 1: bit-parallel lane-per-read kernel, 2: position hints (reads of 257 .. 2 048 bases, differing strides), 0: general */
    uint32_t used_lds_automaton;    /* pass 2: 2 = anchor filter + exact scan of flagged reads,
                                       1 = automaton in LDS over all reads, 0 = automaton in L2   */
    /* HIP-event timings of the last call, milliseconds, measured on the context's stream      */
    float ms_filter, ms_compact, ms_survivor, ms_pass1_total;
    float ms_recruit, ms_recruit_finish, ms_pass2_total;
    float ms_merge_host, ms_sink_host;
    uint64_t bytes_reads_device;    /* packed read bytes resident in HBM                          */
    uint32_t anchor_keys;           /* distinct 16-mer anchor keys of the pattern set (pass 2)    */
    uint32_t anchor_table_kind;     /* 0 exact keys in LDS, 1 fingerprint buckets in LDS,
                                       2 exact keys probed in L2 (key set beyond LDS)             */
    uint32_t used_device_merge;     /* 1: clustering / non-redundant set / pass-2 index built on the
                                       device (dmerge.hip), 0: host merge (merge.cpp)             */
    float ms_merge_device;          /* HIP-event time of the device merge kernels                 */
    uint32_t n_merge_fallbacks;     /* since crass_hip_create: device merges that gave up (key set beyond the table,
                                       cuckoo insertion, a wait that timed out, ...) and were redone on the host     */
    uint32_t last_fallback_bits;    /* the device merge's `fail` word of the last such case (0: none so far)         */
    uint32_t n_bound_overflows[4];  /* since crass_hip_create: stages repeated because a speculation bound (sized from the read
                                       set at load, or learnt from the previous call) was too small: [0] seed-scan survivors,
                                       [1] distinct DR strings (queued merge), [2] reads flagged in pass 2, [3] gathered
                                       distinct strings (multi-rank).  Results are unaffected; each is one repeated stage.  */
    uint32_t used_device_view;      /* 1: crass_merge_view's arrays (tokens, groups, pattern list) of the last merge were
                                       assembled on the device and copied by a DMA engine; 0: built by the host            */
    uint32_t n_view_fallbacks;      /* since crass_hip_create: device merges whose view the host built after all (a group
                                       with more members than the export ranks on the device)                              */
} crass_counters;
int crass_hip_get_counters(const crass_hip_ctx *ctx, crass_counters *out);

/* ---- several GPUs from ONE process (north_star: "C++ host code ... RCCL all-gather over xGMI") ----
 * A group owns one context per device.  The reads shard by contiguous ranges in input order (rank r holds reads
 * [r*n/N, (r+1)*n/N), WorkHorse.cpp:336-393: pass 1 covers every read before pass 2 starts), every rank runs pass 1 on its
 * shard, ONE collective — ncclAllGather inside ncclGroupStart/ncclGroupEnd on the contexts' streams, communicators from
 * ncclCommInitAll — moves every rank's distinct candidate DR strings to every rank (the fixed-size buffers of
 * crass_hip_exchange_setup), each rank merges the rank-ordered concatenation on its device (identical tokens, groups,
 * pattern set and pass-2 index everywhere: no broadcast of tables) and recruits on its shard.  The host view of the
 * merge (crass_merge_view) is built ONCE for the group, by rank 0.  There is no reference counterpart: crass is a
 * single-threaded program; what the group replaces is the same three calls of WorkHorse::parseSeqFiles
 * (searchFile / createNonRedundantSet / findSingletons, WorkHorse.cpp:340,370,386) over all reads.
 * flags: CRASS_GROUP_LOCAL_COPIES replaces the collective by plain device copies (required when a device is listed more
 * than once — RCCL refuses duplicate devices — which is how the group is tested on a one-GPU box).
 * Duplicate headers (readsFound is keyed by header, libcrispr.cpp:138,411) are honoured across shards: pass-1 hits of a
 * header that also occurs in another shard are marked found there before pass 2.
 * Call order as for a context: load_reads -> seed_scan -> merge -> recruit (or step = the three in one dispatch).
 * One host thread calls the group; the group runs one helper thread per additional rank.                              */
typedef struct crass_hip_group crass_hip_group;
#define CRASS_GROUP_LOCAL_COPIES 1u
int  crass_hip_group_create(const crass_params *p, const int *devices, int n_devices, unsigned flags, crass_hip_group **out);
void crass_hip_group_destroy(crass_hip_group *g);
int  crass_hip_group_size(const crass_hip_group *g);
/* ranks of the RCCL communicator (ncclCommCount), 0 when the collective is the local-copy stand-in */
int  crass_hip_group_rccl_ranks(const crass_hip_group *g);
/* text of the last RCCL / group error of this process ("" if none); valid until the next group call */
const char *crass_hip_group_last_error(void);
/* rank r's context, for the per-rank getters (crass_hip_get_counters, crass_hip_get_candidates, ...); owned by the group */
crass_hip_ctx *crass_hip_group_ctx(crass_hip_group *g, int rank);
int  crass_hip_group_load_reads(crass_hip_group *g, const crass_reads *host_reads);
int  crass_hip_group_seed_scan(crass_hip_group *g);
int  crass_hip_group_merge(crass_hip_group *g);
/* extra_found: further found headers as job-level read indices (crass_hip_recruit's argument), routed to the shards */
int  crass_hip_group_recruit(crass_hip_group *g, const uint64_t *extra_found, uint64_t n_extra);
/* an explicit pattern list on every rank instead of the group's own merge (the seam's findSingletons, whose pattern list
 * comes from the caller's createNonRedundantSet; crass_hip_set_patterns).  Found headers of OTHER shards are then the
 * caller's to pass (extra_found): it holds readsFound.                                                          */
int  crass_hip_group_set_patterns(crass_hip_group *g, const char *const *patterns, const uint32_t *lengths, uint32_t n);
int  crass_hip_group_step(crass_hip_group *g);         /* seed_scan + merge + recruit, one dispatch per rank */
/* the hand-off of the whole job, in global read order (rank order == read order): views into group-owned memory, valid
 * until the next group call.  get_merge: tokens / groups / patterns from rank 0's host view, cand_token for the
 * candidates of all ranks (the order of crass_hip_group_get_candidates).                                          */
int  crass_hip_group_get_candidates(crass_hip_group *g, crass_candidates *out);
int  crass_hip_group_get_merge(crass_hip_group *g, crass_merge_view *out);
int  crass_hip_group_get_recruits(crass_hip_group *g, crass_recruits *out);
/* Which host view a context builds after a device merge: 0 (default) the full one, 1 only its own candidates' tokens
 * (ranks other than 0 of a group: tokens, groups and patterns are identical on every rank and are read from rank 0). */
int  crass_hip_set_host_view(crass_hip_ctx *ctx, int light);

/* ---- the stage right behind the hot path: true-DR consensus + start/stop repair ----
 * replaces: int WorkHorse::findConsensusDRs(GroupKmerMap&, int& nextFreeGID)  (WorkHorse.cpp:578-611, called at :403), i.e.
 * per DR group parseGroupedDRs (:1135-1379): Aligner (Aligner.cpp:73-468 over ksw_align, ksw.c:330-360) -> consensus and
 * possible split of the group (calculateDRConsensus :801-938, splitGroupedDR :940-1132) -> ReadHolder::updateStartStops
 * for every read (ReadHolder.cpp:382-511: smithWaterman, SmithWaterman.cpp:151-308, with its Levenshtein filter :283),
 * and combineGroupsWithIdenticalDRs (:416-452) after every group.
 * Input = the hand-off of the search path as flat arrays; mReads[token] = the records of that token in record order
 * (pass-1 records of all files, then pass-2 records: the push order of addReadHolder, libcrispr.cpp:1161).           */
typedef struct {
    const char *seqs; const uint64_t *seq_off; uint64_t n_reads;     /* the input reads (as read from the files)        */
    uint64_t n_rec;                                                  /* records = ReadHolders                           */
    const uint64_t *rec_read;        /* index into seqs                                                                 */
    const uint8_t  *rec_lowlexi;     /* RH_WasLowLexi: 0 = the holder's RH_Seq is the reverse complement of the read    */
    const uint32_t *rec_token;       /* StringToken of the holder's ReadList                                            */
    const uint32_t *rec_nss; const uint64_t *rec_ss_off; const uint32_t *ss_pool;    /* RH_StartStops                   */
    uint32_t n_tokens; const char *tok_chars; const uint64_t *tok_off;               /* StringCheck: token t = entry t-2 */
    uint32_t n_groups; const uint32_t *grp_tokens; const uint64_t *grp_off;          /* mDR2GIDMap: GID g = entry g-1    */
    uint32_t max_read_len;                                                           /* mMaxReadLength                   */
} crass_cons_input;

typedef struct {
    uint64_t n_groups_parsed;        /* parseGroupedDRs calls that reached the coverage stage (incl. split sub-groups)  */
    uint64_t n_ksw_launches, n_ksw_alignments, n_placements, n_flips, n_true_drs, n_sw_tasks, n_partials_added;
} crass_counters_cons;

typedef struct {
    int32_t  error;                  /* != 0: the reference would have thrown / crashed / not terminated on this input  */
    int32_t  next_free_gid;
    uint32_t n_tokens;               /* StringCheck after the stage (reversed slaves and split forms add tokens)        */
    const char *tok_chars; const uint64_t *tok_off;
    uint32_t n_groups;               /* groups with a true DR (mTrueDRs), ascending GID                                 */
    const int32_t *grp_gid; const char *dr_chars; const uint64_t *dr_off;           /* GID, laurenized true DR          */
    const uint32_t *grp_tokens; const uint64_t *grp_off;                            /* mDR2GIDMap[GID], in order        */
    uint64_t n_rec;                  /* per input record:                                                               */
    const uint8_t  *rec_alive;       /* 0: the ReadHolder was deleted                                                   */
    const uint8_t  *rec_rc;          /* 1: RH_Seq is the reverse complement of the input read                           */
    const uint32_t *rec_token;       /* the token whose ReadList holds the record (0: none)                             */
    const uint32_t *rec_nss; const uint64_t *rec_ss_off; const uint32_t *ss_pool;   /* repaired RH_StartStops           */
    const uint64_t *tokread_off; const uint64_t *tokread_idx;                       /* mReads[token]: records in order  */
    const uint8_t  *tok_has_list;
    crass_counters_cons counters;
} crass_cons_view;

typedef struct crass_cons crass_cons;
/* runs the whole stage on `device`; CRASS_OK with view.error != 0 when the reference itself would not have survived the
 * input (the view is then incomplete)                                                                                  */
int  crass_hip_consensus(const crass_params *p, int device, const crass_cons_input *in, crass_cons **out);
int  crass_hip_consensus_view(const crass_cons *c, crass_cons_view *v);
void crass_hip_consensus_free(crass_cons *c);

/* ---- the stages behind findConsensusDRs: spacer graph + crass's output files ----
 * replaces, for a caller that wants crass's outputs from the hand-off: WorkHorse::buildGraph / cleanGraph / makeSpacerGraphs /
 * cleanSpacerGraphs / splitIntoContigs / generateFlankers / removeLowConfidenceNodeManagers (WorkHorse.cpp:196-277 ->
 * NodeManager.cpp, CrisprNode.cpp, SpacerInstance.cpp) and WorkHorse::outputResults (WorkHorse.cpp:1900-2249: crass.crispr
 * through crispr::xml::writer, Group_<gid>_<DR>.fa through NodeManager::dumpReads, Spacers_<gid>_<DR>_spacers.gv and
 * crass.<timestamp>.keys.gv through printSpacerGraph / printSpacerKey).  Host code (<= 10^4 reads per group; serial graph
 * work), no GPU needed.  Input: the groups that have a true DR after crass_hip_consensus, ascending GID, and for each the
 * ReadHolders in buildGraph's order (WorkHorse.cpp:454-505: for every token of mDR2GIDMap[GID], mReads[token] in list
 * order) with RH_Seq as it stands after the consensus stage.
 * The reference's own output depends on heap addresses where a graph has forks or bubbles (edge maps keyed by CrisprNode*):
 * creation order is used there (oracle/crass_graph.py: parity unpinned for these two rows).            */
typedef struct {
    uint32_t n_groups; const int32_t *gid; const char *dr_chars; const uint64_t *dr_off;      /* true DR of group g            */
    const uint64_t *grp_rec_off;                                  /* [n_groups+1]: records of group g, in buildGraph's order  */
    uint64_t n_rec;
    const char *hdr_chars; const uint64_t *hdr_off;               /* RH_Header                                                  */
    const char *com_chars; const uint64_t *com_off;               /* RH_Comment (may be NULL: no comments)                      */
    const char *seq_chars; const uint64_t *seq_off;               /* RH_Seq                                                     */
    const uint32_t *rec_nss; const uint64_t *rec_ss_off; const uint32_t *ss_pool;              /* RH_StartStops                 */
} crass_graph_input;
typedef struct {
    const char *out_dir;          /* options::output_fastq as crass holds it (with the trailing '/'), only used in names/urls  */
    const char *timestamp;        /* mTimeStamp, crass.cpp:476-479 ("%d_%m_%Y_%H%M%S")                                          */
    const char *command_line;     /* mCommandLine, crass.cpp:508-512 (every argv followed by a blank)                           */
    const char *cwd;              /* getcwd() (WorkHorse.cpp:2104)                                                              */
    int32_t log_to_screen;        /* options::logToScreen: no <file type="log"> entry                                           */
    int32_t cov_cutoff;           /* options::covCutoff, default 3 (crassDefines.h:126); <= 0: default                         */
    int32_t node_kmer;            /* options::cNodeKmerLength, default 7 (crassDefines.h:111); <= 0: default                   */
    int32_t show_singles, long_description;                       /* options::showSingles / longDescription, default 0         */
} crass_output_opts;
typedef struct crass_outputs crass_outputs;
typedef struct {
    uint32_t n_files; const char *const *name; const char *const *data; const uint64_t *size;  /* file contents in memory      */
    uint32_t n_groups_kept; const int32_t *kept_gid;              /* the groups in crass.crispr                                 */
    const char *stdout_text;                                      /* what crass prints to stdout in these stages                */
} crass_outputs_view;
int  crass_build_outputs(const crass_graph_input *in, const crass_output_opts *opts, crass_outputs **out);
int  crass_outputs_get(const crass_outputs *o, crass_outputs_view *v);
int  crass_outputs_write(const crass_outputs *o, const char *dir);                              /* every file into dir          */
void crass_outputs_free(crass_outputs *o);

/* raw stream handle (hipStream_t) the context launches on, for callers that time with HIP
 * events or want to order their own work (torch.cuda.ExternalStream) */
void *crass_hip_stream(const crass_hip_ctx *ctx);

/* ---- host utilities: ingest side of the boundary ("next" row f-2) ---- */
/* 2-bit packer.  seqs: concatenated raw bytes, off[n+1].  Allocates the output arrays with
 * malloc (free with crass_free_packed).  pad_uniform: 0 = per-read word offsets, 1 = every
 * read padded to stride_words = ceil(max_len/16), 2 = pad when the reads are short (64..256
 * bases) and padding costs at most twice the words (trimmed reads of differing lengths then
 * take the uniform-stride kernels).                                                        */
typedef struct {
    crass_reads reads;              /* host pointers, owned by this struct                        */
    void *owner;
} crass_packed;
int  crass_pack_reads(const uint8_t *seqs, const uint64_t *off, uint64_t n_reads,
                      int pad_uniform, crass_packed *out);
void crass_free_packed(crass_packed *p);

/* FASTA/FASTQ(.gz) reader with kseq_read record semantics (kseq.cpp:171-226).  Buffers are
 * malloc'd, free with crass_free_fastx.                                                    */
typedef struct {
    uint64_t  n_reads;
    uint8_t  *seq;   uint64_t *seq_off;     /* [n+1] */
    uint8_t  *name;  uint64_t *name_off;
    uint8_t  *comment; uint64_t *comment_off; uint8_t *has_comment; /* stale-pointer semantics,
                                                                       libcrispr.cpp:124-127 */
    uint8_t  *qual;  uint64_t *qual_off;   uint8_t *has_qual;
    uint64_t *header_id;                    /* first read with the same name                     */
    uint32_t  max_len;
    int32_t   last_ret;                     /* kseq_read's final return value (-1 EOF, -2 trunc) */
    uint64_t *name_index; uint64_t name_index_cap;   /* open-addressing table name -> first read (crass_fastx_find) */
} crass_fastx;
int  crass_read_fastx(const char *path, crass_fastx *out);
void crass_free_fastx(crass_fastx *f);
/* index of the FIRST read with this header name (the key of readsFound, libcrispr.cpp:138,411), UINT64_MAX if none */
uint64_t crass_fastx_find(const crass_fastx *f, const char *name, uint64_t len);

/* The same reader as an INDEX over an input kept mapped (plain text) or inflated once (gzip, through libdeflate when the runtime
 * library is there): the records are parsed by the same state machine, every read is
 * 2-bit packed at once (crass_pack_reads' layout rules, mode 2) and its text dropped; what stays is the packed reads, the
 * position of every record's header character, and header_id (first read with the same header, names compared in the mapping:
 * exact).  crass_fastx_index_fetch parses the records that are handed on (the ~1 % that pass 1 / pass 2 find) when they are
 * asked for.  replaces: the getFileHandle / kseq_read loops of searchFile + findSingletons (libcrispr.cpp:84-131, 471-487) and
 * the ReadHolder fields they fill (seq, header, comment, quality) for the reads that reach addReadHolder.  Host memory: 40 bytes
 * of words + 12 of bookkeeping per 150 bp read instead of ~330.  CRASS_ERR_UNSUPPORTED: a gzip'd input without libdeflate or whose text would not fit half the available memory, a file that mixes
 * records with and without a comment / quality line (kseq's stale buffers, libcrispr.cpp:124-131, need the records in order), a
 * file that cannot be mapped — the two readers around this one take those.                                                      */
typedef struct crass_fastx_index crass_fastx_index;
int  crass_index_fastx(const char *path, crass_fastx_index **out);
/* ... over SEVERAL inputs (paired-end files, lanes): the job's reads in (file, read) order in one packed set, header ids across the
 * inputs (the first read of the JOB with the same header: readsFound is keyed by the header string whatever file it came from,
 * libcrispr.cpp:138,411), every input plain text or gzip'd on its own.  max_len / last_ret of crass_fastx_index_reads: the longest
 * read of all inputs, kseq_read's final return value on the LAST one.                                                              */
int  crass_index_fastx_files(const char *const *paths, uint32_t n_paths, crass_fastx_index **out);
/* the packed reads (host pointers owned by the index), the longest read, kseq_read's final return value */
int  crass_fastx_index_reads(const crass_fastx_index *ix, crass_reads *reads, uint32_t *max_len, int *last_ret);
/* records idx[0 .. n) (any order) as a crass_fastx of n records in that order; header_id[k] = idx[k]; free with crass_free_fastx */
int  crass_fastx_index_fetch(const crass_fastx_index *ix, const uint64_t *idx, uint64_t n, crass_fastx *out);
void crass_fastx_index_free(crass_fastx_index *ix);
/* the mappings of the plain-text inputs given back now, over the cores (page tables of gigabytes are slow to take down and block the
 * process's other allocations meanwhile); records fetched afterwards are read from the files — a hand-off fetches what it needs first */
void crass_fastx_index_drop_text(crass_fastx_index *ix);

/* The same reader as a STREAM of chunks, for inputs that should not be held in host memory as a whole — the reference's own
 * memory model: kseq_read hands out one record at a time (kseq.cpp:171-226, libcrispr.cpp:96) and crass reads every input
 * twice, once per pass (WorkHorse.cpp:336-393).  A chunk holds the complete records of about `chunk_bytes` of decompressed text
 * (0: 64 MB; a record larger than a chunk grows it); the fields are those of crass_fastx, valid until the next call; n_reads == 0
 * marks the end of the file.  kseq's cross-record state travels with the stream (stale comment / quality buffers,
 * libcrispr.cpp:124-131).  With a name table, header_id[i] is the JOB-level index (index_base + records before it in this
 * stream) of the first read with the same header — across chunks and, with one table for several streams, across files
 * (readsFound is keyed by the header string, libcrispr.cpp:138,411); names are kept as 128-bit hashes only (two independent
 * 64-bit mixes, both stored and compared in full): two DIFFERENT headers of a job of n reads are taken for one with probability
 * ~ n^2 / 2^129 (1.5e-23 at n = 1e8) — the one place of the path that is exact with that probability rather than by
 * construction; crass_read_fastx (whole-file) compares the header strings themselves.                                       */
typedef struct crass_name_table crass_name_table;
typedef struct crass_fastx_stream crass_fastx_stream;
crass_name_table *crass_name_table_create(void);
void     crass_name_table_destroy(crass_name_table *t);
void     crass_name_table_reserve(crass_name_table *t, uint64_t n_names);    /* sized once for about n_names names (optional) */
uint64_t crass_name_table_first(crass_name_table *t, const char *name, uint64_t len, uint64_t index);
int      crass_fastx_stream_open(const char *path, uint64_t chunk_bytes, crass_name_table *names, uint64_t index_base,
                                 crass_fastx_stream **out);
int      crass_fastx_stream_next(crass_fastx_stream *s, crass_fastx *chunk);
uint64_t crass_fastx_stream_reads_done(const crass_fastx_stream *s);
uint32_t crass_fastx_stream_max_len(const crass_fastx_stream *s);
void     crass_fastx_stream_close(crass_fastx_stream *s);

/* deterministic synthetic metagenome: counter-based, so any shard can be
 * generated independently.  Writes 2-bit packed reads with uniform stride ceil(L/16).      */
typedef struct {
    uint64_t seed;
    uint32_t read_len;          /* 150                                                        */
    uint32_t n_dr;              /* 50 (config 2/3), 500 (config 5)                            */
    uint32_t dr_len_min, dr_len_max;       /* 28..37                                          */
    uint32_t spacer_len_min, spacer_len_max; /* 30..38                                        */
    uint32_t crispr_per_million; /* 10000 = 1 %                                               */
    uint32_t gc_classes;        /* 0/1 = uniform; 4 = GC 30/45/55/70 % background mix         */
    uint32_t array_min_repeats, array_max_repeats; /* 0,0 = short-read mode (the read is a cut
                                   from a tiled DR+spacer stream); else long-read mode (config 4:
                                   20..60): an array of that many units written into a random read */
} crass_synth_spec;
void crass_synth_default(crass_synth_spec *s);
int  crass_synth_packed(const crass_synth_spec *s, uint64_t first_read, uint64_t n_reads,
                        uint32_t *packed_out /* n_reads*ceil(L/16) words */, int n_threads);
/* unpack reads [first, first+n) of a uniform-stride packed buffer to ASCII (for the CPU baseline) */
int  crass_unpack_ascii(const uint32_t *packed, uint32_t stride_words, uint32_t read_len,
                        uint64_t n_reads, uint8_t *ascii_out /* n_reads*read_len */);

int crass_hip_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CRASS_HIP_H */
