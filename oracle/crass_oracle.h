/*
 * crass_oracle.h — CPU restatement (plain C) of the crass v1.0.1 search hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported CPU baseline.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the reference tree, src/crass/... and src/aho-corasick/...).
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - leaf functions (bmpSearch, Levenshtein, similarity, ACISM first-callback,
 *     StringCheck) are checked against the *compiled reference sources*
 *     (oracle/_ref, built by oracle/Makefile from /root/reference);
 *   - scanRight / extendPreRepeat against the reference's own Catch known-answer
 *     tests (tests/golden/kat_libcrispr.json, 139 assertions);
 *   - the whole pipeline against the known answers SURVEY.md §8c recorded from the
 *     compiled reference (read / variant / group / pattern counts on the five
 *     test/ *.gz inputs and on the seeded 1 M-read synthetic set).
 *   libcrispr.cpp / ReadHolder.cpp / WorkHorse.cpp themselves cannot be compiled
 *   here without generated code (autoconf config.h) and Xerces-C headers, so they
 *   are not part of oracle/_ref.
 */
#ifndef CRASS_ORACLE_H
#define CRASS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirror of the hot-path fields of `options` (crassDefines.h:140-170) */
typedef struct {
    uint32_t lowDRsize;          /* -d, default 23 */
    uint32_t highDRsize;         /* -D, default 47 */
    uint32_t lowSpacerSize;      /* -s, default 26 */
    uint32_t highSpacerSize;     /* -S, default 50 */
    uint32_t searchWindowLength; /* -w, default 8 (6..9) */
    uint32_t minNumRepeats;      /* -n, default 2 */
    int32_t  kmer_clust_size;    /* -k, default 6 */
} orc_params;

void orc_default_params(orc_params *p);

/* ---- leaf functions ---------------------------------------------------- */
/* PatternMatcher::bmpSearch, PatternMatcher.cpp:26-59 */
int orc_bmp_search(const char *text, size_t tlen, const char *pat, size_t plen);
/* PatternMatcher::levenstheinDistance, PatternMatcher.cpp:111-195 */
int orc_levenshtein(const char *s, int n, const char *t, int m);
/* PatternMatcher::getStringSimilarity, PatternMatcher.cpp:197-204 */
float orc_similarity(const char *s, int n, const char *t, int m);
/* reverseComplement, SeqUtils.cpp:50-87 (out must hold n bytes) */
void orc_revcomp(const char *in, size_t n, char *out);
/* isRepeatLowComplexity, libcrispr.cpp:1031-1069 */
int orc_is_low_complexity(const char *rep, int n);

/* ---- per-read functions (start/stop list is in/out) -------------------- */
/* ReadHolder::startStopsAdd clamp, ReadHolder.cpp:263-297 */
/* scanRight, libcrispr.cpp:170-263.  ss has *nss entries, capacity cap. */
int orc_scan_right(const char *seq, int L, uint32_t *ss, int *nss, int cap,
                   const char *pat, int plen, uint32_t minSpacer, uint32_t scanRange);
/* extendPreRepeat, libcrispr.cpp:520-772. returns repeat length */
uint32_t orc_extend_pre_repeat(const char *seq, int L, uint32_t *ss, int nss,
                               int window, int minSpacer);
/* qcFoundRepeats, libcrispr.cpp:869-1029. 1 pass, 0 fail, <0 = reference would throw */
int orc_qc_found_repeats(const char *seq, int L, const uint32_t *ss, int nss,
                         int minSpacer, int maxSpacer);
/* searchCore, libcrispr.cpp:265-395. returns 1/0 (<0 error); on 1 fills ss/nss/repeat_len */
int orc_search_core(const char *seq, int L, const orc_params *p,
                    uint32_t *ss, int *nss, int cap, uint32_t *repeat_len);
/* "does searchCore's seed loop find ANY hit on the stride lattice" (the device filter's contract) */
int orc_has_lattice_hit(const char *seq, int L, const orc_params *p);
/* ReadHolder::DRLowLexi + reverseComplementSeq + reverseStartStops,
 * ReadHolder.cpp:513-609,321-380.  seq (L bytes) and ss are rewritten in place when the
 * read is flipped.  dr_out must hold L bytes.  returns dr length; *was_low_lexi as RH_WasLowLexi */
int orc_dr_low_lexi(char *seq, int L, uint32_t *ss, int nss, char *dr_out, int *was_low_lexi);

/* ---- multi-pattern first match (findSingletons/on_match + ACISM semantics) -------- */
typedef struct orc_ac orc_ac;
/* acism_create semantics (acism_create.c:71-131): byte-wise Aho-Corasick */
orc_ac *orc_ac_create(const char *const *pats, const uint32_t *lens, uint32_t n);
void orc_ac_destroy(orc_ac *ac);
/* first callback of acism_scan (acism.c:25-106 via libcrispr.cpp:503,441):
 * returns 1 and (*end_excl, *len) for the occurrence with the smallest end position,
 * ties -> longest pattern; 0 if no pattern occurs. */
int orc_ac_first_match(const orc_ac *ac, const char *text, size_t tlen,
                       uint32_t *end_excl, uint32_t *len);

/* ---- whole pipeline: searchFile* -> createNonRedundantSet -> findSingletons* -------- */
typedef struct orc_result orc_result;

/* reads: concatenated bytes, seq_off[n+1].  headers: concatenated bytes, hdr_off[n+1]
 * (hdr may be NULL => every read has a unique header).  do_pass2 = 0 stops after merge. */
orc_result *orc_pipeline_run(const char *seqs, const uint64_t *seq_off, uint64_t n_reads,
                             const char *hdrs, const uint64_t *hdr_off,
                             const orc_params *p, int do_pass2);
void orc_result_free(orc_result *r);

typedef struct {
    uint64_t n_pass1;        /* records found by pass 1 (searchFile)            */
    uint64_t n_pass2;        /* records recruited by pass 2 (findSingletons)    */
    uint32_t n_tokens;       /* StringCheck entries (tokens 2 .. n_tokens+1)    */
    uint32_t n_groups;       /* mDR2GIDMap size                                 */
    uint32_t n_patterns;     /* non-redundant pattern list incl. revcomps       */
    uint32_t max_read_len;   /* searchFile return value                         */
    int32_t  error;          /* !=0: the reference would have thrown            */
    /* per-record SoA, pass-1 records first, each block in read order */
    const uint64_t *rec_read;     /* read index                                 */
    const uint8_t  *rec_lowlexi;  /* RH_WasLowLexi                              */
    const uint32_t *rec_token;    /* StringToken                                */
    const uint32_t *rec_replen;   /* RH_RepeatLength (0 for pass-2 records)     */
    const uint32_t *rec_nss;      /* RH_StartStops.size()                       */
    const uint64_t *rec_ss_off;   /* offset into ss_pool                        */
    const uint32_t *ss_pool;
    /* token table: token t (>=2) string = tok_chars[tok_off[t-2] .. tok_off[t-1]) */
    const char     *tok_chars;
    const uint64_t *tok_off;
    /* groups: group g (GID g+1) tokens = grp_tokens[grp_off[g] .. grp_off[g+1]) */
    const uint32_t *grp_tokens;
    const uint64_t *grp_off;
    /* patterns in the oracle's canonical order (per group: survivors by (len, token
     * order), then their reverse complements); pat_group[i] = GID */
    const char     *pat_chars;
    const uint64_t *pat_off;
    const uint32_t *pat_group;
} orc_view;

void orc_result_view(const orc_result *r, orc_view *v);

/* timing helper for bench.py's cpu_baseline leg: runs pass1+merge+pass2 and
 * returns seconds for each stage (monotonic clock) */
int orc_pipeline_time(const char *seqs, const uint64_t *seq_off, uint64_t n_reads,
                      const orc_params *p, double *t_pass1, double *t_merge, double *t_pass2,
                      uint64_t *n_pass1, uint64_t *n_pass2, uint32_t *n_patterns);

/* ---- the stage behind the hot path (crass_consensus.c): findConsensusDRs ---------------------- */
/* ksw_align as Aligner calls it (ksw.c:330-360; 16-bit scores); query / target are restored on return */
void orc_ksw_align(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat, int gapo, int gape, int xtra,
                   int *score, int *te, int *qe, int *tb, int *qb);
/* smithWaterman (SmithWaterman.cpp:151-308): 1 = (a_ret, b_ret) returned, 0 = ("", ""), < 0 = the reference would throw;
 * a_ret = seqA[*a_off, +*a_len), b_ret = seqB[*b_off, +*b_len) */
int orc_smith_waterman(const char *seqA, int lenA, const char *seqB, int lenB, int *aStartAlign, int *aEndAlign,
                       int aStartSearch, int aSearchLen, double similarity, int *a_off, int *a_len, int *b_off, int *b_len);

/* the hand-off of the search path, as flat arrays (mReads order = record order: pass-1 records, then pass-2 records) */
typedef struct {
    const char *seqs; const uint64_t *seq_off; uint64_t n_reads;          /* the input reads */
    uint64_t n_rec; const uint64_t *rec_read; const uint8_t *rec_lowlexi; const uint32_t *rec_token;
    const uint32_t *rec_nss; const uint64_t *rec_ss_off; const uint32_t *ss_pool;
    uint32_t n_tokens; const char *tok_chars; const uint64_t *tok_off;      /* StringCheck: token t = entry t - 2 */
    uint32_t n_groups; const uint32_t *grp_tokens; const uint64_t *grp_off; /* mDR2GIDMap: GID g = entry g - 1 */
    uint32_t max_read_len;                                                  /* mMaxReadLength */
} orc_cons_input;

typedef struct {
    int32_t  error;            /* != 0: the reference would have thrown / crashed / not terminated on this input */
    int32_t  next_free_gid;
    uint32_t n_tokens;         /* StringCheck after the stage (reversed slaves and split forms add tokens) */
    const char *tok_chars; const uint64_t *tok_off;
    uint32_t n_groups;         /* groups with a true DR (mTrueDRs), ascending GID */
    const int32_t *grp_gid; const char *dr_chars; const uint64_t *dr_off;   /* GID and laurenized true DR */
    const uint32_t *grp_tokens; const uint64_t *grp_off;                    /* mDR2GIDMap[GID] in order */
    uint64_t n_rec;            /* per input record: */
    const uint8_t *rec_alive;  /* 0: the ReadHolder was deleted */
    const uint8_t *rec_rc;     /* 1: RH_Seq is the reverse complement of the input read */
    const uint32_t *rec_token; /* the token whose ReadList holds the record (0: none) */
    const uint32_t *rec_nss; const uint64_t *rec_ss_off; const uint32_t *ss_pool;   /* repaired RH_StartStops */
    const uint64_t *tokread_off; const uint64_t *tokread_idx;               /* mReads[token]: record ids in order */
    const uint8_t *tok_has_list;
} orc_cons_view;

typedef struct orc_cons orc_cons;
orc_cons *orc_consensus_run(const orc_cons_input *in, const orc_params *p);
void orc_consensus_view(const orc_cons *c, orc_cons_view *v);
void orc_consensus_free(orc_cons *c);

/* calibration loops (see crass_oracle.c): seconds spent, *checksum over the results */
double orc_calib_bmp(const char *seqs, uint64_t n_reads, int L, const orc_params *p, uint64_t *checksum);
double orc_calib_ac(const orc_ac *ac, const char *seqs, uint64_t n_reads, int L, uint64_t *checksum);

#ifdef __cplusplus
}
#endif
#endif
