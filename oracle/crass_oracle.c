/*
 * crass_oracle.c — CPU restatement (plain C) of the crass v1.0.1 search hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see crass_oracle.h).  Same algorithm class as the
 * reference on purpose (per-window Boyer-Moore, full-matrix Levenshtein, byte-wise
 * Aho-Corasick), because it doubles as bench.py's reported single-core CPU baseline.
 *
 * Citations are reference paths relative to /root/reference.
 * Compile with -ffp-contract=off (float evaluation order matters, qcFoundRepeats).
 */
#include "crass_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#include <math.h>

/* ------------------------------------------------------------------------- */
/* small utilities                                                            */
/* ------------------------------------------------------------------------- */

static void *xmalloc(size_t n) { void *p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "oracle: OOM\n"); abort(); } return p; }
static void *xcalloc(size_t n, size_t m) { void *p = calloc(n ? n : 1, m ? m : 1); if (!p) { fprintf(stderr, "oracle: OOM\n"); abort(); } return p; }
static void *xrealloc(void *q, size_t n) { void *p = realloc(q, n ? n : 1); if (!p) { fprintf(stderr, "oracle: OOM\n"); abort(); } return p; }

void orc_default_params(orc_params *p)
{
    /* crass.cpp:430-460, crassDefines.h:56,67,91,121-124 */
    p->lowDRsize = 23; p->highDRsize = 47;
    p->lowSpacerSize = 26; p->highSpacerSize = 50;
    p->searchWindowLength = 8; p->minNumRepeats = 2; p->kmer_clust_size = 6;
}

/* complement table, SeqUtils.cpp:50-59: IUPAC pairs, U->A, everything else identity,
 * plus the table's one oddity: entry 96 ('`') holds 64 ('@'). Built, not transcribed. */
static unsigned char g_comp[128];
static int g_comp_ready = 0;
static void comp_init(void)
{
    if (g_comp_ready) return;
    for (int i = 0; i < 128; i++) g_comp[i] = (unsigned char)i;
    const char *a = "ACBDKRSWN", *b = "TGVHMYSWN";
    for (int i = 0; a[i]; i++) {
        g_comp[(int)a[i]] = (unsigned char)b[i]; g_comp[(int)b[i]] = (unsigned char)a[i];
        g_comp[(int)a[i] + 32] = (unsigned char)(b[i] + 32); g_comp[(int)b[i] + 32] = (unsigned char)(a[i] + 32);
    }
    g_comp['U'] = 'A'; g_comp['u'] = 'a';
    g_comp[96] = 64;
    g_comp_ready = 1;
}

/* reverseComplement, SeqUtils.cpp:61-87 (bytes >= 128 are UB there; masked here) */
void orc_revcomp(const char *in, size_t n, char *out)
{
    comp_init();
    for (size_t i = 0; i < n; i++)
        out[i] = (char)g_comp[(unsigned char)in[n - 1 - i] & 127];
}

/* std::string operator< on raw bytes */
static int str_less(const char *a, size_t na, const char *b, size_t nb)
{
    size_t m = na < nb ? na : nb;
    int c = memcmp(a, b, m);
    if (c) return c < 0;
    return na < nb;
}

/* ------------------------------------------------------------------------- */
/* PatternMatcher                                                             */
/* ------------------------------------------------------------------------- */

/* PatternMatcher.cpp:26-59 + computeBmpLast :99-109 */
/* work counters for tools/longread_phases.py only, compiled in with -DORC_WORK_COUNTERS (the tool builds its own copy of this
 * file; per thread).  The library the tests and bench.py time carries none: in round 5 they were plain globals, and 64 threads of
 * the all-cores baseline writing one cache line twelve times per read ran at 1.5 x one core.
 * [0] bmpSearch calls, [1] of them hits inside searchCore's seed loop, [2] calls from scanRight, [3] extension columns voted on,
 * [4] column votes x repeats (bases read), [5] similarity (Levenshtein) calls, [6] qcFoundRepeats calls, [7] Levenshtein cells */
#ifdef ORC_WORK_COUNTERS
static _Thread_local uint64_t orc_work[8];
#define ORC_WORK(i, n) (orc_work[i] += (uint64_t)(n))
void orc_work_get(uint64_t *out8, int reset)
{
    for (int i = 0; i < 8; i++) { out8[i] = orc_work[i]; if (reset) orc_work[i] = 0; }
}
#else
#define ORC_WORK(i, n) ((void)0)
void orc_work_get(uint64_t *out8, int reset)
{
    (void)reset;
    for (int i = 0; i < 8; i++) out8[i] = 0;
}
#endif

int orc_bmp_search(const char *text, size_t textSize, const char *pattern, size_t patternSize)
{
    ORC_WORK(0, 1);
    if (textSize == 0 || patternSize == 0) return -1;
    if (patternSize > textSize) return -1;
    int bmpLast[128];
    for (int i = 0; i < 128; i++) bmpLast[i] = -1;
    for (size_t i = 0; i < patternSize; i++) bmpLast[(unsigned char)pattern[i] & 127] = (int)i;
    size_t tIdx = patternSize - 1, pIdx = patternSize - 1;
    while (tIdx < textSize) {
        if (pattern[pIdx] == text[tIdx]) {
            if (pIdx == 0) return (int)tIdx;
            tIdx--; pIdx--;
        } else {
            int lastOccur = bmpLast[(unsigned char)text[tIdx] & 127];
            int a = (int)pIdx, b = 1 + lastOccur;
            tIdx = tIdx + patternSize - (size_t)(a < b ? a : b);
            pIdx = patternSize - 1;
        }
    }
    return -1;
}

/* PatternMatcher.cpp:111-195 — full (n+1)x(m+1) matrix, transposition term only for i>2 && j>2 */
int orc_levenshtein(const char *source, int n, const char *target, int m)
{
    if (n == 0) return m;
    if (m == 0) return n;
    int W = m + 1;
    int *M = (int *)xmalloc(sizeof(int) * (size_t)(n + 1) * (size_t)W);
    for (int i = 0; i <= n; i++) M[i * W] = i;
    for (int j = 0; j <= m; j++) M[j] = j;
    for (int i = 1; i <= n; i++) {
        char s_i = source[i - 1];
        for (int j = 1; j <= m; j++) {
            char t_j = target[j - 1];
            int cost = (s_i == t_j) ? 0 : 1;
            int above = M[(i - 1) * W + j], left = M[i * W + j - 1], diag = M[(i - 1) * W + j - 1];
            int cell = left + 1 < diag + cost ? left + 1 : diag + cost;
            if (above + 1 < cell) cell = above + 1;
            if (i > 2 && j > 2) {
                int trans = M[(i - 2) * W + j - 2] + 1;
                if (source[i - 2] != t_j) trans++;
                if (s_i != target[j - 2]) trans++;
                if (cell > trans) cell = trans;
            }
            M[i * W + j] = cell;
        }
    }
    int r = M[n * W + m];
    free(M);
    return r;
}

/* PatternMatcher.cpp:197-204: float ratio, "1.0 -" evaluated in double then narrowed */
float orc_similarity(const char *s1, int n, const char *s2, int m)
{
    float max_length = (float)(size_t)(n > m ? n : m);
    if (n < 3 || m < 3) return 0;
    ORC_WORK(5, 1); ORC_WORK(7, (uint64_t)n * (uint64_t)m);
    float edit_distance = (float)orc_levenshtein(s1, n, s2, m);
    return (float)(1.0 - (double)(edit_distance / max_length));
}

/* ------------------------------------------------------------------------- */
/* ReadHolder restatement                                                     */
/* ------------------------------------------------------------------------- */

typedef struct {
    const char *seq;
    int L;
    uint32_t *ss;
    int nss, cap;
    int repeat_len;     /* RH_RepeatLength */
    int overflow;       /* ss capacity exceeded (oracle artefact, reported as error) */
} rh_t;

/* ReadHolder::startStopsAdd, ReadHolder.cpp:263-297 */
static void rh_add(rh_t *h, uint32_t i, uint32_t j)
{
    if (h->nss + 2 > h->cap) { h->overflow = 1; return; }
    h->ss[h->nss++] = i;
    if (j >= (uint32_t)h->L) j = (uint32_t)h->L - 1;
    h->ss[h->nss++] = j;
}

/* std::string::substr(pos, n) length semantics: throws if pos > size, else clamps */
static int substr_len(int L, uint32_t pos, uint32_t n, uint32_t *out_len)
{
    if (pos > (uint32_t)L) return -1;
    uint32_t avail = (uint32_t)L - pos;
    *out_len = n < avail ? n : avail;
    return 0;
}

/* scanRight, libcrispr.cpp:170-263 */
static void scan_right(rh_t *h, const char *pattern, uint32_t pattern_length,
                       uint32_t minSpacerLength, uint32_t scanRange)
{
    uint32_t start_stops_size = (uint32_t)h->nss;
    uint32_t last_repeat_index = h->ss[start_stops_size - 2];
    uint32_t second_last_repeat_index = h->ss[start_stops_size - 4];
    uint32_t repeat_spacing = last_repeat_index - second_last_repeat_index;
    int candidate_repeat_index, position;
    uint32_t begin_search, end_search;
    uint32_t read_length = (uint32_t)h->L;
    int more_to_search = 1;
    while (more_to_search) {
        candidate_repeat_index = (int)(last_repeat_index + repeat_spacing);
        begin_search = (uint32_t)candidate_repeat_index - scanRange;
        end_search = (uint32_t)candidate_repeat_index + pattern_length + scanRange;
        uint32_t scanRightMinBegin = last_repeat_index + pattern_length + minSpacerLength;
        if (begin_search < scanRightMinBegin) begin_search = scanRightMinBegin;
        if (begin_search > read_length - 1) return;
        if (end_search > read_length) end_search = read_length;
        if (begin_search >= end_search) return;
        ORC_WORK(2, 1);
        position = orc_bmp_search(h->seq + begin_search, end_search - begin_search, pattern, pattern_length);
        if (position >= 0) {
            rh_add(h, begin_search + (uint32_t)position, begin_search + (uint32_t)position + pattern_length - 1);
            if (h->overflow) return;
            second_last_repeat_index = last_repeat_index;
            last_repeat_index = begin_search + (uint32_t)position;
            repeat_spacing = last_repeat_index - second_last_repeat_index;
            if (repeat_spacing < (minSpacerLength + pattern_length)) more_to_search = 0;
        } else {
            more_to_search = 0;
        }
    }
}

int orc_scan_right(const char *seq, int L, uint32_t *ss, int *nss, int cap,
                   const char *pat, int plen, uint32_t minSpacer, uint32_t scanRange)
{
    rh_t h = { seq, L, ss, *nss, cap, 0, 0 };
    if (h.nss < 4) return -1;
    scan_right(&h, pat, (uint32_t)plen, minSpacer, scanRange);
    *nss = h.nss;
    return h.overflow ? -2 : 0;
}

/* extendPreRepeat, libcrispr.cpp:520-772 */
static uint32_t extend_pre_repeat(rh_t *h, int searchWindowLength, int minSpacerLength)
{
    uint32_t num_repeats = (uint32_t)h->nss / 2;
    h->repeat_len = searchWindowLength;
    int cut_off = (int)(num_repeats - 1);
    if (2 > cut_off) cut_off = 2;

    uint32_t first_repeat_start_index = h->ss[0];
    uint32_t last_repeat_start_index = h->ss[h->nss - 2];
    uint32_t shortest_repeat_spacing = (uint32_t)((int)h->ss[2] - (int)h->ss[0]);
    uint32_t end_index = (uint32_t)h->nss;
    for (uint32_t i = 4; i < end_index; i += 2) {
        uint32_t curr_repeat_spacing = (uint32_t)((int)h->ss[i] - (int)h->ss[i - 2]);
        if (curr_repeat_spacing < shortest_repeat_spacing) shortest_repeat_spacing = curr_repeat_spacing;
    }
    uint32_t right_extension_length = 0;
    uint32_t max_right_extension_length = shortest_repeat_spacing - (uint32_t)minSpacerLength;
    uint32_t DR_index_end = end_index;
    int cA = 0, cC = 0, cT = 0, cG = 0;
    uint32_t seqlen = (uint32_t)h->L;

    while (max_right_extension_length > 0) {
        if ((last_repeat_start_index + (uint32_t)searchWindowLength + right_extension_length) >= seqlen)
            DR_index_end -= 2;
        ORC_WORK(3, 1); ORC_WORK(4, DR_index_end / 2);
        for (uint32_t k = 0; k < DR_index_end; k += 2) {
            if ((h->ss[k] + (uint32_t)h->repeat_len) >= seqlen) {
                k = DR_index_end;     /* then k += 2 ends the loop */
            } else {
                switch (h->seq[h->ss[k] + (uint32_t)h->repeat_len]) {
                    case 'A': cA++; break;
                    case 'C': cC++; break;
                    case 'G': cG++; break;
                    case 'T': cT++; break;
                }
            }
        }
        if ((cA >= cut_off) || (cC >= cut_off) || (cG >= cut_off) || (cT >= cut_off)) {
            h->repeat_len++;
            max_right_extension_length--;
            right_extension_length++;
            cA = cC = cT = cG = 0;
        } else {
            break;
        }
    }
    cA = cC = cT = cG = 0;

    uint32_t left_extension_length = 0;
    int test_for_negative = (int)(shortest_repeat_spacing - (uint32_t)h->repeat_len);
    uint32_t max_left_extension_length = (test_for_negative >= 0) ? (uint32_t)test_for_negative : 0;
    uint32_t DR_index_start = 0;
    while (left_extension_length < max_left_extension_length) {
        if ((int)first_repeat_start_index - (int)left_extension_length <= 0)
            DR_index_start += 2;
        ORC_WORK(3, 1); ORC_WORK(4, (end_index - DR_index_start) / 2);
        for (uint32_t k = DR_index_start; k < end_index; k += 2) {
            int idx = (int)(h->ss[k] - left_extension_length - 1);
            switch (h->seq[idx]) {
                case 'A': cA++; break;
                case 'C': cC++; break;
                case 'G': cG++; break;
                case 'T': cT++; break;
            }
        }
        if ((cA >= cut_off) || (cC >= cut_off) || (cG >= cut_off) || (cT >= cut_off)) {
            h->repeat_len++;
            left_extension_length++;
            cA = cC = cT = cG = 0;
        } else {
            break;
        }
    }
    for (int r = 0; r + 1 < h->nss; r += 2) {
        if (h->ss[r] < left_extension_length) h->ss[r] = 0;
        else h->ss[r] -= left_extension_length;
        if (h->ss[r + 1] + right_extension_length >= seqlen) h->ss[r + 1] = seqlen - 1;
        else h->ss[r + 1] += right_extension_length;
    }
    return (uint32_t)h->repeat_len;
}

uint32_t orc_extend_pre_repeat(const char *seq, int L, uint32_t *ss, int nss, int window, int minSpacer)
{
    rh_t h = { seq, L, ss, nss, nss, 0, 0 };
    return extend_pre_repeat(&h, window, minSpacer);
}

/* isRepeatLowComplexity, libcrispr.cpp:1031-1069 */
int orc_is_low_complexity(const char *rep, int n)
{
    int c = 0, g = 0, a = 0, t = 0, o = 0;
    int cut_off = (int)(n * 0.75);
    for (int i = 0; i < n; i++) {
        switch (rep[i]) {
            case 'c': case 'C': c++; break;
            case 't': case 'T': t++; break;
            case 'a': case 'A': a++; break;
            case 'g': case 'G': g++; break;
            default: o++; break;
        }
    }
    if (a > cut_off) return 1;
    else if (t > cut_off) return 1;
    else if (g > cut_off) return 1;
    else if (c > cut_off) return 1;
    else if (o > cut_off) return 1;
    return 0;
}

typedef struct { uint32_t start, len; } span_t;

/* ReadHolder::getAllSpacerStrings via getFirstSpacer/getNextSpacer,
 * ReadHolder.cpp:199-239, 812-952.  Restated as the walk it performs. */
static int get_all_spacers(const rh_t *h, span_t *out, int cap, int *n_out)
{
    int n = 0;
    int size = h->nss;
    int L = h->L;
    int next;  /* RH_NextSpacerStart */
    uint32_t len;
    span_t cur;
    if (size < 1) return -1;
    /* getFirstSpacer -> getNextSpacer with RH_NextSpacerStart == 0 */
    if (0 > size - 1) return -1;
    if (h->ss[0] != 0) {
        if (substr_len(L, 0, h->ss[0], &len)) return -1;
        cur.start = 0; cur.len = len;
        next = 1;
    } else {
        int start_cut = (int)h->ss[1] + 1;
        if (2 < size) {
            if (substr_len(L, (uint32_t)start_cut, h->ss[2] - (uint32_t)start_cut, &len)) return -1;
        } else {
            if (substr_len(L, (uint32_t)start_cut, (uint32_t)(L - start_cut), &len)) return -1;
        }
        cur.start = (uint32_t)start_cut; cur.len = len;
        next = 3;
    }
    if (h->ss[0] == 0) { if (n >= cap) return -2; out[n++] = cur; }
    for (;;) {
        if (next > size - 1) break;
        if (next == size - 1) {
            if (h->ss[next] < (uint32_t)(L - 1)) {
                cur.start = h->ss[next] + 1; cur.len = (uint32_t)L - cur.start;
                next += 2;
                if (n >= cap) return -2;
                out[n++] = cur;
                continue;
            }
            break;
        } else {
            int start_cut = (int)h->ss[next] + 1;
            int length = (int)(h->ss[next + 1] - (uint32_t)start_cut);
            if (substr_len(L, (uint32_t)start_cut, (uint32_t)length, &len)) return -1;
            cur.start = (uint32_t)start_cut; cur.len = len;
            next += 2;
            if (n >= cap) return -2;
            out[n++] = cur;
        }
    }
    if (h->ss[size - 1] != (uint32_t)(L - 1)) { if (n > 0) n--; }
    *n_out = n;
    return 0;
}

/* qcFoundRepeats, libcrispr.cpp:869-1029 (+ test* helpers :773-867) */
static int qc_found_repeats(const rh_t *h, int minSpacerLength, int maxSpacerLength)
{
    ORC_WORK(6, 1);
    int num_repeats = h->nss / 2;
    if (num_repeats < 2) return -1;
    /* repeatStringAt(0), ReadHolder.cpp:99 */
    uint32_t rep_len;
    if (substr_len(h->L, h->ss[0], h->ss[1] - h->ss[0] + 1, &rep_len)) return -1;
    const char *repeat = h->seq + h->ss[0];
    if (orc_is_low_complexity(repeat, (int)rep_len)) return 0;

    int single_compare_index = 0;
    int is_short = (2 > (num_repeats - 1));
    if (!is_short) {
        float ave_spacer_to_spacer_len_difference = 0.0f;
        float ave_repeat_to_spacer_len_difference = 0.0f;
        float ave_spacer_to_spacer_difference = 0.0f;
        float ave_repeat_to_spacer_difference = 0.0f;
        int min_spacer_length = 10000000;
        int max_spacer_length = 0;
        int num_compared = 0;
        span_t *sp = (span_t *)xmalloc(sizeof(span_t) * (size_t)(num_repeats + 2));
        int nsp = 0;
        int rc = get_all_spacers(h, sp, num_repeats + 2, &nsp);
        if (rc || nsp < 1) { free(sp); return -1; }
        for (int i = 0; i + 1 < nsp; i++) {
            num_compared++;
            ave_repeat_to_spacer_difference += orc_similarity(repeat, (int)rep_len, h->seq + sp[i].start, (int)sp[i].len);
            float ss_diff = 0;
            ss_diff += orc_similarity(h->seq + sp[i].start, (int)sp[i].len, h->seq + sp[i + 1].start, (int)sp[i + 1].len);
            ave_spacer_to_spacer_difference += ss_diff;
            ave_spacer_to_spacer_len_difference += ((float)sp[i].len - (float)sp[i + 1].len);
            ave_repeat_to_spacer_len_difference += ((float)rep_len - (float)sp[i].len);
        }
        for (int i = 0; i < nsp; i++) {
            if ((int)sp[i].len < min_spacer_length) min_spacer_length = (int)sp[i].len;
            if ((int)sp[i].len > max_spacer_length) max_spacer_length = (int)sp[i].len;
        }
        free(sp);
        if (num_compared == 0) {
            is_short = 1;
            single_compare_index = 1;
        } else {
            ave_spacer_to_spacer_difference /= (float)num_compared;
            ave_repeat_to_spacer_difference /= (float)num_compared;
            ave_spacer_to_spacer_len_difference /= (float)num_compared;
            ave_spacer_to_spacer_len_difference = fabsf(ave_spacer_to_spacer_len_difference);
            ave_repeat_to_spacer_len_difference /= (float)num_compared;
            ave_repeat_to_spacer_len_difference = fabsf(ave_repeat_to_spacer_len_difference);
            /* testSpacerLength :773-800 */
            if (min_spacer_length < minSpacerLength) return 0;
            if (max_spacer_length > maxSpacerLength) return 0;
            /* testSpacerSpacerSimilarity / testSpacerRepeatSimilarity :802-834 (float vs double 0.82) */
            if ((double)ave_spacer_to_spacer_difference > 0.82) return 0;
            if ((double)ave_repeat_to_spacer_difference > 0.82) return 0;
            /* the two length tests take an int parameter: float -> int truncation (:836,853) */
            if ((int)ave_spacer_to_spacer_len_difference > 12) return 0;
            if ((int)ave_repeat_to_spacer_len_difference > 30) return 0;
        }
    }
    if (is_short) {
        /* spacerStringAt(i), ReadHolder.cpp:102-147: one base short */
        int i = single_compare_index;
        if (i + 2 >= h->nss) return -1;          /* RH_StartStops.at() would throw */
        uint32_t s = h->ss[i + 1] + 1;
        uint32_t e = h->ss[i + 2] - 1;
        uint32_t sp_len;
        if (substr_len(h->L, s, e - s, &sp_len)) return -1;
        const char *spacer = h->seq + s;
        if ((int)sp_len < minSpacerLength) return 0;
        if ((int)sp_len > maxSpacerLength) return 0;
        float similarity = orc_similarity(repeat, (int)rep_len, spacer, (int)sp_len);
        if ((double)similarity > 0.82) return 0;
        int d = (int)sp_len - (int)rep_len;
        if (d < 0) d = -d;
        if (d > 30) return 0;
    }
    return 1;
}

int orc_qc_found_repeats(const char *seq, int L, const uint32_t *ss, int nss, int minSpacer, int maxSpacer)
{
    rh_t h = { seq, L, (uint32_t *)ss, nss, nss, 0, 0 };
    return qc_found_repeats(&h, minSpacer, maxSpacer);
}

/* searchCore, libcrispr.cpp:265-395.  lattice_only: stop at the first seed hit and
 * report it (used to state the device filter's contract). */
static uint64_t orc_stat_rejected = 0, orc_stat_class_switches = 0;     /* (not thread safe: a single-threaded diagnostic) */
static int search_core(rh_t *h, const orc_params *o, int lattice_only)
{
    const char *read = h->seq;
    uint32_t seq_length = (uint32_t)h->L;
    uint32_t skips = o->lowDRsize - (2 * o->searchWindowLength - 1);
    if (skips < 1) skips = 1;
    int searchEnd = (int)(seq_length - o->lowDRsize - o->lowSpacerSize - o->searchWindowLength - 1);
    if (searchEnd < 0) return 0;
    /* lowDR < 2w-1 wraps `skips` to ~4e9: after a failed candidate the reference's j then moves
     * backwards and the loop need not terminate.  Ill-defined there; reported as an error here. */
    if (o->lowDRsize < 2 * o->searchWindowLength - 1) return -3;
    h->nss = 0;
    for (uint32_t j = 0; j <= (uint32_t)searchEnd; j = j + skips) {
        uint32_t beginSearch = j + o->lowDRsize + o->lowSpacerSize;
        uint32_t endSearch = j + o->highDRsize + o->highSpacerSize + o->searchWindowLength;
        if (endSearch >= seq_length) endSearch = seq_length - 1;
        if (endSearch < beginSearch) endSearch = beginSearch;
        uint32_t tlen, plen;
        if (substr_len((int)seq_length, beginSearch, endSearch - beginSearch, &tlen)) return -1;
        if (substr_len((int)seq_length, j, o->searchWindowLength, &plen)) return -1;
        const char *text = read + beginSearch;
        const char *pattern = read + j;
        int pattern_in_text_index = orc_bmp_search(text, tlen, pattern, plen);
        if (pattern_in_text_index >= 0) {
            if (lattice_only) return 1;
            ORC_WORK(1, 1);
            rh_add(h, j, j + o->searchWindowLength - 1);
            uint32_t found = beginSearch + (uint32_t)pattern_in_text_index;
            rh_add(h, found, found + o->searchWindowLength - 1);
            if (h->overflow) return -2;
            scan_right(h, pattern, plen, o->lowSpacerSize, 24);
            if (h->overflow) return -2;
        }
        if ((uint32_t)(h->nss / 2) >= o->minNumRepeats) {
            uint32_t actual_repeat_length = extend_pre_repeat(h, (int)o->searchWindowLength, (int)o->lowSpacerSize);
            if ((actual_repeat_length >= o->lowDRsize) && (actual_repeat_length <= o->highDRsize)) {
                int qc = qc_found_repeats(h, (int)o->lowSpacerSize, (int)o->highSpacerSize);
                if (qc < 0) return -1;
                if (qc) return 1;
            }
            {   /* statistics for tools/class_switches.py only: rejected candidates, and how many of them moved j to another residue class */
                uint32_t nj = h->ss[h->nss - 1] - 1;
                orc_stat_rejected++;
                if (((nj + skips) % skips) != (j % skips)) orc_stat_class_switches++;
            }
            j = h->ss[h->nss - 1] - 1;
        }
        h->nss = 0;
    }
    return 0;
}

void orc_stats_get(uint64_t *rejected, uint64_t *class_switches, int reset)
{
    *rejected = orc_stat_rejected; *class_switches = orc_stat_class_switches;
    if (reset) { orc_stat_rejected = 0; orc_stat_class_switches = 0; }
}

int orc_search_core(const char *seq, int L, const orc_params *p, uint32_t *ss, int *nss, int cap, uint32_t *repeat_len)
{
    rh_t h = { seq, L, ss, 0, cap, 0, 0 };
    int r = search_core(&h, p, 0);
    *nss = h.nss;
    *repeat_len = (uint32_t)h.repeat_len;
    return r;
}

int orc_has_lattice_hit(const char *seq, int L, const orc_params *p)
{
    uint32_t ss[8];
    rh_t h = { seq, L, ss, 0, 8, 0, 0 };
    return search_core(&h, p, 1);
}

/* ReadHolder::DRLowLexi, ReadHolder.cpp:513-591; reverseComplementSeq :593-609;
 * reverseStartStops :321-380 */
int orc_dr_low_lexi(char *seq, int L, uint32_t *ss, int nss, char *dr_out, int *was_low_lexi)
{
    int num_repeats = nss / 2;
    int pick;
    if (num_repeats == 1) pick = 0;
    else if (num_repeats == 2) {
        if (ss[0] == 0) pick = 2;
        else if (ss[nss - 1] == (uint32_t)L) pick = 0;     /* dead branch, kept (:541) */
        else {
            int lenA = (int)(ss[1] - ss[0]);
            int lenB = (int)(ss[3] - ss[2]);
            pick = (lenA > lenB) ? 0 : 2;
        }
    } else pick = 2;
    uint32_t dlen;
    if (substr_len(L, ss[pick], ss[pick + 1] - ss[pick] + 1, &dlen)) return -1;
    const char *tmp_dr = seq + ss[pick];
    char *rev = (char *)xmalloc(dlen + 1);
    orc_revcomp(tmp_dr, dlen, rev);
    if (str_less(tmp_dr, dlen, rev, dlen)) {
        memcpy(dr_out, tmp_dr, dlen);
        *was_low_lexi = 1;
    } else {
        memcpy(dr_out, rev, dlen);
        char *rs = (char *)xmalloc((size_t)L + 1);
        orc_revcomp(seq, (size_t)L, rs);
        memcpy(seq, rs, (size_t)L);
        free(rs);
        /* reverseStartStops: new[k] = (L-1-back) + (back - ss[nss-1-k]) */
        uint32_t *tmp = (uint32_t *)xmalloc(sizeof(uint32_t) * (size_t)nss);
        int true_start_offset = L - (int)ss[nss - 1] - 1;
        uint32_t prev_pos_fixed = (uint32_t)true_start_offset;
        uint32_t prev_pos_orig = ss[nss - 1];
        for (int k = nss - 1, w = 0; k >= 0; k--, w++) {
            uint32_t gap = prev_pos_orig - ss[k];
            prev_pos_fixed += gap;
            tmp[w] = prev_pos_fixed;
            prev_pos_orig = ss[k];
        }
        memcpy(ss, tmp, sizeof(uint32_t) * (size_t)nss);
        free(tmp);
        *was_low_lexi = 0;
    }
    free(rev);
    return (int)dlen;
}

/* ------------------------------------------------------------------------- */
/* string -> int hash map (StringCheck / readsFound / k2GIDMap lookups)       */
/* ------------------------------------------------------------------------- */

typedef struct { char *key; uint32_t klen; int64_t val; } sm_ent;
typedef struct { sm_ent *e; size_t cap, n; } smap;

static uint64_t fnv1a(const char *s, size_t n)
{
    uint64_t h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; i++) { h ^= (unsigned char)s[i]; h *= 1099511628211ULL; }
    return h;
}
static void sm_init(smap *m, size_t cap) { m->cap = 16; while (m->cap < cap) m->cap <<= 1; m->e = (sm_ent *)xcalloc(m->cap, sizeof(sm_ent)); m->n = 0; }
static void sm_free(smap *m) { for (size_t i = 0; i < m->cap; i++) free(m->e[i].key); free(m->e); m->e = NULL; }
static sm_ent *sm_find(const smap *m, const char *k, size_t n)
{
    size_t i = fnv1a(k, n) & (m->cap - 1);
    while (m->e[i].key) {
        if (m->e[i].klen == n && memcmp(m->e[i].key, k, n) == 0) return &m->e[i];
        i = (i + 1) & (m->cap - 1);
    }
    return NULL;
}
static void sm_grow(smap *m);
static sm_ent *sm_put(smap *m, const char *k, size_t n, int64_t v)
{
    if ((m->n + 1) * 2 > m->cap) sm_grow(m);
    size_t i = fnv1a(k, n) & (m->cap - 1);
    while (m->e[i].key) {
        if (m->e[i].klen == n && memcmp(m->e[i].key, k, n) == 0) { m->e[i].val = v; return &m->e[i]; }
        i = (i + 1) & (m->cap - 1);
    }
    m->e[i].key = (char *)xmalloc(n + 1); memcpy(m->e[i].key, k, n); m->e[i].key[n] = 0;
    m->e[i].klen = (uint32_t)n; m->e[i].val = v; m->n++;
    return &m->e[i];
}
static void sm_grow(smap *m)
{
    smap b; sm_init(&b, m->cap * 2);
    for (size_t i = 0; i < m->cap; i++) if (m->e[i].key) {
        size_t j = fnv1a(m->e[i].key, m->e[i].klen) & (b.cap - 1);
        while (b.e[j].key) j = (j + 1) & (b.cap - 1);
        b.e[j] = m->e[i]; b.n++;
    }
    free(m->e); *m = b;
}

/* growable buffers */
typedef struct { char *p; size_t n, cap; } cbuf;
static void cb_push(cbuf *b, const void *d, size_t n)
{
    if (b->n + n > b->cap) { b->cap = (b->n + n) * 2 + 64; b->p = (char *)xrealloc(b->p, b->cap); }
    memcpy(b->p + b->n, d, n); b->n += n;
}
#define VEC(T) struct { T *p; size_t n, cap; }
#define VPUSH(v, x) do { if ((v).n == (v).cap) { (v).cap = (v).cap ? (v).cap * 2 : 64; \
        (v).p = xrealloc((v).p, (v).cap * sizeof(*(v).p)); } (v).p[(v).n++] = (x); } while (0)

/* ------------------------------------------------------------------------- */
/* multi-pattern first match: byte-wise Aho-Corasick                          */
/* ------------------------------------------------------------------------- */

struct orc_ac {
    int sym[256];        /* byte -> symbol (0 = not in any pattern: acism.c:35-40) */
    int nsym;            /* symbols 1..nsym */
    int nstates;
    int32_t *go;         /* [nstates][nsym+1] full goto (failure-resolved) */
    uint32_t *out_len;   /* longest pattern ending at this state (via dict-suffix chain), 0 = none */
};

orc_ac *orc_ac_create(const char *const *pats, const uint32_t *lens, uint32_t n)
{
    orc_ac *ac = (orc_ac *)xcalloc(1, sizeof(orc_ac));
    size_t total = 1;
    for (uint32_t i = 0; i < n; i++) {
        total += lens[i];
        for (uint32_t k = 0; k < lens[i]; k++) {
            unsigned char c = (unsigned char)pats[i][k];
            if (!ac->sym[c]) ac->sym[c] = ++ac->nsym;
        }
    }
    int S = ac->nsym + 1;
    int32_t *go = (int32_t *)xmalloc(sizeof(int32_t) * total * (size_t)S);
    uint32_t *term = (uint32_t *)xcalloc(total, sizeof(uint32_t));
    for (int k = 0; k < S; k++) go[k] = -1;
    int ns = 1;
    for (uint32_t i = 0; i < n; i++) {
        int s = 0;
        for (uint32_t k = 0; k < lens[i]; k++) {
            int c = ac->sym[(unsigned char)pats[i][k]];
            if (go[s * S + c] < 0) {
                for (int q = 0; q < S; q++) go[ns * S + q] = -1;
                go[s * S + c] = ns++;
            }
            s = go[s * S + c];
        }
        if (lens[i] > term[s]) term[s] = lens[i];
    }
    /* BFS: failure links, full goto, propagated output = longest pattern that is a suffix */
    int32_t *fail = (int32_t *)xcalloc((size_t)ns, sizeof(int32_t));
    int32_t *queue = (int32_t *)xmalloc(sizeof(int32_t) * (size_t)ns);
    int qh = 0, qt = 0;
    for (int c = 1; c < S; c++) {
        int t = go[c];
        if (t < 0) go[c] = 0; else { fail[t] = 0; queue[qt++] = t; }
    }
    go[0] = 0;
    while (qh < qt) {
        int s = queue[qh++];
        /* the state's own pattern is deeper (longer) than anything on its fail chain */
        if (!term[s]) term[s] = term[fail[s]];
        go[s * S + 0] = 0;
        for (int c = 1; c < S; c++) {
            int t = go[s * S + c];
            if (t < 0) go[s * S + c] = go[fail[s] * S + c];
            else { fail[t] = go[fail[s] * S + c]; queue[qt++] = t; }
        }
    }
    free(fail); free(queue);
    ac->nstates = ns; ac->go = go; ac->out_len = term;
    return ac;
}

void orc_ac_destroy(orc_ac *ac)
{
    if (!ac) return;
    free(ac->go); free(ac->out_len); free(ac);
}

int orc_ac_first_match(const orc_ac *ac, const char *text, size_t tlen, uint32_t *end_excl, uint32_t *len)
{
    int S = ac->nsym + 1;
    int s = 0;
    for (size_t i = 0; i < tlen; i++) {
        int c = ac->sym[(unsigned char)text[i]];
        s = ac->go[s * S + c];
        if (ac->out_len[s]) { *end_excl = (uint32_t)(i + 1); *len = ac->out_len[s]; return 1; }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* pipeline                                                                   */
/* ------------------------------------------------------------------------- */

struct orc_result {
    orc_view v;
    VEC(uint64_t) rec_read;
    VEC(uint8_t) rec_lowlexi;
    VEC(uint32_t) rec_token;
    VEC(uint32_t) rec_replen;
    VEC(uint32_t) rec_nss;
    VEC(uint64_t) rec_ss_off;
    VEC(uint32_t) ss_pool;
    cbuf tok_chars; VEC(uint64_t) tok_off;
    VEC(uint32_t) grp_tokens; VEC(uint64_t) grp_off;
    cbuf pat_chars; VEC(uint64_t) pat_off; VEC(uint32_t) pat_group;
    smap s2t;           /* StringCheck::mS2T_map */
    uint32_t next_token; /* StringCheck::mNextFreeToken (ctor sets 1, StringCheck.h:55-56) */
};

/* addReadHolder, libcrispr.cpp:1119-1162 (+ StringCheck.cpp:46-81) */
static int add_read_holder(orc_result *r, uint64_t read_idx, const char *seq, int L,
                           const uint32_t *ss, int nss, uint32_t replen, char *scratch_seq,
                           uint32_t *scratch_ss, char *scratch_dr)
{
    memcpy(scratch_seq, seq, (size_t)L);
    memcpy(scratch_ss, ss, sizeof(uint32_t) * (size_t)nss);
    int low = 0;
    int dlen = orc_dr_low_lexi(scratch_seq, L, scratch_ss, nss, scratch_dr, &low);
    if (dlen < 0) return -1;
    uint32_t st;
    sm_ent *e = sm_find(&r->s2t, scratch_dr, (size_t)dlen);
    if (!e) {
        st = ++r->next_token;
        sm_put(&r->s2t, scratch_dr, (size_t)dlen, st);
        cb_push(&r->tok_chars, scratch_dr, (size_t)dlen);
        VPUSH(r->tok_off, (uint64_t)r->tok_chars.n);
    } else st = (uint32_t)e->val;
    VPUSH(r->rec_read, read_idx);
    VPUSH(r->rec_lowlexi, (uint8_t)low);
    VPUSH(r->rec_token, st);
    VPUSH(r->rec_replen, replen);
    VPUSH(r->rec_nss, (uint32_t)nss);
    VPUSH(r->rec_ss_off, (uint64_t)r->ss_pool.n);
    for (int k = 0; k < nss; k++) VPUSH(r->ss_pool, scratch_ss[k]);
    return 0;
}

static const char *tok_str(const orc_result *r, uint32_t tok, uint32_t *len)
{
    uint64_t a = r->tok_off.p[tok - 2], b = r->tok_off.p[tok - 1];
    *len = (uint32_t)(b - a);
    return r->tok_chars.p + a;
}

#define KMER 11   /* CRASS_DEF_KMER_SIZE, crassDefines.h:66 */

/* WorkHorse::clusterDRReads, WorkHorse.cpp:1404-1637.
 * k2gid: laurenized 11-mer -> GID.  Returns the GID the token was put into. */
static int cluster_dr(orc_result *r, uint32_t tok, int *nextFreeGID, smap *k2gid,
                      int min_clust_membership_count, int *err)
{
    uint32_t str_len;
    const char *DR = tok_str(r, tok, &str_len);
    int off = (int)str_len - KMER;
    int num_mers = off + 1;
    if (num_mers <= 0) { *err = 1; return 0; }   /* new char*[<=0] / bad_array_new_length in the reference */
    /* the three-phase cutter (:1490-1532) yields kmers[i] = DR[i .. i+11) */
    VEC(int) gc_gid = {0}; VEC(int) gc_cnt = {0};           /* std::map<int,int> group_count */
    char (*homeless)[KMER] = (char (*)[KMER])xmalloc((size_t)num_mers * KMER);
    int n_homeless = 0;
    int group = 0;
    char km[KMER], rc[KMER];
    for (int i = 0; i < num_mers; ++i) {
        memcpy(km, DR + i, KMER);
        orc_revcomp(km, KMER, rc);
        const char *lau = str_less(km, KMER, rc, KMER) ? km : rc;     /* laurenize, SeqUtils.cpp:89-97 */
        sm_ent *e = sm_find(k2gid, lau, KMER);
        if (!e) {
            memcpy(homeless[n_homeless++], lau, KMER);
        } else if (0 == group) {
            int g = (int)e->val;
            size_t q;
            for (q = 0; q < gc_gid.n; q++) if (gc_gid.p[q] == g) break;
            if (q == gc_gid.n) { VPUSH(gc_gid, g); VPUSH(gc_cnt, 1); }
            else {
                gc_cnt.p[q]++;
                if (min_clust_membership_count <= gc_cnt.p[q]) group = g;
            }
        }
    }
    if (0 == group) group = (*nextFreeGID)++;
    for (int i = 0; i < n_homeless; i++) sm_put(k2gid, homeless[i], KMER, group);
    free(homeless); free(gc_gid.p); free(gc_cnt.p);
    return group;
}

/* includeSubstring, WorkHorse.cpp:78-86 */
static int mem_find(const char *hay, size_t hn, const char *nee, size_t nn)
{
    if (nn > hn) return 0;
    for (size_t i = 0; i + nn <= hn; i++) if (memcmp(hay + i, nee, nn) == 0) return 1;
    return 0;
}

/* createNonRedundantSet + removeRedundantRepeats, WorkHorse.cpp:612-709 */
static void create_non_redundant_set(orc_result *r, const orc_params *p)
{
    smap k2gid; sm_init(&k2gid, 1024);
    int nextFreeGID = 1;
    uint32_t ntok = r->next_token - 1;
    int *tok_gid = (int *)xcalloc(ntok + 2, sizeof(int));
    int err = 0;
    for (uint32_t t = 2; t < 2 + ntok; t++)            /* std::map<StringToken,...> iteration = ascending token */
        tok_gid[t - 2] = cluster_dr(r, t, &nextFreeGID, &k2gid, p->kmer_clust_size, &err);
    if (err) r->v.error = 2;
    int ngroups = nextFreeGID - 1;
    VPUSH(r->grp_off, 0);
    for (int g = 1; g <= ngroups; g++) {
        for (uint32_t t = 2; t < 2 + ntok; t++) if (tok_gid[t - 2] == g) VPUSH(r->grp_tokens, t);
        VPUSH(r->grp_off, (uint64_t)r->grp_tokens.n);
    }
    VPUSH(r->pat_off, 0);
    for (int g = 1; g <= ngroups; g++) {
        uint64_t a = r->grp_off.p[g - 1], b = r->grp_off.p[g];
        size_t n = (size_t)(b - a);
        uint32_t *toks = (uint32_t *)xmalloc(sizeof(uint32_t) * n);
        memcpy(toks, r->grp_tokens.p + a, sizeof(uint32_t) * n);
        /* canonical stable sort by length ascending (reference: std::sort, unstable; only the
         * resulting set is order-independent — see DESIGN.md) */
        for (size_t i = 1; i < n; i++) {
            uint32_t x = toks[i]; uint32_t lx; tok_str(r, x, &lx);
            size_t j = i;
            while (j > 0) { uint32_t ly; tok_str(r, toks[j - 1], &ly); if (ly <= lx) break; toks[j] = toks[j - 1]; j--; }
            toks[j] = x;
        }
        char *blank = (char *)xcalloc(n, 1);
        char rcbuf[4096];
        for (size_t i = 0; i < n; i++) {
            if (blank[i]) continue;
            uint32_t la; const char *sa = tok_str(r, toks[i], &la);
            char *rc = la < sizeof(rcbuf) ? rcbuf : (char *)xmalloc(la);
            orc_revcomp(sa, la, rc);
            for (size_t j = i + 1; j < n; j++) {
                if (blank[j]) continue;
                uint32_t lb; const char *sb = tok_str(r, toks[j], &lb);
                if (mem_find(sb, lb, sa, la) || mem_find(sb, lb, rc, la)) blank[j] = 1;
            }
            if (rc != rcbuf) free(rc);
        }
        size_t first_pat = r->pat_off.n - 1;
        for (size_t i = 0; i < n; i++) if (!blank[i]) {
            uint32_t la; const char *sa = tok_str(r, toks[i], &la);
            cb_push(&r->pat_chars, sa, la);
            VPUSH(r->pat_off, (uint64_t)r->pat_chars.n);
            VPUSH(r->pat_group, (uint32_t)g);
        }
        size_t last_pat = r->pat_off.n - 1;
        for (size_t q = first_pat; q < last_pat; q++) {
            uint64_t pa = r->pat_off.p[q], pb = r->pat_off.p[q + 1];
            size_t l = (size_t)(pb - pa);
            char *rc = (char *)xmalloc(l + 1);
            orc_revcomp(r->pat_chars.p + pa, l, rc);
            cb_push(&r->pat_chars, rc, l);
            free(rc);
            VPUSH(r->pat_off, (uint64_t)r->pat_chars.n);
            VPUSH(r->pat_group, (uint32_t)g);
        }
        free(blank); free(toks);
    }
    free(tok_gid);
    sm_free(&k2gid);
    r->v.n_groups = (uint32_t)ngroups;
    r->v.n_patterns = (uint32_t)(r->pat_off.n - 1);
}

static double now_s(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static orc_result *pipeline(const char *seqs, const uint64_t *seq_off, uint64_t n_reads,
                            const char *hdrs, const uint64_t *hdr_off,
                            const orc_params *p, int do_pass2, double *tm)
{
    orc_result *r = (orc_result *)xcalloc(1, sizeof(orc_result));
    r->next_token = 1;
    sm_init(&r->s2t, 1024);
    VPUSH(r->tok_off, 0);
    smap found; sm_init(&found, 1024);                 /* readsFound (header -> true) */
    uint8_t *found_idx = hdrs ? NULL : (uint8_t *)xcalloc(n_reads, 1);
    uint32_t maxL = 0;
    for (uint64_t i = 0; i < n_reads; i++) { uint64_t l = seq_off[i + 1] - seq_off[i]; if (l > maxL) maxL = (uint32_t)l; }
    int cap = (int)maxL + 8;
    uint32_t *ss = (uint32_t *)xmalloc(sizeof(uint32_t) * (size_t)cap);
    uint32_t *ss2 = (uint32_t *)xmalloc(sizeof(uint32_t) * (size_t)cap);
    char *sseq = (char *)xmalloc((size_t)maxL + 1);
    char *sdr = (char *)xmalloc((size_t)maxL + 1);
    double t0 = now_s();
    /* pass 1: searchFile, libcrispr.cpp:68-166 */
    for (uint64_t i = 0; i < n_reads; i++) {
        const char *seq = seqs + seq_off[i];
        int L = (int)(seq_off[i + 1] - seq_off[i]);
        rh_t h = { seq, L, ss, 0, cap, 0, 0 };
        int f = search_core(&h, p, 0);
        if (f < 0) { r->v.error = 1; break; }
        if (f) {
            if (add_read_holder(r, i, seq, L, ss, h.nss, (uint32_t)h.repeat_len, sseq, ss2, sdr)) { r->v.error = 1; break; }
            if (hdrs) sm_put(&found, hdrs + hdr_off[i], (size_t)(hdr_off[i + 1] - hdr_off[i]), 1);
            else found_idx[i] = 1;
        }
    }
    r->v.n_pass1 = r->rec_read.n;
    r->v.max_read_len = maxL;
    double t1 = now_s();
    /* merge: createNonRedundantSet, WorkHorse.cpp:648-709 */
    create_non_redundant_set(r, p);
    double t2 = now_s();
    /* pass 2: findSingletons + on_match, libcrispr.cpp:399-518 (only if the set is non-empty, WorkHorse.cpp:373) */
    if (do_pass2 && r->v.n_patterns > 0 && !r->v.error) {
        uint32_t np = r->v.n_patterns;
        const char **pp = (const char **)xmalloc(sizeof(char *) * np);
        uint32_t *pl = (uint32_t *)xmalloc(sizeof(uint32_t) * np);
        for (uint32_t q = 0; q < np; q++) { pp[q] = r->pat_chars.p + r->pat_off.p[q]; pl[q] = (uint32_t)(r->pat_off.p[q + 1] - r->pat_off.p[q]); }
        orc_ac *ac = orc_ac_create(pp, pl, np);
        for (uint64_t i = 0; i < n_reads; i++) {
            const char *seq = seqs + seq_off[i];
            int L = (int)(seq_off[i + 1] - seq_off[i]);
            uint32_t e, len;
            if (!orc_ac_first_match(ac, seq, (size_t)L, &e, &len)) continue;
            int isfound = hdrs ? (sm_find(&found, hdrs + hdr_off[i], (size_t)(hdr_off[i + 1] - hdr_off[i])) != NULL)
                               : found_idx[i];
            if (isfound) continue;
            uint32_t DR_end = e - 1;
            if (DR_end >= (uint32_t)L) DR_end = (uint32_t)L - 1;
            rh_t h = { seq, L, ss, 0, cap, 0, 0 };
            rh_add(&h, DR_end - (len - 1), DR_end);
            if (add_read_holder(r, i, seq, L, ss, h.nss, 0, sseq, ss2, sdr)) { r->v.error = 1; break; }
        }
        orc_ac_destroy(ac);
        free(pp); free(pl);
    }
    double t3 = now_s();
    if (tm) { tm[0] = t1 - t0; tm[1] = t2 - t1; tm[2] = t3 - t2; }
    r->v.n_pass2 = r->rec_read.n - r->v.n_pass1;
    r->v.n_tokens = r->next_token - 1;
    free(ss); free(ss2); free(sseq); free(sdr); free(found_idx);
    sm_free(&found);
    r->v.rec_read = r->rec_read.p; r->v.rec_lowlexi = r->rec_lowlexi.p; r->v.rec_token = r->rec_token.p;
    r->v.rec_replen = r->rec_replen.p; r->v.rec_nss = r->rec_nss.p; r->v.rec_ss_off = r->rec_ss_off.p;
    r->v.ss_pool = r->ss_pool.p; r->v.tok_chars = r->tok_chars.p; r->v.tok_off = r->tok_off.p;
    r->v.grp_tokens = r->grp_tokens.p; r->v.grp_off = r->grp_off.p;
    r->v.pat_chars = r->pat_chars.p; r->v.pat_off = r->pat_off.p; r->v.pat_group = r->pat_group.p;
    return r;
}

orc_result *orc_pipeline_run(const char *seqs, const uint64_t *seq_off, uint64_t n_reads,
                             const char *hdrs, const uint64_t *hdr_off,
                             const orc_params *p, int do_pass2)
{
    return pipeline(seqs, seq_off, n_reads, hdrs, hdr_off, p, do_pass2, NULL);
}

void orc_result_view(const orc_result *r, orc_view *v) { *v = r->v; }

void orc_result_free(orc_result *r)
{
    if (!r) return;
    free(r->rec_read.p); free(r->rec_lowlexi.p); free(r->rec_token.p); free(r->rec_replen.p);
    free(r->rec_nss.p); free(r->rec_ss_off.p); free(r->ss_pool.p);
    free(r->tok_chars.p); free(r->tok_off.p); free(r->grp_tokens.p); free(r->grp_off.p);
    free(r->pat_chars.p); free(r->pat_off.p); free(r->pat_group.p);
    sm_free(&r->s2t);
    free(r);
}

int orc_pipeline_time(const char *seqs, const uint64_t *seq_off, uint64_t n_reads,
                      const orc_params *p, double *t_pass1, double *t_merge, double *t_pass2,
                      uint64_t *n_pass1, uint64_t *n_pass2, uint32_t *n_patterns)
{
    double tm[3];
    orc_result *r = pipeline(seqs, seq_off, n_reads, NULL, NULL, p, 1, tm);
    *t_pass1 = tm[0]; *t_merge = tm[1]; *t_pass2 = tm[2];
    *n_pass1 = r->v.n_pass1; *n_pass2 = r->v.n_pass2; *n_patterns = r->v.n_patterns;
    int err = r->v.error;
    orc_result_free(r);
    return err;
}

/* the timed run that also hands its result back (bench.py compares the HIP path's records with it: the parity check costs no
 * second oracle run); the caller frees it with orc_result_free */
orc_result *orc_pipeline_time_keep(const char *seqs, const uint64_t *seq_off, uint64_t n_reads,
                                   const orc_params *p, double *t_pass1, double *t_merge, double *t_pass2)
{
    double tm[3];
    orc_result *r = pipeline(seqs, seq_off, n_reads, NULL, NULL, p, 1, tm);
    *t_pass1 = tm[0]; *t_merge = tm[1]; *t_pass2 = tm[2];
    return r;
}

/* ---- CPU-baseline calibration (bench.py cpu_baseline.calibration, tools/calibrate_cpu.py) ----
 * The two hot loops of the reference on plain inputs, restated here with the oracle's leaf
 * functions; oracle/ref_shim.cpp holds the same two loops over the COMPILED reference leaves
 * (PatternMatcher::bmpSearch, acism_scan).  The ratio of the two timings says how the oracle's
 * speed relates to the reference's on its two dominant functions (40 % + 34 %).
 * Window sequence: searchCore's seed loop without a hit (libcrispr.cpp:295-339). */
static double calib_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double orc_calib_bmp(const char *seqs, uint64_t n_reads, int L, const orc_params *p, uint64_t *checksum)
{
    const int w = (int)p->searchWindowLength;
    unsigned skips = p->lowDRsize - (2 * p->searchWindowLength - 1);
    if (skips < 1) skips = 1;
    const int search_end = L - (int)p->lowDRsize - (int)p->lowSpacerSize - w - 1;
    uint64_t sum = 0;
    const double t0 = calib_now();
    for (uint64_t r = 0; r < n_reads; r++) {
        const char *seq = seqs + r * (uint64_t)L;
        for (int j = 0; j <= search_end; j += (int)skips) {
            int begin = j + (int)p->lowDRsize + (int)p->lowSpacerSize;
            int end = j + (int)p->highDRsize + (int)p->highSpacerSize + w;
            if (begin >= L) begin = L - 1;
            if (end >= L) end = L - 1;
            if (end - begin < w) break;
            sum += (uint64_t)(int64_t)orc_bmp_search(seq + begin, (size_t)(end - begin), seq + j, (size_t)w);
        }
    }
    const double t1 = calib_now();
    *checksum = sum;
    return t1 - t0;
}

double orc_calib_ac(const orc_ac *ac, const char *seqs, uint64_t n_reads, int L, uint64_t *checksum)
{
    uint64_t sum = 0;
    const double t0 = calib_now();
    for (uint64_t r = 0; r < n_reads; r++) {
        uint32_t e = 0, l = 0;
        if (orc_ac_first_match(ac, seqs + r * (uint64_t)L, (size_t)L, &e, &l)) sum += ((uint64_t)e << 8) + l;
    }
    const double t1 = calib_now();
    *checksum = sum;
    return t1 - t0;
}
