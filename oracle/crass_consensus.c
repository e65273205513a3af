/*
 * crass_consensus.c — CPU restatement (plain C) of the stage right behind the search hot path
 * (SURVEY §8f row f-1): true-DR consensus, group splitting and start/stop repair.
 *
 * TEST INFRASTRUCTURE ONLY (see crass_oracle.h).  Restates, with the reference's control flow:
 *   WorkHorse::findConsensusDRs            src/crass/WorkHorse.cpp:578-611
 *   WorkHorse::parseGroupedDRs             :1135-1379
 *   WorkHorse::findMasterDR                :711-748
 *   WorkHorse::populateCoverageArray       :750-798
 *   WorkHorse::calculateDRConsensus        :801-938
 *   WorkHorse::splitGroupedDR              :940-1132
 *   WorkHorse::combineGroupsWithIdenticalDRs :416-452
 *   Aligner (setMasterDR, alignSlave, getOffsetAgainstMaster, placeReadsInCoverageArray, extendSlaveDR,
 *            calculateDRZone, generateConsensus)      src/crass/Aligner.cpp:73-468, Aligner.h:48,112-136
 *   ksw_qinit / ksw_i16 / ksw_align        src/crass/ksw.c:57-101,228-360  (the SSE2 striped layout is simulated
 *                                          lane by lane: E(i+1,j) is taken BEFORE the lazy-F pass and ties of the
 *                                          query end are broken in vector-memory order, so results depend on it)
 *   ReadHolder::updateStartStops           src/crass/ReadHolder.cpp:382-511
 *   ReadHolder::reverseComplementSeq / reverseStartStops   :593-609, :321-380
 *   smithWaterman / findMax                src/crass/SmithWaterman.cpp:68-129,151-308
 *   drHasHighlyAbundantKmers               src/crass/libcrispr.cpp:1077-1117
 *
 * Pinning: ksw_align against the COMPILED ksw.c (oracle/_ref; tests/test_oracle_consensus.py); Levenshtein and
 * reverseComplement as in crass_oracle.c; everything above them by the known answers SURVEY §8c recorded from the
 * compiled reference (true DR strings, group ids and per-group read counts on the reference's regression inputs).
 * Aligner.cpp / SmithWaterman.cpp / ReadHolder.cpp / WorkHorse.cpp themselves include the autoconf config.h and are
 * not buildable here; no record-level vectors of theirs exist: "parity pinned at group level".
 */
#include "crass_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CONS_ARRAY_RL_MULTIPLIER 4          /* crassDefines.h:72 */
#define CONS_ARRAY_START 0.5                /* :73 */
#define MIN_READ_DEPTH 2                    /* :77 */
#define ZONE_EXT_CONS_CUT_OFF 0.55          /* :78 */
#define COLLAPSED_CONS_CUT_OFF 0.75         /* :79 */
#define COLLAPSED_THRESHOLD 0.30            /* :80 */
#define PARTIAL_SIM_CUT_OFF 0.85            /* :81 */
#define MIN_PARTIAL_LENGTH 4                /* :82 */
#define KMER_MAX_ABUNDANCE_CUTOFF 0.23      /* :92 */
#define KSW_XBYTE 0x10000
#define KSW_XSTOP 0x20000
#define KSW_XSUBO 0x40000
#define KSW_XSTART 0x80000

static void *xm(size_t n) { void *p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); } return p; }
static void *xc(size_t n, size_t s) { void *p = calloc(n ? n : 1, s ? s : 1); if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); } return p; }
static void *xr(void *q, size_t n) { void *p = realloc(q, n ? n : 1); if (!p) { fprintf(stderr, "oracle: out of memory\n"); abort(); } return p; }

/* ------------------------------------------------------------------------- */
/* ksw (16-bit striped Smith-Waterman, 8 lanes per vector)                    */
/* ------------------------------------------------------------------------- */
typedef struct { int qlen, slen, m; int16_t *qp; /* [m][slen][8] */ } kq_t;

static kq_t *kq_init(int qlen, const uint8_t *query, int m, const int8_t *mat)
{   /* ksw_qinit, size == 2 (ksw.c:57-101) */
    kq_t *q = (kq_t *)xm(sizeof(kq_t));
    const int p = 8;
    q->slen = (qlen + p - 1) / p; q->qlen = qlen; q->m = m;
    q->qp = (int16_t *)xc((size_t)m * (size_t)(q->slen ? q->slen : 1) * 8, sizeof(int16_t));
    int16_t *t = q->qp;
    for (int a = 0; a < m; ++a) {
        const int nlen = q->slen * p;
        const int8_t *ma = mat + a * m;
        for (int i = 0; i < q->slen; ++i)
            for (int k = i; k < nlen; k += q->slen) *t++ = (int16_t)(k >= qlen ? 0 : ma[query[k]]);
    }
    return q;
}
static void kq_free(kq_t *q) { free(q->qp); free(q); }

static inline int16_t adds16(int16_t a, int16_t b) { int v = (int)a + (int)b; return (int16_t)(v > 32767 ? 32767 : v < -32768 ? -32768 : v); }
static inline int16_t subsu16(int16_t a, int16_t b) { uint16_t x = (uint16_t)a, y = (uint16_t)b; return (int16_t)(x > y ? x - y : 0); }
static inline int16_t max16(int16_t a, int16_t b) { return a > b ? a : b; }

typedef struct { int score, te, qe, tb, qb; } kr_t;

static kr_t k_i16(const kq_t *q, int tlen, const uint8_t *target, int gapo, int gape, int xtra)
{   /* ksw_i16, ksw.c:228-317 (score2 / te2 are never read by crass and are not computed) */
    const int slen = q->slen;
    const int endsc = (xtra & KSW_XSTOP) ? (xtra & 0xffff) : 0x10000;
    int te = -1, gmax = 0;
    const int16_t gapoe = (int16_t)(gapo + gape), gpe = (int16_t)gape;
    const size_t vn = (size_t)(slen ? slen : 1) * 8;
    int16_t *H0 = (int16_t *)xc(vn, 2), *H1 = (int16_t *)xc(vn, 2), *E = (int16_t *)xc(vn, 2), *Hmax = (int16_t *)xc(vn, 2);
    for (int i = 0; i < tlen; ++i) {
        int16_t f[8] = {0}, mx[8] = {0}, h[8], e[8];
        const int16_t *S = q->qp + (size_t)target[i] * slen * 8;
        h[0] = 0;
        for (int l = 1; l < 8; l++) h[l] = slen ? H0[(size_t)(slen - 1) * 8 + (l - 1)] : 0;     /* _mm_slli_si128(h, 2) */
        for (int j = 0; j < slen; ++j) {
            for (int l = 0; l < 8; l++) {
                h[l] = adds16(h[l], S[(size_t)j * 8 + l]);
                e[l] = E[(size_t)j * 8 + l];
                h[l] = max16(h[l], e[l]);
                h[l] = max16(h[l], f[l]);
                mx[l] = max16(mx[l], h[l]);
                H1[(size_t)j * 8 + l] = h[l];
                h[l] = subsu16(h[l], gapoe);
                e[l] = subsu16(e[l], gpe);
                e[l] = max16(e[l], h[l]);
                E[(size_t)j * 8 + l] = e[l];
                f[l] = subsu16(f[l], gpe);
                f[l] = max16(f[l], h[l]);
                h[l] = H0[(size_t)j * 8 + l];
            }
        }
        for (int k = 0; k < 16; ++k) {                     /* the lazy-F loop */
            for (int l = 7; l > 0; l--) f[l] = f[l - 1];
            f[0] = 0;
            for (int j = 0; j < slen; ++j) {
                int any = 0;
                for (int l = 0; l < 8; l++) {
                    int16_t hh = H1[(size_t)j * 8 + l];
                    hh = max16(hh, f[l]);
                    H1[(size_t)j * 8 + l] = hh;
                    hh = subsu16(hh, gapoe);
                    f[l] = subsu16(f[l], gpe);
                    if (f[l] > hh) any = 1;
                }
                if (!any) goto end_loop8;
            }
        }
end_loop8:;
        int imax = mx[0];
        for (int l = 1; l < 8; l++) if (mx[l] > imax) imax = mx[l];
        if (imax > gmax) {
            gmax = imax; te = i;
            memcpy(Hmax, H1, vn * 2);
            if (gmax >= endsc) break;
        }
        { int16_t *T = H1; H1 = H0; H0 = T; }
    }
    kr_t r = { gmax, te, -1, -1, -1 };
    {
        int max = -1;
        const int qlen8 = slen * 8;
        for (int i = 0; i < qlen8; ++i)
            if ((int)(uint16_t)Hmax[i] > max) { max = (uint16_t)Hmax[i]; r.qe = i / 8 + i % 8 * slen; }
    }
    free(H0); free(H1); free(E); free(Hmax);
    return r;
}

static void revseq(int l, uint8_t *s) { for (int i = 0; i < l >> 1; ++i) { uint8_t t = s[i]; s[i] = s[l - 1 - i]; s[l - 1 - i] = t; } }

/* ksw_align with qry == NULL semantics of the result (ksw.c:330-360); query / target are restored on return */
void orc_ksw_align(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat, int gapo, int gape, int xtra,
                   int *score, int *te, int *qe, int *tb, int *qb)
{
    kq_t *q = kq_init(qlen, query, m, mat);
    kr_t r = k_i16(q, tlen, target, gapo, gape, xtra);
    kq_free(q);
    if (!((xtra & KSW_XSTART) == 0 || ((xtra & KSW_XSUBO) && r.score < (xtra & 0xffff)))) {
        revseq(r.qe + 1, query); revseq(r.te + 1, target);
        q = kq_init(r.qe + 1, query, m, mat);
        kr_t rr = k_i16(q, tlen, target, gapo, gape, KSW_XSTOP | r.score);
        revseq(r.qe + 1, query); revseq(r.te + 1, target);
        kq_free(q);
        if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
    }
    *score = r.score; *te = r.te; *qe = r.qe; *tb = r.tb; *qb = r.qb;
}

/* ------------------------------------------------------------------------- */
/* smithWaterman (SmithWaterman.cpp:151-308)                                  */
/* ------------------------------------------------------------------------- */
static double find_max(double a, double b, double c, double d, int *index)
{   /* findMax, SmithWaterman.cpp:68-129 */
    if (b > a) {
        if (c > d) { if (c > b) { *index = 2; return c; } else { *index = 1; return b; } }
        else { if (d > b) { *index = 3; return d; } else { *index = 1; return b; } }
    } else {
        if (c > d) { if (c > a) { *index = 2; return c; } else { *index = 0; return a; } }
        else { if (d > a) { *index = 3; return d; } else { *index = 0; return a; } }
    }
}

/* returns 1 when the (a_ret, b_ret) pair is returned, 0 for ("", "") (then *aStart = *aEnd = 0).
 * a_ret = seqA[*a_off, *a_off + *a_len), b_ret = seqB[*b_off, *b_off + *b_len) */
int orc_smith_waterman(const char *seqA, int lenA, const char *seqB, int lenB, int *aStartAlign, int *aEndAlign,
                       int aStartSearch, int aSearchLen, double similarity, int *a_off, int *a_len, int *b_off, int *b_len)
{
    const int W = lenB + 1;
    double *matrix = (double *)xc((size_t)(aSearchLen + 1) * W, sizeof(double));
    int *Ii = (int *)xc((size_t)(aSearchLen + 1) * W, sizeof(int)), *Ij = (int *)xc((size_t)(aSearchLen + 1) * W, sizeof(int));
    double matrix_max = -1;
    int i_max = 0, j_max = 0;
    for (int i = 1; i <= aSearchLen; i++) {
        for (int j = 1; j <= lenB; j++) {
            int index = -1;
            const double sim = (seqA[i - 1 + aStartSearch] == seqB[j - 1]) ? 1.2 : -1;
            const double v = find_max(matrix[(size_t)(i - 1) * W + j - 1] + sim, matrix[(size_t)(i - 1) * W + j] + (-1),
                                      matrix[(size_t)i * W + j - 1] + (-1), 0, &index);
            matrix[(size_t)i * W + j] = v;
            if (v > matrix_max) { matrix_max = v; i_max = i; j_max = j; }
            switch (index) {
                case 0: Ii[(size_t)i * W + j] = i - 1; Ij[(size_t)i * W + j] = j - 1; break;
                case 1: Ii[(size_t)i * W + j] = i - 1; Ij[(size_t)i * W + j] = j; break;
                case 2: Ii[(size_t)i * W + j] = i; Ij[(size_t)i * W + j] = j - 1; break;
                default: Ii[(size_t)i * W + j] = i; Ij[(size_t)i * W + j] = j; break;
            }
        }
    }
    int current_i = i_max, current_j = j_max;
    int next_i = Ii[(size_t)current_i * W + current_j], next_j = Ij[(size_t)current_i * W + current_j];
    /* (i_max == j_max == 0 when nothing was filled: the reference reads uninitialised I_i[0][0]; callers only reach
     *  this with aSearchLen >= 1 and lenB >= 1) */
    while ((next_j != 0) && (next_i != 0) && ((current_i != next_i) || (current_j != next_j))) {
        current_i = next_i; current_j = next_j;
        next_i = Ii[(size_t)current_i * W + current_j]; next_j = Ij[(size_t)current_i * W + current_j];
    }
    current_i--; current_j--;
    if (0 > current_j) current_j = 0;
    if (0 > current_i) current_i = 0;
    *aStartAlign = current_i + aStartSearch;
    *aEndAlign = (*aStartAlign) + i_max - current_i - 1;
    free(matrix); free(Ii); free(Ij);
    /* a_ret = seqA.substr(current_i + aStartSearch, i_max - current_i + aStartSearch)  (sic: the length includes
     * aStartSearch; substr clamps at the end of the string and throws when pos > size) */
    int apos = current_i + aStartSearch, an = i_max - current_i + aStartSearch;
    if (apos > lenA) return -1;
    if (an < 0 || apos + an > lenA) an = lenA - apos;        /* (a negative count is npos as size_t) */
    int bpos = current_j, bn = j_max - current_j;
    if (bpos > lenB) return -1;
    if (bn < 0 || bpos + bn > lenB) bn = lenB - bpos;
    *a_off = apos; *a_len = an; *b_off = bpos; *b_len = bn;
    if (0 != similarity) {
        const double similarity_ld = 1.0 - (orc_levenshtein(seqA + apos, an, seqB + bpos, bn) / (double)an);
        if (similarity_ld >= similarity) return 1;
        *aStartAlign = 0; *aEndAlign = 0;
        *a_len = 0; *b_len = 0;
        return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------------- */
/* state                                                                      */
/* ------------------------------------------------------------------------- */
typedef struct { int *v; int n, cap; } ivec;
static void iv_push(ivec *a, int x) { if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 8; a->v = (int *)xr(a->v, sizeof(int) * (size_t)a->cap); } a->v[a->n++] = x; }
static ivec *iv_new(void) { return (ivec *)xc(1, sizeof(ivec)); }
static void iv_free(ivec *a) { if (a) { free(a->v); free(a); } }

typedef struct {
    uint64_t read;
    int L; char *seq;               /* RH_Seq in its current orientation */
    uint32_t *ss; int nss, cap;     /* RH_StartStops */
    uint8_t rc;                     /* 1: seq is the reverse complement of the INPUT read */
    uint8_t alive;                  /* 0: the ReadHolder was deleted (clearReadList) */
} crec;

struct orc_cons {
    orc_cons_view v;
    orc_params p;
    int max_read_len;
    crec *rec; int n_rec;
    /* StringCheck: tokens 2 .. next_tok */
    char **tok_str; int *tok_len; int tok_cap; int next_tok;
    ivec **reads_of;                /* mReads: token -> record ids in order; NULL = no list */
    ivec **group; int grp_cap;      /* mDR2GIDMap: GID -> tokens; NULL = none / cleaned */
    char **true_dr;                 /* mTrueDRs: GID -> laurenized DR or NULL */
    int next_gid;
    int error;
    /* flattened output */
    char *o_tok_chars; uint64_t *o_tok_off; int32_t *o_grp_gid; char *o_dr_chars; uint64_t *o_dr_off;
    uint32_t *o_grp_tokens; uint64_t *o_grp_off; uint8_t *o_alive, *o_rc; uint32_t *o_token, *o_nss; uint64_t *o_ss_off;
    uint32_t *o_ss; uint64_t *o_tokread_off; uint64_t *o_tokread_idx; uint8_t *o_tok_has_list;
};

static void grow_tokens(orc_cons *s, int tok)
{
    if (tok < s->tok_cap) return;
    int nc = s->tok_cap ? s->tok_cap : 64;
    while (nc <= tok) nc *= 2;
    s->tok_str = (char **)xr(s->tok_str, sizeof(char *) * (size_t)nc);
    s->tok_len = (int *)xr(s->tok_len, sizeof(int) * (size_t)nc);
    s->reads_of = (ivec **)xr(s->reads_of, sizeof(ivec *) * (size_t)nc);
    for (int i = s->tok_cap; i < nc; i++) { s->tok_str[i] = NULL; s->tok_len[i] = 0; s->reads_of[i] = NULL; }
    s->tok_cap = nc;
}
static int add_string(orc_cons *s, const char *str, int len)
{   /* StringCheck::addString, StringCheck.cpp:46-55: always a NEW token */
    int t = ++s->next_tok;
    grow_tokens(s, t);
    s->tok_str[t] = (char *)xm((size_t)len + 1);
    memcpy(s->tok_str[t], str, (size_t)len); s->tok_str[t][len] = 0;
    s->tok_len[t] = len;
    return t;
}
static void grow_groups(orc_cons *s, int gid)
{
    if (gid < s->grp_cap) return;
    int nc = s->grp_cap ? s->grp_cap : 64;
    while (nc <= gid) nc *= 2;
    s->group = (ivec **)xr(s->group, sizeof(ivec *) * (size_t)nc);
    s->true_dr = (char **)xr(s->true_dr, sizeof(char *) * (size_t)nc);
    for (int i = s->grp_cap; i < nc; i++) { s->group[i] = NULL; s->true_dr[i] = NULL; }
    s->grp_cap = nc;
}
static void clean_group(orc_cons *s, int gid) { if (gid < s->grp_cap && s->group[gid]) { iv_free(s->group[gid]); s->group[gid] = NULL; } }   /* WorkHorse.cpp:1382-1389 */
static void clear_read_list(orc_cons *s, int tok)
{   /* clearReadList (WorkHorse.cpp:127-143): deletes the holders, empties the list */
    ivec *l = s->reads_of[tok];
    if (!l) return;
    for (int i = 0; i < l->n; i++) if (l->v[i] >= 0) s->rec[l->v[i]].alive = 0;
    l->n = 0;
}

static void rec_revcomp(crec *r)
{   /* ReadHolder::reverseComplementSeq + reverseStartStops, ReadHolder.cpp:593-609,321-380 */
    char *t = (char *)xm((size_t)r->L + 1);
    orc_revcomp(r->seq, (size_t)r->L, t);
    memcpy(r->seq, t, (size_t)r->L);
    free(t);
    if (r->nss > 0) {
        uint32_t *tmp = (uint32_t *)xm(sizeof(uint32_t) * (size_t)r->nss);
        const int true_start_offset = r->L - (int)r->ss[r->nss - 1] - 1;
        uint32_t prev_pos_fixed = (uint32_t)true_start_offset, prev_pos_orig = r->ss[r->nss - 1];
        for (int k = r->nss - 1, w = 0; k >= 0; k--, w++) {
            const uint32_t gap = prev_pos_orig - r->ss[k];
            prev_pos_fixed += gap;
            tmp[w] = prev_pos_fixed;
            prev_pos_orig = r->ss[k];
        }
        memcpy(r->ss, tmp, sizeof(uint32_t) * (size_t)r->nss);
        free(tmp);
    }
    r->rc = !r->rc;
}
static void ss_reserve(crec *r, int need)
{
    if (need <= r->cap) return;
    r->cap = need + 8;
    r->ss = (uint32_t *)xr(r->ss, sizeof(uint32_t) * (size_t)r->cap);
}

/* ReadHolder::updateStartStops, ReadHolder.cpp:382-511 */
static void update_start_stops(orc_cons *s, crec *r, int frontOffset, const char *DR, int DR_length)
{
    for (int k = 0; k + 1 < r->nss; k += 2) {
        int usable_length = DR_length - 1;
        if (frontOffset >= (int)r->ss[k]) {
            const int amount_below_zero = frontOffset - (int)r->ss[k];
            usable_length = DR_length - amount_below_zero - 1;
            r->ss[k] = 0;
        } else r->ss[k] -= (uint32_t)frontOffset;
        r->ss[k + 1] = r->ss[k] + (uint32_t)usable_length;
        if (r->ss[k + 1] >= (uint32_t)r->L) r->ss[k + 1] = (uint32_t)r->L - 1;
    }
    if (r->nss == 0) return;
    const uint32_t lowSp = s->p.lowSpacerSize;
    if (r->ss[0] > lowSp) {
        int part_s = 0, part_e = 0, ao, al, bo, bl;
        const int got = orc_smith_waterman(r->seq, r->L, DR, DR_length, &part_s, &part_e, 0, (int)r->ss[0] - (int)lowSp,
                                           PARTIAL_SIM_CUT_OFF, &ao, &al, &bo, &bl);
        if (got < 0) { s->error = 3; return; }
        if (0 != part_e && part_e - part_s >= MIN_PARTIAL_LENGTH) {
            /* (DR->rfind(sp.second) + sp.second.length()) == DR->length() && 0 == part_s */
            int rf = -1;
            for (int q = DR_length - bl; q >= 0; q--) if (memcmp(DR + q, DR + bo, (size_t)bl) == 0) { rf = q; break; }
            if (rf >= 0 && rf + bl == DR_length && 0 == part_s) {
                ss_reserve(r, r->nss + 2);
                memmove(r->ss + 2, r->ss, sizeof(uint32_t) * (size_t)r->nss);
                r->ss[0] = 0; r->ss[1] = (uint32_t)part_e;
                r->nss += 2;
            }
        }
    }
    const uint32_t end_dist = (uint32_t)r->L - r->ss[r->nss - 1];
    if (end_dist > lowSp) {
        int part_s = 0, part_e = 0, ao, al, bo, bl;
        const int got = orc_smith_waterman(r->seq, r->L, DR, DR_length, &part_s, &part_e, (int)(r->ss[r->nss - 1] + lowSp),
                                           (int)(end_dist - lowSp), PARTIAL_SIM_CUT_OFF, &ao, &al, &bo, &bl);
        if (got < 0) { s->error = 3; return; }
        if (0 != part_e && part_e - part_s >= MIN_PARTIAL_LENGTH) {
            /* (len - 1 == part_e) && (0 == DR->find(sp.second)) */
            if ((r->L - 1) == part_e && bl <= DR_length && memcmp(DR, DR + bo, (size_t)bl) == 0) {
                int d = al - bl;
                if (d < 0) d = -d;
                uint32_t i = (uint32_t)(part_s + d), j = (uint32_t)part_e;      /* startStopsAdd, :263-297 */
                ss_reserve(r, r->nss + 2);
                if (j >= (uint32_t)r->L) j = (uint32_t)r->L - 1;
                r->ss[r->nss] = i; r->ss[r->nss + 1] = j;
                r->nss += 2;
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Aligner                                                                    */
/* ------------------------------------------------------------------------- */
typedef struct {
    int length;
    int *cov;                       /* AL_coverage [4][length] */
    char *cons; float *conserv;
    int *off; int off_cap;          /* AL_Offsets by token; INT_MIN = not in the map */
    int master_tok, master_len; uint8_t *master;
    int zone_start, zone_end, zone_set;
    int8_t mat[25];
    int gapo, gape, minsc, xtra;
} aligner;

static uint8_t nt4(char c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; } }
static int char_to_index(char c) { switch (c) { case 'C': case 'c': return 2; case 'G': case 'g': return 3; case 'T': case 't': return 4; default: return 1; } }   /* Aligner.cpp:61-70 */
#define COV(al, i, c) ((al)->cov[(size_t)(char_to_index(c) - 1) * (al)->length + (i)])

static void al_set_off(aligner *al, int tok, int v)
{
    if (tok >= al->off_cap) {
        int nc = al->off_cap ? al->off_cap : 64;
        while (nc <= tok) nc *= 2;
        al->off = (int *)xr(al->off, sizeof(int) * (size_t)nc);
        for (int i = al->off_cap; i < nc; i++) al->off[i] = INT_MIN;
        al->off_cap = nc;
    }
    al->off[tok] = v;
}
static int al_has_off(const aligner *al, int tok) { return tok < al->off_cap && al->off[tok] != INT_MIN; }

static aligner *al_new(int length)
{   /* Aligner ctor, Aligner.h:112-136: gapo 5, gape 2, minsc 5, xtra KSW_XSTART */
    aligner *al = (aligner *)xc(1, sizeof(aligner));
    al->length = length;
    al->cov = (int *)xc((size_t)length * 4, sizeof(int));
    al->cons = (char *)xm((size_t)length); memset(al->cons, 'N', (size_t)length);
    al->conserv = (float *)xc((size_t)length, sizeof(float));
    al->gapo = 5; al->gape = 2; al->minsc = 5; al->xtra = KSW_XSTART;
    if (al->minsc > 0) al->xtra |= KSW_XSUBO | al->minsc;
    int k = 0;
    for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) al->mat[k++] = i == j ? 1 : -3; al->mat[k++] = 0; }
    for (int j = 0; j < 5; ++j) al->mat[k++] = 0;
    return al;
}
static void al_free(aligner *al) { free(al->cov); free(al->cons); free(al->conserv); free(al->off); free(al->master); free(al); }

/* Aligner::placeReadsInCoverageArray, Aligner.cpp:364-418 */
static void place_reads(orc_cons *s, aligner *al, int tok)
{
    ivec *l = s->reads_of[tok];
    if (!l) { s->error = 4; return; }           /* (mReads->at(token) on a NULL list: the reference would crash) */
    const int cur_len = s->tok_len[tok];
    for (int q = 0; q < l->n; q++) {
        crec *r = &s->rec[l->v[q]];
        int a = 0, b = 1;
        while (b < r->nss && ((int)r->ss[b] - (int)r->ss[a]) != (cur_len - 1)) { a += 2; b += 2; }
        if (b >= r->nss) { s->error = 5; return; }      /* startStopsAt(): std::out_of_range in the reference */
        do {
            if (((int)r->ss[b] - (int)r->ss[a]) == (cur_len - 1)) {
                const int pos = al->off[tok] - (int)r->ss[a];
                for (int i = 0; i < r->L; i++) {
                    const int ib = i + pos;
                    if (ib >= al->length || ib < 0) { s->error = 6; return; }      /* "MEMORY CORRUPTION" in the reference */
                    COV(al, ib, r->seq[i])++;
                }
            }
            a += 2; b += 2;
            if (a >= (r->nss / 2) * 2) break;
        } while (((int)r->ss[b] - (int)r->ss[a]) == (cur_len - 1));
    }
}

/* Aligner::calculateDRZone, Aligner.cpp:454-485 */
static void calc_zone(orc_cons *s, aligner *al)
{
    ivec *l = s->reads_of[al->master_tok];
    if (!l) { s->error = 4; return; }
    for (int q = 0; q < l->n; q++) {
        crec *r = &s->rec[l->v[q]];
        int a = 0, b = 1;
        while (b < r->nss && ((int)r->ss[b] - (int)r->ss[a]) != (al->master_len - 1)) { a += 2; b += 2; }
        if (b >= r->nss) { s->error = 5; return; }
        const int pos = al->off[al->master_tok] - (int)r->ss[a];
        al->zone_start = pos + (int)r->ss[a];
        al->zone_end = pos + (int)r->ss[b];
        al->zone_set = 1;
        break;
    }
}

/* Aligner::setMasterDR, Aligner.cpp:73-86 */
static void al_set_master(orc_cons *s, aligner *al, int master)
{
    al->master_tok = master;
    al_set_off(al, master, (int)(al->length * CONS_ARRAY_START));
    al->master_len = s->tok_len[master];
    al->master = (uint8_t *)xm((size_t)al->master_len + 1);
    for (int i = 0; i < al->master_len; i++) al->master[i] = nt4(s->tok_str[master][i]);
    al->master[al->master_len] = 0;
    place_reads(s, al, master);
    calc_zone(s, al);
}

enum { F_REVERSED = 1, F_FAILED = 2, F_EQUAL = 4 };

/* Aligner::getOffsetAgainstMaster, Aligner.cpp:263-362 */
static int offset_against_master(aligner *al, const char *slave, int slen, int *flags)
{
    uint8_t *fw = (uint8_t *)xm((size_t)slen + 1), *rv = (uint8_t *)xm((size_t)slen + 1);
    char *rcs = (char *)xm((size_t)slen + 1);
    orc_revcomp(slave, (size_t)slen, rcs);
    for (int i = 0; i < slen; i++) { fw[i] = nt4(slave[i]); rv[i] = nt4(rcs[i]); }
    fw[slen] = rv[slen] = 0;
    int fs, fte, fqe, ftb, fqb, rs, rte, rqe, rtb, rqb;
    orc_ksw_align(slen, fw, al->master_len, al->master, 5, al->mat, al->gapo, al->gape, al->xtra, &fs, &fte, &fqe, &ftb, &fqb);
    orc_ksw_align(slen, rv, al->master_len, al->master, 5, al->mat, al->gapo, al->gape, al->xtra, &rs, &rte, &rqe, &rtb, &rqb);
    free(fw); free(rv); free(rcs);
    if (rs == fs) { *flags |= F_EQUAL; return 0; }
    int score, tb, qb;
    if (rs > fs) { score = rs; tb = rtb; qb = rqb; *flags |= F_REVERSED; }
    else { score = fs; tb = ftb; qb = fqb; }
    const int min_query_seq_coverage = slen / 2;
    if (min_query_seq_coverage > score) { *flags |= F_FAILED; return 0; }
    if (score < al->minsc) { *flags |= F_FAILED; return 0; }
    return tb - qb;
}

/* Aligner::extendSlaveDR, Aligner.cpp:421-450; returns the length written to out (0: no usable read) */
static int extend_slave(orc_cons *s, int tok, int slave_len, char *out)
{
    ivec *l = s->reads_of[tok];
    if (!l) { s->error = 4; return 0; }
    for (int q = 0; q < l->n; q++) {
        crec *r = &s->rec[l->v[q]];
        int a = 0, b = 1;
        while (b < r->nss && ((int)r->ss[b] - (int)r->ss[a]) != (slave_len - 1)) { a += 2; b += 2; }
        if (b >= r->nss) { s->error = 5; return 0; }
        if ((int)r->ss[a] - 2 < 0 || (int)r->ss[b] + 2 > r->L) continue;
        int pos = (int)r->ss[a] - 2, n = slave_len + 4;
        if (pos + n > r->L) n = r->L - pos;              /* substr clamps */
        memcpy(out, r->seq + pos, (size_t)n);
        return n;
    }
    return 0;
}

/* Aligner::alignSlave, Aligner.cpp:88-153.  *ptok is the group's list entry (replaced when the slave is reversed) */
static void align_slave(orc_cons *s, aligner *al, int *ptok)
{
    int tok = *ptok;
    al_set_off(al, tok, -1);
    int slen = s->tok_len[tok];
    char *slave = (char *)xm((size_t)slen + 8);
    memcpy(slave, s->tok_str[tok], (size_t)slen);
    int flags = 0;
    int offset = offset_against_master(al, slave, slen, &flags);
    if (flags & F_EQUAL) {
        char *ext = (char *)xm((size_t)slen + 8);
        const int elen = extend_slave(s, tok, slen, ext);
        flags = 0;
        offset = offset_against_master(al, ext, elen, &flags);       /* (an empty extension aligns with score 0 both ways) */
        free(ext);
        if (flags & F_EQUAL) flags |= F_FAILED;
    }
    if (flags & F_FAILED) { free(slave); return; }
    if (flags & F_REVERSED) {
        ivec *l = s->reads_of[tok];
        if (!l) { s->error = 4; free(slave); return; }
        for (int q = 0; q < l->n; q++) rec_revcomp(&s->rec[l->v[q]]);
        char *rcs = (char *)xm((size_t)slen + 1);
        orc_revcomp(slave, (size_t)slen, rcs);
        const int st = add_string(s, rcs, slen);
        free(rcs);
        s->reads_of[st] = s->reads_of[tok];
        s->reads_of[tok] = NULL;
        *ptok = st;
        tok = st;
    }
    al_set_off(al, tok, al->off[al->master_tok] + offset);
    place_reads(s, al, tok);
    free(slave);
}

/* Aligner::generateConsensus, Aligner.cpp:155-240 */
static void generate_consensus(orc_cons *s, aligner *al)
{
    static const char alphabet[4] = {'A', 'C', 'G', 'T'};
    int num_GT_zero = 0;
    for (int j = 0; j < al->length; j++) {
        int max_count = 0;
        float total_count = 0.0f;
        for (int i = 0; i < 4; i++) {
            const int c = al->cov[(size_t)i * al->length + j];
            total_count += (float)c;
            if (c > max_count) { max_count = c; al->cons[j] = alphabet[i]; }
        }
        if (total_count > MIN_READ_DEPTH) { al->conserv[j] = (float)max_count / total_count; num_GT_zero++; }
        else al->conserv[j] = 0;
    }
    if (!al->zone_set) { s->error = 7; return; }
    if (num_GT_zero >= MIN_READ_DEPTH) {
        while (al->zone_start > 0) {
            if (al->zone_start - 1 >= al->length) { s->error = 8; return; }     /* the reference reads past the vector */
            if ((double)al->conserv[al->zone_start - 1] < ZONE_EXT_CONS_CUT_OFF) al->zone_start++;
            else break;
        }
        while (al->zone_end < al->length - 1) {
            if (al->zone_end + 1 < 0) { s->error = 8; return; }
            if ((double)al->conserv[al->zone_end + 1] < ZONE_EXT_CONS_CUT_OFF) al->zone_end--;
            else break;
        }
    }
    while (al->zone_start > 0) {
        if (al->zone_start - 1 >= al->length) { s->error = 8; return; }
        if ((double)al->conserv[al->zone_start - 1] >= ZONE_EXT_CONS_CUT_OFF) al->zone_start--;
        else break;
    }
    while (al->zone_end < al->length - 1) {
        if (al->zone_end + 1 < 0) { s->error = 8; return; }
        if ((double)al->conserv[al->zone_end + 1] >= ZONE_EXT_CONS_CUT_OFF) al->zone_end++;
        else break;
    }
}

/* ------------------------------------------------------------------------- */
/* WorkHorse                                                                  */
/* ------------------------------------------------------------------------- */
typedef struct { int present[4]; int val[4]; } cmap4;      /* std::map<char,int> over A C G T (iteration order A < C < G < T) */
static int c4_idx(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }
static int c4_size(const cmap4 *m) { return m->present[0] + m->present[1] + m->present[2] + m->present[3]; }

static int parse_grouped_drs(orc_cons *s, int GID);

/* drHasHighlyAbundantKmers, libcrispr.cpp:1077-1117 */
static int dr_has_abundant_kmers(const char *dr, int len)
{
    if (len < 3) return -1;
    const int max_index = len - 3;
    int total = 0;
    /* std::map<std::string,int>: 3-mers over arbitrary bytes; true DRs are consensus strings over A C G T N */
    char seen[256][3]; int seen_cnt[256]; int n_seen = 0;
    for (int i = 0; i < max_index; i++) {
        int q;
        for (q = 0; q < n_seen; q++) if (memcmp(seen[q], dr + i, 3) == 0) break;
        if (q == n_seen) { if (n_seen == 256) return -1; memcpy(seen[n_seen], dr + i, 3); seen_cnt[n_seen++] = 0; }
        seen_cnt[q]++;
        total++;
    }
    int max_count = 0;
    for (int q = 0; q < n_seen; q++) if (seen_cnt[q] > max_count) max_count = seen_cnt[q];
    const float maxFrequency = (float)max_count / (float)total;
    return (double)maxFrequency > KMER_MAX_ABUNDANCE_CUTOFF ? 1 : 0;
}

/* WorkHorse::calculateDRConsensus, WorkHorse.cpp:801-938 */
static char *calc_dr_consensus(orc_cons *s, int GID, aligner *al, int *collapsedPos, cmap4 *opts, uint8_t *refined /* [length] */, int *true_len)
{
    static const char alphabet[4] = {'A', 'C', 'G', 'T'};
    generate_consensus(s, al);
    char *true_dr = (char *)xm((size_t)al->length + 1);
    int n = 0;
    if (s->error) { *true_len = 0; return true_dr; }
    for (int i = al->zone_start; i <= al->zone_end; i++) {
        if (i < 0 || i >= al->length) { s->error = 8; break; }
        (*collapsedPos)++;
        if ((double)al->conserv[i] >= COLLAPSED_CONS_CUT_OFF) { refined[i] = 1; true_dr[n++] = al->cons[i]; continue; }
        refined[i] = 0;
        const float total_count = (float)(al->cov[i] + al->cov[(size_t)al->length + i] + al->cov[(size_t)2 * al->length + i] + al->cov[(size_t)3 * al->length + i]);
        for (int k = 0; k < 4; k++) {
            const float nt_proportion = (float)((float)al->cov[(size_t)k * al->length + i] / total_count);
            if ((double)nt_proportion >= COLLAPSED_THRESHOLD) {
                /* collapsedOptions[alphabet[k]] = size() + nextFreeGID: operator[] inserts the key BEFORE size() is read */
                if (!opts->present[k]) opts->present[k] = 1;
                opts->val[k] = c4_size(opts) + s->next_gid;
                s->next_gid++;
            }
        }
        if (2 > c4_size(opts)) {
            memset(opts, 0, sizeof(*opts));
            true_dr[n++] = al->cons[i];
            refined[i] = 1;
            continue;
        }
        refined[i] = 0;
        cmap4 opts2; memset(&opts2, 0, sizeof(opts2));
        ivec *g = s->group[GID];
        for (int q = 0; q < g->n; q++) {
            const int tok = g->v[q];
            /* drAligner.offset(*dr_iter) is AL_Offsets[tok]: a token without an entry gets one with value 0 */
            if (!al_has_off(al, tok)) al_set_off(al, tok, 0);
            if (-1 != al->off[tok]) {
                const int p = *collapsedPos + al->zone_start;
                if ((p >= al->off[tok]) && (p - al->off[tok] < s->tok_len[tok])) {
                    const char decision_char = s->tok_str[tok][al->zone_start - al->off[tok] + *collapsedPos];
                    const int di = c4_idx(decision_char);
                    if (di < 0) { s->error = 9; continue; }          /* a non-ACGT decision character: std::map<char,int> with more keys */
                    /* collapsed_options2[c] = collapsedOptions[c]  (inserts c into collapsedOptions with 0 if absent) */
                    if (!opts->present[di]) { opts->present[di] = 1; opts->val[di] = 0; }
                    opts2.present[di] = 1; opts2.val[di] = opts->val[di];
                }
            }
        }
        if (2 > c4_size(&opts2)) {
            true_dr[n++] = al->cons[i];
            refined[i] = 1;
            memset(opts, 0, sizeof(*opts));
        } else {
            *opts = opts2;
            *collapsedPos += al->zone_start;
            i = al->zone_end + 1;
        }
    }
    (void)alphabet;
    true_dr[n] = 0;
    *true_len = n;
    return true_dr;
}

/* one pass over a token's reads: the decision character of the first start/stop that yields one; -1 = none.
 * want: accepted characters (bit mask over ACGT indices) */
static int read_decision_char(const crec *r, int dec_diff, unsigned want)
{
    for (int k = 0; k < r->nss; k += 2) {          /* ss_iter += 2 from begin(): STARTS only */
        const int pos = (int)r->ss[k] + dec_diff;
        if (pos > 0 && pos < r->L) {
            const int di = c4_idx(r->seq[pos]);
            if (di >= 0 && (want & (1u << di))) return di;
        }
    }
    return -1;
}

/* WorkHorse::splitGroupedDR, WorkHorse.cpp:940-1132 */
static void split_grouped_dr(orc_cons *s, const cmap4 *opts, aligner *al, int collapsed_pos, int GID)
{
    int char_gid[4] = {0, 0, 0, 0};
    unsigned opt_mask = 0;
    for (int k = 0; k < 4; k++) if (opts->present[k]) {
        const int group = s->next_gid++;
        grow_groups(s, group);
        s->group[group] = iv_new();
        char_gid[k] = group;
        opt_mask |= 1u << k;
    }
    ivec *g = s->group[GID];
    for (int q = 0; q < g->n; q++) {
        const int tok = g->v[q];
        if (!al_has_off(al, tok)) al_set_off(al, tok, 0);
        if (-1 == al->off[tok]) continue;
        const int off = al->off[tok], tlen = s->tok_len[tok];
        if (off <= collapsed_pos && collapsed_pos < off + tlen) {
            const int di = c4_idx(s->tok_str[tok][collapsed_pos - off]);
            if (di < 0 || !char_gid[di]) { s->error = 10; continue; }     /* mDR2GIDMap[0]->push_back: NULL dereference in the reference */
            iv_push(s->group[char_gid[di]], tok);
            continue;
        }
        const int dec_diff = collapsed_pos - off;
        ivec *l = s->reads_of[tok];
        if (!l) { s->error = 4; continue; }
        unsigned forms = 0;
        for (int i = 0; i < l->n; i++) {
            const int di = read_decision_char(&s->rec[l->v[i]], dec_diff, opt_mask);
            if (di >= 0) forms |= 1u << di;
        }
        const int n_forms = __builtin_popcount(forms);
        if (n_forms == 1) {
            /* the first read that shows a form decides (it can only be that one form) */
            const int di = __builtin_ctz(forms);
            iv_push(s->group[char_gid[di]], tok);
        } else if (n_forms == 0) {
            clear_read_list(s, tok);
            iv_free(s->reads_of[tok]);
            s->reads_of[tok] = NULL;
        } else {
            int form_tok[4] = {0, 0, 0, 0};
            for (int k = 0; k < 4; k++) if (forms & (1u << k)) {
                const int st = add_string(s, s->tok_str[tok], tlen);
                s->reads_of[st] = iv_new();
                form_tok[k] = st;
                iv_push(s->group[char_gid[k]], st);
            }
            l = s->reads_of[tok];                       /* (add_string may have moved the table) */
            for (int i = 0; i < l->n; i++) {
                const int di = read_decision_char(&s->rec[l->v[i]], dec_diff, forms);
                if (di >= 0) { iv_push(s->reads_of[form_tok[di]], l->v[i]); l->v[i] = -1; }
            }
            clear_read_list(s, tok);
            iv_free(s->reads_of[tok]);
            s->reads_of[tok] = NULL;
        }
    }
    clean_group(s, GID);
    for (int k = 0; k < 4; k++) if (char_gid[k]) parse_grouped_drs(s, char_gid[k]);
}

/* WorkHorse::parseGroupedDRs, WorkHorse.cpp:1135-1379 */
static int parse_grouped_drs(orc_cons *s, int GID)
{
    if (s->error) return 0;
    ivec *g = s->group[GID];
    /* findMasterDR, :711-748: the longest DR of the group (first of equals) */
    int master = -1;
    size_t longest = 0;
    for (int q = 0; q < g->n; q++) if ((size_t)s->tok_len[g->v[q]] > longest) { master = g->v[q]; longest = (size_t)s->tok_len[g->v[q]]; }
    if (master < 0) {
        /* an empty group (every member lost its reads in a split): getString(-1) throws in the reference */
        s->error = 11; return 0;
    }
    aligner *al = al_new(CONS_ARRAY_RL_MULTIPLIER * s->max_read_len);
    al_set_master(s, al, master);
    /* populateCoverageArray, :750-798 */
    for (int q = 0; q < g->n && !s->error; q++) {
        if (al->master_tok == g->v[q]) continue;
        align_slave(s, al, &g->v[q]);
    }
    for (int q = 0; q < g->n && !s->error;) {
        const int tok = g->v[q];
        if (al_has_off(al, tok) && al->off[tok] == -1 && s->reads_of[tok] != NULL) {
            clear_read_list(s, tok);
            s->reads_of[tok] = NULL;               /* (the emptied list object leaks in the reference) */
            memmove(g->v + q, g->v + q + 1, sizeof(int) * (size_t)(g->n - q - 1));
            g->n--;
            continue;
        }
        q++;
    }
    if (s->error) { al_free(al); return 0; }
    int collapsed_pos = -1;
    cmap4 opts; memset(&opts, 0, sizeof(opts));
    uint8_t *refined = (uint8_t *)xc((size_t)al->length + 2, 1);
    int true_len = 0;
    char *true_DR = calc_dr_consensus(s, GID, al, &collapsed_pos, &opts, refined, &true_len);
    int ret = 1;
    if (s->error) { ret = 0; goto done; }
    if ((unsigned)true_len > s->p.highDRsize) { clean_group(s, GID); ret = 0; goto done; }
    if (c4_size(&opts) == 0) {
        if ((unsigned)true_len < s->p.lowDRsize) { clean_group(s, GID); ret = 0; goto done; }
        if (orc_is_low_complexity(true_DR, true_len)) { clean_group(s, GID); ret = 0; goto done; }
        const int ab = dr_has_abundant_kmers(true_DR, true_len);
        if (ab < 0) { s->error = 12; ret = 0; goto done; }
        if (ab) { clean_group(s, GID); ret = 0; goto done; }
        int dr_zone_start = al->zone_start, dr_zone_end = al->zone_end;
        int diffs = dr_zone_end - dr_zone_start + 1 - true_len;
        int guard = 0;
        while (0 < diffs) {
            /* refined_DR_ends[x] on a std::map<int,bool>: a missing key reads as false */
            const int re = (dr_zone_end >= 0 && dr_zone_end < al->length) ? refined[dr_zone_end] : 0;
            if (!re) { dr_zone_end--; diffs--; }
            if (0 < diffs) {
                const int rs = (dr_zone_start >= 0 && dr_zone_start < al->length) ? refined[dr_zone_start] : 0;
                if (!rs) { dr_zone_start++; diffs--; }
            }
            if (++guard > 4 * al->length) { s->error = 13; ret = 0; goto done; }     /* both ends refined: the reference spins forever */
        }
        al->zone_start = dr_zone_start; al->zone_end = dr_zone_end;
    }
    if (c4_size(&opts) > 0) {
        split_grouped_dr(s, &opts, al, collapsed_pos, GID);
    } else {
        char *rcd = (char *)xm((size_t)true_len + 1);
        orc_revcomp(true_DR, (size_t)true_len, rcd);
        /* laurenize (SeqUtils.cpp:89-97): seq < revcomp ? seq : revcomp */
        int less = 0;
        { int c = memcmp(true_DR, rcd, (size_t)true_len); less = c < 0; }
        const char *lau = less ? true_DR : rcd;
        const int rev_comp = memcmp(lau, true_DR, (size_t)true_len) != 0;
        grow_groups(s, GID);
        free(s->true_dr[GID]);
        s->true_dr[GID] = (char *)xm((size_t)true_len + 1);
        memcpy(s->true_dr[GID], lau, (size_t)true_len); s->true_dr[GID][true_len] = 0;
        free(rcd);
        g = s->group[GID];
        for (int q = 0; q < g->n && !s->error; q++) {
            const int tok = g->v[q];
            if (!al_has_off(al, tok) || al->off[tok] == -1) continue;           /* logError only */
            ivec *l = s->reads_of[tok];
            if (!l) { s->error = 4; break; }
            for (int i = 0; i < l->n && !s->error; i++) {
                crec *r = &s->rec[l->v[i]];
                update_start_stops(s, r, al->off[tok] - al->zone_start, true_DR, true_len);
                if (rev_comp) rec_revcomp(r);
            }
        }
    }
done:
    free(true_DR); free(refined); al_free(al);
    return ret;
}

/* WorkHorse::combineGroupsWithIdenticalDRs, WorkHorse.cpp:416-452 */
static void combine_groups(orc_cons *s)
{
    for (int gid = 1; gid < s->grp_cap; gid++) {
        if (!s->true_dr[gid]) continue;
        int prev = 0;
        for (int h = 1; h < gid; h++) if (s->true_dr[h] && strcmp(s->true_dr[h], s->true_dr[gid]) == 0) { prev = h; break; }
        if (!prev) continue;
        /* (std::map<std::string,int>: the FIRST group seen with that DR, i.e. the lowest GID) */
        if (!s->group[gid] || !s->group[prev]) { s->error = 14; return; }
        for (int q = 0; q < s->group[gid]->n; q++) iv_push(s->group[prev], s->group[gid]->v[q]);
        iv_free(s->group[gid]); s->group[gid] = NULL;
        free(s->true_dr[gid]); s->true_dr[gid] = NULL;
    }
}

orc_cons *orc_consensus_run(const orc_cons_input *in, const orc_params *p)
{
    orc_cons *s = (orc_cons *)xc(1, sizeof(orc_cons));
    s->p = *p;
    s->max_read_len = (int)in->max_read_len;
    s->n_rec = (int)in->n_rec;
    s->rec = (crec *)xc((size_t)s->n_rec, sizeof(crec));
    s->next_tok = 1;
    for (uint32_t t = 0; t < in->n_tokens; t++)
        add_string(s, in->tok_chars + in->tok_off[t], (int)(in->tok_off[t + 1] - in->tok_off[t]));
    for (int k = 0; k < s->n_rec; k++) {
        crec *r = &s->rec[k];
        r->read = in->rec_read[k];
        const char *src = in->seqs + in->seq_off[r->read];
        r->L = (int)(in->seq_off[r->read + 1] - in->seq_off[r->read]);
        r->seq = (char *)xm((size_t)r->L + 1);
        if (in->rec_lowlexi[k]) memcpy(r->seq, src, (size_t)r->L);
        else orc_revcomp(src, (size_t)r->L, r->seq);              /* DRLowLexi flipped the holder (ReadHolder.cpp:573-590) */
        r->rc = in->rec_lowlexi[k] ? 0 : 1;
        r->nss = (int)in->rec_nss[k];
        r->cap = r->nss + 8;
        r->ss = (uint32_t *)xm(sizeof(uint32_t) * (size_t)r->cap);
        memcpy(r->ss, in->ss_pool + in->rec_ss_off[k], sizeof(uint32_t) * (size_t)r->nss);
        r->alive = 1;
        const int tok = (int)in->rec_token[k];
        if (tok < 2 || tok > s->next_tok) { s->error = 20; continue; }
        if (!s->reads_of[tok]) s->reads_of[tok] = iv_new();
        iv_push(s->reads_of[tok], k);
    }
    s->next_gid = (int)in->n_groups + 1;
    grow_groups(s, s->next_gid + 8);
    for (uint32_t g = 0; g < in->n_groups; g++) {
        /* a GID without tokens = a NULL entry of mDR2GIDMap: the reference `continue`s over it (WorkHorse.cpp:592-595) */
        if (in->grp_off[g + 1] == in->grp_off[g]) continue;
        s->group[g + 1] = iv_new();
        for (uint64_t q = in->grp_off[g]; q < in->grp_off[g + 1]; q++) iv_push(s->group[g + 1], (int)in->grp_tokens[q]);
    }
    /* findConsensusDRs, WorkHorse.cpp:578-611: the ORIGINAL groups in ascending GID order */
    const int n_orig = (int)in->n_groups;
    for (int gid = 1; gid <= n_orig && !s->error; gid++) {
        if (!s->group[gid]) continue;
        parse_grouped_drs(s, gid);
        if (!s->error) combine_groups(s);
    }
    /* ---- flatten ---- */
    orc_cons_view *v = &s->v;
    v->error = s->error; v->next_free_gid = s->next_gid;
    v->n_tokens = (uint32_t)(s->next_tok - 1);
    size_t tc = 0;
    for (int t = 2; t <= s->next_tok; t++) tc += (size_t)s->tok_len[t];
    s->o_tok_chars = (char *)xm(tc + 1); s->o_tok_off = (uint64_t *)xm(sizeof(uint64_t) * ((size_t)v->n_tokens + 1));
    tc = 0; s->o_tok_off[0] = 0;
    for (int t = 2; t <= s->next_tok; t++) { memcpy(s->o_tok_chars + tc, s->tok_str[t], (size_t)s->tok_len[t]); tc += (size_t)s->tok_len[t]; s->o_tok_off[t - 1] = tc; }
    uint32_t ng = 0; size_t ntok_g = 0, dc = 0;
    for (int gid = 1; gid < s->grp_cap; gid++) if (s->group[gid] && s->true_dr[gid]) { ng++; ntok_g += (size_t)s->group[gid]->n; dc += strlen(s->true_dr[gid]); }
    v->n_groups = ng;
    s->o_grp_gid = (int32_t *)xm(sizeof(int32_t) * (ng + 1)); s->o_dr_chars = (char *)xm(dc + 1); s->o_dr_off = (uint64_t *)xm(sizeof(uint64_t) * (ng + 1));
    s->o_grp_tokens = (uint32_t *)xm(sizeof(uint32_t) * (ntok_g + 1)); s->o_grp_off = (uint64_t *)xm(sizeof(uint64_t) * (ng + 1));
    ng = 0; ntok_g = 0; dc = 0; s->o_dr_off[0] = 0; s->o_grp_off[0] = 0;
    for (int gid = 1; gid < s->grp_cap; gid++) if (s->group[gid] && s->true_dr[gid]) {
        s->o_grp_gid[ng] = gid;
        const size_t l = strlen(s->true_dr[gid]);
        memcpy(s->o_dr_chars + dc, s->true_dr[gid], l); dc += l;
        for (int q = 0; q < s->group[gid]->n; q++) s->o_grp_tokens[ntok_g++] = (uint32_t)s->group[gid]->v[q];
        ng++;
        s->o_dr_off[ng] = dc; s->o_grp_off[ng] = ntok_g;
    }
    const size_t nr = (size_t)s->n_rec;
    s->o_alive = (uint8_t *)xc(nr, 1); s->o_rc = (uint8_t *)xc(nr, 1); s->o_token = (uint32_t *)xc(nr, 4); s->o_nss = (uint32_t *)xc(nr, 4);
    s->o_ss_off = (uint64_t *)xc(nr + 1, 8);
    size_t sst = 0;
    for (size_t k = 0; k < nr; k++) sst += (size_t)s->rec[k].nss;
    s->o_ss = (uint32_t *)xm(sizeof(uint32_t) * (sst + 1));
    sst = 0;
    for (size_t k = 0; k < nr; k++) {
        const crec *r = &s->rec[k];
        s->o_alive[k] = r->alive; s->o_rc[k] = r->rc; s->o_nss[k] = (uint32_t)r->nss; s->o_ss_off[k] = sst;
        memcpy(s->o_ss + sst, r->ss, sizeof(uint32_t) * (size_t)r->nss); sst += (size_t)r->nss;
    }
    s->o_tokread_off = (uint64_t *)xc((size_t)v->n_tokens + 1, 8); s->o_tok_has_list = (uint8_t *)xc((size_t)v->n_tokens + 1, 1);
    size_t nl = 0;
    for (int t = 2; t <= s->next_tok; t++) if (s->reads_of[t]) nl += (size_t)s->reads_of[t]->n;
    s->o_tokread_idx = (uint64_t *)xm(sizeof(uint64_t) * (nl + 1));
    nl = 0;
    for (int t = 2; t <= s->next_tok; t++) {
        s->o_tokread_off[t - 2] = nl;
        if (s->reads_of[t]) {
            s->o_tok_has_list[t - 2] = 1;
            for (int q = 0; q < s->reads_of[t]->n; q++) { s->o_tokread_idx[nl++] = (uint64_t)s->reads_of[t]->v[q]; s->o_token[s->reads_of[t]->v[q]] = (uint32_t)t; }
        }
    }
    s->o_tokread_off[v->n_tokens] = nl;
    v->tok_chars = s->o_tok_chars; v->tok_off = s->o_tok_off; v->grp_gid = s->o_grp_gid; v->dr_chars = s->o_dr_chars; v->dr_off = s->o_dr_off;
    v->grp_tokens = s->o_grp_tokens; v->grp_off = s->o_grp_off; v->n_rec = (uint64_t)nr; v->rec_alive = s->o_alive; v->rec_rc = s->o_rc;
    v->rec_token = s->o_token; v->rec_nss = s->o_nss; v->rec_ss_off = s->o_ss_off; v->ss_pool = s->o_ss;
    v->tokread_off = s->o_tokread_off; v->tokread_idx = s->o_tokread_idx; v->tok_has_list = s->o_tok_has_list;
    return s;
}

void orc_consensus_view(const orc_cons *s, orc_cons_view *v) { *v = s->v; }

void orc_consensus_free(orc_cons *s)
{
    if (!s) return;
    for (int k = 0; k < s->n_rec; k++) { free(s->rec[k].seq); free(s->rec[k].ss); }
    free(s->rec);
    for (int t = 0; t < s->tok_cap; t++) { free(s->tok_str[t]); iv_free(s->reads_of[t]); }
    free(s->tok_str); free(s->tok_len); free(s->reads_of);
    for (int g = 0; g < s->grp_cap; g++) { iv_free(s->group[g]); free(s->true_dr[g]); }
    free(s->group); free(s->true_dr);
    free(s->o_tok_chars); free(s->o_tok_off); free(s->o_grp_gid); free(s->o_dr_chars); free(s->o_dr_off); free(s->o_grp_tokens);
    free(s->o_grp_off); free(s->o_alive); free(s->o_rc); free(s->o_token); free(s->o_nss); free(s->o_ss_off); free(s->o_ss);
    free(s->o_tokread_off); free(s->o_tokread_idx); free(s->o_tok_has_list);
    free(s);
}
