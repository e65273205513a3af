/*
 * ref_shim.cpp — extern "C" harness over the *unmodified, compiled reference sources*.
 *
 * TEST INFRASTRUCTURE ONLY.  oracle/Makefile compiles this file together with
 *   /root/reference/src/crass/{PatternMatcher,StringCheck,kseq}.cpp, ksw.c (klib's SSE2 Smith-Waterman, includes only ksw.h)
 *   /root/reference/src/aho-corasick/{acism,acism_create,acism_file,msutil}.c
 * straight from where they lie into oracle/_ref/libcrass_ref.so (git-ignored).
 * Those are the hot-path reference files that compile without generated code
 * (autoconf config.h) or Xerces-C; libcrispr.cpp / ReadHolder.cpp / WorkHorse.cpp
 * need both and are therefore NOT built (see DESIGN.md "Oracle").
 *
 * The functions here only marshal arguments; the algorithms are the reference's.
 */
#include <string>
#include <vector>
#include <cstring>
#include <cstdlib>
#include <cstdint>
#include <zlib.h>
#include <time.h>

#include "PatternMatcher.h"
#include "StringCheck.h"
#include "kseq.h"

extern "C" {
#include "msutil.h"
#include "acism.h"
}
#include "ksw.h"

extern "C" {

/* PatternMatcher::bmpSearch (PatternMatcher.cpp:26) */
int ref_bmp_search(const char *text, size_t tlen, const char *pat, size_t plen)
{
    std::string t(text, tlen), p(pat, plen);
    return PatternMatcher::bmpSearch(t, p);
}

/* PatternMatcher::levenstheinDistance (PatternMatcher.cpp:111) */
int ref_levenshtein(const char *s, int n, const char *t, int m)
{
    std::string a(s, (size_t)n), b(t, (size_t)m);
    return PatternMatcher::levenstheinDistance(a, b);
}

/* PatternMatcher::getStringSimilarity (PatternMatcher.cpp:197) */
float ref_similarity(const char *s, int n, const char *t, int m)
{
    std::string a(s, (size_t)n), b(t, (size_t)m);
    return PatternMatcher::getStringSimilarity(a, b);
}

/* ---- ACISM driven exactly the way findSingletons drives it (libcrispr.cpp:452-469,503,441) */
typedef struct {
    char *concstr;
    MEMREF *pattv;
    int npatts;
    ACISM *psp;
} ref_acism;

typedef struct { int strnum; int textpos; int calls; } ref_hit;

static int ref_on_match(int strnum, int textpos, void *ctx)
{
    ref_hit *h = (ref_hit *)ctx;
    h->strnum = strnum; h->textpos = textpos; h->calls++;
    return 1;                        /* on_match always returns 1 (libcrispr.cpp:441) */
}

void *ref_acism_build(const char *const *pats, const uint32_t *lens, uint32_t n)
{
    std::string conc;
    for (uint32_t i = 0; i < n; i++) { conc.append(pats[i], lens[i]); conc += "\n"; }
    ref_acism *r = (ref_acism *)calloc(1, sizeof(ref_acism));
    r->concstr = (char *)malloc(conc.size() + 1);
    memcpy(r->concstr, conc.data(), conc.size());
    r->concstr[conc.size()] = '\0';
    r->concstr[conc.size() - 1] = '\0';          /* the trailing-newline hack (libcrispr.cpp:458-464) */
    r->pattv = refsplit(r->concstr, '\n', &r->npatts);
    r->psp = acism_create(r->pattv, r->npatts);
    return r;
}

/* returns 1 if the callback fired; (*end_excl, *len) from the first callback */
int ref_acism_first(void *handle, const char *text, size_t tlen, uint32_t *end_excl, uint32_t *len)
{
    ref_acism *r = (ref_acism *)handle;
    ref_hit h = { -1, -1, 0 };
    MEMREF tmp = { text, tlen };
    (void)acism_scan(r->psp, tmp, (ACISM_ACTION *)ref_on_match, &h);
    if (!h.calls) return 0;
    *end_excl = (uint32_t)h.textpos;
    *len = (uint32_t)r->pattv[h.strnum].len;
    return 1;
}

void ref_acism_free(void *handle)
{
    ref_acism *r = (ref_acism *)handle;
    if (!r) return;
    acism_destroy(r->psp);
    free(r->pattv);
    free(r->concstr);
    free(r);
}

/* ---- StringCheck (StringCheck.cpp:46-81): feed strings in order, get tokens back */
void ref_stringcheck_tokens(const char *const *strs, const uint32_t *lens, uint32_t n, int32_t *tokens_out)
{
    StringCheck sc;
    for (uint32_t i = 0; i < n; i++) {
        std::string s(strs[i], lens[i]);
        StringToken t = sc.getToken(s);
        if (0 == t) t = sc.addString(s);
        tokens_out[i] = t;
    }
}

/* ---- kseq (kseq.cpp:171-226): read every record of a FASTA/FASTQ(.gz) file.
 * Output is a flat byte stream: for each record four length-prefixed (u32 LE) fields
 * name, comment, seq, qual — comment/qual are what searchFile would pass to setComment/
 * setQual (i.e. the *stale pointer* semantics of libcrispr.cpp:124-131: the field is emitted
 * whenever the kstring's buffer pointer is non-NULL).  Returns number of records or <0. */
long ref_kseq_dump(const char *path, unsigned char **out, size_t *out_len, int *last_ret)
{
    gzFile fp = gzopen(path, "r");
    if (!fp) return -1;
    kseq_t *seq = kseq_init(fp);
    std::string buf;
    long n = 0;
    int l;
    auto put = [&buf](const char *p) {
        uint32_t len = p ? (uint32_t)strlen(p) : 0xFFFFFFFFu;
        buf.append((const char *)&len, 4);
        if (p) buf.append(p, len);
    };
    while ((l = kseq_read(seq)) >= 0) {
        put(seq->name.s);
        put(seq->comment.s);
        put(seq->seq.s);
        put(seq->qual.s);
        n++;
    }
    if (last_ret) *last_ret = l;
    kseq_destroy(seq);
    gzclose(fp);
    *out = (unsigned char *)malloc(buf.size() ? buf.size() : 1);
    memcpy(*out, buf.data(), buf.size());
    *out_len = buf.size();
    return n;
}

void ref_free(void *p) { free(p); }

/* ksw_align (ksw.c:330-360) exactly as Aligner::getOffsetAgainstMaster calls it (Aligner.cpp:280-301): query profile
 * allocated by the call (*qry == NULL) and freed here */
void ref_ksw_align(int qlen, uint8_t *query, int tlen, uint8_t *target, int m, const int8_t *mat, int gapo, int gape, int xtra,
                   int *score, int *te, int *qe, int *tb, int *qb)
{
    kswq_t *qp = 0;
    kswr_t r = ksw_align(qlen, query, tlen, target, m, mat, gapo, gape, xtra, &qp);
    free(qp);
    *score = r.score; *te = r.te; *qe = r.qe; *tb = r.tb; *qb = r.qb;
}

/* ---- calibration loops (see oracle/crass_oracle.c orc_calib_*): the reference's two hot
 * functions driven the way searchCore (libcrispr.cpp:295-339: std::string substr of the read
 * for text and pattern, PatternMatcher::bmpSearch) and findSingletons (libcrispr.cpp:500-503)
 * drive them. */
static double calib_now()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double ref_calib_bmp(const char *seqs, uint64_t n_reads, int L, unsigned lowDR, unsigned highDR, unsigned lowSp,
                     unsigned highSp, unsigned w, uint64_t *checksum)
{
    unsigned skips = lowDR - (2 * w - 1);
    if (skips < 1) skips = 1;
    const int search_end = L - (int)lowDR - (int)lowSp - (int)w - 1;
    uint64_t sum = 0;
    const double t0 = calib_now();
    for (uint64_t r = 0; r < n_reads; r++) {
        std::string read(seqs + r * (uint64_t)L, (size_t)L);
        for (int j = 0; j <= search_end; j += (int)skips) {
            int begin = j + (int)lowDR + (int)lowSp;
            int end = j + (int)highDR + (int)highSp + (int)w;
            if (begin >= L) begin = L - 1;
            if (end >= L) end = L - 1;
            if (end - begin < (int)w) break;
            std::string text = read.substr((size_t)begin, (size_t)(end - begin));
            std::string pattern = read.substr((size_t)j, w);
            sum += (uint64_t)(int64_t)PatternMatcher::bmpSearch(text, pattern);
        }
    }
    const double t1 = calib_now();
    *checksum = sum;
    return t1 - t0;
}

double ref_calib_acism(void *handle, const char *seqs, uint64_t n_reads, int L, uint64_t *checksum)
{
    uint64_t sum = 0;
    const double t0 = calib_now();
    for (uint64_t r = 0; r < n_reads; r++) {
        uint32_t e = 0, l = 0;
        if (ref_acism_first(handle, seqs + r * (uint64_t)L, (size_t)L, &e, &l)) sum += ((uint64_t)e << 8) + l;
    }
    const double t1 = calib_now();
    *checksum = sum;
    return t1 - t0;
}

} /* extern "C" */
