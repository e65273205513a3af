// ingest.cpp — the ingest side of the boundary: 2-bit packer with an exception list,
// FASTA/FASTQ(.gz) reader with kseq_read record semantics, and the deterministic synthetic
// metagenome generator used by bench.py and the parity tests.  Host-only C++17 (+ zlib).
#include "../../include/crass_hip.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include <zlib.h>

namespace {

inline int base_code(uint8_t c)
{
    switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3; default: return -1; }
}

struct PackedOwner {
    std::vector<uint32_t> packed;
    std::vector<uint64_t> word_off;
    std::vector<uint32_t> lengths;
    std::vector<uint64_t> exc_read, exc_off;
    std::vector<uint8_t> exc_bytes;
};

unsigned hw_threads()
{
    unsigned n = std::thread::hardware_concurrency();
    return n ? n : 1;
}

template <typename F> void parallel_ranges(uint64_t n, unsigned nt, F f)
{
    if (nt <= 1 || n < 4096) { f(0, n, 0u); return; }
    std::vector<std::thread> th;
    uint64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        uint64_t a = std::min<uint64_t>(n, t * per), b = std::min<uint64_t>(n, a + per);
        if (a >= b) break;
        th.emplace_back([=]() { f(a, b, t); });
    }
    for (auto &x : th) x.join();
}

} // namespace

extern "C" {

int crass_pack_reads(const uint8_t *seqs, const uint64_t *off, uint64_t n, int pad_uniform, crass_packed *out)
{
    if (!out || (n && (!seqs || !off))) return CRASS_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    PackedOwner *o = new PackedOwner();
    uint32_t max_len = 0, min_len = 0xFFFFFFFFu;
    for (uint64_t i = 0; i < n; i++) {
        uint64_t l = off[i + 1] - off[i];
        if (l > CRASS_HIP_MAX_READ_LEN) { delete o; return CRASS_ERR_UNSUPPORTED; }
        max_len = std::max<uint32_t>(max_len, (uint32_t)l);
        min_len = std::min<uint32_t>(min_len, (uint32_t)l);
    }
    if (n == 0) min_len = 0;
    const bool uniform_len = (n > 0 && max_len == min_len);
    uint32_t stride = 0;
    if (pad_uniform || uniform_len) stride = std::max<uint32_t>(1, (max_len + 15) / 16);
    if (!stride) {
        o->word_off.resize(n + 1);
        uint64_t w = 0;
        for (uint64_t i = 0; i < n; i++) { o->word_off[i] = w; w += (off[i + 1] - off[i] + 15) / 16; }
        o->word_off[n] = w;
        o->packed.assign(w + 4, 0);
    } else {
        o->packed.assign(n * (uint64_t)stride + 4, 0);
    }
    if (!uniform_len) { o->lengths.resize(n); for (uint64_t i = 0; i < n; i++) o->lengths[i] = (uint32_t)(off[i + 1] - off[i]); }
    const unsigned nt = std::min<unsigned>(hw_threads(), 32);
    std::vector<std::vector<uint64_t>> exc_parts(nt);
    uint32_t *packed = o->packed.data();
    parallel_ranges(n, nt, [&](uint64_t a, uint64_t b, unsigned t) {
        for (uint64_t i = a; i < b; i++) {
            const uint8_t *s = seqs + off[i];
            const uint32_t L = (uint32_t)(off[i + 1] - off[i]);
            uint32_t *w = packed + (stride ? i * (uint64_t)stride : o->word_off[i]);
            bool bad = false;
            for (uint32_t k = 0; k < L; k++) {
                int c = base_code(s[k]);
                if (c < 0) { bad = true; c = 0; }
                w[k >> 4] |= (uint32_t)c << ((k & 15) * 2);
            }
            if (bad) exc_parts[t].push_back(i);
        }
    });
    for (auto &p : exc_parts) o->exc_read.insert(o->exc_read.end(), p.begin(), p.end());
    std::sort(o->exc_read.begin(), o->exc_read.end());
    o->exc_off.push_back(0);
    for (uint64_t r : o->exc_read) {
        o->exc_bytes.insert(o->exc_bytes.end(), seqs + off[r], seqs + off[r + 1]);
        o->exc_off.push_back(o->exc_bytes.size());
    }
    crass_reads &r = out->reads;
    r.n_reads = n; r.packed = o->packed.data(); r.stride_words = stride;
    r.word_off = stride ? nullptr : o->word_off.data();
    r.uniform_len = uniform_len ? max_len : 0;
    r.lengths = uniform_len ? nullptr : o->lengths.data();
    r.n_exceptions = o->exc_read.size();
    r.exc_read = o->exc_read.data(); r.exc_off = o->exc_off.data(); r.exc_bytes = o->exc_bytes.data();
    r.header_id = nullptr; r.read_index_base = 0;
    out->owner = o;
    return CRASS_OK;
}

void crass_free_packed(crass_packed *p)
{
    if (!p) return;
    delete static_cast<PackedOwner *>(p->owner);
    memset(p, 0, sizeof(*p));
}

// ---- FASTA/FASTQ reader: kseq_read record semantics (kseq.cpp:171-226) as driven by
// searchFile (libcrispr.cpp:96-131).  The whole (decompressed) file is parsed from memory.
static bool is_space(int c) { return c == ' ' || (c >= '\t' && c <= '\r'); }

int crass_read_fastx(const char *path, crass_fastx *out)
{
    if (!path || !out) return CRASS_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    gzFile fp = gzopen(path, "r");
    if (!fp) return CRASS_ERR_IO;
    std::vector<uint8_t> data;
    {
        std::vector<uint8_t> buf(1 << 20);
        int got;
        while ((got = gzread(fp, buf.data(), (unsigned)buf.size())) > 0) data.insert(data.end(), buf.begin(), buf.begin() + got);
        gzclose(fp);
        if (got < 0) return CRASS_ERR_IO;
    }
    const size_t n = data.size();
    size_t pos = 0;
    std::vector<uint8_t> seq, name, comment, qual, has_c, has_q;
    std::vector<uint64_t> seq_off{0}, name_off{0}, comment_off{0}, qual_off{0}, header_id;
    std::unordered_map<std::string, uint64_t> first_seen;
    std::string stale_comment, stale_qual;
    bool any_comment = false, any_qual = false;
    int last_char = 0;
    int last_ret = -1;
    uint32_t max_len = 0;
    uint64_t nrec = 0;
    for (;;) {
        if (last_char == 0) {
            while (pos < n && data[pos] != '>' && data[pos] != '@') pos++;
            if (pos >= n) { last_ret = -1; break; }
            last_char = data[pos++];
        }
        if (pos >= n) { last_ret = -1; break; }          // ks_getuntil < 0 at EOF
        size_t st = pos;
        while (pos < n && !is_space(data[pos])) pos++;
        std::string nm((const char *)data.data() + st, pos - st);
        int c = pos < n ? data[pos] : -1;
        pos++;
        if (c != -1 && c != '\n') {
            st = pos;
            while (pos < n && data[pos] != '\n') pos++;
            stale_comment.assign((const char *)data.data() + st, std::min(pos, n) - st);
            any_comment = true;
            pos++;
        }
        std::string sq;
        c = -1;
        while (pos < n) {
            c = data[pos++];
            if (c == '>' || c == '+' || c == '@') break;
            if (c >= 33 && c <= 126) sq.push_back((char)c);       // isgraph
            c = -1;
        }
        if (c == '>' || c == '@') last_char = c;
        bool ok = true;
        if (c == '+') {
            while (pos < n && data[pos] != '\n') pos++;
            if (pos >= n) { last_ret = -2; break; }
            pos++;
            std::string q;
            // `while ((c = ks_getc(ks)) != -1 && seq->qual.l < seq->seq.l)`: consumes one byte past the quality
            while (pos < n) {
                int ch = data[pos++];
                if (!(q.size() < sq.size())) break;
                if (ch >= 33 && ch <= 127) q.push_back((char)ch);
            }
            last_char = 0;
            if (q.size() != sq.size()) { last_ret = -2; ok = false; }
            else { stale_qual = q; any_qual = true; }
        }
        if (!ok) break;
        name.insert(name.end(), nm.begin(), nm.end()); name_off.push_back(name.size());
        seq.insert(seq.end(), sq.begin(), sq.end()); seq_off.push_back(seq.size());
        // stale-pointer semantics: once allocated, comment.s / qual.s keep their old bytes and
        // searchFile passes them on (libcrispr.cpp:124-131)
        has_c.push_back(any_comment ? 1 : 0);
        if (any_comment) comment.insert(comment.end(), stale_comment.begin(), stale_comment.end());
        comment_off.push_back(comment.size());
        has_q.push_back(any_qual ? 1 : 0);
        if (any_qual) qual.insert(qual.end(), stale_qual.begin(), stale_qual.end());
        qual_off.push_back(qual.size());
        auto it = first_seen.find(nm);
        if (it == first_seen.end()) { first_seen.emplace(nm, nrec); header_id.push_back(nrec); }
        else header_id.push_back(it->second);
        max_len = std::max<uint32_t>(max_len, (uint32_t)sq.size());
        nrec++;
        if (c == -1 && pos >= n) { last_ret = -1; break; }
    }
    auto dup8 = [](const std::vector<uint8_t> &v) { uint8_t *p = (uint8_t *)malloc(v.size() + 1); if (!v.empty()) memcpy(p, v.data(), v.size()); return p; };
    auto dup64 = [](const std::vector<uint64_t> &v) { uint64_t *p = (uint64_t *)malloc((v.size() + 1) * 8); if (!v.empty()) memcpy(p, v.data(), v.size() * 8); return p; };
    out->n_reads = nrec;
    out->seq = dup8(seq); out->seq_off = dup64(seq_off);
    out->name = dup8(name); out->name_off = dup64(name_off);
    out->comment = dup8(comment); out->comment_off = dup64(comment_off); out->has_comment = dup8(has_c);
    out->qual = dup8(qual); out->qual_off = dup64(qual_off); out->has_qual = dup8(has_q);
    out->header_id = dup64(header_id);
    out->max_len = max_len;
    out->last_ret = last_ret;
    return CRASS_OK;
}

void crass_free_fastx(crass_fastx *f)
{
    if (!f) return;
    free(f->seq); free(f->seq_off); free(f->name); free(f->name_off); free(f->comment); free(f->comment_off);
    free(f->has_comment); free(f->qual); free(f->qual_off); free(f->has_qual); free(f->header_id);
    memset(f, 0, sizeof(*f));
}

// ---- synthetic metagenome, counter-based ----
static inline uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static inline uint64_t key3(uint64_t seed, uint64_t a, uint64_t b) { return mix64(seed ^ mix64(a * 0xD1342543DE82EF95ull + mix64(b))); }

void crass_synth_default(crass_synth_spec *s)
{
    s->seed = 42; s->read_len = 150; s->n_dr = 50; s->dr_len_min = 28; s->dr_len_max = 37;
    s->spacer_len_min = 30; s->spacer_len_max = 38; s->crispr_per_million = 10000; s->gc_classes = 0;
}

static inline uint32_t gc_word(uint64_t seed, uint64_t i, uint32_t w, uint32_t cls)
{
    // GC-content classes 30/45/55/70 %: 8-bit threshold draw + 1 bit to pick inside the pair
    static const uint32_t thr[4] = {77, 115, 141, 179};
    uint32_t out = 0;
    for (int q = 0; q < 2; q++) {
        uint64_t r0 = key3(seed ^ 0x6C62272E07BB0142ull, i, (uint64_t)w * 4 + q * 2);
        uint64_t r1 = key3(seed ^ 0x6C62272E07BB0142ull, i, (uint64_t)w * 4 + q * 2 + 1);
        for (int k = 0; k < 8; k++) {
            uint32_t u = (uint32_t)(r0 >> (8 * k)) & 0xFF;
            uint32_t pick = (uint32_t)(r1 >> k) & 1;
            uint32_t code = (u < thr[cls & 3]) ? (pick ? 1u : 2u) : (pick ? 0u : 3u);   // C/G vs A/T
            out |= code << (2 * (q * 8 + k));
        }
    }
    return out;
}

int crass_synth_packed(const crass_synth_spec *s, uint64_t first, uint64_t n, uint32_t *packed, int n_threads)
{
    if (!s || (n && !packed)) return CRASS_ERR_INVALID_ARG;
    if (s->read_len == 0 || s->read_len > 4096 || s->n_dr == 0 || s->dr_len_max < s->dr_len_min ||
        s->spacer_len_max < s->spacer_len_min || s->dr_len_max > 200) return CRASS_ERR_INVALID_ARG;
    const uint32_t L = s->read_len, W = (L + 15) / 16;
    const uint64_t seed = s->seed;
    // DR table
    std::vector<std::vector<uint8_t>> drs(s->n_dr);
    for (uint32_t d = 0; d < s->n_dr; d++) {
        uint32_t len = s->dr_len_min + (uint32_t)(key3(seed, 1000, d) % (s->dr_len_max - s->dr_len_min + 1));
        drs[d].resize(len);
        for (uint32_t k = 0; k < len; k++) drs[d][k] = (uint8_t)((key3(seed, 2000 + d, k >> 5) >> (2 * (k & 31))) & 3);
    }
    unsigned nt = n_threads > 0 ? (unsigned)n_threads : hw_threads();
    nt = std::min<unsigned>(nt, 64);
    parallel_ranges(n, nt, [&](uint64_t a, uint64_t b, unsigned) {
        std::vector<uint8_t> sbuf;
        for (uint64_t k = a; k < b; k++) {
            const uint64_t i = first + k;
            uint32_t *w = packed + k * (uint64_t)W;
            const uint64_t h = key3(seed, 1, i);
            if ((h % 1000000ull) < s->crispr_per_million) {
                const uint32_t d = (uint32_t)((h >> 24) % s->n_dr);
                const uint32_t prefix = (uint32_t)(key3(seed, 3, i) % 41);
                const uint32_t cut = (uint32_t)(key3(seed, 4, i) % 41);
                sbuf.clear();
                for (uint32_t p = 0; p < prefix; p++) sbuf.push_back((uint8_t)((key3(seed, 5, i * 8192 + p) >> 7) & 3));
                uint32_t u = 0;
                while (sbuf.size() < (size_t)cut + L) {
                    sbuf.insert(sbuf.end(), drs[d].begin(), drs[d].end());
                    uint32_t sl = s->spacer_len_min + (uint32_t)(key3(seed, 6, i * 64 + u) % (s->spacer_len_max - s->spacer_len_min + 1));
                    for (uint32_t p = 0; p < sl; p++) sbuf.push_back((uint8_t)((key3(seed, 7, i * 8192 + (uint64_t)(u + 1) * 64 + p) >> 11) & 3));
                    u++;
                }
                for (uint32_t q = 0; q < W; q++) w[q] = 0;
                for (uint32_t q = 0; q < L; q++) w[q >> 4] |= (uint32_t)sbuf[cut + q] << ((q & 15) * 2);
            } else {
                const uint32_t cls = s->gc_classes > 1 ? (uint32_t)((h >> 40) % s->gc_classes) : 0;
                for (uint32_t q = 0; q < W; q++) {
                    uint32_t v = s->gc_classes > 1 ? gc_word(seed, i, q, cls) : (uint32_t)key3(seed ^ 0xA24BAED4963EE407ull, i, q);
                    w[q] = v;
                }
                if (L & 15) w[W - 1] &= (1u << ((L & 15) * 2)) - 1u;       // padding bases are zero
            }
        }
    });
    return CRASS_OK;
}

int crass_unpack_ascii(const uint32_t *packed, uint32_t stride_words, uint32_t read_len, uint64_t n, uint8_t *out)
{
    if (n && (!packed || !out || !stride_words)) return CRASS_ERR_INVALID_ARG;
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    parallel_ranges(n, std::min<unsigned>(hw_threads(), 32), [&](uint64_t a, uint64_t b, unsigned) {
        for (uint64_t i = a; i < b; i++) {
            const uint32_t *w = packed + i * (uint64_t)stride_words;
            uint8_t *o = out + i * (uint64_t)read_len;
            for (uint32_t q = 0; q < read_len; q++) o[q] = (uint8_t)acgt[(w[q >> 4] >> ((q & 15) * 2)) & 3];
        }
    });
    return CRASS_OK;
}

} // extern "C"
