// pgzip.cpp — a gzip member inflated by several threads.
//
// A plain .gz input is ONE deflate stream: libdeflate inflates it at ~0.7 GB/s of text on one thread, which is what bounds a
// gzip'd FASTQ end to end (the rest of a 50 M-read run is a second).  A deflate stream can be entered in the middle if one finds
// the first bit of a block and accepts that the 32 KB of text before it are unknown: bytes copied from that unknown window are
// carried as MARKERS (16-bit symbols >= 0x8000: "byte i of the window") and replaced once the chunk before has been inflated.
// (The technique of pugz / rapidgzip; nothing of theirs is used.)
//
//   1. the stream is cut into chunks; every chunk but the first looks for the first bit position behind its nominal start at
//      which a dynamic-Huffman block header parses completely (code-length code, literal/length and distance codes all complete
//      prefix codes with an end-of-block symbol) and whose block then decodes;
//   2. every chunk decodes from there, block by block, into 16-bit symbols, until it stands exactly on the first bit of a later
//      chunk (or on the final block's end); chunks whose start was never stood on are dropped (their start was no block start);
//   3. the last 32 KB of every kept chunk are resolved in order (a chunk's window is the text before it), then all chunks are
//      resolved and narrowed to bytes side by side;
//   4. the result is accepted only if its length and CRC-32 are the member's trailer's — anything else, and every input this does
//      not take (several members, a short input), returns false and the caller inflates serially.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include <zlib.h>
#include <dlfcn.h>

namespace crass {

namespace {

struct Bits {
    const uint8_t *base, *p, *end;
    uint64_t buf = 0;
    int cnt = 0;
    bool over = false;                                  // read past the end
    void seek(const uint8_t *b, const uint8_t *e, uint64_t bitpos)
    {
        base = b; end = e; p = b + (bitpos >> 3); buf = 0; cnt = 0; over = false;
        refill();
        const int skip = (int)(bitpos & 7u);
        buf >>= skip; cnt -= skip;
    }
    inline void refill()
    {
        if (p + 8 <= end) {                              // eight bytes at once: the bits above cnt are OR-ed in again by the next refill
            uint64_t v;
            memcpy(&v, p, 8);
            buf |= v << cnt;
            p += (63 - cnt) >> 3;
            cnt |= 56;
            return;
        }
        while (cnt <= 56) {
            if (p < end) buf |= (uint64_t)(*p++) << cnt;
            else { if (p >= end + 8) { over = true; } p++; }         // (zeros behind the end; a decode that needs them fails its checks)
            cnt += 8;
        }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1ull)); }
    inline void drop(int n) { buf >>= n; cnt -= n; }
    inline uint32_t get(int n) { const uint32_t v = peek(n); drop(n); return v; }
    uint64_t pos() const { return (uint64_t)(p - base) * 8u - (uint64_t)cnt; }
};

// the symbols of a chunk: a plain growing array (no value-initialisation on growth)
struct SymBuf {
    uint16_t *p = nullptr; size_t n = 0, cap = 0;
    SymBuf() = default;
    SymBuf(const SymBuf &) = delete;
    SymBuf &operator=(const SymBuf &) = delete;
    ~SymBuf() { free(p); }
    bool room(size_t want)                              // at least `want` more elements
    {
        if (n + want <= cap) return true;
        // 2 MB-aligned and MADV_HUGEPAGE where the kernel takes the hint: gigabytes of symbols in 4 KB pages were a quarter of a
        // second of first touches while decoding and another to give them back
        const size_t al = 2u << 20;
        const size_t bytes = ((n + want) * 2 * sizeof(uint16_t) + al - 1) / al * al;
        uint16_t *q = (uint16_t *)aligned_alloc(al, bytes);
        if (!q) return false;
        (void)madvise(q, bytes, MADV_HUGEPAGE);
        if (n) memcpy(q, p, n * sizeof(uint16_t));
        free(p);
        p = q; cap = bytes / sizeof(uint16_t);
        return true;
    }
    size_t size() const { return n; }
};

constexpr int kFast = 11;                               // bits of the primary table
struct Huff {
    uint16_t fast[1 << kFast];                          // (symbol << 4) | length, 0 = a longer code
    uint16_t first[16], count[16], offset[16];          // canonical decoding of the codes longer than kFast bits
    uint16_t syms[320];
    int max_len = 0;
    // false: not a complete prefix code (a single code of length 1 is accepted where deflate allows an incomplete distance code)
    bool build(const uint8_t *lens, int n, bool allow_single)
    {
        int cnt[16] = {0};
        for (int i = 0; i < n; i++) cnt[lens[i]]++;
        cnt[0] = 0;
        int used = 0; max_len = 0;
        for (int l = 1; l < 16; l++) { used += cnt[l]; if (cnt[l]) max_len = l; }
        if (used == 0) return false;
        long left = 1;
        for (int l = 1; l < 16; l++) { left <<= 1; left -= cnt[l]; if (left < 0) return false; }
        if (left != 0 && !(allow_single && used == 1 && cnt[1] == 1)) return false;
        uint16_t next[16]; int code = 0, off = 0;
        for (int l = 1; l < 16; l++) { code = (code + cnt[l - 1]) << 1; first[l] = (uint16_t)code; next[l] = (uint16_t)code; count[l] = (uint16_t)cnt[l]; offset[l] = (uint16_t)off; off += cnt[l]; }
        memset(fast, 0, sizeof(fast));
        uint16_t at[16];
        for (int l = 1; l < 16; l++) at[l] = offset[l];
        for (int s = 0; s < n; s++) {
            const int l = lens[s];
            if (!l) continue;
            syms[at[l]++] = (uint16_t)s;
            const uint32_t c = next[l]++;
            if (l <= kFast) {
                uint32_t rev = 0;
                for (int b = 0; b < l; b++) rev |= ((c >> b) & 1u) << (l - 1 - b);
                for (uint32_t k = rev; k < (1u << kFast); k += 1u << l) fast[k] = (uint16_t)((s << 4) | l);
            }
        }
        return true;
    }
    inline int decode(Bits &b) const                    // -1: no such code
    {
        const uint16_t e = fast[b.peek(kFast)];
        if (e) { b.drop(e & 15); return e >> 4; }
        uint32_t code = 0;
        const uint32_t v = b.peek(15);
        for (int l = 1; l <= max_len; l++) {
            code = (code << 1) | ((v >> (l - 1)) & 1u);
            if (l > kFast && code >= first[l] && code - first[l] < count[l]) { b.drop(l); return syms[offset[l] + code - first[l]]; }
        }
        return -1;
    }
};

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
constexpr uint32_t kWin = 32768;

// a dynamic block's header from the bit behind BTYPE on: the two codes; false = this is not one
bool read_dynamic(Bits &b, Huff &lit, Huff &dist)
{
    b.refill();
    const int hlit = (int)b.get(5) + 257, hdist = (int)b.get(5) + 1, hclen = (int)b.get(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; i++) { b.refill(); cl[kClOrder[i]] = (uint8_t)b.get(3); }
    Huff pre;
    if (!pre.build(cl, 19, false)) return false;
    uint8_t lens[320];
    int n = 0;
    while (n < hlit + hdist) {
        b.refill();
        const int s = pre.decode(b);
        if (s < 0) return false;
        if (s < 16) lens[n++] = (uint8_t)s;
        else {
            int rep, val = 0;
            if (s == 16) { if (!n) return false; val = lens[n - 1]; rep = 3 + (int)b.get(2); }
            else if (s == 17) rep = 3 + (int)b.get(3);
            else rep = 11 + (int)b.get(7);
            if (n + rep > hlit + hdist) return false;
            while (rep--) lens[n++] = (uint8_t)val;
        }
    }
    if (b.over || !lens[256]) return false;
    if (!lit.build(lens, hlit, false)) return false;
    bool any = false;
    for (int i = 0; i < hdist; i++) any |= lens[hlit + i] != 0;
    if (!any) { memset(dist.fast, 0, sizeof(dist.fast)); dist.max_len = 0; for (int l = 0; l < 16; l++) dist.count[l] = 0; return true; }      // (literals only)
    return dist.build(lens + hlit, hdist, true);
}

void fixed_codes(Huff &lit, Huff &dist)
{
    uint8_t l[288];
    for (int i = 0; i < 144; i++) l[i] = 8;
    for (int i = 144; i < 256; i++) l[i] = 9;
    for (int i = 256; i < 280; i++) l[i] = 7;
    for (int i = 280; i < 288; i++) l[i] = 8;
    lit.build(l, 288, false);
    uint8_t d[30];
    for (int i = 0; i < 30; i++) d[i] = 5;
    dist.build(d, 30, true);            // (30 codes of 5 bits: incomplete, as the format has it)
    // build() refuses the incomplete set: fill it by hand
    memset(dist.fast, 0, sizeof(dist.fast));
    for (int s = 0; s < 30; s++) {
        uint32_t rev = 0;
        for (int bb = 0; bb < 5; bb++) rev |= (((uint32_t)s >> bb) & 1u) << (4 - bb);
        for (uint32_t k = rev; k < (1u << kFast); k += 32) dist.fast[k] = (uint16_t)((s << 4) | 5);
    }
    dist.max_len = 5;
}

// the symbols of one block appended to out (out[0 .. kWin) is the virtual window in front of the chunk); 0 ok, 1 ok and final, -1 bad
int decode_block(Bits &b, SymBuf &out, Huff &lit, Huff &dist)
{
    b.refill();
    const uint32_t bfinal = b.get(1), btype = b.get(2);
    if (btype == 3) return -1;
    if (btype == 0) {
        b.drop(b.cnt & 7);                              // to the byte boundary
        b.refill();
        const uint32_t len = b.get(16), nlen = b.get(16);
        if ((len ^ 0xFFFFu) != nlen) return -1;
        if (!out.room(len)) return -1;
        for (uint32_t i = 0; i < len; i++) { b.refill(); if (b.over) return -1; out.p[out.n++] = (uint16_t)b.get(8); }
        return bfinal ? 1 : 0;
    }
    if (btype == 1) fixed_codes(lit, dist);
    else if (!read_dynamic(b, lit, dist)) return -1;
    // (the symbols go through a raw cursor into the vector's storage, grown a block of room at a time)
    size_t sz = out.n;
    if (!out.room(1u << 16)) return -1;
    uint16_t *base = out.p;
    size_t lim = out.cap - 300;
    int rc = 0;
    for (;;) {
        if (sz >= lim) { out.n = sz; if (!out.room(1u << 16)) { rc = -1; break; } base = out.p; lim = out.cap - 300; }
        b.refill();
        int s = lit.decode(b);
        if (s < 256) {
            if (s < 0) { rc = -1; break; }
            base[sz++] = (uint16_t)s;
            // (a second literal from the same refill: 57 bits cover two codes)
            s = lit.decode(b);
            if (s < 256) { if (s < 0) { rc = -1; break; } base[sz++] = (uint16_t)s; continue; }
        }
        if (s == 256) break;
        s -= 257;
        if (s >= 29) { rc = -1; break; }
        const uint32_t len = kLenBase[s] + b.get(kLenExtra[s]);
        b.refill();
        const int d = dist.max_len ? dist.decode(b) : -1;
        if (d < 0 || d >= 30) { rc = -1; break; }
        const uint32_t dd = kDistBase[d] + b.get(kDistExtra[d]);
        if (dd > sz) { rc = -1; break; }
        uint16_t *o = base + sz;
        const uint16_t *src = o - dd;
        if (dd >= len) memcpy(o, src, (size_t)len * 2);
        else for (uint32_t i = 0; i < len; i++) o[i] = src[i];
        sz += len;
    }
    if (b.over) rc = -1;
    out.n = sz;
    if (rc < 0) return -1;
    return bfinal ? 1 : 0;
}

struct Chunk {
    uint64_t nominal = 0;                               // bit position the search starts from
    std::atomic<uint64_t> start{~0ull};                 // first bit of the block it decodes from (~0: none found); published by its worker
    std::atomic<int> ready{0};
    SymBuf sym;                                         // [kWin virtual window | the text]
    uint64_t end = 0;                                   // bit position it stopped at
    int link = -1;                                      // the chunk that starts there (-2: the stream's end)
    bool bad = false;
};

} // namespace

// the gzip member in[0, n) -> malloc'd text; false: not taken (the caller inflates serially)
bool parallel_gunzip(const uint8_t *in, size_t n, uint8_t **out_p, size_t *out_n, unsigned threads)
{
    auto no = [](const char *why) { if (getenv("CRASS_TIMING")) fprintf(stderr, "[crass_timing] inflate: the several-thread path gives up: %s\n", why); return false; };
    // (tests: CRASS_PGZIP_MIN_BYTES / CRASS_PGZIP_CHUNK_BYTES bring the thresholds down to files of a megabyte and chunks shorter than
    // a window)
    const size_t min_bytes = getenv("CRASS_PGZIP_MIN_BYTES") ? (size_t)std::max(64ll, atoll(getenv("CRASS_PGZIP_MIN_BYTES"))) : (size_t)(32u << 20);
    const size_t chunk_bytes = getenv("CRASS_PGZIP_CHUNK_BYTES") ? (size_t)std::max(4096ll, atoll(getenv("CRASS_PGZIP_CHUNK_BYTES"))) : (size_t)(4u << 20);
    if (n < min_bytes || threads < 2) return false;
    if (in[0] != 0x1f || in[1] != 0x8b || in[2] != 8) return false;
    const uint8_t flg = in[3];
    size_t h = 10;
    if (flg & 4) { if (h + 2 > n) return false; h += 2 + ((size_t)in[h] | ((size_t)in[h + 1] << 8)); }
    if (flg & 8) { while (h < n && in[h]) h++; h++; }
    if (flg & 16) { while (h < n && in[h]) h++; h++; }
    if (flg & 2) h += 2;
    if (h + 8 >= n) return false;
    const uint8_t *d = in + h;
    const size_t dn = n - h - 8;                        // deflate data, if this is the only member
    uint32_t want_crc, want_size;
    memcpy(&want_crc, in + n - 8, 4); memcpy(&want_size, in + n - 4, 4);
    const unsigned nc = (unsigned)std::min<size_t>(getenv("CRASS_PGZIP_CHUNK_BYTES") ? 4096u : threads * 3u, dn / chunk_bytes);
    if (nc < 2) return false;
    std::unique_ptr<Chunk[]> ch(new Chunk[nc]);
    for (unsigned k = 0; k < nc; k++) ch[k].nominal = (uint64_t)(dn * (uint64_t)k / nc) * 8u;
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now_s();
    std::atomic<unsigned> next{0};
    std::atomic<bool> fail{false};
    auto work = [&]() {
        std::unique_ptr<Huff> lit(new Huff()), dist(new Huff());
        for (;;) {
            const unsigned k = next.fetch_add(1);
            if (k >= nc || fail.load()) break;
            Chunk &c = ch[k];
            Bits b;
            c.sym.n = 0;
            if (!c.sym.room(kWin + (size_t)(dn / nc) * 5)) { c.bad = true; c.start.store(~0ull); c.ready.store(1); fail.store(true); continue; }
            c.sym.n = kWin;
            for (uint32_t i = 0; i < kWin; i++) c.sym.p[i] = (uint16_t)(0x8000u | i);
            uint64_t start = ~0ull;
            if (k == 0) start = 0;
            else {
                // the first bit position from the nominal start on where BFINAL = 0, BTYPE = 2, the header parses and the block decodes
                const uint64_t limit = k + 1 < nc ? ch[k + 1].nominal : (uint64_t)dn * 8u;
                for (uint64_t p = c.nominal; p + 64 < limit; p++) {
                    const uint64_t w = ((uint64_t)d[p >> 3] | ((uint64_t)d[(p >> 3) + 1] << 8) | ((uint64_t)d[(p >> 3) + 2] << 16) | ((uint64_t)d[(p >> 3) + 3] << 24)) >> (p & 7u);
                    if ((w & 7u) != 4u) continue;                       // bits: 0 (not final), then 0 1 (BTYPE 2, low bit first)
                    if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;
                    b.seek(d, d + dn, p + 3);
                    if (!read_dynamic(b, *lit, *dist)) continue;
                    b.seek(d, d + dn, p);
                    c.sym.n = kWin;
                    if (decode_block(b, c.sym, *lit, *dist) != 0) continue;
                    // ... and what follows must be a block too (a header that parses by chance is rare, two in a row are not seen)
                    const uint64_t p2 = b.pos();
                    const size_t keep = c.sym.size();
                    const int r2 = decode_block(b, c.sym, *lit, *dist);
                    if (r2 < 0) continue;
                    c.sym.n = keep;
                    b.seek(d, d + dn, p2);
                    start = p;
                    break;
                }
            }
            c.start.store(start);
            c.ready.store(1);
            if (start == ~0ull) continue;                               // (no block start in its range: the chunk before decodes through it)
            if (k == 0) b.seek(d, d + dn, 0);
            // block by block until it stands on a later chunk's first bit, or the stream ends
            for (;;) {
                const uint64_t pos = b.pos();
                int hit = -1;
                bool wait_more = false;
                for (unsigned j = k + 1; j < nc; j++) {
                    if (ch[j].nominal > pos) { if (!ch[j].ready.load()) wait_more = false; break; }     // (a chunk whose search starts behind pos cannot start at pos)
                    // A chunk that no thread has fetched yet is never waited for: if every thread stood behind such a chunk (a
                    // crafted stream whose in-flight chunks all began on false block starts) nobody would ever fetch it.  This
                    // thread decodes on through its range instead; should the chunk turn out to start here after all, it is
                    // decoded twice and left out of the chain.
                    if (!ch[j].ready.load() && j >= next.load()) break;
                    while (!ch[j].ready.load() && !fail.load()) std::this_thread::yield();
                    if (ch[j].start.load() == pos) { hit = (int)j; break; }
                }
                (void)wait_more;
                if (hit >= 0 && c.sym.size() > kWin) { c.end = pos; c.link = hit; break; }
                const int r = decode_block(b, c.sym, *lit, *dist);
                if (r < 0) { c.bad = true; c.end = pos; break; }
                if (r == 1) { c.end = b.pos(); c.link = -2; break; }
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < threads; t++) th.emplace_back(work);
        work();
        for (auto &x : th) x.join();
    }
    const double t_dec = now_s();
    // the chain of chunks from the first one
    std::vector<unsigned> chain;
    for (int k = 0; k >= 0;) {
        if (ch[k].bad || ch[k].start.load() == ~0ull) return no("a chunk of the chain did not decode");
        chain.push_back((unsigned)k);
        if (ch[k].link == -2) break;
        if (ch[k].link < 0 || chain.size() > nc) return no("the chain is broken");
        k = ch[k].link;
    }
    const Chunk &lastc = ch[chain.back()];
    if (((lastc.end + 7) >> 3) != dn) return no("the member ends before the input does (more members?)");       // (the serial path's business)
    size_t total = 0;
    std::vector<size_t> at(chain.size() + 1, 0);
    for (size_t i = 0; i < chain.size(); i++) { at[i] = total; total += ch[chain[i]].sym.size() - kWin; }
    at[chain.size()] = total;
    if ((uint32_t)total != want_size) return no("length differs from the trailer's");
    uint8_t *out;
    {
        const size_t al = 2u << 20, bytes = (total + 1 + al - 1) / al * al;
        out = (uint8_t *)aligned_alloc(al, bytes);
        if (!out) return false;
        (void)madvise(out, bytes, MADV_HUGEPAGE);
    }
    // the windows, in order: win[i] = the 32 KB of text in front of chain[i] (index kWin - 1 = the byte right before it)
    std::vector<std::vector<uint8_t>> win(chain.size());
    bool ok = true;
    for (size_t i = 0; i < chain.size() && ok; i++) {
        win[i].assign(kWin, 0);
        if (i == 0) continue;
        const Chunk &p = ch[chain[i - 1]];
        const size_t pn = p.sym.size() - kWin;
        const std::vector<uint8_t> &pw = win[i - 1];
        for (uint32_t q = 0; q < kWin; q++) {
            // byte q of this window = text position (end of previous chunk) - kWin + q
            if ((size_t)(kWin - q) <= pn) {
                const uint16_t v = p.sym.p[p.sym.size() - (kWin - q)];
                if (v & 0x8000u) { if (i - 1 == 0) { ok = false; break; } win[i][q] = pw[v & 0x7FFFu]; } else win[i][q] = (uint8_t)v;
            } else win[i][q] = pw[q + pn];                              // (the previous chunk is shorter than a window: older text)
        }
    }
    if (ok) {
        std::atomic<size_t> nx{0};
        std::atomic<bool> bad{false};
        auto fill = [&]() {
            for (;;) {
                const size_t i = nx.fetch_add(1);
                if (i >= chain.size()) break;
                const Chunk &c = ch[chain[i]];
                const uint16_t *s = c.sym.p + kWin;
                const size_t m = c.sym.size() - kWin;
                uint8_t *o = out + at[i];
                const uint8_t *w = win[i].data();
                if (i == 0) { for (size_t q = 0; q < m; q++) { if (s[q] & 0x8000u) { bad.store(true); break; } o[q] = (uint8_t)s[q]; } }
                else for (size_t q = 0; q < m; q++) { const uint16_t v = s[q]; o[q] = (v & 0x8000u) ? w[v & 0x7FFFu] : (uint8_t)v; }
                // (the symbols go back to the allocator here, side by side: 3 GB of them freed by the caller's thread were 0.3 s)
                Chunk &cm = ch[chain[i]];
                free(cm.sym.p); cm.sym.p = nullptr; cm.sym.n = cm.sym.cap = 0;
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < threads; t++) th.emplace_back(fill);
        fill();
        for (auto &x : th) x.join();
        ok = !bad.load();
    }
    const double t_fill = now_s();
    if (ok) {
        // CRC-32 of the text, pieces side by side
        const unsigned np = (unsigned)std::min<size_t>(threads, total / (8u << 20) + 1);
        std::vector<uLong> crcs(np, 0);
        std::vector<size_t> cut(np + 1);
        for (unsigned t = 0; t <= np; t++) cut[t] = total * t / np;
        std::vector<std::thread> th;
        // (libdeflate's CRC-32 where the library is there — carry-less multiplies, ~10 GB/s per core; zlib's table walk is ~1 GB/s,
        // 0.35 s of an 8 GB input on sixteen threads.  The same polynomial: the pieces still combine through crc32_combine)
        typedef uint32_t (*crc_fn)(uint32_t, const void *, size_t);
        static const crc_fn ld_crc = [] {
            if (getenv("CRASS_NO_LIBDEFLATE")) return (crc_fn) nullptr;
            void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
            return h ? (crc_fn)dlsym(h, "libdeflate_crc32") : (crc_fn) nullptr;
        }();
        auto one = [&](unsigned t) {
            if (ld_crc) { crcs[t] = ld_crc(0, out + cut[t], cut[t + 1] - cut[t]); return; }
            uLong c = crc32(0L, Z_NULL, 0);
            for (size_t a = cut[t]; a < cut[t + 1];) { const size_t m = std::min<size_t>(cut[t + 1] - a, 1u << 30); c = crc32(c, out + a, (uInt)m); a += m; }
            crcs[t] = c;
        };
        for (unsigned t = 1; t < np; t++) th.emplace_back(one, t);
        one(0);
        for (auto &x : th) x.join();
        uLong c = crcs[0];
        for (unsigned t = 1; t < np; t++) c = crc32_combine(c, crcs[t], (z_off_t)(cut[t + 1] - cut[t]));
        ok = (uint32_t)c == want_crc;
    }
    if (!ok) { free(out); return no("unresolved markers or a CRC-32 that differs from the trailer's"); }
    if (getenv("CRASS_TIMING"))
        fprintf(stderr, "[crass_timing] inflate: %zu chunks in the chain of %u cut, %u threads: find + decode %.3f s, windows + narrowing %.3f s, CRC-32 %.3f s\n",
                chain.size(), nc, threads, t_dec - t_begin, t_fill - t_dec, now_s() - t_fill);
    *out_p = out; *out_n = total;
    return true;
}

} // namespace crass
