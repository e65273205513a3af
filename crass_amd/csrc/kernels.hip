// kernels.hip — hand-written HIP kernels for gfx950 (MI355X, CDNA4, wave64).
//
// The hot path of crass's WorkHorse search (reference citations are relative to the
// crass v1.0.1 tree):
//   pass 1  searchCore / scanRight / extendPreRepeat / qcFoundRepeats / DRLowLexi
//           (src/crass/libcrispr.cpp:170-395,520-1069, ReadHolder.cpp:513-609,
//            PatternMatcher.cpp:26-204)
//   pass 2  findSingletons / on_match over an Aho-Corasick automaton
//           (src/crass/libcrispr.cpp:399-518, src/aho-corasick/acism.c:25-106)
//
// Integer/byte work, HBM- and issue-bound: no MFMA.  Reads are 2-bit packed and streamed
// once per pass; wave64 ballot/ffs gives Boyer-Moore's "leftmost occurrence" for free;
// the pass-2 automaton is staged in LDS when it fits.  Compile with -ffp-contract=off:
// qcFoundRepeats' float evaluation order is part of the parity contract.
#include "engine_internal.h"
#include <type_traits>
#include <algorithm>

namespace crass {

#define WAVE 64

__constant__ unsigned char c_comp[128];     // reverseComplement table, SeqUtils.cpp:50-59

static __device__ __forceinline__ void wave_sync()
{
    // LDS traffic of one wave is executed in order; this only stops the compiler from
    // moving LDS accesses across the point where lanes exchange data through LDS.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A value that is the same in every lane but was read from LDS (or computed from such a read) lives in a VGPR as far as the
// compiler knows, and everything derived from it — loop counters, branch conditions — becomes vector arithmetic under exec
// masks.  The wave-per-read kernel's control flow is wave-uniform throughout: naming the value once puts it, and what follows
// from it, on the scalar unit.
static __device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
static __device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
static __device__ __forceinline__ uint64_t uni64(uint64_t x) { return ((uint64_t)uni((uint32_t)(x >> 32)) << 32) | (uint64_t)uni((uint32_t)x); }

static __device__ __forceinline__ uint64_t rd_word_off(const DevReads &R, uint64_t r)
{
    return R.stride_words ? r * (uint64_t)R.stride_words : R.word_off[r];
}
static __device__ __forceinline__ uint32_t rd_len(const DevReads &R, uint64_t r)
{
    return R.uniform_len ? R.uniform_len : R.lengths[r];
}
static __device__ __forceinline__ bool rd_is_exc(const DevReads &R, uint64_t r)
{
    return (R.exc_mask[r >> 5] >> (r & 31)) & 1u;
}
// first word of read r's position hints: reads of one length need no table look-up (a dependent global load per read in the
// wave kernel's prefetch otherwise)
static __device__ __forceinline__ uint64_t rd_hint_off(const DevReads &R, uint64_t r)
{
    return R.uniform_len ? r * (uint64_t)((R.uniform_len + 63u) >> 6) : R.pos_hint_off[r];
}
static __device__ __forceinline__ uint64_t rd_header_id(const DevReads &R, uint64_t r)
{
    return R.header_id ? R.header_id[r] : r;
}

// ------------------------------------------------------------------------------------
// exception bit mask
// ------------------------------------------------------------------------------------
__global__ void k_build_exc_mask(const uint64_t *exc_read, uint64_t n_exc, uint32_t *exc_mask)
{
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n_exc) {
        uint64_t r = exc_read[i];
        atomicOr(&exc_mask[r >> 5], 1u << (r & 31));
    }
}

__global__ void k_mark_found(const uint64_t *idx, uint64_t n, const uint64_t *header_id, uint8_t *found_flag)
{
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) { const uint64_t r = idx[i]; found_flag[header_id ? header_id[r] : r] = 1; }
}
hipError_t launch_mark_found(const uint64_t *idx, uint64_t n, const uint64_t *header_id, uint8_t *found_flag, hipStream_t st)
{
    if (!n) return hipSuccess;
    CRASS_LAUNCH(k_mark_found, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, idx, n, header_id, found_flag);
    return hipGetLastError();
}

// Device -> pinned host copy by a handful of workgroups (the link is the bound: ~55 GB/s needs a few hundred stores in
// flight, not a chip): the fall-back of the DMA-engine copy (sdma.cpp).  Like the runtime's blit kernel it costs the kernels
// of the other stream its own duration — PCIe stores from shader waves do, however few waves issue them and on however
// many XCDs (profiles/NOTES_r03.md §9) — so the engine orders it behind the merge kernels.
__global__ __launch_bounds__(256) void k_copy_to_host(const uint4 *src, uint4 *dst, uint64_t n16, const uint8_t *src_tail, uint8_t *dst_tail, uint32_t n_tail)
{
    const uint64_t nth = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += nth) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x < n_tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
hipError_t launch_copy_to_host(const void *d_src, void *h_dst, uint64_t bytes, hipStream_t st)
{
    if (!bytes) return hipSuccess;
    if ((reinterpret_cast<uintptr_t>(d_src) | reinterpret_cast<uintptr_t>(h_dst)) & 15u) return hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st);
    const uint64_t n16 = bytes / 16;
    const unsigned blocks = (unsigned)std::min<uint64_t>(32, (n16 + 255) / 256 + 1);
    CRASS_LAUNCH(k_copy_to_host, dim3(blocks), dim3(256), 0, st, static_cast<const uint4 *>(d_src), static_cast<uint4 *>(h_dst), n16,
                 static_cast<const uint8_t *>(d_src) + n16 * 16, static_cast<uint8_t *>(h_dst) + n16 * 16, (uint32_t)(bytes & 15u));
    return hipGetLastError();
}

hipError_t launch_build_exc_mask(const uint64_t *exc_read, uint64_t n_exc, uint32_t *exc_mask, hipStream_t st)
{
    if (!n_exc) return hipSuccess;
    CRASS_LAUNCH(k_build_exc_mask, dim3((unsigned)((n_exc + 255) / 256)), dim3(256), 0, st, exc_read, n_exc, exc_mask);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// pass 1, step 1: seed-scan filter (general parameters, any read length)
//
// Contract: bit r of hitmask is set if searchCore's seed loop (libcrispr.cpp:295-348) finds a
// w-mer hit for SOME seed j on the stride-`skips` lattice.  searchCore leaves the lattice only
// after a hit (:390), so a clear bit proves searchCore returns false.  A superset is allowed
// (survivors are re-evaluated exactly by k_survivor).
// One wave scans 64 consecutive reads (one mask word); per read the packed words are staged in
// LDS, lanes span the <=64 candidate positions of a seed window and a ballot says "found".
// ------------------------------------------------------------------------------------
#define FG_WAVES 4

static __device__ __forceinline__ uint32_t lds_code(const uint32_t *w, uint32_t p, uint32_t mask)
{
    uint32_t wi = p >> 4, sh = (p & 15u) * 2u;
    uint64_t v = ((uint64_t)w[wi + 1] << 32) | w[wi];
    return (uint32_t)(v >> sh) & mask;
}

__global__ __launch_bounds__(FG_WAVES * WAVE) void k_filter_general(DevReads R, DevParams P, uint64_t *hitmask,
                                                                      uint32_t lds_words_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t fg_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    uint32_t *words = fg_lds + (size_t)wave * lds_words_per_wave;
    const uint64_t n_words = (R.n_reads + 63) / 64;
    const uint32_t w = P.window;
    const uint32_t cmask = (1u << (2 * w)) - 1u;
    for (uint64_t mw = blockIdx.x * (uint64_t)FG_WAVES + wave; mw < n_words; mw += (uint64_t)gridDim.x * FG_WAVES) {
        uint64_t bits = 0;
        for (int k = 0; k < 64; k++) {
            uint64_t r = mw * 64 + k;
            if (r >= R.n_reads) break;
            // exc_survive: exception reads are screened on their packed words like every other read (a non-ACGT byte
            // packs as 'A', so byte-equal seeds are code-equal: still a superset) and evaluated byte-wise afterwards
            if (!P.exc_survive && rd_is_exc(R, r)) continue;
            uint32_t L = rd_len(R, r);
            int searchEnd = (int)(L - P.lowDR - P.lowSp - w - 1);
            if (searchEnd < 0) continue;
            uint32_t nw = (L + 15) >> 4;
            const uint32_t *g = R.packed + rd_word_off(R, r);
            wave_sync();
            for (uint32_t i = lane; i < nw + 1; i += 64) words[i] = (i < nw) ? g[i] : 0u;
            wave_sync();
            bool hit = false;
            for (uint32_t j = 0; j <= (uint32_t)searchEnd && !hit; j += P.skips) {
                uint32_t begin = j + P.lowDR + P.lowSp;
                uint32_t end = j + P.highDR + P.highSp + w;
                if (end >= L) end = L - 1;
                if (end < begin) end = begin;
                uint32_t sj = lds_code(words, j, cmask);
                for (uint32_t p0 = begin; p0 + w <= end; p0 += 64) {
                    uint32_t p = p0 + lane;
                    bool ok = (p + w <= end) && (lds_code(words, p, cmask) == sj);
                    if (__ballot(ok)) { hit = true; break; }
                }
            }
            if (hit) bits |= (1ull << k);
        }
        if (lane == 0) hitmask[mw] = bits;
    }
}

hipError_t launch_filter_general(const DevReads &R, const DevParams &P, uint64_t *hitmask, uint32_t max_len, hipStream_t st)
{
    uint32_t wpw = ((max_len + 15) / 16 + 2 + 3) & ~3u;
    size_t lds = (size_t)FG_WAVES * wpw * 4;
    uint64_t n_words = (R.n_reads + 63) / 64;
    uint64_t blocks = (n_words + FG_WAVES - 1) / FG_WAVES;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks == 0) return hipSuccess;
    CRASS_LAUNCH(k_filter_general, dim3((unsigned)blocks), dim3(FG_WAVES * WAVE), lds, st, R, P, hitmask, wpw);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// pass 1, step 1 (fast path): lane-per-read bit-parallel seed scan.
// Requirements: window == 8 and skips == 8 (defaults: every lattice seed is exactly one
// aligned halfword of the packed read), uniform stride <= 16 words (reads <= 256 bp).
// For a shift d, X = R ^ (R >> 2d) has a zero halfword h  <=>  the 8-mer at seed j=8h
// re-occurs at j+d.  v_pk_min_u16 accumulates "any zero so far" per halfword over all
// shifts d in [lowDR+lowSp, highDR+highSp]; a final per-halfword test yields the hit bit.
// The window's right clamp (libcrispr.cpp:301-304) and bases in the padding are ignored:
// that can only ADD survivors (superset contract), never lose one.
// ------------------------------------------------------------------------------------
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));   // v_pk_min_u16
    return __builtin_bit_cast(uint32_t, r);
}

// Fully unrolled implementation: D0..D1 are compile-time so every register index is static.
// LCT: compile-time read length (0 = unknown).  With it, shifts that would put the window past the
// read end (libcrispr.cpp:301-304: p <= L-9-w... i.e. d <= L-9-16k for the seeds of word k) are dropped
// at compile time: 234 instead of 343 (word, shift) pairs at L = 150.
// min(halfword, 1) for both halfwords: 0 where the halfword is 0, else 1.  Through the builtin the optimiser turns this into
// two compares, two selects and a permute; the instruction itself is what is wanted.
static __device__ __forceinline__ uint32_t pk_nonzero_u16(uint32_t a)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(0x00010001u));
    return r;
}
typedef uint32_t ff_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t ff_u32x2 __attribute__((ext_vector_type(2), aligned(4)));
// One read's row (W words, zero when r is past the end) and its exception flag.
template <int W>
static __device__ __forceinline__ void ff_load_row(const DevReads &R, const DevParams &P, uint64_t r, uint32_t (&w)[W], uint64_t &exc_word)
{
#pragma unroll
    for (int i = 0; i < W; i++) w[i] = 0;
    // the wave's 64 exception bits: one scalar load (the mask has a spare word, engine.cpp load_reads), off the vector memory
    // counter so that waiting for it does not wait for the row
    exc_word = 0;
    const uint64_t wave_word = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(r >> 38)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)(r >> 6));
    if (!P.exc_survive && (wave_word << 6) < R.n_reads) {
        const uint32_t *e = R.exc_mask + 2 * wave_word;
        exc_word = (uint64_t)e[0] | ((uint64_t)e[1] << 32);
    }
    if (r < R.n_reads) {
        const uint32_t *g = R.packed + r * (uint64_t)W;
        // a row is dword-aligned only; global_load_dwordx4 takes that on gfx950, so W words are W/4 wide loads and a tail
#pragma unroll
        for (int i = 0; i + 4 <= W; i += 4) {
            const ff_u32x4 v = *reinterpret_cast<const ff_u32x4 *>(g + i);
            w[i] = v.x; w[i + 1] = v.y; w[i + 2] = v.z; w[i + 3] = v.w;
        }
        if ((W & 3) >= 2) {
            const ff_u32x2 v = *reinterpret_cast<const ff_u32x2 *>(g + (W & ~3));
            w[W & ~3] = v.x; w[(W & ~3) + 1] = v.y;
        }
        if (W & 1) w[W - 1] = g[W - 1];
    }
}

// RPL: reads per lane.  A lane's reads are 256 apart (a wave still covers 64 consecutive reads, one mask word); the row of the
// next read is loaded before the current one is scanned, so a wave's only exposed memory latency is its first load.
template <int W, int D0, int D1, int LCT, int RPL>
__global__ __launch_bounds__(256) void k_filter_fast_impl(DevReads R, DevParams P, uint64_t *hitmask, uint32_t *seed_hint, uint8_t *clear_found)
{
    const uint64_t r_first = blockIdx.x * (uint64_t)(256 * RPL) + threadIdx.x;
    // the step's found flags (one byte per header id, n + 1 of them) are cleared on the way: 64 consecutive bytes per wave and row
    // beside 2.5 KB of loads, instead of a 12-100 MB fill kernel in front of every seed scan (6-30 us of the step)
    if (clear_found && r_first == 0) clear_found[R.n_reads] = 0;
    constexpr int WX = W + (D1 >> 4) + 2;
    uint32_t nxt[W];
    uint64_t nxt_exc;
    ff_load_row<W>(R, P, r_first, nxt, nxt_exc);
#pragma unroll
    for (int it = 0; it < RPL; it++) {
    const uint64_t r = r_first + (uint64_t)it * 256u;
    const bool active = r < R.n_reads;
    if (clear_found && active) clear_found[r] = 0;
    uint32_t w[WX];
#pragma unroll
    for (int i = 0; i < WX; i++) w[i] = i < W ? nxt[i] : 0u;
    const bool exc = (nxt_exc >> (r & 63)) & 1u;        // (exception reads are left to their own pass, see k_filter_general)
    if (it + 1 < RPL) ff_load_row<W>(R, P, r + 256u, nxt, nxt_exc);
    const uint32_t L = active ? rd_len(R, r) : 0u;
    // seeds live in halfwords 0 .. searchEnd/8 with searchEnd = L-58 <= 16W-58: only words < SW hold one
    constexpr int SW = ((16 * W - 58) / 8 + 2) / 2;
    uint32_t acc[SW];
#pragma unroll
    for (int i = 0; i < SW; i++) acc[i] = 0xFFFFFFFFu;
#pragma unroll
    for (int d = D0; d <= D1; d++) {
        const int q = d >> 4;
        const int sh = (d & 15) * 2;
#pragma unroll
        for (int k = 0; k < SW; k++) {
            if (LCT > 0 && d > LCT - 9 - 16 * k) continue;               // window past the read end for both seeds of word k
            uint32_t lo = w[k + q], hi = w[k + q + 1];
            uint32_t s = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;      // v_alignbit_b32
            uint32_t x = s ^ w[k];
            // per-halfword running minimum: a halfword of acc becomes 0 iff some x halfword was 0
            acc[k] = pk_min_u16(acc[k], x);
        }
    }
    bool hit = false;
    // a uniform-length instantiation knows the seed count at compile time (D0 = lowDR + lowSp, checked by the launcher)
    const int searchEnd = LCT > 0 ? (LCT - D0 - 8 - 1) : (int)(L - P.lowDR - P.lowSp - 8 - 1);
    if (active && !exc && searchEnd >= 0) {
        const int n_seed = searchEnd / 8 + 1;           // halfwords 0 .. n_seed-1 hold lattice seeds
        // bit h of the hint: lattice seed j = 8h may have a hit (superset).  min(halfword, 1) is 0 exactly for a hit; the words
        // are folded two bits apart (low halfwords -> bits 2k, high -> bits 2k + 16) and the halves interleaved at the end
        uint32_t t0 = 0, t1 = 0;                        // words 0..7 and 8..15 (a fold holds 16 seeds)
#pragma unroll
        for (int k = 0; k < SW; k++) {
            if (LCT > 0 && 2 * k >= n_seed) continue;
            const uint32_t f = pk_nonzero_u16(acc[k]);
            if (k < 8) t0 |= f << (2 * k); else t1 |= f << (2 * (k - 8));
        }
        uint32_t miss = (t0 | (t0 >> 15)) & 0xFFFFu;
        if (SW > 8) miss |= (t1 | (t1 >> 15)) << 16;
        const uint32_t hint = ~miss & (n_seed >= 32 ? 0xFFFFFFFFu : ((1u << n_seed) - 1u));
        hit = hint != 0;
        if (hit) seed_hint[r] = hint;                   // sparse: ~2 % of the lanes
    }
    uint64_t m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && active) hitmask[r >> 6] = m;
    }
}

// The same scan for ANY shift range (-s / -S / -D: the seed lattice and the window are the defaults', only the distances at
// which a copy counts change): D0 = lowDR + lowSp .. D1 = highDR + highSp arrive at run time.  The fixed-range kernel indexes its
// registers with compile-time shifts; here the WORD part of a shift (d >> 4) stays a compile-time loop — static register
// indices —, the sixteen bit offsets inside it are sixteen copies of the three instructions behind one scalar branch each, and a
// funnel shift takes its amount from a register as readily as from an immediate: the instructions executed are the fixed
// kernel's for the same range.  (k_filter_general, which every non-default option set took until round 5, is a wave per 64
// reads through LDS: 10.6 ms for 10 M reads with -s 20 -S 60 against 0.18 ms for the defaults.)
template <int W>
__global__ __launch_bounds__(256) void k_filter_fast_range(DevReads R, DevParams P, uint64_t *hitmask, uint32_t *seed_hint)
{
    const uint64_t r = blockIdx.x * (uint64_t)256 + threadIdx.x;
    constexpr int WX = 2 * W + 2;                       // shifts up to 16 W + 15 bases: beyond a read of this stride nothing matches
    uint32_t row[W];
    uint64_t exc_word;
    ff_load_row<W>(R, P, r, row, exc_word);
    const bool active = r < R.n_reads;
    uint32_t w[WX];
#pragma unroll
    for (int i = 0; i < WX; i++) w[i] = i < W ? row[i] : 0u;
    const bool exc = (exc_word >> (r & 63)) & 1u;
    const uint32_t L = active ? rd_len(R, r) : 0u;
    constexpr int SW = ((16 * W - 26) / 8 + 2) / 2;     // words that can hold a lattice seed for ANY bounds (searchEnd <= 16 W - 26: lowDR + lowSp >= 17)
    const int D0 = (int)(P.lowDR + P.lowSp), D1 = min((int)(P.highDR + P.highSp), 16 * W + 15);
    uint32_t acc[SW];
#pragma unroll
    for (int i = 0; i < SW; i++) acc[i] = 0xFFFFFFFFu;
    const int n_seed_max = (16 * W - D0 - 9) / 8 + 1;                   // no read of this stride has a seed beyond (wave-uniform)
#pragma unroll
    for (int q = 0; q <= W; q++) {
        if (16 * q + 15 < D0 || 16 * q > D1) continue;                  // (wave-uniform: a scalar branch)
        const int sb_lo = max(0, D0 - 16 * q), sb_hi = min(15, D1 - 16 * q);
        for (int sb = sb_lo; sb <= sb_hi; sb++) {                       // (a run-time loop: the bodies below are q x k, not q x 16 x k)
            const uint32_t sh = (uint32_t)(2 * sb);
#pragma unroll
            for (int k = 0; k < SW; k++) {
                if (2 * k >= n_seed_max) continue;
                const uint32_t lo = w[k + q], hi = w[k + q + 1];
                acc[k] = pk_min_u16(acc[k], __builtin_amdgcn_alignbit(hi, lo, sh) ^ w[k]);
            }
        }
    }
    bool hit = false;
    const int searchEnd = (int)(L - P.lowDR - P.lowSp - 8 - 1);
    if (active && !exc && searchEnd >= 0) {
        const int n_seed = searchEnd / 8 + 1;
        uint32_t t0 = 0, t1 = 0;
#pragma unroll
        for (int k = 0; k < SW; k++) {
            const uint32_t f = pk_nonzero_u16(acc[k]);
            if (k < 8) t0 |= f << (2 * k); else t1 |= f << (2 * (k - 8));
        }
        uint32_t miss = (t0 | (t0 >> 15)) & 0xFFFFu;
        if (SW > 8) miss |= (t1 | (t1 >> 15)) << 16;
        const uint32_t hint = ~miss & (n_seed >= 32 ? 0xFFFFFFFFu : ((1u << n_seed) - 1u));
        hit = hint != 0;
        if (hit) seed_hint[r] = hint;
    }
    const uint64_t m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && active) hitmask[r >> 6] = m;
}

// ... and for ANY window (6 .. 9) and seed lattice (-w, -d: skips = lowDR - (2 w - 1) is then no multiple of a halfword's eight
// bases, and a w-mer no halfword).  For a shift d, X = R xor (R >> 2d) has 2 w zero bits from bit 2p  <=>  the w-mer at p
// re-occurs at p + d: the mismatch flag of a base is the OR of its two bits, a window's flag the OR of its w bases' flags — built by
// doubling (w = 8: windows of 1, 2, 4, 8 bases; three funnel-shift + OR pairs) —, and the flags are AND-ed over the shifts: a zero
// at bit 2p of the result means "some shift matches at p".  11 instructions per word and shift instead of the halfword form's 3,
// for EVERY position at once; the lattice is a mask at the end.  (k_filter_general took 7.3 ms for 10 M reads with -w 7 and 13.7 ms
// with -d 20 -D 40, against 0.18 ms for the defaults: profiles/r05_bench_c1_params_*.json.)
template <int W>
__global__ __launch_bounds__(256) void k_filter_fast_any(DevReads R, DevParams P, uint64_t *hitmask, uint32_t *seed_hint)
{
    const uint64_t r = blockIdx.x * (uint64_t)256 + threadIdx.x;
    constexpr int WX = 2 * W + 3;
    uint32_t row[W];
    uint64_t exc_word;
    ff_load_row<W>(R, P, r, row, exc_word);
    const bool active = r < R.n_reads;
    uint32_t w[WX];
#pragma unroll
    for (int i = 0; i < WX; i++) w[i] = i < W ? row[i] : 0u;
    const bool exc = (exc_word >> (r & 63)) & 1u;
    const uint32_t L = active ? rd_len(R, r) : 0u;
    const int D0 = (int)(P.lowDR + P.lowSp), D1 = min((int)(P.highDR + P.highSp), 16 * W + 15);
    const int wn = (int)P.window;
    // the doubling steps that take a base's flag to a window's (wave-uniform): windows of c bases -> c + s, s = min(c, wn - c)
    const int s1 = min(1, wn - 1), s2 = min(2, wn - 1 - s1), s3 = min(4, wn - 1 - s1 - s2), s4 = wn - 1 - s1 - s2 - s3;      // (wn <= 9: s4 <= 1)
    uint32_t nz[W];                                      // AND over the shifts of the windows' mismatch flags (even bits)
#pragma unroll
    for (int i = 0; i < W; i++) nz[i] = 0xFFFFFFFFu;
#pragma unroll
    for (int q = 0; q <= W; q++) {
        if (16 * q + 15 < D0 || 16 * q > D1) continue;                  // (wave-uniform)
        const int sb_lo = max(0, D0 - 16 * q), sb_hi = min(15, D1 - 16 * q);
        for (int sb = sb_lo; sb <= sb_hi; sb++) {
            const uint32_t sh = (uint32_t)(2 * sb);
            uint32_t z[W + 1];
#pragma unroll
            for (int k = 0; k <= W; k++) {
                const uint32_t x = __builtin_amdgcn_alignbit(w[k + q + 1], w[k + q], sh) ^ w[k];
                z[k] = x | (x >> 1);                     // (odd bits: don't care, they stay on odd bits below)
            }
            // a window of wn bases reaches at most 8 bases = 16 bits into the next word: one word of halo is enough
#pragma unroll
            for (int k = 0; k < W; k++) z[k] |= __builtin_amdgcn_alignbit(z[k + 1], z[k], (uint32_t)(2 * s1));
            if (s2 > 0) {
                z[W] |= z[W] >> (2 * s1);
#pragma unroll
                for (int k = 0; k < W; k++) z[k] |= __builtin_amdgcn_alignbit(z[k + 1], z[k], (uint32_t)(2 * s2));
            }
            if (s3 > 0) {
                z[W] |= z[W] >> (2 * s2);
#pragma unroll
                for (int k = 0; k < W; k++) z[k] |= __builtin_amdgcn_alignbit(z[k + 1], z[k], (uint32_t)(2 * s3));
            }
            if (s4 > 0) {
                z[W] |= z[W] >> (2 * s3);
#pragma unroll
                for (int k = 0; k < W; k++) z[k] |= __builtin_amdgcn_alignbit(z[k + 1], z[k], (uint32_t)(2 * s4));
            }
#pragma unroll
            for (int k = 0; k < W; k++) nz[k] &= z[k];
        }
    }
    bool hit = false;
    const int searchEnd = (int)(L - P.lowDR - P.lowSp - P.window - 1);
    if (active && !exc && searchEnd >= 0) {
        // lattice seeds j = i * skips <= searchEnd; bit i of the hint for the first 32 of them (a superset: the zero padding and the
        // clamp of the window at the read's end are ignored, as in the halfword form)
        const uint32_t skips = P.skips;
        uint32_t hint = 0, more = 0;
        uint32_t j = 0;
        for (uint32_t i = 0; j <= (uint32_t)searchEnd; i++, j += skips) {
            uint32_t word = 0;
#pragma unroll
            for (int k = 0; k < W; k++) word = (j >> 4) == (uint32_t)k ? nz[k] : word;
            const uint32_t m = ((word >> (2u * (j & 15u))) & 1u) ^ 1u;
            if (i < 32u) hint |= m << i; else more |= m;
        }
        hit = (hint | more) != 0;
        if (hit) seed_hint[r] = more ? 0xFFFFFFFFu : hint;      // (a hit beyond the 32nd seed: the hint only says "walk them all")
    }
    const uint64_t m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && active) hitmask[r >> 6] = m;
}

hipError_t launch_filter_fast(const DevReads &R, const DevParams &P, uint64_t *hitmask, uint32_t *seed_hint, hipStream_t st, uint8_t *clear_found, bool *cleared)
{
    if (cleared) *cleared = false;
    if (!R.stride_words || R.n_reads == 0) return hipErrorNotSupported;
    if (P.window != 8 || P.skips != 8) {
        // another window or seed lattice (-w, -d): the every-position form
        if (P.window < 6 || P.window > 9 || P.skips < 1 || P.lowDR + P.lowSp < 17 || P.highDR + P.highSp < P.lowDR + P.lowSp) return hipErrorNotSupported;
        const uint64_t nb = (R.n_reads + 255) / 256;
        if (nb > 0x7FFFFFFFull) return hipErrorNotSupported;
        switch (R.stride_words) {
#define FA_CASE(WW) case WW: CRASS_LAUNCH((k_filter_fast_any<WW>), dim3((unsigned)nb), dim3(256), 0, st, R, P, hitmask, seed_hint); break;
            FA_CASE(4) FA_CASE(5) FA_CASE(6) FA_CASE(7) FA_CASE(8) FA_CASE(9) FA_CASE(10)
            FA_CASE(11) FA_CASE(12) FA_CASE(13) FA_CASE(14) FA_CASE(15) FA_CASE(16)
#undef FA_CASE
            default: return hipErrorNotSupported;
        }
        return hipGetLastError();
    }
    if (P.lowDR + P.lowSp != 49 || P.highDR + P.highSp != 97) {
        // another shift range (-s / -S / -D): the run-time-range form.  (The hint word has 32 bits: reads of up to 16 words.)
        if (P.lowDR + P.lowSp < 17 || P.highDR + P.highSp < P.lowDR + P.lowSp) return hipErrorNotSupported;
        const uint64_t nb = (R.n_reads + 255) / 256;
        if (nb > 0x7FFFFFFFull) return hipErrorNotSupported;
        switch (R.stride_words) {
#define FR_CASE(WW) case WW: CRASS_LAUNCH((k_filter_fast_range<WW>), dim3((unsigned)nb), dim3(256), 0, st, R, P, hitmask, seed_hint); break;
            FR_CASE(4) FR_CASE(5) FR_CASE(6) FR_CASE(7) FR_CASE(8) FR_CASE(9) FR_CASE(10)
            FR_CASE(11) FR_CASE(12) FR_CASE(13) FR_CASE(14) FR_CASE(15) FR_CASE(16)
#undef FR_CASE
            default: return hipErrorNotSupported;
        }
        return hipGetLastError();
    }
    uint64_t blocks = (R.n_reads + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return hipErrorNotSupported;
    // common uniform read lengths get the compile-time clamp and, for big sets, RPL rows per lane with the next row in flight
    // (CRASS_FF_RPL = 1 | 4 forces one form for every size: the A/B and the parity sweep of the other form)
    static const int rpl_env = getenv("CRASS_FF_RPL") ? atoi(getenv("CRASS_FF_RPL")) : 0;
    const int rpl = rpl_env ? rpl_env : (R.n_reads >= (1u << 22) ? 4 : 1);
    dim3 g((unsigned)blocks), b(256), g4((unsigned)((blocks + 3) / 4));
    if (cleared && clear_found) *cleared = true;            // (every path below is k_filter_fast_impl)
#define FF_LEN(LL, WW) if (R.uniform_len == LL && R.stride_words == WW) { \
        if (rpl >= 4) CRASS_LAUNCH((k_filter_fast_impl<WW, 49, 97, LL, 4>), g4, b, 0, st, R, P, hitmask, seed_hint, clear_found); \
        else CRASS_LAUNCH((k_filter_fast_impl<WW, 49, 97, LL, 1>), g, b, 0, st, R, P, hitmask, seed_hint, clear_found); \
        return hipGetLastError(); }
    FF_LEN(100, 7) FF_LEN(101, 7) FF_LEN(125, 8) FF_LEN(126, 8) FF_LEN(150, 10) FF_LEN(151, 10) FF_LEN(250, 16) FF_LEN(251, 16)
#undef FF_LEN
    switch (R.stride_words) {
#define FF_CASE(WW) case WW: CRASS_LAUNCH((k_filter_fast_impl<WW, 49, 97, 0, 1>), g, b, 0, st, R, P, hitmask, seed_hint, clear_found); break;
        FF_CASE(4) FF_CASE(5) FF_CASE(6) FF_CASE(7) FF_CASE(8) FF_CASE(9) FF_CASE(10)
        FF_CASE(11) FF_CASE(12) FF_CASE(13) FF_CASE(14) FF_CASE(15) FF_CASE(16)
#undef FF_CASE
        default: if (cleared) *cleared = false; return hipErrorNotSupported;
    }
    return hipGetLastError();
}

// Hint bits of residue class rho for the 64 positions of one hint word: bit (8 i + rho) is set iff the 8-mer at that position
// has a copy 49 .. 97 bases further on (searchCore's seed test, libcrispr.cpp:295-339; a superset: padding and the right clamp
// are ignored).  w[0..12]: the packed words of the chunk and its halo.  R >> 2 rho puts the seeds 8h + rho on the halfword
// lattice, where the test is the short-read filter's: for a shift d, X = R' ^ (R' >> 2d) has a zero halfword h  <=>  the 8-mer
// at 8h (+ rho) re-occurs d further on — {v_alignbit, v_xor, v_pk_min_u16} per word and shift, 49 x 4 x 3 = 588 instructions
// per 64 positions and CLASS (the all-classes form that k_hint_positions used until round 4 took 2 300 for the eight of them).
static __device__ __forceinline__ uint64_t hint_bits_class(const uint32_t (&w)[13], uint32_t rho)
{
    uint32_t v[12];
    const uint32_t rs = 2u * rho;
#pragma unroll
    for (int i = 0; i < 12; i++) v[i] = rs ? ((w[i] >> rs) | (w[i + 1] << (32u - rs))) : w[i];
    uint32_t acc[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
    for (int d = 49; d <= 97; d++) {
        const int q = d >> 4, sh = (d & 15) * 2;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t lo = v[k + q], hi = v[k + q + 1];
            const uint32_t x = (sh ? ((lo >> sh) | (hi << (32 - sh))) : lo) ^ v[k];
            acc[k] = pk_min_u16(acc[k], x);
        }
    }
    // halfword j of word k -> bit 16 k + 8 j + rho: min(halfword, 1) leaves the miss flags in bytes 0 and 2 of each word, one
    // byte permute gathers those of two words into bytes 0..3
    const uint32_t m0 = pk_nonzero_u16(acc[0]), m1 = pk_nonzero_u16(acc[1]), m2 = pk_nonzero_u16(acc[2]), m3 = pk_nonzero_u16(acc[3]);
    const uint32_t lo = __builtin_amdgcn_perm(m1, m0, 0x06040200u) ^ 0x01010101u;
    const uint32_t hi = __builtin_amdgcn_perm(m3, m2, 0x06040200u) ^ 0x01010101u;
    return ((uint64_t)(hi << rho) << 32) | (uint64_t)(lo << rho);
}

// ... for any shift range D0 .. D1 <= 127 (-s / -S / -D with the defaults' lattice and window): the word part of a shift is a
// compile-time loop, the bit offsets inside it a run-time one (k_filter_fast_range has the reasoning); the defaults keep the
// fully unrolled form above
static __device__ __forceinline__ uint64_t hint_bits_class_range(const uint32_t (&w)[13], uint32_t rho, int D0, int D1)
{
    uint32_t v[12];
    const uint32_t rs = 2u * rho;
#pragma unroll
    for (int i = 0; i < 12; i++) v[i] = rs ? ((w[i] >> rs) | (w[i + 1] << (32u - rs))) : w[i];
    uint32_t acc[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
    for (int q = 0; q < 8; q++) {
        if (16 * q + 15 < D0 || 16 * q > D1) continue;
        const int sb_lo = max(0, D0 - 16 * q), sb_hi = min(15, D1 - 16 * q);
        for (int sb = sb_lo; sb <= sb_hi; sb++) {
            const uint32_t sh = (uint32_t)(2 * sb);
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = pk_min_u16(acc[k], __builtin_amdgcn_alignbit(v[k + q + 1], v[k + q], sh) ^ v[k]);
        }
    }
    const uint32_t m0 = pk_nonzero_u16(acc[0]), m1 = pk_nonzero_u16(acc[1]), m2 = pk_nonzero_u16(acc[2]), m3 = pk_nonzero_u16(acc[3]);
    const uint32_t lo = __builtin_amdgcn_perm(m1, m0, 0x06040200u) ^ 0x01010101u;
    const uint32_t hi = __builtin_amdgcn_perm(m3, m2, 0x06040200u) ^ 0x01010101u;
    return ((uint64_t)(hi << rho) << 32) | (uint64_t)(lo << rho);
}
// (RANGE is a template parameter of the callers: the two forms never share a kernel's register budget)
template <bool RANGE>
static __device__ __forceinline__ uint64_t hint_bits_any(const uint32_t (&w)[13], uint32_t rho, int D0, int D1)
{
    if (RANGE) return hint_bits_class_range(w, rho, D0, D1);
    return hint_bits_class(w, rho);
}

// ------------------------------------------------------------------------------------
// Long reads: per-POSITION seed hints.  A 10 kbp read has ~1 250 lattice seeds and a spurious hit with p ~ 0.93, so a
// per-read filter is useless there; and after a rejected candidate the seed loop leaves the lattice (libcrispr.cpp:390) for
// another residue class mod 8.  Bit p of the read's hint bitmap clear => the 8-mer at p has no copy at p+49 .. p+97 =>
// searchCore's iteration at j = p is a no-op.  This kernel fills the bits of the lattice class (p = 8 i) for every read, one
// lane per 64 positions (4 seed words + halo); the bits of another class are computed by the walking wave itself when — and
// from where — its read's walk moves there (wave_hints_class).  A superset like the short-read filter (padding bases and the
// right clamp are ignored).  Default window / bounds only.
// ------------------------------------------------------------------------------------
// the read a hint tile (64 positions) belongs to and the tile's number inside it: largest r with hint_off[r] <= t
static __device__ __forceinline__ void hint_tile_read(const DevReads &R, const uint64_t *hint_off, const uint32_t *blk_read, uint64_t t, uint64_t n_words,
                                                      uint64_t &r, uint32_t &tile)
{
    // Reads of one length: a division (the search is 20 dependent
    // loads for 1 M reads — about three times the wave's 3.8 us of arithmetic, which six waves per SIMD only just cover)
    if (R.uniform_len) {
        const uint32_t per_read = (R.uniform_len + 63u) >> 6;
        if ((n_words >> 32) == 0) {                                     // (a 32-bit division: a fifth of the 64-bit one's instructions)
            const uint32_t t32 = (uint32_t)t, r32 = t32 / per_read;
            r = r32;
            tile = t32 - r32 * per_read;
        } else {
            r = t / per_read;
            tile = (uint32_t)(t - r * per_read);
        }
    } else if (blk_read) {
        // ragged lengths: the host noted the read of every block's first tile (blk_read[b] = read of tile 256 b); a long read
        // has dozens of tiles, so the tile's own read is a few steps further (short reads mixed in: more steps, same result)
        // (... bisected between this block's first read and the next block's: reads of a few hundred bases are fifty to a block)
        uint64_t lo = blk_read[t >> 8], hi = (uint64_t)blk_read[(t >> 8) + 1] + 1;      // invariant: hint_off[lo] <= t < hint_off[hi]
        if (hi > R.n_reads) hi = R.n_reads;
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (hint_off[mid] <= t) lo = mid; else hi = mid; }
        r = lo;
        tile = (uint32_t)(t - hint_off[r]);
    } else {
        uint64_t lo = 0, hi = R.n_reads;                                // invariant: hint_off[lo] <= t < hint_off[hi]
        while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (hint_off[mid] <= t) lo = mid; else hi = mid; }
        r = lo;
        tile = (uint32_t)(t - hint_off[r]);
    }
}

template <bool RANGE>
__global__ __launch_bounds__(256) void k_hint_positions(DevReads R, const uint64_t *hint_off, const uint32_t *blk_read, uint64_t t0, uint64_t n_words, uint64_t *hint_bits,
                                                        int D0, int D1, uint64_t *hitmask, uint32_t exc_survive)
{
    // (t0: a multiple of 256 — the launch covers the hint words [t0, n_words), see launch_hint_positions)
    const uint64_t t = t0 + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n_words) return;
    uint64_t r;
    uint32_t tile;
    hint_tile_read(R, hint_off, blk_read, t, n_words, r, tile);
    const uint32_t L = rd_len(R, r);
    const uint32_t nw = (L + 15) >> 4;
    const uint32_t *g = R.packed + rd_word_off(R, r) + tile * 4u;
    const uint32_t rem = nw - tile * 4u;                                 // words from this tile's first to the read's end
    uint32_t w[13];
    // 13 words: the tile's four and the halo.  A wave without a lane near its read's end (6 of 10 at 10 kbp) takes them as three
    // 16-byte loads and a word from one address — no bound per word; the last read of the set never does (nothing is read
    // past the reads' buffer)
    typedef uint32_t hp_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
    if (__ballot(rem < 13u || r + 1 >= R.n_reads) == 0ull) {
        const hp_u32x4 a = *reinterpret_cast<const hp_u32x4 *>(g), b = *reinterpret_cast<const hp_u32x4 *>(g + 4), c4 = *reinterpret_cast<const hp_u32x4 *>(g + 8);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        w[8] = c4.x; w[9] = c4.y; w[10] = c4.z; w[11] = c4.w; w[12] = g[12];
    } else {
#pragma unroll
        for (int i = 0; i < 13; i++) w[i] = (uint32_t)i < rem ? g[i] : 0u;
    }
    // the LATTICE class only (positions 8 i): searchCore walks one residue class at a time and leaves it only behind a rejected
    // candidate (libcrispr.cpp:390,295) — 0.76 times per 10 kbp read on BASELINE configs[3] (tools/class_switches.py), so seven
    // of the eight classes this kernel used to cover were never looked at.  The class a walk moves to is covered from there on by
    // the wave that walks (search_core, wave_hints_class)
    uint64_t bits = hint_bits_any<RANGE>(w, 0u, D0, D1);
    // seeds live at j <= searchEnd = L - D0 - 9 (searchCore's loop bound; 58 with the default bounds): the bits behind it — the
    // zero padding matches itself — are cleared, so that a read without a real lattice hint has an all-zero bitmap and the
    // walking wave can drop it without staging it (k_survivor, no_seed)
    const int last = (int)L - D0 - 9 - (int)(tile * 64u);
    bits = last < 0 ? 0ull : (last >= 63 ? bits : (bits & ((2ull << last) - 1ull)));
    hint_bits[t] = bits;
    // the bits as the seed-scan FILTER of a set without a lane-per-read filter (reads of 257 .. 2 048 bases, strides that differ):
    // bit r of the (cleared) hitmask = read r has a hint bit.  With everything behind searchEnd cleared, an all-zero bitmap proves
    // that searchCore's seed loop (libcrispr.cpp:295-348) finds nothing; a superset like the other filters, exception reads as in
    // k_filter_general.  A few per cent of the tiles hold a bit: the atomics are rare
    if (hitmask && bits && (exc_survive || !rd_is_exc(R, r))) atomicOr(reinterpret_cast<unsigned long long *>(hitmask) + (r >> 6), 1ull << (r & 63u));
}

hipError_t launch_hint_positions(const DevReads &R, const DevParams &P, const uint64_t *hint_off, const uint32_t *blk_read, uint64_t n_words,
                                 uint64_t *hint_bits, hipStream_t st, uint64_t w_begin, uint64_t w_end, uint64_t *hitmask)
{
    // the defaults' lattice and window; any shift range the 13-word tile covers (a copy at most 127 bases on)
    if (P.window != 8 || P.skips != 8 || P.lowDR + P.lowSp < 17 || P.highDR + P.highSp > 127 || P.highDR + P.highSp < P.lowDR + P.lowSp) return hipErrorNotSupported;
    if (w_end > n_words) w_end = n_words;
    if ((w_begin & 255u) != 0) return hipErrorInvalidValue;
    if (w_begin >= w_end) return hipSuccess;
    const int D0 = (int)(P.lowDR + P.lowSp), D1 = (int)(P.highDR + P.highSp);
    const dim3 hg((unsigned)((w_end - w_begin + 255) / 256));
    static const int force_hp = getenv("CRASS_HINT_RANGE") ? atoi(getenv("CRASS_HINT_RANGE")) : 0;      // A/B: 1 = the run-time-range form for the hint kernel, 2 = for the light walk, 3 = both
    if (D0 == 49 && D1 == 97 && !(force_hp & 1)) CRASS_LAUNCH(k_hint_positions<false>, hg, dim3(256), 0, st, R, hint_off, blk_read, w_begin, w_end, hint_bits, D0, D1, hitmask, P.exc_survive);
    else CRASS_LAUNCH(k_hint_positions<true>, hg, dim3(256), 0, st, R, hint_off, blk_read, w_begin, w_end, hint_bits, D0, D1, hitmask, P.exc_survive);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// ordered compaction: bit mask -> ascending list of set-bit indices
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mask_count(const uint64_t *mask, uint64_t n_words, uint64_t n_bits,
                                                     uint32_t *word_prefix, uint32_t *block_sums)
{
    __shared__ uint32_t sh[256];
    uint64_t wi = blockIdx.x * 256ull + threadIdx.x;
    uint32_t c = 0;
    if (wi < n_words) {
        uint64_t m = mask[wi];
        uint64_t rem = n_bits - wi * 64;
        if (rem < 64) m &= (1ull << rem) - 1ull;
        c = (uint32_t)__popcll(m);
    }
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint32_t v = (threadIdx.x >= (unsigned)off) ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    if (wi < n_words) word_prefix[wi] = sh[threadIdx.x] - c;
    if (threadIdx.x == 255) block_sums[blockIdx.x] = sh[255];
}

__global__ __launch_bounds__(1024) void k_block_scan(uint32_t *block_sums, uint32_t n_blocks, uint32_t *d_count, uint32_t *zero_a, uint32_t n_a, uint32_t *zero_b, uint32_t n_b)
{
    // counters the NEXT stage accumulates into are cleared here instead of by their own fill launches
    if (threadIdx.x < n_a) zero_a[threadIdx.x] = 0u;
    if (threadIdx.x < n_b) zero_b[threadIdx.x] = 0u;

    __shared__ uint32_t sh[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n_blocks; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t c = (i < n_blocks) ? block_sums[i] : 0;
        sh[threadIdx.x] = c;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            uint32_t v = (threadIdx.x >= (unsigned)off) ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += v;
            __syncthreads();
        }
        uint32_t excl = sh[threadIdx.x] - c + carry;
        if (i < n_blocks) block_sums[i] = excl;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *d_count = carry;
}

__global__ __launch_bounds__(256) void k_mask_scatter(const uint64_t *mask, uint64_t n_words, uint64_t n_bits,
                                                       const uint32_t *word_prefix, const uint32_t *block_sums,
                                                       uint64_t *out_idx, uint64_t out_cap)
{
    uint64_t wi = blockIdx.x * 256ull + threadIdx.x;
    if (wi >= n_words) return;
    uint64_t m = mask[wi];
    uint64_t rem = n_bits - wi * 64;
    if (rem < 64) m &= (1ull << rem) - 1ull;
    uint64_t o = (uint64_t)block_sums[blockIdx.x] + word_prefix[wi];
    while (m) {
        int b = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
        if (o < out_cap) out_idx[o] = wi * 64 + b;
        o++;
    }
}

// ---- single-pass form: decoupled look-back over tiles of 1024 mask words ----
static __device__ __forceinline__ unsigned long long lb_pack(uint32_t epoch, uint32_t flag, uint32_t value)
{
    return ((unsigned long long)epoch << 34) | ((unsigned long long)flag << 32) | value;
}
// Exclusive prefix of this tile's total over the tiles before it.  Called by every thread of the block
// (one __syncthreads inside); wave 0 does the look-back, 64 predecessor tiles per step.
static __device__ uint32_t lb_exclusive_prefix(const Lookback &lb, uint32_t tile, uint32_t total)
{
    __shared__ uint32_t excl_sh;
    if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        if (lane == 0)
            __hip_atomic_store(&lb.status[tile], lb_pack(lb.epoch, tile == 0 ? 2u : 1u, total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        int t = (int)tile - 1;                           // wave-uniform: this step looks at tiles t, t-1, ..., t-63
        while (t >= 0) {
            const int idx = t - lane;
            uint32_t flag = 2u, val = 0u;                // tiles before tile 0: an empty prefix
            if (idx >= 0) {
                flag = 0u;
                for (uint32_t spins = 0; spins < (1u << 24); spins++) {
                    const unsigned long long w = __hip_atomic_load(&lb.status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(w >> 34) == lb.epoch && ((w >> 32) & 3ull) != 0ull) { flag = (uint32_t)(w >> 32) & 3u; val = (uint32_t)w; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (flag == 0u) { *lb.fail = 1u; flag = 2u; }               // gave up (pinned host word): terminate, the host reports it
            }
            const unsigned long long pm = __ballot(flag == 2u);
            const int stop = pm ? __ffsll(pm) - 1 : 64;  // nearest tile whose inclusive prefix is known
            uint32_t v = lane <= stop ? val : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off);
            excl += v;
            if (pm) break;
            t -= 64;
        }
        if (lane == 0) {
            if (tile != 0) __hip_atomic_store(&lb.status[tile], lb_pack(lb.epoch, 2u, excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            excl_sh = excl;
        }
    }
    __syncthreads();
    return excl_sh;
}
// A tile id per block, in start order.  n_act blocks of the launch call this (every block computes the same n_act; the
// others have returned before): the block that draws the last ticket puts the counter back to zero for the next launch — by
// then every other ticket of this launch has been drawn.  (The counter is ONE word for all launches of a context's stream.
// Returning atomics on one address retire every 20-30 ns on this part however many CUs issue them — the whole cost of a
// look-back kernel over a few thousand tiles, rocprofv3 round 4: 1 526 tiles of the read mask 32 us, 4 950 mostly EMPTY
// tiles of the candidate list 50 us — so tiles are fat, and tiles past a device-side count draw no ticket at all.)
static __device__ uint32_t lb_tile_id(const Lookback &lb, uint32_t n_act)
{
    __shared__ uint32_t tile_sh;
    if (threadIdx.x == 0) {
        const uint32_t t = atomicAdd(lb.ticket, 1u);
        if (t + 1u >= n_act) __hip_atomic_store(lb.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tile_sh = t;
    }
    __syncthreads();
    return tile_sh;
}
// exclusive prefix of v over the T threads of the block; *total = block sum
template <int T>
static __device__ uint32_t block_scan_t(uint32_t v, uint32_t *total)
{
    __shared__ uint32_t wsum[T / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)incl, off);
        if (lane >= off) incl += y;
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (int k = 0; k < T / 64; k++) { const uint32_t s = wsum[k]; if (k < wv) base += s; all += s; }
    *total = all;
    return base + incl - v;
}

// T threads x W consecutive mask words per tile: 1024 x 4 for the masks over all reads (few, fat tiles: the ticket is the
// kernel's cost), 256 x 1 for the short dense masks of the later stages (more blocks for the per-word scatter loops)
template <int T, int W>
__global__ __launch_bounds__(T) void k_mask_compact_lb(const uint64_t *mask, uint64_t n_words, uint64_t n_bits, uint32_t *word_prefix,
                                                        uint32_t *block_sums, uint64_t *out_idx, uint64_t out_cap, uint32_t *d_count,
                                                        uint32_t *zero_a, uint32_t n_a, uint32_t *zero_b, uint32_t n_b, Lookback lb,
                                                        uint32_t n_tiles)
{
    const uint32_t tile = lb_tile_id(lb, n_tiles);
    if (tile == 0) {                                    // counters the NEXT stage accumulates into
        if (threadIdx.x < n_a) zero_a[threadIdx.x] = 0u;
        if (threadIdx.x < n_b) zero_b[threadIdx.x] = 0u;
    }
    const uint64_t w0 = ((uint64_t)tile * T + threadIdx.x) * W;
    uint64_t m[W];
    uint32_t cnt = 0;
#pragma unroll
    for (int q = 0; q < W; q++) {
        const uint64_t wi = w0 + q;
        m[q] = 0;
        if (wi < n_words) {
            m[q] = mask[wi];
            const uint64_t rem = n_bits - wi * 64;
            if (rem < 64) m[q] &= (1ull << rem) - 1ull;
        }
        cnt += (uint32_t)__popcll(m[q]);
    }
    uint32_t total;
    const uint32_t in_tile = block_scan_t<T>(cnt, &total);
    const uint32_t excl = lb_exclusive_prefix(lb, tile, total);
    if (tile == n_tiles - 1 && threadIdx.x == 0) *d_count = excl + total;
    uint64_t o = (uint64_t)excl + in_tile;
#pragma unroll
    for (int q = 0; q < W; q++) {
        const uint64_t wi = w0 + q;
        if (wi >= n_words) break;
        if (word_prefix) {                              // same meaning as the three-kernel form: block_sums[w >> 8] + word_prefix[w]
            word_prefix[wi] = (uint32_t)o;
            if ((wi & 255u) == 0) block_sums[wi >> 8] = 0u;
        }
        uint64_t mm = m[q];
        while (mm) {
            const int b = __ffsll((unsigned long long)mm) - 1;
            mm &= mm - 1;
            if (o < out_cap) out_idx[o] = wi * 64 + b;
            o++;
        }
    }
}

// element-wise form: tiles of 4096 elements, 1024 threads x 4 consecutive elements (few, fat tiles keep the look-back
// to one or two steps).  cnt = flagged elements of this thread; returns the rank of the thread's first flagged element
// among all flagged elements before it; *upto = flagged elements up to and including this tile.
static constexpr uint32_t kLbElemsPerTile = 4096;
static __device__ uint32_t lb_rank4(const Lookback &lb, uint32_t tile, uint32_t cnt, uint32_t *upto)
{
    uint32_t all;
    const uint32_t in_tile = block_scan_t<1024>(cnt, &all);
    const uint32_t excl = lb_exclusive_prefix(lb, tile, all);
    *upto = excl + all;
    return excl + in_tile;
}

hipError_t launch_compact(const uint64_t *mask, uint64_t n_words, uint64_t n_bits, uint32_t *word_prefix,
                          uint32_t *block_sums, uint64_t *out_idx, uint64_t out_cap, uint32_t *d_count, hipStream_t st,
                          uint32_t *zero_a, uint32_t n_a, uint32_t *zero_b, uint32_t n_b, const Lookback *lb)
{
    if (lb && n_words) {
        // (the caller reserved ceil(n_words / lookback_tile_words(n_words)) tickets)
        const uint32_t tw = lookback_tile_words(n_words);
        const uint32_t n_tiles = (uint32_t)((n_words + tw - 1) / tw);
        if (tw == 256)
            CRASS_LAUNCH((k_mask_compact_lb<256, 1>), dim3(n_tiles), dim3(256), 0, st, mask, n_words, n_bits, word_prefix, block_sums, out_idx, out_cap,
                               d_count, zero_a, n_a, zero_b, n_b, *lb, n_tiles);
        else
            CRASS_LAUNCH((k_mask_compact_lb<1024, 4>), dim3(n_tiles), dim3(1024), 0, st, mask, n_words, n_bits, word_prefix, block_sums, out_idx, out_cap,
                               d_count, zero_a, n_a, zero_b, n_b, *lb, n_tiles);
        return hipGetLastError();
    }
    if (n_words == 0) {
        if (n_a) (void)hipMemsetAsync(zero_a, 0, 4 * (size_t)n_a, st);
        if (n_b) (void)hipMemsetAsync(zero_b, 0, 4 * (size_t)n_b, st);
        return hipMemsetAsync(d_count, 0, 4, st);
    }
    unsigned nb = (unsigned)((n_words + 255) / 256);
    CRASS_LAUNCH(k_mask_count, dim3(nb), dim3(256), 0, st, mask, n_words, n_bits, word_prefix, block_sums);
    CRASS_LAUNCH(k_block_scan, dim3(1), dim3(1024), 0, st, block_sums, nb, d_count, zero_a, n_a, zero_b, n_b);
    CRASS_LAUNCH(k_mask_scatter, dim3(nb), dim3(256), 0, st, mask, n_words, n_bits, word_prefix, block_sums, out_idx, out_cap);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// pass 1, step 2: the survivor kernel — one wave executes the reference's searchCore for
// one read, byte for byte, with wave-parallel inner operations.
// ------------------------------------------------------------------------------------
struct RH {                 // ReadHolder state (ReadHolder.h:440-451), wave-uniform
    uint8_t *seq;           // LDS, L bytes
    int L;
    uint32_t *ss;           // LDS, RH_StartStops
    int nss, cap;
    int replen;             // RH_RepeatLength
    int err;
    uint16_t *rowA, *rowB;  // Levenshtein block-boundary rows (LDS)
    int row_cap;            // entries per row; a longer string makes qc return -3 (the read is redone with full-size rows)
    float *sims;            // LDS scratch for the QC similarities (ss_cap floats)
    const uint32_t *words;  // LDS copy of the 2-bit packed read (+1 zero word), nullptr for exception reads
    uint32_t cmask;         // (1 << 2w) - 1
    int asc_lo, asc_hi;     // packed reads: seq[] holds the ASCII bases of [asc_lo, asc_hi) only (rh_ascii), wave-uniform
    int asc_pad;            // how far beyond the repeats found so far the byte-wise consumers may read
    uint8_t *seq_buf;       // the LDS bytes behind seq; with a window (seq_win != 0) seq = seq_buf - asc_lo, so that seq[pos] still works
    int seq_win;            // bytes of the ASCII window (0: the whole read fits)
    unsigned long long *lprof;  // diagnostics (CRASS_SURV_PROF): this read's phase cycles / counts in LDS, nullptr normally
};
// phases of one read in the wave kernel (CRASS_SURV_PROF=1, tools/longread_phases.py): cycles unless named n_*
enum { PF_TOTAL = 0, PF_STAGE, PF_FIND, PF_SCAN, PF_EXTEND, PF_QC, PF_HINTS, PF_OUT, PF_N_CAND, PF_N_SCANFIND, PF_N_SWITCH, PF_N_QC, PF_LOOP, PF_N_ITER, PF_MAX, PF_READS, PF_SLOTS };
// LDS behind the layout: PF_SLOTS per-read slots, then 4 categories x PF_SLOTS sums of this wave, then 4 x PF_BINS histogram bins
#define PF_BINS 24
#define PF_LDS_WORDS (PF_SLOTS + 4 * PF_SLOTS + 4 * PF_BINS)
#define PROF_T0(h) const unsigned long long _pt0 = (h).lprof ? (unsigned long long)__builtin_readcyclecounter() : 0ull
#define PROF_ADD(h, ph) do { if ((h).lprof) { const unsigned long long _d = (unsigned long long)__builtin_readcyclecounter() - _pt0; if (lane == 0) (h).lprof[ph] += _d; } } while (0)
#define PROF_CNT(h, ph) do { if ((h).lprof && lane == 0) (h).lprof[ph] += 1ull; } while (0)

// leftmost occurrence of seq[pat, pat+plen) in seq[begin, end): PatternMatcher::bmpSearch
// semantics (PatternMatcher.cpp:26-59: -1 for empty text/pattern or pattern longer than text)
static __device__ int wave_find(const uint8_t *seq, int begin, int end, int pat, int plen, int lane)
{
    int tlen = end - begin;
    if (tlen <= 0 || plen <= 0 || plen > tlen) return -1;
    for (int p0 = begin; p0 + plen <= end; p0 += WAVE) {
        int p = p0 + lane;
        bool ok = (p + plen <= end);
        if (ok) {
            for (int k = 0; k < plen; k++) {
                if (seq[p + k] != seq[pat + k]) { ok = false; break; }
            }
        }
        uint64_t m = __ballot(ok);
        if (m) return p0 + (__ffsll((unsigned long long)m) - 1);
    }
    return -1;
}

// same contract on the 2-bit packed copy: a w-mer is one <=18-bit code, so a window compare is
// one LDS word pair + funnel shift per lane instead of w byte compares
static __device__ int wave_find_packed(const uint32_t *words, uint32_t cmask, int begin, int end, int pat, int plen, int lane)
{
    int tlen = end - begin;
    if (tlen <= 0 || plen <= 0 || plen > tlen) return -1;
    const uint32_t sj = uni(lds_code(words, (uint32_t)pat, cmask));
    for (int p0 = begin; p0 + plen <= end; p0 += WAVE) {
        int p = p0 + lane;
        bool ok = (p + plen <= end) && (lds_code(words, (uint32_t)p, cmask) == sj);
        uint64_t m = __ballot(ok);
        if (m) return p0 + (__ffsll((unsigned long long)m) - 1);
    }
    return -1;
}

static __device__ __forceinline__ int rh_find(const RH &h, int begin, int end, int pat, int plen, int lane)
{
    if (h.words) return wave_find_packed(h.words, h.cmask, begin, end, pat, plen, lane);
    return wave_find(h.seq, begin, end, pat, plen, lane);
}

// The byte-wise consumers (extendPreRepeat's column votes, the QC's string comparisons, DRLowLexi) read the bases around the
// repeats found so far, never far from them — a packed read's ASCII copy is expanded for that REGION when they are entered,
// not for the whole read when it is loaded (a 10 kbp read: 10 KB of LDS writes per read against 2.5 KB for the packed words;
// 61 % of random 10 kbp reads meet a candidate, which covers a few hundred bases).  kAscPad bounds how far the consumers
// reach beyond the first repeat's start / the last one's end: one repeat spacing (highDR + highSp, 97 by default) for the
// extensions; RH::asc_pad = that + 32.
static __device__ __forceinline__ void word_to_ascii(uint32_t v, uint32_t o[4]);
static __device__ void rh_ascii(RH &h, int lane)
{
    if (!h.words || h.nss < 2) return;                  // exception reads: the bytes are the read
    int lo = (int)uni(h.ss[0]) - h.asc_pad, hi = (int)uni(h.ss[h.nss - 1]) + h.asc_pad;
    if (lo < 0) lo = 0;
    if (hi > h.L) hi = h.L;
    if (lo >= h.asc_lo && hi <= h.asc_hi) return;       // (wave-uniform)
    const int w0 = lo >> 4, w1 = (hi + 15) >> 4;
    // (only the candidate in hand is ever read: an earlier candidate's region is dropped.)  With a window: the region
    // starts at the buffer's first byte and seq is moved so that seq[pos] addresses it; a region that does not fit sends the
    // read to the launch with the full layout
    if (h.seq_win) {
        if ((w1 - w0) * 16 > h.seq_win) { h.err = 6; return; }
        h.seq = h.seq_buf - w0 * 16;
    }
    for (int wi = w0 + lane; wi < w1; wi += WAVE) {
        uint32_t o[4];
        word_to_ascii(h.words[wi], o);
        uint32_t *dst = reinterpret_cast<uint32_t *>(h.seq + 16 * wi);   // seq region is 16-B aligned and padded
        dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2]; dst[3] = o[3];
    }
    h.asc_lo = w0 * 16; h.asc_hi = min(w1 * 16, h.L + 16);
    wave_sync();
}

// ReadHolder::startStopsAdd, ReadHolder.cpp:263-297
static __device__ void rh_add(RH &h, uint32_t i, uint32_t j, int lane)
{
    if (h.nss + 2 > h.cap) { h.err = 2; return; }
    if (j >= (uint32_t)h.L) j = (uint32_t)h.L - 1;
    if (lane == 0) { h.ss[h.nss] = i; h.ss[h.nss + 1] = j; }
    h.nss += 2;
    wave_sync();
}

static __device__ void scan_right_masks(RH &h, int pat, uint32_t pattern_length, uint32_t minSpacerLength, uint32_t scanRange,
                                        uint32_t last_repeat_index, uint32_t second_last_repeat_index, int lane);
// scanRight, libcrispr.cpp:170-263
static __device__ void scan_right(RH &h, int pat, uint32_t pattern_length, uint32_t minSpacerLength,
                                  uint32_t scanRange, int lane)
{
    uint32_t last_repeat_index = uni(h.ss[h.nss - 2]);
    uint32_t second_last_repeat_index = uni(h.ss[h.nss - 4]);
    uint32_t repeat_spacing = last_repeat_index - second_last_repeat_index;
    const uint32_t read_length = (uint32_t)h.L;
    bool more_to_search = true;
    while (more_to_search) {
        int candidate_repeat_index = (int)(last_repeat_index + repeat_spacing);
        uint32_t begin_search = (uint32_t)candidate_repeat_index - scanRange;
        uint32_t end_search = (uint32_t)candidate_repeat_index + pattern_length + scanRange;
        uint32_t scanRightMinBegin = last_repeat_index + pattern_length + minSpacerLength;
        if (begin_search < scanRightMinBegin) begin_search = scanRightMinBegin;
        if (begin_search > read_length - 1) return;
        if (end_search > read_length) end_search = read_length;
        if (begin_search >= end_search) return;
        int position = rh_find(h, (int)begin_search, (int)end_search, pat, (int)pattern_length, lane);
        PROF_CNT(h, PF_N_SCANFIND);
        if (position >= 0) {
            uint32_t found = (uint32_t)position;        // wave_find returns absolute positions
            rh_add(h, found, found + pattern_length - 1, lane);
            if (h.err) return;
            second_last_repeat_index = last_repeat_index;
            last_repeat_index = found;
            repeat_spacing = last_repeat_index - second_last_repeat_index;
            if (repeat_spacing < (minSpacerLength + pattern_length)) more_to_search = false;
            else if (h.words && pattern_length <= 9u) {
                // a third repeat: this is an array, not a chance match — the rest of the chain on match masks
                scan_right_masks(h, pat, pattern_length, minSpacerLength, scanRange, last_repeat_index, second_last_repeat_index, lane);
                return;
            }
        } else {
            more_to_search = false;
        }
    }
}

// extendPreRepeat, libcrispr.cpp:520-772.  The per-column A/C/G/T votes run over repeats in lanes.
static __device__ bool extend_pre_repeat_packed(RH &h, int searchWindowLength, int minSpacerLength, int lane);
static __device__ uint32_t extend_pre_repeat(RH &h, int searchWindowLength, int minSpacerLength, int lane)
{
    // packed reads with at most 64 repeats: on the 2-bit words, no ASCII window (defined with the other packed forms below)
    if (h.words && extend_pre_repeat_packed(h, searchWindowLength, minSpacerLength, lane)) return (uint32_t)h.replen;
    rh_ascii(h, lane);
    if (h.err == 6) return 0;                           // (the region does not fit the ASCII window: search_core hands the read over)
    const uint32_t num_repeats = (uint32_t)h.nss / 2;
    h.replen = searchWindowLength;
    int cut_off = (int)(num_repeats - 1);
    if (2 > cut_off) cut_off = 2;
    const uint32_t first_repeat_start_index = h.ss[0];
    const uint32_t last_repeat_start_index = h.ss[h.nss - 2];
    const uint32_t end_index = (uint32_t)h.nss;
    const uint32_t seqlen = (uint32_t)h.L;
    // shortest spacing between consecutive starts (:557-575)
    uint32_t shortest = 0xFFFFFFFFu;
    for (uint32_t i = 2 + 2 * lane; i < end_index; i += 2 * WAVE) {
        uint32_t sp = (uint32_t)((int)h.ss[i] - (int)h.ss[i - 2]);
        shortest = min(shortest, sp);
    }
    for (int off = 32; off > 0; off >>= 1) shortest = min(shortest, (uint32_t)__shfl_xor((int)shortest, off));
    const uint32_t shortest_repeat_spacing = shortest;

    // The reference extends one column per loop iteration (:612-668, :698-740).  Iteration e of the
    // right loop runs with right_extension_length == e, RH_RepeatLength == w+e and, because one more
    // trailing repeat is dropped in EVERY iteration once lastStart+w+e >= L (:614-616),
    // DR_index_end(e) = end_index - 2*max(0, e - (L-lastStart-w) + 1).  Whether column e passes the
    // vote does not depend on the earlier columns, so all columns are evaluated at once — lane = column —
    // and the extension is the number of leading passing columns (capped by max_*_extension_length).
    uint32_t right_extension_length = 0;
    const uint32_t max_right_extension_length = shortest_repeat_spacing - (uint32_t)minSpacerLength;
    {
        const int T = (int)seqlen - (int)last_repeat_start_index - searchWindowLength;     // >= 0
        for (uint32_t e0 = 0; e0 < max_right_extension_length; e0 += WAVE) {
            const uint32_t e = e0 + (uint32_t)lane;
            int drops = (int)e - T + 1;
            if (drops < 0) drops = 0;
            const int dr_index_end = (int)end_index - 2 * drops;
            int cA = 0, cC = 0, cG = 0, cT = 0;
            for (int k = 0; k < dr_index_end; k += 2) {
                const uint32_t pos = h.ss[k] + (uint32_t)searchWindowLength + e;
                if (pos >= seqlen) break;                // starts ascend: every later repeat is off the read too (:624-627)
                const uint8_t ch = h.seq[pos];
                cA += (ch == 'A'); cC += (ch == 'C'); cG += (ch == 'G'); cT += (ch == 'T');
            }
            const bool pass = (e < max_right_extension_length) &&
                              ((cA >= cut_off) || (cC >= cut_off) || (cG >= cut_off) || (cT >= cut_off));
            const uint64_t fails = ~__ballot(pass);
            if (fails) { right_extension_length = e0 + (uint32_t)(__ffsll((unsigned long long)fails) - 1); break; }
            right_extension_length = e0 + WAVE;
        }
    }
    h.replen = searchWindowLength + (int)right_extension_length;

    uint32_t left_extension_length = 0;
    const int test_for_negative = (int)(shortest_repeat_spacing - (uint32_t)h.replen);
    const uint32_t max_left_extension_length = (test_for_negative >= 0) ? (uint32_t)test_for_negative : 0;
    for (uint32_t e0 = 0; e0 < max_left_extension_length; e0 += WAVE) {
        const uint32_t e = e0 + (uint32_t)lane;
        int drops = (int)e - (int)first_repeat_start_index + 1;          // iterations i<=e with firstStart-i <= 0 (:700-704)
        if (drops < 0) drops = 0;
        int cA = 0, cC = 0, cG = 0, cT = 0;
        for (uint32_t k = 2u * (uint32_t)drops; k < end_index; k += 2) {
            const int idx = (int)(h.ss[k] - e - 1);
            uint8_t ch = 0;
            if (idx >= 0 && idx < h.L) ch = h.seq[idx];
            cA += (ch == 'A'); cC += (ch == 'C'); cG += (ch == 'G'); cT += (ch == 'T');
        }
        const bool pass = (e < max_left_extension_length) &&
                          ((cA >= cut_off) || (cC >= cut_off) || (cG >= cut_off) || (cT >= cut_off));
        const uint64_t fails = ~__ballot(pass);
        if (fails) { left_extension_length = e0 + (uint32_t)(__ffsll((unsigned long long)fails) - 1); break; }
        left_extension_length = e0 + WAVE;
    }
    h.replen += (int)left_extension_length;
    wave_sync();
    for (int r = 2 * lane; r + 1 < h.nss; r += 2 * WAVE) {
        uint32_t a = h.ss[r], b = h.ss[r + 1];
        a = (a < left_extension_length) ? 0 : a - left_extension_length;
        b = (b + right_extension_length >= seqlen) ? seqlen - 1 : b + right_extension_length;
        h.ss[r] = a; h.ss[r + 1] = b;
    }
    wave_sync();
    return (uint32_t)h.replen;
}

// isRepeatLowComplexity, libcrispr.cpp:1031-1069
static __device__ bool is_low_complexity(const uint8_t *rep, int n, int lane)
{
    int a = 0, c = 0, g = 0, t = 0, o = 0;
    for (int i0 = 0; i0 < n; i0 += WAVE) {
        int i = i0 + lane;
        bool v = i < n;
        uint8_t ch = v ? rep[i] : 0;
        uint8_t up = ch & 0xDF;          // case-insensitive for letters
        bool isA = v && up == 'A', isC = v && up == 'C', isG = v && up == 'G', isT = v && up == 'T';
        a += __popcll(__ballot(isA)); c += __popcll(__ballot(isC));
        g += __popcll(__ballot(isG)); t += __popcll(__ballot(isT));
        o += __popcll(__ballot(v && !(isA || isC || isG || isT)));
    }
    int cut_off = (int)((double)n * 0.75);
    return (a > cut_off) || (t > cut_off) || (g > cut_off) || (c > cut_off) || (o > cut_off);
}

// value of lane-1 (lane 0 receives 0): one DPP move (v_mov_b32_dpp wave_shr:1), a few cycles,
// instead of a ds_bpermute round trip through the LDS crossbar — the DP below is a dependent
// chain of these, so their latency is the kernel's critical path.
static __device__ __forceinline__ int wave_shr1(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// PatternMatcher::levenstheinDistance, PatternMatcher.cpp:111-195, as an anti-diagonal wavefront:
// lane = DP row inside a 64-row block, one DP column step per iteration; neighbours arrive by
// lane shuffles, block-boundary rows go through LDS.  The recurrence (including the
// non-standard transposition term for i>2 && j>2) is symmetric in its arguments, so the
// shorter string is put on the rows.
static __device__ int wave_lev(const uint8_t *s, int n, const uint8_t *t, int m,
                               uint16_t *rowA, uint16_t *rowB, int lane)
{
    if (n == 0) return m;
    if (m == 0) return n;
    if (n > m) { const uint8_t *x = s; s = t; t = x; int y = n; n = m; m = y; }
    int res = 0;
    const int nblocks = (n + WAVE - 1) / WAVE;
    for (int b = 0; b < nblocks; b++) {
        const int i = b * WAVE + lane + 1;               // DP row (1-based)
        const bool rowvalid = i <= n;
        const uint8_t s_i = rowvalid ? s[i - 1] : 0;                  // source[i-1]
        const uint8_t s_im = (rowvalid && i >= 2) ? s[i - 2] : 0;     // source[i-2]
        int left = i;                 // M[i][j-1]; M[i][0] = i
        int pa = 0, ppa = 0, pppa = 0;   // `above` values used at the previous three active steps
        const int rows_here = min(WAVE, n - b * WAVE);
        const int steps = m + rows_here - 1;
        const bool more = (b + 1 < nblocks);
        for (int d = 0; d < steps; d++) {
            const int j = d - lane + 1;                  // my column this step
            int up_left = wave_shr1(left);               // lane-1: M[i-1][j]
            int up_pppa = wave_shr1(pppa);               // lane-1: M[i-2][j-2]
            const bool act = rowvalid && j >= 1 && j <= m;
            if (lane == 0) {
                if (b == 0) { up_left = j; up_pppa = 0; }
                else {
                    up_left = (j >= 0 && j <= m) ? (int)rowA[j] : 0;
                    up_pppa = (j >= 2 && j - 2 <= m) ? (int)rowB[j - 2] : 0;
                }
            }
            if (act) {
                const uint8_t t_j = t[j - 1];
                const int above = up_left;
                const int diag = (j == 1) ? (i - 1) : pa;
                const int cost = (s_i == t_j) ? 0 : 1;
                int cell = min(above + 1, min(left + 1, diag + cost));
                if (i > 2 && j > 2) {
                    int trans = up_pppa + 1;
                    if (s_im != t_j) trans++;
                    if (s_i != t[j - 2]) trans++;
                    if (cell > trans) cell = trans;
                }
                pppa = ppa; ppa = pa; pa = above;
                left = cell;
                if (i == n && j == m) res = cell;
                if (more) {
                    if (lane == WAVE - 1) rowA[j] = (uint16_t)cell;       // M[64(b+1)][j]
                    if (lane == WAVE - 2) rowB[j] = (uint16_t)cell;       // M[64(b+1)-1][j]
                }
            }
        }
        if (more) {
            if (lane == WAVE - 1) rowA[0] = (uint16_t)(b * WAVE + WAVE);
            if (lane == WAVE - 2) rowB[0] = (uint16_t)(b * WAVE + WAVE - 1);
            wave_sync();
        }
    }
    // broadcast the result from the lane that owns row n
    const int owner = (n - 1) & (WAVE - 1);
    return __shfl(res, owner);
}

// PatternMatcher::getStringSimilarity, PatternMatcher.cpp:197-204
static __device__ float wave_similarity(const uint8_t *s1, int n, const uint8_t *s2, int m,
                                        uint16_t *rowA, uint16_t *rowB, int lane)
{
    float max_length = (float)(n > m ? n : m);
    if (n < 3 || m < 3) return 0.0f;
    float edit_distance = (float)wave_lev(s1, n, s2, m, rowA, rowB, lane);
    return (float)(1.0 - (double)(edit_distance / max_length));
}

// The same distance, one pair per LANE, bit-parallel (Myers / Hyyro 2003 OSA recurrences on one
// 64-bit column word).  The reference's transposition term only ever lowers a cell when both
// crossed characters match (otherwise M[i-2][j-2]+2 or +3 >= the substitution path), i.e. it is
// the OSA transposition restricted to i>2 && j>2: TR is masked for rows 1,2 (bits 0,1) and for
// the first two text columns.  Needs the shorter string to be <= 64 long and ACGT-only (the
// longer one may hold any byte: it simply matches nothing); returns -1 otherwise and the caller
// falls back to the wavefront version.  Checked against the oracle / compiled reference in
// tests (levenshtein batch + every QC decision of the parity suites).
template <typename WORD>
static __device__ __forceinline__ int lane_lev_bp_core(const uint8_t *s, int n, const uint8_t *t, int m)
{
    // straight-line selects only (masks of 0 / ~0): a per-character if-ladder compiles to a chain of
    // exec-mask branches that dominates the loop
    WORD pA = 0, pC = 0, pG = 0, pT = 0;
    uint32_t bad = 0;
    for (int i = 0; i < n; i++) {
        const uint32_t c = s[i];
        const WORD b = (WORD)1 << i;
        const WORD mA = (WORD)0 - (WORD)(c == 'A'), mC = (WORD)0 - (WORD)(c == 'C');
        const WORD mG = (WORD)0 - (WORD)(c == 'G'), mT = (WORD)0 - (WORD)(c == 'T');
        pA |= b & mA; pC |= b & mC; pG |= b & mG; pT |= b & mT;
        bad |= (uint32_t)((mA | mC | mG | mT) == 0);
    }
    WORD VP = ~(WORD)0, VN = 0, D0 = 0, PMold = 0;
    int dist = n;
    const int topbit = n - 1;
    for (int j = 0; j < m; j++) {
        const uint32_t c = t[j];
        const WORD mA = (WORD)0 - (WORD)(c == 'A'), mC = (WORD)0 - (WORD)(c == 'C');
        const WORD mG = (WORD)0 - (WORD)(c == 'G'), mT = (WORD)0 - (WORD)(c == 'T');
        const WORD PMj = (pA & mA) | (pC & mC) | (pG & mG) | (pT & mT);
        WORD TR = (WORD)((((WORD)~D0) & PMj) << 1) & PMold & ~(WORD)3;
        TR &= (WORD)0 - (WORD)(j >= 2);
        D0 = (WORD)((WORD)((WORD)(PMj & VP) + VP) ^ VP) | PMj | VN;
        D0 |= TR;
        WORD HP = VN | (WORD)~(D0 | VP);
        WORD HN = D0 & VP;
        dist += (int)((HP >> topbit) & 1) - (int)((HN >> topbit) & 1);
        HP = (WORD)(HP << 1) | (WORD)1;
        HN = (WORD)(HN << 1);
        VP = HN | (WORD)~(D0 | HP);
        VN = HP & D0;
        PMold = PMj;
    }
    return bad ? -1 : dist;
}

static __device__ int lane_lev_bp(const uint8_t *s, int n, const uint8_t *t, int m)
{
    if (n == 0) return m;
    if (m == 0) return n;
    if (n > m) { const uint8_t *x = s; s = t; t = x; int y = n; n = m; m = y; }
    if (n > 64) return -1;
    // one 32-bit column word when the shorter string fits: half the VALU work of the 64-bit form
    return (n <= 32) ? lane_lev_bp_core<uint32_t>(s, n, t, m) : lane_lev_bp_core<uint64_t>(s, n, t, m);
}

// getStringSimilarity for one pair per lane; *fallback set when the bit-parallel form does not apply
static __device__ float lane_similarity(const uint8_t *s1, int n, const uint8_t *s2, int m, bool &fallback)
{
    fallback = false;
    float max_length = (float)(n > m ? n : m);
    if (n < 3 || m < 3) return 0.0f;
    int d = lane_lev_bp(s1, n, s2, m);
    if (d < 0) { fallback = true; return 0.0f; }
    float edit_distance = (float)d;
    return (float)(1.0 - (double)(edit_distance / max_length));
}

// std::string::substr(pos, n) length: throws if pos > size
static __device__ __forceinline__ bool substr_len(int L, uint32_t pos, uint32_t n, uint32_t &len)
{
    if (pos > (uint32_t)L) return false;
    uint32_t avail = (uint32_t)L - pos;
    len = n < avail ? n : avail;
    return true;
}

// getStringSimilarity (PatternMatcher.cpp:197-204) of read[s0, s0+n) and read[t0, t0+m), both <= 64 bases, on the packed words:
// ten independent LDS reads, then the bit-parallel distance runs on registers (ln_lev_regs: the lane kernel's form)
template <typename WORD> static __device__ __forceinline__ int ln_lev_regs(uint64_t s_lo, uint64_t s_hi, int n, uint64_t t_lo, uint64_t t_hi, int m, int stop_at);
static __device__ __forceinline__ void wv_load128(const uint32_t *words, int start, uint64_t &lo, uint64_t &hi);
// narrow: every lane of the wave holds a pair whose shorter string has at most 32 bases (or none) — the distance then runs on 32-bit
// words, half the instructions (the caller's ballot: the choice is the wave's, not the lane's)
static __device__ float packed_similarity(const uint32_t *words, uint32_t s0, int n, uint32_t t0, int m, bool narrow = false)
{
    const float max_length = (float)(n > m ? n : m);
    if (n < 3 || m < 3) return 0.0f;
    if (n > m) { const uint32_t x = s0; s0 = t0; t0 = x; const int y = n; n = m; m = y; }
    uint64_t sl, sh, tl, th;
    wv_load128(words, (int)s0, sl, sh); wv_load128(words, (int)t0, tl, th);
    const float edit_distance = narrow ? (float)ln_lev_regs<uint32_t>(sl, sh, n, tl, th, m, -1) : (float)ln_lev_regs<uint64_t>(sl, sh, n, tl, th, m, -1);
    return (float)(1.0 - (double)(edit_distance / max_length));
}

// qcFoundRepeats, libcrispr.cpp:869-1029.  1 pass / 0 fail / -1 reference would throw.
// Internal spacer i (getAllSpacerStrings, ReadHolder.cpp:199-239) = seq[ss[2i+1]+1, ss[2i+2]).
static __device__ int qc_found_repeats(RH &h, int minSpacerLength, int maxSpacerLength, int lane, uint32_t dbg = 0)
{
    rh_ascii(h, lane);
    if (h.err == 6) return -3;
    const int num_repeats = h.nss / 2;
    if (num_repeats < 2) return -1;
    uint32_t rep_len;
    const uint32_t rep_start = uni(h.ss[0]);
    if (!substr_len(h.L, rep_start, uni(h.ss[1]) - rep_start + 1, rep_len)) return -1;
    const uint8_t *repeat = h.seq + rep_start;
    if (is_low_complexity(repeat, (int)rep_len, lane)) return 0;

    bool is_short = (2 > (num_repeats - 1));
    if (!is_short) {
        float ave_spacer_to_spacer_len_difference = 0.0f;
        float ave_repeat_to_spacer_len_difference = 0.0f;
        float ave_spacer_to_spacer_difference = 0.0f;
        float ave_repeat_to_spacer_difference = 0.0f;
        int min_spacer_length = 10000000;
        int max_spacer_length = 0;
        int num_compared = 0;
        const int nsp = num_repeats - 1;
        // The reference walks the spacers once, summing similarities and length differences, and only then tests: spacer lengths,
        // the two similarity averages, the two length-difference averages — every failed test is the same `return false`
        // (:773-867).  The length tests need no edit distance, so they are taken FIRST here: a candidate that fails one of them
        // (the common reason: every other repeat of an array missed, spacers of 100 bases) leaves without a single DP.  One read
        // of BASELINE configs[3] spent 5.3 M cycles — 2.3 ms, the tail of the whole launch — on 54 wavefront DPs of 77 x 77 cells
        // for a candidate whose longest spacer decides it.  (A throw in getAllSpacerStrings comes before any of it: -1.)
        // Terms are fetched by the lanes — lane i: spacer i — and summed lane by lane in the reference's order (float addition
        // does not associate): a term is a v_readlane instead of dependent LDS reads per spacer.
        for (int c0 = 0; c0 < nsp; c0 += WAVE) {
            const int i = c0 + lane;
            const bool vi = i < nsp, vc = i + 1 < nsp;
            uint32_t cur_len = 0, nxt_len = 0;
            bool bad = false;
            if (vi) { const uint32_t cs = h.ss[2 * i + 1] + 1; bad = !substr_len(h.L, cs, h.ss[2 * i + 2] - cs, cur_len); }
            if (vc) { const uint32_t ns = h.ss[2 * i + 3] + 1; bad = bad || !substr_len(h.L, ns, h.ss[2 * i + 4] - ns, nxt_len); }
            if (__ballot(bad)) return -1;
            const float t_ssl = (float)cur_len - (float)nxt_len, t_rsl = (float)rep_len - (float)cur_len;
            int mn = vi ? (int)cur_len : 10000000, mx = vi ? (int)cur_len : 0;
            for (int off = 32; off > 0; off >>= 1) { mn = min(mn, __shfl_xor(mn, off)); mx = max(mx, __shfl_xor(mx, off)); }
            if (mn < min_spacer_length) min_spacer_length = mn;
            if (mx > max_spacer_length) max_spacer_length = mx;
            const int i_end = min(c0 + WAVE, nsp - 1);                 // comparisons i = c0 .. i_end - 1 (i + 1 < nsp)
            for (int ii = c0; ii < i_end; ii++) {
                const int src = ii - c0;
                num_compared++;
                ave_spacer_to_spacer_len_difference += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t_ssl), src));
                ave_repeat_to_spacer_len_difference += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t_rsl), src));
            }
        }
        // num_compared == nsp-1 >= 1 here (the reference's num_compared == 0 branch needs <2 spacers)
        ave_spacer_to_spacer_len_difference /= (float)num_compared;
        ave_spacer_to_spacer_len_difference = fabsf(ave_spacer_to_spacer_len_difference);
        ave_repeat_to_spacer_len_difference /= (float)num_compared;
        ave_repeat_to_spacer_len_difference = fabsf(ave_repeat_to_spacer_len_difference);
        if (min_spacer_length < minSpacerLength) return 0;                 // testSpacerLength :773-800
        if (max_spacer_length > maxSpacerLength) return 0;
        if ((int)ave_spacer_to_spacer_len_difference > 12) return 0;       // int parameter: truncation (:836)
        if ((int)ave_repeat_to_spacer_len_difference > 30) return 0;       // (:853)
        // all 2*(nsp-1) similarities are independent: one pair per lane (bit-parallel DP on the packed words), the float
        // sums below then consume them in the reference's order.  pair 2i = (repeat, spacer_i),
        // pair 2i+1 = (spacer_i, spacer_i+1)   (:921-950)
        // A round of 64 lanes takes pairs of ONE kind — first the (repeat, spacer) pairs, then the (spacer, spacer) pairs: the
        // two kinds side by side in one round were two executions of the distance loop with half the lanes each, and the rounds of
        // the first kind share their shorter string's bound — a repeat of up to 32 bases puts the whole round on 32-bit words.
        // (QC was 46 % of the full kernel's cycles on the array reads of BASELINE configs[3]; NOTES r06.)
        const int per_kind = nsp - 1, rounds_per_kind = (per_kind + WAVE - 1) / WAVE;
        for (int rd = 0; rd < 2 * rounds_per_kind; rd++) {
            const int kind = rd >= rounds_per_kind ? 1 : 0;
            const int i_l = (rd - kind * rounds_per_kind) * WAVE + lane;
            const bool valid = i_l < per_kind;
            const int q = valid ? 2 * i_l + kind : 0;
            const int i = valid ? i_l : 0;
            uint32_t a_start = h.ss[2 * i + 1] + 1, a_len = 0, b_start = h.ss[2 * i + 3] + 1, b_len = 0;
            bool bad = !substr_len(h.L, a_start, h.ss[2 * i + 2] - a_start, a_len);
            bad = bad || !substr_len(h.L, b_start, h.ss[2 * i + 4] - b_start, b_len);
            if (__ballot(valid && bad)) return -1;
            bool fb = false;
            float sim = 0.0f;
            if (valid && dbg != 5) {
                // (packed reads, strings of at most 64 bases — always, with anything like the default bounds: on the 2-bit words)
                const bool pk = h.words && rep_len <= 64u && a_len <= 64u && b_len <= 64u;
                const uint32_t x0 = kind ? a_start : rep_start, xn = kind ? a_len : rep_len, y0 = kind ? b_start : a_start, yn = kind ? b_len : a_len;
                const bool narrow = __ballot(valid && pk && min(xn, yn) > 32u) == 0ull;      // (wave-uniform)
                if (pk) sim = packed_similarity(h.words, x0, (int)xn, y0, (int)yn, narrow);
                else sim = lane_similarity(h.seq + x0, (int)xn, h.seq + y0, (int)yn, fb);
            }
            if (valid) h.sims[q] = sim;
            uint64_t fbmask = __ballot(valid && fb);
            while (fbmask) {                              // rare: > 64-long or non-ACGT shorter string
                const int src = __ffsll((unsigned long long)fbmask) - 1;
                fbmask &= fbmask - 1;
                const int qq = 2 * ((rd - kind * rounds_per_kind) * WAVE + src) + kind, ii = qq >> 1;      // (lane src's pair)
                uint32_t as = h.ss[2 * ii + 1] + 1, al = 0, bs = h.ss[2 * ii + 3] + 1, bl = 0;
                (void)substr_len(h.L, as, h.ss[2 * ii + 2] - as, al);
                (void)substr_len(h.L, bs, h.ss[2 * ii + 4] - bs, bl);
                if ((int)std::max(std::max(al, bl), rep_len) + 8 > h.row_cap) return -3;
                float sw = (qq & 1) ? wave_similarity(h.seq + as, (int)al, h.seq + bs, (int)bl, h.rowA, h.rowB, lane)
                                    : wave_similarity(repeat, (int)rep_len, h.seq + as, (int)al, h.rowA, h.rowB, lane);
                if (lane == 0) h.sims[qq] = sw;
            }
        }
        wave_sync();
        for (int c0 = 0; c0 + 1 < nsp; c0 += WAVE) {
            const int i = c0 + lane;
            const bool vc = i + 1 < nsp;
            const float t_rs = vc ? h.sims[2 * i] : 0.0f, t_ss = vc ? h.sims[2 * i + 1] : 0.0f;
            const int i_end = min(c0 + WAVE, nsp - 1);
            for (int ii = c0; ii < i_end; ii++) {
                const int src = ii - c0;
                ave_repeat_to_spacer_difference += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t_rs), src));
                float ss_diff = 0;
                ss_diff += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t_ss), src));
                ave_spacer_to_spacer_difference += ss_diff;
            }
        }
        wave_sync();
        ave_spacer_to_spacer_difference /= (float)num_compared;
        ave_repeat_to_spacer_difference /= (float)num_compared;
        if ((double)ave_spacer_to_spacer_difference > 0.82) return 0;      // :802-834
        if ((double)ave_repeat_to_spacer_difference > 0.82) return 0;
    }
    if (is_short) {
        // spacerStringAt(0), ReadHolder.cpp:102-147 — one base short
        uint32_t s = h.ss[1] + 1;
        uint32_t e = h.ss[2] - 1;
        uint32_t sp_len;
        if (!substr_len(h.L, s, e - s, sp_len)) return -1;
        if ((int)sp_len < minSpacerLength) return 0;
        if ((int)sp_len > maxSpacerLength) return 0;
        bool fb = false;
        const bool pk = h.words && rep_len <= 64u && sp_len <= 64u;
        float similarity = (dbg == 5) ? 0.0f : (pk ? packed_similarity(h.words, rep_start, (int)rep_len, s, (int)sp_len)
                                                   : lane_similarity(repeat, (int)rep_len, h.seq + s, (int)sp_len, fb));     // same pair in every lane
        if (fb && (int)std::max(sp_len, rep_len) + 8 > h.row_cap) return -3;
        if (fb) similarity = wave_similarity(repeat, (int)rep_len, h.seq + s, (int)sp_len, h.rowA, h.rowB, lane);
        if ((double)similarity > 0.82) return 0;
        int dlen = (int)sp_len - (int)rep_len;
        if (dlen < 0) dlen = -dlen;
        if (dlen > 30) return 0;
    }
    return 1;
}

// ------------------------------------------------------------------------------------
// Packed reads in the wave kernel: scanRight, extendPreRepeat and the QC's edit distances on the 2-bit words in LDS — no
// ASCII window, no per-base LDS round trips.  One long read costs the wave (phase counters, CRASS_SURV_PROF, 1 M x 10 kbp):
// an array read 234 k cycles, of which the QC 128 k (two dependent LDS byte reads per DP column and lane), the scanRight chain
// 52 k (38 finds x {two LDS round trips + a ballot + two fenced list appends}), the extension 31 k (a lane per COLUMN walks the
// 36 repeats, two dependent LDS reads each); a read without an array 27 k, 4.3 k of them a two-repeat extension.  The forms
// below keep the reference's decisions and arithmetic (libcrispr.cpp:170-263, 520-772, 869-1029) and change what a lane holds.
// ------------------------------------------------------------------------------------
template <typename WORD> static __device__ __forceinline__ int ln_lev_regs(uint64_t s_lo, uint64_t s_hi, int n, uint64_t t_lo, uint64_t t_hi, int m, int stop_at);
static __device__ __forceinline__ int ln_run_up(uint64_t x0, uint64_t x1, int n);
static __device__ __forceinline__ int ln_run_down(uint64_t x0, uint64_t x1, int n);

// bases [start, start + 64) of the read as two 64-bit words (base `start` in bits 0-1); start >= 0.  Words past the read are
// whatever the LDS holds there: every consumer masks by its own length.
static __device__ __forceinline__ void wv_load128(const uint32_t *words, int start, uint64_t &lo, uint64_t &hi)
{
    const int wi = start >> 4;
    const uint32_t sh = (uint32_t)(start & 15) * 2u;
    const uint32_t a0 = words[wi], a1 = words[wi + 1], a2 = words[wi + 2], a3 = words[wi + 3], a4 = words[wi + 4];
    const uint32_t y0 = __builtin_amdgcn_alignbit(a1, a0, sh), y1 = __builtin_amdgcn_alignbit(a2, a1, sh);
    const uint32_t y2 = __builtin_amdgcn_alignbit(a3, a2, sh), y3 = __builtin_amdgcn_alignbit(a4, a3, sh);
    lo = (uint64_t)y0 | ((uint64_t)y1 << 32); hi = (uint64_t)y2 | ((uint64_t)y3 << 32);
}
// ... the 64 bases that END before `end` (bases [end - 64, end)), base end - 64 in bits 0-1; positions before the read's
// first base read as 0 (the callers never count them)
static __device__ __forceinline__ void wv_load128_before(const uint32_t *words, int end, uint64_t &lo, uint64_t &hi)
{
    const int start = end - 64;
    if (start >= 0) { wv_load128(words, start, lo, hi); return; }
    uint64_t l, h2;
    wv_load128(words, 0, l, h2);
    const int s = -2 * start;                           // bit positions to move up: 2 .. 128
    if (s >= 128) { lo = 0; hi = 0; }
    else if (s >= 64) { lo = 0; hi = s == 64 ? l : (l << (s - 64)); }
    else { hi = (h2 << s) | (l >> (64 - s)); lo = l << s; }
}
static __device__ __forceinline__ uint32_t wv_base(uint64_t lo, uint64_t hi, int i)     // base i of a 128-bit piece, 0 <= i < 64
{
    return (uint32_t)((i < 32 ? lo >> (2 * i) : hi >> (2 * (i - 32))) & 3ull);
}

// extendPreRepeat (libcrispr.cpp:520-772) on the packed words.  Two repeats: the closed form of the lane kernel (ln_extend2:
// cut_off = 2, a column passes iff both copies agree, so the extensions are runs of equal bases).  3 .. 64 repeats: a LANE PER
// REPEAT holds the 64 bases right of its window (then left of it) in registers and a column's vote is four ballots — the
// columns are taken in the reference's order and stop at the first that fails.  Returns false when it does not apply (more than
// 64 repeats: the column-parallel form below takes those).
static __device__ bool extend_pre_repeat_packed(RH &h, int searchWindowLength, int minSpacerLength, int lane)
{
    const int nrep = h.nss / 2;
    if (nrep < 2 || nrep > WAVE) return false;
    const int L = h.L, w = searchWindowLength;
    const uint32_t *words = h.words;
    uint32_t right = 0, left = 0;
    if (nrep == 2) {
        const int j = (int)uni(h.ss[0]), p = (int)uni(h.ss[2]);
        const int spacing = p - j;
        int max_right = spacing - minSpacerLength;                  // (unsigned in the reference; spacing >= minSpacer + w here)
        if (max_right > L - (p + w)) max_right = L - (p + w);       // the second copy's column must lie inside the read
        int rr = 0;
        while (rr < max_right) {
            uint64_t a0, a1, b0, b1;
            wv_load128(words, j + w + rr, a0, a1); wv_load128(words, p + w + rr, b0, b1);
            const int n = max_right - rr < 64 ? max_right - rr : 64;
            const int r = uni(ln_run_up(a0 ^ b0, a1 ^ b1, n));
            rr += r;
            if (r < n) break;
        }
        const int len_r = w + rr;
        int max_left = spacing - len_r;
        if (max_left < 0) max_left = 0;
        if (max_left > j) max_left = j;                             // the first copy's column must lie inside the read
        int ll = 0;
        while (ll < max_left) {
            const int n = max_left - ll < 64 ? max_left - ll : 64;
            uint64_t a0, a1, b0, b1;
            wv_load128(words, j - ll - n, a0, a1); wv_load128(words, p - ll - n, b0, b1);
            const int r = uni(ln_run_down(a0 ^ b0, a1 ^ b1, n));
            ll += r;
            if (r < n) break;
        }
        right = (uint32_t)rr; left = (uint32_t)ll;
    } else {
        const int cut_off = nrep - 1;                               // (max(2, nrep - 1) with nrep >= 3)
        const uint32_t end_index = (uint32_t)h.nss;
        const uint32_t first_start = uni(h.ss[0]), last_start = uni(h.ss[h.nss - 2]);
        const uint32_t seqlen = (uint32_t)L;
        const bool mine = lane < nrep;
        const uint32_t my_start = mine ? h.ss[2 * lane] : 0u;
        uint32_t shortest = 0xFFFFFFFFu;
        if (mine && lane > 0) shortest = (uint32_t)((int)my_start - (int)h.ss[2 * lane - 2]);
        for (int off = 32; off > 0; off >>= 1) shortest = min(shortest, (uint32_t)__shfl_xor((int)shortest, off));
        shortest = uni(shortest);
        const uint32_t max_right = shortest - (uint32_t)minSpacerLength;
        const int T = (int)seqlen - (int)last_start - w;           // >= 0
        bool stop = false;
        for (uint32_t e0 = 0; e0 < max_right && !stop; e0 += 64u) {
            uint64_t lo = 0, hi = 0;
            if (mine) wv_load128(words, (int)(my_start + (uint32_t)w + e0), lo, hi);
            const uint32_t e1 = min(e0 + 64u, max_right);
            for (uint32_t e = e0; e < e1; e++) {
                int drops = (int)e - T + 1;
                if (drops < 0) drops = 0;
                const int n_part = nrep - drops;                    // DR_index_end / 2 of this iteration (:614-616)
                const uint32_t pos = my_start + (uint32_t)w + e;
                const bool valid = mine && lane < n_part && pos < seqlen;     // (starts ascend: behind the first repeat off the read nobody votes, :624-627)
                const uint32_t b = wv_base(lo, hi, (int)(e - e0));
                const int cA = __popcll(__ballot(valid && b == 0u)), cC = __popcll(__ballot(valid && b == 1u));
                const int cG = __popcll(__ballot(valid && b == 2u)), cT = __popcll(__ballot(valid && b == 3u));
                if ((cA >= cut_off) || (cC >= cut_off) || (cG >= cut_off) || (cT >= cut_off)) right++;
                else { stop = true; break; }
            }
        }
        const int replen_r = w + (int)right;
        const int test_for_negative = (int)(shortest - (uint32_t)replen_r);
        const uint32_t max_left = (test_for_negative >= 0) ? (uint32_t)test_for_negative : 0u;
        stop = false;
        for (uint32_t e0 = 0; e0 < max_left && !stop; e0 += 64u) {
            uint64_t lo = 0, hi = 0;
            if (mine) wv_load128_before(words, (int)my_start - (int)e0, lo, hi);      // bases [my_start - e0 - 64, my_start - e0)
            const uint32_t e1 = min(e0 + 64u, max_left);
            for (uint32_t e = e0; e < e1; e++) {
                int drops = (int)e - (int)first_start + 1;          // iterations i <= e with firstStart - i <= 0 (:700-704)
                if (drops < 0) drops = 0;
                const int idx = (int)my_start - (int)e - 1;
                const bool valid = mine && lane >= drops && idx >= 0 && idx < L;
                const uint32_t b = wv_base(lo, hi, 63 - (int)(e - e0));
                const int cA = __popcll(__ballot(valid && b == 0u)), cC = __popcll(__ballot(valid && b == 1u));
                const int cG = __popcll(__ballot(valid && b == 2u)), cT = __popcll(__ballot(valid && b == 3u));
                if ((cA >= cut_off) || (cC >= cut_off) || (cG >= cut_off) || (cT >= cut_off)) left++;
                else { stop = true; break; }
            }
        }
        (void)end_index;
    }
    h.replen = w + (int)right + (int)left;
    wave_sync();
    for (int r = 2 * lane; r + 1 < h.nss; r += 2 * WAVE) {
        uint32_t a = h.ss[r], b = h.ss[r + 1];
        a = (a < left) ? 0 : a - left;
        b = (b + right >= (uint32_t)L) ? (uint32_t)L - 1 : b + right;
        h.ss[r] = a; h.ss[r + 1] = b;
    }
    wave_sync();
    return true;
}

// scanRight's chain (libcrispr.cpp:170-263) once a third repeat exists: instead of one wave-wide find per link — each a pair of
// LDS round trips, a ballot and a fenced append — the wave marks every position of the next 4 096 bases that holds the window's
// w-mer (lane l: the 64 positions from base + 64 l, one bit each) and then follows the chain on those masks: a link is two
// lane reads and a find-first-set.  Entered with last / second-last repeat starts; appends to the start/stop list like rh_add.
static __device__ void scan_right_masks(RH &h, int pat, uint32_t pattern_length, uint32_t minSpacerLength, uint32_t scanRange,
                                        uint32_t last_repeat_index, uint32_t second_last_repeat_index, int lane)
{
    const uint32_t read_length = (uint32_t)h.L;
    const uint32_t sj = uni(lds_code(h.words, (uint32_t)pat, h.cmask));
    last_repeat_index = uni(last_repeat_index); second_last_repeat_index = uni(second_last_repeat_index);
    uint32_t repeat_spacing = last_repeat_index - second_last_repeat_index;
    uint32_t base = 0xFFFFFFFFu;                        // first position the masks cover (a multiple of 64); none yet
    uint64_t mask = 0;
    int nss = h.nss;
    for (;;) {
        // (the chain's state is the same in every lane; said once per link it stays on the scalar unit)
        last_repeat_index = uni(last_repeat_index); repeat_spacing = uni(repeat_spacing); base = uni(base); nss = uni(nss);
        const int candidate_repeat_index = (int)(last_repeat_index + repeat_spacing);
        uint32_t begin_search = (uint32_t)candidate_repeat_index - scanRange;
        uint32_t end_search = (uint32_t)candidate_repeat_index + pattern_length + scanRange;
        const uint32_t scanRightMinBegin = last_repeat_index + pattern_length + minSpacerLength;
        if (begin_search < scanRightMinBegin) begin_search = scanRightMinBegin;
        if (begin_search > read_length - 1) break;
        if (end_search > read_length) end_search = read_length;
        if (begin_search >= end_search) break;
        if (end_search - begin_search < pattern_length) break;          // bmpSearch: pattern longer than the text => -1 => the chain ends
        const uint32_t last_p = end_search - pattern_length;            // the window's last start
        if (base == 0xFFFFFFFFu || begin_search < base || last_p >= base + 4096u) {
            base = begin_search & ~63u;
            const uint32_t p0 = base + 64u * (uint32_t)lane;
            uint64_t m = 0;
            if (p0 + pattern_length <= read_length) {
                const uint32_t wi = p0 >> 4;
                uint32_t a[5];
#pragma unroll
                for (int i = 0; i < 5; i++) a[i] = h.words[wi + i];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint64_t v = ((uint64_t)a[q + 1] << 32) | a[q];
                    uint32_t bits = 0;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        // (w <= 9: a window of 18 bits starting at bit 2 i <= 30 ends inside the 64-bit piece)
                        const uint32_t c = (uint32_t)(v >> (2 * i)) & h.cmask;
                        bits |= (uint32_t)(c == sj) << i;
                    }
                    m |= (uint64_t)bits << (16 * q);
                }
                // positions whose window would pass the read's end hold no match
                const uint32_t n_ok = read_length - pattern_length - p0 + 1u;      // >= 1 here
                if (n_ok < 64u) m &= (1ull << n_ok) - 1ull;
            }
            mask = m;
        }
        const uint32_t lo_i = begin_search - base, hi_i = last_p - base;      // inclusive bit range, < 4096, at most 57 wide
        const int k0 = (int)(lo_i >> 6), k1 = (int)(hi_i >> 6);
        const uint32_t mlo0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mask, k0), mhi0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mask >> 32), k0);
        uint64_t m0 = (((uint64_t)mhi0 << 32) | mlo0) & (~0ull << (lo_i & 63u));
        if (k1 == k0) m0 &= (hi_i & 63u) == 63u ? ~0ull : ((2ull << (hi_i & 63u)) - 1ull);
        int position = -1;
        if (m0) position = (int)(base + 64u * (uint32_t)k0) + (__ffsll((unsigned long long)m0) - 1);
        else if (k1 > k0) {
            const uint32_t mlo1 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mask, k1), mhi1 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mask >> 32), k1);
            const uint64_t m1 = (((uint64_t)mhi1 << 32) | mlo1) & ((hi_i & 63u) == 63u ? ~0ull : ((2ull << (hi_i & 63u)) - 1ull));
            if (m1) position = (int)(base + 64u * (uint32_t)k1) + (__ffsll((unsigned long long)m1) - 1);
        }
        if (position < 0) break;
        const uint32_t found = (uint32_t)position;
        if (nss + 2 > h.cap) { h.err = 2; break; }                      // (rh_add)
        uint32_t stop = found + pattern_length - 1;
        if (stop >= read_length) stop = read_length - 1;
        if (lane == 0) { h.ss[nss] = found; h.ss[nss + 1] = stop; }
        nss += 2;
        second_last_repeat_index = last_repeat_index;
        last_repeat_index = found;
        repeat_spacing = last_repeat_index - second_last_repeat_index;
        if (repeat_spacing < (minSpacerLength + pattern_length)) break;
    }
    h.nss = nss;
    wave_sync();
}

// searchCore, libcrispr.cpp:265-395.  1 found / 0 not / <0 error
// Hint bits of residue class rho for the read in LDS (h.words), hint words [first_word, end): lanes over hint words, OR-ed into
// the read's hint bitmap (classes occupy disjoint bit positions).  Called by search_core when the walk moves to a class whose
// bits are not there yet from that point on.
template <bool RANGE>
static __device__ __attribute__((noinline)) void wave_hints_class(const RH &h, uint32_t rho, uint32_t first_word, uint64_t *l_hint, int lane, int D0, int D1)
{
    const uint32_t nw = ((uint32_t)h.L + 15u) >> 4, nh = ((uint32_t)h.L + 63u) >> 6;
    for (uint32_t base = first_word; base < nh; base += WAVE) {
        const uint32_t t = base + (uint32_t)lane;
        if (t < nh) {
            uint32_t w[13];
#pragma unroll
            for (int i = 0; i < 13; i++) { const uint32_t wi = 4u * t + (uint32_t)i; w[i] = wi < nw ? h.words[wi] : 0u; }
            l_hint[t] |= hint_bits_any<RANGE>(w, rho, D0, D1);
        }
    }
    wave_sync();
}

// pos_hint (long reads, LDS): the read's per-position hint bits — the lattice class from k_hint_positions; cls_from[rho] (LDS):
// first position from which class rho's bits are in pos_hint (0xFFFFFFFF: not at all)
static __device__ int search_core(RH &h, const DevParams &o, uint32_t seed_hint, int lane, uint64_t *pos_hint = nullptr, uint32_t *cls_from = nullptr,
                                  uint32_t j_start = 0)      // j_start: the light walk has done the iterations before it (all no-ops or rejected candidates)
{
    const uint32_t seq_length = (uint32_t)h.L;
    const uint32_t skips = o.skips;
    int searchEnd = (int)(seq_length - o.lowDR - o.lowSp - o.window - 1);
    if (searchEnd < 0) return 0;
    h.nss = 0;
    // seed_hint (from the bit-parallel filter): bit i clear => lattice seed j = i*skips provably has
    // no hit, so its iteration is a no-op in the reference (no start/stops, numRepeats 0) and can be
    // skipped.  Only valid while j is still on the lattice, i.e. before the first `j = back()-1`.
    bool on_lattice = j_start == 0;
    uint32_t lattice_i = 0;
    for (uint32_t j = j_start; j <= (uint32_t)searchEnd; j = j + skips) {
        j = uni(j);
        PROF_CNT(h, PF_N_ITER);
        if (pos_hint) {
            PROF_T0(h);
            // per-position hints: a clear bit makes this iteration a no-op in the reference, on or off the lattice.  The walk
            // stays on one residue class mod 8 until a rejected candidate moves it (j = back() - 1 below); the bits of the class
            // it moves to are computed here, from this position to the end of the read (the walk only moves forward)
            if (cls_from && j < uni(cls_from[j & 7u])) {     // wave-uniform; never true on the lattice class
                const uint32_t first_word = j >> 6;
                { const unsigned long long _ph0 = h.lprof ? (unsigned long long)__builtin_readcyclecounter() : 0ull;
                { const int D0 = (int)(o.lowDR + o.lowSp), D1 = (int)(o.highDR + o.highSp);
                  if (D0 == 49 && D1 == 97) wave_hints_class<false>(h, j & 7u, first_word, pos_hint, lane, D0, D1);
                  else wave_hints_class<true>(h, j & 7u, first_word, pos_hint, lane, D0, D1); }
                if (h.lprof && lane == 0) { h.lprof[PF_HINTS] += (unsigned long long)__builtin_readcyclecounter() - _ph0; h.lprof[PF_N_SWITCH] += 1ull; } }
                if (lane == 0) cls_from[j & 7u] = first_word << 6;
                wave_sync();
            }
            // With skips == 8 the next candidate in the same 64-bit word is one ffs away.
            const uint64_t wbits = uni64(pos_hint[j >> 6]) >> (j & 63u);
            if (!(wbits & 1ull)) {
                if (skips == 8) {
                    const uint64_t m = wbits & 0x0101010101010101ull;        // positions j, j+8, ... inside this word
                    if (m) j += (uint32_t)(__ffsll((unsigned long long)m) - 1) - 8;    // the loop adds 8: lands on it
                    else {
                        // nothing left in this word: the lanes look at the next 64 hint words at once (a random 10 kbp read
                        // has one or two words with a candidate out of 157; walking them one dependent load at a time was
                        // most of this kernel's time on long reads).  64 is a multiple of the stride, so the candidates of
                        // every later word sit at bit offsets (j mod 8) + 8k.
                        const uint64_t resmask = 0x0101010101010101ull << (j & 7u);
                        const uint32_t n_words = ((uint32_t)searchEnd >> 6) + 1u;
                        uint32_t next_j = 0xFFFFFFFFu;
                        for (uint32_t w0 = (j >> 6) + 1u; w0 < n_words; w0 += 64u) {
                            const uint32_t w = w0 + (uint32_t)lane;
                            const uint64_t v = w < n_words ? (pos_hint[w] & resmask) : 0ull;
                            const uint64_t any = __ballot(v != 0ull);
                            if (any) {
                                const uint32_t fw = w0 + (uint32_t)(__ffsll((unsigned long long)any) - 1);
                                const uint64_t fv = uni64(pos_hint[fw]) & resmask;      // (wave-uniform reload)
                                next_j = fw * 64u + (uint32_t)(__ffsll((unsigned long long)fv) - 1);
                                break;
                            }
                        }
                        if (next_j == 0xFFFFFFFFu) { PROF_ADD(h, PF_LOOP); break; }          // no candidate up to searchEnd: the loop ends
                        j = next_j - 8u;                                            // the loop adds 8: lands on it
                    }
                }
                PROF_ADD(h, PF_LOOP);
                continue;
            }
            PROF_ADD(h, PF_LOOP);
        } else if (on_lattice) {
            const uint32_t li = lattice_i++;
            if (li < 32 && !((seed_hint >> li) & 1u)) continue;
        }
        uint32_t beginSearch = j + o.lowDR + o.lowSp;
        uint32_t endSearch = j + o.highDR + o.highSp + o.window;
        if (endSearch >= seq_length) endSearch = seq_length - 1;
        if (endSearch < beginSearch) endSearch = beginSearch;
        if (beginSearch > seq_length) return -1;                       // substr would throw
        int pos;
        { PROF_T0(h);
        pos = rh_find(h, (int)beginSearch, (int)endSearch, (int)j, (int)o.window, lane);
        PROF_ADD(h, PF_FIND); PROF_CNT(h, PF_N_CAND); }
        if (o.debug_stop == 2) pos = -1;
        if (pos >= 0) {
            PROF_T0(h);
            rh_add(h, j, j + o.window - 1, lane);
            rh_add(h, (uint32_t)pos, (uint32_t)pos + o.window - 1, lane);
            if (h.err) return -2;
            scan_right(h, (int)j, o.window, o.lowSp, 24, lane);
            if (h.err) return -2;
            PROF_ADD(h, PF_SCAN);
        }
        if ((uint32_t)(h.nss / 2) >= o.minRepeats) {
            uint32_t actual_repeat_length;
            { PROF_T0(h);
            actual_repeat_length = extend_pre_repeat(h, (int)o.window, (int)o.lowSp, lane);
            PROF_ADD(h, PF_EXTEND); }
            if (h.err == 6) return -3;
            if (o.debug_stop != 3 && (actual_repeat_length >= o.lowDR) && (actual_repeat_length <= o.highDR)) {
                PROF_T0(h);
                int qc = qc_found_repeats(h, (int)o.lowSp, (int)o.highSp, lane, o.debug_stop);
                PROF_ADD(h, PF_QC); PROF_CNT(h, PF_N_QC);
                if (qc == -3) return -3;
                if (qc < 0) return -1;
                if (qc) return 1;
            }
            j = uni(h.ss[h.nss - 1]) - 1;
            on_lattice = false;
        }
        h.nss = 0;
        wave_sync();
    }
    return 0;
}

// ReadHolder::DRLowLexi + reverseStartStops (ReadHolder.cpp:513-591, 321-380).  Writes the
// low-lexi DR to dr_out (global), mirrors ss in LDS if the read is flipped; returns dr length
// (<0 error) and the RH_WasLowLexi flag.
static __device__ int dr_low_lexi(RH &h, char *dr_out, int dr_stride, int &was_low_lexi, int lane)
{
    rh_ascii(h, lane);
    const int num_repeats = h.nss / 2;
    int pick;
    if (num_repeats == 1) pick = 0;
    else if (num_repeats == 2) {
        if (h.ss[0] == 0) pick = 2;
        else if (h.ss[h.nss - 1] == (uint32_t)h.L) pick = 0;          // dead branch, kept (:541)
        else {
            int lenA = (int)(h.ss[1] - h.ss[0]);
            int lenB = (int)(h.ss[3] - h.ss[2]);
            pick = (lenA > lenB) ? 0 : 2;
        }
    } else pick = 2;
    uint32_t dlen;
    const uint32_t dr_start = uni(h.ss[pick]);
    if (!substr_len(h.L, dr_start, uni(h.ss[pick + 1]) - dr_start + 1, dlen)) return -1;
    const uint8_t *dr = h.seq + dr_start;
    // tmp_dr < rev_comp ?  (std::string operator<, unsigned bytes; equal => not less => flip)
    int less = 0;
    for (int i0 = 0; i0 < (int)dlen; i0 += WAVE) {
        int i = i0 + lane;
        uint8_t a = 0, b = 0;
        if (i < (int)dlen) { a = dr[i]; b = c_comp[dr[dlen - 1 - i] & 127]; }
        uint64_t m = __ballot(a != b);
        if (m) {
            int f = __ffsll((unsigned long long)m) - 1;
            int av = __shfl((int)a, f), bv = __shfl((int)b, f);
            less = av < bv;
            break;
        }
    }
    if (less) {
        for (int i = lane; i < dr_stride; i += WAVE) dr_out[i] = (i < (int)dlen) ? (char)dr[i] : (char)0;   // zero padding: slots are bit-reproducible
        was_low_lexi = 1;
    } else {
        for (int i = lane; i < dr_stride; i += WAVE) dr_out[i] = (i < (int)dlen) ? (char)c_comp[dr[dlen - 1 - i] & 127] : (char)0;
        // reverseStartStops: new[k] = L-1 - ss[nss-1-k]
        wave_sync();
        for (int k0 = 0; k0 < h.nss; k0 += WAVE) {       // read everything of a chunk pair-wise before writing
            int k = k0 + lane;
            int mirror = h.nss - 1 - k;
            // swap in place: handle k < mirror pairs only
            if (k < h.nss && k < mirror) {
                uint32_t a = h.ss[k], b = h.ss[mirror];
                h.ss[k] = (uint32_t)h.L - 1 - b;
                h.ss[mirror] = (uint32_t)h.L - 1 - a;
            }
        }
        wave_sync();
        was_low_lexi = 0;
    }
    return (int)dlen;
}

// 16 bases of a packed word as ASCII, four letters per output word: the byte of four codes is placed in both 16-bit
// halves (v_perm), the upper half shifted by 4 (v_pk_lshrrev_b16), `u | u << 6` puts each half's second code into its
// upper byte, and the 2-bit codes select from the letters with a second v_perm: 5 instructions per 4 bases instead of ~20
static __device__ __forceinline__ void word_to_ascii(uint32_t v, uint32_t o[4])
{
    const uint32_t letters = ('A') | ('C' << 8) | ('G' << 16) | ('T' << 24);
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t t = __builtin_amdgcn_perm(0u, v, 0x0C000C00u | ((uint32_t)q << 16) | (uint32_t)q);   // [b, 0, b, 0], b = byte q of v
        us2 u2 = __builtin_bit_cast(us2, t);
        u2.y = (unsigned short)(u2.y >> 4);
        const uint32_t u = __builtin_bit_cast(uint32_t, u2);
        const uint32_t sel = (u | (u << 6)) & 0x03030303u;
        o[q] = __builtin_amdgcn_perm(0u, letters, sel);
    }
}

// words of the read a wave works on NEXT, requested while it searches the current one (wave-per-read kernel, long reads:
// a 10 kbp read is ten dependent rounds of loads per lane otherwise — 5.0 of the kernel's 12.3 ms at 1 M x 10 kbp)
#define SV_PREFETCH_WORDS 12                      // per lane: reads up to 12 * 64 * 16 = 12 288 bases are covered completely
#define SV_PREFETCH_HINTS (SV_PREFETCH_WORDS / 4)  // 64-bit hint words per lane for the same reads (one bit per position)
// (the words travel 16 bytes per lane and load: a read is word-aligned, not 16-byte aligned, and the hardware takes a
// dword-aligned global_load_dwordx4 — a quarter of the load and LDS-store instructions of the one-word form)
typedef uint32_t sv_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
#define SV_PREFETCH_VEC (SV_PREFETCH_WORDS / 4)
struct ReadPrefetch {
    uint4 v[SV_PREFETCH_VEC];                     // lane's 16-byte groups lane, lane + 64, ...: words 4 g .. 4 g + 3
    uint64_t hw[SV_PREFETCH_HINTS];
    uint64_t r;                                   // the read these words belong to (~0: none)
};
// group gi (words 4 gi .. 4 gi + 3) of a read of nw words at g; words past the read are 0 and never touched in memory
static __device__ __forceinline__ uint4 sv_load_group(const uint32_t *g, int gi, int nw)
{
    uint4 v; v.x = v.y = v.z = v.w = 0u;
    const int b = 4 * gi;
    if (b + 3 < nw) { const sv_u32x4 t = *reinterpret_cast<const sv_u32x4 *>(g + b); v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w; }
    else if (b < nw) { v.x = g[b]; if (b + 1 < nw) v.y = g[b + 1]; if (b + 2 < nw) v.z = g[b + 2]; }
    return v;
}
static __device__ __forceinline__ void prefetch_read(const DevReads &R, uint64_t r, int lane, ReadPrefetch &pf)
{
    const uint32_t *g = R.packed + rd_word_off(R, r);
    const int L = (int)uni(rd_len(R, r));
    const int nw = (L + 15) >> 4;
#pragma unroll
    for (int i = 0; i < SV_PREFETCH_VEC; i++) pf.v[i] = sv_load_group(g, lane + i * WAVE, nw);
    if (R.pos_hint) {
        const uint64_t *ph = R.pos_hint + rd_hint_off(R, r);
        const int nh = (L + 63) >> 6;
#pragma unroll
        for (int i = 0; i < SV_PREFETCH_HINTS; i++) {
            const int wi = lane + i * WAVE;
            pf.hw[i] = wi < nh ? ph[wi] : 0ull;
        }
    }
    pf.r = r;
}
// the read's per-position hint words into LDS: search_core looks at one of them per seed candidate and scans them for the
// next candidate — as global loads these were some 25 dependent round trips per 10 kbp read
static __device__ void load_hints_to_lds(const DevReads &R, uint64_t r, int L, uint64_t *l_hint, int lane, const ReadPrefetch &pf)
{
    const int nh = (L + 63) >> 6;
    int first = lane;
    if (pf.r == r) {
#pragma unroll
        for (int i = 0; i < SV_PREFETCH_HINTS; i++) {
            const int wi = lane + i * WAVE;
            if (wi < nh) l_hint[wi] = pf.hw[i];
        }
        first = lane + SV_PREFETCH_HINTS * WAVE;
        if (nh <= SV_PREFETCH_HINTS * WAVE) return;      // (wave-uniform: nothing left, and no look-up of the read's offset)
    }
    const uint64_t *ph = R.pos_hint + rd_hint_off(R, r);
    for (int wi = first; wi < nh; wi += WAVE) l_hint[wi] = ph[wi];
}

static __device__ void load_read_to_lds(const DevReads &R, uint64_t r, uint8_t *seq, uint32_t *words, int L, int lane,
                                        const ReadPrefetch *pf = nullptr)
{
    const uint32_t *g = R.packed + rd_word_off(R, r);
    const int nw = (L + 15) >> 4;
    const int ng = (nw + 4) >> 2;                 // groups up to and including the one that holds word nw (= 0: lds_code reads one word past the last)
    const bool have = pf && pf->r == r;           // wave-uniform
    uint4 *w4 = reinterpret_cast<uint4 *>(words);   // (the words' LDS region is 16-byte aligned and a multiple of four words long)
    int first = lane;
    if (have) {
#pragma unroll
        for (int i = 0; i < SV_PREFETCH_VEC; i++) {
            const int gi = lane + i * WAVE;
            if (gi < ng) w4[gi] = pf->v[i];
        }
        first = lane + SV_PREFETCH_VEC * WAVE;
    }
    for (int gi = first; gi < ng; gi += WAVE) w4[gi] = sv_load_group(g, gi, nw);
    (void)seq;                                          // (the ASCII copy: region by region, when a byte-wise consumer is entered — rh_ascii)
}

#define PROF_READ_DONE(cat_, tot_) do { \
            wave_sync(); \
            if (lane == 0) { h.lprof[PF_TOTAL] = (tot_); h.lprof[PF_READS] = 1ull; } \
            wave_sync(); \
            unsigned long long *ws_ = h.lprof + PF_SLOTS + (cat_) * PF_SLOTS; \
            if (lane < PF_SLOTS && lane != PF_MAX) ws_[lane] += h.lprof[lane]; \
            if (lane == PF_MAX && (tot_) > ws_[PF_MAX]) ws_[PF_MAX] = (tot_); \
            if (lane == 0) { int b_ = 63 - __clzll((long long)((tot_) | 1ull)) - 8; b_ = b_ < 0 ? 0 : (b_ >= PF_BINS ? PF_BINS - 1 : b_); h.lprof[5 * PF_SLOTS + (cat_) * PF_BINS + b_] += 1ull; } \
            wave_sync(); } while (0)
#define PROF_WAVE_FLUSH() do { if (h.lprof) { wave_sync(); \
            if (lane == 0) { atomicAdd(P.prof + 190, (unsigned long long)__builtin_readcyclecounter() - wave_c0); atomicAdd(P.prof + 191, (unsigned long long)__builtin_amdgcn_s_memrealtime() - wave_r0); \
                             atomicMax(P.prof + 189, (unsigned long long)__builtin_amdgcn_s_memrealtime()); atomicAdd(P.prof + 186, 1ull); \
                             if (blockIdx.x < 16384) P.prof[193 + 2 * blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_memrealtime(); } \
            for (int q_ = lane; q_ < 4 * PF_SLOTS + 4 * PF_BINS; q_ += WAVE) { \
                const unsigned long long v_ = h.lprof[PF_SLOTS + q_]; \
                if (v_) { if (q_ < 4 * PF_SLOTS && (q_ % PF_SLOTS) == PF_MAX) atomicMax(P.prof + q_, v_); else atomicAdd(P.prof + q_, v_); } } } } while (0)
template <bool EXC>
// (two waves per SIMD asked for, i.e. up to 256 VGPRs: with one wave per block and 14-58 KB of LDS per block the LDS decides
// the residency — and the next read's prefetched words did not fit the 128 registers of a 4-wave target without spilling)
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_survivor(DevReads R, DevParams P, const uint64_t *surv_idx,
                                                   const uint32_t *d_n_surv, uint64_t n_max, SurvOut *out,
                                                   char *dr_chars, uint32_t dr_stride, uint32_t *ss_pool,
                                                   uint32_t ss_pool_cap, uint32_t *d_ss_used,
                                                   uint8_t *found_flag, const uint32_t *seed_hint, SurvLds lds, int punt_only, uint64_t slot_base, uint64_t slot_total,
                                                   const uint32_t *punt_list, const uint32_t *d_punt_n, uint32_t *redo_list)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sv_lds[];
    const int lane = threadIdx.x;
    RH h;
    h.seq = sv_lds;
    h.seq_buf = sv_lds; h.seq_win = EXC ? 0 : (int)lds.seq_window;
    h.ss = reinterpret_cast<uint32_t *>(sv_lds + lds.seq_bytes);
    h.cap = (int)lds.ss_cap;
    h.rowA = reinterpret_cast<uint16_t *>(h.ss + lds.ss_cap);
    h.rowB = h.rowA + lds.row_elems;
    h.row_cap = (int)lds.row_elems;
    uint32_t *l_words = reinterpret_cast<uint32_t *>(h.rowB + lds.row_elems);
    h.words = EXC ? nullptr : l_words;
    h.sims = reinterpret_cast<float *>(l_words + lds.words_cap);
    uint64_t *l_hint = reinterpret_cast<uint64_t *>(h.sims + lds.ss_cap);
    h.cmask = (1u << (2 * P.window)) - 1u;
    // diagnostics: the launch was given PF_SLOTS x 8 more bytes of LDS behind the layout (launch_survivor)
    h.lprof = P.prof ? reinterpret_cast<unsigned long long *>(sv_lds + ((lds.total_bytes + 7u) & ~7u)) : nullptr;
    if (h.lprof) { for (int q = lane; q < PF_LDS_WORDS; q += WAVE) h.lprof[q] = 0ull; wave_sync(); }
    // (the wave's lifetime on both clocks: s_memtime counts shader cycles, s_memrealtime a constant 100 MHz — their ratio is the
    // clock the launch really ran at; slots 190 / 191 of the counters)
    const unsigned long long wave_c0 = h.lprof ? (unsigned long long)__builtin_readcyclecounter() : 0ull;
    const unsigned long long wave_r0 = h.lprof ? (unsigned long long)__builtin_amdgcn_s_memrealtime() : 0ull;
    if (h.lprof && lane == 0) { atomicMin(P.prof + 188, wave_r0); atomicMax(P.prof + 187, wave_r0); }      // first and last wave START
    if (h.lprof && lane == 0 && blockIdx.x < 16384) P.prof[192 + 2 * blockIdx.x] = wave_r0;
    // EXC with punt_only == 5: exception reads that sit in the survivor list (slot s, read surv_idx[s])
    uint64_t n_surv = (EXC && punt_only != 5) ? R.n_exc : (uint64_t)(*d_n_surv);
    if (n_surv > n_max) n_surv = n_max;
    // punt mode: only the reads an earlier launch handed over (err == punt_only: 4 from the lane kernel, 6 = row buffer).  Each wave looks at 64 slots at once and
    // then walks the (rare) flagged ones, instead of every wave polling its slots one dependent load at a time.
    // ... or, with punt_list, exactly the slots on that list (k_long_light's hand-overs, in any order: every result is addressed by
    // its slot), dealt out one at a time over a grid of resident waves — 64 slots per wave left a wave anything from none to a dozen
    // array reads and the launch took twice its share (3.2 ms for 51 k reads at 10 kbp)
    // (the lane kernel's punts, counted by it: usually none — the scan of every slot's error byte for them was 10-17 us of the step)
    if (!EXC && punt_only == 4 && !punt_list && d_punt_n && *d_punt_n == 0u) { PROF_WAVE_FLUSH(); return; }
    uint64_t punt_base = (uint64_t)blockIdx.x * WAVE, punt_mask = 0;
    uint64_t lp = blockIdx.x;
    const uint64_t n_list = punt_list ? (uint64_t)(*d_punt_n) : 0;
    ReadPrefetch pf;
    pf.r = ~0ull;
    uint64_t next_s = ~0ull, next_r = 0;
    for (uint64_t s = blockIdx.x;; ) {
        if (punt_only && punt_list) {
            if (lp >= n_list) { PROF_WAVE_FLUSH(); return; }
            s = uni(punt_list[lp]);
            lp += gridDim.x;
        } else if (punt_only) {
            while (punt_mask == 0) {
                if (punt_base >= n_surv) { PROF_WAVE_FLUSH(); return; }
                const uint64_t q = punt_base + lane;
                punt_mask = __ballot(q < n_surv && out[q].err == (uint8_t)punt_only);
                if (punt_mask == 0) punt_base += (uint64_t)gridDim.x * WAVE;
            }
            const int b = __ffsll((unsigned long long)punt_mask) - 1;
            punt_mask &= punt_mask - 1;
            s = punt_base + b;
            if (punt_mask == 0) punt_base += (uint64_t)gridDim.x * WAVE;
        } else if (s >= n_surv) { PROF_WAVE_FLUSH(); return; }
        uint64_t r;
        int L;
        wave_sync();
        const unsigned long long prof_t0 = h.lprof ? (unsigned long long)__builtin_readcyclecounter() : 0ull;
        if (h.lprof) { if (lane < PF_SLOTS) h.lprof[lane] = 0ull; wave_sync(); }

        if (EXC) {
            uint64_t e = s;
            if (punt_only == 5) {
                r = uni64(surv_idx[s]);
                uint64_t lo = 0, hi = R.n_exc;                  // exc_read[] is ascending: first entry >= r is r itself
                while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (R.exc_read[mid] < r) lo = mid + 1; else hi = mid; }
                e = lo;
            } else r = uni64(R.exc_read[s]);
            uint64_t o0 = uni64(R.exc_off[e]);
            L = (int)uni((uint32_t)(R.exc_off[e + 1] - o0));
            for (int i = lane; i < L; i += WAVE) h.seq[i] = R.exc_bytes[o0 + i];
        } else {
            // (surv_idx == nullptr: the list is 0, 1, 2, ... — a long-read set without exception reads, every read survives)
            r = (next_s == s) ? next_r : (surv_idx ? uni64(surv_idx[s]) : s + slot_base);    // (the prefetch below already looked it up)
            if (!punt_only && R.n_exc && rd_is_exc(R, r)) {     // left to the exception pass
                if (lane == 0) { SurvOut x; x.found = 0; x.n_ss = 0; x.repeat_len = 0; x.ss_off = 0; x.dr_len = 0; x.low_lexi = 0; x.err = 5; out[s] = x; }
                s += gridDim.x;
                continue;
            }
            L = (int)uni(rd_len(R, r));
            // long reads: a read without a single hinted LATTICE seed (39 % of random 10 kbp reads) never leaves the lattice and
            // searchCore returns false without having looked at a base — neither does this wave (no 10 KB of LDS to fill)
            bool no_seed = false;
            if (R.pos_hint && !punt_only) {
                const int nh = (L + 63) >> 6;
                bool any = false;
                if (pf.r == r) {
#pragma unroll
                    for (int i = 0; i < SV_PREFETCH_HINTS; i++) any |= (lane + i * WAVE < nh) && pf.hw[i] != 0ull;
                    if (nh > SV_PREFETCH_HINTS * WAVE) any = true;                 // (longer than the prefetch covers: walk it)
                } else {
                    const uint64_t *gh = R.pos_hint + rd_hint_off(R, r);
                    for (int wi = lane; wi < nh; wi += WAVE) any |= gh[wi] != 0ull;
                }
                no_seed = __ballot(any) == 0ull;
            }
            if (!no_seed) {
                load_read_to_lds(R, r, h.seq, l_words, L, lane, &pf);
                if (R.pos_hint) load_hints_to_lds(R, r, L, l_hint, lane, pf);
            }
            pf.r = ~0ull;
            if (!punt_only && s + gridDim.x < n_surv) {         // the next read of this wave: its words travel during the search
                const uint64_t r2 = surv_idx ? uni64(surv_idx[s + gridDim.x]) : s + gridDim.x + slot_base;
                next_s = s + gridDim.x; next_r = r2;
                if (!R.n_exc || !rd_is_exc(R, r2)) prefetch_read(R, r2, lane, pf);
            } else if (punt_only && punt_list && lp < n_list) { // (list mode: the slot after this one is known as well)
                const uint64_t s2 = uni(punt_list[lp]);
                const uint64_t r2 = surv_idx ? uni64(surv_idx[s2]) : s2 + slot_base;
                next_s = s2; next_r = r2;
                prefetch_read(R, r2, lane, pf);
            }
            if (no_seed) {
                if (lane == 0) { SurvOut x; x.found = 0; x.n_ss = 0; x.repeat_len = 0; x.ss_off = 0; x.dr_len = 0; x.low_lexi = 0; x.err = 0; out[s] = x; }
                if (h.lprof) { const unsigned long long tot = (unsigned long long)__builtin_readcyclecounter() - prof_t0; PROF_READ_DONE(0, tot); }
                s += gridDim.x;
                continue;
            }
        }
        wave_sync();
        if (h.lprof && lane == 0) h.lprof[PF_STAGE] = (unsigned long long)__builtin_readcyclecounter() - prof_t0;
        h.L = L; h.nss = 0; h.replen = 0; h.err = 0;
        h.asc_lo = 0; h.asc_hi = 0; h.asc_pad = (int)(P.highDR + P.highSp) + 32; h.seq = h.seq_buf;
        const uint32_t hint = (!EXC && seed_hint) ? seed_hint[r] : 0xFFFFFFFFu;
        uint64_t *ph = (!EXC && R.pos_hint) ? l_hint : nullptr;            // (staged in LDS above)
        uint32_t *cls_from = reinterpret_cast<uint32_t *>(l_hint + lds.hint_words - 4);      // (the last four hint slots: 8 x uint32)
        if (ph && lane < 8) cls_from[lane] = (lane == 0 || R.hint_all) ? 0u : 0xFFFFFFFFu;      // (hint_all: every class is there from the start)
        wave_sync();
        // (a read the light walk handed over: on from the seed it stopped at — SurvOut::repeat_len of the slot, k_long_light)
        const uint32_t j_start = (!EXC && punt_only == 7) ? uni(out[s].repeat_len) : 0u;
        int f = (P.debug_stop == 1) ? 0 : search_core(h, P, hint, lane, ph, ph ? cls_from : nullptr, j_start);
        const unsigned long long prof_t1 = h.lprof ? (unsigned long long)__builtin_readcyclecounter() : 0ull;
        SurvOut o;
        o.found = 0; o.n_ss = 0; o.repeat_len = 0; o.ss_off = 0; o.dr_len = 0; o.low_lexi = 0; o.err = 0;
        // (-3: Levenshtein rows too short / the ASCII window too small in this launch's LDS layout; -2 in a layout with a capped
        // start/stop list: the same — the launch with the full layout decides)
        if (f == -3 || (f == -2 && lds.ss_cap < lds.ss_slot)) o.err = 6;
        else if (f < 0) o.err = (f == -2) ? 2 : 1;
        if (f == 1 && P.debug_stop != 4) {
            int low = 0;
            int dlen = dr_low_lexi(h, dr_chars + s * (uint64_t)dr_stride, (int)dr_stride, low, lane);
            if (dlen < 0 || dlen > (int)dr_stride) o.err = 1;
            else {
                // start/stop pool: a fixed slot per survivor when the pool is large enough (short reads),
                // otherwise bump allocation (one contended atomic per found read: long reads only)
                // (slot_base / slot_total: this launch covers the slots [slot_base, slot_base + n) of slot_total — the walk of a
                // long-read set runs slice by slice, `out` and `dr_chars` already point at the slice, the pool is shared)
                uint32_t off = 0;
                if ((slot_total ? slot_total : n_surv) * lds.ss_slot <= ss_pool_cap) off = (uint32_t)(s + slot_base) * lds.ss_slot;
                else {
                    if (lane == 0) off = atomicAdd(d_ss_used, (uint32_t)h.nss);
                    off = (uint32_t)__shfl((int)off, 0);
                }
                if ((uint64_t)off + (uint64_t)h.nss > ss_pool_cap) o.err = 3;
                else {
                    for (int k = lane; k < h.nss; k += WAVE) ss_pool[off + k] = h.ss[k];
                    o.found = 1; o.n_ss = (uint32_t)h.nss; o.repeat_len = (uint32_t)h.replen;
                    o.ss_off = off; o.dr_len = (uint16_t)dlen; o.low_lexi = (uint8_t)low;
                    if (lane == 0) found_flag[rd_header_id(R, r)] = 1;      // readsFound[header] = true (:138)
                }
            }
        }
        if (lane == 0) out[s] = o;
        // (a read this launch's layout cannot hold goes on the list of the launch with the full layout: that launch then visits
        // exactly those slots instead of polling every slot's error byte — 68 us for 1 M slots of which none is flagged)
        if (redo_list && o.err == 6 && lane == 0) redo_list[1u + atomicAdd(redo_list, 1u)] = (uint32_t)s;
        if (h.lprof && lane == 0) h.lprof[PF_OUT] = (unsigned long long)__builtin_readcyclecounter() - prof_t1;
        if (h.lprof) {
            // category 1: walked, nothing found; 2: found; 3: handed over / error (0: skipped, no hinted seed)
            const unsigned long long tot = (unsigned long long)__builtin_readcyclecounter() - prof_t0;
            const int cat = o.err ? 3 : (o.found ? 2 : 1);
            PROF_READ_DONE(cat, tot);
            if (lane == 0) atomicMax(P.prof + 185, (tot << 24) | (unsigned long long)((s + slot_base) & 0xFFFFFFull));      // the slowest read's slot
        }
        if (!punt_only) s += gridDim.x;
    }
}

// ------------------------------------------------------------------------------------
// Long reads, the LIGHT walk.  19 reads in 20 of a long-read set hold no array: their searchCore (libcrispr.cpp:265-395) is a
// walk over ~1 250 seeds of which one or two have a chance copy in range, a two-repeat candidate that the extension rejects, a
// move to another residue class (:390) and more walking.  The full wave kernel above spends 34 k cycles on such a read — all of
// it dependent latency (LDS round trips, fenced list appends, vector arithmetic on wave-uniform values) at three waves per SIMD,
// because it carries the QC, the Levenshtein rows, the ASCII window and a start/stop list for the one read in 20 that needs them.
// This kernel is that walk alone: the read's packed words in LDS (2.5 KB at 10 kbp, nothing else), the CURRENT residue class's
// hint words in registers (lane l: words base + l, base + l + 64, ...; the lattice class from k_hint_positions, any other
// computed here 64 words at a time as the walk reaches them), a candidate's two repeats in scalar registers, the two-repeat
// extension in closed form.  Whatever needs more — a third repeat (an array), a candidate due for qcFoundRepeats, a read beyond
// 12 288 bases, an error path — is HANDED OVER (err = 7) to k_survivor<false>, which redoes that read from its first base with
// the reference's full control flow.  A read that is not handed over is decided here: not found.
// ------------------------------------------------------------------------------------
#define LL_HW 3                                   // hint words per lane: 192 per read = 12 288 bases
#define LL_VEC 3                                  // 16-byte word groups per lane: 768 words
struct LightPrefetch { uint4 v[LL_VEC]; uint64_t hw[LL_HW]; };

static __device__ __forceinline__ void ll_prefetch(const DevReads &R, uint64_t r, int lane, LightPrefetch &pf)
{
    const uint32_t *g = R.packed + rd_word_off(R, r);
    const int L = (int)uni(rd_len(R, r));
    const int nw = (L + 15) >> 4, nh = (L + 63) >> 6;
#pragma unroll
    for (int i = 0; i < LL_VEC; i++) pf.v[i] = sv_load_group(g, lane + i * WAVE, nw);
    const uint64_t *ph = R.pos_hint + rd_hint_off(R, r);
#pragma unroll
    for (int i = 0; i < LL_HW; i++) { const int wi = lane + i * WAVE; pf.hw[i] = wi < nh ? ph[wi] : 0ull; }
}

// first hinted position >= j among the hint words held in registers (word cur_base + lane + 64 i in cur[i], the first `done`
// words valid), 0xFFFFFFFF if none
static __device__ __forceinline__ uint32_t ll_next_hinted(const uint64_t (&cur)[LL_HW], uint32_t cur_base, uint32_t done, uint32_t j, int lane)
{
    const uint32_t jw = j >> 6;
#pragma unroll
    for (int i = 0; i < LL_HW; i++) {
        if (64u * (uint32_t)i >= done) break;                                   // (wave-uniform)
        if (cur_base + 64u * (uint32_t)i + 63u < jw) continue;
        const uint32_t t = cur_base + 64u * (uint32_t)i + (uint32_t)lane;
        uint64_t m = (64u * (uint32_t)i + (uint32_t)lane < done) ? cur[i] : 0ull;
        if (t < jw) m = 0ull;
        else if (t == jw) m &= ~0ull << (j & 63u);
        const uint64_t b = __ballot(m != 0ull);
        if (b) {
            const int l0 = __ffsll((unsigned long long)b) - 1;
            const uint64_t mm = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(m >> 32), l0) << 32) |
                                (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)m, l0);
            return (cur_base + 64u * (uint32_t)i + (uint32_t)l0) * 64u + (uint32_t)(__ffsll((unsigned long long)mm) - 1);
        }
    }
    return 0xFFFFFFFFu;
}

template <bool RANGE>
// surv_idx (optional): slot s holds read surv_idx[s] — the dense path's survivor list (reads of 513 .. 2 048 bases, whose
// survivors the lane kernel does not take: until round 6 every one of them was a walk of the full wave kernel); nullptr: s + slot_base
__global__ __launch_bounds__(WAVE) void k_long_light(DevReads R, DevParams P, const uint32_t *d_n, uint64_t n_max, SurvOut *out, uint64_t slot_base,
                                                     uint32_t *punt_list, uint32_t *d_punt_n, const uint64_t *surv_idx)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t ll_words[];
    const int lane = threadIdx.x;
    uint64_t n = (uint64_t)(*d_n);
    if (n > n_max) n = n_max;
    const uint32_t cmask = (1u << (2 * P.window)) - 1u;
    const int w = (int)P.window;
    LightPrefetch pf;
    auto read_of = [&](uint64_t sl) -> uint64_t { return surv_idx ? uni64(surv_idx[sl]) : sl + slot_base; };
    if (blockIdx.x < n) ll_prefetch(R, read_of(blockIdx.x), lane, pf);
    for (uint64_t s = blockIdx.x; s < n; s += gridDim.x) {
        const uint64_t r = read_of(s);
        if (surv_idx && R.n_exc && rd_is_exc(R, r)) {                             // an exception read in the list: left to the exception pass
            if (lane == 0) { SurvOut x; x.found = 0; x.n_ss = 0; x.repeat_len = 0; x.ss_off = 0; x.dr_len = 0; x.low_lexi = 0; x.err = 5; out[s] = x; }
            if (s + gridDim.x < n) ll_prefetch(R, read_of(s + gridDim.x), lane, pf);
            continue;
        }
        const int L = (int)uni(rd_len(R, r));
        const int nw = (L + 15) >> 4, nh = (L + 63) >> 6;
        uint64_t cur[LL_HW];
        bool any = false;
#pragma unroll
        for (int i = 0; i < LL_HW; i++) { cur[i] = pf.hw[i]; any |= cur[i] != 0ull; }
        const bool too_long = nh > LL_HW * WAVE;
        const bool seeds = __ballot(any) != 0ull;
        uint8_t verdict = 0;                                                     // 0: searchCore returns false; 7: handed over
        uint32_t punt_j = 0;                                                     // ... from this seed on (the iterations before it are done)
        if (too_long) verdict = 7;
        wave_sync();                                                             // (the previous read's LDS accesses are done)
        if (seeds && !too_long) {
            const int ng = (nw + 4) >> 2;                                        // groups up to the one that holds word nw (= 0)
            uint4 *w4 = reinterpret_cast<uint4 *>(ll_words);
#pragma unroll
            for (int i = 0; i < LL_VEC; i++) { const int gi = lane + i * WAVE; if (gi < ng) w4[gi] = pf.v[i]; }
        }
        if (s + gridDim.x < n) ll_prefetch(R, read_of(s + gridDim.x), lane, pf);  // the next read travels during this one's walk
        if (seeds && !too_long) {
            wave_sync();
            const uint32_t seq_length = (uint32_t)L;
            const int searchEnd = (int)(seq_length - P.lowDR - P.lowSp - P.window - 1);
            uint32_t j = 0, rho = 0, cur_base = 0, done = (uint32_t)nh;          // the lattice class: every word is there
            const uint32_t n_hw = searchEnd >= 0 ? ((uint32_t)searchEnd >> 6) + 1u : 0u;      // hint words that hold a seed
            while (searchEnd >= 0 && j <= (uint32_t)searchEnd) {
                j = uni(j); rho = uni(rho); cur_base = uni(cur_base); done = uni(done);
                // the class's hint words are computed 64 at a time, when the walk gets there (a class is left again after
                // 4 000 bases one time in three: most of a class's words would never be looked at)
                uint32_t p = ll_next_hinted(cur, cur_base, done, j, lane);
                while (p == 0xFFFFFFFFu && cur_base + done < n_hw && done < 64u * LL_HW) {
                    const uint32_t t = cur_base + done + (uint32_t)lane;
                    uint64_t bits = 0ull;
                    if (t < (uint32_t)nh) {
                        uint32_t ww[13];
#pragma unroll
                        for (int i = 0; i < 13; i++) { const uint32_t wi = 4u * t + (uint32_t)i; ww[i] = wi < (uint32_t)nw ? ll_words[wi] : 0u; }
                        bits = hint_bits_any<RANGE>(ww, rho, (int)(P.lowDR + P.lowSp), (int)(P.highDR + P.highSp));
                    }
                    const uint32_t round = done >> 6;
#pragma unroll
                    for (int i = 0; i < LL_HW; i++) if ((uint32_t)i == round) cur[i] = bits;
                    done += 64u;
                    p = ll_next_hinted(cur, cur_base, done, j, lane);
                }
                if (p == 0xFFFFFFFFu || p > (uint32_t)searchEnd) break;         // no seed left: false
                j = p;
                uint32_t beginSearch = j + P.lowDR + P.lowSp;
                uint32_t endSearch = j + P.highDR + P.highSp + P.window;
                if (endSearch >= seq_length) endSearch = seq_length - 1;
                if (endSearch < beginSearch) endSearch = beginSearch;
                if (beginSearch > seq_length) { verdict = 7; punt_j = j; break; }   // (the reference throws here: the full kernel reports it)
                const int pos = uni(wave_find_packed(ll_words, cmask, (int)beginSearch, (int)endSearch, (int)j, w, lane));
                if (pos < 0) { j += P.skips; continue; }                         // (the hints are a superset)
                // two repeats at j and pos.  scanRight's first link (libcrispr.cpp:170-263): a third repeat means an array
                {
                    const uint32_t last = (uint32_t)pos, spacing = last - j;
                    const int cand = (int)(last + spacing);
                    uint32_t b3 = (uint32_t)cand - 24u, e3 = (uint32_t)cand + (uint32_t)w + 24u;
                    const uint32_t minb = last + (uint32_t)w + P.lowSp;
                    if (b3 < minb) b3 = minb;
                    if (b3 <= seq_length - 1) {
                        if (e3 > seq_length) e3 = seq_length;
                        if (b3 < e3 && wave_find_packed(ll_words, cmask, (int)b3, (int)e3, (int)j, w, lane) >= 0) { verdict = 7; punt_j = j; break; }
                    }
                }
                if (2u < P.minRepeats) { j += P.skips; continue; }               // (-n 3 and up: two repeats are not a candidate)
                // extendPreRepeat for two repeats, closed form (extend_pre_repeat_packed / ln_extend2)
                uint32_t right, left;
                {
                    const int jj = (int)j, pp = pos, spacing = pp - jj;
                    int max_right = spacing - (int)P.lowSp;
                    if (max_right > L - (pp + w)) max_right = L - (pp + w);
                    int rr = 0;
                    while (rr < max_right) {
                        uint64_t a0, a1, b0, b1;
                        wv_load128(ll_words, jj + w + rr, a0, a1); wv_load128(ll_words, pp + w + rr, b0, b1);
                        const int nn = max_right - rr < 64 ? max_right - rr : 64;
                        const int q = uni(ln_run_up(a0 ^ b0, a1 ^ b1, nn));
                        rr += q;
                        if (q < nn) break;
                    }
                    int max_left = spacing - (w + rr);
                    if (max_left < 0) max_left = 0;
                    if (max_left > jj) max_left = jj;
                    int ll = 0;
                    while (ll < max_left) {
                        const int nn = max_left - ll < 64 ? max_left - ll : 64;
                        uint64_t a0, a1, b0, b1;
                        wv_load128(ll_words, jj - ll - nn, a0, a1); wv_load128(ll_words, pp - ll - nn, b0, b1);
                        const int q = uni(ln_run_down(a0 ^ b0, a1 ^ b1, nn));
                        ll += q;
                        if (q < nn) break;
                    }
                    right = (uint32_t)rr; left = (uint32_t)ll;
                }
                const uint32_t replen = (uint32_t)w + right + left;
                if (replen >= P.lowDR && replen <= P.highDR) { verdict = 7; punt_j = j; break; }          // due for qcFoundRepeats
                // rejected: on behind the last repeat's (extended, clamped) end (:390)
                uint32_t last_end = (uint32_t)pos + (uint32_t)w - 1u;
                if (last_end >= seq_length) last_end = seq_length - 1;                       // (startStopsAdd's clamp)
                last_end = (last_end + right >= seq_length) ? seq_length - 1 : last_end + right;
                j = last_end - 1u + P.skips;
                if ((j & 7u) != rho) { rho = j & 7u; cur_base = j >> 6; done = 0u; }
            }
        }
        if (lane == 0) {
            SurvOut x; x.found = 0; x.n_ss = 0; x.repeat_len = punt_j; x.ss_off = 0; x.dr_len = 0; x.low_lexi = 0; x.err = verdict; out[s] = x;
            if (verdict == 7) punt_list[atomicAdd(d_punt_n, 1u)] = (uint32_t)(s + slot_base);     // (the slot in the whole set's numbering)
        }
    }
}

// ---- the light walk for ANY window and seed lattice (-w / -d on long reads) ----
// The hint forms above know one residue class mod 8 at a time; with another lattice (skips = lowDR - 2 w + 1) or window the
// every-position form of the scan (k_filter_fast_any) gives ONE bit per base — "the w-mer here has a copy D0 .. D1 further on" —
// for whatever lattice the walk is on, so there is no class to switch: the walking wave computes the bits of the next 64 hint words
// when it gets there (lane = hint word: 4 packed words + halo, 11 instructions per word and shift), finds the next set bit at or
// behind j and takes it if it lies on the lattice that starts at j.  No hint kernel, no hint array.  Everything else is
// k_long_light.  (1 M x 10 kbp with -d 20 -D 40: 512 ms on the un-hinted wave kernel.)
static __device__ __forceinline__ uint64_t hint_bits_every(const uint32_t (&w)[13], int wn, int D0, int D1)
{
    const int s1 = min(1, wn - 1), s2 = min(2, wn - 1 - s1), s3 = min(4, wn - 1 - s1 - s2), s4 = wn - 1 - s1 - s2 - s3;
    uint32_t nz[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
#pragma unroll
    for (int q = 0; q < 8; q++) {
        if (16 * q + 15 < D0 || 16 * q > D1) continue;
        const int sb_lo = max(0, D0 - 16 * q), sb_hi = min(15, D1 - 16 * q);
        for (int sb = sb_lo; sb <= sb_hi; sb++) {
            const uint32_t sh = (uint32_t)(2 * sb);
            uint32_t z[5];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const uint32_t x = __builtin_amdgcn_alignbit(w[k + q + 1], w[k + q], sh) ^ w[k];
                z[k] = x | (x >> 1);
            }
#pragma unroll
            for (int k = 0; k < 4; k++) z[k] |= __builtin_amdgcn_alignbit(z[k + 1], z[k], (uint32_t)(2 * s1));
            if (s2 > 0) {
                z[4] |= z[4] >> (2 * s1);
#pragma unroll
                for (int k = 0; k < 4; k++) z[k] |= __builtin_amdgcn_alignbit(z[k + 1], z[k], (uint32_t)(2 * s2));
            }
            if (s3 > 0) {
                z[4] |= z[4] >> (2 * s2);
#pragma unroll
                for (int k = 0; k < 4; k++) z[k] |= __builtin_amdgcn_alignbit(z[k + 1], z[k], (uint32_t)(2 * s3));
            }
            if (s4 > 0) {
                z[4] |= z[4] >> (2 * s3);
#pragma unroll
                for (int k = 0; k < 4; k++) z[k] |= __builtin_amdgcn_alignbit(z[k + 1], z[k], (uint32_t)(2 * s4));
            }
#pragma unroll
            for (int k = 0; k < 4; k++) nz[k] &= z[k];
        }
    }
    // bit 2p of ~nz[k] = a copy exists for position 16 k + p: the even bits of the four words, packed
    auto even16 = [](uint32_t x) {
        x &= 0x55555555u;
        x = (x | (x >> 1)) & 0x33333333u;
        x = (x | (x >> 2)) & 0x0F0F0F0Fu;
        x = (x | (x >> 4)) & 0x00FF00FFu;
        x = (x | (x >> 8)) & 0x0000FFFFu;
        return x;
    };
    const uint32_t lo = even16(~nz[0]) | (even16(~nz[1]) << 16), hi = even16(~nz[2]) | (even16(~nz[3]) << 16);
    return ((uint64_t)hi << 32) | (uint64_t)lo;
}

// The seed-scan FILTER of reads of 257 .. 2 048 bases (and of sets whose strides differ) under another window or seed lattice
// (-w, -d): no position hints are kept for those (the walks' hint forms know the default lattice), so a tile's bits are computed,
// cut to the lattice positions j = i * skips <= searchEnd, and only the read's bit in the (cleared) filter mask is set.  A superset
// like every filter here; 11 instructions per word and shift for every position — four times the lattice-class kernel, a fifth of
// k_filter_general, which these sets took until now.
__global__ __launch_bounds__(256) void k_hint_filter_any(DevReads R, DevParams P, const uint64_t *hint_off, const uint32_t *blk_read, uint64_t n_words,
                                                         uint64_t *hitmask, uint64_t *hint_bits)
{
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (t >= n_words) return;
    uint64_t r;
    uint32_t tile;
    hint_tile_read(R, hint_off, blk_read, t, n_words, r, tile);
    const uint32_t L = rd_len(R, r);
    const int D0 = (int)(P.lowDR + P.lowSp), D1 = (int)(P.highDR + P.highSp);
    const int last = (int)L - D0 - (int)P.window - 1 - (int)(tile * 64u);        // searchEnd, counted from this tile's first position
    if (last < 0) { hint_bits[t] = 0ull; return; }
    const uint32_t nw = (L + 15) >> 4;
    const uint32_t *g = R.packed + rd_word_off(R, r) + tile * 4u;
    const uint32_t rem = nw - tile * 4u;
    uint32_t w[13];
#pragma unroll
    for (int i = 0; i < 13; i++) w[i] = (uint32_t)i < rem ? g[i] : 0u;
    uint64_t bits = hint_bits_every(w, (int)P.window, D0, D1);
    if (last < 63) bits &= (2ull << last) - 1ull;
    // every position's bit is kept: the survivors' walks step over the positions whose bit is clear, on the lattice and — behind a
    // rejected candidate — off it (DevReads.hint_all)
    hint_bits[t] = bits;
    // the lattice: positions that are multiples of skips
    const uint32_t skips = P.skips;
    if (skips > 1) {
        const uint64_t p0 = (uint64_t)tile * 64u;
        uint32_t b = (uint32_t)((skips - (uint32_t)(p0 % skips)) % skips);
        uint64_t m = 0;
        for (; b < 64u; b += skips) m |= 1ull << b;
        bits &= m;
    }
    if (bits && (P.exc_survive || !rd_is_exc(R, r))) atomicOr(reinterpret_cast<unsigned long long *>(hitmask) + (r >> 6), 1ull << (r & 63u));
}

hipError_t launch_hint_filter_any(const DevReads &R, const DevParams &P, const uint64_t *hint_off, const uint32_t *blk_read, uint64_t n_words,
                                  uint64_t *hitmask, uint64_t *hint_bits, hipStream_t st)
{
    if (P.window < 6 || P.window > 9 || P.skips < 1 || P.lowDR + P.lowSp < 17 || P.highDR + P.highSp > 127 || P.highDR + P.highSp < P.lowDR + P.lowSp) return hipErrorNotSupported;
    if (!n_words) return hipSuccess;
    const uint64_t nb = (n_words + 255) / 256;
    if (nb > 0x7FFFFFFFull) return hipErrorNotSupported;
    CRASS_LAUNCH(k_hint_filter_any, dim3((unsigned)nb), dim3(256), 0, st, R, P, hint_off, blk_read, n_words, hitmask, hint_bits);
    return hipGetLastError();
}

__global__ __launch_bounds__(WAVE) void k_long_light_any(DevReads R, DevParams P, const uint32_t *d_n, uint64_t n_max, SurvOut *out, uint64_t slot_base,
                                                         uint32_t *punt_list, uint32_t *d_punt_n, const uint64_t *surv_idx)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t ll_words[];
    const int lane = threadIdx.x;
    uint64_t n = (uint64_t)(*d_n);
    if (n > n_max) n = n_max;
    const uint32_t cmask = (1u << (2 * P.window)) - 1u;
    const int w = (int)P.window;
    const int D0 = (int)(P.lowDR + P.lowSp), D1 = (int)(P.highDR + P.highSp);
    uint4 pfv[LL_VEC];
    auto prefetch = [&](uint64_t r) {
        const uint32_t *g = R.packed + rd_word_off(R, r);
        const int nw = ((int)uni(rd_len(R, r)) + 15) >> 4;
#pragma unroll
        for (int i = 0; i < LL_VEC; i++) pfv[i] = sv_load_group(g, lane + i * WAVE, nw);
    };
    auto read_of = [&](uint64_t sl) -> uint64_t { return surv_idx ? uni64(surv_idx[sl]) : sl + slot_base; };
    if (blockIdx.x < n) prefetch(read_of(blockIdx.x));
    for (uint64_t s = blockIdx.x; s < n; s += gridDim.x) {
        const uint64_t r = read_of(s);
        if (surv_idx && R.n_exc && rd_is_exc(R, r)) {                             // an exception read in the list: left to the exception pass
            if (lane == 0) { SurvOut x; x.found = 0; x.n_ss = 0; x.repeat_len = 0; x.ss_off = 0; x.dr_len = 0; x.low_lexi = 0; x.err = 5; out[s] = x; }
            if (s + gridDim.x < n) prefetch(read_of(s + gridDim.x));
            continue;
        }
        const int L = (int)uni(rd_len(R, r));
        const int nw = (L + 15) >> 4, nh = (L + 63) >> 6;
        const bool too_long = nh > LL_HW * WAVE;
        uint8_t verdict = too_long ? 7 : 0;
        uint32_t punt_j = 0;
        wave_sync();
        if (!too_long) {
            const int ng = (nw + 4) >> 2;
            uint4 *w4 = reinterpret_cast<uint4 *>(ll_words);
#pragma unroll
            for (int i = 0; i < LL_VEC; i++) { const int gi = lane + i * WAVE; if (gi < ng) w4[gi] = pfv[i]; }
        }
        if (s + gridDim.x < n) prefetch(read_of(s + gridDim.x));
        if (!too_long) {
            wave_sync();
            const uint32_t seq_length = (uint32_t)L;
            const int searchEnd = (int)(seq_length - P.lowDR - P.lowSp - P.window - 1);
            const uint32_t n_hw = searchEnd >= 0 ? ((uint32_t)searchEnd >> 6) + 1u : 0u;
            uint64_t cur[LL_HW] = {};
            const uint64_t *stored = (R.pos_hint && R.hint_all) ? R.pos_hint + rd_hint_off(R, r) : nullptr;
            uint32_t j = 0, done = 0, from = 0;          // `from`: the next position to look at (>= j); seeds are j, j + skips, ...
            while (searchEnd >= 0 && j <= (uint32_t)searchEnd) {
                j = uni(j); done = uni(done); from = uni(from);
                uint32_t p = ll_next_hinted(cur, 0u, done, from, lane);
                while (p == 0xFFFFFFFFu && done < n_hw && done < 64u * LL_HW) {
                    const uint32_t t = done + (uint32_t)lane;
                    uint64_t bits = 0ull;
                    if (t < (uint32_t)nh) {
                        if (stored) bits = stored[t];              // (every position's bit is there already: k_hint_filter_any kept them, DevReads.hint_all)
                        else {
                            uint32_t ww[13];
#pragma unroll
                            for (int i = 0; i < 13; i++) { const uint32_t wi = 4u * t + (uint32_t)i; ww[i] = wi < (uint32_t)nw ? ll_words[wi] : 0u; }
                            bits = hint_bits_every(ww, w, D0, D1);
                        }
                    }
                    const uint32_t round = done >> 6;
#pragma unroll
                    for (int i = 0; i < LL_HW; i++) if ((uint32_t)i == round) cur[i] = bits;
                    done += 64u;
                    p = ll_next_hinted(cur, 0u, done, from, lane);
                }
                if (p == 0xFFFFFFFFu || p > (uint32_t)searchEnd) break;
                if ((p - j) % P.skips != 0u) { from = p + 1u; continue; }       // a copy exists there, but it is no seed of this lattice
                j = p;
                uint32_t beginSearch = j + P.lowDR + P.lowSp;
                uint32_t endSearch = j + P.highDR + P.highSp + P.window;
                if (endSearch >= seq_length) endSearch = seq_length - 1;
                if (endSearch < beginSearch) endSearch = beginSearch;
                if (beginSearch > seq_length) { verdict = 7; punt_j = j; break; }
                const int pos = uni(wave_find_packed(ll_words, cmask, (int)beginSearch, (int)endSearch, (int)j, w, lane));
                if (pos < 0) { j += P.skips; from = j; continue; }
                {
                    const uint32_t last = (uint32_t)pos, spacing = last - j;
                    const int cand = (int)(last + spacing);
                    uint32_t b3 = (uint32_t)cand - 24u, e3 = (uint32_t)cand + (uint32_t)w + 24u;
                    const uint32_t minb = last + (uint32_t)w + P.lowSp;
                    if (b3 < minb) b3 = minb;
                    if (b3 <= seq_length - 1) {
                        if (e3 > seq_length) e3 = seq_length;
                        if (b3 < e3 && wave_find_packed(ll_words, cmask, (int)b3, (int)e3, (int)j, w, lane) >= 0) { verdict = 7; punt_j = j; break; }
                    }
                }
                if (2u < P.minRepeats) { j += P.skips; from = j; continue; }
                uint32_t right, left;
                {
                    const int jj = (int)j, pp = pos, spacing = pp - jj;
                    int max_right = spacing - (int)P.lowSp;
                    if (max_right > L - (pp + w)) max_right = L - (pp + w);
                    int rr = 0;
                    while (rr < max_right) {
                        uint64_t a0, a1, b0, b1;
                        wv_load128(ll_words, jj + w + rr, a0, a1); wv_load128(ll_words, pp + w + rr, b0, b1);
                        const int nn = max_right - rr < 64 ? max_right - rr : 64;
                        const int q = uni(ln_run_up(a0 ^ b0, a1 ^ b1, nn));
                        rr += q;
                        if (q < nn) break;
                    }
                    int max_left = spacing - (w + rr);
                    if (max_left < 0) max_left = 0;
                    if (max_left > jj) max_left = jj;
                    int ll = 0;
                    while (ll < max_left) {
                        const int nn = max_left - ll < 64 ? max_left - ll : 64;
                        uint64_t a0, a1, b0, b1;
                        wv_load128(ll_words, jj - ll - nn, a0, a1); wv_load128(ll_words, pp - ll - nn, b0, b1);
                        const int q = uni(ln_run_down(a0 ^ b0, a1 ^ b1, nn));
                        ll += q;
                        if (q < nn) break;
                    }
                    right = (uint32_t)rr; left = (uint32_t)ll;
                }
                const uint32_t replen = (uint32_t)w + right + left;
                if (replen >= P.lowDR && replen <= P.highDR) { verdict = 7; punt_j = j; break; }
                uint32_t last_end = (uint32_t)pos + (uint32_t)w - 1u;
                if (last_end >= seq_length) last_end = seq_length - 1;
                last_end = (last_end + right >= seq_length) ? seq_length - 1 : last_end + right;
                j = last_end - 1u + P.skips;
                from = j;
            }
        }
        if (lane == 0) {
            SurvOut x; x.found = 0; x.n_ss = 0; x.repeat_len = punt_j; x.ss_off = 0; x.dr_len = 0; x.low_lexi = 0; x.err = verdict; out[s] = x;
            if (verdict == 7) punt_list[atomicAdd(d_punt_n, 1u)] = (uint32_t)(s + slot_base);
        }
    }
}

hipError_t launch_long_light(const DevReads &R, const DevParams &P, const uint32_t *d_n, uint64_t n_max, SurvOut *out, uint64_t slot_base,
                             uint32_t max_len, hipStream_t st, uint32_t *punt_list, uint32_t *d_punt_n, const uint64_t *surv_idx)
{
    if (n_max == 0) return hipSuccess;
    const uint32_t words_cap = (((max_len + 15) / 16 + 2) + 3u) & ~3u;
    const uint32_t lds_bytes = (words_cap + 16u) * 4u;                           // (+ the words a 128-base piece reads past the read's last)
    const int grid = (int)std::min<uint64_t>(n_max, 256 * 32);
    if (!R.pos_hint || R.hint_all) {
        // no position hints of the default lattice: another window or seed lattice — the every-position form, hints computed by the
        // walking wave (or read, where the filter kept every position's bit: hint_all)
        if (P.window < 6 || P.window > 9 || P.skips < 1 || P.lowDR + P.lowSp < 17 || P.highDR + P.highSp > 127 || P.highDR + P.highSp < P.lowDR + P.lowSp) return hipErrorNotSupported;
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_long_light_any), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e2 != hipSuccess) return e2;
        CRASS_LAUNCH(k_long_light_any, dim3(grid), dim3(WAVE), lds_bytes, st, R, P, d_n, n_max, out, slot_base, punt_list, d_punt_n, surv_idx);
        return hipGetLastError();
    }
    if (P.skips != 8 || P.window != 8) return hipErrorNotSupported;
    static const int force_ll = getenv("CRASS_HINT_RANGE") ? atoi(getenv("CRASS_HINT_RANGE")) : 0;
    const bool range = P.lowDR + P.lowSp != 49 || P.highDR + P.highSp != 97 || (force_ll & 2);
    hipError_t e = hipFuncSetAttribute(range ? reinterpret_cast<const void *>(&k_long_light<true>) : reinterpret_cast<const void *>(&k_long_light<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    if (range) CRASS_LAUNCH(k_long_light<true>, dim3(grid), dim3(WAVE), lds_bytes, st, R, P, d_n, n_max, out, slot_base, punt_list, d_punt_n, surv_idx);
    else CRASS_LAUNCH(k_long_light<false>, dim3(grid), dim3(WAVE), lds_bytes, st, R, P, d_n, n_max, out, slot_base, punt_list, d_punt_n, surv_idx);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------
// pass 1, step 2 (fast path): the same searchCore, one read per LANE.
//
// The wave-per-read kernel above spends most of its issue slots with one or two useful lanes
// (a serial DP, a serial column vote).  For short packed reads every lane can instead run the
// complete reference control flow for its own read: 64 reads advance together and loops of
// different length simply mask lanes off.  Per-lane state lives in LDS, word-/entry-interleaved
// across lanes ([index][lane]) so that lanes working at the same index hit different banks.
// Anything unusual (string pair with both lengths > 64, start/stop list overflow) is PUNTED
// (err = 4) to the wave-per-read kernel, which re-runs that read from scratch.
// Semantics are line-for-line those of search_core()/oracle; the parity suites cover both.
// ------------------------------------------------------------------------------------
struct LaneRead {
    const uint32_t *w;      // LDS: word i of this lane's read at w[i * 64]
    uint16_t *ss;           // LDS: start/stop entry k at ss[k * 64]
    int L, nss, cap, replen;
    uint32_t cmask;
    int punt;
};

static __device__ __forceinline__ uint32_t ln_word(const LaneRead &h, int i) { return h.w[i * WAVE]; }
static __device__ __forceinline__ uint32_t ln_base(const LaneRead &h, int p) { return (ln_word(h, p >> 4) >> ((p & 15) * 2)) & 3u; }
static __device__ __forceinline__ uint32_t ln_code(const LaneRead &h, int p)
{
    const int wi = p >> 4, sh = (p & 15) * 2;
    const uint64_t v = ((uint64_t)ln_word(h, wi + 1) << 32) | ln_word(h, wi);
    return (uint32_t)(v >> sh) & h.cmask;
}
static __device__ __forceinline__ uint32_t ln_ss(const LaneRead &h, int k) { return h.ss[k * WAVE]; }
static __device__ __forceinline__ void ln_ss_set(LaneRead &h, int k, uint32_t v) { h.ss[k * WAVE] = (uint16_t)v; }

// bases [start, start + 64) of the lane's read as two 64-bit words (2 bits per base, base `start` in bits 0-1); bases past
// the stored words read as 0.  Five independent LDS reads and four funnel shifts: the loops that follow run on registers
// (a per-base ln_base() in a dependent loop is one LDS round trip per iteration: the QC stage was 82 us of 155 that way).
static __device__ __forceinline__ void ln_load128(const LaneRead &h, int start, uint64_t &lo, uint64_t &hi)
{
    const int wi = start >> 4;
    const uint32_t sh = (uint32_t)(start & 15) * 2u;
    const uint32_t a0 = ln_word(h, wi), a1 = ln_word(h, wi + 1), a2 = ln_word(h, wi + 2), a3 = ln_word(h, wi + 3), a4 = ln_word(h, wi + 4);
    const uint32_t y0 = __builtin_amdgcn_alignbit(a1, a0, sh), y1 = __builtin_amdgcn_alignbit(a2, a1, sh);
    const uint32_t y2 = __builtin_amdgcn_alignbit(a3, a2, sh), y3 = __builtin_amdgcn_alignbit(a4, a3, sh);
    lo = (uint64_t)y0 | ((uint64_t)y1 << 32); hi = (uint64_t)y2 | ((uint64_t)y3 << 32);
}
// even bits of a 64-bit word compacted into the low 32 bits
static __device__ __forceinline__ uint32_t ln_even_bits(uint64_t x)
{
    x &= 0x5555555555555555ull;
    x = (x | (x >> 1)) & 0x3333333333333333ull;
    x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
    x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
    x = (x | (x >> 16)) & 0x00000000FFFFFFFFull;
    return (uint32_t)x;
}
// the two bit planes of n <= 64 bases held in (lo, hi): bit i of p0 / p1 = low / high bit of base i
static __device__ __forceinline__ void ln_planes(uint64_t lo, uint64_t hi, int n, uint64_t &p0, uint64_t &p1)
{
    p0 = (uint64_t)ln_even_bits(lo) | ((uint64_t)ln_even_bits(hi) << 32);
    p1 = (uint64_t)ln_even_bits(lo >> 1) | ((uint64_t)ln_even_bits(hi >> 1) << 32);
    const uint64_t m = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    p0 &= m; p1 &= m;
}

// bmpSearch semantics (PatternMatcher.cpp:26-59) for the w-mer at `pat` in [begin, end).  The text is pulled through a
// 64-bit register, 32 bases per refill: no LDS access inside the compare loop (a per-base ln_base() there was one LDS
// round trip per iteration).
static __device__ int ln_find(const LaneRead &h, int begin, int end, int pat, int plen)
{
    if (end - begin <= 0 || plen <= 0 || plen > end - begin) return -1;
    const uint32_t sj = ln_code(h, pat);
    uint32_t code = ln_code(h, begin);
    const int top = 2 * (plen - 1);
    int nb = begin + plen;                               // next base to shift in
    uint64_t buf = 0; int left = 0;                      // bases nb .. nb + left - 1, base nb in bits 0-1
    for (int p = begin;; p++) {
        if (code == sj) return p;
        if (p + 1 + plen > end) return -1;
        if (left == 0) {
            const int wi = nb >> 4;
            const uint32_t sh = (uint32_t)(nb & 15) * 2u;
            const uint32_t a0 = ln_word(h, wi), a1 = ln_word(h, wi + 1), a2 = ln_word(h, wi + 2);
            buf = (uint64_t)__builtin_amdgcn_alignbit(a1, a0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(a2, a1, sh) << 32);
            left = 32;
        }
        code = (code >> 2) | ((uint32_t)(buf & 3ull) << top);
        buf >>= 2; left--; nb++;
    }
}

static __device__ void ln_add(LaneRead &h, uint32_t i, uint32_t j)          // startStopsAdd, ReadHolder.cpp:263-297
{
    if (h.nss + 2 > h.cap) { h.punt = 1; return; }
    if (j >= (uint32_t)h.L) j = (uint32_t)h.L - 1;
    ln_ss_set(h, h.nss, i); ln_ss_set(h, h.nss + 1, j);
    h.nss += 2;
}

static __device__ void ln_scan_right(LaneRead &h, int pat, uint32_t pattern_length, uint32_t minSpacerLength, uint32_t scanRange)
{   // scanRight, libcrispr.cpp:170-263
    uint32_t last_repeat_index = ln_ss(h, h.nss - 2);
    uint32_t second_last_repeat_index = ln_ss(h, h.nss - 4);
    uint32_t repeat_spacing = last_repeat_index - second_last_repeat_index;
    const uint32_t read_length = (uint32_t)h.L;
    for (;;) {
        int candidate_repeat_index = (int)(last_repeat_index + repeat_spacing);
        uint32_t begin_search = (uint32_t)candidate_repeat_index - scanRange;
        uint32_t end_search = (uint32_t)candidate_repeat_index + pattern_length + scanRange;
        uint32_t scanRightMinBegin = last_repeat_index + pattern_length + minSpacerLength;
        if (begin_search < scanRightMinBegin) begin_search = scanRightMinBegin;
        if (begin_search > read_length - 1) return;
        if (end_search > read_length) end_search = read_length;
        if (begin_search >= end_search) return;
        int position = ln_find(h, (int)begin_search, (int)end_search, pat, (int)pattern_length);
        if (position < 0) return;
        ln_add(h, (uint32_t)position, (uint32_t)position + pattern_length - 1);
        if (h.punt) return;
        second_last_repeat_index = last_repeat_index;
        last_repeat_index = (uint32_t)position;
        repeat_spacing = last_repeat_index - second_last_repeat_index;
        if (repeat_spacing < (minSpacerLength + pattern_length)) return;
    }
}

// number of leading equal 2-bit groups of two strings given as xor (x0 low): up to n <= 64 groups
static __device__ __forceinline__ int ln_run_up(uint64_t x0, uint64_t x1, int n)
{
    int r = x0 ? (__ffsll((unsigned long long)x0) - 1) >> 1 : (x1 ? 32 + ((__ffsll((unsigned long long)x1) - 1) >> 1) : 64);
    return r < n ? r : n;
}
// ... of trailing equal groups, counted from group n - 1 downwards (n <= 64, n >= 1)
static __device__ __forceinline__ int ln_run_down(uint64_t x0, uint64_t x1, int n)
{
    int r;
    if (n <= 32) { const uint64_t y = x0 << (64 - 2 * n); r = y ? __clzll((long long)y) >> 1 : n; }
    else {
        const uint64_t y1 = x1 << (64 - 2 * (n - 32));
        if (y1) r = __clzll((long long)y1) >> 1;
        else r = (n - 32) + (x0 ? __clzll((long long)x0) >> 1 : 32);
    }
    return r < n ? r : n;
}

// extendPreRepeat (libcrispr.cpp:520-772) for exactly TWO repeats, in closed form.  With two repeats cut_off = 2 (:538-544),
// so a column is accepted iff both copies hold the same base (:617-651; packed reads hold A/C/G/T only), the right phase
// stops at the first disagreement, after shortest_spacing - minSpacer columns (:581), or when the second copy runs into
// the read end — there the reference drops the last repeat from the vote (:614-616) and the one that is left cannot reach
// the cut-off —; the left phase likewise, bounded by shortest_spacing - length so far (:674-675) and by the read start
// (:700-704).  The agreeing runs are xor + count-zeros on 128-bit pieces of the read instead of a per-column vote.
static __device__ uint32_t ln_extend2(LaneRead &h, int searchWindowLength, int minSpacerLength)
{
    const int j = (int)ln_ss(h, 0), p = (int)ln_ss(h, 2), L = h.L, w = searchWindowLength;
    const int spacing = p - j;
    int max_right = spacing - minSpacerLength;                  // (unsigned in the reference; spacing >= minSpacer + w here)
    if (max_right > L - (p + w)) max_right = L - (p + w);       // the second copy's column must lie inside the read
    int right = 0;
    while (right < max_right) {
        uint64_t a0, a1, b0, b1;
        ln_load128(h, j + w + right, a0, a1); ln_load128(h, p + w + right, b0, b1);
        const int n = max_right - right < 64 ? max_right - right : 64;
        const int r = ln_run_up(a0 ^ b0, a1 ^ b1, n);
        right += r;
        if (r < n) break;
    }
    const int len_r = w + right;
    int max_left = spacing - len_r;
    if (max_left < 0) max_left = 0;
    if (max_left > j) max_left = j;                             // the first copy's column must lie inside the read
    int left = 0;
    while (left < max_left) {
        const int n = max_left - left < 64 ? max_left - left : 64;
        uint64_t a0, a1, b0, b1;
        ln_load128(h, j - left - n, a0, a1); ln_load128(h, p - left - n, b0, b1);
        const int r = ln_run_down(a0 ^ b0, a1 ^ b1, n);
        left += r;
        if (r < n) break;
    }
    h.replen = w + right + left;
    for (int r = 0; r + 1 < h.nss; r += 2) {
        uint32_t a = ln_ss(h, r), b = ln_ss(h, r + 1);
        a = (a < (uint32_t)left) ? 0 : a - (uint32_t)left;
        b = (b + (uint32_t)right >= (uint32_t)L) ? (uint32_t)L - 1 : b + (uint32_t)right;
        ln_ss_set(h, r, a); ln_ss_set(h, r + 1, b);
    }
    return (uint32_t)h.replen;
}

static __device__ uint32_t ln_extend(LaneRead &h, int searchWindowLength, int minSpacerLength)
{   // extendPreRepeat, libcrispr.cpp:520-772 (serial columns, like the reference)
    if (h.nss == 4) return ln_extend2(h, searchWindowLength, minSpacerLength);
    const uint32_t num_repeats = (uint32_t)h.nss / 2;
    h.replen = searchWindowLength;
    int cut_off = (int)(num_repeats - 1);
    if (2 > cut_off) cut_off = 2;
    const uint32_t first_repeat_start_index = ln_ss(h, 0);
    const uint32_t last_repeat_start_index = ln_ss(h, h.nss - 2);
    const uint32_t end_index = (uint32_t)h.nss;
    const uint32_t seqlen = (uint32_t)h.L;
    uint32_t shortest_repeat_spacing = (uint32_t)((int)ln_ss(h, 2) - (int)ln_ss(h, 0));
    for (uint32_t i = 4; i < end_index; i += 2) {
        uint32_t sp = (uint32_t)((int)ln_ss(h, i) - (int)ln_ss(h, i - 2));
        if (sp < shortest_repeat_spacing) shortest_repeat_spacing = sp;
    }
    uint32_t right_extension_length = 0;
    uint32_t max_right_extension_length = shortest_repeat_spacing - (uint32_t)minSpacerLength;
    int DR_index_end = (int)end_index;
    while (max_right_extension_length > 0) {
        if ((last_repeat_start_index + (uint32_t)searchWindowLength + right_extension_length) >= seqlen) DR_index_end -= 2;
        int cnt[4] = {0, 0, 0, 0};
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int k = 0; k < DR_index_end; k += 2) {
            const uint32_t pos = ln_ss(h, k) + (uint32_t)h.replen;
            if (pos >= seqlen) break;
            const uint32_t b = ln_base(h, (int)pos);
            c0 += (b == 0); c1 += (b == 1); c2 += (b == 2); c3 += (b == 3);
        }
        (void)cnt;
        if ((c0 >= cut_off) || (c1 >= cut_off) || (c2 >= cut_off) || (c3 >= cut_off)) {
            h.replen++; max_right_extension_length--; right_extension_length++;
        } else break;
    }
    uint32_t left_extension_length = 0;
    const int test_for_negative = (int)(shortest_repeat_spacing - (uint32_t)h.replen);
    const uint32_t max_left_extension_length = (test_for_negative >= 0) ? (uint32_t)test_for_negative : 0;
    uint32_t DR_index_start = 0;
    while (left_extension_length < max_left_extension_length) {
        if ((int)first_repeat_start_index - (int)left_extension_length <= 0) DR_index_start += 2;
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (uint32_t k = DR_index_start; k < end_index; k += 2) {
            const int idx = (int)(ln_ss(h, (int)k) - left_extension_length - 1);
            if (idx < 0 || idx >= h.L) continue;
            const uint32_t b = ln_base(h, idx);
            c0 += (b == 0); c1 += (b == 1); c2 += (b == 2); c3 += (b == 3);
        }
        if ((c0 >= cut_off) || (c1 >= cut_off) || (c2 >= cut_off) || (c3 >= cut_off)) {
            h.replen++; left_extension_length++;
        } else break;
    }
    for (int r = 0; r + 1 < h.nss; r += 2) {
        uint32_t a = ln_ss(h, r), b = ln_ss(h, r + 1);
        a = (a < left_extension_length) ? 0 : a - left_extension_length;
        b = (b + right_extension_length >= seqlen) ? seqlen - 1 : b + right_extension_length;
        ln_ss_set(h, r, a); ln_ss_set(h, r + 1, b);
    }
    return (uint32_t)h.replen;
}

// Bit-parallel distance (see lane_lev_bp: Hyyro's OSA recurrences with the reference's transposition term restricted to
// i > 2 && j > 2, PatternMatcher.cpp:181-186) between read[s0, s0+n) and read[t0, t0+m), n <= m, n <= 8 * sizeof(WORD),
// m <= 64, both strings in registers.  stop_at >= 0: the caller only wants to know whether the distance is < stop_at —
// the bottom-row score changes by at most 1 per column, so once score - (columns left) >= stop_at the answer is "no" and
// stop_at is returned (any value >= stop_at would do).
template <typename WORD>
static __device__ __forceinline__ int ln_lev_regs(uint64_t s_lo, uint64_t s_hi, int n, uint64_t t_lo, uint64_t t_hi, int m, int stop_at)
{
    uint64_t q0, q1;
    ln_planes(s_lo, s_hi, n, q0, q1);
    const uint64_t nm = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    const WORD pm0 = (WORD)(~q0 & ~q1 & nm), pm1 = (WORD)(q0 & ~q1), pm2 = (WORD)(~q0 & q1 & nm), pm3 = (WORD)(q0 & q1);
    WORD VP = ~(WORD)0, VN = 0, D0 = 0, PMold = 0;
    int dist = n;
    const int topbit = n - 1;
    const WORD top = (WORD)1 << topbit;
    for (int j = 0; j < m; j++) {
        const uint32_t c = (uint32_t)((j < 32 ? t_lo >> (2 * j) : t_hi >> (2 * (j - 32))) & 3ull);
        const WORD PMj = (c & 2u) ? ((c & 1u) ? pm3 : pm2) : ((c & 1u) ? pm1 : pm0);         // three selects instead of four masked ors
        WORD TR = (WORD)((((WORD)~D0) & PMj) << 1) & PMold & ~(WORD)3;
        if (j < 2) TR = 0;
        D0 = (WORD)((WORD)((WORD)(PMj & VP) + VP) ^ VP) | PMj | VN;
        D0 |= TR;
        WORD HP = VN | (WORD)~(D0 | VP);
        WORD HN = D0 & VP;
        dist += (int)((HP & top) != 0) - (int)((HN & top) != 0);
        HP = (WORD)(HP << 1) | (WORD)1;
        HN = (WORD)(HN << 1);
        VP = HN | (WORD)~(D0 | HP);
        VN = HP & D0;
        PMold = PMj;
        if (stop_at >= 0 && dist - (m - 1 - j) >= stop_at) return stop_at;
    }
    return dist;
}

// bit-parallel distance between read[s0, s0+n) and read[t0, t0+m) on 2-bit codes, one base per LDS access (strings
// longer than 64 bases: only reachable with DR / spacer bounds far above the defaults)
template <typename WORD>
static __device__ __forceinline__ int ln_lev_core(const LaneRead &h, int s0, int n, int t0, int m)
{
    WORD pm0 = 0, pm1 = 0, pm2 = 0, pm3 = 0;
    for (int i = 0; i < n; i++) {
        const uint32_t c = ln_base(h, s0 + i);
        const WORD b = (WORD)1 << i;
        pm0 |= b & ((WORD)0 - (WORD)(c == 0)); pm1 |= b & ((WORD)0 - (WORD)(c == 1));
        pm2 |= b & ((WORD)0 - (WORD)(c == 2)); pm3 |= b & ((WORD)0 - (WORD)(c == 3));
    }
    WORD VP = ~(WORD)0, VN = 0, D0 = 0, PMold = 0;
    int dist = n;
    const int topbit = n - 1;
    for (int j = 0; j < m; j++) {
        const uint32_t c = ln_base(h, t0 + j);
        const WORD PMj = (pm0 & ((WORD)0 - (WORD)(c == 0))) | (pm1 & ((WORD)0 - (WORD)(c == 1))) |
                         (pm2 & ((WORD)0 - (WORD)(c == 2))) | (pm3 & ((WORD)0 - (WORD)(c == 3)));
        WORD TR = (WORD)((((WORD)~D0) & PMj) << 1) & PMold & ~(WORD)3;
        TR &= (WORD)0 - (WORD)(j >= 2);
        D0 = (WORD)((WORD)((WORD)(PMj & VP) + VP) ^ VP) | PMj | VN;
        D0 |= TR;
        WORD HP = VN | (WORD)~(D0 | VP);
        WORD HN = D0 & VP;
        dist += (int)((HP >> topbit) & 1) - (int)((HN >> topbit) & 1);
        HP = (WORD)(HP << 1) | (WORD)1;
        HN = (WORD)(HN << 1);
        VP = HN | (WORD)~(D0 | HP);
        VN = HP & D0;
        PMold = PMj;
    }
    return dist;
}

// edit distance of read[s0, s0+n) and read[t0, t0+m); sets h.punt when the pair needs the wavefront DP.  stop_at: see
// ln_lev_regs (-1: the exact distance)
static __device__ int ln_distance(LaneRead &h, int s0, int n, int t0, int m, int stop_at)
{
    if (n > m) { int x = s0; s0 = t0; t0 = x; x = n; n = m; m = x; }
    if (n > 64) { h.punt = 1; return 0; }
    if (m <= 64) {
        uint64_t sl, sh, tl, th;
        ln_load128(h, s0, sl, sh); ln_load128(h, t0, tl, th);
        // (one instantiation for every pair: a wave whose pairs fall on both sides of 32 would issue both loops in turn)
        return ln_lev_regs<uint64_t>(sl, sh, n, tl, th, m, stop_at);
    }
    return (n <= 32) ? ln_lev_core<uint32_t>(h, s0, n, t0, m) : ln_lev_core<uint64_t>(h, s0, n, t0, m);
}

// getStringSimilarity (PatternMatcher.cpp:197-204); sets h.punt when the pair needs the wavefront DP
static __device__ float ln_similarity(LaneRead &h, int s0, int n, int t0, int m)
{
    float max_length = (float)(n > m ? n : m);
    if (n < 3 || m < 3) return 0.0f;
    const int d = ln_distance(h, s0, n, t0, m, -1);
    float edit_distance = (float)d;
    return (float)(1.0 - (double)(edit_distance / max_length));
}

// "(double)getStringSimilarity(a, b) > cut" without the full distance when the answer is no: the similarity falls
// monotonically with the distance, so the smallest distance d_no whose similarity is NOT above the cut is found first
// (with the reference's own float expression) and the DP stops as soon as the distance is known to reach it
static __device__ bool ln_similarity_above(LaneRead &h, int s0, int n, int t0, int m, double cut)
{
    if (n < 3 || m < 3) return 0.0 > cut;
    const float max_length = (float)(n > m ? n : m);
    int d_no = 0;
    while (d_no <= (n > m ? n : m) && (double)(float)(1.0 - (double)((float)d_no / max_length)) > cut) d_no++;
    const int d = ln_distance(h, s0, n, t0, m, d_no);
    if (h.punt) return false;
    return (double)(float)(1.0 - (double)((float)d / max_length)) > cut;
}

static __device__ int ln_qc(LaneRead &h, int minSpacerLength, int maxSpacerLength, uint32_t dbg = 0)
{   // qcFoundRepeats, libcrispr.cpp:869-1029
    if (dbg == 6) return 1;                             // (CRASS_SURV_DEBUG, timing breakdown only)
    const int num_repeats = h.nss / 2;
    if (num_repeats < 2) return -1;
    uint32_t rep_len;
    const uint32_t rep_start = ln_ss(h, 0);
    if (!substr_len(h.L, rep_start, ln_ss(h, 1) - rep_start + 1, rep_len)) return -1;
    {   // isRepeatLowComplexity (:1031-1069); packed reads hold A/C/G/T only
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        if (rep_len <= 64) {                            // base counts from the two bit planes
            uint64_t lo, hi, p0, p1;
            ln_load128(h, (int)rep_start, lo, hi);
            ln_planes(lo, hi, (int)rep_len, p0, p1);
            c3 = __popcll(p0 & p1); c1 = __popcll(p0 & ~p1); c2 = __popcll(~p0 & p1); c0 = (int)rep_len - c1 - c2 - c3;
        } else
            for (uint32_t i = 0; i < rep_len; i++) { const uint32_t b = ln_base(h, (int)(rep_start + i)); c0 += (b == 0); c1 += (b == 1); c2 += (b == 2); c3 += (b == 3); }
        const int cut_off = (int)((double)(int)rep_len * 0.75);
        if (c0 > cut_off || c3 > cut_off || c2 > cut_off || c1 > cut_off) return 0;
    }
    if (dbg == 5) return 1;
    bool is_short = (2 > (num_repeats - 1));
    if (!is_short) {
        float ave_spacer_to_spacer_len_difference = 0.0f, ave_repeat_to_spacer_len_difference = 0.0f;
        float ave_spacer_to_spacer_difference = 0.0f, ave_repeat_to_spacer_difference = 0.0f;
        int min_spacer_length = 10000000, max_spacer_length = 0, num_compared = 0;
        const int nsp = num_repeats - 1;
        uint32_t cur_start = ln_ss(h, 1) + 1, cur_len;
        if (!substr_len(h.L, cur_start, ln_ss(h, 2) - cur_start, cur_len)) return -1;
        if (nsp == 2) {
            // three repeats, ONE comparison: every average is the single value itself (x / 1.0f == x), so the two similarity
            // tests are plain threshold tests and may stop early (ln_similarity_above); the outcome is a boolean either way
            uint32_t nxt_start = ln_ss(h, 3) + 1, nxt_len;
            if (!substr_len(h.L, nxt_start, ln_ss(h, 4) - nxt_start, nxt_len)) return -1;
            const int mn = (int)(cur_len < nxt_len ? cur_len : nxt_len), mx = (int)(cur_len > nxt_len ? cur_len : nxt_len);
            if (mn < minSpacerLength || mx > maxSpacerLength) return 0;
            const bool a = ln_similarity_above(h, (int)cur_start, (int)cur_len, (int)nxt_start, (int)nxt_len, 0.82);
            if (h.punt) return 0;
            if (a) return 0;
            const bool b = ln_similarity_above(h, (int)rep_start, (int)rep_len, (int)cur_start, (int)cur_len, 0.82);
            if (h.punt) return 0;
            if (b) return 0;
            if ((int)fabsf(((float)cur_len - (float)nxt_len) / 1.0f) > 12) return 0;
            if ((int)fabsf(((float)rep_len - (float)cur_len) / 1.0f) > 30) return 0;
            return 1;
        }
        for (int i = 0; i < nsp; i++) {
            if ((int)cur_len < min_spacer_length) min_spacer_length = (int)cur_len;
            if ((int)cur_len > max_spacer_length) max_spacer_length = (int)cur_len;
            if (i + 1 < nsp) {
                uint32_t nxt_start = ln_ss(h, 2 * i + 3) + 1, nxt_len;
                if (!substr_len(h.L, nxt_start, ln_ss(h, 2 * i + 4) - nxt_start, nxt_len)) return -1;
                num_compared++;
                ave_repeat_to_spacer_difference += ln_similarity(h, (int)rep_start, (int)rep_len, (int)cur_start, (int)cur_len);
                float ss_diff = 0;
                ss_diff += ln_similarity(h, (int)cur_start, (int)cur_len, (int)nxt_start, (int)nxt_len);
                ave_spacer_to_spacer_difference += ss_diff;
                ave_spacer_to_spacer_len_difference += ((float)cur_len - (float)nxt_len);
                ave_repeat_to_spacer_len_difference += ((float)rep_len - (float)cur_len);
                cur_start = nxt_start; cur_len = nxt_len;
                if (h.punt) return 0;
            }
        }
        ave_spacer_to_spacer_difference /= (float)num_compared;
        ave_repeat_to_spacer_difference /= (float)num_compared;
        ave_spacer_to_spacer_len_difference /= (float)num_compared;
        ave_spacer_to_spacer_len_difference = fabsf(ave_spacer_to_spacer_len_difference);
        ave_repeat_to_spacer_len_difference /= (float)num_compared;
        ave_repeat_to_spacer_len_difference = fabsf(ave_repeat_to_spacer_len_difference);
        if (min_spacer_length < minSpacerLength) return 0;
        if (max_spacer_length > maxSpacerLength) return 0;
        if ((double)ave_spacer_to_spacer_difference > 0.82) return 0;
        if ((double)ave_repeat_to_spacer_difference > 0.82) return 0;
        if ((int)ave_spacer_to_spacer_len_difference > 12) return 0;
        if ((int)ave_repeat_to_spacer_len_difference > 30) return 0;
    }
    if (is_short) {
        uint32_t s = ln_ss(h, 1) + 1;
        uint32_t e = ln_ss(h, 2) - 1;
        uint32_t sp_len;
        if (!substr_len(h.L, s, e - s, sp_len)) return -1;
        if ((int)sp_len < minSpacerLength) return 0;
        if ((int)sp_len > maxSpacerLength) return 0;
        const bool too_similar = ln_similarity_above(h, (int)rep_start, (int)rep_len, (int)s, (int)sp_len, 0.82);
        if (h.punt) return 0;
        if (too_similar) return 0;
        int dlen = (int)sp_len - (int)rep_len;
        if (dlen < 0) dlen = -dlen;
        if (dlen > 30) return 0;
    }
    return 1;
}

// ---- the QC's threshold tests as TASKS of the block (k_survivor_lanes) ----
// qcFoundRepeats spends its time in one or two "similarity > 0.82" tests per candidate (two repeats: repeat vs spacer; three:
// spacer vs spacer, repeat vs spacer) — bit-parallel edit distances of ~40 columns.  Run by the candidate's own lane they
// occupied the third of a block's lanes that hold real CRISPR reads, once or twice in a row, while the other lanes of the block
// — reads that a chance seed hit let through — had long finished: 180 of the lane kernel's 520 us at 100 M reads.  The tests are
// therefore QUEUED in LDS by the candidates' lanes and run by ALL lanes of the block, a test per lane (any lane reads any
// lane's words: the rows are LDS), tests whose shorter string has at most 32 bases on 32-bit words from the queue's front, the
// others on 64-bit words from its back.  The decisions stay with the candidate's lane, in the reference's order.
struct QcTask { uint32_t a, b; };                 // a: owner thread (8) | s0 (9) | n (7) | d_no (7);  b: t0 (9) | m (7)
struct QcPool {
    QcTask *task;                                 // [2 * threads of the block]: 32-bit tests from the front, 64-bit ones from the back
    uint8_t *res;                                 // [2 * threads]: by task slot, 1 = "similarity above the cut"
    uint32_t *cnt;                                // [0] 32-bit tests queued, [1] 64-bit tests queued
    const uint32_t *rows;                         // the block's LDS rows (sl_lds)
    uint32_t wave_words;                          // words per wave's part of it
    uint32_t cap;                                 // 2 * threads
};
// what a lane remembers between queueing its tests and reading their results
struct QcPending { int kind; uint32_t slot_a, slot_b; uint32_t cur_len, nxt_len, rep_len, sp_len; };

// "(double)getStringSimilarity(a, b) > 0.82" as a queued test; returns the task's slot, or 0xFFFFFFFF with *now = the answer when
// it needs no distance (a string below three bases: similarity 0)
static __device__ uint32_t qc_enqueue_above(const QcPool &q, int s0, int n, int t0, int m, bool *now)
{
    *now = false;
    if (n < 3 || m < 3) { *now = 0.0 > 0.82; return 0xFFFFFFFFu; }
    if (n > m) { int x = s0; s0 = t0; t0 = x; x = n; n = m; m = x; }
    const float max_length = (float)m;
    int d_no = 0;
    while (d_no <= m && (double)(float)(1.0 - (double)((float)d_no / max_length)) > 0.82) d_no++;
    uint32_t slot;
    if (n <= 32) slot = atomicAdd(&q.cnt[0], 1u);
    else slot = q.cap - 1u - atomicAdd(&q.cnt[1], 1u);
    QcTask t;
    t.a = (uint32_t)threadIdx.x | ((uint32_t)s0 << 8) | ((uint32_t)n << 17) | ((uint32_t)d_no << 24);
    t.b = (uint32_t)t0 | ((uint32_t)m << 9);
    q.task[slot] = t;
    return slot;
}

// one queued test, by whichever lane drew it
template <typename WORD>
static __device__ __forceinline__ void qc_run_task(const QcPool &q, uint32_t slot)
{
    const QcTask t = q.task[slot];
    const uint32_t owner = t.a & 0xFFu;
    const int s0 = (int)((t.a >> 8) & 0x1FFu), n = (int)((t.a >> 17) & 0x7Fu), d_no = (int)(t.a >> 24);
    const int t0 = (int)(t.b & 0x1FFu), m = (int)(t.b >> 9);
    LaneRead ho;
    ho.w = q.rows + (size_t)(owner >> 6) * q.wave_words + (owner & 63u);
    uint64_t sl, sh, tl, th;
    ln_load128(ho, s0, sl, sh); ln_load128(ho, t0, tl, th);
    const int d = ln_lev_regs<WORD>(sl, sh, n, tl, th, m, d_no);
    q.res[slot] = (double)(float)(1.0 - (double)((float)d / (float)m)) > 0.82 ? 1 : 0;
}

// qcFoundRepeats (libcrispr.cpp:869-1029) up to its similarity tests: -1 / 0 / 1 = decided here (ln_qc's values); 2 = the
// candidate's tests are queued (pd says which), qc_pool_end decides.  Candidates of four repeats or more, and strings beyond
// 64 bases, take ln_qc as before (rare at these read lengths).
static __device__ int qc_pool_begin(LaneRead &h, int minSpacerLength, int maxSpacerLength, uint32_t dbg, const QcPool &q, QcPending &pd)
{
    pd.kind = 0;
    const int num_repeats = h.nss / 2;
    if (dbg == 5 || dbg == 6 || num_repeats < 2 || num_repeats > 3) return ln_qc(h, minSpacerLength, maxSpacerLength, dbg);
    uint32_t rep_len;
    const uint32_t rep_start = ln_ss(h, 0);
    if (!substr_len(h.L, rep_start, ln_ss(h, 1) - rep_start + 1, rep_len)) return -1;
    if (rep_len > 64) return ln_qc(h, minSpacerLength, maxSpacerLength, dbg);
    {   // isRepeatLowComplexity (:1031-1069); packed reads hold A/C/G/T only
        uint64_t lo, hi, p0, p1;
        ln_load128(h, (int)rep_start, lo, hi);
        ln_planes(lo, hi, (int)rep_len, p0, p1);
        const int c3 = __popcll(p0 & p1), c1 = __popcll(p0 & ~p1), c2 = __popcll(~p0 & p1), c0 = (int)rep_len - c1 - c2 - c3;
        const int cut_off = (int)((double)(int)rep_len * 0.75);
        if (c0 > cut_off || c3 > cut_off || c2 > cut_off || c1 > cut_off) return 0;
    }
    if (num_repeats == 3) {
        uint32_t cur_start = ln_ss(h, 1) + 1, cur_len;
        if (!substr_len(h.L, cur_start, ln_ss(h, 2) - cur_start, cur_len)) return -1;
        uint32_t nxt_start = ln_ss(h, 3) + 1, nxt_len;
        if (!substr_len(h.L, nxt_start, ln_ss(h, 4) - nxt_start, nxt_len)) return -1;
        const int mn = (int)(cur_len < nxt_len ? cur_len : nxt_len), mx = (int)(cur_len > nxt_len ? cur_len : nxt_len);
        if (mn < minSpacerLength || mx > maxSpacerLength) return 0;
        if (cur_len > 64 || nxt_len > 64) return ln_qc(h, minSpacerLength, maxSpacerLength, dbg);
        bool a_now, b_now;
        pd.slot_a = qc_enqueue_above(q, (int)cur_start, (int)cur_len, (int)nxt_start, (int)nxt_len, &a_now);
        if (pd.slot_a == 0xFFFFFFFFu && a_now) return 0;
        pd.slot_b = qc_enqueue_above(q, (int)rep_start, (int)rep_len, (int)cur_start, (int)cur_len, &b_now);
        if (pd.slot_b == 0xFFFFFFFFu && b_now) return 0;          // (its own test only: a queued first test is not waited for — both "return 0")
        pd.kind = 3; pd.cur_len = cur_len; pd.nxt_len = nxt_len; pd.rep_len = rep_len;
        return 2;
    }
    // two repeats
    uint32_t s = ln_ss(h, 1) + 1;
    uint32_t e = ln_ss(h, 2) - 1;
    uint32_t sp_len;
    if (!substr_len(h.L, s, e - s, sp_len)) return -1;
    if ((int)sp_len < minSpacerLength) return 0;
    if ((int)sp_len > maxSpacerLength) return 0;
    if (sp_len > 64) return ln_qc(h, minSpacerLength, maxSpacerLength, dbg);
    bool now;
    pd.slot_a = qc_enqueue_above(q, (int)rep_start, (int)rep_len, (int)s, (int)sp_len, &now);
    if (pd.slot_a == 0xFFFFFFFFu && now) return 0;
    pd.slot_b = 0xFFFFFFFFu;
    pd.kind = 2; pd.sp_len = sp_len; pd.rep_len = rep_len;
    return 2;
}
// ... and from the tests' results on (same values as ln_qc)
static __device__ int qc_pool_end(const QcPool &q, const QcPending &pd)
{
    const bool a = pd.slot_a != 0xFFFFFFFFu && q.res[pd.slot_a] != 0;
    if (pd.kind == 3) {
        if (a) return 0;
        const bool b = pd.slot_b != 0xFFFFFFFFu && q.res[pd.slot_b] != 0;
        if (b) return 0;
        if ((int)fabsf(((float)pd.cur_len - (float)pd.nxt_len) / 1.0f) > 12) return 0;
        if ((int)fabsf(((float)pd.rep_len - (float)pd.cur_len) / 1.0f) > 30) return 0;
        return 1;
    }
    if (a) return 0;
    int dlen = (int)pd.sp_len - (int)pd.rep_len;
    if (dlen < 0) dlen = -dlen;
    if (dlen > 30) return 0;
    return 1;
}

// active: this thread holds a read (the others only take part in the block's barriers and run queued tests)
static __device__ int ln_search_core(LaneRead &h, const DevParams &o, uint64_t seed_hint, bool active, const QcPool &pool)
{   // searchCore, libcrispr.cpp:265-395
    const uint32_t seq_length = (uint32_t)h.L;
    const uint32_t skips = o.skips;
    int searchEnd = (int)(seq_length - o.lowDR - o.lowSp - o.window - 1);
    h.nss = 0;
    bool on_lattice = true;
    uint32_t lattice_i = 0;
    uint32_t j = 0;
    // Two nested loops instead of the reference's one: the inner loop advances a lane from seed to seed (find, chain,
    // extend — cheap since the closed-form extension) until it holds a candidate that is due for qcFoundRepeats or has
    // run out of seeds; only then do the lanes of the wave that hold such a candidate run the QC, together.  The
    // sequence of events per read is the reference's; what changes is that the QC code — by far the longest stretch —
    // is issued once per round for the whole wave instead of once per seed iteration in which some lane needs it
    // (lanes reach their first QC in different iterations: 51 us of the kernel's 117 were QC issue slots).
    bool finished = !active || searchEnd < 0; int result = 0;      // this lane's searchCore has returned `result`
    for (;;) {
        bool ready = false, error = false, out_of_seeds = false;
        // (wave-uniform loop condition: the lanes leave this loop TOGETHER, whatever the compiler makes of the control flow)
        while (__any((int)(!finished && !ready && !out_of_seeds && !error))) {
            if (finished || ready || out_of_seeds || error) continue;
            // Each lane first walks (cheaply) to ITS next seed worth evaluating.  A lattice seed whose hint bit is
            // clear is a no-op iteration in the reference (no hit => no start/stops => numRepeats 0): skipping it
            // changes nothing.
            while (on_lattice && j <= (uint32_t)searchEnd && lattice_i < 64 && !((seed_hint >> lattice_i) & 1ull)) { j += skips; lattice_i++; }
            if (j > (uint32_t)searchEnd) { out_of_seeds = true; continue; }
            if (on_lattice) lattice_i++;
            uint32_t beginSearch = j + o.lowDR + o.lowSp;
            uint32_t endSearch = j + o.highDR + o.highSp + o.window;
            if (endSearch >= seq_length) endSearch = seq_length - 1;
            if (endSearch < beginSearch) endSearch = beginSearch;
            if (beginSearch > seq_length) { error = true; continue; }
            int pos = ln_find(h, (int)beginSearch, (int)endSearch, (int)j, (int)o.window);
            if (o.debug_stop == 2) pos = -1;                // (CRASS_SURV_DEBUG, timing breakdown only: seed finds alone)
            if (pos >= 0) {
                ln_add(h, j, j + o.window - 1);
                ln_add(h, (uint32_t)pos, (uint32_t)pos + o.window - 1);
                if (!h.punt) ln_scan_right(h, (int)j, o.window, o.lowSp, 24);
                if (h.punt) { finished = true; result = 0; continue; }
            }
            if ((uint32_t)(h.nss / 2) >= o.minRepeats) {
                uint32_t actual_repeat_length = ln_extend(h, (int)o.window, (int)o.lowSp);
                if (o.debug_stop != 3 && (actual_repeat_length >= o.lowDR) && (actual_repeat_length <= o.highDR)) { ready = true; continue; }    // (3: no QC)
                j = ln_ss(h, h.nss - 1) - 1;
                on_lattice = false;
            }
            h.nss = 0;
            j = j + skips;
        }
        if (!finished && error) { finished = true; result = -1; }
        if (!finished && out_of_seeds) { finished = true; result = 0; }
        // From here on the BLOCK moves together (three barriers per round, one or two rounds): the lanes that hold a candidate
        // queue its similarity tests, every lane of the block runs queued tests, the candidates' lanes decide.
        if (!__syncthreads_or((int)(!finished))) break;
        QcPending pd;
        pd.kind = 0;
        int qc = 0;
        if (!finished) qc = qc_pool_begin(h, (int)o.lowSp, (int)o.highSp, o.debug_stop, pool, pd);
        __syncthreads();
        {
            const uint32_t n32 = pool.cnt[0], n64 = pool.cnt[1];
            for (uint32_t t = threadIdx.x; t < n32; t += blockDim.x) qc_run_task<uint32_t>(pool, t);
            // (the 64-bit tests from the block's last lanes downwards: the waves that took 32-bit tests take these last)
            for (uint32_t t = blockDim.x - 1u - threadIdx.x; t < n64; t += blockDim.x) qc_run_task<uint64_t>(pool, pool.cap - 1u - t);
        }
        __syncthreads();
        if (threadIdx.x == 0) { pool.cnt[0] = 0u; pool.cnt[1] = 0u; }      // (the next round queues behind this round's first barrier)
        if (!finished) {
            if (qc == 2) qc = qc_pool_end(pool, pd);
            if (h.punt) { finished = true; result = 0; }
            else if (qc < 0) { finished = true; result = -1; }
            else if (qc) { finished = true; result = 1; }
            else {
                j = ln_ss(h, h.nss - 1) - 1;                // a rejected candidate: on with the seeds behind it (:390)
                on_lattice = false;
                h.nss = 0;
                j = j + skips;
            }
        }
    }
    return result;
}

// SL_WAVES waves per block.  A third of the survivors are real CRISPR reads (scan, extension, QC, orientation: the whole
// searchCore), the rest are spurious seed hits that leave after one window — and in slot order every wave holds ~21 of the former,
// so every wave runs for as long as its slowest lane with a third of its lanes busy.  The block therefore deals its 64 x SL_WAVES
// slots out by expected work: reads with at least two hinted lattice seeds (a repeat with a copy one unit on gives several; a
// chance match gives one) first, in slot order, then the others.  Each lane still writes its own slot: the result arrays do not
// notice.  PMC at 100 M reads: 318 M -> 224 M VALU wave instructions (8-wave blocks); 4 waves per block is the measured optimum
// (590 us in slot order with one wave per block -> 550; 8 waves 593, 16 waves 754: a block holds its LDS until its slowest wave ends).
#define SL_WAVES 4
// hint bit i of a lane's read = "lattice seed i may have a copy in its window", for the first 64 seeds; the others are walked.
// From the lane-per-read filter's word (32 seeds: reads of up to 256 bases) or from the read's position hints (bit 8 i of the
// bitmap: k_hint_positions filled the lattice class and cleared what lies behind searchEnd) — bit 0 of each byte, gathered by one
// multiplication per 64 positions (the partial products 2^(56 + 8k - 7j) of bits 8k and multiplier terms 2^(56 - 7j) are all
// distinct, and only those with k == j fall into the top byte)
static __device__ __forceinline__ uint64_t ln_hint64(const DevReads &R, const uint32_t *seed_hint, uint64_t r, int L, uint32_t skips)
{
    if (seed_hint) return 0xFFFFFFFF00000000ull | (uint64_t)seed_hint[r];
    if (!R.pos_hint) return ~0ull;
    const uint64_t *ph = R.pos_hint + rd_hint_off(R, r);
    const int nh = (L + 63) >> 6;
    if (R.hint_all) {
        // every position has a bit (another window or seed lattice): seed i sits at i * skips
        uint64_t hint = 0;
        for (uint32_t i = 0, p = 0; i < 64u && (int)p < L; i++, p += skips) hint |= ((ph[p >> 6] >> (p & 63u)) & 1ull) << i;
        return hint;
    }
    uint64_t hint = 0;
    for (int k = 0; k < 8 && k < nh; k++)
        hint |= (((ph[k] & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56) << (8 * k);
    return hint;
}
__global__ __launch_bounds__(WAVE * SL_WAVES) void k_survivor_lanes(DevReads R, DevParams P, const uint64_t *surv_idx, const uint32_t *d_n_surv,
                                                         uint64_t n_max, SurvOut *out, char *dr_chars, uint32_t dr_stride,
                                                         uint32_t *ss_pool, uint32_t ss_cap, uint8_t *found_flag,
                                                         const uint32_t *seed_hint, uint32_t words_per_read, DevMerge IM, int do_init, int regroup,
                                                         uint32_t *punt_cnt)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sl_lds[];
    __shared__ uint16_t sl_perm[WAVE * SL_WAVES];
    __shared__ uint32_t sl_cnt[SL_WAVES];
    __shared__ uint32_t sl_hint[WAVE * SL_WAVES];       // the slots' filter hints, read once (for the regrouping) and handed to whoever gets the slot
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x >> 6;
    // the tables of the merge that follows this stage are cleared on the way (stores next to an issue-bound kernel)
    if (do_init) dm_init_slice(IM, blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
    uint64_t n_surv = *d_n_surv;
    if (n_surv > n_max) n_surv = n_max;
    uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (regroup) {
        bool heavy = false;
        if (s < n_surv) {
            const uint64_t r0 = surv_idx[s];
            const uint32_t h0 = seed_hint ? seed_hint[r0] : 0u;
            sl_hint[threadIdx.x] = h0;
            heavy = !rd_is_exc(R, r0) && (seed_hint ? __popc(h0) : __popcll(ln_hint64(R, nullptr, r0, (int)rd_len(R, r0), P.skips))) >= 2;
        }
        const uint64_t hb = __ballot(heavy);
        if (lane == 0) sl_cnt[wv] = (uint32_t)__popcll(hb);
        __syncthreads();
        uint32_t heavy_before = 0, heavy_total = 0;
#pragma unroll
        for (int k = 0; k < SL_WAVES; k++) { const uint32_t c = sl_cnt[k]; heavy_total += c; if (k < wv) heavy_before += c; }
        const uint32_t below = (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
        const uint32_t pos = heavy ? heavy_before + below
                                   : heavy_total + ((uint32_t)wv * WAVE - heavy_before) + ((uint32_t)lane - below);
        // (rotating which wave of the block gets the heavy slots, so that one SIMD of a CU does not run every block's long wave:
        // measured, no difference — 114 vs 113 us for an eighth of the 100 M reads, 4.19 vs 4.13 ms at 100 M; NOTES r06)
        sl_perm[pos] = (uint16_t)threadIdx.x;
        __syncthreads();
        s = blockIdx.x * (uint64_t)blockDim.x + sl_perm[threadIdx.x];
    }
    const bool hint_in_lds = regroup && seed_hint;
    const size_t wave_words = (size_t)(words_per_read + 5) * WAVE + ((size_t)ss_cap * WAVE + 1) / 2;      // this wave's part of the LDS
    uint32_t *wbase = sl_lds + (size_t)wv * wave_words;
    uint32_t *lw = wbase + lane;                                       // [word][lane]
    uint16_t *lss = reinterpret_cast<uint16_t *>(wbase + (size_t)(words_per_read + 5) * WAVE) + lane;   // [entry][lane]
    // (no thread leaves before the end: the block's lanes run the candidates' queued similarity tests together, QcPool)
    static_assert(WAVE * SL_WAVES <= 256, "QcTask keeps the owner's thread index in 8 bits");
    __shared__ QcTask sl_task[2 * WAVE * SL_WAVES];
    __shared__ uint8_t sl_res[2 * WAVE * SL_WAVES];
    __shared__ uint32_t sl_qcnt[2];
    if (threadIdx.x < 2) sl_qcnt[threadIdx.x] = 0u;      // (the first queueing is behind ln_search_core's first barrier)
    QcPool pool;
    pool.task = sl_task; pool.res = sl_res; pool.cnt = sl_qcnt; pool.rows = sl_lds; pool.wave_words = (uint32_t)wave_words; pool.cap = 2 * WAVE * SL_WAVES;
    bool active = s < n_surv;
    const uint64_t r = active ? surv_idx[s] : 0;
    if (active && rd_is_exc(R, r)) {                    // raw-byte read: the wave kernel's exception pass (err == 5)
        SurvOut o;
        o.found = 0; o.n_ss = 0; o.repeat_len = 0; o.ss_off = 0; o.dr_len = 0; o.low_lexi = 0; o.err = 5;
        out[s] = o;
        active = false;
    }
    const int L = active ? (int)rd_len(R, r) : 0;
    const uint32_t *g = R.packed + (active ? rd_word_off(R, r) : 0);
    const int nw = (L + 15) >> 4;
    // (a row is dword-aligned only, which global_load_dwordx4 takes on gfx950 — ff_load_row: three or four wide loads per lane
    // instead of ten to sixteen single words, every one of which was 64 lines' worth of look-ups for the wave)
    {
        int i = 0;
        for (; i + 4 <= nw; i += 4) {
            const ff_u32x4 v = *reinterpret_cast<const ff_u32x4 *>(g + i);
            lw[i * WAVE] = v.x; lw[(i + 1) * WAVE] = v.y; lw[(i + 2) * WAVE] = v.z; lw[(i + 3) * WAVE] = v.w;
        }
        for (; i < nw; i++) lw[i * WAVE] = g[i];
        for (; i < (int)words_per_read + 5; i++) lw[i * WAVE] = 0u;                             // (ln_load128 reads up to 4 words past a base)
    }
    LaneRead h;
    h.w = lw; h.ss = lss; h.L = L; h.nss = 0; h.cap = (int)ss_cap; h.replen = 0; h.punt = 0;
    h.cmask = (1u << (2 * P.window)) - 1u;
    const uint64_t hint = !active ? 0ull : hint_in_lds ? (0xFFFFFFFF00000000ull | (uint64_t)sl_hint[(uint32_t)(s - blockIdx.x * (uint64_t)blockDim.x)]) : ln_hint64(R, seed_hint, r, L, P.skips);
    int f = (P.debug_stop == 1) ? 0 : ln_search_core(h, P, hint, active, pool);      // (1: load only)
    if (!active) return;
    if (P.debug_stop == 4 && f == 1) f = 0;                           // (4: no orientation / output)
    SurvOut o;
    o.found = 0; o.n_ss = 0; o.repeat_len = 0; o.ss_off = 0; o.dr_len = 0; o.low_lexi = 0; o.err = 0;
    if (h.punt) { o.err = 4; if (punt_cnt) atomicAdd(punt_cnt, 1u); }      // (counted: the wave kernel's launch behind this one leaves at once when there is none)
    else if (f < 0) o.err = 1;
    else if (f == 1) {
        // ReadHolder::DRLowLexi (ReadHolder.cpp:513-591): representative repeat, orientation
        const int num_repeats = h.nss / 2;
        int pick;
        if (num_repeats == 1) pick = 0;
        else if (num_repeats == 2) {
            if (ln_ss(h, 0) == 0) pick = 2;
            else if (ln_ss(h, h.nss - 1) == (uint32_t)L) pick = 0;
            else pick = ((int)(ln_ss(h, 1) - ln_ss(h, 0)) > (int)(ln_ss(h, 3) - ln_ss(h, 2))) ? 0 : 2;
        } else pick = 2;
        uint32_t dlen = 0;
        const uint32_t st = ln_ss(h, pick);
        if (!substr_len(L, st, ln_ss(h, pick + 1) - st + 1, dlen) || dlen > dr_stride) o.err = 1;
        else {
            int less = 0;
            char *dr = dr_chars + s * (uint64_t)dr_stride;
            const uint32_t off = (uint32_t)s * ss_cap;
            if (dlen <= 64 && (dr_stride & 15u) == 0) {
                // the repeat as a 128-bit value, its reverse complement by bit reversal, DRLowLexi's string comparison as
                // "first differing base from the low end" (as k_recruit_finish), the ASCII string four bases per word
                uint64_t v0, v1;
                ln_load128(h, (int)st, v0, v1);
                const uint64_t m0 = dlen >= 32 ? ~0ull : ((1ull << (2 * dlen)) - 1ull);
                const uint64_t m1 = dlen >= 64 ? ~0ull : (dlen > 32 ? ((1ull << (2 * (dlen - 32))) - 1ull) : 0ull);
                v0 &= m0; v1 &= m1;
                auto rev2 = [](uint64_t t) -> uint64_t { t = __brevll(t); return ((t >> 1) & 0x5555555555555555ull) | ((t & 0x5555555555555555ull) << 1); };
                const uint64_t c0 = rev2(~v1), c1 = rev2(~v0);
                const uint32_t drop = 128u - 2u * dlen;
                uint64_t r0, r1;
                if (drop == 0) { r0 = c0; r1 = c1; }
                else if (drop < 64) { r0 = (c0 >> drop) | (c1 << (64 - drop)); r1 = c1 >> drop; }
                else { r0 = c1 >> (drop - 64); r1 = 0; }
                r0 &= m0; r1 &= m1;
                const uint64_t d0 = v0 ^ r0, d1 = v1 ^ r1;
                if (d0 | d1) {
                    const uint64_t dv = d0 ? d0 : d1, av = d0 ? v0 : v1, bv = d0 ? r0 : r1;
                    const int p = (__ffsll((unsigned long long)dv) - 1) & ~1;
                    less = ((av >> p) & 3ull) < ((bv >> p) & 3ull);
                }
                const uint64_t s0 = less ? v0 : r0, s1 = less ? v1 : r1;
                uint4 *d4 = reinterpret_cast<uint4 *>(dr);
                for (uint32_t q = 0; q < dr_stride / 16; q++) {             // 16 bases = 32 bits of the packed string per uint4
                    const uint32_t bits = q < 2 ? (uint32_t)(s0 >> (32 * q)) : (q < 4 ? (uint32_t)(s1 >> (32 * (q - 2))) : 0u);
                    uint32_t wv[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t b8 = (bits >> (8 * k)) & 0xFFu;
                        const uint32_t Ls = ((b8 & 0x55u) * 0x00041041u) & 0x01010101u, Hs = (((b8 >> 1) & 0x55u) * 0x00041041u) & 0x01010101u;
                        uint32_t asc = 0x41414141u + (Ls << 1) + Hs * 6u + (Ls & Hs) * 0x0Bu;       // A 41, C 43, G 47, T 54
                        const int rem = (int)dlen - (int)(q * 16 + k * 4);                           // bases left from this word on
                        if (rem <= 0) asc = 0u; else if (rem < 4) asc &= (1u << (8 * rem)) - 1u;
                        wv[k] = asc;
                    }
                    uint4 o4; o4.x = wv[0]; o4.y = wv[1]; o4.z = wv[2]; o4.w = wv[3];
                    d4[q] = o4;
                }
                if (less) for (int k = 0; k < h.nss; k++) ss_pool[off + k] = ln_ss(h, k);
                else for (int k = 0; k < h.nss; k++) ss_pool[off + k] = (uint32_t)L - 1 - ln_ss(h, h.nss - 1 - k);   // reverseStartStops
            } else {
            for (uint32_t i = 0; i < dlen; i++) {
                const uint32_t a = ln_base(h, (int)(st + i)), b = 3u - ln_base(h, (int)(st + dlen - 1 - i));
                if (a != b) { less = a < b; break; }
            }
            if (less) {
                for (uint32_t i = 0; i < dr_stride; i++) dr[i] = (i < dlen) ? "ACGT"[ln_base(h, (int)(st + i))] : (char)0;
                for (int k = 0; k < h.nss; k++) ss_pool[off + k] = ln_ss(h, k);
            } else {
                for (uint32_t i = 0; i < dr_stride; i++) dr[i] = (i < dlen) ? "ACGT"[3u - ln_base(h, (int)(st + dlen - 1 - i))] : (char)0;
                for (int k = 0; k < h.nss; k++) ss_pool[off + k] = (uint32_t)L - 1 - ln_ss(h, h.nss - 1 - k);   // reverseStartStops
            }
            }
            o.found = 1; o.n_ss = (uint32_t)h.nss; o.repeat_len = (uint32_t)h.replen; o.ss_off = off;
            o.dr_len = (uint16_t)dlen; o.low_lexi = (uint8_t)less;
            found_flag[rd_header_id(R, r)] = 1;
        }
    }
    out[s] = o;
}

hipError_t launch_survivor_lanes(const DevReads &R, const DevParams &P, const uint64_t *surv_idx, const uint32_t *d_n_surv,
                                 uint64_t n_surv_max, SurvOut *out, char *dr_chars, uint32_t dr_stride, uint32_t *ss_pool,
                                 uint32_t ss_cap, uint8_t *found_flag, const uint32_t *seed_hint, hipStream_t st, const DevMerge *init_merge,
                                 uint32_t max_len, uint32_t *punt_cnt)
{
    if (n_surv_max == 0) return init_merge ? hipErrorNotSupported : hipSuccess;
    // a lane holds its read's words in LDS: reads of up to 512 bases (one stride or not: the rows are addressed per read)
    const uint32_t wpr = R.stride_words ? R.stride_words : (max_len + 15) / 16;
    if (!wpr || wpr > 32 || ss_cap > 64) return hipErrorNotSupported;
    const size_t wave_words = (size_t)(wpr + 5) * WAVE + ((size_t)ss_cap * WAVE + 1) / 2;
    const size_t lds = wave_words * 4 * SL_WAVES;
    static const bool no_regroup = getenv("CRASS_SURV_NO_REGROUP") != nullptr;      // A/B switch: lanes in slot order
    const unsigned bt = WAVE * SL_WAVES;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_survivor_lanes), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    CRASS_LAUNCH(k_survivor_lanes, dim3((unsigned)((n_surv_max + bt - 1) / bt)), dim3(bt), lds, st, R, P, surv_idx, d_n_surv,
                       n_surv_max, out, dr_chars, dr_stride, ss_pool, ss_cap, found_flag, seed_hint, wpr, init_merge ? *init_merge : DevMerge{},
                       init_merge ? 1 : 0, ((seed_hint || R.pos_hint) && !no_regroup) ? 1 : 0, punt_cnt);
    return hipGetLastError();
}

// ---- device-side gather of the found records (fast path: short reads, slot-mode pool) ----
// mask of slots with found != 0; the worst error code is max-reduced into *d_err
__global__ __launch_bounds__(256) void k_found_mask(const SurvOut *out, const uint32_t *d_n, uint64_t n, uint64_t *mask, uint32_t *d_err,
                                                     unsigned long long *dd_keys, uint32_t *dd_first, uint32_t dd_size)
{
    const uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    // the de-duplication table of the next stage is cleared on the way (saves its own launch)
    for (uint64_t i = s; i < dd_size; i += (uint64_t)gridDim.x * blockDim.x) { dd_keys[i] = 0ull; dd_first[i] = 0xFFFFFFFFu; }
    bool f = false;
    if (s < n && s < (uint64_t)*d_n) {                  // slots past the device-side count were never written
        const SurvOut o = out[s];
        f = o.found != 0;
        if (o.err) atomicMax(d_err, (uint32_t)o.err);
    }
    const uint64_t m = __ballot(f);
    if ((threadIdx.x & 63) == 0 && s < n) mask[s >> 6] = m;
}

// ---- device-side de-duplication of the candidates' DR strings (single-GPU merge fast path) ----
// Same 64-bit hash as TokenTable::hash (merge.cpp) so the host can reuse it.  Every distinct
// string gets one table slot; `first` keeps the smallest candidate index (= first occurrence in
// read order).  The host re-checks every (candidate, representative) pair with memcmp, so a hash
// collision between different strings is detected and only costs the fast path.
static __device__ uint64_t dr_hash64(const char *p, uint32_t n)
{
    uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)n * 0xD6E8FEB86659FD93ull);
    while (n >= 8) {
        uint64_t v = 0;
        for (int i = 0; i < 8; i++) v |= (uint64_t)(uint8_t)p[i] << (8 * i);
        h = (h ^ v) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; p += 8; n -= 8;
    }
    if (n) {
        uint64_t v = 0;
        for (uint32_t i = 0; i < n; i++) v |= (uint64_t)(uint8_t)p[i] << (8 * i);
        h = (h ^ v) * 0xC4CEB9FE1A85EC53ull; h ^= h >> 29;
    }
    return h ^ (h >> 31);
}

// found records -> compact hand-off blob + dense DR strings on the device + de-duplication insert
// (one thread per found record; see launch_gather_found in engine_internal.h)
// The insert goes through LDS first: most records carry one of a few popular strings, every resident thread meets the table
// while it is still empty, and returning atomics on one address retire one every 20-30 ns — 5.6 k compare-and-swaps on each
// popular slot were ~100 of this kernel's 125 us at 100 M reads.  A block of 1 024 records claims its strings in an LDS table
// (hash, smallest record index), then ONE thread per distinct string of the block goes to the global table.
#define GF_BLOCK 1024
#define GF_SLOTS 2048
__global__ __launch_bounds__(GF_BLOCK) void k_gather_found(const uint64_t *fidx, const uint32_t *d_nf, uint64_t n_max,
                                                            const SurvOut *out, const uint64_t *surv_idx, uint64_t read_base,
                                                            const char *dr_chars, uint32_t dr_stride, const uint32_t *ss_pool,
                                                            uint32_t ss_cap, uint32_t ss_elem, uint8_t *blob, uint16_t *g_dr_len, char *g_dr,
                                                            unsigned long long *dd_keys, uint32_t *dd_first, uint32_t dd_mask,
                                                            uint64_t *dd_hash, uint32_t *dd_slot, uint32_t *d_mismatch)
{
    __shared__ unsigned long long lkey[GF_SLOTS];
    __shared__ uint32_t lmin[GF_SLOTS], lslot[GF_SLOTS];
    const uint64_t k = blockIdx.x * (uint64_t)GF_BLOCK + threadIdx.x;
    uint64_t n = *d_nf;
    if (n > n_max) n = n_max;
    if (blockIdx.x * (uint64_t)GF_BLOCK >= n) return;    // (the launch is sized for the survivor bound)
    if (dd_keys) {
        for (uint32_t i = threadIdx.x; i < GF_SLOTS; i += GF_BLOCK) { lkey[i] = 0ull; lmin[i] = 0xFFFFFFFFu; }
        __syncthreads();
    }
    uint32_t ls = 0;
    if (k < n) {
        const P1Blob b = p1_blob_layout(n, ss_cap, ss_elem);
        const uint64_t s = fidx[k];
        const SurvOut o = out[s];
        reinterpret_cast<uint64_t *>(blob + b.read)[k] = read_base + surv_idx[s];
        reinterpret_cast<uint16_t *>(blob + b.replen)[k] = (uint16_t)o.repeat_len;
        (blob + b.nss)[k] = (uint8_t)o.n_ss;
        (blob + b.low)[k] = o.low_lexi;
        const uint32_t *ps = ss_pool + o.ss_off;
        if (ss_elem == 1 && (o.ss_off & 3u) == 0u) {         // ss_cap is a multiple of 4: whole words, and 16-byte loads (slot-mode pool)
            uint32_t *pd = reinterpret_cast<uint32_t *>(blob + b.ss + k * (uint64_t)ss_cap);
            const uint4 *p4 = reinterpret_cast<const uint4 *>(ps);
            for (uint32_t i = 0; i < ss_cap; i += 4) {
                const uint4 x = p4[i >> 2];
                const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
                uint32_t v = 0;
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) v |= ((i + q < o.n_ss) ? (xs[q] & 0xFFu) : 0u) << (8 * q);
                pd[i >> 2] = v;
            }
        } else if (ss_elem == 1) {
            uint32_t *pd = reinterpret_cast<uint32_t *>(blob + b.ss + k * (uint64_t)ss_cap);
            for (uint32_t i = 0; i < ss_cap; i += 4) {
                uint32_t v = 0;
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) v |= ((i + q < o.n_ss) ? (ps[i + q] & 0xFFu) : 0u) << (8 * q);
                pd[i >> 2] = v;
            }
        } else {
            uint32_t *pd = reinterpret_cast<uint32_t *>(blob + b.ss + k * (uint64_t)ss_cap * 2);
            for (uint32_t i = 0; i < ss_cap; i += 2) {
                const uint32_t lo = (i < o.n_ss) ? (ps[i] & 0xFFFFu) : 0u, hi = (i + 1 < o.n_ss) ? (ps[i + 1] & 0xFFFFu) : 0u;
                pd[i >> 1] = lo | (hi << 16);
            }
        }
        g_dr_len[k] = o.dr_len;
        // the string is copied 16 bytes at a time and hashed from the same registers (dr_hash64 over a zero-padded slot: a
        // partial last word IS the value its byte loop assembles; 36 byte loads per record were a third of this kernel)
        const uint4 *src = reinterpret_cast<const uint4 *>(dr_chars + s * (uint64_t)dr_stride);
        uint4 *dst = reinterpret_cast<uint4 *>(g_dr + k * (uint64_t)dr_stride);
        uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)o.dr_len * 0xD6E8FEB86659FD93ull);
        uint32_t rem = o.dr_len;
        for (uint32_t i = 0; i < dr_stride / 16; i++) {
            const uint4 v4 = src[i];
            dst[i] = v4;
            const uint64_t w2[2] = {(uint64_t)v4.x | ((uint64_t)v4.y << 32), (uint64_t)v4.z | ((uint64_t)v4.w << 32)};
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (rem >= 8) { h = (h ^ w2[q]) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; rem -= 8; }
                else if (rem) { h = (h ^ w2[q]) * 0xC4CEB9FE1A85EC53ull; h ^= h >> 29; rem = 0; }
            }
        }
        h ^= h >> 31;
        if (dd_keys) {
            // (the same 64-bit hash as TokenTable::hash, merge.cpp); 0 marks an empty slot
            dd_hash[k] = h;
            const unsigned long long key = h | 1ull;
            ls = (uint32_t)(h >> 40) & (GF_SLOTS - 1u);
            for (;;) {                                       // (at most 1 024 distinct keys in 2 048 slots: always ends)
                const unsigned long long old = atomicCAS(&lkey[ls], 0ull, key);
                if (old == 0ull || old == key) break;
                ls = (ls + 1u) & (GF_SLOTS - 1u);
            }
            atomicMin(&lmin[ls], (uint32_t)k);
        }
    }
    if (!dd_keys) return;
    __syncthreads();
    // The global table was cleared by the found-flag compaction.  It is sized for the DISTINCT strings the caller expects
    // (a learnt bound), not for the records: a probe sequence that outlasts kDdMaxProbes means the bound was too small.
    // Bit 2 of the mismatch word tells the host (which then de-duplicates itself and sizes the next call's table for the
    // records); the string keeps the occupied slot it stopped at, so that everything downstream stays in range.
    // Looking at a slot before the CAS / atomicMin pays at 100 M reads but costs two more round trips at 10 M: done for
    // launches sized for more than 2^20 records (the headline workload).
    const bool look = n_max > (1ull << 20);
    for (uint32_t i = threadIdx.x; i < GF_SLOTS; i += GF_BLOCK) {
        const unsigned long long key = lkey[i];
        if (key == 0ull) continue;
        uint32_t slot = (uint32_t)(key >> 17) & dd_mask;
        for (uint32_t probes = 0;; probes++) {
            unsigned long long old = look ? __hip_atomic_load(&dd_keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            if (old == 0ull) old = atomicCAS(&dd_keys[slot], 0ull, key);
            if (old == 0ull || old == key) break;
            if (probes >= kDdMaxProbes) { atomicOr(d_mismatch, 2u); break; }
            slot = (slot + 1) & dd_mask;
        }
        const uint32_t kmin = lmin[i];
        if (!look || __hip_atomic_load(&dd_first[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > kmin) atomicMin(&dd_first[slot], kmin);
        lslot[i] = slot;
    }
    __syncthreads();
    if (k < n) dd_slot[k] = lslot[ls];
}

// k_found_mask + compaction in one pass (decoupled look-back): survivor slot s -> rank among the found records ->
// fidx[rank] = s.  Also clears the de-duplication table of the next stage.  (The gather itself stays a dense kernel:
// with one found record in six slots a fused body would run at a sixth of the lanes.)  Tiles of 16 384 slots (16 per
// thread); the launch is sized for the survivor BOUND, tiles past the device-side count leave at once.
static constexpr uint32_t kFcPerThread = 16, kFcTile = 1024u * kFcPerThread;
// Wave w of the block takes slots [w * 1024, (w + 1) * 1024) of the tile, 64 consecutive slots per step (one per lane: the
// loads of a step cover one contiguous 1 280-byte run); a step's found flags are one ballot.
__global__ __launch_bounds__(1024) void k_found_compact(const SurvOut *out, const uint32_t *d_n, uint64_t n_max, uint32_t *d_err,
                                                         unsigned long long *dd_keys, uint32_t *dd_first, uint32_t dd_size,
                                                         uint64_t *fidx, uint32_t *d_nf, Lookback lb)
{
    uint64_t n = *d_n;                                   // slots past the device-side count were never written
    if (n > n_max) n = n_max;
    const uint32_t n_act = n ? (uint32_t)((n + kFcTile - 1) / kFcTile) : 1u;
    if (blockIdx.x >= n_act) return;
    const uint32_t tile = lb_tile_id(lb, n_act);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t s0 = (uint64_t)tile * kFcTile + (uint64_t)wv * 1024u;
    for (uint64_t i = (uint64_t)tile * 1024u + threadIdx.x; i < dd_size; i += (uint64_t)n_act * 1024u) { dd_keys[i] = 0ull; dd_first[i] = 0xFFFFFFFFu; }
    uint32_t err = 0, cnt = 0;
    uint64_t fm[kFcPerThread];
#pragma unroll
    for (uint32_t e = 0; e < kFcPerThread; e++) {
        const uint64_t sl = s0 + e * 64u + (uint32_t)lane;
        bool f = false;
        if (sl < n) { f = out[sl].found != 0; err = max(err, (uint32_t)out[sl].err); }
        fm[e] = __ballot(f);
        cnt += (uint32_t)__popcll(fm[e]);                // (wave-uniform)
    }
    if (err) atomicMax(d_err, err);
    // ranks: the wave's base from a scan over the 16 wave totals, then step by step
    __shared__ uint32_t wtot[16];
    if (lane == 0) wtot[wv] = cnt;
    __syncthreads();
    uint32_t wbase = 0, all = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) { const uint32_t t = wtot[q]; if (q < wv) wbase += t; all += t; }
    const uint32_t excl = lb_exclusive_prefix(lb, tile, all);
    if (tile == n_act - 1 && threadIdx.x == 0) *d_nf = excl + all;
    uint64_t k = (uint64_t)excl + wbase;
    const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
    for (uint32_t e = 0; e < kFcPerThread; e++) {
        if ((fm[e] >> lane) & 1ull) fidx[k + (uint32_t)__popcll(fm[e] & lt)] = s0 + e * 64u + (uint32_t)lane;
        k += (uint32_t)__popcll(fm[e]);
    }
}
hipError_t launch_found_compact(const SurvOut *out, const uint32_t *d_n, uint64_t n_max, uint32_t *d_err, unsigned long long *dd_keys,
                                uint32_t *dd_first, uint32_t dd_size, uint64_t *fidx, uint32_t *d_nf, const Lookback &lb, hipStream_t st)
{
    if (n_max == 0) return hipSuccess;
    const uint32_t n_tiles = (uint32_t)((n_max + kFcTile - 1) / kFcTile);
    CRASS_LAUNCH(k_found_compact, dim3(n_tiles), dim3(1024), 0, st, out, d_n, n_max, d_err, dd_keys, dd_first, dd_keys ? dd_size : 0u,
                       fidx, d_nf, lb);
    return hipGetLastError();
}

// ---- host-loop sink: select + gather of the found records (engine_internal.h) ----
__global__ __launch_bounds__(256) void k_select_found(const SurvOut *out, uint64_t n, uint64_t *mask, uint32_t *d_err)
{
    const uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    bool f = false;
    if (s < n) {
        const SurvOut o = out[s];
        f = o.found != 0 && o.err == 0;
        if (o.err && o.err != 5) atomicMax(d_err, o.err == 1 ? 2u : 1u);
    }
    const uint64_t m = __ballot(f);
    if ((threadIdx.x & 63) == 0 && s < n) mask[s >> 6] = m;
}
__global__ __launch_bounds__(256) void k_gather_sparse(const uint64_t *fidx, const uint32_t *d_nf, uint64_t n_max, const SurvOut *out, const char *dr_chars,
                                                        uint32_t dr_stride, const uint32_t *ss_pool, SurvOut *g_out, uint64_t *g_slot, char *g_dr,
                                                        uint32_t *g_ss, uint32_t g_ss_cap, uint32_t *d_ss_total, uint16_t *g_dr_len, int ss16)
{
    const uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t n = *d_nf;
    if (n > n_max) n = n_max;
    SurvOut o; o.found = 0; o.n_ss = 0; o.repeat_len = 0; o.ss_off = 0; o.dr_len = 0; o.low_lexi = 0; o.err = 0;
    uint64_t s = 0;
    if (k < n) { s = fidx[k]; o = out[s]; }
    const uint32_t off = block_reserve<256>(k < n ? o.n_ss : 0u, d_ss_total);      // (every thread of the block)
    if (k >= n) return;
    // (ss16: every position of the set fits 16 bits — the packed pool then travels in half the bytes: 50 k records with 80
    // start/stops each are 16 MB of a long-read step's 20 MB of copies)
    if ((uint64_t)off + o.n_ss <= g_ss_cap) {
        if (ss16) { uint16_t *g16 = reinterpret_cast<uint16_t *>(g_ss); for (uint32_t i = 0; i < o.n_ss; i++) g16[off + i] = (uint16_t)ss_pool[o.ss_off + i]; }
        else for (uint32_t i = 0; i < o.n_ss; i++) g_ss[off + i] = ss_pool[o.ss_off + i];
    }
    o.ss_off = off;
    g_out[k] = o;
    g_slot[k] = s;
    if (g_dr_len) g_dr_len[k] = o.dr_len;               // (dense lengths: the de-duplication that may follow on the device)
    const char *src = dr_chars + s * (uint64_t)dr_stride;
    char *dst = g_dr + k * (uint64_t)dr_stride;
    for (uint32_t i = 0; i < dr_stride; i++) dst[i] = src[i];
}
hipError_t launch_select_found(const SurvOut *out, uint64_t n, uint64_t *mask, uint32_t *d_err, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    CRASS_LAUNCH(k_select_found, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n, mask, d_err);
    return hipGetLastError();
}
hipError_t launch_gather_sparse(const uint64_t *fidx, const uint32_t *d_nf, uint64_t n_max, const SurvOut *out, const char *dr_chars, uint32_t dr_stride,
                                const uint32_t *ss_pool, SurvOut *g_out, uint64_t *g_slot, char *g_dr, uint32_t *g_ss, uint32_t g_ss_cap,
                                uint32_t *d_ss_total, hipStream_t st, uint16_t *g_dr_len, int ss16)
{
    if (n_max == 0) return hipSuccess;
    CRASS_LAUNCH(k_gather_sparse, dim3((unsigned)((n_max + 255) / 256)), dim3(256), 0, st, fidx, d_nf, n_max, out, dr_chars, dr_stride, ss_pool,
                       g_out, g_slot, g_dr, g_ss, g_ss_cap, d_ss_total, g_dr_len, ss16);
    return hipGetLastError();
}

hipError_t launch_found_mask(const SurvOut *out, const uint32_t *d_n, uint64_t n, uint64_t *mask, uint32_t *d_err, hipStream_t st,
                             unsigned long long *dd_keys, uint32_t *dd_first, uint32_t dd_size)
{
    if (n == 0) return hipSuccess;
    CRASS_LAUNCH(k_found_mask, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, d_n, n, mask, d_err, dd_keys, dd_first, dd_keys ? dd_size : 0u);
    return hipGetLastError();
}

hipError_t launch_gather_found(const uint64_t *fidx, const uint32_t *d_nf, uint64_t n_max, const SurvOut *out,
                               const uint64_t *surv_idx, uint64_t read_base, const char *dr_chars, uint32_t dr_stride,
                               const uint32_t *ss_pool, uint32_t ss_cap, uint32_t ss_elem, uint8_t *h_blob,
                               uint16_t *g_dr_len, char *g_dr, hipStream_t st,
                               unsigned long long *dd_keys, uint32_t *dd_first, uint32_t dd_size, uint64_t *dd_hash, uint32_t *dd_slot,
                               uint32_t *d_mismatch)
{
    if (n_max == 0) return hipSuccess;
    if ((ss_cap & 3u) || (ss_elem != 1 && ss_elem != 2)) return hipErrorInvalidValue;
    CRASS_LAUNCH(k_gather_found, dim3((unsigned)((n_max + GF_BLOCK - 1) / GF_BLOCK)), dim3(GF_BLOCK), 0, st, fidx, d_nf, n_max, out, surv_idx,
                       read_base, dr_chars, dr_stride, ss_pool, ss_cap, ss_elem, h_blob, g_dr_len, g_dr,
                       dd_keys, dd_first, dd_keys ? dd_size - 1 : 0u, dd_hash, dd_slot, d_mismatch);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_dr_dedupe_clear(unsigned long long *keys, uint32_t *first, uint32_t table_size)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < table_size; i += gridDim.x * blockDim.x) { keys[i] = 0ull; first[i] = 0xFFFFFFFFu; }
}

__global__ __launch_bounds__(256) void k_dr_dedupe_insert(const char *dr, const uint16_t *dr_len, uint32_t stride, const uint32_t *d_n,
                                                           uint32_t n_max, unsigned long long *keys, uint32_t *first, uint32_t mask,
                                                           uint64_t *hash_out, uint32_t *slot_out)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = min(*d_n, n_max);
    if (k >= n) return;
    const uint64_t h = dr_hash64(dr + (uint64_t)k * stride, dr_len[k]);
    hash_out[k] = h;
    const unsigned long long key = h | 1ull;                 // 0 marks an empty slot
    uint32_t slot = (uint32_t)(h >> 17) & mask;
    for (;;) {
        const unsigned long long old = atomicCAS(&keys[slot], 0ull, key);
        if (old == 0ull || old == key) break;
        slot = (slot + 1) & mask;
    }
    atomicMin(&first[slot], k);
    slot_out[k] = slot;
}

// the candidate count lives on the device (*d_n, at most n_max): no host round trip before this launch
hipError_t launch_dr_dedupe(const char *dr, const uint16_t *dr_len, uint32_t stride, const uint32_t *d_n, uint32_t n, unsigned long long *keys,
                            uint32_t *first, uint32_t table_size, uint64_t *hash_out, uint32_t *slot_tmp, uint32_t *rep, hipStream_t st,
                            bool table_cleared)
{
    if (n == 0) return hipSuccess;
    if (!table_cleared) CRASS_LAUNCH(k_dr_dedupe_clear, dim3((unsigned)std::min<uint32_t>((table_size + 255) / 256, 2048u)), dim3(256), 0, st, keys, first, table_size);
    const unsigned nb = (n + 255) / 256;
    CRASS_LAUNCH(k_dr_dedupe_insert, dim3(nb), dim3(256), 0, st, dr, dr_len, stride, d_n, n, keys, first, table_size - 1, hash_out, slot_tmp);
    (void)rep;                                  // rep[] = first occurrence of every candidate: written by k_dx_flag
    return hipGetLastError();
}

// ---- device-side token ranks: distinct strings in first-occurrence order ----
// bit k of `mask` = candidate k is the first occurrence of its string; every other candidate is compared
// byte for byte with its representative, so a 64-bit hash collision between different strings is
// DETECTED (flag) and the host then takes its plain path.
__global__ __launch_bounds__(256) void k_dx_flag(const char *dr, const uint16_t *dr_len, uint32_t stride, const uint32_t *d_n, uint32_t n_max,
                                                  const uint32_t *slot_of, const uint32_t *first, uint32_t *rep, uint64_t *mask, uint32_t *d_mismatch)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n = min(*d_n, n_max);
    bool is_rep = false;
    if (k < n) {
        const uint32_t f = first[slot_of[k]];               // (was k_dr_dedupe_rep: one launch less)
        rep[k] = f;
        is_rep = (f == k);
        if (!is_rep) {
            bool same = f < k && dr_len[f] == dr_len[k];
            if (same) {
                const uint4 *a = reinterpret_cast<const uint4 *>(dr + (uint64_t)k * stride);
                const uint4 *b = reinterpret_cast<const uint4 *>(dr + (uint64_t)f * stride);
                for (uint32_t i = 0; i < stride / 16; i++) {          // slots are zero padded: whole-slot compare
                    const uint4 x = a[i], y = b[i];
                    same = same && x.x == y.x && x.y == y.y && x.z == y.z && x.w == y.w;
                }
            }
            if (!same) atomicOr(d_mismatch, 1u);
        }
    }
    const uint64_t m = __ballot(is_rep);
    if ((threadIdx.x & 63) == 0 && k < n_max) mask[k >> 6] = m;       // words past the count are zero
}

// dmap[k] = rank of k's representative among the first occurrences (token = rank + 2 on one GPU)
__global__ __launch_bounds__(256) void k_dx_assign(const uint32_t *rep, const uint32_t *d_n, uint32_t n_max, const uint64_t *mask,
                                                    const uint32_t *word_prefix, const uint32_t *block_sums, uint32_t *dmap)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= min(*d_n, n_max)) return;
    const uint32_t f = rep[k], w = f >> 6;
    dmap[k] = block_sums[w >> 8] + word_prefix[w] + (uint32_t)__popcll(mask[w] & ((1ull << (f & 63)) - 1ull));
}

__global__ __launch_bounds__(256) void k_dx_gather(const uint64_t *dx_idx, const uint32_t *d_nd, uint32_t n_max, const char *dr,
                                                    const uint16_t *dr_len, const uint64_t *hash, uint32_t stride, char *out_chars,
                                                    uint16_t *out_len, uint64_t *out_hash, char *dev_chars, uint16_t *dev_len,
                                                    const uint32_t *cnt_src, uint32_t *cnt_dst, uint32_t n_cnt)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_cnt) cnt_dst[j] = cnt_src[j];             // the stage's counters, straight into pinned host memory
    uint32_t nd = *d_nd;
    if (nd > n_max) nd = n_max;
    if (j >= nd) return;
    const uint64_t k = dx_idx[j];
    const uint4 *src = reinterpret_cast<const uint4 *>(dr + k * stride);
    uint4 *dst = reinterpret_cast<uint4 *>(out_chars + (uint64_t)j * stride);
    uint4 *dst2 = reinterpret_cast<uint4 *>(dev_chars + (uint64_t)j * stride);      // device copy for the device merge
    // (out_*: pinned host memory, or nullptr when nobody on the host reads the list — the device merge exports its own view)
    for (uint32_t i = 0; i < stride / 16; i++) { const uint4 v = src[i]; if (out_chars) dst[i] = v; if (dev_chars) dst2[i] = v; }
    const uint16_t l = dr_len[k];
    if (out_len) out_len[j] = l;
    if (dev_len) dev_len[j] = l;
    if (out_hash) out_hash[j] = hash[k];
}

// k_dx_flag + compaction in one pass (decoupled look-back over tiles of 1024 candidates, one per thread: the body is a
// chain of dependent loads, so it wants many blocks rather than fat ones): candidate k is a first occurrence iff
// first[slot_of[k]] == k; dx_idx[rank] = k, and the representative's rank is left in slot_of[k] (every thread only
// ever reads its OWN slot_of entry here, so overwriting it is safe) for the assign kernel that follows.
__global__ __launch_bounds__(1024) void k_dx_flag_compact(const char *dr, const uint16_t *dr_len, uint32_t stride, const uint32_t *d_n, uint32_t n_max,
                                                           uint32_t *slot_of, const uint32_t *first, uint32_t *rep, uint64_t *dx_idx, uint32_t *d_nd,
                                                           uint32_t *d_mismatch, Lookback lb)
{
    const uint32_t n = min(*d_n, n_max);
    const uint32_t n_tiles = n ? (n + 1023u) / 1024u : 1u;          // (the launch is sized for a bound: the tiles past the count leave at once)
    if (blockIdx.x >= n_tiles) return;
    const uint32_t tile = lb_tile_id(lb, n_tiles);
    const uint32_t k = tile * 1024u + threadIdx.x;
    bool is_rep = false;
    if (k < n) {
        const uint32_t f = first[slot_of[k]];
        rep[k] = f;
        is_rep = (f == k);
        if (!is_rep) {
            bool same = f < k && dr_len[f] == dr_len[k];
            if (same) {
                const uint4 *a = reinterpret_cast<const uint4 *>(dr + (uint64_t)k * stride);
                const uint4 *b = reinterpret_cast<const uint4 *>(dr + (uint64_t)f * stride);
                for (uint32_t i = 0; i < stride / 16; i++) {          // slots are zero padded: whole-slot compare
                    const uint4 x = a[i], y = b[i];
                    same = same && x.x == y.x && x.y == y.y && x.z == y.z && x.w == y.w;
                }
            }
            if (!same) atomicOr(d_mismatch, 1u);
        }
    }
    uint32_t all;
    const uint32_t in_tile = block_scan_t<1024>(is_rep ? 1u : 0u, &all);
    const uint32_t excl = lb_exclusive_prefix(lb, tile, all);
    if (tile == n_tiles - 1 && threadIdx.x == 0) *d_nd = excl + all;
    if (is_rep) { const uint32_t q = excl + in_tile; dx_idx[q] = k; slot_of[k] = q; }
}

// dense: every candidate's rank (dmap = rank of its representative) and, for the first nd threads, the distinct string's slot
__global__ __launch_bounds__(256) void k_dx_assign_gather(const uint32_t *rep, const uint32_t *d_n, uint32_t n_max, const uint32_t *rank_of,
                                                           uint32_t *dmap, const uint64_t *dx_idx, const uint32_t *d_nd, const char *dr,
                                                           const uint16_t *dr_len, const uint64_t *hash, uint32_t stride, char *out_chars,
                                                           uint16_t *out_len, uint64_t *out_hash, char *dev_chars, uint16_t *dev_len,
                                                           const uint32_t *cnt_src, uint32_t *cnt_dst, uint32_t n_cnt)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_cnt) cnt_dst[j] = cnt_src[j];             // the stage's counters, straight into pinned host memory
    if (j < min(*d_n, n_max)) dmap[j] = rank_of[rep[j]];
    uint32_t nd = *d_nd;
    if (nd > n_max) nd = n_max;
    if (j >= nd) return;
    const uint64_t k = dx_idx[j];
    const uint4 *src = reinterpret_cast<const uint4 *>(dr + k * stride);
    uint4 *dst = reinterpret_cast<uint4 *>(out_chars + (uint64_t)j * stride);
    uint4 *dst2 = reinterpret_cast<uint4 *>(dev_chars + (uint64_t)j * stride);      // device copy for the device merge
    // (out_*: pinned host memory, or nullptr when nobody on the host reads the list — the device merge exports its own view)
    for (uint32_t i = 0; i < stride / 16; i++) { const uint4 v = src[i]; if (out_chars) dst[i] = v; if (dev_chars) dst2[i] = v; }
    const uint16_t l = dr_len[k];
    if (out_len) out_len[j] = l;
    if (dev_len) dev_len[j] = l;
    if (out_hash) out_hash[j] = hash[k];
}

// needs stride % 16 == 0; mask / word_prefix / block_sums / dx_idx are scratch of >= n bits / words.  The candidate
// count is *d_n (<= n).  dmap / out_* may be pinned host memory: the kernels then write the merge's inputs
// straight into it (a few hundred KB; no copy calls on the critical path).
hipError_t launch_dx_tokens(const char *dr, const uint16_t *dr_len, const uint64_t *hash, uint32_t stride, const uint32_t *d_n, uint32_t n, uint32_t *rep,
                            uint32_t *slot_of, const uint32_t *first,
                            uint64_t *mask, uint32_t *word_prefix, uint32_t *block_sums, uint64_t *dx_idx, uint32_t *d_nd,
                            uint32_t *d_mismatch, uint32_t *dmap, char *out_chars, uint16_t *out_len, uint64_t *out_hash,
                            char *dev_chars, uint16_t *dev_len, hipStream_t st, const uint32_t *cnt_src, uint32_t *cnt_dst, uint32_t n_cnt,
                            const Lookback *lb)
{
    if (n == 0) return hipSuccess;
    const unsigned nb = (n + 255) / 256;
    if (lb) {           // two launches: flags + single-pass compaction (element-wise look-back), dense assign + gather
        const uint32_t n_tiles = (n + 1023u) / 1024u;                               // (the caller reserved that many tickets)
        CRASS_LAUNCH(k_dx_flag_compact, dim3(n_tiles), dim3(1024), 0, st, dr, dr_len, stride, d_n, n, slot_of, first, rep, dx_idx, d_nd, d_mismatch,
                           *lb);
        CRASS_LAUNCH(k_dx_assign_gather, dim3(nb), dim3(256), 0, st, rep, d_n, n, (const uint32_t *)slot_of, dmap, dx_idx, d_nd, dr, dr_len,
                           hash, stride, out_chars, out_len, out_hash, dev_chars, dev_len, cnt_src, cnt_dst, cnt_dst ? n_cnt : 0u);
        return hipGetLastError();
    }
    CRASS_LAUNCH(k_dx_flag, dim3(nb), dim3(256), 0, st, dr, dr_len, stride, d_n, n, slot_of, first, rep, mask, d_mismatch);
    hipError_t e = launch_compact(mask, (n + 63) / 64, n, word_prefix, block_sums, dx_idx, n, d_nd, st);
    if (e != hipSuccess) return e;
    CRASS_LAUNCH(k_dx_assign, dim3(nb), dim3(256), 0, st, rep, d_n, n, mask, word_prefix, block_sums, dmap);
    CRASS_LAUNCH(k_dx_gather, dim3(nb), dim3(256), 0, st, dx_idx, d_nd, n, dr, dr_len, hash, stride, out_chars, out_len, out_hash, dev_chars, dev_len,
                       cnt_src, cnt_dst, cnt_dst ? n_cnt : 0u);
    return hipGetLastError();
}

SurvLds survivor_lds_layout(uint32_t max_len, const DevParams &P, uint32_t row_len_cap, uint32_t seq_window_bytes, uint32_t ss_entries_cap)
{
    SurvLds l;
    l.seq_bytes = ((max_len + 16 + 16) + 15u) & ~15u;
    l.seq_window = 0;
    if (seq_window_bytes && ((seq_window_bytes + 15u) & ~15u) + 32u < l.seq_bytes) { l.seq_bytes = ((seq_window_bytes + 15u) & ~15u) + 32u; l.seq_window = l.seq_bytes - 32u; }
    uint32_t reps = max_len / (P.window + P.lowSp) + 4;
    l.ss_cap = ((2 * reps) + 3u) & ~3u;
    l.ss_slot = l.ss_cap;
    if (ss_entries_cap && ((ss_entries_cap + 3u) & ~3u) < l.ss_cap) l.ss_cap = (ss_entries_cap + 3u) & ~3u;
    l.row_elems = ((std::min(max_len, row_len_cap) + 8) + 7u) & ~7u;
    l.words_cap = (((max_len + 15) / 16 + 2) + 3u) & ~3u;
    l.hint_words = (((max_len + 63) / 64 + 2 + 1u) & ~1u) + 4u;      // + 8 x uint32: where each residue class's hint bits start (search_core)
    l.total_bytes = l.seq_bytes + l.ss_cap * 4 + 2 * l.row_elems * 2 + l.words_cap * 4 + l.ss_cap * 4 + l.hint_words * 8;
    return l;
}

hipError_t launch_survivor(const DevReads &R, const DevParams &P, bool exceptions, const uint64_t *surv_idx,
                           const uint32_t *d_n_surv, uint64_t n_surv_max, SurvOut *out, char *dr_chars,
                           uint32_t dr_stride, uint32_t *ss_pool, uint32_t ss_pool_cap, uint32_t *d_ss_used,
                           uint8_t *found_flag, const uint32_t *seed_hint, const SurvLds &lds, int grid, hipStream_t st,
                           int punt_only, uint64_t slot_base, uint64_t slot_total, const uint32_t *punt_list, const uint32_t *d_punt_n, uint32_t *redo_list)
{
    if (n_surv_max == 0) return hipSuccess;
    hipError_t e;
    const uint32_t lds_bytes = ((lds.total_bytes + 7u) & ~7u) + (P.prof ? PF_LDS_WORDS * 8u : 0u);      // (diagnostics: the phase slots)
    if (exceptions) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_survivor<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        CRASS_LAUNCH(k_survivor<true>, dim3(grid), dim3(WAVE), lds_bytes, st, R, P, surv_idx, d_n_surv, n_surv_max,
                           out, dr_chars, dr_stride, ss_pool, ss_pool_cap, d_ss_used, found_flag, seed_hint, lds, punt_only, slot_base, slot_total, punt_list, d_punt_n, redo_list);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_survivor<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        static const bool occ_dbg = getenv("CRASS_OCC_DEBUG") != nullptr;
        if (occ_dbg) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(&k_survivor<false>), WAVE, lds_bytes);
            fprintf(stderr, "[occ] k_survivor<false>: %d blocks per CU with %u bytes of LDS, grid %d, punt %d\n", nb, lds_bytes, grid, punt_only);
        }
        CRASS_LAUNCH(k_survivor<false>, dim3(grid), dim3(WAVE), lds_bytes, st, R, P, surv_idx, d_n_surv, n_surv_max,
                           out, dr_chars, dr_stride, ss_pool, ss_pool_cap, d_ss_used, found_flag, seed_hint, lds, punt_only, slot_base, slot_total, punt_list, d_punt_n, redo_list);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// pass 2: first-match multi-pattern scan (findSingletons/on_match semantics: the first ACISM
// callback = occurrence with the smallest end position, ties -> longest pattern;
// libcrispr.cpp:441, acism.c:73-102).  Lane per read, 64 consecutive reads per wave so the
// ballot is the mask word.  hit_info[r] = (end_exclusive << 8) | pattern_length.
// ------------------------------------------------------------------------------------
template <bool LDS_TABLE, int THREADS>
__global__ __launch_bounds__(THREADS) void k_recruit(DevReads R, DevAutomaton A, const uint8_t *found_flag,
                                                 uint64_t *hitmask, uint32_t *hit_info)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t rc_lds[];
    const uint16_t *go4 = A.go4;
    const uint16_t *outl = A.out_len;
    if (LDS_TABLE) {
        // stage [n_states][4] transitions + out_len in LDS
        uint16_t *l_go = rc_lds;
        uint16_t *l_out = rc_lds + (size_t)A.n_states * 4;
        for (uint32_t i = threadIdx.x; i < A.n_states * 4; i += blockDim.x) l_go[i] = A.go4[i];
        for (uint32_t i = threadIdx.x; i < A.n_states; i += blockDim.x) l_out[i] = A.out_len[i];
        __syncthreads();
        go4 = l_go; outl = l_out;
    }
    const uint64_t n_tiles = (R.n_reads + 63) / 64;
    const int lane = threadIdx.x & 63;
    const uint64_t wave_global = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t wave_total = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t tile = wave_global; tile < n_tiles; tile += wave_total) {
        const uint64_t r = tile * 64 + lane;
        bool hit = false;
        if (r < R.n_reads && !rd_is_exc(R, r) && !found_flag[rd_header_id(R, r)]) {
            const uint32_t L = rd_len(R, r);
            const uint32_t *g = R.packed + rd_word_off(R, r);
            uint32_t state = 0;
            uint32_t word = 0;
            for (uint32_t i = 0; i < L; i++) {
                if ((i & 15u) == 0) word = g[i >> 4];
                uint32_t c = word & 3u;
                word >>= 2;
                state = go4[state * 4 + c];
                uint32_t ol = outl[state];
                if (ol) { hit_info[r] = ((i + 1) << 8) | ol; hit = true; break; }
            }
        }
        uint64_t m = __ballot(hit);
        if (lane == 0) hitmask[tile] = m;
    }
}

// generic transition tables (any symbol count / state count), global memory
__global__ __launch_bounds__(256) void k_recruit_wide(DevReads R, DevAutomaton A, const uint8_t *found_flag,
                                                       uint64_t *hitmask, uint32_t *hit_info)
{
    const uint64_t n_tiles = (R.n_reads + 63) / 64;
    const int lane = threadIdx.x & 63;
    const uint64_t wave_global = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t wave_total = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t symA = A.sym['A'], symC = A.sym['C'], symG = A.sym['G'], symT = A.sym['T'];
    for (uint64_t tile = wave_global; tile < n_tiles; tile += wave_total) {
        const uint64_t r = tile * 64 + lane;
        bool hit = false;
        if (r < R.n_reads && !rd_is_exc(R, r) && !found_flag[rd_header_id(R, r)]) {
            const uint32_t L = rd_len(R, r);
            const uint32_t *g = R.packed + rd_word_off(R, r);
            uint32_t state = 0, word = 0;
            for (uint32_t i = 0; i < L; i++) {
                if ((i & 15u) == 0) word = g[i >> 4];
                uint32_t c = word & 3u;
                word >>= 2;
                uint32_t sy = c == 0 ? symA : c == 1 ? symC : c == 2 ? symG : symT;
                state = A.go16 ? (uint32_t)A.go16[(size_t)state * A.n_sym1 + sy] : A.go32[(size_t)state * A.n_sym1 + sy];
                uint32_t ol = A.out_len[state];
                if (ol) { hit_info[r] = ((i + 1) << 8) | ol; hit = true; break; }
            }
        }
        uint64_t m = __ballot(hit);
        if (lane == 0) hitmask[tile] = m;
    }
}

hipError_t launch_recruit_general(const DevReads &R, const DevAutomaton &A, const uint8_t *found_flag,
                                  uint64_t *hitmask, uint32_t *hit_info, hipStream_t st)
{
    if (R.n_reads == 0) return hipSuccess;
    uint64_t n_tiles = (R.n_reads + 63) / 64;
    uint64_t blocks = (n_tiles + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (A.acgt_ok && A.go4)
        CRASS_LAUNCH((k_recruit<false, 256>), dim3((unsigned)blocks), dim3(256), 0, st, R, A, found_flag, hitmask, hit_info);
    else
        CRASS_LAUNCH(k_recruit_wide, dim3((unsigned)blocks), dim3(256), 0, st, R, A, found_flag, hitmask, hit_info);
    return hipGetLastError();
}

hipError_t launch_recruit_lds(const DevReads &R, const DevAutomaton &A, const uint8_t *found_flag,
                              uint64_t *hitmask, uint32_t *hit_info, hipStream_t st)
{
    if (R.n_reads == 0) return hipSuccess;
    if (!A.acgt_ok || !A.go4) return hipErrorNotSupported;
    size_t lds = (size_t)A.n_states * 10;       // 4 x u16 transitions + u16 out_len
    if (lds > 160 * 1024) return hipErrorNotSupported;
    uint64_t n_tiles = (R.n_reads + 63) / 64;
    // one workgroup per CU-slot; the LDS footprint decides how many fit, so size the block to fill the CU
    const int threads = lds > 80 * 1024 ? 1024 : (lds > 40 * 1024 ? 512 : 256);
    uint64_t waves_per_block = threads / 64;
    uint64_t blocks = (n_tiles + waves_per_block - 1) / waves_per_block;
    uint64_t cap = lds > 80 * 1024 ? 256 : (lds > 40 * 1024 ? 512 : (lds > 20 * 1024 ? 1024 : 2048));
    if (blocks > cap) blocks = cap;
    hipError_t e;
#define RC_LAUNCH(T)                                                                                                   \
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_recruit<true, T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return e;                                                                                     \
    CRASS_LAUNCH((k_recruit<true, T>), dim3((unsigned)blocks), dim3(T), lds, st, R, A, found_flag, hitmask, hit_info);
    if (threads == 1024) { RC_LAUNCH(1024) }
    else if (threads == 512) { RC_LAUNCH(512) }
    else { RC_LAUNCH(256) }
#undef RC_LAUNCH
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// pass 2 fast path: anchor filter + exact verification of the flagged reads.
//
// Every pattern has length >= 23.  If pattern P occurs at offset o of a read, let a be the
// smallest multiple of 8 with a >= o (a <= o+7): bases [a, a+16) lie inside the occurrence
// (a+16 <= o+23 <= o+|P|) and equal P[a-o .. a-o+16).  So the halfword-aligned 32-bit window
// of the read at base a is one of the keys {P[r..r+16) : r = 0..7}.  The filter probes every
// aligned window (ceil(L/8)-1 per read, all independent) in an exact LDS hash set of the keys:
// no false negatives by construction; the rare false positives (a key occurring by chance,
// ~n_keys * L/8 / 4^16) are removed by the exact automaton scan of the flagged reads
// (k_recruit_list), which also yields ACISM's first-callback (end, length).
// ------------------------------------------------------------------------------------
// MODE 0: exact keys in LDS, 1: buckets of two 16-bit fingerprints in LDS, 2: exact keys in global memory
template <int MODE>
static __device__ __forceinline__ bool anchor_probe(const uint32_t *tab, uint32_t V, const DevAnchors &K, uint32_t rshift)
{
    const uint32_t h1 = ak_hash(V, K.m1);
    const uint32_t h2 = ak_hash(V, K.m2);
    const uint32_t a = tab[h1 >> rshift], b = tab[h2 >> rshift];   // both probes always issued: independent reads, no branches
    if (MODE == 1) {
        // fingerprint = HIGH halfword of h1 ^ h2 (the bits that depend on every base of the key; the low halfword only
        // sees the first eight, see anchor_probe_fp), replicated into both halves
        const uint32_t hx = h1 ^ h2;
        const uint32_t ff = __builtin_amdgcn_perm(hx, hx, 0x03020302u);
        // a halfword of (slot ^ ff) is zero <=> that fingerprint matches; min(x, 1) per halfword keeps 1 unless zero
        const uint32_t t = pk_min_u16(a ^ ff, 0x00010001u) & pk_min_u16(b ^ ff, 0x00010001u);
        return t != 0x00010001u;
    }
    return (a == V) | (b == V);
}
// MODE 4 (device-built tables beyond the LDS tiers): 2^20-bit Bloom filter in LDS, exact keys in global memory
static __device__ __forceinline__ bool anchor_probe_bloom(const uint32_t *bloom, const uint32_t *gtab, uint32_t V, const DevAnchors &K, uint32_t rshift)
{
    const uint32_t h1 = ak_hash(V, K.m1);
    const uint32_t wd = bloom[ak_bloom_word(h1)];
    bool hit = false;
    if (((wd >> ((h1 >> 12) & 31u)) & (wd >> ((h1 >> 7) & 31u)) & 1u) != 0u)          // ~7 % of the probes at 150 k keys
        hit = (gtab[h1 >> rshift] == V) | (gtab[ak_hash(V, K.m2) >> rshift] == V);
    return hit;
}
// MODE 3 (device-built tables): 2^16 slots, 16 bits per slot in LDS — the OTHER slot index of the key that sits there
// (partial-key cuckoo: slot h1(K) stores h2(K) and the other way round), so a window matches when one of its two slots
// names the other.  Sixteen bits that depend on every base of the key through an independent hash; the low halfword of
// h1 ^ h2 — the first form — only depends on the key's first eight bases, which the hundreds of variants of one repeat
// share: every read carrying a near-copy of a repeat then met that value in ~30 slots instead of one (k_dm_verify
// 364 -> 464 us at 100 M reads, profiles/NOTES_r03.md).
static __device__ __forceinline__ bool anchor_probe_fp(const uint16_t *tab, uint32_t V, const DevAnchors &K)
{
    const uint32_t i1 = ak_hash(V, K.m1) >> 16, i2 = ak_hash(V, K.m2) >> 16;
    const uint32_t a = tab[i1], b = tab[i2];
    return (a == i2) | (b == i1);
}

// ASH: log2 of the windows' alignment — 3: every 8 bases (halfword positions; patterns of >= 23 bases), 2: every 4 bases (byte
// positions; patterns of 19 .. 22 bases, `-d 19` .. `-d 22`: twice the windows per read, see kDevMinDR)
template <int W, int THREADS, int MODE, int ASH = 3>     // W = uniform stride in words (0: ragged / any stride)
static __device__ __forceinline__ void anchor_filter_body(const DevReads &R, const DevAnchors &K, const uint32_t *ak_lds,
                                                          const uint8_t *found_flag, uint64_t *hitmask)
{
    constexpr uint32_t PW = 16u >> ASH;              // windows per packed word (2 or 4)
    constexpr uint32_t WB = 32u / PW;                // bits between two windows (16 or 8)
    const uint32_t mask = 32u - K.log_size;          // right shift that keeps the top log_size bits
    const uint64_t n_tiles = (R.n_reads + 63) / 64;
    const int lane = threadIdx.x & 63;
    const uint64_t wave_global = (blockIdx.x * (uint64_t)THREADS + threadIdx.x) >> 6;
    const uint64_t wave_total = ((uint64_t)gridDim.x * THREADS) >> 6;
    if (W == 0 && R.wave_walk) {
        // long reads: a lane walking its own 10 kbp read
        // touches one word per 2.5 KB row, 258 GB/s; here the WAVE walks one read, lane = window, so the loads are
        // consecutive words, and a tile's 64 reads are taken one after the other (bit k of the mask word = read k)
        auto probe = [&](uint32_t V) {
            return MODE == 4 ? anchor_probe_bloom(ak_lds, K.table, V, K, mask)
                             : MODE == 3 ? anchor_probe_fp(reinterpret_cast<const uint16_t *>(ak_lds), V, K)
                                         : anchor_probe<(MODE == 3 || MODE == 4) ? 0 : MODE>(ak_lds, V, K, mask);
        };
        for (uint64_t tile = wave_global; tile < n_tiles; tile += wave_total) {
            uint64_t bits = 0;
            for (int k = 0; k < 64; k++) {
                const uint64_t r = tile * 64 + (uint64_t)k;                     // wave-uniform
                if (r >= R.n_reads) break;
                if (!(K.with_exc || !rd_is_exc(R, r)) || found_flag[rd_header_id(R, r)]) continue;
                const uint32_t L = rd_len(R, r);
                if (L < 16) continue;
                const uint32_t *g = R.packed + rd_word_off(R, r);
                const uint32_t nw = (L + 15) >> 4, h_max = (L - 16) >> ASH;
                // a lane takes FOUR consecutive windows (halfword positions 4q .. 4q+3 = words 2q, 2q+1 and the low half of
                // 2q+2): three loads serve four probes, one ballot decides 256 windows, and the words of the next round are
                // requested before this round is probed (one window per lane and round was 20 dependent round trips per
                // 10 kbp read: 3.3 ms for 1 M reads)
                auto fetch3 = [&](uint32_t q, uint32_t &a, uint32_t &b, uint32_t &c3) {
                    const uint32_t w0 = 2u * q;
                    a = w0 < nw ? g[w0] : 0u; b = w0 + 1u < nw ? g[w0 + 1u] : 0u; c3 = w0 + 2u < nw ? g[w0 + 2u] : 0u;
                };
                uint32_t na, nb, nc;
                fetch3((uint32_t)lane, na, nb, nc);
                // (windows every 4 bases: the same three words serve EIGHT probes per lane)
                constexpr uint32_t PL = 2u * PW;                                // windows per lane and round
                for (uint32_t h0 = 0; h0 <= h_max; h0 += 64u * PL) {
                    const uint32_t q = (h0 / PL) + (uint32_t)lane;
                    const uint32_t a = na, b = nb, c3 = nc;
                    if (h0 + 64u * PL <= h_max) fetch3(q + 64u, na, nb, nc);
                    const uint32_t h = PL * q;
                    bool f = false;
#pragma unroll
                    for (uint32_t i = 0; i < PL; i++) {
                        const uint32_t V = i < PW ? __builtin_amdgcn_alignbit(b, a, (i * WB) & 31u) : __builtin_amdgcn_alignbit(c3, b, ((i - PW) * WB) & 31u);
                        if (h + i <= h_max) f = f | probe(V);
                    }
                    if (__ballot(f)) { bits |= 1ull << k; break; }              // one window is enough to flag the read
                }
            }
            if (lane == 0) hitmask[tile] = bits;
        }
        return;
    }
    // uniform stride: the words of the wave's NEXT tile are requested before the current one is hashed and probed, so a
    // wave never sits idle for the ~1-2 us of its own loads (4 waves per SIMD — the table takes 128 KB of LDS — were
    // not enough to cover them: the kernel ran at 62 % of its VALU issue time)
    uint32_t pre[W > 0 ? W : 1];
    auto prefetch = [&](uint64_t tile) {
        if (W > 0) {
            const uint64_t rr = tile * 64 + lane;
            if (tile < n_tiles && rr < R.n_reads) {
                const uint32_t *gp = R.packed + rr * (uint64_t)W;
#pragma unroll
                for (int i = 0; i < (W > 0 ? W : 1); i++) pre[i] = gp[i];
            }
        }
    };
    prefetch(wave_global);
    for (uint64_t tile = wave_global; tile < n_tiles; tile += wave_total) {
        const uint64_t r = tile * 64 + lane;
        bool flag = false;
        uint32_t cur[W > 0 ? W : 1];
        if (W > 0) {
#pragma unroll
            for (int i = 0; i < (W > 0 ? W : 1); i++) cur[i] = pre[i];
            prefetch(tile + wave_total);
        }
        // with_exc: every pattern is pure ACGT, so an occurrence in an exception read lies in a stretch whose packed
        // codes are the real bases — the probe stays a superset filter; the verification checks the bytes
        if (r < R.n_reads && (K.with_exc || !rd_is_exc(R, r)) && !found_flag[rd_header_id(R, r)]) {
            const uint32_t L = rd_len(R, r);
            const uint32_t *g = R.packed + rd_word_off(R, r);
            if (L >= 16) {
                const uint32_t h_max = (L - 16) >> ASH;          // last window position (halfword, or byte) whose 16-mer is inside the read
                if (W > 0) {
                    uint32_t w[W + 1];
#pragma unroll
                    for (int i = 0; i < W; i++) w[i] = cur[i];
                    w[W] = 0;
                    if (MODE == 4) {
                        // Bloom filter in LDS, exact keys in global memory.  ~7 % of the windows pass the Bloom filter, i.e.
                        // in nearly every one of the 2W-1 unrolled windows SOME lane of the wave does, and a conditional
                        // pair of global loads per window made the wave wait for 19 round trips.  So: all Bloom tests
                        // first (LDS only, a bit per window), then every lane resolves ITS positives one per round —
                        // the wave needs as many rounds as its busiest lane has positives (4-5).
                        typedef typename std::conditional<ASH == 3, uint32_t, uint64_t>::type pm_t;      // (up to 61 windows every 4 bases)
                        pm_t pm = 0;
#pragma unroll
                        for (int h = 0; h < (int)PW * (W - 1) + 1; h++) {
                            const uint32_t V = __builtin_amdgcn_alignbit(w[h / (int)PW + 1], w[h / (int)PW], ((uint32_t)h % PW) * WB);
                            // (blocked Bloom: ONE hash, one LDS word, both bits from it; a shift by a register takes the register's low
                            // five bits, so the two positions cost a shift each and the window's flag joins pm with one v_lshl_or)
                            const uint32_t h1 = ak_hash(V, K.m1);
                            const uint32_t wd = ak_lds[ak_bloom_word(h1)];
                            const uint32_t bit = (wd >> ((h1 >> 12) & 31u)) & (wd >> ((h1 >> 7) & 31u)) & 1u;
                            if ((uint32_t)h <= h_max) pm |= (pm_t)bit << h;
                        }
                        while (pm) {                                   // (divergent: lanes with fewer positives idle)
                            const uint32_t h = (uint32_t)(ASH == 3 ? __ffs((int)(uint32_t)pm) : __ffsll((unsigned long long)pm)) - 1u;
                            pm &= pm - 1u;
                            const uint32_t kk = h / PW;
                            uint32_t lo = 0, hi = 0;
#pragma unroll
                            for (int i = 0; i < W; i++) { lo = kk == (uint32_t)i ? w[i] : lo; hi = kk == (uint32_t)i ? w[i + 1] : hi; }
                            const uint32_t V = __builtin_amdgcn_alignbit(hi, lo, (h % PW) * WB);
                            const uint32_t h1 = ak_hash(V, K.m1), h2 = ak_hash(V, K.m2);
                            if ((K.table[h1 >> mask] == V) | (K.table[h2 >> mask] == V)) { flag = true; pm = 0; }
                        }
                    } else {
                    // (uniform read length: the last window is a scalar, and "window inside the read" costs no vector compare)
                    auto scan = [&](const uint32_t hm) {
#pragma unroll
                        for (int h = 0; h < (int)PW * (W - 1) + 1; h++) {
                            uint32_t V = __builtin_amdgcn_alignbit(w[h / (int)PW + 1], w[h / (int)PW], ((uint32_t)h % PW) * WB);
                            bool hit = MODE == 3 ? anchor_probe_fp(reinterpret_cast<const uint16_t *>(ak_lds), V, K)
                                                 : anchor_probe<(MODE == 3 || MODE == 4) ? 0 : MODE>(ak_lds, V, K, mask);
                            flag = flag | (hit & ((uint32_t)h <= hm));
                            // (16 LDS reads in flight are plenty; left alone the scheduler hoists all 4W-2 of them and, from
                            // W = 12, spills)
                            if ((h & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                        }
                    };
                    if (R.uniform_len) scan((R.uniform_len - 16u) >> ASH);
                    else scan(h_max);
                    }
                } else {
                    // (four words per round, requested together: one word per round was one dependent round trip per 16 bases — reads of
                    // 300 .. 800 bases, lane per read, took twice the time of the register form per base)
                    const uint32_t nw = (L + 15) >> 4;
                    auto probe = [&](uint32_t V) {
                        return MODE == 4 ? anchor_probe_bloom(ak_lds, K.table, V, K, mask)
                                         : MODE == 3 ? anchor_probe_fp(reinterpret_cast<const uint16_t *>(ak_lds), V, K)
                                                     : anchor_probe<(MODE == 3 || MODE == 4) ? 0 : MODE>(ak_lds, V, K, mask);
                    };
                    uint32_t lo = g[0];
                    for (uint32_t h = 0; h <= h_max && !flag; h += 4u * PW) {
                        const uint32_t wi = (h / PW) + 1;
                        uint32_t x[4];
#pragma unroll
                        for (uint32_t q = 0; q < 4; q++) x[q] = wi + q < nw ? g[wi + q] : 0u;
#pragma unroll
                        for (uint32_t q = 0; q < 4; q++) {
#pragma unroll
                            for (uint32_t i = 0; i < PW; i++)
                                if (h + PW * q + i <= h_max && probe(__builtin_amdgcn_alignbit(x[q], lo, i * WB))) flag = true;
                            lo = x[q];
                        }
                    }
                }
            }
        }
        uint64_t m = __ballot(flag);
        if (lane == 0) hitmask[tile] = m;
    }
}

template <int W, int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void k_anchor_filter(DevReads R, DevAnchors K, const uint8_t *found_flag, uint64_t *hitmask)
{
    // (1 024 threads per block and, with its table in LDS, one block per CU.  Until round 4 a CRASS_VGPR_FLOOR(120) kept every
    // instantiation off a multiple of 8 registers: 4 waves x 128 allocated registers = a SIMD's whole file, so no wave of any
    // other kernel could share the CU — the view export beside it then cost the probe 144 -> 216 us.  The build's guard is exact
    // now, crass_amd/vgpr_guard.py, and these kernels hold no 64-bit shift by their last register.)
    extern __shared__ __attribute__((aligned(16))) uint32_t ak_lds_buf[];
    const uint32_t tsize = 1u << K.log_size;
    const uint32_t *ak_lds = K.table;                   // key sets too large for LDS are probed in global memory (L2)
    if (MODE != 2) {
        for (uint32_t i = threadIdx.x; i < tsize; i += THREADS) ak_lds_buf[i] = K.table[i];
        __syncthreads();
        ak_lds = ak_lds_buf;
    }
    anchor_filter_body<W, THREADS, MODE>(R, K, ak_lds, found_flag, hitmask);
}

// the same filter when the key table was built on the device (dmerge.hip): its size is only known there
// (ASH is a template parameter of the KERNEL: with both forms in one kernel the default one was allocated the other's registers —
// 99 instead of 56 — and no other kernel's waves fitted beside its four per SIMD any more)
template <int W, int THREADS, int ASH>
__global__ __launch_bounds__(THREADS) void k_anchor_filter_dev(DevReads R, DevMerge M, const uint8_t *found_flag, uint64_t *hitmask)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t ak_lds_buf[];
    if (M.flag_post && blockIdx.x == 0 && threadIdx.x == 0) stage_flag_store(M.flag_post, M.flag_post_val);      // (the merge's kernels are complete)
    DevAnchors K;
    K.table = M.anchor_tab; K.log_size = M.st->log_size; K.mode = 0; K.m1 = M.m1; K.m2 = M.m2; K.n_keys = 0;
    K.with_exc = 1;
    if (M.st->fail != 0 || K.log_size == 0) {            // the host redoes the merge; flag nothing
        const uint64_t n_tiles = (R.n_reads + 63) / 64;
        for (uint64_t t = blockIdx.x * (uint64_t)THREADS + threadIdx.x; t < n_tiles; t += (uint64_t)gridDim.x * THREADS) hitmask[t] = 0ull;
        return;
    }
    if (K.log_size <= 15) {
        const uint32_t tsize = 1u << K.log_size;
        for (uint32_t i = threadIdx.x; i < tsize; i += THREADS) ak_lds_buf[i] = K.table[i];
        __syncthreads();
        anchor_filter_body<W, THREADS, 0, ASH>(R, K, ak_lds_buf, found_flag, hitmask);
    } else if (M.st->tab_mode == 3) {
        for (uint32_t i = threadIdx.x; i < (1u << 15); i += THREADS) ak_lds_buf[i] = M.anchor_fp[i];
        __syncthreads();
        anchor_filter_body<W, THREADS, 3, ASH>(R, K, ak_lds_buf, found_flag, hitmask);
    } else {
        for (uint32_t i = threadIdx.x; i < (1u << 15); i += THREADS) ak_lds_buf[i] = M.anchor_fp[i];
        __syncthreads();
        anchor_filter_body<W, THREADS, 4, ASH>(R, K, ak_lds_buf, found_flag, hitmask);
    }
}

hipError_t launch_anchor_filter_dev(const DevReads &R, const DevMerge &M, const uint8_t *found_flag, uint64_t *hitmask, hipStream_t st)
{
    if (R.n_reads == 0) return hipSuccess;
    const size_t lds = 128 * 1024;
    const uint64_t n_tiles = (R.n_reads + 63) / 64;
    constexpr int T = 1024;
    uint64_t blocks = (n_tiles + (T / 64) - 1) / (T / 64);
    if (blocks > 256) blocks = 256;
    hipError_t e;
#define AKD_LAUNCH1(WW, AA)                                                                                             \
    {                                                                                                                   \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_anchor_filter_dev<WW, T, AA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                  \
        CRASS_LAUNCH((k_anchor_filter_dev<WW, T, AA>), dim3((unsigned)blocks), dim3(T), lds, st, R, M, found_flag, hitmask); \
    }
#define AKD_LAUNCH(WW) { if (M.akey_shift == 2u) AKD_LAUNCH1(WW, 2) else AKD_LAUNCH1(WW, 3) }
    switch (R.stride_words) {
        case 4: AKD_LAUNCH(4) break;  case 5: AKD_LAUNCH(5) break;  case 6: AKD_LAUNCH(6) break;  case 7: AKD_LAUNCH(7) break;
        case 8: AKD_LAUNCH(8) break;  case 9: AKD_LAUNCH(9) break;  case 10: AKD_LAUNCH(10) break; case 11: AKD_LAUNCH(11) break;
        case 12: AKD_LAUNCH(12) break; case 13: AKD_LAUNCH(13) break; case 14: AKD_LAUNCH(14) break; case 15: AKD_LAUNCH(15) break;
        case 16: AKD_LAUNCH(16) break;
        default: AKD_LAUNCH(0) break;
    }
#undef AKD_LAUNCH
#undef AKD_LAUNCH1
    return hipGetLastError();
}

hipError_t launch_anchor_filter(const DevReads &R, const DevAnchors &K, const uint8_t *found_flag, uint64_t *hitmask, hipStream_t st)
{
    if (R.n_reads == 0) return hipSuccess;
    const size_t tbytes = (size_t)4 << K.log_size;
    const bool in_lds = tbytes <= 128 * 1024;
    if (K.mode == 1 && !in_lds) return hipErrorInvalidValue;
    const size_t lds = in_lds ? tbytes : 0;
    const uint64_t n_tiles = (R.n_reads + 63) / 64;
    constexpr int T = 1024;
    uint64_t blocks = (n_tiles + (T / 64) - 1) / (T / 64);
    const uint64_t cap = lds > 80 * 1024 ? 256 : (lds > 40 * 1024 ? 512 : 1024);
    if (blocks > cap) blocks = cap;
    hipError_t e;
#define AK_LAUNCH_M(WW, MM)                                                                                             \
    {                                                                                                                   \
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_anchor_filter<WW, T, MM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                  \
        CRASS_LAUNCH((k_anchor_filter<WW, T, MM>), dim3((unsigned)blocks), dim3(T), lds, st, R, K, found_flag, hitmask); \
    }
#define AK_LAUNCH(WW)                                                                                                   \
    if (!in_lds) AK_LAUNCH_M(WW, 2) else if (K.mode == 1) AK_LAUNCH_M(WW, 1) else AK_LAUNCH_M(WW, 0)
    switch (R.stride_words) {
        case 4: AK_LAUNCH(4) break;  case 5: AK_LAUNCH(5) break;  case 6: AK_LAUNCH(6) break;  case 7: AK_LAUNCH(7) break;
        case 8: AK_LAUNCH(8) break;  case 9: AK_LAUNCH(9) break;  case 10: AK_LAUNCH(10) break; case 11: AK_LAUNCH(11) break;
        case 12: AK_LAUNCH(12) break; case 13: AK_LAUNCH(13) break; case 14: AK_LAUNCH(14) break; case 15: AK_LAUNCH(15) break;
        case 16: AK_LAUNCH(16) break;
        default: AK_LAUNCH(0) break;
    }
#undef AK_LAUNCH
#undef AK_LAUNCH_M
    return hipGetLastError();
}

// exact first-match scan of the flagged reads (lane per flagged read), transition table in global
// memory (L2-resident).  info_by_slot[k] = (end_exclusive << 8) | length, 0 = no pattern occurs.
__global__ __launch_bounds__(256) void k_recruit_list(DevReads R, DevAutomaton A, const uint64_t *idx, const uint32_t *d_n,
                                                       uint64_t n_max, uint32_t *info_by_slot, uint32_t *pid_by_slot)
{
    uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t n = *d_n;
    if (n > n_max) n = n_max;
    if (k >= n) return;
    const uint64_t r = idx[k];
    const uint32_t L = rd_len(R, r);
    const uint32_t *g = R.packed + rd_word_off(R, r);
    const uint32_t symA = A.sym['A'], symC = A.sym['C'], symG = A.sym['G'], symT = A.sym['T'];
    uint32_t state = 0, word = 0, info = 0, pid = 0;
    for (uint32_t i = 0; i < L; i++) {
        if ((i & 15u) == 0) word = g[i >> 4];
        uint32_t c = word & 3u;
        word >>= 2;
        if (A.go4) state = A.go4[state * 4 + c];
        else if (A.go4w) state = A.go4w[(size_t)state * 4 + c];
        else {
            uint32_t sy = c == 0 ? symA : c == 1 ? symC : c == 2 ? symG : symT;
            state = A.go16 ? (uint32_t)A.go16[(size_t)state * A.n_sym1 + sy] : A.go32[(size_t)state * A.n_sym1 + sy];
        }
        uint32_t ol = A.out_len[state];
        if (ol) { info = ((i + 1) << 8) | ol; pid = A.out_pid[state]; break; }
    }
    info_by_slot[k] = info;
    pid_by_slot[k] = pid;
}

// The same for long reads: one WAVE per flagged read.  The automaton's state at a position only depends on the last
// max_pat_len bases (the depth of the trie), so lane l scans its own slice [l * seg, (l + 1) * seg) after a warm-up of
// max_pat_len bases from the start state and is in the exact state for every position it reports; the first callback
// of the whole read is the smallest reported position over the lanes (a lane per 10 kbp read walked 10 000 dependent
// table look-ups: 2.8 ms for a few hundred reads).
__global__ __launch_bounds__(256) void k_recruit_list_wave(DevReads R, DevAutomaton A, const uint64_t *idx, const uint32_t *d_n,
                                                            uint64_t n_max, uint32_t *info_by_slot, uint32_t *pid_by_slot)
{
    const int lane = threadIdx.x & 63;
    const uint64_t k = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    uint64_t n = *d_n;
    if (n > n_max) n = n_max;
    if (k >= n) return;
    const uint64_t r = idx[k];
    const uint32_t L = rd_len(R, r);
    const uint32_t *g = R.packed + rd_word_off(R, r);
    const uint32_t symA = A.sym['A'], symC = A.sym['C'], symG = A.sym['G'], symT = A.sym['T'];
    const uint32_t seg = (L + 63u) / 64u;
    const uint32_t s0 = (uint32_t)lane * seg, e0 = min(L, s0 + seg);
    uint32_t first = 0xFFFFFFFFu, ol_found = 0, pid = 0;
    if (s0 < L) {
        const uint32_t p0 = s0 >= A.max_pat_len ? s0 - A.max_pat_len : 0u;        // warm-up (exact from the read start anyway)
        uint32_t state = 0, word = 0;
        for (uint32_t i = p0; i < e0; i++) {
            if ((i & 15u) == 0 || i == p0) word = g[i >> 4] >> ((i & 15u) * 2u);
            const uint32_t c = word & 3u;
            word >>= 2;
            if (A.go4) state = A.go4[state * 4 + c];
            else if (A.go4w) state = A.go4w[(size_t)state * 4 + c];
            else {
                const uint32_t sy = c == 0 ? symA : c == 1 ? symC : c == 2 ? symG : symT;
                state = A.go16 ? (uint32_t)A.go16[(size_t)state * A.n_sym1 + sy] : A.go32[(size_t)state * A.n_sym1 + sy];
            }
            if (i >= s0) {
                const uint32_t ol = A.out_len[state];
                if (ol) { first = i; ol_found = ol; pid = A.out_pid[state]; break; }
            }
        }
    }
    uint32_t best = first;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, off));
    if (best == 0xFFFFFFFFu) { if (lane == 0) { info_by_slot[k] = 0; pid_by_slot[k] = 0; } return; }
    if (first == best) {                                  // exactly one lane owns that position
        info_by_slot[k] = ((best + 1) << 8) | ol_found;
        pid_by_slot[k] = pid;
    }
}

hipError_t launch_recruit_list(const DevReads &R, const DevAutomaton &A, const uint64_t *idx, const uint32_t *d_n,
                               uint64_t n_max, uint32_t *info_by_slot, uint32_t *pid_by_slot, hipStream_t st)
{
    if (n_max == 0) return hipSuccess;
    if (R.wave_walk && A.max_pat_len)                      // long reads
        CRASS_LAUNCH(k_recruit_list_wave, dim3((unsigned)((n_max + 3) / 4)), dim3(256), 0, st, R, A, idx, d_n, n_max, info_by_slot, pid_by_slot);
    else
        CRASS_LAUNCH(k_recruit_list, dim3((unsigned)((n_max + 255) / 256)), dim3(256), 0, st, R, A, idx, d_n, n_max, info_by_slot, pid_by_slot);
    return hipGetLastError();
}

// exception reads: raw bytes through the byte-symbol automaton, lane per exception read
__global__ __launch_bounds__(256) void k_recruit_exc(DevReads R, DevAutomaton A, const uint8_t *found_flag, uint32_t *exc_hit_info)
{
    uint64_t s = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (s >= R.n_exc) return;
    uint64_t r = R.exc_read[s];
    uint32_t info = 0;
    if (!found_flag[rd_header_id(R, r)]) {
        uint64_t o0 = R.exc_off[s];
        uint32_t L = (uint32_t)(R.exc_off[s + 1] - o0);
        uint32_t state = 0;
        for (uint32_t i = 0; i < L; i++) {
            uint32_t sy = A.sym[R.exc_bytes[o0 + i]];
            state = A.go16 ? (uint32_t)A.go16[(size_t)state * A.n_sym1 + sy] : A.go32[(size_t)state * A.n_sym1 + sy];
            uint32_t ol = A.out_len[state];
            if (ol) { info = ((i + 1) << 8) | ol; break; }
        }
    }
    exc_hit_info[s] = info;
}

hipError_t launch_recruit_exceptions(const DevReads &R, const DevAutomaton &A, const uint8_t *found_flag,
                                     uint32_t *exc_hit_info, hipStream_t st)
{
    if (R.n_exc == 0) return hipSuccess;
    CRASS_LAUNCH(k_recruit_exc, dim3((unsigned)((R.n_exc + 255) / 256)), dim3(256), 0, st, R, A, found_flag, exc_hit_info);
    return hipGetLastError();
}

// on_match + addReadHolder's DRLowLexi for the single recruited repeat
// (libcrispr.cpp:408-442, ReadHolder.cpp:524-528,573-590).  Thread per hit.
template <bool EXC>
__global__ __launch_bounds__(256) void k_recruit_finish(DevReads R, const uint64_t *hit_idx, const uint32_t *d_n_hits,
                                                        uint64_t n_max, const uint32_t *hit_info, int info_by_slot,
                                                        const uint32_t *pid_by_slot, const uint32_t *pat_token,
                                                        RecruitOut *out, char *dr_chars, uint32_t dr_stride, const uint64_t *pat_mask)
{
    if constexpr (!EXC) CRASS_VGPR_FLOOR(24);      // 24 VGPRs with the 128-bit shift amount in v23: the kernel that exposed the erratum
    uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t n = EXC ? R.n_exc : (uint64_t)(*d_n_hits);
    if (n > n_max) n = n_max;
    if (k >= n) return;
    RecruitOut o; o.start = 0; o.end = 0; o.token = 0; o.dr_len = 0; o.low_lexi = 0; o.pad = 0;
    uint32_t info;
    uint32_t L;
    uint64_t r = 0, o0 = 0;
    const uint32_t *g = nullptr;
    if (EXC) {
        info = hit_info[k];
        o0 = R.exc_off[k];
        L = (uint32_t)(R.exc_off[k + 1] - o0);
    } else {
        r = hit_idx[k];
        info = info_by_slot ? hit_info[k] : hit_info[r];
        L = rd_len(R, r);
        g = R.packed + rd_word_off(R, r);
    }
    if (info == 0) { out[k] = o; return; }              // no match (exception read / anchor false positive)
    uint32_t textpos = info >> 8, len = info & 0xFFu;
    uint32_t DR_end = textpos - 1;
    if (DR_end >= L) DR_end = L - 1;
    uint32_t start = DR_end - (len - 1);
    // a pattern with an 'N' (device merge, dmerge.hip) matched an exception read: the packed words hold 'A' there,
    // so the repeat is read from the read's bytes
    const uint8_t *raw = nullptr;
    if (!EXC && pat_mask && pid_by_slot && pat_mask[pid_by_slot[k]] != 0ull && R.n_exc) {
        uint64_t lo = 0, hi = R.n_exc - 1;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (R.exc_read[mid] < r) lo = mid + 1; else hi = mid; }
        raw = R.exc_bytes + R.exc_off[lo];
    }
    if (!EXC && len <= 64 && !raw) {
        // packed reads: the repeat as a 128-bit value (base i in bits 2i..2i+1), its reverse complement by bit
        // reversal, and DRLowLexi's string comparison as "first differing base from the low end"
        const uint32_t nw = (L + 15) >> 4, w0 = start >> 4, sh = (start & 15u) * 2u;
        uint32_t x[5];
#pragma unroll
        for (int q = 0; q < 5; q++) x[q] = (w0 + q < nw) ? g[w0 + q] : 0u;
        uint32_t y[4];
#pragma unroll
        for (int q = 0; q < 4; q++) y[q] = sh ? ((x[q] >> sh) | (x[q + 1] << (32 - sh))) : x[q];
        uint64_t v0 = (uint64_t)y[0] | ((uint64_t)y[1] << 32), v1 = (uint64_t)y[2] | ((uint64_t)y[3] << 32);
        const uint64_t m0 = len >= 32 ? ~0ull : ((1ull << (2 * len)) - 1ull);
        const uint64_t m1 = len >= 64 ? ~0ull : (len > 32 ? ((1ull << (2 * (len - 32))) - 1ull) : 0ull);
        v0 &= m0; v1 &= m1;
        auto rev2 = [](uint64_t t) -> uint64_t {          // reverse the order of the 32 two-bit groups
            t = __brevll(t);
            return ((t >> 1) & 0x5555555555555555ull) | ((t & 0x5555555555555555ull) << 1);
        };
        // complement, reverse all 64 groups of the 128-bit value, then shift the len groups down to bit 0
        const uint64_t c0 = rev2(~v1), c1 = rev2(~v0);     // (c1:c0) = reversed 128 bits
        const uint32_t drop = 128u - 2u * len;             // unused high groups became low groups
        uint64_t r0, r1;
        if (drop == 0) { r0 = c0; r1 = c1; }
        else if (drop < 64) { r0 = (c0 >> drop) | (c1 << (64 - drop)); r1 = c1 >> drop; }
        else { r0 = c1 >> (drop - 64); r1 = 0; }
        r0 &= m0; r1 &= m1;
        int less = 0;
        const uint64_t d0 = v0 ^ r0, d1 = v1 ^ r1;
        if (d0 | d1) {
            const uint64_t dv = d0 ? d0 : d1, av = d0 ? v0 : v1, bv = d0 ? r0 : r1;
            const int p = (__ffsll((unsigned long long)dv) - 1) & ~1;
            less = ((av >> p) & 3ull) < ((bv >> p) & 3ull);
        }
        const uint64_t s0 = less ? v0 : r0, s1 = less ? v1 : r1;
        if (dr_chars) {
            char *dr = dr_chars + k * (uint64_t)dr_stride;
            for (uint32_t i = 0; i < dr_stride; i++) {
                const uint32_t c = (uint32_t)(((i < 32 ? s0 : s1) >> (2 * (i & 31))) & 3ull);
                dr[i] = i < len ? "ACGT"[c] : (char)0;
            }
        }
        if (less) { o.start = start; o.end = DR_end; o.low_lexi = 1; }
        else { o.start = L - 1 - DR_end; o.end = L - 1 - start; o.low_lexi = 0; }
        o.dr_len = (uint16_t)len;
        if (pid_by_slot && pat_token) o.token = pat_token[pid_by_slot[k]];
        out[k] = o;
        return;
    }
    auto base_at = [&](uint32_t i) -> uint8_t {
        if (EXC) return R.exc_bytes[o0 + i];
        if (raw) return raw[i];
        uint32_t c = (g[i >> 4] >> ((i & 15u) * 2u)) & 3u;
        return (uint8_t)("ACGT"[c]);
    };
    int less = 0;
    for (uint32_t i = 0; i < len; i++) {
        uint8_t a = base_at(start + i);
        uint8_t b = c_comp[base_at(start + len - 1 - i) & 127];
        if (a != b) { less = a < b; break; }
    }
    char *dr = dr_chars ? dr_chars + k * (uint64_t)dr_stride : nullptr;
    if (less) {
        if (dr) for (uint32_t i = 0; i < len; i++) dr[i] = (char)base_at(start + i);
        o.start = start; o.end = DR_end; o.low_lexi = 1;
    } else {
        if (dr) for (uint32_t i = 0; i < len; i++) dr[i] = (char)c_comp[base_at(start + len - 1 - i) & 127];
        o.start = L - 1 - DR_end; o.end = L - 1 - start; o.low_lexi = 0;
    }
    if (dr) for (uint32_t i = len; i < dr_stride; i++) dr[i] = 0;
    o.dr_len = (uint16_t)len;
    // the matched pattern's low-lexi form is a stored DR variant: its token was resolved once per
    // pattern on the host (addReadHolder's lookup, libcrispr.cpp:1137)
    if (pid_by_slot && pat_token) o.token = pat_token[pid_by_slot[k]];
    out[k] = o;
}

hipError_t launch_recruit_finish(const DevReads &R, const uint64_t *hit_idx, const uint32_t *d_n_hits, uint64_t n_hits_max,
                                 const uint32_t *hit_info, bool info_by_slot, bool exceptions,
                                 const uint32_t *pid_by_slot, const uint32_t *pat_token, RecruitOut *out,
                                 char *dr_chars, uint32_t dr_stride, hipStream_t st, const uint64_t *pat_mask)
{
    if (n_hits_max == 0) return hipSuccess;
    unsigned nb = (unsigned)((n_hits_max + 255) / 256);
    if (exceptions)
        CRASS_LAUNCH(k_recruit_finish<true>, dim3(nb), dim3(256), 0, st, R, hit_idx, d_n_hits, n_hits_max, hit_info, 1, pid_by_slot, pat_token, out, dr_chars, dr_stride, (const uint64_t *)nullptr);
    else
        CRASS_LAUNCH(k_recruit_finish<false>, dim3(nb), dim3(256), 0, st, R, hit_idx, d_n_hits, n_hits_max, hit_info, info_by_slot ? 1 : 0, pid_by_slot, pat_token, out, dr_chars, dr_stride, pat_mask);
    return hipGetLastError();
}

// ---- pass-2 sink on the device: drop the slots without a match, pack the rest (read order) ----
__global__ __launch_bounds__(256) void k_recruit_valid_mask(const RecruitOut *rec, const uint32_t *d_n, uint64_t n_max, uint64_t *mask)
{
    const uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t n = *d_n;
    if (n > n_max) n = n_max;
    const bool v = k < n && rec[k].dr_len != 0;
    const uint64_t m = __ballot(v);
    if ((threadIdx.x & 63) == 0 && k < n_max) mask[k >> 6] = m;
}
__global__ __launch_bounds__(256) void k_pack_p2_blob(const RecruitOut *rec, const uint64_t *hit_idx, uint64_t read_base, const uint64_t *vidx,
                                                       const uint32_t *d_nv, uint64_t cap, uint8_t *blob, const uint32_t *d_n_hits, uint32_t *h_n_hits,
                                                       uint32_t narrow)
{
    const uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t nv = *d_nv;
    if (nv > cap) nv = cap;
    if (q == 0) {
        reinterpret_cast<uint64_t *>(blob)[0] = nv; reinterpret_cast<uint64_t *>(blob)[1] = cap;
        if (h_n_hits) *h_n_hits = *d_n_hits;            // the flagged-read count the host checks its bound against
    }
    if (q >= nv) return;
    const P2Blob b = p2_blob_layout(cap, narrow);
    const uint64_t k = vidx[q];
    const RecruitOut o = rec[k];
    if (narrow == 2) {
        reinterpret_cast<uint32_t *>(blob + b.token)[q] = o.token | ((uint32_t)(o.low_lexi != 0) << 31);
        reinterpret_cast<uint32_t *>(blob + b.read)[q] = (uint32_t)hit_idx[k];
        (blob + b.start)[q] = (uint8_t)o.start;
        return;
    }
    reinterpret_cast<uint32_t *>(blob + b.token)[q] = o.token;
    if (narrow) {
        reinterpret_cast<uint32_t *>(blob + b.read)[q] = (uint32_t)hit_idx[k];          // (local index: the host adds the base)
        (blob + b.start)[q] = (uint8_t)o.start;
        (blob + b.end)[q] = (uint8_t)o.end;
    } else {
        reinterpret_cast<uint64_t *>(blob + b.read)[q] = read_base + hit_idx[k];
        reinterpret_cast<uint16_t *>(blob + b.start)[q] = (uint16_t)o.start;
        reinterpret_cast<uint16_t *>(blob + b.end)[q] = (uint16_t)o.end;
    }
    (blob + b.dr_len)[q] = (uint8_t)o.dr_len;
    (blob + b.low)[q] = o.low_lexi;
}
// k_recruit_valid_mask + compaction in one pass (decoupled look-back): vidx[rank] = slot of the rank-th valid hit
__global__ __launch_bounds__(1024) void k_valid_compact(const RecruitOut *rec, const uint32_t *d_n_hits, uint64_t cap, uint64_t *vidx, uint32_t *d_nv,
                                                         Lookback lb)
{
    uint64_t n = *d_n_hits;
    if (n > cap) n = cap;
    const uint32_t n_tiles = n ? (uint32_t)((n + kLbElemsPerTile - 1) / kLbElemsPerTile) : 1u;      // (sized for a bound: the tiles past the count leave at once)
    if (blockIdx.x >= n_tiles) return;
    const uint32_t tile = lb_tile_id(lb, n_tiles);
    const uint64_t k0 = (uint64_t)tile * kLbElemsPerTile + 4u * threadIdx.x;
    uint32_t fm = 0;
#pragma unroll
    for (int e = 0; e < 4; e++) if (k0 + e < n && rec[k0 + e].dr_len != 0) fm |= 1u << e;
    uint32_t upto;
    uint64_t q = lb_rank4(lb, tile, (uint32_t)__popc(fm), &upto);
    if (tile == n_tiles - 1 && threadIdx.x == 0) *d_nv = upto;
#pragma unroll
    for (int e = 0; e < 4; e++) if (fm & (1u << e)) vidx[q++] = k0 + e;
}

hipError_t launch_pack_p2_blob(const RecruitOut *rec, const uint64_t *hit_idx, uint64_t read_base,
                               const uint32_t *d_n_hits, uint64_t n_hits_max, uint64_t *mask, uint32_t *word_prefix, uint32_t *block_sums,
                               uint64_t *vidx, uint32_t *d_nv, uint8_t *blob, hipStream_t st, uint32_t *h_n_hits, const Lookback *lb, uint32_t narrow)
{
    if (n_hits_max == 0) return hipSuccess;
    const unsigned nb = (unsigned)((n_hits_max + 255) / 256);
    if (lb) {           // (the caller reserved nb tiles)
        const unsigned nt = (unsigned)((n_hits_max + kLbElemsPerTile - 1) / kLbElemsPerTile);
        CRASS_LAUNCH(k_valid_compact, dim3(nt), dim3(1024), 0, st, rec, d_n_hits, n_hits_max, vidx, d_nv, *lb);
        CRASS_LAUNCH(k_pack_p2_blob, dim3(nb), dim3(256), 0, st, rec, hit_idx, read_base, vidx, d_nv, n_hits_max, blob, d_n_hits, h_n_hits, narrow);
        return hipGetLastError();
    }
    CRASS_LAUNCH(k_recruit_valid_mask, dim3(nb), dim3(256), 0, st, rec, d_n_hits, n_hits_max, mask);
    hipError_t e = launch_compact(mask, (n_hits_max + 63) / 64, n_hits_max, word_prefix, block_sums, vidx, n_hits_max, d_nv, st);
    if (e != hipSuccess) return e;
    CRASS_LAUNCH(k_pack_p2_blob, dim3(nb), dim3(256), 0, st, rec, hit_idx, read_base, vidx, d_nv, n_hits_max, blob, d_n_hits, h_n_hits, narrow);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Levenshtein / similarity batch: one wave per string pair
// ------------------------------------------------------------------------------------
// lane per pair, bit-parallel; pairs it cannot handle are marked dist = -1 for the wave kernel
__global__ __launch_bounds__(256) void k_lev_batch_lanes(const uint8_t *chars, const uint64_t *a_off, const uint32_t *a_len,
                                                          const uint64_t *b_off, const uint32_t *b_len, uint64_t n_pairs,
                                                          int32_t *dist, float *sim)
{
    uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (k >= n_pairs) return;
    const uint8_t *s = chars + a_off[k], *t = chars + b_off[k];
    const int n = (int)a_len[k], m = (int)b_len[k];
    int d = lane_lev_bp(s, n, t, m);
    dist[k] = d;
    if (sim && d >= 0) {
        float max_length = (float)(n > m ? n : m);
        sim[k] = (n < 3 || m < 3) ? 0.0f : (float)(1.0 - (double)((float)d / max_length));
    }
}

__global__ __launch_bounds__(WAVE) void k_lev_batch(const uint8_t *chars, const uint64_t *a_off, const uint32_t *a_len,
                                                     const uint64_t *b_off, const uint32_t *b_len, uint64_t n_pairs,
                                                     int32_t *dist, float *sim, uint32_t row_elems, uint32_t str_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lv_lds[];
    uint8_t *sa = lv_lds;
    uint8_t *sb = lv_lds + str_bytes;
    uint16_t *rowA = reinterpret_cast<uint16_t *>(lv_lds + 2 * (size_t)str_bytes);
    uint16_t *rowB = rowA + row_elems;
    const int lane = threadIdx.x;
    for (uint64_t k = blockIdx.x; k < n_pairs; k += gridDim.x) {
        int n = (int)a_len[k], m = (int)b_len[k];
        if (dist[k] != -1) continue;                  // already done by the lane kernel
        wave_sync();
        for (int i = lane; i < n; i += WAVE) sa[i] = chars[a_off[k] + i];
        for (int i = lane; i < m; i += WAVE) sb[i] = chars[b_off[k] + i];
        wave_sync();
        int d = wave_lev(sa, n, sb, m, rowA, rowB, lane);
        float s = 0.0f;
        if (sim) s = wave_similarity(sa, n, sb, m, rowA, rowB, lane);
        if (lane == 0) { dist[k] = d; if (sim) sim[k] = s; }
    }
}

hipError_t launch_levenshtein_batch(const uint8_t *chars, const uint64_t *a_off, const uint32_t *a_len,
                                    const uint64_t *b_off, const uint32_t *b_len, uint64_t n_pairs,
                                    int32_t *dist, float *sim, uint32_t max_len, hipStream_t st)
{
    if (n_pairs == 0) return hipSuccess;
    uint32_t str_bytes = (max_len + 16 + 15u) & ~15u;
    uint32_t row_elems = ((max_len + 8) + 7u) & ~7u;
    size_t lds = 2 * (size_t)str_bytes + 2 * (size_t)row_elems * 2;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lev_batch), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    // `dist` is always a device buffer here (engine.cpp allocates it even if the caller wants only sims)
    CRASS_LAUNCH(k_lev_batch_lanes, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, st, chars, a_off, a_len, b_off, b_len, n_pairs, dist, sim);
    uint64_t grid = n_pairs < 4096 ? n_pairs : 4096;
    CRASS_LAUNCH(k_lev_batch, dim3((unsigned)grid), dim3(WAVE), lds, st, chars, a_off, a_len, b_off, b_len, n_pairs, dist, sim, row_elems, str_bytes);
    return hipGetLastError();
}

hipError_t upload_comp_table(const unsigned char *tab128)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(c_comp), tab128, 128);
}

} // namespace crass
