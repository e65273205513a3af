// dmerge.hip — the DR merge on the device (gfx950, wave64): WorkHorse::createNonRedundantSet
// (src/crass/WorkHorse.cpp:648-709) = clusterDRReads (:1404-1637) + removeRedundantRepeats
// (:612-645, includeSubstring :78-86), followed by the construction of everything pass 2 needs
// (pattern list, anchor key set, exact verification index).  Input: the DISTINCT candidate DR
// strings in first-occurrence order (= StringCheck token order, StringCheck.cpp:46-81) that the
// pass-1 tail already produced on the device; nothing here waits for the host.
//
// Preconditions (checked by the host before it takes this path, re-checked here -> fail word):
// every token is over {A,C,G,T,N}, 23 <= length <= 64.  Anything else uses the host merge (merge.cpp).
// An 'N' (reads with an undetermined base do produce a few such DR variants) is packed as 'A' plus a bit in a
// 64-bit position mask that every comparison carries along; the handful of 11-mers that contain one get their
// identity from a small open-addressing table instead of the 22-bit code (k_dm_pack_codes).
//
// All integer work on a few 10^4 short strings: no MFMA, no HBM roofline to speak of — the point of
// running it here is that the step no longer leaves the device between pass 1 and pass 2.
#include "engine_internal.h"
#include <algorithm>

namespace crass {

#define WAVE 64
static constexpr uint32_t kUnres = 0xFFFFFFFFu;      // root_of[t] not decided yet
static constexpr uint32_t kNoLane = 0xFFFFFFFEu;     // lane holds no earlier-owned k-mer
static constexpr int kClusterK = 11;                 // CRASS_DEF_KMER_SIZE (crassDefines.h:66)

// (h0,h1) >> bits, low 64 bits; bits in [0,127]
static __device__ __forceinline__ uint64_t shr128_lo(uint64_t h0, uint64_t h1, uint32_t bits)
{
    if (bits == 0) return h0;
    if (bits < 64) return (h0 >> bits) | (h1 << (64 - bits));
    return h1 >> (bits - 64);
}
static __device__ __forceinline__ uint64_t shr128_hi(uint64_t h1, uint32_t bits)
{
    return bits < 64 ? (h1 >> bits) : 0ull;
}
// mask of the low 2*len bits of a 128-bit value, len in [1,64]
static __device__ __forceinline__ void mask128(uint32_t len, uint64_t &m0, uint64_t &m1)
{
    if (len >= 32) { m0 = ~0ull; m1 = len >= 64 ? ~0ull : ((1ull << (2 * (len - 32))) - 1ull); }
    else { m0 = (1ull << (2 * len)) - 1ull; m1 = 0ull; }
}
static __device__ __forceinline__ uint32_t kset_hash(uint32_t key, uint32_t log_size)
{
    return (key * 0x9E3779B1u) >> (32u - log_size);
}
static __device__ __forceinline__ uint32_t ak_h(uint32_t v, uint32_t m, uint32_t rsh)
{
    return ak_hash(v, m) >> rsh;                              // same hash as anchor_probe (kernels.hip)
}

// number of tokens: M.n_tok, or — when the merge is launched before the host knows the count (crass_hip_seed_scan
// queues it right behind pass 1) — the device-side count, of which M.n_tok is then only the bound every buffer
// and grid was sized for
static __device__ __forceinline__ uint32_t dm_ntok(const DevMerge &M)
{
    return M.d_ntok ? min(*M.d_ntok, M.n_tok) : M.n_tok;
}

// A merge whose input is unusable (a token outside ACGTN / 23..64 bases: fail bit 1; more tokens than the launch was sized
// for: bit 64) is decided by the FIRST kernel; its results are never used (the host merges instead), so every later kernel
// leaves at once — uniformly: the bits are set before it starts.  Without this, garbage lengths (the rows of an exchange that
// overflowed are never unpacked) sent k_dm_redundant into a ~2^32-iteration window loop (`lenj - 22` wraps for lenj < 22).
// (A plain load: the bits were written by an EARLIER kernel, so every cache level shows them, and a uniform address makes it one
// scalar load per wave.  An agent-scope atomic load here — every thread of 670 k asking L2 for the same word — cost 100 us per
// kernel at 100 M reads: k_dm_redundant 54 -> 170 us, k_dm_keys 61 -> 165 us, profiles/NOTES_r03.md.)
static __device__ __forceinline__ bool dm_abandoned(const DevMerge &M)
{
    return (M.st->fail & (1u | 64u)) != 0u;
}

// ---- 0. initialise every word a later kernel polls, counts into or probes (dm_init_slice, engine_internal.h) ----
__global__ __launch_bounds__(256) void k_dm_init(DevMerge M)
{
    dm_init_slice(M, blockIdx.x * (uint64_t)blockDim.x + threadIdx.x, (uint64_t)gridDim.x * blockDim.x);
}

// ---- 1. 2-bit packing (forward and reverse complement) + laurenized 11-mer codes + k-mer owners ----
// owner[code] = smallest token containing the k-mer = the token whose group the k-mer belongs to in the
// reference's k2GIDMap (homeless k-mers are assigned to their first token's group, WorkHorse.cpp:1612-1617).
// bits of x (32) spread to the even bit positions of a 64-bit word
static __device__ __forceinline__ uint64_t spread32(uint32_t v)
{
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
// the 11 low bits of v spread to the even positions of a 22-bit value
static __device__ __forceinline__ uint32_t spread11(uint32_t v)
{
    uint32_t x = v & 0x7FFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

// One WAVE per token, lane i = base i: the two bit planes of the 2-bit codes come from ballots, the packed forms are
// their interleave, and lane i builds the 11-mer that starts at i from an 11-bit window of each plane (first base
// most significant: integer order == lexicographic order, laurenize() == min, SeqUtils.cpp:89-97).  'N' packs as
// 'A' + a position-mask bit.  An 11-mer with an 'N' (a handful per merge, none for reads without N) cannot be a 22-bit
// code: it gets the identity the reference's std::map<std::string,int> would give it — the laurenized 11-mer over
// A < C < G < N < T (ASCII order; comp_tab maps N to N, SeqUtils.cpp:50-59) as a 33-bit key, claimed in a small
// open-addressing table whose slot index is the id: code = (1 << 22) + slot.
__global__ __launch_bounds__(256) void k_dm_pack_codes(DevMerge M)
{
    const int lane = threadIdx.x & 63;
    const uint32_t t = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (t == 0 && lane == 0 && M.d_ntok && *M.d_ntok > M.n_tok) atomicOr(&M.st->fail, 64u);      // more tokens than the launch was sized for
    if (t == 0 && lane == 0 && M.flag_pre) stage_flag_store(M.flag_pre, M.flag_pre_val);          // (everything in front of the merge is complete)
    if (t < dm_ntok(M)) {
        const uint32_t len = M.dx_len[t];
        if (len > 64 || len < M.min_len || M.min_len < kDevMinDR || M.stride > 64) { if (lane == 0) atomicOr(&M.st->fail, 1u); }
        else {
            const bool in = (uint32_t)lane < len;
            const uint32_t ch = in ? (uint32_t)(uint8_t)M.dx_chars[(uint64_t)t * M.stride + lane] : 0u;
            const bool isn = in && ch == 'N';
            const uint32_t c = ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 0u;
            const bool bad = in && ch != 'A' && ch != 'C' && ch != 'G' && ch != 'T' && ch != 'N';
            const uint32_t cr = (in && !isn) ? 3u - c : 0u;
            const uint64_t b0 = __ballot(in && (c & 1u)), b1 = __ballot(in && (c & 2u));      // forward planes, bit i = base i
            const uint64_t q0 = __ballot(cr & 1u), q1 = __ballot(cr & 2u);                    // complement planes, bit i = base i
            const uint64_t mf = __ballot(isn);
            const uint64_t any_bad = __ballot(bad);
            const uint32_t sh = 64u - len;                                                   // reversal: base i -> position len-1-i
            const uint64_t rq0 = __brevll(q0) >> sh, rq1 = __brevll(q1) >> sh, mr = __brevll(mf) >> sh;
            if (lane == 0) {
                uint64_t *pk = M.packed + (uint64_t)t * 4;
                pk[0] = spread32((uint32_t)b0) | (spread32((uint32_t)b1) << 1);
                pk[1] = spread32((uint32_t)(b0 >> 32)) | (spread32((uint32_t)(b1 >> 32)) << 1);
                pk[2] = spread32((uint32_t)rq0) | (spread32((uint32_t)rq1) << 1);
                pk[3] = spread32((uint32_t)(rq0 >> 32)) | (spread32((uint32_t)(rq1 >> 32)) << 1);
                M.tmask[(uint64_t)t * 2] = mf; M.tmask[(uint64_t)t * 2 + 1] = mr;
                M.pat_token[2 * t] = t + 2; M.pat_token[2 * t + 1] = t + 2;
                if (any_bad) atomicOr(&M.st->fail, 1u);
            }
            if ((uint32_t)lane + kClusterK <= len) {                                          // the 11-mer that starts at this lane
                const uint32_t x0 = (uint32_t)(b0 >> lane) & 0x7FFu, x1 = (uint32_t)(b1 >> lane) & 0x7FFu;
                const uint32_t xn = (uint32_t)(mf >> lane) & 0x7FFu;
                uint32_t code;
                if (xn == 0u) {
                    const uint32_t fwd = spread11(__brev(x0) >> 21) | (spread11(__brev(x1) >> 21) << 1);
                    const uint32_t rev = spread11(~x0) | (spread11(~x1) << 1);
                    code = fwd < rev ? fwd : rev;
                } else {
                    uint64_t fk = 0, rk = 0;
                    for (int i = 0; i < kClusterK; i++) {
                        const uint32_t c2 = ((x0 >> i) & 1u) | (((x1 >> i) & 1u) << 1);
                        const uint64_t c5 = ((xn >> i) & 1u) ? 3u : (c2 == 3u ? 4u : c2);          // A C G N T
                        const uint64_t cc = c5 == 0u ? 4u : c5 == 1u ? 2u : c5 == 2u ? 1u : c5 == 3u ? 3u : 0u;      // T G C N A
                        fk = (fk << 3) | c5;
                        rk |= cc << (3 * i);
                    }
                    const unsigned long long want = (fk < rk ? fk : rk) | (1ull << 40);
                    uint32_t h = (uint32_t)((want * 0x9E3779B97F4A7C15ull) >> 52) & (kDmBadSlots - 1u);
                    code = 0;
                    for (uint32_t probes = 0; probes < kDmBadSlots; probes++) {
                        const unsigned long long old = atomicCAS(&M.bk_key[h], 0ull, want);
                        if (old == 0ull) {
                            if (atomicAdd(&M.st->n_badk, 1u) >= kDmBadKmerCap) atomicOr(&M.st->fail, 1u);
                            code = (1u << 22) + h; break;
                        }
                        if (old == want) { code = (1u << 22) + h; break; }
                        h = (h + 1) & (kDmBadSlots - 1u);
                    }
                    if (!code) { atomicOr(&M.st->fail, 1u); code = 1u << 22; }
                }
                // look before the atomic: the variants of one repeat family share most 11-mers (1 400-way contention on a
                // few hundred addresses at 100 M reads), the owner is an early token and everybody after it only confirms
                if (!(M.ablate & 1u) && __hip_atomic_load(&M.owner[code], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > t) atomicMin(&M.owner[code], t);
                if (!(M.ablate & 2u)) M.codes[(uint64_t)t * M.kmax + lane] = code;
            }
        }
    }
}

// ---- 2. the greedy, order-dependent group assignment (clusterDRReads) ----
// One wave per token; wave w of W takes tokens w, w+W, ...  A wave only ever waits for tokens with a
// smaller index, i.e. for a wave that is in the same or an earlier round, so with all W waves resident
// (the grid is at most one block per CU) the smallest undecided token can always finish.
// Lane q owns the token's q-th k-mer: if an EARLIER token owns that k-mer the lane fetches that token's
// root (the first token of its group); the reference's scan ("first group whose count reaches
// kmer_clust_size on a repeated sighting", :1573-1590) is then a prefix count across lanes.  root_of[]
// words are their own flags: written once with an agent-scope store, polled with agent-scope loads
// (8 XCDs, private L2s).  Every spin is bounded: on a time-out the fail word is set and the host merges.
// The token's needle key of removeRedundantRepeats — (root, first 16 bases) — only needs its own root, so it is
// claimed and counted right here (step 4a).
static __device__ __forceinline__ uint32_t rset_hash(uint32_t g, uint32_t w, uint32_t log_size)
{
    return ((w * 0x9E3779B1u) ^ (g * 0x85EBCA6Bu)) >> (32u - log_size);
}
__global__ __launch_bounds__(1024) void k_dm_greedy(DevMerge M)
{
    if (dm_abandoned(M)) return;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t t = wave; t < dm_ntok(M); t += n_waves) {
        int nk = (int)M.dx_len[t] - kClusterK + 1;
        if (nk < 0) nk = 0;
        if (nk > 64) nk = 64;
        bool valid = false;
        uint32_t o = 0;
        if (lane < nk) {
            o = M.owner[M.codes[(uint64_t)t * M.kmax + lane] & 0x7FFFFFu];
            valid = o < t;                      // o == t: homeless k-mer (first seen in this token), not counted
        }
        uint32_t r = valid ? kUnres : kNoLane;
        bool gave_up = false;
        for (uint32_t spins = 0;; spins++) {
            if (valid && r == kUnres) r = __hip_atomic_load(&M.root_of[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((M.ablate & 8u) && r == kUnres) r = kNoLane;
            if (__ballot(valid && r == kUnres) == 0ull) break;
            if ((spins & 255u) == 255u) {
                const uint32_t f = __hip_atomic_load(&M.st->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (f != 0 || spins > (1u << 17)) { gave_up = true; break; }
            }
            __builtin_amdgcn_s_sleep(2);
        }
        if (gave_up) {
            if (lane == 0) atomicOr(&M.st->fail, 8u);
            if (r == kUnres) r = kNoLane;
        }
        // c = sightings of this lane's group among lanes 0..lane
        uint32_t c = 0;
        for (int j = 0; j < nk; j++) {
            const uint32_t rj = (uint32_t)__shfl((int)r, j);
            if (j <= lane && rj == r) c++;
        }
        const uint64_t win = __ballot(valid && r != kNoLane && c >= M.thr);
        uint32_t root = t;                      // no group reached the threshold: new group (:1595-1606)
        if (win) root = (uint32_t)__shfl((int)r, __ffsll((unsigned long long)win) - 1);
        if (lane == 0) {
            __hip_atomic_store(&M.root_of[t], root, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (M.ablate & 4u) continue;
            // 4a. needle key of this member
            const uint32_t g = root + 1, w = (uint32_t)M.packed[(uint64_t)t * 4];
            const unsigned long long want = ((unsigned long long)g << 32) | w;            // g >= 1: never 0
            const uint32_t mask = (1u << M.rset_log) - 1u;
            uint32_t h = rset_hash(g, w, M.rset_log);
            bool winner = false;
            for (;;) {
                unsigned long long old = __hip_atomic_load(&M.rset_key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (look first)
                if (old == 0ull) old = atomicCAS(&M.rset_key[h], 0ull, want);
                if (old == 0ull) { winner = true; break; }
                if (old == want) break;
                h = (h + 1) & mask;
            }
            atomicAdd(&M.rset_cnt[h], 1u);
            M.rd_slot[t] = winner ? (h | 0x80000000u) : h;
        }
    }
}

// (block_reserve: engine_internal.h)
// ---- 4. removeRedundantRepeats: a member is dropped iff a strictly shorter member of its group, or
// that member's reverse complement, occurs in it (equal-length members are distinct strings; the
// relation is transitive, so "blanked earlier" never matters).  t or rc(t) in s <=> t in s or in rc(s).
// The needles are indexed by (group root, first 16 bases): every member claims its key in an open-addressing
// set (4a, above) and the members of one key are laid out contiguously ({len | token << 32, bits lo, bits hi, N mask}):
// 4b allocates the keys' ranges, 4c fills them.  A member j then probes the index with every window of its own string
// and of its reverse complement in which a member (>= 23 bases) could still start, and compares only the few
// candidates that share the window's first 16 bases — instead of trying every shorter member at every shift.
__global__ __launch_bounds__(256) void k_dm_rd_bases(DevMerge M)
{
    if (dm_abandoned(M)) return;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t hs = 0;
    if (t < dm_ntok(M)) hs = M.rd_slot[t];
    const bool claim = (hs & 0x80000000u) != 0;         // this thread claimed the key: it allocates the key's range
    const uint32_t h = hs & 0x7FFFFFFFu;
    const uint32_t cnt = claim ? M.rset_cnt[h] : 0u;
    if (cnt > M.group_cap) atomicOr(&M.st->fail, 32u);
    const uint32_t base = block_reserve<256>(cnt, &M.hot[dm_hot(kHotRdCursor, 0)]);
    if (claim) M.rset_base[h] = base;
}
__global__ __launch_bounds__(256) void k_dm_rd_fill(DevMerge M)
{
    if (dm_abandoned(M)) return;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= dm_ntok(M)) return;
    const uint32_t h = M.rd_slot[t] & 0x7FFFFFFFu;
    const uint32_t pos = M.rset_base[h] + atomicAdd(&M.rset_fill[h], 1u);
    uint64_t *d = M.rents + (uint64_t)pos * 4;
    d[0] = (uint64_t)M.dx_len[t] | ((uint64_t)t << 32);
    d[1] = M.packed[(uint64_t)t * 4];
    d[2] = M.packed[(uint64_t)t * 4 + 1];
    d[3] = M.tmask[(uint64_t)t * 2];
}
// One wave per member j: the lanes probe j's windows (both orientations) in parallel, then every window that
// hit a key has its candidates compared 64 at a time.
static __device__ __forceinline__ uint64_t shfl64(uint64_t v, int src)
{
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
__global__ __launch_bounds__(256) void k_dm_redundant(DevMerge M)
{
    if (dm_abandoned(M)) return;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t mask = (1u << M.rset_log) - 1u;
    uint32_t n_surv = 0;                                // this wave's contribution to the state word
    for (uint32_t j = wave; j < dm_ntok(M); j += n_waves) {
        const uint32_t g = M.root_of[j] + 1;
        const uint32_t lenj = M.dx_len[j];
        const uint64_t *pj = M.packed + (uint64_t)j * 4;
        const uint64_t f0 = pj[0], f1 = pj[1], r0 = pj[2], r1 = pj[3];
        const uint64_t mfj = M.tmask[(uint64_t)j * 2], mrj = M.tmask[(uint64_t)j * 2 + 1];
        if (lenj < M.min_len || lenj > 64u) continue;   // (cannot happen once k_dm_pack_codes has passed the token: belt and braces)
        const uint32_t nwin = lenj - (M.min_len - 1u);  // starts 0 .. lenj - min_len: a member is >= min_len (23 by default) long and shorter than lenj
        bool found = false;
        for (uint32_t wb = 0; wb < 2 * nwin && !found; wb += 64) {
            const uint32_t wq = wb + lane;
            const bool act = wq < 2 * nwin;
            const uint32_t o = wq >= nwin ? 1u : 0u, p = wq - o * nwin;
            const uint64_t h0 = o ? r0 : f0, h1 = o ? r1 : f1;
            const uint64_t w0 = shr128_lo(h0, h1, 2 * (act ? p : 0u)), w1 = shr128_hi(h1, 2 * (act ? p : 0u));
            const uint64_t wm = (o ? mrj : mfj) >> (act ? p : 0u);           // 'N' positions of the window
            uint32_t cnt = 0, base = 0;
            if (act && !(M.ablate & 32u)) {
                const uint32_t w = (uint32_t)w0;
                const unsigned long long want = ((unsigned long long)g << 32) | w;
                uint32_t h = rset_hash(g, w, M.rset_log);
                for (;;) {
                    // (the slot's range is requested with its key: one round trip instead of two)
                    const unsigned long long kk = M.rset_key[h];
                    const uint32_t c_h = M.rset_cnt[h], b_h = M.rset_base[h];
                    if (kk == 0ull) break;
                    if (kk == want) { cnt = c_h; base = b_h; break; }
                    h = (h + 1) & mask;
                }
            }
            uint64_t hits = __ballot(cnt > 0);
            if (M.ablate & 16u) hits = 0;
            while (hits && !found) {
                const int src = __ffsll((unsigned long long)hits) - 1;
                hits &= hits - 1;
                const uint32_t cnt_s = (uint32_t)__shfl((int)cnt, src), base_s = (uint32_t)__shfl((int)base, src);
                const uint32_t p_s = (uint32_t)__shfl((int)p, src);
                const uint64_t w0_s = shfl64(w0, src), w1_s = shfl64(w1, src), wm_s = shfl64(wm, src);
                for (uint32_t c0 = 0; c0 < cnt_s && !found; c0 += 64) {
                    const uint32_t c = c0 + lane;
                    bool hit = false;
                    if (c < cnt_s) {
                        const ulonglong2 *ent = reinterpret_cast<const ulonglong2 *>(M.rents) + (uint64_t)(base_s + c) * 2;
                        const ulonglong2 ea = ent[0], eb = ent[1];          // (the whole entry at once)
                        const uint32_t leni = (uint32_t)ea.x & 0xFFu;
                        if (leni < lenj && p_s + leni <= lenj) {
                            uint64_t m0, m1;
                            mask128(leni, m0, m1);
                            const uint64_t lm = leni >= 64 ? ~0ull : ((1ull << leni) - 1ull);
                            hit = (w0_s & m0) == ea.y && (w1_s & m1) == eb.x && (wm_s & lm) == eb.y;
                        }
                    }
                    if (__ballot(hit)) found = true;
                }
            }
        }
        if (lane == 0) M.blank[j] = found ? 1 : 0;
        if (!found) n_surv++;
    }
    // (one atomic per block, on the block's stripe of the counter: dm_hot, engine_internal.h)
    __shared__ uint32_t surv_w[4];
    if (lane == 0) surv_w[threadIdx.x >> 6] = n_surv;
    __syncthreads();
    if (threadIdx.x == 0) { const uint32_t tot = surv_w[0] + surv_w[1] + surv_w[2] + surv_w[3]; if (tot) atomicAdd(&M.hot[dm_hot(kHotSurvivors, blockIdx.x)], tot); }
}

// ---- 6a. anchor keys: every 16-mer at offset 0..7 of a pattern (see kernels.hip, pass-2 fast path).  A member that
// survived is a pattern (pid 2t) and so is its reverse complement (pid 2t + 1, WorkHorse.cpp:690-697); entry
// e = pid * 8 + r = 16 t + 8 o + r.  Distinct keys are claimed in an open-addressing set and counted (thread per entry:
// a wave-per-member form inside k_dm_redundant ran the claims at a quarter of the lanes behind that kernel's probe
// latency, 13 + 11 us -> 48 us).
__global__ __launch_bounds__(256) void k_dm_keys(DevMerge M)
{
    if (M.flag_blank && blockIdx.x == 0 && threadIdx.x == 0) stage_flag_store(M.flag_blank, M.flag_blank_val);      // (k_dm_redundant is complete: blank[] is final)
    if (dm_abandoned(M)) return;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = e >> 4, o = (e >> 3) & 1u, r = e & 7u;
    uint32_t slot = 0xFFFFFFFFu, my_key = 0xFFFFFFFFu, won = 0u;
    const bool in_range = t < dm_ntok(M);               // (no early return: the block meets at the end)
    // (anchor windows every 4 bases — patterns of 19 .. 22 bases — : offsets 0 .. 3 only; the entry numbering stays 16 per token)
    if (in_range && !M.blank[t] && r < (1u << M.akey_shift) && !(M.ablate & 512u)) {
        const uint32_t key = (uint32_t)shr128_lo(M.packed[(uint64_t)t * 4 + 2 * o], M.packed[(uint64_t)t * 4 + 2 * o + 1], 2 * r);
        const uint32_t kmask = (1u << M.kset_log) - 1u;
        const unsigned long long want = (unsigned long long)key | (1ull << 32);
        uint32_t h = kset_hash(key, M.kset_log);
        bool winner = false;
        for (;;) {
            unsigned long long old = __hip_atomic_load(&M.kset_key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // (look first)
            if (M.ablate & 256u) break;
            if (old == 0ull) old = atomicCAS(&M.kset_key[h], 0ull, want);
            if (old == 0ull) { winner = true; break; }
            if (old == want) break;
            h = (h + 1) & kmask;
        }
        if (!(M.ablate & 64u)) atomicAdd(&M.kset_cnt[h], 1u);
        slot = winner ? (h | 0x80000000u) : h;
        if (winner) {
            my_key = key; won = 1u;
            if (key == 0xFFFFFFFFu) atomicOr(&M.st->all_t, 1u);
        }
    }
    if (in_range) M.ent_slot[e] = slot;
    // the key count and the smallest key: one atomic each per block, on the block's stripe (dm_hot)
    __shared__ uint32_t kc_w[4], km_w[4];
    const int lane = threadIdx.x & 63;
    const uint32_t cw = (uint32_t)__popcll(__ballot(won != 0u));
    uint32_t mw = my_key;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mw = min(mw, (uint32_t)__shfl_xor((int)mw, off));
    if (lane == 0) { kc_w[threadIdx.x >> 6] = cw; km_w[threadIdx.x >> 6] = mw; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t tot = kc_w[0] + kc_w[1] + kc_w[2] + kc_w[3];
        if (tot && !(M.ablate & 128u)) {
            atomicAdd(&M.hot[dm_hot(kHotKeys, blockIdx.x)], tot);
            atomicMin(&M.hot[dm_hot(kHotK0, blockIdx.x)], min(min(km_w[0], km_w[1]), min(km_w[2], km_w[3])));
        }
    }
}

// table size: load <= 1/3 (<= 1/2 at the limits), as build_anchors (merge.cpp).  Up to 2^14 keys: exact keys in
// LDS; up to 2^15: a 2^16-slot table whose 16-bit fingerprints are staged in LDS (a superset filter, 2 * 2^-16
// false positives per probe — the flagged reads are verified exactly anyway); beyond: exact keys probed in L2.
static __device__ __forceinline__ void dm_table_params(uint32_t n, uint32_t tab_log_alloc, uint32_t &ls, uint32_t &mode)
{
    ls = 0; mode = 0;
    for (uint32_t log = 10; log <= 15 && !ls; log++) {
        const uint32_t size = 1u << log;
        if (n * 3 > size && log != 15) continue;
        if (n * 2 > size) continue;
        ls = log;
    }
    if (!ls && n * 2 <= 65536u && tab_log_alloc >= 16) { ls = 16; mode = 3; }
    for (uint32_t log = 17; log <= tab_log_alloc && !ls; log++) {
        const uint32_t size = 1u << log;
        if (n * 3 > size && log != tab_log_alloc) continue;
        if (n * 2 > size) continue;
        ls = log; mode = 2;
    }
    if (ls > tab_log_alloc) ls = 0;
}

// ---- 6b. the keys' entry ranges of the verification index (count -> block allocation -> fill), and the keys
// themselves into the cuckoo table.  Two-choice cuckoo insertion, all keys at once: a key is always either in the
// table or in exactly one thread's hand (atomicExch).  0xFFFFFFFF marks a free slot; the all-T key itself is handled
// by the last kernel.
// (1 024 threads per block: the cursor atomic RETURNS a value, and returning atomics on one address retire every ~35 ns,
// not every 3 — 2 600 blocks of 256 were 90 of this kernel's 102 us at 100 M reads)
__global__ __launch_bounds__(1024) void k_dm_key_bases_insert(DevMerge M)
{
    if (dm_abandoned(M)) return;
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tok = e >> 4;
    // every thread derives the table shape from the key count (a handful of scalar instructions); thread 0 records it
    uint32_t ls, mode, n_keys = 0;
#pragma unroll
    for (uint32_t q = 0; q < kDmHotStripes; q++) n_keys += M.hot[dm_hot(kHotKeys, q)];          // (uniform addresses: scalar loads)
    dm_table_params(n_keys, M.tab_log_alloc, ls, mode);
    if (e == 0) {
        uint32_t n_surv = 0, k0 = 0xFFFFFFFFu;
        for (uint32_t q = 0; q < kDmHotStripes; q++) { n_surv += M.hot[dm_hot(kHotSurvivors, q)]; k0 = min(k0, M.hot[dm_hot(kHotK0, q)]); }
        if (!ls || n_keys == 0) atomicOr(&M.st->fail, 2u);
        M.st->log_size = ls; M.st->tab_mode = mode; M.st->n_survivors = n_surv; M.st->n_patterns = 2 * n_surv;
        M.st->n_keys = n_keys; M.st->k0 = k0;
    }
    uint32_t hs = 0xFFFFFFFFu;
    if (tok < dm_ntok(M)) hs = M.ent_slot[e];
    const bool claim = hs != 0xFFFFFFFFu && (hs & 0x80000000u);      // the key's claimant allocates its entry range
    const uint32_t h = hs & 0x7FFFFFFFu;
    const uint32_t ebase = (M.ablate & 2048u) ? 0u : block_reserve<1024>(claim ? M.kset_cnt[h] : 0u, &M.hot[dm_hot(kHotEntCursor, 0)]);
    if (!claim) return;
    M.kset_base[h] = ebase;
    if (!ls || (M.st->fail & ~2u) || (M.ablate & 1024u)) return;
    uint32_t cur = (uint32_t)M.kset_key[h];
    if (mode == 2) {
        // key sets beyond the LDS tiers: the exact table is probed in L2, behind a 2^20-bit Bloom filter in LDS
        // BLOCKED: one hash — its top 15 bits pick the word, bits 12..16 and 7..11 two bits inside it (ak_bloom_word / ak_bloom_bits,
        // engine_internal.h): the probe is one LDS read and 13 instructions per window instead of two reads and 21, for 7.1 % false
        // positives instead of 6.2 % at 150 k keys (configs[4]'s probe is VALU-issue bound, NOTES r05)
        const uint32_t hb = ak_hash(cur, M.m1);
        atomicOr(&M.anchor_fp[ak_bloom_word(hb)], ak_bloom_bits(hb));
    }
    if (cur == 0xFFFFFFFFu) return;
    const uint32_t rsh = 32u - ls;
    uint32_t pos = ak_h(cur, M.m1, rsh);
    // a free slot of its own first: only a key whose two slots are both taken starts an eviction chain (every link of a
    // chain is a dependent atomic round trip; near load 1/2 — 30 k keys in 2^16 slots at 100 M reads — exchanging
    // unconditionally made 46 % of the keys start one and the longest took ~100 us)
    if (atomicCAS(&M.anchor_tab[pos], 0xFFFFFFFFu, cur) == 0xFFFFFFFFu) return;
    {
        const uint32_t p2 = ak_h(cur, M.m2, rsh);
        if (p2 != pos && atomicCAS(&M.anchor_tab[p2], 0xFFFFFFFFu, cur) == 0xFFFFFFFFu) return;
    }
    for (int kicks = 0; kicks < 1000; kicks++) {
        const uint32_t old = atomicExch(&M.anchor_tab[pos], cur);
        if (old == 0xFFFFFFFFu) return;
        cur = old;
        const uint32_t p1 = ak_h(cur, M.m1, rsh), p2 = ak_h(cur, M.m2, rsh);
        pos = (pos == p1) ? p2 : p1;
    }
    atomicOr(&M.st->fail, 4u);
}
// ---- 6c. fill the verification index: {r | len << 3 | pid << 32, pattern bits lo, pattern bits hi, N mask} — and, in the
// same launch (both only need 6b): unused table slots get a member key, so that a probe never matches by accident;
// tab_mode 3: 16 bits per slot = the other slot index of the slot's key (anchor_probe_fp), two per word; the per-token
// results + state words go straight into pinned host memory (a few 10 KB over PCIe, no copy calls): the helper thread
// rebuilds the host view from them
__global__ __launch_bounds__(256) void k_dm_fill_finish(DevMerge M)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    const bool dead = dm_abandoned(M);                  // (the kernels before this one left at once: only the state words are exported)
    if (!dead && (w >> 4) < dm_ntok(M)) {
        const uint32_t hs = M.ent_slot[w];
        if (hs != 0xFFFFFFFFu) {
            const uint32_t h = hs & 0x7FFFFFFFu, pid = w >> 3, r = w & 7u, t = pid >> 1, o = pid & 1u;
            const uint32_t pos = M.kset_base[h] + atomicAdd(&M.kset_fill[h], 1u);
            // two planes of 16 bytes per entry: {meta, bases 0..31} decides nearly every comparison; {bases 32..63, N mask}
            // (plane 2, M.ent_cap entries further) is only read for an entry whose first 32 bases equal the read's
            uint64_t *d = M.ents + (uint64_t)pos * 2, *d2 = M.ents + ((uint64_t)M.ent_cap + pos) * 2;
            d[0] = (uint64_t)(r | ((uint32_t)M.dx_len[t] << 3)) | ((uint64_t)pid << 32);
            d[1] = M.packed[(uint64_t)t * 4 + 2 * o];
            d2[0] = M.packed[(uint64_t)t * 4 + 2 * o + 1];
            d2[1] = M.tmask[(uint64_t)t * 2 + o];
        }
    }
    if (w == 0) *M.h_st = *M.st;
    if (dead) return;
    if (!M.x_on && w < dm_ntok(M)) { M.h_root[w] = M.root_of[w]; M.h_blank[w] = M.blank[w]; }      // (x_on: the view comes from k_dmx_*; the host fetches these if it must)
    const uint32_t ls = M.st->log_size, k0 = M.st->k0;
    const bool fill = ls && !M.st->all_t;
    if (fill && w < (1u << ls) && M.anchor_tab[w] == 0xFFFFFFFFu) M.anchor_tab[w] = k0;
    if (M.st->tab_mode != 3 || w >= (1u << 15)) return;
    uint32_t out = 0;
#pragma unroll
    for (int q = 0; q < 2; q++) {
        uint32_t v = M.anchor_tab[2 * w + q];
        if (fill && v == 0xFFFFFFFFu) v = k0;           // (the slot's own thread stores the same value)
        const uint32_t p1 = ak_hash(v, M.m1) >> 16, p2 = ak_hash(v, M.m2) >> 16, slot = 2 * w + q;
        out |= (slot == p1 ? p2 : p1) << (16 * q);       // the key's other slot (anchor_probe_fp, kernels.hip)
    }
    M.anchor_fp[w] = out;
}

// ---- 7. the host view of the merge (crass_merge_view, include/crass_hip.h) assembled on the device ----
// What the host used to rebuild from root_of[] / blank[] on a helper thread (token arena, mDR2GIDMap as flat arrays, the
// pattern list of createNonRedundantSet in WorkHorse.cpp:690-697 order) — 0.6 ms for 42 k tokens, the critical path of a
// rank whose shard of the reads is an eighth of the job.  Five small launches on a side stream, beside the merge's last
// three kernels; the result is ONE dense blob that a DMA engine moves to pinned host memory.
//   k_dmx_count   per root: members, survivors, survivors' characters (one atomic per distinct root and wave)
//   k_dmx_tiles   per tile of 1024 tokens: sums of {is root, group size, survivors, survivors' chars (at roots), length}
//   k_dmx_apply   exclusive prefixes -> GIDs (roots in token order = nextFreeGID++ order, WorkHorse.cpp:1598), group / pattern
//                 bases, tok_off; copies the token strings; publishes the totals and the blob layout
//   k_dmx_place   every token claims a slot in its group's member list (any order)
//   k_dmx_rank    every member counts the members of its group that precede it — by token for grp_tokens, by (length, token)
//                 among the survivors for the pattern list (remove_redundant's order, merge.cpp) — and writes its outputs
// A group with more than x_group_cap members is not ranked here (the counting is quadratic in the group): ok = 0, and the
// host builds the view as before.
static __device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off);
    return v;
}
static __device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)v, off);
        if (lane >= off) v += u;
    }
    return v;
}

// A block's tokens by root in LDS (open addressing, 2 x the tokens of a block): the global words of one root then see one
// atomic per BLOCK instead of one per token.  The roots are early tokens, i.e. a few hundred neighbouring words: 78 k atomics on
// them took 107 us of a launch over 42 k tokens, and 26 k returning ones 104 us (rocprofv3, round 4).
#define DMX_BLOCK 1024
#define DMX_SLOTS 2048
static __device__ __forceinline__ uint32_t dmx_claim(uint32_t *hkey, uint32_t r)
{
    uint32_t slot = (r * 0x9E3779B1u) >> 21;            // 11 bits
    for (;;) {
        const uint32_t old = atomicCAS(&hkey[slot], 0xFFFFFFFFu, r);
        if (old == 0xFFFFFFFFu || old == r) return slot;
        slot = (slot + 1u) & (DMX_SLOTS - 1u);
    }
}

__global__ __launch_bounds__(DMX_BLOCK) void k_dmx_count(DevMerge M)
{
    if (dm_abandoned(M)) return;
    // (group sizes only: the export runs behind k_dm_greedy, beside the kernels that decide which members survive — the pattern
    // list, which needs blank[], is put together by the host from this view and blank[]; see launch_device_merge)
    __shared__ uint32_t hkey[DMX_SLOTS], hcnt[DMX_SLOTS];
    for (uint32_t i = threadIdx.x; i < DMX_SLOTS; i += DMX_BLOCK) { hkey[i] = 0xFFFFFFFFu; hcnt[i] = 0u; }
    __syncthreads();
    const uint32_t t = blockIdx.x * DMX_BLOCK + threadIdx.x;
    if (t < dm_ntok(M)) {
        const uint32_t slot = dmx_claim(hkey, M.root_of[t]);
        atomicAdd(&hcnt[slot], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < DMX_SLOTS; i += DMX_BLOCK) {
        const uint32_t r = hkey[i];
        if (r == 0xFFFFFFFFu) continue;
        atomicAdd(&M.x_size[r], hcnt[i]);
    }
}

// the five scan values of token t (zero beyond the token count)
static __device__ __forceinline__ void dmx_vals(const DevMerge &M, uint32_t t, uint32_t n, uint32_t v[kDmxVals])
{
    v[0] = v[1] = v[2] = v[3] = v[4] = 0u;
    if (t >= n) return;
    v[4] = M.dx_len[t];
    if (M.root_of[t] == t) { v[0] = 1u; v[1] = M.x_size[t]; }      // (v[2], v[3] — survivors and their characters — stay 0: the host's part)
}

__global__ __launch_bounds__(1024) void k_dmx_tiles(DevMerge M)
{
    if (dm_abandoned(M)) return;
    __shared__ uint32_t part[16][kDmxVals + 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t t = blockIdx.x * 1024u + threadIdx.x;
    uint32_t v[kDmxVals];
    dmx_vals(M, t, dm_ntok(M), v);
    uint32_t mx = v[1];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
#pragma unroll
    for (uint32_t q = 0; q < kDmxVals; q++) { const uint32_t s = wave_sum(v[q]); if (lane == 0) part[w][q] = s; }
    if (lane == 0) part[w][kDmxVals] = mx;
    __syncthreads();
    if (threadIdx.x < kDmxVals) {
        uint32_t s = 0;
        for (int i = 0; i < 16; i++) s += part[i][threadIdx.x];
        M.x_tile[blockIdx.x * kDmxVals + threadIdx.x] = s;
    }
    if (threadIdx.x == kDmxVals) {
        uint32_t m = 0;
        for (int i = 0; i < 16; i++) m = max(m, part[i][kDmxVals]);
        if (m) atomicMax(&M.x_tile[kDmxTiles * kDmxVals], m);
    }
}

__global__ __launch_bounds__(1024) void k_dmx_apply(DevMerge M)
{
    __shared__ uint32_t part[16][kDmxVals], ptot[16][kDmxVals], pbef[16][kDmxVals];
    __shared__ uint32_t before[kDmxVals], total[kDmxVals];
    __shared__ DevViewTotals T;
    const bool dead = dm_abandoned(M);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t n = dm_ntok(M);
    const uint32_t n_tiles = gridDim.x, tile = blockIdx.x;
    if (dead) {
        if (tile == 0 && threadIdx.x == 0) { DevViewTotals z{}; *M.x_tot = z; M.x_htot[0] = z; M.x_htot[1] = z; }
        return;
    }
    // (a) this tile's exclusive prefix and the totals, from the tile sums of the launch before (<= 1024 tiles: one per thread)
#pragma unroll
    for (uint32_t q = 0; q < kDmxVals; q++) {
        const uint32_t tv = threadIdx.x < n_tiles ? M.x_tile[threadIdx.x * kDmxVals + q] : 0u;
        const uint32_t st = wave_sum(tv), sb = wave_sum(threadIdx.x < tile ? tv : 0u);
        if (lane == 0) { ptot[w][q] = st; pbef[w][q] = sb; }
    }
    __syncthreads();
    if (threadIdx.x < kDmxVals) {
        uint32_t a = 0, b = 0;
        for (int i = 0; i < 16; i++) { a += ptot[i][threadIdx.x]; b += pbef[i][threadIdx.x]; }
        total[threadIdx.x] = a; before[threadIdx.x] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        DevViewTotals t{};
        t.n_tok = n; t.n_groups = total[0]; t.n_kept = total[2]; t.tok_chars = total[4]; t.kept_chars = total[3];
        t.max_group = M.x_tile[kDmxTiles * kDmxVals];
        t.ok = (t.max_group <= M.x_group_cap && total[1] == n) ? 1u : 0u;
        t.lay = view_layout(n, t.n_groups, 2ull * t.n_kept, t.tok_chars, 2ull * t.kept_chars);
        T = t;
        // (ok: the host mirror is written by the LAST kernel, behind the blob; x_htot[1] is this kernel's: the host starts the copy
        // of the token half of the blob — tok_off, the token strings, grp_off: everything in front of lay.grp_tokens is complete
        // when this kernel ends — while the kernels behind it rank the members)
        if (tile == 0) { *M.x_tot = t; if (!t.ok) M.x_htot[0] = t; M.x_htot[1] = t; }
    }
    __syncthreads();
    if (!T.ok) return;
    // (b) exclusive scan inside the tile
    const uint32_t t = tile * 1024u + threadIdx.x;
    uint32_t v[kDmxVals], ex[kDmxVals];
    dmx_vals(M, t, n, v);
#pragma unroll
    for (uint32_t q = 0; q < kDmxVals; q++) {
        const uint32_t inc = wave_incl_scan(v[q], lane);
        ex[q] = inc - v[q];
        if (lane == 63) part[w][q] = inc;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < kDmxVals; q++) {
        uint32_t s = before[q];
        for (int i = 0; i < w; i++) s += part[i][q];
        ex[q] += s;
    }
    uint8_t *blob = M.x_blob;
    uint64_t *tok_off = reinterpret_cast<uint64_t *>(blob + T.lay.tok_off);
    uint64_t *grp_off = reinterpret_cast<uint64_t *>(blob + T.lay.grp_off);
    if (t < n) {
        tok_off[t] = ex[4];
        if (v[0]) {                                      // a root: its group's dense id and bases
            M.x_gid[t] = ex[0]; M.x_goff[t] = ex[1]; M.x_pat0[t] = 2u * ex[2]; M.x_pch0[t] = 2u * ex[3];
            grp_off[ex[0]] = ex[1];
        }
    }
    if (t == 0) {
        tok_off[n] = T.tok_chars; grp_off[T.n_groups] = n;
        reinterpret_cast<uint64_t *>(blob + T.lay.pat_off)[2ull * T.n_kept] = 2ull * T.kept_chars;
    }
    // (c) the token strings, back to back.  Every (token, character) pair of the tile is one independent load + store (a loop
    // over the wave's tokens with lane = character waits for each row in turn: 64 dependent round trips, 38 us of this kernel)
    __shared__ uint32_t s_off[1024];
    __shared__ uint8_t s_len[1024];
    s_off[threadIdx.x] = ex[4]; s_len[threadIdx.x] = (uint8_t)v[4];
    __syncthreads();
    char *tok_chars = reinterpret_cast<char *>(blob + T.lay.tok_chars);
    const uint32_t stride = M.stride, cpr = stride >> 3;      // 8-byte chunks per row (the stride is a multiple of 16)
    for (uint32_t q = threadIdx.x; q < 1024u * cpr; q += 1024u) {
        const uint32_t tk = q / cpr, c8 = (q - tk * cpr) * 8u, len = s_len[tk];
        if (c8 >= len) continue;
        const uint2 w = *reinterpret_cast<const uint2 *>(M.dx_chars + (uint64_t)(tile * 1024u + tk) * stride + c8);
        char *dst = tok_chars + s_off[tk] + c8;
        const uint32_t nb = min(8u, len - c8);
#pragma unroll
        for (uint32_t b = 0; b < 8; b++) if (b < nb) dst[b] = (char)(((b < 4 ? w.x : w.y) >> (8u * (b & 3u))) & 0xFFu);
    }
}

__global__ __launch_bounds__(DMX_BLOCK) void k_dmx_place(DevMerge M)
{
    if (!M.x_tot->ok) return;
    __shared__ uint32_t hkey[DMX_SLOTS], hcnt[DMX_SLOTS], hbase[DMX_SLOTS];
    for (uint32_t i = threadIdx.x; i < DMX_SLOTS; i += DMX_BLOCK) { hkey[i] = 0xFFFFFFFFu; hcnt[i] = 0u; }
    __syncthreads();
    const uint32_t t = blockIdx.x * DMX_BLOCK + threadIdx.x;
    const bool act = t < M.x_tot->n_tok;
    uint32_t r = 0, slot = 0, local = 0;
    if (act) {
        r = M.root_of[t];
        slot = dmx_claim(hkey, r);
        local = atomicAdd(&hcnt[slot], 1u);             // rank among the block's members of the group (any order)
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < DMX_SLOTS; i += DMX_BLOCK)
        if (hkey[i] != 0xFFFFFFFFu) hbase[i] = atomicAdd(&M.x_fill[hkey[i]], hcnt[i]);      // one returning atomic per root and block
    __syncthreads();
    if (act) M.x_members[M.x_goff[r] + hbase[slot] + local] = t;
}

static __device__ __forceinline__ char dmx_comp(char c)                // SeqUtils.cpp:50-59 over the device merge's alphabet
{
    return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
}

// The member lists lie in GID order, so 64 consecutive entries belong to a run of consecutive groups whose lists are one
// contiguous range [lo, hi).  A block takes 64 entries (lane = entry, the same in each of its 16 waves) and its waves share the
// range 64 keys at a time: one coalesced load, then every key goes to all lanes through an SGPR (v_readlane) and each lane
// counts it if it lies in its own group's part of the range; the 16 partial counts per entry meet in LDS.
// (Measured on the way, 42 k tokens in groups of up to 1 800: a wave per 64 entries walking the whole range alone — 650 waves,
// one per SIMD, nothing to hide the dependent adds behind — 260 us; a tile of the range in LDS read by all lanes at once 78 us
// with 32-bit reads, 133 us with 128-bit ones (no broadcast), and the LDS it held kept pass 2's probe kernel, which stages its
// table there, off the CUs: 144 -> 243 us.)
// A group of 65 .. DMX_SORT_MAX members is ranked by SORTING its member list in LDS, one block per group, by token (grp_tokens):
// a bitonic sort of <= 2 048 keys (66 steps at the most) instead of group-size^2 comparisons — 42 k tokens in 64 groups of up to
// 1 800 were 27-50 M counting steps over every CU, 65 us alone and 93 us beside pass 2's probe, which lost 50 us to them
// (profiles/NOTES_r06.md).  The pattern list's order (length, then token, among the survivors) is the host's business since the
// export moved in front of k_dm_redundant.
// (2 048 keys: 16 KB of LDS for the two arrays — pass 2's probe holds 128 KB of a CU's 160 KB, and with 32 KB here whichever of
// the two kernels reached a CU first kept the other off it: the probe took 149 or 190 us depending on the launch's luck)
#define DMX_SORT_MAX 2048u
static __device__ __forceinline__ void dmx_bitonic(uint32_t *keys, uint32_t n_pad)
{
    for (uint32_t k = 2; k <= n_pad; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < n_pad; i += blockDim.x) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint32_t a = keys[i], b = keys[l];
                    const bool up = (i & k) == 0u;
                    if ((a > b) == up) { keys[i] = b; keys[l] = a; }
                }
            }
            __syncthreads();
        }
}
__global__ __launch_bounds__(1024) void k_dmx_sort(DevMerge M)
{
    const DevViewTotals T = *M.x_tot;
    if (!T.ok) return;
    __shared__ uint32_t keys[DMX_SORT_MAX];
    uint8_t *blob = M.x_blob;
    const uint64_t *grp_off = reinterpret_cast<const uint64_t *>(blob + T.lay.grp_off);
    uint32_t *grp_tokens = reinterpret_cast<uint32_t *>(blob + T.lay.grp_tokens);
    for (uint32_t gid = blockIdx.x; gid < T.n_groups; gid += gridDim.x) {
        const uint32_t g0 = (uint32_t)grp_off[gid], gs = (uint32_t)grp_off[gid + 1] - g0;
        if (gs <= 64u || gs > M.x_sort_max) continue;                   // (uniform: the waves' loop below / k_dmx_rank take those)
        uint32_t n_pad = 128;
        while (n_pad < gs) n_pad <<= 1;
        for (uint32_t i = threadIdx.x; i < n_pad; i += 1024u) keys[i] = i < gs ? M.x_members[g0 + i] : 0xFFFFFFFFu;
        __syncthreads();
        dmx_bitonic(keys, n_pad);
        for (uint32_t i = threadIdx.x; i < gs; i += 1024u) grp_tokens[g0 + i] = keys[i] + 2u;
        __syncthreads();                                                // (the next group reuses keys)
    }
    // groups of up to 64 members: a wave each, a member per lane, every lane counting the smaller tokens of its group
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 16u + (threadIdx.x >> 6), n_waves = gridDim.x * 16u;
    for (uint32_t gid = wave; gid < T.n_groups; gid += n_waves) {
        const uint32_t g0 = (uint32_t)grp_off[gid], gs = (uint32_t)grp_off[gid + 1] - g0;
        if (gs > 64u) continue;
        const uint32_t t = (uint32_t)lane < gs ? M.x_members[g0 + lane] : 0xFFFFFFFFu;
        uint32_t ra = 0;
        for (uint32_t q = 0; q < gs; q++) ra += (uint32_t)__builtin_amdgcn_readlane((int)t, (int)q) < t ? 1u : 0u;
        if ((uint32_t)lane < gs) grp_tokens[g0 + ra] = t + 2u;
    }
}

// the groups k_dmx_sort leaves out (more than x_sort_max members): every member counts the members of its group with a smaller
// token.  The member lists lie in GID order, so 64 consecutive entries belong to a run of consecutive groups whose
// lists are one contiguous range [lo, hi): a block takes 64 entries (lane = entry, the same in each of its 16 waves) and its
// waves share the range 64 keys at a time — one coalesced load, every key to all lanes through an SGPR (v_readlane), each lane
// counting it if it lies in its own group's part of the range; the 16 partial counts per entry meet in LDS.
#define DMX_RW 16                                    // waves per block
__global__ __launch_bounds__(64 * DMX_RW) void k_dmx_rank(DevMerge M)
{
    // (the totals are read field by field: a private copy of the struct is "promoted" to LDS by the compiler — 80 bytes x 1 024
    // threads = 80 KB for this kernel, which then could not share a CU with pass 2's probe (128 KB) and sat out the probe's
    // whole 150 us behind it, whatever its stream's priority)
    const DevViewTotals *TT = M.x_tot;
    if (!TT->ok) return;
    const uint32_t t_ntok = TT->n_tok;
    const uint64_t t_grp_tokens = TT->lay.grp_tokens;
    auto publish = [&]() {
        if (blockIdx.x == 0 && threadIdx.x < sizeof(DevViewTotals) / 4)
            reinterpret_cast<uint32_t *>(M.x_htot)[threadIdx.x] = reinterpret_cast<const uint32_t *>(TT)[threadIdx.x];      // (the host only reads the blob after the stream's event)
    };
    if (TT->max_group <= M.x_sort_max) { publish(); return; }      // (nothing for this kernel: k_dmx_sort has ranked every group)
    __shared__ uint32_t acc[DMX_RW][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint8_t *blob = M.x_blob;
    for (uint32_t chunk = blockIdx.x; chunk * 64u < t_ntok; chunk += gridDim.x) {
        const uint32_t i = chunk * 64u + (uint32_t)lane;
        const bool act = i < t_ntok;
        uint32_t t = 0, g0 = 0, gs = 0;
        if (act) {
            t = M.x_members[i];
            const uint32_t r = M.root_of[t];
            g0 = M.x_goff[r]; gs = M.x_size[r];
        }
        // (the lanes of a chunk that belong to such groups: their lists form one contiguous range, groups of up to x_sort_max members
        // in between are skipped by the per-lane test below)
        const bool part = act && gs > M.x_sort_max;
        const uint64_t am = __ballot(part);
        if (am == 0ull) continue;                       // (uniform over the block: every wave holds the same 64 entries)
        const int first = __ffsll((unsigned long long)am) - 1, lastl = 63 - __clzll((unsigned long long)am);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)g0, first);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(g0 + gs), lastl);
        uint32_t ra = 0;
        for (uint32_t base = lo + 64u * (uint32_t)w; base < hi; base += 64u * DMX_RW) {
            const uint32_t cur = base + (uint32_t)lane < hi ? M.x_members[base + lane] : 0u;
            const uint32_t cnt = min(hi - base, 64u);
            const uint32_t rel = base - g0;             // per lane: key q of this chunk is member (rel + q) of the lane's group iff < gs
            for (uint32_t q = 0; q < cnt; q++) {
                const uint32_t k2 = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)q);
                ra += (rel + q < gs && k2 < t) ? 1u : 0u;      // (unsigned: also false for keys in front of the lane's group)
            }
        }
        acc[w][lane] = ra;
        __syncthreads();
        if (w == 0 && part) {
            ra = 0;
#pragma unroll
            for (int k = 0; k < DMX_RW; k++) ra += acc[k][lane];
            reinterpret_cast<uint32_t *>(blob + t_grp_tokens)[g0 + ra] = t + 2u;
        }
        __syncthreads();                                // (the next chunk reuses acc)
    }
    publish();
}

// ---- one-collective exchange ----
static constexpr uint64_t kXgRedo = 1ull << 63;      // header count, bit 63: this rank's pass 1 has to be repeated (deferred pass 1)
__global__ __launch_bounds__(256) void k_xg_fill(const char *dx_chars, const uint16_t *dx_len, const uint32_t *d_nd, uint32_t stride,
                                                  uint64_t cap_rows, uint32_t slot_bytes, uint8_t *send, const uint32_t *p1c, uint64_t surv_bound,
                                                  uint32_t *h_flag, uint32_t flag_val)
{
    if (h_flag && blockIdx.x == 0 && threadIdx.x == 0) stage_flag_store(h_flag, flag_val);      // (pass 1's kernels in front of this one are complete)
    const uint64_t j = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t nd = *d_nd;
    if (j == 0) {
        uint64_t *h = reinterpret_cast<uint64_t *>(send);
        const bool redo = p1c && (p1c[0] > surv_bound || p1c[0] == 0u || p1c[3] != 0u || (p1c[2] != 0u && p1c[5] != 0u));
        h[0] = nd | (redo ? kXgRedo : 0ull);
        reinterpret_cast<uint32_t *>(send)[2] = stride;
        reinterpret_cast<uint32_t *>(send)[3] = (uint32_t)cap_rows;
    }
    if (j >= nd || j >= cap_rows) return;
    uint8_t *row = send + (j + 1) * (uint64_t)slot_bytes;
    const uint4 *src = reinterpret_cast<const uint4 *>(dx_chars + j * (uint64_t)stride);
    uint4 *dst = reinterpret_cast<uint4 *>(row);
    for (uint32_t i = 0; i < stride / 16; i++) dst[i] = src[i];
    uint4 tail; tail.x = dx_len[j]; tail.y = tail.z = tail.w = 0;
    dst[stride / 16] = tail;
}
hipError_t launch_xg_fill(const char *dx_chars, const uint16_t *dx_len, const uint32_t *d_nd, uint32_t stride, uint64_t cap_rows,
                          uint32_t slot_bytes, uint8_t *send, hipStream_t st, const uint32_t *p1_counts, uint64_t surv_bound, uint32_t *h_flag, uint32_t flag_val)
{
    CRASS_LAUNCH(k_xg_fill, dim3((unsigned)((cap_rows + 255) / 256)), dim3(256), 0, st, dx_chars, dx_len, d_nd, stride, cap_rows, slot_bytes, send,
                 p1_counts, surv_bound, h_flag, flag_val);
    return hipGetLastError();
}
// rank order == global read order: rank r's rows go to [off_r, off_r + n_r)
__global__ __launch_bounds__(256) void k_xg_unpack(const uint8_t *recv, uint32_t world, uint32_t rank, uint32_t stride, uint64_t cap_rows,
                                                    uint32_t slot_bytes, char *g_chars, uint16_t *g_len, uint32_t *xinfo, uint32_t *h_xinfo, uint32_t *zero2,
                                                    unsigned long long *dd_keys, uint32_t *dd_first, uint32_t dd_size)
{
    if (zero2 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 2) zero2[threadIdx.x] = 0u;      // the de-duplication's two counters (no fill launch)
    // the table of the de-duplication that follows is cleared on the way (its own launch was 5-6 us of a rank's step)
    if (dd_keys) {
        const uint64_t nth = (uint64_t)gridDim.x * gridDim.y * blockDim.x;
        for (uint64_t i = ((uint64_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; i < dd_size; i += nth) { dd_keys[i] = 0ull; dd_first[i] = 0xFFFFFFFFu; }
    }
    const uint64_t send_bytes = (cap_rows + 1) * (uint64_t)slot_bytes;
    const uint32_t r = blockIdx.y;
    uint64_t off = 0, total = 0, mx = 0, mine = 0;
    bool redo = false;                                          // some rank's (deferred) pass 1 has to be repeated: nothing is unpacked
    for (uint32_t q = 0; q < world; q++) {                      // a handful of ranks: every thread sums the headers
        uint64_t nq = *reinterpret_cast<const uint64_t *>(recv + q * send_bytes);
        redo |= (nq & kXgRedo) != 0ull;
        nq &= ~kXgRedo;
        if (q < r) off += nq;
        if (q < rank) mine += nq;
        total += nq;
        mx = nq > mx ? nq : mx;
    }
    if (redo && mx <= cap_rows) mx = cap_rows + 1;              // (reported as a list that did not fit; the rows asked for stay within what there is: below)
    const uint64_t n_r = *reinterpret_cast<const uint64_t *>(recv + r * send_bytes) & ~kXgRedo;
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (r == 0 && i == 0) {
        // (a list that did not fit: no row is unpacked, and the kernels queued behind this one see an EMPTY global list —
        // with the sum of the headers they would walk rows of g_chars / g_len that nobody wrote)
        // ([3]: the rows the largest list needs; 1 when the lists fit and only a pass 1 has to be repeated — the caller keeps its capacity)
        const uint32_t need = redo && mx == cap_rows + 1 ? 1u : (uint32_t)mx;
        xinfo[0] = mx > cap_rows ? 0u : (uint32_t)total; xinfo[1] = (uint32_t)mine; xinfo[2] = mx > cap_rows ? 1u : 0u; xinfo[3] = need;
        if (h_xinfo) { h_xinfo[0] = (uint32_t)total; h_xinfo[1] = (uint32_t)mine; h_xinfo[2] = mx > cap_rows ? 1u : 0u; h_xinfo[3] = need; }   // pinned mirror
    }
    if (mx > cap_rows || i >= n_r) return;
    const uint8_t *row = recv + r * send_bytes + (i + 1) * (uint64_t)slot_bytes;
    const uint4 *src = reinterpret_cast<const uint4 *>(row);
    uint4 *dst = reinterpret_cast<uint4 *>(g_chars + (off + i) * (uint64_t)stride);
    for (uint32_t k = 0; k < stride / 16; k++) dst[k] = src[k];
    g_len[off + i] = (uint16_t)src[stride / 16].x;
}
hipError_t launch_xg_unpack(const uint8_t *recv, uint32_t world, uint32_t rank, uint32_t stride, uint64_t cap_rows, uint32_t slot_bytes,
                            char *g_chars, uint16_t *g_len, uint32_t *xinfo, hipStream_t st, uint32_t *h_xinfo, uint32_t *zero2,
                            unsigned long long *dd_keys, uint32_t *dd_first, uint32_t dd_size)
{
    CRASS_LAUNCH(k_xg_unpack, dim3((unsigned)((cap_rows + 255) / 256), world), dim3(256), 0, st, recv, world, rank, stride, cap_rows,
                       slot_bytes, g_chars, g_len, xinfo, h_xinfo, zero2, dd_keys, dd_first, dd_keys ? dd_size : 0u);
    return hipGetLastError();
}

hipError_t launch_device_merge(const DevMerge &M, hipStream_t st, bool init_done, hipStream_t view_st, hipEvent_t ev_fork, hipEvent_t ev_view,
                               hipEvent_t ev_apply)
{
    if (M.n_tok == 0) return hipErrorInvalidValue;
    const unsigned nb = (M.n_tok + 255) / 256;
    if (!init_done) CRASS_LAUNCH(k_dm_init, dim3(1024), dim3(256), 0, st, M);
    CRASS_LAUNCH(k_dm_pack_codes, dim3((M.n_tok + 3) / 4), dim3(256), 0, st, M);       // one wave per token
    // every wave must be resident: at most one block per CU (16 waves of the CU's 32 wave slots, no LDS)
    unsigned gb = (M.n_tok + 15) / 16;
    static const unsigned greedy_per_cu = getenv("CRASS_DM_GREEDY_PER_CU") ? (unsigned)std::max(1, atoi(getenv("CRASS_DM_GREEDY_PER_CU"))) : 1u;      // (experiment)
    if (gb > M.n_cu * greedy_per_cu) gb = M.n_cu * greedy_per_cu;
    CRASS_LAUNCH(k_dm_greedy, dim3(gb), dim3(1024), 0, st, M);
    if (M.x_on && view_st) {
        // The view's device part — tokens, groups, every group's tokens in order — only needs root_of[]: it is assembled beside
        // removeRedundantRepeats and the kernels that build pass 2's index, and is through before pass 2's probe starts (forked
        // behind k_dm_redundant, as until round 6, its last kernels ran beside the probe and the verification: + 40 us for them
        // and a view that was ready 20 us after the step's last kernel).  The pattern list needs blank[] as well: the host puts
        // it together from this view and blank[] (build_host_merge), which k_dm_keys' stage flag announces.
        hipError_t e = hipEventRecord(ev_fork, st);
        if (e == hipSuccess) e = hipStreamWaitEvent(view_st, ev_fork, 0);
        if (e != hipSuccess) return e;
        const unsigned nt = (M.n_tok + 1023) / 1024;
        CRASS_LAUNCH(k_dmx_count, dim3(nt), dim3(DMX_BLOCK), 0, view_st, M);
        CRASS_LAUNCH(k_dmx_tiles, dim3(nt), dim3(1024), 0, view_st, M);
        CRASS_LAUNCH(k_dmx_apply, dim3(nt), dim3(1024), 0, view_st, M);
        if (ev_apply) { e = hipEventRecord(ev_apply, view_st); if (e != hipSuccess) return e; }
        CRASS_LAUNCH(k_dmx_place, dim3(nt), dim3(DMX_BLOCK), 0, view_st, M);
        CRASS_LAUNCH(k_dmx_sort, dim3(std::min<unsigned>((M.n_tok + 64) / 65, std::max(1u, M.n_cu))), dim3(1024), 0, view_st, M);      // (a sorted group has >= 65 members)
        CRASS_LAUNCH(k_dmx_rank, dim3(std::min<unsigned>((M.n_tok + 63) / 64, std::max(1u, M.n_cu))), dim3(64 * DMX_RW), 0, view_st, M);
        e = hipEventRecord(ev_view, view_st);
        if (e != hipSuccess) return e;
    }
    CRASS_LAUNCH(k_dm_rd_bases, dim3(nb), dim3(256), 0, st, M);
    CRASS_LAUNCH(k_dm_rd_fill, dim3(nb), dim3(256), 0, st, M);
    unsigned rb = (M.n_tok + 3) / 4;
    if (rb > 4096) rb = 4096;
    CRASS_LAUNCH(k_dm_redundant, dim3(rb), dim3(256), 0, st, M);
    const unsigned ne = (16u * M.n_tok + 255) / 256;
    CRASS_LAUNCH(k_dm_keys, dim3(ne), dim3(256), 0, st, M);
    CRASS_LAUNCH(k_dm_key_bases_insert, dim3((16u * M.n_tok + 1023) / 1024), dim3(1024), 0, st, M);
    CRASS_LAUNCH(k_dm_fill_finish, dim3(std::max(std::max(128u, ne), (unsigned)(((1ull << M.tab_log_alloc) + 255) / 256))), dim3(256), 0, st, M);
    return hipGetLastError();
}

// ---- pass 2, exact verification of the reads the anchor filter flagged (replaces the automaton scan
// on this path).  Every occurrence of a pattern P at offset o is found through exactly one aligned
// window a = ceil8(o): the key at a is P[a-o .. a-o+16), so the entry (P, r = a-o) is in that key's
// chain.  Wanted: ACISM's first callback = smallest end position, ties -> longest pattern
// (libcrispr.cpp:441, acism.c:73-102).  An occurrence found through window a ends at >= a+16, so the
// scan stops as soon as the best end so far is <= a+15.
// info_by_slot[k] = (end_exclusive << 8) | length, 0 = none; pid_by_slot[k] = pattern index.
static __device__ __forceinline__ uint32_t dm_rd_len(const DevReads &R, uint64_t r) { return R.uniform_len ? R.uniform_len : R.lengths[r]; }
static __device__ __forceinline__ uint64_t dm_rd_off(const DevReads &R, uint64_t r) { return R.stride_words ? r * (uint64_t)R.stride_words : R.word_off[r]; }

// One wave per flagged read.  Lanes 0..h_max probe the read's aligned windows in parallel; the windows that
// hit a key are then taken in ascending order and their candidates compared 64 at a time.  The read's words
// sit in LDS ([wave][word]) so that the dynamically indexed window extraction costs an LDS read.
#define DV_MAXW 20                                   // words staged per read (reads up to 304 bases); longer reads use global loads
#define DV_G 3                                       // reads a wave verifies side by side when their windows fit 21 lanes each
#define DV_GL 21                                     // lanes per read in that form (reads up to 183 bases: h_max <= 20)

// the candidates of one hit window (key entries [base, base + cnt)) against the read whose words `word` returns; keeps the
// wave-uniform best (end, len, pid): ACISM's first callback = smallest end, ties -> longest pattern
template <typename WordFn>
static __device__ __forceinline__ void dv_candidates(const DevMerge &M, WordFn word, const uint8_t *raw, uint32_t L, uint32_t a, uint32_t cnt_s,
                                                     uint32_t base_s, int lane, uint32_t &best_end, uint32_t &best_len, uint32_t &best_pid)
{
    for (uint32_t c0 = 0; c0 < cnt_s; c0 += 64) {
        const uint32_t c = c0 + lane;
        uint32_t cand = 0xFFFFFFFFu, cpid = 0;                // (end << 8) | (255 - len): smaller is better
        if (c < cnt_s) {
            const uint64_t *ent = M.ents + (uint64_t)(base_s + c) * 2;
            const uint64_t *ent2 = M.ents + ((uint64_t)M.ent_cap + base_s + c) * 2;
            const uint64_t e0 = ent[0], e1 = ent[1], f0 = ent2[0], f1 = ent2[1];      // (both planes requested at once)
            const uint32_t rr = (uint32_t)e0 & 7u, len = ((uint32_t)e0 >> 3) & 0x7Fu;
            if (a >= rr && a - rr + len <= L) {
                const uint32_t start = a - rr;
                const uint32_t w0 = start >> 4, sh = (start & 15u) * 2u;
                uint32_t x[5];
#pragma unroll
                for (int q = 0; q < 5; q++) x[q] = word(w0 + q);
                uint32_t y[4];
#pragma unroll
                for (int q = 0; q < 4; q++) y[q] = sh ? ((x[q] >> sh) | (x[q + 1] << (32 - sh))) : x[q];
                const uint64_t v0 = (uint64_t)y[0] | ((uint64_t)y[1] << 32), v1 = (uint64_t)y[2] | ((uint64_t)y[3] << 32);
                uint64_t m0, m1;
                mask128(len, m0, m1);
                bool eq = (v0 & m0) == e1 && (v1 & m1) == f0;
                const uint64_t e_mask = f1;
                if (eq && raw) {
                    uint64_t rm = 0;
                    bool other = false;
                    for (uint32_t i = 0; i < len; i++) {
                        const uint8_t ch = raw[start + i];
                        if (!((ch == 'A') | (ch == 'C') | (ch == 'G') | (ch == 'T'))) { rm |= 1ull << i; other |= ch != 'N'; }
                    }
                    eq = !other && rm == e_mask;
                } else if (eq) eq = e_mask == 0ull;
                if (eq) { cand = ((start + len) << 8) | (255u - len); cpid = (uint32_t)(e0 >> 32); }
            }
        }
        // wave minimum of cand (ties: any lane — equal (end, len) means equal strings)
        uint32_t mn = cand;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, off));
        if (mn != 0xFFFFFFFFu) {
            const uint32_t e_end = mn >> 8, e_len = 255u - (mn & 0xFFu);
            if (e_end < best_end || (e_end == best_end && e_len > best_len)) {
                const uint64_t who = __ballot(cand == mn);
                best_end = e_end; best_len = e_len;
                best_pid = (uint32_t)__shfl((int)cpid, __ffsll((unsigned long long)who) - 1);
            }
        }
    }
}

// the key set's entry range for window value V: (cnt, base), cnt == 0 when V is no key
static __device__ __forceinline__ void dv_probe(const DevMerge &M, uint32_t V, uint32_t kmask, uint32_t &cnt, uint32_t &base)
{
    const unsigned long long want = (unsigned long long)V | (1ull << 32);
    uint32_t s = kset_hash(V, M.kset_log);
    for (;;) {
        // (the slot's range is requested with its key — three loads in flight, one round trip — instead of after the comparison)
        const unsigned long long kk = M.kset_key[s];
        const uint32_t c_s = M.kset_cnt[s], b_s = M.kset_base[s];
        if (kk == 0ull) break;
        if (kk == want) { cnt = c_s; base = b_s; break; }
        s = (s + 1) & kmask;
    }
}

static __device__ __forceinline__ const uint8_t *dv_raw(const DevReads &R, uint64_t r)
{
    if (!(R.n_exc && ((R.exc_mask[r >> 5] >> (r & 31)) & 1u))) return nullptr;
    uint64_t lo = 0, hi = R.n_exc - 1;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (R.exc_read[mid] < r) lo = mid + 1; else hi = mid; }
    return R.exc_bytes + R.exc_off[lo];
}

// SHORT: every read of the set has at most 183 bases (uniform length, the short-read layouts): a wave takes DV_G reads per
// round, DV_GL lanes each — the kernel is latency bound (index -> words -> key set -> entries: four dependent round trips
// per read, 18 of 64 lanes busy in the one-read form), and the rounds of three reads overlap; the candidate comparison stays
// 64 lanes wide, one hit window at a time.  100 M reads (400 k flagged): 440 us in the one-read form.
template <bool SHORT, int U>
__global__ __launch_bounds__(256) void k_dm_verify(DevReads R, DevMerge M, const uint64_t *idx, const uint32_t *d_n, uint64_t n_max,
                                                    uint32_t *info_by_slot, uint32_t *pid_by_slot)
{
    __shared__ uint32_t rw_all[4][DV_G][DV_MAXW + 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t wave = __builtin_amdgcn_readfirstlane((uint32_t)((blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6));
    const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint64_t n = *d_n;
    if (n > n_max) n = n_max;
    const uint32_t kmask = (1u << M.kset_log) - 1u;
    if (SHORT) {
        const int gi = lane / DV_GL, pl = lane % DV_GL;                     // (lane 63: group 3 = idle)
        const uint32_t L = R.uniform_len, nw = (L + 15) >> 4, h_max = (L - 16) >> 3;
        const bool grp = gi < DV_G;
        const uint64_t stride = n_waves * DV_G;
        const int g0 = (grp ? gi : 0) * DV_GL;                              // first lane of this lane's group
        const uint64_t gmask = grp ? (((1ull << DV_GL) - 1ull) << g0) : 0ull;
        uint32_t *rw = rw_all[wv][grp ? gi : 0];
        const ulonglong2 *ents1 = reinterpret_cast<const ulonglong2 *>(M.ents), *ents2 = ents1 + M.ent_cap;
        // The round's two leading round trips (index -> packed words) are taken one round ahead: while this round's reads are
        // probed and compared, the next round's words and the index of the round after are already in flight (software
        // pipeline in registers).
        uint64_t k = wave * DV_G + (uint64_t)gi;
        bool have = grp && k < n;
        uint64_t r = have ? idx[k] : 0;
        uint32_t wreg = (have && (uint32_t)pl < nw) ? R.packed[dm_rd_off(R, r) + pl] : 0u;
        uint64_t k_n = k + stride;
        bool have_n = grp && k_n < n;
        uint64_t r_n = have_n ? idx[k_n] : 0;
        for (uint64_t k0 = wave * DV_G; k0 < n; k0 += stride) {
            if (have && (uint32_t)pl <= nw + 4u) rw[pl] = wreg;             // (words nw .. nw + 4 = 0 — wreg is 0 there —: a window's five words need no clamp)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t wreg_n = (have_n && (uint32_t)pl < nw) ? R.packed[dm_rd_off(R, r_n) + pl] : 0u;
            const uint64_t k_nn = k_n + stride;
            const bool have_nn = grp && k_nn < n;
            const uint64_t r_nn = have_nn ? idx[k_nn] : 0;
            uint32_t cnt = 0, base = 0;
            if (have && (uint32_t)pl <= h_max) {
                const uint32_t wi = (uint32_t)pl >> 1;
                const uint32_t lo = rw[wi], hi = rw[wi + 1];
                dv_probe(M, (pl & 1) ? ((lo >> 16) | (hi << 16)) : lo, kmask, cnt, base);
            }
            // The three reads of the round walk their hit windows SIDE BY SIDE: read g's 21 lanes compare the candidates of
            // read g's next hit window (42 at a time, two per lane, both planes of both entries requested at once) while the
            // other two groups do the same for theirs — as many dependent {entry load, compare, reduce} steps per round as the
            // busiest read has windows (~4), not as the three have together (~12, a quarter of the lanes busy).
            // be / bl / bp: the group's best so far, the same value in each of its lanes.
            const uint8_t *raw = nullptr;                                    // exception reads (rare): the read's raw bytes
            if (R.n_exc && have) raw = dv_raw(R, r);
            const uint64_t hits_all = __ballot(cnt > 0);
            uint32_t my = grp ? (uint32_t)(hits_all >> (gi * DV_GL)) & ((1u << DV_GL) - 1u) : 0u;      // this read's hit windows
            uint32_t be = 0xFFFFFFFFu, bl = 0, bp = 0;
            while (__ballot(my != 0u)) {
                bool act = my != 0u;
                const int wsrc = act ? __ffs((int)my) - 1 : 0;               // the group's next hit window
                if (act) my &= my - 1u;
                const uint32_t a = 8u * (uint32_t)wsrc;
                if (act && be <= a + 15u) { act = false; my = 0u; }          // no later window of this read can end earlier
                const uint32_t cnt_s = (uint32_t)__shfl((int)cnt, g0 + wsrc), base_s = (uint32_t)__shfl((int)base, g0 + wsrc);
                for (uint32_t c0 = 0; __ballot(act && c0 < cnt_s); c0 += U * DV_GL) {
                    uint32_t cand = 0xFFFFFFFFu, cpid = 0;                   // (end << 8) | (255 - len): smaller is better
                    uint32_t cq[U];
                    bool vq[U];
                    ulonglong2 e1[U], e2[U];
#pragma unroll
                    for (int q = 0; q < U; q++) {                            // (entry 0 stands in for a lane without a candidate)
                        cq[q] = c0 + (uint32_t)(q * DV_GL + pl);
                        vq[q] = act && cq[q] < cnt_s;
                        const uint32_t ei = vq[q] ? base_s + cq[q] : 0u;
                        e1[q] = ents1[ei]; e2[q] = ents2[ei];
                    }
#pragma unroll
                    for (int q = 0; q < U; q++) {
                        const uint64_t e0 = e1[q].x;
                        const uint32_t rr = (uint32_t)e0 & 7u, len = ((uint32_t)e0 >> 3) & 0x7Fu;
                        if (vq[q] && a >= rr && a - rr + len <= L) {
                            const uint32_t start = a - rr;
                            const uint32_t w0 = start >> 4, sh = (start & 15u) * 2u;
                            uint32_t x[5];
#pragma unroll
                            for (int i = 0; i < 5; i++) x[i] = rw[w0 + i];
                            uint32_t y[4];
#pragma unroll
                            for (int i = 0; i < 4; i++) y[i] = __builtin_amdgcn_alignbit(x[i + 1], x[i], sh);
                            const uint64_t v0 = (uint64_t)y[0] | ((uint64_t)y[1] << 32), v1 = (uint64_t)y[2] | ((uint64_t)y[3] << 32);
                            uint64_t m0, m1;
                            mask128(len, m0, m1);
                            bool eq = (v0 & m0) == e1[q].y && (v1 & m1) == e2[q].x;
                            const uint64_t e_mask = e2[q].y;
                            if (eq && raw) {
                                uint64_t rm = 0;
                                bool other = false;
                                for (uint32_t i = 0; i < len; i++) {
                                    const uint8_t ch = raw[start + i];
                                    if (!((ch == 'A') | (ch == 'C') | (ch == 'G') | (ch == 'T'))) { rm |= 1ull << i; other |= ch != 'N'; }
                                }
                                eq = !other && rm == e_mask;
                            } else if (eq) eq = e_mask == 0ull;
                            if (eq) {
                                const uint32_t cv = ((start + len) << 8) | (255u - len);
                                if (cv < cand) { cand = cv; cpid = (uint32_t)(e0 >> 32); }
                            }
                        }
                    }
                    // minimum over the group's lanes (ties: any lane — equal (end, len) means equal strings) — only when one of
                    // them matched at all: nearly every step compares a window's variants and finds none equal, and the
                    // reduction is a quarter of a step's instructions (the kernel is VALU-issue bound: 109 M wave instructions =
                    // 178 of its 246 us at 100 M reads, PMC round 4)
                    if (__ballot(cand != 0xFFFFFFFFu) & gmask) {
                        uint32_t mn = cand;
#pragma unroll
                        for (int off = 16; off > 0; off >>= 1) {
                            const uint32_t o = (uint32_t)__shfl_down((int)mn, off);
                            if (pl + off < DV_GL) mn = min(mn, o);
                        }
                        mn = (uint32_t)__shfl((int)mn, g0);
                        const uint64_t who = __ballot(cand == mn) & gmask;
                        const uint32_t pid = (uint32_t)__shfl((int)cpid, who ? __ffsll((unsigned long long)who) - 1 : 0);
                        const uint32_t e_end = mn >> 8, e_len = 255u - (mn & 0xFFu);
                        if (e_end < be || (e_end == be && e_len > bl)) { be = e_end; bl = e_len; bp = pid; }
                    }
                }
            }
            if (have && pl == 0) {
                info_by_slot[k] = bl ? ((be << 8) | bl) : 0u;
                pid_by_slot[k] = bp;
            }
            __builtin_amdgcn_wave_barrier();                // the next round reuses rw
            k = k_n; have = have_n; r = r_n; wreg = wreg_n;
            k_n = k_nn; have_n = have_nn; r_n = r_nn;
        }
        return;
    }
    uint32_t *rw = rw_all[wv][0];                       // (DV_G rows of DV_MAXW + 2 words, contiguous: room for DV_MAXW + 1)
    for (uint64_t k = wave; k < n; k += n_waves) {
        const uint64_t r = idx[k];
        const uint32_t L = dm_rd_len(R, r);
        const uint32_t *g = R.packed + dm_rd_off(R, r);
        const uint32_t nw = (L + 15) >> 4;
        const bool staged = nw <= DV_MAXW;
        // exception read (non-ACGT bytes pack as 'A'): a bit-equal candidate only counts if its non-ACGT bytes are
        // exactly the pattern's 'N' positions and are 'N' themselves — the automaton the reference runs over the
        // bytes (libcrispr.cpp:503) matches byte for byte
        const uint8_t *raw = dv_raw(R, r);               // wave-uniform
        if (staged) {
            if ((uint32_t)lane < nw) rw[lane] = g[lane];
            if ((uint32_t)lane == nw) rw[lane] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        auto word = [&](uint32_t i) -> uint32_t {
            if (staged) return rw[min(i, nw)];
            return i < nw ? g[i] : 0u;
        };
        uint32_t best_end = 0xFFFFFFFFu, best_len = 0, best_pid = 0;      // wave-uniform
        if (L >= 16) {
            const uint32_t ash = M.akey_shift;                              // windows every 8 (or, patterns of 19 .. 22 bases, 4) bases
            const uint32_t h_max = (L - 16) >> ash;
            for (uint32_t hb = 0; hb <= h_max; hb += 64) {
                if (best_end <= (hb << ash) + 15) break;
                const uint32_t h = hb + lane;
                uint32_t cnt = 0, base = 0;
                if (h <= h_max) {
                    const uint32_t pos = h << ash, wi = pos >> 4;
                    const uint32_t lo = word(wi), hi = word(wi + 1);
                    dv_probe(M, __builtin_amdgcn_alignbit(hi, lo, (pos & 15u) * 2u), kmask, cnt, base);
                }
                uint64_t hits = __ballot(cnt > 0);
                while (hits) {
                    const int src = __ffsll((unsigned long long)hits) - 1;
                    hits &= hits - 1;
                    const uint32_t a = (hb + (uint32_t)src) << ash;
                    if (best_end <= a + 15) { hits = 0; break; }              // later windows cannot end earlier
                    const uint32_t cnt_s = (uint32_t)__shfl((int)cnt, src), base_s = (uint32_t)__shfl((int)base, src);
                    dv_candidates(M, word, raw, L, a, cnt_s, base_s, lane, best_end, best_len, best_pid);
                }
            }
        }
        if (lane == 0) {
            info_by_slot[k] = best_len ? ((best_end << 8) | best_len) : 0u;
            pid_by_slot[k] = best_pid;
        }
        __builtin_amdgcn_wave_barrier();                // the next read reuses rw
    }
}

// The runtime loads a code object on the first launch of one of its kernels (this file's: ~15 ms, which a group's first step
// paid inside its first merge — 18.4 ms against 4.9 in steady state).  Asking for a kernel's attributes loads it.
hipError_t warm_dmerge_module()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, reinterpret_cast<const void *>(&k_xg_unpack));
}

hipError_t launch_dm_verify(const DevReads &R, const DevMerge &M, const uint64_t *idx, const uint32_t *d_n, uint64_t n_max,
                            uint32_t *info_by_slot, uint32_t *pid_by_slot, hipStream_t st)
{
    if (n_max == 0) return hipSuccess;
    static const bool dv_one = getenv("CRASS_DV_ONE") != nullptr;      // A/B switch, read once per process
    // every read at most 183 bases (uniform length): three reads per wave and round
    const bool shortr = M.akey_shift == 3 && R.uniform_len >= 16 && ((R.uniform_len - 16) >> 3) < DV_GL && ((R.uniform_len + 15) >> 4) <= DV_MAXW && !dv_one;
    uint64_t nb = shortr ? (n_max + 4 * DV_G - 1) / (4 * DV_G) : (n_max + 3) / 4;
    if (nb > 8192) nb = 8192;
    static const bool dv_pair = getenv("CRASS_DV_SINGLE") == nullptr;  // two candidates per lane and step (CRASS_DV_SINGLE: the A/B switch, one)
    if (shortr && dv_pair) CRASS_LAUNCH((k_dm_verify<true, 2>), dim3((unsigned)nb), dim3(256), 0, st, R, M, idx, d_n, n_max, info_by_slot, pid_by_slot);
    else if (shortr) CRASS_LAUNCH((k_dm_verify<true, 1>), dim3((unsigned)nb), dim3(256), 0, st, R, M, idx, d_n, n_max, info_by_slot, pid_by_slot);
    else CRASS_LAUNCH((k_dm_verify<false, 1>), dim3((unsigned)nb), dim3(256), 0, st, R, M, idx, d_n, n_max, info_by_slot, pid_by_slot);
    return hipGetLastError();
}

} // namespace crass
