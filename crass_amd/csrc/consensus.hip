// consensus.hip — device kernels of the stage right behind the search hot path:
// WorkHorse::findConsensusDRs (src/crass/WorkHorse.cpp:578-611) = per group, Aligner (src/crass/Aligner.cpp) +
// calculateDRConsensus / splitGroupedDR, then ReadHolder::updateStartStops (src/crass/ReadHolder.cpp:382-511) for every
// read of the group.  The control flow (which is sequential by construction: greedy GID / token numbering, recursion
// into split groups) runs on the host (consensus.cpp); the data-parallel work is here:
//   k_cons_flip    ReadHolder::reverseComplementSeq of a list of records (Aligner::alignSlave's reversed slaves)
//   k_cons_cover   Aligner::placeReadsInCoverageArray: every placed read adds its bases to the coverage columns
//   k_cons_ksw     Aligner::getOffsetAgainstMaster: ksw_align (src/crass/ksw.c:228-360) of every DR variant of a group and
//                  of its reverse complement against the group's master DR
//   k_cons_sw      smithWaterman (src/crass/SmithWaterman.cpp:151-308): the partial-repeat search at both read ends, one
//                  wave per (read end, true DR); its Levenshtein filter (:283) runs through the engine's batch kernel
// gfx950, wave64, integer / fp64 add-compare work: no MFMA.  Records are RH_Seq bytes (raw ASCII, so N / IUPAC bytes keep
// the reference's byte semantics) in one device buffer: record k at roff[k], length rlen[k].
#include "engine_internal.h"
#include "consensus_internal.h"

namespace crass {

#define WAVE 64

// ---- ReadHolder::reverseComplementSeq (ReadHolder.cpp:593-609; comp_tab of SeqUtils.cpp:50-59 passed in) ----
__global__ __launch_bounds__(256) void k_cons_flip(uint8_t *seq, const uint64_t *roff, const uint32_t *rlen, const uint32_t *list, uint32_t n,
                                                    const uint8_t *comp)
{
    const int lane = threadIdx.x & 63;
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t q = w; q < n; q += nw) {
        const uint32_t k = list[q];
        uint8_t *s = seq + roff[k];
        const uint32_t L = rlen[k];
        for (uint32_t i = lane; i < (L + 1) / 2; i += WAVE) {
            const uint32_t j = L - 1 - i;
            const uint8_t a = s[i], b = s[j];
            s[i] = comp[b & 127]; s[j] = comp[a & 127];         // (i == j: the middle base, complemented once)
        }
    }
}

// ---- Aligner::placeReadsInCoverageArray (Aligner.cpp:364-418): coverageIndex(i, c) = (CHAR_TO_INDEX[c] - 1) * length + i,
// any byte that is not C / G / T counts as A (Aligner.cpp:61-70, Aligner.h:48) ----
__global__ __launch_bounds__(256) void k_cons_cover(const uint8_t *seq, const uint64_t *roff, const uint32_t *rlen, const uint32_t *plc_rec,
                                                     const int32_t *plc_pos, uint32_t n_plc, int *cov, int length)
{
    const int lane = threadIdx.x & 63;
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t q = w; q < n_plc; q += nw) {
        const uint32_t k = plc_rec[q];
        const uint8_t *s = seq + roff[k];
        const int L = (int)rlen[k], pos = plc_pos[q];
        for (int i = lane; i < L; i += WAVE) {
            const uint8_t c = s[i];
            const int row = (c == 'C' || c == 'c') ? 1 : (c == 'G' || c == 'g') ? 2 : (c == 'T' || c == 't') ? 3 : 0;
            atomicAdd(&cov[(size_t)row * length + (i + pos)], 1);        // (the host checked 0 <= i + pos < length)
        }
    }
}

__global__ __launch_bounds__(256) void k_cons_cover_lds(const uint8_t *seq, const uint64_t *roff, const uint32_t *rlen, const uint32_t *plc_rec,
                                                         const int32_t *plc_pos, uint32_t n_plc, int *cov, int length)
{
    extern __shared__ int cov_lds[];                    // [4][length]
    for (int i = threadIdx.x; i < 4 * length; i += 256) cov_lds[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t q = w; q < n_plc; q += nw) {
        const uint32_t k = plc_rec[q];
        const uint8_t *s = seq + roff[k];
        const int L = (int)rlen[k], pos = plc_pos[q];
        for (int i = lane; i < L; i += WAVE) {
            const uint8_t c = s[i];
            const int row = (c == 'C' || c == 'c') ? 1 : (c == 'G' || c == 'g') ? 2 : (c == 'T' || c == 't') ? 3 : 0;
            atomicAdd(&cov_lds[row * length + (i + pos)], 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * length; i += 256) { const int v = cov_lds[i]; if (v) atomicAdd(&cov[i], v); }
}

// ---- ksw_i16 (ksw.c:228-317) with the SSE2 striped layout simulated lane by lane: 8 int16 lanes per vector, slen =
// ceil(qlen / 8) vectors, query position of (vector j, lane l) = j + l * slen.  The layout is part of the result:
// E(i+1, j) is taken from H before the lazy-F pass, and the query end is the first maximum in vector-memory order.
// One thread per alignment; its H0 / H1 / E / Hmax vectors live in LDS, [entry][thread].
struct KswIO { int score, te, qe; };
static __device__ __forceinline__ int16_t k_adds(int16_t a, int16_t b) { int v = (int)a + (int)b; return (int16_t)(v > 32767 ? 32767 : v < -32768 ? -32768 : v); }
static __device__ __forceinline__ int16_t k_subsu(int16_t a, int16_t b) { const uint16_t x = (uint16_t)a, y = (uint16_t)b; return (int16_t)(x > y ? x - y : 0); }
static __device__ __forceinline__ int16_t k_max(int16_t a, int16_t b) { return a > b ? a : b; }

// QF(k) / TF(i): query / target codes (0..4); mat: the 5 x 5 scoring matrix of the Aligner
template <typename QF, typename TF>
static __device__ KswIO ksw_i16_dev(QF qf, int qlen, TF tf, int tlen, int gapo, int gape, int endsc, const int8_t *mat, int16_t *lds, int nvec_max)
{
    const int T = blockDim.x, me = threadIdx.x;
    const int slen = (qlen + 7) / 8, nv = slen * 8;
    int16_t *A0 = lds + me, *A1 = lds + (size_t)nvec_max * T + me, *E = lds + (size_t)2 * nvec_max * T + me, *HM = lds + (size_t)3 * nvec_max * T + me;
    for (int x = 0; x < nv; x++) { A0[(size_t)x * T] = 0; E[(size_t)x * T] = 0; HM[(size_t)x * T] = 0; }
    int16_t *H0 = A0, *H1 = A1;
    const int16_t gapoe = (int16_t)(gapo + gape), gpe = (int16_t)gape;
    int te = -1, gmax = 0;
    for (int i = 0; i < tlen; ++i) {
        int16_t f[8], mx[8], h[8];
        const int8_t *mrow = mat + tf(i) * 5;
#pragma unroll
        for (int l = 0; l < 8; l++) { f[l] = 0; mx[l] = 0; }
        h[0] = 0;
#pragma unroll
        for (int l = 1; l < 8; l++) h[l] = slen ? H0[(size_t)((slen - 1) * 8 + l - 1) * T] : (int16_t)0;
        for (int j = 0; j < slen; ++j) {
#pragma unroll
            for (int l = 0; l < 8; l++) {
                const int k = j + l * slen;
                const int16_t S = (int16_t)(k >= qlen ? 0 : mrow[qf(k)]);
                int16_t hh = k_adds(h[l], S);
                int16_t e = E[(size_t)(j * 8 + l) * T];
                hh = k_max(hh, e); hh = k_max(hh, f[l]);
                mx[l] = k_max(mx[l], hh);
                H1[(size_t)(j * 8 + l) * T] = hh;
                hh = k_subsu(hh, gapoe);
                e = k_subsu(e, gpe); e = k_max(e, hh);
                E[(size_t)(j * 8 + l) * T] = e;
                f[l] = k_subsu(f[l], gpe); f[l] = k_max(f[l], hh);
                h[l] = H0[(size_t)(j * 8 + l) * T];
            }
        }
        bool done = false;
        for (int k = 0; k < 16 && !done; ++k) {             // the lazy-F loop
#pragma unroll
            for (int l = 7; l > 0; l--) f[l] = f[l - 1];
            f[0] = 0;
            for (int j = 0; j < slen; ++j) {
                bool any = false;
#pragma unroll
                for (int l = 0; l < 8; l++) {
                    int16_t hh = H1[(size_t)(j * 8 + l) * T];
                    hh = k_max(hh, f[l]);
                    H1[(size_t)(j * 8 + l) * T] = hh;
                    hh = k_subsu(hh, gapoe);
                    f[l] = k_subsu(f[l], gpe);
                    any |= f[l] > hh;
                }
                if (!any) { done = true; break; }
            }
        }
        int imax = mx[0];
#pragma unroll
        for (int l = 1; l < 8; l++) imax = mx[l] > imax ? mx[l] : imax;
        if (imax > gmax) {
            gmax = imax; te = i;
            for (int x = 0; x < nv; x++) HM[(size_t)x * T] = H1[(size_t)x * T];
            if (gmax >= endsc) break;
        }
        int16_t *t2 = H1; H1 = H0; H0 = t2;
    }
    KswIO r; r.score = gmax; r.te = te; r.qe = -1;
    int mxv = -1;
    for (int x = 0; x < nv; x++) { const int v = (uint16_t)HM[(size_t)x * T]; if (v > mxv) { mxv = v; r.qe = x / 8 + x % 8 * slen; } }
    return r;
}

// ksw_align (ksw.c:330-360) with xtra = KSW_XSTART | KSW_XSUBO | minsc: the second pass over the reversed prefixes finds the
// start positions.  Task a = 2 * v + o: string v of the batch, o = 1: its reverse complement (codes 3 - c reversed; 4 stays 4)
__global__ __launch_bounds__(64) void k_cons_ksw(const uint8_t *q_codes, const uint32_t *q_off, const uint32_t *q_len, const uint32_t *q_tgt, uint32_t n_str,
                                                  const uint8_t *t_codes, const uint32_t *t_off, const uint32_t *t_len, ConsKswParams P, int nvec_max,
                                                  int32_t *out /* [2 n][3] score, tb, qb */)
{
    extern __shared__ __attribute__((aligned(16))) int16_t ksw_lds[];
    const uint32_t a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= 2 * n_str) return;
    const uint32_t v = a >> 1, o = a & 1u;
    const uint8_t *q = q_codes + q_off[v];
    const int qlen = (int)q_len[v];
    const uint8_t *target = t_codes + t_off[q_tgt[v]];          // the master DR of the string's group
    const int tlen = (int)t_len[q_tgt[v]];
    auto qf = [&](int k) -> int { if (!o) return q[k]; const int c = q[qlen - 1 - k]; return c < 4 ? 3 - c : 4; };
    auto tf = [&](int i) -> int { return target[i]; };
    KswIO r = ksw_i16_dev(qf, qlen, tf, tlen, P.gapo, P.gape, 0x10000, P.mat, ksw_lds, nvec_max);
    int tb = -1, qb = -1;
    if (r.score >= P.minsc) {
        const int qe = r.qe, te = r.te;
        auto qr = [&](int k) -> int { return qf(qe - k); };                               // revseq(qe + 1, query)
        auto tr = [&](int i) -> int { return i <= te ? target[te - i] : target[i]; };     // revseq(te + 1, target)
        const KswIO rr = ksw_i16_dev(qr, qe + 1, tr, tlen, P.gapo, P.gape, r.score, P.mat, ksw_lds, nvec_max);
        if (r.score == rr.score) { tb = r.te - rr.te; qb = r.qe - rr.qe; }
    }
    out[(size_t)a * 3] = r.score; out[(size_t)a * 3 + 1] = tb; out[(size_t)a * 3 + 2] = qb;
}

// ---- smithWaterman (SmithWaterman.cpp:151-308): seqA window [aStartSearch, aStartSearch + aSearchLen) of a record against
// the whole true DR; match 1.2, mismatch -1, gap -1 in double precision, findMax's comparison order (:68-129), the first
// maximum in row-major order, the reference's traceback (which also takes in the zero cell in front of the alignment) and
// its substring arithmetic (a_ret's length includes aStartSearch: :279).  One wave per task, anti-diagonal wavefront: lane =
// DR column, so a DR of up to 64 bases; longer DRs (only with -D > 64) take the serial form in lane 0.
static __device__ __forceinline__ double sw_find_max(double a, double b, double c, double d, int &index)
{
    if (b > a) {
        if (c > d) { if (c > b) { index = 2; return c; } index = 1; return b; }
        if (d > b) { index = 3; return d; }
        index = 1; return b;
    }
    if (c > d) { if (c > a) { index = 2; return c; } index = 0; return a; }
    if (d > a) { index = 3; return d; }
    index = 0; return a;
}

__global__ __launch_bounds__(256) void k_cons_sw(const uint8_t *seq, const uint64_t *roff, const uint32_t *rlen, const ConsSwTask *tasks, uint32_t n_tasks,
                                                  const uint8_t *dr_chars, const uint32_t *dr_off, const uint32_t *dr_len, uint8_t *dirs,
                                                  ConsSwOut *out)
{
    // A task whose direction matrix fits kSwLdsDir bytes (every task of a short-read job: <= ~100 x 38 cells) keeps it — and its
    // window of the read — in LDS: the traceback is one DEPENDENT byte load per step in lane 0 (each a ~1 us round trip from
    // global memory: it was most of this kernel's 5.1 ms for 108 k tasks), and the fill reads one byte of the read per step.
    // Larger tasks (long reads) use the global scratch as before.
    __shared__ uint8_t sw_dir[4][kSwLdsDir];
    __shared__ uint8_t sw_a[4][kSwLdsA];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t q = w; q < n_tasks; q += nw) {
        const ConsSwTask t = tasks[q];
        const uint8_t *A = seq + roff[t.rec];
        const int lenA = (int)rlen[t.rec];
        const uint8_t *B = dr_chars + dr_off[t.dr];
        const int lenB = (int)dr_len[t.dr];
        const int S0 = t.start, SL = t.len, W = lenB + 1;
        const bool in_lds = cons_sw_in_lds(SL, lenB);                   // (the host sized the scratch with the same test)
        uint8_t *dir = in_lds ? &sw_dir[wv][0] : dirs + t.dir_off;
        const uint8_t *Aseg = A + S0;                                    // Aseg[i - 1] = the read's base of row i
        if (in_lds) {
            for (int x = lane; x < SL; x += WAVE) sw_a[wv][x] = A[S0 + x];
            Aseg = &sw_a[wv][0];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        double best = -1.0; int bi = 0, bj = 0;
        if (lenB <= WAVE) {
            const int j = lane + 1;
            const bool col = lane < lenB;
            const uint8_t bch = col ? B[lane] : 0;
            double v1 = 0.0, v2 = 0.0;                                  // this lane's cells of the previous two steps
            double my_best = -1.0; int my_i = 0;
            const int steps = SL + lenB - 1;
            for (int s = 0; s < steps; s++) {
                const int i = s - lane + 1;
                double left = __shfl_up(v1, 1), diag = __shfl_up(v2, 1);
                if (lane == 0) { left = 0.0; diag = 0.0; }
                const bool act = col && i >= 1 && i <= SL;
                double cell = 0.0;
                if (act) {
                    const double up = (i == 1) ? 0.0 : v1;
                    if (i == 1) diag = 0.0;
                    int index;
                    const double sim = (Aseg[i - 1] == bch) ? 1.2 : -1.0;
                    cell = sw_find_max(diag + sim, up + (-1.0), left + (-1.0), 0.0, index);
                    dir[(size_t)i * W + j] = (uint8_t)index;
                    if (cell > my_best) { my_best = cell; my_i = i; }
                }
                // a lane that is not active yet / any more contributes the boundary value 0 only while i < 1; past the last
                // row nobody reads it
                v2 = v1;
                v1 = act ? cell : (i < 1 ? 0.0 : v1);
            }
            // the first maximum in row-major order: largest value, then smallest i, then smallest j
            double bv = my_best; int ri = my_i, rj = j;
            if (!col) { bv = -2.0; ri = 0x7fffffff; rj = 0x7fffffff; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_xor(bv, off); const int oi = __shfl_xor(ri, off), oj = __shfl_xor(rj, off);
                if (ov > bv || (ov == bv && (oi < ri || (oi == ri && oj < rj)))) { bv = ov; ri = oi; rj = oj; }
            }
            best = bv; bi = ri; bj = rj;
            if (best < 0.0) { bi = 0; bj = 0; }                          // (nothing filled)
        } else if (lane == 0) {
            // serial form with two rolling rows
            double *rows = reinterpret_cast<double *>(dirs + t.dir_off + (size_t)(SL + 1) * W + 8 - ((t.dir_off + (size_t)(SL + 1) * W) & 7));
            double *prev = rows, *cur = rows + W;
            for (int jj = 0; jj <= lenB; jj++) prev[jj] = 0.0;
            for (int i = 1; i <= SL; i++) {
                cur[0] = 0.0;
                for (int jj = 1; jj <= lenB; jj++) {
                    int index;
                    const double sim = (A[i - 1 + S0] == B[jj - 1]) ? 1.2 : -1.0;
                    const double cell = sw_find_max(prev[jj - 1] + sim, prev[jj] + (-1.0), cur[jj - 1] + (-1.0), 0.0, index);
                    cur[jj] = cell;
                    dir[(size_t)i * W + jj] = (uint8_t)index;
                    if (cell > best) { best = cell; bi = i; bj = jj; }
                }
                double *x = prev; prev = cur; cur = x;
            }
        }
        if (lenB > WAVE) { bi = __shfl(bi, 0); bj = __shfl(bj, 0); }
        __threadfence();
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            ConsSwOut o; o.err = 0;
            int ci = bi, cj = bj;
            auto nxt = [&](int i, int jj, int &ni, int &nj) {
                const int d = (i >= 1 && jj >= 1) ? dir[(size_t)i * W + jj] : 3;
                ni = (d == 0 || d == 1) ? i - 1 : i; nj = (d == 0 || d == 2) ? jj - 1 : jj;
            };
            int ni, nj;
            nxt(ci, cj, ni, nj);
            while ((nj != 0) && (ni != 0) && ((ci != ni) || (cj != nj))) { ci = ni; cj = nj; nxt(ci, cj, ni, nj); }
            ci--; cj--;
            if (0 > cj) cj = 0;
            if (0 > ci) ci = 0;
            o.a_start = ci + S0;
            o.a_end = o.a_start + bi - ci - 1;
            int apos = ci + S0, an = bi - ci + S0;                        // (sic: substr(current_i + aStartSearch, i_max - current_i + aStartSearch))
            if (apos > lenA) o.err = 1;
            else if (an < 0 || apos + an > lenA) an = lenA - apos;
            int bpos = cj, bn = bj - cj;
            if (bpos > lenB) o.err = 1;
            else if (bn < 0 || bpos + bn > lenB) bn = lenB - bpos;
            o.a_off = apos; o.a_len = an; o.b_off = bpos; o.b_len = bn;
            out[q] = o;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- launch wrappers ----
hipError_t launch_cons_flip(uint8_t *seq, const uint64_t *roff, const uint32_t *rlen, const uint32_t *list, uint32_t n, const uint8_t *comp, hipStream_t st)
{
    if (!n) return hipSuccess;
    CRASS_LAUNCH(k_cons_flip, dim3(std::min<uint32_t>((n + 3) / 4, 4096u)), dim3(256), 0, st, seq, roff, rlen, list, n, comp);
    return hipGetLastError();
}
hipError_t launch_cons_cover(const uint8_t *seq, const uint64_t *roff, const uint32_t *rlen, const uint32_t *plc_rec, const int32_t *plc_pos,
                             uint32_t n_plc, int *cov, int length, hipStream_t st)
{
    if (!n_plc) return hipSuccess;
    // short reads: the whole coverage array (4 rows x length columns) fits LDS, a block adds its placements there and
    // flushes the columns it touched — all reads of a group pile onto the same ~150 columns, and as global atomics
    // (~600 adds per address) that was 54 us per group
    const size_t lds = (size_t)length * 16;
    if (lds <= 64 * 1024) {
        const uint32_t nb = std::max<uint32_t>(1u, std::min<uint32_t>((n_plc + 63) / 64, 64u));
        CRASS_LAUNCH(k_cons_cover_lds, dim3(nb), dim3(256), lds, st, seq, roff, rlen, plc_rec, plc_pos, n_plc, cov, length);
        return hipGetLastError();
    }
    CRASS_LAUNCH(k_cons_cover, dim3(std::min<uint32_t>((n_plc + 3) / 4, 8192u)), dim3(256), 0, st, seq, roff, rlen, plc_rec, plc_pos, n_plc, cov, length);
    return hipGetLastError();
}
hipError_t launch_cons_ksw(const uint8_t *q_codes, const uint32_t *q_off, const uint32_t *q_len, const uint32_t *q_tgt, uint32_t n_str, uint32_t max_qlen,
                           const uint8_t *t_codes, const uint32_t *t_off, const uint32_t *t_len, const ConsKswParams &P, int32_t *out, hipStream_t st)
{
    if (!n_str) return hipSuccess;
    const int nvec_max = (int)((max_qlen + 7) / 8) * 8;
    const size_t lds = (size_t)4 * nvec_max * 64 * sizeof(int16_t);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_cons_ksw), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    CRASS_LAUNCH(k_cons_ksw, dim3((2 * n_str + 63) / 64), dim3(64), lds, st, q_codes, q_off, q_len, q_tgt, n_str, t_codes, t_off, t_len, P, nvec_max, out);
    return hipGetLastError();
}
hipError_t launch_cons_sw(const uint8_t *seq, const uint64_t *roff, const uint32_t *rlen, const ConsSwTask *tasks, uint32_t n_tasks,
                          const uint8_t *dr_chars, const uint32_t *dr_off, const uint32_t *dr_len, uint8_t *dirs, ConsSwOut *out, hipStream_t st)
{
    if (!n_tasks) return hipSuccess;
    CRASS_LAUNCH(k_cons_sw, dim3(std::min<uint32_t>((n_tasks + 3) / 4, 16384u)), dim3(256), 0, st, seq, roff, rlen, tasks, n_tasks, dr_chars, dr_off,
                       dr_len, dirs, out);
    return hipGetLastError();
}

} // namespace crass
