// consensus_internal.h — types shared by consensus.hip (kernels) and consensus.cpp (host stage).  Not part of the ABI.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace crass {

struct ConsKswParams { int gapo, gape, minsc; int8_t mat[25]; };          // Aligner's ksw setup (Aligner.h:112-136)
struct ConsSwTask { uint32_t rec, dr; int32_t start, len; uint64_t dir_off; };   // smithWaterman(RH_Seq, DR, .., start, len, 0.85)
struct ConsSwOut { int32_t a_start, a_end, a_off, a_len, b_off, b_len, err; };

hipError_t launch_cons_flip(uint8_t *seq, const uint64_t *roff, const uint32_t *rlen, const uint32_t *list, uint32_t n, const uint8_t *comp, hipStream_t st);
hipError_t launch_cons_cover(const uint8_t *seq, const uint64_t *roff, const uint32_t *rlen, const uint32_t *plc_rec, const int32_t *plc_pos,
                             uint32_t n_plc, int *cov, int length, hipStream_t st);
// strings as codes 0..4 (seq_nt4_table, Aligner.cpp:40-58), string v against target q_tgt[v] (its group's master DR);
// out[2 v + o][3] = score, tb, qb of string v (o = 1: its reverse complement)
hipError_t launch_cons_ksw(const uint8_t *q_codes, const uint32_t *q_off, const uint32_t *q_len, const uint32_t *q_tgt, uint32_t n_str, uint32_t max_qlen,
                           const uint8_t *t_codes, const uint32_t *t_off, const uint32_t *t_len, const ConsKswParams &P, int32_t *out, hipStream_t st);
// A task whose direction matrix and read window fit the kernel's per-wave LDS keeps them there (k_cons_sw) and needs no scratch
static constexpr uint32_t kSwLdsDir = 8192, kSwLdsA = 256;      // per wave: direction bytes, read window
__host__ __device__ static inline bool cons_sw_in_lds(int len, int dr_len)
{
    return dr_len <= 64 && len >= 0 && (uint64_t)(len + 1) * (uint64_t)(dr_len + 1) <= kSwLdsDir && (uint32_t)len <= kSwLdsA;
}
// dirs: scratch, task t uses [dir_off, dir_off + cons_sw_scratch_bytes(len, dr_len))
static inline uint64_t cons_sw_scratch_bytes(uint32_t len, uint32_t dr_len)
{
    if (cons_sw_in_lds((int)len, (int)dr_len)) return 0;
    uint64_t b = (uint64_t)(len + 1) * (dr_len + 1);
    if (dr_len > 64) b += 16 + 2ull * (dr_len + 1) * 8;                    // the serial form's two rolling rows
    return (b + 15) & ~15ull;
}
hipError_t launch_cons_sw(const uint8_t *seq, const uint64_t *roff, const uint32_t *rlen, const ConsSwTask *tasks, uint32_t n_tasks,
                          const uint8_t *dr_chars, const uint32_t *dr_off, const uint32_t *dr_len, uint8_t *dirs, ConsSwOut *out, hipStream_t st);

} // namespace crass
