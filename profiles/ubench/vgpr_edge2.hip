// vgpr_edge2.hip — the exact instruction sequence of k_recruit_finish's "shift the reversed 128 bits down" step (ISA pasted
// from the hipcc 7.2 output, fixed registers v14..v23 / s[12:13]) in a kernel whose .vgpr_count is exactly 24 (k_edge0) or
// 32 (k_edge1: dead clobber of v31).  The host checks every result; wrong ones are counted for first-generation
// blocks (< 256) and later ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define EDGE_BODY(NAME, PADREG, DROPREG, NOPS, ZERO1819)                                                              \
__global__ __launch_bounds__(256) void NAME(const uint32_t *lens, const uint4 *cs, uint32_t n, uint4 *out, uint64_t magic)  \
{                                                                                                                        \
    if (magic == 0x1234567ull) asm volatile("v_mov_b32 " PADREG ", 0" ::: PADREG);                                       \
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;                                                                  \
    if (k >= n) return;                                                                                                  \
    const uint32_t len = lens[k];                                                                                        \
    const uint4 c = cs[k];                                                                                               \
    uint32_t a0, a1, a2, a3;                                                                                             \
    asm volatile(                                                                                                        \
        "v_mov_b32 v20, %4\n\t"                                                                                          \
        "v_lshlrev_b32 v22, 1, %4\n\t"                                                                                   \
        "v_mov_b32 v14, %5\n\tv_mov_b32 v15, %6\n\tv_mov_b32 v16, %7\n\tv_mov_b32 v17, %8\n\t"                          \
        "v_cmp_gt_u32_e32 vcc, 33, v20\n\t"                                                                              \
        "s_and_saveexec_b64 s[12:13], vcc\n\t"                                                                           \
        "s_xor_b64 s[12:13], exec, s[12:13]\n\t"                                                                         \
        "v_sub_u32_e32 v14, 64, v22\n\t"                                                                                 \
        "v_lshrrev_b64 v[14:15], v14, v[16:17]\n\t"                                                                      \
        "s_or_saveexec_b64 s[12:13], s[12:13]\n\t"                                                                       \
        ZERO1819                                                                                                         \
        "s_xor_b64 exec, exec, s[12:13]\n\t"                                                                             \
        NOPS                                                                                                             \
        "v_sub_u32_e32 " DROPREG ", 0x80, v22\n\t"                                                                       \
        "v_subrev_u32_e32 v18, 64, v22\n\t"                                                                              \
        "v_lshrrev_b64 v[14:15], " DROPREG ", v[14:15]\n\t"                                                              \
        "v_lshlrev_b64 v[18:19], v18, v[16:17]\n\t"                                                                      \
        "v_or_b32_e32 v15, v15, v19\n\t"                                                                                 \
        "v_or_b32_e32 v14, v14, v18\n\t"                                                                                 \
        "v_lshrrev_b64 v[18:19], " DROPREG ", v[16:17]\n\t"                                                              \
        "s_or_b64 exec, exec, s[12:13]\n\t"                                                                              \
        "v_mov_b64_e32 v[16:17], v[18:19]\n\t"                                                                           \
        "v_mov_b32 %0, v14\n\tv_mov_b32 %1, v15\n\tv_mov_b32 %2, v16\n\tv_mov_b32 %3, v17\n\t"                          \
        : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3)                                                                         \
        : "v"(len), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w)                                                               \
        : "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "s12", "s13", "vcc");                     \
    out[k] = make_uint4(a0, a1, a2, a3);                                                                                 \
}
#define Z64 "v_mov_b64_e32 v[18:19], 0\n\t"
#define Z32 "v_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\t"
EDGE_BODY(k_edge0, "v23", "v23", "", Z64)            // as compiled: drop in v23, 24 VGPRs
EDGE_BODY(k_edge1, "v31", "v23", "", Z64)            // same, 32 VGPRs
EDGE_BODY(k_edge2, "v23", "v21", "", Z64)            // drop in v21, v23 unused, still 24 VGPRs
EDGE_BODY(k_edge3, "v23", "v23", "s_nop 7\n\t", Z64)   // wait states after the EXEC write
EDGE_BODY(k_edge4, "v23", "v23", "", Z32)            // two 32-bit moves instead of v_mov_b64
EDGE_BODY(k_edge5, "v24", "v23", "", Z64)            // 25 VGPRs
EDGE_BODY(k_edge6, "v31", "v31", "", Z64)            // drop in v31 = last register of a 32-register allocation
EDGE_BODY(k_edge7, "v39", "v39", "", Z64)            // drop in v39 = last register of a 40-register allocation
EDGE_BODY(k_edge8, "v39", "v31", "", Z64)            // drop in v31, 40 registers allocated
int main()
{
    const uint32_t nblk = 517, n = 132344, cap = nblk * 256;
    uint32_t *lens; uint4 *cs, *out;
    (void)hipMalloc(&lens, cap * 4); (void)hipMalloc(&cs, cap * 16); (void)hipMalloc(&out, cap * 16);
    std::vector<uint32_t> h(cap); std::vector<uint4> hc(cap), ho(cap);
    for (uint32_t i = 0; i < cap; i++) {
        uint32_t x = i * 2654435761u >> 9; h[i] = (x % 5 == 0) ? 33 + (x >> 3) % 3 : 26 + (x >> 3) % 7;
        const uint64_t c0 = 0x9E3779B97F4A7C15ull * (i + 1), c1 = 0xD1B54A32D192ED03ull * (i + 7);
        hc[i] = make_uint4((uint32_t)c0, (uint32_t)(c0 >> 32), (uint32_t)c1, (uint32_t)(c1 >> 32));
    }
    (void)hipMemcpy(lens, h.data(), cap * 4, hipMemcpyHostToDevice); (void)hipMemcpy(cs, hc.data(), cap * 16, hipMemcpyHostToDevice);
    typedef void (*kern_t)(const uint32_t *, const uint4 *, uint32_t, uint4 *, uint64_t);
    const kern_t kerns[] = {k_edge0, k_edge1, k_edge2, k_edge3, k_edge4, k_edge5, k_edge6, k_edge7, k_edge8};
    const char *names[] = {"24 VGPRs, drop in v23 (as compiled)", "32 VGPRs (dead clobber of v31)", "24 VGPRs, drop in v21 (v23 unused)",
                           "24 VGPRs, s_nop 7 after the EXEC write", "24 VGPRs, 2x v_mov_b32 instead of v_mov_b64", "25 VGPRs (dead clobber of v24)",
                           "32 VGPRs, drop in v31 (last of 32)", "40 VGPRs, drop in v39 (last of 40)", "40 VGPRs, drop in v31"};
    for (int kv = 0; kv < 9; kv++) {
        uint64_t tot0 = 0, tot1 = 0, first_bad = ~0ull, badA = 0, badB = 0;
        for (int rep = 0; rep < 100; rep++) {
            (void)hipMemset(out, 0xFF, cap * 16);
            hipLaunchKernelGGL(kerns[kv], dim3(nblk), dim3(256), 0, 0, lens, cs, n, out, 0ull);
            (void)hipMemcpy(ho.data(), out, cap * 16, hipMemcpyDeviceToHost);
            for (uint32_t i = 0; i < n; i++) {
                const uint64_t c0 = hc[i].x | ((uint64_t)hc[i].y << 32), c1 = hc[i].z | ((uint64_t)hc[i].w << 32);
                const uint32_t drop = 128u - 2u * h[i];
                uint64_t e0, e1;
                if (drop < 64) { e0 = (c0 >> drop) | (c1 << (64 - drop)); e1 = c1 >> drop; } else { e0 = c1 >> (drop - 64); e1 = 0; }
                const uint64_t r0 = ho[i].x | ((uint64_t)ho[i].y << 32), r1 = ho[i].z | ((uint64_t)ho[i].w << 32);
                if (r0 != e0 || r1 != e1) { (i < 65536 ? tot0 : tot1)++; if (i < first_bad) first_bad = i; (h[i] < 33 ? badA : badB)++; }
            }
        }
        printf("%-46s wrong results over 100 launches: slots<65536 %llu, slots>=65536 %llu (len<33: %llu, len>=33: %llu), first bad slot %lld (%s)\n", names[kv],
               (unsigned long long)tot0, (unsigned long long)tot1, (unsigned long long)badA, (unsigned long long)badB, (long long)first_bad, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
