// How long does the host wait for "a tiny kernel finished"?  hipStreamSynchronize vs polling a pinned word the
// kernel writes.  (Decides whether replacing the step's three synchronisations is worth anything.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void k_signal(volatile uint32_t *flag, uint32_t v) { __threadfence_system(); *flag = v; }
__global__ void k_work(uint32_t *p) { p[threadIdx.x] += 1; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    uint32_t *d; hipMalloc(&d, 4096);
    volatile uint32_t *h; hipHostMalloc((void **)&h, 64, hipHostMallocDefault); *h = 0;
    const int N = 2000;
    for (int mode = 0; mode < 2; mode++) {
        double tot = 0;
        for (int i = 1; i <= N; i++) {
            const double t0 = now();
            hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d);
            if (mode == 0) hipStreamSynchronize(st);
            else { hipLaunchKernelGGL(k_signal, dim3(1), dim3(1), 0, st, h, (uint32_t)i); while (*h != (uint32_t)i) { __builtin_ia32_pause(); } }
            tot += now() - t0;
        }
        printf("%s: %.2f us per launch+wait\n", mode == 0 ? "hipStreamSynchronize" : "pinned flag polling ", tot / N);
    }
    return 0;
}
