// How much does a software grid barrier cost on MI355X (256 CUs, 8 XCDs with private L2s) against a kernel boundary?
// Decides whether the ~19 dependent micro-kernels of the device merge are better off as phases of ONE persistent kernel.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

// monotonically increasing counter: phase p is complete when counter >= (p+1)*gridDim.x
__device__ __forceinline__ void grid_sync(uint32_t *bar, uint32_t &phase)
{
    __syncthreads();
    phase++;
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t target = phase * gridDim.x;
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_barriers(uint32_t *bar, uint32_t *data, int n_phases)
{
    uint32_t phase = 0;
    for (int p = 0; p < n_phases; p++) {
        data[blockIdx.x * blockDim.x + threadIdx.x] += p;         // a little work with a cross-block dependency
        grid_sync(bar, phase);
    }
}
__global__ __launch_bounds__(1024) void k_tiny(uint32_t *data, int p) { data[blockIdx.x * blockDim.x + threadIdx.x] += p; }

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    uint32_t *bar, *data;
    hipMalloc(&bar, 64); hipMalloc(&data, 256 * 1024 * 4);
    hipMemset(data, 0, 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 1024}) {
        for (int blocks : {64, 256}) {
            for (int phases : {1, 20, 200}) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; rep++) {
                    hipMemsetAsync(bar, 0, 4, st);
                    hipEventRecord(e0, st);
                    hipLaunchKernelGGL(k_barriers, dim3(blocks), dim3(threads), 0, st, bar, data, phases);
                    hipEventRecord(e1, st);
                    hipStreamSynchronize(st);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("persistent %4d blocks x %4d threads, %3d barriers: %8.2f us total, %6.2f us per barrier\n", blocks, threads, phases, best * 1e3f, best * 1e3f / phases);
            }
        }
    }
    for (int blocks : {64, 256}) {
        float best = 1e9f;
        const int K = 200;
        for (int rep = 0; rep < 5; rep++) {
            hipEventRecord(e0, st);
            for (int p = 0; p < K; p++) hipLaunchKernelGGL(k_tiny, dim3(blocks), dim3(1024), 0, st, data, p);
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("kernel chain %4d blocks x 1024 threads, %d launches: %6.2f us per launch\n", blocks, K, best * 1e3f / K);
    }
    // cooperative launch of the same kernel (guaranteed co-residency): what does the launch itself cost?
    {
        int phases = 20, blocks = 256;
        void *args[] = {&bar, &data, &phases};
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            hipMemsetAsync(bar, 0, 4, st);
            hipEventRecord(e0, st);
            hipError_t e = hipLaunchCooperativeKernel((const void *)k_barriers, dim3(blocks), dim3(1024), args, 0, st);
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            if (e != hipSuccess) { printf("cooperative launch failed: %s\n", hipGetErrorString(e)); break; }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("cooperative launch, 256 x 1024, 20 barriers: %8.2f us total\n", best * 1e3f);
    }
    return 0;
}
