// exit_cost.cpp — what a process's END costs after _exit(): anonymous host memory (4 KB pages or THP), device memory, pinned host
// memory, a HIP context alone.  The parent (tools/exit_cost.py) measures _exit -> reaped.
//   exit_cost <host_gb> <thp 0|1> <dev_gb> <pinned_gb> <hip 0|1> [drop_threads [streams]]   (drop_threads: MADV_DONTNEED the host memory in
//   64 MB slices over that many threads before _exit, and print how long that took)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <sys/mman.h>
#include <unistd.h>
#include <thread>
#include <vector>
#include <atomic>
int main(int argc, char **argv)
{
    if (argc < 6) return 1;
    const size_t host = (size_t)(atof(argv[1]) * (1ull << 30)), dev = (size_t)(atof(argv[3]) * (1ull << 30)), pin = (size_t)(atof(argv[4]) * (1ull << 30));
    const int thp = atoi(argv[2]), hip = atoi(argv[5]);
    char *p = nullptr;
    if (host) {
        p = (char *)mmap(nullptr, host, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (p == MAP_FAILED) return 2;
        madvise(p, host, thp ? MADV_HUGEPAGE : MADV_NOHUGEPAGE);
        for (size_t i = 0; i < host; i += 4096) p[i] = 1;
    }
    if (hip || dev || pin) {
        if (hipSetDevice(0) != hipSuccess) return 3;
        hipFree(nullptr);
        if (dev) { void *d = nullptr; if (hipMalloc(&d, dev) != hipSuccess || hipMemset(d, 1, dev) != hipSuccess) return 4; hipDeviceSynchronize(); }
        if (argc > 7) {                                   // n streams, each made to own a hardware queue by a small fill
            const int ns = atoi(argv[7]);
            void *d = nullptr;
            if (hipMalloc(&d, 4096) != hipSuccess) return 6;
            for (int i = 0; i < ns; i++) { hipStream_t st; if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return 7; (void)hipMemsetAsync(d, 0, 4096, st); (void)hipStreamSynchronize(st); }
        }
        if (pin) { void *h = nullptr; if (hipHostMalloc(&h, pin, hipHostMallocDefault) != hipSuccess) return 5; for (size_t i = 0; i < pin; i += 4096) ((char *)h)[i] = 1; }
    }
    if (argc > 6 && host) {
        const int nt = atoi(argv[6]);
        const double d0 = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
        std::atomic<size_t> next{0};
        const size_t slice = 64u << 20;
        std::vector<std::thread> th;
        for (int t = 0; t < nt; t++) th.emplace_back([&] { for (;;) { const size_t a = next.fetch_add(slice); if (a >= host) break; madvise(p + a, std::min(slice, host - a), MADV_DONTNEED); } });
        for (auto &x : th) x.join();
        fprintf(stderr, "dropped by %d threads in %.3f s\n", nt, std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count() - d0);
    }
    const double e = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    printf("%.6f\n", e); fflush(stdout);
    _exit(0);
}
