// How fast does the HOST read pinned memory (hipHostMalloc) — default flags, non-coherent, and plain malloc?  (round 6: the helper
// thread's pattern build read 2 MB of pinned memory and took 1.1 ms.)   hipcc -O2 pinned_read.cpp -o pinned_read
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void run(const char *tag, unsigned char *p, size_t n)
{
    memset(p, 1, n);
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        unsigned long long s = 0;
        for (size_t i = 0; i < n; i += 8) s += *(const unsigned long long *)(p + i);
        double t1 = now();
        unsigned char *q = (unsigned char *)malloc(n);
        double t2 = now();
        memcpy(q, p, n);
        double t3 = now();
        printf("%-28s %zu bytes: 8-byte loads %.1f us, memcpy to malloc'd %.1f us (sum %llu)\n", tag, n, t1 - t0, t3 - t2, s + q[5]);
        free(q);
    }
}
int main()
{
    const size_t n = 2 << 20;
    unsigned char *a = nullptr, *b = nullptr, *c = nullptr;
    hipHostMalloc((void **)&a, n, hipHostMallocDefault);
    hipHostMalloc((void **)&b, n, hipHostMallocNonCoherent);
    hipHostMalloc((void **)&c, n, hipHostMallocCoherent);
    unsigned char *m = (unsigned char *)malloc(n);
    run("malloc", m, n);
    run("hipHostMallocDefault", a, n);
    run("hipHostMallocNonCoherent", b, n);
    run("hipHostMallocCoherent", c, n);
    return 0;
}
