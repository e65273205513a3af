// open_cost.cpp — how long open(O_CREAT) takes in /dev/shm: alone, beside a thread writing a 300 MB file, and in a process that
// holds N GB of anonymous memory and the input's page cache mapped (what crass-hip looks like when it writes its outputs)
//   open_cost <dir> <anon_gb> <big_write 0|1> <threads>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const std::string dir = argv[1];
    const size_t anon = (size_t)(atof(argv[2]) * (1ull << 30));
    const int big = atoi(argv[3]), nt = atoi(argv[4]);
    if (anon) { char *p = (char *)mmap(nullptr, anon, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0); for (size_t i = 0; i < anon; i += 4096) p[i] = 1; }
    const size_t bn = 300u << 20;
    char *blob = (char *)malloc(bn); memset(blob, 'x', bn);
    std::vector<char> small(1 << 20, 'y');
    std::atomic<int> next{0};
    std::vector<double> t_open(101, 0);
    const double t0 = now();
    auto work = [&]() {
        for (;;) {
            const int i = next.fetch_add(1);
            if (i > 100) break;
            const std::string path = dir + "/f" + std::to_string(i);
            const double a = now();
            const int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
            t_open[i] = now() - a;
            if (fd < 0) { perror("open"); exit(1); }
            if (i == 0 && big) { size_t left = bn; const char *p = blob; while (left) { ssize_t w = write(fd, p, left); if (w <= 0) break; p += w; left -= w; } }
            else if (write(fd, small.data(), small.size()) < 0) {}
            close(fd);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
    double sum = 0, mx = 0;
    for (double d : t_open) { sum += d; if (d > mx) mx = d; }
    printf("anon %.0f GB, big write %d, %d threads: total %.4f s; opens summed %.4f s, slowest %.4f s\n", anon / 1073741824.0, big, nt, now() - t0, sum, mx);
    for (int i = 0; i <= 100; i++) unlink((dir + "/f" + std::to_string(i)).c_str());
}
