// What does the first touch of a few GB of anonymous memory cost on this host? (The traversal image's frame is written once, by the commit's
// workers, into pages nobody has touched: profiles/round6/README.md, first commit.)   touchbench <GiB> <threads>
// mode 0: plain mmap; 1: + MADV_HUGEPAGE; 2: + MADV_POPULATE_WRITE by the threads, a slice each, before the writes
#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static long anon_huge_kb() {
    FILE* f = std::fopen("/proc/self/smaps_rollup", "r");
    if (!f) return -1;
    char line[256];
    long kb = -1;
    while (std::fgets(line, sizeof line, f))
        if (std::sscanf(line, "AnonHugePages: %ld kB", &kb) == 1) break;
    std::fclose(f);
    return kb;
}
int main(int argc, char** argv) {
    const size_t bytes = size_t(argc > 1 ? atof(argv[1]) * (1 << 30) : size_t(2) << 30);
    const unsigned threads = argc > 2 ? atoi(argv[2]) : 16;
    for (int mode = 0; mode < 3; ++mode) {
        char* p = static_cast<char*>(mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
        if (p == MAP_FAILED) return 1;
        if (mode >= 1) madvise(p, bytes, MADV_HUGEPAGE);
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> pool;
        const size_t slice = (bytes / threads + 4095) / 4096 * 4096;
        int populate_failed = 0;
        for (unsigned t = 0; t < threads; ++t)
            pool.emplace_back([&, t] {
                const size_t from = std::min(bytes, t * slice), to = std::min(bytes, from + slice);
                if (mode == 2 && madvise(p + from, to - from, MADV_POPULATE_WRITE) != 0) populate_failed = 1;
                for (size_t o = from; o < to; o += 32768) std::memset(p + o, 1, std::min<size_t>(32768, to - o));  // (a chunk's image at a time)
            });
        for (auto& th : pool) th.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("mode %d: %.2f GiB first written by %u threads in %.3f s = %.1f GB/s; AnonHugePages %ld kB%s\n", mode, bytes / double(1 << 30), threads, s,
                    bytes / s / 1e9, anon_huge_kb(), populate_failed ? " (MADV_POPULATE_WRITE refused)" : "");
        munmap(p, bytes);
    }
    char buf[256];
    for (const char* f : {"/sys/kernel/mm/transparent_hugepage/enabled", "/sys/kernel/mm/transparent_hugepage/defrag"}) {
        FILE* fh = std::fopen(f, "r");
        if (fh && std::fgets(buf, sizeof buf, fh)) std::printf("%s: %s", f, buf);
        if (fh) std::fclose(fh);
    }
}
