// The traversal image's build (WorldImage::update, what vx_commit runs on the host) on a dumped world frame (tools/dump_world.py), by phase.
//   imgbench <file> <svo_type 1|2> <threads> [repeats] [checksum 0|1] [pinned flags]
// Built with hipcc and -DVX_PINNED the world's bytes lie in hipHostMalloc'ed memory like a context's staging buffer (flags: 0 default, 0x20000000 NumaUser).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include "traversal_image.hpp"
#ifdef VX_PINNED
#include <hip/hip_runtime.h>
#endif
int main(int argc, char** argv) {
    const char* path = argv[1];
    int svo = atoi(argv[2]);
    unsigned threads = atoi(argv[3]);
    int repeats = argc > 4 ? atoi(argv[4]) : 2;
    FILE* f = fopen(path, "rb");
    uint64_t used;
    if (fread(&used, 8, 1, f) != 1) return 1;
    fseek(f, 0, SEEK_END);
    size_t n = ftell(f) - 8;
    fseek(f, 8, SEEK_SET);
#ifdef VX_PINNED
    struct { uint8_t* p; uint8_t* data() { return p; } } w{nullptr};
    if (hipHostMalloc(reinterpret_cast<void**>(&w.p), n, argc > 6 ? strtoul(argv[6], nullptr, 0) : 0) != hipSuccess) return 2;
#else
    std::vector<uint8_t> w(n);
#endif
    if (fread(w.data(), 1, n, f) != n) return 1;
    fclose(f);
    for (int r = 0; r < repeats; ++r) {
        vximg::WorldImage img(svo, vximg::kOct64);
        auto t0 = std::chrono::steady_clock::now();
        bool ok = img.update(w.data(), used, nullptr, 0, threads);
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double* t = img.last_timing();
        uint64_t sum = 0;
        if (argc > 5 && atoi(argv[5])) { const auto& fr = img.frame(); for (size_t i = 0; i < fr.size(); ++i) sum = sum * 1099511628211ull + fr[i]; }
        { FILE* fh = fopen("/proc/self/smaps_rollup", "r"); char line[256]; while (fh && fgets(line, sizeof line, fh)) if (!strncmp(line, "AnonHuge", 8)) fputs(line, stdout); if (fh) fclose(fh); }
        printf("ok %d total %.3f s: root %.3f walk %.3f place %.3f encode %.3f header %.3f; frame %.1f MB chunks %zu sum %016llx\n", ok, s, t[0], t[1], t[2], t[3], t[4],
               img.frame_bytes() / 1e6, img.chunk_count(), (unsigned long long)sum);
    }
}
