// How long the traversal image of a whole world takes to build, step by step and by the number of worker threads (host only: no GPU).
//   g++ -O2 -std=c++17 -pthread -Ivoxel-rs_amd/csrc/hip -o /tmp/image_build_time profiles/tools/image_build_time.cpp
//   /tmp/image_build_time <world frame file> <1 = ESVO | 2 = CSVO> <layout 1 | 2> <threads> ...
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "traversal_image.hpp"

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(size_t(n) + 64);
    if (fread(buf.data(), 1, size_t(n), f) != size_t(n)) return 1;
    fclose(f);
    const int fmt = atoi(argv[2]), layout = atoi(argv[3]);
    const uint64_t head = 4 + (fmt == 1 ? 20 : 4);
    for (int a = 4; a < argc; ++a) {
        const unsigned threads = unsigned(atoi(argv[a]));
        vximg::WorldImage img(fmt, layout == 2 ? vximg::kOct64Wide : vximg::kOct64);
        const auto t0 = std::chrono::steady_clock::now();
        const bool ok = img.update(buf.data(), uint64_t(n) - head, nullptr, 0, threads);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double* t = img.last_timing();
        printf("{\"format\": %d, \"threads\": %u, \"ok\": %d, \"seconds\": %.3f, \"root_walk\": %.3f, \"chunk_walk\": %.3f, \"place\": %.3f, \"encode\": %.3f, \"root_encode\": %.3f, \"image_MB\": %.1f, \"chunks\": %zu}\n",
               fmt, threads, ok ? 1 : 0, s, t[0], t[1], t[2], t[3], t[4], img.frame_bytes() / 1e6, img.chunk_count());
        fflush(stdout);
    }
    return 0;
}
