// dev harness: the chunk walk alone, single thread, N chunks
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#include "traversal_image.hpp"
using namespace vximg;
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb");
    int svo = atoi(argv[2]);
    uint64_t used;
    if (fread(&used, 8, 1, f) != 1) return 1;
    fseek(f, 0, SEEK_END);
    size_t n = ftell(f) - 8;
    fseek(f, 8, SEEK_SET);
    std::vector<uint8_t> wv(n);
    if (fread(wv.data(), 1, n, f) != n) return 1;
    fclose(f);
    const uint8_t* world = wv.data();
    const Bytes b{world + 8, size_t(used)};
    const Words w{reinterpret_cast<const uint32_t*>(world + 4), size_t(used / 4 + 5)};
    uint32_t scale_bits; memcpy(&scale_bits, world, 4);
    const uint32_t depth = 127u - ((scale_bits >> 23) & 0xffu);
    Tree root; std::vector<ChunkRef> refs;
    if (svo == 1) { const uint32_t p = w.at(4); EsvoWalker(w, root, &refs).run((p & 0x80000000u) ? 4u + (p & 0x7fffffffu) : p, w.at(0) & 0xffffu, depth); }
    else { uint32_t root_ptr; memcpy(&root_ptr, world + 4, 4); root.root = walk_root(b, root_ptr, depth, root, refs); }
    std::sort(refs.begin(), refs.end(), [](const ChunkRef& x, const ChunkRef& y) { return x.key < y.key; });
    size_t N = argc > 3 ? atoi(argv[3]) : 50000;
    if (N > refs.size()) N = refs.size();
    size_t stride = refs.size() / N;
    for (int rep = 0; rep < 3; ++rep) {
        Emitted e; uint64_t words = 0, relocs = 0, sum = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (size_t i = 0; i < N; ++i) {
            const ChunkRef& r = refs[i * stride];
            if (svo == 1) EsvoEmitter(w, e).run(r.key, r.masks, r.levels); else ChunkEmitter(b, e).run(r.key);
            words += e.n_words; relocs += e.n_relocs;
            sum += e.words[e.n_words / 2];
        }
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%zu chunks %.3f s = %.2f us a chunk; %.0f words %.0f relocs a chunk; sum %016llx\n", N, s, s / N * 1e6, double(words) / N, double(relocs) / N, (unsigned long long)sum);
    }
}
