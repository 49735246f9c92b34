// Everything of the product that runs on HOST THREADS, driven hard under the sanitizers (make -C voxel-rs_amd sanitize; CPU build only -- the
// GPU boxes run no sanitizer): the scene builder's workers, the chunk streamer's background workers feeding pump(), the traversal image's
// Workers (whole-world build and incremental WorldImage::update at 1 / 4 / 16 threads), and the device header's traversal (vx_device.hpp
// behind the plain-C++ platform shims, tests/cpp/device_on_host.cpp) walking what they built. Test infrastructure: nothing of the product
// links against it. What the reference's design relies on (src/systems/worldsvo.rs:90-151): worker threads only BUILD chunks, the caller's
// thread owns the buffer -- a data race or a stray write here is a bug in that hand-over.
//
// Not covered (needs a HIP device): the pipelined vx_commit's worker inside runtime.cpp; its queue is exercised on the GPU by
// tests/test_streaming.py.
//
// Exit code 0 = every check passed (and the sanitizer, which aborts on a finding, had nothing to say).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "voxel_hip.h"

#include "csvo.hpp"
#include "esvo.hpp"
#include "scene.hpp"
#include "stream.hpp"
#include "traversal_image.hpp"

// the streamer's device half is never reached here (pump(nullptr, ..)): symbols for the linker, an abort for a caller
extern "C" {
uint8_t* vx_staging_ptr(vx_context*) { std::abort(); }
size_t vx_arena_capacity(const vx_context*) { std::abort(); }
int vx_commit(vx_context*, uint32_t, const vx_range*, uint32_t, uint64_t) { std::abort(); }
const char* vx_last_error(void) { return "stub"; }
// tests/cpp/device_on_host.cpp, compiled into this binary
void devhost_picker(int svo_type, const uint8_t* world, uint64_t world_bytes, const vx_material* mats, uint32_t n_mats, const uint8_t* tex, uint32_t tw, uint32_t th,
                    uint32_t layers, uint32_t levels, const uint32_t* level_offset, const vx_picker_task* tasks, uint32_t n, vx_picker_result* results, int cast_translucent);
void devhost_image_cast(int svo_type, int layout, int shallow, int walk_mode, const uint8_t* world, uint64_t world_bytes, const uint8_t* image, uint64_t image_bytes,
                        const uint8_t* origin, const vx_material* mats, uint32_t n_mats, const uint8_t* tex, uint32_t tw, uint32_t th, uint32_t layers, uint32_t levels,
                        const uint32_t* level_offset, const vx_picker_task* tasks, uint32_t n, int cast_translucent, vx_result* results, uint32_t* steps);
}

using namespace vx;

namespace {

int g_failures = 0;
#define CHECK(cond, ...)                                 \
    do {                                                 \
        if (!(cond)) {                                   \
            std::fprintf(stderr, "FAILED %s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__);           \
            std::fprintf(stderr, "\n");                  \
            ++g_failures;                                \
        }                                                \
    } while (0)

uint32_t rng_state = 0x5EED0006u;
uint32_t rnd() {  // xorshift32: the same rays in every build
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 17;
    rng_state ^= rng_state << 5;
    return rng_state;
}
float rnd01() { return float(rnd() >> 8) * (1.0f / 16777216.0f); }

// [f32 2^-depth][header][arena] + the 16 zero bytes a context keeps behind the world buffer
template <class WorldT>
std::vector<uint8_t> frame_of(const WorldT& w, size_t header) {
    std::vector<uint8_t> f(4 + header + w.size_in_bytes() + 16, 0);
    const float scale = std::ldexp(1.0f, -int(w.depth()));
    std::memcpy(f.data(), &scale, 4);
    w.write_to(f.data() + 4);
    return f;
}

struct Look {
    std::vector<vx_material> mats;
    std::vector<uint8_t> tex;  // 4x4 texels, 2 layers, 1 level, opaque
    uint32_t level_offset[16] = {};
    Look() {
        mats.resize(8);
        for (size_t i = 0; i < mats.size(); ++i) {
            vx_material& m = mats[i];
            m.specular_pow = 8.0f; m.specular_strength = 0.25f;
            m.tex_top = 0; m.tex_side = 1; m.tex_bottom = 1;
            m.tex_top_normal = m.tex_side_normal = m.tex_bottom_normal = -1;
        }
        tex.assign(4 * 4 * 4 * 2, 0xff);
        for (size_t i = 0; i < tex.size(); i += 4) tex[i] = uint8_t(40 + 3 * i);
    }
};

std::vector<vx_picker_task> rays_into(uint32_t depth, uint32_t h_max, uint32_t n) {
    std::vector<vx_picker_task> t(n);
    const float edge = float(1u << depth);
    for (vx_picker_task& r : t) {
        std::memset(&r, 0, sizeof r);
        r.max_dst = -1.0f;
        r.pos[0] = rnd01() * edge; r.pos[1] = float(h_max) + 2.0f + rnd01() * 20.0f; r.pos[2] = rnd01() * edge;
        float d[3] = {rnd01() - 0.5f, -0.2f - rnd01(), rnd01() - 0.5f};
        const float len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (int k = 0; k < 3; ++k) r.dir[k] = d[k] / len;
    }
    // a few that start inside the terrain (rays led into voxels: the walk on the world's bytes)
    for (uint32_t i = 0; i < n / 8; ++i) t[i].pos[1] = float(h_max) * rnd01();
    return t;
}

// the whole-world image at 1 / 4 / 16 worker threads: the same bytes (placement happens on the caller's thread, in the chunks' order)
template <class WorldT>
void image_builds(const char* name, int svo_type, const WorldT& world, size_t header, uint32_t h_max, const Look& look) {
    const std::vector<uint8_t> frame = frame_of(world, header);
    const uint64_t used = world.size_in_bytes();
    std::vector<uint32_t> first;
    for (unsigned threads : {1u, 4u, 16u}) {
        vximg::WorldImage img(svo_type, vximg::kOct64);
        const bool ok = img.update(frame.data(), used, nullptr, 0, threads);
        CHECK(ok, "%s: image build with %u threads failed", name, threads);
        if (!ok) return;
        std::vector<uint32_t> words(img.frame().data(), img.frame().data() + img.frame().size());
        if (first.empty()) first = words;
        CHECK(words == first, "%s: the image built on %u threads differs from the one built on 1", name, threads);
    }
    // the device header walks both encodings: the same hits from the world's bytes and from its image (16 resident levels: the deep build's stack)
    const std::vector<vx_picker_task> tasks = rays_into(world.depth(), h_max, 1500);
    std::vector<vx_picker_result> on_bytes(tasks.size());
    devhost_picker(svo_type, frame.data(), frame.size(), look.mats.data(), uint32_t(look.mats.size()), look.tex.data(), 4, 4, 2, 1, look.level_offset, tasks.data(),
                   uint32_t(tasks.size()), on_bytes.data(), 1);
    first.resize(first.size() + 16, 0u);  // (the padding a context keeps behind the image)
    for (int shallow : {1, 2}) {
        std::vector<vx_result> on_image(tasks.size());
        std::vector<uint32_t> steps(tasks.size());
        devhost_image_cast(svo_type, 1, shallow, 2, frame.data(), frame.size(), reinterpret_cast<const uint8_t*>(first.data()), first.size() * 4,
                           reinterpret_cast<const uint8_t*>(first.data()) /* a CSVO world's origins are units of the image itself */, look.mats.data(), uint32_t(look.mats.size()), look.tex.data(), 4, 4, 2, 1, look.level_offset,
                           tasks.data(), uint32_t(tasks.size()), 1, on_image.data(), steps.data());
        size_t hits = 0, differ = 0;
        for (size_t i = 0; i < tasks.size(); ++i) {
            const bool hit = on_image[i].t > 0.0f;
            hits += hit;
            if (hit != (on_bytes[i].dst > 0.0f) || (hit && (std::memcmp(&on_image[i].t, &on_bytes[i].dst, 4) != 0 || std::memcmp(on_image[i].pos, on_bytes[i].pos, 12) != 0))) ++differ;
        }
        CHECK(differ == 0, "%s: %zu of %zu rays differ between the world's bytes and its image (stack build %d)", name, differ, tasks.size(), shallow);
        CHECK(hits > tasks.size() / 2, "%s: only %zu of %zu rays hit", name, hits, tasks.size());
    }
    std::printf("%s: whole-world image on 1 / 4 / 16 threads identical (%zu words); %zu rays agree on bytes and image\n", name, first.size(), tasks.size());
}

// Worlds the serializer did NOT write: a few bytes of a good world overwritten at random, 300 times over. The image build may refuse such a world (the
// context then traverses its bytes) or image what it reads; what it may not do is read or write out of bounds or never end -- the emitters write a chunk's
// words through bare pointers into room they asked for up front (traversal_image.hpp, Emitted::room), and the sanitizer watches them here.
template <class WorldT>
void damaged_worlds(const char* name, int svo_type, const WorldT& world, size_t header) {
    const std::vector<uint8_t> good = frame_of(world, header);
    const uint64_t used = world.size_in_bytes();
    int imaged = 0, refused = 0;
    for (int round = 0; round < 300; ++round) {
        std::vector<uint8_t> f = good;
        const int hits = 1 + int(rnd() % 24);
        for (int k = 0; k < hits; ++k) {
            // (half of them where the chunks' nodes lie: the last third of the arena holds the latest chunks, and the root octree behind them)
            const size_t at = 4 + header + size_t((rnd() & 1u) ? rnd() % used : used - 1 - rnd() % (used / 3));
            f[at] = (rnd() & 3u) ? uint8_t(rnd()) : uint8_t(0xff);
        }
        for (vximg::Layout layout : {vximg::kOct64, vximg::kEsvo48}) {
            vximg::WorldImage img(svo_type, layout);
            if (img.update(f.data(), used, nullptr, 0, 4)) ++imaged;
            else ++refused;
        }
    }
    CHECK(imaged + refused == 600, "%s: damaged worlds", name);
    std::printf("%s: 300 damaged worlds, both layouts: %d imaged, %d refused, none out of bounds\n", name, imaged, refused);
}

// the streamer (its own worker threads build chunks) feeding an incrementally maintained image (Workers at 1 / 4 / 16 threads per commit) for `steps` moves of
// a camera, until `steps` commits have been made; at the end the image must hold the tree a from-scratch build of the final world holds
template <class WorldT, class SerializedT>
void stream_and_update(const char* name, int svo_type, size_t header, uint32_t streamer_threads, int steps) {
    const uint32_t depth = 9, radius = 3;
    systems::WorldStreamer<WorldT, SerializedT> streamer(depth, 0x5EED0001u, radius, 0, int32_t((1u << depth) / 32u), streamer_threads);
    std::vector<uint8_t> mirror(size_t(96) << 20, 0);
    vximg::WorldImage image(svo_type, vximg::kOct64);
    bool image_ok = false;
    int commits = 0;
    streamer.on_dry_commit = [&](WorldT& world, const std::vector<vx_range>& ranges) {
        const float scale = std::ldexp(1.0f, -int(world.depth()));
        std::memcpy(mirror.data(), &scale, 4);
        if (!world.write_changes_to(mirror.data() + 4, mirror.size() - 5, true)) {
            CHECK(false, "%s: the mirror is too small", name);
            return;
        }
        std::vector<vximg::Range> changed;
        for (const vx_range& r : ranges) changed.push_back(vximg::Range{r.start, r.length});
        const unsigned threads[3] = {1u, 4u, 16u};
        image_ok = image.update(mirror.data(), world.size_in_bytes(), changed.data(), changed.size(), threads[commits % 3]);
        ++commits;
    };
    const float mid = float(1u << (depth - 1));
    uint64_t events = 0;
    int moves = 0;
    for (; commits < steps && moves < 20 * steps; ++moves) {
        // a walk that re-centres every few steps (chunks load, unload, change their level of detail), its events applied a dozen at a time
        const float a = 0.05f * float(moves), x = mid + 120.0f * std::cos(a), z = mid + 120.0f * std::sin(a), y = 40.0f + float(moves % 7);  // (a circle inside the world)
        events += streamer.move_to(x, y, z);
        (void)streamer.pump(nullptr, 12, (moves & 1) != 0);  // (alternately: only what is ready / wait for the workers)
    }
    while (streamer.pending_events()) streamer.pump(nullptr, 400, true);
    CHECK(commits >= steps, "%s: only %d commits in %d moves", name, commits, moves);
    CHECK(image_ok, "%s: the incremental image was lost", name);
    if (!image_ok) return;
    // from scratch, of the world as it stands
    const std::vector<uint8_t> frame = frame_of(streamer.world(), header);
    vximg::WorldImage fresh(svo_type, vximg::kOct64);
    const bool ok = fresh.update(frame.data(), streamer.world().size_in_bytes(), nullptr, 0, 4);
    CHECK(ok, "%s: the from-scratch image failed", name);
    if (!ok) return;
    const int same = vximg::oct64_same_tree(image.frame().data(), image.frame().size(), fresh.frame().data(), fresh.frame().size());
    CHECK(same == 1, "%s: after %d commits the incremental image and a fresh one disagree (%d)", name, commits, same);
    std::printf("%s: %d moves, %llu events, %d commits with incremental image updates (streamer workers %u): same tree as a fresh build, %zu resident chunks\n", name,
                moves, (unsigned long long)events, commits, streamer_threads, streamer.resident_chunks());
}

}  // namespace

int main(int argc, char** argv) {
    const int steps = argc > 1 ? std::atoi(argv[1]) : 200;
    const Look look;
    {
        Esvo<EsvoSerializedChunk> world;
        const SceneStats st = build_heightfield_scene(world, 8, 0x5EED0001u, 4);
        image_builds("esvo depth 8", 1, world, 20, st.h_max, look);
        damaged_worlds("esvo depth 8", 1, world, 20);
    }
    {
        Csvo world;
        const SceneStats st = build_heightfield_scene(world, 8, 0x5EED0001u, 4);
        image_builds("csvo depth 8", 2, world, 4, st.h_max, look);
        damaged_worlds("csvo depth 8", 2, world, 4);
    }
    for (uint32_t threads : {1u, 4u, 16u}) {
        stream_and_update<Esvo<EsvoSerializedChunk>, EsvoSerializedChunk>("esvo stream", 1, 20, threads, steps);
        stream_and_update<Csvo, CsvoSerializedChunk>("csvo stream", 2, 4, threads, steps);
    }
    if (g_failures) {
        std::fprintf(stderr, "%d check(s) failed\n", g_failures);
        return 1;
    }
    std::printf("sanitize_stress: all checks passed\n");
    return 0;
}
