// Seeded synthetic heightfield scenes for benchmarks and at-scale parity runs.
//
// Not part of the reference: its terrain comes from the un-vendored `noise` crate (Perlin), so scenes
// here use an all-integer value-noise heightfield that any language reproduces bit for bit:
//   h(x,z) = clamp(1 + sum_{o<5} (bilerp16(o,x,z) * amp_o) >> 16, 1, N/4),  N = 2^depth
//   lattice value  = hash32(seed, o, i, j) & 0xffff,  cell size lambda_o = max(1, N >> (2+o)),
//   amp_o = (N/8) >> o.
// Only surface-shell voxels are stored (a voxel with an empty 6-neighbour; outside the domain counts as
// "same height", so the domain border grows no walls). Block ids follow the reference's generator
// (src/gamelogic/worldgen.rs:303-310): grass on top, dirt up to 3 below, stone underneath.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

#include "chunk.hpp"
#include "csvo.hpp"
#include "esvo.hpp"

namespace vx {

struct SceneStats {
    uint64_t chunks = 0;
    uint64_t leaves = 0;
    uint32_t h_max = 0;
};

inline uint32_t scene_hash32(uint32_t seed, uint32_t o, uint32_t i, uint32_t j) {
    uint32_t h = seed ^ (o * 0x9E3779B1u);
    h = (h ^ i) * 0x85EBCA6Bu;
    h ^= h >> 13;
    h = (h ^ j) * 0xC2B2AE35u;
    h ^= h >> 16;
    h *= 0x27D4EB2Fu;
    h ^= h >> 15;
    return h;
}

inline uint32_t heightfield_height(uint32_t depth, uint32_t seed, uint32_t x, uint32_t z) {
    const uint32_t n = 1u << depth;
    uint64_t height = 1;
    for (uint32_t o = 0; o < 5; ++o) {
        uint32_t lambda = n >> (2 + o);
        if (lambda < 1) lambda = 1;
        const uint32_t amp = (n / 8) >> o;
        const uint32_t i = x / lambda, j = z / lambda;
        const uint64_t fx = x % lambda, fz = z % lambda;
        const uint64_t v00 = scene_hash32(seed, o, i, j) & 0xffffu, v10 = scene_hash32(seed, o, i + 1, j) & 0xffffu;
        const uint64_t v01 = scene_hash32(seed, o, i, j + 1) & 0xffffu, v11 = scene_hash32(seed, o, i + 1, j + 1) & 0xffffu;
        const uint64_t a = v00 * (lambda - fx) + v10 * fx;
        const uint64_t b = v01 * (lambda - fx) + v11 * fx;
        const uint64_t v = (a * (lambda - fz) + b * fz) / (uint64_t(lambda) * lambda);  // 0..65535
        height += (v * amp) >> 16;
    }
    const uint64_t cap = n / 4 ? n / 4 : 1;
    if (height > cap) height = cap;
    if (height < 1) height = 1;
    return uint32_t(height);
}

namespace detail {
inline EsvoSerializedChunk make_serialized(const Chunk& c, const Esvo<EsvoSerializedChunk>*) { return EsvoSerializedChunk(c); }
inline CsvoSerializedChunk make_serialized(const Chunk& c, const Csvo*) { return CsvoSerializedChunk(c); }
}  // namespace detail

// Builds every non-empty chunk of the scene (worker threads serialize chunks, as the reference does on its
// job system, src/systems/worldsvo.rs:90-99), places them in `world` and serializes the world.
template <class WorldT>
SceneStats build_heightfield_scene(WorldT& world, uint32_t depth, uint32_t seed, uint32_t n_threads) {
    using SerializedT = decltype(detail::make_serialized(std::declval<const Chunk&>(), static_cast<const WorldT*>(nullptr)));
    struct Built {
        Position pos;
        SerializedT chunk;
    };
    const uint32_t n = 1u << depth;
    const uint32_t chunks_per_axis = depth > 5 ? 1u << (depth - 5) : 1u;
    const uint32_t edge = n < 32 ? n : 32;  // voxels per chunk edge actually inside the domain
    if (n_threads == 0) n_threads = 1;

    std::vector<std::vector<Built>> per_thread(n_threads);
    std::vector<uint64_t> leaves(n_threads, 0);
    std::vector<uint32_t> hmax(n_threads, 0);
    std::atomic<uint32_t> next_column{0};

    auto worker = [&](uint32_t tid) {
        std::vector<uint32_t> h(34 * 34);
        for (;;) {
            const uint32_t col = next_column.fetch_add(1);
            if (col >= chunks_per_axis * chunks_per_axis) break;
            const uint32_t cx = col % chunks_per_axis, cz = col / chunks_per_axis;
            for (int dz = -1; dz <= 32; ++dz)
                for (int dx = -1; dx <= 32; ++dx) {
                    int64_t wx = int64_t(cx) * 32 + dx, wz = int64_t(cz) * 32 + dz;
                    wx = std::clamp<int64_t>(wx, 0, int64_t(n) - 1);
                    wz = std::clamp<int64_t>(wz, 0, int64_t(n) - 1);
                    h[(dz + 1) * 34 + (dx + 1)] = heightfield_height(depth, seed, uint32_t(wx), uint32_t(wz));
                }
            uint32_t lo_min = UINT32_MAX, hi_max = 0;
            std::vector<uint32_t> ylo(32 * 32), yhi(32 * 32);
            for (uint32_t z = 0; z < edge; ++z)
                for (uint32_t x = 0; x < edge; ++x) {
                    const uint32_t c = h[(z + 1) * 34 + (x + 1)];
                    const uint32_t m = std::min(std::min(h[(z + 1) * 34 + x], h[(z + 1) * 34 + x + 2]), std::min(h[z * 34 + x + 1], h[(z + 2) * 34 + x + 1]));
                    const uint32_t lo = std::min(c, m + 1);
                    ylo[z * 32 + x] = lo;
                    yhi[z * 32 + x] = c;
                    lo_min = std::min(lo_min, lo);
                    hi_max = std::max(hi_max, c);
                }
            hmax[tid] = std::max(hmax[tid], hi_max);
            for (uint32_t cy = lo_min / 32; cy <= hi_max / 32 && cy < chunks_per_axis; ++cy) {
                Chunk chunk(ChunkPos{int32_t(cx), int32_t(cy), int32_t(cz)}, 5);
                uint64_t count = 0;
                for (uint32_t z = 0; z < edge; ++z)
                    for (uint32_t x = 0; x < edge; ++x) {
                        const uint32_t top = yhi[z * 32 + x];
                        const uint32_t a = std::max(ylo[z * 32 + x], cy * 32), b = std::min(top, cy * 32 + 31);
                        for (uint32_t wy = a; wy <= b && a <= b; ++wy) {
                            const BlockId id = wy >= top ? 1u : (wy + 3 >= top ? 2u : 3u);  // grass / dirt / stone
                            chunk.storage.set_leaf(Position{x, wy - cy * 32, z}, id);
                            ++count;
                        }
                    }
                if (!count) continue;
                chunk.storage.compact();
                leaves[tid] += count;
                per_thread[tid].push_back(Built{Position{cx, cy, cz}, detail::make_serialized(chunk, static_cast<const WorldT*>(nullptr))});
            }
        }
    };

    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& t : pool) t.join();

    // deterministic placement order regardless of thread scheduling: sort by (z, x, y)
    std::vector<Built> all;
    for (auto& v : per_thread) {
        for (auto& b : v) all.push_back(std::move(b));
        v.clear();
    }
    std::sort(all.begin(), all.end(), [](const Built& a, const Built& b) {
        if (a.pos.z != b.pos.z) return a.pos.z < b.pos.z;
        if (a.pos.x != b.pos.x) return a.pos.x < b.pos.x;
        return a.pos.y < b.pos.y;
    });

    SceneStats st;
    for (auto& b : all) {
        world.set_leaf(b.pos, std::move(b.chunk), true);
        ++st.chunks;
    }
    world.serialize();
    for (uint32_t t = 0; t < n_threads; ++t) {
        st.leaves += leaves[t];
        st.h_max = std::max(st.h_max, hmax[t]);
    }
    return st;
}

}  // namespace vx
