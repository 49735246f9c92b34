// 32^3 voxel chunk and its coordinate types (mirror of src/world/chunk.rs:94-297).
#pragma once

#include <cmath>
#include <cstdint>
#include <functional>

#include "octree.hpp"

namespace vx {

using BlockId = uint32_t;
constexpr BlockId NO_BLOCK = 0;
using ChunkStorage = Octree<BlockId>;

// Chunk coordinates: one step = 32 blocks (chunk.rs:138-181).
struct ChunkPos {
    int32_t x = 0, y = 0, z = 0;
    static ChunkPos from_block_pos(int32_t bx, int32_t by, int32_t bz) { return {bx >> 5, by >> 5, bz >> 5}; }
    bool operator==(const ChunkPos& o) const { return x == o.x && y == o.y && z == o.z; }
    bool operator!=(const ChunkPos& o) const { return !(*this == o); }
    ChunkPos operator-(const ChunkPos& o) const { return {x - o.x, y - o.y, z - o.z}; }
    float dst_sq(const ChunkPos& o) const {
        const float dx = float(o.x - x), dy = float(o.y - y), dz = float(o.z - z);
        return std::fma(dz, dz, std::fma(dx, dx, dy * dy));
    }
    float dst_2d_sq(const ChunkPos& o) const {
        const float dx = float(o.x - x), dz = float(o.z - z);
        return std::fma(dx, dx, dz * dz);
    }
};

struct ChunkPosHash {
    size_t operator()(const ChunkPos& p) const {
        uint64_t h = 0x9E3779B97F4A7C15ull;
        for (int32_t v : {p.x, p.y, p.z}) h = (h ^ uint32_t(v)) * 0x100000001B3ull + (h >> 29);
        return size_t(h);
    }
};

// Block position split into chunk + in-chunk offset; negative coordinates wrap into the chunk
// (x = -1 is block 31 of chunk -1), chunk.rs:251-297.
struct BlockPos {
    ChunkPos chunk;
    float rel_x = 0, rel_y = 0, rel_z = 0;

    static BlockPos from_ints(int32_t x, int32_t y, int32_t z) {
        return {ChunkPos::from_block_pos(x, y, z), float(x & 31), float(y & 31), float(z & 31)};
    }

    static BlockPos from_point(float px, float py, float pz) {
        const int32_t x = int32_t(std::floor(px)), y = int32_t(std::floor(py)), z = int32_t(std::floor(pz));
        // Rust's f32::fract keeps the sign of the input (x - trunc(x)).
        float fx = px - std::trunc(px), fy = py - std::trunc(py), fz = pz - std::trunc(pz);
        if (fx != 0.0f && px < 0.0f) fx += 1.0f;
        if (fy != 0.0f && py < 0.0f) fy += 1.0f;
        if (fz != 0.0f && pz < 0.0f) fz += 1.0f;
        return {ChunkPos::from_block_pos(x, y, z), float(x & 31) + fx, float(y & 31) + fy, float(z & 31) + fz};
    }

    void to_point(float out[3]) const {
        // (the shift on the unsigned value: a negative chunk coordinate shifted left is undefined before C++20 -- UBSan, make sanitize)
        const int32_t bx = int32_t(uint32_t(chunk.x) << 5) | (int32_t(rel_x) & 31);
        const int32_t by = int32_t(uint32_t(chunk.y) << 5) | (int32_t(rel_y) & 31);
        const int32_t bz = int32_t(uint32_t(chunk.z) << 5) | (int32_t(rel_z) & 31);
        out[0] = float(bx) + (rel_x - std::trunc(rel_x));
        out[1] = float(by) + (rel_y - std::trunc(rel_y));
        out[2] = float(bz) + (rel_z - std::trunc(rel_z));
    }
};

// A chunk owns a depth-5 octree of block ids. Fresh storage is pre-expanded to depth 5 exactly like
// the reference's pooled storage (chunk.rs:27-33), i.e. it starts as a chain of five empty octants.
struct Chunk {
    ChunkPos pos;
    uint8_t lod = 5;  // 5 = full detail (chunk.rs:96-98)
    ChunkStorage storage;

    Chunk() { storage.expand_to(5); }
    Chunk(ChunkPos p, uint8_t l) : pos(p), lod(l) { storage.expand_to(5); }

    BlockId get_block(uint32_t x, uint32_t y, uint32_t z) const {
        const BlockId* v = storage.get_leaf(Position{x, y, z});
        return v ? *v : NO_BLOCK;
    }

    void set_block(uint32_t x, uint32_t y, uint32_t z, BlockId block) {
        if (block == NO_BLOCK) storage.remove_leaf(Position{x, y, z});
        else storage.set_leaf(Position{x, y, z}, block);
    }

    // rebuilds the storage bottom-up from a voxel function (chunk.rs:124-130)
    void fill_with(const std::function<std::optional<BlockId>(uint32_t, uint32_t, uint32_t)>& f) {
        storage.construct_octants_with(5, [&](Position p) { return f(p.x, p.y, p.z); });
    }
};

// SipHash-1-3 with zero keys over the three little-endian i32 coordinates: the value Rust's
// `DefaultHasher` yields for `#[derive(Hash)] ChunkPos`, which the reference uses as the chunk's unique id
// (src/world/hds/esvo.rs:357-360, csvo.rs:404-407). Only a map key, but reproducing it keeps ids comparable
// with the reference's tests (csvo.rs:376: ChunkPos(0,0,0) -> 2435999049025295583).
inline uint64_t chunk_pos_hash(const ChunkPos& p) {
    auto rotl = [](uint64_t v, int b) { return (v << b) | (v >> (64 - b)); };
    uint64_t v0 = 0x736f6d6570736575ull, v1 = 0x646f72616e646f6dull, v2 = 0x6c7967656e657261ull, v3 = 0x7465646279746573ull;
    auto round = [&]() {
        v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
        v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
        v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
        v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
    };
    const uint64_t m0 = uint64_t(uint32_t(p.x)) | (uint64_t(uint32_t(p.y)) << 32);
    v3 ^= m0; round(); v0 ^= m0;
    const uint64_t tail = uint64_t(uint32_t(p.z)) | (uint64_t(12) << 56);
    v3 ^= tail; round(); v0 ^= tail;
    v2 ^= 0xff;
    round(); round(); round();
    return v0 ^ v1 ^ v2 ^ v3;
}

}  // namespace vx
