// C entry points over the C++ host mirror (chunks, world SVOs), for the Python harness (tests, bench, smoke).
// Nothing here is on the render path: it builds the same serialized buffers the reference's Rust host code
// hands to `graphics::Svo::update` (src/graphics/svo.rs:171-189).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <new>

#include "chunk.hpp"
#include "csvo.hpp"
#include "esvo.hpp"
#include "scene.hpp"

using namespace vx;

namespace {

struct World {
    int svo_type;  // 1 = ESVO, 2 = CSVO (svo.rs:35-36)
    Esvo<EsvoSerializedChunk> esvo;
    Csvo csvo;
};

}  // namespace

extern "C" {

// ---- chunks -------------------------------------------------------------------------------------------

void* vxh_chunk_new(int32_t x, int32_t y, int32_t z, uint32_t lod) { return new (std::nothrow) Chunk(ChunkPos{x, y, z}, uint8_t(lod)); }
void vxh_chunk_free(void* c) { delete static_cast<Chunk*>(c); }
void vxh_chunk_set_block(void* c, uint32_t x, uint32_t y, uint32_t z, uint32_t id) { static_cast<Chunk*>(c)->set_block(x, y, z, id); }
uint32_t vxh_chunk_get_block(void* c, uint32_t x, uint32_t y, uint32_t z) { return static_cast<Chunk*>(c)->get_block(x, y, z); }
void vxh_chunk_compact(void* c) { static_cast<Chunk*>(c)->storage.compact(); }
void vxh_chunk_set_lod(void* c, uint32_t lod) { static_cast<Chunk*>(c)->lod = uint8_t(lod); }

// ids[x + 32 * (y + 32 * z)], 0 = empty; rebuilds the storage bottom-up like Chunk::fill_with
void vxh_chunk_fill_dense(void* c, const uint32_t* ids) {
    static_cast<Chunk*>(c)->fill_with([ids](uint32_t x, uint32_t y, uint32_t z) -> std::optional<BlockId> {
        const uint32_t v = ids[x + 32u * (y + 32u * z)];
        if (v == NO_BLOCK) return std::nullopt;
        return v;
    });
}

uint64_t vxh_chunk_pos_hash(int32_t x, int32_t y, int32_t z) { return chunk_pos_hash(ChunkPos{x, y, z}); }

// ---- world SVO (world::hds::WorldSvo, src/world/hds/common.rs:3-15) -------------------------------------

void* vxh_world_new(int svo_type) {
    if (svo_type != 1 && svo_type != 2) return nullptr;
    World* w = new (std::nothrow) World();
    if (w) w->svo_type = svo_type;
    return w;
}
void vxh_world_free(void* w) { delete static_cast<World*>(w); }

// serializes the chunk (SerializedChunk::new) and places it at the given SVO-space chunk position
int vxh_world_set_chunk(void* wp, uint32_t px, uint32_t py, uint32_t pz, const void* chunk, int serialize) {
    World* w = static_cast<World*>(wp);
    const Chunk& c = *static_cast<const Chunk*>(chunk);
    if (w->svo_type == 1) w->esvo.set_leaf(Position{px, py, pz}, EsvoSerializedChunk(c), serialize != 0);
    else w->csvo.set_leaf(Position{px, py, pz}, CsvoSerializedChunk(c), serialize != 0);
    return 0;
}

void vxh_world_serialize(void* wp) {
    World* w = static_cast<World*>(wp);
    if (w->svo_type == 1) w->esvo.serialize();
    else w->csvo.serialize();
}

uint32_t vxh_world_depth(const void* wp) {
    const World* w = static_cast<const World*>(wp);
    return w->svo_type == 1 ? w->esvo.depth() : w->csvo.depth();
}

size_t vxh_world_size_in_bytes(const void* wp) {
    const World* w = static_cast<const World*>(wp);
    return w->svo_type == 1 ? w->esvo.size_in_bytes() : w->csvo.size_in_bytes();
}

// bytes of the header the writer puts in front of the arena (ESVO preamble 20, CSVO root pointer 4)
size_t vxh_world_header_bytes(const void* wp) { return static_cast<const World*>(wp)->svo_type == 1 ? 20 : 4; }

size_t vxh_world_write_to(const void* wp, uint8_t* dst) {
    const World* w = static_cast<const World*>(wp);
    return w->svo_type == 1 ? w->esvo.write_to(dst) : w->csvo.write_to(dst);
}

int vxh_world_write_changes_to(void* wp, uint8_t* dst, size_t dst_len, int reset) {
    World* w = static_cast<World*>(wp);
    const bool ok = w->svo_type == 1 ? w->esvo.write_changes_to(dst, dst_len, reset != 0) : w->csvo.write_changes_to(dst, dst_len, reset != 0);
    return ok ? 0 : 1;
}

// copies up to `max` (start,length) pairs of the dirty ranges (arena offsets); returns the count
size_t vxh_world_updated_ranges(const void* wp, uint64_t* out_pairs, size_t max) {
    const World* w = static_cast<const World*>(wp);
    const std::vector<Range>& r = w->svo_type == 1 ? w->esvo.buffer.updated_ranges : w->csvo.buffer.updated_ranges;
    for (size_t i = 0; i < r.size() && i < max; ++i) {
        out_pairs[2 * i] = r[i].start;
        out_pairs[2 * i + 1] = r[i].length;
    }
    return r.size();
}

// The whole mapped-buffer image `graphics::Svo::update` produces: [f32 2^-depth][header][arena].
// Returns the bytes needed; writes only if `cap` suffices.
size_t vxh_world_frame(const void* wp, uint8_t* dst, size_t cap) {
    const World* w = static_cast<const World*>(wp);
    const size_t need = 4 + vxh_world_header_bytes(wp) + vxh_world_size_in_bytes(wp);
    if (!dst || cap < need) return need;
    const float scale = std::exp2(-float(vxh_world_depth(wp)));
    std::memcpy(dst, &scale, 4);
    if (w->svo_type == 1) w->esvo.write_to(dst + 4);
    else w->csvo.write_to(dst + 4);
    return need;
}

// ---- synthetic scenes (bench / parity at scale) --------------------------------------------

// Fills `world` with the seeded heightfield scene of the given depth; returns the number of chunks placed.
uint64_t vxh_scene_build_heightfield(void* wp, uint32_t depth, uint32_t seed, uint32_t n_threads, uint64_t* out_leaves, uint32_t* out_hmax) {
    World* w = static_cast<World*>(wp);
    SceneStats st;
    if (w->svo_type == 1) st = build_heightfield_scene(w->esvo, depth, seed, n_threads);
    else st = build_heightfield_scene(w->csvo, depth, seed, n_threads);
    if (out_leaves) *out_leaves = st.leaves;
    if (out_hmax) *out_hmax = st.h_max;
    return st.chunks;
}

uint32_t vxh_scene_height(uint32_t depth, uint32_t seed, uint32_t x, uint32_t z) { return heightfield_height(depth, seed, x, z); }

}  // extern "C"
