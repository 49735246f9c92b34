// World -> SVO-space mapper (src/systems/worldsvo.rs): keeps the camera's chunk at the centre of the SVO ("chunk
// shifting", worldsvo.rs:42-47,161-196), converts positions between world and SVO space for render / raycast
// (worldsvo.rs:397-435) and feeds serialized chunks to graphics::Svo. Chunk serialization runs synchronously here (the
// reference uses its job system, worldsvo.rs:90-99; out of scope).
#pragma once

#include <cmath>
#include <optional>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "chunk.hpp"
#include "graphics_svo.hpp"
#include "octree.hpp"

namespace vx {
namespace systems {

// worldsvo.rs:437-503
struct SvoCoordSpace {
    ChunkPos center;
    uint32_t dst = 0;  // render distance in chunks

    // world block position -> SVO position: re-base the chunk coordinate to dst + (chunk - center)
    Vec3 cnv_block_pos(Vec3 pos) const {
        BlockPos bp = BlockPos::from_point(pos.x, pos.y, pos.z);
        const ChunkPos delta = bp.chunk - center;
        const int32_t rd = int32_t(dst);
        bp.chunk = ChunkPos{rd + delta.x, rd + delta.y, rd + delta.z};
        float out[3];
        bp.to_point(out);
        return Vec3{out[0], out[1], out[2]};
    }

    Vec3 cnv_svo_pos(Vec3 pos) const {
        BlockPos bp = BlockPos::from_point(pos.x, pos.y, pos.z);
        const int32_t rd = int32_t(dst);
        const ChunkPos delta = bp.chunk - ChunkPos{rd, rd, rd};
        bp.chunk = ChunkPos{center.x + delta.x, center.y + delta.y, center.z + delta.z};
        float out[3];
        bp.to_point(out);
        return Vec3{out[0], out[1], out[2]};
    }

    // chunk position -> SVO chunk position, none outside the cylinder of radius dst (full height dst up and down)
    std::optional<Position> cnv_chunk_pos(ChunkPos pos) const {
        const float r = float(dst);
        auto times32 = [](int32_t c) { return float(int32_t(uint32_t(c) << 5)); };  // (a negative coordinate shifted left is undefined before C++20)
        const Vec3 p = cnv_block_pos(Vec3{times32(pos.x), times32(pos.y), times32(pos.z)});
        const float x = p.x / 32.0f, y = p.y / 32.0f, z = p.z / 32.0f;
        const float dcy = y - r;
        if (dcy < -r || dcy > r) return std::nullopt;
        const float dcx = x - r, dcz = z - r;
        if (std::fma(dcx, dcx, dcz * dcz) > r * r) return std::nullopt;
        return Position{uint32_t(x), uint32_t(y), uint32_t(z)};
    }
};

// worldsvo.rs:161-196. WorldT needs set_leaf(pos, leaf, serialize) / move_leaf(id, pos) / remove_leaf(id).
template <class WorldT>
void shift_chunks(const SvoCoordSpace& cs, std::unordered_map<ChunkPos, LeafId, ChunkPosHash>& leaf_ids, WorldT& world) {
    using LeafT = typename decltype(world.remove_leaf(LeafId{}))::value_type;
    struct LeafIdHash { size_t operator()(const LeafId& l) const { return size_t(l.parent) * 8 + l.idx; } };
    std::unordered_map<LeafId, LeafT, LeafIdHash> overridden;
    std::vector<ChunkPos> removed;
    for (auto& kv : leaf_ids) {
        LeafId& leaf_id = kv.second;
        const std::optional<Position> np = cs.cnv_chunk_pos(kv.first);
        if (!np) {
            if (!overridden.count(leaf_id)) world.remove_leaf(leaf_id);
            overridden.erase(leaf_id);
            removed.push_back(kv.first);
            continue;
        }
        std::pair<LeafId, std::optional<LeafT>> r;
        auto it = overridden.find(leaf_id);
        if (it != overridden.end()) {
            LeafT value = std::move(it->second);
            overridden.erase(it);
            r = world.set_leaf(*np, std::move(value), false);  // moved only: try to bypass re-serialization
        } else {
            r = world.move_leaf(leaf_id, *np);
        }
        leaf_id = r.first;
        if (r.second) overridden.emplace(r.first, std::move(*r.second));
    }
    for (const ChunkPos& p : removed) leaf_ids.erase(p);
}

// worldsvo.rs:48-224, 390-435
template <class WorldT, class SerializedT>
class Svo {
public:
    Svo(graphics::Svo& gfx, uint32_t render_distance) : gfx_(gfx) { cs_.dst = render_distance; }

    void set_chunk(const Chunk& chunk) {
        SerializedT sc(chunk);
        const std::optional<Position> p = cs_.cnv_chunk_pos(chunk.pos);
        if (!p) return;
        auto r = world_.set_leaf(*p, std::move(sc), true);
        leaf_ids_[chunk.pos] = r.first;
        has_changed_ = true;
    }

    void remove_chunk(ChunkPos pos) {
        auto it = leaf_ids_.find(pos);
        if (it == leaf_ids_.end()) return;
        world_.remove_leaf(it->second);
        leaf_ids_.erase(it);
        has_changed_ = true;
    }

    // worldsvo.rs:133-151
    void update(ChunkPos world_center) {
        if (cs_.center != world_center) {
            cs_.center = world_center;
            has_changed_ = true;
            shift_chunks(cs_, leaf_ids_, world_);
        }
        if (!has_changed_) return;
        has_changed_ = false;
        world_.serialize();
        graphics::WorldSvoRef<WorldT> ref(world_);
        gfx_.update(ref);
    }

    void render(graphics::RenderParams params, graphics::Framebuffer& target) const {
        params.cam_pos = cs_.cnv_block_pos(params.cam_pos);
        if (params.selected_voxel) params.selected_voxel = cs_.cnv_block_pos(*params.selected_voxel);
        gfx_.render(params, target);
    }

    // Raycaster impl (worldsvo.rs:419-435): world space in, world space out
    void raycast(PickerBatch& batch, PickerBatchResult& result) const {
        for (Ray& r : batch.rays) r.pos = cs_.cnv_block_pos(r.pos);
        for (Aabb& a : batch.aabbs) a.pos = cs_.cnv_block_pos(a.pos);
        gfx_.raycast(batch, result);
        for (RayResult& r : result.rays) r.pos = cs_.cnv_svo_pos(r.pos);
    }

    graphics::Stats get_stats() const { return gfx_.get_stats(); }
    WorldT& world() { return world_; }
    const SvoCoordSpace& coord_space() const { return cs_; }

private:
    graphics::Svo& gfx_;
    WorldT world_;
    std::unordered_map<ChunkPos, LeafId, ChunkPosHash> leaf_ids_;
    bool has_changed_ = false;
    SvoCoordSpace cs_;
};

}  // namespace systems
}  // namespace vx
