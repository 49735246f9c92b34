// World streaming in front of the device buffer: chunk loader events -> generated chunks at their LOD -> SVO leaves ->
// dirty ranges -> vx_commit. What systems::worldsvo::Svo::update does with the chunk loader's output in the reference
// (src/systems/worldsvo.rs:133-196 with src/gamelogic/world.rs:132-262 as the producer), on a raw vx_context:
//   move_to(pos)   chunk loader events for the new position are queued (nearest first); a new centre chunk re-bases the
//                  SVO coordinate space and shifts the resident chunks (move_leaf, no re-serialization)
//   pump(ctx, n)   applies up to n queued events (the reference takes at most 400 serialized chunks per frame,
//                  worldsvo.rs:139), re-serializes the root, writes only the changed ranges into the pinned staging mirror
//                  and commits them: the H2D copies are ordered after in-flight frames by an event, not by a CPU stall
// Terrain is the build's integer value-noise heightfield (scene.hpp), evaluated per chunk in world coordinates.
#pragma once

#include <atomic>
#include <chrono>
#include <deque>
#include <functional>
#include <optional>
#include <stdexcept>
#include <thread>
#include <unordered_map>
#include <vector>

#include "camera.hpp"
#include "chunkloader.hpp"
#include "scene.hpp"
#include "voxel_hip.h"
#include "worldsvo.hpp"

namespace vx {
namespace systems {

// One chunk of the heightfield scene at world chunk position `pos` (surface shell only, ids as scene.hpp); nullopt when the
// chunk holds no voxel or lies outside the [0, 2^depth) domain.
inline std::optional<Chunk> generate_heightfield_chunk(uint32_t depth, uint32_t seed, ChunkPos pos, uint8_t lod) {
    const int64_t n = int64_t(1) << depth;
    if (pos.x < 0 || pos.y < 0 || pos.z < 0 || int64_t(pos.x) * 32 >= n || int64_t(pos.y) * 32 >= n || int64_t(pos.z) * 32 >= n) return std::nullopt;
    uint32_t h[34 * 34];
    for (int dz = -1; dz <= 32; ++dz)
        for (int dx = -1; dx <= 32; ++dx) {
            const int64_t wx = std::clamp<int64_t>(int64_t(pos.x) * 32 + dx, 0, n - 1), wz = std::clamp<int64_t>(int64_t(pos.z) * 32 + dz, 0, n - 1);
            h[(dz + 1) * 34 + (dx + 1)] = heightfield_height(depth, seed, uint32_t(wx), uint32_t(wz));
        }
    Chunk chunk(pos, lod);
    const uint32_t y0 = uint32_t(pos.y) * 32;
    uint64_t count = 0;
    for (uint32_t z = 0; z < 32; ++z)
        for (uint32_t x = 0; x < 32; ++x) {
            const uint32_t c = h[(z + 1) * 34 + (x + 1)];
            const uint32_t m = std::min(std::min(h[(z + 1) * 34 + x], h[(z + 1) * 34 + x + 2]), std::min(h[z * 34 + x + 1], h[(z + 2) * 34 + x + 1]));
            const uint32_t lo = std::min(c, m + 1);  // the shell: down to one above the lowest neighbour column
            const uint32_t a = std::max(lo, y0), b = std::min(c, y0 + 31);
            for (uint32_t wy = a; a <= b && wy <= b; ++wy) {
                const BlockId id = wy >= c ? 1u : (wy + 3 >= c ? 2u : 3u);  // grass / dirt / stone
                chunk.storage.set_leaf(Position{x, wy - y0, z}, id);
                ++count;
            }
        }
    if (!count) return std::nullopt;
    chunk.storage.compact();
    return chunk;
}

struct PumpStats {
    uint32_t events = 0, loads = 0, unloads = 0, lod_changes = 0;
    uint32_t ranges = 0;        // dirty ranges handed to vx_commit
    uint64_t bytes = 0;         // their total length
    uint64_t arena_bytes = 0;   // size_in_bytes() after the commit
    uint32_t depth = 0;
    uint32_t pending = 0;       // events still queued
    double build_ms = 0, apply_ms = 0, commit_ms = 0;  // host time: chunk generation + serialization (workers), set_leaf / root, staging write + vx_commit
};

template <class WorldT, class SerializedT>
class WorldStreamer {
public:
    WorldStreamer(uint32_t scene_depth, uint32_t seed, uint32_t radius, int32_t start_y, int32_t end_y, uint32_t threads = 0)
        : scene_depth_(scene_depth), seed_(seed), loader_(radius, start_y, end_y) {
        cs_.dst = radius;
        threads_ = threads ? threads : std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    }

    // pump(nullptr, ..) hands the dirty ranges here instead of to vx_commit (tests)
    std::function<void(WorldT&, const std::vector<vx_range>&)> on_dry_commit;

    // returns the number of events the move produced
    // camera (optional): the view the events of this move are ordered by; its position is (x, y, z)
    size_t move_to(float x, float y, float z, const graphics::Camera* camera = nullptr) {
        const ChunkPos centre = ChunkPos::from_block_pos(int32_t(std::floor(x)), int32_t(std::floor(y)), int32_t(std::floor(z)));
        if (!(centre == cs_.center) || !has_centre_) {
            cs_.center = centre;
            has_centre_ = true;
            shift_chunks(cs_, leaf_ids_, world_);  // worldsvo.rs:161-196
            dirty_ = true;
        }
        std::vector<ChunkEvent> events = loader_.update(x, y, z);
        // what the player looks at first, the rest from front to back (src/gamelogic/world.rs:132-137)
        if (camera) events = sort_chunks_by_view_frustum(events, *camera);
        for (const ChunkEvent& e : events) queue_.push_back(e);
        return events.size();
    }

    PumpStats pump(vx_context* ctx, uint32_t max_events) {
        PumpStats st;
        using clock = std::chrono::steady_clock;
        auto ms_since = [](clock::time_point t0) { return std::chrono::duration<double, std::milli>(clock::now() - t0).count(); };
        clock::time_point t0 = clock::now();
        // take the batch, build its chunks on worker threads (the reference serializes chunks on its job system,
        // worldsvo.rs:90-99), then apply the results in event order on this thread
        std::vector<ChunkEvent> batch;
        while (!queue_.empty() && batch.size() < max_events) {
            batch.push_back(queue_.front());
            queue_.pop_front();
        }
        std::vector<std::optional<Position>> where(batch.size());
        std::vector<std::optional<SerializedT>> built(batch.size());
        for (size_t i = 0; i < batch.size(); ++i)
            if (batch[i].kind != ChunkEvent::Unload) where[i] = cs_.cnv_chunk_pos(batch[i].pos);  // none: outside the cylinder by now
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < batch.size();) {
                if (!where[i]) continue;
                std::optional<Chunk> chunk = generate_heightfield_chunk(scene_depth_, seed_, batch[i].pos, batch[i].lod);
                if (chunk) built[i].emplace(*chunk);
            }
        };
        const uint32_t n_workers = std::min<uint32_t>(threads_, uint32_t(batch.size() / 8 + 1));
        std::vector<std::thread> pool;
        for (uint32_t t = 1; t < n_workers; ++t) pool.emplace_back(worker);
        worker();
        for (std::thread& t : pool) t.join();
        st.build_ms = ms_since(t0);
        t0 = clock::now();

        for (size_t i = 0; i < batch.size(); ++i) {
            const ChunkEvent& e = batch[i];
            ++st.events;
            if (e.kind == ChunkEvent::Unload) {
                ++st.unloads;
                remove(e.pos);
                continue;
            }
            if (e.kind == ChunkEvent::Load) ++st.loads; else ++st.lod_changes;
            if (!where[i]) continue;
            if (!built[i]) {
                remove(e.pos);  // an LOD change of a chunk that has nothing to show
                continue;
            }
            auto r = world_.set_leaf(*where[i], std::move(*built[i]), true);
            leaf_ids_[e.pos] = r.first;
            dirty_ = true;
        }
        st.pending = uint32_t(queue_.size());
        if (dirty_) {
            dirty_ = false;
            world_.serialize();
            st.apply_ms = ms_since(t0);
            t0 = clock::now();
            std::vector<vx_range> ranges;
            for (const Range& r : world_.buffer.updated_ranges) {
                ranges.push_back(vx_range{r.start, r.length});  // RangeBuffer counts bytes for both formats here
                st.bytes += r.length;
            }
            st.ranges = uint32_t(ranges.size());
            if (ctx) {
                if (!world_.write_changes_to(vx_staging_ptr(ctx) + 4, vx_arena_capacity(ctx), true)) throw std::runtime_error("world buffer capacity exceeded");
                if (vx_commit(ctx, world_.depth(), ranges.data(), uint32_t(ranges.size()), world_.size_in_bytes()) != VX_OK) throw std::runtime_error(vx_last_error());
            } else if (on_dry_commit) {
                on_dry_commit(world_, ranges);  // host-only tests: someone else plays the device (must consume the ranges)
                world_.buffer.updated_ranges.clear();
            } else {
                world_.buffer.updated_ranges.clear();  // dry run (host-only tests): the world is updated, nothing is uploaded
            }
            st.commit_ms = ms_since(t0);
        }
        st.arena_bytes = world_.size_in_bytes();
        st.depth = world_.depth();
        return st;
    }

    WorldT& world() { return world_; }
    const SvoCoordSpace& coord_space() const { return cs_; }
    size_t resident_chunks() const { return leaf_ids_.size(); }
    size_t pending_events() const { return queue_.size(); }

private:
    void remove(ChunkPos pos) {
        auto it = leaf_ids_.find(pos);
        if (it == leaf_ids_.end()) return;
        world_.remove_leaf(it->second);
        leaf_ids_.erase(it);
        dirty_ = true;
    }

    uint32_t scene_depth_, seed_, threads_ = 1;
    ChunkLoader loader_;
    SvoCoordSpace cs_;
    bool has_centre_ = false, dirty_ = false;
    WorldT world_;
    std::unordered_map<ChunkPos, LeafId, ChunkPosHash> leaf_ids_;
    std::deque<ChunkEvent> queue_;
};

}  // namespace systems
}  // namespace vx
