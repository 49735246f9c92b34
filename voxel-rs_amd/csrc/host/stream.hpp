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
#include <condition_variable>
#include <memory>
#include <mutex>
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

// The terrain's heights under one (x, z) column of chunks and around it (34 x 34: the shell of a voxel looks at its four neighbours), shared by the
// chunks of every layer of the column, and the span of layers that hold any voxel at all.
struct ColumnHeights {
    uint32_t h[34 * 34];
    uint32_t lowest = 0, highest = 0;  // world y of the lowest and the highest voxel of the column's shell
};
inline void heightfield_column(uint32_t depth, uint32_t seed, int32_t cx, int32_t cz, ColumnHeights& out) {
    const int64_t n = int64_t(1) << depth;
    for (int dz = -1; dz <= 32; ++dz)
        for (int dx = -1; dx <= 32; ++dx) {
            const int64_t wx = std::clamp<int64_t>(int64_t(cx) * 32 + dx, 0, n - 1), wz = std::clamp<int64_t>(int64_t(cz) * 32 + dz, 0, n - 1);
            out.h[(dz + 1) * 34 + (dx + 1)] = heightfield_height(depth, seed, uint32_t(wx), uint32_t(wz));
        }
    out.lowest = UINT32_MAX;
    out.highest = 0;
    for (uint32_t z = 0; z < 32; ++z)
        for (uint32_t x = 0; x < 32; ++x) {
            const uint32_t* h = out.h;
            const uint32_t c = h[(z + 1) * 34 + (x + 1)];
            const uint32_t m = std::min(std::min(h[(z + 1) * 34 + x], h[(z + 1) * 34 + x + 2]), std::min(h[z * 34 + x + 1], h[(z + 2) * 34 + x + 1]));
            out.lowest = std::min(out.lowest, std::min(c, m + 1));
            out.highest = std::max(out.highest, c);
        }
}

// One chunk of the heightfield scene at world chunk position `pos` (surface shell only, ids as scene.hpp); nullopt when the
// chunk holds no voxel or lies outside the [0, 2^depth) domain. `column` (optional): the heights of the chunk's column, made once for all its layers.
inline std::optional<Chunk> generate_heightfield_chunk(uint32_t depth, uint32_t seed, ChunkPos pos, uint8_t lod, const ColumnHeights* column = nullptr) {
    const int64_t n = int64_t(1) << depth;
    if (pos.x < 0 || pos.y < 0 || pos.z < 0 || int64_t(pos.x) * 32 >= n || int64_t(pos.y) * 32 >= n || int64_t(pos.z) * 32 >= n) return std::nullopt;
    ColumnHeights own;
    if (!column) {
        heightfield_column(depth, seed, pos.x, pos.z, own);
        column = &own;
    }
    const uint32_t* h = column->h;
    const uint32_t y0 = uint32_t(pos.y) * 32;
    if (column->highest < y0 || column->lowest > y0 + 31) return std::nullopt;  // (most chunks of a column: air above, rock below)
    Chunk chunk(pos, lod);
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
    ~WorldStreamer() {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        cv_jobs_.notify_all();
        for (std::thread& t : pool_) t.join();
    }
    WorldStreamer(const WorldStreamer&) = delete;
    WorldStreamer& operator=(const WorldStreamer&) = delete;

    // --no-lod (src/gamelogic/world.rs:141,151): every chunk is loaded at full detail and LOD changes are ignored
    bool no_lod = false;

    // pump(nullptr, ..) hands the dirty ranges here instead of to vx_commit (tests)
    std::function<void(WorldT&, const std::vector<vx_range>&)> on_dry_commit;

    // returns the number of events the move produced
    // camera (optional): the view the events of this move are ordered by; its position is (x, y, z)
    size_t move_to(float x, float y, float z, const graphics::Camera* camera = nullptr) {
        using clock = std::chrono::steady_clock;
        auto ms_since = [](clock::time_point t0) { return std::chrono::duration<double, std::milli>(clock::now() - t0).count(); };
        clock::time_point t0 = clock::now();
        const ChunkPos centre = ChunkPos::from_block_pos(int32_t(std::floor(x)), int32_t(std::floor(y)), int32_t(std::floor(z)));
        if (!(centre == cs_.center) || !has_centre_) {
            cs_.center = centre;
            has_centre_ = true;
            shift_chunks(cs_, leaf_ids_, world_);  // worldsvo.rs:161-196
            dirty_ = true;
        }
        last_move_ms[0] = ms_since(t0);
        t0 = clock::now();
        std::vector<ChunkEvent> events = loader_.update(x, y, z);
        // what the player looks at first, the rest from front to back (src/gamelogic/world.rs:132-137)
        if (camera) events = sort_chunks_by_view_frustum(events, *camera);
        last_move_ms[1] = ms_since(t0);
        t0 = clock::now();
        // Every event gets a slot in the queue; chunks are generated and serialized in the background from now on (the reference
        // hands them to its job system as soon as they are known, worldsvo.rs:90-99) and applied, in event order, by pump().
        {
            // (a deque's elements stay where they are while others are pushed behind and popped in front of them: the workers hold plain pointers;
            // a re-centre at radius 40 queues 25 K events, and an allocation per event was a millisecond of the frame loop's step)
            std::lock_guard<std::mutex> lock(m_);
            for (const ChunkEvent& e : events) {
                if (no_lod && e.kind == ChunkEvent::LodChange) continue;  // (world.rs:151)
                queue_.emplace_back();
                Slot& slot = queue_.back();
                slot.event = e;
                if (no_lod && e.kind == ChunkEvent::Load) slot.event.lod = 5;  // (world.rs:141)
                if (e.kind == ChunkEvent::Unload) slot.ready = true;  // nothing to build
                else jobs_.push_back(&slot);
            }
        }
        if (pool_.empty() && !events.empty())
            for (uint32_t t = 0; t < threads_; ++t) pool_.emplace_back([this]() { build_loop(); });
        cv_jobs_.notify_all();
        last_move_ms[2] = ms_since(t0);
        return events.size();
    }
    // what the last move_to spent, in ms: shifting the resident chunks (a new centre chunk only), the chunk loader's events, queueing them
    double last_move_ms[3] = {0, 0, 0};

    // wait = true: the next max_events events are applied, whether their chunks are built yet or not (the call waits for them: what
    // the tests want, the same events per call whatever the machine). wait = false: only events whose chunks are ready are applied,
    // in order (what a frame loop wants: a chunk that is not built yet arrives with a later frame, like in the reference, whose
    // update() takes whatever its job system has finished, worldsvo.rs:139).
    PumpStats pump(vx_context* ctx, uint32_t max_events, bool wait = true) {
        PumpStats st;
        using clock = std::chrono::steady_clock;
        auto ms_since = [](clock::time_point t0) { return std::chrono::duration<double, std::milli>(clock::now() - t0).count(); };
        clock::time_point t0 = clock::now();
        std::vector<ChunkEvent> batch;
        std::vector<std::unique_ptr<SerializedT>> built;
        {
            std::unique_lock<std::mutex> lock(m_);
            while (!queue_.empty() && batch.size() < max_events) {
                Slot& slot = queue_.front();
                if (!slot.ready) {
                    if (!wait) break;
                    cv_done_.wait(lock, [&]() { return slot.ready; });
                }
                batch.push_back(slot.event);
                built.push_back(std::move(slot.built));
                if (slot.event.kind != ChunkEvent::Unload) --built_ahead_;
                queue_.pop_front();
            }
        }
        cv_jobs_.notify_all();  // (room for the workers to build further ahead)
        std::vector<std::optional<Position>> where(batch.size());
        for (size_t i = 0; i < batch.size(); ++i)
            if (batch[i].kind != ChunkEvent::Unload) where[i] = cs_.cnv_chunk_pos(batch[i].pos);  // none: outside the cylinder by now
        st.build_ms = ms_since(t0);  // (time spent waiting for chunks that were not built yet)
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
        {
            std::lock_guard<std::mutex> lock(m_);
            st.pending = uint32_t(queue_.size());
        }
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
    size_t pending_events() {
        std::lock_guard<std::mutex> lock(m_);
        return queue_.size();
    }

private:
    // one queued event and, once a worker is through with it, its serialized chunk (none: the chunk holds no voxel)
    struct Slot {
        ChunkEvent event;
        bool ready = false;
        std::unique_ptr<SerializedT> built;  // (a pointer: a re-centre queues 25 K slots, nine in ten of which never hold a chunk)
    };

    void build_loop() {
        for (;;) {
            Slot* slot = nullptr;  // (pump() pops a slot only once it is ready, i.e. after this thread's last touch of it)
            {
                std::unique_lock<std::mutex> lock(m_);
                // not too far ahead of pump(): finished chunks wait in memory until they are applied
                cv_jobs_.wait(lock, [&]() { return stop_ || (!jobs_.empty() && built_ahead_ < kMaxBuiltAhead); });
                if (stop_) return;
                slot = jobs_.front();
                jobs_.pop_front();
                ++built_ahead_;
            }
            std::unique_ptr<SerializedT> built;
            std::shared_ptr<const ColumnHeights> column = column_heights(slot->event.pos.x, slot->event.pos.z);
            std::optional<Chunk> chunk = generate_heightfield_chunk(scene_depth_, seed_, slot->event.pos, slot->event.lod, column.get());
            if (chunk) built = std::make_unique<SerializedT>(*chunk);
            {
                std::lock_guard<std::mutex> lock(m_);
                slot->built = std::move(built);
                slot->ready = true;
            }
            cv_done_.notify_all();
        }
    }

    // The heights of a column of chunks, evaluated once for its (up to 81) layers: a re-centre's events are mostly layers of the same few hundred
    // columns, nine in ten of them air. Kept for the most recently used columns.
    std::shared_ptr<const ColumnHeights> column_heights(int32_t cx, int32_t cz) {
        const uint64_t key = (uint64_t(uint32_t(cx)) << 32) | uint32_t(cz);
        {
            std::lock_guard<std::mutex> lock(columns_m_);
            auto it = columns_.find(key);
            if (it != columns_.end()) return it->second;
        }
        auto made = std::make_shared<ColumnHeights>();
        heightfield_column(scene_depth_, seed_, cx, cz, *made);  // (outside the lock; two workers may make the same column once in a while)
        std::lock_guard<std::mutex> lock(columns_m_);
        if (columns_.size() >= kMaxColumns) columns_.clear();  // (crude: the window moves on, what was made for the old one goes)
        return columns_.emplace(key, std::move(made)).first->second;
    }

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
    // events in arrival order; jobs_ = those of them no worker has taken yet
    std::deque<Slot> queue_;
    std::deque<Slot*> jobs_;
    static constexpr size_t kMaxBuiltAhead = 16384;
    size_t built_ahead_ = 0;  // chunks taken by workers and not applied yet
    std::mutex m_;
    std::mutex columns_m_;
    std::unordered_map<uint64_t, std::shared_ptr<const ColumnHeights>> columns_;
    static constexpr size_t kMaxColumns = 16384;
    std::condition_variable cv_jobs_, cv_done_;
    std::vector<std::thread> pool_;
    bool stop_ = false;
};

}  // namespace systems
}  // namespace vx
