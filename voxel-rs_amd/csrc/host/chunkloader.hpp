// Which chunks should be resident around a moving position, at which level of detail, and in which order to fetch
// them: src/systems/chunkloader.rs. This is the producer in front of the SVO mapper (worldsvo.hpp): its Load / LodChange
// events become serialized chunks and, through vx_commit, dirty ranges of the device world buffer; the LOD of a chunk
// is the depth the traversal descends to inside it (CSVO: `depth = lod` at the chunk boundary, svo.csvo.glsl:402-409).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "chunk.hpp"

namespace vx {
namespace systems {

// chunkloader.rs:17-30. Ordering = the derived Ord of the reference's enum: variant (Load < Unload < LodChange), then
// position (x, y, z), then lod.
struct ChunkEvent {
    enum Kind : uint8_t { Load = 0, Unload = 1, LodChange = 2 };
    Kind kind = Load;
    ChunkPos pos;
    uint8_t lod = 0;  // unused for Unload

    bool operator==(const ChunkEvent& o) const { return kind == o.kind && pos == o.pos && (kind == Unload || lod == o.lod); }
    bool operator<(const ChunkEvent& o) const {
        if (kind != o.kind) return kind < o.kind;
        if (pos.x != o.pos.x) return pos.x < o.pos.x;
        if (pos.y != o.pos.y) return pos.y < o.pos.y;
        if (pos.z != o.pos.z) return pos.z < o.pos.z;
        return kind != Unload && lod < o.lod;
    }
};

class ChunkLoader {
public:
    // chunkloader.rs:33-44 (asserts start_y < end_y)
    ChunkLoader(uint32_t radius, int32_t start_y, int32_t end_y) : radius_(radius), start_y_(start_y), end_y_(end_y) {
        if (!(start_y < end_y)) throw std::invalid_argument("ChunkLoader: start_y must be below end_y");
    }

    uint32_t get_radius() const { return radius_; }
    // chunkloader.rs:50-54: forgets the last position so that the next update re-checks every chunk
    void set_radius(uint32_t radius) {
        radius_ = radius;
        last_pos_.reset();
        rescan_ = true;
    }

    // chunkloader.rs:58-128: the events caused by the target moving to `pos` (block coordinates); empty while it stays in
    // the same chunk. Sorted by the chunk's distance to the target (stable: equal distances keep discovery order; the
    // reference's Unload events come out of a hash map, so their order among equals is unspecified there).
    std::vector<ChunkEvent> update(float px, float py, float pz) {
        std::vector<ChunkEvent> events;
        const ChunkPos current = ChunkPos::from_block_pos(to_i32(px), to_i32(py), to_i32(pz));
        // The reference never writes last_pos back (chunkloader.rs:60-63), so its early return never fires and every call
        // re-scans the whole radius; a re-scan from an unchanged chunk with an unchanged resident set finds nothing (the
        // scan is idempotent). Returning early in exactly that case gives the same events without the scan, which at
        // radius 40 is ~200k map look-ups per frame.
        if (last_pos_ && *last_pos_ == current && !rescan_) return events;
        last_pos_ = current;
        rescan_ = false;

        const int32_t r = int32_t(radius_);
        for (int32_t dx = -r; dx <= r; ++dx) {
            for (int32_t dz = -r; dz <= r; ++dz) {
                if (dx * dx + dz * dz > r * r) continue;  // only inside the radius
                ChunkPos pos{current.x + dx, 0, current.z + dz};
                const uint8_t lod = calculate_lod(current, pos);
                for (int32_t y = start_y_; y < end_y_; ++y) {
                    const int32_t dy = y - current.y;
                    if (dy < -r || dy > r) continue;
                    pos.y = y;
                    auto it = loaded_.find(pos);
                    if (it != loaded_.end()) {
                        if (it->second != lod) {
                            events.push_back(ChunkEvent{ChunkEvent::LodChange, pos, lod});
                            it->second = lod;
                        }
                    } else {
                        events.push_back(ChunkEvent{ChunkEvent::Load, pos, lod});
                        loaded_.emplace(pos, lod);
                        order_.push_back(pos);
                    }
                }
            }
        }

        // unload what fell out of the radius (in load order, for a deterministic event list)
        std::vector<ChunkPos> keep;
        keep.reserve(order_.size());
        for (const ChunkPos& pos : order_) {
            if (!loaded_.count(pos)) continue;
            const int32_t dx = std::abs(pos.x - current.x), dy = std::abs(pos.y - current.y), dz = std::abs(pos.z - current.z);
            if (dy > r || dx * dx + dz * dz > r * r) {
                events.push_back(ChunkEvent{ChunkEvent::Unload, pos, 0});
                loaded_.erase(pos);
            } else {
                keep.push_back(pos);
            }
        }
        order_.swap(keep);

        std::stable_sort(events.begin(), events.end(), [&](const ChunkEvent& a, const ChunkEvent& b) { return a.pos.dst_sq(current) < b.pos.dst_sq(current); });
        return events;
    }

    // chunkloader.rs:130-137
    static uint8_t calculate_lod(const ChunkPos& center, const ChunkPos& pos) {
        const int32_t d = to_i32(std::sqrt(pos.dst_2d_sq(center)));
        if (d <= 6) return 5;
        if (d <= 12) return 4;
        if (d <= 19) return 3;
        return 2;
    }

    bool is_loaded(const ChunkPos& pos) const { return loaded_.count(pos) != 0; }
    void add_loaded_chunk(const ChunkPos& pos, uint8_t lod) {
        if (loaded_.emplace(pos, lod).second) order_.push_back(pos);
        else loaded_[pos] = lod;
        rescan_ = true;  // the resident set changed behind update()'s back: the next call has to look again
    }
    size_t loaded_count() const { return loaded_.size(); }

private:
    // Rust's `f32 as i32`: truncation toward zero, saturating, NaN -> 0
    static int32_t to_i32(float v) {
        if (std::isnan(v)) return 0;
        if (v >= 2147483648.0f) return INT32_MAX;
        if (v <= -2147483648.0f) return INT32_MIN;
        return int32_t(v);
    }

    uint32_t radius_;
    int32_t start_y_, end_y_;
    std::optional<ChunkPos> last_pos_;
    bool rescan_ = false;
    std::unordered_map<ChunkPos, uint8_t, ChunkPosHash> loaded_;
    std::vector<ChunkPos> order_;  // load order of the keys of loaded_
};

}  // namespace systems
}  // namespace vx
