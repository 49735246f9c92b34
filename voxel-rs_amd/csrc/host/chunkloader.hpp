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
    //
    // The reference keeps one hash-map entry per loaded chunk and looks every chunk of the cylinder up on every call (at radius 40 with 81
    // layers: 407 K look-ups, 35-45 ms each time the target enters a new chunk -- two dropped frames of the caller's loop, profiles/round4
    // stream_d14_*.json's host_ms_per_step_max). The same events come out of a table per COLUMN (x, z) of the cylinder -- a byte per layer: the
    // LOD the chunk is loaded at, 0 = not loaded -- at 5 K look-ups and a byte read per chunk: under a millisecond.
    std::vector<ChunkEvent> update(float px, float py, float pz) {
        std::vector<ChunkEvent> events;
        const ChunkPos current = ChunkPos::from_block_pos(to_i32(px), to_i32(py), to_i32(pz));
        // The reference never writes last_pos back (chunkloader.rs:60-63), so its early return never fires and every call
        // re-scans the whole radius; a re-scan from an unchanged chunk with an unchanged resident set finds nothing (the
        // scan is idempotent). Returning early in exactly that case gives the same events without the scan.
        if (last_pos_ && *last_pos_ == current && !rescan_) return events;
        last_pos_ = current;
        rescan_ = false;

        const int32_t r = int32_t(radius_);
        // the layers of a column that are within reach of the target's own layer
        const int32_t y_lo = std::max(start_y_, sat_add(current.y, -r)), y_hi = std::min(end_y_ - 1, sat_add(current.y, r));
        for (int32_t dx = -r; dx <= r; ++dx) {
            for (int32_t dz = -r; dz <= r; ++dz) {
                if (dx * dx + dz * dz > r * r) continue;  // only inside the radius
                if (y_lo > y_hi) continue;
                const ChunkPos base{current.x + dx, 0, current.z + dz};
                const uint8_t lod = calculate_lod(current, base);
                Column& col = column_at(base.x, base.z);
                // (what the last scan left in this column -- these layers, at this LOD, nothing else -- is what this one would: most columns, most times)
                if (col.settled && col.settled_lod == lod && col.settled_lo == y_lo && col.settled_hi == y_hi) continue;
                for (int32_t y = y_lo; y <= y_hi; ++y) {
                    uint8_t& at = col.lod[size_t(y - start_y_)];
                    if (at == lod) continue;
                    events.push_back(ChunkEvent{at ? ChunkEvent::LodChange : ChunkEvent::Load, ChunkPos{base.x, y, base.z}, lod});
                    if (!at) { ++col.loaded; ++loaded_count_; }
                    at = lod;
                }
                // settled if nothing of the column is loaded outside the window (the unload pass below removes such layers; it re-marks then)
                col.settled = col.loaded == uint32_t(y_hi - y_lo + 1);
                col.settled_lod = lod; col.settled_lo = y_lo; col.settled_hi = y_hi;
            }
        }

        // unload what fell out of the radius: columns in the order they were first loaded, a column's layers from the bottom
        size_t kept = 0;
        for (size_t i = 0; i < column_order_.size(); ++i) {
            const ColumnKey key = column_order_[i];
            auto it = columns_.find(key);
            Column& col = it->second;
            const int64_t dx = int64_t(key.x) - current.x, dz = int64_t(key.z) - current.z;
            const bool outside = dx * dx + dz * dz > int64_t(r) * r;
            if (col.loaded && !outside && col.settled && col.settled_lo == y_lo && col.settled_hi == y_hi) {
                column_order_[kept++] = key;  // (exactly the window's layers are loaded: nothing to unload)
                continue;
            }
            if (col.loaded && (outside || col.first_layer() < y_lo - start_y_ || col.last_layer() > y_hi - start_y_)) {
                col.settled = false;
                for (int32_t y = start_y_; y < end_y_; ++y) {
                    uint8_t& at = col.lod[size_t(y - start_y_)];
                    if (!at || (!outside && y >= y_lo && y <= y_hi)) continue;
                    events.push_back(ChunkEvent{ChunkEvent::Unload, ChunkPos{key.x, y, key.z}, 0});
                    at = 0;
                    --col.loaded;
                    --loaded_count_;
                }
            }
            if (col.loaded) column_order_[kept++] = key;
            else columns_.erase(it);
        }
        column_order_.resize(kept);
        // (chunks outside [start_y, end_y) only ever get here through add_loaded_chunk: the scan above never visits them, the reach test does)
        for (auto it = stray_.begin(); it != stray_.end();) {
            const int64_t dx = int64_t(it->first.x) - current.x, dy = std::abs(int64_t(it->first.y) - current.y), dz = int64_t(it->first.z) - current.z;
            if (dy > r || dx * dx + dz * dz > int64_t(r) * r) {
                events.push_back(ChunkEvent{ChunkEvent::Unload, it->first, 0});
                it = stray_.erase(it);
                --loaded_count_;
            } else {
                ++it;
            }
        }

        sort_by_distance(events, current);
        return events;
    }

    // nearest first, equal distances in discovery order (chunkloader.rs:123-126 sorts by dst_sq). The squared distances are small integers: a
    // stable counting sort, linear in the events (a re-centre at radius 40 makes 25 K of them; std::stable_sort is the fall-back for far-away ones)
    static void sort_by_distance(std::vector<ChunkEvent>& events, const ChunkPos& current) {
        if (events.size() < 2) return;
        std::vector<int64_t> key(events.size());
        int64_t top = 0;
        for (size_t i = 0; i < events.size(); ++i) {
            key[i] = int64_t(events[i].pos.dst_sq(current));
            top = std::max(top, key[i]);
        }
        if (top < 0 || top >= (int64_t(1) << 22)) {
            std::stable_sort(events.begin(), events.end(), [&](const ChunkEvent& a, const ChunkEvent& b) { return a.pos.dst_sq(current) < b.pos.dst_sq(current); });
            return;
        }
        std::vector<uint32_t> first(size_t(top) + 2, 0);
        for (int64_t k : key) ++first[size_t(k) + 1];
        for (size_t k = 1; k < first.size(); ++k) first[k] += first[k - 1];
        std::vector<ChunkEvent> sorted(events.size());
        for (size_t i = 0; i < events.size(); ++i) sorted[first[size_t(key[i])]++] = events[i];
        events.swap(sorted);
    }

    // chunkloader.rs:130-137
    static uint8_t calculate_lod(const ChunkPos& center, const ChunkPos& pos) {
        const int32_t d = to_i32(std::sqrt(pos.dst_2d_sq(center)));
        if (d <= 6) return 5;
        if (d <= 12) return 4;
        if (d <= 19) return 3;
        return 2;
    }

    bool is_loaded(const ChunkPos& pos) const {
        if (pos.y < start_y_ || pos.y >= end_y_) return stray_.count(pos) != 0;
        auto it = columns_.find(ColumnKey{pos.x, pos.z});
        return it != columns_.end() && it->second.lod[size_t(pos.y - start_y_)] != 0;
    }
    void add_loaded_chunk(const ChunkPos& pos, uint8_t lod) {
        if (pos.y < start_y_ || pos.y >= end_y_) {
            if (stray_.emplace(pos, lod).second) ++loaded_count_;
            else stray_[pos] = lod;
        } else {
            Column& col = column_at(pos.x, pos.z);
            uint8_t& at = col.lod[size_t(pos.y - start_y_)];
            if (!at) { ++col.loaded; ++loaded_count_; }
            at = lod ? lod : uint8_t(255);  // (0 means "not loaded" in the table; the reference's LODs are 2..5)
            col.settled = false;
        }
        rescan_ = true;  // the resident set changed behind update()'s back: the next call has to look again
    }
    size_t loaded_count() const { return loaded_count_; }

private:
    // Rust's `f32 as i32`: truncation toward zero, saturating, NaN -> 0
    static int32_t to_i32(float v) {
        if (std::isnan(v)) return 0;
        if (v >= 2147483648.0f) return INT32_MAX;
        if (v <= -2147483648.0f) return INT32_MIN;
        return int32_t(v);
    }

    static int32_t sat_add(int32_t a, int32_t b) {
        const int64_t v = int64_t(a) + b;
        return v > INT32_MAX ? INT32_MAX : (v < INT32_MIN ? INT32_MIN : int32_t(v));
    }

    // one (x, z) column of the cylinder: per layer of [start_y, end_y) the LOD its chunk is loaded at, 0 = not loaded
    struct ColumnKey {
        int32_t x, z;
        bool operator==(const ColumnKey& o) const { return x == o.x && z == o.z; }
    };
    struct ColumnKeyHash {
        size_t operator()(const ColumnKey& k) const { return size_t(uint32_t(k.x)) * 0x9E3779B97F4A7C15ull ^ (size_t(uint32_t(k.z)) << 32 | uint32_t(k.z)); }
    };
    struct Column {
        std::vector<uint8_t> lod;
        uint32_t loaded = 0;
        // update()'s short cut: the column holds exactly the layers [settled_lo, settled_hi], every one at settled_lod
        bool settled = false;
        uint8_t settled_lod = 0;
        int32_t settled_lo = 0, settled_hi = 0;
        int32_t first_layer() const { for (size_t i = 0; i < lod.size(); ++i) if (lod[i]) return int32_t(i); return INT32_MAX; }
        int32_t last_layer() const { for (size_t i = lod.size(); i-- > 0;) if (lod[i]) return int32_t(i); return INT32_MIN; }
    };
    Column& column_at(int32_t x, int32_t z) {
        auto r = columns_.try_emplace(ColumnKey{x, z});
        if (r.second) {
            r.first->second.lod.assign(size_t(int64_t(end_y_) - start_y_), 0);
            column_order_.push_back(ColumnKey{x, z});
        }
        return r.first->second;
    }

    uint32_t radius_;
    int32_t start_y_, end_y_;
    std::optional<ChunkPos> last_pos_;
    bool rescan_ = false;
    std::unordered_map<ColumnKey, Column, ColumnKeyHash> columns_;
    std::vector<ColumnKey> column_order_;  // the keys of columns_ in the order they were first loaded
    std::unordered_map<ChunkPos, uint8_t, ChunkPosHash> stray_;  // add_loaded_chunk outside [start_y, end_y)
    size_t loaded_count_ = 0;
};

}  // namespace systems
}  // namespace vx
