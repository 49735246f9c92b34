// ESVO serializer: octree -> flat u32 buffer the traversal kernels walk.
//
// Format (reference src/world/hds/esvo.rs:74-101): one octant = 12 u32 = 4 header words + 8 body words.
// Header word i/2 carries, in its low (even i) or high (odd i) 16 bits, `(child_mask << 8) | leaf_mask`
// OF CHILD i, i.e. the masks describe the child's own children. Body word i is, for an inner child, a
// relative pointer (bit 31 set, offset counted from the body word itself) and, for a leaf child, the
// leaf value. The world-level octree ("octree of octrees") stores whole serialized chunks as leaves and
// points at them with ABSOLUTE u32 indices (bit 31 clear), esvo.rs:151-175, so a chunk can be moved by
// rewriting one word.
//
// This file restates esvo.rs:122-512 (Esvo, SerializedChunk, serialize_octant). Deviation: the
// reference drains its change set from a hash set (arbitrary order, esvo.rs:246); here changes apply in
// insertion order, so multi-chunk layouts are deterministic.
#pragma once

#include <cstdint>
#include <optional>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "chunk.hpp"
#include "octree.hpp"
#include "range_buffer.hpp"

namespace vx {

struct EsvoResult {
    uint8_t child_mask = 0;  // bit per non-empty child
    uint8_t leaf_mask = 0;   // bit per child that is a leaf value
    uint8_t depth = 0;       // 0 = nothing serialized, 1 = leaves only, n = levels below
    bool operator==(const EsvoResult& o) const { return child_mask == o.child_mask && leaf_mask == o.leaf_mask && depth == o.depth; }
};

// Depth-first octant writer shared by chunks and the world root (esvo.rs:439-512).
// `encode(idx, content, octant_words, result)` is called for every leaf (or LOD cut-off) child.
template <class T, class Encoder>
EsvoResult esvo_serialize_octant(const Octree<T>& octree, OctantId octant_id, std::vector<uint32_t>& dst, uint8_t lod,
                                 const Encoder& encode) {
    const size_t start = dst.size();
    dst.insert(dst.end(), 12, 0u);

    EsvoResult result;
    for (uint32_t idx = 0; idx < 8; ++idx) {
        const Child<T>& child = octree.octants[octant_id].children[idx];
        if (child.is_none()) continue;
        result.child_mask |= uint8_t(1u << idx);

        if (child.is_leaf() || lod == 1) {
            const T* content = child.leaf_value();
            if (!content && child.is_octant()) content = pick_leaf_for_lod(octree, octree.octants[child.octant]);
            if (!content) continue;
            encode(uint8_t(idx), *content, dst.data() + start, result);
        } else {
            const uint8_t child_lod = lod > 0 ? uint8_t(lod - 1) : uint8_t(0);
            const uint32_t child_offset = uint32_t(dst.size() - start);
            const EsvoResult cr = esvo_serialize_octant(octree, child.octant, dst, child_lod, encode);

            uint32_t mask = (uint32_t(cr.child_mask) << 8) | cr.leaf_mask;
            if (idx & 1) mask <<= 16;
            dst[start + idx / 2] |= mask;
            dst[start + 4 + idx] = (child_offset - 4 - idx) | (1u << 31);
            if (uint8_t(cr.depth + 1) > result.depth) result.depth = uint8_t(cr.depth + 1);
        }
    }
    return result;
}

// A chunk serialized once at construction; the world SVO later copies the words into its arena
// (esvo.rs:343-413).
class EsvoSerializedChunk {
public:
    ChunkPos pos;
    uint64_t pos_hash = 0;
    uint8_t lod = 0;
    std::optional<std::vector<uint32_t>> buffer;
    EsvoResult result;

    EsvoSerializedChunk() = default;
    explicit EsvoSerializedChunk(const Chunk& chunk) : pos(chunk.pos), pos_hash(chunk_pos_hash(chunk.pos)), lod(chunk.lod) {
        std::vector<uint32_t> words;
        result = serialize_storage(chunk.storage, words, lod);
        if (result.depth > 0) buffer = std::move(words);
    }

    // esvo.rs:369-383: leaves write their block id into the body and flag the leaf bit
    static EsvoResult serialize_storage(const Octree<BlockId>& octree, std::vector<uint32_t>& dst, uint8_t lod) {
        if (!octree.root) return {};
        return esvo_serialize_octant(octree, *octree.root, dst, lod,
                                     [](uint8_t idx, const BlockId& value, uint32_t* words, EsvoResult& res) {
                                         res.leaf_mask |= uint8_t(1u << idx);
                                         words[4 + idx] = value;
                                         res.depth = 1;
                                     });
    }

    bool has_data() const { return buffer.has_value(); }
    uint64_t unique_id() const { return pos_hash; }

    // hands the cached words over (once) and reports the cached masks, esvo.rs:401-412
    EsvoResult serialize(std::vector<uint32_t>& dst, uint8_t /*lod*/) {
        if (buffer) {
            dst.insert(dst.end(), buffer->begin(), buffer->end());
            buffer.reset();
        }
        return result;
    }
};

// Plain u32 leaf, as the reference's tests use to exercise the world SVO without chunks
// (src/systems/worldsvo.rs:236-245).
struct EsvoU32Leaf {
    uint32_t value = 0;
    uint64_t unique_id() const { return value; }
    EsvoResult serialize(std::vector<uint32_t>& dst, uint8_t) {
        dst.push_back(value);
        return {1, 1, 1};
    }
    bool operator==(const EsvoU32Leaf& o) const { return value == o.value; }
};

template <class T>
class Esvo {
public:
    static constexpr uint32_t kPreambleU32 = 5;  // esvo.rs:134

    struct LeafInfo {
        size_t buf_offset = 0;  // in u32 units
        EsvoResult serialization;
    };

    Octree<T> octree;
    RangeBuffer buffer;
    std::unordered_map<uint64_t, LeafInfo> leaf_info;
    std::optional<LeafInfo> root_info;

    Esvo() = default;
    explicit Esvo(size_t capacity_bytes) : buffer(capacity_bytes) {}

    void clear() {
        octree.reset();
        changes_.clear();
        change_keys_.clear();
        buffer.clear();
        leaf_info.clear();
        root_info.reset();
    }

    // esvo.rs:203-212
    std::pair<LeafId, std::optional<T>> set_leaf(Position pos, T leaf, bool serialize) {
        const uint64_t uid = leaf.unique_id();
        auto r = octree.set_leaf(pos, std::move(leaf));
        if (serialize || !leaf_info.count(uid)) push_change({true, uid, r.first});
        return r;
    }

    std::pair<LeafId, std::optional<T>> move_leaf(LeafId leaf, Position to) { return octree.move_leaf(leaf, to); }

    // esvo.rs:221-228
    std::optional<T> remove_leaf(LeafId leaf) {
        auto v = octree.remove_leaf_by_id(leaf);
        if (v) push_change({false, v->unique_id(), LeafId{}});
        return v;
    }

    const T* get_leaf(Position pos) const { return octree.get_leaf(pos); }

    // Applies pending adds/removes to the arena, then re-serializes the root octree (esvo.rs:237-276).
    void serialize() {
        if (!octree.root) return;
        std::vector<uint32_t>& tmp = tmp_;

        std::vector<Change> changes;
        changes.swap(changes_);
        change_keys_.clear();
        for (const Change& c : changes) {
            if (c.add) {
                T* content = nullptr;
                if (c.leaf.parent < octree.octants.size()) content = octree.octants[c.leaf.parent].children[c.leaf.idx].leaf_value();
                if (!content) continue;  // reference would panic (esvo.rs:251); the leaf moved or vanished meanwhile
                const EsvoResult res = content->serialize(tmp, 0);
                if (res.depth > 0) {
                    const size_t off = buffer.insert(c.uid, reinterpret_cast<const uint8_t*>(tmp.data()), tmp.size() * 4);
                    tmp.clear();
                    leaf_info[c.uid] = LeafInfo{off / 4, res};
                }
            } else {
                buffer.remove(c.uid);
                leaf_info.erase(c.uid);
            }
        }

        const EsvoResult res = serialize_root(tmp);
        const size_t off = buffer.insert(UINT64_MAX, reinterpret_cast<const uint8_t*>(tmp.data()), tmp.size() * 4);
        tmp.clear();
        root_info = LeafInfo{off / 4, res};
    }

    uint8_t depth() const { return root_info ? root_info->serialization.depth : 0; }
    size_t size_in_bytes() const { return buffer.size_in_bytes(); }

    // Full image: 5-word preamble + arena. Returns bytes written (esvo.rs:291-305).
    size_t write_to(uint8_t* dst) const {
        if (!root_info) return 0;
        write_preamble(*root_info, dst);
        copy_bytes(dst + kPreambleU32 * 4, buffer.bytes.data(), buffer.bytes.size());
        return kPreambleU32 * 4 + buffer.bytes.size();
    }

    // Preamble + dirty ranges only; `false` replaces the reference's capacity assert (esvo.rs:310-339).
    bool write_changes_to(uint8_t* dst, size_t dst_len, bool reset) {
        if (!root_info || buffer.updated_ranges.empty()) return true;
        write_preamble(*root_info, dst);
        uint8_t* body = dst + kPreambleU32 * 4;
        for (const Range& r : buffer.updated_ranges) {
            if (!(r.start + r.length < dst_len)) return false;
            copy_bytes(body + r.start, buffer.bytes.data() + r.start, r.length);
        }
        if (reset) buffer.updated_ranges.clear();
        return true;
    }

    // The entry point into the structure is a fake octant whose child 0 is the root octree (esvo.rs:179-188).
    static void write_preamble(const LeafInfo& info, uint8_t* dst) {
        const uint32_t words[kPreambleU32] = {uint32_t(info.serialization.child_mask) << 8, 0, 0, 0,
                                              uint32_t(info.buf_offset) + kPreambleU32};
        std::memcpy(dst, words, sizeof(words));
    }

private:
    struct Change {
        bool add;
        uint64_t uid;
        LeafId leaf;
    };
    std::vector<Change> changes_;
    std::unordered_set<uint64_t> change_keys_;  // set semantics of the reference's FxHashSet<OctantChange>
    std::vector<uint32_t> tmp_;

    void push_change(const Change& c) {
        // key = (kind, uid, leaf) folded; collisions only cost a duplicate (idempotent) change
        const uint64_t key = (c.uid * 0x9E3779B97F4A7C15ull) ^ (uint64_t(c.leaf.parent) << 8) ^ c.leaf.idx ^ (c.add ? 0x8000000000000000ull : 0);
        if (!change_keys_.insert(key).second) {
            for (const Change& o : changes_)
                if (o.add == c.add && o.uid == c.uid && o.leaf == c.leaf) return;
        }
        changes_.push_back(c);
    }

    // esvo.rs:151-175
    EsvoResult serialize_root(std::vector<uint32_t>& dst) const {
        return esvo_serialize_octant(octree, *octree.root, dst, 0, [this](uint8_t idx, const T& content, uint32_t* words, EsvoResult& res) {
            auto it = leaf_info.find(content.unique_id());
            if (it == leaf_info.end()) return;
            const LeafInfo& info = it->second;
            uint32_t mask = (uint32_t(info.serialization.child_mask) << 8) | info.serialization.leaf_mask;
            if (idx & 1) mask <<= 16;
            words[idx / 2] |= mask;
            words[4 + idx] = uint32_t(info.buf_offset) + kPreambleU32;
            const uint8_t d = uint8_t(info.serialization.depth + 1);
            if (d > res.depth) res.depth = d;
        });
    }
};

}  // namespace vx
