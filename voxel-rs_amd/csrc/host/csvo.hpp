// CSVO serializer: byte-packed clustered octree (reference src/world/hds/csvo.rs).
//
// Node kinds, chosen by the number of levels left below the node (csvo.rs:434-546):
//   depth > 3  internal : u16 header, 2 bits per child (0 = none, 1/2/3 = child offset stored in 1/2/4 bytes),
//                         then the offset table, then the children back to back
//   depth == 3 pre-leaf : u8 child mask, one u8 offset per present child, children
//   depth == 2 leaf node: u8 child mask, u16 index of this node's first material in the chunk's material
//                         list, then one u8 leaf mask per present child
//   depth == 1          : the u8 leaf mask itself; block ids go to the per-chunk material list in DFS order
// A chunk is framed in the world arena as [lod:u8][material_bytes:u32][materials u32...][nodes...]
// (csvo.rs:217-227). The world root octree is made of internal nodes whose lowest level holds 4-byte
// ABSOLUTE arena offsets with bit 31 set (csvo.rs:68-139).
//
// Restates csvo.rs:28-313 (Csvo) and csvo.rs:393-555 (SerializedChunk). Deviation: changes apply in
// insertion order instead of hash-set order (csvo.rs:198).
#pragma once

#include <cstdint>
#include <cstring>
#include <optional>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "chunk.hpp"
#include "octree.hpp"
#include "range_buffer.hpp"

namespace vx {

namespace detail {

// Shared layout of "internal" nodes (csvo.rs:107-137 and 509-543): offsets are relative to the end of
// the offset table; an offset needs 1, 2 or 4 bytes by magnitude (tag = ilog2(max(off,1)) / 8 + 1).
inline bool csvo_pack_internal(const std::vector<std::pair<uint32_t, std::vector<uint8_t>>>& children, std::vector<uint8_t>& out) {
    uint16_t header = 0;
    out.assign(2, 0);
    std::vector<uint32_t> offsets;
    uint32_t running = 0;
    for (const auto& c : children) {
        offsets.push_back(running);
        running += uint32_t(c.second.size());
    }
    for (size_t i = 0; i < children.size(); ++i) {
        uint32_t v = offsets[i] > 1 ? offsets[i] : 1, bits = 0;
        while (v >>= 1) ++bits;
        const uint32_t tag = bits / 8 + 1;
        if (tag > 3 || (tag == 3 && (offsets[i] & (1u << 31)))) return false;  // reference: unreachable!/assert (csvo.rs:123,126)
        header |= uint16_t(tag << (children[i].first * 2));
        const uint32_t nbytes = tag == 3 ? 4 : tag;
        for (uint32_t b = 0; b < nbytes; ++b) out.push_back(uint8_t(offsets[i] >> (8 * b)));
    }
    for (const auto& c : children) out.insert(out.end(), c.second.begin(), c.second.end());
    out[0] = uint8_t(header & 0xff);
    out[1] = uint8_t(header >> 8);
    return true;
}

}  // namespace detail

class CsvoSerializedChunk {
public:
    ChunkPos pos;
    uint64_t pos_hash = 0;
    uint8_t lod = 0;
    std::optional<std::vector<uint8_t>> buffer;
    std::optional<std::vector<BlockId>> materials;

    CsvoSerializedChunk() = default;

    // csvo.rs:403-432
    explicit CsvoSerializedChunk(const Chunk& chunk) : pos(chunk.pos), pos_hash(chunk_pos_hash(chunk.pos)) {
        const Octree<BlockId>& storage = chunk.storage;
        if (storage.root) {
            uint8_t depth = storage.depth();
            if (chunk.lod != 0 && chunk.lod < depth) depth = chunk.lod;
            materials.emplace();
            buffer = serialize_octant(storage, *storage.root, depth, 0, *materials);
        }
        lod = chunk.lod != 0 ? chunk.lod : storage.depth();
    }

    // csvo.rs:434-546
    static std::vector<uint8_t> serialize_octant(const Octree<BlockId>& octree, OctantId octant_id, uint8_t depth,
                                                 uint16_t material_offset, std::vector<BlockId>& materials) {
        const Octant<BlockId>& octant = octree.octants[octant_id];

        if (depth == 1) {
            uint8_t leaf_mask = 0;
            for (uint32_t idx = 0; idx < 8; ++idx) {
                const Child<BlockId>& child = octant.children[idx];
                if (child.is_none()) continue;
                const BlockId* content = child.leaf_value();
                if (!content && child.is_octant()) content = pick_leaf_for_lod(octree, octree.octants[child.octant]);
                if (!content) continue;
                materials.push_back(*content);
                leaf_mask |= uint8_t(1u << idx);
            }
            return {leaf_mask};
        }

        std::vector<std::pair<uint32_t, std::vector<uint8_t>>> children;
        for (uint32_t idx = 0; idx < 8; ++idx) {
            const Child<BlockId>& child = octant.children[idx];
            if (!child.is_octant()) continue;  // leaves above the uniform leaf level are an assert in the reference (csvo.rs:471)
            children.emplace_back(idx, serialize_octant(octree, child.octant, uint8_t(depth - 1), uint16_t(materials.size()), materials));
        }

        std::vector<uint8_t> out;
        if (depth == 2) {
            out.push_back(0);
            if (!children.empty()) {
                out.push_back(uint8_t(material_offset & 0xff));
                out.push_back(uint8_t(material_offset >> 8));
            }
            for (const auto& c : children) {
                out[0] |= uint8_t(1u << c.first);
                out.insert(out.end(), c.second.begin(), c.second.end());
            }
        } else if (depth == 3) {
            out.assign(1 + children.size(), 0);
            uint8_t running = 0;
            for (size_t i = 0; i < children.size(); ++i) {
                out[0] |= uint8_t(1u << children[i].first);
                out[1 + i] = running;
                running = uint8_t(running + children[i].second.size());
                out.insert(out.end(), children[i].second.begin(), children[i].second.end());
            }
        } else {
            detail::csvo_pack_internal(children, out);
        }
        return out;
    }

    bool has_data() const { return buffer.has_value() && materials.has_value(); }
    uint64_t unique_id() const { return pos_hash; }
};

class Csvo {
public:
    struct LeafInfo {
        size_t buf_offset = 0;  // in bytes
    };

    Octree<CsvoSerializedChunk> octree;
    uint8_t child_depth = 0;
    RangeBuffer buffer;
    std::unordered_map<uint64_t, LeafInfo> leaf_info;
    std::optional<LeafInfo> root_info;

    Csvo() = default;
    explicit Csvo(size_t capacity_bytes) : buffer(capacity_bytes) {}

    void clear() {
        octree.reset();
        changes_.clear();
        child_depth = 0;
        buffer.clear();
        leaf_info.clear();
        root_info.reset();
    }

    // csvo.rs:155-164
    std::pair<LeafId, std::optional<CsvoSerializedChunk>> set_leaf(Position pos, CsvoSerializedChunk leaf, bool serialize) {
        const uint64_t uid = leaf.pos_hash;
        auto r = octree.set_leaf(pos, std::move(leaf));
        if (serialize || !leaf_info.count(uid)) push_change({true, uid, r.first});
        return r;
    }

    std::pair<LeafId, std::optional<CsvoSerializedChunk>> move_leaf(LeafId leaf, Position to) { return octree.move_leaf(leaf, to); }

    std::optional<CsvoSerializedChunk> remove_leaf(LeafId leaf) {
        auto v = octree.remove_leaf_by_id(leaf);
        if (v) push_change({false, v->pos_hash, LeafId{}});
        return v;
    }

    const CsvoSerializedChunk* get_leaf(Position pos) const { return octree.get_leaf(pos); }

    // csvo.rs:189-250
    void serialize() {
        if (!octree.root) return;
        std::vector<Change> changes;
        changes.swap(changes_);
        for (const Change& c : changes) {
            if (c.add) {
                CsvoSerializedChunk* content = nullptr;
                if (c.leaf.parent < octree.octants.size()) content = octree.octants[c.leaf.parent].children[c.leaf.idx].leaf_value();
                if (!content) continue;
                if (content->lod > child_depth) child_depth = content->lod;
                if (content->buffer) {
                    std::vector<uint8_t> nodes = std::move(*content->buffer);
                    content->buffer.reset();
                    std::vector<BlockId> mats = content->materials ? std::move(*content->materials) : std::vector<BlockId>{};
                    content->materials.reset();

                    const uint32_t material_bytes = uint32_t(mats.size() * sizeof(BlockId));
                    tmp_.clear();
                    tmp_.reserve(1 + 4 + material_bytes + nodes.size());
                    tmp_.push_back(content->lod);
                    for (int b = 0; b < 4; ++b) tmp_.push_back(uint8_t(material_bytes >> (8 * b)));
                    for (BlockId m : mats)
                        for (int b = 0; b < 4; ++b) tmp_.push_back(uint8_t(m >> (8 * b)));
                    tmp_.insert(tmp_.end(), nodes.begin(), nodes.end());

                    const size_t off = buffer.insert(c.uid, tmp_.data(), tmp_.size());
                    tmp_.clear();
                    leaf_info[c.uid] = LeafInfo{off};
                }
            } else {
                buffer.remove(c.uid);
                leaf_info.erase(c.uid);
            }
        }

        root_.clear();
        serialize_root(*octree.root, octree.depth(), root_);
        const size_t off = buffer.insert(UINT64_MAX, root_.data(), root_.size());
        root_info = LeafInfo{off};
    }

    uint8_t depth() const { return uint8_t(octree.depth() + child_depth); }  // csvo.rs:252-254
    size_t size_in_bytes() const { return buffer.size_in_bytes(); }

    // [root_ptr:u32 LE][arena] (csvo.rs:262-277)
    size_t write_to(uint8_t* dst) const {
        if (!root_info) return 0;
        write_root_ptr(dst);
        copy_bytes(dst + 4, buffer.bytes.data(), buffer.bytes.size());
        return 4 + buffer.bytes.size();
    }

    // csvo.rs:282-312; `false` replaces the capacity assert
    bool write_changes_to(uint8_t* dst, size_t dst_len, bool reset) {
        if (!root_info || buffer.updated_ranges.empty()) return true;
        write_root_ptr(dst);
        for (const Range& r : buffer.updated_ranges) {
            if (!(r.start + r.length < dst_len)) return false;
            copy_bytes(dst + 4 + r.start, buffer.bytes.data() + r.start, r.length);
        }
        if (reset) buffer.updated_ranges.clear();
        return true;
    }

private:
    struct Change {
        bool add;
        uint64_t uid;
        LeafId leaf;
    };
    std::vector<Change> changes_;
    std::vector<uint8_t> tmp_;

    void push_change(const Change& c) {
        // set semantics; recent duplicates are the only ones that occur in practice
        for (size_t i = changes_.size(); i-- > 0 && changes_.size() - i <= 8;) {
            const Change& o = changes_[i];
            if (o.add == c.add && o.uid == c.uid && o.leaf == c.leaf) return;
        }
        changes_.push_back(c);
    }

    void write_root_ptr(uint8_t* dst) const {
        const uint32_t p = uint32_t(root_info->buf_offset);
        for (int b = 0; b < 4; ++b) dst[b] = uint8_t(p >> (8 * b));
    }

    // csvo.rs:68-139: the root octree as internal nodes, appended to `out` -- the node's header, its offset table (1/2/4-byte offsets by magnitude,
    // relative to the table's end: detail::csvo_pack_internal's layout), its children back to back; the lowest level holds 4-byte absolute
    // arena offsets with bit 31 set. Every commit rewrites the root (csvo.rs:237-249), and a vector per node and per child was a third of a
    // streaming step's host time: the children are emitted straight behind a placeholder for the largest possible table and moved up once
    // the table's size is known -- no allocation per node.
    void serialize_root(OctantId octant_id, uint8_t depth, std::vector<uint8_t>& out) const {
        const Octant<CsvoSerializedChunk>& octant = octree.octants[octant_id];
        constexpr size_t kMaxTable = 8 * 4;
        const size_t start = out.size();
        out.resize(start + 2 + kMaxTable);
        size_t child_at[8];
        uint32_t child_idx[8], n = 0;
        for (uint32_t idx = 0; idx < 8; ++idx) {
            const Child<CsvoSerializedChunk>& child = octant.children[idx];
            if (child.is_none()) continue;
            if (depth == 1) {
                if (const CsvoSerializedChunk* content = child.leaf_value()) {
                    auto it = leaf_info.find(content->pos_hash);
                    if (it != leaf_info.end()) {
                        const uint32_t pointer = uint32_t(it->second.buf_offset) | (1u << 31);
                        child_at[n] = out.size();
                        child_idx[n++] = idx;
                        for (int b = 0; b < 4; ++b) out.push_back(uint8_t(pointer >> (8 * b)));
                    }
                }
                continue;
            }
            if (!child.is_octant()) continue;  // reference asserts uniform leaf level (csvo.rs:87)
            child_at[n] = out.size();
            child_idx[n++] = idx;
            serialize_root(child.octant, uint8_t(depth - 1), out);
        }
        const size_t body = start + 2 + kMaxTable;  // where the children's bytes begin for now
        uint16_t header = 0;
        size_t table = 0;
        uint8_t entries[kMaxTable];
        if (depth == 1) {
            // (the entries ARE the pointers: tag 3 per present child, nothing behind the table)
            for (uint32_t i = 0; i < n; ++i) header |= uint16_t(3u << (child_idx[i] * 2));
        } else {
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t offset = uint32_t(child_at[i] - body);
                uint32_t v = offset > 1 ? offset : 1, bits = 0;
                while (v >>= 1) ++bits;
                const uint32_t tag = bits / 8 + 1;  // (an offset of 2^24 and more cannot occur: a root octree is kilobytes)
                header |= uint16_t((tag > 3 ? 3u : tag) << (child_idx[i] * 2));
                const uint32_t nbytes = tag >= 3 ? 4 : tag;
                for (uint32_t b = 0; b < nbytes; ++b) entries[table++] = uint8_t(offset >> (8 * b));
            }
        }
        out[start] = uint8_t(header & 0xff);
        out[start + 1] = uint8_t(header >> 8);
        std::memcpy(out.data() + start + 2, entries, table);
        const size_t body_bytes = out.size() - body;
        std::memmove(out.data() + start + 2 + table, out.data() + body, body_bytes);
        out.resize(start + 2 + table + body_bytes);
    }

    std::vector<uint8_t> root_;  // serialize()'s scratch: the root octree's bytes
};

}  // namespace vx
