// CSVO -> traversal image. The reference's compressed node format (src/world/hds/csvo.rs:434-546) costs the traversal a
// bit-field decode, two popcount sums and a dependent table read per descent; its 48-byte-octant sibling (esvo.rs:74-101)
// costs two independent loads. A CSVO world is therefore re-laid out, on the host at commit time, as octants of that second
// kind -- the "traversal image" -- and rays walk the image. Structure is preserved node for node (including the empty
// octants the reference's never-compacted root octree carries), so every ray that starts outside a voxel takes the same
// iterations to the same leaf with the same floats. A ray that starts INSIDE a voxel makes the reference wander through
// leaf bytes as if they were nodes, which is format specific: the kernel hands exactly those rays to the CSVO traversal on
// the original bytes (render_persistent, kForeign).
//
// Host-only, no HIP calls: vx_api.hip owns the device side. Image layout = the ESVO frame Trav<VX_SVO_ESVO> reads:
//   [f32 2^-depth][5-word preamble: root masks, 0, 0, 0, absolute index of the root octant][arena of 12-word octants]
// Octant of node N: words 0..3 = masks of N's children, two per word (child_mask << 8 | leaf_mask); words 4..11 = per child
// the relative pointer (bit 31) to its octant, or the absolute index of a chunk's root octant, or the leaf value.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace vximg {

struct Range {
    uint64_t start, length;
};

// byte reader over descriptors[] (world + 8), reads beyond the end return 0 like the traversal's
struct Bytes {
    const uint8_t* p;
    size_t n;
    uint32_t u8(uint64_t o) const { return o < n ? p[o] : 0u; }
    uint32_t u16(uint64_t o) const { return u8(o) | (u8(o + 1) << 8); }
    uint32_t u32(uint64_t o) const { return u16(o) | (u16(o + 2) << 16); }
};

inline uint32_t tag_bytes(uint32_t m) {  // bytes of the table entries a 2-bit-per-child mask selects: tag 0,1,2,3 -> 0,1,2,4
    uint32_t s = 0;
    for (int c = 0; c < 8; ++c) {
        const uint32_t t = (m >> (2 * c)) & 3u;
        s += t == 3 ? 4u : t;
    }
    return s;
}

struct NodeMasks {
    uint32_t child_mask = 0, leaf_mask = 0;
    uint32_t packed() const { return (child_mask << 8) | leaf_mask; }
};

// One chunk frame [lod:u8][material_bytes:u32][materials][nodes] (csvo.rs:217-227) -> position-independent octants.
struct ChunkImage {
    std::vector<uint32_t> words;  // octant 0 = the chunk's root node
    NodeMasks root;
    uint64_t csvo_end = 0;        // one past the last byte the chunk's nodes and materials occupy
};

class ChunkTranscoder {
public:
    ChunkTranscoder(Bytes b, ChunkImage& out) : b_(b), out_(out) {}

    void run(uint64_t frame) {
        const uint32_t lod = b_.u8(frame);
        const uint32_t material_bytes = b_.u32(frame + 1);
        materials_ = frame + 5;
        end_ = materials_ + material_bytes;
        out_.words.clear();
        out_.root = node(materials_ + material_bytes, lod, 0);
        out_.csvo_end = end_;
    }

private:
    // emits the octant of the node at `ptr` (appended to out_.words, children after it) and returns the node's masks
    NodeMasks node(uint64_t ptr, uint32_t depth, uint64_t pre_leaf) {
        const size_t at = out_.words.size();
        out_.words.resize(at + 12, 0u);
        NodeMasks m;
        if (depth == 0 || depth > 32) return m;  // malformed: an octant without children
        if (depth == 1) {
            // a leaf-mask byte inside its depth-2 parent: the children are voxels (read_leaf, svo.csvo.glsl:119-133)
            const uint32_t mask = b_.u8(ptr);
            end_ = std::max(end_, ptr + 1);
            const uint32_t material_offset = b_.u16(pre_leaf + 1);
            const uint64_t leaf_index = ptr - (pre_leaf + 3);
            for (uint32_t c = 0; c < 8; ++c) {
                if (!((mask >> c) & 1u)) continue;
                const uint64_t bit_mark = leaf_index * 8 + c;  // leaves preceding this one under the depth-2 node
                uint32_t preceding = 0;
                for (uint64_t k = 0; k < bit_mark; ++k) preceding += (b_.u8(pre_leaf + 3 + k / 8) >> (k % 8)) & 1u;
                out_.words[at + 4 + c] = b_.u32(materials_ + uint64_t(material_offset) * 4 + uint64_t(preceding) * 4);
            }
            m.child_mask = m.leaf_mask = mask;
            return m;
        }
        for (uint32_t c = 0; c < 8; ++c) {
            uint64_t child = 0;
            if (depth > 3) {  // internal node: u16 header, 2 bits per child, 1/2/4-byte forward offsets (svo.csvo.glsl:56-97)
                const uint32_t header = b_.u16(ptr);
                const uint32_t tag = (header >> (2 * c)) & 3u;
                if (!tag) continue;
                const uint32_t offset = tag_bytes(header & ((1u << (2 * c)) - 1u)), table = tag_bytes(header);
                const uint32_t width = tag == 3 ? 4u : tag;
                uint32_t e = 0;
                for (uint32_t k = 0; k < width; ++k) e |= b_.u8(ptr + 2 + offset + k) << (8 * k);
                end_ = std::max(end_, ptr + 2 + table);
                child = ptr + 2 + table + e;  // (an absolute pointer, bit 31, only exists in the root octree: see RootTranscoder)
            } else if (depth == 3) {  // pre-leaf node: u8 mask, u8 offsets (svo.csvo.glsl:107-112)
                const uint32_t header = b_.u8(ptr);
                if (!((header >> c) & 1u)) continue;
                const uint32_t offset = uint32_t(__builtin_popcount(header & ((1u << c) - 1u))), table = uint32_t(__builtin_popcount(header));
                end_ = std::max(end_, ptr + 1 + table);
                child = ptr + 1 + table + b_.u8(ptr + 1 + offset);
            } else {  // depth 2, leaf node: u8 mask, u16 material offset, one leaf-mask byte per child (svo.csvo.glsl:114-115)
                const uint32_t header = b_.u8(ptr);
                if (!((header >> c) & 1u)) continue;
                child = ptr + 3 + uint32_t(__builtin_popcount(header & ((1u << c) - 1u)));
                end_ = std::max(end_, ptr + 3 + uint32_t(__builtin_popcount(header)));
            }
            const size_t child_at = out_.words.size();
            const NodeMasks cm = node(child, depth - 1, depth == 2 ? ptr : pre_leaf);
            m.child_mask |= 1u << c;
            out_.words[at + (c >> 1)] |= cm.packed() << ((c & 1u) * 16);
            out_.words[at + 4 + c] = 0x80000000u | uint32_t(child_at - (at + 4 + c));  // relative to this body word (esvo.rs:501-504)
        }
        return m;
    }

    Bytes b_;
    ChunkImage& out_;
    uint64_t materials_ = 0, end_ = 0;
};

// first-fit word allocator over the image arena (the reference's RangeBuffer idea, internal.rs:163-277)
class WordAllocator {
public:
    uint64_t alloc(uint64_t n) {
        for (size_t i = 0; i < free_.size(); ++i) {
            if (free_[i].length < n) continue;
            const uint64_t at = free_[i].start;
            free_[i].start += n;
            free_[i].length -= n;
            if (!free_[i].length) free_.erase(free_.begin() + long(i));
            return at;
        }
        const uint64_t at = end_;
        end_ += n;
        return at;
    }
    void release(uint64_t at, uint64_t n) {
        if (!n) return;
        auto it = std::lower_bound(free_.begin(), free_.end(), at, [](const Range& r, uint64_t v) { return r.start < v; });
        it = free_.insert(it, Range{at, n});
        if (it + 1 != free_.end() && it->start + it->length == (it + 1)->start) {
            it->length += (it + 1)->length;
            free_.erase(it + 1);
        }
        if (it != free_.begin() && (it - 1)->start + (it - 1)->length == it->start) {
            (it - 1)->length += it->length;
            free_.erase(it);
        }
    }
    uint64_t end() const { return end_; }
    void reset(uint64_t first) { free_.clear(); end_ = first; }

private:
    std::vector<Range> free_;
    uint64_t end_ = 0;
};

// The image of a whole CSVO world, kept up to date commit by commit.
class WorldImage {
public:
    static constexpr uint64_t kPreambleWords = 5;

    // host mirror of the image frame: byte 0 = f32 scale, then descriptors[] words
    const std::vector<uint32_t>& frame() const { return frame_; }  // frame_[0] = scale bits, frame_[1 + i] = descriptors[i]
    uint64_t frame_bytes() const { return frame_.size() * 4; }
    // byte ranges of frame() changed by the last update(), sorted and merged
    std::vector<Range> dirty_bytes() const {
        std::vector<Range> r = dirty_;
        std::sort(r.begin(), r.end(), [](const Range& a, const Range& b) { return a.start < b.start; });
        std::vector<Range> out;
        for (const Range& x : r) {
            if (!out.empty() && x.start <= out.back().start + out.back().length) {
                out.back().length = std::max(out.back().start + out.back().length, x.start + x.length) - out.back().start;
            } else {
                out.push_back(x);
            }
        }
        return out;
    }
    size_t chunk_count() const { return chunks_.size(); }

    // `world` = the CSVO frame as committed: [f32 scale][u32 root_ptr][descriptor bytes]; `used` = bytes of the arena in use;
    // `changed` = byte ranges (relative to the arena, like vx_commit's) rewritten since the last call, or empty + `all` = true.
    // Returns false when the world cannot be imaged (malformed or image beyond 4 GiB): the caller then traverses the CSVO bytes.
    bool update(const uint8_t* world, uint64_t used, const Range* changed, size_t n_changed, bool all, unsigned threads) {
        dirty_.clear();
        if (used < 2) return false;
        const Bytes b{world + 8, size_t(used)};
        uint32_t scale_bits, root_ptr;
        std::memcpy(&scale_bits, world, 4);
        std::memcpy(&root_ptr, world + 4, 4);
        const uint32_t depth = 127u - ((scale_bits >> 23) & 0xffu);  // svo.csvo.glsl:254
        if (depth < 1 || depth > 23) return false;
        if (frame_.empty()) {
            frame_.assign(1 + kPreambleWords, 0u);
            alloc_.reset(kPreambleWords);
            all = true;
        }
        if (all) {
            for (auto& kv : chunks_) alloc_.release(kv.second.at, kv.second.words);
            chunks_.clear();
        }

        // 1. walk the root octree: which chunk frames does it reference, and where
        std::vector<uint32_t> root_words;
        std::vector<std::pair<size_t, uint32_t>> chunk_refs;  // (index of the body word in root_words, chunk frame offset)
        const NodeMasks root_masks = root_node(b, root_ptr, depth, root_words, chunk_refs);

        // 2. drop images of chunks that are gone or whose bytes were rewritten
        std::unordered_set<uint32_t> referenced;
        for (auto& r : chunk_refs) referenced.insert(r.second);
        for (auto it = chunks_.begin(); it != chunks_.end();) {
            bool stale = !referenced.count(it->first);
            for (size_t i = 0; i < n_changed && !stale; ++i)
                stale = changed[i].start < it->second.csvo_end && it->first < changed[i].start + changed[i].length;
            if (stale) {
                alloc_.release(it->second.at, it->second.words);
                it = chunks_.erase(it);
            } else {
                ++it;
            }
        }

        // 3. transcode what is missing (worker threads), then place it
        std::vector<uint32_t> todo;
        for (uint32_t off : referenced)
            if (!chunks_.count(off)) todo.push_back(off);
        std::sort(todo.begin(), todo.end());
        std::vector<ChunkImage> built(todo.size());
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < todo.size();) ChunkTranscoder(b, built[i]).run(todo[i]);
        };
        const unsigned n_workers = std::max(1u, std::min<unsigned>(threads, unsigned(todo.size() / 16 + 1)));
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < n_workers; ++t) pool.emplace_back(worker);
        worker();
        for (auto& t : pool) t.join();
        for (size_t i = 0; i < todo.size(); ++i) {
            Placed pl;
            pl.words = built[i].words.size();
            pl.at = alloc_.alloc(pl.words);
            pl.masks = built[i].root.packed();
            pl.csvo_end = built[i].csvo_end;
            write(pl.at, built[i].words.data(), pl.words);
            chunks_[todo[i]] = pl;
        }

        // 4. the root octree is rewritten by every commit (csvo.rs:68-139 re-serializes it): so is its image
        alloc_.release(root_at_, root_words_);
        for (auto& r : chunk_refs) {
            const Placed& pl = chunks_.at(r.second);
            const size_t body = r.first;
            root_words[body] = uint32_t(pl.at);  // absolute index of the chunk's root octant (bit 31 clear, esvo.rs:164-171)
            // the chunk's masks go into the header half-word of the same child slot of the same octant
            const size_t oct = (body / 12) * 12, c = body - oct - 4;
            root_words[oct + (c >> 1)] |= pl.masks << ((c & 1u) * 16);
        }
        root_words_ = root_words.size();
        root_at_ = alloc_.alloc(root_words_);
        write(root_at_, root_words.data(), root_words_);
        const uint32_t preamble[5] = {root_masks.packed(), 0, 0, 0, uint32_t(root_at_)};
        write(0, preamble, 5);
        if (frame_[0] != scale_bits) {
            frame_[0] = scale_bits;
            dirty_.push_back(Range{0, 4});
        }
        return (alloc_.end() + 1) * 4 < (uint64_t(1) << 32) - 64;
    }

private:
    struct Placed {
        uint64_t at = 0, words = 0, csvo_end = 0;
        uint32_t masks = 0;
    };

    // Root octree nodes are internal nodes (their depth is always above a chunk's); a 4-byte entry with bit 31 is the frame
    // offset of a chunk (csvo.rs:76-86,100-105). Octants are appended to `out` with relative pointers; chunk slots are
    // recorded and patched by the caller once the chunks have been placed.
    NodeMasks root_node(const Bytes& b, uint64_t ptr, uint32_t depth, std::vector<uint32_t>& out, std::vector<std::pair<size_t, uint32_t>>& refs) {
        const size_t at = out.size();
        out.resize(at + 12, 0u);
        NodeMasks m;
        if (depth <= 3 || depth > 32) return m;
        const uint32_t header = b.u16(ptr);
        const uint32_t table = tag_bytes(header);
        for (uint32_t c = 0; c < 8; ++c) {
            const uint32_t tag = (header >> (2 * c)) & 3u;
            if (!tag) continue;
            const uint32_t offset = tag_bytes(header & ((1u << (2 * c)) - 1u)), width = tag == 3 ? 4u : tag;
            uint32_t e = 0;
            for (uint32_t k = 0; k < width; ++k) e |= b.u8(ptr + 2 + offset + k) << (8 * k);
            m.child_mask |= 1u << c;
            if (e & 0x80000000u) {
                refs.emplace_back(at + 4 + c, e ^ 0x80000000u);
            } else {
                const size_t child_at = out.size();
                const NodeMasks cm = root_node(b, ptr + 2 + table + e, depth - 1, out, refs);
                out[at + (c >> 1)] |= cm.packed() << ((c & 1u) * 16);
                out[at + 4 + c] = 0x80000000u | uint32_t(child_at - (at + 4 + c));
            }
        }
        return m;
    }

    void write(uint64_t at, const uint32_t* src, uint64_t n) {
        if (!n) return;
        if (frame_.size() < 1 + at + n) frame_.resize(1 + at + n, 0u);
        std::memcpy(frame_.data() + 1 + at, src, n * 4);
        dirty_.push_back(Range{(1 + at) * 4, n * 4});
    }

    std::vector<uint32_t> frame_;
    std::vector<Range> dirty_;
    WordAllocator alloc_;
    std::unordered_map<uint32_t, Placed> chunks_;
    uint64_t root_at_ = 0, root_words_ = 0;
};

}  // namespace vximg
