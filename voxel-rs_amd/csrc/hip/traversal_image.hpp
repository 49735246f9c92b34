// World -> traversal image. The reference's node formats make a descent expensive: CSVO (src/world/hds/csvo.rs:434-546) costs
// a bit-field decode, two popcount sums and a dependent table read, ESVO (esvo.rs:74-101) two loads, a relative/absolute
// pointer resolve and a half-word select. A world is therefore re-laid out, on the host at commit time, as fixed-size octants
// -- the "traversal image" -- and rays walk the image. Structure is preserved node for node (including the empty octants the
// reference's never-compacted root octree carries), so every ray that starts outside a voxel takes the same iterations to the
// same leaf with the same floats. A ray that starts INSIDE a voxel makes the reference wander through leaf data as if they
// were nodes, which is format specific: the kernel re-renders exactly those pixels on the original bytes (render_persistent,
// kForeign).
//
// Two encodings of the same octant tree:
//   kOct64   what the renderer walks (the name is round 2's, when an octant took 64 bytes; since round 6 it takes what its children take). Frame = 64-byte
//            header [f32 2^-depth][u32 root masks][u32 `lo` of the root octant][0...] followed by octants. Everything is addressed in 8-byte UNITS from the
//            frame start. A node's octant holds one {lo, hi} entry (one unit) per EXISTING child, the children in reverse order (child 7 first): lo = where
//            the child's own octant lies -- the unit BEFORE its first entry --, or the leaf's value; hi = the child's masks (oct64_masks()). With m = the
//            node's masks shifted left by the child's index (which the traversal forms anyway: "is a leaf" in the sign, "exists" in bit 23, below it the "exists" bits of the children above), the entry of that child is unit
//            lo + popcount(m & 0xffffff): a descent is ONE aligned 8-byte load that yields the new pointer and the new masks, every pointer in the image is valid by
//            construction. An octant ALL of whose children are leaves -- the majority: every voxel's parent -- holds 4-byte values only, again for the existing
//            children in reverse order from unit lo + 1 on, padded to whole units (its masks, which live in its parent's entry, say so: child bits == leaf
//            bits); in the image of a CSVO world unit `lo` itself holds where that leaf-mask byte lies in the world's own bytes (Octant::origin: what a ray
//            that is led into a voxel needs, vx_device.hpp walk_voxel_on_bytes). An octant without children (the reference's root octree keeps such) takes
//            no room. For a surface shell about half of what eight entries per octant took (rounds 2-5: 64 / 32 bytes an octant plus a separate origin table):
//            0.4 x the bytes of an ESVO world, 2 x those of a CSVO world.
//   kOct64Wide  the same bytes, walked through a 64-bit pointer instead of a buffer resource (whose offsets end at 4 GiB): images from 4 GiB up to 32 GiB,
//            at two more instructions per descent. A context switches to it when kOct64 no longer fits.
//   kEsvo48  the reference's ESVO format: [f32][5-word preamble][12-word octants], relative pointers. Kept because any ESVO
//            traversal can walk it: tests/test_traversal_image.py checks the tree walk with the oracle.
// The image never has more levels than the world's depth says (a world that breaks this is not imaged): the kernel relies on it.
// Host-only, no HIP calls: vx_api.hip owns the device side.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <sys/mman.h>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace vximg {

// The image's words on the host: a growing array whose new part reads as zero WITHOUT being written -- anonymous pages, extended with
// mremap. (A std::vector zero-fills what it grows by on the calling thread: 7 GB for the depth-14 terrain, two seconds of a whole-world
// commit; here the kernel zeroes a page when one of the encoding threads first touches it.)
class ZeroedWords {
public:
    ZeroedWords() = default;
    ZeroedWords(const ZeroedWords& o) { *this = o; }
    ZeroedWords(ZeroedWords&& o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
    ZeroedWords& operator=(const ZeroedWords& o) {
        if (this != &o) {
            clear();
            resize(o.n_, 0u);
            if (o.n_) std::memcpy(p_, o.p_, o.n_ * 4);
        }
        return *this;
    }
    ZeroedWords& operator=(ZeroedWords&& o) noexcept {
        if (this != &o) {
            clear();
            p_ = o.p_; n_ = o.n_; cap_ = o.cap_;
            o.p_ = nullptr; o.n_ = o.cap_ = 0;
        }
        return *this;
    }
    ~ZeroedWords() { clear(); }
    const uint32_t* data() const { return p_; }
    uint32_t* data() { return p_; }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    uint32_t& operator[](size_t i) { return p_[i]; }
    const uint32_t& operator[](size_t i) const { return p_[i]; }
    void clear() {
        if (p_) ::munmap(p_, cap_ * 4);
        p_ = nullptr;
        n_ = cap_ = 0;
    }
    void assign(size_t n, uint32_t /*zero*/) {
        clear();
        resize(n, 0u);
    }
    // address space for n words (untouched pages cost nothing): a whole world's image grows to its size without being moved on the way
    void reserve(size_t n) {
        if (n <= cap_) return;
        try {
            grow_to(n);
        } catch (const std::bad_alloc&) {
            // (a hint: where the address space is not to be had up front the frame grows in steps as before)
        }
    }
    // grows only (the image never shrinks between clear()s); the value is always zero
    void resize(size_t n, uint32_t /*zero*/) {
        if (n <= n_) return;
        if (n > cap_) grow_to(std::max(n, cap_ + cap_ / 2));
        n_ = n;
    }

private:
    void grow_to(size_t want) {
        want = (want * 4 + 4095) / 4096 * 4096 / 4;  // whole pages
        void* q = p_ ? ::mremap(p_, cap_ * 4, want * 4, MREMAP_MAYMOVE) : ::mmap(nullptr, want * 4, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (q == MAP_FAILED) throw std::bad_alloc();
        p_ = static_cast<uint32_t*>(q);
        cap_ = want;
        // (gigabytes that the encoding threads touch for the first time: as huge pages where the system gives them -- a hint, ignored otherwise)
        (void)::madvise(p_, cap_ * 4, MADV_HUGEPAGE);
    }

    uint32_t* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;  // words
};


enum Layout : int { kEsvo48 = 0, kOct64 = 1, kOct64Wide = 2 };

// The CPUs this process may keep busy: the logical CPUs, or what a cgroup's quota grants of them (v2 cpu.max, v1 cfs quota / period). A container
// that sees 256 logical CPUs behind a quota of 16 is throttled for every thread beyond the sixteenth (measured with the oracle: 32 threads 8 % below 16,
// 128 threads half of it).
inline unsigned granted_cpus() {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    auto read_pair = [](const char* path, double& a, double& b) -> bool {
        FILE* f = std::fopen(path, "r");
        if (!f) return false;
        char first[64] = {0};
        double second = 0.0;
        const int got = std::fscanf(f, "%63s %lf", first, &second);
        std::fclose(f);
        if (got < 1 || first[0] == 'm') return false;  // "max": no quota
        a = std::atof(first);
        b = second;
        return got == 2;
    };
    double quota = 0.0, period = 0.0;
    if (read_pair("/sys/fs/cgroup/cpu.max", quota, period) && quota > 0.0 && period > 0.0) {
        n = std::min(n, std::max(1u, unsigned(quota / period + 0.5)));
    } else {
        double q = 0.0, unused = 0.0, pr = 0.0;
        FILE* f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
        if (f) { if (std::fscanf(f, "%lf", &q) != 1) q = 0.0; std::fclose(f); }
        f = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (f) { if (std::fscanf(f, "%lf", &pr) != 1) pr = 0.0; std::fclose(f); }
        (void)unused;
        if (q > 0.0 && pr > 0.0) n = std::min(n, std::max(1u, unsigned(q / pr + 0.5)));
    }
    return n;
}

struct Range {
    uint64_t start, length;
};

// byte reader over descriptors[] (world + 8), reads beyond the end return 0 like the traversal's
struct Bytes {
    const uint8_t* p;
    size_t n;
    uint32_t u8(uint64_t o) const { return o < n ? p[o] : 0u; }
    uint32_t u16(uint64_t o) const {
        if (o + 2 <= n) return uint32_t(p[o]) | (uint32_t(p[o + 1]) << 8);
        return u8(o) | (u8(o + 1) << 8);
    }
    uint32_t u32(uint64_t o) const {
        if (o + 4 <= n) {
            uint32_t v;
            std::memcpy(&v, p + o, 4);  // (little-endian hosts, like the buffer itself)
            return v;
        }
        return u16(o) | (u16(o + 2) << 16);
    }
};

inline uint32_t tag_bytes(uint32_t m) {  // bytes of the table entries a 2-bit-per-child mask selects: tag 0,1,2,3 -> 0,1,2,4
    const uint32_t lo = m & 0x5555u, hi = (m >> 1) & 0x5555u;  // tag = hi:lo per child
    return uint32_t(__builtin_popcount(lo & ~hi)) + 2u * uint32_t(__builtin_popcount(hi & ~lo)) + 4u * uint32_t(__builtin_popcount(lo & hi));
}

struct NodeMasks {
    uint32_t child_mask = 0, leaf_mask = 0;
    uint32_t packed() const { return (child_mask << 8) | leaf_mask; }
};

// kOct64's form of packed masks: child c's "is a leaf" bit at 31 - c and its "exists" bit at 23 - c, nothing in the lower half. The traversal shifts the
// word left by the child's index: "leaf" of that child is then the sign, "exists" bit 23, and bits 22..0 hold the "exists" bits of the children ABOVE it and
// nothing else (the leaf bits have moved up, the lower half was empty) -- so popcount(m & 0xffffff) counts this child and the existing ones above it, which is
// where its entry lies in the octant (entries of the existing children only, child 7 first). (Rounds 2-5 had the two bytes the other way round; with "exists"
// in the upper byte a shift brings leaf bits in below it and the count needs an instruction more.)
// A leaf bit without its child bit means nothing to the traversal (svo.esvo.glsl:168-173) and is dropped: "all children are
// leaves" must read the same on the device (child bits == leaf bits) as here (oct64_words()).
inline uint32_t oct64_masks(uint32_t packed) {
    auto reversed = [](uint32_t b) {  // bit 0 <-> bit 7, ...
        b = ((b & 0xf0u) >> 4) | ((b & 0x0fu) << 4);
        b = ((b & 0xccu) >> 2) | ((b & 0x33u) << 2);
        return ((b & 0xaau) >> 1) | ((b & 0x55u) << 1);
    };
    const uint32_t children = (packed >> 8) & 0xffu, leaves = packed & children;
    return (reversed(leaves) << 24) | (reversed(children) << 16);
}

// One node of the walked tree, layout independent. Per child: nothing, a leaf (value), a node (index of its octant in the
// same tree) or -- root octree only -- a chunk (its frame offset in the CSVO bytes).
struct Octant {
    uint32_t lo[8] = {};
    uint16_t masks[8] = {};  // the child's own masks (nodes), filled in for chunks at placement
    uint8_t node_mask = 0, leaf_mask = 0, chunk_mask = 0;
    // CSVO voxel parents (leaf-mask bytes, svo.csvo.glsl:114-115): where the byte is in the world, for the unit in front of the octant's values --
    // [0] = its byte pointer L, [1] = k << 29 | (L - the chunk's material section), k = its place among its depth-2 parent's bytes
    uint32_t origin[2] = {};
};

// frame words an octant takes in the kOct64 layouts: a {pointer | value, masks} entry per existing child; or (all children leaves) their values, padded to
// whole 8-byte units, behind the unit of the origin where the image keeps one; or nothing
inline uint32_t oct64_words(const Octant& o, bool with_origin) {
    if (o.node_mask | o.chunk_mask) return 2u * uint32_t(__builtin_popcount(uint32_t(o.node_mask | o.chunk_mask | o.leaf_mask)));
    if (!o.leaf_mask) return 0;
    return ((uint32_t(__builtin_popcount(uint32_t(o.leaf_mask))) + 1u) & ~1u) + (with_origin ? 2u : 0u);
}
// ... and its `lo` (what an entry that points to it holds) when it is placed at frame word `at`: the unit before its first entry / value
inline uint32_t oct64_lo(const Octant& o, uint64_t at, bool with_origin) {
    const bool values_only = !(o.node_mask | o.chunk_mask) && o.leaf_mask;
    return uint32_t(at / 2) - ((values_only && with_origin) ? 0u : 1u);
}

struct Tree {
    std::vector<Octant> octants;  // octant 0 = the root of this tree
    NodeMasks root;
    bool too_deep = false;        // more levels than the slot the tree hangs in leaves it (or a runaway walk)
    uint64_t src_begin = 0, src_end = 0;  // chunks: the arena bytes the chunk was read from
};

// A chunk hanging in the root octree: `key` identifies its bytes (CSVO: arena offset of its frame; ESVO: descriptors[] index of
// its root octant), `levels` is what the slot leaves it, `masks` its root's masks where the format keeps them outside the chunk.
struct ChunkRef {
    uint32_t key, masks, levels;
};

// One chunk frame [lod:u8][material_bytes:u32][materials][nodes] (csvo.rs:217-227).
class ChunkWalker {
public:
    ChunkWalker(Bytes b, Tree& out) : b_(b), out_(out) {}

    void run(uint64_t frame) {
        const uint32_t lod = b_.u8(frame);
        const uint32_t material_bytes = b_.u32(frame + 1);
        materials_ = frame + 5;
        end_ = materials_ + material_bytes;
        out_.octants.clear();
        out_.root = node(materials_ + material_bytes, lod, 0);
        out_.src_begin = frame;
        out_.src_end = end_;
    }

private:
    NodeMasks node(uint64_t ptr, uint32_t depth, uint64_t pre_leaf) {
        const size_t at = out_.octants.size();
        out_.octants.emplace_back();
        NodeMasks m;
        if (depth == 0 || depth > 32) return m;  // malformed: an octant without children
        if (depth == 1) {
            // a leaf-mask byte inside its depth-2 parent: the children are voxels (read_leaf, svo.csvo.glsl:119-133)
            const uint32_t mask = b_.u8(ptr);
            end_ = std::max(end_, ptr + 1);
            const uint32_t material_offset = b_.u16(pre_leaf + 1);
            const uint64_t leaf_index = ptr - (pre_leaf + 3);
            if (leaf_index > 7 || ptr >= (uint64_t(1) << 31) || ptr - materials_ >= (uint64_t(1) << 29)) {
                out_.too_deep = true;  // not a chunk the serializer wrote (csvo.rs:481-493): the origin could not say where it is
                return m;
            }
            out_.octants[at].origin[0] = uint32_t(ptr);
            out_.octants[at].origin[1] = uint32_t(leaf_index << 29) | uint32_t(ptr - materials_);
            uint32_t preceding = 0;  // leaves before this byte under the depth-2 node
            for (uint64_t k = 0; k < leaf_index; ++k) preceding += uint32_t(__builtin_popcount(b_.u8(pre_leaf + 3 + k)));
            for (uint32_t c = 0; c < 8; ++c) {
                if (!((mask >> c) & 1u)) continue;
                const uint32_t before = preceding + uint32_t(__builtin_popcount(mask & ((1u << c) - 1u)));
                out_.octants[at].lo[c] = b_.u32(materials_ + uint64_t(material_offset) * 4 + uint64_t(before) * 4);
            }
            out_.octants[at].leaf_mask = uint8_t(mask);
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
                child = ptr + 2 + table + e;  // (an absolute pointer, bit 31, only exists in the root octree: RootWalker)
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
            const size_t child_at = out_.octants.size();
            const NodeMasks cm = node(child, depth - 1, depth == 2 ? ptr : pre_leaf);
            m.child_mask |= 1u << c;
            Octant& o = out_.octants[at];
            o.node_mask |= uint8_t(1u << c);
            o.lo[c] = uint32_t(child_at);
            o.masks[c] = uint16_t(cm.packed());
        }
        return m;
    }

    Bytes b_;
    Tree& out_;
    uint64_t materials_ = 0, end_ = 0;
};

// Root octree: internal nodes only (their depth is always above a chunk's); a 4-byte entry with bit 31 is the frame offset of
// a chunk (csvo.rs:76-86,100-105).
inline NodeMasks walk_root(const Bytes& b, uint64_t ptr, uint32_t depth, Tree& out, std::vector<ChunkRef>& refs) {
    const size_t at = out.octants.size();
    out.octants.emplace_back();
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
            out.octants[at].chunk_mask |= uint8_t(1u << c);
            out.octants[at].lo[c] = e ^ 0x80000000u;
            // a child of a node at `depth` has depth - 1 levels to itself: the chunk's lod byte says how many it uses
            if (b.u8(e ^ 0x80000000u) > depth - 1) out.too_deep = true;
            refs.push_back(ChunkRef{e ^ 0x80000000u, 0u, depth - 1});
        } else {
            const size_t child_at = out.octants.size();
            const NodeMasks cm = walk_root(b, ptr + 2 + table + e, depth - 1, out, refs);
            Octant& o = out.octants[at];
            o.node_mask |= uint8_t(1u << c);
            o.lo[c] = uint32_t(child_at);
            o.masks[c] = uint16_t(cm.packed());
        }
    }
    return m;
}

// ---- ESVO source (esvo.rs:74-101): descriptors[] = [5-word preamble][arena of 12-word octants] ----
struct Words {
    const uint32_t* p;
    size_t n;
    uint32_t at(uint64_t i) const { return i < n ? p[i] : 0u; }  // reads beyond the end return 0 like the traversal's
};

// Walks the subtree of the octant at descriptors[] index `octant`; its masks (child_mask << 8 | leaf_mask) live in its parent's
// header (esvo.rs:74-86) and are handed in. `levels` = levels below this node (1 = its children can only be voxels). With
// `refs`, absolute pointers end the tree (the root octree: they are the chunks, esvo.rs:164-171); without, they are followed.
class EsvoWalker {
public:
    EsvoWalker(Words w, Tree& out, std::vector<ChunkRef>* refs) : w_(w), out_(out), refs_(refs) {}

    void run(uint64_t octant, uint32_t masks, uint32_t levels) {
        out_.octants.clear();
        lo_ = ~uint64_t(0);
        hi_ = 0;
        out_.root.child_mask = (masks >> 8) & 0xffu;
        out_.root.leaf_mask = masks & 0xffu;
        node(octant, masks, levels);
        // descriptors[] index -> arena byte (the preamble's 5 words come first)
        out_.src_begin = lo_ == ~uint64_t(0) ? 0 : (lo_ < 5 ? 0 : (lo_ - 5) * 4);
        out_.src_end = hi_ < 5 ? 0 : (hi_ - 5) * 4;
    }

private:
    void node(uint64_t octant, uint32_t masks, uint32_t levels) {
        const size_t at = out_.octants.size();
        out_.octants.emplace_back();
        if (at > kMaxOctants) {  // shared or cyclic subtrees would never end: not a world the serializer wrote
            out_.too_deep = true;
            return;
        }
        lo_ = std::min(lo_, octant);
        hi_ = std::max(hi_, octant + 12);
        const uint32_t child_mask = (masks >> 8) & 0xffu, leaf_mask = masks & 0xffu;
        for (uint32_t c = 0; c < 8 && !out_.too_deep; ++c) {
            if (!((child_mask >> c) & 1u)) continue;
            const uint32_t body = w_.at(octant + 4 + c);
            if ((leaf_mask >> c) & 1u) {
                // A voxel has no masks of its own in anything the serializer writes (esvo.rs:465-485 ORs a child's masks into the header
                // only for octants), so a ray led INTO a voxel walks an empty node (svo.esvo.glsl:183-185) -- which is what the image's
                // traversal does for every voxel. A world where that is not so is traversed as bytes.
                if ((w_.at(octant + (c >> 1)) >> ((c & 1u) * 16)) & 0xffffu) {
                    out_.too_deep = true;
                    return;
                }
                out_.octants[at].leaf_mask |= uint8_t(1u << c);
                out_.octants[at].lo[c] = body;
                continue;
            }
            if (levels <= 1) {  // a node where only voxels fit
                out_.too_deep = true;
                return;
            }
            const uint32_t cm = (w_.at(octant + (c >> 1)) >> ((c & 1u) * 16)) & 0xffffu;
            const bool relative = (body & 0x80000000u) != 0;
            const uint64_t target = relative ? octant + 4 + c + (body & 0x7fffffffu) : body;  // svo.esvo.glsl:283-290
            if (!relative && refs_) {
                out_.octants[at].chunk_mask |= uint8_t(1u << c);
                out_.octants[at].lo[c] = uint32_t(target);
                refs_->push_back(ChunkRef{uint32_t(target), cm, levels - 1});
                continue;
            }
            const size_t child_at = out_.octants.size();
            node(target, cm, levels - 1);
            Octant& o = out_.octants[at];
            o.node_mask |= uint8_t(1u << c);
            o.lo[c] = uint32_t(child_at);
            o.masks[c] = uint16_t(cm);
        }
    }

    static constexpr size_t kMaxOctants = size_t(1) << 26;  // (a runaway walk: no world the serializer wrote has as many)
    Words w_;
    Tree& out_;
    std::vector<ChunkRef>* refs_;
    uint64_t lo_ = 0, hi_ = 0;
};

// ---- chunks straight into image words (the kOct64 layouts) ----------------------------------------------------------------------------
//
// A whole world's first commit walks 400,000 chunks; through Tree (60 bytes an octant, written by the walk, read again by encode()) that was a
// second and the larger half of the commit. The emitters below make the same walk -- same order, same checks, same words: octants in depth-first
// order, a node's octant in front of its children's -- and write each octant's words when they meet it, relative to the chunk's first word; `relocs`
// lists the words that hold a `lo` pointing into the chunk, to which the copy into the frame adds where the chunk was placed. (The root octree and
// the kEsvo48 layout still go through Tree: ChunkWalker / EsvoWalker + encode() say the same thing in the other form, and
// tests/test_traversal_image.py holds the two against each other.)
struct Emitted {
    std::vector<uint32_t> words, relocs;  // (room to write into, grown by room(): what counts is n_words / n_relocs, not their size())
    size_t n_words = 0, n_relocs = 0;
    NodeMasks root;
    bool too_deep = false;
    uint64_t src_begin = 0, src_end = 0;
    void start() {
        n_words = n_relocs = 0;
        root = NodeMasks();
        too_deep = false;
        src_begin = src_end = 0;
    }
    // room for `extra` more words and `extra` more relocs (a node asks once for itself and what it writes for its children)
    void room(size_t extra) {
        if (n_words + extra > words.size()) words.resize(std::max(words.size() * 2, n_words + extra + 4096));
        if (n_relocs + extra > relocs.size()) relocs.resize(std::max(relocs.size() * 2, n_relocs + extra + 1024));
    }
};

inline uint32_t entries_above(uint32_t mask, uint32_t c) { return uint32_t(__builtin_popcount((mask & 0xffu) >> (c + 1u))); }  // entries lie child 7 first

// oct64_masks() of a node whose children are all voxels, by its leaf-mask byte
inline const uint32_t* oct64_voxel_parent_masks() {
    static const struct Table {
        uint32_t m[256];
        Table() { for (uint32_t i = 0; i < 256; ++i) m[i] = oct64_masks((i << 8) | i); }
    } table;
    return table.m;
}

// One chunk frame [lod:u8][material_bytes:u32][materials][nodes] (csvo.rs:217-227): ChunkWalker's walk.
class ChunkEmitter {
public:
    ChunkEmitter(Bytes b, Emitted& out) : b_(b), out_(out), voxel_parent_masks_(oct64_voxel_parent_masks()) {}

    void run(uint64_t frame) {
        out_.start();
        const uint32_t lod = b_.u8(frame);
        const uint32_t material_bytes = b_.u32(frame + 1);
        materials_ = frame + 5;
        end_ = materials_ + material_bytes;
        out_.root = node(materials_ + material_bytes, lod);
        out_.src_begin = frame;
        out_.src_end = end_;
    }

private:
    static constexpr size_t kMaxWords = size_t(1) << 26;  // (a runaway walk: no chunk the serializer wrote comes near)

    NodeMasks node(uint64_t ptr, uint32_t depth) {
        NodeMasks m;
        if (depth == 0 || depth > 32 || out_.too_deep) return m;  // malformed: an octant without children
        if (depth == 1) {
            // a leaf-mask byte without a depth-2 parent (a chunk of one level): nothing the serializer writes (csvo.rs:481-493), and no origin could
            // say where it is
            out_.too_deep = true;
            return m;
        }
        if (out_.n_words > kMaxWords) {
            out_.too_deep = true;
            return m;
        }
        out_.room(2 * 8 + 8 * (2 + 8) + 2);
        const size_t at = out_.n_words;
        if (depth == 2) {
            // leaf node: u8 mask, u16 material offset, one leaf-mask byte per child (svo.csvo.glsl:114-115); each of those is a voxel parent (read_leaf,
            // svo.csvo.glsl:119-133) whose octant is [origin][the existing children's values, child 7 first][padding] -- the values of all of them lie
            // one after the other, in child order, in the material section
            const uint32_t header = b_.u8(ptr), material_offset = b_.u16(ptr + 1);
            const uint32_t n_children = uint32_t(__builtin_popcount(header));
            end_ = std::max(end_, ptr + 3 + n_children);
            if (ptr + 3 + n_children > (uint64_t(1) << 31) || ptr + 3 + n_children - materials_ > (uint64_t(1) << 29)) {
                out_.too_deep = true;  // (the origin could not say where the bytes are)
                return m;
            }
            uint32_t* w = out_.words.data();
            uint32_t* relocs = out_.relocs.data();
            size_t top = at + 2u * n_children;
            uint64_t value = materials_ + uint64_t(material_offset) * 4;
            uint32_t entry = uint32_t(at) + 2u * n_children;  // (child 7 first: the entries are written from the back)
            for (uint32_t j = 0; j < n_children; ++j) {
                const uint64_t byte = ptr + 3 + j;
                const uint32_t mask = b_.u8(byte);
                entry -= 2u;
                w[entry] = uint32_t(top / 2) - (mask ? 0u : 1u);  // (an octant of values is pointed to by the unit of its origin: oct64_lo)
                w[entry + 1] = voxel_parent_masks_[mask];
                relocs[out_.n_relocs++] = entry;
                if (!mask) continue;
                const uint32_t n = uint32_t(__builtin_popcount(mask));
                w[top] = uint32_t(byte);
                w[top + 1] = (j << 29) | uint32_t(byte - materials_);
                uint32_t* values = w + top + 2;
                values[n] = 0u;  // (the padding, where n is odd)
                if (value + 32 <= b_.n) {
                    uint32_t in[8];
                    std::memcpy(in, b_.p + value, 32);
                    for (uint32_t k = 0; k < n; ++k) values[n - 1 - k] = in[k];
                } else {
                    for (uint32_t k = 0; k < n; ++k) values[n - 1 - k] = b_.u32(value + uint64_t(k) * 4);
                }
                value += uint64_t(n) * 4;
                top += 2u + ((n + 1u) & ~1u);
            }
            out_.n_words = top;
            m.child_mask = header;
            return m;
        }
        // which children there are, before any of them is walked: the octant's entries come first
        uint32_t header, present = 0;
        if (depth > 3) {  // internal node: u16 header, 2 bits per child, 1/2/4-byte forward offsets (svo.csvo.glsl:56-97)
            header = b_.u16(ptr);
            const uint32_t any = (header | (header >> 1)) & 0x5555u;
            for (uint32_t c = 0; c < 8; ++c) present |= ((any >> (2 * c)) & 1u) << c;
        } else {  // pre-leaf node: u8 mask, u8 offsets (svo.csvo.glsl:107-112)
            header = b_.u8(ptr);
            present = header;
        }
        out_.n_words = at + 2u * uint32_t(__builtin_popcount(present));
        const uint32_t table = depth > 3 ? tag_bytes(header) : uint32_t(__builtin_popcount(header));
        end_ = std::max(end_, ptr + (depth > 3 ? 2u : 1u) + (present ? table : 0u));
        for (uint32_t c = 0; c < 8; ++c) {
            if (!((present >> c) & 1u)) continue;
            uint64_t child = 0;
            if (depth > 3) {
                const uint32_t tag = (header >> (2 * c)) & 3u;
                const uint32_t offset = tag_bytes(header & ((1u << (2 * c)) - 1u));
                const uint32_t width = tag == 3 ? 4u : tag;
                uint32_t e = 0;
                for (uint32_t k = 0; k < width; ++k) e |= b_.u8(ptr + 2 + offset + k) << (8 * k);
                child = ptr + 2 + table + e;  // (an absolute pointer, bit 31, only exists in the root octree: walk_root)
            } else {
                child = ptr + 1 + table + b_.u8(ptr + 1 + uint32_t(__builtin_popcount(header & ((1u << c) - 1u))));
            }
            const size_t child_at = out_.n_words;
            const NodeMasks cm = node(child, depth - 1);
            m.child_mask |= 1u << c;
            const size_t entry = at + 2u * entries_above(present, c);
            out_.words[entry] = uint32_t(child_at / 2) - 1u;
            out_.words[entry + 1] = oct64_masks(cm.packed());
            out_.relocs[out_.n_relocs++] = uint32_t(entry);
        }
        return m;
    }

    Bytes b_;
    Emitted& out_;
    const uint32_t* voxel_parent_masks_;
    uint64_t materials_ = 0, end_ = 0;
};

// The subtree of an ESVO octant (esvo.rs:74-101), absolute pointers followed: EsvoWalker's walk of a chunk.
class EsvoEmitter {
public:
    EsvoEmitter(Words w, Emitted& out) : w_(w), out_(out) {}

    void run(uint64_t octant, uint32_t masks, uint32_t levels) {
        out_.start();
        lo_ = ~uint64_t(0);
        hi_ = 0;
        octants_ = 0;
        out_.root.child_mask = (masks >> 8) & 0xffu;
        out_.root.leaf_mask = masks & 0xffu;
        node(octant, masks, levels);
        out_.src_begin = lo_ == ~uint64_t(0) ? 0 : (lo_ < 5 ? 0 : (lo_ - 5) * 4);
        out_.src_end = hi_ < 5 ? 0 : (hi_ - 5) * 4;
    }

private:
    void node(uint64_t octant, uint32_t masks, uint32_t levels) {
        if (octants_++ > kMaxOctants) {  // shared or cyclic subtrees would never end: not a world the serializer wrote
            out_.too_deep = true;
            return;
        }
        lo_ = std::min(lo_, octant);
        hi_ = std::max(hi_, octant + 12);
        const uint32_t child_mask = (masks >> 8) & 0xffu, leaf_mask = masks & child_mask;
        // the octant's twelve words (reads beyond the end return 0 like the traversal's)
        uint32_t o[12];
        if (octant + 12 <= w_.n) std::memcpy(o, w_.p + octant, 48);
        else for (uint32_t k = 0; k < 12; ++k) o[k] = w_.at(octant + k);
        out_.room(16);
        const size_t at = out_.n_words;
        const uint32_t n = uint32_t(__builtin_popcount(child_mask));
        if (child_mask == leaf_mask) {
            // voxels only: their values, child 7 first, padded to a unit
            uint32_t* v = out_.words.data() + at;
            uint32_t k = 0;
            for (int c = 7; c >= 0; --c) {
                if (!((child_mask >> c) & 1u)) continue;
                // (a voxel with masks of its own: a world the image's traversal would walk differently -- EsvoWalker::node)
                if ((o[c >> 1] >> ((c & 1) * 16)) & 0xffffu) out_.too_deep = true;
                v[k++] = o[4 + c];
            }
            v[k] = 0u;
            out_.n_words = at + ((n + 1u) & ~1u);
            return;
        }
        out_.n_words = at + 2u * n;
        for (uint32_t c = 0; c < 8 && !out_.too_deep; ++c) {
            if (!((child_mask >> c) & 1u)) continue;
            const uint32_t body = o[4 + c];
            const size_t entry = at + 2u * entries_above(child_mask, c);
            const uint32_t cm = (o[c >> 1] >> ((c & 1u) * 16)) & 0xffffu;
            if ((leaf_mask >> c) & 1u) {
                if (cm) {
                    out_.too_deep = true;
                    return;
                }
                out_.words[entry] = body;
                out_.words[entry + 1] = 0u;
                continue;
            }
            if (levels <= 1) {  // a node where only voxels fit
                out_.too_deep = true;
                return;
            }
            const bool relative = (body & 0x80000000u) != 0;
            const uint64_t target = relative ? octant + 4 + c + (body & 0x7fffffffu) : body;  // svo.esvo.glsl:283-290
            const size_t child_at = out_.n_words;
            node(target, cm, levels - 1);
            out_.words[entry] = uint32_t(child_at / 2) - 1u;
            out_.words[entry + 1] = oct64_masks(cm);
            out_.relocs[out_.n_relocs++] = uint32_t(entry);
        }
    }

    static constexpr size_t kMaxOctants = size_t(1) << 26;
    Words w_;
    Emitted& out_;
    uint64_t lo_ = 0, hi_ = 0;
    size_t octants_ = 0;
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

// Worker threads that live as long as one update(): a whole world is walked and encoded in a hundred batches, and a thread per worker and batch
// was a fifth of a second of thread creation on 48 workers.
class Workers {
public:
    explicit Workers(unsigned n) {
        for (unsigned t = 1; t < n; ++t) pool_.emplace_back([this] { loop(); });
    }
    ~Workers() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : pool_) t.join();
    }
    unsigned size() const { return unsigned(pool_.size()) + 1u; }
    // f(i) for i in [0, n), on at most `limit` of the workers (the caller's thread is one of them)
    void run(size_t n, unsigned limit, const std::function<void(size_t)>& f) {
        if (n == 0) return;
        {
            std::lock_guard<std::mutex> lk(m_);
            f_ = &f;
            n_ = n;
            next_.store(0);
            active_ = 0;
            slots_ = limit > 1 ? limit - 1 : 0;
            ++generation_;
        }
        cv_.notify_all();
        // An exception on any thread (bad_alloc while a walker grows its tree) ends the round early -- nobody takes another index -- and
        // is rethrown HERE, on the caller's thread, once every worker has left f: the caller's vectors outlive the workers' use of them and a
        // commit that runs out of memory fails with an error code instead of std::terminate.
        try {
            for (size_t i; (i = next_.fetch_add(1)) < n;) f(i);
        } catch (...) {
            note_failure(n);
        }
        std::exception_ptr thrown;
        {
            std::unique_lock<std::mutex> lk(m_);
            slots_ = 0;  // (nobody else joins this round)
            done_.wait(lk, [&] { return active_ == 0; });
            f_ = nullptr;
            thrown = failure_;
            failure_ = nullptr;
        }
        if (thrown) std::rethrow_exception(thrown);
    }

private:
    void loop() {
        unsigned seen = 0;
        for (;;) {
            const std::function<void(size_t)>* f = nullptr;
            size_t n = 0;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || (generation_ != seen && slots_ > 0); });
                if (stop_) return;
                seen = generation_;
                --slots_;
                ++active_;
                f = f_;
                n = n_;
            }
            try {
                for (size_t i; (i = next_.fetch_add(1)) < n;) (*f)(i);
            } catch (...) {
                note_failure(n);
            }
            {
                std::lock_guard<std::mutex> lk(m_);
                --active_;
            }
            done_.notify_one();
        }
    }
    // (inside a catch block) keeps the first exception of the round and stops the hand-out of indices
    void note_failure(size_t n) {
        std::lock_guard<std::mutex> lk(m_);
        if (!failure_) failure_ = std::current_exception();
        next_.store(n);
    }
    std::vector<std::thread> pool_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::exception_ptr failure_;
    const std::function<void(size_t)>* f_ = nullptr;
    size_t n_ = 0;
    std::atomic<size_t> next_{0};
    unsigned generation_ = 0, slots_ = 0, active_ = 0;
    bool stop_ = false;
};

// The image of a whole world, kept up to date commit by commit.
class WorldImage {
public:
    // svo_type: VX_SVO_ESVO (1) or VX_SVO_CSVO (2), the format of the worlds handed to update()
    // first_word: where the arena starts in the frame (tests start a wide image beyond 4 GiB to exercise its 64-bit addresses)
    explicit WorldImage(int svo_type = 2, Layout layout = kOct64, uint64_t first_word = 0)
        : esvo_(svo_type == 1), layout_(layout), first_word_(first_word) {}

    // host mirror of the image frame, as 32-bit words
    const ZeroedWords& frame() const { return frame_; }
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
    // CSVO worlds in the kOct64 layouts: every voxel-parent octant is preceded by a unit that says where that leaf-mask byte is in the world's own bytes
    // (Octant::origin); the renderer reads it when a ray is led into a voxel (vx_device.hpp, walk_voxel_on_bytes). (Rounds 3-5 kept a table of its own beside
    // the image, a quarter of its size; origin() is what is left of that interface: empty.)
    const ZeroedWords& origin() const { return origin_; }
    uint64_t origin_bytes() const { return 0; }
    bool has_origin() const { return !esvo_ && layout_ != kEsvo48; }
    size_t chunk_count() const { return chunks_.size(); }
    // levels of the imaged octree (the world's depth); no path of the image is longer
    uint32_t depth() const { return depth_; }
    Layout layout() const { return layout_; }
    // the last update() failed only because the image outgrew what this layout's pointers can reach
    bool too_big() const { return too_big_; }
    // seconds the last update() spent: [0] walking the root, [1] walking the chunks (workers), [2] placing them, [3] encoding them (workers), [4] the root and the header
    const double* last_timing() const { return timing_; }

    // Is the point (x, y, z), in the octree's [1, 2)^3 coordinates, inside a voxel of the imaged world? A walk down the host
    // mirror, one child per level by the coordinates' mantissa bits. (The renderer asks this about the eye: every primary ray of
    // an eye inside a voxel is an inside-voxel ray, which the image cannot serve.)
    bool point_in_voxel(float x, float y, float z) const {
        if (layout_ == kEsvo48 || frame_.size() < 16 || !(x >= 1.0f && x < 2.0f && y >= 1.0f && y < 2.0f && z >= 1.0f && z < 2.0f)) return false;
        uint32_t bx, by, bz;
        std::memcpy(&bx, &x, 4);
        std::memcpy(&by, &y, 4);
        std::memcpy(&bz, &z, 4);
        uint32_t masks = frame_[1];
        uint64_t lo = frame_[2];  // the node's octant: the unit before its first entry
        for (int scale = 22; scale >= 0; --scale) {
            const uint32_t c = ((bx >> scale) & 1u) | (((by >> scale) & 1u) << 1) | (((bz >> scale) & 1u) << 2);
            const uint32_t m = masks << c;
            if (!(m & 0x00800000u)) return false;
            if (m & 0x80000000u) return true;
            const uint64_t entry = (lo + uint64_t(__builtin_popcount(m & 0x00ffffffu))) * 2;  // frame word of child c's entry
            if (entry + 1 >= frame_.size()) return false;
            masks = frame_[entry + 1];
            lo = frame_[entry];
        }
        return false;
    }

    // `world` = the frame as committed: [f32 scale][CSVO: u32 root_ptr | ESVO: 5-word preamble][arena]; `used` = bytes of the
    // arena in use; `changed` = byte ranges (relative to the arena, like vx_commit's) rewritten since the last call.
    // Returns false when the world cannot be imaged (malformed or image beyond 4 GiB): the caller then traverses the world's bytes.
    bool update(const uint8_t* world, uint64_t used, const Range* changed, size_t n_changed, unsigned threads) {
        // Nothing is thrown across the C ABI: running out of host memory in the middle of a build (a walker's tree, the frame's growth; on a
        // worker thread: rethrown by Workers::run on this one) leaves no image, like any other failure, and the context renders from the world's bytes.
        try {
            return update_or_throw(world, used, changed, n_changed, threads);
        } catch (const std::exception&) {
            return fail();
        }
    }

private:
    bool update_or_throw(const uint8_t* world, uint64_t used, const Range* changed, size_t n_changed, unsigned threads) {
        dirty_.clear();
        too_big_ = false;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(now() - t).count(); };
        auto t_step = now();
        if (used < 2) return false;
        const Bytes b{world + 8, size_t(used)};                                                        // CSVO arena
        const Words w{reinterpret_cast<const uint32_t*>(world + 4), size_t(used / 4 + 5)};             // ESVO descriptors[]
        uint32_t scale_bits;
        std::memcpy(&scale_bits, world, 4);
        const uint32_t depth = 127u - ((scale_bits >> 23) & 0xffu);  // svo.csvo.glsl:254
        if (depth < 1 || depth > 23) return false;
        if (frame_.empty()) {
            frame_.assign(header_words(), 0u);
            alloc_.reset(std::max(header_words(), first_word_ / 16 * 16));
        }

        // 1. walk the root octree: which chunks does it reference
        Tree root;
        std::vector<ChunkRef> refs;
        root.octants.reserve(last_root_octants_ + 64);  // (every commit rebuilds the root: no growing in steps)
        refs.reserve(chunks_.size() + 64);
        if (esvo_) {
            // the preamble is an octant whose child 0 is the root (esvo.rs:179-188, svo.esvo.glsl:139-141)
            const uint32_t p = w.at(4);
            EsvoWalker(w, root, &refs).run((p & 0x80000000u) ? 4u + (p & 0x7fffffffu) : p, w.at(0) & 0xffffu, depth);
        } else {
            uint32_t root_ptr;
            std::memcpy(&root_ptr, world + 4, 4);
            root.root = walk_root(b, root_ptr, depth, root, refs);
        }
        if (root.too_deep) return fail();  // (what follows relies on the image having at most `depth` levels, like the world says)
        depth_ = depth;
        last_root_octants_ = root.octants.size();

        timing_[0] = since(t_step);
        t_step = now();
        // 2. drop images of chunks whose bytes were rewritten, note which of the others are still referenced (and whether they
        //    still hang in the same kind of slot), drop the rest; what is referenced and has no image is to be walked
        for (auto it = chunks_.begin(); n_changed && it != chunks_.end();) {
            bool rewritten = false;
            for (size_t i = 0; i < n_changed && !rewritten; ++i)
                rewritten = changed[i].start < it->second.src_end && it->second.src_begin < changed[i].start + changed[i].length;
            if (rewritten) {
                alloc_.release(it->second.at, it->second.words);
                it = chunks_.erase(it);
            } else {
                ++it;
            }
        }
        ++epoch_;
        std::vector<ChunkRef> todo;
        for (const ChunkRef& r : refs) {
            const auto it = chunks_.find(r.key);
            if (it == chunks_.end()) {
                todo.push_back(r);
            } else if (esvo_ && (it->second.masks != r.masks || it->second.levels != r.levels)) {
                if (it->second.seen == epoch_) return fail();  // one chunk, two different slots
                alloc_.release(it->second.at, it->second.words);
                chunks_.erase(it);
                todo.push_back(r);
            } else {
                it->second.seen = epoch_;
            }
        }
        for (auto it = chunks_.begin(); it != chunks_.end();) {
            if (it->second.seen != epoch_) {
                alloc_.release(it->second.at, it->second.words);
                it = chunks_.erase(it);
            } else {
                ++it;
            }
        }

        // 3. walk what is missing (worker threads), place it (this thread), encode it in place (worker threads)
        std::sort(todo.begin(), todo.end(), [](const ChunkRef& x, const ChunkRef& y) { return x.key < y.key; });
        for (size_t i = 1; i < todo.size(); ++i)
            if (todo[i].key == todo[i - 1].key && (todo[i].masks != todo[i - 1].masks || todo[i].levels != todo[i - 1].levels)) return fail();
        todo.erase(std::unique(todo.begin(), todo.end(), [](const ChunkRef& x, const ChunkRef& y) { return x.key == y.key; }), todo.end());
        // In batches: a tree is ~60 bytes an octant until it is encoded -- all of a whole world's at once were 9 GB of first-touched memory for the
        // depth-14 terrain, and the workers' heap growth serialised them (the first build of a process walked its chunks no faster on sixteen
        // threads than on one). A batch's trees are walked into vectors that keep their capacity, placed, encoded in place, and reused.
        const size_t batch = 4096;
        // (workers: a sixteenth of the chunks at most -- an incremental commit stays on a few --; the encoding, whose first touch of the frame's
        // pages contends in the kernel, on sixteen at most: measured on the depth-14 terrain, profiles/round4/pass_p)
        Workers workers(std::max(1u, std::min<unsigned>(threads, unsigned(todo.size() / 16 + 1))));
        const unsigned encoders = std::min(16u, workers.size());
        const bool direct = layout_ != kEsvo48;  // (the renderer's layouts: a chunk's words straight from the walk)
        const size_t width = std::min(todo.size(), batch);
        std::vector<Tree> built(direct ? 0 : width);
        std::vector<Emitted> emitted(direct ? width : 0);
        // (two sets: a batch's book-keeping -- its entries in chunks_, its dirty range -- is one more task of the NEXT batch's walk, done by whichever
        // worker draws it while the others walk: 400,000 map insertions were a tenth of a second of everybody waiting for this thread)
        std::vector<Placed> placed_sets[2] = {std::vector<Placed>(width), std::vector<Placed>(width)};
        chunks_.reserve(chunks_.size() + todo.size());
        // a whole world: the frame's address space up front (about 2.3 x a CSVO world's bytes, 0.37 x an ESVO world's), so that it is not moved as it grows
        if (direct && todo.size() >= batch) frame_.reserve(frame_.size() + size_t(double(used) * (esvo_ ? 0.45 : 2.6) / 4.0));
        uint64_t top = frame_.size();
        double t_walk = since(t_step), t_place = 0.0, t_encode = 0.0;
        size_t kept_base = 0, kept_n = 0;  // the batch whose book-keeping is still to be done
        auto keep_books = [&](size_t from, size_t count, const std::vector<Placed>& of) {
            for (size_t i = 0; i < count; ++i) {
                const Placed& pl = of[i];
                if (!dirty_.empty() && dirty_.back().start + dirty_.back().length == pl.at * 4) dirty_.back().length += pl.words * 4;
                else dirty_.push_back(Range{pl.at * 4, pl.words * 4});
                chunks_[todo[from + i].key] = pl;
            }
        };
        for (size_t base = 0, round = 0; base < todo.size(); base += batch, ++round) {
            const size_t n = std::min(batch, todo.size() - base);
            std::vector<Placed>& placed = placed_sets[round & 1];
            const std::vector<Placed>& placed_before = placed_sets[(round & 1) ^ 1];
            t_step = now();
            workers.run(n + 1, workers.size(), [&](size_t task) {
                if (task == 0) {
                    keep_books(kept_base, kept_n, placed_before);
                    return;
                }
                const size_t i = task - 1;
                const ChunkRef& r = todo[base + i];
                if (direct) {
                    if (esvo_) EsvoEmitter(w, emitted[i]).run(r.key, r.masks, r.levels);
                    else ChunkEmitter(b, emitted[i]).run(r.key);
                    placed[i].words = emitted[i].n_words;
                    return;
                }
                built[i].too_deep = false;
                if (esvo_) EsvoWalker(w, built[i], nullptr).run(r.key, r.masks, r.levels);
                else ChunkWalker(b, built[i]).run(r.key);
                placed[i].words = tree_words(built[i]);  // (here, not in the placing loop: it looks at every octant)
            });
            kept_n = 0;
            t_walk += since(t_step);
            t_step = now();
            for (size_t i = 0; i < n; ++i) {
                if (direct ? emitted[i].too_deep : built[i].too_deep) return fail();
                Placed& pl = placed[i];
                pl.at = alloc_.alloc(pl.words);
                if (direct) {
                    const Emitted& e = emitted[i];
                    // (oct64_lo of the chunk's root octant: the unit before its first entry -- a root of values with an origin in front of them would be a CSVO
                    // chunk of one level, which ChunkEmitter refuses)
                    pl.root_lo = uint32_t(pl.at / 2) - 1u;
                    pl.masks = e.root.packed();
                    pl.src_begin = e.src_begin;
                    pl.src_end = e.src_end;
                } else {
                    pl.root_lo = 0u;
                    pl.masks = built[i].root.packed();
                    pl.src_begin = built[i].src_begin;
                    pl.src_end = built[i].src_end;
                }
                pl.levels = todo[base + i].levels;
                pl.seen = epoch_;
                top = std::max(top, pl.at + pl.words);
            }
            kept_base = base;
            kept_n = n;
            if (frame_.size() < top) frame_.resize(top, 0u);
            t_place += since(t_step);
            t_step = now();
            // (tasks are drawn in order: numbered across `encoders` stretches of the batch, the workers start in as many different places of the frame --
            // neighbours in one huge page wait for whoever touched it first to have it zeroed)
            const size_t rows = (n + encoders - 1) / encoders;
            workers.run(rows * encoders, encoders, [&](size_t task) {
                const size_t i = (task % encoders) * rows + task / encoders;
                if (i >= n) return;
                if (!direct) {
                    encode(built[i], placed[i].at);
                    return;
                }
                // the chunk's words where it was placed; what points into the chunk, from there
                const Emitted& e = emitted[i];
                uint32_t* dst = frame_.data() + placed[i].at;
                if (e.n_words) std::memcpy(dst, e.words.data(), e.n_words * 4);
                const uint32_t unit = uint32_t(placed[i].at / 2);
                for (size_t k = 0; k < e.n_relocs; ++k) dst[e.relocs[k]] += unit;
            });
            t_encode += since(t_step);
        }
        t_step = now();
        keep_books(kept_base, kept_n, placed_sets[((todo.size() + batch - 1) / batch + 1) & 1]);
        t_place += since(t_step);
        timing_[1] = t_walk;
        timing_[2] = t_place;
        timing_[3] = t_encode;
        t_step = now();

        // 4. the root octree is rewritten by every commit (csvo.rs:68-139 re-serializes it): so is its image
        alloc_.release(root_at_, root_words_);
        root_words_ = tree_words(root);
        root_at_ = alloc_.alloc(root_words_);
        if (frame_.size() < root_at_ + root_words_) frame_.resize(root_at_ + root_words_, 0u);
        for (Octant& o : root.octants)
            for (uint32_t c = 0; c < 8; ++c)
                if ((o.chunk_mask >> c) & 1u) {
                    const Placed& pl = chunks_.at(o.lo[c]);
                    o.lo[c] = uint32_t(layout_ == kEsvo48 ? pl.at : pl.root_lo);  // the chunk's root octant: frame word / its `lo`
                    o.masks[c] = uint16_t(pl.masks);
                }
        encode(root, root_at_);
        dirty_.push_back(Range{root_at_ * 4, root_words_ * 4});

        // 5. header
        frame_[0] = scale_bits;
        if (layout_ != kEsvo48) {
            frame_[1] = oct64_masks(root.root.packed());
            frame_[2] = root.octants.empty() ? 0u : oct64_lo(root.octants[0], root_at_, has_origin());
        } else {
            frame_[1] = root.root.packed();  // preamble: a fake octant whose child 0 is the root (esvo.rs:179-188)
            frame_[2] = frame_[3] = frame_[4] = 0;
            frame_[5] = uint32_t(root_at_ - 1);  // descriptors[] index = frame word index - 1
        }
        dirty_.push_back(Range{0, header_words() * 4});
        // what the pointers can reach: 32-bit byte offsets (and the buffer resource's) / 32-bit octant indices / 31-bit word offsets
        const uint64_t end = alloc_.end();
        timing_[4] = since(t_step);
        too_big_ = layout_ == kOct64 ? end * 4 + 4096 >= (uint64_t(1) << 32) : (layout_ == kOct64Wide ? end / 2 + 16 >= (uint64_t(1) << 32) : end >= (uint64_t(1) << 31));
        return !too_big_;
    }

private:
    struct Placed {
        uint64_t at = 0, words = 0;             // in frame words
        uint64_t src_begin = 0, src_end = 0;    // arena bytes it was read from
        uint32_t masks = 0, levels = 0;
        uint32_t root_lo = 0;  // kOct64*: what an entry that points to the chunk's root octant holds
        uint32_t seen = 0;  // the update() that last found it referenced
    };

    // nothing of a half-made update may survive: the next one starts from an empty image
    bool fail() {
        chunks_.clear();
        alloc_.reset(0);
        frame_.clear();
        origin_.clear();
        dirty_.clear();
        root_at_ = root_words_ = 0;
        return false;
    }

    uint64_t header_words() const { return layout_ == kEsvo48 ? 6 : 16; }
    uint64_t tree_words(const Tree& t) const {
        if (layout_ == kEsvo48) return t.octants.size() * 12;
        uint64_t n = 0;
        for (const Octant& o : t.octants) n += oct64_words(o, has_origin());
        return n;
    }

    // writes the tree's octants at frame word `at` (in walk order, each as large as its layout makes it); chunk children hold
    // the frame word index (kEsvo48) / the `lo` (kOct64*) of the chunk's root octant by now
    void encode(const Tree& t, uint64_t at) {
        uint32_t* dst = frame_.data() + at;
        std::vector<uint64_t> where;  // kOct64*: word offset of octant i inside the tree
        const bool with_origin = has_origin();
        if (layout_ != kEsvo48) {
            where.resize(t.octants.size());
            uint64_t n = 0;
            for (size_t i = 0; i < t.octants.size(); ++i) {
                where[i] = n;
                n += oct64_words(t.octants[i], with_origin);
            }
        }
        for (size_t i = 0; i < t.octants.size(); ++i) {
            const Octant& o = t.octants[i];
            if (layout_ != kEsvo48) {
                uint32_t* w = dst + where[i];
                const uint32_t words = oct64_words(o, with_origin);
                if (words == 0) continue;
                if (!(o.node_mask | o.chunk_mask)) {  // values only: [origin][the existing children's values, child 7 first][padding]
                    if (with_origin) {
                        w[0] = o.origin[0];
                        w[1] = o.origin[1];
                        w += 2;
                    }
                    uint32_t k = 0;
                    for (int c = 7; c >= 0; --c)
                        if ((o.leaf_mask >> c) & 1u) w[k++] = o.lo[c];
                    if (k & 1u) w[k] = 0u;
                    continue;
                }
                uint32_t k = 0;
                for (int c = 7; c >= 0; --c) {  // an entry per existing child, child 7 first
                    const uint32_t bit = 1u << c;
                    uint32_t lo = 0, hi = 0;
                    if (o.node_mask & bit) {
                        // (an empty child octant takes no room: wherever the next octant starts -- nothing ever reads it)
                        lo = oct64_lo(t.octants[o.lo[c]], at + where[o.lo[c]], with_origin);
                        hi = oct64_masks(o.masks[c]);
                    } else if (o.chunk_mask & bit) { lo = o.lo[c]; hi = oct64_masks(o.masks[c]); }
                    else if (o.leaf_mask & bit) { lo = o.lo[c]; }
                    else continue;
                    w[2 * k] = lo;
                    w[2 * k + 1] = hi;
                    ++k;
                }
            } else {
                uint32_t* w = dst + i * 12;
                for (int k = 0; k < 12; ++k) w[k] = 0;
                for (uint32_t c = 0; c < 8; ++c) {
                    const uint32_t bit = 1u << c;
                    if (o.node_mask & bit) {
                        const uint64_t target = at + uint64_t(o.lo[c]) * 12, body = at + i * 12 + 4 + c;
                        w[4 + c] = 0x80000000u | uint32_t(target - body);  // relative to this body word (esvo.rs:501-504)
                        w[c >> 1] |= uint32_t(o.masks[c]) << ((c & 1u) * 16);
                    } else if (o.chunk_mask & bit) {
                        w[4 + c] = o.lo[c] - 1u;  // absolute descriptors[] index of the chunk's root octant (esvo.rs:164-171)
                        w[c >> 1] |= uint32_t(o.masks[c]) << ((c & 1u) * 16);
                    } else if (o.leaf_mask & bit) {
                        w[4 + c] = o.lo[c];
                    }
                }
            }
        }
    }

    bool esvo_;
    Layout layout_;
    uint64_t first_word_;
    ZeroedWords frame_;
    ZeroedWords origin_;
    std::vector<Range> dirty_;
    WordAllocator alloc_;
    std::unordered_map<uint32_t, Placed> chunks_;
    uint64_t root_at_ = 0, root_words_ = 0;
    uint32_t depth_ = 0;
    size_t last_root_octants_ = 0;
    uint32_t epoch_ = 0;
    bool too_big_ = false;
    double timing_[5] = {};
};

// Do two kOct64 images ([64-byte header][octants]) hold the same tree -- same masks, same leaf values, same shape -- wherever their octants
// were placed? 1 / 0; -1 = a pointer out of range. (Tests: an image kept up to date commit by commit against one built from scratch.)
// with_origin: images of CSVO worlds (a unit in front of every octant of values).
inline int oct64_same_tree(const uint32_t* a, uint64_t na, const uint32_t* b, uint64_t nb, bool with_origin = false) {
    if (na < 16 || nb < 16 || a[0] != b[0] || a[1] != b[1]) return 0;
    struct Pair { uint32_t la, lb, masks; };
    std::vector<Pair> todo{{a[2], b[2], a[1]}};
    while (!todo.empty()) {
        const Pair p = todo.back();
        todo.pop_back();
        const uint32_t children = (p.masks >> 16) & 0xffu, leaves = p.masks >> 24;
        if (!children) continue;  // an octant without children takes no room
        const uint32_t n = uint32_t(__builtin_popcount(children));
        const bool values_only = children == leaves;
        const uint64_t wa = (uint64_t(p.la) + 1) * 2, wb = (uint64_t(p.lb) + 1) * 2;  // frame word of the first entry / value
        const uint64_t words = values_only ? n : 2u * n;
        if (wa + words > na || wb + words > nb) return -1;
        if (values_only) {
            if (with_origin && (wa < 2 || wb < 2)) return -1;
            for (uint32_t k = 0; k < n; ++k)
                if (a[wa + k] != b[wb + k]) return 0;
            continue;  // (the origins say where the bytes lie in worlds that may be laid out differently: not compared)
        }
        uint32_t k = 0;
        for (int c = 7; c >= 0; --c) {  // (bit 7 - c of `children` = child c: reversed masks)
            if (!((children >> (7 - c)) & 1u)) continue;
            const bool leaf = (leaves >> (7 - c)) & 1u;
            const uint32_t *ea = a + wa + 2 * k, *eb = b + wb + 2 * k;
            ++k;
            if (leaf) {
                if (ea[0] != eb[0]) return 0;
            } else {
                if (ea[1] != eb[1]) return 0;
                todo.push_back(Pair{ea[0], eb[0], ea[1]});
            }
        }
    }
    return 1;
}

}  // namespace vximg
