// Known-answer tests for the host-side mirror (octree, RangeBuffer, ESVO/CSVO serializers, coordinate types).
// Each case restates the expected values of one of the reference's own #[test]s; the case name carries the
// reference location. Run by tests/test_host_kats.py (one pytest case per KAT).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "camera.hpp"
#include "chunk.hpp"
#include "chunkloader.hpp"
#include "csvo.hpp"
#include "esvo.hpp"
#include "octree.hpp"
#include "physics.hpp"
#include "range_buffer.hpp"
#include "worldsvo.hpp"

using namespace vx;

static int g_failures = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::printf("  CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond);    \
            ++g_failures;                                                            \
        }                                                                            \
    } while (0)

template <class A, class B>
static void check_seq(const A& a, const B& b, const char* what, int line) {
    bool ok = a.size() == b.size();
    size_t bad = 0;
    if (ok)
        for (size_t i = 0; i < a.size(); ++i)
            if (!(uint64_t(a[i]) == uint64_t(b[i]))) { ok = false; bad = i; break; }
    if (!ok) {
        std::printf("  sequence mismatch (%s) line %d: sizes %zu vs %zu, first diff at %zu\n", what, line, a.size(), b.size(), bad);
        ++g_failures;
    }
}
#define CHECK_SEQ(a, b) check_seq(a, b, #a " == " #b, __LINE__)

static std::vector<uint8_t> u32_bytes(const std::vector<uint32_t>& w) {
    std::vector<uint8_t> out(w.size() * 4);
    std::memcpy(out.data(), w.data(), out.size());
    return out;
}
static std::vector<uint32_t> cat(std::initializer_list<std::vector<uint32_t>> parts) {
    std::vector<uint32_t> out;
    for (auto& p : parts) out.insert(out.end(), p.begin(), p.end());
    return out;
}

constexpr uint32_t REL = 1u << 31;

// ---------------------------------------------------------------------------------------------------------
// octree  (src/world/hds/octree.rs tests)
// ---------------------------------------------------------------------------------------------------------

// Whole-structure dump so the reference's `assert_eq!(octree, Octree {...})` checks carry over verbatim:
// "root depth free | id:parent count [children]" with children N (none), O<id>, L<value>.
static std::string dump(const Octree<uint32_t>& o) {
    std::ostringstream s;
    s << "root=" << (o.root ? std::to_string(*o.root) : "-") << " depth=" << int(o.depth()) << " free=[";
    for (size_t i = 0; i < o.free_list.size(); ++i) s << (i ? "," : "") << o.free_list[i];
    s << "]";
    for (size_t i = 0; i < o.octants.size(); ++i) {
        const auto& oc = o.octants[i];
        s << " | " << i << ":p" << (oc.parent ? std::to_string(*oc.parent) : "-") << " c" << int(oc.children_count) << " [";
        for (int c = 0; c < 8; ++c) {
            const auto& ch = oc.children[c];
            if (c) s << ",";
            if (ch.is_none()) s << "N";
            else if (ch.is_octant()) s << "O" << ch.octant;
            else s << "L" << *ch.leaf;
        }
        s << "]";
    }
    return s.str();
}
#define CHECK_DUMP(o, expected)                                                                         \
    do {                                                                                                \
        const std::string got = dump(o);                                                                \
        if (got != (expected)) {                                                                        \
            std::printf("  dump mismatch line %d:\n    got      %s\n    expected %s\n", __LINE__, got.c_str(), expected); \
            ++g_failures;                                                                               \
        }                                                                                               \
    } while (0)

static void octree_add_leaf_single() {  // octree.rs:515-545
    Octree<uint32_t> o;
    auto r = o.set_leaf(Position{1, 1, 3}, 20);
    CHECK(r.first.parent == 2 && r.first.idx == 7 && !r.second);
    CHECK_DUMP(o, "root=1 depth=2 free=[] | 0:p1 c0 [N,N,N,N,N,N,N,N] | 1:p- c2 [O0,N,N,N,O2,N,N,N] | 2:p1 c1 [N,N,N,N,N,N,N,L20]");
    CHECK(o.get_leaf(Position{1, 1, 3}) && *o.get_leaf(Position{1, 1, 3}) == 20);
    CHECK(o.get_leaf(Position{1, 1, 1}) == nullptr);
}

static void octree_add_leaf_multiple() {  // octree.rs:548-612
    Octree<uint32_t> o;
    auto a = o.set_leaf(Position{6, 7, 5}, 10);
    CHECK(a.first.parent == 4 && a.first.idx == 6 && !a.second);
    auto b = o.set_leaf(Position{0, 0, 0}, 20);
    CHECK(b.first.parent == 0 && b.first.idx == 0 && !b.second);
    auto c = o.set_leaf(Position{1, 0, 6}, 30);
    CHECK(c.first.parent == 6 && c.first.idx == 1 && !c.second);
    CHECK_DUMP(o, "root=2 depth=3 free=[]"
                  " | 0:p1 c1 [L20,N,N,N,N,N,N,N]"
                  " | 1:p2 c1 [O0,N,N,N,N,N,N,N]"
                  " | 2:p- c3 [O1,N,N,N,O5,N,N,O3]"
                  " | 3:p2 c1 [N,N,N,O4,N,N,N,N]"
                  " | 4:p3 c1 [N,N,N,N,N,N,L10,N]"
                  " | 5:p2 c1 [N,N,N,N,O6,N,N,N]"
                  " | 6:p5 c1 [N,L30,N,N,N,N,N,N]");
    CHECK(*o.get_leaf(Position{6, 7, 5}) == 10 && *o.get_leaf(Position{0, 0, 0}) == 20 && *o.get_leaf(Position{1, 0, 6}) == 30);
    CHECK(o.get_leaf(Position{1, 1, 1}) == nullptr);
    auto d = o.set_leaf(Position{0, 0, 0}, 40);  // replace by adding
    CHECK(d.first.parent == 0 && d.first.idx == 0 && d.second && *d.second == 20);
    CHECK(*o.get_leaf(Position{0, 0, 0}) == 40);
}

static void octree_remove_and_add_leaf() {  // octree.rs:616-686
    Octree<uint32_t> o;
    o.set_leaf(Position{0, 0, 0}, 10);
    o.set_leaf(Position{1, 0, 0}, 20);
    CHECK_DUMP(o, "root=0 depth=1 free=[] | 0:p- c2 [L10,L20,N,N,N,N,N,N]");
    auto rm = o.remove_leaf(Position{0, 0, 0});
    CHECK(rm.first && *rm.first == 10 && rm.second && rm.second->parent == 0 && rm.second->idx == 0);
    auto rm2 = o.remove_leaf_by_id(LeafId{0, 1});
    CHECK(rm2 && *rm2 == 20);
    CHECK_DUMP(o, "root=0 depth=1 free=[] | 0:p- c0 [N,N,N,N,N,N,N,N]");
    o.set_leaf(Position{0, 0, 0}, 30);
    CHECK_DUMP(o, "root=0 depth=1 free=[] | 0:p- c1 [L30,N,N,N,N,N,N,N]");
    CHECK(!o.remove_leaf(Position{100, 0, 0}).first);  // outside current depth
}

static void octree_move_leaf() {  // octree.rs:690-772
    Octree<uint32_t> o;
    o.set_leaf(Position{0, 0, 0}, 10);
    o.set_leaf(Position{1, 1, 1}, 20);
    auto m = o.move_leaf(LeafId{0, 0}, Position{1, 0, 0});  // into an empty slot
    CHECK(m.first.parent == 0 && m.first.idx == 1 && !m.second);
    CHECK_DUMP(o, "root=0 depth=1 free=[] | 0:p- c2 [N,L10,N,N,N,N,N,L20]");
    m = o.move_leaf(LeafId{0, 1}, Position{1, 0, 0});  // onto itself
    CHECK(m.first.parent == 0 && m.first.idx == 1 && !m.second);
    CHECK_DUMP(o, "root=0 depth=1 free=[] | 0:p- c2 [N,L10,N,N,N,N,N,L20]");
    m = o.move_leaf(LeafId{0, 1}, Position{1, 1, 1});  // onto an existing leaf
    CHECK(m.first.parent == 0 && m.first.idx == 7 && m.second && *m.second == 20);
    CHECK_DUMP(o, "root=0 depth=1 free=[] | 0:p- c1 [N,N,N,N,N,N,N,L10]");
    m = o.move_leaf(LeafId{0, 7}, Position{2, 0, 0});  // into a new parent (tree grows)
    CHECK(m.first.parent == 2 && m.first.idx == 0 && !m.second);
    CHECK_DUMP(o, "root=1 depth=2 free=[] | 0:p1 c0 [N,N,N,N,N,N,N,N] | 1:p- c2 [O0,O2,N,N,N,N,N,N] | 2:p1 c1 [L10,N,N,N,N,N,N,N]");
}

static void octree_construct_octants() {  // octree.rs:775-815
    Octree<uint32_t> o;
    o.set_leaf(Position{1, 1, 3}, 2);
    o.construct_octants_with(2, [](Position) -> std::optional<uint32_t> { return std::nullopt; });
    CHECK_DUMP(o, "root=- depth=0 free=[]");
    o.construct_octants_with(2, [](Position p) -> std::optional<uint32_t> {
        if (p.x == 2 && p.y == 2 && p.z == 2) return 1u;
        return std::nullopt;
    });
    CHECK_DUMP(o, "root=1 depth=2 free=[] | 0:p1 c1 [L1,N,N,N,N,N,N,N] | 1:p- c1 [N,N,N,N,N,N,N,O0]");
    CHECK(*o.get_leaf(Position{2, 2, 2}) == 1 && o.get_leaf(Position{1, 1, 1}) == nullptr);
}

static void octree_compact() {  // octree.rs:818-893
    Octree<uint32_t> o;
    o.set_leaf(Position{0, 1, 3}, 10);
    o.set_leaf(Position{1, 1, 3}, 20);
    CHECK_DUMP(o, "root=1 depth=2 free=[] | 0:p1 c0 [N,N,N,N,N,N,N,N] | 1:p- c2 [O0,N,N,N,O2,N,N,N] | 2:p1 c2 [N,N,N,N,N,N,L10,L20]");
    o.compact();
    CHECK_DUMP(o, "root=1 depth=2 free=[0] | 0:p- c0 [N,N,N,N,N,N,N,N] | 1:p- c1 [N,N,N,N,O2,N,N,N] | 2:p1 c2 [N,N,N,N,N,N,L10,L20]");
    o.remove_leaf(Position{0, 1, 3});
    o.remove_leaf(Position{1, 1, 3});
    o.compact();
    CHECK_DUMP(o, "root=- depth=0 free=[]");
}

static void octree_expand_wraps_root_as_child0() {  // octree.rs:311-336 and SURVEY §8c last paragraph
    Octree<uint32_t> o;
    o.set_leaf(Position{15, 15, 15}, 1);  // required depth 4
    CHECK(o.depth() == 4);
    // expand(4) on an empty tree leaves a chain root -> child0 -> child0 -> child0 of empty octants
    CHECK(o.octants.size() == 4 + 3);
    const auto& root = o.octants[*o.root];
    CHECK(root.children[0].is_octant() && root.children[7].is_octant());
    o.compact();
    CHECK(!o.octants[*o.root].children[0].is_octant());
    CHECK(o.get_leaf(Position{15, 15, 15}) && *o.get_leaf(Position{15, 15, 15}) == 1);
}

static void position_required_depth() {  // octree.rs:25-28
    CHECK((Position{0, 0, 0}.required_depth() == 1));
    CHECK((Position{1, 0, 0}.required_depth() == 1));
    CHECK((Position{2, 0, 0}.required_depth() == 2));
    CHECK((Position{0, 3, 0}.required_depth() == 2));
    CHECK((Position{0, 0, 4}.required_depth() == 3));
    CHECK((Position{15, 15, 15}.required_depth() == 4));
    CHECK((Position{31, 0, 0}.required_depth() == 5));
    CHECK((Position{32, 0, 0}.required_depth() == 6));
}

// ---------------------------------------------------------------------------------------------------------
// RangeBuffer  (src/world/hds/internal.rs:289-453)
// ---------------------------------------------------------------------------------------------------------

static std::vector<uint8_t> B(std::initializer_list<int> v) { return std::vector<uint8_t>(v.begin(), v.end()); }

static void range_buffer_insert_remove() {  // internal.rs:289-385 (u32 elements there, bytes here)
    RangeBuffer buf(10);
    CHECK(buf.free_ranges.size() == 1 && (buf.free_ranges[0] == Range{0, 10}));
    auto ins = [&](uint64_t id, std::vector<uint8_t> v) { return buf.insert(id, v.data(), v.size()); };
    ins(1, B({0, 1, 2, 3, 4}));
    ins(2, B({5, 6}));
    ins(3, B({7, 8, 9}));
    CHECK_SEQ(buf.bytes, B({0, 1, 2, 3, 4, 5, 6, 7, 8, 9}));
    CHECK(buf.free_ranges.empty());
    CHECK(buf.updated_ranges.size() == 1 && (buf.updated_ranges[0] == Range{0, 10}));
    CHECK((buf.octant_to_range[2] == Range{5, 2}) && (buf.octant_to_range[3] == Range{7, 3}));

    ins(4, B({10}));  // exceeds initial capacity -> append
    CHECK_SEQ(buf.bytes, B({0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10}));
    CHECK((buf.updated_ranges[0] == Range{0, 11}));

    ins(3, B({11}));  // replace existing: old range freed first, first fit reuses its head
    CHECK_SEQ(buf.bytes, B({0, 1, 2, 3, 4, 5, 6, 11, 8, 9, 10}));
    CHECK(buf.free_ranges.size() == 1 && (buf.free_ranges[0] == Range{8, 2}));
    CHECK((buf.octant_to_range[3] == Range{7, 1}));

    buf.remove(2);
    buf.remove(3);
    CHECK(buf.free_ranges.size() == 1 && (buf.free_ranges[0] == Range{5, 5}));

    ins(5, B({12, 13, 14}));
    CHECK_SEQ(buf.bytes, B({0, 1, 2, 3, 4, 12, 13, 14, 8, 9, 10}));
    CHECK(buf.free_ranges.size() == 1 && (buf.free_ranges[0] == Range{8, 2}));
    CHECK((buf.octant_to_range[5] == Range{5, 3}));

    buf.remove(5);
    buf.remove(4);
    buf.remove(1);
    CHECK(buf.free_ranges.size() == 1 && (buf.free_ranges[0] == Range{0, 11}));
    CHECK(buf.octant_to_range.empty());
}

static void range_buffer_merge_ranges() {  // internal.rs:388-453
    struct Case { std::vector<Range> in, out; };
    std::vector<Case> cases = {
        {{{0, 1}, {1, 1}, {2, 1}}, {{0, 3}}},
        {{{0, 1}, {2, 1}}, {{0, 1}, {2, 1}}},
        {{{0, 5}, {3, 1}}, {{0, 5}}},
        {{{0, 5}, {3, 5}}, {{0, 8}}},
        {{{3, 5}, {0, 5}}, {{0, 8}}},
    };
    for (auto& c : cases) {
        RangeBuffer::merge_ranges(c.in);
        CHECK(c.in == c.out);
    }
}

// ---------------------------------------------------------------------------------------------------------
// ESVO  (src/world/hds/esvo.rs:562-1228)
// ---------------------------------------------------------------------------------------------------------

static Octree<BlockId> three_corner_octree(bool expand_to5) {
    Octree<BlockId> o;
    o.set_leaf(Position{31, 0, 0}, 1);
    o.set_leaf(Position{0, 31, 0}, 2);
    o.set_leaf(Position{0, 0, 31}, 3);
    if (expand_to5) o.expand_to(5);
    o.compact();
    return o;
}

// one arm of the three-corner chunk: `levels` inner octants whose only child is `idx`, then the leaf octant
static std::vector<uint32_t> esvo_arm(uint32_t idx, uint32_t value, int inner_levels) {
    std::vector<uint32_t> out;
    for (int l = 0; l < inner_levels; ++l) {
        std::vector<uint32_t> oct(12, 0);
        uint32_t mask = (1u << idx) << 8;
        if (l == inner_levels - 1) mask |= (1u << idx);  // the child below is the leaf octant: its leaf mask
        oct[idx / 2] = (idx & 1) ? mask << 16 : mask;
        oct[4 + idx] = REL | (12 - 4 - idx);
        out.insert(out.end(), oct.begin(), oct.end());
    }
    std::vector<uint32_t> leaf(12, 0);
    leaf[4 + idx] = value;
    out.insert(out.end(), leaf.begin(), leaf.end());
    return out;
}

static std::vector<uint32_t> esvo_three_corner_expected(int lod) {  // esvo.rs:873-1222
    if (lod == 1) return {0, 0, 0, 0, 0, 1, 2, 0, 3, 0, 0, 0};
    const int inner = lod - 2;                  // octants between core and leaf octant
    const uint32_t arm = uint32_t(inner + 1) * 12;  // words per arm
    std::vector<uint32_t> core(12, 0);
    auto mask_for = [&](uint32_t idx) {
        uint32_t m = (1u << idx) << 8;
        if (inner == 0) m |= (1u << idx);
        return (idx & 1) ? m << 16 : m;
    };
    core[0] = mask_for(1);
    core[1] = mask_for(2);
    core[2] = mask_for(4);
    core[4 + 1] = REL | (12 - 4 - 1);
    core[4 + 2] = REL | (12 + arm - 4 - 2);
    core[4 + 4] = REL | (12 + 2 * arm - 4 - 4);
    return cat({core, esvo_arm(1, 1, inner), esvo_arm(2, 2, inner), esvo_arm(4, 3, inner)});
}

static void esvo_serialize_with_lod() {  // esvo.rs:861-1228
    Octree<BlockId> o = three_corner_octree(true);
    for (int lod = 5; lod >= 1; --lod) {
        std::vector<uint32_t> buf;
        EsvoResult r = EsvoSerializedChunk::serialize_storage(o, buf, uint8_t(lod));
        CHECK_SEQ(buf, esvo_three_corner_expected(lod));
        CHECK(r.child_mask == (2 | 4 | 16));
        CHECK(r.leaf_mask == (lod == 1 ? (2 | 4 | 16) : 0));
        CHECK(r.depth == lod);
    }
    // spot values spelled out in the reference (LOD 5): body pointers 7, 6+4*12, 4+8*12
    std::vector<uint32_t> buf;
    EsvoSerializedChunk::serialize_storage(o, buf, 5);
    CHECK(buf[0] == ((2u << 8) << 16) && buf[1] == (4u << 8) && buf[2] == (16u << 8) && buf[3] == 0);
    CHECK(buf[5] == (REL | 7) && buf[6] == (REL | (6 + 4 * 12)) && buf[8] == (REL | (4 + 8 * 12)));
    CHECK(buf[12 + 24] == (((2u << 8) | 2) << 16));  // header 3 of the first arm
    CHECK(buf[12 + 36 + 5] == 1);                    // leaf body
}

static void esvo_serialize_world() {  // esvo.rs:562-742
    Octree<BlockId> o = three_corner_octree(true);
    EsvoSerializedChunk sc;
    sc.pos = ChunkPos{1, 0, 0};
    sc.lod = 0;
    sc.pos_hash = 100;
    std::vector<uint32_t> words;
    sc.result = EsvoSerializedChunk::serialize_storage(o, words, 0);
    sc.buffer = words;

    Esvo<EsvoSerializedChunk> esvo;
    esvo.set_leaf(Position{1, 0, 0}, std::move(sc), true);
    esvo.serialize();

    CHECK(esvo.root_info && esvo.root_info->buf_offset == 156);
    CHECK((esvo.root_info->serialization == EsvoResult{2, 0, 6}));

    std::vector<uint32_t> root(12, 0);
    root[0] = ((2u | 4u | 16u) << 8) << 16;
    root[4 + 1] = 5;  // absolute: chunk offset 0 + preamble
    std::vector<uint32_t> expected = cat({esvo_three_corner_expected(5), root});
    CHECK_SEQ(esvo.buffer.bytes, u32_bytes(expected));
    CHECK(esvo.buffer.bytes.size() == 672);
    CHECK(esvo.buffer.free_ranges.empty());
    CHECK(esvo.buffer.updated_ranges.size() == 1 && (esvo.buffer.updated_ranges[0] == Range{0, 672}));
    CHECK((esvo.buffer.octant_to_range[100] == Range{0, 624}));
    CHECK((esvo.buffer.octant_to_range[UINT64_MAX] == Range{624, 48}));

    std::vector<uint8_t> out(800, 0);
    const size_t n = esvo.write_to(out.data());
    out.resize(n);
    CHECK_SEQ(out, u32_bytes(cat({{2u << 8, 0, 0, 0, 156 + 5}, expected})));
}

static void esvo_serialize_with_remove_and_move() {  // esvo.rs:745-858
    Esvo<EsvoU32Leaf> esvo;
    esvo.set_leaf(Position{0, 0, 0}, EsvoU32Leaf{10}, true);
    esvo.serialize();
    esvo.set_leaf(Position{1, 0, 0}, EsvoU32Leaf{20}, true);
    esvo.serialize();

    CHECK(esvo.root_info && esvo.root_info->buf_offset == 1);
    CHECK((esvo.root_info->serialization == EsvoResult{3, 0, 2}));

    std::vector<uint32_t> expected = {10, (((1u << 8) | 1) << 16) | ((1u << 8) | 1), 0, 0, 0, 5, 18, 0, 0, 0, 0, 0, 0, 20};
    CHECK_SEQ(esvo.buffer.bytes, u32_bytes(expected));
    CHECK(esvo.buffer.free_ranges.empty());
    CHECK(esvo.buffer.updated_ranges.size() == 1 && (esvo.buffer.updated_ranges[0] == Range{0, 56}));
    CHECK((esvo.buffer.octant_to_range[10] == Range{0, 4}) && (esvo.buffer.octant_to_range[20] == Range{52, 4}) &&
          (esvo.buffer.octant_to_range[UINT64_MAX] == Range{4, 48}));
    esvo.buffer.updated_ranges.clear();

    std::vector<uint8_t> out(800, 0);
    const size_t size = esvo.write_to(out.data());
    CHECK_SEQ(std::vector<uint8_t>(out.begin(), out.begin() + size), u32_bytes(cat({{(2u | 1u) << 8, 0, 0, 0, 1 + 5}, expected})));

    auto mv = esvo.move_leaf(LeafId{0, 1}, Position{1, 1, 1});
    CHECK(mv.first.parent == 0 && mv.first.idx == 7 && !mv.second);
    auto old = esvo.remove_leaf(LeafId{0, 0});
    CHECK(old && old->value == 10);
    esvo.serialize();

    CHECK(esvo.root_info->buf_offset == 0);
    CHECK((esvo.root_info->serialization == EsvoResult{uint8_t(1u << 7), 0, 2}));
    std::vector<uint32_t> expected2 = {0, 0, 0, ((1u << 8) | 1) << 16, 0, 0, 0, 0, 0, 0, 0, 18, 0, 20};
    CHECK_SEQ(esvo.buffer.bytes, u32_bytes(expected2));
    CHECK(esvo.buffer.free_ranges.size() == 1 && (esvo.buffer.free_ranges[0] == Range{48, 4}));
    CHECK(esvo.buffer.updated_ranges.size() == 1 && (esvo.buffer.updated_ranges[0] == Range{0, 48}));

    CHECK(esvo.write_changes_to(out.data(), out.size(), true));
    CHECK_SEQ(std::vector<uint8_t>(out.begin(), out.begin() + size), u32_bytes(cat({{(1u << 7) << 8, 0, 0, 0, 5}, expected2})));
    CHECK(esvo.buffer.updated_ranges.empty());
}

// ---------------------------------------------------------------------------------------------------------
// CSVO  (src/world/hds/csvo.rs:329-388, 600-711)
// ---------------------------------------------------------------------------------------------------------

static void csvo_serialize_octant_single_leaf() {  // csvo.rs:600-616
    Octree<BlockId> o;
    o.set_leaf(Position{0, 0, 0}, 1);
    o.expand_to(4);
    o.compact();
    std::vector<BlockId> mats;
    auto data = CsvoSerializedChunk::serialize_octant(o, *o.root, o.depth(), 0, mats);
    CHECK_SEQ(data, B({1, 0, 0, 1, 0, 1, 0, 0, 1}));
    CHECK_SEQ(mats, std::vector<BlockId>({1}));
}

static void csvo_serialize_octant_multiple_leaves() {  // csvo.rs:618-638
    Octree<BlockId> o;
    o.set_leaf(Position{0, 0, 0}, 1);
    o.set_leaf(Position{3, 3, 3}, 2);
    o.set_leaf(Position{5, 4, 4}, 1);
    o.set_leaf(Position{6, 7, 7}, 2);
    o.expand_to(4);
    o.compact();
    std::vector<BlockId> mats;
    auto data = CsvoSerializedChunk::serialize_octant(o, *o.root, o.depth(), 0, mats);
    CHECK_SEQ(data, B({1, 0, 0, 1 | (1 << 7), 0, 5, 1 | (1 << 7), 0, 0, 1, 1 << 7, 1 | (1 << 7), 2, 0, 2, 1 << 6}));
    CHECK_SEQ(mats, std::vector<BlockId>({1, 2, 1, 2}));
}

static const std::vector<uint8_t> kCsvoChunkBytes = {
    0b00010100, 0b00000001, 0, 9, 18,
    0b00000100, 0, 0, 2, 0, 2, 0, 0, 2,
    0b00010000, 0, 0, 4, 0, 4, 1, 0, 4,
    0, 0b00000001, 0, 16, 0, 16, 2, 0, 16,
};

static void csvo_serialize_octant_chunk() {  // csvo.rs:640-664
    Octree<BlockId> o = three_corner_octree(false);
    std::vector<BlockId> mats;
    auto data = CsvoSerializedChunk::serialize_octant(o, *o.root, o.depth(), 0, mats);
    CHECK_SEQ(data, kCsvoChunkBytes);
    CHECK_SEQ(mats, std::vector<BlockId>({1, 2, 3}));
}

static void csvo_serialize_octant_chunk_with_lod() {  // csvo.rs:666-711
    Octree<BlockId> o = three_corner_octree(false);
    std::vector<BlockId> mats;
    auto d1 = CsvoSerializedChunk::serialize_octant(o, *o.root, uint8_t(o.depth() - 1), 0, mats);
    CHECK_SEQ(d1, B({0b00010100, 0b00000001, 0, 6, 12, 2, 0, 2, 0, 0, 2, 4, 0, 4, 1, 0, 4, 16, 0, 16, 2, 0, 16}));
    CHECK_SEQ(mats, std::vector<BlockId>({1, 2, 3}));
    mats.clear();
    auto d2 = CsvoSerializedChunk::serialize_octant(o, *o.root, uint8_t(o.depth() - 2), 0, mats);
    CHECK_SEQ(d2, B({0b00010110, 0, 4, 8, 2, 0, 0, 2, 4, 1, 0, 4, 16, 2, 0, 16}));
    mats.clear();
    auto d3 = CsvoSerializedChunk::serialize_octant(o, *o.root, uint8_t(o.depth() - 3), 0, mats);
    CHECK_SEQ(d3, B({0b00010110, 0, 0, 2, 4, 16}));
    mats.clear();
    auto d4 = CsvoSerializedChunk::serialize_octant(o, *o.root, uint8_t(o.depth() - 4), 0, mats);
    CHECK_SEQ(d4, B({22}));
    CHECK_SEQ(mats, std::vector<BlockId>({1, 2, 3}));
}

static void csvo_serialize_world() {  // csvo.rs:329-388
    Chunk chunk(ChunkPos{0, 0, 0}, 5);
    chunk.set_block(31, 0, 0, 1);
    chunk.set_block(0, 31, 0, 2);
    chunk.set_block(0, 0, 31, 3);
    chunk.storage.compact();
    CsvoSerializedChunk sc(chunk);
    CHECK(sc.pos_hash == 2435999049025295583ull);  // Rust DefaultHasher of ChunkPos(0,0,0), csvo.rs:376

    Csvo csvo;
    csvo.set_leaf(Position{1, 0, 0}, std::move(sc), true);
    csvo.serialize();
    CHECK(csvo.root_info && csvo.root_info->buf_offset == 49);
    CHECK(csvo.depth() == 6);

    std::vector<uint8_t> expected = {5, 12, 0, 0, 0, 1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0};
    expected.insert(expected.end(), kCsvoChunkBytes.begin(), kCsvoChunkBytes.end());
    const std::vector<uint8_t> root = {0b00001100, 0, 0, 0, 0, 1 << 7};
    expected.insert(expected.end(), root.begin(), root.end());
    CHECK_SEQ(csvo.buffer.bytes, expected);
    CHECK(csvo.buffer.free_ranges.empty());
    CHECK(csvo.buffer.updated_ranges.size() == 1 && (csvo.buffer.updated_ranges[0] == Range{0, 55}));
    CHECK((csvo.buffer.octant_to_range[2435999049025295583ull] == Range{0, 49}));
    CHECK((csvo.buffer.octant_to_range[UINT64_MAX] == Range{49, 6}));

    std::vector<uint8_t> out(200, 0);
    const size_t n = csvo.write_to(out.data());
    out.resize(n);
    std::vector<uint8_t> framed = {49, 0, 0, 0};
    framed.insert(framed.end(), expected.begin(), expected.end());
    CHECK_SEQ(out, framed);
}

// ---------------------------------------------------------------------------------------------------------
// chunk / coordinates  (src/world/chunk.rs:201-358)
// ---------------------------------------------------------------------------------------------------------

static void chunk_pos_from_block_pos() {  // chunk.rs:201-244
    CHECK((ChunkPos::from_block_pos(15, 28, 35) == ChunkPos{0, 0, 1}));
    CHECK((ChunkPos::from_block_pos(-15, -28, -35) == ChunkPos{-1, -1, -2}));
    CHECK((ChunkPos::from_block_pos(-32, 32, -33) == ChunkPos{-1, 1, -2}));
    CHECK(((ChunkPos{0, -1, 1} - ChunkPos{-1, 2, 0}) == ChunkPos{1, -3, 1}));
}

static void block_pos_roundtrip() {  // chunk.rs:301-358
    BlockPos p = BlockPos::from_ints(15, -28, 35);
    CHECK((p.chunk == ChunkPos{0, -1, 1}) && p.rel_x == 15.0f && p.rel_y == 4.0f && p.rel_z == 3.0f);
    float pt[3];
    p.to_point(pt);
    CHECK(pt[0] == 15.0f && pt[1] == -28.0f && pt[2] == 35.0f);

    BlockPos q = BlockPos::from_point(0.25f, 32.75f, 8.5f);
    CHECK((q.chunk == ChunkPos{0, 1, 0}) && q.rel_x == 0.25f && q.rel_y == 0.75f && q.rel_z == 8.5f);
    q.to_point(pt);
    CHECK(pt[0] == 0.25f && pt[1] == 32.75f && pt[2] == 8.5f);

    BlockPos n = BlockPos::from_point(-0.25f, -32.75f, -8.5f);
    CHECK((n.chunk == ChunkPos{-1, -2, -1}) && n.rel_x == 31.75f && n.rel_y == 31.25f && n.rel_z == 23.5f);
    n.to_point(pt);
    CHECK(pt[0] == -0.25f && pt[1] == -32.75f && pt[2] == -8.5f);

    BlockPos e = BlockPos::from_point(-1.0f, -1.0f, -1.0f);
    CHECK((e.chunk == ChunkPos{-1, -1, -1}) && e.rel_x == 31.0f && e.rel_y == 31.0f && e.rel_z == 31.0f);
    e.to_point(pt);
    CHECK(pt[0] == -1.0f && pt[1] == -1.0f && pt[2] == -1.0f);
}

static void chunk_storage_depth_and_blocks() {  // chunk.rs:60-90, 110-131
    Chunk c(ChunkPos{0, 0, 0}, 5);
    CHECK(c.storage.depth() == 5);
    CHECK(c.get_block(3, 4, 5) == NO_BLOCK);
    c.set_block(3, 4, 5, 7);
    CHECK(c.get_block(3, 4, 5) == 7);
    c.set_block(3, 4, 5, NO_BLOCK);
    CHECK(c.get_block(3, 4, 5) == NO_BLOCK);
    c.fill_with([](uint32_t x, uint32_t y, uint32_t z) -> std::optional<BlockId> {
        if ((x + y + z) % 2 == 0) return 1u;
        return std::nullopt;
    });
    CHECK(c.storage.depth() == 5 && c.get_block(0, 0, 0) == 1 && c.get_block(1, 0, 0) == NO_BLOCK && c.get_block(31, 31, 0) == 1);
}

// ---------------------------------------------------------------------------------------------------------
// world <-> SVO space, chunk shifting  (src/systems/worldsvo.rs:226-386, 505-558)
// ---------------------------------------------------------------------------------------------------------

using vx::systems::SvoCoordSpace;

static void coord_space_positive() {  // worldsvo.rs:513-524
    SvoCoordSpace cs{ChunkPos{4, 5, 12}, 2};
    const Vec3 world{std::fma(32.0f, 5.0f, 16.25f), std::fma(32.0f, 3.0f, 4.25f), std::fma(32.0f, 10.0f, 20.5f)};
    const Vec3 svo = cs.cnv_block_pos(world);
    CHECK((svo == Vec3{std::fma(32.0f, 3.0f, 16.25f), std::fma(32.0f, 0.0f, 4.25f), std::fma(32.0f, 0.0f, 20.5f)}));
    CHECK(cs.cnv_svo_pos(svo) == world);
}

static void coord_space_negative() {  // worldsvo.rs:526-537
    SvoCoordSpace cs{ChunkPos{-1, -1, -1}, 2};
    const Vec3 world{-16.25f, -4.25f, -20.5f};
    const Vec3 svo = cs.cnv_block_pos(world);
    CHECK((svo == Vec3{std::fma(32.0f, 2.0f, 15.75f), std::fma(32.0f, 2.0f, 27.75f), std::fma(32.0f, 2.0f, 11.5f)}));
    CHECK(cs.cnv_svo_pos(svo) == world);
}

static void coord_space_cnv_chunk_pos() {  // worldsvo.rs:539-557
    SvoCoordSpace cs{ChunkPos{0, 0, 0}, 1};
    auto eq = [](std::optional<Position> p, Position q) { return p && *p == q; };
    CHECK(eq(cs.cnv_chunk_pos(ChunkPos{-1, 0, 0}), Position{0, 1, 1}));
    CHECK(eq(cs.cnv_chunk_pos(ChunkPos{0, 0, 0}), Position{1, 1, 1}));
    CHECK(eq(cs.cnv_chunk_pos(ChunkPos{1, 0, 0}), Position{2, 1, 1}));
    CHECK(!cs.cnv_chunk_pos(ChunkPos{-2, 0, 0}));
    CHECK(!cs.cnv_chunk_pos(ChunkPos{2, 0, 0}));
    CHECK(!cs.cnv_chunk_pos(ChunkPos{1, 0, 1}));
}

struct ShiftFixture {
    std::unordered_map<ChunkPos, LeafId, ChunkPosHash> leaf_ids;
    Esvo<EsvoU32Leaf> world;
    LeafId c0, c1, c2;
    ShiftFixture() {
        c0 = world.set_leaf(Position{0, 1, 1}, EsvoU32Leaf{1}, true).first; leaf_ids[ChunkPos{-1, 0, 0}] = c0;
        c1 = world.set_leaf(Position{1, 1, 1}, EsvoU32Leaf{2}, true).first; leaf_ids[ChunkPos{0, 0, 0}] = c1;
        c2 = world.set_leaf(Position{2, 1, 1}, EsvoU32Leaf{3}, true).first; leaf_ids[ChunkPos{1, 0, 0}] = c2;
    }
    uint32_t at(uint32_t x) const { const EsvoU32Leaf* l = world.get_leaf(Position{x, 1, 1}); return l ? l->value : 0; }
    bool ids_are(std::initializer_list<std::pair<ChunkPos, LeafId>> want) const {
        if (leaf_ids.size() != want.size()) return false;
        for (auto& kv : want) {
            auto it = leaf_ids.find(kv.first);
            if (it == leaf_ids.end() || it->second != kv.second) return false;
        }
        return true;
    }
};

static void shift_chunks_x_positive() {  // worldsvo.rs:247-300
    ShiftFixture f;
    CHECK(f.at(0) == 1 && f.at(1) == 2 && f.at(2) == 3);
    vx::systems::shift_chunks(SvoCoordSpace{ChunkPos{1, 0, 0}, 1}, f.leaf_ids, f.world);
    CHECK(f.ids_are({{ChunkPos{0, 0, 0}, f.c0}, {ChunkPos{1, 0, 0}, f.c1}}));
    CHECK(f.at(0) == 2 && f.at(1) == 3 && f.at(2) == 0);
    vx::systems::shift_chunks(SvoCoordSpace{ChunkPos{2, 0, 0}, 1}, f.leaf_ids, f.world);
    CHECK(f.ids_are({{ChunkPos{1, 0, 0}, f.c0}}));
    CHECK(f.at(0) == 3 && f.at(1) == 0 && f.at(2) == 0);
    vx::systems::shift_chunks(SvoCoordSpace{ChunkPos{3, 0, 0}, 1}, f.leaf_ids, f.world);
    CHECK(f.leaf_ids.empty());
    CHECK(f.at(0) == 0 && f.at(1) == 0 && f.at(2) == 0);
}

static void shift_chunks_x_negative() {  // worldsvo.rs:302-355
    ShiftFixture f;
    vx::systems::shift_chunks(SvoCoordSpace{ChunkPos{-1, 0, 0}, 1}, f.leaf_ids, f.world);
    CHECK(f.ids_are({{ChunkPos{-1, 0, 0}, f.c1}, {ChunkPos{0, 0, 0}, f.c2}}));
    CHECK(f.at(0) == 0 && f.at(1) == 1 && f.at(2) == 2);
    vx::systems::shift_chunks(SvoCoordSpace{ChunkPos{-2, 0, 0}, 1}, f.leaf_ids, f.world);
    CHECK(f.ids_are({{ChunkPos{-1, 0, 0}, f.c2}}));
    CHECK(f.at(0) == 0 && f.at(1) == 0 && f.at(2) == 1);
    vx::systems::shift_chunks(SvoCoordSpace{ChunkPos{-3, 0, 0}, 1}, f.leaf_ids, f.world);
    CHECK(f.leaf_ids.empty());
    CHECK(f.at(0) == 0 && f.at(1) == 0 && f.at(2) == 0);
}

static void shift_chunks_x_out_of_range() {  // worldsvo.rs:357-385
    ShiftFixture f;
    vx::systems::shift_chunks(SvoCoordSpace{ChunkPos{3, 0, 0}, 1}, f.leaf_ids, f.world);
    CHECK(f.leaf_ids.empty());
    CHECK(f.at(0) == 0 && f.at(1) == 0 && f.at(2) == 0);
}

// ---------------------------------------------------------------------------------------------------------


// ---------------------------------------------------------------------------------------------------------
// physics  (src/systems/physics.rs:216-494)
// ---------------------------------------------------------------------------------------------------------

namespace {
using vx::Vec3;
using vx::systems::AABBDef;
using vx::systems::Entity;
using vx::systems::EntityCapabilities;

// MockRaycaster (physics.rs:226-250): asserts the batch it is handed and answers with canned AABB results
struct MockRaycaster final : vx::systems::Raycaster {
    std::vector<vx::Aabb> expected;
    std::vector<vx::AabbResult> answer;
    mutable int calls = 0;
    void raycast(vx::PickerBatch& batch, vx::PickerBatchResult& result) const override {
        ++calls;
        CHECK(batch.rays.empty());
        CHECK(batch.aabbs.size() == expected.size());
        for (size_t i = 0; i < expected.size() && i < batch.aabbs.size(); ++i) {
            CHECK(batch.aabbs[i].pos == expected[i].pos);
            CHECK(batch.aabbs[i].offset == expected[i].offset);
            CHECK(batch.aabbs[i].extents == expected[i].extents);
        }
        result.rays.clear();
        result.aabbs = answer;
    }
};

const EntityCapabilities kTestCaps{false, false, 0.008f, 3.0f};
const AABBDef kUnitBox{Vec3{0, 0, 0}, Vec3{1, 1, 1}};
}  // namespace

static void physics_step() {  // physics.rs:254-290
    Entity e(Vec3{0, 0, 0}, kUnitBox);
    e.caps = kTestCaps;
    MockRaycaster mock;
    mock.expected = {vx::Aabb{e.position, e.aabb_def.offset, e.aabb_def.extents}};
    mock.answer = {vx::AabbResult{}};
    vx::systems::Physics physics;
    physics.step(1.0f, mock, e);
    CHECK(mock.calls == 1);
    CHECK((e.position == Vec3{0.0f, -0.008f, 0.0f}));
    CHECK((e.velocity == Vec3{0.0f, -0.008f, 0.0f}));
    CHECK(e.state.is_grounded == false);
}

static void physics_step_many() {  // physics.rs:294-493
    struct Case {
        const char* name;
        Vec3 position, velocity;
        EntityCapabilities caps;
        vx::AabbResult aabb_result;
        Vec3 expected_position, expected_velocity;
        bool expected_grounded;
    };
    const EntityCapabilities clip{true, false, 0.008f, 3.0f}, fly{false, true, 0.008f, 3.0f};
    const Vec3 none{-1, -1, -1};
    const std::vector<Case> cases = {
        {"falling - first time", {0, 0, 0}, {0, 0, 0}, kTestCaps, {{-1, 1, -1}, none}, {0, -0.008f, 0}, {0, -0.008f, 0}, false},
        {"falling - second time", {0, -0.008f, 0}, {0, -0.008f, 0}, kTestCaps, {{-1, 1, -1}, none}, {0, -0.024f, 0}, {0, -0.016f, 0}, false},
        {"falling - hitting floor", {0, -0.024f, 0}, {0, -0.016f, 0}, kTestCaps, {{-1, 0.01f, -1}, none}, {0, -0.0335f, 0}, {0, 0, 0}, true},
        {"falling - hitting floor with wall clip enabled", {0, -0.024f, 0}, {0, -0.016f, 0}, clip, {{-1, 0.01f, -1}, none}, {0, -0.0335f, 0}, {0, 0, 0}, true},
        {"falling - max velocity", {0, 0, 0}, {0, -4, 0}, kTestCaps, {{-1, 10, -1}, none}, {0, -3, 0}, {0, -3, 0}, false},
        {"jumping - no velocity limit", {0, 0, 0}, {0, 5, 0}, kTestCaps, {none, none}, {0, 4.992f, 0}, {0, 4.992f, 0}, false},
        {"jumping - with collision", {0, 0, 0}, {0, 5, 0}, kTestCaps, {none, {-1, 2, -1}}, {0, 1.9995f, 0}, {0, 4.992f, 0}, false},
        {"jumping - after collision for velocity reset", {0, 1.9995f, 0}, {0, 1.9995f, 0}, kTestCaps, {none, {-1, 0.0005f, -1}}, {0, 1.9995f, 0}, {0, 1.9915f, 0}, false},
        {"jumping - with collision and wall clip enabled", {0, 0, 0}, {0, 5, 0}, clip, {none, {-1, 2, -1}}, {0, 1.9995f, 0}, {0, 4.992f, 0}, false},
        {"flying - ground state not set", {0, 5, 0}, {3, -5, 3}, fly, {{-1, 5, -1}, {2, -1, 2}}, {3, 0, 3}, {3, -5, 3}, false},
        {"horizontal positive collision", {0, 0, 0}, {2, 0, 2}, kTestCaps, {{-1, 0, -1}, {1, -1, 1}}, {0.9995f, 0, 0.9995f}, {2, 0, 2}, true},
        {"horizontal negative collision", {0, 0, 0}, {-2, 0, -2}, kTestCaps, {{1, 0, 1}, none}, {-0.9995f, 0, -0.9995f}, {-2, 0, -2}, true},
        {"horizontal positive collision - with wall clip enabled", {0, 0, 0}, {2, 0, 2}, clip, {{-1, 0, -1}, {1, -1, 1}}, {2, 0, 2}, {2, 0, 2}, true},
    };
    std::vector<Entity> entities;
    MockRaycaster mock;
    for (const Case& c : cases) {
        Entity e(c.position, kUnitBox);
        e.velocity = c.velocity;
        e.caps = c.caps;
        mock.expected.push_back(vx::Aabb{c.position, kUnitBox.offset, kUnitBox.extents});
        mock.answer.push_back(c.aabb_result);
        entities.push_back(e);
    }
    vx::systems::Physics physics;
    physics.step_many(1.0f, mock, entities);
    CHECK(mock.calls == 1);
    for (size_t i = 0; i < cases.size(); ++i) {
        const bool ok = entities[i].position == cases[i].expected_position && entities[i].velocity == cases[i].expected_velocity &&
                        entities[i].state.is_grounded == cases[i].expected_grounded;
        if (!ok) {
            std::printf("  entity case '%s': pos (%.9g %.9g %.9g) vel (%.9g %.9g %.9g) grounded %d\n", cases[i].name, entities[i].position.x,
                        entities[i].position.y, entities[i].position.z, entities[i].velocity.x, entities[i].velocity.y, entities[i].velocity.z,
                        int(entities[i].state.is_grounded));
            ++g_failures;
        }
    }
}

static void physics_apply_axial() {  // physics.rs:173-185, the branches one by one
    using vx::systems::Physics;
    CHECK(Physics::apply_axial_physics(0.5f, -1.0f, 3.0f) == 0.5f);       // no obstacle in the direction of travel
    CHECK(Physics::apply_axial_physics(-0.5f, 3.0f, -1.0f) == -0.5f);
    CHECK(Physics::apply_axial_physics(0.5f, 0.0009f, -1.0f) == 0.0f);    // closer than 2 epsilon: stop
    CHECK(Physics::apply_axial_physics(0.5f, 0.25f, -1.0f) == 0.25f - 0.0005f);
    CHECK(Physics::apply_axial_physics(-0.5f, -1.0f, 0.25f) == -(0.25f - 0.0005f));
    CHECK(Physics::apply_axial_physics(0.125f, 0.25f, -1.0f) == 0.125f);  // slower than the distance: unchanged
    CHECK(Physics::apply_axial_physics(0.0f, 5.0f, 0.0001f) == 0.0f);     // speed 0 looks at the negative side
}


// ---------------------------------------------------------------------------------------------------------
// chunk loader  (src/systems/chunkloader.rs:146-267)
// ---------------------------------------------------------------------------------------------------------

namespace {
using vx::systems::ChunkEvent;
ChunkEvent ev_load(int x, int y, int z, uint8_t lod) { return ChunkEvent{ChunkEvent::Load, vx::ChunkPos{x, y, z}, lod}; }
ChunkEvent ev_unload(int x, int y, int z) { return ChunkEvent{ChunkEvent::Unload, vx::ChunkPos{x, y, z}, 0}; }
void check_events(std::vector<ChunkEvent> got, const std::vector<ChunkEvent>& want, int line) {
    std::sort(got.begin(), got.end());  // the reference's tests sort by the derived Ord before comparing
    bool ok = got.size() == want.size();
    for (size_t i = 0; ok && i < got.size(); ++i) ok = got[i] == want[i];
    if (!ok) {
        std::printf("  event list mismatch at line %d (%zu events, expected %zu)\n", line, got.size(), want.size());
        ++g_failures;
    }
}
// chunkloader.rs:243-266
std::vector<int> lod_scale_on_x_axis(const std::vector<ChunkEvent>& events, int z) {
    std::map<int, int> columns;
    for (const ChunkEvent& e : events)
        if (e.kind != ChunkEvent::Unload && e.pos.z == z) columns[e.pos.x] = e.lod;
    std::vector<int> scale;
    for (auto& kv : columns) scale.push_back(kv.second);
    return scale;
}
}  // namespace

static void chunkloader_load_and_unload() {  // chunkloader.rs:150-213
    vx::systems::ChunkLoader cl(1, 0, 1);
    check_events(cl.update(0, 0, 0), {ev_load(-1, 0, 0, 5), ev_load(0, 0, -1, 5), ev_load(0, 0, 0, 5), ev_load(0, 0, 1, 5), ev_load(1, 0, 0, 5)}, __LINE__);
    CHECK(cl.update(16, 16, 16).empty());  // same chunk
    check_events(cl.update(32, 0, 0), {ev_load(1, 0, -1, 5), ev_load(1, 0, 1, 5), ev_load(2, 0, 0, 5), ev_unload(-1, 0, 0), ev_unload(0, 0, -1), ev_unload(0, 0, 1)},
                 __LINE__);
    check_events(cl.update(128, 0, 0), {ev_load(3, 0, 0, 5), ev_load(4, 0, -1, 5), ev_load(4, 0, 0, 5), ev_load(4, 0, 1, 5), ev_load(5, 0, 0, 5), ev_unload(0, 0, 0),
                                        ev_unload(1, 0, -1), ev_unload(1, 0, 0), ev_unload(1, 0, 1), ev_unload(2, 0, 0)}, __LINE__);
    check_events(cl.update(128, 64, 0), {ev_unload(3, 0, 0), ev_unload(4, 0, -1), ev_unload(4, 0, 0), ev_unload(4, 0, 1), ev_unload(5, 0, 0)}, __LINE__);
    CHECK(cl.update(0, 64, 0).empty());  // y still out of range: nothing to do anywhere
    CHECK(cl.loaded_count() == 0);
}

static void chunkloader_changing_lod() {  // chunkloader.rs:217-241
    vx::systems::ChunkLoader cl(25, 0, 1);
    const std::vector<ChunkEvent> first = cl.update(0, 0, 0);
    const std::vector<int> z0 = {2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 4, 4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 2, 2, 2, 2, 2, 2};
    const std::vector<int> z1 = {2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 5, 4, 4, 4, 4, 4, 4, 3, 3, 3, 3, 3, 3, 3, 2, 2, 2, 2, 2};
    CHECK_SEQ(lod_scale_on_x_axis(first, -1), z1);
    CHECK_SEQ(lod_scale_on_x_axis(first, 0), z0);
    CHECK_SEQ(lod_scale_on_x_axis(first, 1), z1);
    // nearest first: the list is ordered by distance to the target chunk
    for (size_t i = 1; i < first.size(); ++i) CHECK(first[i - 1].pos.dst_sq(vx::ChunkPos{0, 0, 0}) <= first[i].pos.dst_sq(vx::ChunkPos{0, 0, 0}));
    CHECK(first.front().pos == (vx::ChunkPos{0, 0, 0}));
    // one chunk further in +x: one chunk per LOD ring changes on each row, plus one new chunk on z = 0
    const std::vector<ChunkEvent> second = cl.update(32, 0, 0);
    const std::vector<int> change = {2, 3, 4, 5, 4, 3, 2};
    CHECK_SEQ(lod_scale_on_x_axis(second, -1), change);
    CHECK_SEQ(lod_scale_on_x_axis(second, 0), change);
    CHECK_SEQ(lod_scale_on_x_axis(second, 1), change);
}

// The reference's own formulation (chunkloader.rs:58-128): one map entry per loaded chunk, every chunk of the cylinder looked up on every call.
// ChunkLoader keeps a table per column instead (40 ms -> 2 ms for a re-centre at radius 40): the same events must come out, step by step,
// over a walk with vertical moves, jumps, radius changes and chunks loaded behind the loader's back.
namespace {
struct PerChunkLoader {
    uint32_t radius; int32_t start_y, end_y;
    std::map<std::tuple<int, int, int>, uint8_t> loaded;
    std::vector<ChunkEvent> update(float px, float py, float pz) {
        std::vector<ChunkEvent> events;
        const vx::ChunkPos current = vx::ChunkPos::from_block_pos(int32_t(px), int32_t(py), int32_t(pz));
        const int32_t r = int32_t(radius);
        for (int32_t dx = -r; dx <= r; ++dx)
            for (int32_t dz = -r; dz <= r; ++dz) {
                if (dx * dx + dz * dz > r * r) continue;
                vx::ChunkPos pos{current.x + dx, 0, current.z + dz};
                const uint8_t lod = vx::systems::ChunkLoader::calculate_lod(current, pos);
                for (int32_t y = start_y; y < end_y; ++y) {
                    if (y - current.y < -r || y - current.y > r) continue;
                    auto key = std::make_tuple(pos.x, y, pos.z);
                    auto it = loaded.find(key);
                    if (it != loaded.end()) {
                        if (it->second != lod) { events.push_back(ChunkEvent{ChunkEvent::LodChange, vx::ChunkPos{pos.x, y, pos.z}, lod}); it->second = lod; }
                    } else {
                        events.push_back(ChunkEvent{ChunkEvent::Load, vx::ChunkPos{pos.x, y, pos.z}, lod});
                        loaded.emplace(key, lod);
                    }
                }
            }
        for (auto it = loaded.begin(); it != loaded.end();) {
            const int32_t dx = std::get<0>(it->first) - current.x, dy = std::abs(std::get<1>(it->first) - current.y), dz = std::get<2>(it->first) - current.z;
            if (dy > r || dx * dx + dz * dz > r * r) {
                events.push_back(ChunkEvent{ChunkEvent::Unload, vx::ChunkPos{std::get<0>(it->first), std::get<1>(it->first), std::get<2>(it->first)}, 0});
                it = loaded.erase(it);
            } else {
                ++it;
            }
        }
        return events;
    }
};
}  // namespace

static void chunkloader_matches_the_per_chunk_map() {
    vx::systems::ChunkLoader cl(5, -2, 7);
    PerChunkLoader ref{5, -2, 7, {}};
    uint32_t rng = 12345u;
    auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    float x = 3.0f, y = 40.0f, z = -7.0f;
    for (int step = 0; step < 400; ++step) {
        const uint32_t kind = next() % 16;
        if (kind < 9) { x += float(int(next() % 41) - 20); z += float(int(next() % 41) - 20); }          // a walk: often into a neighbouring chunk
        else if (kind < 12) y += float(int(next() % 161) - 80);                                             // up and down: layers enter and leave the window
        else if (kind == 12) { x += float(int(next() % 2001) - 1000); z += float(int(next() % 2001) - 1000); }  // a jump: nothing stays
        else if (kind == 13) { const uint32_t r = 3 + next() % 5; cl.set_radius(r); ref.radius = r; }
        else if (kind >= 14) {  // a chunk loaded behind the loader's back, in reach or not, in or outside the layers
            const vx::ChunkPos c = vx::ChunkPos::from_block_pos(int32_t(x), int32_t(y), int32_t(z));
            const vx::ChunkPos p{c.x + int(next() % 15) - 7, int(next() % 13) - 4, c.z + int(next() % 15) - 7};
            const uint8_t lod = uint8_t(2 + next() % 4);
            cl.add_loaded_chunk(p, lod);
            ref.loaded[std::make_tuple(p.x, p.y, p.z)] = lod;
        }
        std::vector<ChunkEvent> got = cl.update(x, y, z), want = ref.update(x, y, z);
        // nearest first (chunkloader.rs:123-126)
        const vx::ChunkPos current = vx::ChunkPos::from_block_pos(int32_t(x), int32_t(y), int32_t(z));
        for (size_t i = 1; i < got.size(); ++i) CHECK(got[i - 1].pos.dst_sq(current) <= got[i].pos.dst_sq(current));
        std::sort(want.begin(), want.end());
        check_events(got, want, __LINE__);
        CHECK(cl.loaded_count() == ref.loaded.size());
        if (g_failures) { std::printf("  first mismatch at step %d (kind %u)\n", step, kind); return; }
    }
    for (const auto& kv : ref.loaded) CHECK(cl.is_loaded(vx::ChunkPos{std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first)}));
}

static void camera_is_in_frustum() {  // camera.rs:106-140
    vx::graphics::Camera camera(72.0f, 1.0f, 0.01f, 30.0f);
    camera.position[0] = camera.position[1] = camera.position[2] = 0.0f;
    camera.forward[0] = 0.0f; camera.forward[1] = 0.0f; camera.forward[2] = 1.0f;
    auto in = [&](float x, float y, float z, float r) {
        const float p[3] = {x, y, z};
        return camera.is_in_frustum(p, r);
    };
    CHECK(!in(0, 0, 0, 0)); CHECK(in(0, 0, 10, 0)); CHECK(in(0, 0, 29, 0)); CHECK(!in(0, 0, 31, 0)); CHECK(in(0, 0, 0, 1)); CHECK(in(0, 0, 31, 1));
    CHECK(in(0, 0, 3, 0)); CHECK(in(0, 2, 3, 0)); CHECK(!in(0, 3, 3, 0)); CHECK(in(0, -2, 3, 0)); CHECK(!in(0, -3, 3, 0)); CHECK(in(0, 3, 3, 1)); CHECK(in(0, -3, 3, 1));
    CHECK(in(0, 0, 3, 0)); CHECK(in(2, 0, 3, 0)); CHECK(!in(3, 0, 3, 0)); CHECK(in(-2, 0, 3, 0)); CHECK(!in(-3, 0, 3, 0)); CHECK(in(3, 0, 3, 1)); CHECK(in(-3, 0, 3, 1));
}

static void sort_chunks_by_view_frustum() {  // world.rs:233-262 (the reference has no test of its own for it)
    vx::systems::ChunkLoader cl(6, 0, 1);
    const std::vector<ChunkEvent> events = cl.update(16.0f, 16.0f, 16.0f);  // nearest first around chunk (0,0,0)
    vx::graphics::Camera camera(72.0f, 16.0f / 9.0f, 0.01f, 1024.0f);
    camera.position[0] = 16.0f; camera.position[1] = 16.0f; camera.position[2] = 16.0f;
    camera.forward[0] = 1.0f; camera.forward[1] = 0.0f; camera.forward[2] = 0.0f;
    const std::vector<ChunkEvent> sorted = vx::systems::sort_chunks_by_view_frustum(events, camera);
    CHECK(sorted.size() == events.size());
    // a permutation
    std::vector<ChunkEvent> a = events, b = sorted;
    std::sort(a.begin(), a.end());
    std::sort(b.begin(), b.end());
    CHECK(a == b);
    // the visible ones first, in the loader's order; then the rest, front to back
    size_t n_visible = 0;
    std::vector<ChunkEvent> visible_in_loader_order;
    for (const ChunkEvent& e : events) {
        const float c[3] = {float(e.pos.x * 32 + 16), float(e.pos.y * 32 + 16), float(e.pos.z * 32 + 16)};
        if (camera.is_in_frustum(c, 32.0f)) visible_in_loader_order.push_back(e);
    }
    n_visible = visible_in_loader_order.size();
    CHECK(n_visible > 10 && n_visible < events.size() - 10);
    for (size_t i = 0; i < n_visible; ++i) CHECK(sorted[i] == visible_in_loader_order[i]);
    float last = -2.0f;
    for (size_t i = n_visible; i < sorted.size(); ++i) {
        float t[3] = {float(sorted[i].pos.x * 32) - 16.0f, float(sorted[i].pos.y * 32) - 16.0f, float(sorted[i].pos.z * 32) - 16.0f};
        const float len = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        const float k = -(t[0] / len);  // -dot(normalize(t), forward)
        CHECK(k >= last);
        last = k;
    }
    CHECK(sorted.back().pos.x < 0);  // what is behind the camera comes last
    CHECK(sorted.front().pos == (vx::ChunkPos{0, 0, 0}));  // the chunk the camera is in: its sphere reaches the near plane
}

// The writers' large ranges are copied on several threads (range_buffer.hpp, copy_bytes: a whole depth-14 ESVO world is 6.7 GB): every byte arrives,
// whatever the length does to the last thread's share; nothing of the pieces' book-keeping in the reference (its write_to is one memcpy, esvo.rs:291-339).
static void copy_bytes_in_pieces() {
    for (size_t n : {size_t(0), size_t(5), (size_t(64) << 20) - 1, size_t(64) << 20, (size_t(70) << 20) + 4097, (size_t(96) << 20) + 1}) {
        std::vector<uint8_t> src(n), dst(n + 8, 0xEE);
        for (size_t i = 0; i < n; ++i) src[i] = uint8_t((i * 2654435761u) >> 13);
        vx::copy_bytes(dst.data() + 4, src.data(), n);
        CHECK(n == 0 || std::memcmp(dst.data() + 4, src.data(), n) == 0);
        for (size_t k = 0; k < 4; ++k) CHECK(dst[k] == 0xEE && dst[n + 4 + k] == 0xEE);  // (nothing in front of it or behind it)
    }
}

int main(int argc, char** argv) {
    const std::map<std::string, std::function<void()>> cases = {
        {"copy_bytes_in_pieces", copy_bytes_in_pieces},
        {"camera_is_in_frustum", camera_is_in_frustum},
        {"sort_chunks_by_view_frustum", sort_chunks_by_view_frustum},
        {"octree_add_leaf_single", octree_add_leaf_single},
        {"octree_add_leaf_multiple", octree_add_leaf_multiple},
        {"octree_remove_and_add_leaf", octree_remove_and_add_leaf},
        {"octree_expand_wraps_root_as_child0", octree_expand_wraps_root_as_child0},
        {"octree_move_leaf", octree_move_leaf},
        {"octree_construct_octants", octree_construct_octants},
        {"octree_compact", octree_compact},
        {"position_required_depth", position_required_depth},
        {"range_buffer_insert_remove", range_buffer_insert_remove},
        {"range_buffer_merge_ranges", range_buffer_merge_ranges},
        {"esvo_serialize_with_lod", esvo_serialize_with_lod},
        {"esvo_serialize_world", esvo_serialize_world},
        {"esvo_serialize_with_remove_and_move", esvo_serialize_with_remove_and_move},
        {"csvo_serialize_octant_single_leaf", csvo_serialize_octant_single_leaf},
        {"csvo_serialize_octant_multiple_leaves", csvo_serialize_octant_multiple_leaves},
        {"csvo_serialize_octant_chunk", csvo_serialize_octant_chunk},
        {"csvo_serialize_octant_chunk_with_lod", csvo_serialize_octant_chunk_with_lod},
        {"csvo_serialize_world", csvo_serialize_world},
        {"chunk_pos_from_block_pos", chunk_pos_from_block_pos},
        {"block_pos_roundtrip", block_pos_roundtrip},
        {"chunk_storage_depth_and_blocks", chunk_storage_depth_and_blocks},
        {"coord_space_positive", coord_space_positive},
        {"coord_space_negative", coord_space_negative},
        {"coord_space_cnv_chunk_pos", coord_space_cnv_chunk_pos},
        {"shift_chunks_x_positive", shift_chunks_x_positive},
        {"shift_chunks_x_negative", shift_chunks_x_negative},
        {"shift_chunks_x_out_of_range", shift_chunks_x_out_of_range},
        {"chunkloader_load_and_unload", chunkloader_load_and_unload},
        {"chunkloader_changing_lod", chunkloader_changing_lod},
        {"chunkloader_matches_the_per_chunk_map", chunkloader_matches_the_per_chunk_map},
        {"physics_step", physics_step},
        {"physics_step_many", physics_step_many},
        {"physics_apply_axial", physics_apply_axial},
    };
    if (argc >= 2 && std::string(argv[1]) == "--list") {
        for (auto& c : cases) std::printf("%s\n", c.first.c_str());
        return 0;
    }
    int failed_cases = 0;
    for (auto& c : cases) {
        if (argc >= 2 && c.first != argv[1]) continue;
        g_failures = 0;
        c.second();
        std::printf("%s %s\n", g_failures ? "FAIL" : "ok  ", c.first.c_str());
        if (g_failures) ++failed_cases;
    }
    return failed_cases ? 1 : 0;
}
