// Pointer-free octree with an octant free list: the host-side container the ESVO/CSVO serializers walk.
//
// Mirrors the behaviour of the reference's `world::hds::octree::Octree<T>` (src/world/hds/octree.rs:56-418):
// same child index (x + 2y + 4z, octree.rs:21-23), same `required_depth` (octree.rs:25-28), same octant-id
// allocation order (sequential ids, LIFO free list, octree.rs:379-412) so that `LeafId`s and whole-structure
// known-answer tests carry over, and the same quirks the serializers depend on:
//   * `expand` wraps the previous root as child 0 of a new root, also when that root is empty
//     (octree.rs:311-324) -- the empty chain IS serialized by ESVO/CSVO (last paragraph);
//   * removing leaves never prunes parents; only `compact` does (octree.rs:236-238, 341-376).
#pragma once

#include <array>
#include <cstdint>
#include <functional>
#include <optional>
#include <utility>
#include <vector>

namespace vx {

using OctantId = uint32_t;

struct Position {
    uint32_t x = 0, y = 0, z = 0;

    // child slot of this (already reduced to 0/1 per axis) position, octree.rs:21-23
    uint8_t idx() const { return static_cast<uint8_t>(x + y * 2 + z * 4); }

    // number of levels an octree needs so that this position is addressable, octree.rs:25-28
    uint8_t required_depth() const {
        uint32_t m = x > y ? x : y;
        m = m > z ? m : z;
        if (m < 1) m = 1;
        uint8_t bits = 0;
        while (m) { ++bits; m >>= 1; }
        return bits;  // floor(log2(m)) + 1
    }

    bool operator==(const Position& o) const { return x == o.x && y == o.y && z == o.z; }
};

// Identifies a leaf by the octant holding it and the child slot, octree.rs:11-15.
struct LeafId {
    OctantId parent = 0;
    uint8_t idx = 0;
    bool operator==(const LeafId& o) const { return parent == o.parent && idx == o.idx; }
    bool operator!=(const LeafId& o) const { return !(*this == o); }
};

enum class ChildKind : uint8_t { None = 0, Octant = 1, Leaf = 2 };

template <class T>
struct Child {
    ChildKind kind = ChildKind::None;
    OctantId octant = 0;
    std::optional<T> leaf;

    static Child none() { return Child{}; }
    static Child make_octant(OctantId id) { Child c; c.kind = ChildKind::Octant; c.octant = id; return c; }
    static Child make_leaf(T v) { Child c; c.kind = ChildKind::Leaf; c.leaf.emplace(std::move(v)); return c; }

    bool is_none() const { return kind == ChildKind::None; }
    bool is_octant() const { return kind == ChildKind::Octant; }
    bool is_leaf() const { return kind == ChildKind::Leaf; }
    const T* leaf_value() const { return is_leaf() ? &*leaf : nullptr; }
    T* leaf_value() { return is_leaf() ? &*leaf : nullptr; }
    std::optional<T> into_leaf_value() { return is_leaf() ? std::move(leaf) : std::optional<T>{}; }
};

template <class T>
struct Octant {
    std::optional<OctantId> parent;
    uint8_t children_count = 0;
    std::array<Child<T>, 8> children;

    // swaps in the new child, keeps `children_count` in step, returns the previous child (octree.rs:490-503)
    Child<T> set_child(uint8_t idx, Child<T> child) {
        if (children[idx].is_none() && !child.is_none()) ++children_count;
        if (!children[idx].is_none() && child.is_none()) --children_count;
        std::swap(child, children[idx]);
        return child;
    }
};

template <class T>
class Octree {
public:
    std::optional<OctantId> root;
    std::vector<Octant<T>> octants;
    std::vector<OctantId> free_list;

    uint8_t depth() const { return depth_; }

    void reset() {
        root.reset();
        octants.clear();
        free_list.clear();
        depth_ = 0;
    }

    // octree.rs:101-122
    std::pair<LeafId, std::optional<T>> set_leaf(Position pos, T leaf) {
        expand_to(pos.required_depth());
        OctantId it = *root;
        uint32_t size = 1u << depth_;
        while (size >= 1) {
            size /= 2;
            const uint8_t idx = Position{pos.x / size, pos.y / size, pos.z / size}.idx();
            pos.x %= size; pos.y %= size; pos.z %= size;
            if (size == 1) {
                Child<T> prev = octants[it].set_child(idx, Child<T>::make_leaf(std::move(leaf)));
                return {LeafId{it, idx}, prev.into_leaf_value()};
            }
            it = step_into_or_create(it, idx);
        }
        return {LeafId{}, std::nullopt};  // unreachable
    }

    // Bottom-up construction from a voxel predicate; branches without leaves are never allocated
    // (octree.rs:127-172). Octant ids come out in post-order, as in the reference.
    void construct_octants_with(uint8_t depth, const std::function<std::optional<T>(Position)>& f) {
        reset();
        const uint32_t size = 1u << depth;
        if (auto id = construct_impl(size, Position{0, 0, 0}, f)) {
            root = *id;
            depth_ = depth;
        }
    }

    // octree.rs:177-218
    std::pair<LeafId, std::optional<T>> move_leaf(LeafId leaf_id, Position to_pos) {
        expand_to(to_pos.required_depth());
        OctantId it = *root;
        Position pos = to_pos;
        uint32_t size = 1u << depth_;
        while (size >= 1) {
            size /= 2;
            const uint8_t idx = Position{pos.x / size, pos.y / size, pos.z / size}.idx();
            pos.x %= size; pos.y %= size; pos.z %= size;
            if (size == 1) {
                if (it == leaf_id.parent && idx == leaf_id.idx) return {leaf_id, std::nullopt};
                Child<T> old_leaf = octants[it].set_child(idx, Child<T>::none());
                Child<T> moved = octants[leaf_id.parent].set_child(leaf_id.idx, Child<T>::none());
                if (moved.is_leaf()) octants[it].set_child(idx, std::move(moved));
                return {LeafId{it, idx}, old_leaf.into_leaf_value()};
            }
            it = step_into_or_create(it, idx);
        }
        return {LeafId{}, std::nullopt};  // unreachable
    }

    // octree.rs:239-267
    std::pair<std::optional<T>, std::optional<LeafId>> remove_leaf(Position pos) {
        if (pos.required_depth() > depth_ || !root) return {std::nullopt, std::nullopt};
        OctantId it = *root;
        uint32_t size = 1u << depth_;
        while (size >= 1) {
            size /= 2;
            if (size == 0) break;
            const uint8_t idx = Position{pos.x / size, pos.y / size, pos.z / size}.idx();
            pos.x %= size; pos.y %= size; pos.z %= size;
            Child<T>& c = octants[it].children[idx];
            if (c.is_none()) break;
            if (c.is_octant()) { it = c.octant; continue; }
            Child<T> prev = octants[it].set_child(idx, Child<T>::none());
            return {prev.into_leaf_value(), LeafId{it, idx}};
        }
        return {std::nullopt, std::nullopt};
    }

    // octree.rs:270-281
    std::optional<T> remove_leaf_by_id(LeafId id) {
        if (id.parent >= octants.size() || !octants[id.parent].children[id.idx].is_leaf()) return std::nullopt;
        return octants[id.parent].set_child(id.idx, Child<T>::none()).into_leaf_value();
    }

    // octree.rs:284-307
    const T* get_leaf(Position pos) const {
        if (!root) return nullptr;
        OctantId it = *root;
        uint32_t size = 1u << depth_;
        while (size > 1) {
            size /= 2;
            const uint8_t idx = Position{pos.x / size, pos.y / size, pos.z / size}.idx();
            pos.x %= size; pos.y %= size; pos.z %= size;
            const Child<T>& c = octants[it].children[idx];
            if (c.is_none()) return nullptr;
            if (c.is_leaf()) return c.leaf_value();
            it = c.octant;
        }
        return nullptr;
    }

    // octree.rs:311-324: every step allocates a new root and hangs the old one (if any) under child 0
    void expand(uint8_t by) {
        for (uint8_t i = 0; i < by; ++i) {
            const OctantId new_root = new_octant(std::nullopt);
            if (root) {
                octants[*root].parent = new_root;
                octants[new_root].set_child(0, Child<T>::make_octant(*root));
            }
            root = new_root;
        }
        depth_ = static_cast<uint8_t>(depth_ + by);
    }

    void expand_to(uint8_t to) {
        if (depth_ > to) return;
        if (to > depth_) expand(static_cast<uint8_t>(to - depth_));
    }

    // Drops every octant without content, depth first; an all-empty tree resets (octree.rs:341-376).
    void compact() {
        if (!root) return;
        compact_octant(*root);
        if (octants[*root].children_count != 0) return;
        reset();
    }

private:
    uint8_t depth_ = 0;

    OctantId new_octant(std::optional<OctantId> parent) {
        if (!free_list.empty()) {
            const OctantId id = free_list.back();
            free_list.pop_back();
            octants[id].parent = parent;
            return id;
        }
        const OctantId id = static_cast<OctantId>(octants.size());
        octants.emplace_back();
        octants.back().parent = parent;
        return id;
    }

    void delete_octant(OctantId id) {
        if (octants[id].parent) {
            Octant<T>& p = octants[*octants[id].parent];
            for (uint8_t i = 0; i < 8; ++i) {
                if (p.children[i].is_octant() && p.children[i].octant == id) {
                    p.set_child(i, Child<T>::none());
                    break;
                }
            }
        }
        Octant<T>& o = octants[id];
        o.parent.reset();
        o.children_count = 0;
        for (auto& c : o.children) c = Child<T>::none();
        free_list.push_back(id);
    }

    void compact_octant(OctantId octant_id) {
        for (uint8_t i = 0; i < 8; ++i) {
            if (!octants[octant_id].children[i].is_octant()) continue;
            const OctantId id = octants[octant_id].children[i].octant;
            compact_octant(id);
            if (octants[id].children_count == 0) {
                delete_octant(id);
                octants[octant_id].set_child(i, Child<T>::none());
            }
        }
    }

    OctantId step_into_or_create(OctantId it, uint8_t idx) {
        const Child<T>& c = octants[it].children[idx];
        if (c.is_octant()) return c.octant;
        // Child::Leaf on the way down is `unreachable!` in the reference (octree.rs:232); here it is overwritten.
        const OctantId next = new_octant(it);
        octants[it].set_child(idx, Child<T>::make_octant(next));
        return next;
    }

    std::optional<OctantId> construct_impl(uint32_t size, Position pos,
                                           const std::function<std::optional<T>(Position)>& f) {
        size /= 2;
        std::optional<OctantId> new_parent;
        for (uint32_t i = 0; i < 8; ++i) {
            const Position child_pos{pos.x + size * (i & 1), pos.y + size * ((i >> 1) & 1), pos.z + size * ((i >> 2) & 1)};
            if (size > 1) {
                auto child_id = construct_impl(size, child_pos, f);
                if (!child_id) continue;
                if (!new_parent) new_parent = new_octant(std::nullopt);
                octants[*new_parent].set_child(static_cast<uint8_t>(i), Child<T>::make_octant(*child_id));
                octants[*child_id].parent = *new_parent;
                continue;
            }
            if (auto value = f(child_pos)) {
                if (!new_parent) new_parent = new_octant(std::nullopt);
                octants[*new_parent].set_child(static_cast<uint8_t>(i), Child<T>::make_leaf(std::move(*value)));
            }
        }
        return new_parent;
    }
};

// Breadth-first search for a representative leaf below `parent`, preferring the y=1 half
// (visiting order [2,3,6,7,0,1,4,5]): used when an LOD cut-off lands on an inner octant
// (src/world/hds/internal.rs:461-485).
template <class T>
const T* pick_leaf_for_lod(const Octree<T>& octree, const Octant<T>& parent) {
    static constexpr uint8_t kOrder[8] = {2, 3, 6, 7, 0, 1, 4, 5};
    for (uint8_t i : kOrder) {
        if (parent.children[i].is_leaf()) return parent.children[i].leaf_value();
    }
    for (uint8_t i : kOrder) {
        if (!parent.children[i].is_octant()) continue;
        if (const T* r = pick_leaf_for_lod(octree, octree.octants[parent.children[i].octant])) return r;
    }
    return nullptr;
}

}  // namespace vx
