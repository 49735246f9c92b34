// Ray / AABB pick batches over the picker kernel: what `graphics::svo_picker` does in the reference
// (src/graphics/svo_picker.rs). A batch is flattened into PickerTasks (rays first, then every AABB's
// ray fan), sent through vx_raycast, and the PickerResults are folded back per ray / per AABB.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "voxel_hip.h"

namespace vx {

constexpr size_t MAX_SVO_PICKER_JOBS = 100;  // svo_picker.rs:5 (capacity hint only here; vx_raycast has no cap)

struct Vec3 {
    float x = 0, y = 0, z = 0;
    bool operator==(const Vec3& o) const { return x == o.x && y == o.y && z == o.z; }
};

struct Ray {
    Vec3 pos, dir;
    float max_dst = 0;
};

// svo_picker.rs:138-154: only if dst != -1 are the other fields valid
struct RayResult {
    float dst = -1.0f;
    bool inside_voxel = false;
    Vec3 pos, normal;
    bool did_hit() const { return dst != -1.0f; }
};

// svo_picker.rs:163-176: shortest hit distance per axis in negative / positive direction, -1 = none
struct AabbResult {
    Vec3 neg{-1.0f, -1.0f, -1.0f}, pos{-1.0f, -1.0f, -1.0f};
};

struct Aabb {
    Vec3 pos, offset, extents;

    // svo_picker.rs:183-243: grid points every <= 1 block across the box; at each point one ray per axis on
    // which the point lies on the box boundary (3 per corner, 2 per edge point, 1 per face point), max_dst 10
    size_t generate_picker_tasks(std::vector<vx_picker_task>& dst) const {
        const int blocks[3] = {int(std::ceil(extents.x)), int(std::ceil(extents.y)), int(std::ceil(extents.z))};
        const float step[3] = {extents.x / float(blocks[0]), extents.y / float(blocks[1]), extents.z / float(blocks[2])};
        size_t n = 0;
        for (int x = 0; x <= blocks[0]; ++x)
            for (int y = 0; y <= blocks[1]; ++y)
                for (int z = 0; z <= blocks[2]; ++z) {
                    const int axes[3] = {x, y, z};
                    for (int i = 0; i < 3; ++i) {
                        const int v = axes[i];
                        if (v != 0 && v != blocks[i]) continue;
                        vx_picker_task t;
                        std::memset(&t, 0, sizeof t);
                        t.max_dst = 10.0f;
                        t.pos[0] = pos.x + offset.x + float(x) * step[0];
                        t.pos[1] = pos.y + offset.y + float(y) * step[1];
                        t.pos[2] = pos.z + offset.z + float(z) * step[2];
                        t.dir[i] = v == 0 ? -1.0f : 1.0f;
                        dst.push_back(t);
                        ++n;
                    }
                }
        return n;
    }

    // svo_picker.rs:245-299: same walk, keeping the minimum hit distance per axis and direction
    size_t parse_picker_results(const vx_picker_result* data, AabbResult& out) const {
        const int blocks[3] = {int(std::ceil(extents.x)), int(std::ceil(extents.y)), int(std::ceil(extents.z))};
        out = AabbResult{};
        float* refs[6] = {&out.pos.x, &out.neg.x, &out.pos.y, &out.neg.y, &out.pos.z, &out.neg.z};
        size_t k = 0;
        for (int x = 0; x <= blocks[0]; ++x)
            for (int y = 0; y <= blocks[1]; ++y)
                for (int z = 0; z <= blocks[2]; ++z) {
                    const int axes[3] = {x, y, z};
                    for (int i = 0; i < 3; ++i) {
                        const int v = axes[i];
                        if (v != 0 && v != blocks[i]) continue;
                        const float dst = data[k++].dst;
                        if (dst == -1.0f) continue;
                        float& r = *refs[i * 2 + (v == 0 ? 1 : 0)];
                        r = r == -1.0f ? dst : std::fmin(r, dst);
                    }
                }
        return k;
    }
};

struct PickerBatchResult {
    std::vector<RayResult> rays;
    std::vector<AabbResult> aabbs;
    void reset() { rays.clear(); aabbs.clear(); }
};

struct PickerBatch {
    std::vector<Ray> rays;
    std::vector<Aabb> aabbs;

    void reset() { rays.clear(); aabbs.clear(); }
    void add_ray(Vec3 pos, Vec3 dir, float max_dst) { rays.push_back(Ray{pos, dir, max_dst}); }
    void add_aabb(const Aabb& a) { aabbs.push_back(a); }

    // svo_picker.rs:63-80
    size_t serialize_tasks(std::vector<vx_picker_task>& tasks) const {
        tasks.clear();
        for (const Ray& r : rays) {
            vx_picker_task t;
            std::memset(&t, 0, sizeof t);
            t.max_dst = r.max_dst;
            t.pos[0] = r.pos.x; t.pos[1] = r.pos.y; t.pos[2] = r.pos.z;
            t.dir[0] = r.dir.x; t.dir[1] = r.dir.y; t.dir[2] = r.dir.z;
            tasks.push_back(t);
        }
        for (const Aabb& a : aabbs) a.generate_picker_tasks(tasks);
        return tasks.size();
    }

    // svo_picker.rs:84-104
    void deserialize_results(const vx_picker_result* results, PickerBatchResult& dst) const {
        size_t off = 0;
        for (size_t i = 0; i < rays.size(); ++i, ++off) {
            const vx_picker_result& r = results[off];
            dst.rays.push_back(RayResult{r.dst, r.inside_voxel != 0, Vec3{r.pos[0], r.pos[1], r.pos[2]}, Vec3{r.normal[0], r.normal[1], r.normal[2]}});
        }
        for (const Aabb& a : aabbs) {
            AabbResult res;
            off += a.parse_picker_results(results + off, res);
            dst.aabbs.push_back(res);
        }
    }
};

}  // namespace vx
