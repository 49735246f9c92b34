// The camera as far as the chunk stream needs it: radar frustum culling (src/graphics/camera.rs:69-99) and the order in which the
// chunk loader's events are worked off -- chunks inside the view frustum first, the rest from the forward to the backward side of
// the camera (src/gamelogic/world.rs:233-262). The view matrix itself stays with the caller (svo.rs:197).
#pragma once

#include <algorithm>
#include <cmath>
#include <vector>

#include "chunkloader.hpp"

namespace vx {
namespace graphics {

struct Camera {
    float position[3] = {0.0f, 0.0f, 0.0f};
    float forward[3] = {0.0f, 0.0f, -1.0f};
    float up[3] = {0.0f, 1.0f, 0.0f};
    float fov_y_deg = 72.0f, aspect_ratio = 1.0f, near = 0.01f, far = 1024.0f;  // src/gamelogic/world.rs:103, src/main.rs:97

    Camera() = default;
    Camera(float fov_y_deg_, float aspect_ratio_, float near_, float far_) : fov_y_deg(fov_y_deg_), aspect_ratio(aspect_ratio_), near(near_), far(far_) {}

    static float dot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
    static void cross(const float a[3], const float b[3], float out[3]) {
        out[0] = a[1] * b[2] - a[2] * b[1];
        out[1] = a[2] * b[0] - a[0] * b[2];
        out[2] = a[0] * b[1] - a[1] * b[0];
    }

    // camera.rs:43-45
    void right(float out[3]) const {
        cross(forward, up, out);
        const float len = std::sqrt(dot(out, out));
        out[0] /= len; out[1] /= len; out[2] /= len;
    }

    // camera.rs:75-99: is the sphere (point, r) inside the frustum? The point goes to view space; the frustum's half height and
    // half width at its depth follow from the distance to the near plane and the field of view.
    bool is_in_frustum(const float point[3], float r) const {
        const float cp[3] = {point[0] - position[0], point[1] - position[1], point[2] - position[2]};
        float cz = dot(cp, forward);
        if (cz + r < near || cz - r > far) return false;
        cz = cz - near;
        float rt[3], upv[3];
        right(rt);
        cross(forward, rt, upv);
        const float cy = dot(cp, upv);
        const float hh = cz * std::tan(fov_y_deg * (3.14159265358979323846f / 180.0f) / 2.0f);
        if (cy + r < -hh || cy - r > hh) return false;
        const float cx = dot(cp, rt);
        const float wh = hh * aspect_ratio;
        if (cx + r < -wh || cx - r > wh) return false;
        return true;
    }
};

}  // namespace graphics

namespace systems {

// world.rs:233-262: events of chunks whose centre (block position + 16) is in the frustum with a 32-block radius come first, in
// the order the loader gave them (nearest first); the others follow, ordered by the angle between the camera's forward vector
// and the direction to the chunk's origin (stable, like the reference's sort_by).
inline std::vector<ChunkEvent> sort_chunks_by_view_frustum(const std::vector<ChunkEvent>& events, const graphics::Camera& camera) {
    std::vector<ChunkEvent> visible, other;
    for (const ChunkEvent& e : events) {
        const float centre[3] = {float(e.pos.x * 32 + 16), float(e.pos.y * 32 + 16), float(e.pos.z * 32 + 16)};
        (camera.is_in_frustum(centre, 32.0f) ? visible : other).push_back(e);
    }
    auto key = [&](const ChunkEvent& e) {
        float t[3] = {float(e.pos.x * 32) - camera.position[0], float(e.pos.y * 32) - camera.position[1], float(e.pos.z * 32) - camera.position[2]};
        const float len = std::sqrt(graphics::Camera::dot(t, t));
        t[0] /= len; t[1] /= len; t[2] /= len;
        return -graphics::Camera::dot(t, camera.forward);
    };
    // f32::total_cmp orders NaN (a chunk AT the camera position) above every number
    std::stable_sort(other.begin(), other.end(), [&](const ChunkEvent& a, const ChunkEvent& b) {
        const float ka = key(a), kb = key(b);
        if (std::isnan(ka) || std::isnan(kb)) return !std::isnan(ka) && std::isnan(kb);
        return ka < kb;
    });
    visible.insert(visible.end(), other.begin(), other.end());
    return visible;
}

}  // namespace systems
}  // namespace vx
