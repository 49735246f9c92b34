// `graphics::Svo` of the reference (src/graphics/svo.rs:56-256) rebuilt over the C ABI of libvoxelhip.so: same methods,
// same argument meaning, same order of effects. The OpenGL objects it owned (mapped SSBO, programs, fences, RGBA32F
// framebuffer) become vx_* calls; nothing here computes a pixel.
#pragma once

#include <cmath>
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "png_io.hpp"
#include "range_buffer.hpp"
#include "svo_picker.hpp"
#include "svo_registry.hpp"
#include "voxel_hip.h"

namespace vx {
namespace graphics {

enum class SvoType : int { Esvo = VX_SVO_ESVO, Csvo = VX_SVO_CSVO };  // svo.rs:18-39

// svo.rs:75-83
struct Stats {
    size_t used_bytes = 0, capacity_bytes = 0;
    uint8_t depth = 0;
};

// svo.rs:85-106
struct RenderParams {
    float ambient_intensity = 0.3f;
    Vec3 light_dir{-1, -1, -1};
    Vec3 cam_pos, cam_fwd{0, 0, -1}, cam_up{0, 1, 0};
    float fov_y_rad = 1.2566371f;
    float aspect_ratio = 1.0f;
    std::optional<Vec3> selected_voxel;
    bool render_shadows = true;
    float shadow_distance = 500.0f;
};

// What the reference needs from a `dyn WorldSvo` in Svo::update (src/world/hds/common.rs:11-14), plus the dirty ranges the
// mapped-buffer design made implicit (the one accessor a Rust maintainer adds).
struct WorldSvoSource {
    virtual ~WorldSvoSource() = default;
    virtual uint8_t depth() const = 0;
    virtual size_t size_in_bytes() const = 0;
    virtual std::vector<Range> updated_ranges() const = 0;
    virtual bool write_changes_to(uint8_t* dst, size_t dst_len, bool reset) = 0;
};

template <class W>
struct WorldSvoRef final : WorldSvoSource {
    W& w;
    explicit WorldSvoRef(W& world) : w(world) {}
    uint8_t depth() const override { return w.depth(); }
    size_t size_in_bytes() const override { return w.size_in_bytes(); }
    std::vector<Range> updated_ranges() const override { return w.buffer.updated_ranges; }
    bool write_changes_to(uint8_t* dst, size_t dst_len, bool reset) override { return w.write_changes_to(dst, dst_len, reset); }
};

// RGBA32F colour target (src/graphics/framebuffer.rs:10-118); row 0 = bottom like the GL image it replaces.
class Framebuffer {
public:
    Framebuffer(int width, int height) : width_(width), height_(height), rgba_(size_t(width) * height * 4, 0.0f) {}
    int width() const { return width_; }
    int height() const { return height_; }
    float* data() { return rgba_.data(); }
    const float* data() const { return rgba_.data(); }
    void clear(float r, float g, float b, float a) {
        for (size_t i = 0; i < rgba_.size(); i += 4) { rgba_[i] = r; rgba_[i + 1] = g; rgba_[i + 2] = b; rgba_[i + 3] = a; }
    }
    // as_image (framebuffer.rs:96-111): RGBA8 read-back, flipped vertically
    Image8 as_image() const {
        Image8 img;
        img.width = uint32_t(width_); img.height = uint32_t(height_);
        img.rgba.resize(rgba_.size());
        for (int y = 0; y < height_; ++y)
            for (int x = 0; x < width_ * 4; ++x) {
                float v = rgba_[size_t(height_ - 1 - y) * width_ * 4 + x];
                v = v != v ? 0.0f : (v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v));
                img.rgba[size_t(y) * width_ * 4 + x] = uint8_t(v * 255.0f + 0.5f);
            }
        return img;
    }

private:
    int width_, height_;
    std::vector<float> rgba_;
};

// framebuffer.rs:120-134: mean absolute RGB difference as a fraction of full scale
inline double diff_images(const Image8& a, const Image8& b) {
    if (a.width != b.width || a.height != b.height) return 1.0;
    uint64_t acc = 0;
    for (size_t i = 0; i < size_t(a.width) * a.height; ++i)
        for (int c = 0; c < 3; ++c) acc += uint64_t(std::abs(int(a.rgba[i * 4 + c]) - int(b.rgba[i * 4 + c])));
    return double(acc) / (255.0 * 3.0 * double(a.width) * double(a.height));
}

inline Vec3 normalize(Vec3 v) {
    const float m = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    return Vec3{v.x / m, v.y / m, v.z / m};
}
inline Vec3 cross(Vec3 a, Vec3 b) { return Vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// u_view = Matrix4::look_to_rh(eye, dir, up).invert() (svo.rs:197; cgmath 0.18.0, not vendored). look_to_rh builds the
// orthonormal rows s = normalize(f x up), u = s x f, -f with translation (-eye.s, -eye.u, eye.f); the inverse of that
// rigid transform has columns [s, u, -f, eye]. Column-major, as the uniform is uploaded.
inline void view_matrix(Vec3 eye, Vec3 dir, Vec3 up, float out[16]) {
    const Vec3 f = normalize(dir), s = normalize(cross(f, up)), u = cross(s, f);
    const float m[16] = {s.x, s.y, s.z, 0, u.x, u.y, u.z, 0, -f.x, -f.y, -f.z, 0, eye.x, eye.y, eye.z, 1};
    for (int i = 0; i < 16; ++i) out[i] = m[i];
}

class Svo {
public:
    // Svo::new (svo.rs:109-149)
    Svo(const VoxelRegistry& registry, SvoType type, size_t size_mb, int device = 0) {
        check(vx_create(int(type), size_mb * 1000 * 1000, device, &ctx_));
        std::string err;
        TextureArrayBuilder tex(6, 4.0f);
        if (!registry.build_texture_array(tex, err)) fail(err);
        if (tex.layers()) {
            const std::vector<uint8_t> base = tex.base_level();
            check(vx_set_textures(ctx_, base.data(), tex.width(), tex.height(), tex.layers(), tex.mip_levels()));
        }
        const std::vector<vx_material> mats = registry.build_material_buffer(tex);
        if (!mats.empty()) check(vx_set_materials(ctx_, mats.data(), uint32_t(mats.size())));
    }
    ~Svo() { vx_destroy(ctx_); }
    Svo(const Svo&) = delete;
    Svo& operator=(const Svo&) = delete;

    // Svo::update (svo.rs:171-189)
    void update(WorldSvoSource& svo) {
        const std::vector<Range> dirty = svo.updated_ranges();
        // dst_len: the room behind the writer's header, which is what the writers check their ranges against (esvo.rs:328)
        if (!svo.write_changes_to(vx_staging_ptr(ctx_) + 4, vx_arena_capacity(ctx_), true)) fail("dst is not large enough");
        std::vector<vx_range> ranges;
        for (const Range& r : dirty) ranges.push_back(vx_range{r.start, r.length});
        check(vx_commit(ctx_, svo.depth(), ranges.data(), uint32_t(ranges.size()), svo.size_in_bytes()));
    }

    Stats get_stats() const {
        vx_stats s;
        check(vx_get_stats(ctx_, &s));
        return Stats{size_t(s.used_bytes), size_t(s.capacity_bytes), uint8_t(s.depth)};
    }

    // Svo::render (svo.rs:196-229)
    void render(const RenderParams& p, Framebuffer& target) const {
        vx_uniforms u;
        view_matrix(p.cam_pos, p.cam_fwd, p.cam_up, u.view);
        u.fovy = p.fov_y_rad;
        u.aspect = p.aspect_ratio;
        u.ambient = p.ambient_intensity;
        u.light_dir[0] = p.light_dir.x; u.light_dir[1] = p.light_dir.y; u.light_dir[2] = p.light_dir.z;
        u.cam_pos[0] = p.cam_pos.x; u.cam_pos[1] = p.cam_pos.y; u.cam_pos[2] = p.cam_pos.z;
        u.render_shadows = p.render_shadows ? 1 : 0;
        u.shadow_distance = p.shadow_distance;
        const float nan = std::nanf("");
        const Vec3 sel = p.selected_voxel.value_or(Vec3{nan, nan, nan});  // svo.rs:211
        u.highlight_pos[0] = sel.x; u.highlight_pos[1] = sel.y; u.highlight_pos[2] = sel.z;
        vx_target t{target.data(), nullptr, VX_MEM_HOST, 0, 1, VX_FORMAT_RGBA32F};
        check(vx_render(ctx_, &u, uint32_t(target.width()), uint32_t(target.height()), &t));
    }

    // Svo::raycast (svo.rs:233-255)
    void raycast(const PickerBatch& batch, PickerBatchResult& result) const {
        std::vector<vx_picker_task> tasks;
        const size_t n = batch.serialize_tasks(tasks);
        std::vector<vx_picker_result> out(n);
        check(vx_raycast(ctx_, tasks.data(), uint32_t(n), out.data()));
        batch.deserialize_results(out.data(), result);
    }

    vx_context* handle() const { return ctx_; }

private:
    vx_context* ctx_ = nullptr;
    [[noreturn]] static void fail(const std::string& m) { throw std::runtime_error(m); }  // the reference panics (svo.rs:112,120,127)
    static void check(int rc) {
        if (rc != VX_OK) fail(std::string("libvoxelhip: ") + vx_last_error());
    }
};

}  // namespace graphics
}  // namespace vx
