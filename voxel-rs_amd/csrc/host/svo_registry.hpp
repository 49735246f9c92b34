// Block registry: materials by block id and the texture array they index (src/graphics/svo_registry.rs:20-165,
// src/graphics/texture_array.rs:43-153). Output is what vx_set_materials / vx_set_textures take.
#pragma once

#include <cstdint>
#include <optional>
#include <string>
#include <unordered_map>
#include <vector>

#include "png_io.hpp"
#include "voxel_hip.h"

namespace vx {

// svo_registry.rs:18-97 (builder style: Material().specular(..).all_sides("stone").with_normals())
struct Material {
    float specular_pow = 0.0f, specular_strength = 0.0f;
    std::optional<std::string> tex_top, tex_side, tex_bottom, tex_top_normal, tex_side_normal, tex_bottom_normal;

    Material& specular(float pow, float strength) { specular_pow = pow; specular_strength = strength; return *this; }
    Material& all_sides(const std::string& n) { return top(n).side(n).bottom(n); }
    Material& top(const std::string& n) { tex_top = n; return *this; }
    Material& side(const std::string& n) { tex_side = n; return *this; }
    Material& bottom(const std::string& n) { tex_bottom = n; return *this; }
    // normal maps are the side's texture name + "_normal" (svo_registry.rs:84-96)
    Material& with_normals() {
        if (tex_top) tex_top_normal = *tex_top + "_normal";
        if (tex_side) tex_side_normal = *tex_side + "_normal";
        if (tex_bottom) tex_bottom_normal = *tex_bottom + "_normal";
        return *this;
    }
};

// TextureArrayBuilder (texture_array.rs:43-153): named RGBA8 layers of equal size, flipped vertically on the way in.
class TextureArrayBuilder {
public:
    TextureArrayBuilder(uint8_t mip_levels, float max_anisotropy) : mip_levels_(mip_levels), max_anisotropy_(max_anisotropy) {}

    bool add_file(const std::string& name, const std::string& path, std::string& err) {
        Image8 img;
        if (!png_read(path, img, err)) return false;
        return add_rgba8(name, img.width, img.height, std::move(img.rgba), err);
    }

    // rows given top to bottom; stored bottom to top (flip_image_v, texture_array.rs:155-176)
    bool add_rgba8(const std::string& name, uint32_t w, uint32_t h, std::vector<uint8_t> bytes, std::string& err) {
        if (index_.count(name)) { err = "name '" + name + "' is already registered"; return false; }
        if (bytes.size() != size_t(w) * h * 4) { err = "bad pixel buffer size for '" + name + "'"; return false; }
        if (!layers_.empty() && (w != width_ || h != height_)) { err = "image does not match base dimensions"; return false; }
        for (uint32_t y = 0; y < h / 2; ++y)
            for (size_t i = 0; i < size_t(w) * 4; ++i) std::swap(bytes[size_t(y) * w * 4 + i], bytes[size_t(h - 1 - y) * w * 4 + i]);
        width_ = w; height_ = h;
        index_[name] = uint32_t(layers_.size());
        layers_.push_back(std::move(bytes));
        return true;
    }

    std::optional<uint32_t> lookup(const std::string& name) const {
        auto it = index_.find(name);
        if (it == index_.end()) return std::nullopt;
        return it->second;
    }

    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    uint32_t layers() const { return uint32_t(layers_.size()); }
    uint8_t mip_levels() const { return mip_levels_; }  // vx_set_textures clamps to ilog2(min(w,h)) like :108
    float max_anisotropy() const { return max_anisotropy_; }  // explicit-LOD sampling ignores anisotropy; kept for the record

    std::vector<uint8_t> base_level() const {
        std::vector<uint8_t> out;
        for (const auto& l : layers_) out.insert(out.end(), l.begin(), l.end());
        return out;
    }

private:
    uint8_t mip_levels_;
    float max_anisotropy_;
    uint32_t width_ = 0, height_ = 0;
    std::unordered_map<std::string, uint32_t> index_;
    std::vector<std::vector<uint8_t>> layers_;
};

class VoxelRegistry {
public:
    VoxelRegistry& add_texture(const std::string& name, const std::string& path) { textures_.push_back({name, path}); return *this; }
    VoxelRegistry& add_material(uint32_t block, const Material& m) { materials_.push_back({block, m}); return *this; }

    // build_texture_array (svo_registry.rs:122-133): 6 mip levels, anisotropy 4
    bool build_texture_array(TextureArrayBuilder& out, std::string& err) const {
        out = TextureArrayBuilder(6, 4.0f);
        for (const auto& t : textures_)
            if (!out.add_file(t.first, t.second, err)) return false;
        return true;
    }

    // build_material_buffer (svo_registry.rs:135-165): dense table up to the largest block id; unknown texture
    // names map to layer 0, unset ones to -1
    std::vector<vx_material> build_material_buffer(const TextureArrayBuilder& tex) const {
        uint32_t max_id = 0;
        for (const auto& e : materials_) max_id = e.first > max_id ? e.first : max_id;
        std::vector<vx_material> rows(materials_.empty() ? 0 : max_id + 1, vx_material{0, 0, 0, 0, 0, 0, 0, 0});
        auto lookup = [&](const std::optional<std::string>& n) -> int32_t { return n ? int32_t(tex.lookup(*n).value_or(0)) : -1; };
        for (const auto& e : materials_) {
            const Material& m = e.second;
            rows[e.first] = vx_material{m.specular_pow, m.specular_strength, lookup(m.tex_top), lookup(m.tex_side), lookup(m.tex_bottom),
                                        lookup(m.tex_top_normal), lookup(m.tex_side_normal), lookup(m.tex_bottom_normal)};
        }
        return rows;
    }

private:
    std::vector<std::pair<std::string, std::string>> textures_;
    std::vector<std::pair<uint32_t, Material>> materials_;
};

}  // namespace vx
