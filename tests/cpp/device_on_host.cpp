// TEST HARNESS ONLY: compiles the DEVICE header (voxel-rs_amd/csrc/hip/vx_device.hpp) for the host with shims for the
// HIP built-ins, so the device-side logic can be stepped against the oracle without a GPU. Never linked into the
// product libraries; the product has no CPU path.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#define __device__
#define __forceinline__ inline
#define __constant__ static const
#define __restrict__
struct uint4 { uint32_t x, y, z, w; };
struct uint2 { uint32_t x, y; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline uint32_t __popc(uint32_t v) { return uint32_t(__builtin_popcount(v)); }
static inline int __clz(uint32_t v) { return v ? __builtin_clz(v) : 32; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline int32_t __float_as_int(float f) { int32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
#define HIP_INCLUDE_HIP_HIP_RUNTIME_H  // keep <hip/hip_runtime.h> out
#define VX_DEVICE_ON_HOST 1
#include "vx_device.hpp"

namespace vxd { unsigned char* vx_smem = nullptr; }
using namespace vxd;

extern "C" void devhost_picker(int svo_type, const uint8_t* world, uint64_t world_bytes, const vx_material* mats, uint32_t n_mats,
                               const uint8_t* tex, uint32_t tw, uint32_t th, uint32_t layers, uint32_t levels, const uint32_t* level_offset,
                               const vx_picker_task* tasks, uint32_t n, vx_picker_result* results, int cast_translucent) {
    SceneArgs sa = {};
    sa.world = world; sa.world_bytes = uint32_t(world_bytes); sa.materials = mats; sa.n_materials = n_mats;
    sa.tex = tex; sa.tex_bytes = 0;
    sa.width = tw; sa.height = th; sa.layers = layers; sa.levels = levels;
    for (uint32_t l = 0; l < levels && l < 16; ++l) {
        sa.level_offset[l] = level_offset[l];
        const uint32_t w = (tw >> l) ? (tw >> l) : 1, h = (th >> l) ? (th >> l) : 1;
        sa.tex_bytes = level_offset[l] + layers * w * h * 4;
    }
    const DevScene sc = make_scene(sa);
    std::vector<unsigned char> lds(7 * 10 + 64);
    vx_smem = lds.data();
    StackSpill spill;
    Stack st;
    st.stride = 1; st.tid = 0; st.levels = 7; st.spill = &spill;
    for (uint32_t i = 0; i < n; ++i) {
        Result res;
        uint32_t steps = 0, nf = 0;
        if (svo_type == 1) intersect<1, false, false>(sc, tasks[i].pos, tasks[i].dir, tasks[i].max_dst, cast_translucent != 0, st, res, steps, nullptr, 0, nf, nullptr);
        else intersect<2, false, false>(sc, tasks[i].pos, tasks[i].dir, tasks[i].max_dst, cast_translucent != 0, st, res, steps, nullptr, 0, nf, nullptr);
        vx_picker_result r;
        std::memset(&r, 0, sizeof r);
        if (res.t > 0.0f) {
            r.dst = res.t; r.inside_voxel = res.inside_voxel;
            std::memcpy(r.pos, res.pos, 12);
            std::memcpy(r.normal, kFaceNormals[res.face_id], 12);
        } else r.dst = -1.0f;
        results[i] = r;
    }
}

// Image traversal on the host: builds the 64-byte-octant image with the same per-entry rule as esvo_image_kernel and runs
// Trav<VX_SVO_IMAGE>, falling back to the reference-format traversal exactly like render_persistent does.
extern "C" void devhost_picker_image(const uint8_t* world, uint64_t world_bytes, const vx_material* mats, uint32_t n_mats, const uint8_t* tex,
                                     uint32_t tw, uint32_t th, uint32_t layers, const vx_picker_task* tasks, uint32_t n,
                                     vx_picker_result* results, int cast_translucent, uint32_t* n_fallbacks) {
    const uint32_t* arena = reinterpret_cast<const uint32_t*>(world + 24);
    const uint64_t n_oct = (world_bytes - 24) / 48;
    std::vector<uint2> image(n_oct * 8);
    for (uint64_t k = 0; k < n_oct; ++k)
        for (uint32_t j = 0; j < 8; ++j) {
            const uint32_t* o = arena + k * 12;
            const uint32_t masks = (o[j >> 1] >> ((j & 1u) * 16u)) & 0xffffu;
            uint32_t lo = o[4 + j];
            if (lo & 0x80000000u) lo = 0x80000000u | uint32_t((12ull * k + 4 + j + (lo & 0x7fffffffu)) / 12);
            image[k * 8 + j] = uint2{lo, masks};
        }
    uint32_t pre[5];
    std::memcpy(pre, world + 4, sizeof pre);
    SceneArgs sa = {};
    sa.world = world; sa.world_bytes = uint32_t(world_bytes); sa.materials = mats; sa.n_materials = n_mats;
    sa.tex = tex; sa.tex_bytes = layers * tw * th * 4; sa.width = tw; sa.height = th; sa.layers = layers; sa.levels = 1;
    sa.image = reinterpret_cast<const uint8_t*>(image.data()); sa.image_bytes = uint32_t(image.size() * 8);
    sa.image_root = (pre[4] - 5) / 12; sa.image_root_masks = pre[0] & 0xffffu;
    const DevScene sc = make_scene(sa);
    std::vector<unsigned char> lds(7 * 10 + 64);
    vx_smem = lds.data();
    StackSpill spill;
    Stack st;
    st.stride = 1; st.tid = 0; st.levels = 7; st.spill = &spill;
    *n_fallbacks = 0;
    for (uint32_t i = 0; i < n; ++i) {
        Result res;
        uint32_t nf = 0;
        Trav<VX_SVO_IMAGE> tr;
        tr.init(sc, tasks[i].pos, tasks[i].dir, tasks[i].max_dst);
        for (;;) {
            const TravStatus s = tr.step<false, false>(sc, st, nullptr, 0, nf, nullptr);
            if (s == kTravContinue) continue;
            if (s == kTravAtLeaf && tr.leaf_test<false>(sc, cast_translucent != 0, res, nullptr)) break;
            if (s == kTravFinished) { result_miss(res, tr.inside_voxel); break; }
            if (s == kTravNeedsReference) {
                tr.init(sc, tasks[i].pos, tasks[i].dir, tasks[i].max_dst, true);
                ++*n_fallbacks;
            }
        }
        vx_picker_result r;
        std::memset(&r, 0, sizeof r);
        if (res.t > 0.0f) {
            r.dst = res.t; r.inside_voxel = res.inside_voxel;
            std::memcpy(r.pos, res.pos, 12);
            std::memcpy(r.normal, kFaceNormals[res.face_id], 12);
        } else r.dst = -1.0f;
        results[i] = r;
    }
}
