// TEST HARNESS ONLY: compiles the DEVICE header (voxel-rs_amd/csrc/hip/vx_device.hpp) for the host with shims for the
// HIP built-ins, so the device-side logic can be stepped against the oracle without a GPU. Never linked into the
// product libraries; the product has no CPU path.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#define __device__
#define __host__
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
    std::vector<unsigned char> lds(Stack<1>::kBytes + 64);
    // Stack<1>::slot0 is "negative" (it is relative to scale 0 of a full-height plane): bias the base so that the sums land in lds
    vx_smem = lds.data();
    StackSpill spill;
    Stack<1> st;
    st.init(0, &spill);
    for (uint32_t i = 0; i < n; ++i) {
        Result res;
        uint32_t steps = 0;
        if (svo_type == 1) intersect<1, false, false, true>(sc, tasks[i].pos, tasks[i].dir, tasks[i].max_dst, cast_translucent != 0, st, res, steps, nullptr, nullptr);
        else intersect<2, false, false, true>(sc, tasks[i].pos, tasks[i].dir, tasks[i].max_dst, cast_translucent != 0, st, res, steps, nullptr, nullptr);
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

