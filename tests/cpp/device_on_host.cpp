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
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct uint2 { uint32_t x, y; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
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
static uint32_t g_opaque_lo = 0, g_opaque_hi = 0;  // walk mode 3: the block ids whose textures are opaque throughout (RenderParams::opaque_*)
extern "C" void devhost_set_opaque(uint32_t lo, uint32_t hi) { g_opaque_lo = lo; g_opaque_hi = hi; }

// The sub-tile queue's arithmetic (vx_args.hpp), as the kernel and the runtime use it.
// queue_subtile: the numbers dispenser c hands out, k = 0, 1, ... until the first beyond the launch -- every sub-tile of the launch exactly once? Returns 1 / 0.
extern "C" int devhost_queue_covers(uint32_t total, uint32_t stripe) {
    std::vector<uint8_t> seen(total, 0);
    uint32_t n = 0;
    for (uint32_t c = 0; c < vxk::kQueues; ++c) {
        uint32_t last = 0;
        for (uint32_t k = 0;; ++k) {
            const uint32_t t = vxk::queue_subtile(k, c, stripe);
            // (a power-of-two stretch is handed to the kernel as its logarithm -- PersistentArgs::stripe_shift --: the same number without the division)
            if ((stripe & (stripe - 1u)) == 0u && vxk::queue_subtile(k, c, stripe, uint32_t(__builtin_ctz(stripe))) != t) return 0;
            if (k && t <= last) return 0;  // (a dispenser's numbers grow: the first beyond the launch means it is dry)
            last = t;
            if (t >= total) break;
            if (seen[t]++) return 0;
            ++n;
        }
    }
    return n == total ? 1 : 0;
}
// tile_place / tile_number (RenderParams::tile_numbering) as launch_render sets them up: a permutation of the launch's tiles and its inverse? Returns 1 / 0;
// `place_of` (n_local entries, may be null) receives the places in queue order.
extern "C" int devhost_tile_numbering(uint32_t tiles_x, uint32_t tiles_y, uint32_t tile_count, uint32_t n_local, int numbering, int strip, uint32_t* place_of) {
    vxd::RenderParams p = {};
    p.tiles_x = tiles_x; p.tiles_y = tiles_y; p.tile_count = tile_count; p.n_local_tiles = n_local;
    vxd::set_tile_numbering(p, numbering, strip);
    std::vector<uint8_t> seen(n_local, 0);
    for (uint32_t i = 0; i < n_local; ++i) {
        const uint32_t place = vxd::tile_place(p, i);
        if (place >= n_local || seen[place]++ || vxd::tile_number(p, place) != i) return 0;
        if (place_of) place_of[i] = place;
    }
    return 1;
}
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
            face_vector<0>(uint32_t(res.face_id), r.normal);
        } else r.dst = -1.0f;
        results[i] = r;
    }
}


// One ray on the traversal IMAGE of a world, driven the way a lane of render_persistent is driven (vx_api.hip): the fast stack in
// the loop, the hand-over to the full stack below the LDS-resident levels, leaf tests, and -- for the image of a CSVO world --
// the excursion onto the world's own bytes when the ray is led into a voxel. Returns the reference's OctreeResult and the
// number of loop iterations.
// `along` (optional): the ray is the shadow ray of a primary that has just hit a voxel on THIS stack -- {un-mirrored corner of the voxel (3 floats as bits), the
// parent's octant, masks, scale}: the cursor takes the levels down to the voxel's parent in one go (Trav::descend_along), as render_persistent's set-up does;
// *along_taken says whether it took any level that way. `leaf_state` (optional): where the cursor stood when the ray hit, in the same form, for the next ray.
template <int IMG, int FOREIGN, bool SHALLOW, int LV = kLdsLevels>
static void image_cast(const DevScene& sc, const DevScene& sc_bytes, const float pos[3], const float dir[3], float max_dst, bool cast_translucent,
                       int walk_mode, vx_result* out, uint32_t* steps, uint32_t* given_up = nullptr, const uint32_t* along = nullptr, bool* along_taken = nullptr,
                       uint32_t* leaf_state = nullptr) {
    StackSpill spill;
    typedef Stack<1, false, false, LV, (LV > kLdsLevels)> FullStack;  // (16 levels: the 16-bit third plane, like the kernel's)
    typedef Stack<1, true, SHALLOW, LV, (LV > kLdsLevels)> FastStack;
    FullStack st;
    FastStack fast_st;
    st.init(0, &spill);
    fast_st.init(0, &spill);
    constexpr int kFastFloor = FullStack::kBaseScale - 1;
    Trav<IMG> tr;
    tr.init(sc, pos, dir, max_dst);
    bool walked = false;  // the hit (if any) was found by a walk inside a voxel: the cursor does not stand at a voxel of the image
    if (along) {
        float q[3];
        std::memcpy(q, along, 12);
        // (the primary's cursor stood in the voxel's parent when it hit: the parent joins its ancestors on the stack, as render_persistent does at the hit)
        fast_st.push(int(along[5]), along[3], 0.0f, along[4]);
        tr.descend_along(fast_st, int(along[5]), q);
        if (along_taken) *along_taken = tr.iter > 0;  // (levels taken in one go)
    }
    Result res;
    bool hit = false;
    TravStatus s = kTravContinue;
    for (;;) {
        if (s == kTravContinue) {
            if (SHALLOW || tr.scale >= kFastFloor) {
                s = tr.template step<false, false, true, FastStack, true, FOREIGN != 0>(sc, fast_st, nullptr, nullptr);
                if (s == kTravDeep || s == kTravForeign) --tr.iter;  // repeated by whoever takes over
                if (s != kTravContinue) tr.sync_idx();
            } else {
                tr.sync_idx();
                s = tr.template step<false, false, true, FullStack, true, FOREIGN != 0>(sc, st, nullptr, nullptr);
                if (s == kTravForeign) --tr.iter;
            }
            continue;
        }
        if (s == kTravDeep) {
            s = tr.template step<false, false, true, FullStack, true, FOREIGN != 0>(sc, st, nullptr, nullptr);
            if (s == kTravForeign) --tr.iter;
            continue;
        }
        if (s == kTravForeign) {
            {
                // the lean walk (walk_voxel_on_bytes): what it gives up on is run whole on the world's own bytes, like the render kernel does
                // (3: the image-only render kernels' build -- phantom leaves of opaque blocks are hits without their sample, whose colour
                // is sampled when the hit is shaded; any other phantom leaf is given up)
                bool color_pending = false;
                if (walk_mode == 3) {
                    s = walk_voxel_on_bytes<IMG, FullStack, true, true, false>(sc, sc_bytes.world, tr, st, cast_translucent, res, g_opaque_lo, g_opaque_hi, &color_pending);
                    if (s == kTravAtLeaf && color_pending) {
                        const vx_material mat = material_at(sc, res.value);
                        int tex_id = mat.tex_side;
                        if (res.face_id == 3) tex_id = mat.tex_top;
                        else if (res.face_id == 2) tex_id = mat.tex_bottom;
                        texture_lod(sc.tex, res.uv[0], res.uv[1], float(tex_id), res.lod, res.color);
                    }
                } else {
                    s = walk_voxel_on_bytes<IMG, FullStack, true, false, true>(sc, sc_bytes.world, tr, st, cast_translucent, res);
                }
                walked = true;
                if (given_up && s == kTravForeign) ++*given_up;
                if (s == kTravForeign) {
                    uint32_t n = 0;
                    Stack<1, false> st2;
                    st2.init(0, &spill);
                    intersect<VX_SVO_CSVO, false, false, true>(sc_bytes, pos, dir, max_dst, cast_translucent, st2, res, n, nullptr, nullptr);
                    out->t = res.t; out->value = res.value; out->face_id = res.face_id;
                    std::memcpy(out->pos, res.pos, 12); std::memcpy(out->uv, res.uv, 8); std::memcpy(out->color, res.color, 16);
                    out->lod = res.lod; out->inside_voxel = res.inside_voxel ? 1 : 0;
                    *steps = n;
                    return;
                }
            }
            if (s == kTravAtLeaf) { hit = true; break; }
            continue;
        }
        if (s == kTravAtLeaf) {
            const LeafOutcome o = tr.template leaf_test<false, false>(sc, st, cast_translucent, res, nullptr, nullptr);
            if (o == kLeafHit) { hit = true; break; }
            s = o == kLeafPassed ? kTravContinue : kTravFinished;
            continue;
        }
        break;  // kTravFinished
    }
    if (!hit) result_miss(res, tr.inside_voxel());
    out->t = res.t; out->value = res.value; out->face_id = res.face_id;
    std::memcpy(out->pos, res.pos, 12); std::memcpy(out->uv, res.uv, 8); std::memcpy(out->color, res.color, 16);
    out->lod = res.lod; out->inside_voxel = res.inside_voxel ? 1 : 0;
    *steps = tr.iter;
    if (leaf_state) {
        leaf_state[6] = hit && !walked ? 1u : 0u;
        float q[3];
        tr.cell_corner(q);
        std::memcpy(leaf_state, q, 12);
        leaf_state[3] = tr.ptr; leaf_state[4] = tr.node; leaf_state[5] = uint32_t(tr.scale);
    }
}

// layout: 1 = byte-offset image (VX_SVO_IMAGE), 2 = wide image (VX_SVO_IMAGE_WIDE); svo_type = the world's own format
static uint32_t g_given_up = 0;
extern "C" uint32_t devhost_given_up(void) { const uint32_t n = g_given_up; g_given_up = 0; return n; }

// walk_mode (images of CSVO worlds, a ray led into a voxel: walk_voxel_on_bytes; what it gives up on is run whole on the world's bytes):
// 2 = with the full leaf test inside the voxel, 3 = the image-only render kernels' build
extern "C" void devhost_image_cast(int svo_type, int layout, int shallow, int walk_mode, const uint8_t* world, uint64_t world_bytes, const uint8_t* image, uint64_t image_bytes,
                                   const uint8_t* origin, const vx_material* mats, uint32_t n_mats, const uint8_t* tex, uint32_t tw, uint32_t th,
                                   uint32_t layers, uint32_t levels, const uint32_t* level_offset, const vx_picker_task* tasks, uint32_t n,
                                   int cast_translucent, vx_result* results, uint32_t* steps) {
    SceneArgs sa = {};
    sa.world = world; sa.world_bytes = world_bytes; sa.materials = mats; sa.n_materials = n_mats;
    sa.tex = tex; sa.tex_bytes = 0;
    sa.width = tw; sa.height = th; sa.layers = layers; sa.levels = levels;
    for (uint32_t l = 0; l < levels && l < 16; ++l) {
        sa.level_offset[l] = level_offset[l];
        const uint32_t w = (tw >> l) ? (tw >> l) : 1, h = (th >> l) ? (th >> l) : 1;
        sa.tex_bytes = level_offset[l] + layers * w * h * 4;
    }
    sa.image = image; sa.image_bytes = image_bytes; sa.origin = origin;
    const DevScene sc = make_image_scene(sa), sc_bytes = make_scene(sa);
    std::vector<unsigned char> lds(Stack<1, false, false, 16, false>::kBytes + 64);
    vx_smem = lds.data();
    for (uint32_t i = 0; i < n; ++i) {
        const bool ct = cast_translucent != 0;
#define CAST(IMG, FOREIGN, SHALLOW) image_cast<IMG, FOREIGN, SHALLOW>(sc, sc_bytes, tasks[i].pos, tasks[i].dir, tasks[i].max_dst, ct, walk_mode, &results[i], &steps[i], &g_given_up)
#define CAST16(IMG, FOREIGN) image_cast<IMG, FOREIGN, true, 16>(sc, sc_bytes, tasks[i].pos, tasks[i].dir, tasks[i].max_dst, ct, walk_mode, &results[i], &steps[i], &g_given_up)
        if (shallow == 2) {  // the kernel build with 16 resident stack levels
            if (svo_type == 1) { if (layout == 1) CAST16(VX_SVO_IMAGE, 0); else CAST16(VX_SVO_IMAGE_WIDE, 0); }
            else { if (layout == 1) CAST16(VX_SVO_IMAGE, VX_SVO_CSVO); else CAST16(VX_SVO_IMAGE_WIDE, VX_SVO_CSVO); }
            continue;
        }
        if (svo_type == 1) {
            if (layout == 1) { if (shallow) CAST(VX_SVO_IMAGE, 0, true); else CAST(VX_SVO_IMAGE, 0, false); }
            else { if (shallow) CAST(VX_SVO_IMAGE_WIDE, 0, true); else CAST(VX_SVO_IMAGE_WIDE, 0, false); }
        } else {
            if (layout == 1) { if (shallow) CAST(VX_SVO_IMAGE, VX_SVO_CSVO, true); else CAST(VX_SVO_IMAGE, VX_SVO_CSVO, false); }
            else { if (shallow) CAST(VX_SVO_IMAGE_WIDE, VX_SVO_CSVO, true); else CAST(VX_SVO_IMAGE_WIDE, VX_SVO_CSVO, false); }
        }
#undef CAST
#undef CAST16
    }
}


// A pixel's two rays on the image of a world, the shadow ray both ways: from the root like any ray, and down the primary's path in one go (Trav::descend_along,
// what render_persistent's ray set-up does). For every task: the primary ray; if it hits, the shadow ray from hit + normal * 0.001 (world.glsl:79-84) toward
// `to_light` -- results[2 i] = the plain one, results[2 i + 1] = the one that started on the path, steps likewise. Returns how many shadow rays took at least one level
// along the path. Byte-offset images, the stack build with 16 resident levels; walk_mode 3.
extern "C" uint32_t devhost_image_shadow_pairs(int svo_type, const uint8_t* world, uint64_t world_bytes, const uint8_t* image, uint64_t image_bytes, const uint8_t* origin,
                                               const vx_material* mats, uint32_t n_mats, const uint8_t* tex, uint32_t tw, uint32_t th, uint32_t layers, uint32_t levels,
                                               const uint32_t* level_offset, const vx_picker_task* tasks, uint32_t n, const float* to_light, vx_result* results, uint32_t* steps) {
    SceneArgs sa = {};
    sa.world = world; sa.world_bytes = world_bytes; sa.materials = mats; sa.n_materials = n_mats;
    sa.tex = tex; sa.tex_bytes = 0;
    sa.width = tw; sa.height = th; sa.layers = layers; sa.levels = levels;
    for (uint32_t l = 0; l < levels && l < 16; ++l) {
        sa.level_offset[l] = level_offset[l];
        const uint32_t w = (tw >> l) ? (tw >> l) : 1, h = (th >> l) ? (th >> l) : 1;
        sa.tex_bytes = level_offset[l] + layers * w * h * 4;
    }
    sa.image = image; sa.image_bytes = image_bytes; sa.origin = origin;
    const DevScene sc = make_image_scene(sa), sc_bytes = make_scene(sa);
    std::vector<unsigned char> lds(Stack<1, false, false, 16, false>::kBytes + 64);
    vx_smem = lds.data();
    uint32_t taken = 0;
    for (uint32_t i = 0; i < n; ++i) {
        vx_result primary;
        uint32_t primary_steps = 0, leaf[7] = {};
        std::memset(&results[2 * i], 0, 2 * sizeof(vx_result));
        steps[2 * i] = steps[2 * i + 1] = 0;
        if (svo_type == 1) image_cast<VX_SVO_IMAGE, 0, true, 16>(sc, sc_bytes, tasks[i].pos, tasks[i].dir, tasks[i].max_dst, true, 3, &primary, &primary_steps, nullptr, nullptr, nullptr, leaf);
        else image_cast<VX_SVO_IMAGE, VX_SVO_CSVO, true, 16>(sc, sc_bytes, tasks[i].pos, tasks[i].dir, tasks[i].max_dst, true, 3, &primary, &primary_steps, nullptr, nullptr, nullptr, leaf);
        if (!(primary.t >= 0.0f) || !leaf[6]) continue;
        float normal[3];
        face_vector<0>(uint32_t(primary.face_id), normal);
        const float so[3] = {primary.pos[0] + normal[0] * 0.001f, primary.pos[1] + normal[1] * 0.001f, primary.pos[2] + normal[2] * 0.001f};
        // the path first: the primary's stack is still in place
        bool ok = false;
        if (svo_type == 1) image_cast<VX_SVO_IMAGE, 0, true, 16>(sc, sc_bytes, so, to_light, -1.0f, true, 3, &results[2 * i + 1], &steps[2 * i + 1], nullptr, leaf, &ok);
        else image_cast<VX_SVO_IMAGE, VX_SVO_CSVO, true, 16>(sc, sc_bytes, so, to_light, -1.0f, true, 3, &results[2 * i + 1], &steps[2 * i + 1], nullptr, leaf, &ok);
        taken += ok ? 1u : 0u;
        if (svo_type == 1) image_cast<VX_SVO_IMAGE, 0, true, 16>(sc, sc_bytes, so, to_light, -1.0f, true, 3, &results[2 * i], &steps[2 * i]);
        else image_cast<VX_SVO_IMAGE, VX_SVO_CSVO, true, 16>(sc, sc_bytes, so, to_light, -1.0f, true, 3, &results[2 * i], &steps[2 * i]);
    }
    return taken;
}
