// Device-side ray path for gfx950: SVO traversal (ESVO and CSVO node formats), software texture sampler,
// shading, sky. Written for wave64 / LDS-resident per-ray stacks; no MFMA (nothing here is a contraction).
//
// What it computes is the reference's GLSL (assets/shaders/svo.esvo.glsl:50-393, svo.csvo.glsl:151-509,
// world.glsl:27-141, picker.glsl:30-51); how it computes it is not a translation: per-ray stacks live in LDS
// in a [level][thread] layout (bank = lane, conflict-free at any mix of levels), the ESVO child descriptor is
// kept in a register between PUSH/POP instead of being re-fetched every iteration, CSVO bytes are read with
// single unaligned dword loads, divisions by powers of two are multiplications.
//
// Numerics contract (checked bit for bit against oracle/svo_oracle.c): fp32 only, no implicit contraction
// (-ffp-contract=off); the plane-distance expressions use explicit fmaf exactly where the oracle does.
#pragma once

#include <stdint.h>

#include <vx_platform.hpp>  // the gfx950 primitives this file is written in (buffer loads, LDS address spaces, v_min3 ...)

#include "voxel_hip.h"
#include "vx_args.hpp"

// 1 = the render loop's step on a traversal image is Trav::step_image (PUSH and ADVANCE merged into one instruction stream); 0 = step_with's
// own paths (the round-2 loop, kept for A/B builds)
#ifndef VX_MERGED_STEP
#define VX_MERGED_STEP 1
#endif

namespace vxd {

constexpr float kEps = 1.1920929e-7f;  // exp2(-23), svo.esvo.glsl:24
constexpr uint32_t kInvalidPtr = 0xffffffffu;

struct DevTextures {
    buf_t buf;                // mip chain, level l at level_offset[l], layout [layer][y][x][4]
    uint32_t width, height, layers, levels;
    const uint32_t* level_offset;  // [16], stays in the kernel-argument segment (no private copy: it is indexed per lane)
    // The kernel vouches that `height` is a power of two (a literal `true` in kernels that are only launched for such textures: the sampler's
    // REPEAT is then a mask and nothing of the general wrap is compiled in); false: the sampler looks at `height`.
    bool pow2_height;
};

struct DevScene {
    buf_t world;              // device copy of the mapped world buffer, byte 0 = f32 octree_scale (the first 4 GiB of it)
    buf_t records8;           // traversal images: the same memory as 8-byte records addressed by index (what the hand-scheduled loop loads entries through)
    buf_t materials;          // vx_material rows
    float octree_scale;       // = world[0]
    uint32_t root_ptr;        // CSVO: world[1]
    uint32_t image_root_masks, image_root_octant;  // traversal images: the header's two words every ray starts from (read once, by make_image_scene)
    DevTextures tex;
    // buffers that outgrow a V#'s 32-bit offsets are read through a plain 64-bit pointer: the traversal image in its wide layout
    // (VX_SVO_IMAGE_WIDE), and an ESVO world of 4 GiB and more (VX_SVO_ESVO_BIG; descriptors[] indices stay 32 bits: 16 GiB)
    const uint8_t* wide;
    uint64_t wide_bytes;
    // CSVO worlds only: the image again, as a plain pointer -- where a voxel-parent octant comes from in the world's own bytes is the unit in front of its
    // values (traversal_image.hpp), read when a ray is led INTO a voxel (walk_voxel_on_bytes)
    const uint8_t* origin;
};

__device__ __forceinline__ uint32_t clamp_u32(uint64_t v) { return v < 0xffffffffull ? uint32_t(v) : 0xffffffffu; }

__device__ __forceinline__ DevScene make_scene(const SceneArgs& a) {
    DevScene sc;
    sc.world = make_buf(a.world, clamp_u32(a.world_bytes));
    sc.records8 = sc.world;  // (unused on the world's own bytes)
    sc.materials = make_buf(a.materials, a.n_materials * uint32_t(sizeof(vx_material)));
    sc.tex.buf = make_buf(a.tex, a.tex_bytes);
    sc.tex.width = a.width; sc.tex.height = a.height; sc.tex.layers = a.layers; sc.tex.levels = a.levels;
    sc.tex.level_offset = a.level_offset;
    sc.tex.pow2_height = false;
    sc.octree_scale = __uint_as_float(buf_u32(sc.world, 0));
    sc.root_ptr = buf_u32(sc.world, 4);
    sc.wide = a.world;
    sc.wide_bytes = a.world_bytes;
    sc.origin = nullptr;
    sc.image_root_masks = sc.image_root_octant = 0;
    return sc;
}

// the same scene with the traversal image in place of the world buffer
__device__ __forceinline__ DevScene make_image_scene(const SceneArgs& a) {
    DevScene sc;
    sc.world = make_buf(a.image, clamp_u32(a.image_bytes));  // (a wide image is not read through this)
    sc.records8 = make_buf_records8(a.image, clamp_u32(a.image_bytes));
    sc.materials = make_buf(a.materials, a.n_materials * uint32_t(sizeof(vx_material)));
    sc.tex.buf = make_buf(a.tex, a.tex_bytes);
    sc.tex.width = a.width; sc.tex.height = a.height; sc.tex.layers = a.layers; sc.tex.levels = a.levels;
    sc.tex.level_offset = a.level_offset;
    sc.tex.pow2_height = false;
    sc.octree_scale = __uint_as_float(buf_u32(sc.world, 0));
    sc.root_ptr = 0;
    sc.wide = a.image;
    sc.wide_bytes = a.image_bytes;
    sc.origin = a.origin;
    // header of the image: root masks, root octant (byte offset / index). (A wide image's header is within the first 4 GiB: the resource reaches it.)
    sc.image_root_masks = buf_u32(sc.world, 4);
    sc.image_root_octant = buf_u32(sc.world, 8);
    return sc;
}

// Traversal images (traversal_image.hpp): everything is addressed in 8-byte units from the start of the image. A node's octant holds an entry per EXISTING
// child, child 7 first; `lo` = the unit before the first. With m = the node's masks shifted left by the child's index ("exists" of that child in bit 23, of
// the children above it below, "is a leaf" in the sign), the child's entry is unit lo + popcount(m & 0xffffff). An octant of values only (all children leaves) holds 4-byte values in the
// same order from unit lo + 1 on. Byte-offset layout: through the buffer resource (range-checked: any `lo` is harmless); wide layout (beyond 4 GiB): a
// 64-bit pointer, so `lo` has to stay inside the image.
template <bool WIDE>
__device__ __forceinline__ uint2 image_entry(const DevScene& sc, uint32_t lo, uint32_t m) {
    const uint32_t unit = lo + uint32_t(__popc(m & 0x00ffffffu));
    return WIDE ? mem_u64(sc.wide + (uint64_t(unit) << 3)) : buf_u64(sc.world, unit << 3);
}
template <bool WIDE>
__device__ __forceinline__ uint32_t image_u32(const DevScene& sc, uint32_t unit, uint32_t byte) {
    return WIDE ? mem_u32(sc.wide + (uint64_t(unit) << 3) + byte) : buf_u32(sc.world, (unit << 3) + byte);
}

struct Result {
    float t;
    uint32_t value;
    int face_id;
    float pos[3];
    float uv[2];
    float color[4];
    float lod;
    bool inside_voxel;
};

struct Counters {  // per-thread, reduced by the instrumented kernel
    uint32_t rays, iterations, pushes, leaf_tests, leaf_tests_trilinear, boundaries, csvo_header_bytes, csvo_pointer_bytes;
};

// Per-ray traversal stack. The levels an octree of depth <= LEVELS - 1 can legitimately reach live in LDS as three planes
// [plane][level][thread]: every lane's slot for a level sits in its own bank whatever mix of levels the lanes are on, and the
// planes are a compile-time distance apart, so one address register serves the accesses (the first two fuse into
// ds_write2st64_b32 / ds_read2st64_b32). A ray that starts INSIDE a voxel makes the reference descend "below" the leaves,
// interpreting leaf bytes as nodes (svo.esvo.glsl:183-185 / svo.csvo.glsl:293-295 only treat a leaf as a hit when t_min > 0);
// its stack arrays hold MAX_SCALE + 1 entries (svo.esvo.glsl:28-30), so those pushes are kept too -- in a per-thread spill
// array that ordinary rays never touch.
constexpr int kLdsBaseScale = kMaxScale - kLdsLevels;   // scales [kLdsBaseScale, 22] are LDS resident

// per-thread backing store for the levels below the LDS-resident ones (lives in scratch)
struct StackSpill {
    uint32_t ptr[kMaxScale];
    float t_max[kMaxScale];
    uint32_t aux[kMaxScale];
};

// What a slot holds -- ESVO: {own-octant pointer, t_max, child masks}; CSVO: {node byte pointer, t_max, depth << 16 | header};
// image: {octant offset, t_max, masks (the upper half of a word)}.
// FAST = the caller guarantees LDS-resident scales only (see Trav::step: a ray that is about to leave them reports
// kTravDeep instead), so push/pop are the bare LDS accesses.
// BOUNDED (fast stacks): the caller also guarantees that no PUSH goes below them -- an octree of at most LEVELS levels whose
// rays cannot be led beyond its leaves (a validated traversal image) -- so step() does not even look.
// AUX16 (image cursors only: their third word has its lower half free): the third plane holds 16 bits per slot. Ten bytes per
// level and lane instead of twelve: 16 levels in the 10 KB a wave can have at 16 waves per CU (LEVELS = 16: worlds of up to 16
// levels without the deep-push hand-over), at one more address computation per push and pop.
// HOT (experiment X1, "hot upper octree levels staged in LDS"; image cursors, one wave per block): behind the planes the block keeps
// a copy of the image's root octant and of the root's eight child octants (9 x 64 bytes, Stack::load_hot); a PUSH out of one of
// those nodes takes its entry from there instead of from memory (Trav::step_with).
template <int THREADS, bool FAST = false, bool BOUNDED = false, int LEVELS = kLdsLevels, bool AUX16 = false, bool HOT = false>
struct Stack {
    static constexpr bool kFast = FAST;
    static constexpr bool kHot = HOT;
    static constexpr bool kCanOverflow = FAST && !BOUNDED;
    static constexpr bool kAux16 = AUX16;
    static constexpr int kLevels = LEVELS;
    static constexpr int kBaseScale = kMaxScale - LEVELS;                   // scales [kBaseScale, 22] are LDS resident
    static constexpr uint32_t kPlane = uint32_t(LEVELS) * THREADS * 4;      // bytes between the first two planes
    static constexpr uint32_t kStackBytes = 2 * kPlane + (AUX16 ? kPlane / 2 : kPlane);
    static constexpr uint32_t kHotBytes = HOT ? 9u * 64u : 0u;
    static constexpr uint32_t kBytes = kStackBytes + kHotBytes;  // dynamic LDS a block of THREADS threads needs
    uint32_t slot0;  // byte offset of this thread's slot for scale 0 of a (virtual) full-height plane; may be "negative"
    VX_AS_PRIVATE StackSpill* spill;

    __device__ __forceinline__ void init(uint32_t tid, StackSpill* sp) {
        slot0 = tid * 4u - uint32_t(kBaseScale) * THREADS * 4u;
        spill = (VX_AS_PRIVATE StackSpill*)sp;
    }
    __device__ __forceinline__ VX_AS_LDS uint32_t* at(uint32_t byte) const { return (VX_AS_LDS uint32_t*)((VX_AS_LDS unsigned char*)vx_smem + byte); }
    __device__ __forceinline__ VX_AS_LDS uint16_t* at16(uint32_t byte) const { return (VX_AS_LDS uint16_t*)((VX_AS_LDS unsigned char*)vx_smem + byte); }
    // HOT: entry `child` (0..7) of hot node `n` (0 = the root, 1 + c = the root's child c)
    __device__ __forceinline__ uint2 hot_entry(uint32_t n, uint32_t child) const {
        const VX_AS_LDS uint32_t* e = at(kStackBytes + n * 64u + child * 8u);
        return make_uint2(e[0], e[1]);
    }
    // HOT: fills the copy (one wave; `lane` 0..63), every octant EXPANDED to eight entries (zeros where a child does not exist) so that hot_entry() needs no
    // popcount. The root's entries first, then entry (lane & 7) of the root's child (lane >> 3) -- whatever that child is: only octants of inner nodes are
    // ever read back.
    template <class SCENE>
    __device__ __forceinline__ void load_hot(const SCENE& sc, uint32_t lane) const {
        const uint32_t root_masks = sc.image_root_masks, root_lo = sc.image_root_octant;
        auto entry_of = [&](uint32_t lo, uint32_t masks, uint32_t child) -> uint2 {
            const uint32_t m = masks << child;
            if (!(m & 0x00800000u)) return make_uint2(0u, 0u);
            return buf_u64(sc.world, (lo + uint32_t(__popc(m & 0x00ffffffu))) << 3);
        };
        if (lane < 8) {
            const uint2 e = entry_of(root_lo, root_masks, lane);
            VX_AS_LDS uint32_t* d = at(kStackBytes + lane * 8u);
            d[0] = e.x; d[1] = e.y;
        }
        const uint2 parent = entry_of(root_lo, root_masks, lane >> 3);  // (the root's child: its `lo` and masks; a leaf's value is no octant: nothing of it is read back)
        const bool inner = ((root_masks << (lane >> 3)) & 0x80800000u) == 0x00800000u;
        const uint2 e = inner ? entry_of(parent.x, parent.y, lane & 7u) : make_uint2(0u, 0u);
        VX_AS_LDS uint32_t* d = at(kStackBytes + 64u + lane * 8u);
        d[0] = e.x; d[1] = e.y;
    }
    __device__ __forceinline__ bool resident(int scale) const { return FAST || (scale >= kBaseScale && scale < kMaxScale); }

    __device__ __forceinline__ void push(int scale, uint32_t p, float t, uint32_t a) const {
        if (resident(scale)) {
            const uint32_t s = uint32_t(scale) * (THREADS * 4u) + slot0;
            *at(s) = p; *at(s + kPlane) = __float_as_uint(t);
            if (AUX16) *at16(2 * kPlane + (s >> 1)) = uint16_t(a >> 16);
            else *at(s + 2 * kPlane) = a;
        } else if (uint32_t(scale) < uint32_t(kMaxScale)) {
            spill->ptr[scale] = p; spill->t_max[scale] = t; spill->aux[scale] = a;
        }
    }
    // the t_max of a resident slot alone: pointer and masks stay what they are (Trav::descend_along)
    __device__ __forceinline__ void set_t_max(int scale, float t) const {
        const uint32_t s = uint32_t(scale) * (THREADS * 4u) + slot0;
        *at(s + kPlane) = __float_as_uint(t);
    }
    // scale is in [0, kMaxScale) here (the caller has already left the octree otherwise)
    __device__ __forceinline__ void pop(int scale, uint32_t& p, float& t, uint32_t& a) const {
        if (resident(scale)) {
            const uint32_t s = uint32_t(scale) * (THREADS * 4u) + slot0;
            p = *at(s); t = __uint_as_float(*at(s + kPlane));
            if (AUX16) a = uint32_t(*at16(2 * kPlane + (s >> 1))) << 16;
            else a = *at(s + 2 * kPlane);
        } else {
            p = spill->ptr[scale]; t = spill->t_max[scale]; a = spill->aux[scale];
        }
    }
};

__device__ __forceinline__ float gmin(float x, float y) { return y < x ? y : x; }  // GLSL min
__device__ __forceinline__ float gmax(float x, float y) { return x < y ? y : x; }  // GLSL max
__device__ __forceinline__ float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
__device__ __forceinline__ uint32_t low_bits(int n) { return n >= 32 ? 0xffffffffu : (n <= 0 ? 0u : ((1u << n) - 1u)); }
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float(uint32_t(e + 127) << 23); }
__device__ __forceinline__ float smoothstepf(float e0, float e1, float x) {
    const float t = gclamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}

// ---- world buffer access --------------------------------------------------------------------------------

// ESVO: descriptors[] starts at byte 4 (svo.esvo.glsl:3-6)
__device__ __forceinline__ uint32_t esvo_word(const DevScene& sc, uint32_t index) {
    index = index < 0x3ffffffeu ? index : 0x3ffffffeu;  // keep 4 + 4*index from wrapping: wild indices must read 0, not alias
    return buf_u32(sc.world, 4u + index * 4u);
}

// CSVO: descriptors[] starts at byte 8 and is addressed in bytes (svo.csvo.glsl:1-5, 25-49); the hardware reads
// unaligned dwords directly, so read_uint's two-loads-and-shift collapses into one load
__device__ __forceinline__ uint32_t csvo_clamp(uint32_t byte_ptr) { return byte_ptr < 0xfffffff0u ? byte_ptr : 0xfffffff0u; }  // no wrap of 8 + ptr
__device__ __forceinline__ uint32_t csvo_u32(const DevScene& sc, uint32_t byte_ptr) { return buf_u32(sc.world, 8u + csvo_clamp(byte_ptr)); }
__device__ __forceinline__ uint32_t csvo_u16(const DevScene& sc, uint32_t p) { return csvo_u32(sc, p) & 0xffffu; }
__device__ __forceinline__ uint32_t csvo_u8(const DevScene& sc, uint32_t byte_ptr) { return buf_u8(sc.world, 8u + csvo_clamp(byte_ptr)); }

// bytes taken by the pointer-table entries a 2-bit-per-child mask selects: tag 0,1,2,3 -> 0,1,2,4 bytes
__device__ __forceinline__ uint32_t csvo_tag_bytes(uint32_t m) {
    return __popc(m & 0x5555u) + 2u * __popc(m & 0xAAAAu) + __popc(m & (m >> 1) & 0x5555u);
}

// read_next_ptr (svo.csvo.glsl:53-116)
__device__ __forceinline__ uint32_t csvo_next_ptr(const DevScene& sc, uint32_t ptr, uint32_t depth, uint32_t idx, bool& crossed,
                                                  uint32_t& header_bytes, uint32_t& pointer_bytes) {
    crossed = false;
    if (depth > 3) {
        header_bytes = 2;
        const uint32_t header = csvo_u16(sc, ptr);
        const uint32_t child = (header >> (idx * 2)) & 3u;
        if (child == 0) return kInvalidPtr;
        const uint32_t offset = csvo_tag_bytes(header & ((1u << (idx * 2)) - 1u));
        const uint32_t ptr_bytes = csvo_tag_bytes(header);
        uint32_t ptr_offset = csvo_u32(sc, ptr + 2 + offset);
        ptr_offset &= low_bits(int(1u << (child - 1)) * 8);
        pointer_bytes = (1u << child) >> 1;
        if (ptr_offset & (1u << 31)) {
            crossed = true;
            return ptr_offset ^ (1u << 31);
        }
        return ptr + 2 + ptr_bytes + ptr_offset;
    }
    header_bytes = 1;
    const uint32_t header = csvo_u8(sc, ptr);
    if (((header >> idx) & 1u) == 0) return kInvalidPtr;
    const uint32_t offset = __popc(header & ((1u << idx) - 1u));
    if (depth == 3) {
        pointer_bytes = 1;
        return ptr + 1 + __popc(header) + csvo_u8(sc, ptr + 1 + offset);
    }
    pointer_bytes = 0;
    return ptr + 3 + offset;
}

// read_leaf (svo.csvo.glsl:119-133)
__device__ __forceinline__ uint32_t csvo_read_leaf(const DevScene& sc, uint32_t material_section_ptr, uint32_t pre_leaf_ptr, uint32_t ptr,
                                                   uint32_t idx) {
    const uint32_t material_section_offset = csvo_u16(sc, pre_leaf_ptr + 1);
    const int leaf_index = int(ptr - (pre_leaf_ptr + 3));
    const int bit_mark = leaf_index * 8 + int(idx);
    const uint32_t v0 = csvo_u32(sc, pre_leaf_ptr + 3) & low_bits(bit_mark < 32 ? bit_mark : 32);
    const uint32_t v1 = csvo_u32(sc, pre_leaf_ptr + 7) & low_bits(bit_mark - 32 > 0 ? bit_mark - 32 : 0);
    const uint32_t preceding = __popc(v0) + __popc(v1);
    return csvo_u32(sc, material_section_ptr + material_section_offset * 4 + preceding * 4);
}

// the same on a bare buffer resource over the world's bytes (walk_voxel_on_bytes)
__device__ __forceinline__ uint32_t csvo_read_leaf_at(buf_t world, uint32_t material_section_ptr, uint32_t pre_leaf_ptr, uint32_t ptr, uint32_t idx) {
    auto u32_at = [&](uint32_t p) -> uint32_t { return buf_u32(world, 8u + csvo_clamp(p)); };
    const uint32_t material_section_offset = u32_at(pre_leaf_ptr + 1) & 0xffffu;
    const int leaf_index = int(ptr - (pre_leaf_ptr + 3));
    const int bit_mark = leaf_index * 8 + int(idx);
    const uint32_t v0 = u32_at(pre_leaf_ptr + 3) & low_bits(bit_mark < 32 ? bit_mark : 32);
    const uint32_t v1 = u32_at(pre_leaf_ptr + 7) & low_bits(bit_mark - 32 > 0 ? bit_mark - 32 : 0);
    const uint32_t preceding = __popc(v0) + __popc(v1);
    return u32_at(material_section_ptr + material_section_offset * 4 + preceding * 4);
}

__device__ __forceinline__ vx_material material_at(const DevScene& sc, uint32_t value) {
    // 32-byte rows: two 16-byte loads; block ids beyond the table read as an all-zero row (range-checked by the V#)
    const uint32_t off = value < 0x07ffffffu ? value * 32u : 0xffffffe0u;
    const uint4 a = buf_u128(sc.materials, off), b = buf_u128(sc.materials, off + 16u);
    vx_material m;
    m.specular_pow = __uint_as_float(a.x); m.specular_strength = __uint_as_float(a.y);
    m.tex_top = int(a.z); m.tex_side = int(a.w); m.tex_bottom = int(b.x);
    m.tex_top_normal = int(b.y); m.tex_side_normal = int(b.z); m.tex_bottom_normal = int(b.w);
    return m;
}

// ---- software sampler: textureLod(sampler2DArray) with the state of texture_array.rs:200-203 ---------------
//   MAG NEAREST, MIN LINEAR_MIPMAP_LINEAR, WRAP_S CLAMP_TO_EDGE, WRAP_T REPEAT (never set -> GL default)

// float(b) / 255.0f for b in [0, 255], bit for bit (checked for all 256 values with exact rational arithmetic): one
// Newton step on b * RN(1/255) instead of the ~10-instruction IEEE division sequence, four times per texel
__device__ __forceinline__ float unorm8(uint32_t b) {
    const float x = float(b), r = 1.0f / 255.0f;
    const float q0 = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q0, 255.0f, x), r, q0);
}

__device__ __forceinline__ void unpack_rgba8(uint32_t rgba, float out[4]) {
    out[0] = unorm8(rgba & 0xffu);
    out[1] = unorm8((rgba >> 8) & 0xffu);
    out[2] = unorm8((rgba >> 16) & 0xffu);
    out[3] = unorm8(rgba >> 24);
}

// one mip level of one layer: dimensions, byte offset of the layer, and the two wrap modes
struct TexLevel {
    int w, h;
    uint32_t base;
    bool pow2_h;  // wave-uniform: REPEAT is a mask instead of a signed modulo
    __device__ __forceinline__ TexLevel(const DevTextures& t, uint32_t level, uint32_t layer) : TexLevel(t, level, layer, t.level_offset[level]) {}
    // (the level's offset handed in: read ahead of the chain of accesses the sample hangs on, texture_lod_pair)
    __device__ __forceinline__ TexLevel(const DevTextures& t, uint32_t level, uint32_t layer, uint32_t level_offset) {
        const uint32_t ww = t.width >> level, hh = t.height >> level;
        w = int(ww ? ww : 1);
        h = int(hh ? hh : 1);
        base = level_offset + layer * uint32_t(h) * uint32_t(w) * 4u;
        pow2_h = t.pow2_height || (t.height & (t.height - 1)) == 0;
    }
    __device__ __forceinline__ int clamp_s(int x) const { x = x < 0 ? 0 : x; return x > w - 1 ? w - 1 : x; }
    __device__ __forceinline__ int repeat_t(int y) const {
        if (pow2_h) return y & (h - 1);
        // Any other height: |y| mod h by binary long division from |y|'s highest bit down -- a loop of five instructions, not `%` (the
        // compiler's 32-bit modulo is ~35 instructions, inlined at every tap of every sample). Coordinates are within a row or two of
        // the texture, so the loop makes about as many trips as h has bits.
        const uint32_t uh = uint32_t(h), a = y < 0 ? 0u - uint32_t(y) : uint32_t(y);
        uint32_t r = 0;
#pragma unroll 1
        for (int i = 31 - __builtin_clz(a | 1u); i >= 0; --i) {
            r = (r << 1) | ((a >> i) & 1u);
            r = r >= uh ? r - uh : r;
        }
        return int((y < 0 && r != 0u) ? uh - r : r);
    }
    __device__ __forceinline__ uint32_t offset(int x, int y) const { return base + (uint32_t(y) * uint32_t(w) + uint32_t(x)) * 4u; }
};

// Bilinear tap of one level in BYTE units (0..255 as floats): the common factor 1/255 of the four texels is applied once, by
// the caller, after all blending -- 4 multiplications per sample instead of 32 exact divisions. The result differs from
// "convert each texel, then blend" only in rounding (< 4e-7 absolute), which GL leaves to the implementation anyway, and
// is > 0 in exactly the same cases (all terms are non-negative and none can underflow), so the alpha test is unaffected.
__device__ __forceinline__ void sample_linear_bytes(const DevTextures& t, uint32_t level, uint32_t layer, float u, float v, float out[4]) {
    const TexLevel L(t, level, layer);
    const float x = u * float(L.w) - 0.5f, y = v * float(L.h) - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float ax = x - fx, ay = y - fy;
    const int i0 = int(fx), j0 = int(fy);
    const int x0 = L.clamp_s(i0), x1 = L.clamp_s(i0 + 1), y0 = L.repeat_t(j0), y1 = y0 + 1 == L.h ? 0 : y0 + 1;  // (= repeat_t(j0 + 1))
    const uint32_t r00 = buf_u32(t.buf, L.offset(x0, y0)), r10 = buf_u32(t.buf, L.offset(x1, y0));
    const uint32_t r01 = buf_u32(t.buf, L.offset(x0, y1)), r11 = buf_u32(t.buf, L.offset(x1, y1));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float c00 = float((r00 >> (8 * k)) & 0xffu), c10 = float((r10 >> (8 * k)) & 0xffu);
        const float c01 = float((r01 >> (8 * k)) & 0xffu), c11 = float((r11 >> (8 * k)) & 0xffu);
        const float lo = c00 * (1.0f - ax) + c10 * ax;
        const float hi = c01 * (1.0f - ax) + c11 * ax;
        out[k] = lo * (1.0f - ay) + hi * ay;
    }
}

__device__ __forceinline__ void texture_lod(const DevTextures& t, float u, float v, float layer_f, float lod, float rgba[4]) {
    if (t.levels == 0 || t.layers == 0) {
        rgba[0] = rgba[1] = rgba[2] = rgba[3] = 0.0f;
        return;
    }
    const float lf = floorf(layer_f + 0.5f);
    const uint32_t layer = lf <= 0.0f ? 0u : (lf >= float(t.layers - 1) ? t.layers - 1 : uint32_t(lf));
    if (!(lod > 0.0f)) {  // magnification: NEAREST on the base level (exact texel values)
        const TexLevel L(t, 0, layer);
        unpack_rgba8(buf_u32(t.buf, L.offset(L.clamp_s(int(floorf(u * float(L.w)))), L.repeat_t(int(floorf(v * float(L.h)))))), rgba);
        return;
    }
    const float q = float(t.levels - 1);
    const float lam = lod > q ? q : lod;
    const float fl = floorf(lam);
    const uint32_t d1 = uint32_t(fl);
    const uint32_t d2 = d1 + 1 > t.levels - 1 ? t.levels - 1 : d1 + 1;
    const float frac = lam - fl;
    float a[4], b[4];
    sample_linear_bytes(t, d1, layer, u, v, a);
    sample_linear_bytes(t, d2, layer, u, v, b);
#pragma unroll
    for (int k = 0; k < 4; ++k) rgba[k] = (a[k] * (1.0f - frac) + b[k] * frac) * (1.0f / 255.0f);
}

// Two layers sampled at the same coordinates and level of detail -- a shaded hit's normal map and its colour: texture_lod twice, value for value, but
// with the texels of BOTH samples (and of both mip levels of each) requested before any of them is looked at. A service phase lasts as long as its
// longest chain of dependent memory accesses, and one sample after the other was six of them (level offsets, the four taps of the finer level, the
// four of the coarser, twice over); side by side they are two. The layers of a texture array share their dimensions, so the taps' places within a
// layer are worked out once. `want_a` / `want_b`: which of the two this lane wants at all (the other's result is left alone).
// the mip levels a sample at `lod` blends, and where they start in the chain (two words of the level table, indexed per lane: memory accesses -- made
// where this is called, ahead of whatever the sample's layer still waits for)
struct TexLod {
    uint32_t d1, d2, offset1, offset2;
    float frac;
    bool nearest;
};
__device__ __forceinline__ TexLod texture_levels(const DevTextures& t, float lod) {
    TexLod l = {0u, 0u, 0u, 0u, 0.0f, true};
    if (t.levels == 0 || t.layers == 0) return l;
    l.nearest = !(lod > 0.0f);
    if (!l.nearest) {
        const float q = float(t.levels - 1);
        const float lam = lod > q ? q : lod;
        const float fl = floorf(lam);
        l.d1 = uint32_t(fl);
        l.d2 = l.d1 + 1 > t.levels - 1 ? t.levels - 1 : l.d1 + 1;
        l.frac = lam - fl;
    }
    l.offset1 = t.level_offset[l.d1];
    l.offset2 = t.level_offset[l.d2];
    return l;
}

__device__ __forceinline__ void texture_lod_pair(const DevTextures& t, const TexLod& lv, float u, float v, float layer_a_f, float layer_b_f, bool want_a, bool want_b,
                                                 float a_rgba[4], float b_rgba[4]) {
    if (t.levels == 0 || t.layers == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (want_a) a_rgba[k] = 0.0f;
            if (want_b) b_rgba[k] = 0.0f;
        }
        return;
    }
    auto layer_of = [&](float layer_f) -> uint32_t {
        const float lf = floorf(layer_f + 0.5f);
        return lf <= 0.0f ? 0u : (lf >= float(t.layers - 1) ? t.layers - 1 : uint32_t(lf));
    };
    const uint32_t layer_a = layer_of(layer_a_f), layer_b = layer_of(layer_b_f);
    if (lv.nearest) {  // magnification: NEAREST on the base level (exact texel values)
        const TexLevel L(t, 0, 0, lv.offset1);
        const uint32_t texel = (uint32_t(L.repeat_t(int(floorf(v * float(L.h))))) * uint32_t(L.w) + uint32_t(L.clamp_s(int(floorf(u * float(L.w)))))) * 4u;
        const uint32_t layer_bytes = uint32_t(L.h) * uint32_t(L.w) * 4u;
        const uint32_t ra = buf_u32(t.buf, L.base + layer_a * layer_bytes + texel), rb = buf_u32(t.buf, L.base + layer_b * layer_bytes + texel);
        if (want_a) unpack_rgba8(ra, a_rgba);
        if (want_b) unpack_rgba8(rb, b_rgba);
        return;
    }
    const float frac = lv.frac;
    // the four taps of a level: their places within a layer and their weights (sample_linear_bytes)
    struct Taps { uint32_t base, layer_bytes, o00, o10, o01, o11; float ax, ay; };
    auto taps_of = [&](uint32_t level, uint32_t level_offset) -> Taps {
        const TexLevel L(t, level, 0, level_offset);
        const float x = u * float(L.w) - 0.5f, y = v * float(L.h) - 0.5f;
        const float fx = floorf(x), fy = floorf(y);
        const int i0 = int(fx), j0 = int(fy);
        const int x0 = L.clamp_s(i0), x1 = L.clamp_s(i0 + 1), y0 = L.repeat_t(j0), y1 = y0 + 1 == L.h ? 0 : y0 + 1;
        Taps tp;
        tp.base = L.base;
        tp.layer_bytes = uint32_t(L.h) * uint32_t(L.w) * 4u;
        tp.o00 = (uint32_t(y0) * uint32_t(L.w) + uint32_t(x0)) * 4u; tp.o10 = (uint32_t(y0) * uint32_t(L.w) + uint32_t(x1)) * 4u;
        tp.o01 = (uint32_t(y1) * uint32_t(L.w) + uint32_t(x0)) * 4u; tp.o11 = (uint32_t(y1) * uint32_t(L.w) + uint32_t(x1)) * 4u;
        tp.ax = x - fx; tp.ay = y - fy;
        return tp;
    };
    const Taps t1 = taps_of(lv.d1, lv.offset1), t2 = taps_of(lv.d2, lv.offset2);
    uint32_t ra[8] = {}, rb[8] = {};  // [level][tap]
    auto request = [&](uint32_t layer, uint32_t r[8]) {
        const uint32_t b1 = t1.base + layer * t1.layer_bytes, b2 = t2.base + layer * t2.layer_bytes;
        r[0] = buf_u32(t.buf, b1 + t1.o00); r[1] = buf_u32(t.buf, b1 + t1.o10); r[2] = buf_u32(t.buf, b1 + t1.o01); r[3] = buf_u32(t.buf, b1 + t1.o11);
        r[4] = buf_u32(t.buf, b2 + t2.o00); r[5] = buf_u32(t.buf, b2 + t2.o10); r[6] = buf_u32(t.buf, b2 + t2.o01); r[7] = buf_u32(t.buf, b2 + t2.o11);
    };
    // (both layers' texels are requested whether wanted or not -- an unwanted sample's sixteen bytes cost nothing beside its neighbour's, and a request
    // under its own condition would be waited for before the other is made)
    request(layer_a, ra);
    request(layer_b, rb);
    auto blend = [&](const uint32_t r[8], float rgba[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float level[2];
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                const float ax = l ? t2.ax : t1.ax, ay = l ? t2.ay : t1.ay;
                const float c00 = float((r[4 * l] >> (8 * k)) & 0xffu), c10 = float((r[4 * l + 1] >> (8 * k)) & 0xffu);
                const float c01 = float((r[4 * l + 2] >> (8 * k)) & 0xffu), c11 = float((r[4 * l + 3] >> (8 * k)) & 0xffu);
                const float lo = c00 * (1.0f - ax) + c10 * ax;
                const float hi = c01 * (1.0f - ax) + c11 * ax;
                level[l] = lo * (1.0f - ay) + hi * ay;
            }
            rgba[k] = (level[0] * (1.0f - frac) + level[1] * frac) * (1.0f / 255.0f);
        }
    };
    if (want_a) blend(ra, a_rgba);
    if (want_b) blend(rb, b_rgba);
}

// ---- intersect_octree as a resumable per-lane state machine ------------------------------------------------------
//
// The reference's loop body has three very different costs: the common descend/advance/pop step, the rare leaf test
// (material row + texture sample, svo.esvo.glsl:185-265) and termination. Run as written, one lane at a leaf stalls
// the other 63. Here one ray is a `Trav` whose step() performs exactly one loop iteration of the reference and
// reports when the ray is AT a leaf instead of testing it, so a wavefront can park such lanes and test them
// together (render kernel), while the picker / debug kernels simply call step() and leaf_test() back to back.
//
// What a Trav keeps between iterations is chosen so that an iteration that neither descends nor pops touches no memory:
//   ESVO  the reference identifies the node being examined as (ptr, parent_octant_idx) = "slot of my parent's octant"
//         and, every iteration, re-reads that slot's masks; on PUSH it reads the slot's pointer and then, dependent
//         on it, the child's masks (svo.esvo.glsl:168-173, 283-290). Here the node is identified by `ptr` = the
//         pointer to its OWN octant (what the reference calls the child pointer) plus `node` = its masks. Both
//         words a PUSH needs (child pointer, child masks) are then in the octant at `ptr`: two independent loads.
//         A leaf's value is one load, and POP restores the masks from the stack instead of re-reading them.
//         Same words, same values -- only read once, and earlier. (ptr, parent_octant_idx) are still tracked for
//         the debug trace, which reports them per iteration.
//   CSVO  the node header (1 or 2 bytes, svo.csvo.glsl:53-116) is read when the node is entered and kept in `node`
//         (and on the stack); the pointer-table entry is only read by the iteration that descends through it.
// kTravDeep (fast stacks only): the next PUSH would leave the LDS-resident levels; the cursor is untouched and the caller
// continues this ray with a full stack.
// kTravForeign (FOREIGN steps only: the traversal image of a CSVO world): the ray is about to be led INTO a voxel (it started
// inside it, svo.csvo.glsl:293-295). What the reference does in there depends on the bytes that follow the voxel's parent in the
// world's own buffer (read_next_ptr, svo.csvo.glsl:107-115, applied below the leaves), so the caller lets the ray make that
// excursion on the world's own bytes and bring it back to the image (walk_voxel_on_bytes). The iteration is repeated there: the
// caller takes `iter` back by one, as for kTravDeep. (An ESVO world's image needs none of this: see the PUSH in step_with.)
// (the values are the render kernel's lane states -- kTrav, kLeaf, kMissed, kDeep, kForeign -- so that it can store a status as is)
enum TravStatus : int { kTravContinue = 1, kTravAtLeaf = 2, kTravFinished = 4, kTravDeep = 5, kTravForeign = 6 };
enum LeafOutcome : int { kLeafHit = 0, kLeafPassed = 1, kLeafPassedAndFinished = 2 };

// Debug-trace state (trace kernel only): the output frames, and the reference's (ptr, parent_octant_idx) view of the
// ESVO cursor with its own shadow stack.
struct TraceSink {
    vx_frame* frames;
    uint32_t max_frames, n_frames;
    uint32_t ref_ptr, ref_aux;
    uint32_t stack_ptr[kMaxScale];
    uint8_t stack_aux[kMaxScale];
};

typedef VX_AS_PRIVATE TraceSink* TracePtr;

template <int SVO>
struct Trav {
    static constexpr bool CSVO = SVO == VX_SVO_CSVO;
    static constexpr bool WIDE = SVO == VX_SVO_IMAGE_WIDE;
    static constexpr bool IMG = SVO == VX_SVO_IMAGE || WIDE;
    static constexpr bool BIG = SVO == VX_SVO_ESVO_BIG;

    // ESVO: descriptors[index] (svo.esvo.glsl:3-6); beyond the buffer reads 0 either way
    __device__ __forceinline__ static uint32_t word(const DevScene& sc, uint32_t index) {
        if (BIG) {
            const uint64_t off = 4ull + uint64_t(index) * 4ull;
            return off + 4ull <= sc.wide_bytes ? mem_u32(sc.wide + off) : 0u;
        }
        return esvo_word(sc, index);
    }

    float rox, roy, roz, rdx, rdy, rdz;   // origin in [1,2) space, epsilon-clamped direction
    float tcx, tcy, tcz, tbx, tby, tbz;   // t(x) = x * t_coef - t_bias per axis
    float px, py, pz;                     // current octant corner
    float t_min, t_max, h, scale_exp2;
    float max_dst;                        // already scaled to [0,1]; < 0 = unlimited
    uint32_t ptr;                         // ESVO: word index of the examined node's own octant; CSVO: its byte pointer; image: byte offset of its octant
    uint32_t node;                        // ESVO: child_mask << 8 | leaf_mask of the examined node; CSVO: its header
    uint32_t depth;                       // CSVO only (svo.csvo.glsl:254); wraps below 0 exactly like the reference's uint
    uint32_t material_section_ptr, pre_leaf_pointer;  // CSVO only
    // svo.esvo.glsl:241-265 keeps adjacent_leaf_count and last_leaf_value; only "count != 0" is ever observed, and the value
    // is only compared while the count is non-zero, so the reset needs to touch the flag alone
    uint32_t last_leaf_value;
    uint32_t flags;                       // kInsideVoxel | kHasAdjacentLeaf (one word: no per-flag storage for the optimiser to split)
    static constexpr uint32_t kInsideVoxel = 1u, kHasAdjacentLeaf = 2u;
    __device__ __forceinline__ bool inside_voxel() const { return (flags & kInsideVoxel) != 0; }
    int scale, idx, octant_mask;
    uint32_t iter;                        // loop iterations executed (the reference's `i`)

    // CSVO header of the node at `ptr`, normalised to the internal-node form (2 bits per child): the 1-bit-per-child
    // headers of the three lowest levels are spread to tag 01 per present child, after which child lookup and the
    // popcount-style table offsets are the same expressions at every depth (popcount(h & below) == tag_bytes(h' & below')).
    __device__ __forceinline__ uint32_t csvo_header(const DevScene& sc) const {
        const uint32_t raw = csvo_u32(sc, ptr);
        uint32_t x = raw & 0xffu;
        x = (x | (x << 4)) & 0x0f0fu;
        x = (x | (x << 2)) & 0x3333u;
        x = (x | (x << 1)) & 0x5555u;
        return depth > 3 ? raw & 0xffffu : x;
    }

    template <bool TRACE = false>
    __device__ __forceinline__ void init(const DevScene& sc, const float ro_in[3], const float rd_in[3], float max_dst_in, TracePtr tk = nullptr) {
        const float octree_scale = sc.octree_scale;
        rox = ro_in[0] * octree_scale; roy = ro_in[1] * octree_scale; roz = ro_in[2] * octree_scale;
        max_dst = max_dst_in * octree_scale;
        rox += 1.0f; roy += 1.0f; roz += 1.0f;
        aim<TRACE>(sc, rd_in, tk);
    }

    // A ray whose origin is already where init() puts it (octree space, [1, 2)^3): what a cursor on another encoding of the same
    // world took down for it (render_persistent's list of rays for the world's own bytes). max_dst as for init().
    __device__ __forceinline__ void init_in_octree_space(const DevScene& sc, float ox, float oy, float oz, const float rd_in[3], float max_dst_in) {
        rox = ox; roy = oy; roz = oz;
        max_dst = max_dst_in * sc.octree_scale;
        aim<false>(sc, rd_in, nullptr);
    }

    // the direction-dependent half of the set-up (svo.esvo.glsl:60-125) and the cursor at the root
    template <bool TRACE = false>
    __device__ __forceinline__ void aim(const DevScene& sc, const float rd_in[3], TracePtr tk = nullptr) {
        rdx = rd_in[0]; rdy = rd_in[1]; rdz = rd_in[2];
        const uint32_t eps_bits = __float_as_uint(kEps) & 0x7fffffffu;
        if (fabsf(rdx) < kEps) rdx = __uint_as_float(eps_bits | (__float_as_uint(rdx) & 0x80000000u));
        if (fabsf(rdy) < kEps) rdy = __uint_as_float(eps_bits | (__float_as_uint(rdy) & 0x80000000u));
        if (fabsf(rdz) < kEps) rdz = __uint_as_float(eps_bits | (__float_as_uint(rdz) & 0x80000000u));

        tcx = 1.0f / -fabsf(rdx); tcy = 1.0f / -fabsf(rdy); tcz = 1.0f / -fabsf(rdz);
        tbx = tcx * rox; tby = tcy * roy; tbz = tcz * roz;

        octant_mask = 0;
        if (rdx > 0.0f) { octant_mask ^= 1; tbx = __builtin_fmaf(3.0f, tcx, -tbx); }
        if (rdy > 0.0f) { octant_mask ^= 2; tby = __builtin_fmaf(3.0f, tcy, -tby); }
        if (rdz > 0.0f) { octant_mask ^= 4; tbz = __builtin_fmaf(3.0f, tcz, -tbz); }
        start<TRACE>(sc, tk);
    }

    // The cursor at the root, for the ray whose constants (origin, direction, t_coef, t_bias, octant_mask, max_dst) are in place:
    // the second half of the reference's set-up (svo.esvo.glsl:126-150).
    template <bool TRACE = false>
    __device__ __forceinline__ void start(const DevScene& sc, TracePtr tk = nullptr) {
        scale = kMaxScale - 1;
        scale_exp2 = 0.5f;
        last_leaf_value = 0xffffffffu;
        flags = 0;
        material_section_ptr = kInvalidPtr;
        pre_leaf_pointer = kInvalidPtr;
        iter = 0;

        t_min = gmax(gmax(__builtin_fmaf(2.0f, tcx, -tbx), __builtin_fmaf(2.0f, tcy, -tby)), __builtin_fmaf(2.0f, tcz, -tbz));
        t_min = gmax(0.0f, t_min);
        t_max = gmin(gmin(tcx - tbx, tcy - tby), tcz - tbz);
        h = t_max;

        idx = 0;
        px = 1.0f; py = 1.0f; pz = 1.0f;
        if (t_min < __builtin_fmaf(1.5f, tcx, -tbx)) { idx ^= 1; px = 1.5f; }
        if (t_min < __builtin_fmaf(1.5f, tcy, -tby)) { idx ^= 2; py = 1.5f; }
        if (t_min < __builtin_fmaf(1.5f, tcz, -tbz)) { idx ^= 4; pz = 1.5f; }

        if (CSVO) {
            ptr = sc.root_ptr;
            depth = 127u - ((__float_as_uint(sc.octree_scale) >> 23) & 0xffu);  // svo.csvo.glsl:254
            node = csvo_header(sc);
            if (depth == 2) pre_leaf_pointer = ptr;
        } else if (IMG) {
            depth = 0;
            node = sc.image_root_masks;  // header of the image: root masks, root octant (byte offset / index) -- read once per kernel, not per ray
            ptr = sc.image_root_octant;
        } else {
            // the reference starts at (ptr 0, parent_octant_idx 0): the preamble is an octant whose only child is the root
            depth = 0;
            node = word(sc, 0);
            const uint32_t w = word(sc, 4);
            ptr = (w & 0x80000000u) ? 4u + (w & 0x7fffffffu) : w;
            if (TRACE) { tk->ref_ptr = 0; tk->ref_aux = 0; }
        }
    }

    // ADVANCE + POP (svo.esvo.glsl:324-390). Returns false when the ray left the octree.
    template <bool TRACE, class ST>
    __device__ __forceinline__ bool advance(const DevScene& sc, const ST& st, float tcrx, float tcry, float tcrz, float tc_max, TracePtr tk) {
        (void)sc;
        const uint32_t ox = __float_as_uint(px), oy = __float_as_uint(py), oz = __float_as_uint(pz);
        t_min = tc_max;
        bool pop;
        uint32_t differing_bits;
        if (ST::kFast) {
            // scale >= 0 and the child index IS the position bit at `scale` (see step()): stepping an axis whose bit is set clears
            // just that bit, stepping one whose bit is clear borrows from above it. So "idx & step_mask after the flip"
            // (svo.esvo.glsl:337-343) == "the old and new corners differ above bit `scale`", and the differing bits are the POP's own
            // (svo.esvo.glsl:345-349: per stepped axis bits(pos) ^ bits(pos + scale_exp2), the subtraction being exact).
            // (written so that the stepped corner is the LAST thing computed from the old one: the new value can then live in the old
            // one's register, and the loop needs no copies where its paths meet. The old coordinate of a stepped axis is the new one
            // plus the step, exactly.)
            const float ax = tc_max >= tcrx ? scale_exp2 : 0.0f, ay = tc_max >= tcry ? scale_exp2 : 0.0f, az = tc_max >= tcrz ? scale_exp2 : 0.0f;
            px -= ax; py -= ay; pz -= az;
            differing_bits = (__float_as_uint(px + ax) ^ __float_as_uint(px)) | (__float_as_uint(py + ay) ^ __float_as_uint(py)) |
                             (__float_as_uint(pz + az) ^ __float_as_uint(pz));
            pop = differing_bits >= (2u << scale);
        } else {
            int step_mask = 0;
            if (tc_max >= tcrx) { step_mask ^= 1; px -= scale_exp2; }
            if (tc_max >= tcry) { step_mask ^= 2; py -= scale_exp2; }
            if (tc_max >= tcrz) { step_mask ^= 4; pz -= scale_exp2; }
            idx ^= step_mask;
            pop = (idx & step_mask) != 0;
            differing_bits = 0;
            if (pop) {
                // While scale >= 0 every coordinate is a multiple of scale_exp2 >= 2^-23 in [1, 2), the subtraction above was exact
                // and pos + scale_exp2 is the old coordinate again (unstepped axes contribute 0 on their own); below that (a ray
                // that started inside a voxel and was taken more than `depth` levels further down) the sums round and are formed
                // as written.
                if (scale >= 0) {
                    differing_bits = (ox ^ __float_as_uint(px)) | (oy ^ __float_as_uint(py)) | (oz ^ __float_as_uint(pz));
                } else {
                    if (step_mask & 1) differing_bits |= __float_as_uint(px) ^ __float_as_uint(px + scale_exp2);
                    if (step_mask & 2) differing_bits |= __float_as_uint(py) ^ __float_as_uint(py + scale_exp2);
                    if (step_mask & 4) differing_bits |= __float_as_uint(pz) ^ __float_as_uint(pz + scale_exp2);
                }
            }
        }

        bool inside = true;
        if (pop) {
            scale = (ST::kFast || differing_bits) ? 31 - __builtin_clz(differing_bits) : -1;  // (fast: pop implies bits above `scale`)
            inside = uint32_t(scale) < uint32_t(kMaxScale);
            if (inside) {
                scale_exp2 = pow2i(scale - kMaxScale);

                uint32_t a;
                st.pop(scale, ptr, t_max, a);
                if (CSVO) {
                    node = a & 0xffffu;
                    depth = uint32_t(int32_t(a) >> 16);  // sign-extended: depth spans [-23, 255] once a ray is below the leaves
                } else {
                    node = a;
                    if (TRACE) { tk->ref_ptr = tk->stack_ptr[scale]; tk->ref_aux = tk->stack_aux[scale]; }
                }

                // svo.esvo.glsl:372-385: drop the position bits below the new scale; the bit at the scale is the child index
                const uint32_t bx = __float_as_uint(px), by = __float_as_uint(py), bz = __float_as_uint(pz);
                const uint32_t keep = 0xffffffffu << scale;
                px = __uint_as_float(bx & keep);
                py = __uint_as_float(by & keep);
                pz = __uint_as_float(bz & keep);
                if (!ST::kFast) idx = int(((bx >> scale) & 1u) | (((by >> scale) & 1u) << 1) | (((bz >> scale) & 1u) << 2));
                h = 0.0f;
            }
        }
        return inside;
    }

    // One iteration of the reference's loop (svo.esvo.glsl:152-391 / svo.csvo.glsl:261-508) minus the leaf test.
    // LIMIT = the ray has a maximum distance (picker); render rays are unlimited and skip the test.
    // CAPPED = test the iteration cap here (a caller that already did passes false). FOREIGN: see kTravForeign.
    template <bool TRACE, bool STATS, bool LIMIT, class ST, bool CAPPED = true, bool FOREIGN = false>
    __device__ __forceinline__ TravStatus step(const DevScene& sc, const ST& st, TracePtr tk, Counters* ctr) {
        TravStatus status = kTravContinue;
        step_with<TRACE, STATS, LIMIT, ST, CAPPED, FOREIGN>(sc, st, tk, ctr, [&](TravStatus s) { status = s; });
        return status;
    }

    // The same with the outcome delivered to `on_exit(status)` -- called only when the ray stops being a plain traversal
    // (never with kTravContinue), from the spot where that is found out. The render kernel's loop records its lane state
    // there, which keeps the common paths free of a status value to merge.
    template <bool TRACE, bool STATS, bool LIMIT, class ST, bool CAPPED, bool FOREIGN, class EXIT>
    __device__ __forceinline__ void step_with(const DevScene& sc, const ST& st, TracePtr tk, Counters* ctr, EXIT&& on_exit) {
#if VX_MERGED_STEP
        if constexpr (IMG && ST::kFast && !ST::kHot && !TRACE && !STATS) return step_image<LIMIT, ST, CAPPED, FOREIGN>(sc, st, on_exit);
#endif
        bool live = !CAPPED || iter < uint32_t(kMaxSteps);
        if (LIMIT) live = live && !(max_dst >= 0.0f && t_min > max_dst);
        if (!live) return on_exit(kTravFinished);
        ++iter;
        if (STATS) ctr->iterations++;
        // While scale >= 0 the child index is the position's mantissa bit at `scale` on each axis (the corner is a multiple
        // of scale_exp2; PUSH adds half a cell = sets the bit, ADVANCE subtracts a cell = flips it, POP masks below it), so
        // the loop that only ever sees LDS-resident scales reads it off the position instead of maintaining it in three places.
        // (The member `idx` is not kept up to date by fast steps: sync_idx() before handing the ray to anything else.)
        const int cur_idx = ST::kFast ? idx_from_position() : idx;

        const float tcrx = __builtin_fmaf(px, tcx, -tbx), tcry = __builtin_fmaf(py, tcy, -tby), tcrz = __builtin_fmaf(pz, tcz, -tbz);
        const float tc_max = gmin3(tcrx, tcry, tcrz);
        const uint32_t octant_idx = uint32_t(cur_idx ^ octant_mask);

        // Image cursors (byte-offset layout): the entry a PUSH out of this node into this child would read is requested NOW, for every
        // lane, whether the lane turns out to descend or not -- the address is known, and everything the iteration does (the tests
        // below, the stack access, the other lanes' ADVANCE and POP) runs while it is in flight; it is looked at once, at the very end.
        // Requested for nothing by the lanes that do not descend: the same one load instruction per trip of a wave, and any address
        // is harmless (the unit in front of the octant for a child that does not exist, a unit among an octant's values; a byte-offset image is
        // read through a buffer resource with its range check; the wide layout's 64-bit addresses are kept valid: see `ptr` below).
        constexpr bool kAhead = IMG && !ST::kHot;
        uint2 ahead = make_uint2(0u, 0u);
        if (kAhead) ahead = image_entry<WIDE>(sc, ptr, node << octant_idx);
        bool pushed = false;
        const float cell_before = scale_exp2;

        bool is_child, is_leaf;
        uint32_t tag = 0;  // CSVO: the child's 2-bit pointer-width tag (01 for every present child of the 1-bit levels)
        if (IMG) {
            // image masks (traversal_image.hpp, oct64_masks): child c's "is a leaf" bit at 31 - c, its "exists" bit at 23 - c, so
            // that one shift brings both to fixed places ("leaf" into the sign) and leaves the existing children above c below bit 23
            const uint32_t m = node << octant_idx;
            is_child = (m & 0x00800000u) != 0;
            is_leaf = int32_t(m) < 0;
        } else if (!CSVO) {
            is_child = (node & (0x100u << octant_idx)) != 0;
            is_leaf = (node & (1u << octant_idx)) != 0;
        } else {
            tag = (node >> (octant_idx * 2)) & 3u;
            is_child = tag != 0;
            is_leaf = is_child && depth < 2;
            // (the reference's per-iteration `if (depth == 2) pre_leaf_pointer = ptr`, svo.csvo.glsl:283, is done once, when
            // a depth-2 node is entered: nothing else can change either value while the ray is inside that node's subtree)
            if (STATS) ctr->csvo_header_bytes += depth > 3 ? 2u : 1u;
        }

        static_assert(!(TRACE && IMG), "the debug trace reports the reference's own pointers: it runs on the world's own bytes");
        if (TRACE) {
            if (tk->n_frames < tk->max_frames) {
                const float octree_scale = sc.octree_scale;
                vx_frame& f = tk->frames[tk->n_frames];
                f.t_min = t_min * __uint_as_float(0x7f000000u - __float_as_uint(octree_scale));
                f.ptr = CSVO ? ptr : tk->ref_ptr;
                f.idx = octant_idx;
                f.parent_octant_idx = CSVO ? depth : tk->ref_aux;
                f.scale = scale;
                f.is_child = is_child;
                f.is_leaf = is_leaf;
                bool crossed = false;
                uint32_t hb = 0, pb = 0;
                f.next_ptr = CSVO ? csvo_next_ptr(sc, ptr, depth, octant_idx, crossed, hb, pb) : 0u;
                f.crossed_boundary = crossed;
            }
            ++tk->n_frames;
        }

        const bool descend = is_child && t_min <= t_max;
        if (!descend) flags &= ~kHasAdjacentLeaf;
        if (descend && is_leaf) {
            if (t_min > 0.0f) return on_exit(kTravAtLeaf);  // leaf_test() decides; the cursor is left untouched
            if (FOREIGN) return on_exit(kTravForeign);
            if (t_min == 0.0f) flags |= kInsideVoxel;
        }
        const float tv_max = gmin(t_max, tc_max);
        if (descend && t_min <= tv_max) {
            // ---- PUSH (svo.esvo.glsl:280-311, svo.csvo.glsl:387-426) ----
            if (ST::kCanOverflow && scale < ST::kBaseScale) {
                // this PUSH would write a slot below the resident ones: hand over. The iteration is repeated by the caller's
                // full-stack step, so the CALLER takes `iter` back by one (undoing it here would make the counter's update
                // path dependent, which costs every iteration a register copy).
                if (STATS) {
                    ctr->iterations--;
                    if (CSVO) ctr->csvo_header_bytes -= depth > 3 ? 2u : 1u;
                }
                return on_exit(kTravDeep);
            }
            if (STATS) ctr->pushes++;
            // Order of the block: (1) request what the descent reads, (2) the stack write and all arithmetic that does not
            // depend on it -- new scale, child centre distances, first child index and corner -- while the request is in
            // flight, (3) the dependent part. The scheduling fences keep the compiler from pulling (3) up to the loads.
            uint32_t w0 = 0, w1 = 0, table = 0, offset = 0;
            if (IMG) {
                // one aligned 8-byte entry: the child's octant and the child's masks (no clamp: image pointers are valid by construction)
                uint2 e = ahead;
                if (kAhead) {
                    // (requested at the top of the iteration)
                } else if (ST::kHot && scale >= kMaxScale - 2) {
                    // out of the root (scale 22) or of the root's child the ray is in (scale 21; which one: the position's bit 22)
                    const uint32_t at_root = uint32_t(bit_at(__float_as_uint(px), 22) | (bit_at(__float_as_uint(py), 22) << 1) | (bit_at(__float_as_uint(pz), 22) << 2)) ^ uint32_t(octant_mask);
                    e = st.hot_entry(scale == kMaxScale - 1 ? 0u : 1u + at_root, octant_idx);
                } else {
                    e = image_entry<WIDE>(sc, ptr, node << octant_idx);
                }
                if (!kAhead) w0 = e.x;
                // A ray that starts inside a voxel is led INTO it (the leaf was not accepted above: t_min <= 0). In an ESVO world a voxel's
                // own masks are zero in everything the serializer writes (esvo.rs:465-485 never ORs a leaf's masks into its parent's
                // header; the transcoder refuses worlds where that is not so), so the reference walks the voxel as an empty node: so
                // do we, whatever the entry holds (an octant of voxels has values where others have entries). The pointer is never used.
                // The image of a CSVO world never gets here (FOREIGN: walk_voxel_on_bytes).
                if (!kAhead) w1 = (!FOREIGN && is_leaf) ? 0u : e.y;
            } else if (!CSVO) {
                // the child's pointer word and the header word with its masks, both in the octant at `ptr`
                w0 = word(sc, ptr + 4 + octant_idx);
                w1 = word(sc, ptr + (octant_idx >> 1));
            } else {
                // read_next_ptr (svo.csvo.glsl:53-116) with the (normalised) header already at hand: internal nodes (2-byte
                // header, 1/2/4-byte table entries) and the depth-3 level (1-byte header, 1-byte entries) are the same
                // computation; the two lowest levels have no table
                offset = csvo_tag_bytes(node & ((1u << (octant_idx * 2)) - 1u));
                if (depth >= 3) {
                    table = ptr + (depth > 3 ? 2u : 1u);
                    w0 = csvo_u32(sc, table + offset);
                }
            }
            sched_fence();
            // The reference writes the parent's entry only where the ray leaves the child before it leaves the parent (tc_max < h, svo.esvo.glsl:292-296). A
            // write it skips is either of an entry that is never popped (tc_max == h: the ray leaves the parent with the child, the next POP goes above
            // this level) or of the very entry the slot already holds (h == 0 after a POP to this parent: same pointer, same masks, the t_max that was
            // just popped). So a cursor on an image writes at EVERY push and keeps no `h`: three instructions a trip of the render loop less, and a
            // ray's stack holds every ancestor of its cell (the shadow ray's start relies on it: Trav::descend_along).
            if (IMG || tc_max < h) {
                st.push(scale, ptr, t_max, CSVO ? (depth << 16) | node : node);
                if (TRACE && !CSVO) { tk->stack_ptr[scale] = tk->ref_ptr; tk->stack_aux[scale] = uint8_t(tk->ref_aux); }
            }
            const float half_scale = scale_exp2 * 0.5f;
            const float tcenx = __builtin_fmaf(half_scale, tcx, tcrx), tceny = __builtin_fmaf(half_scale, tcy, tcry),
                        tcenz = __builtin_fmaf(half_scale, tcz, tcrz);
            h = tc_max;
            --scale;
            scale_exp2 = half_scale;
            const bool upper_x = t_min < tcenx, upper_y = t_min < tceny, upper_z = t_min < tcenz;
            if (upper_x) px += scale_exp2;
            if (upper_y) py += scale_exp2;
            if (upper_z) pz += scale_exp2;
            if (!ST::kFast) idx = int(upper_x) | (int(upper_y) << 1) | (int(upper_z) << 2);
            t_max = tv_max;
            sched_fence();
            if (kAhead) {
                pushed = true;  // (pointer and masks: below, where the other lanes' ADVANCE has been done too)
            } else if (IMG) {
                ptr = w0;
                node = w1;
            } else if (!CSVO) {
                if (TRACE) { tk->ref_ptr = ptr; tk->ref_aux = octant_idx; }
                ptr = (w0 & 0x80000000u) ? ptr + 4 + octant_idx + (w0 & 0x7fffffffu) : w0;
                node = (octant_idx & 1u) ? w1 >> 16 : w1;
            } else {
                uint32_t next_ptr = ptr + 3 + offset;
                bool crossed = false;
                if (depth >= 3) {
                    const uint32_t e = w0 & (0xffffffffu >> ((0x001018u >> ((tag - 1) * 8)) & 0xffu));
                    if (STATS) ctr->csvo_pointer_bytes += (1u << tag) >> 1;
                    crossed = (e & 0x80000000u) != 0;
                    next_ptr = crossed ? e ^ 0x80000000u : table + csvo_tag_bytes(node) + e;
                }
                --depth;
                ptr = next_ptr;
                if (crossed) {
                    if (STATS) ctr->boundaries++;
                    const uint32_t child_lod = csvo_u8(sc, ptr);
                    const uint32_t material_bytes = csvo_u32(sc, ptr + 1);
                    ptr += 5;
                    material_section_ptr = ptr;
                    ptr += material_bytes;
                    depth = child_lod;
                }
                node = csvo_header(sc);
                if (depth == 2) pre_leaf_pointer = ptr;
            }
            if (!kAhead) return;
        }
        if (!kAhead) {
            if (!advance<TRACE>(sc, st, tcrx, tcry, tcrz, tc_max, tk)) on_exit(kTravFinished);
            return;
        }
        if (!pushed && !advance<TRACE>(sc, st, tcrx, tcry, tcrz, tc_max, tk)) return on_exit(kTravFinished);
        // The entry requested at the top: the child's octant and the child's masks (a voxel's: none -- see the PUSH above). "This lane
        // descended" is read off the cell size (PUSH halves it, POP grows it) rather than off the path taken: a condition the compiler
        // cannot trace back to the branch keeps these two selects -- and with them the wait for the entry -- here, behind both paths,
        // instead of inside the PUSH path in front of the other lanes' ADVANCE.
        const bool descended = scale_exp2 < cell_before;
        // (wide layout: what a voxel's place in an octant of values holds is no pointer, and the next iteration requests an entry behind
        // whatever `ptr` is -- the image's first octant will do; the byte-offset layout's range check makes any value harmless)
        ptr = descended ? ((WIDE && !FOREIGN && is_leaf) ? 0u : ahead.x) : ptr;
        node = descended ? ((!FOREIGN && is_leaf) ? 0u : ahead.y) : node;
    }

    // The same iteration for a cursor on a traversal image with a fast stack -- the render loop's step -- as ONE instruction stream for
    // the lanes that PUSH and the lanes that ADVANCE. Measured on gfx950 (profiles/tools/valu_issue.hip, profiles/round3/): a SIMD
    // issues one instruction of a wave every ~2 cycles whatever its kind -- a scalar mask operation or a branch costs as much as a
    // vector instruction, a compare into a scalar pair and a select out of one half as much again -- and the loop as the compiler lays
    // out step_with (81 vector + 50 scalar instructions: the PUSH, ADVANCE and POP paths one after the other, each behind its own
    // execution-mask bookkeeping) runs at 91 % of what that costs. So the two paths are merged, not branched around:
    //   PUSH     corner += (t_min  <  t(centre plane)) ? half a cell : 0      per axis
    //   ADVANCE  corner -= (tc_max >= t(corner plane)) ? a cell      : 0      per axis
    // are the same compare-and-select with the operands chosen per lane: t(centre) = fma(half, t_coef, t(corner)), so with `hm` = half a
    // cell for a PUSH lane and 0 for an ADVANCE lane one fma gives either plane's distance (fma(0, c, x) == x exactly), L < R is the
    // PUSH's condition and the negation of the ADVANCE's, and the select's two values (hm : other) = (half : 0) or (0 : -cell) are per-lane
    // constants. Every float is produced by the operation the reference uses on the operands the reference uses: results are bit for
    // bit those of step_with (tests/test_device_on_host.py steps this very code against the oracle; test_kernel_versions_agree
    // compares the builds on the GPU).
    template <bool LIMIT, class ST, bool CAPPED, bool FOREIGN, class EXIT>
    __device__ __forceinline__ void step_image(const DevScene& sc, const ST& st, EXIT&& on_exit) {
        static_assert(IMG && ST::kFast && !ST::kHot, "image cursors on the loop's stack");
        bool live = !CAPPED || iter < uint32_t(kMaxSteps);
        if (LIMIT) live = live && !(max_dst >= 0.0f && t_min > max_dst);
        if (!live) return on_exit(kTravFinished);
        ++iter;
        const uint32_t bx = __float_as_uint(px), by = __float_as_uint(py), bz = __float_as_uint(pz);
        const uint32_t octant_idx = (bit_at(bx, scale) | (bit_at(by, scale) << 1) | (bit_at(bz, scale) << 2)) ^ uint32_t(octant_mask);
        const uint32_t m = node << octant_idx;
        // the entry a PUSH into this child reads, requested for every lane (see step_with): looked at last
        const uint2 ahead = image_entry<WIDE>(sc, ptr, m);
        const float tcrx = __builtin_fmaf(px, tcx, -tbx), tcry = __builtin_fmaf(py, tcy, -tby), tcrz = __builtin_fmaf(pz, tcz, -tbz);
        const float tc_max = gmin3(tcrx, tcry, tcrz);
        const bool is_child = (m & 0x00800000u) != 0, is_leaf = int32_t(m) < 0;
        const bool descend = is_child && t_min <= t_max;
        flags = descend ? flags : (flags & ~kHasAdjacentLeaf);
        if (descend && is_leaf) {
            if (t_min > 0.0f) return on_exit(kTravAtLeaf);
            if (FOREIGN) return on_exit(kTravForeign);
            if (t_min == 0.0f) flags |= kInsideVoxel;
        }
        const float tv_max = gmin(t_max, tc_max);
        const bool push = descend && t_min <= tv_max;
        if (ST::kCanOverflow && push && scale < ST::kBaseScale) return on_exit(kTravDeep);  // (the caller takes `iter` back by one)

        const float half = scale_exp2 * 0.5f;
        const float hm = push ? half : 0.0f, other = push ? 0.0f : -scale_exp2;
        const float lhs = push ? t_min : tc_max;
        const float rx = __builtin_fmaf(hm, tcx, tcrx), ry = __builtin_fmaf(hm, tcy, tcry), rz = __builtin_fmaf(hm, tcz, tcrz);
        if (push) st.push(scale, ptr, t_max, node);  // (at every push, no `h`: see step_with)
        px += lhs < rx ? hm : other;
        py += lhs < ry ? hm : other;
        pz += lhs < rz ? hm : other;
        // ADVANCE lanes: the stepped corner differs from the old one above bit `scale` exactly when the step left the parent (see advance())
        const uint32_t differing_bits = (bx ^ __float_as_uint(px)) | (by ^ __float_as_uint(py)) | (bz ^ __float_as_uint(pz));
        const bool pop = !push && differing_bits >= (2u << scale);
        t_min = push ? t_min : tc_max;
        t_max = push ? tv_max : t_max;
        scale_exp2 = push ? half : scale_exp2;
        scale = push ? scale - 1 : scale;
        bool inside = true;
        if (pop) {
            scale = 31 - __builtin_clz(differing_bits);  // (pop implies bits above the old scale: never 0)
            inside = uint32_t(scale) < uint32_t(kMaxScale);
            if (inside) {
                scale_exp2 = pow2i(scale - kMaxScale);
                uint32_t a;
                st.pop(scale, ptr, t_max, a);
                node = a;
                const uint32_t keep = 0xffffffffu << scale;
                px = __uint_as_float(__float_as_uint(px) & keep);
                py = __uint_as_float(__float_as_uint(py) & keep);
                pz = __uint_as_float(__float_as_uint(pz) & keep);
            }
        }
        // the child's octant and masks. (A ray led into a voxel of an ESVO world walks it as an empty node whatever the entry holds, and
        // the wide layout's next request needs a valid octant: see step_with.)
        ptr = push ? ((WIDE && !FOREIGN && is_leaf) ? 0u : ahead.x) : ptr;
        node = push ? ((!FOREIGN && is_leaf) ? 0u : ahead.y) : node;
        if (!inside) return on_exit(kTravFinished);
    }

    // The un-mirrored corner of the cell the cursor is at (svo.esvo.glsl:205-207), in the octree's [1, 2)^3: for a ray at a leaf, the voxel's.
    __device__ __forceinline__ void cell_corner(float q[3]) const {
        q[0] = (octant_mask & 1) ? 3.0f - scale_exp2 - px : px;
        q[1] = (octant_mask & 2) ? 3.0f - scale_exp2 - py : py;
        q[2] = (octant_mask & 4) ? 3.0f - scale_exp2 - pz : pz;
    }

    // A ray that starts in or next to a voxel another ray of this lane has just hit -- a pixel's shadow ray (world.glsl:79-84: from the hit, 0.001 along the
    // normal) -- descends from the root through nodes the first ray's stack still holds: every cursor on an image writes its parent's entry at EVERY push
    // (step_with), so when the first ray stood at the voxel, slot s held the node at scale s on the voxel's path for every scale above the voxel's parent, and
    // the caller has put the parent itself into its slot (`parent_scale`). The reference spends an iteration of its loop per level on that descent
    // (svo.esvo.glsl:152-311: fetch the node, test the child, PUSH) -- 10 of a depth-12 shadow ray's ~30, each a trip of the render loop for the whole wave.
    // Here the levels are run through in one go for as long as the ray stays ON the path: the child cell it is about to enter is the path's (`q`: the
    // voxel's un-mirrored corner; cells are nested, so a cell is on the path iff its corner is q's prefix) -- that child exists and is a node, nothing need be
    // fetched to know it -- and the reference would PUSH (t_min <= t_max, t_min <= min(t_max, tc_max)). Per level exactly the floats the PUSH computes: the
    // cell's exit distances, tc_max, min(t_max, tc_max), the centre planes' distances, the child chosen by t_min against them; the stack gets this ray's t_max
    // (pointer and masks are in place). Where the ray leaves the path (its origin lies in a neighbouring cell: at depth <= 12 the offset takes it out of the
    // voxel; or the reference would ADVANCE) the cursor stands in the last node of the path it reached, about to examine its child -- an ordinary state of
    // the traversal, and the loop takes over. Called on a cursor fresh from init().
    template <class ST>
    __device__ __forceinline__ void descend_along(const ST& st, int parent_scale, const float q[3]) {
        static_assert(IMG && ST::kFast, "image cursors on the loop's stack");
        if (!(parent_scale >= ST::kBaseScale && parent_scale < kMaxScale)) return;
        const float cell = pow2i(parent_scale - kMaxScale);
        // the voxel's corner in THIS ray's mirrored coordinates
        const uint32_t ex = __float_as_uint((octant_mask & 1) ? 3.0f - cell - q[0] : q[0]);
        const uint32_t ey = __float_as_uint((octant_mask & 2) ? 3.0f - cell - q[1] : q[1]);
        const uint32_t ez = __float_as_uint((octant_mask & 4) ? 3.0f - cell - q[2] : q[2]);
        const int from = scale;
        while (scale > parent_scale) {
            // the child cell the cursor is at: on the path? (its corner's bits from `scale` up are the voxel's)
            const uint32_t off = (__float_as_uint(px) ^ ex) | (__float_as_uint(py) ^ ey) | (__float_as_uint(pz) ^ ez);
            if ((off >> scale) != 0u) break;
            const float tcrx = __builtin_fmaf(px, tcx, -tbx), tcry = __builtin_fmaf(py, tcy, -tby), tcrz = __builtin_fmaf(pz, tcz, -tbz);
            const float tc_max = gmin3(tcrx, tcry, tcrz);
            const float tv_max = gmin(t_max, tc_max);
            if (!(t_min <= t_max && t_min <= tv_max)) break;  // (the reference ADVANCEs here)
            st.set_t_max(scale, t_max);
            const float half = scale_exp2 * 0.5f;
            const float rx = __builtin_fmaf(half, tcx, tcrx), ry = __builtin_fmaf(half, tcy, tcry), rz = __builtin_fmaf(half, tcz, tcrz);
            px += t_min < rx ? half : 0.0f;
            py += t_min < ry ? half : 0.0f;
            pz += t_min < rz ? half : 0.0f;
            t_max = tv_max;
            scale_exp2 = half;
            --scale;
            ++iter;
        }
        if (scale != from) {  // the node the cursor has reached: the path's at this scale
            float unused;
            st.pop(scale, ptr, unused, node);
        }
    }

    // image octants: an octant all of whose children are leaves (child bits 23..16 == leaf bits 31..24 of its masks) holds u32 values only -- for the
    // existing children, child 7 first, from unit ptr + 1 on --, any other a {value | pointer, masks} entry per existing child (traversal_image.hpp)
    __device__ __forceinline__ uint32_t image_leaf_value(const DevScene& sc, uint32_t octant_idx) const {
        const uint32_t k = uint32_t(__popc((node << octant_idx) & 0x00ffffffu));  // this child and the existing ones above it
        const bool values_only = (((node >> 8) ^ node) & 0x00ff0000u) == 0u;
        return values_only ? image_u32<WIDE>(sc, ptr + 1u, (k - 1u) * 4u) : image_u32<WIDE>(sc, ptr + k, 0u);
    }

    // child index from the corner's mantissa bits (scale >= 0)
    __device__ __forceinline__ int idx_from_position() const {
        return int(bit_at(__float_as_uint(px), scale) | (bit_at(__float_as_uint(py), scale) << 1) | (bit_at(__float_as_uint(pz), scale) << 2));
    }
    __device__ __forceinline__ void sync_idx() { idx = idx_from_position(); }

    // The voxel's value (block id) for a ray whose step() returned kTravAtLeaf.
    __device__ __forceinline__ uint32_t leaf_value(const DevScene& sc) const {
        const uint32_t octant_idx = uint32_t(idx ^ octant_mask);
        return CSVO ? csvo_read_leaf(sc, material_section_ptr, pre_leaf_pointer, ptr, octant_idx)
               : IMG ? image_leaf_value(sc, octant_idx)
                     : word(sc, ptr + 4 + octant_idx);
    }

    // Which face of the voxel the ray enters through and where on it (svo.esvo.glsl:196-233): arithmetic on the cursor only.
    struct LeafSurface {
        int face_id;
        float uvx, uvy;
        float qx, qy, qz;  // the voxel's un-mirrored corner
    };
    __device__ __forceinline__ LeafSurface leaf_surface() const {
        LeafSurface f;
        const float ex = __builtin_fmaf(px + scale_exp2, tcx, -tbx);
        const float ey = __builtin_fmaf(py + scale_exp2, tcy, -tby);
        const float ez = __builtin_fmaf(pz + scale_exp2, tcz, -tbz);
        const float tc_min = gmax(gmax(ex, ey), ez);

        f.qx = px; f.qy = py; f.qz = pz;
        if (octant_mask & 1) f.qx = 3.0f - scale_exp2 - f.qx;
        if (octant_mask & 2) f.qy = 3.0f - scale_exp2 - f.qy;
        if (octant_mask & 4) f.qz = 3.0f - scale_exp2 - f.qz;

        const float inv_s = __uint_as_float(0x7f000000u - __float_as_uint(scale_exp2));  // exact 1/scale_exp2
        if (tc_min == ex) {
            f.face_id = int((__float_as_uint(rdx) >> 31) & 1u);
            f.uvx = (__builtin_fmaf(rdz, ex, roz) - f.qz) * inv_s;
            f.uvy = (__builtin_fmaf(rdy, ex, roy) - f.qy) * inv_s;
            if (rdx > 0.0f) f.uvx = 1.0f - f.uvx;
        } else if (tc_min == ey) {
            f.face_id = 2 | int((__float_as_uint(rdy) >> 31) & 1u);
            f.uvx = (__builtin_fmaf(rdx, ey, rox) - f.qx) * inv_s;
            f.uvy = (__builtin_fmaf(rdz, ey, roz) - f.qz) * inv_s;
            if (rdy > 0.0f) f.uvy = 1.0f - f.uvy;
        } else {
            f.face_id = 4 | int((__float_as_uint(rdz) >> 31) & 1u);
            f.uvx = (__builtin_fmaf(rdx, ez, rox) - f.qx) * inv_s;
            f.uvy = (__builtin_fmaf(rdy, ez, roy) - f.qy) * inv_s;
            if (rdz < 0.0f) f.uvx = 1.0f - f.uvx;
        }
        return f;
    }
    // the texture level of detail at the hit (svo.esvo.glsl:238-239) and the hit record minus its colour (svo.esvo.glsl:246-262)
    __device__ __forceinline__ static float leaf_lod(float dst) { return smoothstepf(15.0f, 25.0f, dst) * (dst - 15.0f) * 0.05f; }
    __device__ __forceinline__ void leaf_record(const DevScene& sc, const LeafSurface& f, uint32_t value, float dst, float tex_lod, Result& res) const {
        const float inv_scale = __uint_as_float(0x7f000000u - __float_as_uint(sc.octree_scale));  // 2^depth, exact
        res.t = dst;
        res.face_id = f.face_id;
        res.uv[0] = f.uvx; res.uv[1] = f.uvy;
        res.value = value;
        res.lod = tex_lod;
        const float hx = gmin(gmax(__builtin_fmaf(t_min, rdx, rox), f.qx + kEps), f.qx + scale_exp2 - kEps);
        const float hy = gmin(gmax(__builtin_fmaf(t_min, rdy, roy), f.qy + kEps), f.qy + scale_exp2 - kEps);
        const float hz = gmin(gmax(__builtin_fmaf(t_min, rdz, roz), f.qz + kEps), f.qz + scale_exp2 - kEps);
        res.pos[0] = (hx - 1.0f) * inv_scale;
        res.pos[1] = (hy - 1.0f) * inv_scale;
        res.pos[2] = (hz - 1.0f) * inv_scale;
        res.inside_voxel = inside_voxel();
    }

    // A voxel whose every texel of every face and mip level has alpha > 0 (the host's `opaque` set of block ids: vx_api.hip, opaque_blocks)
    // is a hit whatever the sample: the HIT phase is then the value and arithmetic -- no material row, no texels. The record is
    // leaf_test()'s except for its colour, which the caller samples when (and if: a shadow ray's is never looked at) it shades the hit,
    // with the same function on the same arguments (hit_color()).
    __device__ __forceinline__ void leaf_hit_opaque(const DevScene& sc, uint32_t value, Result& res) const {
        const LeafSurface f = leaf_surface();
        const float inv_scale = __uint_as_float(0x7f000000u - __float_as_uint(sc.octree_scale));
        const float dst = t_min * inv_scale;
        leaf_record(sc, f, value, dst, leaf_lod(dst), res);
    }

    // The HIT phase proper (svo.esvo.glsl:185-265) for a voxel of block `value`: true = the leaf is the result (res filled in); false = a
    // translucent leaf that is not, recorded as the last one passed -- the caller runs the ADVANCE half of the iteration.
    template <bool STATS = false>
    __device__ __forceinline__ bool leaf_test_value(const DevScene& sc, uint32_t value, bool cast_translucent, Result& res, Counters* ctr = nullptr) {
        const float octree_scale = sc.octree_scale;
        const float inv_scale = __uint_as_float(0x7f000000u - __float_as_uint(octree_scale));  // 2^depth, exact
        const LeafSurface f = leaf_surface();

        const vx_material mat = material_at(sc, value);
        int tex_id = mat.tex_side;
        if (f.face_id == 3) tex_id = mat.tex_top;
        else if (f.face_id == 2) tex_id = mat.tex_bottom;

        const float dst = t_min * inv_scale;
        const float tex_lod = leaf_lod(dst);
        if (STATS && tex_lod > 0.0f) ctr->leaf_tests_trilinear++;

        float tex_color[4];
        texture_lod(sc.tex, f.uvx, f.uvy, float(tex_id), tex_lod, tex_color);

        const bool first_of_kind = !(flags & kHasAdjacentLeaf) || value != last_leaf_value;
        if ((tex_color[3] > 0.0f || !cast_translucent) && first_of_kind) {
            leaf_record(sc, f, value, dst, tex_lod, res);
            res.color[0] = tex_color[0]; res.color[1] = tex_color[1]; res.color[2] = tex_color[2]; res.color[3] = tex_color[3];
            return true;
        }
        flags |= kHasAdjacentLeaf;
        last_leaf_value = value;
        return false;
    }

    // HIT phase for a ray whose step() returned kTravAtLeaf. kLeafHit: the leaf is the result (res filled in). Otherwise the translucent
    // leaf is recorded and the ADVANCE half of the iteration is run (svo.esvo.glsl:324 onwards).
    template <bool TRACE, bool STATS, class ST>
    __device__ __forceinline__ LeafOutcome leaf_test(const DevScene& sc, const ST& st, bool cast_translucent, Result& res, TracePtr tk,
                                                     Counters* ctr) {
        if (STATS) ctr->leaf_tests++;
        if (leaf_test_value<STATS>(sc, leaf_value(sc), cast_translucent, res, ctr)) return kLeafHit;
        const float tcrx = __builtin_fmaf(px, tcx, -tbx), tcry = __builtin_fmaf(py, tcy, -tby), tcrz = __builtin_fmaf(pz, tcz, -tbz);
        return advance<TRACE>(sc, st, tcrx, tcry, tcrz, gmin3(tcrx, tcry, tcrz), tk) ? kLeafPassed : kLeafPassedAndFinished;
    }
};

// ---- CSVO worlds on their image: a ray that is led into a voxel ------------------------------------------------------
//
// The image cursor `tr` stopped with kTravForeign: child `idx` of the node it examines (a voxel's parent, i.e. the image of a
// leaf-mask byte L of the world, svo.csvo.glsl:114-115) is a voxel the ray's origin lies in. The reference now PUSHes into the
// voxel: it takes the byte at L + 3 + popcount(L's mask below idx) for the voxel's "node" N0 and keeps going on whatever follows
// (phantom leaves with materials looked up through read_leaf included) until the ray steps out of the voxel. That walk only makes
// sense on the world's own bytes, so it is made there -- as a lean state machine of its own (round 4; until then the reference's
// whole byte cursor ran here, every node kind, a cursor of thirty fields copied in and out, a scratch-backed stack), for a walk
// that is 3.2 iterations long on average (profiles/round4/tools/excursion_stats.py: 58 % of the walks are PUSH + one ADVANCE that
// pops out again, 69 % never leave N0, 97 % never cross a phantom chunk boundary). What a walk inside a voxel can meet is much less
// than what the byte cursor can decode:
//   * N0 is read as a child mask whose children are all leaves (depth 0 < 2);
//   * below N0 the depth counter has wrapped (0 - 1 = 0xffffffff, svo.csvo.glsl:399): every node down there is an internal node --
//     u16 header, 1/2/4-byte offset table (svo.csvo.glsl:56-97) -- and none of its children is a leaf;
//   * depth never comes back to 2 or 3 in there, so `pre_leaf_pointer` is never touched; `material_section_ptr` only by a phantom chunk
//     boundary (a 4-byte table entry with bit 31 set) -- and a walk that crosses one is GIVEN UP (kTravForeign: the caller runs the
//     ray on the world's own bytes, with the reference's own cursor). So is a walk that would push below scale 0 (the child
//     index is read off the position's mantissa bits here) and, unless FULL_LEAF, one that meets a phantom leaf whose block is not in
//     the opaque set. Giving up is always correct; it only costs time.
// Without boundaries a node's depth is a function of its scale (N0: depth 0 at the parent's scale - 1, one less per level), so a stack
// entry inside the voxel is {byte pointer, t_max, 16-bit header}: it fits the image cursor's own LDS slots below the voxel's parent --
// which this ray's image cursor never uses -- 16-bit third plane included. The cursor's floats live in the image cursor `tr`
// throughout (nothing is copied), and the voxel's parent takes part as a byte node whose header is the image node's child mask.
// What a walk phase costs is its slowest lane. Measured (the timeline build's walk probe, 4K depth-14 frame, profiles/round4/pass_q): a wave
// makes 16 walk phases, 31 walkers each, whose loop makes 10 trips a phase -- a walk is 3 iterations long on average, but the slowest of 31
// decides -- and cycles = 3,700 a phase + 2,700 a trip: a trip is two or three memory accesses one after the other (table entry, child
// header, a stack slot below the LDS-resident levels), not its instructions -- with half as many waves on a CU it still takes 2,200.
// Tried and not kept: the voxels' N0 bytes kept beside the origin entry so that two walks in three read nothing of the world (no gain:
// the stragglers decide); nodes below N0 fetched as 16 bytes, a PUSH's table entry taken from them (3 % slower, pass_i); a cap on a
// shadow ray's walk, the capped rays run on the world's bytes at the end of the wave's life (what the cap saves the rerun costs: 4K
// depth 13 -2 %, 8K +2 %, pass_d / pass_q) or on the image again, walking together there (slower than on the bytes, pass_q).
// It ends when a POP brings the ray back to the voxel's parent or above: from there on every node is a real one again and the ray
// continues on the image (the stack slots at and above the parent's scale hold image entries, the walk only ever writes below them).
// One loop, whose iterations are the reference's iterations (`tr.iter` counts them; the iteration `tr` stopped in is the first one
// here: the caller took it back). `st` is a full stack (the walk can go below the LDS-resident levels).
// Returns kTravContinue (back on the image: `tr` is the cursor to go on with), kTravAtLeaf (`res` is the hit; OPAQUE: a phantom leaf
// whose value is in the host's set of blocks that are opaque throughout -- `opaque_lo/hi`, RenderParams -- is a hit without its sample,
// as in the kernel's own leaf tests, and `*color_pending` says that its colour is still to be sampled), kTravFinished (a miss) or
// kTravForeign (given up).
template <int IMGSVO, class ST, bool LIMIT = false, bool OPAQUE = false, bool FULL_LEAF = true>
__device__ __forceinline__ TravStatus walk_voxel_on_bytes(const DevScene& img, buf_t world, Trav<IMGSVO>& tr, const ST& st, bool cast_translucent,
                                                          Result& res, uint32_t opaque_lo = 0u, uint32_t opaque_hi = 0u, bool* color_pending = nullptr) {
    static_assert(!ST::kFast, "the walk needs the levels below the voxel");
    typedef Trav<IMGSVO> T;
    auto u32_at = [&](uint32_t p) -> uint32_t { return buf_u32(world, 8u + csvo_clamp(p)); };
    const int parent_scale = tr.scale;
    const uint32_t img_ptr = tr.ptr, img_node = tr.node;
    // the origin of the voxel-parent's octant, the unit in front of its values (unit `ptr`: traversal_image.hpp): [0] = byte pointer of L, [1] = k << 29 |
    // (L - material section), k = L's place among its depth-2 parent's leaf-mask bytes
    const uint64_t unit = uint64_t(tr.ptr);
    const uint32_t o0 = mem_u32(img.origin + unit * 8u), o1 = mem_u32(img.origin + unit * 8u + 4u);
    uint32_t bp = o0;  // the byte node the cursor examines: L first
    // its header in the 2-bits-per-child form (tag 01 per present child of a 1-bit level, csvo_header()): L's is the image node's child mask
    // (child c at bit 23 - c there)
    auto spread8 = [](uint32_t x) -> uint32_t {
        x = (x | (x << 4)) & 0x0f0fu;
        x = (x | (x << 2)) & 0x3333u;
        return (x | (x << 1)) & 0x5555u;
    };
    uint32_t hd = spread8((rev_bits32(img_node) >> 8) & 0xffu);
    if (tr.iter >= uint32_t(kMaxSteps)) return kTravFinished;
    // One iteration is ONE stretch of code for every lane: the PUSH's and the ADVANCE / POP's values are both worked out and the cursor takes one set
    // or the other by selects; only the memory operations (table entry, child header, stack slot) sit under their lanes' predicate, and a lane whose
    // walk ends notes how (`status`) and leaves at the bottom. Round 4's first form -- the reference's if / else if tree with a return wherever
    // a walk can end -- compiled to ~490 instructions an iteration, 170 of them mask bookkeeping and register copies at the joins
    // (profiles/round4/pass_q); a walk phase costs the iterations of its slowest lane, so that is what counts.
    TravStatus status = kTravContinue;
    for (;;) {
        if (LIMIT && tr.max_dst >= 0.0f && tr.t_min > tr.max_dst) return kTravFinished;
        ++tr.iter;
        const int dp = tr.scale - parent_scale + 1;  // the node's depth (svo.csvo.glsl:254): 1 = L, 0 = N0, below: wrapped
        const uint32_t oct = uint32_t(tr.idx_from_position() ^ tr.octant_mask);
        const float tcrx = __builtin_fmaf(tr.px, tr.tcx, -tr.tbx), tcry = __builtin_fmaf(tr.py, tr.tcy, -tr.tby), tcrz = __builtin_fmaf(tr.pz, tr.tcz, -tr.tbz);
        const float tc_max = gmin3(tcrx, tcry, tcrz);
        const uint32_t tag = (hd >> (oct * 2u)) & 3u;
        const bool is_child = tag != 0u, is_leaf = is_child && dp >= 0;
        const bool descend = is_child && tr.t_min <= tr.t_max;
        if (!descend) tr.flags &= ~T::kHasAdjacentLeaf;
        bool out = false;  // this lane's walk ends with this iteration
        const bool leaf_case = descend && is_leaf && tr.t_min > 0.0f;
        if (leaf_case) {
            // a phantom leaf (a child of N0: the voxel's parent is left for the image as soon as the cursor is back at it), svo.csvo.glsl:296-372
            const uint32_t value = csvo_read_leaf_at(world, o0 - (o1 & 0x1fffffffu), o0 - 3u - (o1 >> 29), bp, oct);
            bool hit = false;
            if constexpr (OPAQUE) {
                const uint32_t set = value < 32u ? opaque_lo : opaque_hi;
                if (value < 64u && ((set >> (value & 31u)) & 1u) != 0u && !(tr.flags & T::kHasAdjacentLeaf)) {
                    tr.leaf_hit_opaque(img, value, res);
                    *color_pending = true;
                    hit = true;
                }
            }
            if constexpr (FULL_LEAF) {
                if (!hit && tr.leaf_test_value(img, value, cast_translucent, res)) hit = true;
                // (else passed: a translucent leaf, recorded; the rest of the iteration is the ADVANCE)
            }
            if (hit) { status = kTravAtLeaf; out = true; }
            else if (!FULL_LEAF) { status = kTravForeign; out = true; }
        }
        const bool to_voxel = descend && !leaf_case;  // (a lane that is `out` already runs the rest of the iteration to no effect: it neither pushes nor pops)
        if (to_voxel && is_leaf && tr.t_min == 0.0f) tr.flags |= T::kInsideVoxel;
        const float tv_max = gmin(tr.t_max, tc_max);
        const bool push = to_voxel && tr.t_min <= tv_max;

        // ---- PUSH (svo.csvo.glsl:387-426): the child's node, the stack entry, the child cell ----
        const bool wrapped = dp < 0;
        const uint32_t offset = csvo_tag_bytes(hd & ((1u << (oct * 2u)) - 1u));
        const uint32_t table = bp + 2u;
        uint32_t word = 0u;
        if (push && wrapped) word = u32_at(table + offset);
        // (tag 0 -- no child, nothing is pushed and `e` is not looked at -- must still shift by less than 32: UBSan on the host harness, make sanitize)
        const uint32_t e = word & (0xffffffffu >> ((0x001018u >> (((tag - 1u) & 3u) * 8u)) & 0xffu));
        // out of L or N0: read_next_ptr's leaf-node case; below: the table entry's
        const uint32_t next = wrapped ? table + csvo_tag_bytes(hd) + e : bp + 3u + offset;
        // given up: below scale 0 the child index is no longer a mantissa bit; a phantom chunk boundary
        if (push && (tr.scale == 0 || (wrapped && (e & 0x80000000u) != 0u))) { status = kTravForeign; out = true; }
        // (written at every push, like the image cursor's: it keeps no `h` -- step_with)
        if (push) st.push(tr.scale, dp == 1 ? img_ptr : bp, tr.t_max, dp == 1 ? img_node : hd << 16);
        const float half = tr.scale_exp2 * 0.5f;
        const float tcenx = __builtin_fmaf(half, tr.tcx, tcrx), tceny = __builtin_fmaf(half, tr.tcy, tcry), tcenz = __builtin_fmaf(half, tr.tcz, tcrz);
        const float cx = tr.t_min < tcenx ? tr.px + half : tr.px, cy = tr.t_min < tceny ? tr.py + half : tr.py, cz = tr.t_min < tcenz ? tr.pz + half : tr.pz;
        // the child's header: N0's one byte spread to tags, a u16 below it
        uint32_t raw = 0u;
        if (push) raw = u32_at(next);
        const uint32_t child_hd = dp == 1 ? spread8(raw & 0xffu) : (raw & 0xffffu);

        // ---- ADVANCE, POP (svo.csvo.glsl:432-506; the child index is the position's bit at `scale`: see Trav::advance) ----
        const float ax = tc_max >= tcrx ? tr.scale_exp2 : 0.0f, ay = tc_max >= tcry ? tr.scale_exp2 : 0.0f, az = tc_max >= tcrz ? tr.scale_exp2 : 0.0f;
        const float sx = tr.px - ax, sy = tr.py - ay, sz = tr.pz - az;
        const uint32_t differing_bits = (__float_as_uint(sx + ax) ^ __float_as_uint(sx)) | (__float_as_uint(sy + ay) ^ __float_as_uint(sy)) |
                                        (__float_as_uint(sz + az) ^ __float_as_uint(sz));
        const bool pop = !push && !out && differing_bits >= (2u << tr.scale);
        const int up = 31 - __builtin_clz(differing_bits | 1u);
        const bool gone = pop && uint32_t(up) >= uint32_t(kMaxScale);
        if (gone) { status = kTravFinished; out = true; }
        uint32_t p = 0u, a = 0u;
        float popped_t_max = 0.0f;
        if (pop && !gone) st.pop(up, p, popped_t_max, a);
        const uint32_t keep = 0xffffffffu << (up & 31);
        if (pop && !gone && up >= parent_scale) {  // back among real nodes: the slot holds an image entry
            tr.ptr = p;
            tr.node = a;
            out = true;
        }
        // (the voxel's span was empty: a step to a sibling voxel, still at the voxel's parent -- ptr and node untouched)
        if (!push && !pop && dp == 1) out = true;  // (status: kTravContinue, or what the leaf test above has set)

        // ---- the cursor's next state ----
        tr.t_min = push ? tr.t_min : tc_max;
        tr.px = push ? cx : (pop ? __uint_as_float(__float_as_uint(sx) & keep) : sx);
        tr.py = push ? cy : (pop ? __uint_as_float(__float_as_uint(sy) & keep) : sy);
        tr.pz = push ? cz : (pop ? __uint_as_float(__float_as_uint(sz) & keep) : sz);
        tr.t_max = push ? tv_max : (pop ? popped_t_max : tr.t_max);
        tr.scale_exp2 = push ? half : (pop ? pow2i((up & 31) - kMaxScale) : tr.scale_exp2);
        tr.scale = push ? tr.scale - 1 : (pop ? up : tr.scale);
        bp = push ? next : (pop ? p : bp);
        hd = push ? child_hd : (pop ? a >> 16 : hd);
        if (!out && tr.iter >= uint32_t(kMaxSteps)) { status = kTravFinished; out = true; }
        if (out) break;
    }
    return status;
}

__device__ __forceinline__ void result_miss(Result& res, bool inside_voxel) {
    res.t = -1.0f;
    res.value = 0;
    res.face_id = 0;
    res.pos[0] = res.pos[1] = res.pos[2] = 0.0f;
    res.uv[0] = res.uv[1] = 0.0f;
    res.color[0] = res.color[1] = res.color[2] = res.color[3] = 0.0f;
    res.lod = 0.0f;
    res.inside_voxel = inside_voxel;
}

// Whole-ray form (picker and debug kernels; the v1 render kernel): step and test back to back.
template <int SVO, bool TRACE, bool STATS, bool LIMIT, class ST>
__device__ __forceinline__ void intersect(const DevScene& sc, const float ro_in[3], const float rd_in[3], float max_dst, bool cast_translucent,
                                          const ST& st, Result& res, uint32_t& steps, TracePtr tk, Counters* ctr) {
    if (STATS) ctr->rays++;
    Trav<SVO> tr;
    tr.template init<TRACE>(sc, ro_in, rd_in, max_dst, tk);
    for (;;) {
        TravStatus s = tr.template step<TRACE, STATS, LIMIT>(sc, st, tk, ctr);
        if (s == kTravAtLeaf) {
            const LeafOutcome o = tr.template leaf_test<TRACE, STATS>(sc, st, cast_translucent, res, tk, ctr);
            if (o == kLeafHit) break;
            s = o == kLeafPassed ? kTravContinue : kTravFinished;
        }
        if (s == kTravFinished) {
            result_miss(res, tr.inside_voxel());
            break;
        }
    }
    steps += tr.iter;
}

// ---- world.glsl -----------------------------------------------------------------------------------------------

// A pixel's place in the target and its value there. RGBA32F: the image2D of world.glsl:10 (row 0 = bottom). RGBA8: what
// Framebuffer::as_image makes of it (src/graphics/framebuffer.rs:97-111) -- glReadPixels(RGBA, UNSIGNED_BYTE), i.e. clamp to [0,1]
// and round to the nearest of 255 steps (NaN -> 0), rows flipped so that the top row comes first; tile lists keep their tile-local
// order in both formats.
__device__ __forceinline__ uint32_t image_index(const RenderParams& p, uint32_t x, uint32_t y) { return (p.rgba8 ? p.height - 1u - y : y) * p.width + x; }
__device__ __forceinline__ uint32_t pack_rgba8(const float c[4]) {
    uint32_t v = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float f = c[k];
        f = f != f ? 0.0f : (f < 0.0f ? 0.0f : (f > 1.0f ? 1.0f : f));
        v |= uint32_t(f * 255.0f + 0.5f) << (8 * k);
    }
    return v;
}
__device__ __forceinline__ void store_pixel(const RenderParams& p, float4* out, size_t index, const float c[4]) {
    if (p.rgba8) reinterpret_cast<uint32_t*>(out)[index] = pack_rgba8(c);
    else out[index] = make_float4(c[0], c[1], c[2], c[3]);
}

__device__ __forceinline__ float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void normalize3(const float v[3], float out[3]) {
    const float len = sqrtf(dot3(v, v));
    out[0] = v[0] / len; out[1] = v[1] / len; out[2] = v[2] / len;
}

// world.glsl:110-129
__device__ __forceinline__ void primary_ray(const RenderParams& p, uint32_t x, uint32_t y, float ro[3], float rd[3]) {
    float uvx = float(x) / float(p.width), uvy = float(y) / float(p.height);
    uvx = uvx * 2.0f - 1.0f;
    uvy = uvy * 2.0f - 1.0f;
    uvx *= p.u.aspect;
    uvx *= p.tan_half_fovy;
    uvy *= p.tan_half_fovy;
    const float* m = p.u.view;
    float d[3];
    if (p.affine_view) {
        // lw = 0*uvx + 0*uvy + 0*-1 + 1 is exactly 1 and x / 1 == x: the three divisions drop out
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            ro[r] = p.ray_origin[r];
            d[r] = (m[r] * uvx + m[4 + r] * uvy + m[8 + r] * -1.0f + m[12 + r] * 1.0f) - ro[r];
        }
    } else {
        const float lw = m[3] * uvx + m[7] * uvy + m[11] * -1.0f + m[15] * 1.0f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            ro[r] = p.ray_origin[r];
            const float l = (m[r] * uvx + m[4 + r] * uvy + m[8 + r] * -1.0f + m[12 + r] * 1.0f) / lw;
            d[r] = l - ro[r];
        }
    }
    normalize3(d, rd);
}

// world.glsl:92-108
__device__ __forceinline__ void sky_color(const float rd[3], float out[3]) {
    const float SKY[3] = {135.0f / 255.0f, 206.0f / 255.0f, 235.0f / 255.0f};
    const float flat[3] = {rd[0], 0.0f, rd[2]};
    float p[3];
    normalize3(flat, p);
    // argument clamped like the oracle does: the reference's expected image has no undefined (acos(1+)) horizon pixels
    const float a = sky_acos(gclamp(dot3(rd, p) / fabsf(sqrtf(dot3(rd, rd))) * fabsf(sqrtf(dot3(p, p))), -1.0f, 1.0f));
    float grad = a / 1.570796f;
    const float g1 = 1.0f - grad;
    grad = 1.0f - g1 * g1 * g1;  // pow(x, 3.0): two multiplications are within an ulp of any pow() and an order of magnitude cheaper
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float horizon = 1.0f * (1.0f - 0.3f) + SKY[k] * 0.3f;
        out[k] = horizon * (1.0f - grad) + SKY[k] * grad;
    }
}

// A face's normal, tangent and bitangent (svo.glsl:2-9, 12-19, 22-29): components of -1, 0 or 1, two bits each (1 = 1, 2 = -1), the six faces of a component in
// one 12-bit constant -- arithmetic on the face's number instead of three look-ups in constant memory, each a round trip of its own in the middle of a
// shading phase (profiles/round4/pass_t).
constexpr int kFaceNormalsI[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
constexpr int kFaceTangentsI[6][3] = {{0, 0, 1}, {0, 0, -1}, {1, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {1, 0, 0}};
constexpr int kFaceBitangentsI[6][3] = {{0, 1, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 1}, {0, 1, 0}, {0, 1, 0}};
constexpr uint32_t face_pack(const int (&t)[6][3], int k) {
    uint32_t v = 0;
    for (int f = 0; f < 6; ++f) v |= uint32_t(t[f][k] == 1 ? 1 : (t[f][k] == -1 ? 2 : 0)) << (2 * f);
    return v;
}
template <int TABLE>  // 0 normals, 1 tangents, 2 bitangents
__device__ __forceinline__ void face_vector(uint32_t face, float out[3]) {
    constexpr uint32_t px = TABLE == 0 ? face_pack(kFaceNormalsI, 0) : (TABLE == 1 ? face_pack(kFaceTangentsI, 0) : face_pack(kFaceBitangentsI, 0));
    constexpr uint32_t py = TABLE == 0 ? face_pack(kFaceNormalsI, 1) : (TABLE == 1 ? face_pack(kFaceTangentsI, 1) : face_pack(kFaceBitangentsI, 1));
    constexpr uint32_t pz = TABLE == 0 ? face_pack(kFaceNormalsI, 2) : (TABLE == 1 ? face_pack(kFaceTangentsI, 2) : face_pack(kFaceBitangentsI, 2));
    const uint32_t sh = (face < 6u ? face : 5u) * 2u;
    const uint32_t cx = (px >> sh) & 3u, cy = (py >> sh) & 3u, cz = (pz >> sh) & 3u;
    out[0] = float(int(cx & 1u) - int(cx >> 1)); out[1] = float(int(cy & 1u) - int(cy >> 1)); out[2] = float(int(cz & 1u) - int(cz >> 1));
}

// world.glsl:87-88
__device__ __forceinline__ void apply_light(const RenderParams& p, float color[4], float ds, float shadow) {
    const float light = gclamp(p.u.ambient + ds * shadow, 0.0f, 1.0f);
    color[0] *= light; color[1] *= light; color[2] *= light;
}

// What trace_ray does with a finished PRIMARY ray (world.glsl:31-84), split from the traversal so that a wavefront
// can run it for several lanes at once. Outcome: either the pixel's final colour, or "cast this shadow ray" plus
// the two values needed after it (surface colour and diffuse+specular).
struct PrimaryOutcome {
    bool final_color;     // color[] is the pixel's value; no shadow ray
    float color[4];       // final colour, or the surface colour to be lit
    float ds;             // diffuse + specular
    float shadow_origin[3];
    uint32_t flags;       // vx_hit flags accumulated so far
};

// COLOR_PENDING (a hit accepted by Trav::leaf_hit_opaque): res.color is not there yet -- it is sampled here, where the hit's material row is
// at hand anyway, with leaf_test()'s own arguments: the face's texture at (uv, lod).
template <bool COLOR_PENDING = false>
__device__ __forceinline__ void shade_primary(const DevScene& sc, const RenderParams& p, const Result& res, PrimaryOutcome& o, bool color_pending = false) {
    o.flags = res.t != -1.0f ? 1u : 0u;
    o.final_color = true;
    o.ds = 0.0f;
    o.color[0] = o.color[1] = o.color[2] = o.color[3] = 0.0f;
    o.shadow_origin[0] = o.shadow_origin[1] = o.shadow_origin[2] = 0.0f;
    if (res.t < 0.0f) return;  // miss: the caller paints the sky

    if (floorf(res.pos[0]) == floorf(p.u.highlight_pos[0]) && floorf(res.pos[1]) == floorf(p.u.highlight_pos[1]) &&
        floorf(res.pos[2]) == floorf(p.u.highlight_pos[2])) {
        const float lx = fabsf(res.uv[0] - 0.5f) * 2.0f, ly = fabsf(res.uv[1] - 0.5f) * 2.0f;
        if (gmax(lx, ly) > 1.0f - 1.0f / 16.0f) {
            o.color[0] = o.color[1] = o.color[2] = o.color[3] = 1.0f;
            o.flags |= 8u;
            return;
        }
    }

    const TexLod levels = texture_levels(sc.tex, res.lod);  // (its two words requested beside the material's row)
    const vx_material mat = material_at(sc, res.value);
    int tex_normal_id = mat.tex_side_normal;
    if (res.face_id == 3) tex_normal_id = mat.tex_top_normal;
    else if (res.face_id == 2) tex_normal_id = mat.tex_bottom_normal;

    float normal[3];
    face_vector<0>(uint32_t(res.face_id), normal);
    // the normal map's sample and -- where the hit was found without it (an opaque block) -- the colour's, side by side (texture_lod_pair)
    int tex_id = mat.tex_side;
    if (res.face_id == 3) tex_id = mat.tex_top;
    else if (res.face_id == 2) tex_id = mat.tex_bottom;
    const bool want_color = COLOR_PENDING && color_pending;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f}, sampled_color[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (tex_normal_id != -1 || want_color)
        texture_lod_pair(sc.tex, levels, res.uv[0], res.uv[1], float(tex_normal_id), float(tex_id), tex_normal_id != -1, want_color, s, sampled_color);
    if (tex_normal_id != -1) {
        const float tex[3] = {s[0] * 2.0f - 1.0f, s[2] * 2.0f - 1.0f, s[1] * 2.0f - 1.0f};  // .xzy
        float n[3];
        normalize3(tex, n);
        const float base[3] = {normal[0], normal[1], normal[2]};
        float tangent[3], bitangent[3];
        face_vector<1>(uint32_t(res.face_id), tangent);
        face_vector<2>(uint32_t(res.face_id), bitangent);
#pragma unroll
        for (int k = 0; k < 3; ++k) normal[k] = n[0] * tangent[k] + n[1] * base[k] + n[2] * bitangent[k];
    }

    const float neg_l[3] = {-p.u.light_dir[0], -p.u.light_dir[1], -p.u.light_dir[2]};
    const float diffuse = gmax(dot3(normal, neg_l), 0.0f);
    const float vd[3] = {res.pos[0] - p.u.cam_pos[0], res.pos[1] - p.u.cam_pos[1], res.pos[2] - p.u.cam_pos[2]};
    float view_dir[3];
    normalize3(vd, view_dir);
    const float dn = dot3(normal, neg_l);
    float reflect_dir[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) reflect_dir[k] = neg_l[k] - 2.0f * dn * normal[k];
    const float specular = glsl_pow(gmax(dot3(view_dir, reflect_dir), 0.0f), mat.specular_pow) * mat.specular_strength;

    o.ds = diffuse + specular;
    o.color[0] = res.color[0]; o.color[1] = res.color[1]; o.color[2] = res.color[2]; o.color[3] = res.color[3];
    if (want_color) { o.color[0] = sampled_color[0]; o.color[1] = sampled_color[1]; o.color[2] = sampled_color[2]; o.color[3] = sampled_color[3]; }
    if (p.u.render_shadows && res.t < p.u.shadow_distance) {
        o.final_color = false;
        o.flags |= 2u;
        o.shadow_origin[0] = res.pos[0] + normal[0] * 0.001f;
        o.shadow_origin[1] = res.pos[1] + normal[1] * 0.001f;
        o.shadow_origin[2] = res.pos[2] + normal[2] * 0.001f;
    } else {
        apply_light(p, o.color, o.ds, 1.0f);
    }
}

// trace_ray + sky (world.glsl:27-90, 132-138) for one pixel
template <int SVO, bool STATS, class ST>
__device__ __forceinline__ void shade_pixel(const DevScene& sc, const RenderParams& p, uint32_t x, uint32_t y, const ST& st, float color[4],
                                            vx_hit* rec, Counters* ctr, uint32_t* lit, uint32_t* shadow_rays) {
    float ro[3], rd[3];
    primary_ray(p, x, y, ro, rd);

    Result res;
    uint32_t steps = 0;
    intersect<SVO, false, STATS, false>(sc, ro, rd, -1.0f, true, st, res, steps, nullptr, ctr);

    const bool hit = res.t != -1.0f;
    uint32_t flags = hit ? 1u : 0u;
    float shadow_t = -1.0f;
    color[0] = color[1] = color[2] = color[3] = 0.0f;

    bool done = res.t < 0.0f;
    if (!done) {
        if (floorf(res.pos[0]) == floorf(p.u.highlight_pos[0]) && floorf(res.pos[1]) == floorf(p.u.highlight_pos[1]) &&
            floorf(res.pos[2]) == floorf(p.u.highlight_pos[2])) {
            const float lx = fabsf(res.uv[0] - 0.5f) * 2.0f, ly = fabsf(res.uv[1] - 0.5f) * 2.0f;
            if (gmax(lx, ly) > 1.0f - 1.0f / 16.0f) {
                color[0] = color[1] = color[2] = color[3] = 1.0f;
                flags |= 8u;
                done = true;
            }
        }
    }
    if (!done) {
        if (STATS) ++*lit;
        const vx_material mat = material_at(sc, res.value);
        int tex_normal_id = mat.tex_side_normal;
        if (res.face_id == 3) tex_normal_id = mat.tex_top_normal;
        else if (res.face_id == 2) tex_normal_id = mat.tex_bottom_normal;

        float normal[3];
    face_vector<0>(uint32_t(res.face_id), normal);
        if (tex_normal_id != -1) {
            float s[4];
            texture_lod(sc.tex, res.uv[0], res.uv[1], float(tex_normal_id), res.lod, s);
            const float tex[3] = {s[0] * 2.0f - 1.0f, s[2] * 2.0f - 1.0f, s[1] * 2.0f - 1.0f};  // .xzy
            float n[3];
            normalize3(tex, n);
            const float base[3] = {normal[0], normal[1], normal[2]};
            float tangent[3], bitangent[3];
            face_vector<1>(uint32_t(res.face_id), tangent);
            face_vector<2>(uint32_t(res.face_id), bitangent);
#pragma unroll
            for (int k = 0; k < 3; ++k) normal[k] = n[0] * tangent[k] + n[1] * base[k] + n[2] * bitangent[k];
        }

        const float neg_l[3] = {-p.u.light_dir[0], -p.u.light_dir[1], -p.u.light_dir[2]};
        const float diffuse = gmax(dot3(normal, neg_l), 0.0f);

        const float vd[3] = {res.pos[0] - p.u.cam_pos[0], res.pos[1] - p.u.cam_pos[1], res.pos[2] - p.u.cam_pos[2]};
        float view_dir[3];
        normalize3(vd, view_dir);
        const float dn = dot3(normal, neg_l);
        float reflect_dir[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) reflect_dir[k] = neg_l[k] - 2.0f * dn * normal[k];
        const float specular = glsl_pow(gmax(dot3(view_dir, reflect_dir), 0.0f), mat.specular_pow) * mat.specular_strength;

        float shadow = 1.0f;
        if (p.u.render_shadows && res.t < p.u.shadow_distance) {
            const float so[3] = {res.pos[0] + normal[0] * 0.001f, res.pos[1] + normal[1] * 0.001f, res.pos[2] + normal[2] * 0.001f};
            Result sres;
            if (STATS) ++*shadow_rays;
            intersect<SVO, false, STATS, false>(sc, so, neg_l, -1.0f, true, st, sres, steps, nullptr, ctr);
            shadow = sres.t < 0.0f ? 1.0f : 0.0f;
            flags |= 2u;
            if (!(sres.t < 0.0f)) flags |= 4u;
            shadow_t = sres.t;
        }

        const float light = gclamp(p.u.ambient + (diffuse + specular) * shadow, 0.0f, 1.0f);
        color[0] = res.color[0] * light;
        color[1] = res.color[1] * light;
        color[2] = res.color[2] * light;
        color[3] = res.color[3];
    }

    if (!hit) {
        float sky[3];
        sky_color(rd, sky);
        color[0] = sky[0]; color[1] = sky[1]; color[2] = sky[2]; color[3] = 1.0f;
    }

    if (rec) {
        rec->t = res.t;
        rec->value = res.value;
        rec->face_id = res.face_id;
        rec->flags = flags;
        rec->pos[0] = res.pos[0]; rec->pos[1] = res.pos[1]; rec->pos[2] = res.pos[2];
        rec->lod = res.lod;
        rec->uv[0] = res.uv[0]; rec->uv[1] = res.uv[1];
        rec->shadow_t = shadow_t;
        rec->steps = steps;
    }
}

}  // namespace vxd
