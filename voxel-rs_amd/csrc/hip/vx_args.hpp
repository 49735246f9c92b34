// What the host runtime (runtime.cpp, comm.cpp) and the kernels (kernels_render.hip, kernels_aux.hip) share: the argument blocks a
// launch passes, the units of the screen and of the sub-tile queue, the node formats' internal ids. Plain data, no device code: the
// runtime is ordinary C++ and never sees a kernel's body.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "voxel_hip.h"

#if defined(__HIPCC__)
#define VX_HOST_DEVICE __host__ __device__
#else
#define VX_HOST_DEVICE
#endif

// third node format, internal to the library: the traversal image of a world (traversal_image.hpp, kOct64)
#define VX_SVO_IMAGE 3
// the same with octant INDICES for pointers and 64-bit addressing: images beyond 4 GiB (traversal_image.hpp, kOct64Wide)
#define VX_SVO_IMAGE_WIDE 4
// an ESVO world buffer of 4 GiB and more: the reference's format unchanged (descriptors[] indices are 32 bits, esvo.rs:74-101),
// read through a 64-bit pointer with an explicit range check instead of the V#'s
#define VX_SVO_ESVO_BIG 5

namespace vxd {

constexpr int kMaxSteps = 1000;       // svo.esvo.glsl:18
constexpr int kMaxScale = 23;         // svo.esvo.glsl:21
constexpr int kLdsLevels = 13;        // per-ray stack levels resident in LDS by default: three u32 planes, 9.75 KB per wave, 16 waves per CU

// what the host passes to a kernel; expanded into a DevScene (descriptors in SGPRs) at kernel entry
struct SceneArgs {
    const uint8_t* world;
    uint64_t world_bytes;
    const vx_material* materials;
    uint32_t n_materials;
    const uint8_t* tex;
    uint32_t tex_bytes;
    uint32_t width, height, layers, levels;
    uint32_t level_offset[16];
    const uint8_t* image;   // the traversal image of the world (traversal_image.hpp), or null
    uint64_t image_bytes;
    const uint8_t* origin;  // CSVO worlds: the image again (the origin of a voxel-parent octant is the unit in front of its values), else null
};

struct RenderParams {
    vx_uniforms u;
    float tan_half_fovy;   // tanf(fovy * 0.5f), evaluated on the host (world.glsl:115)
    float ray_origin[3];   // (view * vec4(0,0,0,1)).xyz / .w, the same for every pixel: evaluated on the host (world.glsl:118)
    uint32_t affine_view;  // the view matrix's last row is (0,0,0,1): the per-pixel perspective divide is a division by exactly 1
    uint32_t width, height;
    uint32_t tiles_x, tiles_y;
    // How the queue's tile numbers i lie in the launch's list of n_local_tiles tiles (a whole image: its tiles in row-major order) -- 0: number =
    // place; 1 (whole images): strips of strip_w columns one after the other, a strip's tiles along its rows (1: down the columns); 2: place (i * tile_stride) % n_local_tiles,
    // tile_stride prime to the number of tiles and about 0.618 of it -- consecutive numbers lie far apart in both directions (the tile at place
    // t has number (t * tile_stride_inv) % n_local_tiles). What a frame costs depends on the ORDER its tiles are handed out in: along the rows
    // all waves are in the sky first and on the ground last -- the memory side idles, then queues -- and the frame ends on its dearest tiles
    // (profiles/round4/pass_r: C3 0.261 -> 0.239 ms with frames in flight, 0.367 -> 0.348 one at a time, for either of 1 and 2).
    uint32_t tile_numbering, tile_stride, tile_stride_inv, strip_w;
    uint32_t tile_rank, tile_count, n_local_tiles;
    // ... as two tables the HOST makes once per (image size, rank, numbering) with tile_place / tile_number below (device memory, n_local_tiles entries
    // each): the kernel's refill reads one entry per sub-tile instead of working the place, the Morton look-up and two divisions by tiles_x out
    // (a quarter of a refill's instructions, profiles/round5). tile_table[number] = {tile x | tile y << 16, place in the launch's list};
    // number_of_place[place] = the tile's queue number (the cost notes).
    const uint2* tile_table;
    const uint32_t* number_of_place;
    // screen sharding (tile_count > 1): the image's 32x32 tiles in Morton order of their (x, y) -- this context renders the tiles
    // tile_order[k * tile_count + tile_rank], k = 0 .. n_local_tiles - 1 (device memory; null when the whole image is rendered)
    const uint32_t* tile_order;
    uint32_t rgba8;  // the target holds RGBA8 pixels (vx_target.format): 4 bytes each, and a whole image has its TOP row first
    // block ids 0..63 whose textures -- all three faces, every texel of every mip level -- have alpha > 0: a voxel of such a block is a hit
    // whatever the sample (Trav::leaf_hit_opaque). The host derives it from the material rows and the mip chain (runtime.cpp).
    uint32_t opaque_lo, opaque_hi;
};

// (view * vec4(0,0,0,1)).xyz / .w (world.glsl:118), in the operation order primary_ray uses for the look-at point
VX_HOST_DEVICE inline void view_origin(const float* m, float ro[3]) {
    const float ow = m[3] * 0.0f + m[7] * 0.0f + m[11] * 0.0f + m[15] * 1.0f;
    for (int r = 0; r < 3; ++r) ro[r] = (m[r] * 0.0f + m[4 + r] * 0.0f + m[8 + r] * 0.0f + m[12 + r] * 1.0f) / ow;
}

// RenderParams::tile_numbering, both ways: the place in the launch's list of the tile with queue number `number`, and back.
VX_HOST_DEVICE inline uint32_t tile_place(const RenderParams& p, uint32_t number) {
    if (p.tile_numbering == 2u) return uint32_t((uint64_t(number) * p.tile_stride) % p.n_local_tiles);
    if (p.tile_numbering == 1u && p.tile_count <= 1) {
        // strips of strip_w columns, the tiles of a strip along its rows (the last strip is narrower)
        const uint32_t per_strip = p.strip_w * p.tiles_y, full = p.tiles_x / p.strip_w;
        const uint32_t strip = number / per_strip < full ? number / per_strip : full;
        const uint32_t first = strip * p.strip_w, wide = p.tiles_x - first < p.strip_w ? p.tiles_x - first : p.strip_w;
        const uint32_t in = number - first * p.tiles_y;
        return (in / wide) * p.tiles_x + first + in % wide;
    }
    return number;
}
VX_HOST_DEVICE inline uint32_t tile_number(const RenderParams& p, uint32_t place) {
    if (p.tile_numbering == 2u) return uint32_t((uint64_t(place) * p.tile_stride_inv) % p.n_local_tiles);
    if (p.tile_numbering == 1u && p.tile_count <= 1) {
        const uint32_t tcol = place % p.tiles_x, trow = place / p.tiles_x, strip = tcol / p.strip_w;
        const uint32_t first = strip * p.strip_w, wide = p.tiles_x - first < p.strip_w ? p.tiles_x - first : p.strip_w;
        return first * p.tiles_y + trow * wide + (tcol - first);
    }
    return place;
}
// ... and what launch_render puts there: strips of `strip` columns for a whole image, the stride for a tile list (which has no columns) -- g prime to
// the n tiles and about 0.618 n, and its inverse modulo n
inline void set_tile_numbering(RenderParams& p, int numbering, int strip) {
    p.tile_numbering = uint32_t(numbering < 0 || numbering > 2 ? 0 : numbering);
    if (p.tile_numbering == 1u && p.tile_count > 1) p.tile_numbering = 2u;
    p.tile_stride = p.tile_stride_inv = 1;
    p.strip_w = strip > 0 ? uint32_t(strip) : 1u;
    if (p.strip_w > p.tiles_x) p.strip_w = p.tiles_x ? p.tiles_x : 1u;
    if (p.tile_numbering == 2u && p.n_local_tiles < 3) p.tile_numbering = 0u;
    if (p.tile_numbering != 2u) return;
    const uint64_t n = p.n_local_tiles;
    uint64_t g = uint64_t(double(n) * 0.6180339887498949);
    if (g < 1) g = 1;
    for (;; ++g) {  // (n - 1 is prime to n: the search ends)
        uint64_t a = g, b = n;
        while (b) { const uint64_t t = a % b; a = b; b = t; }
        if (a == 1) break;
    }
    long long t0 = 0, t1 = 1, r0 = (long long)n, r1 = (long long)g;  // extended Euclid: t0 = 1 / g modulo n
    while (r1 > 0) { const long long q = r0 / r1; long long t = t0 - q * t1; t0 = t1; t1 = t; t = r0 - q * r1; r0 = r1; r1 = t; }
    if (t0 < 0) t0 += (long long)n;
    p.tile_stride = uint32_t(g);
    p.tile_stride_inv = uint32_t(t0);
}

}  // namespace vxd

namespace vxk {

constexpr uint32_t kTile = 32;         // multi-GPU sharding unit (pixels per edge)
constexpr uint32_t kBlockEdge = 16;    // the one-thread-per-pixel kernel: a 256-thread workgroup shades 16x16 pixels, 4 waves x (8x8)
constexpr uint32_t kBlockThreads = 256;

constexpr uint32_t kQueues = 8, kQueueStride = 64;  // dispensers of the sub-tile queue, words between them
// The k-th sub-tile of dispenser c is sub-tile ((k / S) * 8 + c) * S + k % S: the launch's sub-tiles in stretches of S (`stripe`), dealt
// out to the dispensers in turn. Workgroups b and b + 8 run on the same XCD and wave b draws from dispenser b & 7, so with long stretches (a
// band of the screen; a compact piece of a tile list's Morton order) what an XCD's waves traverse is the part of the world behind its own
// stretches, and that is what its L2 holds -- not, eight times over, the nodes behind the whole screen, as with S = 1 (round 3: sub-tiles c,
// c + 8, c + 16, ... to dispenser c). S = 1 is still what a cost-ordered launch uses: its tickets are places in a table sorted by cost.
// (`shift`: log2 of the stretch where it is a power of two -- the default 16 is --, else 0xffffffff: a 32-bit division is ~35 instructions, per ticket)
VX_HOST_DEVICE inline uint32_t queue_subtile(uint32_t k, uint32_t c, uint32_t stripe, uint32_t shift = 0xffffffffu) {
    const uint32_t q = shift < 32u ? k >> shift : k / stripe;  // (once per sub-tile and wave)
    return (q * kQueues + c) * stripe + (k - q * stripe);
}

struct PersistentArgs {
    // The sub-tile queue: eight dispensers (kQueueStride words apart: one memory-side atomic unit each), dispenser c hands out the sub-tiles
    // queue_subtile(k, c) beyond its waves' first ones (wave w = 8 j + c starts on dispenser c's j-th sub-tile without asking). A
    // wave draws from dispenser (w & 7) and, when that one is empty, from the next. One dispenser for 4096 waves is 70 M atomic adds per second on one address:
    // more than the memory side carries out there -- a ticket took tens of microseconds. A stream has two sets: a launch uses one and
    // clears the other for its successor (which does not start before this one has ended).
    uint32_t* work_counter;   // this launch's set
    uint32_t* next_counter;   // the set to clear
    uint32_t total_subtiles;  // n_local_tiles * 16
    uint32_t refill_min, service_min;
    uint32_t stripe;          // the length of the stretches the sub-tiles are dealt out to the dispensers in (queue_subtile), at least 1
    uint32_t stripe_shift;    // log2(stripe) where that is a power of two, else 0xffffffff
    // Expensive sub-tiles first. A ray is a chain of dependent steps -- about 0.8 us per iteration on a busy device -- so a frame cannot
    // end before its longest rays do (up to ~300 iterations against a mean of ~30): handed out in the order of their numbers they start in mid-frame
    // and the frame ends with a long tail of waves that wait for a few of them (profiles/timeline.py). So every ray that ends notes
    // its iteration count in its sub-tile's entry of `cost_cur` (atomic max), a small kernel behind the frame sorts the sub-tiles into
    // sixteen cost classes, most expensive first, the order of their numbers within a class (order_kernel), and the NEXT frame of the same view on this
    // stream draws its tickets through that table: `order` (null: the order of their numbers). Order only: no pixel's value depends on it.
    const uint32_t* order;          // [total_subtiles] sub-tile ids, or null
    uint32_t* cost_cur;             // [total_subtiles] tag << 12 | iterations of the sub-tile's longest ray this frame; null = do not note
    uint32_t cur_tag;               // frame tag (20 bits, never 0): entries with another tag are stale (no clearing between frames)
    uint32_t ticket_ahead;          // 1 + g: waves draw their next sub-tile's ticket when they start on one (its round trip runs under the traversal),
                                    // except for the frame's last g quarter-grids of tickets
    uint32_t timeline_part;         // measurement: which part of the service phases the timeline's tick count covers (0 all, 1 leaf tests, 2 finished rays, 3 refill, 4 ray set-up, 5 walks inside voxels)
    unsigned long long* timeline;   // measurement (the timeline build of the library), else null: per wave {start, queue found empty, exit} in 10 ns ticks, pixels taken
    unsigned long long* excursions;  // counted on request only (vx_excursion_counters), else null: [0] rays that walked inside a voxel on the world's bytes, [1] of which were given up and run on the bytes, [2] service phases that ran such walks, [3] loop iterations made on the bytes
};

// Images of CSVO worlds: what a wave cannot finish on the image goes on a list of its own and is run on the world's own bytes once the
// tile queue is empty and the wave's rays are done (a second phase of the same kernel, not a second kernel: its registers overlay the
// first phase's, and a frame stays one command).
struct PixelList {
    // Chunks that the wave chains together, from a ring (`mask` + 1 of them, a power of two) through one counter that only ever grows; a
    // wave touches nothing but its own chunks, so no wave ever waits for another.
    //   renders with hit records: PIXELS -- chunks of 128 dwords: [0] previous chunk + 1 (0 = none), [1] entries, [2..127] out_index values
    //   image-only renders: RAYS -- chunks of 1024 dwords: [0] previous chunk + 1, [1] records, from dword 16 on up to 64 records of 12
    //   dwords {pixel | shadow << 31, origin in octree space, colour to be lit, diffuse + specular}
    uint32_t* chunks;
    uint32_t* next_chunk;
    uint32_t mask;
};
constexpr uint32_t kChunkDwords = 128, kChunkEntries = 126;
constexpr uint32_t kRayChunkDwords = 1024, kRayChunkRecords = 64, kRayRecordDwords = 12, kRayChunkHeader = 16;
// render_persistent's FOREIGN: 0 = no ray of this image is ever led into a voxel's own bytes (ESVO worlds), VX_SVO_CSVO = such rays walk the
// voxel on the world's bytes in a service phase (walk_voxel_on_bytes), kForeignRerun = they are listed and run on the bytes afterwards
constexpr int kForeignRerun = 3;

constexpr uint32_t kCostClasses = 16, kCostStep = 16;  // classes of the order table: iterations / kCostStep, capped
constexpr uint32_t kCostFloor = 64;  // rays that end sooner (the mean is about 30; one in eight gets here) leave their sub-tile in the cheapest class: no note
constexpr uint32_t kOrderThreads = 1024;

struct TraceArgs {
    float pos[3], dir[3];
    float max_dst;
    int cast_translucent;
};

}  // namespace vxk
