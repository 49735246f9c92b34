// libvoxelhip.so: kernels and the host runtime behind include/voxel_hip.h.
//
// Replaces the OpenGL side of the reference's graphics::Svo (src/graphics/svo.rs:56-256): persistently mapped
// SSBO -> pinned staging + hipMalloc'd world buffer with range uploads; glDispatchCompute(world.glsl) ->
// render kernel; glDispatchCompute(picker.glsl) -> picker kernel; glFenceSync/glClientWaitSync -> HIP events.
// There is no CPU path: without a HIP device every entry point fails with VX_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: RCCL itself is opened at run time by vx_comm_init (a single-GPU deployment needs none)

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "traversal_image.hpp"
#include "voxel_hip.h"
#include "vx_device.hpp"
#include "vx_loop_gfx950.hpp"

// 1 = render_persistent's traversal loop on a byte-offset image whose depth the LDS-resident stack covers is the hand-scheduled one
// (vx_loop_gfx950.hpp); 0 = the compiler's loop everywhere (A/B builds)
#ifndef VX_ASM_LOOP
#define VX_ASM_LOOP 1
#endif
// the same for the 12-level stack of the five-waves build (VX_FIVE_WAVES=1, an experiment)
#ifndef VX_ASM_LOOP_12
#define VX_ASM_LOOP_12 1
#endif

using namespace vxd;

// =================================================================================================================
// kernels
// =================================================================================================================

namespace {

constexpr uint32_t kTile = 32;         // multi-GPU sharding unit (pixels per edge)
constexpr uint32_t kBlockEdge = 16;    // one 256-thread workgroup shades 16x16 pixels: 4 waves x (8x8)
constexpr uint32_t kBlockThreads = 256;

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2). Remap so that each XCD shades a
// contiguous run of screen blocks and its L2 keeps that region's octree nodes (speed only, never correctness).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n) {
    constexpr uint32_t X = 8;
    const uint32_t per = n / X, rem = n % X;
    const uint32_t xcd = b % X, slot = b / X;
    // XCDs [0, rem) own per+1 blocks, the rest own per
    const uint32_t start = xcd * per + (xcd < rem ? xcd : rem);
    return start + slot;
}

// morton decode of the low 6 bits into (x, y) in [0,8)
__device__ __forceinline__ void lane_to_xy(uint32_t lane, uint32_t& x, uint32_t& y) {
    x = (lane & 1u) | ((lane >> 1) & 2u) | ((lane >> 2) & 4u);
    y = ((lane >> 1) & 1u) | ((lane >> 2) & 2u) | ((lane >> 3) & 4u);
}

template <int SVO, bool HITS, bool STATS>
__global__ __launch_bounds__(kBlockThreads) void render_kernel(SceneArgs sa, RenderParams p, float4* __restrict__ out, vx_hit* __restrict__ hits,
                                                               unsigned long long* __restrict__ counters) {
    const DevScene sc = make_scene(sa);
    const uint32_t tid = threadIdx.x;
    StackSpill spill;
    Stack<kBlockThreads> st;
    st.init(tid, &spill);

    // block -> (local tile, 16x16 sub-block) -> pixel
    const uint32_t b = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t local_tile = b >> 2, sub = b & 3u;
    const uint32_t seq = local_tile * p.tile_count + p.tile_rank;
    const bool tile_valid = seq < p.tiles_x * p.tiles_y;
    const uint32_t tile = p.tile_count > 1 ? (tile_valid ? p.tile_order[seq] : 0u) : seq;
    const uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const uint32_t wave = tid >> 6, lane = tid & 63u;
    uint32_t lx, ly;
    lane_to_xy(lane, lx, ly);
    const uint32_t in_x = (sub & 1u) * kBlockEdge + (wave & 1u) * 8 + lx;  // position inside the 32x32 tile
    const uint32_t in_y = (sub >> 1) * kBlockEdge + (wave >> 1) * 8 + ly;
    const uint32_t x = tx * kTile + in_x, y = ty * kTile + in_y;
    const bool active = tile_valid && x < p.width && y < p.height;

    Counters ctr = {};
    uint32_t lit = 0, shadow_rays = 0;
    if (active) {
        float color[4];
        vx_hit rec;
        shade_pixel<SVO, STATS>(sc, p, x, y, st, color, HITS ? &rec : nullptr, STATS ? &ctr : nullptr, &lit, &shadow_rays);
        const size_t index = p.tile_count > 1 ? size_t(local_tile) * (kTile * kTile) + in_y * kTile + in_x : size_t(image_index(p, x, y));
        if (out) store_pixel(p, out, index, color);
        if (HITS) hits[index] = rec;
    } else if (tile_valid && p.tile_count > 1) {
        // pixels of an edge tile that fall outside the image: keep the compact tile list fully defined
        const size_t index = size_t(local_tile) * (kTile * kTile) + in_y * kTile + in_x;
        const float zero[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (out) store_pixel(p, out, index, zero);
        if (HITS) memset(&hits[index], 0, sizeof(vx_hit));
    }

    if (STATS) {
        uint32_t v[11] = {ctr.rays, ctr.iterations, ctr.pushes, ctr.leaf_tests, ctr.leaf_tests_trilinear, ctr.boundaries, ctr.csvo_header_bytes,
                          ctr.csvo_pointer_bytes, active ? 1u : 0u, lit, shadow_rays};
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            unsigned long long s = v[k];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane == 0 && s) atomicAdd(&counters[k], s);
        }
    }
}

// ---- v2: persistent wavefront kernel ---------------------------------------------------------------------------
//
// One workgroup = one wave64 that keeps its 64 lanes fed from a global queue of 8x8-pixel sub-tiles. A lane's ray is a
// small state machine (Trav): IDLE -> TRAV (primary) -> [LEAF -> TRAV]* -> DONE -> shade -> TRAV (shadow) -> ... ->
// DONE -> pixel written -> IDLE. The expensive, rare phases (leaf test = material row + texture sample; shading; ray
// set-up) are not executed the moment one lane needs them: lanes park, and the wave services them when a ballot
// shows that `service_min` lanes are waiting (or nobody is left traversing); idle lanes are re-filled with new pixels
// when `refill_min` of them are free. The common descend/advance/pop step therefore runs with most lanes active
// instead of the ~30 % the one-thread-per-pixel kernel reached (profiles/round1/v1_*).
enum LaneState : int { kIdle = 0, kTrav = 1, kLeaf = 2, kDone = 3, kMissed = 4, kDeep = 5, kForeign = 6, kHeld = 7 };
static_assert(int(kTrav) == int(vxd::kTravContinue) && int(kLeaf) == int(vxd::kTravAtLeaf) && int(kMissed) == int(vxd::kTravFinished) &&
                  int(kDeep) == int(vxd::kTravDeep) && int(kForeign) == int(vxd::kTravForeign),
              "a TravStatus is stored as the lane's state");

struct PersistentArgs {
    // The sub-tile queue: eight dispensers (kQueueStride words apart: one memory-side atomic unit each), dispenser c hands out the sub-tiles
    // c, c + 8, c + 16, ... beyond the waves' first ones (wave w starts on sub-tile w without asking). A wave draws from dispenser
    // (w & 7) and, when that one is empty, from the next. One dispenser for 4096 waves is 70 M atomic adds per second on one address:
    // more than the memory side carries out there -- a ticket took tens of microseconds. A stream has two sets: a launch uses one and
    // clears the other for its successor (which does not start before this one has ended).
    uint32_t* work_counter;   // this launch's set
    uint32_t* next_counter;   // the set to clear
    uint32_t total_subtiles;  // n_local_tiles * 16
    uint32_t refill_min, service_min;
    uint32_t foreign_min;     // images of CSVO worlds: rays led into a voxel wait until this many of a wave's lanes are, and go together
    // Expensive sub-tiles first. A ray is a chain of dependent steps -- about 0.8 us per iteration on a busy device -- so a frame cannot
    // end before its longest rays do (up to ~300 iterations against a mean of ~30): handed out in screen order they start in mid-frame
    // and the frame ends with a long tail of waves that wait for a few of them (profiles/timeline.py). So every ray that ends notes
    // its iteration count in its sub-tile's entry of `cost_cur` (atomic max), a small kernel behind the frame sorts the sub-tiles into
    // sixteen cost classes, most expensive first, screen order within a class (order_kernel), and the NEXT frame of the same view on this
    // stream draws its tickets through that table: `order` (null: screen order). Order only: no pixel's value depends on it.
    const uint32_t* order;          // [total_subtiles] sub-tile ids, or null
    uint32_t* cost_cur;             // [total_subtiles] tag << 12 | iterations of the sub-tile's longest ray this frame; null = do not note
    uint32_t cost_floor;            // a wave notes its sub-tile's longest ray from this many iterations on (note_cost_wave)
    // SORTED builds (render_persistent): the unit of the queue is a PASS -- 64 pixels of a block of four sub-tiles (16x16 pixels), put together by
    // what the block's pixels cost two frames ago in this view on this stream (unit = 4 x block + pass). `perm_in` [unit][lane] = the pixel a
    // lane takes (a byte: sub-tile of the block << 6 | place in the sub-tile's Morton order). A wave that has rendered a pass leaves its pixels
    // and their costs in `pass_out` [unit][lane] (cost << 8 | pixel); the wave that takes a block's pass 0 also reads the block's four records
    // of the LAST frame (`pass_in`) and makes the NEXT frame's passes of the block (partition_block -> `perm_out`). Between writer and reader
    // of every table lies a kernel boundary: no wave ever waits for another, nothing is fenced. Any permutation is a correct frame; a view's
    // first frames read tables that say "sub-tile by sub-tile".
    const uint32_t* pass_in;
    uint32_t* pass_out;
    const uint8_t* perm_in;
    uint8_t* perm_out;
    uint32_t sort_turn, sort_mask;  // a block is re-sorted when ((block + sort_turn) & sort_mask) == 0 -- every fourth frame of its stream: mask 3 --, its passes kept otherwise
    uint32_t cur_tag;               // frame tag (20 bits, never 0): entries with another tag are stale (no clearing between frames)
    uint32_t ticket_ahead;          // 0 = no; 1 + g = waves draw their next sub-tile's ticket when they start on one (its round trip runs under the traversal),
                                    // except for the frame's last g quarter-grids of tickets
    uint32_t timeline_part;         // measurement: which part of the service phases the timeline's tick count covers (0 all, 1 leaf tests, 2 finished rays, 3 refill, 4 ray set-up, 5 excursions into voxels)
    unsigned long long* timeline;   // measurement (VX_TIMELINE=1), else null: per wave {start, queue found empty, exit} in 10 ns ticks, pixels taken
    // BATCH kernels: per wave a ring of ray records and a ring of result records (kWaveBatchBytes each wave), see render_persistent
    uint8_t* batch;
    unsigned long long* excursions;  // [0] rays that made the excursion into a voxel on the world's bytes, [1] of which started over, [2] service phases that ran excursions, [3] loop iterations made on the bytes
};

// Images of CSVO worlds: pixels a wave gave up on the image (a ray's walk inside the voxel it started in overwrote what the rest of
// the ray depends on, vx_device.hpp: walk_voxel_on_bytes) -- a few in a hundred. The wave renders them whole on the world's own
// bytes once the tile queue is empty and its rays are done (a second phase of the same kernel, not a second kernel: its registers
// overlay the first phase's, and a frame stays one command).
struct PixelList {
    // in chunks of 128 dwords that the wave chains together: [0] previous chunk + 1 (0 = none), [1] entries, [2..127] out_index
    // values. Chunks come from a ring (`mask` + 1 of them, a power of two) through one counter that only ever grows; a wave touches
    // nothing but its own chunks, so no wave ever waits for another.
    uint32_t* chunks;
    uint32_t* next_chunk;
    uint32_t mask;
};
constexpr uint32_t kChunkDwords = 128, kChunkEntries = 126;
// FOREIGN = kForeignRerun: the list holds RAYS instead (below): chunks of 1024 dwords, [0] previous chunk + 1, [1] records, from dword 16 on up to
// 64 records of 12 dwords {pixel | shadow << 31, origin in octree space, colour to be lit, diffuse + specular}
constexpr int kForeignRerun = 3;
constexpr uint32_t kRayChunkDwords = 1024, kRayChunkRecords = 64, kRayRecordDwords = 12, kRayChunkHeader = 16;

// compact / row-major output index -> pixel coordinates (the inverse of the index computation in the refill)
__device__ __forceinline__ void out_index_to_xy(const RenderParams& p, uint32_t out_index, uint32_t& x, uint32_t& y) {
    if (p.tile_count > 1) {
        const uint32_t local_tile = out_index >> 10, in_y = (out_index >> 5) & 31u, in_x = out_index & 31u;
        const uint32_t tile = p.tile_order[local_tile * p.tile_count + p.tile_rank];
        x = (tile % p.tiles_x) * kTile + in_x;
        y = (tile / p.tiles_x) * kTile + in_y;
    } else {
        x = out_index % p.width;
        y = out_index / p.width;
        if (p.rgba8) y = p.height - 1u - y;
    }
}

constexpr uint32_t kQueues = 8, kQueueStride = 64;  // dispensers of the sub-tile queue, words between them
constexpr uint32_t kCostFloor = 64;  // rays that end sooner (the mean is about 30; one in eight gets here) leave their sub-tile in the cheapest class: no note

// the sub-tile (8x8 pixels: the unit of the queue) a pixel's output index lies in
__device__ __forceinline__ uint32_t subtile_of(const RenderParams& p, uint32_t out_index) {
    uint32_t local_tile, in_x, in_y;
    if (p.tile_count > 1) {
        local_tile = out_index >> 10;
        in_y = (out_index >> 5) & 31u;
        in_x = out_index & 31u;
    } else {
        uint32_t x, y;
        out_index_to_xy(p, out_index, x, y);
        local_tile = (y / kTile) * p.tiles_x + x / kTile;
        in_x = x & 31u;
        in_y = y & 31u;
    }
    const uint32_t sx = in_x >> 3, sy = in_y >> 3;  // 4x4 sub-tiles in Morton order (see the refill)
    return local_tile * 16u + ((sx & 1u) | ((sy & 1u) << 1) | ((sx & 2u) << 1) | ((sy & 2u) << 2));
}

// the maximum of a value over the wave's 64 lanes (every lane active), as a wave-uniform value: four DPP steps inside a row of 16 lanes, the four
// rows' results through scalar registers
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    uint32_t o;
    o = uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0xB1, 0xF, 0xF, false)); v = v > o ? v : o;   // quad_perm [1,0,3,2]
    o = uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x4E, 0xF, 0xF, false)); v = v > o ? v : o;   // quad_perm [2,3,0,1]
    o = uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x141, 0xF, 0xF, false)); v = v > o ? v : o;  // row_half_mirror
    o = uint32_t(__builtin_amdgcn_update_dpp(0, int(v), 0x140, 0xF, 0xF, false)); v = v > o ? v : o;  // row_mirror
    const uint32_t r0 = uint32_t(__builtin_amdgcn_readlane(int(v), 0)), r1 = uint32_t(__builtin_amdgcn_readlane(int(v), 16));
    const uint32_t r2 = uint32_t(__builtin_amdgcn_readlane(int(v), 32)), r3 = uint32_t(__builtin_amdgcn_readlane(int(v), 48));
    const uint32_t r01 = r0 > r1 ? r0 : r1, r23 = r2 > r3 ? r2 : r3;
    return r01 > r23 ? r01 : r23;
}

// this lane's rank among the set lanes of a wave mask
__device__ __forceinline__ uint32_t rank_in_mask(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)); }

// a ray of this pixel has just ended after `iterations` loop iterations: the sub-tile's entry keeps the maximum
__device__ __forceinline__ void note_cost(const PersistentArgs& a, const RenderParams& p, uint32_t out_index, uint32_t iterations) {
    if (iterations < kCostFloor || !a.cost_cur) return;
    atomicMax(&a.cost_cur[subtile_of(p, out_index)], (a.cur_tag << 12) | (iterations < 4095u ? iterations : 4095u));
}

// The same for a whole wave (every lane of the wave calls it; `done` = this lane's ray has just ended). With the lanes in lockstep the rays
// that end in a service phase are one sub-tile's: their maximum is found in registers (four DPP steps inside a row of 16 lanes, the four
// rows' results through scalar registers) and ONE lane notes it -- an atomic is carried out at the memory side of the L2s, 32 bytes of HBM
// write traffic each, and a wave's next wait for memory waits for it too: 375 K of them a C3 frame, 12 MB. Lanes of several sub-tiles (any
// other service_min): a note per lane, as before.
// (UNIT: the wave's lanes are all of one unit of the queue, `unit` -- a SORTED build's pass)
template <bool UNIT = false>
__device__ __forceinline__ void note_cost_wave(const PersistentArgs& a, const RenderParams& p, bool done, uint32_t out_index, uint32_t iterations, uint32_t unit = 0u) {
    if (!a.cost_cur) return;
    const bool noting = done && iterations >= a.cost_floor;
    const unsigned long long m = __ballot(noting);
    if (m == 0ull) return;
    const uint32_t st = UNIT ? unit : subtile_of(p, out_index);
    const uint32_t st0 = UNIT ? unit : uint32_t(__builtin_amdgcn_readlane(int(st), int(__builtin_ctzll(m))));
    uint32_t v = noting ? (iterations < 4095u ? iterations : 4095u) : 0u;
    if (UNIT || __ballot(noting && st != st0) == 0ull) {
        const uint32_t top = wave_max_u32(v);
        if (threadIdx.x == 0) atomicMax(&a.cost_cur[st0], (a.cur_tag << 12) | top);
    } else if (noting && iterations >= kCostFloor) {
        atomicMax(&a.cost_cur[st], (a.cur_tag << 12) | v);
    }
}

// SORTED builds: a block's 256 pixels into four passes of 64 by what their rays cost. v[r] = the record of the pixel lane `lane` of pass r rendered
// (cost << 8 | pixel); out = the block's four passes in the next frame's table, 64 bytes each: the cheapest 64 pixels are pass 0 ... the most
// expensive pass 3 -- with the lanes in lockstep a pass costs what its longest ray costs, so rays of a kind go together (the C3 frame: a
// quarter fewer trips of the traversal loop than sub-tile by sub-tile, profiles/round3/pass_al). Three boundaries by bisection on the cost
// (wave-wide counts are ballots), ties split by (r, lane) so that every pass gets exactly 64 pixels; a pixel's place in its pass = its rank
// there. ~700 instructions a block, once a block and frame.
__device__ __forceinline__ void partition_block(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, uint8_t* out) {
    const uint32_t v[4] = {v0, v1, v2, v3};
    uint32_t c[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) c[r] = (v[r] >> 8) < 1023u ? (v[r] >> 8) : 1023u;
    auto count_le = [&](uint32_t t) -> uint32_t {
        uint32_t n = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) n += uint32_t(__popcll(__ballot(c[r] <= t)));
        return n;
    };
    // the smallest T in [lo, hi] with count_le(T) >= target (count_le(hi) >= target holds)
    auto boundary = [&](uint32_t lo, uint32_t hi, uint32_t target) -> uint32_t {
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (count_le(mid) >= target) hi = mid;
            else lo = mid + 1u;
        }
        return lo;
    };
    uint32_t T[3];
    T[1] = boundary(0u, 1023u, 128u);
    T[0] = boundary(0u, T[1], 64u);
    T[2] = boundary(T[1], 1023u, 192u);
    uint32_t g[4] = {0u, 0u, 0u, 0u};  // the pass each of this lane's four pixels goes to
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        uint32_t below = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) below += uint32_t(__popcll(__ballot(c[r] < T[k])));
        const uint32_t need = 64u * uint32_t(k + 1) - below;  // of the pixels that cost exactly T[k], this many stay below the boundary
        uint32_t seen = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool eq = c[r] == T[k];
            const unsigned long long m = __ballot(eq);
            g[r] += (c[r] > T[k] || (eq && seen + rank_in_mask(m) >= need)) ? 1u : 0u;
            seen += uint32_t(__popcll(m));
        }
    }
#pragma unroll
    for (uint32_t pass = 0; pass < 4; ++pass) {
        uint32_t seen = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool mine = g[r] == pass;
            const unsigned long long m = __ballot(mine);
            if (mine) out[pass * 64u + seen + rank_in_mask(m)] = uint8_t(v[r]);
            seen += uint32_t(__popcll(m));
        }
    }
}

// BATCH kernels: a record is four 16-byte words; per wave 256 ray records and 128 result records
constexpr uint32_t kRayRing = 256, kHitRing = 128;
constexpr size_t kWaveBatchBytes = size_t(kRayRing + kHitRing) * 64;
__device__ __forceinline__ uint32_t rank_in(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)); }
__device__ __forceinline__ uint32_t fbits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float bitsf(uint32_t u) { return __uint_as_float(u); }

// IMAGE = the rays walk the traversal image of the world (traversal_image.hpp) instead of its own bytes. FOREIGN (an image of a
// CSVO world; = VX_SVO_CSVO): a ray that is about to be led into the voxel it started in makes that excursion on the world's own
// bytes and comes back to the image (vx_device.hpp, walk_voxel_on_bytes) -- in the service phase, like every other rare and
// expensive thing a ray can need. (The image of an ESVO world serves such rays itself.)
// SHALLOW: no ray can push below the LDS-resident stack levels (the host knows the image's depth): no hand-over test in the loop.
// LV: stack levels resident in LDS -- 13 (three u32 planes); 16 with a 16-bit third plane (image cursors: worlds of 14 to 16 levels
// without the hand-over; the same 10 KB per wave); or 12 with a 16-bit third plane (images of at most 12 levels: 7.5 KB per wave, so
// that 20 waves fit a CU's LDS -- the build for five waves per SIMD, MINW = 5: 96 VGPRs, the spills stay in the service phases).
// HOT (experiment X1): the image's root octant and its eight child octants copied into LDS, PUSHes out of them served from there.
// BATCH (image-only renders): what does not need the lane's own traversal state is not done by the few lanes that happen to need it
// in a service phase (about 30 % of the wave), but 64 at a time, by all lanes, from records in two rings of the wave's own (global
// memory; no other wave touches them, so there is nothing to wait for and nothing to synchronise): a finished ray leaves a result
// record and its lane takes the next ray record at once; 64 result records are shaded together (primary hits: material, normal
// map, light -> the pixel, or a shadow-ray record; misses: the sky; shadow results: the lit pixel); primary rays are generated for
// a whole sub-tile at a time. Leaf tests (they continue the traversal) and ray set-up stay with the lane. Same arithmetic per
// pixel, same pixels.
template <int SVO, bool HITS, bool STATS, int MINW = 1, int FOREIGN = 0, bool SHALLOW = false, int LV = kLdsLevels, bool HOT = false, bool BATCH = false, bool TL = false, bool SORTED = false>
__global__ __launch_bounds__(64, MINW) void render_persistent(SceneArgs sa, RenderParams p, PersistentArgs a, float4* __restrict__ out,
                                                        vx_hit* __restrict__ hits, unsigned long long* __restrict__ counters, PixelList todo) {
    constexpr bool IMAGE = SVO == VX_SVO_IMAGE || SVO == VX_SVO_IMAGE_WIDE;
    static_assert(!(IMAGE && STATS), "the instrumented kernel counts the reference's own fetches: it walks the world's own bytes");
    // FOREIGN = kForeignRerun (image-only renders): a ray that is about to be led into the voxel it started in is not walked here at all. Its
    // lane takes it down -- pixel, origin, and for a shadow ray what the pixel's colour still needs -- in a list of the wave's own and is
    // free; when the wave's queue is empty and its rays are done, the wave runs the listed rays, 64 at a time, from their origins on the
    // world's own bytes with the reference's own cursor (a primary ray: the whole pixel) -- the same iterations the image cursor made,
    // then the walk inside the voxel, then the rest: the same ray. The render loop carries no code for the walk (whose registers its
    // service phases used to spill around: 11 % of a C3 frame), no lane waits for company, nothing stalls a wave in mid-frame, and the
    // listed rays run with every lane busy.
    static_assert(FOREIGN == 0 || (IMAGE && (FOREIGN == VX_SVO_CSVO || FOREIGN == kForeignRerun)), "FOREIGN: the image of a CSVO world");
    static_assert(FOREIGN != kForeignRerun || (!HITS && !STATS), "rays for the world's bytes are listed by image-only renders");
    static_assert(!SHALLOW || IMAGE, "only a traversal image bounds how deep a ray can get");
    static_assert(!BATCH || (!HITS && !STATS), "batched service phases: image-only renders");
    // SORTED: the queue hands out passes -- 64 pixels of a block of four sub-tiles that last frame's costs put together (PersistentArgs::perm_in,
    // partition_block) -- instead of sub-tiles; lanes are refilled only when all 64 are idle: a pass is a batch
    static_assert(!SORTED || (IMAGE && !HITS && !STATS && !BATCH && !HOT && FOREIGN != VX_SVO_CSVO), "sorted passes: image-only renders without the excursion");
    // (the image kernels are only launched for textures whose height is a power of two -- launch_render -- and say so to the sampler, a literal the
    // compiler folds: REPEAT is a mask, nothing of the general wrap is in these kernels' code -- 1-3 % of a frame, profiles/round3/pass_af)
    // TL: the build that fills in the wave timeline (VX_TIMELINE=1; profiles/timeline.py). Everywhere else the instrumentation is compiled out, not
    // switched off: its stamps and counters are wave-uniform state that lives through the whole kernel, and with them the ESVO image kernel
    // spilled 123 scalar registers instead of 50 and was 13 % longer (C3 +2 % without: profiles/round3/pass_ag).
    if constexpr (!TL) a.timeline = nullptr;
    auto vouched = [](DevScene s) { s.tex.pow2_height = IMAGE; return s; };
    const DevScene sc = vouched(IMAGE ? make_image_scene(sa) : make_scene(sa));
    const uint32_t lane = threadIdx.x;
    static_assert(LV == kLdsLevels || IMAGE, "only an image cursor's third stack word fits 16 bits");
    StackSpill spill;
    static_assert(!HOT || (SVO == VX_SVO_IMAGE && LV == kLdsLevels), "the LDS copy of the top levels: byte-offset images, 13 stack levels");
    typedef Stack<64, false, false, LV, (LV != kLdsLevels) || HOT, HOT> FullStack;
    FullStack st;       // all 23 levels: LDS, then the per-lane spill array
    // the same LDS slots, no range checks: what the traversal loop uses. SHALLOW (an image of at most LV levels): no ray
    // ever needs anything else
    typedef Stack<64, true, SHALLOW, LV, (LV != kLdsLevels) || HOT, HOT> FastStack;
    FastStack fast_st;
    st.init(lane, &spill);
    fast_st.init(lane, &spill);
    if (HOT) fast_st.load_hot(sc, lane);
    // a ray may use fast_st while every level it can pop to is LDS resident
    constexpr int kFastFloor = FullStack::kBaseScale - 1;
    // set in Trav::iter while the lane is not traversing, so that "iter < kMaxSteps" alone says "run one more step"
    constexpr uint32_t kParked = 0x80000000u;

    Trav<SVO> tr;
    tr.iter = kParked;
    int state = kIdle;
    bool shadow_ray = false;
    uint32_t out_index = 0;
    float primary_rd[3] = {0, 0, 0};  // kept while a primary ray is in flight: the sky needs it if the ray misses (world.glsl:135-138)
    float keep_color[4] = {0, 0, 0, 0}, keep_ds = 0.0f;
    float held_t = -1.0f;  // FOREIGN = VX_SVO_CSVO: the distance of a shadow ray that ended inside its voxel (kHeld)
    vx_hit rec;            // HITS only
    uint32_t steps = 0;    // HITS only
    Counters ctr = {};
    uint32_t n_pixels = 0, lit = 0, shadow_rays = 0;
    uint32_t wave_steps = 0, services = 0, refills = 0, tail_wave_steps = 0, tail_iterations = 0;  // STATS only, wave-uniform

    uint32_t cursor = 64, sub = 0;  // wave-uniform: position inside the current sub-tile
    // SORTED: a pass is in flight (its unit: `sub`); this lane's record of it: what its pixel has cost so far << 8 | the pixel
    bool have_unit = false;
    uint32_t rec_now = 0;
    bool queue_empty = false;
    const unsigned long long t_start = a.timeline ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long t_empty = 0ull;
    uint32_t taken = 0;
    uint32_t in_service = 0, service_phases = 0;  // timeline only, wave-uniform
    // timeline only: shader-clock stamps (s_memtime) -- the wave's whole life and the part of it spent in the traversal loop -- and the loop's trips
    const unsigned long long c_start = a.timeline ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long loop_cycles = 0ull;
    uint32_t loop_trips = 0;
    // the sub-tile queue: a ticket is this launch's sub-tile number (lane 0's value counts)
    // Wave w starts on sub-tile w without asking (a launch never has more waves than sub-tiles): 4096 waves do not open the frame by
    // queueing at one counter. The counter hands out the sub-tiles from gridDim.x on.
    uint32_t my_queue = blockIdx.x & (kQueues - 1u);  // wave-uniform: the dispenser this wave draws from
    // A ticket is DRAWN (the atomic issued, its raw count -- lane 0's -- left in a vector register) and, later, SETTLED (the count waited for and
    // made the sub-tile's number). Nothing between the two may touch the raw value: round 2's draw did the arithmetic at once, so the wave
    // waited out the atomic's round trip (2-3 us) in every refill -- "ahead" in name only (the refill was 24 of the 70 us a wave spends in its
    // service phases, profiles/round3/pass_p). (A wave's memory operations complete in order as far as its wait counter goes: the first wait
    // for ANY later load waits for the atomic too. Drawn where it is, that is after the ray generation and set-up of a whole sub-tile.)
    auto draw_raw = [&]() -> uint32_t {
        uint32_t raw = 0;
        if (lane == 0) raw = atomicAdd(a.work_counter + my_queue * kQueueStride, 1u);
        return raw;
    };
    // the n-th ticket of dispenser c is sub-tile (first_c + n) * 8 + c, first_c = how many of the waves' own first sub-tiles are c's
    auto ticket_of = [&](uint32_t raw, uint32_t queue) -> uint32_t {
        return (((gridDim.x + kQueues - 1u - queue) >> 3) + uint32_t(__builtin_amdgcn_readfirstlane(raw))) * kQueues + queue;
    };
    uint32_t ticket_raw = 0, ticket_queue = my_queue;  // the ticket drawn ahead: its raw count and the dispenser it came from
    bool ticket_ahead = true, ticket_first = true;   // wave-uniform; the wave's first ticket is its own number: nothing was drawn
    // the ticket as the wave's value; a dispenser that has run dry sends the wave on to the next one (the frame's last stretch only)
    auto settle_ticket = [&]() -> uint32_t {
        uint32_t t = ticket_ahead ? (ticket_first ? blockIdx.x : ticket_of(ticket_raw, ticket_queue)) : ticket_of(draw_raw(), my_queue);
        ticket_ahead = false;
        ticket_first = false;
        for (uint32_t tried = 1; t >= a.total_subtiles && tried < kQueues; ++tried) {
            my_queue = (my_queue + 1u) & (kQueues - 1u);
            t = ticket_of(draw_raw(), my_queue);
        }
        return t;
    };
    // SORTED: the unit of a ticket -- most expensive first, or screen order with a block's four passes on consecutive tickets of ONE dispenser
    // (ticket t = 8 n + c is dispenser c's n-th): a dispenser serves the waves of one XCD, and a pass spans its whole block -- four L2s would
    // each fetch the block's part of the world otherwise (cycles per trip of the loop 832 against 788, profiles/round3/pass_an)
    auto unit_of = [&](uint32_t t) -> uint32_t {
        const uint32_t in_turn = t < (a.total_subtiles & ~31u) ? (((t >> 5) * 8u + (t & 7u)) << 2) | ((t >> 3) & 3u) : t;
        const uint32_t u = a.order ? uint32_t(__builtin_amdgcn_readfirstlane(a.order[t])) : in_turn;
        return u < a.total_subtiles ? u : t;
    };
    if (blockIdx.x == 0 && lane < kQueues) a.next_counter[lane * kQueueStride] = 0u;
    uint32_t my_chunk = 0, my_fill = 0;  // FOREIGN, wave-uniform: newest chunk of this wave's list (+ 1) and its fill
    // BATCH, wave-uniform: the wave's two rings
    uint4* const ring_r = BATCH ? reinterpret_cast<uint4*>(a.batch + size_t(blockIdx.x) * kWaveBatchBytes) : nullptr;
    uint4* const ring_h = BATCH ? ring_r + size_t(kRayRing) * 4 : nullptr;
    uint32_t r_head = 0, r_count = 0, h_head = 0, h_count = 0;

    for (;;) {
        // ---- traverse until enough lanes wait for service (none are traversing on the first trip) ----
        // idle lanes are not waiting for anything; lanes that wait for company before their excursion (FOREIGN) are not waiting for service
        const uint32_t park_limit = a.service_min + uint32_t(__popcll(__ballot(state == kIdle || (FOREIGN == VX_SVO_CSVO && state == kForeign))));
        // the loop goes on while more than this many lanes traverse (64 - popcount(trav) < park_limit, and trav != 0)
        const uint32_t keep_going = park_limit >= 64u ? 0u : 64u - park_limit;
        // (the lanes that traverse, as the wave's mask: one compare per trip serves the loop's exit test and the next trip's execution mask)
        unsigned long long trav = __ballot(tr.iter < uint32_t(kMaxSteps));
        const unsigned long long c_loop = a.timeline ? __builtin_amdgcn_s_memtime() : 0ull;
        // The hand-scheduled loop (vx_loop_gfx950.hpp) for cursors on a byte-offset image that the resident stack levels cover. It does not
        // clear kHasAdjacentLeaf: a wave with a traversing ray that has just passed a translucent voxel takes the compiler's loop this time.
        constexpr bool kAsmLoop = VX_ASM_LOOP != 0 && IMAGE && SHALLOW && (LV == kLdsLevels || LV == 16 || (LV == 12 && VX_ASM_LOOP_12 != 0)) && !HOT && !STATS;
        bool by_hand = false;
        if constexpr (kAsmLoop) {
            by_hand = __ballot((tr.flags & Trav<SVO>::kHasAdjacentLeaf) != 0 && tr.iter < uint32_t(kMaxSteps)) == 0;
            // (the wide layout's entry index -- 4 x the octant index + the child -- is formed in 32 bits)
            if (SVO == VX_SVO_IMAGE_WIDE && sc.wide_bytes >= (uint64_t(1) << 35)) by_hand = false;
        }
        if (kAsmLoop && by_hand) {
            if constexpr (kAsmLoop) {
                const uint32_t lds_base = uint32_t(reinterpret_cast<uintptr_t>(fast_st.at(0)));
                const uint32_t lds_slot0 = lds_base + fast_st.slot0;
                // the 16-bit third plane: 2 * kPlane + (slot >> 1), the slot's offset being even and, for the resident scales, not negative
                const uint32_t lds_aux0 = lds_base + 2u * FastStack::kPlane + uint32_t(int32_t(fast_st.slot0) >> 1);
                // (FOREIGN = VX_SVO_CSVO: the loop also ends when foreign_min lanes wait for their walk into a voxel; the build that lists such rays
                // instead never waits for anything)
                const uint32_t f_waiting = FOREIGN == VX_SVO_CSVO ? uint32_t(__popcll(__ballot(state == kForeign))) : 0u;
                const uint32_t f_min = FOREIGN == VX_SVO_CSVO ? a.foreign_min : 0xffffffffu;
                if (a.timeline) traverse_loop_gfx950<SVO, FOREIGN != 0, true, LV>(tr, sc.world, sc.wide, lds_slot0, lds_aux0, keep_going, f_waiting, f_min, loop_trips);
                else traverse_loop_gfx950<SVO, FOREIGN != 0, false, LV>(tr, sc.world, sc.wide, lds_slot0, lds_aux0, keep_going, f_waiting, f_min, loop_trips);
                // a lane the loop parked says why in bits 28..30 of its iteration count
                const uint32_t why = (tr.iter >> 28) & 7u;
                if (why) {
                    state = LaneState(why);
                    tr.iter &= 0x8fffffffu;
                }
            }
        } else
        for (;;) {
            ++loop_trips;  // (one scalar add, unconditionally: a test of a.timeline here would cost the loop more than the count does)
            if (__builtin_amdgcn_inverse_ballot_w64(trav)) {  // traversing and below the iteration cap (svo.esvo.glsl:152)
                tr.template step_with<false, STATS, false, FastStack, false, FOREIGN != 0>(sc, fast_st, nullptr, STATS ? &ctr : nullptr, [&](TravStatus s) {
                    state = LaneState(s);
                    tr.iter = (s == kTravDeep || s == kTravForeign ? tr.iter - 1 : tr.iter) | kParked;  // a handed-over iteration is counted by the step that repeats it
                });
            }
            trav = __ballot(tr.iter < uint32_t(kMaxSteps));
            if (STATS) {
                ++wave_steps;
                if (queue_empty) { ++tail_wave_steps; tail_iterations += uint32_t(__popcll(trav)); }
            }
            if (uint32_t(__popcll(trav)) <= keep_going) break;
        }
        if (a.timeline) loop_cycles += __builtin_amdgcn_s_memtime() - c_loop;
        if (STATS) ++services;
        const unsigned long long t_service = a.timeline ? __builtin_amdgcn_s_memrealtime() : 0ull;
        unsigned long long t_part = 0ull;
#define VX_PART_BEGIN(n) if (a.timeline && a.timeline_part == (n)) t_part = __builtin_amdgcn_s_memrealtime()
#define VX_PART_END(n) if (a.timeline && a.timeline_part == (n)) in_service += uint32_t(__builtin_amdgcn_s_memrealtime() - t_part)
        // what a ray found: produced (leaf test, miss) and consumed (shading) within this service phase, never carried into the loop
        Result res;
        if (state == kTrav && tr.iter >= uint32_t(kMaxSteps)) {  // the cap ended this ray
            state = kMissed;
            tr.iter |= kParked;
        }

        // ---- rays below the LDS-resident levels (they started inside a voxel and were led on by leaf data): full stack ----
        if (!SHALLOW && state == kDeep) {
            tr.iter &= ~kParked;
            tr.sync_idx();
            for (;;) {
                const TravStatus s = tr.template step<false, STATS, false, FullStack, true, FOREIGN != 0>(sc, st, nullptr, STATS ? &ctr : nullptr);
                if (s == kTravContinue && tr.scale < kFastFloor) continue;
                state = LaneState(s);
                break;
            }
            if (FOREIGN && state == kForeign) --tr.iter;  // the excursion repeats this iteration
            if (state != kTrav) tr.iter |= kParked;
        }

        // ---- rays for the world's own bytes: taken down in the wave's list (chunks of 64 records: pixel | shadow << 31, the origin in octree space,
        // the colour to be lit, diffuse + specular), their lanes freed; the wave runs them when its queue is empty and its rays are done ----
        auto list_rays = [&](bool listed) {
            const unsigned long long fm = __ballot(listed);
            if (!fm) return;
            const uint32_t k = uint32_t(__popcll(fm)), r = rank_in(fm);
            const uint32_t room = my_chunk ? kRayChunkRecords - my_fill : 0u;
            uint32_t* cur = todo.chunks + size_t(my_chunk ? my_chunk - 1 : 0u) * kRayChunkDwords;
            uint32_t* dst = cur + kRayChunkHeader + (my_fill + r) * kRayRecordDwords;
            if (k > room) {  // (the records beyond the chunk's 64 start the next one)
                uint32_t c = 0;
                if (lane == 0) c = atomicAdd(todo.next_chunk, 1u);
                c = __builtin_amdgcn_readfirstlane(c) & todo.mask;
                uint32_t* fresh = todo.chunks + size_t(c) * kRayChunkDwords;
                if (lane == 0) {
                    fresh[0] = my_chunk;
                    fresh[1] = k - room;
                    if (my_chunk) cur[1] = kRayChunkRecords;
                }
                if (r >= room) dst = fresh + kRayChunkHeader + (r - room) * kRayRecordDwords;
                my_chunk = c + 1;
                my_fill = k - room;
            } else {
                my_fill += k;
                if (lane == 0) cur[1] = my_fill;
            }
            if (listed) {
                uint4* w = reinterpret_cast<uint4*>(dst);
                w[0] = make_uint4(out_index | (shadow_ray ? 0x80000000u : 0u), fbits(tr.rox), fbits(tr.roy), fbits(tr.roz));
                w[1] = make_uint4(fbits(keep_color[0]), fbits(keep_color[1]), fbits(keep_color[2]), fbits(keep_color[3]));
                w[2] = make_uint4(fbits(keep_ds), 0u, 0u, 0u);
                state = kIdle;
            }
            if (a.excursions && lane == 0) atomicAdd(&a.excursions[FOREIGN == kForeignRerun ? 0 : 1], (unsigned long long)k);
        };
        // (FOREIGN = kForeignRerun: every ray that is led into a voxel -- a few dozen a frame in worlds of at most 12 levels)
        if (FOREIGN == kForeignRerun) list_rays(state == kForeign);

        // ---- ... or (FOREIGN = VX_SVO_CSVO) the excursion on the world's own bytes ----
        // The walk runs with only these lanes active, so they go together: a lane waits (parked, at no cost to the loop) until
        // foreign_min lanes of the wave are there, or until no lane is left that could traverse meanwhile.
        VX_PART_BEGIN(5);
        constexpr bool kOpaqueFastPath = !STATS && !BATCH;
        bool color_pending = false;  // this lane's hit is of an opaque block and was found without its sample: its colour is still to be sampled
        // (wave-uniform) This phase walks rays into their voxels -- and does nothing else: the shadow rays of a sub-tile reach their voxels in the same
        // trip of the loop, the wave leaves the loop for their walk at once (foreign_min 1), and the lanes that are parked for another reason at that
        // moment -- at a leaf, finished -- stay parked until the phase in which everybody is (round 3 served them here: a lane whose primary ray was
        // shaded in a walk phase started its shadow ray in the middle of its sub-tile's batch, and the wave made 16 % more trips of the loop than for
        // the ESVO world, profiles/round4/pass_e). A shadow ray whose walk ENDS in the voxel (a phantom leaf is hit, the ray leaves the octree) is held
        // likewise -- its distance in a register, kHeld -- until then. (A primary ray that ends there -- an eye inside a voxel -- is served here: its
        // result lives in this phase.)
        bool walk_phase = false;
        bool walked = false;  // this lane made a walk in this phase
        if (FOREIGN == VX_SVO_CSVO) {
            const unsigned long long fm = __ballot(state == kForeign);
            // (unlikely: tells the register allocator that what the walk needs may be spilled around it, not across the phase)
            if (__builtin_expect(fm && (uint32_t(__popcll(fm)) >= a.foreign_min || __ballot(state == kTrav || state == kLeaf || state == kDone || state == kMissed || state == kDeep) == 0), 0)) {
                uint32_t on_bytes = 0;
                bool given_up = false;
                walk_phase = true;
                if (state == kForeign) {
                    walked = true;
                    tr.iter &= ~kParked;
                    const uint32_t before = tr.iter;
                    // (As a real call -- a register allocation of its own for the walk, the cursor and the result handed over through scratch: shared by
                    // the kernels it takes 198 VGPRs and the kernel's occupancy with it; one copy per kernel build keeps the kernel's bound, and
                    // the frame is a quarter slower -- 4K depth 13 2.58 -> 3.22 ms, what is saved and restored around the call is more than what
                    // the service phases spill: profiles/round3/pass_ab. At three waves per SIMD (168 registers, 2 spilled) a service phase is a
                    // third shorter -- three quarters of that because each of three waves gets a third of the SIMD's issue slots instead of a quarter --
                    // and the frame 4-7 % longer for the waves that are missing: VX_DEEP_WAVES=3, pass_aa. What a lane carries through the walk put away
                    // by hand around it -- ten values, 58 -> 38 spilled registers -- changes nothing: pass_ad. The spills are not what the walk costs.)
                    // (round 4: the walk as its own lean state machine on the image cursor's own registers and LDS slots, vx_device.hpp)
                    const TravStatus s = walk_voxel_on_bytes<SVO, FullStack, false, kOpaqueFastPath, !kOpaqueFastPath>(sc, make_buf(sa.world, clamp_u32(sa.world_bytes)), tr, st, true, res,
                                                                                                                        p.opaque_lo, p.opaque_hi, &color_pending);
                    on_bytes = tr.iter - before;
                    // back on the image / a phantom leaf inside the voxel was hit / the ray ended in there / given up (the pixel's turn comes later)
                    given_up = s == kTravForeign;
                    state = s == kTravContinue ? (SHALLOW || tr.scale >= kFastFloor ? kTrav : kDeep)
                                               : (s == kTravAtLeaf ? kDone : (s == kTravFinished ? kMissed : kIdle));
                    if (state != kTrav) tr.iter |= kParked;
                    if (shadow_ray && (state == kDone || state == kMissed)) {
                        held_t = state == kDone ? res.t : -1.0f;
                        state = kHeld;
                    }
                }
                // What the walk gave up on (a phantom chunk boundary, a phantom leaf of a block with holes, a straggler: walk_voxel_on_bytes) is run on the
                // world's own bytes at the end of the wave's life: image-only renders list the RAY (a shadow ray: only the shadow ray); renders
                // with hit records list the pixel, which is then rendered whole -- record and all.
                const unsigned long long gm = __ballot(given_up);
                if constexpr (!HITS) {
                    list_rays(given_up);
                } else if (gm) {
                    const uint32_t k = uint32_t(__popcll(gm));
                    if (my_chunk == 0 || my_fill + k > kChunkEntries) {  // a fresh chunk always has room for a whole wave
                        uint32_t c = 0;
                        if (lane == 0) c = atomicAdd(todo.next_chunk, 1u);
                        c = __builtin_amdgcn_readfirstlane(c) & todo.mask;
                        if (lane == 0) todo.chunks[size_t(c) * kChunkDwords] = my_chunk;
                        my_chunk = c + 1;
                        my_fill = 0;
                    }
                    uint32_t* chunk = todo.chunks + size_t(my_chunk - 1) * kChunkDwords;
                    if (given_up) chunk[2 + my_fill + __builtin_amdgcn_mbcnt_hi(uint32_t(gm >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(gm), 0u))] = out_index;
                    my_fill += k;
                    if (lane == 0) chunk[1] = my_fill;
                }
                // Counted only on request (vx_excursion_counters): four atomics on ONE line from every walk phase of every wave -- 65 M a second in a
                // depth-14 frame -- are more than the memory side carries out there, and the wave's next wait for memory waits for them (as for the
                // sub-tile queue's single counter in round 2)
                if (a.excursions) {
                    unsigned long long sum = on_bytes;
                    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
                    if (lane == 0) {
                        atomicAdd(&a.excursions[0], (unsigned long long)__popcll(fm));
                        if (HITS && gm) atomicAdd(&a.excursions[1], (unsigned long long)__popcll(gm));  // (image-only renders: counted by list_rays)
                        atomicAdd(&a.excursions[2], 1ull);
                        atomicAdd(&a.excursions[3], sum);
                    }
                }
            }
        }

        if (FOREIGN == VX_SVO_CSVO && !walk_phase && state == kHeld) {  // a held shadow ray: all that is ever looked at is its distance
            result_miss(res, false);
            res.t = held_t;
            state = kDone;
        }
        VX_PART_END(5);
        const bool serve = FOREIGN != VX_SVO_CSVO || !walk_phase || walked;  // (a walk phase serves nobody but its walkers)
        // ---- leaf tests (svo.esvo.glsl:185-265) for the parked lanes ----
        VX_PART_BEGIN(1);
        // A voxel of a block whose textures are opaque throughout is a hit whatever the sample says (RenderParams::opaque_*): its leaf test
        // is the value and arithmetic. The hit's colour is sampled when the hit is shaded (a shadow ray's never is).
        if (state == kLeaf && serve) {
            tr.iter &= ~kParked;
            tr.sync_idx();
            bool tested = false;
            if constexpr (kOpaqueFastPath) {
                const uint32_t value = tr.leaf_value(sc);
                const uint32_t set = value < 32u ? p.opaque_lo : p.opaque_hi;
                if (value < 64u && ((set >> (value & 31u)) & 1u) != 0u && !(tr.flags & Trav<SVO>::kHasAdjacentLeaf)) {
                    tr.leaf_hit_opaque(sc, value, res);
                    color_pending = true;
                    tested = true;
                    state = kDone;
                }
            }
            if (!tested) {
                const LeafOutcome o = tr.template leaf_test<false, STATS>(sc, st, true, res, nullptr, STATS ? &ctr : nullptr);
                state = o == kLeafHit ? kDone : (o == kLeafPassed ? (SHALLOW || tr.scale >= kFastFloor ? kTrav : kDeep) : kMissed);
            }
            if (state != kTrav) tr.iter |= kParked;
        }
        if (state == kMissed && serve) {
            result_miss(res, tr.inside_voxel());
            state = kDone;
        }
        VX_PART_END(1);

        // A lane that gets a new ray in this service phase -- the shadow ray of a shaded pixel, or the primary ray of a
        // freshly assigned pixel -- only records origin and direction; one Trav::init below serves both kinds together.
        float new_ro[3] = {0, 0, 0}, new_rd[3] = {0, 0, 0};
        bool new_ray = false;

        if constexpr (BATCH) {
            // ---- finished rays leave their result in the wave's ring; the lane is free ----
            {
                const bool done = state == kDone;
                const unsigned long long dm = __ballot(done);
                if (done) {
                    note_cost(a, p, out_index, tr.iter & ~kParked);
                    uint4* r = ring_h + size_t((h_head + h_count + rank_in(dm)) & (kHitRing - 1u)) * 4;
                    if (!shadow_ray) {
                        // (a miss needs the ray's direction for the sky and nothing of the hit: it travels in the position's place)
                        const bool sky = res.t == -1.0f;
                        r[0] = make_uint4(out_index, fbits(res.t), res.value, uint32_t(res.face_id));
                        r[1] = make_uint4(fbits(sky ? primary_rd[0] : res.pos[0]), fbits(sky ? primary_rd[1] : res.pos[1]), fbits(sky ? primary_rd[2] : res.pos[2]), fbits(res.uv[0]));
                        r[2] = make_uint4(fbits(res.uv[1]), fbits(res.lod), fbits(res.color[0]), fbits(res.color[1]));
                        r[3] = make_uint4(fbits(res.color[2]), fbits(res.color[3]), 0u, 0u);
                    } else {
                        r[0] = make_uint4(out_index | 0x80000000u, fbits(res.t), fbits(keep_color[0]), fbits(keep_color[1]));
                        r[1] = make_uint4(fbits(keep_color[2]), fbits(keep_color[3]), fbits(keep_ds), 0u);
                    }
                    state = kIdle;
                }
                h_count += uint32_t(__popcll(dm));
            }
            // ---- 64 results at a time, all lanes: shading, sky, light -> pixels and shadow-ray records. (Fewer only when nothing
            // else could feed the idle lanes: the queue and the ray ring are empty.) ----
            while (h_count >= 64u || (h_count != 0u && queue_empty && r_count == 0u)) {
                const uint32_t n = h_count < 64u ? h_count : 64u;
                bool cast = false;
                uint32_t pixel = 0;
                float so[3] = {0, 0, 0}, kc[4] = {0, 0, 0, 0}, kds = 0.0f;
                if (lane < n) {
                    const uint4* r = ring_h + size_t((h_head + lane) & (kHitRing - 1u)) * 4;
                    const uint4 w0 = r[0], w1 = r[1];
                    pixel = w0.x & 0x7fffffffu;
                    float color[4];
                    bool write = true;
                    if (!(w0.x >> 31)) {
                        const uint4 w2 = r[2], w3 = r[3];
                        Result rs;
                        rs.t = bitsf(w0.y); rs.value = w0.z; rs.face_id = int(w0.w);
                        rs.pos[0] = bitsf(w1.x); rs.pos[1] = bitsf(w1.y); rs.pos[2] = bitsf(w1.z);
                        rs.uv[0] = bitsf(w1.w); rs.uv[1] = bitsf(w2.x); rs.lod = bitsf(w2.y);
                        rs.color[0] = bitsf(w2.z); rs.color[1] = bitsf(w2.w); rs.color[2] = bitsf(w3.x); rs.color[3] = bitsf(w3.y);
                        rs.inside_voxel = false;
                        PrimaryOutcome o;
                        shade_primary(sc, p, rs, o);
                        if (rs.t == -1.0f) {  // no hit: sky (world.glsl:135-138)
                            const float rd[3] = {rs.pos[0], rs.pos[1], rs.pos[2]};
                            float sky[3];
                            sky_color(rd, sky);
                            color[0] = sky[0]; color[1] = sky[1]; color[2] = sky[2]; color[3] = 1.0f;
                        } else {
                            color[0] = o.color[0]; color[1] = o.color[1]; color[2] = o.color[2]; color[3] = o.color[3];
                            if (!o.final_color) {
                                cast = true;
                                write = false;
                                so[0] = o.shadow_origin[0]; so[1] = o.shadow_origin[1]; so[2] = o.shadow_origin[2];
                                kc[0] = o.color[0]; kc[1] = o.color[1]; kc[2] = o.color[2]; kc[3] = o.color[3];
                                kds = o.ds;
                            }
                        }
                    } else {
                        color[0] = bitsf(w0.z); color[1] = bitsf(w0.w); color[2] = bitsf(w1.x); color[3] = bitsf(w1.y);
                        apply_light(p, color, bitsf(w1.z), bitsf(w0.y) < 0.0f ? 1.0f : 0.0f);
                    }
                    if (write && out) store_pixel(p, out, pixel, color);
                }
                h_head += n;
                h_count -= n;
                const unsigned long long cm = __ballot(cast);
                if (cast) {
                    uint4* r = ring_r + size_t((r_head + r_count + rank_in(cm)) & (kRayRing - 1u)) * 4;
                    r[0] = make_uint4(pixel | 0x80000000u, fbits(so[0]), fbits(so[1]), fbits(so[2]));
                    r[1] = make_uint4(fbits(kc[0]), fbits(kc[1]), fbits(kc[2]), fbits(kc[3]));
                    r[2] = make_uint4(fbits(kds), 0u, 0u, 0u);
                }
                r_count += uint32_t(__popcll(cm));
            }
            // ---- idle lanes take the oldest ray records; primary rays are made a sub-tile at a time, by all lanes ----
            const unsigned long long idle_mask = __ballot(state == kIdle);
            const uint32_t n_idle = uint32_t(__popcll(idle_mask));
            if (idle_mask && (n_idle >= a.refill_min || idle_mask == ~0ull || queue_empty)) {
                while (r_count < n_idle && !queue_empty) {
                    const uint32_t t = settle_ticket();
                    if (t >= a.total_subtiles) {
                        queue_empty = true;
                        if (a.timeline) t_empty = __builtin_amdgcn_s_memrealtime();
                        break;
                    }
                    sub = a.order ? uint32_t(__builtin_amdgcn_readfirstlane(a.order[t])) : t;  // most expensive first, or screen order
                    if (sub >= a.total_subtiles) sub = t;  // (never: a table of another view is not used)
                    ++taken;
                    // sub-tile -> pixel: 32x32 tile (sharding unit), 4x4 sub-tiles in Morton order, 8x8 pixels in Morton order
                    const uint32_t local_tile = sub >> 4, sq = sub & 15u;
                    const uint32_t tile = p.tile_count > 1 ? p.tile_order[local_tile * p.tile_count + p.tile_rank] : local_tile;
                    const uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;
                    const uint32_t sx = (sq & 1u) | ((sq >> 1) & 2u), sy = ((sq >> 1) & 1u) | ((sq >> 2) & 2u);
                    uint32_t lx, ly;
                    lane_to_xy(lane, lx, ly);
                    const uint32_t in_x = sx * 8 + lx, in_y = sy * 8 + ly;
                    const uint32_t px_x = tx * kTile + in_x, px_y = ty * kTile + in_y;
                    const uint32_t index = p.tile_count > 1 ? local_tile * (kTile * kTile) + in_y * kTile + in_x : image_index(p, px_x, px_y);
                    const bool inside = px_x < p.width && px_y < p.height;
                    const unsigned long long im = __ballot(inside);
                    if (inside) {
                        float ro[3], rd[3];
                        primary_ray(p, px_x, px_y, ro, rd);
                        uint4* r = ring_r + size_t((r_head + r_count + rank_in(im)) & (kRayRing - 1u)) * 4;
                        r[0] = make_uint4(index, fbits(rd[0]), fbits(rd[1]), fbits(rd[2]));
                    } else if (p.tile_count > 1) {
                        // padding pixel of an edge tile: keep the compact tile list fully defined
                        const float zero[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (out) store_pixel(p, out, index, zero);
                    }
                    r_count += uint32_t(__popcll(im));
                }
                const uint32_t take = n_idle < r_count ? n_idle : r_count;
                const uint32_t rk = rank_in(idle_mask);
                if (state == kIdle && rk < take) {
                    const uint4* r = ring_r + size_t((r_head + rk) & (kRayRing - 1u)) * 4;
                    const uint4 w0 = r[0];
                    out_index = w0.x & 0x7fffffffu;
                    shadow_ray = (w0.x >> 31) != 0u;
                    if (shadow_ray) {
                        const uint4 w1 = r[1];
                        const uint32_t w2 = r[2].x;
                        new_ro[0] = bitsf(w0.y); new_ro[1] = bitsf(w0.z); new_ro[2] = bitsf(w0.w);
                        new_rd[0] = -p.u.light_dir[0]; new_rd[1] = -p.u.light_dir[1]; new_rd[2] = -p.u.light_dir[2];
                        keep_color[0] = bitsf(w1.x); keep_color[1] = bitsf(w1.y); keep_color[2] = bitsf(w1.z); keep_color[3] = bitsf(w1.w);
                        keep_ds = bitsf(w2);
                    } else {
                        new_ro[0] = p.ray_origin[0]; new_ro[1] = p.ray_origin[1]; new_ro[2] = p.ray_origin[2];
                        new_rd[0] = bitsf(w0.y); new_rd[1] = bitsf(w0.z); new_rd[2] = bitsf(w0.w);
                        primary_rd[0] = new_rd[0]; primary_rd[1] = new_rd[1]; primary_rd[2] = new_rd[2];
                    }
                    new_ray = true;
                }
                r_head += take;
                r_count -= take;
            }
            if (new_ray) {
                tr.init(sc, new_ro, new_rd, -1.0f);  // iter = 0: not parked
                state = kTrav;
            }
            if (__ballot(state != kIdle) == 0 && queue_empty && r_count == 0u && h_count == 0u) break;
            continue;
        }

        // ---- finished rays ----
        VX_PART_BEGIN(2);
        if constexpr (!SORTED) note_cost_wave(a, p, state == kDone && serve, out_index, tr.iter & ~kParked);  // (a SORTED build notes a pass when it has ended)
        if (state == kDone && serve) {
            float color[4];
            bool write = true;
            if constexpr (SORTED) rec_now += (tr.iter & ~kParked) << 8;
            if (!shadow_ray) {
                PrimaryOutcome o;
                shade_primary<kOpaqueFastPath>(sc, p, res, o, color_pending);
                if (HITS) {
                    rec.t = res.t; rec.value = res.value; rec.face_id = res.face_id; rec.flags = o.flags;
                    rec.pos[0] = res.pos[0]; rec.pos[1] = res.pos[1]; rec.pos[2] = res.pos[2];
                    rec.lod = res.lod; rec.uv[0] = res.uv[0]; rec.uv[1] = res.uv[1];
                    rec.shadow_t = -1.0f;
                    steps = tr.iter & ~kParked;
                }
                if (res.t == -1.0f) {  // no hit: sky (world.glsl:135-138)
                    float sky[3];
                    sky_color(primary_rd, sky);
                    color[0] = sky[0]; color[1] = sky[1]; color[2] = sky[2]; color[3] = 1.0f;
                } else {
                    if (STATS && !(o.flags & 8u)) ++lit;
                    color[0] = o.color[0]; color[1] = o.color[1]; color[2] = o.color[2]; color[3] = o.color[3];
                    if (!o.final_color) {
                        keep_color[0] = o.color[0]; keep_color[1] = o.color[1]; keep_color[2] = o.color[2]; keep_color[3] = o.color[3];
                        keep_ds = o.ds;
                        new_ro[0] = o.shadow_origin[0]; new_ro[1] = o.shadow_origin[1]; new_ro[2] = o.shadow_origin[2];
                        new_rd[0] = -p.u.light_dir[0]; new_rd[1] = -p.u.light_dir[1]; new_rd[2] = -p.u.light_dir[2];
                        new_ray = true;
                        shadow_ray = true;
                        write = false;
                        if (STATS) { ctr.rays++; ++shadow_rays; }
                    }
                }
            } else {
                color[0] = keep_color[0]; color[1] = keep_color[1]; color[2] = keep_color[2]; color[3] = keep_color[3];
                apply_light(p, color, keep_ds, res.t < 0.0f ? 1.0f : 0.0f);
                if (HITS) {
                    if (!(res.t < 0.0f)) rec.flags |= 4u;
                    rec.shadow_t = res.t;
                    steps += tr.iter & ~kParked;
                }
            }
            if (write) {
                if (out) store_pixel(p, out, out_index, color);
                if (HITS) {
                    rec.steps = steps;
                    hits[out_index] = rec;
                }
                state = kIdle;
            }
        }

        VX_PART_END(2);
        // ---- refill idle lanes from the sub-tile queue ----
        VX_PART_BEGIN(3);
        if constexpr (SORTED) {
            if (!queue_empty && __ballot(state != kIdle || new_ray) == 0ull) {  // every pixel of the pass is stored
                // the pass that has ended: its pixels and what they cost, for the next frame of this view on this stream
                if (have_unit) {
                    a.pass_out[size_t(sub) * 64u + lane] = rec_now;
                    // ... and what the pass cost -- its dearest pixel, primary and shadow ray together: half of it is what the order table's
                    // classes are of -- for "expensive passes first" (this wave is the pass's only writer: a plain store)
                    if (a.cost_cur) {
                        const uint32_t top = wave_max_u32(rec_now >> 8) >> 1;
                        if (lane == 0) a.cost_cur[sub] = (a.cur_tag << 12) | (top < 4095u ? top : 4095u);
                    }
                }
                const uint32_t t = settle_ticket();
                if (t >= a.total_subtiles) {
                    queue_empty = true;
                    if (a.timeline) t_empty = __builtin_amdgcn_s_memrealtime();
                } else {
                    sub = unit_of(t);
                    const uint32_t pid = a.perm_in[size_t(sub) * 64u + lane];
                    rec_now = pid;
                    ++taken;
                    if (a.ticket_ahead && t + ((a.ticket_ahead - 1u) * gridDim.x >> 2) < a.total_subtiles) {
                        ticket_raw = draw_raw();
                        ticket_queue = my_queue;
                        ticket_ahead = true;
                    }
                    if ((sub & 3u) == 0u) {  // a block's pass 0: the next frame's passes of the block
                        if ((((sub >> 2) + a.sort_turn) & a.sort_mask) == 0u) {  // ... from what the last frame's cost (its turn: 6 us of a wave's time, every fourth frame)
                            const uint32_t* rec4 = a.pass_in + size_t(sub) * 64u + lane;
                            partition_block(rec4[0], rec4[64], rec4[128], rec4[192], a.perm_out + size_t(sub) * 64u);
                        } else {  // ... as they are
                            reinterpret_cast<uint32_t*>(a.perm_out + size_t(sub) * 64u)[lane] = reinterpret_cast<const uint32_t*>(a.perm_in + size_t(sub) * 64u)[lane];
                        }
                    }
                    const uint32_t local_tile = sub >> 4, s = (sub & 12u) | (pid >> 6);
                    const uint32_t tile = p.tile_count > 1 ? p.tile_order[local_tile * p.tile_count + p.tile_rank] : local_tile;
                    const uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;
                    const uint32_t sx = (s & 1u) | ((s >> 1) & 2u), sy = ((s >> 1) & 1u) | ((s >> 2) & 2u);
                    uint32_t lx, ly;
                    lane_to_xy(pid & 63u, lx, ly);
                    const uint32_t in_x = sx * 8 + lx, in_y = sy * 8 + ly;
                    const uint32_t px_x = tx * kTile + in_x, px_y = ty * kTile + in_y;
                    out_index = p.tile_count > 1 ? local_tile * (kTile * kTile) + in_y * kTile + in_x : image_index(p, px_x, px_y);
                    if (px_x < p.width && px_y < p.height) {
                        primary_ray(p, px_x, px_y, new_ro, new_rd);
                        primary_rd[0] = new_rd[0]; primary_rd[1] = new_rd[1]; primary_rd[2] = new_rd[2];
                        new_ray = true;
                        shadow_ray = false;
                    } else if (p.tile_count > 1) {
                        const float zero[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // padding pixel of an edge tile: keep the compact tile list fully defined
                        if (out) store_pixel(p, out, out_index, zero);
                    }
                }
                have_unit = !queue_empty;
            }
        }
        unsigned long long idle_mask = SORTED ? 0ull : __ballot(state == kIdle);
        if (!queue_empty && idle_mask && !walk_phase && (uint32_t(__popcll(idle_mask)) >= a.refill_min || idle_mask == ~0ull)) {
            if (STATS) ++refills;
            for (int round = 0; round < 2 && idle_mask && !queue_empty; ++round) {
                if (cursor >= 64) {
                    // the ticket drawn ahead, if there is one (its round trip -- an atomic is carried out at the memory side -- ran under the
                    // traversal since)
                    const uint32_t t = settle_ticket();
                    if (t >= a.total_subtiles) {
                        queue_empty = true;
                        if (a.timeline) t_empty = __builtin_amdgcn_s_memrealtime();
                        break;
                    }
                    sub = a.order ? uint32_t(__builtin_amdgcn_readfirstlane(a.order[t])) : t;  // most expensive first, or screen order
                    if (sub >= a.total_subtiles) sub = t;  // (never: a table of another view is not used)
                    cursor = 0;
                    ++taken;
                    // one ahead (its round trip runs under the ray generation and set-up that follow, and the loop's first trip) -- but not in
                    // the frame's last stretch, where a sub-tile reserved by a busy wave is one an idle wave cannot take
                    if (a.ticket_ahead && t + ((a.ticket_ahead - 1u) * gridDim.x >> 2) < a.total_subtiles) {
                        ticket_raw = draw_raw();
                        ticket_queue = my_queue;
                        ticket_ahead = true;
                    }
                }
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(idle_mask), 0u));
                const uint32_t k = cursor + rank;
                if (state == kIdle && !new_ray && k < 64) {
                    // sub-tile -> pixel: 32x32 tile (sharding unit), 4x4 sub-tiles in Morton order, 8x8 pixels in Morton order
                    const uint32_t local_tile = sub >> 4, s = sub & 15u;
                    const uint32_t tile = p.tile_count > 1 ? p.tile_order[local_tile * p.tile_count + p.tile_rank] : local_tile;
                    const uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;
                    const uint32_t sx = (s & 1u) | ((s >> 1) & 2u), sy = ((s >> 1) & 1u) | ((s >> 2) & 2u);
                    uint32_t lx, ly;
                    lane_to_xy(k, lx, ly);
                    const uint32_t in_x = sx * 8 + lx, in_y = sy * 8 + ly;
                    const uint32_t px_x = tx * kTile + in_x, px_y = ty * kTile + in_y;
                    out_index = p.tile_count > 1 ? local_tile * (kTile * kTile) + in_y * kTile + in_x : image_index(p, px_x, px_y);
                    if (px_x < p.width && px_y < p.height) {
                        primary_ray(p, px_x, px_y, new_ro, new_rd);
                        primary_rd[0] = new_rd[0]; primary_rd[1] = new_rd[1]; primary_rd[2] = new_rd[2];
                        new_ray = true;
                        shadow_ray = false;
                        steps = 0;
                        if (STATS) { ctr.rays++; ++n_pixels; }
                    } else if (p.tile_count > 1) {
                        // padding pixel of an edge tile: keep the compact tile list fully defined
                        const float zero[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (out) store_pixel(p, out, out_index, zero);
                        if (HITS) memset(&hits[out_index], 0, sizeof(vx_hit));
                    }
                }
                const uint32_t n_idle = uint32_t(__popcll(idle_mask));
                cursor += n_idle < 64 - cursor ? n_idle : 64 - cursor;
                idle_mask = __ballot(state == kIdle && !new_ray);
            }
        }

        VX_PART_END(3);
        // ---- ray set-up (svo.esvo.glsl:50-150) for every lane that got a ray above ----
        VX_PART_BEGIN(4);
        if (new_ray) {
            tr.init(sc, new_ro, new_rd, -1.0f);  // iter = 0: not parked
            state = kTrav;
        }
        VX_PART_END(4);
#undef VX_PART_BEGIN
#undef VX_PART_END
        if (a.timeline) { if (a.timeline_part == 0) in_service += uint32_t(__builtin_amdgcn_s_memrealtime() - t_service); ++service_phases; }
        if (__ballot(state != kIdle) == 0 && queue_empty) break;
    }

    if (a.timeline && lane == 0) {
        unsigned long long* row = a.timeline + size_t(blockIdx.x) * 8;
        row[0] = t_start; row[1] = t_empty; row[2] = __builtin_amdgcn_s_memrealtime();
        row[3] = taken | ((unsigned long long)(service_phases & 0xfffu) << 20) | ((unsigned long long)in_service << 32);  // sub-tiles, service phases, ticks spent in them
        row[4] = __builtin_amdgcn_s_memtime() - c_start; row[5] = loop_cycles; row[6] = loop_trips; row[7] = 0;
    }
    // ---- second phase (image-only renders of a CSVO world): the rays this wave listed, on the world's own bytes ----
    if (FOREIGN == kForeignRerun || (FOREIGN == VX_SVO_CSVO && !HITS)) {
        const DevScene sc_bytes = vouched(make_scene(sa));
        Stack<64, false, false, int(FullStack::kStackBytes / (64u * 12u))> st2;  // (three full words per slot, over the same LDS: the first phase is over)
        st2.init(lane, &spill);
        const float to_light[3] = {-p.u.light_dir[0], -p.u.light_dir[1], -p.u.light_dir[2]};
        for (uint32_t c = my_chunk; c != 0;) {
            const uint32_t* chunk = todo.chunks + size_t(c - 1) * kRayChunkDwords;
            const uint32_t n = __builtin_amdgcn_readfirstlane(chunk[1]);
            c = __builtin_amdgcn_readfirstlane(chunk[0]);
            if (lane < n) {
                const uint4* w = reinterpret_cast<const uint4*>(chunk + kRayChunkHeader + lane * kRayRecordDwords);
                const uint4 w0 = w[0];
                const uint32_t index = w0.x & 0x7fffffffu;
                float color[4];
                if (w0.x >> 31) {
                    const uint4 w1 = w[1];
                    const float ds = bitsf(w[2].x);
                    // a shadow ray: from its origin on the world's bytes, with the reference's own cursor
                    Trav<VX_SVO_CSVO> tb;
                    tb.init_in_octree_space(sc_bytes, bitsf(w0.y), bitsf(w0.z), bitsf(w0.w), to_light, -1.0f);
                    Result rs;
                    for (;;) {
                        TravStatus s2 = tb.template step<false, false, false>(sc_bytes, st2, nullptr, nullptr);
                        if (s2 == kTravAtLeaf) {
                            const LeafOutcome o = tb.template leaf_test<false, false>(sc_bytes, st2, true, rs, nullptr, nullptr);
                            if (o == kLeafHit) break;
                            s2 = o == kLeafPassed ? kTravContinue : kTravFinished;
                        }
                        if (s2 == kTravFinished) {
                            result_miss(rs, tb.inside_voxel());
                            break;
                        }
                    }
                    color[0] = bitsf(w1.x); color[1] = bitsf(w1.y); color[2] = bitsf(w1.z); color[3] = bitsf(w1.w);
                    apply_light(p, color, ds, rs.t < 0.0f ? 1.0f : 0.0f);
                } else {
                    uint32_t x, y;
                    out_index_to_xy(p, index, x, y);
                    shade_pixel<VX_SVO_CSVO, false>(sc_bytes, p, x, y, st2, color, nullptr, nullptr, nullptr, nullptr);
                }
                if (out) store_pixel(p, out, index, color);
            }
        }
    }
    // ---- second phase (FOREIGN = VX_SVO_CSVO, renders with hit records): the pixels this wave gave up on the image, whole, on the world's own bytes ----
    if (FOREIGN == VX_SVO_CSVO && HITS) {
        const DevScene sc_bytes = vouched(make_scene(sa));
        // (the byte cursor's stack entries are three full words: the plain layout, as many levels as fit the same LDS -- the first phase is over)
        Stack<64, false, false, int(FullStack::kStackBytes / (64u * 12u))> st2;
        st2.init(lane, &spill);
        for (uint32_t c = my_chunk; c != 0;) {
            const uint32_t* chunk = todo.chunks + size_t(c - 1) * kChunkDwords;
            const uint32_t n = __builtin_amdgcn_readfirstlane(chunk[1]);
            c = __builtin_amdgcn_readfirstlane(chunk[0]);
            for (uint32_t i0 = 0; i0 < n; i0 += 64) {
                if (i0 + lane < n) {
                    const uint32_t index = chunk[2 + i0 + lane];
                    uint32_t x, y;
                    out_index_to_xy(p, index, x, y);
                    float color[4];
                    vx_hit r;
                    shade_pixel<VX_SVO_CSVO, false>(sc_bytes, p, x, y, st2, color, HITS ? &r : nullptr, nullptr, nullptr, nullptr);
                    if (out) store_pixel(p, out, index, color);
                    if (HITS) hits[index] = r;
                }
            }
        }
    }

    if (STATS) {
        uint32_t v[11] = {ctr.rays, ctr.iterations, ctr.pushes, ctr.leaf_tests, ctr.leaf_tests_trilinear, ctr.boundaries, ctr.csvo_header_bytes,
                          ctr.csvo_pointer_bytes, n_pixels, lit, shadow_rays};
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            unsigned long long sum = v[k];
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
            if (lane == 0 && sum) atomicAdd(&counters[k], sum);
        }
        if (lane == 0) {
            atomicAdd(&counters[11], (unsigned long long)wave_steps);
            atomicAdd(&counters[12], (unsigned long long)services);
            atomicAdd(&counters[13], (unsigned long long)refills);
            atomicAdd(&counters[14], (unsigned long long)tail_wave_steps);
            atomicAdd(&counters[15], (unsigned long long)tail_iterations);
        }
    }
}

// 2x2 ordered-grid supersampling (BASELINE.json C5): box filter of a (2w x 2h) render down to (w x h); one thread per
// output pixel, float4 loads and stores
__global__ __launch_bounds__(256) void resolve_2x2_kernel(const float4* __restrict__ src, uint32_t w, uint32_t h, float4* __restrict__ dst) {
    const uint32_t x = blockIdx.x * 16 + (threadIdx.x & 15u), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= w || y >= h) return;
    const size_t sw = size_t(w) * 2;
    const float4 a = src[size_t(2 * y) * sw + 2 * x], b = src[size_t(2 * y) * sw + 2 * x + 1];
    const float4 c = src[size_t(2 * y + 1) * sw + 2 * x], d = src[size_t(2 * y + 1) * sw + 2 * x + 1];
    dst[size_t(y) * w + x] = make_float4(((a.x + b.x) + (c.x + d.x)) * 0.25f, ((a.y + b.y) + (c.y + d.y)) * 0.25f, ((a.z + b.z) + (c.z + d.z)) * 0.25f,
                                         ((a.w + b.w) + (c.w + d.w)) * 0.25f);
}

template <int SVO>
__global__ __launch_bounds__(64) void picker_kernel(SceneArgs sa, const vx_picker_task* __restrict__ tasks, uint32_t n,
                                                    vx_picker_result* __restrict__ results) {
    const DevScene sc = make_scene(sa);
    StackSpill spill;
    Stack<64> st;
    st.init(threadIdx.x, &spill);
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    // picker.glsl:30-51
    const vx_picker_task task = tasks[i];
    Result res;
    uint32_t steps = 0;
    intersect<SVO, false, false, true>(sc, task.pos, task.dir, task.max_dst, false, st, res, steps, nullptr, nullptr);
    vx_picker_result r;
    memset(&r, 0, sizeof r);
    if (res.t > 0.0f) {
        r.dst = res.t;
        r.inside_voxel = res.inside_voxel ? 1u : 0u;
        r.pos[0] = res.pos[0]; r.pos[1] = res.pos[1]; r.pos[2] = res.pos[2];
        r.normal[0] = kFaceNormals[res.face_id][0]; r.normal[1] = kFaceNormals[res.face_id][1]; r.normal[2] = kFaceNormals[res.face_id][2];
    } else {
        r.dst = -1.0f;
    }
    results[i] = r;
}

struct TraceArgs {
    float pos[3], dir[3];
    float max_dst;
    int cast_translucent;
};

template <int SVO>
__global__ __launch_bounds__(64) void trace_kernel(SceneArgs sa, TraceArgs a, vx_result* __restrict__ result, vx_frame* __restrict__ frames,
                                                   uint32_t max_frames, uint32_t* __restrict__ n_frames) {
    const DevScene sc = make_scene(sa);
    StackSpill spill;
    Stack<64> st;
    st.init(threadIdx.x, &spill);
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Result res;
    uint32_t steps = 0;
    TraceSink tk;
    tk.frames = frames; tk.max_frames = max_frames; tk.n_frames = 0;
    intersect<SVO, true, false, true>(sc, a.pos, a.dir, a.max_dst, a.cast_translucent != 0, st, res, steps, (TracePtr)&tk, nullptr);
    vx_result r;
    r.t = res.t; r.value = res.value; r.face_id = res.face_id;
    r.pos[0] = res.pos[0]; r.pos[1] = res.pos[1]; r.pos[2] = res.pos[2];
    r.uv[0] = res.uv[0]; r.uv[1] = res.uv[1];
    r.color[0] = res.color[0]; r.color[1] = res.color[1]; r.color[2] = res.color[2]; r.color[3] = res.color[3];
    r.lod = res.lod;
    r.inside_voxel = res.inside_voxel ? 1 : 0;
    *result = r;
    *n_frames = tk.n_frames;
}

// SORTED builds' tables before a view's first frame: every pass = a sub-tile of its block in Morton order (what an unsorted build renders), nothing
// has cost anything
__global__ __launch_bounds__(256) void pass_identity_kernel(uint32_t* __restrict__ rec, uint8_t* __restrict__ perm0, uint8_t* __restrict__ perm1, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;  // [unit][lane]
    if (i < n) {
        const uint32_t pixel = (((i >> 6) & 3u) << 6) | (i & 63u);
        rec[i] = pixel;
        perm0[i] = uint8_t(pixel);
        perm1[i] = uint8_t(pixel);
    }
}

// Sorts a frame's sub-tiles for the next frame's queue (PersistentArgs::order): sixteen classes by the iteration count of the sub-tile's
// longest ray (class = min(15, iterations / 16); entries without this frame's tag are class 0), the highest class first, screen
// order within a class (a stable counting sort: the rays of neighbouring sub-tiles walk the same nodes). ONE workgroup of 1024 threads:
// thread t owns the contiguous run of sub-tiles [t * per, (t + 1) * per); (1) it counts its run's members of each class, (2) the
// counts are scanned class by class across the threads -- a wave-level scan by lane shuffles, the sixteen waves' totals through LDS --
// which gives every (class, thread) its first place in the table, (3) it walks its run again and puts every sub-tile in its place.
// 32 K sub-tiles (1080p) are 32 per thread: a few microseconds (round 2's version -- four waves, a ballot per class and block of 64 --
// took 180; it runs behind a frame on a stream of its own, but on the compute units the next frame wants).
constexpr uint32_t kCostClasses = 16, kCostStep = 16;  // classes of the order table: iterations / kCostStep, capped
constexpr uint32_t kOrderThreads = 1024;
__global__ __launch_bounds__(kOrderThreads) void order_kernel(const uint32_t* __restrict__ cost, uint32_t tag, uint32_t n, uint32_t* __restrict__ order, uint32_t step) {
    __shared__ uint32_t wave_totals[kCostClasses][kOrderThreads / 64];  // [slot][wave], slot 0 = the most expensive class
    __shared__ uint32_t slot_base[kCostClasses];
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63u;
    const uint32_t per = (n + kOrderThreads - 1u) / kOrderThreads;
    const uint32_t first = t * per < n ? t * per : n, last = first + per < n ? first + per : n;
    auto slot_of = [&](uint32_t i) -> uint32_t {
        const uint32_t c = cost[i];
        const uint32_t cls = (c >> 12) == tag ? ((c & 0xfffu) / step < kCostClasses - 1u ? (c & 0xfffu) / step : kCostClasses - 1u) : 0u;
        return kCostClasses - 1u - cls;
    };
    // (1) this thread's members of each class: sixteen 16-bit counters in eight words (a run is shorter than 65536)
    uint32_t packed[kCostClasses / 2] = {};
#pragma unroll 8  // (eight loads in flight: a run is read by one thread, one dependent round trip per element otherwise)
    for (uint32_t i = first; i < last; ++i) {
        const uint32_t slot = slot_of(i);
#pragma unroll
        for (uint32_t w = 0; w < kCostClasses / 2; ++w) packed[w] += (slot >> 1) == w ? (1u << ((slot & 1u) * 16u)) : 0u;
    }
    // (2) for every class: where this thread's members start = (members of more expensive classes) + (this class's members of the threads before)
    uint32_t mine[kCostClasses], before[kCostClasses];
#pragma unroll
    for (uint32_t k = 0; k < kCostClasses; ++k) {
        mine[k] = (packed[k >> 1] >> ((k & 1u) * 16u)) & 0xffffu;
        uint32_t inc = mine[k];  // inclusive scan over the wave's lanes
#pragma unroll
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = __shfl_up(inc, d, 64);
            inc += lane >= d ? up : 0u;
        }
        before[k] = inc - mine[k];
        if (lane == 63) wave_totals[k][wave] = inc;
    }
    __syncthreads();
    if (t < kCostClasses) {  // thread k: class k's total; then the classes' bases by a scan over sixteen values (one wave)
        uint32_t total = 0;
        for (uint32_t w = 0; w < kOrderThreads / 64; ++w) total += wave_totals[t][w];
        uint32_t inc = total;
#pragma unroll
        for (uint32_t d = 1; d < kCostClasses; d <<= 1) {
            const uint32_t up = __shfl_up(inc, d, 64);
            inc += t >= d ? up : 0u;
        }
        slot_base[t] = inc - total;
    }
    __syncthreads();
    uint32_t at[kCostClasses];
#pragma unroll
    for (uint32_t k = 0; k < kCostClasses; ++k) {
        uint32_t waves_before = 0;
        for (uint32_t w = 0; w < wave; ++w) waves_before += wave_totals[k][w];
        at[k] = slot_base[k] + waves_before + before[k];
    }
    // (3) every sub-tile of the run into its place
#pragma unroll 8
    for (uint32_t i = first; i < last; ++i) {
        const uint32_t slot = slot_of(i);
        uint32_t place = 0;
#pragma unroll
        for (uint32_t k = 0; k < kCostClasses; ++k) {
            place = slot == k ? at[k] : place;
            at[k] += slot == k ? 1u : 0u;
        }
        order[place] = i;
    }
}

// vx_commit's packed uploads: piece b of the table = {device address, offset in the packed payload, bytes}; source and
// destination agree modulo 16 (the packer pads), so the middle of a piece moves as 16-byte words
__global__ __launch_bounds__(256) void scatter_kernel(const uint64_t* __restrict__ table, const uint8_t* __restrict__ packed) {
    const uint64_t dst = table[blockIdx.x * 3], src = table[blockIdx.x * 3 + 1], len = table[blockIdx.x * 3 + 2];
    uint8_t* d = reinterpret_cast<uint8_t*>(dst);
    const uint8_t* s = packed + src;
    const uint32_t t = threadIdx.x;
    const uint64_t to_aligned = (16u - (dst & 15u)) & 15u;
    const uint32_t head = uint32_t(len < to_aligned ? len : to_aligned);
    if (t < head) d[t] = s[t];
    const uint64_t body = (len - head) / 16;
    const uint4* s4 = reinterpret_cast<const uint4*>(s + head);
    uint4* d4 = reinterpret_cast<uint4*>(d + head);
    for (uint64_t i = t; i < body; i += 256) d4[i] = s4[i];
    const uint32_t tail = uint32_t((len - head) & 15u);
    if (t < tail) d[head + body * 16 + t] = s[head + body * 16 + t];
}

// Scatters gathered compact tile lists back into a row-major image: ONE workgroup per 32x32 tile, a thread moves four neighbouring pixels
// of a row (RGBA8: one 16-byte word; RGBA32F: four) -- whole 4 KB / 16 KB tiles are read in order, whole 128- / 512-byte row segments
// written. (Round 2's version, a thread per pixel in 16x16 blocks, was 8160 workgroups at 1080p and took 230 us beside the render kernel's
// persistent waves: the next frame on its stream waits for it.) `inverse` = place of every tile in the Morton sequence the ranks share out
// (place j: rank j % tile_count, its local tile j / tile_count).
__global__ __launch_bounds__(256) void assemble_kernel(const float4* __restrict__ tiles, uint64_t stride_px, uint32_t tile_count, uint32_t width,
                                                       uint32_t height, uint32_t tiles_x, const uint32_t* __restrict__ inverse, float4* __restrict__ out) {
    const uint32_t tile = blockIdx.x, tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t j = inverse[tile];
    const float4* src = tiles + (j % tile_count) * stride_px + size_t(j / tile_count) * (kTile * kTile);
    const uint32_t row = threadIdx.x >> 3, x4 = (threadIdx.x & 7u) * 4u;  // 32 rows x 8 groups of four pixels
    const uint32_t y = ty * kTile + row, x = tx * kTile + x4;
    if (y >= height) return;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k)
        if (x + k < width) out[size_t(y) * width + x + k] = src[row * kTile + x4 + k];
}

// the same for RGBA8 tile lists and image (vx_target.format = VX_FORMAT_RGBA8; an RGBA8 image has its top row first)
__global__ __launch_bounds__(256) void assemble_kernel_rgba8(const uint32_t* __restrict__ tiles, uint64_t stride_px, uint32_t tile_count, uint32_t width,
                                                             uint32_t height, uint32_t tiles_x, const uint32_t* __restrict__ inverse, uint32_t* __restrict__ out) {
    const uint32_t tile = blockIdx.x, tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t j = inverse[tile];
    const uint32_t* src = tiles + (j % tile_count) * stride_px + size_t(j / tile_count) * (kTile * kTile);
    const uint32_t row = threadIdx.x >> 3, x4 = (threadIdx.x & 7u) * 4u;
    const uint32_t y = ty * kTile + row, x = tx * kTile + x4;
    if (y >= height) return;
    uint32_t* dst = out + size_t(height - 1u - y) * width + x;
    const uint4 v = *reinterpret_cast<const uint4*>(src + row * kTile + x4);  // (16-byte aligned: tiles are 4 KB, x4 a multiple of 4)
    if (x + 3 < width && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        *reinterpret_cast<uint4*>(dst) = v;
    } else {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        for (uint32_t k = 0; k < 4; ++k)
            if (x + k < width) dst[k] = w[k];
    }
}

}  // namespace

// =================================================================================================================
// host runtime
// =================================================================================================================

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(call)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (call);                                                                                    \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? VX_ERR_OUT_OF_MEMORY : VX_ERR_HIP,           \
                                          std::string(#call) + ": " + hipGetErrorString(e_));                     \
    } while (0)

struct ProfiledLaunch {
    hipEvent_t start, stop;
};

}  // namespace

struct vx_context {
    int svo_type = 0;
    int device = 0;
    size_t capacity = 0;
    uint8_t* staging = nullptr;   // pinned host mirror of the world buffer
    uint8_t* d_world = nullptr;
    hipStream_t stream = nullptr;         // render / raycast launches
    hipStream_t upload_stream = nullptr;  // range uploads
    hipEvent_t upload_done = nullptr, render_done = nullptr;
    bool committed = false, render_recorded = false;
    // Frames in flight: image-only renders into device memory rotate over `frames_in_flight` streams, so that the first waves
    // of the next frames fill the CUs the last long rays of frame k leave idle (each frame is one persistent kernel whose tail runs at
    // low occupancy). Everything else (picker, hit records, host targets, counters) stays on `stream`.
    static constexpr int kFrameStreams = 8;
    hipStream_t frame_stream[kFrameStreams] = {};
    hipEvent_t frame_done[kFrameStreams] = {};
    bool frame_recorded[kFrameStreams] = {};
    uint32_t* d_frame_counter[kFrameStreams] = {};
    // Launches on each stream so far: a stream's two sets of ticket dispensers take turns (PersistentArgs::work_counter).
    uint32_t frame_tickets[kFrameStreams] = {};
    uint32_t* d_frame_todo[kFrameStreams] = {};  // images of CSVO worlds: [chunk counter][ring of 128-dword chunks] per stream (PixelList)
    size_t frame_todo_chunks[kFrameStreams] = {};
    uint32_t* d_main_todo = nullptr;
    size_t main_todo_chunks = 0;
    uint32_t main_tickets = 0;
    // expensive sub-tiles first (PersistentArgs::order): per stream three generations of {cost per sub-tile, order table}. Frame j of a
    // view on a stream notes costs in generation j % 3; the order kernel for it runs on `order_stream`, behind the frame and beside
    // the next one; frame j + 2 draws its tickets through that table (every step ordered by events: nothing is read while written).
    struct HotState {
        uint32_t* cost[3] = {nullptr, nullptr, nullptr};
        uint32_t* order[3] = {nullptr, nullptr, nullptr};
        hipEvent_t order_done[3] = {nullptr, nullptr, nullptr};
        size_t subtiles = 0;        // capacity of each
        uint32_t tag = 0;           // of the generation written last
        uint32_t frames = 0;        // frames of the current view issued on this stream
        uint32_t width = 0, height = 0, tile_rank = 0, tile_count = 0;  // the view (0 = none)
    };
    hipStream_t order_stream = nullptr;
    HotState hot[kFrameStreams + 1];  // [slot + 1]
    // SORTED builds: per stream, the records and pass tables (PersistentArgs::pass_in / pass_out, perm_in / perm_out) of the view rendered there; each
    // pair takes turns
    struct SortState {
        uint32_t* rec[2] = {nullptr, nullptr};  // [unit][lane]: cost << 8 | pixel
        uint8_t* perm[2] = {nullptr, nullptr};  // [unit][lane]: pixel
        size_t units = 0;   // capacity
        int cur = 0;        // the tables the next frame reads
        uint32_t frames = 0;  // frames of the view issued on this stream
        // The uniforms of the last image-only frame on this stream: passes are sorted by what pixels cost in EARLIER frames, which says something
        // about this frame only if the view has not moved -- under a camera that turns by a tenth of a degree a frame the stale passes are 8 %
        // slower than plain sub-tiles (a random 64 of a block's 256 pixels hold one of its long rays almost surely; 8 x 8 neighbours often do
        // not: profiles/round3/pass_aq). So a frame is rendered by a SORTED build only if its uniforms are the last frame's, bit for bit.
        vx_uniforms last_u = {};
        bool last_u_valid = false;
        bool live = false;    // the last frame here was a SORTED build's: its records and passes are there
        uint32_t width = 0, height = 0, tile_rank = 0, tile_count = 0;  // the view (0 = none)
    };
    SortState sorted_state[kFrameStreams + 1];  // [slot + 1]
    bool sorted_passes = true;                  // VX_SORTED=0: the unsorted builds everywhere (A/B)
    bool sorted_always = false;                 // VX_SORTED=2: sorted passes also for views that move (measurement)
    uint32_t sort_mask = 3;                     // VX_SORT_PERIOD (a power of two, default 4; VX_SORT_EVERY_FRAME=1 = 1): a block is re-sorted every so many frames of its stream
    bool hot_first = true;            // VX_HOT_FIRST=0: sub-tiles in plain order (A/B)
    bool hot_use = true, hot_note = true, hot_sort = true;  // VX_HOT_FIRST bits (measurement): 1 use the table, 2 note costs, 4 run the order kernel
    hipEvent_t pending_wait = nullptr;  // vx_wait_event: what the next pipelined render has to wait for
    hipEvent_t pending_gather = nullptr;  // vx_wait_gather: the gather that still reads the tile list the next render overwrites
    int frames_in_flight = 2;           // 1 serialises frames on `stream` again (vx_set_frames_in_flight / VX_FRAMES_IN_FLIGHT)
    unsigned frame_index = 0;
    int last_frame_slot = -1;           // slot of the most recent pipelined render, -1 = it ran on `stream`
    vx_stats stats = {};

    vx_material* d_materials = nullptr;
    uint32_t n_materials = 0;
    uint8_t* d_tex = nullptr;
    uint32_t tex_bytes = 0;
    struct { uint32_t width, height, layers, levels, level_offset[16]; } tex = {};

    // scratch
    float* d_frame = nullptr;  size_t d_frame_bytes = 0;
    vx_hit* d_hits = nullptr;  size_t d_hits_bytes = 0;
    vx_picker_task* d_tasks = nullptr;  vx_picker_result* d_results = nullptr;  uint32_t picker_cap = 0;
    vx_result* d_trace_result = nullptr;  vx_frame* d_trace_frames = nullptr;  uint32_t* d_trace_count = nullptr;  uint32_t trace_cap = 0;
    unsigned long long* d_counters = nullptr;

    uint32_t* d_work_counter = nullptr;
    unsigned long long* d_excursions = nullptr;  // [4], see PersistentArgs
    bool count_excursions = false;               // vx_excursion_counters(.., 1) switches the counting on (it costs: see render_persistent)
    unsigned long long* d_timeline = nullptr;    // VX_TIMELINE=1: [8192][8], the last launch's waves (PersistentArgs::timeline)
    uint32_t timeline_waves = 0;
    uint32_t timeline_part = 0;   // VX_TIMELINE_PART
    uint32_t ahead_guard = 2;     // VX_AHEAD_GUARD: quarter-grids of tickets at the end of a frame that are not drawn ahead
    int ticket_ahead = 1;         // VX_TICKET_AHEAD=0: waves draw a sub-tile's ticket when they need it (A/B)
    // the traversal image of the world (traversal_image.hpp), rebuilt for the changed chunks by every commit
    vximg::WorldImage image;
    uint8_t* d_image = nullptr;
    size_t d_image_capacity = 0;
    uint8_t* d_origin = nullptr;  // CSVO worlds: the image's origin table (a quarter of the image's size)
    size_t d_origin_capacity = 0;
    size_t image_cap_bytes = 0;   // VX_IMAGE_CAP_BYTES: never allocate more than this for the image (tests of the fall-back)
    // vx_commit's packed uploads: a small ring of pinned host buffers with their device twins, each guarded by an event
    struct DeltaSlot { uint8_t* host = nullptr; uint8_t* dev = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; };
    static constexpr int kDeltaSlots = 3;
    DeltaSlot delta[kDeltaSlots];
    unsigned delta_next = 0;
    bool hot_levels = false;      // VX_HOT_LEVELS=1 (experiment X1): the image's top two levels served from an LDS copy (ESVO worlds, image-only renders)
    int foreign_rerun = -1;       // VX_FOREIGN_RERUN: 1 / 0 = image-only renders of a CSVO world always / never list their inside-voxel rays for the bytes
                                  // (kForeignRerun) instead of making the excursion in the render loop; default: worlds of at most 12 levels do
    bool batch_service = false;   // VX_BATCH=1 (experiment): image-only renders by the build that shades, lights and generates rays 64 records at a time
    uint8_t* d_batch[kFrameStreams + 1] = {};  // [slot + 1]: the waves' record rings of a BATCH kernel (PersistentArgs::batch)
    size_t batch_waves[kFrameStreams + 1] = {};
    uint32_t cost_floor = 64, cost_step = kCostStep;  // VX_COST_FLOOR, VX_COST_STEP: the order table's classes (note_cost_wave, order_kernel)
    int deep_waves = 4;           // VX_DEEP_WAVES=3 (experiment, 4-7 % slower): the kernel with the excursion code (CSVO worlds of 13 and 14 levels) at three waves per SIMD, 168 VGPRs
    bool five_waves = false;      // VX_FIVE_WAVES=1 (experiment, 3 % slower): images of up to 12 levels on a 12-level stack with a 16-bit third plane, five waves per SIMD
    bool deep_stack = true;       // VX_DEEP_STACK=0: images of 14 to 16 levels on the 13-level stack with the hand-over (A/B)
    bool no_excursion = false;    // VX_NO_EXCURSION=1 (MEASUREMENT ONLY, wrong pixels): a CSVO world's image walked by the kernel without the excursion code
    bool big = false;             // an ESVO world buffer of 4 GiB and more: kernels on its own bytes use 64-bit addresses (VX_SVO_ESVO_BIG)
    bool image_enabled = true;  // VX_TRAVERSAL_IMAGE=0: traverse the world's own bytes
    bool image_ok = false;
    int kernel_version = 2;               // 2 = persistent wavefront kernel, 1 = one thread per pixel (kept for A/B runs)
    // service_min = 64: a wave's lanes move in LOCKSTEP -- a sub-tile's 64 primary rays are traversed until the last of them has ended, then
    // served together (hits shaded, misses painted), then the shadow rays, then the pixels lit and stored and the next sub-tile taken. Until
    // round 3 it was 32 (lanes served when half the wave waited). With the hand-scheduled loop a trip costs a third less, and what a wave
    // pays for is its service phases: 23 per frame at 32, 14 at 56, 12 at 64 -- and at 64 only every batch of rays is ONE sub-tile's (at 63
    // a straggler carried into the next round shifts every later batch across two sub-tiles: 8 rounds for 7 sub-tiles): 9.2 -> 10.9 Grays/s
    // from 63 to 64 (profiles/round3/pass_m, pass_o). Lanes whose rays end early idle until the slowest ray of the batch has ended (the rays
    // of 8 x 8 neighbouring pixels are of similar length: 41 % of the loop's lane slots are used, against 62 % at 56 -- in fewer instructions).
    // (foreign_min -- deep CSVO worlds: how many lanes must wait for their walk into a voxel before a wave leaves the loop for it: 32 in
    // round 2; what a frame pays for is the number of such phases, profiles/round3/pass_s/foreign_min.txt, pass_y with held rays)
    uint32_t refill_min = 4, service_min = 64, foreign_min = 40;
    // Block ids 0..63 all of whose textures are opaque throughout (RenderParams::opaque_*): from host copies of what vx_set_materials and
    // vx_set_textures were given. opaque_layer[l] = every texel of layer l, on every mip level, has alpha > 0.
    std::vector<vx_material> host_materials;
    std::vector<uint8_t> opaque_layer;
    uint64_t opaque_blocks = 0;
    int min_waves = 4;                    // 4 = the image-only kernel is the build for 4 waves per SIMD (<= 128 VGPRs); 1 = compiler's choice
    int waves_per_cu_cap = 0;             // experiment: fewer persistent waves than the occupancy limit
    int comm_headroom = 4;                // VX_COMM_HEADROOM: wave slots per CU a context with a communicator of more than one rank leaves free (LDS for RCCL's kernels)
    int cu_count = 256;
    // VX_COMM_RESERVE_CUS=n (read at vx_create; bench.py sets it for ranks that exchange tiles): the streams renders run on are created with a CU
    // mask that leaves n compute units out -- whole CUs for the exchange's and the assembly's workgroups instead of `comm_headroom` wave slots on
    // every CU -- and a launch has (cu_count - n) x 16 persistent waves
    int reserve_cus = 0;
    std::unordered_map<const void*, int> persistent_blocks;  // kernel -> resident 64-thread workgroups per CU, queried once

    // screen sharding: the Morton order of an image's tiles and its inverse, on the device, per image size seen
    struct TileTable { uint32_t tiles_x = 0, tiles_y = 0; uint32_t* d_order = nullptr; uint32_t* d_inverse = nullptr; };
    std::vector<TileTable> tile_tables;

    // multi-GPU: the RCCL communicator over which the finished tiles are gathered (vx_comm_init), its stream and the events that
    // say when a gather has read its tile list
    ncclComm_t comm = nullptr;
    int comm_ranks = 0, comm_rank = 0;
    hipStream_t comm_stream = nullptr;
    static constexpr int kGatherEvents = 16;
    hipEvent_t gather_done[kGatherEvents] = {};
    unsigned gather_index = 0;
    std::vector<ProfiledLaunch> gathers;  // vx_profile_enable: the exchanges' event pairs (vx_comm_profile_read)

    // pipelined presentation (vx_present_begin / vx_present_wait): per slot a device frame and its pinned host twin; the read-back
    // runs on its own stream behind the frame's kernel, beside the next frame's
    static constexpr int kPresentSlots = 4;
    struct PresentSlot { void* dev = nullptr; void* host = nullptr; size_t cap = 0, bytes = 0; hipEvent_t copied = nullptr; bool busy = false; };
    PresentSlot present[kPresentSlots];
    unsigned present_next = 0;
    hipStream_t copy_stream = nullptr;

    bool profile = false;
    std::vector<ProfiledLaunch> launches;
    std::vector<ProfiledLaunch> event_pool;

    // Every entry point that queues device work, or reads what a commit publishes, holds `mutex` while it does. A context is still
    // driven by ONE caller thread; the second party is the context's own commit worker (vx_set_commit_mode), which takes the
    // mutex for the part of a commit that touches the device and the published state -- so a render is either wholly before a
    // commit (which then waits for it on the device) or wholly after it (and waits for its uploads).
    std::recursive_mutex mutex;
    // what renders need to know of the traversal image, as of the last commit that reached the device (`image` itself belongs to
    // whoever runs the commit)
    struct ImagePublished { uint64_t frame_bytes = 0, origin_bytes = 0, chunks = 0; uint32_t depth = 0; vximg::Layout layout = vximg::kOct64; } pub;
    // pipelined commits: one posted job at a time, run by `worker`
    int commit_mode = VX_COMMIT_INLINE;
    struct CommitJob { uint32_t depth = 0; std::vector<vx_range> ranges; uint64_t used_bytes = 0; } job;
    std::thread worker;
    std::mutex job_mutex;
    std::condition_variable job_cv;
    bool job_posted = false, job_running = false, worker_stop = false;
    int async_rc = VX_OK;       // of the last pipelined commit, reported by the next vx_commit / vx_commit_wait / vx_sync
    std::string async_error;
};

#define VX_LOCK(ctx) std::lock_guard<std::recursive_mutex> vx_lock_((ctx)->mutex)

namespace {
void wait_commit_idle(vx_context* ctx);
void stop_commit_worker(vx_context* ctx);
}  // namespace

namespace {

constexpr size_t kWorldPad = 16;
constexpr size_t kImagePad = 64;  // zero bytes behind the image: an 8-byte entry load at the last octant's last child stays inside
constexpr size_t kStagingSlack = 64;

uint32_t header_bytes(const vx_context* c) { return c->svo_type == VX_SVO_ESVO ? 20u : 4u; }

SceneArgs scene_of(const vx_context* c) {
    SceneArgs s = {};
    s.world = c->d_world;
    s.world_bytes = uint64_t(c->capacity) + kWorldPad;
    s.materials = c->d_materials;
    s.n_materials = c->n_materials;
    s.tex = c->d_tex;
    s.tex_bytes = c->tex_bytes;
    s.width = c->tex.width; s.height = c->tex.height; s.layers = c->tex.layers; s.levels = c->tex.levels;
    for (int l = 0; l < 16; ++l) s.level_offset[l] = c->tex.level_offset[l];
    s.image = c->image_ok ? c->d_image : nullptr;
    s.image_bytes = c->image_ok ? c->pub.frame_bytes + kImagePad : 0u;
    s.origin = c->image_ok ? c->d_origin : nullptr;
    return s;
}

// every stream a kernel of this context can be running on has drained (before freeing or replacing what kernels read)
int drain_streams(vx_context* c) {
    if (c->upload_stream) HIP_TRY(hipStreamSynchronize(c->upload_stream));
    if (c->stream) HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < vx_context::kFrameStreams; ++i)
        if (c->frame_stream[i]) HIP_TRY(hipStreamSynchronize(c->frame_stream[i]));
    if (c->copy_stream) HIP_TRY(hipStreamSynchronize(c->copy_stream));
    if (c->comm_stream) HIP_TRY(hipStreamSynchronize(c->comm_stream));
    if (c->order_stream) HIP_TRY(hipStreamSynchronize(c->order_stream));
    return VX_OK;
}

// Morton (Z-order) sequence of an image's 32x32 tiles: tile (tx, ty) sorts by the interleaved bits of its coordinates, so any run
// of consecutive places covers a compact patch of the screen and the ranks that share the places out round-robin each get an
// even sample of every region. order[j] = row-major id of the tile at place j; inverse[id] = j.
void tile_order_host(uint32_t tiles_x, uint32_t tiles_y, std::vector<uint32_t>& order, std::vector<uint32_t>& inverse) {
    auto spread = [](uint32_t v) {  // bits of a 16-bit value to the even bit positions
        v &= 0xffffu;
        v = (v | (v << 8)) & 0x00ff00ffu;
        v = (v | (v << 4)) & 0x0f0f0f0fu;
        v = (v | (v << 2)) & 0x33333333u;
        return (v | (v << 1)) & 0x55555555u;
    };
    const uint32_t n = tiles_x * tiles_y;
    std::vector<uint64_t> keyed(n);
    for (uint32_t t = 0; t < n; ++t) keyed[t] = (uint64_t(spread(t % tiles_x) | (spread(t / tiles_x) << 1)) << 32) | t;
    std::sort(keyed.begin(), keyed.end());
    order.resize(n);
    inverse.resize(n);
    for (uint32_t j = 0; j < n; ++j) {
        order[j] = uint32_t(keyed[j]);
        inverse[order[j]] = j;
    }
}

int tile_table(vx_context* ctx, uint32_t tiles_x, uint32_t tiles_y, const vx_context::TileTable** out) {
    for (const auto& t : ctx->tile_tables)
        if (t.tiles_x == tiles_x && t.tiles_y == tiles_y) {
            *out = &t;
            return VX_OK;
        }
    std::vector<uint32_t> order, inverse;
    tile_order_host(tiles_x, tiles_y, order, inverse);
    vx_context::TileTable t;
    t.tiles_x = tiles_x;
    t.tiles_y = tiles_y;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t.d_order), order.size() * 4));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&t.d_inverse), inverse.size() * 4));
    HIP_TRY(hipMemcpy(t.d_order, order.data(), order.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(t.d_inverse, inverse.data(), inverse.size() * 4, hipMemcpyHostToDevice));
    ctx->tile_tables.reserve(16);  // (pointers into the vector are handed out: a context sees a handful of sizes)
    ctx->tile_tables.push_back(t);
    *out = &ctx->tile_tables.back();
    return VX_OK;
}

int ensure(void** p, size_t* have, size_t need) {
    if (*have >= need && *p) return VX_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *have = 0;
    HIP_TRY(hipMalloc(p, need));
    *have = need;
    return VX_OK;
}

int check_ready(vx_context* ctx) {
    if (!ctx) return fail(VX_ERR_INVALID_ARGUMENT, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    if (!ctx->committed) return fail(VX_ERR_STATE, "no SVO committed yet (call vx_commit / vx_commit_all first)");
    return VX_OK;
}

// The persistent render kernel for a context's world: on the world's own bytes (ESVO, ESVO beyond 4 GiB, CSVO), or on its
// traversal image (byte-offset or wide layout; with the excursion onto the bytes for CSVO worlds; without the stack hand-over test
// where the image's depth rules deep pushes out). The image-only build of the hot variants is held to 128 VGPRs (4 waves per SIMD).
// levels: the LDS-resident stack levels of an image kernel -- kLdsLevels, or 16 (16-bit third plane) for images of 14 to 16 levels
template <bool HITS, bool STATS>
const void* persistent_kernel(const vx_context* ctx, bool imaged, bool shallow, int levels, bool batch, bool rerun, bool still, bool* sorted) {
    const bool esvo = ctx->svo_type == VX_SVO_ESVO;
    *sorted = false;
#define VX_K(...) reinterpret_cast<const void*>(&render_persistent<__VA_ARGS__>)
    // The SORTED builds (the queue's units are passes: 64 pixels of a block of four sub-tiles, chosen by earlier frames' costs): image-only renders of a
    // view that stands still (`still`), with the lanes in lockstep,
    // on 13- and 16-level stacks, of worlds whose image needs no excursion (ESVO; CSVO of at most 12 levels, which list such rays)
    if constexpr (!HITS && !STATS) {
        if (ctx->sorted_passes && still && ctx->service_min >= 64 && imaged && shallow && !batch && (levels == kLdsLevels || levels == 16) && !ctx->hot_levels &&
            (esvo || ctx->no_excursion || (rerun && levels == kLdsLevels))) {
            const bool wide = ctx->pub.layout == vximg::kOct64Wide;
            const bool tl = ctx->d_timeline != nullptr;
            *sorted = true;
#define VX_SK(IMAGE, FOREIGN, LV) (tl ? VX_K(IMAGE, false, false, 4, FOREIGN, true, LV, false, false, true, true) : VX_K(IMAGE, false, false, 4, FOREIGN, true, LV, false, false, false, true))
            if (esvo || ctx->no_excursion) return levels == 16 ? (wide ? VX_SK(VX_SVO_IMAGE_WIDE, 0, 16) : VX_SK(VX_SVO_IMAGE, 0, 16)) : (wide ? VX_SK(VX_SVO_IMAGE_WIDE, 0, kLdsLevels) : VX_SK(VX_SVO_IMAGE, 0, kLdsLevels));
            return wide ? VX_SK(VX_SVO_IMAGE_WIDE, kForeignRerun, kLdsLevels) : VX_SK(VX_SVO_IMAGE, kForeignRerun, kLdsLevels);
#undef VX_SK
        }
    }
    // VX_TIMELINE=1: the instrumented builds of the image-only kernels on 13- and 16-level stacks (what profiles/timeline.py renders with); any
    // other kernel leaves the timeline's rows untouched
    if constexpr (!HITS && !STATS) {
        if (ctx->d_timeline && imaged && shallow && !batch && (levels == kLdsLevels || levels == 16) && !ctx->hot_levels && ctx->deep_waves != 3) {
            const bool wide = ctx->pub.layout == vximg::kOct64Wide;
#define VX_TLK(IMAGE, FOREIGN) (levels == 16 ? VX_K(IMAGE, false, false, 4, FOREIGN, true, 16, false, false, true) : VX_K(IMAGE, false, false, 4, FOREIGN, true, kLdsLevels, false, false, true))
            if (esvo || ctx->no_excursion) return wide ? VX_TLK(VX_SVO_IMAGE_WIDE, 0) : VX_TLK(VX_SVO_IMAGE, 0);
            if (rerun && levels == kLdsLevels) return wide ? VX_K(VX_SVO_IMAGE_WIDE, false, false, 4, kForeignRerun, true, kLdsLevels, false, false, true) : VX_K(VX_SVO_IMAGE, false, false, 4, kForeignRerun, true, kLdsLevels, false, false, true);
            return wide ? VX_TLK(VX_SVO_IMAGE_WIDE, VX_SVO_CSVO) : VX_TLK(VX_SVO_IMAGE, VX_SVO_CSVO);
#undef VX_TLK
        }
    }
    if (!imaged) {
        constexpr int W = (!HITS && !STATS) ? 4 : 1;
        if (!HITS && !STATS && ctx->min_waves != 4)
            return esvo ? (ctx->big ? VX_K(VX_SVO_ESVO_BIG, false, false, 1) : VX_K(VX_SVO_ESVO, false, false, 1)) : VX_K(VX_SVO_CSVO, false, false, 1);
        return esvo ? (ctx->big ? VX_K(VX_SVO_ESVO_BIG, HITS, STATS, W) : VX_K(VX_SVO_ESVO, HITS, STATS, W)) : VX_K(VX_SVO_CSVO, HITS, STATS, W);
    }
    constexpr int W = HITS ? 1 : 4;
    const bool wide = ctx->pub.layout == vximg::kOct64Wide;  // an image beyond 4 GiB: octant indices, 64-bit addresses
#define VX_IMG(IMAGE, FOREIGN)                                                                                                       \
    (batch ? (levels == 16 ? VX_K(IMAGE, false, false, 4, FOREIGN, true, 16, false, true)                                           \
                           : (shallow ? VX_K(IMAGE, false, false, 4, FOREIGN, true, kLdsLevels, false, true) : VX_K(IMAGE, false, false, 4, FOREIGN, false, kLdsLevels, false, true))) \
           : (levels == 12 ? VX_K(IMAGE, false, false, 5, FOREIGN, true, 12)                                                        \
                           : (levels == 16 ? VX_K(IMAGE, HITS, false, W, FOREIGN, true, 16) : (shallow ? VX_K(IMAGE, HITS, false, W, FOREIGN, true) : VX_K(IMAGE, HITS, false, W, FOREIGN, false)))))
    if (ctx->hot_levels && !HITS && !wide && shallow && levels == kLdsLevels && (esvo || ctx->no_excursion)) return VX_K(VX_SVO_IMAGE, false, false, 4, 0, true, kLdsLevels, true);
    if (esvo || ctx->no_excursion) return wide ? VX_IMG(VX_SVO_IMAGE_WIDE, 0) : VX_IMG(VX_SVO_IMAGE, 0);
    if constexpr (!HITS && !STATS) {
        // image-only renders of a CSVO world: rays that start inside a voxel are listed and run on the world's bytes afterwards (kForeignRerun)
        if (rerun && shallow && levels == kLdsLevels) return wide ? VX_K(VX_SVO_IMAGE_WIDE, false, false, 4, kForeignRerun, true) : VX_K(VX_SVO_IMAGE, false, false, 4, kForeignRerun, true);
    }
    if constexpr (!HITS && !STATS) {
        if (ctx->deep_waves == 3 && !batch && shallow && levels == kLdsLevels)
            return wide ? VX_K(VX_SVO_IMAGE_WIDE, false, false, 3, VX_SVO_CSVO, true) : VX_K(VX_SVO_IMAGE, false, false, 3, VX_SVO_CSVO, true);
    }
    return wide ? VX_IMG(VX_SVO_IMAGE_WIDE, VX_SVO_CSVO) : VX_IMG(VX_SVO_IMAGE, VX_SVO_CSVO);
#undef VX_IMG
#undef VX_K
}

template <bool HITS, bool STATS>
int launch_render(vx_context* ctx, const RenderParams& p, float* out, vx_hit* hits, unsigned long long* counters, int slot = -1) {
    const hipStream_t stream = slot >= 0 ? ctx->frame_stream[slot] : ctx->stream;
    uint32_t* const work_counter = slot >= 0 ? ctx->d_frame_counter[slot] : ctx->d_work_counter;
    const size_t lds = Stack<kBlockThreads>::kBytes;
    const dim3 grid(p.n_local_tiles * 4), block(kBlockThreads);
    if (grid.x == 0) return VX_OK;
    const SceneArgs sc = scene_of(ctx);

    uint32_t& tickets = slot >= 0 ? ctx->frame_tickets[slot] : ctx->main_tickets;
    vx_context::HotState* order_after = nullptr;
    uint32_t order_subtiles = 0;
    ProfiledLaunch ev{};
    if (ctx->profile) {
        if (!ctx->event_pool.empty()) {
            ev = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else {
            HIP_TRY(hipEventCreate(&ev.start));
            HIP_TRY(hipEventCreate(&ev.stop));
        }
        HIP_TRY(hipEventRecord(ev.start, stream));
    }
    if (ctx->kernel_version == 1 && !ctx->big) {
        if (ctx->svo_type == VX_SVO_ESVO)
            hipLaunchKernelGGL((render_kernel<VX_SVO_ESVO, HITS, STATS>), grid, block, lds, stream, sc, p, reinterpret_cast<float4*>(out), hits, counters);
        else
            hipLaunchKernelGGL((render_kernel<VX_SVO_CSVO, HITS, STATS>), grid, block, lds, stream, sc, p, reinterpret_cast<float4*>(out), hits, counters);
    } else {
        // persistent waves: as many 64-thread workgroups as the device keeps resident, fed from the sub-tile queue
        // Worlds are rendered from their traversal image; the instrumented variant stays on the world's own bytes so that its
        // counters are the reference's own fetches
        // (... and only for textures whose height is a power of two: the image kernels' sampler wraps with a mask. Any other height renders
        // on the world's own bytes, whose kernels carry the general wrap.)
        const bool imaged = !STATS && ctx->image_ok && (ctx->tex.height & (ctx->tex.height - 1u)) == 0u;
        // The image holds at most `depth` levels (traversal_image.hpp). The deepest PUSH of a ray on the image of an ESVO world is
        // into a voxel (a ray that started inside it walks it as an empty node), out of a node at scale 23 - depth; on the image of a
        // CSVO world such a ray leaves for its excursion instead, and the deepest PUSH is one level higher. Where that is an LDS
        // resident slot (scales >= kLdsBaseScale) the loop needs no hand-over test.
        const uint32_t depth = ctx->pub.depth;
        const uint32_t slack = ctx->svo_type == VX_SVO_CSVO ? 1u : 0u;
        bool shallow = imaged && depth <= uint32_t(kLdsLevels) + slack;
        // deeper images, up to 16 levels: the kernel build with 16 resident levels (16-bit third stack plane) -- no hand-over either
        int levels = kLdsLevels;
        if (imaged && !shallow && depth <= 16u + slack && ctx->deep_stack) {
            levels = 16;
            shallow = true;
        }
        // Deep CSVO worlds (13 levels and more: half of the shadow rays or all of them start inside their voxel and walk it on the world's bytes,
        // walk_voxel_on_bytes): the 16-level stack as well -- what the walk pushes inside a voxel then lands in LDS slots (three levels below
        // the voxel's parent at depth 13, two at 14: 96 % / 82 % of the walks go no deeper) instead of the scratch-backed spill array
        if (imaged && ctx->svo_type == VX_SVO_CSVO && depth >= 13u && depth <= 16u + slack && ctx->deep_stack) {
            levels = 16;
            shallow = true;
        }
        // shallower images, up to 12 levels, image-only renders: 12 resident levels in 7.5 KB, five waves per SIMD
        if (!HITS && shallow && levels == kLdsLevels && depth <= 12u + slack && ctx->five_waves && !ctx->hot_levels) levels = 12;
        // image-only renders: the build with batched service phases (render_persistent, BATCH)
        const bool batch = imaged && !HITS && ctx->batch_service && levels != 12 && !ctx->hot_levels;
        // Which way a CSVO world's inside-voxel rays go is a matter of how many there are, and that of the world's size: below 4096 the
        // reference's 0.001 shadow offset survives rounding and they are a few dozen per frame -- listed, and run on the bytes when
        // the wave is done (kForeignRerun: +8 % at C3, for a render loop without the walk's code). From 8192 on every second shadow ray
        // is one (2 M per 4K frame): there the excursion inside the render loop, batched, is a fifth faster than running them whole on
        // the bytes (profiles/round2/foreign_rerun/).
        const bool rerun = imaged && ctx->svo_type == VX_SVO_CSVO && !HITS && !STATS && !batch && shallow && levels == kLdsLevels && !ctx->no_excursion &&
                           (ctx->foreign_rerun == 1 || (ctx->foreign_rerun < 0 && depth <= 12u));
        // (a view that has not moved since the last image-only frame on this stream -- or VX_SORTED=2: always -- may be rendered in sorted passes)
        vx_context::SortState& ss_view = ctx->sorted_state[slot + 1];
        // (... and only if it casts shadow rays: primary rays alone are too much of a length for sorting to pay -- 7 % fewer trips against 18 %,
        // less than the tables cost: C2 0.178 against 0.172 ms)
        const bool still = !HITS && !STATS && p.u.render_shadows != 0 &&
                           (ctx->sorted_always || (ss_view.last_u_valid && std::memcmp(&ss_view.last_u, &p.u, sizeof(vx_uniforms)) == 0));
        if (!HITS && !STATS) {
            ss_view.last_u = p.u;
            ss_view.last_u_valid = true;
        }
        bool sorted = false;
        const void* fn = persistent_kernel<HITS, STATS>(ctx, imaged, shallow, levels, batch, rerun, still, &sorted);
        if (!sorted && !HITS && !STATS) ss_view.live = false;
        size_t wave_lds = levels == 16 ? Stack<64, false, false, 16, true>::kBytes : (levels == 12 ? Stack<64, false, false, 12, true>::kBytes : Stack<64>::kBytes);
        if (fn == reinterpret_cast<const void*>(&render_persistent<VX_SVO_IMAGE, false, false, 4, 0, true, kLdsLevels, true>))
            wave_lds = Stack<64, false, false, kLdsLevels, true, true>::kBytes;
        int& per_cu = ctx->persistent_blocks[fn];
        if (per_cu == 0) {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 64, wave_lds) != hipSuccess || n <= 0) n = 8;
            per_cu = n;
        }
        PersistentArgs a;
        // (`tickets` counts this stream's launches: its sets of dispensers take turns)
        a.work_counter = work_counter + size_t(tickets & 1u) * (kQueues * kQueueStride);
        a.next_counter = work_counter + size_t((tickets & 1u) ^ 1u) * (kQueues * kQueueStride);
        a.total_subtiles = p.n_local_tiles * 16;  // (the queue's units: sub-tiles, or a SORTED build's passes)
        a.pass_in = nullptr;
        a.pass_out = nullptr;
        a.perm_in = nullptr;
        a.perm_out = nullptr;
        a.sort_turn = 0;
        a.sort_mask = 3;
        if (sorted) {
            vx_context::SortState& ss = ctx->sorted_state[slot + 1];
            const size_t units = a.total_subtiles;
            const bool same = ss.live && ss.width == p.width && ss.height == p.height && ss.tile_rank == p.tile_rank && ss.tile_count == p.tile_count && ss.units >= units;
            ss.live = true;
            if (ss.units < units) {  // (grow: earlier frames of this stream use the old tables)
                HIP_TRY(hipStreamSynchronize(stream));
                void* old_tables[] = {ss.rec[0], ss.rec[1], ss.perm[0], ss.perm[1]};
                for (void* q : old_tables)
                    if (q) (void)hipFree(q);
                ss.rec[0] = ss.rec[1] = nullptr;
                ss.perm[0] = ss.perm[1] = nullptr;
                ss.units = 0;
                const size_t cap = units + units / 4 + 64;
                for (int g = 0; g < 2; ++g) {
                    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ss.rec[g]), cap * 64 * sizeof(uint32_t)));
                    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ss.perm[g]), cap * 64));
                }
                ss.units = cap;
            }
            if (!same) {  // a view's first frame on this stream: sub-tile by sub-tile
                const uint32_t n = uint32_t(units * 64);
                hipLaunchKernelGGL(pass_identity_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, ss.rec[ss.cur], ss.perm[0], ss.perm[1], n);
                HIP_TRY(hipGetLastError());
                ss.width = p.width; ss.height = p.height; ss.tile_rank = p.tile_rank; ss.tile_count = p.tile_count;
            }
            a.pass_in = ss.rec[ss.cur];
            a.pass_out = ss.rec[ss.cur ^ 1];
            a.perm_in = ss.perm[ss.cur];
            a.perm_out = ss.perm[ss.cur ^ 1];
            if (!same) ss.frames = 0;
            a.sort_turn = ss.frames & 0xffffu;
            a.sort_mask = ctx->sort_mask;
            ss.frames += 1;
            ss.cur ^= 1;
        }
        a.refill_min = ctx->refill_min;
        a.service_min = ctx->service_min;
        a.foreign_min = ctx->foreign_min;
        a.excursions = ctx->count_excursions ? ctx->d_excursions : nullptr;
        a.timeline = ctx->d_timeline;
        a.timeline_part = ctx->timeline_part;
        a.ticket_ahead = ctx->ticket_ahead != 0 ? 1u + ctx->ahead_guard : 0u;
        a.order = nullptr;
        a.cost_cur = nullptr;
        a.cost_floor = ctx->cost_floor;
        a.cur_tag = 0xfffffu;
        vx_context::HotState* hs = nullptr;
        // One frame at a time only (the context's own stream): with several frames in flight the next frame's waves fill the tail anyway,
        // and noting costs (+6 %) and sorting them (+7 %: four wave slots for most of a frame) would be all cost (profiles/round2).
        if (ctx->hot_first && !STATS && slot < 0) {
            hs = &ctx->hot[slot + 1];
            const size_t n_sub = a.total_subtiles;
            if (!ctx->order_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->order_stream, hipStreamNonBlocking));
            if (hs->subtiles < n_sub) {  // (grow: earlier frames of this stream and their order kernels use the old arrays)
                HIP_TRY(hipStreamSynchronize(stream));
                HIP_TRY(hipStreamSynchronize(ctx->order_stream));
                for (int g = 0; g < 3; ++g) {
                    if (hs->cost[g]) (void)hipFree(hs->cost[g]);
                    if (hs->order[g]) (void)hipFree(hs->order[g]);
                    hs->cost[g] = hs->order[g] = nullptr;
                }
                hs->subtiles = 0;
                const size_t cap = n_sub + n_sub / 4 + 1024;
                for (int g = 0; g < 3; ++g) {
                    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&hs->cost[g]), cap * 4));
                    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&hs->order[g]), cap * 4));
                    HIP_TRY(hipMemsetAsync(hs->cost[g], 0, cap * 4, stream));
                    if (!hs->order_done[g]) HIP_TRY(hipEventCreateWithFlags(&hs->order_done[g], hipEventDisableTiming));
                }
                hs->subtiles = cap;
                hs->width = 0;
                hs->tag = 0;
            }
            if (hs->tag >= 0xffff0u) {  // (tags are 20 bits: start over once in a million frames)
                HIP_TRY(hipStreamSynchronize(ctx->order_stream));
                for (int g = 0; g < 3; ++g) HIP_TRY(hipMemsetAsync(hs->cost[g], 0, hs->subtiles * 4, stream));
                hs->tag = 0;
                hs->width = 0;
            }
            const bool same_view = hs->width == p.width && hs->height == p.height && hs->tile_rank == p.tile_rank && hs->tile_count == p.tile_count;
            if (!same_view) hs->frames = 0;
            if (hs->frames >= 2 && ctx->hot_sort) {  // the table made from this view's frame before last on this stream (its kernel had a whole frame's time)
                const uint32_t g = (hs->frames - 2) % 3;
                HIP_TRY(hipStreamWaitEvent(stream, hs->order_done[g], 0));
                if (ctx->hot_use) a.order = hs->order[g];
            } else if (hs->frames == 0 && hs->order_done[0]) {
                // a new view starts over in generation 0: what the old view's order kernels still have to write comes first
                for (int g = 0; g < 3; ++g) HIP_TRY(hipStreamWaitEvent(stream, hs->order_done[g], 0));
            }
            hs->tag += 1;
            if (ctx->hot_note) a.cost_cur = hs->cost[hs->frames % 3];
            a.cur_tag = hs->tag;
            hs->width = p.width; hs->height = p.height; hs->tile_rank = p.tile_rank; hs->tile_count = p.tile_count;
        }
        // Persistent waves per CU: all that fit -- the stacks fill a CU's LDS to the last hundred bytes. A context that gathers its tiles over
        // RCCL leaves `comm_headroom` (4) of them out: 40 KB of every CU's LDS stay free, in one piece, for the communication kernels'
        // workgroups, which otherwise find room only when a whole frame has drained (measured cost of twelve waves instead of sixteen: 8 %
        // of a rank's render rate; the gather itself could not be measured here -- one GPU per box).
        int per_cu_used = per_cu;
        if (ctx->waves_per_cu_cap > 0 && ctx->waves_per_cu_cap < per_cu) per_cu_used = ctx->waves_per_cu_cap;
        else if (ctx->reserve_cus == 0 && ctx->comm_ranks > 1 && ctx->comm_headroom > 0 && per_cu > ctx->comm_headroom + 4) per_cu_used = per_cu - ctx->comm_headroom;
        // (whole CUs reserved instead -- VX_COMM_RESERVE_CUS: the render streams' CU mask leaves them out -- : every other CU is filled)
        uint32_t waves = uint32_t(ctx->cu_count - ctx->reserve_cus) * uint32_t(per_cu_used);
        if (waves > a.total_subtiles) waves = a.total_subtiles;
        if (waves > 8192) a.timeline = nullptr;
        ctx->timeline_waves = a.timeline ? waves : 0;
        PixelList todo = {nullptr, nullptr, 0};
        if (imaged && ctx->svo_type == VX_SVO_CSVO) {
            // per stream: a ring of chunks behind a counter that only ever grows -- nothing to reset between frames. A finished
            // chunk holds at least 63 pixels, every wave can have one unfinished one: pixels / 63 + waves chunks per launch at most.
            uint32_t*& ring = slot >= 0 ? ctx->d_frame_todo[slot] : ctx->d_main_todo;
            size_t& have = slot >= 0 ? ctx->frame_todo_chunks[slot] : ctx->main_todo_chunks;
            // (kForeignRerun: chunks of 64 rays, eight times the size: every pixel can list one ray; a chunk is full before the wave starts
            // the next)
            // (image-only renders list rays -- whichever build --, renders with hit records pixels)
            const bool ray_list = !HITS;
            const size_t need = ray_list ? (size_t(p.n_local_tiles) * kTile * kTile / kRayChunkRecords + waves + 1) * (kRayChunkDwords / kChunkDwords)
                                         : size_t(p.n_local_tiles) * kTile * kTile / 63 + waves + 1;
            if (have < need) {
                if (ring) {
                    HIP_TRY(hipStreamSynchronize(stream));
                    (void)hipFree(ring);
                    ring = nullptr;
                    have = 0;
                }
                size_t cap = 64;
                while (cap < need) cap <<= 1;
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ring), (cap * kChunkDwords + 32) * sizeof(uint32_t)));  // [counter, pad][chunks...]
                HIP_TRY(hipMemsetAsync(ring, 0, 32 * sizeof(uint32_t), stream));
                have = cap;
            }
            todo.next_chunk = ring;
            todo.chunks = ring + 32;
            todo.mask = uint32_t((ray_list ? have / (kRayChunkDwords / kChunkDwords) : have) - 1);
        }
        a.batch = nullptr;
        if (batch) {
            uint8_t*& rings = ctx->d_batch[slot + 1];
            size_t& have = ctx->batch_waves[slot + 1];
            if (have < waves) {
                if (rings) {
                    HIP_TRY(hipStreamSynchronize(stream));
                    (void)hipFree(rings);
                    rings = nullptr;
                    have = 0;
                }
                const size_t all = size_t(ctx->cu_count) * size_t(per_cu);  // (the most a launch of this kernel ever has)
                const size_t n = all > waves ? all : waves;
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&rings), n * kWaveBatchBytes));
                have = n;
            }
            a.batch = rings;
        }
        void* kargs[] = {const_cast<SceneArgs*>(&sc), const_cast<RenderParams*>(&p), &a, &out, &hits, &counters, &todo};
        HIP_TRY(hipLaunchKernel(fn, dim3(waves), dim3(64), kargs, wave_lds, stream));
        tickets += 1u;
        order_after = hs;
        order_subtiles = a.total_subtiles;
    }
    HIP_TRY(hipGetLastError());
    if (ctx->profile) {
        HIP_TRY(hipEventRecord(ev.stop, stream));
        ctx->launches.push_back(ev);
    }
    if (slot >= 0) {
        HIP_TRY(hipEventRecord(ctx->frame_done[slot], stream));
        ctx->frame_recorded[slot] = true;
    } else {
        HIP_TRY(hipEventRecord(ctx->render_done, stream));
        ctx->render_recorded = true;
    }
    if (order_after) {
        // behind this frame but not in its stream's way: the table the frame after next draws its tickets through (PersistentArgs::order)
        vx_context::HotState* hs = order_after;
        const uint32_t g = hs->frames % 3;
        if (ctx->hot_sort) {
            HIP_TRY(hipStreamWaitEvent(ctx->order_stream, slot >= 0 ? ctx->frame_done[slot] : ctx->render_done, 0));
            hipLaunchKernelGGL(order_kernel, dim3(1), dim3(kOrderThreads), 0, ctx->order_stream, hs->cost[g], hs->tag, order_subtiles, hs->order[g], ctx->cost_step);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(hs->order_done[g], ctx->order_stream));
        }
        hs->frames += 1;
    }
    ctx->last_frame_slot = slot;
    return VX_OK;
}

// vx_wait_event / vx_wait_gather order the NEXT frame, whichever entry point issues it (vx_render, vx_present_begin): one-shot
int apply_pending_waits(vx_context* ctx, hipStream_t stream) {
    if (ctx->pending_wait) HIP_TRY(hipStreamWaitEvent(stream, ctx->pending_wait, 0));
    if (ctx->pending_gather) HIP_TRY(hipStreamWaitEvent(stream, ctx->pending_gather, 0));
    ctx->pending_wait = nullptr;
    ctx->pending_gather = nullptr;
    return VX_OK;
}

int fill_params(vx_context* ctx, const vx_uniforms* u, uint32_t w, uint32_t h, uint32_t tile_rank, uint32_t tile_count, int format, RenderParams& p) {
    if (!u || w == 0 || h == 0) return fail(VX_ERR_INVALID_ARGUMENT, "bad uniforms or size");
    if (uint64_t(w) * h >= (uint64_t(1) << 31)) return fail(VX_ERR_INVALID_ARGUMENT, "image too large (pixel indices are 31 bits)");
    if (format != VX_FORMAT_RGBA32F && format != VX_FORMAT_RGBA8) return fail(VX_ERR_INVALID_ARGUMENT, "unknown target format");
    if (tile_count == 0) tile_count = 1;
    if (tile_rank >= tile_count) return fail(VX_ERR_INVALID_ARGUMENT, "tile_rank >= tile_count");
    p.u = *u;
    p.tan_half_fovy = tanf(u->fovy * 0.5f);
    view_origin(u->view, p.ray_origin);
    p.affine_view = (u->view[3] == 0.0f && u->view[7] == 0.0f && u->view[11] == 0.0f && u->view[15] == 1.0f && std::isfinite(p.tan_half_fovy) &&
                     std::isfinite(u->aspect)) ? 1u : 0u;
    p.width = w;
    p.height = h;
    p.tiles_x = (w + kTile - 1) / kTile;
    p.tiles_y = (h + kTile - 1) / kTile;
    p.tile_rank = tile_rank;
    p.tile_count = tile_count;
    p.n_local_tiles = vx_local_tile_count(w, h, tile_rank, tile_count);
    p.tile_order = nullptr;
    p.rgba8 = format == VX_FORMAT_RGBA8 ? 1u : 0u;
    p.opaque_lo = uint32_t(ctx->opaque_blocks);
    p.opaque_hi = uint32_t(ctx->opaque_blocks >> 32);
    if (tile_count > 1) {
        const vx_context::TileTable* t = nullptr;
        if (int rc = tile_table(ctx, p.tiles_x, p.tiles_y, &t)) return rc;
        p.tile_order = t->d_order;
    }
    return VX_OK;
}

// one contiguous piece of a commit: `bytes` from host memory `src` to device memory `dst`
struct Upload {
    uint8_t* dst;
    const uint8_t* src;
    uint64_t bytes;
};
constexpr uint64_t kDeltaLimit = 64ull << 20;  // commits up to this size travel packed (one transfer, one scatter kernel)
constexpr uint64_t kPiece = 32768;             // bytes one workgroup of the scatter kernel moves

template <class WAIT>
int upload_packed(vx_context* ctx, const std::vector<Upload>& up, WAIT&& wait_for_frames) {
    // layout: [piece table: 3 x u64 each][payload, every upload padded so that it starts at its destination's address modulo 16]
    uint64_t pieces = 0;
    for (const Upload& u : up) pieces += (u.bytes + kPiece - 1) / kPiece;
    if (pieces == 0) return wait_for_frames();
    uint64_t at = (pieces * 24 + 15) & ~uint64_t(15);
    std::vector<uint64_t> where(up.size());
    for (size_t i = 0; i < up.size(); ++i) {
        at = ((at + 15) & ~uint64_t(15)) + (reinterpret_cast<uintptr_t>(up[i].dst) & 15u);
        where[i] = at;
        at += up[i].bytes;
    }
    const uint64_t total = at;
    vx_context::DeltaSlot& slot = ctx->delta[ctx->delta_next++ % vx_context::kDeltaSlots];
    if (slot.used) HIP_TRY(hipEventSynchronize(slot.done));  // (three commits ago: long done)
    if (slot.cap < total) {
        if (slot.host) (void)hipHostFree(slot.host);
        if (slot.dev) (void)hipFree(slot.dev);
        slot.host = slot.dev = nullptr;
        slot.cap = 0;
        const size_t cap = size_t(total + total / 2 + 4096);
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&slot.host), cap, hipHostMallocDefault));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&slot.dev), cap));
        slot.cap = cap;
    }
    if (!slot.done) HIP_TRY(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    uint64_t* table = reinterpret_cast<uint64_t*>(slot.host);
    uint64_t k = 0;
    for (size_t i = 0; i < up.size(); ++i) {
        if (!up[i].bytes) continue;
        std::memcpy(slot.host + where[i], up[i].src, up[i].bytes);
        for (uint64_t off = 0; off < up[i].bytes; off += kPiece, ++k) {
            table[k * 3] = reinterpret_cast<uintptr_t>(up[i].dst) + off;
            table[k * 3 + 1] = where[i] + off;
            table[k * 3 + 2] = std::min(kPiece, up[i].bytes - off);
        }
    }
    // the transfer needs no fence (the device twin is private to this commit): it runs while frames in flight finish
    HIP_TRY(hipMemcpyAsync(slot.dev, slot.host, total, hipMemcpyHostToDevice, ctx->upload_stream));
    if (int rc = wait_for_frames()) return rc;
    hipLaunchKernelGGL(scatter_kernel, dim3(uint32_t(pieces)), dim3(256), 0, ctx->upload_stream, reinterpret_cast<const uint64_t*>(slot.dev), slot.dev);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(slot.done, ctx->upload_stream));
    slot.used = true;
    return VX_OK;
}

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;

// RCCL is opened when the first communicator is asked for. By its soname: a process that already has one loaded (PyTorch brings
// its own copy) shares that one.
int rccl_open() {
    if (g_rccl.lib) return VX_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(VX_ERR_STATE, std::string("RCCL is not available: ") + dlerror());
#define VX_SYM(name)                                                                                  \
    g_rccl.name = reinterpret_cast<decltype(g_rccl.name)>(dlsym(h, "nccl" #name));                     \
    if (!g_rccl.name) return fail(VX_ERR_STATE, "RCCL lacks nccl" #name)
    VX_SYM(GetUniqueId); VX_SYM(CommInitRank); VX_SYM(CommDestroy); VX_SYM(GroupStart); VX_SYM(GroupEnd); VX_SYM(Send); VX_SYM(Recv); VX_SYM(GetErrorString);
#undef VX_SYM
    g_rccl.lib = h;
    return VX_OK;
}
#define NCCL_TRY(call)                                                                                                   \
    do {                                                                                                                 \
        ncclResult_t r_ = (call);                                                                                        \
        if (r_ != ncclSuccess) return fail(VX_ERR_HIP, std::string(#call) + ": " + g_rccl.GetErrorString(r_));         \
    } while (0)

}  // namespace

extern "C" {

const char* vx_last_error(void) { return g_last_error.c_str(); }
const char* vx_version(void) { return "voxel-hip 0.1 (gfx950)"; }

uint32_t vx_local_tile_count(uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_count) {
    if (tile_count == 0) tile_count = 1;
    const uint32_t total = ((width + kTile - 1) / kTile) * ((height + kTile - 1) / kTile);
    if (tile_rank >= tile_count) return 0;
    return (total - tile_rank + tile_count - 1) / tile_count;
}

int vx_create(int svo_type, size_t capacity_bytes, int device, vx_context** out) {
    if (!out) return fail(VX_ERR_INVALID_ARGUMENT, "out is null");
    *out = nullptr;
    if (svo_type != VX_SVO_ESVO && svo_type != VX_SVO_CSVO) return fail(VX_ERR_INVALID_ARGUMENT, "svo_type must be VX_SVO_ESVO or VX_SVO_CSVO");
    if (capacity_bytes < 64) return fail(VX_ERR_INVALID_ARGUMENT, "capacity_bytes too small");
    // CSVO pointers are 31-bit byte offsets (bit 31 flags an absolute one, csvo.rs:100-105), ESVO pointers 32-bit indices of
    // 4-byte words (esvo.rs:74-101): nothing beyond 4 GiB / 16 GiB could be referenced
    if (svo_type == VX_SVO_CSVO && capacity_bytes >= (size_t(1) << 32) - 64)
        return fail(VX_ERR_CAPACITY, "a CSVO world buffer cannot exceed 4 GiB: its pointers are byte offsets of at most 31 bits");
    if (svo_type == VX_SVO_ESVO && capacity_bytes > (size_t(1) << 34))
        return fail(VX_ERR_CAPACITY, "an ESVO world buffer cannot exceed 16 GiB: its pointers are 32-bit indices of 4-byte words");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(VX_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(VX_ERR_NO_DEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    vx_context* c = new (std::nothrow) vx_context();
    if (!c) return fail(VX_ERR_OUT_OF_MEMORY, "context allocation failed");
    c->svo_type = svo_type;
    c->device = device;
    c->capacity = (capacity_bytes + 15) & ~size_t(15);
    c->stats.capacity_bytes = capacity_bytes;
    c->big = c->capacity + kWorldPad >= (size_t(1) << 32);  // beyond a buffer resource's 32-bit offsets
    auto cleanup = [&](int code) {
        vx_destroy(c);
        return code;
    };
#define CREATE_TRY(call)                                                                                                   \
    do {                                                                                                                   \
        hipError_t e_ = (call);                                                                                            \
        if (e_ != hipSuccess) {                                                                                            \
            g_last_error = std::string(#call) + ": " + hipGetErrorString(e_);                                              \
            return cleanup(e_ == hipErrorOutOfMemory ? VX_ERR_OUT_OF_MEMORY : VX_ERR_HIP);                                 \
        }                                                                                                                  \
    } while (0)
    // (kStagingSlack bytes more than the caller may use: the reference's own call, write_changes_to(ptr + 4, len - 1, ..), checks its
    // ranges against a length that ignores the writer's header (svo.rs:180-181, esvo.rs:328) and can run that far past the end)
    CREATE_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->staging), c->capacity + kStagingSlack, hipHostMallocDefault));
    std::memset(c->staging, 0, c->capacity + kStagingSlack);
    // kWorldPad zero bytes follow the buffer and are inside the descriptor's range: an unaligned dword read that straddles
    // the end then returns the real bytes plus zeros (what the word-wise reference reads), not an all-zero dword
    CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_world), c->capacity + kWorldPad));
    CREATE_TRY(hipMemset(c->d_world, 0, c->capacity + kWorldPad));
    std::vector<uint32_t> cu_mask;  // (empty: no reservation)
    {
        hipDeviceProp_t prop;
        CREATE_TRY(hipGetDeviceProperties(&prop, device));
        c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (const char* e = std::getenv("VX_COMM_RESERVE_CUS")) c->reserve_cus = std::max(0, std::min(std::atoi(e), c->cu_count / 2));
        if (c->reserve_cus > 0) {
            // the mask's LAST n bits stay clear (which CUs those are is the driver's numbering: a handful of whole CUs is all that matters)
            cu_mask.assign(size_t(c->cu_count + 31) / 32, 0u);
            for (int b = 0; b < c->cu_count - c->reserve_cus; ++b) cu_mask[size_t(b) >> 5] |= 1u << (b & 31);
        }
    }
    if (!cu_mask.empty()) CREATE_TRY(hipExtStreamCreateWithCUMask(&c->stream, uint32_t(cu_mask.size()), cu_mask.data()));
    else CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CREATE_TRY(hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreateWithFlags(&c->upload_done, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&c->render_done, hipEventDisableTiming));
    // Frame streams must sit on different hardware queues or their kernels serialise. The runtime hands out queues from a
    // small pool per priority level (GPU_MAX_HW_QUEUES, 4 by default), least used first, with no way to ask which one a
    // stream got; the default-priority pool is already shared with this context's other streams and the caller's. The frame
    // streams therefore alternate between the lowest and the highest priority level, whose pools nothing else uses.
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    for (int i = 0; i < vx_context::kFrameStreams; ++i) {
        const int prio = (i & 1) ? prio_greatest : prio_least;
        // (a stream with a CU mask has a hardware queue of its own)
        if (!cu_mask.empty()) CREATE_TRY(hipExtStreamCreateWithCUMask(&c->frame_stream[i], uint32_t(cu_mask.size()), cu_mask.data()));
        else CREATE_TRY(hipStreamCreateWithPriority(&c->frame_stream[i], hipStreamNonBlocking, prio));
        CREATE_TRY(hipEventCreateWithFlags(&c->frame_done[i], hipEventDisableTiming));
        CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_frame_counter[i]), 2 * kQueues * kQueueStride * sizeof(uint32_t)));
        CREATE_TRY(hipMemset(c->d_frame_counter[i], 0, 2 * kQueues * kQueueStride * sizeof(uint32_t)));
    }
    CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_counters), 16 * sizeof(unsigned long long)));
    CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_work_counter), 2 * kQueues * kQueueStride * sizeof(uint32_t)));
    CREATE_TRY(hipMemset(c->d_work_counter, 0, 2 * kQueues * kQueueStride * sizeof(uint32_t)));
    CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_excursions), 8 * sizeof(unsigned long long)));
    CREATE_TRY(hipMemset(c->d_excursions, 0, 8 * sizeof(unsigned long long)));
    {
        if (const char* e = std::getenv("VX_RENDER_KERNEL")) c->kernel_version = std::atoi(e) == 1 ? 1 : 2;
        if (const char* e = std::getenv("VX_MIN_WAVES")) c->min_waves = std::atoi(e);
        if (const char* e = std::getenv("VX_FRAMES_IN_FLIGHT")) c->frames_in_flight = std::atoi(e);
        if (c->frames_in_flight < 1) c->frames_in_flight = 1;
        if (c->frames_in_flight > vx_context::kFrameStreams) c->frames_in_flight = vx_context::kFrameStreams;
        if (const char* e = std::getenv("VX_TRAVERSAL_IMAGE")) c->image_enabled = std::atoi(e) != 0;
        if (const char* e = std::getenv("VX_NO_EXCURSION")) c->no_excursion = std::atoi(e) != 0;
        if (const char* e = std::getenv("VX_DEEP_STACK")) c->deep_stack = std::atoi(e) != 0;
        if (const char* e = std::getenv("VX_FIVE_WAVES")) c->five_waves = std::atoi(e) != 0;
        if (const char* e = std::getenv("VX_SORT_PERIOD")) { uint32_t n = uint32_t(std::max(1, std::atoi(e))), m = 1; while (m * 2 <= n && m < 1024) m *= 2; c->sort_mask = m - 1; }
        if (std::getenv("VX_SORT_EVERY_FRAME")) c->sort_mask = 0;
        if (const char* e = std::getenv("VX_SORTED")) { c->sorted_passes = std::atoi(e) != 0; c->sorted_always = std::atoi(e) == 2; }
        if (const char* e = std::getenv("VX_COST_FLOOR")) c->cost_floor = uint32_t(std::atoi(e));
        if (const char* e = std::getenv("VX_COST_STEP")) c->cost_step = std::atoi(e) > 0 ? uint32_t(std::atoi(e)) : kCostStep;
        if (const char* e = std::getenv("VX_DEEP_WAVES")) c->deep_waves = std::atoi(e) == 3 ? 3 : 4;
        if (const char* e = std::getenv("VX_BATCH")) c->batch_service = std::atoi(e) != 0;
        if (const char* e = std::getenv("VX_FOREIGN_RERUN")) c->foreign_rerun = std::atoi(e) != 0 ? 1 : 0;
        if (const char* e = std::getenv("VX_HOT_LEVELS")) c->hot_levels = std::atoi(e) != 0;
        if (const char* e = std::getenv("VX_HOT_FIRST")) {
            const int v = std::atoi(e);
            c->hot_first = v != 0;
            c->hot_use = (v & 1) != 0; c->hot_note = (v & 2) != 0; c->hot_sort = (v & 4) != 0;
        }
        if (const char* e = std::getenv("VX_TIMELINE"))
            if (std::atoi(e) != 0) CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_timeline), 8192 * 8 * sizeof(unsigned long long)));
        if (const char* e = std::getenv("VX_TIMELINE_PART")) c->timeline_part = uint32_t(std::atoi(e));
        if (const char* e = std::getenv("VX_TICKET_AHEAD")) c->ticket_ahead = std::atoi(e) != 0 ? 1 : 0;
        if (const char* e = std::getenv("VX_AHEAD_GUARD")) c->ahead_guard = uint32_t(std::max(0, std::min(64, std::atoi(e))));
        if (const char* e = std::getenv("VX_IMAGE_CAP_BYTES")) c->image_cap_bytes = size_t(std::strtoull(e, nullptr, 10));
        // VX_WIDE_IMAGE=1: the layout for images beyond 4 GiB from the start; 2: and its arena starts 5 GiB into the frame, so that
        // every pointer needs more than 32 bits of byte offset (tests)
        int wide_image = 0;
        if (const char* e = std::getenv("VX_WIDE_IMAGE")) wide_image = std::atoi(e);
        c->image = vximg::WorldImage(svo_type, wide_image ? vximg::kOct64Wide : vximg::kOct64, wide_image == 2 ? (uint64_t(5) << 30) / 4 : 0);
        if (const char* e = std::getenv("VX_WAVES_PER_CU")) c->waves_per_cu_cap = std::atoi(e);
        if (const char* e = std::getenv("VX_COMM_HEADROOM")) c->comm_headroom = std::max(0, std::atoi(e));
        if (const char* e = std::getenv("VX_REFILL_MIN")) c->refill_min = uint32_t(std::atoi(e));
        if (const char* e = std::getenv("VX_SERVICE_MIN")) c->service_min = uint32_t(std::atoi(e));
        if (const char* e = std::getenv("VX_FOREIGN_MIN")) c->foreign_min = uint32_t(std::max(1, std::min(64, std::atoi(e))));
        if (c->refill_min < 1) c->refill_min = 1;
        if (c->refill_min > 64) c->refill_min = 64;
        if (c->service_min < 1) c->service_min = 1;
        if (c->service_min > 64) c->service_min = 64;
    }
    // one all-zero material and a 1x1 transparent-black texture so that rendering works before any registry is set
    const vx_material zero_mat = {0, 0, -1, -1, -1, -1, -1, -1};
    CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_materials), sizeof zero_mat));
    CREATE_TRY(hipMemcpy(c->d_materials, &zero_mat, sizeof zero_mat, hipMemcpyHostToDevice));
    c->n_materials = 1;
#undef CREATE_TRY
    *out = c;
    return VX_OK;
}

void vx_destroy(vx_context* c) {
    if (!c) return;
    stop_commit_worker(c);
    (void)hipSetDevice(c->device);
    (void)drain_streams(c);  // nothing may still be reading what is freed below
    for (auto& l : c->launches) { (void)hipEventDestroy(l.start); (void)hipEventDestroy(l.stop); }
    for (auto& l : c->event_pool) { (void)hipEventDestroy(l.start); (void)hipEventDestroy(l.stop); }
    if (c->staging) (void)hipHostFree(c->staging);
    void* dev[] = {c->d_world, c->d_materials, c->d_tex, c->d_frame, c->d_hits, c->d_tasks, c->d_results, c->d_trace_result, c->d_trace_frames,
                   c->d_trace_count, c->d_counters, c->d_work_counter, c->d_image, c->d_origin, c->d_excursions, c->d_main_todo, c->d_timeline};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    for (int i = 0; i <= vx_context::kFrameStreams; ++i)
        if (c->d_batch[i]) (void)hipFree(c->d_batch[i]);
    for (int i = 0; i < vx_context::kFrameStreams; ++i) {
        if (c->d_frame_counter[i]) (void)hipFree(c->d_frame_counter[i]);
        if (c->d_frame_todo[i]) (void)hipFree(c->d_frame_todo[i]);
        if (c->frame_done[i]) (void)hipEventDestroy(c->frame_done[i]);
        if (c->frame_stream[i]) (void)hipStreamDestroy(c->frame_stream[i]);
    }
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    for (auto& hs : c->hot) {
        for (int g = 0; g < 3; ++g) {
            if (hs.cost[g]) (void)hipFree(hs.cost[g]);
            if (hs.order[g]) (void)hipFree(hs.order[g]);
            if (hs.order_done[g]) (void)hipEventDestroy(hs.order_done[g]);
        }
    }
    for (auto& ss : c->sorted_state) {
        void* tables[] = {ss.rec[0], ss.rec[1], ss.perm[0], ss.perm[1]};
        for (void* q : tables)
            if (q) (void)hipFree(q);
    }
    for (auto& e : c->gather_done)
        if (e) (void)hipEventDestroy(e);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    if (c->order_stream) (void)hipStreamDestroy(c->order_stream);
    for (auto& ps : c->present) {
        if (ps.dev) (void)hipFree(ps.dev);
        if (ps.host) (void)hipHostFree(ps.host);
        if (ps.copied) (void)hipEventDestroy(ps.copied);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (auto& t : c->tile_tables) {
        if (t.d_order) (void)hipFree(t.d_order);
        if (t.d_inverse) (void)hipFree(t.d_inverse);
    }
    for (auto& d : c->delta) {
        if (d.host) (void)hipHostFree(d.host);
        if (d.dev) (void)hipFree(d.dev);
        if (d.done) (void)hipEventDestroy(d.done);
    }
    if (c->upload_done) (void)hipEventDestroy(c->upload_done);
    if (c->render_done) (void)hipEventDestroy(c->render_done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->upload_stream) (void)hipStreamDestroy(c->upload_stream);
    delete c;
}

}  // extern "C"

namespace {
// Which block ids (< 64) are opaque throughout: all three face textures of the block's material row -- the layers texture_lod() picks for
// them (round to nearest, clamped to the array) -- have alpha > 0 in every texel of every mip level. Then every sample's alpha is > 0
// too (NEAREST: a texel; LINEAR_MIPMAP_LINEAR: a blend of texels with non-negative weights that sum to 1, of which one is >= 1/4 on
// each level), i.e. a voxel of the block is a hit for any ray that reaches it (svo.esvo.glsl:241: `tex_color.a > 0`).
void update_opaque_blocks(vx_context* ctx) {
    uint64_t set = 0;
    const uint32_t layers = uint32_t(ctx->opaque_layer.size());
    if (layers != 0 && ctx->tex.levels != 0) {
        auto layer_of = [&](int32_t id) -> uint32_t {  // texture_lod(): floor(float(id) + 0.5), clamped
            const float lf = std::floor(float(id) + 0.5f);
            return lf <= 0.0f ? 0u : (lf >= float(layers - 1) ? layers - 1 : uint32_t(lf));
        };
        for (size_t v = 0; v < ctx->host_materials.size() && v < 64; ++v) {
            const vx_material& m = ctx->host_materials[v];
            if (ctx->opaque_layer[layer_of(m.tex_top)] && ctx->opaque_layer[layer_of(m.tex_side)] && ctx->opaque_layer[layer_of(m.tex_bottom)]) set |= uint64_t(1) << v;
        }
    }
    ctx->opaque_blocks = std::getenv("VX_NO_OPAQUE_SET") ? 0 : set;  // (measurement: every leaf test samples its texture, as before round 3)
}
}  // namespace

extern "C" {

int vx_set_materials(vx_context* ctx, const vx_material* rows, uint32_t count) {
    if (!ctx || !rows || count == 0) return fail(VX_ERR_INVALID_ARGUMENT, "materials: null or empty");
    VX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    // the new table is complete before anything is swapped; the old one is freed once every frame in flight (they hold its address
    // in their kernel arguments, on any of the frame streams) has finished
    vx_material* fresh = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&fresh), size_t(count) * sizeof(vx_material)));
    if (hipMemcpy(fresh, rows, size_t(count) * sizeof(vx_material), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(fresh);
        return fail(VX_ERR_HIP, "materials: upload failed");
    }
    if (int rc = drain_streams(ctx)) {
        (void)hipFree(fresh);
        return rc;
    }
    if (ctx->d_materials) (void)hipFree(ctx->d_materials);
    ctx->d_materials = fresh;
    ctx->n_materials = count;
    ctx->host_materials.assign(rows, rows + count);
    update_opaque_blocks(ctx);
    return VX_OK;
}

int vx_set_textures(vx_context* ctx, const uint8_t* rgba8, uint32_t width, uint32_t height, uint32_t layers, uint32_t mip_levels) {
    if (!ctx || !rgba8 || !width || !height || !layers) return fail(VX_ERR_INVALID_ARGUMENT, "textures: null or empty");
    VX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    // mip_levels = min(requested, ilog2(min(w, h))), never below 1 (texture_array.rs:108, :193)
    uint32_t m = width < height ? width : height, lg = 0;
    while (m >>= 1) ++lg;
    uint32_t levels = mip_levels < lg ? mip_levels : lg;
    if (levels < 1) levels = 1;
    if (levels > 16) levels = 16;

    decltype(ctx->tex) t = {};
    t.width = width; t.height = height; t.layers = layers; t.levels = levels;
    size_t total = 0;
    for (uint32_t l = 0; l < levels; ++l) {
        const uint32_t w = (width >> l) ? (width >> l) : 1, h = (height >> l) ? (height >> l) : 1;
        t.level_offset[l] = uint32_t(total);
        total += size_t(layers) * w * h * 4;
    }
    std::vector<uint8_t> chain(total);
    std::memcpy(chain.data(), rgba8, size_t(layers) * width * height * 4);
    // glGenerateMipmap (texture_array.rs:259): 2x2 box filter, each level from the previous one
    for (uint32_t l = 1; l < levels; ++l) {
        const uint32_t sw = (width >> (l - 1)) ? (width >> (l - 1)) : 1, sh = (height >> (l - 1)) ? (height >> (l - 1)) : 1;
        const uint32_t dw = (width >> l) ? (width >> l) : 1, dh = (height >> l) ? (height >> l) : 1;
        const uint8_t* src = chain.data() + t.level_offset[l - 1];
        uint8_t* dst = chain.data() + t.level_offset[l];
        for (uint32_t layer = 0; layer < layers; ++layer)
            for (uint32_t y = 0; y < dh; ++y)
                for (uint32_t x = 0; x < dw; ++x) {
                    const uint32_t xa = 2 * x, xb = xa + 1 < sw ? xa + 1 : sw - 1, ya = 2 * y, yb = ya + 1 < sh ? ya + 1 : sh - 1;
                    const uint8_t* s = src + size_t(layer) * sw * sh * 4;
                    for (uint32_t ch = 0; ch < 4; ++ch) {
                        const uint32_t sum = s[(size_t(ya) * sw + xa) * 4 + ch] + s[(size_t(ya) * sw + xb) * 4 + ch] + s[(size_t(yb) * sw + xa) * 4 + ch] +
                                             s[(size_t(yb) * sw + xb) * 4 + ch];
                        dst[((size_t(layer) * dh + y) * dw + x) * 4 + ch] = uint8_t((sum + 2) / 4);
                    }
                }
    }
    if (total >= (size_t(1) << 32)) return fail(VX_ERR_CAPACITY, "textures: the mip chain exceeds 4 GiB");
    uint8_t* fresh = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&fresh), total));
    if (hipMemcpy(fresh, chain.data(), total, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(fresh);
        return fail(VX_ERR_HIP, "textures: upload failed");
    }
    if (int rc = drain_streams(ctx)) {  // frames in flight sample the old chain (see vx_set_materials)
        (void)hipFree(fresh);
        return rc;
    }
    if (ctx->d_tex) (void)hipFree(ctx->d_tex);
    ctx->d_tex = fresh;
    ctx->tex_bytes = uint32_t(total);
    ctx->tex = t;
    ctx->opaque_layer.assign(layers, 1);
    for (uint32_t l = 0; l < levels; ++l) {
        const uint32_t w = (width >> l) ? (width >> l) : 1, h = (height >> l) ? (height >> l) : 1;
        for (uint32_t layer = 0; layer < layers; ++layer) {
            const uint8_t* texels = chain.data() + t.level_offset[l] + size_t(layer) * w * h * 4;
            for (size_t i = 0; i < size_t(w) * h && ctx->opaque_layer[layer]; ++i)
                if (texels[i * 4 + 3] == 0) ctx->opaque_layer[layer] = 0;
        }
    }
    update_opaque_blocks(ctx);
    return VX_OK;
}

uint8_t* vx_staging_ptr(vx_context* ctx) {
    if (!ctx) return nullptr;
    // a pipelined commit reads the mirror on the worker thread: whoever asks for the pointer (to write the next changes) waits for it
    if (ctx->worker.joinable()) wait_commit_idle(ctx);
    return ctx->staging;
}
size_t vx_capacity(const vx_context* ctx) { return ctx ? size_t(ctx->stats.capacity_bytes) : 0; }
size_t vx_arena_capacity(const vx_context* ctx) { return ctx ? size_t(ctx->stats.capacity_bytes) - 4 - header_bytes(ctx) : 0; }

}  // extern "C"

namespace {

// The commit proper (arguments already checked): image update and packing on the calling thread -- the caller's, or the
// context's commit worker --, then, under the context's mutex, everything that touches the device or what renders read.
int commit_now(vx_context* ctx, uint32_t depth, const vx_range* ranges, uint32_t count, uint64_t used_bytes) {
    HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t head = 4 + header_bytes(ctx);

    // octree_scale = 2^-depth as f32 at byte 0 (svo.rs:173-175)
    const float scale = std::exp2(-float(depth));
    std::memcpy(ctx->staging, &scale, 4);

    // What goes to the device: the writer's header, the dirty arena ranges (neighbours closer than 4 KiB travel as one: the
    // staging mirror holds the whole world, so the bytes between them are the device's own), and the parts of the traversal image
    // and its origin table that the image update below rewrites.
    std::vector<Upload> up;
    up.push_back(Upload{ctx->d_world, ctx->staging, head});
    {
        std::vector<vx_range> r(ranges, ranges + count);
        std::sort(r.begin(), r.end(), [](const vx_range& x, const vx_range& y) { return x.start < y.start; });
        for (size_t i = 0; i < r.size();) {
            uint64_t lo = r[i].start, hi = r[i].start + r[i].length;
            size_t j = i + 1;
            while (j < r.size() && r[j].start <= hi + 4096) {
                hi = std::max(hi, r[j].start + r[j].length);
                ++j;
            }
            if (hi > lo) up.push_back(Upload{ctx->d_world + head + lo, ctx->staging + head + lo, hi - lo});
            i = j;
        }
    }

    bool image_ok = false;
    if (ctx->image_enabled && ctx->kernel_version != 1) {
        // re-lay the changed chunks (and the root octree, which every commit rewrites) out as octants
        std::vector<vximg::Range> changed(count);
        for (uint32_t i = 0; i < count; ++i) changed[i] = vximg::Range{ranges[i].start, ranges[i].length};
        const unsigned threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        // A world whose image will not fit 32-bit byte offsets starts in the wide layout instead of finding that out at the end of a whole
        // build (an image is about 0.84 x the bytes of an ESVO world, 3.9 x those of a CSVO world; the wide layout serves any size)
        if (ctx->image.chunk_count() == 0 && ctx->image.layout() == vximg::kOct64 &&
            double(used_bytes) * (ctx->svo_type == VX_SVO_ESVO ? 0.95 : 4.4) >= 3.5 * double(1ull << 30))
            ctx->image = vximg::WorldImage(ctx->svo_type, vximg::kOct64Wide);
        image_ok = ctx->image.update(ctx->staging, used_bytes, changed.data(), changed.size(), threads);
        if (!image_ok && ctx->image.too_big() && ctx->image.layout() == vximg::kOct64) {
            // past what 32-bit byte offsets reach: from here on octant indices (the image is rebuilt once, whole)
            ctx->image = vximg::WorldImage(ctx->svo_type, vximg::kOct64Wide);
            image_ok = ctx->image.update(ctx->staging, used_bytes, nullptr, 0, threads);
        }
    }
    VX_LOCK(ctx);  // from here on: device memory, streams, events and the state renders read
    // The image is an accelerator: whatever goes wrong with it (a world that cannot be imaged, no device memory for it), the
    // context falls back to traversing the world's own bytes -- with nothing of a half-made image left behind.
    auto drop_image = [&]() {
        image_ok = false;
        if (ctx->d_image || ctx->d_origin) (void)drain_streams(ctx);  // frames in flight still walk it
        if (ctx->d_image) (void)hipFree(ctx->d_image);
        if (ctx->d_origin) (void)hipFree(ctx->d_origin);
        ctx->d_image = ctx->d_origin = nullptr;
        ctx->d_image_capacity = ctx->d_origin_capacity = 0;
        const vximg::Layout layout = ctx->image.layout();
        ctx->image = vximg::WorldImage(ctx->svo_type, layout);  // the next commit rebuilds it whole
    };
    bool whole_image = false;
    if (image_ok) {
        const size_t need = ctx->image.frame_bytes() + kImagePad;
        const size_t need_origin = ctx->image.has_origin() ? ctx->image.origin_bytes() + kImagePad : 0;
        if (need > ctx->d_image_capacity || need_origin > ctx->d_origin_capacity) {
            // grow both (frames in flight still read the old ones: wait for them), then everything is uploaded again
            (void)drain_streams(ctx);
            if (ctx->d_image) (void)hipFree(ctx->d_image);
            if (ctx->d_origin) (void)hipFree(ctx->d_origin);
            ctx->d_image = ctx->d_origin = nullptr;
            ctx->d_image_capacity = ctx->d_origin_capacity = 0;
            const size_t cap = std::min(need + need / 2 + (1 << 20), ctx->image_cap_bytes ? ctx->image_cap_bytes : ~size_t(0));
            const size_t cap_origin = need_origin ? cap / 4 + kImagePad : 0;
            bool ok = cap >= need && hipMalloc(reinterpret_cast<void**>(&ctx->d_image), cap) == hipSuccess;
            if (ok && cap_origin) ok = hipMalloc(reinterpret_cast<void**>(&ctx->d_origin), cap_origin) == hipSuccess;
            if (ok) ok = hipMemsetAsync(ctx->d_image, 0, cap, ctx->upload_stream) == hipSuccess;
            if (ok && cap_origin) ok = hipMemsetAsync(ctx->d_origin, 0, cap_origin, ctx->upload_stream) == hipSuccess;
            if (ok) {
                ctx->d_image_capacity = cap;
                ctx->d_origin_capacity = cap_origin;
                whole_image = true;
            } else {
                (void)hipGetLastError();  // (an allocation failure is not the caller's error: the bytes path serves)
                drop_image();
            }
        }
    } else if (ctx->image_enabled && ctx->kernel_version != 1) {
        drop_image();
    }
    if (image_ok) {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(ctx->image.frame().data());
        const uint8_t* osrc = reinterpret_cast<const uint8_t*>(ctx->image.origin().data());
        if (whole_image) {
            up.push_back(Upload{ctx->d_image, src, ctx->image.frame_bytes()});
            if (ctx->image.has_origin()) up.push_back(Upload{ctx->d_origin, osrc, ctx->image.origin_bytes()});
        } else {
            for (const vximg::Range& r : ctx->image.dirty_bytes()) {
                up.push_back(Upload{ctx->d_image + r.start, src + r.start, r.length});
                // two origin words per 32-byte unit of the frame
                if (ctx->image.has_origin()) up.push_back(Upload{ctx->d_origin + r.start / 4, osrc + r.start / 4, (r.length + 3) / 4});
            }
        }
    }

    // render_fence.wait() (svo.rs:178), on the device: nothing of this commit may land while a frame in flight is still traversing
    // the nodes it replaces
    auto wait_for_frames = [&]() -> int {
        if (ctx->render_recorded) HIP_TRY(hipStreamWaitEvent(ctx->upload_stream, ctx->render_done, 0));
        for (int i = 0; i < vx_context::kFrameStreams; ++i)
            if (ctx->frame_recorded[i]) HIP_TRY(hipStreamWaitEvent(ctx->upload_stream, ctx->frame_done[i], 0));
        return VX_OK;
    };
    uint64_t total = 0;
    for (const Upload& u : up) total += u.bytes;
    int rc = VX_OK;
    if (total <= kDeltaLimit) {
        // many small pieces: ONE packed transfer + one scatter kernel, and the caller's staging mirror is free as soon as this
        // function returns (no wait for the device)
        rc = upload_packed(ctx, up, wait_for_frames);
    } else {
        rc = wait_for_frames();
        for (size_t i = 0; i < up.size() && rc == VX_OK; ++i)
            if (up[i].bytes && hipMemcpyAsync(up[i].dst, up[i].src, up[i].bytes, hipMemcpyHostToDevice, ctx->upload_stream) != hipSuccess)
                rc = fail(VX_ERR_HIP, std::string("commit: upload failed: ") + hipGetErrorString(hipGetLastError()));
        // the caller may rewrite the staging mirror as soon as we return: wait for the copies to have read it
        if (rc == VX_OK && hipStreamSynchronize(ctx->upload_stream) != hipSuccess) rc = fail(VX_ERR_HIP, "commit: upload failed");
    }
    if (rc != VX_OK) {
        // the device copy of the image can no longer be trusted; the world's own bytes may be incomplete too, which the caller
        // learns from the error -- a later commit of the same ranges repairs both
        const std::string why = g_last_error;
        drop_image();
        ctx->image_ok = false;
        g_last_error = why;
        return rc;
    }
    ctx->image_ok = image_ok;
    if (image_ok) {
        ctx->pub.frame_bytes = ctx->image.frame_bytes();
        ctx->pub.origin_bytes = ctx->image.has_origin() ? ctx->image.origin_bytes() : 0u;
        ctx->pub.chunks = ctx->image.chunk_count();
        ctx->pub.depth = ctx->image.depth();
        ctx->pub.layout = ctx->image.layout();
    }
    HIP_TRY(hipEventRecord(ctx->upload_done, ctx->upload_stream));
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->upload_done, 0));
    for (int i = 0; i < vx_context::kFrameStreams; ++i) HIP_TRY(hipStreamWaitEvent(ctx->frame_stream[i], ctx->upload_done, 0));

    ctx->stats.depth = depth;
    ctx->stats.used_bytes = used_bytes;
    ctx->committed = true;
    return VX_OK;
}

// ---- pipelined commits: the context's worker thread ------------------------------------------------------------------------

void wait_commit_idle(vx_context* ctx) {
    std::unique_lock<std::mutex> lk(ctx->job_mutex);
    ctx->job_cv.wait(lk, [&] { return !ctx->job_posted && !ctx->job_running; });
}

// the error of the last pipelined commit, once (VX_OK if there was none)
int take_async_error(vx_context* ctx) {
    std::unique_lock<std::mutex> lk(ctx->job_mutex);
    const int rc = ctx->async_rc;
    if (rc == VX_OK) return VX_OK;
    ctx->async_rc = VX_OK;
    return fail(rc, "pipelined commit failed: " + ctx->async_error);
}

void commit_worker(vx_context* ctx) {
    (void)hipSetDevice(ctx->device);
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(ctx->job_mutex);
            ctx->job_cv.wait(lk, [&] { return ctx->job_posted || ctx->worker_stop; });
            if (!ctx->job_posted) return;
            ctx->job_posted = false;
            ctx->job_running = true;
        }
        const vx_context::CommitJob& j = ctx->job;  // (not rewritten before job_running is false again)
        const int rc = commit_now(ctx, j.depth, j.ranges.data(), uint32_t(j.ranges.size()), j.used_bytes);
        {
            std::unique_lock<std::mutex> lk(ctx->job_mutex);
            if (rc != VX_OK && ctx->async_rc == VX_OK) {
                ctx->async_rc = rc;
                ctx->async_error = g_last_error;
            }
            ctx->job_running = false;
        }
        ctx->job_cv.notify_all();
    }
}

void stop_commit_worker(vx_context* ctx) {
    if (!ctx->worker.joinable()) return;
    wait_commit_idle(ctx);
    {
        std::unique_lock<std::mutex> lk(ctx->job_mutex);
        ctx->worker_stop = true;
    }
    ctx->job_cv.notify_all();
    ctx->worker.join();
    ctx->worker_stop = false;
}

}  // namespace

extern "C" {

int vx_commit(vx_context* ctx, uint32_t depth, const vx_range* ranges, uint32_t count, uint64_t used_bytes) {
    if (!ctx || (count && !ranges)) return fail(VX_ERR_INVALID_ARGUMENT, "commit: null argument");
    if (depth > uint32_t(kMaxScale)) return fail(VX_ERR_INVALID_ARGUMENT, "depth exceeds the traversal's 23-level limit (svo.esvo.glsl:21)");
    const uint64_t head = 4 + header_bytes(ctx);
    const uint64_t arena = ctx->stats.capacity_bytes - head;
    if (used_bytes > arena) return fail(VX_ERR_CAPACITY, "dst is not large enough: used_bytes exceeds the world buffer");
    for (uint32_t i = 0; i < count; ++i)
        if (ranges[i].start + ranges[i].length > arena || ranges[i].start + ranges[i].length < ranges[i].start)
            return fail(VX_ERR_CAPACITY, "dst is not large enough: a dirty range exceeds the world buffer");
    if (ctx->commit_mode == VX_COMMIT_PIPELINED && ctx->committed) {
        // (the first commit of a context is always done here and now: nothing can be rendered before it)
        wait_commit_idle(ctx);
        if (int rc = take_async_error(ctx)) return rc;
        {
            std::unique_lock<std::mutex> lk(ctx->job_mutex);
            ctx->job.depth = depth;
            ctx->job.ranges.assign(ranges, ranges + count);
            ctx->job.used_bytes = used_bytes;
            ctx->job_posted = true;
        }
        ctx->job_cv.notify_all();
        return VX_OK;
    }
    wait_commit_idle(ctx);
    return commit_now(ctx, depth, ranges, count, used_bytes);
}

int vx_set_commit_mode(vx_context* ctx, int mode) {
    if (!ctx || (mode != VX_COMMIT_INLINE && mode != VX_COMMIT_PIPELINED)) return fail(VX_ERR_INVALID_ARGUMENT, "commit mode: VX_COMMIT_INLINE or VX_COMMIT_PIPELINED");
    wait_commit_idle(ctx);
    if (mode == VX_COMMIT_PIPELINED && !ctx->worker.joinable()) ctx->worker = std::thread(commit_worker, ctx);
    if (mode == VX_COMMIT_INLINE) stop_commit_worker(ctx);
    ctx->commit_mode = mode;
    return VX_OK;
}

int vx_commit_wait(vx_context* ctx) {
    if (!ctx) return fail(VX_ERR_INVALID_ARGUMENT, "null context");
    wait_commit_idle(ctx);
    return take_async_error(ctx);
}

int vx_commit_all(vx_context* ctx, uint32_t depth, uint64_t used_bytes) {
    const vx_range all = {0, used_bytes};
    return vx_commit(ctx, depth, &all, 1, used_bytes);
}

int vx_get_stats(const vx_context* ctx, vx_stats* out) {
    if (!ctx || !out) return fail(VX_ERR_INVALID_ARGUMENT, "stats: null argument");
    VX_LOCK(const_cast<vx_context*>(ctx));
    *out = ctx->stats;
    return VX_OK;
}

int vx_render(vx_context* ctx, const vx_uniforms* uniforms, uint32_t width, uint32_t height, const vx_target* target) {
    if (int rc = check_ready(ctx)) return rc;
    VX_LOCK(ctx);
    if (!target || !target->rgba32f) return fail(VX_ERR_INVALID_ARGUMENT, "render: null target");
    RenderParams p;
    if (int rc = fill_params(ctx, uniforms, width, height, target->tile_rank, target->tile_count, target->format, p)) return rc;
    if (p.rgba8 && ctx->kernel_version == 1) return fail(VX_ERR_INVALID_ARGUMENT, "the one-thread-per-pixel kernel (VX_RENDER_KERNEL=1) writes RGBA32F only");
    const size_t pixels = p.tile_count > 1 ? size_t(p.n_local_tiles) * kTile * kTile : size_t(width) * height;
    const size_t pixel_bytes = p.rgba8 ? 4 : 16;

    float* out = static_cast<float*>(target->rgba32f);
    vx_hit* hits = target->hits;
    if (target->memory == VX_MEM_HOST) {
        if (int rc = ensure(reinterpret_cast<void**>(&ctx->d_frame), &ctx->d_frame_bytes, pixels * pixel_bytes)) return rc;
        out = ctx->d_frame;
        if (hits) {
            if (int rc = ensure(reinterpret_cast<void**>(&ctx->d_hits), &ctx->d_hits_bytes, pixels * sizeof(vx_hit))) return rc;
            hits = ctx->d_hits;
        }
    }
    int slot = -1;
    if (!hits && target->memory == VX_MEM_DEVICE && ctx->frames_in_flight > 1 && ctx->kernel_version != 1) {
        slot = int(ctx->frame_index++ % unsigned(ctx->frames_in_flight));
        // ordered after whatever the caller put on `stream` before the PREVIOUS frame on this slot was issued is implied by
        // stream order; explicit cross-stream dependencies come in through vx_wait_event
    }
    if (int rc = apply_pending_waits(ctx, slot >= 0 ? ctx->frame_stream[slot] : ctx->stream)) return rc;
    const int rc = hits ? launch_render<true, false>(ctx, p, out, hits, nullptr) : launch_render<false, false>(ctx, p, out, nullptr, nullptr, slot);
    if (rc) return rc;
    if (target->memory == VX_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(target->rgba32f, ctx->d_frame, pixels * pixel_bytes, hipMemcpyDeviceToHost, ctx->stream));
        if (target->hits) HIP_TRY(hipMemcpyAsync(target->hits, ctx->d_hits, pixels * sizeof(vx_hit), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return VX_OK;
}

// ---- pipelined presentation ------------------------------------------------------------------------------------------

int vx_present_begin(vx_context* ctx, const vx_uniforms* uniforms, uint32_t width, uint32_t height, int format, int* out_slot) {
    if (int rc = check_ready(ctx)) return rc;
    VX_LOCK(ctx);
    if (!out_slot) return fail(VX_ERR_INVALID_ARGUMENT, "present: null slot");
    RenderParams p;
    if (int rc = fill_params(ctx, uniforms, width, height, 0, 1, format, p)) return rc;
    if (ctx->kernel_version == 1) return fail(VX_ERR_STATE, "presentation needs the persistent kernel");
    const size_t bytes = size_t(width) * height * (p.rgba8 ? 4 : 16);
    const int k = int(ctx->present_next++ % unsigned(vx_context::kPresentSlots));
    vx_context::PresentSlot& ps = ctx->present[k];
    if (ps.busy) HIP_TRY(hipEventSynchronize(ps.copied));  // the slot's previous image has to have left the device frame
    if (ps.cap < bytes) {
        if (ps.dev) (void)hipFree(ps.dev);
        if (ps.host) (void)hipHostFree(ps.host);
        ps.dev = ps.host = nullptr;
        ps.cap = 0;
        HIP_TRY(hipMalloc(&ps.dev, bytes));
        HIP_TRY(hipHostMalloc(&ps.host, bytes, hipHostMallocDefault));
        ps.cap = bytes;
    }
    if (!ps.copied) HIP_TRY(hipEventCreateWithFlags(&ps.copied, hipEventDisableTiming));
    if (!ctx->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    // the frame on a frame stream (in rotation with the other frames in flight), its read-back on the copy stream behind it
    const int slot = ctx->frames_in_flight > 1 ? int(ctx->frame_index++ % unsigned(ctx->frames_in_flight)) : -1;
    if (int rc = apply_pending_waits(ctx, slot >= 0 ? ctx->frame_stream[slot] : ctx->stream)) return rc;
    if (int rc = launch_render<false, false>(ctx, p, static_cast<float*>(ps.dev), nullptr, nullptr, slot)) return rc;
    HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, slot >= 0 ? ctx->frame_done[slot] : ctx->render_done, 0));
    HIP_TRY(hipMemcpyAsync(ps.host, ps.dev, bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
    HIP_TRY(hipEventRecord(ps.copied, ctx->copy_stream));
    ps.bytes = bytes;
    ps.busy = true;
    *out_slot = k;
    return VX_OK;
}

int vx_present_wait(vx_context* ctx, int slot, const void** pixels, size_t* bytes) {
    if (!ctx || slot < 0 || slot >= vx_context::kPresentSlots || !pixels) return fail(VX_ERR_INVALID_ARGUMENT, "present_wait: bad argument");
    VX_LOCK(ctx);
    vx_context::PresentSlot& ps = ctx->present[slot];
    if (!ps.busy) return fail(VX_ERR_STATE, "present_wait: nothing was begun on this slot");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(ps.copied));
    *pixels = ps.host;
    if (bytes) *bytes = ps.bytes;
    return VX_OK;
}

int vx_render_counters(vx_context* ctx, const vx_uniforms* uniforms, uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_count,
                       vx_counters* out) {
    if (int rc = check_ready(ctx)) return rc;
    VX_LOCK(ctx);
    if (!out) return fail(VX_ERR_INVALID_ARGUMENT, "counters: null output");
    RenderParams p;
    if (int rc = fill_params(ctx, uniforms, width, height, tile_rank, tile_count, VX_FORMAT_RGBA32F, p)) return rc;
    HIP_TRY(hipMemsetAsync(ctx->d_counters, 0, 16 * sizeof(unsigned long long), ctx->stream));
    const bool was = ctx->profile;
    ctx->profile = false;
    const int rc = launch_render<false, true>(ctx, p, nullptr, nullptr, ctx->d_counters);
    ctx->profile = was;
    if (rc) return rc;
    unsigned long long h[16];
    HIP_TRY(hipMemcpyAsync(h, ctx->d_counters, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    out->rays = h[0]; out->iterations = h[1]; out->pushes = h[2]; out->leaf_tests = h[3]; out->leaf_tests_trilinear = h[4];
    out->boundaries = h[5]; out->csvo_header_bytes = h[6]; out->csvo_pointer_bytes = h[7];
    out->pixels = h[8]; out->lit_pixels = h[9]; out->shadow_rays = h[10];
    out->wave_steps = h[11]; out->services = h[12]; out->refills = h[13];
    out->tail_wave_steps = h[14]; out->tail_iterations = h[15];
    return VX_OK;
}

int vx_raycast(vx_context* ctx, const vx_picker_task* tasks, uint32_t count, vx_picker_result* results) {
    if (int rc = check_ready(ctx)) return rc;
    VX_LOCK(ctx);
    if (count == 0) return VX_OK;
    if (!tasks || !results) return fail(VX_ERR_INVALID_ARGUMENT, "raycast: null argument");
    if (ctx->picker_cap < count) {
        if (ctx->d_tasks) (void)hipFree(ctx->d_tasks);
        if (ctx->d_results) (void)hipFree(ctx->d_results);
        ctx->d_tasks = nullptr; ctx->d_results = nullptr; ctx->picker_cap = 0;
        const uint32_t cap = count < 128 ? 128 : count;  // the reference sizes these for 100 tasks (svo.rs:138-139)
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_tasks), size_t(cap) * sizeof(vx_picker_task)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_results), size_t(cap) * sizeof(vx_picker_result)));
        ctx->picker_cap = cap;
    }
    HIP_TRY(hipMemcpyAsync(ctx->d_tasks, tasks, size_t(count) * sizeof(vx_picker_task), hipMemcpyHostToDevice, ctx->stream));
    const size_t lds = Stack<64>::kBytes;
    const SceneArgs sc = scene_of(ctx);
    const dim3 grid((count + 63) / 64), block(64);
    if (ctx->big)
        hipLaunchKernelGGL((picker_kernel<VX_SVO_ESVO_BIG>), grid, block, lds, ctx->stream, sc, ctx->d_tasks, count, ctx->d_results);
    else if (ctx->svo_type == VX_SVO_ESVO)
        hipLaunchKernelGGL((picker_kernel<VX_SVO_ESVO>), grid, block, lds, ctx->stream, sc, ctx->d_tasks, count, ctx->d_results);
    else
        hipLaunchKernelGGL((picker_kernel<VX_SVO_CSVO>), grid, block, lds, ctx->stream, sc, ctx->d_tasks, count, ctx->d_results);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->render_done, ctx->stream));
    ctx->render_recorded = true;
    HIP_TRY(hipMemcpyAsync(results, ctx->d_results, size_t(count) * sizeof(vx_picker_result), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the reference blocks on its fence too (svo.rs:248-249)
    return VX_OK;
}

int vx_debug_trace(vx_context* ctx, const float pos[3], const float dir[3], float max_dst, int cast_translucent, vx_result* result,
                   vx_frame* frames, uint32_t max_frames, uint32_t* n_frames) {
    if (int rc = check_ready(ctx)) return rc;
    VX_LOCK(ctx);
    if (!pos || !dir || !result) return fail(VX_ERR_INVALID_ARGUMENT, "debug_trace: null argument");
    if (max_frames > 1024) max_frames = 1024;
    if (!frames) max_frames = 0;
    if (!ctx->d_trace_result) {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_trace_result), sizeof(vx_result)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_trace_count), sizeof(uint32_t)));
    }
    if (ctx->trace_cap < max_frames || !ctx->d_trace_frames) {
        if (ctx->d_trace_frames) (void)hipFree(ctx->d_trace_frames);
        ctx->d_trace_frames = nullptr;
        const uint32_t cap = max_frames < 128 ? 128 : max_frames;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_trace_frames), size_t(cap) * sizeof(vx_frame)));
        ctx->trace_cap = cap;
    }
    TraceArgs a;
    std::memcpy(a.pos, pos, sizeof a.pos);
    std::memcpy(a.dir, dir, sizeof a.dir);
    a.max_dst = max_dst;
    a.cast_translucent = cast_translucent;
    const size_t lds = Stack<64>::kBytes;
    const SceneArgs sc = scene_of(ctx);
    if (ctx->big)
        hipLaunchKernelGGL((trace_kernel<VX_SVO_ESVO_BIG>), dim3(1), dim3(64), lds, ctx->stream, sc, a, ctx->d_trace_result, ctx->d_trace_frames, max_frames,
                           ctx->d_trace_count);
    else if (ctx->svo_type == VX_SVO_ESVO)
        hipLaunchKernelGGL((trace_kernel<VX_SVO_ESVO>), dim3(1), dim3(64), lds, ctx->stream, sc, a, ctx->d_trace_result, ctx->d_trace_frames, max_frames,
                           ctx->d_trace_count);
    else
        hipLaunchKernelGGL((trace_kernel<VX_SVO_CSVO>), dim3(1), dim3(64), lds, ctx->stream, sc, a, ctx->d_trace_result, ctx->d_trace_frames, max_frames,
                           ctx->d_trace_count);
    HIP_TRY(hipGetLastError());
    uint32_t n = 0;
    HIP_TRY(hipMemcpyAsync(result, ctx->d_trace_result, sizeof(vx_result), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&n, ctx->d_trace_count, sizeof n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (frames && max_frames) {
        const uint32_t k = n < max_frames ? n : max_frames;
        if (k) HIP_TRY(hipMemcpy(frames, ctx->d_trace_frames, size_t(k) * sizeof(vx_frame), hipMemcpyDeviceToHost));
    }
    if (n_frames) *n_frames = n;
    return VX_OK;
}

int vx_sync(vx_context* ctx) {
    if (!ctx) return fail(VX_ERR_INVALID_ARGUMENT, "null context");
    wait_commit_idle(ctx);
    if (int rc = take_async_error(ctx)) return rc;
    VX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->upload_stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < vx_context::kFrameStreams; ++i) HIP_TRY(hipStreamSynchronize(ctx->frame_stream[i]));
    if (ctx->copy_stream) HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
    if (ctx->comm_stream) HIP_TRY(hipStreamSynchronize(ctx->comm_stream));
    if (ctx->order_stream) HIP_TRY(hipStreamSynchronize(ctx->order_stream));
    return VX_OK;
}

int vx_set_frames_in_flight(vx_context* ctx, int frames) {
    if (!ctx || frames < 1 || frames > vx_context::kFrameStreams) return fail(VX_ERR_INVALID_ARGUMENT, "frames in flight: 1..8");
    VX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    for (int i = 0; i < vx_context::kFrameStreams; ++i) HIP_TRY(hipStreamSynchronize(ctx->frame_stream[i]));
    ctx->frames_in_flight = frames;
    ctx->frame_index = 0;
    return VX_OK;
}

int vx_wait_event(vx_context* ctx, void* hip_event) {
    if (!ctx) return fail(VX_ERR_INVALID_ARGUMENT, "null context");
    VX_LOCK(ctx);
    ctx->pending_wait = static_cast<hipEvent_t>(hip_event);
    return VX_OK;
}

int vx_stream_wait_render(vx_context* ctx, void* stream) {
    if (!ctx) return fail(VX_ERR_INVALID_ARGUMENT, "null context");
    VX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    const int slot = ctx->last_frame_slot;
    if (slot >= 0) {
        if (ctx->frame_recorded[slot]) HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), ctx->frame_done[slot], 0));
    } else if (ctx->render_recorded) {
        HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), ctx->render_done, 0));
    }
    return VX_OK;
}

int vx_assemble_tiles(vx_context* ctx, const float* tiles, uint64_t stride_floats, uint32_t tile_count, uint32_t width, uint32_t height,
                      float* out_rgba32f) {
    if (!ctx) return fail(VX_ERR_INVALID_ARGUMENT, "null context");
    // on the context's own stream the tile lists may come from frames still in flight on the frame streams: order after them
    for (int i = 0; i < vx_context::kFrameStreams; ++i)
        if (ctx->frame_recorded[i]) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->frame_done[i], 0));
    return vx_assemble_tiles_on(ctx, tiles, stride_floats, tile_count, width, height, out_rgba32f, ctx->stream);
}

int vx_assemble_tiles_on(vx_context* ctx, const float* tiles, uint64_t stride_floats, uint32_t tile_count, uint32_t width, uint32_t height,
                         float* out_rgba32f, void* stream) {
    if (stride_floats & 3) return fail(VX_ERR_INVALID_ARGUMENT, "assemble_tiles: the stride between the ranks' lists must be whole pixels (a multiple of 4 floats)");
    return vx_assemble_tiles_format(ctx, tiles, stride_floats / 4, tile_count, width, height, out_rgba32f, VX_FORMAT_RGBA32F, stream);
}

int vx_assemble_tiles_format(vx_context* ctx, const void* tiles, uint64_t stride_pixels, uint32_t tile_count, uint32_t width, uint32_t height, void* out,
                             int format, void* stream) {
    if (!ctx || !tiles || !out || !tile_count || !width || !height || (format != VX_FORMAT_RGBA32F && format != VX_FORMAT_RGBA8))
        return fail(VX_ERR_INVALID_ARGUMENT, "assemble_tiles: bad argument");
    VX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t tiles_x = (width + kTile - 1) / kTile, tiles_y = (height + kTile - 1) / kTile;
    const vx_context::TileTable* t = nullptr;
    if (int rc = tile_table(ctx, tiles_x, tiles_y, &t)) return rc;
    const dim3 grid(tiles_x * tiles_y), block(256);  // a workgroup per tile
    if (format == VX_FORMAT_RGBA8)
        hipLaunchKernelGGL(assemble_kernel_rgba8, grid, block, 0, static_cast<hipStream_t>(stream), static_cast<const uint32_t*>(tiles), stride_pixels, tile_count,
                           width, height, tiles_x, t->d_inverse, static_cast<uint32_t*>(out));
    else
        hipLaunchKernelGGL(assemble_kernel, grid, block, 0, static_cast<hipStream_t>(stream), static_cast<const float4*>(tiles), stride_pixels, tile_count, width,
                           height, tiles_x, t->d_inverse, static_cast<float4*>(out));
    HIP_TRY(hipGetLastError());
    // On the communicator's stream the assembly reads the gathered lists -- the root's own among them, which the root renders straight
    // into (vx_gather_tiles). The newest gather's ticket therefore covers the assembly too: whoever waits for the ticket before
    // rendering into a list again (vx_wait_gather) waits for the kernel that still reads it.
    if (stream && static_cast<hipStream_t>(stream) == ctx->comm_stream && ctx->gather_index > 0) {
        const int ticket = int((ctx->gather_index - 1) % unsigned(vx_context::kGatherEvents));
        HIP_TRY(hipEventRecord(ctx->gather_done[ticket], ctx->comm_stream));
    }
    return VX_OK;
}

uint32_t vx_tile_order(uint32_t width, uint32_t height, uint32_t* out, uint32_t capacity) {
    const uint32_t tiles_x = (width + kTile - 1) / kTile, tiles_y = (height + kTile - 1) / kTile;
    if (out && capacity >= tiles_x * tiles_y) {
        std::vector<uint32_t> order, inverse;
        tile_order_host(tiles_x, tiles_y, order, inverse);
        std::memcpy(out, order.data(), order.size() * 4);
    }
    return tiles_x * tiles_y;
}

// ---- multi-GPU: the gather of the finished tiles over RCCL ---------------------------------------------------------------------

int vx_comm_unique_id(void* out_id, size_t bytes) {
    if (!out_id || bytes < sizeof(ncclUniqueId)) return fail(VX_ERR_INVALID_ARGUMENT, "comm_unique_id: needs VX_COMM_ID_BYTES (128) bytes");
    if (int rc = rccl_open()) return rc;
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    std::memcpy(out_id, &id, sizeof id);
    return VX_OK;
}

int vx_comm_init(vx_context* ctx, int nranks, int rank, const void* unique_id) {
    if (!ctx || !unique_id || nranks < 1 || rank < 0 || rank >= nranks) return fail(VX_ERR_INVALID_ARGUMENT, "comm_init: bad argument");
    VX_LOCK(ctx);
    if (ctx->comm) return fail(VX_ERR_STATE, "comm_init: this context already has a communicator");
    if (int rc = rccl_open()) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    ncclComm_t comm = nullptr;
    NCCL_TRY(g_rccl.CommInitRank(&comm, nranks, id, rank));
    if (!ctx->comm_stream) {
        // highest priority: the gather's and the assembly's few waves get the first compute-unit slots that the frames in flight
        // (persistent kernels that fill the device) give up, instead of queueing behind the next frame's waves
        int prio_least = 0, prio_greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
        HIP_TRY(hipStreamCreateWithPriority(&ctx->comm_stream, hipStreamNonBlocking, prio_greatest));
    }
    for (auto& e : ctx->gather_done)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->comm = comm;
    ctx->comm_ranks = nranks;
    ctx->comm_rank = rank;
    return VX_OK;
}

int vx_comm_destroy(vx_context* ctx) {
    if (!ctx) return fail(VX_ERR_INVALID_ARGUMENT, "null context");
    VX_LOCK(ctx);
    if (!ctx->comm) return VX_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->comm_stream) HIP_TRY(hipStreamSynchronize(ctx->comm_stream));
    NCCL_TRY(g_rccl.CommDestroy(ctx->comm));
    ctx->comm = nullptr;
    ctx->comm_ranks = ctx->comm_rank = 0;
    return VX_OK;
}

int vx_comm_info(const vx_context* ctx, int* nranks, int* rank) {
    if (!ctx) return fail(VX_ERR_INVALID_ARGUMENT, "null context");
    if (nranks) *nranks = ctx->comm_ranks;
    if (rank) *rank = ctx->comm_rank;
    return VX_OK;
}

int vx_gather_tiles(vx_context* ctx, const void* tiles, uint64_t bytes_per_rank, void* gathered, int root, int* out_ticket) {
    if (!ctx || !tiles || !bytes_per_rank || (bytes_per_rank & 3)) return fail(VX_ERR_INVALID_ARGUMENT, "gather_tiles: bad argument");
    VX_LOCK(ctx);
    if (!ctx->comm) return fail(VX_ERR_STATE, "gather_tiles: no communicator (vx_comm_init)");
    if (root < 0 || root >= ctx->comm_ranks) return fail(VX_ERR_INVALID_ARGUMENT, "gather_tiles: bad root");
    if (ctx->comm_rank == root && !gathered) return fail(VX_ERR_INVALID_ARGUMENT, "gather_tiles: the root needs a destination");
    HIP_TRY(hipSetDevice(ctx->device));
    // behind the renders issued so far (the list's among them), on the communicator's own stream: the frame streams go on with the
    // next frames meanwhile
    if (ctx->render_recorded) HIP_TRY(hipStreamWaitEvent(ctx->comm_stream, ctx->render_done, 0));
    for (int i = 0; i < vx_context::kFrameStreams; ++i)  // (every frame issued so far: a list may hold a group of frames)
        if (ctx->frame_recorded[i]) HIP_TRY(hipStreamWaitEvent(ctx->comm_stream, ctx->frame_done[i], 0));
    const size_t words = size_t(bytes_per_rank / 4);
    ProfiledLaunch ev{};
    if (ctx->profile) {  // (vx_profile_enable: the exchange bracketed by events on the communicator's stream, vx_comm_profile_read)
        if (!ctx->event_pool.empty()) {
            ev = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else {
            HIP_TRY(hipEventCreate(&ev.start));
            HIP_TRY(hipEventCreate(&ev.stop));
        }
        HIP_TRY(hipEventRecord(ev.start, ctx->comm_stream));
    }
    if (ctx->comm_rank == root) {
        uint8_t* dst = static_cast<uint8_t*>(gathered);
        // its own share (nothing to move when the root renders straight into its place in `gathered`)
        if (tiles != dst + size_t(root) * bytes_per_rank)
            HIP_TRY(hipMemcpyAsync(dst + size_t(root) * bytes_per_rank, tiles, bytes_per_rank, hipMemcpyDeviceToDevice, ctx->comm_stream));
        if (ctx->comm_ranks > 1) {
            // one receive per peer, grouped: every peer sends over its own xGMI link at the same time
            NCCL_TRY(g_rccl.GroupStart());
            for (int r = 0; r < ctx->comm_ranks; ++r)
                if (r != root) NCCL_TRY(g_rccl.Recv(dst + size_t(r) * bytes_per_rank, words, ncclUint32, r, ctx->comm, ctx->comm_stream));
            NCCL_TRY(g_rccl.GroupEnd());
        }
    } else {
        NCCL_TRY(g_rccl.Send(tiles, words, ncclUint32, root, ctx->comm, ctx->comm_stream));
    }
    if (ctx->profile) {
        HIP_TRY(hipEventRecord(ev.stop, ctx->comm_stream));
        ctx->gathers.push_back(ev);
    }
    const int ticket = int(ctx->gather_index++ % unsigned(vx_context::kGatherEvents));
    HIP_TRY(hipEventRecord(ctx->gather_done[ticket], ctx->comm_stream));
    if (out_ticket) *out_ticket = ticket;
    return VX_OK;
}

int vx_gather_query(vx_context* ctx, int ticket) {
    if (!ctx || ticket < 0 || ticket >= vx_context::kGatherEvents || !ctx->gather_done[ticket]) return -1;
    VX_LOCK(ctx);
    const hipError_t e = hipEventQuery(ctx->gather_done[ticket]);
    if (e == hipSuccess) return 1;
    (void)hipGetLastError();  // (hipErrorNotReady is not an error)
    return e == hipErrorNotReady ? 0 : -1;
}

int vx_comm_profile_read(vx_context* ctx, double* gather_ms_sum, uint32_t* gathers) {
    if (!ctx || !gather_ms_sum || !gathers) return fail(VX_ERR_INVALID_ARGUMENT, "comm_profile_read: null argument");
    VX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->comm_stream) HIP_TRY(hipStreamSynchronize(ctx->comm_stream));
    double sum = 0.0;
    for (const ProfiledLaunch& l : ctx->gathers) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, l.start, l.stop));
        sum += ms;
        ctx->event_pool.push_back(l);
    }
    *gather_ms_sum = sum;
    *gathers = uint32_t(ctx->gathers.size());
    ctx->gathers.clear();
    return VX_OK;
}

int vx_wait_gather(vx_context* ctx, int ticket) {
    if (!ctx || ticket < 0 || ticket >= vx_context::kGatherEvents) return fail(VX_ERR_INVALID_ARGUMENT, "wait_gather: bad ticket");
    VX_LOCK(ctx);
    if (!ctx->gather_done[ticket]) return fail(VX_ERR_STATE, "wait_gather: no communicator");
    ctx->pending_gather = ctx->gather_done[ticket];
    return VX_OK;
}

void* vx_comm_stream(vx_context* ctx) { return ctx ? static_cast<void*>(ctx->comm_stream) : nullptr; }

uint64_t vx_traversal_image_with_origin(int svo_type, const uint8_t* world_frame, uint64_t used_bytes, int layout, uint32_t* out_words,
                                        uint64_t capacity_words, uint32_t* out_origin_words, uint64_t origin_capacity_words) {
    if (!world_frame || layout < 0 || layout > 2 || (svo_type != VX_SVO_ESVO && svo_type != VX_SVO_CSVO)) return 0;
    vximg::WorldImage img(svo_type, layout == 0 ? vximg::kEsvo48 : (layout == 1 ? vximg::kOct64 : vximg::kOct64Wide));
    if (!img.update(world_frame, used_bytes, nullptr, 0, std::max(1u, std::min(16u, std::thread::hardware_concurrency())))) return 0;
    const vximg::ZeroedWords& f = img.frame();
    if (out_words && capacity_words >= f.size()) std::memcpy(out_words, f.data(), f.size() * 4);
    const vximg::ZeroedWords& o = img.origin();
    if (out_origin_words && origin_capacity_words >= o.size() && !o.empty()) std::memcpy(out_origin_words, o.data(), o.size() * 4);
    return f.size();
}

uint64_t vx_traversal_image(int svo_type, const uint8_t* world_frame, uint64_t used_bytes, int layout, uint32_t* out_words, uint64_t capacity_words) {
    return vx_traversal_image_with_origin(svo_type, world_frame, used_bytes, layout, out_words, capacity_words, nullptr, 0);
}

int vx_resolve_2x2(vx_context* ctx, const float* src_rgba32f, uint32_t width, uint32_t height, float* dst_rgba32f, void* stream) {
    if (!ctx || !src_rgba32f || !dst_rgba32f || !width || !height) return fail(VX_ERR_INVALID_ARGUMENT, "resolve_2x2: bad argument");
    VX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    const dim3 grid((width + 15) / 16, (height + 15) / 16), block(256);
    hipLaunchKernelGGL(resolve_2x2_kernel, grid, block, 0, static_cast<hipStream_t>(stream), reinterpret_cast<const float4*>(src_rgba32f), width, height,
                       reinterpret_cast<float4*>(dst_rgba32f));
    HIP_TRY(hipGetLastError());
    return VX_OK;
}

int vx_profile_enable(vx_context* ctx, int enabled) {
    if (!ctx) return fail(VX_ERR_INVALID_ARGUMENT, "null context");
    VX_LOCK(ctx);
    ctx->profile = enabled != 0;
    return VX_OK;
}

int vx_profile_read(vx_context* ctx, double* kernel_ms_sum, uint32_t* launches) {
    if (!ctx || !kernel_ms_sum || !launches) return fail(VX_ERR_INVALID_ARGUMENT, "profile_read: null argument");
    VX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < vx_context::kFrameStreams; ++i) HIP_TRY(hipStreamSynchronize(ctx->frame_stream[i]));
    double sum = 0.0;
    for (auto& l : ctx->launches) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, l.start, l.stop));
        sum += ms;
        ctx->event_pool.push_back(l);
    }
    *kernel_ms_sum = sum;
    *launches = uint32_t(ctx->launches.size());
    ctx->launches.clear();
    return VX_OK;
}

uint32_t vx_timeline_read(vx_context* ctx, uint64_t* out, uint32_t capacity_waves) {
    if (!ctx || !ctx->d_timeline || !out) return 0;
    VX_LOCK(ctx);
    if (hipSetDevice(ctx->device) != hipSuccess || drain_streams(ctx) != VX_OK) return 0;
    const uint32_t n = ctx->timeline_waves < capacity_waves ? ctx->timeline_waves : capacity_waves;
    if (n && hipMemcpy(out, ctx->d_timeline, size_t(n) * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
}

int vx_image_info(const vx_context* ctx, uint64_t out[4]) {
    if (!ctx || !out) return fail(VX_ERR_INVALID_ARGUMENT, "image_info: null argument");
    VX_LOCK(const_cast<vx_context*>(ctx));
    out[0] = ctx->image_ok ? (ctx->pub.layout == vximg::kOct64Wide ? 2u : 1u) : 0u;
    out[1] = ctx->image_ok ? ctx->pub.frame_bytes : 0u;
    out[2] = ctx->image_ok ? ctx->pub.origin_bytes : 0u;
    out[3] = ctx->image_ok ? ctx->pub.chunks : 0u;
    return VX_OK;
}

int vx_excursion_counters(vx_context* ctx, uint64_t out[4], int reset) {
    if (!ctx || !out) return fail(VX_ERR_INVALID_ARGUMENT, "excursion_counters: null argument");
    VX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    if (int rc = drain_streams(ctx)) return rc;
    unsigned long long h[8] = {};
    HIP_TRY(hipMemcpy(h, ctx->d_excursions, sizeof h, hipMemcpyDeviceToHost));
    out[0] = h[0]; out[1] = h[1]; out[2] = h[2]; out[3] = h[3];
    if (reset) HIP_TRY(hipMemset(ctx->d_excursions, 0, sizeof h));
    if (reset == 1) ctx->count_excursions = true;
    if (reset == 2) ctx->count_excursions = false;
    return VX_OK;
}

void* vx_stream(vx_context* ctx) { return ctx ? static_cast<void*>(ctx->stream) : nullptr; }
int vx_device(const vx_context* ctx) { return ctx ? ctx->device : -1; }

}  // extern "C"
