// The render kernels of libvoxelhip.so: the persistent wavefront kernel (render_persistent: ray generation, primary traversal, shading,
// shadow traversal, sky, pixel store -- world.glsl:27-141 over svo.esvo.glsl / svo.csvo.glsl) and the first, one-thread-per-pixel version
// kept as a cross-check for the tests. The host runtime (runtime.cpp) sees none of this: it asks render_persistent_fn() for the build a
// launch needs and launches it through the HIP API.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "vx_device.hpp"
#include "vx_loop_gfx950.hpp"

// 1 = render_persistent's traversal loop on a traversal image is the hand-scheduled one (vx_loop_gfx950.hpp); 0 = the compiler's loop
// everywhere (A/B builds)
#ifndef VX_ASM_LOOP
#define VX_ASM_LOOP 1
#endif
// 1 = the library's timeline build: the image-only kernels fill in the wave timeline (PersistentArgs::timeline; make tl)
#ifndef VX_TIMELINE_BUILD
#define VX_TIMELINE_BUILD 0
#endif

using namespace vxd;
using namespace vxk;

namespace {


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
        if (HITS && hits) hits[index] = rec;
    } else if (tile_valid && p.tile_count > 1) {
        // pixels of an edge tile that fall outside the image: keep the compact tile list fully defined
        const size_t index = size_t(local_tile) * (kTile * kTile) + in_y * kTile + in_x;
        const float zero[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (out) store_pixel(p, out, index, zero);
        if (HITS && hits) memset(&hits[index], 0, sizeof(vx_hit));
    }

    if (STATS && counters) {
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

// the sub-tile (8x8 pixels: the unit of the queue) a pixel's output index lies in
__device__ __forceinline__ uint32_t subtile_of(const RenderParams& p, uint32_t out_index) {
    uint32_t local_tile, in_x, in_y;  // local_tile: the tile's place in this launch's list (a whole image: its row-major number)
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
    local_tile = p.number_of_place[local_tile];  // ... and its number in the queue
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

// A ray of a pixel has just ended after `iterations` loop iterations: the sub-tile's entry keeps the maximum.
// For a whole wave (every lane of the wave calls it; `done` = this lane's ray has just ended). With the lanes in lockstep the rays
// that end in a service phase are one sub-tile's: their maximum is found in registers (four DPP steps inside a row of 16 lanes, the four
// rows' results through scalar registers) and ONE lane notes it -- an atomic is carried out at the memory side of the L2s, 32 bytes of HBM
// write traffic each, and a wave's next wait for memory waits for it too: 375 K of them a C3 frame, 12 MB. Lanes of several sub-tiles (any
// other service_min): a note per lane, as before.
__device__ __forceinline__ void note_cost_wave(const PersistentArgs& a, const RenderParams& p, bool done, uint32_t out_index, uint32_t iterations) {
    if (!a.cost_cur) return;
    const bool noting = done && iterations >= kCostFloor;
    const unsigned long long m = __ballot(noting);
    if (m == 0ull) return;
    const uint32_t st = subtile_of(p, out_index);
    const uint32_t st0 = uint32_t(__builtin_amdgcn_readlane(int(st), int(__builtin_ctzll(m))));
    uint32_t v = noting ? (iterations < 4095u ? iterations : 4095u) : 0u;
    if (__ballot(noting && st != st0) == 0ull) {
        const uint32_t top = wave_max_u32(v);
        if (threadIdx.x == 0) atomicMax(&a.cost_cur[st0], (a.cur_tag << 12) | top);
    } else if (noting && iterations >= kCostFloor) {
        atomicMax(&a.cost_cur[st], (a.cur_tag << 12) | v);
    }
}

__device__ __forceinline__ uint32_t rank_in(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)); }
__device__ __forceinline__ uint32_t fbits(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float bitsf(uint32_t u) { return __uint_as_float(u); }

// IMAGE = the rays walk the traversal image of the world (traversal_image.hpp) instead of its own bytes: a build per way of addressing it (a buffer
// resource / a 64-bit base). An image kernel is only launched for worlds whose depth its LDS-resident stack levels
// cover (LV: 13 three-word levels, or 16 with a 16-bit third plane -- image cursors need no more of the third word), so its traversal loop
// never hands a ray over to the spill-backed stack; deeper worlds are rendered on their own bytes.
// FOREIGN (an image of a CSVO world): what happens to a ray that is about to be led into the voxel it started in -- VX_SVO_CSVO: it walks the
// voxel on the world's own bytes in a service phase and comes back to the image (vx_device.hpp, walk_voxel_on_bytes); kForeignRerun: it is
// listed and run on the bytes afterwards. (The image of an ESVO world serves such rays itself.)
// HOT (experiment X1): the image's root octant and its eight child octants copied into LDS, PUSHes out of them served from there.
// HITS / STATS: hit records and the instrumented counters are written where the pointers are not null; the kernels
// on the world's own bytes are built once per format with both (they are the fall-back and the instrumented path, not the fast one).
constexpr int min_waves_of(int svo, bool hits) { return ((svo == VX_SVO_IMAGE || svo == VX_SVO_IMAGE_WIDE) && !hits) ? 4 : 1; }
template <int SVO, bool HITS, bool STATS, int FOREIGN = 0, int LV = kLdsLevels, bool HOT = false>
__global__ __launch_bounds__(64, min_waves_of(SVO, HITS)) void render_persistent(SceneArgs sa, RenderParams p, PersistentArgs a, float4* __restrict__ out,
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
    constexpr bool SHALLOW = IMAGE;  // no ray can push below the LDS-resident stack levels (the host launches an image kernel only for worlds they cover)
    // (the image kernels are only launched for textures whose height is a power of two -- launch_render -- and say so to the sampler, a literal the
    // compiler folds: REPEAT is a mask, nothing of the general wrap is in these kernels' code -- 1-3 % of a frame, profiles/round3/pass_af)
    // TL: the library's timeline build (make tl: -DVX_TIMELINE_BUILD=1; profiles/timeline.py) fills in the wave timeline. Everywhere else the
    // instrumentation is compiled out, not switched off: its stamps and counters are wave-uniform state that lives through the whole kernel, and
    // with them the ESVO image kernel spilled 123 scalar registers instead of 50 and was 13 % longer (C3 +2 % without: profiles/round3/pass_ag).
    constexpr bool TL = VX_TIMELINE_BUILD != 0 && IMAGE && !HITS && !STATS && !HOT;
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
    // A pixel's shadow ray starts in (or at) the voxel its primary ray hit, and the primary's stack holds that voxel's ancestors: the shadow ray takes
    // the levels down to the voxel's parent in one go instead of a trip of the loop each (Trav::descend_along). Image cursors on the loop's own stack;
    // the build with the LDS copy of the top levels keeps the plain descent (it is a cross-check).
    constexpr bool kDescendAlong = IMAGE && !HOT;
    vx_hit rec;            // HITS only
    uint32_t steps = 0;    // HITS only
    Counters ctr = {};
    uint32_t n_pixels = 0, lit = 0, shadow_rays = 0;
    uint32_t wave_steps = 0, services = 0, refills = 0, tail_wave_steps = 0, tail_iterations = 0;  // STATS only, wave-uniform

    uint32_t cursor = 64, sub = 0;  // wave-uniform: position inside the current sub-tile
    bool queue_empty = false;
    const unsigned long long t_start = a.timeline ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long t_empty = 0ull;
    uint32_t taken = 0;
    uint32_t in_service = 0, service_phases = 0;  // timeline only, wave-uniform
    // timeline only: shader-clock stamps (s_memtime) -- the wave's whole life and the part of it spent in the traversal loop -- and the loop's trips
    const unsigned long long c_start = a.timeline ? __builtin_amdgcn_s_memtime() : 0ull;
    unsigned long long loop_cycles = 0ull;
    uint32_t loop_trips = 0;
    unsigned long long loop_tails = 0ull;  // timeline only: of the hand-scheduled loop's trips, those that took its ADVANCE-only tail | its PUSH-only tail << 32
    // the sub-tile queue: a ticket is this launch's sub-tile number (lane 0's value counts)
    // Wave w = 8 j + c starts on dispenser c's j-th sub-tile without asking: 4096 waves do not open the frame by queueing at the counters. The
    // dispenser hands out its sub-tiles from there on.
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
    // the n-th ticket of dispenser c is its (first_c + n)-th sub-tile (queue_subtile), first_c = how many of the waves' own first sub-tiles are c's
    // (beyond the launch's last sub-tile: the dispenser is dry -- its sub-tiles' numbers grow with k)
    auto region_ticket = [&](uint32_t k, uint32_t queue) -> uint32_t { return queue_subtile(k, queue, a.stripe, a.stripe_shift); };
    auto ticket_of = [&](uint32_t raw, uint32_t queue) -> uint32_t {
        return region_ticket(((gridDim.x + kQueues - 1u - queue) >> 3) + uint32_t(__builtin_amdgcn_readfirstlane(raw)), queue);
    };
    uint32_t ticket_raw = 0, ticket_queue = my_queue;  // the ticket drawn ahead: its raw count and the dispenser it came from
    bool ticket_ahead = true, ticket_first = true;   // wave-uniform; the wave's first ticket is its own number: nothing was drawn
    // the ticket as the wave's value; a dispenser that has run dry sends the wave on to the next one (the frame's last stretch only)
    auto settle_ticket = [&]() -> uint32_t {
        uint32_t t = ticket_ahead ? (ticket_first ? region_ticket(blockIdx.x >> 3, my_queue) : ticket_of(ticket_raw, ticket_queue)) : ticket_of(draw_raw(), my_queue);
        ticket_ahead = false;
        ticket_first = false;
        for (uint32_t tried = 1; t >= a.total_subtiles && tried < kQueues; ++tried) {
            my_queue = (my_queue + 1u) & (kQueues - 1u);
            t = ticket_of(draw_raw(), my_queue);
        }
        return t;
    };
    if (blockIdx.x == 0 && lane < kQueues) a.next_counter[lane * kQueueStride] = 0u;
    uint32_t walk_cycles = 0, walk_trips = 0, walk_phases = 0;  // timeline only, wave-uniform: the walks inside voxels -- shader-clock cycles, trips of the walk's loop (its slowest lane's iterations), phases
    uint32_t my_chunk = 0, my_fill = 0;  // FOREIGN, wave-uniform: newest chunk of this wave's list (+ 1) and its fill

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
        constexpr bool kAsmLoop = VX_ASM_LOOP != 0 && IMAGE && (LV == kLdsLevels || LV == 16) && !HOT && !STATS;
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
                // (FOREIGN = VX_SVO_CSVO: the loop also ends as soon as a lane waits for its walk into a voxel -- the shadow rays of a sub-tile get there
                // in the same trip; the build that lists such rays instead never waits for anything)
                const uint32_t f_waiting = FOREIGN == VX_SVO_CSVO ? uint32_t(__popcll(__ballot(state == kForeign))) : 0u;
                const uint32_t f_min = FOREIGN == VX_SVO_CSVO ? 1u : 0xffffffffu;
                constexpr int kForeignKind = FOREIGN == VX_SVO_CSVO ? 1 : (FOREIGN == kForeignRerun ? 2 : 0);  // (waits for its walk / is listed / ESVO world)
                if (a.timeline) traverse_loop_gfx950<SVO, kForeignKind, true, LV>(tr, sc.records8, sc.wide, lds_slot0, lds_aux0, keep_going, f_waiting, f_min, loop_trips, &loop_tails);
                else if (keep_going == 0u) traverse_loop_gfx950<SVO, kForeignKind, false, LV, true>(tr, sc.records8, sc.wide, lds_slot0, lds_aux0, keep_going, f_waiting, f_min, loop_trips);
                else traverse_loop_gfx950<SVO, kForeignKind, false, LV>(tr, sc.records8, sc.wide, lds_slot0, lds_aux0, keep_going, f_waiting, f_min, loop_trips);
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

        // ---- ... or (FOREIGN = VX_SVO_CSVO) the walk inside the voxel on the world's own bytes ----
        VX_PART_BEGIN(5);
        constexpr bool kOpaqueFastPath = !STATS;
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
            if (__builtin_expect(fm != 0ull, 0)) {
                uint32_t on_bytes = 0;
                bool given_up = false;
                walk_phase = true;
                const unsigned long long c_walk = a.timeline ? __builtin_amdgcn_s_memtime() : 0ull;
                if (state == kForeign) {
                    walked = true;
                    tr.iter &= ~kParked;
                    const uint32_t before = tr.iter;
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
                if (a.timeline) {
                    uint32_t most = on_bytes;
                    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(most, off, 64); most = o > most ? o : most; }
                    walk_cycles += uint32_t(__builtin_amdgcn_s_memtime() - c_walk);
                    walk_trips += uint32_t(__builtin_amdgcn_readfirstlane(most));
                    ++walk_phases;
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
        // (kDescendAlong) a shadow ray set up in this phase: the voxel its primary hit -- its un-mirrored corner, its parent's scale
        bool along = false;
        float along_q[3] = {0, 0, 0};
        int along_scale = 0;

        // ---- finished rays ----
        VX_PART_BEGIN(2);
        note_cost_wave(a, p, state == kDone && serve, out_index, tr.iter & ~kParked);
        if (state == kDone && serve) {
            float color[4];
            bool write = true;
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
                        if constexpr (kDescendAlong) {
                            // the primary's cursor still stands at the voxel (a hit found by a walk inside a voxel does not: its cursor is among phantom
                            // nodes): the voxel's parent joins its ancestors on the stack, the voxel's corner says where the path leads
                            along = !walked;
                            tr.cell_corner(along_q);
                            along_scale = tr.scale;
                            if (along) fast_st.push(tr.scale, tr.ptr, tr.t_max, tr.node);
                        }
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
                if (HITS && hits) {
                    rec.steps = steps;
                    hits[out_index] = rec;
                }
                state = kIdle;
            }
        }

        VX_PART_END(2);
        // ---- refill idle lanes from the sub-tile queue ----
        VX_PART_BEGIN(3);
        unsigned long long idle_mask = __ballot(state == kIdle);
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
                    sub = a.order ? uint32_t(__builtin_amdgcn_readfirstlane(a.order[t])) : t;  // most expensive first, or the order of their numbers
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
                    // (the queue's tile number -> the tile's place in the launch's list: RenderParams::tile_numbering)
                    const uint32_t number = sub >> 4, s = sub & 15u;
                    const uint2 entry = p.tile_table[number];  // (wave-uniform: one scalar load)
                    const uint32_t local_tile = entry.y;
                    const uint32_t tx = entry.x & 0xffffu, ty = entry.x >> 16;
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
                        if (HITS && hits) memset(&hits[out_index], 0, sizeof(vx_hit));
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
        if constexpr (kDescendAlong) {
            if (__ballot(along) != 0ull) {  // (wave-uniform: a batch of shadow rays)
                if (along) {
                    constexpr bool kByHand = VX_ASM_LOOP != 0 && (LV == kLdsLevels || LV == 16);
                    if constexpr (kByHand) {
                        if (along_scale >= FastStack::kBaseScale && along_scale < kMaxScale) {
                            const int from = tr.scale;
                            descend_along_gfx950<LV>(tr, uint32_t(reinterpret_cast<uintptr_t>(fast_st.at(0))) + fast_st.slot0, along_scale, along_q);
                            if (tr.scale != from) {  // the node the cursor has reached: the path's at this scale
                                float unused;
                                fast_st.pop(tr.scale, tr.ptr, unused, tr.node);
                            }
                        }
                    } else {
                        tr.descend_along(fast_st, along_scale, along_q);
                    }
                }
            }
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
        row[4] = __builtin_amdgcn_s_memtime() - c_start; row[5] = loop_cycles;
        // trips of the loop | of which every lane ADVANCEd << 20 | of which every lane PUSHed << 40 (the hand-scheduled loop's cheap tails)
        row[6] = (unsigned long long)(loop_trips & 0xfffffu) | ((loop_tails & 0xfffffull) << 20) | (((loop_tails >> 32) & 0xfffffull) << 40);
        row[7] = ((unsigned long long)(walk_phases & 0xfffu) << 52) | ((unsigned long long)(walk_trips & 0xfffffu) << 32) | walk_cycles;  // the walks inside voxels: phases, trips of their loop, cycles
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
                    if (HITS && hits) hits[index] = r;
                }
            }
        }
    }

    if (STATS && counters) {
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


}  // namespace

namespace vxk {

#define VX_K(...) reinterpret_cast<const void*>(&render_persistent<__VA_ARGS__>)
// The builds that exist (14): on the world's own bytes one per format, with hit records and counters; on a traversal image, image-only: worlds
// without walks (ESVO) on 13 levels / 16 levels / 16 levels in the wide layout; CSVO worlds whose inside-voxel rays are listed (at most 12
// levels: 13-level stack) or walk (16 levels, both layouts); with hit records: 16 levels, both layouts, with and without the walk; the LDS
// copy of the top levels (experiment X1).
const void* render_persistent_fn(const RenderBuild& b) {
    const bool image = b.svo == VX_SVO_IMAGE || b.svo == VX_SVO_IMAGE_WIDE, wide = b.svo == VX_SVO_IMAGE_WIDE;
    if (!image) {
        if (b.foreign != 0 || b.hot) return nullptr;
        return b.svo == VX_SVO_ESVO ? VX_K(VX_SVO_ESVO, true, true) : (b.svo == VX_SVO_ESVO_BIG ? VX_K(VX_SVO_ESVO_BIG, true, true) : (b.svo == VX_SVO_CSVO ? VX_K(VX_SVO_CSVO, true, true) : nullptr));
    }
    if (b.hot) return (!wide && !b.hits && b.foreign == 0 && b.levels == kLdsLevels) ? VX_K(VX_SVO_IMAGE, false, false, 0, kLdsLevels, true) : nullptr;
    if (b.hits) {
        if (b.levels != 16) return nullptr;
        if (b.foreign == 0) return wide ? VX_K(VX_SVO_IMAGE_WIDE, true, false, 0, 16) : VX_K(VX_SVO_IMAGE, true, false, 0, 16);
        if (b.foreign == VX_SVO_CSVO) return wide ? VX_K(VX_SVO_IMAGE_WIDE, true, false, VX_SVO_CSVO, 16) : VX_K(VX_SVO_IMAGE, true, false, VX_SVO_CSVO, 16);
        return nullptr;
    }
    if (b.foreign == 0) {
        if (b.levels == 16) return wide ? VX_K(VX_SVO_IMAGE_WIDE, false, false, 0, 16) : VX_K(VX_SVO_IMAGE, false, false, 0, 16);
        return (b.levels == kLdsLevels && !wide) ? VX_K(VX_SVO_IMAGE, false, false, 0, kLdsLevels) : nullptr;
    }
    if (b.foreign == kForeignRerun) return (!wide && b.levels == kLdsLevels) ? VX_K(VX_SVO_IMAGE, false, false, kForeignRerun, kLdsLevels) : nullptr;
    if (b.foreign == VX_SVO_CSVO && b.levels == 16) return wide ? VX_K(VX_SVO_IMAGE_WIDE, false, false, VX_SVO_CSVO, 16) : VX_K(VX_SVO_IMAGE, false, false, VX_SVO_CSVO, 16);
    return nullptr;
}
#undef VX_K

size_t render_persistent_lds(const RenderBuild& b) {
    if (b.hot) return Stack<64, false, false, kLdsLevels, true, true>::kBytes;
    return b.levels == 16 ? Stack<64, false, false, 16, true>::kBytes : Stack<64>::kBytes;
}

bool timeline_build() { return VX_TIMELINE_BUILD != 0; }

hipError_t launch_render_v1(int svo, uint32_t blocks, hipStream_t stream, const SceneArgs& sc, const RenderParams& p, void* out, vx_hit* hits, unsigned long long* counters) {
    const size_t lds = Stack<kBlockThreads>::kBytes;
    if (svo == VX_SVO_ESVO)
        hipLaunchKernelGGL((render_kernel<VX_SVO_ESVO, true, true>), dim3(blocks), dim3(kBlockThreads), lds, stream, sc, p, static_cast<float4*>(out), hits, counters);
    else if (svo == VX_SVO_CSVO)
        hipLaunchKernelGGL((render_kernel<VX_SVO_CSVO, true, true>), dim3(blocks), dim3(kBlockThreads), lds, stream, sc, p, static_cast<float4*>(out), hits, counters);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace vxk
