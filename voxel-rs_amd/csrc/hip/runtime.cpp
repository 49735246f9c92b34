// libvoxelhip.so: the host runtime behind include/voxel_hip.h.
//
// Replaces the OpenGL side of the reference's graphics::Svo (src/graphics/svo.rs:56-256): persistently mapped SSBO -> pinned staging +
// hipMalloc'd world buffer with range uploads; glDispatchCompute(world.glsl) -> render kernel; glDispatchCompute(picker.glsl) -> picker
// kernel; glFenceSync/glClientWaitSync -> HIP events. Ordinary C++ over the HIP API: the kernels are in kernels_render.hip /
// kernels_aux.hip (kernels.h says what can be launched), the RCCL exchange in comm.cpp.
// There is no CPU path: without a HIP device every entry point fails with VX_ERR_NO_DEVICE.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>

#include "vx_context.hpp"

using namespace vxd;
using namespace vxk;
using vxrt::fail;
using vxrt::g_last_error;
using vxrt::ProfiledLaunch;

namespace vxrt {
thread_local std::string g_last_error;
int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}
}  // namespace vxrt


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
    // (images of CSVO worlds: where a voxel-parent octant comes from in the world's bytes is in the image itself, the unit in front of its values)
    s.origin = c->image_ok && c->svo_type == VX_SVO_CSVO ? c->d_image : nullptr;
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

// RenderParams::tile_table / number_of_place for this launch shape: made once with tile_place / tile_number (vx_args.hpp) and kept -- the most
// recently used first, at most kLaunchTables of them (a context sees a handful of shapes: image size x rank x numbering; a caller that keeps
// resizing its window must not grow the list, and its search, without bound). Both arrays live in ONE allocation: nothing leaks when the
// second half of a set-up fails.
int launch_table(vx_context* ctx, RenderParams& p) {
    constexpr size_t kLaunchTables = 32;
    auto& tables = ctx->launch_tables;
    for (size_t i = 0; i < tables.size(); ++i) {
        const auto& t = tables[i];
        if (t.tiles_x == p.tiles_x && t.tiles_y == p.tiles_y && t.rank == p.tile_rank && t.count == p.tile_count && t.numbering == p.tile_numbering && t.strip == p.strip_w) {
            if (i) std::rotate(tables.begin(), tables.begin() + i, tables.begin() + i + 1);
            p.tile_table = tables[0].d_table;
            p.number_of_place = tables[0].d_number;
            return VX_OK;
        }
    }
    std::vector<uint32_t> order, inverse;
    if (p.tile_count > 1) tile_order_host(p.tiles_x, p.tiles_y, order, inverse);
    // [n x uint2 table][n x u32 number]
    const size_t n = p.n_local_tiles;
    std::vector<uint32_t> host(n * 3);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t place = tile_place(p, i);
        const uint32_t tile = p.tile_count > 1 ? order[size_t(place) * p.tile_count + p.tile_rank] : place;
        host[2 * size_t(i)] = (tile % p.tiles_x) | ((tile / p.tiles_x) << 16);
        host[2 * size_t(i) + 1] = place;
        host[2 * n + place] = i;
    }
    if (tables.size() >= kLaunchTables) {  // the least recently used goes; launches in flight may still read it
        if (int rc = drain_streams(ctx)) return rc;
        (void)hipFree(tables.back().d_table);
        tables.pop_back();
    }
    vx_context::LaunchTable t;
    t.tiles_x = p.tiles_x; t.tiles_y = p.tiles_y; t.rank = p.tile_rank; t.count = p.tile_count; t.numbering = p.tile_numbering; t.strip = p.strip_w;
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, host.size() * 4));
    if (const hipError_t e = hipMemcpy(d, host.data(), host.size() * 4, hipMemcpyHostToDevice); e != hipSuccess) {
        (void)hipFree(d);
        HIP_TRY(e);
    }
    t.d_table = static_cast<uint2*>(d);
    t.d_number = static_cast<uint32_t*>(d) + 2 * n;
    tables.insert(tables.begin(), t);
    p.tile_table = t.d_table;
    p.number_of_place = t.d_number;
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

// One frame: `hits` / `counters` null = an image-only render (HITS / STATS below: what the caller asked for)
int launch_render(vx_context* ctx, const RenderParams& p, float* out, vx_hit* hits, unsigned long long* counters, int slot = -1) {
    const bool HITS = hits != nullptr, STATS = counters != nullptr;
    const hipStream_t stream = slot >= 0 ? ctx->frame_stream[slot] : ctx->stream;
    uint32_t* const work_counter = slot >= 0 ? ctx->d_frame_counter[slot] : ctx->d_work_counter;
    if (p.n_local_tiles == 0) return VX_OK;
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
        HIP_TRY(vxk::launch_render_v1(ctx->svo_type, p.n_local_tiles * 4, stream, sc, p, out, hits, counters));
    } else {
        // persistent waves: as many 64-thread workgroups as the device keeps resident, fed from the sub-tile queue
        // Worlds are rendered from their traversal image; the instrumented variant stays on the world's own bytes so that its
        // counters are the reference's own fetches
        // (... and only for textures whose height is a power of two: the image kernels' sampler wraps with a mask. Any other height renders
        // on the world's own bytes, whose kernels carry the general wrap.)
        // The image holds at most `depth` levels (traversal_image.hpp). The deepest PUSH of a ray on the image of an ESVO world is into a voxel (a
        // ray that started inside it walks it as an empty node), out of a node at scale 23 - depth; on the image of a CSVO world such a ray leaves
        // for its walk on the bytes instead, and the deepest PUSH is one level higher. An image kernel's loop knows no stack but its LDS-resident
        // levels, 13 or 16: deeper worlds are rendered on their own bytes.
        const uint32_t depth = ctx->pub.depth;
        const bool csvo = ctx->svo_type == VX_SVO_CSVO;
        const uint32_t slack = csvo ? 1u : 0u;
        const bool imaged = !STATS && ctx->image_ok && (ctx->tex.height & (ctx->tex.height - 1u)) == 0u && depth <= 16u + slack;
        const bool wide = imaged && ctx->pub.layout == vximg::kOct64Wide;  // an image beyond 4 GiB: octant indices, 64-bit addresses
        // Which way a CSVO world's inside-voxel rays go is a matter of how many there are, and that of the world's size: below 4096 the
        // reference's 0.001 shadow offset survives rounding and they are a few dozen per frame -- listed, and run on the bytes when the wave is
        // done (kForeignRerun: a render loop without the walk's code). From 8192 on every second shadow ray is one (1 M per 4K frame): they walk
        // their voxels in the render loop, a sub-tile's together (walk_voxel_on_bytes) -- on the 16-level stack, whose slots below a voxel's
        // parent take what the walk pushes (three levels at depth 13, two at 14: 96 % / 82 % of the walks go no deeper).
        const bool rerun = imaged && csvo && !HITS && !wide && depth <= 12u && ctx->foreign_rerun != 0;
        vxk::RenderBuild build = {};
        build.svo = !imaged ? (ctx->big ? VX_SVO_ESVO_BIG : ctx->svo_type) : (wide ? VX_SVO_IMAGE_WIDE : VX_SVO_IMAGE);
        build.hits = imaged && HITS;
        build.foreign = !imaged || !csvo ? 0 : (rerun ? vxk::kForeignRerun : VX_SVO_CSVO);
        // 13 three-word levels where they suffice and no walk wants the room below the leaves; 16 (16-bit third plane) else
        build.levels = !imaged ? kLdsLevels : ((HITS || wide || depth > uint32_t(kLdsLevels) || (csvo && !rerun)) ? 16 : kLdsLevels);
        build.hot = imaged && ctx->hot_levels && !HITS && !wide && !csvo && build.levels == kLdsLevels;
        const void* fn = vxk::render_persistent_fn(build);
        if (!fn) return fail(VX_ERR_STATE, "no render kernel for this world (library built without it)");
        const size_t wave_lds = vxk::render_persistent_lds(build);
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
        a.total_subtiles = p.n_local_tiles * 16;
        a.refill_min = ctx->refill_min;
        a.service_min = ctx->service_min;
        a.excursions = ctx->count_excursions ? ctx->d_excursions : nullptr;
        a.timeline = vxk::timeline_build() && imaged && !HITS && !build.hot ? ctx->d_timeline : nullptr;
        a.timeline_part = ctx->timeline_part;
        a.ticket_ahead = 1u + 2u;  // (the frame's last two quarter-grids of tickets are not drawn ahead)
        a.order = nullptr;
        a.cost_cur = nullptr;
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
            if (hs->frames >= 2) {  // the table made from this view's frame before last on this stream (its kernel had a whole frame's time)
                const uint32_t g = (hs->frames - 2) % 3;
                HIP_TRY(hipStreamWaitEvent(stream, hs->order_done[g], 0));
                a.order = hs->order[g];
            } else if (hs->frames == 0 && hs->order_done[0]) {
                // a new view starts over in generation 0: what the old view's order kernels still have to write comes first
                for (int g = 0; g < 3; ++g) HIP_TRY(hipStreamWaitEvent(stream, hs->order_done[g], 0));
            }
            hs->tag += 1;
            a.cost_cur = hs->cost[hs->frames % 3];
            a.cur_tag = hs->tag;
            hs->width = p.width; hs->height = p.height; hs->tile_rank = p.tile_rank; hs->tile_count = p.tile_count;
        }
        // The queue's stretches (queue_subtile): a tile each -- or one sub-tile, where the tickets are places in the cost-ordered table. (With the
        // tiles numbered along the rows an eighth of the launch per dispenser -- every XCD its own band of the screen -- was worth 3-4 % with
        // frames in flight and cost 7 % one frame at a time; numbered in strips it is worth nothing: profiles/round4/pass_r.)
        a.stripe = 1;
        if (!a.order) a.stripe = ctx->queue_stripe > 0 ? uint32_t(ctx->queue_stripe) : 16u;
        // (a stretch longer than the launch behaves like one of its length; clamped so that queue_subtile's 32-bit products cannot wrap)
        if (a.stripe > a.total_subtiles) a.stripe = a.total_subtiles;
        if (a.stripe < 1u) a.stripe = 1u;
        a.stripe_shift = 0xffffffffu;
        for (uint32_t b = 0; b < 32u; ++b)
            if (a.stripe == (1u << b)) a.stripe_shift = b;
        // Persistent waves per CU: all that fit -- the stacks fill a CU's LDS to the last hundred bytes. A context that gathers its tiles over
        // RCCL leaves `comm_headroom` of them out: LDS of every CU stays free, in one piece, for the communication kernels' workgroups, which
        // otherwise find room only when a whole frame has drained.
        // A tile-list render (a rank's share, also of a forced one-rank run) leaves ONE slot out: what a one-wave workgroup needs -- the assembly of
        // the gathered lists on rank 0 (assemble_kernel) then starts when it is issued (-0.8 % for the fifteen waves, against 5.8 % for an
        // assembly that waits for a frame to drain: profiles/round4/pass_o).
        int per_cu_used = per_cu;
        if (ctx->waves_per_cu_cap > 0 && ctx->waves_per_cu_cap < per_cu) per_cu_used = ctx->waves_per_cu_cap;
        else if (ctx->comm_ranks > 1 && ctx->comm_headroom > 0 && per_cu > ctx->comm_headroom + 4) per_cu_used = per_cu - ctx->comm_headroom;
        else if (p.tile_count > 1 || ctx->comm) per_cu_used = per_cu > 4 ? per_cu - 1 : per_cu;
        uint32_t waves = uint32_t(ctx->cu_count) * uint32_t(per_cu_used);
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
        {
            HIP_TRY(hipStreamWaitEvent(ctx->order_stream, slot >= 0 ? ctx->frame_done[slot] : ctx->render_done, 0));
            HIP_TRY(vxk::launch_order(ctx->order_stream, hs->cost[g], hs->tag, order_subtiles, hs->order[g]));
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
    set_tile_numbering(p, ctx->tile_numbering, ctx->tile_strip);
    p.tile_order = nullptr;
    p.rgba8 = format == VX_FORMAT_RGBA8 ? 1u : 0u;
    p.opaque_lo = uint32_t(ctx->opaque_blocks);
    p.opaque_hi = uint32_t(ctx->opaque_blocks >> 32);
    if (tile_count > 1) {
        const vx_context::TileTable* t = nullptr;
        if (int rc = tile_table(ctx, p.tiles_x, p.tiles_y, &t)) return rc;
        p.tile_order = t->d_order;
    }
    p.tile_table = nullptr;
    p.number_of_place = nullptr;
    if (p.n_local_tiles) {
        if (p.tiles_x > 0xffffu || p.tiles_y > 0xffffu) return fail(VX_ERR_INVALID_ARGUMENT, "image too large (tile coordinates are 16 bits)");
        if (int rc = launch_table(ctx, p)) return rc;
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

// A whole world (gigabytes): the staging mirror is pinned and travels as it is; the traversal image lives in pageable
// memory, from which a copy command crawls (2.8 GB/s measured: 2.5 s for the depth-14 terrain's 7 GB) -- they go through a ring of pinned
// bounce buffers instead, filled by a handful of threads while the previous ones are in flight. Returns when everything has been read.
int upload_big(vx_context* ctx, const std::vector<Upload>& up) {
    constexpr size_t kSlot = size_t(32) << 20;
    constexpr int kSlots = 4;
    struct Bounce { uint8_t* host = nullptr; hipEvent_t done = nullptr; bool used = false; } ring[kSlots];
    auto release = [&]() {
        for (Bounce& b : ring) {
            if (b.host) (void)hipHostFree(b.host);
            if (b.done) (void)hipEventDestroy(b.done);
        }
    };
    const unsigned workers = std::max(1u, std::min(8u, vximg::granted_cpus()));
    int next = 0;
    for (const Upload& u : up) {
        if (!u.bytes) continue;
        const bool pinned = u.src >= ctx->staging && u.src + u.bytes <= ctx->staging + ctx->capacity + kStagingSlack;
        if (pinned) {
            if (hipMemcpyAsync(u.dst, u.src, u.bytes, hipMemcpyHostToDevice, ctx->upload_stream) != hipSuccess) {
                release();
                return fail(VX_ERR_HIP, std::string("commit: upload failed: ") + hipGetErrorString(hipGetLastError()));
            }
            continue;
        }
        for (uint64_t off = 0; off < u.bytes; off += kSlot) {
            Bounce& b = ring[next];
            next = (next + 1) % kSlots;
            if (!b.host) {
                if (hipHostMalloc(reinterpret_cast<void**>(&b.host), kSlot, hipHostMallocDefault) != hipSuccess ||
                    hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess) {
                    release();
                    return fail(VX_ERR_OUT_OF_MEMORY, "commit: no pinned memory for the upload");
                }
            }
            if (b.used && hipEventSynchronize(b.done) != hipSuccess) {
                release();
                return fail(VX_ERR_HIP, "commit: upload failed");
            }
            const size_t n = size_t(std::min<uint64_t>(kSlot, u.bytes - off));
            const size_t share = ((n + workers - 1) / workers + 4095) & ~size_t(4095);  // (rounded up first: the shares must cover n)
            std::vector<std::thread> pool;
            for (unsigned t = 1; t < workers && t * share < n; ++t)
                pool.emplace_back([&, t] { std::memcpy(b.host + t * share, u.src + off + t * share, std::min(share, n - t * share)); });
            std::memcpy(b.host, u.src + off, std::min(share, n));
            for (auto& t : pool) t.join();
            if (hipMemcpyAsync(u.dst + off, b.host, n, hipMemcpyHostToDevice, ctx->upload_stream) != hipSuccess ||
                hipEventRecord(b.done, ctx->upload_stream) != hipSuccess) {
                release();
                return fail(VX_ERR_HIP, std::string("commit: upload failed: ") + hipGetErrorString(hipGetLastError()));
            }
            b.used = true;
        }
    }
    // the caller may rewrite the staging mirror as soon as we return, and the bounce buffers go: wait for the copies to have read them
    const hipError_t e = hipStreamSynchronize(ctx->upload_stream);
    release();
    return e == hipSuccess ? VX_OK : fail(VX_ERR_HIP, "commit: upload failed");
}

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
        // (twice what is asked for, 4 MB at least: a pinned allocation takes milliseconds, and a world that streams in asks for a little more every time)
        const size_t cap = std::max(size_t(2 * total + 4096), size_t(4) << 20);
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
    // the transfer needs no fence (the device twin is private to this commit): it runs while frames in flight finish. (A kernel that reads the pinned
    // slot, not a copy command: kernels_aux.hip, copy16_kernel.)
    HIP_TRY(vxk::launch_copy16(ctx->upload_stream, slot.dev, slot.host, total));
    if (int rc = wait_for_frames()) return rc;
    HIP_TRY(vxk::launch_scatter(ctx->upload_stream, uint32_t(pieces), reinterpret_cast<const uint64_t*>(slot.dev), slot.dev));
    HIP_TRY(hipEventRecord(slot.done, ctx->upload_stream));
    slot.used = true;
    return VX_OK;
}

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
    {
        hipDeviceProp_t prop;
        CREATE_TRY(hipGetDeviceProperties(&prop, device));
        c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    CREATE_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
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
        CREATE_TRY(hipStreamCreateWithPriority(&c->frame_stream[i], hipStreamNonBlocking, prio));
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
        // The environment knobs of the product build (eleven; none of them changes a pixel): which kernel, how many frames in flight, the image on /
        // off / wide / capped / its first buffer, where a CSVO world's inside-voxel rays go, the LDS copy of the top levels, the wave slots left to a communicator,
        // the lockstep threshold, how the tiles are numbered.
        if (const char* e = std::getenv("VX_RENDER_KERNEL")) c->kernel_version = std::atoi(e) == 1 ? 1 : 2;
        if (const char* e = std::getenv("VX_FRAMES_IN_FLIGHT")) c->frames_in_flight = std::atoi(e);
        if (c->frames_in_flight < 1) c->frames_in_flight = 1;
        if (c->frames_in_flight > vx_context::kFrameStreams) c->frames_in_flight = vx_context::kFrameStreams;
        if (const char* e = std::getenv("VX_TRAVERSAL_IMAGE")) c->image_enabled = std::atoi(e) != 0;
        if (const char* e = std::getenv("VX_FOREIGN_RERUN")) c->foreign_rerun = std::atoi(e) != 0 ? 1 : 0;
        if (const char* e = std::getenv("VX_HOT_LEVELS")) c->hot_levels = std::atoi(e) != 0;
        if (const char* e = std::getenv("VX_IMAGE_CAP_BYTES")) c->image_cap_bytes = size_t(std::strtoull(e, nullptr, 10));
        // (tests: a first buffer of a few KB makes a small streamed world's image outgrow it again and again -- the carry-over on the device, commit_now)
        if (const char* e = std::getenv("VX_IMAGE_FIRST_BYTES")) c->image_first_bytes = std::max<size_t>(4096, size_t(std::strtoull(e, nullptr, 10)));
        // VX_WIDE_IMAGE=1: the layout for images beyond 4 GiB from the start; 2: and its arena starts 5 GiB into the frame, so that
        // every pointer needs more than 32 bits of byte offset (tests)
        int wide_image = 0;
        if (const char* e = std::getenv("VX_WIDE_IMAGE")) wide_image = std::atoi(e);
        c->image = vximg::WorldImage(svo_type, wide_image ? vximg::kOct64Wide : vximg::kOct64, wide_image == 2 ? (uint64_t(5) << 30) / 4 : 0);
        if (const char* e = std::getenv("VX_COMM_HEADROOM")) c->comm_headroom = std::max(0, std::atoi(e));
        if (const char* e = std::getenv("VX_SERVICE_MIN")) c->service_min = uint32_t(std::atoi(e));
        // (the order tiles are handed out in: strips of eight columns for a CSVO world -- its walks' reads of the world's bytes stay close --, places a
        // golden-ratio stride apart for an ESVO world: 0.4 % at C3, 1.0 % at 4K, nothing at 8K; the other way round CSVO loses up to 1.9 %:
        // profiles/round6/tile_numbering.sh)
        c->tile_numbering = svo_type == VX_SVO_ESVO ? 2 : 1;
        if (const char* e = std::getenv("VX_TILE_NUMBERING")) c->tile_numbering = std::atoi(e);
        // The knobs of experiments exist in the library's MEASUREMENT build only (make tl: lib/lib_tl, loaded through VX_LIB_DIR by
        // profiles/timeline.py, profiles/sweep.py and the tests that vary them): the wave timeline, the order table off, the refill threshold,
        // a cap on the resident waves, the queue's stretches, the width of the tile strips.
        if (vxk::timeline_build()) {
            if (const char* e = std::getenv("VX_HOT_FIRST")) c->hot_first = std::atoi(e) != 0;
            if (const char* e = std::getenv("VX_TIMELINE"))
                if (std::atoi(e) != 0) CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_timeline), 8192 * 8 * sizeof(unsigned long long)));
            if (const char* e = std::getenv("VX_TIMELINE_PART")) c->timeline_part = uint32_t(std::atoi(e));
            if (const char* e = std::getenv("VX_WAVES_PER_CU")) c->waves_per_cu_cap = std::atoi(e);
            if (const char* e = std::getenv("VX_REFILL_MIN")) c->refill_min = uint32_t(std::atoi(e));
            if (const char* e = std::getenv("VX_QUEUE_STRIPE")) c->queue_stripe = std::atoi(e);
            if (const char* e = std::getenv("VX_TILE_STRIP")) c->tile_strip = std::atoi(e);
        }
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
    for (auto& l : c->gathers) { (void)hipEventDestroy(l.start); (void)hipEventDestroy(l.stop); }
    if (c->staging) (void)hipHostFree(c->staging);
    if (c->h_pick_tasks) (void)hipHostFree(c->h_pick_tasks);
    if (c->h_pick_results) (void)hipHostFree(c->h_pick_results);
    void* dev[] = {c->d_world, c->d_materials, c->d_tex, c->d_frame, c->d_hits, c->d_tasks, c->d_results, c->d_trace_result, c->d_trace_frames,
                   c->d_trace_count, c->d_counters, c->d_work_counter, c->d_image, c->d_origin, c->d_excursions, c->d_main_todo, c->d_timeline};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    for (int i = 0; i < vx_context::kFrameStreams; ++i) {
        if (c->d_frame_counter[i]) (void)hipFree(c->d_frame_counter[i]);
        if (c->d_frame_todo[i]) (void)hipFree(c->d_frame_todo[i]);
        if (c->frame_done[i]) (void)hipEventDestroy(c->frame_done[i]);
        if (c->frame_stream[i]) (void)hipStreamDestroy(c->frame_stream[i]);
    }
    for (auto& t : c->launch_tables) {
        if (t.d_table) (void)hipFree(t.d_table);  // (d_number lies in the same allocation)
    }
    vxrt::comm_release(c);
    for (auto& hs : c->hot) {
        for (int g = 0; g < 3; ++g) {
            if (hs.cost[g]) (void)hipFree(hs.cost[g]);
            if (hs.order[g]) (void)hipFree(hs.order[g]);
            if (hs.order_done[g]) (void)hipEventDestroy(hs.order_done[g]);
        }
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
    if (c->probe_stream) (void)hipStreamDestroy(c->probe_stream);
    if (c->h_probe) (void)hipHostFree(c->h_probe);
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
    ctx->opaque_blocks = set;
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
    // that the image update below rewrites.
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
        // (up to 32 workers for the chunk walk of a whole world, 16 for its encoding; WorldImage::update takes no more than a sixteenth of the chunks
        // it has to walk, so an incremental commit stays on a few)
        const unsigned threads = std::max(1u, std::min(32u, vximg::granted_cpus()));
        // A world whose image will not fit a buffer resource's 32-bit offsets starts in the wide layout instead of finding that out at the end of a whole
        // build (an image is about 0.37 x the bytes of an ESVO world, 2.3 x those of a CSVO world; the wide layout serves any size)
        if (ctx->image.chunk_count() == 0 && ctx->image.layout() == vximg::kOct64 &&
            double(used_bytes) * (ctx->svo_type == VX_SVO_ESVO ? 0.42 : 2.45) >= 3.7 * double(1ull << 30))
            ctx->image = vximg::WorldImage(ctx->svo_type, vximg::kOct64Wide);
        image_ok = ctx->image.update(ctx->staging, used_bytes, changed.data(), changed.size(), threads);
        if (!image_ok && ctx->image.too_big() && ctx->image.layout() == vximg::kOct64) {
            // past what 32-bit byte offsets reach: from here on octant indices (the image is rebuilt once, whole)
            ctx->image = vximg::WorldImage(ctx->svo_type, vximg::kOct64Wide);
            image_ok = ctx->image.update(ctx->staging, used_bytes, nullptr, 0, threads);
        }
    }
    const auto t_built = std::chrono::steady_clock::now();
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
        const size_t need_origin = 0;  // (the origins of a CSVO world's voxel parents are units of the image itself since round 6: no table beside it)
        if (need > ctx->d_image_capacity || need_origin > ctx->d_origin_capacity) {
            // Grow: a buffer of twice what is needed (a streamed world's image grows commit by commit: at 1.5 x, and with everything sent again from
            // the host each time, the initial fill of the depth-14 terrain's surroundings spent 1-2 of its 0.6-2.5 s re-allocating pinned transfer
            // buffers for ever larger whole images, profiles/round6/fill_timing.sh). What the old buffer holds is the frame as the last commit left it:
            // it travels on the device, and this commit's dirty ranges follow as in any other commit. Frames in flight go on reading the old buffer,
            // which is freed when they and the copy are done.
            // (a context's first image -- a whole world's, as a rule -- gets a quarter of headroom)
            const size_t want = ctx->d_image ? 2 * need : need + need / 4;
            const size_t cap = std::min(std::max(want, ctx->image_first_bytes), ctx->image_cap_bytes ? ctx->image_cap_bytes : ~size_t(0));
            const size_t cap_origin = need_origin ? cap / 4 + kImagePad : 0;
            uint8_t* fresh = nullptr;
            bool ok = cap >= need && hipMalloc(reinterpret_cast<void**>(&fresh), cap) == hipSuccess;
            const bool carry = ok && ctx->d_image && ctx->image_ok && ctx->d_image_capacity <= cap;
            if (carry) {
                ok = hipMemcpyAsync(fresh, ctx->d_image, ctx->d_image_capacity, hipMemcpyDeviceToDevice, ctx->upload_stream) == hipSuccess &&
                     hipMemsetAsync(fresh + ctx->d_image_capacity, 0, cap - ctx->d_image_capacity, ctx->upload_stream) == hipSuccess;
            } else if (ok) {
                ok = hipMemsetAsync(fresh, 0, cap, ctx->upload_stream) == hipSuccess;
            }
            (void)drain_streams(ctx);  // (the frames that still walk the old buffer, and the copy out of it)
            if (ctx->d_image) (void)hipFree(ctx->d_image);
            if (ctx->d_origin) (void)hipFree(ctx->d_origin);
            ctx->d_image = ctx->d_origin = nullptr;
            ctx->d_image_capacity = ctx->d_origin_capacity = 0;
            if (ok && cap_origin) {
                ok = hipMalloc(reinterpret_cast<void**>(&ctx->d_origin), cap_origin) == hipSuccess &&
                     hipMemsetAsync(ctx->d_origin, 0, cap_origin, ctx->upload_stream) == hipSuccess;
            }
            if (ok) {
                ctx->d_image = fresh;
                ctx->d_image_capacity = cap;
                ctx->d_origin_capacity = cap_origin;
                whole_image = !carry;
            } else {
                if (fresh) (void)hipFree(fresh);
                (void)hipGetLastError();  // (an allocation failure is not the caller's error: the bytes path serves)
                drop_image();
            }
        }
    } else if (ctx->image_enabled && ctx->kernel_version != 1) {
        drop_image();
    }
    if (image_ok) {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(ctx->image.frame().data());
        if (whole_image) {
            up.push_back(Upload{ctx->d_image, src, ctx->image.frame_bytes()});
        } else {
            for (const vximg::Range& r : ctx->image.dirty_bytes()) up.push_back(Upload{ctx->d_image + r.start, src + r.start, r.length});
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
        if (rc == VX_OK) rc = upload_big(ctx, up);
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
    if (std::getenv("VX_COMMIT_TIMING")) {  // (measurement aid: where a commit's host time goes)
        const double* t = ctx->image.last_timing();
        const double upload_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_built).count();
        std::fprintf(stderr, "[vx_commit] %llu bytes to upload; image: root walk %.6f s, chunk walk %.6f, place %.6f, encode %.6f, root + header %.6f; allocation + upload %.6f s\n",
                     (unsigned long long)total, t[0], t[1], t[2], t[3], t[4], upload_s);
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
    const int rc = launch_render(ctx, p, out, hits, nullptr, hits ? -1 : slot);
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
    if (int rc = launch_render(ctx, p, static_cast<float*>(ps.dev), nullptr, nullptr, slot)) return rc;
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
    const int rc = launch_render(ctx, p, nullptr, nullptr, ctx->d_counters);
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
    const int svo = ctx->big ? VX_SVO_ESVO_BIG : ctx->svo_type;
    // The game's batches are small (80 tasks for an entity's AABB fan, svo_picker.rs:311-536; the reference sizes its buffers for 100, svo.rs:138-139)
    // and a call is synchronous, 250 times a second: what it costs is its round trips, not its rays. Up to kPickerDirect tasks the kernel reads
    // the tasks from, and writes the results to, pinned host memory the device sees (no copy commands: one launch, one wait); larger batches
    // (C1's 65,536 rays: 3 MB each way) travel by DMA.
    if (count <= vx_context::kPickerDirect) {
        if (!ctx->h_pick_tasks) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pick_tasks), vx_context::kPickerDirect * sizeof(vx_picker_task), hipHostMallocMapped));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_pick_results), vx_context::kPickerDirect * sizeof(vx_picker_result), hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->d_pick_tasks), ctx->h_pick_tasks, 0));
            HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->d_pick_results), ctx->h_pick_results, 0));
        }
        std::memcpy(ctx->h_pick_tasks, tasks, size_t(count) * sizeof(vx_picker_task));
        HIP_TRY(vxk::launch_picker(svo, ctx->stream, scene_of(ctx), ctx->d_pick_tasks, count, ctx->d_pick_results));
        HIP_TRY(hipStreamSynchronize(ctx->stream));  // the reference blocks on its fence too (svo.rs:248-249)
        std::memcpy(results, ctx->h_pick_results, size_t(count) * sizeof(vx_picker_result));
        return VX_OK;
    }
    if (ctx->picker_cap < count) {
        if (ctx->d_tasks) (void)hipFree(ctx->d_tasks);
        if (ctx->d_results) (void)hipFree(ctx->d_results);
        ctx->d_tasks = nullptr; ctx->d_results = nullptr; ctx->picker_cap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_tasks), size_t(count) * sizeof(vx_picker_task)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->d_results), size_t(count) * sizeof(vx_picker_result)));
        ctx->picker_cap = count;
    }
    HIP_TRY(hipMemcpyAsync(ctx->d_tasks, tasks, size_t(count) * sizeof(vx_picker_task), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(vxk::launch_picker(svo, ctx->stream, scene_of(ctx), ctx->d_tasks, count, ctx->d_results));
    HIP_TRY(hipMemcpyAsync(results, ctx->d_results, size_t(count) * sizeof(vx_picker_result), hipMemcpyDeviceToHost, ctx->stream));
    // (no event for later commits to wait for: the call returns when the stream has drained -- the reference blocks on its fence too)
    HIP_TRY(hipStreamSynchronize(ctx->stream));
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
    HIP_TRY(vxk::launch_trace(ctx->big ? VX_SVO_ESVO_BIG : ctx->svo_type, ctx->stream, scene_of(ctx), a, ctx->d_trace_result, ctx->d_trace_frames, max_frames, ctx->d_trace_count));
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
    // (pixels are moved as 16-byte words where they can be: RGBA32F lists and images are arrays of float4, RGBA8 ones of u32)
    const uintptr_t align = format == VX_FORMAT_RGBA32F ? 15u : 3u;
    if ((reinterpret_cast<uintptr_t>(tiles) & align) || (reinterpret_cast<uintptr_t>(out) & align))
        return fail(VX_ERR_INVALID_ARGUMENT, "assemble_tiles: tile lists and image must be aligned to a pixel (16 bytes RGBA32F, 4 bytes RGBA8)");
    VX_LOCK(ctx);
    HIP_TRY(hipSetDevice(ctx->device));
    const uint32_t tiles_x = (width + kTile - 1) / kTile, tiles_y = (height + kTile - 1) / kTile;
    const vx_context::TileTable* t = nullptr;
    if (int rc = tile_table(ctx, tiles_x, tiles_y, &t)) return rc;
    HIP_TRY(vxk::launch_assemble(static_cast<hipStream_t>(stream), format, tiles, stride_pixels, tile_count, width, height, tiles_x, t->d_inverse, out));
    // On the communicator's stream the assembly reads gathered lists -- the root's own among them, which the root renders straight into
    // (vx_gather_tiles). The tickets of every gather issued since the last assembly therefore cover this assembly too: whoever waits for
    // one of them before rendering into a list again (vx_wait_gather) waits for the kernel that may still read it. (Round 3 re-recorded the
    // newest ticket only: a caller that queued two gathers before assembling the first could overwrite a list under the assembly.)
    if (stream && static_cast<hipStream_t>(stream) == ctx->comm_stream) {
        for (unsigned g = ctx->assembled_index; g < ctx->gather_index; ++g)
            if (ctx->gather_index - g <= unsigned(vx_context::kGatherEvents))
                HIP_TRY(hipEventRecord(ctx->gather_done[g % unsigned(vx_context::kGatherEvents)], ctx->comm_stream));
        ctx->assembled_index = ctx->gather_index;
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

uint64_t vx_traversal_image_with_origin(int svo_type, const uint8_t* world_frame, uint64_t used_bytes, int layout, uint32_t* out_words,
                                        uint64_t capacity_words, uint32_t* out_origin_words, uint64_t origin_capacity_words) {
    if (!world_frame || layout < 0 || layout > 2 || (svo_type != VX_SVO_ESVO && svo_type != VX_SVO_CSVO)) return 0;
    vximg::WorldImage img(svo_type, layout == 0 ? vximg::kEsvo48 : (layout == 1 ? vximg::kOct64 : vximg::kOct64Wide));
    if (!img.update(world_frame, used_bytes, nullptr, 0, std::max(1u, std::min(32u, vximg::granted_cpus())))) return 0;
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
    HIP_TRY(vxk::launch_resolve_2x2(static_cast<hipStream_t>(stream), src_rgba32f, width, height, dst_rgba32f));
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

int vx_clock_probe(vx_context* ctx, uint32_t microseconds, double* shader_mhz) {
    if (!ctx || !shader_mhz) return fail(VX_ERR_INVALID_ARGUMENT, "clock_probe: null argument");
    if (microseconds < 10u || microseconds > 100000u) return fail(VX_ERR_INVALID_ARGUMENT, "clock_probe: 10 .. 100000 microseconds");
    std::lock_guard<std::mutex> one(ctx->probe_mutex);
    HIP_TRY(hipSetDevice(ctx->device));
    {
        VX_LOCK(ctx);  // (launching only: the wait below is outside, the context's own thread goes on rendering)
        if (!ctx->probe_stream) {
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            HIP_TRY(hipStreamCreateWithPriority(&ctx->probe_stream, hipStreamNonBlocking, greatest));
        }
        if (!ctx->h_probe) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_probe), 2 * sizeof(unsigned long long), hipHostMallocDefault));
        ctx->h_probe[0] = ctx->h_probe[1] = 0ull;
        HIP_TRY(vxk::launch_clock_probe(ctx->probe_stream, microseconds * 100u, ctx->h_probe));
    }
    HIP_TRY(hipStreamSynchronize(ctx->probe_stream));
    if (ctx->h_probe[1] == 0ull) return fail(VX_ERR_HIP, "clock_probe: the probe did not report");
    *shader_mhz = double(ctx->h_probe[0]) / double(ctx->h_probe[1]) * 100.0;  // cycles per 10 ns tick x 100 = MHz
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

int vx_debug_knobs(const vx_context* ctx, uint32_t out[8]) {
    if (!ctx || !out) return fail(VX_ERR_INVALID_ARGUMENT, "debug_knobs: null argument");
    VX_LOCK(const_cast<vx_context*>(ctx));
    out[0] = ctx->refill_min; out[1] = ctx->service_min; out[2] = uint32_t(ctx->waves_per_cu_cap); out[3] = uint32_t(ctx->queue_stripe);
    out[4] = uint32_t(ctx->tile_strip); out[5] = ctx->hot_first ? 1u : 0u; out[6] = uint32_t(ctx->comm_headroom); out[7] = vxk::timeline_build() ? 1u : 0u;
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

