// C entry points over the C++ host mirror (chunks, world SVOs), for the Python harness (tests, bench, smoke).
// Nothing here is on the render path: it builds the same serialized buffers the reference's Rust host code
// hands to `graphics::Svo::update` (src/graphics/svo.rs:171-189).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <cstring>
#include <memory>
#include <new>

#include "chunk.hpp"
#include "csvo.hpp"
#include "esvo.hpp"
#include "graphics_svo.hpp"
#include "physics.hpp"
#include "scene.hpp"
#include "stream.hpp"
#include "svo_picker.hpp"
#include "svo_registry.hpp"
#include "traversal_image.hpp"  // (csrc/hip: host-only header)
#include "worldsvo.hpp"

using namespace vx;

namespace {

struct World {
    int svo_type;  // 1 = ESVO, 2 = CSVO (svo.rs:35-36)
    Esvo<EsvoSerializedChunk> esvo;
    Csvo csvo;
};

}  // namespace

extern "C" {

// ---- chunks -------------------------------------------------------------------------------------------

void* vxh_chunk_new(int32_t x, int32_t y, int32_t z, uint32_t lod) { return new (std::nothrow) Chunk(ChunkPos{x, y, z}, uint8_t(lod)); }
void vxh_chunk_free(void* c) { delete static_cast<Chunk*>(c); }
void vxh_chunk_set_block(void* c, uint32_t x, uint32_t y, uint32_t z, uint32_t id) { static_cast<Chunk*>(c)->set_block(x, y, z, id); }
uint32_t vxh_chunk_get_block(void* c, uint32_t x, uint32_t y, uint32_t z) { return static_cast<Chunk*>(c)->get_block(x, y, z); }
void vxh_chunk_compact(void* c) { static_cast<Chunk*>(c)->storage.compact(); }
void vxh_chunk_set_lod(void* c, uint32_t lod) { static_cast<Chunk*>(c)->lod = uint8_t(lod); }

// ids[x + 32 * (y + 32 * z)], 0 = empty; rebuilds the storage bottom-up like Chunk::fill_with
void vxh_chunk_fill_dense(void* c, const uint32_t* ids) {
    static_cast<Chunk*>(c)->fill_with([ids](uint32_t x, uint32_t y, uint32_t z) -> std::optional<BlockId> {
        const uint32_t v = ids[x + 32u * (y + 32u * z)];
        if (v == NO_BLOCK) return std::nullopt;
        return v;
    });
}

uint64_t vxh_chunk_pos_hash(int32_t x, int32_t y, int32_t z) { return chunk_pos_hash(ChunkPos{x, y, z}); }

// ---- world SVO (world::hds::WorldSvo, src/world/hds/common.rs:3-15) -------------------------------------

void* vxh_world_new(int svo_type) {
    if (svo_type != 1 && svo_type != 2) return nullptr;
    World* w = new (std::nothrow) World();
    if (w) w->svo_type = svo_type;
    return w;
}
void vxh_world_free(void* w) { delete static_cast<World*>(w); }

// serializes the chunk (SerializedChunk::new) and places it at the given SVO-space chunk position
int vxh_world_set_chunk(void* wp, uint32_t px, uint32_t py, uint32_t pz, const void* chunk, int serialize) {
    World* w = static_cast<World*>(wp);
    const Chunk& c = *static_cast<const Chunk*>(chunk);
    if (w->svo_type == 1) w->esvo.set_leaf(Position{px, py, pz}, EsvoSerializedChunk(c), serialize != 0);
    else w->csvo.set_leaf(Position{px, py, pz}, CsvoSerializedChunk(c), serialize != 0);
    return 0;
}

void vxh_world_serialize(void* wp) {
    World* w = static_cast<World*>(wp);
    if (w->svo_type == 1) w->esvo.serialize();
    else w->csvo.serialize();
}

uint32_t vxh_world_depth(const void* wp) {
    const World* w = static_cast<const World*>(wp);
    return w->svo_type == 1 ? w->esvo.depth() : w->csvo.depth();
}

size_t vxh_world_size_in_bytes(const void* wp) {
    const World* w = static_cast<const World*>(wp);
    return w->svo_type == 1 ? w->esvo.size_in_bytes() : w->csvo.size_in_bytes();
}

// bytes of the header the writer puts in front of the arena (ESVO preamble 20, CSVO root pointer 4)
size_t vxh_world_header_bytes(const void* wp) { return static_cast<const World*>(wp)->svo_type == 1 ? 20 : 4; }

size_t vxh_world_write_to(const void* wp, uint8_t* dst) {
    const World* w = static_cast<const World*>(wp);
    return w->svo_type == 1 ? w->esvo.write_to(dst) : w->csvo.write_to(dst);
}

int vxh_world_write_changes_to(void* wp, uint8_t* dst, size_t dst_len, int reset) {
    World* w = static_cast<World*>(wp);
    const bool ok = w->svo_type == 1 ? w->esvo.write_changes_to(dst, dst_len, reset != 0) : w->csvo.write_changes_to(dst, dst_len, reset != 0);
    return ok ? 0 : 1;
}

// copies up to `max` (start,length) pairs of the dirty ranges (arena offsets); returns the count
size_t vxh_world_updated_ranges(const void* wp, uint64_t* out_pairs, size_t max) {
    const World* w = static_cast<const World*>(wp);
    const std::vector<Range>& r = w->svo_type == 1 ? w->esvo.buffer.updated_ranges : w->csvo.buffer.updated_ranges;
    for (size_t i = 0; i < r.size() && i < max; ++i) {
        out_pairs[2 * i] = r[i].start;
        out_pairs[2 * i + 1] = r[i].length;
    }
    return r.size();
}

// The whole mapped-buffer image `graphics::Svo::update` produces: [f32 2^-depth][header][arena].
// Returns the bytes needed; writes only if `cap` suffices.
size_t vxh_world_frame(const void* wp, uint8_t* dst, size_t cap) {
    const World* w = static_cast<const World*>(wp);
    const size_t need = 4 + vxh_world_header_bytes(wp) + vxh_world_size_in_bytes(wp);
    if (!dst || cap < need) return need;
    const float scale = std::exp2(-float(vxh_world_depth(wp)));
    std::memcpy(dst, &scale, 4);
    if (w->svo_type == 1) w->esvo.write_to(dst + 4);
    else w->csvo.write_to(dst + 4);
    return need;
}

// ---- synthetic scenes (bench / parity at scale) --------------------------------------------

// Fills `world` with the seeded heightfield scene of the given depth; returns the number of chunks placed.
uint64_t vxh_scene_build_heightfield(void* wp, uint32_t depth, uint32_t seed, uint32_t n_threads, uint64_t* out_leaves, uint32_t* out_hmax) {
    World* w = static_cast<World*>(wp);
    SceneStats st;
    if (w->svo_type == 1) st = build_heightfield_scene(w->esvo, depth, seed, n_threads);
    else st = build_heightfield_scene(w->csvo, depth, seed, n_threads);
    if (out_leaves) *out_leaves = st.leaves;
    if (out_hmax) *out_hmax = st.h_max;
    return st.chunks;
}

uint32_t vxh_scene_height(uint32_t depth, uint32_t seed, uint32_t x, uint32_t z) { return heightfield_height(depth, seed, x, z); }
uint32_t vxh_scene_hash32(uint32_t seed, uint32_t o, uint32_t i, uint32_t j) { return scene_hash32(seed, o, i, j); }

// ---- picker batches (src/graphics/svo_picker.rs) --------------------------------------------------------------------

namespace {
PickerBatch make_batch(const float* rays, uint32_t n_rays, const float* aabbs, uint32_t n_aabbs) {
    PickerBatch b;
    for (uint32_t i = 0; i < n_rays; ++i) {
        const float* r = rays + 7 * i;  // pos, dir, max_dst
        b.add_ray(Vec3{r[0], r[1], r[2]}, Vec3{r[3], r[4], r[5]}, r[6]);
    }
    for (uint32_t i = 0; i < n_aabbs; ++i) {
        const float* a = aabbs + 9 * i;  // pos, offset, extents
        b.add_aabb(Aabb{Vec3{a[0], a[1], a[2]}, Vec3{a[3], a[4], a[5]}, Vec3{a[6], a[7], a[8]}});
    }
    return b;
}
}  // namespace

// PickerBatch::serialize_tasks; returns the task count (writes at most `max`)
uint32_t vxh_picker_serialize(const float* rays, uint32_t n_rays, const float* aabbs, uint32_t n_aabbs, vx_picker_task* out, uint32_t max) {
    std::vector<vx_picker_task> tasks;
    make_batch(rays, n_rays, aabbs, n_aabbs).serialize_tasks(tasks);
    for (size_t i = 0; i < tasks.size() && i < max; ++i) out[i] = tasks[i];
    return uint32_t(tasks.size());
}

// PickerBatch::deserialize_results: out_rays = n_rays x {dst, inside, pos[3], normal[3]}, out_aabbs = n_aabbs x {neg[3], pos[3]}
void vxh_picker_deserialize(const float* rays, uint32_t n_rays, const float* aabbs, uint32_t n_aabbs, const vx_picker_result* results,
                            float* out_rays, float* out_aabbs) {
    PickerBatchResult res;
    make_batch(rays, n_rays, aabbs, n_aabbs).deserialize_results(results, res);
    for (size_t i = 0; i < res.rays.size(); ++i) {
        const RayResult& r = res.rays[i];
        const float v[8] = {r.dst, r.inside_voxel ? 1.0f : 0.0f, r.pos.x, r.pos.y, r.pos.z, r.normal.x, r.normal.y, r.normal.z};
        std::memcpy(out_rays + 8 * i, v, sizeof v);
    }
    for (size_t i = 0; i < res.aabbs.size(); ++i) {
        const AabbResult& a = res.aabbs[i];
        const float v[6] = {a.neg.x, a.neg.y, a.neg.z, a.pos.x, a.pos.y, a.pos.z};
        std::memcpy(out_aabbs + 6 * i, v, sizeof v);
    }
}

// ---- physics (src/systems/physics.rs) -----------------------------------------------------------------------------------
// entity record = 17 floats: position[3], velocity[3], aabb offset[3], aabb extents[3], wall_clip, flying, gravity,
// max_fall_velocity, is_grounded (in: ignored, out: state after the step)

namespace {
constexpr uint32_t kEntityFloats = 17;
systems::Entity entity_from(const float* e) {
    systems::Entity out(Vec3{e[0], e[1], e[2]}, systems::AABBDef{Vec3{e[6], e[7], e[8]}, Vec3{e[9], e[10], e[11]}});
    out.velocity = Vec3{e[3], e[4], e[5]};
    out.caps = systems::EntityCapabilities{e[12] != 0.0f, e[13] != 0.0f, e[14], e[15]};
    return out;
}
void entity_to(const systems::Entity& in, float* e) {
    e[0] = in.position.x; e[1] = in.position.y; e[2] = in.position.z;
    e[3] = in.velocity.x; e[4] = in.velocity.y; e[5] = in.velocity.z;
    e[16] = in.state.is_grounded ? 1.0f : 0.0f;
}

// Raycaster over a raw vx_context: what graphics::Svo::raycast does (svo.rs:233-255)
struct ContextRaycaster final : systems::Raycaster {
    vx_context* ctx;
    mutable std::vector<vx_picker_task> tasks;
    mutable std::vector<vx_picker_result> results;
    mutable uint64_t total_tasks = 0;
    explicit ContextRaycaster(vx_context* c) : ctx(c) {}
    void raycast(PickerBatch& batch, PickerBatchResult& result) const override {
        const size_t n = batch.serialize_tasks(tasks);
        results.resize(n);
        total_tasks += n;
        if (n && vx_raycast(ctx, tasks.data(), uint32_t(n), results.data()) != VX_OK) throw std::runtime_error(vx_last_error());
        batch.deserialize_results(results.data(), result);
    }
};
}  // namespace

// Physics::step_many `steps` times through vx_raycast on `ctx` (one picker launch per step for all entities).
// Returns the number of picker tasks cast in total, or -1 (see vx_last_error()).
int64_t vxh_physics_step_many(void* ctx, float delta_time, uint32_t steps, float* entities, uint32_t n) {
    try {
        std::vector<systems::Entity> es;
        for (uint32_t i = 0; i < n; ++i) es.push_back(entity_from(entities + kEntityFloats * i));
        systems::Physics physics;
        ContextRaycaster rc(static_cast<vx_context*>(ctx));
        for (uint32_t s = 0; s < steps; ++s) physics.step_many(delta_time, rc, es);
        for (uint32_t i = 0; i < n; ++i) entity_to(es[i], entities + kEntityFloats * i);
        return int64_t(rc.total_tasks);
    } catch (const std::exception&) {
        return -1;
    }
}

// Physics::update_entity with externally supplied AabbResults (6 floats each: neg[3], pos[3]) -- lets a test stand an
// oracle in for the raycaster
void vxh_physics_update(float delta_time, float* entities, const float* aabb_results, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) {
        systems::Entity e = entity_from(entities + kEntityFloats * i);
        const float* r = aabb_results + 6 * i;
        systems::Physics::update_entity(e, AabbResult{Vec3{r[0], r[1], r[2]}, Vec3{r[3], r[4], r[5]}}, delta_time);
        entity_to(e, entities + kEntityFloats * i);
    }
}

// ---- world streaming (chunk loader -> generated chunks -> dirty ranges -> vx_commit) --------------------------------------

namespace {
struct Streamer {
    int svo_type;
    systems::WorldStreamer<Esvo<EsvoSerializedChunk>, EsvoSerializedChunk> esvo;
    systems::WorldStreamer<Csvo, CsvoSerializedChunk> csvo;
    Streamer(int t, uint32_t depth, uint32_t seed, uint32_t radius, int32_t y0, int32_t y1)
        : svo_type(t), esvo(depth, seed, radius, y0, y1, workers()), csvo(depth, seed, radius, y0, y1, workers()) {}
    // (what the process is GRANTED, not what the machine has: a cgroup quota of 16 on 256 logical CPUs throttles every thread beyond the sixteenth)
    static uint32_t workers() { return std::max(1u, std::min(16u, vximg::granted_cpus())); }
    // tests: a host-side stand-in for the device world buffer and the traversal image vx_commit keeps next to it
    std::vector<uint8_t> mirror;
    std::unique_ptr<vximg::WorldImage> image;
    bool image_ok = false;
};

extern "C++" {
template <class S>
void mirror_commits(Streamer* s, S& streamer) {
    streamer.on_dry_commit = [s](auto& world, const std::vector<vx_range>& ranges) {
        const float scale = std::ldexp(1.0f, -int(world.depth()));
        std::memcpy(s->mirror.data(), &scale, 4);
        if (!world.write_changes_to(s->mirror.data() + 4, s->mirror.size() - 5, true)) throw std::runtime_error("mirror too small");
        std::vector<vximg::Range> changed;
        for (const vx_range& r : ranges) changed.push_back(vximg::Range{r.start, r.length});
        s->image_ok = s->image->update(s->mirror.data(), world.size_in_bytes(), changed.data(), changed.size(), 2);
    };
}
}  // extern "C++"
}  // namespace

void* vxh_stream_new(int svo_type, uint32_t scene_depth, uint32_t seed, uint32_t radius, int32_t start_y, int32_t end_y) {
    if ((svo_type != 1 && svo_type != 2) || !(start_y < end_y)) return nullptr;
    return new (std::nothrow) Streamer(svo_type, scene_depth, seed, radius, start_y, end_y);
}
void vxh_stream_free(void* s) { delete static_cast<Streamer*>(s); }
// --no-lod of the reference's command line (src/main.rs:106, src/gamelogic/world.rs:141,151): before the first move
void vxh_stream_set_no_lod(void* sp, int no_lod) {
    Streamer* s = static_cast<Streamer*>(sp);
    s->esvo.no_lod = s->csvo.no_lod = no_lod != 0;
}

uint64_t vxh_stream_move_to(void* sp, float x, float y, float z) {
    Streamer* s = static_cast<Streamer*>(sp);
    return s->svo_type == 1 ? s->esvo.move_to(x, y, z) : s->csvo.move_to(x, y, z);
}

// the same with the chunk events ordered by the camera's view (frustum first, then front to back: world.rs:233-262);
// view = forward[3], up[3], fov_y_deg, aspect_ratio, near, far
uint64_t vxh_stream_move_to_view(void* sp, float x, float y, float z, const float* view) {
    Streamer* s = static_cast<Streamer*>(sp);
    graphics::Camera cam(view[6], view[7], view[8], view[9]);
    cam.position[0] = x; cam.position[1] = y; cam.position[2] = z;
    for (int k = 0; k < 3; ++k) { cam.forward[k] = view[k]; cam.up[k] = view[3 + k]; }
    return s->svo_type == 1 ? s->esvo.move_to(x, y, z, &cam) : s->csvo.move_to(x, y, z, &cam);
}

// out[11] = events, loads, unloads, lod_changes, ranges, bytes, arena_bytes, pending, build_us, apply_us, commit_us;
// returns 0 or -1 (capacity / HIP error)
// wait = 0: only events whose chunks the background workers have finished are applied (a frame loop); 1: the call waits for them
int vxh_stream_pump_mode(void* sp, void* ctx, uint32_t max_events, int wait, uint64_t* out);
int vxh_stream_pump(void* sp, void* ctx, uint32_t max_events, uint64_t* out) { return vxh_stream_pump_mode(sp, ctx, max_events, 1, out); }
int vxh_stream_pump_mode(void* sp, void* ctx, uint32_t max_events, int wait, uint64_t* out) {
    Streamer* s = static_cast<Streamer*>(sp);
    try {
        const systems::PumpStats st = s->svo_type == 1 ? s->esvo.pump(static_cast<vx_context*>(ctx), max_events, wait != 0)
                                                       : s->csvo.pump(static_cast<vx_context*>(ctx), max_events, wait != 0);
        const uint64_t v[11] = {st.events, st.loads, st.unloads, st.lod_changes, st.ranges, st.bytes, st.arena_bytes, st.pending,
                                uint64_t(st.build_ms * 1000.0), uint64_t(st.apply_ms * 1000.0), uint64_t(st.commit_ms * 1000.0)};
        std::memcpy(out, v, sizeof v);
        return 0;
    } catch (const std::exception&) {
        return -1;
    }
}

// the streamer's whole world as one frame [f32 2^-depth][header][arena] (what a full upload would send), for the oracle
size_t vxh_stream_frame(void* sp, uint8_t* dst, size_t cap) {
    Streamer* s = static_cast<Streamer*>(sp);
    auto emit = [&](auto& w, size_t header) -> size_t {
        const size_t need = 4 + header + w.size_in_bytes();
        if (!dst || cap < need) return need;
        const float scale = std::ldexp(1.0f, -int(w.depth()));
        std::memcpy(dst, &scale, 4);
        w.write_to(dst + 4);
        return need;
    };
    return s->svo_type == 1 ? emit(s->esvo.world(), 20) : emit(s->csvo.world(), 4);
}

// Tests: from now on pump(ctx = NULL) applies its dirty ranges to a host mirror of `capacity` bytes exactly like vx_commit
// applies them to the staging mirror, and keeps a traversal image (layout as vx_traversal_image) up to date incrementally.
int vxh_stream_mirror_image(void* sp, uint64_t capacity, int layout) {
    Streamer* s = static_cast<Streamer*>(sp);
    if (capacity < 64 || (layout != 0 && layout != 1)) return -1;
    s->mirror.assign(capacity, 0);
    s->image.reset(new vximg::WorldImage(s->svo_type, layout == 0 ? vximg::kEsvo48 : vximg::kOct64));
    s->image_ok = false;
    if (s->svo_type == 1) mirror_commits(s, s->esvo); else mirror_commits(s, s->csvo);
    return 0;
}
// the incrementally maintained image, in 32-bit words (0 = none / the world could not be imaged)
uint64_t vxh_stream_image(void* sp, uint32_t* dst, uint64_t cap_words) {
    Streamer* s = static_cast<Streamer*>(sp);
    if (!s->image || !s->image_ok) return 0;
    const vximg::ZeroedWords& f = s->image->frame();
    if (dst && cap_words >= f.size()) std::memcpy(dst, f.data(), f.size() * 4);
    return f.size();
}

// Tests: do two images of layout 1 ([64-byte header][octants], traversal_image.hpp) hold the same tree -- same masks, same leaf
// values, same shape -- wherever their octants were placed? 1 / 0; -1 = a pointer out of range.
int vxh_oct64_same_tree(const uint32_t* a, uint64_t na, const uint32_t* b, uint64_t nb) { return vximg::oct64_same_tree(a, na, b, nb); }

// world block position -> SVO position of the streamer's current coordinate space
void vxh_stream_to_svo(void* sp, const float* world_pos, float* svo_pos) {
    Streamer* s = static_cast<Streamer*>(sp);
    const systems::SvoCoordSpace& cs = s->svo_type == 1 ? s->esvo.coord_space() : s->csvo.coord_space();
    const Vec3 p = cs.cnv_block_pos(Vec3{world_pos[0], world_pos[1], world_pos[2]});
    svo_pos[0] = p.x; svo_pos[1] = p.y; svo_pos[2] = p.z;
}

uint64_t vxh_stream_resident_chunks(void* sp) {
    Streamer* s = static_cast<Streamer*>(sp);
    return s->svo_type == 1 ? s->esvo.resident_chunks() : s->csvo.resident_chunks();
}

// ---- graphics::Svo end to end (src/graphics/svo.rs:342-449), needs a GPU -----------------------------------------------

// The reference's `render` test through the C++ mirror: registry from PNG files, one uncompacted chunk, Svo::new(.., 10 MB),
// update, render 640x490, as_image, diff_images against the expected PNG. Returns 0 and the diff fraction, or -1 (see `err`).
int vxh_reference_render_test(int svo_type, const char* texture_dir, const char* expected_png, const char* actual_png_out, double* diff, char* err,
                              size_t err_len) {
    try {
        const std::string dir = texture_dir;
        VoxelRegistry reg;
        reg.add_texture("stone", dir + "/stone.png").add_texture("stone_normal", dir + "/stone_n.png").add_texture("dirt", dir + "/dirt.png")
            .add_texture("dirt_normal", dir + "/dirt_n.png").add_texture("grass_side", dir + "/grass_side.png")
            .add_texture("grass_side_normal", dir + "/grass_side_n.png").add_texture("grass_top", dir + "/grass_top.png")
            .add_texture("grass_top_normal", dir + "/grass_top_n.png")
            .add_material(0, Material())
            .add_material(1, Material().specular(70.0f, 0.4f).all_sides("stone").with_normals())
            .add_material(2, Material().specular(14.0f, 0.4f).top("grass_top").side("grass_side").bottom("dirt").with_normals());

        Chunk chunk(ChunkPos{0, 0, 0}, 5);
        for (uint32_t x = 0; x < 5; ++x)
            for (uint32_t z = 0; z < 5; ++z) chunk.set_block(x, 0, z, 1);
        for (uint32_t z : {1u, 3u})
            for (auto xy : {std::pair<uint32_t, uint32_t>{1, 1}, {3, 1}, {1, 3}, {3, 3}}) chunk.set_block(xy.first, xy.second, z, 2);

        graphics::Svo svo(reg, svo_type == 1 ? graphics::SvoType::Esvo : graphics::SvoType::Csvo, 10);
        Esvo<EsvoSerializedChunk> esvo;
        Csvo csvo;
        if (svo_type == 1) {
            esvo.set_leaf(Position{0, 0, 0}, EsvoSerializedChunk(chunk), true);
            esvo.serialize();
            graphics::WorldSvoRef<Esvo<EsvoSerializedChunk>> ref(esvo);
            svo.update(ref);
        } else {
            csvo.set_leaf(Position{0, 0, 0}, CsvoSerializedChunk(chunk), true);
            csvo.serialize();
            graphics::WorldSvoRef<Csvo> ref(csvo);
            svo.update(ref);
        }
        const int w = 640, h = 490;
        graphics::Framebuffer fb(w, h);
        fb.clear(0, 0, 0, 1);
        graphics::RenderParams p;
        p.ambient_intensity = 0.3f;
        p.light_dir = graphics::normalize(Vec3{-1, -1, -1});
        p.cam_pos = Vec3{2.5f, 2.5f, 7.5f};
        p.cam_fwd = Vec3{0, 0, -1};
        p.cam_up = Vec3{0, 1, 0};
        p.fov_y_rad = 72.0f * 3.14159265358979323846f / 180.0f;
        p.aspect_ratio = float(w) / float(h);
        p.selected_voxel = Vec3{1, 1, 3};
        p.render_shadows = true;
        p.shadow_distance = 500.0f;
        svo.render(p, fb);
        const Image8 actual = fb.as_image();
        if (actual_png_out && *actual_png_out) png_write(actual_png_out, actual);
        Image8 expected;
        std::string e;
        if (!png_read(expected_png, expected, e)) throw std::runtime_error(e);
        *diff = graphics::diff_images(actual, expected);
        const graphics::Stats st = svo.get_stats();
        if (st.depth != 6 || st.capacity_bytes != 10u * 1000 * 1000) throw std::runtime_error("unexpected stats");
        return 0;
    } catch (const std::exception& ex) {
        if (err && err_len) std::snprintf(err, err_len, "%s", ex.what());
        return -1;
    }
}

// worldsvo::Svo end to end: chunks at world positions around `center`, raycasts in WORLD space straight down at (x, z)
// pairs from height y0; out = n x {dst, pos.y}. Exercises coordinate conversion both ways, chunk shifting and updates.
int vxh_mapper_raycast_test(int svo_type, uint32_t render_distance, const int32_t* chunk_pos, const uint32_t* floor_height, uint32_t n_chunks,
                            const int32_t centers[6], const float* xz, uint32_t n_rays, float y0, float* out_first, float* out_second, char* err,
                            size_t err_len) {
    try {
        VoxelRegistry reg;
        reg.add_material(0, Material()).add_material(1, Material());
        graphics::Svo gfx(reg, svo_type == 1 ? graphics::SvoType::Esvo : graphics::SvoType::Csvo, 64);
        auto run = [&](auto& mapper) {
            for (uint32_t i = 0; i < n_chunks; ++i) {
                Chunk c(ChunkPos{chunk_pos[3 * i], chunk_pos[3 * i + 1], chunk_pos[3 * i + 2]}, 5);
                c.fill_with([&](uint32_t, uint32_t y, uint32_t) -> std::optional<BlockId> { return y < floor_height[i] ? std::optional<BlockId>(1u) : std::nullopt; });
                mapper.set_chunk(c);
            }
            for (int pass = 0; pass < 2; ++pass) {
                mapper.update(ChunkPos{centers[3 * pass], centers[3 * pass + 1], centers[3 * pass + 2]});
                PickerBatch batch;
                for (uint32_t i = 0; i < n_rays; ++i) batch.add_ray(Vec3{xz[2 * i], y0, xz[2 * i + 1]}, Vec3{0, -1, 0}, -1.0f);
                PickerBatchResult res;
                mapper.raycast(batch, res);
                float* out = pass == 0 ? out_first : out_second;
                for (uint32_t i = 0; i < n_rays; ++i) {
                    out[2 * i] = res.rays[i].dst;
                    out[2 * i + 1] = res.rays[i].pos.y;
                }
            }
        };
        if (svo_type == 1) {
            systems::Svo<Esvo<EsvoSerializedChunk>, EsvoSerializedChunk> mapper(gfx, render_distance);
            run(mapper);
        } else {
            systems::Svo<Csvo, CsvoSerializedChunk> mapper(gfx, render_distance);
            run(mapper);
        }
        return 0;
    } catch (const std::exception& ex) {
        if (err && err_len) std::snprintf(err, err_len, "%s", ex.what());
        return -1;
    }
}

}  // extern "C"
