/* voxel_hip.h -- C ABI of libvoxelhip.so: the MI355X (gfx950) drop-in for voxel-rs's `graphics::Svo`.
 *
 * Every entry point replaces one piece of the reference's OpenGL render/raycast surface
 * (/root/reference/src/graphics/svo.rs); the citation next to each declaration names it. All positions are in
 * SVO space [0, 2^depth) exactly as `graphics::Svo` expects (svo.rs:55); world<->SVO conversion stays with the
 * caller (src/systems/worldsvo.rs:397-435). Single caller thread per context, like the reference (svo.rs:56-73
 * holds RefCell'd GL state). No exceptions cross this boundary: functions return a vx_status, details via
 * vx_last_error(). Plain pointers and sizes only.
 *
 * The reference-side binding (Rust `extern "C"` block + safe wrapper) is shown in INTEGRATION.md.
 */
#ifndef VOXEL_HIP_H
#define VOXEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SvoType (svo.rs:18-39): the values are the reference's shader defines SVO_TYPE_ESVO / SVO_TYPE_CSVO
 * (assets/shaders/svo.glsl:65-66). */
#define VX_SVO_ESVO 1
#define VX_SVO_CSVO 2

typedef enum vx_status {
    VX_OK = 0,
    VX_ERR_INVALID_ARGUMENT = 1,
    VX_ERR_NO_DEVICE = 2,      /* no usable HIP device: there is NO CPU fallback */
    VX_ERR_OUT_OF_MEMORY = 3,
    VX_ERR_CAPACITY = 4,       /* a range does not fit the world buffer (reference: assert!, esvo.rs:328 / csvo.rs:301) */
    VX_ERR_HIP = 5,            /* a HIP runtime call failed */
    VX_ERR_STATE = 6           /* e.g. render before any commit */
} vx_status;

typedef enum vx_memory { VX_MEM_HOST = 0, VX_MEM_DEVICE = 1 } vx_memory;
/* Pixel format of a render target. RGBA32F is the reference's framebuffer (the image2D of world.glsl:10, row 0 = bottom); RGBA8 is
 * what Framebuffer::as_image reads back from it (src/graphics/framebuffer.rs:97-111): glReadPixels(RGBA, UNSIGNED_BYTE) -- clamp
 * to [0,1], round to the nearest of 255 steps, NaN -> 0 -- flipped vertically, so a whole image has its TOP row first. A quarter
 * of the bytes for a presenting embedder's read-back and for the multi-GPU gather. */
typedef enum vx_format { VX_FORMAT_RGBA32F = 0, VX_FORMAT_RGBA8 = 1 } vx_format;
/* What "identical to the reference" means for a pixel. Everything a ray's PATH depends on -- every traversal float (t, position, uv,
 * the per-iteration t_min), hit identity, flags, step counts, and the shadow ray's origin and with it the normal-mapped normal -- is
 * bit for bit the reference's arithmetic (fp32, no contraction, explicit fma where the shader says so, IEEE division and square root).
 * The COLOUR of a pixel agrees with the CPU restatement of the shaders to 5e-6 absolute, for four stated reasons, all colour-only:
 *   - pow() of the specular term is exp2(y * log2(x)) on the hardware's 1-ulp v_exp_f32 / v_log_f32 (GLSL's own definition) instead of a
 *     correctly rounded powf: within 1.2e-7 absolute on its domain;
 *   - acos() of the sky gradient is a degree-7 polynomial (Abramowitz & Stegun 4.4.46): 4.3e-7 rad;
 *   - acos' argument is clamped to [-1, 1] (world.glsl:98 does not; the reference's expected image shows no undefined horizon pixels).
 *     The restatement clamps too, so only the comparison with the reference's own PNG (tests/test_render_png.py, 1e-3) can see this one;
 *   - the four texels of a bilinear tap are blended in byte units and the common 1/255 applied once (< 4e-7; > 0 in exactly the same
 *     cases, so the alpha test that decides a hit is unaffected). */

typedef struct vx_context vx_context; /* replaces `struct Svo` (svo.rs:56-73): owns every device object */

/* MaterialInstance, 32-byte rows indexed by BlockId (src/graphics/svo_registry.rs:29-40; svo.glsl:48-59).
 * Texture fields are array layers, -1 = none. */
typedef struct vx_material {
    float specular_pow, specular_strength;
    int32_t tex_top, tex_side, tex_bottom;
    int32_t tex_top_normal, tex_side_normal, tex_bottom_normal;
} vx_material;

/* One dirty range of the serialized SVO arena, in the coordinates of the reference's
 * `RangeBuffer::updated_ranges` (src/world/hds/internal.rs:151-154,166): byte offsets relative to the
 * first byte AFTER the writer's header (ESVO: 20-byte preamble, esvo.rs:179-188; CSVO: 4-byte root pointer,
 * csvo.rs:291-292). */
typedef struct vx_range {
    uint64_t start, length;
} vx_range;

/* The uniforms `Svo::render` sets on world.glsl (svo.rs:201-215; assets/shaders/world.glsl:12-25).
 * `view` is u_view = look_to_rh(cam_pos, cam_fwd, cam_up)^-1, column-major (svo.rs:197).
 * highlight_pos = NaN when nothing is selected (svo.rs:211). light_dir must be normalised by the caller
 * (src/gamelogic/world.rs:106). */
typedef struct vx_uniforms {
    float view[16];
    float fovy, aspect;
    float ambient;
    float light_dir[3];
    float cam_pos[3];
    int32_t render_shadows;
    float shadow_distance;
    float highlight_pos[3];
} vx_uniforms;

/* PickerTask / PickerResult with the std430 layout of assets/shaders/picker.glsl:9-27
 * (src/graphics/svo_picker.rs:13-32): 48 bytes each, vec3s at 16 and 32. */
typedef struct vx_picker_task {
    float max_dst, _pad0[3];
    float pos[3], _pad1;
    float dir[3], _pad2;
} vx_picker_task;

typedef struct vx_picker_result {
    float dst;             /* -1 = no hit */
    uint32_t inside_voxel; /* GLSL bool */
    float _pad0[2];
    float pos[3], _pad1;
    float normal[3], _pad2;
} vx_picker_result;

/* Optional per-pixel record of what trace_ray saw (world.glsl:27-90) -- the "hit position, depth" outputs
 * used for parity checks; not part of the reference's surface. */
typedef struct vx_hit {
    float t;          /* primary hit distance in SVO units, -1 = miss */
    uint32_t value;   /* BlockId of the hit voxel */
    int32_t face_id;  /* 0..5 = -x,+x,-y,+y,-z,+z */
    uint32_t flags;   /* bit0 hit, bit1 shadow ray cast, bit2 in shadow, bit3 highlight outline */
    float pos[3];
    float lod;
    float uv[2];
    float shadow_t;   /* hit distance of the shadow ray, -1 = unoccluded / not cast */
    uint32_t steps;   /* traversal loop iterations, primary + shadow */
} vx_hit;

/* OctreeResult and StackFrame of the debug harness (assets/shaders/svo.glsl:31-40, svo.test.glsl:13-33). */
typedef struct vx_result {
    float t;
    uint32_t value;
    int32_t face_id;
    float pos[3];
    float uv[2];
    float color[4];
    float lod;
    int32_t inside_voxel;
} vx_result;

typedef struct vx_frame {
    float t_min;
    uint32_t ptr, idx, parent_octant_idx; /* CSVO: 4th field = depth (svo.csvo.glsl:285) */
    int32_t scale, is_child, is_leaf, crossed_boundary;
    uint32_t next_ptr;
} vx_frame;

/* graphics::svo::Stats (svo.rs:75-83) */
typedef struct vx_stats {
    uint64_t used_bytes, capacity_bytes;
    uint32_t depth;
} vx_stats;

/* Step counters of the instrumented kernel variant (algorithmic-bytes model). */
typedef struct vx_counters {
    uint64_t rays, iterations, pushes, leaf_tests, leaf_tests_trilinear, boundaries;
    uint64_t csvo_header_bytes, csvo_pointer_bytes;
    uint64_t pixels, lit_pixels, shadow_rays;
    /* occupancy of the persistent wavefront kernel: trips of its traversal loop summed over waves (iterations / (64 *
     * wave_steps) = fraction of lanes traversing), service phases run, queue refills (0 for the per-pixel kernel) */
    uint64_t wave_steps, services, refills;
    /* the same after a wave found the queue empty (the frame's tail): loop trips and lane-iterations */
    uint64_t tail_wave_steps, tail_iterations;
} vx_counters;

/* Where vx_render writes. Tiles are 32x32 pixels; the image's tiles are taken in MORTON order of their (x, y) (vx_tile_order: any
 * run of consecutive places is a compact patch of the screen) and shared out round-robin: a context renders the tiles at the places
 * j with j % tile_count == tile_rank (multi-GPU screen sharding; 0/1 = whole image).
 *   tile_count <= 1: the target is width*height pixels -- RGBA32F: row 0 = bottom, the image2D of world.glsl:10; RGBA8: row 0 = top.
 *   tile_count  > 1: the target is a compact tile list: local tile k (the tile at place k*tile_count + tile_rank) occupies pixels
 *                    [k*1024, (k+1)*1024), pixel (x,y) of the tile (y counted from the bottom) at y*32+x.
 * hits (optional) uses the same indexing with vx_hit elements. */
typedef struct vx_target {
    void* rgba32f;  /* the pixels, in `format` */
    vx_hit* hits;
    int32_t memory; /* vx_memory of both pointers */
    uint32_t tile_rank, tile_count;
    int32_t format; /* vx_format; 0 = RGBA32F */
} vx_target;

/* ---- lifetime ------------------------------------------------------------------------------------------ */

/* Svo::new (svo.rs:109-149): capacity_bytes = size_mb * 1000 * 1000 of world buffer (svo.rs:133). Allocates
 * the pinned staging mirror, the device world buffer, streams and events on HIP device `device`. */
int vx_create(int svo_type, size_t capacity_bytes, int device, vx_context** out);
/* Drop for Svo (src/graphics/buffer.rs:41-47,95-101) */
void vx_destroy(vx_context* ctx);

/* ---- resources ------------------------------------------------------------------------------------------ */

/* VoxelRegistry::build_material_buffer (svo_registry.rs:135-165): rows indexed by BlockId. */
int vx_set_materials(vx_context* ctx, const vx_material* rows, uint32_t count);
/* TextureArrayBuilder::build + TextureArray::new (src/graphics/texture_array.rs:83-153,191-236): `rgba8` is
 * layers*height*width*4 bytes, base level only, ALREADY flipped vertically (row 0 = bottom, :92,126);
 * mip_levels is clamped to min(mip_levels, ilog2(min(w,h))) like :108 and the chain is generated here
 * (2x2 box filter, the glGenerateMipmap of :259). Sampler state is fixed to the reference's (:200-203). */
int vx_set_textures(vx_context* ctx, const uint8_t* rgba8, uint32_t width, uint32_t height, uint32_t layers, uint32_t mip_levels);

/* ---- SVO upload ------------------------------------------------------------------------------------------ */

/* MappedBuffer::cast / offset (svo.rs:175,181): host-pinned mirror of the world buffer, capacity_bytes long,
 * valid until vx_destroy. The caller's `WorldSvo::write_changes_to(ptr + 4, vx_arena_capacity(ctx), reset)` writes here
 * exactly as it writes into the mapped SSBO. */
uint8_t* vx_staging_ptr(vx_context* ctx);
size_t vx_capacity(const vx_context* ctx);
/* Bytes the serialized arena may take: vx_capacity() minus the 4-byte scale and the writer's header (ESVO: 20-byte preamble,
 * CSVO: 4-byte root pointer). This is the exact `dst_len` for write_changes_to(ptr + 4, dst_len, reset): its check
 * `start + length < dst_len` (esvo.rs:328, csvo.rs:301) is relative to the first byte AFTER that header. (The reference itself
 * passes `len - 1` (svo.rs:180), which lets a nearly full world run up to 22 bytes past the mapped buffer; the staging mirror
 * is allocated 64 bytes larger than vx_capacity() so that this call pattern cannot write outside it, and vx_commit rejects
 * any range or used_bytes beyond vx_arena_capacity() with VX_ERR_CAPACITY.) */
size_t vx_arena_capacity(const vx_context* ctx);
/* Svo::update (svo.rs:171-189): stores f32 2^-depth at byte 0 (:173-175), waits for in-flight renders like
 * render_fence.wait() (:178), then copies the writer's header and the given dirty arena ranges to the device
 * (asynchronously; later renders/raycasts are ordered after it). used_bytes = WorldSvo::size_in_bytes() for
 * vx_get_stats (:183-187).
 * A context also keeps a traversal image of the world (vx_traversal_image): the chunks inside the given ranges and the root
 * octree are re-laid out as octants of one 8-byte entry per existing child on host worker threads and the changed parts uploaded;
 * vx_render walks the image (device memory on top of the world's own bytes: 0.37x them for ESVO, about 2.3x for CSVO;
 * VX_TRAVERSAL_IMAGE=0 in the environment turns it off). The staging mirror must hold the whole current world, i.e. every change has to go through
 * vx_staging_ptr (it does when write_changes_to is the only writer). */
int vx_commit(vx_context* ctx, uint32_t depth, const vx_range* ranges, uint32_t count, uint64_t used_bytes);
/* Same, treating [0, used_bytes) of the arena as dirty (what the first write_changes_to after write_to does). */
int vx_commit_all(vx_context* ctx, uint32_t depth, uint64_t used_bytes);
/* Pipelined commits. VX_COMMIT_INLINE (default): vx_commit does the image update and queues the uploads before it returns, and
 * every later render sees the new world. VX_COMMIT_PIPELINED: vx_commit only posts the job to a worker thread of the context
 * (0.01 ms) and returns; the worker updates the image, packs and queues the uploads while the caller goes on -- renders issued
 * meanwhile show the world as of the commit before, WHOLLY (world bytes and image change together, ordered on the
 * device behind the frames in flight and before every later one), so a change becomes visible at most one frame late, like the
 * reference's own one-frame lag between Svo::update and the next fence (svo.rs:171-189, 196-229). One job at a time: the next
 * vx_commit, vx_staging_ptr (the worker reads the mirror: ask for the pointer before writing the next changes, as
 * write_changes_to's callers do), vx_commit_wait and vx_sync wait for the posted one. A pipelined commit's error (VX_ERR_HIP
 * etc.) is reported by the next vx_commit, vx_commit_wait or vx_sync. The first commit of a context is always done inline. */
typedef enum vx_commit_mode { VX_COMMIT_INLINE = 0, VX_COMMIT_PIPELINED = 1 } vx_commit_mode;
int vx_set_commit_mode(vx_context* ctx, int mode);
/* wait until the posted commit has been queued on the device (not for the device itself: vx_sync), return its error */
int vx_commit_wait(vx_context* ctx);
/* Svo::get_stats (svo.rs:191-193) */
int vx_get_stats(const vx_context* ctx, vx_stats* out);

/* ---- the hot path ---------------------------------------------------------------------------------------- */

/* Svo::render (svo.rs:196-229) = world.glsl main for every pixel: primary ray, shading, <=1 shadow ray, sky.
 * Device targets are written asynchronously on the context's stream (vx_sync = the render fence); host
 * targets return when the image is in place.
 * Scheduling only, never a pixel's value: image-only renders use what earlier frames of the same view on the same stream cost -- the
 * order in which work is handed out, and (a view whose uniforms are bit for bit the last frame's) which pixels a wave's lanes take
 * together -- so a view that stands still renders faster from its third frame on; a moving view is not affected. */
int vx_render(vx_context* ctx, const vx_uniforms* uniforms, uint32_t width, uint32_t height, const vx_target* target);
/* Svo::raycast (svo.rs:233-255) = picker.glsl over `count` tasks (no 100-task cap); synchronous like the
 * reference's fence wait (:248-249). Host pointers. */
int vx_raycast(vx_context* ctx, const vx_picker_task* tasks, uint32_t count, vx_picker_result* results);
/* svo.test.glsl (assets/shaders/svo.test.glsl:63-76): one ray with a StackFrame per loop iteration.
 * n_frames receives the number of iterations (may exceed max_frames). Host pointers. */
int vx_debug_trace(vx_context* ctx, const float pos[3], const float dir[3], float max_dst, int cast_translucent, vx_result* result,
                   vx_frame* frames, uint32_t max_frames, uint32_t* n_frames);
/* Fence::wait for everything enqueued on this context (src/graphics/fence.rs:8-42). */
int vx_sync(vx_context* ctx);
/* Frames in flight. vx_render of an image (no hit records) into DEVICE memory returns after enqueueing and consecutive
 * frames rotate over a few internal streams (two by default), so that frame k+1 starts on the compute units frame k's last rays leave
 * idle. Two calls order such a render against the caller's own streams:
 *   vx_wait_event(ctx, e)           the NEXT vx_render waits for the hipEvent_t `e` (e.g. "the buffer I am about to
 *                                   render into has been consumed"); one-shot
 *   vx_stream_wait_render(ctx, s)   the caller's hipStream_t `s` waits for the most recently issued vx_render
 * vx_sync waits for everything; vx_commit orders uploads after every frame in flight (Svo::update's fence, svo.rs:178). */
/* 1 = every render on the context's stream again; 2 (default) .. 8 = that many frame streams in rotation. More frames in
 * flight hide more of each frame's tail, which matters when a context renders only a share of the tiles (multi-GPU:
 * an eighth of a 1080p frame takes 0.24 ms per frame with 1, 0.085 ms with 3, 0.068 ms with 4 or more frames in flight). */
int vx_set_frames_in_flight(vx_context* ctx, int frames);
int vx_wait_event(vx_context* ctx, void* hip_event);
int vx_stream_wait_render(vx_context* ctx, void* stream);

/* ---- pipelined presentation --------------------------------------------------------------------------------- */

/* For an embedder that shows every frame (the reference blits its framebuffer, src/gamelogic/world.rs:269-283): vx_present_begin
 * enqueues the frame on one of the frame streams and its read-back into pinned host memory on a copy stream behind it, and
 * returns a slot (0..3, in rotation); vx_present_wait blocks until that slot's image is in place and returns it (valid until the
 * slot comes round again, i.e. for the next three vx_present_begin calls). Begin frame k+1, then wait for frame k: the read-back
 * of one frame runs beside the kernel of the next. */
int vx_present_begin(vx_context* ctx, const vx_uniforms* uniforms, uint32_t width, uint32_t height, int format, int* out_slot);
int vx_present_wait(vx_context* ctx, int slot, const void** pixels, size_t* bytes);

/* ---- multi-GPU: tiles, gather, assembly ------------------------------------------------------------------------- */

/* The shared sequence of an image's tiles: out[j] = row-major id (ty * tiles_x + tx, from the bottom-left) of the tile at place j
 * of the Morton order. Returns the number of tiles; fills `out` when capacity suffices. Pure host function. */
uint32_t vx_tile_order(uint32_t width, uint32_t height, uint32_t* out, uint32_t capacity);

/* Which RCCL to open: by default the library is opened by its soname (librccl.so.1: the copy the process already has loaded, if any) when the
 * first communicator is asked for. A deployment that ships its own build -- and the tests, whose stand-in lets several ranks share one GPU
 * (tests/stub_rccl) -- names the file here, before the first vx_comm_* call. Process-wide. No counterpart in the reference (one GL context). */
int vx_comm_library(const char* path);
/* The handle owns the RCCL communicator the finished tiles travel over (SURVEY.md 8b "Ownership"; one process per GPU):
 *   vx_comm_unique_id   on ONE rank: a fresh 128-byte id (ncclGetUniqueId), which the caller hands to every rank by its own means
 *   vx_comm_init        on every rank, collectively: ncclCommInitRank on this context's device
 *   vx_gather_tiles     the one exchange step of the path: this rank's compact tile list (`bytes_per_rank` bytes of device
 *                       memory, the same on every rank) goes to `root`, which receives rank r's list at gathered + r *
 *                       bytes_per_rank (the root's own list is copied there unless it already is there:
 *                       tiles == gathered + root * bytes_per_rank) -- grouped ncclSend / ncclRecv, every peer straight to the root over its own xGMI link, on
 *                       the communicator's stream and ordered behind every vx_render issued so far (the list's among them). Returns
 *                       after enqueueing; *out_ticket (optional) names the gather for vx_wait_gather
 *   vx_wait_gather      the NEXT vx_render waits (on the device) for that gather: call it before rendering into a tile list a
 *                       gather may still be reading
 *   vx_comm_stream      the communicator's hipStream_t: vx_assemble_tiles_on(.., vx_comm_stream(ctx)) is ordered behind the gather
 * RCCL is opened at run time (dlopen of librccl.so.1) by the first of these calls; a single-GPU deployment needs none. */
#define VX_COMM_ID_BYTES 128
int vx_comm_unique_id(void* out_id, size_t bytes);
int vx_comm_init(vx_context* ctx, int nranks, int rank, const void* unique_id);
int vx_comm_destroy(vx_context* ctx);
int vx_comm_info(const vx_context* ctx, int* nranks, int* rank);
/* Wave slots per CU that a context with a communicator of more than one rank leaves to RCCL's own workgroups (its persistent render waves would
 * otherwise fill every CU's LDS, and a communication kernel would find room only when a whole frame has drained). 0 .. 8; default 4 (or
 * VX_COMM_HEADROOM at vx_create). Read by the next render: a caller can time a few frames at each setting and keep the best (bench.py does,
 * at world_size > 1). No counterpart in the reference (one GL context, src/graphics/svo.rs:196-229). */
int vx_set_comm_headroom(vx_context* ctx, int waves_per_cu);
int vx_gather_tiles(vx_context* ctx, const void* tiles, uint64_t bytes_per_rank, void* gathered, int root, int* out_ticket);
int vx_wait_gather(vx_context* ctx, int ticket);
/* Has that gather (and an assembly issued behind it on the communicator's stream: see vx_assemble_tiles_format) finished? 1 = yes,
 * 0 = not yet, -1 = bad ticket or a device error. Never blocks: a caller that must not hang on a collective a peer never joins polls
 * this against a deadline instead of calling vx_sync. */
int vx_gather_query(vx_context* ctx, int ticket);
void* vx_comm_stream(vx_context* ctx);
/* One sharded frame in ONE call -- what a rank does per frame, in the order it has to be done: vx_wait_gather(wait_ticket) unless wait_ticket < 0 (the
 * exchange that last read this tile list), vx_render of this rank's tiles into `target` (device memory; tile_rank / tile_count / format as for
 * vx_render), vx_gather_tiles of the list (bytes_per_rank of it) to `root`, and, on the root, vx_assemble_tiles_format of the gathered lists
 * (`gathered`: [ranks][bytes_per_rank]; the root's own list should be target->rgba32f = gathered + root * bytes_per_rank: rendered in place)
 * into `image` on the communicator's stream (image NULL: no assembly). *out_ticket names the exchange (and covers the assembly). Saves the
 * caller three of four trips through the ABI: at eight ranks a rank's share of a 1080p frame is 0.03 ms of GPU time, and a frame loop
 * that spends longer than that per frame on calls is the bound (bench.py's host_issue_ms_per_step). The reference has no counterpart. */
int vx_render_gather(vx_context* ctx, const vx_uniforms* uniforms, uint32_t width, uint32_t height, const vx_target* target, uint64_t bytes_per_rank,
                     void* gathered, int root, void* image, int wait_ticket, int* out_ticket);
/* vx_assemble_tiles for either pixel format: stride in PIXELS between the ranks' lists; an RGBA8 image comes out top row first.
 * Issued on vx_comm_stream(ctx), it extends the ticket of every gather issued since the last assembly on that stream: vx_wait_gather(ticket)
 * then also waits for this kernel, which reads every rank's list -- the root's own included, which the root renders into in place. Lists and
 * image must be aligned to a pixel (16 bytes RGBA32F, 4 bytes RGBA8). */
int vx_assemble_tiles_format(vx_context* ctx, const void* tiles, uint64_t stride_pixels, uint32_t tile_count, uint32_t width, uint32_t height, void* out,
                             int format, void* stream);

/* Scatters `tile_count` gathered compact tile lists (rank r's list at tiles + r*stride_floats) into a
 * width*height RGBA32F image; all pointers are device memory on this context's device. */
int vx_assemble_tiles(vx_context* ctx, const float* tiles, uint64_t stride_floats, uint32_t tile_count, uint32_t width, uint32_t height,
                      float* out_rgba32f);
/* The same on a caller-supplied hipStream_t (e.g. the stream the gather ran on, so that the context's own stream is free to
 * render the next frame meanwhile); NULL = the legacy default stream. */
int vx_assemble_tiles_on(vx_context* ctx, const float* tiles, uint64_t stride_floats, uint32_t tile_count, uint32_t width, uint32_t height,
                         float* out_rgba32f, void* stream);
/* The traversal image of a world: the octant tree a context traverses instead of the world's own bytes.
 * `world_frame` = the world as committed ([f32 scale][ESVO: 5-word preamble | CSVO: u32 root_ptr][arena]), `used_bytes` = arena
 * bytes in use. layout 1 = what the renderer walks ([64-byte header][octants: a {pointer | value, masks} entry per EXISTING child, child 7
 * first, or just the existing children's values when every child is a voxel -- behind a unit that says where the node lies in a CSVO world's bytes], everything
 * addressed in 8-byte units); layout 2 = the same bytes, walked through a 64-bit pointer instead of a buffer resource: what the renderer switches to when
 * the image outgrows 4 GiB; layout 0 = the same tree as an ESVO frame ([f32 2^-depth][5-word preamble][12-word octants], esvo.rs:74-101),
 * which any ESVO traversal can walk (the tests do, with the oracle). Returns the image size in 32-bit words (0 = cannot be
 * imaged) and fills `out_words` when it is large enough. Pure host function (no device needed): vx_commit does this itself. */
uint64_t vx_traversal_image(int svo_type, const uint8_t* world_frame, uint64_t used_bytes, int layout, uint32_t* out_words, uint64_t capacity_words);
/* Rounds 3-5 kept an ORIGIN TABLE beside the image of a CSVO world and this call returned it. Since round 6 the origin of a voxel-parent octant -- where
 * that node (a leaf-mask byte, svo.csvo.glsl:114-115) lies in the world's own bytes: [0] = its byte pointer, [1] = k << 29 | (pointer - the chunk's material
 * section), k = its place among its depth-2 parent's leaf-mask bytes -- is the 8-byte unit of the image in front of the octant's values, and this call is
 * vx_traversal_image (nothing is written to `out_origin_words`). The renderer consults the origin when a ray that started inside a voxel is led into it
 * (svo.csvo.glsl:293-295): that walk is format specific and is made on the world's bytes. */
uint64_t vx_traversal_image_with_origin(int svo_type, const uint8_t* world_frame, uint64_t used_bytes, int layout, uint32_t* out_words,
                                        uint64_t capacity_words, uint32_t* out_origin_words, uint64_t origin_capacity_words);
/* 2x2 ordered-grid supersampling (BASELINE.json C5): box-filters a (2*width) x (2*height) RGBA32F render down to
 * width x height, both in device memory, on the caller's hipStream_t (NULL = legacy default stream). Render the large
 * image with vx_render first (order it with vx_stream_wait_render). */
int vx_resolve_2x2(vx_context* ctx, const float* src_rgba32f, uint32_t width, uint32_t height, float* dst_rgba32f, void* stream);
/* Number of tiles (32x32) rank `tile_rank` of `tile_count` owns for a width x height image. */
uint32_t vx_local_tile_count(uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_count);

/* ---- measurement ------------------------------------------------------------------------------------------ */

/* Instrumented variant of vx_render (same rays, per-phase step counters, no image kept). */
int vx_render_counters(vx_context* ctx, const vx_uniforms* uniforms, uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_count,
                       vx_counters* out);
/* While enabled, every render-kernel launch is bracketed by HIP events on the stream it is launched on. */
int vx_profile_enable(vx_context* ctx, int enabled);
/* Sum of the bracketed kernel durations (ms) and their count since the last call; synchronises. */
int vx_profile_read(vx_context* ctx, double* kernel_ms_sum, uint32_t* launches);
/* The shader clock the device runs at while this call lasts, in MHz: a one-lane kernel on a stream of its own, beside whatever the context
 * has in flight, compares the shader-clock counter with the device's constant 100 MHz counter over `microseconds` (10 .. 100000). Blocks the
 * calling thread until the probe has run; may be called from a second thread while the context's own thread renders (it takes the
 * context's lock only to launch). No counterpart in the reference: bench.py samples it during its sustained block. */
int vx_clock_probe(vx_context* ctx, uint32_t microseconds, double* shader_mhz);
/* The same for the exchanges (vx_gather_tiles calls made while profiling was enabled): time on the communicator's stream from the
 * first send / receive to the last, i.e. including the wait for the slowest peer. Synchronises the communicator's stream. */
int vx_comm_profile_read(vx_context* ctx, double* gather_ms_sum, uint32_t* gathers);
/* Measurement (contexts created with VX_TIMELINE=1 in the environment by the library's timeline build, lib/lib_tl; else returns 0): per wave of the most recent render launch
 * EIGHT words -- [0] when it started, [1] when it found the sub-tile queue empty, [2] when it left (all in 10 ns ticks of the device's
 * constant clock), [3] sub-tiles taken | service phases << 20 | ticks spent in them << 32, [4] its life in shader-clock cycles
 * (s_memtime: [4] / ([2] - [0]) x 100 MHz is the clock the kernel ran at), [5] the cycles of it spent in the traversal loop, [6] the
 * loop's trips | those in which every traversing lane ADVANCEd << 20 | those in which every one PUSHed << 40 (20 bits each), [7] the walks inside voxels (images of CSVO worlds): walk phases << 52 | trips of the walk's loop << 32 | shader-clock cycles
 * in the walks. `out` holds capacity_waves x 8 words. Returns the number of waves copied. Waits for every frame in flight. */
uint32_t vx_timeline_read(vx_context* ctx, uint64_t* out, uint32_t capacity_waves);
/* What vx_render walks after the last commit: [0] 0 = the world's own bytes (no image: switched off, or the world cannot be
 * imaged), 1 = the traversal image walked through a buffer resource, 2 = walked through a 64-bit pointer (images beyond 4 GiB);
 * [1] bytes of the image, [2] 0 (rounds 3-5: bytes of a CSVO world's origin table; the origins are units of the image now), [3] chunks it holds. */
int vx_image_info(const vx_context* ctx, uint64_t out[4]);
/* The scheduling knobs this context runs with (they reorder a frame's work and change no pixel): [0] refill threshold, [1] service threshold, [2] cap on
 * the persistent waves per CU (0 = none), [3] length of the sub-tile queue's stretches (0 = by the launch), [4] width of the tile numbering's strips, [5] cost-ordered
 * queue on / off, [6] wave slots per CU left to RCCL (vx_set_comm_headroom), [7] 1 = this is the library's measurement build (the only one that reads the
 * VX_REFILL_MIN / VX_WAVES_PER_CU / VX_QUEUE_STRIPE / VX_TILE_STRIP / VX_HOT_FIRST environment variables). For the tests. */
int vx_debug_knobs(const vx_context* ctx, uint32_t out[8]);
/* CSVO worlds rendered from their traversal image: since creation (or the last reset), [0] rays that were led into the voxel they
 * started in and made that walk on the world's own bytes, [1] those of them whose pixel was rendered again on the bytes (the walk
 * overwrote cursor state the rest of the ray depends on), [2] service phases of the render kernel that ran such walks (the rays of a
 * wave go together), [3] loop iterations made on the world's own bytes. Waits for every frame in flight. Counting costs frame time (atomics
 * on one line from every wave) and is OFF until asked for: reset = 1 zeroes the counters and counts from now on, 2 zeroes them and stops
 * counting, 0 only reads. */
int vx_excursion_counters(vx_context* ctx, uint64_t out[4], int reset);
/* hipStream_t the context launches on (as void*), for callers that order their own work after it. */
void* vx_stream(vx_context* ctx);
int vx_device(const vx_context* ctx);

const char* vx_last_error(void);
const char* vx_version(void);

#ifdef __cplusplus
}
#endif
#endif
