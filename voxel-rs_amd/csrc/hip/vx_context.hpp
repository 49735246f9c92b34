// Internal to the library's host side (runtime.cpp, comm.cpp): the render context behind the opaque vx_context of include/voxel_hip.h.
#pragma once

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>  // types only: RCCL itself is opened at run time by vx_comm_init (a single-GPU deployment needs none)

#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "kernels.h"
#include "traversal_image.hpp"
#include "voxel_hip.h"
#include "vx_args.hpp"

namespace vxrt {

// the calling thread's last error message (vx_last_error); fail() sets it and returns the code
extern thread_local std::string g_last_error;
int fail(int code, const std::string& msg);

#define HIP_TRY(call)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (call);                                                                                    \
        if (e_ != hipSuccess) return vxrt::fail(e_ == hipErrorOutOfMemory ? VX_ERR_OUT_OF_MEMORY : VX_ERR_HIP,     \
                                                std::string(#call) + ": " + hipGetErrorString(e_));               \
    } while (0)

struct ProfiledLaunch {
    hipEvent_t start, stop;
};

}  // namespace vxrt

struct vx_context;
namespace vxrt {
void comm_release(vx_context* c);  // comm.cpp: the context's communicator, if it has one, is destroyed (vx_destroy)
}

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
    bool hot_first = true;            // VX_HOT_FIRST=0: sub-tiles in plain order (A/B)
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
    // small picker batches: pinned host memory the kernel reads and writes directly (vx_raycast)
    static constexpr uint32_t kPickerDirect = 2048;
    vx_picker_task* h_pick_tasks = nullptr;  vx_picker_result* h_pick_results = nullptr;
    vx_picker_task* d_pick_tasks = nullptr;  vx_picker_result* d_pick_results = nullptr;
    vx_result* d_trace_result = nullptr;  vx_frame* d_trace_frames = nullptr;  uint32_t* d_trace_count = nullptr;  uint32_t trace_cap = 0;
    unsigned long long* d_counters = nullptr;

    uint32_t* d_work_counter = nullptr;
    unsigned long long* d_excursions = nullptr;  // [4], see PersistentArgs
    bool count_excursions = false;               // vx_excursion_counters(.., 1) switches the counting on (it costs: see render_persistent)
    unsigned long long* d_timeline = nullptr;    // VX_TIMELINE=1 with the library's timeline build: [8192][8], the last launch's waves (PersistentArgs::timeline)
    uint32_t timeline_waves = 0;
    uint32_t timeline_part = 0;   // VX_TIMELINE_PART
    // the traversal image of the world (traversal_image.hpp), rebuilt for the changed chunks by every commit
    vximg::WorldImage image;
    uint8_t* d_image = nullptr;
    size_t d_image_capacity = 0;
    uint8_t* d_origin = nullptr;  // (rounds 3-5: a CSVO world's origin table beside the image; the origins are units of the image now -- never allocated)
    size_t d_origin_capacity = 0;
    size_t image_cap_bytes = 0;   // VX_IMAGE_CAP_BYTES: never allocate more than this for the image (tests of the fall-back)
    size_t image_first_bytes = size_t(32) << 20;  // VX_IMAGE_FIRST_BYTES: the least the image's device buffer is given (it doubles from there as a streamed world's image grows)
    // vx_commit's packed uploads: a small ring of pinned host buffers with their device twins, each guarded by an event
    struct DeltaSlot { uint8_t* host = nullptr; uint8_t* dev = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; };
    static constexpr int kDeltaSlots = 3;
    DeltaSlot delta[kDeltaSlots];
    unsigned delta_next = 0;
    bool hot_levels = false;      // VX_HOT_LEVELS=1 (experiment X1): the image's top two levels served from an LDS copy (ESVO worlds of at most 13 levels, image-only renders)
    int foreign_rerun = -1;       // VX_FOREIGN_RERUN=0: image-only renders of a CSVO world never list their inside-voxel rays for the bytes (kForeignRerun)
                                  // but walk them in the render loop; default: worlds of at most 12 levels list them
    bool big = false;             // an ESVO world buffer of 4 GiB and more: kernels on its own bytes use 64-bit addresses (VX_SVO_ESVO_BIG)
    bool image_enabled = true;  // VX_TRAVERSAL_IMAGE=0: traverse the world's own bytes
    bool image_ok = false;
    int kernel_version = 2;               // 2 = persistent wavefront kernel, 1 = one thread per pixel (kept for A/B runs)
    // service_min = 64: a wave's lanes move in LOCKSTEP -- a sub-tile's 64 primary rays are traversed until the last of them has ended, then
    // served together (hits shaded, misses painted), then the shadow rays, then the pixels lit and stored and the next sub-tile taken (round 3:
    // 9.2 -> 10.9 Grays/s from 63 to 64, profiles/round3/pass_m, pass_o). Smaller values (lanes served when that many wait) are kept for the tests.
    // refill_min = 64 (round 5): idle lanes are refilled when ALL of the wave's lanes are -- a sub-tile at a time, so that no batch mixes the next
    // sub-tile's primary rays with this one's shadow rays, and no service phase has to shade AND light (4 -> 64: C3 -1.1 %, 4K depth 13 -1.6 %,
    // profiles/round5/pass_a/refill.txt, tails_detail.txt). Smaller: the measurement build's VX_REFILL_MIN.
    uint32_t refill_min = 64, service_min = 64;
    int tile_strip = 8;  // VX_TILE_STRIP: tile numbering 1's strips are this many tiles wide
    int tile_numbering = 1;  // VX_TILE_NUMBERING: how a whole-image render's tile numbers lie on the screen (RenderParams::tile_numbering); vx_create: 2 for ESVO worlds
    int queue_stripe = 0;  // VX_QUEUE_STRIPE: the length of the stretches the sub-tile queue deals out to its dispensers (0: by the launch, launch_render)
    // Block ids 0..63 all of whose textures are opaque throughout (RenderParams::opaque_*): from host copies of what vx_set_materials and
    // vx_set_textures were given. opaque_layer[l] = every texel of layer l, on every mip level, has alpha > 0.
    std::vector<vx_material> host_materials;
    std::vector<uint8_t> opaque_layer;
    uint64_t opaque_blocks = 0;
    int waves_per_cu_cap = 0;             // experiment: fewer persistent waves than the occupancy limit
    int comm_headroom = 4;                // VX_COMM_HEADROOM: wave slots per CU a context with a communicator of more than one rank leaves free (LDS for RCCL's kernels)
    int cu_count = 256;
    std::unordered_map<const void*, int> persistent_blocks;  // kernel -> resident 64-thread workgroups per CU, queried once

    // screen sharding: the Morton order of an image's tiles and its inverse, on the device, per image size seen
    struct TileTable { uint32_t tiles_x = 0, tiles_y = 0; uint32_t* d_order = nullptr; uint32_t* d_inverse = nullptr; };
    std::vector<TileTable> tile_tables;
    // ... and what a launch's queue numbers mean on the screen (RenderParams::tile_table / number_of_place), per (image size, rank, numbering) seen
    struct LaunchTable { uint32_t tiles_x = 0, tiles_y = 0, rank = 0, count = 0, numbering = 0, strip = 0; uint2* d_table = nullptr; uint32_t* d_number = nullptr; };
    std::vector<LaunchTable> launch_tables;

    // multi-GPU: the RCCL communicator over which the finished tiles are gathered (vx_comm_init), its stream and the events that
    // say when a gather has read its tile list
    ncclComm_t comm = nullptr;
    int comm_ranks = 0, comm_rank = 0;
    hipStream_t comm_stream = nullptr;
    static constexpr int kGatherEvents = 16;
    hipEvent_t gather_done[kGatherEvents] = {};
    unsigned gather_index = 0;      // gathers issued so far (ticket = index % kGatherEvents)
    unsigned assembled_index = 0;   // ... of which an assembly on the communicator's stream has been issued behind
    std::vector<vxrt::ProfiledLaunch> gathers;  // vx_profile_enable: the exchanges' event pairs (vx_comm_profile_read)

    // pipelined presentation (vx_present_begin / vx_present_wait): per slot a device frame and its pinned host twin; the read-back
    // runs on its own stream behind the frame's kernel, beside the next frame's
    static constexpr int kPresentSlots = 4;
    struct PresentSlot { void* dev = nullptr; void* host = nullptr; size_t cap = 0, bytes = 0; hipEvent_t copied = nullptr; bool busy = false; };
    PresentSlot present[kPresentSlots];
    unsigned present_next = 0;
    hipStream_t copy_stream = nullptr;

    // vx_clock_probe: its stream and its two words of pinned host memory the kernel writes to
    hipStream_t probe_stream = nullptr;
    unsigned long long* h_probe = nullptr;
    std::mutex probe_mutex;  // one probe at a time

    bool profile = false;
    std::vector<vxrt::ProfiledLaunch> launches;
    std::vector<vxrt::ProfiledLaunch> event_pool;

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
