// libvoxelhip.so, multi-GPU: the render context's RCCL communicator and the gather of the finished tile lists to rank 0 (SURVEY.md 8e:
// rays shard by screen tile, the SVO is replicated per GPU, only the final image travels). RCCL is opened at run time: a single-GPU
// deployment needs none.
#include <dlfcn.h>

#include <cstring>

#include "vx_context.hpp"

using vxrt::fail;
using vxrt::ProfiledLaunch;

namespace {

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
std::string g_rccl_path;  // vx_comm_library: the library to open instead of librccl.so.1

// RCCL is opened when the first communicator is asked for. By its soname: a process that already has one loaded (PyTorch brings
// its own copy) shares that one.
int rccl_open() {
    if (g_rccl.lib) return VX_OK;
    void* h = nullptr;
    if (!g_rccl_path.empty()) {
        h = dlopen(g_rccl_path.c_str(), RTLD_NOW | RTLD_LOCAL);  // (a path: another build than the process may already have loaded by soname)
    } else {
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    }
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

namespace vxrt {
void comm_release(vx_context* c) {
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    c->comm = nullptr;
}
}  // namespace vxrt

extern "C" {

// ---- multi-GPU: the gather of the finished tiles over RCCL ---------------------------------------------------------------------

int vx_comm_library(const char* path) {
    if (!path || !*path) return fail(VX_ERR_INVALID_ARGUMENT, "comm_library: empty path");
    if (g_rccl.lib) return fail(VX_ERR_STATE, "comm_library: RCCL is already open (call this before the first vx_comm_* call)");
    g_rccl_path = path;
    return VX_OK;
}

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

int vx_set_comm_headroom(vx_context* ctx, int waves_per_cu) {
    if (!ctx || waves_per_cu < 0 || waves_per_cu > 8) return fail(VX_ERR_INVALID_ARGUMENT, "comm headroom: 0..8 waves per CU");
    VX_LOCK(ctx);
    ctx->comm_headroom = waves_per_cu;  // (launch_render reads it per launch)
    return VX_OK;
}

namespace {
// the exchange; only_slot >= 0: the list was rendered by the frame just issued on that frame stream (vx_render_gather knows) -- one event to wait for
// instead of every stream's (a HIP call each: a third of what a sharded frame costs its host thread)
int gather_tiles(vx_context* ctx, const void* tiles, uint64_t bytes_per_rank, void* gathered, int root, int* out_ticket, int only_slot);
}  // namespace

int vx_gather_tiles(vx_context* ctx, const void* tiles, uint64_t bytes_per_rank, void* gathered, int root, int* out_ticket) {
    return gather_tiles(ctx, tiles, bytes_per_rank, gathered, root, out_ticket, -1);
}

namespace {
int gather_tiles(vx_context* ctx, const void* tiles, uint64_t bytes_per_rank, void* gathered, int root, int* out_ticket, int only_slot) {
    if (!ctx || !tiles || !bytes_per_rank || (bytes_per_rank & 3)) return fail(VX_ERR_INVALID_ARGUMENT, "gather_tiles: bad argument");
    VX_LOCK(ctx);
    if (!ctx->comm) return fail(VX_ERR_STATE, "gather_tiles: no communicator (vx_comm_init)");
    if (root < 0 || root >= ctx->comm_ranks) return fail(VX_ERR_INVALID_ARGUMENT, "gather_tiles: bad root");
    if (ctx->comm_rank == root && !gathered) return fail(VX_ERR_INVALID_ARGUMENT, "gather_tiles: the root needs a destination");
    HIP_TRY(hipSetDevice(ctx->device));
    // behind the renders issued so far (the list's among them), on the communicator's own stream: the frame streams go on with the
    // next frames meanwhile
    if (only_slot >= 0 && only_slot < vx_context::kFrameStreams && ctx->frame_recorded[only_slot]) {
        HIP_TRY(hipStreamWaitEvent(ctx->comm_stream, ctx->frame_done[only_slot], 0));
    } else {
        if (ctx->render_recorded) HIP_TRY(hipStreamWaitEvent(ctx->comm_stream, ctx->render_done, 0));
        for (int i = 0; i < vx_context::kFrameStreams; ++i)  // (every frame issued so far: a list may hold a group of frames)
            if (ctx->frame_recorded[i]) HIP_TRY(hipStreamWaitEvent(ctx->comm_stream, ctx->frame_done[i], 0));
    }
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
    // (an error between the two records hands the pair back to the pool)
    auto exchange = [&]() -> int {
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
        return VX_OK;
    };
    if (const int rc = exchange()) {
        if (ctx->profile) ctx->event_pool.push_back(ev);
        return rc;
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
}  // namespace

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

int vx_render_gather(vx_context* ctx, const vx_uniforms* uniforms, uint32_t width, uint32_t height, const vx_target* target, uint64_t bytes_per_rank,
                     void* gathered, int root, void* image, int wait_ticket, int* out_ticket) {
    if (!ctx || !target || !target->rgba32f) return fail(VX_ERR_INVALID_ARGUMENT, "render_gather: null argument");
    if (target->memory != VX_MEM_DEVICE) return fail(VX_ERR_INVALID_ARGUMENT, "render_gather: the tile list lives in device memory");
    VX_LOCK(ctx);  // (one frame's four steps as one: nothing of another thread's in between)
    // every argument is checked BEFORE anything is issued: an error below this block leaves a gather in flight, and its ticket is the
    // caller's only handle on it
    if (!ctx->comm) return fail(VX_ERR_STATE, "render_gather: no communicator (vx_comm_init)");
    if (root < 0 || root >= ctx->comm_ranks) return fail(VX_ERR_INVALID_ARGUMENT, "render_gather: bad root");
    if (!bytes_per_rank || (bytes_per_rank & 3)) return fail(VX_ERR_INVALID_ARGUMENT, "render_gather: bytes_per_rank must be a positive multiple of 4");
    if (ctx->comm_rank == root && !gathered) return fail(VX_ERR_INVALID_ARGUMENT, "render_gather: the root needs a destination");
    // the list is this rank's share of a frame cut for the communicator's ranks (a one-rank communicator: the frame itself, tile_count 0 or 1)
    if ((target->tile_count > 1 || ctx->comm_ranks > 1) && target->tile_count != uint32_t(ctx->comm_ranks))
        return fail(VX_ERR_INVALID_ARGUMENT, "render_gather: target->tile_count must be the communicator's size");
    const uint64_t pixel = target->format == VX_FORMAT_RGBA8 ? 4u : 16u;
    const bool assemble = image && ctx->comm_rank == root;
    if (assemble && bytes_per_rank % pixel) return fail(VX_ERR_INVALID_ARGUMENT, "render_gather: bytes_per_rank must be whole pixels");
    if (wait_ticket >= 0)
        if (int rc = vx_wait_gather(ctx, wait_ticket)) return rc;
    if (int rc = vx_render(ctx, uniforms, width, height, target)) return rc;
    int ticket = -1;
    // (the list was written by the render just issued: behind that frame's event alone -- a render on the context's own stream: behind everything)
    if (int rc = gather_tiles(ctx, target->rgba32f, bytes_per_rank, gathered, root, &ticket, ctx->last_frame_slot)) return rc;
    if (out_ticket) *out_ticket = ticket;  // (the gather is recorded: whatever happens next, the caller can wait for it)
    if (assemble) {
        // (vx_render with tile_count 1 writes the whole frame: the "list" of a one-rank run is the frame, and the assembly runs for its cost)
        if (int rc = vx_assemble_tiles_format(ctx, gathered, bytes_per_rank / pixel, uint32_t(ctx->comm_ranks), width, height, image, target->format, ctx->comm_stream)) return rc;
    }
    return VX_OK;
}


}  // extern "C"
