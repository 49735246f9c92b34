// What the host runtime knows of the kernels: which builds of the render kernel exist and how the small kernels are launched. The kernels
// themselves live in kernels_render.hip and kernels_aux.hip.
#pragma once

#include <hip/hip_runtime_api.h>

#include "vx_args.hpp"

namespace vxk {

// One build of render_persistent (kernels_render.hip). svo: the format the rays walk -- VX_SVO_ESVO / VX_SVO_CSVO / VX_SVO_ESVO_BIG (the
// world's own bytes: one build per format, it writes hit records and counters where the pointers it is given are not null) or VX_SVO_IMAGE /
// VX_SVO_IMAGE_WIDE (the traversal image). For an image: hits = the build that also writes hit records (looser register bound); foreign =
// 0 (the image of an ESVO world), VX_SVO_CSVO (rays led into a voxel walk it on the world's bytes) or kForeignRerun (they are listed and
// run afterwards); levels = LDS-resident stack levels, 13 or 16; hot = the LDS copy of the top two levels.
struct RenderBuild {
    int svo;
    bool hits;
    int foreign;
    int levels;
    bool hot;
};
const void* render_persistent_fn(const RenderBuild& b);  // null: there is no such build
size_t render_persistent_lds(const RenderBuild& b);      // dynamic LDS of one wave
bool timeline_build();                                   // this library's image-only kernels fill in PersistentArgs::timeline

// the one-thread-per-pixel kernel (ESVO / CSVO worlds on their own bytes; hits / counters may be null)
hipError_t launch_render_v1(int svo, uint32_t blocks, hipStream_t stream, const vxd::SceneArgs& sc, const vxd::RenderParams& p, void* out_rgba, vx_hit* hits,
                            unsigned long long* counters);

// kernels_aux.hip
hipError_t launch_clock_probe(hipStream_t stream, uint32_t ticks_10ns, unsigned long long* out2);  // out2 (device-visible): {shader cycles, 10 ns ticks}
hipError_t launch_resolve_2x2(hipStream_t stream, const void* src_rgba32f, uint32_t w, uint32_t h, void* dst_rgba32f);
hipError_t launch_picker(int svo, hipStream_t stream, const vxd::SceneArgs& sc, const vx_picker_task* tasks, uint32_t n, vx_picker_result* results);
hipError_t launch_trace(int svo, hipStream_t stream, const vxd::SceneArgs& sc, const TraceArgs& a, vx_result* result, vx_frame* frames, uint32_t max_frames, uint32_t* n_frames);
hipError_t launch_order(hipStream_t stream, const uint32_t* cost, uint32_t tag, uint32_t n, uint32_t* order);
hipError_t launch_scatter(hipStream_t stream, uint32_t pieces, const uint64_t* table, const uint8_t* packed);
// `bytes` (rounded up to 16) from pinned host memory or device memory to device memory, both 16-byte aligned, by a kernel instead of a copy command
hipError_t launch_copy16(hipStream_t stream, void* dst, const void* src, uint64_t bytes);
hipError_t launch_assemble(hipStream_t stream, int format, const void* tiles, uint64_t stride_px, uint32_t tile_count, uint32_t width, uint32_t height, uint32_t tiles_x,
                           const uint32_t* inverse, void* out);

}  // namespace vxk
