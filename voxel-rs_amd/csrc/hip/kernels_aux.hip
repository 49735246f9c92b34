// The small kernels of libvoxelhip.so: picker and debug trace (picker.glsl:30-51, svo.test.glsl:63-76) on the world's own bytes, the
// 2x2 resolve, the order table's counting sort, the scatter of a commit's packed dirty ranges, the assembly of gathered tile lists.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "vx_device.hpp"

using namespace vxd;
using namespace vxk;

namespace {

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

// Measurement: the shader clock this device runs at NOW -- one lane watches the constant 100 MHz counter (s_memrealtime) for `ticks` of it and
// reports how far the shader-clock counter (s_memtime) moved meanwhile. A wave of its own beside whatever else runs (bench.py samples it
// while the render kernels of its sustained block run: the clock those kernels hold over seconds).
__global__ __launch_bounds__(64) void clock_probe_kernel(uint32_t ticks, unsigned long long* __restrict__ out) {
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1;
    do {
        __builtin_amdgcn_s_sleep(16);
        r1 = __builtin_amdgcn_s_memrealtime();
    } while (r1 - r0 < ticks);
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    out[0] = c1 - c0;
    out[1] = r1 - r0;
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
        face_vector<0>(uint32_t(res.face_id), r.normal);
    } else {
        r.dst = -1.0f;
    }
    results[i] = r;
}

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

// Sorts a frame's sub-tiles for the next frame's queue (PersistentArgs::order): sixteen classes by the iteration count of the sub-tile's
// longest ray (class = min(15, iterations / 16); entries without this frame's tag are class 0), the highest class first, screen
// order within a class (a stable counting sort: the rays of neighbouring sub-tiles walk the same nodes). ONE workgroup of 1024 threads:
// thread t owns the contiguous run of sub-tiles [t * per, (t + 1) * per); (1) it counts its run's members of each class, (2) the
// counts are scanned class by class across the threads -- a wave-level scan by lane shuffles, the sixteen waves' totals through LDS --
// which gives every (class, thread) its first place in the table, (3) it walks its run again and puts every sub-tile in its place.
// 32 K sub-tiles (1080p) are 32 per thread: a few microseconds (round 2's version -- four waves, a ballot per class and block of 64 --
// took 180; it runs behind a frame on a stream of its own, but on the compute units the next frame wants).
__global__ __launch_bounds__(kOrderThreads) void order_kernel(const uint32_t* __restrict__ cost, uint32_t tag, uint32_t n, uint32_t* __restrict__ order) {
    constexpr uint32_t step = kCostStep;
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

// ... and the packed payload's way from the pinned slot to its device twin, as a kernel: a copy command goes to the DMA engines, and on a shared host those
// answer a small request tens of milliseconds late now and then -- the streamer's initial fill (2,000 commits of a few KB to a few MB) took 0.6 to 6 s
// with them, 0.5-0.8 s without (profiles/round6/fill_sdma.sh). 16-byte words, both ends 16-byte aligned.
__global__ __launch_bounds__(256) void copy16_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, uint64_t words) {
    for (uint64_t i = uint64_t(blockIdx.x) * 256 + threadIdx.x; i < words; i += uint64_t(gridDim.x) * 256) dst[i] = src[i];
}

// Scatters gathered compact tile lists back into a row-major image: ONE WAVE per 32x32 tile -- a workgroup of 64 threads needs one free wave slot
// on a compute unit, which a context that renders tile lists keeps free everywhere (runtime.cpp: fifteen persistent waves a CU instead of sixteen,
// -0.8 %): the assembly starts when it is issued instead of when a frame has drained (a 256-thread workgroup needs four slots on ONE unit, and
// with frames in flight the render kernels' queued waves take every slot the moment it is free: measured 190 us late, profiles/round3/pass_t).
// Four times, a lane moves four neighbouring pixels of a row (RGBA8: one 16-byte word; RGBA32F: four): whole 4 KB / 16 KB tiles are read in
// order, whole 128- / 512-byte row segments written. `inverse` = place of every tile in the Morton sequence the ranks share out (place j: rank
// j % tile_count, its local tile j / tile_count). (Any base and stride: 16-byte accesses only where source and destination are aligned.)
__global__ __launch_bounds__(64) void assemble_kernel(const float4* __restrict__ tiles, uint64_t stride_px, uint32_t tile_count, uint32_t width,
                                                      uint32_t height, uint32_t tiles_x, const uint32_t* __restrict__ inverse, float4* __restrict__ out) {
    const uint32_t tile = blockIdx.x, tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t j = inverse[tile];
    const float4* src = tiles + (j % tile_count) * stride_px + size_t(j / tile_count) * (kTile * kTile);
#pragma unroll
    for (uint32_t part = 0; part < 4; ++part) {
        const uint32_t t = part * 64u + threadIdx.x, row = t >> 3, x4 = (t & 7u) * 4u;  // 32 rows x 8 groups of four pixels
        const uint32_t y = ty * kTile + row, x = tx * kTile + x4;
        if (y >= height) continue;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k)
            if (x + k < width) out[size_t(y) * width + x + k] = src[row * kTile + x4 + k];
    }
}

// the same for RGBA8 tile lists and image (vx_target.format = VX_FORMAT_RGBA8; an RGBA8 image has its top row first)
__global__ __launch_bounds__(64) void assemble_kernel_rgba8(const uint32_t* __restrict__ tiles, uint64_t stride_px, uint32_t tile_count, uint32_t width,
                                                            uint32_t height, uint32_t tiles_x, const uint32_t* __restrict__ inverse, uint32_t* __restrict__ out) {
    const uint32_t tile = blockIdx.x, tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t j = inverse[tile];
    const uint32_t* src = tiles + (j % tile_count) * stride_px + size_t(j / tile_count) * (kTile * kTile);
    const bool src_aligned = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
#pragma unroll
    for (uint32_t part = 0; part < 4; ++part) {
        const uint32_t t = part * 64u + threadIdx.x, row = t >> 3, x4 = (t & 7u) * 4u;
        const uint32_t y = ty * kTile + row, x = tx * kTile + x4;
        if (y >= height) continue;
        uint32_t* dst = out + size_t(height - 1u - y) * width + x;
        const uint32_t* from = src + row * kTile + x4;
        if (src_aligned && x + 3 < width && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(from);
        } else {
            for (uint32_t k = 0; k < 4; ++k)
                if (x + k < width) dst[k] = from[k];
        }
    }
}

}  // namespace

namespace vxk {

hipError_t launch_clock_probe(hipStream_t stream, uint32_t ticks, unsigned long long* out) {
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, stream, ticks, out);
    return hipGetLastError();
}

hipError_t launch_resolve_2x2(hipStream_t stream, const void* src, uint32_t w, uint32_t h, void* dst) {
    hipLaunchKernelGGL(resolve_2x2_kernel, dim3((w + 15) / 16, (h + 15) / 16), dim3(256), 0, stream, static_cast<const float4*>(src), w, h, static_cast<float4*>(dst));
    return hipGetLastError();
}

hipError_t launch_picker(int svo, hipStream_t stream, const SceneArgs& sc, const vx_picker_task* tasks, uint32_t n, vx_picker_result* results) {
    const size_t lds = Stack<64>::kBytes;
    const dim3 grid((n + 63) / 64), block(64);
    if (svo == VX_SVO_ESVO_BIG) hipLaunchKernelGGL((picker_kernel<VX_SVO_ESVO_BIG>), grid, block, lds, stream, sc, tasks, n, results);
    else if (svo == VX_SVO_ESVO) hipLaunchKernelGGL((picker_kernel<VX_SVO_ESVO>), grid, block, lds, stream, sc, tasks, n, results);
    else if (svo == VX_SVO_CSVO) hipLaunchKernelGGL((picker_kernel<VX_SVO_CSVO>), grid, block, lds, stream, sc, tasks, n, results);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_trace(int svo, hipStream_t stream, const SceneArgs& sc, const TraceArgs& a, vx_result* result, vx_frame* frames, uint32_t max_frames, uint32_t* n_frames) {
    const size_t lds = Stack<64>::kBytes;
    if (svo == VX_SVO_ESVO_BIG) hipLaunchKernelGGL((trace_kernel<VX_SVO_ESVO_BIG>), dim3(1), dim3(64), lds, stream, sc, a, result, frames, max_frames, n_frames);
    else if (svo == VX_SVO_ESVO) hipLaunchKernelGGL((trace_kernel<VX_SVO_ESVO>), dim3(1), dim3(64), lds, stream, sc, a, result, frames, max_frames, n_frames);
    else if (svo == VX_SVO_CSVO) hipLaunchKernelGGL((trace_kernel<VX_SVO_CSVO>), dim3(1), dim3(64), lds, stream, sc, a, result, frames, max_frames, n_frames);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_order(hipStream_t stream, const uint32_t* cost, uint32_t tag, uint32_t n, uint32_t* order) {
    hipLaunchKernelGGL(order_kernel, dim3(1), dim3(kOrderThreads), 0, stream, cost, tag, n, order);
    return hipGetLastError();
}

hipError_t launch_copy16(hipStream_t stream, void* dst, const void* src, uint64_t bytes) {
    const uint64_t words = (bytes + 15) / 16;
    if (!words) return hipSuccess;
    const uint32_t blocks = uint32_t(std::min<uint64_t>((words + 255) / 256, 2048));
    hipLaunchKernelGGL(copy16_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<uint4*>(dst), static_cast<const uint4*>(src), words);
    return hipGetLastError();
}

hipError_t launch_scatter(hipStream_t stream, uint32_t pieces, const uint64_t* table, const uint8_t* packed) {
    hipLaunchKernelGGL(scatter_kernel, dim3(pieces), dim3(256), 0, stream, table, packed);
    return hipGetLastError();
}

hipError_t launch_assemble(hipStream_t stream, int format, const void* tiles, uint64_t stride_px, uint32_t tile_count, uint32_t width, uint32_t height, uint32_t tiles_x,
                           const uint32_t* inverse, void* out) {
    const uint32_t tiles_y = (height + kTile - 1) / kTile;
    const dim3 grid(tiles_x * tiles_y), block(64);  // a wave per tile
    if (format == VX_FORMAT_RGBA8)
        hipLaunchKernelGGL(assemble_kernel_rgba8, grid, block, 0, stream, static_cast<const uint32_t*>(tiles), stride_px, tile_count, width, height, tiles_x, inverse, static_cast<uint32_t*>(out));
    else
        hipLaunchKernelGGL(assemble_kernel, grid, block, 0, stream, static_cast<const float4*>(tiles), stride_px, tile_count, width, height, tiles_x, inverse, static_cast<float4*>(out));
    return hipGetLastError();
}

}  // namespace vxk
