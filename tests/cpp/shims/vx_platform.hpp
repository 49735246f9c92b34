// TEST HARNESS ONLY: plain-C++ stand-ins for voxel-rs_amd/csrc/hip/vx_platform.hpp (the gfx950 primitives the device-side
// ray path is written in), so that tests/cpp/device_on_host.cpp can step the DEVICE header against the oracle on a machine
// without a GPU. Found instead of the real one because the harness puts this directory first on its include path. Nothing
// of the product includes or links this.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

namespace vxd {

struct buf_t { const uint8_t* p; uint32_t bytes; };
inline buf_t make_buf(const void* p, uint32_t bytes) { return buf_t{static_cast<const uint8_t*>(p), bytes}; }
inline buf_t make_buf_records8(const void* p, uint32_t bytes) { return make_buf(p, bytes); }  // (only the hand-scheduled loop reads through it: not in this build)
// like the V#'s range check: a read that does not fit the buffer's range WHOLE returns 0. (The product keeps zero padding behind every buffer
// it reads unaligned, inside the descriptor's range -- kWorldPad, kImagePad in runtime.cpp --, so that a straddling read returns the real bytes
// plus zeros; the harness pads the arrays it hands over the same way. A buffer made without its padding shows up here as it would on the GPU.)
inline void buf_read(buf_t b, uint32_t off, void* out, uint32_t n) {
    std::memset(out, 0, n);
    if (uint64_t(off) + n <= b.bytes) std::memcpy(out, b.p + off, n);
}
inline uint32_t buf_u32(buf_t b, uint32_t off) { uint32_t v; buf_read(b, off, &v, 4); return v; }
inline uint32_t buf_u8(buf_t b, uint32_t off) { return off < b.bytes ? b.p[off] : 0u; }
inline uint4 buf_u128(buf_t b, uint32_t off) { uint4 v; buf_read(b, off, &v, 16); return v; }
inline uint2 buf_u64(buf_t b, uint32_t off) { uint2 v; buf_read(b, off, &v, 8); return v; }
inline uint2 mem_u64(const uint8_t* p) { uint2 v; std::memcpy(&v, p, 8); return v; }
inline uint32_t mem_u32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }

extern unsigned char* vx_smem;
#define VX_AS_LDS
#define VX_AS_PRIVATE
#define VX_NOINLINE

inline uint32_t bit_at(uint32_t v, int pos) { return (v >> pos) & 1u; }
inline uint32_t rev_bits32(uint32_t v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    return (v >> 16) | (v << 16);
}
inline void sched_fence() {}
inline float sky_acos(float x) { return acosf(x); }
inline float gmin3(float x, float y, float z) { float m = y < x ? y : x; return z < m ? z : m; }
inline float gmax3(float x, float y, float z) { float m = x < y ? y : x; return m < z ? z : m; }
inline float glsl_pow(float x, float y) { return powf(x, y); }

}  // namespace vxd
