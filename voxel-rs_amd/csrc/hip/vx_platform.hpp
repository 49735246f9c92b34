// gfx950 primitives the device-side ray path (vx_device.hpp) is written in: buffer-resource loads, address-space-qualified
// LDS / scratch access, single-instruction bit-field and min3/max3 forms, hardware exp2/log2. vx_device.hpp includes this
// file as <vx_platform.hpp>; the product is built with this directory on the include path. (The CPU test harness
// tests/cpp/device_on_host.cpp puts a directory with plain-C++ stand-ins of the same names in front of it -- the harness is
// never linked into the product, which has no CPU path.)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vxd {

// Buffer resources (128-bit V#) for everything the rays read: 32-bit byte offsets instead of 64-bit pointers and the
// hardware's range check instead of explicit clamps -- an out-of-range read returns 0, which is also what the CPU
// oracle defines for reads beyond the world buffer, unknown block ids and missing texels.
typedef __amdgpu_buffer_rsrc_t buf_t;
__device__ __forceinline__ buf_t make_buf(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, int(bytes), 0x00020000);
}
// the same memory as records of 8 bytes addressed by their index (`idxen`): what the hand-scheduled loop reads a traversal image's entries through --
// unit index in, no shift; an index beyond the records reads 0
__device__ __forceinline__ buf_t make_buf_records8(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 8, int(bytes >> 3), 0x00020000);
}
__device__ __forceinline__ uint32_t buf_u32(buf_t b, uint32_t off) { return uint32_t(__builtin_amdgcn_raw_buffer_load_b32(b, int(off), 0, 0)); }
__device__ __forceinline__ uint32_t buf_u8(buf_t b, uint32_t off) { return uint32_t(__builtin_amdgcn_raw_buffer_load_b8(b, int(off), 0, 0)); }
__device__ __forceinline__ uint4 buf_u128(buf_t b, uint32_t off) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(b, int(off), 0, 0);
    return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ uint2 buf_u64(buf_t b, uint32_t off) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(b, int(off), 0, 0);
    return make_uint2(v[0], v[1]);
}

// plain 64-bit-addressed loads (buffers beyond what a V#'s 32-bit offsets reach); `p` is suitably aligned by construction
__device__ __forceinline__ uint2 mem_u64(const uint8_t* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ uint32_t mem_u32(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }

// LDS and scratch are reached through address-space-qualified pointers only: a generic pointer would turn every stack
// access into a flat_ instruction plus an aperture test.
// the one dynamic-LDS array of every kernel in this library (16-byte aligned base, cdna guide G17)
extern __shared__ __attribute__((aligned(16))) unsigned char vx_smem[];
#define VX_AS_LDS __attribute__((address_space(3)))
#define VX_AS_PRIVATE __attribute__((address_space(5)))

// a real call instead of inlined code: for rare, register-hungry paths that must not weigh on the loops around them
#define VX_NOINLINE __attribute__((noinline))

// bit `pos` of `v` (one v_bfe_u32)
__device__ __forceinline__ uint32_t bit_at(uint32_t v, int pos) { return __builtin_amdgcn_ubfe(v, uint32_t(pos), 1u); }

// the 32 bits of `v` in reverse order (one v_bfrev_b32)
__device__ __forceinline__ uint32_t rev_bits32(uint32_t v) { return __builtin_bitreverse32(v); }

// instruction-scheduling fence: nothing is moved across it (orders memory requests against the arithmetic that hides them)
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// min / max of three plane distances. The operands are never NaN or -0 for finite rays (p >= 1, t_coef != 0: see
// Trav::init), where the hardware's single v_min3/v_max3 and the GLSL chain min(min(x, y), z) agree bit for bit.
__device__ __forceinline__ float gmin3(float x, float y, float z) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}
__device__ __forceinline__ float gmax3(float x, float y, float z) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}

// pow() as GLSL defines it -- exp2(y * log2(x)), x >= 0 -- on the hardware's log2/exp2 (1 ulp each): within 1.2e-7 absolute of
// the correctly rounded x^y for x in [0, 1 + 1e-4], y in [0, 1000] (27 M points measured on gfx950), at 5 instructions instead
// of the ~170 of a correctly rounded powf. Colour only (the specular term), inside the stated colour tolerance.
__device__ __forceinline__ float glsl_pow(float x, float y) { return y == 0.0f ? 1.0f : __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }

// acos() for the sky's gradient (colour only): Abramowitz & Stegun 4.4.46, sqrt(1 - |x|) times a degree-7 polynomial, absolute error below
// 2e-8 + rounding (measured against a double-precision acos on 16 M points of [-1, 1]: 4.3e-7 rad, i.e. below 5e-7 in the colour), a quarter of the instructions of the library's
// correctly rounded acosf. x in [-1, 1].
__device__ __forceinline__ float sky_acos(float x) {
    const float a = __builtin_fabsf(x);
    float p = -0.0012624911f;
    p = __builtin_fmaf(p, a, 0.0066700901f);
    p = __builtin_fmaf(p, a, -0.0170881256f);
    p = __builtin_fmaf(p, a, 0.0308918810f);
    p = __builtin_fmaf(p, a, -0.0501743046f);
    p = __builtin_fmaf(p, a, 0.0889789874f);
    p = __builtin_fmaf(p, a, -0.2145988016f);
    p = __builtin_fmaf(p, a, 1.5707963050f);
    const float r = __builtin_sqrtf(1.0f - a) * p;
    return x < 0.0f ? 3.14159265358979f - r : r;
}

}  // namespace vxd
