// render_persistent's traversal loop for cursors on a byte-offset traversal image, hand-scheduled for gfx950.
//
// What it computes is Trav<VX_SVO_IMAGE>::step_image (vx_device.hpp) for every traversing lane, trip after trip, until at most `keep_going`
// lanes of the wave still traverse -- the loop `for (;;) { step; trav = ballot(iter < kMaxSteps); if (popc(trav) <= keep_going) break; }` of
// vx_api.hip -- bit for bit: every float comes out of the operation step_image uses, on the same operands (the parity tests run both
// builds: VX_ASM_LOOP=0 keeps the compiler's loop).
//
// Why by hand. Measured (profiles/tools/valu_issue.hip, profiles/round3/pass_a): with four waves on a SIMD an instruction of ANY kind
// costs the SIMD about 2 cycles -- a scalar mask operation 3.4, a taken branch 8, a compare into a scalar pair 3.4, a select out of
// one 3.0 -- and the traversal loop as the compiler lays it out (78-81 vector + 41-50 scalar instructions per trip, most of the scalar
// ones execution-mask bookkeeping around its PUSH / ADVANCE / POP regions, compares parked in scalar pairs) runs at 91 % of what
// that mix costs: the loop is issue bound, and the way to a faster frame is fewer and cheaper instructions per trip. Here a trip is
// 72 vector + 13 scalar + 6 memory instructions:
//   * compares go to VCC and are consumed by the next instruction (v_cndmask_e32, v_cmpx), one mask (`push`) lives in a scalar pair;
//   * "is a child AND ..." is folded into the compared value (t_min or +inf) instead of and-ing masks;
//   * lanes that stop traversing (at a leaf / led into a voxel / out of the octree) are marked in `iter` itself -- bit 31 = parked,
//     bits 28..30 = the TravStatus -- by one v_add under a narrowed execution mask; the caller decodes it once per service phase;
//   * the cell size is derived from the scale (one shift-add) instead of being carried and selected;
//   * PUSH's five register updates and POP's run under their lanes' execution masks as plain moves / loads into the state registers.
// Variants: the image read through a buffer resource of 8-byte records or (beyond 4 GiB) a 64-bit base; a stack of 13 three-word levels or of 16 levels with
// a 16-bit third plane; worlds in ESVO or CSVO (what happens to a ray that is led into a voxel); with or without a trip counter.
// Hazards (no hazard recognizer looks inside an asm block): no DPP / SDWA / packed / transcendental / lane-access instructions, no SGPR
// written by a VALU is read by a memory instruction, s_cbranch_execz only behind a SALU write of EXEC.
#pragma once

#include "vx_device.hpp"

namespace vxd {

// bits 28..31 of Trav::iter as the loop leaves them on a lane it parked
constexpr uint32_t kLoopParked = 0x80000000u;
__device__ __forceinline__ constexpr uint32_t loop_exit_bits(TravStatus s) { return kLoopParked | (uint32_t(s) << 28); }

// the trip, as one string: LEAF_EXITS / TAKE_MASKS differ by the world's format, COUNT is empty or the trip counter
#define VX_LEAF_EXITS_ESVO                                                                                                         \
    "v_cmpx_lt_f32_e32 vcc, 0, %[tmin]\n"                        /* exec = at a leaf with t_min > 0: leaf_test() decides */       \
    "v_add_u32_e32 %[iter], 0xa0000000, %[iter]\n"               /* parked | kTravAtLeaf << 28 */                                  \
    "s_andn2_b64 exec, %[s_trav], exec\n"                        /* the others go on (into the voxel, if that is where they are led) */
#define VX_LEAF_EXITS_CSVO                                                                                                         \
    "s_mov_b64 %[s_save], exec\n"                                                                                                  \
    "v_add_u32_e32 %[iter], 0xdfffffff, %[iter]\n"               /* parked | kTravForeign << 28, and the iteration taken back */   \
    "v_cmpx_lt_f32_e32 vcc, 0, %[tmin]\n"                                                                                          \
    "v_add_u32_e32 %[iter], 0xc0000001, %[iter]\n"               /* t_min > 0 after all: parked | kTravAtLeaf << 28, the iteration counted (sum: + 0xa0000000) */ \
    "s_andn2_b64 vcc, %[s_save], exec\n"                         /* the lanes that are led into a voxel: they wait for that walk now */ \
    "s_or_b64 %[waiting], %[waiting], vcc\n"                                                                                       \
    "s_andn2_b64 exec, %[s_trav], %[s_save]\n"
// (worlds of at most 12 levels, whose rays led into a voxel are LISTED and run on the world's bytes afterwards -- render_persistent's kForeignRerun build, C3's:
// nothing waits for anything, so nothing is counted and the loop has no second exit: five instructions a trip less than the walk's build)
#define VX_LEAF_EXITS_CSVO_LISTED                                                                                                  \
    "s_mov_b64 %[s_save], exec\n"                                                                                                  \
    "v_add_u32_e32 %[iter], 0xdfffffff, %[iter]\n"               /* parked | kTravForeign << 28, and the iteration taken back */   \
    "v_cmpx_lt_f32_e32 vcc, 0, %[tmin]\n"                                                                                          \
    "v_add_u32_e32 %[iter], 0xc0000001, %[iter]\n"               /* t_min > 0 after all: parked | kTravAtLeaf << 28, the iteration counted */ \
    "s_andn2_b64 exec, %[s_trav], %[s_save]\n"
// ... and as soon as one of them waits (`waiting`: a mask; at entry non-zero if lanes wait from earlier rounds), the wave leaves the loop for the service phase that walks them together (the others' rays pause
// where they are): a shadow ray that starts inside its voxel gets there after `depth` trips, all such rays of a batch in the same trip --
// waiting for the other rays to END first would run the batch's two halves one after the other
#define VX_FOREIGN_EXIT "s_cmp_lg_u64 %[waiting], 0\n s_cbranch_scc1 9f\n"
// (an ESVO world's voxel is walked as an empty node whatever its place in the octant holds: vx_device.hpp, step_image)
// (... and in the wide layout, where `ptr` is dereferenced unchecked, the image's first octant for its pointer)
#define VX_TAKE_ENTRY_ESVO_BYTES "v_mov_b32_e32 %[ptr], v" VX_E0 "\n v_cmp_le_i32_e32 vcc, 0, %[m]\n v_cndmask_b32_e32 %[node], 0, v" VX_E1 ", vcc\n"
#define VX_TAKE_ENTRY_ESVO_UNITS "v_cmp_le_i32_e32 vcc, 0, %[m]\n v_cndmask_b32_e32 %[ptr], 0, v" VX_E0 ", vcc\n v_cndmask_b32_e32 %[node], 0, v" VX_E1 ", vcc\n"
#define VX_TAKE_ENTRY_CSVO "v_mov_b32_e32 %[ptr], v" VX_E0 "\n v_mov_b32_e32 %[node], v" VX_E1 "\n"
#define VX_COUNT_TRIP "s_add_u32 %[trips], %[trips], 1\n"
// (measurement build: which tail a trip took, in ten-bit fields of the same counter -- a loop call makes at most kMaxSteps = 1000 trips)
#define VX_COUNT_TRIP_ADVANCE_ONLY "s_add_u32 %[trips], %[trips], 0x401\n"
#define VX_COUNT_TRIP_PUSH_ONLY "s_add_u32 %[trips], %[trips], 0x100001\n"
// the entry of the child the ray is in: unit ptr + popcount(%[t2]) of the image (8-byte units; %[t2] = "exists" of this child and of the existing ones above
// it: traversal_image.hpp) -- through a buffer resource of 8-byte records (out of range reads 0: any `ptr` is harmless) | (images beyond 4 GiB, which no
// buffer resource reaches -- its offsets are 32 bits, measured) behind a 64-bit base, so `ptr` has to stay inside the image (smaller than 32 GiB)
#define VX_LOAD_ENTRY_BYTES "v_bcnt_u32_b32 %[t1], %[t2], %[ptr]\n buffer_load_dwordx2 v[" VX_E0 ":" VX_E1 "], %[t1], %[rsrc], 0 idxen\n"
#define VX_LOAD_ENTRY_UNITS                                                                                                        \
    "v_bcnt_u32_b32 v" VX_A0 ", %[t2], %[ptr]\n v_mov_b32_e32 v" VX_A1 ", 0\n v_lshl_add_u64 v[" VX_A0 ":" VX_A1 "], v[" VX_A0 ":" VX_A1 "], 3, %[base]\n"         \
    "global_load_dwordx2 v[" VX_E0 ":" VX_E1 "], v[" VX_A0 ":" VX_A1 "], off\n"
// the stack: 13 levels of three words (planes 13 x 256 bytes apart) | 16 levels with a 16-bit third plane (the masks' upper half: all a
// cursor on an image needs), planes 16 x 256 bytes apart, the third at half the slot's offset behind them
#define VX_STACK_WRITE_13 "ds_write2st64_b32 %[t0], %[ptr], %[tmax] offset1:13\n ds_write_b32 %[t0], %[node] offset:6656\n"
#define VX_STACK_WRITE_16 "v_lshl_add_u32 %[t1], %[sc], 7, %[lds16]\n ds_write2st64_b32 %[t0], %[ptr], %[tmax] offset1:16\n ds_write_b16_d16_hi %[t1], %[node]\n"
#define VX_STACK_READ_13 "ds_read_b32 %[ptr], %[oct]\n ds_read_b32 %[tmax], %[oct] offset:3328\n ds_read_b32 %[node], %[oct] offset:6656\n"
#define VX_STACK_READ_16 "v_lshl_add_u32 %[m], %[sc], 7, %[lds16]\n ds_read_b32 %[ptr], %[oct]\n ds_read_b32 %[tmax], %[oct] offset:4096\n ds_read_u16_d16_hi %[node], %[m]\n"
// who still traverses, and whether the wave goes on
#define VX_LOOP_CONTROL_false(COUNT)                                                                                               \
        "s_mov_b64 exec, %[s_trav]\n"                                                                                              \
        COUNT    /* (also carries the FOREIGN builds' second exit) */                                                              \
        "v_cmp_gt_u32_e32 vcc, 0x3e8, %[iter]\n"                                                                                   \
        "s_bcnt1_i32_b64 %[s_n], vcc\n"                                                                                            \
        "s_cmp_gt_u32 %[s_n], %[keep]\n"                                                                                           \
        "s_cbranch_scc1 1b\n"
// ... in lockstep (keep_going == 0, the product's setting: the wave goes on while ANY lane traverses): two instructions less
#define VX_LOOP_CONTROL_true(COUNT)                                                                                                \
        "s_mov_b64 exec, %[s_trav]\n"                                                                                              \
        COUNT                                                                                                                      \
        "v_cmp_gt_u32_e32 vcc, 0x3e8, %[iter]\n"                                                                                   \
        "s_cbranch_vccnz 1b\n"
// POP (the execution mask = the lanes whose ADVANCE left the parent; %[t0] = the bits in which their corner changed)
#define VX_TRIP_POP(STACK_READ)                                                                                                    \
        "v_ffbh_u32_e32 %[t1], %[t0]\n"                                                                                            \
        "v_sub_u32_e32 %[sc], 31, %[t1]\n"                       /* the highest differing bit */                                   \
        "v_lshl_add_u32 %[oct], %[sc], 8, %[lds]\n"                                                                                \
        STACK_READ                                                                                                                 \
        "v_lshlrev_b32_e64 %[t1], %[sc], -1\n"                                                                                     \
        "v_and_b32_e32 %[px], %[t1], %[px]\n"                                                                                      \
        "v_and_b32_e32 %[py], %[t1], %[py]\n"                                                                                      \
        "v_and_b32_e32 %[pz], %[t1], %[pz]\n"                                                                                      \
        "v_cmpx_lt_u32_e32 vcc, 22, %[sc]\n"                     /* out of the octree */                                           \
        "v_add_u32_e32 %[iter], 0xc0000000, %[iter]\n"           /* parked | kTravFinished << 28 */
// The rays of a sub-tile's 64 pixels walk the upper levels of the tree together: in a fifth of the trips every traversing lane PUSHes, or every
// one ADVANCEs (measured, C3 in lockstep: 11.4 % / 10.7 %, profiles/round5/pass_a/tails.txt). The trip therefore has three tails behind its common part (the child, the entry request, the plane distances, the leaf
// exits, the PUSH mask): the merged one (both kinds of lane), and the two it degenerates to when the mask is all or none -- 31 and 22
// instructions shorter, the ADVANCE-only one without the wait for the entry it requested for nothing.
#define VX_LOOP_ASM(LEAF_EXITS, TAKE_MASKS, COUNT, COUNT_A, COUNT_P, LOAD_ENTRY, STACK_WRITE, STACK_READ, LOCKSTEP)                          \
        "v_cmp_gt_u32_e32 vcc, 0x3e8, %[iter]\n"                                                                                   \
        "s_cmp_eq_u64 vcc, 0\n"                                                                                                    \
        "s_cbranch_scc1 9f\n"                                                                                                      \
        "1:\n"                                                                                                                     \
        "s_mov_b64 exec, vcc\n"                                                                                                    \
        "s_mov_b64 %[s_trav], vcc\n"                                                                                               \
        /* ---- the child the ray is in, its entry requested, the planes' distances ---- */                                       \
        "v_add_u32_e32 %[iter], 1, %[iter]\n"                                                                                      \
        "v_bfe_u32 %[t0], %[px], %[sc], 1\n"                                                                                       \
        "v_bfe_u32 %[t1], %[py], %[sc], 1\n"                                                                                       \
        "v_bfe_u32 %[t2], %[pz], %[sc], 1\n"                                                                                       \
        "v_lshlrev_b32_e32 %[t1], 1, %[t1]\n"                                                                                      \
        "v_lshl_or_b32 %[t0], %[t2], 2, %[t0]\n"                                                                                   \
        "v_bitop3_b32 %[oct], %[t0], %[om], %[t1] bitop3:0x36\n" /* (t0 | t1) ^ octant_mask */                                     \
        "v_lshlrev_b32_e32 %[m], %[oct], %[node]\n"              /* the child's "is a leaf" bit in the sign, its "exists" bit at 23 ... */ \
        "v_and_b32_e32 %[t2], 0xffffff, %[m]\n"                  /* ... and below it "exists" of the children above: their count = where the entry lies */ \
        LOAD_ENTRY                                                                                                                 \
        "v_fma_f32 %[crx], %[px], %[tcx], -%[tbx]\n"                                                                               \
        "v_fma_f32 %[cry], %[py], %[tcy], -%[tby]\n"                                                                               \
        "v_fma_f32 %[crz], %[pz], %[tcz], -%[tbz]\n"                                                                               \
        "v_min3_f32 %[tcm], %[crx], %[cry], %[crz]\n"                                                                              \
        /* (min(t_max, tc_max), the t_max a PUSH hands on, IS tc_max: the child's cell lies in the node's, t_max is the node's own exit distance and the  */ \
        /* plane distances are monotone in the corner -- fma rounds once --, so the child's exit is never later. vx_device.hpp keeps the reference's min.) */ \
        "v_lshl_add_u32 %[sx], %[sc], 23, %[k_cell]\n"                                                                             \
        "v_lshl_add_u32 %[hf], %[sc], 23, %[k_half]\n"                                                                             \
        /* ---- a leaf the ray reaches (is a leaf, t_min <= t_max): the lane stops here ---- */                                    \
        "v_cmpx_gt_i32_e32 vcc, 0, %[m]\n"                       /* exec: ... whose child is a leaf (a leaf is a child) */         \
        "v_cmpx_le_f32_e32 vcc, %[tmin], %[tmax]\n"              /* ... and is reached */                                          \
        LEAF_EXITS                                                                                                                 \
        /* ---- PUSH or ADVANCE ---- */                                                                                            \
        "v_cmp_le_u32_e32 vcc, 0x800000, %[t2]\n"                /* the child exists (bit 23 of what is left of m) */              \
        "v_cndmask_b32_e32 %[tq], %[inf], %[tmin], vcc\n"        /* t_min, or +inf where there is no child */                      \
        "v_cmp_le_f32_e64 %[s_push], %[tq], %[tcm]\n"            /* PUSH: a child, and t_min <= min(t_max, tc_max) = tc_max */      \
        "s_cmp_eq_u64 %[s_push], 0\n"                                                                                              \
        "s_cbranch_scc1 4f\n"                                    /* nobody: the ADVANCE-only tail */                               \
        "s_cmp_eq_u64 %[s_push], exec\n"                                                                                           \
        "s_cbranch_scc1 5f\n"                                    /* everybody: the PUSH-only tail */                               \
        /* ======== both kinds of lane ======== */                                                                                 \
        "v_cndmask_b32_e64 %[hm], 0, %[hf], %[s_push]\n"         /* half a cell | 0 */                                             \
        "v_cndmask_b32_e64 %[ot], -%[sx], 0, %[s_push]\n"        /* 0 | minus a cell */                                            \
        "v_cndmask_b32_e64 %[tmin], %[tcm], %[tmin], %[s_push]\n" /* ADVANCE: t_min = tc_max */                                    \
        "v_fmac_f32_e32 %[crx], %[hm], %[tcx]\n"                 /* a PUSH lane's centre planes, the others' corner planes still */ \
        "v_fmac_f32_e32 %[cry], %[hm], %[tcy]\n"                                                                                   \
        "v_fmac_f32_e32 %[crz], %[hm], %[tcz]\n"                                                                                   \
        /* the parent's entry goes on the stack -- at EVERY push (the reference writes it only where the ray leaves the child before it  */ \
        /* leaves the parent, tc_max < h: the writes it skips are of entries that are never popped, or of the very entry the slot holds:  */ \
        /* vx_device.hpp, step_image -- so the image cursor keeps no h at all)                                                             */ \
        "v_lshl_add_u32 %[t0], %[sc], 8, %[lds]\n"                                                                                 \
        "s_and_saveexec_b64 %[s_save], %[s_push]\n"                                                                                \
        STACK_WRITE                                                                                                                \
        "s_mov_b64 exec, %[s_save]\n"                                                                                              \
        /* the corner: += half a cell where t_min < t(centre) | -= a cell where tc_max >= t(corner) */                             \
        "v_cmp_lt_f32_e32 vcc, %[tmin], %[crx]\n"                                                                                  \
        "v_cndmask_b32_e32 %[t1], %[ot], %[hm], vcc\n"                                                                             \
        "v_add_f32_e32 %[nx], %[px], %[t1]\n"                                                                                      \
        "v_cmp_lt_f32_e32 vcc, %[tmin], %[cry]\n"                                                                                  \
        "v_cndmask_b32_e32 %[t1], %[ot], %[hm], vcc\n"                                                                             \
        "v_add_f32_e32 %[ny], %[py], %[t1]\n"                                                                                      \
        "v_cmp_lt_f32_e32 vcc, %[tmin], %[crz]\n"                                                                                  \
        "v_cndmask_b32_e32 %[t1], %[ot], %[hm], vcc\n"                                                                             \
        "v_add_f32_e32 %[nz], %[pz], %[t1]\n"                                                                                      \
        /* the bits in which the corner changed: above bit `scale` exactly when an ADVANCE left the parent (a PUSH changes bit scale - 1) */ \
        "v_xor_b32_e32 %[t0], %[px], %[nx]\n"                                                                                      \
        "v_bitop3_b32 %[t0], %[t0], %[py], %[ny] bitop3:0xf6\n"  /* a | (b ^ c) */                                                 \
        "v_bitop3_b32 %[t0], %[t0], %[pz], %[nz] bitop3:0xf6\n"                                                                    \
        "v_mov_b32_e32 %[px], %[nx]\n"                                                                                             \
        "v_mov_b32_e32 %[py], %[ny]\n"                                                                                             \
        "v_mov_b32_e32 %[pz], %[nz]\n"                                                                                             \
        "v_lshlrev_b32_e64 %[t1], %[sc], 2\n"                                                                                      \
        "v_cmp_ge_u32_e32 vcc, %[t0], %[t1]\n"                                                                                     \
        "s_and_saveexec_b64 %[s_save], vcc\n"                                                                                      \
        "s_cbranch_execz 2f\n"                                                                                                     \
        VX_TRIP_POP(STACK_READ)                                                                                                    \
        "2:\n"                                                                                                                     \
        /* PUSH: the child becomes the node */                                                                                     \
        "s_and_b64 exec, %[s_save], %[s_push]\n"                                                                                   \
        "v_add_u32_e32 %[sc], -1, %[sc]\n"                                                                                         \
        "v_mov_b32_e32 %[tmax], %[tcm]\n"                                                                                          \
        "s_waitcnt vmcnt(0)\n"                                                                                                     \
        TAKE_MASKS                                                                                                                 \
        "s_waitcnt lgkmcnt(0)\n"                                                                                                   \
        VX_LOOP_CONTROL_##LOCKSTEP(COUNT)                                                                                                     \
        "s_branch 9f\n"                                                                                                            \
        /* ======== every lane ADVANCEs ======== */                                                                                \
        "4:\n"                                                                                                                     \
        "v_mov_b32_e32 %[tmin], %[tcm]\n"                                                                                          \
        "v_cmp_ge_f32_e32 vcc, %[tcm], %[crx]\n"                                                                                   \
        "v_cndmask_b32_e32 %[t1], 0, %[sx], vcc\n"                                                                                 \
        "v_sub_f32_e32 %[nx], %[px], %[t1]\n"                                                                                      \
        "v_cmp_ge_f32_e32 vcc, %[tcm], %[cry]\n"                                                                                   \
        "v_cndmask_b32_e32 %[t1], 0, %[sx], vcc\n"                                                                                 \
        "v_sub_f32_e32 %[ny], %[py], %[t1]\n"                                                                                      \
        "v_cmp_ge_f32_e32 vcc, %[tcm], %[crz]\n"                                                                                   \
        "v_cndmask_b32_e32 %[t1], 0, %[sx], vcc\n"                                                                                 \
        "v_sub_f32_e32 %[nz], %[pz], %[t1]\n"                                                                                      \
        "v_xor_b32_e32 %[t0], %[px], %[nx]\n"                                                                                      \
        "v_bitop3_b32 %[t0], %[t0], %[py], %[ny] bitop3:0xf6\n"                                                                    \
        "v_bitop3_b32 %[t0], %[t0], %[pz], %[nz] bitop3:0xf6\n"                                                                    \
        "v_mov_b32_e32 %[px], %[nx]\n"                                                                                             \
        "v_mov_b32_e32 %[py], %[ny]\n"                                                                                             \
        "v_mov_b32_e32 %[pz], %[nz]\n"                                                                                             \
        "v_lshlrev_b32_e64 %[t1], %[sc], 2\n"                                                                                      \
        "v_cmp_ge_u32_e32 vcc, %[t0], %[t1]\n"                                                                                     \
        "s_and_saveexec_b64 %[s_save], vcc\n"                                                                                      \
        "s_cbranch_execz 6f\n"                                                                                                     \
        VX_TRIP_POP(STACK_READ)                                                                                                    \
        "s_waitcnt lgkmcnt(0)\n"                                                                                                   \
        "6:\n"                                                                                                                     \
        VX_LOOP_CONTROL_##LOCKSTEP(COUNT_A)                                                                                                   \
        "s_branch 9f\n"                                                                                                            \
        /* ======== every lane PUSHes ======== */                                                                                  \
        "5:\n"                                                                                                                     \
        "v_fmac_f32_e32 %[crx], %[hf], %[tcx]\n"                 /* the centre planes' distances */                                \
        "v_fmac_f32_e32 %[cry], %[hf], %[tcy]\n"                                                                                   \
        "v_fmac_f32_e32 %[crz], %[hf], %[tcz]\n"                                                                                   \
        "v_lshl_add_u32 %[t0], %[sc], 8, %[lds]\n"                                                                                 \
        STACK_WRITE                                                                                                                \
        "v_cmp_lt_f32_e32 vcc, %[tmin], %[crx]\n"                                                                                  \
        "v_cndmask_b32_e32 %[t1], 0, %[hf], vcc\n"                                                                                 \
        "v_add_f32_e32 %[px], %[px], %[t1]\n"                                                                                      \
        "v_cmp_lt_f32_e32 vcc, %[tmin], %[cry]\n"                                                                                  \
        "v_cndmask_b32_e32 %[t1], 0, %[hf], vcc\n"                                                                                 \
        "v_add_f32_e32 %[py], %[py], %[t1]\n"                                                                                      \
        "v_cmp_lt_f32_e32 vcc, %[tmin], %[crz]\n"                                                                                  \
        "v_cndmask_b32_e32 %[t1], 0, %[hf], vcc\n"                                                                                 \
        "v_add_f32_e32 %[pz], %[pz], %[t1]\n"                                                                                      \
        "v_add_u32_e32 %[sc], -1, %[sc]\n"                                                                                         \
        "v_mov_b32_e32 %[tmax], %[tcm]\n"                                                                                          \
        "s_waitcnt vmcnt(0)\n"                                                                                                     \
        TAKE_MASKS                                                                                                                 \
        VX_LOOP_CONTROL_##LOCKSTEP(COUNT_P)                                                                                                   \
        "9:\n"                                                                                                                     \
        /* (the ADVANCE-only tail leaves its entry request in flight: it must have landed before the compiler's code reuses the pair -- */ \
        /* where every register is in use it did not always: one pixel in a few frames differed) */                                    \
        "s_waitcnt vmcnt(0)\n"                                                                                                     \
        "s_mov_b64 exec, %[entry_exec]\n"

// FOREIGN != 0: the image of a CSVO world -- a ray about to be led into a voxel leaves the loop (kTravForeign, its iteration not counted);
// otherwise (ESVO world) it walks the voxel as an empty node. COUNT: count the trips in `trips` (measurement).
// SVO: VX_SVO_IMAGE (`image` = a resource of 8-byte records over the image: make_buf_records8) or VX_SVO_IMAGE_WIDE (8-byte units behind `image_base`; the image
// must be smaller than 32 GiB). LEVELS: 13 (three-word slots) or 16 (the
// 16-bit third plane); lds_slot0 / lds_aux0 = the LDS addresses of this lane's slot for scale 0 in the first and in the third plane.
// The caller guarantees that no traversing lane has kHasAdjacentLeaf set (the loop does not clear it; such rays -- they have just
// passed a translucent voxel -- are rare and take the compiler's loop); kInsideVoxel is not maintained (nothing in a render reads it).
// FOREIGN: 0 = the image of an ESVO world, 1 = of a CSVO world whose rays walk inside voxels in a service phase (the loop counts the lanes that wait for that
// and leaves when `foreign_min` do), 2 = of a CSVO world whose such rays are listed (no waiting, no second exit).
// LOCKSTEP: the caller has found keep_going == 0 (wave-uniform): the loop's control is then a branch on the traversing lanes' mask alone. (Product builds only:
// the measurement build's loop, which counts its trips, keeps the general control.)
template <int SVO, int FOREIGN, bool COUNT, int LEVELS, bool LOCKSTEP = false>
__device__ __forceinline__ void traverse_loop_gfx950(Trav<SVO>& tr, buf_t image, const uint8_t* image_base, uint32_t lds_slot0, uint32_t lds_aux0, uint32_t keep_going,
                                                     uint32_t foreign_waiting, uint32_t foreign_min, uint32_t& trips, unsigned long long* tails = nullptr) {
    static_assert(SVO == VX_SVO_IMAGE || SVO == VX_SVO_IMAGE_WIDE, "cursors on a traversal image");
    static_assert(LEVELS == 13 || LEVELS == 16, "stack layouts: Stack<64, true, true, 13>, Stack<64, true, true, 16, true>");
    constexpr bool UNITS = SVO == VX_SVO_IMAGE_WIDE;
    uint32_t t0, t1, t2, oct, m, nx, ny, nz;
    float crx, cry, crz, tcm, tq, hf, hm, ot, sx;
    unsigned long long s_trav, s_push, s_save;
    uint32_t s_n;
    uint32_t px = __float_as_uint(tr.px), py = __float_as_uint(tr.py), pz = __float_as_uint(tr.pz);
    uint32_t scale = uint32_t(tr.scale);
    const uint32_t k_cell = 0x34000000u, k_half = 0x33800000u;  // 2^(scale - 23) = (scale + 104) << 23, half of it = (scale + 103) << 23
    const unsigned long long entry_exec = __builtin_amdgcn_read_exec();
    // (wave-uniform by construction; the compiler is told so)
    keep_going = uint32_t(__builtin_amdgcn_readfirstlane(int(keep_going)));
    uint32_t n_trips = 0;
    // FOREIGN: lanes that wait for their walk into a voxel (at entry: those of earlier rounds), and how many of them make the wave leave the loop
    // (a mask of the lanes that have come to wait in this call; at entry: whether any waits already. The caller's `foreign_min` is 1 wherever lanes wait at all:
    // the loop leaves for the walk as soon as one lane does)
    unsigned long long waiting = (unsigned long long)uint32_t(__builtin_amdgcn_readfirstlane(int(foreign_waiting)));  // (their number: non-zero is all that counts)
    (void)foreign_min;
#define VX_LOOP_OPERANDS                                                                                                                                   \
        : [px] "+v"(px), [py] "+v"(py), [pz] "+v"(pz), [tmin] "+v"(tr.t_min), [tmax] "+v"(tr.t_max), [sc] "+v"(scale), [ptr] "+v"(tr.ptr),  \
          [node] "+v"(tr.node), [iter] "+v"(tr.iter), [trips] "+s"(n_trips), [waiting] "+s"(waiting), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [oct] "=&v"(oct), [m] "=&v"(m), \
          [nx] "=&v"(nx), [ny] "=&v"(ny), [nz] "=&v"(nz), [crx] "=&v"(crx), [cry] "=&v"(cry), [crz] "=&v"(crz), [tcm] "=&v"(tcm),         \
          [tq] "=&v"(tq), [hf] "=&v"(hf), [hm] "=&v"(hm), [ot] "=&v"(ot), [sx] "=&v"(sx), [s_trav] "=&s"(s_trav), [s_push] "=&s"(s_push),                  \
          [s_save] "=&s"(s_save), [s_n] "=&s"(s_n)                                                                                                         \
        : [tcx] "v"(tr.tcx), [tcy] "v"(tr.tcy), [tcz] "v"(tr.tcz), [tbx] "v"(tr.tbx), [tby] "v"(tr.tby), [tbz] "v"(tr.tbz), [om] "v"(uint32_t(tr.octant_mask)), \
          [lds] "v"(lds_slot0), [lds16] "v"(lds_aux0), [inf] "v"(0x7f800000u), [rsrc] "s"(image), [base] "s"(image_base), [keep] "s"(keep_going), [k_cell] "s"(k_cell),           \
          [k_half] "s"(k_half), [entry_exec] "s"(entry_exec)                                                                                               \
        : "v" VX_A0, "v" VX_A1, "v" VX_E0, "v" VX_E1, "vcc", "scc", "memory"
#define VX_LOOP_VARIANT(F, C, U, L, K)                                                                                                                     \
    if constexpr (FOREIGN == F && COUNT == C && UNITS == U && LEVELS == L && (LOCKSTEP && !COUNT) == K)                                                     \
        asm volatile(VX_LOOP_ASM(VX_F_LEAF_##F, VX_F_TAKE_##F(U),                                                                                         \
                                 VX_LOOP_PICK_##C(VX_COUNT_TRIP, "") VX_F_EXIT_##F, VX_LOOP_PICK_##C(VX_COUNT_TRIP_ADVANCE_ONLY, "") VX_F_EXIT_##F,     \
                                 VX_LOOP_PICK_##C(VX_COUNT_TRIP_PUSH_ONLY, "") VX_F_EXIT_##F, VX_LOOP_PICK_##U(VX_LOAD_ENTRY_UNITS, VX_LOAD_ENTRY_BYTES),                 \
                                 VX_STACK_WRITE_##L, VX_STACK_READ_##L, K) VX_LOOP_OPERANDS)
#define VX_F_LEAF_0 VX_LEAF_EXITS_ESVO
#define VX_F_LEAF_1 VX_LEAF_EXITS_CSVO
#define VX_F_LEAF_2 VX_LEAF_EXITS_CSVO_LISTED
#define VX_F_TAKE_0(U) VX_LOOP_PICK_##U(VX_TAKE_ENTRY_ESVO_UNITS, VX_TAKE_ENTRY_ESVO_BYTES)
#define VX_F_TAKE_1(U) VX_TAKE_ENTRY_CSVO
#define VX_F_TAKE_2(U) VX_TAKE_ENTRY_CSVO
#define VX_F_EXIT_0 ""
#define VX_F_EXIT_1 VX_FOREIGN_EXIT
#define VX_F_EXIT_2 ""
#define VX_LOOP_PICK_true(a, b) a
#define VX_LOOP_PICK_false(a, b) b
    // (two pairs of fixed registers -- the entry a trip requests, the 64-bit address of it in the wide layout: an asm operand cannot name the
    // halves of a pair -- at the top of the build's register budget: 128 at four waves per SIMD)
#define VX_E0 "124"
#define VX_E1 "125"
#define VX_A0 "122"
#define VX_A1 "123"
    VX_LOOP_VARIANT(0, false, false, 13, false); VX_LOOP_VARIANT(0, false, false, 13, true); VX_LOOP_VARIANT(0, false, false, 16, false); VX_LOOP_VARIANT(0, false, false, 16, true); VX_LOOP_VARIANT(0, false, true, 13, false); VX_LOOP_VARIANT(0, false, true, 13, true); VX_LOOP_VARIANT(0, false, true, 16, false); VX_LOOP_VARIANT(0, false, true, 16, true);
    VX_LOOP_VARIANT(0, true, false, 13, false); VX_LOOP_VARIANT(0, true, false, 16, false); VX_LOOP_VARIANT(0, true, true, 13, false); VX_LOOP_VARIANT(0, true, true, 16, false);
    VX_LOOP_VARIANT(1, false, false, 13, false); VX_LOOP_VARIANT(1, false, false, 13, true); VX_LOOP_VARIANT(1, false, false, 16, false); VX_LOOP_VARIANT(1, false, false, 16, true); VX_LOOP_VARIANT(1, false, true, 13, false); VX_LOOP_VARIANT(1, false, true, 13, true); VX_LOOP_VARIANT(1, false, true, 16, false); VX_LOOP_VARIANT(1, false, true, 16, true);
    VX_LOOP_VARIANT(1, true, false, 13, false); VX_LOOP_VARIANT(1, true, false, 16, false); VX_LOOP_VARIANT(1, true, true, 13, false); VX_LOOP_VARIANT(1, true, true, 16, false);
    VX_LOOP_VARIANT(2, false, false, 13, false); VX_LOOP_VARIANT(2, false, false, 13, true); VX_LOOP_VARIANT(2, false, false, 16, false); VX_LOOP_VARIANT(2, false, false, 16, true); VX_LOOP_VARIANT(2, false, true, 13, false); VX_LOOP_VARIANT(2, false, true, 13, true); VX_LOOP_VARIANT(2, false, true, 16, false); VX_LOOP_VARIANT(2, false, true, 16, true);
    VX_LOOP_VARIANT(2, true, false, 13, false); VX_LOOP_VARIANT(2, true, false, 16, false); VX_LOOP_VARIANT(2, true, true, 13, false); VX_LOOP_VARIANT(2, true, true, 16, false);
#undef VX_E0
#undef VX_E1
#undef VX_A0
#undef VX_A1
#undef VX_LOOP_VARIANT
#undef VX_F_LEAF_0
#undef VX_F_LEAF_1
#undef VX_F_LEAF_2
#undef VX_F_TAKE_0
#undef VX_F_TAKE_1
#undef VX_F_TAKE_2
#undef VX_F_EXIT_0
#undef VX_F_EXIT_1
#undef VX_F_EXIT_2
#undef VX_LOOP_PICK_true
#undef VX_LOOP_PICK_false
#undef VX_LOOP_OPERANDS
    if constexpr (COUNT) {
        // (measurement build) n_trips = trips | ADVANCE-only trips << 10 | PUSH-only trips << 20 of this call; *tails accumulates the two kinds in 32-bit halves
        trips += n_trips & 0x3ffu;
        if (tails) *tails += (unsigned long long)((n_trips >> 10) & 0x3ffu) | ((unsigned long long)((n_trips >> 20) & 0x3ffu) << 32);
    } else {
        trips += n_trips;
    }
    tr.px = __uint_as_float(px); tr.py = __uint_as_float(py); tr.pz = __uint_as_float(pz);
    tr.scale = int(scale);
    tr.scale_exp2 = pow2i(tr.scale - kMaxScale);  // (the loop derives the cell size from the scale; the service phases read the member)
}

// Trav::descend_along (vx_device.hpp) by hand, for the same reason as the loop: as the compiler lays the per-level loop out -- three regions behind
// execution-mask bookkeeping, three taken branches, packed fmas at 3.4 cycles each -- a level costs a wave ~150 cycles, 640 at four waves a SIMD, and the
// levels saved (a trip of the render loop each) barely pay for it (measured: profiles/round6, timelines before / after). Here a level is 33
// instructions and one taken branch. Bit for bit the C++ (the parity tests run this one on the GPU, the host harness steps the other against the oracle).
// LEVELS: 13 (three-word slots) or 16 (16-bit third plane): where a slot's t_max lies behind its pointer. Called by the lanes whose shadow ray has a path
// to follow (any execution mask); `lds_slot0` as for the loop. Leaves the cursor's pointer and masks to the caller (Stack::pop of the scale reached).
template <int LEVELS, class TRAV>
__device__ __forceinline__ void descend_along_gfx950(TRAV& tr, uint32_t lds_slot0, int parent_scale, const float q[3]) {
    static_assert(LEVELS == 13 || LEVELS == 16, "stack layouts: Stack<64, true, true, 13>, Stack<64, true, true, 16, true>");
    const float cell = pow2i(parent_scale - kMaxScale);
    const uint32_t ex = __float_as_uint((tr.octant_mask & 1) ? 3.0f - cell - q[0] : q[0]);
    const uint32_t ey = __float_as_uint((tr.octant_mask & 2) ? 3.0f - cell - q[1] : q[1]);
    const uint32_t ez = __float_as_uint((tr.octant_mask & 4) ? 3.0f - cell - q[2] : q[2]);
    uint32_t px = __float_as_uint(tr.px), py = __float_as_uint(tr.py), pz = __float_as_uint(tr.pz);
    uint32_t scale = uint32_t(tr.scale);
    uint32_t t0, t1, t2;
    float crx, cry, crz, tcm, hf;
    unsigned long long s_entry;
    const uint32_t k_half = 0x33800000u;  // half a cell at `scale`: 2^(scale - 24) = (scale + 103) << 23
#define VX_DESCEND_ASM(TMAX_OFFSET)                                                                                                 \
        "s_mov_b64 %[s_entry], exec\n"                                                                                             \
        "1:\n"                                                                                                                     \
        "v_xor_b32_e32 %[t0], %[px], %[ex]\n"                                                                                      \
        "v_xor_b32_e32 %[t1], %[py], %[ey]\n"                                                                                      \
        "v_xor_b32_e32 %[t2], %[pz], %[ez]\n"                                                                                      \
        "v_or3_b32 %[t0], %[t0], %[t1], %[t2]\n"                                                                                   \
        "v_lshrrev_b32_e32 %[t0], %[sc], %[t0]\n"                                                                                  \
        "v_cmpx_lt_i32_e32 vcc, %[ps], %[sc]\n"                  /* above the voxel's parent */                                    \
        "v_cmpx_eq_u32_e32 vcc, 0, %[t0]\n"                      /* the cell the cursor is at is on the path */                    \
        "v_fma_f32 %[crx], %[px], %[tcx], -%[tbx]\n"                                                                               \
        "v_fma_f32 %[cry], %[py], %[tcy], -%[tby]\n"                                                                               \
        "v_fma_f32 %[crz], %[pz], %[tcz], -%[tbz]\n"                                                                               \
        "v_min3_f32 %[tcm], %[crx], %[cry], %[crz]\n"                                                                              \
        "v_cmpx_le_f32_e32 vcc, %[tmin], %[tmax]\n"              /* the reference PUSHes: t_min <= t_max ... */                    \
        "v_cmp_le_f32_e32 vcc, %[tmin], %[tcm]\n"                /* ... and t_min <= min(t_max, tc_max), which is tc_max (see the loop) */ \
        "s_and_b64 exec, exec, vcc\n"                                                                                              \
        "s_cbranch_execz 9f\n"                                                                                                     \
        "v_lshl_add_u32 %[t0], %[sc], 8, %[lds]\n"                                                                                 \
        "ds_write_b32 %[t0], %[tmax] offset:" #TMAX_OFFSET "\n"  /* this ray's t_max in the parent's slot (pointer and masks are in place) */ \
        "v_lshl_add_u32 %[hf], %[sc], 23, %[k_half]\n"                                                                             \
        "v_fmac_f32_e32 %[crx], %[hf], %[tcx]\n"                 /* the centre planes' distances */                                \
        "v_fmac_f32_e32 %[cry], %[hf], %[tcy]\n"                                                                                   \
        "v_fmac_f32_e32 %[crz], %[hf], %[tcz]\n"                                                                                   \
        "v_cmp_lt_f32_e32 vcc, %[tmin], %[crx]\n"                                                                                  \
        "v_cndmask_b32_e32 %[t1], 0, %[hf], vcc\n"                                                                                 \
        "v_add_f32_e32 %[px], %[px], %[t1]\n"                                                                                      \
        "v_cmp_lt_f32_e32 vcc, %[tmin], %[cry]\n"                                                                                  \
        "v_cndmask_b32_e32 %[t1], 0, %[hf], vcc\n"                                                                                 \
        "v_add_f32_e32 %[py], %[py], %[t1]\n"                                                                                      \
        "v_cmp_lt_f32_e32 vcc, %[tmin], %[crz]\n"                                                                                  \
        "v_cndmask_b32_e32 %[t1], 0, %[hf], vcc\n"                                                                                 \
        "v_add_f32_e32 %[pz], %[pz], %[t1]\n"                                                                                      \
        "v_mov_b32_e32 %[tmax], %[tcm]\n"                                                                                          \
        "v_add_u32_e32 %[sc], -1, %[sc]\n"                                                                                         \
        "v_add_u32_e32 %[iter], 1, %[iter]\n"                                                                                      \
        "s_branch 1b\n"                                                                                                            \
        "9:\n"                                                                                                                     \
        "s_waitcnt lgkmcnt(0)\n"                                                                                                   \
        "s_mov_b64 exec, %[s_entry]\n"
#define VX_DESCEND_OPERANDS                                                                                                                        \
        : [px] "+v"(px), [py] "+v"(py), [pz] "+v"(pz), [tmax] "+v"(tr.t_max), [sc] "+v"(scale), [iter] "+v"(tr.iter), [t0] "=&v"(t0), [t1] "=&v"(t1),  \
          [t2] "=&v"(t2), [crx] "=&v"(crx), [cry] "=&v"(cry), [crz] "=&v"(crz), [tcm] "=&v"(tcm), [hf] "=&v"(hf), [s_entry] "=&s"(s_entry) \
        : [tmin] "v"(tr.t_min), [tcx] "v"(tr.tcx), [tcy] "v"(tr.tcy), [tcz] "v"(tr.tcz), [tbx] "v"(tr.tbx), [tby] "v"(tr.tby), [tbz] "v"(tr.tbz),       \
          [ex] "v"(ex), [ey] "v"(ey), [ez] "v"(ez), [ps] "v"(parent_scale), [lds] "v"(lds_slot0), [k_half] "s"(k_half)                                  \
        : "vcc", "scc", "memory"
    if constexpr (LEVELS == 13) asm volatile(VX_DESCEND_ASM(3328) VX_DESCEND_OPERANDS);
    else asm volatile(VX_DESCEND_ASM(4096) VX_DESCEND_OPERANDS);
#undef VX_DESCEND_ASM
#undef VX_DESCEND_OPERANDS
    tr.px = __uint_as_float(px); tr.py = __uint_as_float(py); tr.pz = __uint_as_float(pz);
    tr.scale = int(scale);
    tr.scale_exp2 = pow2i(tr.scale - kMaxScale);
}

}  // namespace vxd
