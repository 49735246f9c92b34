/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE. See svo_oracle.h.
 *
 * Scalar fp32 restatement of the reference's GLSL ray path. Build with -ffp-contract=off (see Makefile):
 * every +,-,*,/, sqrt and fmaf below is a single IEEE-754 binary32 operation, evaluated left to right as written,
 * which is what the HIP kernels are required to reproduce bit for bit.
 *
 * FMA placement. GLSL lets the driver fuse a*b+c. The traversal's plane-distance expressions are written as
 * explicit fmaf() here -- `pos * t_coef - t_bias` and its siblings, which the shader itself describes as "one
 * FMA-operation per axis" (svo.esvo.glsl:97-99), plus `ro + rd * t` -- because with exactly these fused the
 * restatement reproduces the reference's golden vectors BIT FOR BIT (e.g. t = 51.095497, uv = 0.099998474 in
 * cast_inside_outside_all_axes "diagonal pos", svo_shader_tests.rs:445-449); unfused it only lands within the
 * tests' 1e-5. Everything else stays unfused. GLSL built-ins whose precision the GL driver defines (normalize,
 * tan, pow, acos, textureLod) are restated with libm / explicit formulas; the golden vectors pin them to 1e-5
 * (tests/test_oracle_golden.py).
 */
#include "svo_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAX_STEPS 1000           /* svo.esvo.glsl:18 */
#define MAX_SCALE 23             /* svo.esvo.glsl:21 */
#define EPSILON 0.00000011920929f /* svo.esvo.glsl:24 = exp2(-23) */
#define INVALID_PTR 0xffffffffu  /* svo.csvo.glsl:15 */

/* ---------------------------------------------------------------------------------------------------- */
/* GLSL scalar helpers                                                                                   */
/* ---------------------------------------------------------------------------------------------------- */

static inline float gmin(float x, float y) { return y < x ? y : x; } /* GLSL min: y if y < x else x */
static inline float gmax(float x, float y) { return x < y ? y : x; } /* GLSL max: y if x < y else x */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline int32_t f2i(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float i2f(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
static inline int find_msb(uint32_t v) { return v ? 31 - __builtin_clz(v) : -1; }
static inline uint32_t low_bits(int n) { return n >= 32 ? 0xffffffffu : (n <= 0 ? 0u : ((1u << n) - 1u)); } /* bitfieldInsert(0,~0,0,n) */
static inline float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
static inline float smoothstepf(float e0, float e1, float x) {
    float t = gclamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
static inline float pow2i(int e) { return u2f((uint32_t)(e + 127) << 23); } /* exp2(int), exact */

static inline uint32_t word_at(const uint32_t* w, size_t n, size_t i) { return i < n ? w[i] : 0u; }

int or_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------------------------------------------- */
/* software sampler: textureLod(sampler2DArray) with the state of texture_array.rs:200-203               */
/*   MAG NEAREST, MIN LINEAR_MIPMAP_LINEAR, WRAP_S CLAMP_TO_EDGE, WRAP_T left at its default REPEAT      */
/* ---------------------------------------------------------------------------------------------------- */

static inline void texel(const or_textures* t, uint32_t level, uint32_t layer, int32_t x, int32_t y, float out[4]) {
    uint32_t w = t->width >> level, h = t->height >> level;
    if (w == 0) w = 1;
    if (h == 0) h = 1;
    /* S clamps to the edge, T repeats */
    if (x < 0) x = 0;
    if (x > (int32_t)w - 1) x = (int32_t)w - 1;
    y %= (int32_t)h;
    if (y < 0) y += (int32_t)h;
    const uint8_t* p = t->level[level] + (((size_t)layer * h + (uint32_t)y) * w + (uint32_t)x) * 4;
    out[0] = (float)p[0] / 255.0f;
    out[1] = (float)p[1] / 255.0f;
    out[2] = (float)p[2] / 255.0f;
    out[3] = (float)p[3] / 255.0f;
}

static void sample_nearest(const or_textures* t, uint32_t level, uint32_t layer, float u, float v, float out[4]) {
    uint32_t w = t->width >> level, h = t->height >> level;
    if (w == 0) w = 1;
    if (h == 0) h = 1;
    texel(t, level, layer, (int32_t)floorf(u * (float)w), (int32_t)floorf(v * (float)h), out);
}

static void sample_linear(const or_textures* t, uint32_t level, uint32_t layer, float u, float v, float out[4]) {
    uint32_t w = t->width >> level, h = t->height >> level;
    if (w == 0) w = 1;
    if (h == 0) h = 1;
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float ax = x - fx, ay = y - fy;
    int32_t i0 = (int32_t)fx, j0 = (int32_t)fy;
    float c00[4], c10[4], c01[4], c11[4];
    texel(t, level, layer, i0, j0, c00);
    texel(t, level, layer, i0 + 1, j0, c10);
    texel(t, level, layer, i0, j0 + 1, c01);
    texel(t, level, layer, i0 + 1, j0 + 1, c11);
    for (int k = 0; k < 4; ++k) {
        float lo = c00[k] * (1.0f - ax) + c10[k] * ax;
        float hi = c01[k] * (1.0f - ax) + c11[k] * ax;
        out[k] = lo * (1.0f - ay) + hi * ay;
    }
}

void or_texture_lod(const or_textures* t, float u, float v, float layer_f, float lod, float rgba[4]) {
    if (t->levels == 0 || t->layers == 0) {
        rgba[0] = rgba[1] = rgba[2] = rgba[3] = 0.0f;
        return;
    }
    /* array layer: round to nearest, clamp to [0, layers-1] (GL 4.5 spec 8.14.2) */
    float lf = floorf(layer_f + 0.5f);
    uint32_t layer = lf <= 0.0f ? 0u : (lf >= (float)(t->layers - 1) ? t->layers - 1 : (uint32_t)lf);
    float q = (float)(t->levels - 1);
    if (!(lod > 0.0f)) { /* magnification (also NaN): NEAREST on the base level */
        sample_nearest(t, 0, layer, u, v, rgba);
        return;
    }
    float lam = lod > q ? q : lod;
    float fl = floorf(lam);
    uint32_t d1 = (uint32_t)fl;
    uint32_t d2 = d1 + 1 > t->levels - 1 ? t->levels - 1 : d1 + 1;
    float frac = lam - fl;
    float a[4], b[4];
    sample_linear(t, d1, layer, u, v, a);
    sample_linear(t, d2, layer, u, v, b);
    for (int k = 0; k < 4; ++k) rgba[k] = a[k] * (1.0f - frac) + b[k] * frac;
}

size_t or_build_mips(const uint8_t* base, uint32_t w, uint32_t h, uint32_t layers, uint32_t levels, uint8_t* out) {
    /* glGenerateMipmap is driver-defined; a 2x2 box filter with round-to-nearest is the de facto behaviour */
    const uint8_t* src = base;
    uint8_t* dst = out;
    uint32_t sw = w, sh = h;
    for (uint32_t l = 1; l < levels; ++l) {
        uint32_t dw = sw / 2 ? sw / 2 : 1, dh = sh / 2 ? sh / 2 : 1;
        for (uint32_t layer = 0; layer < layers; ++layer)
            for (uint32_t y = 0; y < dh; ++y)
                for (uint32_t x = 0; x < dw; ++x)
                    for (uint32_t c = 0; c < 4; ++c) {
                        uint32_t x0 = 2 * x, x1 = 2 * x + 1 < sw ? 2 * x + 1 : sw - 1;
                        uint32_t y0 = 2 * y, y1 = 2 * y + 1 < sh ? 2 * y + 1 : sh - 1;
                        const uint8_t* s = src + (size_t)layer * sw * sh * 4;
                        uint32_t sum = s[(y0 * sw + x0) * 4 + c] + s[(y0 * sw + x1) * 4 + c] + s[(y1 * sw + x0) * 4 + c] +
                                       s[(y1 * sw + x1) * 4 + c];
                        dst[(((size_t)layer * dh + y) * dw + x) * 4 + c] = (uint8_t)((sum + 2) / 4);
                    }
        src = dst;
        dst += (size_t)layers * dw * dh * 4;
        sw = dw;
        sh = dh;
    }
    return (size_t)(dst - out);
}

/* ---------------------------------------------------------------------------------------------------- */
/* CSVO byte readers, svo.csvo.glsl:25-133. `words` = descriptors[] (after octree_scale and root_ptr).   */
/* ---------------------------------------------------------------------------------------------------- */

uint32_t or_csvo_read_uint(const uint32_t* w, size_t n, uint32_t ptr) {
    uint32_t index = ptr / 4, mod = ptr % 4;
    uint32_t lo = word_at(w, n, index);
    if (mod == 0) return lo; /* lshift would be 32; the GLSL masks the second word away entirely (:29-32) */
    uint32_t hi = word_at(w, n, (size_t)index + 1);
    return (lo >> (mod * 8)) | (hi << ((4 - mod) * 8));
}
static inline uint32_t csvo_read_ushort(const uint32_t* w, size_t n, uint32_t ptr) { return or_csvo_read_uint(w, n, ptr) & 0xffffu; }
static inline uint32_t csvo_read_byte(const uint32_t* w, size_t n, uint32_t ptr) { return (word_at(w, n, ptr / 4) >> ((ptr % 4) * 8)) & 0xffu; }

/* bytes occupied by the pointer table entries selected by a 2-bit-per-child mask: sum of (1<<tag)>>1 */
static inline uint32_t csvo_tag_bytes(uint32_t mask16) {
    uint32_t total = 0;
    for (int i = 0; i < 8; ++i) total += (1u << ((mask16 >> (i * 2)) & 3u)) >> 1;
    return total;
}

static uint32_t csvo_next_ptr(const uint32_t* w, size_t n, uint32_t ptr, uint32_t depth, uint32_t idx, int* crossed,
                              uint32_t* header_bytes, uint32_t* pointer_bytes) {
    *crossed = 0;
    if (depth > 3) { /* internal node, :56-97 */
        *header_bytes = 2;
        uint32_t header = csvo_read_ushort(w, n, ptr);
        uint32_t child = (header >> (idx * 2)) & 3u;
        if (child == 0) return INVALID_PTR;
        uint32_t offset = csvo_tag_bytes(header & ((1u << (idx * 2)) - 1u));
        uint32_t ptr_bytes = csvo_tag_bytes(header);
        uint32_t ptr_offset = or_csvo_read_uint(w, n, ptr + 2 + offset);
        ptr_offset &= low_bits((int)(1u << (child - 1)) * 8);
        *pointer_bytes = (1u << child) >> 1;
        if (ptr_offset & (1u << 31)) {
            *crossed = 1;
            return ptr_offset ^ (1u << 31);
        }
        return ptr + 2 + ptr_bytes + ptr_offset;
    }
    *header_bytes = 1;
    uint32_t header = csvo_read_byte(w, n, ptr);
    if (((header >> idx) & 1u) == 0) return INVALID_PTR;
    uint32_t offset = (uint32_t)__builtin_popcount(header & ((1u << idx) - 1u));
    if (depth == 3) { /* pre-leaf node, :107-112 */
        uint32_t ptr_bytes = (uint32_t)__builtin_popcount(header);
        uint32_t ptr_offset = csvo_read_byte(w, n, ptr + 1 + offset);
        *pointer_bytes = 1;
        return ptr + 1 + ptr_bytes + ptr_offset;
    }
    *pointer_bytes = 0;
    return ptr + 1 + 2 + offset; /* leaf node: header + u16 material offset, :114-115 */
}

uint32_t or_csvo_read_next_ptr(const uint32_t* w, size_t n, uint32_t ptr, uint32_t depth, uint32_t idx, int* crossed) {
    uint32_t hb, pb;
    return csvo_next_ptr(w, n, ptr, depth, idx, crossed, &hb, &pb);
}

uint32_t or_csvo_read_leaf(const uint32_t* w, size_t n, uint32_t material_section_ptr, uint32_t pre_leaf_ptr, uint32_t ptr, uint32_t idx) {
    uint32_t material_section_offset = csvo_read_ushort(w, n, pre_leaf_ptr + 1);
    int leaf_index = (int)(ptr - (pre_leaf_ptr + 3));
    int bit_mark = leaf_index * 8 + (int)idx;
    uint32_t v0 = or_csvo_read_uint(w, n, pre_leaf_ptr + 3) & low_bits(bit_mark < 32 ? bit_mark : 32);
    uint32_t v1 = or_csvo_read_uint(w, n, pre_leaf_ptr + 3 + 4) & low_bits(bit_mark - 32 > 0 ? bit_mark - 32 : 0);
    uint32_t preceding = (uint32_t)__builtin_popcount(v0) + (uint32_t)__builtin_popcount(v1);
    return or_csvo_read_uint(w, n, material_section_ptr + material_section_offset * 4 + preceding * 4);
}

/* ---------------------------------------------------------------------------------------------------- */
/* intersect_octree                                                                                      */
/* ---------------------------------------------------------------------------------------------------- */

static inline void push_frame(or_frame* frames, int max_frames, int* n, float t_min, uint32_t ptr, uint32_t idx, uint32_t p4,
                              int scale, int is_child, int is_leaf, int crossed, uint32_t next_ptr) {
    if (!frames) return;
    if (*n < max_frames) {
        or_frame* f = &frames[*n];
        f->t_min = t_min; f->ptr = ptr; f->idx = idx; f->parent_octant_idx = p4; f->scale = scale;
        f->is_child = is_child; f->is_leaf = is_leaf; f->crossed_boundary = crossed; f->next_ptr = next_ptr;
    }
    ++*n;
}

static inline or_material material_at(const or_scene* s, uint32_t value) {
    or_material zero;
    memset(&zero, 0, sizeof zero); /* out-of-range SSBO reads are undefined in GL; defined as zeros here */
    return value < s->n_materials ? s->materials[value] : zero;
}

void or_intersect(const or_scene* scene, const float ro_in[3], const float rd_in[3], float max_dst, int cast_translucent,
                  or_result* res, or_frame* frames, int max_frames, int* n_frames, or_counters* ctr) {
    const int csvo = scene->svo_type == OR_SVO_CSVO;
    const float octree_scale = u2f(scene->world[0]);
    const uint32_t* desc = scene->world + (csvo ? 2 : 1);
    const size_t n_desc = scene->world_words > (size_t)(csvo ? 2 : 1) ? scene->world_words - (csvo ? 2 : 1) : 0;
    int nf = 0;
    if (ctr) ctr->rays++;

    /* rescale inputs to [0;1], then shift to [1;2) (esvo :52-66) */
    float rox = ro_in[0] * octree_scale, roy = ro_in[1] * octree_scale, roz = ro_in[2] * octree_scale;
    max_dst *= octree_scale;

    memset(res, 0, sizeof *res);
    res->t = -1.0f;

    rox += 1.0f; roy += 1.0f; roz += 1.0f;

    uint32_t ptr = csvo ? scene->world[1] : 0u;
    uint32_t parent_octant_idx = 0;
    int scale = MAX_SCALE - 1;
    float scale_exp2 = 0.5f;

    uint32_t last_leaf_value = 0xffffffffu;
    int adjacent_leaf_count = 0;

    float rdx = rd_in[0], rdy = rd_in[1], rdz = rd_in[2];
    const uint32_t eps_bits = f2u(EPSILON) & 0x7fffffffu;
    if (fabsf(rdx) < EPSILON) rdx = u2f(eps_bits | (f2u(rdx) & 0x80000000u));
    if (fabsf(rdy) < EPSILON) rdy = u2f(eps_bits | (f2u(rdy) & 0x80000000u));
    if (fabsf(rdz) < EPSILON) rdz = u2f(eps_bits | (f2u(rdz) & 0x80000000u));

    float tcx = 1.0f / -fabsf(rdx), tcy = 1.0f / -fabsf(rdy), tcz = 1.0f / -fabsf(rdz);
    float tbx = tcx * rox, tby = tcy * roy, tbz = tcz * roz;

    int octant_mask = 0;
    if (rdx > 0.0f) { octant_mask ^= 1; tbx = fmaf(3.0f, tcx, -tbx); }
    if (rdy > 0.0f) { octant_mask ^= 2; tby = fmaf(3.0f, tcy, -tby); }
    if (rdz > 0.0f) { octant_mask ^= 4; tbz = fmaf(3.0f, tcz, -tbz); }

    float t_min = gmax(gmax(fmaf(2.0f, tcx, -tbx), fmaf(2.0f, tcy, -tby)), fmaf(2.0f, tcz, -tbz));
    t_min = gmax(0.0f, t_min);
    float t_max = gmin(gmin(tcx - tbx, tcy - tby), tcz - tbz);
    float h = t_max;

    int idx = 0;
    float px = 1.0f, py = 1.0f, pz = 1.0f;
    if (t_min < fmaf(1.5f, tcx, -tbx)) { idx ^= 1; px = 1.5f; }
    if (t_min < fmaf(1.5f, tcy, -tby)) { idx ^= 2; py = 1.5f; }
    if (t_min < fmaf(1.5f, tcz, -tbz)) { idx ^= 4; pz = 1.5f; }

    /* CSVO state (csvo :252-258) */
    uint32_t depth = 127u - ((f2u(octree_scale) >> 23) & 0xffu);
    uint32_t material_section_ptr = INVALID_PTR;
    uint32_t pre_leaf_pointer = INVALID_PTR;

    uint32_t ptr_stack[MAX_SCALE + 1];
    uint32_t aux_stack[MAX_SCALE + 1]; /* ESVO: parent_octant_idx, CSVO: depth */
    float t_max_stack[MAX_SCALE + 1];
    memset(ptr_stack, 0, sizeof ptr_stack);
    memset(aux_stack, 0, sizeof aux_stack);
    memset(t_max_stack, 0, sizeof t_max_stack);

    for (int i = 0; i < MAX_STEPS; ++i) {
        if (max_dst >= 0.0f && t_min > max_dst) break;
        if (ctr) ctr->iterations++;

        float tcrx = fmaf(px, tcx, -tbx), tcry = fmaf(py, tcy, -tby), tcrz = fmaf(pz, tcz, -tbz);
        float tc_max = gmin(gmin(tcrx, tcry), tcrz);

        uint32_t octant_idx = (uint32_t)(idx ^ octant_mask);

        int is_child, is_leaf, crossed_boundary = 0;
        uint32_t next_ptr = 0, iter_ptr_bytes = 0;
        if (!csvo) { /* esvo :167-173 */
            uint32_t bit = 1u << octant_idx;
            uint32_t descriptor = word_at(desc, n_desc, (size_t)ptr + parent_octant_idx / 2);
            if (parent_octant_idx % 2 != 0) descriptor >>= 16;
            is_child = (descriptor & (bit << 8)) != 0;
            is_leaf = (descriptor & bit) != 0;
            push_frame(frames, max_frames, &nf, t_min / octree_scale, ptr, octant_idx, parent_octant_idx, scale, is_child, is_leaf, 0, 0);
        } else { /* csvo :276-285 */
            uint32_t hb = 0;
            next_ptr = csvo_next_ptr(desc, n_desc, ptr, depth, octant_idx, &crossed_boundary, &hb, &iter_ptr_bytes);
            is_child = next_ptr != INVALID_PTR;
            is_leaf = is_child && depth < 2;
            if (depth == 2) pre_leaf_pointer = ptr;
            if (ctr) ctr->csvo_header_bytes += hb;
            push_frame(frames, max_frames, &nf, t_min / octree_scale, ptr, octant_idx, depth, scale, is_child, is_leaf, crossed_boundary, next_ptr);
        }

        int advance = 1;
        if (is_child && t_min <= t_max) {
            if (is_leaf && t_min == 0.0f) res->inside_voxel = 1;

            if (is_leaf && t_min > 0.0f) {
                /* phase: HIT (esvo :185-265, csvo :295-371) */
                if (ctr) ctr->leaf_tests++;
                uint32_t value;
                if (!csvo) {
                    uint32_t np = word_at(desc, n_desc, (size_t)ptr + 4 + parent_octant_idx);
                    if (np & (1u << 31)) np = ptr + 4 + parent_octant_idx + (np & 0x7fffffffu);
                    np = np + 4 + octant_idx;
                    value = word_at(desc, n_desc, np);
                } else {
                    value = or_csvo_read_leaf(desc, n_desc, material_section_ptr, pre_leaf_pointer, ptr, octant_idx);
                }

                float tex_corner_x = fmaf(px + scale_exp2, tcx, -tbx);
                float tex_corner_y = fmaf(py + scale_exp2, tcy, -tby);
                float tex_corner_z = fmaf(pz + scale_exp2, tcz, -tbz);
                float tc_min = gmax(gmax(tex_corner_x, tex_corner_y), tex_corner_z);

                float qx = px, qy = py, qz = pz; /* un-mirrored voxel position */
                if (octant_mask & 1) qx = 3.0f - scale_exp2 - qx;
                if (octant_mask & 2) qy = 3.0f - scale_exp2 - qy;
                if (octant_mask & 4) qz = 3.0f - scale_exp2 - qz;

                int face_id;
                float uvx, uvy;
                if (tc_min == tex_corner_x) {
                    face_id = (int)((f2u(rdx) >> 31) & 1u);
                    uvx = (fmaf(rdz, tex_corner_x, roz) - qz) / scale_exp2;
                    uvy = (fmaf(rdy, tex_corner_x, roy) - qy) / scale_exp2;
                    if (rdx > 0.0f) uvx = 1.0f - uvx;
                } else if (tc_min == tex_corner_y) {
                    face_id = 2 | (int)((f2u(rdy) >> 31) & 1u);
                    uvx = (fmaf(rdx, tex_corner_y, rox) - qx) / scale_exp2;
                    uvy = (fmaf(rdz, tex_corner_y, roz) - qz) / scale_exp2;
                    if (rdy > 0.0f) uvy = 1.0f - uvy;
                } else {
                    face_id = 4 | (int)((f2u(rdz) >> 31) & 1u);
                    uvx = (fmaf(rdx, tex_corner_z, rox) - qx) / scale_exp2;
                    uvy = (fmaf(rdy, tex_corner_z, roy) - qy) / scale_exp2;
                    if (rdz < 0.0f) uvx = 1.0f - uvx;
                }

                or_material mat = material_at(scene, value);
                int tex_id = mat.tex_side;
                if (face_id == 3) tex_id = mat.tex_top;
                else if (face_id == 2) tex_id = mat.tex_bottom;

                float dst = t_min / octree_scale;
                float tex_lod = smoothstepf(15.0f, 25.0f, dst) * (dst - 15.0f) * 0.05f;
                if (ctr && tex_lod > 0.0f) ctr->leaf_tests_trilinear++;

                float tex_color[4];
                or_texture_lod(&scene->tex, uvx, uvy, (float)tex_id, tex_lod, tex_color);

                int first_of_kind = adjacent_leaf_count == 0 || value != last_leaf_value;
                if ((tex_color[3] > 0.0f || !cast_translucent) && first_of_kind) {
                    res->t = dst;
                    res->face_id = face_id;
                    res->uv[0] = uvx; res->uv[1] = uvy;
                    res->value = value;
                    memcpy(res->color, tex_color, sizeof tex_color);
                    res->lod = tex_lod;
                    res->pos[0] = gmin(gmax(fmaf(t_min, rdx, rox), qx + EPSILON), qx + scale_exp2 - EPSILON);
                    res->pos[1] = gmin(gmax(fmaf(t_min, rdy, roy), qy + EPSILON), qy + scale_exp2 - EPSILON);
                    res->pos[2] = gmin(gmax(fmaf(t_min, rdz, roz), qz + EPSILON), qz + scale_exp2 - EPSILON);
                    for (int k = 0; k < 3; ++k) {
                        res->pos[k] -= 1.0f;
                        res->pos[k] /= octree_scale;
                    }
                    break;
                }
                ++adjacent_leaf_count;
                last_leaf_value = value;
            } else {
                float half_scale = scale_exp2 * 0.5f;
                float tcenx = fmaf(half_scale, tcx, tcrx), tceny = fmaf(half_scale, tcy, tcry), tcenz = fmaf(half_scale, tcz, tcrz);
                float tv_max = gmin(t_max, tc_max);

                if (t_min <= tv_max) {
                    /* phase: PUSH (esvo :280-311, csvo :387-426) */
                    if (ctr) ctr->pushes++;
                    if (tc_max < h) {
                        ptr_stack[scale] = ptr;
                        aux_stack[scale] = csvo ? depth : parent_octant_idx;
                        t_max_stack[scale] = t_max;
                    }
                    h = tc_max;

                    if (!csvo) {
                        uint32_t np = word_at(desc, n_desc, (size_t)ptr + 4 + parent_octant_idx);
                        if (np & (1u << 31)) np = ptr + 4 + parent_octant_idx + (np & 0x7fffffffu);
                        ptr = np;
                        parent_octant_idx = octant_idx;
                    } else {
                        if (ctr) ctr->csvo_pointer_bytes += iter_ptr_bytes; /* pointer bytes count when the ray descends */
                        --depth;
                        ptr = next_ptr;
                        if (crossed_boundary) {
                            if (ctr) ctr->boundaries++;
                            uint32_t child_lod = csvo_read_byte(desc, n_desc, ptr);
                            uint32_t material_bytes = or_csvo_read_uint(desc, n_desc, ptr + 1);
                            ptr += 5;
                            material_section_ptr = ptr;
                            ptr += material_bytes;
                            depth = child_lod;
                        }
                    }

                    --scale;
                    scale_exp2 = half_scale;

                    idx = 0;
                    if (t_min < tcenx) { idx ^= 1; px += scale_exp2; }
                    if (t_min < tceny) { idx ^= 2; py += scale_exp2; }
                    if (t_min < tcenz) { idx ^= 4; pz += scale_exp2; }

                    t_max = tv_max;
                    advance = 0;
                }
            }
        } else {
            adjacent_leaf_count = 0;
            last_leaf_value = 0xffffffffu;
        }
        if (!advance) continue;

        /* phase: ADVANCE (esvo :324-331) */
        int step_mask = 0;
        if (tc_max >= tcrx) { step_mask ^= 1; px -= scale_exp2; }
        if (tc_max >= tcry) { step_mask ^= 2; py -= scale_exp2; }
        if (tc_max >= tcrz) { step_mask ^= 4; pz -= scale_exp2; }

        t_min = tc_max;
        idx ^= step_mask;

        if ((idx & step_mask) != 0) {
            /* phase: POP (esvo :347-390) */
            uint32_t differing_bits = 0;
            if (step_mask & 1) differing_bits |= f2u(px) ^ f2u(px + scale_exp2);
            if (step_mask & 2) differing_bits |= f2u(py) ^ f2u(py + scale_exp2);
            if (step_mask & 4) differing_bits |= f2u(pz) ^ f2u(pz + scale_exp2);

            scale = find_msb(differing_bits);
            if (scale >= MAX_SCALE || scale < 0) break; /* left the octree (findMSB(0) = -1 cannot index the stacks) */
            scale_exp2 = pow2i(scale - MAX_SCALE);

            ptr = ptr_stack[scale];
            if (csvo) depth = aux_stack[scale]; else parent_octant_idx = aux_stack[scale];
            t_max = t_max_stack[scale];

            int32_t shx = f2i(px) >> scale, shy = f2i(py) >> scale, shz = f2i(pz) >> scale;
            px = i2f(shx << scale);
            py = i2f(shy << scale);
            pz = i2f(shz << scale);
            idx = (shx & 1) | ((shy & 1) << 1) | ((shz & 1) << 2);

            h = 0.0f;
        }
    }
    if (n_frames) *n_frames = nf;
}

/* ---------------------------------------------------------------------------------------------------- */
/* world.glsl                                                                                            */
/* ---------------------------------------------------------------------------------------------------- */

static const float FACE_NORMALS[6][3] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};    /* svo.glsl:2-9 */
static const float FACE_TANGENTS[6][3] = {{0, 0, 1}, {0, 0, -1}, {1, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {1, 0, 0}};    /* svo.glsl:12-19 */
static const float FACE_BITANGENTS[6][3] = {{0, 1, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 1}, {0, 1, 0}, {0, 1, 0}};    /* svo.glsl:22-29 */

static inline float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void normalize3(const float v[3], float out[3]) {
    float len = sqrtf(dot3(v, v));
    out[0] = v[0] / len; out[1] = v[1] / len; out[2] = v[2] / len;
}

void or_primary_ray(const or_uniforms* u, uint32_t w, uint32_t h, uint32_t x, uint32_t y, float ro[3], float rd[3]) {
    /* world.glsl:112-129 */
    float uvx = (float)x / (float)w, uvy = (float)y / (float)h;
    uvx = uvx * 2.0f - 1.0f;
    uvy = uvy * 2.0f - 1.0f;
    uvx *= u->aspect;
    float tan_half = tanf(u->fovy * 0.5f);
    uvx *= tan_half;
    uvy *= tan_half;

    const float* m = u->view;
    /* u_view * vec4(0,0,0,1) and u_view * vec4(uv, -1, 1), rows accumulated left to right */
    float ow = m[3] * 0.0f + m[7] * 0.0f + m[11] * 0.0f + m[15] * 1.0f;
    float o[3], l[3];
    for (int r = 0; r < 3; ++r) o[r] = (m[r] * 0.0f + m[4 + r] * 0.0f + m[8 + r] * 0.0f + m[12 + r] * 1.0f) / ow;
    float lw = m[3] * uvx + m[7] * uvy + m[11] * -1.0f + m[15] * 1.0f;
    for (int r = 0; r < 3; ++r) l[r] = (m[r] * uvx + m[4 + r] * uvy + m[8 + r] * -1.0f + m[12 + r] * 1.0f) / lw;

    float d[3] = {l[0] - o[0], l[1] - o[1], l[2] - o[2]};
    normalize3(d, rd);
    ro[0] = o[0]; ro[1] = o[1]; ro[2] = o[2];
}

static void sky_color(const float rd[3], float out[3]) { /* world.glsl:92-108 */
    const float SKY[3] = {135.0f / 255.0f, 206.0f / 255.0f, 235.0f / 255.0f};
    float HORIZON[3];
    for (int k = 0; k < 3; ++k) HORIZON[k] = 1.0f * (1.0f - 0.3f) + SKY[k] * 0.3f;
    float flat[3] = {rd[0], 0.0f, rd[2]}, p[3];
    normalize3(flat, p);
    /* acos is undefined beyond [-1,1] in GLSL; at the horizon rounding pushes the argument an ulp above 1. The reference's
     * expected image (assets/tests/graphics_svo_render_expected.png) shows the horizon row as plain horizon colour,
     * i.e. its driver returns acos(1+) = 0, which clamping reproduces. */
    float a = acosf(gclamp(dot3(rd, p) / fabsf(sqrtf(dot3(rd, rd))) * fabsf(sqrtf(dot3(p, p))), -1.0f, 1.0f));
    float grad = a / 1.570796f;
    grad = 1.0f - powf(1.0f - grad, 3.0f);
    for (int k = 0; k < 3; ++k) out[k] = HORIZON[k] * (1.0f - grad) + SKY[k] * grad;
}

static void add_counters(or_counters* dst, const or_counters* src) {
    dst->rays += src->rays; dst->iterations += src->iterations; dst->pushes += src->pushes;
    dst->leaf_tests += src->leaf_tests; dst->leaf_tests_trilinear += src->leaf_tests_trilinear;
    dst->boundaries += src->boundaries; dst->csvo_header_bytes += src->csvo_header_bytes;
    dst->csvo_pointer_bytes += src->csvo_pointer_bytes;
}

static void trace_ray(const or_scene* scene, const or_uniforms* u, const float ro[3], const float rd[3], float color[4], int* hit,
                      or_hit* rec, or_counters* ctr) { /* world.glsl:27-90 */
    or_result res;
    or_counters pc; /* this pixel's own step counters: iterations are reported per pixel for parity checks */
    memset(&pc, 0, sizeof pc);
    or_intersect(scene, ro, rd, -1.0f, 1, &res, NULL, 0, NULL, &pc);
    *hit = res.t != -1.0f;
    if (rec) {
        memset(rec, 0, sizeof *rec);
        rec->t = res.t; rec->value = res.value; rec->face_id = res.face_id;
        rec->pos[0] = res.pos[0]; rec->pos[1] = res.pos[1]; rec->pos[2] = res.pos[2];
        rec->lod = res.lod; rec->uv[0] = res.uv[0]; rec->uv[1] = res.uv[1];
        rec->shadow_t = -1.0f;
        if (*hit) rec->flags |= 1u;
    }
    color[0] = color[1] = color[2] = color[3] = 0.0f;
    if (res.t < 0.0f) {
        if (rec) rec->steps = (uint32_t)pc.iterations;
        if (ctr) add_counters(ctr, &pc);
        return;
    }

    if (floorf(res.pos[0]) == floorf(u->highlight_pos[0]) && floorf(res.pos[1]) == floorf(u->highlight_pos[1]) &&
        floorf(res.pos[2]) == floorf(u->highlight_pos[2])) {
        const float thickness = 1.0f / 16.0f;
        float lx = fabsf(res.uv[0] - 0.5f) * 2.0f, ly = fabsf(res.uv[1] - 0.5f) * 2.0f;
        if (gmax(lx, ly) > 1.0f - thickness) {
            color[0] = color[1] = color[2] = color[3] = 1.0f;
            if (rec) { rec->flags |= 8u; rec->steps = (uint32_t)pc.iterations; }
            if (ctr) add_counters(ctr, &pc);
            return;
        }
    }

    or_material mat = material_at(scene, res.value);
    int tex_normal_id = mat.tex_side_normal;
    if (res.face_id == 3) tex_normal_id = mat.tex_top_normal;
    else if (res.face_id == 2) tex_normal_id = mat.tex_bottom_normal;

    float normal[3] = {FACE_NORMALS[res.face_id][0], FACE_NORMALS[res.face_id][1], FACE_NORMALS[res.face_id][2]};
    const float* tangent = FACE_TANGENTS[res.face_id];
    const float* bitangent = FACE_BITANGENTS[res.face_id];

    if (tex_normal_id != -1) {
        float s[4];
        or_texture_lod(&scene->tex, res.uv[0], res.uv[1], (float)tex_normal_id, res.lod, s);
        float tex[3] = {s[0] * 2.0f - 1.0f, s[2] * 2.0f - 1.0f, s[1] * 2.0f - 1.0f}; /* .xzy */
        float n[3];
        normalize3(tex, n);
        const float base[3] = {normal[0], normal[1], normal[2]};
        for (int k = 0; k < 3; ++k) normal[k] = n[0] * tangent[k] + n[1] * base[k] + n[2] * bitangent[k];
    }

    const float neg_l[3] = {-u->light_dir[0], -u->light_dir[1], -u->light_dir[2]};
    float diffuse = gmax(dot3(normal, neg_l), 0.0f);

    float vd[3] = {res.pos[0] - u->cam_pos[0], res.pos[1] - u->cam_pos[1], res.pos[2] - u->cam_pos[2]}, view_dir[3];
    normalize3(vd, view_dir);
    /* reflect(I, N) = I - 2 dot(N, I) N with I = -light_dir */
    float dn = dot3(normal, neg_l);
    float reflect_dir[3];
    for (int k = 0; k < 3; ++k) reflect_dir[k] = neg_l[k] - 2.0f * dn * normal[k];
    float specular = powf(gmax(dot3(view_dir, reflect_dir), 0.0f), mat.specular_pow) * mat.specular_strength;

    float shadow = 1.0f;
    if (u->render_shadows && res.t < u->shadow_distance) {
        float so[3] = {res.pos[0] + normal[0] * 0.001f, res.pos[1] + normal[1] * 0.001f, res.pos[2] + normal[2] * 0.001f};
        or_result sres;
        or_intersect(scene, so, neg_l, -1.0f, 1, &sres, NULL, 0, NULL, &pc);
        shadow = sres.t < 0.0f ? 1.0f : 0.0f;
        if (rec) {
            rec->flags |= 2u;
            if (!(sres.t < 0.0f)) rec->flags |= 4u;
            rec->shadow_t = sres.t;
        }
    }

    float light = gclamp(u->ambient + (diffuse + specular) * shadow, 0.0f, 1.0f);
    color[0] = res.color[0] * light;
    color[1] = res.color[1] * light;
    color[2] = res.color[2] * light;
    color[3] = res.color[3];
    if (rec) rec->steps = (uint32_t)pc.iterations;
    if (ctr) add_counters(ctr, &pc);
}

void or_render(const or_scene* scene, const or_uniforms* u, uint32_t w, uint32_t h, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
               float* out_rgba, or_hit* hits, or_counters* ctr, int n_threads) {
    if (x1 > w) x1 = w;
    if (y1 > h) y1 = h;
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads > 1 ? n_threads : 1)
#endif
    {
        or_counters local;
        memset(&local, 0, sizeof local);
        /* the unit of work is a 32x32 tile of the rectangle ("all host cores ... over 32x32 tiles"), handed out one at a time: rows of
         * the sky cost a fraction of rows of the ground, and 1080 rows in chunks of four were too few units for a hundred threads */
        const int64_t tx0 = x0 / 32, ty0 = y0 / 32, ntx = ((int64_t)x1 + 31) / 32 - tx0, nty = ((int64_t)y1 + 31) / 32 - ty0;
        const int64_t n_tiles = ntx > 0 && nty > 0 ? ntx * nty : 0;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int64_t t = 0; t < n_tiles; ++t) {
            const uint32_t bx = (uint32_t)(tx0 + t % ntx) * 32u, by = (uint32_t)(ty0 + t / ntx) * 32u;
            const uint32_t xa = bx > x0 ? bx : x0, xb = bx + 32u < x1 ? bx + 32u : x1;
            const uint32_t ya = by > y0 ? by : y0, yb = by + 32u < y1 ? by + 32u : y1;
            for (uint32_t y = ya; y < yb; ++y) {
                for (uint32_t x = xa; x < xb; ++x) {
                    float ro[3], rd[3], color[4];
                    int hit = 0;
                    or_primary_ray(u, w, h, x, y, ro, rd);
                    trace_ray(scene, u, ro, rd, color, &hit, hits ? &hits[(size_t)y * w + x] : NULL, ctr ? &local : NULL);
                    if (!hit) {
                        float sky[3];
                        sky_color(rd, sky);
                        color[0] = sky[0]; color[1] = sky[1]; color[2] = sky[2]; color[3] = 1.0f;
                    }
                    memcpy(out_rgba + ((size_t)y * w + x) * 4, color, sizeof color);
                }
            }
        }
        if (ctr) {
#ifdef _OPENMP
#pragma omp critical
#endif
            add_counters(ctr, &local);
        }
    }
}

void or_picker(const or_scene* scene, const or_picker_task* tasks, uint32_t n, or_picker_result* results, int n_threads) {
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads > 1 ? n_threads : 1)
#endif
    for (int64_t i = 0; i < (int64_t)n; ++i) { /* picker.glsl:30-51 */
        or_result res;
        or_intersect(scene, tasks[i].pos, tasks[i].dir, tasks[i].max_dst, 0, &res, NULL, 0, NULL, NULL);
        or_picker_result* r = &results[i];
        memset(r, 0, sizeof *r);
        if (res.t > 0.0f) {
            r->dst = res.t;
            r->inside_voxel = (uint32_t)res.inside_voxel;
            memcpy(r->pos, res.pos, sizeof r->pos);
            memcpy(r->normal, FACE_NORMALS[res.face_id], sizeof r->normal);
        } else {
            r->dst = -1.0f;
        }
    }
}
