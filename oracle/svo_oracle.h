/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, scalar fp32, no FMA contraction) of the reference's GLSL ray path:
 *   assets/shaders/svo.esvo.glsl, svo.csvo.glsl, svo.glsl, world.glsl, picker.glsl, svo.test.glsl.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker. Parity status: PINNED against the reference's own golden vectors
 * (src/graphics/svo_shader_tests.rs, src/graphics/svo.rs tests) by tests/test_oracle_golden.py.
 */
#ifndef SVO_ORACLE_H
#define SVO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OR_SVO_ESVO 1 /* SVO_TYPE_ESVO, svo.glsl:65 */
#define OR_SVO_CSVO 2 /* SVO_TYPE_CSVO, svo.glsl:66 */

/* svo.glsl:48-59, 32-byte rows (src/graphics/svo_registry.rs:29-40) */
typedef struct {
    float specular_pow, specular_strength;
    int32_t tex_top, tex_side, tex_bottom;
    int32_t tex_top_normal, tex_side_normal, tex_bottom_normal;
} or_material;

/* RGBA8 2D array texture with a full mip chain; level l is [layers][h>>l][w>>l][4], row 0 = bottom
 * (images are flipped on load, src/graphics/texture_array.rs:92,155-176). */
typedef struct {
    uint32_t width, height, layers, levels;
    const uint8_t* level[16];
} or_textures;

typedef struct {
    int svo_type;
    const uint32_t* world; /* [f32 octree_scale][payload], the mapped buffer of src/graphics/svo.rs:171-189 */
    size_t world_words;    /* readable u32 words in `world` (reads beyond return 0) */
    const or_material* materials;
    uint32_t n_materials;
    or_textures tex;
} or_scene;

/* svo.glsl:31-40 */
typedef struct {
    float t;
    uint32_t value;
    int32_t face_id;
    float pos[3];
    float uv[2];
    float color[4];
    float lod;
    int32_t inside_voxel;
} or_result;

/* svo.test.glsl:23-33; for CSVO the 4th field carries `depth` (svo.csvo.glsl:285) */
typedef struct {
    float t_min;
    uint32_t ptr, idx, parent_octant_idx;
    int32_t scale, is_child, is_leaf, crossed_boundary;
    uint32_t next_ptr;
} or_frame;

/* step counters feeding the algorithmic-bytes model of SURVEY.md §8(d) */
typedef struct {
    uint64_t rays, iterations, pushes, leaf_tests, leaf_tests_trilinear, boundaries;
    uint64_t csvo_header_bytes, csvo_pointer_bytes;
} or_counters;

/* the uniforms of world.glsl:12-25 as set by src/graphics/svo.rs:201-215 */
typedef struct {
    float view[16]; /* u_view, column-major */
    float fovy, aspect;
    float ambient;
    float light_dir[3];
    float cam_pos[3];
    int32_t render_shadows;
    float shadow_distance;
    float highlight_pos[3];
} or_uniforms;

/* per-pixel record of what trace_ray saw (for parity checks) */
typedef struct {
    float t;          /* primary hit distance, -1 = miss */
    uint32_t value;
    int32_t face_id;
    uint32_t flags;   /* bit0 hit, bit1 shadow ray cast, bit2 in shadow, bit3 highlighted outline */
    float pos[3];
    float lod;
    float uv[2];
    float shadow_t;   /* t of the shadow ray's hit, -1 = unoccluded or not cast */
    uint32_t steps;   /* traversal loop iterations, primary + shadow */
} or_hit;

/* picker.glsl:19-35, std430 (48-byte) layouts of src/graphics/svo_picker.rs:13-32 */
typedef struct { float max_dst; float _p0[3]; float pos[3]; float _p1; float dir[3]; float _p2; } or_picker_task;
typedef struct { float dst; uint32_t inside_voxel; float _p0[2]; float pos[3]; float _p1; float normal[3]; float _p2; } or_picker_result;

/* intersect_octree (svo.esvo.glsl:50-393 / svo.csvo.glsl:151-509). frames/ctr may be NULL. */
void or_intersect(const or_scene* scene, const float ro[3], const float rd[3], float max_dst, int cast_translucent,
                  or_result* res, or_frame* frames, int max_frames, int* n_frames, or_counters* ctr);

/* textureLod on the software sampler (sampler state of texture_array.rs:200-203) */
void or_texture_lod(const or_textures* tex, float u, float v, float layer, float lod, float rgba[4]);

/* Box-filter mip chain below `base`; returns bytes written to `out` (levels 1..levels-1, concatenated). */
size_t or_build_mips(const uint8_t* base, uint32_t w, uint32_t h, uint32_t layers, uint32_t levels, uint8_t* out);

/* world.glsl main for the pixel rectangle [x0,x1) x [y0,y1) of a w x h image; out_rgba is w*h*4 floats
 * (row 0 = bottom, as imageStore writes it), hits may be NULL; n_threads <= 1 runs serially. */
void or_render(const or_scene* scene, const or_uniforms* u, uint32_t w, uint32_t h, uint32_t x0, uint32_t y0, uint32_t x1,
               uint32_t y1, float* out_rgba, or_hit* hits, or_counters* ctr, int n_threads);

/* primary ray of pixel (x,y): world.glsl:110-129 */
void or_primary_ray(const or_uniforms* u, uint32_t w, uint32_t h, uint32_t x, uint32_t y, float ro[3], float rd[3]);

/* picker.glsl main over n tasks */
void or_picker(const or_scene* scene, const or_picker_task* tasks, uint32_t n, or_picker_result* results, int n_threads);

/* CSVO byte readers (svo.csvo.glsl:25-133), exposed for the bit-reader known-answer tests */
uint32_t or_csvo_read_uint(const uint32_t* words, size_t n_words, uint32_t byte_ptr);
uint32_t or_csvo_read_next_ptr(const uint32_t* words, size_t n_words, uint32_t ptr, uint32_t depth, uint32_t idx, int* crossed);
uint32_t or_csvo_read_leaf(const uint32_t* words, size_t n_words, uint32_t material_section_ptr, uint32_t pre_leaf_ptr, uint32_t ptr, uint32_t idx);

int or_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
