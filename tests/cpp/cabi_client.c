/* A plain-C client of include/voxel_hip.h: the call sequence a foreign-language binding of `graphics::Svo` makes (what
 * integration/rust/svo_hip.rs does through voxel_hip_sys.rs), with nothing but the header's declarations -- no Python, no C++
 * mirror. It writes a tiny ESVO world by hand (the 12-word octants of src/world/hds/esvo.rs:74-101), commits it, renders a
 * frame, casts picker rays and checks a handful of facts that need no oracle. Exit code 0 = all good; 2 = no HIP device
 * (the library has no CPU path, which is itself checked: vx_create must say VX_ERR_NO_DEVICE); anything else = a failure.
 *
 *   gcc -std=c11 -O1 -Iinclude tests/cpp/cabi_client.c -Lvoxel-rs_amd/lib -lvoxelhip -Wl,-rpath,$PWD/voxel-rs_amd/lib -lm -o cabi_client
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "voxel_hip.h"

#define CHECK(cond)                                                                      \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            fprintf(stderr, "cabi_client: line %d: %s failed (last error: %s)\n", __LINE__, #cond, vx_last_error()); \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

/* An ESVO world of depth 2 (4x4x4 voxels) with ONE voxel of block id 7 at (1, 0, 2), written the way Esvo::write_to writes it
 * (esvo.rs:291-308): [preamble: 5 words][arena]. Octant = 4 header words (per child: child_mask << 8 | leaf_mask, two children per
 * word) + 8 body words (relative pointer with bit 31, or the leaf's value). */
static size_t tiny_world(uint32_t* words) {
    /* arena: octant A (root, at arena word 0), octant B (its child 4 = the 2x2x2 cell at x 0..1, y 0..1, z 2..3, at arena word 12) */
    uint32_t* preamble = words;
    uint32_t* a = words + 5;
    uint32_t* b = a + 12;
    memset(words, 0, (5 + 24) * 4);
    /* B: voxel (1, 0, 2) is child index x + 2y + 4z = 1 of the cell (its local position is (1, 0, 0)): a leaf with value 7 */
    b[4 + 1] = 7;
    /* A: child 4 (local cell (0, 0, 1)) is the octant B, 12 - 4 - 4 words ahead of its body word; B's masks go into A's header */
    a[4 + 4] = 0x80000000u | (12 - 4 - 4);
    a[4 / 2] |= ((1u << 1) << 8 | (1u << 1)) << 0; /* child 4 is even: low half of header word 2; child_mask = leaf_mask = bit 1 */
    /* preamble (esvo.rs:179-188): a fake octant whose child 0 is the root -- its header holds the ROOT's child mask, its body word 0
     * the absolute descriptors[] index of the root octant (= 5, right behind the preamble) */
    preamble[0] = (1u << 4) << 8;
    preamble[4] = 0 + 5;
    return (5 + 24) * 4;
}

int main(void) {
    vx_context* ctx = NULL;
    int rc = vx_create(VX_SVO_ESVO, 1u << 20, 0, &ctx);
    if (rc == VX_ERR_NO_DEVICE) {
        CHECK(ctx == NULL && strstr(vx_last_error(), "no HIP device") != NULL);
        printf("cabi_client: no HIP device: vx_create refuses (there is no CPU path)\n");
        return 2;
    }
    CHECK(rc == VX_OK && ctx != NULL);
    CHECK(strncmp(vx_version(), "voxel-hip", 9) == 0);

    /* Svo::new: one opaque white texture layer, material rows by block id (row 7 = the block used below) */
    uint8_t texel[4 * 4 * 4];
    memset(texel, 255, sizeof texel);
    CHECK(vx_set_textures(ctx, texel, 4, 4, 1, 1) == VX_OK);
    vx_material mats[8];
    memset(mats, 0, sizeof mats);
    for (int i = 0; i < 8; ++i) {
        mats[i].tex_top = mats[i].tex_side = mats[i].tex_bottom = 0;
        mats[i].tex_top_normal = mats[i].tex_side_normal = mats[i].tex_bottom_normal = -1;
    }
    CHECK(vx_set_materials(ctx, mats, 8) == VX_OK);

    /* Svo::update: write the world behind the 4-byte scale in the staging mirror, commit all of it */
    CHECK(vx_render(ctx, NULL, 8, 8, NULL) != VX_OK); /* nothing committed yet, and null arguments: an error code, no crash */
    uint8_t* staging = vx_staging_ptr(ctx);
    CHECK(staging != NULL && vx_capacity(ctx) == (1u << 20) && vx_arena_capacity(ctx) == (1u << 20) - 24);
    const size_t frame_bytes = tiny_world((uint32_t*)(staging + 4));
    CHECK(vx_commit_all(ctx, 2, frame_bytes - 20) == VX_OK);
    vx_stats st;
    CHECK(vx_get_stats(ctx, &st) == VX_OK && st.depth == 2 && st.used_bytes == frame_bytes - 20 && st.capacity_bytes == (1u << 20));
    vx_range too_far = {(1u << 20) - 8, 64};
    CHECK(vx_commit(ctx, 2, &too_far, 1, frame_bytes - 20) == VX_ERR_CAPACITY); /* the reference asserts here (esvo.rs:328) */

    /* Svo::raycast: straight down onto the voxel, and a ray that misses everything */
    vx_picker_task tasks[2];
    vx_picker_result results[2];
    memset(tasks, 0, sizeof tasks);
    tasks[0].max_dst = 10.0f; tasks[0].pos[0] = 1.5f; tasks[0].pos[1] = 3.5f; tasks[0].pos[2] = 2.5f; tasks[0].dir[1] = -1.0f;
    tasks[1].max_dst = 10.0f; tasks[1].pos[0] = 3.5f; tasks[1].pos[1] = 3.5f; tasks[1].pos[2] = 0.5f; tasks[1].dir[1] = -1.0f;
    CHECK(vx_raycast(ctx, tasks, 2, results) == VX_OK);
    CHECK(fabsf(results[0].dst - 2.5f) < 1e-4f && results[0].normal[1] == 1.0f && results[0].inside_voxel == 0);
    CHECK(fabsf(results[0].pos[0] - 1.5f) < 1e-4f && fabsf(results[0].pos[1] - 1.0f) < 1e-3f && fabsf(results[0].pos[2] - 2.5f) < 1e-4f);
    CHECK(results[1].dst == -1.0f);

    /* Svo::render: a camera above the voxel looking straight down (u_view = look_to_rh(eye, -y, +z)^-1, columns [s, u, -f, eye]) */
    vx_uniforms u;
    memset(&u, 0, sizeof u);
    const float s[3] = {-1, 0, 0}, up[3] = {0, 0, 1}, f[3] = {0, -1, 0}, eye[3] = {1.5f, 3.8f, 2.5f};
    for (int i = 0; i < 3; ++i) { u.view[i] = s[i]; u.view[4 + i] = up[i]; u.view[8 + i] = -f[i]; u.view[12 + i] = eye[i]; u.cam_pos[i] = eye[i]; }
    u.view[15] = 1.0f;
    u.fovy = 1.2566371f; u.aspect = 1.0f; u.ambient = 0.3f;
    u.light_dir[0] = u.light_dir[1] = u.light_dir[2] = -0.57735026f;
    u.render_shadows = 1; u.shadow_distance = 500.0f;
    u.highlight_pos[0] = u.highlight_pos[1] = u.highlight_pos[2] = NAN;
    enum { W = 64, H = 64 };
    static float image[W * H * 4];
    static vx_hit hits[W * H];
    vx_target target = {image, hits, VX_MEM_HOST, 0, 1, VX_FORMAT_RGBA32F};
    CHECK(vx_render(ctx, &u, W, H, &target) == VX_OK);
    const vx_hit* centre = &hits[(H / 2) * W + W / 2];
    CHECK((centre->flags & 1u) && centre->value == 7 && centre->face_id == 3 && fabsf(centre->t - 2.8f) < 1e-3f);
    CHECK(!(hits[0].flags & 1u) && image[3] == 1.0f && image[2] > 0.8f); /* a corner pixel sees the sky */
    size_t hit_pixels = 0;
    for (int i = 0; i < W * H; ++i) hit_pixels += hits[i].flags & 1u;
    CHECK(hit_pixels > 50 && hit_pixels < W * H / 4);

    /* the same frame as RGBA8 through the presentation ring: top row first, 4 bytes per pixel */
    int slot = -1;
    const void* pixels = NULL;
    size_t bytes = 0;
    CHECK(vx_present_begin(ctx, &u, W, H, VX_FORMAT_RGBA8, &slot) == VX_OK && slot >= 0);
    CHECK(vx_present_wait(ctx, slot, &pixels, &bytes) == VX_OK && bytes == (size_t)W * H * 4);
    const uint8_t* px = (const uint8_t*)pixels;
    const float* bottom_left = image;                 /* float image: row 0 = bottom */
    const uint8_t* same_pixel = px + (size_t)(H - 1) * W * 4; /* RGBA8 image: row H-1 = bottom */
    for (int c = 0; c < 4; ++c) CHECK(same_pixel[c] == (uint8_t)(fminf(fmaxf(bottom_left[c], 0.0f), 1.0f) * 255.0f + 0.5f));

    /* pipelined commits (what svo_hip.rs switches on): the same world committed again through the worker thread; the frame is the same */
    CHECK(vx_set_commit_mode(ctx, 7) == VX_ERR_INVALID_ARGUMENT);
    CHECK(vx_set_commit_mode(ctx, VX_COMMIT_PIPELINED) == VX_OK);
    CHECK(vx_staging_ptr(ctx) != NULL);
    CHECK(vx_commit_all(ctx, 2, frame_bytes - 20) == VX_OK); /* posted */
    CHECK(vx_commit_wait(ctx) == VX_OK);                     /* queued on the device */
    static float image2[W * H * 4];
    vx_target target2 = {image2, NULL, VX_MEM_HOST, 0, 1, VX_FORMAT_RGBA32F};
    CHECK(vx_render(ctx, &u, W, H, &target2) == VX_OK);
    CHECK(memcmp(image, image2, sizeof image) == 0);
    CHECK(vx_set_commit_mode(ctx, VX_COMMIT_INLINE) == VX_OK);

    CHECK(vx_sync(ctx) == VX_OK);
    vx_destroy(ctx);
    printf("cabi_client: ok (%zu of %d pixels hit the voxel)\n", hit_pixels, W * H);
    return 0;
}
