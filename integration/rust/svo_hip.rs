//! `graphics::Svo` on an MI355X: the same public surface as `src/graphics/svo.rs` (new / update / get_stats / render / raycast /
//! reload_resources / bind_buffers_globally), implemented over `libvoxelhip.so` instead of OpenGL compute shaders.
//!
//! Goes to `src/graphics/svo_hip.rs`; `src/graphics/mod.rs` picks it with a cargo feature so that `systems::worldsvo::Svo`,
//! `systems::physics` and the game keep compiling against `graphics::Svo` unchanged:
//!
//! ```ignore
//! #[cfg(feature = "hip")]      mod voxel_hip_sys;
//! #[cfg(feature = "hip")]      #[path = "svo_hip.rs"] pub mod svo;
//! #[cfg(not(feature = "hip"))] pub mod svo;
//! ```
//!
//! What stays exactly as it is: the serializers (`world::hds`), the mapper (`systems::worldsvo`), `PickerBatch` and its task
//! tables, `VoxelRegistry`, `Framebuffer` (the render target is read from / written to host memory here; see `render`).
//! What is needed besides: `WorldSvo::updated_ranges` (worldsvo_updated_ranges.diff) and two accessors on `VoxelRegistry` that
//! hand out what `build_texture_array` / `build_material_buffer` (svo_registry.rs:122-165) compute -- `texture_layers()` and
//! `material_rows()` below are written against those.
use std::ops::Deref;
use std::os::raw::c_int;
use std::ptr;

use cgmath::{EuclideanSpace, Matrix4, Point3, SquareMatrix, Vector3};

use crate::graphics::framebuffer::Framebuffer;
use crate::graphics::svo_picker::{PickerBatch, PickerBatchResult, PickerResult, PickerTask};
use crate::graphics::svo_registry::VoxelRegistry;
use crate::graphics::voxel_hip_sys::*;
use crate::world::hds::WorldSvo;

#[derive(Debug, Copy, Clone)]
pub enum SvoType {
    Esvo,
    Csvo,
}

#[derive(Debug, Copy, Clone)]
pub struct SvoTypeProperties {
    pub name: &'static str,
    pub shader_type_define: &'static str,
}

impl Deref for SvoType {
    type Target = SvoTypeProperties;

    fn deref(&self) -> &Self::Target {
        match self {
            Self::Esvo => &SvoTypeProperties { name: "ESVO", shader_type_define: "1" },
            Self::Csvo => &SvoTypeProperties { name: "CSVO", shader_type_define: "2" },
        }
    }
}

#[derive(Clone, Copy, Debug)]
pub struct Stats {
    pub used_bytes: usize,
    pub capacity_bytes: usize,
    pub depth: u8,
}

pub struct RenderParams {
    pub ambient_intensity: f32,
    pub light_dir: Vector3<f32>,
    pub cam_pos: Point3<f32>,
    pub cam_fwd: Vector3<f32>,
    pub cam_up: Vector3<f32>,
    pub fov_y_rad: f32,
    pub aspect_ratio: f32,
    pub selected_voxel: Option<Point3<f32>>,
    pub render_shadows: bool,
    pub shadow_distance: f32,
}

/// Owns the device context (world buffer, traversal image, textures, materials, streams). Single-threaded like the original,
/// which holds `RefCell`'d GL state.
pub struct Svo {
    ctx: *mut vx_context,
    stats: Stats,
}

fn check(rc: c_int) {
    // the reference panics on shader / texture / capacity errors (svo.rs:112,120,127; esvo.rs:328): so does this
    assert!(rc == VX_OK, "voxelhip: {}", last_error());
}

impl Svo {
    /// svo.rs:109-149. `size_mb` megabytes of world buffer on HIP device 0.
    pub fn new(registry: &VoxelRegistry, typ: SvoType, size_mb: usize) -> Self {
        let svo_type: c_int = typ.shader_type_define.parse().unwrap();
        let mut ctx = ptr::null_mut();
        check(unsafe { vx_create(svo_type, size_mb * 1000 * 1000, 0, &mut ctx) });

        // The layers exactly as TextureArrayBuilder::build uploads them (texture_array.rs:83-153): decoded, flipv()'d, RGBA8, all
        // of one size, in registration order -- which is what `TextureArray::lookup` indexes. 6 mip levels, clamped by the library
        // like texture_array.rs:108 (svo_registry.rs:126).
        let (width, height, layers, rgba8) = registry.texture_layers().unwrap();
        check(unsafe { vx_set_textures(ctx, rgba8.as_ptr(), width, height, layers, 6) });
        // MaterialInstance rows by BlockId (svo_registry.rs:135-165)
        let rows = registry.material_rows();
        check(unsafe { vx_set_materials(ctx, rows.as_ptr(), rows.len() as u32) });

        // Svo::update is called from the frame loop (systems/worldsvo.rs:397-409): let the library's worker thread do the image
        // update and the uploads, so that `update` costs the frame loop what write_changes_to costs. A change shows one frame
        // late at most -- the original's update() + render_fence pair has the same lag.
        check(unsafe { vx_set_commit_mode(ctx, 1) });

        Self { ctx, stats: Stats { used_bytes: 0, capacity_bytes: size_mb * 1000 * 1000, depth: 0 } }
    }

    /// svo.rs:151-156: there is nothing to bind (no global GL buffer bindings on this path).
    pub fn bind_buffers_globally(&self) {}

    /// svo.rs:158-168: shaders are compiled into the library.
    pub fn reload_resources(&mut self) {}

    /// svo.rs:171-189. Writes all changes of `svo` into the pinned staging mirror -- the very call the original makes on the mapped
    /// SSBO -- and hands the library the byte ranges that changed. `vx_commit` stores 2^-depth at byte 0 (:173-175), orders the
    /// upload after the frames in flight like `render_fence.wait()` (:178), and returns without waiting for the device.
    pub fn update<T: WorldSvo<U> + ?Sized, U>(&mut self, svo: &mut T) {
        let ranges: Vec<vx_range> = svo.updated_ranges().iter().map(|r| vx_range { start: r.start as u64, length: r.length as u64 }).collect();
        unsafe {
            let dst = vx_staging_ptr(self.ctx);
            svo.write_changes_to(dst.add(4), vx_arena_capacity(self.ctx), true);
            check(vx_commit(self.ctx, svo.depth() as u32, ranges.as_ptr(), ranges.len() as u32, svo.size_in_bytes() as u64));
        }
        self.stats = Stats { used_bytes: svo.size_in_bytes(), capacity_bytes: unsafe { vx_capacity(self.ctx) }, depth: svo.depth() };
    }

    pub fn get_stats(&self) -> Stats {
        self.stats
    }

    fn uniforms(params: &RenderParams) -> vx_uniforms {
        let view: Matrix4<f32> = Matrix4::look_to_rh(params.cam_pos, params.cam_fwd, params.cam_up).invert().unwrap(); // svo.rs:197
        let view: &[f32; 16] = view.as_ref();
        let highlight = params.selected_voxel.map_or([f32::NAN; 3], |p| [p.x, p.y, p.z]); // svo.rs:211-215
        vx_uniforms {
            view: *view,
            fovy: params.fov_y_rad,
            aspect: params.aspect_ratio,
            ambient: params.ambient_intensity,
            light_dir: params.light_dir.into(),
            cam_pos: params.cam_pos.to_vec().into(),
            render_shadows: params.render_shadows as i32,
            shadow_distance: params.shadow_distance,
            highlight_pos: highlight,
        }
    }

    /// svo.rs:196-229: the frame into `target`'s colour attachment. The pixels arrive in host memory (RGBA32F, row 0 = bottom, the
    /// image2D of world.glsl:10) and are uploaded into the framebuffer's texture, so `blit_to_default` and `as_image` keep working.
    pub fn render(&self, params: &RenderParams, target: &Framebuffer) {
        let (width, height) = (target.width() as u32, target.height() as u32);
        let mut pixels = vec![0f32; (width * height * 4) as usize];
        let u = Self::uniforms(params);
        let t = vx_target { rgba32f: pixels.as_mut_ptr().cast(), hits: ptr::null_mut(), memory: VX_MEM_HOST, tile_rank: 0, tile_count: 1, format: VX_FORMAT_RGBA32F };
        check(unsafe { vx_render(self.ctx, &u, width, height, &t) }); // host target: returns when the image is in place
        unsafe {
            gl::BindTexture(gl::TEXTURE_2D, target.color_attachment());
            gl::TexSubImage2D(gl::TEXTURE_2D, 0, 0, 0, width as i32, height as i32, gl::RGBA, gl::FLOAT, pixels.as_ptr().cast());
            gl::BindTexture(gl::TEXTURE_2D, 0);
        }
    }

    /// The presenting embedder's fast path (no equivalent in the original, which is bound by its fence): begins frame k+1 and returns
    /// frame k as RGBA8 rows top to bottom -- `Framebuffer::as_image`'s bytes (framebuffer.rs:97-111) -- from the library's pinned
    /// ring. `previous` is the slot the last call returned.
    pub fn present(&self, params: &RenderParams, width: u32, height: u32, previous: Option<i32>) -> (i32, Option<&[u8]>) {
        let u = Self::uniforms(params);
        let mut slot: c_int = 0;
        check(unsafe { vx_present_begin(self.ctx, &u, width, height, VX_FORMAT_RGBA8, &mut slot) });
        let frame = previous.map(|p| {
            let (mut pixels, mut bytes) = (ptr::null(), 0usize);
            check(unsafe { vx_present_wait(self.ctx, p, &mut pixels, &mut bytes) });
            unsafe { std::slice::from_raw_parts(pixels.cast::<u8>(), bytes) }
        });
        (slot, frame)
    }

    /// svo.rs:233-255: same task tables (`PickerBatch::serialize_tasks`, svo_picker.rs:63-80), any number of tasks in one launch
    /// (no `MAX_SVO_PICKER_JOBS` cap), synchronous like the original's fence wait (:248-249).
    pub fn raycast(&self, batch: &PickerBatch, result: &mut PickerBatchResult) {
        let mut tasks = vec![PickerTask::default(); batch.task_capacity()];
        let task_count = batch.serialize_tasks(&mut tasks);
        let mut out = vec![PickerResult::default(); task_count];
        check(unsafe { vx_raycast(self.ctx, tasks.as_ptr(), task_count as u32, out.as_mut_ptr()) });
        batch.deserialize_results(&out[..task_count], result);
    }
}

impl Drop for Svo {
    fn drop(&mut self) {
        unsafe { vx_destroy(self.ctx) }
    }
}

// ---- the two accessors `Svo::new` needs from `VoxelRegistry` (add to src/graphics/svo_registry.rs) ------------------------------------
//
// impl VoxelRegistry {
//     /// (width, height, layers, RGBA8 bytes of all layers): what `TextureArrayBuilder::build` uploads (texture_array.rs:83-153).
//     pub(super) fn texture_layers(&self) -> Result<(u32, u32, u32, Vec<u8>), TextureArrayError> {
//         let mut out = Vec::new();
//         let (mut width, mut height) = (0, 0);
//         for tex in &self.textures {
//             let data = assets::read(&tex.path)?;
//             let image = image::load_from_memory_with_format(&data, ImageFormat::from_path(&tex.path)?)?.flipv();
//             assert!(out.is_empty() || (image.width() == width && image.height() == height), "image does not match base dimensions");
//             (width, height) = (image.width(), image.height());
//             out.extend_from_slice(image.to_rgba8().as_raw());
//         }
//         Ok((width, height, self.textures.len() as u32, out))
//     }
//
//     /// `build_material_buffer`'s rows (svo_registry.rs:135-165) with texture names resolved by registration order
//     /// (`TextureArray::lookup`, texture_array.rs:262-264): missing name -> 0, unset -> -1.
//     pub(super) fn material_rows(&self) -> Vec<MaterialInstance> {
//         let lookup = |name: Option<&String>| name.map_or(-1, |n| self.textures.iter().position(|t| &t.name == n).unwrap_or(0) as i32);
//         let max_block_id = self.materials.iter().map(|e| e.block).max().unwrap();
//         let mut rows = vec![MaterialInstance::default(); max_block_id as usize + 1];
//         for entry in &self.materials {
//             let m = &entry.material;
//             rows[entry.block as usize] = MaterialInstance {
//                 specular_pow: m.specular_pow, specular_strength: m.specular_strength,
//                 tex_top: lookup(m.tex_top.as_ref()), tex_side: lookup(m.tex_side.as_ref()), tex_bottom: lookup(m.tex_bottom.as_ref()),
//                 tex_top_normal: lookup(m.tex_top_normal.as_ref()), tex_side_normal: lookup(m.tex_side_normal.as_ref()),
//                 tex_bottom_normal: lookup(m.tex_bottom_normal.as_ref()),
//             };
//         }
//         rows
//     }
// }
//
// and on `PickerBatch` (src/graphics/svo_picker.rs), next to `serialize_tasks`:
//
//     /// Upper bound of the tasks this batch serializes to: one per ray, `Aabb::task_count()` per box (svo_picker.rs:183-299).
//     pub(super) fn task_capacity(&self) -> usize { self.rays.len() + self.aabbs.iter().map(Aabb::task_count).sum::<usize>() }
