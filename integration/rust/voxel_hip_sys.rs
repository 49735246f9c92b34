//! Raw FFI of `libvoxelhip.so` (include/voxel_hip.h): the MI355X drop-in for the OpenGL side of `graphics::Svo`.
//!
//! Goes to `src/graphics/voxel_hip_sys.rs`. Every item mirrors one declaration of the C header, in the header's order; layouts are
//! `#[repr(C)]` and checked at compile time against the sizes the header's structs have. Link with
//! `println!("cargo:rustc-link-lib=dylib=voxelhip");` (and a `rustc-link-search` to where the library is installed) in `build.rs`.
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_int, c_void};

use crate::graphics::svo_picker::{PickerResult, PickerTask};
use crate::graphics::svo_registry::MaterialInstance;

pub const VX_SVO_ESVO: c_int = 1; // = SvoType::Esvo.shader_type_define (svo.rs:35)
pub const VX_SVO_CSVO: c_int = 2; // = SvoType::Csvo.shader_type_define (svo.rs:36)

pub const VX_OK: c_int = 0;
pub const VX_ERR_INVALID_ARGUMENT: c_int = 1;
pub const VX_ERR_NO_DEVICE: c_int = 2;
pub const VX_ERR_OUT_OF_MEMORY: c_int = 3;
pub const VX_ERR_CAPACITY: c_int = 4;
pub const VX_ERR_HIP: c_int = 5;
pub const VX_ERR_STATE: c_int = 6;

pub const VX_MEM_HOST: i32 = 0;
pub const VX_MEM_DEVICE: i32 = 1;
pub const VX_FORMAT_RGBA32F: i32 = 0;
pub const VX_FORMAT_RGBA8: i32 = 1;
pub const VX_COMM_ID_BYTES: usize = 128;

/// Opaque: replaces `struct Svo`'s GL objects (svo.rs:56-73).
#[repr(C)]
pub struct vx_context {
    _private: [u8; 0],
}

/// One dirty range of the serialized arena: `RangeBuffer::updated_ranges` (internal.rs:151-154,166), in bytes.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct vx_range {
    pub start: u64,
    pub length: u64,
}

/// The uniforms `Svo::render` sets on world.glsl (svo.rs:201-215).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct vx_uniforms {
    pub view: [f32; 16],
    pub fovy: f32,
    pub aspect: f32,
    pub ambient: f32,
    pub light_dir: [f32; 3],
    pub cam_pos: [f32; 3],
    pub render_shadows: i32,
    pub shadow_distance: f32,
    pub highlight_pos: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct vx_hit {
    pub t: f32,
    pub value: u32,
    pub face_id: i32,
    pub flags: u32,
    pub pos: [f32; 3],
    pub lod: f32,
    pub uv: [f32; 2],
    pub shadow_t: f32,
    pub steps: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct vx_stats {
    pub used_bytes: u64,
    pub capacity_bytes: u64,
    pub depth: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct vx_target {
    pub rgba32f: *mut c_void,
    pub hits: *mut vx_hit,
    pub memory: i32,
    pub tile_rank: u32,
    pub tile_count: u32,
    pub format: i32,
}

const _: () = assert!(std::mem::size_of::<vx_range>() == 16);
const _: () = assert!(std::mem::size_of::<vx_uniforms>() == 4 * (16 + 3 + 3 + 3 + 1 + 1 + 3));
const _: () = assert!(std::mem::size_of::<vx_hit>() == 48);
const _: () = assert!(std::mem::size_of::<MaterialInstance>() == 32); // svo_registry.rs:29-40 is #[repr(C)]
const _: () = assert!(std::mem::size_of::<PickerTask>() == 48 && std::mem::size_of::<PickerResult>() == 48); // svo_picker.rs:13-32

extern "C" {
    // ---- lifetime (Svo::new, Drop) ------------------------------------------------------------------------------------
    pub fn vx_create(svo_type: c_int, capacity_bytes: usize, device: c_int, out: *mut *mut vx_context) -> c_int;
    pub fn vx_destroy(ctx: *mut vx_context);
    // ---- resources (VoxelRegistry::build_material_buffer, TextureArrayBuilder::build) ---------------------------------------
    pub fn vx_set_materials(ctx: *mut vx_context, rows: *const MaterialInstance, count: u32) -> c_int;
    pub fn vx_set_textures(ctx: *mut vx_context, rgba8: *const u8, width: u32, height: u32, layers: u32, mip_levels: u32) -> c_int;
    // ---- SVO upload (Svo::update) ---------------------------------------------------------------------------------------
    pub fn vx_staging_ptr(ctx: *mut vx_context) -> *mut u8;
    pub fn vx_capacity(ctx: *const vx_context) -> usize;
    pub fn vx_arena_capacity(ctx: *const vx_context) -> usize;
    pub fn vx_commit(ctx: *mut vx_context, depth: u32, ranges: *const vx_range, count: u32, used_bytes: u64) -> c_int;
    pub fn vx_commit_all(ctx: *mut vx_context, depth: u32, used_bytes: u64) -> c_int;
    /// VX_COMMIT_INLINE = 0, VX_COMMIT_PIPELINED = 1 (a worker thread of the context does the image update and the uploads)
    pub fn vx_set_commit_mode(ctx: *mut vx_context, mode: c_int) -> c_int;
    pub fn vx_commit_wait(ctx: *mut vx_context) -> c_int;
    pub fn vx_get_stats(ctx: *const vx_context, out: *mut vx_stats) -> c_int;
    // ---- the hot path (Svo::render, Svo::raycast) ---------------------------------------------------------------------------
    pub fn vx_render(ctx: *mut vx_context, uniforms: *const vx_uniforms, width: u32, height: u32, target: *const vx_target) -> c_int;
    pub fn vx_raycast(ctx: *mut vx_context, tasks: *const PickerTask, count: u32, results: *mut PickerResult) -> c_int;
    pub fn vx_sync(ctx: *mut vx_context) -> c_int;
    pub fn vx_set_frames_in_flight(ctx: *mut vx_context, frames: c_int) -> c_int;
    pub fn vx_wait_event(ctx: *mut vx_context, hip_event: *mut c_void) -> c_int;
    pub fn vx_stream_wait_render(ctx: *mut vx_context, stream: *mut c_void) -> c_int;
    // ---- pipelined presentation (render + blit_to_default, world.rs:269-283) ------------------------------------------------
    pub fn vx_present_begin(ctx: *mut vx_context, uniforms: *const vx_uniforms, width: u32, height: u32, format: c_int, out_slot: *mut c_int) -> c_int;
    pub fn vx_present_wait(ctx: *mut vx_context, slot: c_int, pixels: *mut *const c_void, bytes: *mut usize) -> c_int;
    // ---- multi-GPU: one process per GPU, screen tiles, RCCL gather --------------------------------------------------------------
    pub fn vx_tile_order(width: u32, height: u32, out: *mut u32, capacity: u32) -> u32;
    pub fn vx_local_tile_count(width: u32, height: u32, tile_rank: u32, tile_count: u32) -> u32;
    pub fn vx_comm_library(path: *const c_char) -> c_int;
    pub fn vx_comm_unique_id(out_id: *mut c_void, bytes: usize) -> c_int;
    pub fn vx_comm_init(ctx: *mut vx_context, nranks: c_int, rank: c_int, unique_id: *const c_void) -> c_int;
    pub fn vx_comm_destroy(ctx: *mut vx_context) -> c_int;
    pub fn vx_comm_info(ctx: *const vx_context, nranks: *mut c_int, rank: *mut c_int) -> c_int;
    pub fn vx_set_comm_headroom(ctx: *mut vx_context, waves_per_cu: c_int) -> c_int;
    pub fn vx_gather_tiles(ctx: *mut vx_context, tiles: *const c_void, bytes_per_rank: u64, gathered: *mut c_void, root: c_int, out_ticket: *mut c_int) -> c_int;
    pub fn vx_wait_gather(ctx: *mut vx_context, ticket: c_int) -> c_int;
    pub fn vx_gather_query(ctx: *mut vx_context, ticket: c_int) -> c_int;
    pub fn vx_render_gather(ctx: *mut vx_context, uniforms: *const vx_uniforms, width: u32, height: u32, target: *const vx_target, bytes_per_rank: u64,
                            gathered: *mut c_void, root: c_int, image: *mut c_void, wait_ticket: c_int, out_ticket: *mut c_int) -> c_int;
    pub fn vx_comm_stream(ctx: *mut vx_context) -> *mut c_void;
    pub fn vx_assemble_tiles(ctx: *mut vx_context, tiles: *const f32, stride_floats: u64, tile_count: u32, width: u32, height: u32, out_rgba32f: *mut f32) -> c_int;
    pub fn vx_assemble_tiles_format(ctx: *mut vx_context, tiles: *const c_void, stride_pixels: u64, tile_count: u32, width: u32, height: u32, out: *mut c_void,
                                    format: c_int, stream: *mut c_void) -> c_int;
    pub fn vx_resolve_2x2(ctx: *mut vx_context, src_rgba32f: *const f32, width: u32, height: u32, dst_rgba32f: *mut f32, stream: *mut c_void) -> c_int;
    // ---- diagnostics ------------------------------------------------------------------------------------------------------
    pub fn vx_image_info(ctx: *const vx_context, out: *mut [u64; 4]) -> c_int;
    pub fn vx_stream(ctx: *mut vx_context) -> *mut c_void;
    pub fn vx_device(ctx: *const vx_context) -> c_int;
    pub fn vx_last_error() -> *const c_char;
    pub fn vx_version() -> *const c_char;
}

/// `vx_last_error()` as a `String`.
pub fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(vx_last_error()) }.to_string_lossy().into_owned()
}
