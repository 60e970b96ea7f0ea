//! FFI binding of include/rt_mi355x.h (librt_mi355x.so). UNCOMPILED in this environment — see rust/README.md.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const RT_OK: i32 = 0;
pub const RT_SPHERE: u8 = 0;
pub const RT_PLANE: u8 = 1; // extension, no counterpart in the reference
pub const RT_POINT: u8 = 0;
pub const RT_AMBIENT: u8 = 1;
pub const RT_SHADOWS: u32 = 1;
pub const RT_OUT_RGB8: u32 = 2;
pub const RT_TIMED: u32 = 4; // rt_render_rows_device: record the events rt_last_kernel_ms / rt_last_phase_ms read
pub const RT_CULL: u32 = 8; // opt-in, same bytes out: primary rays through the cone hierarchy instead of the walk over the whole list

#[repr(C)]
pub struct rt_objects {
    pub n: i32,
    pub kind: *const u8,
    pub cx: *const f32,
    pub cy: *const f32,
    pub cz: *const f32,
    pub radius: *const f32,
    pub refl_r: *const f32,
    pub refl_g: *const f32,
    pub refl_b: *const f32,
    /// RT_PLANE only (extension): unit normal; may be null when no element is a plane
    pub nx: *const f32,
    pub ny: *const f32,
    pub nz: *const f32,
}

#[repr(C)]
pub struct rt_lights {
    pub n: i32,
    pub kind: *const u8,
    pub px: *const f32,
    pub py: *const f32,
    pub pz: *const f32,
    pub cr: *const f32,
    pub cg: *const f32,
    pub cb: *const f32,
    pub intensity: *const f32,
}

#[repr(C)]
pub struct rt_ctx {
    _private: [u8; 0],
}

#[link(name = "rt_mi355x")]
extern "C" {
    pub fn rt_version() -> *const c_char;
    pub fn rt_ctx_create(out: *mut *mut rt_ctx, device_id: i32) -> i32;
    /// One context over a device list: every entry renders one row band, gathered to device_ids[0] by RCCL inside the library.
    pub fn rt_ctx_create_multi(out: *mut *mut rt_ctx, device_ids: *const i32, n_devices: i32) -> i32;
    pub fn rt_ctx_destroy(ctx: *mut rt_ctx) -> i32;
    pub fn rt_last_error(ctx: *const rt_ctx) -> *const c_char;
    pub fn rt_camera_transform(fov: f32, z_near: f32, z_far: f32, aspect_ratio: f32, m: *mut f32) -> i32;
    pub fn rt_camera_default(fov: *mut f32, z_near: *mut f32, z_far: *mut f32) -> i32;
    /// Transform::apply (math/mod.rs:40-43) in the library's arithmetic; a Rust host calls `Transform::apply` itself instead.
    pub fn rt_transform_apply(position: *const f32, versor: *const f32, scale: *const f32, v: *const f32, out: *mut f32) -> i32;
    pub fn rt_scene_upload(ctx: *mut rt_ctx, objects: *const rt_objects, lights: *const rt_lights) -> i32;
    pub fn rt_render(ctx: *mut rt_ctx, m: *const f32, w: u32, h: u32, depth: u32, flags: u32, out_host: *mut u8) -> i32;
    pub fn rt_render_rows(ctx: *mut rt_ctx, m: *const f32, w: u32, h: u32, y0: u32, y1: u32, depth: u32, flags: u32,
                          out_host: *mut u8, hit_idx_host: *mut i32) -> i32;
    pub fn rt_render_rows_device(ctx: *mut rt_ctx, m: *const f32, w: u32, h: u32, y0: u32, y1: u32, depth: u32, flags: u32,
                                 d_out: *mut c_void, d_hit_idx: *mut c_void, stream: *mut c_void) -> i32;
    /// No counterpart in the reference: same as rt_render_rows_device but allocates nothing (scratch = a fixed per-GPU array of
    /// RT_NOALLOC_SCRATCH_BYTES = 32 MiB); for callers that cannot call hipMalloc (graph capture, serving loops). RT_ERR_TOO_LARGE beyond.
    pub fn rt_render_rows_device_noalloc(ctx: *mut rt_ctx, m: *const f32, w: u32, h: u32, y0: u32, y1: u32, depth: u32, flags: u32,
                                 d_out: *mut c_void, d_hit_idx: *mut c_void, stream: *mut c_void) -> i32;
    pub fn rt_host_alloc(ctx: *mut rt_ctx, bytes: usize, out: *mut *mut c_void) -> i32;
    pub fn rt_host_free(ctx: *mut rt_ctx, host_ptr: *mut c_void) -> i32;
    pub fn rt_last_phase_ms(ctx: *mut rt_ctx, ms: *mut f32) -> i32; // ms[3]: emit, shadow mask, shade + bounce trace (general launches)
    pub fn rt_last_kernel_ms(ctx: *mut rt_ctx, ms: *mut f32) -> i32;
    pub fn rt_debug_tile_plan(w: u32, rows: u32, slots: u32, plan: *mut u32) -> i32;
    pub fn rt_debug_prepare_spheres(objects: *const rt_objects, filt: *mut f32, cls: *mut u8) -> i32;
    pub fn rt_debug_prepare_shadow_cull(objects: *const rt_objects, lights: *const rt_lights, light: i32, order: *mut i32, lfilt: *mut f32,
                                        cl_a: *mut f32, cl_b: *mut f32, always: *mut i32, n_always: *mut i32) -> i32;
    pub fn rt_debug_prepare_primary_cull(objects: *const rt_objects, order: *mut i32, pfilt: *mut f32, cl_a: *mut f32, cl_b: *mut f32) -> i32;
}
