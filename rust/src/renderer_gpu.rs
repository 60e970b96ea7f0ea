//! `Renderer::render` routed through librt_mi355x.so — replaces the body at renderer.rs:32-82.
//! UNCOMPILED in this environment (no Rust toolchain in the image, see rust/README.md); written as real items, to be dropped into
//! the crate as `src/renderer_gpu.rs` next to `mod ffi; mod flatten;`.  The same host logic is compiled and tested as C++
//! (raytracing_amd/host/raytracing.hpp) and Python (raytracing_amd/scene.py).
//!
//! Changes the crate needs besides this file:
//!   * renderer.rs: rename the existing `pub fn render` to `pub(crate) fn render_cpu` (the CPU closure stays as the fallback for
//!     scenes that cannot be flattened) and make the `frame` field `pub(crate)`;
//!   * display.rs, next to `clear()`:
//!         #[inline] pub(crate) fn pixels_mut(&mut self) -> &mut [u8] { &mut self.pixels }   // ImageBuffer derefs to [u8]
//!         #[inline] pub(crate) fn aspect_ratio(&self) -> f32 { self.aspect_ratio }
//!         #[inline] pub(crate) fn camera(&self) -> &Camera { &self.camera }
//!   * object/mod.rs, light/mod.rs: the defaulted `flatten()` trait methods of flatten.rs.
use std::borrow::Borrow;
use std::ffi::CStr;
use std::ptr;

use crate::element::Element;
use crate::ffi::*;
use crate::flatten::FlatScene;
use crate::light::{DynLight, Light};
use crate::object::{DynObject, Object};
use crate::renderer::Renderer;

fn check(rc: i32, ctx: *const rt_ctx) -> anyhow::Result<()> {
    if rc == RT_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(rt_last_error(ctx)) }.to_string_lossy().into_owned();
    anyhow::bail!("rt_mi355x error {rc}: {msg}")
}

/// Owns one `rt_ctx` (single device, or one row band per entry of a device list).  Not `Sync`: a context is used by one caller at a
/// time, the same contract as `&mut self` on `Renderer::render` (renderer.rs:32).
pub struct GpuContext {
    raw: *mut rt_ctx,
}

impl GpuContext {
    /// `devices = [0]`: one GPU.  `[0, 1, .., 7]`: the frame is cut into row bands (the rayon row split of display.rs:90-93 across
    /// GPUs) and gathered to the first device inside the library (RCCL).
    pub fn new(devices: &[i32]) -> anyhow::Result<Self> {
        anyhow::ensure!(!devices.is_empty(), "at least one device");
        let mut raw: *mut rt_ctx = ptr::null_mut();
        let rc = if devices.len() == 1 {
            unsafe { rt_ctx_create(&mut raw, devices[0]) }
        } else {
            unsafe { rt_ctx_create_multi(&mut raw, devices.as_ptr(), devices.len() as i32) }
        };
        check(rc, ptr::null())?;
        Ok(Self { raw })
    }

    pub fn upload(&mut self, s: &FlatScene) -> anyhow::Result<()> {
        let objects = rt_objects {
            n: s.cx.len() as i32,
            kind: s.okind.as_ptr(),
            cx: s.cx.as_ptr(), cy: s.cy.as_ptr(), cz: s.cz.as_ptr(), radius: s.radius.as_ptr(),
            refl_r: s.refl_r.as_ptr(), refl_g: s.refl_g.as_ptr(), refl_b: s.refl_b.as_ptr(),
            nx: s.nx.as_ptr(), ny: s.ny.as_ptr(), nz: s.nz.as_ptr(),
        };
        let lights = rt_lights {
            n: s.cr.len() as i32,
            kind: s.lkind.as_ptr(),
            px: s.px.as_ptr(), py: s.py.as_ptr(), pz: s.pz.as_ptr(),
            cr: s.cr.as_ptr(), cg: s.cg.as_ptr(), cb: s.cb.as_ptr(), intensity: s.intensity.as_ptr(),
        };
        check(unsafe { rt_scene_upload(self.raw, &objects, &lights) }, self.raw)
    }

    /// Writes `h` rows of `w` RGB8 pixels (the layout of `Framebuffer.pixels`, display.rs:16-20) into `out`.
    /// `flags`: 0 = the reference's per-pixel walk of the element list; `RT_CULL` = same bytes through the cone hierarchy (opt-in);
    /// `RT_SHADOWS` = the occlusion extension.
    pub fn render_rgb8(&mut self, m: &[f32; 16], w: u32, h: u32, depth: u32, flags: u32, out: &mut [u8]) -> anyhow::Result<()> {
        anyhow::ensure!(out.len() == 3 * (w as usize) * (h as usize), "pixel store has the wrong size");
        check(unsafe { rt_render(self.raw, m.as_ptr(), w, h, depth, flags | RT_OUT_RGB8, out.as_mut_ptr()) }, self.raw)
    }
}

impl Drop for GpuContext {
    fn drop(&mut self) {
        unsafe { rt_ctx_destroy(self.raw) };
    }
}

/// Flatten `&[Element<DynObject>]` / `&[DynLight]` in list order; `None` as soon as one entry has no flat form (a user-defined
/// `impl Object` without `flatten()`): the caller then keeps the CPU closure.
pub fn flatten_scene<'a>(elements: &[Element<DynObject<'a>>], lights: &[DynLight<'a>]) -> Option<FlatScene> {
    let mut s = FlatScene::default();
    for e in elements {
        s.push_object(e.object.flatten()?, e.material.reflectiveness.to_array());
    }
    for l in lights {
        s.push_light(Light::flatten(l)?);
    }
    Some(s)
}

impl<'a, E, L> Renderer<E, L>
where
    E: Borrow<[Element<DynObject<'a>>]>,
    L: Borrow<[DynLight<'a>]>,
{
    /// renderer.rs:32-82 with the pixel loop (renderer.rs:40-77 inside Framebuffer::update, display.rs:74-105) on the GPU.
    pub fn render(&mut self, depth: usize) -> anyhow::Result<()>
    where
        E: Send + Sync,
    {
        self.render_on(&[0], depth)
    }

    /// Same over a list of devices (one row band each).
    pub fn render_on(&mut self, devices: &[i32], depth: usize) -> anyhow::Result<()>
    where
        E: Send + Sync,
    {
        self.render_flags(devices, depth, 0)
    }

    /// Same with library flags: `RT_CULL` (same image, primary rays through the cone hierarchy) and / or `RT_SHADOWS` (extension).
    pub fn render_flags(&mut self, devices: &[i32], depth: usize, flags: u32) -> anyhow::Result<()>
    where
        E: Send + Sync,
    {
        anyhow::ensure!(depth >= 1, "depth must be >= 1"); // the original computes `depth - 1` on usize (renderer.rs:36)
        anyhow::ensure!(depth <= u32::MAX as usize, "depth too large");
        let scene = match flatten_scene(self.elements.borrow(), self.lights.borrow()) {
            Some(s) => s,
            None => return self.render_cpu(depth), // the original body, renamed
        };
        // A long-lived Renderer would keep the context in a field; one per call keeps this file free of struct changes.
        let mut ctx = GpuContext::new(devices)?;
        ctx.upload(&scene)?;
        // Camera::transform (display.rs:30-41) stays in Rust; Mat4 is four row Vec4s (mat4.rs:7-12) -> row-major [f32; 16]
        let t = self.frame.camera().transform(self.frame.aspect_ratio());
        let rows = [t.x.to_array(), t.y.to_array(), t.z.to_array(), t.w.to_array()]; // vec4.rs:194
        let mut m = [0f32; 16];
        for (i, r) in rows.iter().enumerate() {
            m[4 * i..4 * i + 4].copy_from_slice(r);
        }
        let (w, h) = (self.frame.width(), self.frame.height());
        // RT_OUT_RGB8 is Framebuffer.pixels' own layout (3 B/pixel, row 0 = top): no conversion pass
        ctx.render_rgb8(&m, w, h, depth as u32, flags, self.frame.pixels_mut())?;
        drop(ctx);
        self.frame.display()?; // renderer.rs:79, unchanged
        self.frame.clear(); // renderer.rs:80, unchanged
        Ok(())
    }
}
