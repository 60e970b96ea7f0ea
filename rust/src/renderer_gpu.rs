//! Replacement body of `Renderer::render` (renderer.rs:32-82). UNCOMPILED in this environment — see rust/README.md.
//!
//! display.rs needs one accessor next to `clear()`:
//!     #[inline] pub(crate) fn pixels_mut(&mut self) -> &mut [u8] { &mut self.pixels }   // ImageBuffer derefs to [u8]
//!     #[inline] pub(crate) fn aspect_ratio(&self) -> f32 { self.aspect_ratio }
use std::ffi::CStr;

use crate::ffi::*;
use crate::flatten::FlatScene;

fn check(rc: i32, ctx: *const rt_ctx) -> anyhow::Result<()> {
    if rc == RT_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(rt_last_error(ctx)) }.to_string_lossy().into_owned();
    anyhow::bail!("rt_mi355x error {rc}: {msg}")
}

// impl<'a, E, L> Renderer<E, L> where E: Borrow<[Element<DynObject<'a>>]>, L: Borrow<[DynLight<'a>]>
//
// pub fn render(&mut self, depth: usize) -> anyhow::Result<()> where E: Send + Sync {
//     anyhow::ensure!(depth >= 1, "depth must be >= 1");            // renderer.rs:36 would underflow
//     let elements: &[Element<DynObject>] = self.elements.borrow();
//     let lights: &[DynLight] = self.lights.borrow();
//
//     // 1. flatten; anything that cannot be flattened keeps the CPU closure of the original render()
//     let mut s = FlatScene::default();
//     for e in elements {
//         match e.object.flatten() {
//             Some(o) => s.push_object(o, e.material.reflectiveness.to_array()),
//             None => return self.render_cpu(depth),                // the original body, renamed
//         }
//     }
//     for l in lights {
//         match Light::flatten(l) { Some(f) => s.push_light(f), None => return self.render_cpu(depth) }
//     }
//
//     // 2. one context per Renderer (cache it in a field in real code), scene upload, render, copy out
//     let mut ctx: *mut rt_ctx = std::ptr::null_mut();
//     check(unsafe { rt_ctx_create(&mut ctx, 0) }, std::ptr::null())?;
//     let objects = rt_objects { n: s.cx.len() as i32, kind: s.okind.as_ptr(), cx: s.cx.as_ptr(), cy: s.cy.as_ptr(), cz: s.cz.as_ptr(),
//                                radius: s.radius.as_ptr(), refl_r: s.refl_r.as_ptr(), refl_g: s.refl_g.as_ptr(), refl_b: s.refl_b.as_ptr(),
//                                nx: s.nx.as_ptr(), ny: s.ny.as_ptr(), nz: s.nz.as_ptr() };
//     let lts = rt_lights { n: s.cr.len() as i32, kind: s.lkind.as_ptr(), px: s.px.as_ptr(), py: s.py.as_ptr(), pz: s.pz.as_ptr(),
//                           cr: s.cr.as_ptr(), cg: s.cg.as_ptr(), cb: s.cb.as_ptr(), intensity: s.intensity.as_ptr() };
//     let res = (|| {
//         check(unsafe { rt_scene_upload(ctx, &objects, &lts) }, ctx)?;
//         let t = self.frame.camera.transform(self.frame.aspect_ratio());          // display.rs:30-41 stays in Rust
//         let m: [f32; 16] = unsafe { core::mem::transmute([t.x.to_inner().to_array(), t.y.to_inner().to_array(),
//                                                            t.z.to_inner().to_array(), t.w.to_inner().to_array()]) };
//         let (w, h) = (self.frame.width(), self.frame.height());
//         // RT_OUT_RGB8 writes Framebuffer.pixels' own layout (3 B/pixel, row 0 = top): no conversion pass
//         check(unsafe { rt_render(ctx, m.as_ptr(), w, h, depth as u32, RT_OUT_RGB8, self.frame.pixels_mut().as_mut_ptr()) }, ctx)
//     })();
//     unsafe { rt_ctx_destroy(ctx) };
//     res?;
//
//     self.frame.display()?;                                           // renderer.rs:79, unchanged
//     self.frame.clear();                                              // renderer.rs:80, unchanged
//     Ok(())
// }
