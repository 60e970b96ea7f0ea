//! `flatten()` for the scene traits. UNCOMPILED in this environment — see rust/README.md.
//!
//! A `Box<dyn Object>` cannot be downcast, so each trait gets one defaulted method; the blanket impls for
//! `&T`, `Box<T>`, `Arc<T>` (object/mod.rs:12-46, light/mod.rs:12-31) forward it like the existing methods.
use crate::ffi::{RT_AMBIENT, RT_POINT, RT_SPHERE};

#[derive(Debug, Clone, Copy, PartialEq)]
pub struct FlatObject {
    pub kind: u8,
    pub center: [f32; 3],
    pub radius: f32,
    /// RT_PLANE only (extension): unit normal; [0.0; 3] for spheres
    pub normal: [f32; 3],
}

#[derive(Debug, Clone, Copy, PartialEq)]
pub struct FlatLight {
    pub kind: u8,
    pub point: [f32; 3],
    pub color: [f32; 3],
    pub intensity: f32,
}

// --- object/mod.rs: add to `pub trait Object` ---------------------------------------------------------
//     fn flatten(&self) -> Option<FlatObject> { None }
// and to each blanket impl (`&T`, `Box<T>`, `Arc<T>`):
//     #[inline] fn flatten(&self) -> Option<FlatObject> { T::flatten(self) }

// --- object/sphere.rs: add to `impl Object for Sphere` -------------------------------------------------
//     #[inline]
//     fn flatten(&self) -> Option<FlatObject> {
//         Some(FlatObject { kind: RT_SPHERE, center: self.center.to_array(), radius: self.radius, normal: [0.0; 3] })
//     }

// --- light/mod.rs: add to `pub trait Light` (+ blanket impls) ------------------------------------------
//     fn flatten(&self) -> Option<FlatLight> { None }

// --- light/point.rs ------------------------------------------------------------------------------------
//     #[inline]
//     fn flatten(&self) -> Option<FlatLight> {
//         Some(FlatLight { kind: RT_POINT, point: self.point.to_array(), color: self.color.to_array(), intensity: self.intensity })
//     }

// --- light/ambient.rs ----------------------------------------------------------------------------------
//     #[inline]
//     fn flatten(&self) -> Option<FlatLight> {
//         Some(FlatLight { kind: RT_AMBIENT, point: [0.0; 3], color: self.color.to_array(), intensity: 0.0 })
//     }

/// SoA buffers handed to `rt_scene_upload`; element / light order is preserved (renderer.rs:49-53, :60-64).
#[derive(Default)]
pub struct FlatScene {
    pub okind: Vec<u8>,
    pub cx: Vec<f32>, pub cy: Vec<f32>, pub cz: Vec<f32>, pub radius: Vec<f32>,
    pub refl_r: Vec<f32>, pub refl_g: Vec<f32>, pub refl_b: Vec<f32>,
    pub nx: Vec<f32>, pub ny: Vec<f32>, pub nz: Vec<f32>,
    pub lkind: Vec<u8>,
    pub px: Vec<f32>, pub py: Vec<f32>, pub pz: Vec<f32>,
    pub cr: Vec<f32>, pub cg: Vec<f32>, pub cb: Vec<f32>, pub intensity: Vec<f32>,
}

impl FlatScene {
    pub fn push_object(&mut self, o: FlatObject, reflectiveness: [f32; 3]) {
        self.okind.push(o.kind);
        self.cx.push(o.center[0]); self.cy.push(o.center[1]); self.cz.push(o.center[2]); self.radius.push(o.radius);
        self.refl_r.push(reflectiveness[0]); self.refl_g.push(reflectiveness[1]); self.refl_b.push(reflectiveness[2]);
        self.nx.push(o.normal[0]); self.ny.push(o.normal[1]); self.nz.push(o.normal[2]);
    }
    pub fn push_light(&mut self, l: FlatLight) {
        self.lkind.push(l.kind);
        self.px.push(l.point[0]); self.py.push(l.point[1]); self.pz.push(l.point[2]);
        self.cr.push(l.color[0]); self.cg.push(l.color[1]); self.cb.push(l.color[2]); self.intensity.push(l.intensity);
    }
}
