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

// The trait methods themselves have to live in the trait definitions (a foreign module cannot add a method to `dyn Object`), and
// `Ambient.color` is private to light/ambient.rs, so these are the lines to paste there:
//
//   object/mod.rs, in `pub trait Object`:            fn flatten(&self) -> Option<FlatObject> { None }
//   object/mod.rs, in each blanket impl (&T, Box<T>, Arc<T>):   #[inline] fn flatten(&self) -> Option<FlatObject> { T::flatten(self) }
//   object/sphere.rs, in `impl Object for Sphere`:   #[inline] fn flatten(&self) -> Option<FlatObject> { Some(self.flat()) }
//   light/mod.rs, in `pub trait Light`:              fn flatten(&self) -> Option<FlatLight> { None }
//   light/mod.rs, in each blanket impl:              #[inline] fn flatten(&self) -> Option<FlatLight> { T::flatten(self) }
//   light/point.rs, in `impl Light for Point`:       #[inline] fn flatten(&self) -> Option<FlatLight> { Some(self.flat()) }
//   light/ambient.rs, in `impl Light for Ambient`:   #[inline] fn flatten(&self) -> Option<FlatLight> {
//                                                        Some(FlatLight { kind: RT_AMBIENT, point: [0.0; 3], color: self.color.to_array(), intensity: 0.0 }) }
//
// The flat forms of the types whose fields are public are real items here:

impl crate::object::Sphere {
    /// object/sphere.rs:6-9
    #[inline]
    pub fn flat(&self) -> FlatObject {
        FlatObject { kind: RT_SPHERE, center: self.center.to_array(), radius: self.radius, normal: [0.0; 3] }
    }

    /// The sphere after an object `Transform` (math/mod.rs:5-51): centre through `Transform::apply` (math/mod.rs:40-43), radius times
    /// the scale — defined only for a uniform scale (anything else is an ellipsoid, which `Sphere` cannot express): `None` then.
    /// Nothing in the reference attaches a Transform to an object; this is the host-side pre-transform of SURVEY 8(f) row 4.
    pub fn transformed(&self, t: crate::math::Transform) -> Option<Self> {
        let s = t.scale.to_array();
        if !(s[0] == s[1] && s[1] == s[2]) {
            return None;
        }
        Some(Self::new(t.apply(self.center), self.radius * s[0].abs()))
    }
}

impl crate::light::Point {
    /// light/point.rs:5-9
    #[inline]
    pub fn flat(&self) -> FlatLight {
        FlatLight { kind: RT_POINT, point: self.point.to_array(), color: self.color.to_array(), intensity: self.intensity }
    }
}

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
