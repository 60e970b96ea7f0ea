"""Host-side mirror of the reference's scene / renderer interface (same names, argument meaning and error behaviour).

Reference (Rust, /root/reference/src): math/vec3.rs (Vec3), object/sphere.rs (Sphere), light/point.rs (Point),
light/ambient.rs (Ambient), element.rs (Element, Material), display.rs (Camera, Framebuffer), renderer.rs (Renderer).
Only the per-pixel loop of `Renderer::render` runs on the GPU; everything here is plumbing that flattens the scene
into the SoA buffers of include/rt_mi355x.h.  No arithmetic of the path is done in Python.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import capi


@dataclass(frozen=True)
class Vec3:
    """math/vec3.rs:22-30 — three f32 lanes (the reference's 4th lane is always 0)."""
    x: float = 0.0
    y: float = 0.0
    z: float = 0.0

    @staticmethod
    def splat(v: float) -> "Vec3":
        return Vec3(v, v, v)

    def scaled(self, s: float) -> "Vec3":
        """`s * Vec3` (vec3.rs:467-474), rounded to f32 per lane like the reference."""
        s = np.float32(s)
        return Vec3(float(s * np.float32(self.x)), float(s * np.float32(self.y)), float(s * np.float32(self.z)))

    def to_array(self):
        return (self.x, self.y, self.z)


Vec3.ZERO = Vec3(0.0, 0.0, 0.0)


@dataclass(frozen=True)
class Sphere:
    """object/sphere.rs:6-16"""
    center: Vec3
    radius: float

    def flatten(self):
        return (capi.RT_SPHERE, self.center.x, self.center.y, self.center.z, self.radius)

    def transformed(self, t: "Transform") -> "Sphere":
        """The sphere after an object Transform (math/mod.rs:5-51): centre through Transform::apply, radius times the scale.
        Uniform scale only — anything else is an ellipsoid, which Sphere (object/sphere.rs:6-9) cannot express."""
        s = t.scale
        if not (s.x == s.y == s.z):
            raise ValueError("a Sphere can only take a uniform scale")
        c = capi.transform_apply(t.position.to_array(), t.rotation, t.scale.to_array(), self.center.to_array())
        return Sphere(Vec3(float(c[0]), float(c[1]), float(c[2])), float(np.float32(self.radius) * np.float32(abs(s.x))))


@dataclass(frozen=True)
class Transform:
    """math/mod.rs:5-51 — position, rotation (a versor as (r, i, j, k), quat.rs), scale.  The reference never attaches one to an
    object; the mirrors apply it on the host before flattening (Sphere.transformed)."""
    position: Vec3 = Vec3(0.0, 0.0, 0.0)
    rotation: tuple = (1.0, 0.0, 0.0, 0.0)
    scale: Vec3 = Vec3(1.0, 1.0, 1.0)

    def apply(self, v: Vec3) -> Vec3:
        o = capi.transform_apply(self.position.to_array(), self.rotation, self.scale.to_array(), v.to_array())
        return Vec3(float(o[0]), float(o[1]), float(o[2]))


@dataclass(frozen=True)
class Plane:
    """EXTENSION (no reference counterpart): a two-sided infinite plane through `point` with normal `normal`, an `impl Object`
    in the shape of object/sphere.rs:19-45.  The normal is normalised once here in f32 (the role `UnitVec3` plays in the reference:
    Object::normal returns a UnitVec3, object/mod.rs:8): v / sqrt((xx + yy) + (zz + 0)) with true divisions, vec3.rs:195-197."""
    point: Vec3
    normal: Vec3

    def unit_normal(self):
        f = np.float32
        x, y, z = f(self.normal.x), f(self.normal.y), f(self.normal.z)
        n = np.sqrt(f(f(x * x) + f(y * y)) + f(f(z * z) + f(0.0)), dtype=f)
        return float(x / n), float(y / n), float(z / n)

    def flatten(self):
        return (capi.RT_PLANE, self.point.x, self.point.y, self.point.z, 0.0)


@dataclass(frozen=True)
class Material:
    """element.rs:15-19 — only `reflectiveness` is read by the renderer (renderer.rs:66)."""
    color: Vec3
    reflectiveness: Vec3


@dataclass(frozen=True)
class Element:
    """element.rs:10-13"""
    object: object  # Sphere (reference) or Plane (extension)
    material: Material

    @staticmethod
    def new_unzise(obj, material) -> "Element":
        """element.rs:38-40. HEAD's main.rs:39-46 passes a Vec3 here (it does not type-check); the value is the
        reflectiveness, which is what reproduces result.png, so a bare Vec3 is accepted with that meaning."""
        if isinstance(material, Vec3):
            material = Material(material, material)
        return Element(obj, material)


@dataclass(frozen=True)
class Point:
    """light/point.rs:5-21"""
    point: Vec3
    color: Vec3
    intensity: float

    def flatten(self):
        return (capi.RT_POINT, self.point.x, self.point.y, self.point.z, self.color.x, self.color.y, self.color.z, self.intensity)


@dataclass(frozen=True)
class Ambient:
    """light/ambient.rs:5-19"""
    color: Vec3

    def flatten(self):
        return (capi.RT_AMBIENT, 0.0, 0.0, 0.0, self.color.x, self.color.y, self.color.z, 0.0)


def flatten_elements(elements: Sequence[Element]) -> capi.FlatObjects:
    rows, kinds = [], []
    for e in elements:
        fl = getattr(e.object, "flatten", None)
        if fl is None:
            raise TypeError(f"object {type(e.object).__name__} cannot be flattened (Sphere is the reference's primitive, Plane the extension)")
        kind, cx, cy, cz, r = fl()
        rf = e.material.reflectiveness
        nx, ny, nz = e.object.unit_normal() if kind == capi.RT_PLANE else (0.0, 0.0, 0.0)
        rows.append((cx, cy, cz, r, rf.x, rf.y, rf.z, nx, ny, nz))
        kinds.append(kind)
    a = np.asarray(rows, dtype=np.float32).reshape(-1, 10)
    if not any(k == capi.RT_PLANE for k in kinds):  # the reference's case: spheres only, no kind / normal arrays needed
        return capi.FlatObjects(a[:, 0], a[:, 1], a[:, 2], a[:, 3], a[:, 4], a[:, 5], a[:, 6])
    return capi.FlatObjects(a[:, 0], a[:, 1], a[:, 2], a[:, 3], a[:, 4], a[:, 5], a[:, 6], kind=np.asarray(kinds, dtype=np.uint8),
                            nx=a[:, 7], ny=a[:, 8], nz=a[:, 9])


def flatten_lights(lights: Sequence) -> capi.FlatLights:
    rows, kinds = [], []
    for l in lights:
        fl = getattr(l, "flatten", None)
        if fl is None:
            raise TypeError(f"light {type(l).__name__} cannot be flattened")
        k, *vals = fl()
        kinds.append(k)
        rows.append(vals)
    a = np.asarray(rows, dtype=np.float32).reshape(-1, 7)
    return capi.FlatLights(a[:, 0], a[:, 1], a[:, 2], a[:, 3], a[:, 4], a[:, 5], a[:, 6], kind=np.asarray(kinds, dtype=np.uint8))


@dataclass
class Camera:
    """display.rs:8-13,118-123 — default = 60 degrees, 0.01, 1000."""
    fov: float = None
    z_near: float = None
    z_far: float = None

    def __post_init__(self):
        if self.fov is None or self.z_near is None or self.z_far is None:
            f, n, fr = capi.camera_default()
            self.fov = f if self.fov is None else self.fov
            self.z_near = n if self.z_near is None else self.z_near
            self.z_far = fr if self.z_far is None else self.z_far

    def transform(self, aspect_ratio: float) -> np.ndarray:
        """display.rs:30-41"""
        return capi.camera_transform(self.fov, self.z_near, self.z_far, aspect_ratio)


class Framebuffer:
    """display.rs:16-20,46-115 — RGB8, row-major, row 0 = top."""

    def __init__(self, width: int, height: int, camera: Optional[Camera] = None):
        self.camera = camera if camera is not None else Camera()
        self.aspect_ratio = np.float32(width) / np.float32(height)  # display.rs:47
        self.pixels = np.zeros((height, width, 3), dtype=np.uint8)

    def width(self) -> int:
        return self.pixels.shape[1]

    def height(self) -> int:
        return self.pixels.shape[0]

    def clear(self):
        self.pixels.fill(0)  # display.rs:108-110

    def display(self, path: str = "result.png"):
        """display.rs:113-115 — the reference always writes "result.png" in the CWD."""
        from PIL import Image
        Image.fromarray(self.pixels, mode="RGB").save(path, format="PNG")


class Renderer:
    """renderer.rs:12-82. `render(depth)` runs the pixel loop on the GPU, then display() and clear() like the reference."""

    def __init__(self, frame: Framebuffer, elements: Sequence[Element], lights: Sequence, device_id: int = 0,
                 device_ids: Optional[Sequence[int]] = None):
        self.frame = frame
        self.elements = list(elements)
        self.lights = list(lights)
        self._ctx: Optional[capi.Context] = None
        self._device_id = device_id
        self._device_ids = list(device_ids) if device_ids is not None else None  # one row band per entry (rt_ctx_create_multi)
        self.last_frame: Optional[np.ndarray] = None

    def _context(self) -> capi.Context:
        if self._ctx is None:
            self._ctx = capi.Context(self._device_id, device_ids=self._device_ids)
        return self._ctx

    def render(self, depth: int, flags: int = 0, display_path: Optional[str] = "result.png"):
        if depth < 1:
            # renderer.rs:36 `depth - 1` underflows on usize (panic in debug builds)
            raise ValueError("depth must be >= 1")
        ctx = self._context()
        ctx.scene_upload(flatten_elements(self.elements), flatten_lights(self.lights))
        m = self.frame.camera.transform(self.frame.aspect_ratio)
        w, h = self.frame.width(), self.frame.height()
        self.frame.pixels[...] = ctx.render(m, w, h, depth, flags | capi.RT_OUT_RGB8)
        self.last_frame = self.frame.pixels.copy()
        if display_path is not None:
            self.frame.display(display_path)  # renderer.rs:79
        self.frame.clear()  # renderer.rs:80
        return None


def default_scene():
    """The literals of main.rs:35-52 (config 1): 2 spheres, Point(0, white, 0.5), Ambient(0.1 * white)."""
    elements = [
        Element.new_unzise(Sphere(Vec3(1.0, -1.0, -1.0), 0.5), Vec3(1.0, 0.0, 0.0)),
        Element.new_unzise(Sphere(Vec3(1.0, 0.0, -2.0), 0.5), Vec3(0.0, 1.0, 0.0)),
    ]
    lights = [
        Point(Vec3.ZERO, Vec3(1.0, 1.0, 1.0), 0.5),
        Ambient(Vec3(1.0, 1.0, 1.0).scaled(0.1)),
    ]
    return elements, lights
