"""ctypes binding of librt_mi355x.so (include/rt_mi355x.h). Plain pointers and sizes only; no torch types cross."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

RT_OK = 0
RT_ERR_INVALID = -1
RT_ERR_NO_DEVICE = -2
RT_ERR_HIP = -3
RT_ERR_NO_SCENE = -4
RT_ERR_TOO_LARGE = -5
RT_ERR_RCCL = -6

RT_SPHERE = 0
RT_PLANE = 1  # extension, no reference counterpart
RT_POINT = 0
RT_AMBIENT = 1

RT_SHADOWS = 1
RT_OUT_RGB8 = 2
RT_TIMED = 4  # rt_render_rows_device: record the events rt_last_kernel_ms / rt_last_phase_ms read
RT_CULL = 8  # opt-in: primary rays through the cone hierarchy instead of the walk over the whole list (same bytes out)
RT_SHADOW_EPS = 1.0e-3
RT_NOALLOC_SCRATCH_BYTES = 32 << 20

# RT_MI355X_LIB overrides the in-tree library (used to A/B kernel builds on the GPU box)
LIB_PATH = os.environ.get("RT_MI355X_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "librt_mi355x.so")

EXPORTS = (
    "rt_version", "rt_ctx_create", "rt_ctx_destroy", "rt_last_error", "rt_camera_transform", "rt_camera_default",
    "rt_scene_upload", "rt_render", "rt_render_rows", "rt_render_rows_device", "rt_last_kernel_ms",
    "rt_debug_prepare_spheres", "rt_host_alloc", "rt_host_free", "rt_debug_tile_plan", "rt_render_rows_device_noalloc", "rt_last_phase_ms", "rt_ctx_create_multi", "rt_transform_apply",
    "rt_debug_prepare_shadow_cull",
    "rt_debug_prepare_primary_cull",
)

_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)


class RtObjects(C.Structure):
    _fields_ = [("n", C.c_int32), ("kind", _u8p), ("cx", _f32p), ("cy", _f32p), ("cz", _f32p), ("radius", _f32p),
                ("refl_r", _f32p), ("refl_g", _f32p), ("refl_b", _f32p), ("nx", _f32p), ("ny", _f32p), ("nz", _f32p)]


class RtLights(C.Structure):
    _fields_ = [("n", C.c_int32), ("kind", _u8p), ("px", _f32p), ("py", _f32p), ("pz", _f32p), ("cr", _f32p),
                ("cg", _f32p), ("cb", _f32p), ("intensity", _f32p)]


class RtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rt_mi355x error {code}: {msg}")
        self.code = code


_lib = None


def load() -> C.CDLL:
    """Load the shared library (fails loudly if it has not been built: `make -C raytracing_amd/csrc`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} is missing — build it with `make -C raytracing_amd/csrc` "
                                "(or __graft_entry__.build()); there is no CPU fallback")
    # If torch is (or will be) in the process, its bundled libamdhip64 must be the one HIP runtime: import it first so
    # the loader resolves our DT_NEEDED libamdhip64.so.7 to the copy already mapped.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the plain C ABI
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    lib.rt_version.restype = C.c_char_p
    lib.rt_last_error.restype = C.c_char_p
    lib.rt_last_error.argtypes = [vp]
    lib.rt_ctx_create.argtypes = [C.POINTER(vp), C.c_int32]
    lib.rt_ctx_create_multi.argtypes = [C.POINTER(vp), C.POINTER(C.c_int32), C.c_int32]
    lib.rt_ctx_destroy.argtypes = [vp]
    lib.rt_camera_transform.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, _f32p]
    lib.rt_camera_default.argtypes = [_f32p, _f32p, _f32p]
    lib.rt_transform_apply.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p]
    lib.rt_scene_upload.argtypes = [vp, C.POINTER(RtObjects), C.POINTER(RtLights)]
    lib.rt_debug_tile_plan.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.rt_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.rt_host_free.argtypes = [vp, vp]
    lib.rt_render.argtypes = [vp, _f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    lib.rt_render_rows.argtypes = [vp, _f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp]
    lib.rt_render_rows_device.argtypes = [vp, _f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]
    lib.rt_render_rows_device_noalloc.argtypes = lib.rt_render_rows_device.argtypes
    lib.rt_last_kernel_ms.argtypes = [vp, _f32p]
    lib.rt_last_phase_ms.argtypes = [vp, _f32p]
    lib.rt_debug_prepare_spheres.argtypes = [C.POINTER(RtObjects), _f32p, _u8p]
    _i32p = C.POINTER(C.c_int32)
    lib.rt_debug_prepare_shadow_cull.argtypes = [C.POINTER(RtObjects), C.POINTER(RtLights), C.c_int32, _i32p, _f32p, _f32p, _f32p, _i32p, _i32p]
    lib.rt_debug_prepare_primary_cull.argtypes = [C.POINTER(RtObjects), _i32p, _f32p, _f32p, _f32p]
    for name in EXPORTS:
        if name not in ("rt_version", "rt_last_error"):
            getattr(lib, name).restype = C.c_int32
    _lib = lib
    return lib


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: Optional[np.ndarray], ty):
    return a.ctypes.data_as(ty) if a is not None and a.size else C.cast(None, ty)


class FlatObjects:
    """SoA view of `&[Element<DynObject>]` (element.rs:10-19, object/sphere.rs:6-9); keeps the numpy arrays alive."""

    def __init__(self, cx, cy, cz, radius, refl_r, refl_g, refl_b, kind=None, nx=None, ny=None, nz=None):
        self.cx, self.cy, self.cz, self.radius = _f32(cx), _f32(cy), _f32(cz), _f32(radius)
        self.refl_r, self.refl_g, self.refl_b = _f32(refl_r), _f32(refl_g), _f32(refl_b)
        n = self.cx.size
        for a in (self.cy, self.cz, self.radius, self.refl_r, self.refl_g, self.refl_b):
            if a.size != n:
                raise ValueError("object arrays differ in length")
        self.kind = None if kind is None else np.ascontiguousarray(kind, dtype=np.uint8)
        # RT_PLANE (extension): unit normals, read only for plane elements
        self.nx, self.ny, self.nz = (None if a is None else _f32(a) for a in (nx, ny, nz))
        for a in (self.nx, self.ny, self.nz):
            if a is not None and a.size != n:
                raise ValueError("normal arrays differ in length from the object arrays")
        self.n = n

    def c_struct(self) -> RtObjects:
        return RtObjects(self.n, _ptr(self.kind, _u8p), _ptr(self.cx, _f32p), _ptr(self.cy, _f32p), _ptr(self.cz, _f32p),
                         _ptr(self.radius, _f32p), _ptr(self.refl_r, _f32p), _ptr(self.refl_g, _f32p), _ptr(self.refl_b, _f32p),
                         _ptr(self.nx, _f32p), _ptr(self.ny, _f32p), _ptr(self.nz, _f32p))


class FlatLights:
    """SoA view of `&[DynLight]` (light/point.rs:5-9, light/ambient.rs:5-7)."""

    def __init__(self, px, py, pz, cr, cg, cb, intensity, kind=None):
        self.px, self.py, self.pz = _f32(px), _f32(py), _f32(pz)
        self.cr, self.cg, self.cb, self.intensity = _f32(cr), _f32(cg), _f32(cb), _f32(intensity)
        n = self.cr.size
        for a in (self.px, self.py, self.pz, self.cg, self.cb, self.intensity):
            if a.size != n:
                raise ValueError("light arrays differ in length")
        self.kind = None if kind is None else np.ascontiguousarray(kind, dtype=np.uint8)
        self.n = n

    def c_struct(self) -> RtLights:
        return RtLights(self.n, _ptr(self.kind, _u8p), _ptr(self.px, _f32p), _ptr(self.py, _f32p), _ptr(self.pz, _f32p),
                        _ptr(self.cr, _f32p), _ptr(self.cg, _f32p), _ptr(self.cb, _f32p), _ptr(self.intensity, _f32p))


def tile_plan(w: int, rows: int, slots: int) -> dict:
    """Host-only: the tile queue launch_render builds for a rows x w band on a device with `slots` resident workgroups."""
    plan = (C.c_uint32 * 6)()
    rc = load().rt_debug_tile_plan(w, rows, slots, plan)
    if rc != RT_OK:
        raise RtError(rc, "rt_debug_tile_plan")
    return dict(zip(("strips", "rows4", "rows2", "tiles4", "tiles2", "tiles1"), (int(v) for v in plan)))


def camera_transform(fov: float, z_near: float, z_far: float, aspect: float) -> np.ndarray:
    """Camera::transform (display.rs:30-41) -> row-major float32[16]."""
    m = np.zeros(16, dtype=np.float32)
    rc = load().rt_camera_transform(fov, z_near, z_far, aspect, m.ctypes.data_as(_f32p))
    if rc != RT_OK:
        raise RtError(rc, "rt_camera_transform")
    return m


def transform_apply(position, versor, scale, v) -> np.ndarray:
    """Transform::apply (math/mod.rs:40-43): position + rotate(versor = (r, i, j, k), v) * scale, in the library's f32 arithmetic."""
    a = [_f32(x) for x in (position, versor, scale, v)]
    if [x.size for x in a] != [3, 4, 3, 3]:
        raise ValueError("position[3], versor[4], scale[3], v[3]")
    out = np.zeros(3, dtype=np.float32)
    rc = load().rt_transform_apply(*[x.ctypes.data_as(_f32p) for x in a], out.ctypes.data_as(_f32p))
    if rc != RT_OK:
        raise RtError(rc, "rt_transform_apply")
    return out


def camera_default():
    """Camera::default() (display.rs:118-123) -> (fov, z_near, z_far) as float32."""
    a, b, c = C.c_float(), C.c_float(), C.c_float()
    rc = load().rt_camera_default(C.byref(a), C.byref(b), C.byref(c))
    if rc != RT_OK:
        raise RtError(rc, "rt_camera_default")
    return np.float32(a.value), np.float32(b.value), np.float32(c.value)


def debug_prepare_spheres(objs: FlatObjects):
    filt = np.zeros((objs.n, 4), dtype=np.float32)
    cls = np.zeros(objs.n, dtype=np.uint8)
    s = objs.c_struct()
    rc = load().rt_debug_prepare_spheres(C.byref(s), filt.ctypes.data_as(_f32p), cls.ctypes.data_as(_u8p))
    if rc != RT_OK:
        raise RtError(rc, "rt_debug_prepare_spheres")
    return filt, cls


def debug_prepare_shadow_cull(objs: FlatObjects, lights: FlatLights, light: int) -> dict:
    """Host-only: the any-hit hierarchy of RT_SHADOWS for one point light (rt_debug_prepare_shadow_cull)."""
    n_pad = max(16, (objs.n + 15) // 16 * 16)
    order = np.zeros(n_pad, dtype=np.int32)
    lfilt = np.zeros((n_pad, 4), dtype=np.float32)
    cl_a = np.zeros((n_pad // 16, 4), dtype=np.float32)
    cl_b = np.zeros((n_pad // 16, 4), dtype=np.float32)
    always = np.zeros(max(1, objs.n), dtype=np.int32)
    n_always = C.c_int32()
    o, l = objs.c_struct(), lights.c_struct()
    i32p = C.POINTER(C.c_int32)
    rc = load().rt_debug_prepare_shadow_cull(C.byref(o), C.byref(l), light, order.ctypes.data_as(i32p), lfilt.ctypes.data_as(_f32p),
                                             cl_a.ctypes.data_as(_f32p), cl_b.ctypes.data_as(_f32p), always.ctypes.data_as(i32p), C.byref(n_always))
    if rc != RT_OK:
        raise RtError(rc, "rt_debug_prepare_shadow_cull")
    return {"order": order, "lfilt": lfilt, "cl_a": cl_a, "cl_b": cl_b, "always": always[: n_always.value]}


def debug_prepare_primary_cull(objs: FlatObjects) -> dict:
    """Host-only: the cone hierarchy of RT_CULL for the primary rays (rt_debug_prepare_primary_cull)."""
    n_pad = max(16, (objs.n + 15) // 16 * 16)
    order = np.zeros(n_pad, dtype=np.int32)
    pfilt = np.zeros((n_pad, 4), dtype=np.float32)
    cl_a = np.zeros((n_pad // 16, 4), dtype=np.float32)
    cl_b = np.zeros((n_pad // 16, 4), dtype=np.float32)
    o = objs.c_struct()
    rc = load().rt_debug_prepare_primary_cull(C.byref(o), order.ctypes.data_as(C.POINTER(C.c_int32)), pfilt.ctypes.data_as(_f32p),
                                              cl_a.ctypes.data_as(_f32p), cl_b.ctypes.data_as(_f32p))
    if rc != RT_OK:
        raise RtError(rc, "rt_debug_prepare_primary_cull")
    return {"order": order, "pfilt": pfilt, "cl_a": cl_a, "cl_b": cl_b}


class Context:
    """One rt_ctx == one device. Not thread-safe (mirrors `&mut self`, renderer.rs:32)."""

    def __init__(self, device_id: int = 0, device_ids: Optional[Sequence[int]] = None):
        """Context(d): one device.  Context(device_ids=[...]): rt_ctx_create_multi — one row band per entry, gathered to device_ids[0]
        inside the library (RCCL); a device may repeat (virtual bands)."""
        self._lib = load()
        self._h = C.c_void_p()
        if device_ids is not None:
            ids = (C.c_int32 * len(device_ids))(*[int(d) for d in device_ids])
            rc = self._lib.rt_ctx_create_multi(C.byref(self._h), ids, len(device_ids))
            device_id = int(device_ids[0]) if len(device_ids) else -1
        else:
            rc = self._lib.rt_ctx_create(C.byref(self._h), device_id)
        if rc != RT_OK:
            raise RtError(rc, (self._lib.rt_last_error(None) or b"").decode())
        self.device_id = device_id
        self.device_ids = list(device_ids) if device_ids is not None else [device_id]
        self._pinned = {}

    def _check(self, rc: int):
        if rc != RT_OK:
            raise RtError(rc, (self._lib.rt_last_error(self._h) or b"").decode())

    def close(self):
        if self._h:
            self._lib.rt_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def scene_upload(self, objs: FlatObjects, lights: FlatLights):
        o, l = objs.c_struct(), lights.c_struct()
        self._check(self._lib.rt_scene_upload(self._h, C.byref(o), C.byref(l)))

    def render(self, m: Sequence[float], w: int, h: int, depth: int = 1, flags: int = 0, out: Optional[np.ndarray] = None) -> np.ndarray:
        return self.render_rows(m, w, h, 0, h, depth, flags, out=out)[0]

    def host_frame(self, shape) -> np.ndarray:
        """A page-locked uint8 frame store (rt_host_alloc) as a numpy array; render(..., out=frame) then copies at PCIe rate.
        Release it with host_free(frame) before closing the context."""
        n = int(np.prod(shape))
        p = C.c_void_p()
        self._check(self._lib.rt_host_alloc(self._h, n, C.byref(p)))
        arr = np.ctypeslib.as_array((C.c_uint8 * n).from_address(p.value)).reshape(shape)
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr: np.ndarray) -> None:
        p = self._pinned.pop(arr.ctypes.data)
        self._check(self._lib.rt_host_free(self._h, C.c_void_p(p)))

    def render_rows(self, m, w: int, h: int, y0: int, y1: int, depth: int = 1, flags: int = 0, want_hit_idx: bool = False,
                    out: Optional[np.ndarray] = None):
        m = _f32(m)
        if m.size != 16:
            raise ValueError("camera matrix must have 16 entries")
        ch = 3 if (flags & RT_OUT_RGB8) else 4
        rows = max(0, y1 - y0)
        if out is None:
            out = np.zeros((rows, w, ch), dtype=np.uint8)
        elif out.dtype != np.uint8 or out.shape != (rows, w, ch) or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous uint8 array of shape {(rows, w, ch)}")
        hit = np.full((rows, w), -2, dtype=np.int32) if want_hit_idx else None
        self._check(self._lib.rt_render_rows(self._h, m.ctypes.data_as(_f32p), w, h, y0, y1, depth, flags,
                                             out.ctypes.data_as(C.c_void_p), hit.ctypes.data_as(C.c_void_p) if hit is not None else None))
        return out, hit

    def render_rows_device(self, m, w: int, h: int, y0: int, y1: int, depth: int, flags: int, d_out: int, d_hit_idx: int = 0, stream: int = 0):
        """Enqueue on `stream` (hipStream_t as int; 0 = the context's own stream); d_out/d_hit_idx are device addresses."""
        m = _f32(m)
        self._check(self._lib.rt_render_rows_device(self._h, m.ctypes.data_as(_f32p), w, h, y0, y1, depth, flags,
                                                    C.c_void_p(d_out), C.c_void_p(d_hit_idx) if d_hit_idx else None,
                                                    C.c_void_p(stream) if stream else None))

    def render_rows_device_noalloc(self, m, w: int, h: int, y0: int, y1: int, depth: int, flags: int, d_out: int, d_hit_idx: int = 0, stream: int = 0):
        """rt_render_rows_device_noalloc: same result, scratch in the library's fixed per-GPU array, no allocation in the call."""
        m = _f32(m)
        self._check(self._lib.rt_render_rows_device_noalloc(self._h, m.ctypes.data_as(_f32p), w, h, y0, y1, depth, flags,
                                                            C.c_void_p(d_out), C.c_void_p(d_hit_idx) if d_hit_idx else None,
                                                            C.c_void_p(stream) if stream else None))

    def last_phase_ms(self):
        """(primary/emit, shadow mask, shade + bounce trace) kernel ms of the most recent general launch."""
        ms = (C.c_float * 3)()
        self._check(self._lib.rt_last_phase_ms(self._h, ms))
        return float(ms[0]), float(ms[1]), float(ms[2])

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        self._check(self._lib.rt_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)
