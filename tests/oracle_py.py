"""ctypes loader of the CPU oracle (oracle/librt_oracle.so). TEST INFRASTRUCTURE ONLY — never imported by raytracing_amd."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "librt_oracle.so")

_f32p = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)

FLAG_SHADOWS = 1


class RtoObjects(C.Structure):
    _fields_ = [("n", C.c_int32), ("kind", _u8p), ("cx", _f32p), ("cy", _f32p), ("cz", _f32p), ("radius", _f32p),
                ("refl_r", _f32p), ("refl_g", _f32p), ("refl_b", _f32p), ("nx", _f32p), ("ny", _f32p), ("nz", _f32p)]


class RtoLights(C.Structure):
    _fields_ = [("n", C.c_int32), ("kind", _u8p), ("px", _f32p), ("py", _f32p), ("pz", _f32p), ("cr", _f32p),
                ("cg", _f32p), ("cb", _f32p), ("intensity", _f32p)]


def _p(a, ty):
    return a.ctypes.data_as(ty) if a is not None and a.size else C.cast(None, ty)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.rto_default_fov.restype = C.c_float
        lib.rto_camera_transform.argtypes = [C.c_float] * 4 + [_f32p]
        lib.rto_mat4_mul_vec4.argtypes = [_f32p, _f32p, _f32p]
        lib.rto_dot3.restype = C.c_float
        lib.rto_dot3.argtypes = [_f32p, _f32p]
        lib.rto_unit3.argtypes = [_f32p, _f32p]
        lib.rto_sphere_is_hit_by.argtypes = [_f32p, _f32p, _f32p, C.c_float, _f32p]
        lib.rto_sphere_is_hit_by.restype = C.c_int
        lib.rto_plane_is_hit_by.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p]
        lib.rto_plane_is_hit_by.restype = C.c_int
        lib.rto_point_hits.argtypes = [_f32p, _f32p, C.c_float, _f32p, _f32p]
        lib.rto_point_hits.restype = C.c_int
        lib.rto_pixel_position.argtypes = [_f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _f32p]
        lib.rto_quantise.argtypes = [C.c_float]
        lib.rto_quantise.restype = C.c_uint8
        lib.rto_render.argtypes = [C.POINTER(RtoObjects), C.POINTER(RtoLights), _f32p] + [C.c_uint32] * 8 + [C.c_void_p, C.c_void_p, C.c_int]
        lib.rto_render.restype = C.c_int
        lib.rto_default_threads.restype = C.c_int
        lib.rto_transform_apply.argtypes = [_f32p] * 5

    @staticmethod
    def _v(a):
        return np.ascontiguousarray(a, dtype=np.float32)

    def default_fov(self):
        return np.float32(self.lib.rto_default_fov())

    def camera_transform(self, fov, zn, zf, aspect):
        m = np.zeros(16, np.float32)
        self.lib.rto_camera_transform(fov, zn, zf, aspect, m.ctypes.data_as(_f32p))
        return m

    def default_camera(self, w, h):
        return self.camera_transform(self.default_fov(), np.float32(0.01), np.float32(1000.0), np.float32(w) / np.float32(h))

    def mat4_mul_vec4(self, m, v):
        m, v, o = self._v(m), self._v(v), np.zeros(4, np.float32)
        self.lib.rto_mat4_mul_vec4(m.ctypes.data_as(_f32p), v.ctypes.data_as(_f32p), o.ctypes.data_as(_f32p))
        return o

    def unit3(self, v):
        v, o = self._v(v), np.zeros(3, np.float32)
        self.lib.rto_unit3(v.ctypes.data_as(_f32p), o.ctypes.data_as(_f32p))
        return o

    def sphere_is_hit_by(self, origin, direction, center, radius):
        o, d, c = self._v(origin), self._v(direction), self._v(center)
        t = C.c_float()
        hit = self.lib.rto_sphere_is_hit_by(o.ctypes.data_as(_f32p), d.ctypes.data_as(_f32p), c.ctypes.data_as(_f32p), radius, C.byref(t))
        return np.float32(t.value) if hit else None

    def plane_is_hit_by(self, origin, direction, point, normal):
        o, d, p, n = self._v(origin), self._v(direction), self._v(point), self._v(normal)
        t = C.c_float()
        hit = self.lib.rto_plane_is_hit_by(o.ctypes.data_as(_f32p), d.ctypes.data_as(_f32p), p.ctypes.data_as(_f32p), n.ctypes.data_as(_f32p), C.byref(t))
        return np.float32(t.value) if hit else None

    def point_hits(self, lp, lc, intensity, p):
        lp, lc, p, o = self._v(lp), self._v(lc), self._v(p), np.zeros(3, np.float32)
        r = self.lib.rto_point_hits(lp.ctypes.data_as(_f32p), lc.ctypes.data_as(_f32p), intensity, p.ctypes.data_as(_f32p), o.ctypes.data_as(_f32p))
        return o if r else None

    def pixel_position(self, m, j, i, w, h):
        m, o = self._v(m), np.zeros(3, np.float32)
        self.lib.rto_pixel_position(m.ctypes.data_as(_f32p), j, i, w, h, o.ctypes.data_as(_f32p))
        return o

    def quantise(self, c):
        return int(self.lib.rto_quantise(C.c_float(c)))

    def render(self, objs, lights, m, w, h, y0=0, y1=None, row_step=1, depth=1, flags=0, channels=4, want_hit_idx=False, nthreads=0):
        """objs/lights: raytracing_amd.capi.FlatObjects/FlatLights (plain numpy SoA holders)."""
        y1 = h if y1 is None else y1
        m = self._v(m)
        o = RtoObjects(objs.n, _p(objs.kind, _u8p), _p(objs.cx, _f32p), _p(objs.cy, _f32p), _p(objs.cz, _f32p), _p(objs.radius, _f32p),
                       _p(objs.refl_r, _f32p), _p(objs.refl_g, _f32p), _p(objs.refl_b, _f32p),
                       _p(getattr(objs, "nx", None), _f32p), _p(getattr(objs, "ny", None), _f32p), _p(getattr(objs, "nz", None), _f32p))
        l = RtoLights(lights.n, _p(lights.kind, _u8p), _p(lights.px, _f32p), _p(lights.py, _f32p), _p(lights.pz, _f32p),
                      _p(lights.cr, _f32p), _p(lights.cg, _f32p), _p(lights.cb, _f32p), _p(lights.intensity, _f32p))
        out = np.zeros((y1 - y0, w, channels), np.uint8)
        hit = np.full((y1 - y0, w), -2, np.int32) if want_hit_idx else None
        rc = self.lib.rto_render(C.byref(o), C.byref(l), m.ctypes.data_as(_f32p), w, h, y0, y1, row_step, depth, flags, channels,
                                 out.ctypes.data_as(C.c_void_p), hit.ctypes.data_as(C.c_void_p) if hit is not None else None, nthreads)
        if rc != 0:
            raise ValueError("rto_render rejected the arguments")
        return (out, hit) if want_hit_idx else out

    def transform_apply(self, position, versor, scale, v):
        a = [self._v(x) for x in (position, versor, scale, v)]
        out = np.zeros(3, np.float32)
        self.lib.rto_transform_apply(*[_p(x, _f32p) for x in a], _p(out, _f32p))
        return out

    def default_threads(self):
        return int(self.lib.rto_default_threads())


def build():
    subprocess.run(["make", "-C", ORACLE_DIR, "-s"], check=True)


def load() -> Oracle:
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ORACLE_DIR, "rt_oracle.c")):
        build()
    return Oracle(C.CDLL(LIB))
