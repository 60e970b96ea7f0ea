"""raytracing_amd — MI355X (gfx950) implementation of the per-pixel render path of Aandreba/raytracing.

The product is ``librt_mi355x.so`` (C ABI in ``include/rt_mi355x.h``, HIP kernels in ``raytracing_amd/csrc``).
This package is the Python host side used by the tests and ``bench.py``: a ctypes binding (``capi``), a mirror of the
reference's scene/renderer interface (``scene``), the deterministic synthetic scene generator (``scenegen``) and the
row-band multi-GPU driver (``bands``).  There is no CPU fallback: without the built library and a gfx950 device every
render call raises.
"""
from . import capi, scene, scenegen, bands  # noqa: F401
from .scene import (Vec3, Sphere, Point, Ambient, Material, Element, Camera, Framebuffer, Renderer)  # noqa: F401

__all__ = ["capi", "scene", "scenegen", "bands", "Vec3", "Sphere", "Point", "Ambient", "Material", "Element", "Camera",
           "Framebuffer", "Renderer"]
