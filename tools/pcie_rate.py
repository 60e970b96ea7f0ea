#!/usr/bin/env python3
"""Measure the PCIe-inclusive rate of the host-buffer entry point rt_render on config C3 (noted in DESIGN.md; never bench.py's value)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import capi, scenegen

w, h, objs, lights = scenegen.config("C3")
m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
with capi.Context(0) as ctx:
    ctx.scene_upload(objs, lights)
    for _ in range(3):
        ctx.render(m, w, h)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        ctx.render(m, w, h)
    dt = (time.perf_counter() - t0) / n
    print("rt_render C3 host-buffer (kernel + D2H of %.1f MB into pageable numpy memory): %.3f ms/frame = %.3e rays/s; kernel alone %.3f ms"
          % (w * h * 4 / 1e6, dt * 1e3, w * h / dt, ctx.last_kernel_ms()))
