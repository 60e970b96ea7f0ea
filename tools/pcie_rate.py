#!/usr/bin/env python3
"""Measure the PCIe-inclusive rate of the host-buffer entry point rt_render on config C3 (noted in DESIGN.md; never bench.py's value):
into a reused pageable frame store, and into a page-locked store from rt_host_alloc."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import capi, scenegen

w, h, objs, lights = scenegen.config("C3")
m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))


def timed(ctx, out, n=30):
    for _ in range(3):
        ctx.render(m, w, h, out=out)
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.render(m, w, h, out=out)
    return (time.perf_counter() - t0) / n


with capi.Context(0) as ctx:
    ctx.scene_upload(objs, lights)
    frame = np.zeros((h, w, 4), np.uint8)
    dt = timed(ctx, frame)
    print("rt_render C3, %.1f MB frame into pageable memory:    %.3f ms/frame = %.3e rays/s (kernels %.3f ms)" % (frame.nbytes / 1e6, dt * 1e3, w * h / dt, ctx.last_kernel_ms()))
    ref = frame.copy()
    pinned = ctx.host_frame((h, w, 4))
    dt = timed(ctx, pinned)
    print("rt_render C3, frame store from rt_host_alloc (pinned):     %.3f ms/frame = %.3e rays/s (kernels %.3f ms)" % (dt * 1e3, w * h / dt, ctx.last_kernel_ms()))
    assert np.array_equal(pinned, ref)
    ctx.host_free(pinned)
