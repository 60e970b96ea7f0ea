#!/usr/bin/env python3
"""Measure the PCIe-inclusive rate of the host-buffer entry point rt_render on config C3 (noted in DESIGN.md; never bench.py's value):
into a reused pageable frame store, and into a page-locked store from rt_host_alloc."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import capi, scenegen

w, h, objs, lights = scenegen.config("C3")
m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))


def timed(ctx, out, n=30, flags=0):
    for _ in range(3):
        ctx.render(m, w, h, 1, flags, out=out)
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.render(m, w, h, 1, flags, out=out)
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
    # RT_OUT_RGB8 = the layout of Framebuffer.pixels (display.rs:16-20), what the Rust adapter asks for: 3/4 of the bytes over PCIe
    pinned3 = ctx.host_frame((h, w, 3))
    dt = timed(ctx, pinned3, flags=capi.RT_OUT_RGB8)
    print("rt_render C3, RGB8 frame store from rt_host_alloc (pinned): %.3f ms/frame = %.3e rays/s (kernels %.3f ms)" % (dt * 1e3, w * h / dt, ctx.last_kernel_ms()))
    assert np.array_equal(pinned3, ref[..., :3])
    # with RT_CULL the render is 0.06 ms: what is left is the copy
    dt = timed(ctx, pinned3, flags=capi.RT_OUT_RGB8 | capi.RT_CULL)
    print("rt_render C3, RGB8 pinned, RT_CULL:                          %.3f ms/frame = %.3e rays/s (kernels %.3f ms)" % (dt * 1e3, w * h / dt, ctx.last_kernel_ms()))
    assert np.array_equal(pinned3, ref[..., :3])
    ctx.host_free(pinned3)
with capi.Context(device_ids=[0, 0, 0, 0]) as ctx:  # the multi-device context (four virtual bands on this GPU): assembled on the root, one copy out
    ctx.scene_upload(objs, lights)
    pinned = ctx.host_frame((h, w, 4))
    dt = timed(ctx, pinned)
    print("rt_render C3, multi-device context (4 virtual bands), pinned: %.3f ms/frame = %.3e rays/s" % (dt * 1e3, w * h / dt))
    assert np.array_equal(pinned, ref)
    ctx.host_free(pinned)
