#!/usr/bin/env python3
"""Time one configuration of the render kernel on the GPU: tools/time_config.py C5 [--shadows] [--depth D] [--frames N] [--rows y0:y1]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import capi, scenegen

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--shadows", action="store_true")
ap.add_argument("--depth", type=int, default=1)
ap.add_argument("--frames", type=int, default=5)
ap.add_argument("--rows", default=None)
ap.add_argument("--cull", action="store_true", help="RT_CULL: primary rays through the cone hierarchy")
ap.add_argument("--rgb8", action="store_true", help="RT_OUT_RGB8: 3 B/pixel, the layout of Framebuffer.pixels")
a = ap.parse_args()
w, h, objs, lights = scenegen.config(a.config)
y0, y1 = (0, h) if a.rows is None else map(int, a.rows.split(":"))
m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
import torch
out = torch.zeros(((y1 - y0), w, 4), dtype=torch.uint8, device="cuda")
flags = (capi.RT_SHADOWS if a.shadows else 0) | (capi.RT_OUT_RGB8 if a.rgb8 else 0) | (capi.RT_CULL if a.cull else 0) | capi.RT_TIMED
with capi.Context(0) as ctx:
    ctx.scene_upload(objs, lights)
    import time
    t_end = time.perf_counter() + 0.06  # bring the clocks out of idle first (a cold GPU is ~15 % slower for the first tens of ms)
    while time.perf_counter() < t_end:
        ctx.render_rows_device(m, w, h, y0, y1, a.depth, flags, out.data_ptr())
        ctx.last_kernel_ms()
    ms = []
    for _ in range(a.frames + 1):
        ctx.render_rows_device(m, w, h, y0, y1, a.depth, flags, out.data_ptr())
        ms.append(ctx.last_kernel_ms())
    ms = ms[1:]
    rays = (y1 - y0) * w
    print("%s rows %d:%d depth=%d shadows=%d rgb8=%d: kernel ms min/med = %.3f / %.3f  -> %.3e primary rays/s; lit pixels %.1f%%"
          % (a.config, y0, y1, a.depth, int(a.shadows), int(a.rgb8), min(ms), sorted(ms)[len(ms) // 2], rays / (sorted(ms)[len(ms) // 2] * 1e-3),
             100.0 * float((out[..., :3].sum(dim=-1) > 0).float().mean())))
