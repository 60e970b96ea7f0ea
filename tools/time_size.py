#!/usr/bin/env python3
"""DEV: kernel time of the C3 scene (1024 spheres) at several frame heights: separates per-pixel cost from launch/quantisation effects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import capi, scenegen
w = 3840
_, _, objs, lights = scenegen.config("C3")
with capi.Context(0) as ctx:
    ctx.scene_upload(objs, lights)
    for h in [135, 270, 540, 1080, 2160]:
        m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(2160))  # fixed aspect
        out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        ms = []
        for _ in range(12):
            ctx.render_rows_device(m, w, 2160, 1000 - h // 2 if h < 2160 else 0, (1000 - h // 2 if h < 2160 else 0) + h, 1, capi.RT_TIMED, out.data_ptr())
            ms.append(ctx.last_kernel_ms())
        t = sorted(ms[2:])[len(ms[2:]) // 2]
        wgs = (w // 64) * ((h + 15) // 16)
        print("h=%5d  WGs=%6d  ms=%.3f  ns/pixel=%.4f  TF=%.1f" % (h, wgs, t, t * 1e6 / (w * h), w * h * 1024 * 6 / (t * 1e-3) / 1e12))
