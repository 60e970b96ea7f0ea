#!/usr/bin/env python3
"""DEV: fixed per-launch cost: kernel time vs sphere count for a 270-row band and the full frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import capi, scenegen
w, h = 3840, 2160
m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
with capi.Context(0) as ctx:
    for n in (16, 128, 256, 512, 1024, 2048):
        objs, lights = scenegen.generate(w, h, n, 4)
        ctx.scene_upload(objs, lights)
        res = []
        for rows in (270, 2160):
            out = torch.zeros((rows, w, 4), dtype=torch.uint8, device="cuda")
            ms = []
            for _ in range(12):
                ctx.render_rows_device(m, w, h, 945 if rows < h else 0, (945 if rows < h else 0) + rows, 1, capi.RT_TIMED, out.data_ptr())
                ms.append(ctx.last_kernel_ms())
            res.append(sorted(ms[2:])[5])
        print("N=%5d  band270 %.4f ms   full %.4f ms" % (n, res[0], res[1]))
