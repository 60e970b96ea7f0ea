#!/usr/bin/env python3
"""DEV: RT_CULL against the default path on scenes far beyond the BASELINE sizes (1e5 / 3e5 spheres): whole frame + hit plane equality and times."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raytracing_amd import capi, scenegen
for n, w, h in ((100000, 1280, 720), (300000, 1920, 1080)):
    objs, lights = scenegen.generate(w, h, n, 3, seed=77)
    m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
    with capi.Context(0) as ctx:
        ctx.scene_upload(objs, lights)
        for depth, sh in ((1, 0), (2, capi.RT_SHADOWS)):
            res = []
            for fl in (sh, sh | capi.RT_CULL):
                out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda"); hit = torch.full((h, w), -7, dtype=torch.int32, device="cuda")
                for _ in range(2):
                    ctx.render_rows_device(m, w, h, 0, h, depth, fl | capi.RT_TIMED, out.data_ptr(), hit.data_ptr())
                    ms = ctx.last_kernel_ms()
                res.append((out, hit, ms))
            print(n, w, h, depth, sh, "default %.2f ms, cull %.3f ms, identical %s, hits %d" % (res[0][2], res[1][2], bool(torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])), int((res[0][1] >= 0).sum())), flush=True)
