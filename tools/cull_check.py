#!/usr/bin/env python3
"""DEV: RT_CULL against the default path on the GPU, whole frames byte for byte (+ hit planes), and the time of both: tools/cull_check.py [C3 C5 ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import capi, scenegen

def run(ctx, m, w, h, depth, flags, frames=6):
    out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    hit = torch.full((h, w), -7, dtype=torch.int32, device="cuda")
    ms = []
    for _ in range(frames):
        ctx.render_rows_device(m, w, h, 0, h, depth, flags | capi.RT_TIMED, out.data_ptr(), hit.data_ptr())
        ms.append(ctx.last_kernel_ms())
    return out, hit, sorted(ms)[len(ms) // 2]

for name in (sys.argv[1:] or ["C3", "C5", "C4"]):
    w, h, objs, lights = scenegen.config(name)
    m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
    with capi.Context(0) as ctx:
        ctx.scene_upload(objs, lights)
        for depth, sh in ((1, 0), (2, 0), (1, capi.RT_SHADOWS)):
            if name == "C4" and (depth > 1 or sh):
                continue
            a, ha, ta = run(ctx, m, w, h, depth, sh)
            b, hb, tb = run(ctx, m, w, h, depth, sh | capi.RT_CULL)
            same = bool(torch.equal(a, b)) and bool(torch.equal(ha, hb))
            print("%s depth=%d shadows=%d: default %.3f ms, RT_CULL %.3f ms (x%.2f), identical=%s, differing pixels %d" % (
                name, depth, int(bool(sh)), ta, tb, ta / tb, same, int((a != b).any(dim=-1).sum())), flush=True)
            if not same:
                sys.exit(1)
