#!/usr/bin/env python3
"""DEV probe: rt_render_rows_device_noalloc under HIP graph capture (torch.cuda.CUDAGraph): capture one primary-only frame and
general bands, replay each several times into cleared buffers, compare with the eager result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import capi, scenegen

w, h = 1280, 720
objs, lights = scenegen.generate(w, h, 400, 4, seed=3)
m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
st = torch.cuda.Stream()
ok = True
with capi.Context(0) as ctx:
    ctx.scene_upload(objs, lights)
    for depth, flags, rows in ((1, 0, h), (1, capi.RT_SHADOWS, 300), (2, 0, 300), (2, capi.RT_SHADOWS, 300)):
        eager = torch.zeros((rows, w, 4), dtype=torch.uint8, device="cuda")
        ctx.render_rows_device(m, w, h, 0, rows, depth, flags, eager.data_ptr(), 0, st.cuda_stream)
        st.synchronize()
        out = torch.zeros((rows, w, 4), dtype=torch.uint8, device="cuda")
        for rep in range(3):  # eager, same size, three times in a row
            out.zero_()
            torch.cuda.synchronize()
            ctx.render_rows_device_noalloc(m, w, h, 0, rows, depth, flags, out.data_ptr(), 0, st.cuda_stream)
            torch.cuda.synchronize()
            d = int((out != eager).any(dim=-1).sum())
            print("depth %d flags %d EAGER  %d: %d pixels differ" % (depth, flags, rep, d), flush=True)
            ok = ok and d == 0
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            ctx.render_rows_device_noalloc(m, w, h, 0, rows, depth, flags, out.data_ptr(), 0, st.cuda_stream)
        for rep in range(4):
            out.zero_()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            diff = (out != eager).any(dim=-1)
            d = int(diff.sum())
            darker = int((out.int().sum(dim=-1) < eager.int().sum(dim=-1)).sum())
            print("depth %d flags %d REPLAY %d: %d pixels differ (%d darker)" % (depth, flags, rep, d, darker), flush=True)
            ok = ok and d == 0
print("ok" if ok else "MISMATCH")
