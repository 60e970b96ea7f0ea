#!/usr/bin/env python3
"""DEV: host-side cost of one bench step (ctypes call + 2 event records) vs kernel time for a 270-row band (what a rank renders at 8 GPUs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import capi, scenegen
w, h, objs, lights = scenegen.config("C3")
m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
ctx = capi.Context(0); ctx.scene_upload(objs, lights)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
for rows in (270, 540, 2160):
    out = torch.zeros((rows, w, 4), dtype=torch.uint8, device="cuda")
    n = 300
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for _ in range(20):
        ctx.render_rows_device(m, w, h, 0, rows, 1, 0, out.data_ptr(), 0, stream.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        evs[k][0].record(stream)
        ctx.render_rows_device(m, w, h, 0, rows, 1, 0, out.data_ptr(), 0, stream.cuda_stream)
        evs[k][1].record(stream)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    km = sum(a.elapsed_time(b) for a, b in evs) / n
    print("rows=%4d: host issue %.1f us/step, wall %.1f us/step, kernel (events) %.1f us" % (rows, t_issue / n * 1e6, t_all / n * 1e6, km * 1e3))
