#!/usr/bin/env python3
"""DEV: does alternating two streams for consecutive frames raise throughput of small (band-sized) launches?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import capi, scenegen
w, h, objs, lights = scenegen.config("C3")
m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
ctx = capi.Context(0); ctx.scene_upload(objs, lights)
for rows in (270, 540, 1080, 2160):
    for nstreams in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        outs = [torch.zeros((rows, w, 4), dtype=torch.uint8, device="cuda") for _ in range(nstreams)]
        n = 400
        for k in range(20):
            ctx.render_rows_device(m, w, h, 945 if rows < h else 0, (945 if rows < h else 0) + rows, 1, 0, outs[k % nstreams].data_ptr(), 0, streams[k % nstreams].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            ctx.render_rows_device(m, w, h, 945 if rows < h else 0, (945 if rows < h else 0) + rows, 1, 0, outs[k % nstreams].data_ptr(), 0, streams[k % nstreams].cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("rows=%4d streams=%d: %.1f us/frame" % (rows, nstreams, dt * 1e6))
