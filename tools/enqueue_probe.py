#!/usr/bin/env python3
"""DEV: host-side cost of every rt_render_rows_device call in a long back-to-back run of band-sized launches on one stream:
where does the host stall once it runs ahead of the GPU?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import capi, scenegen
w, h, objs, lights = scenegen.config("C3")
rows = int(os.environ.get("ROWS", "270"))
n = int(os.environ.get("N", "600"))
m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
st = torch.cuda.Stream()
with capi.Context(0) as ctx:
    ctx.scene_upload(objs, lights)
    out = torch.zeros((rows, w, 4), dtype=torch.uint8, device="cuda")
    for _ in range(300):
        ctx.render_rows_device(m, w, h, 945, 945 + rows, 1, 0, out.data_ptr(), 0, st.cuda_stream)
    torch.cuda.synchronize()
    t = np.zeros(n + 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    t[0] = time.perf_counter()
    for i in range(n):
        ctx.render_rows_device(m, w, h, 945, 945 + rows, 1, capi.RT_TIMED, out.data_ptr(), 0, st.cuda_stream)
        t[i + 1] = time.perf_counter()
    e1.record(st)
    e1.synchronize()
    d = np.diff(t) * 1e6
    print("rows %d: GPU %.1f us/frame over %d launches; host enqueue us: median %.1f  p90 %.1f  max %.1f; total host %.1f ms"
          % (rows, e0.elapsed_time(e1) * 1e3 / n, n, np.median(d), np.percentile(d, 90), d.max(), (t[-1] - t[0]) * 1e3))
    slow = np.nonzero(d > 5 * np.median(d))[0]
    print("calls slower than 5x the median:", len(slow), "first at", slow[:12].tolist(), "costs", np.round(d[slow[:12]]).tolist())
    print("kernel alone (last launch, own events): %.1f us" % (ctx.last_kernel_ms() * 1e3))
