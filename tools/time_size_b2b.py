#!/usr/bin/env python3
"""DEV: per-frame time of the C3 scene at several frame heights with launches issued BACK TO BACK on one stream (as bench.py does):
separates the per-row cost from whatever is fixed per launch when the chip never idles between launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracing_amd import capi, scenegen
w = 3840
_, _, objs, lights = scenegen.config("C3")
st = torch.cuda.Stream()
with capi.Context(0) as ctx:
    ctx.scene_upload(objs, lights)
    m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(2160))
    for h in [int(x) for x in os.environ.get('HS', '135,270,540,1080,2160,4320,8640').split(',')]:
        out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        n = max(20, min(400, int(100 * 2160 / h)))
        torch.cuda.synchronize()
        for rep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(n):
                # rows of a taller virtual frame keep the same per-row content statistics irrelevant here: cost per (pixel, sphere) is uniform
                ctx.render_rows_device(m, w, max(h, 2160), 0, h, 1, 0, out.data_ptr(), 0, st.cuda_stream)
            e1.record(st)
            e1.synchronize()
        ctx.render_rows_device(m, w, max(h, 2160), 0, h, 1, capi.RT_TIMED, out.data_ptr(), 0, st.cuda_stream)
        t = e0.elapsed_time(e1) / n
        print("h=%5d  n=%3d  ms/frame=%.4f  us/row=%.4f  TF=%.1f  last launch alone by its own events: %.4f ms" % (h, n, t, t * 1e3 / h, w * h * 1024 * 6 / (t * 1e-3) / 1e12, ctx.last_kernel_ms()), flush=True)
