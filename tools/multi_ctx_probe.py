#!/usr/bin/env python3
"""One-GPU rehearsal of the in-library multi-device path (rt_ctx_create_multi with virtual bands on device 0): host time per
rt_render_rows_device call and wall time per frame for 1, 2, 4, 8 bands of config C3, against the single-device context.
Every band beyond the first is staged and travels through the library's RCCL group (a send of the device to itself)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import capi, scenegen

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
w, h, objs, lights = scenegen.config(cfg)
m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()


def run(ctx, label):
    ctx.scene_upload(objs, lights)
    for _ in range(100):
        ctx.render_rows_device(m, w, h, 0, h, 1, 0, out.data_ptr(), 0, st.cuda_stream)
    st.synchronize()
    per = []
    t0 = time.perf_counter()
    for _ in range(frames):
        a = time.perf_counter()
        ctx.render_rows_device(m, w, h, 0, h, 1, 0, out.data_ptr(), 0, st.cuda_stream)
        per.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    st.synchronize()
    t2 = time.perf_counter()
    per = np.array(per) * 1e6
    # the MEDIAN is the cost of issuing a call; the mean also holds the calls that blocked because the device's queue was full (the
    # host runs ahead of the GPU in this loop, so with enough frames it is throttled to the GPU's frame time)
    print("%-28s host median %.1f us per call (p10 %.1f, p90 %.1f, mean %.1f), %.4f ms per frame" %
          (label, np.median(per), np.percentile(per, 10), np.percentile(per, 90), (t1 - t0) / frames * 1e6, (t2 - t0) / frames * 1e3), flush=True)


with capi.Context(0) as c:
    run(c, "single-device context")
for g in (1, 2, 4, 8):
    with capi.Context(device_ids=[0] * g) as c:
        run(c, "multi context, %d band(s)" % g)
