import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from raytracing_amd import capi, scenegen
w, h, objs, lights = scenegen.config("C3")
m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
st = torch.cuda.Stream()
with capi.Context(0) as ctx:
    ctx.scene_upload(objs, lights)
    for _ in range(150):
        ctx.render_rows_device(m, w, h, 0, h, 1, 0, out.data_ptr(), 0, st.cuda_stream)
    st.synchronize()
    for rep in range(3):
        for mode in ("no torch events", "2 torch events per step"):
            n = 300
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            st.synchronize()
            t0 = time.perf_counter()
            for k in range(n):
                if mode[0] == "2": evs[k][0].record(st)
                ctx.render_rows_device(m, w, h, 0, h, 1, 0, out.data_ptr(), 0, st.cuda_stream)
                if mode[0] == "2": evs[k][1].record(st)
            st.synchronize()
            dt = (time.perf_counter() - t0) / n
            print("%-26s %.4f ms per step" % (mode, dt * 1e3), flush=True)
