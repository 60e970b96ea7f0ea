#!/usr/bin/env python3
"""DEV: host cost of the pieces of one multi-rank bench step (one rank, nccl process group of size 1, 270-row band)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")
import numpy as np, torch, torch.distributed as dist
from raytracing_amd import capi, scenegen
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
w, h, objs, lights = scenegen.config("C3")
m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
ctx = capi.Context(0); ctx.scene_upload(objs, lights)
rows = 270
st = torch.cuda.Stream(dev)
band = torch.zeros((rows, w, 4), dtype=torch.uint8, device=dev)
outl = [torch.empty_like(band)]
N = 300
def timeit(name, fn):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N): fn()
    t = (time.perf_counter() - t0) / N * 1e6
    torch.cuda.synchronize()
    print("%-60s %6.1f us" % (name, t))
def f_ctx():
    with torch.cuda.stream(st): pass
timeit("with torch.cuda.stream(st): pass", f_ctx)
timeit("ctx.render_rows_device", lambda: ctx.render_rows_device(m, w, h, 945, 945 + rows, 1, 0, band.data_ptr(), 0, st.cuda_stream))
evs = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
timeit("event.record(st)", lambda: evs[0].record(st))
works = []
def f_g():
    wk = dist.gather(band, gather_list=outl, dst=0, async_op=True)
    works.append(wk)
timeit("dist.gather(async_op=True), 1 output", f_g)
def f_w():
    if works: works.pop().wait()
timeit("Work.wait()", f_w)
# direct RCCL (raytracing_amd.rccl_gather): host cost of one grouped send + recv of a 270-row band with itself
from raytracing_amd import rccl_gather
g = rccl_gather.RcclBandGather(1, 0, 0)
frame = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
timeit("rccl loopback (ncclGroupStart, Recv, Send, GroupEnd) 4.1 MB", lambda: g.loopback(band.data_ptr(), frame.data_ptr(), band.numel(), st.cuda_stream))
small = 64 * 1024
timeit("rccl loopback 64 KB", lambda: g.loopback(band.data_ptr(), frame.data_ptr(), small, st.cuda_stream))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record(st)
for _ in range(100): g.loopback(band.data_ptr(), frame.data_ptr(), band.numel(), st.cuda_stream)
e1.record(st); e1.synchronize()
print("GPU time of one 4.1 MB loopback: %.1f us" % (e0.elapsed_time(e1) * 10))
g.close()
dist.destroy_process_group()
