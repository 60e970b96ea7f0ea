#!/usr/bin/env python3
"""DEV: statistics of shadow_mask_kernel's any-hit hierarchy from a -DRT_MASK_STATS build (RT_MI355X_LIB=build_variants/stats_*.so)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from raytracing_amd import capi, scenegen

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
w, h, objs, lights = scenegen.config(cfg)
m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
ctx = capi.Context(0)
ctx.scene_upload(objs, lights)
out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
lib = ctx._lib
st = (C.c_uint32 * 16)()
lib.rt_debug_read_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
lib.rt_debug_read_stats(ctx._h, st)
ctx.render_rows_device(m, w, h, 0, h, 1, capi.RT_SHADOWS, out.data_ptr(), 0, 0)
torch.cuda.synchronize()
lib.rt_debug_read_stats(ctx._h, st)
v = list(st)
units = max(1, v[2])
print("units (wave, light) %d, wide %d (%.3f), clusters opened per unit %.1f, spheres listed per unit %.1f, exact calls per unit %.2f, early-outs %d (%.3f), mean sin(theta) %.4f"
      % (v[2], v[3], v[3] / units, v[4] / units, v[5] / units, v[6] / units, v[7], v[7] / units, v[8] / units / 1000.0))
