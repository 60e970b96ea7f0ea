import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import capi, scenegen
w, h, objs, lights = scenegen.config("C5")
m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
f = np.float32
with capi.Context(0) as ctx:
    # per light: brightness with and without shadows -> occluded fraction of (hit, light) pairs, weighted by the light's contribution
    tot_l = tot_s = 0.0
    for l in range(0, lights.n, 4):
        one = capi.FlatLights(lights.px[l:l+1], lights.py[l:l+1], lights.pz[l:l+1], lights.cr[l:l+1], lights.cg[l:l+1], lights.cb[l:l+1], lights.intensity[l:l+1] * f(8))
        ctx.scene_upload(objs, one)
        lit = ctx.render_rows(m, w, h, 0, h, 1, 0)[0][..., :3].astype(np.float64)
        sh = ctx.render_rows(m, w, h, 0, h, 1, capi.RT_SHADOWS)[0][..., :3].astype(np.float64)
        n_lit = (lit.sum(-1) > 0).sum(); n_sh = (sh.sum(-1) > 0).sum()
        print("light %2d: pixels lit without shadows %d, with shadows %d -> occluded fraction %.3f" % (l, n_lit, n_sh, 1 - n_sh / max(1, n_lit)), flush=True)
        tot_l += n_lit; tot_s += n_sh
    print("overall occluded fraction of (hit pixel, light) pairs ~ %.3f" % (1 - tot_s / tot_l))
