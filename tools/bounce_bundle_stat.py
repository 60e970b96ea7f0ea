"""CPU study (round 3): how narrow are the bundles of bounce rays a wave holds?  Hit plane from the oracle, records sorted by the spatial
rank of their sphere, groups of G; per group the origin ball, the direction cone and the share of the spheres any ray of the group can
reach with the margins of the bounce filter (what trace_bundle lists and walks).  float64 statistics only.  Usage: <config> <group size>"""
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from raytracing_amd import scenegen
from tests import oracle_py
cfg = sys.argv[1]; G = int(sys.argv[2])
w, h, objs, lights = scenegen.config(cfg)
orc = oracle_py.load(); m32 = orc.default_camera(w, h); m = m32.astype(np.float64)
C = np.stack([objs.cx, objs.cy, objs.cz], 1).astype(np.float64); R = objs.radius.astype(np.float64)
cache = "/tmp/hit_%s.npy" % cfg
if os.path.exists(cache): hit = np.load(cache)
else:
    _, hit = orc.render(objs, lights, m32, w, h, want_hit_idx=True); np.save(cache, hit)
ys, xs = np.nonzero(hit >= 0); k = hit[ys, xs]
vx = 2 * (xs / w) - 1; vy = 2 * (ys / h) - 1
pos = np.stack([m[0] * vx, m[5] * vy, np.full(vx.shape, m[10] + m[11])], -1)
d = pos / np.linalg.norm(pos, axis=-1, keepdims=True)
ck = C[k]; rk = R[k]
a = (d * ck).sum(-1); disc = a * a - ((ck * ck).sum(-1) - rk ** 2)
t = a - np.sqrt(np.maximum(disc, 0)); hp = d * t[:, None]
nrm = -ck / np.linalg.norm(ck, axis=1, keepdims=True)      # reference quirk: normal at the OLD origin (0) = unit(-c)
dn = (d * nrm).sum(-1); refl = d - 2 * dn[:, None] * nrm; refl /= np.linalg.norm(refl, axis=1, keepdims=True)
# spatial rank = Morton order of centres
lo, hi = C.min(0), C.max(0); q = ((C - lo) / (hi - lo) * 1023).astype(np.uint64)
def spread(v):
    r = np.zeros_like(v)
    for b in range(10): r |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
    return r
code = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
rank = np.empty(len(C), np.int64); rank[np.argsort(code, kind="stable")] = np.arange(len(C))
# emission order within a sphere ~ screen tile order
tile = (ys // 16) * ((w + 15) // 16) + xs // 16
o = np.lexsort((xs, ys, tile, rank[k]))
O = hp[o]; D = refl[o]
ng = len(O) // G
rng = np.random.default_rng(0); sel = rng.choice(ng, 400, replace=False)
fr = []; wide = 0; thetas = []
for g in sel:
    og = O[g * G:(g + 1) * G]; dg = D[g * G:(g + 1) * G]
    oc = og.mean(0); ro = np.linalg.norm(og - oc, axis=1).max()
    ax = dg.sum(0); ax /= np.linalg.norm(ax); cmin = (dg @ ax).min()
    th = np.arccos(np.clip(cmin, -1, 1)); thetas.append(th)
    if cmin <= 0.17: wide += 1; fr.append(1.0); continue
    v = C - oc; dist = np.linalg.norm(v, axis=1); Rr = R * 1.05 + 6e-3 * dist + ro
    ang = np.arccos(np.clip((v @ ax) / dist, -1, 1))
    phi = np.arcsin(np.minimum(1, Rr / dist))
    fr.append(((dist <= Rr) | (ang <= th + phi)).mean())
print(cfg, "group", G, "mean theta deg %.1f" % np.degrees(np.mean(thetas)), "wide frac %.3f" % (wide / len(sel)), "mean shortlist fraction %.3f" % np.mean(fr), "median %.3f" % np.median(fr))
