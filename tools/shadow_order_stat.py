"""CPU study for the shadow any-hit ordering (round 3): for sampled (hit point, light) pairs of C5, where in a per-light
ordered sphere list does the FIRST blocker sit?  float64 geometry (statistics only, not parity)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import capi, scenegen
from tests import oracle_py

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
w, h, objs, lights = scenegen.config(cfg)
orc = oracle_py.load()
m = orc.default_camera(w, h)
# sample hit points: every 24th row, every 16th column, via the oracle hit index + a float64 recomputation of the hit point
step_r, step_c = 24, 16
C = np.stack([objs.cx, objs.cy, objs.cz], 1).astype(np.float64); R = objs.radius.astype(np.float64)
pts = []
for y in range(0, h, step_r):
    img, hit = orc.render(objs, lights, m, w, h, y0=y, y1=y + 1, want_hit_idx=True)
    for x in range(0, w, step_c):
        k = hit[0, x]
        if k < 0: continue
        pos = orc.pixel_position(m, x, y, w, h).astype(np.float64)
        d = pos / np.linalg.norm(pos)
        a = d @ C[k]; disc = a * a - (C[k] @ C[k] - R[k] ** 2)
        if disc < 0: continue
        t = a - np.sqrt(disc)
        pts.append(d * t)
P = np.array(pts)
print("hit points sampled:", len(P))
L = np.stack([lights.px, lights.py, lights.pz], 1).astype(np.float64)
N = len(R)
res = {}
for name in ("scene", "solid", "dist", "solid_w"):
    res[name] = []
first_rank = {k: [] for k in res}
n_pairs = 0; n_occ = 0
for l in range(len(L)):
    lp = L[l]
    v = lp - P; sd = np.linalg.norm(v, axis=1); u = v / sd[:, None]
    # blockers: t in (1e-3, sd)
    q = P[:, None, :] - C[None, :, :]                       # [p, s, 3]
    al = -(q * u[:, None, :]).sum(-1)
    disc = al * al - ((q * q).sum(-1) - R[None, :] ** 2)
    ok = disc >= 0
    sq = np.sqrt(np.where(ok, disc, 0))
    t = np.where(al > sq, al - sq, al + sq)
    blk = ok & (t > 1e-3) & (t < sd[:, None])               # [p, s]
    occ = blk.any(1)
    n_pairs += len(P); n_occ += occ.sum()
    dl = np.linalg.norm(C - lp, axis=1)
    orders = {
        "scene": np.arange(N),
        "solid": np.argsort(-(R / dl) ** 2, kind="stable"),
        "dist": np.argsort(dl - R, kind="stable"),
    }
    # data-weighted: solid angle x fraction of hit points farther from the light than the sphere
    far = (sd[:, None] > dl[None, :]).mean(0)
    orders["solid_w"] = np.argsort(-((R / dl) ** 2) * far, kind="stable")
    for name, o in orders.items():
        b = blk[:, o][occ]
        first_rank[name].append(b.argmax(1))
print("pairs %d, occluded %.3f" % (n_pairs, n_occ / n_pairs))
for name in first_rank:
    fr = np.concatenate(first_rank[name])
    qs = [np.mean(fr < k) for k in (64, 128, 256, 512, 1024, 2048)]
    print("%-8s frac of occluded pairs with a blocker in first K=64/128/256/512/1024/2048: %s  mean rank %.0f" % (name, " ".join("%.3f" % x for x in qs), fr.mean()))
    focc = n_occ / n_pairs
    for K in (128, 256, 512, 1024):
        found = np.mean(fr < K) * focc
        cost = K / N + (1 - found) * (1 - K / N)
        print("    two-phase K=%d: relative work %.3f" % (K, cost))
    # three-phase
    for K1, K2 in ((128, 512), (128, 1024), (256, 1024), (64, 512)):
        f1 = np.mean(fr < K1) * focc; f2 = np.mean(fr < K2) * focc
        cost = K1 / N + (1 - f1) * (K2 - K1) / N + (1 - f2) * (1 - K2 / N)
        print("    three-phase K=%d,%d: relative work %.3f" % (K1, K2, cost))
