"""CPU study (round 3): per-tile box vote for shadow rays.  Emulates the record order of render_kernel<EMIT> on a few 16-row bands of
C5, groups records in 64s, bounds the reversed ray directions w of each group by a box, and counts the spheres whose per-light filter
f = w.c'' - 1 can be >= 0 for some w in the box (the shortlist a tile would walk).  float64 statistics only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import capi, scenegen
from tests import oracle_py

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
w, h, objs, lights = scenegen.config(cfg)
orc = oracle_py.load()
m = orc.default_camera(w, h).astype(np.float64)
C = np.stack([objs.cx, objs.cy, objs.cz], 1).astype(np.float64); R = objs.radius.astype(np.float64)
L = np.stack([lights.px, lights.py, lights.pz], 1).astype(np.float64)
rng = np.random.default_rng(1)
blocks = []   # per primary wave: array of hit points in emission order
for y0 in (320, 800, 1072, 1360, 1840):
    img, hit = orc.render(objs, lights, orc.default_camera(w, h), w, h, y0=y0, y1=y0 + 16, want_hit_idx=True)
    ys, xs = np.mgrid[y0:y0 + 16, 0:w]
    vx = 2 * (xs / w) - 1; vy = 2 * (ys / h) - 1
    pos = np.stack([m[0] * vx, m[5] * vy, np.full(vx.shape, m[10] + m[11])], -1)
    d = pos / np.linalg.norm(pos, axis=-1, keepdims=True)
    k = hit
    ck = C[np.maximum(k, 0)]; rk = R[np.maximum(k, 0)]
    a = (d * ck).sum(-1); disc = a * a - ((ck * ck).sum(-1) - rk ** 2)
    t = a - np.sqrt(np.maximum(disc, 0))
    hp = d * t[..., None]
    for x0 in range(0, w, 16):           # one wave = 16 columns x 16 rows; order p (rows ly + 4p), lane = ly*16 + lx
        pts = []
        for p in range(4):
            for ly in range(4):
                row = ly + 4 * p
                sel = k[row, x0:x0 + 16] >= 0
                pts.append(hp[row, x0:x0 + 16][sel])
        pts = np.concatenate(pts)
        if len(pts): blocks.append(pts)
order = rng.permutation(len(blocks)) if os.environ.get("SHUFFLE") else np.arange(len(blocks))
P = np.concatenate([blocks[i] for i in order])
print("records:", len(P), "wave blocks:", len(blocks))
ngroups = len(P) // 64
P = P[:ngroups * 64]
tot_short = []; tot_group = []
for l in range(0, len(L), 4):
    lp = L[l]
    q = C - lp; kk = (q * q).sum(1) - R ** 2
    okf = kk > 0
    cf = np.where(okf[:, None], q / np.sqrt(np.where(okf, kk, 1))[:, None], 0)     # c''
    wv = (P - lp); wv /= np.linalg.norm(wv, axis=1, keepdims=True)                  # reversed ray direction: from the light to the hit point
    wg = wv.reshape(ngroups, 64, 3)
    lo = wg.min(1); hi = wg.max(1)                                                  # [g, 3]
    # per group: f_box = sum_i max(lo_i c_i, hi_i c_i) - 1
    fb = np.maximum(lo[:, None, :] * cf[None], hi[:, None, :] * cf[None]).sum(-1) - 1  # [g, s]
    passes = fb >= 0
    tot_group.append(passes.sum(1).mean())
    nt = ngroups // 16
    tile = passes[:nt * 16].reshape(nt, 16, -1).any(1)
    tot_short.append(tile.sum(1).mean())
    # per-wave (4 groups) union
    wave = passes[:nt * 16].reshape(nt * 4, 4, -1).any(1)
    # exact per-ray candidates for comparison
    fr = (wg[:, :, None, :] * cf[None, None]).sum(-1) - 1 if ngroups * 64 * len(R) < 4e8 else None
    print("light %2d: spheres passing per 64-ray group %.1f, per 256-ray wave %.1f, per 1024-ray tile (union) %.1f (+%d always)" % (l, passes.sum(1).mean(), wave.sum(1).mean(), tile.sum(1).mean(), (~okf).sum()), flush=True)
print("mean shortlist per tile: %.1f of %d spheres (%.3f); per group %.1f" % (np.mean(tot_short), len(R), np.mean(tot_short) / len(R), np.mean(tot_group)))
