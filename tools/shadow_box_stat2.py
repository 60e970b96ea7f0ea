"""CPU study (round 3): box vote with records SORTED BY HIT SPHERE.  Full C5 frame hit plane from the oracle; records grouped in 64s in
(sphere, pixel) order; per group a box on the reversed shadow-ray directions and the largest ray length; shortlist = spheres whose
per-light filter can pass for some direction of the box and that are not entirely farther from the light than every hit point."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracing_amd import capi, scenegen
from tests import oracle_py

cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
w, h, objs, lights = scenegen.config(cfg)
orc = oracle_py.load()
m32 = orc.default_camera(w, h)
m = m32.astype(np.float64)
C = np.stack([objs.cx, objs.cy, objs.cz], 1).astype(np.float64); R = objs.radius.astype(np.float64)
L = np.stack([lights.px, lights.py, lights.pz], 1).astype(np.float64)
cache = "/tmp/hit_%s.npy" % cfg
if os.path.exists(cache):
    hit = np.load(cache)
else:
    img, hit = orc.render(objs, lights, m32, w, h, want_hit_idx=True)
    np.save(cache, hit)
ys, xs = np.nonzero(hit >= 0)
k = hit[ys, xs]
print("records", len(k), "coverage %.3f" % (len(k) / (w * h)))
vx = 2 * (xs / w) - 1; vy = 2 * (ys / h) - 1
pos = np.stack([m[0] * vx, m[5] * vy, np.full(vx.shape, m[10] + m[11])], -1)
d = pos / np.linalg.norm(pos, axis=-1, keepdims=True)
ck = C[k]; rk = R[k]
a = (d * ck).sum(-1); disc = a * a - ((ck * ck).sum(-1) - rk ** 2)
t = a - np.sqrt(np.maximum(disc, 0))
hp = d * t[:, None]
mode = os.environ.get("ORDER", "sphere")
if mode == "sphere":
    o = np.argsort(k, kind="stable")
elif mode == "tile":   # 64 x 16 tiles, row-major over tiles
    o = np.lexsort((xs, ys, (ys // 16) * ((w + 63) // 64) + xs // 64))
P = hp[o]
G = int(os.environ.get("GROUP", "64"))
ngroups = len(P) // G
# sample groups to keep it quick
rng = np.random.default_rng(0)
tiles = rng.choice(ngroups // (1024 // G) - 1, 300, replace=False)
gsel = (tiles[:, None] * (1024 // G) + np.arange(1024 // G)[None]).ravel()
Pg = P[: ngroups * G].reshape(ngroups, G, 3)[gsel]          # [g, G, 3]
res_g = []; res_t = []; res_g_nolen = []
for l in range(0, len(L), 4):
    lp = L[l]
    q = C - lp; dl = np.linalg.norm(q, axis=1); kk = dl ** 2 - R ** 2
    okf = kk > 0
    cf = np.where(okf[:, None], q / np.sqrt(np.where(okf, kk, 1))[:, None], 0)
    v = Pg - lp; sd = np.linalg.norm(v, axis=-1); wv = v / sd[..., None]
    lo = wv.min(1); hi = wv.max(1); sdmax = sd.max(1)
    fb = np.maximum(lo[:, None, :] * cf[None], hi[:, None, :] * cf[None]).sum(-1) - 1
    p_dir = fb >= 0
    p_len = (dl - R)[None, :] <= sdmax[:, None] + 4e-3 * 40
    passes = p_dir & p_len
    nt = len(tiles)
    tile = passes.reshape(nt, 1024 // G, -1).any(1)
    res_g.append(passes.sum(1).mean()); res_t.append(tile.sum(1).mean()); res_g_nolen.append(p_dir.sum(1).mean())
    print("light %2d: per %d-ray group %.1f (direction only %.1f), per 1024-ray tile union %.1f" % (l, G, passes.sum(1).mean(), p_dir.sum(1).mean(), tile.sum(1).mean()), flush=True)
print("order=%s group=%d: mean per group %.1f (dir only %.1f), per tile %.1f of %d" % (mode, G, np.mean(res_g), np.mean(res_g_nolen), np.mean(res_t), len(R)))
