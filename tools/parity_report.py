#!/usr/bin/env python3
"""Parity report GPU vs CPU oracle (north star: +-1 LSB per channel on >= 99.9 % of pixels + list of grazing-edge hit/miss differences).
Writes gpurun_out/parity_report.md.  Run on the GPU box: python tools/parity_report.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from raytracing_amd import capi, scene, scenegen
from tests import oracle_py

orc = oracle_py.load()
lines = ["# Parity report: HIP path (C ABI) vs CPU oracle", "",
         "| case | pixels compared | exact | within +-1 LSB | max abs diff | hit/miss or winner differences |", "|---|---|---|---|---|---|"]
edge_lists = []


def compare(name, gpu, ref, ghit, rhit):
    d = np.abs(gpu.astype(np.int32) - ref.astype(np.int32)).max(axis=-1)
    n = d.size
    diff_hit = np.argwhere(ghit != rhit)
    lines.append(f"| {name} | {n} | {100.0 * (d == 0).mean():.4f} % | {100.0 * (d <= 1).mean():.4f} % | {int(d.max())} | {len(diff_hit)} |")
    for y, x in diff_hit[:50]:
        edge_lists.append(f"{name}: pixel (row {y}, col {x}): gpu element {int(ghit[y, x])}, oracle element {int(rhit[y, x])}")


with capi.Context(0) as ctx:
    el, li = scene.default_scene()
    o, l = scene.flatten_elements(el), scene.flatten_lights(li)
    for name, w, h in (("C1 default scene 100x100", 100, 100), ("C2 default scene 1920x1080", 1920, 1080)):
        m = orc.default_camera(w, h)
        ctx.scene_upload(o, l)
        g, gh = ctx.render_rows(m, w, h, 0, h, 1, 0, want_hit_idx=True)
        r, rh = orc.render(o, l, m, w, h, want_hit_idx=True)
        compare(name, g, r, gh, rh)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "result_100x100.npy"))
    m = orc.default_camera(100, 100)
    g, gh = ctx.render_rows(m, 100, 100, 0, 100, 1, 0, want_hit_idx=True)
    compare("C1 vs the reference's result.png", g[..., :3], gold, gh, gh)
    g, gh2 = ctx.render_rows(m, 100, 100, 0, 100, 1, capi.RT_CULL, want_hit_idx=True)
    compare("C1 with RT_CULL vs the reference's result.png", g[..., :3], gold, gh2, gh)
    for cfg, step, flags, depth in (("C3", 8, 0, 1), ("C4", 48, 0, 1), ("C5", 24, 0, 1), ("C5", 60, 1, 1), ("C3", 24, 0, 2)):
        w, h, o, l = scenegen.config(cfg)
        m = orc.default_camera(w, h)
        ctx.scene_upload(o, l)
        t0 = time.time()
        g, gh = ctx.render_rows(m, w, h, 0, h, depth, flags, want_hit_idx=True)
        r, rh = orc.render(o, l, m, w, h, row_step=step, depth=depth, flags=flags, want_hit_idx=True)
        compare(f"{cfg} {w}x{h} N={o.n} L={l.n} depth={depth} shadows={flags} (every {step}th row)", g[::step], r[::step], gh[::step], rh[::step])
        gc, ghc = ctx.render_rows(m, w, h, 0, h, depth, flags | capi.RT_CULL, want_hit_idx=True)  # opt-in: primary rays through the cone hierarchy
        compare(f"{cfg} depth={depth} shadows={flags} with RT_CULL vs the oracle (every {step}th row)", gc[::step], r[::step], ghc[::step], rh[::step])
        compare(f"{cfg} depth={depth} shadows={flags} with RT_CULL vs the default launch (WHOLE frame)", gc, g, ghc, gh)
        print(cfg, flags, depth, "%.1f s" % (time.time() - t0), flush=True)

lines += ["", "## Grazing-edge hit/miss differences", ""]
lines += edge_lists if edge_lists else ["None: every compared pixel has the same winning element and the same bytes. The kernel's FMA filter only selects candidates;",
                                        "every accepted hit is computed with the reference's unfused f32 arithmetic (DESIGN.md §3.1), so there is no grazing-edge class."]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "parity_report.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
