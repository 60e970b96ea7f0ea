#!/usr/bin/env python3
"""Turn the rocprofv3 rocpd databases under gpurun_out/prof into the text summaries committed under profiles/."""
import glob
import os
import sqlite3
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r03_c3"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (kernel_source_sha256: ties the summary to the kernel sources it was taken with)
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)

lines = []
db = os.path.join(src, "trace", "c3_results.db")
con = sqlite3.connect(db)
lines.append("# rocprofv3 --kernel-trace --stats (top_kernels view of the rocpd database)")
lines.append("# kernel, calls, total_us, average_us, percent")
for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    lines.append(f"\"{name[:140]}\",{calls},{total:.3f},{avg:.3f},{pct:.3f}")
# Which launches do those averages cover?  top_kernels averages EVERY launch of the run: bench.py's priming and warm-up steps, the
# timed steps and, for the general path, the launches after the timed region that read the per-kernel events.  The second table gives,
# per rtamd kernel, the same figures over the last 60 % of its launches (steady state: clocks up, past priming and warm-up).
lines.append("# per rtamd kernel: all launches (priming + warm-up + timed + post-run probes) vs the last 60 % of its launches (steady state)")
lines.append("# kernel, launches, mean_us_all, median_us_all, launches_steady, mean_us_steady, median_us_steady")
per = {}
for name, dur in con.execute("select name, duration from kernels where name like '%rtamd%' order by start"):
    per.setdefault(name, []).append(dur / 1000.0)
for name, d in per.items():
    sd = d[int(len(d) * 0.4):]
    srt, ssd = sorted(d), sorted(sd)
    lines.append(f"\"{name[:140]}\",{len(d)},{sum(d) / len(d):.3f},{srt[len(srt) // 2]:.3f},{len(sd)},{sum(sd) / len(sd):.3f},{ssd[len(ssd) // 2]:.3f}")
cmd = ""
try:
    cmd = open(os.path.join(src, "command.txt")).read().strip()
except OSError:
    pass
lines.insert(0, "# command: " + cmd)
lines.insert(1, "kernel source sha256 = " + bench.kernel_source_sha256())
open(dst + "_kernel_stats.csv", "w").write("\n".join(lines) + "\n")
out = ["# rocprofv3 --pmc passes, averages per dispatch, one block per rtamd kernel", "# command: " + cmd,
       "kernel source sha256 = " + bench.kernel_source_sha256(),
       "# (VGPRs below is rocprofv3's vgpr_count column; the code object's allocation is in profiles/r03_kernel_resource_usage.txt)", ""]
per_kernel = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "c3_results.db"))):
    c = sqlite3.connect(f)
    q = ("select kernel_name, counter_name, avg(value), count(*), max(grid_size_x), max(grid_size_y), max(workgroup_size_x), max(lds_block_size), max(vgpr_count), "
         "max(sgpr_count), max(scratch_size) from counters_collection where kernel_name like '%rtamd%' group by kernel_name, counter_name")
    for kname, name, v, n, gx, gy, wx, lds, vg, sg, scr in c.execute(q):
        d = per_kernel.setdefault(kname, {"vals": {}, "lines": [], "geom": None})
        d["vals"][name] = v
        d["geom"] = (gx, gy, wx, lds, vg, sg, scr)
        d["lines"].append(f"{os.path.basename(os.path.dirname(f)):10s} {name:28s} {v:18.1f}   (n={n})")
for kname, d in per_kernel.items():
    geom, vals = d["geom"], d["vals"]
    out.append("## " + kname)
    out.append(f"grid={geom[0]}x{geom[1]} threads, workgroup={geom[2]}, LDS/block={geom[3]} B, VGPRs={geom[4]}, SGPRs={geom[5]}, scratch={geom[6]} B")
    out += d["lines"]
    g = vals.get
    if g("SQ_INSTS_VALU"):
        out.append("# derived")
        fl = 64 * (2 * g("SQ_INSTS_VALU_FMA_F32", 0) + g("SQ_INSTS_VALU_ADD_F32", 0) + g("SQ_INSTS_VALU_MUL_F32", 0) + g("SQ_INSTS_VALU_TRANS_F32", 0))
        out.append(f"counter-derived f32 flops per dispatch (64 x (2 FMA + ADD + MUL + TRANS)) = {fl:.4e}")
        out.append(f"VALU wave-instructions per dispatch = {g('SQ_INSTS_VALU'):.4e}; FMA share = {g('SQ_INSTS_VALU_FMA_F32', 0) / g('SQ_INSTS_VALU'):.3f}")
        if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            out.append(f"VALU lane utilisation (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)) = {g('SQ_THREAD_CYCLES_VALU') / (64 * g('SQ_ACTIVE_INST_VALU')):.3f}")
        if g("GRBM_GUI_ACTIVE"):
            out.append(f"GRBM_GUI_ACTIVE / 8 XCDs = {g('GRBM_GUI_ACTIVE') / 8:.4e} cycles per dispatch")
        if g("GRBM_GUI_ACTIVE") and g("SQ_WAVE_CYCLES"):
            # SQ_WAVE_CYCLES counts quad-cycles summed over waves (MI355X_MICROARCH.md); 1024 SIMDs on the chip, 8 wave slots each
            occ = g("SQ_WAVE_CYCLES") * 4 / (1024 * g("GRBM_GUI_ACTIVE") / 8)
            out.append(f"wave occupancy (4 x SQ_WAVE_CYCLES / (1024 SIMDs x cycles)) = {occ:.2f} resident waves per SIMD on average ({occ / 8 * 100:.0f} % of the 8 slots)")
        if g("SQ_LDS_IDX_ACTIVE") is not None:
            out.append(f"LDS bank-conflict cycles / LDS active cycles = {g('SQ_LDS_BANK_CONFLICT', 0):.0f} / {g('SQ_LDS_IDX_ACTIVE'):.4e}")
        if g("TCC_EA0_WRREQ_sum") is not None:
            out.append(f"HBM write bytes = TCC_EA0_WRREQ_sum x 64 B = {g('TCC_EA0_WRREQ_sum') * 64:.4e}")
        if g("TCC_EA0_RDREQ_sum") is not None:
            out.append(f"HBM read bytes  = TCC_EA0_RDREQ_sum x 64 B x 2 (gfx950 correction, MI355X_MICROARCH.md) = {g('TCC_EA0_RDREQ_sum') * 128:.4e}")
    out.append("")
open(dst + "_pmc.txt", "w").write("\n".join(out) + "\n")
print("\n".join(lines[:6]))
print("\n".join(out))
