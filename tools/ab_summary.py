#!/usr/bin/env python3
"""DEV: condense the output of tools/ab_variants.sh over bench.py lines (stdin) to one line per run."""
import json
import sys

for line in sys.stdin:
    if line.startswith("=="):
        print(line.strip(), end=" ")
        continue
    try:
        d = json.loads(line)
    except ValueError:
        continue
    ks = d.get("general_path", {}).get("kernels", [])
    print("ms_per_step %.4f" % d["ms_per_step"], " ".join("%s=%.4f" % (k["kernel"].split("<")[0][:18] + ("<emit>" if "emit" in k["kernel"] else ""), k["ms"]) for k in ks), flush=True)
