"""profiles/hbm_traffic.json from the per-kernel PMC table of tools/pmc_summary.py.
usage: hbm_traffic.py <pmc_summary.csv> <out.json> [model=BAT] [batch=48]
GEMM family = the kernels behind bench.py's roofline launches: direct_gemm_kernel, wgrad2_kernel and the
generic conv_{fwd,dgrad,wgrad}_kernel; the bytes of the wgrad slice-reduce kernels belong to their
weight-gradient launch (counted in the bytes, not in the launches)."""
import csv
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
model = sys.argv[3] if len(sys.argv) > 3 else "BAT"
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 48
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 5          # eager steps in the profiled run (warm-up + timed)
PRIMARY = ("direct_gemm_kernel", "wgrad2_kernel", "conv_fwd_kernel", "conv_dgrad_kernel", "conv_wgrad_kernel")
EXTRA = ("wgrad_reduce",)
tot_bytes, launches, per = 0.0, 0, {}
for r in csv.DictReader(open(src)):
    k = r["kernel"]
    prim = any(p in k for p in PRIMARY)
    if not prim and not any(p in k for p in EXTRA):
        continue
    n = int(r["dispatches"])
    rd = float(r["hbm_read_bytes_per_launch(2xFETCH_SIZE)"] or 0)
    wr = float(r["hbm_write_bytes_per_launch(WRITE_SIZE)"] or 0)
    tot_bytes += n * (rd + wr)
    if prim:
        launches += n
    m = re.search(r"(\w*kernel\w*(<[^>]*>)?)", k)
    short = m.group(1) if m else k[:60]
    per[short] = {"dispatches": n, "read_bytes": int(rd), "write_bytes": int(wr)}
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3dsot_amd import build as _build  # noqa: E402

json.dump({"model": model, "workload_batch": batch, "kernel_source_sha256": _build.source_hash(),
           "gemm_family_hbm_bytes_per_launch": int(tot_bytes / max(launches, 1)),
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/gpu_round2.sh): the L2's "
                     "memory-side (fabric) request bytes, Infinity-Cache hits INCLUDED (an upper bound of HBM bytes); "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts wide reads at half size); average over "
                     "the %d GEMM-family dispatches of the eager steps of that run (tools/hbm_traffic.py)" % launches,
           "gemm_family_bytes_per_step": int(tot_bytes / max(steps, 1)),
           "per_kernel": per}, open(dst, "w"), indent=1)
print("gemm family: %d launches, %.1f MB per launch" % (launches, tot_bytes / max(launches, 1) / 1e6))
