"""Per-kernel SQ counter table from a rocprofv3 --pmc counter_collection.csv (own GEMM-family kernels).
usage: sq_summary.py <counter_collection.csv> [out.csv]
wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of resident wave time spent in s_waitcnt / barriers);
issue = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; mfma = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES: a ratio of two raw
counters with different aggregation (matrix-pipe busy cycles are summed over every SIMD, SQ busy cycles are counted once
per shader engine), so its unit is "busy SIMD-cycles per busy SE-cycle" -- dimensionless, NOT a fraction of peak (values
of 4-19 are normal); it ranks kernels against each other, the roofline fraction is the FLOP/s figure of bench.py.
Several input CSVs (one per --pmc pass) may be given: counters are merged by kernel name."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3dsot_amd.fused import GEMM_KERNEL_SYMBOLS, kernel_symbol  # noqa: E402  (the one definition of the GEMM family)

STREAMING = ("reduce_gather_kernel", "pool_t_kernel", "pool_c_kernel", "expand_c_kernel", "sa_eval_kernel", "pool_bwd_c_kernel",
             "dw0_xyz_kernel", "row_mlp_fwd_kernel", "row_mlp_bwd_kernel")

srcs = [a for a in sys.argv[1:] if a.endswith("counter_collection.csv") or a.endswith("_in.csv")] or [sys.argv[1]]
outs = [a for a in sys.argv[1:] if a not in srcs]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
rows = []
for src in srcs:
    rows += [dict(r, _src=src) for r in csv.DictReader(open(src))]
for r in rows:
    k = r["Kernel_Name"]
    sym = kernel_symbol(k)
    if sym not in GEMM_KERNEL_SYMBOLS and sym not in STREAMING and not (sym or "").startswith("xcorr_"):
        continue
    key = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    n[key].add((r["_src"], r["Dispatch_Id"]))
npass = max(len(srcs), 1)
out = csv.writer(open(outs[0], "w") if outs else sys.stdout)
out.writerow(["kernel", "dispatches", "wave_cycles_per_dispatch", "wait_any/wave_cycles", "wait_inst_any/wave_cycles",
              "active_inst_any/wave_cycles", "mfma_busy_simd_cycles_per_busy_se_cycle", "mfma_mops_f32_per_dispatch"])
for key, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    d = max(len(n[key]) // npass, 1)           # every pass re-runs the same dispatches
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    out.writerow([key, d, int(wc / d), "%.3f" % (c.get("SQ_WAIT_ANY", 0) / wc), "%.3f" % (c.get("SQ_WAIT_INST_ANY", 0) / wc),
                  "%.3f" % (c.get("SQ_ACTIVE_INST_ANY", 0) / wc),
                  "%.3f" % (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c.get("SQ_BUSY_CYCLES", 0) or 1)),
                  int(c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) / d)])
