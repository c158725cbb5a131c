"""Per-kernel SQ counter table from a rocprofv3 --pmc counter_collection.csv (own GEMM-family kernels).
usage: sq_summary.py <counter_collection.csv> [out.csv]
wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of resident wave time spent in s_waitcnt / barriers);
issue = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; mfma = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES (matrix-pipe busy
cycles summed over the SIMDs per busy SQ cycle, as the counters report them -- compare kernels, not absolutes)."""
import collections
import csv
import re
import sys

src = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(src)):
    k = r["Kernel_Name"]
    if not any(t in k for t in ("direct_gemm_kernel", "wgrad2_kernel", "conv_fwd_kernel", "conv_dgrad_kernel",
                                "conv_wgrad_kernel", "reduce_c_kernel", "pool_c_kernel", "expand_c_kernel")):
        continue
    m = re.search(r"(\w*kernel\w*(<[^>]*>)?)", k)
    key = m.group(1) if m else k[:60]
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    n[key].add(r["Dispatch_Id"])
out = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout)
out.writerow(["kernel", "dispatches", "wave_cycles_per_dispatch", "wait_any/wave_cycles", "wait_inst_any/wave_cycles",
              "active_inst_any/wave_cycles", "mfma_busy/busy_cycles", "mfma_mops_f32_per_dispatch"])
for key, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    d = max(len(n[key]), 1)
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    out.writerow([key, d, int(wc / d), "%.3f" % (c.get("SQ_WAIT_ANY", 0) / wc), "%.3f" % (c.get("SQ_WAIT_INST_ANY", 0) / wc),
                  "%.3f" % (c.get("SQ_ACTIVE_INST_ANY", 0) / wc),
                  "%.3f" % (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (c.get("SQ_BUSY_CYCLES", 0) or 1)),
                  int(c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0) / d)])
