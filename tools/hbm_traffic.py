"""profiles/hbm_traffic.json from the per-kernel PMC table of tools/pmc_summary.py.
usage: hbm_traffic.py <pmc_summary.csv> <out.json> [model=BAT] [batch=48] [eager steps in the profiled run]
GEMM family = the device kernels behind bench.py's roofline launches, as listed ONCE in the product
(open3dsot_amd/fused.py::GEMM_KERNEL_SYMBOLS); the bytes of the weight-gradient slice reductions (GEMM_REDUCE_SYMBOLS)
belong to their launch (counted in the bytes, not in the dispatches).  The json carries BYTES PER STEP; bench.py divides
by its own `roofline.launches_per_step` (C-ABI launches of the family), so that traffic x launches_per_step equals the PMC
table's family rows by construction.  A family kernel of the csv that matches no symbol is an error, not a silent
omission (round 3: fused_bwd_kernel, wgrad2_group_kernel and direct_gemm_pair_kernel fell through a substring list)."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open3dsot_amd import build as _build  # noqa: E402
from open3dsot_amd.fused import GEMM_KERNEL_SYMBOLS, GEMM_REDUCE_SYMBOLS, kernel_symbol  # noqa: E402


def summarise(rows, steps):
    tot_bytes, dispatches, per, other = 0.0, 0, {}, 0.0
    for r in rows:
        sym = kernel_symbol(r["kernel"])
        n = int(r["dispatches"])
        rd = float(r["hbm_read_bytes_per_launch(2xFETCH_SIZE)"] or 0)
        wr = float(r["hbm_write_bytes_per_launch(WRITE_SIZE)"] or 0)
        if sym in GEMM_KERNEL_SYMBOLS:
            dispatches += n
        elif sym not in GEMM_REDUCE_SYMBOLS:
            if sym and ("gemm" in sym or "wgrad" in sym or "fused_bwd" in sym or sym.startswith("conv_")):
                raise SystemExit("hbm_traffic.py: kernel %r looks like a GEMM-family kernel but is in neither symbol list of "
                                 "open3dsot_amd/fused.py" % sym)
            other += n * (rd + wr)
            continue
        tot_bytes += n * (rd + wr)
        short = r["kernel"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        per[short] = {"dispatches": n, "read_bytes": int(rd), "write_bytes": int(wr),
                      "gb_per_step": round(n * (rd + wr) / max(steps, 1) / 1e9, 4)}
    return tot_bytes, dispatches, per, other


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    model = sys.argv[3] if len(sys.argv) > 3 else "BAT"
    batch = int(sys.argv[4]) if len(sys.argv) > 4 else 48
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 5          # eager steps in the profiled run (warm-up + timed)
    tot_bytes, dispatches, per, other = summarise(list(csv.DictReader(open(src))), steps)
    json.dump({"model": model, "workload_batch": batch, "kernel_source_sha256": _build.source_hash(),
               "gemm_family_bytes_per_step": int(tot_bytes / max(steps, 1)),
               "gemm_family_device_dispatches_per_step": dispatches // max(steps, 1),
               "all_kernels_bytes_per_step": int((tot_bytes + other) / max(steps, 1)),
               "gemm_family_symbols": list(GEMM_KERNEL_SYMBOLS) + list(GEMM_REDUCE_SYMBOLS),
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/gpu_round4.sh pmc): the L2's "
                         "memory-side (fabric) request bytes, Infinity-Cache hits INCLUDED (an upper bound of HBM bytes); "
                         "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts wide reads at half size); summed over the "
                         "GEMM-family kernels of open3dsot_amd/fused.py::GEMM_KERNEL_SYMBOLS + GEMM_REDUCE_SYMBOLS in the %d eager "
                         "steps of that run, divided by the steps (tools/hbm_traffic.py)" % steps,
               "per_kernel": per}, open(dst, "w"), indent=1)
    print("gemm family: %d device dispatches per step, %.3f GB per step (all kernels %.3f GB)" %
          (dispatches // max(steps, 1), tot_bytes / max(steps, 1) / 1e9, (tot_bytes + other) / max(steps, 1) / 1e9))
