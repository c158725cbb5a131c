#!/bin/bash
# round-2 first GPU call: full gpu test suite (new full-size tests), bench A/B of the layer-0 reduce variants
TAG=${1:-r2a}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{ nproc; lscpu | grep -E "Model name|^CPU\(s\)|Socket|Thread"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit|Max Clock" | head -8; } > $OUT/env.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider --durations=15 > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
grep -E "passed|failed|^E  |exit|^FAILED" $OUT/pytest.log | cut -c1-300 | head -40
timeout 600 python bench.py --steps 100 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
timeout 300 env O3D_REDUCE_GATHER=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_rg.json 2> $OUT/bench_rg.err; echo "bench rg exit $?"
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --dense > $OUT/bench_dense.json 2> $OUT/bench_dense.err; echo "bench dense exit $?"
timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --model P2B > $OUT/bench_p2b.json 2> $OUT/bench_p2b.err; echo "bench p2b exit $?"
for f in bench bench_rg bench_dense bench_p2b; do python - <<PY
import json
try:
    d=json.load(open("$OUT/$f.json")); r=d.get("roofline",{})
    print("$f", d["value"], d["ms_per_step"], "frac", r.get("frac"), "live", r.get("live_fraction"), "cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("cores"), d.get("cpu_baseline",{}).get("thread_sweep_pairs_per_s"))
except Exception as e:
    print("$f", "ERR", e); print(open("$OUT/$f.err").read()[-1500:])
PY
done
