#!/bin/bash
# GPU pass: all gpu tests, smoke, graph + eager bench, rocprofv3 kernel stats (csv summary)
TAG=${1:-run}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{ nproc; lscpu | grep -E "Model name|^CPU\(s\)|Socket|Thread"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit|Max Clock" | head -8; } > $OUT/env.txt 2>&1
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log
tail -3 $OUT/smoke.log
timeout 1200 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
grep -E "passed|failed|^E  |exit" $OUT/pytest.log | cut -c1-250 | head -30
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench graph exit $?"
grep -E "bench\]|Error" $OUT/bench.err | tail -4
timeout 600 python bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline > $OUT/bench_eager.json 2> $OUT/bench_eager.err; echo "bench eager exit $?"
grep -E "bench\]|Error" $OUT/bench_eager.err | tail -4
python - <<PY
import json
for f in ("bench_eager.json","bench.json"):
    try:
        d=json.loads(open("$OUT/"+f).read())
        print(f, d["value"], d["ms_per_step"], d["config"].get("hip_graph"), d["roofline"]["achieved"], d["roofline"].get("gemm_ms_per_step"))
        for k,v in d["roofline"].get("per_kernel",{}).items(): print("   ",k,v)
        print(d.get("cpu_baseline"))
    except Exception as e: print(f, "ERR", e)
PY
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline > $OUT/rocprof.log 2>&1; echo "rocprof exit $?"
cd $REPO
F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/kernel_stats.csv && head -45 "$F" | cut -c1-200
# keep only the summaries (the raw trace is large)
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
