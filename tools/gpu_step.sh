#!/bin/bash
# one GPU iteration: selected gpu tests, bench (BAT, graph), steady-state kernel trace.  usage: gpu_step.sh <tag> [pytest -k expr] [model]
TAG=${1:-step}; KEXPR=${2:-}; MODEL=${3:-BAT}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -n "$KEXPR" ]; then
  timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider -k "$KEXPR" > $OUT/pytest.log 2>&1
else
  timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest.log 2>&1
fi
echo "pytest exit $?" >> $OUT/pytest.log
grep -E "passed|failed|^E  |exit|^FAILED|Error" $OUT/pytest.log | cut -c1-400 | head -40
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --model $MODEL > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
tail -3 $OUT/bench.err | cut -c1-300
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_graph -o bench -- python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline --model $MODEL > $OUT/rocprof_graph.log 2>&1; echo "trace graph exit $?"
cd $REPO
G=$(find $OUT/trace_graph -name "*kernel_trace.csv" | head -1); [ -n "$G" ] && python tools/trace_steps.py "$G" 20 90 > $OUT/steady_state_per_step.txt
rm -rf $OUT/trace_graph
head -45 $OUT/steady_state_per_step.txt | cut -c1-150
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json")); r=d.get("roofline",{})
    print("BENCH", d["value"], d["ms_per_step"], "frac", r.get("frac"), "gemm ms", r.get("gemm_ms_per_step"), "fused ms", r.get("fused_kernels_ms_per_step"))
except Exception as e:
    print("ERR", e)
PY
