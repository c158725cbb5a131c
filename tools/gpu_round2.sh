#!/bin/bash
# Round-2 artefacts on the GPU box: full gpu test suite, smoke, the bench lines of every BASELINE config, rocprofv3
# kernel stats of the eager step + steady-state graph trace + FETCH/WRITE PMC passes.  usage: gpu_round2.sh <tag>
TAG=${1:-round2}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{ nproc; lscpu | grep -E "Model name|^CPU\(s\)|Socket|Thread"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit|Max Clock" | head -8; } > $OUT/env.txt 2>&1
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider --durations=10 > $OUT/pytest.log 2>&1
echo "pytest exit $?" >> $OUT/pytest.log
grep -E "passed|failed|^E  |exit|^FAILED" $OUT/pytest.log | cut -c1-250 | head -20
timeout 900 python bench.py --per-launch $OUT/per_launch_roofline_bat.txt > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
timeout 300 python bench.py --no-cpu-baseline --dense --per-launch $OUT/per_launch_roofline_dense.txt > $OUT/bench_dense.json 2>/dev/null; echo "dense exit $?"
timeout 400 python bench.py --model P2B --cpu-budget 40 --per-launch $OUT/per_launch_roofline_p2b.txt > $OUT/bench_p2b.json 2>/dev/null; echo "p2b exit $?"
timeout 400 python bench.py --model P2B --batch 1 --cpu-budget 30 > $OUT/bench_p2b_b1.json 2>/dev/null; echo "p2b b1 exit $?"
timeout 300 python bench.py --model M2TRACK --per-launch $OUT/per_launch_roofline_m2track.txt > $OUT/bench_m2track.json 2>/dev/null; echo "m2track exit $?"
timeout 300 python bench.py --no-cpu-baseline --search-size 2048 > $OUT/bench_nuscenes2048.json 2>/dev/null; echo "nuscenes exit $?"
timeout 200 python bench.py --infer --steps 500 --warmup 20 > $OUT/bench_infer.json 2>/dev/null; echo "infer exit $?"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_eager -o bench -- python $REPO/bench.py --steps 5 --warmup 3 --no-graph --no-cpu-baseline > $OUT/rocprof_eager.log 2>&1; echo "trace eager exit $?"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_graph -o bench -- python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/rocprof_graph.log 2>&1; echo "trace graph exit $?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py --steps 2 --warmup 3 --no-graph --no-cpu-baseline > $OUT/rocprof_fetch.log 2>&1; echo "pmc fetch exit $?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $REPO/bench.py --steps 2 --warmup 3 --no-graph --no-cpu-baseline > $OUT/rocprof_write.log 2>&1; echo "pmc write exit $?"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -o bench -- python $REPO/bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline > $OUT/rocprof_sq1.log 2>&1; echo "pmc sq1 exit $?"
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_sq2 -o bench -- python $REPO/bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline > $OUT/rocprof_sq2.log 2>&1; echo "pmc sq2 exit $?"
cd $REPO
S1=$(find $OUT/pmc_sq1 -name "*counter_collection.csv" | head -1); S2=$(find $OUT/pmc_sq2 -name "*counter_collection.csv" | head -1)
[ -n "$S1" ] && [ -n "$S2" ] && python tools/sq_summary.py "$S1" "$S2" $OUT/sq_counters.csv
rm -rf $OUT/pmc_sq1 $OUT/pmc_sq2
F=$(find $OUT/trace_eager -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/kernel_stats_eager.csv
G=$(find $OUT/trace_graph -name "*kernel_trace.csv" | head -1); [ -n "$G" ] && python tools/trace_steps.py "$G" 20 100 > $OUT/steady_state_per_step.txt
mkdir -p $OUT/pmc; find $OUT/pmc_fetch $OUT/pmc_write -name "*counter_collection.csv" | while read f; do cp "$f" $OUT/pmc/$(echo $f | grep -o "pmc_[a-z]*")_$(basename $f); done
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.csv
python tools/hbm_traffic.py $OUT/pmc_summary.csv $OUT/hbm_traffic.json BAT 48 8   # 3 warm-up + 2 timed + 3 roofline-profile eager steps
rm -rf $OUT/trace_eager $OUT/trace_graph $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc
head -3 $OUT/steady_state_per_step.txt
for f in bench bench_dense bench_p2b bench_p2b_b1 bench_m2track bench_nuscenes2048 bench_infer; do python - <<PY
import json
try:
    d=json.load(open("$OUT/$f.json")); r=d.get("roofline") or {}
    print("$f", d["value"], d["unit"], d["ms_per_step"], "frac", r.get("frac"), "live", r.get("live_fraction"), "cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"))
except Exception as e:
    print("$f", "ERR", e)
PY
done
