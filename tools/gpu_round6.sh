#!/bin/bash
# One GPU call (gpurun).  usage: gpu_round6.sh <tag> [parts]   parts = any of: smoke tests ref bench qbench stats trace pmc sq probe:<name> ab:<env> hook:<module.dict.key>[:MODEL] k:<name+name+...> m:<MODEL>
# (default: smoke tests ref bench trace).  Everything lands in gpurun_out/<tag>/.
TAG=${1:-r4}; shift
PARTS="${*:-smoke tests ref bench trace}"
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
{ nproc; lscpu | grep -E "Model name|^CPU\(s\)|Socket|Thread"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit|Max Clock" | head -8; } > $OUT/env.txt 2>&1
rm -f $REPO/gpurun_out/flip_proof.txt
if has smoke; then
  timeout 400 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log
fi
for p in $PARTS; do case "$p" in k:*)
  timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider -s -k "$(echo "${p#k:}" | sed 's/+/ or /g')" > $OUT/pytest_k.log 2>&1
  echo "pytest -k exit $?" >> $OUT/pytest_k.log
  grep -E "passed|failed|^E  |exit|^FAILED|Error" $OUT/pytest_k.log | cut -c1-600 | head -60 ;;
esac; done
for p in $PARTS; do case "$p" in probe:*)
  ( cd $REPO && timeout 300 tools/exp/${p#probe:} > $OUT/probe_${p#probe:}.txt 2>&1; echo "probe exit $?" >> $OUT/probe_${p#probe:}.txt; tail -60 $OUT/probe_${p#probe:}.txt ) ;;
esac; done
if has tests; then
  timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider --durations=12 > $OUT/pytest.log 2>&1
  echo "pytest exit $?" >> $OUT/pytest.log
  grep -E "passed|failed|^E  |exit|^FAILED" $OUT/pytest.log | cut -c1-600 | head -60
  [ -f $REPO/gpurun_out/flip_proof.txt ] && cp $REPO/gpurun_out/flip_proof.txt $OUT/flip_proof.txt
fi
if has ref && [ -d $REPO/.refscratch ]; then
  O3D_REFERENCE_ROOT=$REPO/.refscratch timeout 900 python -m pytest tests/test_reference_modules_gpu.py -m gpu -v -s --timeout=600 -p no:cacheprovider > $OUT/reference_modules_over_hip_ext.log 2>&1
  echo "reference-modules pytest exit $?" | tee -a $OUT/reference_modules_over_hip_ext.log
  grep -E "PASSED|FAILED|SKIPPED|passed|failed|^E  |reference .* class over" $OUT/reference_modules_over_hip_ext.log | cut -c1-400 | head -30
fi
if has pmc; then
  cd /tmp
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py --steps 2 --warmup 3 --no-graph --no-cpu-baseline --no-secondary > $OUT/rocprof_fetch.log 2>&1; echo "pmc fetch exit $?"
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $REPO/bench.py --steps 2 --warmup 3 --no-graph --no-cpu-baseline --no-secondary > $OUT/rocprof_write.log 2>&1; echo "pmc write exit $?"
  cd $REPO
  mkdir -p $OUT/pmc; find $OUT/pmc_fetch $OUT/pmc_write -name "*counter_collection.csv" | while read f; do cp "$f" $OUT/pmc/$(echo $f | grep -o "pmc_[a-z]*")_$(basename $f); done
  python tools/pmc_summary.py $OUT/pmc $OUT/fabric_pmc_per_kernel.csv
  python tools/hbm_traffic.py $OUT/fabric_pmc_per_kernel.csv $OUT/hbm_traffic.json BAT 48 8   # 3 warm-up + 2 timed + 3 roofline-profile eager steps
  rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc
  # (on the box only: the bench part below then reports `traffic` from counters of exactly this library)
  [ -f $OUT/hbm_traffic.json ] && cp $OUT/hbm_traffic.json $REPO/profiles/hbm_traffic.json
fi
if has bench; then
  timeout 1200 python bench.py --per-launch $OUT/per_launch_roofline_bat.txt > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
  tail -3 $OUT/bench.err | cut -c1-300
fi
if has qbench; then
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --per-launch $OUT/per_launch_roofline_bat.txt > $OUT/bench.json 2> $OUT/bench.err; echo "qbench exit $?"
  tail -2 $OUT/bench.err | cut -c1-300
fi
for p in $PARTS; do case "$p" in hook:*)     # hook:<module>.<dict>.<key>[:MODEL] -> same-box A/B of a test hook (tools/ab_hook.py)
  H=${p#hook:}; HM=${H#*:}; [ "$HM" = "$H" ] && HM=BAT; H=${H%%:*}
  timeout 1200 python tools/ab_hook.py $H 3 --model $HM > $OUT/ab_hook_${H}_$HM.txt 2>&1; cat $OUT/ab_hook_${H}_$HM.txt | tail -4 ;;
esac; done
for p in $PARTS; do case "$p" in ab:*)
  bash tools/ab.sh "$(echo "${p#ab:}" | tr '+' ' ')" 2 > $OUT/ab_$(echo "${p#ab:}" | tr -c 'A-Za-z0-9_=' '_').txt 2>&1; cat $OUT/ab_$(echo "${p#ab:}" | tr -c 'A-Za-z0-9_=' '_').txt | tail -3 ;;
esac; done
if has stats; then     # the native rocprofv3 --kernel-trace --stats table of the bench command (40 timed steps + warm-up + capture)
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_run -o bench -- python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/rocprof_stats.log 2>&1; echo "stats exit $?"
  cd $REPO
  K=$(find $OUT/stats_run -name "*kernel_stats.csv" | head -1); [ -n "$K" ] && cp "$K" $OUT/rocprof_kernel_stats.csv && head -6 $OUT/rocprof_kernel_stats.csv | cut -c1-160
  rm -rf $OUT/stats_run
fi
if has trace; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_graph -o bench -- python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/rocprof_graph.log 2>&1; echo "trace graph exit $?"
  cd $REPO
  G=$(find $OUT/trace_graph -name "*kernel_trace.csv" | head -1); [ -n "$G" ] && python tools/trace_steps.py "$G" 20 100 > $OUT/steady_state_per_step.txt
  [ -n "$G" ] && python tools/trace_families.py $OUT/steady_state_per_step.txt > $OUT/steady_state_families.txt 2>/dev/null
  rm -rf $OUT/trace_graph
  head -4 $OUT/steady_state_per_step.txt | cut -c1-200
fi
for p in $PARTS; do case "$p" in m:*)
  M=${p#m:}; ML=$(echo $M | tr A-Z a-z)
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$ML -o bench -- python $REPO/bench.py --model $M --steps 40 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/rocprof_$ML.log 2>&1; echo "trace $M exit $?"
  cd $REPO
  G=$(find $OUT/trace_$ML -name "*kernel_trace.csv" | head -1); [ -n "$G" ] && python tools/trace_steps.py "$G" 20 100 > $OUT/steady_state_per_step_$ML.txt
  [ -n "$G" ] && python tools/trace_families.py $OUT/steady_state_per_step_$ML.txt > $OUT/steady_state_families_$ML.txt 2>/dev/null
  rm -rf $OUT/trace_$ML
  head -3 $OUT/steady_state_per_step_$ML.txt | cut -c1-200
  timeout 300 python bench.py --model $M --no-cpu-baseline --no-secondary --per-launch $OUT/per_launch_roofline_$ML.txt > $OUT/bench_$ML.json 2> $OUT/bench_$ML.err; echo "bench $M exit $?" ;;
esac; done
if has sq; then
  cd /tmp
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -o bench -- python $REPO/bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline --no-secondary > $OUT/rocprof_sq1.log 2>&1; echo "pmc sq1 exit $?"
  timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_sq2 -o bench -- python $REPO/bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline --no-secondary > $OUT/rocprof_sq2.log 2>&1; echo "pmc sq2 exit $?"
  cd $REPO
  S1=$(find $OUT/pmc_sq1 -name "*counter_collection.csv" | head -1); S2=$(find $OUT/pmc_sq2 -name "*counter_collection.csv" | head -1)
  [ -n "$S1" ] && [ -n "$S2" ] && python tools/sq_summary.py "$S1" "$S2" $OUT/sq_counters.csv
  rm -rf $OUT/pmc_sq1 $OUT/pmc_sq2
fi
python - <<PY
import json
try:
    d=json.load(open("$OUT/bench.json")); r=d.get("roofline") or {}
    print("BENCH", d["value"], d["unit"], d["ms_per_step"], "frac", r.get("frac"), "live", r.get("live_fraction"), "traffic", r.get("traffic"), r.get("traffic_note"))
    print("per_launch", r.get("per_launch_roofline"))
    c=d.get("cpu_baseline") or {}
    print("cpu", c.get("value"), c.get("cores"), c.get("timed_iterations"), "all_cores", (c.get("all_cores") or {}).get("value"), (c.get("all_cores") or {}).get("processes"))
    for k,v in (d.get("secondary") or {}).items(): print("  ", k, v if "error" in v or not isinstance(v, dict) else (v.get("value"), v.get("ms_per_step")))
except Exception as e:
    print("no bench line:", e)
PY
