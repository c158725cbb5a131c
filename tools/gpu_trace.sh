#!/bin/bash
# steady-state per-step kernel breakdown (rocprofv3 kernel trace of a graph-replayed bench).  usage: gpu_trace.sh <tag> <model> [bench args]
TAG=${1:-trace}; MODEL=${2:-BAT}; shift; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$MODEL -o bench -- python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline --model $MODEL "$@" > $OUT/rocprof_$MODEL.log 2>&1; echo "trace exit $?"
cd $REPO
G=$(find $OUT/trace_$MODEL -name "*kernel_trace.csv" | head -1); [ -n "$G" ] && python tools/trace_steps.py "$G" 20 120 > $OUT/steady_state_$MODEL.txt
rm -rf $OUT/trace_$MODEL
head -3 $OUT/steady_state_$MODEL.txt | cut -c1-200
