#!/bin/bash
# rocprofv3 passes over bench.py (eager launches): kernel-trace stats, then FETCH_SIZE, then WRITE_SIZE.
# usage: tools/gpu_prof.sh <tag> [bench args]
TAG=${1:-prof}; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-graph --no-cpu-baseline $@"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $REPO/bench.py $ARGS > $OUT/rocprof_trace.log 2>&1; echo "trace exit $?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline $@ > $OUT/rocprof_fetch.log 2>&1; echo "pmc fetch exit $?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python $REPO/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline $@ > $OUT/rocprof_write.log 2>&1; echo "pmc write exit $?"
cd $REPO
F=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp "$F" $OUT/kernel_stats.csv
mkdir -p $OUT/pmc; find $OUT/pmc_fetch $OUT/pmc_write -name "*counter_collection.csv" | while read f; do cp "$f" $OUT/pmc/$(echo $f | grep -o "pmc_[a-z]*")_$(basename $f); done
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_summary.csv; head -30 $OUT/pmc_summary.csv | cut -c1-220
find $OUT -name "*kernel_trace.csv" -size +8M -delete; rm -rf $OUT/pmc_fetch $OUT/pmc_write
ls -la $OUT
