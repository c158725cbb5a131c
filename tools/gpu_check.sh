#!/bin/bash
# Runs on the GPU box through gpurun: environment probe, GPU tests, bench, rocprof summary.
# usage: tools/gpu_check.sh [tag]   -> everything is written under gpurun_out/<tag>/
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
{
  echo "== host"; nproc; lscpu | grep -E "Model name|^CPU\(s\)|Socket|Thread" ; free -g | head -2
  echo "== gpu"; rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit|Max Clock" | head -12
  rocm-smi --showmeminfo vram 2>/dev/null | head -8
} > $OUT/env.txt 2>&1
python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
tail -5 $OUT/smoke.log
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -25 $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
cat $OUT/bench.json; tail -5 $OUT/bench.err
