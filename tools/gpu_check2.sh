#!/bin/bash
# fused-kernel bring-up on the GPU box: unit tests first, then model parity, bench, profile
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_fused_gpu.py -m gpu -q --timeout=300 -p no:cacheprovider > $OUT/pytest_fused.log 2>&1
echo "pytest fused exit $?" >> $OUT/pytest_fused.log
grep -E "passed|failed|Error|assert|^E " $OUT/pytest_fused.log | head -60
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_index_ops_gpu.py -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_model.log 2>&1
echo "pytest model exit $?" >> $OUT/pytest_model.log
grep -E "passed|failed|^E " $OUT/pytest_model.log | head -30
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
cat $OUT/bench.json; grep -E "bench\]|Error|error" $OUT/bench.err | tail -8
timeout 600 python bench.py --steps 10 --warmup 3 --composed --no-cpu-baseline > $OUT/bench_composed.json 2> $OUT/bench_composed.err; echo "bench composed exit $?"
cat $OUT/bench_composed.json; grep -E "bench\]" $OUT/bench_composed.err | tail -3
