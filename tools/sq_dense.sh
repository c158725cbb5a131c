set -x
REPO=$(pwd); OUT=$REPO/gpurun_out/r5q; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -o bench -- python $REPO/bench.py --dense --steps 2 --warmup 2 --no-graph --no-cpu-baseline --no-secondary > $OUT/rocprof_sq1.log 2>&1; echo "pmc sq1 exit $?"
timeout 600 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $OUT/pmc_sq2 -o bench -- python $REPO/bench.py --dense --steps 2 --warmup 2 --no-graph --no-cpu-baseline --no-secondary > $OUT/rocprof_sq2.log 2>&1; echo "pmc sq2 exit $?"
cd $REPO
S1=$(find $OUT/pmc_sq1 -name "*counter_collection.csv" | head -1); S2=$(find $OUT/pmc_sq2 -name "*counter_collection.csv" | head -1)
python tools/sq_summary.py "$S1" "$S2" $OUT/sq_counters_dense.csv
rm -rf $OUT/pmc_sq1 $OUT/pmc_sq2
timeout 300 python bench.py --dense --no-cpu-baseline --no-secondary --per-launch $OUT/per_launch_roofline_dense.txt > $OUT/bench_dense.json 2>/dev/null
