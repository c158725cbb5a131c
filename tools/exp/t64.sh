for X in 65536 300000 600000 1200000; do
  O3D_TILE64_MAX=$X timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --per-launch gpurun_out/pl_$X.txt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$X', d['ms_per_step'])"
done
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --model P2B --per-launch gpurun_out/pl_p2b.txt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('p2b', d['ms_per_step'])"
timeout 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --model M2TRACK --per-launch gpurun_out/pl_m2.txt 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('m2', d['ms_per_step'])"
