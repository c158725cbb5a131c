#!/bin/bash
# same-box alternating A/B of the plain step against the multi-GPU exchange path at world size 1 (bench.py --exchange): tools/ab_exchange.sh [reps]
reps="${1:-3}"
ms() { grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('max_parameter_divergence'), d['config'].get('exchange_path'), d['config'].get('rccl_world_size'))"; }
for i in $(seq "$reps"); do
  a=$(timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | ms)
  b=$(timeout 300 python bench.py --exchange --steps 200 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | ms)
  echo "plain $a   exchange $b"
done
