#!/bin/bash
# A/B on one box: tools/ab.sh "<env for B>" [reps] [bench args...] -> ms_per_step of A (no env) and B, alternating
envb="$1"; reps="${2:-3}"; shift; shift
ms() { python -c "import json,sys; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])"; }
for i in $(seq "$reps"); do
  a=$(timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | ms)
  b=$(env $envb timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | ms)
  echo "A $a   B($envb) $b"
done
