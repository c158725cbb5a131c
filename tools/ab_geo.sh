#!/bin/bash
# same-box alternating A/B of the geometry prefetch: tools/ab_geo.sh [reps] [bench args...]
reps="${1:-3}"; shift
ms() { grep '^{' | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in $(seq "$reps"); do
  a=$(timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | ms)
  b=$(timeout 300 python tools/ab_hook.py --child trackers._GEOMETRY_PREFETCH.on 0 "$@" 2>/dev/null | ms)
  echo "A (geometry prefetch) $a   B (inline) $b"
done
