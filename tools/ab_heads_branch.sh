#!/bin/bash
# same-box alternating A/B: only the heads' grouped weight gradients on the side branch.  tools/ab_heads_branch.sh [reps] [bench args]
reps="${1:-3}"; shift
ms() { grep '^{' | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in $(seq "$reps"); do
  a=$(timeout 300 python -c "
import sys; sys.argv=['bench.py','--steps','200','--warmup','10','--no-cpu-baseline','--no-secondary']+sys.argv[1:]
from open3dsot_amd import fused; fused._WGRAD_BRANCH['on']=True; fused._WGRAD_BRANCH['heads_only']=True
import bench; bench.main()" "$@" 2>/dev/null | ms)
  b=$(timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | ms)
  echo "A (heads' grouped wgrads on the side branch) $a   B (shipped) $b"
done
