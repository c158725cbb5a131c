#!/bin/bash
# round 6: step time of the BAT bench with the geometry prefetch's in-place / stream-priority variants (same box)
run() { python -c "
import sys; sys.argv=['bench.py','--steps','200','--warmup','10','--no-cpu-baseline','--no-secondary']
from open3dsot_amd import dist, trackers
dist._PREFETCH['inplace']=bool($1); dist._PREFETCH['high_priority']=bool($2); trackers._GEOMETRY_PREFETCH['on']=bool($3)
import bench; bench.main()" 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for rep in 1 2; do
  echo "geometry off: $(run 0 0 0) | geo, copy, prio0: $(run 0 0 1) | geo, copy, prio-1: $(run 0 1 1) | geo, inplace, prio0: $(run 1 0 1) | geo, inplace, prio-1: $(run 1 1 1)"
done
