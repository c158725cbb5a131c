#!/bin/bash
# same-box: the secondary lines with the geometry prefetch on (default) and off
line() { grep '^{' | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(' '.join('%s=%s' % (k.replace('bat_','').replace('nuscenes_','ns_')[:16], v.get('ms_per_step')) for k,v in d.items() if 'infer' not in k))"; }
for rep in 1 2; do
  echo "geo ON : $(python bench.py --secondary-only --secondary-steps 40 --pool 4 2>/dev/null | line)"
  echo "geo OFF: $(python -c "
import sys; sys.argv=['bench.py','--secondary-only','--secondary-steps','40','--pool','4']
from open3dsot_amd import trackers; trackers._GEOMETRY_PREFETCH['on']=False
import bench; bench.main()" 2>/dev/null | line)"
done
