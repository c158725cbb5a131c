#!/bin/bash
# compact per-kernel resource usage: tools/kernel_usage.sh <file.hip> [extra flags]
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-function "$@" \
  -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/ku.o 2>&1 | \
  awk '/error|warning:/ {print} /Function Name/ {n=$NF; sub(/\[.*/,"",n); name=$(NF-1)} /VGPRs:/ {v=$(NF-1)} /AGPRs:/ {ag=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /Occupancy/ {oc=$(NF-1)} /SGPRs Spill/ {ss=$(NF-1)} /VGPRs Spill/ {vs=$(NF-1)} /LDS Size/ {print name, "vgpr="v, "agpr="ag, "scratch="sc, "occ="oc, "sgprspill="ss, "vgprspill="vs}' | \
  sed 's/_ZN12_GLOBAL__N_1[0-9]*//' | c++filt 2>/dev/null | cut -c1-150
