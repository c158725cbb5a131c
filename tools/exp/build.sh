#!/bin/bash
# builds the standalone probes (GPU-box micro-benchmarks, not part of the product)
cd "$(dirname "$0")/../.."
for f in tools/exp/*.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Wno-unused-result "$f" -o "${f%.hip}" \
    -Lopen3dsot_amd/_lib -lo3dsot_hip -Wl,-rpath,'$ORIGIN/../../open3dsot_amd/_lib' || exit 1
done
