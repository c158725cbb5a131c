#!/bin/bash
# tools/build_variant.sh <name> <source.hip> <extra hipcc flags...>
# A/B build: recompile ONE source of open3dsot_amd/csrc with extra flags (e.g. -DO3D_RING_FWD=3), link it with the
# regular objects into open3dsot_amd/_lib/libo3dsot_hip.<name>.so; select it with O3D_LIB_VARIANT=<name>
# (tools/ab.sh "O3D_LIB_VARIANT=<name>").  Build the regular library first (python -m open3dsot_amd.build).
set -e
name="$1"; src="$2"; shift; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
lib="$root/open3dsot_amd/_lib"
obj="$lib/${src%.hip}.$name.o"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function "$@" \
      -c "$root/open3dsot_amd/csrc/${SRC_OVERRIDE:-$src}" -o "$obj"
objs=$(ls "$lib"/*.o | grep -v "\.[A-Za-z0-9_]*\.o$" | grep -v "/${src%.hip}\.o$")
hipcc --offload-arch=gfx950 -shared -fPIC -o "$lib/libo3dsot_hip.$name.so" $objs "$obj"
echo "$lib/libo3dsot_hip.$name.so"
