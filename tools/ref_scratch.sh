#!/bin/bash
# Scratch copy of the nine reference files tests/test_reference_modules_gpu.py imports, so that the reference's OWN
# operator layer and tracker classes can be executed once on a GPU box over this repo's pointnet2_ops._ext
# (/root/reference itself does not travel).  .refscratch/ is git-ignored: the copy ships with the gpurun snapshot only,
# is never committed, and `tools/ref_scratch.sh clean` removes it after the run.   usage: ref_scratch.sh make|clean
set -e
cd "$(dirname "$0")/.."
SRC=${O3D_REFERENCE_SRC:-/root/reference}
case "$1" in
  make)
    rm -rf .refscratch
    for f in pointnet2/__init__.py pointnet2/utils/__init__.py pointnet2/utils/pytorch_utils.py \
             pointnet2/utils/pointnet2_utils.py pointnet2/utils/pointnet2_modules.py \
             models/backbone/pointnet.py models/head/rpn.py models/head/xcorr.py \
             models/base_model.py models/bat.py models/p2b.py; do
      if [ -f "$SRC/$f" ]; then mkdir -p ".refscratch/$(dirname "$f")"; cp "$SRC/$f" ".refscratch/$f"; fi
    done
    find .refscratch -type f | sort ;;
  clean) rm -rf .refscratch ;;
  *) echo "usage: $0 make|clean"; exit 1 ;;
esac
