#!/usr/bin/env python
"""Same-box A/B of a module-level test hook: tools/ab_hook.py <module>.<dict>.<key> [reps] [bench args...]
alternates bench.py runs (one process each) with the hook on (A, the shipped default) and off (B) and prints ms_per_step
of each.  Example: tools/ab_hook.py fused_pointwise._POOLED_GMAX.on 3 --model M2TRACK"""
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(spec, value, argv):
    mod, dct, key = spec.rsplit(".", 2)
    getattr(importlib.import_module("open3dsot_amd." + mod), dct)[key] = bool(int(value))
    import bench
    sys.argv = ["bench.py", "--steps", "200", "--warmup", "10", "--no-cpu-baseline", "--no-secondary"] + argv
    bench.main()


def one(spec, value, argv):
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", spec, str(value)] + argv, cwd=ROOT,
                         capture_output=True, text=True, timeout=600).stdout
    return json.loads(out.strip().splitlines()[-1])["ms_per_step"]


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3], sys.argv[4:])
    else:
        spec = sys.argv[1]
        reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
        for _ in range(reps):
            a, b = one(spec, 1, sys.argv[3:]), one(spec, 0, sys.argv[3:])
            print("A (hook on) %.3f   B (%s off) %.3f" % (a, spec, b), flush=True)
