#!/usr/bin/env python3
"""tools/asm_waits.py file.s [...]: per kernel the static count of global loads / stores, `s_waitcnt vmcnt(0)` and
scratch (spill) instructions -- many full waits next to as many loads is the signature of a serialised
load -> wait -> use chain (what the direct GEMM epilogue had: HISTORY.md section 7)."""
import re
import sys

for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"(_Z\w+):\s", lines[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        end = next((j for j in range(i + 1, len(lines)) if lines[j].startswith(".Lfunc_end")), len(lines))
        f = [x.strip() for x in lines[i:end] if x.startswith("\t") and not x.startswith("\t.") and not x.startswith("\t;")]
        nl = sum(1 for x in f if x.startswith(("global_load", "buffer_load")))
        ns = sum(1 for x in f if x.startswith("global_store"))
        w0 = sum(1 for x in f if re.match(r"s_waitcnt vmcnt\(0\)", x))
        sc = sum(1 for x in f if x.startswith("scratch_"))
        tm = re.search(r"\d+([a-z_0-9]*kernel)(I(?:L[ib]\d+E)+E)?", name)
        label = name[:40]
        if tm:
            label = tm.group(1) + ("<" + ",".join(re.findall(r"L[ib](\d+)E", tm.group(2))) + ">" if tm.group(2) else "")
        print("%-14s %-40s instr %5d loads %3d stores %3d vmcnt(0) %3d scratch %d" % (
            path.split("/")[-1], label, len(f), nl, ns, w0, sc))
        i = end + 1
