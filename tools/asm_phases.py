#!/usr/bin/env python3
"""tools/asm_phases.py file.s [template-args-substring ...]: per kernel that uses MFMA, the instruction count before the
first MFMA, between first and last, and after the last (prologue / main loop / epilogue) with a few instruction classes
of the epilogue -- a quick check that an epilogue edit did not bloat every instantiation."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2:]
i = 0
while i < len(lines):
    l = lines[i]
    m = re.match(r"(_Z\w+):\s", l)
    if not m:
        i += 1
        continue
    name = m.group(1)
    end = next(j for j in range(i + 1, len(lines)) if lines[j].startswith(".Lfunc_end"))
    tm = re.search(r"(\d+)([a-z_0-9]*kernel)I((?:L[ib]\d+E)+)E", name)
    label = (tm.group(2) + "<" + ",".join(re.findall(r"L[ib](\d+)E", tm.group(3))) + ">") if tm else name[:50]
    f = [x for x in lines[i:end] if x.startswith("\t") and not x.startswith("\t.") and not x.startswith("\t;")]
    mf = [k for k, x in enumerate(f) if "v_mfma" in x]
    if mf and (not want or any(w in label for w in want)):
        def cnt(pat, a=mf[-1], b=len(f)):
            return sum(1 for x in f[a:b] if re.search(pat, x))
        print("%-40s total %5d  pre %4d  loop %4d (%d mfma)  epi %5d [v_mov %d, loads %d, stores %d, scratch %d, waitcnt %d]" % (
            label, len(f), mf[0], mf[-1] - mf[0], len(mf), len(f) - mf[-1], cnt(r"v_mov"), cnt(r"global_load"),
            cnt(r"global_store"), cnt(r"scratch_"), cnt(r"s_waitcnt")))
    i = end + 1
