#!/usr/bin/env python3
"""tools/asm_seq.py file.s <kernel-name-substring>: the order of memory instructions, waits and branches of one kernel
(runs of equal lines compressed) -- shows at a glance whether loads are batched ahead of their uses or serialised."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
for i, l in enumerate(lines):
    m = re.match(r"(_Z\w+):\s", l)
    if not m or want not in m.group(1):
        continue
    end = next(j for j in range(i + 1, len(lines)) if lines[j].startswith(".Lfunc_end"))
    f = [x.strip() for x in lines[i:end] if x.startswith("\t") and not x.startswith("\t.") and not x.startswith("\t;")]
    seq = []
    for x in f:
        if re.match(r"global_load|global_store|buffer_|s_waitcnt|s_cbranch|s_load|ds_|s_barrier|s_endpgm|scratch_", x):
            seq.append(x.split()[0] + (" " + x.split()[1] if x.startswith("s_waitcnt") else ""))
        elif "v_mfma" in x:
            seq.append("mfma")
        elif "dpp" in x:
            seq.append("dpp")
    out, prev, cnt = [], None, 0
    for x in seq + [None]:
        if x == prev:
            cnt += 1
        else:
            if prev:
                out.append("%s x%d" % (prev, cnt) if cnt > 1 else prev)
            prev, cnt = x, 1
    print(m.group(1)[:80], "(%d instructions)" % len(f))
    print(" | ".join(out))
    print()
