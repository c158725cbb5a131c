#!/usr/bin/env python3
"""tools/kernel_regs.py <remarks.txt> [...]: table of -Rpass-analysis=kernel-resource-usage remarks
(template arguments, VGPRs, AGPRs, scratch bytes, occupancy, spills) -- one column group per file."""
import re
import sys


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur:
            out[cur][m.group(1).strip()] = int(m.group(2))
    return out


def targs(mangled):
    m = re.search(r"kernelI((?:L[ib]\d+E)+)E", mangled)
    name = re.search(r"\d+([a-z_0-9]+kernel)", mangled)
    args = ",".join(re.findall(r"L[ib](\d+)E", m.group(1))) if m else ""
    return (name.group(1) if name else mangled[:40]) + "<" + args + ">"


tabs = [parse(p) for p in sys.argv[1:]]
for k in tabs[0]:
    cols = []
    for t in tabs:
        r = t.get(k, {})
        cols.append("v%3d a%3d scr%4d occ%d sp%d" % (r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("ScratchSize", -1),
                                                     r.get("Occupancy", -1), r.get("VGPRs Spill", -1)))
    print("%-44s %s" % (targs(k), " | ".join(cols)))
