"""Per-kernel time of the steady state from a rocprofv3 kernel trace: keeps the dispatches of the last
`frac` of the wall-clock span (after MIOpen's first-call searches and graph capture) and reports
ms per step.   usage: trace_last_step.py <kernel_trace.csv> <steps_in_window> [frac]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - (t1 - t0) * frac
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    if int(r["Start_Timestamp"]) >= cut:
        a = agg[r["Kernel_Name"]]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(a[1] for a in agg.values())
print("window %.1f ms, kernel time %.1f ms, steps %.1f -> %.3f ms kernel time per step" % ((t1 - cut) / 1e6, tot / 1e6, steps, tot / 1e6 / steps))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%8.3f ms/step %6.1f calls/step  %s" % (a[1] / 1e6 / steps, a[0] / steps, k[:120]))
