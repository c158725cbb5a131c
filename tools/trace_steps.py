"""Steady-state per-step kernel breakdown from a rocprofv3 kernel trace of bench.py (graph mode):
the window between the loss-kernel launch of step -(n+1) and the loss-kernel launch of the last step.
usage: trace_steps.py <kernel_trace.csv> [n_steps=20] [top=60]"""
import csv, sys, collections
F = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20; top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows = list(csv.DictReader(open(F)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fps = [r for r in rows if "adam_step_kernel" in r["Kernel_Name"]]            # one per training step (every model)
if len(fps) <= steps:
    fps = [r for r in rows if "track_loss_sums_kernel" in r["Kernel_Name"]]  # BAT and P2B
if len(fps) <= steps:
    fps = [r for r in rows if "fps_reg_kernel<16" in r["Kernel_Name"]]
start = int(fps[-steps - 1]["Start_Timestamp"]); end = int(fps[-1]["Start_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    s = int(r["Start_Timestamp"])
    if start <= s < end:
        a = agg[r["Kernel_Name"]]; a[0] += 1; a[1] += int(r["End_Timestamp"]) - s
tot = sum(a[1] for a in agg.values())
# union of the kernel intervals: consecutive kernels of a replayed graph overlap (the next one is dispatched while the
# previous one drains), so the SUM of the durations overstates what the small kernels cost
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if start <= int(r["Start_Timestamp"]) < end)
busy, cur_s, cur_e = 0, None, None
for s_, e_ in iv:
    if cur_e is None or s_ > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
if cur_e is not None:
    busy += cur_e - cur_s
own = sum(a[1] for k, a in agg.items() if "anonymous namespace" in k and "at::native" not in k)
print("wall %.3f ms/step | GPU busy (union of kernel intervals) %.3f ms/step | sum of kernel durations %.3f ms/step | %d kernels/step | own kernels %.3f ms/step (%d launches)" % (
    (end - start) / 1e6 / steps, busy / 1e6 / steps, tot / 1e6 / steps, sum(a[0] for a in agg.values()) / steps, own / 1e6 / steps,
    sum(a[0] for k, a in agg.items() if "anonymous namespace" in k and "at::native" not in k) / steps))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%7.3f ms %5.1f x  %s" % (a[1] / 1e6 / steps, a[0] / steps, k[:140]))
