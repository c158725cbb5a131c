"""Which Python line launches which torch kernel: one eager training step under torch.profiler with
stacks, kernels grouped by (aten op, innermost frame inside this repo).  usage: trace_glue.py [BAT|P2B] [top]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from open3dsot_amd import dist as D, synth, trackers  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "BAT"
top = int(sys.argv[2]) if len(sys.argv) > 2 else 70
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = trackers.get_model(name)().to(dev).train()
trainer = D.DataParallelStep(model, world=1, graph=False)
batch = synth.to_torch(synth.make_batch(0, 48), dev)


def step():
    trainer._forward_backward(batch)
    trainer.reduce_gradients()
    trainer.optimizer.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()

agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    ks = getattr(e, "kernels", None)
    if not ks:
        continue
    frame = "?"
    for f in (e.stack or []):
        if "/open3dsot_amd/" in f or "bench.py" in f or "/tools/" in f:
            frame = f.split("/root/repo/")[-1] if "/root/repo/" in f else f[-70:]
            break
    for k in ks:
        a = agg[(e.name, frame)]
        a[0] += 1
        a[1] += k.duration
tot = sum(a[0] for a in agg.values())
print("%d kernels launched by aten ops in one step" % tot)
for (op, frame), a in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print("%4d x %8.1f us  %-28s %s" % (a[0], a[1], op[:28], frame[:110]))
