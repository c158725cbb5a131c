"""Tracking-inference latency: BAT forward in eval mode at batch 1 (the reference's
per-frame loop, models/base_model.py:59-86), eager vs one HIP graph replay.  GPU box only.
usage: python tools/bench_infer.py [--batch 1] [--iters 200]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from open3dsot_amd import synth, trackers

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--iters", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = trackers.BAT().to(dev).eval()
frames = [synth.to_torch(synth.make_batch(100 + i, a.batch), dev) for i in range(4)]


def fwd(b):
    with torch.no_grad():
        return model(b)["estimation_boxes"]


for i in range(5):
    fwd(frames[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.iters):
    fwd(frames[i % 4])
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / a.iters * 1e3
static = {k: v.clone() for k, v in frames[0].items()}
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    fwd(static)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = fwd(static)
torch.cuda.synchronize()
ref = fwd(frames[1]).clone()
for k, v in frames[1].items():
    static[k].copy_(v)
g.replay()
torch.cuda.synchronize()
same = bool(torch.allclose(out, ref, rtol=1e-5, atol=1e-6))
t0 = time.perf_counter()
for i in range(a.iters):
    for k, v in frames[i % 4].items():
        static[k].copy_(v, non_blocking=True)
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / a.iters * 1e3
print(json.dumps({"metric": "BAT eval forward latency, template 512 / search 1024", "batch": a.batch,
                  "eager_ms": round(eager, 3), "hip_graph_ms": round(graph, 3), "graph_matches_eager": same,
                  "frames_per_s_graph": round(a.batch / graph * 1e3, 1)}))
