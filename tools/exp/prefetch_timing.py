#!/usr/bin/env python
"""GPU diagnostic (round 6): how long does the prefetch (FPS [+ geometry]) take on its stream while the step's graph replays,
and does the next step wait for it?  HIP events on both streams, 60 steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from open3dsot_amd import dist as D, synth, trackers

geo = int(sys.argv[1]) if len(sys.argv) > 1 else 1
trackers._GEOMETRY_PREFETCH["on"] = bool(geo)
D._PREFETCH["inplace"] = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = trackers.BAT().to(dev).train()
tr = D.DataParallelStep(model, world=1, graph=True, graph_warmup=2)
pool = [synth.to_torch(synth.make_batch(48 * s, 48), dev) for s in range(4)]
flat = False
for i in range(10):
    if tr.graph is not None and not flat:
        pool, flat = [tr.make_batch(b) for b in pool], True
    tr.step(pool[i % 4], next_batch=pool[(i + 1) % 4])
torch.cuda.synchronize()
rec = []
orig_prefetch, orig_replay = tr._prefetch, tr.graph.replay
def prefetch(nb):
    main = torch.cuda.current_stream()
    orig_prefetch(nb)
    # events: start = when the side stream may begin (recorded on main at the point wait_stream captured), end on side
    e1 = torch.cuda.Event(enable_timing=True); e1.record(tr._side)
    cur["p_end"] = e1
def replay():
    g0 = torch.cuda.Event(enable_timing=True); g1 = torch.cuda.Event(enable_timing=True)
    g0.record(); orig_replay(); g1.record()
    cur["g0"], cur["g1"] = g0, g1
tr._prefetch = prefetch
tr.graph.replay = replay
t_all0 = torch.cuda.Event(enable_timing=True); t_all0.record()
for i in range(60):
    cur = {}
    s0 = torch.cuda.Event(enable_timing=True); s0.record()
    tr.step(pool[i % 4], next_batch=pool[(i + 1) % 4])
    cur["s0"] = s0
    rec.append(cur)
t_all1 = torch.cuda.Event(enable_timing=True); t_all1.record()
torch.cuda.synchronize()
import statistics as st
from open3dsot_amd import fused
print("pair_geometry outputs:", fused._GEO_STATS, "| pool[1] is FlatBatch:", isinstance(pool[1], D.FlatBatch), "own keys:", len(getattr(pool[1], "extra_keys", ())), "takes out:", tr._sampling_takes_out)
gd = [r["g0"].elapsed_time(r["g1"]) for r in rec]
pe = [r["g0"].elapsed_time(r["p_end"]) for r in rec]          # prefetch end relative to the graph's start
gap = [rec[i]["g1"].elapsed_time(rec[i + 1]["g0"]) for i in range(len(rec) - 1)]   # end of graph t -> start of graph t+1
print("geometry prefetch %d inplace %d | step %.3f ms | graph replay %.3f ms (min %.3f max %.3f) | prefetch ends %.3f ms after the graph starts (max %.3f) | graph-to-graph gap %.3f ms (max %.3f)"
      % (geo, int(D._PREFETCH["inplace"]), t_all0.elapsed_time(t_all1) / 60, st.median(gd), min(gd), max(gd), st.median(pe), max(pe), st.median(gap), max(gap)))
