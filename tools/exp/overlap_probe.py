"""Does a forked branch of a captured HIP graph overlap with the main branch on this machine?
FPS of a batch (latency bound: 96 waves, 0.27 ms) beside a chain of fp32 GEMMs.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from open3dsot_amd import ext, synth

dev = torch.device("cuda", 0)
b = synth.to_torch(synth.make_batch(0, 48), dev)
t, s = b["template_points"], b["search_points"]
A = torch.randn(2048, 2048, device=dev)
Bm = torch.randn(2048, 2048, device=dev)
side = torch.cuda.Stream()


def gemms(n=12):
    x = A
    for _ in range(n):
        x = torch.mm(x, Bm)
    return x


def fps():
    return ext.furthest_point_sampling_pair(t, 256, s, 512)


def serial():
    fps(); return gemms()


def forked():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        r = fps()
    g = gemms()
    main.wait_stream(side)
    return g, r


def timeit(fn, graph):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if graph:
        s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            fn()
        torch.cuda.current_stream().wait_stream(s2); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            keep = fn()
        run = g.replay
    else:
        run = fn
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 10.0


for graph in (False, True):
    print("graph" if graph else "eager", "fps only %.3f ms | gemms only %.3f ms | serial %.3f ms | forked %.3f ms" % (
        timeit(fps, graph), timeit(gemms, graph), timeit(serial, graph), timeit(forked, graph)))
