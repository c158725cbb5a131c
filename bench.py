#!/usr/bin/env python
"""bench.py -- template/search pairs per second (forward + backward + Adam) of the BAT tracker
at KITTI-Car shapes (template 512 / search 1024 points), BASELINE.json's metric.

    python bench.py --gpus 1 --steps K --warmup W          (one MI355X)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N ...        (no torchrun environment: spawns the N ranks itself, like the
                                         reference's `pl.Trainer(gpus=-1, accelerator='ddp')`, main.py:53-64,82)

One "step" = one full training step of the hot path on one synthetic batch of `--batch`
pairs PER GPU (weak scaling; default 48 = BASELINE config 2): FPS, ball queries, the fused
grouped-MLP stack, BoxCloud xcorr, vote heads, losses, backward, the one-message RCCL
gradient all-reduce and the Adam update.  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line (contract in the task statement) that also carries
  "roofline"     -- the dominant kernel family (fp32-MFMA grouped-MLP GEMMs) measured live
                    with HIP events on the launch stream: algorithmic FLOPs / time vs the
                    157.3 TFLOP/s dense fp32 matrix peak of gfx950;
  "cpu_baseline" -- the oracle's PyTorch restatement of the same step timed on this host's
                    cores on a bounded sample (kind "port": the reference has no CPU path).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from open3dsot_amd import dist as D  # noqa: E402
from open3dsot_amd import synth, trackers  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak (= vector peak)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=48, help="pairs per GPU (BASELINE config 2: 48)")
    ap.add_argument("--model", default="BAT", choices=["BAT", "P2B", "M2TRACK"])
    ap.add_argument("--pool", type=int, default=4, help="distinct resident synthetic batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=150.0, help="seconds of host time the CPU baseline may take")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short driver-visible lines of the other BASELINE "
                    "configs (P2B batch 48 / batch 1, M2-Track, tracking inference) appended to the main line as `secondary`")
    ap.add_argument("--secondary-steps", type=int, default=20)
    ap.add_argument("--secondary-only", action="store_true", help=argparse.SUPPRESS)     # the child process of `secondary`
    ap.add_argument("--infer", action="store_true", help="tracking-inference latency instead of the training step: eval-mode "
                    "forward of one frame (SURVEY.md section 8f-4, models/base_model.py:59-86), one HIP graph replay per frame")
    ap.add_argument("--infer-batch", type=int, default=1, help="frames per forward in --infer mode (the reference tracks at 1)")
    ap.add_argument("--search-size", type=int, default=1024, help="search-cloud points (BASELINE config 5, BAT_CAR_NUSCENES: 2048)")
    ap.add_argument("--dense", action="store_true", help="worst-case clouds: every ball full of distinct neighbours "
                    "(live_fraction 1.0) instead of the KITTI-like crops")
    ap.add_argument("--per-launch", default=None, metavar="FILE", help="write the per-launch roofline table of the GEMM "
                    "launches (algorithmic FLOPs / bytes, HIP-event time, share of max(MFMA, HBM) roofline) to FILE")
    ap.add_argument("--exchange", action="store_true", help="run the MULTI-GPU code path at --gpus 1: a one-rank RCCL process "
                    "group, gradient pack inside the captured graph, all_reduce(AVG) on the flat buffer, FlatAdam on its views "
                    "(DataParallelStep(exchange=True)); prices the exchange path's per-step overhead on a one-GPU box")
    ap.add_argument("--composed", action="store_true", help="disable the fused kernels (debug A/B only)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying one HIP graph")
    return ap.parse_args()


def mlp_flops_per_pair(model_name):
    """Algorithmic forward FLOPs per pair of every 1x1-conv layer: 2*Cin*Cout*positions
    (BAT 5.462 GFLOP, P2B 8.462 GFLOP)."""
    def mlp(spec, pos):
        return sum(2 * a * b * pos for a, b in zip(spec[:-1], spec[1:]))
    total = 0
    for M in (512, 1024):
        total += mlp([3, 64, 64, 128], (M // 2) * 32)
        total += mlp([131, 128, 128, 256], (M // 4) * 32)
        total += mlp([259, 256, 256, 256], (M // 8) * 32)
        total += mlp([256, 256], M // 8)                      # conv_final
    total += mlp([260, 256, 256, 256], 64 * 16)               # vote aggregation SA
    total += mlp([256, 256, 256, 1], 128) + mlp([259, 256, 256, 259], 128) + mlp([256, 256, 256, 5], 64)
    if model_name == "BAT":
        total += mlp([259, 256, 256, 9], 128)                 # mlp_bc
        total += mlp([268, 256, 256, 256], 128 * 4) + mlp([256, 256, 256], 128)
    else:
        total += mlp([260, 256, 256, 256], 64 * 128) + mlp([256, 256, 256], 128)
    return total


def _cpu_step_fn(model_name, sd):
    """-> step(batch) : one fwd+bwd+Adam iteration of the oracle restatement on the host, seconds"""
    from oracle import torch_ref
    w = {k: v for k, v in (trackers.BAT_CAR if model_name == "BAT" else trackers.P2B_CAR).items() if k.endswith("_weight")}
    params = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in sd.items()}
    opt = torch.optim.Adam([v for v in params.values() if v.requires_grad], lr=1e-3, betas=(0.5, 0.999), eps=1e-6)

    def step(batch):
        t0 = time.perf_counter()
        opt.zero_grad()
        out = (torch_ref.bat_forward if model_name == "BAT" else torch_ref.p2b_forward)(params, batch, True)
        loss, _ = torch_ref.matching_loss(batch, out, w, bat=model_name == "BAT")
        loss.backward()
        opt.step()
        return time.perf_counter() - t0
    return step


def _cpu_worker(rank, model_name, sd, batch_size, threads, iters, barrier, out):
    """one replica of the process-parallel CPU figure: `threads` cores, its own batches, 1 warm-up + `iters` timed"""
    torch.set_num_threads(threads)
    step = _cpu_step_fn(model_name, sd)
    step(synth.to_torch(synth.make_batch(7000 + 1000 * rank, batch_size)))
    batches = [synth.to_torch(synth.make_batch(7100 + 1000 * rank + it * batch_size, batch_size)) for it in range(iters)]
    barrier.wait()
    t0 = time.perf_counter()
    for b in batches:
        step(b)
    out.put((rank, time.perf_counter() - t0))


def cpu_process_parallel(model_name, sd, batch_size, threads, cores, iters=2, timeout_s=120.0):
    """the same-host figure that uses ALL physical cores: cores // threads independent replicas of the port (one process
    each, `threads` intra-op threads, its own batch-`batch_size` stream: what a CPU data-parallel run of the reference
    would be), started together behind a barrier; value = all pairs / the slowest replica's time"""
    import torch.multiprocessing as mp
    nproc = max(1, cores // threads)
    ctx = mp.get_context("spawn")
    barrier, out = ctx.Barrier(nproc), ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(r, model_name, sd, batch_size, threads, iters, barrier, out), daemon=True)
             for r in range(nproc)]
    for pr in procs:
        pr.start()
    times, deadline = [], time.perf_counter() + timeout_s
    while len(times) < nproc:
        try:
            times.append(out.get(timeout=1.0)[1])
        except Exception:      # nothing yet: a replica that died (or the deadline) ends the measurement without a figure
            if time.perf_counter() > deadline or any(pr.exitcode not in (None, 0) for pr in procs):
                for pr in procs:
                    if pr.is_alive():
                        pr.terminate()
                return None
    for pr in procs:
        pr.join(timeout=10)
    return {"value": round(nproc * iters * batch_size / max(times), 3), "unit": "pairs/s", "processes": nproc,
            "threads_per_process": threads, "cores": nproc * threads, "iterations_per_process": iters,
            "sample": "%d processes x %d threads, each %d timed batch-%d steps after 1 warm-up, started behind a barrier; "
                      "all pairs / slowest process" % (nproc, threads, iters, batch_size)}


def cpu_baseline(model_name, sd, batch_size, budget_s=150.0):
    """Oracle restatement (oracle/torch_ref.py) of the same training step on the host cores (kind "port": the reference
    has no CPU path).  More threads than the problem has parallel work THRASH (round 1's 128-thread figure was 4x slower
    than 8 threads), so the thread count is swept on a batch-8 probe first; the best count then runs the bench batch
    itself -- SURVEY.md section 8d: 3 warm-up (2 when the budget is short) + >= 10 timed iterations, median -- and batch 1
    (3 + 10); last, the process-parallel figure over all physical cores (cpu_process_parallel)."""
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count() or 1
    except ImportError:
        cores = os.cpu_count() or 1
    t_begin = time.perf_counter()
    step = _cpu_step_fn(model_name, sd)
    probe = [synth.to_torch(synth.make_batch(5000 + 8 * i, 8)) for i in range(2)]
    sweep = {}
    for n in sorted({c for c in (4, 8, 16, 32) if c <= cores} | ({cores} if cores <= 16 else set())):
        torch.set_num_threads(n)
        step(probe[0])                                   # warm-up at this thread count
        sweep[n] = round(8 / min(step(probe[1]), step(probe[0])), 2)
        if time.perf_counter() - t_begin > 0.12 * budget_s:
            break

    def sample(bs, warm, iters, until):
        times = []
        for it in range(warm + iters):
            dt = step(synth.to_torch(synth.make_batch(6000 + it * bs, bs)))
            if it >= warm:
                times.append(dt)
            if len(times) >= 3 and time.perf_counter() > until:
                break
        return bs / sorted(times)[len(times) // 2], len(times)

    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    vfull, nfull = sample(batch_size, 2, 10, t_begin + 0.7 * budget_s)
    v1, n1 = sample(1, 3, 10, t_begin + 0.75 * budget_s)
    par = None
    if cores >= 16:      # every physical core: 8 processes x cores/8 threads (its own time limit: 1 warm-up + 2 steps each, all
        par = cpu_process_parallel(model_name, sd, batch_size, cores // 8, cores, timeout_s=180.0)      # processes at once)
    res = {"value": round(vfull, 3), "unit": "pairs/s", "cores": best, "kind": "port",
           "batch1_value": round(v1, 3), "host_physical_cores": cores, "thread_sweep_batch8_pairs_per_s": sweep,
           "timed_iterations": nfull,
           "sample": "%s fwd+bwd+Adam on oracle/torch_ref.py (C index ops + PyTorch fp32 CPU convs); threads swept on a "
                     "batch-8 probe, the best (%d threads) timed on batch %d: median of %d iterations after 2 warm-up; "
                     "batch 1: median of %d after 3 warm-up" % (model_name, best, batch_size, nfull, n1)}
    if par is not None:
        res["all_cores"] = par
    return res


def _flush_c_stdio():
    """RCCL prints a start-up banner through C stdio; into a pipe that is fully buffered and would come out at process
    exit, AFTER the JSON line.  Flushed on every rank once the communicator exists, and again before the line is printed."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def _spawned_rank(rank, args, port):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    run(args)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: one process per GPU, spawned here (the reference's Trainer does the same
        # with gpus=-1 / accelerator='ddp', main.py:53-64,82)
        if not torch.cuda.is_available() or torch.cuda.device_count() < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible -- refusing to report a smaller job" %
                             (args.gpus, torch.cuda.device_count() if torch.cuda.is_available() else 0))
        import socket
        import torch.multiprocessing as mp
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        mp.spawn(_spawned_rank, args=(args, port), nprocs=args.gpus, join=True)
        return
    run(args)


def run_infer(args):
    """`--infer`: eval-mode forward latency of one tracked frame.  The reference's per-frame loop (models/base_model.py:
    59-86) is strictly sequential (frame t's search region depends on frame t-1's box), so the figure of merit is the
    latency of ONE batch-1 forward: BatchNorm on running statistics, no autograd, the whole forward (FPS, ball queries,
    fused MLPs, xcorr, heads) replayed as one HIP graph.  Inputs resident in HBM; `steps` frames are timed."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library is the only compute path (no CPU fallback)")
    from open3dsot_amd import capi
    capi.load()
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    B = args.infer_batch
    if args.model == "M2TRACK":
        # the network half of BaseModel.evaluate_one_sample for the motion tracker (models/base_model.py:44-57 on the input
        # of MotionBaseModel.build_input_dict, :255-304): the two-stage forward on the stacked previous / current crops ->
        # the refined box (x, y, z, theta)
        from open3dsot_amd import m2track
        model = m2track.M2TRACK().to(dev).eval()
        keep = ("points", "candidate_bc")
        frames = [{k: v for k, v in synth.to_torch(synth.make_motion_batch(100 + i * B, B, 1024), dev).items() if k in keep}
                  for i in range(max(2, args.pool))]
        shapes = "2 x 1024 pts"

        def fwd(b):
            with torch.no_grad():
                return (model(b)["estimation_boxes"],)
    else:
        model = trackers.get_model(args.model)().to(dev).eval()
        frames = [synth.to_torch(synth.make_batch(100 + i * B, B), dev) for i in range(max(2, args.pool))]
        shapes = "512/1024 pts"

        def fwd(b):      # the network half of evaluate_one_sample (models/base_model.py:44-57): forward + the best proposal's
            with torch.no_grad():      # (x, y, z, theta), selected on the device (no (64,5) copy to the host)
                return model.evaluate_one_sample(b)

    for i in range(max(args.warmup, 3)):
        fwd(frames[i % len(frames)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_eager = min(args.steps, 100)
    for i in range(n_eager):
        fwd(frames[i % len(frames)])
    torch.cuda.synchronize()
    eager_ms = (time.perf_counter() - t0) / n_eager * 1e3
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
    ref = [t.clone() for t in fwd(frames[1])]
    for k, v in frames[1].items():
        static[k].copy_(v)
    g.replay()
    torch.cuda.synchronize()
    same = all(torch.allclose(a.float(), b.float(), rtol=1e-5, atol=1e-6) for a, b in zip(out, ref))
    for i in range(args.warmup):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        for k, v in frames[i % len(frames)].items():
            static[k].copy_(v, non_blocking=True)
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    return {
        "metric": "tracked frames/sec (eval forward, %s KITTI-Car %s, batch %d)" % (model.__class__.__name__, shapes, B),
        "value": round(B / ms * 1e3, 1), "unit": "frames/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic KITTI-Car-like pairs (open3dsot_amd/synth.py), random-init weights, BatchNorm on running statistics",
        "config": {"workload": "%s tracking inference, KITTI-Car %s, batch %d, eval forward only, fp32"
                               % (model.__class__.__name__, shapes, B), "hip_graph": True, "eager_ms_per_frame": round(eager_ms, 4),
                   "best_proposal_on_device": args.model != "M2TRACK", "graph_replay_matches_eager": bool(same)}}


def run(args):
    if args.secondary_only:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU")
        torch.cuda.set_device(0)
        print(json.dumps(secondary_lines(args)))
        return
    if args.infer:
        print(json.dumps(run_infer(args)))
        return
    rank, local_rank, world = D.init_distributed()
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE is %d: launch with torch.distributed.run "
                         "--nproc-per-node %d (or without a torchrun environment to let bench.py spawn the ranks)" %
                         (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library is the only compute path (no CPU fallback)")
    line = measure(args, rank, local_rank, world, full=True)
    if rank == 0:
        default_cfg = (args.model == "BAT" and not args.dense and args.search_size == 1024 and world == 1 and
                       not args.composed and not args.no_graph)
        if default_cfg and not args.no_secondary:
            # in a child process: a fault in one of the short side runs must not take the main line down
            import subprocess
            try:
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--secondary-only", "--secondary-steps",
                                     str(args.secondary_steps), "--pool", str(args.pool)], capture_output=True, text=True,
                                    timeout=420)
                rows = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
                line["secondary"] = json.loads(rows[-1]) if rows else {"error": "exit %d: %s" % (cp.returncode, cp.stderr[-300:])}
            except Exception as e:
                line["secondary"] = {"error": "%s: %s" % (type(e).__name__, e)}
    _flush_c_stdio()                # (RCCL's start-up banner, if it is still buffered, goes out BEFORE the line)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist.is_initialized():       # after the line: a teardown that hangs or aborts must not cost the measurement
        dist.destroy_process_group()


def secondary_lines(args):
    """Short driver-visible lines of the other BASELINE configs (the driver runs ONE command): P2B at batch 48 and at
    batch 1 (config 1's shape on the GPU), M2-Track (config 4), BAT at the NuScenes shapes (config 5: search 2048, and the
    YAML's per-GPU batch 100 at search 1024), the worst-case dense clouds, and the tracking-inference latency (section
    8f-4).  Each: `--secondary-steps` timed steps after the graph warm-up, same code path as the main line."""
    import copy
    out = {}

    def one(tag, **over):
        a = copy.copy(args)
        a.steps, a.warmup, a.no_cpu_baseline, a.per_launch = args.secondary_steps, 5, True, None
        over = dict(over)
        a.full = over.pop("full", False)
        for k, v in over.items():
            setattr(a, k, v)
        try:
            if over.get("infer"):
                a.steps, a.warmup = 200, 20
                r = run_infer(a)
            else:
                r = measure(a, 0, 0, 1, full=a.full)
            out[tag] = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                        "workload": r["config"]["workload"], "hip_graph": bool(r["config"].get("hip_graph"))}
            for k in ("rccl_world_size", "max_parameter_divergence", "gradient_exchange", "exchange_path", "self_check_error"):
                if over.get("exchange") and k in r["config"]:
                    out[tag][k] = r["config"][k]
            if a.full and r.get("roofline"):       # the worst-case line carries its own roofline fractions
                out[tag]["roofline"] = {k: r["roofline"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac",
                                                                           "whole_step_frac", "live_fraction")}
        except Exception as e:  # a secondary line must never take the main line down
            out[tag] = {"error": "%s: %s" % (type(e).__name__, e)}
        if over.get("exchange") and dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:
                pass
        torch.cuda.empty_cache()
    one("p2b_batch48", model="P2B")
    one("p2b_batch1", model="P2B", batch=1)
    one("m2track_batch48", model="M2TRACK")
    one("bat_nuscenes_search2048_batch48", search_size=2048)
    one("bat_nuscenes_yaml_batch100", batch=100)
    one("bat_nuscenes_search2048_batch100", search_size=2048, batch=100)       # config 5 as one workload
    one("bat_dense_worst_case", dense=True, full=True)
    # the multi-GPU step at world size 1: what the exchange path (pack in the graph + all_reduce(AVG) + FlatAdam on views of
    # the exchange buffer) costs per step beside the headline; NOT a scaling measurement
    one("bat_rccl_world1_exchange", exchange=True)
    one("bat_infer_batch1", infer=True)
    one("p2b_infer_batch1", infer=True, model="P2B")
    one("m2track_infer_batch1", infer=True, model="M2TRACK")     # (last: a fault while capturing it cannot touch the lines above)
    return out


def measure(args, rank, local_rank, world, full=True):
    """one timed run of the training step described by `args`; -> the JSON line as a dict (rank 0; {} elsewhere).
    full=False: no roofline instrumentation and no CPU baseline (the `secondary` lines)."""
    dev = torch.device("cuda", local_rank)
    from open3dsot_amd import capi, sa_modules
    capi.load()
    sa_modules.set_fused(not args.composed)

    torch.manual_seed(1234)
    if args.model == "M2TRACK":      # BASELINE config 4 (parity case; no pointnet2 operator on this path)
        from open3dsot_amd import m2track
        model = m2track.M2TRACK().to(dev).train()
        make = lambda first, n: synth.make_motion_batch(first, n, 1024)
        args.no_cpu_baseline = True
    else:
        model = trackers.get_model(args.model)().to(dev).train()
        base_make = synth.make_dense_batch if args.dense else synth.make_batch
        make = lambda first, n: base_make(first, n, 512, args.search_size)
    sd_cpu = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    exchange = bool(getattr(args, "exchange", False))
    if exchange and world == 1 and not dist.is_initialized():
        import socket
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        os.environ.update(RANK="0", LOCAL_RANK=str(local_rank), WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                          MASTER_PORT=str(sock.getsockname()[1]))
        sock.close()
        D.init_distributed(force=True)          # a one-rank RCCL communicator
    trainer = D.DataParallelStep(model, world=world, graph=not args.no_graph, graph_warmup=2,
                                 exchange=True if exchange else None)

    # resident synthetic batches (distinct per rank and per pool slot)
    pool = []
    for s in range(args.pool):
        first, n = D.shard_indices(s, rank, world, args.batch)
        pool.append(synth.to_torch(make(first, n), dev))
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    tw = time.perf_counter()
    # (next_batch: the loop knows which resident batch comes next, like a data loader with one batch of look-ahead; its
    # farthest-point sampling -- a function of the input clouds only -- then runs beside the current step)
    nwarm = max(args.warmup, 8 if not args.no_graph else 0)
    flat = False
    for i in range(nwarm):   # graph capture happens in here
        if trainer.graph is not None and not flat:
            # the resident batches in the layout of the captured step's static inputs: one device copy per step instead
            # of one copy node per field (what a loader that fills one staging buffer per batch gives)
            pool, flat = [trainer.make_batch(b) for b in pool], True
        trainer.step(pool[i % len(pool)], next_batch=pool[(i + 1) % len(pool)] if i + 1 < nwarm else pool[0])
    torch.cuda.synchronize()
    _flush_c_stdio()
    log("warm-up: %.2f s; hip graph: %s %s" % (time.perf_counter() - tw, trainer.graph is not None,
                                              trainer.graph_error or ""))
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        trainer.step(pool[i % len(pool)], next_batch=pool[(i + 1) % len(pool)])
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    log("timed %d steps: %.3f s" % (args.steps, elapsed))
    multi = None
    if world > 1:
        # self-checks of the multi-GPU run (the driver computes the scaling efficiency itself; these only say whether the
        # run was the run it claims to be): every rank's own rate, and how far the replicas' parameters are apart after
        # the timed steps -- identical gradients after the all-reduce and the same Adam update must leave them bitwise equal
        # The timing exchange first; the self-check is a fixed sequence of collectives on every rank (its fallible local
        # work happens before them and the ranks agree on an ok flag: dist.replica_self_check) -- no per-rank try/except
        # around collectives, which would leave the other ranks blocked in them
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        own_elapsed, elapsed = elapsed, float(t.item())
        multi = D.replica_self_check(model, trainer, own_elapsed, args.batch * args.steps)
    elif exchange:
        multi = D.replica_self_check(model, trainer, elapsed, args.batch * args.steps)
        multi["exchange_path"] = bool(trainer.exchange and all(p.grad is v for p, v in zip(trainer.grads.params, trainer.grads.views)))

    # ---- roofline of the dominant kernel family, measured live with HIP events ------------
    roofline = None
    try:
        from open3dsot_amd import fused
        if full and sa_modules.fused_enabled() and hasattr(fused, "profile_step"):
            def eager_step():      # event-bracketed launches cannot be replayed from a graph
                trainer._forward_backward(pool[0])
                trainer.reduce_gradients(); trainer.optimizer.step()
            roofline = fused.profile_step(eager_step, PEAK_FP32_MFMA_TFLOPS)
    except ImportError:
        roofline = None
    if roofline is not None:
        rows = roofline.pop("per_launch", None) or []
        if rows:          # both rooflines per launch, summed: the time the step's GEMM launches would take AT the roofline
            roof_ms, ms = sum(r["roof_ms"] for r in rows), sum(r["ms"] for r in rows)
            roofline["per_launch_roofline"] = {"launches": len(rows), "ms": round(ms, 4), "roof_ms": round(roof_ms, 4),
                                               "frac": round(roof_ms / ms, 4),
                                               "hbm_bound_launches": sum(1 for r in rows if r["bound"] == "hbm")}
            for fam, pick in (("grouped_mlp", lambda k: not k.startswith("pw_") and not k.endswith("_points")),
                              ("per_point_layer0", lambda k: k.endswith("_points")), ("heads", lambda k: k.startswith("pw_"))):
                sel = [r for r in rows if pick(r["kernel"])]
                if sel:
                    roofline["per_launch_roofline"][fam] = {
                        "launches": len(sel), "ms": round(sum(r["ms"] for r in sel), 4),
                        "frac": round(sum(r["roof_ms"] for r in sel) / sum(r["ms"] for r in sel), 4)}
        if args.per_launch and rank == 0:
            with open(args.per_launch, "w") as fh:
                fh.write("kernel i Cin Cout cols ms TFLOP/s GB/s bound roof_ms frac\n")
                for r in rows:
                    fh.write("%-18s %2d %4d %4d %8d %.4f %6.1f %6.0f %-4s %.4f %.3f\n" % (
                        r["kernel"], r["i"], r["Cin"], r["Cout"], r["cols"], r["ms"], r["tflops"], r["gbs"], r["bound"],
                        r["roof_ms"], r["frac"]))
    if roofline is not None and args.model != "M2TRACK" and args.search_size == 1024:
        # the reference gathers first (layer 0 on npoint*nsample positions): rate in those terms as well
        ref_gflop = 3.0 * mlp_flops_per_pair(args.model) * args.batch / 1e9
        roofline["reference_formula_gflop_per_step"] = round(ref_gflop, 2)
        roofline["reference_formula_tflops"] = round(ref_gflop / roofline["gemm_ms_per_step"], 3)
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # rocprofv3 --pmc passes (tools/gpu_prof.sh)
        if os.path.exists(tfile):
            t = json.load(open(tfile))
            from open3dsot_amd import build as _build
            if t.get("kernel_source_sha256") != _build.source_hash():
                # the counters were taken on other kernels than the ones that just ran: no number rather than a stale one
                roofline["traffic_note"] = ("profiles/hbm_traffic.json was recorded for kernel sources %s, this library is "
                                            "built from %s: not reported" % (str(t.get("kernel_source_sha256"))[:12],
                                                                            _build.source_hash()[:12]))
            elif t.get("workload_batch") == args.batch and t.get("model") == args.model:
                # bytes per STEP of the family's device kernels (tools/hbm_traffic.py, same symbol list as the launches
                # counted here) over THIS line's launch count: traffic x launches_per_step = the PMC table's family rows
                roofline["traffic_bytes_per_step"] = t["gemm_family_bytes_per_step"]
                roofline["traffic"] = int(t["gemm_family_bytes_per_step"] / max(1, roofline["launches_per_step"]))
                roofline["traffic_source"] = t["source"]
                # the same launches against the other roof.  FETCH_SIZE / WRITE_SIZE count the L2's memory-side (fabric)
                # requests, Infinity-Cache hits included (MI355X_MICROARCH.md, HBM section): an UPPER bound of the HBM
                # bytes, so the fraction below is "fabric traffic against the HBM peak", not proven HBM bandwidth
                gbs = roofline["traffic"] / (roofline["avg_launch_ms"] * 1e-3) / 1e9
                roofline["fabric_gbs"] = round(gbs, 1)
                roofline["fabric_frac_of_hbm_peak"] = round(gbs / PEAK_HBM_GBS, 4)
    if roofline is not None and roofline.get("gemm_gflop_per_step"):
        # the same executed FLOPs over the WHOLE timed step (every kernel, launch gaps included): what the step as a unit
        # reaches of the matrix peak, beside `frac` (the GEMM launches' own time)
        step_ms = 1e3 * elapsed / args.steps
        roofline["whole_step_frac"] = round(roofline["gemm_gflop_per_step"] / step_ms / PEAK_FP32_MFMA_TFLOPS, 4)
        if roofline.get("reference_formula_gflop_per_step"):
            # SURVEY 8(d)'s gather-first formula over the FLOPs the kernels execute (distinct-neighbour compaction, layer 0
            # on points): > 1 means the formula's work is NOT what runs; credit is on executed FLOPs only
            roofline["formula_over_executed"] = round(roofline["reference_formula_gflop_per_step"] / roofline["gemm_gflop_per_step"], 3)
    if roofline is None and full:  # no instrumented kernels yet: whole-step algorithmic rate (labelled as such)
        flops = 3.0 * mlp_flops_per_pair(args.model) * args.batch
        ach = flops / (elapsed / args.steps) / 1e12
        roofline = {"bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
                    "kernel": "whole training step (3x forward conv FLOPs / step time)"}

    if rank == 0:
        pairs = world * args.batch * args.steps
        line = {
            "metric": {"BAT": "template/search pairs/sec (fwd+bwd), BAT KITTI-Car 512/%d pts" % args.search_size,
                       "P2B": "template/search pairs/sec (fwd+bwd), P2B KITTI-Car 512/%d pts" % args.search_size,
                       "M2TRACK": "frame pairs/sec (fwd+bwd), M2-Track KITTI 2x1024 pts"}[args.model],
            "value": round(pairs / elapsed, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": ("synthetic worst-case pairs: every ball full of distinct neighbours (open3dsot_amd/synth.py "
                     "make_dense_batch), random-init weights") if args.dense else
                    "synthetic KITTI-Car-like pairs (open3dsot_amd/synth.py, seed 1234+index), random-init weights",
            "config": {"workload": ("M2_track_kitti.yaml, 2x1024 pts, batch %d per GPU, fwd+bwd+Adam, fp32" % args.batch)
                       if args.model == "M2TRACK" else
                       "%s KITTI-Car, template 512 / search %d pts, batch %d per GPU, fwd+bwd+Adam, fp32%s" % (
                           "%s_Car.yaml" % args.model if args.search_size == 1024 else "BAT_CAR_NUSCENES.yaml shapes,",
                           args.search_size, args.batch,
                           ", worst-case DENSE clouds (every ball full of distinct neighbours)" if args.dense else
                           " (the per-GPU batch of cfgs/BAT_CAR_NUSCENES.yaml:56)" if args.batch == 100 else ""),
                       "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                       "fused_kernels": bool(sa_modules.fused_enabled()),
                       "hip_graph": trainer.graph is not None},      # (false = the capture failed and the step ran eagerly: see graph_error)
            "roofline": roofline,
        }
        if full and not args.no_cpu_baseline and world == 1 and args.search_size == 1024:
            line["cpu_baseline"] = cpu_baseline(args.model, sd_cpu, args.batch, args.cpu_budget)
        if trainer.graph is None and not args.no_graph:
            line["config"]["graph_error"] = (trainer.graph_error or "no capture attempted")[:300]
        line["config"]["rccl_world_size"] = dist.get_world_size() if ((world > 1 or exchange) and dist.is_initialized()) else 1
        if exchange:
            line["config"]["rccl_backend"] = dist.get_backend() if dist.is_initialized() else None
        if multi is not None:
            line["config"].update(multi)
        return line
    return {}


if __name__ == "__main__":
    main()
