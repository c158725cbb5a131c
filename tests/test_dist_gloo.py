"""world_size-2 gloo test of the data-parallel step (CPU): two ranks on disjoint shards must
end with identical parameters, equal to a single process that averages the two shard
gradients itself.  Uses a small stand-in model with the tracker's `training_loss` protocol
(the real tracker needs the GPU operator set); a second test drives the REAL BAT tracker through the
same step with the oracle's operator shim standing in for the HIP library (test only)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Linear(6, 16)
        self.bn = torch.nn.BatchNorm1d(16)
        self.b = torch.nn.Linear(16, 1)

    def training_loss(self, batch):
        y = self.b(torch.relu(self.bn(self.a(batch["x"])))).squeeze(1)
        loss = ((y - batch["y"]) ** 2).mean()
        return loss, {"l": loss}

    def configure_optimizers(self):
        return {"optimizer": torch.optim.Adam(self.parameters(), lr=1e-2, betas=(0.5, 0.999), eps=1e-6)}


def _data(first, n):
    g = torch.Generator().manual_seed(1000 + first)
    x = torch.randn(n, 6, generator=g)
    return {"x": x, "y": x.sum(1) * 0.3}


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from open3dsot_amd import dist as D
    r, _, w = D.init_distributed("gloo")
    torch.manual_seed(100 + rank)         # different init per rank: broadcast must fix it
    model = Toy()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(rank * 0.1)
    trainer = D.DataParallelStep(model)
    for step in range(3):
        first, n = D.shard_indices(step, r, w, 8)
        trainer.step(_data(first, n))
    torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(out, "rank%d.pt" % rank))
    # bench.py's multi-GPU self-check (collective): replicas identical -> divergence exactly 0, one rate per rank
    chk = D.replica_self_check(model, trainer, 0.5 + rank, 3 * 8)
    with torch.no_grad():
        next(model.parameters()).add_(0.25 * rank)          # ... and a broken exchange would show
    chk2 = D.replica_self_check(model, trainer, 1.0, 24)
    torch.save({"ok": chk, "broken": chk2}, os.path.join(out, "check%d.pt" % rank))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_matches_manual_average(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0 = torch.load(tmp_path / "rank0.pt")
    s1 = torch.load(tmp_path / "rank1.pt")
    for r in (0, 1):
        c = torch.load(tmp_path / ("check%d.pt" % r))
        assert c["ok"]["max_parameter_divergence"] == 0.0 and c["ok"]["per_rank_pairs_per_s"] == [48.0, 16.0], c["ok"]
        assert abs(c["broken"]["max_parameter_divergence"] - 0.25) < 1e-6 and "all_reduce" in c["ok"]["gradient_exchange"]
    for k in s0:
        if "running" in k or "num_batches" in k:
            continue  # BatchNorm statistics are per-rank by design (no sync-BN in the reference)
        assert torch.equal(s0[k], s1[k]), k
    # single-process emulation: average of the two shard gradients, per-shard BatchNorm stats
    from open3dsot_amd import dist as D
    ref = Toy()
    opt = ref.configure_optimizers()["optimizer"]
    import copy
    for step in range(3):
        grads = []
        for r in range(2):
            m = copy.deepcopy(ref)
            first, n = D.shard_indices(step, r, 2, 8)
            loss, _ = m.training_loss(_data(first, n))
            loss.backward()
            grads.append([p.grad.clone() for p in m.parameters()])
        for p, g0, g1 in zip(ref.parameters(), *grads):
            p.grad = (g0 + g1) / 2
        opt.step()
    for (k, v), p in zip([(k, v) for k, v in s0.items() if "running" not in k and "num_batches" not in k],
                         ref.parameters()):
        assert torch.allclose(v, p.detach(), rtol=1e-5, atol=1e-6), k


def _world1_worker(rank, port, out):
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from open3dsot_amd import dist as D
    D.init_distributed("gloo", force=True)
    model, twin = Toy(), Toy()
    forced = D.DataParallelStep(model, exchange=True)          # the multi-rank path on one rank
    plain = D.DataParallelStep(twin, world=1)
    assert forced.exchange and not plain.exchange and forced.world == 1
    for step in range(3):
        forced.step(_data(step * 8, 8))
        plain.step(_data(step * 8, 8))
    res = {"views": all(p.grad is v for p, v in zip(forced.grads.params, forced.grads.views)),
           "plain_views": any(p.grad is v for p, v in zip(plain.grads.params, plain.grads.views)),
           "equal": all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), twin.state_dict().values())),
           "check": D.replica_self_check(model, forced, 1.0, 24)}
    torch.save(res, os.path.join(out, "w1.pt"))
    torch.distributed.destroy_process_group()


def test_forced_exchange_at_world_size_one_equals_the_plain_step(tmp_path):
    """DataParallelStep(exchange=True) on a one-rank process group runs the multi-rank code (pack, all-reduce, views of the
    exchange buffer as p.grad) and must leave exactly the plain single-process parameters; the GPU twin
    (tests/test_model_gpu.py::test_world_size_one_rccl_drives_the_multi_gpu_step) does it over RCCL with the HIP graph"""
    mp.spawn(_world1_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    r = torch.load(tmp_path / "w1.pt")
    assert r["views"] and not r["plain_views"] and r["equal"], r
    assert r["check"]["max_parameter_divergence"] == 0.0 and r["check"]["per_rank_pairs_per_s"] == [24.0]
    import pytest
    from open3dsot_amd import dist as D
    with pytest.raises(RuntimeError):
        D.DataParallelStep(Toy(), exchange=True)               # no process group in this process


def test_shards_are_disjoint():
    from open3dsot_amd import dist as D
    seen = set()
    for step in range(3):
        for r in range(4):
            first, n = D.shard_indices(step, r, 4, 48)
            ids = set(range(first, first + n))
            assert not (ids & seen)
            seen |= ids
    assert len(seen) == 3 * 4 * 48


# ---- the real tracker (BAT, small clouds) through DataParallelStep, world_size 2 ----------------------------
def _patch_ext_with_oracle():
    """what tests/conftest.py::cpu_ext does, for a spawned worker: the oracle's CPU operators stand in for the HIP
    library (TEST ONLY) and the fused kernels are off"""
    from oracle import ext_shim, ops as oops
    import open3dsot_amd.ext as ext
    from open3dsot_amd import sa_modules
    for name in ("furthest_point_sampling", "gather_points", "gather_points_grad", "gather_rows", "three_nn", "three_interpolate",
                 "three_interpolate_grad", "ball_query", "group_points", "group_points_grad"):
        setattr(ext, name, getattr(ext_shim, name))
    ext.knn = lambda q, r, k: torch.from_numpy(oops.knn(q.detach().numpy(), r.detach().numpy(), k))
    sa_modules.set_fused(False)


def _bat_batch(first, n):
    from open3dsot_amd import synth
    return synth.to_torch(synth.make_batch(first, n, 128, 256))


def _bat_model(seed):
    from open3dsot_amd import trackers
    torch.manual_seed(seed)
    return trackers.BAT(trackers.make_config(trackers.BAT_CAR, num_proposal=32, optimizer="sgd", lr=0.01)).train()


def _bat_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    _patch_ext_with_oracle()
    from open3dsot_amd import dist as D
    r, _, w = D.init_distributed("gloo")
    model = _bat_model(7 + rank)          # different init per rank: the constructor's broadcast must fix it
    trainer = D.DataParallelStep(model)
    assert trainer.scheduler is not None  # StepLR of configure_optimizers is kept (base_model.py:34-35)
    losses = []
    for step in range(2):
        first, n = D.shard_indices(step, r, w, 2)
        losses.append(float(trainer.step(_bat_batch(first, n))))
    trainer.epoch_end()
    torch.save({"sd": {k: v.clone() for k, v in model.state_dict().items()}, "losses": losses,
                "lr": trainer.optimizer.param_groups[0]["lr"]}, os.path.join(out, "bat%d.pt" % rank))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_real_tracker(tmp_path):
    port = _free_port()
    mp.spawn(_bat_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "bat0.pt"), torch.load(tmp_path / "bat1.pt")
    stat = lambda k: "running" in k or "num_batches" in k
    for k in r0["sd"]:
        if not stat(k):
            assert torch.equal(r0["sd"][k], r1["sd"][k]), k          # replicas stay identical
    assert any(not torch.equal(r0["sd"][k], r1["sd"][k]) for k in r0["sd"] if "running_mean" in k)   # per-rank BN
    assert r0["lr"] == r1["lr"] and r0["lr"] > 0
    # single-process emulation: rank 0's initial weights, mean of the two shard gradients, the same Adam
    _patch_ext_with_oracle()
    try:
        import copy
        from open3dsot_amd import dist as D
        ref = _bat_model(7)
        opt = ref.configure_optimizers()["optimizer"]
        reps = [copy.deepcopy(ref) for _ in range(2)]               # per-rank BatchNorm buffers
        for step in range(2):
            grads = []
            for r in range(2):
                reps[r].load_state_dict({k: (v if stat(k) else ref.state_dict()[k]) for k, v in reps[r].state_dict().items()})
                reps[r].zero_grad(set_to_none=True)
                first, n = D.shard_indices(step, r, 2, 2)
                loss, _ = reps[r].training_loss(_bat_batch(first, n))
                loss.backward()
                if r == 0:
                    assert abs(float(loss) - r0["losses"][step]) < 1e-4 * (1 + abs(float(loss)))
                grads.append([p.grad.clone() if p.grad is not None else torch.zeros_like(p) for p in reps[r].parameters()])
            for p, g0, g1 in zip(ref.parameters(), *grads):
                p.grad = (g0 + g1) / 2
            opt.step()
        for k, p in ref.named_parameters():
            # (SGD configuration of base_model.py:29-31 on purpose: Adam normalises every step to +-lr, which turns
            # the rounding noise of near-zero gradients into sign flips of whole steps)
            assert torch.allclose(r0["sd"][k], p.detach(), rtol=1e-4, atol=1e-5), k
    finally:
        from open3dsot_amd import sa_modules
        sa_modules.set_fused(True)


def _m2_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from open3dsot_amd import dist as D, m2track, synth
    r, _, w = D.init_distributed("gloo")
    torch.manual_seed(3 + rank)           # different init per rank: the constructor's broadcast must fix it
    model = m2track.M2TRACK().train()
    trainer = D.DataParallelStep(model, optimizer=torch.optim.SGD(model.parameters(), lr=0.01))
    losses = []
    for step in range(2):
        first, n = D.shard_indices(step, r, w, 4)
        losses.append(float(trainer.step(synth.to_torch(synth.make_motion_batch(first, n, 128)))))
    torch.save({"sd": {k: v.clone() for k, v in model.state_dict().items()}, "losses": losses},
               os.path.join(out, "m2%d.pt" % rank))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_m2track(tmp_path):
    """the M2-Track step (no pointnet2 operator, so its CPU mirror runs as is) through the same
    data-parallel step on two gloo ranks: identical parameters after two steps from different initialisations, per-rank
    BatchNorm statistics, finite losses that differ between the shards"""
    port = _free_port()
    mp.spawn(_m2_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "m20.pt"), torch.load(tmp_path / "m21.pt")
    stat = lambda k: "running" in k or "num_batches" in k
    assert all(torch.equal(r0["sd"][k], r1["sd"][k]) for k in r0["sd"] if not stat(k))
    assert any(not torch.equal(r0["sd"][k], r1["sd"][k]) for k in r0["sd"] if "running_mean" in k)
    assert all(int(v) == 2 for k, v in r0["sd"].items() if "num_batches" in k)
    assert all(l == l and abs(l) < 1e4 for l in r0["losses"] + r1["losses"]) and r0["losses"] != r1["losses"]


def test_bench_refuses_to_report_fewer_gpus_than_asked():
    """`python bench.py --gpus N` outside torchrun spawns N ranks itself; with fewer than N GPUs visible it must stop
    with an error, never print a line for a smaller job (here: no GPU at all, or one)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    n = 64                       # more than any node has
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "refusing" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout
