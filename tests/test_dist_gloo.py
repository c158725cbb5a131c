"""world_size-2 gloo test of the data-parallel step (CPU): two ranks on disjoint shards must
end with identical parameters, equal to a single process that averages the two shard
gradients itself.  Uses a small stand-in model with the tracker's `training_loss` protocol
(the real tracker needs the GPU operator set)."""
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
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_matches_manual_average(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    s0 = torch.load(tmp_path / "rank0.pt")
    s1 = torch.load(tmp_path / "rank1.pt")
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
