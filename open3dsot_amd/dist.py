"""One-process-per-GPU data-parallel training step (RCCL over xGMI on MI355X).

Restates the only parallelism the reference uses -- PyTorch-Lightning `accelerator='ddp'`
(main.py:82: one process per GPU, per-GPU batch, gradients summed with NCCL, BatchNorm
statistics NOT synchronised) -- without Lightning and without torch's bucketed DDP wrapper:
every template/search pair is independent, so the pairs are sharded over ranks and the one
exchange per step is a single all-reduce of the 1.48 M fp32 gradients (5.9 MB).  All
parameter gradients live in ONE flat device buffer (each `p.grad` is a view into it), so the
exchange is one collective on one contiguous message: over xGMI's 7 point-to-point links a
5.9 MB all-reduce is a few tens of microseconds against a multi-millisecond step, so it is
issued once after backward instead of being chopped into overlap buckets.
Backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU (tests, world_size 2).
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world)."""
    rank, local_rank, world = env_rank()
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if use_cuda:
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend or ("nccl" if use_cuda else "gloo"), rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_indices(step, rank, world, per_rank_batch):
    """Global sample indices of `rank` at `step` (disjoint across ranks, weak scaling)."""
    first = (step * world + rank) * per_rank_batch
    return first, per_rank_batch


class FlatGrads:
    """All parameter gradients as views of one contiguous fp32 buffer."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero(self):
        self.flat.zero_()

    def rebind(self):
        """re-attach views if something replaced p.grad (e.g. zero_grad(set_to_none=True))"""
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                if p.grad is not None:
                    view.copy_(p.grad)
                p.grad = view
            off += n


class DataParallelStep:
    """model + optimizer + one-message gradient all-reduce.  `step(batch)` = forward, backward,
    all-reduce (mean over ranks), optimizer update; returns the detached loss (device tensor).

    graph=True captures forward+backward (every kernel of the hot path: FPS, ball queries, the
    MFMA GEMMs, losses, autograd) into ONE HIP graph after `graph_warmup` eager steps and replays
    it afterwards -- the step has no host synchronisation and fixed shapes, so a replay is the
    same work with none of the per-launch host overhead.  The all-reduce and the optimizer stay
    outside the graph (eagerly enqueued while the graph runs)."""

    def __init__(self, model, optimizer=None, world=None, graph=False, graph_warmup=3):
        self.model = model
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        if self.world > 1:  # identical replicas: rank 0's parameters and buffers everywhere
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0)
        self.grads = FlatGrads(model.parameters())
        self.optimizer = optimizer if optimizer is not None else model.configure_optimizers()["optimizer"]
        self.graph_requested = bool(graph) and torch.cuda.is_available()
        self.graph_warmup = graph_warmup
        self.graph = None
        self.graph_error = None
        self._static = None
        self._static_loss = None
        self._eager_steps = 0

    def reduce_gradients(self):
        if self.world > 1:
            dist.all_reduce(self.grads.flat, op=dist.ReduceOp.SUM)
            self.grads.flat.div_(self.world)

    def _forward_backward(self, batch):
        self.grads.zero()
        loss, _ = self.model.training_loss(batch)
        loss.backward()
        return loss.detach()

    def _capture(self, batch):
        self._static = {k: v.clone() for k, v in batch.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # one more eager pass on the capture stream's allocator
            self._forward_backward(self._static)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._static_loss = self._forward_backward(self._static)
        self.graph = g

    def step(self, batch):
        if self.graph is not None:
            for k, v in batch.items():
                self._static[k].copy_(v, non_blocking=True)
            self.graph.replay()
            loss = self._static_loss
        else:
            if self.graph_requested and self._eager_steps >= self.graph_warmup:
                try:
                    self._capture(batch)
                except Exception as e:  # capture unsupported for some op: stay eager, say so
                    self.graph_error = "%s: %s" % (type(e).__name__, e)
                    self.graph_requested = False
                    self.graph = None
                    torch.cuda.synchronize()
            if self.graph is not None:
                return self.step(batch)
            loss = self._forward_backward(batch)
            self._eager_steps += 1
        self.grads.rebind()
        self.reduce_gradients()
        self.optimizer.step()
        return loss
