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
from torch.autograd.graph import increment_version


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_distributed(backend=None, force=False):
    """Initialise torch.distributed from the torchrun environment; returns (rank, local_rank, world).
    force: create the process group at world size 1 as well (a one-rank RCCL communicator: what
    tests/test_model_gpu.py::test_world_size_one_rccl_* drives the multi-GPU code path with on a one-GPU box)."""
    rank, local_rank, world = env_rank()
    use_cuda = torch.cuda.is_available()
    if use_cuda:
        torch.cuda.set_device(local_rank)
    if (world > 1 or force) and not dist.is_initialized():
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


def _defer_wgrads():
    """the backward's scope: the heads' weight gradients in grouped launches (fused_heads.defer_wgrads) and the weight-gradient
    side branch (fused.wgrad_branch: the set-abstraction levels' wgrad launches on a second stream / graph branch, joined when
    the scope ends, i.e. before anything reads a gradient)"""
    import contextlib
    if not torch.cuda.is_available():
        return contextlib.nullcontext()
    from . import fused, fused_heads
    stack = contextlib.ExitStack()
    stack.enter_context(fused.wgrad_branch())          # exits LAST: joins after the heads' final flush
    stack.enter_context(fused_heads.defer_wgrads())
    return stack


# prefetch variants (round 6; gpurun_out/prefetch_matrix2.txt: in place / by copy and default / high stream priority all within
# 0.03 ms of each other on the BAT step): extras written straight into the next FlatBatch's own fields; default stream priority
_PREFETCH = {"inplace": True, "high_priority": False}


class FlatBatch(dict):
    """A batch whose tensors are views of ONE flat device buffer (fields 256-byte aligned).  A captured step reads its
    inputs from static buffers; handing it a new batch is then one device copy of `flat` instead of one copy node per
    field (nine 5 us nodes per BAT step).  `extra`: {name: (shape, dtype)} of fields computed later on the device (the
    prefetched sampling indices) that travel in the same buffer."""

    def __init__(self, batch, extra=None):
        super().__init__()
        fields = [(k, tuple(v.shape), v.dtype) for k, v in batch.items()]
        fields += [(k, tuple(shape), dtype) for k, (shape, dtype) in (extra or {}).items()]
        dev = next(iter(batch.values())).device
        offs, off = [], 0
        for k, shape, dtype in fields:
            n = 1
            for d in shape:
                n *= d
            offs.append(off)
            off += -(-n * torch.empty((), dtype=dtype).element_size() // 256) * 256
        self.flat = torch.zeros(off, dtype=torch.uint8, device=dev)
        self.layout = tuple(fields)
        self.extra_keys = tuple((extra or {}).keys())
        for (k, shape, dtype), o in zip(fields, offs):
            n = 1
            for d in shape:
                n *= d
            nbytes = n * torch.empty((), dtype=dtype).element_size()
            self[k] = self.flat[o:o + nbytes].view(dtype).view(shape)
        for k, v in batch.items():
            self[k].copy_(v)


class FlatGrads:
    """One contiguous fp32 buffer for the gradient exchange.  Autograd is left to ASSIGN each p.grad
    (p.grad is None before backward: AccumulateGrad then keeps the produced tensor, no `grad += new`
    kernel per parameter -- 76 launches per step here); `gather()` packs them into the flat buffer with
    one foreach copy for the all-reduce, `views` are the per-parameter slices of that buffer."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.views = []
        off = 0
        for p in self.params:
            n = p.numel()
            self.views.append(self.flat[off:off + n].view_as(p))
            off += n

        # device job table of the captured step's pack launch (csrc/heads.hip::prep_weights_kernel: {src, dst, rows, cols, ld,
        # transpose} per parameter): allocated here, outside any capture; filled once the capture has fixed the addresses of
        # the gradient tensors the captured backward writes (`fill_table`)
        self.table = torch.zeros((len(self.params), 6), dtype=torch.int64, device=ref.device) if ref.is_cuda else None
        self._captured_pack = False

    def clear(self):
        for p in self.params:
            p.grad = None

    def gather(self, grads):
        """flat buffer <- the gradients autograd produced (None -> zeros).  Eagerly: one foreach copy (two multi-tensor
        launches, ~70 us for the 76 tensors of BAT).  While a HIP graph is being captured: ONE launch of the library's
        job-table copy kernel -- the captured backward writes the same gradient buffers at every replay, so their addresses
        go into the table once, right after the capture (round 6: the pack was 0.07 of the 0.16 ms the exchange path costs
        per step at world size 1, gpurun_out/r6x)"""
        if (self.table is not None and torch.cuda.is_current_stream_capturing() and
                all(g is not None and g.is_contiguous() and g.dtype == v.dtype and g.numel() == v.numel()
                    for g, v in zip(grads, self.views))):
            from . import capi, fused_heads  # noqa: F401  (fused_heads registers o3d_prep_weights' signature)
            lib = capi.load()
            capi.check(lib.o3d_prep_weights(self.table.data_ptr(), len(self.params),
                                            torch.cuda.current_stream(self.flat.device).cuda_stream), "pack_grads")
            self._captured_pack = True
            return
        src = [g if g is not None else torch.zeros_like(v) for g, v in zip(grads, self.views)]
        torch._foreach_copy_(self.views, src)

    def fill_table(self, grads):
        """after the capture: the job table of the captured pack launch <- the addresses of the gradient tensors the captured
        backward writes (they live in the graph's memory pool and are held by DataParallelStep._static_grads)"""
        if not self._captured_pack:
            return
        rows = [[g.data_ptr(), v.data_ptr(), 1, v.numel(), v.numel(), 0] for g, v in zip(grads, self.views)]
        self.table.copy_(torch.tensor(rows, dtype=torch.int64))

    def bind_views(self):
        for p, v in zip(self.params, self.views):
            p.grad = v


class DataParallelStep:
    """model + optimizer + one-message gradient all-reduce.  `step(batch)` = forward, backward,
    all-reduce (mean over ranks), optimizer update; returns the detached loss (device tensor).

    graph=True captures forward+backward (every kernel of the hot path: FPS, ball queries, the
    MFMA GEMMs, losses, autograd) into ONE HIP graph after `graph_warmup` eager steps and replays
    it afterwards -- the step has no host synchronisation and fixed shapes, so a replay is the
    same work with none of the per-launch host overhead.  The all-reduce and the optimizer stay
    outside the graph (eagerly enqueued while the graph runs)."""

    def __init__(self, model, optimizer=None, world=None, graph=False, graph_warmup=3, prefetch_sampling=True,
                 require_graph=None, exchange=None):
        self.model = model
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        # exchange: run the multi-rank path -- parameter broadcast, gradient pack inside the captured step, the one flat
        # all-reduce, p.grad = views of the exchange buffer.  Default: whenever there is more than one rank; True forces
        # it at world size 1 (needs an initialised process group), so that the very code an 8-GPU run executes can be
        # exercised on one GPU
        self.exchange = (self.world > 1) if exchange is None else bool(exchange)
        if self.exchange and not dist.is_initialized():
            raise RuntimeError("DataParallelStep(exchange=True) needs an initialised torch.distributed process group")
        if self.exchange:  # identical replicas: rank 0's parameters and buffers everywhere
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0)
        self.grads = FlatGrads(model.parameters())
        self.scheduler = None
        if optimizer is None:
            conf = model.configure_optimizers()
            optimizer, self.scheduler = conf["optimizer"], conf.get("lr_scheduler")   # StepLR of base_model.py:34-35
        self.optimizer = optimizer
        self.graph_requested = bool(graph) and torch.cuda.is_available()
        # require_graph: a failed capture raises instead of falling back to the (several times slower) eager step;
        # default: the O3D_REQUIRE_GRAPH=1 environment switch (CI / benchmarking)
        self.require_graph = (os.environ.get("O3D_REQUIRE_GRAPH", "0") == "1") if require_graph is None else bool(require_graph)
        self._views_bound = False
        self.graph_warmup = graph_warmup
        self.graph = None
        self.graph_error = None
        self._static = None
        self._static_loss = None
        self._static_grads = None      # the gradient tensors the captured backward writes (graph pool)
        self._eager_steps = 0
        self._buffers = [b for b in model.buffers()]
        # sampling prefetch (HISTORY.md section 7): the farthest-point-sampling indices of batch t+1 are computed on a second
        # stream while the graph of step t replays -- 766 strictly serial rounds on 96 of 1024 SIMDs otherwise head every
        # step with the rest of the chip idle.  prefetch_sampling=False keeps the sampling inside the captured step (what
        # tests/test_model_gpu.py::test_sampling_prefetch_* compares against).
        self._sampling = getattr(model, "sampling_inputs", None) if prefetch_sampling else None
        self._sampling_takes_out = False
        if self._sampling is not None:
            import inspect
            self._sampling_takes_out = "out" in inspect.signature(self._sampling).parameters
        self._side = None
        self._prefetched = None        # (batch object, {key: tensor}, event)
        # the constant 1 that seeds `loss.backward`: allocated here, outside any capture (open3dsot_amd/fused_loss.py::one)
        first = next(iter(model.parameters()), None)
        self._one = None
        if first is not None and first.is_cuda:
            from . import fused_loss
            self._one = fused_loss.one(first.device)

    def reduce_gradients(self):
        """flat exchange buffer (packed by `_forward_backward`) -> mean over ranks, p.grad = its views.  One collective on
        one contiguous 5.9 MB message; RCCL averages in the collective itself (ReduceOp.AVG: no separate division
        launch), gloo (CPU tests) sums and divides."""
        if self.exchange:
            if dist.get_backend() == "nccl":
                dist.all_reduce(self.grads.flat, op=dist.ReduceOp.AVG)
            else:
                dist.all_reduce(self.grads.flat, op=dist.ReduceOp.SUM)
                self.grads.flat.div_(self.world)
            # p.grad = the views of the exchange buffer.  An eager step cleared them (autograd ASSIGNS fresh tensors, which
            # `_forward_backward` packs into the buffer), so they are bound again; a replayed graph never touches p.grad,
            # so once bound they stay bound: no per-step host loop over the 76 parameters (round-3 review)
            # (p.grad is OWNED by this class once a graph is captured; the check walks every parameter -- a caller that
            # cleared or replaced some of them, e.g. zero_grad on a subset, gets them bound again: ~10 us of host time)
            if self.graph is None or not self._views_bound or \
                    any(p.grad is not v for p, v in zip(self.grads.params, self.grads.views)):
                self.grads.bind_views()
                self._views_bound = self.graph is not None

    def _forward_backward(self, batch):
        if isinstance(batch, FlatBatch) and batch is not self._static and batch.extra_keys:
            batch = {k: v for k, v in batch.items() if k not in batch.extra_keys}   # an eager step samples for itself
        self.grads.clear()
        loss, _ = self.model.training_loss(batch)
        with _defer_wgrads():         # the heads' weight gradients of the whole backward in a few grouped launches
            if self._one is not None and loss.dim() == 0 and loss.device == self._one.device:
                loss.backward(gradient=self._one)      # no ones_like launch; the fused loss skips its scaling launch
            else:
                loss.backward()
        if self.exchange:
            # pack the gradients autograd produced into the exchange buffer: one multi-tensor copy, part of the captured
            # HIP graph when there is one (so a replayed step ends with the message ready to be reduced)
            self.grads.gather([p.grad for p in self.grads.params])
        return loss.detach()

    def _capture(self, batch):
        extra = {}
        if self._sampling is not None:
            # the sampling indices become INPUTS of the captured step (static buffers): the graph holds no FPS launch
            with torch.no_grad():
                extra = self._sampling({k: v for k, v in batch.items() if k not in getattr(batch, "extra_keys", ())})
        src = {k: v for k, v in batch.items() if k not in extra}
        self._static = FlatBatch(src, {k: (v.shape, v.dtype) for k, v in extra.items()})
        for k, v in extra.items():
            self._static[k].copy_(v)
        # the allocator warm-up pass below is NOT a training step: the BatchNorm running statistics and
        # num_batches_tracked it touches are put back, so a graph run sees exactly one update per step()
        buffers = list(self.model.buffers())
        saved = [b.clone() for b in buffers]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # eager passes on the capture stream's allocator; two, so that whatever
            self._forward_backward(self._static)      # the first one registered (weight-preparation jobs, cached
            self._forward_backward(self._static)      # constants) is in its steady state when the capture begins
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for b, v in zip(buffers, saved):
            b.copy_(v)
        g = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (the RCCL watchdog of torch.distributed polls events) may
        # keep calling the runtime while this thread captures
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._static_loss = self._forward_backward(self._static)
        self._static_grads = [p.grad for p in self.grads.params]
        if self.exchange:
            self.grads.fill_table(self._static_grads)
        self.graph = g

    def _sampling_for(self, batch):
        """sampling inputs of `batch`: the prefetched ones when `batch` is the very object announced as `next_batch`
        of the previous step (the main stream then waits for the side stream's event), else computed here and now"""
        main = torch.cuda.current_stream()
        if self._prefetched is not None and self._prefetched[0] is batch:
            _, extra, ev = self._prefetched
            main.wait_event(ev)
            for t in extra.values():
                t.record_stream(main)          # allocated on the side stream, read by the copy below on this one
        else:
            with torch.no_grad():
                extra = self._sampling({k: v for k, v in batch.items() if k not in getattr(batch, "extra_keys", ())})
        self._prefetched = None
        return extra

    def _prefetch(self, next_batch):
        if self._side is None:
            self._side = torch.cuda.Stream(priority=-1 if _PREFETCH["high_priority"] else 0)
        own = getattr(next_batch, "extra_keys", ()) if next_batch is not self._static else ()
        # behind everything the main stream has been given so far (the input copy of THIS step; not its graph, which is
        # replayed after this call): nothing on the main stream still reads the buffers written here
        self._side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._side), torch.no_grad():
            src = {k: v for k, v in next_batch.items() if k not in own}
            if own and self._sampling_takes_out and _PREFETCH["inplace"]:      # straight into the FlatBatch's own fields
                extra = self._sampling(src, out={k: next_batch[k] for k in own})
            else:
                extra = self._sampling(src)
            if own:                    # a FlatBatch with room for them: they travel with its one flat copy
                for k in own:
                    if extra[k] is not next_batch[k]:
                        next_batch[k].copy_(extra[k])
                extra = {k: next_batch[k] for k in own}
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._prefetched = (next_batch, extra, ev)

    def make_batch(self, batch):
        """batch (dict of device tensors) -> a FlatBatch laid out like the captured step's static inputs (one device copy
        per step instead of one per field); before the capture, or for another layout, the dict itself"""
        if self._static is None:
            return batch
        fb = FlatBatch({k: v for k, v in batch.items() if k not in self._static.extra_keys},
                       {k: (self._static[k].shape, self._static[k].dtype) for k in self._static.extra_keys})
        return fb if fb.layout == self._static.layout else batch

    def step(self, batch, next_batch=None):
        """one training step on `batch`.  next_batch: the batch the NEXT call will be given (the very same dict object),
        when the caller knows it -- its input-only preprocessing (farthest-point sampling) then runs beside this step"""
        if self.graph is not None:
            src = batch
            flat = isinstance(batch, FlatBatch) and batch.layout == self._static.layout
            if self._sampling is not None and self._static.extra_keys:
                extra = self._sampling_for(batch)
                if set(extra) != set(self._static.extra_keys):
                    # (e.g. trackers.set_geometry_prefetch flipped after the capture: a missing key would leave the PREVIOUS
                    # batch's value in the captured step's input buffers)
                    raise RuntimeError("DataParallelStep: the model's sampling_inputs() returns other keys than the captured "
                                       "step was built with (%s vs %s); build a new trainer" % (
                                           sorted(extra)[:4], sorted(self._static.extra_keys)[:4]))
                if not (flat and all(extra[k] is batch[k] for k in extra)):
                    flat = False
                    src = dict(batch)
                    src.update(extra)
            if flat:
                if batch is not self._static:
                    self._static.flat.copy_(batch.flat, non_blocking=True)       # every input field in one device copy
            else:
                keys = [k for k in src if src[k] is not self._static[k]]
                if keys:
                    torch._foreach_copy_([self._static[k] for k in keys], [src[k] for k in keys], non_blocking=True)
            if next_batch is not None and self._sampling is not None and self._static.extra_keys:
                self._prefetch(next_batch)        # enqueued before the replay: it starts with the step
            self.graph.replay()
            # the replayed finalize kernels rewrote the BatchNorm running statistics through raw pointers: bump their
            # version counters (host only) so version-keyed caches -- eval-mode constants -- see a training step
            increment_version(self._buffers)
            if not self.exchange and (not self._views_bound or
                                      any(p.grad is not g for p, g in zip(self.grads.params, self._static_grads))):
                # (with an exchange the replay packed them into the exchange buffer, reduce_gradients binds its views)
                for p, g in zip(self.grads.params, self._static_grads):   # the replay rewrites these very buffers: bound once
                    p.grad = g
                self._views_bound = True
            loss = self._static_loss.clone()      # the graph rewrites its own buffer at the next replay
        else:
            if self.graph_requested and self._eager_steps >= self.graph_warmup:
                try:
                    self._capture(batch)
                except Exception as e:  # capture unsupported for some op: stay eager, say so
                    self.graph_error = "%s: %s" % (type(e).__name__, e)
                    self.graph_requested = False
                    self.graph = None
                    torch.cuda.synchronize()
                    if self.require_graph:
                        raise RuntimeError("HIP-graph capture of the training step failed: " + self.graph_error) from e
                    # loudly: an eager step is several times slower on the launch-bound models, and a failed capture went
                    # unnoticed for two rounds on M2-Track (HISTORY.md 8b)
                    import warnings
                    warnings.warn("open3dsot_amd: HIP-graph capture of the training step failed, running eagerly (%s)"
                                  % self.graph_error.splitlines()[0][:300], RuntimeWarning, stacklevel=2)
            if self.graph is not None:
                return self.step(batch, next_batch)
            loss = self._forward_backward(batch)
            self._eager_steps += 1
        self.reduce_gradients()
        self.optimizer.step()
        return loss

    def epoch_end(self):
        """step the learning-rate schedule (the reference's Lightning loop steps StepLR once per epoch)"""
        if self.scheduler is not None:
            self.scheduler.step()

    def sync_buffers(self):
        """BatchNorm statistics are per rank during training (no sync-BN in the reference; its DDP wrapper
        re-broadcasts rank 0's buffers every forward): call before checkpointing so every rank holds rank 0's"""
        if self.exchange:
            for b in self.model.buffers():
                dist.broadcast(b.data, src=0)


def replica_self_check(model, trainer, elapsed_s, pairs_per_rank):
    """Collective (every rank must call it): what a multi-GPU bench run can say about itself -- each rank's own rate and how
    far the replicas' parameters are apart after the timed steps (identical gradients after the all-reduce and the same Adam
    update must leave them bitwise equal: anything else is a broken exchange).  -> dict for the bench line's `config`."""
    world = dist.get_world_size()
    first = next(model.parameters())
    # every fallible LOCAL step (allocations: the flat parameter copy is 3 x 5.9 MB) comes before the first collective, and
    # the ranks agree on an ok flag first: a rank that failed alone would otherwise skip collectives the others block in
    err = None
    try:
        mine = torch.tensor([elapsed_s], device=first.device, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        flat = torch.cat([p.detach().reshape(-1).float() for p in model.parameters()])
        hi, lo = flat.clone(), flat.clone()
    except Exception as e:
        err = "%s: %s" % (type(e).__name__, e)
    ok = torch.tensor([0 if err else 1], device=first.device, dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        return {"self_check_error": err or "another rank could not allocate the self-check buffers"}
    dist.all_gather(every, mine)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    return {"per_rank_pairs_per_s": [round(pairs_per_rank / max(float(t.item()), 1e-12), 1) for t in every],
            "max_parameter_divergence": float((hi - lo).abs().max().item()),
            "gradient_exchange": "one flat %.2f MB all_reduce per step (%s), outside the captured graph" % (
                trainer.grads.flat.numel() * trainer.grads.flat.element_size() / 1e6, dist.get_backend())}
