"""Adam for the trackers on flat buffers: one kernel launch per step.

`MatchingBaseModel.configure_optimizers` (models/base_model.py:28-36) builds `torch.optim.Adam(betas=(0.5, 0.999),
eps=1e-6)`.  On this model -- 76 small parameter tensors, 1.48 M values -- torch's fused multi-tensor Adam takes three
launches and 0.13 ms per step.  `FlatAdam` is the same update rule with every parameter and both moments as views of
flat buffers and the gradients read where autograd left them (device job table, csrc/heads.hip::adam_step_kernel):
one ~10 us launch.  `state_dict()` has torch.optim.Adam's layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), so
checkpoints are interchangeable with the reference's optimizer.
"""
import ctypes

import torch
from torch.autograd.graph import increment_version

from . import capi

_vp, _i, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
capi.register("o3d_adam_step", [_vp, _i, _vp, _vp, _vp, _d, _d, _d, _d, _d, _d, _d, _vp])


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        ps = [p for g in self.param_groups for p in g["params"]]
        if not ps or not all(p.is_cuda and p.dtype == torch.float32 for p in ps):
            raise ValueError("FlatAdam: fp32 CUDA parameters only (use torch.optim.Adam elsewhere)")
        dev = ps[0].device
        total = sum(p.numel() for p in ps)
        self._flat = torch.empty(total, device=dev, dtype=torch.float32)
        self._m = torch.zeros(total, device=dev, dtype=torch.float32)
        self._v = torch.zeros(total, device=dev, dtype=torch.float32)
        self._step = torch.zeros((), dtype=torch.float32)          # shared by every parameter (torch keeps one each)
        self._offsets = {}
        off = 0
        with torch.no_grad():
            for p in ps:
                n = p.numel()
                view = self._flat[off:off + n].view_as(p)
                view.copy_(p)
                p.data = view                       # the parameter now lives in the flat buffer
                self.state[p] = {"step": self._step, "exp_avg": self._m[off:off + n].view_as(p),
                                 "exp_avg_sq": self._v[off:off + n].view_as(p)}
                self._offsets[p] = (off, n)
                off += n
        self._key, self._table = None, None

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = capi.load()
        base = self._flat.data_ptr()
        for p, (off, _) in self._offsets.items():
            # every parameter must still be a view of the flat buffer: a second optimizer built on the same model,
            # module.to(dtype) or any `p.data = ...` re-homes it and this update would silently go nowhere
            if p.data_ptr() != base + 4 * off:
                raise RuntimeError("FlatAdam: a parameter no longer lives in this optimizer's flat buffer (re-homed by "
                                   "another FlatAdam, module.to(...) or a p.data assignment); rebuild the optimizer")
        self._step += 1
        t = float(self._step)
        for gi, group in enumerate(self.param_groups):
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in live]
            key = (gi,) + tuple(g.data_ptr() for g in grads)
            if self._key is None or self._key.get(gi) != key:
                # (eager autograd hands out new gradient tensors every step: one small upload; a replayed HIP graph
                # writes the same buffers every step: the table is built once)
                rows = [[g.data_ptr(), self._offsets[p][0], self._offsets[p][1]] for p, g in zip(live, grads)]
                self._key = dict(self._key or {})
                self._key[gi] = key
                self._table = dict(self._table or {})
                self._table[gi] = torch.tensor(rows, dtype=torch.int64, device=self._flat.device)
            b1, b2 = group["betas"]
            tab = self._table[gi]
            dev = self._flat.device
            with torch.cuda.device(dev):
                capi.check(lib.o3d_adam_step(tab.data_ptr(), tab.shape[0], self._flat.data_ptr(), self._m.data_ptr(),
                                             self._v.data_ptr(), float(group["lr"]), float(b1), float(b2),
                                             float(group["eps"]), float(group["weight_decay"]), 1.0 - b1 ** t,
                                             1.0 - b2 ** t, torch.cuda.current_stream(dev).cuda_stream), "adam_step")
            self._keep = grads                      # the launch reads them: alive until the next step
            # the kernel wrote the parameters through the flat buffer: bump their version counters like an in-place
            # torch op would, so the forward/backward guards of the fused operators and the eval-constant cache see it
            increment_version(live)
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        missing = [p for p in self._offsets if not self.state.get(p)]
        if missing and len(missing) < len(self._offsets):
            # torch.optim.Adam omits parameters that never received a gradient and would start them at step 0 on their first
            # update; FlatAdam keeps ONE step counter for all parameters, so their bias correction continues at the
            # checkpoint's step -- the updates of exactly those parameters differ from the reference optimizer's after resume
            import warnings
            warnings.warn("FlatAdam.load_state_dict: the checkpoint holds state for %d of %d parameters; the others get zero "
                          "moments but share the checkpoint's step counter (torch.optim.Adam would restart them at step 0)"
                          % (len(self._offsets) - len(missing), len(self._offsets)), RuntimeWarning, stacklevel=2)
        with torch.no_grad():                       # back into the flat buffers (the base class swapped in copies)
            step = None
            for p, (off, n) in self._offsets.items():
                st = self.state.get(p)
                if not st:          # e.g. a torch.optim.Adam checkpoint taken before its first step: fresh moments
                    self._m[off:off + n].zero_()
                    self._v[off:off + n].zero_()
                    self.state[p] = {"step": self._step, "exp_avg": self._m[off:off + n].view_as(p),
                                     "exp_avg_sq": self._v[off:off + n].view_as(p)}
                    continue
                self._m[off:off + n].view_as(p).copy_(st["exp_avg"])
                self._v[off:off + n].view_as(p).copy_(st["exp_avg_sq"])
                step = float(st["step"]) if step is None else step
                st["exp_avg"], st["exp_avg_sq"] = self._m[off:off + n].view_as(p), self._v[off:off + n].view_as(p)
            self._step.fill_(step if step is not None else 0.0)
            for st in self.state.values():
                st["step"] = self._step
        self._key = self._table = None


def make_adam(params, lr, weight_decay, betas=(0.5, 0.999), eps=1e-6):
    """the reference's Adam (models/base_model.py:32-33): FlatAdam when every parameter is an fp32 GPU tensor, else
    torch.optim.Adam (the CPU mirror used by the tests)"""
    params = list(params)
    if params and all(q.is_cuda and q.dtype == torch.float32 for q in params) and _ON["on"]:
        return FlatAdam(params, lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)
    return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay, betas=betas, eps=eps,
                            fused=True if params and all(q.is_cuda for q in params) else None)


_ON = {"on": True}      # set_flat_adam(False): torch.optim.Adam, for the tests that compare the two update rules


def set_flat_adam(enabled):
    _ON["on"] = bool(enabled)
