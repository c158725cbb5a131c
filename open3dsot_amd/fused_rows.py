"""`Linear -> BatchNorm1d -> ReLU` rows on a handful of samples as ONE launch per layer each way (csrc/rowmlp.hip).

What it replaces (same numbers within fp32 rounding, tests/test_rows_gpu.py):
  the four heads of M2-Track          models/m2track.py:43-71   (Linear(256,128) BN ReLU Linear(128,128) BN ReLU Linear(128,out))
  the hidden rows of MiniPointNet     models/backbone/pointnet.py:118-126   (Linear BN ReLU x 2 behind the global max)
The reference runs them as nn.Sequential on (B, C) activations with B = the per-GPU batch (48 frame pairs in the bench):
addmm + batch_norm statistics / transform / running update + threshold and their mirrors, ~14 launches per row, 144 of
M2-Track's 511 launches per step.  A BatchNorm1d over the rows normalises every feature on its own, so a workgroup that
owns 16 output features needs nobody else: the layer is one launch, the backward of a stack is one launch per layer plus
one for the input gradient (the data gradient of layer l is computed inside layer l-1's launch).
Parameters stay in the caller's modules (state_dict keys unchanged); rows <= 64 (csrc/rowmlp.hip RM_RMAX).
"""
import ctypes

import torch
from torch import nn

from . import capi
from .fused import _call, _ptr, _stream, count_batches

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
capi.register("o3d_row_mlp_fwd", [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp])
capi.register("o3d_row_mlp_bwd", [_vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i,
                                  _vp, _vp, _vp, _vp, _vp, _vp])

RMAX = 64
_ON = {"on": True}


def set_fused_rows(enabled):
    """row stacks on the library's kernel (default) or as the nn.Sequential they are (the specification they are tested against)"""
    _ON["on"] = bool(enabled)


def parse(seq):
    """nn.Sequential of Linear [BatchNorm1d] [ReLU] groups -> [(linear, bn | None, relu)] or None when it is something else"""
    layers, mods, i = [], list(seq), 0
    while i < len(mods):
        if not isinstance(mods[i], nn.Linear):
            return None
        lin, bn, relu = mods[i], None, False
        i += 1
        if i < len(mods) and isinstance(mods[i], nn.BatchNorm1d):
            bn = mods[i]
            i += 1
            if not (bn.affine and bn.track_running_stats and bn.momentum is not None):
                return None
        if i < len(mods) and isinstance(mods[i], nn.ReLU):
            relu = True
            i += 1
        if relu and bn is None:
            return None
        layers.append((lin, bn, relu))
    return layers or None


FMAX = 1024        # csrc/rowmlp.hip::RM_KMAX: a workgroup's weight slice lives in LDS; wider layers return EINVAL there


def supported(layers, x):
    """the kernel's own limits, so that anything outside them takes the nn.Sequential path instead of a RuntimeError from
    the library: <= RMAX rows, every Linear <= FMAX features wide (in and out: the backward stages the layer above), and
    no training-mode BatchNorm over a single row (torch raises there; the kernel would normalise with var = 0)"""
    if not (_ON["on"] and layers is not None and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and
            0 < x.shape[0] <= RMAX and x.shape[1] == layers[0][0].in_features):
        return False
    if any(lin.in_features > FMAX or lin.out_features > FMAX for lin, _, _ in layers):
        return False
    if x.shape[0] == 1 and any(bn is not None and bn.training for _, bn, _ in layers):
        return False
    return True


class _Cfg:
    __slots__ = ("bns", "relus", "training")


class RowStack(torch.autograd.Function):
    """apply(cfg, x (R, Cin), [W, b | None, gamma | None, beta | None] per layer) -> (R, Cout_last)"""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, cfg, x, *params):
        lib = capi.load()
        L = len(params) // 4
        dev, f32 = x.device, torch.float32
        st = _stream()
        need_bwd = any(ctx.needs_input_grad)
        h = x.detach()
        if h.stride(1) != 1:
            h = h.contiguous()
        R = h.shape[0]
        saved = []
        for l in range(L):
            W, b, gamma, beta = params[4 * l:4 * l + 4]
            bn = cfg.bns[l]
            Cout, Cin = W.shape
            Y = torch.empty((R, Cout), device=dev, dtype=f32)
            Z = torch.empty((R, Cout), device=dev, dtype=f32) if (bn is not None and need_bwd) else None
            stat = torch.empty((2, Cout), device=dev, dtype=f32) if (bn is not None and need_bwd) else None
            _call("row_mlp_fwd", 0.0, lib.o3d_row_mlp_fwd, h.data_ptr(), h.stride(0), W.data_ptr(), _ptr(b),
                  _ptr(gamma), _ptr(beta), _ptr(bn.running_mean) if bn is not None else None,
                  _ptr(bn.running_var) if bn is not None else None, float(bn.momentum) if bn is not None else 0.0,
                  float(bn.eps) if bn is not None else 0.0, int(cfg.training), int(cfg.relus[l]), R, Cin, Cout, _ptr(Z),
                  Y.data_ptr(), _ptr(stat[0]) if stat is not None else None, _ptr(stat[1]) if stat is not None else None, st)
            saved.append((h, Z, stat))
            h = Y
        if cfg.training:
            bns = [bn for bn in cfg.bns if bn is not None]
            if bns:
                count_batches(bns, 1)
        if need_bwd:
            ctx.cfg = cfg
            ctx.saved = saved
            ctx.params = params
            ctx.versions = [(p, p._version) for p in params if p is not None]
        return h

    @staticmethod
    @capi.on_tensor_device
    def backward(ctx, dOut):
        lib = capi.load()
        cfg, params = ctx.cfg, ctx.params
        for p, v in ctx.versions:
            if p._version != v:
                raise RuntimeError("a parameter of a fused row stack was modified in place between forward and backward")
        L = len(params) // 4
        dev, f32 = dOut.device, torch.float32
        st = _stream()
        g = dOut if dOut.stride(1) == 1 else dOut.contiguous()
        R = g.shape[0]
        grads = [None] * (4 * L)
        dZup, Wup = None, None
        for l in range(L - 1, -1, -1):
            W, b, gamma, beta = params[4 * l:4 * l + 4]
            X, Z, stat = ctx.saved[l]
            C, Cin = W.shape
            dZ = torch.empty((R, C), device=dev, dtype=f32)
            dW = torch.empty((C, Cin), device=dev, dtype=f32)
            db = torch.empty((C,), device=dev, dtype=f32) if b is not None else None
            dgb = torch.empty((2, C), device=dev, dtype=f32) if gamma is not None else None
            _call("row_mlp_bwd", 0.0, lib.o3d_row_mlp_bwd, g.data_ptr() if dZup is None else None, g.stride(0) if dZup is None else 0,
                  _ptr(dZup), _ptr(Wup), Wup.shape[0] if Wup is not None else 0, 0, None, 0, _ptr(Z), _ptr(gamma), _ptr(beta),
                  _ptr(stat[0]) if stat is not None else None, _ptr(stat[1]) if stat is not None else None, int(cfg.training),
                  int(cfg.relus[l]), X.data_ptr(), X.stride(0), R, Cin, C, dZ.data_ptr(), dW.data_ptr(), _ptr(db),
                  _ptr(dgb[0]) if dgb is not None else None, _ptr(dgb[1]) if dgb is not None else None, st)
            grads[4 * l] = dW
            grads[4 * l + 1] = db
            if dgb is not None:
                grads[4 * l + 2], grads[4 * l + 3] = dgb[0], dgb[1]
            dZup, Wup = dZ, W
        dx = None
        if ctx.needs_input_grad[1]:
            Cin0 = params[0].shape[1]
            dx = torch.empty((R, Cin0), device=dev, dtype=f32)
            _call("row_mlp_bwd", 0.0, lib.o3d_row_mlp_bwd, None, 0, dZup.data_ptr(), Wup.data_ptr(), Wup.shape[0], 1, dx.data_ptr(),
                  Cin0, None, None, None, None, None, 0, 0, None, 0, R, 0, Cin0, None, None, None, None, None, st)
        return (None, dx, *grads)


def run(layers, x):
    """[(linear, bn | None, relu)] applied to x (R, Cin); caller has checked `supported`"""
    cfg = _Cfg()
    cfg.bns = [bn for _, bn, _ in layers]
    cfg.relus = [bool(r) for _, _, r in layers]
    cfg.training = bool(any(bn is not None and bn.training for bn in cfg.bns))
    params = []
    for lin, bn, _ in layers:
        params += [lin.weight, lin.bias, bn.weight if bn is not None else None, bn.bias if bn is not None else None]
    return RowStack.apply(cfg, x, *params)


def seq_rows(seq, x):
    """nn.Sequential of Linear / BatchNorm1d / ReLU modules on x (R, Cin): the fused stack when it applies, else the modules"""
    mods = seq if isinstance(seq, (list, tuple)) else list(seq)
    layers = parse(mods)
    if supported(layers, x):
        return run(layers, x)
    for m in mods:
        x = m(x)
    return x


# ---- several stacks of the same depth on the same rows: one launch per layer for all of them ----------------------------------
class _RowFwdArgs(ctypes.Structure):      # o3d_row_fwd_args
    _fields_ = [("X", _vp), ("ldx", _i), ("W", _vp), ("bias", _vp), ("gamma", _vp), ("beta", _vp), ("running_mean", _vp),
                ("running_var", _vp), ("momentum", _f), ("eps", _f), ("training", _i), ("relu", _i), ("R", _i), ("Cin", _i),
                ("Cout", _i), ("Z", _vp), ("Y", _vp), ("mean", _vp), ("invstd", _vp)]


class _RowBwdArgs(ctypes.Structure):      # o3d_row_bwd_args
    _fields_ = [("dY", _vp), ("lddy", _i), ("dZup", _vp), ("Wup", _vp), ("Cup", _i), ("input_mode", _i), ("dX", _vp),
                ("lddx", _i), ("Z", _vp), ("gamma", _vp), ("beta", _vp), ("mean", _vp), ("invstd", _vp), ("training", _i),
                ("relu", _i), ("X", _vp), ("ldx", _i), ("R", _i), ("Cin", _i), ("C", _i), ("dZ", _vp), ("dW", _vp), ("db", _vp),
                ("dgamma", _vp), ("dbeta", _vp)]


for _n in ("o3d_row_mlp_fwd_group", "o3d_row_mlp_bwd_group", "o3d_row_mlp_input_grad"):
    capi.register(_n, [_vp, _i, _vp])
GROUP_MAX = 4


class RowStackGroup(torch.autograd.Function):
    """apply(cfgs, x (R, Cin), *params) -> one (R, Cout_last) per stack; params = the stacks' [W, b, gamma, beta] x L
    concatenated, every stack L layers deep on the SAME rows x: layer l of all stacks is one launch (csrc/rowmlp.hip,
    row_mlp_{fwd,bwd}_group_kernel), and so is the gradient of x (the sum over the stacks).  M2-Track's motion, motion-state
    and previous-box heads (models/m2track.py:60-71): 9 + 12 launches become 3 + 4."""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, cfgs, x, *params):
        lib = capi.load()
        G = len(cfgs)
        L = len(params) // (4 * G)
        dev, f32 = x.device, torch.float32
        st = _stream()
        need_bwd = any(ctx.needs_input_grad)
        h0 = x.detach()
        if h0.stride(1) != 1:
            h0 = h0.contiguous()
        R = h0.shape[0]
        hs = [h0] * G
        saved = [[] for _ in range(G)]
        for l in range(L):
            jobs = (_RowFwdArgs * G)()
            outs = []
            for g, cfg in enumerate(cfgs):
                W, b, gamma, beta = params[4 * (g * L + l):4 * (g * L + l) + 4]
                bn = cfg.bns[l]
                Cout, Cin = W.shape
                Y = torch.empty((R, Cout), device=dev, dtype=f32)
                Z = torch.empty((R, Cout), device=dev, dtype=f32) if (bn is not None and need_bwd) else None
                stat = torch.empty((2, Cout), device=dev, dtype=f32) if (bn is not None and need_bwd) else None
                h = hs[g]
                jobs[g] = _RowFwdArgs(h.data_ptr(), h.stride(0), W.data_ptr(), _ptr(b), _ptr(gamma), _ptr(beta),
                                      _ptr(bn.running_mean) if bn is not None else None,
                                      _ptr(bn.running_var) if bn is not None else None,
                                      float(bn.momentum) if bn is not None else 0.0, float(bn.eps) if bn is not None else 0.0,
                                      int(cfg.training), int(cfg.relus[l]), R, Cin, Cout, _ptr(Z), Y.data_ptr(),
                                      _ptr(stat[0]) if stat is not None else None, _ptr(stat[1]) if stat is not None else None)
                saved[g].append((h, Z, stat))
                outs.append(Y)
            _call("row_mlp_fwd", 0.0, lib.o3d_row_mlp_fwd_group, ctypes.addressof(jobs), G, st)
            hs = outs
        bns = [bn for cfg in cfgs if cfg.training for bn in cfg.bns if bn is not None]
        if bns:
            count_batches(bns, 1)
        if need_bwd:
            ctx.cfgs, ctx.saved, ctx.params, ctx.L = cfgs, saved, params, L
            ctx.versions = [(p, p._version) for p in params if p is not None]
        ctx.set_materialize_grads(False)
        return tuple(hs)

    @staticmethod
    @capi.on_tensor_device
    def backward(ctx, *dOuts):
        lib = capi.load()
        cfgs, params, L = ctx.cfgs, ctx.params, ctx.L
        for p, v in ctx.versions:
            if p._version != v:
                raise RuntimeError("a parameter of a fused row stack was modified in place between forward and backward")
        G = len(cfgs)
        x0 = ctx.saved[0][0][0]
        dev, f32 = x0.device, torch.float32
        st = _stream()
        R = x0.shape[0]
        tops = []
        for g in range(G):
            d = dOuts[g]
            if d is None:               # a head nothing downstream read
                d = torch.zeros((R, params[4 * (g * L + L - 1)].shape[0]), device=dev, dtype=f32)
            tops.append(d if d.stride(1) == 1 else d.contiguous())
        grads = [None] * (4 * L * G)
        ups = [None] * G                # (dZ, W) of the layer above
        for l in range(L - 1, -1, -1):
            jobs = (_RowBwdArgs * G)()
            keep = []
            for g, cfg in enumerate(cfgs):
                W, b, gamma, beta = params[4 * (g * L + l):4 * (g * L + l) + 4]
                X, Z, stat = ctx.saved[g][l]
                C, Cin = W.shape
                dZ = torch.empty((R, C), device=dev, dtype=f32)
                dW = torch.empty((C, Cin), device=dev, dtype=f32)
                db = torch.empty((C,), device=dev, dtype=f32) if b is not None else None
                dgb = torch.empty((2, C), device=dev, dtype=f32) if gamma is not None else None
                up = ups[g]
                jobs[g] = _RowBwdArgs(tops[g].data_ptr() if up is None else None, tops[g].stride(0) if up is None else 0,
                                      up[0].data_ptr() if up is not None else None, up[1].data_ptr() if up is not None else None,
                                      up[1].shape[0] if up is not None else 0, 0, None, 0, _ptr(Z), _ptr(gamma), _ptr(beta),
                                      _ptr(stat[0]) if stat is not None else None, _ptr(stat[1]) if stat is not None else None,
                                      int(cfg.training), int(cfg.relus[l]), X.data_ptr(), X.stride(0), R, Cin, C, dZ.data_ptr(),
                                      dW.data_ptr(), _ptr(db), _ptr(dgb[0]) if dgb is not None else None,
                                      _ptr(dgb[1]) if dgb is not None else None)
                o = 4 * (g * L + l)
                grads[o], grads[o + 1] = dW, db
                if dgb is not None:
                    grads[o + 2], grads[o + 3] = dgb[0], dgb[1]
                keep.append((dZ, W))
            _call("row_mlp_bwd", 0.0, lib.o3d_row_mlp_bwd_group, ctypes.addressof(jobs), G, st)
            ups = keep
        dx = None
        if ctx.needs_input_grad[1]:
            Cin0 = params[0].shape[1]
            dx = torch.empty((R, Cin0), device=dev, dtype=f32)
            jobs = (_RowBwdArgs * G)()
            for g in range(G):
                dZ, W = ups[g]
                jobs[g] = _RowBwdArgs(None, 0, dZ.data_ptr(), W.data_ptr(), W.shape[0], 1, dx.data_ptr(), Cin0, None, None, None,
                                      None, None, 0, 0, None, 0, R, 0, Cin0, None, None, None, None, None)
            _call("row_mlp_bwd", 0.0, lib.o3d_row_mlp_input_grad, ctypes.addressof(jobs), G, st)
        return (None, dx, *grads)


def seq_rows_group(seqs, x):
    """[nn.Sequential of Linear / BatchNorm1d / ReLU] applied to the same rows x -> one output per sequence: one launch per
    layer for all of them when they are equally deep (RowStackGroup), else one stack at a time (seq_rows)"""
    parsed = [parse(s if isinstance(s, (list, tuple)) else list(s)) for s in seqs]
    if (1 < len(seqs) <= GROUP_MAX and all(supported(p, x) for p in parsed) and len({len(p) for p in parsed}) == 1):
        cfgs, params = [], []
        for layers in parsed:
            cfg = _Cfg()
            cfg.bns = [bn for _, bn, _ in layers]
            cfg.relus = [bool(r) for _, _, r in layers]
            cfg.training = bool(any(bn is not None and bn.training for bn in cfg.bns))
            cfgs.append(cfg)
            for lin, bn, _ in layers:
                params += [lin.weight, lin.bias, bn.weight if bn is not None else None, bn.bias if bn is not None else None]
        return list(RowStackGroup.apply(tuple(cfgs), x, *params))
    return [seq_rows(s, x) for s in seqs]
