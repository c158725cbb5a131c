"""Per-point MLP chains (Conv1d 1x1 -> BatchNorm1d -> ReLU)* on the library's GEMM kernels.

The M2-Track stacks (models/backbone/pointnet.py:91-204) are the grouped MLP with ONE ball per cloud
(SURVEY.md section 8f-1): no gather, flat (C, P = B*N) activations, BatchNorm+ReLU applied by the
consumer kernel on load, backward folded into per-channel constants exactly as in open3dsot_amd/fused.py.
A chain ends either in its materialised activation ("act") or in the global max over the N points of
each cloud ("gmax", AdaptiveMaxPool1d(1)).  Parameters stay in the caller's nn.Conv1d / nn.BatchNorm1d
modules.  A Conv1d bias in front of a training-mode BatchNorm cancels in the output and has zero
gradient; it only shifts the running mean, which is updated accordingly.
"""
import contextlib
import ctypes

import torch

from . import capi
from .fused import _call, _check_versions, _const_vec, _eval_consts, _ptr, _stream, _versions, count_batches, POOL_BWD_SPLIT, TILE
from .fused import bias_fix as _fused_bias_fix

_vp, _i, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
capi.register("o3d_pw_tile", [_l, _i])
capi.register("o3d_pw_fwd", [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, _vp, _vp, _vp, _vp])
capi.register("o3d_pw_dgrad", [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
capi.register("o3d_gmax_bwd_pk", [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp])
capi.register("o3d_thin_bwd_scratch", [])
capi.register("o3d_thin_bwd", [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, _vp, _vp, _vp, _vp])
capi.register("o3d_bn_relu_apply", [_vp, _vp, _vp, _i, _l, _vp, _vp])
capi.register("o3d_act_bwd_partials", [_vp, _vp, _vp, _vp, _vp, _i, _l, _vp, _vp, _vp])
capi.register("o3d_gmax_fwd", [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp])
capi.register("o3d_pw_fwd_cloud", [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp])
capi.register("o3d_cloud_sum_dy", [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp])


class _Cfg:
    __slots__ = ("mode", "training", "bns")


_POOLED_GMAX = {"on": True}      # test hook: the dense-gradient backward of a "gmax" stack (the path it replaced)


def _flat_ok(rows, k, P):
    """the flat-layout entry points of csrc/mlp_direct.hip (o3d_pw_fwd / o3d_pw_dgrad): they pick the wave tile -- and with it
    the number of statistics partial rows, P // o3d_pw_tile(P, rows) -- from the size of the launch"""
    return rows % 64 == 0 and k % 16 == 0 and P % 128 == 0


def supported(x, layers):
    B, C, N = x.shape
    return x.is_cuda and N % TILE == 0 and all(c.kernel_size == (1,) and c.stride == (1,) and c.padding == (0,) and
                                               c.groups == 1 for c, _ in layers)


class FusedPointwiseChain(torch.autograd.Function):
    """(x (B,Cin,N), cbias (C_0,B) | None, cfg, W0,b0,g0,beta0, W1,...) -> act (B,C_L,N) | pooled (B,C_L)
    `cbias`: per-cloud bias added to layer 0's output before its BatchNorm (see `chain_cloud`)."""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, x, cbias, cfg, *params):
        lib = capi.load()
        L = len(params) // 4
        Ws = [params[4 * l].detach()[:, :, 0].contiguous() for l in range(L)]
        biases = [params[4 * l + 1] for l in range(L)]
        gammas = [params[4 * l + 2].detach().contiguous() for l in range(L)]
        betas = [params[4 * l + 3].detach().contiguous() for l in range(L)]
        B, Cin0, N = x.shape
        P = B * N
        dev, f32 = x.device, torch.float32
        st = _stream()
        # flat (Cin, B*N) operand: a (B,C,N) view of another stack's flat buffer (the usual case between M2-Track's stacks) is
        # that buffer, no copy -- round 4 made it (B,C,N)-contiguous first and flat again here: two 25 MB copies per hand-over
        xp = x.detach().permute(1, 0, 2)
        X0 = (xp if xp.is_contiguous() else xp.contiguous()).view(Cin0, P)
        ntiles = P // TILE
        Ys, means, invstds, scales, shifts = [], [], [], [], []
        bias_fix = {}
        for l in range(L):
            Cout, Cin = Ws[l].shape
            bn = cfg.bns[l]
            Y = torch.empty((Cout, P), device=dev, dtype=f32)
            flat = _flat_ok(Cout, Cin, P) and not (l == 0 and cbias is not None)
            nrows = P // lib.o3d_pw_tile(P, Cout) if flat else ntiles
            part = torch.empty((nrows, 2, Cout), device=dev, dtype=f32) if cfg.training else None
            stat_c = bn.running_mean if cfg.training else None
            src = X0 if l == 0 else Ys[-1]
            if flat:
                _call("pw_conv_fwd", 2.0 * Cin * Cout * P, lib.o3d_pw_fwd, src.data_ptr(), Ws[l].data_ptr(),
                      None if l == 0 else scales[-1].data_ptr(), None if l == 0 else shifts[-1].data_ptr(), None, None, Cin, Cout,
                      P, Y.data_ptr(), _ptr(part), _ptr(stat_c), st, dims=(Cin, Cout))
            elif l == 0 and cbias is not None:
                cb = cbias.detach().contiguous()
                _call("pw_conv_fwd", 2.0 * Cin * Cout * P, lib.o3d_pw_fwd_cloud, src.data_ptr(), Ws[0].data_ptr(), cb.data_ptr(),
                      B, N, Cin, Cout, Y.data_ptr(), _ptr(part), _ptr(stat_c), st, dims=(Cin, Cout))
            else:
                _call("pw_conv_fwd", 2.0 * Cin * Cout * P, lib.o3d_mlp_conv_fwd, src.data_ptr(), Ws[l].data_ptr(),
                      None if l == 0 else scales[-1].data_ptr(), None if l == 0 else shifts[-1].data_ptr(), 1, Cin, Cout, P,
                      Y.data_ptr(), _ptr(part), _ptr(stat_c), st, dims=(Cin, Cout))
            vec = torch.empty((4, Cout), device=dev, dtype=f32)
            b = biases[l].detach() if biases[l] is not None else None
            if cfg.training:
                fold = torch.empty((64, Cout), device=dev, dtype=f32)
                _call("bn_finalize", 0.0, lib.o3d_bn_finalize, part.data_ptr(), nrows, Cout, float(P), stat_c.data_ptr(),
                      gammas[l].data_ptr(), betas[l].data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                      float(bn.momentum), float(bn.eps), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
                      vec[3].data_ptr(), fold.data_ptr(), st)
                if b is not None:       # statistics were taken without the bias: mean(Y + b) = mean(Y) + b
                    bias_fix.setdefault(float(bn.momentum), ([], []))
                    bias_fix[float(bn.momentum)][0].append(bn.running_mean)
                    bias_fix[float(bn.momentum)][1].append(b)
            else:
                _eval_consts(lib, bn, gammas[l], betas[l], vec, 1, st, conv_bias=b)
            Ys.append(Y)
            means.append(vec[0]); invstds.append(vec[1]); scales.append(vec[2]); shifts.append(vec[3])
        for mom, (rms, bs) in bias_fix.items():      # one multi-tensor launch per FORWARD inside a tracker scope, else per stack
            _fused_bias_fix(mom, rms, bs)
        if cfg.training:
            count_batches(cfg.bns, 1)
        Cl = Ws[-1].shape[0]
        need_bwd = any(ctx.needs_input_grad)
        argq = yarg = None
        if cfg.mode == "act":
            act = torch.empty((Cl, P), device=dev, dtype=f32)
            _call("pw_apply", 0.0, lib.o3d_bn_relu_apply, Ys[-1].data_ptr(), scales[-1].data_ptr(), shifts[-1].data_ptr(),
                  Cl, P, act.data_ptr(), st)
            out = act.view(Cl, B, N).permute(1, 0, 2)
        else:
            out = torch.empty((B, Cl), device=dev, dtype=f32)
            argq = torch.empty((B, Cl), device=dev, dtype=torch.int32) if need_bwd else None
            yarg = torch.empty((B, Cl), device=dev, dtype=f32) if need_bwd else None
            _call("pw_gmax", 0.0, lib.o3d_gmax_fwd, Ys[-1].data_ptr(), scales[-1].data_ptr(), shifts[-1].data_ptr(), B, Cl,
                  N, out.data_ptr(), _ptr(argq), _ptr(yarg), st)
        if need_bwd:
            # W^T of the inner layers for the data gradient: from the device's WeightPrep table (inside a tracker forward
            # all of them were refreshed by ONE launch) instead of one transposing copy per layer in the backward
            ctx.Wts = None
            if all(isinstance(params[4 * l], torch.nn.Parameter) for l in range(1, L)):
                from .fused_heads import prep_for
                prep = prep_for(dev)
                ctx.Wts = [None] + [prep.get(params[4 * l], Ws[l].shape[1], Ws[l].shape[0], transpose=True) for l in range(1, L)]
            ctx.cfg = cfg
            ctx.versions = _versions(params)
            if X0.data_ptr() == x.data_ptr():
                # X0 ALIASES the caller's storage (a view of the producing stack's flat buffer, no copy): an in-place write to
                # that tensor between forward and backward would silently corrupt dW0 and dX -- tracked like the parameters
                ctx.versions.append((x, x._version))
            ctx.dims = (B, N, L)
            # (identity only: which parameter a weight gradient belongs to; a sliced weight -- chain_cloud's W[:, :C] -- is not
            # a leaf autograd merely stores, so it is never branched)
            ctx.wparams = [params[4 * l] if isinstance(params[4 * l], torch.nn.Parameter) else None for l in range(L)]
            ctx.has_bias = [b is not None for b in biases]
            ctx.has_cbias = cbias is not None
            ctx.saved = (X0, Ws, gammas, Ys, means, invstds, scales, shifts, out.detach() if cfg.mode != "act" else None,
                         argq, yarg)
        return out

    @staticmethod
    @capi.on_tensor_device
    def backward(ctx, dOut):
        lib = capi.load()
        cfg = ctx.cfg
        _check_versions(ctx.versions, "FusedPointwiseChain")
        B, N, L = ctx.dims
        X0, Ws, gammas, Ys, means, invstds, scales, shifts, out, argq, yarg = ctx.saved
        P = B * N
        dev, f32 = dOut.device, torch.float32
        st = _stream()
        ntiles = P // TILE
        Cl = Ws[-1].shape[0]
        dN = torch.empty((Cl, P), device=dev, dtype=f32) if not (cfg.mode != "act" and L >= 2 and Cl % 64 == 0 and
                                                                   Ws[-1].shape[1] % 64 == 0 and N % 128 == 0 and
                                                                   _POOLED_GMAX["on"]) else None
        pk = None
        if cfg.mode == "act":
            g = dOut.permute(1, 0, 2).reshape(Cl, P).contiguous()
            part = torch.empty((ntiles, 2, Cl), device=dev, dtype=f32)
            _call("pw_act_bwd", 0.0, lib.o3d_act_bwd_partials, g.data_ptr(), Ys[-1].data_ptr(), scales[-1].data_ptr(),
                  shifts[-1].data_ptr(), means[-1].data_ptr(), Cl, P, dN.data_ptr(), part.data_ptr(), st)
            nparts = ntiles
        elif L >= 2 and Cl % 64 == 0 and Ws[-1].shape[1] % 64 == 0 and N % 128 == 0 and _POOLED_GMAX["on"]:
            # the last layer's GEMMs build the one-hot gradient of the max on the fly (csrc/pointwise.hip::gmax_bwd_pk_kernel)
            dOut = dOut.contiguous()
            pk = torch.empty((Cl, B, 2), device=dev, dtype=f32)
            part = torch.empty((1, 2, Cl), device=dev, dtype=f32)
            _call("pool_bwd", 0.0, lib.o3d_gmax_bwd_pk, dOut.data_ptr(), out.data_ptr(), argq.data_ptr(), yarg.data_ptr(),
                  means[-1].data_ptr(), B, Cl, N, pk.data_ptr(), part.data_ptr(), st)
            nparts, dN = 1, None
        else:
            dOut = dOut.contiguous()
            part = torch.empty((POOL_BWD_SPLIT, 2, Cl), device=dev, dtype=f32)
            meta = _meta_full(dev, P, B)       # every column is live: one "ball" of N columns per cloud
            _call("pool_bwd", 0.0, lib.o3d_pool_bwd_c, dOut.data_ptr(), out.data_ptr(), argq.data_ptr(), yarg.data_ptr(),
                  means[-1].data_ptr(), B, Cl, 1, 0, meta.data_ptr(), 0, P, dN.data_ptr(), part.data_ptr(), st)
            nparts = POOL_BWD_SPLIT
        grads = [None] * (4 * L)
        dx = dcb = None
        zero_bias = None      # dL/db behind a training-mode BatchNorm is exactly zero: one fill for all layers of the stack
        if cfg.training and any(ctx.has_bias):
            widths = [Ws[l].shape[0] if ctx.has_bias[l] else 0 for l in range(L)]
            flat0 = torch.zeros((sum(widths),), device=dev, dtype=f32)
            zero_bias, o = [], 0
            for w in widths:
                zero_bias.append(flat0[o:o + w])
                o += w
        for l in range(L - 1, -1, -1):
            Cout, Cin = Ws[l].shape
            coef = torch.empty((5, Cout), device=dev, dtype=f32)
            fold = torch.empty((64, Cout), device=dev, dtype=f32)
            _call("bn_bwd_finalize", 0.0, lib.o3d_bn_bwd_finalize, part.data_ptr(), nparts, Cout, float(P),
                  gammas[l].data_ptr(), means[l].data_ptr(), invstds[l].data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(),
                  coef[2].data_ptr(), coef[3].data_ptr(), coef[4].data_ptr(), fold.data_ptr(), st)
            if not cfg.training:
                coef[3].zero_()
                coef[4].zero_()
            grads[4 * l + 2], grads[4 * l + 3] = coef[0], coef[1]
            if ctx.has_bias[l]:      # dL/db = sum dY: zero behind a training-mode BatchNorm, A1 * sum dN in eval mode
                if cfg.training:
                    grads[4 * l + 1] = zero_bias[l] if zero_bias is not None else torch.zeros_like(coef[0])
                else:
                    grads[4 * l + 1] = coef[2] * coef[1]
            A = (coef[2].data_ptr(), coef[3].data_ptr(), coef[4].data_ptr())
            dW = torch.empty((Cout, Cin), device=dev, dtype=f32)
            flops = 2.0 * Cin * Cout * P
            if l == 0 and ctx.has_cbias and ctx.needs_input_grad[1]:
                dcb = torch.empty((Cout, B), device=dev, dtype=f32)       # gradient of the per-cloud bias
                _call("cloud_sum", 0.0, lib.o3d_cloud_sum_dy, dN.data_ptr(), Ys[0].data_ptr(), A[0], A[1], A[2], Cout, B, N,
                      dcb.data_ptr(), st)
            thin = l == 0 and Cout == 64 and Cin <= 16       # weight and input gradient in one pass (csrc/pointwise.hip)
            if thin:
                wpart = torch.empty((lib.o3d_thin_bwd_scratch(),), device=dev, dtype=f32)
                dX = torch.empty((Cin, B, N), device=dev, dtype=f32) if ctx.needs_input_grad[0] else None
                _call("pw_conv_wgrad", flops * (2 if dX is not None else 1), lib.o3d_thin_bwd, dN.data_ptr(), Ys[0].data_ptr(), A[0],
                      A[1], A[2], X0.data_ptr(), Ws[0].data_ptr(), Cin, Cout, P, wpart.data_ptr(), dW.data_ptr(), _ptr(dX), st,
                      dims=(Cin, Cout))
                if dX is not None:
                    dx = dX.permute(1, 0, 2)
            elif Cin % 64 == 0 and Cout % 64 == 0:
                wpart = torch.empty((lib.o3d_mlp_conv_wgrad2_scratch(1, Cin, Cout, P),), device=dev, dtype=f32)
                xs = (X0.data_ptr(), None, None) if l == 0 else \
                     (Ys[l - 1].data_ptr(), scales[l - 1].data_ptr(), shifts[l - 1].data_ptr())
                # the weight gradient is a leaf of the backward: on the side branch when one is open (fused.wgrad_branch)
                from .fused import _branch_side
                side = _branch_side([ctx.wparams[l]], (dN, pk, coef, wpart, ctx.saved)) if ctx.wparams[l] is not None else None
                with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                    _call("pw_conv_wgrad", flops, lib.o3d_mlp_conv_wgrad2, _ptr(dN), _ptr(pk) if dN is None else None,
                          N if dN is None else 4, Ys[l].data_ptr(), A[0], A[1],
                          A[2], xs[0], xs[1], xs[2], 1, Cin, Cout, P, wpart.data_ptr(), dW.data_ptr(),
                          side.cuda_stream if side is not None else st, dims=(Cin, Cout))
            else:
                tiles = ((Cin + 127) // 128) * ((Cout + 127) // 128)
                nsl = max(1, min(P // 32 // 4, 768 // tiles))
                wpart = torch.empty((nsl + 16, Cout, Cin), device=dev, dtype=f32)
                xs = (X0.data_ptr(), None, None) if l == 0 else \
                     (Ys[l - 1].data_ptr(), scales[l - 1].data_ptr(), shifts[l - 1].data_ptr())
                _call("pw_conv_wgrad", flops, lib.o3d_mlp_conv_wgrad, dN.data_ptr(), None, None, None, 4, Ys[l].data_ptr(),
                      A[0], A[1], A[2], xs[0], xs[1], xs[2], None, None, None, None, 0, 0, 0, 1.0, 1, Cin, Cout, P, nsl,
                      wpart.data_ptr(), dW.data_ptr(), st, dims=(Cin, Cout))
            grads[4 * l] = dW.unsqueeze(-1)
            if l >= 1:
                Wt = ctx.Wts[l] if ctx.Wts is not None else Ws[l].t().contiguous()
                dNp = torch.empty((Cin, P), device=dev, dtype=f32)
                if dN is None:           # pooled source: 128-column tiles, one ball of N columns per cloud
                    nrows = ntiles
                    part = torch.empty((ntiles, 2, Cin), device=dev, dtype=f32)
                    _call("pw_conv_dgrad", flops, lib.o3d_mlp_conv_dgrad_wt, None, None, None, None, N,
                          Ys[l].data_ptr(), A[0], A[1], A[2], Ws[l].data_ptr(), Wt.data_ptr(), pk.data_ptr(), 1, Cin, Cout, P,
                          Ys[l - 1].data_ptr(), scales[l - 1].data_ptr(), shifts[l - 1].data_ptr(), means[l - 1].data_ptr(),
                          dNp.data_ptr(), part.data_ptr(), st, dims=(Cin, Cout))
                elif _flat_ok(Cin, Cout, P):
                    nrows = P // lib.o3d_pw_tile(P, Cin)
                    part = torch.empty((nrows, 2, Cin), device=dev, dtype=f32)
                    _call("pw_conv_dgrad", flops, lib.o3d_pw_dgrad, dN.data_ptr(), Ys[l].data_ptr(), A[0], A[1], A[2],
                          Wt.data_ptr(), Cin, Cout, P, Ys[l - 1].data_ptr(), scales[l - 1].data_ptr(), shifts[l - 1].data_ptr(),
                          means[l - 1].data_ptr(), None, dNp.data_ptr(), part.data_ptr(), st, dims=(Cin, Cout))
                else:
                    nrows = ntiles
                    part = torch.empty((ntiles, 2, Cin), device=dev, dtype=f32)
                    _call("pw_conv_dgrad", flops, lib.o3d_mlp_conv_dgrad_wt, dN.data_ptr(), None, None, None, 4,
                          Ys[l].data_ptr(), A[0], A[1], A[2], Ws[l].data_ptr(), Wt.data_ptr(), None, 1, Cin, Cout, P,
                          Ys[l - 1].data_ptr(), scales[l - 1].data_ptr(), shifts[l - 1].data_ptr(), means[l - 1].data_ptr(),
                          dNp.data_ptr(), part.data_ptr(), st, dims=(Cin, Cout))
                dN, nparts = dNp, nrows
            elif ctx.needs_input_grad[0] and not thin:
                dX = torch.empty((Cin, B, N), device=dev, dtype=f32)
                # (the LDS-staged kernel: on 64 <- 512 x 98 304 columns it takes 0.113 ms, the direct 128-column tile 0.150 ms --
                # 768 one-wave chains of 512 MFMAs -- and the split-K tile 0.183 ms; profiles/r04_ab_m2track_narrow_tiles.txt)
                _call("pw_conv_dgrad", flops, lib.o3d_mlp_conv_dgrad_plain, dN.data_ptr(), Ys[0].data_ptr(), A[0], A[1],
                      A[2], Ws[0].data_ptr(), 1, Cin, Cout, P, dX.data_ptr(), st, dims=(Cin, Cout))
                dx = dX.permute(1, 0, 2)
        return (dx, dcb, None, *grads)


_META = {}


def _meta_full(dev, P, B):
    key = (str(dev), P, B)
    if key not in _META:
        _META[key] = torch.tensor([P, P, B, 0], dtype=torch.int32, device=dev)
    return _META[key]


def chain(x, layers, mode):
    """layers = [(conv1d, batchnorm1d)]; mode "act" -> (B,C_L,N), "gmax" -> (B,C_L)"""
    cfg = _Cfg()
    cfg.mode, cfg.training, cfg.bns = mode, bool(layers[0][1].training), [bn for _, bn in layers]
    params = []
    for conv, bn in layers:
        params += [conv.weight, conv.bias, bn.weight, bn.bias]
    return FusedPointwiseChain.apply(x, None, cfg, *params)


def cloud_supported(x, pooled, layers):
    conv0 = layers[0][0]
    B, C, N = x.shape
    return supported(x, layers) and pooled.dim() == 2 and pooled.shape[0] == B and conv0.in_channels == C + pooled.shape[1] \
        and C % 64 == 0 and conv0.out_channels % 64 == 0 and N % 128 == 0


def chain_cloud(x, pooled, layers, mode="act"):
    """chain(cat([x, pooled expanded over the N points]), layers): SegPointNet's second per-point stack
    (models/backbone/pointnet.py:188-193).  The pooled feature is the same at every point of a cloud, so its 1024 input
    channels of the first layer are NOT run through the GEMM at all 2048 points: W_b . pooled[b] is a per-cloud bias
    (C_0, B) -- one small matmul -- added in the GEMM's epilogue before the BatchNorm statistics.  Exact; removes
    1024/1088 of that layer's FLOPs (40 % of the whole M2-Track step).  Backward: the bias gradient is the per-cloud
    column sum of dY, which autograd carries through the small matmul into W_b and the pooled feature."""
    conv0, bn0 = layers[0]
    C = x.shape[1]
    w = conv0.weight                                  # (C_0, C + Cp, 1)
    cbias = torch.mm(w[:, C:, 0], pooled.t())         # (C_0, B)
    cfg = _Cfg()
    cfg.mode, cfg.training, cfg.bns = mode, bool(bn0.training), [bn for _, bn in layers]
    params = [w[:, :C], conv0.bias, bn0.weight, bn0.bias]
    for conv, bn in layers[1:]:
        params += [conv.weight, conv.bias, bn.weight, bn.bias]
    return FusedPointwiseChain.apply(x, cbias, cfg, *params)
