"""P2B's template<->search fusion (models/head/xcorr.py:25-53: cosine similarity + template xyz + template
features -> SharedMLP [4+f,h,h,h] -> max over the template axis) on the library's kernels, without the
(B, 4+f, M, N) fusion tensor the reference materialises (8.5 MB per pair in, 3 x 8.4 MB per pair of activations).

Only the similarity channel depends on the search point, so layer 0 splits exactly like the set-abstraction layer 0
(csrc/xcorr.hip): Z = W0[:, 1:] . [xyz ; feat] is a GEMM over the M template points and
Y0[c, (j,i)] = Z[c, i] + W0[c, 0] * sim[j, i].  The M template points of one search point are a "ball" of M contiguous
columns, so layers 1.. and the max-pool are the kernels of the set-abstraction levels (csrc/mlp_direct.hip,
csrc/mlp_wgrad.hip, csrc/compact.hip) with every column live.  The cosine similarity itself (a (B,N,M) map from two
small GEMM-shaped products) is one kernel each way (`CosineSimMap`, round 4; torch.bmm + norms until then); its gradient
arrives from the layer-0 backward kernel.
"""
import ctypes

import torch

from . import capi
from .fused import _call, _const_vec, _eval_consts, _layers, _ptr, _stream, count_batches
from .fused_heads import _up, pack_rows, prep_for

_vp, _i, _l = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
capi.register("o3d_xcorr_expand", [_vp, _l, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp])
capi.register("o3d_xcorr_reduce_groups", [_i])
capi.register("o3d_pool_bwd_partials_split", [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp])
capi.register("o3d_xcorr_reduce", [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _l, _vp, _vp, _vp])


def supported(mlp, t_feat, s_feat):
    layers = _layers(mlp)
    if layers is None or len(layers) < 2 or not t_feat.is_cuda:
        return False
    B, f, M = t_feat.shape
    N = s_feat.shape[2]
    if M < 4 or M > 64 or (M & (M - 1)) or (M * N) % 1024 or layers[0][0].in_channels != f + 4:
        return False
    return all(conv.out_channels % 64 == 0 for conv, _ in layers) and B * M * N <= 0x7fffffff and (B * M) % 64 == 0


_STATIC = {}


def _zero_row(dev, B, M):
    """the similarity channel's slot in the per-point operand: a (B,1,M) block of zeros, built once per shape"""
    key = (str(dev), B, M)
    if key not in _STATIC:
        _STATIC[key] = torch.zeros((B, 1, M), dtype=torch.float32, device=dev)
    return _STATIC[key]


class _Cfg:
    __slots__ = ("bns", "training")


class FusedP2BXCorr(torch.autograd.Function):
    """apply(cfg, sim (B,N,M), t_xyz (B,M,3), t_feat (B,f,M), W0,g0,b0, W1,g1,b1, ...) -> (B, C_last, N)"""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, cfg, sim, t_xyz, t_feat, *params):
        lib = capi.load()
        L = len(params) // 3
        Ws = [params[3 * l].detach().reshape(params[3 * l].shape[0], -1) for l in range(L)]
        gammas = [params[3 * l + 1].detach() for l in range(L)]
        betas = [params[3 * l + 2].detach() for l in range(L)]
        B, N, M = sim.shape
        f = t_feat.shape[1]
        P = B * N * M
        Pm = B * M
        dev, f32 = sim.device, torch.float32
        st = _stream()
        prep = prep_for(dev)
        zero1 = _zero_row(dev, B, M)
        simf = sim.detach().contiguous()
        need_bwd = any(ctx.needs_input_grad)
        C0 = Ws[0].shape[0]
        K0 = 4 + f
        K0p = _up(K0, 64)
        # per-point operand [0 ; xyz ; feat] (row 0 = the similarity channel's slot, zero): Z = W0 . X0
        X0 = pack_rows([zero1, t_xyz.detach().transpose(1, 2), t_feat.detach()], K0p)
        W0p = prep.get(params[0], C0, K0p)
        Z = torch.empty((C0, Pm), device=dev, dtype=f32)
        _call("conv_fwd_points", 2.0 * K0p * C0 * Pm, lib.o3d_pw_fwd, X0.data_ptr(), W0p.data_ptr(), None, None, None, None,
              K0p, C0, Pm, Z.data_ptr(), None, None, st)
        Ys, vecs, Wts = [], [], []
        tile = lib.o3d_pw_tile(P, 256)          # the xcorr stage has far more than 65536 columns: 128, whatever the rows
        for l in range(L):
            Cout, Cin = Ws[l].shape
            bn = cfg.bns[l]
            Y = torch.empty((Cout, P), device=dev, dtype=f32)
            nparts = P // 256 if l == 0 else P // tile
            part = torch.empty((nparts, 2, Cout), device=dev, dtype=f32) if cfg.training else None
            statc = bn.running_mean if cfg.training else None
            if l == 0:
                _call("xcorr_expand", 0.0, lib.o3d_xcorr_expand, Z.data_ptr(), Pm, simf.data_ptr(), Ws[0].data_ptr(), K0, B, M, N,
                      C0, Y.data_ptr(), _ptr(part), _ptr(statc), st)
            else:
                _call("conv_fwd", 2.0 * Cin * Cout * P, lib.o3d_pw_fwd, Ys[-1].data_ptr(), Ws[l].data_ptr(),
                      vecs[-1][2].data_ptr(), vecs[-1][3].data_ptr(), None, None, Cin, Cout, P, Y.data_ptr(), _ptr(part),
                      _ptr(statc), st, dims=(Cin, Cout))
                if need_bwd:
                    Wts.append(prep.get(params[3 * l], Cin, Cout, transpose=True))
            vec = torch.empty((4, Cout), device=dev, dtype=f32)
            if cfg.training:
                fold = torch.empty((64, Cout), device=dev, dtype=f32)
                _call("bn_finalize", 0.0, lib.o3d_bn_finalize, part.data_ptr(), nparts, Cout, float(P), statc.data_ptr(),
                      gammas[l].data_ptr(), betas[l].data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                      float(bn.momentum), float(bn.eps), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
                      vec[3].data_ptr(), fold.data_ptr(), st)
            else:
                _eval_consts(lib, bn, gammas[l], betas[l], vec, 1, st)
            Ys.append(Y)
            vecs.append(vec)
        if cfg.training:
            count_batches(cfg.bns, 1)
        Cl = Ws[-1].shape[0]
        # max over the template axis = over the M contiguous columns of a ball: the regular (slot) pooling kernel on the
        # flat layout (one "cloud" of B*N balls); the pooled tensor comes out flat (Cl, B*N), the layout the
        # fea_layer stack consumes without a pack launch
        out = torch.empty((Cl, B * N), device=dev, dtype=f32)
        argq = torch.empty((Cl, B * N), device=dev, dtype=torch.int32) if need_bwd else None
        yarg = torch.empty((Cl, B * N), device=dev, dtype=f32) if need_bwd else None
        _call("pool_fwd", 0.0, lib.o3d_bn_relu_maxpool_fwd, Ys[-1].data_ptr(), vecs[-1][2].data_ptr(), vecs[-1][3].data_ptr(),
              1, Cl, B * N, M, out.data_ptr(), _ptr(argq), _ptr(yarg), st)
        if need_bwd:
            ctx.cfg = cfg
            ctx.geom = (B, N, M, f, K0, K0p, tile)
            ctx.versions = [(p, p._version) for p in params]
            W0t = prep.get(params[0], K0p, C0, transpose=True)
            ctx.saved = (X0, simf, Ys, vecs, Ws, Wts, W0t, gammas, out.detach(), argq, yarg)
        return out.view(Cl, B, N).permute(1, 0, 2)

    @staticmethod
    @capi.on_tensor_device
    def backward(ctx, dOut):
        lib = capi.load()
        cfg = ctx.cfg
        B, N, M, f, K0, K0p, tile = ctx.geom
        for p, v in ctx.versions:
            if p._version != v:
                raise RuntimeError("a parameter of the fused xcorr MLP was modified in place between forward and backward")
        X0, simf, Ys, vecs, Ws, Wts, W0t, gammas, out, argq, yarg = ctx.saved
        L = len(Ws)
        P, Pm = B * N * M, B * M
        dev, f32 = dOut.device, torch.float32
        st = _stream()
        Cl = Ws[-1].shape[0]
        g = dOut.permute(1, 0, 2)
        g = g.reshape(Cl, B * N) if g.is_contiguous() else g.contiguous().view(Cl, B * N)
        # the dense (Cl, P) gradient of the pooled layer (one non-zero per ball and channel) is never written: the last
        # layer's data / weight gradient kernels read the packed {masked gradient, arg-max slot} pairs
        nparts = 8
        part = torch.empty((nparts, 2, Cl), device=dev, dtype=f32)
        pk = torch.empty((Cl, B * N, 2), device=dev, dtype=f32)
        _call("pool_bwd", 0.0, lib.o3d_pool_bwd_partials_split, g.data_ptr(), out.data_ptr(), yarg.data_ptr(),
              vecs[-1][0].data_ptr(), Cl, B * N, nparts, part.data_ptr(), argq.data_ptr(), pk.data_ptr(), st)
        dN = None
        grads = [None] * (3 * L)
        dsim = dxyz = dfeat = None
        for l in range(L - 1, -1, -1):
            Cout, Cin = Ws[l].shape
            coef = torch.empty((5, Cout), device=dev, dtype=f32)
            fold = torch.empty((64, Cout), device=dev, dtype=f32)
            _call("bn_bwd_finalize", 0.0, lib.o3d_bn_bwd_finalize, part.data_ptr(), nparts, Cout, float(P), gammas[l].data_ptr(),
                  vecs[l][0].data_ptr(), vecs[l][1].data_ptr(), coef[0].data_ptr(), coef[1].data_ptr(), coef[2].data_ptr(),
                  coef[3].data_ptr(), coef[4].data_ptr(), fold.data_ptr(), st)
            if not cfg.training:
                coef[3].zero_()
                coef[4].zero_()
            grads[3 * l + 1], grads[3 * l + 2] = coef[0], coef[1]
            A = (coef[2].data_ptr(), coef[3].data_ptr(), coef[4].data_ptr())
            if l == 0:
                S = torch.empty((Cout, Pm), device=dev, dtype=f32)
                ng = lib.o3d_xcorr_reduce_groups(Cout)
                dsp = torch.empty((ng, P), device=dev, dtype=f32)
                dwp = torch.empty((B, Cout), device=dev, dtype=f32)
                _call("xcorr_reduce", 0.0, lib.o3d_xcorr_reduce, dN.data_ptr(), Ys[0].data_ptr(), A[0], A[1], A[2],
                      simf.data_ptr(), Ws[0].data_ptr(), K0, B, M, N, Cout, S.data_ptr(), Pm, dsp.data_ptr(), dwp.data_ptr(), st)
                one, zero = _const_vec(dev, Cout, 1.0), _const_vec(dev, Cout, 0.0)
                dWm = torch.empty((Cout, K0p), device=dev, dtype=f32)
                scratch = torch.empty((lib.o3d_mlp_conv_wgrad2_scratch(1, K0p, Cout, Pm),), device=dev, dtype=f32)
                _call("conv_wgrad_points", 2.0 * K0p * Cout * Pm, lib.o3d_mlp_conv_wgrad2, S.data_ptr(), None, 4, S.data_ptr(),
                      one.data_ptr(), zero.data_ptr(), zero.data_ptr(), X0.data_ptr(), None, None, 1, K0p, Cout, Pm,
                      scratch.data_ptr(), dWm.data_ptr(), st)
                dW0 = dWm[:, :K0].clone()
                dW0[:, 0] = dwp.sum(0)                     # the similarity channel's weight (its X0 row is zero)
                grads[0] = dW0
                if ctx.needs_input_grad[1]:
                    dsim = dsp.sum(0).view(B, N, M)
                if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
                    dX0 = torch.empty((K0p, Pm), device=dev, dtype=f32)
                    _call("conv_dgrad_points", 2.0 * K0p * Cout * Pm, lib.o3d_pw_dgrad, S.data_ptr(), None, None, None, None,
                          W0t.data_ptr(), K0p, Cout, Pm, None, None, None, None, None, dX0.data_ptr(), None, st)
                    if ctx.needs_input_grad[2]:
                        dxyz = dX0[1:4].view(3, B, M).permute(1, 2, 0)
                    if ctx.needs_input_grad[3]:
                        dfeat = dX0[4:4 + f].view(f, B, M).permute(1, 0, 2)
                continue
            dW = torch.empty((Cout, Cin), device=dev, dtype=f32)
            scratch = torch.empty((lib.o3d_mlp_conv_wgrad2_scratch(1, Cin, Cout, P),), device=dev, dtype=f32)
            _call("conv_wgrad", 2.0 * Cin * Cout * P, lib.o3d_mlp_conv_wgrad2, _ptr(dN), pk.data_ptr() if dN is None else None,
                  M if dN is None else 4, Ys[l].data_ptr(), A[0], A[1], A[2], Ys[l - 1].data_ptr(), vecs[l - 1][2].data_ptr(),
                  vecs[l - 1][3].data_ptr(), 1, Cin, Cout, P, scratch.data_ptr(), dW.data_ptr(), st,
                  dims=(Cin, Cout, dN is None))
            grads[3 * l] = dW
            dNp = torch.empty((Cin, P), device=dev, dtype=f32)
            vp = vecs[l - 1]
            if dN is None:         # pooled source: 128-column tiles (one statistics partial row each)
                part = torch.empty((P // 128, 2, Cin), device=dev, dtype=f32)
                _call("conv_dgrad", 2.0 * Cin * Cout * P, lib.o3d_mlp_conv_dgrad_wt, None, g.data_ptr(), out.data_ptr(),
                      argq.data_ptr(), M, Ys[l].data_ptr(), A[0], A[1], A[2], Ws[l].data_ptr(), Wts[l - 1].data_ptr(),
                      pk.data_ptr(), 1, Cin, Cout, P, Ys[l - 1].data_ptr(), vp[2].data_ptr(), vp[3].data_ptr(), vp[0].data_ptr(),
                      dNp.data_ptr(), part.data_ptr(), st, dims=(Cin, Cout, True))
                nparts = P // 128
            else:
                part = torch.empty((P // tile, 2, Cin), device=dev, dtype=f32)
                _call("conv_dgrad", 2.0 * Cin * Cout * P, lib.o3d_pw_dgrad, dN.data_ptr(), Ys[l].data_ptr(), A[0], A[1], A[2],
                      Wts[l - 1].data_ptr(), Cin, Cout, P, Ys[l - 1].data_ptr(), vp[2].data_ptr(), vp[3].data_ptr(),
                      vp[0].data_ptr(), None, dNp.data_ptr(), part.data_ptr(), st, dims=(Cin, Cout))
                nparts = P // tile
            dN = dNp
        gw = []
        for l in range(L):
            gw += [grads[3 * l].view(Ws[l].shape[0], Ws[l].shape[1], 1, 1), grads[3 * l + 1], grads[3 * l + 2]]
        return (None, dsim, dxyz, dfeat, *gw)


capi.register("o3d_cosine_sim_fwd", [_vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp, _vp, _vp, _vp])
capi.register("o3d_cosine_sim_bwd", [_vp, _vp, _vp, _vp, _vp, _l, _l, _l, _vp, _l, _l, _l, _i, _i, _i, _i, _vp, _vp, _vp])


class CosineSimMap(torch.autograd.Function):
    """apply(t (B,f,M), s (B,f,N)) -> sim (B,N,M) = <t_i, s_j> / (max(|t_i|, eps) * max(|s_j|, eps)), eps = 1e-8:
    nn.CosineSimilarity(dim=1) of models/head/xcorr.py:37-38 on the never-materialised (B,f,M,N) expansion, one kernel
    each way (csrc/xcorr.hip); any strides (the features are (B,C,N) views of the flat conv_final output)."""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, t, s):
        lib = capi.load()
        B, f, M = t.shape
        N = s.shape[2]
        dev = t.device
        td, sd = t.detach(), s.detach()
        sim = torch.empty((B, N, M), device=dev, dtype=torch.float32)
        tn = torch.empty((B, M), device=dev, dtype=torch.float32)
        sn = torch.empty((B, N), device=dev, dtype=torch.float32)
        _call("cosine_sim", 0.0, lib.o3d_cosine_sim_fwd, td.data_ptr(), *td.stride(), sd.data_ptr(), *sd.stride(), B, f, M, N,
              sim.data_ptr(), tn.data_ptr(), sn.data_ptr(), _stream())
        if any(ctx.needs_input_grad):
            # NB: a detached alias, never `sim` itself: the returned tensor's grad_fn is this node, so storing it on ctx is a
            # reference cycle that keeps the step's whole autograd graph alive until the GC runs (and the HIP-graph capture
            # of the P2B step died in capture_end on exactly that)
            ctx.saved = (td, sd, sim.detach(), tn, sn)
        return sim

    @staticmethod
    @capi.on_tensor_device
    def backward(ctx, dsim):
        lib = capi.load()
        td, sd, sim, tn, sn = ctx.saved
        B, f, M = td.shape
        N = sd.shape[2]
        dsim = dsim.contiguous()
        dt = torch.empty((B, f, M), device=td.device, dtype=torch.float32)
        ds = torch.empty((B, f, N), device=td.device, dtype=torch.float32)
        _call("cosine_sim_bwd", 0.0, lib.o3d_cosine_sim_bwd, dsim.data_ptr(), sim.data_ptr(), tn.data_ptr(), sn.data_ptr(),
              td.data_ptr(), *td.stride(), sd.data_ptr(), *sd.stride(), B, f, M, N, dt.data_ptr(), ds.data_ptr(), _stream())
        return (dt if ctx.needs_input_grad[0] else None), (ds if ctx.needs_input_grad[1] else None)


def cosine_sim_supported(t, s):
    return (t.is_cuda and t.dtype == torch.float32 and s.dtype == torch.float32 and t.shape[1] % 32 == 0 and
            t.shape[2] <= 64 and t.shape[2] % 4 == 0 and s.shape[2] <= 128)


def p2b_xcorr_mlp_pool(mlp, template_feature, search_feature, template_xyz):
    """(B,f,M), (B,f,N), (B,M,3) -> (B, h, N): SharedMLP over [sim ; xyz ; feat] + max over the template axis"""
    # cosine similarity as the (B,N,M) map the kernels index by column: <t,s> / (max(|t|,eps) * max(|s|,eps))
    # (nn.CosineSimilarity, eps = 1e-8, xcorr.py:37-38)
    if cosine_sim_supported(template_feature, search_feature):
        sim = CosineSimMap.apply(template_feature, search_feature)
    else:       # shapes outside the kernel's tile limits: the same map on torch ops
        tn = template_feature.norm(dim=1).clamp_min(1e-8)                      # (B,M)
        sn = search_feature.norm(dim=1).clamp_min(1e-8)                        # (B,N)
        sim = torch.bmm(search_feature.transpose(1, 2), template_feature) / (sn.unsqueeze(2) * tn.unsqueeze(1))
    layers = _layers(mlp)
    cfg = _Cfg()
    cfg.training, cfg.bns = bool(mlp.training), [bn for _, bn in layers]
    params = []
    for conv, bn in layers:
        params += [conv.weight, bn.weight, bn.bias]
    return FusedP2BXCorr.apply(cfg, sim, template_xyz, template_feature, *params)
