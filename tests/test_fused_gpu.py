"""GPU parity of the fused grouped-MLP kernels (csrc/mlp.hip) against plain PyTorch references
of the same operators (this is the floating-point kernel of the path, so a torch reference is
the yardstick).  Forward features: 1e-4 relative to the tensor scale, as north_star states
(measured ~1e-6).  Gradients are judged against an fp64 shadow of the composed operator chain:
ReLU masks and max-pool winners are discrete, so two correct fp32 evaluations whose forward
values differ by 1e-6 can route a gradient differently at a near-tie; an fp32-vs-fp32 max-norm
comparison is therefore noisy (measured on the GPU box: torch's own CPU fp32 backward is 2e-2
away from its fp64 run on some RPN weights, tools/diag_grad2.py).  Criterion: relative L2 error
<= 5e-4 and max error <= 1e-2 of the tensor scale, against fp64."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-20))


def l2rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def assert_grad_close(a, b, what, l2tol=5e-4, maxtol=1e-2):
    # (eval-mode BatchNorm has no normalisation to damp a routing flip: callers pass l2tol=3e-3 there)
    assert l2rel(a, b) < l2tol and rel(a, b) < maxtol, (what, l2rel(a, b), rel(a, b))


def shadow64(mlp, xyz, new_xyz, feats, idx, train, inv_radius=1.0, ties=None):
    """fp64 restatement of QueryAndGroup -> SharedMLP -> max over nsample on torch ops
    (pointnet2_utils.py:299-339, pytorch_utils.py:12-37, pointnet2_modules.py:69-73).
    -> (out, leaves: dict name -> fp64 leaf, buffers: dict name -> fp64 running stat)
    ties: a list -> receives the per-ball near-tie margin (B, npoint) in fp32 ulps (tests/flip_proof.py): the minimum over
    every ReLU unit of the ball (all layers, channels, slots) and over the max-pool candidates"""
    B, npoint, ns = idx.shape
    flat = idx.long().reshape(B, 1, npoint * ns)
    leaves, bufs = {}, {}

    def leaf(name, t):
        if t is None:
            return None
        leaves[name] = t.detach().double().clone().requires_grad_(True)
        return leaves[name]
    parts = []
    if xyz is not None:
        x64, n64 = leaf("xyz", xyz), leaf("new_xyz", new_xyz)
        g = x64.transpose(1, 2).gather(2, flat.expand(B, 3, -1)).reshape(B, 3, npoint, ns)
        parts.append((g - n64.transpose(1, 2).unsqueeze(-1)) * inv_radius)
    if feats is not None:
        f64 = leaf("feats", feats)
        parts.append(f64.gather(2, flat.expand(B, f64.shape[1], -1)).reshape(B, -1, npoint, ns))
    x = torch.cat(parts, dim=1)
    for name, layer in mlp.named_children():
        conv, bn = layer.conv, layer.bn.bn
        x = F.conv2d(x, leaf(name + ".conv.weight", conv.weight))
        rm, rv = bn.running_mean.detach().double().clone(), bn.running_var.detach().double().clone()
        x = F.batch_norm(x, rm, rv, leaf(name + ".bn.bn.weight", bn.weight), leaf(name + ".bn.bn.bias", bn.bias),
                         train, bn.momentum, bn.eps)
        bufs[name + ".bn.bn.running_mean"], bufs[name + ".bn.bn.running_var"] = rm, rv
        if ties is not None:
            import flip_proof
            m = flip_proof.relu_margin_ulps(x.detach(), 1).amin(dim=-1)
            ties[:] = [m if not ties else torch.minimum(ties[0], m)]
        x = F.relu(x)
    if ties is not None:
        ties[:] = [torch.minimum(ties[0], flip_proof.pool_margin_ulps(x.detach(), 1))]
    return x.max(dim=-1)[0], leaves, bufs


@pytest.fixture(scope="module")
def lib():
    from open3dsot_amd import capi, fused  # noqa: F401  (registers the signatures)
    return capi.load()


def st():
    return torch.cuda.current_stream().cuda_stream


@pytest.mark.parametrize("B,Cin,Cout,P,xform", [(2, 64, 64, 256, True), (3, 64, 128, 128, True), (2, 128, 256, 384, True),
                                                (1, 256, 256, 128, False), (2, 16, 64, 128, True), (2, 40, 200, 256, True)])
def test_conv_fwd_and_stats(lib, B, Cin, Cout, P, xform):
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + Cin + Cout)
    X = torch.randn(B, Cin, P, device="cuda", generator=g)
    W = torch.randn(Cout, Cin, device="cuda", generator=g) * 0.2
    sc = torch.rand(Cin, device="cuda", generator=g) + 0.5
    sh = torch.randn(Cin, device="cuda", generator=g) * 0.3
    c = torch.randn(Cout, device="cuda", generator=g) * 0.1
    Y = torch.full((B, Cout, P), float("nan"), device="cuda")
    ntiles = B * (P // 128)
    part = torch.full((ntiles, 2, Cout), float("nan"), device="cuda")
    rc = lib.o3d_mlp_conv_fwd(X.data_ptr(), W.data_ptr(), sc.data_ptr() if xform else None,
                              sh.data_ptr() if xform else None, B, Cin, Cout, P, Y.data_ptr(), part.data_ptr(),
                              c.data_ptr(), st())
    assert rc == 0
    Xin = F.relu(X * sc[None, :, None] + sh[None, :, None]) if xform else X
    ref = torch.einsum("oc,bcp->bop", W.double(), Xin.double())
    assert rel(Y, ref) < 2e-6 * max(1, Cin ** 0.5), rel(Y, ref)
    s = part[:, 0, :].double().sum(0)
    q = part[:, 1, :].double().sum(0)
    assert rel(s, ref.sum((0, 2))) < 1e-5
    assert rel(q, ((ref - c.double()[None, :, None]) ** 2).sum((0, 2))) < 1e-5


@pytest.mark.parametrize("B,C,P", [(4, 96, 256), (40, 96, 512)])     # 8 parts / 160 parts (folded first)
def test_bn_finalize_matches_torch(lib, B, C, P):
    torch.manual_seed(0)
    Y = torch.randn(B, C, P, device="cuda") * 2 + 3
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.1)
        bn.running_mean.normal_(0, 0.5); bn.running_var.uniform_(0.5, 2)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    ntiles = B * P // 128
    Yt = Y.view(B, C, P // 128, 128)
    part = torch.stack([Yt.sum(3), ((Yt - rm[None, :, None, None]) ** 2).sum(3)], 0)  # (2,B,C,T)
    part = part.permute(1, 3, 0, 2).reshape(ntiles, 2, C).contiguous()
    vec = torch.empty(4, C, device="cuda")
    fold = torch.empty(64, C, device="cuda")
    rc = lib.o3d_bn_finalize(part.data_ptr(), ntiles, C, float(B * P), rm.data_ptr(), bn.weight.data_ptr(),
                             bn.bias.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5, vec[0].data_ptr(),
                             vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), fold.data_ptr(), st())
    assert rc == 0
    ref = bn(Y.view(B, C, P, 1))
    got = Y * vec[2][None, :, None] + vec[3][None, :, None]
    assert rel(got, ref.squeeze(-1)) < 1e-5
    assert rel(rm, bn.running_mean) < 1e-5 and rel(rv, bn.running_var) < 1e-5


def make_case(kind, B=3, train=True, seed=0, full=False):
    """(grouper, mlp, xyz, new_xyz, feats) on the GPU for a layer shape family of the tracker; `full`: the SEARCH
    branch's point counts at BASELINE config 2 (1024-point search cloud) instead of the half-size test clouds"""
    from open3dsot_amd import nn_blocks, ops, synth
    torch.manual_seed(seed)
    if kind == "sa1":
        N, npoint, ns, r, C, spec = 512, 256, 32, 0.3, 0, [3, 64, 64, 128]
    elif kind == "sa2":
        N, npoint, ns, r, C, spec = 256, 128, 32, 0.5, 128, [131, 128, 128, 256]
    elif kind == "sa3":
        N, npoint, ns, r, C, spec = 128, 64, 32, 0.7, 256, [259, 256, 256, 256]
    elif kind == "rpn":
        N, npoint, ns, r, C, spec = 128, 64, 16, 0.3, 257, [260, 256, 256, 256]
    else:
        raise ValueError(kind)
    if full and kind != "rpn":
        N, npoint = 2 * N, 2 * npoint
    b = synth.make_batch(900 + seed, B, 512, 1024)
    xyz = torch.from_numpy(b["search_points"][:, :N, :]).cuda()
    if kind != "sa1" and not full:
        xyz = xyz * 0.5
    new_xyz = xyz[:, :npoint, :].contiguous()
    feats = torch.randn(B, C, N, device="cuda") if C else None
    grouper = ops.QueryAndGroup(r, ns, use_xyz=True)
    mlp = nn_blocks.SharedMLP(list(spec), bn=True).cuda().train(train)
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.empty(m.weight.shape).uniform_(0.5, 1.5, generator=g))
                m.bias.copy_(torch.empty(m.bias.shape).normal_(0, 0.2, generator=g))
                m.running_mean.copy_(torch.empty(m.bias.shape).normal_(0, 0.2, generator=g))
                m.running_var.copy_(torch.empty(m.bias.shape).uniform_(0.5, 1.5, generator=g))
    return grouper, mlp, xyz, new_xyz, feats


def composed(grouper, mlp, xyz, new_xyz, feats):
    x = mlp(grouper(xyz, new_xyz, feats))
    return F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1)


@pytest.mark.parametrize("kind", ["sa1", "sa2", "sa3", "rpn"])
@pytest.mark.parametrize("train", [True, False])
def test_fused_sa_matches_composed(kind, train):
    import copy
    from open3dsot_amd import fused
    grouper, mlp, xyz, new_xyz, feats = make_case(kind, train=train)
    mlp_ref = copy.deepcopy(mlp)
    want_xyz = kind == "rpn"
    leaves = []
    for t in (xyz, new_xyz, feats):
        leaves.append(t.clone().requires_grad_(True) if t is not None and (t is feats or want_xyz) else t)
    leaves_ref = [t.detach().clone().requires_grad_(t.requires_grad) if t is not None else None for t in leaves]
    idx = grouper.query(xyz, new_xyz)
    ref64, l64, b64 = shadow64(mlp_ref, xyz, new_xyz, feats, idx, train)
    out = fused.sa_group_mlp_pool(grouper, mlp, *leaves)
    ref = composed(grouper, mlp_ref, *leaves_ref)          # torch fp32 (MIOpen) on the same HIP index ops
    assert out.shape == ref.shape
    assert rel(out, ref) < 1e-4, ("forward vs torch fp32", rel(out, ref))
    assert rel(out, ref64) < 2e-5, ("forward vs fp64 shadow", rel(out, ref64))
    go = torch.randn(out.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    if not train and not any(t is not None and t.requires_grad for t in leaves):
        return
    out.backward(go)
    ref64.backward(go.double())
    for n1, p1 in mlp.named_parameters():
        assert p1.grad is not None, n1
        assert_grad_close(p1.grad, l64[n1].grad, n1, l2tol=5e-4 if train else 3e-3)
    for nm, a in zip(("xyz", "new_xyz", "feats"), leaves):
        if a is not None and a.requires_grad:
            assert_grad_close(a.grad, l64[nm].grad, nm, l2tol=5e-4 if train else 3e-3)
    if train:
        for n1, b1 in mlp.named_buffers():
            if b1.dtype.is_floating_point:
                assert rel(b1, b64[n1]) < 1e-5, n1
            else:
                assert int(b1) == 1, n1


@pytest.mark.parametrize("kind", ["sa1", "sa2", "sa3"])
def test_paired_segments_full_size_batch48(kind):
    """BASELINE config 2 itself: 48 pairs, template 512 / search 1024 points -- the per-level column counts of the
    benchmarked step (search SA1 48*512*32 = 786 432 slots), where the launch geometry differs from the small cases
    (128-column wave tiles above 65 536 slots, segment 1 at a non-zero `start1`, the weight-gradient slice plan):
    forward <= 2e-5 and running statistics <= 1e-5 against the fp64 shadow; gradients against fp64 with torch's own
    fp32 evaluation as the yardstick for the discrete routing flips (see _paired_case)."""
    _paired_case(kind, True, B=48, full=True)


@pytest.mark.parametrize("kind", ["sa1", "sa2", "sa3"])
@pytest.mark.parametrize("train", [True, False])
def test_paired_segments_match_two_calls(kind, train):
    """The template and the search cloud through one shared SA module in ONE set of launches
    (fused.sa_group_mlp_pool_pair) = two consecutive calls, template first: separate batch statistics,
    running statistics updated in order, parameter gradients summed (models/bat.py:89-90)."""
    _paired_case(kind, train)


def _paired_case(kind, train, B=3, full=False):
    import copy
    from open3dsot_amd import fused
    grouper, mlp, xyz_s, new_s, feats_s = make_case(kind, B=B, train=train, full=full)
    N, npoint = xyz_s.shape[1], new_s.shape[1]
    xyz_t = (xyz_s[:, :N // 2, :] * 0.9 + 0.05).contiguous()          # the template: half the points
    new_t = xyz_t[:, :npoint // 2, :].contiguous()
    feats_t = torch.randn(feats_s.shape[0], feats_s.shape[1], N // 2, device="cuda") if feats_s is not None else None
    mlp_ref = copy.deepcopy(mlp)
    mlp_ref0 = copy.deepcopy(mlp)
    want_xyz = kind == "sa2"
    segs, refs, outs64, idxs, margins = [], [], [], [], []
    prove = full and train          # stage 1 / stage 2 of tests/flip_proof.py
    for xyz, new_xyz, feats in ((xyz_t, new_t, feats_t), (xyz_s, new_s, feats_s)):
        segs.append([t.clone().requires_grad_(True) if t is not None and (t is feats or want_xyz) else t
                     for t in (xyz, new_xyz, feats)])
        idx = grouper.query(xyz, new_xyz)
        ties = [] if prove else None
        o64, l64, b64 = shadow64(mlp_ref, xyz, new_xyz, feats, idx, train, ties=ties)
        idxs.append(idx)
        margins.append(ties[0] if prove else None)
        outs64.append(o64)
        refs.append(l64)
        if train:      # the second call starts from the running statistics the first one left
            with torch.no_grad():
                for n1, b1 in mlp_ref.named_buffers():
                    if n1 in b64:
                        b1.copy_(b64[n1])
    outs = fused.sa_group_mlp_pool_pair(grouper, mlp, tuple(segs[0]), tuple(segs[1]))
    assert outs is not None and len(outs) == 2
    for o, o64 in zip(outs, outs64):
        assert o.shape == o64.shape and o.is_contiguous()
        assert rel(o, o64) < 2e-5, ("forward vs fp64 shadow", rel(o, o64))
    if train:
        for n1, b1 in mlp.named_buffers():
            if b1.dtype.is_floating_point:
                assert rel(b1, b64[n1]) < 1e-5, n1
            else:
                assert int(b1) == 2, n1
    if not train and not any(t is not None and t.requires_grad for sg in segs for t in sg):
        return
    gen = torch.Generator(device="cuda").manual_seed(3)
    gos = [torch.randn(o.shape, device="cuda", generator=gen) for o in outs]
    torch.autograd.backward(list(outs), gos)
    for o64, go in zip(outs64, gos):
        o64.backward(go.double(), retain_graph=prove)
    tol = 5e-4 if train else 6e-3      # eval: nothing damps an argmax flip, and two clouds contribute flips
    yard = {}
    if full:
        # ReLU masks and max-pool winners are discrete: at 48 pairs a layer has 10^7-10^8 activations, a handful of which
        # sit within fp32 rounding of the decision boundary and are routed differently than in fp64 -- each such flip
        # moves a parameter gradient by ~1e-3 L2 (profiles/r02_relu_flip_diag.txt locates one: a single column carries
        # the whole error, the rest agrees to 5e-7).  torch's own fp32 evaluation (MIOpen convolutions + BatchNorm on
        # the same HIP index operators, the two module calls in order) flips elsewhere: measured on the MI355X at
        # B = 48 it is 8e-3 from fp64 on SA1 (fused: 2e-3), 4e-4 on SA2 (fused: 1e-3), 3e-4 on SA3 (fused: 4e-4).
        # Bound: 5e-4, or three times torch's error, or 3e-3 (a few flips), whichever is largest.
        m32 = copy.deepcopy(mlp_ref0)
        segs32 = [[t.detach().clone().requires_grad_(t.requires_grad) if t is not None else None for t in sg] for sg in segs]
        outs32 = [composed(grouper, m32, *sg) for sg in segs32]
        torch.autograd.backward(outs32, gos)
        for n1, p32 in m32.named_parameters():
            yard[n1] = l2rel(p32.grad, refs[0][n1].grad + refs[1][n1].grad)
        for si, (sg, l64) in enumerate(zip(segs32, refs)):
            for nm, a in zip(("xyz", "new_xyz", "feats"), sg):
                if a is not None and a.requires_grad:
                    yard["%d.%s" % (si, nm)] = l2rel(a.grad, l64[nm].grad)
    report, checks = {}, []
    for n1, p1 in mlp.named_parameters():
        want = refs[0][n1].grad + refs[1][n1].grad
        report[n1] = (l2rel(p1.grad, want), yard.get(n1))
        checks.append((p1.grad, want, n1, max(tol, 3 * yard.get(n1, 0.0), 3e-3 if full else 0.0), (3e-2 if full else 1e-2) if train else 2e-2))
    for si, (sg, l64) in enumerate(zip(segs, refs)):
        for nm, a in zip(("xyz", "new_xyz", "feats"), sg):
            if a is not None and a.requires_grad:
                key = "%d.%s" % (si, nm)
                report[key] = (l2rel(a.grad, l64[nm].grad), yard.get(key))
                checks.append((a.grad, l64[nm].grad, key, max(tol, 3 * yard.get(key, 0.0), 3e-3 if full else 0.0), (1e-1 if full else 1e-2) if train else 4e-2))
    if full:
        print("B=%d %s gradient L2 error vs fp64 (fused, torch fp32):" % (B, kind),
              {k: ("%.1e" % v[0], "%.1e" % v[1]) for k, v in report.items()})
    for got, want, what, l2tol, maxtol in checks:
        assert_grad_close(got, want, what, l2tol=l2tol, maxtol=maxtol)
    if prove:
        _prove_flips("paired %s B=%d" % (kind, B), mlp_ref0, segs, refs, outs64, idxs, margins, gos,
                     lambda m, sg: fused.sa_group_mlp_pool_pair(grouper, m, tuple(sg[0]), tuple(sg[1])))


@pytest.mark.parametrize("kind", ["rpn", "xcorr"])
def test_full_size_batch48_single_segment_flip_proof(kind):
    """The two grouped MLPs of the step that are not a template/search pair, at the benchmarked 48 pairs: the RPN's vote
    aggregation (64 balls x 16 of 128 votes, every input gradient live: rpn.py:55-60) and BoxAwareXCorr's k-NN group
    (128 x 4 of 64 template points, xcorr.py:89-100).  Forward 2e-5 against fp64; gradients by tests/flip_proof.py."""
    import copy
    from open3dsot_amd import fused, nn_blocks
    B = 48
    if kind == "rpn":
        grouper, mlp, xyz, new_xyz, feats = make_case("rpn", B=B, train=True)
        seg = [t.clone().requires_grad_(True) for t in (xyz, new_xyz, feats)]
        idx = grouper.query(xyz, new_xyz)
        run = lambda m, sg: [fused.sa_group_mlp_pool(grouper, m, *sg[0])]
    else:
        torch.manual_seed(4)
        M, N, k, f = 64, 128, 4, 256
        bundle = torch.randn(B, 3 + 9 + f, M, device="cuda")
        idx = torch.randint(0, M, (B, N, k), device="cuda", dtype=torch.int32)
        mlp = nn_blocks.SharedMLP([3 + 9 + f, 256, 256, 256], bn=True).cuda().train()
        xyz = new_xyz = None
        feats = bundle
        seg = [None, None, bundle.clone().requires_grad_(True)]
        run = lambda m, sg: [fused.group_mlp_pool(m, sg[0][2], idx)]
    mlp0 = copy.deepcopy(mlp)
    ties = []
    o64, l64, b64 = shadow64(copy.deepcopy(mlp), xyz, new_xyz, feats, idx, True, ties=ties)
    out = run(mlp, [seg])[0]
    assert rel(out, o64) < 2e-5, rel(out, o64)
    for n1, b1 in mlp.named_buffers():
        if b1.dtype.is_floating_point:
            assert rel(b1, b64[n1]) < 1e-5, n1
    go = torch.randn(out.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    out.backward(go)
    o64.backward(go.double(), retain_graph=True)
    for n1, p1 in mlp.named_parameters():        # loose first (flips included), then the proof
        assert l2rel(p1.grad, l64[n1].grad) < 3e-3, (n1, l2rel(p1.grad, l64[n1].grad))
    _prove_flips("%s B=%d" % (kind, B), mlp0, [seg], [l64], [o64], [idx], [ties[0]], [go], run)


def _prove_flips(tag, mlp0, segs, refs, outs64, idxs, margins, gos, run):
    """tests/flip_proof.py on a paired full-size case: (1) every input-gradient outlier sits on a point of a ball with an
    fp64 near-tie; (2) with the cotangent zeroed on those balls, every gradient meets the TIGHT bound"""
    import copy
    import flip_proof as fp
    from open3dsot_amd import fused
    flagged = [m < fp.TIE_ULPS for m in margins]                       # (B, npoint) per segment
    for si, (sg, l64, idx, fl, m) in enumerate(zip(segs, refs, idxs, flagged, margins)):
        B, npoint, ns = idx.shape
        N = sg[0].shape[1] if sg[0] is not None else sg[2].shape[2]
        # a point is flagged when any flagged ball lists it; its margin = the smallest margin of the balls listing it
        pmargin = torch.full((B, N), float("inf"), device=idx.device, dtype=torch.float64)
        pmargin.scatter_reduce_(1, idx.long().reshape(B, -1), m[:, :, None].expand(B, npoint, ns).reshape(B, -1), "amin")
        pflag = pmargin < fp.TIE_ULPS
        # a flip also moves every OTHER column a little, through the BatchNorm-backward batch means: ~1/positions of its
        # own size.  At the benchmarked sizes (>= 10^5 positions) that is far below OUTLIER; the small cases (12 288
        # positions: measured 1.8e-3 of the rms beside a flipped column at 7e-2) get the floor 40 / positions
        outlier = max(fp.OUTLIER, 40.0 / (B * npoint * ns))
        for nm, a, cols, flg, mar in (("xyz", sg[0], (0, 1), pflag, pmargin), ("new_xyz", sg[1], (0, 1), fl, m),
                                      ("feats", sg[2], (0, 2), pflag, pmargin)):
            if a is not None and a.requires_grad:
                out, emap = fp.outlier_columns(a.grad, l64[nm].grad, cols, outlier)
                fp.check_outliers_flagged("%s seg %d d%s" % (tag, si, nm), out, emap, flg, mar, outlier)
    # stage 2: both evaluations again with the cotangent zeroed on the flagged balls
    mlp2 = copy.deepcopy(mlp0)
    segs2 = [[t.detach().clone().requires_grad_(t.requires_grad) if t is not None else None for t in sg] for sg in segs]
    gos2 = [go * (~fl)[:, None, :].to(go.dtype) for go, fl in zip(gos, flagged)]
    for l64 in refs:
        for t in l64.values():
            t.grad = None
    outs2 = run(mlp2, segs2)
    torch.autograd.backward(list(outs2), gos2)
    for o64, go in zip(outs64, gos2):
        o64.backward(go.double())
    pairs = [(n1, p1.grad, sum(l64[n1].grad for l64 in refs)) for n1, p1 in mlp2.named_parameters()]
    for si, (sg, l64) in enumerate(zip(segs2, refs)):
        for nm, a in zip(("xyz", "new_xyz", "feats"), sg):
            if a is not None and a.requires_grad:
                pairs.append(("%d.%s" % (si, nm), a.grad, l64[nm].grad))
    fp.record("%s: flagged balls %s of %s" % (tag, [int(f.sum()) for f in flagged], [f.numel() for f in flagged]))
    fp.check_tight(tag, pairs)


@pytest.mark.parametrize("mode", ["single", "paired", "eval"])
def test_fused_sa_normalize_xyz(mode):
    """QueryAndGroup(normalize_xyz=True) (pointnet2_utils.py:321-322: grouped_xyz /= radius): the scaling branch of the
    fused path (per-point operand and ball centres scaled by 1/radius, xyz gradients scaled back) against the fp64
    shadow and the composed torch chain -- one call, the paired template/search launch, and the one-kernel eval path"""
    import copy
    from open3dsot_amd import fused, ops
    train = mode != "eval"
    _, mlp, xyz, new_xyz, feats = make_case("sa2", train=train)
    grouper = ops.QueryAndGroup(0.5, 32, use_xyz=True, normalize_xyz=True)
    inv_r = 1.0 / 0.5
    mlp_ref, mlp32, mlp0 = copy.deepcopy(mlp), copy.deepcopy(mlp), copy.deepcopy(mlp)
    idxs, margins = [], []
    segs = [(xyz, new_xyz, feats)]
    if mode == "paired":
        N, npoint = xyz.shape[1], new_xyz.shape[1]
        xt = (xyz[:, :N // 2, :] * 0.9 + 0.05).contiguous()
        segs = [(xt, xt[:, :npoint // 2, :].contiguous(), torch.randn(feats.shape[0], feats.shape[1], N // 2, device="cuda")),
                (xyz, new_xyz, feats)]
    leaves = [[t.clone().requires_grad_(train) for t in sg] for sg in segs]
    outs64, refs = [], []
    for sg in segs:
        idx = grouper.query(sg[0], sg[1])
        ties = []
        o64, l64, b64 = shadow64(mlp_ref, *sg, idx, train, inv_radius=inv_r, ties=ties)
        outs64.append(o64)
        refs.append(l64)
        idxs.append(idx)
        margins.append(ties[0])
        if train:
            with torch.no_grad():
                for n1, b1 in mlp_ref.named_buffers():
                    if n1 in b64:
                        b1.copy_(b64[n1])
    if mode == "paired":
        outs = fused.sa_group_mlp_pool_pair(grouper, mlp, tuple(leaves[0]), tuple(leaves[1]))
    else:
        outs = [fused.sa_group_mlp_pool(grouper, mlp, *leaves[0])]
    with torch.no_grad():
        ref32 = [composed(grouper, mlp32, *sg) for sg in segs]
    for o, o32, o64 in zip(outs, ref32, outs64):
        assert rel(o, o32) < 1e-4 and rel(o, o64) < 2e-5, (rel(o, o32), rel(o, o64))
    if not train:
        return
    gen = torch.Generator(device="cuda").manual_seed(2)
    gos = [torch.randn(o.shape, device="cuda", generator=gen) for o in outs]
    torch.autograd.backward(list(outs), gos)
    for o64, go in zip(outs64, gos):
        o64.backward(go.double(), retain_graph=True)
    for n1, b1 in mlp.named_buffers():
        if b1.dtype.is_floating_point:
            assert rel(b1, b64[n1]) < 1e-5, n1
    for n1, p1 in mlp.named_parameters():        # loose (a routing flip may be in it), then the proof with the tight bound
        assert l2rel(p1.grad, sum(l64[n1].grad for l64 in refs)) < 5e-3, n1
    run = (lambda m, sg: fused.sa_group_mlp_pool_pair(grouper, m, tuple(sg[0]), tuple(sg[1]))) if mode == "paired" else \
        (lambda m, sg: [fused.sa_group_mlp_pool(grouper, m, *sg[0])])
    _prove_flips("normalize_xyz %s" % mode, mlp0, leaves, refs, outs64, idxs, margins, gos, run)


def test_prefix_indices_are_shared_and_intact():
    """the arange(npoint) sample indices of the non-FPS levels (pointnet2_modules.py:56) are one cached tensor per
    (B, npoint, device), never written by the module; the cache is BOUNDED (round 4: inference with varying batch sizes
    grew it for ever) except for entries a captured graph may hold the address of (pinned), which are never evicted"""
    from open3dsot_amd import sa_modules
    dev = torch.device("cuda", 0)
    a = sa_modules._prefix_idx(3, 16, dev)
    assert sa_modules._prefix_idx(3, 16, dev).data_ptr() == a.data_ptr()          # shared
    pinned = sa_modules._prefix_idx(5, 24, dev)
    sa_modules._PREFIX_PINNED.add((5, 24, str(dev)))                               # what a capture that reads it does
    try:
        for i in range(3 * sa_modules._PREFIX_MAX):
            sa_modules._prefix_idx(2, 100 + i, dev)
        assert len(sa_modules._PREFIX_IDX) <= sa_modules._PREFIX_MAX + len(sa_modules._PREFIX_PINNED)
        assert sa_modules._prefix_idx(5, 24, dev).data_ptr() == pinned.data_ptr()
        b = sa_modules._prefix_idx(3, 16, dev)                                     # evicted and rebuilt: same content
        assert torch.equal(a, b)
    finally:
        sa_modules._PREFIX_PINNED.discard((5, 24, str(dev)))
    grouper, mlp, xyz, new_xyz, feats = make_case("sa2")
    mod = sa_modules.PointnetSAModule(mlp=[128, 128, 128, 256], radius=0.5, nsample=32).cuda()
    _, _, idx = mod(xyz, feats, 128, True)
    want = torch.arange(128, dtype=torch.int32, device=dev).repeat(xyz.shape[0], 1)
    assert torch.equal(idx, want) and torch.equal(sa_modules._prefix_idx(xyz.shape[0], 128, dev), want)


def test_fused_bwd_with_zero_gamma():
    """a zero BatchNorm scale in the PRODUCER layer (zero-initialised / pruned channel): the fused data + weight gradient
    kernel cannot recover sum g*(y - mean) from relu(bn(y)) = const there and re-reads the row instead
    (csrc/mlp_wgrad.hip) -- dgamma of that channel must equal the fp64 value, as with the unfused kernel pair"""
    import copy
    from open3dsot_amd import fused
    grouper, mlp, xyz, new_xyz, feats = make_case("sa1", B=4, train=True)
    bns = [m for m in mlp.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    with torch.no_grad():
        bns[0].weight[3] = 0.0            # beta > 0: the channel passes the ReLU as a constant
        bns[0].bias[3] = 0.4
        bns[0].weight[7] = 0.0            # beta < 0: masked everywhere
        bns[0].bias[7] = -0.4
    idx = grouper.query(xyz, new_xyz)
    o64, l64, _ = shadow64(copy.deepcopy(mlp), xyz, new_xyz, feats, idx, True)
    out = fused.sa_group_mlp_pool(grouper, mlp, xyz, new_xyz, feats)
    go = torch.randn(out.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    out.backward(go)
    o64.backward(go.double())
    for n1, p1 in mlp.named_parameters():
        assert_grad_close(p1.grad, l64[n1].grad, n1)
    name0 = [n for n, _ in mlp.named_parameters() if n.endswith("bn.bn.weight")][0]
    g, g64 = dict(mlp.named_parameters())[name0].grad, l64[name0].grad
    for c in (3, 7):
        assert abs(float(g[c]) - float(g64[c])) <= 1e-4 * float(g64.abs().max()), (c, float(g[c]), float(g64[c]))
    assert float(g64[3].abs()) > 1e-3 * float(g64.abs().max())       # the case is not vacuous


def test_paired_backbone_matches_sequential():
    """Pointnet_Backbone.forward_pair = the two backbone calls of the trackers, outputs and sampling indices"""
    import copy
    from open3dsot_amd import backbone, sa_modules, synth
    torch.manual_seed(0)
    net = backbone.Pointnet_Backbone(use_fps=True, normalize_xyz=False).cuda().train()
    ref = copy.deepcopy(net)
    b = synth.to_torch(synth.make_batch(77, 4, 512, 1024), torch.device("cuda"))
    t, s = b["template_points"], b["search_points"]
    ra, rb = net.forward_pair(t, [256, 128, 64], s, [512, 256, 128])
    sa_modules.set_paired(False)
    try:
        qa, qb = ref.forward_pair(t, [256, 128, 64], s, [512, 256, 128])
    finally:
        sa_modules.set_paired(True)
    for x, y in zip(ra + rb, qa + qb):
        if x.dtype.is_floating_point:
            assert rel(x, y) < 1e-4, rel(x, y)
        else:
            assert torch.equal(x, y)
    for (n1, b1), (_, b2) in zip(net.named_buffers(), ref.named_buffers()):
        assert rel(b1.float(), b2.float()) < 1e-5, n1


@pytest.mark.parametrize("use_fps", [True, False])
@pytest.mark.parametrize("train", [True, False])
def test_sample_query_launch_matches_gather_ball_query_cat(use_fps, train):
    """fused.sa_pair_sampled (round 5: sampling gather + both ball queries + the centres' layout in ONE launch,
    csrc/index_ops.hip::sample_query_kernel) against the route it replaces (gather_xyz / prefix copy, two ball queries, a
    concatenation): sampling indices, centres and every pooled feature of the paired backbone BIT-identical (the same
    kernels downstream on the same indices), running statistics and parameter gradients equal to run-to-run rounding"""
    import copy
    from open3dsot_amd import backbone, fused, synth
    torch.manual_seed(0)
    net = backbone.Pointnet_Backbone(use_fps=use_fps, normalize_xyz=False).cuda().train(train)
    ref = copy.deepcopy(net)
    b = synth.to_torch(synth.make_batch(77, 6, 512, 1024), torch.device("cuda"))
    t, s = b["template_points"], b["search_points"]
    with torch.set_grad_enabled(train):
        ra, rb = net.forward_pair(t, [256, 128, 64], s, [512, 256, 128])
        fused.set_sample_query(False)
        try:
            qa, qb = ref.forward_pair(t, [256, 128, 64], s, [512, 256, 128])
        finally:
            fused.set_sample_query(True)
    for x, y in zip(ra + rb, qa + qb):
        assert x.shape == y.shape and x.dtype == y.dtype
        assert torch.equal(x, y), float((x.float() - y.float()).abs().max())
    for (n1, b1), (_, b2) in zip(net.named_buffers(), ref.named_buffers()):
        assert rel(b1.float(), b2.float()) < 1e-6, n1
    if train:
        gen = torch.Generator(device="cuda").manual_seed(5)
        go = [torch.randn(x.shape, device="cuda", generator=gen) for x in (ra[1], rb[1])]
        torch.autograd.backward([ra[1], rb[1]], go)
        torch.autograd.backward([qa[1], qb[1]], go)
        for (n1, p1), (_, p2) in zip(net.named_parameters(), ref.named_parameters()):
            assert l2rel(p1.grad, p2.grad) < 1e-5, n1


@pytest.mark.parametrize("kind", ["sa1", "sa2", "sa3", "rpn"])
def test_reduce_gather_matches_atomic_reduce(kind):
    """o3d_group_reduce_gather (transposed index + LDS gather) against o3d_group_reduce_c (LDS atomics): same S / T,
    i.e. the same input and layer-0 weight gradients"""
    import copy
    from open3dsot_amd import fused
    grouper, mlp, xyz, new_xyz, feats = make_case(kind)
    grads = []
    was = fused.reduce_gather_enabled()
    for gather in (False, True):
        fused.set_reduce_gather(gather)
        try:
            m = copy.deepcopy(mlp)
            leaves = [t.clone().requires_grad_(True) if t is not None else None for t in (xyz, new_xyz, feats)]
            out = fused.sa_group_mlp_pool(grouper, m, *leaves)
            go = torch.randn(out.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
            out.backward(go)
            grads.append([p.grad for p in m.parameters()] + [t.grad for t in leaves if t is not None])
        finally:
            fused.set_reduce_gather(was)
    for a, b in zip(*grads):
        assert l2rel(a, b) < 1e-5, l2rel(a, b)


@pytest.mark.parametrize("kind,B,full", [("sa1", 3, False), ("sa2", 3, False), ("sa3", 3, False), ("sa1", 48, True),
                                         ("sa2", 48, True)])
def test_reduce_gather_matches_atomic_reduce_paired(kind, B, full):
    """the same comparison through the two-segment (template + search) call, also at BASELINE config 2's sizes"""
    import copy
    from open3dsot_amd import fused
    grouper, mlp, xyz_s, new_s, feats_s = make_case(kind, B=B, full=full)
    N, npoint = xyz_s.shape[1], new_s.shape[1]
    xyz_t = (xyz_s[:, :N // 2, :] * 0.9 + 0.05).contiguous()
    new_t = xyz_t[:, :npoint // 2, :].contiguous()
    feats_t = torch.randn(feats_s.shape[0], feats_s.shape[1], N // 2, device="cuda") if feats_s is not None else None
    grads = []
    was = fused.reduce_gather_enabled()
    for gather in (False, True):
        fused.set_reduce_gather(gather)
        try:
            m = copy.deepcopy(mlp)
            segs = [[t.clone().requires_grad_(True) if t is not None else None for t in sg]
                    for sg in ((xyz_t, new_t, feats_t), (xyz_s, new_s, feats_s))]
            outs = fused.sa_group_mlp_pool_pair(grouper, m, tuple(segs[0]), tuple(segs[1]))
            gen = torch.Generator(device="cuda").manual_seed(4)
            torch.autograd.backward(list(outs), [torch.randn(o.shape, device="cuda", generator=gen) for o in outs])
            grads.append([p.grad for p in m.parameters()] + [t.grad for sg in segs for t in sg if t is not None])
        finally:
            fused.set_reduce_gather(was)
    for a, b in zip(*grads):      # two fp32 summation orders of the same terms: grows with the number of terms
        assert l2rel(a, b) < (1e-5 if B <= 4 else 5e-5), l2rel(a, b)


@pytest.mark.parametrize("kind", ["sa1", "sa2", "sa3", "rpn"])
def test_pool_bwd_one_pass_matches_zero_fill_and_scatter(kind):
    """o3d_pool_bwd_dense (round 5: the pooled layer's dense gradient written once, column by column, statistics rows per
    512 columns) against o3d_pool_bwd_c (zero fill of the live columns + scatter, 8 statistics rows per segment): the same
    dense gradient, so every parameter / input gradient agrees to the order of two fp32 sums of the statistics"""
    import copy
    from open3dsot_amd import fused
    grouper, mlp, xyz, new_xyz, feats = make_case(kind)
    grads = []
    for dense in (False, True):
        fused.set_pool_bwd_dense(dense)
        try:
            m = copy.deepcopy(mlp)
            leaves = [t.clone().requires_grad_(True) if t is not None else None for t in (xyz, new_xyz, feats)]
            out = fused.sa_group_mlp_pool(grouper, m, *leaves)
            go = torch.randn(out.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
            out.backward(go)
            grads.append([p.grad for p in m.parameters()] + [t.grad for t in leaves if t is not None])
        finally:
            fused.set_pool_bwd_dense(True)
    for a, b in zip(*grads):
        assert l2rel(a, b) < 1e-5, l2rel(a, b)


@pytest.mark.parametrize("kind,B,full", [("sa1", 3, False), ("sa2", 3, False), ("sa3", 3, False), ("sa1", 48, True),
                                         ("sa2", 48, True), ("sa3", 48, True)])
def test_pool_bwd_one_pass_matches_zero_fill_and_scatter_paired(kind, B, full):
    """the same comparison through the two-segment (template + search) call, also at BASELINE config 2's sizes (segment 1
    at a non-zero `start1`, chunks whose last 256 columns are dead, padding columns of the dummy ball)"""
    import copy
    from open3dsot_amd import fused
    grouper, mlp, xyz_s, new_s, feats_s = make_case(kind, B=B, full=full)
    N, npoint = xyz_s.shape[1], new_s.shape[1]
    xyz_t = (xyz_s[:, :N // 2, :] * 0.9 + 0.05).contiguous()
    new_t = xyz_t[:, :npoint // 2, :].contiguous()
    feats_t = torch.randn(feats_s.shape[0], feats_s.shape[1], N // 2, device="cuda") if feats_s is not None else None
    grads = []
    for dense in (False, True):
        fused.set_pool_bwd_dense(dense)
        try:
            m = copy.deepcopy(mlp)
            segs = [[t.clone().requires_grad_(True) if t is not None else None for t in sg]
                    for sg in ((xyz_t, new_t, feats_t), (xyz_s, new_s, feats_s))]
            outs = fused.sa_group_mlp_pool_pair(grouper, m, tuple(segs[0]), tuple(segs[1]))
            gen = torch.Generator(device="cuda").manual_seed(4)
            torch.autograd.backward(list(outs), [torch.randn(o.shape, device="cuda", generator=gen) for o in outs])
            grads.append([p.grad for p in m.parameters()] + [t.grad for sg in segs for t in sg if t is not None])
        finally:
            fused.set_pool_bwd_dense(True)
    for a, b in zip(*grads):
        assert l2rel(a, b) < (1e-5 if B <= 4 else 5e-5), l2rel(a, b)


@pytest.mark.parametrize("slots", [8, 16, 24, 40, 64])
@pytest.mark.parametrize("kind,B", [("sa2", 8), ("sa3", 12)])
def test_remainder_tiles_in_column_blocks_match_the_plain_launch(kind, B, slots):
    """Round 5: the last T mod S live tiles of a 128-column GEMM launch cut into 2 or 4 column blocks that run as workgroups
    of the same launch (csrc/mlp_direct.hip::direct_gemm_tail_kernel, csrc/mlp_common.hpp::tail_plan; their statistics in
    extra rows the finalize kernels follow).  With the slot count forced small, problems of a few dozen tiles meet every
    branch of the plan (no remainder, quarter blocks, half blocks, too large a remainder, both segments of a paired call):
    pooled features, running statistics and every gradient against the unsplit launch -- the same MFMA chains on the same
    operands, only the statistics are summed in another order."""
    import copy
    from open3dsot_amd import fused
    grouper, mlp, xyz_s, new_s, feats_s = make_case(kind, B=B, full=True)
    N, npoint = xyz_s.shape[1], new_s.shape[1]
    xyz_t = (xyz_s[:, :N // 2, :] * 0.9 + 0.05).contiguous()
    new_t = xyz_t[:, :npoint // 2, :].contiguous()
    feats_t = torch.randn(feats_s.shape[0], feats_s.shape[1], N // 2, device="cuda") if feats_s is not None else None
    res = []
    for s in (0, slots):
        fused.set_tail_split(s)
        try:
            m = copy.deepcopy(mlp)
            segs = [[t.clone().requires_grad_(True) if t is not None else None for t in sg]
                    for sg in ((xyz_t, new_t, feats_t), (xyz_s, new_s, feats_s))]
            outs = fused.sa_group_mlp_pool_pair(grouper, m, tuple(segs[0]), tuple(segs[1]))
            gen = torch.Generator(device="cuda").manual_seed(4)
            torch.autograd.backward(list(outs), [torch.randn(o.shape, device="cuda", generator=gen) for o in outs])
            res.append(([o.detach() for o in outs] + [b.clone() for b in m.buffers() if b.dtype.is_floating_point],
                        [p.grad for p in m.parameters()] + [t.grad for sg in segs for t in sg if t is not None]))
        finally:
            fused.set_tail_split(-1)
    for a, b in zip(res[0][0], res[1][0]):
        assert rel(a, b) < 2e-6, rel(a, b)
    for a, b in zip(res[0][1], res[1][1]):
        assert l2rel(a, b) < 2e-5, l2rel(a, b)


def test_shapes_outside_the_compact_layout_run_operator_by_operator():
    """what the distinct-neighbour layout does not take (here nsample = 128 > 64; likewise channel counts that are no
    multiple of 64) runs the reference's own sequence -- ball query, grouping_operation, SharedMLP, max-pool -- on the
    library's index operators and torch's convolutions (fused._composed): still correct against fp64, and NOT on a
    fused kernel (round 4 retired the slot-per-neighbour fused path that used to sit in between)"""
    import copy
    from open3dsot_amd import fused, ops
    grouper, mlp, xyz, new_xyz, feats = make_case("sa2", train=True)
    grouper = ops.QueryAndGroup(0.5, 128, use_xyz=True)
    mlp_ref = copy.deepcopy(mlp)
    f = feats.clone().requires_grad_(True)
    idx = grouper.query(xyz, new_xyz)
    ref64, l64, b64 = shadow64(mlp_ref, xyz, new_xyz, feats, idx, True)
    out = fused.sa_group_mlp_pool(grouper, mlp, xyz, new_xyz, f)
    assert "FusedGroupedMLP" not in out.grad_fn.name()
    assert rel(out, ref64) < 2e-5
    go = torch.randn(out.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    out.backward(go)
    ref64.backward(go.double())
    for n1, p1 in mlp.named_parameters():
        assert_grad_close(p1.grad, l64[n1].grad, n1)
    assert_grad_close(f.grad, l64["feats"].grad, "feats")


@pytest.mark.parametrize("train", [True, False])
def test_fused_xcorr_group_mlp_pool(train):
    from open3dsot_amd import fused, nn_blocks, ops
    torch.manual_seed(4)
    B, M, N, k, f = 3, 64, 128, 4, 256
    bundle = torch.randn(B, 3 + 9 + f, M, device="cuda", requires_grad=True)
    idx = torch.randint(0, M, (B, N, k), device="cuda", dtype=torch.int32)
    mlp = nn_blocks.SharedMLP([3 + 9 + f, 256, 256, 256], bn=True).cuda().train(train)
    ref64, l64, _ = shadow64(mlp, None, None, bundle, idx, train)
    with torch.no_grad():
        ref32 = copy_eval(mlp, ops.grouping_operation(bundle.detach(), idx), train)
    out = fused.group_mlp_pool(mlp, bundle, idx)
    assert rel(out, ref32) < 1e-4
    assert rel(out, ref64) < 2e-5
    go = torch.randn_like(out)
    out.backward(go)
    ref64.backward(go.double())
    assert_grad_close(bundle.grad, l64["feats"].grad, "bundle")
    for n1, p1 in mlp.named_parameters():
        assert_grad_close(p1.grad, l64[n1].grad, n1)


def copy_eval(mlp, x, train):
    """torch fp32 forward of a deep copy (so the running statistics of `mlp` are not advanced twice)"""
    import copy
    m = copy.deepcopy(mlp).train(train)
    return m(x).max(dim=-1)[0]


def test_fused_full_size_layer_properties():
    """BASELINE config-2 size (B=48, search SA1: 512 centres x 32 neighbours): properties that do
    not need a second implementation -- pooled output >= 0, BN statistics consistent with the
    stored raw output, linearity of the weight gradient in dOut."""
    from open3dsot_amd import fused
    grouper, mlp, xyz, new_xyz, feats = make_case("sa1", B=48)
    out = fused.sa_group_mlp_pool(grouper, mlp, xyz, new_xyz, feats)
    assert out.shape == (48, 128, 256) and bool((out >= 0).all()) and bool(torch.isfinite(out).all())
    g1 = torch.randn_like(out)
    out.backward(g1, retain_graph=True)
    w = [p.grad.clone() for p in mlp.parameters()]
    for p in mlp.parameters():
        p.grad = None
    out.backward(2 * g1)
    for a, p in zip(w, mlp.parameters()):
        assert rel(p.grad, 2 * a) < 1e-5


@pytest.mark.parametrize("ns", [4, 16, 32])
def test_compact_build(lib, ns):
    """csrc/compact.hip: one column per distinct neighbour, first hit weighted by its copies"""
    g = torch.Generator().manual_seed(ns)
    B, npoint, N, ld = 4, 64, 100, 128
    idx = torch.zeros(B, npoint, ns, dtype=torch.int32)
    ref_cnt = []
    for b in range(B):
        for j in range(npoint):
            cnt = int(torch.randint(1, ns + 1, (1,), generator=g))
            hits = torch.randperm(N, generator=g)[:cnt].sort()[0].int()
            idx[b, j, :cnt] = hits
            idx[b, j, cnt:] = hits[0]
            ref_cnt.append(cnt)
    dev_idx = idx.cuda()
    nballs, Pmax = B * npoint, B * npoint * ns
    i32 = dict(device="cuda", dtype=torch.int32)
    ball_cnt, ball_off = torch.empty(nballs, **i32), torch.empty(nballs + 1, **i32)
    gp, cball, meta = torch.empty(Pmax, **i32), torch.empty(Pmax, **i32), torch.empty(4, **i32)
    cw = torch.empty(Pmax, device="cuda")
    # one segment: columns, points and balls all based at 0; padding columns point at ball `nballs`
    assert lib.o3d_compact_build(dev_idx.data_ptr(), B, npoint, ns, ld, 0, 0, 0, nballs, ball_cnt.data_ptr(),
                                 ball_off.data_ptr(), gp.data_ptr(), cball.data_ptr(), cw.data_ptr(), meta.data_ptr(),
                                 st()) == 0
    ref_cnt = torch.tensor(ref_cnt, dtype=torch.int32)
    assert torch.equal(ball_cnt.cpu(), ref_cnt)
    off = torch.cat([torch.zeros(1, dtype=torch.int64), ref_cnt.long().cumsum(0)])
    assert torch.equal(ball_off.cpu().long()[:nballs], off[:nballs])      # one offset per ball (ends = off + cnt)
    tot = int(off[-1])
    assert meta.cpu().tolist()[:3] == [(tot + 255) // 256 * 256, tot, nballs]
    gp_c, cball_c, cw_c = gp.cpu(), cball.cpu(), cw.cpu()
    for ball in (0, 7, nballs - 1):
        b, j, q0, cnt = ball // npoint, ball % npoint, int(off[ball]), int(ref_cnt[ball])
        assert gp_c[q0:q0 + cnt].tolist() == (idx[b, j, :cnt].long() + b * ld).tolist()
        assert cball_c[q0:q0 + cnt].tolist() == [ball] * cnt
        assert cw_c[q0:q0 + cnt].tolist() == [float(1 + ns - cnt)] + [1.0] * (cnt - 1)
    pad = slice(tot, int(meta[0]))
    assert bool((cw_c[pad] == 0).all()) and bool((cball_c[pad] == nballs).all())
    assert abs(float(cw_c[:tot].sum()) - Pmax) < 0.5      # the weights account for every slot


@pytest.mark.parametrize("ns", [16, 32])
def test_compact_build2_equals_two_single_builds(lib, ns):
    """o3d_compact_build2 (both segments of a paired call in three launches) writes exactly what two o3d_compact_build
    calls write: counts, offsets, per-column source point / ball / weight, the live counts of both segments"""
    g = torch.Generator().manual_seed(100 + ns)
    B, np0, np1, N0, N1 = 4, 32, 64, 100, 200
    ld0, ld1 = 128, 256

    def make(npoint, N):
        idx = torch.zeros(B, npoint, ns, dtype=torch.int32)
        for b in range(B):
            for j in range(npoint):
                cnt = int(torch.randint(1, ns + 1, (1,), generator=g))
                hits = torch.randperm(N, generator=g)[:cnt].sort()[0].int()
                idx[b, j, :cnt] = hits
                idx[b, j, cnt:] = hits[0]
        return idx.cuda()
    i0, i1 = make(np0, N0), make(np1, N1)
    nb0, nb1 = B * np0, B * np1
    nballs, P0, P1 = nb0 + nb1, nb0 * ns, nb1 * ns
    i32 = dict(device="cuda", dtype=torch.int32)

    def bufs():
        return (torch.full((nballs,), -1, **i32), torch.full((nballs + 1,), -1, **i32), torch.full((P0 + P1,), -1, **i32),
                torch.full((P0 + P1,), -1, **i32), torch.full((P0 + P1,), -1.0, device="cuda"), torch.full((2, 4), -1, **i32))
    a = bufs()
    assert lib.o3d_compact_build(i0.data_ptr(), B, np0, ns, ld0, 0, 0, 0, nballs, a[0].data_ptr(), a[1].data_ptr(),
                                 a[2].data_ptr(), a[3].data_ptr(), a[4].data_ptr(), a[5][0].data_ptr(), st()) == 0
    assert lib.o3d_compact_build(i1.data_ptr(), B, np1, ns, ld1, P0, B * ld0, nb0, nballs, a[0][nb0:].data_ptr(),
                                 a[1][nb0:].data_ptr(), a[2].data_ptr(), a[3].data_ptr(), a[4].data_ptr(),
                                 a[5][1].data_ptr(), st()) == 0
    b = bufs()
    assert lib.o3d_compact_build2(i0.data_ptr(), np0, ld0, i1.data_ptr(), np1, ld1, B, ns, P0, B * ld0, nballs,
                                  b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), b[3].data_ptr(), b[4].data_ptr(),
                                  b[5].data_ptr(), st()) == 0
    torch.cuda.synchronize()
    for x, y, nm in zip(a, b, ("ball_cnt", "ball_off", "gp", "cball", "cw", "meta")):
        assert torch.equal(x, y), nm
    assert int(b[5][1, 1]) > 0 and int(b[5][0, 1]) > 0


def test_in_place_weight_update_between_forward_and_backward_is_refused():
    """the fused functions keep views of the live parameter storage for backward: an in-place update in between must
    raise (as autograd's version counter does for saved tensors) instead of yielding wrong gradients"""
    from open3dsot_amd import fused
    grouper, mlp, xyz, new_xyz, feats = make_case("sa2")
    out = fused.sa_group_mlp_pool(grouper, mlp, xyz, new_xyz, feats.clone().requires_grad_(True))
    with torch.no_grad():
        next(mlp.parameters()).mul_(1.5)
    with pytest.raises(RuntimeError, match="modified in place"):
        out.sum().backward()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the round-end scaling node has them)")
def test_fused_path_on_a_non_current_device():
    """a module on cuda:1 while cuda:0 is the current device: streams, scratch and launches follow the tensors"""
    from open3dsot_amd import fused
    grouper, mlp, xyz, new_xyz, feats = make_case("sa2")
    dev1 = torch.device("cuda", 1)
    mlp1 = __import__("copy").deepcopy(mlp).to(dev1)
    want = fused.sa_group_mlp_pool(grouper, mlp, xyz, new_xyz, feats)
    assert torch.cuda.current_device() == 0
    got = fused.sa_group_mlp_pool(grouper, mlp1, xyz.to(dev1), new_xyz.to(dev1), feats.to(dev1))
    assert got.device == dev1 and rel(got, want) < 1e-6


@pytest.mark.parametrize("kind", ["sa1", "sa2", "sa3", "rpn"])
def test_eval_fused_kernel_matches_layerwise_path(kind):
    """tracking inference (eval mode, no autograd): the one-kernel set abstraction of csrc/sa_eval.hip against the
    layer-by-layer kernels and the fp64 shadow; single call and the template + search pair"""
    from open3dsot_amd import fused
    grouper, mlp, xyz, new_xyz, feats = make_case(kind, train=False)
    idx = grouper.query(xyz, new_xyz)
    ref64, _, _ = shadow64(mlp, xyz, new_xyz, feats, idx, False)
    N, npoint = xyz.shape[1], new_xyz.shape[1]
    xyz_t = (xyz[:, :N // 2, :] * 0.9 + 0.05).contiguous()
    new_t = xyz_t[:, :npoint // 2, :].contiguous()
    feats_t = torch.randn(feats.shape[0], feats.shape[1], N // 2, device="cuda") if feats is not None else None
    res = {}
    was = fused._EVAL_FUSED["on"]
    try:
        for on in (False, True):
            fused.set_eval_fused(on)
            with torch.no_grad():
                one = fused.sa_group_mlp_pool(grouper, mlp, xyz, new_xyz, feats)
                pair = fused.sa_group_mlp_pool_pair(grouper, mlp, (xyz_t, new_t, feats_t), (xyz, new_xyz, feats))
            res[on] = (one, pair)
    finally:
        fused.set_eval_fused(was)
    assert rel(res[True][0], ref64) < 2e-5, rel(res[True][0], ref64)
    assert rel(res[True][0], res[False][0]) < 1e-5
    if res[False][1] is not None:
        assert res[True][1] is not None
        for a, b in zip(res[True][1], res[False][1]):
            assert a.shape == b.shape and rel(a, b) < 1e-5, rel(a, b)
        assert rel(res[True][1][1], ref64) < 2e-5


def test_eval_fused_kernel_knn_groups():
    """the BoxCloud xcorr grouping (k-NN lists of 4, no xyz channels: models/head/xcorr.py:89-100) through the same kernel"""
    from open3dsot_amd import fused, nn_blocks, ops
    torch.manual_seed(3)
    B, M, N, k = 4, 64, 128, 4
    bundle = torch.randn(B, 268, M, device="cuda")
    idx = ops.knn_point(k, torch.rand(B, N, 9, device="cuda"), torch.rand(B, M, 9, device="cuda"))
    mlp = nn_blocks.SharedMLP([268, 256, 256, 256], bn=True).cuda().eval()
    with torch.no_grad():
        for m in mlp.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2)
        want, _, _ = shadow64(mlp, None, None, bundle, idx, False)
        got = fused.group_mlp_pool(mlp, bundle, idx)
        fused.set_eval_fused(False)
        try:
            layerwise = fused.group_mlp_pool(mlp, bundle, idx)
        finally:
            fused.set_eval_fused(True)
    assert rel(got, want) < 2e-5 and rel(got, layerwise) < 1e-5
