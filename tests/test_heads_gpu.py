"""GPU parity of the flat-layout 1-D conv stacks (open3dsot_amd/fused_heads.py on csrc/mlp_direct.hip,
csrc/mlp_wgrad.hip, csrc/heads.hip) against the same pt_utils.Seq modules evaluated by torch in fp64
(pointnet2/utils/pytorch_utils.py:124-155,300-457; the stacks of models/head/rpn.py:16-39, models/head/xcorr.py:14-17,
models/bat.py:22-26).  Forward 1e-4 of the tensor scale (north_star; measured ~1e-6), gradients 5e-4 L2 / 1e-2 max."""
import copy
import os
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-20))


def l2rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def test_pack_rows_matches_cat():
    from open3dsot_amd import fused_heads
    g = torch.Generator(device="cuda").manual_seed(0)
    B, N = 3, 40
    xyz = torch.randn(B, N, 3, device="cuda", generator=g)
    flat = torch.randn(7, B, N, device="cuda", generator=g)              # a flat (C,B,N) buffer seen as (B,C,N)
    wide = torch.randn(B, 12, 2 * N, device="cuda", generator=g)[:, 2:7, ::2]      # strided slice
    parts = [xyz.transpose(1, 2), flat.permute(1, 0, 2), wide]
    X = fused_heads.pack_rows(parts, 64)
    want = torch.cat([p.contiguous() for p in parts], dim=1).permute(1, 0, 2).reshape(15, B * N)
    assert torch.equal(X[:15], want) and float(X[15:].abs().max()) == 0.0


def test_prep_weights_pads_and_transposes():
    from open3dsot_amd import fused_heads
    dev = torch.device("cuda", 0)
    prep = fused_heads.WeightPrep(dev)
    w = torch.nn.Parameter(torch.randn(9, 259, 1, device=dev))
    b = torch.nn.Parameter(torch.randn(9, device=dev))
    wp, wt, bp = prep.get(w, 64, 320), prep.get(w, 320, 64, transpose=True), prep.get(b, 1, 64)

    def check():
        assert torch.equal(wp[:9, :259], w.detach()[:, :, 0]) and float(wp[9:].abs().max()) == 0 and float(wp[:, 259:].abs().max()) == 0
        assert torch.equal(wt[:259, :9], w.detach()[:, :, 0].t()) and float(wt[259:].abs().max()) == 0 and float(wt[:, 9:].abs().max()) == 0
        assert torch.equal(bp[0, :9], b.detach()) and float(bp[0, 9:].abs().max()) == 0
    check()
    with torch.no_grad():
        w.mul_(2.0).add_(1.0)
        b.add_(3.0)
    prep.refresh()              # one launch for the three jobs
    check()
    same = torch.nn.Parameter(torch.randn(256, 256, 1, device=dev))
    assert prep.get(same, 256, 256).data_ptr() == same.data_ptr()        # already aligned: handed out as is


CASES = {
    # name: (channels of the sources, widths, residual)
    "cla": ([256], [256, 256, 1], False),                 # FC_layer_cla        rpn.py:16-22
    "vote": ([3, 256], [256, 256, 259], True),            # vote_layer + seeds  rpn.py:23-28,50-54
    "proposal": ([256], [256, 256, 5], False),            # FC_proposal         rpn.py:34-39
    "mlp_bc": ([3, 256], [256, 256, 9], False),           # mlp_bc              bat.py:23-26,94
    "fea": ([256], [256, 256], False),                    # fea_layer           xcorr.py:15-17
}


def build_seq(widths, cin, seed):
    from open3dsot_amd import nn_blocks
    torch.manual_seed(seed)
    seq = nn_blocks.Seq(cin)
    for i, w in enumerate(widths):
        seq = seq.conv1d(w, bn=True) if i < len(widths) - 1 else seq.conv1d(w, activation=None)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in seq.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(torch.empty(m.weight.shape).uniform_(0.5, 1.5, generator=g))
                m.bias.copy_(torch.empty(m.bias.shape).normal_(0, 0.2, generator=g))
                m.running_mean.copy_(torch.empty(m.bias.shape).normal_(0, 0.2, generator=g))
                m.running_var.copy_(torch.empty(m.bias.shape).uniform_(0.5, 1.5, generator=g))
            if isinstance(m, torch.nn.Conv1d) and m.bias is not None:
                m.bias.copy_(torch.empty(m.bias.shape).normal_(0, 0.3, generator=g))
    return seq.cuda()


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("B,N", [(4, 64), (48, 128)])
def test_flat_chain_vs_fp64(name, train, B, N):
    from open3dsot_amd import fused_heads, nn_blocks
    src_C, widths, residual = CASES[name]
    seq = build_seq(widths, sum(src_C), 3).train(train)
    seq0 = copy.deepcopy(seq)
    ref = copy.deepcopy(seq).double()
    zs = []       # per hidden layer: (B, N) margin of the fp64 BatchNorm output to zero, in fp32 ulps of the layer's rms
    if B * N > 1024:
        import flip_proof
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                # (the fp64 module runs the flat torch path: BatchNorm1d sees (1, C, B*N), column b*N + n)
                m.register_forward_hook(lambda mod, inp, outp: zs.append(
                    flip_proof.relu_margin_ulps(outp.detach(), 1).reshape(B, N)))
    g = torch.Generator(device="cuda").manual_seed(5)
    parts = []
    for C in src_C:     # xyz arrives as a transposed (B,N,3) tensor, features as (B,C,N)
        t = torch.randn(B, N, C, device="cuda", generator=g).transpose(1, 2) if C == 3 else \
            torch.randn(B, C, N, device="cuda", generator=g)
        parts.append(t.requires_grad_(True))
    parts64 = [t.detach().double().requires_grad_(True) for t in parts]
    assert fused_heads.chain_supported(parts, seq._flat_units())
    out = nn_blocks.seq_apply(seq, parts, residual)
    x64 = torch.cat(parts64, dim=1)
    ref_out = ref(x64) + (x64 if residual else 0)
    assert out.shape == ref_out.shape
    assert rel(out, ref_out) < 2e-5, rel(out, ref_out)
    ct = torch.randn(out.shape, device="cuda", generator=g)
    big = B * N > 1024
    (out * ct).sum().backward()
    (ref_out * ct.double()).sum().backward(retain_graph=big)
    # ReLU masks are discrete: a pre-activation within fp32 rounding of zero is routed differently than in fp64, which
    # changes that ONE column's gradient by O(1) and every parameter gradient by ~1/sqrt(columns).  At the small size no
    # unit is that close and the tight bounds apply directly.  At the benchmarked size (48 x 128 columns) the test PROVES
    # the reading instead of tolerating it (tests/flip_proof.py): stage 1 -- every column whose input gradient is off by
    # more than 1e-4 of the rms has an fp64 pre-activation within TIE_ULPS fp32 ulps of zero; stage 2 -- with the
    # cotangent zeroed on the flagged columns in both evaluations, every gradient meets the tight bounds.
    if not big:
        for a, b in zip(parts, parts64):
            assert l2rel(a.grad, b.grad) < 5e-4 and rel(a.grad, b.grad) < 1e-2, ("input", l2rel(a.grad, b.grad))
        for (n1, p), (_, q) in zip(seq.named_parameters(), ref.named_parameters()):
            assert p.grad is not None and p.grad.shape == p.shape, n1
            assert l2rel(p.grad, q.grad) < 5e-4 and rel(p.grad, q.grad) < 1e-2, (n1, l2rel(p.grad, q.grad), rel(p.grad, q.grad))
    else:
        import flip_proof as fp
        tag = "flat chain %s %s (%d,%d)" % (name, "train" if train else "eval", B, N)
        margin = torch.stack(zs).amin(dim=0)                       # (B, N): min over the hidden layers' ReLU units, in ulps
        flagged = margin < fp.TIE_ULPS
        for i, (a, b) in enumerate(zip(parts, parts64)):
            outl, emap = fp.outlier_columns(a.grad, b.grad, (0, 2))
            fp.check_outliers_flagged("%s d(source %d)" % (tag, i), outl, emap, flagged, margin)
        for (n1, p), (_, q) in zip(seq.named_parameters(), ref.named_parameters()):     # loose: includes the flips
            assert p.grad is not None and p.grad.shape == p.shape, n1
            assert l2rel(p.grad, q.grad) < 3e-3, (n1, l2rel(p.grad, q.grad))
        seq2 = copy.deepcopy(seq0)
        parts2 = [t.detach().clone().requires_grad_(True) for t in parts]
        ct2 = ct * (~flagged)[:, None, :].to(ct.dtype)
        for t in parts64 + list(ref.parameters()):
            t.grad = None
        (nn_blocks.seq_apply(seq2, parts2, residual) * ct2).sum().backward()
        (ref_out * ct2.double()).sum().backward()
        pairs = [("source %d" % i, a.grad, b.grad) for i, (a, b) in enumerate(zip(parts2, parts64))]
        pairs += [(n1, p.grad, q.grad) for (n1, p), (_, q) in zip(seq2.named_parameters(), ref.named_parameters())]
        fp.check_tight(tag, pairs)
    if train:
        for (n1, b1), (_, b2) in zip(seq.named_buffers(), ref.named_buffers()):
            if b1.dtype.is_floating_point:
                assert rel(b1, b2) < 1e-5, n1
            else:
                assert int(b1) == int(b2) == 1, n1


@pytest.mark.parametrize("B,N", [(4, 64), (48, 64), (48, 128)])
def test_conv_final_flat(B, N):
    """nn.Conv1d(256, 256, 1) with bias (models/bat.py:22, :91-92) through the same kernels"""
    from open3dsot_amd import nn_blocks
    torch.manual_seed(1)
    conv = torch.nn.Conv1d(256, 256, 1).cuda()
    ref = copy.deepcopy(conv).double()
    x = torch.randn(B, 256, N, device="cuda", requires_grad=True)
    x64 = x.detach().double().requires_grad_(True)
    out = nn_blocks.pointwise_conv1d(conv, x)
    want = ref(x64)
    assert rel(out, want) < 2e-5
    ct = torch.randn(out.shape, device="cuda")
    (out * ct).sum().backward()
    (want * ct.double()).sum().backward()
    assert l2rel(x.grad, x64.grad) < 5e-4
    assert l2rel(conv.weight.grad, ref.weight.grad) < 5e-4 and l2rel(conv.bias.grad, ref.bias.grad) < 5e-4


def test_flat_chain_falls_back_when_unaligned():
    """B*N not a multiple of 64: the torch path (same numbers) runs instead"""
    from open3dsot_amd import fused_heads, nn_blocks
    seq = build_seq([256, 256, 5], 256, 2).train()
    x = torch.randn(3, 256, 30, device="cuda")
    assert not fused_heads.chain_supported([x], seq._flat_units())
    assert nn_blocks.seq_apply(seq, [x]).shape == (3, 5, 30)


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("B,M,N", [(2, 32, 64), (4, 64, 128)])
def test_p2b_xcorr_fused_vs_fp64(train, B, M, N):
    """P2B_XCorr's SharedMLP + max over the template axis (models/head/xcorr.py:37-49) on the split layer-0 kernels of
    csrc/xcorr.hip against the reference formulation -- the materialised (B,4+f,M,N) fusion tensor through the same
    SharedMLP module -- evaluated in fp64; then the whole module (fea_layer behind it) in forward."""
    from open3dsot_amd import fused_xcorr, sa_modules, xcorr
    torch.manual_seed(2)
    mod = xcorr.P2B_XCorr(256, 256, 256).cuda().train(train)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.weight.copy_(torch.empty(m.weight.shape).uniform_(0.5, 1.5, generator=g))
                m.bias.copy_(torch.empty(m.bias.shape).normal_(0, 0.2, generator=g))
                m.running_mean.copy_(torch.empty(m.bias.shape).normal_(0, 0.2, generator=g))
                m.running_var.copy_(torch.empty(m.bias.shape).uniform_(0.5, 1.5, generator=g))
    ref = copy.deepcopy(mod).double()
    gg = torch.Generator(device="cuda").manual_seed(6)
    t_feat = torch.randn(B, 256, M, device="cuda", generator=gg).requires_grad_(True)
    s_feat = torch.randn(B, 256, N, device="cuda", generator=gg).requires_grad_(True)
    t_xyz = torch.randn(B, M, 3, device="cuda", generator=gg).requires_grad_(True)
    leaves = (t_feat, s_feat, t_xyz)
    tf, sf, tx = [t.detach().double().requires_grad_(True) for t in leaves]
    assert fused_xcorr.supported(mod.mlp, t_feat, s_feat)
    out = fused_xcorr.p2b_xcorr_mlp_pool(mod.mlp, t_feat, s_feat, t_xyz)
    sim = torch.nn.functional.cosine_similarity(tf.unsqueeze(-1).expand(B, 256, M, N),
                                                sf.unsqueeze(2).expand(B, 256, M, N), dim=1)          # xcorr.py:37-38
    x = torch.cat((sim.unsqueeze(1), tx.transpose(1, 2).unsqueeze(-1).expand(B, 3, M, N),
                   tf.unsqueeze(-1).expand(B, 256, M, N)), dim=1)                                      # :40-45
    # fp64 near-ties (tests/flip_proof.py): a ReLU whose fp64 pre-activation, or a max over the template axis whose two best
    # candidates, lie within fp32 rounding of each other can route differently in ANY fp32 evaluation (round 4: the
    # similarity map moved from torch.bmm to a kernel with another summation order and this very test flipped one routing
    # decision -- a 2 % gradient change from a 1e-7 change of `sim`).  The search points holding such a unit are taken out of
    # the cotangent on both sides; they must be few, and everything else meets the tight bound.
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import flip_proof
    margins, acts = [], []
    hooks = [m.register_forward_hook(lambda _m, _i, o: acts.append(o.detach().clone()))      # (the ReLU behind it works in place)
             for m in ref.mlp.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    y = ref.mlp(x)                                                                                     # (B,C,M,N)
    for h in hooks:
        h.remove()
    for z in acts:              # BatchNorm outputs in front of the ReLUs: min over channels and template points -> (B,N)
        margins.append((z.abs().amin(dim=(1, 2)) / (flip_proof.ULP * flip_proof.rms(z))))
    margins.append(flip_proof.pool_margin_ulps(y.detach().permute(0, 1, 3, 2), 1))                    # the max over M
    margin = torch.stack(margins).amin(dim=0)                                                          # (B,N)
    # 16 ulps here (flip_proof's 64 is sized for balls of 32 slots): a search point carries 3 layers x 256 channels x M template
    # points = up to 49 152 units, and at 64 ulps a third of the search points would hold one
    flagged = margin < 16.0
    assert float(flagged.float().mean()) < 0.25, float(flagged.float().mean())
    want = y.max(dim=2)[0]                                                                             # :47-49
    assert out.shape == want.shape
    assert rel(out, want) < 2e-5, rel(out, want)
    ct = torch.randn(out.shape, device="cuda", generator=gg) * (~flagged)[:, None, :].to(out.dtype)
    (out * ct).sum().backward()
    (want * ct.double()).sum().backward()
    tol = 5e-4 if train else 3e-3        # eval: no normalisation damps a max-pool routing flip
    for a, b in zip(leaves, (tf, sf, tx)):
        assert l2rel(a.grad, b.grad) < tol and rel(a.grad, b.grad) < 2e-2, (l2rel(a.grad, b.grad), rel(a.grad, b.grad))
    for (n1, p), (_, q) in zip(mod.mlp.named_parameters(), ref.mlp.named_parameters()):
        assert p.grad is not None and p.grad.shape == p.shape, n1
        assert l2rel(p.grad, q.grad) < tol and rel(p.grad, q.grad) < 2e-2, (n1, l2rel(p.grad, q.grad), rel(p.grad, q.grad))
    if train:
        for (n1, b1), (_, b2) in zip(mod.mlp.named_buffers(), ref.mlp.named_buffers()):
            if b1.dtype.is_floating_point:
                assert rel(b1, b2) < 1e-5, n1
    # the whole module: fused stage + fea_layer (its training-mode BatchNorm over only B*N = 128..512 columns divides
    # by a batch deviation of nearly constant pooled features: rounding is amplified to ~5e-4)
    mod2 = copy.deepcopy(ref).float().train(train)
    sa_modules.set_fused(False)
    try:
        whole = ref(tf.detach(), sf.detach(), tx.detach())
    finally:
        sa_modules.set_fused(True)
    got = mod2(t_feat.detach(), s_feat.detach(), t_xyz.detach())
    assert rel(got, whole) < 2e-3, rel(got, whole)


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("B,N", [(4, 64), (48, 128)])
def test_chain_pair_equals_two_single_chains(train, B, N):
    """FC_layer_cla and vote_layer (models/head/rpn.py:44-54) advanced side by side in merged launches
    (fused_heads.run_chain_pair: splitk_gemm_pair_kernel / direct_gemm_pair_kernel, bn_*finalize_pair_kernel) against the two
    stacks run one after the other (two `seq_apply` calls): the same kernels' arithmetic on the same operands, so outputs, every gradient and the running
    statistics must be bitwise equal"""
    from open3dsot_amd import fused_heads, nn_blocks
    cla = build_seq(CASES["cla"][1], 256, 11).train(train)
    vote = build_seq(CASES["vote"][1], 259, 12).train(train)
    cla2, vote2 = copy.deepcopy(cla), copy.deepcopy(vote)
    g = torch.Generator(device="cuda").manual_seed(8)
    feat = torch.randn(B, 256, N, device="cuda", generator=g)
    xyz = torch.randn(B, N, 3, device="cuda", generator=g)
    f1, x1 = feat.clone().requires_grad_(True), xyz.clone().requires_grad_(True)
    f2, x2 = feat.clone().requires_grad_(True), xyz.clone().requires_grad_(True)
    oa, ob = nn_blocks.seq_apply_pair((cla, [f1], False), (vote, [x1.transpose(1, 2), f1], True))
    assert "FlatChainPair" in oa.grad_fn.name()
    ra = nn_blocks.seq_apply(cla2, [f2])
    rb = nn_blocks.seq_apply(vote2, [x2.transpose(1, 2), f2], residual=True)
    assert oa.shape == (B, 1, N) and ob.shape == (B, 259, N)
    assert torch.equal(oa, ra) and torch.equal(ob, rb)
    ca, cb = torch.randn(oa.shape, device="cuda", generator=g), torch.randn(ob.shape, device="cuda", generator=g)
    ((oa * ca).sum() + (ob * cb).sum()).backward()
    ((ra * ca).sum() + (rb * cb).sum()).backward()
    assert torch.equal(f1.grad, f2.grad) and torch.equal(x1.grad, x2.grad)
    for m1, m2 in ((cla, cla2), (vote, vote2)):
        for (n1, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            assert p1.grad is not None and torch.equal(p1.grad, p2.grad), n1
        for (n1, b1), (_, b2) in zip(m1.named_buffers(), m2.named_buffers()):
            assert torch.equal(b1, b2), n1
    # one output unused by the loss: the other stack's gradients still arrive
    f3 = feat.clone().requires_grad_(True)
    oa3, ob3 = nn_blocks.seq_apply_pair((cla, [f3], False), (vote, [xyz.transpose(1, 2), f3], True))
    (oa3 * ca).sum().backward()
    assert f3.grad is not None and torch.isfinite(f3.grad).all()


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("B,N", [(3, 64), (48, 128)])
def test_rpn_glue_kernels_equal_the_reference_torch_ops(train, B, N):
    """The element-wise glue of P2BVoteNetRPN.forward (models/head/rpn.py:47-56,62-66: sigmoid, transposed vote
    coordinates, cat(score, vote features); offsets + centres, cat, transpose) as RpnVotes / BoxAssemble (round 5, one launch
    each way) against the torch-op form: every output and every gradient of the whole head"""
    from open3dsot_amd import rpn
    torch.manual_seed(4)
    head = rpn.P2BVoteNetRPN(256, vote_channel=256, num_proposal=N // 2).cuda().train(train)
    ref = copy.deepcopy(head)
    g = torch.Generator(device="cuda").manual_seed(6)
    xyz = torch.randn(B, N, 3, device="cuda", generator=g) * 0.5
    feat = torch.randn(B, 256, N, device="cuda", generator=g)
    outs, grads = [], []
    for m, on in ((head, True), (ref, False)):
        rpn.set_fused_glue(on)
        try:
            x, f = xyz.clone().requires_grad_(True), feat.clone().requires_grad_(True)
            boxes, cla, vote_xyz, centers = m(x, f)
            assert boxes.shape == (B, N // 2, 5) and boxes.is_contiguous() and vote_xyz.is_contiguous()
            cts = [torch.randn(t.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(30 + i))
                   for i, t in enumerate((boxes, cla, vote_xyz, centers))]
            sum((t * c).sum() for t, c in zip((boxes, cla, vote_xyz, centers), cts)).backward()
            outs.append([boxes, cla, vote_xyz, centers])
            grads.append([x.grad, f.grad] + [p.grad for p in m.parameters()])
        finally:
            rpn.set_fused_glue(True)
    for a, b in zip(outs[0], outs[1]):
        assert rel(a, b) < 2e-6, rel(a, b)
    for i, (a, b) in enumerate(zip(grads[0], grads[1])):
        assert (a is None) == (b is None), i
        if a is not None:
            assert l2rel(a, b) < 2e-5, (i, l2rel(a, b))      # (the list sums of the vote aggregation use LDS float atomics)


@pytest.mark.parametrize("B,Na,Nb", [(4, 64, 128), (48, 64, 128), (2, 32, 96)])
def test_shared_conv_pair_equals_two_convs(B, Na, Nb):
    """conv_final on the template and on the search feature (models/bat.py:91-92) as ONE GEMM over the columns of both
    (fused_heads.SharedConvPair) against the two separate calls and against nn.Conv1d in fp64: outputs (bitwise the
    single calls': every column is the same dot product), input gradients, the weight gradient summed over both sets,
    the bias gradient"""
    from open3dsot_amd import fused_heads, nn_blocks
    torch.manual_seed(5)
    conv = torch.nn.Conv1d(256, 256, 1).cuda()
    conv2 = copy.deepcopy(conv)
    ref = copy.deepcopy(conv).double()
    g = torch.Generator(device="cuda").manual_seed(9)
    xa = torch.randn(B, 256, Na, device="cuda", generator=g)
    xb = torch.randn(Nb, B, 256, device="cuda", generator=g).permute(1, 2, 0)          # a strided (B, C, N) view
    a1, b1 = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True)
    a2, b2 = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True)
    a3, b3 = xa.double().requires_grad_(True), xb.double().requires_grad_(True)
    assert fused_heads.shared_conv_pair_supported(conv, a1, b1) == ((B * (Na + Nb)) % 128 == 0)
    ya, yb = nn_blocks.pointwise_conv1d_pair(conv, a1, b1)
    ra, rb = nn_blocks.pointwise_conv1d(conv2, a2), nn_blocks.pointwise_conv1d(conv2, b2)
    wa, wb = ref(a3), ref(b3)
    assert ya.shape == ra.shape and yb.shape == rb.shape
    assert torch.equal(ya, ra) and torch.equal(yb, rb)
    assert rel(ya, wa) < 2e-5 and rel(yb, wb) < 2e-5
    ca, cb = torch.randn(ya.shape, device="cuda", generator=g), torch.randn(yb.shape, device="cuda", generator=g)
    ((ya * ca).sum() + (yb * cb).sum()).backward()
    ((ra * ca).sum() + (rb * cb).sum()).backward()
    ((wa * ca.double()).sum() + (wb * cb.double()).sum()).backward()
    assert torch.equal(a1.grad, a2.grad) and torch.equal(b1.grad, b2.grad)
    assert l2rel(a1.grad, a3.grad) < 5e-4 and l2rel(b1.grad, b3.grad) < 5e-4
    assert l2rel(conv.weight.grad, ref.weight.grad) < 5e-4 and l2rel(conv.bias.grad, ref.bias.grad) < 5e-4
    assert l2rel(conv.weight.grad, conv2.weight.grad) < 1e-5


def _poison_allocator(dev, sizes):
    """fill and free blocks of the sizes the backward is about to `torch.empty`: a gradient buffer that is handed out
    before it is written then holds NaNs, not plausible stale numbers"""
    for n in sizes:
        t = torch.full((n,), float("nan"), device=dev)
        del t


@pytest.mark.parametrize("which", ["shared_conv", "chain"])
def test_deferred_wgrads_equal_immediate_wgrads(which):
    """inside fused_heads.defer_wgrads() (the scope DataParallelStep wraps loss.backward() in) the stacks only QUEUE
    their weight / bias gradient jobs; what autograd stored in p.grad must be the tensors the flush fills, not clones taken
    before it (round-3 advisor finding: SharedConvPair returned `dbias` itself, AccumulateGrad cloned the unfilled
    buffer): every parameter gradient deferred == not deferred, bitwise (same launches, same order of summation)"""
    from open3dsot_amd import fused_heads, nn_blocks
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    g = torch.Generator(device="cuda").manual_seed(4)
    if which == "shared_conv":
        mods = [torch.nn.Conv1d(256, 256, 1).cuda() for _ in range(1)]
        xa = torch.randn(4, 256, 64, device=dev, generator=g)
        xb = torch.randn(4, 256, 128, device=dev, generator=g)

        def run(conv):
            ya, yb = nn_blocks.pointwise_conv1d_pair(conv, xa, xb)
            return (ya * ca).sum() + (yb * cb).sum()
        ca = torch.randn(4, 256, 64, device=dev, generator=g)
        cb = torch.randn(4, 256, 128, device=dev, generator=g)
        net = mods[0]
    else:
        net = build_seq([256, 256, 9], 259, 7).cuda().train()
        xyz = torch.randn(4, 3, 128, device=dev, generator=g)
        feat = torch.randn(4, 256, 128, device=dev, generator=g)
        ct = torch.randn(4, 9, 128, device=dev, generator=g)

        def run(seq):
            return (nn_blocks.seq_apply(seq, [xyz, feat]) * ct).sum()
    twin = copy.deepcopy(net)
    run(twin).backward()                              # immediate launches
    torch.cuda.synchronize()
    loss = run(net)
    _poison_allocator(dev, [256, 256 * 256, 9, 64, 259 * 256, 16])
    with fused_heads.defer_wgrads():
        loss.backward()
        _poison_allocator(dev, [256, 256 * 256, 9, 64])
    torch.cuda.synchronize()
    for (k, p), (_, q) in zip(net.named_parameters(), twin.named_parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        assert torch.equal(p.grad, q.grad), (k, float((p.grad - q.grad).abs().max()))
    if which == "shared_conv":
        assert float(net.bias.grad.abs().max()) > 1.0          # a real row sum, not rounding noise


def test_deferred_wgrads_are_not_deferred_into_an_existing_gradient():
    """a parameter that already holds a gradient gets `grad += new` from AccumulateGrad the moment the backward returns:
    its jobs must run at once, not at the end of the scope (fused_heads._deferrable)"""
    from open3dsot_amd import fused_heads, nn_blocks
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    conv = torch.nn.Conv1d(256, 256, 1).cuda()
    twin = copy.deepcopy(conv)
    g = torch.Generator(device="cuda").manual_seed(4)
    xa, xb = torch.randn(4, 256, 64, device=dev, generator=g), torch.randn(4, 256, 128, device=dev, generator=g)

    def run(c):
        ya, yb = nn_blocks.pointwise_conv1d_pair(c, xa, xb)
        return ya.sum() + 2.0 * yb.sum()
    run(twin).backward()
    run(twin).backward()                               # accumulated twice, immediate launches
    run(conv).backward()
    loss = run(conv)
    with fused_heads.defer_wgrads():
        loss.backward()                                # p.grad is set: must flush at once
    torch.cuda.synchronize()
    assert torch.equal(conv.weight.grad, twin.weight.grad) and torch.equal(conv.bias.grad, twin.bias.grad)


@pytest.mark.parametrize("B,M,N", [(48, 64, 128), (3, 32, 64), (2, 64, 96)])
def test_cosine_sim_map_matches_nn_cosine_similarity(B, M, N):
    """P2B_XCorr's similarity map (models/head/xcorr.py:37-38: nn.CosineSimilarity(dim=1) on the (B,f,M,N) expansion)
    as one kernel each way, against the reference formulation itself in fp64 -- values and both gradients; the inputs
    are strided views like the trackers' (B,C,N) features"""
    from open3dsot_amd import fused_xcorr
    g = torch.Generator(device="cuda").manual_seed(12)
    f = 256
    t = torch.randn(f, B, M, device="cuda", generator=g).permute(1, 0, 2).requires_grad_(True)        # (B,f,M) view
    s = torch.randn(B, N, f, device="cuda", generator=g).permute(0, 2, 1).requires_grad_(True)
    assert fused_xcorr.cosine_sim_supported(t, s)
    sim = fused_xcorr.CosineSimMap.apply(t, s)
    ct = torch.randn(B, N, M, device="cuda", generator=g)
    gt, gs = torch.autograd.grad((sim * ct).sum(), (t, s))
    t64, s64 = t.detach().double().requires_grad_(True), s.detach().double().requires_grad_(True)
    ref = torch.nn.CosineSimilarity(dim=1)(t64.unsqueeze(-1).expand(B, f, M, N), s64.unsqueeze(2).expand(B, f, M, N))
    ref = ref.transpose(1, 2)                                                                           # (B,N,M)
    rt, rs = torch.autograd.grad((ref * ct.double()).sum(), (t64, s64))
    assert rel(sim, ref) < 2e-6, rel(sim, ref)
    assert l2rel(gt, rt) < 2e-6 and l2rel(gs, rs) < 2e-6, (l2rel(gt, rt), l2rel(gs, rs))
