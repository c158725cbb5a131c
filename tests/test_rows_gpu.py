"""GPU parity of the row stacks (open3dsot_amd/fused_rows.py on csrc/rowmlp.hip) against the nn.Sequential they are,
evaluated by torch in fp64: the heads of M2-Track (models/m2track.py:43-71) and the hidden rows of MiniPointNet
(models/backbone/pointnet.py:118-126).  Forward 1e-5 of the tensor scale, gradients 1e-4 L2, running statistics 1e-6."""
import copy

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def l2rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def build(kind, seed):
    from open3dsot_amd.nn_blocks import RowBatchNorm1d
    torch.manual_seed(seed)
    if kind == "head4":
        seq = nn.Sequential(nn.Linear(256, 128), RowBatchNorm1d(128), nn.ReLU(), nn.Linear(128, 128), RowBatchNorm1d(128), nn.ReLU(),
                            nn.Linear(128, 4))
        cin = 256
    elif kind == "hidden":
        seq = nn.Sequential(nn.Linear(512, 512), RowBatchNorm1d(512), nn.ReLU(), nn.Linear(512, 256), RowBatchNorm1d(256), nn.ReLU())
        cin = 512
    else:       # odd widths: partial feature slices, unaligned K
        seq = nn.Sequential(nn.Linear(70, 40), RowBatchNorm1d(40), nn.ReLU(), nn.Linear(40, 9))
        cin = 70
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in seq.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.weight.copy_(torch.empty(m.weight.shape).uniform_(0.5, 1.5, generator=g))
                m.bias.copy_(torch.empty(m.bias.shape).normal_(0, 0.2, generator=g))
                m.running_mean.copy_(torch.empty(m.bias.shape).normal_(0, 0.2, generator=g))
                m.running_var.copy_(torch.empty(m.bias.shape).uniform_(0.5, 1.5, generator=g))
    return seq, cin


@pytest.mark.parametrize("kind", ["head4", "hidden", "odd"])
@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("R", [48, 64, 5])
def test_row_stack_vs_fp64(kind, train, R):
    from open3dsot_amd import fused_rows
    seq, cin = build(kind, 3)
    ref = copy.deepcopy(seq).double().train(train)
    seq = seq.cuda().train(train)
    g = torch.Generator().manual_seed(R)
    x = torch.randn(R, cin, generator=g)
    xg = x.cuda().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    layers = fused_rows.parse(seq)
    assert fused_rows.supported(layers, xg)
    out = fused_rows.seq_rows(seq, xg)
    assert out.grad_fn is not None and "RowStack" in out.grad_fn.name()
    want = ref(xr)
    assert rel(out, want) < 1e-5, rel(out, want)
    ct = torch.randn(out.shape, generator=g)
    (out * ct.cuda()).sum().backward()
    (want * ct.double()).sum().backward()
    assert l2rel(xg.grad, xr.grad) < 1e-4, l2rel(xg.grad, xr.grad)
    for (n1, p), (_, q) in zip(seq.named_parameters(), ref.named_parameters()):
        scale = float(q.grad.abs().max())
        if scale < 1e-9:                # a bias in front of a training-mode BatchNorm: its true gradient is zero
            assert float(p.grad.abs().max()) < 1e-4, n1
            continue
        assert l2rel(p.grad, q.grad) < 1e-4, (n1, l2rel(p.grad, q.grad))
    for (n1, b1), (_, b2) in zip(seq.named_buffers(), ref.named_buffers()):
        if b1.dtype.is_floating_point:
            assert rel(b1, b2) < 1e-6, n1
        else:
            assert int(b1) == int(b2), n1           # num_batches_tracked


def test_row_stack_takes_strided_rows_and_falls_back_beyond_64_rows():
    from open3dsot_amd import fused_rows
    seq, cin = build("head4", 5)
    seq = seq.cuda().train()
    wide = torch.randn(48, cin + 8, device="cuda")
    x = wide[:, 4:4 + cin]                       # rows with a stride larger than their length
    a = fused_rows.seq_rows(seq, x)
    twin = copy.deepcopy(seq)
    # (the twin's running statistics start where seq's were BEFORE the call above: restore them for the comparison)
    b = seq_ref_forward(twin, x.contiguous())
    assert rel(a, b) < 1e-5
    big = torch.randn(65, cin, device="cuda")
    out = fused_rows.seq_rows(seq, big)
    assert "RowStack" not in (out.grad_fn.name() if out.grad_fn is not None else "")


def seq_ref_forward(seq, x):
    ref = copy.deepcopy(seq).double()
    return ref(x.double())


@pytest.mark.gpu
@pytest.mark.parametrize("with_prev", [True, False])
def test_motion_merge_matches_the_reference_helpers(with_prev):
    """csrc/boxcloud.hip::motion_merge (M2-Track between its stages, one launch each way) against the chain of
    datasets/points_utils.py's tensor helpers (box_utils.motion_merge_reference) evaluated in fp64: values and the gradients
    with respect to the previous box and the motion, with gradient arriving through both outputs."""
    from open3dsot_amd import box_utils
    g = torch.Generator().manual_seed(5)
    B, N = 48, 1024
    pts = torch.randn(B, 4, N, generator=g) * 2.0
    pts[:, :, ::7] = 0.0                                     # masked-out points sit at the origin
    prev = torch.randn(B, 4, generator=g) * 0.7 if with_prev else None
    motion = torch.randn(B, 4, generator=g) * 0.5
    gm, ga = torch.randn(B, 3, N, generator=g), torch.randn(B, 4, generator=g)
    p64 = prev.double().requires_grad_() if with_prev else None
    m64 = motion.double().requires_grad_()
    ref_m, ref_a = box_utils.motion_merge_reference(pts.double(), p64, m64)
    ((ref_m * gm.double()).sum() + (ref_a * ga.double()).sum()).backward()
    pg = prev.cuda().requires_grad_() if with_prev else None
    mg = motion.cuda().requires_grad_()
    big = torch.cat([pts, torch.zeros(B, 9, N)], 1).cuda()   # the model hands over a channel slice of a wider tensor
    view = big[:, :4]
    assert box_utils.motion_merge_supported(view, mg)
    merged, aux = box_utils.MotionMerge.apply(view, pg, mg)
    assert merged.shape == (B, 3, N) and aux.shape == (B, 4)
    scale = float(ref_m.abs().max())
    assert float((merged.cpu().double() - ref_m).abs().max()) <= 2e-6 * scale
    assert float((aux.cpu().double() - ref_a).abs().max()) <= 2e-6 * float(ref_a.abs().max())
    ((merged * gm.cuda()).sum() + (aux * ga.cuda()).sum()).backward()
    pairs = [(mg, m64)] + ([(pg, p64)] if with_prev else [])
    for got, ref in pairs:
        err = float((got.grad.cpu().double() - ref.grad).abs().max())
        assert err <= 1e-5 * float(ref.grad.abs().max()), (err, float(ref.grad.abs().max()))
    # gradient through the first-stage box only (the second stage switched off upstream): the points' sums are zero
    mg2 = motion.cuda().requires_grad_()
    _, aux2 = box_utils.MotionMerge.apply(view, pg.detach() if with_prev else None, mg2)
    (aux2 * ga.cuda()).sum().backward()
    m3 = motion.double().requires_grad_()
    (box_utils.get_offset_box_tensor(prev.double() if with_prev else torch.zeros(B, 4, dtype=torch.float64), m3) * ga.double()).sum().backward()
    assert float((mg2.grad.cpu().double() - m3.grad).abs().max()) <= 1e-5 * float(m3.grad.abs().max())


@pytest.mark.gpu
def test_offset_box_kernel_matches_the_torch_form():
    """box_utils.OffsetBox (one launch each way) against get_offset_box_tensor_reference (datasets/points_utils.py:420-436
    restated with torch ops) in fp64; gradient to either operand alone as well"""
    from open3dsot_amd import box_utils
    g = torch.Generator().manual_seed(11)
    for B in (1, 48, 100):
        ref, off, gb = torch.randn(B, 4, generator=g) * 2, torch.randn(B, 4, generator=g), torch.randn(B, 4, generator=g)
        r64, o64 = ref.double().requires_grad_(), off.double().requires_grad_()
        want = box_utils.get_offset_box_tensor_reference(r64, o64)
        (want * gb.double()).sum().backward()
        rg, og = ref.cuda().requires_grad_(), off.cuda().requires_grad_()
        got = box_utils.get_offset_box_tensor(rg, og)
        assert got.grad_fn is not None and type(got.grad_fn).__name__.startswith("OffsetBox")
        (got * gb.cuda()).sum().backward()
        assert float((got.detach().cpu().double() - want.detach()).abs().max()) <= 2e-6 * float(want.abs().max())
        for a, b in ((rg, r64), (og, o64)):
            assert float((a.grad.cpu().double() - b.grad).abs().max()) <= 4e-6 * float(b.grad.abs().max())
        og2 = off.cuda().requires_grad_()
        (box_utils.get_offset_box_tensor(ref.cuda(), og2) * gb.cuda()).sum().backward()
        assert torch.allclose(og2.grad, og.grad, rtol=0, atol=0)


@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("unused_head", [False, True])
def test_row_stack_group_equals_the_single_stacks(train, unused_head):
    """fused_rows.seq_rows_group (three heads of different output widths on the same rows, one launch per layer) against
    the same heads one at a time and against fp64: outputs, every parameter gradient, the summed input gradient, running
    statistics; a head whose output nothing reads contributes exactly zero."""
    from open3dsot_amd import fused_rows
    from open3dsot_amd.nn_blocks import RowBatchNorm1d

    def head(out, seed):
        torch.manual_seed(seed)
        seq = nn.Sequential(nn.Linear(256, 128), RowBatchNorm1d(128), nn.ReLU(), nn.Linear(128, 128), RowBatchNorm1d(128), nn.ReLU(),
                            nn.Linear(128, out))
        with torch.no_grad():
            for m in seq.modules():
                if isinstance(m, nn.BatchNorm1d):
                    m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.2); m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
        return seq
    heads = [head(4, 1), head(2, 2), head(4, 3)]
    refs = [copy.deepcopy(h).double().train(train) for h in heads]
    singles = [copy.deepcopy(h).cuda().train(train) for h in heads]
    heads = [h.cuda().train(train) for h in heads]
    R = 48
    x = torch.randn(R, 256, generator=torch.Generator().manual_seed(7))
    xg, xs, xr = x.cuda().requires_grad_(True), x.cuda().requires_grad_(True), x.double().requires_grad_(True)
    outs = fused_rows.seq_rows_group(heads, xg)
    assert type(outs[0].grad_fn).__name__.startswith("RowStackGroup")
    souts = [fused_rows.seq_rows(h, xs) for h in singles]
    routs = [h(xr) for h in refs]
    gos = [torch.randn(R, o.shape[1], generator=torch.Generator().manual_seed(20 + i)) for i, o in enumerate(outs)]
    live = [0, 2] if unused_head else [0, 1, 2]
    sum((outs[i] * gos[i].cuda()).sum() for i in live).backward()
    sum((souts[i] * gos[i].cuda()).sum() for i in live).backward()
    sum((routs[i] * gos[i].double()).sum() for i in live).backward()
    for o, s, r in zip(outs, souts, routs):
        assert torch.equal(o, s)                       # the same kernel body: bit-identical
        assert rel(o, r) < 1e-5
    assert l2rel(xg.grad, xr.grad) < 1e-4 and l2rel(xg.grad, xs.grad) < 1e-6
    for i, (h, s, r) in enumerate(zip(heads, singles, refs)):
        for (n, p), (_, q), (_, t) in zip(h.named_parameters(), s.named_parameters(), r.named_parameters()):
            if i not in live:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
                continue
            assert torch.equal(p.grad, q.grad), n
            # (a Linear bias in front of a training-mode BatchNorm has an exactly zero gradient: judged on the head's scale)
            scale = max(float(u.grad.abs().max()) for u in r.parameters())
            assert float((p.grad.cpu().double() - t.grad).abs().max()) < 1e-4 * scale, n
        if train:
            for (n, b), (_, c) in zip(h.named_buffers(), r.named_buffers()):
                if b.dtype.is_floating_point:
                    assert rel(b, c) < 1e-6, n
                else:
                    assert int(b) == int(c), n
