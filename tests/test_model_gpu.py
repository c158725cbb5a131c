"""GPU parity of the whole hot path: BAT / P2B forward + loss + backward on the MI355X
(HIP operator set; composed and fused execution paths) vs the CPU oracle restatement
(oracle/torch_ref.py, itself pinned to the reference's Python layers by tests/golden).
Indices bit-exact; features and losses within 1e-4 relative-to-scale against the fp32 oracle.
Gradients are judged against the SAME oracle evaluated in fp64, with the fp32 oracle's own
distance to fp64 as the yardstick.  Why not a fixed 1e-4: every BatchNorm backward projects
out the mean and the x-hat component of its incoming gradient, which amplifies fp32 rounding
noise layer after layer (some 30 BatchNorm layers deep), and the objectness loss at random
initialisation is nearly constant over the proposals, so almost all of its gradient is
projected away.  Measured on the GPU box (tools/diag_grad2.py, diag_grad5.py): torch's own
CPU fp32 backward sits 0.5-2e-2 from its fp64 run, torch+MIOpen on the GPU likewise, and the
per-term differences between any two fp32 evaluations are 2e-3 (box/seg/vote) to 6e-2
(objectness).  The tight gradient checks live in tests/test_fused_gpu.py (one SA module at a
time, L2 error <= 5e-4 against fp64)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref

pytestmark = pytest.mark.gpu


OUT_KEYS = ("estimation_boxes", "estimation_cla", "vote_xyz", "center_xyz", "pred_search_bc")


def make_model(model_name, seed, train=True):
    from open3dsot_amd import trackers
    dev = torch.device("cuda", 0)
    torch.manual_seed(seed)
    model = trackers.get_model(model_name)().to(dev).train(train)
    # non-trivial BatchNorm affine/running statistics
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.weight.copy_(torch.empty(m.weight.shape).uniform_(0.5, 1.5, generator=g))
                m.bias.copy_(torch.empty(m.bias.shape).normal_(0, 0.1, generator=g))
                m.running_mean.copy_(torch.empty(m.bias.shape).normal_(0, 0.1, generator=g))
                m.running_var.copy_(torch.empty(m.bias.shape).uniform_(0.5, 1.5, generator=g))
    return model


def cotangents(out, seed=11):
    """fixed random cotangent per output tensor: a well-conditioned scalar for backward parity"""
    g = torch.Generator().manual_seed(seed)
    return {k: torch.randn(out[k].shape, generator=g, dtype=torch.float64) for k in OUT_KEYS if k in out}


def oracle_run(model_name, sd, host, dtype, objective):
    """CPU oracle (oracle/torch_ref.py) in `dtype`; objective 'loss' | 'proj' -> (scalar, loss dict, grads, sd)"""
    from open3dsot_amd import synth, trackers
    sdx = {k: (v.detach().clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    for k, v in sdx.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    b = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in synth.to_torch(host).items()}
    fwd = torch_ref.bat_forward if model_name == "BAT" else torch_ref.p2b_forward
    out = fwd(sdx, b, True)
    cfg = trackers.BAT_CAR if model_name == "BAT" else trackers.P2B_CAR
    w = {k: v for k, v in cfg.items() if k.endswith("_weight")}
    loss, ld = torch_ref.matching_loss(b, out, w, bat=model_name == "BAT")
    if objective == "proj":
        ct = cotangents(out)
        scalar = sum((out[k] * ct[k].to(dtype)).sum() for k in ct)
        ld = {"abs_scale": sum((out[k] * ct[k].to(dtype)).abs().sum() for k in ct)}   # size of the summed terms
    else:
        scalar = loss
    scalar.backward()
    return float(scalar.detach()), {k: float(v.detach()) for k, v in ld.items()}, \
        {k: v.grad.double() for k, v in sdx.items() if v.requires_grad and v.grad is not None}, sdx


def gpu_run(model, sd, host, fused, objective):
    from open3dsot_amd import sa_modules, synth
    dev = torch.device("cuda", 0)
    batch = synth.to_torch(host, dev)
    was = sa_modules.fused_enabled()
    sa_modules.set_fused(fused)
    try:
        model.load_state_dict(sd)
        model.zero_grad(set_to_none=True)
        if objective == "proj":
            out = model(batch)
            ct = cotangents(out)
            scalar = sum((out[k] * ct[k].to(dev, torch.float32)).sum() for k in ct)
            ld = {}
        else:
            scalar, ld = model.training_loss(batch)
        scalar.backward()
    finally:
        sa_modules.set_fused(was)
    torch.cuda.synchronize()
    return float(scalar.detach()), {k: float(v.detach()) for k, v in ld.items()}, \
        {k: p.grad.detach().cpu().double() for k, p in model.named_parameters() if p.grad is not None}


def grad_errors(g, truth):
    """per-parameter relative L2 error, judged only where the true gradient is not rounding noise
    (rms above 1e-3 of the largest rms: a bias in front of a training-mode BatchNorm has none)"""
    rms = {k: float(v.norm() / v.numel() ** 0.5) for k, v in truth.items()}
    top = max(rms.values())
    return {k: float((g[k] - truth[k]).norm() / truth[k].norm()) for k in truth if rms[k] > 1e-3 * top}


def flat_cos(g, truth):
    a = torch.cat([g[k].flatten() for k in truth])
    b = torch.cat([truth[k].flatten() for k in truth])
    return float(a @ b / (a.norm() * b.norm())), float(a.norm() / b.norm())


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("model_name", ["BAT", "P2B"])
@pytest.mark.parametrize("fused", [False, True])
def test_training_step_matches_oracle(model_name, fused):
    """loss terms + BatchNorm running statistics vs the fp32 oracle (1e-4); training-loss gradient
    direction and norm vs the fp64 shadow (loose: the objectness term is ill-conditioned at random
    initialisation, see the module docstring)."""
    from open3dsot_amd import synth
    model = make_model(model_name, 0)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    host = synth.make_batch(300, 3, 256, 512)
    loss, ld, g = gpu_run(model, sd, host, fused, "loss")
    ref_loss, ref_ld, _, sd32 = oracle_run(model_name, sd, host, torch.float32, "loss")
    _, _, g64, _ = oracle_run(model_name, sd, host, torch.float64, "loss")
    assert abs(loss - ref_loss) <= 1e-4 * (1 + abs(ref_loss)), (loss, ref_loss)
    for k in ref_ld:
        assert abs(ld[k] - ref_ld[k]) <= 1e-4 * (1 + abs(ref_ld[k])), k
    cos, ratio = flat_cos(g, g64)
    assert cos > 0.995 and abs(ratio - 1) < 0.03, (cos, ratio)
    for k, v in model.state_dict().items():   # BatchNorm running statistics advanced identically
        if "running" in k:
            assert rel(v, sd32[k]) < 1e-4, k


@pytest.mark.parametrize("model_name", ["BAT", "P2B"])
def test_backward_random_cotangent(model_name):
    """Backward parity on a well-conditioned scalar (fixed random cotangent on every output) of the
    fused kernels and of the composed operator path, both against the fp64 shadow of the oracle.
    The fp32 CPU oracle's own distance to fp64 is the yardstick: an implementation passes when its
    median per-parameter L2 error is within 3x that (+2e-3) and no parameter is beyond 5e-2."""
    from open3dsot_amd import synth
    model = make_model(model_name, 1)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    host = synth.make_batch(340, 4, 256, 512)
    _, _, g64, _ = oracle_run(model_name, sd, host, torch.float64, "proj")
    s32, info, g32, _ = oracle_run(model_name, sd, host, torch.float32, "proj")
    e32 = grad_errors(g32, g64)
    yard = float(np.median(list(e32.values())))
    report = {"cpu32": (yard, max(e32.values()))}
    for fused in (False, True):
        s, _, g = gpu_run(model, sd, host, fused, "proj")
        assert abs(s - s32) <= 1e-5 * info["abs_scale"], (fused, s, s32, info)   # a +- sum of 1e5 terms
        e = grad_errors(g, g64)
        report["fused" if fused else "composed"] = (float(np.median(list(e.values()))), max(e.values()))
    print("median / worst per-parameter L2 error vs fp64:", report)
    for k in ("composed", "fused"):
        assert report[k][0] <= 3 * yard + 2e-3 and report[k][1] <= 5e-2, report


def run_pair(model_name, fused, B=3, M=256, N=512, seed=0, train=True):
    from open3dsot_amd import sa_modules, synth
    dev = torch.device("cuda", 0)
    model = make_model(model_name, seed, train)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    host = synth.make_batch(300 + seed, B, M, N)
    batch = synth.to_torch(host, dev)
    was = sa_modules.fused_enabled()
    sa_modules.set_fused(fused)
    try:
        out = model(batch)
    finally:
        sa_modules.set_fused(was)
    torch.cuda.synchronize()
    fwd = torch_ref.bat_forward if model_name == "BAT" else torch_ref.p2b_forward
    return model, out, fwd(sd, synth.to_torch(host), train)


@pytest.mark.parametrize("fused", [False, True])
def test_eval_forward_matches_oracle(fused):
    model, out, ref = run_pair("BAT", fused, train=False)
    assert np.array_equal(out["sample_idxs"].cpu().numpy(), ref["sample_idxs"].numpy())
    for k in ("estimation_boxes", "estimation_cla", "vote_xyz", "center_xyz", "pred_search_bc"):
        assert rel(out[k], ref[k]) < 1e-4, k


def oracle_forward_loss(model_name, sd, host, train=True):
    """fp32 CPU oracle, forward + loss terms only (no backward) -> (end points, loss dict, state_dict after)"""
    from open3dsot_amd import synth, trackers
    sdx = {k: v.detach().clone() for k, v in sd.items()}
    b = synth.to_torch(host)
    with torch.no_grad():
        out = (torch_ref.bat_forward if model_name == "BAT" else torch_ref.p2b_forward)(sdx, b, train)
        cfg = trackers.BAT_CAR if model_name == "BAT" else trackers.P2B_CAR
        loss, ld = torch_ref.matching_loss(b, out, {k: v for k, v in cfg.items() if k.endswith("_weight")},
                                           bat=model_name == "BAT")
    ld = dict(ld, total=loss)
    return out, {k: float(v) for k, v in ld.items()}, sdx


@pytest.mark.parametrize("model_name", ["BAT", "P2B"])
def test_full_size_batch48_forward_losses_stats(model_name):
    """BASELINE config 2 itself (48 pairs, template 512 / search 1024 points), the shape bench.py times: sampling
    indices equal, every end point, every loss term and every BatchNorm running statistic within 1e-4 of the fp32
    CPU oracle (oracle/torch_ref.py, pinned on the reference's own BAT / P2B classes by tests/golden)."""
    from open3dsot_amd import synth
    dev = torch.device("cuda", 0)
    model = make_model(model_name, 3)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    host = synth.make_batch(100, 48)
    batch = synth.to_torch(host, dev)
    out = model(batch)
    n_seed = out["estimation_cla"].shape[1]
    data = dict(batch)
    sidx = out["sample_idxs"][:, :n_seed].long()
    data["seg_label"] = batch["seg_label"].gather(1, sidx)
    if model_name == "BAT":
        data["points2cc_dist_s"] = batch["points2cc_dist_s"].gather(1, sidx[:, :, None].expand(-1, -1, 9))
    ld = model.compute_loss(data, out)            # the torch-op specification of the loss, on the GPU end points
    from open3dsot_amd import fused_loss
    total, ld_fused = fused_loss.track_loss(model.config, data, out, with_bc=model_name == "BAT")
    torch.cuda.synchronize()
    ref, ref_ld, sd_after = oracle_forward_loss(model_name, sd, host)
    assert np.array_equal(out["sample_idxs"].cpu().numpy(), ref["sample_idxs"].numpy())
    for k in OUT_KEYS:
        if k in ref:
            assert rel(out[k], ref[k]) < 1e-4, (k, rel(out[k], ref[k]))
    for k, v in ref_ld.items():
        if k == "total":
            assert abs(float(total) - v) <= 1e-4 * (1 + abs(v)), (k, float(total), v)
            continue
        assert abs(float(ld[k]) - v) <= 1e-4 * (1 + abs(v)), (k, float(ld[k]), v)
        assert abs(float(ld_fused[k]) - v) <= 1e-4 * (1 + abs(v)), ("fused", k, float(ld_fused[k]), v)
    for k, v in model.state_dict().items():
        if "running" in k:
            assert rel(v, sd_after[k]) < 1e-4, (k, rel(v, sd_after[k]))
        elif "num_batches" in k:
            assert int(v) == int(sd_after[k]), k


@pytest.mark.parametrize("model_name", ["BAT", "P2B"])
def test_full_size_batch48_gradients_vs_fp64(model_name):
    """Training-loss gradient of the benchmarked step (48 pairs, 512/1024) against the fp64 evaluation of the oracle,
    PER PARAMETER (round 4; rounds 2-3 asserted only the whole vector's direction and norm here): every key within
    max(2e-2, 3 x the fp32 CPU oracle's own distance to fp64 on that key) -- the bar tests/test_golden_trackers_b8.py holds
    at batch 8 against the reference's own classes, now at the tile / slice plans / segment offsets of the benchmarked
    batch.  Keys whose true gradient is zero (a bias in front of a training-mode BatchNorm) must be rounding noise.  The
    module docstring explains why end-to-end gradients are not a 1e-4 quantity."""
    from open3dsot_amd import synth
    _batch48_gradients_vs_fp64(model_name, synth.make_batch(148, 48), "B=48")


def _batch48_gradients_vs_fp64(model_name, host, tag, seed=4, gpu_yardstick=False):
    """gpu_yardstick: the operator-by-operator GPU path (torch's own fp32 GPU kernels for convolution and BatchNorm) against
    fp64 as a second yardstick beside the fp32 CPU oracle's -- for ill-conditioned batches (one pair), where what an fp32
    GPU GEMM can reach is not what the CPU's fp32 run reaches (test_batch1_training_step_...)"""
    model = make_model(model_name, seed)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    gt = gpu_run(model, sd, host, False, "loss")[2] if gpu_yardstick else None
    loss, ld, g = gpu_run(model, sd, host, True, "loss")
    l64, _, g64, _ = oracle_run(model_name, sd, host, torch.float64, "loss")
    _, _, g32, _ = oracle_run(model_name, sd, host, torch.float32, "loss")
    assert abs(loss - l64) <= 1e-4 * (1 + abs(l64)), (loss, l64)
    assert set(g) == set(g64), set(g) ^ set(g64)
    gnorm = float(torch.cat([v.flatten() for v in g64.values()]).norm())
    num = den = num32 = 0.0
    worst = ("", 0.0, 0.0)
    for k, want in g64.items():
        num += float((g[k] - want).pow(2).sum())
        num32 += float((g32[k] - want).pow(2).sum())
        den += float(want.pow(2).sum())
        if float(want.norm()) < 1e-5 * gnorm:          # mathematically zero
            assert float(g[k].norm()) < 1e-4 * gnorm, (k, float(g[k].norm()), gnorm)
            continue
        err = float((g[k] - want).norm() / want.norm())
        yard = float((g32[k] - want).norm() / want.norm())
        if gt is not None:
            yard = max(yard, float((gt[k] - want).norm() / want.norm()))
        if err > worst[1]:
            worst = (k, err, yard)
        assert err <= max(2e-2, 3.0 * yard), (k, err, "fp32 CPU oracle (/ torch GPU kernels) vs fp64 on this key:", yard)
    whole, whole32 = (num / den) ** 0.5, (num32 / den) ** 0.5
    if gt is not None:
        whole32 = max(whole32, (sum(float((gt[k] - w).pow(2).sum()) for k, w in g64.items()) / den) ** 0.5)
    cos, ratio = flat_cos(g, g64)
    print("%s %s gradient vs fp64: whole-gradient L2 error %.2e (fp32 CPU oracle: %.2e), cos %.6f, norm ratio %.4f; worst "
          "key %s %.2e (fp32 CPU oracle on it: %.2e)" % (model_name, tag, whole, whole32, cos, ratio, *worst))
    assert whole <= max(2e-2, 1.5 * whole32), (whole, whole32)
    # direction and norm of the whole vector: 0.995 / 3 % wherever the yardstick itself is a percent quantity; on an
    # ill-conditioned batch (one pair: the fp32 CPU oracle itself is 8e-2 from fp64) what the yardstick allows
    cos_min, ratio_tol = (0.995, 0.03) if whole32 < 3e-2 else (1.0 - 2.0 * whole32 ** 2, 1.5 * whole32)
    assert cos > cos_min and abs(ratio - 1) < ratio_tol, (cos, ratio, cos_min, ratio_tol)


def oracle_forward(model_name, sd, host, dtype, train=True):
    """CPU oracle forward in `dtype` (the index operators see the float32 coordinates either way) -> (end points, loss dict
    incl. 'total', state dict after)"""
    from open3dsot_amd import synth, trackers
    sdx = {k: (v.detach().clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    b = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in synth.to_torch(host).items()}
    with torch.no_grad():
        out = (torch_ref.bat_forward if model_name == "BAT" else torch_ref.p2b_forward)(sdx, b, train)
        cfg = trackers.BAT_CAR if model_name == "BAT" else trackers.P2B_CAR
        loss, ld = torch_ref.matching_loss(b, out, {k: v for k, v in cfg.items() if k.endswith("_weight")},
                                           bat=model_name == "BAT")
    return out, {k: float(v) for k, v in dict(ld, total=loss).items()}, sdx


def _forward_losses_stats(model, model_name, batch, fused):
    """one training forward of `model` (its state dict is put back afterwards) -> (end points, loss terms incl. 'total',
    running statistics after)"""
    from open3dsot_amd import fused_loss, sa_modules
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    was = sa_modules.fused_enabled()
    sa_modules.set_fused(fused)
    try:
        with torch.no_grad():
            out = model(batch)
            data = dict(batch)
            sidx = out["sample_idxs"][:, :out["estimation_cla"].shape[1]].long()
            data["seg_label"] = batch["seg_label"].gather(1, sidx)
            if model_name == "BAT":
                data["points2cc_dist_s"] = batch["points2cc_dist_s"].gather(1, sidx[:, :, None].expand(-1, -1, 9))
            total, ld = fused_loss.track_loss(model.config, data, out, with_bc=model_name == "BAT")
        torch.cuda.synchronize()
        after = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    finally:
        sa_modules.set_fused(was)
        model.load_state_dict(sd)
    return ({k: v.detach().cpu() for k, v in out.items()}, dict({k: float(v) for k, v in ld.items()}, total=float(total)), after)


@pytest.mark.parametrize("model_name", ["P2B", "BAT"])
def test_batch1_training_step_forward_losses_stats_gradients(model_name):
    """BASELINE config 1's shape -- ONE template / search pair, 512 / 1 024 points, training mode (cfgs/P2B_Car.yaml with
    batch 1; the `p2b_batch1` bench line times exactly this step): the split-K / 32-row tile plans, BatchNorm partial
    lists of a single cloud and 128-column heads that no other training parity test reaches.  Sampling indices exact.  Every
    end point, loss term and running statistic against the fp64 evaluation of the CPU oracle under TWO yardsticks, because
    one pair is an ill-conditioned batch: `xcorr.fea_layer`'s BatchNorm normalises 128 nearly identical pooled features
    (per-channel std / |mean| down to 1e-3 at random initialisation: tools/exp/diag_p2b_b1.py, gpurun_out/diag_p2b_b1.txt)
    and amplifies ANY fp32 GEMM's rounding a thousandfold -- torch's own GPU kernels (the operator-by-operator path:
    rocBLAS / MIOpen convolutions, torch BatchNorm) sit 1-4e-3 from fp64 behind it where the CPU's fp32 run sits at 1e-4, with
    every stage in front of it at 3e-6 on all three.  Bound per quantity: max(1e-4, 3 x the fp32 CPU oracle's distance to
    fp64, 5 x the distance of torch's GPU kernels to fp64) -- 5, not 3: behind that BatchNorm this path's split-K summation
    order sits at twice torch-GPU's distance (4.4e-3 against 2.3e-3 on the fused feature) and torch's own distance moves by
    10 % from run to run (8.2e-5 / 9.1e-5 on the vote loss in two runs of this test).  Quantities BEHIND the vote aggregation's ball query (a discrete
    decision on predicted coordinates) are only compared when the query groups the same points as the oracle's.
    Gradients: every parameter under the rule of test_full_size_batch48_gradients_vs_fp64, same two yardsticks."""
    from open3dsot_amd import synth
    from oracle import ops as oops
    dev = torch.device("cuda", 0)
    host = synth.make_batch(171, 1)
    model = make_model(model_name, 6)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    batch = synth.to_torch(host, dev)
    assert batch["template_points"].shape == (1, 512, 3) and batch["search_points"].shape == (1, 1024, 3)
    out, ld, after = _forward_losses_stats(model, model_name, batch, True)
    out_t, ld_t, after_t = _forward_losses_stats(model, model_name, batch, False)     # torch's GPU kernels: second yardstick
    ref32, ld32, sd32 = oracle_forward(model_name, sd, host, torch.float32)
    ref64, ld64, sd64 = oracle_forward(model_name, sd, host, torch.float64)
    assert np.array_equal(out["sample_idxs"].numpy(), ref32["sample_idxs"].numpy())

    def vote_groups(o):      # the vote aggregation's grouping (models/head/rpn.py:58-60): first 64 votes, radius 0.3, 16 samples
        v = o["vote_xyz"].detach().float().contiguous().numpy()
        return oops.ball_query(np.ascontiguousarray(v[:, :64]), v, 0.3, 16)
    same_groups = np.array_equal(vote_groups(out), vote_groups(ref64))
    behind = ("estimation_boxes", "loss.loss_box", "loss.loss_objective", "loss.total", "rpn.vote_aggregation", "rpn.FC_proposal")
    rows = []
    for k in OUT_KEYS:
        if k in ref64:
            rows.append((k, rel(out[k], ref64[k]), rel(ref32[k], ref64[k]), rel(out_t[k], ref64[k])))
    for k, v in ld64.items():
        rows.append(("loss." + k, abs(ld[k] - v) / (1 + abs(v)), abs(ld32[k] - v) / (1 + abs(v)), abs(ld_t[k] - v) / (1 + abs(v))))
    for k, v in after.items():
        if "running" in k:
            rows.append((k, rel(v, sd64[k]), rel(sd32[k], sd64[k]), rel(after_t[k], sd64[k])))
        elif "num_batches" in k:
            assert int(v) == int(sd32[k]), k
    if not same_groups:
        print("%s batch 1: the vote aggregation's ball query groups other points than the oracle's (a decision on predicted "
              "coordinates): the %d quantities behind it are not compared" % (model_name, sum(r[0].startswith(behind) for r in rows)))
        rows = [r for r in rows if not r[0].startswith(behind)]
    bound = lambda r: max(1e-4, 3 * r[2], 5 * r[3])
    worst = max(rows, key=lambda r: r[1] / bound(r))
    print("%s batch 1: %d quantities, worst vs fp64: %s err %.2e (fp32 CPU oracle: %.2e, torch's GPU kernels: %.2e)"
          % ((model_name, len(rows)) + worst))
    tight = [r for r in rows if r[1] <= max(1e-4, 3 * r[2])]
    print("   %d of %d within max(1e-4, 3 x the CPU yardstick) alone" % (len(tight), len(rows)))
    bad = [r for r in rows if not r[1] <= bound(r)]
    for r in bad:
        print("   BAD %-60s err vs fp64 %.3e | fp32 CPU oracle vs fp64 %.3e | torch GPU kernels vs fp64 %.3e" % r)
    assert not bad, [r[0] for r in bad]
    if same_groups:
        _batch48_gradients_vs_fp64(model_name, host, "B=1", seed=6, gpu_yardstick=True)


@pytest.mark.parametrize("model_name", ["P2B", "BAT"])
def test_batch1_graph_replay_equals_eager_step(model_name):
    """the captured batch-1 training step (what `bench.py --model p2b --batch 1` replays) on a NEW pair against an eager
    forward + backward of a twin with the same weights and running statistics: loss and every parameter gradient"""
    import copy
    from open3dsot_amd import dist as D, synth, trackers
    dev = torch.device("cuda", 0)
    torch.manual_seed(7)
    model = trackers.get_model(model_name)().to(dev).train()
    twin = copy.deepcopy(model)
    b0, b1 = [synth.to_torch(synth.make_batch(960 + i, 1), dev) for i in range(2)]
    step = D.DataParallelStep(model, optimizer=torch.optim.SGD(model.parameters(), lr=0.0), world=1, graph=True,
                              graph_warmup=0, require_graph=True)
    step.step(b0)
    assert step.graph is not None, step.graph_error
    loss_g = float(step.step(b1))
    torch.cuda.synchronize()
    grads_g = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    twin.training_loss(b0)
    loss_e, _ = twin.training_loss(b1)
    loss_e.backward()
    assert abs(loss_g - float(loss_e)) <= 1e-5 * (1 + abs(float(loss_e))), (loss_g, float(loss_e))
    top = max(float(p.grad.abs().max()) for p in twin.parameters() if p.grad is not None)
    worst = 0.0
    for k, p in twin.named_parameters():
        if p.grad is None:
            assert k not in grads_g
            continue
        scale = float(p.grad.abs().max())
        if scale < 1e-4 * top:
            assert float(grads_g[k].abs().max()) < 1e-3 * top, k
            continue
        err = float((grads_g[k] - p.grad).abs().max()) / scale
        worst = max(worst, err)
        assert err < 5e-3, (k, err)
    print("%s batch 1, graph replay vs eager: worst per-parameter max-norm gradient difference %.1e" % (model_name, worst))


def test_bat_dense_worst_case_batch48():
    """The worst case of the data-dependent work -- `synth.make_dense_batch`, the clouds `bench.py --dense` and the
    `bat_dense_worst_case` secondary line time: every ball of every set-abstraction level full of DISTINCT neighbours,
    so the distinct-neighbour layout compacts nothing (live fraction 1.0: 1.18 M-column launches at level 0, the
    tile / slice plans and the segment offset `start1` of that size).  48 pairs, 512 / 1024 points, through the fused
    path: sampling indices bit-exact, every end point / loss term / running statistic within 1e-4 of the fp32 CPU
    oracle, every parameter gradient against the oracle's fp64 evaluation under the rule of
    `test_full_size_batch48_gradients_vs_fp64` (max(2e-2, 3 x the fp32 oracle's own distance to fp64 on that key))."""
    from open3dsot_amd import fused, fused_loss, synth
    dev = torch.device("cuda", 0)
    host = synth.make_dense_batch(100, 48)
    model = make_model("BAT", 3)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    batch = synth.to_torch(host, dev)
    out = model(batch)
    data = dict(batch)
    sidx = out["sample_idxs"][:, :out["estimation_cla"].shape[1]].long()
    data["seg_label"] = batch["seg_label"].gather(1, sidx)
    data["points2cc_dist_s"] = batch["points2cc_dist_s"].gather(1, sidx[:, :, None].expand(-1, -1, 9))
    total, ld_fused = fused_loss.track_loss(model.config, data, out, with_bc=True)
    torch.cuda.synchronize()
    ref, ref_ld, sd_after = oracle_forward_loss("BAT", sd, host)
    assert np.array_equal(out["sample_idxs"].cpu().numpy(), ref["sample_idxs"].numpy())
    for k in OUT_KEYS:
        assert rel(out[k], ref[k]) < 1e-4, (k, rel(out[k], ref[k]))
    for k, v in ref_ld.items():
        got = float(total) if k == "total" else float(ld_fused[k])
        assert abs(got - v) <= 1e-4 * (1 + abs(v)), (k, got, v)
    for k, v in model.state_dict().items():
        if "running" in k:
            assert rel(v, sd_after[k]) < 1e-4, (k, rel(v, sd_after[k]))
    # nothing compacts: the backbone's paired levels computed every ball slot
    model.load_state_dict(sd)
    prof = fused.profile_step(lambda: model.training_loss(batch)[0].backward(), 157.3, repeats=1)
    assert prof is not None and prof["live_fraction"] is not None and prof["live_fraction"] > 0.999, prof and prof["live_fraction"]
    _batch48_gradients_vs_fp64("BAT", host, "DENSE B=48")


def test_bat_nuscenes_search_2048():
    """BASELINE config 5 shapes (BAT_CAR_NUSCENES: template 512 / search 2048) at batch 8: forward + losses vs the
    CPU oracle, indices bit-exact; the FPS register kernel holds 32 points per lane at N=2048."""
    from open3dsot_amd import synth
    model = make_model("BAT", 2)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    host = synth.make_batch(500, 8, 512, 2048)
    loss, ld, g = gpu_run(model, sd, host, True, "loss")
    ref_loss, ref_ld, _, _ = oracle_run("BAT", sd, host, torch.float32, "loss")
    assert abs(loss - ref_loss) <= 1e-4 * (1 + abs(ref_loss)), (loss, ref_loss)
    for k in ref_ld:
        assert abs(ld[k] - ref_ld[k]) <= 1e-4 * (1 + abs(ref_ld[k])), k
    assert all(torch.isfinite(v).all() for v in g.values())
    model.eval()
    with torch.no_grad():
        out = model(synth.to_torch(host, torch.device("cuda", 0)))
    ref = torch_ref.bat_forward(sd, synth.to_torch(host), False)
    assert np.array_equal(out["sample_idxs"].cpu().numpy(), ref["sample_idxs"].numpy())


def test_m2track_gpu_matches_cpu_mirror():
    """M2-Track (config 4) on the GPU (flat-GEMM per-point stacks) vs the golden-pinned CPU mirror"""
    from open3dsot_amd import m2track, synth
    torch.manual_seed(5)
    cpu = m2track.M2TRACK().train()
    gpu = m2track.M2TRACK().train()
    gpu.load_state_dict(cpu.state_dict())
    gpu = gpu.cuda()
    host = synth.make_motion_batch(3, 16, 256)
    lc, ldc = cpu.training_loss(synth.to_torch(host))
    lg, ldg = gpu.training_loss(synth.to_torch(host, torch.device("cuda", 0)))
    lg.backward()
    lc.backward()
    for k in ldc:
        assert abs(float(ldg[k]) - float(ldc[k])) <= 2e-3 * (1 + abs(float(ldc[k]))), (k, float(ldg[k]), float(ldc[k]))
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in gpu.parameters())
    # gradients of the fused per-point chains vs the torch CPU mirror: direction and norm of the whole vector
    a = torch.cat([p.grad.detach().cpu().flatten() for p in gpu.parameters()]).double()
    b = torch.cat([p.grad.detach().flatten() for p in cpu.parameters()]).double()
    cos = float(a @ b / (a.norm() * b.norm()))
    assert cos > 0.995 and abs(float(a.norm() / b.norm()) - 1) < 0.03, (cos, float(a.norm() / b.norm()))
    for (k, v), (_, w) in zip(gpu.state_dict().items(), cpu.state_dict().items()):
        if "running" in k:
            assert rel(v, w) < 1e-3, k


def test_m2track_step_is_captured_and_replays_like_the_eager_step():
    """The M2-Track training step must CAPTURE (until round 3 it silently ran eagerly: the class weights of its
    segmentation loss were uploaded from the host inside the step, `operation not permitted when stream is capturing`):
    graph present after the warm-up, replayed losses equal the eager trainer's, every BatchNorm counter advanced once per
    step by the forward's single counter launch."""
    import copy
    from open3dsot_amd import dist as D, m2track, synth
    dev = torch.device("cuda", 0)
    torch.manual_seed(21)
    model_g = m2track.M2TRACK().to(dev).train()
    model_e = copy.deepcopy(model_g)
    tg = D.DataParallelStep(model_g, optimizer=torch.optim.SGD(model_g.parameters(), lr=0.0), world=1, graph=True, graph_warmup=2)
    te = D.DataParallelStep(model_e, optimizer=torch.optim.SGD(model_e.parameters(), lr=0.0), world=1, graph=False)
    batches = [synth.to_torch(synth.make_motion_batch(40 + 8 * i, 8, 512), dev) for i in range(3)]
    for i in range(6):
        lg, le = float(tg.step(batches[i % 3])), float(te.step(batches[i % 3]))
        assert abs(lg - le) <= 2e-4 * (1 + abs(le)), (i, lg, le)
    assert tg.graph is not None, tg.graph_error
    counters = {k: int(v) for k, v in model_g.state_dict().items() if k.endswith("num_batches_tracked")}
    assert len(counters) >= 20 and set(counters.values()) == {6}, counters
    for (k, v), (_, w) in zip(model_g.state_dict().items(), model_e.state_dict().items()):
        if "running" in k:
            assert rel(v, w) < 1e-3, k


@pytest.mark.parametrize("mode", ["act", "gmax"])
@pytest.mark.parametrize("train", [True, False])
@pytest.mark.parametrize("shape", [(4, 256, [14, 64, 128, 64]), (48, 2048, [64, 64, 128, 256])])
def test_fused_pointwise_chain_vs_fp64(mode, train, shape):
    """open3dsot_amd/fused_pointwise.py against an fp64 torch evaluation of Conv1d -> BatchNorm1d -> ReLU (x3)
    [-> global max]: outputs, parameter / input gradients, running statistics.  Second shape: M2-Track's 98 304 columns
    (64- and 128-row layers on the 64-column / split-K tiles, the 256-row layer on 128-column tiles, aligned input: the
    stack's input gradient on the flat entry point too)"""
    from open3dsot_amd import fused_pointwise
    torch.manual_seed(9)
    B, N, widths = shape
    convs = [torch.nn.Conv1d(a, b, 1).cuda() for a, b in zip(widths[:-1], widths[1:])]
    bns = [torch.nn.BatchNorm1d(b).cuda().train(train) for b in widths[1:]]
    with torch.no_grad():
        for bn in bns:
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    x = torch.randn(B, widths[0], N, device="cuda", requires_grad=True)
    x64 = x.detach().double().requires_grad_(True)
    h = x64
    p64 = []
    rms = []
    for conv, bn in zip(convs, bns):
        w, b_, g_, be = (t.detach().double().requires_grad_(True) for t in (conv.weight, conv.bias, bn.weight, bn.bias))
        p64 += [w, b_, g_, be]
        rm, rv = bn.running_mean.double().clone(), bn.running_var.double().clone()
        rms += [rm, rv]
        h = torch.relu(torch.nn.functional.batch_norm(torch.nn.functional.conv1d(h, w, b_), rm, rv, g_, be, train, 0.1, 1e-5))
    ref = h.amax(dim=2) if mode == "gmax" else h
    out = fused_pointwise.chain(x, list(zip(convs, bns)), mode)
    assert rel(out, ref) < 2e-5
    go = torch.randn_like(out)
    out.backward(go)
    ref.backward(go.double())
    got = [t for conv, bn in zip(convs, bns) for t in (conv.weight, conv.bias, bn.weight, bn.bias)]
    gmax = max(float(t.grad.abs().max()) for t in p64)
    for a, b in zip(got, p64):
        err = float((a.grad.detach().cpu().double() - b.grad.cpu()).norm() / (b.grad.norm().cpu() + 1e-3 * gmax * b.numel() ** 0.5))
        assert err < 2e-3, err
    assert float((x.grad.double() - x64.grad).norm() / x64.grad.norm()) < 2e-3
    if train:
        for bn, rm, rv in zip(bns, rms[0::2], rms[1::2]):
            assert rel(bn.running_mean, rm) < 1e-5 and rel(bn.running_var, rv) < 1e-5


@pytest.mark.parametrize("with_bc", [True, False])
@pytest.mark.parametrize("seed", [0, 1])
def test_fused_track_loss_matches_compute_loss(with_bc, seed):
    """csrc/loss.hip (one launch: five losses, weighted total, gradients) against the torch-op restatement of
    models/base_model.py:122-164 + models/bat.py:57-65 evaluated in fp64."""
    from open3dsot_amd import fused_loss, trackers
    g = torch.Generator().manual_seed(seed)
    B, N, P, K = 48, 128, 64, 9
    model = trackers.get_model("BAT" if with_bc else "P2B")()
    box_label = torch.randn(B, 4, generator=g) * 0.5
    centers = box_label[:, None, :3] + torch.randn(B, P, 3, generator=g) * 0.35      # near / masked / far proposals
    out = {"estimation_cla": torch.randn(B, N, generator=g) * 2, "vote_xyz": box_label[:, None, :3] + torch.randn(B, N, 3, generator=g),
           "estimation_boxes": torch.cat([box_label[:, None, :] + torch.randn(B, P, 4, generator=g), torch.randn(B, P, 1, generator=g) * 2], 2),
           "center_xyz": centers, "pred_search_bc": torch.randn(B, N, K, generator=g) * 1.5}
    data = {"seg_label": (torch.rand(B, N, generator=g) < 0.3).float(), "box_label": box_label,
            "points2cc_dist_s": torch.randn(B, N, K, generator=g)}
    if seed == 1:      # empty denominators: no foreground seed, no proposal within 0.3 m
        data["seg_label"].zero_()
        out["center_xyz"] = centers + 5.0
    names = ["estimation_cla", "vote_xyz", "estimation_boxes"] + (["pred_search_bc"] if with_bc else [])
    o64 = {k: v.double().requires_grad_(k in names) for k, v in out.items()}
    d64 = {k: v.double() for k, v in data.items()}
    ld = model.compute_loss(d64, o64)
    c = model.config
    total = (ld["loss_objective"] * c.objectiveness_weight + ld["loss_box"] * c.box_weight + ld["loss_seg"] * c.seg_weight
             + ld["loss_vote"] * c.vote_weight + (ld["loss_bc"] * c.bc_weight if with_bc else 0.0))
    (total * 0.7).backward()
    og = {k: v.cuda().requires_grad_(k in names) for k, v in out.items()}
    dg = {k: v.cuda() for k, v in data.items()}
    tot, parts = fused_loss.track_loss(c, dg, og, with_bc)
    (tot * 0.7).backward()
    assert abs(float(tot) - float(total)) <= 2e-6 * (1 + abs(float(total))), (float(tot), float(total))
    for k, v in ld.items():
        assert abs(float(parts[k]) - float(v)) <= 2e-6 * (1 + abs(float(v))), (k, float(parts[k]), float(v))
    for k in names:
        ref = o64[k].grad
        err = float((og[k].grad.cpu().double() - ref).abs().max())
        assert err <= 2e-6 * float(ref.abs().max()) + 1e-9, (k, err, float(ref.abs().max()))
    assert og["center_xyz"].grad is None


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [(True, True, True, True), (False, False, False, False), (True, False, True, False)])
@pytest.mark.parametrize("seed", [0, 1])
def test_fused_m2_loss_matches_compute_loss_reference(flags, seed):
    """csrc/loss.hip::o3d_m2track_loss (two launches: every term, the weighted total, all gradients) against the term-by-term
    restatement of models/m2track.py:153-231 evaluated in fp64, for the reference's flag combinations (box_aware,
    use_motion_cls, use_second_stage, use_prev_refinement); seed 1: no moving sample (the 1e-6 denominator)."""
    from open3dsot_amd import fused_loss, m2track
    box_aware, use_cls, second, prev_ref = flags
    g = torch.Generator().manual_seed(100 + seed)
    B, N, K = 48, 256, 9
    model = m2track.M2TRACK(box_aware=box_aware, use_motion_cls=use_cls, use_second_stage=second, use_prev_refinement=prev_ref)
    box = torch.randn(B, 4, generator=g)
    out = {"seg_logits": torch.randn(B, 2, N, generator=g) * 2, "pred_bc": torch.randn(B, N, K, generator=g) * 1.5,
           "motion_cls": torch.randn(B, 2, generator=g), "motion_pred": torch.randn(B, 4, generator=g) * 1.5,
           "aux_estimation_boxes": box + torch.randn(B, 4, generator=g) * 0.8,
           "estimation_boxes": box + torch.randn(B, 4, generator=g) * 1.5,
           "estimation_boxes_prev": torch.randn(B, 4, generator=g) * 1.2}
    data = {"seg_label": (torch.rand(B, N, generator=g) < 0.3).long(), "prev_bc": torch.randn(B, N // 2, K, generator=g),
            "this_bc": torch.randn(B, N // 2, K, generator=g), "motion_state_label": (torch.rand(B, generator=g) < 0.5).long(),
            "motion_label": torch.randn(B, 4, generator=g), "box_label": box, "box_label_prev": torch.randn(B, 4, generator=g)}
    if seed == 1:
        data["motion_state_label"].zero_()
    names = ["seg_logits", "motion_pred", "aux_estimation_boxes"] + (["pred_bc"] if box_aware else []) + \
        (["motion_cls"] if use_cls else []) + (["estimation_boxes"] if second else []) + (["estimation_boxes_prev"] if prev_ref else [])
    o64 = {k: v.double().requires_grad_(k in names) for k, v in out.items()}
    d64 = {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}
    ref = model.compute_loss_reference(d64, o64)
    (ref["loss_total"] * 0.7).backward()
    og = {k: v.cuda().requires_grad_(k in names) for k, v in out.items()}
    dg = {k: v.cuda() for k, v in data.items()}
    assert fused_loss.enabled()
    ld = model.compute_loss(dg, og)
    assert set(ld) == set(ref), (sorted(ld), sorted(ref))
    (ld["loss_total"] * 0.7).backward()
    for k, v in ref.items():
        assert ld[k].dim() == 0
        assert abs(float(ld[k]) - float(v)) <= 3e-6 * (1 + abs(float(v))), (k, float(ld[k]), float(v))
    for k in names:
        r = o64[k].grad
        err = float((og[k].grad.cpu().double() - r).abs().max())
        assert err <= 3e-6 * float(r.abs().max()) + 1e-9, (k, err, float(r.abs().max()))
    # the constant-1 seed DataParallelStep uses skips the scaling launch: same gradients as a plain backward
    og2 = {k: v.cuda().requires_grad_(k in names) for k, v in out.items()}
    tot = model.compute_loss(dg, og2)["loss_total"]
    tot.backward(gradient=fused_loss.one(tot.device))
    for k in names:
        assert torch.allclose(og2[k].grad * 0.7, og[k].grad, rtol=1e-6, atol=1e-12), k


def test_backward_seeded_with_the_constant_one_equals_plain_backward():
    """DataParallelStep seeds `loss.backward` with fused_loss.one(device): no `ones_like`, and the fused loss recognises the
    constant and skips its scaling launch.  Same gradients as the plain `loss.backward()` (to run-to-run rounding); any OTHER seed
    (here 0.5 at a different address, and a fresh tensor holding 1.0) still goes through the multiply."""
    from open3dsot_amd import fused_loss, synth
    dev = torch.device("cuda", 0)
    model = make_model("BAT", 12)
    batch = synth.to_torch(synth.make_batch(77, 4, 512, 1024), dev)

    saved = {k: v.clone() for k, v in model.state_dict().items()}

    def grads(seed):
        model.load_state_dict(saved)          # running statistics (the shift of the second moments) as in the first run
        model.zero_grad(set_to_none=True)
        loss, _ = model.training_loss(batch)
        assert loss.dim() == 0 and loss.requires_grad
        loss.backward() if seed is None else loss.backward(gradient=seed)
        return [p.grad.clone() for p in model.parameters()]

    plain = grads(None)
    one = fused_loss.one(dev)
    assert one.dim() == 0 and float(one) == 1.0 and fused_loss.one(dev) is one
    seeded = grads(one)
    fresh = grads(torch.ones((), device=dev))
    half = grads(torch.full((), 0.5, device=dev))
    # not bit for bit: the layer-0 list sums use LDS float atomics (csrc/compact.hip::reduce_gather_kernel), so two runs of
    # the SAME backward differ in the last bits; the seeds must not add anything beyond that
    gmax = max(float(a.abs().max()) for a in plain)     # conv biases in front of a BatchNorm: the true gradient is 0, the computed one noise
    for a, b, c, d in zip(plain, seeded, fresh, half):
        tol = 4e-6 * float(a.abs().max()) + 1e-6 * gmax
        assert float((b - a).abs().max()) <= tol and float((c - a).abs().max()) <= tol
        assert float((d * 2 - a).abs().max()) <= tol


@pytest.mark.parametrize("model_name", ["BAT", "M2TRACK"])
def test_second_backward_under_the_constant_one_seed_returns_the_same_gradients(model_name):
    """Under the constant-1 seed FusedTrackLoss / FusedM2Loss hand their stored gradient tensors to autograd UNCOPIED and
    keep them for a second backward (retain_graph=True).  That is sound only while no consumer modifies its incoming
    gradient in place (round-4 advisor): this test enforces it -- two backward passes over one retained graph must give
    the same parameter gradients (to the run-to-run rounding of the LDS-atomic list sums), and the loss functions' stored
    tensors must be bitwise what they were before either pass."""
    from open3dsot_amd import fused_loss, m2track, synth
    dev = torch.device("cuda", 0)
    if model_name == "BAT":
        model = make_model("BAT", 12)
        batch = synth.to_torch(synth.make_batch(77, 4, 512, 1024), dev)
    else:
        torch.manual_seed(3)
        model = m2track.M2TRACK().to(dev).train()
        batch = synth.to_torch(synth.make_motion_batch(5, 8, 256), dev)
    one = fused_loss.one(dev)
    loss, _ = model.training_loss(batch)
    fn = loss.grad_fn
    assert type(fn).__name__.startswith(("FusedTrackLoss", "FusedM2Loss")), type(fn).__name__
    before = [t.clone() if t is not None else None for t in fn.grads]
    model.zero_grad(set_to_none=True)
    loss.backward(gradient=one, retain_graph=True)
    first = [p.grad.clone() for p in model.parameters()]
    for t, b in zip(fn.grads, before):
        assert t is None or torch.equal(t, b)          # nobody wrote into the handed-out tensors
    model.zero_grad(set_to_none=True)
    loss.backward(gradient=one)
    second = [p.grad.clone() for p in model.parameters()]
    gmax = max(float(a.abs().max()) for a in first)
    for a, b in zip(first, second):
        assert float((b - a).abs().max()) <= 4e-6 * float(a.abs().max()) + 1e-6 * gmax


@pytest.mark.parametrize("model_name", ["BAT", "P2B"])
def test_inference_graph_replay_matches_eager_and_oracle(model_name):
    """the tracking-inference path -- eval mode (BatchNorm on running statistics), batch 1,
    template 512 / search 1024 points, no autograd -- captured as ONE HIP graph and replayed on new frames:
    replay == eager launch-by-launch == fp32 CPU oracle (1e-4), sampling indices equal (models/base_model.py:59-86)."""
    from open3dsot_amd import synth
    dev = torch.device("cuda", 0)
    model = make_model(model_name, 7, train=False)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    hosts = [synth.make_batch(700 + i, 1) for i in range(3)]
    frames = [synth.to_torch(h, dev) for h in hosts]
    keys = [k for k in OUT_KEYS if model_name == "BAT" or k != "pred_search_bc"]

    def fwd(b):
        with torch.no_grad():
            out = model(b)
        return [out[k] for k in keys] + [out["sample_idxs"]]

    eager = [[t.clone() for t in fwd(f)] for f in frames]
    static = {k: v.clone() for k, v in frames[0].items()}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd(static)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = fwd(static)
    for h, f, e in zip(hosts, frames, eager):
        for k, v in f.items():
            static[k].copy_(v)
        g.replay()
        torch.cuda.synchronize()
        for a, b in zip(outs, e):
            assert torch.equal(a, b)                                   # the same kernels on the same data: bitwise
        ref = (torch_ref.bat_forward if model_name == "BAT" else torch_ref.p2b_forward)(sd, synth.to_torch(h), False)
        assert np.array_equal(outs[-1].cpu().numpy(), ref["sample_idxs"].numpy())
        for k, a in zip(keys, outs):
            assert rel(a, ref[k]) < 1e-4, (k, rel(a, ref[k]))
    for k, v in model.state_dict().items():                            # eval mode leaves every buffer untouched
        assert torch.equal(v.cpu(), sd[k]), k


def test_graph_step_equals_eager_step_bookkeeping():
    """DataParallelStep(graph=True): the capture's allocator warm-up pass is not a training step -- after k steps the
    BatchNorm `num_batches_tracked` counters and running statistics equal those of k eager steps on the same batches
    (the template and the search cloud count as two calls of the shared backbone, models/bat.py:89-90), and the losses
    returned by earlier steps are not overwritten by later replays"""
    from open3dsot_amd import dist as D, synth, trackers
    dev = torch.device("cuda", 0)
    batches = [synth.to_torch(synth.make_batch(900 + 4 * i, 4, 256, 512), dev) for i in range(5)]
    results = {}
    for graph in (False, True):
        torch.manual_seed(11)
        model = trackers.BAT().to(dev).train()
        # plain SGD: Adam normalises every step to +-lr, which turns the run-to-run rounding noise of the backward's
        # LDS atomics into diverging trajectories within three steps (measured: 1e-5 -> 2e-3 -> 1e-1 on the loss)
        opt = torch.optim.SGD(model.parameters(), lr=1e-3)
        step = D.DataParallelStep(model, optimizer=opt, world=1, graph=graph, graph_warmup=2)
        losses = [step.step(b) for b in batches]
        torch.cuda.synchronize()
        assert (step.graph is not None) == graph, step.graph_error
        results[graph] = ([float(l) for l in losses], {k: v.detach().clone() for k, v in model.state_dict().items()})
    le, lg = results[False][0], results[True][0]
    assert len(set(lg)) == len(lg)                      # distinct values: nothing was overwritten in place
    # up to and including the capture step the two runs see the same weights: equal losses.  Later steps are not
    # compared: the objectness / box terms threshold the predicted centres (dist < 0.3, > 0.6, base_model.py:140-147),
    # so with 4 pairs a 1e-6 weight difference (atomics in the backward) can flip a label and move the loss by 1 %
    # (eager runs themselves scatter by ~1e-4 here from step 1 on, occasionally more: step 0 is the tight one)
    # (round 3: 0.6 % seen at step 2 on one box -- one flipped objectness label; 2 % bounds a couple of flips, a capture
    # that replayed stale weights or a wrong batch is off by tens of percent; the exact check of a replay against an
    # eager step is test_graph_replay_gradients_equal_eager_gradients, at lr = 0)
    for i, (a, b) in enumerate(zip(le[:3], lg[:3])):
        assert abs(a - b) <= (1e-4 if i == 0 else 2e-2) * (1 + abs(a)), (le, lg)
    assert all(np.isfinite(v) for v in lg)
    for k, v in results[False][1].items():
        w = results[True][1][k]
        if "num_batches_tracked" in k:          # exact: the warm-up pass of the capture did not count as a step
            assert int(v) == int(w), (k, int(v), int(w))
        elif "running" in k:                    # (the two trajectories have diverged by now: only sanity)
            assert torch.isfinite(w).all(), k


def test_graph_replay_gradients_equal_eager_gradients():
    """the captured training step replayed on a NEW batch computes what an eager forward + backward on that batch
    computes (same weights: lr = 0): loss and every parameter gradient -- the timed region of bench.py is this replay"""
    import copy
    from open3dsot_amd import dist as D, synth, trackers
    dev = torch.device("cuda", 0)
    torch.manual_seed(5)
    model = trackers.BAT().to(dev).train()
    twin = copy.deepcopy(model)
    b0, b1 = [synth.to_torch(synth.make_batch(950 + 6 * i, 6, 256, 512), dev) for i in range(2)]
    step = D.DataParallelStep(model, optimizer=torch.optim.SGD(model.parameters(), lr=0.0), world=1, graph=True,
                              graph_warmup=0)
    step.step(b0)                       # captures on b0, replays it
    assert step.graph is not None, step.graph_error
    loss_g = float(step.step(b1))       # replay on the new batch
    torch.cuda.synchronize()
    grads_g = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    twin.training_loss(b0)              # the same BatchNorm running statistics (they shift the variance sums) as the
    loss_e, _ = twin.training_loss(b1)  # graph run had when it saw b1
    loss_e.backward()
    assert abs(loss_g - float(loss_e)) <= 1e-5 * (1 + abs(float(loss_e))), (loss_g, float(loss_e))
    worst = 0.0
    top = max(float(p.grad.abs().max()) for p in twin.parameters() if p.grad is not None)
    for k, p in twin.named_parameters():
        if p.grad is None:
            assert k not in grads_g
            continue
        scale = float(p.grad.abs().max())
        if scale < 1e-4 * top:           # e.g. conv_final.bias: every consumer starts with a training-mode BatchNorm,
            assert float(grads_g[k].abs().max()) < 1e-3 * top, k     # so its true gradient is zero (rounding noise)
            continue
        err = float((grads_g[k] - p.grad).abs().max()) / scale
        worst = max(worst, err)
        assert err < 5e-3, (k, err)          # LDS-atomic summation order differs run to run; nothing else may
    print("graph replay vs eager: worst per-parameter max-norm gradient difference %.1e" % worst)


@pytest.mark.parametrize("model_name", ["BAT", "P2B"])
@pytest.mark.parametrize("graph", [False, True])
def test_wgrad_side_branch_equals_inline_launches(model_name, graph):
    """fused.wgrad_branch (round 6): inside DataParallelStep's backward the set-abstraction levels' weight-gradient launches
    (and full groups of the heads' deferred ones) run on a side stream / second branch of the captured graph, joined before
    anything reads a gradient.  Same step with the branch switched off (every launch inline on the launch stream): loss and
    every parameter gradient equal to the run-to-run noise of the backward's LDS atomics; the branch really forked."""
    import copy
    from open3dsot_amd import dist as D, fused, synth, trackers
    dev = torch.device("cuda", 0)
    torch.manual_seed(9)
    model = trackers.get_model(model_name)().to(dev).train()
    twin = copy.deepcopy(model)
    b0, b1 = [synth.to_torch(synth.make_batch(970 + 6 * i, 6, 256, 512), dev) for i in range(2)]
    res = {}
    was = fused._WGRAD_BRANCH["on"]
    for on, m in ((True, model), (False, twin)):
        fused.set_wgrad_branch(on)
        fused._BRANCH["last_launches"] = 0
        try:
            step = D.DataParallelStep(m, optimizer=torch.optim.SGD(m.parameters(), lr=0.0), world=1, graph=graph,
                                      graph_warmup=0, require_graph=graph)
            step.step(b0)
            forks = fused._BRANCH["last_launches"]
            loss = float(step.step(b1))
            torch.cuda.synchronize()
            assert (step.graph is not None) == graph, step.graph_error
        finally:
            fused.set_wgrad_branch(was)
        assert (forks > 0) == on, (on, forks)
        res[on] = (loss, {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    (la, ga), (lb, gb) = res[True], res[False]
    assert abs(la - lb) <= 1e-5 * (1 + abs(lb)), (la, lb)
    assert set(ga) == set(gb)
    top = max(float(v.abs().max()) for v in gb.values())
    worst = 0.0
    for k, want in gb.items():
        scale = float(want.abs().max())
        if scale < 1e-4 * top:
            assert float(ga[k].abs().max()) < 1e-3 * top, k
            continue
        err = float((ga[k] - want).abs().max()) / scale
        worst = max(worst, err)
        assert err < 5e-3, (k, err)
    print("%s graph=%s: side branch vs inline, worst per-parameter max-norm gradient difference %.1e" % (model_name, graph, worst))


def test_dy_written_once_equals_the_two_operand_data_gradient():
    """fused._DY_ONCE (round-6 experiment, off by default): the weight-gradient launch writes the operand it stages,
    dY = A1*dN + w*(A2*Y + A3), and the data gradient loads that one tensor (o3d_mlp_conv_wgrad2_c_dy +
    o3d_mlp_conv_dgrad_c with Y = NULL) instead of rebuilding it from dN and Y: same loss, same gradients (the summation
    order inside the kernels is unchanged; what differs run to run is the LDS-atomic noise of the list sums)"""
    import copy
    from open3dsot_amd import fused, synth, trackers
    dev = torch.device("cuda", 0)
    torch.manual_seed(13)
    model = trackers.BAT().to(dev).train()
    twin = copy.deepcopy(model)
    batch = synth.to_torch(synth.make_batch(990, 6, 256, 512), dev)
    res = {}
    for on, m in ((True, model), (False, twin)):
        fused._DY_ONCE["on"] = on
        try:
            loss, _ = m.training_loss(batch)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            fused._DY_ONCE["on"] = False
        res[on] = (float(loss.detach()), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None})
    (la, ga), (lb, gb) = res[True], res[False]
    assert la == lb, (la, lb)
    top = max(float(v.abs().max()) for v in gb.values())
    worst = 0.0
    for k, want in gb.items():
        scale = float(want.abs().max())
        if scale < 1e-4 * top:
            assert float(ga[k].abs().max()) < 1e-3 * top, k
            continue
        err = float((ga[k] - want).abs().max()) / scale
        worst = max(worst, err)
        assert err < 5e-3, (k, err)
    print("dY written once vs two-operand data gradient: worst per-parameter max-norm gradient difference %.1e" % worst)


@pytest.mark.parametrize("train", [True, False])
def test_segpointnet_cloud_bias_matches_broadcast_concat(train):
    """SegPointNet (models/backbone/pointnet.py:144-204): the pooled feature broadcast to every point and concatenated
    (:188-190) as a per-cloud bias (fused_pointwise.chain_cloud) against the literal concatenation through the same
    fused stack, and against the module evaluated in fp64 on torch ops: outputs, input gradient, every parameter
    gradient, running statistics"""
    import copy
    from open3dsot_amd import backbone, nn_blocks
    torch.manual_seed(21)
    net = backbone.SegPointNet(input_channel=14, per_point_mlp1=[64, 64, 64, 128, 1024],
                               per_point_mlp2=[512, 256, 128, 128], output_size=11).cuda().train(train)
    g = torch.Generator().manual_seed(22)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(torch.empty(m.weight.shape).uniform_(0.5, 1.5, generator=g))
                m.bias.copy_(torch.empty(m.bias.shape).normal_(0, 0.2, generator=g))
                m.running_mean.copy_(torch.empty(m.bias.shape).normal_(0, 0.2, generator=g))
                m.running_var.copy_(torch.empty(m.bias.shape).uniform_(0.5, 1.5, generator=g))
    x = torch.randn(6, 14, 256, device="cuda")
    ct = torch.randn(6, 11, 256, device="cuda")
    runs = {}
    for mode in ("cloud", "concat", "fp64"):
        m = copy.deepcopy(net)
        xi = x.clone().requires_grad_(True)
        if mode == "fp64":
            m, xi = m.double(), x.double().requires_grad_(True)
            nn_blocks.set_flat_pointwise(False)          # the nn.Sequential path of the mirror, torch ops in fp64
        backbone.set_cloud_bias(mode == "cloud")
        try:
            out = m(xi)
            (out * ct.to(out.dtype)).sum().backward()
        finally:
            backbone.set_cloud_bias(True)
            nn_blocks.set_flat_pointwise(True)
        runs[mode] = (out.detach(), xi.grad, {k: p.grad for k, p in m.named_parameters()},
                      {k: b for k, b in m.named_buffers() if b.dtype.is_floating_point})
    for mode in ("cloud", "concat"):
        out, dx, gp, bufs = runs[mode]
        ref_out, ref_dx, ref_gp, ref_bufs = runs["fp64"]
        # gradients: nine ReLU layers and 6 x 1024 global arg-max decisions on 1 536 columns -- a routing flip against
        # fp64 moves a gradient by 1e-3..5e-3 here (measured 1.4e-3 on one path and 4.5e-3 on the
        # other in the same run): both fp32 paths are held to the same robust bounds
        gtol = 2e-2 if train else 5e-4
        assert rel(out, ref_out) < 2e-5, (mode, rel(out, ref_out))
        assert float((dx.double() - ref_dx).norm() / ref_dx.norm()) < gtol, mode
        for k, gq in ref_gp.items():
            if gq is None or float(gq.abs().max()) < 1e-9 * float(ref_dx.abs().max()):
                continue
            err = float((gp[k].double() - gq).norm() / (gq.norm() + 1e-30))
            if float(gq.norm()) < 1e-6 * float(ref_gp["fc.weight"].norm()):      # a bias in front of a training-mode BN
                continue
            assert err < gtol, (mode, k, err)
        if train:
            for k, b in ref_bufs.items():
                assert rel(bufs[k], b) < 1e-5, (mode, k)
    assert rel(runs["cloud"][0], runs["concat"][0]) < 5e-5          # the two formulations of the same layer


def test_flat_adam_matches_torch_adam():
    """open3dsot_amd.optim.FlatAdam (one launch on flat buffers) against torch.optim.Adam with the reference's
    hyper-parameters (models/base_model.py:32-33) over several steps with real gradients, lr schedule included;
    state_dict layouts are interchangeable"""
    import copy
    from open3dsot_amd import optim, synth, trackers
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    a = trackers.BAT().to(dev).train()
    b = copy.deepcopy(a)
    conf = a.configure_optimizers()
    oa, sa = conf["optimizer"], conf["lr_scheduler"]
    assert isinstance(oa, optim.FlatAdam)
    ob = torch.optim.Adam(b.parameters(), lr=1e-3, weight_decay=0, betas=(0.5, 0.999), eps=1e-6)
    sb = torch.optim.lr_scheduler.StepLR(ob, step_size=12, gamma=0.2)
    g = torch.Generator(device=dev).manual_seed(9)
    for it in range(15):
        grads = [torch.randn(p.shape, device=dev, generator=g) * (0.1 if it % 3 else 1e-4) for p in a.parameters()]
        for (p, q, gr) in zip(a.parameters(), b.parameters(), grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        oa.step(); ob.step(); sa.step(); sb.step()
    for (k, p), q in zip(a.named_parameters(), b.parameters()):
        assert rel(p, q) < 2e-5, (k, rel(p, q))      # lerp vs b1*m+(1-b1)*g rounding
    sda, sdb = oa.state_dict(), ob.state_dict()
    assert sda["param_groups"][0]["lr"] == sdb["param_groups"][0]["lr"]
    for i in sdb["state"]:
        assert float(sda["state"][i]["step"]) == float(sdb["state"][i]["step"]) == 15
        assert rel(sda["state"][i]["exp_avg"], sdb["state"][i]["exp_avg"]) < 2e-6
        assert rel(sda["state"][i]["exp_avg_sq"], sdb["state"][i]["exp_avg_sq"]) < 2e-6
    # torch's state loads into FlatAdam and the next step agrees again
    oa.load_state_dict(copy.deepcopy(sdb))
    for (p, q) in zip(a.parameters(), b.parameters()):
        gr = torch.randn(p.shape, device=dev, generator=g)
        p.grad, q.grad = gr.clone(), gr.clone()
    oa.step(); ob.step()
    for (k, p), q in zip(a.named_parameters(), b.parameters()):
        assert rel(p, q) < 2e-5, (k, rel(p, q))      # lerp vs b1*m+(1-b1)*g rounding
    out = a(synth.to_torch(synth.make_batch(5, 2, 256, 512), dev))      # the re-pointed parameters still drive the model
    assert torch.isfinite(out["estimation_boxes"]).all()


def test_flat_adam_behind_the_flat_gradient_exchange():
    """the multi-rank order of DataParallelStep.step on one GPU: backward assigns p.grad, `FlatGrads.gather` packs the
    gradients into the exchange buffer (where the all-reduce would run), `bind_views` re-points every p.grad at its
    slice, then FlatAdam steps -- its job table must follow the re-pointed gradients.  Against torch.optim.Adam fed
    the same gradients, eager steps and a captured step."""
    import copy
    from open3dsot_amd import dist as D, optim, synth, trackers
    dev = torch.device("cuda", 0)
    torch.manual_seed(8)
    model = trackers.BAT().to(dev).train()
    twin = copy.deepcopy(model)
    step = D.DataParallelStep(model, world=1, graph=False)
    assert isinstance(step.optimizer, optim.FlatAdam)
    ob = torch.optim.Adam(twin.parameters(), lr=1e-3, weight_decay=0, betas=(0.5, 0.999), eps=1e-6)
    for it in range(4):
        batch = synth.to_torch(synth.make_batch(700 + 4 * it, 4, 256, 512), dev)
        step._forward_backward(batch)
        for p, q in zip(model.parameters(), twin.parameters()):
            q.grad = p.grad.detach().clone()
        step.grads.gather([p.grad for p in step.grads.params])        # what reduce_gradients does at world > 1 ...
        step.grads.flat.div_(1)                                       # ... around the all-reduce
        step.grads.bind_views()
        step.optimizer.step()
        ob.step()
        for (k, p), q in zip(model.named_parameters(), twin.parameters()):
            assert rel(p, q) < 2e-5, (it, k, rel(p, q))


@pytest.mark.parametrize("graph", [False, True])
def test_train_eval_train_eval_uses_fresh_batchnorm_constants(graph):
    """the reference's fit loop validates every epoch (train -> eval -> train -> eval).  The one-kernel eval path caches
    its BatchNorm constants on the module keyed on version counters; the training kernels and FlatAdam write through raw
    pointers, so they bump those counters explicitly (fused.touch / increment_version) -- also behind a replayed HIP
    graph.  The second eval must equal the layer-wise eval path (which computes its constants afresh) and must differ
    from the first eval."""
    from open3dsot_amd import dist as D, fused, optim, synth
    dev = torch.device("cuda", 0)
    model = make_model("BAT", 12)
    step = D.DataParallelStep(model, world=1, graph=graph, graph_warmup=1)
    assert isinstance(step.optimizer, optim.FlatAdam)
    val = synth.to_torch(synth.make_batch(31, 2, 512, 1024), dev)

    def evaluate():
        model.eval()
        with torch.no_grad():
            out = {k: v.clone() for k, v in model(val).items() if v.dtype.is_floating_point}
        model.train()
        return out
    first = evaluate()
    for it in range(4):
        step.step(synth.to_torch(synth.make_batch(40 + 4 * it, 4, 512, 1024), dev))
    assert (step.graph is not None) == graph, step.graph_error
    second = evaluate()
    fused.set_eval_fused(False)
    try:
        want = evaluate()
    finally:
        fused.set_eval_fused(True)
    for k in want:
        assert rel(second[k], want[k]) < 1e-4, (k, rel(second[k], want[k]))
    assert rel(second["estimation_boxes"], first["estimation_boxes"]) > 1e-3      # four Adam steps moved the model


def test_flat_adam_step_between_forward_and_backward_is_refused():
    """FlatAdam updates the parameters through a raw-pointer kernel; it bumps their version counters so the fused
    operators' saved-parameter guard fires exactly as for torch optimizers"""
    from open3dsot_amd import optim, synth
    dev = torch.device("cuda", 0)
    model = make_model("BAT", 5)
    opt = model.configure_optimizers()["optimizer"]
    assert isinstance(opt, optim.FlatAdam)
    batch = synth.to_torch(synth.make_batch(3, 2, 256, 512), dev)
    loss, _ = model.training_loss(batch)
    loss.backward()
    v0 = [p._version for p in model.parameters()]
    opt.step()
    assert all(p._version > v for p, v in zip(model.parameters(), v0))
    loss, _ = model.training_loss(batch)
    opt.step()                      # a second update before the backward of the new forward
    with pytest.raises(RuntimeError, match="modified in place|modified by an inplace"):
        loss.backward()


def test_flat_adam_detects_rehomed_parameters_and_fresh_torch_state():
    """a second FlatAdam on the same model re-homes the parameters: the first one must refuse to step instead of
    updating memory the model no longer reads; loading a torch.optim.Adam state taken before its first step resets the
    moments"""
    from open3dsot_amd import optim
    dev = torch.device("cuda", 0)
    lin = torch.nn.Linear(8, 4).to(dev)
    a = optim.FlatAdam(lin.parameters(), lr=1e-2)
    for p in lin.parameters():
        p.grad = torch.ones_like(p)
    a.step()
    m_before = a._m.clone()
    assert float(m_before.abs().max()) > 0
    fresh = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in lin.parameters()], lr=1e-2)
    a.load_state_dict(fresh.state_dict())            # empty state: moments and step start over
    assert float(a._m.abs().max()) == 0 and float(a._v.abs().max()) == 0 and float(a._step) == 0
    a.step()
    assert float(a._step) == 1
    b = optim.FlatAdam(lin.parameters(), lr=1e-2)    # re-homes the parameters into b's buffer
    with pytest.raises(RuntimeError, match="no longer lives"):
        a.step()
    b.step()


def test_best_proposal_on_device_matches_the_references_numpy_selection():
    """MatchingBaseModel.evaluate_one_sample (models/base_model.py:44-57): `estimation_box.cpu().numpy()`,
    `[:, 4].argmax()`, `[best, 0:4]` -- here one launch on the device (csrc/heads.hip), numpy's tie rule included, and
    inside the tracked frame's HIP graph"""
    from open3dsot_amd import synth, trackers
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    for B, P in ((1, 64), (5, 64), (3, 100), (2, 1)):
        boxes = torch.randn(B, P, 5, device=dev, generator=g)
        if P >= 64:
            boxes[0, 40, 4] = boxes[0, 7, 4] = 9.0          # a tie: numpy's argmax takes the first maximum
        got, idx = trackers.best_proposal(boxes)
        cpu = boxes.cpu().numpy()
        for b in range(B):
            i = cpu[b][:, 4].argmax()                         # the reference's own two lines (:47-52)
            assert int(idx[b]) == int(i) and np.array_equal(got[b].cpu().numpy(), cpu[b][i, 0:4]), (B, P, b)
    model = make_model("BAT", 3, train=False)
    frame = synth.to_torch(synth.make_batch(9, 1, 512, 1024), dev)
    with torch.no_grad():
        best, idx = model.evaluate_one_sample(frame)
        ref = model(frame)["estimation_boxes"].squeeze(0).cpu().numpy()
    i = ref[:, 4].argmax()
    assert int(idx[0]) == int(i) and np.array_equal(best[0].cpu().numpy(), ref[i, 0:4])
    static = {k: v.clone() for k, v in frame.items()}
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        model.evaluate_one_sample(static)
    torch.cuda.current_stream().wait_stream(side)
    with torch.no_grad(), torch.cuda.graph(gr):
        gbest, gidx = model.evaluate_one_sample(static)
    frame2 = synth.to_torch(synth.make_batch(10, 1, 512, 1024), dev)
    for k, v in frame2.items():
        static[k].copy_(v)
    gr.replay()
    with torch.no_grad():
        ref2 = model(frame2)["estimation_boxes"].squeeze(0).cpu().numpy()
    i2 = ref2[:, 4].argmax()
    assert int(gidx[0]) == int(i2) and np.allclose(gbest[0].cpu().numpy(), ref2[i2, 0:4], rtol=0, atol=1e-6)


def _rccl_worker(rank, world, port, out):
    import os
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    from open3dsot_amd import dist as D, synth, trackers
    r, lr, w = D.init_distributed()
    dev = torch.device("cuda", lr)
    torch.manual_seed(100 + rank)            # different initialisation per rank: the constructor's broadcast must fix it
    model = trackers.BAT().to(dev).train()
    trainer = D.DataParallelStep(model, graph=True, graph_warmup=1)
    losses = []
    for step in range(4):                    # 1 eager step, the capture, 2 replays
        first, n = D.shard_indices(step, r, w, 4)
        losses.append(float(trainer.step(synth.to_torch(synth.make_batch(first, n, 256, 512), dev))))
    torch.cuda.synchronize()
    torch.save({"sd": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "graph": trainer.graph is not None,
                "err": trainer.graph_error, "backend": torch.distributed.get_backend(), "world": torch.distributed.get_world_size(),
                "losses": losses}, os.path.join(out, "rank%d.pt" % rank))
    torch.distributed.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the round-end scaling node has them)")
def test_two_rank_rccl_step_keeps_replicas_identical(tmp_path):
    """one process per GPU over RCCL (main.py:53-64,82's DDP), world size 2, HIP graph + the one-message all-reduce
    (ReduceOp.AVG) + FlatAdam: after four steps on disjoint shards the two replicas hold identical parameters;
    BatchNorm statistics stay per rank (no sync-BN in the reference)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rccl_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0["backend"] == "nccl" and r0["world"] == 2
    assert r0["graph"] and r1["graph"], (r0["err"], r1["err"])
    diff = 0
    for k in r0["sd"]:
        if "running" in k or "num_batches" in k:
            diff += int(not torch.equal(r0["sd"][k], r1["sd"][k]))
            continue
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
    assert diff > 0                          # per-rank statistics really differ (disjoint shards)
    assert all(np.isfinite(v) for v in r0["losses"] + r1["losses"])


def _rccl_world1_worker(rank, port, out):
    import os
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    from open3dsot_amd import dist as D, synth, trackers
    try:
        r, lr, w = D.init_distributed(force=True)       # a one-rank RCCL communicator (watchdog thread and all)
        probe = torch.ones(4, device=torch.device("cuda", lr))
        torch.distributed.all_reduce(probe)              # the communicator really works on this box
        torch.cuda.synchronize()
    except Exception as e:      # environmental (no RCCL transport on this box): reported as a skip, not as a failure of the path
        torch.save({"unavailable": "%s: %s" % (type(e).__name__, e)}, os.path.join(out, "world1.pt"))
        return
    dev = torch.device("cuda", lr)
    res = {"backend": torch.distributed.get_backend(), "world": torch.distributed.get_world_size()}
    torch.manual_seed(100)
    model = trackers.BAT().to(dev).train()
    twin = trackers.BAT().to(dev).train()
    twin.load_state_dict(model.state_dict())
    # exchange=True: the code path of a multi-GPU run -- broadcast, gradient pack INSIDE the captured graph (captured while
    # the process group's watchdog thread is alive: capture_error_mode="thread_local"), all_reduce(AVG) on the flat
    # buffer, p.grad = views of that buffer, FlatAdam on them, the replicas' self-check
    trainer = D.DataParallelStep(model, graph=True, graph_warmup=1, exchange=True, require_graph=True)
    plain = D.DataParallelStep(twin, world=1, graph=True, graph_warmup=1, exchange=False, require_graph=True)
    assert trainer.exchange and not plain.exchange
    batches = [synth.to_torch(synth.make_batch(40 + 4 * i, 4, 256, 512), dev) for i in range(4)]
    losses, plain_losses = [], []
    for i, b in enumerate(batches):                     # 1 eager step, the capture, 2 replays
        nxt = batches[i + 1] if i + 1 < len(batches) else None
        losses.append(float(trainer.step(b, next_batch=nxt)))
        plain_losses.append(float(plain.step(b, next_batch=nxt)))
    torch.cuda.synchronize()
    res["graph"], res["err"] = trainer.graph is not None, trainer.graph_error
    res["losses"], res["plain_losses"] = losses, plain_losses
    # after a replayed step: the exchange buffer holds what the captured backward wrote (world size 1: AVG is the identity)
    res["flat_equals_graph_grads"] = all(
        torch.allclose(v, g, rtol=1e-6, atol=0) for v, g in zip(trainer.grads.views, trainer._static_grads))
    res["grads_are_views"] = all(p.grad is v for p, v in zip(trainer.grads.params, trainer.grads.views))
    res["views_bound_once"] = bool(trainer._views_bound)
    # a caller clearing a SUBSET of the gradients gets them bound again at the next step (round-4 advisor)
    some = trainer.grads.params[5]
    some.grad = None
    trainer.step(batches[0])
    res["rebound_after_partial_clear"] = some.grad is trainer.grads.views[5]
    plain.step(batches[0])
    torch.cuda.synchronize()
    res["self_check"] = D.replica_self_check(model, trainer, 1.0, 4)
    res["param_gap"] = max(float((a.detach() - b.detach()).abs().max()) for a, b in zip(model.parameters(), twin.parameters()))
    res["finite"] = all(bool(torch.isfinite(p).all()) for p in model.parameters())
    torch.save(res, os.path.join(out, "world1.pt"))
    torch.distributed.destroy_process_group()


def test_world_size_one_rccl_drives_the_multi_gpu_step(tmp_path):
    """The sequence a multi-GPU run executes -- HIP-graph replay with the gradient pack inside, one RCCL
    `all_reduce(AVG)` on the flat 5.9 MB buffer, `FlatAdam` on its views, `replica_self_check` -- on ONE GPU: an NCCL (=RCCL)
    process group of world size 1 and `DataParallelStep(exchange=True)`.  Everything but the inter-GPU transport is the
    code `bench.py --gpus 8` runs (main.py:53-64,82's DDP); no scaling is measured or claimed.  Compared with the plain
    single-GPU trainer on the same batches: same losses on the first two steps (later ones diverge chaotically between any
    two fp32 runs: Adam's sign-like first updates), every loss finite, replicas' self-check divergence exactly 0."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.spawn(_rccl_world1_worker, args=(port, str(tmp_path)), nprocs=1, join=False)
    import time
    t0 = time.time()
    while not ctx.join(timeout=5):                       # a hung collective must not hang the whole test tier
        if time.time() - t0 > 600:
            for p in ctx.processes:
                p.terminate()
            pytest.fail("the one-rank RCCL worker did not finish within 600 s")
    r = torch.load(tmp_path / "world1.pt")
    if "unavailable" in r:
        pytest.skip("no working one-rank RCCL process group on this box: " + r["unavailable"])
    assert r["backend"] == "nccl" and r["world"] == 1
    assert r["graph"], r["err"]
    assert r["flat_equals_graph_grads"] and r["grads_are_views"] and r["views_bound_once"] and r["rebound_after_partial_clear"]
    assert r["self_check"].get("max_parameter_divergence") == 0.0 and "self_check_error" not in r["self_check"], r["self_check"]
    assert "all_reduce" in r["self_check"]["gradient_exchange"] and "nccl" in r["self_check"]["gradient_exchange"]
    assert r["finite"] and all(np.isfinite(v) for v in r["losses"])
    # the same losses as the plain trainer while the comparison means something: step 0 (identical parameters) to rounding,
    # step 1 to 1e-3; after that Adam's first updates (m / sqrt(v) = +-1 wherever a gradient is rounding noise) make two
    # fp32 runs diverge chaotically -- measured 2 % and 8 % at step 3 on two boxes, with the LDS-atomic list sums alone
    assert abs(r["losses"][0] - r["plain_losses"][0]) <= 1e-5 * (1 + abs(r["plain_losses"][0])), (r["losses"], r["plain_losses"])
    assert abs(r["losses"][1] - r["plain_losses"][1]) <= 1e-3 * (1 + abs(r["plain_losses"][1])), (r["losses"], r["plain_losses"])
    print("world-size-1 RCCL step: losses %s | plain %s | parameter gap to the plain trainer %.2e" % (
        ["%.5f" % v for v in r["losses"]], ["%.5f" % v for v in r["plain_losses"]], r["param_gap"]))


def test_bat_nuscenes_yaml_batch100():
    """cfgs/BAT_CAR_NUSCENES.yaml:56 trains at 100 pairs per GPU (search 1024 as the YAML says): the column counts of
    the fused path at 2.08x the benchmarked batch (2.46 M slots in SA1) -- forward, every loss term and the sampling
    indices against the CPU oracle, finite gradients"""
    from open3dsot_amd import synth
    model = make_model("BAT", 6)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    host = synth.make_batch(900, 100, 512, 1024)
    loss, ld, g = gpu_run(model, sd, host, True, "loss")
    ref_loss, ref_ld, _, _ = oracle_run("BAT", sd, host, torch.float32, "loss")
    assert abs(loss - ref_loss) <= 1e-4 * (1 + abs(ref_loss)), (loss, ref_loss)
    for k in ref_ld:
        assert abs(ld[k] - ref_ld[k]) <= 1e-4 * (1 + abs(ref_ld[k])), k
    assert all(torch.isfinite(v).all() for v in g.values())


def test_bat_nuscenes_2048_batch100():
    """BASELINE config 5 as ONE workload: search 2048 points (BASELINE.json) at the YAML's per-GPU batch of 100
    (cfgs/BAT_CAR_NUSCENES.yaml:11,56) -- 204 800 search points per launch, 102 400 balls in the search cloud's SA level 0
    (above the 65 536 the compact layout accepted until round 4), 4.1 M worst-case columns in the paired level: forward,
    every loss term and the sampling indices against the CPU oracle, finite gradients, and the distinct-neighbour path
    really taken (not the slot-wise fallback)."""
    from open3dsot_amd import fused, synth
    model = make_model("BAT", 6)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    host = synth.make_batch(1900, 100, 512, 2048)
    seen = []
    orig = fused.sa_pair_sampled        # (round 5: the paired levels enter through the sampling + ball query launch)

    def spy(grouper, mlp, a, b, geo=None):
        outs = orig(grouper, mlp, a, b, geo=geo)
        seen.append((a[2], b[2], outs is not None and "FusedGroupedMLPCompact" in outs[1].grad_fn.name()))
        return outs
    fused.sa_pair_sampled = spy
    try:
        loss, ld, g = gpu_run(model, sd, host, True, "loss")
    finally:
        fused.sa_pair_sampled = orig
    assert (256, 1024, True) in seen, seen          # SA level 0: both clouds in one set of launches, compact layout
    ref_loss, ref_ld, _, _ = oracle_run("BAT", sd, host, torch.float32, "loss")
    assert abs(loss - ref_loss) <= 1e-4 * (1 + abs(ref_loss)), (loss, ref_loss)
    for k in ref_ld:
        assert abs(ld[k] - ref_ld[k]) <= 1e-4 * (1 + abs(ref_ld[k])), k
    assert all(torch.isfinite(v).all() for v in g.values())
    model.eval()
    with torch.no_grad():
        out = model(synth.to_torch(host, torch.device("cuda", 0)))
    ref = torch_ref.bat_forward(sd, synth.to_torch(host), False)
    assert np.array_equal(out["sample_idxs"].cpu().numpy(), ref["sample_idxs"].numpy())


def test_captured_step_survives_other_models_on_the_device():
    """Round 6 regression: the weight-preparation table (fused_heads.WeightPrep: every padded / transposed weight copy of a
    forward in one launch from a DEVICE job table) was one per device.  A second model's first forward rebuilt and freed the
    table that the first model's captured step still launched with, and a model that died left rows pointing at freed
    buffers in the survivor's table -- a replay then read a dangling table (memory access fault in a test that kept two
    trainers alive).  Now every tracker owns its table and tables are never freed while their owner lives.  Here: capture A;
    build, capture and step B; drop a third model that existed while A captured; replay A on a new batch == an eager twin."""
    import copy
    import gc
    from open3dsot_amd import dist as D, synth, trackers
    dev = torch.device("cuda", 0)
    torch.manual_seed(17)
    bystander = trackers.BAT().to(dev).train()
    b0, b1, b2 = [synth.to_torch(synth.make_batch(1200 + 4 * i, 4, 256, 512), dev) for i in range(3)]
    bystander.training_loss(b0)[0].backward()                 # registered its weight copies on the device
    model_a = trackers.BAT().to(dev).train()
    twin = copy.deepcopy(model_a)
    ta = D.DataParallelStep(model_a, optimizer=torch.optim.SGD(model_a.parameters(), lr=0.0), world=1, graph=True,
                            graph_warmup=0, require_graph=True)
    ta.step(b0)
    assert ta.graph is not None
    model_b = trackers.P2B().to(dev).train()
    tb = D.DataParallelStep(model_b, optimizer=torch.optim.SGD(model_b.parameters(), lr=1e-3), world=1, graph=True,
                            graph_warmup=0, require_graph=True)
    for b in (b0, b1, b2):
        tb.step(b)
    del bystander
    gc.collect()
    junk = [torch.full((1 << 20,), float("nan"), device=dev) for _ in range(8)]      # whatever was freed gets overwritten
    torch.cuda.synchronize()
    loss_a = float(ta.step(b1))
    torch.cuda.synchronize()
    twin.training_loss(b0)
    loss_e = float(twin.training_loss(b1)[0].detach())
    assert abs(loss_a - loss_e) <= 1e-5 * (1 + abs(loss_e)), (loss_a, loss_e)
    del junk


@pytest.mark.parametrize("model_name", ["BAT", "P2B"])
def test_geometry_prefetch_equals_the_inline_step(model_name, request):
    """Round 6: beside the farthest-point sampling, EVERYTHING of the backbone that depends on the input coordinates only
    -- per level the centres, both ball queries and the distinct-neighbour layout (fused.pair_geometry) -- is computed for
    the next batch beside the current step and enters the captured step as inputs ("geo<level>.<name>").  Against a trainer
    whose step computes it inline (trackers.set_geometry_prefetch(False)): the same layout arrays bit for bit, the same
    losses step by step, the same gradients (to the LDS-atomic noise of the backward); a batch that was not announced is
    handled on the spot."""
    import copy
    from open3dsot_amd import dist as D, fused, synth, trackers
    dev = torch.device("cuda", 0)
    model_a = make_model(model_name, 9)
    model_b = copy.deepcopy(model_a)
    pool = [synth.to_torch(synth.make_batch(310 + 4 * i, 4, 512, 1024), dev) for i in range(3)]
    monkey_min = trackers._GEOMETRY_PREFETCH["min_batch"]
    trackers._GEOMETRY_PREFETCH["min_batch"] = 1          # (the default leaves batches below 8 alone: host-bound steps,
    trackers._GEOMETRY_PREFETCH["without_fps"] = True     # and a backbone that does not sample: P2B)
    request.addfinalizer(lambda: trackers._GEOMETRY_PREFETCH.update(min_batch=monkey_min, without_fps=False))
    trackers.set_geometry_prefetch(False)
    try:
        ta = D.DataParallelStep(model_a, optimizer=torch.optim.SGD(model_a.parameters(), lr=0.0), world=1, graph=True,
                                graph_warmup=1, require_graph=True)
        la = [float(ta.step(pool[i % 3], next_batch=pool[(i + 1) % 3])) for i in range(5)]
        torch.cuda.synchronize()
        ga = {k: p.grad.detach().clone() for k, p in model_a.named_parameters() if p.grad is not None}
    finally:
        trackers.set_geometry_prefetch(True)
    assert not any(k.startswith("geo") for k in ta._static)
    tb = D.DataParallelStep(model_b, optimizer=torch.optim.SGD(model_b.parameters(), lr=0.0), world=1, graph=True,
                            graph_warmup=1, require_graph=True)
    idx_prefetch = True
    lb = []
    for i in range(5):
        cur, nxt = pool[i % 3], pool[(i + 1) % 3]
        lb.append(float(tb.step(cur, next_batch=nxt)))
        if tb.graph is not None:          # the layout the replay consumed == the layout computed from `cur` on the spot
            torch.cuda.synchronize()
            with torch.no_grad():
                want = model_b.sampling_inputs(cur)
            assert sum(k.startswith("geo") for k in want) == 3 * len(fused.GEO_KEYS)
            for k, v in want.items():
                if k.endswith((".gp", ".cball", ".cw")):          # (worst-case buffers: only the live columns are defined)
                    live = int(want[k.rsplit(".", 1)[0] + ".meta"].view(-1)[0])
                    assert torch.equal(tb._static[k][:live], v[:live]), (i, k)
                elif k.endswith(".ball_off"):                       # (nballs + 1 slots; the kernels define the first nballs)
                    assert torch.equal(tb._static[k][:-1], v[:-1]), (i, k)
                else:
                    assert torch.equal(tb._static[k], v), (i, k)
    torch.cuda.synchronize()
    gb = {k: p.grad.detach().clone() for k, p in model_b.named_parameters() if p.grad is not None}
    for a, b in zip(la, lb):
        assert abs(a - b) <= 1e-5 * (1 + abs(a)), (la, lb)
    top = max(float(v.abs().max()) for v in ga.values())
    for k, want in ga.items():
        scale = float(want.abs().max())
        if scale < 1e-4 * top:
            continue
        assert float((gb[k] - want).abs().max()) / scale < 5e-3, k
    if idx_prefetch:
        # batches laid out like the captured step's inputs (DataParallelStep.make_batch, what bench.py's loop hands over): the
        # prefetch then writes the geometry straight INTO the next batch's own fields -- no allocation, no copy launch -- and
        # the batch travels into the step as one flat copy: same losses as the per-field route above
        flat_pool = [tb.make_batch(b) for b in pool]
        assert all(isinstance(b, D.FlatBatch) and "geo0.gp" in b.extra_keys for b in flat_pool)
        before = dict(fused._GEO_STATS)
        lf = [float(tb.step(flat_pool[i % 3], next_batch=flat_pool[(i + 1) % 3])) for i in range(5)]
        torch.cuda.synchronize()
        assert fused._GEO_STATS["in_place"] - before["in_place"] >= 3 * 4, (before, fused._GEO_STATS)
        for a, b in zip(la, lf):
            assert abs(a - b) <= 1e-5 * (1 + abs(a)), (la, lf)
    other = synth.to_torch(synth.make_batch(910, 4, 512, 1024), dev)      # not announced: computed on the spot
    l_other = float(tb.step(other))
    if ta._static.extra_keys:      # (BAT: the inline trainer's captured step takes the sampling indices but no geometry inputs;
        with pytest.raises(RuntimeError, match="sampling_inputs"):      # P2B's takes nothing and never asks the model)
            ta.step(other)
    trackers.set_geometry_prefetch(False)
    try:
        assert abs(l_other - float(ta.step(other))) <= 1e-5 * (1 + abs(l_other))
    finally:
        trackers.set_geometry_prefetch(True)


def test_sampling_prefetch_feeds_the_same_indices_as_the_inline_step():
    """DataParallelStep.step(batch, next_batch=...): the farthest-point sampling of the NEXT batch runs on a second
    stream beside the replayed graph of this step and enters the next replay as an input.  The indices the captured step
    consumes must be bit-identical to the operator's own output for that batch (pointnet2_utils.py:37-58), and the
    losses must follow the trajectory of a trainer that samples inside its step."""
    import copy
    from open3dsot_amd import dist as D, ext, synth
    dev = torch.device("cuda", 0)
    model_a = make_model("BAT", 9)
    model_b = copy.deepcopy(model_a)
    pool = [synth.to_torch(synth.make_batch(300 + 4 * i, 4, 512, 1024), dev) for i in range(3)]
    # (learning rate 0: every step's loss is then a function of the batch alone, so the two trainers are comparable step
    # by step -- with Adam the first updates are +-lr whatever the gradient's size, and the run-to-run rounding noise of
    # the scatter sums on near-zero gradients sends two identical trainers apart by percents within three steps)
    ta = D.DataParallelStep(model_a, optimizer=torch.optim.SGD(model_a.parameters(), lr=0.0), world=1, graph=True, graph_warmup=1)
    tb = D.DataParallelStep(model_b, optimizer=torch.optim.SGD(model_b.parameters(), lr=0.0), world=1, graph=True, graph_warmup=1)
    ta._sampling = None                                         # A: sampling inside the captured step (round 2's form)
    assert tb._sampling is not None
    la, lb = [], []
    for i in range(7):
        cur, nxt = pool[i % 3], pool[(i + 1) % 3]
        la.append(float(ta.step(cur)))
        lb.append(float(tb.step(cur, next_batch=nxt)))
        if tb.graph is not None:
            torch.cuda.synchronize()
            want = ext.furthest_point_sampling(cur["search_points"].contiguous(), 512)
            assert torch.equal(tb._static["fps_idx_s"], want), i
            want_t = ext.furthest_point_sampling(cur["template_points"].contiguous(), 256)
            assert torch.equal(tb._static["fps_idx_t"], want_t), i
            if i >= 2:
                assert tb._prefetched is not None and tb._prefetched[0] is nxt
    assert ta.graph is not None and tb.graph is not None, (ta.graph_error, tb.graph_error)
    assert "fps_idx_s" not in ta._static
    for a, b in zip(la, lb):                                     # same losses (scatter sums are not bitwise reproducible)
        assert abs(a - b) <= 1e-5 * (1 + abs(a)), (la, lb)
    assert abs(la[0] - la[3]) <= 1e-5 * (1 + abs(la[0])) and abs(la[0] - la[1]) > 1e-4      # batch 0 again / another batch
    # a batch that was NOT announced: sampled on the spot, same result
    other = synth.to_torch(synth.make_batch(900, 4, 512, 1024), dev)
    tb.step(other)
    torch.cuda.synchronize()
    assert torch.equal(tb._static["fps_idx_s"], ext.furthest_point_sampling(other["search_points"].contiguous(), 512))


def test_flat_batch_single_copy_step_equals_per_field_step():
    """DataParallelStep.make_batch: the inputs of a captured step as views of one flat buffer (one device copy per step);
    same losses as the per-field copies, prefetched sampling indices travelling inside the flat buffer"""
    import copy
    from open3dsot_amd import dist as D, ext, synth
    dev = torch.device("cuda", 0)
    model_a = make_model("BAT", 10)
    model_b = copy.deepcopy(model_a)
    raw = [synth.to_torch(synth.make_batch(500 + 4 * i, 4, 512, 1024), dev) for i in range(3)]
    ta = D.DataParallelStep(model_a, optimizer=torch.optim.SGD(model_a.parameters(), lr=0.0), world=1, graph=True, graph_warmup=1)
    tb = D.DataParallelStep(model_b, optimizer=torch.optim.SGD(model_b.parameters(), lr=0.0), world=1, graph=True, graph_warmup=1)
    for i in range(2):
        ta.step(raw[i]); tb.step(raw[i])
    assert ta.graph is not None and tb.graph is not None
    pool = [tb.make_batch(b) for b in raw]
    assert all(isinstance(b, D.FlatBatch) and b.layout == tb._static.layout for b in pool)
    for i in range(6):
        la = float(ta.step(raw[i % 3]))
        lb = float(tb.step(pool[i % 3], next_batch=pool[(i + 1) % 3]))
        assert abs(la - lb) <= 1e-5 * (1 + abs(la)), (i, la, lb)
        torch.cuda.synchronize()
        assert torch.equal(tb._static["fps_idx_s"], ext.furthest_point_sampling(raw[i % 3]["search_points"].contiguous(), 512))
        assert torch.equal(tb._static["search_points"], raw[i % 3]["search_points"])
    # an eager step on a FlatBatch ignores its (possibly stale) index fields
    pool[0]["fps_idx_s"].zero_()
    l_eager = float(tb._forward_backward(pool[0]))
    assert abs(l_eager - float(ta.step(raw[0]))) <= 1e-5 * (1 + abs(l_eager))


def test_m2track_inference_graph_replay_matches_eager_and_cpu_mirror():
    """M2-Track's tracking-inference forward (SURVEY.md section 8f-4 for config 4): eval mode, batch 1, the stacked 2 x 1024-point
    crops, no autograd, captured as ONE HIP graph and replayed on new frames: replay == eager (bitwise) == the golden-pinned
    CPU mirror of the model (1e-4 of the box scale; the hard masks -- argmax of the segmentation / motion-state logits --
    equal); eval mode leaves every buffer untouched."""
    from open3dsot_amd import m2track, synth
    dev = torch.device("cuda", 0)
    torch.manual_seed(31)
    cpu = m2track.M2TRACK().eval()
    g = torch.Generator().manual_seed(32)
    with torch.no_grad():
        for m in cpu.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(torch.empty(m.weight.shape).uniform_(0.5, 1.5, generator=g))
                m.bias.copy_(torch.empty(m.bias.shape).normal_(0, 0.1, generator=g))
                m.running_mean.copy_(torch.empty(m.bias.shape).normal_(0, 0.1, generator=g))
                m.running_var.copy_(torch.empty(m.bias.shape).uniform_(0.5, 1.5, generator=g))
    gpu = m2track.M2TRACK().eval()
    gpu.load_state_dict(cpu.state_dict())
    gpu = gpu.to(dev)
    sd = {k: v.detach().cpu().clone() for k, v in gpu.state_dict().items()}
    keep = ("points", "candidate_bc")
    hosts = [{k: v for k, v in synth.to_torch(synth.make_motion_batch(900 + i, 1, 1024)).items() if k in keep} for i in range(3)]
    frames = [{k: v.to(dev) for k, v in h.items()} for h in hosts]
    keys = ("estimation_boxes", "aux_estimation_boxes", "motion_pred", "seg_logits", "motion_cls", "estimation_boxes_prev")

    def fwd(b):
        with torch.no_grad():
            out = gpu(b)
        return [out[k] for k in keys]

    eager = [[t.clone() for t in fwd(f)] for f in frames]
    static = {k: v.clone() for k, v in frames[0].items()}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd(static)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        outs = fwd(static)
    for h, f, e in zip(hosts, frames, eager):
        for k, v in f.items():
            static[k].copy_(v)
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(outs, e):
            assert torch.equal(a, b)
        with torch.no_grad():
            ref = cpu(h)
        got = dict(zip(keys, outs))
        same_mask = bool((got["seg_logits"].argmax(1).cpu() == ref["seg_logits"].argmax(1)).all()) and \
            bool((got["motion_cls"].argmax(1).cpu() == ref["motion_cls"].argmax(1)).all())
        if not same_mask:
            continue               # a logit pair within rounding of a tie flips the hard mask: not comparable downstream
        for k in keys:
            assert rel(got[k], ref[k]) < 1e-4, (k, rel(got[k], ref[k]))
    for k, v in gpu.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k
