"""GPU parity of the whole hot path: BAT / P2B forward + loss + backward on the MI355X
(HIP operator set; composed and fused execution paths) vs the CPU oracle restatement
(oracle/torch_ref.py, itself pinned to the reference's Python layers by tests/golden).
Indices bit-exact; features / losses / gradients within 1e-4 relative-to-scale (fp32)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref

pytestmark = pytest.mark.gpu


def run_pair(model_name, fused, B=3, M=256, N=512, seed=0, train=True):
    from open3dsot_amd import sa_modules, synth, trackers
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
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for k in sd:
        if sd[k].dtype.is_floating_point and "running" not in k:
            sd[k].requires_grad_(True)
    host = synth.make_batch(300 + seed, B, M, N)
    batch = synth.to_torch(host, dev)
    was = sa_modules.fused_enabled()
    sa_modules.set_fused(fused)
    try:
        if train:
            loss, ld = model.training_loss(batch)
            loss.backward()
        out = model(batch) if not train else None
    finally:
        sa_modules.set_fused(was)
    torch.cuda.synchronize()
    cpu_batch = synth.to_torch(host)
    fwd = torch_ref.bat_forward if model_name == "BAT" else torch_ref.p2b_forward
    ref_out = fwd(sd, cpu_batch, train)
    if not train:
        return model, out, ref_out
    w = {k: v for k, v in vars(model.config).items() if k.endswith("_weight")}
    ref_loss, ref_ld = torch_ref.matching_loss(cpu_batch, ref_out, w, bat=model_name == "BAT")
    ref_loss.backward()
    return model, (loss, ld), (ref_loss, ref_ld, sd)


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("model_name", ["BAT", "P2B"])
@pytest.mark.parametrize("fused", [False, True])
def test_training_step_matches_oracle(model_name, fused):
    model, (loss, ld), (ref_loss, ref_ld, sd) = run_pair(model_name, fused)
    assert abs(float(loss.detach()) - float(ref_loss.detach())) <= 1e-4 * (1 + abs(float(ref_loss))), (float(loss), float(ref_loss))
    for k in ref_ld:
        assert abs(float(ld[k]) - float(ref_ld[k])) <= 1e-4 * (1 + abs(float(ref_ld[k]))), k
    # error of each gradient relative to its own scale, floored at 1e-3 of the largest gradient in
    # the model: parameters whose true gradient is ~0 (a bias in front of a training-mode
    # BatchNorm) hold only rounding noise and must not be judged against that noise.
    gmax = max(float(sd[k].grad.abs().max()) for k, _ in model.named_parameters())
    worst, worst_k = 0.0, None
    for k, p in model.named_parameters():
        a, b = p.grad.detach().cpu().double(), sd[k].grad.double()
        r = float((a - b).abs().max() / (b.abs().max() + 1e-3 * gmax))
        if r > worst:
            worst, worst_k = r, k
    assert worst < 2e-3, (worst, worst_k)
    for k, v in model.state_dict().items():   # BatchNorm running statistics advanced identically
        if "running" in k:
            assert rel(v, sd[k]) < 1e-4, k


@pytest.mark.parametrize("fused", [False, True])
def test_eval_forward_matches_oracle(fused):
    model, out, ref = run_pair("BAT", fused, train=False)
    assert np.array_equal(out["sample_idxs"].cpu().numpy(), ref["sample_idxs"].numpy())
    for k in ("estimation_boxes", "estimation_cla", "vote_xyz", "center_xyz", "pred_search_bc"):
        assert rel(out[k], ref[k]) < 1e-4, k
