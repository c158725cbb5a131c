"""Generate tests/golden/ref_m2track_grad.npz: the GRADIENT of the reference's own M2TRACK training loss, in fp32 and in
fp64, on the two M2-Track fixture batches (8 clouds x 256 points, 48 clouds x 512 points -- the inputs of ref_m2track.npz /
ref_m2track_b48.npz, regenerated from open3dsot_amd/synth.py with the same seeds; the state dict is the one stored in
ref_m2track.npz).

Run from the repo root, only where /root/reference exists:  python tests/golden/make_golden_m2track_grad.py

Reference code executed (read-only, /root/reference): models/m2track.py (M2TRACK.forward :73-151, compute_loss :153-231),
models/backbone/pointnet.py, datasets/points_utils.py -- loaded by tests/golden/make_golden_m2track.py (imported here for its
stubs; its main() is not run, so the three fixtures it writes stay byte-identical).

The fp64 run is the reference model `.double()` on the same inputs with its two hard-mask decisions (`torch.argmax`,
models/m2track.py:95,113) REPLAYED from the fp32 run, exactly as ref_m2track_f64.npz does it: its gradient is the true
gradient of the graph the fp32 run executed.  Stored per parameter key:

    <tag>.grad64.<key>     the fp64 gradient rounded once to float32 -- in full for tensors of <= 40 000 elements, the
                           deterministic sample flat[::stride] (stride = ceil(numel / 32 768)) for the nine larger ones
    <tag>.stride.<key>     that stride (1 = stored in full)
    <tag>.norm64.<key>     L2 norm of the FULL fp64 gradient of the key
    <tag>.ref32err.<key>   || g32 - g64 || / || g64 ||  of the reference's own fp32 gradient over the FULL tensor: the yardstick
    <tag>.gradnorm64, <tag>.ref32err_whole, <tag>.loss32, <tag>.loss64
    <tag>.mask.seg / .motion     the reference's fp32 HARD-MASK decisions (argmax of the segmentation logits per point, bit-packed,
                           and of the motion-state logits per cloud): a run under test REPLAYS them, so that its gradient is the
                           gradient of the same graph (one flipped point gates a different set of points into the second stage)
    <tag>.margin.seg / .motion   |logit_1 - logit_0| of the fp64 run (float16 / float32): a decision of the run under test may only
                           differ from the stored one where this margin is within rounding of zero
    <tag>.relu.<module>    the fp64 run's input z of every ReLU behind a Linear -> BatchNorm1d row of the heads (mini_pointnet[2].
                           features.16/19, box_mlp / final_mlp / motion_mlp / motion_state_mlp .2/.5: (B, C) each) as float32 --
                           the ROUTING of the gradient through these ~100 000 units is discrete; a unit whose |z| is within the
                           forward rounding of the run under test (a BatchNorm over 48 rows amplifies it to ~1e-4) may
                           legitimately be routed the other way, which moves every gradient behind it by percents

Third tag `b48x2048`: the BENCHMARKED M2-Track batch (bench.py `m2track_batch48`: 48 frame pairs x 2 048 points,
synth.make_motion_batch(211, 48, point_sample_size=1024)).  Its inputs are not stored (5.9 MB): the test regenerates them from
open3dsot_amd/synth.py and checks `b48x2048.in_sha256` (SHA-256 over the float32 / int64 bytes of the batch in key order);
every key is stored as a sample of <= 8 192 elements.

tests/test_golden_m2track_gpu.py holds the GPU step to  err(key) <= max(2e-2, 3 x ref32err(key))  and the whole vector to the
same rule -- the bar tests/test_golden_trackers_b8.py sets for BAT / P2B.
"""
import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_m2track as base  # noqa: E402  (installs the stubs, loads the reference modules; does not run main())

from open3dsot_amd import m2track as ours, synth  # noqa: E402

FULL_LIMIT, SAMPLE = 40000, 32768


def reference_model():
    from types import SimpleNamespace
    gold = np.load(os.path.join(HERE, "ref_m2track.npz"))
    torch.manual_seed(77)
    net = base.ref_m2.M2TRACK(SimpleNamespace(**ours.M2_KITTI))
    net.load_state_dict({k[3:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith("sd.")}, strict=True)
    return net


def batch_digest(bt):
    import hashlib
    h = hashlib.sha256()
    for k in sorted(bt):
        h.update(k.encode())
        h.update(np.ascontiguousarray(bt[k]).tobytes())
    return h.digest()


def grads_of(net, batch):
    out = net({k: v.clone() for k, v in batch.items()})
    loss = net.compute_loss(batch, out)["loss_total"]
    loss.backward()
    return float(loss.detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}


def main():
    net = reference_model()
    pu = base.points_utils
    fix = {}
    for tag, bt in (("b8", synth.make_motion_batch(11, 8, point_sample_size=128)),
                    ("b48", synth.make_motion_batch(111, 48, point_sample_size=256)),
                    ("b48x2048", synth.make_motion_batch(211, 48, point_sample_size=1024))):
        full_limit, sample = (FULL_LIMIT, SAMPLE) if tag != "b48x2048" else (8192, 8192)
        if tag == "b48x2048":
            fix[tag + ".in_sha256"] = np.frombuffer(batch_digest(bt), dtype=np.uint8).copy()
        tb = synth.to_torch(bt)
        tb64 = {k: (v.double() if v.dtype.is_floating_point else v) for k, v in tb.items()}
        tape, real_argmax = [], torch.argmax
        torch.argmax = lambda *a, **k: (tape.append(real_argmax(*a, **k)), tape[-1])[1]
        try:
            loss32, g32 = grads_of(copy.deepcopy(net).train(), tb)
        finally:
            torch.argmax = real_argmax
        assert len(tape) == 2 and tape[0].shape == (tb["points"].shape[0], 1, tb["points"].shape[1]) and tape[1].shape[1] == 1
        fix[tag + ".mask.seg"] = np.packbits(tape[0].numpy().astype(np.uint8).reshape(-1))
        fix[tag + ".mask.motion"] = tape[1].numpy().astype(np.int8).reshape(-1)
        margins = []
        flips = []

        def replay(*a, **k):
            mine, theirs = real_argmax(*a, **k), tape.pop(0)
            flips.append(int((mine != theirs).sum()))
            margins.append((a[0].detach().select(1, 1) - a[0].detach().select(1, 0)).abs())
            return theirs
        torch.argmax = replay
        torch.set_default_dtype(torch.float64)          # m2track.py:171 builds the class weights with torch.tensor([...])
        real_rotz = pu.rotz_batch_tensor                # datasets/points_utils.py:379 hard-codes float32
        pu.rotz_batch_tensor = base.rotz_like_input
        n64 = copy.deepcopy(net).double().train()
        hooks = []
        for name, mod in n64.named_modules():       # ReLUs behind the Linear -> BatchNorm1d rows (inputs (B, C))
            if isinstance(mod, torch.nn.ReLU):
                def rec(m, inp, out, _n=name):
                    if inp[0].dim() == 2:
                        fix["%s.relu.%s" % (tag, _n)] = inp[0].detach().numpy().astype(np.float32)
                hooks.append(mod.register_forward_hook(rec))
        try:
            loss64, g64 = grads_of(n64, tb64)
        finally:
            for h in hooks:
                h.remove()
            torch.argmax = real_argmax
            torch.set_default_dtype(torch.float32)
            pu.rotz_batch_tensor = real_rotz
        assert not tape and sum(flips) == 0, flips      # no hard-mask decision near a tie (ref_m2track_f64.npz says the same)
        fix[tag + ".margin.seg"] = margins[0].numpy().astype(np.float16).reshape(-1)
        fix[tag + ".margin.motion"] = margins[1].numpy().astype(np.float32).reshape(-1)
        print("  %s smallest hard-mask margins (fp64 logits): seg %.2e, motion %.2e" % (tag, float(margins[0].min()), float(margins[1].min())))
        assert set(g32) == set(g64) == {k for k, _ in net.named_parameters()}
        gn = sum(float(g.pow(2).sum()) for g in g64.values()) ** 0.5
        yard = {}
        for k, g in g64.items():
            assert g.dtype == torch.float64, k
            stride = 1 if g.numel() <= full_limit else -(-g.numel() // sample)
            fix["%s.grad64.%s" % (tag, k)] = g.flatten()[::stride].numpy().astype(np.float32)
            fix["%s.stride.%s" % (tag, k)] = np.int64(stride)
            fix["%s.norm64.%s" % (tag, k)] = np.float64(float(g.norm()))
            yard[k] = float((g32[k].double() - g).norm() / (g.norm() + 1e-300))
            fix["%s.ref32err.%s" % (tag, k)] = np.float64(yard[k])
        whole = sum(float((g32[k].double() - g64[k]).pow(2).sum()) for k in g64) ** 0.5 / gn
        fix["%s.gradnorm64" % tag], fix["%s.ref32err_whole" % tag] = np.float64(gn), np.float64(whole)
        fix["%s.loss32" % tag], fix["%s.loss64" % tag] = np.float64(loss32), np.float64(loss64)
        live = [v for k, v in yard.items() if float(g64[k].norm()) > 1e-6 * gn]
        print("%s loss32 %.6f loss64 %.6f | |g| %.4f | reference fp32 gradient vs its fp64 truth: whole %.2e, per-key median %.2e "
              "max %.2e (keys above 1e-6 of the whole norm: %d of %d)" % (tag, loss32, loss64, gn, whole, float(np.median(live)),
                                                                        max(live), len(live), len(yard)))
    path = os.path.join(HERE, "ref_m2track_grad.npz")
    np.savez_compressed(path, **fix)
    print("wrote", path, "%.2f MB" % (os.path.getsize(path) / 1e6), len(fix), "arrays")


if __name__ == "__main__":
    main()
