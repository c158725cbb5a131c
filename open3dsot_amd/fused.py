"""Fused grouped-MLP path: gather + 1x1 conv + BatchNorm + ReLU + max-pool (+ full backward) on the distinct-neighbour
(compact) layout of csrc/compact.hip and the fp32-MFMA kernels of csrc/mlp_direct.hip / csrc/mlp_wgrad.hip, behind the
reference's module boundary.  ONE fused path (round 4): a shape it does not take runs operator by operator (`_composed`).

What it replaces, numerically identical within fp32 rounding (tests/test_fused_gpu.py):
  QueryAndGroup.forward            pointnet2/utils/pointnet2_utils.py:299-339
  SharedMLP (Conv2d+BN2d+ReLU x L) pointnet2/utils/pytorch_utils.py:12-37,68-121
  F.max_pool2d over nsample        pointnet2/utils/pointnet2_modules.py:69-73
  BoxAwareXCorr's group+mlp+max    models/head/xcorr.py:89-100
Parameters stay in the reference's module tree (state_dict keys unchanged); this module only
reads `conv.weight`, `bn.bn.{weight,bias,running_mean,running_var,num_batches_tracked}`.

Data flow per layer (training): GEMM writes the raw conv output Y_l and per-tile partial
statistics; `bn_finalize` turns them into mean/invstd/scale/shift and updates the running
statistics exactly like torch.nn.BatchNorm2d (biased variance to normalise, unbiased into
running_var, momentum 0.1); the NEXT kernel applies BN+ReLU while loading.  Saved for
backward: Y_l and four per-channel vectors per layer, plus arg-max of the pool.
"""
import contextlib
import ctypes

import torch
from torch import nn
from torch.autograd.graph import increment_version as _increment_version

from . import capi
from .ops import QueryAndGroup

_vp, _i, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_double
capi.register("o3d_mlp_conv_fwd", [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp])
capi.register("o3d_mlp_conv_dgrad_wt", [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i,
                                        _vp, _vp, _vp, _vp, _vp, _vp, _vp])
capi.register("o3d_mlp_conv_dgrad_plain", [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp])
capi.register("o3d_mlp_conv_wgrad2", [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp])
capi.register("o3d_mlp_conv_wgrad2_scratch", [_i, _i, _i, _i])
capi.register("o3d_mlp_conv_bwd_fused_rows", [_i, _i, ctypes.c_long])
capi.register("o3d_mlp_conv_bwd_fused_scratch", [_i, _i, ctypes.c_long])
capi.register("o3d_mlp_conv_bwd_fused_c", [_vp] * 10 + [_i, _i, ctypes.c_long, _vp, _vp, ctypes.c_long, _vp, _vp, _vp, _vp, _vp])
capi.register("o3d_bn_finalize", [_vp, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp])
capi.register("o3d_bn_relu_maxpool_fwd", [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp])
capi.register("o3d_bn_bwd_finalize", [_vp, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
capi.register("o3d_mlp_conv_wgrad", [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _i, _i, _i, _f, _i, _i, _i, _i, _i, _vp, _vp, _vp])

_l = ctypes.c_long
capi.register("o3d_compact_build", [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
capi.register("o3d_group_expand_c3", [_vp, _l, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _l, _l, _vp, _vp, _vp, _vp])
capi.register("o3d_group_dw0_xyz", [_vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _l, _i, _vp, _vp, _vp])
capi.register("o3d_compact_build2", [_vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
capi.register("o3d_group_expand_c", [_vp, _l, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _l, _l, _vp, _vp, _vp, _vp])
POOL_BWD_SPLIT = 8      # O3D_POOL_BWD_SPLIT of include/o3dsot.h
capi.register("o3d_bn_finalize_c2", [_vp, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp])
capi.register("o3d_bn_bwd_finalize_c2", [_vp, _i, _i, _i, _d, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp])
capi.register("o3d_group_reduce_gather_scratch", [_i, _i, _i, _i, _i, _i, _i])
capi.register("o3d_group_reduce_gather", [_vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp,
                                          _vp, _vp, _vp, _vp])
capi.register("o3d_pack_points", [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp])
capi.register("o3d_center_term", [_vp, _vp, _i, _i, _i, _vp, _vp])
capi.register("o3d_center_grad", [_vp, _vp, _i, _i, _i, _f, _vp, _vp])
capi.register("o3d_center_term_out", [_vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp])
capi.register("o3d_pool_fwd_c", [_vp, _l, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp])
capi.register("o3d_pool_fwd_ct", [_vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp])
capi.register("o3d_pool_bwd_c", [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _l, _l, _vp, _vp, _vp])
capi.register("o3d_pool_bwd_dense", [_vp, _l, _l, _vp, _l, _l] + [_vp] * 7 + [_l, _l, _i, _i, _i, _i, _vp, _vp, _vp])
capi.register("o3d_sa_eval_fused", [_vp, _l, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _l, _vp, _vp])
capi.register("o3d_group_reduce_c", [_vp, _vp, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp,
                                     _vp, _vp])
capi.register("o3d_direct_tile", [ctypes.c_long, _i, _i])
capi.register("o3d_bn_eval_consts", [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp])
capi.register("o3d_mlp_conv_fwd_c", [_vp, _vp, _vp, _vp, _i, _i, _l, _vp, _vp, _l, _i, _vp, _vp, _vp, _vp])
capi.register("o3d_mlp_conv_dgrad_c", [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, _vp, _vp, _l, _i, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp])
capi.register("o3d_mlp_conv_wgrad2_c", [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, _vp, _vp, _l, _vp, _vp, _vp])
capi.register("o3d_mlp_conv_wgrad2_c_dy", [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, _vp, _vp, _l, _vp, _vp, _vp, _vp])
capi.register("o3d_bn_finalize_c", [_vp, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _i, _vp])
capi.register("o3d_bn_bwd_finalize_c", [_vp, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp])

TILE = 128   # point columns are padded to this (the widest wave tile of the GEMM kernels, csrc/mlp_direct.hip)
ETILE = 256  # columns per workgroup tile of the layer-0 expand kernel (csrc/compact.hip expand_c_kernel)

# ---- optional per-kernel timing (bench.py's roofline leg) ---------------------------------
_PROF = {"on": False, "events": []}


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _call(name, flops, fn, *args, dims=None):
    """Launch one C-ABI entry; when profiling, bracket it with HIP events on the launch stream.
    dims = (Cin, Cout[, pooled]) of a GEMM launch: only used for the per-launch roofline table of `profile_step`."""
    if _PROF["on"]:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(*args)
        e1.record()
        _PROF["events"].append((name, flops, e0, e1, dims))
    else:
        rc = fn(*args)
    capi.check(rc, name)


def _eval_consts(lib, bn, gamma, beta, vec, nrep, st, conv_bias=None):
    """eval-mode BatchNorm -> vec (4, nrep, C) = mean, invstd, scale, shift (one launch, csrc/heads.hip)"""
    _call("bn_eval_consts", 0.0, lib.o3d_bn_eval_consts, bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
          gamma.data_ptr(), beta.data_ptr(), _ptr(conv_bias), float(bn.eps), bn.running_mean.numel(), nrep, vec.data_ptr(), st)


# BatchNorm `num_batches_tracked` increments: inside a tracker forward (fused_heads.prep_scope) they are collected and
# applied by ONE multi-tensor launch at the end of the forward instead of one launch per fused operator
_COUNTERS = {"pending": None}
# running-mean corrections `running_mean += momentum * conv_bias` of the per-point stacks (their statistics are taken without the
# bias): collected the same way, one multi-tensor launch per forward (round 5; was one per stack)
_BIASFIX = {"pending": None}


def bias_fix(momentum, running_means, biases):
    """running_mean_i += momentum * bias_i, now or -- inside a tracker forward -- with every other stack's at its end"""
    if not running_means:
        return
    if _BIASFIX["pending"] is None:
        torch._foreach_add_(running_means, biases, alpha=momentum)
        return
    ent = _BIASFIX["pending"].setdefault(float(momentum), ([], []))
    ent[0].extend(running_means)
    ent[1].extend(biases)


def touch(tensors):
    """bump the autograd version counters of tensors a raw-pointer kernel has just written (running statistics in the
    finalize kernels, parameters in FlatAdam): everything keyed on `_version` -- the eval-mode constant cache below,
    the forward/backward parameter guards -- then sees those writes like any in-place torch op (host only, no launch)"""
    tensors = [t for t in tensors if t is not None]
    if tensors:
        _increment_version(tensors)


def count_batches(bns, inc=1, raw_writes=True):
    """a training forward through `bns` has happened: its finalize kernels wrote the running statistics through raw
    pointers (version counters bumped here) and `num_batches_tracked` is due `inc` increments.  raw_writes=False: the
    statistics were updated by a torch op that did its own version bookkeeping (nn_blocks.RowBatchNorm1d) -- a second bump
    would trip autograd's saved-tensor check of that op's backward"""
    if raw_writes:
        touch([t for bn in bns for t in (bn.running_mean, bn.running_var)])
    tensors = [bn.num_batches_tracked for bn in bns if bn.num_batches_tracked is not None]
    if not tensors:
        return
    if _COUNTERS["pending"] is None:
        torch._foreach_add_(tensors, inc)
    else:
        _COUNTERS["pending"].setdefault(inc, []).extend(tensors)


def counters_begin():
    if _COUNTERS["pending"] is not None:
        return False
    _COUNTERS["pending"] = {}
    _BIASFIX["pending"] = {}
    return True


def counters_end():
    fixes, _BIASFIX["pending"] = _BIASFIX["pending"], None
    for mom, (rms, bs) in (fixes or {}).items():
        torch._foreach_add_(rms, bs, alpha=mom)
    pending, _COUNTERS["pending"] = _COUNTERS["pending"], None
    if not pending:
        return
    # one multi-tensor launch for every increment of the forward (was: one per distinct increment -- 1 for a module
    # called once, 2 for a paired one); a counter that was collected twice gets the sum, not two racing updates
    total = {}
    for inc, tensors in pending.items():
        for t in tensors:
            ent = total.setdefault(id(t), [t, 0])
            ent[1] += inc
    torch._foreach_add_([e[0] for e in total.values()], [e[1] for e in total.values()])


def _versions(params):
    """[(tensor, version)]: the fused functions keep detached views of the live parameter storage for their backward
    (no copy), so autograd's own in-place check does not see them -- `_check_versions` restores it"""
    return [(p, p._version) for p in params if isinstance(p, torch.Tensor)]


def _check_versions(saved, what):
    for p, v in saved:
        if p._version != v:
            raise RuntimeError("%s: a parameter was modified in place between forward and backward (its saved view "
                               "would silently yield wrong gradients)" % what)


GEMM_KERNELS = ("conv_fwd", "conv_fwd_points", "conv_dgrad", "conv_dgrad_points", "conv_wgrad", "conv_wgrad_points", "conv_bwd_fused",
                "pw_conv_fwd", "pw_conv_dgrad", "pw_conv_wgrad")
# The device kernels behind those launch names -- the ONE definition of the "GEMM family" that bench.py's roofline, the PMC
# traffic accounting (tools/hbm_traffic.py) and the SQ-counter table (tools/sq_summary.py) share.  Matched as
# "<name>(" or "<name><" against the demangled kernel name, so `direct_gemm_kernel` does not swallow (or, as in round 3,
# silently miss) `direct_gemm_pair_kernel`.  GEMM_REDUCE_SYMBOLS: the slice reductions of the weight gradients -- their
# bytes belong to the family, they are not counted as launches.
GEMM_KERNEL_SYMBOLS = ("direct_gemm_kernel", "direct_gemm_tail_kernel", "direct_gemm_pair_kernel", "splitk_gemm_kernel", "splitk_gemm_pair_kernel", "wgrad2_kernel", "wgrad2_group_kernel", "fused_bwd_kernel",
                       "conv_fwd_kernel", "conv_dgrad_kernel", "conv_wgrad_kernel")
GEMM_REDUCE_SYMBOLS = ("wgrad_reduce_kernel", "wgrad_reduce1_kernel", "wgrad_reduce_group_kernel")


def kernel_symbol(demangled):
    """'void (anonymous namespace)::wgrad2_kernel<128, 64, 0>(...)' -> 'wgrad2_kernel' (None when it is not a kernel name)"""
    import re
    m = re.search(r"(\w+)\s*(<[^()]*>)?\s*\(", demangled.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else None


def profile_step(step_fn, peak_tflops, repeats=3):
    """Run `step_fn` with every fused launch bracketed by HIP events (on the launch stream); returns
    the roofline dict of the dominant kernel family: the fp32-MFMA GEMM kernels (forward / data
    gradient / weight gradient, on positions and on points).  `achieved` counts the FLOPs those
    launches EXECUTE (layer 0 runs on the N points, not on the npoint*nsample positions)."""
    torch.cuda.synchronize()
    _PROF["events"] = []
    _PROF["on"] = True
    try:
        for _ in range(repeats):
            step_fn()
        torch.cuda.synchronize()
    finally:
        _PROF["on"] = False
    agg = {}
    live_cols = slot_cols = 0.0
    launches_tab = {}
    seq = {}
    for name, flops, e0, e1, dims in _PROF["events"]:
        if isinstance(flops, tuple):       # (per-column FLOPs, meta, slots): the live column count is data dependent
            live = float(flops[1].view(-1, 4)[:, 0].sum().item())      # live columns of every segment
            if name == "conv_fwd":
                live_cols += live
                slot_cols += float(flops[2])
            flops = flops[0] * live
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += flops
        ms_l = e0.elapsed_time(e1)
        a[2] += ms_l
        if dims is not None and flops > 0:          # per-launch roofline row, keyed by (kind, position in the step)
            i = seq[name] = seq.get(name, -1) + 1
            row = launches_tab.setdefault((name, i % max(1, _launches_per_repeat(_PROF["events"], name, repeats))),
                                          [dims, flops, 0.0, 0])
            row[2] += ms_l
            row[3] += 1
    _PROF["events"] = []
    gemm = {k: v for k, v in agg.items() if k in GEMM_KERNELS}
    if not gemm:
        return None
    launches = sum(v[0] for v in gemm.values())
    flops = sum(v[1] for v in gemm.values())
    ms = sum(v[2] for v in gemm.values())
    ach = flops / (ms * 1e-3) / 1e12
    # the set-abstraction / xcorr grouped MLPs alone (round 1's GEMM family, before the heads' launch-latency-bound
    # 1-D conv stacks joined it): comparable across rounds
    sa = {k: v for k, v in gemm.items() if not k.startswith("pw_")}
    sa_ms, sa_fl = sum(v[2] for v in sa.values()), sum(v[1] for v in sa.values())
    per_kernel = {k: {"launches": v[0] // repeats, "ms_per_step": round(v[2] / repeats, 4),
                      "tflops": round(v[1] / (v[2] * 1e-3) / 1e12, 2) if v[2] > 0 and v[1] > 0 else None}
                  for k, v in sorted(agg.items())}
    return {"bound": "mfma", "achieved": round(ach, 3), "peak": peak_tflops, "unit": "TFLOP/s",
            "frac": round(ach / peak_tflops, 4), "traffic": None,
            "kernel": "fp32-MFMA grouped-MLP GEMM kernels (direct_gemm_kernel fwd/dgrad of csrc/mlp_direct.hip, "
                      "wgrad2_kernel of csrc/mlp_wgrad.hip, conv_*_kernel of csrc/mlp.hip for the unaligned "
                      "per-point layer 0); executed FLOPs per launch / HIP-event time of the launch",
            # share of the ball slots (B*npoint*nsample, what the reference's kernels process) that hold a DISTINCT
            # neighbour and are therefore computed here: data dependent -- `value` scales with it (bench.py --dense = 1.0)
            "live_fraction": round(live_cols / slot_cols, 4) if slot_cols else None,
            "launches_per_step": launches // repeats, "avg_launch_ms": round(ms / launches, 5),
            "gemm_ms_per_step": round(ms / repeats, 4), "gemm_gflop_per_step": round(flops / repeats / 1e9, 2),
            "fused_kernels_ms_per_step": round(sum(v[2] for v in agg.values()) / repeats, 4),
            "grouped_mlp_only": {"launches_per_step": sum(v[0] for v in sa.values()) // repeats,
                                 "ms_per_step": round(sa_ms / repeats, 4), "gflop_per_step": round(sa_fl / repeats / 1e9, 2),
                                 "achieved": round(sa_fl / (sa_ms * 1e-3) / 1e12, 3) if sa_ms > 0 else None,
                                 "frac": round(sa_fl / (sa_ms * 1e-3) / 1e12 / peak_tflops, 4) if sa_ms > 0 else None},
            "per_kernel": per_kernel, "per_launch": _per_launch_rows(launches_tab, peak_tflops)}


def _launches_per_repeat(events, name, repeats):
    return sum(1 for e in events if e[0] == name and e[4] is not None) // repeats


HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md


def _per_launch_rows(tab, peak_tflops):
    """one row per GEMM launch of a step: algorithmic FLOPs and bytes (fp32 operands read once, result written once),
    the time both rooflines allow (max of FLOPs / MFMA peak and bytes / HBM peak) and the share of it achieved"""
    rows = []
    for (name, i), (dims, flops, ms, n) in sorted(tab.items()):
        cin, cout = dims[0], dims[1]
        pooled = len(dims) > 2 and dims[2]
        cols = flops / (2.0 * cin * cout)
        if name == "conv_dgrad_points":      # a plain GEMM W0^T . S: S in, the point gradient out
            nrows = cin + cout
        elif "bwd_fused" in name:  # dN, Y, the producer's raw output in; the data gradient out (the weight gradient is small)
            nrows = 2 * cout + 2 * cin
        elif "wgrad" in name:    # dY (+ the raw output its BatchNorm backward needs) and the layer input
            nrows = (cout if pooled else 2 * cout) + cin
        elif "dgrad" in name:    # the same two operands, the result and the producer's raw output for the ReLU mask
            nrows = (cout if pooled else 2 * cout) + 2 * cin
        else:                    # forward: input rows in, output rows out
            nrows = cin + cout
        byts = 4.0 * nrows * cols
        ms1 = ms / n
        t_mfma = flops / (peak_tflops * 1e12) * 1e3
        t_hbm = byts / (HBM_PEAK_GBS * 1e9) * 1e3
        rows.append({"kernel": name, "i": i, "Cin": cin, "Cout": cout, "cols": int(cols), "ms": round(ms1, 4),
                     "tflops": round(flops / ms1 / 1e9, 1), "gbs": round(byts / ms1 / 1e6, 0),
                     "bound": "mfma" if t_mfma >= t_hbm else "hbm", "roof_ms": round(max(t_mfma, t_hbm), 4),
                     "frac": round(max(t_mfma, t_hbm) / ms1, 3)})
    return rows


# ---- module introspection --------------------------------------------------------------
def _layers(mlp):
    """[(conv, bn)] of a SharedMLP whose layers are conv(1x1,no bias) -> BatchNorm2d -> ReLU."""
    out = []
    for layer in mlp.children():
        kids = dict(layer.named_children())
        conv, bnw, act = kids.get("conv"), kids.get("bn"), kids.get("activation")
        if conv is None or bnw is None or not isinstance(act, nn.ReLU) or list(kids)[0] != "conv":
            return None
        bn = getattr(bnw, "bn", None)
        if not isinstance(conv, nn.Conv2d) or not isinstance(bn, nn.BatchNorm2d):
            return None
        if conv.bias is not None or conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.groups != 1:
            return None
        if not bn.affine or not bn.track_running_stats or bn.momentum is None:
            return None
        out.append((conv, bn))
    return out or None


def supports_mlp(mlp):
    layers = _layers(mlp)
    return layers is not None and len(layers) >= 2     # layer 0 runs on the points, layers >= 1 on positions


def supports(grouper, mlp, features):
    return isinstance(grouper, QueryAndGroup) and grouper.use_xyz and supports_mlp(mlp)


def _shape_ok(npoint, ns):
    return 4 <= ns <= 256 and (ns & (ns - 1)) == 0 and (npoint * ns) % ETILE == 0


def _const_vec(dev, n, value):
    key = (str(dev), n, value)
    v = _CONST.get(key)
    if v is None:
        v = _CONST[key] = torch.full((n,), float(value), device=dev, dtype=torch.float32)
    return v


_CONST = {}


# ---- weight gradients on a second branch of the step (round 6) ---------------------------------------------------------
# The weight gradient of a layer is a LEAF of the backward: nothing in the step reads dW_l before the optimizer, while the
# data gradient of the same layer heads the critical chain  dgrad_l -> bn_bwd_finalize_{l-1} -> dgrad_{l-1} -> ...  (a chain of
# launches with wave-quantised tails and ~7 us dependent finalizes during which most of the chip idles).  Inside
# `wgrad_branch()` -- DataParallelStep wraps loss.backward() in it -- the wgrad launches (+ their slice reductions, the layer-0
# per-point weight gradient and its centre term) go to a SIDE stream forked from the launch stream behind the layer's
# bn_bwd_finalize; the scope's end joins it.  Captured into the step's HIP graph the fork/join are graph edges: the wgrad
# nodes form a second branch that the hardware schedules into the main chain's idle slots.
# Soundness = the contract of fused_heads.defer_wgrads (the gradient tensors autograd is handed are FILLED LATER): a parameter
# that already holds a gradient or carries hooks is never branched (`_deferrable`), a parameter met twice in one scope joins
# first, and everything a side launch reads is kept alive in the scope until the join (a tensor freed on the launch stream
# could be handed out again while the side launch still reads it).
# MEASURED AND NOT KEPT AS THE DEFAULT (profiles/r06_ab_wgrad_branch.txt, same-box alternating A/B on the final kernels): the
# branch overlaps 1.07 ms of kernel time per BAT step (sum of kernel durations 6.57 ms against a busy union of 5.11; without
# it 5.50 / 5.20) and the busy time does shrink by 0.09 ms -- the tails are filled -- but every fork / join is a cross-queue
# dependency of the replayed graph (~7-10 us each, 14 forks per step) and the overlapped kernels stretch (the 14 slice
# reductions 0.11 -> 0.36 ms): BAT 5.31 -> 5.39 ms, P2B 8.71 -> 8.79, M2-Track 6.30 -> 6.52.  The mechanism stays as a tested
# switch (tests/test_model_gpu.py::test_wgrad_side_branch_equals_inline_launches), off.
_WGRAD_BRANCH = {"on": False,      # tools/ab_hook.py fused._WGRAD_BRANCH.on flips it for the same-box A/B; tests run both
                 "heads_only": False}   # True (with "on"): only the heads' grouped launches branch (ONE fork per full group)
_BRANCH = {"scope": None, "last_launches": 0}      # last_launches: side-branch forks of the scope that closed last (tests)
_SIDE_STREAMS = {}


def set_wgrad_branch(enabled):
    _WGRAD_BRANCH["on"] = bool(enabled)


def _branch_join(sc):
    if sc["side"] is not None and sc["forked"]:
        sc["main"].wait_stream(sc["side"])
    sc["forked"] = False
    sc["keep"] = []
    sc["keys"] = set()


@contextlib.contextmanager
def wgrad_branch():
    if not _WGRAD_BRANCH["on"] or _BRANCH["scope"] is not None or not torch.cuda.is_available():
        yield
        return
    sc = _BRANCH["scope"] = {"main": None, "side": None, "keep": [], "keys": set(), "forked": False, "launches": 0}
    try:
        yield
    finally:
        _BRANCH["scope"] = None
        _BRANCH["last_launches"] = sc["launches"]
        _branch_join(sc)


def branch_join():
    """the launch stream waits for everything the open scope has put on the side branch (no-op without one)"""
    if _BRANCH["scope"] is not None:
        _branch_join(_BRANCH["scope"])


def _branch_side(params, keep):
    """-> the side stream the weight-gradient launches of `params` may go to (forked behind everything the launch stream
    has been given so far), or None: launch inline.  keep: every tensor those launches read or use as scratch."""
    sc = _BRANCH["scope"]
    if sc is None:
        return None
    from .fused_heads import _deferrable
    params = [p for p in params if p is not None]
    if params and _WGRAD_BRANCH["heads_only"]:      # a set-abstraction level's own weight gradient: inline in this mode
        return None
    if not _deferrable(params):
        return None
    main = torch.cuda.current_stream()
    if sc["main"] is None:
        key = (main.device.index, main.cuda_stream)
        side = _SIDE_STREAMS.get(key)
        if side is None:
            side = _SIDE_STREAMS[key] = torch.cuda.Stream(device=main.device)
        sc["main"], sc["side"] = main, side
    elif main != sc["main"]:
        return None
    keys = {id(p) for p in params}
    if sc["keys"] & keys:      # second use of a parameter: autograd ADDS this gradient to the first as soon as the backward
        _branch_join(sc)       # returns -- the first must be complete, and this one is launched inline
        return None
    sc["keys"] |= keys
    sc["side"].wait_stream(main)
    sc["keep"].append(keep)
    sc["forked"] = True
    sc["launches"] += 1
    return sc["side"]


# ---- the autograd function ---------------------------------------------------------------
class _Cfg:
    __slots__ = ("nxyz", "inv_radius", "training", "bns", "eps", "momentum", "centers", "geo", "geo_dims")


def _direct_tile(lib, pmax, m):
    """columns per wave tile (= per statistics partial row) of the GEMM launches of a level with `pmax` worst-case columns"""
    return lib.o3d_direct_tile(pmax, m, 1)


capi.register("o3d_direct_tail_slots", [_i])
capi.register("o3d_direct_tail_override", [_i])
_TAIL = {"on": True, "applied": -1}       # "on": False = no split (tools/ab_hook.py fused._TAIL.on flips it for the same-box A/B)


def set_tail_split(slots):
    """TEST hook for the remainder-tile split of the large GEMM launches: -1 automatic (default), 0 off, S > 0 pretend S
    resident slots (small problems then meet every branch of csrc/mlp_common.hpp::tail_plan)"""
    capi.load().o3d_direct_tail_override(int(slots))
    _TAIL["on"], _TAIL["applied"] = int(slots) != 0, int(slots)


def _tail_rows(lib, tile, rows, nseg, layer):
    """extra statistics rows behind the regular ones of a 128-column direct GEMM launch over the compact layout with `rows`
    output rows (layers >= 1; layer 0's rows come from the expand kernel): one block of resident-slot size per segment"""
    if not _TAIL["on"] and _TAIL["applied"] != 0:          # switched off through the dict (A/B tool): tell the library
        lib.o3d_direct_tail_override(0)
        _TAIL["applied"] = 0
    return nseg * lib.o3d_direct_tail_slots(rows) if (tile == 128 and layer >= 1) else 0


class FusedGroupedMLPCompact(torch.autograd.Function):
    """QueryAndGroup + SharedMLP + max-pool (+ the whole backward) on the compact layout of csrc/compact.hip: one column per
    DISTINCT neighbour of a ball (ball_query pads with copies of the first hit; copies are folded into
    a weight), flat (C, ldp) activations with the live column counts in device memory.

    `nseg` independent sets of clouds ("segments": the template and the search branch of the backbone,
    models/bat.py:89-90) go through the SAME weights in one set of launches with SEPARATE BatchNorm
    statistics, applied in order like two consecutive module calls.
    apply(cfg, nseg, xyz_0, new_xyz_0, feats_0, idx_0, [xyz_1, ...], W0,g0,b0, W1,...) -> pooled_0 [, pooled_1]"""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, cfg, nseg, *args):
        lib = capi.load()
        segs = [args[4 * s:4 * s + 4] for s in range(nseg)]
        params = args[4 * nseg:]
        L = len(params) // 3
        Ws = [params[3 * l].detach().reshape(params[3 * l].shape[0], -1).contiguous() for l in range(L)]
        gammas = [params[3 * l + 1].detach().contiguous() for l in range(L)]
        betas = [params[3 * l + 2].detach().contiguous() for l in range(L)]
        nxyz = cfg.nxyz
        geo = getattr(cfg, "geo", None)       # the level's coordinate-only part computed ahead of the step (pair_geometry)
        dev = segs[0][3].device if geo is None else geo["gp"].device
        f32, i32 = torch.float32, torch.int32
        st = _stream()
        if geo is None:
            B, _, ns = segs[0][3].shape
            npoints = [sg[3].shape[1] for sg in segs]
        else:
            B, ns, npoints = cfg.geo_dims[0], cfg.geo_dims[1], list(cfg.geo_dims[2])
        C = segs[0][2].shape[1] if segs[0][2] is not None else 0
        Cin0, C0 = nxyz + C, Ws[0].shape[0]
        Ns = [sg[2].shape[2] if sg[2] is not None else sg[0].shape[1] for sg in segs]
        Npads = [-(-n // TILE) * TILE for n in Ns]
        Pmaxs = [B * npt * ns for npt in npoints]
        starts = [0, Pmaxs[0]][:nseg]
        start1 = starts[1] if nseg == 2 else 0
        ldp = sum(Pmaxs)
        pt_bases = [0, B * Npads[0]][:nseg]
        ldz = sum(B * n for n in Npads)
        nballs_s = [B * npt for npt in npoints]
        ball_bases = [0, nballs_s[0]][:nseg]
        nballs = sum(nballs_s)
        need_bwd = any(ctx.needs_input_grad)
        unit = cfg.inv_radius == 1.0          # normalize_xyz False in every tracker config: no scaling launches
        # ---- compaction of the grouping indices (per segment, into joint arrays)
        if geo is not None:
            ball_cnt, ball_off, gp, cball, cw, meta = (geo[k] for k in ("ball_cnt", "ball_off", "gp", "cball", "cw", "meta"))
            assert nseg == 2 and gp.numel() == ldp and ball_cnt.numel() == nballs and meta.numel() == 8
        else:
            ball_cnt = torch.empty((nballs,), device=dev, dtype=i32)
            ball_off = torch.empty((nballs + 1,), device=dev, dtype=i32)
            gp = torch.empty((ldp,), device=dev, dtype=i32)
            cball = torch.empty((ldp,), device=dev, dtype=i32)
            cw = torch.empty((ldp,), device=dev, dtype=f32)
            meta = torch.empty((nseg, 4), device=dev, dtype=i32)
            if nseg == 2:      # both segments: three launches instead of six
                _call("compact_build", 0.0, lib.o3d_compact_build2, segs[0][3].data_ptr(), npoints[0], Npads[0],
                      segs[1][3].data_ptr(), npoints[1], Npads[1], B, ns, starts[1], pt_bases[1], nballs, ball_cnt.data_ptr(),
                      ball_off.data_ptr(), gp.data_ptr(), cball.data_ptr(), cw.data_ptr(), meta.data_ptr(), st)
            else:
                for s_, sg in enumerate(segs):
                    _call("compact_build", 0.0, lib.o3d_compact_build, sg[3].data_ptr(), B, npoints[s_], ns, Npads[s_],
                          starts[s_], pt_bases[s_], ball_bases[s_], nballs, ball_cnt[ball_bases[s_]:].data_ptr(),
                          ball_off[ball_bases[s_]:].data_ptr(), gp.data_ptr(), cball.data_ptr(), cw.data_ptr(),
                          meta[s_].data_ptr(), st)
        # ---- layer 0 on the points: Z = W0 . [xyz * inv_radius ; feats], flat (C0, ldz)
        padded = any(n != npd for n, npd in zip(Ns, Npads))
        # rows padded to a multiple of 16 (zero rows, zero weight columns): the per-point GEMM then runs on the
        # direct MFMA kernel (csrc/mlp_direct.hip needs K % 16 == 0) instead of the generic LDS-staged one
        # (and the buffer itself to a multiple of 64 rows for the weight-gradient kernel of csrc/mlp_wgrad.hip)
        Cin0p = -(-Cin0 // 16) * 16 if Cin0 > 16 else Cin0
        Cin0m = -(-Cin0 // 64) * 64 if Cin0 > 16 else Cin0
        X0n = torch.empty((Cin0m, ldz), device=dev, dtype=f32)
        xs = [sg[0].detach().contiguous() if nxyz else None for sg in segs]
        fs = [sg[2].detach().contiguous() if C else None for sg in segs]
        _call("pack_points", 0.0, lib.o3d_pack_points, _ptr(xs[0]), _ptr(fs[0]), Ns[0], Npads[0],
              _ptr(xs[-1]) if nseg == 2 else None, _ptr(fs[-1]) if nseg == 2 else None, Ns[-1] if nseg == 2 else 0,
              Npads[-1] if nseg == 2 else 0, B, nxyz, C, float(cfg.inv_radius), Cin0m, X0n.data_ptr(), st)
        centers = None
        if nxyz:       # ball centres of every segment + the dummy ball (origin) of the padding columns
            centers = getattr(cfg, "centers", None)         # laid out like that by o3d_sample_query (sa_pair_sampled)
            if centers is None:
                centers = torch.cat([sg[1].detach().reshape(-1, 3) for sg in segs] + [_const_vec(dev, 3, 0.0).view(1, 3)])
            if not unit:
                centers = centers * cfg.inv_radius
        # zero-padded / transposed weight copies come from the device's WeightPrep table (open3dsot_amd/fused_heads.py):
        # inside a tracker forward they were all refreshed by ONE launch, otherwise `get` copies on the spot
        from .fused_heads import prep_for
        prep = prep_for(dev)
        direct3 = C == 0 and nxyz == 3        # xyz-only layer 0: no per-point GEMM, see group_expand below
        Z = None
        if not direct3:
            Z = torch.empty((C0, ldz), device=dev, dtype=f32)
            W0p = prep.get(params[0], C0, Cin0p)
            _call("conv_fwd_points", 2.0 * Cin0p * C0 * ldz, lib.o3d_mlp_conv_fwd, X0n.data_ptr(), W0p.data_ptr(),
                  None, None, 1, Cin0p, C0, ldz, Z.data_ptr(), None, None, st, dims=(Cin0p, C0))
        counts = [float(pm) for pm in Pmaxs]          # BatchNorm counts every slot (copies included)
        Ys, means, invstds, scales, shifts = [], [], [], [], []
        if cfg.training:       # shift of the second moment: the running means before this call, one row per segment
            if nseg == 1:
                statcs = [bn.running_mean.detach().unsqueeze(0) for bn in cfg.bns]
            else:
                # one row per segment, through the device's WeightPrep table (refreshed by the ONE launch at the start of a
                # tracker forward; was a torch.cat per level).  The shift only has to be the SAME vector in the producer's
                # epilogue and in the finalize -- any value near the mean conditions the second moment -- so a snapshot taken
                # at the start of the forward serves both segments
                statcs = [prep.snapshot(bn.running_mean, nseg) for bn in cfg.bns]
        for l in range(L):
            Cout, Cin = Ws[l].shape
            bn = cfg.bns[l]
            tile = ETILE if l == 0 else _direct_tile(lib, ldp, Cout)
            Y = torch.empty((Cout, ldp), device=dev, dtype=f32)
            # (+ the rows of a direct GEMM launch's remainder tiles cut into column blocks: csrc/mlp_common.hpp::tail_plan)
            part = torch.empty((ldp // tile + _tail_rows(lib, tile, Cout, nseg, l), 2, Cout), device=dev, dtype=f32) \
                if cfg.training else None
            statc = statcs[l] if cfg.training else None
            if l == 0 and direct3:
                _call("group_expand", 0.0, lib.o3d_group_expand_c3, X0n.data_ptr(), ldz, gp.data_ptr(), cball.data_ptr(),
                      cw.data_ptr(), centers.data_ptr(), Ws[0].data_ptr(), Cin0, C0, meta.data_ptr(), start1, ldp,
                      Y.data_ptr(), _ptr(part), _ptr(statc), st)
            elif l == 0:
                _call("group_expand", 0.0, lib.o3d_group_expand_c, Z.data_ptr(), ldz, gp.data_ptr(), cball.data_ptr(),
                      cw.data_ptr(), _ptr(centers), Ws[0].data_ptr(), Cin0, C0, meta.data_ptr(), start1, ldp, Y.data_ptr(),
                      _ptr(part), _ptr(statc), st)
            else:
                _call("conv_fwd", (2.0 * Cin * Cout, meta, ldp), lib.o3d_mlp_conv_fwd_c, Ys[-1].data_ptr(), Ws[l].data_ptr(),
                      scales[-1].data_ptr(), shifts[-1].data_ptr(), Cin, Cout, ldp, cw.data_ptr(), meta.data_ptr(), start1,
                      tile, Y.data_ptr(), _ptr(part), _ptr(statc), st, dims=(Cin, Cout))
            vec = torch.empty((4, nseg, Cout), device=dev, dtype=f32)      # mean, invstd, scale, shift per segment
            if cfg.training and nseg == 1:
                _call("bn_finalize", 0.0, lib.o3d_bn_finalize_c, part.data_ptr(), Pmaxs[0] // tile, Cout, counts[0],
                      statc.data_ptr(), gammas[l].data_ptr(), betas[l].data_ptr(), bn.running_mean.data_ptr(),
                      bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps), vec[0].data_ptr(), vec[1].data_ptr(),
                      vec[2].data_ptr(), vec[3].data_ptr(), meta.data_ptr(), tile, st)
            elif cfg.training:       # both segments in one launch; the running statistics see segment 0's update first
                _call("bn_finalize", 0.0, lib.o3d_bn_finalize_c2, part.data_ptr(), Pmaxs[0] // tile, Pmaxs[1] // tile, Cout,
                      counts[0], counts[1], statc.data_ptr(), gammas[l].data_ptr(), betas[l].data_ptr(),
                      bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps),
                      vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(), vec[3].data_ptr(), meta.data_ptr(), tile, st)
            else:
                _eval_consts(lib, bn, gammas[l], betas[l], vec, nseg, st)
            Ys.append(Y)
            means.append(vec[0]); invstds.append(vec[1]); scales.append(vec[2]); shifts.append(vec[3])
        if cfg.training:
            count_batches(cfg.bns, nseg)
        Cl = Ws[-1].shape[0]
        # pooled tensors: one (B, Cl, npoint_s) block per segment in one buffer
        out = torch.empty((nballs * Cl,), device=dev, dtype=f32)
        argq = torch.empty((nballs * Cl,), device=dev, dtype=i32) if need_bwd else None
        yarg = torch.empty((nballs * Cl,), device=dev, dtype=f32) if need_bwd else None
        np1 = npoints[1] if nseg == 2 else 0
        if Cl % 64 == 0 and ns <= 32 and ldp >= 288:
            # lane = channel, a wave per ball over an LDS-transposed tile (csrc/compact.hip::pool_t_kernel)
            _call("pool_fwd", 0.0, lib.o3d_pool_fwd_ct, Ys[-1].data_ptr(), ldp, scales[-1].data_ptr(), shifts[-1].data_ptr(),
                  ball_off.data_ptr(), ball_cnt.data_ptr(), cball.data_ptr(), meta.data_ptr(), start1, B, Cl, npoints[0], np1,
                  ns, out.data_ptr(), _ptr(argq), _ptr(yarg), st)
        else:       # shapes the transposed tile does not take (nsample 64, channels % 64): lanes along the ball's columns
            _call("pool_fwd", 0.0, lib.o3d_pool_fwd_c, Ys[-1].data_ptr(), ldp, scales[-1].data_ptr(), shifts[-1].data_ptr(),
                  ball_off.data_ptr(), ball_cnt.data_ptr(), B, Cl, npoints[0], np1, out.data_ptr(), _ptr(argq), _ptr(yarg), st)
        if need_bwd:
            ctx.cfg = cfg
            ctx.versions = _versions(params)
            ctx.wparams = [params[3 * l] for l in range(L)]       # (identity only: which parameter a weight gradient belongs to)
            # weights as the backward's GEMMs want them: W_l^T for the data gradients, W0^T padded to 64 rows
            ctx.Wts = [None] + [prep.get(params[3 * l], Ws[l].shape[1], Ws[l].shape[0], transpose=True) for l in range(1, L)]
            want_in = any(ctx.needs_input_grad[2:2 + 4 * nseg])
            ctx.W0t = prep.get(params[0], -(-Cin0 // 64) * 64, C0, transpose=True) if want_in else None
            ctx.geom = (B, ns, C, nseg, Ns, Npads, npoints, Pmaxs, starts, pt_bases, ball_bases, nballs_s)
            ctx.saved = (X0n, centers, ball_off, ball_cnt, gp, cball, cw, meta, Ws, gammas, Ys, means, invstds, scales,
                         shifts, out.detach(), argq, yarg)
        outs = tuple(out[ball_bases[s_] * Cl:(ball_bases[s_] + nballs_s[s_]) * Cl].view(B, Cl, npoints[s_])
                     for s_ in range(nseg))
        return outs if nseg > 1 else outs[0]

    @staticmethod
    @capi.on_tensor_device
    def backward(ctx, *dOuts):
        lib = capi.load()
        cfg = ctx.cfg
        _check_versions(ctx.versions, "FusedGroupedMLPCompact")
        B, ns, C, nseg, Ns, Npads, npoints, Pmaxs, starts, pt_bases, ball_bases, nballs_s = ctx.geom
        (X0n, centers, ball_off, ball_cnt, gp, cball, cw, meta, Ws, gammas, Ys, means, invstds, scales, shifts, out,
         argq, yarg) = ctx.saved
        L = len(Ws)
        dev = out.device
        st = _stream()
        f32 = torch.float32
        nxyz = cfg.nxyz
        ldp, ldz, nballs = sum(Pmaxs), X0n.shape[1], sum(nballs_s)
        start1 = starts[1] if nseg == 2 else 0
        counts = [float(pm) for pm in Pmaxs]
        grads = [None] * (3 * L)
        needs = ctx.needs_input_grad[2:2 + 4 * nseg]
        want_xyz = nxyz > 0 and any(needs[4 * s_] or needs[4 * s_ + 1] for s_ in range(nseg))
        want_feats = C > 0 and any(needs[4 * s_ + 2] for s_ in range(nseg))
        Cl = Ws[-1].shape[0]
        np1 = npoints[1] if nseg == 2 else 0
        dN = torch.empty((Cl, ldp), device=dev, dtype=f32)          # dense class-sum gradient of the pooled layer
        # one pass over the live columns (no zero fill; csrc/compact.hip::pool_bwd_dense_kernel), partial rows per 512 columns;
        # the segments' gradients are read where they lie (any batch / channel strides)
        pool_dense = _POOL_BWD_DENSE["on"] and Cl % 8 == 0 and ldp % POOL_DENSE_COLS == 0 and start1 % POOL_DENSE_COLS == 0
        if pool_dense:
            gsrc = []
            for s_ in range(nseg):
                g = dOuts[s_]
                if g is not None and g.stride(2) != 1:
                    g = g.contiguous()
                gsrc += [_ptr(g), g.stride(0) if g is not None else 0, g.stride(1) if g is not None else 0]
            if nseg == 1:
                gsrc += [None, 0, 0]
            part = torch.empty((ldp // POOL_DENSE_COLS, 2, Cl), device=dev, dtype=f32)
            _call("pool_bwd", 0.0, lib.o3d_pool_bwd_dense, *gsrc, out.data_ptr(), argq.data_ptr(), yarg.data_ptr(),
                  means[-1].data_ptr(), cball.data_ptr(), ball_off.data_ptr(), meta.data_ptr(), start1, ldp, B, Cl, npoints[0],
                  np1, dN.data_ptr(), part.data_ptr(), st)
        else:
            # pooled-layer gradient: the segments' (B, Cl, npoint) blocks back to back, like `out`
            if nseg == 1:
                dOut = dOuts[0].contiguous()
            else:
                dOut = torch.empty((nballs * Cl,), device=dev, dtype=f32)
                for s_ in range(nseg):
                    dst = dOut[ball_bases[s_] * Cl:(ball_bases[s_] + nballs_s[s_]) * Cl].view(B, Cl, npoints[s_])
                    if dOuts[s_] is None:
                        dst.zero_()
                    else:
                        dst.copy_(dOuts[s_])
            part = torch.empty((nseg, POOL_BWD_SPLIT, 2, Cl), device=dev, dtype=f32)
            _call("pool_bwd", 0.0, lib.o3d_pool_bwd_c, dOut.data_ptr(), out.data_ptr(), argq.data_ptr(), yarg.data_ptr(),
                  means[-1].data_ptr(), B, Cl, npoints[0], np1, meta.data_ptr(), start1, ldp, dN.data_ptr(),
                  part.data_ptr(), st)
        dtile = POOL_DENSE_COLS if pool_dense else 0
        part_rows = 0    # > 0: `part` comes from the fused data + weight gradient kernel (rows per segment block)
        seg_grads = [[None, None, None] for _ in range(nseg)]
        for l in range(L - 1, -1, -1):
            Cout, Cin = Ws[l].shape
            coef = torch.empty((5, nseg, Cout), device=dev, dtype=f32)  # dgamma dbeta (row 0: all segments) A1 A2 A3
            cp = [coef[k].data_ptr() for k in range(5)]
            if l == L - 1 and not pool_dense:       # partials of the pool backward: POOL_BWD_SPLIT rows per segment, all live
                if nseg == 1:
                    _call("bn_bwd_finalize", 0.0, lib.o3d_bn_bwd_finalize, part.data_ptr(), POOL_BWD_SPLIT, Cout, counts[0],
                          gammas[l].data_ptr(), means[l].data_ptr(), invstds[l].data_ptr(), *cp, None, st)
                else:
                    _call("bn_bwd_finalize", 0.0, lib.o3d_bn_bwd_finalize_c2, part.data_ptr(), POOL_BWD_SPLIT, POOL_BWD_SPLIT,
                          Cout, counts[0], counts[1], gammas[l].data_ptr(), means[l].data_ptr(), invstds[l].data_ptr(), *cp,
                          None, 1, st)
            elif part_rows:      # partials of the fused data + weight gradient kernel: part_rows rows per segment block, all live
                if nseg == 1:
                    _call("bn_bwd_finalize", 0.0, lib.o3d_bn_bwd_finalize, part.data_ptr(), part_rows, Cout, counts[0],
                          gammas[l].data_ptr(), means[l].data_ptr(), invstds[l].data_ptr(), *cp, None, st)
                else:
                    _call("bn_bwd_finalize", 0.0, lib.o3d_bn_bwd_finalize_c2, part.data_ptr(), part_rows, part_rows,
                          Cout, counts[0], counts[1], gammas[l].data_ptr(), means[l].data_ptr(), invstds[l].data_ptr(), *cp,
                          None, 1, st)
            elif nseg == 1:
                _call("bn_bwd_finalize", 0.0, lib.o3d_bn_bwd_finalize_c, part.data_ptr(), Pmaxs[0] // dtile, Cout, counts[0],
                      gammas[l].data_ptr(), means[l].data_ptr(), invstds[l].data_ptr(), *cp, meta.data_ptr(), dtile, st)
            else:
                _call("bn_bwd_finalize", 0.0, lib.o3d_bn_bwd_finalize_c2, part.data_ptr(), Pmaxs[0] // dtile,
                      Pmaxs[1] // dtile, Cout, counts[0], counts[1], gammas[l].data_ptr(), means[l].data_ptr(),
                      invstds[l].data_ptr(), *cp, meta.data_ptr(), dtile, st)
            if not cfg.training:
                coef[3].zero_()
                coef[4].zero_()
            grads[3 * l + 1], grads[3 * l + 2] = coef[0, 0], coef[1, 0]      # summed over the segments by the kernel
            A = (coef[2].data_ptr(), coef[3].data_ptr(), coef[4].data_ptr())
            if l == 0 and C == 0 and nxyz == 3 and not (want_xyz or want_feats):
                # xyz-only layer 0, nobody wants the input gradient (SA level 0): dW0 straight from the columns, no list
                # sums, no K = 3 GEMM, no centre term (csrc/compact.hip::dw0_xyz_kernel)
                part0 = torch.empty((ldp // 256, Cout, 3), device=dev, dtype=f32)
                dW = torch.empty((Cout, 3), device=dev, dtype=f32)
                _call("group_dw0", 0.0, lib.o3d_group_dw0_xyz, dN.data_ptr(), Ys[0].data_ptr(), ldp, A[0], A[1], A[2],
                      gp.data_ptr(), cball.data_ptr(), cw.data_ptr(), X0n.data_ptr(), ldz, centers.data_ptr(), meta.data_ptr(),
                      start1, Cout, part0.data_ptr(), dW.data_ptr(), st)
                grads[0] = dW
                continue
            if l == 0:
                S = torch.empty((Cout, ldz), device=dev, dtype=f32)
                T = torch.empty((Cout, nballs), device=dev, dtype=f32) if nxyz else None
                spanmax = max(npoints) * ns
                npo = lib.o3d_group_reduce_gather_scratch(B, nseg, npoints[0], Npads[0], npoints[-1], Npads[-1],
                                                          spanmax) if _REDUCE_GATHER["on"] else -1
                if npo >= 0:     # the cloud's columns through a transposed index: one LDS atomic per run of equal points
                    perm = torch.empty((ldp,), device=dev, dtype=torch.int32)
                    poff = torch.empty((npo,), device=dev, dtype=torch.int32)
                    _call("group_reduce", 0.0, lib.o3d_group_reduce_gather, dN.data_ptr(), Ys[0].data_ptr(), ldp, A[0], A[1],
                          A[2], gp.data_ptr(), cw.data_ptr(), ball_off.data_ptr(), ball_cnt.data_ptr(), B, nseg, npoints[0],
                          Npads[0], npoints[-1], Npads[-1], Cout, spanmax, perm.data_ptr(), poff.data_ptr(), S.data_ptr(),
                          _ptr(T), st)
                else:            # the index does not fit the LDS budget (o3d_group_reduce_gather_scratch < 0): one LDS atomic per column
                    _call("group_reduce", 0.0, lib.o3d_group_reduce_c, dN.data_ptr(), Ys[0].data_ptr(), ldp, A[0], A[1], A[2],
                          gp.data_ptr(), cball.data_ptr(), cw.data_ptr(), ball_off.data_ptr(), ball_cnt.data_ptr(), B, nseg,
                          npoints[0], Npads[0], npoints[-1], Npads[-1], Cout, S.data_ptr(), _ptr(T), st)
                one, zero = _const_vec(dev, Cout, 1.0), _const_vec(dev, Cout, 0.0)
                Cinm = X0n.shape[0]                  # rows of the padded per-point operand
                aligned = Cinm % 64 == 0
                if aligned:       # the tile-matched MFMA weight-gradient kernel, X as stored
                    wpart = torch.empty((lib.o3d_mlp_conv_wgrad2_scratch(1, Cinm, Cout, ldz),), device=dev, dtype=f32)
                    dWm = torch.empty((Cout, Cinm), device=dev, dtype=f32)
                else:
                    tiles = ((Cin + 127) // 128) * ((Cout + 127) // 128)
                    total_chunks = ldz // 32
                    nsl = max(1, min(total_chunks // 4 if total_chunks >= 4 else 1, 768 // tiles))
                    wpart = torch.empty((nsl + 16, Cout, Cin), device=dev, dtype=f32)
                    dWm = torch.empty((Cout, Cin), device=dev, dtype=f32)
                fold = bool(nxyz) and dWm.shape[1] != Cin
                dW = torch.empty((Cout, Cin), device=dev, dtype=f32) if fold else None
                # dW0 is a leaf: the per-point weight gradient and its centre term on the side branch (the data gradient
                # W0^T . S below, which the next level's backward waits for, stays on the launch stream)
                side = _branch_side([ctx.wparams[0]], (S, T, wpart, dWm, dW, one, zero, ctx.saved))
                st0 = side.cuda_stream if side is not None else st
                with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                    if aligned:
                        _call("conv_wgrad_points", 2.0 * Cinm * Cout * ldz, lib.o3d_mlp_conv_wgrad2, S.data_ptr(), None, 4,
                              S.data_ptr(), one.data_ptr(), zero.data_ptr(), zero.data_ptr(), X0n.data_ptr(), None, None, 1,
                              Cinm, Cout, ldz, wpart.data_ptr(), dWm.data_ptr(), st0, dims=(Cinm, Cout, True))
                    else:
                        _call("conv_wgrad_points", 2.0 * Cin * Cout * ldz, lib.o3d_mlp_conv_wgrad, S.data_ptr(), None, None,
                              None, 4, S.data_ptr(), one.data_ptr(), zero.data_ptr(), zero.data_ptr(), X0n.data_ptr(), None,
                              None, None, None, None, None, 0, 0, 0, 1.0, 1, Cin, Cout, ldz, nsl, wpart.data_ptr(),
                              dWm.data_ptr(), st0)
                    if fold:
                        # the centre term of grouped_xyz = xyz[idx] - new_xyz and the compaction of the padded rows in one
                        # launch (was: the term in place, then a strided torch copy of dWm[:, :Cin])
                        _call("center_term", 0.0, lib.o3d_center_term_out, T.data_ptr(), centers.data_ptr(), Cout, nballs,
                              dWm.shape[1], dWm.data_ptr(), Cin, dW.data_ptr(), st0)
                    else:
                        if nxyz:      # the centre term of grouped_xyz = xyz[idx] - new_xyz
                            _call("center_term", 0.0, lib.o3d_center_term, T.data_ptr(), centers.data_ptr(), Cout, nballs,
                                  dWm.shape[1], dWm.data_ptr(), st0)
                        dW = dWm if dWm.shape[1] == Cin else dWm[:, :Cin].contiguous()
                grads[0] = dW
                if want_xyz or want_feats:
                    # dX = W0^T . S as a plain forward GEMM on the direct MFMA kernel: rows padded to a multiple of 64
                    Cinm = -(-Cin // 64) * 64
                    W0t = ctx.W0t                                                          # (Cinm, Cout), rows >= Cin zero
                    dX = torch.empty((Cinm, ldz), device=dev, dtype=f32)
                    _call("conv_dgrad_points", 2.0 * Cinm * Cout * ldz, lib.o3d_mlp_conv_fwd, S.data_ptr(), W0t.data_ptr(),
                          None, None, 1, Cout, Cinm, ldz, dX.data_ptr(), None, None, st, dims=(Cout, Cinm))
                    dnew_all = None
                    if want_xyz:      # -inv_radius * W0[:, :3]^T . T (3, balls): one small launch (was a torch.addmm on rocBLAS)
                        dnew_all = torch.empty((3, nballs), device=dev, dtype=f32)
                        _call("center_grad", 0.0, lib.o3d_center_grad, T.data_ptr(), Ws[0].data_ptr(), Ws[0].shape[1], Cout, nballs,
                              -float(cfg.inv_radius), dnew_all.data_ptr(), st)
                    for s_ in range(nseg):
                        view = dX[:Cin, pt_bases[s_]:pt_bases[s_] + B * Npads[s_]].view(Cin, B, Npads[s_])
                        if want_feats:
                            seg_grads[s_][2] = view[nxyz:, :, :Ns[s_]].permute(1, 0, 2)
                        if want_xyz:
                            dxyz = view[:3, :, :Ns[s_]].permute(1, 2, 0)
                            seg_grads[s_][0] = dxyz if cfg.inv_radius == 1.0 else dxyz * cfg.inv_radius
                            seg_grads[s_][1] = dnew_all[:, ball_bases[s_]:ball_bases[s_] + nballs_s[s_]].reshape(
                                3, B, npoints[s_]).permute(1, 2, 0)
                continue
            rows_fb = lib.o3d_mlp_conv_bwd_fused_rows(Cin, Cout, ldp) if Cout <= FUSED_BWD_MAX_COUT else -1
            if rows_fb > 0:      # data + weight gradient in one launch (64-channel layers; csrc/mlp_wgrad.hip)
                dW = torch.empty((Cout, Cin), device=dev, dtype=f32)
                wpart = torch.empty((lib.o3d_mlp_conv_bwd_fused_scratch(Cin, Cout, ldp),), device=dev, dtype=f32)
                dNp = torch.empty((Cin, ldp), device=dev, dtype=f32)
                part = torch.empty((2, rows_fb, 2, Cin), device=dev, dtype=f32)
                _call("conv_bwd_fused", (4.0 * Cin * Cout, meta, ldp), lib.o3d_mlp_conv_bwd_fused_c, dN.data_ptr(),
                      Ys[l].data_ptr(), A[0], A[1], A[2], Ys[l - 1].data_ptr(), scales[l - 1].data_ptr(),
                      shifts[l - 1].data_ptr(), means[l - 1].data_ptr(), ctx.Wts[l].data_ptr(), Cin, Cout, ldp, cw.data_ptr(),
                      meta.data_ptr(), start1, wpart.data_ptr(), dW.data_ptr(), part.data_ptr(), dNp.data_ptr(), st,
                      dims=(Cin, Cout))
                grads[3 * l] = dW
                dN, part_rows = dNp, rows_fb
                continue
            part_rows = 0
            flops = (2.0 * Cin * Cout, meta, ldp)      # executed FLOPs = per live column (count read back when profiling)
            dW = torch.empty((Cout, Cin), device=dev, dtype=f32)
            wpart = torch.empty((lib.o3d_mlp_conv_wgrad2_scratch(1, Cin, Cout, ldp),), device=dev, dtype=f32)
            dYm = None
            if _DY_ONCE["on"]:
                # (experiment, review item 2-ii) the weight gradient writes the operand it stages, dY, once; the data gradient
                # below then loads that ONE tensor instead of rebuilding dY from dN and Y
                dYm = torch.empty((Cout, ldp), device=dev, dtype=f32)
                _call("conv_wgrad", flops, lib.o3d_mlp_conv_wgrad2_c_dy, dN.data_ptr(), Ys[l].data_ptr(), A[0], A[1], A[2],
                      Ys[l - 1].data_ptr(), scales[l - 1].data_ptr(), shifts[l - 1].data_ptr(), Cin, Cout, ldp,
                      cw.data_ptr(), meta.data_ptr(), start1, wpart.data_ptr(), dW.data_ptr(), dYm.data_ptr(), st,
                      dims=(Cin, Cout))
            else:
                side = _branch_side([ctx.wparams[l]], (dN, coef, wpart, ctx.saved))
                with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                    _call("conv_wgrad", flops, lib.o3d_mlp_conv_wgrad2_c, dN.data_ptr(), Ys[l].data_ptr(), A[0], A[1], A[2],
                          Ys[l - 1].data_ptr(), scales[l - 1].data_ptr(), shifts[l - 1].data_ptr(), Cin, Cout, ldp,
                          cw.data_ptr(), meta.data_ptr(), start1, wpart.data_ptr(), dW.data_ptr(),
                          side.cuda_stream if side is not None else st, dims=(Cin, Cout))
            grads[3 * l] = dW
            Wt = ctx.Wts[l]
            dNp = torch.empty((Cin, ldp), device=dev, dtype=f32)
            dtile = _direct_tile(lib, ldp, Cin)
            part = torch.empty((ldp // dtile + _tail_rows(lib, dtile, Cin, nseg, 1), 2, Cin), device=dev, dtype=f32)
            _call("conv_dgrad", flops, lib.o3d_mlp_conv_dgrad_c, (dYm if dYm is not None else dN).data_ptr(),
                  None if dYm is not None else Ys[l].data_ptr(), A[0], A[1], A[2],
                  Wt.data_ptr(), Cin, Cout, ldp, cw.data_ptr(), meta.data_ptr(), start1, dtile, Ys[l - 1].data_ptr(),
                  scales[l - 1].data_ptr(), shifts[l - 1].data_ptr(), means[l - 1].data_ptr(), dNp.data_ptr(),
                  part.data_ptr(), st, dims=(Cin, Cout))
            dN = dNp
        gw = []
        for l in range(L):
            shape = (Ws[l].shape[0], Ws[l].shape[1], 1, 1)
            gw += [grads[3 * l].view(shape), grads[3 * l + 1], grads[3 * l + 2]]
        gin = []
        for s_ in range(nseg):
            gin += [seg_grads[s_][0] if needs[4 * s_] else None, seg_grads[s_][1] if needs[4 * s_ + 1] else None,
                    seg_grads[s_][2] if needs[4 * s_ + 2] else None, None]
        return (None, None, *gin, *gw)


# pool backward in one pass (no zero fill of the dense gradient; csrc/compact.hip::pool_bwd_dense_kernel); `set_pool_bwd_dense(False)`
# = the zero fill + scatter pair it replaces: a TEST hook (both must give the same gradients), not a tuning switch
POOL_DENSE_COLS = 512
_POOL_BWD_DENSE = {"on": True}


def set_pool_bwd_dense(enabled):
    _POOL_BWD_DENSE["on"] = bool(enabled)


# (experiment, round 6) dY written once by the weight gradient, the data gradient loads one operand tensor: see
# profiles/r06_ab_dy_once.txt.  tools/ab_hook.py fused._DY_ONCE.on flips it; tests run both.
_DY_ONCE = {"on": False}


# data + weight gradient of a 64-input-channel inner layer in ONE kernel (csrc/mlp_wgrad.hip::fused_bwd_kernel: SA level 0's
# 64 -> 64 and 64 -> 128 layers); wider outputs take the wgrad2 + direct-dgrad pair
FUSED_BWD_MAX_COUT = 128
# layer-0 backward reduce through a transposed index (csrc/compact.hip::reduce_gather_kernel) whenever its index fits the LDS
# budget; the per-column LDS-atomic kernel (reduce_c_kernel) is the fallback for larger clouds.  `set_reduce_gather(False)`
# forces the fallback: a TEST hook (tests/test_fused_gpu.py runs both), not a tuning switch.
_REDUCE_GATHER = {"on": True}


def set_reduce_gather(enabled):
    _REDUCE_GATHER["on"] = bool(enabled)


def reduce_gather_enabled():
    return _REDUCE_GATHER["on"]


# ---- tracking inference: one kernel per set abstraction (csrc/sa_eval.hip) --------------------------------------
_EVAL_FUSED = {"on": True}      # (set_eval_fused(False): the layer-wise kernels, for the tests that compare the two)


def set_eval_fused(enabled):
    _EVAL_FUSED["on"] = bool(enabled)


def _eval_vec(lib, bn, gamma, beta, st):
    """(4, C) eval-mode BatchNorm constants, cached on the module until one of its tensors changes"""
    key = (bn.running_mean._version, bn.running_var._version, gamma._version, beta._version, bn.running_mean.data_ptr(),
           gamma.data_ptr(), float(bn.eps))
    hit = getattr(bn, "_o3d_eval_vec", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    vec = torch.empty((4, 1, bn.running_mean.numel()), device=bn.running_mean.device, dtype=torch.float32)
    _eval_consts(lib, bn, gamma.detach(), beta.detach(), vec, 1, st)
    object.__setattr__(bn, "_o3d_eval_vec", (key, vec))
    return vec


def _eval_fused_ok(mlp, layers, segs, nxyz):
    if not _EVAL_FUSED["on"] or mlp.training or layers is None or len(layers) != 3:
        return False
    if torch.is_grad_enabled() and (any(t is not None and t.requires_grad for sg in segs for t in sg[:3]) or
                                    any(p.requires_grad for p in mlp.parameters())):
        return False
    c = [conv.out_channels for conv, _ in layers]
    for sg in segs:
        B, npoint, ns = sg[3].shape
        if ns > 32 or (ns & (ns - 1)) or (B * npoint * ns) % 32:
            return False
    return c[0] % 8 == 0 and c[1] % 32 == 0 and c[2] % 32 == 0


def _run_eval(mlp, layers, segs, nxyz, inv_radius):
    """eval-mode QueryAndGroup + SharedMLP + max for one or two sets of clouds: per-point layer-0 GEMM for all of them,
    then ONE kernel per set (csrc/sa_eval.hip).  -> [pooled (B, C_last, npoint_s)]"""
    lib = capi.load()
    dev = segs[0][3].device
    f32 = torch.float32
    nseg = len(segs)
    B, _, ns = segs[0][3].shape
    C = segs[0][2].shape[1] if segs[0][2] is not None else 0
    Ws = [conv.weight.detach().reshape(conv.out_channels, -1) for conv, _ in layers]
    Cin0, C0 = nxyz + C, Ws[0].shape[0]
    Ns = [sg[2].shape[2] if sg[2] is not None else sg[0].shape[1] for sg in segs]
    Npads = [-(-n // TILE) * TILE for n in Ns]
    pt_bases = [0, B * Npads[0]][:nseg]
    ldz = sum(B * n for n in Npads)
    Cin0p = -(-Cin0 // 16) * 16 if Cin0 > 16 else Cin0
    with torch.cuda.device(dev):
        st = _stream()
        X0n = torch.empty((Cin0p, ldz), device=dev, dtype=f32)
        xs = [sg[0].detach().contiguous() if nxyz else None for sg in segs]
        fs = [sg[2].detach().contiguous() if C else None for sg in segs]
        _call("pack_points", 0.0, lib.o3d_pack_points, _ptr(xs[0]), _ptr(fs[0]), Ns[0], Npads[0],
              _ptr(xs[-1]) if nseg == 2 else None, _ptr(fs[-1]) if nseg == 2 else None, Ns[-1] if nseg == 2 else 0,
              Npads[-1] if nseg == 2 else 0, B, nxyz, C, float(inv_radius), Cin0p, X0n.data_ptr(), st)
        from .fused_heads import prep_for
        W0p = prep_for(dev).get(layers[0][0].weight, C0, Cin0p)
        Z = torch.empty((C0, ldz), device=dev, dtype=f32)
        _call("conv_fwd_points", 2.0 * Cin0p * C0 * ldz, lib.o3d_mlp_conv_fwd, X0n.data_ptr(), W0p.data_ptr(), None, None, 1,
              Cin0p, C0, ldz, Z.data_ptr(), None, None, st)
        vecs = [_eval_vec(lib, bn, bn.weight, bn.bias, st) for _, bn in layers]
        outs = []
        for s_, sg in enumerate(segs):
            npoint = sg[3].shape[1]
            centers = None
            if nxyz:
                centers = sg[1].detach().reshape(-1, 3)
                centers = (centers * inv_radius if inv_radius != 1.0 else centers).contiguous()
            out = torch.empty((B, Ws[2].shape[0], npoint), device=dev, dtype=f32)
            _call("sa_eval_fused", 0.0, lib.o3d_sa_eval_fused, Z.data_ptr(), ldz, sg[3].data_ptr(), _ptr(centers),
                  Ws[0].data_ptr(), Cin0, vecs[0].data_ptr(), Ws[1].data_ptr(), vecs[1].data_ptr(), Ws[2].data_ptr(),
                  vecs[2].data_ptr(), C0, Ws[1].shape[0], Ws[2].shape[0], B, npoint, ns, Npads[s_], pt_bases[s_],
                  out.data_ptr(), st)
            outs.append(out)
    return outs


def _compact_ok(layers, npoint, ns, B):
    if ns > 64 or B * npoint > (1 << 20) or B * npoint * ns >= (1 << 30):      # O3D_MAX_BALLS of csrc/compact.hip
        return False
    for i, (conv, _) in enumerate(layers):
        if i >= 1 and (conv.in_channels % 64 or conv.out_channels % 64):
            return False
    return layers[0][0].out_channels % 64 == 0


def _run(mlp, xyz, new_xyz, feats, idx, nxyz, inv_radius):
    layers = _layers(mlp)
    cfg = _Cfg()
    cfg.nxyz, cfg.inv_radius, cfg.training = nxyz, float(inv_radius), bool(mlp.training)
    cfg.bns = [bn for _, bn in layers]
    params = []
    for conv, bn in layers:
        params += [conv.weight, bn.weight, bn.bias]
    B, npoint, ns = idx.shape
    if _eval_fused_ok(mlp, layers, [(xyz, new_xyz, feats, idx)], nxyz):
        return _run_eval(mlp, layers, [(xyz, new_xyz, feats, idx)], nxyz, float(inv_radius))[0]
    if _compact_ok(layers, npoint, ns, B):
        return FusedGroupedMLPCompact.apply(cfg, 1, xyz, new_xyz, feats, idx, *params)
    return None       # the caller runs the block operator by operator (nsample > 64, unaligned channel counts)


def sa_group_mlp_pool(grouper, mlp, xyz, new_xyz, features):
    """QueryAndGroup + SharedMLP + max over nsample -> (B, C_last, npoint)."""
    idx = grouper.query(xyz, new_xyz)                      # ball query (B,npoint,nsample) int32
    _, npoint, ns = idx.shape
    if not _shape_ok(npoint, ns):
        return _composed(grouper, mlp, xyz, new_xyz, features, idx)
    inv_r = 1.0 / grouper.radius if grouper.normalize_xyz else 1.0
    out = _run(mlp, xyz, new_xyz, features, idx, 3, inv_r)
    return out if out is not None else _composed(grouper, mlp, xyz, new_xyz, features, idx)


def sa_group_mlp_pool_pair(grouper, mlp, a, b):
    """Two independent sets of clouds (a, b = (xyz, new_xyz, features)) through the same grouper and
    SharedMLP in ONE set of launches, numerically two consecutive calls (a first): separate BatchNorm
    batch statistics, running statistics updated twice.  Returns (pooled_a, pooled_b), or None when the
    joint layout does not apply (the caller then makes the two calls)."""
    if a[0].shape[0] != b[0].shape[0] or (a[2] is None) != (b[2] is None):
        return None
    if a[2] is not None and a[2].shape[1] != b[2].shape[1]:
        return None
    idx_a, idx_b = grouper.query(a[0], a[1]), grouper.query(b[0], b[1])
    B, np_a, ns = idx_a.shape
    np_b = idx_b.shape[1]
    layers = _layers(mlp)
    if not (_shape_ok(np_a, ns) and _shape_ok(np_b, ns) and _compact_ok(layers, np_a, ns, B) and
            _compact_ok(layers, np_b, ns, B) and (B * np_a * ns) % 256 == 0):
        return None
    inv_r = float(1.0 / grouper.radius if grouper.normalize_xyz else 1.0)
    if _eval_fused_ok(mlp, layers, [(a[0], a[1], a[2], idx_a), (b[0], b[1], b[2], idx_b)], 3):
        return tuple(_run_eval(mlp, layers, [(a[0], a[1], a[2], idx_a), (b[0], b[1], b[2], idx_b)], 3, inv_r))
    cfg = _Cfg()
    cfg.nxyz, cfg.training = 3, bool(mlp.training)
    cfg.inv_radius = inv_r
    cfg.bns = [bn for _, bn in layers]
    params = []
    for conv, bn in layers:
        params += [conv.weight, bn.weight, bn.bias]
    return FusedGroupedMLPCompact.apply(cfg, 2, a[0], a[1], a[2], idx_a, b[0], b[1], b[2], idx_b, *params)


capi.register("o3d_sample_query", [_vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _f, _i, _vp, _vp])


GEO_KEYS = ("centers", "ball_cnt", "ball_off", "gp", "cball", "cw", "meta")
_GEO_STATS = {"in_place": 0, "allocated": 0}      # (tests / diagnostics: how pair_geometry's outputs were provided)


def _pair_shapes_ok(layers, B, np_a, np_b, ns):
    return (_shape_ok(np_a, ns) and _shape_ok(np_b, ns) and _compact_ok(layers, np_a, ns, B) and
            _compact_ok(layers, np_b, ns, B) and (B * np_a * ns) % 256 == 0)


def pair_geometry(grouper, mlp, xyz_a, np_a, si_a, xyz_b, np_b, si_b, out=None):
    """Everything of a paired set-abstraction level that depends on the COORDINATES only (pointnet2_modules.py:52-62 and
    pointnet2_utils.py:299-320 up to the grouping indices): the centres (sampling gather), both ball queries and the
    distinct-neighbour layout of csrc/compact.hip -- sample_query + the three compaction launches -- as a dict of tensors
    (GEO_KEYS; written into `out`'s tensors when given) that `sa_pair_sampled(..., geo=...)` takes instead of computing them.  A training loop computes it for batch
    t+1 beside step t, like the farthest-point sampling (round 6: 12 launches, 0.14 ms per BAT step, off the step's serial
    chain).  -> None when the joint layout does not apply (the caller then lets the step compute it inline)."""
    if not (_SAMPLE_QUERY["on"] and xyz_a.is_cuda) or xyz_a.shape[0] != xyz_b.shape[0]:
        return None
    B, ns = xyz_a.shape[0], grouper.nsample
    if not _pair_shapes_ok(_layers(mlp), B, np_a, np_b, ns):
        return None
    if (si_a is None and np_a > xyz_a.shape[1]) or (si_b is None and np_b > xyz_b.shape[1]):
        return None
    dev, f32, i32 = xyz_a.device, torch.float32, torch.int32
    for si, npt in ((si_a, np_a), (si_b, np_b)):
        if si is not None and not (si.dtype == i32 and si.is_contiguous() and tuple(si.shape) == (B, npt) and si.device == dev):
            return None
    lib = capi.load()
    xa, xb = xyz_a.detach().contiguous(), xyz_b.detach().contiguous()
    nb_a, nb_b = B * np_a, B * np_b
    nballs, ldp = nb_a + nb_b, (nb_a + nb_b) * ns
    Npad_a, Npad_b = -(-xa.shape[1] // TILE) * TILE, -(-xb.shape[1] // TILE) * TILE
    shapes = {"centers": ((nballs + 1, 3), f32), "ball_cnt": ((nballs,), i32), "ball_off": ((nballs + 1,), i32),
              "gp": ((ldp,), i32), "cball": ((ldp,), i32), "cw": ((ldp,), f32), "meta": ((2, 4), i32)}
    if out is not None and all(k in out and tuple(out[k].shape) == sh and out[k].dtype == dt and out[k].device == dev and
                               out[k].is_contiguous() for k, (sh, dt) in shapes.items()):
        geo = {k: out[k] for k in shapes}        # written in place: the caller's buffers (a FlatBatch's own fields)
        _GEO_STATS["in_place"] += 1
    else:
        geo = {k: torch.empty(sh, device=dev, dtype=dt) for k, (sh, dt) in shapes.items()}
        _GEO_STATS["allocated"] += 1
    idx_a = torch.empty((B, np_a, ns), device=dev, dtype=i32)
    idx_b = torch.empty((B, np_b, ns), device=dev, dtype=i32)
    with torch.cuda.device(dev):
        st = _stream()
        _call("sample_query", 0.0, lib.o3d_sample_query, xa.data_ptr(), _ptr(si_a), xa.shape[1], np_a, idx_a.data_ptr(),
              xb.data_ptr(), _ptr(si_b), xb.shape[1], np_b, idx_b.data_ptr(), B, float(grouper.radius), ns,
              geo["centers"].data_ptr(), st)
        _call("compact_build", 0.0, lib.o3d_compact_build2, idx_a.data_ptr(), np_a, Npad_a, idx_b.data_ptr(), np_b, Npad_b, B,
              ns, nb_a * ns, B * Npad_a, nballs, geo["ball_cnt"].data_ptr(), geo["ball_off"].data_ptr(), geo["gp"].data_ptr(),
              geo["cball"].data_ptr(), geo["cw"].data_ptr(), geo["meta"].data_ptr(), st)
    return geo


def sa_pair_sampled(grouper, mlp, a, b, geo=None):
    """sa_group_mlp_pool_pair with the sampling folded in: a, b = (xyz (B,N,3), features | None, npoint, sample_idx (B,npoint)
    int32 | None = the arange(npoint) prefix).  ONE launch gathers the centres of both sets, writes them in the layout the
    fused path wants and runs both ball queries (csrc/index_ops.hip::sample_query_kernel; were 2 gathers or strided copies
    + 2 ball queries + 1 concatenation per level).  -> (new_xyz_a, pooled_a, new_xyz_b, pooled_b) or None when the joint
    layout does not apply or a coordinate tensor carries a gradient (the caller then takes the operator-by-operator route)."""
    (xyz_a, f_a, np_a, si_a), (xyz_b, f_b, np_b, si_b) = a, b
    if not (_SAMPLE_QUERY["on"] and xyz_a.is_cuda) or xyz_a.shape[0] != xyz_b.shape[0] or (f_a is None) != (f_b is None):
        return None
    if f_a is not None and f_a.shape[1] != f_b.shape[1]:
        return None
    if torch.is_grad_enabled() and (xyz_a.requires_grad or xyz_b.requires_grad):
        return None
    B, ns = xyz_a.shape[0], grouper.nsample
    layers = _layers(mlp)
    if not (_shape_ok(np_a, ns) and _shape_ok(np_b, ns) and _compact_ok(layers, np_a, ns, B) and
            _compact_ok(layers, np_b, ns, B) and (B * np_a * ns) % 256 == 0):
        return None
    if (si_a is None and np_a > xyz_a.shape[1]) or (si_b is None and np_b > xyz_b.shape[1]):
        return None
    dev, f32, i32 = xyz_a.device, torch.float32, torch.int32
    xa, xb = xyz_a.detach().contiguous(), xyz_b.detach().contiguous()
    # sample_query_kernel reads sidx[b * npoint + j]: exactly a contiguous (B, npoint) int32 tensor on the clouds' device (a
    # prefetched or caller-supplied index of another width, batch or device takes the operator-by-operator route, which
    # follows idx.shape)
    for si, npt in ((si_a, np_a), (si_b, np_b)):
        if si is not None and not (si.dtype == i32 and si.is_contiguous() and tuple(si.shape) == (B, npt) and si.device == dev):
            return None
    nb_a, nb_b = B * np_a, B * np_b
    if geo is not None and not (mlp.training and tuple(geo["centers"].shape) == (nb_a + nb_b + 1, 3) and
                                geo["gp"].numel() == (nb_a + nb_b) * ns and geo["gp"].device == dev):
        geo = None          # (eval mode takes the one-kernel set abstraction, which wants the grouping indices themselves)
    if geo is not None:     # the coordinate-only part came with the batch (pair_geometry, computed beside the previous step)
        centers, idx_a, idx_b = geo["centers"], None, None
    else:
        centers = torch.empty((nb_a + nb_b + 1, 3), device=dev, dtype=f32)
        idx_a = torch.empty((B, np_a, ns), device=dev, dtype=i32)
        idx_b = torch.empty((B, np_b, ns), device=dev, dtype=i32)
        with torch.cuda.device(dev):
            _call("sample_query", 0.0, capi.load().o3d_sample_query, xa.data_ptr(), _ptr(si_a), xa.shape[1], np_a,
                  idx_a.data_ptr(), xb.data_ptr(), _ptr(si_b), xb.shape[1], np_b, idx_b.data_ptr(), B, float(grouper.radius), ns,
                  centers.data_ptr(), _stream())
    new_a, new_b = centers[:nb_a].view(B, np_a, 3), centers[nb_a:nb_a + nb_b].view(B, np_b, 3)
    inv_r = float(1.0 / grouper.radius if grouper.normalize_xyz else 1.0)
    if geo is None and _eval_fused_ok(mlp, layers, [(xa, new_a, f_a, idx_a), (xb, new_b, f_b, idx_b)], 3):
        outs = _run_eval(mlp, layers, [(xa, new_a, f_a, idx_a), (xb, new_b, f_b, idx_b)], 3, inv_r)
        return new_a, outs[0], new_b, outs[1]
    cfg = _Cfg()
    cfg.nxyz, cfg.training = 3, bool(mlp.training)
    cfg.inv_radius = inv_r
    cfg.bns = [bn for _, bn in layers]
    cfg.centers = centers
    cfg.geo, cfg.geo_dims = geo, (B, ns, (np_a, np_b))
    params = []
    for conv, bn in layers:
        params += [conv.weight, bn.weight, bn.bias]
    outs = FusedGroupedMLPCompact.apply(cfg, 2, xa, new_a, f_a, idx_a, xb, new_b, f_b, idx_b, *params)
    return new_a, outs[0], new_b, outs[1]


# (TEST hook: False = gather / ball query / concatenation as separate launches, the route sa_pair_sampled is tested against)
_SAMPLE_QUERY = {"on": True}


def set_sample_query(enabled):
    _SAMPLE_QUERY["on"] = bool(enabled)


def group_mlp_pool(mlp, bundle, idx):
    """grouping_operation(bundle, idx) + SharedMLP + max over the last axis -> (B, C_last, idx.shape[1])."""
    _, npoint, ns = idx.shape
    out = _run(mlp, None, None, bundle, idx, 0, 1.0) if _shape_ok(npoint, ns) else None
    if out is None:
        from . import ops
        out = mlp(ops.grouping_operation(bundle, idx)).max(dim=-1)[0]
    return out


def _composed(grouper, mlp, xyz, new_xyz, features, idx):
    """operator-by-operator path for shapes the tile kernels do not cover (npoint*nsample % 128 != 0)"""
    import torch.nn.functional as F
    from . import ops
    rel = ops.grouping_operation(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
    if grouper.normalize_xyz:
        rel = rel / grouper.radius
    x = rel if features is None else torch.cat([rel, ops.grouping_operation(features, idx)], dim=1)
    x = mlp(x)
    return F.max_pool2d(x, kernel_size=[1, x.size(3)]).squeeze(-1)
