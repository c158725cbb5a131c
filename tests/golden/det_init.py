"""Deterministic, storage-free parameter values shared by tests/golden/make_golden_trackers.py (run on the reference's
own BAT / P2B classes) and tests/test_golden_trackers.py (run on the host mirror): every tensor of a state_dict is a
closed-form function of its key and shape, so the 1.4 M weights need not be stored in the fixture."""
import zlib

import torch


def fill_state_dict(module):
    sd = module.state_dict()
    new = {}
    for key in sorted(sd):
        t = sd[key]
        if not t.dtype.is_floating_point:          # num_batches_tracked
            new[key] = torch.zeros_like(t)
            continue
        n = t.numel()
        phase = (zlib.crc32(key.encode()) % 1000) * 0.0173
        i = torch.arange(n, dtype=torch.float64)
        wave = torch.cos(0.7310 * i + phase) * 0.6 + torch.sin(0.1937 * i * 1.618 + 2.0 * phase) * 0.4
        if key.endswith("running_var"):
            v = 1.0 + 0.4 * wave
        elif key.endswith("running_mean"):
            v = 0.1 * wave
        elif key.endswith("bn.weight") or key.endswith("bn.bn.weight"):
            v = 1.0 + 0.3 * wave
        elif key.endswith("bias"):
            v = 0.05 * wave
        else:                                      # conv weights: variance-preserving scale
            fan_in = max(1, n // t.shape[0])
            v = wave * (1.7 / fan_in) ** 0.5
        new[key] = v.to(t.dtype).reshape(t.shape)
    module.load_state_dict(new, strict=True)
    return module


def fill_state_dict_random(module, seed=0):
    """Storage-free like `fill_state_dict`, but non-degenerate: every tensor is drawn from a numpy PCG64 stream seeded by
    (crc32(key), seed) -- He-normal conv weights, BatchNorm gamma in [0.7, 1.3], small biases / running means, running
    variances in [0.6, 1.4].  The cos/sin waves above are periodic along the input channels, which makes the heads'
    BatchNorm batch variances tiny for many channels (1/sigma then amplifies fp32 rounding); these are not."""
    import numpy as np
    sd = module.state_dict()
    new = {}
    for key in sorted(sd):
        t = sd[key]
        if not t.dtype.is_floating_point:
            new[key] = torch.zeros_like(t)
            continue
        rng = np.random.default_rng([zlib.crc32(key.encode()), seed])
        shape = tuple(t.shape)
        if key.endswith("running_var"):
            v = rng.uniform(0.6, 1.4, shape)
        elif key.endswith("running_mean"):
            v = rng.normal(0.0, 0.1, shape)
        elif key.endswith("bn.weight") or key.endswith("bn.bn.weight"):
            v = rng.uniform(0.7, 1.3, shape)
        elif key.endswith("bias"):
            v = rng.normal(0.0, 0.05, shape)
        else:
            fan_in = max(1, t.numel() // t.shape[0])
            v = rng.normal(0.0, (2.0 / fan_in) ** 0.5, shape)
        new[key] = torch.from_numpy(np.asarray(v)).to(t.dtype)
    module.load_state_dict(new, strict=True)
    return module


def fill_by_module_type(module, seed=0):
    """Storage-free like `fill_state_dict_random`, for trees whose BatchNorm keys carry no `bn.` marker (M2-Track's
    nn.Sequential stacks): what a tensor is comes from the TYPE of the module that owns it, the values from a PCG64 stream
    seeded by (crc32(state-dict key), seed) -- so the reference's class and the mirror (same keys) get the same numbers."""
    import numpy as np
    from torch import nn
    with torch.no_grad():
        for mname, m in module.named_modules():
            for pname, t in list(m.named_parameters(recurse=False)) + list(m.named_buffers(recurse=False)):
                key = (mname + "." if mname else "") + pname
                if not t.dtype.is_floating_point:
                    t.zero_()
                    continue
                rng = np.random.default_rng([zlib.crc32(key.encode()), seed])
                shape = tuple(t.shape)
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    v = {"weight": rng.uniform(0.7, 1.3, shape), "bias": rng.normal(0.0, 0.05, shape),
                         "running_mean": rng.normal(0.0, 0.1, shape), "running_var": rng.uniform(0.6, 1.4, shape)}[pname]
                elif pname == "bias":
                    v = rng.normal(0.0, 0.05, shape)
                else:
                    fan_in = max(1, t.numel() // t.shape[0])
                    v = rng.normal(0.0, (2.0 / fan_in) ** 0.5, shape)
                t.copy_(torch.from_numpy(np.asarray(v)).to(t.dtype))
    return module
