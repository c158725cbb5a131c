"""Conv/BN/ReLU building blocks with the reference's module tree (state_dict compatible).

Mirror of pointnet2/utils/pytorch_utils.py: `SharedMLP` :12 (stack of 1x1 Conv2d ->
BatchNorm2d -> ReLU), `Conv1d/2d/3d` :124-223 (bias dropped when bn=True :88, kaiming_normal
weights :97, module order conv->bn->activation or the pre-activation order :101-121),
`BatchNorm1d/2d/3d` :50-65 (weight 1, bias 0, nested as `<name>bn.bn`), `FC` :226, `Seq`
:300 (fluent builder with children "0","1",...), `BNMomentumScheduler` :272.

Checkpoint keys are therefore identical to the reference's, e.g.
`SA_modules.0.mlps.0.layer0.conv.weight`, `...layer0.bn.bn.running_mean`.
The fused MI355X path (open3dsot_amd.fused) reads parameters out of these modules; it
never changes the tree.
"""
import torch
import torch.nn as nn

_FLAT = {"on": True}


def set_flat_pointwise(enabled):
    """1x1 Conv1d stacks as plain GEMMs on the flat (C, B*N) layout (default) or as F.conv1d calls."""
    _FLAT["on"] = bool(enabled)


def pointwise_conv1d(conv, x):
    """nn.Conv1d with kernel 1 on (B,C,N) as ONE (Cout,Cin) x (Cin,B*N) GEMM: on the GPU the library's fp32-MFMA
    kernel on the flat layout (open3dsot_amd/fused_heads.py), else torch.mm.  Same arithmetic as F.conv1d."""
    if _FLAT["on"] and x.is_cuda:
        from . import fused_heads
        units = [(conv, None, None)]
        if fused_heads.chain_supported([x], units):
            return fused_heads.run_chain([x], units)
    B, C, N = x.shape
    h = torch.mm(conv.weight.view(conv.out_channels, C), x.permute(1, 0, 2).reshape(C, B * N))
    if conv.bias is not None:
        h = h + conv.bias[:, None]
    return h.reshape(-1, B, N).permute(1, 0, 2).contiguous()


def pointwise_conv1d_pair(conv, xa, xb):
    """(pointwise_conv1d(conv, xa), pointwise_conv1d(conv, xb)) for two sets of clouds of different sizes through the SAME
    convolution (conv_final on the template and on the search feature, models/bat.py:91-92): on the GPU one GEMM over the
    columns of both (open3dsot_amd/fused_heads.py::SharedConvPair)"""
    if _FLAT["on"] and xa.is_cuda:
        from . import fused_heads
        if fused_heads.shared_conv_pair_supported(conv, xa, xb):
            return fused_heads.SharedConvPair.apply(conv.weight, conv.bias, xa, xb)
    return pointwise_conv1d(conv, xa), pointwise_conv1d(conv, xb)


def seq_apply(seq, parts, residual=False):
    """seq(torch.cat(parts, dim=1)) (+ the concatenated input when `residual`: `seeds + vote_layer(seeds)`,
    models/head/rpn.py:50-54) for a pt_utils.Seq of kernel-1 Conv1d units.  On the GPU the parts go straight into the
    stack's packed operand (no cat, no transposes: parts may be any strided (B,C_i,N) views)."""
    units = seq._flat_units() if (_FLAT["on"] and isinstance(seq, Seq)) else None
    if units is not None and parts[0].is_cuda:
        from . import fused_heads
        if fused_heads.chain_supported(parts, units):
            return fused_heads.run_chain(parts, units, residual)
    x = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
    y = seq(x)
    return x + y if residual else y


def seq_apply_pair(a, b):
    """(seq_a, parts_a, residual_a), (seq_b, parts_b, residual_b): two independent pt_utils.Seq stacks over the same
    (B, N) -- FC_layer_cla and vote_layer of models/head/rpn.py:44-54 -- -> (out_a, out_b) = (seq_apply(*a), seq_apply(*b)).
    On the GPU the two stacks advance layer by layer in merged launches (open3dsot_amd/fused_heads.py::run_chain_pair)."""
    ua = a[0]._flat_units() if (_FLAT["on"] and isinstance(a[0], Seq)) else None
    ub = b[0]._flat_units() if (_FLAT["on"] and isinstance(b[0], Seq)) else None
    if ua is not None and ub is not None and a[1][0].is_cuda:
        from . import fused_heads
        if fused_heads.chain_supported(a[1], ua) and fused_heads.chain_supported(b[1], ub):
            return fused_heads.run_chain_pair((a[1], ua, a[2]), (b[1], ub, b[2]))
    return seq_apply(*a), seq_apply(*b)


_CONV = {1: nn.Conv1d, 2: nn.Conv2d, 3: nn.Conv3d}
_BN = {1: nn.BatchNorm1d, 2: nn.BatchNorm2d, 3: nn.BatchNorm3d}


class RowBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d (same parameters, buffers and state_dict keys) for the (B, C) rows of the M2-Track heads
    (models/m2track.py:47-71, models/backbone/pointnet.py:118-126).  Inside a tracker forward on the GPU its
    `num_batches_tracked += 1` joins the forward's ONE multi-tensor counter update (open3dsot_amd/fused.py::count_batches)
    instead of launching its own: twelve such modules per M2-Track step.  Everywhere else it IS nn.BatchNorm1d."""

    def forward(self, x):
        if self.training and self.track_running_stats and self.momentum is not None and x.is_cuda:
            from . import fused
            if fused._COUNTERS["pending"] is not None:
                self._check_input_dim(x)
                out = nn.functional.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, True,
                                               self.momentum, self.eps)
                fused.count_batches([self], 1, raw_writes=False)
                return out
        return super().forward(x)


class _BNWrap(nn.Sequential):
    """BatchNorm nested one level down (`bn`) with gamma=1, beta=0."""

    def __init__(self, channels, dims, name=""):
        super().__init__()
        norm = _BN[dims](channels)
        nn.init.constant_(norm.weight, 1.0)
        nn.init.constant_(norm.bias, 0.0)
        self.add_module(name + "bn", norm)


class BatchNorm1d(_BNWrap):
    def __init__(self, in_size, *, name=""):
        super().__init__(in_size, 1, name)


class BatchNorm2d(_BNWrap):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, 2, name)


class BatchNorm3d(_BNWrap):
    def __init__(self, in_size, name=""):
        super().__init__(in_size, 3, name)


_BN_WRAP = {1: BatchNorm1d, 2: BatchNorm2d, 3: BatchNorm3d}


class _ConvUnit(nn.Sequential):
    """conv (+bn) (+activation); `preact` puts bn/activation in front of the conv."""

    DIMS = 2

    def __init__(self, in_size, out_size, *, kernel_size=None, stride=None, padding=None,
                 activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_,
                 bias=True, preact=False, name=""):
        super().__init__()
        d = self.DIMS
        one, zero = (1,) * d, (0,) * d
        if d == 1:
            one, zero = 1, 0
        conv = _CONV[d](in_size, out_size,
                        kernel_size=one if kernel_size is None else kernel_size,
                        stride=one if stride is None else stride,
                        padding=zero if padding is None else padding,
                        bias=bias and not bn)
        init(conv.weight)
        if conv.bias is not None:
            nn.init.constant_(conv.bias, 0)
        norm = None
        if bn:
            norm = _BN_WRAP[d](in_size if preact else out_size)
        head = []
        if norm is not None:
            head.append(("bn", norm))
        if activation is not None:
            head.append(("activation", activation))
        order = head + [("conv", conv)] if preact else [("conv", conv)] + head
        for key, mod in order:
            self.add_module(name + key, mod)


class Conv1d(_ConvUnit):
    DIMS = 1


class Conv2d(_ConvUnit):
    DIMS = 2


class Conv3d(_ConvUnit):
    DIMS = 3


class SharedMLP(nn.Sequential):
    """Pointwise MLP over a (B,C,npoint,nsample) tensor: children `layer0`, `layer1`, ..."""

    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True), preact=False,
                 first=False, name=""):
        super().__init__()
        for i in range(len(args) - 1):
            plain = first and preact and i == 0  # first pre-activation layer: conv only
            self.add_module(
                name + "layer{}".format(i),
                Conv2d(args[i], args[i + 1], bn=bn and not plain,
                       activation=None if plain else activation, preact=preact))


class FC(nn.Sequential):
    def __init__(self, in_size, out_size, *, activation=nn.ReLU(inplace=True), bn=False,
                 init=None, preact=False, name=""):
        super().__init__()
        fc = nn.Linear(in_size, out_size, bias=not bn)
        if init is not None:
            init(fc.weight)
        if not bn:
            nn.init.constant_(fc.bias, 0)
        head = []
        if bn:
            head.append(("bn", BatchNorm1d(in_size if preact else out_size)))
        if activation is not None:
            head.append(("activation", activation))
        order = head + [("fc", fc)] if preact else [("fc", fc)] + head
        for key, mod in order:
            self.add_module(name + key, mod)


def feature_dropout_no_scaling(x, p, train=False, inplace=False):
    """Whole-channel dropout WITHOUT the 1/(1-p) rescaling: a Bernoulli(1 - p) keep mask per (sample, channel), broadcast
    over the remaining dimensions.  `pointnet2_utils.RandomDropout` (pointnet2/utils/pointnet2_utils.py:24-32) calls
    `pt_utils.feature_dropout_no_scaling`, which the reference's own pytorch_utils.py does not define (the class cannot run
    there; no tracker uses it): this is the upstream `pointnet2_ops` function of that name, restated with torch ops."""
    if p < 0 or p > 1:
        raise ValueError("dropout probability has to be between 0 and 1, but got {}".format(p))
    if not train or float(p) == 0.0:
        return x
    keep = torch.empty(x.shape[:2] + (1,) * (x.dim() - 2), device=x.device, dtype=x.dtype).bernoulli_(1.0 - float(p))
    return x.mul_(keep) if inplace else x * keep


def set_bn_momentum_default(bn_momentum):
    def fn(m):
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = bn_momentum
    return fn


class BNMomentumScheduler(object):
    def __init__(self, model, bn_lambda, last_epoch=-1, setter=set_bn_momentum_default):
        if not isinstance(model, nn.Module):
            raise RuntimeError("Class '{}' is not a PyTorch nn Module".format(type(model).__name__))
        self.model, self.setter, self.lmbd = model, setter, bn_lambda
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        self.model.apply(self.setter(self.lmbd(epoch)))


class Seq(nn.Sequential):
    """Fluent builder: Seq(c).conv1d(c2, bn=True).conv1d(c3, activation=None) ..."""

    def __init__(self, input_channels):
        super().__init__()
        self.count = 0
        self.current_channels = input_channels

    def _push(self, module, out_size=None):
        self.add_module(str(self.count), module)
        self.count += 1
        if out_size is not None:
            self.current_channels = out_size
        return self

    def _conv(self, cls, out_size, kw):
        kw.pop("dilation", None)
        kw.pop("norm_layer", None)
        return self._push(cls(self.current_channels, out_size, **kw), out_size)

    def conv1d(self, out_size, **kw):
        return self._conv(Conv1d, out_size, kw)

    def conv2d(self, out_size, **kw):
        return self._conv(Conv2d, out_size, kw)

    def conv3d(self, out_size, **kw):
        return self._conv(Conv3d, out_size, kw)

    def fc(self, out_size, **kw):
        return self._push(FC(self.current_channels, out_size, **kw), out_size)

    def dropout(self, p=0.5):
        return self._push(nn.Dropout(p=0.5))  # the reference ignores p as well (:431)

    def _flat_units(self):
        """[(conv, bn | None, activation | None)] when every child is a kernel-1 Conv1d unit, else None"""
        units = []
        for unit in self.children():
            if not isinstance(unit, Conv1d):
                return None
            kids = dict(unit.named_children())
            conv = kids.get("conv")
            if conv is None or list(kids)[0] != "conv" or conv.kernel_size != (1,) or conv.stride != (1,) or \
                    conv.padding != (0,) or conv.groups != 1:
                return None
            bn = kids.get("bn")
            units.append((conv, bn.bn if bn is not None else None, kids.get("activation")))
        return units or None

    def forward(self, x):
        units = self._flat_units() if (_FLAT["on"] and x.dim() == 3) else None
        if units is None:
            return super().forward(x)
        if x.is_cuda:
            from . import fused_heads
            if fused_heads.chain_supported([x], units):
                return fused_heads.run_chain([x], units)
        # flat layout (C, B*N): each layer is one GEMM; BatchNorm1d sees (1, C, B*N), i.e. the same
        # per-channel statistics over all B*N positions
        B, C, N = x.shape
        h = x.permute(1, 0, 2).reshape(C, B * N)
        for conv, bn, act in units:
            w = conv.weight.view(conv.out_channels, -1)      # (a view: indexing [:, :, 0] costs a zero-fill + copy in backward)
            if w.shape[0] < 32:     # hipBLASLt picks a 16-row macro tile (87 us per call, rocprofv3) for the
                w = torch.nn.functional.pad(w, (0, 0, 0, 32 - w.shape[0]))   # 1/5/9-channel outputs: pad to 32
                h = torch.mm(w, h)[:conv.out_channels]
            else:
                h = torch.mm(w, h)
            if conv.bias is not None:
                h = h + conv.bias[:, None]
            if bn is not None:
                h = bn(h.unsqueeze(0)).squeeze(0)
            if act is not None:
                h = act(h)
        return h.reshape(-1, B, N).permute(1, 0, 2).contiguous()

    def maxpool2d(self, kernel_size, stride=None, padding=0, dilation=1, return_indices=False,
                  ceil_mode=False):
        return self._push(nn.MaxPool2d(kernel_size=kernel_size, stride=stride, padding=padding,
                                       dilation=dilation, return_indices=return_indices,
                                       ceil_mode=ceil_mode))
