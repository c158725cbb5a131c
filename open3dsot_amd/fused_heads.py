"""The trackers' 1-D conv stacks on the library's GEMM kernels, flat (C, P = B*N) layout end to end.

What it replaces (same numbers within fp32 rounding, tests/test_heads_gpu.py):
  pt_utils.Seq of Conv1d(+BatchNorm1d+ReLU) units   pointnet2/utils/pytorch_utils.py:124-155,300-457
    FC_layer_cla, vote_layer (+ the residual `seeds + vote_layer(seeds)`), FC_proposal   models/head/rpn.py:16-39,50-54
    fea_layer                                          models/head/xcorr.py:14-17
    mlp_bc, conv_final                                 models/bat.py:22-26,91-94
The reference runs each unit as conv1d -> batch_norm -> relu on (B,C,N) tensors with torch.cat / transpose
around them; here a stack is: ONE pack launch that stacks its (arbitrarily strided) sources into the zero-padded
(K, P) operand, then per layer one fp32-MFMA GEMM (csrc/mlp_direct.hip; BatchNorm + ReLU of the producer
applied on load, statistics partials in the epilogue, bias / residual in the last layer's epilogue) and one
BatchNorm finalize; the backward mirrors it (data gradient with the BatchNorm-backward constants folded in,
tile-matched weight gradient of csrc/mlp_wgrad.hip).  Parameters stay in the caller's modules.

`WeightPrep`: every zero-padded / transposed weight copy the GEMMs need (K % 16, M % 64, W^T for the data
gradient) is a job in one table; inside a `prep_scope` all jobs of the device are refreshed by ONE launch at scope
entry (the weights only change in the optimizer step) instead of one small copy per layer per step.
"""
import contextlib
import ctypes
import weakref

import torch

from . import capi
from .fused import _call, _const_vec, _eval_consts, _ptr, _stream, count_batches, counters_begin, counters_end

_vp, _i, _l, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double
capi.register("o3d_pack_rows", [_vp, _i, _i, _i, _i, _vp, _vp])
capi.register("o3d_pack_rows_ld", [_vp, _i, _i, _i, _i, _vp, _l, _vp])
capi.register("o3d_prep_weights", [_vp, _i, _vp])
capi.register("o3d_row_sum", [_vp, _i, _l, _vp, _vp])
capi.register("o3d_pw_tile", [_l, _i])
capi.register("o3d_pw_fwd", [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, _vp, _vp, _vp, _vp])
capi.register("o3d_pw_dgrad", [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
capi.register("o3d_mlp_conv_wgrad2_group", [_vp, _i, _vp])


class _WgradJob(ctypes.Structure):        # o3d_wgrad_job of include/o3dsot.h
    _fields_ = [("dN", _vp), ("Y", _vp), ("A1", _vp), ("A2", _vp), ("A3", _vp), ("X", _vp), ("in_scale", _vp),
                ("in_shift", _vp), ("Cin", _i), ("Cout", _i), ("P", _l), ("scratch", _vp), ("dW", _vp), ("out_rows", _i),
                ("out_cols", _i)]


# the weight gradients of a stack are ONE grouped launch (+ one reduction launch) at the end of its backward instead of a
# launch + reduction per layer (csrc/mlp_wgrad.hip::wgrad2_group_kernel)
_MAXJOBS = 8                # WG_MAXJOBS of csrc/mlp_wgrad.hip
# Deferred weight gradients: inside `defer_wgrads()` (the training step of open3dsot_amd/dist.py wraps loss.backward() in
# it) the stacks do not launch their grouped weight gradients at the end of their own backward but queue the jobs; the
# queue is flushed in groups of 8 when the scope ends -- the heads' 23 jobs of a BAT step in 3 launches + 3 reductions
# instead of 7 + 7.  The gradient tensors autograd was handed are filled by the flush: the scope must end before anything
# reads them (AccumulateGrad only stores them; `_deferrable` lists what that requires of the parameters).
_DEFER = {"queue": None, "keys": None}


def _flush_jobs(jobs, st):
    lib = capi.load()
    for j0 in range(0, len(jobs), _MAXJOBS):
        chunk = jobs[j0:j0 + _MAXJOBS]
        arr = (_WgradJob * len(chunk))(*[_WgradJob(*j[1]) for j in chunk])
        _call("pw_conv_wgrad", sum(j[0] for j in chunk), lib.o3d_mlp_conv_wgrad2_group, ctypes.addressof(arr), len(chunk), st)


def _flush_queue(early=False):
    """early: FULL groups of jobs (multiples of _MAXJOBS) while the scope is still open -- launched on the weight-gradient side
    branch when one is open (open3dsot_amd/fused.py::wgrad_branch), beside the rest of the backward instead of behind it; what
    does not fill a group stays queued; the parameter keys stay recorded, so that a later second use of one of these
    parameters still finds it"""
    q = _DEFER["queue"]
    _DEFER["queue"] = []
    if not early:
        _DEFER["keys"] = set()
    by_stream = {}
    for jobs, keep, st in q:
        by_stream.setdefault(st, []).extend(jobs)
    for st, jobs in by_stream.items():
        side = None
        if early:
            n = (len(jobs) // _MAXJOBS) * _MAXJOBS
            if n < len(jobs):        # the remainder waits for the next full group / the end of the scope (operands: held by
                _DEFER["queue"].append((jobs[n:], q, st))      # `q`, which the re-queued entry keeps alive)
                jobs = jobs[:n]
            if jobs and st == _stream():
                from . import fused
                side = fused._branch_side([], q)       # (the scope keeps `q` -- the operands -- alive until the join)
        if not jobs:
            continue
        if side is not None:
            with torch.cuda.stream(side):
                _flush_jobs(jobs, side.cuda_stream)
        else:
            _flush_jobs(jobs, st)
    del q


def _deferrable(params):
    """Deferral hands autograd gradient tensors that are FILLED LATER (at the flush).  That is only sound when autograd
    does nothing with them but store them: the parameter has no gradient yet (else AccumulateGrad runs `grad += new` on
    the unfilled buffer) and carries no tensor / post-accumulate hooks (which would read it)."""
    for p in params:
        if p is None:
            continue
        if p.grad is not None or getattr(p, "_backward_hooks", None) or getattr(p, "_post_accumulate_grad_hooks", None):
            return False
    return True


def _submit_jobs(jobs, keep, st, params):
    """launch the jobs now, or queue them inside a `defer_wgrads` scope.  params: the parameters whose gradients the
    jobs produce.  Contract of the deferred path (tests/test_heads_gpu.py::test_deferred_wgrads_*):
      * a parameter that is used TWICE in the step (conv_final on the template and on the search feature when the two
        calls are not merged) gets its second gradient ADDED to the first by autograd as soon as this backward returns,
        so both must be complete by then: a repeated parameter flushes the queue on the spot;
      * a parameter that already holds a gradient, or carries hooks, is never deferred (`_deferrable`);
      * a parameter that is ALSO used by a non-fused torch op in the same step is not supported inside the scope
        (autograd's input buffer would sum an unfilled tensor): the trackers have none, DataParallelStep is the only
        caller that opens the scope, and it clears the gradients first."""
    if not jobs:
        return
    if _DEFER["queue"] is None:
        _flush_jobs(jobs, st)
        return
    keys = {id(p) for p in params if p is not None}
    _DEFER["queue"].append((jobs, keep, st))      # (the queue keeps the operands alive until the flush)
    if (_DEFER["keys"] & keys) or not _deferrable(params):
        _flush_queue()
        from . import fused
        fused.branch_join()         # an earlier group with this parameter may be running on the side branch
    else:
        _DEFER["keys"] |= keys
        from . import fused
        if fused._BRANCH["scope"] is not None and sum(len(e[0]) for e in _DEFER["queue"]) >= _MAXJOBS:
            # only with the weight-gradient side branch open (off by default): full groups go out beside the rest of the
            # backward.  Without it an early flush only splits the step's 23 jobs into more launches than the 3 of the final one
            _flush_queue(early=True)


@contextlib.contextmanager
def defer_wgrads():
    if _DEFER["queue"] is not None:          # nested scopes: the outer one flushes
        yield
        return
    _DEFER["queue"], _DEFER["keys"] = [], set()
    try:
        yield
    finally:
        _flush_queue()
        _DEFER["queue"] = _DEFER["keys"] = None

_ON = {"on": True}


def set_fused_heads(enabled):
    """1-D conv stacks on the library's kernels (default) or on torch ops (the specification they are tested against)"""
    _ON["on"] = bool(enabled)


def enabled():
    return _ON["on"]


def _up(v, m):
    return -(-v // m) * m


class _RowsSrc(ctypes.Structure):
    _fields_ = [("p", ctypes.c_void_p), ("sb", ctypes.c_long), ("sc", ctypes.c_long), ("sn", ctypes.c_long),
                ("C", ctypes.c_int)]


def pack_rows(sources, rows):
    """[(B,C_i,N) tensors, any strides] -> X (rows, B*N): the sources stacked along the rows, zero rows below"""
    lib = capi.load()
    B, _, N = sources[0].shape
    X = torch.empty((rows, B * N), device=sources[0].device, dtype=torch.float32)
    arr = (_RowsSrc * len(sources))()
    for i, t in enumerate(sources):
        sb, sc, sn = t.stride()
        arr[i].p, arr[i].sb, arr[i].sc, arr[i].sn, arr[i].C = t.data_ptr(), sb, sc, sn, t.shape[1]
    _call("pack_rows", 0.0, lib.o3d_pack_rows, ctypes.addressof(arr), len(sources), B, N, rows, X.data_ptr(), _stream())
    return X


def pack_rows_into(t, X, col0):
    """(B,C,N) tensor (any strides) -> the columns [col0, col0 + B*N) of the rows [0, C) of X (rows, ld)"""
    lib = capi.load()
    B, C, N = t.shape
    arr = (_RowsSrc * 1)()
    sb, sc, sn = t.stride()
    arr[0].p, arr[0].sb, arr[0].sc, arr[0].sn, arr[0].C = t.data_ptr(), sb, sc, sn, C
    _call("pack_rows", 0.0, lib.o3d_pack_rows_ld, ctypes.addressof(arr), 1, B, N, C, X[0, col0:].data_ptr(), X.shape[1], _stream())


def _as_flat(t):
    """(B,C,N) view of a contiguous (C,B,N) buffer -> that buffer as (C, B*N), else None"""
    p = t.permute(1, 0, 2)
    return p.reshape(p.shape[0], -1) if p.is_contiguous() else None


# ---- weight preparation ------------------------------------------------------------------------------------
class _Job:
    __slots__ = ("ref", "dst", "rows", "cols", "transpose", "src_ptr", "snap")


class WeightPrep:
    def __init__(self, dev):
        self.dev = dev
        self.jobs = {}
        self.table = None
        self.tables = []      # every table a launch has ever been given: a captured HIP graph replays that launch with THAT
        self.dirty = False    # address, so a table is never freed while this object lives (round 6: a second model on the
        self.active = False   # device used to rebuild -- and free -- the one table a captured step of the first still read)
        self.fresh = set()

    @staticmethod
    def _copy_now(job, param):
        src = param.detach().reshape(job.rows, job.cols)
        if job.transpose:
            job.dst[:job.cols, :job.rows].copy_(src.t())
        else:
            job.dst[:job.rows, :job.cols].copy_(src)

    def get(self, param, rows_p, cols_p, transpose=False):
        """param viewed as (param.shape[0], rest) -> zero-padded (rows_p, cols_p) copy of it (or of its transpose)"""
        rows = param.shape[0] if param.dim() > 1 else 1
        if not transpose and (rows, param.numel() // rows) == (rows_p, cols_p):
            return param.detach().reshape(rows_p, cols_p)           # already in shape: no copy at all
        key = (id(param), rows_p, cols_p, bool(transpose))
        job = self.jobs.get(key)
        if job is None or job.ref() is not param:
            job = _Job()
            job.ref = weakref.ref(param)
            job.rows = param.shape[0] if param.dim() > 1 else 1
            job.cols = param.numel() // job.rows
            job.transpose = bool(transpose)
            job.dst = torch.zeros((rows_p, cols_p), device=param.device, dtype=torch.float32)
            job.src_ptr = param.data_ptr()
            self.jobs[key] = job
            self.dirty = True
            self._copy_now(job, param)
        elif not (self.active and key in self.fresh) or job.src_ptr != param.data_ptr():
            self._copy_now(job, param)
        return job.dst

    def snapshot(self, buf, nrows):
        """(nrows, C) tensor whose every row is a copy of the 1-D `buf` taken by the table's launch at the start of the
        forward (or on the spot outside a prep scope): the BatchNorm running means as they were before the forward's
        own updates, one row per segment of a paired set abstraction (fused.py; was a torch.cat per level)"""
        n = buf.numel()
        key0 = (id(buf), "snap", nrows, 0)
        job0 = self.jobs.get(key0)
        if job0 is None or job0.ref() is not buf:
            snap = torch.zeros((nrows, n), device=buf.device, dtype=torch.float32)
            for r in range(nrows):
                job = _Job()
                job.ref = weakref.ref(buf)
                job.rows, job.cols, job.transpose = 1, n, False
                job.dst = snap[r:r + 1]
                job.snap = snap
                job.src_ptr = buf.data_ptr()
                self.jobs[(id(buf), "snap", nrows, r)] = job
            self.dirty = True
            job0 = self.jobs[key0]
            job0.snap.copy_(buf.detach().reshape(1, n).expand(nrows, n))
        elif not (self.active and key0 in self.fresh) or job0.src_ptr != buf.data_ptr():
            job0.snap.copy_(buf.detach().reshape(1, n).expand(nrows, n))
        return job0.snap

    def refresh(self):
        """all registered copies in one launch"""
        for key in [k for k, j in self.jobs.items() if j.ref() is None]:
            del self.jobs[key]
            self.dirty = True
        for j in self.jobs.values():
            ptr = j.ref().data_ptr()
            if ptr != j.src_ptr:
                j.src_ptr, self.dirty = ptr, True
        if not self.jobs:
            return
        if self.dirty or self.table is None:
            if torch.cuda.is_current_stream_capturing():
                # the job table is uploaded from the host, which a capturing stream cannot do: this forward copies per
                # `get` instead (run one eager forward before capturing and the table is already in place)
                self.fresh = set()
                return
            rows = [[j.src_ptr, j.dst.data_ptr(), j.rows, j.cols, j.dst.shape[1], int(j.transpose)] for j in self.jobs.values()]
            self.table = torch.tensor(rows, dtype=torch.int64, device=self.dev)       # only while the job set changes
            self.tables.append(self.table)
            self.dirty = False
        with torch.cuda.device(self.dev):
            _call("prep_weights", 0.0, capi.load().o3d_prep_weights, self.table.data_ptr(), self.table.shape[0],
                  torch.cuda.current_stream(self.dev).cuda_stream)
        self.fresh = set(self.jobs)


_PREP = {}           # device -> the default table (used outside any scope: copies on the spot, never launches a table)
_OWNED = weakref.WeakKeyDictionary()      # owner module -> {device: WeightPrep}
_ACTIVE = {}         # device -> the WeightPrep of the scope that is open on it


def _dev_key(dev):
    return (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())


def prep_for(dev):
    """the weight-preparation table in force on `dev`: the open scope's (its owner's own table), else the device's default"""
    key = _dev_key(dev)
    if key in _ACTIVE:
        return _ACTIVE[key]
    if key not in _PREP:
        _PREP[key] = WeightPrep(torch.device(*key))
    return _PREP[key]


@contextlib.contextmanager
def prep_scope(dev, owner=None):
    """Between entry and exit the weights do not change (one forward of a tracker): every prepared copy is refreshed by
    one launch now and `get` hands the buffers out without further copies.  owner: the module whose forward this is -- it
    gets a table of ITS OWN (round 6).  One table per device served every model on it: a second model's first forward
    rebuilt (and freed) the table a captured step of the first model still launched with, and a model that died left
    rows pointing at freed buffers in the table of the one that lived on (memory fault in a test with two live trainers)."""
    if dev.type != "cuda" or not _ON["on"]:
        yield
        return
    key = _dev_key(dev)
    if key in _ACTIVE:          # nested scopes: the outer one did the work
        yield
        return
    if owner is not None:
        per_dev = _OWNED.setdefault(owner, {})
        if key not in per_dev:
            per_dev[key] = WeightPrep(torch.device(*key))
        prep = per_dev[key]
    else:
        prep = prep_for(dev)
    prep.refresh()
    prep.active = True
    _ACTIVE[key] = prep
    own_counters = counters_begin()
    try:
        yield
    finally:
        prep.active = False
        _ACTIVE.pop(key, None)
        if own_counters:
            counters_end()


# ---- the stack ---------------------------------------------------------------------------------------------
class _Cfg:
    __slots__ = ("bns", "training", "residual", "nsrc")


def chain_supported(sources, units):
    """sources: [(B,C_i,N) GPU tensors]; units: [(conv1d, batchnorm1d | None, activation | None)]"""
    if not _ON["on"] or not units:
        return False
    x = sources[0]
    if not x.is_cuda or any(t.dtype != torch.float32 or t.dim() != 3 or t.shape[0] != x.shape[0] or
                            t.shape[2] != x.shape[2] for t in sources) or len(sources) > 4:
        return False
    B, _, N = x.shape
    P = B * N
    if P == 0 or P % 64 or (P > 65536 and P % 128):
        return False
    if sum(t.shape[1] for t in sources) != units[0][0].in_channels:
        return False
    for i, (conv, bn, act) in enumerate(units):
        if conv.kernel_size != (1,) or conv.stride != (1,) or conv.padding != (0,) or conv.groups != 1:
            return False
        last = i == len(units) - 1
        if last:
            if bn is not None or act is not None:
                return False
        else:
            if bn is None or not isinstance(act, torch.nn.ReLU) or conv.bias is not None or conv.out_channels % 64 or \
                    not bn.affine or not bn.track_running_stats or bn.momentum is None:
                return False
    return True


# ---- launch plumbing: a stack's forward / backward is a GENERATOR of launches --------------------------------------------
# (name, flops, C entry, argument list, dims).  Driven alone, every launch is issued as it comes; two stacks driven side
# by side (`run_chain_pair`: FC_layer_cla and vote_layer read the same seeds, models/head/rpn.py:44-54) advance in
# lockstep and a pair of launches of the same kind goes out as ONE launch (csrc/mlp_direct.hip::direct_gemm_pair_kernel,
# csrc/mlp.hip::bn_*finalize_pair_kernel) -- the serial chain of ~15 us sub-round launches is then half as long.
class _PwFwdArgs(ctypes.Structure):       # o3d_pw_fwd_args
    _fields_ = [("X", _vp), ("W", _vp), ("in_scale", _vp), ("in_shift", _vp), ("bias", _vp), ("resid", _vp), ("Cin", _i),
                ("Cout", _i), ("P", _l), ("Y", _vp), ("part", _vp), ("stat_c", _vp)]


class _PwDgradArgs(ctypes.Structure):     # o3d_pw_dgrad_args
    _fields_ = [("dN", _vp), ("Y", _vp), ("A1", _vp), ("A2", _vp), ("A3", _vp), ("Wt", _vp), ("Cin", _i), ("Cout", _i),
                ("P", _l), ("Yprev", _vp), ("scale_p", _vp), ("shift_p", _vp), ("mean_p", _vp), ("resid", _vp),
                ("dNprev", _vp), ("part", _vp)]


class _BnFinArgs(ctypes.Structure):       # o3d_bn_fin_args
    _fields_ = [("part", _vp), ("nparts", _i), ("C", _i), ("count", _d), ("stat_c", _vp), ("gamma", _vp), ("beta", _vp),
                ("running_mean", _vp), ("running_var", _vp), ("momentum", _f), ("eps", _f), ("mean", _vp), ("invstd", _vp),
                ("scale", _vp), ("shift", _vp)]


class _BnBwdFinArgs(ctypes.Structure):    # o3d_bn_bwd_fin_args
    _fields_ = [("part", _vp), ("nparts", _i), ("C", _i), ("count", _d), ("gamma", _vp), ("mean", _vp), ("invstd", _vp),
                ("dgamma", _vp), ("dbeta", _vp), ("A1", _vp), ("A2", _vp), ("A3", _vp)]


for _n in ("o3d_pw_fwd_pair", "o3d_pw_dgrad_pair", "o3d_bn_finalize_pair", "o3d_bn_bwd_finalize_pair"):
    capi.register(_n, [_vp, _vp, _vp])
# single entry -> (pair entry, argument struct, trailing arguments of the single call that the struct does not carry)
_PAIRABLE = {"o3d_pw_fwd": ("o3d_pw_fwd_pair", _PwFwdArgs, 1), "o3d_pw_dgrad": ("o3d_pw_dgrad_pair", _PwDgradArgs, 1),
             "o3d_bn_finalize": ("o3d_bn_finalize_pair", _BnFinArgs, 2),
             "o3d_bn_bwd_finalize": ("o3d_bn_bwd_finalize_pair", _BnBwdFinArgs, 2)}


def _drive(gens):
    """run the launch generators to completion -> their return values; two generators advance in lockstep and matching
    pairable launches are merged"""
    lib = capi.load()
    n = len(gens)
    pend, done, res = [None] * n, [False] * n, [None] * n

    def advance(k):
        try:
            pend[k] = next(gens[k])
        except StopIteration as e:
            pend[k], done[k], res[k] = None, True, e.value

    def single(k):
        name, flops, entry, args, dims = pend[k]
        _call(name, flops, getattr(lib, entry), *args, dims=dims)
        advance(k)
    for k in range(n):
        advance(k)
    while not all(done):
        live = [k for k in range(n) if not done[k]]
        if len(live) == 2 and pend[0][2] == pend[1][2] and pend[0][2] in _PAIRABLE and pend[0][3][-1] == pend[1][3][-1]:
            pair_entry, struct, tail = _PAIRABLE[pend[0][2]]
            sa, sb = struct(*pend[0][3][:-tail]), struct(*pend[1][3][:-tail])
            _call(pend[0][0], pend[0][1] + pend[1][1], getattr(lib, pair_entry), ctypes.addressof(sa), ctypes.addressof(sb),
                  pend[0][3][-1])
            advance(0)
            advance(1)
        elif len(live) == 2:
            # not aligned: issue the launch that has no partner (a pack, a row sum ...) or, failing that, the first one
            k = next((k for k in live if pend[k][2] not in _PAIRABLE), live[0])
            single(k)
        else:
            single(live[0])
    return res


class _State:
    pass


def _chain_forward(cfg, tensors, need_bwd):
    """generator: launches of one stack's forward; returns (output view (B,Cout,N), state for the backward | None)"""
    lib = capi.load()
    srcs, params = tensors[:cfg.nsrc], tensors[cfg.nsrc:]
    L = len(params) // 4
    B, _, N = srcs[0].shape
    P = B * N
    dev, f32 = srcs[0].device, torch.float32
    st = _stream()
    prep = prep_for(dev)
    K0 = sum(t.shape[1] for t in srcs)
    K0p = _up(K0, 64)
    X0 = _as_flat(srcs[0].detach()) if (cfg.nsrc == 1 and K0 == K0p) else None
    if X0 is None:
        X0 = pack_rows([t.detach() for t in srcs], K0p)
    Ys, vecs, Wts = [], [], []
    Kp = K0p
    for l in range(L):
        W, bias, gamma, beta = params[4 * l:4 * l + 4]
        Cout = W.shape[0]
        Mp = _up(Cout, 64)
        Wp = prep.get(W, Mp, Kp)
        if need_bwd:
            Wts.append(prep.get(W, Kp, Mp, transpose=True))
        src = X0 if l == 0 else Ys[-1]
        sc = None if l == 0 else vecs[-1][2].data_ptr()
        sh = None if l == 0 else vecs[-1][3].data_ptr()
        Y = torch.empty((Mp, P), device=dev, dtype=f32)
        if l < L - 1:
            bn = cfg.bns[l]
            vec = torch.empty((4, Mp), device=dev, dtype=f32)            # mean, invstd, scale, shift
            if cfg.training:
                nparts = P // lib.o3d_pw_tile(P, Mp)
                part = torch.empty((nparts, 2, Mp), device=dev, dtype=f32)
                yield ("pw_conv_fwd", 2.0 * Kp * Mp * P, "o3d_pw_fwd", [src.data_ptr(), Wp.data_ptr(), sc, sh, None, None, Kp, Mp,
                                                                       P, Y.data_ptr(), part.data_ptr(),
                                                                       bn.running_mean.data_ptr(), st], (Kp, Mp))
                fold = torch.empty((64, Mp), device=dev, dtype=f32)
                yield ("bn_finalize", 0.0, "o3d_bn_finalize", [part.data_ptr(), nparts, Mp, float(P), bn.running_mean.data_ptr(),
                                                               gamma.data_ptr(), beta.data_ptr(), bn.running_mean.data_ptr(),
                                                               bn.running_var.data_ptr(), float(bn.momentum), float(bn.eps),
                                                               vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
                                                               vec[3].data_ptr(), fold.data_ptr(), st], None)
            else:
                yield ("pw_conv_fwd", 2.0 * Kp * Mp * P, "o3d_pw_fwd", [src.data_ptr(), Wp.data_ptr(), sc, sh, None, None, Kp, Mp,
                                                                       P, Y.data_ptr(), None, None, st], (Kp, Mp))
                _eval_consts(lib, bn, gamma, beta, vec, 1, st)
            vecs.append(vec)
        else:
            bp = prep.get(bias, 1, Mp) if bias is not None else None
            if cfg.residual and Mp != K0p:
                raise ValueError("residual stack: padded output rows %d != padded input rows %d" % (Mp, K0p))
            if bp is None and not cfg.residual:       # plain store without statistics
                bp = _const_vec(dev, Mp, 0.0)
            yield ("pw_conv_fwd", 2.0 * Kp * Mp * P, "o3d_pw_fwd", [src.data_ptr(), Wp.data_ptr(), sc, sh, _ptr(bp),
                                                                   X0.data_ptr() if cfg.residual else None, Kp, Mp, P,
                                                                   Y.data_ptr(), None, None, st], (Kp, Mp))
        Ys.append(Y)
        Kp = Mp
    if cfg.training and L > 1:
        count_batches(cfg.bns[:L - 1], 1)
    Cl = params[4 * (L - 1)].shape[0]
    state = None
    if need_bwd:
        state = _State()
        state.cfg = cfg
        state.geom = (B, N, L, K0, K0p, [t.shape[1] for t in srcs])
        state.versions = [(p, p._version) for p in params if p is not None]
        state.saved = (X0, Ys, vecs, Wts, [params[4 * l + 2] for l in range(L)], [params[4 * l] for l in range(L)])
        state.biases = [params[4 * l + 1] for l in range(L)]
    return Ys[-1][:Cl].view(Cl, B, N).permute(1, 0, 2), state


def _chain_backward(state, dOut, needs):
    """generator: launches of one stack's backward; `needs` = needs_input_grad of (sources..., 4 parameters per layer);
    returns the list of gradients in that order"""
    lib = capi.load()
    cfg = state.cfg
    B, N, L, K0, K0p, src_C = state.geom
    for p, v in state.versions:
        if p._version != v:
            raise RuntimeError("a parameter of a fused conv stack was modified in place between forward and backward")
    X0, Ys, vecs, Wts, gammas, Ws = state.saved
    P = B * N
    dev, f32 = dOut.device, torch.float32
    st = _stream()
    nparts = 0
    Cl = Ws[-1].shape[0]
    Mp = Ys[-1].shape[0]
    G = _as_flat(dOut) if Cl == Mp else None
    if G is None:
        G = pack_rows([dOut], Mp)
    grads = [None] * (4 * L)
    want_x = any(needs[:cfg.nsrc])
    one, zero = _const_vec(dev, Mp, 1.0), _const_vec(dev, Mp, 0.0)
    dX0 = None

    keep = []            # what the (possibly deferred) grouped launch reads or writes: alive until it has run
    jobs = []            # grouped weight gradients: launched together behind the data-gradient chain

    def wgrad(l, dN, Y, A, Cout_p, coef=None):
        Xs = X0 if l == 0 else Ys[l - 1]
        Kp = Xs.shape[0]
        sc = None if l == 0 else vecs[l - 1][2].data_ptr()
        sh = None if l == 0 else vecs[l - 1][3].data_ptr()
        Wl = Ws[l]
        Cout, Cin = Wl.shape[0], Wl.shape[1]
        # the gradient in the parameter's OWN (Cout, Cin) shape, written compactly by the group's reduction: what
        # autograd receives is contiguous (a slice of the padded buffer is cloned by AccumulateGrad -- one more launch,
        # and with deferred launches a clone of a buffer that is not filled yet)
        dW = torch.empty((Cout, Cin), device=dev, dtype=f32)
        scratch = torch.empty((lib.o3d_mlp_conv_wgrad2_scratch(1, Kp, Cout_p, P),), device=dev, dtype=f32)
        jobs.append((2.0 * Kp * Cout_p * P, (dN.data_ptr(), Y.data_ptr(), A[0], A[1], A[2], Xs.data_ptr(), sc, sh, Kp,
                                             Cout_p, P, scratch.data_ptr(), dW.data_ptr(), Cout, Cin)))
        # everything the (possibly deferred) launch reads: the layer input and its BatchNorm constants belong to the
        # autograd node, which is released -- and its memory reused -- as soon as this backward returns
        keep.extend((dN, Y, scratch, dW, coef, Xs, vecs[l - 1] if l > 0 else None))
        return dW.view(Wl.shape)

    # ---- last layer: plain conv (+ bias, + residual)
    l = L - 1
    if needs[cfg.nsrc + 4 * l + 1]:
        db = torch.empty((Mp,), device=dev, dtype=f32)       # the bias gradient rides in the stack's grouped weight-gradient launch
        jobs.append((0.0, (G.data_ptr(), None, None, None, None, None, None, None, 0, Mp, P, None, db.data_ptr(), 0, 0)))
        keep.extend((G, db))
        grads[4 * l + 1] = db[:Cl]
    grads[4 * l] = wgrad(l, G, G, (one.data_ptr(), zero.data_ptr(), zero.data_ptr()), Mp)
    dN, part = None, None
    if L > 1:
        Cp = Ys[l - 1].shape[0]
        dN = torch.empty((Cp, P), device=dev, dtype=f32)
        nparts = P // lib.o3d_pw_tile(P, Cp)
        part = torch.empty((nparts, 2, Cp), device=dev, dtype=f32)
        v = vecs[l - 1]
        yield ("pw_conv_dgrad", 2.0 * Cp * Mp * P, "o3d_pw_dgrad", [G.data_ptr(), None, None, None, None, Wts[l].data_ptr(), Cp,
                                                                   Mp, P, Ys[l - 1].data_ptr(), v[2].data_ptr(), v[3].data_ptr(),
                                                                   v[0].data_ptr(), None, dN.data_ptr(), part.data_ptr(), st],
               (Cp, Mp, True))
    elif want_x:
        dX0 = torch.empty((K0p, P), device=dev, dtype=f32)
        yield ("pw_conv_dgrad", 2.0 * K0p * Mp * P, "o3d_pw_dgrad", [G.data_ptr(), None, None, None, None, Wts[0].data_ptr(), K0p,
                                                                    Mp, P, None, None, None, None,
                                                                    G.data_ptr() if cfg.residual else None, dX0.data_ptr(), None,
                                                                    st], (K0p, Mp, True))
    # ---- hidden layers: conv -> BatchNorm -> ReLU
    for l in range(L - 2, -1, -1):
        Cp = Ys[l].shape[0]
        v = vecs[l]
        coef = torch.empty((5, Cp), device=dev, dtype=f32)          # dgamma dbeta A1 A2 A3
        fold = torch.empty((64, Cp), device=dev, dtype=f32)
        yield ("bn_bwd_finalize", 0.0, "o3d_bn_bwd_finalize", [part.data_ptr(), nparts, Cp, float(P), gammas[l].data_ptr(),
                                                               v[0].data_ptr(), v[1].data_ptr(), coef[0].data_ptr(),
                                                               coef[1].data_ptr(), coef[2].data_ptr(), coef[3].data_ptr(),
                                                               coef[4].data_ptr(), fold.data_ptr(), st], None)
        if not cfg.training:
            coef[3].zero_()
            coef[4].zero_()
        grads[4 * l + 2], grads[4 * l + 3] = coef[0], coef[1]
        A = (coef[2].data_ptr(), coef[3].data_ptr(), coef[4].data_ptr())
        grads[4 * l] = wgrad(l, dN, Ys[l], A, Cp, coef)
        if l > 0:
            Cq = Ys[l - 1].shape[0]
            dNp = torch.empty((Cq, P), device=dev, dtype=f32)
            nparts_next = P // lib.o3d_pw_tile(P, Cq)
            part_next = torch.empty((nparts_next, 2, Cq), device=dev, dtype=f32)
            vp = vecs[l - 1]
            yield ("pw_conv_dgrad", 2.0 * Cq * Cp * P, "o3d_pw_dgrad", [dN.data_ptr(), Ys[l].data_ptr(), A[0], A[1], A[2],
                                                                       Wts[l].data_ptr(), Cq, Cp, P, Ys[l - 1].data_ptr(),
                                                                       vp[2].data_ptr(), vp[3].data_ptr(), vp[0].data_ptr(), None,
                                                                       dNp.data_ptr(), part_next.data_ptr(), st], (Cq, Cp))
            keep.append(dN)
            dN, part, nparts = dNp, part_next, nparts_next
        elif want_x:
            dX0 = torch.empty((K0p, P), device=dev, dtype=f32)
            yield ("pw_conv_dgrad", 2.0 * K0p * Cp * P, "o3d_pw_dgrad", [dN.data_ptr(), Ys[0].data_ptr(), A[0], A[1], A[2],
                                                                        Wts[0].data_ptr(), K0p, Cp, P, None, None, None, None,
                                                                        G.data_ptr() if cfg.residual else None, dX0.data_ptr(),
                                                                        None, st], (K0p, Cp))
    _submit_jobs(jobs, keep, st, list(Ws) + [b_ for b_ in state.biases if b_ is not None])
    del keep
    gsrc, off = [], 0
    for i, C in enumerate(src_C):
        gsrc.append(dX0[off:off + C].view(C, B, N).permute(1, 0, 2) if (dX0 is not None and needs[i]) else None)
        off += C
    return gsrc + grads


class FlatChain(torch.autograd.Function):
    """apply(cfg, src_0..src_{nsrc-1}, [W, bias | None, gamma | None, beta | None] per layer) -> (B, Cout, N) view of the
    flat (Cout_pad, B*N) output.  Hidden layers: conv -> BatchNorm -> ReLU; last layer: conv + bias (+ residual)."""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, cfg, *tensors):
        out, ctx.state = _drive([_chain_forward(cfg, tensors, any(ctx.needs_input_grad))])[0]
        return out

    @staticmethod
    @capi.on_tensor_device
    def backward(ctx, dOut):
        return (None, *_drive([_chain_backward(ctx.state, dOut, ctx.needs_input_grad[1:])])[0])


class FlatChainPair(torch.autograd.Function):
    """two independent stacks over the same columns, advanced side by side: apply(cfg_a, cfg_b, n_a, tensors_a...,
    tensors_b...) -> (out_a, out_b), each as FlatChain.apply(cfg_x, *tensors_x) would return it (same launches' worth of
    arithmetic, the same numbers; launches of the same kind merged two by two)"""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, cfg_a, cfg_b, n_a, *tensors):
        ta, tb = tensors[:n_a], tensors[n_a:]
        na, nb = ctx.needs_input_grad[3:3 + n_a], ctx.needs_input_grad[3 + n_a:]
        (oa, sa), (ob, sb) = _drive([_chain_forward(cfg_a, ta, any(na)), _chain_forward(cfg_b, tb, any(nb))])
        ctx.states, ctx.n_a = (sa, sb), n_a
        return oa, ob

    @staticmethod
    @capi.on_tensor_device
    def backward(ctx, da, db):
        sa, sb = ctx.states
        na, nb = ctx.needs_input_grad[3:3 + ctx.n_a], ctx.needs_input_grad[3 + ctx.n_a:]
        gens, order = [], []
        for state, d, needs, n in ((sa, da, na, ctx.n_a), (sb, db, nb, len(ctx.needs_input_grad) - 3 - ctx.n_a)):
            if state is None:
                order.append([None] * n)
                continue
            if d is None:      # this output did not reach the loss
                B, N, L = state.geom[:3]
                d = torch.zeros((B, state.saved[5][-1].shape[0], N), device=state.saved[0].device, dtype=torch.float32)
            order.append(len(gens))
            gens.append(_chain_backward(state, d, needs))
        res = _drive(gens) if gens else []
        out = []
        for o in order:
            out += o if isinstance(o, list) else res[o]
        return (None, None, None, *out)


def _chain_cfg(sources, units, residual):
    cfg = _Cfg()
    cfg.nsrc = len(sources)
    cfg.training = bool(units[0][1].training) if units[0][1] is not None else False
    cfg.residual = bool(residual)
    cfg.bns = [bn for _, bn, _ in units]
    params = []
    for conv, bn, _ in units:
        params += [conv.weight, conv.bias, bn.weight if bn is not None else None, bn.bias if bn is not None else None]
    return cfg, params


def run_chain(sources, units, residual=False):
    """sources [(B,C_i,N)] stacked along the channels -> the Conv1d stack `units` -> (B,Cout,N) (+ sources when
    `residual`).  Caller has checked chain_supported."""
    cfg, params = _chain_cfg(sources, units, residual)
    return FlatChain.apply(cfg, *sources, *params)


def run_chain_pair(a, b):
    """a, b = (sources, units, residual) of two independent stacks over the same (B, N): -> (out_a, out_b), launches
    merged two by two where they match.  Caller has checked chain_supported for both."""
    cfg_a, pa = _chain_cfg(*a)
    cfg_b, pb = _chain_cfg(*b)
    if a[0][0].shape[0] != b[0][0].shape[0] or a[0][0].shape[2] != b[0][0].shape[2]:
        return FlatChain.apply(cfg_a, *a[0], *pa), FlatChain.apply(cfg_b, *b[0], *pb)
    ta = (*a[0], *pa)
    return FlatChainPair.apply(cfg_a, cfg_b, len(ta), *ta, *b[0], *pb)


class SharedConvPair(torch.autograd.Function):
    """One nn.Conv1d (kernel 1, bias) applied to TWO sets of clouds of different sizes -- `conv_final` on the template
    and on the search feature (models/bat.py:91-92, models/p2b.py:35-36) -- as ONE GEMM over the columns of both:
    apply(W (Cout,Cin,1), bias, xa (B,Cin,Na), xb (B,Cin,Nb)) -> (ya (B,Cout,Na), yb (B,Cout,Nb)), views of one flat
    (Cout, B*Na + B*Nb) output.  Backward: one data gradient, one weight gradient (the sum over both sets: the weights
    are shared) and one bias row sum, the latter two as jobs of the grouped weight-gradient launch."""

    @staticmethod
    @capi.on_tensor_device
    def forward(ctx, W, bias, xa, xb):
        lib = capi.load()
        B, Cin, Na = xa.shape
        Nb = xb.shape[2]
        Cout = W.shape[0]
        Pa, Pb = B * Na, B * Nb
        P = Pa + Pb
        dev, f32 = xa.device, torch.float32
        st = _stream()
        prep = prep_for(dev)
        X0 = torch.empty((Cin, P), device=dev, dtype=f32)
        pack_rows_into(xa.detach(), X0, 0)
        pack_rows_into(xb.detach(), X0, Pa)
        Wp = prep.get(W, Cout, Cin)
        bp = prep.get(bias, 1, Cout) if bias is not None else _const_vec(dev, Cout, 0.0)
        Y = torch.empty((Cout, P), device=dev, dtype=f32)
        _call("pw_conv_fwd", 2.0 * Cin * Cout * P, lib.o3d_pw_fwd, X0.data_ptr(), Wp.data_ptr(), None, None, bp.data_ptr(), None,
              Cin, Cout, P, Y.data_ptr(), None, None, st, dims=(Cin, Cout))
        if any(ctx.needs_input_grad):
            ctx.saved = (X0, prep.get(W, Cin, Cout, transpose=True), W, bias)
            ctx.geom = (B, Cin, Cout, Na, Nb)
            ctx.versions = [(p_, p_._version) for p_ in (W, bias) if p_ is not None]
        return Y[:, :Pa].view(Cout, B, Na).permute(1, 0, 2), Y[:, Pa:].view(Cout, B, Nb).permute(1, 0, 2)

    @staticmethod
    @capi.on_tensor_device
    def backward(ctx, da, db):
        lib = capi.load()
        X0, Wt, W, bias = ctx.saved
        B, Cin, Cout, Na, Nb = ctx.geom
        for p_, v in ctx.versions:
            if p_._version != v:
                raise RuntimeError("a parameter of a fused conv stack was modified in place between forward and backward")
        Pa, Pb = B * Na, B * Nb
        P = Pa + Pb
        dev, f32 = X0.device, torch.float32
        st = _stream()
        G = torch.empty((Cout, P), device=dev, dtype=f32)
        for d, col0, n in ((da, 0, Na), (db, Pa, Nb)):
            if d is None:
                G[:, col0:col0 + B * n].zero_()
            else:
                pack_rows_into(d, G, col0)
        dxa = dxb = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            dX = torch.empty((Cin, P), device=dev, dtype=f32)
            _call("pw_conv_dgrad", 2.0 * Cin * Cout * P, lib.o3d_pw_dgrad, G.data_ptr(), None, None, None, None, Wt.data_ptr(), Cin,
                  Cout, P, None, None, None, None, None, dX.data_ptr(), None, st, dims=(Cin, Cout, True))
            dxa = dX[:, :Pa].view(Cin, B, Na).permute(1, 0, 2) if ctx.needs_input_grad[2] else None
            dxb = dX[:, Pa:].view(Cin, B, Nb).permute(1, 0, 2) if ctx.needs_input_grad[3] else None
        one, zero = _const_vec(dev, Cout, 1.0), _const_vec(dev, Cout, 0.0)
        dW = torch.empty((Cout, Cin), device=dev, dtype=f32)
        scratch = torch.empty((lib.o3d_mlp_conv_wgrad2_scratch(1, Cin, Cout, P),), device=dev, dtype=f32)
        jobs = [(2.0 * Cin * Cout * P, (G.data_ptr(), G.data_ptr(), one.data_ptr(), zero.data_ptr(), zero.data_ptr(),
                                       X0.data_ptr(), None, None, Cin, Cout, P, scratch.data_ptr(), dW.data_ptr(), 0, 0))]
        dbias = None
        if bias is not None and ctx.needs_input_grad[1]:
            dbias = torch.empty((Cout,), device=dev, dtype=f32)
            jobs.append((0.0, (G.data_ptr(), None, None, None, None, None, None, None, 0, Cout, P, None, dbias.data_ptr(), 0, 0)))
        _submit_jobs(jobs, [G, X0, scratch, dW, dbias, one, zero], st, [W, bias])
        # fresh views: autograd must hold the ONLY reference to what it is handed, or AccumulateGrad clones it -- with
        # deferred launches a clone of a buffer the flush has not filled yet (round-3 advisor finding on dbias)
        return dW.view(W.shape), dbias.view(Cout) if dbias is not None else None, dxa, dxb


def shared_conv_pair_supported(conv, xa, xb):
    return (_ON["on"] and xa.is_cuda and xa.dtype == torch.float32 and xb.dtype == torch.float32 and
            xa.dim() == 3 and xb.dim() == 3 and xa.shape[0] == xb.shape[0] and xa.shape[1] == xb.shape[1] == conv.in_channels and
            conv.kernel_size == (1,) and conv.stride == (1,) and conv.padding == (0,) and conv.groups == 1 and
            conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0 and
            (xa.shape[0] * (xa.shape[2] + xb.shape[2])) % 128 == 0)
