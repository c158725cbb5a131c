"""The nine `pointnet2_ops._ext` operators on torch tensors, backed by libo3dsot_hip.so.

Drop-in for the pybind module the reference imports at
pointnet2/utils/pointnet2_utils.py:17 -- same names, positional arguments, shapes and
dtypes as its call sites (:56,:92,:98,:125,:162,:184,:217,:237,:268) and the upstream
error behaviour: tensors must be on the GPU ("CPU not supported"), contiguous, fp32 /
int32, otherwise RuntimeError.  Outputs are freshly allocated on the input device; the
kernels are enqueued on torch's current stream without any host synchronisation.
"""
import torch

from . import capi


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk_f(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s: CPU not supported (tensor must be a GPU tensor)" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if t.dtype != torch.float32:
        raise RuntimeError("%s must be a float tensor" % name)


def _chk_i(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s: CPU not supported (tensor must be a GPU tensor)" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if t.dtype != torch.int32:
        raise RuntimeError("%s must be an int tensor" % name)


def furthest_point_sampling(xyz, npoint, _variant="dpp"):
    """xyz (B,N,3) f32 -> (B,npoint) i32   [pointnet2_utils.py:56]"""
    _chk_f(xyz, "xyz")
    B, N, _ = xyz.shape
    npoint = int(npoint)
    out = torch.empty((B, npoint), dtype=torch.int32, device=xyz.device)
    temp = torch.empty((B, N), dtype=torch.float32, device=xyz.device) if N > 8192 else None
    lib = capi.load()
    fn = lib.o3d_furthest_point_sampling if _variant == "dpp" else lib.o3d_furthest_point_sampling_shfl
    with torch.cuda.device(xyz.device):
        capi.check(fn(xyz.data_ptr(), B, N, npoint, temp.data_ptr() if temp is not None else None,
                      out.data_ptr(), _stream()), "furthest_point_sampling")
    return out


def furthest_point_sampling_pair(xyz_a, npoint_a, xyz_b, npoint_b):
    """furthest_point_sampling of two independent sets of clouds (same batch size) in one launch -> (idx_a, idx_b),
    each bit-identical to the single call; falls back to two launches beyond 2048 points per cloud."""
    _chk_f(xyz_a, "xyz_a")
    _chk_f(xyz_b, "xyz_b")
    B, Na, _ = xyz_a.shape
    Nb = xyz_b.shape[1]
    if xyz_b.shape[0] != B or max(Na, Nb) > 2048 or min(int(npoint_a), int(npoint_b)) < 1 or B == 0:
        return furthest_point_sampling(xyz_a, npoint_a), furthest_point_sampling(xyz_b, npoint_b)
    out_a = torch.empty((B, int(npoint_a)), dtype=torch.int32, device=xyz_a.device)
    out_b = torch.empty((B, int(npoint_b)), dtype=torch.int32, device=xyz_a.device)
    with torch.cuda.device(xyz_a.device):
        capi.check(capi.load().o3d_furthest_point_sampling_pair(xyz_a.data_ptr(), Na, int(npoint_a), out_a.data_ptr(),
                                                                xyz_b.data_ptr(), Nb, int(npoint_b), out_b.data_ptr(), B,
                                                                _stream()), "furthest_point_sampling_pair")
    return out_a, out_b


def gather_points(features, idx):
    """features (B,C,N), idx (B,npoint) -> (B,C,npoint)   [pointnet2_utils.py:92]"""
    _chk_f(features, "features")
    _chk_i(idx, "idx")
    B, C, N = features.shape
    npoint = idx.shape[1]
    out = torch.empty((B, C, npoint), dtype=torch.float32, device=features.device)
    with torch.cuda.device(features.device):
        capi.check(capi.load().o3d_gather_points(features.data_ptr(), idx.data_ptr(), B, C, N, npoint,
                                                 out.data_ptr(), _stream()), "gather_points")
    return out


def gather_rows(src, idx):
    """src (B,N,D) point-major, idx (B,npoint) -> (B,npoint,D): out[b,j] = src[b,idx[b,j]]"""
    _chk_f(src, "src")
    _chk_i(idx, "idx")
    B, N, D = src.shape
    npoint = idx.shape[1]
    out = torch.empty((B, npoint, D), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        capi.check(capi.load().o3d_gather_rows(src.data_ptr(), idx.data_ptr(), B, N, D, npoint, out.data_ptr(),
                                               _stream()), "gather_rows")
    return out


def gather_rows2(a, b, idx, npoint):
    """a (B,N,Da), b (B,N,Db) | None, idx (B, >= npoint) int32 (row stride free) -> a[b, idx[b, :npoint]] (B,npoint,Da)
    [, the same of b]: the trackers' label re-indexing by the sampling indices in one launch (no gradient: labels)"""
    _chk_f(a, "a")
    if not (idx.dtype == torch.int32 and idx.dim() == 2 and idx.stride(1) == 1 and idx.shape[1] >= npoint and idx.is_cuda):
        raise RuntimeError("gather_rows2: idx must be a CUDA int32 (B, >= npoint) tensor with unit inner stride")
    B, N, Da = a.shape
    Db = 0
    if b is not None:
        _chk_f(b, "b")
        Db = b.shape[2]
    outa = torch.empty((B, npoint, Da), dtype=torch.float32, device=a.device)
    outb = torch.empty((B, npoint, Db), dtype=torch.float32, device=a.device) if b is not None else None
    with torch.cuda.device(a.device):
        capi.check(capi.load().o3d_gather_rows2(a.data_ptr(), Da, b.data_ptr() if b is not None else None, Db, idx.data_ptr(),
                                                idx.stride(0), B, N, npoint, outa.data_ptr(),
                                                outb.data_ptr() if outb is not None else None, _stream()), "gather_rows2")
    return outa, outb


def gather_points_grad(grad_out, idx, n):
    """grad_out (B,C,npoint), idx (B,npoint) -> (B,C,n)   [pointnet2_utils.py:98]"""
    _chk_f(grad_out, "grad_out")
    _chk_i(idx, "idx")
    B, C, npoint = grad_out.shape
    n = int(n)
    out = torch.empty((B, C, n), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        capi.check(capi.load().o3d_gather_points_grad(grad_out.data_ptr(), idx.data_ptr(), B, C, n, npoint,
                                                      out.data_ptr(), _stream()), "gather_points_grad")
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    """new_xyz (B,npoint,3), xyz (B,N,3) -> (B,npoint,nsample) i32   [pointnet2_utils.py:268]"""
    _chk_f(new_xyz, "new_xyz")
    _chk_f(xyz, "xyz")
    B, N, _ = xyz.shape
    npoint = new_xyz.shape[1]
    nsample = int(nsample)
    out = torch.empty((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        capi.check(capi.load().o3d_ball_query(new_xyz.data_ptr(), xyz.data_ptr(), B, N, npoint, float(radius),
                                              nsample, out.data_ptr(), _stream()), "ball_query")
    return out


def group_points(features, idx):
    """features (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample)   [pointnet2_utils.py:217]"""
    _chk_f(features, "features")
    _chk_i(idx, "idx")
    B, C, N = features.shape
    _, npoint, nsample = idx.shape
    out = torch.empty((B, C, npoint, nsample), dtype=torch.float32, device=features.device)
    with torch.cuda.device(features.device):
        capi.check(capi.load().o3d_group_points(features.data_ptr(), idx.data_ptr(), B, C, N, npoint, nsample,
                                                out.data_ptr(), _stream()), "group_points")
    return out


def group_points_grad(grad_out, idx, n):
    """grad_out (B,C,npoint,nsample), idx -> (B,C,n)   [pointnet2_utils.py:237]"""
    _chk_f(grad_out, "grad_out")
    _chk_i(idx, "idx")
    B, C, npoint, nsample = grad_out.shape
    n = int(n)
    out = torch.empty((B, C, n), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        capi.check(capi.load().o3d_group_points_grad(grad_out.data_ptr(), idx.data_ptr(), B, C, n, npoint,
                                                     nsample, out.data_ptr(), _stream()), "group_points_grad")
    return out


def three_nn(unknown, known):
    """unknown (B,n,3), known (B,m,3) -> (dist2 (B,n,3) f32, idx (B,n,3) i32)   [pointnet2_utils.py:125]"""
    _chk_f(unknown, "unknown")
    _chk_f(known, "known")
    B, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = torch.empty((B, n, 3), dtype=torch.float32, device=unknown.device)
    idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknown.device)
    with torch.cuda.device(unknown.device):
        capi.check(capi.load().o3d_three_nn(unknown.data_ptr(), known.data_ptr(), B, n, m, dist2.data_ptr(),
                                            idx.data_ptr(), _stream()), "three_nn")
    return dist2, idx


def three_interpolate(features, idx, weight):
    """features (B,c,m), idx (B,n,3), weight (B,n,3) -> (B,c,n)   [pointnet2_utils.py:162]"""
    _chk_f(features, "features")
    _chk_i(idx, "idx")
    _chk_f(weight, "weight")
    B, c, m = features.shape
    n = idx.shape[1]
    out = torch.empty((B, c, n), dtype=torch.float32, device=features.device)
    with torch.cuda.device(features.device):
        capi.check(capi.load().o3d_three_interpolate(features.data_ptr(), idx.data_ptr(), weight.data_ptr(),
                                                     B, c, m, n, out.data_ptr(), _stream()), "three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """grad_out (B,c,n), idx, weight -> (B,c,m)   [pointnet2_utils.py:184]"""
    _chk_f(grad_out, "grad_out")
    _chk_i(idx, "idx")
    _chk_f(weight, "weight")
    B, c, n = grad_out.shape
    m = int(m)
    out = torch.empty((B, c, m), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        capi.check(capi.load().o3d_three_interpolate_grad(grad_out.data_ptr(), idx.data_ptr(), weight.data_ptr(),
                                                          B, c, n, m, out.data_ptr(), _stream()),
                   "three_interpolate_grad")
    return out


def knn(query, ref, k):
    """query (B,Q,D), ref (B,R,D) -> (B,Q,k) i32: k nearest refs, ascending, ties -> lowest index.

    Pins the tie order torch.argsort leaves open at models/head/xcorr.py:87 and
    pointnet2_utils.py:400 (knn_point)."""
    _chk_f(query, "query")
    _chk_f(ref, "ref")
    B, Q, D = query.shape
    R = ref.shape[1]
    k = int(k)
    out = torch.empty((B, Q, k), dtype=torch.int32, device=query.device)
    with torch.cuda.device(query.device):
        capi.check(capi.load().o3d_knn(query.data_ptr(), ref.data_ptr(), B, Q, R, D, k, out.data_ptr(),
                                       _stream()), "knn")
    return out
