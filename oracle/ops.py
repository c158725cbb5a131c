"""numpy front-end of oracle/pointnet2_oracle.c (TEST INFRASTRUCTURE, see that file's header).

Each function mirrors one `pointnet2_ops._ext` entry point as the reference calls it
(pointnet2/utils/pointnet2_utils.py:56,92,98,125,162,184,217,237,268).  PARITY UNPINNED at
this boundary -- see the C file header.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libo3d_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "pointnet2_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(_SO), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared",
                               "-o", _SO, src, "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError("oracle %s failed rc=%d" % (name, rc))


def opt_n_threads(n):
    return lib().o3d_oracle_opt_n_threads(int(n))


def furthest_point_sampling(xyz, npoint):
    xyz, p = _f(xyz)
    B, N, _ = xyz.shape
    out = np.zeros((B, npoint), np.int32)
    _chk(lib().o3d_oracle_furthest_point_sampling(p, B, N, int(npoint), out.ctypes.data_as(ctypes.c_void_p)), "fps")
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    new_xyz, pn = _f(new_xyz)
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    npoint = new_xyz.shape[1]
    out = np.zeros((B, npoint, nsample), np.int32)
    _chk(lib().o3d_oracle_ball_query(pn, px, B, N, npoint, ctypes.c_float(radius), int(nsample),
                                     out.ctypes.data_as(ctypes.c_void_p)), "ball_query")
    return out


def group_points(feats, idx):
    feats, pf = _f(feats)
    idx, pi = _i(idx)
    B, C, N = feats.shape
    _, npoint, nsample = idx.shape
    out = np.empty((B, C, npoint, nsample), np.float32)
    _chk(lib().o3d_oracle_group_points(pf, pi, B, C, N, npoint, nsample, out.ctypes.data_as(ctypes.c_void_p)), "group")
    return out


def group_points_grad(grad_out, idx, N):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, npoint, nsample = grad_out.shape
    out = np.empty((B, C, N), np.float32)
    _chk(lib().o3d_oracle_group_points_grad(pg, pi, B, C, int(N), npoint, nsample,
                                            out.ctypes.data_as(ctypes.c_void_p)), "group_grad")
    return out


def gather_points(feats, idx):
    feats, pf = _f(feats)
    idx, pi = _i(idx)
    B, C, N = feats.shape
    npoint = idx.shape[1]
    out = np.empty((B, C, npoint), np.float32)
    _chk(lib().o3d_oracle_gather_points(pf, pi, B, C, N, npoint, out.ctypes.data_as(ctypes.c_void_p)), "gather")
    return out


def gather_points_grad(grad_out, idx, N):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, npoint = grad_out.shape
    out = np.empty((B, C, N), np.float32)
    _chk(lib().o3d_oracle_gather_points_grad(pg, pi, B, C, int(N), npoint,
                                             out.ctypes.data_as(ctypes.c_void_p)), "gather_grad")
    return out


def three_nn(unknown, known):
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.empty((B, n, 3), np.float32)
    idx = np.empty((B, n, 3), np.int32)
    _chk(lib().o3d_oracle_three_nn(pu, pk, B, n, m, d2.ctypes.data_as(ctypes.c_void_p),
                                   idx.ctypes.data_as(ctypes.c_void_p)), "three_nn")
    return d2, idx


def three_interpolate(feats, idx, weight):
    feats, pf = _f(feats)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, c, m = feats.shape
    n = idx.shape[1]
    out = np.empty((B, c, n), np.float32)
    _chk(lib().o3d_oracle_three_interpolate(pf, pi, pw, B, c, m, n, out.ctypes.data_as(ctypes.c_void_p)), "interp")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, c, n = grad_out.shape
    out = np.empty((B, c, m), np.float32)
    _chk(lib().o3d_oracle_three_interpolate_grad(pg, pi, pw, B, c, n, int(m),
                                                 out.ctypes.data_as(ctypes.c_void_p)), "interp_grad")
    return out


def knn(query, ref, k):
    """k nearest rows of ref (B,R,D) for each query (B,Q,D): ascending, ties -> lowest index."""
    query, pq = _f(query)
    ref, pr = _f(ref)
    B, Q, D = query.shape
    R = ref.shape[1]
    out = np.empty((B, Q, k), np.int32)
    _chk(lib().o3d_oracle_knn(pq, pr, B, Q, R, D, int(k), out.ctypes.data_as(ctypes.c_void_p)), "knn")
    return out
