"""Property tests of the oracle -- size-independent invariants
that the GPU tests re-use at full BASELINE sizes."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st

from oracle import ops

f32 = np.float32


def cloud(seed, B, N, dup=False, grid=False):
    rng = np.random.default_rng(seed)
    p = rng.normal(0, 1.2, (B, N, 3)).astype(f32)
    if dup:
        src = rng.integers(0, min(N, max(2, N // 3)), (B, N))
        p = np.take_along_axis(p, src[:, :, None], 1)
    if grid:
        p = (np.round(p * 2) / 2).astype(f32)
    return p


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 10**6), st.integers(1, 3), st.integers(2, 300), st.booleans(), st.booleans())
def test_fps_properties(seed, B, N, dup, grid):
    xyz = cloud(seed, B, N, dup, grid)
    m = max(1, N // 2)
    idx = ops.furthest_point_sampling(xyz, m)
    assert idx.shape == (B, m) and idx.dtype == np.int32
    assert (idx[:, 0] == 0).all() and (idx >= 0).all() and (idx < N).all()
    for b in range(B):
        # greedy max-min property: the j-th pick maximises the min distance to earlier picks
        # (checked in float64 with slack for fp32 rounding); near-origin points are excluded
        p = xyz[b].astype(np.float64)
        ok = (xyz[b].astype(f32) ** 2).sum(1) > 1e-3
        md = np.full(N, np.inf)
        for j in range(1, m):
            md = np.minimum(md, ((p - p[idx[b, j - 1]]) ** 2).sum(1))
            cand = np.where(ok, md, -1.0)
            best = cand.max()
            if best <= 0:
                break
            assert md[idx[b, j]] >= best * (1 - 1e-5) - 1e-6


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 10**6), st.integers(1, 3), st.integers(1, 200), st.integers(1, 40),
       st.sampled_from([0.2, 0.5, 1.0, 3.0]), st.sampled_from([1, 4, 16, 70]))
def test_ball_query_properties(seed, B, N, npoint, radius, nsample):
    xyz = cloud(seed, B, N, dup=seed % 2 == 0)
    new_xyz = cloud(seed + 1, B, npoint)
    idx = ops.ball_query(new_xyz, xyz, radius, nsample)
    r2 = f32(radius) * f32(radius)
    for b in range(B):
        d = xyz[b][None, :, :] - new_xyz[b][:, None, :]
        # same fma chain in float64 emulation is overkill here: use a tolerance band
        d2 = (d.astype(np.float64) ** 2).sum(2)
        for j in range(npoint):
            row = idx[b, j]
            inside = np.nonzero(d2[j] < r2 * (1 - 1e-5))[0]
            maybe = np.nonzero(d2[j] < r2 * (1 + 1e-5))[0]
            if len(maybe) == 0:
                assert (row == 0).all()
                continue
            uniq = row[:min(nsample, len(inside))] if len(inside) == len(maybe) else None
            if uniq is not None:
                assert uniq.tolist() == inside[:nsample].tolist()          # first hits, ascending
                assert (row[len(uniq):] == (inside[0] if len(inside) else 0)).all()  # padded with first hit
            assert np.isin(row, np.concatenate([maybe, [0]])).all()


@settings(max_examples=20, deadline=None)
@given(st.integers(0, 10**6), st.integers(1, 3), st.integers(1, 9), st.integers(1, 50), st.integers(1, 12), st.integers(1, 6))
def test_group_matches_torch_gather_and_autograd(seed, B, C, N, npoint, nsample):
    rng = np.random.default_rng(seed)
    feats = rng.normal(size=(B, C, N)).astype(f32)
    idx = rng.integers(0, N, (B, npoint, nsample)).astype(np.int32)
    got = ops.group_points(feats, idx)
    t = torch.from_numpy(feats).requires_grad_(True)
    flat = torch.from_numpy(idx).long().reshape(B, 1, -1).expand(B, C, -1)
    ref = t.gather(2, flat).reshape(B, C, npoint, nsample)
    assert np.array_equal(got, ref.detach().numpy())
    go = rng.normal(size=got.shape).astype(f32)
    ref.backward(torch.from_numpy(go))
    np.testing.assert_allclose(ops.group_points_grad(go, idx, N), t.grad.numpy(), rtol=1e-5, atol=1e-5)
    # gather == group with nsample 1
    assert np.array_equal(ops.gather_points(feats, idx[:, :, 0]), got[:, :, :, 0])


@settings(max_examples=20, deadline=None)
@given(st.integers(0, 10**6), st.integers(1, 2), st.integers(1, 40), st.integers(3, 60), st.integers(1, 5))
def test_three_nn_interpolate(seed, B, n, m, c):
    unknown, known = cloud(seed, B, n), cloud(seed + 7, B, m)
    d2, idx = ops.three_nn(unknown, known)
    full = ((unknown[:, :, None, :].astype(np.float64) - known[:, None, :, :]) ** 2).sum(3)
    srt = np.sort(full, axis=2)[:, :, :3]
    np.testing.assert_allclose(d2, srt, rtol=1e-4, atol=1e-6)
    assert (np.diff(d2, axis=2) >= 0).all()
    rng = np.random.default_rng(seed)
    feats = rng.normal(size=(B, c, m)).astype(f32)
    w = rng.uniform(size=(B, n, 3)).astype(f32)
    out = ops.three_interpolate(feats, idx, w)
    ref = sum(np.take_along_axis(feats, np.broadcast_to(idx[:, None, :, t].astype(np.int64), (B, c, n)), 2)
              * w[:, None, :, t] for t in range(3))
    np.testing.assert_allclose(out, ref, rtol=1e-5, atol=1e-5)
    # linearity of the adjoint: <interp(f), g> == <f, interp_grad(g)>
    g = rng.normal(size=out.shape).astype(f32)
    lhs = (out.astype(np.float64) * g).sum()
    rhs = (feats.astype(np.float64) * ops.three_interpolate_grad(g, idx, w, m)).sum()
    assert abs(lhs - rhs) <= 1e-3 * (1 + abs(lhs))


@settings(max_examples=20, deadline=None)
@given(st.integers(0, 10**6), st.integers(1, 2), st.integers(1, 30), st.integers(1, 40), st.integers(1, 9))
def test_knn_matches_stable_argsort(seed, B, Q, R, D):
    rng = np.random.default_rng(seed)
    q = (np.round(rng.normal(size=(B, Q, D)) * 2) / 2).astype(f32)   # grid -> many exact ties
    r = (np.round(rng.normal(size=(B, R, D)) * 2) / 2).astype(f32)
    k = min(R, 4)
    got = ops.knn(q, r, k)
    d = ((q[:, :, None, :].astype(np.float64) - r[:, None, :, :]) ** 2).sum(3)  # exact on a 0.5-grid
    ref = np.argsort(d, axis=2, kind="stable")[:, :, :k]
    assert np.array_equal(got, ref)
