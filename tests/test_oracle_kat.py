"""Hand-computable known-answer tests of the CPU oracle.

Each case is small enough to verify by hand against the upstream kernel semantics
(SURVEY.md Appendix A): ties, duplicates, all-zero clouds, empty balls, balls with more
than nsample hits, the strict `d2 < r2` boundary.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import ops

f32 = np.float32


def test_opt_n_threads():
    assert [ops.opt_n_threads(n) for n in (1, 2, 3, 5, 511, 512, 513, 1024, 100000)] == \
        [1, 2, 2, 4, 256, 512, 512, 512, 512]


def test_fps_line_by_hand():
    # points on a line at x = 1,2,4,8,16 (none near the origin): start 0 -> farthest 16 (idx 4)
    # -> then max min-dist: x=8 (d=49 vs 64 -> min 49), x=4 (9,144->9), x=2(1,196->1) => idx 3
    # -> then x=4: min(9, 16)=9 ; x=2: min(1,36)=1 => idx 2 ; then idx 1
    xyz = np.array([[[1, 0, 0], [2, 0, 0], [4, 0, 0], [8, 0, 0], [16, 0, 0]]], f32)
    assert ops.furthest_point_sampling(xyz, 5).tolist() == [[0, 4, 3, 2, 1]]


def test_fps_skips_near_origin_and_all_zero_cloud():
    # |p|^2 <= 1e-3 is never selected: (0.01,0,0) has mag 1e-4
    xyz = np.array([[[5, 0, 0], [0.01, 0, 0], [6, 0, 0], [0, 0, 0]]], f32)
    assert ops.furthest_point_sampling(xyz, 4).tolist() == [[0, 2, 0, 0]]
    # after both valid points are taken every min-dist is 0 -> tie at 0 -> lowest-rank valid thread
    zero = np.zeros((2, 16, 3), f32)
    assert (ops.furthest_point_sampling(zero, 8) == 0).all()


def near_origin_literal_case():
    """A cloud whose point 1 has |p|^2 == (float)1e-3 == 0x3A83126F EXACTLY under the canonical chain
    fmaf(z,z, fmaf(y,y, x*x)), point 3 one float below; -> (xyz, expected indices under upstream's double literal)."""
    from fractions import Fraction
    x = np.uint32(0x3cab68e6).view(f32)
    y = np.uint32(0x3cc23c60).view(f32)
    t = np.uint32(0x3A83126F).view(f32)
    xx = f32(x * x)                                          # one rounding, as __fmul_rn
    exact = Fraction(float(y)) * Fraction(float(y)) + Fraction(float(xx))     # what the fma rounds ONCE
    ulp = Fraction(2) ** (-10 - 23)                          # t is in [2^-10, 2^-9)
    assert abs(exact - Fraction(float(t))) < ulp / 2, "the constructed point does not hit 0x3A83126F"
    assert t == f32(1e-3) and float(t) > 1e-3               # above the double literal: upstream KEEPS the point
    # a point one float BELOW the threshold float is <= 0.001 in double as well: skipped under either literal
    lo = np.array([0.0316227, 0, 0], f32)                    # 0.0316227^2 = 9.99995e-4
    assert float(f32(lo[0] * lo[0])) <= 1e-3
    xyz = np.array([[[1, 0, 0], [x, y, 0], [0.5, 0, 0], lo]], f32)
    # from point 0: point 1 is at d2 = 0.959 (kept -> farthest), point 2 at 0.25, point 3 never a candidate;
    # then point 2 (min(0.25, 0.23) = 0.23 > 0 of the taken ones).  With `mag <= 1e-3f` it would be [0, 2, 0].
    return xyz, [[0, 1, 2]]


def test_fps_near_origin_literal_is_double():
    """upstream: `if (mag <= 1e-3) continue;` -- float against the DOUBLE literal (round-4 review, fidelity nit)"""
    xyz, want = near_origin_literal_case()
    assert ops.furthest_point_sampling(xyz, 3).tolist() == want
    # the same decision at a size where the product's register kernel holds several points per lane
    big = np.concatenate([xyz, np.tile(np.array([[[0.75, 0, 0]]], f32), (1, 296, 1))], 1)
    got = ops.furthest_point_sampling(big, 3)[0].tolist()
    assert got[:2] == [0, 1], got


def test_fps_tie_follows_upstream_tree_order():
    # N=4 -> block of 4 threads; points 1,2,3 are all at distance 1 from point 0 (a tie).
    # Tree: stride 2 merges (0,2),(1,3) keeping the lower slot on ties; stride 1 merges (0,1).
    # thread bests (round 1): t0: d=0 (idx0), t1:1, t2:1, t3:1 -> slots after s=2: [max(0,1)->idx2, tie(1,1)->idx1]
    # s=1: tie(1,1) keeps slot 0 -> idx 2.  So the winner is 2, NOT the lowest index 1.
    xyz = np.array([[[1, 1, 1], [2, 1, 1], [1, 2, 1], [1, 1, 2]]], f32)
    out = ops.furthest_point_sampling(xyz, 2)
    assert out.tolist() == [[0, 2]]


def test_fps_duplicates():
    # exact duplicates of the farthest point: bit-reversed thread order decides which copy wins
    base = np.array([[1, 1, 1], [9, 1, 1]], f32)
    xyz = base[[0, 1, 1, 1, 1, 1, 1, 1]][None]           # N=8, block 8
    out = ops.furthest_point_sampling(xyz, 3)[0]
    # candidates tids 1..7 tie; bit-reversal rank (3 bits): tid4 -> 001 is the lowest among 1..7
    assert out[1] == 4
    # afterwards every min-dist is 0: tie over ALL threads incl. tid 0 -> idx 0
    assert out[2] == 0


def test_ball_query_cases():
    xyz = np.array([[[0, 0, 0], [0.1, 0, 0], [0.2, 0, 0], [0.3, 0, 0], [5, 5, 5], [0.05, 0, 0]]], f32)
    centres = np.array([[[0, 0, 0], [5, 5, 5], [9, 9, 9]]], f32)
    r = 0.3
    got = ops.ball_query(centres, xyz, r, 4)
    r2 = f32(r) * f32(r)
    # centre 0: d2 = 0, .01, .04, .09(f32) , - , .0025 ; strict d2 < r2
    d2_3 = f32(0.3) * f32(0.3)
    exp0 = [0, 1, 2] + ([3] if d2_3 < r2 else []) + [5]
    exp0 = (exp0 + [exp0[0]] * 4)[:4] if len(exp0) < 4 else exp0[:4]
    assert got[0, 0].tolist() == exp0          # more than nsample hits -> first 4 ascending
    assert got[0, 1].tolist() == [4, 4, 4, 4]  # single hit pads every slot with it
    assert got[0, 2].tolist() == [0, 0, 0, 0]  # empty ball -> zeros
    # boundary: a point exactly at distance r is EXCLUDED (strict <)
    xyz2 = np.array([[[0, 0, 0], [0.5, 0, 0]]], f32)
    assert ops.ball_query(np.zeros((1, 1, 3), f32), xyz2, 0.5, 2).tolist() == [[[0, 0]]]


def test_group_gather_and_grads():
    feats = np.arange(2 * 3 * 5, dtype=f32).reshape(2, 3, 5)
    idx = np.array([[[0, 4], [2, 2]], [[1, 1], [3, 0]]], np.int32)
    g = ops.group_points(feats, idx)
    assert g.shape == (2, 3, 2, 2)
    assert g[0, 1].tolist() == [[5, 9], [7, 7]] and g[1, 2].tolist() == [[26, 26], [28, 25]]
    go = np.ones((2, 3, 2, 2), f32)
    gg = ops.group_points_grad(go, idx, 5)
    assert gg[0, 0].tolist() == [1, 0, 2, 0, 1] and gg[1, 0].tolist() == [1, 2, 0, 1, 0]
    gi = np.array([[4, 0, 0], [1, 2, 3]], np.int32)
    ga = ops.gather_points(feats, gi)
    assert ga[0, 0].tolist() == [4, 0, 0] and ga[1, 1].tolist() == [21, 22, 23]
    assert ops.gather_points_grad(np.ones((2, 3, 3), f32), gi, 5)[0, 2].tolist() == [2, 0, 0, 0, 1]


def test_three_nn_and_interpolate():
    known = np.array([[[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [1, 0, 0]]], f32)
    unknown = np.array([[[0.9, 0, 0], [0, 0, 0]]], f32)
    d2, idx = ops.three_nn(unknown, known)
    # duplicate known points 1 and 4: strict '<' keeps the earlier index first
    assert idx[0, 0].tolist() == [1, 4, 0]
    np.testing.assert_allclose(d2[0, 0], [f32(0.1) ** 2, f32(0.1) ** 2, f32(0.9) ** 2], rtol=1e-5)
    assert idx[0, 1].tolist() == [0, 1, 4]
    feats = np.array([[[10, 20, 30, 40, 50]]], f32)
    w = np.array([[[0.5, 0.25, 0.25], [1, 0, 0]]], f32)
    out = ops.three_interpolate(feats, idx, w)
    assert out[0, 0].tolist() == [0.5 * 20 + 0.25 * 50 + 0.25 * 10, 10]
    gr = ops.three_interpolate_grad(np.ones((1, 1, 2), f32), idx, w, 5)
    assert gr[0, 0].tolist() == [0.25 + 1, 0.5 + 0, 0, 0, 0.25 + 0]
    # fewer than 3 known points: remaining slots stay (1e40 -> inf in f32, index 0)
    d2b, idxb = ops.three_nn(unknown, known[:, :1])
    assert idxb[0, 0].tolist() == [0, 0, 0] and np.isinf(d2b[0, 0, 1])


def test_knn_stable_ties():
    ref = np.array([[[0, 0], [1, 0], [1, 0], [0, 1], [3, 3]]], f32)
    q = np.array([[[0, 0], [1, 0.0]]], f32)
    out = ops.knn(q, ref, 3)
    assert out[0, 0].tolist() == [0, 1, 2]   # 1,2,3 tie at d=1 -> lowest indices
    assert out[0, 1].tolist() == [1, 2, 0]   # 1 and 2 tie at 0


def test_frozen_oracle_outputs(golden_index):
    """the committed oracle outputs (tests/golden/oracle_index_ops.npz) are reproduced exactly"""
    g = golden_index
    assert np.array_equal(ops.furthest_point_sampling(g["xyz_t"], 256), g["fps_t"])
    assert np.array_equal(ops.furthest_point_sampling(g["xyz_s"], 512), g["fps_s"])
    new_s = np.take_along_axis(g["xyz_s"], g["fps_s"][:, :, None].astype(np.int64), 1)
    assert np.array_equal(ops.ball_query(new_s, g["xyz_s"], 0.3, 32), g["ball_s_r03"])
    assert np.array_equal(ops.ball_query(new_s[:, :64], new_s[:, :128], 0.3, 16), g["ball_rpn"])
    h = hashlib.sha256()
    for k in sorted(g.files):
        h.update(k.encode()); h.update(np.ascontiguousarray(g[k]).tobytes())
    here = os.path.dirname(os.path.abspath(__file__))
    assert h.hexdigest() == open(os.path.join(here, "golden", "oracle_index_ops.npz.sha256")).read().strip()
