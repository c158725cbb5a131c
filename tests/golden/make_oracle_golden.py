"""Freeze oracle outputs on seeded inputs -> tests/golden/oracle_index_ops.npz.

The index operators have no upstream golden vectors (PARITY UNPINNED, see
oracle/pointnet2_oracle.c); this file pins the ORACLE itself against regressions and gives
the GPU tests committed vectors that do not need the oracle to be re-run.  Inputs come from
open3dsot_amd.synth (seed = 1234 + index), which includes duplicate-heavy and all-zero
clouds.  Run from the repo root:  python tests/golden/make_oracle_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ops  # noqa: E402
from open3dsot_amd import synth  # noqa: E402


def main():
    out = {}
    b = synth.make_batch(0, 12, 512, 1024)
    xyz_t, xyz_s = b["template_points"], b["search_points"]
    out["xyz_t"], out["xyz_s"] = xyz_t, xyz_s
    out["fps_t"] = ops.furthest_point_sampling(xyz_t, 256)
    out["fps_s"] = ops.furthest_point_sampling(xyz_s, 512)
    new_s = np.take_along_axis(xyz_s, out["fps_s"][:, :, None].astype(np.int64), 1)
    out["ball_s_r03"] = ops.ball_query(new_s, xyz_s, 0.3, 32)
    out["ball_s_r05"] = ops.ball_query(new_s[:, :256], new_s, 0.5, 32)
    out["ball_s_r07"] = ops.ball_query(new_s[:, :128], new_s[:, :256], 0.7, 32)
    out["ball_rpn"] = ops.ball_query(new_s[:, :64], new_s[:, :128], 0.3, 16)
    bc_s, bc_t = b["points2cc_dist_s"][:, :128], b["points2cc_dist_t"][:, :64]
    out["knn_k4"] = ops.knn(bc_s, bc_t, 4)
    d2, i3 = ops.three_nn(xyz_s[:, :200], new_s[:, :77])
    out["three_nn_d2"], out["three_nn_idx"] = d2, i3
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_index_ops.npz")
    np.savez_compressed(path, **out)
    h = hashlib.sha256()
    for k in sorted(out):
        h.update(k.encode()); h.update(np.ascontiguousarray(out[k]).tobytes())
    with open(path + ".sha256", "w") as f:
        f.write(h.hexdigest() + "\n")
    print("wrote", path, h.hexdigest()[:16], "%.2f MB" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
