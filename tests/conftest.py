import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "ref_python_layers.npz"))


@pytest.fixture(scope="session")
def golden_index():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "oracle_index_ops.npz"))


@pytest.fixture()
def cpu_ext(monkeypatch):
    """Run the host-side mirror on the CPU by swapping the HIP operator set for the oracle
    shim (TEST ONLY -- the product has no CPU path) and disabling the fused kernels."""
    from oracle import ext_shim
    import open3dsot_amd.ext as ext
    from open3dsot_amd import sa_modules
    for name in ("furthest_point_sampling", "gather_points", "gather_points_grad", "gather_rows", "three_nn",
                 "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
                 "group_points_grad"):
        monkeypatch.setattr(ext, name, getattr(ext_shim, name))
    import torch
    from oracle import ops as oops
    monkeypatch.setattr(ext, "knn", lambda q, r, k: torch.from_numpy(
        oops.knn(q.detach().numpy(), r.detach().numpy(), k)))
    was = sa_modules.fused_enabled()
    sa_modules.set_fused(False)
    yield
    sa_modules.set_fused(was)
