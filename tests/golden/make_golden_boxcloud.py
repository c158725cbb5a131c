"""Generate tests/golden/ref_boxcloud.npz by running the REFERENCE'S OWN code in this container.

Run from the repo root, only where /root/reference exists:  python tests/golden/make_golden_boxcloud.py
Reference code executed (read-only, from /root/reference): datasets/data_classes.py (Box.__init__,
Box.corners :226-250) and datasets/points_utils.py (get_point_to_box_distance :127-143, regularize_pc
:24-40).  Stubbed because the packages are absent from the sandbox: nuscenes (imported, never called) and
pyquaternion -- `Quaternion` is replaced by a minimal class whose `rotation_matrix` is the standard
unit-quaternion -> matrix formula (the only attribute Box.corners reads).
"""
import importlib.util
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"


class Quaternion:
    def __init__(self, axis=None, radians=None, elements=None):
        if elements is not None:
            q = np.asarray(elements, dtype=np.float64)
        else:
            ax = np.asarray(axis, dtype=np.float64)
            ax = ax / np.linalg.norm(ax)
            q = np.concatenate([[np.cos(radians / 2)], np.sin(radians / 2) * ax])
        self.elements = q / np.linalg.norm(q)

    @property
    def rotation_matrix(self):
        w, x, y, z = self.elements
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


stub("nuscenes"); stub("nuscenes.utils"); stub("nuscenes.utils.geometry_utils")
stub("pyquaternion", Quaternion=Quaternion)
stub("datasets")
dc = load("datasets.data_classes", "datasets/data_classes.py")
pu = load("datasets.points_utils", "datasets/points_utils.py")

rng = np.random.default_rng(20260926)
out = {}
cases = [(512, 1.0), (1024, 1.0), (37, 1.25), (3, 1.0)]
for i, (n, factor) in enumerate(cases):
    center = rng.normal(0, 3.0, 3)
    wlh = rng.uniform(0.5, 4.5, 3)
    axis = rng.normal(0, 1, 3) if i == 2 else np.array([0.0, 0.0, 1.0])       # KITTI boxes: yaw only; one general case
    q = Quaternion(axis=axis, radians=float(rng.uniform(-np.pi, np.pi)))
    box = dc.Box(center, wlh, q)
    pts = (center + rng.normal(0, 2.0, (n, 3))).astype(np.float32)
    out["points_%d" % i] = pts
    out["center_%d" % i] = center
    out["wlh_%d" % i] = wlh
    out["rot_%d" % i] = q.rotation_matrix
    out["factor_%d" % i] = np.float64(factor)
    out["corners_%d" % i] = box.corners(wlh_factor=factor)
    out["bc_%d" % i] = pu.get_point_to_box_distance(pts, box, wlh_factor=factor)
# regularize_pc with the inference seed (models/bat.py:42-45): more, fewer, equal, degenerate
for j, (n, size) in enumerate([(700, 512), (300, 512), (512, 512), (2, 512), (1500, 1024)]):
    pts = rng.normal(0, 1, (n, 3)).astype(np.float32)
    res, idx = pu.regularize_pc(pts, size, seed=1)
    out["reg_in_%d" % j] = pts
    out["reg_out_%d" % j] = res
    out["reg_idx_%d" % j] = np.asarray(idx if idx is not None else [-1])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_boxcloud.npz"), **out)
print("wrote ref_boxcloud.npz:", len(out), "arrays")
