"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/o3dsot.h declares; the Python binding refuses CPU tensors (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "o3dsot.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(o3d_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_header():
    from open3dsot_amd import build, capi
    so = build.build()
    lib = ctypes.CDLL(so)
    names = declared_symbols()
    assert len(names) >= 11
    for n in names:
        assert hasattr(lib, n), "libo3dsot_hip.so does not export %s" % n
    assert capi.version().startswith("o3dsot-hip")
    for n in capi.SIGNATURES:  # everything the binding calls is exported as well
        assert hasattr(lib, n), n


def test_code_object_is_gfx950():
    so = os.path.join(ROOT, "open3dsot_amd", "_lib", "libo3dsot_hip.so")
    blob = open(so, "rb").read()
    assert b"gfx950" in blob
    assert b"gfx942" not in blob and b"gfx90a" not in blob   # single-target build, no fat multi-arch


def test_pointnet2_ops_ext_is_a_dropin():
    import pointnet2_ops._ext as ext
    for n in ("furthest_point_sampling", "gather_points", "gather_points_grad", "three_nn", "three_interpolate",
              "three_interpolate_grad", "ball_query", "group_points", "group_points_grad"):
        assert callable(getattr(ext, n))


def test_cpu_tensors_are_refused():
    import pointnet2_ops._ext as ext
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ext.furthest_point_sampling(torch.zeros(1, 8, 3), 4)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        ext.group_points(torch.zeros(1, 2, 8), torch.zeros(1, 2, 2, dtype=torch.int32))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from open3dsot_amd import capi
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "SO_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(capi.O3DError, match="no CPU fallback"):
        capi.load()
