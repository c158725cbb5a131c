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


def header_prototypes():
    """name -> list of parameter kinds ('p' pointer, 'i' int, 'l' long, 'f' float, 'd' double) from include/o3dsot.h"""
    src = open(os.path.join(ROOT, "include", "o3dsot.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|long|const char\s*\*)\s+(o3d_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        params = [p.strip() for p in m.group(2).split(",") if p.strip() and p.strip() != "void"]
        kinds = []
        for p in params:
            if "*" in p:
                kinds.append("p")
            else:
                t = p.split()
                kinds.append({"int": "i", "long": "l", "float": "f", "double": "d"}[t[-2] if len(t) > 1 else t[0]])
        protos[m.group(1)] = kinds
    return protos


def test_ctypes_signatures_match_the_header():
    """every argtypes list registered by the Python binding has the arity and the argument kinds of its
    prototype in include/o3dsot.h (a silent ctypes mismatch corrupts arguments instead of failing)"""
    from open3dsot_amd import capi, fused, fused_loss, fused_pointwise, points_utils  # noqa: F401  (they register)
    protos = header_prototypes()
    kind = {ctypes.c_void_p: "p", ctypes.c_int: "i", ctypes.c_long: "l", ctypes.c_float: "f", ctypes.c_double: "d"}
    assert len(capi.SIGNATURES) >= 40
    for name, argtypes in capi.SIGNATURES.items():
        assert name in protos, "%s is bound but not declared in include/o3dsot.h" % name
        got = [kind[a] for a in argtypes]
        assert got == protos[name], (name, got, protos[name])
