"""ctypes binding of include/o3dsot.h (libo3dsot_hip.so).

The library is the product's only compute path: if it is missing or fails to load this
module raises -- there is NO CPU or PyTorch fallback.  `import torch` happens first on
purpose: torch brings its own ROCm runtime (libamdhip64.so.7) and the kernels must run on
the same runtime instance that owns torch's device pointers and streams.
"""
import ctypes
import functools
import os

import torch  # noqa: F401  (must be loaded before the HIP library, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "_lib", "libo3dsot_hip.so")
if os.environ.get("O3D_LIB_VARIANT"):      # A/B builds of tools/build_variant.sh (same ABI, different -D switches)
    SO_PATH = os.path.join(_HERE, "_lib", "libo3dsot_hip.%s.so" % os.environ["O3D_LIB_VARIANT"])

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float

# name -> argtypes (every function returns int except o3d_version)
SIGNATURES = {
    "o3d_furthest_point_sampling": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "o3d_furthest_point_sampling_shfl": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "o3d_furthest_point_sampling_pair": [_vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp],
    "o3d_gather_points": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "o3d_gather_points_grad": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "o3d_gather_rows": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "o3d_gather_rows2": [_vp, _i, _vp, _i, _vp, ctypes.c_long, _i, _i, _i, _vp, _vp, _vp],
    "o3d_ball_query": [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp],
    "o3d_group_points": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "o3d_group_points_grad": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "o3d_three_nn": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "o3d_three_interpolate": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "o3d_three_interpolate_grad": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "o3d_knn": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
}

_lib = None


class O3DError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises O3DError when it is absent -- by design."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise O3DError(
            "open3dsot_amd: %s is missing. Build it with `python -m open3dsot_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % SO_PATH)
    try:
        lib = ctypes.CDLL(SO_PATH)
    except OSError as e:  # pragma: no cover
        raise O3DError("open3dsot_amd: cannot load %s: %s" % (SO_PATH, e))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.argtypes = argtypes
        fn.restype = ctypes.c_long if name.endswith("_scratch") else ctypes.c_int
    lib.o3d_version.restype = ctypes.c_char_p
    lib.o3d_version.argtypes = []
    _lib = lib
    return lib


def register(name, argtypes):
    """Declare one more entry point (used by the fused-layer modules)."""
    SIGNATURES[name] = argtypes
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_long if name.endswith("_scratch") else ctypes.c_int


_ERR = {-1: "invalid argument (shape / null pointer / unsupported size)",
        -2: "HIP launch failed"}


def check(rc, what):
    if rc != 0:
        raise O3DError("%s failed: %s (code %d)" % (what, _ERR.get(rc, "unknown"), rc))


def version():
    return load().o3d_version().decode()


def on_tensor_device(fn):
    """Run `fn` with the CUDA device of its first GPU-tensor argument current: the launch stream, the scratch
    allocations and the kernel launches of the fused operators then all belong to the device that owns the
    pointers, whichever device the caller left current (a model on cuda:1 while cuda:0 is current)."""
    @functools.wraps(fn)
    def wrapped(*args):
        for a in args:
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index == torch.cuda.current_device():
                    break
                with torch.cuda.device(a.device):
                    return fn(*args)
        return fn(*args)
    return wrapped
