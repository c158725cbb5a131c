"""GPU-box probe: group_reduce / group_expand time on the BAT shapes, with and without LDS atomics."""
import ctypes, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from open3dsot_amd import synth, ext
here = os.path.dirname(os.path.abspath(__file__))
vp, i = ctypes.c_void_p, ctypes.c_int
libs = {k: ctypes.CDLL(os.path.join(here, "libgroup_%s.so" % k)) for k in ("atomic", "noatomic")}
for l in libs.values():
    l.o3d_group_reduce_bwd.argtypes = [vp, vp, i, i, i, i, i, vp, vp, vp, vp, vp, vp]
    l.o3d_group_expand_fwd.argtypes = [vp, i, vp, vp, vp, i, i, i, i, i, vp, vp, vp, vp, vp]
b = synth.make_batch(0, 48, 512, 1024)
xyz = torch.from_numpy(b["search_points"]).cuda()
st = torch.cuda.current_stream().cuda_stream
def timeit(f, n=10):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, N, npoint, ns, r, C0 in (("S-SA1", 1024, 512, 32, 0.3, 64), ("S-SA2", 512, 256, 32, 0.5, 128), ("S-SA3", 256, 128, 32, 0.7, 256)):
    pts = xyz[:, :N].contiguous()
    ctr = pts[:, :npoint].contiguous()
    idx = ext.ball_query(ctr, pts, r, ns)
    P = npoint * ns
    pad = float((idx[:, :, 1:] == idx[:, :, :1]).float().mean())
    dN = torch.randn(48, C0, P, device="cuda")
    S = torch.empty(48, C0, N, device="cuda"); T = torch.empty(48, C0, npoint, device="cuda")
    cnt = torch.empty(48, N, device="cuda"); R = torch.empty(48, N, 3, device="cuda")
    Z = torch.randn(48, C0, N, device="cuda"); W0 = torch.randn(C0, 3 + C0, device="cuda")
    Y0 = torch.empty(48, C0, P, device="cuda"); part = torch.empty(48 * P // 256, 2, C0, device="cuda"); GY = torch.empty(48, C0, npoint, device="cuda")
    res = []
    for k, l in libs.items():
        t = timeit(lambda: l.o3d_group_reduce_bwd(dN.data_ptr(), idx.data_ptr(), 48, C0, N, npoint, ns, S.data_ptr(), T.data_ptr(), ctr.data_ptr(), cnt.data_ptr(), R.data_ptr(), st))
        res.append("%s %.3f ms" % (k, t))
    te = timeit(lambda: libs["atomic"].o3d_group_expand_fwd(Z.data_ptr(), N, idx.data_ptr(), ctr.data_ptr(), W0.data_ptr(), 3 + C0, 48, C0, npoint, ns, Y0.data_ptr(), part.data_ptr(), None, GY.data_ptr(), st))
    gb = dN.numel() * 4 / 1e9
    print("%s: padded-slot fraction %.2f | dN %.3f GB | reduce %s | expand %.3f ms (%.2f TB/s write)" % (name, pad, gb, " | ".join(res), te, gb / te))
