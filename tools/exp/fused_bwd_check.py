#!/usr/bin/env python3
"""tools/exp/fused_bwd_check.py -- numerical check and timing of the EXPERIMENTAL fused data + weight gradient kernel
(tools/exp/fused_bwd.hip) against an fp64 torch evaluation and against the production pair
(o3d_mlp_conv_dgrad_c + o3d_mlp_conv_wgrad2_c).  Build first in the CPU container (the .so travels with gpurun):
    python tools/exp/fused_bwd_check.py --build
then on the GPU box:
    python tools/exp/fused_bwd_check.py [--big]
"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
V4 = False          # (round 2's never-run v3 / v4 variants were deleted in round 3)
SRC = "fused_bwd.hip"
SO = os.path.join(HERE, "_" + SRC.replace(".hip", ".so"))
sys.path.insert(0, ROOT)


def build():
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
           "-I", os.path.join(ROOT, "open3dsot_amd", "csrc"), os.path.join(HERE, SRC), "-o", SO]
    subprocess.check_call(cmd)
    print("built", SO)


def main(big):
    import torch
    from open3dsot_amd import capi, fused  # noqa: F401 (registers the production entry points)
    lib = capi.load()
    exp = ctypes.CDLL(SO)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    exp.o3d_exp_bwd_fused.argtypes = [vp] * 10 + [i32, i32, i64, vp, vp, i64, i32, vp, vp, vp, vp]
    if V4:
        exp.o3d_exp_bwd_fused_slices_v4.argtypes = [i32, i32, i64]
    else:
        exp.o3d_exp_bwd_fused_slices.argtypes = [i32, i64]
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    st = torch.cuda.current_stream().cuda_stream
    cases = [(64, 64, 256 * 40, 256 * 13, 256 * 17), (64, 128, 256 * 40, 256 * 13, 256 * 17), (64, 64, 256 * 8, 256 * 3, 256 * 1)]
    if V4:
        cases += [(128, 128, 256 * 40, 256 * 13, 256 * 17)]
    if big:
        cases += [(64, 64, 256 * 4608, 256 * 500, 256 * 1000), (64, 128, 256 * 4608, 256 * 500, 256 * 1000)]
        if V4:      # SA2 layer 1 at the benchmarked batch: 201 472 live columns of 589 824
            cases += [(128, 128, 256 * 2304, 256 * 262, 256 * 525)]
    for Cin, Cout, ldp, live0, live1 in cases:
        start1 = (ldp // 512) * 256
        rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
        dN, Y, X = rnd(Cout, ldp), rnd(Cout, ldp), rnd(Cin, ldp)
        w = torch.rand(ldp, device=dev, generator=g) * 3
        A = [rnd(2, Cout) * 0.5 for _ in range(3)]
        sc, sh, mu = rnd(2, Cin) * 0.5 + 1.0, rnd(2, Cin) * 0.3, rnd(2, Cin) * 0.3
        W = rnd(Cout, Cin) * 0.1
        Wt = W.t().contiguous()
        meta = torch.tensor([live0, 0, 0, 0, live1, 0, 0, 0], dtype=torch.int32, device=dev)
        segs = [(0, live0), (start1, live1)]
        for s_, (b0, n) in enumerate(segs):       # no activation within rounding of the ReLU threshold: the mask is then the
            sl = slice(b0, b0 + n)                # same in fp32 (fma or not) and fp64, and every difference below is arithmetic
            pre = sc[s_][:, None] * X[:, sl] + sh[s_][:, None]
            near = pre.abs() < 1e-3
            X[:, sl] = torch.where(near, (torch.where(pre >= 0, 1e-2, -1e-2) - sh[s_][:, None]) / sc[s_][:, None], X[:, sl])
        # ---- fp64 truth
        dW64 = torch.zeros(Cout, Cin, dtype=torch.float64, device=dev)
        dX64 = torch.zeros(Cin, ldp, dtype=torch.float64, device=dev)
        st64 = torch.zeros(2, 2, Cin, dtype=torch.float64, device=dev)
        for s_, (b0, n) in enumerate(segs):
            sl = slice(b0, b0 + n)
            dY = A[0][s_].double()[:, None] * dN[:, sl].double() + w[sl].double()[None] * (
                A[1][s_].double()[:, None] * Y[:, sl].double() + A[2][s_].double()[:, None])
            pre = torch.addcmul(sh[s_][:, None], X[:, sl], sc[s_][:, None])          # fp32 fma like the kernels' mask
            Xt = torch.clamp(sc[s_].double()[:, None] * X[:, sl].double() + sh[s_].double()[:, None], min=0) * (pre > 0)
            dW64 += dY @ Xt.t()
            gq = (Wt.double() @ dY) * (pre > 0)
            dX64[:, sl] = gq
            st64[s_, 0] = gq.sum(1)
            st64[s_, 1] = (gq * (X[:, sl].double() - mu[s_].double()[:, None])).sum(1)
        # ---- experimental kernel
        nsl = exp.o3d_exp_bwd_fused_slices_v4(Cin, Cout, ldp) if V4 else exp.o3d_exp_bwd_fused_slices(Cout, ldp)
        WK = 4 // ((Cout // 64) * (Cin // 64))
        NPT = (2 if Cin == 64 else 1) if V4 else (2 if (Cout == 64 or V3) else 1)
        part_w = torch.full((nsl * WK, Cout, Cin), float("nan"), device=dev)
        part_s = torch.full((2, nsl * NPT, 2, Cin), float("nan"), device=dev)
        dX = torch.zeros(Cin, ldp, device=dev)
        args = (dN.data_ptr(), Y.data_ptr(), A[0].data_ptr(), A[1].data_ptr(), A[2].data_ptr(), X.data_ptr(), sc.data_ptr(),
                sh.data_ptr(), mu.data_ptr(), Wt.data_ptr(), Cin, Cout, ldp, w.data_ptr(), meta.data_ptr(), start1, nsl,
                part_w.data_ptr(), part_s.data_ptr(), dX.data_ptr(), st)
        rc = exp.o3d_exp_bwd_fused(*args)
        torch.cuda.synchronize()
        rel = lambda a, b: float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))
        live = torch.zeros(ldp, dtype=torch.bool, device=dev)
        for b0, n in segs:
            live[b0:b0 + n] = True
        print("Cin %3d Cout %3d ldp %8d live %d+%d nslices %d rc %d" % (Cin, Cout, ldp, live0, live1, nsl, rc))
        print("   fused   : dW %.2e   dX %.2e   sum g %.2e   sum g(y-mean) %.2e" % (
            rel(part_w.sum(0), dW64), rel(dX[:, live], dX64[:, live]), rel(part_s.sum(1)[:, 0], st64[:, 0]),
            rel(part_s.sum(1)[:, 1], st64[:, 1])))
        # ---- production pair
        tile = lib.o3d_direct_tile(ldp, Cin, 1)
        scr = torch.empty(lib.o3d_mlp_conv_wgrad2_scratch(1, Cin, Cout, ldp), device=dev)
        dWp = torch.empty(Cout, Cin, device=dev)
        dXp = torch.zeros(Cin, ldp, device=dev)
        partp = torch.zeros(ldp // tile, 2, Cin, device=dev)

        def prod():
            capi.check(lib.o3d_mlp_conv_wgrad2_c(dN.data_ptr(), Y.data_ptr(), A[0].data_ptr(), A[1].data_ptr(), A[2].data_ptr(),
                                                 X.data_ptr(), sc.data_ptr(), sh.data_ptr(), Cin, Cout, ldp, w.data_ptr(),
                                                 meta.data_ptr(), start1, scr.data_ptr(), dWp.data_ptr(), st), "wgrad2_c")
            capi.check(lib.o3d_mlp_conv_dgrad_c(dN.data_ptr(), Y.data_ptr(), A[0].data_ptr(), A[1].data_ptr(), A[2].data_ptr(),
                                                Wt.data_ptr(), Cin, Cout, ldp, w.data_ptr(), meta.data_ptr(), start1, tile,
                                                X.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), dXp.data_ptr(),
                                                partp.data_ptr(), st), "dgrad_c")
        prod()
        torch.cuda.synchronize()
        print("   product : dW %.2e   dX %.2e" % (rel(dWp, dW64), rel(dXp[:, live], dX64[:, live])))

        def timeit(fn, n=20):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n
        for ns in sorted({nsl, min(512, nsl * 2)}):
            pw = torch.empty((ns * WK, Cout, Cin), device=dev)
            ps = torch.empty((2, ns * NPT, 2, Cin), device=dev)
            a2 = args[:16] + (ns, pw.data_ptr(), ps.data_ptr(), dX.data_ptr(), st)
            print("   time    : fused (%d workgroups) %.4f ms" % (ns, timeit(lambda: exp.o3d_exp_bwd_fused(*a2))))
        print("   time    : production wgrad2_c + dgrad_c %.4f ms" % timeit(prod))


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    else:
        main("--big" in sys.argv)
