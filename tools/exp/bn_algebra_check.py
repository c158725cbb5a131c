"""CPU check of HISTORY.md (round 1-2) section 9.1: the data gradient of an inner layer WITHOUT reading the layer's own output Y.

    layer:   Y = W X            X = relu(bn(Y_prev)) >= 0, (Cin, L) columns with class weights w (first hit of a ball
                                carries its padding copies)
    BN bwd:  dY = A1*dN + w*(A2*Y + A3)      (per-channel A1, A2, A3 from bn_bwd_finalize)
    today:   dX = W^T dY                      reads dN (Cout rows) and Y (Cout rows)
    algebra: dX = [W^T diag(A1) | G] [dN ; w*X] + w (x) g3,   G = W^T diag(A2) W  (Cin x Cin),  g3 = W^T A3
             reads dN (Cout rows) and X (Cin rows -- needed for the ReLU mask anyway), never Y.

Prints the distance of both formulations (evaluated in fp32) from the fp64 result, for the tracker's layer shapes.
Runs on the CPU: python tools/exp/bn_algebra_check.py
"""
import torch

torch.manual_seed(0)
for cin, cout, L in ((64, 64, 20000), (64, 128, 20000), (128, 128, 12000), (128, 256, 12000), (256, 256, 6000)):
    g = torch.Generator().manual_seed(cin + cout)
    X64 = torch.relu(torch.randn(cin, L, generator=g, dtype=torch.float64))
    W64 = torch.randn(cout, cin, generator=g, dtype=torch.float64) / cin ** 0.5
    w64 = torch.where(torch.rand(L, generator=g) < 0.1, torch.randint(2, 28, (L,), generator=g).double(), torch.ones(L, dtype=torch.float64))
    Y64 = W64 @ X64
    dN64 = torch.randn(cout, L, generator=g, dtype=torch.float64)
    # BatchNorm-backward constants as bn_bwd_finalize_kernel forms them (weighted statistics, count = sum w)
    cnt = w64.sum()
    mu = (Y64 * w64).sum(1) / cnt
    var = ((Y64 - mu[:, None]) ** 2 * w64).sum(1) / cnt
    istd = 1.0 / torch.sqrt(var + 1e-5)
    gamma = 0.5 + torch.rand(cout, generator=g, dtype=torch.float64)
    s = dN64.sum(1)
    q = (dN64 * (Y64 - mu[:, None])).sum(1)
    A1 = gamma * istd
    A2 = -A1 * istd * istd * q / cnt
    A3 = -A1 * s / cnt - A2 * mu
    truth = W64.t() @ (A1[:, None] * dN64 + w64 * (A2[:, None] * Y64 + A3[:, None]))

    f = lambda t: t.float()
    X, W, w, Y, dN, a1, a2, a3 = map(f, (X64, W64, w64, Y64, dN64, A1, A2, A3))
    today = W.t() @ (a1[:, None] * dN + w * (a2[:, None] * Y + a3[:, None]))
    G = (W.t() * a2[None, :]) @ W
    g3 = W.t() @ a3
    Acat = torch.cat([W.t() * a1[None, :], G], dim=1)              # (Cin, Cout + Cin)
    Bcat = torch.cat([dN, w * X], dim=0)                           # (Cout + Cin, L)
    alg = Acat @ Bcat + g3[:, None] * w[None, :]
    scale = truth.abs().max().item()
    e_today = (today.double() - truth).abs().max().item() / scale
    e_alg = (alg.double() - truth).abs().max().item() / scale
    by_today, by_alg = 2 * cout + 2 * cin, cout + 2 * cin          # floats per column incl. mask read and dX write
    print("Cin %3d Cout %3d: fp32 error vs fp64  today %.2e  algebra %.2e | floats/column %d -> %d (%.0f %%), "
          "FLOPs x%.2f" % (cin, cout, e_today, e_alg, by_today, by_alg, 100.0 * by_alg / by_today, (cout + cin) / cout))
