"""GPU diagnostics for round 2 (not a test): where do the flat-chain / P2B-xcorr errors sit?"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_heads_gpu as T
from open3dsot_amd import fused_heads, nn_blocks


def l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def chain_case(name, B, N, train=True):
    src_C, widths, residual = T.CASES[name]
    seq = T.build_seq(widths, sum(src_C), 3).train(train)
    ref = copy.deepcopy(seq).double()
    g = torch.Generator(device="cuda").manual_seed(5)
    parts = []
    for C in src_C:
        t = torch.randn(B, N, C, device="cuda", generator=g).transpose(1, 2) if C == 3 else torch.randn(B, C, N, device="cuda", generator=g)
        parts.append(t.requires_grad_(True))
    parts64 = [t.detach().double().requires_grad_(True) for t in parts]
    out = nn_blocks.seq_apply(seq, parts, residual)
    x64 = torch.cat(parts64, dim=1)
    ref_out = ref(x64) + (x64 if residual else 0)
    ct = torch.randn(out.shape, device="cuda", generator=g)
    (out * ct).sum().backward()
    (ref_out * ct.double()).sum().backward()
    print("== chain %s B=%d N=%d: fwd l2 %.2e" % (name, B, N, l2(out, ref_out)))
    for i, (a, b) in enumerate(zip(parts, parts64)):
        d = (a.grad.double() - b.grad).abs()
        per_row = [(l2(a.grad[:, r], b.grad[:, r])) for r in range(min(a.shape[1], 4))]
        flat = d.flatten()
        top = torch.topk(flat, 5)
        idx = [tuple(int(v) for v in torch.unravel_index(k, d.shape)) for k in top.indices]
        print("  src%d l2 %.2e rows %s | top |err| %s at %s (ref rms %.3f) | frac>1e-3*rms %.4f" % (
            i, l2(a.grad, b.grad), ["%.1e" % v for v in per_row], ["%.2e" % float(v) for v in top.values], idx,
            float(b.grad.pow(2).mean().sqrt()), float((d > 1e-3 * b.grad.pow(2).mean().sqrt()).double().mean())))
        # error per column (b, n): is it a few columns?
        col = d.pow(2).sum(1).sqrt()          # (B,N)
        print("      worst columns:", [tuple(int(v) for v in torch.unravel_index(k, col.shape)) for k in torch.topk(col.flatten(), 6).indices],
              "col err max/median %.2e / %.2e" % (float(col.max()), float(col.median())))
    for (n1, p), (_, q) in zip(seq.named_parameters(), ref.named_parameters()):
        print("  %-28s l2 %.2e" % (n1, l2(p.grad, q.grad)))


for name, B, N in (("mlp_bc", 48, 128), ("mlp_bc", 4, 64), ("cla", 48, 128), ("vote", 48, 128)):
    chain_case(name, B, N)

# ---- P2B xcorr forward error: per-layer statistics
from open3dsot_amd import fused_xcorr, sa_modules, xcorr
torch.manual_seed(2)
mod = xcorr.P2B_XCorr(256, 256, 256).cuda().train()
ref = copy.deepcopy(mod).double()
gg = torch.Generator(device="cuda").manual_seed(6)
B, M, N = 4, 64, 128
t_feat = torch.randn(B, 256, M, device="cuda", generator=gg)
s_feat = torch.randn(B, 256, N, device="cuda", generator=gg)
t_xyz = torch.randn(B, M, 3, device="cuda", generator=gg)
f = fused_xcorr.p2b_xcorr_mlp_pool(mod.mlp, t_feat, s_feat, t_xyz)
sa_modules.set_fused(False)
# fp64 reference of the mlp+pool part only
tf, sf, tx = t_feat.double(), s_feat.double(), t_xyz.double()
sim = torch.nn.functional.cosine_similarity(tf.unsqueeze(-1).expand(B, 256, M, N), sf.unsqueeze(2).expand(B, 256, M, N), dim=1)
x = torch.cat((sim.unsqueeze(1), tx.transpose(1, 2).unsqueeze(-1).expand(B, 3, M, N), tf.unsqueeze(-1).expand(B, 256, M, N)), 1)
y = ref.mlp(x).max(dim=2)[0]
sa_modules.set_fused(True)
print("== p2b xcorr mlp+pool fwd l2 %.2e max-rel %.2e" % (l2(f, y), float((f.double() - y).abs().max() / y.abs().max())))
for (n1, b1), (_, b2) in zip(mod.mlp.named_buffers(), ref.mlp.named_buffers()):
    if b1.dtype.is_floating_point:
        print("  %-32s l2 %.2e" % (n1, l2(b1, b2)))
