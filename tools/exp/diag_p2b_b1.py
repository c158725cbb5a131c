#!/usr/bin/env python
"""GPU diagnostic (round 6): P2B training forward at batch 1, stage by stage against the CPU oracle in fp32 and fp64 -- where
does the 5e-3 difference the batch-1 parity test found come from?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F

from open3dsot_amd import fused_heads, nn_blocks as pt_utils, sa_modules, synth
from oracle import torch_ref as R
import test_model_gpu as T

name = "P2B"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
first = int(sys.argv[2]) if len(sys.argv) > 2 else 171
host = synth.make_batch(first, B)
model = T.make_model(name, 6)
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
dev = torch.device("cuda", 0)
batch = synth.to_torch(host, dev)


def oracle(dtype):
    sdx = {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    b = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in synth.to_torch(host).items()}
    st = R.State(sdx, True)
    o = {}
    with torch.no_grad():
        template, search = b["template_points"], b["search_points"]
        M, N = template.shape[1], search.shape[1]
        t_xyz, t_feat, _ = R.backbone(st, template, [M // 2, M // 4, M // 8], False)
        s_xyz, s_feat, s_idx = R.backbone(st, search, [N // 2, N // 4, N // 8], False)
        o["backbone.t_feat"], o["backbone.s_feat"] = t_feat, s_feat
        t_feat = F.conv1d(t_feat, sdx["conv_final.weight"], sdx["conv_final.bias"])
        s_feat = F.conv1d(s_feat, sdx["conv_final.weight"], sdx["conv_final.bias"])
        o["conv_final.t"], o["conv_final.s"] = t_feat, s_feat
        Bq, f, n1 = t_feat.shape
        n2 = s_feat.shape[2]
        sim = F.cosine_similarity(t_feat.unsqueeze(-1).expand(Bq, f, n1, n2), s_feat.unsqueeze(2).expand(Bq, f, n1, n2), dim=1)
        o["sim"] = sim
        x = torch.cat((sim.unsqueeze(1), t_xyz.transpose(1, 2).unsqueeze(-1).expand(Bq, 3, n1, n2),
                       t_feat.unsqueeze(-1).expand(Bq, f, n1, n2)), dim=1)
        x = st.shared_mlp("xcorr.mlp", x)
        o["xcorr.mlp_pool"] = x.max(dim=2)[0]
        fusion = st.seq("xcorr.fea_layer", o["xcorr.mlp_pool"], 2)
        o["xcorr.out"] = fusion
        boxes, cla, vote_xyz, centers = R.rpn(st, s_xyz, fusion, 64)
        o["rpn.cla"], o["rpn.vote_xyz"], o["rpn.boxes"] = cla, vote_xyz, boxes
    return o


def gpu(fused):
    model.load_state_dict(sd)
    sa_modules.set_fused(fused)
    o = {}
    with torch.no_grad(), fused_heads.prep_scope(dev):
        template, search = batch["template_points"], batch["search_points"]
        M, N = template.shape[1], search.shape[1]
        (t_xyz, t_feat, _), (s_xyz, s_feat, sidx) = model.backbone.forward_pair(
            template, [M // 2, M // 4, M // 8], search, [N // 2, N // 4, N // 8], model._given_sampling(batch))
        o["backbone.t_feat"], o["backbone.s_feat"] = t_feat, s_feat
        t_feat, s_feat = pt_utils.pointwise_conv1d_pair(model.conv_final, t_feat, s_feat)
        o["conv_final.t"], o["conv_final.s"] = t_feat, s_feat
        tn = t_feat.norm(dim=1).clamp_min(1e-8)
        sn = s_feat.norm(dim=1).clamp_min(1e-8)
        o["sim"] = torch.bmm(t_feat.transpose(1, 2), s_feat) / (tn.unsqueeze(2) * sn.unsqueeze(1))
        xc = model.xcorr
        if fused:
            from open3dsot_amd import fused_xcorr
            pooled = fused_xcorr.p2b_xcorr_mlp_pool(xc.mlp, t_feat, s_feat, t_xyz)
        else:
            Bq, f, n1 = t_feat.shape
            n2 = s_feat.shape[2]
            x = torch.cat((o["sim"].unsqueeze(1), t_xyz.transpose(1, 2).unsqueeze(-1).expand(Bq, 3, n1, n2),
                           t_feat.unsqueeze(-1).expand(Bq, f, n1, n2)), dim=1)
            pooled = F.max_pool2d(xc.mlp(x), kernel_size=[n1, 1]).squeeze(2)
        o["xcorr.mlp_pool"] = pooled
        fusion = xc.fea_layer(pooled)
        o["xcorr.out"] = fusion
        boxes, cla, vote_xyz, centers = model.rpn(s_xyz, fusion)
        o["rpn.cla"], o["rpn.vote_xyz"], o["rpn.boxes"] = cla, vote_xyz, boxes
    sa_modules.set_fused(True)
    return {k: v.detach().float().cpu() for k, v in o.items()}


o64, o32 = oracle(torch.float64), oracle(torch.float32)
gf, gc = gpu(True), gpu(False)
print("stage                    fused-vs-fp64  composed-vs-fp64  fp32-oracle-vs-fp64   (max abs err / max abs)")
for k in o64:
    print("  %-22s %.2e       %.2e          %.2e" % (k, T.rel(gf[k], o64[k]), T.rel(gc[k], o64[k]), T.rel(o32[k], o64[k])))
# conditioning of the fea_layer BatchNorm over the B*128 points: per-channel batch std of its input relative to the mean
x = o64["xcorr.mlp_pool"]
w = sd["xcorr.fea_layer.0.conv.weight"].double()[:, :, 0]
y = torch.einsum("oc,bcn->bon", w, x)
print("xcorr.fea_layer.0 conv output: per-channel std/|mean| min %.3e median %.3e; pooled-input per-channel std min %.3e" % (
    float((y.std(dim=(0, 2)) / y.mean(dim=(0, 2)).abs()).min()), float((y.std(dim=(0, 2)) / y.mean(dim=(0, 2)).abs()).median()),
    float(x.std(dim=(0, 2)).min())))
print("xcorr.mlp_pool per-channel std over the points (fp64): min %.3e, count below 1e-3: %d of %d" % (
    float(x.std(dim=(0, 2)).min()), int((x.std(dim=(0, 2)) < 1e-3).sum()), x.shape[1]))
