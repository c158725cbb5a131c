import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from open3dsot_amd import dist as D, synth, trackers, fused
dev = torch.device("cuda", 0)
model = trackers.BAT().to(dev).train()
b = synth.to_torch(synth.make_batch(0, 48), dev)
with torch.no_grad():
    ex = model.sampling_inputs(b)
fb = D.FlatBatch(b, {k: (v.shape, v.dtype) for k, v in ex.items()})
orig = fused.pair_geometry
def spy(grouper, mlp, xyz_a, np_a, si_a, xyz_b, np_b, si_b, out=None):
    print("pair_geometry out:", None if out is None else {k: (tuple(v.shape), v.dtype, v.device, v.is_contiguous()) for k, v in out.items()}, "xyz dev", xyz_a.device)
    return orig(grouper, mlp, xyz_a, np_a, si_a, xyz_b, np_b, si_b, out=out)
fused.pair_geometry = spy
with torch.no_grad():
    ex2 = model.sampling_inputs({k: v for k, v in fb.items() if k not in fb.extra_keys}, out={k: fb[k] for k in fb.extra_keys})
print(fused._GEO_STATS, [ex2[k] is fb[k] for k in ("geo0.gp", "geo1.cw", "geo2.meta")])
