import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from open3dsot_amd import synth, ext
b = synth.make_batch(0, 48, 512, 1024)
for key, m in (("search_points", 512), ("template_points", 256)):
    x = torch.from_numpy(b[key]).cuda()
    for _ in range(3): ext.furthest_point_sampling(x, m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ext.furthest_point_sampling(x, m)
    e1.record(); torch.cuda.synchronize()
    print(key, x.shape, "->", m, "%.3f ms" % (e0.elapsed_time(e1) / 20))
