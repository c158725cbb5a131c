"""Which Python line issues which aten op in one eager training step: a TorchDispatchMode logs every aten call that
launches work (copies, cats, element-wise arithmetic, gathers, fills, matmuls), with operand shapes / strides and the
innermost frame inside this repo.  Forward ops and the backward of the repo's own autograd Functions carry a frame; ops
issued by C++ autograd nodes (CatBackward, TransposeBackward, ...) run without a Python stack and print `<autograd>` plus
the forward/backward phase -- their shapes identify them.  torch.profiler's with_stack attribution comes back empty on
this ROCm build (tools/trace_glue.py prints `?`), hence this tool.
usage: trace_dispatch.py [BAT|P2B|M2TRACK] > profiles/rNN_torch_glue.txt"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from open3dsot_amd import dist as D, synth, trackers  # noqa: E402

SKIP = ("aten.view", "aten._unsafe_view", "aten.reshape", "aten.t.", "aten.transpose", "aten.permute", "aten.slice.", "aten.slice_copy",
        "aten.select.int", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.detach", "aten.alias", "aten.as_strided",
        "aten.split", "aten.unbind", "aten.narrow", "aten.empty", "aten.new_empty", "aten._local_scalar", "aten.is_",
        "aten.sym_", "aten.stride", "aten.size", "aten.numel", "aten.dim", "aten.storage_offset", "aten.lift",
        "aten.record_stream", "aten.empty_like", "aten.empty_strided", "aten.new_empty_strided", "aten.unfold",
        "aten._to_copy.default_meta", "aten.split_with_sizes", "aten.view_as", "aten.diagonal", "aten.flatten")


def _desc(a):
    if isinstance(a, torch.Tensor):
        contiguous = a.is_contiguous()
        return "%s%s" % (tuple(a.shape), "" if contiguous else "s" + str(tuple(a.stride())))
    if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
        return "[" + ", ".join(_desc(x) for x in a[:4]) + (", ..." if len(a) > 4 else "") + "]"
    return None


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = []
        self.phase = "forward"

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        out = func(*args, **(kwargs or {}))
        if name.startswith(SKIP):
            return out
        frame = "<autograd>"
        for f in reversed(traceback.extract_stack()):
            fn = f.filename
            if ("/open3dsot_amd/" in fn or fn.endswith("bench.py")) and "/tools/" not in fn:
                frame = "%s:%d %s" % (fn.split("open3dsot_amd/")[-1], f.lineno, f.name)
                break
        shapes = [d for d in (_desc(a) for a in args) if d]
        self.rows.append((self.phase, name.replace("aten.", ""), " ".join(shapes)[:110], frame))
        return out


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "BAT"
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    if name.upper() == "M2TRACK":
        from open3dsot_amd import m2track
        model = m2track.M2TRACK().to(dev).train()
        batch = synth.to_torch(synth.make_motion_batch(0, 48, 1024), dev)
    else:
        model = trackers.get_model(name)().to(dev).train()
        batch = synth.to_torch(synth.make_batch(0, 48), dev)
    trainer = D.DataParallelStep(model, world=1, graph=False)
    for _ in range(2):
        trainer._forward_backward(batch)
        trainer.optimizer.step()
    torch.cuda.synchronize()
    log = Log()
    real_backward = torch.Tensor.backward

    def backward(self, *a, **k):
        log.phase = "backward"
        return real_backward(self, *a, **k)

    torch.Tensor.backward = backward
    try:
        with log:
            trainer._forward_backward(batch)
            log.phase = "optimizer"
            trainer.reduce_gradients()
            trainer.optimizer.step()
    finally:
        torch.Tensor.backward = real_backward
    torch.cuda.synchronize()
    print("%d aten calls that do work in one eager %s step (views / allocations not listed)" % (len(log.rows), name))
    agg = collections.Counter((p, n) for p, n, _, _ in log.rows)
    for (p, n), c in sorted(agg.items(), key=lambda kv: (-kv[1], kv[0])):
        print("  %3d x %-9s %s" % (c, p, n))
    print()
    for p, n, shapes, frame in log.rows:
        print("%-9s %-28s %-60s %s" % (p, n[:28], frame[:60], shapes))


if __name__ == "__main__":
    main()
