#!/usr/bin/env python
"""Which stock torch (aten) ops still run inside the joint step, and from which line of aps_amd?
One eager 32-utterance step under a TorchDispatchMode: every aten op that touches a GPU tensor and is not a pure
view / metadata op is listed with the innermost aps_amd frames of its Python stack.
    python scripts/step_torch_ops.py
"""
import os
import sys
import traceback
from collections import Counter

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

VIEWS = ("view", "reshape", "transpose", "permute", "slice", "select", "unsqueeze", "squeeze", "expand", "as_strided",
         "detach", "alias", "t.default", "unbind", "split", "narrow", "_unsafe_view", "empty", "size", "stride",
         "is_", "sym_", "dim", "numel", "_to_copy.default_meta", "lift_fresh", "unfold", "chunk", "view_as")


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.seen = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        flat = [a for a in torch.utils._pytree.tree_leaves((args, kwargs)) if isinstance(a, torch.Tensor)]
        outs = [a for a in torch.utils._pytree.tree_leaves(out) if isinstance(a, torch.Tensor)]
        if not any(t.is_cuda for t in flat + outs):
            return out
        short = name.replace("aten.", "")
        if any(short.startswith(v) for v in VIEWS):
            return out
        frames = [f"{os.path.relpath(fr.filename)}:{fr.lineno} {fr.name}" for fr in traceback.extract_stack()
                  if "aps_amd" in fr.filename][-3:]
        self.seen[(short, " <- ".join(reversed(frames)))] += 1
        return out


dev = torch.device("cuda:0")
_, d = bench.build_joint(dev, 0, 1, 1)
net, wav, lens = d["net"], d["wavs"][0], d["lens"]
net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"   # as the captured step runs
with torch.no_grad():
    for _ in range(3):
        net(wav, lens)
    torch.cuda.synchronize()
    with Census() as c:
        net(wav, lens)
    torch.cuda.synchronize()
print(f"{sum(c.seen.values())} aten ops on GPU tensors in one step (views and allocations left out):")
for (op, where), n in sorted(c.seen.items(), key=lambda kv: -kv[1]):
    print(f"  {n:3d} x {op:32s} {where}")
