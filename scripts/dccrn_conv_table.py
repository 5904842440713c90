#!/usr/bin/env python
"""Every convolution launch of one DCCRN forward (bench geometry): shape, device time, TFLOP/s.
   python scripts/dccrn_conv_table.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from aps_amd import nn_ops  # noqa: E402

dev = torch.device("cuda:0")
cpu, d = bench.build_dccrn(dev, 0)
net, mix = d["net"], d["mix"]
with torch.no_grad():
    for _ in range(2):
        net(mix)
    torch.cuda.synchronize()
    nn_ops.CONV_TIMELINE = tl = []
    for _ in range(3):
        net(mix)
    torch.cuda.synchronize()
    nn_ops.CONV_TIMELINE = None
n = len(tl) // 3
tot = 0.0
for i in range(n):
    ms = sum(tl[i + k * n][0].elapsed_time(tl[i + k * n][1]) for k in range(3)) / 3
    tot += ms
    print(f"{tl[i][3]:60s} {ms * 1e3:8.1f} us {tl[i][2] / ms / 1e9:7.1f} TF")
print(f"total {tot:.3f} ms")
