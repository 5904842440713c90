"""Per-launch conv / deconv durations of the DCCRN forward (BASELINE configs[2]) with the
residue-class row ordering of transposed convolutions on and off (same box, same process).
   python scripts/dccrn_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from aps_amd import nn_ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    _, d = bench.build_dccrn(dev, 0)
    net, mix = d["net"], d["mix"]
    res = {}
    with torch.no_grad():
        for mode in ["class", "plain", "class"]:
            if mode == "plain":
                os.environ["APS_CONV_NO_CLASS"] = "1"
            else:
                os.environ.pop("APS_CONV_NO_CLASS", None)
            for _ in range(2):
                out = net(mix)
            torch.cuda.synchronize()
            nn_ops.CONV_TIMELINE = tl = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                out = net(mix)
            e1.record()
            torch.cuda.synchronize()
            nn_ops.CONV_TIMELINE = None
            per = len(tl) // 5
            rows = []
            for i in range(per):
                us = sum(tl[i + per * k][0].elapsed_time(tl[i + per * k][1]) for k in range(5)) / 5 * 1e3
                rows.append((tl[i][3], us, tl[i][2] / us / 1e6))
            res[mode] = (rows, e0.elapsed_time(e1) / 5, [o.clone() for o in out])
    a, b = res["class"], res["plain"]
    for (d1, u1, tf1), (_, u2, tf2) in zip(a[0], b[0]):
        print(f"{d1:60s} class {u1:8.1f} us {tf1:6.1f} TF | plain {u2:8.1f} us {tf2:6.1f} TF")
    print(f"conv total: class {sum(r[1] for r in a[0]):.0f} us, plain {sum(r[1] for r in b[0]):.0f} us; "
          f"eager step: class {a[1]:.2f} ms, plain {b[1]:.2f} ms")
    err = max((x - y).abs().max().item() for x, y in zip(a[2], b[2]))
    print(f"max |class - plain| over outputs: {err:.3e}")


if __name__ == "__main__":
    main()
