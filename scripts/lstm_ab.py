"""time the joint step's mask-estimator LSTM (2 x 512, N = 32, T = 249) with whatever library
APS_AMD_LIB points at: same-box A/B of two builds"""
import os, sys
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd import nn_ops
th.manual_seed(0)
dev = th.device("cuda:0")
rnn = th.nn.LSTM(512, 512, num_layers=2, batch_first=True).eval().to(dev)
x = th.randn(32, 249, 512, device=dev)
with th.no_grad():
    for _ in range(5):
        nn_ops.lstm_forward(rnn, x)
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        nn_ops.lstm_forward(rnn, x)
    e1.record()
    th.cuda.synchronize()
print(os.environ.get("APS_AMD_LIB", "in-tree"), f"{e0.elapsed_time(e1) / 50 * 1e3:.1f} us per forward (input GEMM included)")
