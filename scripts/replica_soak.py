#!/usr/bin/env python
"""Soak test for the capture-time corruption GraphReplicas guards against (replicas.py): R graphs of
a small LSTM step captured the way round 1's FAILING order did it -- warm-up on the capture stream
itself, no device-wide stop before the capture -- then replayed for many rounds and compared bit for
bit with the eager result.  Variants isolate what the step contains:
    lstm     the 2 x 128 LSTM stack (persistent kernel, sentinel memset node)
    layers   the same LSTM, one launch per layer
    gemm     GEMMs only
    python scripts/replica_soak.py"""
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd import nn_ops  # noqa: E402


def capture_like_round1(fn, replicas):
    """the order that corrupted replica 0 in round 1"""
    streams, graphs, outs = [], [], []
    nn_ops.push_lstm_share(replicas)
    for _ in range(replicas):
        stream = th.cuda.Stream()
        with th.cuda.stream(stream):
            fn()  # warm-up on the capture stream, not followed by a device synchronise
        graph = th.cuda.CUDAGraph()
        with th.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
            out = fn()
        streams.append(stream), graphs.append(graph), outs.append(out)
    return streams, graphs, outs


def run(tag, fn, replicas=2, rounds=300):
    with th.no_grad():
        nn_ops.push_lstm_share(replicas)
        eager = fn().clone()
        nn_ops.pop_lstm_share(replicas)
        th.cuda.synchronize()
        streams, graphs, outs = capture_like_round1(fn, replicas)
        bad, first = [0] * replicas, None
        for rnd in range(rounds):
            for k in range(8):
                i = k % replicas
                with th.cuda.stream(streams[i]):
                    graphs[i].replay()
            th.cuda.synchronize()
            scratch = th.empty(1 + 37 * (rnd % 7), device="cuda")  # allocator traffic
            for i, out in enumerate(outs):
                if not th.equal(out, eager):
                    bad[i] += 1
                    first = first or (rnd, i)
            del scratch
        nn_ops.pop_lstm_share(replicas)
        print(f"{tag}: mismatching checks per replica {bad} of {rounds}, first {first}, hand-off "
              f"timeouts {nn_ops.lstm_timeouts(check=False)}", flush=True)


def main():
    th.manual_seed(3)
    dev = th.device("cuda:0")
    rnn = th.nn.LSTM(128, 128, num_layers=2, batch_first=True).eval().to(dev)
    lin = th.nn.Linear(128, 96).eval().to(dev)
    x = th.randn(8, 20, 128, device=dev)
    run("lstm stack + gemm", lambda: nn_ops.linear(nn_ops.lstm_forward(rnn, x), lin.weight, lin.bias))
    nn_ops.LSTM_STACK = False
    run("lstm layers + gemm", lambda: nn_ops.linear(nn_ops.lstm_forward(rnn, x), lin.weight, lin.bias))
    nn_ops.LSTM_STACK = True
    run("gemm only", lambda: nn_ops.linear(nn_ops.linear(x, rnn.weight_ih_l0, act="relu")[..., :128],
                                           lin.weight, lin.bias))


if __name__ == "__main__":
    main()
