import os, sys
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aps_amd import nn_ops
from aps_amd.replicas import GraphReplicas, concurrent_launches

th.manual_seed(3)
dev = th.device("cuda:0")
rnn_s = th.nn.LSTM(128, 128, num_layers=2, batch_first=True).eval().to(dev)
x_s = th.randn(8, 20, 128, device=dev)
fn = lambda: nn_ops.lstm_forward(rnn_s, x_s)


def run(tag, serial=False, stack=True, rounds=200):
    nn_ops.LSTM_STACK = stack
    with th.no_grad():
        with concurrent_launches(2):
            eager = fn()
        reps = GraphReplicas(fn, replicas=2)
        bad = [0, 0]
        first = None
        for rnd in range(rounds):
            for _ in range(8):
                reps.submit()
                if serial:
                    th.cuda.synchronize()
            th.cuda.synchronize()
            for i, out in enumerate(reps.outputs):
                if not th.equal(out, eager):
                    bad[i] += 1
                    if first is None:
                        first = (rnd, i)
        print(f"{tag}: bad checks per replica {bad} of {rounds}, first {first}", flush=True)
        del reps


run("first run in the process: " + " ".join(k for k in os.environ if k.startswith("APS_REPL")))
