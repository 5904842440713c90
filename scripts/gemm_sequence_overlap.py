#!/usr/bin/env python
"""Do the projections of two batches overlap when NOTHING else is in the way?  The two-plane GEMM launches
of one 32-utterance joint step are recorded per resident batch (nn_ops.GEMM_RECORD) and re-issued (a) on one
stream, batch after batch, (b) on two streams, one batch each, the second started half a sequence late (so
the two streams sit in different layers, as two replicas do).  Also: the same for the WHOLE step as captured
graphs with the LSTM taken out of the comparison by timing the GEMM-only lists beside them.
    python scripts/gemm_sequence_overlap.py            (on an MI355X)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from aps_amd import nn_ops  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    _, d = bench.build_joint(dev, 0, 4, 1)
    net, wavs, lens = d["net"], d["wavs"], d["lens"]
    net.enh_transform.nan_policy = net.asr_transform.nan_policy = "deferred"
    recs = []
    with torch.no_grad():
        for w in wavs[:2]:
            net(w, lens)
        torch.cuda.synchronize()
        for w in wavs:
            nn_ops.GEMM_RECORD = rec = []
            net(w, lens)
            torch.cuda.synchronize()
            nn_ops.GEMM_RECORD = None
            recs.append([c for c, _, _, _ in rec])
            keep = rec  # noqa: F841  (operands stay alive through `recs`' closures)
    n = len(recs[0])
    print(f"{n} two-plane GEMM launches per step")
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    reps = 8

    def run(two: bool):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if not two:
            with torch.cuda.stream(streams[0]):
                for r in range(reps):
                    for b in range(2):
                        for c in recs[b]:
                            c_stream(c, streams[0])
        else:
            # interleave the host-side issue so that both queues stay fed
            seqs = [[c for _ in range(reps) for c in recs[0]], [c for _ in range(reps) for c in recs[1]]]
            off = n // 2
            for i in range(len(seqs[0]) + off):
                if i < len(seqs[0]):
                    c_stream(seqs[0][i], streams[0])
                if i >= off:
                    c_stream(seqs[1][i - off], streams[1])
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / (2 * reps)

    # the recorded closures carry the stream they were recorded on (the default one): re-issue through the
    # C-ABI with another stream by patching the last argument
    def c_stream(c, stream):
        fn, fargs = c.__defaults__
        fn(*fargs[:-1], torch.cuda.current_stream().cuda_stream if stream is None else stream.cuda_stream)

    for label, two in (("one stream ", False), ("two streams", True), ("one stream ", False), ("two streams", True)):
        ms = run(two)
        print(f"{label}: {ms:.3f} ms per batch of GEMMs ({1e3 * ms / n:.1f} us per launch)")

    # the same single-stream GEMM sequence while the mask estimator's LSTM stack (one persistent launch of
    # 256 workgroups, 0.85 ms) runs back to back on another stream: what the LSTM of one replica costs the
    # projections of the other
    rnn = None
    for m in net.enh_net.modules():
        if isinstance(m, torch.nn.LSTM):
            rnn = m
    x = torch.randn(32, 249, rnn.input_size, device=dev)
    lstm_stream = torch.cuda.Stream()
    nn_ops.push_lstm_share(2)
    with torch.no_grad():
        nn_ops.lstm_forward(rnn, x)
        torch.cuda.synchronize()
        for with_lstm in (False, True, False, True):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            k = 0
            for r in range(reps):
                for b in range(2):
                    if with_lstm:
                        with torch.cuda.stream(lstm_stream):
                            nn_ops.lstm_forward(rnn, x)
                            k += 1
                    for c in recs[b]:
                        c_stream(c, streams[0])
            streams[0].synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / (2 * reps)
            torch.cuda.synchronize()
            print(f"GEMMs of one stream, LSTM stack on another: {'yes' if with_lstm else 'no '}  {ms:.3f} ms per batch "
                  f"({1e3 * ms / n:.1f} us per launch; {k} LSTM launches queued)")
    nn_ops.pop_lstm_share(2)


if __name__ == "__main__":
    main()
