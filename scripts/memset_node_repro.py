#!/usr/bin/env python
"""No aps_amd kernel in here: does a small hipMemsetAsync captured into a hipGraph (what round 1's
LSTM launcher recorded per launch: a 4-byte clear of its timeout word, allocated from torch's
small-block pool during capture) disturb its neighbours in the same pool segment when the graph is
replayed?  Builds a graph of [memset 4 B][copy kernel over neighbouring small tensors] on a side
stream, after a warm-up on that same stream (round 1's failing order), replays it and checks the
neighbours.     python scripts/memset_node_repro.py"""
import ctypes

import torch as th


def main():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    dev = th.device("cuda:0")
    src = th.arange(4096, dtype=th.float32, device=dev)
    for size in (4, 16, 64):
        for warm_on_capture_stream in (True, False):
            stream = th.cuda.Stream()

            def step():
                word = th.empty(size // 4, dtype=th.int32, device=dev)   # small-pool block
                a = src[:1000] * 2.0                                      # neighbours in the pool
                rc = hip.hipMemsetAsync(word.data_ptr(), 0, size,
                                        th.cuda.current_stream().cuda_stream)
                assert rc == 0
                b = a + 1.0
                c = th.empty(300, device=dev).copy_(b[:300])
                return a, b, c, word

            if warm_on_capture_stream:
                with th.cuda.stream(stream):
                    step()
            else:
                step()
                th.cuda.synchronize()
            g = th.cuda.CUDAGraph()
            with th.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                a, b, c, word = step()
            want_a, want_b = src[:1000] * 2.0, src[:1000] * 2.0 + 1.0
            th.cuda.synchronize()
            bad = 0
            for rnd in range(300):
                with th.cuda.stream(stream):
                    g.replay()
                th.cuda.synchronize()
                junk = th.empty(1 + 37 * (rnd % 7), device=dev)
                ok = th.equal(a, want_a) and th.equal(b, want_b) and th.equal(c, want_b[:300]) and \
                    int(word.abs().sum()) == 0
                bad += 0 if ok else 1
                del junk
            print(f"memset {size} B, warm-up on the capture stream: {warm_on_capture_stream}: "
                  f"{bad} of 300 replays disturbed", flush=True)


if __name__ == "__main__":
    main()
