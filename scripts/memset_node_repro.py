#!/usr/bin/env python
"""No aps_amd kernel in here.  Round 1 found graph replicas whose outputs went wrong "a few replays
later" and worked around it by ordering (warm-up on the caller's stream).  This isolates the cause:
a hipMemsetAsync captured into a hipGraph stops taking effect after about a hundred replays.
  graph = [ hipMemsetAsync(buf, 0xff, size) ] -> [ count = (buf == 0xffffffff).sum() ] -> [ buf.fill_(1) ]
Every replay must count every word of buf as re-armed; a replay whose memset node did nothing sees
the 1.0 the previous replay left behind.  The control arm re-arms with a fill KERNEL instead.
This is the protocol of the persistent LSTM kernels (lstm.hip: the layer output is pre-filled with a
sentinel and "the data is the flag"), which is why the launcher re-arms with a kernel, not a memset.
    python scripts/memset_node_repro.py"""
import ctypes

import torch as th


def main():
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    dev = th.device("cuda:0")
    print(f"torch {th.__version__}, hip {th.version.hip}", flush=True)
    for words in (1, 16, 4096, 1 << 20, 8 << 20):
        for arm in ("memset node", "fill kernel"):
            for warm_on_capture_stream in (True, False):
                stream = th.cuda.Stream()
                buf = th.empty(words, dtype=th.int32, device=dev)
                count = th.zeros(1, dtype=th.int64, device=dev)

                def step():
                    if arm == "memset node":
                        rc = hip.hipMemsetAsync(buf.data_ptr(), 0xff, 4 * words,
                                                th.cuda.current_stream().cuda_stream)
                        assert rc == 0
                    else:
                        buf.fill_(-1)
                    count.copy_((buf == -1).sum())
                    buf.fill_(1)

                if warm_on_capture_stream:
                    with th.cuda.stream(stream):
                        step()
                else:
                    step()
                    th.cuda.synchronize()
                g = th.cuda.CUDAGraph()
                with th.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                    step()
                th.cuda.synchronize()
                bad, first = 0, None
                for rnd in range(400):
                    with th.cuda.stream(stream):
                        g.replay()
                    th.cuda.synchronize()
                    if int(count.item()) != words:
                        bad += 1
                        first = rnd if first is None else first
                print(f"{4 * words:9d} B, {arm:11s}, warm-up on the capture stream {str(warm_on_capture_stream):5s}: "
                      f"{bad:3d} of 400 replays NOT re-armed (first: {first})", flush=True)


if __name__ == "__main__":
    main()
