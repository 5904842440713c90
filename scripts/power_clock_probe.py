#!/usr/bin/env python
"""What the chip draws and what clock its CUs run at, stage by stage, for the joint step of BASELINE configs[4]
(32 utterances per batch): every stage of the staged pipeline replayed ALONE in a loop, the encoder stage on 1 / 6
streams, and the headline pipeline itself -- board power and sclk from the hwmon files while the loop runs, the
shader clock the conformer-stack kernel really ran at from inside it (s_memtime ticks / s_memrealtime's constant
100 MHz, workgroup 0 of every launch; APS_MEGA_TRACE=1).
    APS_MEGA_TRACE=1 python scripts/power_clock_probe.py"""
import ctypes
import glob
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("APS_MEGA_TRACE", "1")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from aps_amd import _native  # noqa: E402
from aps_amd.replicas import PipelinedReplicas  # noqa: E402

W, P = 6, 12
SECONDS = float(os.environ.get("PROBE_SECONDS", "2.5"))
dev = torch.device("cuda:0")
lib = _native.load()
tbuf = (ctypes.c_ulonglong * 32)()


# (the box may show several cards: every card's files are sampled and the busiest one is reported)
CARDS = [os.path.dirname(f) for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))]
print("hwmon directories:", CARDS)


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.on, self.rows = True, []

    def run(self):
        while self.on:
            row = [time.perf_counter()]
            for c in CARDS:
                try:
                    row.append((int(open(c + "/power1_input").read()) * 1e-6, int(open(c + "/freq1_input").read()) * 1e-9))
                except Exception:  # noqa: BLE001
                    row.append((float("nan"), float("nan")))
            self.rows.append(row)
            time.sleep(0.02)


_, d = bench.build_joint(dev, 0, P, 1)
net, wavs, lens = d["net"], d["wavs"], d["lens"]
net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
with torch.no_grad():
    for b in range(2):
        net(wavs[b], lens)
    torch.cuda.synchronize()
    reps = PipelinedReplicas([lambda b=b: net(wavs[b], lens) for b in range(P)], workers=W, lstm_share=2,
                             front="worker", mid="worker", lookahead=True)
kinds = reps.kinds[0]
print("stages of a batch:", kinds)


def stage(i, kind, stream):
    with torch.cuda.stream(stream):
        reps.pipelines[i][kinds.index(kind)][0].replay()


def measure(name, one, streams_used):
    """`one(j)` issues the j-th unit of work (a stage replay / a submit); batches per call = 1"""
    for j in range(2 * P):
        one(j)
    torch.cuda.synchronize()
    lib.aps_debug_conformer_trace(tbuf)
    s = Sampler()
    s.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < SECONDS:
        for _ in range(P):
            one(n)
            n += 1
        if n % (4 * P) == 0:   # (keep the queues a few rounds deep, not unbounded)
            torch.cuda.synchronize()
    reps.flush()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    s.on = False
    s.join()
    rows = [r for r in s.rows if t0 + 0.5 < r[0] < t1 - 0.1]
    per_card = [(sum(r[1 + c][0] for r in rows) / max(len(rows), 1), sum(r[1 + c][1] for r in rows) / max(len(rows), 1))
                for c in range(len(CARDS))] or [(float("nan"), float("nan"))]
    pw, fq = max(per_card)
    ms = 1e3 * (t1 - t0) / max(n, 1)
    assert lib.aps_debug_conformer_trace(tbuf) == 0
    clk = f"{tbuf[13] / tbuf[14] * 0.1:.3f} GHz inside aps_conformer_stack" if tbuf[14] else "-"
    print(f"{name:44s} {ms:7.3f} ms / batch   {pw:7.0f} W   sclk file {fq:5.2f} GHz   kernel clock {clk}   "
          f"({len(rows)} samples)", flush=True)
    if tbuf[14]:
        busy, wall = sum(tbuf[i] for i in range(12)), sum(tbuf[16 + i] for i in range(12))
        print(f"      workgroup 0: wave 0 busy {busy / tbuf[13]:.3f} of the kernel's ticks, phases barrier to barrier "
              f"{wall / tbuf[13]:.3f}; busy / wall by phase kind: "
              + " ".join(f"{PHASES[i]} {tbuf[i] / max(1, tbuf[16 + i]):.2f}" for i in range(12)))
        print("      share of the kernel by phase kind (barrier to barrier): "
              + " ".join(f"{PHASES[i]} {tbuf[16 + i] / tbuf[13]:.3f}" for i in range(12)) + f" | staging {tbuf[12] / tbuf[13]:.3f}")
    return ms, pw


PHASES = ["ff1_up", "ff1_dn0", "ff1_dn1", "qkv", "att", "out", "pw1", "dwconv", "pw2", "ff2_up", "ff2_dn0", "ff2_dn1"]


def idle(j):
    time.sleep(0.02)


base_ms, base_pw = measure("idle (no launches)", idle, 0)
st = reps.streams
res = {}
res["a"] = measure("stage A alone (STFT, features, in-proj)", lambda j: stage(j % P, "a", st[0]), 1)
res["l"] = measure("stage L alone (persistent LSTM stack)", lambda j: stage(j % P, "l", reps.lstm_stream), 1)
if "m" in kinds:
    res["m"] = measure("stage M alone (masks, MVDR, beamform, log-mel)", lambda j: stage(j % P, "m", st[0]), 1)
res["b1"] = measure("stage B alone, one stream", lambda j: stage(j % P, "b", st[0]), 1)
res["b6"] = measure(f"stage B on {W} streams", lambda j: stage(j % P, "b", st[j % W]), W)
res["pipe"] = measure("the headline pipeline", lambda j: reps.submit(after_caller=False), W + 1)
res["b1_again"] = measure("stage B alone, one stream (again)", lambda j: stage(j % P, "b", st[0]), 1)
print(f"\nenergy per batch above the idle draw ({base_pw:.0f} W): (P - idle) x ms per batch")
for k, (ms, pw) in res.items():
    print(f"   {k:10s} {(pw - base_pw) * ms * 1e-3:8.3f} J   ({pw:.0f} W x {ms:.3f} ms)")
reps.close()
