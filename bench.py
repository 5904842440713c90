#!/usr/bin/env python
"""
bench.py -- throughput of the aps joint front-end hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
        N > 1 without a torchrun environment: bench.py re-executes itself through
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`
        (one rank per GPU, RCCL); launched by torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.

A "step" is one pass of the hot path over one batch of synthetic utterances already resident in
HBM.  Default workload = BASELINE.json configs[4], the configuration the metric
"utterances/sec (4-ch 16 kHz 4 s) STFT->MVDR->encoder fwd" is quoted on; per GPU a step covers 32
utterances (BASELINE's global batch 256 over 8 GPUs; --global-batch B sets group = B / (32 x ranks)):
    EnhTransform STFT + log-magnitude/CMVN + cos-IPD -> RNNMaskMvdr (LSTM mask estimator, mask
    MVDR: covariance x2, channel attention, per-bin complex solve, beamform) ->
    AsrTransform abs-mel-log-cmvn -> 12-layer conformer encoder (conf/asr/chime4/1a.yaml) + CTC head
The same line carries `merged_batch`: 4 x 32 utterances fused into one launch sequence per GPU
(utterances are independent; a larger per-GPU batch than BASELINE's, reported as an extra).
Other workloads: --workload frontend (configs[1]: STFT + features + MVDR with given masks, the
HBM-bound stage), --workload encoder (configs[3]) and --workload dccrn (configs[2]).

One process per GPU, utterances sharded by rank (weak scaling, no collective on the data path).
The inputs ROTATE: --batches P distinct batches are resident (12 x 32.8 MB of waveforms: more than the
256 MB Infinity Cache; every batch also owns its intermediates),
so no replay finds its input in a cache.  Every batch's step is captured once -- joint workload (rounds 5 - 6):
as FOUR hipGraphs cut at the mask estimator's persistent LSTM launch and behind the front end: the LSTM launches of
all batches one after the other on the head stream, STFT + features, the front end's tail and the encoder stage -- whose
12 conformer layers are ONE launch per batch, a workgroup per utterance (aps_conformer_stack, round 6) -- round-robin
on --pipeline (6) worker streams, a batch's front launched --pipeline submissions ahead of its back
(aps_amd.replicas.PipelinedReplicas(lookahead=True)); with --replicas R (or --pipeline 0) as ONE hipGraph replayed
round-robin on R streams (GraphReplicas: rounds 2-4's mode, still measured and reported as `whole_step_replicas`).
W untimed warm-up steps, then --repeats (default 5) timed regions of exactly K steps, each between
barrier + synchronize pairs and reduced with max over ranks; `ms_per_step` / `value` come from the
MEDIAN region, min / max are reported next to it.  A region pays the staged pipeline's fill and drain once (~4.4 ms
whatever K): a line with K < 100 also carries `steady_state`, the same measurement on regions of 100 steps.

The JSON line also carries
  roofline       : the dominant kernel (round 6: conformer_stack_kernel): MFMA flops executed per second with the
                   headline's launches in flight against the dense f16 peak, the launch ALONE (32 of 256 CUs by design)
                   under `per_launch`, the projections still launched one by one under `per_launch_projections`
  stage_roofline : STFT / features / MVDR weights / beamform: ALGORITHMIC bytes /
                   event-timed duration per launch over the rotating batches, against the HBM peak
  host_input     : the PCIe-inclusive rate (every step's waveforms copied from pinned host memory; never `value`),
  latency_ms_per_batch, stage_ms_under_load : GPU-clock latency of a batch and of every stage inside the pipeline
  parity         : the GPU outputs of the first utterances of batch 0 against the CPU oracle
                   (scaled max error; the run FAILS above 1e-4)
  cpu_baseline   : the CPU oracle (a torch-CPU port of the reference, oracle/) timed on this box's
                   host cores on a bounded sample of the same workload (rank 0, N = 1)
  ranks_seen     : all-reduced count of the ranks that ran the timed regions
"""
import argparse
import json
import os
import socket
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The joint headline keeps EIGHT streams busy (six worker streams, the head stream, the caller's): each needs a
# hardware queue of its own, and the HIP runtime multiplexes streams onto 4 unless told otherwise when it starts
# (aps_amd/replicas.py: PipelinedReplicas; profiles/r05_pipeline_sweep.txt: 11.3 k against 14.7 k utt/s)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

METRIC = "utterances/sec (4-ch 16 kHz 4 s) STFT→MVDR→encoder fwd, 1/2/4/8 MI355X"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 MFMA peak
# dense bf16 MFMA peak (16 x the fp32 rate, MI355X_MICROARCH.md): the pipe gemm_split_kernel runs on,
# 6 bf16 products per fp32 product
MFMA_BF16_PEAK_TFLOPS = 16 * MFMA_F32_PEAK_TFLOPS
SPLIT_PRODUCTS = 6
PARITY_TOL = 1e-4

# workload constants
BATCH, CH, SAMPLES = 32, 4, 64000
FRAME_LEN, FRAME_HOP, BINS, PAIRS = 512, 256, 257, 3
FRAMES = (SAMPLES - FRAME_LEN) // FRAME_HOP + 1  # 249

# ALGORITHMIC bytes per utterance and stage (fp32; SURVEY.md 8d, DESIGN.md "bytes per unit"): what each
# launch group of THIS build must read and write at the least (per-kernel figures), and next to them the
# survey's minimum-traffic three-pass schedule the judge prices the whole stage against
X_BYTES = CH * BINS * FRAMES * 8
FEATS_BYTES = FRAMES * BINS * (1 + PAIRS) * 4
ALGO_BYTES = {
    "stft": CH * SAMPLES * 4 + X_BYTES,                                   # R wav + W X
    "features": X_BYTES + FEATS_BYTES,                                    # R X + W feats
    "stft_features": CH * SAMPLES * 4 + X_BYTES + FEATS_BYTES,            # R wav + W X + W feats (8d P1)
    "mvdr_weights": X_BYTES + 2 * FRAMES * BINS * 4 + BINS * CH * 8 + CH * 4,  # R X, masks; W w, u
    "beamform": X_BYTES + BINS * CH * 8 + FRAMES * BINS * 8,              # R X, w; W Y
    "asr_features": FRAMES * BINS * 8 + FRAMES * 80 * 4,                  # R Y; W log-mel
    "beamform_features": X_BYTES + BINS * CH * 8 + FRAMES * 80 * 4,       # R X, w; W log-mel (8d P3: no Y)
}
# P1 STFT + features 4 095 664, P2 covariance 2 625 512, P3 solve + beamform + |.| + mel +
# log 2 193 248, + W Y 511 944 when the beam output is returned (EnhASRBase does) = 9 426 368 B / utterance
SURVEY_8D_BYTES = (CH * SAMPLES * 4 + X_BYTES + FEATS_BYTES) + \
    (X_BYTES + 2 * FRAMES * BINS * 4 + 2 * BINS * CH * CH * 8) + \
    (2 * BINS * CH * CH * 8 + X_BYTES + FRAMES * 80 * 4) + FRAMES * BINS * 8
assert SURVEY_8D_BYTES == 9426368, SURVEY_8D_BYTES
STAGE_KERNELS = {
    "stft": "stft512_wave_kernel",
    "features": "features_rows_kernel<5>",
    "stft_features": "stft512_frame_feat_kernel (STFT + log-magnitude / CMVN + IPD in one pass, X written once)",
    "mvdr_weights": "covariance_partial_kernel<4,64> + covariance_finalize_kernel + attention_partial_kernel + weight_kernel",
    "beamform": "beamform_kernel<4>",
    "asr_features": "features_kernel<1> (|Y| -> 80 mel -> log -> cmvn)",
    "beamform_features": "beamform_features_kernel<4> (beamform -> |Y| -> 80 mel -> log -> cmvn in one pass, Y not written)",
}


# ---------------------------------------------------------------------------------------------
# launch / timing plumbing shared by the workloads
# ---------------------------------------------------------------------------------------------
# where host_input_rate queues the host -> device copies.  Round 6 (front end and LSTM launches on the workers /
# the head stream, lookahead; profiles/r06_pipeline_sweep.txt): the CALLER's stream ("null": nothing else runs on it,
# the batch's first stage waits for its head) 19.06 k utt/s against 19.71 k resident; the batch's worker stream 14.8 -
# 15.2 k, a pool stream of its own 13.0 k (it shares a hardware queue with a worker), the head stream 11.2 k (that
# stream carries the persistent LSTM launches back to back now).  Round 5's pipeline had the head stream idle enough
# (12.9 - 13.3 k there against 11.7 k on the worker).
HOST_INPUT_STREAM = "null"


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int) -> None:
    """`python bench.py --gpus N` outside torchrun: become the launcher of N ranks (one per GPU),
    the shape of scripts/distributed_train.sh:62-113 of the reference"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] launching", " ".join(cmd), file=sys.stderr, flush=True)
    os.execvp(cmd[0], cmd)


class Ranks(object):
    """process-group context of one bench process"""

    def __init__(self, args):
        from aps_amd import distributed as D
        self.D = D
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.cpu_only = args.selftest_launch and not torch.cuda.is_available()
        local = D.local_rank()
        if self.world > 1:
            if not self.cpu_only:
                if local >= torch.cuda.device_count():
                    raise RuntimeError(f"rank with LOCAL_RANK={local} but only "
                                       f"{torch.cuda.device_count()} GPU(s) visible: one rank per GPU")
                torch.cuda.set_device(local)
            D.init("torch", "gloo" if self.cpu_only else "nccl")
            self.cores = D.bind_rank_to_cores()  # each rank on its own share of the host cores
        self.rank = D.rank()
        self.device = torch.device("cpu") if self.cpu_only else \
            torch.device("cuda", local if self.world > 1 else 0)
        if not self.cpu_only:
            torch.cuda.set_device(self.device)
        if self.world != args.gpus and self.rank == 0:
            print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={self.world}",
                  file=sys.stderr)

    def sync(self):
        if not self.cpu_only:
            torch.cuda.synchronize()

    def ranks_seen(self) -> int:
        return int(round(self.D.reduce_sum(1.0, self.device)))

    def inputs_distinct(self, checksum: float):
        """do the ranks hold DIFFERENT input batches?  (sum and sum of squares of one checksum per rank:
        equal checksums on every rank would mean the same shard N times)  None on one rank."""
        if self.world == 1:
            return None
        s1 = self.D.reduce_sum(float(checksum), self.device)
        s2 = self.D.reduce_sum(float(checksum) ** 2, self.device)
        return bool(s2 * self.world - s1 * s1 > 1e-6 * max(s2, 1e-30))


def timed_regions(R: Ranks, steps: int, repeats: int, submit, units_per_step: int, flush=None):
    """`repeats` regions of exactly `steps` steps, each between barrier + synchronize pairs, the
    elapsed time of a region = max over ranks.  Returns (region seconds list, units all ranks
    processed per region).  flush: called behind the region's last submit (a submitter that launches part of a step
    ahead -- PipelinedReplicas(lookahead=True) -- launches what is still pending: every region is whole steps)."""
    regions = []
    for _ in range(repeats):
        R.D.barrier()
        R.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            submit()
        if flush is not None:
            flush()
        R.sync()
        R.D.barrier()
        regions.append(R.D.reduce_max(time.perf_counter() - t0, R.device))
    units = R.D.reduce_sum(float(units_per_step * steps), R.device)
    return regions, units


def region_stats(regions, steps):
    ms = sorted(1e3 * r / steps for r in regions)
    return {"median": round(statistics.median(ms), 4), "min": round(ms[0], 4),
            "max": round(ms[-1], 4), "repeats": len(ms)}


def base_line(args, R: Ranks, regions, units, workload: dict, dtype="f32"):
    med = statistics.median(regions)
    return {
        "metric": METRIC, "value": round(units / med, 1), "unit": "utt/s", "n_gpus": R.world,
        "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * med / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype,
        "data": "synthetic (random-init weights, seeded noise waveforms, rotating resident batches)",
        "config": workload, "ms_per_step_regions": region_stats(regions, args.steps),
    }


def hold_stream(cycles: int) -> None:
    """keep the current stream busy for `cycles` spin cycles (no-op when 0)"""
    if cycles > 0:
        torch.cuda._sleep(cycles)


def spin_cycles_for(ms: float) -> int:
    """cycle count that makes torch.cuda._sleep occupy the stream for about `ms` milliseconds
    (0 when the spin kernel is not available: the brackets then include host launch gaps)"""
    if not hasattr(torch.cuda, "_sleep"):
        return 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    probe = 2_000_000
    torch.cuda._sleep(probe)
    torch.cuda.synchronize()
    e0.record()
    torch.cuda._sleep(probe)
    e1.record()
    torch.cuda.synchronize()
    per_ms = probe / max(e0.elapsed_time(e1), 1e-3)
    return int(min(per_ms * ms, 2_000_000_000))


def empty_bracket_us(spin: int) -> float:
    """what an event bracket costs by itself on a busy queue (median of 64)"""
    hold_stream(spin)
    empty = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
             for _ in range(64)]
    for a, b in empty:
        a.record()
        b.record()
    torch.cuda.synchronize()
    return sorted(1e3 * a.elapsed_time(b) for a, b in empty)[len(empty) // 2]


def scaled_err(got: torch.Tensor, ref: torch.Tensor) -> float:
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def cpu_info() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def timed_cpu(fn, units: int, threads: int, budget_s: float, max_iters: int = 20, warm=True):
    """utt/s of fn() (which processes `units` utterances) with `threads` torch threads"""
    before = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        if warm:
            fn()
        t0, iters = time.perf_counter(), 0
        while True:
            fn()
            iters += 1
            el = time.perf_counter() - t0
            if el > budget_s or iters >= max_iters:
                break
    finally:
        torch.set_num_threads(before)
    return units * iters / el, iters, el


def cpu_baseline_of(fn, units: int, what: str, budget_s: float = 12.0):
    """The CPU oracle on the host cores.  Op-by-op torch on a few utterances stops scaling well below
    the 128-256 cores of the GPU boxes (round 1 measured 0.35 utt/s at 128 threads against 25 at 32),
    so the thread count is PROBED -- one timed call each at 16 / 32 / 64 threads -- and the best one
    is then timed for `budget_s` seconds; the 1-thread rate is reported next to it."""
    cores = os.cpu_count() or 1
    probes = {}
    for t in (16, 32, 64):
        if t <= cores:
            probes[t] = timed_cpu(fn, units, t, 0.0, max_iters=1)[0]
    many = max(probes, key=probes.get) if probes else cores
    v_many, it, el = timed_cpu(fn, units, many, budget_s, max_iters=60, warm=False)
    v_one, it1, el1 = timed_cpu(fn, units, 1, budget_s / 3, max_iters=4, warm=False)
    return {"value": round(v_many, 2), "unit": "utt/s", "cores": many, "kind": "port",
            "one_thread_value": round(v_one, 3), "host_cores": cores, "cpu": cpu_info(),
            "torch": torch.__version__,
            "thread_probe_utt_s": {str(k): round(v, 2) for k, v in probes.items()},
            "sample": f"{it} {what} of {units} utterances ({el:.1f} s, torch-CPU oracle, {many} "
                      f"threads = the best of the probed counts; 1 thread: {it1} in {el1:.1f} s)"}


# ---------------------------------------------------------------------------------------------
GEMM_KERNELS = {
    # kind in nn_ops.GEMM_TIMELINE -> (kernel, MFMA instruction, MFMA products per fp32 product, peak)
    "f32": ("gemm_f32_kernel", "v_mfma_f32_32x32x2_f32", 1, MFMA_F32_PEAK_TFLOPS),
    "split-bf16": ("gemm_split_bd_kernel", "v_mfma_f32_32x32x16_bf16", SPLIT_PRODUCTS, MFMA_BF16_PEAK_TFLOPS),
    "split-fp16": ("gemm_fp16x2_kernel", "v_mfma_f32_32x32x16_f16", 3, MFMA_BF16_PEAK_TFLOPS),
    "split-panel": ("gemm_panel_kernel", "v_mfma_f32_32x32x16_f16", 3, MFMA_BF16_PEAK_TFLOPS),
    "split-kgroup": ("gemm_kgroup_kernel", "v_mfma_f32_32x32x16_f16", 3, MFMA_BF16_PEAK_TFLOPS),
}
DTYPES = {"f32": "f32 (fp32 MFMA)", "split-bf16": "f32 io / 6 x bf16 MFMA exact split, f32 accumulate",
          "split-fp16": "f32 io / 3 x f16 MFMA two-plane split, f32 accumulate",
          "split-panel": "f32 io / 3 x f16 MFMA two-plane split, f32 accumulate",
          "split-kgroup": "f32 io / 3 x f16 MFMA two-plane split, f32 accumulate"}
_KIND_NAMES = {"panel": "split-panel", "kgroup": "split-kgroup"}
_KIND_KEYS = {v: k for k, v in _KIND_NAMES.items()}


def gemm_roofline(timeline, passes, bracket_us, where, replayed=None):
    """roofline object of the dominant GEMM kernel from (start, stop, flops, kernel) brackets.
    `achieved` / `peak` / `frac` price the MFMA flops the kernel EXECUTES against the dense peak of
    the matrix pipe it runs on (the fp16 two-plane form executes three f16 products per fp32
    product, the bf16 form six): a utilisation figure, always <= 1.  The ALGORITHMIC rate (2 M N K
    per launch, what earlier rounds divided by the fp32 MFMA peak) stays next to it as
    `algorithmic`.  Durations: summed launch brackets minus the cost of an empty bracket."""
    from aps_amd import nn_ops
    # ("split" under layout 3 = the launches nn_ops.linear hands to the planes-pass kernel: many column tiles)
    split_kind = {1: "split-bf16", 2: "split-fp16", 3: "split-fp16"}.get(nn_ops.SPLIT_LAYOUT, "split-bf16")
    kinds = {}
    for a, b, f, kind in timeline:
        k = kinds.setdefault({"split": split_kind, **_KIND_NAMES}.get(kind, kind), [0.0, 0.0, 0])
        k[0] += a.elapsed_time(b)
        k[1] += f
        k[2] += 1
    name = max(kinds, key=lambda k: kinds[k][0])
    raw_ms, flop, n = (v / passes for v in kinds[name])
    launches = int(round(n))
    bracket_ms = ms = raw_ms - launches * bracket_us * 1e-3
    # (replayed: kind of nn_ops -> the back-to-back re-issue of the step's launches, see measure_joint)
    rkey = _KIND_KEYS.get(name, "split")
    rep = (replayed or {}).get(rkey)
    if rep is not None and rep["launches"] == launches:
        ms = rep["ms_per_step"]
    algo = flop / (ms * 1e-3) / 1e12
    kernel, instr, products, peak = GEMM_KERNELS[name]
    executed = products * algo
    out = {"kernel": f"{kernel} ({launches} launches / {where})", "bound": "mfma",
           "achieved": round(executed, 1), "peak": peak, "unit": "TFLOP/s",
           "frac": round(executed / peak, 4), "traffic": None,
           "instruction": instr, "mfma_products_per_fp32_product": products,
           "algorithmic": {"achieved": round(algo, 2), "unit": "TFLOP/s",
                           "vs_fp32_mfma_peak": round(algo / MFMA_F32_PEAK_TFLOPS, 4),
                           "note": "2 M N K per launch / kernel time; the figure rounds 1-2 called "
                                   "roofline.frac (it exceeds 1 once the arithmetic leaves the fp32 pipe)"},
           "algo_flops_per_step": flop, "kernel_ms_per_step": round(ms, 4),
           "kernel_us_per_launch": round(1e3 * ms / max(launches, 1), 2),
           "timing": ("the step's two-plane GEMM launches re-issued back to back, in their own order, between one "
                      "pair of HIP events (6 passes); this kernel's share of that time = its share of the "
                      "per-launch brackets.  A WARM-CACHE figure: re-issued alone, the launches find operands and "
                      "weight images warmer than inside a real step (nothing else runs between them), so the rate "
                      "is an upper bound on the in-step one -- bracket_corrected_ms_per_step is the in-step "
                      "measurement, the rocprof averages of the real step are under profiles/" if ms is not bracket_ms else "a HIP-event pair around every launch, "
                      "empty bracket subtracted"),
           "bracketed_ms_per_step": round(raw_ms, 4), "bracket_corrected_ms_per_step": round(bracket_ms, 4),
           "empty_bracket_us": round(bracket_us, 2),
           "dtype": DTYPES[name]}
    if name == "split-kgroup":
        out["note"] = ("fp32 in / fp32 out on 3 f16 MFMA products of two-plane operand splits; a 32 x 128 tile is owned "
                       "by 16 waves: K is cut into 4 groups of 128 / 256 columns, every group forms the planes of ITS "
                       "columns (a power-of-two scale per row and group), multiplies them against the weight image and "
                       "the 4 partial tiles are summed in LDS in group order (bit-reproducible); cross terms in their "
                       "own accumulator; every output within 2^-19 sum|a||w| for any finite input (tiles whose operands "
                       "leave the planes' range are recomputed on the fp32 MFMA inside the launch); no pass over A "
                       "outside the kernel")
    elif name == "split-panel":
        out["note"] = ("fp32 in / fp32 out on 3 f16 MFMA products of two-plane operand splits; the planes of A are "
                       "formed inside the kernel per 256- / 128-wide K chunk (a power-of-two scale per row and "
                       "chunk, the chunks folded into an fp32 sum), those of W come from its image; cross terms in "
                       "their own accumulator; every output within 2^-19 sum|a||w| for any finite input (tiles "
                       "whose operands leave the planes' range are recomputed on the fp32 MFMA inside the "
                       "launch); no pass over A outside the kernel")
    elif name == "split-fp16":
        out["note"] = ("fp32 in / fp32 out on 3 f16 MFMA products of two-plane operand splits, a power-of-two "
                       "scale per operand row, cross terms in their own accumulator; every output within "
                       "2^-19 sum|a||w| for any finite input (tiles whose operands leave the planes' range "
                       "are recomputed on the fp32 MFMA inside the launch; 2^-20.5 measured otherwise); the "
                       "row-exponent pass over A is inside the brackets")
    elif name == "split-bf16":
        out["note"] = ("fp32 in / fp32 out, as accurate as the fp32 MFMA, evaluated as 6 bf16 MFMA "
                       "products of exact three-way operand splits")
    others = {}
    for k, v in kinds.items():
        if k == name:
            continue
        o_ms, o_n = v[0] / passes - v[2] / passes * bracket_us * 1e-3, int(round(v[2] / passes))
        o_rep = (replayed or {}).get(_KIND_KEYS.get(k, "split"))
        if o_rep is not None and o_rep["launches"] == o_n:
            o_ms = o_rep["ms_per_step"]  # (the same back-to-back timing as the dominant kernel's)
        others[k] = {"launches": o_n, "ms_per_step": round(o_ms, 4),
                     "TFLOP/s algorithmic": round(v[1] / passes / max(o_ms, 1e-9) / 1e9, 2)}
    if others:
        out["other_gemm_kernels"] = others
    if replayed and "_all" in replayed:
        out["all_two_plane_gemms"] = {"launches": replayed["_all"]["launches"],
                                      "ms_per_step": round(replayed["_all"]["ms_per_step"], 4)}
    return out


def mega_roofline(record, streams: int):
    """roofline object of aps_conformer_stack (csrc/conformer_mega.hip), the dominant kernel when the conformer stack
    runs as one launch per batch: HIP events on the launch stream around `reps` back-to-back re-issues of the recorded
    launch (its own clone of the input each time) -> the kernel ALONE: a launch is 32 workgroups = 32 of the 256 CUs
    by design, so `per_launch.frac` is bounded by 1 / 8; then the same launch on `streams` streams at once (what the
    pipelined headline keeps in flight) -> `achieved` / `frac`: the MFMA flops the chip executes per second with the
    headline's concurrency against the dense f16 peak."""
    from aps_amd import mega
    enc, x, lens, rel = record
    N, T = int(x.shape[0]), int(x.shape[1])
    algo = mega.projection_flops(enc, N, T)
    for _ in range(2):
        mega.conformer_stack(enc, x, lens, rel)
    torch.cuda.synchronize()
    reps = 6
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        mega.conformer_stack(enc, x, lens, rel)
    e1.record()
    torch.cuda.synchronize()
    alone_ms = e0.elapsed_time(e1) / reps
    from aps_amd.replicas import replica_streams
    ss = replica_streams(x.device, streams)
    xs = [x.clone() for _ in range(streams)]
    rounds = 4

    def round_():
        for i, st in enumerate(ss):
            with torch.cuda.stream(st):
                mega.conformer_stack(enc, xs[i], lens, rel)
    round_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rounds):
        round_()
    torch.cuda.synchronize()
    flight_ms = 1e3 * (time.perf_counter() - t0) / rounds
    peak = GEMM_KERNELS["split-panel"][3]
    ex_alone = 3 * algo / (alone_ms * 1e-3) / 1e12
    ex_flight = 3 * algo * streams / (flight_ms * 1e-3) / 1e12
    wgs = N
    return {"kernel": f"conformer_stack_kernel (aps_conformer_stack: {len(enc.layers)} conformer layers of a batch in ONE "
                      f"launch, a workgroup per utterance: {wgs} workgroups of 512 threads = {wgs} of 256 CUs)",
            "bound": "mfma", "achieved": round(ex_flight, 1), "peak": peak, "unit": "TFLOP/s",
            "frac": round(ex_flight / peak, 4), "traffic": None,
            "launches_in_flight": streams, "ms_per_round_in_flight": round(flight_ms, 3),
            "per_launch": {"ms": round(alone_ms, 3), "achieved": round(ex_alone, 1), "frac": round(ex_alone / peak, 4),
                           "cus_held": wgs, "frac_of_the_cus_held": round(ex_alone / peak * 256 / wgs, 4),
                           "note": "the launch ALONE on the chip (HIP events around 6 back-to-back launches on the "
                                   "launch stream): 32 workgroups hold 32 CUs, the other 224 idle by design"},
            "instruction": "v_mfma_f32_32x32x16_f16", "mfma_products_per_fp32_product": 3,
            "algorithmic": {"flops_per_launch": algo, "achieved_in_flight": round(algo * streams / (flight_ms * 1e-3) / 1e12, 2),
                            "unit": "TFLOP/s",
                            "vs_fp32_mfma_peak": round(algo * streams / (flight_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                            "note": "2 T sum(N K) over the projections of the launch (attention / depthwise conv on "
                                    "the fp32 pipes not counted)"},
            "measured": (f"`achieved` / `frac`: {streams} launches at once on {streams} streams (the headline keeps that "
                         f"many encoder stages in flight), {rounds} rounds between a synchronise pair (host clock); "
                         "executed MFMA flops = 3 x algorithmic (two f16 planes per operand, three products) against the "
                         "dense f16 peak"),
            "dtype": DTYPES["split-panel"],
            "note": "fp32 in / fp32 out on 3 f16 MFMA products of two-plane operand splits (a power-of-two scale per row "
                    "of a 512-wide phase and per weight row; cross terms in their own accumulator; blocks whose operands "
                    "leave the planes' range recomputed on the fp32 MFMA inside the launch), LayerNorm folds, rel-pos "
                    "attention (exact fp32 MFMA) and GLU . dwconv . BN . swish inside the same launch"}


# front-end stages (BASELINE configs[1]) and their HBM roofline
# ---------------------------------------------------------------------------------------------
def synth_wav(gen, batch=BATCH):
    """correlated 4-channel noise: one source seen with small per-channel delays + sensor noise"""
    src = 0.1 * torch.randn(batch, SAMPLES + 16, generator=gen)
    wav = torch.stack([src[:, d:d + SAMPLES] for d in (0, 2, 5, 9)], 1)
    return (wav + 0.05 * torch.randn(batch, CH, SAMPLES, generator=gen)).contiguous()


def build_frontend(device, rank, batches):
    from aps_amd.asr.filter.mvdr import MvdrBeamformer
    from aps_amd.transform import EnhTransform
    torch.manual_seed(3)
    mvdr = MvdrBeamformer(BINS, att_dim=512, mask_norm=True)
    enh = EnhTransform(feats="spectrogram-log-cmvn-ipd", frame_len=FRAME_LEN, frame_hop=FRAME_HOP,
                       window="sqrthann", center=False, ipd_index="0,1;0,2;0,3", cos_ipd=True)
    enh.nan_policy = "deferred"  # NaN scan still runs in-kernel every step; host does not stall
    xs, ms, mn = [], [], []
    for b in range(batches):
        g = torch.Generator().manual_seed(1 + 1000 * rank + 17 * b)
        xs.append(0.1 * torch.randn(BATCH, CH, SAMPLES, generator=g))
        masks = torch.sigmoid(torch.randn(BATCH, FRAMES, 2 * BINS, generator=g))
        s, n = [m.contiguous() for m in torch.chunk(masks, 2, -1)]
        ms.append(s)
        mn.append(n)
    cpu = dict(x=xs[0], mask_s=ms[0], mask_n=mn[0],
               att=[p.detach().clone() for p in (mvdr.ref.proj.weight, mvdr.ref.proj.bias,
                                                 mvdr.ref.gvec.weight, mvdr.ref.gvec.bias)])
    dev = dict(x=[t.to(device) for t in xs], mask_s=[t.to(device) for t in ms],
               mask_n=[t.to(device) for t in mn], enh=enh.to(device), mvdr=mvdr.to(device))
    return cpu, dev


class FrontendStages(object):
    """The front-end stages over P resident batches (every batch owns its store / feats / weights /
    beam output, so a stage's input was written a whole rotation earlier).  Stages = the launch groups
    of this build: STFT + features in one launch when EnhTransform fuses them (the default for 2 .. 4
    channels), the MVDR weights (covariance + solve, channel attention, projection), the beamformer,
    and -- when an AsrTransform is given (the joint model) -- |Y| -> mel -> log -> cmvn."""

    def __init__(self, enh, mvdr, wavs, masks_s, masks_n, asr=None):
        from aps_amd.asr.filter import mvdr as M
        from aps_amd.cplx import ComplexTensor
        from aps_amd.spectrogram import packed_view
        self.enh, self.mvdr, self.M, self.packed_view, self.asr = enh, mvdr, M, packed_view, asr
        self.ComplexTensor = ComplexTensor
        self.wavs, self.masks_s, self.masks_n = wavs, masks_s, masks_n
        self.P = len(wavs)
        self.batch = int(wavs[0].shape[0])  # utterances per launch
        self.state = [dict() for _ in wavs]
        fused = enh.fuse_encode_features is not False
        # the joint model forms the ASR features in the beamforming pass (EnhASRBase._enhance_fused) when the
        # transform is the abs-chain; the complex beam output is then not written at all
        self.beam_chain = asr.abs_chain() if (asr is not None and hasattr(asr, "abs_chain") and
                                               not os.environ.get("APS_NO_BEAM_FEATURES")) else None
        tail = ["beamform"] if asr is None else (["beamform_features"] if self.beam_chain is not None
                                                 else ["beamform", "asr_features"])
        self.ORDER = (["stft_features"] if fused else ["stft", "features"]) + ["mvdr_weights"] + tail

    def run_stage(self, name, b):
        st = self.state[b]
        if name == "stft_features":
            packed, _ = self.enh.encode(self.wavs[b], None)
            st["feats"] = self.enh(packed)
            from aps_amd.spectrogram import store_of
            st["store"] = store_of(packed)
        elif name == "stft":
            st["store"] = self.enh.forward_stft.to_store(self.wavs[b])
        elif name == "features":
            st["feats"] = self.enh(self.packed_view(st["store"]))
        elif name == "mvdr_weights":
            st["u"], st["wgt"] = self.mvdr.weights_from_masks(st["store"], self.masks_s[b],
                                                              self.masks_n[b])
        elif name == "beamform":
            st["y"] = self.M.beamform_store(st["store"], st["wgt"])
        elif name == "asr_features":
            y = st["y"]
            st["mel"], _ = self.asr(self.ComplexTensor(y[..., 0], y[..., 1]), None)
        elif name == "beamform_features":
            plan, eps = self.beam_chain
            st["mel"], st["y"] = self.M.beamform_features(st["store"], st["wgt"], plan, eps,
                                                          self.asr.nan_pointer(st["store"].device))

    def step(self, b):
        for name in self.ORDER:
            self.run_stage(name, b)
        return self.state[b]["feats"], self.state[b].get("y")

    def roofline(self, rounds=4):
        """per stage: ALGORITHMIC bytes of one launch group (batch of 32) / its mean duration.
        The P launch groups of a round sit back to back in the queue (the stream is held busy by a
        spin kernel while the host enqueues them) between ONE pair of events, so the mean includes
        the dispatch gaps and excludes host launch latency; the cost of the empty bracket is
        subtracted.  `all_stages` prices the whole stage twice: on the sum of this build's own
        per-group bytes, and -- `survey_8d`, the figure SURVEY.md 8(d) and the judge use -- on the
        survey's minimum-traffic schedule (9 426 368 B / utterance with Y and the log-mel rows)."""
        for b in range(self.P):
            self.step(b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in range(self.P):
            self.step(b)
        torch.cuda.synchronize()
        host_ms = 1e3 * (time.perf_counter() - t0)
        spin = spin_cycles_for(1.5 * host_ms)
        bracket_us = empty_bracket_us(spin)
        out, total_us = {}, 0.0
        for name in self.ORDER:
            samples = []
            for _ in range(rounds):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                hold_stream(spin)
                e0.record()
                for b in range(self.P):
                    self.run_stage(name, b)
                e1.record()
                torch.cuda.synchronize()
                samples.append((1e3 * e0.elapsed_time(e1) - bracket_us) / self.P)
            us = statistics.median(samples)
            algo = ALGO_BYTES[name] * self.batch
            gbs = algo / (us * 1e-6) / 1e9
            total_us += us
            out[name] = {"kernel": STAGE_KERNELS[name], "us_per_launch": round(us, 2),
                         "algo_bytes_per_launch": algo, "achieved": round(gbs, 1),
                         "frac": round(gbs / HBM_PEAK_GBS, 4)}
        algo_all = sum(ALGO_BYTES[k] for k in self.ORDER) * self.batch
        gbs = algo_all / (total_us * 1e-6) / 1e9
        # (8(d): 8 914 424 B without the beam output, + 511 944 when Y is returned: the joint model does not
        # return it and, with the fused P3, does not write it; the front-end workload's output IS Y)
        s8d = SURVEY_8D_BYTES - (0 if self.asr is not None else FRAMES * 80 * 4) - \
            (FRAMES * BINS * 8 if "beamform_features" in self.ORDER else 0)
        gbs8 = s8d * self.batch / (total_us * 1e-6) / 1e9
        out["all_stages"] = {"us_per_batch": round(total_us, 2), "algo_bytes_per_batch": algo_all,
                             "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                             "survey_8d": {"bytes_per_utterance": s8d, "achieved": round(gbs8, 1),
                                           "frac": round(gbs8 / HBM_PEAK_GBS, 4),
                                           "note": "SURVEY.md 8(d) minimum-traffic schedule (P1 + P2 + P3"
                                                   + ("" if "beamform_features" in self.ORDER else " + Y")
                                                   + (", log-mel rows" if self.asr is not None else "") +
                                                   ") / the summed stage times of this build"}}
        out["bound"], out["peak"], out["unit"] = "hbm", HBM_PEAK_GBS, "GB/s"
        out["utterances_per_launch"] = self.batch
        out["measured"] = (f"{rounds} rounds x {self.P} back-to-back launch groups on {self.P} distinct "
                           f"resident batches between one pair of HIP events per round (median), "
                           f"minus the empty bracket ({bracket_us:.1f} us) / {self.P}")
        return out


def frontend_in_flight(fs, s8d_bytes: int, streams: int = 3, steps: int = 96):
    """`stage_roofline.in_flight`: the same front-end launch groups with `streams` batches in flight (one hipGraph per
    resident batch, replayed round-robin on `streams` streams: what the pipelined headline does with its stages) --
    SURVEY 8(d) bytes / wall time between one synchronise pair, against the HBM peak.  An extra: never fails the line."""
    try:
        from aps_amd.replicas import GraphReplicas
        reps = GraphReplicas([lambda b=b: fs.step(b) for b in range(fs.P)], replicas=streams, verify=True, guard_every=0)
        for _ in range(2 * fs.P):
            reps.submit(after_caller=False)
        reps.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            reps.submit(after_caller=False)
        reps.synchronize()
        dt = time.perf_counter() - t0
        reps.close()
        gbs = s8d_bytes * fs.batch * steps / dt / 1e9
        return {"streams": streams, "steps": steps, "us_per_batch": round(1e6 * dt / steps, 2),
                "utt_per_s": round(fs.batch * steps / dt, 1), "achieved": round(gbs, 1),
                "frac": round(gbs / HBM_PEAK_GBS, 4),
                "note": f"SURVEY 8(d) bytes x {fs.batch} utterances per launch group / wall time of {steps} whole-stage "
                        f"hipGraph replays, {streams} batches in flight on {streams} streams (host clock around a "
                        "synchronise pair; rotating resident batches)"}
    except Exception as exc:  # noqa: BLE001
        return {"error": str(exc)[:300]}


def frontend_cpu_baseline(cpu):
    """oracle on the host cores, bounded sample of the same workload"""
    from oracle import aps_oracle as orc
    n = 8
    x, ms, mn = cpu["x"][:n], cpu["mask_s"][:n], cpu["mask_n"][:n]
    out = {}

    def once():
        packed = orc.stft(x, FRAME_LEN, FRAME_HOP, "sqrthann")
        out["feats"] = orc.enh_features(packed, "spectrogram-log-cmvn-ipd", "0,1;0,2;0,3")
        out["yr"], out["yi"], _ = orc.mvdr_forward(ms, packed[..., 0], packed[..., 1], cpu["att"],
                                                   mn)

    base = cpu_baseline_of(once, n, "passes over the first 8")
    return base, out


def run_frontend(args, R: Ranks):
    cpu, dev = build_frontend(R.device, R.rank, args.batches)
    if args.no_fuse_features:
        dev["enh"].fuse_encode_features = False
    stages = FrontendStages(dev["enh"], dev["mvdr"], dev["x"], dev["mask_s"], dev["mask_n"])
    enh = dev["enh"]
    P = args.batches
    with torch.no_grad():
        for i in range(max(args.warmup, 2)):
            stages.step(i % P)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in range(P):
            stages.step(b)
        torch.cuda.synchronize()
        eager_ms = 1e3 * (time.perf_counter() - t0) / P
        stage_roofline = stages.roofline() if R.rank == 0 else None
        enh._nan_guard.flush()
        graph, launch = None, "eager, 1 stream"
        if not args.eager:
            try:
                from aps_amd.replicas import GraphReplicas
                enh.nan_policy = "manual"  # host reads are illegal while capturing
                graph = GraphReplicas([lambda b=b: stages.step(b) for b in range(P)],
                                      replicas=args.replicas)
                launch = (f"hipGraph replay of the whole step, one graph per resident batch ({P}), "
                          f"{args.replicas} in flight on as many streams")
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] graph capture failed ({exc}); timing eager launches",
                      file=sys.stderr)
                graph = None
                enh.nan_policy = "deferred"
                torch.cuda.synchronize()
        count = [0]

        def submit():
            if graph is not None:
                graph.submit(after_caller=False)  # resident inputs
            else:
                stages.step(count[0] % P)
                count[0] += 1

        for _ in range(args.warmup):
            submit()
        regions, units = timed_regions(R, args.steps, args.repeats, submit, BATCH)
        if graph is not None:
            assert enh._nan_guard.count() == 0, "NaN in the features"
            graph.check_outputs(graph.eager_outputs, "after the timed regions")
        else:
            enh._nan_guard.flush()
        feats0, y0 = [t.clone() for t in (graph.outputs[0] if graph is not None
                                          else stages.step(0))]
    seen = R.ranks_seen()
    if R.rank != 0:
        return
    line = base_line(args, R, regions, units, {
        "workload": "BASELINE configs[1]: EnhTransform 4-ch 16 kHz 4 s -> STFT + log-mag/CMVN + "
                    "cos-IPD(3 pairs) + mask-MVDR (cov x2, attention, solve, beamform), masks given; "
                    "the encoder is in the default (joint) workload",
        "batch_per_gpu": BATCH, "global_batch": BATCH * R.world, "resident_batches": P,
        "batches_in_flight": args.replicas if graph is not None else 1,
        "frame": "512/256 sqrthann", "launch": launch,
        "parallelism": f"dp{R.world} (utterance sharding, no collective)"})
    line["ranks_seen"] = seen
    line["eager_ms_per_step"] = round(eager_ms, 4)
    line["algo_gbs_all_stages"] = round(sum(ALGO_BYTES[k] for k in stages.ORDER) * BATCH /
                                        (line["ms_per_step"] * 1e-3) / 1e9, 1)
    dominant = max(stages.ORDER, key=lambda k: stage_roofline[k]["us_per_launch"])
    dom = stage_roofline[dominant]
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get(dominant)
        except Exception:  # noqa: BLE001
            traffic = None
    line["roofline"] = {"kernel": dom["kernel"], "stage": dominant, "bound": "hbm",
                        "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": dom["frac"], "traffic": traffic,
                        "algo_bytes_per_launch": dom["algo_bytes_per_launch"],
                        "kernel_us": dom["us_per_launch"]}
    line["stage_roofline"] = stage_roofline
    if not args.no_cpu_baseline:
        base, ref = frontend_cpu_baseline(cpu)
        n = ref["feats"].shape[0]
        line["parity"] = {"feats": scaled_err(feats0[:n], ref["feats"]),
                          "beam_re": scaled_err(y0[:n, ..., 0], ref["yr"]),
                          "beam_im": scaled_err(y0[:n, ..., 1], ref["yi"]), "n": n,
                          "tol": PARITY_TOL, "vs": "CPU oracle, batch 0"}
        if R.world == 1:
            line["cpu_baseline"] = base
        bad = {k: v for k, v in line["parity"].items() if k in ("feats", "beam_re", "beam_im")
               and not v <= PARITY_TOL}
        if bad:
            print(json.dumps(line))
            raise SystemExit(f"[bench] PARITY FAILURE vs the CPU oracle: {bad}")
    print(json.dumps(line))


# ---------------------------------------------------------------------------------------------
# --workload encoder : BASELINE configs[3], transformer encoder (12 x 512, FF 2048, conv2d 256 x 2,
# 80-mel, 400 frames -> 100) forward, batch 128 per GPU.  MFMA-bound; fp32 MFMA peak 157.3 TFLOP/s.
# ---------------------------------------------------------------------------------------------
ENC_BATCH, ENC_FRAMES, ENC_MELS = 128, 400, 80
ENC_FLOP_PER_UTT = 10.72e9  # torch flop counter on the reference module


def build_encoder(device, rank):
    from aps_amd.asr.transformer import TransformerEncoder
    torch.manual_seed(5)
    enc = TransformerEncoder("xfmr", ENC_MELS, num_layers=12, proj="conv2d",
                             proj_kwargs={"conv_channels": 256, "num_layers": 2}, pose="abs",
                             pose_kwargs={"dropout": 0},
                             arch_kwargs={"att_dim": 512, "nhead": 8, "feedforward_dim": 2048,
                                          "att_dropout": 0, "ffn_dropout": 0,
                                          "pre_norm": False}).eval()
    g = torch.Generator().manual_seed(6 + 1000 * rank)
    x = torch.randn(ENC_BATCH, ENC_FRAMES, ENC_MELS, generator=g)
    lens = torch.tensor([ENC_FRAMES] * ENC_BATCH)
    sd = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    return dict(x=x, lens=lens, sd=sd), dict(x=x.to(device), lens=lens.to(device),
                                             enc=enc.to(device))


def encoder_cpu_baseline(cpu, n):
    """the CPU oracle: reference outputs for the parity check + its rate on the host cores"""
    from oracle import encoder_oracle as eo
    ref = {}

    def once():
        ref["out"], _ = eo.xfmr_abs_encoder(cpu["sd"], cpu["x"][:n], cpu["lens"][:n], 12, 8)

    return ref, cpu_baseline_of(once, n, "forwards")


def run_encoder(args, R: Ranks):
    from aps_amd import nn_ops
    cpu, dev = build_encoder(R.device, R.rank)
    enc, x, lens = dev["enc"], dev["x"], dev["lens"]
    with torch.no_grad():
        for _ in range(max(args.warmup, 2)):
            out0 = enc(x, lens)
        torch.cuda.synchronize()
        probe_steps = min(args.steps, 5)
        nn_ops.GEMM_TIMELINE = timeline = []
        for _ in range(probe_steps):
            enc(x, lens)
        torch.cuda.synchronize()
        nn_ops.GEMM_TIMELINE = None
        regions, units = timed_regions(R, args.steps, args.repeats, lambda: enc(x, lens), ENC_BATCH)
    seen = R.ranks_seen()
    if R.rank != 0:
        return
    line = base_line(args, R, regions, units, {
        "workload": "BASELINE configs[3]: asr transformer encoder (12x512, FF 2048, conv2d 256x2 "
                    "subsampling, 80-mel, 400 frames) forward only",
        "batch_per_gpu": ENC_BATCH, "global_batch": ENC_BATCH * R.world,
        "parallelism": f"dp{R.world} (utterance sharding, forward: no collective)"})
    line["ranks_seen"] = seen
    line["encoder_tflops_end_to_end"] = round(
        ENC_FLOP_PER_UTT * ENC_BATCH / (line["ms_per_step"] * 1e-3) / 1e12, 2)
    line["roofline"] = gemm_roofline(timeline, probe_steps, 0.0, "step, all nn.Linear")
    line["dtype"] = line["roofline"].pop("dtype")
    line["fp32_path_tiles"] = nn_ops.fp16x2_wide_tiles(R.device)
    if not args.no_cpu_baseline:
        n = 8
        ref, base = encoder_cpu_baseline(cpu, n)
        got = out0[0] if isinstance(out0, (tuple, list)) else out0
        line["parity"] = {"enc_out": scaled_err(got[:n], ref["out"]), "n": n, "tol": PARITY_TOL,
                          "vs": "CPU oracle"}
        if R.world == 1:
            line["cpu_baseline"] = base
        if not line["parity"]["enc_out"] <= PARITY_TOL:
            print(json.dumps(line))
            raise SystemExit(f"[bench] PARITY FAILURE vs the CPU oracle: {line['parity']}")
    print(json.dumps(line))


# ---------------------------------------------------------------------------------------------
# --workload joint : BASELINE configs[4], the joint front end (SURVEY.md 8d "Config 5"):
# EnhTransform(spectrogram-log-cmvn-ipd) -> RNNMaskMvdr (1028 -> 512 -> 2 x LSTM 512 -> 514 masks,
# MVDR att 512) -> AsrTransform(abs-mel-log-cmvn, 80 mel) -> conformer (conf/asr/chime4/1a.yaml:
# 12 layers, conv2d 128 x 2, rel pose r = 256, 512 / 8 heads / FF 1024, k = 15) + CTC head,
# 32 utterances of 4 ch x 4 s per GPU (global batch 256 on 8 GPUs).
# ---------------------------------------------------------------------------------------------
JOINT_VOCAB = 5000


def build_joint(device, rank, batches=1, group=1):
    """`batches` resident input batches of `group` x 32 utterances each"""
    from aps_amd.asr.ctc import CtcASR
    from aps_amd.asr.enh_att import EnhASRBase
    from aps_amd.transform import AsrTransform, EnhTransform
    torch.manual_seed(7)
    enh_transform = EnhTransform(feats="spectrogram-log-cmvn-ipd", frame_len=FRAME_LEN,
                                 frame_hop=FRAME_HOP, window="sqrthann", ipd_index="0,1;0,2;0,3",
                                 cos_ipd=True)
    asr_transform = AsrTransform(feats="abs-mel-log-cmvn", frame_len=FRAME_LEN,
                                 frame_hop=FRAME_HOP, window="sqrthann", num_mels=80)
    enc_kwargs = dict(num_layers=12, proj="conv2d",
                      proj_kwargs={"conv_channels": 128, "num_layers": 2}, pose="rel",
                      pose_kwargs={"dropout": 0, "lradius": 256, "rradius": 256},
                      arch_kwargs={"att_dim": 512, "nhead": 8, "feedforward_dim": 1024,
                                   "att_dropout": 0, "ffn_dropout": 0})
    asr = CtcASR(input_size=80, vocab_size=JOINT_VOCAB, ctc=True, ead=True, enc_type="cfmr",
                 enc_kwargs=enc_kwargs)
    enh_kwargs = dict(num_bins=BINS, rnn_inp_proj=512, rnn="lstm", num_layers=2, hidden_size=512,
                      dropout=0.0, bidirectional=False, mvdr_att_dim=512, mask_norm=True)
    net = EnhASRBase(asr, enh_input_size=BINS * 4, enh_transform=enh_transform,
                     asr_transform=asr_transform, enh_type="rnn_mask_mvdr",
                     enh_kwargs=enh_kwargs).eval()
    wavs = []
    for b in range(batches):
        g = torch.Generator().manual_seed(8 + 1000 * rank + 17 * b)
        wavs.append(synth_wav(g, BATCH * group))
    lens = torch.tensor([SAMPLES] * (BATCH * group))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    cpu = dict(wav=wavs[0], lens=lens, sd=sd)
    if device.type == "cpu":
        return cpu, dict(wavs=wavs, lens=lens, net=net)
    return cpu, dict(wavs=[w.to(device) for w in wavs], lens=lens.to(device), net=net.to(device))


def joint_cpu_baseline(cpu, n_parity, n_timed):
    """the CPU oracle of the joint path: reference outputs of the first n_parity utterances of
    batch 0 (the parity check) and, when n_timed > 0, its rate on the host cores"""
    from oracle import joint_oracle as jo

    def forward(n):
        return jo.joint_forward(cpu["sd"], cpu["wav"][:n], cpu["lens"][:n], num_mels=80,
                                rnn_layers=2, enc_layers=12, nhead=8)

    before = torch.get_num_threads()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    try:
        ref = forward(n_parity)
    finally:
        torch.set_num_threads(before)
    base = None
    if n_timed > 0:
        base = cpu_baseline_of(lambda: forward(n_timed), n_timed, "joint forwards")
    return ref, base


def host_input_rate(reps, wavs, units_per_step: int, steps: int, where=None, chunks: int = 1):
    """The PCIe-INCLUSIVE rate, measured (never `value`): every step's waveforms start in page-locked HOST memory
    and are copied into the batch's resident device tensor in front of the batch's first stage, beside the kernels
    of the steps in flight (the double-buffer rule of distributed.PinnedStager: a slot is refilled only behind its
    last reader's event).  The outputs are NOT
    compared afterwards (the inputs are the same values as before, so the bench's later checks still hold)."""
    try:
        P = len(wavs)
        host = [w.detach().cpu().pin_memory() for w in wavs]
        # where the copy is queued: HOST_INPUT_STREAM (the caller's stream; the measurements are at its definition);
        # `chunks` > 1 cuts a batch's copy into that many pieces (4 pieces: 15.4 k against 19.4 k utt/s in one piece --
        # more DMA packets to schedule between the kernels' queue entries, nothing gained)
        where = where or HOST_INPUT_STREAM
        own = torch.cuda.Stream() if where == "own" else None
        torch.cuda.synchronize()
        keep_mid = reps.mid
        if where == "head":
            reps.mid = "worker"   # the head stream carries the copies: the front end's tail goes to the workers (13.3 against 11.7 k)

        def one():
            b = reps.next_index   # the batch the submission below launches: ITS waveforms are refilled
            copy = {"own": own, "null": torch.cuda.current_stream(), "head": reps.lstm_stream,
                    "worker": reps.streams[b % reps.workers]}[where]
            done = reps.done_event(b)
            if done is not None:
                copy.wait_event(done)   # the batch's previous pass has read its waveforms
            with torch.cuda.stream(copy):
                if chunks > 1:
                    for dst, src in zip(wavs[b].chunk(chunks, 0), host[b].chunk(chunks, 0)):
                        dst.copy_(src, non_blocking=True)
                else:
                    wavs[b].copy_(host[b], non_blocking=True)
                index, _ = reps.submit(after_caller=True)   # (its first stage waits for the copy stream's head = this copy)
            assert index == b or getattr(reps, "lookahead", False)

        try:
            for _ in range(P):
                one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                one()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            reps.mid = keep_mid
        mb = wavs[0].numel() * wavs[0].element_size() / 1e6
        return {"what": "the same steps with every batch's waveforms copied host -> device (pinned memory) in front of "
                        "its first stage, beside the steps in flight; the PCIe-inclusive rate, NOT `value`",
                "value": round(units_per_step * steps / dt, 1), "unit": "utt/s",
                "ms_per_step": round(1e3 * dt / steps, 4), "host_to_device_mb_per_step": round(mb, 1),
                "h2d_gb_per_s_needed": round(mb / 1e3 / (dt / steps), 1), "steps": steps, "copy_stream": where}
    except Exception as exc:  # noqa: BLE001  (an extra: never fails the line)
        return {"error": str(exc)[:300]}


def measure_joint(args, R: Ranks, G: int, P: int, steps: int, warmup: int, repeats: int):
    """One measurement of the joint step at a per-GPU batch of G x 32 utterances per launch sequence
    over P resident batches: timed regions = `steps` passes each, every pass a replay of one resident
    batch's captured hipGraph, args.replicas of them in flight on as many streams (--eager keeps plain
    launches).  The dominant kernel (the projections' GEMM) is timed with HIP events on the launch
    stream in instrumented eager passes of the same step right before the timed regions: event
    records cannot sit inside a graph replay."""
    from aps_amd import nn_ops
    cpu, dev = build_joint(R.device, R.rank, P, G)
    net, wavs, lens = dev["net"], dev["wavs"], dev["lens"]
    if args.no_fuse_features:
        net.enh_transform.fuse_encode_features = False
    units_per_step = BATCH * G
    # the NaN scan of check_valid runs inside the feature kernels every step; its counter is read
    # without stalling the stream (eager) / after the replays (graph), never skipped
    net.enh_transform.nan_policy = net.asr_transform.nan_policy = "deferred"
    m = {"G": G, "P": P, "cpu": cpu,
         "inputs_distinct": R.inputs_distinct(float(wavs[0][:, :, :1000].double().abs().sum().item()))}
    m["G"] = G
    # every pass of this function launches what the headline mode launches: with R batches in flight the library
    # sizes its persistent launches for 1 / R of the chip and keeps the four-wave GEMM tiles (nn_ops.lstm_share)
    pipeline = 0 if args.eager else int(getattr(args, "pipeline", 0) or 0)
    in_flight = 1 if args.eager else (args.pipe_share if pipeline else args.replicas)
    nn_ops.push_lstm_share(in_flight)
    if pipeline:  # (stages on `pipeline` worker streams + the LSTM stream: four-wave GEMM tiles, full-chip LSTM launches)
        nn_ops.STREAMS_IN_FLIGHT = pipeline + 1
    with torch.no_grad():
        for i in range(max(warmup, 2)):
            net(wavs[i % P], lens)
        torch.cuda.synchronize()
        # ---- eager passes: host-bound step time, then per-GEMM events (roofline) + stage times
        probe_steps = max(2, min(steps, 8))
        t0 = time.perf_counter()
        for i in range(probe_steps):
            net(wavs[i % P], lens)
        torch.cuda.synchronize()
        eager_ms = 1e3 * (time.perf_counter() - t0) / probe_steps
        # Instrumented passes.  Eager launches are host bound (the GPU idles between kernels), and
        # an event bracket would then time the host's launch gap with the kernel.  So the stream is
        # first held busy by a spin kernel for longer than the host needs to enqueue the step: the
        # launches then sit back to back in the queue, like the nodes of the replayed graph, and a
        # bracket sees the kernel's duration.
        spin = spin_cycles_for(1.5 * eager_ms)
        nn_ops.GEMM_TIMELINE = timeline = []
        for i in range(probe_steps):
            hold_stream(spin)
            net(wavs[i % P], lens)
            torch.cuda.synchronize()
        nn_ops.GEMM_TIMELINE = None
        bracket_us = empty_bracket_us(spin)
        # The same launches WITHOUT an event pair around each: one pass records every two-plane GEMM call
        # of the step (nn_ops.GEMM_RECORD), the launches of a kind are then re-issued back to back -- the
        # same operands, results rewritten in place -- between ONE pair of events.  What comes out is what
        # rocprof's per-kernel durations add up to (kernel + its boundary); the per-launch brackets above
        # carry a second dispatch gap each that the empty-bracket correction does not remove (7 us per
        # launch at 32 utterances, nothing at 128).
        from aps_amd import mega
        nn_ops.GEMM_RECORD = record = []
        mega.RECORD = mega_calls = []
        net(wavs[0], lens)
        torch.cuda.synchronize()
        nn_ops.GEMM_RECORD = None
        mega.RECORD = None
        # (the WHOLE sequence in its own order: the panel launches hand each other's weight images to the LDS
        # prefetch, a chain that re-issuing one kind at a time would break; a kind's share of the total is its
        # share of the bracketed time)
        calls = [c for c, _, _, _ in record]
        reps_ = 6
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for c in calls:   # warm (code, descriptors)
            c()
        hold_stream(spin_cycles_for(0.3))
        e0.record()
        for _ in range(reps_):
            for c in calls:
                c()
        e1.record()
        torch.cuda.synchronize()
        total_ms = e0.elapsed_time(e1) / reps_
        by_kind = {}
        for a_, b_, _, k in timeline:
            by_kind[k] = by_kind.get(k, 0.0) + a_.elapsed_time(b_) - bracket_us * 1e-3
        share_all = max(sum(by_kind.get(k, 0.0) for k in {k for _, _, k, _ in record}), 1e-9)
        replayed = {}
        for kind in sorted({k for _, _, k, _ in record}):
            replayed[kind] = {"launches": sum(1 for _, _, k, _ in record if k == kind),
                              "ms_per_step": total_ms * by_kind.get(kind, 0.0) / share_all,
                              "flops": sum(f for _, f, k, _ in record if k == kind)}
        replayed["_all"] = {"launches": len(calls), "ms_per_step": total_ms}
        del record
        # the conformer stack as one launch per batch (aps_amd.mega): the dominant kernel when several streams launch
        m["mega_roofline"] = None
        if mega_calls and R.rank == 0:
            m["mega_roofline"] = mega_roofline(mega_calls[0], max(pipeline, args.replicas, 1))
        del mega_calls
        net.enh_transform._nan_guard.flush()
        net.asr_transform._nan_guard.flush()
        stage_roofline = None
        if R.rank == 0:
            # the HBM-bound front-end stages of THIS model on the batches the timed steps run on (G x 32
            # utterances per launch; > 256 MB of waveforms in rotation), masks = what the model's mask
            # estimator emits for each batch
            masks = [torch.chunk(net.enh_net.mask_net(net.enh_transform(
                net.enh_transform.encode(w, lens)[0]), None)[0], 2, dim=-1) for w in wavs]
            fs = FrontendStages(net.enh_transform, net.enh_net.mvdr_net, wavs,
                                [k[0].contiguous() for k in masks],
                                [k[1].contiguous() for k in masks], asr=net.asr_transform)
            stage_roofline = fs.roofline()
            stage_roofline["in_flight"] = frontend_in_flight(
                fs, stage_roofline["all_stages"]["survey_8d"]["bytes_per_utterance"])
            del fs, masks
            net.enh_transform._nan_guard.flush()
        # ---- one hipGraph per resident batch (torch's capture API is only the recorder: every node
        # is one of our launches): ~160 host launches per step -> 1.  --replicas R: R batches in
        # flight on R streams (aps_amd/replicas.py): the latency-bound LSTM mask estimator of one
        # hides behind the GEMM-bound conformer of the other.
        reps, launch, single_ms = None, "eager, one stream", None
        if not args.eager:
            try:
                from aps_amd.replicas import GraphReplicas
                net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
                if args.replicas > 1 or pipeline:
                    # what the library default gives (GraphReplicas(replicas=1): nothing else launching, so the
                    # one-tile-per-CU projections take the K-group form): its own capture of two of the batches
                    nn_ops.pop_lstm_share(in_flight)
                    saved_streams, nn_ops.STREAMS_IN_FLIGHT = nn_ops.STREAMS_IN_FLIGHT, 1
                    try:
                        one = GraphReplicas([lambda b=b: net(wavs[b], lens) for b in range(min(P, 4))], replicas=1)
                        for _ in range(2 * len(one)):
                            one.submit(after_caller=False)
                        one.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(probe_steps):
                            one.submit(after_caller=False)
                        one.synchronize()
                        m["single_default_ms"] = 1e3 * (time.perf_counter() - t0) / probe_steps
                        one.close()
                        del one
                        if pipeline:
                            # rounds 2-4's headline mode, for continuity: two WHOLE steps in flight on two streams
                            two = GraphReplicas([lambda b=b: net(wavs[b], lens) for b in range(min(P, 4))], replicas=2)
                            for _ in range(2 * len(two)):
                                two.submit(after_caller=False)
                            two.synchronize()
                            t0 = time.perf_counter()
                            for _ in range(4 * probe_steps):
                                two.submit(after_caller=False)
                            two.synchronize()
                            m["whole_step_ms"] = 1e3 * (time.perf_counter() - t0) / (4 * probe_steps)
                            two.close()
                            del two
                    finally:
                        nn_ops.push_lstm_share(in_flight)
                        nn_ops.STREAMS_IN_FLIGHT = saved_streams
                if pipeline:
                    from aps_amd.replicas import PipelinedReplicas
                    try:
                        reps = PipelinedReplicas([lambda b=b: net(wavs[b], lens) for b in range(P)], workers=pipeline,
                                                 lstm_share=in_flight, front=args.pipe_front, mid=args.pipe_mid,
                                                 lookahead=args.pipe_lookahead, heads=args.pipe_heads)
                    except Exception as exc:  # noqa: BLE001  (say so and measure rounds 2-4's mode instead)
                        print(f"[bench] the staged capture failed ({exc}); whole-step graphs on {args.replicas} streams",
                              file=sys.stderr)
                        torch.cuda.synchronize()
                        nn_ops.pop_lstm_share(in_flight)
                        pipeline, in_flight, nn_ops.STREAMS_IN_FLIGHT = 0, args.replicas, 1
                        nn_ops.push_lstm_share(in_flight)
                if pipeline:
                    from aps_amd import mega as _mega
                    launch = (f"the step cut at the mask estimator's persistent LSTM launch into {reps.stages} hipGraphs per "
                              f"resident batch ({P}): the LSTM launches of all batches one after the other on the head "
                              f"stream"
                              + (" with the stage in front of them" if args.pipe_front == "head" else "")
                              + (" and the front end's tail behind them" if reps.stages == 4 and args.pipe_mid == "head" else "")
                              + f", the other stages round-robin on {pipeline} worker streams"
                              + (f", a batch's front launched {pipeline} submissions ahead of its back" if args.pipe_lookahead else "")
                              + " (aps_amd.replicas.PipelinedReplicas)"
                              + ("; the 12 conformer layers of a batch are ONE launch, a workgroup per utterance "
                                 "(aps_conformer_stack)" if _mega.wanted() else ""))
                else:
                    reps = GraphReplicas([lambda b=b: net(wavs[b], lens) for b in range(P)],
                                         replicas=args.replicas)
                    # one stream alone, back to back: the step time without a second batch in flight
                    t0 = time.perf_counter()
                    for i in range(probe_steps):
                        with torch.cuda.stream(reps.streams[0]):
                            reps.graphs[i % P].replay()
                    torch.cuda.synchronize()
                    single_ms = 1e3 * (time.perf_counter() - t0) / probe_steps
                    launch = (f"hipGraph replay of the whole step, one graph per resident batch ({P}), "
                              f"round-robin on {args.replicas} stream(s) = batches in flight")
            except Exception as exc:  # noqa: BLE001  (capture unsupported: stay eager, say so)
                print(f"[bench] graph capture failed ({exc}); timing eager launches",
                      file=sys.stderr)
                reps = None
                torch.cuda.synchronize()
                net.enh_transform.nan_policy = net.asr_transform.nan_policy = "deferred"
        count = [0]

        def submit():
            if reps is not None and pipeline:
                reps.submit(after_caller=False)  # resident inputs
            elif reps is not None:
                reps.submit(after_caller=False)  # resident inputs
            else:
                net(wavs[count[0] % P], lens)
                count[0] += 1

        for _ in range(warmup):
            submit()
        regions, units = timed_regions(R, steps, repeats, submit, units_per_step,
                                       flush=getattr(reps, "flush", None))
        if pipeline and steps < 100 and G == 1:
            # A region of K steps holds the pipeline's fill and drain (the first front end before the first LSTM
            # launch, the last batch's encoder stage behind the last one: ~5.7 ms per region whatever K): at the
            # driver's K = 20 that is 15 % of the region.  The steady-state step time next to it, measured the same way
            # on regions of 100 steps (never `value`).
            long_regions, long_units = timed_regions(R, 100, 3, submit, units_per_step, flush=getattr(reps, "flush", None))
            med = statistics.median(long_regions)
            m["steady_state"] = {"steps": 100, "repeats": 3, "value": round(long_units / med, 1), "unit": "utt/s",
                                 "ms_per_step": round(1e3 * med / 100, 4),
                                 "fill_and_drain_ms_per_region": round(
                                     1e3 * (statistics.median(regions) - steps * med / 100), 3),
                                 "note": f"`value` times regions of exactly --steps = {steps} steps between synchronise pairs; "
                                         "each such region pays the staged pipeline's fill and drain once; this is the same "
                                         "measurement on regions of 100 steps"}
        if reps is not None:
            reps.synchronize()
            reps.check_outputs(reps.eager_outputs, "after the timed regions")
            out0 = [t.clone() for t in reps.outputs[0][:2]]
            m["replay_checks"] = getattr(reps, "checks_run", None)
            if pipeline:
                # latency of a batch in the headline mode, steady state: GPU time from the start of its first stage to
                # the end of its last one with the other batches in flight (HIP events; 2 rounds over the batches)
                for _ in range(P):
                    reps.submit(after_caller=False)
                for _ in range(2 * P):
                    reps.submit(after_caller=False, timed=True)
                reps.synchronize()
                lat = [reps.latency_ms(b) for b in range(P)]
                m["latency_ms"] = {"mean": round(statistics.mean(lat), 3), "max": round(max(lat), 3)}
                # ... and how long every stage takes UNDER LOAD (an event pair around every stage of 3 rounds)
                reps.stage_log = log = []
                for _ in range(3 * P):
                    reps.submit(after_caller=False)
                reps.synchronize()
                reps.stage_log = None
                by = {}
                for kind, e0, e1 in log[P * reps.stages:]:
                    by.setdefault(kind, []).append(e0.elapsed_time(e1))
                names = {"a": "A_stft_features_inproj", "l": "L_lstm_stack", "m": "M_masks_mvdr_beamform", "b": "B_encoder"}
                m["stage_ms_under_load"] = {names.get(k, k): round(statistics.mean(v), 3) for k, v in by.items()}
            if pipeline and G == 1 and R.world == 1 and not args.no_host_input:
                m["host_input"] = host_input_rate(reps, wavs, units_per_step, min(steps, 60),
                                                  args.host_input_stream, args.host_input_chunks)
        else:
            out0 = [t.clone() for t in net(wavs[0], lens)[:2]]
        nans = net.enh_transform._nan_guard.count() + net.asr_transform._nan_guard.count()
        assert nans == 0, f"{nans} NaN rows in the features"
        m["timeouts"] = nn_ops.lstm_timeouts(R.device)
        m["wide_tiles"] = nn_ops.fp16x2_wide_tiles(R.device)
    nn_ops.pop_lstm_share(in_flight)
    nn_ops.STREAMS_IN_FLIGHT = 1
    if reps is not None and pipeline:
        reps.close()
    m.update(regions=regions, units=units, eager_ms=eager_ms, single_ms=single_ms, launch=launch,
             in_flight=(pipeline if pipeline else args.replicas) if reps is not None else 1, out0=out0,
             stage_roofline=stage_roofline, steps=steps,
             roofline=gemm_roofline(timeline, probe_steps, bracket_us,
                                    "launch sequence: mask-net, conformer and CTC projections", replayed)
             if R.rank == 0 else None)
    if m["roofline"] is not None:
        # HBM-side bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE x 2 +
        # WRITE_SIZE, profiles/pmc_traffic.json; taken at 32 utterances per launch, so only quoted there)
        kname = m["roofline"]["kernel"].split(" ")[0].replace("_kernel", "")
        if G == 1 and kname in ("gemm_panel", "gemm_kgroup"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                m["roofline"]["traffic"] = pmc.get(kname)
                m["roofline"]["traffic_note"] = pmc.get(f"_{kname}_line_note", (
                    "bytes per aps_linear_panel launch at the L2s' fabric side (Infinity-Cache hits included), "
                    "rocprofv3 --pmc, profiles/pmc_traffic.json"))
            except Exception:  # noqa: BLE001
                pass
        m["roofline"]["measured"] = (
            "kernel_ms_per_step: see `timing`; bracketed_ms_per_step: HIP events around every launch in "
            f"{probe_steps} queued-ahead eager passes of the same step over the rotating batches "
            "(bracket_corrected: minus the cost of an empty bracket measured the same way)")
    del reps, net, wavs, dev
    torch.cuda.empty_cache()
    return m


def run_joint(args, R: Ranks):
    """default workload: the joint step at BASELINE's own per-GPU share -- 32 utterances per launch
    sequence (configs[4]: batch 256 over 8 GPUs; `--global-batch 256` on 8 ranks is that literally) -- is
    the headline `value`; the SAME line carries `merged_batch`, the measurement with --merged-group (4)
    x 32 utterances fused into one launch sequence (utterances are independent: their results do not
    depend on the batch they ride in), unless --merged-group is 0 or --group is not 1."""
    G = args.group
    m = measure_joint(args, R, G, args.batches, args.steps, args.warmup, args.repeats)
    merged = None
    if G == 1 and args.merged_group > 1:
        Gm = args.merged_group
        Pm = max(args.replicas, -(-max(3, 12 // Gm) // args.replicas) * args.replicas)
        # the merged batch stays on whole-step graphs (128 utterances per launch fill the chip from one stream;
        # APS_BENCH_MERGED_PIPELINE=W tries the three-graph pipeline there)
        keep, args.pipeline = args.pipeline, 0   # (the merged batch gains nothing from the staged pipeline: whole-step graphs)
        try:
            merged = measure_joint(args, R, Gm, Pm, max(10, args.steps // 3), max(3, args.warmup // 2), 3)
        finally:
            args.pipeline = keep
    seen = R.ranks_seen()
    if R.rank != 0:
        return
    line = base_line(args, R, m["regions"], m["units"], {
        "workload": "BASELINE configs[4]: joint front end, 4-ch 4 s -> STFT + IPD features -> LSTM "
                    "masks -> MVDR -> 80-mel log/cmvn -> 12-layer conformer (chime4/1a geometry) + "
                    "CTC head, forward only",
        "batch_per_gpu": BATCH * G, "global_batch": BATCH * G * R.world,
        "batch_note": ("BASELINE's batch 256 / 8 GPUs = 32 utterances per GPU and launch sequence" if G == 1 else
                       f"{G} x the 32 utterances per GPU of BASELINE's batch 256 / 8 GPUs, fused into one "
                       "launch sequence (utterances are independent)"),
        "resident_batches": m["P"],
        "batches_in_flight": m["in_flight"],
        "hardware_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
        "frames": FRAMES, "encoder_frames": ((FRAMES - 1) // 2 // 2) + 1,
        "parallelism": f"dp{R.world} (utterance sharding, forward: no collective)"})
    line["ms_per_32_utterances"] = round(line["ms_per_step"] / G, 4)
    line["launch"] = m["launch"]
    line["ranks_seen"] = seen
    line["inputs_distinct_across_ranks"] = m["inputs_distinct"]
    line["lstm_handoff_timeouts"] = m["timeouts"]
    line["fp32_path_tiles"] = m["wide_tiles"]
    line["replay_checks"] = m.get("replay_checks")
    line["eager_ms_per_step"] = round(m["eager_ms"], 3)
    line["single_stream_ms_per_step"] = None if m["single_ms"] is None else round(m["single_ms"], 3)
    if m["single_ms"] is not None:
        # the headline's graphs replayed on ONE stream (captured for two in flight: four-wave GEMM tiles, half-chip
        # LSTM launches) and, next to it, what a caller of the library default gets (GraphReplicas(replicas=1):
        # its own capture, one batch in flight)
        line["single_stream_value"] = round(BATCH * G * R.world / (m["single_ms"] * 1e-3), 1)
        if m.get("single_default_ms"):
            line["single_stream_default_ms_per_step"] = round(m["single_default_ms"], 3)
            line["single_stream_default_value"] = round(BATCH * G * R.world / (m["single_default_ms"] * 1e-3), 1)
    elif m.get("single_default_ms"):
        # pipeline mode: one stream = the library default (GraphReplicas(replicas=1), its own capture)
        line["single_stream_ms_per_step"] = round(m["single_default_ms"], 3)
        line["single_stream_value"] = round(BATCH * G * R.world / (m["single_default_ms"] * 1e-3), 1)
    if m.get("latency_ms"):
        line["latency_ms_per_batch"] = {
            "headline": m["latency_ms"]["mean"], "headline_max": m["latency_ms"]["max"],
            "one_stream": line.get("single_stream_ms_per_step"),
            "note": "headline: GPU time from the start of a batch's first stage to the end of its last, the other batches "
                    "of the pipeline in flight (HIP events, mean / max over the resident batches); one_stream: the step "
                    "time of one batch in flight"}
    if m.get("stage_ms_under_load"):
        line["stage_ms_under_load"] = m["stage_ms_under_load"]
    if m.get("host_input"):
        line["host_input"] = m["host_input"]
    if m.get("steady_state"):
        line["steady_state"] = m["steady_state"]
    if m.get("whole_step_ms"):
        line["whole_step_replicas"] = {
            "what": "rounds 2-4's headline mode: two WHOLE steps in flight on two streams (GraphReplicas(replicas=2)), "
                    "measured in this run on 4 of the resident batches",
            "ms_per_step": round(m["whole_step_ms"], 3),
            "value": round(BATCH * G * R.world / (m["whole_step_ms"] * 1e-3), 1)}
    line["roofline"] = m["roofline"]
    line["dtype"] = line["roofline"].pop("dtype")
    if m.get("mega_roofline"):
        # the encoder stack runs as ONE launch per batch: that kernel is the dominant one; the projections still
        # launched one by one (mask estimator, conv2d subsampling's linear, CTC head) stay next to it
        per_launch = line["roofline"]
        line["roofline"] = m["mega_roofline"]
        line["roofline"].pop("dtype", None)
        if G == 1:
            # bytes at the L2s' fabric side per launch from the committed PMC passes (2 x FETCH_SIZE + WRITE_SIZE,
            # profiles/pmc_traffic.json; taken at 32 utterances per launch)
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
                line["roofline"]["traffic"] = pmc.get("conformer_stack")
                line["roofline"]["traffic_note"] = pmc.get("_conformer_stack_line_note")
            except Exception:  # noqa: BLE001
                pass
        line["roofline"]["per_launch_projections"] = per_launch
    line["stage_roofline"] = m["stage_roofline"]
    if merged is not None:
        med = statistics.median(merged["regions"])
        rf = merged["roofline"]
        line["merged_batch"] = {
            "what": f"the same model and step with {merged['G']} x 32 utterances fused into one launch sequence "
                    "per GPU (a larger per-GPU batch than BASELINE's 256 / 8: NOT the headline), measured "
                    "in this run after it",
            "batch_per_gpu": BATCH * merged["G"], "value": round(merged["units"] / med, 1), "unit": "utt/s",
            "ms_per_step": round(1e3 * med / merged["steps"], 4),
            "ms_per_step_regions": region_stats(merged["regions"], merged["steps"]),
            "single_stream_ms_per_step": None if merged["single_ms"] is None else round(merged["single_ms"], 3),
            "batches_in_flight": merged["in_flight"], "resident_batches": merged["P"],
            "roofline": {k: rf[k] for k in ("kernel", "achieved", "peak", "unit", "frac", "algorithmic",
                                            "kernel_ms_per_step", "kernel_us_per_launch", "timing",
                                            "bracket_corrected_ms_per_step", "other_gemm_kernels", "all_two_plane_gemms")
                         if k in rf},
            "stage_roofline": merged["stage_roofline"]}
    if not args.no_cpu_baseline:
        n = 4
        ref, base = joint_cpu_baseline(m["cpu"], n, 16 if R.world == 1 else 0)
        out0 = m["out0"]
        line["parity"] = {"enc_out": scaled_err(out0[0][:n], ref["enc_out"]),
                          "enc_ctc": scaled_err(out0[1][:n], ref["enc_ctc"]), "n": n,
                          "tol": PARITY_TOL,
                          "vs": "CPU oracle on the first utterances of batch 0, full 12-layer model"}
        if merged is not None:
            om = merged["out0"]  # batch 0 of the merged run: its own seeds, its own reference
            refm, _ = joint_cpu_baseline(merged["cpu"], 2, 0)
            line["merged_batch"]["parity"] = {"enc_out": scaled_err(om[0][:2], refm["enc_out"]),
                                              "enc_ctc": scaled_err(om[1][:2], refm["enc_ctc"]), "n": 2}
        if base is not None:
            line["cpu_baseline"] = base
        bad = {k: v for k, v in line["parity"].items() if k.startswith("enc_")
               and not v <= PARITY_TOL}
        if merged is not None:
            bad.update({"merged_batch." + k: v for k, v in line["merged_batch"]["parity"].items()
                        if k.startswith("enc_") and not v <= PARITY_TOL})
        if bad:
            print(json.dumps(line))
            raise SystemExit(f"[bench] PARITY FAILURE vs the CPU oracle: {bad}")
    print(json.dumps(line))


# ---------------------------------------------------------------------------------------------
# --workload dccrn : BASELINE configs[2], DCCRN mask estimator + separation on 2-speaker 8 kHz
# mixtures (4 s = 32000 samples, 512/256 STFT -> 124 frames), batch 64 per GPU, time-domain output
# (STFT -> 7 complex conv blocks -> complex LSTM 2 x 512 -> 7 complex deconv blocks -> complex ratio
# masks -> masking -> iSTFT x 2 speakers).  8.54 GFLOP / utterance.
# ---------------------------------------------------------------------------------------------
DCCRN_BATCH, DCCRN_SAMPLES = 64, 32000
DCCRN_FLOP_PER_UTT = 8.54e9


def build_dccrn(device, rank):
    from aps_amd.sse.bss.dccrn import DCCRN
    from aps_amd.transform import EnhTransform
    torch.manual_seed(9)
    enh = EnhTransform(feats="spectrogram-log-cmvn", frame_len=512, frame_hop=256,
                       window="sqrthann")
    net = DCCRN(enh_transform=enh, training_mode="time").eval()
    g = torch.Generator().manual_seed(10 + 1000 * rank)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(0.05 * torch.randn(m.num_features, generator=g))
            m.running_var.copy_(0.8 + 0.4 * torch.rand(m.num_features, generator=g))
    mix = 0.3 * torch.randn(DCCRN_BATCH, DCCRN_SAMPLES, generator=g)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    return dict(mix=mix, sd=sd), dict(mix=mix.to(device), net=net.to(device))


def dccrn_cpu_baseline(cpu, n):
    """the CPU oracle: reference outputs for the parity check + its rate on the host cores"""
    from oracle import dccrn_oracle as do
    ref = {}
    cfg = dict(K="3,3;3,3;3,3;3,3;3,3;3,3;3,3", S="2,1;2,1;2,1;2,1;2,1;2,1;2,1",
               P="1,1,1,1,1,1,1", O="0,0,0,0,0,0,0")

    def once():
        ref["out"] = do.dccrn_forward(cpu["sd"], cpu["mix"][:n], **cfg)

    return ref, cpu_baseline_of(once, n, "DCCRN forwards")


def run_dccrn(args, R: Ranks):
    from aps_amd import nn_ops
    cpu, dev = build_dccrn(R.device, R.rank)
    net, mix = dev["net"], dev["mix"]
    with torch.no_grad():
        for _ in range(max(args.warmup, 2)):
            net(mix)
        torch.cuda.synchronize()
        probe_steps = max(1, min(args.steps, 5))
        nn_ops.CONV_TIMELINE = timeline = []
        t0 = time.perf_counter()
        for _ in range(probe_steps):
            net(mix)
        torch.cuda.synchronize()
        eager_ms = 1e3 * (time.perf_counter() - t0) / probe_steps
        nn_ops.CONV_TIMELINE = None
        reps, launch = None, "eager, one stream"
        if not args.eager:
            try:
                from aps_amd.replicas import GraphReplicas
                reps = GraphReplicas(lambda: net(mix), replicas=args.replicas)
                launch = ("hipGraph replay of the whole step" if args.replicas == 1 else
                          f"{args.replicas} hipGraph replicas of the whole step, round-robin on "
                          f"{args.replicas} streams ({args.replicas} batches in flight)")
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] graph capture failed ({exc}); timing eager launches",
                      file=sys.stderr)
                reps = None
                torch.cuda.synchronize()

        def submit():
            if reps is not None:
                reps.submit(after_caller=False)  # static inputs
            else:
                net(mix)

        regions, units = timed_regions(R, args.steps, args.repeats, submit, DCCRN_BATCH)
        if reps is not None:
            reps.synchronize()
            reps.check_outputs(reps.eager_outputs, "after the timed regions")
        out0 = net(mix)
    seen = R.ranks_seen()
    if R.rank != 0:
        return
    conv_ms = sum(t[0].elapsed_time(t[1]) for t in timeline) / probe_steps
    conv_flop = sum(t[2] for t in timeline) / probe_steps
    launches = len(timeline) // probe_steps
    achieved = conv_flop / (conv_ms * 1e-3) / 1e12
    line = base_line(args, R, regions, units, {
        "workload": "BASELINE configs[2]: sse DCCRN forward on 2-spk 8 kHz 4 s mixtures (STFT -> "
                    "mask estimator -> masking -> iSTFT), NOT the headline metric's utterance type",
        "batch_per_gpu": DCCRN_BATCH, "global_batch": DCCRN_BATCH * R.world,
        "parallelism": f"dp{R.world} (utterance sharding, forward: no collective)"})
    line["launch"] = launch
    line["ranks_seen"] = seen
    line["eager_ms_per_step"] = round(eager_ms, 3)
    line["model_tflops_end_to_end"] = round(
        DCCRN_FLOP_PER_UTT * DCCRN_BATCH / (line["ms_per_step"] * 1e-3) / 1e12, 2)
    from aps_amd import nn_ops
    conv16 = nn_ops.CONV_FP16X2 is not False  # (the DCCRN blocks ask for the fp16 two-plane form)
    products = 3 if conv16 else SPLIT_PRODUCTS
    pipe = products * achieved
    line["roofline"] = {
        "kernel": f"{'conv_fp16x2_kernel' if conv16 else 'conv_split_kernel'} (+ conv_smallk_kernel for the 2-channel first layer, "
                  f"conv_fewout_kernel for the 4-channel mask layer; {launches} launches / step: the "
                  "complex conv / deconv blocks)",
        "bound": "mfma", "achieved": round(pipe, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": round(pipe / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None,
        "instruction": "v_mfma_f32_32x32x16_f16" if conv16 else "v_mfma_f32_32x32x16_bf16",
        "mfma_products_per_fp32_product": products,
        "algorithmic": {"achieved": round(achieved, 2), "unit": "TFLOP/s",
                        "vs_fp32_mfma_peak": round(achieved / MFMA_F32_PEAK_TFLOPS, 4)},
        "algo_flops_per_step": conv_flop, "kernel_ms_per_step": round(conv_ms, 4),
        "note": "executed MFMA flops against the peak of the 16-bit matrix pipe (an upper bound: as if "
                "all 14 layers ran the split form; 12 do): " +
                ("3 fp16 MFMA products of two-plane splits of operands scaled per pixel / weight row by a "
                 "power of two, tiles outside the planes' range recomputed in fp32 (the per-pixel exponent "
                 "pass is inside the brackets)" if conv16 else
                 "6 bf16 MFMA products of exact operand splits") +
                "; `algorithmic` = fp32 flops of the useful taps",
        "measured": f"HIP events around every launch, {probe_steps} eager passes"}
    line["dtype"] = DTYPES["split-fp16" if conv16 else "split-bf16"]
    line["fp32_path_tiles"] = nn_ops.fp16x2_wide_tiles(R.device)
    if not args.no_cpu_baseline:
        n = 4
        ref, base = dccrn_cpu_baseline(cpu, n)
        want = ref["out"]
        errs = {f"spk{i}": scaled_err(out0[i][:n], want[i]) for i in range(len(want))}
        line["parity"] = dict(errs, n=n, tol=PARITY_TOL, vs="CPU oracle")
        if R.world == 1:
            line["cpu_baseline"] = base
        if any(not v <= PARITY_TOL for v in errs.values()):
            print(json.dumps(line))
            raise SystemExit(f"[bench] PARITY FAILURE vs the CPU oracle: {errs}")
    print(json.dumps(line))


# ---------------------------------------------------------------------------------------------
# --workload train : one optimiser step of the joint model per step (NOT the headline metric): what
# aps/trainer/ddp.py:124-200 does per batch -- forward in train() mode (BatchNorm on batch
# statistics), torch CTC loss on the CTC head, backward through the HIP autograd functions
# (aps_amd/grad_ops.py), DistributedDataParallel gradient all-reduce over RCCL when N > 1, SGD step.
# ---------------------------------------------------------------------------------------------
def run_train(args, R: Ranks):
    import torch.nn.functional as F
    cpu, dev = build_joint(R.device, R.rank, batches=min(args.batches, 4), group=1)
    net, wavs, lens = dev["net"].train(), dev["wavs"], dev["lens"]
    net.enh_transform.nan_policy = net.asr_transform.nan_policy = "deferred"
    model = net
    if R.world > 1:
        from torch.nn.parallel import DistributedDataParallel as DDP
        model = DDP(net, device_ids=[R.device.index], **R.D.ddp_kwargs())
    g = torch.Generator().manual_seed(11 + R.rank)
    tgt = torch.randint(1, JOINT_VOCAB, (BATCH, 12), generator=g).to(R.device)
    tgt_len = torch.full((BATCH,), 12, dtype=torch.int64, device=R.device)
    opt = torch.optim.SGD(net.parameters(), lr=1e-3)
    P = len(wavs)
    count, losses = [0], []

    def step():
        opt.zero_grad(set_to_none=True)
        _, enc_ctc, enc_len = model(wavs[count[0] % P], lens)
        logp = F.log_softmax(enc_ctc, -1).transpose(0, 1)
        loss = F.ctc_loss(logp, tgt, enc_len, tgt_len, blank=0, reduction="mean", zero_infinity=True)
        loss.backward()
        opt.step()
        losses.append(loss.detach())
        count[0] += 1

    for _ in range(max(args.warmup, 1)):
        step()
    regions, units = timed_regions(R, args.steps, args.repeats, step, BATCH)
    seen = R.ranks_seen()
    if R.rank != 0:
        return
    line = base_line(args, R, regions, units, {
        "workload": "training step of the BASELINE configs[4] model (forward in train() mode + torch "
                    "CTC loss + HIP backward + SGD; DDP gradient all-reduce over RCCL when N > 1) -- "
                    "NOT the headline forward metric",
        "batch_per_gpu": BATCH, "global_batch": BATCH * R.world,
        "parallelism": f"ddp{R.world} (gradient all-reduce, {sum(p.numel() for p in net.parameters()) * 4 / 1e6:.0f} MB fp32)"})
    line["metric"] = "utterances/sec, TRAINING step (fwd + bwd + SGD) of the joint model"
    line["ranks_seen"] = seen
    line["loss_first_last"] = [round(float(losses[0]), 4), round(float(losses[-1]), 4)]
    print(json.dumps(line))


def run_selftest_launch(args, R: Ranks):
    """exercise the launch path only (self-launch, rendezvous, rank -> device binding, barrier,
    max / sum reductions): runs on CPU with gloo when there is no GPU"""
    work = torch.ones(4, device=R.device)
    regions, units = timed_regions(R, args.steps, args.repeats, lambda: work.add_(1.0), BATCH)
    seen = R.ranks_seen()
    if R.rank != 0:
        return
    line = base_line(args, R, regions, units, {"workload": "launch self-test (no kernels)",
                                               "backend": "gloo" if R.cpu_only else "nccl"})
    line["ranks_seen"] = seen
    print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of --steps steps each; the median region is reported")
    ap.add_argument("--batches", type=int, default=None,
                    help="distinct resident input batches the steps rotate over (joint / frontend; "
                         "default: 12 batches of 32, 4 of 128)")
    ap.add_argument("--group", type=int, default=1,
                    help="joint: per-GPU batch = group x 32 utterances in one launch sequence; 1 = BASELINE "
                         "configs[4]'s 32 per GPU (the headline)")
    ap.add_argument("--merged-group", type=int, default=4,
                    help="joint, with --group 1: also measure this many x 32 utterances fused into one "
                         "launch sequence (`merged_batch` in the line); 0 skips it")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="joint: utterances per launch sequence over ALL ranks (BASELINE configs[4] is 256 "
                         "on 8 GPUs); sets --group = global batch / (32 x ranks)")
    ap.add_argument("--no-cpu-baseline", action="store_true",
                    help="skip the CPU oracle legs (cpu_baseline AND the parity check)")
    ap.add_argument("--workload", default="joint",
                    choices=["joint", "frontend", "encoder", "dccrn", "train"],
                    help="joint = BASELINE configs[4], STFT -> MVDR -> encoder forward, the "
                         "configuration the metric is quoted on (default); frontend = configs[1] "
                         "(STFT + features + MVDR with given masks); encoder = configs[3]; "
                         "dccrn = configs[2]")
    ap.add_argument("--no-fuse-features", action="store_true",
                    help="joint / frontend: STFT and the enhancement features as two launches (A/B against "
                         "the fused STFT + features launch, the default for 2 .. 4 channels)")
    ap.add_argument("--eager", action="store_true",
                    help="time plain launches instead of the captured hipGraph")
    ap.add_argument("--replicas", type=int, default=None,
                    help="joint / frontend / dccrn workloads: batches in flight per GPU = streams "
                         "the captured graphs are replayed on (default 2 for joint, 3 for frontend, "
                         "1 for dccrn)")
    ap.add_argument("--pipeline", type=int, default=None,
                    help="joint workload: W > 0 = the step cut at the persistent LSTM launch, the LSTM launches of all "
                         "batches on one stream, the other stages on W worker streams (aps_amd.replicas.PipelinedReplicas); "
                         "0 = whole-step graphs on --replicas streams")
    ap.add_argument("--pipe-share", type=int, default=2,
                    help="--pipeline: what the persistent LSTM launch is sized for (1 / share of the chip); experiments")
    ap.add_argument("--pipe-front", default="worker", choices=["head", "worker"],
                    help="--pipeline: the stream of the stage in front of the LSTM launch; experiments")
    ap.add_argument("--pipe-mid", default="worker", choices=["head", "worker"],
                    help="--pipeline: the stream of the front end's tail behind the LSTM launch; experiments")
    ap.add_argument("--pipe-heads", type=int, default=1,
                    help="--pipeline: streams the persistent LSTM launches alternate over (experiments; needs "
                         "--pipe-share >= 2 x heads)")
    ap.add_argument("--pipe-lookahead", type=int, default=1,
                    help="--pipeline: 1 = a batch's front (stage A + LSTM launch) is launched `workers` submissions "
                         "ahead of its back (PipelinedReplicas(lookahead=True))")
    ap.add_argument("--no-host-input", action="store_true", help="skip the host-fed (PCIe-inclusive) extra")
    ap.add_argument("--host-input-stream", default=None, choices=("head", "worker", "own", "null"),
                    help="where the host-fed extra queues its copies (default: HOST_INPUT_STREAM)")
    ap.add_argument("--host-input-chunks", type=int, default=1, help="pieces a batch's host -> device copy is cut into")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="only exercise the N-rank launch path (gloo on a CPU-only box)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)  # does not return
    if args.global_batch is not None:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if args.global_batch % (BATCH * world):
            ap.error(f"--global-batch must be a multiple of {BATCH} x {world} ranks")
        args.group = args.global_batch // (BATCH * world)
    defaults = {"joint": (100, 10), "encoder": (20, 5), "frontend": (200, 20),
                "dccrn": (20, 3), "train": (5, 1)}[args.workload]
    replicas_given = args.replicas is not None
    if args.replicas is None:
        args.replicas = {"joint": 2, "frontend": 3}.get(args.workload, 1)
    if args.pipeline is None:
        # the joint headline: three worker streams + the LSTM stream, unless the caller asked for --replicas R
        args.pipeline = 0 if replicas_given else 6
    if args.workload != "joint":
        args.pipeline = 0
    if args.steps is None:
        args.steps = defaults[0]
    if args.warmup is None:
        args.warmup = defaults[1]
    if args.batches is None:
        per = args.group if args.workload == "joint" else 1
        args.batches = max(3, 12 // per)
    # every stream gets the same number of graphs; at least one per stream (the pipeline: per WORKER stream, and at
    # least two rounds of them for the lookahead)
    per = args.pipeline if args.pipeline else args.replicas
    args.batches = max(per * (2 if args.pipeline else 1), -(-args.batches // per) * per)
    R = Ranks(args)
    try:
        if args.selftest_launch:
            return run_selftest_launch(args, R)
        if args.workload == "encoder":
            return run_encoder(args, R)
        if args.workload == "joint":
            return run_joint(args, R)
        if args.workload == "dccrn":
            return run_dccrn(args, R)
        if args.workload == "train":
            return run_train(args, R)
        return run_frontend(args, R)
    finally:
        if R.D.is_initialized():
            import torch.distributed as dist
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
