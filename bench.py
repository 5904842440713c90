#!/usr/bin/env python
"""
bench.py -- throughput of the aps joint front-end hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic utterances already resident in
HBM.  Default workload = BASELINE.json configs[4], the configuration the metric
"utterances/sec (4-ch 16 kHz 4 s) STFT->MVDR->encoder fwd" is quoted on (32 utterances per GPU =
its global batch 256 over 8 GPUs):
    EnhTransform STFT + log-magnitude/CMVN + cos-IPD -> RNNMaskMvdr (LSTM mask estimator, mask
    MVDR: covariance x2, channel attention, per-bin complex solve, beamform) ->
    AsrTransform abs-mel-log-cmvn -> 12-layer conformer encoder (conf/asr/chime4/1a.yaml) + CTC head
Other workloads: --workload frontend (configs[1]: STFT + features + MVDR with given masks, the
HBM-bound stage, with its HBM roofline), --workload encoder (configs[3]) and --workload dccrn
(configs[2]).
One process per GPU, utterances sharded by rank (weak scaling, no collective on the data path);
W untimed warm-up steps, then exactly K steps between barrier + synchronize pairs, max over ranks,
one JSON line from rank 0.  Every timed step is a replay of the whole step captured as one
hipGraph; the joint and front-end workloads keep --replicas batches in flight per GPU (2 / 3
captured copies of the step on as many streams, aps_amd/replicas.py), `ms_per_step` = timed
region / K.

The JSON line also carries
  roofline     : the dominant kernel's ALGORITHMIC flops (bytes for the front-end workload) per
                 launch / its mean duration measured with HIP events on the launch stream, against
                 the fp32 MFMA peak (HBM peak)
  cpu_baseline : the CPU oracle (a torch-CPU port of the reference, oracle/) timed on this box's
                 host cores on a bounded sample of the same workload (rank 0, N=1)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# workload constants
BATCH, CH, SAMPLES = 32, 4, 64000
FRAME_LEN, FRAME_HOP, BINS, PAIRS = 512, 256, 257, 3
FRAMES = (SAMPLES - FRAME_LEN) // FRAME_HOP + 1  # 249

# ALGORITHMIC bytes per utterance and kernel (fp32; SURVEY.md 8d, DESIGN.md "bytes per unit")
X_BYTES = CH * BINS * FRAMES * 8
ALGO_BYTES = {
    "stft": CH * SAMPLES * 4 + X_BYTES,                                   # R wav + W X
    "features": X_BYTES + FRAMES * BINS * (1 + PAIRS) * 4,                # R X + W feats
    "mvdr_weights": X_BYTES + 2 * FRAMES * BINS * 4 + BINS * CH * 8 + CH * 4,  # R X, masks; W w, u
    "beamform": X_BYTES + BINS * CH * 8 + FRAMES * BINS * 8,              # R X, w; W Y
}


def build_workload(device, rank):
    from aps_amd.asr.filter.mvdr import MvdrBeamformer
    from aps_amd.transform import EnhTransform
    g = torch.Generator().manual_seed(1 + 1000 * rank)
    x = 0.1 * torch.randn(BATCH, CH, SAMPLES, generator=g)
    g = torch.Generator().manual_seed(2 + 1000 * rank)
    masks = torch.sigmoid(torch.randn(BATCH, FRAMES, 2 * BINS, generator=g))
    mask_s, mask_n = [m.contiguous() for m in torch.chunk(masks, 2, -1)]
    torch.manual_seed(3)
    mvdr = MvdrBeamformer(BINS, att_dim=512, mask_norm=True)
    enh = EnhTransform(feats="spectrogram-log-cmvn-ipd", frame_len=FRAME_LEN, frame_hop=FRAME_HOP,
                       window="sqrthann", center=False, ipd_index="0,1;0,2;0,3", cos_ipd=True)
    enh.nan_policy = "deferred"  # NaN scan still runs in-kernel every step; host does not stall
    cpu = dict(x=x, mask_s=mask_s, mask_n=mask_n,
               att=[p.detach().clone() for p in (mvdr.ref.proj.weight, mvdr.ref.proj.bias,
                                                 mvdr.ref.gvec.weight, mvdr.ref.gvec.bias)])
    dev = dict(x=x.to(device), mask_s=mask_s.to(device), mask_n=mask_n.to(device),
               enh=enh.to(device), mvdr=mvdr.to(device))
    return cpu, dev


class Stages(object):
    """One step of the front-end, stage by stage.

    Dataflow of BASELINE config 2 (masks are given):   STFT ──► features            (stream B)
                                                          └──► covariance ► fold ► attention ►
                                                               weights ► beamform  (stream A)
    The feature kernel and the MVDR chain only share the spectrogram, so they run on two HIP
    streams; each is a chain of latency-bound launches at batch 32 and they overlap almost
    perfectly (measured: features 25 us hidden behind the 60 us MVDR chain).
    """

    ORDER = ["stft", "features", "mvdr_weights", "beamform"]
    KERNELS = {
        "stft": "stft512_wave_kernel<false, false>",
        "features": "features_rows_kernel<5>",
        "mvdr_weights": "covariance_partial_kernel<4, 64> + covariance_finalize_kernel<4, 4> + "
                        "attention_partial_kernel<4, true> + weight_kernel<4, true, true>",
        "beamform": "beamform_kernel<4, 0>",
    }

    def __init__(self, w, two_streams=True):
        from aps_amd.asr.filter import mvdr as M
        from aps_amd.spectrogram import packed_view
        self.w, self.M, self.packed_view = w, M, packed_view
        self.state = {}
        self.side = torch.cuda.Stream() if two_streams else None
        self.ready = torch.cuda.Event()

    def run_stage(self, name):
        w, M, st = self.w, self.M, self.state
        enh, mvdr = w["enh"], w["mvdr"]
        if name == "stft":
            st["store"] = enh.forward_stft.to_store(w["x"])
        elif name == "features":
            st["feats"] = enh(self.packed_view(st["store"]))
        elif name == "mvdr_weights":
            st["u"], st["wgt"] = mvdr.weights_from_masks(st["store"], w["mask_s"], w["mask_n"])
        elif name == "beamform":
            st["y"] = M.beamform_store(st["store"], st["wgt"])

    def step(self, probe=None, ev=None):
        """one full step; `probe` names the stage bracketed by the (start, stop) events, which
        are recorded on the stream that stage is launched on"""

        def stage(name):
            if probe == name:
                ev[0].record()
                self.run_stage(name)
                ev[1].record()
            else:
                self.run_stage(name)

        main = torch.cuda.current_stream()
        stage("stft")
        if self.side is not None:
            self.ready.record(main)
            self.state["store"].record_stream(self.side)
            with torch.cuda.stream(self.side):
                self.side.wait_event(self.ready)
                stage("features")
        else:
            stage("features")
        stage("mvdr_weights")
        stage("beamform")
        if self.side is not None:
            main.wait_stream(self.side)
        return self.state["feats"], self.state["y"]


# ---------------------------------------------------------------------------------------------
# --workload encoder : BASELINE configs[3], transformer encoder (12 x 512, FF 2048, conv2d 256 x 2,
# 80-mel, 400 frames -> 100) forward, batch 128 per GPU.  MFMA-bound; fp32 MFMA peak 157.3 TFLOP/s.
# ---------------------------------------------------------------------------------------------
ENC_BATCH, ENC_FRAMES, ENC_MELS = 128, 400, 80
ENC_FLOP_PER_UTT = 10.72e9  # torch flop counter on the reference module
MFMA_F32_PEAK_TFLOPS = 157.3


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


def encoder_cpu_baseline(cpu, budget_s=15.0):
    from oracle import encoder_oracle as eo
    n = 8
    x, lens = cpu["x"][:n], cpu["lens"][:n]
    eo.xfmr_abs_encoder(cpu["sd"], x, lens, 12, 8)
    t0, iters = time.perf_counter(), 0
    while True:
        eo.xfmr_abs_encoder(cpu["sd"], x, lens, 12, 8)
        iters += 1
        el = time.perf_counter() - t0
        if el > budget_s or iters >= 20:
            break
    return {"value": round(n * iters / el, 2), "unit": "utt/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{iters} forwards of {n} utterances ({el:.1f} s, torch-CPU oracle, "
                      f"{torch.get_num_threads()} threads)"}


def run_encoder(args, D, world, rank, device):
    from aps_amd import nn_ops
    cpu, dev = build_encoder(device, rank)
    enc, x, lens = dev["enc"], dev["x"], dev["lens"]
    with torch.no_grad():
        for _ in range(max(args.warmup, 2)):
            enc(x, lens)
        torch.cuda.synchronize()
        nn_ops.GEMM_TIMELINE = timeline = []
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            enc(x, lens)
        torch.cuda.synchronize()
        D.barrier()
        elapsed = time.perf_counter() - t0
        nn_ops.GEMM_TIMELINE = None
    elapsed = D.reduce_max(elapsed, device)
    total = D.reduce_sum(float(ENC_BATCH * args.steps), device)
    if rank != 0:
        return
    gemm_ms = sum(a.elapsed_time(b) for a, b, _ in timeline) / args.steps
    gemm_flop = sum(f for _, _, f in timeline) / args.steps
    launches = len(timeline) // args.steps
    achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12
    ms_per_step = 1e3 * elapsed / args.steps
    line = {
        "metric": "utterances/sec (4-ch 16 kHz 4 s) STFT→MVDR→encoder fwd, 1/2/4/8 MI355X",
        "value": round(total / elapsed, 1), "unit": "utt/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: asr transformer encoder (12x512, FF 2048, "
                               "conv2d 256x2 subsampling, 80-mel, 400 frames) forward only",
                   "batch_per_gpu": ENC_BATCH, "global_batch": ENC_BATCH * world,
                   "parallelism": f"dp{world} (utterance sharding, forward: no collective)"},
        "encoder_tflops_end_to_end": round(ENC_FLOP_PER_UTT * ENC_BATCH / (ms_per_step * 1e-3) / 1e12,
                                           2),
        "roofline": {"kernel": f"gemm_f32_kernel ({launches} launches / step, all nn.Linear)",
                     "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                     "traffic": None, "algo_flops_per_step": gemm_flop,
                     "kernel_ms_per_step": round(gemm_ms, 4)},
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = encoder_cpu_baseline(cpu)
    print(json.dumps(line))


# ---------------------------------------------------------------------------------------------
# --workload joint : BASELINE configs[4], the joint front end (SURVEY.md 8d "Config 5"):
# EnhTransform(spectrogram-log-cmvn-ipd) -> RNNMaskMvdr (1028 -> 512 -> 2 x LSTM 512 -> 514 masks,
# MVDR att 512) -> AsrTransform(abs-mel-log-cmvn, 80 mel) -> conformer (conf/asr/chime4/1a.yaml:
# 12 layers, conv2d 128 x 2, rel pose r = 256, 512 / 8 heads / FF 1024, k = 15) + CTC head,
# 32 utterances of 4 ch x 4 s per GPU (global batch 256 on 8 GPUs).
# ---------------------------------------------------------------------------------------------
JOINT_VOCAB = 5000


def build_joint(device, rank):
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
    g = torch.Generator().manual_seed(8 + 1000 * rank)
    src = 0.1 * torch.randn(BATCH, SAMPLES + 16, generator=g)
    wav = torch.stack([src[:, d:d + SAMPLES] for d in (0, 2, 5, 9)], 1)
    wav = (wav + 0.05 * torch.randn(BATCH, CH, SAMPLES, generator=g)).contiguous()
    lens = torch.tensor([SAMPLES] * BATCH)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    return dict(wav=wav, lens=lens, sd=sd), dict(wav=wav.to(device), lens=lens.to(device),
                                                 net=net.to(device))


def joint_cpu_baseline(cpu, budget_s=12.0):
    from oracle import joint_oracle as jo
    n = 4
    wav, lens = cpu["wav"][:n], cpu["lens"][:n]
    t0, iters = time.perf_counter(), 0
    while True:
        jo.joint_forward(cpu["sd"], wav, lens, num_mels=80, rnn_layers=2, enc_layers=12, nhead=8)
        iters += 1
        el = time.perf_counter() - t0
        if el > budget_s or iters >= 10:
            break
    return {"value": round(n * iters / el, 2), "unit": "utt/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{iters} joint forwards of {n} utterances ({el:.1f} s, torch-CPU oracle, "
                      f"{torch.get_num_threads()} threads)"}


def joint_stage_times(net, wav, lens, reps=5):
    """per-stage device time of one joint step (events on the launch stream, outside the timed
    region): where the step goes"""
    from aps_amd.cplx import ComplexTensor
    names = ["stft", "enh_features", "mask_net", "mvdr", "asr_features", "encoder+ctc"]
    acc = dict.fromkeys(names, 0.0)
    for _ in range(reps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        ev[0].record()
        packed, n = net.enh_transform.encode(wav, lens)
        ev[1].record()
        feats = net.enh_transform(packed)
        ev[2].record()
        mask, _ = net.enh_net.mask_net(feats, n)
        ev[3].record()
        mask_s, mask_n = torch.chunk(mask, 2, dim=-1)
        y = net.enh_net.mvdr_net(mask_s, ComplexTensor(packed[..., 0], packed[..., 1]), x_len=n,
                                 mask_n=mask_n)
        ev[4].record()
        x, _ = net.asr_transform(y, None)
        ev[5].record()
        net.asr(x, n)
        ev[6].record()
        torch.cuda.synchronize()
        for i, k in enumerate(names):
            acc[k] += ev[i].elapsed_time(ev[i + 1])
    return {k: round(1e3 * v / reps, 1) for k, v in acc.items()}


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


def run_joint(args, D, world, rank, device):
    """default workload.  Timed region = K passes of the joint step, each a replay of the whole step
    as one hipGraph, --replicas of them in flight on as many streams (--eager keeps plain launches).  The dominant kernel (the fp32 MFMA GEMM) is
    timed with HIP events on the launch stream in an instrumented eager pass of the same step
    right before the timed region: event records cannot sit inside a graph replay."""
    from aps_amd import nn_ops
    cpu, dev = build_joint(device, rank)
    net, wav, lens = dev["net"], dev["wav"], dev["lens"]
    # the NaN scan of check_valid runs inside the feature kernels every step; its counter is read
    # without stalling the stream (eager) / after the replays (graph), never skipped
    net.enh_transform.nan_policy = net.asr_transform.nan_policy = "deferred"
    with torch.no_grad():
        for _ in range(max(args.warmup, 2)):
            net(wav, lens)
        torch.cuda.synchronize()
        # ---- eager passes: host-bound step time, then per-GEMM events (roofline) + stage times
        probe_steps = max(1, min(args.steps, 10))
        t0 = time.perf_counter()
        for _ in range(probe_steps):
            net(wav, lens)
        torch.cuda.synchronize()
        eager_ms = 1e3 * (time.perf_counter() - t0) / probe_steps
        # Instrumented passes.  Eager launches are host bound (the GPU idles between kernels), and
        # an event bracket would then time the host's launch gap with the kernel.  So the stream is
        # first held busy by a spin kernel for longer than the host needs to enqueue the step: the
        # launches then sit back to back in the queue, like the nodes of the replayed graph, and a
        # bracket sees the kernel's duration.
        spin = spin_cycles_for(1.5 * eager_ms)
        nn_ops.GEMM_TIMELINE = timeline = []
        for _ in range(probe_steps):
            hold_stream(spin)
            net(wav, lens)
            torch.cuda.synchronize()
        nn_ops.GEMM_TIMELINE = None
        # what a bracket costs by itself (two event packets on a busy queue): empty brackets under
        # the same conditions; subtracted from every GEMM bracket below
        hold_stream(spin)
        empty = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                 for _ in range(64)]
        for a, b in empty:
            a.record()
            b.record()
        torch.cuda.synchronize()
        bracket_us = sorted(1e3 * a.elapsed_time(b) for a, b in empty)[len(empty) // 2]
        net.enh_transform._nan_guard.flush()
        net.asr_transform._nan_guard.flush()
        stages = joint_stage_times(net, wav, lens) if rank == 0 else None
        # ---- the whole step as ONE hipGraph (torch's capture API is only the recorder: every node
        # is one of our launches / memsets or a MIOpen conv): ~230 host launches per step -> 1
        # Two batches in flight (aps_amd/replicas.py): the latency-bound LSTM mask estimator of one
        # hides behind the GEMM-bound conformer of the other.  --replicas 1 = one graph, one stream.
        reps, launch, single_ms = None, "eager, one stream", None
        if not args.eager:
            try:
                from aps_amd.replicas import GraphReplicas, concurrent_launches
                net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
                with concurrent_launches(args.replicas):  # same LSTM decomposition as the replicas
                    ref_out = net(wav, lens)
                reps = GraphReplicas(lambda: net(wav, lens), replicas=args.replicas)
                for _ in range(2 * args.replicas):
                    reps.submit()
                torch.cuda.synchronize()
                for out in reps.outputs:
                    assert torch.equal(out[0], ref_out[0]), "graph replay differs from eager"
                # one replica alone, back to back: the step time without a second batch in flight
                t0 = time.perf_counter()
                for _ in range(probe_steps):
                    with torch.cuda.stream(reps.streams[0]):
                        reps.graphs[0].replay()
                torch.cuda.synchronize()
                single_ms = 1e3 * (time.perf_counter() - t0) / probe_steps
                launch = ("hipGraph replay of the whole step" if args.replicas == 1 else
                          f"{args.replicas} hipGraph replicas of the whole step, round-robin on "
                          f"{args.replicas} streams ({args.replicas} batches in flight)")
            except Exception as exc:  # noqa: BLE001  (capture unsupported: stay eager, say so)
                print(f"[bench] graph capture failed ({exc}); timing eager launches",
                      file=sys.stderr)
                reps = None
                torch.cuda.synchronize()
                net.enh_transform.nan_policy = net.asr_transform.nan_policy = "deferred"
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if reps is not None:
                reps.submit(after_caller=False)  # static inputs
            else:
                net(wav, lens)
        torch.cuda.synchronize()
        D.barrier()
        elapsed = time.perf_counter() - t0
        if reps is not None:
            for out in reps.outputs:
                assert torch.equal(out[0], ref_out[0]), "graph replay differs from eager"
        nans = net.enh_transform._nan_guard.count() + net.asr_transform._nan_guard.count()
        assert nans == 0, f"{nans} NaN rows in the features"
    elapsed = D.reduce_max(elapsed, device)
    total = D.reduce_sum(float(BATCH * args.steps), device)
    if rank != 0:
        return
    raw_ms = sum(a.elapsed_time(b) for a, b, _ in timeline) / probe_steps
    gemm_flop = sum(f for _, _, f in timeline) / probe_steps
    launches = len(timeline) // probe_steps
    gemm_ms = raw_ms - launches * bracket_us * 1e-3  # minus the brackets' own cost
    achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12
    ms_per_step = 1e3 * elapsed / args.steps
    line = {
        "metric": "utterances/sec (4-ch 16 kHz 4 s) STFT→MVDR→encoder fwd, 1/2/4/8 MI355X",
        "value": round(total / elapsed, 1), "unit": "utt/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "launch": launch,
        "config": {"workload": "BASELINE configs[4]: joint front end, 4-ch 4 s -> STFT + IPD "
                               "features -> LSTM masks -> MVDR -> 80-mel log/cmvn -> 12-layer "
                               "conformer (chime4/1a geometry) + CTC head, forward only",
                   "batch_per_gpu": BATCH, "global_batch": BATCH * world,
                   "batches_in_flight": args.replicas if reps is not None else 1,
                   "frames": FRAMES, "encoder_frames": ((FRAMES - 1) // 2 // 2) + 1,
                   "parallelism": f"dp{world} (utterance sharding, forward: no collective)"},
        "eager_ms_per_step": round(eager_ms, 3),
        "single_stream_ms_per_step": None if single_ms is None else round(single_ms, 3),
        "stage_us": stages,
        "roofline": {"kernel": f"gemm_f32_kernel ({launches} launches / step: mask-net, conformer "
                               "and CTC projections)",
                     "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                     "traffic": None, "algo_flops_per_step": gemm_flop,
                     "kernel_ms_per_step": round(gemm_ms, 4),
                     "bracketed_ms_per_step": round(raw_ms, 4),
                     "empty_bracket_us": round(bracket_us, 2),
                     "measured": f"HIP events around every launch in {probe_steps} queued-ahead eager "
                                 "passes of the same step, minus the cost of an empty bracket "
                                 "measured the same way"},
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = joint_cpu_baseline(cpu)
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


def dccrn_cpu_baseline(cpu, budget_s=12.0):
    from oracle import dccrn_oracle as do
    n = 4
    mix = cpu["mix"][:n]
    cfg = dict(K="3,3;3,3;3,3;3,3;3,3;3,3;3,3", S="2,1;2,1;2,1;2,1;2,1;2,1;2,1",
               P="1,1,1,1,1,1,1", O="0,0,0,0,0,0,0")
    t0, iters = time.perf_counter(), 0
    while True:
        do.dccrn_forward(cpu["sd"], mix, **cfg)
        iters += 1
        el = time.perf_counter() - t0
        if el > budget_s or iters >= 10:
            break
    return {"value": round(n * iters / el, 2), "unit": "utt/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{iters} DCCRN forwards of {n} mixtures ({el:.1f} s, torch-CPU oracle, "
                      f"{torch.get_num_threads()} threads)"}


def run_dccrn(args, D, world, rank, device):
    from aps_amd import nn_ops
    cpu, dev = build_dccrn(device, rank)
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
                from aps_amd.replicas import GraphReplicas, concurrent_launches
                with concurrent_launches(args.replicas):
                    ref_out = net(mix)
                reps = GraphReplicas(lambda: net(mix), replicas=args.replicas)
                launch = ("hipGraph replay of the whole step" if args.replicas == 1 else
                          f"{args.replicas} hipGraph replicas of the whole step, round-robin on "
                          f"{args.replicas} streams ({args.replicas} batches in flight)")
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] graph capture failed ({exc}); timing eager launches",
                      file=sys.stderr)
                reps = None
                torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if reps is not None:
                reps.submit(after_caller=False)  # static inputs
            else:
                net(mix)
        torch.cuda.synchronize()
        D.barrier()
        elapsed = time.perf_counter() - t0
        if reps is not None:
            for out in reps.outputs:
                assert torch.equal(out[0], ref_out[0]), "graph replay differs from eager"
    elapsed = D.reduce_max(elapsed, device)
    total = D.reduce_sum(float(DCCRN_BATCH * args.steps), device)
    if rank != 0:
        return
    conv_ms = sum(t[0].elapsed_time(t[1]) for t in timeline) / probe_steps
    conv_flop = sum(t[2] for t in timeline) / probe_steps
    launches = len(timeline) // probe_steps
    achieved = conv_flop / (conv_ms * 1e-3) / 1e12
    ms_per_step = 1e3 * elapsed / args.steps
    line = {
        "metric": "utterances/sec (4-ch 16 kHz 4 s) STFT→MVDR→encoder fwd, 1/2/4/8 MI355X",
        "value": round(total / elapsed, 1), "unit": "utt/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "launch": launch,
        "config": {"workload": "BASELINE configs[2]: sse DCCRN forward on 2-spk 8 kHz 4 s mixtures "
                               "(STFT -> mask estimator -> masking -> iSTFT), NOT the headline "
                               "metric's utterance type",
                   "batch_per_gpu": DCCRN_BATCH, "global_batch": DCCRN_BATCH * world,
                   "parallelism": f"dp{world} (utterance sharding, forward: no collective)"},
        "eager_ms_per_step": round(eager_ms, 3),
        "model_tflops_end_to_end": round(DCCRN_FLOP_PER_UTT * DCCRN_BATCH / (ms_per_step * 1e-3) / 1e12,
                                         2),
        "roofline": {"kernel": f"conv_mfma_kernel / conv_direct_kernel ({launches} launches / step: "
                               "the complex conv / deconv blocks)",
                     "bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                     "traffic": None, "algo_flops_per_step": conv_flop,
                     "kernel_ms_per_step": round(conv_ms, 4),
                     "measured": f"HIP events around every launch, {probe_steps} eager passes"},
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = dccrn_cpu_baseline(cpu)
    print(json.dumps(line))


def cpu_baseline(cpu, budget_s=12.0):
    """oracle on the host cores, bounded sample of the same workload"""
    from oracle import aps_oracle as orc
    n = 8
    x, ms, mn = cpu["x"][:n], cpu["mask_s"][:n], cpu["mask_n"][:n]
    threads = torch.get_num_threads()

    def once():
        packed = orc.stft(x, FRAME_LEN, FRAME_HOP, "sqrthann")
        feats = orc.enh_features(packed, "spectrogram-log-cmvn-ipd", "0,1;0,2;0,3")
        yr, yi, _ = orc.mvdr_forward(ms, packed[..., 0], packed[..., 1], cpu["att"], mn)
        return feats, yr, yi

    once()  # warm-up
    t0 = time.perf_counter()
    iters = 0
    while True:
        once()
        iters += 1
        el = time.perf_counter() - t0
        if el > budget_s or iters >= 20:
            break
    return {
        "value": round(n * iters / el, 2),
        "unit": "utt/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{iters} passes over {n} of the batch's {BATCH} utterances "
                  f"({el:.1f} s, torch-CPU oracle, {threads} threads)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="joint", choices=["joint", "frontend", "encoder", "dccrn"],
                    help="joint = BASELINE configs[4], STFT -> MVDR -> encoder forward, the "
                         "configuration the metric is quoted on (default); frontend = configs[1] "
                         "(STFT + features + MVDR with given masks); encoder = configs[3]; "
                         "dccrn = configs[2]")
    ap.add_argument("--eager", action="store_true",
                    help="time plain launches instead of the captured hipGraph")
    ap.add_argument("--replicas", type=int, default=None,
                    help="joint / frontend / dccrn workloads: batches in flight per GPU, each a "
                         "captured hipGraph on its own stream (1 = a single graph on one stream; "
                         "default 2 for joint, 3 for frontend, 1 for dccrn)")
    ap.add_argument("--two-streams", action="store_true", help="run the feature kernel beside the MVDR chain on a second stream (measured slower)")
    args = ap.parse_args()

    from aps_amd import distributed as D
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        torch.cuda.set_device(D.local_rank())
        D.init("torch", "nccl")
    rank = D.rank()
    if world != args.gpus and rank == 0:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    device = torch.device("cuda", D.local_rank() if world > 1 else 0)
    torch.cuda.set_device(device)

    defaults = {"joint": (100, 10), "encoder": (20, 5), "frontend": (200, 20),
                "dccrn": (20, 3)}[args.workload]
    if args.replicas is None:
        args.replicas = {"joint": 2, "frontend": 3}.get(args.workload, 1)
    if args.steps is None:
        args.steps = defaults[0]
    if args.warmup is None:
        args.warmup = defaults[1]
    if args.workload == "encoder":
        return run_encoder(args, D, world, rank, device)
    if args.workload == "joint":
        return run_joint(args, D, world, rank, device)
    if args.workload == "dccrn":
        return run_dccrn(args, D, world, rank, device)

    cpu, dev = build_workload(device, rank)
    stages = Stages(dev, two_streams=args.two_streams)
    order = Stages.ORDER
    enh = dev["enh"]
    enh.nan_policy = "deferred"  # the NaN scan runs in-kernel every step; the host does not stall

    with torch.no_grad():
        # ---- warm-up: W full steps, then time every stage alone to find the dominant kernel ----
        for _ in range(max(args.warmup, 2)):
            stages.step()
        torch.cuda.synchronize()
        stage_ms = {}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for name in order:
            for _ in range(3):
                stages.run_stage(name)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(20):
                stages.run_stage(name)
            e1.record()
            torch.cuda.synchronize()
            stage_ms[name] = e0.elapsed_time(e1) / 20
        single = {"stft", "features", "beamform"}  # stages that are exactly one kernel launch
        dominant = max(single, key=lambda k: stage_ms[k])

        # the dominant kernel is bracketed with HIP events in instrumented eager passes (event
        # records cannot sit inside a graph replay); the timed region replays the captured step
        probes = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for _ in range(min(args.steps, 50))]
        for _ in range(3):
            stages.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in probes:
            stages.step()
        torch.cuda.synchronize()
        eager_ms = 1e3 * (time.perf_counter() - t0) / len(probes)
        # queued-ahead instrumented steps (the stream is held busy while the host enqueues, see
        # run_joint) and the cost of an empty bracket under the same conditions
        spin = spin_cycles_for(3 * eager_ms)
        for ev in probes:
            hold_stream(spin)
            stages.step(probe=dominant, ev=ev)
        hold_stream(spin)
        empty = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                 for _ in range(64)]
        for a, b in empty:
            a.record()
            b.record()
        torch.cuda.synchronize()
        bracket_ms = sorted(a.elapsed_time(b) for a, b in empty)[len(empty) // 2]
        enh._nan_guard.flush()
        graph = None
        if not args.eager and not args.two_streams:
            try:
                from aps_amd.replicas import GraphReplicas
                enh.nan_policy = "manual"  # host reads are illegal while capturing
                ref_feats, ref_y = [t.clone() for t in stages.step()]
                graph = GraphReplicas(stages.step, replicas=args.replicas)
                for g_feats, g_y in graph.outputs:
                    assert torch.equal(g_feats, ref_feats) and torch.equal(g_y, ref_y), \
                        "graph replay differs from eager"
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] graph capture failed ({exc}); timing eager launches",
                      file=sys.stderr)
                graph = None
                enh.nan_policy = "deferred"
                torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            if graph is not None:
                graph.submit(after_caller=False)  # static inputs
            else:
                stages.step()
        torch.cuda.synchronize()
        D.barrier()
        elapsed = time.perf_counter() - t0
        if graph is not None:
            assert enh._nan_guard.count() == 0, "NaN in the features"
            for g_feats, g_y in graph.outputs:
                assert torch.equal(g_feats, ref_feats) and torch.equal(g_y, ref_y), \
                    "graph replay differs from eager"
        else:
            enh._nan_guard.flush()

    elapsed = D.reduce_max(elapsed, device)
    total_utts = D.reduce_sum(float(BATCH * args.steps), device)
    if rank != 0:
        return
    ms_per_step = 1e3 * elapsed / args.steps
    value = total_utts / elapsed
    raw_kern_ms = sum(a.elapsed_time(b) for a, b in probes) / len(probes)
    kern_ms = raw_kern_ms - bracket_ms  # minus the bracket's own cost (closing event packet)
    algo = ALGO_BYTES[dominant] * BATCH
    achieved = algo / (kern_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get(dominant)
        except Exception:
            traffic = None
    line = {
        "metric": "utterances/sec (4-ch 16 kHz 4 s) STFT→MVDR→encoder fwd, 1/2/4/8 MI355X",
        "value": round(value, 1),
        "unit": "utt/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE configs[1]: EnhTransform 4-ch 16 kHz 4 s -> STFT + log-mag/CMVN "
                        "+ cos-IPD(3 pairs) + mask-MVDR (cov x2, attention, solve, beamform), "
                        "masks given; the encoder is in the default (joint) workload",
            "batch_per_gpu": BATCH,
            "global_batch": BATCH * world,
            "batches_in_flight": args.replicas if graph is not None else 1,
            "frame": "512/256 sqrthann",
            "parallelism": f"dp{world} (utterance sharding, no collective)",
            "launch": "eager, 2 streams (features || covariance..beamform)" if args.two_streams
                      else ((f"hipGraph replay of the whole step, {args.replicas} batch(es) in "
                             f"flight on as many streams") if graph is not None
                            else "eager, 1 stream"),
        },
        "eager_ms_per_step": round(eager_ms, 4),
        "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
        "algo_gbs_all_stages": round(sum(ALGO_BYTES.values()) * BATCH / (ms_per_step * 1e-3) / 1e9,
                                     1),
        "roofline": {
            "kernel": Stages.KERNELS[dominant],
            "stage": dominant,
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "algo_bytes_per_launch": algo,
            "kernel_ms": round(kern_ms, 5),
            "bracketed_ms": round(raw_kern_ms, 5),
            "empty_bracket_ms": round(bracket_ms, 5),
        },
    }
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(cpu)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
