"""
CPU: host-side logic of the product path -- registry surface, ctor contracts, parameter names and
shapes (checkpoint compatibility), frame arithmetic, the C-ABI library's exported symbols -- with
no compute calls (there is no GPU here and the product has no CPU fallback).
"""
import ctypes
import os
import re

import pytest
import torch

from tests.conftest import golden, ROOT


def test_library_exports_every_declared_symbol():
    from aps_amd import _native
    lib = _native.load()
    header = open(os.path.join(ROOT, "include", "aps_amd.h")).read()
    declared = set(re.findall(r"\b(aps_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_native.SIGNATURES), (declared ^ set(_native.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/aps_amd.h but not exported"
    assert lib.aps_abi_version() == _native.ABI_VERSION
    assert lib.aps_status_string(-2).decode().startswith("configuration not supported")


def test_every_entry_point_is_documented():
    """INTEGRATION.md names the reference interface behind every declared entry point"""
    header = open(os.path.join(ROOT, "include", "aps_amd.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    declared = set(re.findall(r"\b(aps_[a-z0-9_]+)\s*\(", header))
    assert not [n for n in sorted(declared) if n not in doc]


def test_num_frames_c_abi_matches_reference_table():
    from aps_amd import _native
    lib = _native.load()
    for fl, fh, kaldi, center, S, T, L, nb in golden("num_frames")["table"].tolist():
        W = (nb - 1) * 2
        p = _native.StftParams(W, L, fh, nb, center, 0, 0.0, 0.0, 1.0)
        assert lib.aps_stft_num_frames(S, ctypes.byref(p)) == T
    p = _native.StftParams(512, 512, 256, 257, 0, 0, 0.0, 0.0, 1.0)
    assert lib.aps_stft_num_frames(100, ctypes.byref(p)) == 0


def test_invalid_arguments_are_rejected_without_a_gpu():
    from aps_amd import _native
    lib = _native.load()
    p = _native.StftParams(512, 512, 256, 257, 0, 0, 0.0, 0.0, 1.0)
    rc = lib.aps_stft_forward(None, 1, 1000, None, ctypes.byref(p), None, 0, 0, 1, None)
    assert rc == -1


def test_registry_surface():
    from aps_amd.libs import ApsRegisters, aps_transform, aps_nnet, aps_specific_nnet
    asr = aps_transform("asr")
    enh = aps_transform("enh")
    from aps_amd.transform import AsrTransform, EnhTransform
    assert asr is AsrTransform and enh is EnhTransform
    assert set(r.name for r in ApsRegisters.container) == {"asr", "sse", "task", "loader",
                                                           "trainer", "transform"}
    with pytest.raises(RuntimeError):
        aps_transform("nope")
    with pytest.raises(RuntimeError):
        aps_nnet("foo@bar")
    with pytest.warns(UserWarning):
        ApsRegisters.transform.register("asr")(AsrTransform)


def test_kernel_and_window_parameters_match_reference():
    from aps_amd.transform.utils import init_kernel, init_window, STFT, iSTFT
    g = golden("windows")
    for key, ref in g.items():
        name, n = key.rsplit("_", 1)
        assert torch.equal(init_window(name, int(n)), ref), key
    g = golden("kernels")
    for mode in ["librosa", "kaldi"]:
        for nrm in [0, 1]:
            for inv in [0, 1]:
                K, w = init_kernel(30, 10, init_window("hamm", 30), True, bool(nrm), bool(inv), mode)
                assert torch.equal(K, g[f"K30_{mode}_n{nrm}_i{inv}"])
                assert torch.equal(w, g[f"w30_{mode}_n{nrm}_i{inv}"])
    m = STFT(400, 160, window="sqrthann", mode="kaldi")
    assert list(m.K.shape) == g["K400_kaldi_shape"].tolist()
    assert torch.equal(m.K[::37, 0, ::29], g["K400_kaldi_probe"])
    assert not m.K.requires_grad and not m.w.requires_grad
    assert set(iSTFT(512, 256).state_dict()) == {"K", "w"}
    assert set(STFT(512, 256, mode="torch").state_dict()) == {"w"}


def test_transform_ctor_contract():
    from aps_amd.transform import AsrTransform, EnhTransform
    a = AsrTransform(feats="fbank-log-cmvn", frame_len=400, frame_hop=160, window="hamm",
                     pre_emphasis=0.97, num_mels=80)
    assert a.feats_dim == a.dim() == 80 and a.spectra_index == 0 and a.perturb_index == -1
    assert [type(m).__name__ for m in a.transform] == [
        "SpectrogramTransform", "MagnitudeTransform", "TFTransposeTransform", "PowerTransform",
        "MelTransform", "LogTransform", "CmvnTransform"]
    assert set(a.state_dict()) == {"transform.0.K", "transform.0.w", "transform.4.filters"}
    assert torch.equal(a.transform[4].filters, golden("asr_cfg1_fbank_log_cmvn")["mel_filters"])
    assert a.num_frames(torch.tensor([64000, 8000])).tolist() == [397, 47]
    e = EnhTransform(feats="spectrogram-log-cmvn-ipd", ipd_index="0,1;0,2;0,3")
    assert e.feats_dim == 257 * 4 and e.forward_stft.num_bins == 257
    assert [type(m).__name__ for m in e.mag_transform] == [
        "RefChannelTransform", "MagnitudeTransform", "TFTransposeTransform", "PowerTransform",
        "LogTransform", "CmvnTransform"]
    assert set(e.state_dict()) == {"forward_stft.K", "forward_stft.w", "inverse_stft.K",
                                   "inverse_stft.w"}
    assert type(e.ctx("inverse_stft")).__name__ == "iSTFT"
    with pytest.raises(ValueError):
        e.ctx("foo")
    with pytest.raises(RuntimeError):
        AsrTransform(feats="fbank-nope")
    with pytest.raises(ValueError):
        AsrTransform(feats="")
    # training-time randomised tokens: the reference's layers and frozen parameters, identity in
    # eval mode; in training mode there is no CPU path (loud without the GPU / on host tensors)
    g = golden("perturb_aug_eval")
    p = AsrTransform(feats="perturb-fbank-log-cmvn-aug", frame_len=400, frame_hop=160, window="hamm",
                     num_mels=40, speed_perturb="0.9,1.0,1.1", aug_prob=0.5)
    assert p.perturb_index == 0 and p.spectra_index == 1
    shapes = {k[6:]: v.tolist() for k, v in g.items() if k.startswith("shape.")}
    assert {k: list(v.shape) for k, v in p.state_dict().items()} == shapes
    assert torch.equal(p.transform[0].weights[0], g["filt_09"])
    assert torch.equal(p.transform[0].weights[1], g["filt_11"])
    x = torch.randn(2, 50)
    p.eval()
    assert p.transform[0](x) is x and p.transform[-1](x) is x
    p.train()
    p.transform[-1].p = 1.0
    with pytest.raises(Exception):
        p.transform[0](x)
    with pytest.raises(Exception):
        p.transform[-1](torch.randn(2, 30, 40))
    m = AsrTransform(feats="mfcc-cmvn-delta-splice", num_mels=40, num_ceps=13, lifter=22, lctx=1,
                     rctx=1, subsampling_factor=2)
    assert m.feats_dim == 13 * 3 * 3 and m.subsampling_factor == 2
    assert [type(x).__name__ for x in m.transform][4:] == [
        "MelTransform", "LogTransform", "DiscreteCosineTransform", "CmvnTransform",
        "DeltaTransform", "SpliceTransform"]
    assert set(m.state_dict()) == {"transform.0.K", "transform.0.w", "transform.4.filters",
                                   "transform.6.dct", "transform.6.cepstral_lifter",
                                   "transform.8.scale"}  # the reference's frozen parameters
    assert m.num_frames(torch.tensor([8000])).tolist() == [23]


def test_mel_band_form_reproduces_dense_matrix():
    from aps_amd.ops import MelBands
    from aps_amd.transform.utils import mel_filter
    w = mel_filter(400, num_mels=40, fmin=20, fmax=-400, norm=True)
    b = MelBands(w)
    dense = torch.zeros_like(w)
    for m in range(b.num_mels):
        s, n, o = int(b.start[m]), int(b.length[m]), int(b.offset[m])
        dense[m, s:s + n] = b.weight[o:o + n]
    assert torch.equal(dense, w)


def test_spectrogram_store_views():
    from aps_amd.spectrogram import alloc_store, packed_view, store_of, store_of_pair
    st = alloc_store((2, 3), 5, 9, "cpu").normal_()
    pk = packed_view(st)
    assert pk.shape == (2, 3, 9, 5, 2)
    assert store_of(pk).data_ptr() == st.data_ptr()          # zero-copy for our own views
    v = store_of_pair(pk[..., 0], pk[..., 1])
    assert v.data_ptr() == st.data_ptr() and torch.equal(v, st)
    foreign = pk.contiguous()
    assert torch.equal(store_of(foreign), st)
    assert torch.equal(store_of_pair(foreign[..., 0].contiguous(), foreign[..., 1].contiguous()), st)


def test_complex_tensor_algebra():
    """the reference's ComplexTensor surface (aps/cplx.py) against numpy complex arithmetic"""
    import numpy as np
    from aps_amd.cplx import ComplexTensor
    g = torch.Generator().manual_seed(4)
    ar, ai, br, bi = [torch.randn(3, 4, 4, generator=g, dtype=torch.float64) for _ in range(4)]
    a, b = ComplexTensor(ar, ai), ComplexTensor(br, bi)
    na, nb = ar.numpy() + 1j * ai.numpy(), br.numpy() + 1j * bi.numpy()

    def same(x, y):
        assert np.allclose(x.real.numpy() + 1j * x.imag.numpy(), y, atol=1e-10)

    same(a + b, na + nb)
    same(a - b, na - nb)
    same(2.0 - a, 2.0 - na)
    same(a * b, na * nb)
    same(a / b, na / nb)
    same(3.0 / b, 3.0 / nb)
    same(a @ b, na @ nb)
    same(a @ br, na @ br.numpy())
    same(a.inverse(), np.linalg.inv(na))
    same(a.conj_transpose(-1, -2), np.conj(np.swapaxes(na, -1, -2)))
    assert np.allclose(a.abs().numpy(), np.abs(na)) and np.allclose(a.angle().numpy(), np.angle(na))
    same(ComplexTensor(a.abs(), a.angle(), polar=True), na)


def test_concurrent_launches_scopes_the_library_state():
    """aps_amd/replicas.py: the sizing hint for memory-synchronised grids is library state (the
    explicit `share` argument of aps_lstm_layer / aps_lstm_stack), held only inside the context /
    for the lifetime of its holder and released in any order; no environment variable is involved;
    a replica count below 1 is refused before anything touches the GPU"""
    import os
    from aps_amd import nn_ops
    from aps_amd.replicas import GraphReplicas, concurrent_launches
    assert nn_ops.lstm_share() == 1
    with concurrent_launches(2):
        assert nn_ops.lstm_share() == 2
        with concurrent_launches(3):
            assert nn_ops.lstm_share() == 3
        assert nn_ops.lstm_share() == 2
    assert nn_ops.lstm_share() == 1
    nn_ops.push_lstm_share(2)
    nn_ops.push_lstm_share(4)
    nn_ops.pop_lstm_share(2)  # holders go away in any order
    assert nn_ops.lstm_share() == 4
    nn_ops.pop_lstm_share(4)
    assert nn_ops.lstm_share() == 1
    assert "APS_LSTM_CONCURRENT" not in os.environ
    with pytest.raises(ValueError):
        nn_ops.push_lstm_share(0)
    with pytest.raises(ValueError):
        GraphReplicas(lambda: None, replicas=0)
    from aps_amd.replicas import PipelinedReplicas, hardware_queues
    for bad in (dict(workers=0), dict(lstm_share=0), dict(front="own"), dict(mid="own")):
        with pytest.raises(ValueError):
            PipelinedReplicas([lambda: None], **bad)
    assert nn_ops.lstm_share() == 1 and nn_ops.STREAMS_IN_FLIGHT == 1 and nn_ops.STAGE_HOOK is None
    # the one-launch conformer stack is the library's choice while FOUR or more streams launch (the staged pipeline with
    # three or more workers); one stream and two whole steps in flight keep one launch per projection
    from aps_amd import mega
    if mega.ENABLED == "auto":
        for streams, share, want in ((1, 1, False), (1, 2, False), (3, 2, False), (4, 2, True), (7, 2, True)):
            nn_ops.STREAMS_IN_FLIGHT = streams
            nn_ops.push_lstm_share(share)
            try:
                assert mega.wanted() is want, (streams, share)
            finally:
                nn_ops.pop_lstm_share(share)
                nn_ops.STREAMS_IN_FLIGHT = 1
    keep = os.environ.pop("GPU_MAX_HW_QUEUES", None)
    try:
        assert hardware_queues() == 4   # the HIP default
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
        assert hardware_queues() == 8
    finally:
        os.environ.pop("GPU_MAX_HW_QUEUES", None)
        if keep is not None:
            os.environ["GPU_MAX_HW_QUEUES"] = keep


def test_mvdr_mask_operands_pitch_rule():
    """aps_amd/asr/filter/mvdr.py:_mask_operands -- which masks the MVDR kernels read in place (mask_ld) and which
    are made dense first: host logic on strides and dtypes only"""
    import torch as th
    from aps_amd.asr.filter.mvdr import _mask_operands
    N, T, F = 3, 7, 5
    est = th.rand(N, T, 2 * F)
    ms, mn = th.chunk(est, 2, dim=-1)
    a, b, ld = _mask_operands(ms, mn, N, T, F)
    assert ld == 2 * F and a.data_ptr() == ms.data_ptr() and b.data_ptr() == mn.data_ptr()
    a, b, ld = _mask_operands(ms, None, N, T, F)
    assert ld == 2 * F and b is None
    dense = th.rand(N, T, F)
    assert _mask_operands(dense, None, N, T, F)[2] == F
    a, b, ld = _mask_operands(ms, dense, N, T, F)               # two pitches: both dense
    assert ld == F and a.is_contiguous() and b.is_contiguous() and th.equal(a, ms)
    assert _mask_operands(est[:, :, :F].double(), None, N, T, F)[2] == F          # not fp32: converted
    assert _mask_operands(th.rand(N, T + 2, 2 * F)[:, :T, :F], None, N, T, F)[2] == F   # utterances not T ld apart
    assert _mask_operands(th.rand(N, F, T).transpose(1, 2), None, N, T, F)[2] == F      # bins not unit-stride
    grad = th.rand(N, T, 2 * F, requires_grad=True)[..., :F]   # under autograd: a dense copy, as before
    assert _mask_operands(grad, None, N, T, F)[2] == F
    with pytest.raises(RuntimeError):
        _mask_operands(th.rand(N, T, F + 1), None, N, T, F)
    with pytest.raises(RuntimeError):
        _mask_operands(dense, th.rand(N, T + 1, F), N, T, F)


def test_split_gemm_dispatch_rules():
    """aps_amd/nn_ops.py: which launches take the bf16-split kernels and whose weights may carry a
    cached planes image (no GPU involved: the rules are host logic)"""
    import torch.nn as nn
    from aps_amd import nn_ops
    saved = nn_ops.SPLIT_MODE
    try:
        nn_ops.SPLIT_MODE = None  # default rule: >= SPLIT_MIN_TILES tiles of 64 x 128 and K >= 128
        assert nn_ops._use_split(8064, 512, 512)          # the merged-batch conformer projections
        assert nn_ops._use_split(31872, 2048, 512)        # the mask estimator's input projections
        assert nn_ops._use_split(2016, 1536, 512)         # 32 utterances: the QKV projection (384 tiles)
        assert nn_ops._use_split(2016, 1024, 512)         # ... and the N = 1024 projections (256: one per CU)
        assert nn_ops._use_split(2016, 512, 512)          # ... and N = 512 (128 tiles, run as 256 of 64 x 64)
        assert not nn_ops._use_split(1008, 512, 512)      # 64 tiles: fp32 kernel
        saved_layout = nn_ops.SPLIT_LAYOUT
        nn_ops.SPLIT_LAYOUT = 2  # the planes-pass form: 64 x 64 tiles up to 400 tiles of 64 x 128
        assert nn_ops.fp16x2_tiles(2016, 512) == 256 and nn_ops.fp16x2_tiles(8064, 512) == 504
        nn_ops.SPLIT_LAYOUT = 3  # the panel form: 32-row panels below 512 tiles of 64 x 128, 64-row ones above
        assert nn_ops.fp16x2_tiles(2016, 512) == 63 * 4 and nn_ops.fp16x2_tiles(8064, 1024) == 126 * 8
        nn_ops.SPLIT_LAYOUT = saved_layout
        assert not nn_ops._use_split(8064, 512, 64)       # short K
        nn_ops.SPLIT_MODE = "1"
        assert nn_ops._use_split(1, 1, 4)
        nn_ops.SPLIT_MODE = "0"
        assert not nn_ops._use_split(8064, 512, 512)
    finally:
        nn_ops.SPLIT_MODE = saved
    lin = nn.Linear(8, 4)
    conv = nn.Conv1d(8, 4, 1)
    assert nn_ops._weight_owner(lin.weight) is lin.weight
    assert nn_ops._weight_owner(conv.weight.view(4, 8)) is conv.weight      # a view of a Parameter
    assert nn_ops._weight_owner(lin.weight * 2.0) is None                   # a temporary: never cached
    assert nn_ops._weight_owner(lin.weight.detach().clone()) is None
    kept = lin.weight.detach().clone()
    kept._aps_persistent = True  # what a module's own cached re-layout of a weight declares
    assert nn_ops._weight_owner(kept) is kept


def test_spec_augment_draws_follow_the_reference():
    """draw_tf_bands consumes Python's `random` like tf_mask / random_mask (augment.py:13-83): the
    bands of a seeded run are the zero regions of the reference's recorded output"""
    import random
    from aps_amd.transform.asr import draw_tf_bands
    g = golden("spec_augment_train")
    x, y = g["zero.x"], g["zero.y"]
    seed = int(g["zero.seed"])
    torch.manual_seed(seed)
    random.seed(seed)
    assert torch.rand(1).item() < 1.0  # the layer's coin flip comes first
    bands, nf, nt = draw_tf_bands(3, (50, 40), max_bands=8, max_frame=12, num_freq_masks=2,
                                  num_time_masks=2)
    assert (nf, nt) == (2, 2) and all(len(b) == 4 for b in bands)
    keep = torch.ones(3, 50, 40, dtype=torch.bool)
    for n, per_utt in enumerate(bands):
        for q, (beg, dur) in enumerate(per_utt):
            if q < nf:
                keep[n, :, beg:beg + dur] = False
            else:
                keep[n, beg:beg + dur, :] = False
    assert torch.equal(keep, y != 0)
    assert torch.equal(x * keep, y)
    # adaptive limits (pm, ps) and the skipped draw (band longer than the axis)
    random.seed(1)
    bands, nf, nt = draw_tf_bands(1, (60, 16), pm=0.04, ps=0.1, max_bands=10, max_frame=40,
                                  num_freq_masks=1, num_time_masks=4)
    assert (nf, nt) == (1, 2) and all(dur <= 5 for _, dur in bands[0][1:])


def test_product_never_reaches_for_the_oracle_and_fails_loudly_without_the_library():
    """the oracle is test infrastructure: no module under aps_amd/ imports it (bench.py only inside
    its cpu_baseline functions); without the built extension the product raises, it has no CPU
    path to fall back to"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pattern = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for folder, _, files in os.walk(os.path.join(root, "aps_amd")):
        for name in files:
            if name.endswith(".py"):
                text = open(os.path.join(folder, name)).read()
                assert not pattern.search(text), f"{name} imports the oracle"
    bench = open(os.path.join(root, "bench.py")).read()
    for m in pattern.finditer(bench):
        head = bench[:m.start()]
        owner = re.findall(r"^def (\w+)\(", head, re.M)[-1]
        assert owner.endswith("cpu_baseline"), f"bench.py imports the oracle in {owner}()"
    code = ("import os; os.environ['APS_AMD_LIB'] = '/nonexistent/libaps_amd.so'\n"
            "import torch\n"
            "from aps_amd import _native\n"
            "from aps_amd.transform import STFT\n"
            "try:\n    STFT(512, 256)(torch.randn(1, 4000))\n"
            "except RuntimeError as e:\n    print('HOST TENSOR:', e)\n"
            "try:\n    _native.load()\n"
            "except _native.NativeLibraryError as e:\n    print('NO LIBRARY:', e)\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True,
                         timeout=300)
    lines = out.stdout.splitlines()
    assert any(l.startswith("HOST TENSOR:") and "no CPU fallback" in l for l in lines), \
        out.stdout + out.stderr
    assert any(l.startswith("NO LIBRARY:") and "no CPU fallback" in l for l in lines), \
        out.stdout + out.stderr


def test_bindings_match_the_header_prototypes():
    """every ctypes signature in aps_amd/_native.py has the parameter list of its prototype in
    include/aps_amd.h: same count, pointers where the header has pointers, the integer / float
    width the header names"""
    import ctypes as C
    from aps_amd import _native
    header = open(os.path.join(ROOT, "include", "aps_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    protos = re.findall(r"\b([a-z_0-9]+\s*\*?)\s*(aps_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header)
    assert len(protos) == len(_native.SIGNATURES)

    def kind(decl: str) -> str:
        decl = decl.strip()
        if "*" in decl:
            return "ptr"
        base = decl.rsplit(None, 1)[0] if " " in decl else decl
        return {"int64_t": "i64", "int32_t": "i32", "int": "i32", "float": "f32"}[base.replace(
            "const ", "").strip()]

    def ckind(t) -> str:
        if t in (C.c_void_p, C.c_char_p) or hasattr(t, "contents") or hasattr(t, "_type_") and \
                isinstance(getattr(t, "_type_"), type):
            return "ptr"
        return {C.c_int64: "i64", C.c_int32: "i32", C.c_int: "i32", C.c_float: "f32"}[t]

    for _, name, params in protos:
        params = params.strip()
        want = [] if params in ("", "void") else [kind(p) for p in params.split(",")]
        _, args = _native.SIGNATURES[name]
        got = [ckind(a) for a in args]
        assert got == want, f"{name}: binding {got} vs header {want}"


def test_weight_gradient_workspace_rule():
    """aps_gemm_tn_workspace: which x^T y products are cut into row slabs (no GPU: a size query).  Few
    output tiles and a long contraction -> slabs (S (I J + I) floats of partials); many tiles -> none"""
    from aps_amd import _native as nat
    lib = nat.load()
    ws = lib.aps_gemm_tn_workspace
    assert ws(160000, 128, 12) == 64 * (128 * 12 + 128) * 4     # the first conv2d layer: 2 tiles, 64 slabs
    assert ws(2016, 512, 512) == 8 * (512 * 512 + 512) * 4       # a conformer projection: 64 tiles, 8 slabs
    assert ws(2016, 2048, 512) == 2 * (2048 * 512 + 2048) * 4
    assert ws(2 ** 20, 2048, 2048) == 0                          # 1024 tiles fill the chip: the plain product
    assert ws(40, 16, 12) == 0 and ws(0, 1, 1) == 0              # a short contraction is one slab
