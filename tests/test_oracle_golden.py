"""
CPU: pins the oracle (oracle/aps_oracle.py) against the golden vectors recorded from the real
reference (tests/golden/make_golden.py).  Oracle and reference run the same torch-CPU ops in the
same order, so the bar here is tight (1e-6 scaled) except where BLAS blocking may differ.
"""
import numpy as np
import pytest
import torch

from oracle import aps_oracle as orc
from tests.conftest import golden, golden_names, assert_close

TIGHT = 2e-6


def test_windows():
    g = golden("windows")
    for key, ref in g.items():
        name, n = key.rsplit("_", 1)
        assert torch.equal(orc.window(name, int(n)), ref), key


def test_dft_basis():
    g = golden("kernels")
    for mode in ["librosa", "kaldi"]:
        for nrm in [0, 1]:
            for inv in [0, 1]:
                K, w = orc.dft_basis(30, orc.window("hamm", 30), True, bool(nrm), bool(inv), mode)
                tag = f"{mode}_n{nrm}_i{inv}"
                assert torch.equal(K, g["K30_" + tag]) and torch.equal(w, g["w30_" + tag]), tag
    K, w = orc.dft_basis(30, orc.window("hann", 30), False, mode="librosa")
    assert torch.equal(K, g["K30_nopow2"]) and torch.equal(w, g["w30_nopow2"])
    for fl, mode in [(400, "librosa"), (400, "kaldi"), (512, "librosa")]:
        K, w = orc.dft_basis(fl, orc.window("sqrthann", fl), mode=mode)
        assert list(K.shape) == g[f"K{fl}_{mode}_shape"].tolist()
        assert torch.equal(K[::37, 0, ::29], g[f"K{fl}_{mode}_probe"])
        assert torch.equal(w, g[f"w{fl}_{mode}"])


def test_num_frames_table():
    tab = golden("num_frames")["table"].tolist()
    for fl, fh, kaldi, center, S, T, L, nb in tab:
        mode = "kaldi" if kaldi else "librosa"
        W = orc.fft_size_of(fl, True, mode)
        width = fl if kaldi else W
        assert width == L and W // 2 + 1 == nb
        assert orc.num_frames(S, width, fh, bool(center)) == T


@pytest.mark.parametrize("name", golden_names("stft_"))
def test_stft(name):
    g = golden(name)
    c = g.cfg
    out = orc.stft(g["wav"], c["frame_len"], c["frame_hop"], c["window"], True, c["normalized"],
                   c["pre_emphasis"], c["onesided"], c["center"], c["mode"], c["polar"])
    if c["polar"]:
        assert_close(out[..., 0], g["out"][..., 0], TIGHT, name + " mag")
        d = (out[..., 1] - g["out"][..., 1]).abs()
        d = torch.minimum(d, (2 * np.pi - d).abs())
        strong = g["out"][..., 0] > 1e-3 * g["out"][..., 0].max()
        assert d[strong].max() < 1e-4
    else:
        assert_close(out, g["out"], TIGHT, name)
    if "inv" in g:
        inv = orc.istft(g["out"], c["frame_len"], c["frame_hop"], c["window"], True,
                        c["normalized"], c["onesided"], c["center"], c["mode"], c["polar"])
        assert_close(inv, g["inv"], TIGHT, name + " inverse")


@pytest.mark.parametrize("name", [n for n in golden_names("asr_") if n != "asr_abs_mel_log_cmvn"])
def test_asr_features(name):
    g = golden(name)
    c = dict(g.cfg)
    kw = dict(feats=c.pop("feats"), frame_len=c.pop("frame_len"), frame_hop=c.pop("frame_hop"),
              window_name=c.pop("window", "hamm"))
    kw.update(c)
    kw.pop("gcmvn", None)
    if "gmean" in g:  # global statistics travel in the fixture (the reference read them from a file)
        kw["gcmvn"] = (g["gmean"], g["gstd"])
    if "mel_filters" in g:
        mel = orc.mel_weights(kw["frame_len"], kw.get("round_pow_of_two", True), None,
                              kw.get("sr", 16000), kw.get("num_mels", 80), kw.get("min_freq", 0),
                              kw.get("max_freq", None), kw.get("mel_coeff_norm", False))
        assert torch.equal(mel, g["mel_filters"])
    for src in ["randn", "egs1"]:
        out = orc.asr_features(g["in_" + src], **kw)
        assert_close(out, g["out_" + src], 1e-5, f"{name}/{src}")


def test_abs_mel_log_cmvn():
    g = golden("asr_abs_mel_log_cmvn")
    out = orc.abs_mel_log_cmvn(g["yr"], g["yi"], g["mel_filters"])
    assert_close(out, g["out"], 1e-5)


@pytest.mark.parametrize("name", golden_names("enh_"))
def test_enh_features(name):
    g = golden(name)
    if name == "enh_mono_decode":
        packed = orc.stft(g["inp"], 512, 256)
        assert_close(packed, g["packed"], TIGHT)
        assert_close(orc.istft(packed, 512, 256), g["wav"], TIGHT)
        assert_close(orc.enh_features(packed, "spectrogram-log-cmvn"), g["feats"], 1e-5)
        return
    c = dict(g.cfg)
    for src in ["randn", "egs3"]:
        packed = orc.stft(g["in_" + src], c["frame_len"], c["frame_hop"],
                          c.get("window", "sqrthann"), center=c.get("center", False))
        assert_close(packed, g["packed_" + src], TIGHT, name)
        T = orc.num_frames(g["in_" + src].shape[-1], orc.fft_size_of(c["frame_len"]),
                           c["frame_hop"], c.get("center", False))
        assert g["len_" + src].tolist() == [T] * packed.shape[0]
        chain = {}
        if "fbank" in c["feats"]:
            chain["mel_w"] = orc.mel_weights(c["frame_len"], num_mels=c.get("num_mels", 80))
        feats = orc.enh_features(g["packed_" + src], c["feats"], c.get("ipd_index", ""),
                                 c.get("cos_ipd", True), c.get("sin_ipd", False),
                                 c.get("ref_channel", 0), **chain)
        assert_close(feats, g["feats_" + src], 1e-5, f"{name}/{src}")


@pytest.mark.parametrize("name", ["mvdr_full", "mvdr_ragged", "mvdr_no_noise_mask"])
def test_mvdr_pieces(name):
    b, g = golden("mvdr_base"), golden(name)
    xr, xi = b["packed"][..., 0], b["packed"][..., 1]
    xl = g.get("x_len")
    mn = None if name == "mvdr_no_noise_mask" else b["mask_n"]
    att = (b["proj_w"], b["proj_b"], b["gvec_w"], b["gvec_b"])
    assert_close(orc.process_mask(b["mask_s"], xl), g["pmask_s"], TIGHT)
    yr, yi, it = orc.mvdr_forward(b["mask_s"], xr, xi, att, mn, xl)
    assert_close(it["Rs"][0], g["Rs_r"], 1e-5)
    assert_close(it["Rs"][1], g["Rs_i"], 1e-5)
    assert_close(it["Rn"][0], g["Rn_r"], 1e-5)
    assert_close(it["u"], g["u"], 1e-5)
    assert_close(it["w"][0].transpose(1, 2), g["w_r"], 1e-4)
    assert_close(it["w"][1].transpose(1, 2), g["w_i"], 1e-4)
    assert_close(yr, g["y_r"], 1e-4)
    assert_close(yi, g["y_i"], 1e-4)


def test_mvdr_nonorm_and_channels():
    b, g = golden("mvdr_base"), golden("mvdr_nonorm")
    att = (b["proj_w"], b["proj_b"], b["gvec_w"], b["gvec_b"])
    yr, yi, _ = orc.mvdr_forward(b["mask_s"], b["packed"][..., 0], b["packed"][..., 1], att,
                                 b["mask_n"], None, mask_norm=False)
    assert_close(yr, g["y_r"], 1e-4)
    assert_close(yi, g["y_i"], 1e-4)
    for C in (2, 6):
        g = golden(f"mvdr_c{C}")
        att = (g["proj_w"], g["proj_b"], g["gvec_w"], g["gvec_b"])
        yr, yi, _ = orc.mvdr_forward(g["mask_s"], g["packed"][..., 0], g["packed"][..., 1], att,
                                     g["mask_n"])
        assert_close(yr, g["y_r"], 1e-4)
        assert_close(yi, g["y_i"], 1e-4)


def test_tf_masking():
    g = golden("tf_masking")
    assert torch.equal(orc.tf_masking(g["packed"], g["rmask"], 1), g["out_real"])
    assert_close(orc.tf_masking(g["packed"], g["cmask"], 0), g["out_cplx"], TIGHT)
    assert torch.equal(orc.tf_masking(g["packed"][:, 2], g["rmask"]), g["out_4d"])


def test_config1_full_size_oracle_matches_reference():
    """the oracle's AsrTransform restatement at BASELINE configs[0]'s own size (batch 8 x 64 000 samples ->
    8 x 397 x 80) against the reference's recorded outputs; first two utterances (the oracle is a dense DFT)"""
    import torch
    from oracle import aps_oracle as orc
    g = golden("cfg1full_fbank_log_cmvn")
    c = dict(g.cfg)
    kw = dict(feats=c.pop("feats"), frame_len=c.pop("frame_len"), frame_hop=c.pop("frame_hop"),
              window_name=c.pop("window"))
    kw.update(c)
    x = g["in_q"].float() / 32768
    out = orc.asr_features(x[:2], **kw)
    assert out.shape == (2, 397, 80)
    assert_close(out, g["out_randn"][:2], 2e-5, "oracle at config 1 size")
    assert_close(orc.asr_features(g["in_egs1"], **kw), g["out_egs1"], 2e-5, "oracle, egs1.wav 4 s")
