"""
GPU parity tests proper: the HIP path (through the C-ABI, via the reference-shaped Python surface)
against (a) the golden vectors recorded from the reference and (b) the CPU oracle on seeded
inputs, plus size-independent properties at the benchmark's full sizes.

Tolerance (north star): bit-exact for framing / indexing / shapes; <= 1e-4 of the activation
scale for STFT / feature / beamformer values (see tests/conftest.py:rel_err).
"""
import numpy as np
import pytest
import torch

from tests.conftest import golden, golden_names, assert_close, assert_as_accurate, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(autouse=True)
def _inference_mode():
    """forward path only: the reference's eval entry points run under no_grad as well"""
    with torch.no_grad():
        yield


@pytest.fixture(scope="module")
def lib(device):
    from aps_amd import _native
    return _native.load()


def _to(dev, *ts):
    return [None if t is None else t.to(dev) for t in ts]


# ------------------------------------------------------------------------------------------
# STFT / iSTFT
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", golden_names("stft_"))
def test_stft_golden(name, device, lib):
    from aps_amd.transform.utils import forward_stft, inverse_stft
    g = golden(name)
    c = g.cfg
    kw = dict(window=c["window"], mode=c["mode"], center=c["center"], normalized=c["normalized"],
              onesided=c["onesided"])
    out = forward_stft(g["wav"].to(device), c["frame_len"], c["frame_hop"],
                       pre_emphasis=c["pre_emphasis"], return_polar=c["polar"], **kw)
    assert out.shape == g["out"].shape  # framing: exact
    if c["polar"]:
        assert_close(out[..., 0], g["out"][..., 0], TOL, name + " mag")
        d = (out[..., 1].cpu() - g["out"][..., 1]).abs()
        d = torch.minimum(d, (2 * np.pi - d).abs())
        strong = g["out"][..., 0] > 1e-2 * g["out"][..., 0].max()
        assert d[strong].max() < 1e-3
    else:
        assert_close(out, g["out"], TOL, name)
    if "inv" in g:
        inv = inverse_stft(g["out"].to(device), c["frame_len"], c["frame_hop"],
                           return_polar=c["polar"], **kw)
        assert inv.shape == g["inv"].shape
        assert_close(inv, g["inv"], TOL, name + " inverse")


def test_stft_module_state_and_frames(device):
    from aps_amd.transform.utils import STFT
    tab = golden("num_frames")["table"].tolist()
    for fl, fh, kaldi, center, S, T, L, nb in tab:
        m = STFT(fl, fh, mode="kaldi" if kaldi else "librosa", center=bool(center))
        assert m.win_length == L and m.num_bins == nb
        lens = torch.tensor([S])
        assert m.num_frames(lens).item() == T
        assert lens.item() == S  # not mutated
    m = STFT(400, 160, mode="kaldi").to(device)
    x = torch.randn(3, 2, 4000, device=device)
    out = m(x)
    assert out.shape == (3, 2, 257, m.num_frames(torch.tensor([4000])).item(), 2)


def test_stft_linearity_and_roundtrip_full_size(device):
    """benchmark size (32 x 4 x 64000): linearity and STFT->iSTFT identity (centre mode)"""
    from aps_amd.transform.utils import forward_stft, inverse_stft
    g = torch.Generator().manual_seed(1)
    x = (0.1 * torch.randn(32, 4, 64000, generator=g)).to(device)
    y = (0.1 * torch.randn(32, 4, 64000, generator=g)).to(device)
    X = forward_stft(x, 512, 256)
    assert X.shape == (32, 4, 257, 249, 2)
    Y = forward_stft(y, 512, 256)
    Z = forward_stft(2.0 * x - 3.0 * y, 512, 256)
    assert_close(Z, 2.0 * X - 3.0 * Y, 1e-5, "linearity")
    # Parseval per frame against the windowed time frames of one sequence
    w = torch.hann_window(512, periodic=True, device=device)**0.5
    fr = x[5, 2].unfold(0, 512, 256) * w  # T x 512
    e_t = (fr**2).sum(-1)
    P = (X[5, 2]**2).sum(-1)  # F x T
    e_f = (2 * P.sum(0) - P[0] - P[-1]) / 512
    assert_close(e_f, e_t, 1e-5, "parseval")
    Xc = forward_stft(x[:, 0], 512, 256, center=True)
    xr = inverse_stft(Xc, 512, 256, center=True)
    n = min(xr.shape[-1], 64000)
    assert_close(xr[:, :n], x[:, 0, :n], 1e-5, "roundtrip")


def test_stft_edge_cases(device):
    from aps_amd.transform.utils import forward_stft
    from oracle import aps_oracle as orc
    # shortest legal signal: exactly one frame; odd hop; non power-of-two DFT; two sided
    for S, fl, fh, kw in [(513, 512, 256, {}), (1000, 512, 255, {}),
                          (900, 200, 80, dict(round_pow_of_two=False)),
                          (2000, 128, 64, dict(onesided=False, window="rect"))]:
        g = torch.Generator().manual_seed(S)
        x = torch.randn(2, S, generator=g)
        ref = orc.stft(x, fl, fh, kw.get("window", "sqrthann"),
                       kw.get("round_pow_of_two", True), onesided=kw.get("onesided", True))
        out = forward_stft(x.to(device), fl, fh, **kw)
        assert out.shape == ref.shape
        assert_close(out, ref, TOL, f"edge {S}/{fl}/{fh}")
    with pytest.raises(RuntimeError):
        forward_stft(torch.randn(1, 100, device=device), 512, 256)
    with pytest.raises(RuntimeError):
        forward_stft(torch.randn(1, 2, 3, 4000, device=device), 512, 256)


@pytest.mark.parametrize("S,fl,fh,kw", [(64000, 512, 256, {}), (9001, 512, 256, dict(center=True)),
                                        (4000, 400, 160, dict(pre_emphasis=0.97)), (1000, 512, 255, {}),
                                        (900, 200, 80, dict(round_pow_of_two=False))])
def test_stft_int16_pcm_equals_the_normalised_float_waveform(device, S, fl, fh, kw):
    """int16 waveforms are PCM: the kernels form sample / 32768 themselves (aps_stft_forward_pcm16) -- what the
    reference's reader does on the host (read_audio(norm=True), aps/io/audio.py:41-44).  The value is exact in fp32,
    so the transform equals the float path's bit for bit: interior frames (4-byte loads of two samples), reflect-padded
    and tail frames (the staged loader), the odd-hop kernel, pre-emphasis, an odd base address; and the CPU oracle's
    transform of the normalised waveform within the tolerance"""
    from aps_amd.transform.utils import forward_stft
    from oracle import aps_oracle as orc
    g = torch.Generator().manual_seed(S + fl)
    pcm = torch.randint(-32768, 32768, (3, S), generator=g, dtype=torch.int16)
    pcm[0, :5] = torch.tensor([-32768, 32767, 0, 1, -1], dtype=torch.int16)
    want = pcm.float() / 32768.0
    out = forward_stft(pcm.to(device), fl, fh, **kw)
    ref = forward_stft(want.to(device), fl, fh, **kw)
    assert out.dtype == torch.float32 and torch.equal(out, ref)
    if "pre_emphasis" not in kw:
        assert_close(out, orc.stft(want, fl, fh, "sqrthann", kw.get("round_pow_of_two", True),
                                   center=kw.get("center", False)), TOL, "int16 PCM vs the oracle")
    # rows that start at an odd sample of a larger buffer (2-byte aligned only: the binding re-packs them)
    view = torch.cat([pcm.reshape(-1), pcm.new_zeros(1)]).to(device)[1:1 + 2 * S].view(2, S)
    assert torch.equal(forward_stft(view, fl, fh, **kw),
                       forward_stft(view.float() / 32768.0, fl, fh, **kw))


def test_enh_and_asr_front_ends_take_int16_pcm(device):
    """the fused STFT + feature launches on int16 PCM (aps_stft_features_pcm16): EnhTransform.encode of 4 channels
    (the frame-major kernel) and the waveform-rooted AsrTransform chain equal the float path bit for bit"""
    from aps_amd.transform import AsrTransform, EnhTransform
    g = torch.Generator().manual_seed(11)
    pcm = torch.randint(-20000, 20000, (3, 4, 16000), generator=g, dtype=torch.int16)
    flt = (pcm.float() / 32768.0).to(device)
    enh = EnhTransform(feats="spectrogram-log-cmvn-ipd", frame_len=512, frame_hop=256, window="sqrthann",
                       ipd_index="0,1;0,2;0,3", cos_ipd=True).to(device)
    p16, _ = enh.encode(pcm.to(device), None)
    assert enh._fused is not None, "encode() did not take the fused launch"
    f16 = enh(p16)
    p32, _ = enh.encode(flt, None)
    f32 = enh(p32)
    assert torch.equal(p16, p32) and torch.equal(f16, f32)
    asr = AsrTransform(feats="fbank-log-cmvn", frame_len=400, frame_hop=160, window="hamm", num_mels=80,
                       pre_emphasis=0.97).to(device).eval()
    a16, _ = asr(pcm[:, 0].to(device), None)
    a32, _ = asr(flt[:, 0], None)
    assert torch.equal(a16, a32)


# ------------------------------------------------------------------------------------------
# AsrTransform / EnhTransform
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", [n for n in golden_names("asr_") if n != "asr_abs_mel_log_cmvn"])
def test_asr_transform_golden(name, device, tmp_path):
    from aps_amd.transform import AsrTransform
    from oracle import aps_oracle as orc
    g = golden(name)
    cfg = dict(g.cfg)
    if "gmean" in g:  # global CMVN statistics are read from a file, as the reference does
        path = str(tmp_path / "gcmvn.pt")
        torch.save([g["gmean"], g["gstd"]], path)
        cfg["gcmvn"] = path
    t = AsrTransform(**cfg).to(device)
    if "mel_filters" in g:
        assert torch.equal(t.transform[4].filters.cpu(), g["mel_filters"])
    for src in ["randn", "egs1"]:
        lens = g.get("len_" + src)
        inp_len = None
        if lens is not None:
            inp_len = torch.tensor([8000, 6000])
        out, n = t(g["in_" + src].to(device), inp_len)
        assert out.shape == g["out_" + src].shape
        if lens is not None:
            assert torch.equal(n.cpu(), lens)
        c = dict(g.cfg)
        kw = dict(feats=c.pop("feats"), frame_len=c.pop("frame_len"), frame_hop=c.pop("frame_hop"),
                  window_name=c.pop("window", "hamm"))
        kw.update(c)
        kw.pop("gcmvn", None)
        if "gmean" in g:
            kw["gcmvn"] = (g["gmean"].double(), g["gstd"].double())
        truth = orc.asr_features(g["in_" + src], dtype=torch.float64, **kw)
        # spectrogram features: hand over |X| behind every output element, so that an error above
        # the tolerance is only accepted on bins below 1e-4 of the spectral peak (hann window, no
        # pre-emphasis, real speech with silence: a few bins ~1e-5 of the frame peak carry ~2e-7
        # ABSOLUTE error in |X| in both implementations, which the log turns into ~2e-4)
        mag = None
        if kw["feats"].startswith("spectrogram"):
            p64 = orc.stft(g["in_" + src], kw["frame_len"], kw["frame_hop"], kw["window_name"],
                           center=kw.get("center", False), pre_emphasis=kw.get("pre_emphasis", 0.97),
                           mode=kw.get("stft_mode", "librosa"), dtype=torch.float64)
            mag = p64.pow(2).sum(-1).sqrt().transpose(-1, -2)
        assert_as_accurate(out, g["out_" + src], truth, TOL, what=f"{name}/{src}", magnitude=mag)


def test_asr_abs_mel_log_cmvn(device):
    from aps_amd.cplx import ComplexTensor
    from aps_amd.transform import AsrTransform
    g = golden("asr_abs_mel_log_cmvn")
    t = AsrTransform(feats="abs-mel-log-cmvn", frame_len=512, frame_hop=256,
                     window="sqrthann").to(device)
    out, _ = t(ComplexTensor(g["yr"].to(device), g["yi"].to(device)), None)
    assert_close(out, g["out"], TOL)


def test_asr_standalone_layers_match_fused(device):
    from aps_amd.transform import AsrTransform
    g = torch.Generator().manual_seed(3)
    x = (0.1 * torch.randn(2, 2, 8000, generator=g)).to(device)
    t = AsrTransform(feats="fbank-log-cmvn", use_power=True).to(device)
    fused, _ = t(x, None)
    y = x
    for layer in t.transform:
        y = layer(y)
    assert fused.shape == y.shape == (2, 2, 47, 80)
    assert_close(y, fused, 1e-4, "layer-by-layer vs fused")


def test_speed_perturb_training(device):
    """SpeedPerturbTransform.train(): seeded like the reference's recorded run, every factor drawn"""
    from aps_amd.transform.asr import SpeedPerturbTransform
    g = golden("speed_perturb_train")
    sp = SpeedPerturbTransform(sr=16000, perturb="0.9,1.0,1.1").to(device).train()
    torch.manual_seed(int(g["seed"]))
    out = sp(g["wav"].to(device))
    assert sp.last_choice.tolist() == g["choice"].tolist()
    assert out.shape == g["out"].shape
    assert_close(out, g["out"], 1e-4, "resampled batch")
    assert torch.equal((out == 0).cpu(), g["out"] == 0)  # the zero padding past each new length
    assert sp.output_length(g["lens"].to(device)).cpu().tolist() == g["out_len"].tolist()
    with pytest.raises(RuntimeError):
        sp(torch.randn(2, 2, 4000, device=device))
    sp.eval()
    x = g["wav"].to(device)
    assert sp(x) is x and sp.output_length(g["lens"]) is g["lens"]


@pytest.mark.parametrize("tag,kwargs", [
    ("zero", dict(p=1.0, time_args=(12, 2), freq_args=(8, 2), mask_zero=True)),
    ("mean", dict(p=1.0, time_args=(40, 1), freq_args=(30, 1), mask_zero=False)),
    ("adaptive", dict(p=1.0, adaptive_args=(0.04, 0.1), time_args=(40, 4), freq_args=(10, 1),
                      mask_zero=True)),
    ("coin", dict(p=0.5, time_args=(12, 1), freq_args=(8, 1), mask_zero=True))])
def test_spec_augment_training(device, tag, kwargs):
    """SpecAugTransform.train(): two consecutive seeded calls give the reference's two outputs"""
    import random
    from aps_amd.transform.asr import SpecAugTransform
    g = golden("spec_augment_train")
    aug = SpecAugTransform(**kwargs).to(device).train()
    x = g[f"{tag}.x"].to(device)
    seed = int(g[f"{tag}.seed"])
    torch.manual_seed(seed)
    random.seed(seed)
    for key in ("y", "y2"):
        y = aug(x)
        want = g[f"{tag}.{key}"]
        assert y.shape == want.shape
        assert_close(y, want, 1e-6, f"{tag} {key}")
        if tag != "mean":
            assert torch.equal(y.cpu(), want)
    assert aug.eval()(x) is x


def test_asr_transform_training_mode(device):
    """the whole token chain perturb-fbank-log-cmvn-aug in training mode, seeded like the reference"""
    import random
    from aps_amd.transform import AsrTransform
    g = golden("train_perturb_aug")
    t = AsrTransform(feats="perturb-fbank-log-cmvn-aug", frame_len=400, frame_hop=160, window="hamm",
                     num_mels=40, speed_perturb="0.9,1.0,1.1", aug_prob=1.0, aug_time_args=(6, 1),
                     aug_freq_args=(8, 2)).to(device).train()
    seed = int(g["seed"])
    torch.manual_seed(seed)
    random.seed(seed)
    feats, n = t(g["wav"].to(device), g["lens"].to(device))
    assert t.transform[0].last_choice.tolist() == g["choice"].tolist()
    assert n.cpu().tolist() == g["num_frames"].tolist()
    assert feats.shape == g["feats"].shape
    assert torch.equal((feats == 0).cpu(), g["feats"] == 0)
    assert_close(feats, g["feats"], 1e-4, "training-mode features")


def test_context_layers_standalone(device):
    """SpliceTransform / DeltaTransform / DiscreteCosineTransform / CmvnTransform variants as
    stand-alone layers on multi-channel features, against the oracle"""
    from aps_amd.transform.asr import (CmvnTransform, DeltaTransform, DiscreteCosineTransform,
                                       SpliceTransform)
    from oracle import aps_oracle as orc
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 37, 24, generator=g)
    xd = x.to(device)
    assert_close(SpliceTransform(3, 2, 2)(xd), orc.splice_transform(x.double(), 3, 2, 2), 1e-6)
    assert_close(SpliceTransform(0, 0, 4)(xd), orc.splice_transform(x.double(), 0, 0, 4), 1e-6)
    assert_close(DeltaTransform(2, 2).to(device)(xd), orc.delta_transform(x.double(), 2, 2), 1e-5)
    assert_close(DeltaTransform(1, 3).to(device)(xd), orc.delta_transform(x.double(), 1, 3), 1e-5)
    assert_close(DeltaTransform(2, 2, True).to(device)(xd[:, 0]),
                 orc.delta_transform(x[:, 0].double(), 2, 2, True), 1e-5)
    assert_close(DiscreteCosineTransform(10, 24, 22).to(device)(xd), orc.dct(x.double(), 10, 22), 1e-5)
    for nm, nv in ((True, True), (True, False), (False, True)):
        out = CmvnTransform(nm, nv, per_band=False, eps=1e-5)(xd)
        assert_close(out, orc.cmvn(x.double(), nm, nv, False, 1e-5), 1e-5, f"all-band {nm} {nv}")


def test_nan_detection(device):
    from aps_amd.transform import AsrTransform
    t = AsrTransform(feats="spectrogram-log-cmvn").to(device)
    x = torch.randn(2, 4000, device=device)
    x[1, 2000] = float("nan")
    with pytest.raises(ValueError):
        t(x, None)
    t.nan_policy = "deferred"
    t(x, None)
    with pytest.raises(ValueError):
        t._nan_guard.flush()
    good, _ = t(torch.randn(2, 4000, device=device), None)
    t._nan_guard.flush()
    assert not torch.isnan(good).any()


@pytest.mark.parametrize("name", golden_names("enh_"))
def test_enh_transform_golden(name, device):
    from aps_amd.transform import EnhTransform
    from oracle import aps_oracle as orc
    g = golden(name)
    if name == "enh_mono_decode":
        t = EnhTransform(feats="spectrogram-log-cmvn", frame_len=512, frame_hop=256).to(device)
        packed, _ = t.encode(g["inp"].to(device), None)
        assert_close(packed, g["packed"], TOL)
        assert_close(t(packed), g["feats"], TOL)
        assert_close(t.decode([packed])[0], g["wav"], TOL)
        # foreign (reference-layout, contiguous) packed tensors are accepted too
        assert_close(t(g["packed"].to(device)), g["feats"], TOL)
        assert_close(t.decode([g["packed"].to(device)])[0], g["wav"], TOL)
        return
    t = EnhTransform(**g.cfg).to(device)
    for src in ["randn", "egs3"]:
        x = g["in_" + src].to(device)
        packed, n = t.encode(x, torch.tensor([x.shape[-1]] * x.shape[0]))
        assert packed.shape == g["packed_" + src].shape
        assert torch.equal(n.cpu(), g["len_" + src])
        assert_close(packed, g["packed_" + src], TOL, f"{name}/{src} packed")
        feats = t(packed)
        assert feats.shape == g["feats_" + src].shape == (x.shape[0], packed.shape[-2], t.feats_dim)
        c = g.cfg
        chain = {}
        if "fbank" in c["feats"]:
            chain["mel_w"] = orc.mel_weights(c["frame_len"], num_mels=c.get("num_mels", 80))
        p64 = orc.stft(g["in_" + src], c["frame_len"], c["frame_hop"], c.get("window", "sqrthann"),
                       center=c.get("center", False), dtype=torch.float64)
        truth = orc.enh_features(p64, c["feats"], c.get("ipd_index", ""), c.get("cos_ipd", True),
                                 c.get("sin_ipd", False), c.get("ref_channel", 0), **chain)
        assert_as_accurate(feats, g["feats_" + src], truth, TOL, what=f"{name}/{src} feats")


# ------------------------------------------------------------------------------------------
# MVDR
# ------------------------------------------------------------------------------------------
def _mvdr_from(g, device, num_bins, att_dim, **kw):
    from aps_amd.asr.filter.mvdr import MvdrBeamformer
    m = MvdrBeamformer(num_bins, att_dim=att_dim, **kw)
    m.ref.proj.weight.data.copy_(g["proj_w"])
    m.ref.proj.bias.data.copy_(g["proj_b"])
    m.ref.gvec.weight.data.copy_(g["gvec_w"])
    m.ref.gvec.bias.data.copy_(g["gvec_b"])
    return m.to(device)


@pytest.mark.parametrize("name", ["mvdr_full", "mvdr_ragged", "mvdr_no_noise_mask"])
def test_mvdr_golden(name, device):
    from aps_amd.asr.filter import mvdr as M
    from aps_amd.cplx import ComplexTensor
    from aps_amd.spectrogram import store_of
    b, g = golden("mvdr_base"), golden(name)
    m = _mvdr_from(b, device, 257, 64)
    packed = b["packed"].to(device)
    ms, mn = b["mask_s"].to(device), b["mask_n"].to(device)
    if name == "mvdr_no_noise_mask":
        mn = None
    xl = g.get("x_len")
    xl = None if xl is None else xl.to(device)
    store = store_of(packed)
    cs, cn, ps, _ = M.covariance(store, ms, mn, xl, True, return_masks=True)
    assert_close(ps, g["pmask_s"], 1e-6, "processed mask")
    assert_close(cs[..., 0], g["Rs_r"], TOL, "Rs real")
    assert_close(cs[..., 1], g["Rs_i"], TOL, "Rs imag")
    assert_close(cn[..., 0], g["Rn_r"], TOL, "Rn real")
    assert_close(cn[..., 1], g["Rn_i"], TOL, "Rn imag")
    u = m.ref.attend(cs)
    assert_close(u, g["u"], TOL, "u")
    w = m.derive_weight(cs, cn, u, m.eps)
    assert_close(w[..., 0], g["w_r"], TOL, "w real")
    assert_close(w[..., 1], g["w_i"], TOL, "w imag")
    y = m(ms, ComplexTensor(packed[..., 0], packed[..., 1]), mask_n=mn, x_len=xl)
    assert y.shape == g["y_r"].shape
    assert_close(y.real, g["y_r"], TOL, "y real")
    assert_close(y.imag, g["y_i"], TOL, "y imag")
    # reference-layout (foreign, contiguous) complex input gives the same answer
    pc = b["packed"].to(device)
    y2 = m(ms, ComplexTensor(pc[..., 0].contiguous(), pc[..., 1].contiguous()), mask_n=mn,
           x_len=xl)
    assert_close(y2.real, g["y_r"], TOL)


def test_mvdr_variants_golden(device):
    from aps_amd.cplx import ComplexTensor
    b, g = golden("mvdr_base"), golden("mvdr_nonorm")
    m = _mvdr_from(b, device, 257, 64, mask_norm=False)
    p = b["packed"].to(device)
    y = m(b["mask_s"].to(device), ComplexTensor(p[..., 0], p[..., 1]),
          mask_n=b["mask_n"].to(device))
    assert_close(y.real, g["y_r"], TOL)
    assert_close(y.imag, g["y_i"], TOL)
    for C_ in (2, 6):
        g = golden(f"mvdr_c{C_}")
        m = _mvdr_from(g, device, 129, 32)
        p = g["packed"].to(device)
        y = m(g["mask_s"].to(device), ComplexTensor(p[..., 0], p[..., 1]),
              mask_n=g["mask_n"].to(device))
        assert_close(y.real, g["y_r"], TOL, f"C={C_}")
        assert_close(y.imag, g["y_i"], TOL, f"C={C_}")


def test_mvdr_functional_surface(device):
    """estimate_covar / beamform / trace keep the reference's ComplexTensor signatures"""
    from aps_amd.asr.filter import mvdr as M
    from aps_amd.cplx import ComplexTensor
    b, g = golden("mvdr_base"), golden("mvdr_full")
    p = b["packed"].to(device)
    x = ComplexTensor(p[..., 0], p[..., 1])
    Rs = M.estimate_covar(g["pmask_s"].to(device), x)
    assert_close(Rs.real, g["Rs_r"], TOL)
    w = ComplexTensor(g["w_r"].to(device), g["w_i"].to(device)).transpose(1, 2)
    y = M.beamform(w, x).transpose(1, 2)
    assert_close(y.real, g["y_r"], TOL)
    tr = M.trace(Rs)
    ref = torch.diagonal(g["Rs_r"], dim1=-2, dim2=-1).sum(-1)
    assert_close(tr.real, ref, TOL)


@pytest.mark.parametrize("mask_norm", [True, False])
@pytest.mark.parametrize("lens", [False, True])
def test_mvdr_reads_the_halves_of_one_mask_estimate_in_place(device, mask_norm, lens):
    """the masks of mvdr.py:132-135 are the two halves of ONE N x T x 2F tensor (th.chunk): the kernels read them
    where they are (mask_ld = 2F) and give the bits of the dense copies; so do masks padded to any other pitch, and
    one half alone (noise mask = 1 - speech mask)"""
    from aps_amd import _native as nat
    from aps_amd.asr.filter import mvdr as M
    torch.manual_seed(17)
    N, C, T, F = 5, 4, 61, 257
    store = torch.randn(N, C, T, F, 2, device=device)
    est = torch.rand(N, T, 2 * F, device=device)
    ms, mn = torch.chunk(est, 2, dim=-1)
    x_len = torch.tensor([T, 40, T, 1, 33], device=device) if lens else None
    assert M._mask_operands(ms, mn, N, T, F)[2] == 2 * F and M._mask_operands(ms.contiguous(), None, N, T, F)[2] == F
    assert M._mask_operands(ms, mn.contiguous(), N, T, F)[2] == F   # mixed pitches: made dense
    bf = M.MvdrBeamformer(F, att_dim=64, mask_norm=mask_norm).to(device)
    bf.eval()
    with torch.no_grad():
        dense = bf.weights_from_masks(store, ms.contiguous(), mn.contiguous(), x_len, return_cov=True)
        place = bf.weights_from_masks(store, ms, mn, x_len, return_cov=True)
        for a, b in zip(dense, place):
            assert torch.equal(a, b)
        one_d = bf.weights_from_masks(store, ms.contiguous(), None, x_len)
        one_p = bf.weights_from_masks(store, ms, None, x_len)
        assert torch.equal(one_d[1], one_p[1])
        wide = torch.rand(N, T, F + 7, device=device)
        odd = wide[..., :F]
        assert torch.equal(bf.weights_from_masks(store, odd, None, x_len)[1],
                           bf.weights_from_masks(store, odd.contiguous(), None, x_len)[1])
        cd = M.covariance(store, ms.contiguous(), mn.contiguous(), x_len, mask_norm=mask_norm, return_masks=True)
        cp = M.covariance(store, ms, mn, x_len, mask_norm=mask_norm, return_masks=True)
        for a, b in zip(cd, cp):
            assert torch.equal(a, b)


def test_config2_against_oracle(device):
    """BASELINE config 2 (EnhTransform + IPD + MVDR) at N=4, full 4-ch / 4 s utterances,
    ragged lengths, against the CPU oracle end to end."""
    from aps_amd.asr.filter.mvdr import MvdrBeamformer
    from aps_amd.cplx import ComplexTensor
    from aps_amd.transform import EnhTransform
    from oracle import aps_oracle as orc
    N = 4
    g = torch.Generator().manual_seed(1)
    x = 0.1 * torch.randn(N, 4, 64000, generator=g)
    g = torch.Generator().manual_seed(2)
    masks = torch.sigmoid(torch.randn(N, 249, 514, generator=g))
    ms, mn = [m.contiguous() for m in torch.chunk(masks, 2, -1)]
    torch.manual_seed(3)
    mvdr = MvdrBeamformer(257, att_dim=512, mask_norm=True)
    att = [p.detach().clone() for p in (mvdr.ref.proj.weight, mvdr.ref.proj.bias,
                                        mvdr.ref.gvec.weight, mvdr.ref.gvec.bias)]
    xl = torch.tensor([249, 249, 200, 200])
    # oracle
    rp = orc.stft(x, 512, 256, "sqrthann")
    rf = orc.enh_features(rp, "spectrogram-log-cmvn-ipd", "0,1;0,2;0,3")
    ryr, ryi, _ = orc.mvdr_forward(ms, rp[..., 0], rp[..., 1], att, mn, xl)
    # HIP path
    t = EnhTransform(feats="spectrogram-log-cmvn-ipd", frame_len=512, frame_hop=256,
                     window="sqrthann", center=False, ipd_index="0,1;0,2;0,3",
                     cos_ipd=True).to(device)
    mvdr = mvdr.to(device)
    packed, n = t.encode(x.to(device), torch.tensor([64000] * N))
    assert packed.shape == (N, 4, 257, 249, 2) and n.tolist() == [249] * N
    feats = t(packed)
    assert feats.shape == (N, 249, 1028)
    y = mvdr(ms.to(device), ComplexTensor(packed[..., 0], packed[..., 1]), mask_n=mn.to(device),
             x_len=xl.to(device))
    assert_close(packed, rp, TOL, "packed")
    p64 = orc.stft(x, 512, 256, "sqrthann", dtype=torch.float64)
    tf = orc.enh_features(p64, "spectrogram-log-cmvn-ipd", "0,1;0,2;0,3")
    assert_as_accurate(feats[..., :257], rf[..., :257], tf[..., :257], TOL, what="log-mag cmvn")
    assert_as_accurate(feats[..., 257:], rf[..., 257:], tf[..., 257:], TOL, what="cos ipd")
    assert_close(y.real, ryr, TOL, "beam real")
    assert_close(y.imag, ryi, TOL, "beam imag")


def test_mvdr_properties_full_size(device):
    """N=32 benchmark size: (i) distortionless constraint w^H Rs u-column scaling -- MVDR output is
    invariant to a common positive scaling of both masks when mask_norm is on; (ii) permuting the
    batch permutes the output; (iii) covariance matrices are Hermitian with real diagonal."""
    from aps_amd.asr.filter import mvdr as M
    from aps_amd.cplx import ComplexTensor
    from aps_amd.transform.utils import forward_stft
    from aps_amd.spectrogram import store_of
    g = torch.Generator().manual_seed(1)
    x = (0.1 * torch.randn(32, 4, 64000, generator=g)).to(device)
    g = torch.Generator().manual_seed(2)
    masks = torch.sigmoid(torch.randn(32, 249, 514, generator=g)).to(device)
    ms, mn = [m.contiguous() for m in torch.chunk(masks, 2, -1)]
    torch.manual_seed(3)
    mv = M.MvdrBeamformer(257, att_dim=512).to(device)
    packed = forward_stft(x, 512, 256)
    cx = ComplexTensor(packed[..., 0], packed[..., 1])
    y = mv(ms, cx, mask_n=mn)
    assert y.shape == (32, 249, 257)
    y2 = mv(0.5 * ms, cx, mask_n=0.25 * mn)
    assert_close(y2.real, y.real, 1e-4, "mask scale invariance")
    perm = torch.randperm(32, device=device)
    pk = packed[perm]
    y3 = mv(ms[perm], ComplexTensor(pk[..., 0], pk[..., 1]), mask_n=mn[perm])
    assert torch.equal(y3.real, y.real[perm]) and torch.equal(y3.imag, y.imag[perm])
    cs, cn = M.covariance(store_of(packed), ms, mn)
    assert torch.equal(cs[..., 0], cs[..., 0].transpose(-1, -2))
    assert torch.equal(cs[..., 1], -cs[..., 1].transpose(-1, -2))
    assert (torch.diagonal(cn[..., 1], dim1=-2, dim2=-1) == 0).all()


def test_tf_masking(device):
    from aps_amd.ops import tf_mask_store
    from aps_amd.spectrogram import packed_view, store_of
    g = golden("tf_masking")
    packed = g["packed"].to(device)
    out = packed_view(tf_mask_store(store_of(packed[:, 1]), g["rmask"].to(device)))
    assert_close(out, g["out_real"], 1e-6)
    out = packed_view(tf_mask_store(store_of(packed[:, 0]), g["cmask"].to(device)))
    assert_close(out, g["out_cplx"], 1e-6)


def test_cpu_tensors_are_refused():
    """the product path has no CPU fallback"""
    from aps_amd.transform import EnhTransform
    t = EnhTransform(feats="spectrogram-log-cmvn")
    with pytest.raises(RuntimeError):
        t.encode(torch.randn(2, 4000), None)


def test_asr_transform_perturb_aug_eval(device):
    """'perturb-fbank-log-cmvn-aug' in eval mode = 'fbank-log-cmvn' (both random layers are the
    identity outside training), against the reference's recorded output"""
    from aps_amd.transform import AsrTransform
    g = golden("perturb_aug_eval")
    t = AsrTransform(feats="perturb-fbank-log-cmvn-aug", frame_len=400, frame_hop=160, window="hamm",
                     num_mels=40, speed_perturb="0.9,1.0,1.1", aug_prob=0.5).eval().to(device)
    with torch.no_grad():
        feats, n = t(g["wav"].to(device), g["lens"].to(device))
    assert n.cpu().tolist() == g["num_frames"].tolist()
    from oracle import aps_oracle as orc
    truth = orc.asr_features(g["wav"], "fbank-log-cmvn", frame_len=400, frame_hop=160,
                             window_name="hamm", num_mels=40, dtype=torch.float64)
    assert_as_accurate(feats, g["feats"], truth, TOL, what="perturb-fbank-log-cmvn-aug (eval)")


@pytest.mark.parametrize("C,S,center,sin,ipd", [(4, 64000, False, False, "0,1;0,2;0,3"),
                                                (4, 9000, True, True, "0,1;2,3;1,3"),
                                                (3, 5000, False, True, "0,1;1,2"),
                                                (2, 3333 * 2, True, False, "1,0")])
def test_enh_encode_fused_with_features_equals_two_launches(device, C, S, center, sin, ipd):
    """EnhTransform.encode() computes the features in the STFT's launch for 2 .. 4 channels
    (stft512_frame_feat_kernel: a wavefront owns one frame of every channel) and forward(packed) hands
    them out: the spectrogram and the features equal the two stand-alone launches' (the same butterflies
    and split: bit for bit for X, the features through the same formulas) and the CPU oracle's; interior
    frames, reflect-padded edges, frame counts that are not a multiple of the frames per wavefront"""
    from aps_amd.transform import EnhTransform
    from oracle import aps_oracle as orc
    g = torch.Generator().manual_seed(C * 7 + S)
    x = 0.1 * torch.randn(3, C, S, generator=g)
    kw = dict(feats="spectrogram-log-cmvn-ipd", frame_len=512, frame_hop=256, window="sqrthann",
              center=center, ipd_index=ipd, cos_ipd=True, sin_ipd=sin)
    fused, plain = EnhTransform(**kw).to(device), EnhTransform(**kw).to(device)
    assert fused.fuse_encode_features is None      # automatic: on for this shape
    plain.fuse_encode_features = False
    xd = x.to(device)
    pf, _ = fused.encode(xd, None)
    assert fused._fused is not None, "encode() did not take the fused launch"
    ff = fused(pf)
    pp, _ = plain.encode(xd, None)
    assert plain._fused is None
    fp = plain(pp)
    assert pf.shape == pp.shape and ff.shape == fp.shape
    assert torch.equal(pf, pp), "the fused launch's spectrogram differs from the stand-alone STFT's"
    assert_close(ff, fp, 2e-6, "fused features vs the stand-alone feature kernel")
    rp = orc.stft(x, 512, 256, "sqrthann", center=center)
    assert_close(pf, rp, TOL, "packed")
    p64 = orc.stft(x, 512, 256, "sqrthann", center=center, dtype=torch.float64)
    rf = orc.enh_features(rp, "spectrogram-log-cmvn-ipd", ipd, sin_ipd=sin)
    tf = orc.enh_features(p64, "spectrogram-log-cmvn-ipd", ipd, sin_ipd=sin)
    assert_as_accurate(ff[..., :257], rf[..., :257], tf[..., :257], TOL, what="log-mag cmvn (fused)")
    assert_as_accurate(ff[..., 257:], rf[..., 257:], tf[..., 257:], TOL, what="ipd (fused)")
    # a second forward() on the same tensor, and a forward() on a tensor encode() did not make, take the
    # stand-alone feature launch
    assert_close(fused(pf), fp, 2e-6)
    assert_close(fused(pp), fp, 2e-6)


def test_enh_layers_called_on_their_own(device):
    """The reference lets a caller run the chain's layers as modules (enh.py:21-143, asr.py:280-303,
    mvdr.py:103-116): PhaseTransform (any axis), IpdTransform on a phase tensor (cos, cos + sin, 3-D
    input), MagnitudeTransform with another axis / eps, RefChannelTransform(-1) inside EnhTransform, and
    MvdrBeamformer._process_mask -- against the reference's formulas in float64"""
    from aps_amd.asr.filter.mvdr import MvdrBeamformer
    from aps_amd.transform import EnhTransform
    from aps_amd.transform.asr import MagnitudeTransform
    from aps_amd.transform.enh import IpdTransform, PhaseTransform
    g = torch.Generator().manual_seed(77)
    packed = torch.randn(2, 3, 33, 17, 2, generator=g)
    pd = packed.to(device)
    pha = PhaseTransform(dim=-1)(pd)
    assert pha.shape == (2, 3, 33, 17)
    want = torch.atan2(packed[..., 1].double(), packed[..., 0].double())
    assert_close(pha, want, 2e-6, "phase, last axis")
    mid = packed.permute(0, 4, 1, 2, 3).contiguous()  # the (re, im) axis second
    assert_close(PhaseTransform(dim=1)(mid.to(device)), want, 2e-6, "phase, axis 1")
    # IPD on the phase N x C x T x F
    p = pha.transpose(-1, -2).contiguous()
    for sin in (False, True):
        ipd = IpdTransform("0,1;2,0;1,2", cos=True, sin=sin)
        got = ipd(p)
        pt = want.transpose(-1, -2).transpose(1, 2)            # N x T x C x F
        dif = pt[..., [0, 2, 1], :] - pt[..., [1, 0, 2], :]
        ref = torch.cos(dif) if not sin else torch.cat([torch.cos(dif), torch.sin(dif)], 2)
        assert got.shape == (2, 17, (6 if sin else 3) * 33)
        assert_close(got, ref.reshape(2, 17, -1), 1e-5, f"ipd from phase, sin={sin}")
    assert_close(IpdTransform("1,0")(p[0]), torch.cos(want[0, 1] - want[0, 0]).T[None], 1e-5, "3-D phase")
    with pytest.raises(NameError):
        IpdTransform("1,0", cos=False)(p)
    # magnitude with an eps and along another axis
    m = MagnitudeTransform(dim=1, eps=1e-3)(mid.to(device))
    assert_close(m, torch.sqrt((mid.double() ** 2).sum(1) + 1e-3), 2e-6, "magnitude, axis 1, eps")
    m = MagnitudeTransform(dim=-1, eps=0.5)(pd)
    assert_close(m, torch.sqrt((packed.double() ** 2).sum(-1) + 0.5), 2e-6, "magnitude, eps")
    # every channel through the magnitude chain (ref_channel < 0, no IPD): N x C x T x F
    t = EnhTransform(feats="spectrogram-log-cmvn", frame_len=64, frame_hop=32, window="hann",
                     ref_channel=-1).to(device)
    x = 0.1 * torch.randn(2, 3, 4000, generator=g)
    pk, _ = t.encode(x.to(device), None)
    feats = t(pk)
    assert feats.shape == (2, 3, pk.shape[-2], 33)
    one = EnhTransform(feats="spectrogram-log-cmvn", frame_len=64, frame_hop=32, window="hann",
                       ref_channel=1).to(device)
    assert_close(feats[:, 1], one(pk), 1e-6, "all channels vs one")
    # _process_mask
    mask = torch.rand(3, 21, 40, generator=g)
    xl = torch.tensor([21, 13, 7])
    for norm in (True, False):
        mv = MvdrBeamformer(40, att_dim=16, mask_norm=norm).to(device)
        got = mv._process_mask(mask.to(device), xl.to(device))
        md = mask.double().masked_fill((torch.arange(21)[None] >= xl[:, None])[..., None], 0)
        if norm:
            md = md / (md.abs().amax(1, keepdim=True) + float(torch.finfo(torch.float32).eps))
        assert_close(got, md.transpose(1, 2), 1e-6, f"_process_mask norm={norm}")
        assert mv._process_mask(None, None) is None
    assert_close(mv._process_mask(mask.to(device), None), mask.transpose(1, 2), 1e-6)


def test_config1_full_size_against_the_reference_fixture(device):
    """BASELINE configs[0] at its own size: AsrTransform fbank-log-cmvn on batch 8 x 4 s (64 000 samples,
    16-bit audio) -> 8 x 397 x 80, against outputs recorded from the reference at that size
    (tests/golden/cfg1full_fbank_log_cmvn.npz), plus egs1.wav cropped to 4 s"""
    from aps_amd.transform import AsrTransform
    from oracle import aps_oracle as orc
    g = golden("cfg1full_fbank_log_cmvn")
    cfg = dict(g.cfg)
    t = AsrTransform(**cfg).to(device)
    assert torch.equal(t.transform[4].filters.cpu(), g["mel_filters"])
    x = g["in_q"].float() / 32768
    out, n = t(x.to(device), torch.tensor([64000] * 8))
    assert out.shape == (8, 397, 80) and n.tolist() == [397] * 8 and torch.equal(n.cpu(), g["len_randn"])
    c = dict(cfg)
    kw = dict(feats=c.pop("feats"), frame_len=c.pop("frame_len"), frame_hop=c.pop("frame_hop"),
              window_name=c.pop("window"))
    kw.update(c)
    truth = orc.asr_features(x[:2], dtype=torch.float64, **kw)
    assert_as_accurate(out[:2], g["out_randn"][:2], truth, TOL, what="config 1, batch 8 x 64000 (first two)")
    assert_close(out, g["out_randn"], TOL, "config 1, batch 8 x 64000")
    out1, _ = t(g["in_egs1"].to(device), None)
    assert_close(out1, g["out_egs1"], TOL, "config 1, egs1.wav 4 s")
