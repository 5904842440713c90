#!/usr/bin/env python
"""
Golden-vector generator.  Runs ONLY in the build container (needs /root/reference).

Imports the real reference (funcwj/aps, read-only, no bytecode written) and records
inputs -> outputs of the hot-path functions as small .npz fixtures next to this script.
Nothing here travels as source of the reference: the fixtures are data only.

Disclosed substitutions (also written to MANIFEST.json):
  * `librosa` and `kaldi_python_io` are not installed.  They are stubbed in sys.modules.
    `librosa.filters.mel` is served by oracle.aps_oracle.librosa_mel_htk, our restatement of
    the published librosa 0.8.1 algorithm -> every fixture that goes through a mel matrix is
    "mel weights: parity unpinned" (the mel matrix used is stored in the fixture).
  * wav files are read with scipy.io.wavfile and scaled by 1/32768 (== soundfile float32
    normalisation the reference uses, aps/io/audio.py:41-44).

Usage:  python tests/golden/make_golden.py
"""
import json
import math
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch as th  # noqa: E402
from scipy.io import wavfile  # noqa: E402

from oracle import aps_oracle as orc  # noqa: E402


def _install_stubs():
    lib = types.ModuleType("librosa")
    fil = types.ModuleType("librosa.filters")

    def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm="slaney"):
        assert htk, "reference always passes htk=True (aps/transform/utils.py:153)"
        if fmax is None:
            fmax = float(sr) / 2
        return orc.librosa_mel_htk(sr, n_fft, n_mels, fmin, fmax, norm == "slaney")

    fil.mel = mel
    lib.filters = fil
    sys.modules["librosa"] = lib
    sys.modules["librosa.filters"] = fil
    kio = types.ModuleType("kaldi_python_io")
    kfn = types.ModuleType("kaldi_python_io.functional")
    kfn.read_kaldi_mat = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
    kio.functional = kfn
    sys.modules["kaldi_python_io"] = kio
    sys.modules["kaldi_python_io.functional"] = kfn


_install_stubs()
sys.path.insert(0, REF)

from aps.transform.utils import (init_window, init_kernel, forward_stft, inverse_stft, STFT,  # noqa: E402
                                 iSTFT)
from aps.transform.asr import FeatureTransform as RefAsrTransform  # noqa: E402
from aps.transform.enh import FeatureTransform as RefEnhTransform  # noqa: E402
from aps.asr.filter.mvdr import MvdrBeamformer, estimate_covar, beamform  # noqa: E402
from aps.cplx import ComplexTensor  # noqa: E402
from aps.sse.base import tf_masking  # noqa: E402

MANIFEST = {
    "generator": "tests/golden/make_golden.py",
    "reference": "funcwj/aps @ /root/reference (read-only)",
    "torch": th.__version__,
    "numpy": np.__version__,
    "substitutions": {
        "librosa.filters.mel": "oracle.aps_oracle.librosa_mel_htk (restated librosa 0.8.1, "
                               "htk=True) -- mel weights parity unpinned",
        "kaldi_python_io": "stub (never called)",
        "audio reading": "scipy.io.wavfile int16 / 32768",
    },
    "files": {},
}


def save(name, desc, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, th.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    MANIFEST["files"][name + ".npz"] = {
        "desc": desc,
        "keys": {k: [str(np.asarray(v).dtype), list(np.shape(v))] for k, v in out.items()},
    }
    print(f"{name}.npz: {os.path.getsize(path) / 1024:.0f} KiB")


def read_wav(name):
    sr, data = wavfile.read(os.path.join(REF, "tests/data/transform", name))
    assert sr == 16000 and data.dtype == np.int16
    data = data.astype(np.float32) / 32768.0
    return data.T.copy() if data.ndim == 2 else data


def gen_windows():
    arrs = {}
    for name in ["bartlett", "hann", "hamm", "blackman", "rect", "sqrthann"]:
        for n in [256, 400, 512]:
            arrs[f"{name}_{n}"] = init_window(name, n)
    save("windows", "init_window(name, n) (aps/transform/utils.py:30-59)", **arrs)


def gen_kernels():
    arrs = {}
    for mode in ["librosa", "kaldi"]:
        for normalized in [False, True]:
            for inverse in [False, True]:
                K, w = init_kernel(30, 10, init_window("hamm", 30), round_pow_of_two=True,
                                   normalized=normalized, inverse=inverse, mode=mode)
                tag = f"{mode}_n{int(normalized)}_i{int(inverse)}"
                arrs["K30_" + tag] = K
                arrs["w30_" + tag] = w
    # non power of two, librosa
    K, w = init_kernel(30, 10, init_window("hann", 30), round_pow_of_two=False, mode="librosa")
    arrs["K30_nopow2"] = K
    arrs["w30_nopow2"] = w
    # full-size probes: shapes + a strided sample of the entries
    for (fl, fh, mode) in [(400, 160, "librosa"), (400, 160, "kaldi"), (512, 256, "librosa")]:
        K, w = init_kernel(fl, fh, init_window("sqrthann", fl), mode=mode)
        arrs[f"K{fl}_{mode}_shape"] = np.array(K.shape)
        arrs[f"K{fl}_{mode}_probe"] = K[::37, 0, ::29]
        arrs[f"w{fl}_{mode}"] = w
    save("kernels", "init_kernel K,w (aps/transform/utils.py:62-112); probe = K[::37,0,::29]",
         **arrs)


STFT_CASES = [
    # tag, source, frame_len, hop, window, mode, center, pre_emph, polar, normalized, onesided
    ("egs1_512_sqrthann", "egs1", 512, 256, "sqrthann", "librosa", False, 0, False, False, True),
    ("egs1_512_sqrthann_center", "egs1", 512, 256, "sqrthann", "librosa", True, 0, False, False,
     True),
    ("egs1_512_sqrthann_polar", "egs1", 512, 256, "sqrthann", "librosa", False, 0, True, False,
     True),
    ("egs1_400_hamm_pe", "egs1", 400, 160, "hamm", "librosa", False, 0.97, False, False, True),
    ("egs1_400_hamm_kaldi_pe", "egs1", 400, 160, "hamm", "kaldi", False, 0.97, False, False, True),
    ("egs1_400_hann_kaldi_center", "egs1", 400, 160, "hann", "kaldi", True, 0, False, False, True),
    ("egs1_256_hann", "egs1", 256, 128, "hann", "librosa", False, 0, False, False, True),
    ("egs1_1024_hamm_center", "egs1", 1024, 256, "hamm", "librosa", True, 0, False, False, True),
    ("egs1_512_norm_twosided", "egs1", 512, 256, "hann", "librosa", False, 0, False, True, False),
    ("egs2_512_sqrthann", "egs2", 512, 256, "sqrthann", "librosa", False, 0, False, False, True),
    ("egs3_512_hann_center", "egs3", 512, 256, "hann", "librosa", True, 0, False, False, True),
    ("randn_512_sqrthann", "randn", 512, 256, "sqrthann", "librosa", False, 0, False, False, True),
    ("randn_400_hamm_pe_center", "randn", 400, 160, "hamm", "librosa", True, 0.97, False, False,
     True),
    ("randn_200_blackman", "randn", 200, 80, "blackman", "librosa", False, 0, False, False, True),
]


def stft_source(src):
    if src == "egs1":
        return th.from_numpy(read_wav("egs1.wav")[4000:12000])[None]  # 1 x S
    if src == "egs2":
        return th.from_numpy(read_wav("egs2.wav")[:, 20000:25000].copy())[None]  # 1 x 5 x S
    if src == "egs3":
        return th.from_numpy(read_wav("egs3.wav")[:, 10000:15000].copy())[None]  # 1 x 4 x S
    g = th.Generator().manual_seed(11)
    return 0.1 * th.randn(2, 3, 4000, generator=g)


def gen_stft():
    for case in STFT_CASES:
        tag, src, fl, fh, wnd, mode, center, pe, polar, normalized, onesided = case
        wav = stft_source(src)
        out = forward_stft(wav, fl, fh, window=wnd, mode=mode, center=center, pre_emphasis=pe,
                           return_polar=polar, normalized=normalized, onesided=onesided)
        arrs = {"wav": wav, "out": out,
                "cfg": np.array(json.dumps(dict(frame_len=fl, frame_hop=fh, window=wnd, mode=mode,
                                                center=center, pre_emphasis=pe, polar=polar,
                                                normalized=normalized, onesided=onesided)))}
        # inverse where the reference supports it (no pre-emphasis)
        if pe == 0 and wav.dim() == 2:
            inv = inverse_stft(out, fl, fh, window=wnd, mode=mode, center=center,
                               return_polar=polar, normalized=normalized, onesided=onesided)
            arrs["inv"] = inv
        save("stft_" + tag, f"forward_stft/inverse_stft on {src} (utils.py:227-360)", **arrs)


def gen_num_frames():
    rows = []
    for (fl, fh, mode) in [(512, 256, "librosa"), (400, 160, "librosa"), (400, 160, "kaldi"),
                           (256, 128, "librosa"), (1024, 256, "librosa")]:
        for center in [False, True]:
            m = STFT(fl, fh, mode=mode, center=center)
            for S in [1025, 16000, 64000, 94010, 129536]:
                n = m.num_frames(th.tensor([S]))
                rows.append([fl, fh, int(mode == "kaldi"), int(center), S, int(n.item()),
                             int(m.win_length), int(m.num_bins)])
    save("num_frames",
         "STFTBase.num_frames (utils.py:653-662); cols: frame_len hop kaldi center S T "
         "win_length num_bins", table=np.array(rows, dtype=np.int64))
    # shape known-answers of tests/python/test_transform.py:102-150 on the full files
    egs1 = th.from_numpy(read_wav("egs1.wav"))[None]
    egs2 = th.from_numpy(read_wav("egs2.wav"))[None]
    t1 = RefAsrTransform(feats="spectrogram-log", frame_len=400, frame_hop=160, use_power=True,
                         pre_emphasis=0.96)
    f1, _ = t1(egs1, None)
    t2 = RefEnhTransform(feats="ipd", frame_len=512, frame_hop=256, ipd_index="0,1;0,2;0,3;0,4")
    p2, _ = t2.encode(egs2, None)
    MANIFEST["shape_known_answers"] = {
        "egs1 129536 samples, 400/160 -> spectrogram-log": list(f1.shape),
        "egs2 5ch 94010 samples, 512/256 -> packed": list(p2.shape),
        "egs2 -> ipd feats": list(t2(p2).shape),
    }


def gen_asr_transform():
    g = th.Generator().manual_seed(0)
    x = 0.1 * th.randn(2, 8000, generator=g)
    egs1 = th.from_numpy(read_wav("egs1.wav")[30000:38000])[None]
    lens = th.tensor([8000, 6000])
    cases = {
        "cfg1_fbank_log_cmvn": dict(feats="fbank-log-cmvn", frame_len=400, frame_hop=160,
                                    window="hamm", round_pow_of_two=True, stft_mode="librosa",
                                    pre_emphasis=0.97, use_power=False, num_mels=80, sr=16000,
                                    norm_per_band=True),
        "spectrogram_log_cmvn": dict(feats="spectrogram-log-cmvn", frame_len=400, frame_hop=160),
        "fbank_log_power_kaldi": dict(feats="fbank-log", frame_len=400, frame_hop=160,
                                      stft_mode="kaldi", use_power=True, num_mels=40,
                                      min_freq=20, max_freq=-400, mel_coeff_norm=True),
        "spectrogram_cmvn_allband": dict(feats="spectrogram-log-cmvn", frame_len=512,
                                         frame_hop=256, window="hann", pre_emphasis=0,
                                         norm_per_band=False, center=True),
        "fbank_log_lower_bound": dict(feats="fbank-log-cmvn", frame_len=400, frame_hop=160,
                                      log_lower_bound=1.0, norm_var=False),
        "mfcc_cmvn_delta": dict(feats="mfcc-cmvn-delta", frame_len=400, frame_hop=160,
                                num_mels=40, num_ceps=13, lifter=22, delta_ctx=2, delta_order=2),
        "fbank_splice_sub3": dict(feats="fbank-log-cmvn-splice", frame_len=400, frame_hop=160,
                                  num_mels=40, lctx=2, rctx=1, subsampling_factor=3),
        "fbank_delta_channel": dict(feats="fbank-log-delta", frame_len=400, frame_hop=160,
                                    num_mels=40, delta_ctx=3, delta_order=1,
                                    delta_as_channel=True),
        "fbank_log_dct20": dict(feats="fbank-log-dct-cmvn", frame_len=400, frame_hop=160,
                                num_mels=40, num_ceps=20, norm_mean=False),
        "fbank_gcmvn": dict(feats="fbank-log-cmvn", frame_len=400, frame_hop=160, num_mels=40,
                            gcmvn="<tmp>"),
    }
    import tempfile
    for tag, kw in cases.items():
        extra = {}
        if kw.get("gcmvn") == "<tmp>":
            gg = th.Generator().manual_seed(77)
            gmean, gstd = th.randn(40, generator=gg) - 5, th.rand(40, generator=gg) + 1
            path = os.path.join(tempfile.mkdtemp(), "gcmvn.pt")
            th.save([gmean, gstd], path)
            kw = dict(kw, gcmvn=path)
            extra = {"gmean": gmean, "gstd": gstd}
        t = RefAsrTransform(**kw)
        if extra:
            kw = dict(kw, gcmvn="")  # the statistics travel inside the fixture, not as a path
        out = {}
        for nm, inp in [("randn", x), ("egs1", egs1)]:
            f, n = t(inp.clone(), lens.clone() if nm == "randn" else None)
            out[f"in_{nm}"] = inp
            out[f"out_{nm}"] = f
            if n is not None:
                out[f"len_{nm}"] = n
        mel = [m for m in t.transform if hasattr(m, "filters")]
        if mel:
            out["mel_filters"] = mel[0].filters.data
        out["cfg"] = np.array(json.dumps(kw))
        out.update(extra)
        save("asr_" + tag, "AsrTransform forward (asr.py:837-1033)" +
             (" [mel weights parity unpinned]" if mel else ""), **out)
    # complex input chain used by EnhASRBase (enh_att.py:92-93)
    t = RefAsrTransform(feats="abs-mel-log-cmvn", frame_len=512, frame_hop=256, window="sqrthann")
    g = th.Generator().manual_seed(7)
    yr, yi = th.randn(2, 30, 257, generator=g), th.randn(2, 30, 257, generator=g)
    f, _ = t(ComplexTensor(yr, yi), None)
    save("asr_abs_mel_log_cmvn", "AsrTransform('abs-mel-log-cmvn') on ComplexTensor "
         "(asr.py:330-332) [mel weights parity unpinned]", yr=yr, yi=yi, out=f,
         mel_filters=t.transform[1].filters.data)


def gen_asr_cfg1_full():
    """BASELINE configs[0] at its full size (SURVEY.md 8d "Config 1"): batch 8 of 4 s at 16 kHz ->
    397 x 80 log-mel frames.  The waveforms are 16-bit audio (0.1 randn quantised to 1 / 32768, stored as
    int16: what a wav file holds), and egs1.wav cropped to 64 000 samples"""
    g = th.Generator().manual_seed(0)
    q = th.round(0.1 * th.randn(8, 64000, generator=g) * 32768).clamp(-32768, 32767).to(th.int16)
    x = q.float() / 32768
    egs1 = th.from_numpy(read_wav("egs1.wav")[:64000].copy())[None]
    kw = dict(feats="fbank-log-cmvn", frame_len=400, frame_hop=160, window="hamm", round_pow_of_two=True,
              stft_mode="librosa", pre_emphasis=0.97, use_power=False, num_mels=80, sr=16000,
              norm_per_band=True)
    t = RefAsrTransform(**kw)
    lens = th.tensor([64000] * 8)
    f, n = t(x.clone(), lens.clone())
    f1, _ = t(egs1.clone(), None)
    save("cfg1full_fbank_log_cmvn", "AsrTransform forward at BASELINE configs[0]'s size: batch 8 x 64 000 "
         "samples -> 8 x 397 x 80 (asr.py:837-1033) [mel weights parity unpinned]",
         in_q=q, out_randn=f, len_randn=n, in_egs1=egs1, out_egs1=f1,
         mel_filters=[m for m in t.transform if hasattr(m, "filters")][0].filters.data,
         cfg=np.array(json.dumps(kw)))


def gen_enh_transform():
    g = th.Generator().manual_seed(1)
    x = 0.1 * th.randn(2, 4, 5000, generator=g)
    egs3 = th.from_numpy(read_wav("egs3.wav")[:, 20000:25000].copy())[None]
    cases = {
        "cfg2_spec_log_cmvn_ipd": dict(feats="spectrogram-log-cmvn-ipd", frame_len=512,
                                       frame_hop=256, window="sqrthann", center=False,
                                       ipd_index="0,1;0,2;0,3", cos_ipd=True),
        "ipd_cos_sin": dict(feats="ipd", frame_len=512, frame_hop=256,
                            ipd_index="1,0;2,0;3,1", cos_ipd=True, sin_ipd=True),
        "fbank_log_ipd_ref2": dict(feats="fbank-log-cmvn-ipd", frame_len=400, frame_hop=160,
                                   window="hann", ref_channel=2, num_mels=40, ipd_index="0,2",
                                   center=True),
        "spectrogram_only": dict(feats="spectrogram-log-cmvn", frame_len=512, frame_hop=256),
    }
    for tag, kw in cases.items():
        t = RefEnhTransform(**kw)
        out = {"cfg": np.array(json.dumps(kw))}
        for nm, inp in [("randn", x), ("egs3", egs3)]:
            packed, n = t.encode(inp, th.tensor([inp.shape[-1]] * inp.shape[0]))
            out[f"in_{nm}"] = inp
            out[f"packed_{nm}"] = packed
            out[f"feats_{nm}"] = t(packed)
            out[f"len_{nm}"] = n
        save("enh_" + tag, "EnhTransform encode/forward (enh.py:387-613)", **out)
    # single channel input & decode round trip
    t = RefEnhTransform(feats="spectrogram-log-cmvn", frame_len=512, frame_hop=256)
    mono = x[:, 0].contiguous()
    packed, _ = t.encode(mono, None)
    wav = t.decode([packed])[0]
    save("enh_mono_decode", "EnhTransform encode->decode on N x S (enh.py:571-593)", inp=mono,
         packed=packed, feats=t(packed), wav=wav)


def gen_mvdr():
    th.manual_seed(3)
    mvdr = MvdrBeamformer(257, att_dim=64, mask_norm=True)
    for p in mvdr.parameters():
        p.requires_grad = False
    g = th.Generator().manual_seed(1)
    x = 0.1 * th.randn(3, 4, 8000, generator=g)
    packed = forward_stft(x, 512, 256, window="sqrthann")
    T = packed.shape[-2]
    g = th.Generator().manual_seed(2)
    masks = th.sigmoid(th.randn(3, T, 514, generator=g))
    ms, mn = th.chunk(masks, 2, -1)
    ms, mn = ms.contiguous(), mn.contiguous()
    cx = ComplexTensor(packed[..., 0], packed[..., 1])
    base = dict(packed=packed, mask_s=ms, mask_n=mn, proj_w=mvdr.ref.proj.weight,
                proj_b=mvdr.ref.proj.bias, gvec_w=mvdr.ref.gvec.weight, gvec_b=mvdr.ref.gvec.bias)
    save("mvdr_base", "shared MVDR inputs: packed STFT, masks, ChannelAttention weights", **base)
    for tag, xl, use_n in [("full", None, True), ("ragged", th.tensor([T, T - 9, 20]), True),
                           ("no_noise_mask", None, False)]:
        pms = mvdr._process_mask(ms, xl)
        pmn = mvdr._process_mask(mn, xl) if use_n else None
        Rs = estimate_covar(pms, cx)
        Rn = estimate_covar(pmn if use_n else 1 - pms, cx)
        u = mvdr.ref(Rs)
        w = mvdr._derive_weight(Rs, Rn, u, eps=mvdr.eps)
        y = mvdr(ms, cx, mask_n=mn if use_n else None, x_len=xl)
        extra = dict(pmask_s=pms, Rs_r=Rs.real, Rs_i=Rs.imag, Rn_r=Rn.real, Rn_i=Rn.imag, u=u,
                     w_r=w.real, w_i=w.imag, y_r=y.real, y_i=y.imag)
        if xl is not None:
            extra["x_len"] = xl
        save("mvdr_" + tag, "MvdrBeamformer pieces + forward (mvdr.py:42-174), att_dim=64, seed 3; "
             "inputs in mvdr_base.npz", **extra)
    # no mask normalisation
    mv2 = MvdrBeamformer(257, att_dim=64, mask_norm=False)
    mv2.load_state_dict(mvdr.state_dict())
    y = mv2(ms, cx, mask_n=mn)
    save("mvdr_nonorm", "MvdrBeamformer(mask_norm=False) forward; inputs in mvdr_base.npz",
         y_r=y.real, y_i=y.imag)
    # 2-channel / 6-channel shapes for the templated solver
    for C in [2, 6]:
        g = th.Generator().manual_seed(20 + C)
        xc = 0.1 * th.randn(2, C, 4000, generator=g)
        pk = forward_stft(xc, 256, 128, window="hann")
        Tc = pk.shape[-2]
        mk = th.sigmoid(th.randn(2, Tc, 258, generator=g))
        a, b = [m.contiguous() for m in th.chunk(mk, 2, -1)]
        th.manual_seed(30 + C)
        mv = MvdrBeamformer(129, att_dim=32)
        y = mv(a, ComplexTensor(pk[..., 0], pk[..., 1]), mask_n=b)
        save(f"mvdr_c{C}", f"MvdrBeamformer forward, C={C}, 256/128 hann", packed=pk, mask_s=a,
             mask_n=b, proj_w=mv.ref.proj.weight, proj_b=mv.ref.proj.bias,
             gvec_w=mv.ref.gvec.weight, gvec_b=mv.ref.gvec.bias, y_r=y.real, y_i=y.imag)


def gen_masking():
    g = th.Generator().manual_seed(5)
    packed = th.randn(2, 3, 129, 20, 2, generator=g)
    rmask = th.rand(2, 129, 20, generator=g)
    cmask = th.randn(2, 129, 20, 2, generator=g)
    save("tf_masking", "tf_masking (sse/base.py:23-47)", packed=packed, rmask=rmask, cmask=cmask,
         out_real=tf_masking(packed, rmask, 1), out_cplx=tf_masking(packed, cmask, 0),
         out_4d=tf_masking(packed[:, 2], rmask))


def gen_encoder():
    from aps.asr.transformer.encoder import TransformerEncoder
    cases = {
        "encoder_xfmr_abs_post": dict(pre_norm=False, output_proj=-1),
        "encoder_xfmr_abs_pre": dict(pre_norm=True, output_proj=96),
    }
    for tag, opt in cases.items():
        th.manual_seed(5)
        enc = TransformerEncoder("xfmr", 40, output_proj=opt["output_proj"], num_layers=2,
                                 proj="conv2d", proj_kwargs={"conv_channels": 16, "num_layers": 2},
                                 pose="abs", pose_kwargs={"dropout": 0},
                                 arch_kwargs={"att_dim": 128, "nhead": 4, "feedforward_dim": 256,
                                              "att_dropout": 0, "ffn_dropout": 0,
                                              "pre_norm": opt["pre_norm"]})
        # non-trivial BatchNorm statistics
        g = th.Generator().manual_seed(9)
        for m in enc.modules():
            if isinstance(m, th.nn.BatchNorm2d):
                m.running_mean.copy_(0.1 * th.randn(m.num_features, generator=g))
                m.running_var.copy_(0.5 + th.rand(m.num_features, generator=g))
                m.weight.data.copy_(0.5 + th.rand(m.num_features, generator=g))
                m.bias.data.copy_(0.1 * th.randn(m.num_features, generator=g))
        enc.eval()
        g = th.Generator().manual_seed(6)
        x = th.randn(3, 50, 40, generator=g)
        lens = th.tensor([50, 41, 30])
        with th.no_grad():
            out_full, _ = enc(x, None)
            out_len, n = enc(x, lens.clone())
        sd = {"sd." + k: v for k, v in enc.state_dict().items() if "num_batches" not in k}
        save(tag, "TransformerEncoder('xfmr', conv2d proj, abs pose) eval forward "
             "(asr/transformer/encoder.py:55-106), 2 layers x 128, 4 heads; keys sd.* = state_dict",
             x=x, lens=lens, out_full=out_full, out_len=out_len, num_frames=n, **sd)


def gen_conformer():
    from aps.asr.transformer.encoder import TransformerEncoder
    th.manual_seed(15)
    enc = TransformerEncoder("cfmr", 40, num_layers=2, proj="conv2d",
                             proj_kwargs={"conv_channels": 16, "num_layers": 2}, pose="rel",
                             pose_kwargs={"dropout": 0, "lradius": 6, "rradius": 9},
                             arch_kwargs={"att_dim": 128, "nhead": 4, "feedforward_dim": 256,
                                          "att_dropout": 0, "ffn_dropout": 0, "kernel_size": 7})
    g = th.Generator().manual_seed(19)
    for m in enc.modules():
        if isinstance(m, (th.nn.BatchNorm2d, th.nn.BatchNorm1d)):
            m.running_mean.copy_(0.1 * th.randn(m.num_features, generator=g))
            m.running_var.copy_(0.5 + th.rand(m.num_features, generator=g))
            m.weight.data.copy_(0.5 + th.rand(m.num_features, generator=g))
            m.bias.data.copy_(0.1 * th.randn(m.num_features, generator=g))
    enc.eval()
    g = th.Generator().manual_seed(16)
    x = th.randn(3, 70, 40, generator=g)
    lens = th.tensor([70, 55, 41])
    with th.no_grad():
        out_full, _ = enc(x, None)
        out_len, n = enc(x, lens.clone())
    sd = {"sd." + k: v for k, v in enc.state_dict().items() if "num_batches" not in k}
    variants = {
        "encoder_cfmr_abs_plain": ("cfmr", "abs", {"macaron": False, "kernel_size": 5}, {}),
        "encoder_cfmr_rel_post": ("cfmr", "rel", {"pre_norm": False, "kernel_size": 5}, {}),
        "encoder_xfmr_rel_pre": ("xfmr", "rel", {"pre_norm": True}, {}),
        # Transformer-XL attention, context windows, the other projections
        "encoder_xfmr_xl_ctx": ("xfmr", "xl", {}, dict(proj="linear", proj_kwargs={}, lctx=2, rctx=1,
                                                      chunk_size=2, num_layers=2)),
        "encoder_cfmr_xl_tie": ("cfmr", "xl", {"kernel_size": 5, "tie": True},
                                dict(proj="conv1d", proj_kwargs={"dim": 32, "num_layers": 2},
                                     num_layers=2)),
        "encoder_xfmr_abs_lctx": ("xfmr", "abs", {}, dict(lctx=3, rctx=0, chunk_size=1)),
    }
    for tag, (arch, pose, kw, top) in variants.items():
        th.manual_seed(21)
        pose_kwargs = {"dropout": 0, "lradius": 5, "rradius": 3} if pose == "rel" else {"dropout": 0}
        top = dict(dict(num_layers=1, proj="conv2d",
                        proj_kwargs={"conv_channels": 8, "num_layers": 2}), **top)
        small = TransformerEncoder(arch, 24, pose=pose, pose_kwargs=pose_kwargs,
                                   arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 96,
                                                "att_dropout": 0, "ffn_dropout": 0, **kw}, **top)
        gg = th.Generator().manual_seed(23)
        for m in small.modules():
            if isinstance(m, (th.nn.BatchNorm2d, th.nn.BatchNorm1d)):
                m.running_mean.copy_(0.1 * th.randn(m.num_features, generator=gg))
                m.running_var.copy_(0.5 + th.rand(m.num_features, generator=gg))
            if isinstance(m, th.nn.GroupNorm):
                m.weight.data.copy_(0.5 + th.rand(m.num_channels, generator=gg))
                m.bias.data.copy_(0.1 * th.randn(m.num_channels, generator=gg))
        small.eval()
        xs = th.randn(2, 45, 24, generator=gg)
        ls = th.tensor([45, 31])
        with th.no_grad():
            o_full, _ = small(xs, None)
            o_len, nn_ = small(xs, ls.clone())
        ssd = {"sd." + k: v for k, v in small.state_dict().items() if "num_batches" not in k}
        save(tag, f"TransformerEncoder('{arch}', pose '{pose}', {kw}) eval forward, 1 layer x 64, "
             "2 heads, rel radius 5/3; keys sd.* = state_dict",
             x=xs, lens=ls, out_full=o_full, out_len=o_len, num_frames=nn_, **ssd)
    save("encoder_cfmr_rel", "TransformerEncoder('cfmr', conv2d proj, rel pose lradius 6 / rradius 9,"
         " kernel 7) eval forward, 2 layers x 128, 4 heads; keys sd.* = state_dict",
         x=x, lens=lens, out_full=out_full, out_len=out_len, num_frames=n, **sd)


def gen_conformer_t100():
    """relative-position encoders at 100 encoder frames (400 input frames through the conv2d
    subsampling: the chime4 conformer's geometry) with 64-wide heads -- the sequence lengths
    64 < T' <= 128 of the short-sequence attention kernel's second form"""
    from aps.asr.transformer.encoder import TransformerEncoder
    cases = {
        "encoder_cfmr_rel_t100": ("cfmr", "rel", {"dropout": 0, "lradius": 20, "rradius": 12},
                                  {"kernel_size": 7}),
        "encoder_xfmr_xl_t100": ("xfmr", "xl", {"dropout": 0}, {}),
    }
    for tag, (arch, pose, pose_kwargs, kw) in cases.items():
        th.manual_seed(171)
        enc = TransformerEncoder(arch, 40, num_layers=2, proj="conv2d",
                                 proj_kwargs={"conv_channels": 8, "num_layers": 2}, pose=pose,
                                 pose_kwargs=pose_kwargs,
                                 arch_kwargs={"att_dim": 128, "nhead": 2, "feedforward_dim": 192,
                                              "att_dropout": 0, "ffn_dropout": 0, **kw})
        g = th.Generator().manual_seed(173)
        for m in enc.modules():
            if isinstance(m, (th.nn.BatchNorm2d, th.nn.BatchNorm1d)):
                m.running_mean.copy_(0.1 * th.randn(m.num_features, generator=g))
                m.running_var.copy_(0.5 + th.rand(m.num_features, generator=g))
        enc.eval()
        x = th.randn(2, 400, 40, generator=g)
        lens = th.tensor([400, 333])
        with th.no_grad():
            out_full, _ = enc(x, None)
            out_len, n = enc(x, lens.clone())
        assert out_full.shape[1] == 100, out_full.shape
        sd = {"sd." + k: v for k, v in enc.state_dict().items() if "num_batches" not in k}
        save(tag, f"TransformerEncoder('{arch}', conv2d proj 8 ch, pose '{pose}' {pose_kwargs}, {kw}) "
             "eval forward, 2 layers x 128, 2 heads of 64, 2 x 400 frames -> 100 encoder frames "
             "(lens 400 / 333); keys sd.* = state_dict",
             x=x, lens=lens, out_full=out_full, out_len=out_len, num_frames=n, **sd)


def gen_joint():
    """data path of EnhASRBase.forward (asr/enh_att.py:83-95) with the encoder-side model of
    asr/ctc.py:113-134 as `asr`, assembled from the reference's own modules"""
    from aps.transform import AsrTransform, EnhTransform
    from aps.asr.filter.mvdr import RNNMaskMvdr
    from aps.asr.ctc import CtcASR
    from aps.cplx import ComplexTensor
    th.manual_seed(31)
    enh_transform = EnhTransform(feats="spectrogram-log-cmvn-ipd", frame_len=512, frame_hop=256,
                                 window="sqrthann", ipd_index="0,1;0,2;0,3", cos_ipd=True)
    asr_transform = AsrTransform(feats="abs-mel-log-cmvn", frame_len=512, frame_hop=256,
                                 window="sqrthann", num_mels=40)
    enh_net = RNNMaskMvdr(257 * 4, num_bins=257, rnn_inp_proj=48, rnn="lstm", num_layers=2,
                          hidden_size=64, dropout=0.0, bidirectional=False, mvdr_att_dim=32,
                          mask_norm=True)
    asr = CtcASR(input_size=40, vocab_size=50, ctc=True, ead=True, enc_type="cfmr",
                 enc_kwargs=dict(num_layers=2, proj="conv2d",
                                 proj_kwargs={"conv_channels": 8, "num_layers": 2}, pose="rel",
                                 pose_kwargs={"dropout": 0, "lradius": 4, "rradius": 4},
                                 arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 96,
                                              "att_dropout": 0, "ffn_dropout": 0,
                                              "kernel_size": 5}))
    mods = th.nn.ModuleDict({"enh_transform": enh_transform, "asr_transform": asr_transform,
                             "enh_net": enh_net, "asr": asr}).eval()
    g = th.Generator().manual_seed(33)
    for m in mods.modules():
        if isinstance(m, (th.nn.BatchNorm2d, th.nn.BatchNorm1d)):
            m.running_mean.copy_(0.1 * th.randn(m.num_features, generator=g))
            m.running_var.copy_(0.5 + th.rand(m.num_features, generator=g))
    # correlated channels: a common source with per-channel delay + noise
    src = th.randn(2, 9000, generator=g)
    wav = th.stack([src[:, d:d + 8000] for d in (0, 3, 7, 12)], 1)
    wav = wav + 0.3 * th.randn(2, 4, 8000, generator=g)
    out = {}
    for tag, x_len in (("full", None), ("ragged", th.tensor([8000, 6500]))):
        with th.no_grad():
            packed, n = enh_transform.encode(wav, None if x_len is None else x_len.clone())
            cstft = ComplexTensor(packed[..., 0], packed[..., 1])
            feats = enh_transform(packed)
            x_enh = enh_net(feats, cstft, inp_len=n)
            asr_feats, _ = asr_transform(x_enh, None)
            enc_out, enc_ctc, enc_len = asr(asr_feats, n)
        out.update({f"{tag}.enh_real": x_enh.real, f"{tag}.enh_imag": x_enh.imag,
                    f"{tag}.asr_feats": asr_feats, f"{tag}.enc_out": enc_out,
                    f"{tag}.enc_ctc": enc_ctc})
        if n is not None:
            out.update({f"{tag}.num_frames": n, f"{tag}.enc_len": enc_len})
    sd = {"sd." + k: v for k, v in mods.state_dict().items() if "num_batches" not in k}
    save("joint_mvdr_cfmr", "EnhASRBase.forward data path: EnhTransform(spectrogram-log-cmvn-ipd) -> "
         "RNNMaskMvdr(lstm 2x64) -> AsrTransform(abs-mel-log-cmvn, 40 mel) -> CtcASR(cfmr rel, "
         "2 x 64, ctc head 50), 2 utts x 4 ch x 8000 samples, full + ragged lengths; sd.* = "
         "state_dict of ModuleDict(enh_transform, asr_transform, enh_net, asr)",
         wav=wav, lens=th.tensor([8000, 6500]), **out, **sd)


def gen_joint_grad():
    """the same reference modules under autograd (eval-mode statistics, ragged lengths): a fixed probe
    of the encoder output and the CTC logits is the loss; grad.* of every parameter of the mask
    estimator, the MVDR front end's attention, the conv2d subsampling, the conformer and the CTC head"""
    from aps.transform import AsrTransform, EnhTransform
    from aps.asr.filter.mvdr import RNNMaskMvdr
    from aps.asr.ctc import CtcASR
    from aps.cplx import ComplexTensor
    fwd = np.load(os.path.join(HERE, "joint_mvdr_cfmr.npz"))
    enh_transform = EnhTransform(feats="spectrogram-log-cmvn-ipd", frame_len=512, frame_hop=256,
                                 window="sqrthann", ipd_index="0,1;0,2;0,3", cos_ipd=True)
    asr_transform = AsrTransform(feats="abs-mel-log-cmvn", frame_len=512, frame_hop=256,
                                 window="sqrthann", num_mels=40)
    enh_net = RNNMaskMvdr(257 * 4, num_bins=257, rnn_inp_proj=48, rnn="lstm", num_layers=2,
                          hidden_size=64, dropout=0.0, bidirectional=False, mvdr_att_dim=32,
                          mask_norm=True)
    asr = CtcASR(input_size=40, vocab_size=50, ctc=True, ead=True, enc_type="cfmr",
                 enc_kwargs=dict(num_layers=2, proj="conv2d",
                                 proj_kwargs={"conv_channels": 8, "num_layers": 2}, pose="rel",
                                 pose_kwargs={"dropout": 0, "lradius": 4, "rradius": 4},
                                 arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 96,
                                              "att_dropout": 0, "ffn_dropout": 0,
                                              "kernel_size": 5}))
    mods = th.nn.ModuleDict({"enh_transform": enh_transform, "asr_transform": asr_transform,
                             "enh_net": enh_net, "asr": asr})
    sd = {k[3:]: th.from_numpy(fwd[k]) for k in fwd.files if k.startswith("sd.")}
    missing, unexpected = mods.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    mods.eval()
    wav, x_len = th.from_numpy(fwd["wav"]), th.from_numpy(fwd["lens"])
    with th.no_grad():
        packed, n = enh_transform.encode(wav, x_len.clone())
        feats = enh_transform(packed)
    cstft = ComplexTensor(packed[..., 0], packed[..., 1])
    x_enh = enh_net(feats, cstft, inp_len=n)
    asr_feats, _ = asr_transform(x_enh, None)
    enc_out, enc_ctc, enc_len = asr(asr_feats, n)
    assert np.allclose(enc_out.detach().numpy(), fwd["ragged.enc_out"], atol=1e-5)
    g = th.Generator().manual_seed(227)
    valid = (th.arange(enc_out.shape[1])[None] < enc_len[:, None])[..., None]
    p_out = th.randn(enc_out.shape, generator=g) * valid
    p_ctc = th.randn(enc_ctc.shape, generator=g) * valid
    loss = (th.where(valid, enc_out, th.zeros_like(enc_out)) * p_out).sum() + \
        (th.where(valid, enc_ctc, th.zeros_like(enc_ctc)) * p_ctc).sum()
    loss.backward()
    grads = {"grad." + k: v.grad for k, v in mods.named_parameters()
             if v.requires_grad and v.grad is not None}
    save("joint_mvdr_cfmr_grad", "joint_mvdr_cfmr.npz (ragged lengths) under autograd: loss = <enc_out, "
         "probe_out> + <enc_ctc, probe_ctc> over the valid encoder frames; grad.* = d loss / d parameter "
         "of ModuleDict(enh_transform, asr_transform, enh_net, asr)", probe_out=p_out, probe_ctc=p_ctc,
         loss=loss, **grads)


def gen_dccrn():
    import aps.sse.bss.dccrn as ref_dccrn
    from aps.sse.bss.dccrn import DCCRN
    from aps.transform.enh import FeatureTransform as RefEnh
    # torch 2.10: th.einsum / th.chunk hand LSTMP.forward a strided view and its `.view(N, T, -1)`
    # (dccrn.py:47) raises; the reference code is left untouched, its input is made contiguous
    orig_forward = ref_dccrn.LSTMP.forward
    ref_dccrn.LSTMP.forward = lambda self, inp: orig_forward(self, inp.contiguous())
    th.manual_seed(51)  # one seed ahead of the loop: the variants draw their weights in this order
    variants = {
        "dccrn_shared": dict(share_decoder=True, non_linear="tanh"),
        "dccrn_split": dict(share_decoder=False, non_linear="sigmoid"),
        # causal convolutions (CasualTruncated) + "cat" connections, complex
        "dccrn_cat_causal": dict(share_decoder=True, non_linear="tanh", connection="cat",
                                 causal_conv=True),
        # real-valued network on the magnitude spectrogram
        "dccrn_real": dict(cplx=False, share_decoder=True, non_linear="sigmoid"),
        "dccrn_real_cat": dict(cplx=False, share_decoder=False, non_linear="relu", connection="cat",
                               causal_conv=True),
    }
    for tag, kw in variants.items():
        enh = RefEnh(feats="spectrogram-log-cmvn", frame_len=64, frame_hop=32, window="hann")
        kw = dict(kw)
        cplx = kw.pop("cplx", True)
        # rnn_resize = last channel count x remaining frequency bins (x 2 halves if complex)
        net = DCCRN(cplx=cplx, K="3,3;3,3;3,3", S="2,1;2,1;2,1", P="1,1,1", O="0,0,0",
                    C="16,32,32", num_spks=2, rnn_hidden=64, rnn_layers=2,
                    rnn_resize=320 if cplx else 160,
                    enh_transform=enh, training_mode="time", **kw)
        g = th.Generator().manual_seed(53)
        for m in net.modules():
            if isinstance(m, th.nn.BatchNorm2d):
                m.running_mean.copy_(0.1 * th.randn(m.num_features, generator=g))
                m.running_var.copy_(0.5 + th.rand(m.num_features, generator=g))
                m.weight.data.copy_(0.5 + th.rand(m.num_features, generator=g))
                m.bias.data.copy_(0.1 * th.randn(m.num_features, generator=g))
        net.eval()
        mix = 0.5 * th.randn(2, 2000, generator=g)
        with th.no_grad():
            wav = net(mix)
            net.training_mode = "freq"
            masks = net(mix)
            stft = net.forward_stft(mix, return_polar=False).transpose(1, 2)  # N x T x F x 2
            pred = net.mask_predict(stft)
        sd = {"sd." + k: v for k, v in net.state_dict().items() if "num_batches" not in k}
        save(tag, f"DCCRN (sse/bss/dccrn.py:139-349) cplx={cplx} {kw}: 3 conv blocks 16/32/32, "
             "(complex) LSTM 2 x 64, 2 speakers, 64/32 hann STFT; forward in time / freq mode + "
             "mask_predict", mix=mix, wav0=wav[0], wav1=wav[1], mask0=masks[0], mask1=masks[1],
             pred=pred, **sd)


def gen_dccrn_train():
    """the reference's DCCRN in train() mode (BatchNorm with batch statistics): the weights of the
    forward fixtures, a fixed linear probe of the outputs as the loss, recorded: the mixture, the
    outputs, the gradient of every parameter and the running statistics after the step"""
    import aps.sse.bss.dccrn as ref_dccrn
    from aps.sse.bss.dccrn import DCCRN
    from aps.transform.enh import FeatureTransform as RefEnh
    orig_forward = ref_dccrn.LSTMP.forward  # (see gen_dccrn)
    ref_dccrn.LSTMP.forward = lambda self, inp: orig_forward(self, inp.contiguous())
    variants = {
        "dccrn_shared": dict(share_decoder=True, non_linear="tanh"),
        "dccrn_split": dict(share_decoder=False, non_linear="sigmoid"),
        "dccrn_cat_causal": dict(share_decoder=True, non_linear="tanh", connection="cat",
                                 causal_conv=True),
        "dccrn_real": dict(cplx=False, share_decoder=True, non_linear="sigmoid"),
        "dccrn_real_cat": dict(cplx=False, share_decoder=False, non_linear="relu", connection="cat",
                               causal_conv=True),
    }
    for tag, kw in variants.items():
        fwd = np.load(os.path.join(HERE, tag + ".npz"))
        enh = RefEnh(feats="spectrogram-log-cmvn", frame_len=64, frame_hop=32, window="hann")
        kw = dict(kw)
        cplx = kw.pop("cplx", True)
        # (the masks' own gradient path, mode "freq", once per mask kind)
        for mode in ["time", "freq"] if tag in ("dccrn_shared", "dccrn_real") else ["time"]:
            net = DCCRN(cplx=cplx, K="3,3;3,3;3,3", S="2,1;2,1;2,1", P="1,1,1", O="0,0,0",
                        C="16,32,32", num_spks=2, rnn_hidden=64, rnn_layers=2,
                        rnn_resize=320 if cplx else 160, enh_transform=enh, training_mode=mode,
                        **kw)
            sd = {k[3:]: th.from_numpy(fwd[k]) for k in fwd.files if k.startswith("sd.")}
            missing, unexpected = net.load_state_dict(sd, strict=False)
            assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
            net.train()
            # LeakyReLU has a kink at 0: an input within rounding distance of it makes the gradient
            # a coin toss between implementations (slope 1 or 0.01 for that element, nothing to
            # compare).  The step is recorded on a mixture whose pre-activations all keep a margin
            # of 2e-6 from the kink (post-BatchNorm units, i.e. ~20 fp32 ulps of the typical
            # value): the forward fixture's mixture if it has it, otherwise the first seeded
            # redraw that does.  The margin found is stored.
            margins = []
            hooks = [m.register_forward_pre_hook(lambda _, a: margins.append(a[0].abs().min().item()))
                     for m in net.modules() if isinstance(m, th.nn.LeakyReLU)]
            state = {k: v.clone() for k, v in net.state_dict().items()}
            mix, attempt = th.from_numpy(fwd["mix"]), 0
            while True:
                del margins[:]
                net.load_state_dict(state)  # (the running statistics of a rejected attempt)
                out = net(mix)
                if min(margins) >= 2e-6:
                    break
                attempt += 1
                mix = 0.5 * th.randn(mix.shape, generator=th.Generator().manual_seed(157 + attempt))
            for h in hooks:
                h.remove()
            print(f"{tag} {mode}: LeakyReLU margin {min(margins):.2e} (attempt {attempt})")
            g = th.Generator().manual_seed(157)
            probes = [th.randn(o.shape, generator=g) for o in out]
            loss = sum((o * p).sum() for o, p in zip(out, probes))
            loss.backward()
            grads = {"grad." + k: v.grad for k, v in net.named_parameters() if v.grad is not None}
            stats = {"stat." + k: v for k, v in net.state_dict().items() if "running_" in k}
            save(f"{tag}_train_{mode}", f"DCCRN {tag} (see {tag}.npz: weights sd.*, input mix) in "
                 f"train() mode, training_mode={mode!r}: outputs out0 / out1, loss = sum_s <out_s, "
                 "probe_s>, grad.* = d loss / d parameter, stat.* = BatchNorm running statistics "
                 "after the forward; mix = the mixture (the forward fixture's, or a redraw that keeps "
                 "every LeakyReLU input >= 2e-6 away from the kink), relu_margin = the smallest "
                 "LeakyReLU |input|", out0=out[0], out1=out[1], probe0=probes[0], probe1=probes[1],
                 loss=loss, mix=mix, relu_margin=th.tensor(min(margins)), **grads, **stats)


def gen_train_grads():
    """gradients of the reference's own modules (autograd enabled, eval-mode statistics, dropout 0) on
    the inputs and weights of the forward fixtures: the transformer decoder, the Transformer-XL encoder
    under a context window, the windowed absolute transformer, the causal conformer layer.  A fixed
    random probe of the (valid) outputs is the loss; recorded: probe, loss, grad.* of every parameter
    and the gradient that reaches the input."""
    from aps.asr.transformer.decoder import TorchTransformerDecoder
    from aps.asr.transformer.encoder import TransformerEncoder
    from aps.asr.transformer.impl import ApsConformerEncoderLayer, ApsMultiheadAttention
    _drop_causal_hints()

    def load(net, fwd):
        sd = {k[3:]: th.from_numpy(fwd[k]) for k in fwd.files if k.startswith("sd.")}
        missing, unexpected = net.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing,
                                                                                           unexpected)
        return net.eval()

    def grads(net):
        return {"grad." + k: v.grad for k, v in net.named_parameters()
                if v.requires_grad and v.grad is not None}

    g = th.Generator().manual_seed(211)
    # ---- decoder (decoder.py:102-186)
    for tag, pre_norm in {"decoder_xfmr_post": False, "decoder_xfmr_pre": True}.items():
        fwd = np.load(os.path.join(HERE, tag + ".npz"))
        dec = load(TorchTransformerDecoder(
            40, pose_kwargs={"dropout": 0}, num_layers=2,
            arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 128, "pre_norm": pre_norm,
                         "att_dropout": 0, "ffn_dropout": 0}), fwd)
        enc_out = th.from_numpy(fwd["enc_out"]).requires_grad_(True)
        enc_len, tgt_pad, tgt_len = (th.from_numpy(fwd[k]) for k in ("enc_len", "tgt_pad", "tgt_len"))
        out = dec(enc_out, enc_len, tgt_pad, tgt_len)
        valid = (th.arange(out.shape[1])[None] < tgt_len[:, None])[..., None]
        probe = th.randn(out.shape, generator=g) * valid
        loss = (th.where(valid, out, th.zeros_like(out)) * probe).sum()
        loss.backward()
        save(tag + "_grad", f"{tag}.npz under autograd: loss = <out_len (valid target positions), probe>; "
             "grad.* = d loss / d parameter, g_enc_out = d loss / d enc_out", probe=probe, loss=loss,
             g_enc_out=enc_out.grad, **grads(dec))
    # ---- encoders (encoder.py:57-106): Transformer-XL under a chunked window, windowed absolute
    variants = {
        "encoder_xfmr_xl_ctx": ("xfmr", "xl", {}, dict(proj="linear", proj_kwargs={}, lctx=2, rctx=1,
                                                      chunk_size=2, num_layers=2)),
        "encoder_xfmr_abs_lctx": ("xfmr", "abs", {}, dict(lctx=3, rctx=0, chunk_size=1)),
    }
    for tag, (arch, pose, kw, top) in variants.items():
        fwd = np.load(os.path.join(HERE, tag + ".npz"))
        top = dict(dict(num_layers=1, proj="conv2d",
                        proj_kwargs={"conv_channels": 8, "num_layers": 2}), **top)
        enc = load(TransformerEncoder(arch, 24, pose=pose, pose_kwargs={"dropout": 0},
                                      arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 96,
                                                   "att_dropout": 0, "ffn_dropout": 0, **kw}, **top), fwd)
        x = th.from_numpy(fwd["x"]).requires_grad_(True)
        out, _ = enc(x, None)  # (no lengths: a padded query whose window is all padding is NaN)
        probe = th.randn(out.shape, generator=g)
        loss = (out * probe).sum()
        loss.backward()
        save(tag + "_grad", f"{tag}.npz under autograd (no lengths): loss = <out_full, probe>; grad.* = "
             "d loss / d parameter, g_x = d loss / d input features", probe=probe, loss=loss,
             g_x=x.grad, **grads(enc))
    # ---- the causal conformer layer (impl.py:432-541, casual_conv1d=True)
    fwd = np.load(os.path.join(HERE, "cfmr_layer_causal.npz"))
    layer = load(ApsConformerEncoderLayer(64, ApsMultiheadAttention(64, 2, dropout=0),
                                          feedforward_dim=96, dropout=0, kernel_size=5,
                                          casual_conv1d=True), fwd)
    src = th.from_numpy(fwd["src"]).requires_grad_(True)
    lens = th.from_numpy(fwd["lens"])
    pad = th.arange(src.shape[0])[None, :] >= lens[:, None]
    out = layer(src, src_key_padding_mask=pad)
    valid = (~pad).transpose(0, 1)[..., None]  # T x N x 1
    probe = th.randn(out.shape, generator=g) * valid
    loss = (th.where(valid, out, th.zeros_like(out)) * probe).sum()
    loss.backward()
    save("cfmr_layer_causal_grad", "cfmr_layer_causal.npz under autograd: loss = <out (valid frames), "
         "probe>; grad.* = d loss / d parameter, g_src = d loss / d src", probe=probe, loss=loss,
         g_src=src.grad, **grads(layer))


def gen_causal_conformer_layer():
    """`casual_conv1d` is an option of the base conformer layer only (impl.py:446): the registered
    cfmr_* classes do not pass it on, so it is pinned at layer level"""
    from aps.asr.transformer.impl import ApsConformerEncoderLayer, ApsMultiheadAttention
    th.manual_seed(41)
    layer = ApsConformerEncoderLayer(64, ApsMultiheadAttention(64, 2, dropout=0),
                                     feedforward_dim=96, dropout=0, kernel_size=5,
                                     casual_conv1d=True).eval()
    g = th.Generator().manual_seed(43)
    bn = layer.convolution[3]
    bn.running_mean.copy_(0.1 * th.randn(64, generator=g))
    bn.running_var.copy_(0.5 + th.rand(64, generator=g))
    src = th.randn(21, 2, 64, generator=g)  # T x N x D
    pad = th.arange(21)[None, :] >= th.tensor([21, 15])[:, None]
    with th.no_grad():
        out = layer(src, src_key_padding_mask=pad)
        conv = layer.conv(src)
    sd = {"sd." + k: v for k, v in layer.state_dict().items() if "num_batches" not in k}
    save("cfmr_layer_causal", "ApsConformerEncoderLayer(64, ApsMultiheadAttention(64, 2), FF 96, "
         "kernel 5, casual_conv1d=True) eval forward (impl.py:432-541) + its conv module alone",
         src=src, lens=th.tensor([21, 15]), out=out, conv=conv, **sd)


def gen_perturb_aug():
    """training-time randomised tokens: eval-mode identity + the frozen resampling filters"""
    from aps.transform.asr import FeatureTransform as RefAsr
    from aps.transform.utils import speed_perturb_filter
    ref = RefAsr(feats="perturb-fbank-log-cmvn-aug", frame_len=400, frame_hop=160, window="hamm",
                 num_mels=40, speed_perturb="0.9,1.0,1.1", aug_prob=0.5).eval()
    g = th.Generator().manual_seed(83)
    wav = 0.1 * th.randn(2, 8000, generator=g)
    lens = th.tensor([8000, 6000])
    with th.no_grad():
        feats, n = ref(wav, lens.clone())
    sd = ref.state_dict()
    save("perturb_aug_eval", "AsrTransform('perturb-fbank-log-cmvn-aug') in eval mode "
         "(asr.py:116-195, 621-684: both layers are the identity) + speed_perturb_filter "
         "(utils.py:159-190) for 16000 -> 14400 / 17600; keys.* = state_dict key names and shapes",
         wav=wav, lens=lens, feats=feats, num_frames=n,
         filt_09=speed_perturb_filter(16000, 14400), filt_11=speed_perturb_filter(16000, 17600),
         **{"shape." + k: th.tensor(list(v.shape)) for k, v in sd.items()})


def gen_spatial():
    """geometry-dependent layers of the enh transform: FixedBeamformer and DfTransform"""
    from aps.transform.enh import DfTransform, FixedBeamformer
    g = th.Generator().manual_seed(97)
    bf = FixedBeamformer(5, 4, 33)
    with th.no_grad():
        bf.real.copy_(th.randn(5, 4, 33, 1, generator=g) * 0.3)
        bf.imag.copy_(th.randn(5, 4, 33, 1, generator=g) * 0.3)
    xr, xi = th.randn(3, 4, 33, 19, generator=g), th.randn(3, 4, 33, 19, generator=g)
    beams = th.tensor([1, 4, 0])
    with th.no_grad():
        all_r, all_i = bf(xr, xi)
        one_r, one_i = bf(xr, xi, beam=2)
        sel_r, sel_i = bf(xr, xi, beam=beams)
        tr_r, tr_i = bf(xr, xi, beam=0, trans=True)
    save("fixed_beamformer", "FixedBeamformer (transform/enh.py:303-384) B=5 C=4 F=33 T=19: all "
         "beams, beam=2, per-utterance beams, beam=0 with trans=True", xr=xr, xi=xi,
         w_real=bf.real, w_imag=bf.imag, beams=beams, all_r=all_r, all_i=all_i, one_r=one_r,
         one_i=one_i, sel_r=sel_r, sel_i=sel_i, tr_r=tr_r, tr_i=tr_i)
    phase = (th.rand(3, 7, 13, 33, generator=g) * 2 - 1) * math.pi
    doa_a, doa_b = th.rand(3, generator=g) * 2 * math.pi, th.rand(3, generator=g) * 2 * math.pi
    known = DfTransform(num_bins=33, num_doas=1)
    pairs = DfTransform(num_bins=33, num_doas=1, af_index="1,4;2,5;3,6;0,2")
    sampled = DfTransform(num_bins=33, num_doas=8, sr=8000, velocity=343)
    with th.no_grad():
        af_known = known(phase, doa_a)
        af_two = pairs(phase, [doa_a, doa_b])
        af_sampled = sampled(phase, doa_a)
        af_single = known(phase[0], doa_a[:1])
    save("df_transform", "DfTransform (transform/enh.py:146-300) geometry 7@, F=33: known DoA "
         "(default pairs), two speakers with af_index 1,4;2,5;3,6;0,2, 8 sampled DoAs at sr=8000 "
         "velocity=343, a 3-D (single utterance) phase", phase=phase, doa_a=doa_a, doa_b=doa_b,
         omega=known.omega, af_known=af_known, af_two=af_two, af_sampled=af_sampled,
         af_single=af_single)


def gen_streaming():
    """frame-by-frame (i)STFT: whole-signal forward, single steps and the cache protocol"""
    from aps.transform.streaming import StreamingSTFT, StreamingiSTFT
    g = th.Generator().manual_seed(101)
    wav = 0.3 * th.randn(2, 2100, generator=g)
    cases = {
        "streaming_512": dict(frame_len=512, frame_hop=256, window="sqrthann", mode="librosa"),
        "streaming_400_librosa": dict(frame_len=400, frame_hop=160, window="hamm", mode="librosa",
                                      normalized=True),
        "streaming_400_kaldi": dict(frame_len=400, frame_hop=160, window="hann", mode="kaldi",
                                    round_pow_of_two=False),
    }
    for tag, cfg in cases.items():
        fwd, inv = StreamingSTFT(**cfg), StreamingiSTFT(**cfg)
        W = fwd.win_length
        with th.no_grad():
            packed = fwd(wav)
            polar = fwd(wav, return_polar=True)
            first = fwd.step(wav[:, :W])
            third_polar = fwd.step(wav[:, 2 * cfg["frame_hop"]:2 * cfg["frame_hop"] + W],
                                   return_polar=True)
            rebuilt = inv(packed)
            rebuilt_polar = inv(polar, return_polar=True)
            inv.reset()
            steps = [inv.step(packed[..., t, :].clone()) for t in range(3)]
            tail = inv.flush()
        save(tag, f"StreamingSTFT / StreamingiSTFT (transform/streaming.py:13-152) {cfg}: forward, "
             "polar forward, step on frames 0 and 2, inverse forward (rect / polar), three inverse "
             "steps after reset() and the flush() that follows them", wav=wav, w=fwd.w,
             packed=packed, polar=polar, first=first, third_polar=third_polar, rebuilt=rebuilt,
             rebuilt_polar=rebuilt_polar, step0=steps[0], step1=steps[1], step2=steps[2],
             tail=tail, win_length=th.tensor(W))


def gen_augment_train():
    """the training-mode behaviour of the randomised tokens, seeded"""
    import random
    from aps.transform.asr import FeatureTransform as RefAsr
    from aps.transform.asr import SpecAugTransform, SpeedPerturbTransform
    g = th.Generator().manual_seed(107)
    wav = 0.1 * th.randn(6, 4000, generator=g)
    lens = th.tensor([4000, 3500, 3000, 2500, 4000, 3999])
    sp = SpeedPerturbTransform(sr=16000, perturb="0.9,1.0,1.1").train()
    th.manual_seed(5)
    with th.no_grad():
        out = sp(wav)
    choice = sp.last_choice.clone()
    assert sorted(set(choice.tolist())) == [0, 1, 2], choice  # every factor is exercised
    save("speed_perturb_train", "SpeedPerturbTransform('0.9,1.0,1.1').train() (asr.py:166-195) after "
         "th.manual_seed(5): output, the drawn choices, output_length of `lens`",
         wav=wav, lens=lens, seed=th.tensor(5), out=out, choice=choice,
         out_len=sp.output_length(lens.clone()))
    cases = {
        "zero": (dict(p=1.0, time_args=(12, 2), freq_args=(8, 2), mask_zero=True), (3, 50, 40)),
        "mean": (dict(p=1.0, time_args=(40, 1), freq_args=(30, 1), mask_zero=False), (2, 2, 30, 23)),
        "adaptive": (dict(p=1.0, adaptive_args=(0.04, 0.1), time_args=(40, 4), freq_args=(10, 1),
                          mask_zero=True), (3, 60, 16)),
        "coin": (dict(p=0.5, time_args=(12, 1), freq_args=(8, 1), mask_zero=True), (2, 50, 40)),
    }
    arrays = {}
    for tag, (kw, shape) in cases.items():
        aug = SpecAugTransform(**kw).train()
        x = th.randn(*shape, generator=g)
        seed = 13 + len(tag)
        th.manual_seed(seed)
        random.seed(seed)
        with th.no_grad():
            y = aug(x)
            y2 = aug(x)  # the generators move on: a second call draws different bands
        arrays.update({f"{tag}.x": x, f"{tag}.y": y, f"{tag}.y2": y2, f"{tag}.seed": th.tensor(seed)})
    save("spec_augment_train", "SpecAugTransform.train() (asr.py:621-684) after th.manual_seed(s) + "
         "random.seed(s), two consecutive calls: zero fill with 2 + 2 bands, mean fill on 4-D input, "
         "adaptive (pm, ps) = (0.04, 0.1), p = 0.5 coin", **arrays)
    ref = RefAsr(feats="perturb-fbank-log-cmvn-aug", frame_len=400, frame_hop=160, window="hamm",
                 num_mels=40, speed_perturb="0.9,1.0,1.1", aug_prob=1.0, aug_time_args=(6, 1),
                 aug_freq_args=(8, 2)).train()
    th.manual_seed(23)
    random.seed(23)
    with th.no_grad():
        feats, n = ref(wav, lens.clone())
    save("train_perturb_aug", "AsrTransform('perturb-fbank-log-cmvn-aug', aug_prob=1).train() "
         "after th.manual_seed(23) + random.seed(23): features, frame counts, the drawn speed "
         "choices", wav=wav, lens=lens, seed=th.tensor(23), feats=feats, num_frames=n,
         choice=ref.transform[0].last_choice)


def gen_checkpoints():
    """the registered joint nets asr@enh_xfmr / asr@enh_att built the way a user gets them: a
    checkpoint directory (train.yaml + best.pt.tar) loaded by the reference's load_checkpoint"""
    import tempfile
    import yaml
    from aps.libs import aps_nnet, aps_transform
    from aps.eval.wrapper import load_checkpoint
    _drop_causal_hints()
    enh_transform = dict(feats="spectrogram-log-cmvn-ipd", frame_len=256, frame_hop=128,
                         window="sqrthann", ipd_index="0,1;0,2", cos_ipd=True)
    asr_transform = dict(feats="abs-mel-log-cmvn", frame_len=256, frame_hop=128, window="sqrthann",
                         num_mels=24, sr=16000)
    enh_kwargs = dict(num_bins=129, rnn_inp_proj=32, rnn="lstm", num_layers=2, hidden_size=64,
                      dropout=0.0, bidirectional=False, mvdr_att_dim=24, mask_norm=True)
    xfmr_enc = dict(num_layers=2, proj="conv2d", proj_kwargs={"conv_channels": 8, "num_layers": 2},
                    pose="abs", pose_kwargs={"dropout": 0},
                    arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 96,
                                 "att_dropout": 0, "ffn_dropout": 0})
    recipes = {
        "checkpoint_enh_xfmr": dict(
            nnet="asr@enh_xfmr", enh_transform=enh_transform, asr_transform=asr_transform,
            nnet_conf=dict(asr_input_size=24, enh_input_size=129 * 3, vocab_size=40, sos=1, eos=2,
                           ctc=True, enh_type="rnn_mask_mvdr", enh_kwargs=enh_kwargs,
                           enc_type="xfmr", dec_type="xfmr", enc_kwargs=xfmr_enc,
                           dec_kwargs=dict(num_layers=2, pose_kwargs={"dropout": 0},
                                           arch_kwargs={"att_dim": 64, "nhead": 2,
                                                        "feedforward_dim": 96, "att_dropout": 0,
                                                        "ffn_dropout": 0}))),
        "checkpoint_enh_att": dict(
            nnet="asr@enh_att", enh_transform=enh_transform, asr_transform=asr_transform,
            nnet_conf=dict(asr_input_size=24, enh_input_size=129 * 3, vocab_size=40, sos=1, eos=2,
                           ctc=False, enh_type="rnn_mask_mvdr", enh_kwargs=enh_kwargs,
                           att_type="mhloc",
                           att_kwargs=dict(att_dim=16, att_head=2, conv_channels=3, loc_context=6),
                           enc_type="pytorch_rnn", enc_proj=48, dec_dim=64,
                           enc_kwargs=dict(rnn="lstm", num_layers=2, hidden=64, dropout=0.0,
                                           bidirectional=True),
                           dec_kwargs=dict(rnn="lstm", num_layers=2, hidden=64, dropout=0.0,
                                           input_feeding=True))),
    }
    g = th.Generator().manual_seed(113)
    src = th.randn(2, 5000, generator=g)
    wav = th.stack([src[:, d:d + 4800] for d in (0, 3, 8)], 1) + 0.3 * th.randn(2, 3, 4800,
                                                                                 generator=g)
    lens = th.tensor([4800, 4000])
    tgt = th.randint(3, 40, (2, 7), generator=g)
    tgt[:, 0] = 1
    tgt_len = th.tensor([7, 5])
    for tag, conf in recipes.items():
        th.manual_seed(117)
        built = aps_nnet(conf["nnet"])(
            enh_transform=aps_transform("enh")(**conf["enh_transform"]),
            asr_transform=aps_transform("asr")(**conf["asr_transform"]), **conf["nnet_conf"])
        for m in built.modules():
            if isinstance(m, (th.nn.BatchNorm2d, th.nn.BatchNorm1d)):
                m.running_mean.copy_(0.1 * th.randn(m.num_features, generator=g))
                m.running_var.copy_(0.5 + th.rand(m.num_features, generator=g))
        with tempfile.TemporaryDirectory() as tmp:
            with open(os.path.join(tmp, "train.yaml"), "w") as f:
                yaml.safe_dump(conf, f)
            th.save({"model_state": built.state_dict(), "epoch": 7}, os.path.join(tmp, "best.pt.tar"))
            stats = load_checkpoint(tmp)
        net = stats["nnet"].eval()
        with th.no_grad():
            dec_out, enc_ctc, enc_len = net(wav, lens.clone(), tgt, tgt_len)
        sd = {"sd." + k: v for k, v in net.state_dict().items()}
        save(tag, f"{conf['nnet']} (asr/enh_att.py:121-220) through eval/wrapper.py:load_checkpoint: "
             "3-ch 4800-sample mixtures, rnn_mask_mvdr front end, teacher-forced forward; cfg = the "
             "train.yaml recipe, sd.* = model_state", cfg=json.dumps(conf), wav=wav, lens=lens,
             tgt=tgt, tgt_len=tgt_len, dec_out=dec_out, enc_ctc=enc_ctc, enc_len=enc_len,
             epoch=th.tensor(stats["epoch"]), accept_raw=th.tensor(int(stats["accept_raw"])), **sd)


def gen_concat_encoder():
    """the classic LAS encoder of the reference's recipes: conv1d (time reduction) + BLSTM, and
    conv2d + LSTM, built by encoder_instance("concat", ...)"""
    from aps.asr.base.encoder import BaseEncoder, encoder_instance
    cases = {
        "concat_conv1d_blstm": {"conv1d": dict(dim=48, num_layers=3, stride=[2, 2, 1],
                                               dilation=[1, 1, 2], kernel=3, norm="BN", dropout=0),
                                "pytorch_rnn": dict(rnn="lstm", num_layers=2, hidden=64, dropout=0.0,
                                                    bidirectional=True)},
        "concat_conv2d_lstm": {"conv2d": dict(channel=[4, 8], num_layers=2, kernel=3, stride=2,
                                              norm="BN"),
                               "pytorch_rnn": dict(rnn="lstm", num_layers=2, hidden=64, dropout=0.0,
                                                   input_proj=32)},
    }
    g = th.Generator().manual_seed(127)
    x = th.randn(3, 61, 40, generator=g)
    lens = th.tensor([61, 50, 37])
    for tag, kwargs in cases.items():
        th.manual_seed(131)
        enc = encoder_instance("concat", 40, 56, kwargs, BaseEncoder).eval()
        for m in enc.modules():
            if isinstance(m, (th.nn.BatchNorm2d, th.nn.BatchNorm1d)):
                m.running_mean.copy_(0.1 * th.randn(m.num_features, generator=g))
                m.running_var.copy_(0.5 + th.rand(m.num_features, generator=g))
        with th.no_grad():
            out, out_len = enc(x, lens.clone())
            out_full, _ = enc(x, None)
        sd = {"sd." + k: v for k, v in enc.state_dict().items() if "num_batches" not in k}
        save(tag, f"encoder_instance('concat', 40, 56, {list(kwargs)}) (asr/base/encoder.py:21-72): "
             "forward with / without lengths; cfg = the enc_kwargs", cfg=json.dumps(kwargs), x=x,
             lens=lens, out=out, out_len=out_len, out_full=out_full, **sd)


def gen_variant_rnn():
    """VariantRNNEncoder: projection + BatchNorm + tanh between BLSTM layers, and the pyramidal
    stack with LayerNorm-style (GroupNorm) normalisation and summed directions"""
    from aps.asr.base.encoder import BaseEncoder, encoder_instance
    cases = {
        "variant_rnn_bn": dict(rnn="lstm", hidden=64, num_layers=3, bidirectional=True, project=48,
                               non_linear="tanh", norm="BN"),
        "variant_rnn_pyramid": dict(rnn="lstm", hidden=64, num_layers=3, bidirectional=True,
                                    project=-1, non_linear="relu", norm="LN", pyramid_stack=True,
                                    add_forward_backward=True),
        "variant_rnn_plain": dict(rnn="lstm", hidden=64, num_layers=2, bidirectional=False,
                                  project=40, non_linear="sigmoid", norm=""),
    }
    g = th.Generator().manual_seed(137)
    x = th.randn(3, 45, 40, generator=g)
    lens = th.tensor([45, 38, 20])
    for tag, kwargs in cases.items():
        th.manual_seed(139)
        enc = encoder_instance("variant_rnn", 40, 56, kwargs, BaseEncoder).eval()
        for m in enc.modules():
            if isinstance(m, th.nn.BatchNorm1d):
                m.running_mean.copy_(0.1 * th.randn(m.num_features, generator=g))
                m.running_var.copy_(0.5 + th.rand(m.num_features, generator=g))
            if isinstance(m, (th.nn.BatchNorm1d, th.nn.GroupNorm)):
                m.weight.data.copy_(0.5 + th.rand(m.weight.shape, generator=g))
                m.bias.data.copy_(0.1 * th.randn(m.bias.shape, generator=g))
        with th.no_grad():
            out, out_len = enc(x, lens.clone())
            out_full, _ = enc(x, None)
        sd = {"sd." + k: v for k, v in enc.state_dict().items() if "num_batches" not in k}
        save(tag, f"encoder_instance('variant_rnn', 40, 56, ...) (asr/base/encoder.py:225-308, "
             "component.py:389-449): forward with / without lengths; cfg = the enc_kwargs",
             cfg=json.dumps(kwargs), x=x, lens=lens, out=out, out_len=out_len, out_full=out_full,
             **sd)


def gen_mask_nonlinear():
    from aps.sse.base import MaskNonLinear
    g = th.Generator().manual_seed(149)
    x3 = 4 * th.randn(2, 30, 17, generator=g)
    x4 = 4 * th.randn(3, 2, 20, 9, generator=g)
    x4[0, 0, 0, :3] = th.tensor([25.0, -30.0, 0.0])
    cases = {"relu_scaled": ("relu", dict(scale=2.0, vmax=3.0)),
             "sigmoid": ("sigmoid", dict()),
             "softplus": ("softplus", dict(vmax=10.0)),
             "tanh_clamped": ("tanh", dict(enable="all", scale=1.5, vmax=1.2, vmin=-0.5)),
             "softmax": ("softmax", dict(scale=1.0, vmin=0.05)),
             "none": ("none", dict(enable="all", vmax=2.0, vmin=-2.0))}
    arrays = {"x3": x3, "x4": x4}
    for tag, (name, kw) in cases.items():
        layer = MaskNonLinear(name, **kw)
        arrays[f"{tag}.y3"] = layer(x3)
        arrays[f"{tag}.y4"] = layer(x4)
    save("mask_nonlinear", "MaskNonLinear (sse/base.py:112-156) on a 3-D and a 4-D input: relu x 2 "
         "clamped at 3, sigmoid, softplus clamped at 10, tanh x 1.5 clamped to [-0.5, 1.2], softmax "
         "over the sources floored at 0.05, identity clamped to [-2, 2]", **arrays)


def gen_att_decoder():
    from aps.asr.base.attention import att_instance
    from aps.asr.base.decoder import TorchRNNDecoder
    cases = {
        "att_decoder_ctx": ("ctx", {"att_dim": 32}, False),
        "att_decoder_dot": ("dot", {"att_dim": 32, "scaled": True}, True),
        "att_decoder_loc": ("loc", {"att_dim": 32, "conv_channels": 4, "loc_context": 5}, False),
        "att_decoder_mhctx": ("mhctx", {"att_dim": 16, "att_head": 3}, False),
        "att_decoder_mhdot": ("mhdot", {"att_dim": 16, "att_head": 4, "scaled": True}, True),
        "att_decoder_mhloc": ("mhloc", {"att_dim": 16, "att_head": 2, "conv_channels": 3,
                                        "loc_context": 4}, False),
        # round 4: the other cells / wrappers of TorchRNNDecoder (decoder.py:18-110): GRU, LayerNormRNN,
        # projected LSTM (the attention then sees the projection's width), one-hot "embedding"
        "att_decoder_gru": ("ctx", {"att_dim": 32}, False, {"rnn": "gru"}),
        "att_decoder_lstm_ln": ("loc", {"att_dim": 32, "conv_channels": 4, "loc_context": 5}, False,
                                {"rnn": "lstm", "add_ln": True}),
        "att_decoder_lstmp": ("dot", {"att_dim": 32, "scaled": True}, True, {"rnn": "lstm", "proj_size": 24}),
        "att_decoder_onehot": ("ctx", {"att_dim": 32}, False, {"rnn": "lstm", "onehot_embed": True}),
        "att_decoder_tanh_ln": ("dot", {"att_dim": 32, "scaled": False}, True,
                                {"rnn": "rnn_tanh", "add_ln": True}),
        "att_decoder_lstmp_ln": ("ctx", {"att_dim": 32}, False,
                                 {"rnn": "lstm", "add_ln": True, "proj_size": 24}),
    }
    only = [a for a in sys.argv[2:] if a in cases] if len(sys.argv) > 2 else None
    for tag, case in cases.items():
        kind, att_kwargs, feeding = case[:3]
        dec_kwargs = dict(case[3]) if len(case) > 3 else {"rnn": "lstm"}
        if only and tag not in only:
            continue
        th.manual_seed(61)
        dec_dim = dec_kwargs["proj_size"] if dec_kwargs.get("proj_size", -1) > 0 else 64
        att = att_instance(kind, 48, dec_dim, **att_kwargs)
        dec = TorchRNNDecoder(48, 30, num_layers=2, hidden=64, dropout=0.0, input_feeding=feeding,
                              **dec_kwargs)
        net = th.nn.ModuleDict({"att_net": att, "decoder": dec}).eval()
        g = th.Generator().manual_seed(63)
        enc_out = th.randn(3, 20, 48, generator=g)
        enc_len = th.tensor([20, 15, 11])
        tgt_pad = th.randint(0, 30, (3, 6), generator=g)
        with th.no_grad():
            att.clear()
            outs, alis = dec(att, enc_out, enc_len, tgt_pad)
            att.clear()
            outs_full, alis_full = dec(att, enc_out, None, tgt_pad)
        sd = {"sd." + k: v for k, v in net.state_dict().items()}
        save(tag, f"TorchRNNDecoder (asr/base/decoder.py:69-218) + '{kind}' attention "
             f"(asr/base/attention.py) {att_kwargs}, input_feeding={feeding}: enc 48, 2 x {dec_kwargs} 64, "
             "vocab 30; teacher-forced forward with / without encoder lengths",
             enc_out=enc_out, enc_len=enc_len, tgt_pad=tgt_pad, outs=outs, alis=alis,
             outs_full=outs_full, alis_full=alis_full, **sd)


def gen_att_decoder_grad():
    """gradients of the reference's RNN attention decoder (what cmd/train_am.py back-propagates through on an
    `att` recipe, aps/asr/base/decoder.py:165-218): the cases of gen_att_decoder (same seeds, same inputs,
    same state dicts -- the forward fixtures att_decoder_<case>.npz hold them), teacher forcing with encoder
    lengths, loss = sum(outs * up): the gradient w.r.t. the encoder output and every parameter"""
    from aps.asr.base.attention import att_instance
    from aps.asr.base.decoder import TorchRNNDecoder
    cases = {
        "ctx": ("ctx", {"att_dim": 32}, False),
        "dot": ("dot", {"att_dim": 32, "scaled": True}, True),
        "loc": ("loc", {"att_dim": 32, "conv_channels": 4, "loc_context": 5}, False),
        "mhctx": ("mhctx", {"att_dim": 16, "att_head": 3}, False),
        "mhdot": ("mhdot", {"att_dim": 16, "att_head": 4, "scaled": True}, True),
        "mhloc": ("mhloc", {"att_dim": 16, "att_head": 2, "conv_channels": 3, "loc_context": 4}, False),
        "gru": ("ctx", {"att_dim": 32}, False, {"rnn": "gru"}),
        "lstm_ln": ("loc", {"att_dim": 32, "conv_channels": 4, "loc_context": 5}, False,
                    {"rnn": "lstm", "add_ln": True}),
        "lstmp": ("dot", {"att_dim": 32, "scaled": True}, True, {"rnn": "lstm", "proj_size": 24}),
        "onehot": ("ctx", {"att_dim": 32}, False, {"rnn": "lstm", "onehot_embed": True}),
        "tanh_ln": ("dot", {"att_dim": 32, "scaled": False}, True, {"rnn": "rnn_tanh", "add_ln": True}),
        "lstmp_ln": ("ctx", {"att_dim": 32}, False, {"rnn": "lstm", "add_ln": True, "proj_size": 24}),
    }
    arrays = {}
    for tag, case in cases.items():
        kind, att_kwargs, feeding = case[:3]
        dec_kwargs = dict(case[3]) if len(case) > 3 else {"rnn": "lstm"}
        th.manual_seed(61)
        dec_dim = dec_kwargs["proj_size"] if dec_kwargs.get("proj_size", -1) > 0 else 64
        att = att_instance(kind, 48, dec_dim, **att_kwargs)
        dec = TorchRNNDecoder(48, 30, num_layers=2, hidden=64, dropout=0.0, input_feeding=feeding,
                              **dec_kwargs)
        net = th.nn.ModuleDict({"att_net": att, "decoder": dec}).train()
        g = th.Generator().manual_seed(63)
        enc_out = th.randn(3, 20, 48, generator=g).requires_grad_(True)
        enc_len = th.tensor([20, 15, 11])
        tgt_pad = th.randint(0, 30, (3, 6), generator=g)
        up = th.randn(3, 6, 30, generator=th.Generator().manual_seed(64))
        att.clear()
        outs, _ = dec(att, enc_out, enc_len, tgt_pad)
        (outs * up).sum().backward()
        arrays[f"{tag}.up"] = up
        arrays[f"{tag}.g.enc_out"] = enc_out.grad
        for name, p in net.named_parameters():
            arrays[f"{tag}.g.{name}"] = th.zeros_like(p) if p.grad is None else p.grad
    save("att_decoder_grads", "gradients of TorchRNNDecoder + attention (asr/base/decoder.py:69-218, "
         "asr/base/attention.py) for the 12 cases of att_decoder_<case>.npz (same seeds: same inputs and "
         "parameters): loss = sum(outs * up) under teacher forcing with encoder lengths; <case>.g.enc_out and "
         "<case>.g.<parameter>", **arrays)


def _drop_causal_hints():
    """torch 2.10 compatibility of the reference's decoder layer (see gen_decoder): swallow the
    `tgt_is_causal` / `memory_is_causal` hints in front of the untouched reference forward"""
    import aps.asr.transformer.decoder as ref_dec
    if getattr(ref_dec.TransformerDncoderLayer, "_hints_dropped", False):
        return
    orig = ref_dec.TransformerDncoderLayer.forward

    def forward(self, tgt, memory, tgt_is_causal=None, memory_is_causal=None, **kwargs):
        return orig(self, tgt, memory, **kwargs)

    ref_dec.TransformerDncoderLayer.forward = forward
    ref_dec.TransformerDncoderLayer._hints_dropped = True


def gen_decoder():
    import aps.asr.transformer.decoder as ref_dec
    from aps.asr.transformer.decoder import TorchTransformerDecoder
    # torch 2.10: nn.TransformerDecoder.forward hands every layer `tgt_is_causal` /
    # `memory_is_causal` hints that the reference's layer (written for torch 1.x,
    # decoder.py:46-52) does not accept; the reference code is left untouched, the two hints are
    # dropped in front of it (the masks themselves are still passed and applied)
    _drop_causal_hints()
    for tag, pre_norm in {"decoder_xfmr_post": False, "decoder_xfmr_pre": True}.items():
        th.manual_seed(71)
        dec = TorchTransformerDecoder(
            40, pose_kwargs={"dropout": 0}, num_layers=2,
            arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 128, "pre_norm": pre_norm,
                         "att_dropout": 0, "ffn_dropout": 0}).eval()
        g = th.Generator().manual_seed(73)
        enc_out = th.randn(3, 17, 64, generator=g)
        enc_len = th.tensor([17, 12, 9])
        tgt_pad = th.randint(0, 40, (3, 9), generator=g)
        tgt_len = th.tensor([9, 7, 4])
        with th.no_grad():
            out_len = dec(enc_out, enc_len, tgt_pad, tgt_len)
            out_full = dec(enc_out, None, tgt_pad, None)
            # step with a prefix of embeddings (the decoding-side call pattern)
            _, emb = dec.step(enc_out.transpose(0, 1), tgt_pad[:, :4])
            step_out, _ = dec.step(enc_out.transpose(0, 1), tgt_pad[:, 4:6], pre_emb=emb,
                                   out_idx=-1)
        sd = {"sd." + k: v for k, v in dec.state_dict().items()}
        save(tag, f"TorchTransformerDecoder (asr/transformer/decoder.py:102-186) pre_norm={pre_norm}: "
             "vocab 40, 2 layers x 64, 2 heads, FF 128; teacher-forced forward with / without "
             "lengths + step(pre_emb, out_idx=-1)", enc_out=enc_out, enc_len=enc_len,
             tgt_pad=tgt_pad, tgt_len=tgt_len, out_len=out_len, out_full=out_full,
             step_out=step_out, **sd)


def gen_xfmr_asr():
    """the reference's encoder-decoder model asr@xfmr (asr/att.py:216-260) end to end, forward and under
    autograd: fbank features -> conv2d projection -> transformer encoder (+ CTC branch) -> transformer
    decoder, teacher forced; outputs, a probe loss over the valid positions and the gradient of every
    parameter"""
    from aps.asr.att import XfmrASR
    from aps.transform import AsrTransform
    _drop_causal_hints()
    th.manual_seed(91)
    arch = {"att_dim": 64, "nhead": 2, "feedforward_dim": 128, "att_dropout": 0, "ffn_dropout": 0}
    net = XfmrASR(
        40, 41, sos=39, eos=39, ctc=True,
        asr_transform=AsrTransform(feats="fbank-log-cmvn", frame_len=400, frame_hop=160,
                                   window="hamm", num_mels=40),
        enc_type="xfmr",
        enc_kwargs=dict(num_layers=2, proj="conv2d", proj_kwargs={"conv_channels": 8, "num_layers": 2},
                        pose="abs", pose_kwargs={"dropout": 0}, arch_kwargs=dict(arch)),
        dec_kwargs=dict(num_layers=2, pose_kwargs={"dropout": 0}, arch_kwargs=dict(arch))).eval()
    g = th.Generator().manual_seed(92)
    for m in net.modules():
        if isinstance(m, th.nn.BatchNorm2d):
            m.running_mean.copy_(0.05 * th.randn(m.num_features, generator=g))
            m.running_var.copy_(0.8 + 0.4 * th.rand(m.num_features, generator=g))
    wav = 0.1 * th.randn(3, 12000, generator=g)
    wav_len = th.tensor([12000, 9000, 7000])
    y = th.randint(0, 40, (3, 7), generator=g)
    y_len = th.tensor([7, 5, 3])
    dec_out, enc_ctc, enc_len = net(wav, wav_len.clone(), y, y_len)
    vd = (th.arange(dec_out.shape[1])[None] < y_len[:, None])[..., None]
    ve = (th.arange(enc_ctc.shape[1])[None] < enc_len[:, None])[..., None]
    p_dec = th.randn(dec_out.shape, generator=g) * vd
    p_ctc = th.randn(enc_ctc.shape, generator=g) * ve
    loss = (th.where(vd, dec_out, th.zeros_like(dec_out)) * p_dec).sum() + \
        (th.where(ve, enc_ctc, th.zeros_like(enc_ctc)) * p_ctc).sum()
    loss.backward()
    grads = {"grad." + k: v.grad for k, v in net.named_parameters()
             if v.requires_grad and v.grad is not None}
    sd = {"sd." + k: v for k, v in net.state_dict().items() if "num_batches" not in k}
    save("xfmr_asr", "asr@xfmr (asr/att.py:216-260): AsrTransform(fbank-log-cmvn, 40 mel) -> TransformerEncoder"
         "(xfmr abs, conv2d 8 x 2, 2 x 64) + CTC head 41 -> TorchTransformerDecoder(2 x 64), 3 utterances "
         "(12000 / 9000 / 7000 samples), 7 target tokens; dec_out / enc_ctc / enc_len = forward(wav, wav_len, "
         "y, y_len), loss = <dec_out, probe_dec> + <enc_ctc, probe_ctc> over the valid positions, grad.* = "
         "d loss / d parameter; sd.* = state_dict", wav=wav, wav_len=wav_len, y=y, y_len=y_len,
         dec_out=dec_out, enc_ctc=enc_ctc, enc_len=enc_len, probe_dec=p_dec, probe_ctc=p_ctc, loss=loss,
         **grads, **sd)


def gen_decoder_memory_mask():
    """the reference's decoder layer called with a `memory_mask` (decoder.py:51, 85: handed to the
    cross attention as attn_mask) -- its own decoder never passes one, so the layer is driven
    directly: a boolean mask (a diagonal band of the encoder frames) and an additive float one"""
    from aps.asr.transformer.decoder import TorchTransformerDecoder
    from aps.asr.transformer.utils import prep_sub_mask
    _drop_causal_hints()
    for tag, pre_norm in {"decoder_layer_memmask_post": False, "decoder_layer_memmask_pre": True}.items():
        th.manual_seed(71)
        dec = TorchTransformerDecoder(
            40, pose_kwargs={"dropout": 0}, num_layers=2,
            arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 128, "pre_norm": pre_norm,
                         "att_dropout": 0, "ffn_dropout": 0}).eval()
        layer = dec.decoder.layers[1]
        g = th.Generator().manual_seed(79)
        T, S, N = 9, 17, 3
        tgt, memory = th.randn(T, N, 64, generator=g), th.randn(S, N, 64, generator=g)
        tgt_len, mem_len = th.tensor([9, 7, 4]), th.tensor([17, 12, 9])
        tpad = th.arange(T)[None] >= tgt_len[:, None]
        mpad = th.arange(S)[None] >= mem_len[:, None]
        centre = (th.arange(T)[:, None] * (S - 1) / (T - 1)).round()
        band = (th.arange(S)[None] - centre).abs() > 4  # True = not visible
        bias = th.where(band, th.tensor(float("-inf")), 0.3 * th.randn(T, S, generator=g))
        with th.no_grad():
            out_bool = layer(tgt, memory, tgt_mask=prep_sub_mask(T), memory_mask=band,
                             tgt_key_padding_mask=tpad, memory_key_padding_mask=mpad)
            out_float = layer(tgt, memory, tgt_mask=prep_sub_mask(T), memory_mask=bias,
                              tgt_key_padding_mask=None, memory_key_padding_mask=None)
        # round 5: the same additive-mask call under autograd: loss = sum(out * up)
        up = th.randn(T, N, 64, generator=th.Generator().manual_seed(80))
        tg, mg = tgt.clone().requires_grad_(True), memory.clone().requires_grad_(True)
        layer.zero_grad()
        (layer(tg, mg, tgt_mask=prep_sub_mask(T), memory_mask=bias) * up).sum().backward()
        grads = {"g." + k: v.grad for k, v in layer.named_parameters()}
        sd = {"sd." + k: v for k, v in layer.state_dict().items()}
        save(tag, f"TransformerDncoderLayer (decoder.py:46-99) pre_norm={pre_norm} called with a "
             "memory_mask: 64 wide, 2 heads, tgt 9 x 3, memory 17 x 3; out_bool = boolean band mask "
             "+ both padding masks, out_float = additive mask, no padding; sd.* = the layer; up, g_tgt, "
             "g_memory, g.* = the gradients of sum(out_float * up) w.r.t. both inputs and every parameter",
             tgt=tgt, memory=memory, tgt_len=tgt_len, mem_len=mem_len, band=band, bias=bias,
             out_bool=out_bool, out_float=out_float, up=up, g_tgt=tg.grad, g_memory=mg.grad, **grads, **sd)


def gen_tasks():
    """Task-side consumers recorded from the reference's own task classes:
    LinearFreqSaTask / MelFreqSaTask (aps/task/sse.py:207-455) on a stub network that returns
    fixed masks, MlEnhTask (aps/task/ml.py:64-122) on a stub network that returns fixed (obs, ms)"""
    import torch.nn as nn
    from aps.task.sse import LinearFreqSaTask, MelFreqSaTask
    from aps.task.ml import MlEnhTask
    g = th.Generator().manual_seed(77)
    N, S, C = 3, 6000, 2
    enh = RefEnhTransform(feats="spectrogram-log-cmvn", frame_len=512, frame_hop=256,
                          window="sqrthann")
    mix = 0.1 * th.randn(N, C, S, generator=g)
    refs = [0.1 * th.randn(N, S, generator=g) for _ in range(2)]
    T = (S - 512) // 256 + 1
    masks = [th.sigmoid(th.randn(N, 257, T, generator=g)) for _ in range(2)]

    class MaskNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.enh_transform = enh

        def forward(self, mix):
            return masks

    egs = {"mix": mix, "ref": refs}
    out = {"mix": mix, "ref0": refs[0], "ref1": refs[1], "mask0": masks[0], "mask1": masks[1]}
    cases = {
        "linear_l2": (LinearFreqSaTask, dict()),
        "linear_l1_psa": (LinearFreqSaTask, dict(objf="L1", phase_sensitive=True, truncated=1.0)),
        "linear_fixed_order": (LinearFreqSaTask, dict(permute=False, weight="0.7,0.3")),
        "mel_log": (MelFreqSaTask, dict(num_mels=40, mel_log=True, power_mag=True, mel_scale=2)),
    }
    for tag, (cls, kw) in cases.items():
        task = cls(MaskNet(), **kw)
        with th.no_grad():
            out["loss." + tag] = task(egs)["loss"]
        if tag == "mel_log":
            out["mel"] = task.mel[..., 0]
    out["cfg"] = np.array(json.dumps({k: v[1] for k, v in cases.items()}))
    save("task_freq_sa", "LinearFreqSaTask / MelFreqSaTask.forward(egs)['loss'] (task/sse.py:207-455) "
         "for fixed masks; 2-channel mixture (channel 0 is the reference), 2 speakers", **out)
    # ---- MlEnhTask
    Fb, Tm, Cm = 33, 40, 4
    obs_r, obs_i = th.randn(2, Cm, Fb, Tm, generator=g), th.randn(2, Cm, Fb, Tm, generator=g)
    ms = th.sigmoid(th.randn(2, Tm, Fb, generator=g))

    class MlNet(nn.Module):
        def forward(self, mix):
            return ComplexTensor(obs_r, obs_i), ms

    task = MlEnhTask(MlNet())
    with th.no_grad():
        loss = task({"mix": None})["loss"]
        lp = task.log_pdf(ms.transpose(-1, -2), ComplexTensor(obs_r, obs_i).transpose(1, 2))
    save("task_enh_ml", "MlEnhTask (task/ml.py:64-122): loss and log_pdf(ms, obs) for a fixed "
         "4-channel observation (F = 33, T = 40) and speech mask", obs_r=obs_r, obs_i=obs_i, ms=ms,
         loss=loss, log_pdf=lp)


if __name__ == "__main__":
    th.set_num_threads(4)
    if len(sys.argv) > 1:
        # subset run (e.g. `make_golden.py gen_dccrn`): only these generators, their entries are
        # merged into the existing MANIFEST.json
        for name in sys.argv[1:]:
            if name.startswith("gen_"):  # anything else is an argument of a generator
                globals()[name]()
        path = os.path.join(HERE, "MANIFEST.json")
        old = json.load(open(path))
        old["files"].update(MANIFEST["files"])
        with open(path, "w") as f:
            json.dump(old, f, indent=1)
        print("done (subset)")
        sys.exit(0)
    gen_windows()
    gen_kernels()
    gen_stft()
    gen_num_frames()
    gen_asr_transform()
    gen_asr_cfg1_full()
    gen_enh_transform()
    gen_mvdr()
    gen_masking()
    gen_encoder()
    gen_conformer()
    gen_conformer_t100()
    gen_joint()
    gen_joint_grad()
    gen_dccrn()
    gen_dccrn_train()
    gen_decoder()
    gen_decoder_memory_mask()
    gen_xfmr_asr()
    gen_causal_conformer_layer()
    gen_train_grads()
    gen_att_decoder()
    gen_att_decoder_grad()
    gen_perturb_aug()
    gen_spatial()
    gen_streaming()
    gen_augment_train()
    gen_checkpoints()
    gen_concat_encoder()
    gen_variant_rnn()
    gen_mask_nonlinear()
    gen_tasks()
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(MANIFEST, f, indent=1)
    print("done")
