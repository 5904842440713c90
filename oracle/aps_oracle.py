"""
CPU oracle for the aps front-end hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a functional restatement (plain torch-CPU / numpy, fp32, same operation order)
of the reference algorithms for the path  waveform -> framed STFT -> (|X|, mel, log, cmvn, IPD)
-> mask-weighted covariance -> MVDR solve -> beamform -> iSTFT.  It exists only so that
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg can check / time the
HIP path against it.  Nothing under `aps_amd/` may import it.

Pinned by: tests/golden/*.npz, generated in the build container by tests/golden/make_golden.py
from the real reference imported read-only (see tests/golden/MANIFEST.json), and by
oracle/_ref (the reference's own C++ FFT/STFT/window sources compiled in place).

Parity status of the single third-party piece: `librosa.filters.mel` (librosa==0.8.1 per the
reference's requirements.txt:4) is NOT installed here and the reference ships no stored mel
matrix, so `mel_weights` below restates the published librosa algorithm  --  mel weights:
PARITY UNPINNED.  Everything else is pinned by golden vectors produced by reference code.

Every function cites the reference file:line (under /root/reference) it restates.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# aps/const.py:17
EPSILON = float(np.finfo(np.float32).eps)
MATH_PI = math.pi

_WINDOWS = ("bartlett", "hann", "hamm", "blackman", "rect", "sqrthann")


# ----------------------------------------------------------------------------------------------
# a1  windows  (aps/transform/utils.py:30-59)
# ----------------------------------------------------------------------------------------------
def window(name: str, n: int) -> torch.Tensor:
    """All windows periodic=True (librosa convention) except the rectangular one."""
    if name not in _WINDOWS:
        raise RuntimeError(f"Unknown window type: {name}")
    if name == "rect":
        return torch.ones(n)
    if name == "sqrthann":
        return torch.hann_window(n, periodic=True)**0.5
    gen = {
        "hann": torch.hann_window,
        "hamm": torch.hamming_window,
        "blackman": torch.blackman_window,
        "bartlett": torch.bartlett_window,
    }[name]
    return gen(n, periodic=True)


def fft_size_of(frame_len: int, round_pow_of_two: bool = True, mode: str = "librosa") -> int:
    """aps/transform/utils.py:83-86"""
    if round_pow_of_two or mode == "kaldi":
        return 2**math.ceil(math.log2(frame_len))
    return frame_len


# ----------------------------------------------------------------------------------------------
# a2  dense DFT basis  (aps/transform/utils.py:62-112)
# ----------------------------------------------------------------------------------------------
def dft_basis(frame_len: int,
              win: torch.Tensor,
              round_pow_of_two: bool = True,
              normalized: bool = False,
              inverse: bool = False,
              mode: str = "librosa",
              dtype: torch.dtype = torch.float32):
    """Returns (K [2W,1,L], w [L]).  librosa: window centre-padded to W, L = W.
    kaldi: window untouched, basis rows truncated to frame_len, L = frame_len."""
    if mode not in ("librosa", "kaldi"):
        raise ValueError(f"Unsupported mode: {mode}")
    W = fft_size_of(frame_len, round_pow_of_two, mode)
    if mode == "librosa" and W != frame_len:
        lpad = (W - frame_len) // 2
        win = F.pad(win, (lpad, W - frame_len - lpad))
    scale = W**0.5 if normalized else 1
    spec = torch.fft.fft(torch.eye(W, dtype=dtype) / scale, dim=-1)  # [W(time), W(freq)]
    win = win.to(dtype)
    basis = torch.stack([spec.real, spec.imag], dim=-1)  # time x freq x 2
    if mode == "kaldi":
        basis = basis[:frame_len]
    if inverse and not normalized:
        basis = basis / W
    basis = basis.transpose(0, 2)  # 2 x freq x time
    basis = basis.reshape(W * 2, 1, basis.shape[-1])
    return basis, win


def num_frames(num_samples: int, kernel_width: int, hop: int, center: bool) -> int:
    """aps/transform/utils.py:653-662 (integer, exact)"""
    if center:
        num_samples = num_samples + kernel_width
    return (num_samples - kernel_width) // hop + 1


# ----------------------------------------------------------------------------------------------
# a3  forward STFT as strided dense DFT  (aps/transform/utils.py:227-290)
# ----------------------------------------------------------------------------------------------
def stft(wav: torch.Tensor,
         frame_len: int,
         frame_hop: int,
         window_name: str = "sqrthann",
         round_pow_of_two: bool = True,
         normalized: bool = False,
         pre_emphasis: float = 0,
         onesided: bool = True,
         center: bool = False,
         mode: str = "librosa",
         polar: bool = False,
         eps: float = EPSILON,
         dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """wav N x (C) x S  ->  N x (C) x F x T x 2 (re,im) or (mag,phase).
    dtype=float64 evaluates the same fp32 window / samples with a float64 basis and float64
    accumulation: the exact-arithmetic yardstick the tests measure both implementations against."""
    if wav.dim() not in (2, 3):
        raise RuntimeError(f"STFT expect 2D/3D tensor, but got {wav.dim():d}D")
    K, w = dft_basis(frame_len, window(window_name, frame_len), round_pow_of_two, normalized,
                     False, mode, dtype=dtype)
    wav = wav.to(dtype)
    lead = wav.shape[:-1]
    x = wav.reshape(-1, 1, wav.shape[-1])
    L = K.shape[-1]
    if center:
        x = F.pad(x, (L // 2, L // 2), mode="reflect")
    Kw = K * w
    if pre_emphasis > 0:
        # per-frame (Kaldi style) pre-emphasis, then dense DFT by matmul (:263-272)
        fr = F.unfold(x[:, None], (1, L), stride=frame_hop, padding=0)  # NC x L x T
        head = fr[:, :1] * (1 - pre_emphasis)
        tail = fr[:, 1:] - pre_emphasis * fr[:, :-1]
        fr = torch.cat([head, tail], 1)
        out = torch.matmul(Kw[:, 0][None], fr)
    else:
        out = F.conv1d(x, Kw, stride=frame_hop, padding=0)
    out = out.reshape(*lead, out.shape[-2], out.shape[-1])
    re, im = torch.chunk(out, 2, dim=-2)
    if onesided:
        nb = K.shape[0] // 4 + 1
        re, im = re[..., :nb, :], im[..., :nb, :]
    if polar:
        return torch.stack([(re**2 + im**2 + eps)**0.5, torch.atan2(im, re)], -1)
    return torch.stack([re, im], -1)


# ----------------------------------------------------------------------------------------------
# a5  inverse STFT  (aps/transform/utils.py:293-360)
# ----------------------------------------------------------------------------------------------
def istft(spec: torch.Tensor,
          frame_len: int,
          frame_hop: int,
          window_name: str = "sqrthann",
          round_pow_of_two: bool = True,
          normalized: bool = False,
          onesided: bool = True,
          center: bool = False,
          mode: str = "librosa",
          polar: bool = False,
          eps: float = EPSILON) -> torch.Tensor:
    """spec (N) x F x T x 2 -> N x S via transposed dense DFT + window^2 overlap-add normaliser"""
    if spec.dim() == 3:
        spec = spec[None]
    if spec.dim() != 4:
        raise RuntimeError(f"Expect 4D tensor, but got {spec.dim()}D")
    K, w = dft_basis(frame_len, window(window_name, frame_len), round_pow_of_two, normalized,
                     True, mode)
    if polar:
        re = spec[..., 0] * torch.cos(spec[..., 1])
        im = spec[..., 0] * torch.sin(spec[..., 1])
    else:
        re, im = spec[..., 0], spec[..., 1]
    if onesided:
        mirror = list(range(K.shape[0] // 4 - 1, 0, -1))
        re = torch.cat([re, re[:, mirror]], 1)
        im = torch.cat([im, -im[:, mirror]], 1)
    packed = torch.cat([re, im], 1)
    wav = F.conv_transpose1d(packed, K * w, stride=frame_hop, padding=0)
    T = packed.shape[-1]
    L = w.shape[0]
    wsq = (w**2)[:, None].repeat(1, T)[None]  # 1 x L x T
    eye = torch.eye(L)[:, None]
    den = F.conv_transpose1d(wsq, eye, stride=frame_hop, padding=0)
    if center:
        pad = K.shape[-1] // 2
        wav, den = wav[..., pad:-pad], den[..., pad:-pad]
    return (wav / (den + eps)).squeeze(1)


# ----------------------------------------------------------------------------------------------
# a7  mel filterbank.  aps side: aps/transform/utils.py:115-156.  librosa side: PARITY UNPINNED
# restatement of librosa 0.8.1 `filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=True, norm)`.
# ----------------------------------------------------------------------------------------------
def _hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asanyarray(f, dtype=np.float64) / 700.0)


def _mel_to_hz_htk(m):
    return 700.0 * (10.0**(np.asanyarray(m, dtype=np.float64) / 2595.0) - 1.0)


def librosa_mel_htk(sr, n_fft, n_mels, fmin, fmax, slaney_norm) -> np.ndarray:
    fft_f = np.linspace(0, float(sr) / 2, 1 + n_fft // 2, endpoint=True)
    mel_f = _mel_to_hz_htk(np.linspace(_hz_to_mel_htk(fmin), _hz_to_mel_htk(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fft_f)
    wts = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        wts[i] = np.maximum(0, np.minimum(lower, upper))
    if slaney_norm:
        enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
        wts *= enorm[:, None]  # float64 band norms applied in place to the float32 matrix
    return wts


def mel_weights(frame_len: int,
                round_pow_of_two: bool = True,
                num_bins=None,
                sr: int = 16000,
                num_mels: int = 80,
                fmin: float = 0.0,
                fmax=None,
                norm: bool = False) -> torch.Tensor:
    if num_bins is None:
        n_fft = 2**math.ceil(math.log2(frame_len)) if round_pow_of_two else frame_len
    else:
        n_fft = (num_bins - 1) * 2
    upper = sr // 2
    if fmax is None:
        fmax = upper
    else:
        fmax = min(fmax + upper if fmax < 0 else fmax, upper)
    fmin = max(0, fmin)
    return torch.tensor(librosa_mel_htk(sr, n_fft, num_mels, fmin, fmax, norm),
                        dtype=torch.float32)


# ----------------------------------------------------------------------------------------------
# a6/a8/a9  spectral feature layers  (aps/transform/asr.py:226-357, 431-464, 520-618)
# ----------------------------------------------------------------------------------------------
def magnitude(packed: torch.Tensor, eps: float = 0.0) -> torch.Tensor:
    return torch.sqrt(torch.sum(packed**2, -1) + eps)  # asr.py:303


def log_feature(x: torch.Tensor, eps: float = EPSILON, lower_bound: float = 0.0):
    if lower_bound > 0:  # asr.py:460-464
        return torch.log(lower_bound + x)
    return torch.log(torch.clamp(x, min=eps))


def cmvn(x, norm_mean=True, norm_var=True, per_band=True, eps=EPSILON, gmean=None, gstd=None):
    """asr.py:576-618.  NB: `per_band` reduces over the LAST (feature) dim, i.e. per frame."""
    if not norm_mean and not norm_var:
        return x
    if gmean is not None:
        if norm_mean:
            x = x - gmean
        if norm_var:
            x = x / gstd
        return x
    dims = -1 if per_band else (-1, -2)
    if norm_mean:
        x = x - torch.mean(x, dims, keepdim=True)
    if norm_var:
        if norm_mean:
            var = torch.mean(x**2, dims, keepdim=True)
        else:
            var = torch.var(x, dims, unbiased=False, keepdim=True)
        x = x / torch.sqrt(var + eps)
    return x


def dct_matrix(num_ceps: int, num_mels: int) -> torch.Tensor:
    """rows of the orthonormal DCT-II, = scipy.fftpack.dct(eye(M), norm="ortho")[:, :P].T which the
    reference builds its frozen `dct` parameter from (asr.py:484-488); closed form in float64"""
    n = torch.arange(num_mels, dtype=torch.float64)
    k = torch.arange(num_ceps, dtype=torch.float64)[:, None]
    mat = torch.cos(math.pi * (2 * n + 1) * k / (2 * num_mels)) * math.sqrt(2.0 / num_mels)
    mat[0] *= math.sqrt(0.5)
    return mat.float()


def dct(log_mel, num_ceps=13, lifter=0.0):
    """DiscreteCosineTransform.forward (asr.py:507-517)"""
    mfcc = F.linear(log_mel, dct_matrix(num_ceps, log_mel.shape[-1]).to(log_mel.dtype))
    if lifter > 0:
        lift = 1 + lifter * 0.5 * torch.sin(math.pi * torch.arange(1, 1 + num_ceps) / lifter)
        mfcc = mfcc * lift.to(log_mel.dtype)
    return mfcc


def splice(feats, lctx=1, rctx=1, op="cat"):
    """splice_feature (utils.py:193-224): context frames with the edges repeated"""
    if lctx + rctx == 0:
        return feats
    T = feats.shape[-2]
    ctx = []
    for c in range(-lctx, rctx + 1):
        idx = torch.clamp(torch.arange(c, c + T), min=0, max=T - 1)
        ctx.append(torch.index_select(feats, -2, idx))
    return torch.cat(ctx, -1) if op == "cat" else torch.stack(ctx, -1)


def splice_transform(feats, lctx=0, rctx=0, subsampling_factor=1):
    """SpliceTransform.forward (asr.py:715-728)"""
    feats = splice(feats, max(lctx, 0), max(rctx, 0))
    end = (feats.shape[-2] // subsampling_factor) * subsampling_factor
    if subsampling_factor != 1:
        feats = feats[..., :end:subsampling_factor, :]
    return feats


def delta_transform(feats, ctx=2, order=2, delta_as_channel=False):
    """DeltaTransform.forward (asr.py:760-782)"""
    scale = torch.arange(-ctx, ctx + 1, dtype=torch.float32)
    scale = (scale / sum(i * i for i in range(-ctx, ctx + 1))).to(feats.dtype)
    delta = [feats]
    for _ in range(order):
        delta.append(torch.sum(splice(delta[-1], ctx, ctx, op="stack") * scale, -1))
    return torch.stack(delta, 1) if delta_as_channel else torch.cat(delta, -1)


def spectral_chain(packed: torch.Tensor,
                   tokens,
                   mel_w=None,
                   use_power=False,
                   eps=EPSILON,
                   log_lower_bound=0.0,
                   norm_mean=True,
                   norm_var=True,
                   norm_per_band=True,
                   gcmvn=None,
                   num_ceps=13,
                   lifter=0.0,
                   lctx=0,
                   rctx=0,
                   subsampling_factor=1,
                   delta_ctx=2,
                   delta_order=2,
                   delta_as_channel=False) -> torch.Tensor:
    """packed N x (C) x F x T x 2 -> N x (C) x T x D for a token list starting with
    "spectrogram", "fbank" or "mfcc" followed by any of "log", "cmvn", "dct", "splice", "delta"
    (asr.py:902-990).  gcmvn = (gmean, gstd) or None."""
    head, rest = tokens[0], tokens[1:]
    x = magnitude(packed).transpose(-1, -2)
    x = x**(2 if use_power else 1)
    if head in ("fbank", "mfcc"):
        x = F.linear(x, mel_w.to(x.dtype))  # asr.py:427
        if head == "mfcc":
            x = dct(log_feature(x, eps, log_lower_bound), num_ceps, lifter)
    elif head != "spectrogram":
        raise RuntimeError(f"oracle: unsupported head token {head}")
    for tok in rest:
        if tok == "log":
            x = log_feature(x, eps, log_lower_bound)
        elif tok == "cmvn":
            gm, gs = gcmvn if gcmvn is not None else (None, None)
            x = cmvn(x, norm_mean, norm_var, norm_per_band, eps, gm, gs)
        elif tok == "dct":
            x = dct(x, num_ceps, lifter)
        elif tok == "splice":
            x = splice_transform(x, lctx, rctx, subsampling_factor)
        elif tok == "delta":
            x = delta_transform(x, delta_ctx, delta_order, delta_as_channel)
        else:
            raise RuntimeError(f"oracle: unsupported token {tok}")
    return x


def asr_features(wav,
                 feats="fbank-log-cmvn",
                 frame_len=400,
                 frame_hop=160,
                 window_name="hamm",
                 center=False,
                 round_pow_of_two=True,
                 stft_normalized=False,
                 stft_mode="librosa",
                 pre_emphasis=0.97,
                 use_power=False,
                 sr=16000,
                 log_lower_bound=0.0,
                 num_mels=80,
                 mel_coeff_norm=False,
                 min_freq=0,
                 max_freq=None,
                 norm_mean=True,
                 norm_var=True,
                 norm_per_band=True,
                 eps=EPSILON,
                 dtype=torch.float32,
                 **context):
    """AsrTransform forward for waveform-rooted chains (asr.py:837-1033); context = the dct /
    splice / delta / gcmvn keyword arguments of spectral_chain."""
    tokens = feats.split("-")
    packed = stft(wav, frame_len, frame_hop, window_name, round_pow_of_two, stft_normalized,
                  pre_emphasis, True, center, stft_mode, dtype=dtype)
    mel_w = None
    if tokens[0] in ("fbank", "mfcc"):
        mel_w = mel_weights(frame_len, round_pow_of_two, None, sr, num_mels, min_freq, max_freq,
                            mel_coeff_norm)
    return spectral_chain(packed, tokens, mel_w, use_power, eps, log_lower_bound, norm_mean,
                          norm_var, norm_per_band, **context)


def abs_mel_log_cmvn(yr, yi, mel_w, eps=EPSILON, tokens=("abs", "mel", "log", "cmvn")):
    """AsrTransform("abs-mel-log-cmvn") on a ComplexTensor N x T x F (asr.py:330-332, 950-971):
    eps is added to the REAL part before |.|."""
    x = None
    for tok in tokens:
        if tok == "abs":
            x = ((yr + eps)**2 + yi**2).sqrt()
        elif tok == "mel":
            x = F.linear(x, mel_w.to(x.dtype))
        elif tok == "log":
            x = log_feature(x, eps)
        elif tok == "cmvn":
            x = cmvn(x, eps=eps)
    return x


# ----------------------------------------------------------------------------------------------
# a11/a12  spatial features  (aps/transform/enh.py:52-143, 595-613)
# ----------------------------------------------------------------------------------------------
def parse_pairs(ipd_index: str):
    pairs = [tuple(map(int, p.split(","))) for p in ipd_index.split(";")]
    return [p[0] for p in pairs], [p[1] for p in pairs]


def ipd_features(packed, ipd_index, cos=True, sin=False):
    """packed N x C x F x T x 2 -> N x T x (P*F)"""
    pha = torch.atan2(packed[..., 1], packed[..., 0]).transpose(-1, -2)  # N x C x T x F
    N, C, T, _ = pha.shape
    pha = pha.transpose(1, 2).contiguous()  # N x T x C x F
    il, ir = parse_pairs(ipd_index)
    dif = pha[..., il, :] - pha[..., ir, :]
    if not cos:
        raise NameError("reference raises NameError for cos=False (enh.py:138-141)")
    out = torch.cos(dif)
    if sin:
        out = torch.cat([out, torch.sin(dif)], 2)
    return out.reshape(N, T, -1)


def enh_features(packed,
                 feats="spectrogram-log-cmvn-ipd",
                 ipd_index="",
                 cos_ipd=True,
                 sin_ipd=False,
                 ref_channel=0,
                 **chain_kwargs):
    toks = feats.split("-")
    mag_toks = [t for t in toks if t != "ipd"]
    out = []
    if mag_toks:
        ref = packed[:, ref_channel] if (packed.dim() == 5 and ref_channel >= 0) else packed
        out.append(spectral_chain(ref, mag_toks, **chain_kwargs))
    if "ipd" in toks and ipd_index:
        out.append(ipd_features(packed, ipd_index, cos_ipd, sin_ipd))
    return torch.cat(out, -1)


# ----------------------------------------------------------------------------------------------
# a13-a18  mask based MVDR  (aps/asr/filter/mvdr.py:19-174, aps/cplx.py:242-278,
#                            aps/asr/base/attention.py:18-36)
# ----------------------------------------------------------------------------------------------
def pad_mask(lens: torch.Tensor) -> torch.Tensor:
    steps = torch.arange(int(lens.max().item()))[None]
    return steps >= lens[:, None]


def process_mask(mask, x_len=None, mask_norm=True):
    """N x T x F -> N x F x T (mvdr.py:103-116)"""
    if x_len is not None:
        mask = mask.masked_fill(pad_mask(x_len)[..., None], 0)
    if mask_norm:
        peak = torch.norm(mask, float("inf"), dim=1, keepdim=True)
        mask = mask / (peak + EPSILON)
    return mask.transpose(1, 2)


def covar(mask, xr, xi):
    """mask N x F x T (real), X N x C x F x T -> (Rr, Ri) N x F x C x C   (mvdr.py:42-61)"""
    sr, si = xr.transpose(1, 2), xi.transpose(1, 2)  # N x F x C x T
    m = mask.unsqueeze(-2)
    ar, ai = sr * m, si * m
    br, bi = sr.transpose(-1, -2), -1.0 * si.transpose(-1, -2)  # conj transpose
    rr = torch.matmul(ar, br) - torch.matmul(ai, bi)
    ri = torch.matmul(ai, br) + torch.matmul(ar, bi)
    den = torch.clamp(m.sum(-1, keepdim=True), min=EPSILON)
    return rr / den, ri / den


def cplx_inverse(ar, ai):
    """2C x 2C real block inverse (cplx.py:268-278)"""
    top = torch.cat([ar, -1.0 * ai], -1)
    bot = torch.cat([ai, ar], -1)
    inv = torch.cat([top, bot], -2).inverse()
    upper, _ = torch.chunk(inv, 2, dim=-2)
    re, im = torch.chunk(upper, 2, dim=-1)
    return re, -im


def channel_attention(rr, ri, proj_w, proj_b, gvec_w, gvec_b):
    """Rs N x F x C x C -> u N x C   (mvdr.py:148-174)"""
    C = rr.shape[-1]
    diag = torch.eye(C, dtype=torch.bool)
    mr = rr.masked_fill(diag, 0).sum(-1) / (C - 1)
    mi = ri.masked_fill(diag, 0).sum(-1) / (C - 1)
    mag = (mr**2 + mi**2).sqrt().transpose(1, 2)  # N x C x F
    hid = torch.tanh(F.linear(mag, proj_w, proj_b))
    score = F.linear(hid, gvec_w, gvec_b).squeeze(-1)
    return torch.softmax(score, -1)


def mvdr_weight(rs, rn, u, eps=1e-5):
    """(Rs, Rn) pairs of N x F x C x C, u N x C -> w (wr, wi) N x F x C   (mvdr.py:75-101)"""
    (sr, si), (nr, ni) = rs, rn
    C = nr.shape[-1]
    nr = nr + torch.eye(C) * eps
    ir, ii = cplx_inverse(nr, ni)
    yr = torch.matmul(ir, sr) - torch.matmul(ii, si)
    yi = torch.matmul(ii, sr) + torch.matmul(ir, si)
    eye = torch.eye(C, dtype=torch.bool).expand(*yr.shape)
    shape = yr.shape[:-1]
    tr_r = yr.masked_select(eye).view(*shape).sum(-1) + eps
    tr_i = yi.masked_select(eye).view(*shape).sum(-1)
    vr = (yr * u[:, None, None, :]).sum(-1)
    vi = (yi * u[:, None, None, :]).sum(-1)
    tr_r, tr_i = tr_r[..., None], tr_i[..., None]
    den = tr_r**2 + tr_i**2
    return (vr * tr_r + vi * tr_i) / den, (vi * tr_r - vr * tr_i) / den


def beamform(wr, wi, xr, xi):
    """w N x C x F, X N x C x F x T -> y N x F x T   (mvdr.py:29-39)"""
    cr, ci = wr[..., None], -1.0 * wi[..., None]
    return (cr * xr - ci * xi).sum(1), (ci * xr + cr * xi).sum(1)


def mvdr_forward(mask_s, xr, xi, att, mask_n=None, x_len=None, mask_norm=True, eps=1e-5):
    """MvdrBeamformer.forward (mvdr.py:118-145).  att = (proj_w, proj_b, gvec_w, gvec_b).
    Returns (yr, yi) N x T x F and the intermediates dict."""
    ms = process_mask(mask_s, x_len, mask_norm)
    mn = process_mask(mask_n, x_len, mask_norm) if mask_n is not None else None
    rs = covar(ms, xr, xi)
    rn = covar(1 - ms if mn is None else mn, xr, xi)
    u = channel_attention(rs[0], rs[1], *att)
    wr, wi = mvdr_weight(rs, rn, u, eps)
    wr, wi = wr.transpose(1, 2), wi.transpose(1, 2)
    yr, yi = beamform(wr, wi, xr, xi)
    inter = {"Rs": rs, "Rn": rn, "u": u, "w": (wr, wi)}
    return yr.transpose(1, 2), yi.transpose(1, 2), inter


# ----------------------------------------------------------------------------------------------
# a21  TF masking  (aps/sse/base.py:23-47)
# ----------------------------------------------------------------------------------------------
def tf_masking(packed, mask, ref_channel=0):
    """packed N x (C) x F x T x 2, mask N x F x T (real) or N x F x T x 2 (complex)"""
    if packed.dim() == 5:
        packed = packed[:, ref_channel]
    re, im = packed[..., 0], packed[..., 1]
    if mask.dim() == 4:
        mr, mi = mask[..., 0], mask[..., 1]
        return torch.stack([re * mr - im * mi, im * mr + re * mi], -1)
    return torch.stack([re * mask, im * mask], -1)


# ----------------------------------------------------------------------------------------------
# 8f row 3  geometry-dependent layers  (aps/transform/enh.py:146-384)
# ----------------------------------------------------------------------------------------------
def fixed_beamform(xr, xi, wr, wi, beam=None):
    """FixedBeamformer.forward (enh.py:349-384): y = sum_c conj(w[b, c, f]) x[n, c, f, t].
    xr / xi [N, C, F, T], wr / wi [B, C, F]; beam None -> [N, B, F, T], an index or N indices
    -> [N, F, T].  Restated as one complex contraction (the reference spells out four real sums)."""
    x = torch.complex(xr.double(), xi.double())
    w = torch.complex(wr.double(), wi.double())
    if beam is None:
        y = torch.einsum("bcf,ncft->nbft", w.conj(), x)
    else:
        sel = torch.as_tensor(beam, dtype=torch.int64).reshape(-1).expand(x.shape[0])
        y = torch.einsum("ncf,ncft->nft", w.conj()[sel], x)
    return y.real.float(), y.imag.float()


def directional_feature(phase, doa, index_l, index_r, num_doas=1, sr=16000, velocity=340.0,
                        radius=0.0425):
    """DfTransform.forward for the "7@" array (enh.py:195-300).  phase [N, C, T, F]; doa [N] (or a
    list of them: speakers side by side on the last axis) when num_doas == 1, otherwise ignored
    and num_doas directions are sampled on [0, 2 pi).  Returns [N, T, F x speakers] or
    [N, D, T, F].  Microphone c > 0 sits at angle (c - 1) 60 degrees on the circle, the centre is
    microphone 0: the delay of a plane wave from `doa` is -R cos(doa - angle_c) / v up to the
    reference's own sign convention (enh.py:218-226), restated here from that geometry."""
    N, C, T, F = phase.shape
    omega = torch.tensor([math.pi * sr * f / (F - 1) for f in range(F)], dtype=torch.float32)

    def taus(angles):  # [...]-> [..., 7]
        cols = [torch.zeros_like(angles)]
        for c in range(1, 7):
            # enh.py:218-226: -cos(a), -cos(pi/3 - a), -cos(2pi/3 - a), +cos(a), +cos(pi/3 - a), ...
            sign = -1.0 if c <= 3 else 1.0
            cols.append(sign * torch.cos(((c - 1) % 3) * MATH_PI / 3 - angles) if (c - 1) % 3
                        else sign * torch.cos(angles))
        return radius * torch.stack(cols, -1) / velocity

    ipd = phase[:, index_l] - phase[:, index_r]  # N x P x T x F

    def one(angles):
        if num_doas != 1:
            angles = torch.linspace(0, MATH_PI * 2, num_doas + 1)[:-1].repeat(N, 1)  # N x D
        phi = taus(angles.float()).unsqueeze(-1) * (-omega)  # N x (D) x 7 x F
        if num_doas == 1:
            dif = phi[:, index_l] - phi[:, index_r]  # N x P x F
            return torch.cos(ipd - dif[:, :, None, :]).mean(1)
        dif = phi[:, :, index_l] - phi[:, :, index_r]  # N x D x P x F
        return torch.cos(ipd[:, None] - dif[:, :, :, None, :]).mean(2)

    if isinstance(doa, (list, tuple)):
        if num_doas != 1:
            raise RuntimeError("known_doa=False, no need to pass doa as a Sequence object")
        return torch.cat([one(d) for d in doa], -1)
    return one(doa)


# ----------------------------------------------------------------------------------------------
# 8f row 3  frame-by-frame (i)STFT  (aps/transform/streaming.py:13-152)
# ----------------------------------------------------------------------------------------------
def streaming_stft(wav, w, hop, normalized=False, polar=False, eps=EPSILON):
    """StreamingSTFT.forward (streaming.py:45-64): frames of len(w) samples at 0, hop, 2 hop, ...,
    rfft(frame * w) of that same size.  wav N x (C) x S -> N x (C) x F x T x 2.  Restated with
    unfold instead of the reference's Python loop over frames."""
    W = w.shape[0]
    frames = wav.unfold(-1, W, hop) * w  # ... x T x W
    spec = torch.fft.rfft(frames, W, dim=-1, norm="ortho" if normalized else "backward")
    out = torch.view_as_real(spec)  # ... x T x F x 2
    if polar:
        mag = (torch.sum(out**2, -1) + eps)**0.5
        out = torch.stack([mag, torch.atan2(out[..., 1], out[..., 0])], -1)
    return out.transpose(-2, -3)


def streaming_istft_frames(packed, w, normalized=False, polar=False):
    """irfft(frame) * w for every frame (streaming.py:90-98): N x F x T x 2 -> N x T x W"""
    x = packed.transpose(-2, -3)
    if polar:
        x = torch.stack([x[..., 0] * torch.cos(x[..., 1]), x[..., 0] * torch.sin(x[..., 1])], -1)
    W = w.shape[0]
    frames = torch.fft.irfft(torch.view_as_complex(x.contiguous()), W, dim=-1,
                             norm="ortho" if normalized else "backward")
    return frames * w


def streaming_istft(packed, w, hop, normalized=False, polar=False, eps=EPSILON):
    """StreamingiSTFT.forward (streaming.py:132-152): all steps + flush = overlap-add of the
    windowed frames divided by the overlap-added window energy + eps, T hop + (W - hop) samples.
    Restated as two scatter-adds instead of the reference's running caches."""
    frames = streaming_istft_frames(packed, w, normalized, polar)  # N x T x W
    N, T, W = frames.shape
    S = (T - 1) * hop + W
    wav = torch.zeros(N, S)
    den = torch.zeros(S)
    for t in range(T):
        wav[:, t * hop:t * hop + W] += frames[:, t]
        den[t * hop:t * hop + W] += w**2
    return wav / (den + eps)


# ----------------------------------------------------------------------------------------------
# 8f row 3  training-time augmentation  (aps/transform/augment.py, aps/transform/asr.py:116-195,
# 621-684).  The random draws are part of the algorithm: same generators, same order.
# ----------------------------------------------------------------------------------------------
def perturb_speed_row(wav, weight):
    """perturb_speed for one utterance (augment.py:86-109): wav [S], weight [dst, src, K] ->
    [(S // src) dst].  Restated as a contraction over zero-padded block windows instead of conv1d."""
    dst, src, K = weight.shape
    B = wav.shape[0] // src
    if B == 0:
        raise RuntimeError(f"Input wav is too short to be perturbed, length = {wav.shape[0]}")
    pad = (K - 1) // 2
    blocks = F.pad(wav[:B * src].view(B, src), (0, 0, pad, pad))  # (B + 2 pad) x src
    windows = blocks.unfold(0, K, 1)  # B x src x K
    return torch.einsum("bik,jik->bj", windows, weight).reshape(-1)


def speed_perturb(wav, weights, choice):
    """SpeedPerturbTransform.forward in training mode given the drawn choices (asr.py:178-195):
    choice[n] == len(weights) keeps utterance n; the batch is zero padded to its longest member."""
    rows = [wav[n] if c == len(weights) else perturb_speed_row(wav[n], weights[c])
            for n, c in enumerate(choice.tolist())]
    out = torch.zeros(len(rows), max(r.shape[0] for r in rows))
    for n, r in enumerate(rows):
        out[n, :r.shape[0]] = r
    return out


def tf_bands(batch, T, Fdim, pm=0.0, ps=0.0, max_bands=30, max_frame=40, num_freq_masks=2,
             num_time_masks=2):
    """the draws of tf_mask / random_mask (augment.py:13-83) as (begin, length) pairs per utterance,
    frequency bands first; consumes Python's `random` exactly like the reference"""
    import random
    max_bands = min(max_bands, Fdim)
    if ps > 0:
        max_frame = min(max_frame, int(T * ps))
    if pm > 0:
        num_time_masks = min(num_time_masks, int(T * pm))
    out = []
    for _ in range(batch):
        fb, tb = [], []
        for store, size, limit, count in ((fb, Fdim, max_bands, num_freq_masks),
                                          (tb, T, max_frame, num_time_masks)):
            for _ in range(count):
                dur = random.randint(1, limit - 1)
                if size - dur <= 0:
                    continue
                store.append((random.randint(0, size - dur - 1), dur))
        out.append((fb, tb))
    return out


def spec_augment(x, bands, mask_zero=True):
    """SpecAugTransform.forward after the coin flip (asr.py:660-684): x N x (C) x T x F, bands from
    tf_bands; masked cells -> 0 (x * mask) or the mean of the whole input"""
    keep = torch.ones(x.shape[0], x.shape[-2], x.shape[-1], dtype=torch.bool)
    for n, (fb, tb) in enumerate(bands):
        for beg, dur in fb:
            keep[n, :, beg:beg + dur] = False
        for beg, dur in tb:
            keep[n, beg:beg + dur, :] = False
    if x.dim() == 4:
        keep = keep[:, None]
    return x * keep if mask_zero else torch.where(keep, x, x.mean())


# ----------------------------------------------------------------------------------------------
# a21  MaskNonLinear  (aps/sse/base.py:112-156)
# ----------------------------------------------------------------------------------------------
def mask_nonlinear(x, name, scale=1.0, vmax=None, vmin=None):
    """out = clamp(f(x) * scale): f elementwise, softmax over the leading (source) axis"""
    if name == "softmax":
        e = torch.exp(x - x.max(0, keepdim=True).values)
        y = e / e.sum(0, keepdim=True)
    elif name == "softplus":
        y = torch.where(x > 20, x, torch.log1p(torch.exp(x)))
    else:
        y = {"none": lambda v: v, "relu": lambda v: v.clamp_min(0), "tanh": torch.tanh,
             "sigmoid": lambda v: 1 / (1 + torch.exp(-v))}[name](x)
    y = y * scale
    if vmax is not None:
        y = y.clamp(max=vmax)
    if vmin is not None:
        y = y.clamp(min=vmin)
    return y
