"""
Mask based MVDR front-end -- the surface of aps/asr/filter/mvdr.py on the kernels of
aps_amd/csrc/mvdr.hip.

Call graph of MvdrBeamformer.forward (mvdr.py:118-145) here: 5 launches over the bin-fastest store
    covariance partials (speech + noise, mask padding / normalisation folded in)
    fold     (segments -> packed Rs | Rn triangles + |off-diagonal mean Rs|, all coalesced)
    attention partial scores
    weight   (softmax over channels -> u, per-bin complex solve -> w)
    beamform                              -> y  (N x T x F complex)
The stage-by-stage functions (covariance, ChannelAttention.attend, derive_weight, ...) run the same
arithmetic piecewise for callers that want the intermediates.
No tensor is transposed or copied in between; the reference materialises ~12 intermediates
(`aten::copy_` of the transposed operands is 50 % of its CPU time).
"""
import ctypes as C
from typing import Optional

import torch as th
import torch.nn as nn

from aps_amd import _native as nat
from aps_amd.cplx import ComplexTensor
from aps_amd.libs import Register
from aps_amd.ops import mvdr_singular_check, mvdr_singular_flag
from aps_amd.spectrogram import store_of_pair

EnhFrontEnds = Register("enh_filter")


def _store5(x: ComplexTensor) -> th.Tensor:
    """ComplexTensor N x C x F x T -> store N x C x T x F x 2"""
    if x.dim() != 4:
        raise RuntimeError(f"expect N x C x F x T complex spectrogram, got {x.dim()}D")
    return store_of_pair(x.real, x.imag)


def _cplx_of(t: th.Tensor) -> ComplexTensor:
    """interleaved (..., 2) -> ComplexTensor of the two strided views"""
    return ComplexTensor(t[..., 0], t[..., 1])


def trace(cplx_mat: ComplexTensor) -> ComplexTensor:
    """trace of (..., C, C) complex matrices (mvdr.py:19-26); view + one reduction"""
    return ComplexTensor(th.diagonal(cplx_mat.real, dim1=-2, dim2=-1).sum(-1),
                         th.diagonal(cplx_mat.imag, dim1=-2, dim2=-1).sum(-1))



def _mask_operands(mask_s: th.Tensor, mask_n: Optional[th.Tensor], N: int, T: int, F: int):
    """(mask_s, mask_n, ld): the masks as the kernels read them -- fp32, frames `ld` floats apart.  The two halves of
    one N x T x 2F estimate (th.chunk of the mask net's output, mvdr.py:132-135) are read in place (ld = 2F): no copy
    of 2 x N T F floats per step; anything else is made dense (ld = F)."""
    for m in (mask_s, mask_n):
        if m is not None and tuple(m.shape) != (N, T, F):
            raise RuntimeError(f"mask shape {tuple(m.shape)} != {(N, T, F)}")

    def pitch(m):
        ok = m.dtype == th.float32 and not m.requires_grad and m.stride(2) == 1 and m.stride(1) >= F and \
            m.stride(0) == T * m.stride(1)
        return m.stride(1) if ok else 0

    ld = pitch(mask_s)
    if ld and (mask_n is None or pitch(mask_n) == ld):
        return mask_s.detach(), None if mask_n is None else mask_n.detach(), ld
    return nat.f32c(mask_s), None if mask_n is None else nat.f32c(mask_n), F

def covariance(store: th.Tensor,
               mask_s: th.Tensor,
               mask_n: Optional[th.Tensor] = None,
               x_len: Optional[th.Tensor] = None,
               mask_norm: bool = True,
               return_masks: bool = False,
               return_offdiag: bool = False):
    """store N x C x T x F x 2, masks N x T x F (raw, as the mask net emits them)
    -> Rs, Rn  N x F x C x C x 2  (+ processed masks N x F x T when asked)"""
    nat.require_device(store, mask_s, mask_n, x_len)
    lib = nat.load()
    N, Cn, T, F, _ = store.shape
    mask_s, mask_n, mask_ld = _mask_operands(mask_s, mask_n, N, T, F)
    if x_len is not None:
        x_len = x_len.to(device=store.device, dtype=th.int64).contiguous()
    dev = store.device
    cov_s = th.empty(N, F, Cn, Cn, 2, device=dev, dtype=th.float32)
    cov_n = th.empty(N, F, Cn, Cn, 2, device=dev, dtype=th.float32)
    pm_s = th.empty(N, F, T, device=dev, dtype=th.float32) if return_masks else None
    pm_n = th.empty(N, F, T, device=dev, dtype=th.float32) if return_masks else None
    offd = th.empty(N, Cn, F, device=dev, dtype=th.float32) if return_offdiag else None
    nbytes = int(lib.aps_mvdr_covariance_workspace(N, Cn, T, F))
    if nbytes < 0:
        raise RuntimeError(f"MVDR supports 2..8 channels, got {Cn}")
    work = th.empty(nbytes // 4, device=dev, dtype=th.float32)
    rc = lib.aps_mvdr_covariance(nat.ptr(store), N, Cn, T, F, store.stride(0), store.stride(1),
                                 store.stride(2), nat.ptr(mask_s), nat.ptr(mask_n), mask_ld, nat.ptr(x_len),
                                 int(mask_norm), nat.ptr(cov_s), nat.ptr(cov_n), nat.ptr(offd),
                                 nat.ptr(pm_s), nat.ptr(pm_n), nat.ptr(work), nat.stream_of(store))
    nat.check(rc, "aps_mvdr_covariance")
    out = (cov_s, cov_n)
    if return_masks:
        out = out + (pm_s, pm_n)
    if return_offdiag:
        out = out + (offd,)
    return out


def estimate_covar(mask: th.Tensor, spectrogram: ComplexTensor) -> ComplexTensor:
    """Covariance estimation with an already processed mask N x F x T (mvdr.py:42-61)"""
    store = _store5(spectrogram)
    cov, _ = covariance(store, mask.transpose(1, 2), None, None, mask_norm=False)
    return _cplx_of(cov)


def beamform_store(store: th.Tensor, weight: th.Tensor) -> th.Tensor:
    """store N x C x T x F x 2, weight N x F x C x 2 -> y N x T x F x 2"""
    nat.require_device(store, weight)
    lib = nat.load()
    N, Cn, T, F, _ = store.shape
    y = th.empty(N, T, F, 2, device=store.device, dtype=th.float32)
    rc = lib.aps_mvdr_beamform(nat.ptr(store), nat.ptr(nat.f32c(weight)), N, Cn, T, F,
                               store.stride(0), store.stride(1), store.stride(2), nat.ptr(y),
                               nat.stream_of(store))
    nat.check(rc, "aps_mvdr_beamform")
    return y


def beamform_features(store: th.Tensor, weight: th.Tensor, plan, abs_eps: float,
                      nan_flag: Optional[th.Tensor] = None, want_beam: bool = False):
    """beamform + AbsTransform + [mel] [log] [row cmvn] in ONE launch (aps_mvdr_beamform_features, SURVEY
    8(d) P3): store N x C x T x F x 2, weight N x F x C x 2 -> (features N x T x D, beam N x T x F x 2 or
    None).  The beam output is written only when asked for.  None when the kernel does not take the shape."""
    from aps_amd.ops import _feat_params, _mel_ptrs
    nat.require_device(store, weight)
    lib = nat.load()
    N, Cn, T, F, _ = store.shape
    D = plan.mel.num_mels if plan.mel else F
    y = th.empty(N, T, F, 2, device=store.device, dtype=th.float32) if want_beam else None
    out = th.empty(N, T, D, device=store.device, dtype=th.float32)
    p = _feat_params(F, 1, 0, plan, 0, False)
    ms, ml, mo, mw = _mel_ptrs(plan)
    import ctypes as C
    rc = lib.aps_mvdr_beamform_features(nat.ptr(store), nat.ptr(nat.f32c(weight)), N, Cn, T, F,
                                        store.stride(0), store.stride(1), store.stride(2), float(abs_eps),
                                        C.byref(p), ms, ml, mo, mw, nat.ptr(y), nat.ptr(out),
                                        nat.ptr(nan_flag), nat.stream_of(store))
    if rc == nat.ERR_UNSUPPORTED:
        return None
    nat.check(rc, "aps_mvdr_beamform_features")
    return out, y


def beamform(weight: ComplexTensor, spectrogram: ComplexTensor) -> ComplexTensor:
    """weight N x C x F, spectrogram N x C x F x T -> beam N x F x T (mvdr.py:29-39)"""
    w = th.stack([weight.real, weight.imag], -1).transpose(1, 2)  # N x F x C x 2
    y = beamform_store(_store5(spectrogram), w)
    return _cplx_of(y).transpose(1, 2)


class ChannelAttention(nn.Module):
    """Reference-channel attention u for MVDR (mvdr.py:148-174); parameters proj, gvec"""

    def __init__(self, num_bins: int, att_dim: int) -> None:
        super(ChannelAttention, self).__init__()
        self.proj = nn.Linear(num_bins, att_dim)
        self.gvec = nn.Linear(att_dim, 1)

    def attend(self, cov_s: th.Tensor) -> th.Tensor:
        """Rs N x F x C x C x 2 -> u N x C"""
        nat.require_device(cov_s, self.proj.weight)
        lib = nat.load()
        N, F, Cn = cov_s.shape[:3]
        A = self.proj.weight.shape[0]
        if self.proj.weight.shape[1] != F:
            raise RuntimeError(f"ChannelAttention built for {self.proj.weight.shape[1]} bins, "
                               f"covariance has {F}")
        nbytes = int(lib.aps_mvdr_attention_scratch(N, Cn, A))
        if nbytes < 0:
            raise RuntimeError(f"MVDR supports 2..8 channels, got {Cn}")
        scratch = th.empty(nbytes // 4, device=cov_s.device, dtype=th.float32)
        u = th.empty(N, Cn, device=cov_s.device, dtype=th.float32)
        rc = lib.aps_mvdr_channel_attention(nat.ptr(nat.f32c(cov_s)), N, Cn, F, A,
                                            nat.ptr(self.proj.weight.data.contiguous()),
                                            nat.ptr(self.proj.bias.data),
                                            nat.ptr(self.gvec.weight.data.contiguous()),
                                            nat.ptr(self.gvec.bias.data), nat.ptr(scratch),
                                            nat.ptr(u), nat.stream_of(cov_s))
        nat.check(rc, "aps_mvdr_channel_attention")
        return u

    def forward(self, Rs: ComplexTensor) -> th.Tensor:
        """Rs complex N x F x C x C -> u N x C"""
        return self.attend(th.stack([Rs.real, Rs.imag], -1).contiguous())


class MvdrBeamformer(nn.Module):
    """MVDR beamformer (mvdr.py:64-145)"""

    def __init__(self, num_bins, att_dim=512, mask_norm=True, eps=1e-5):
        super(MvdrBeamformer, self).__init__()
        self.ref = ChannelAttention(num_bins, att_dim)
        self.mask_norm = mask_norm
        self.eps = eps
        # a singular Rn + eps I (mvdr.py:89-92 -> cplx.py:268-278 -> th.inverse raises): counted by the solve
        # kernels; "deferred" reads the count back without stalling the stream and raises
        # torch.linalg.LinAlgError at a later call (ops.MVDR_SINGULAR.flush() forces it), "sync" raises at
        # the call like the reference (a host stall per forward), "manual" / "off": ops.MVDR_SINGULAR.count()
        self.singular_policy = "deferred"

    @staticmethod
    def check_singular() -> None:
        """Raise NOW (torch.linalg.LinAlgError) for every singular system the solve kernels have counted so far, on
        any device: what `singular_policy = "deferred"` postpones.  The deferred error surfaces at a LATER solve call
        (or here), not at the call the reference's `Rn.inverse()` raises from: a serving loop calls this at its batch
        boundary, a caller that needs the reference's call-site semantics sets `singular_policy = "sync"`."""
        from aps_amd.ops import MVDR_SINGULAR
        MVDR_SINGULAR.flush()

    def derive_weight(self, cov_s: th.Tensor, cov_n: th.Tensor, u: th.Tensor,
                      eps: float = 1e-5) -> th.Tensor:
        """Rs, Rn N x F x C x C x 2, u N x C -> w N x F x C x 2"""
        nat.require_device(cov_s, cov_n, u)
        lib = nat.load()
        N, F, Cn = cov_s.shape[:3]
        w = th.empty(N, F, Cn, 2, device=cov_s.device, dtype=th.float32)
        rc = lib.aps_mvdr_weight(nat.ptr(nat.f32c(cov_s)), nat.ptr(nat.f32c(cov_n)),
                                 nat.ptr(nat.f32c(u)), N, Cn, F, float(eps), nat.ptr(w),
                                 nat.ptr(mvdr_singular_flag(cov_s.device)), nat.stream_of(cov_s))
        nat.check(rc, "aps_mvdr_weight")
        mvdr_singular_check(self.singular_policy, tuple(cov_n.shape), cov_n.device)
        return w

    def weights_from_masks(self, store: th.Tensor, mask_s: th.Tensor,
                           mask_n: Optional[th.Tensor] = None,
                           x_len: Optional[th.Tensor] = None, return_cov: bool = False):
        """store N x C x T x F x 2 + raw masks N x T x F -> (u N x C, w N x F x C x 2) in two
        launches (covariance partials; fold + attention + solve).  mvdr.py:132-140"""
        ref = self.ref
        nat.require_device(store, mask_s, mask_n, x_len, ref.proj.weight)
        lib = nat.load()
        N, Cn, T, F, _ = store.shape
        A = ref.proj.weight.shape[0]
        if ref.proj.weight.shape[1] != F:
            raise RuntimeError(f"ChannelAttention built for {ref.proj.weight.shape[1]} bins, "
                               f"spectrogram has {F}")
        mask_s, mask_n, mask_ld = _mask_operands(mask_s, mask_n, N, T, F)
        if x_len is not None:
            x_len = x_len.to(device=store.device, dtype=th.int64).contiguous()
        nbytes = int(lib.aps_mvdr_weights_workspace(N, Cn, T, F, A))
        if nbytes < 0:
            raise RuntimeError(f"MVDR supports 2..8 channels, got {Cn}")
        dev = store.device
        work = th.empty(nbytes // 4, device=dev, dtype=th.float32)
        u = th.empty(N, Cn, device=dev, dtype=th.float32)
        w = th.empty(N, F, Cn, 2, device=dev, dtype=th.float32)
        cov_s = th.empty(N, F, Cn, Cn, 2, device=dev, dtype=th.float32) if return_cov else None
        cov_n = th.empty(N, F, Cn, Cn, 2, device=dev, dtype=th.float32) if return_cov else None
        rc = lib.aps_mvdr_weights(nat.ptr(store), N, Cn, T, F, store.stride(0), store.stride(1),
                                  store.stride(2), nat.ptr(mask_s), nat.ptr(mask_n), mask_ld,
                                  nat.ptr(x_len), int(self.mask_norm), A,
                                  nat.ptr(ref.proj.weight.data.contiguous()),
                                  nat.ptr(ref.proj.bias.data),
                                  nat.ptr(ref.gvec.weight.data.contiguous()),
                                  nat.ptr(ref.gvec.bias.data), float(self.eps), nat.ptr(work),
                                  nat.ptr(cov_s), nat.ptr(cov_n), nat.ptr(u), nat.ptr(w),
                                  nat.ptr(mvdr_singular_flag(dev)), nat.stream_of(store))
        nat.check(rc, "aps_mvdr_weights")
        mvdr_singular_check(self.singular_policy, (N, F, Cn, Cn), dev)
        if return_cov:
            return u, w, cov_s, cov_n
        return u, w

    def attend_and_derive(self, cov_s: th.Tensor, cov_n: th.Tensor, eps: float = 1e-5,
                          offdiag: Optional[th.Tensor] = None):
        """Rs, Rn N x F x C x C x 2 -> (u N x C, w N x F x C x 2): attention + solve, 2 launches.
        offdiag: the N x C x F by-product of covariance(..., return_offdiag=True), if at hand."""
        ref = self.ref
        nat.require_device(cov_s, cov_n, ref.proj.weight)
        lib = nat.load()
        N, F, Cn = cov_s.shape[:3]
        A = ref.proj.weight.shape[0]
        if ref.proj.weight.shape[1] != F:
            raise RuntimeError(f"ChannelAttention built for {ref.proj.weight.shape[1]} bins, "
                               f"covariance has {F}")
        nbytes = int(lib.aps_mvdr_attention_scratch(N, Cn, A))
        if nbytes < 0:
            raise RuntimeError(f"MVDR supports 2..8 channels, got {Cn}")
        dev = cov_s.device
        scratch = th.empty(nbytes // 4, device=dev, dtype=th.float32)
        u = th.empty(N, Cn, device=dev, dtype=th.float32)
        w = th.empty(N, F, Cn, 2, device=dev, dtype=th.float32)
        rc = lib.aps_mvdr_attention_weight(nat.ptr(nat.f32c(cov_s)), nat.ptr(nat.f32c(cov_n)),
                                           nat.ptr(offdiag), N, Cn, F, A, nat.ptr(ref.proj.weight.data.contiguous()),
                                           nat.ptr(ref.proj.bias.data),
                                           nat.ptr(ref.gvec.weight.data.contiguous()),
                                           nat.ptr(ref.gvec.bias.data), float(eps),
                                           nat.ptr(scratch), nat.ptr(u), nat.ptr(w),
                                           nat.ptr(mvdr_singular_flag(dev)), nat.stream_of(cov_s))
        nat.check(rc, "aps_mvdr_attention_weight")
        mvdr_singular_check(self.singular_policy, tuple(cov_n.shape), cov_n.device)
        return u, w

    def _derive_weight(self, Rs: ComplexTensor, Rn: ComplexTensor, u: th.Tensor,
                       eps: float = 1e-5) -> ComplexTensor:
        """ComplexTensor flavour of derive_weight (mvdr.py:75-101): -> weight N x F x C"""
        w = self.derive_weight(th.stack([Rs.real, Rs.imag], -1), th.stack([Rn.real, Rn.imag], -1),
                               u, eps)
        return _cplx_of(w)

    def _process_mask(self, mask: Optional[th.Tensor],
                      x_len: Optional[th.Tensor]) -> Optional[th.Tensor]:
        """N x T x F -> padded-zeroed, max-normalised, transposed N x F x T (mvdr.py:103-116).
        Stand-alone form (aps_mvdr_process_mask); `forward` folds this into the covariance kernel."""
        if mask is None:
            return mask
        nat.require_device(mask, x_len)
        N, T, F = mask.shape
        if x_len is not None:
            x_len = x_len.to(device=mask.device, dtype=th.int64).contiguous()
        out = th.empty(N, F, T, device=mask.device, dtype=th.float32)
        rc = nat.load().aps_mvdr_process_mask(nat.ptr(nat.f32c(mask)), nat.ptr(x_len), N, T, F,
                                              int(self.mask_norm), 0, nat.ptr(out), nat.stream_of(mask))
        nat.check(rc, "aps_mvdr_process_mask")
        return out

    def forward(self,
                mask_s: th.Tensor,
                x: ComplexTensor,
                mask_n: Optional[th.Tensor] = None,
                x_len: Optional[th.Tensor] = None) -> ComplexTensor:
        """mask_s/mask_n N x T x F, x complex N x C x F x T -> y complex N x T x F"""
        store = _store5(x)
        if nat.needs_grad(mask_s, mask_n, *self.ref.parameters()):
            return _cplx_of(self.forward_trainable(store, mask_s, mask_n, x_len))
        _, w = self.weights_from_masks(store, mask_s, mask_n, x_len)
        return _cplx_of(beamform_store(store, w))

    def forward_trainable(self, store: th.Tensor, mask_s: th.Tensor, mask_n: Optional[th.Tensor],
                          x_len: Optional[th.Tensor]) -> th.Tensor:
        """The same arithmetic stage by stage, every stage an autograd.Function with a HIP backward
        (aps_amd/grad_ops.py): gradients reach the masks (hence the mask estimator) and the
        ChannelAttention parameters; the spectrogram is data.  mask_n None: the implicit noise mask
        1 - processed speech mask (mvdr.py:135), whose gradient reaches the speech mask as well.
        -> y N x T x F x 2"""
        from aps_amd.grad_ops import (BeamformFn, CovarianceFn, OffdiagAbsFn, SoftmaxRowsFn,
                                      WeightFn)
        from aps_amd.nn_ops import linear
        if store.requires_grad:
            raise NotImplementedError("aps_amd MVDR: no gradient w.r.t. the spectrogram (it is the "
                                      "STFT of the input: data)")
        if x_len is not None:
            x_len = x_len.to(device=store.device, dtype=th.int64).contiguous()
        cov_s, cov_n = CovarianceFn.apply(store, mask_s, mask_n, x_len, self.mask_norm)
        v = OffdiagAbsFn.apply(cov_s)  # N x C x F
        hid = linear(v, self.ref.proj.weight, self.ref.proj.bias, act="tanh")  # N x C x A
        score = linear(hid, self.ref.gvec.weight, self.ref.gvec.bias)  # N x C x 1
        u = SoftmaxRowsFn.apply(score.squeeze(-1))  # N x C
        w = WeightFn.apply(cov_s, cov_n, u, self.eps)
        return BeamformFn.apply(store, w)


@EnhFrontEnds.register("rnn_mask_mvdr")
class RNNMaskMvdr(nn.Module):
    """Mask based MVDR with an RNN mask estimator (mvdr.py:177-234).  The mask network is the
    reference's PyTorchRNNEncoder (Linear+ReLU -> LSTM stack -> Linear -> sigmoid) on
    aps_linear / the persistent LSTM kernels."""

    def __init__(self,
                 enh_input_size: int,
                 num_bins: int = 257,
                 rnn_inp_proj: int = None,
                 rnn: str = "lstm",
                 num_layers: int = 3,
                 dropout: float = 0.0,
                 hidden_size: int = 640,
                 bidirectional: bool = True,
                 mask_net_noise: bool = True,
                 mvdr_att_dim: int = 512,
                 mask_norm: bool = True):
        super(RNNMaskMvdr, self).__init__()
        from aps_amd.asr.base.encoder import PyTorchRNNEncoder
        self.mask_net = PyTorchRNNEncoder(enh_input_size,
                                          num_bins * 2 if mask_net_noise else num_bins,
                                          input_proj=rnn_inp_proj,
                                          rnn=rnn,
                                          num_layers=num_layers,
                                          hidden=hidden_size,
                                          dropout=dropout,
                                          bidirectional=bidirectional,
                                          non_linear="sigmoid")
        self.mvdr_net = MvdrBeamformer(num_bins, att_dim=mvdr_att_dim, mask_norm=mask_norm)
        self.mask_net_noise = mask_net_noise

    def forward(self,
                feats: th.Tensor,
                cstft: ComplexTensor,
                eps: float = 1e-5,
                inp_len: Optional[th.Tensor] = None) -> ComplexTensor:
        """feats N x T x D, cstft complex N x C x F x T -> enhanced complex N x T x F"""
        mask, _ = self.mask_net(feats, inp_len)
        if self.mask_net_noise:
            mask_s, mask_n = th.chunk(mask, 2, dim=-1)
        else:
            mask_s, mask_n = mask, None
        return self.mvdr_net(mask_s, cstft, x_len=inp_len, mask_n=mask_n)

    def beam_weights(self, feats: th.Tensor, cstft: ComplexTensor, inp_len: Optional[th.Tensor] = None):
        """the same up to the beamformer's weights: (store N x C x T x F x 2, w N x F x C x 2) -- for a
        consumer that forms its features in the beamforming pass (EnhASRBase.enhance ->
        beamform_features); inference only"""
        mask, _ = self.mask_net(feats, inp_len)
        if self.mask_net_noise:
            mask_s, mask_n = th.chunk(mask, 2, dim=-1)
        else:
            mask_s, mask_n = mask, None
        store = _store5(cstft)
        _, w = self.mvdr_net.weights_from_masks(store, mask_s, mask_n, inp_len)
        return store, w
