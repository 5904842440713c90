"""
Geometry-dependent layers of the enhancement front end: DfTransform (angle / directional feature)
and FixedBeamformer -- the surface of aps/transform/enh.py:146-384, on the kernels of
csrc/spatial.hip (aps_directional_feature, aps_fixed_beamform).  Parameter names and shapes are the
reference's (`omega` [1, F]; `real` / `imag` [B, C, F, 1]), so its checkpoints load unchanged.
"""
import ctypes as C
import math
from typing import List, Optional, Tuple, Union

import torch as th
import torch.nn as nn

from aps_amd import _native as nat


def _split_index(sstr: str) -> Tuple[List[int], List[int]]:
    pair = [tuple(map(int, p.split(","))) for p in sstr.split(";")]
    return [t[0] for t in pair], [t[1] for t in pair]


class DfTransform(nn.Module):
    """
    Angle / directional feature (enh.py:146-300): the mean over microphone pairs of
    cos(observed IPD - the IPD a source at the given DoA would produce on the array).
        num_doas == 1: the DoA of the target speaker(s) is given
        num_doas != 1: the DoA argument is ignored, num_doas directions are sampled on the circle

    Args:
        geometric: array geometry, "7@" (centre + 6 on a 4.25 cm circle) as in the reference
        sr, velocity, num_bins: sample rate, speed of sound, FFT bins
        num_doas: how many directions to point at
        af_index: microphone pairs "l,r;l,r;..."
    """
    RADIUS = {"7@": 0.0425}

    def __init__(self, geometric: str = "7@", sr: int = 16000, velocity: int = 340,
                 num_bins: int = 257, num_doas: int = 1,
                 af_index: str = "1,0;2,0;3,0;4,0;5,0;6,0") -> None:
        super().__init__()
        if geometric not in self.RADIUS:
            raise RuntimeError(f"Unsupported array geometric: {geometric}")
        self.geometric = geometric
        self.sr = sr
        self.num_bins = num_bins
        self.num_doas = num_doas
        self.velocity = velocity
        self.index_l, self.index_r = _split_index(af_index)
        self.af_index = af_index
        omega = th.tensor([math.pi * sr * f / (num_bins - 1) for f in range(num_bins)])
        self.omega = nn.Parameter(omega[None, :], requires_grad=False)  # 1 x F

    def extra_repr(self) -> str:
        return (f"geometric={self.geometric}, af_index={self.af_index}, sr={self.sr}, "
                f"num_bins={self.num_bins}, velocity={self.velocity}, "
                f"known_doa={self.num_doas == 1}")

    def exportable(self) -> bool:
        return True

    def _sampled_doas(self, device) -> th.Tensor:
        # enh.py:204-209: num_doas directions on [0, 2 pi)
        return th.linspace(0, math.pi * 2, self.num_doas + 1, device=device)[:-1].contiguous()

    def forward(self, p: th.Tensor, doa: Union[th.Tensor, List[th.Tensor]]) -> th.Tensor:
        """
        Args:
            p: phase, (N x) C x T x F
            doa: DoA of the target / of each speaker (N, or a list of N), ignored if num_doas != 1
        Return:
            af: N x T x F (x speakers along F) or N x D x T x F (num_doas != 1)
        """
        if p.dim() not in [3, 4]:
            raise RuntimeError(f"{self.__class__.__name__} expect 3/4D tensor, but got {p.dim():d} "
                               "instead")
        if p.dim() == 3:
            p = p.unsqueeze(0)
        speakers = doa if isinstance(doa, list) else [doa]
        if isinstance(doa, list) and self.num_doas != 1:
            raise RuntimeError("known_doa=False, no need to pass doa as a Sequence object")
        nat.require_device(p, self.omega, *speakers)
        lib = nat.load()
        p = nat.f32c(p)
        N, Cn, T, F = p.shape
        if F != self.num_bins:
            raise RuntimeError(f"phase has {F} bins, the layer was built for {self.num_bins}")
        if max(self.index_l + self.index_r) >= Cn:
            raise IndexError(f"af_index {self.af_index} exceeds the {Cn} channels of the phase")
        P = len(self.index_l)
        idx_l = (C.c_int32 * P)(*self.index_l)
        idx_r = (C.c_int32 * P)(*self.index_r)
        neg_omega = (-self.omega).reshape(-1).contiguous()
        S, D = len(speakers), self.num_doas
        if D == 1:
            out = th.empty(N, T, S * F, device=p.device, dtype=th.float32)
        else:
            out = th.empty(N, D, T, F, device=p.device, dtype=th.float32)
            grid = self._sampled_doas(p.device)
        for s, spk in enumerate(speakers):
            if D == 1:
                angles, stride = nat.f32c(spk.reshape(-1)), 1
                if angles.numel() != N:
                    raise RuntimeError(f"doa has {angles.numel()} entries for {N} utterances")
            else:
                angles, stride = grid, 0
            rc = lib.aps_directional_feature(nat.ptr(p), nat.ptr(angles), stride,
                                             nat.ptr(neg_omega), idx_l, idx_r, P, nat.ptr(out),
                                             S * F, s * F, N, Cn, T, F, D,
                                             float(self.RADIUS[self.geometric]),
                                             float(self.velocity), nat.stream_of(p))
            nat.check(rc, "aps_directional_feature")
        return out


class FixedBeamformer(nn.Module):
    """
    Fixed beamformer as a layer (enh.py:303-384).

    Args:
        num_beams, num_channels, num_bins: B, C, F of the coefficient tensor
        weight: path of a saved coefficient tensor (2, B, C, F), else random initialisation
        requires_grad: train the coefficients (grad_ops.FixedBeamFn: aps_fixed_beamform_backward)
    """

    def __init__(self, num_beams: int, num_channels: int, num_bins: int,
                 weight: Optional[str] = None, requires_grad: bool = False) -> None:
        super().__init__()
        if weight:
            w = th.load(weight)
            if w.shape[1] != num_beams:
                raise RuntimeError(f"Number of beam got from {w.shape[1]} don't match parameter "
                                   f"{num_beams}")
            self.init_weight = weight
        else:
            self.init_weight = None
            w = th.zeros(2, num_beams, num_channels, num_bins)
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.real = nn.Parameter(w[0].unsqueeze(-1), requires_grad=requires_grad)  # B x C x F x 1
        self.imag = nn.Parameter(w[1].unsqueeze(-1), requires_grad=requires_grad)
        self.requires_grad = requires_grad

    def extra_repr(self) -> str:
        B, M, F, _ = self.real.shape
        return (f"num_beams={B}, num_channels={M}, num_bins={F}, init_weight={self.init_weight}, "
                f"requires_grad={self.requires_grad}")

    def exportable(self) -> bool:
        return True

    def forward(self, real: th.Tensor, imag: th.Tensor,
                beam: Union[None, int, th.Tensor] = None, squeeze: bool = False,
                trans: bool = False, cplx: bool = True) -> Tuple[th.Tensor, th.Tensor]:
        """
        Args:
            real, imag: N x C x F x T
            beam: None (all beams), one beam index, or N indices (one per utterance)
        Return:
            real, imag: N x (B) x F x T
        """
        r, i = real, imag
        if r.dim() != 4 or i.dim() != 4:
            raise RuntimeError(f"FixBeamformer accept 4D tensor, got {r.dim()}")
        if self.real.shape[1] != r.shape[1]:
            raise RuntimeError(f"Number of channels mismatch: {r.shape[1]} vs {self.real.shape[1]}")
        train = nat.needs_grad(real, imag, self.real, self.imag)
        nat.require_device(r.detach(), i.detach(), self.real.detach(), self.imag.detach())
        lib = nat.load()
        r, i = nat.f32c(r.detach()), nat.f32c(i.detach())
        N, Cn, F, T = r.shape
        B = self.real.shape[0]
        if self.real.shape[2] != F:
            raise RuntimeError(f"Number of bins mismatch: {F} vs {self.real.shape[2]}")
        sel = None
        if beam is not None:
            sel = th.as_tensor(beam, device=r.device, dtype=th.int64).reshape(-1)
            if sel.numel() == 1:
                sel = sel.expand(N)
            if sel.numel() != N:
                raise RuntimeError(f"beam has {sel.numel()} entries for {N} utterances")
            sel = sel.contiguous()
            # host-side range check only for host-side indices (a device tensor is trusted like
            # the reference's advanced indexing, which raises asynchronously)
            if not isinstance(beam, th.Tensor) and not 0 <= int(beam) < B:
                raise IndexError(f"beam {beam} out of range for {B} beams")
        if train:
            # trainable coefficients (requires_grad=True) or a differentiable producer: HIP adjoint
            from aps_amd.grad_ops import FixedBeamFn
            br, bi = FixedBeamFn.apply(real, imag, self.real, self.imag, sel)
        else:
            shape = (N, F, T) if sel is not None else (N, B, F, T)
            br = th.empty(*shape, device=r.device, dtype=th.float32)
            bi = th.empty(*shape, device=r.device, dtype=th.float32)
            rc = lib.aps_fixed_beamform(nat.ptr(r), nat.ptr(i), nat.ptr(nat.f32c(self.real)),
                                        nat.ptr(nat.f32c(self.imag)), nat.ptr(sel), nat.ptr(br),
                                        nat.ptr(bi), N, Cn, F, T, B, nat.stream_of(r))
            nat.check(rc, "aps_fixed_beamform")
        if squeeze:
            br, bi = br.squeeze(), bi.squeeze()
        if trans:
            br, bi = br.transpose(-1, -2), bi.transpose(-1, -2)
        return br, bi
