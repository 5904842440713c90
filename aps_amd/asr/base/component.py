"""
Conv2d block (Conv2d -> Norm -> ReLU) and the 2-D normalisation wrapper of
aps/asr/base/component.py:117-142, 251-307 (parameter names `conv`, `norm.norm`).  With BatchNorm
in eval mode the whole block is ONE launch of the channels-last implicit-GEMM convolution
(aps_conv2d_nhwc: conv + folded BatchNorm affine + ReLU); a dilated kernel runs as the equivalent
dense filter (zeros between its taps); an InstanceNorm block is the convolution, the per-(utterance,
channel) normalisation of aps_cmvn_utterance (the same formula as the reference's all-band CMVN)
and a ReLU launch; only InstanceNorm under autograd keeps the torch path.  The output-length
arithmetic is integer exact.
"""
from typing import Tuple, Union

import torch as th
import torch.nn as nn
import torch.nn.functional as tf

from aps_amd.ops import length_map


class Normalize1d(nn.Module):
    """BatchNorm1d / "LN" wrapper on N x T x F (component.py:85-114).  "LN" is GroupNorm(1, F) on
    N x F x T: mean / variance over the whole T x F utterance matrix, then a per-feature affine."""

    def __init__(self, name: str, inp_features: int):
        super(Normalize1d, self).__init__()
        name = name.upper()
        if name not in ["BN", "LN"]:
            raise ValueError(f"Unknown type of Normalize1d: {name}")
        self.norm = nn.BatchNorm1d(inp_features) if name == "BN" else nn.GroupNorm(1, inp_features)

    def __repr__(self) -> str:
        return str(self.norm)

    def affine(self):
        """eval-mode BatchNorm1d as (scale, shift)"""
        bn = self.norm
        if bn.training or bn.running_mean is None:
            raise NotImplementedError("aps_amd: BatchNorm1d forward (eval) path only")
        scale = th.rsqrt(bn.running_var.detach().float() + bn.eps)
        if bn.weight is not None:
            scale = scale * bn.weight.detach().float()
        shift = -bn.running_mean.detach().float() * scale
        if bn.bias is not None:
            shift = shift + bn.bias.detach().float()
        return scale, shift

    def run(self, inp: th.Tensor, relu: bool = False) -> th.Tensor:
        """N x T x F -> N x T x F (+ ReLU)"""
        from aps_amd import _native as nat
        from aps_amd.ops import cmvn_utterance
        m = self.norm
        if (m.training and isinstance(m, nn.BatchNorm1d)) or nat.needs_grad(inp, *m.parameters()):
            # train() / autograd: every link with a HIP backward (aps_amd/grad_ops.py)
            from aps_amd.grad_ops import UtteranceNormFn, activation, batchnorm_rows, row_affine
            if isinstance(m, nn.GroupNorm):
                out = row_affine(UtteranceNormFn.apply(inp, m.eps), m.weight, m.bias)
            else:
                out = batchnorm_rows(inp, m)
            return activation(out, "relu") if relu else out
        if isinstance(m, nn.GroupNorm):
            out = cmvn_utterance(inp, True, True, m.eps)  # utterance statistics kernel
            if m.weight is not None:
                out = th.addcmul(m.bias.detach(), out, m.weight.detach())  # per-feature affine
        else:
            scale, shift = self.affine()
            out = th.addcmul(shift, inp, scale)
        return th.relu_(out) if relu else out

    def forward(self, inp: th.Tensor) -> th.Tensor:
        return self.run(inp)


class Conv1d(nn.Module):
    """TDNN layer: Conv1d -> Norm -> ReLU on N x T x F (component.py:192-248).  With BatchNorm (eval)
    it is one launch of the channels-last conv kernel (H = 1); dilation runs as the equivalent
    dense filter with zero taps."""

    def __init__(self, inp_features: int, out_features: int, kernel_size: int = 3, stride: int = 2,
                 dilation: int = 1, norm: str = "BN", dropout: float = 0,
                 for_streaming: bool = False):
        super(Conv1d, self).__init__()
        padding = 0 if for_streaming else (dilation * (kernel_size - 1)) // 2
        self.conv = nn.Conv1d(inp_features, out_features, kernel_size, stride=stride,
                              padding=padding, dilation=dilation)
        self.norm = Normalize1d(norm, out_features)
        self.drop = nn.Dropout(p=dropout)
        self.stride, self.kernel_size = stride, kernel_size
        self.dilation, self.padding = dilation, padding

    def compute_outp_dim(self, dim: th.Tensor) -> th.Tensor:
        return length_map(dim, 2 * self.padding - self.dilation * (self.kernel_size - 1) - 1,
                          self.stride, 1)

    def forward(self, inp: th.Tensor) -> th.Tensor:
        """N x T x F -> N x T' x O"""
        from aps_amd.grad_ops import dropout
        return dropout(self._run(inp), self.drop)  # drop(relu(norm(conv))), component.py:247

    def _run(self, inp: th.Tensor) -> th.Tensor:
        from aps_amd import _native as nat
        from aps_amd.nn_ops import conv2d_nhwc
        conv = self.conv
        bn = isinstance(self.norm.norm, nn.BatchNorm1d)
        if (bn and self.norm.norm.training) or nat.needs_grad(inp, *self.parameters()):
            return self._trainable_chain(inp)
        w = conv.weight.detach().float().permute(0, 2, 1)[:, None].contiguous()  # Co x 1 x K x Ci
        if self.dilation != 1:
            # a dilated K-tap filter is a dense filter of d (K - 1) + 1 taps with zeros between
            # the live ones: same padding, same outputs, (K_eff / K) x the MACs on the conv kernel
            dense = w.new_zeros(w.shape[0], 1, self.dilation * (self.kernel_size - 1) + 1,
                                w.shape[-1])
            dense[:, :, ::self.dilation] = w
            w = dense
        if bn:
            scale, shift = self.norm.affine()
            if conv.bias is not None:
                shift = shift + conv.bias.detach().float() * scale
            out = conv2d_nhwc(inp[:, None], w, scale.contiguous(), shift.contiguous(),
                              (1, self.stride), (0, self.padding), act="relu")
            return out[:, 0]
        out = conv2d_nhwc(inp[:, None], w, None, conv.bias, (1, self.stride), (0, self.padding))
        return self.norm.run(out[:, 0], relu=True)

    def _trainable_chain(self, inp: th.Tensor) -> th.Tensor:
        """conv -> (+ bias) -> norm (batch statistics in train()) -> ReLU, every link with a HIP
        backward; the Conv1d weight Co x Ci x K enters the channels-last kernel as a view Co x 1 x K x Ci
        of the parameter (a dilated one as its dense equivalent, a differentiable slice assignment)"""
        from aps_amd.nn_ops import conv2d_nhwc
        w = self.conv.weight.permute(0, 2, 1)[:, None]
        if self.dilation != 1:
            dense = w.new_zeros(w.shape[0], 1, self.dilation * (self.kernel_size - 1) + 1, w.shape[-1])
            dense[:, :, ::self.dilation] = w
            w = dense
        y = conv2d_nhwc(inp[:, None], w, None, self.conv.bias, (1, self.stride), (0, self.padding))
        return self.norm.run(y[:, 0], relu=True)


class Normalize2d(nn.Module):
    """BatchNorm2d / InstanceNorm2d wrapper"""

    def __init__(self, name: str, inp_features: int):
        super(Normalize2d, self).__init__()
        name = name.upper()
        if name not in ["BN", "IN"]:
            raise ValueError(f"Unknown type of Normalize2d: {name}")
        self.norm = nn.BatchNorm2d(inp_features) if name == "BN" else nn.InstanceNorm2d(inp_features)

    def __repr__(self) -> str:
        return str(self.norm)

    def forward(self, inp: th.Tensor) -> th.Tensor:
        return self.norm(inp)


class Conv2d(nn.Module):
    """... -> Conv2d -> Norm -> ReLU -> ..."""
    Conv2dParam = Union[int, Tuple[int, int]]

    def __init__(self,
                 in_channels: int,
                 out_channels: int,
                 kernel_size: Conv2dParam = 3,
                 stride: Conv2dParam = 2,
                 dilation: Conv2dParam = 1,
                 norm: str = "BN",
                 for_streaming: bool = False):
        super(Conv2d, self).__init__()

        def int2tuple(inp):
            return (inp, inp) if isinstance(inp, int) else inp

        kernel_size = int2tuple(kernel_size)
        dilation = int2tuple(dilation)
        padding = tuple((d * (k - 1)) // 2 for d, k in zip(dilation, kernel_size))
        if for_streaming:
            padding = (0, padding[-1])
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride,
                              padding=padding, dilation=dilation)
        self.norm = Normalize2d(norm, out_channels)
        self.kernel_size = kernel_size
        self.padding = padding
        self.dilation = dilation
        self.stride = int2tuple(stride)

    def compute_outp_dim(self, dim: th.Tensor, axis: int) -> th.Tensor:
        """output length along `axis`; NB dilation * kernel as in the reference (:290-297)"""
        return length_map(dim, 2 * self.padding[axis] - self.dilation[axis] * self.kernel_size[axis],
                          self.stride[axis], 1)

    def fusible(self) -> bool:
        """does the block run on aps_conv2d_nhwc? (on the GPU: BatchNorm2d in eval mode takes the
        fused launch, train() / autograd the un-fused chain of `run_nhwc`; InstanceNorm2d the
        convolution + aps_cmvn_utterance + ReLU)"""
        return self.conv.weight.is_cuda

    def _inflate(self, w: th.Tensor) -> th.Tensor:
        """Co x Ci x KH x KW -> the dense kernel a dilated convolution is equivalent to: d (k - 1) + 1
        taps per axis, the weights d apart, zeros between them (a differentiable slice assignment)"""
        dh, dw = self.dilation
        if (dh, dw) == (1, 1):
            return w
        Co, Ci, KH, KW = w.shape
        full = w.new_zeros(Co, Ci, (KH - 1) * dh + 1, (KW - 1) * dw + 1)
        full[:, :, ::dh, ::dw] = w
        return full

    def _dense_weight(self) -> th.Tensor:
        """the (inflated) kernel as Co x KH' x KW' x Ci, refreshed when the parameter changes"""
        w = self.conv.weight
        key = (w.data_ptr(), w._version)
        cache = getattr(self, "_dense_cache", None)
        if cache is None or cache[0] != key:
            wd = self._inflate(w.detach().float()).permute(0, 2, 3, 1).contiguous()
            wd._aps_persistent = True  # lives as long as this cache entry (split planes may hang on it)
            cache = (key, wd)
            self._dense_cache = cache
        return cache[1]

    def _trainable_chain(self, inp: th.Tensor) -> th.Tensor:
        """conv -> (+ bias) -> BatchNorm2d (batch statistics in train()) -> ReLU on channels-last
        activations, every link with a HIP backward (aps_amd/grad_ops.py)"""
        from aps_amd.grad_ops import RowBiasAddFn, activation, batchnorm_rows
        from aps_amd.nn_ops import conv2d_nhwc
        w = self._inflate(self.conv.weight).permute(0, 2, 3, 1)  # Co x KH x KW x Ci view of the parameter
        y = conv2d_nhwc(inp, w, None, None, self.stride, self.padding)
        if self.conv.bias is not None:
            y = RowBiasAddFn.apply(y, self.conv.bias)
        return activation(batchnorm_rows(y, self.norm.norm), "relu")

    def _folded(self):
        """(weight Co x KH x KW x Ci, scale, shift) with the conv bias and the eval-mode BatchNorm
        folded into one per-channel affine; refreshed when any source tensor changes"""
        bn, conv = self.norm.norm, self.conv
        parts = [conv.weight, conv.bias, bn.running_mean, bn.running_var, bn.weight, bn.bias]
        key = tuple((t.data_ptr(), t._version) for t in parts if t is not None)
        cache = getattr(self, "_fold_cache", None)
        if cache is None or cache[0] != key:
            scale = th.rsqrt(bn.running_var.detach().float() + bn.eps)
            if bn.weight is not None:
                scale = scale * bn.weight.detach().float()
            shift = -bn.running_mean.detach().float() * scale
            if conv.bias is not None:
                shift = shift + conv.bias.detach().float() * scale
            if bn.bias is not None:
                shift = shift + bn.bias.detach().float()
            cache = (key, self._dense_weight(), scale.contiguous(), shift.contiguous())
            self._fold_cache = cache
        return cache[1:]

    def run_nhwc(self, inp: th.Tensor) -> th.Tensor:
        """channels-last N x T x F x C -> N x T' x F' x C' (one launch)"""
        from aps_amd import _native as nat
        from aps_amd.nn_ops import conv2d_nhwc
        bn = self.norm.norm
        if isinstance(bn, nn.InstanceNorm2d):
            if bn.affine or bn.track_running_stats or nat.needs_grad(inp, *self.parameters()):
                # (never built by the reference's constructor / no HIP adjoint of the normalisation)
                return tf.relu(self.norm(self.conv(inp.permute(0, 3, 1, 2)))).permute(0, 2, 3, 1)
            from aps_amd.grad_ops import activation
            from aps_amd.ops import cmvn_utterance
            # (the conv bias is constant over an (utterance, channel) plane: the normalisation removes it)
            y = conv2d_nhwc(inp, self._dense_weight(), None, None, self.stride, self.padding)
            # N x C x T' x F': every plane contiguous = one "utterance-channel" of the all-band CMVN,
            # (x - mean) / sqrt(mean((x - mean)^2) + eps) = InstanceNorm2d without affine
            z = cmvn_utterance(y.permute(0, 3, 1, 2).contiguous(), True, True, bn.eps)
            return activation(z, "relu").permute(0, 2, 3, 1)
        if bn.training or bn.running_mean is None or nat.needs_grad(inp, *self.parameters()):
            return self._trainable_chain(inp)
        w, scale, shift = self._folded()
        # (the conv2d subsampling of the encoders: the MFMA-sized layers take the fp16 two-plane form,
        # like the DCCRN blocks -- round 2 profiled this layer on the bf16 x 6 form: 300 us per launch)
        return conv2d_nhwc(inp, w, scale, shift, self.stride, self.padding, act="relu", fp16=True)

    def forward(self, inp: th.Tensor) -> th.Tensor:
        """N x C x T x F -> N x C' x T' x F'"""
        inp = inp[:, None] if inp.dim() == 3 else inp
        if self.fusible():
            return self.run_nhwc(inp.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return tf.relu(self.norm(self.conv(inp)))
