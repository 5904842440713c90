"""
Complex / real convolutional UNet blocks (aps/sse/enh/dcunet.py:16-275) for DCCRN.

Parameters keep the reference's names and shapes (`block.0.real` / `.imag` Conv2d or
ConvTranspose2d, `block.1.real_bn` / `.imag_bn` BatchNorm2d), so checkpoints load; the forward path
does not call them.  Activations travel channels-last, N x T x F x 2C with the real channels first
(the reference stacks real | imag along the frequency axis of N x C x 2F x T), which makes

  * a complex layer ONE real convolution with the block weight [[Wr, -Wi], [Wi, Wr]],
  * conv + (complex) BatchNorm2d (eval) + LeakyReLU one launch of aps_conv2d_nhwc,
  * the decoder's skip connection x + enc_h the residual input of the producing launch
    ("sum"; a "cat" connection concatenates channels-last and the consuming layer's folded weight
    has its input axis ordered to match),
  * the causal variant (time padding k - 1 on both sides, last k - 1 frames cut,
    dcunet.py:90-100, 118-133) a launch that does not compute the cut frames.

Real-valued blocks (cplx = False) are the same launches on plain Conv2d / BatchNorm2d parameters.

Training (train() mode, or any parameter / input that requires grad): the block is the convolution
with its adjoints (grad_ops.Conv2dNhwcFn: forward and transposed form, bias, causal crop), the
BatchNorm over the rows of the channels-last activation with batch statistics (one pass over the
2C real | imag channels: the reference's real_bn / imag_bn are independent per-channel norms, their
running statistics are the two halves), LeakyReLU and the skip addition as stand-alone passes that
keep what their backward needs.  The block weight [[Wr, -Wi], [Wi, Wr]] is assembled from the
parameters by differentiable concatenation, so the gradients land on `real.weight` / `imag.weight`.
"""
from typing import List, Optional, Tuple

import torch as th
import torch.nn as nn

from aps_amd import _native as nat
from aps_amd.nn_ops import conv2d_nhwc


def parse_1dstr(sstr: str) -> List[int]:
    return list(map(int, sstr.split(",")))


def parse_2dstr(sstr: str) -> List[List[int]]:
    return [parse_1dstr(tok) for tok in sstr.split(";")]


class _Pair(nn.Module):
    """holder of a `real` / `imag` pair of torch layers (parameters only)"""

    def __init__(self, cls, *args, **kwargs):
        super(_Pair, self).__init__()
        self.real = cls(*args, **kwargs)
        self.imag = cls(*args, **kwargs)


class ComplexConv2d(_Pair):
    def __init__(self, *args, **kwargs):
        super(ComplexConv2d, self).__init__(nn.Conv2d, *args, **kwargs)


class ComplexConvTranspose2d(_Pair):
    def __init__(self, *args, **kwargs):
        super(ComplexConvTranspose2d, self).__init__(nn.ConvTranspose2d, *args, **kwargs)


class ComplexBatchNorm2d(nn.Module):
    def __init__(self, *args, **kwargs):
        super(ComplexBatchNorm2d, self).__init__()
        self.real_bn = nn.BatchNorm2d(*args, **kwargs)
        self.imag_bn = nn.BatchNorm2d(*args, **kwargs)


class CasualTruncated(nn.Module):
    """drops the last `padding` frames (dcunet.py:90-100); in the kernel path those frames are not
    computed in the first place"""

    def __init__(self, casual_padding: int) -> None:
        super(CasualTruncated, self).__init__()
        self.padding = casual_padding

    def forward(self, inp: th.Tensor) -> th.Tensor:
        return inp[..., :-self.padding]


def _bn_affine(bn: nn.BatchNorm2d) -> Tuple[th.Tensor, th.Tensor]:
    if bn.training or bn.running_mean is None:
        raise RuntimeError("_bn_affine folds running statistics (eval mode); training-mode blocks "
                           "run _Block._train_run")
    scale = th.rsqrt(bn.running_var.detach().float() + bn.eps)
    if bn.weight is not None:
        scale = scale * bn.weight.detach().float()
    shift = -bn.running_mean.detach().float() * scale
    if bn.bias is not None:
        shift = shift + bn.bias.detach().float()
    return scale, shift


class _Block(nn.Module):
    """conv (or transposed conv) [+ BatchNorm + LeakyReLU]; `self.block` mirrors the reference's
    nn.Sequential indices"""

    transposed = False
    cat_input = False  # input = two tensors concatenated channels-last ("cat" connection)
    crop_t = 0         # causal: trailing frames that are cut

    def _layout(self, w: th.Tensor) -> th.Tensor:
        """torch weight -> Co x KT x KF x Ci (the kernel's H axis is time, its W axis frequency)"""
        if self.transposed:  # [Ci, Co, KF, KT]
            return w.permute(1, 3, 2, 0)
        return w.permute(0, 3, 2, 1)  # [Co, Ci, KF, KT]

    def _folded(self):
        conv = self.block[0]
        norms = [m for m in self.block if isinstance(m, (ComplexBatchNorm2d, nn.BatchNorm2d))]
        norm = norms[0] if norms else None
        tensors = list(conv.parameters()) + ([] if norm is None else list(norm.parameters()) +
                                             list(norm.buffers()))
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        cache = getattr(self, "_fold_cache", None)
        if cache is not None and cache[0] == key:
            return cache[1:]
        if self.cplx:
            wr, wi = self._layout(conv.real.weight.detach().float()), \
                self._layout(conv.imag.weight.detach().float())
            # rows = output (real | imag), last axis = input (real | imag)
            w = th.cat([th.cat([wr, -wi], -1), th.cat([wi, wr], -1)], 0)
            if self.cat_input:
                # the data arrives as [r1 | i1 | r2 | i2] (two real|imag tensors side by side), the
                # block weight's input axis is [r1 r2 | i1 i2]
                c = wr.shape[-1] // 2
                idx = th.arange(4 * c, device=w.device).view(2, 2, c).transpose(0, 1).reshape(-1)
                w = w[..., idx]
            w = w.contiguous()
            br = conv.real.bias.detach().float() if conv.real.bias is not None else 0
            bi = conv.imag.bias.detach().float() if conv.imag.bias is not None else 0
            bias = th.cat([br - bi + th.zeros_like(wr[:, 0, 0, 0]),
                           br + bi + th.zeros_like(wr[:, 0, 0, 0])])
            if norm is not None:
                sr, tr = _bn_affine(norm.real_bn)
                si, ti = _bn_affine(norm.imag_bn)
                scale, shift = th.cat([sr, si]), th.cat([tr, ti])
        else:
            w = self._layout(conv.weight.detach().float()).contiguous()
            bias = conv.bias.detach().float() if conv.bias is not None else th.zeros_like(w[:, 0, 0, 0])
            if norm is not None:
                scale, shift = _bn_affine(norm)
        if norm is None:
            scale, shift = None, bias.contiguous()
        else:
            shift = (shift + bias * scale).contiguous()
            scale = scale.contiguous()
        w._aps_persistent = True  # lives as long as this cache entry (split planes may hang on it)
        self._fold_cache = (key, w, scale, shift)
        return w, scale, shift

    def _norm(self):
        norms = [m for m in self.block if isinstance(m, (ComplexBatchNorm2d, nn.BatchNorm2d))]
        return norms[0] if norms else None

    def _train_weight(self) -> Tuple[th.Tensor, Optional[th.Tensor]]:
        """the block weight and bias as differentiable functions of the parameters"""
        conv = self.block[0]
        if not self.cplx:
            return self._layout(conv.weight).contiguous(), conv.bias
        wr, wi = self._layout(conv.real.weight), self._layout(conv.imag.weight)
        w = th.cat([th.cat([wr, -wi], -1), th.cat([wi, wr], -1)], 0)
        if self.cat_input:  # (see _folded)
            c = wr.shape[-1] // 2
            idx = th.arange(4 * c, device=w.device).view(2, 2, c).transpose(0, 1).reshape(-1)
            w = w[..., idx]
        br, bi = conv.real.bias, conv.imag.bias
        bias = None if br is None else th.cat([br - bi, br + bi])
        return w.contiguous(), bias

    def _train_norm(self, y: th.Tensor, norm) -> th.Tensor:
        """BatchNorm2d (batch statistics in train(), running statistics in eval()) on the
        channels-last activation, differentiable"""
        from aps_amd.grad_ops import BatchNormRowsFn, batchnorm_rows
        if not self.cplx:
            return batchnorm_rows(y, norm)
        re, im = norm.real_bn, norm.imag_bn
        training = re.training or re.running_mean is None
        tracked = re.running_mean is not None
        momentum = 0.0
        if training and tracked:
            for bn in (re, im):
                if bn.num_batches_tracked is not None:
                    bn.num_batches_tracked.add_(1)
            momentum = re.momentum if re.momentum is not None else \
                1.0 / float(re.num_batches_tracked)
        cat = lambda a, b: None if a is None else th.cat([a, b])  # noqa: E731
        mean, var = cat(re.running_mean, im.running_mean), cat(re.running_var, im.running_var)
        out = BatchNormRowsFn.apply(y, cat(re.weight, im.weight), cat(re.bias, im.bias), mean, var,
                                    training, momentum, re.eps)
        if training and tracked:  # the kernel updated the concatenated copies
            c = re.running_mean.shape[0]
            with th.no_grad():
                re.running_mean.copy_(mean[:c]), im.running_mean.copy_(mean[c:])
                re.running_var.copy_(var[:c]), im.running_var.copy_(var[c:])
        return out

    def _train_run(self, x: th.Tensor, residual: Optional[th.Tensor]) -> th.Tensor:
        from aps_amd.grad_ops import ScaleAddFn, activation
        w, bias = self._train_weight()
        y = conv2d_nhwc(x, w, None, bias, stride=self.stride_tf, padding=self.padding_tf,
                        transposed=self.transposed, output_padding=self.outpad_tf,
                        crop=(self.crop_t, 0))
        norm = self._norm()
        if norm is not None:
            y = self._train_norm(y, norm)
        if any(isinstance(m, nn.LeakyReLU) for m in self.block):
            y = activation(y, "leaky_relu")
        return y if residual is None else ScaleAddFn.apply(y, residual, 1.0)

    def _in_training(self, x: th.Tensor, residual: Optional[th.Tensor]) -> bool:
        norm = self._norm()
        bns = [] if norm is None else ([norm.real_bn] if self.cplx else [norm])
        return any(bn.training or bn.running_mean is None for bn in bns) or \
            nat.needs_grad(x, residual, *self.parameters())

    def run(self, x: th.Tensor, residual: Optional[th.Tensor] = None) -> th.Tensor:
        """channels-last N x T x F x C' -> N x T x F' x C'' (+ residual: the next layer's skip)"""
        if self._in_training(x, residual):
            return self._train_run(x, residual)
        w, scale, shift = self._folded()
        act = "leaky_relu" if any(isinstance(m, nn.LeakyReLU) for m in self.block) else None
        return conv2d_nhwc(x, w, scale, shift, stride=self.stride_tf, padding=self.padding_tf,
                           transposed=self.transposed, output_padding=self.outpad_tf, act=act,
                           slope=0.01, residual=residual, crop=(self.crop_t, 0), fp16=True)

    def forward(self, x: th.Tensor) -> th.Tensor:
        """reference layout N x C x (2)F x T -> N x C' x (2)F' x T"""
        return from_nhwc(self.run(to_nhwc(x, self.cplx)), self.cplx)


def to_nhwc(x: th.Tensor, cplx: bool) -> th.Tensor:
    """N x C x (2)F x T -> N x T x F x (2)C, real channels first"""
    if cplx:
        xr, xi = th.chunk(x, 2, -2)
        x = th.cat([xr, xi], 1)
    return x.permute(0, 3, 2, 1).contiguous()


def from_nhwc(x: th.Tensor, cplx: bool) -> th.Tensor:
    x = x.permute(0, 3, 2, 1)
    if cplx:
        xr, xi = th.chunk(x, 2, 1)
        x = th.cat([xr, xi], -2)
    return x


class EncoderBlock(_Block):
    """Conv2d -> BatchNorm2d -> LeakyReLU (dcunet.py:103-143)"""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: Tuple[int],
                 stride: int = 1, padding: int = 0, causal: bool = False, cplx: bool = True) -> None:
        super(EncoderBlock, self).__init__()
        time_axis_pad = kernel_size[-1] - 1
        if not causal:
            time_axis_pad = time_axis_pad // 2
        pad = (padding, time_axis_pad)
        ConvClass = ComplexConv2d if cplx else nn.Conv2d
        NormClass = ComplexBatchNorm2d if cplx else nn.BatchNorm2d
        block = [ConvClass(in_channels, out_channels, tuple(kernel_size), stride=tuple(stride),
                           padding=pad)]
        if causal:
            block += [CasualTruncated(time_axis_pad)]
        block += [NormClass(out_channels), nn.LeakyReLU()]
        self.block = nn.Sequential(*block)  # same indices as the reference's Sequential
        self.cplx = cplx
        self.crop_t = time_axis_pad if causal else 0
        self.stride_tf = (stride[1], stride[0])
        self.padding_tf = (pad[1], pad[0])
        self.outpad_tf = (0, 0)


class DecoderBlock(_Block):
    """ConvTranspose2d [-> BatchNorm2d -> LeakyReLU] (dcunet.py:146-192)"""

    transposed = True

    def __init__(self, in_channels: int, out_channels: int, kernel_size: Tuple[int],
                 stride: int = 1, padding: int = 0, output_padding: int = 0, causal: bool = False,
                 cplx: bool = True, last_layer: bool = False, cat_input: bool = False) -> None:
        super(DecoderBlock, self).__init__()
        time_axis_pad = kernel_size[-1] - 1
        if not causal:
            time_axis_pad = time_axis_pad // 2
        pad = (padding, kernel_size[1] - 1 - time_axis_pad)
        ConvClass = ComplexConvTranspose2d if cplx else nn.ConvTranspose2d
        NormClass = ComplexBatchNorm2d if cplx else nn.BatchNorm2d
        block = [ConvClass(in_channels, out_channels, tuple(kernel_size), stride=tuple(stride),
                           padding=pad, output_padding=(output_padding, 0))]
        if causal:
            block += [CasualTruncated(time_axis_pad)]
        if not last_layer:
            block += [NormClass(out_channels), nn.LeakyReLU()]
        self.block = nn.Sequential(*block)
        self.cplx = cplx
        self.stride_tf = (stride[1], stride[0])
        self.padding_tf = (pad[1], pad[0])
        self.outpad_tf = (0, output_padding)
        self.crop_t = time_axis_pad if causal else 0
        self.cat_input = cat_input and cplx


class Encoder(nn.Module):
    """encoder of the UNet (dcunet.py:195-229): returns the skip list and the bottleneck"""

    def __init__(self, cplx: bool, K, S, C, P, causal: bool = False) -> None:
        super(Encoder, self).__init__()
        self.layers = nn.ModuleList([
            EncoderBlock(C[i], C[i + 1], k, stride=S[i], padding=P[i], cplx=cplx, causal=causal)
            for i, k in enumerate(K)
        ])
        self.num_layers = len(self.layers)
        self.cplx = cplx

    def run(self, x: th.Tensor):
        enc_h = []
        for index, layer in enumerate(self.layers):
            x = layer.run(x)
            if index + 1 != self.num_layers:
                enc_h.append(x)
        return enc_h, x

    def forward(self, x: th.Tensor):
        enc_h, h = self.run(to_nhwc(x, self.cplx))
        return [from_nhwc(e, self.cplx) for e in enc_h], from_nhwc(h, self.cplx)


class Decoder(nn.Module):
    """decoder of the UNet (dcunet.py:232-275).  "sum" connections: layer i's launch adds the skip
    enc_h[i] that the next layer's input needs; "cat": the skip is concatenated channels-last"""

    def __init__(self, cplx: bool, K, S, C, P, O, causal: bool = False,
                 connection: str = "sum") -> None:
        super(Decoder, self).__init__()
        if connection not in ["cat", "sum"]:
            raise ValueError(f"Unknown connection mode: {connection}")
        cat = connection == "cat"
        self.layers = nn.ModuleList([
            DecoderBlock(C[i] * 2 if cat and i != 0 else C[i], C[i + 1], k, stride=S[i],
                         padding=P[i], output_padding=O[i], causal=causal, cplx=cplx,
                         last_layer=(i == len(K) - 1),
                         cat_input=cat and i != 0)
            for i, k in enumerate(K)
        ])
        self.connection = connection
        self.cplx = cplx

    def run(self, x: th.Tensor, enc_h: List[th.Tensor]) -> th.Tensor:
        last = len(self.layers) - 1
        for index, layer in enumerate(self.layers):
            if self.connection == "sum":
                x = layer.run(x, residual=enc_h[index] if index != last else None)
            else:
                x = layer.run(x if index == 0 else th.cat([x, enc_h[index - 1]], -1))
        return x

    def forward(self, x: th.Tensor, enc_h: List[th.Tensor]) -> th.Tensor:
        skips = [to_nhwc(e, self.cplx) for e in enc_h]
        return from_nhwc(self.run(to_nhwc(x, self.cplx), skips), self.cplx)
