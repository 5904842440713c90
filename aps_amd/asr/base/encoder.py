"""
RNN encoder used as the TF-mask estimator of RNNMaskMvdr: (Linear+ReLU) -> RNN stack -> (Linear) ->
(non-linearity), the `pytorch_rnn` encoder of aps/asr/base/encoder.py:87-184 with
`var_len_rnn_forward` (aps/asr/base/component.py:26-55) and the `PyTorchRNN` factory
(component.py:145-190).  Parameter names (`proj`, `impl`, `outp`) follow the reference so
checkpoints load.  The input projection + ReLU and the output projection + non-linearity are one
MFMA GEMM each with the activation in the epilogue (aps_linear*); an nn.LSTM stack of a supported
width runs as one batched input GEMM + one persistent recurrence kernel per layer and direction
(aps_lstm_layer / aps_lstm_stack, csrc/lstm.hip), forward AND backward (grad_ops.LstmFn); GRU, tanh /
relu RNNs, LSTMs of other widths and projected LSTMs run step by step on aps_rnn_step in the forward
pass and step by step backwards under autograd (grad_ops.RnnStepFn / LstmProjStepFn:
aps_rnn_step_backward + the GEMMs; round 4).  `_torch_rnn_under_autograd` -- torch's own recurrent layer on
the GPU -- is reached only by what is not a batch-first torch.nn.RNNBase at all (a user's own recurrent
module handed to var_len_rnn_forward).  CPU tensors raise, like every other op of the package.
"""
from typing import Optional, Tuple

import torch as th
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from aps_amd import _native as nat
from aps_amd.libs import Register
from aps_amd.grad_ops import dropout
from aps_amd.nn_ops import (linear, lstm_forward, lstm_supported, rnn_step_forward,
                            rnn_step_supported, rnn_step_train, rnn_step_trainable)

BaseEncoder = Register("base_encoder")
EncRetType = Tuple[th.Tensor, Optional[th.Tensor]]

rnn_output_nonlinear = {
    "relu": th.relu,
    "sigmoid": th.sigmoid,
    "tanh": th.tanh,
    "none": None,
}


def var_len_rnn_forward(rnn_impl: nn.Module,
                        inp: th.Tensor,
                        inp_len: Optional[th.Tensor] = None,
                        enforce_sorted: bool = False,
                        add_forward_backward: bool = False) -> th.Tensor:
    """N x T x D (+ lengths) -> N x T x H through a packed sequence when lengths are given"""
    if inp.dim() != 3:
        raise ValueError(f"RNN forward needs 3D tensor, got {inp.dim()} instead")
    if lstm_supported(rnn_impl, inp):
        # persistent-kernel recurrence (aps_lstm_layer); padded frames come out as zeros, the
        # time axis is trimmed to the longest utterance like pad_packed_sequence does
        out = lstm_forward(rnn_impl, inp, inp_len)
        # trimming needs max(len) on the host (a sync, like the reference's inp_len.tolist());
        # under stream capture the caller guarantees max(len) == T
        if inp_len is not None and not th.cuda.is_current_stream_capturing():
            out = out[:, :int(inp_len.max())]
        if add_forward_backward:
            prev, last = th.chunk(out, 2, dim=-1)
            out = prev + last
        return out
    if rnn_step_supported(rnn_impl, inp):
        # GRU / tanh- and relu-RNN / LSTMs without a persistent kernel: one cell launch per step
        out = rnn_step_forward(rnn_impl, inp, inp_len)
        if inp_len is not None and not th.cuda.is_current_stream_capturing():
            out = out[:, :int(inp_len.max())]
        if add_forward_backward:
            prev, last = th.chunk(out, 2, dim=-1)
            out = prev + last
        return out
    if not inp.is_cuda:
        raise RuntimeError("aps_amd kernels run on the GPU only (got a CPU tensor); there is no CPU fallback")
    if rnn_step_trainable(rnn_impl, inp):
        # the same recurrences under autograd / in train() mode: BPTT step by step on HIP
        out = rnn_step_train(rnn_impl, inp, inp_len)
        if inp_len is not None and not th.cuda.is_current_stream_capturing():
            out = out[:, :int(inp_len.max())]
        if add_forward_backward:
            prev, last = th.chunk(out, 2, dim=-1)
            out = prev + last
        return out
    return _torch_rnn_under_autograd(rnn_impl, inp, inp_len, enforce_sorted, add_forward_backward)


def _torch_rnn_under_autograd(rnn_impl: nn.Module, inp: th.Tensor, inp_len: Optional[th.Tensor],
                              enforce_sorted: bool, add_forward_backward: bool) -> th.Tensor:
    """The documented torch fall-through (module docstring): a recurrent module that is not a batch-first
    torch.nn.RNNBase runs as itself, packed exactly like the reference does (component.py:26-55)"""
    if inp_len is not None:
        inp = pack_padded_sequence(inp, inp_len.tolist(), batch_first=True,
                                   enforce_sorted=enforce_sorted)
    out, _ = rnn_impl(inp)
    if inp_len is not None:
        out, _ = pad_packed_sequence(out, batch_first=True)
    if add_forward_backward:
        prev, last = th.chunk(out, 2, dim=-1)
        out = prev + last
    return out


def PyTorchRNN(mode: str,
               input_size: int,
               hidden_size: int,
               num_layers: int = 1,
               bias: bool = True,
               dropout: float = 0.,
               proj_size: int = -1,
               bidirectional: bool = False) -> nn.Module:
    """LSTM / GRU / RNN_TANH / RNN_RELU, batch_first"""
    mode = mode.upper()
    kwargs = dict(bias=bias, dropout=dropout, batch_first=True, bidirectional=bidirectional)
    if mode == "LSTM":
        if proj_size > 0:
            kwargs["proj_size"] = proj_size
        return nn.LSTM(input_size, hidden_size, num_layers, **kwargs)
    if mode == "GRU":
        return nn.GRU(input_size, hidden_size, num_layers, **kwargs)
    if mode in ("RNN_TANH", "RNN_RELU"):
        return nn.RNN(input_size, hidden_size, num_layers,
                      nonlinearity="tanh" if mode == "RNN_TANH" else "relu", **kwargs)
    raise ValueError(f"Unsupported RNNs: {mode}")


@BaseEncoder.register("pytorch_rnn")
class PyTorchRNNEncoder(nn.Module):
    """(Linear) -> RNN -> (Linear) -> (NonLinear)"""

    def __init__(self,
                 inp_features: int,
                 out_features: int,
                 input_proj: int = -1,
                 rnn: str = "lstm",
                 num_layers: int = 3,
                 hidden: int = 512,
                 hidden_proj: int = -1,
                 dropout: float = 0.2,
                 bidirectional: bool = False,
                 non_linear: str = "none"):
        super(PyTorchRNNEncoder, self).__init__()
        if non_linear not in rnn_output_nonlinear:
            raise ValueError(f"Unsupported output non-linear function: {non_linear}")
        self.inp_features = inp_features
        self.out_features = out_features
        self.proj = nn.Linear(inp_features, input_proj) if input_proj > 0 else None
        self.impl = PyTorchRNN(rnn,
                               input_proj if input_proj > 0 else inp_features,
                               hidden,
                               num_layers=num_layers,
                               dropout=dropout,
                               proj_size=hidden_proj,
                               bidirectional=bidirectional)
        width = (hidden_proj if hidden_proj > 0 else hidden) * (2 if bidirectional else 1)
        if out_features > 0:
            self.outp = nn.Linear(width, out_features)
            self.non_linear = rnn_output_nonlinear[non_linear]
            self.non_linear_name = non_linear
        else:
            self.outp = None
            self.non_linear = None
            self.non_linear_name = "none"
            self.out_features = width

    def flat(self):
        self.impl.flatten_parameters()

    def forward(self, inp: th.Tensor, inp_len: Optional[th.Tensor]) -> EncRetType:
        """N x T x F -> N x T x out_features"""
        if self.proj is not None:
            inp = linear(inp, self.proj.weight, self.proj.bias, act="relu")
        out = var_len_rnn_forward(self.impl, inp, inp_len=inp_len, enforce_sorted=False)
        if self.outp is not None:
            out = linear(out, self.outp.weight, self.outp.bias, act=self.non_linear_name)
        return out, inp_len


class EncoderBase(nn.Module):
    """inp_features / out_features attributes (encoder.py:87-96)"""

    def __init__(self, inp_features: int, out_features: int):
        super(EncoderBase, self).__init__()
        self.inp_features = inp_features
        self.out_features = out_features


class VariantRNN(nn.Module):
    """-> RNN -> (Linear) -> (Norm) -> (NonLinear) -> (Dropout) -> on N x T x F
    (component.py:389-449; parameter names `rnn`, `proj`, `norm.norm`).  The one-layer recurrence
    runs on the persistent LSTM kernel; projection, an eval-mode BatchNorm (folded into the
    projection's rows) and the non-linearity are ONE GEMM launch.  The "LN" form (GroupNorm over the
    whole utterance matrix) is the utterance-statistics kernel followed by its affine."""

    def __init__(self, input_size: int, rnn: str = "lstm", norm: str = "", hidden: int = 512,
                 project: int = -1, non_linear: str = "relu", dropout: float = 0.0,
                 bidirectional: bool = False, add_forward_backward: bool = False):
        super(VariantRNN, self).__init__()
        from aps_amd.asr.base.component import Normalize1d
        if non_linear not in rnn_output_nonlinear:
            raise ValueError(f"Unsupported non_linear: {non_linear}")
        self.non_linear_name = non_linear
        self.rnn = PyTorchRNN(rnn, input_size, hidden, num_layers=1, dropout=0,
                              bidirectional=bidirectional)
        self.add_forward_backward = add_forward_backward and bidirectional
        if bidirectional and not add_forward_backward:
            hidden *= 2
        self.proj = nn.Linear(hidden, project) if project > 0 else None
        self.norm = Normalize1d(norm, project if project > 0 else hidden) if norm else None
        self.drop = nn.Dropout(dropout) if dropout != 0 else None

    def forward(self, inp: th.Tensor, inp_len: Optional[th.Tensor]) -> th.Tensor:
        """N x Ti x F (+ lengths) -> N x Ti x O"""
        out = self._run(inp, inp_len)
        return out if self.drop is None else dropout(out, self.drop)

    def _run(self, inp: th.Tensor, inp_len: Optional[th.Tensor]) -> th.Tensor:
        out = var_len_rnn_forward(self.rnn, inp, inp_len=inp_len, enforce_sorted=False,
                                  add_forward_backward=self.add_forward_backward)
        act = self.non_linear_name
        batchnorm = self.norm is not None and isinstance(self.norm.norm, nn.BatchNorm1d)
        if self.proj is not None and (self.norm is None or batchnorm):
            w, b = self.proj.weight, self.proj.bias
            if batchnorm:  # y = scale (W x + b) + shift
                scale, shift = self.norm.affine()
                w, b = w * scale[:, None], shift + (0 if b is None else b * scale)
            return linear(out, w, b, act=act)
        if self.proj is not None:
            out = linear(out, self.proj.weight, self.proj.bias)
        if self.norm is not None:
            out = self.norm.run(out)
        fn = rnn_output_nonlinear[act]
        return out if fn is None else fn(out)


@BaseEncoder.register("variant_rnn")
class VariantRNNEncoder(nn.Module):
    """stack of VariantRNN layers, optionally pyramidal (every layer after the first sees pairs of
    frames side by side at half the rate) (encoder.py:225-308)"""

    def __init__(self, inp_features: int, out_features: int, rnn: str = "lstm", hidden: int = 512,
                 num_layers: int = 3, bidirectional: bool = True, dropout: float = 0.0,
                 dropout_input: bool = True, project: int = -1, non_linear: str = "tanh",
                 norm: str = "", pyramid_stack: bool = False,
                 add_forward_backward: bool = False):
        super(VariantRNNEncoder, self).__init__()
        self.inp_features = inp_features
        factor = 2 if (bidirectional and not add_forward_backward) else 1

        def width_into(layer: int) -> int:
            if layer == 0:
                return inp_features
            if project > 0:
                return project  # (sic) not doubled by pyramid_stack, encoder.py:252-253
            return hidden * factor * (2 if pyramid_stack else 1)

        self.pyramid = pyramid_stack
        self.out_features = out_features if out_features > 0 else hidden * factor
        last = num_layers - 1
        self.enc_layers = nn.ModuleList([
            VariantRNN(width_into(i), rnn=rnn, norm=norm if i != last else "", hidden=hidden,
                       project=project if i != last else self.out_features,
                       dropout=dropout if i != last else 0, bidirectional=bidirectional,
                       non_linear=non_linear if i != last else "none",
                       add_forward_backward=add_forward_backward) for i in range(num_layers)
        ])

    def forward(self, inp: th.Tensor, inp_len: Optional[th.Tensor]) -> EncRetType:
        """N x Ti x F -> N x To x O"""
        for i, layer in enumerate(self.enc_layers):
            if i != 0 and self.pyramid:
                if inp.shape[1] % 2:
                    inp = inp[:, :-1]
                inp = th.cat([inp[:, ::2], inp[:, 1::2]], -1)
                inp_len = None if inp_len is None else inp_len // 2
            inp = layer(inp, inp_len)
        return inp, inp_len


@BaseEncoder.register("conv1d")
class Conv1dEncoder(EncoderBase):
    """stack of TDNN (conv1d) layers with optional time reduction (encoder.py:310-364)"""

    def __init__(self, inp_features: int, out_features: int, dim: int = 512, norm: str = "BN",
                 num_layers: int = 3, kernel=3, stride=2, dilation=1, dropout: float = 0,
                 for_streaming: bool = False):
        super(Conv1dEncoder, self).__init__(inp_features, out_features)
        from aps_amd.asr.base.component import Conv1d

        def int2list(param, repeat):
            return [param] * repeat if isinstance(param, int) else param

        self.kernel = int2list(kernel, num_layers)
        self.stride = int2list(stride, num_layers)
        self.dilation = int2list(dilation, num_layers)
        self.out_features = out_features if out_features > 0 else dim
        self.enc_layers = nn.ModuleList([
            Conv1d(inp_features if i == 0 else dim,
                   dim if i != num_layers - 1 else self.out_features, norm=norm,
                   kernel_size=self.kernel[i], stride=self.stride[i], dilation=self.dilation[i],
                   dropout=dropout, for_streaming=for_streaming) for i in range(num_layers)
        ])
        self.out_features = dim  # (sic) as in the reference, encoder.py:347

    def forward(self, inp: th.Tensor, inp_len: Optional[th.Tensor]) -> EncRetType:
        """N x Ti x F -> N x To x O"""
        for conv1d in self.enc_layers:
            inp = conv1d(inp)
            if inp_len is not None:
                inp_len = conv1d.compute_outp_dim(inp_len)
        return inp, inp_len


@BaseEncoder.register("conv2d")
class Conv2dEncoder(EncoderBase):
    """Stack of Conv2d blocks with time reduction + output projection (encoder.py:367-441): one
    channels-last implicit-GEMM launch per block (conv + BatchNorm + ReLU fused), activations stay
    N x T x F x C between blocks, and the output projection (fp32 MFMA GEMM) consumes them directly
    through a column-permuted copy of its weight (the reference flattens channel-major)."""

    def __init__(self,
                 inp_features: int,
                 out_features: int,
                 channel=32,
                 in_channels: int = 1,
                 norm: str = "BN",
                 num_layers: int = 3,
                 kernel=3,
                 stride=2,
                 for_streaming: bool = False):
        super(Conv2dEncoder, self).__init__(inp_features, out_features)
        from aps_amd.asr.base.component import Conv2d

        def param2need(param, num_layers):
            if isinstance(param, int):
                return [(param, param)] * num_layers
            if isinstance(param[0], int):
                return [(p, p) for p in param]
            return param

        self.kernel = param2need(kernel, num_layers)
        self.stride = param2need(stride, num_layers)
        if isinstance(channel, int):
            channel = [channel] * num_layers
        self.enc_layers = nn.ModuleList([
            Conv2d(in_channels if i == 0 else channel[i - 1], channel[i],
                   kernel_size=self.kernel[i], norm=norm, stride=self.stride[i],
                   for_streaming=for_streaming) for i in range(num_layers)
        ])
        freq_dim = th.IntTensor([inp_features])
        for conv2d in self.enc_layers:
            freq_dim = conv2d.compute_outp_dim(freq_dim, 1)
        freq_x_channel = freq_dim.item() * channel[-1]
        if out_features > 0:
            self.out_features = out_features
            self.outp = nn.Linear(freq_x_channel, out_features)
        else:
            self.out_features = freq_x_channel
            self.outp = None

    def forward(self, inp: th.Tensor, inp_len: Optional[th.Tensor]) -> EncRetType:
        """N x (C) x T x F -> N x T' x D"""
        inp = inp[:, None] if inp.dim() == 3 else inp
        if all(c.fusible() for c in self.enc_layers) and inp.is_cuda:
            x = inp.permute(0, 2, 3, 1)  # N x T x F x C (a view; C = 1 is already contiguous)
            for conv2d in self.enc_layers:
                x = conv2d.run_nhwc(x)
                if inp_len is not None:
                    inp_len = conv2d.compute_outp_dim(inp_len, 0)
            N, T, Fo, Co = x.shape
            if self.outp is None:  # reference feature order: channel-major
                return x.permute(0, 1, 3, 2).reshape(N, T, -1), inp_len
            if nat.needs_grad(x, *self.outp.parameters()):
                # autograd: the column permutation as a differentiable view chain of the parameter
                wp = self.outp.weight.view(-1, Co, Fo).transpose(1, 2).reshape(-1, Fo * Co)
            else:
                wp = self._outp_weight_nhwc(Fo, Co)
            out = linear(x.reshape(N, T, Fo * Co), wp, self.outp.bias)
            return out, inp_len
        for conv2d in self.enc_layers:
            inp = conv2d(inp)
            if inp_len is not None:
                inp_len = conv2d.compute_outp_dim(inp_len, 0)
        N, _, T, _ = inp.shape
        out = inp.transpose(1, 2).contiguous().view(N, T, -1)
        if self.outp is not None:
            out = linear(out, self.outp.weight, self.outp.bias)
        return out, inp_len

    def _outp_weight_nhwc(self, Fo: int, Co: int) -> th.Tensor:
        """outp.weight with its input columns reordered from (c, f) to (f, c)"""
        w = self.outp.weight
        key = (w.data_ptr(), w._version, Fo, Co)
        cache = getattr(self, "_outp_cache", None)
        if cache is None or cache[0] != key:
            wp = w.detach().float().view(-1, Co, Fo).transpose(1, 2).reshape(-1, Fo * Co).contiguous()
            wp._aps_persistent = True  # lives as long as this cache entry: derived images may hang on it
            cache = (key, wp)
            self._outp_cache = cache
        return cache[1]
