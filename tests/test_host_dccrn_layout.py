"""
CPU: the channels-last layout / weight folding logic of the DCCRN host code (real | imag channel
stacking, complex block weights, "cat" input ordering, causal crop, per-speaker decoders) checked
against the CPU oracle with the device kernels EMULATED by torch ops inside this test (the product
itself has no CPU path: the emulation is patched in here, test-side only).
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import golden
from tests.test_oracle_encoder import DCCRN_SMALL, DCCRN_VARIANTS


def _conv_emul(x, weight, scale=None, shift=None, stride=(1, 1), padding=(0, 0), transposed=False,
               output_padding=(0, 0), act=None, slope=0.01, residual=None, crop=(0, 0), fp16=False):
    """aps_conv2d_nhwc's contract (include/aps_amd.h) in torch"""
    xn = x.permute(0, 3, 1, 2)
    if transposed:
        y = F.conv_transpose2d(xn, weight.permute(3, 0, 1, 2), None, stride, padding, output_padding)
    else:
        y = F.conv2d(xn, weight.permute(0, 3, 1, 2), None, stride, padding)
    y = y[:, :, :y.shape[2] - crop[0], :y.shape[3] - crop[1]].permute(0, 2, 3, 1)
    if scale is not None:
        y = y * scale
    if shift is not None:
        y = y + shift
    if act == "leaky_relu":
        y = F.leaky_relu(y, slope)
    if residual is not None:
        y = y + residual
    return y.contiguous()


def _linear_emul(x, w, b=None, act=None, alpha=1.0, residual=None, ln=None):
    y = F.linear(x, w, b) * alpha
    return y if residual is None else y + residual


class _FakeLib:
    def aps_store_magnitude(self, store, out, rows, eps, stream):
        s = np.ctypeslib.as_array((ctypes.c_float * (2 * rows)).from_address(store.value))
        o = np.ctypeslib.as_array((ctypes.c_float * rows).from_address(out.value))
        s = s.reshape(rows, 2)
        o[:] = np.sqrt(s[:, 0]**2 + s[:, 1]**2 + np.float32(eps))
        return 0


@pytest.mark.parametrize("tag,kw", DCCRN_VARIANTS)
def test_dccrn_decode_layout(monkeypatch, tag, kw):
    import aps_amd.sse.bss.dccrn as dc
    import aps_amd.sse.enh.dcunet as du
    from aps_amd import _native as nat
    from aps_amd.transform import EnhTransform
    from oracle import aps_oracle as ao
    from oracle import dccrn_oracle as do
    monkeypatch.setattr(du, "conv2d_nhwc", _conv_emul)
    monkeypatch.setattr(dc, "linear", _linear_emul)
    monkeypatch.setattr(dc, "lstm_supported", lambda *a, **k: False)  # torch's CPU nn.LSTM
    monkeypatch.setattr(nat, "require_device", lambda *a, **k: None)
    monkeypatch.setattr(nat, "load", lambda: _FakeLib())
    monkeypatch.setattr(nat, "stream_of", lambda t: None)
    kw = dict(kw)
    cplx = kw.pop("cplx", True)
    g = golden(tag)
    enh = EnhTransform(feats="spectrogram-log-cmvn", frame_len=64, frame_hop=32, window="hann")
    net = dc.DCCRN(cplx=cplx, K="3,3;3,3;3,3", S="2,1;2,1;2,1", P="1,1,1", O="0,0,0", C="16,32,32",
                   num_spks=2, rnn_hidden=64, rnn_layers=2, rnn_resize=320 if cplx else 160,
                   enh_transform=enh, training_mode="time", **kw)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    missing, unexpected = net.load_state_dict(sd, strict=False)  # the reference's own key names
    assert not unexpected, unexpected
    assert all(k.endswith("num_batches_tracked") for k in missing), missing
    net.eval()
    store = ao.stft(g["mix"], 64, 32, "hann").permute(0, 2, 1, 3).contiguous()  # N x T x F x 2
    with torch.no_grad():
        dec = net._decode(store)
        ref = do.dccrn_forward(sd, g["mix"], mode="freq", cplx=cplx, **DCCRN_SMALL,
                               **{**kw, "non_linear": "none"})  # = the decoder output itself
    for s in range(2):
        mine = torch.stack([dec[..., s], dec[..., 2 + s]], -1) if cplx else dec[..., s]
        mine = mine.transpose(1, 2)
        err = (mine - ref[s]).abs().max().item() / ref[s].abs().max().item()
        assert err < 1e-5, (tag, s, err)
