"""
DCCRN in train() mode on the GPU (aps/sse/bss/dccrn.py under autograd, what cmd/train_ss.py drives):

  * the convolution's adjoints (forward and transposed form, bias, causal crop) against torch's
    F.conv2d / F.conv_transpose2d under autograd,
  * one UNet block in train() mode against the oracle's batch-statistics block,
  * the whole network -- STFT, encoder, (complex) LSTM, decoder, masks, iSTFT -- against the
    REFERENCE module's own train()-mode step (fixtures <tag>_train_{time,freq}: outputs, the
    gradient of every parameter, the BatchNorm running statistics after the step), every variant:
    complex / real, shared / per-speaker decoders, "sum" / "cat", causal,
  * a few SGD steps on an SI-SNR loss.

Gradient tolerance 2e-5 of each parameter's gradient scale (tests/test_oracle_encoder.py holds
the CPU oracle to the same fixtures).
"""
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import assert_close
from tests.test_gpu_dccrn import small_net
from tests.test_oracle_encoder import DCCRN_TRAIN_CASES, assert_grad_close, dccrn_train_reference

pytestmark = pytest.mark.gpu
TOL = 2e-5  # measured: <= 2e-6 on every parameter of every variant (scripts/dccrn_grad_errors.py)


@pytest.mark.parametrize("transposed,stride,padding,outpad,crop,bias", [
    (False, (1, 2), (1, 1), (0, 0), (0, 0), True),
    (False, (1, 2), (2, 1), (0, 0), (2, 0), True),    # causal encoder block
    (False, (2, 2), (0, 1), (0, 0), (0, 0), False),
    (True, (1, 2), (1, 1), (0, 0), (0, 0), True),
    (True, (1, 2), (0, 1), (0, 1), (2, 0), True),     # causal decoder block, output padding
    (True, (2, 2), (1, 0), (1, 1), (0, 0), False),
])
def test_conv2d_nhwc_autograd(device, transposed, stride, padding, outpad, crop, bias):
    """x N x H x W x Ci, w Co x KH x KW x Ci: output, g_x, g_w, g_b vs torch's own layers"""
    from aps_amd.nn_ops import conv2d_nhwc
    torch.manual_seed(3)
    N, H, W, Ci, Co, KH, KW = 2, 9, 14, 6, 10, 3, 3
    x = torch.randn(N, H, W, Ci, requires_grad=True)
    w = (0.3 * torch.randn(Co, KH, KW, Ci)).requires_grad_(True)
    b = torch.randn(Co, requires_grad=True) if bias else None
    xt = x.permute(0, 3, 1, 2)
    if transposed:  # torch: weight [Ci, Co, KH, KW]
        ref = F.conv_transpose2d(xt, w.permute(3, 0, 1, 2), b, stride, padding, outpad)
    else:
        ref = F.conv2d(xt, w.permute(0, 3, 1, 2), b, stride, padding)
    ref = ref.permute(0, 2, 3, 1)
    ref = ref[:, :ref.shape[1] - crop[0], :ref.shape[2] - crop[1]]
    up = torch.randn(ref.shape)
    ref.backward(up)
    xd, wd = x.detach().to(device).requires_grad_(True), w.detach().to(device).requires_grad_(True)
    bd = None if b is None else b.detach().to(device).requires_grad_(True)
    out = conv2d_nhwc(xd, wd, None, bd, stride=stride, padding=padding, transposed=transposed,
                      output_padding=outpad, crop=crop)
    assert_close(out, ref, 1e-5, "conv forward")
    out.backward(up.to(device))
    assert_close(xd.grad, x.grad, 2e-5, "conv g_x")
    assert_close(wd.grad, w.grad, 2e-5, "conv g_w")
    if bias:
        assert_close(bd.grad, b.grad, 2e-5, "conv g_b")


@pytest.mark.parametrize("act", [None, "leaky_relu", "relu"])
def test_conv2d_nhwc_autograd_epilogue(device, act):
    """activation and residual behind the differentiable convolution"""
    from aps_amd.nn_ops import conv2d_nhwc
    torch.manual_seed(4)
    x = torch.randn(2, 7, 12, 8, requires_grad=True)
    w = (0.3 * torch.randn(12, 3, 3, 8)).requires_grad_(True)
    res = torch.randn(2, 7, 6, 12, requires_grad=True)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), None, (1, 2), (1, 1))
    ref = {None: lambda v: v, "leaky_relu": lambda v: F.leaky_relu(v, 0.01), "relu": F.relu}[act](ref)
    ref = ref.permute(0, 2, 3, 1) + res
    up = torch.randn(ref.shape)
    ref.backward(up)
    dev = [t.detach().to(device).requires_grad_(True) for t in (x, w, res)]
    out = conv2d_nhwc(dev[0], dev[1], stride=(1, 2), padding=(1, 1), act=act, residual=dev[2])
    out.backward(up.to(device))
    assert_close(out, ref, 1e-5, "forward")
    for d, c, name in zip(dev, (x, w, res), ("g_x", "g_w", "g_residual")):
        assert_close(d.grad, c.grad, 2e-5, name)
    with pytest.raises(NotImplementedError):  # a folded BatchNorm scale is an eval-mode construct
        conv2d_nhwc(dev[0], dev[1], torch.ones(12, device=device), stride=(1, 2), padding=(1, 1))


def test_block_train_mode_vs_oracle(device):
    """an encoder and a decoder block in train() mode (reference layout in and out): batch
    statistics, gradients of input and parameters, running statistics"""
    from oracle import dccrn_oracle as do
    net = small_net().train()
    torch.manual_seed(8)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    for k in sd:
        if sd[k].is_floating_point() and "running_" not in k:
            sd[k].requires_grad_(True)
    x = torch.randn(2, 16, 34, 9, requires_grad=True)   # into encoder layer 1
    h = torch.randn(2, 32, 10, 9, requires_grad=True)   # into decoder layer 0 (32 -> 32 channels)
    pe, pd = "encoder.layers.1.block.", "decoder.0.layers.0.block."
    ref_e = F.leaky_relu(do.cplx_bn(sd, pe + "1.", do.cplx_conv(sd, pe + "0.", x, (2, 1), (1, 1)),
                                    True), 0.01)
    ref_d = F.leaky_relu(do.cplx_bn(sd, pd + "1.", do.cplx_conv(sd, pd + "0.", h, (2, 1), (1, 1),
                                                               True, (0, 0)), True), 0.01)
    ue, ud = torch.randn(ref_e.shape), torch.randn(ref_d.shape)
    (ref_e * ue).sum().backward()
    (ref_d * ud).sum().backward()
    net = net.to(device)
    xd, hd = x.detach().to(device).requires_grad_(True), h.detach().to(device).requires_grad_(True)
    out_e = net.encoder.layers[1](xd)
    out_d = net.decoder[0].layers[0](hd)
    assert_close(out_e, ref_e, 2e-5, "encoder block (batch statistics)")
    assert_close(out_d, ref_d, 2e-5, "decoder block (batch statistics)")
    (out_e * ue.to(device)).sum().backward()
    (out_d * ud.to(device)).sum().backward()
    assert_close(xd.grad, x.grad, TOL, "encoder block g_x")
    assert_close(hd.grad, h.grad, TOL, "decoder block g_x")
    params = dict(net.named_parameters())
    for p in (pe, pd):
        for k in [k for k in sd if k.startswith(p)]:
            if "running_" in k:
                assert_close(net.state_dict()[k], sd[k], 1e-5, k)
            elif sd[k].grad is not None and not k.endswith("0.real.bias") and \
                    not k.endswith("0.imag.bias"):  # (conv bias in front of batch statistics: 0)
                assert_close(params[k].grad, sd[k].grad, TOL, k)
    for m in (net.encoder.layers[1].block[1], net.decoder[0].layers[0].block[1]):
        assert int(m.real_bn.num_batches_tracked) == 1 and int(m.imag_bn.num_batches_tracked) == 1


@pytest.mark.parametrize("tag,kw,mode", DCCRN_TRAIN_CASES)
def test_dccrn_train_step_vs_reference(device, tag, kw, mode):
    sd, mix, ref = dccrn_train_reference(tag, kw, mode)
    net = small_net(**dict(kw))
    net.training_mode = mode
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    net = net.train().to(device)
    out = net(mix.to(device))
    loss = sum((o * ref[f"probe{s}"].to(device)).sum() for s, o in enumerate(out))
    loss.backward()
    for s in range(2):
        assert_close(out[s], ref[f"out{s}"], TOL, f"{tag} train() output {s}")
    assert_close(loss, ref["loss"], TOL, "loss")
    params = dict(net.named_parameters())
    names = [k[5:] for k in ref if k.startswith("grad.")]
    assert len(names) > 20
    for k in names:
        assert params[k].grad is not None, k
        assert_grad_close(params[k].grad, ref, k, TOL, f"{tag} {mode}")
    state = net.state_dict()
    for k in [k[5:] for k in ref if k.startswith("stat.")]:
        assert_close(state[k], ref["stat." + k], 1e-5, f"{tag} running statistic {k}")
    # eval() after train(): the folded-weight launches pick the moved statistics up
    net.eval()
    with torch.no_grad():
        again = net(mix.to(device))
    assert all(torch.isfinite(a).all() for a in again)


def si_snr(est, ref, eps=1e-8):
    est, ref = est - est.mean(-1, keepdim=True), ref - ref.mean(-1, keepdim=True)
    proj = (est * ref).sum(-1, keepdim=True) * ref / (ref.square().sum(-1, keepdim=True) + eps)
    return 10 * torch.log10(proj.square().sum(-1) / ((est - proj).square().sum(-1) + eps) + eps)


def test_dccrn_trains_on_si_snr(device):
    """forward -> SI-SNR of both speakers -> backward -> Adam, a few steps on one batch: the loss
    goes down, every parameter has a finite gradient, the running statistics move"""
    torch.manual_seed(21)
    net = small_net().train().to(device)
    g = torch.Generator().manual_seed(22)
    src = 0.3 * torch.randn(2, 4, 2000, generator=g)  # speakers x batch x samples
    mix = (src[0] + src[1]).to(device)
    src = src.to(device)
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=2e-3)
    bn = [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    before = [m.running_var.clone() for m in bn]
    losses = []
    for _ in range(8):
        opt.zero_grad()
        out = net(mix)
        S = out[0].shape[-1]  # (the iSTFT returns whole frames: 1984 of the 2000 samples)
        loss = -(si_snr(out[0], src[0, :, :S]) + si_snr(out[1], src[1, :, :S])).mean()
        loss.backward()
        for name, p in net.named_parameters():
            if p.requires_grad and not name.startswith(("enh_transform", "forward_stft",
                                                        "inverse_stft")):
                assert p.grad is not None and torch.isfinite(p.grad).all(), name
        opt.step()
        losses.append(loss.item())
    print("[train] DCCRN -SI-SNR per step:", [f"{v:.3f}" for v in losses])
    assert losses[-1] < losses[0]
    assert all(not torch.equal(a, m.running_var) for a, m in zip(before, bn))
