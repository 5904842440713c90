"""
GPU: the training path.  Gradients of the HIP autograd functions
(aps_amd/grad_ops.py over grad.hip) against torch autograd on the CPU -- through the torch layer the
reference itself uses for single operators, and through the CPU oracle (the restatement of the
reference's arithmetic, oracle/) for RNNMaskMvdr.forward and the whole EnhASRBase data path.
Then the things the reference's trainer does with them (aps/trainer/ddp.py:107-200): one optimiser
step on a CTC loss in train() mode, and DistributedDataParallel gradient averaging over two ranks.
Tolerance 1e-4 of each gradient tensor's scale, like the forward activations.
"""
import os
import socket

import pytest
import torch
import torch.nn.functional as F

from tests.conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


def check(got, want, what, tol=TOL):
    err = rel_err(got, want)
    print(f"[grad] {what}: {err:.2e}")
    assert err <= tol, f"{what}: scaled max error {err:.3e} > {tol:.1e}"


@pytest.mark.parametrize("act", [None, "relu", "swish", "sigmoid", "tanh"])
def test_linear_backward(device, act):
    from aps_amd.nn_ops import linear
    torch.manual_seed(0)
    x = torch.randn(5, 37, 96)
    lin = torch.nn.Linear(96, 70)
    res = torch.randn(5, 37, 70)
    up = torch.randn(5, 37, 70)
    fn = {None: lambda v: v, "relu": torch.relu, "swish": F.silu, "sigmoid": torch.sigmoid,
          "tanh": torch.tanh}[act]
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    (fn(lin(xr)) * 0.5 + rr).backward(up)
    want = (xr.grad, lin.weight.grad.clone(), lin.bias.grad.clone(), rr.grad)
    lin.zero_grad()
    lin_d = lin.to(device)
    xd, rd = x.to(device).requires_grad_(True), res.to(device).requires_grad_(True)
    out = linear(xd, lin_d.weight, lin_d.bias, residual=rd, act=act, alpha=0.5)
    out.backward(up.to(device))
    for g, w, name in zip((xd.grad, lin_d.weight.grad, lin_d.bias.grad, rd.grad), want,
                          ("g_x", "g_W", "g_b", "g_residual")):
        check(g, w, f"linear[{act}] {name}")


def test_linear_with_folded_layernorm_backward(device):
    from aps_amd.nn_ops import linear
    torch.manual_seed(1)
    x = torch.randn(3, 20, 64)
    ln, lin = torch.nn.LayerNorm(64), torch.nn.Linear(64, 48)
    ln.weight.data.uniform_(0.5, 1.5)
    ln.bias.data.normal_()
    up = torch.randn(3, 20, 48)
    xr = x.clone().requires_grad_(True)
    F.silu(lin(ln(xr))).backward(up)
    want = [xr.grad] + [p.grad.clone() for p in (*ln.parameters(), *lin.parameters())]
    ln.zero_grad(), lin.zero_grad()
    ln, lin = ln.to(device), lin.to(device)
    xd = x.to(device).requires_grad_(True)
    linear(xd, lin.weight, lin.bias, act="swish", ln=ln).backward(up.to(device))
    got = [xd.grad] + [p.grad for p in (*ln.parameters(), *lin.parameters())]
    for g, w, name in zip(got, want, ("g_x", "g_gamma", "g_beta", "g_W", "g_b")):
        check(g, w, f"LN + linear {name}")


@pytest.mark.parametrize("training", [True, False])
def test_batchnorm_rows(device, training):
    from aps_amd.grad_ops import batchnorm_rows
    torch.manual_seed(2)
    x = torch.randn(4, 50, 24) * 2 + 1
    bn = torch.nn.BatchNorm1d(24)
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_()
    bn.running_mean.normal_()
    bn.running_var.uniform_(0.5, 2.0)
    bn.train(training)
    import copy
    bn_d = copy.deepcopy(bn).to(device)
    up = torch.randn(4, 50, 24)
    xr = x.clone().requires_grad_(True)
    y = bn(xr.transpose(1, 2)).transpose(1, 2)  # BatchNorm1d wants N x C x T
    y.backward(up)
    xd = x.to(device).requires_grad_(True)
    yd = batchnorm_rows(xd, bn_d)
    yd.backward(up.to(device))
    check(yd, y, "bn forward")
    check(xd.grad, xr.grad, "bn g_x")
    check(bn_d.weight.grad, bn.weight.grad, "bn g_gamma")
    check(bn_d.bias.grad, bn.bias.grad, "bn g_beta")
    check(bn_d.running_mean, bn.running_mean, "running mean")
    check(bn_d.running_var, bn.running_var, "running var")
    assert int(bn_d.num_batches_tracked) == int(bn.num_batches_tracked)


def test_conformer_convolution_module_backward(device):
    """pointwise -> GLU -> depthwise -> BatchNorm1d (batch statistics) -> Swish -> pointwise
    (impl.py:478-489) in train() mode against the same torch modules on the CPU"""
    import copy
    from aps_amd.asr.transformer.impl import ApsConformerEncoderLayer, RelMultiheadAttention
    torch.manual_seed(3)
    layer = ApsConformerEncoderLayer(64, RelMultiheadAttention(64, 2), feedforward_dim=96,
                                     kernel_size=5, dropout=0).train()
    conv = layer.convolution
    ref = copy.deepcopy(conv)
    x = torch.randn(3, 17, 64)
    up = torch.randn(3, 17, 64)
    xr = x.clone().requires_grad_(True)
    h = ref[3](ref[2](ref[1](ref[0](xr.transpose(1, 2)))))  # Conv1d modules: N x D x T
    ref[5](F.silu(h)).transpose(1, 2).backward(up)
    layer = layer.to(device)
    xd = x.to(device).requires_grad_(True)
    layer.conv_run(xd, None).backward(up.to(device))
    check(xd.grad, xr.grad, "conv module g_x")
    for (name, p), q in zip(conv.named_parameters(), ref.parameters()):
        if name == "2.bias":  # a bias in front of batch statistics: the exact gradient is zero
            assert p.grad.abs().max().item() < 1e-4 and q.grad.abs().max().item() < 1e-4
            continue
        check(p.grad, q.grad, f"conv module {name}")


@pytest.mark.parametrize("use_lens", [False, True])
def test_rel_attention_backward(device, use_lens):
    from aps_amd.nn_ops import attention_core
    from tests.test_grad_host import rel_attention_reference
    torch.manual_seed(4)
    N, T, H, dh = 3, 21, 2, 64
    qkv = torch.randn(N, T, 3 * H * dh)
    rel = torch.randn(2 * T - 1, dh)
    lens = torch.tensor([21, 15, 9]) if use_lens else None
    up = torch.randn(N, T, H * dh)
    qr, rr = qkv.clone().requires_grad_(True), rel.clone().requires_grad_(True)
    rel_attention_reference(qr, lens, rr, T - 1, H).backward(up)
    qd, rd = qkv.to(device).requires_grad_(True), rel.to(device).requires_grad_(True)
    out = attention_core(qd, H, None if lens is None else lens.to(device), rel=rd)
    out.backward(up.to(device))
    check(qd.grad, qr.grad, "attention g_qkv")
    check(rd.grad, rr.grad, "attention g_rel")


def test_conv2d_subsampling_block_backward(device):
    """Conv2d -> BatchNorm2d -> ReLU blocks + output projection of Conv2dEncoder
    (component.py:251-307, encoder.py:367-441) in train() mode vs the torch modules"""
    import copy
    from aps_amd.asr.base.encoder import Conv2dEncoder
    torch.manual_seed(5)
    enc = Conv2dEncoder(40, 48, channel=32, num_layers=2).train()
    ref = copy.deepcopy(enc)
    x = torch.randn(2, 37, 40)
    up = torch.randn(2, 10, 48)
    h = x[:, None]
    for blk in ref.enc_layers:  # the torch modules of the same block
        h = F.relu(blk.norm.norm(blk.conv(h)))
    ref_out = F.linear(h.transpose(1, 2).contiguous().view(2, h.shape[2], -1), ref.outp.weight,
                       ref.outp.bias)
    ref_out.backward(up)
    enc = enc.to(device)
    out, _ = enc(x.to(device), None)
    check(out, ref_out, "conv2d encoder forward (batch statistics)")
    out.backward(up.to(device))
    for (name, p), q in zip(enc.named_parameters(), ref.parameters()):
        if name.endswith("conv.bias"):  # in front of batch statistics: the exact gradient is zero
            assert p.grad.abs().max().item() < 1e-4 and q.grad.abs().max().item() < 1e-4
            continue
        check(p.grad, q.grad, f"conv2d encoder {name}", tol=2e-4)


@pytest.mark.parametrize("bidir", [False, True])
@pytest.mark.parametrize("use_lens", [False, True])
def test_lstm_stack_backward(device, use_lens, bidir):
    import copy
    from aps_amd.nn_ops import lstm_forward
    torch.manual_seed(6)
    N, T, D, H = 5, 23, 48, 64
    rnn = torch.nn.LSTM(D, H, num_layers=2, batch_first=True, bidirectional=bidir)
    x = torch.randn(N, T, D)
    lens = torch.tensor([23, 23, 17, 9, 4]) if use_lens else None
    up = torch.randn(N, T, H * (2 if bidir else 1))
    xr = x.clone().requires_grad_(True)
    if use_lens:
        packed = torch.nn.utils.rnn.pack_padded_sequence(xr, lens.tolist(), batch_first=True,
                                                         enforce_sorted=False)
        y, _ = torch.nn.utils.rnn.pad_packed_sequence(rnn(packed)[0], batch_first=True,
                                                      total_length=T)
    else:
        y, _ = rnn(xr)
    y.backward(up)
    rnn_d = copy.deepcopy(rnn).to(device)
    rnn_d.zero_grad()
    xd = x.to(device).requires_grad_(True)
    yd = lstm_forward(rnn_d, xd, None if lens is None else lens.to(device))
    check(yd, y, "lstm forward")
    yd.backward(up.to(device))
    check(xd.grad, xr.grad, "lstm g_x")
    for (name, p), q in zip(rnn_d.named_parameters(), rnn.parameters()):
        check(p.grad, q.grad, f"lstm {name}")


@pytest.mark.parametrize("N,T,bidir,use_lens", [(20, 31, False, True), (32, 40, False, False),
                                                (5, 17, True, True), (130, 9, False, True)])
def test_lstm_backward_hidden_512(device, N, T, bidir, use_lens):
    """H = 512 (the mask estimator's width): the reverse sweep runs as ONE persistent launch
    (lstm_backward_team_kernel: N <= 128; 130 utterances go through the launch-per-step form) --
    gradients of the input and every parameter against torch's own LSTM under autograd, ragged
    lengths, partial 16-utterance teams"""
    import copy
    from aps_amd.nn_ops import lstm_forward
    torch.manual_seed(N + T)
    D, H = 40, 512
    rnn = torch.nn.LSTM(D, H, num_layers=1, batch_first=True, bidirectional=bidir)
    x = torch.randn(N, T, D)
    lens = None
    if use_lens:
        lens = torch.randint(1, T + 1, (N,))
        lens[0] = T
    up = torch.randn(N, T, H * (2 if bidir else 1))
    xr = x.clone().requires_grad_(True)
    if use_lens:
        packed = torch.nn.utils.rnn.pack_padded_sequence(xr, lens.tolist(), batch_first=True,
                                                         enforce_sorted=False)
        y, _ = torch.nn.utils.rnn.pad_packed_sequence(rnn(packed)[0], batch_first=True,
                                                      total_length=T)
    else:
        y, _ = rnn(xr)
    y.backward(up)
    rnn_d = copy.deepcopy(rnn).to(device)
    rnn_d.zero_grad()
    xd = x.to(device).requires_grad_(True)
    yd = lstm_forward(rnn_d, xd, None if lens is None else lens.to(device))
    check(yd, y, "lstm forward")
    yd.backward(up.to(device))
    check(xd.grad, xr.grad, "lstm g_x")
    for (name, p), q in zip(rnn_d.named_parameters(), rnn.parameters()):
        check(p.grad, q.grad, f"lstm {name}")


def small_joint(seed=41):
    from tests.test_gpu_joint import SMALL_ENC, build_joint
    torch.manual_seed(seed)
    net = build_joint(40, 48, 64, 32, 50, SMALL_ENC)
    for m in net.modules():  # non-trivial running statistics
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.7, 1.4)
    return net


def joint_inputs(seed=42, N=3, S=9000):
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(N, S + 16, generator=g)
    wav = torch.stack([src[:, d:d + S] for d in (0, 2, 5, 9)], 1)
    wav = 0.1 * (wav + 0.5 * torch.randn(N, 4, S, generator=g))
    lens = torch.tensor([S, S - 1500, S - 4000][:N])
    return wav, lens, g


def oracle_grads(net, wav, lens, loss_of):
    """autograd through the CPU oracle: every entry of the state dict is a leaf"""
    from oracle import joint_oracle as jo
    trainable = {n for n, p in net.named_parameters() if p.requires_grad}
    sd = {k: (v.detach().clone().requires_grad_(True) if k in trainable else v.detach().clone())
          for k, v in net.state_dict().items()}
    ref = jo.joint_forward(sd, wav, lens, num_mels=40, rnn_layers=2, enc_layers=2, nhead=2,
                           lradius=4, rradius=4, kernel_size=5)
    loss_of(ref).backward()
    return ref, {k: v.grad for k, v in sd.items() if k in trainable and v.grad is not None}


def test_rnn_mask_mvdr_backward_vs_oracle(device):
    """RNNMaskMvdr.forward (mvdr.py:177-234) end to end: mask estimator (GEMM + LSTM stack + GEMM)
    -> covariances -> channel attention -> per-bin solve -> beamformer, gradients of every
    parameter of the front end against autograd through the oracle"""
    from aps_amd.cplx import ComplexTensor
    net = small_joint().eval()
    wav, lens, g = joint_inputs()
    T = (9000 - 512) // 256 + 1
    ur, ui = torch.randn(3, T, 257, generator=g), torch.randn(3, T, 257, generator=g)

    def loss_of(ref):
        yr, yi = ref["enh"]
        return (yr * ur).sum() + (yi * ui).sum()

    ref, want = oracle_grads(net, wav, lens, loss_of)
    net = net.to(device)
    with torch.no_grad():
        packed, n = net.enh_transform.encode(wav.to(device), lens.to(device))
        feats = net.enh_transform(packed)
    y = net.enh_net(feats, ComplexTensor(packed[..., 0], packed[..., 1]), inp_len=n)
    check(y.real, ref["enh"][0], "enhanced real")
    check(y.imag, ref["enh"][1], "enhanced imag")
    ((y.real * ur.to(device)).sum() + (y.imag * ui.to(device)).sum()).backward()
    seen = 0
    for name, p in net.enh_net.named_parameters():
        seen += 1
        if name.endswith("gvec.bias"):  # softmax over the channels is shift invariant: exactly 0
            assert p.grad.abs().max().item() < 1e-6 * net.enh_net.mvdr_net.ref.gvec.weight.grad.abs().max().item() + 1e-12
            continue
        check(p.grad, want["enh_net." + name], f"enh_net.{name}")
    assert seen >= 12  # mask net (proj, 2 LSTM layers, outp) + ChannelAttention


def test_joint_backward_vs_oracle(device):
    """EnhASRBase.forward (enh_att.py:83-95) with gradients enabled (eval-mode statistics): the
    gradient of EVERY trainable parameter -- mask estimator, ChannelAttention, conv2d subsampling,
    relative position table, conformer layers, CTC head -- against autograd through the oracle"""
    net = small_joint().eval()
    wav, lens, g = joint_inputs(seed=43)

    def shapes(ref):
        return ref["enc_out"].shape, ref["enc_ctc"].shape

    from oracle import joint_oracle as jo
    with torch.no_grad():
        probe = jo.joint_forward({k: v.detach() for k, v in net.state_dict().items()}, wav, lens,
                                 num_mels=40, rnn_layers=2, enc_layers=2, nhead=2, lradius=4,
                                 rradius=4, kernel_size=5)
    u1 = torch.randn(probe["enc_out"].shape, generator=g)
    u2 = torch.randn(probe["enc_ctc"].shape, generator=g)

    def loss_of(ref):
        return (ref["enc_out"] * u1).sum() + (ref["enc_ctc"] * u2).sum()

    ref, want = oracle_grads(net, wav, lens, loss_of)
    net = net.to(device)
    enc_out, enc_ctc, enc_len = net(wav.to(device), lens.to(device))
    check(enc_out, ref["enc_out"], "encoder output")
    ((enc_out * u1.to(device)).sum() + (enc_ctc * u2.to(device)).sum()).backward()
    missing = [n for n, p in net.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, f"no gradient reached {missing}"
    worst = 0.0
    for name, p in net.named_parameters():
        if not p.requires_grad or name not in want or name.endswith("gvec.bias"):
            continue  # (gvec.bias: shift of a softmax input, exact gradient zero)
        err = rel_err(p.grad, want[name])
        worst = max(worst, err)
        assert err <= 2e-4, f"{name}: gradient error {err:.3e}"
    print(f"[grad] joint: worst parameter-gradient error {worst:.2e}")


def test_joint_trains_one_step_with_ctc(device):
    """what aps/trainer/ddp.py:124-200 does per batch, in train() mode (BatchNorm on batch
    statistics): forward -> torch CTC loss on the CTC head -> backward -> SGD step; the loss on the
    same batch goes down and the running statistics moved"""
    net = small_joint(seed=44).train().to(device)
    wav, lens, g = joint_inputs(seed=45)
    wav, lens = wav.to(device), lens.to(device)
    tgt = torch.randint(1, 50, (3, 4), generator=g).to(device)
    tgt_len = torch.tensor([4, 3, 2], device=device)
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    bn = [m for m in net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    before = [m.running_mean.clone() for m in bn]
    losses = []
    for _ in range(4):
        opt.zero_grad()
        _, enc_ctc, enc_len = net(wav, lens)
        logp = F.log_softmax(enc_ctc, -1).transpose(0, 1)  # T x N x V
        loss = F.ctc_loss(logp, tgt, enc_len, tgt_len, blank=0, reduction="mean",
                          zero_infinity=True)
        loss.backward()
        for name, p in net.named_parameters():
            if p.requires_grad:
                assert p.grad is not None and torch.isfinite(p.grad).all(), name
        opt.step()
        losses.append(loss.item())
    print("[train] CTC loss per step:", [f"{v:.4f}" for v in losses])
    assert losses[-1] < losses[0]
    assert any(not torch.equal(a, m.running_mean) for a, m in zip(before, bn))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ddp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from aps_amd import distributed as D
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    # one GPU on this box: both ranks share it, so the process group is gloo (RCCL refuses two
    # ranks on one device); the DDP reducer, its bucketing and the all-reduce of gradients are the
    # same code path the nccl backend drives on a multi-GPU node
    D.init("torch", "gloo")
    net = small_joint(seed=46).eval().to(device)
    # the settings the training path uses (aps_amd/distributed.py:ddp_kwargs: gradients as views of the
    # reducer's buckets, static graph, 32 MB buckets); two steps, so that the static-graph reducer's
    # second (re-ordered) iteration is the one whose gradients are compared
    ddp = DDP(net, device_ids=[0], **D.ddp_kwargs())
    wav, lens, g = joint_inputs(seed=100 + rank)  # a different shard per rank
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        enc_out, enc_ctc, _ = ddp(wav.to(device), lens.to(device))
        (enc_out.square().mean() + enc_ctc.square().mean()).backward()
    grads = {n: p.grad.detach().cpu().numpy() for n, p in net.named_parameters()
             if p.grad is not None}  # numpy: pickled by value (the worker exits right after)
    dist.barrier()
    out.put((rank, grads))
    dist.destroy_process_group()


def test_ddp_gradient_all_reduce_two_ranks(device):
    """DistributedDataParallel over the HIP modules (aps/trainer/ddp.py:107-119): after backward
    both ranks hold the same gradients = the mean of the two shards' single-process gradients"""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process gradients of each shard
    single = []
    for rank in range(world):
        net = small_joint(seed=46).eval().to(device)
        wav, lens, g = joint_inputs(seed=100 + rank)
        enc_out, enc_ctc, _ = net(wav.to(device), lens.to(device))
        (enc_out.square().mean() + enc_ctc.square().mean()).backward()
        single.append({n: p.grad.detach().cpu() for n, p in net.named_parameters()
                       if p.grad is not None})
    assert res[0].keys() == res[1].keys() == single[0].keys()
    worst = 0.0
    for name in res[0]:
        a, b = torch.from_numpy(res[0][name]), torch.from_numpy(res[1][name])
        assert torch.equal(a, b), f"{name}: ranks disagree"
        mean = 0.5 * (single[0][name] + single[1][name])
        scale = max(single[0][name].abs().max().item(), single[1][name].abs().max().item(), 1e-30)
        err = (a - mean).abs().max().item() / scale
        worst = max(worst, err)
        assert err <= 1e-5, f"{name}: {err:.2e}"
    print(f"[ddp] {len(res[0])} gradients averaged over 2 ranks, worst deviation {worst:.1e}")


# ---------------------------------------------------------------- dropout (train() mode)
def _keep(seed, numel, p):
    import numpy as np
    from tests.test_grad_host import keep_scale_reference
    with np.errstate(over="ignore"):
        return torch.from_numpy(keep_scale_reference(seed, np.arange(numel), p))


def test_dropout_is_the_counter_mask_in_both_directions(device):
    """nn.Dropout in train(): element i survives iff hash(seed, i) says so (grad_core.h:keep_scale),
    scaled by 1 / (1 - p); the backward applies the same mask without having stored it"""
    from aps_amd.grad_ops import DropoutFn, dropout
    torch.manual_seed(1)
    x = torch.randn(7, 33, 129)
    seed, p = 31415926535897, 0.2
    ks = _keep(seed, x.numel(), p).view_as(x)
    xd = x.to(device).requires_grad_(True)
    out = DropoutFn.apply(xd, p, seed)
    assert torch.equal(out.cpu(), x * ks)
    g = torch.randn_like(x)
    out.backward(g.to(device))
    assert torch.equal(xd.grad.cpu(), g * ks)
    drop = torch.nn.Dropout(0.2)
    torch.manual_seed(9)
    a = dropout(xd.detach(), drop)
    torch.manual_seed(9)
    b = dropout(xd.detach(), drop)
    c = dropout(xd.detach(), drop)
    assert torch.equal(a, b) and not torch.equal(a, c)  # torch.manual_seed reproduces a run
    assert abs((a != 0).float().mean().item() - 0.8) < 1e-2
    assert dropout(xd, drop.eval()) is xd  # eval(): identity, no launch


@pytest.mark.parametrize("dh,use_rel,use_lens", [(64, True, True), (32, False, False),
                                                 (32, True, False)])
def test_attention_weight_dropout_backward(device, dh, use_rel, use_lens):
    """dropout on the attention weights (reference impl.py:104 / torch's MHA): forward and backward
    with the mask recomputed from (seed, n, h, i, j), vs torch autograd with that mask applied"""
    from aps_amd.grad_ops import AttentionFn
    torch.manual_seed(23)
    N, T, H = 3, 19, 2
    seed, p = 271828182845, 0.25
    qkv = torch.randn(N, T, 3 * H * dh)
    rel = torch.randn(2 * T - 1, dh) if use_rel else None
    lens = torch.tensor([19, 12, 7]) if use_lens else None
    mask = _keep(seed, N * H * T * T, p).view(N, H, T, T)
    qr = qkv.clone().requires_grad_(True)
    rr = None if rel is None else rel.clone().requires_grad_(True)
    q, k, v = qr.view(N, T, 3, H, dh).unbind(2)
    s = torch.einsum("nihd,njhd->nhij", q, k)
    if rr is not None:
        idx = torch.arange(T)[None, :] - torch.arange(T)[:, None] + T - 1
        s = s + torch.einsum("nihd,ijd->nhij", q, rr[idx])
    s = s / dh**0.5
    if lens is not None:
        s = s.masked_fill((torch.arange(T)[None, :] >= lens[:, None])[:, None, None, :],
                          float("-inf"))
    want = torch.einsum("nhij,njhd->nihd", torch.softmax(s, -1) * mask, v).reshape(N, T, -1)
    up = torch.randn_like(want)
    want.backward(up)
    qd = qkv.to(device).requires_grad_(True)
    rd = None if rel is None else rel.to(device).requires_grad_(True)
    out = AttentionFn.apply(qd, rd, None if lens is None else lens.to(device), H, None, p, seed)
    check(out, want, "attention with weight dropout")
    out.backward(up.to(device))
    check(qd.grad, qr.grad, "dropout attention g_qkv")
    if rel is not None:
        check(rd.grad, rr.grad, "dropout attention g_rel")


def test_conformer_feedforward_with_dropout_backward(device):
    """the macaron feed-forward in train() mode with dropout 0.3: Linear - Swish - Dropout - Linear
    - Dropout, * 0.5 + src (reference impl.py:515-520).  The two seeds are the next two draws of
    torch's CPU generator, so the test rebuilds both masks and differentiates the same function
    with torch"""
    from aps_amd.asr.transformer.impl import ApsConformerEncoderLayer, RelMultiheadAttention
    from aps_amd.grad_ops import draw_seed
    torch.manual_seed(5)
    D, F_, N, T, p = 64, 96, 3, 11, 0.3
    layer = ApsConformerEncoderLayer(D, RelMultiheadAttention(D, 2), feedforward_dim=F_,
                                     dropout=p, kernel_size=5).train()
    x = torch.randn(N, T, D)
    up = torch.randn(N, T, D)
    ffn, ln = layer.feedforward1, layer.norm_ffn1
    torch.manual_seed(77)
    s1, s2 = draw_seed(), draw_seed()
    xr = x.clone().requires_grad_(True)
    h = F.silu(ffn[0](ln(xr))) * _keep(s1, N * T * F_, p).view(N, T, F_)
    want = ffn[3](h) * _keep(s2, N * T * D, p).view(N, T, D) * 0.5 + xr
    want.backward(up)
    ref = {n: q.grad.clone() for n, q in layer.named_parameters() if q.grad is not None}
    layer.zero_grad()
    layer = layer.to(device)
    xd = x.to(device).requires_grad_(True)
    torch.manual_seed(77)
    out = layer._ffn(layer.feedforward1, xd, xd, ln=layer.norm_ffn1)
    check(out, want, "ffn with dropout")
    out.backward(up.to(device))
    check(xd.grad, xr.grad, "ffn dropout g_x")
    for n, q in layer.named_parameters():
        if n in ref:
            check(q.grad, ref[n], f"ffn dropout {n}")


def test_lstm_dropout_between_layers(device):
    """nn.LSTM(dropout=p).train(): the output of every layer but the last is dropped.  Rebuilt
    with torch from single-layer LSTMs and the same counter mask"""
    from aps_amd.grad_ops import draw_seed
    from aps_amd.nn_ops import lstm_forward
    torch.manual_seed(6)
    N, T, D, H, p = 3, 13, 24, 64, 0.2
    rnn = torch.nn.LSTM(D, H, num_layers=2, batch_first=True, dropout=p).train()
    x = torch.randn(N, T, D)
    up = torch.randn(N, T, H)
    l0 = torch.nn.LSTM(D, H, batch_first=True)
    l1 = torch.nn.LSTM(H, H, batch_first=True)
    for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
        getattr(l0, name + "_l0").data.copy_(getattr(rnn, name + "_l0"))
        getattr(l1, name + "_l0").data.copy_(getattr(rnn, name + "_l1"))
    torch.manual_seed(78)
    seed = draw_seed()
    xr = x.clone().requires_grad_(True)
    want = l1(l0(xr)[0] * _keep(seed, N * T * H, p).view(N, T, H))[0]
    want.backward(up)
    rnn_d = rnn.to(device)
    xd = x.to(device).requires_grad_(True)
    torch.manual_seed(78)
    out = lstm_forward(rnn_d, xd)
    check(out, want, "lstm with layer dropout")
    out.backward(up.to(device))
    check(xd.grad, xr.grad, "lstm layer dropout g_x")
    for name in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
        check(getattr(rnn_d, name + "_l0").grad, getattr(l0, name + "_l0").grad, f"lstm drop {name}_l0")
        check(getattr(rnn_d, name + "_l1").grad, getattr(l1, name + "_l0").grad, f"lstm drop {name}_l1")
    rnn_d.eval()
    with torch.no_grad():
        check(lstm_forward(rnn_d, xd.detach()), rnn.cpu()(x)[0], "eval(): no dropout")


def dropout_joint(seed=51):
    """the small joint model with the reference's default kind of dropouts switched on: encoder
    positional / attention-weight / feed-forward / layer dropouts and the mask RNN's"""
    from tests.test_gpu_joint import SMALL_ENC, build_joint
    torch.manual_seed(seed)
    enc = dict(SMALL_ENC, pose_kwargs=dict(SMALL_ENC["pose_kwargs"], dropout=0.1),
               arch_kwargs=dict(SMALL_ENC["arch_kwargs"], att_dropout=0.2, ffn_dropout=0.2))
    return build_joint(40, 48, 64, 32, 50, enc, enh_dropout=0.2)


def test_joint_trains_with_dropout(device):
    """train() mode with every dropout of the encoder active: runs are reproducible under
    torch.manual_seed, differ between seeds, eval() is the deterministic forward, and SGD on the CTC
    loss still goes down"""
    net = dropout_joint().train().to(device)
    wav, lens, g = joint_inputs(seed=52)
    wav, lens = wav.to(device), lens.to(device)
    torch.manual_seed(1)
    a = net(wav, lens)[0]
    torch.manual_seed(1)
    b = net(wav, lens)[0]
    c = net(wav, lens)[0]
    assert torch.equal(a, b) and not torch.equal(a, c)
    net.eval()
    with torch.no_grad():
        d, e = net(wav, lens)[0], net(wav, lens)[0]
    assert torch.equal(d, e) and not torch.equal(d, a.detach())
    net.train()
    tgt = torch.randint(1, 50, (3, 4), generator=g).to(device)
    tgt_len = torch.tensor([4, 3, 2], device=device)
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        _, enc_ctc, enc_len = net(wav, lens)
        logp = F.log_softmax(enc_ctc, -1).transpose(0, 1)
        loss = F.ctc_loss(logp, tgt, enc_len, tgt_len, blank=0, reduction="mean", zero_infinity=True)
        loss.backward()
        for name, q in net.named_parameters():
            if q.requires_grad:
                assert q.grad is not None and torch.isfinite(q.grad).all(), name
        opt.step()
        losses.append(loss.item())
    print("[train] CTC loss per step with dropout:", [f"{v:.4f}" for v in losses])
    assert min(losses[-2:]) < losses[0]


@pytest.mark.parametrize("M,I,J", [(160000, 128, 12), (40320, 128, 100), (20000, 24, 12),
                                   (16384, 200, 300), (9000, 16, 12), (2016, 512, 512), (2016, 2048, 512),
                                   (7968, 2048, 257), (33, 70, 65), (1, 3, 5), (2017, 513, 130)])
def test_weight_gradient_product(device, M, I, J):
    """x^T y over the row axis with the column sums of x riding along (aps_gemm_tn: the weight and bias
    gradient of a projection in one call; the conv2d layers' 128 x 12 over 160 000 output pixels; ragged
    edges in every dimension; a row pitch on either operand) against float64"""
    from aps_amd.grad_ops import xty
    g = torch.Generator().manual_seed(M + I)
    x, y = torch.randn(M, I + 3, generator=g), torch.randn(M, J + 1, generator=g)
    xd, yd = x.to(device)[:, :I], y.to(device)[:, 1:]          # pitched views: lda = I + 3, ldb = J + 1
    ref = x[:, :I].double().T @ y[:, 1:].double()
    out, cs = xty(xd, yd, colsum=True)
    assert out.shape == ref.shape
    check(out, ref, f"x^T y {M} x {I} x {J}", tol=2e-5)
    check(cs, x[:, :I].double().sum(0), f"column sums {M} x {I}", tol=2e-5)
    again, cs2 = xty(xd, yd, colsum=True)
    assert torch.equal(out, again) and torch.equal(cs, cs2)   # slab sums in a fixed order: bit-reproducible
    check(xty(xd, yd), ref, f"x^T y {M} x {I} x {J} (no column sums)", tol=2e-5)


# ------------------------------------------------------------------------------------------------
# Transformer-XL attention, context windows, the causal conformer convolution
# ------------------------------------------------------------------------------------------------
def test_causal_convolution_module_backward(device):
    """casual_conv1d (impl.py:446, 491-505) in train() mode: K - 1 zero frames in front of the first
    pointwise convolution, so the depthwise convolution sees glu(bias) there and the bias collects a
    gradient through them"""
    import copy
    from aps_amd.asr.transformer.impl import ApsConformerEncoderLayer, RelMultiheadAttention
    torch.manual_seed(13)
    layer = ApsConformerEncoderLayer(64, RelMultiheadAttention(64, 2), feedforward_dim=96,
                                     kernel_size=5, dropout=0, casual_conv1d=True).train()
    conv = layer.convolution
    with torch.no_grad():
        conv[0].bias.normal_(0, 0.5)
    ref = copy.deepcopy(conv)
    x = torch.randn(3, 17, 64)
    up = torch.randn(3, 17, 64)
    xr = x.clone().requires_grad_(True)
    h = ref[3](ref[2](ref[1](ref[0](F.pad(xr.transpose(1, 2), (4, 0))))))  # Conv1d modules: N x D x T
    ref[5](F.silu(h)).transpose(1, 2).backward(up)
    layer = layer.to(device)
    xd = x.to(device).requires_grad_(True)
    out = layer.conv_run(xd, None)
    assert out.shape == (3, 17, 64)
    out.backward(up.to(device))
    check(xd.grad, xr.grad, "causal conv module g_x")
    for (name, p), q in zip(conv.named_parameters(), ref.parameters()):
        if name == "2.bias":  # a bias in front of batch statistics: the exact gradient is zero
            assert p.grad.abs().max().item() < 1e-4 and q.grad.abs().max().item() < 1e-4
            continue
        check(p.grad, q.grad, f"causal conv module {name}")


@pytest.mark.parametrize("case,T", [("window", 21), ("rel_window", 40), ("xl_shared", 21),
                                    ("xl_per_head_value_query", 33), ("xl_window", 100),
                                    ("xl_window", 150)])
def test_attention_xl_window_backward(device, case, T):
    """attention_core under autograd with context windows, per-head tables, the XL biases and the
    query read from the value projection (XlMultiheadAttention.dot_att, impl.py:322-374) against
    autograd through the explicit float64 form"""
    from aps_amd.nn_ops import attention_core
    from tests.test_grad_host import xl_window_reference
    torch.manual_seed(T + len(case))
    N, H, dh = 3, 2, 32
    cfg = {"window": dict(window=(2, 1, 0)), "rel_window": dict(window=(4, 2, 1), table=True),
           "xl_shared": dict(xl=True), "xl_per_head_value_query": dict(xl=True, per_head=True, qslot=2),
           "xl_window": dict(xl=True, per_head=True, qslot=2, window=(8, 2, 1))}[case]
    window = cfg.get("window", (1, -1, -1))
    qslot = cfg.get("qslot", 0)
    qkv = torch.randn(N, T, 3 * H * dh)
    lens = torch.tensor([T, T - 6, T - 11])
    R = 2 * T - 1
    table = u = v = None
    if cfg.get("xl") or cfg.get("table"):
        table = torch.randn(*((H, R, dh) if cfg.get("per_head") else (R, dh)))
    if cfg.get("xl"):
        u, v = torch.randn(H, dh), torch.randn(H, dh)

    def leaf(t, dev=None, dtype=None):
        return None if t is None else t.to(device=dev, dtype=dtype).clone().requires_grad_(True)

    r = [leaf(t, dtype=torch.float64) for t in (qkv, table, u, v)]
    # (a query whose window holds no valid key -- padded frames of the "window" case -- gives a zero
    # context row and no gradient on both sides)
    ctx = xl_window_reference(r[0], lens, r[1], r[2], r[3], T - 1, H, qslot, window)
    up = torch.randn(N, T, H * dh)
    (ctx * up.double()).sum().backward()
    d = [leaf(t, dev=device) for t in (qkv, table, u, v)]
    out = attention_core(d[0], H, lens.to(device), rel=d[1], rel_u=d[2], rel_v=d[3],
                         query_from_value=qslot == 2, chunk_size=window[0], lctx=window[1],
                         rctx=window[2])
    check(out, ctx.detach().float(), f"{case} context", 1e-5)
    out.backward(up.to(device))
    for got, want, what in zip(d, r, ("g_qkv", "g_table", "g_u", "g_v")):
        if want is not None:
            check(got.grad, want.grad.float(), f"{case} T={T} {what}")


@pytest.mark.parametrize("case,T,dh", [("xl_window", 40, 32), ("window", 21, 16), ("xl_shared", 70, 64)])
def test_attention_xl_window_dropout_backward(device, case, T, dh):
    """the same general attention in train() mode: dropout on the weights (impl.py:104), the mask
    recomputed from (seed, n, h, i, j) in the backward; any head size"""
    from aps_amd.grad_ops import AttentionXlFn
    from tests.test_grad_host import xl_window_reference
    torch.manual_seed(T + dh)
    N, H = 2, 2
    seed, p = 1618033988749, 0.2
    cfg = {"window": dict(window=(2, 1, 0)), "xl_shared": dict(xl=True),
           "xl_window": dict(xl=True, per_head=True, qslot=2, window=(8, 2, 1))}[case]
    window = cfg.get("window", (1, -1, -1))
    qslot = cfg.get("qslot", 0)
    qkv = torch.randn(N, T, 3 * H * dh)
    lens = torch.tensor([T, T - 6])
    table = u = v = None
    if cfg.get("xl"):
        table = torch.randn(*((H, 2 * T - 1, dh) if cfg.get("per_head") else (2 * T - 1, dh)))
        u, v = torch.randn(H, dh), torch.randn(H, dh)
    keep = _keep(seed, N * H * T * T, p).view(N, H, T, T).double()

    def leaf(t, dev=None, dtype=None):
        return None if t is None else t.to(device=dev, dtype=dtype).clone().requires_grad_(True)

    r = [leaf(t, dtype=torch.float64) for t in (qkv, table, u, v)]
    ctx = xl_window_reference(r[0], lens, r[1], r[2], r[3], T - 1, H, qslot, window, keep)
    up = torch.randn(N, T, H * dh)
    (ctx * up.double()).sum().backward()
    d = [leaf(t, dev=device) for t in (qkv, table, u, v)]
    out = AttentionXlFn.apply(d[0], d[1], d[2], d[3], lens.to(device), H, None, qslot == 2, *window, p,
                              seed)
    check(out, ctx.detach().float(), f"{case} context with weight dropout", 1e-5)
    out.backward(up.to(device))
    for got, want, what in zip(d, r, ("g_qkv", "g_table", "g_u", "g_v")):
        if want is not None:
            check(got.grad, want.grad.float(), f"{case} dropout {what}")


def test_xl_encoder_layer_trains_with_dropout(device):
    """a Transformer-XL encoder in train() mode with the reference's default dropouts: one step runs,
    every parameter receives a finite gradient, and the same seed gives the same step"""
    from aps_amd.asr.transformer import TransformerEncoder
    enc = TransformerEncoder("xfmr", 24, num_layers=2, proj="linear", proj_kwargs={}, pose="xl",
                             pose_kwargs={"dropout": 0.1}, chunk_size=2, lctx=2, rctx=1,
                             arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 96,
                                          "att_dropout": 0.1, "ffn_dropout": 0.1}).to(device).train()
    x = torch.randn(3, 40, 24, device=device)
    grads = []
    for _ in range(2):
        torch.manual_seed(5)
        enc.zero_grad()
        out, _ = enc(x, None)
        out.square().mean().backward()
        grads.append({n: p.grad.clone() for n, p in enc.named_parameters() if p.requires_grad})
    for n, g in grads[0].items():
        assert torch.isfinite(g).all(), n
        assert torch.equal(g, grads[1][n]), n
    assert any(g.abs().max() > 0 for g in grads[0].values())


@pytest.mark.parametrize("arch,pose,kw,top", [
    ("xfmr", "xl", {}, dict(proj="linear", proj_kwargs={}, lctx=2, rctx=1, chunk_size=2)),
    ("cfmr", "xl", {"kernel_size": 5}, dict(proj="conv1d", proj_kwargs={"dim": 32, "num_layers": 2})),
    ("cfmr", "rel", {"kernel_size": 5, "casual_conv1d": True},
     dict(proj="conv2d", proj_kwargs={"conv_channels": 8, "num_layers": 2}, lctx=3, rctx=0,
          chunk_size=1)),
    ("xfmr", "abs", {}, dict(proj="linear", proj_kwargs={}, lctx=1, rctx=1, chunk_size=4))])
def test_encoder_xl_window_causal_backward_vs_oracle(device, arch, pose, kw, top):
    """TransformerEncoder.forward (encoder.py:57-106) with Transformer-XL positions, context windows
    and the causal conformer convolution: the gradient of every parameter against autograd through
    the CPU oracle (eval-mode statistics)"""
    from aps_amd.asr.transformer import TransformerEncoder
    from oracle import encoder_oracle as eo
    torch.manual_seed(17)
    pose_kwargs = {"dropout": 0, "lradius": 5, "rradius": 3} if pose == "rel" else {"dropout": 0}
    kw = dict(kw)
    causal = kw.pop("casual_conv1d", False)
    enc = TransformerEncoder(arch, 24, num_layers=2, pose=pose, pose_kwargs=pose_kwargs,
                             arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 96,
                                          "att_dropout": 0, "ffn_dropout": 0, **kw}, **top).eval()
    if causal:  # (the registered layers do not pass the flag on, in the reference either)
        for layer in enc.encoder.layers:
            layer.padding = kw["kernel_size"] - 1
            layer.convolution[2].padding = (0,)
    g = torch.Generator().manual_seed(18)
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(0.1 * torch.randn(m.num_features, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.num_features, generator=g))
    x = torch.randn(3, 60, 24, generator=g)
    window = None
    lens = torch.tensor([60, 47, 33])
    if "lctx" in top:  # (padded queries whose window holds no valid key are NaN in the reference)
        window, lens = (top["chunk_size"], top["lctx"], top["rctx"]), None
    trainable = {n for n, p in enc.named_parameters() if p.requires_grad}
    sd = {k: (v.detach().clone().requires_grad_(True) if k in trainable else v.detach().clone())
          for k, v in enc.state_dict().items()}
    ref, _ = eo.generic_encoder(sd, x, lens, arch=arch, pose=pose, num_layers=2, nhead=2, lradius=5,
                                rradius=3, kernel_size=kw.get("kernel_size", 15),
                                pre_norm=arch == "cfmr", proj=top["proj"], window=window,
                                casual_conv1d=causal)
    valid = torch.ones(ref.shape[:2], dtype=torch.bool)
    if lens is not None:
        rn = _
        valid = torch.arange(ref.shape[1])[None] < rn[:, None]
    up = torch.randn(ref.shape, generator=g) * valid[..., None]
    (torch.where(valid[..., None], ref, torch.zeros_like(ref)) * up).sum().backward()
    enc = enc.to(device)
    out, _ = enc(x.to(device), None if lens is None else lens.to(device))
    check(out.cpu()[valid], ref.detach()[valid], f"{arch}_{pose} output")
    (out * up.to(device)).sum().backward()
    missing = [n for n, p in enc.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, f"no gradient reached {missing}"
    worst = 0.0
    for name, p in enc.named_parameters():
        if not p.requires_grad:  # (the frozen sinusoid frequencies)
            continue
        want = sd[name].grad
        assert want is not None, name
        err = rel_err(p.grad, want)
        worst = max(worst, err)
        assert err <= 2e-4, f"{name}: gradient error {err:.3e}"
    print(f"[grad] {arch}_{pose} {top}: worst parameter-gradient error {worst:.2e}")


@pytest.mark.parametrize("kind,norm,train", [("linear", "LN", False), ("linear", "BN", True),
                                             ("conv1d", "BN", True), ("conv1d", "LN", False),
                                             ("conv1d", "BN", False), ("linear_long", "LN", False)])
def test_projection_backward(device, kind, norm, train):
    """the linear and conv1d (TDNN) projections in front of the encoder (proj.py:31-101) under
    autograd: Linear / Conv1d -> Normalize1d -> ReLU with "LN" = GroupNorm(1, D) over the whole
    utterance and "BN" on batch statistics in train(), against the same torch modules on the CPU"""
    import copy
    from aps_amd.asr.transformer.proj import get_xfmr_proj
    torch.manual_seed(21)
    T = 400 if kind == "linear_long" else 37  # (400 x 64 values per utterance: the wide-row kernel)
    if kind.startswith("linear"):
        proj = get_xfmr_proj("linear", 24, 64, norm=norm)
    else:
        proj = get_xfmr_proj("conv1d", 24, 64, norm=norm, dim=32, num_layers=2)
    proj = proj.train(train)
    for m in proj.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.GroupNorm)):
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.3)
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.7, 1.4)
    ref = copy.deepcopy(proj)
    x = torch.randn(3, T, 24)
    xr = x.clone().requires_grad_(True)
    if kind.startswith("linear"):
        h = ref.norm.norm(ref.proj(xr).transpose(1, 2)).transpose(1, 2)
        want = F.relu(h)
    else:
        h = xr.transpose(1, 2)
        for blk in ref.conv.enc_layers:
            h = F.relu(blk.norm.norm(blk.conv(h)))
        want = h.transpose(1, 2)
    up = torch.randn(want.shape)
    want.backward(up)
    proj = proj.to(device)
    xd = x.to(device).requires_grad_(True)
    out, _ = proj(xd, None)
    check(out, want, f"{kind} {norm} output")
    out.backward(up.to(device))
    check(xd.grad, xr.grad, f"{kind} {norm} g_x")
    for (name, p), q in zip(proj.named_parameters(), ref.parameters()):
        if q.grad.abs().max().item() < 1e-5 * max(1.0, up.abs().max().item()):
            # a bias in front of batch / utterance statistics: the exact gradient is zero
            assert p.grad.abs().max().item() < 1e-3, name
            continue
        check(p.grad, q.grad, f"{kind} {norm} {name}")
    if train and norm == "BN":
        for (name, b), c in zip(proj.named_buffers(), ref.buffers()):
            check(b.float(), c.float(), f"{kind} {name}")


@pytest.mark.parametrize("feats,use_power", [("fbank-log-cmvn", False), ("fbank-log", False),
                                             ("fbank-log-cmvn", True)])
def test_trainable_mel_filters(device, feats, use_power):
    """AsrTransform(requires_grad=True) (asr.py:383-400, 833): the mel projection leaves the fused
    feature launch and runs as a GEMM under autograd; features and the gradient of the filters against
    autograd through the oracle's chain with the same matrix as a leaf"""
    from aps_amd.transform import AsrTransform
    from oracle import aps_oracle as ao
    g = torch.Generator().manual_seed(31)
    wav = 0.1 * torch.randn(3, 6000, generator=g)
    t = AsrTransform(feats=feats, frame_len=400, frame_hop=160, window="hamm", num_mels=40,
                     use_power=use_power, requires_grad=True)
    mel = [m for m in t.transform if hasattr(m, "filters")][0]
    assert mel.filters.requires_grad
    w = mel.filters.detach().clone().requires_grad_(True)
    packed = ao.stft(wav, 400, 160, "hamm", True, False, 0.97, True, False, "librosa")
    want = ao.spectral_chain(packed, feats.split("-"), w, use_power)
    up = torch.randn(want.shape, generator=g)
    (want * up).sum().backward()
    t = t.to(device)
    out, _ = t(wav.to(device), None)
    assert out.requires_grad
    check(out, want, f"{feats} with trainable filters")
    (out * up.to(device)).sum().backward()
    check(mel.filters.grad, w.grad, f"{feats} g_filters")
    with torch.no_grad():  # the fused launch again once nothing records
        fused, _ = t(wav.to(device), None)
    check(fused, want, f"{feats} fused")


def test_abs_pow_mel_chain_backward(device):
    """AsrTransform("abs-pow-mel-log-cmvn") on an enhanced spectrogram that requires grad (the joint
    model's feature chain with the power spectrum, asr.py:946-971): output and the gradient that
    reaches the spectrogram against autograd through the explicit chain"""
    from aps_amd.cplx import ComplexTensor
    from aps_amd.transform import AsrTransform
    from oracle import aps_oracle as ao
    g = torch.Generator().manual_seed(37)
    yr, yi = torch.randn(2, 30, 257, generator=g), torch.randn(2, 30, 257, generator=g)
    t = AsrTransform(feats="abs-pow-mel-log-cmvn", frame_len=512, frame_hop=256, window="sqrthann",
                     num_mels=40)
    mel = [m for m in t.transform if hasattr(m, "filters")][0].filters.detach()
    rr, ri = yr.clone().requires_grad_(True), yi.clone().requires_grad_(True)
    x = ((rr + ao.EPSILON)**2 + ri**2).sqrt()**2
    want = ao.cmvn(ao.log_feature(F.linear(x, mel), ao.EPSILON), eps=ao.EPSILON)
    up = torch.randn(want.shape, generator=g)
    (want * up).sum().backward()
    t = t.to(device)
    dr, di = yr.to(device).requires_grad_(True), yi.to(device).requires_grad_(True)
    out, _ = t(ComplexTensor(dr, di), None)
    check(out, want, "abs-pow-mel-log-cmvn")
    (out * up.to(device)).sum().backward()
    check(dr.grad, rr.grad, "abs-pow chain g_real")
    check(di.grad, ri.grad, "abs-pow chain g_imag")


# ------------------------------------------------------------------------------------------------
# transformer decoder (the attention over the encoder output, the token embedding)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dh,drop", [(32, 0.0), (64, 0.25), (16, 0.25)])
def test_cross_attention_backward(device, dh, drop):
    """attention_cross under autograd (decoder.py:78-86), with and without dropout on the weights,
    against autograd through softmax(q k^T / sqrt(dh)) (* mask) v in float64"""
    from aps_amd.grad_ops import AttentionCrossFn
    torch.manual_seed(41 + dh)
    N, Tq, Tk, H = 3, 9, 23, 2
    seed = 31415926535
    q = torch.randn(N, Tq, H * dh)
    kv = torch.randn(N, Tk, 2 * H * dh)
    lens = torch.tensor([Tk, 14, 6])
    keep = _keep(seed, N * H * Tq * Tk, drop).view(N, H, Tq, Tk).double() if drop > 0 else 1.0
    qr, kr = q.double().requires_grad_(True), kv.double().requires_grad_(True)
    k, v = kr.view(N, Tk, 2, H, dh).unbind(2)
    s = torch.einsum("nihd,njhd->nhij", qr.view(N, Tq, H, dh), k) / dh**0.5
    s = s.masked_fill((torch.arange(Tk)[None] >= lens[:, None])[:, None, None, :], float("-inf"))
    want = torch.einsum("nhij,njhd->nihd", torch.softmax(s, -1) * keep, v).reshape(N, Tq, H * dh)
    up = torch.randn(N, Tq, H * dh)
    (want * up.double()).sum().backward()
    qd, kd = q.to(device).requires_grad_(True), kv.to(device).requires_grad_(True)
    out = AttentionCrossFn.apply(qd, kd, lens.to(device), H, drop, seed)
    check(out, want.detach().float(), "cross attention", 1e-5)
    out.backward(up.to(device))
    check(qd.grad, qr.grad.float(), "cross attention g_q")
    check(kd.grad, kr.grad.float(), "cross attention g_kv")


@pytest.mark.parametrize("drop", [0.0, 0.25])
def test_attention_with_additive_mask_tensors_under_autograd(device, drop):
    """additive mask TENSORS in training (round 5; the reference passes `src_mask` / `tgt_mask` /
    `memory_mask` to its attentions under autograd, aps/asr/transformer/impl.py:104-114,
    decoder.py:150-186; rounds 1-4 raised NotImplementedError): nn_ops.attention_core(add_mask = a causal
    0 / -inf mask plus a bias) and nn_ops.attention_cross(add_mask = a bias with a masked column and a fully
    masked query row), with and without dropout on the weights, against autograd through the explicit
    float64 forms.  The masks are data: no gradient into them, -inf pairs weigh 0, a row without keys is 0."""
    from aps_amd import nn_ops
    torch.manual_seed(61)
    N, T, H, dh = 3, 17, 2, 32
    seed = 1357911
    qkv = torch.randn(N, T, 3 * H * dh)
    lens = torch.tensor([T, 11, 5])
    mask = 0.3 * torch.randn(T, T)
    mask = mask.masked_fill(torch.arange(T)[None, :] > torch.arange(T)[:, None], float("-inf"))
    mask[3, :] = float("-inf")   # a query row without any visible key: 0 (and no gradient) in the eval AND the
    #                              training kernels -- the library's one semantic (torch's softmax gives NaN there)
    drop_mod = torch.nn.Dropout(drop).train()
    keep = 1.0
    if drop > 0:
        import aps_amd.grad_ops as go
        saved = go.draw_seed
        go.draw_seed = lambda: seed
        keep = _keep(seed, N * H * T * T, drop).view(N, H, T, T).double()
    try:
        qr = qkv.double().requires_grad_(True)
        q, k, v = [m.reshape(N, T, H, dh) for m in qr.chunk(3, -1)]
        s = torch.einsum("nlhd,nshd->nhls", q, k) / dh**0.5 + mask.double()[None, None]
        s = s.masked_fill((torch.arange(T)[None] >= lens[:, None])[:, None, None, :], float("-inf"))
        dead0 = torch.isinf(s).all(-1, keepdim=True)
        p0 = torch.softmax(s.masked_fill(dead0, 0.0), -1) * (~dead0)
        want = torch.einsum("nhls,nshd->nlhd", p0 * keep, v).reshape(N, T, H * dh)
        up = torch.randn(N, T, H * dh)
        (want * up.double()).sum().backward()
        qd = qkv.to(device).requires_grad_(True)
        out = nn_ops.attention_core(qd, H, lens.to(device), add_mask=mask.to(device), dropout=drop_mod)
        check(out, want.detach().float(), "self attention with an additive mask", 1e-5)
        assert float(out[:, 3].detach().abs().max()) == 0
        with torch.no_grad():   # the eval kernels (nn.hip) on the same mask: the same zeros, the same values at p = 0
            ev = nn_ops.attention_core(qkv.to(device), H, lens.to(device), add_mask=mask.to(device))
        assert float(ev[:, 3].abs().max()) == 0 and not torch.isnan(ev).any()
        if drop == 0:
            check(ev, want.detach().float(), "eval kernels with the same additive mask", 1e-5)
        with pytest.raises(RuntimeError):   # a mask on another device is refused before any pointer is taken
            nn_ops.attention_core(qd, H, lens.to(device), add_mask=mask, dropout=drop_mod)
        out.backward(up.to(device))
        check(qd.grad, qr.grad.float(), "self attention with an additive mask: g_qkv")
        # cross attention with a memory_mask
        Tq, Tk = 7, 19
        qc, kv = torch.randn(N, Tq, H * dh), torch.randn(N, Tk, 2 * H * dh)
        klens = torch.tensor([Tk, 12, 4])
        mmask = 0.3 * torch.randn(Tq, Tk)
        mmask[:, 1] = float("-inf")
        mmask[4, :] = float("-inf")
        keep2 = _keep(seed, N * H * Tq * Tk, drop).view(N, H, Tq, Tk).double() if drop > 0 else 1.0
        qr2, kr2 = qc.double().requires_grad_(True), kv.double().requires_grad_(True)
        kk, vv = kr2.view(N, Tk, 2, H, dh).unbind(2)
        s2 = torch.einsum("nihd,njhd->nhij", qr2.view(N, Tq, H, dh), kk) / dh**0.5 + mmask.double()[None, None]
        s2 = s2.masked_fill((torch.arange(Tk)[None] >= klens[:, None])[:, None, None, :], float("-inf"))
        dead = torch.isinf(s2).all(-1, keepdim=True)
        p2 = torch.softmax(s2.masked_fill(dead, 0.0), -1) * (~dead)
        want2 = torch.einsum("nhij,njhd->nihd", p2 * keep2, vv).reshape(N, Tq, H * dh)
        up2 = torch.randn(N, Tq, H * dh)
        (want2 * up2.double()).sum().backward()
        qd2, kd2 = qc.to(device).requires_grad_(True), kv.to(device).requires_grad_(True)
        out2 = nn_ops.attention_cross(qd2, kd2, H, klens.to(device), add_mask=mmask.to(device), dropout=drop_mod)
        check(out2, want2.detach().float(), "cross attention with a memory_mask", 1e-5)
        assert float(out2[:, 4].abs().max()) == 0
        out2.backward(up2.to(device))
        check(qd2.grad, qr2.grad.float(), "cross attention with a memory_mask: g_q")
        check(kd2.grad, kr2.grad.float(), "cross attention with a memory_mask: g_kv")
    finally:
        if drop > 0:
            go.draw_seed = saved


@pytest.mark.parametrize("pre_norm", [False, True])
def test_transformer_decoder_backward_vs_oracle(device, pre_norm):
    """TorchTransformerDecoder.forward (decoder.py:128-186) under autograd: the gradient of every
    parameter (embedding, both attentions, feed-forward, norms, output projection) and of the encoder
    output against autograd through the oracle"""
    from aps_amd.asr.transformer.decoder import TorchTransformerDecoder
    from oracle import encoder_oracle as eo
    torch.manual_seed(51)
    dec = TorchTransformerDecoder(
        40, pose_kwargs={"dropout": 0}, num_layers=2,
        arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 96, "pre_norm": pre_norm,
                     "att_dropout": 0, "ffn_dropout": 0}).eval()
    g = torch.Generator().manual_seed(52)
    enc_out = torch.randn(3, 21, 64, generator=g)
    enc_len = torch.tensor([21, 17, 9])
    tgt = torch.randint(0, 38, (3, 8), generator=g)  # (tokens 38, 39 never occur: zero embedding rows)
    tgt_len = torch.tensor([8, 5, 8])
    trainable = {n for n, p in dec.named_parameters() if p.requires_grad}
    sd = {k: (v.detach().clone().requires_grad_(True) if k in trainable else v.detach().clone())
          for k, v in dec.state_dict().items()}
    er = enc_out.clone().requires_grad_(True)
    ref = eo.transformer_decoder(sd, er, enc_len, tgt, tgt_len, 2, 2, pre_norm=pre_norm)
    valid = (torch.arange(8)[None] < tgt_len[:, None])[..., None]
    up = torch.randn(ref.shape, generator=g) * valid
    (torch.where(valid, ref, torch.zeros_like(ref)) * up).sum().backward()
    dec = dec.to(device)
    ed = enc_out.to(device).requires_grad_(True)
    out = dec(ed, enc_len.to(device), tgt.to(device), tgt_len.to(device))
    check(out.cpu() * valid, ref.detach() * valid, "decoder output")
    (out * up.to(device)).sum().backward()
    check(ed.grad, er.grad, "decoder g_enc_out")
    worst = 0.0
    for name, p in dec.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, name
        err = rel_err(p.grad, sd[name].grad)
        worst = max(worst, err)
        assert err <= 2e-4, f"{name}: gradient error {err:.3e}"
    print(f"[grad] transformer decoder (pre_norm={pre_norm}): worst parameter-gradient error {worst:.2e}")


def test_transformer_decoder_trains_with_dropout(device):
    """train() mode with the reference's default dropouts (attention weights 0.1, feed-forward /
    residual 0.1, position 0.1): a step runs, every parameter that takes part receives a finite gradient,
    the same seed gives the same step, another seed another mask"""
    from aps_amd.asr.transformer.decoder import TorchTransformerDecoder
    dec = TorchTransformerDecoder(
        40, pose_kwargs={"dropout": 0.1}, num_layers=2,
        arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 96}).to(device).train()
    g = torch.Generator().manual_seed(53)
    enc_out = torch.randn(3, 21, 64, generator=g).to(device)
    tgt = torch.randint(0, 40, (3, 8), generator=g).to(device)
    enc_len, tgt_len = torch.tensor([21, 17, 9]).to(device), torch.tensor([8, 5, 8]).to(device)
    runs = []
    for seed in (7, 7, 8):
        torch.manual_seed(seed)
        dec.zero_grad()
        out = dec(enc_out, enc_len, tgt, tgt_len)
        out.square().mean().backward()
        runs.append((out.detach().clone(), {n: p.grad.clone() for n, p in dec.named_parameters()
                                            if p.grad is not None}))
    assert torch.equal(runs[0][0], runs[1][0]) and not torch.equal(runs[0][0], runs[2][0])
    for n, gr in runs[0][1].items():
        assert torch.isfinite(gr).all(), n
        assert torch.equal(gr, runs[1][1][n]), n
    assert len(runs[0][1]) >= 30


def test_xfmr_asr_backward_vs_oracle(device):
    """asr@xfmr end to end under autograd: fbank features -> conv2d projection -> transformer encoder
    (+ CTC branch) -> transformer decoder; the gradient of every parameter of the model against autograd
    through the oracle's encoder + decoder on the same weights"""
    from aps_amd.libs import aps_asr_nnet
    from aps_amd.transform import AsrTransform
    from oracle import aps_oracle as orc
    from oracle import encoder_oracle as eo
    torch.manual_seed(91)
    arch = {"att_dim": 64, "nhead": 2, "feedforward_dim": 128, "att_dropout": 0, "ffn_dropout": 0}
    net = aps_asr_nnet("asr@xfmr")(
        40, 41, sos=39, eos=39, ctc=True,
        asr_transform=AsrTransform(feats="fbank-log-cmvn", frame_len=400, frame_hop=160,
                                   window="hamm", num_mels=40),
        enc_type="xfmr",
        enc_kwargs=dict(num_layers=2, proj="conv2d", proj_kwargs={"conv_channels": 8, "num_layers": 2},
                        pose="abs", pose_kwargs={"dropout": 0}, arch_kwargs=dict(arch)),
        dec_kwargs=dict(num_layers=2, pose_kwargs={"dropout": 0}, arch_kwargs=dict(arch))).eval()
    g = torch.Generator().manual_seed(92)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(0.05 * torch.randn(m.num_features, generator=g))
            m.running_var.copy_(0.8 + 0.4 * torch.rand(m.num_features, generator=g))
    wav = 0.1 * torch.randn(3, 12000, generator=g)
    wav_len = torch.tensor([12000, 9000, 7000])
    y = torch.randint(0, 40, (3, 7), generator=g)
    y_len = torch.tensor([7, 5, 3])
    trainable = {n for n, p in net.named_parameters() if p.requires_grad}
    sd = {k: (v.detach().clone().requires_grad_(True) if k in trainable else v.detach().clone())
          for k, v in net.state_dict().items()}
    feats = orc.asr_features(wav, "fbank-log-cmvn", frame_len=400, frame_hop=160,
                             window_name="hamm", num_mels=40)
    n = torch.tensor([orc.num_frames(int(v), 512, 160, False) for v in wav_len])
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    enc_out, enc_len = eo.generic_encoder(enc, feats, n, "xfmr", "abs", 2, 2)
    ref = eo.transformer_decoder(sd, enc_out, enc_len, y, y_len, 2, 2, prefix="decoder.")
    ref_ctc = F.linear(enc_out, sd["ctc.weight"], sd["ctc.bias"])
    vd = (torch.arange(7)[None] < y_len[:, None])[..., None]
    ve = (torch.arange(enc_out.shape[1])[None] < enc_len[:, None])[..., None]
    u1 = torch.randn(ref.shape, generator=g) * vd
    u2 = torch.randn(ref_ctc.shape, generator=g) * ve
    ((torch.where(vd, ref, torch.zeros_like(ref)) * u1).sum() +
     (torch.where(ve, ref_ctc, torch.zeros_like(ref_ctc)) * u2).sum()).backward()
    net = net.to(device)
    dec_out, enc_ctc, out_len = net(wav.to(device), wav_len.to(device), y.to(device), y_len.to(device))
    assert out_len.cpu().tolist() == enc_len.tolist()
    ((dec_out * u1.to(device)).sum() + (enc_ctc * u2.to(device)).sum()).backward()
    worst, seen = 0.0, 0
    for name, p in net.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, f"no gradient reached {name}"
        want = sd[name].grad
        assert want is not None, name
        err = rel_err(p.grad, want)
        worst, seen = max(worst, err), seen + 1
        assert err <= 2e-4, f"{name}: gradient error {err:.3e}"
    print(f"[grad] asr@xfmr: {seen} tensors, worst parameter-gradient error {worst:.2e}")


def test_xfmr_asr_against_the_reference_models_own_step(device):
    """asr@xfmr with the weights of the reference's own model (tests/golden/xfmr_asr.npz, recorded by
    make_golden.py:gen_xfmr_asr): the forward outputs and the gradient of every parameter against what
    the reference's modules computed under torch autograd -- no oracle in between"""
    from aps_amd.libs import aps_asr_nnet
    from aps_amd.transform import AsrTransform
    from tests.conftest import golden
    g = golden("xfmr_asr")
    arch = {"att_dim": 64, "nhead": 2, "feedforward_dim": 128, "att_dropout": 0, "ffn_dropout": 0}
    net = aps_asr_nnet("asr@xfmr")(
        40, 41, sos=39, eos=39, ctc=True,
        asr_transform=AsrTransform(feats="fbank-log-cmvn", frame_len=400, frame_hop=160,
                                   window="hamm", num_mels=40),
        enc_type="xfmr",
        enc_kwargs=dict(num_layers=2, proj="conv2d", proj_kwargs={"conv_channels": 8, "num_layers": 2},
                        pose="abs", pose_kwargs={"dropout": 0}, arch_kwargs=dict(arch)),
        dec_kwargs=dict(num_layers=2, pose_kwargs={"dropout": 0}, arch_kwargs=dict(arch)))
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    net = net.eval().to(device)
    dec_out, enc_ctc, enc_len = net(g["wav"].to(device), g["wav_len"].to(device), g["y"].to(device),
                                    g["y_len"].to(device))
    assert enc_len.cpu().tolist() == g["enc_len"].tolist()
    vd = (torch.arange(dec_out.shape[1])[None] < g["y_len"][:, None])[..., None]
    ve = (torch.arange(enc_ctc.shape[1])[None] < g["enc_len"][:, None])[..., None]
    check(dec_out.cpu() * vd, g["dec_out"] * vd, "asr@xfmr decoder output vs the reference")
    check(enc_ctc.cpu() * ve, g["enc_ctc"] * ve, "asr@xfmr CTC branch vs the reference")
    ((dec_out * g["probe_dec"].to(device)).sum() + (enc_ctc * g["probe_ctc"].to(device)).sum()).backward()
    names = [k[5:] for k in g if k.startswith("grad.")]
    params = dict(net.named_parameters())
    worst = 0.0
    for k in names:
        assert params[k].grad is not None, f"no gradient reached {k}"
        err = rel_err(params[k].grad, g["grad." + k])
        worst = max(worst, err)
        assert err <= 2e-4, f"{k}: gradient error {err:.3e}"
    print(f"[grad] asr@xfmr vs the reference's own step: {len(names)} tensors, worst {worst:.2e}")


def test_joint_against_the_reference_modules_own_step(device):
    """the north-star model with the weights of the reference's own modules (tests/golden/
    joint_mvdr_cfmr.npz) under autograd, ragged lengths: the gradient of every parameter against what the
    reference's EnhTransform -> RNNMaskMvdr -> AsrTransform -> CtcASR computed (joint_mvdr_cfmr_grad.npz,
    make_golden.py:gen_joint_grad) -- no oracle in between"""
    from tests.conftest import golden
    from tests.test_gpu_joint import SMALL_ENC, build_joint
    g, ref = golden("joint_mvdr_cfmr"), golden("joint_mvdr_cfmr_grad")
    net = build_joint(40, 48, 64, 32, 50, SMALL_ENC)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    net = net.eval().to(device)
    enc_out, enc_ctc, enc_len = net(g["wav"].to(device), g["lens"].to(device))
    assert torch.equal(enc_len.cpu(), g["ragged.enc_len"])
    check(enc_out, g["ragged.enc_out"], "joint encoder output vs the reference")
    ((enc_out * ref["probe_out"].to(device)).sum() + (enc_ctc * ref["probe_ctc"].to(device)).sum()).backward()
    names = [k[5:] for k in ref if k.startswith("grad.")]
    params = dict(net.named_parameters())
    assert len(names) >= 60 and not [k for k in names if k not in params]
    worst = 0.0
    for k in names:
        if k.endswith("gvec.bias"):  # (a shift of a softmax input: exact gradient zero)
            continue
        assert params[k].grad is not None, f"no gradient reached {k}"
        err = rel_err(params[k].grad, ref["grad." + k])
        worst = max(worst, err)
        assert err <= 3e-4, f"{k}: gradient error {err:.3e}"
    print(f"[grad] joint vs the reference's own step: {len(names)} tensors, worst {worst:.2e}")


@pytest.mark.parametrize("ragged,mask_norm", [(False, True), (True, True), (True, False)])
def test_mvdr_backward_with_the_implicit_noise_mask(device, ragged, mask_norm):
    """MvdrBeamformer.forward(mask_s, x) without a noise mask under autograd (mvdr.py:131-135:
    Rn = estimate_covar(1 - processed speech mask, X), what RNNMaskMvdr(mask_net_noise=False) trains
    through): the gradients of the speech mask and of the ChannelAttention parameters against torch
    autograd through the oracle"""
    import oracle.aps_oracle as orc
    from aps_amd.asr.filter.mvdr import MvdrBeamformer
    from aps_amd.cplx import ComplexTensor
    g = torch.Generator().manual_seed(77)
    N, Cn, F, T = 3, 4, 33, 50
    xr, xi = torch.randn(N, Cn, F, T, generator=g), torch.randn(N, Cn, F, T, generator=g)
    mask = torch.sigmoid(torch.randn(N, T, F, generator=g))
    lens = torch.tensor([50, 41, 30]) if ragged else None
    ur, ui = torch.randn(N, T, F, generator=g), torch.randn(N, T, F, generator=g)
    torch.manual_seed(5)
    mvdr = MvdrBeamformer(F, att_dim=24, mask_norm=mask_norm)
    att = [p.detach().clone().requires_grad_(True) for p in
           (mvdr.ref.proj.weight, mvdr.ref.proj.bias, mvdr.ref.gvec.weight, mvdr.ref.gvec.bias)]
    mref = mask.clone().requires_grad_(True)
    yr, yi, _ = orc.mvdr_forward(mref, xr, xi, att, None, lens, mask_norm)
    ((yr * ur).sum() + (yi * ui).sum()).backward()
    mvdr = mvdr.to(device)
    md = mask.to(device).requires_grad_(True)
    y = mvdr(md, ComplexTensor(xr.to(device), xi.to(device)), None,
             None if lens is None else lens.to(device))
    check(y.real, yr, "beam output, real")
    check(y.imag, yi, "beam output, imag")
    ((y.real * ur.to(device)).sum() + (y.imag * ui.to(device)).sum()).backward()
    check(md.grad, mref.grad, "g_mask_s (through Rs and, with the opposite sign, through Rn)")
    check(mvdr.ref.proj.weight.grad, att[0].grad, "ChannelAttention proj.weight")
    check(mvdr.ref.proj.bias.grad, att[1].grad, "ChannelAttention proj.bias")
    check(mvdr.ref.gvec.weight.grad, att[2].grad, "ChannelAttention gvec.weight")


@pytest.mark.parametrize("kind,hidden,layers,bidir,ragged,proj", [
    ("GRU", 48, 1, False, False, 0), ("GRU", 40, 2, True, True, 0), ("RNN_TANH", 32, 2, False, True, 0),
    ("RNN_RELU", 24, 1, True, True, 0), ("LSTM", 48, 2, True, True, 0), ("GRU", 512, 1, False, True, 0),
    ("LSTM", 64, 1, False, False, 24), ("LSTM", 48, 2, True, True, 16)])
def test_step_recurrences_train_on_hip(device, kind, hidden, layers, bidir, ragged, proj):
    """var_len_rnn_forward (aps/asr/base/component.py:26-55) under autograd for the recurrences that have no
    persistent kernel -- nn.GRU, nn.RNN (tanh / relu), nn.LSTM of a width the LSTM kernels do not take, nn.LSTM
    with a projection (proj_size): output,
    input gradient and every parameter gradient against torch's own layer in float64 on the CPU, packed
    sequences included.  No torch recurrent kernel runs on the GPU side (the step path is HIP)."""
    import copy
    from aps_amd.asr.base.encoder import PyTorchRNN, var_len_rnn_forward
    torch.manual_seed(len(kind) + hidden)
    N, T, D = 4, 11, 20
    rnn = PyTorchRNN(kind, D, hidden, num_layers=layers, bidirectional=bidir, proj_size=proj if proj else -1)
    x = torch.randn(N, T, D)
    lens = torch.tensor([11, 7, 9, 4]) if ragged else None
    dirs = 2 if bidir else 1
    up = torch.randn(N, T, (proj if proj else hidden) * dirs)
    ref = copy.deepcopy(rnn).double()
    xr = x.double().requires_grad_(True)
    if lens is not None:
        packed = torch.nn.utils.rnn.pack_padded_sequence(xr, lens.tolist(), batch_first=True,
                                                         enforce_sorted=False)
        yr, _ = torch.nn.utils.rnn.pad_packed_sequence(ref(packed)[0], batch_first=True, total_length=T)
    else:
        yr, _ = ref(xr)
    (yr * up.double()).sum().backward()
    net = copy.deepcopy(rnn).to(device)
    xd = x.to(device).requires_grad_(True)
    called = []
    orig = net.forward
    net.forward = lambda *a, **k: called.append(1) or orig(*a, **k)   # torch's own layer must not run
    yd = var_len_rnn_forward(net, xd, None if lens is None else lens.to(device))
    assert not called, "the torch recurrent layer ran"
    assert yd.shape == yr.shape
    check(yd, yr, f"{kind} output")
    (yd * up.to(device)).sum().backward()
    check(xd.grad, xr.grad, f"{kind} g_x")
    for (name, p), q in zip(net.named_parameters(), ref.parameters()):
        check(p.grad, q.grad, f"{kind} {name}")
