"""
GPU parity of the Transformer decoder forward (aps/asr/transformer/decoder.py) and the `asr@xfmr`
encoder-decoder model: activations recorded from the reference module (fixtures
decoder_xfmr_post / decoder_xfmr_pre) and the CPU oracle for the composition.  Tolerance 1e-4 of
the output scale.
"""
import pytest
import torch

from tests.conftest import golden, assert_close, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def small_decoder(pre_norm):
    from aps_amd.asr.transformer.decoder import TorchTransformerDecoder
    return TorchTransformerDecoder(
        40, pose_kwargs={"dropout": 0}, num_layers=2,
        arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 128, "pre_norm": pre_norm,
                     "att_dropout": 0, "ffn_dropout": 0})


@pytest.mark.parametrize("tag,pre_norm", [("decoder_xfmr_post", False), ("decoder_xfmr_pre", True)])
def test_decoder_golden(device, tag, pre_norm):
    g = golden(tag)
    dec = small_decoder(pre_norm)
    dec.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd.")}, strict=True)
    dec = dec.eval().to(device)
    enc_out, tgt = g["enc_out"].to(device), g["tgt_pad"].to(device)
    out = dec(enc_out, g["enc_len"].to(device), tgt, g["tgt_len"].to(device))
    assert out.shape == g["out_len"].shape
    assert_close(out, g["out_len"], TOL, tag + " with lengths")
    assert_close(dec(enc_out, None, tgt, None), g["out_full"], TOL, tag + " without lengths")
    _, emb = dec.step(enc_out.transpose(0, 1), tgt[:, :4])
    step_out, _ = dec.step(enc_out.transpose(0, 1), tgt[:, 4:6], pre_emb=emb, out_idx=-1)
    assert_close(step_out, g["step_out"], TOL, tag + " step(pre_emb, out_idx)")
    # the reference's T x N x D layer call convention
    layer = dec.decoder.layers[0]
    x = torch.randn(9, 3, 64, device=device)
    pad = torch.arange(9, device=device)[None, :] >= g["tgt_len"].to(device)[:, None]
    mpad = torch.arange(17, device=device)[None, :] >= g["enc_len"].to(device)[:, None]
    a = layer(x, enc_out.transpose(0, 1), tgt_key_padding_mask=pad, memory_key_padding_mask=mpad)
    b = layer.run(x.transpose(0, 1).contiguous(), enc_out, g["tgt_len"].to(device),
                  g["enc_len"].to(device)).transpose(0, 1)
    assert torch.equal(a, b)


def test_decoder_wide_vs_oracle(device):
    """512-wide, 8 heads (head_dim 64), 6 layers, 30 target tokens over 100 encoder frames"""
    from aps_amd.asr.transformer.decoder import TorchTransformerDecoder
    from oracle import encoder_oracle as eo
    torch.manual_seed(81)
    dec = TorchTransformerDecoder(
        500, pose_kwargs={"dropout": 0}, num_layers=6,
        arch_kwargs={"att_dim": 512, "nhead": 8, "feedforward_dim": 1024, "pre_norm": True,
                     "att_dropout": 0, "ffn_dropout": 0}).eval()
    g = torch.Generator().manual_seed(82)
    enc_out = torch.randn(4, 100, 512, generator=g)
    enc_len = torch.tensor([100, 90, 64, 33])
    tgt = torch.randint(0, 500, (4, 30), generator=g)
    tgt_len = torch.tensor([30, 21, 30, 5])
    sd = {k: v.detach() for k, v in dec.state_dict().items()}
    ref = eo.transformer_decoder(sd, enc_out, enc_len, tgt, tgt_len, 6, 8, pre_norm=True)
    out = dec.to(device)(enc_out.to(device), enc_len.to(device), tgt.to(device), tgt_len.to(device))
    assert_close(out, ref, TOL, "wide decoder")


def test_xfmr_asr_forward(device):
    """asr@xfmr: fbank features -> transformer encoder (+ CTC branch) -> transformer decoder,
    against the oracle's encoder + decoder on the same weights"""
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
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    feats = orc.asr_features(wav, "fbank-log-cmvn", frame_len=400, frame_hop=160,
                             window_name="hamm", num_mels=40)
    n = torch.tensor([orc.num_frames(int(v), 512, 160, False) for v in wav_len])
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    enc_out, enc_len = eo.generic_encoder(enc, feats, n, "xfmr", "abs", 2, 2)
    ref = eo.transformer_decoder(sd, enc_out, enc_len, y, y_len, 2, 2, prefix="decoder.")
    ref_ctc = torch.nn.functional.linear(enc_out, sd["ctc.weight"], sd["ctc.bias"])
    net = net.to(device)
    dec_out, enc_ctc, out_len = net(wav.to(device), wav_len.to(device), y.to(device),
                                    y_len.to(device))
    assert out_len.cpu().tolist() == enc_len.tolist()
    assert_close(enc_ctc, ref_ctc, TOL, "CTC branch")
    assert_close(dec_out, ref, TOL, "decoder output")


# ------------------------------------------------------------------------------------------------
# RNN attention decoder (asr@att)
# ------------------------------------------------------------------------------------------------
ATT_CASES = {"att_decoder_ctx": ("ctx", {"att_dim": 32}, False),
             "att_decoder_dot": ("dot", {"att_dim": 32, "scaled": True}, True),
             "att_decoder_loc": ("loc", {"att_dim": 32, "conv_channels": 4, "loc_context": 5}, False),
             "att_decoder_mhctx": ("mhctx", {"att_dim": 16, "att_head": 3}, False),
             "att_decoder_mhdot": ("mhdot", {"att_dim": 16, "att_head": 4, "scaled": True}, True),
             "att_decoder_mhloc": ("mhloc", {"att_dim": 16, "att_head": 2, "conv_channels": 3,
                                             "loc_context": 4}, False),
             # the other cells / wrappers of TorchRNNDecoder (decoder.py:18-110), recorded from the reference
             "att_decoder_gru": ("ctx", {"att_dim": 32}, False, {"rnn": "gru"}),
             "att_decoder_lstm_ln": ("loc", {"att_dim": 32, "conv_channels": 4, "loc_context": 5}, False,
                                     {"rnn": "lstm", "add_ln": True}),
             "att_decoder_lstmp": ("dot", {"att_dim": 32, "scaled": True}, True,
                                   {"rnn": "lstm", "proj_size": 24}),
             "att_decoder_onehot": ("ctx", {"att_dim": 32}, False, {"rnn": "lstm", "onehot_embed": True}),
             "att_decoder_tanh_ln": ("dot", {"att_dim": 32, "scaled": False}, True,
                                     {"rnn": "rnn_tanh", "add_ln": True}),
             "att_decoder_lstmp_ln": ("ctx", {"att_dim": 32}, False,
                                      {"rnn": "lstm", "add_ln": True, "proj_size": 24})}


@pytest.mark.parametrize("tag", sorted(ATT_CASES))
def test_att_decoder_golden(device, tag):
    from aps_amd.asr.base.attention import att_instance
    from aps_amd.asr.base.decoder import TorchRNNDecoder
    kind, att_kwargs, feeding = ATT_CASES[tag][:3]
    dec_kwargs = dict(ATT_CASES[tag][3]) if len(ATT_CASES[tag]) > 3 else {"rnn": "lstm"}
    g = golden(tag)
    dec_dim = dec_kwargs["proj_size"] if dec_kwargs.get("proj_size", -1) > 0 else 64
    att = att_instance(kind, 48, dec_dim, **att_kwargs)
    dec = TorchRNNDecoder(48, 30, num_layers=2, hidden=64, dropout=0.0, input_feeding=feeding,
                          **dec_kwargs)
    net = torch.nn.ModuleDict({"att_net": att, "decoder": dec})
    net.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd.")}, strict=True)
    net = net.eval().to(device)
    enc_out, tgt = g["enc_out"].to(device), g["tgt_pad"].to(device)
    att.clear()
    outs, alis = dec(att, enc_out, g["enc_len"].to(device), tgt)
    assert outs.shape == g["outs"].shape and alis.shape == g["alis"].shape
    assert_close(outs, g["outs"], TOL, tag + " outs")
    assert_close(alis, g["alis"], TOL, tag + " alis")
    att.clear()
    outs, alis = dec(att, enc_out, None, tgt)
    assert_close(outs, g["outs_full"], TOL, tag + " outs (no lengths)")
    assert_close(alis, g["alis_full"], TOL, tag + " alis (no lengths)")
    # scheduled sampling (decoder.py:196-200): the same `random` draws on both sides -> the same
    # positions feed back the arg-max of the previous prediction
    import random
    from oracle import att_oracle as ato
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    random.seed(5)
    want, _ = ato.rnn_att_decoder(sd, g["enc_out"], g["enc_len"], g["tgt_pad"], kind, 2,
                                  input_feeding=feeding, scaled=att_kwargs.get("scaled", True),
                                  loc_context=att_kwargs.get("loc_context", 0),
                                  heads=att_kwargs.get("att_head", 1), schedule_sampling=0.8,
                                  rnn=dec_kwargs["rnn"], add_ln=dec_kwargs.get("add_ln", False),
                                  onehot_embed=dec_kwargs.get("onehot_embed", False))
    att.clear()
    random.seed(5)
    outs, _ = dec(att, enc_out, g["enc_len"].to(device), tgt, schedule_sampling=0.8)
    assert_close(outs, want, TOL, tag + " outs (scheduled sampling)")
    assert rel_err(want, g["outs"]) > 1e-3  # the sampled run does differ from teacher forcing


GRAD_TAGS = {"ctx": "att_decoder_ctx", "dot": "att_decoder_dot", "loc": "att_decoder_loc",
             "mhctx": "att_decoder_mhctx", "mhdot": "att_decoder_mhdot", "mhloc": "att_decoder_mhloc",
             "gru": "att_decoder_gru", "lstm_ln": "att_decoder_lstm_ln", "lstmp": "att_decoder_lstmp",
             "onehot": "att_decoder_onehot", "tanh_ln": "att_decoder_tanh_ln", "lstmp_ln": "att_decoder_lstmp_ln"}


@pytest.mark.parametrize("case", sorted(GRAD_TAGS))
def test_att_decoder_trains_gradients_of_the_reference(device, case):
    """autograd THROUGH the RNN attention decoder (round 5; rounds 1-4 raised): what `cmd/train_am.py` does on
    an `att` recipe (aps/asr/base/decoder.py:165-218).  The cell steps on grad_ops.RnnCellFn (aps_rnn_step /
    aps_rnn_step_backward), every projection / LayerNorm on its HIP adjoint, the attention step's scores,
    softmax and context on torch's differentiable ops (attention.py: the documented training fall-through).
    Against gradients recorded from the reference's own modules (att_decoder_grads.npz: same parameters and
    inputs as the forward fixtures): the encoder output's and EVERY parameter's, each within 1e-4 of the largest
    gradient of the model (a gradient that is analytically zero -- the dot attention's key bias -- is rounding
    noise on both sides)."""
    from aps_amd.asr.base.attention import att_instance
    from aps_amd.asr.base.decoder import TorchRNNDecoder
    tag = GRAD_TAGS[case]
    kind, att_kwargs, feeding = ATT_CASES[tag][:3]
    dec_kwargs = dict(ATT_CASES[tag][3]) if len(ATT_CASES[tag]) > 3 else {"rnn": "lstm"}
    g, gg = golden(tag), golden("att_decoder_grads")
    dec_dim = dec_kwargs["proj_size"] if dec_kwargs.get("proj_size", -1) > 0 else 64
    att = att_instance(kind, 48, dec_dim, **att_kwargs)
    dec = TorchRNNDecoder(48, 30, num_layers=2, hidden=64, dropout=0.0, input_feeding=feeding, **dec_kwargs)
    net = torch.nn.ModuleDict({"att_net": att, "decoder": dec})
    net.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd.")}, strict=True)
    net = net.train().to(device)
    with torch.enable_grad():   # (this module runs under no_grad)
        enc_out = g["enc_out"].to(device).requires_grad_(True)
        att.clear()
        outs, alis = dec(att, enc_out, g["enc_len"].to(device), g["tgt_pad"].to(device))
        assert_close(outs, g["outs"], TOL, case + ": outs under autograd")
        assert_close(alis, g["alis"], TOL, case + ": alis under autograd")
        (outs * gg[f"{case}.up"].to(device)).sum().backward()
    att.clear()
    want = {k[len(case) + 3:]: v for k, v in gg.items() if k.startswith(case + ".g.")}
    scale = max(float(v.abs().max()) for v in want.values())
    got = {"enc_out": enc_out.grad}
    got.update({n: (torch.zeros_like(p) if p.grad is None else p.grad) for n, p in net.named_parameters()})
    assert set(got) == set(want), set(got) ^ set(want)
    worst = max((float((got[k].cpu() - want[k]).abs().max()) / scale, k) for k in want)
    print(f"[att decoder grads] {case}: {len(want)} tensors, worst {worst[0]:.2e} of the largest gradient ({worst[1]})")
    assert worst[0] <= 1e-4, worst
    # and tensor by tensor where the gradient is not (numerically) zero
    for k, v in want.items():
        if float(v.abs().max()) > 1e-3 * scale:
            assert rel_err(got[k], v) <= TOL, (k, rel_err(got[k], v))


def test_att_asr_forward(device):
    """asr@att at recipe widths (encoder projection 512, 3 x LSTM 512 decoder, location aware
    attention 512 / 10 channels / context 64) on an RNN encoder, against the oracle"""
    from aps_amd.libs import aps_asr_nnet
    from oracle import att_oracle as ao
    torch.manual_seed(95)
    net = aps_asr_nnet("asr@att")(
        40, 60, sos=58, eos=59, ctc=True, att_type="loc",
        att_kwargs={"att_dim": 512, "conv_channels": 10, "loc_context": 64},
        enc_type="pytorch_rnn", enc_proj=512,
        enc_kwargs=dict(rnn="lstm", num_layers=2, hidden=256, dropout=0, bidirectional=True),
        dec_dim=512, dec_kwargs=dict(num_layers=3, hidden=512, dropout=0)).eval()
    g = torch.Generator().manual_seed(96)
    x = torch.randn(4, 90, 40, generator=g)
    x_len = torch.tensor([90, 77, 60, 33])
    y = torch.randint(0, 58, (4, 12), generator=g)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    # encoder reference: torch's own CPU LSTM with packed-sequence semantics (zeros past len), then
    # the encoder's output projection (PyTorchRNNEncoder, aps/asr/base/encoder.py:143-184)
    import torch.nn.utils.rnn as R
    rnn = torch.nn.LSTM(40, 256, 2, batch_first=True, bidirectional=True)
    rnn.load_state_dict({k[len("encoder.impl."):]: v for k, v in sd.items()
                         if k.startswith("encoder.impl.")})
    with torch.no_grad():
        packed = R.pack_padded_sequence(x, x_len, batch_first=True, enforce_sorted=True)
        out, _ = R.pad_packed_sequence(rnn(packed)[0], batch_first=True)
        ref_enc = torch.nn.functional.linear(out, sd["encoder.outp.weight"], sd["encoder.outp.bias"])
    ref_len = x_len
    ref_ctc = torch.nn.functional.linear(ref_enc, sd["ctc.weight"], sd["ctc.bias"])
    ref, _ = ao.rnn_att_decoder(sd, ref_enc, ref_len, y, "loc", 3, loc_context=64)
    net = net.to(device)
    dec_out, enc_ctc, enc_len = net(x.to(device), x_len.to(device), y.to(device), None)
    assert enc_len.cpu().tolist() == ref_len.tolist()
    assert_close(enc_ctc, ref_ctc, TOL, "CTC branch")
    assert_close(dec_out, ref, TOL, "decoder output")


@pytest.mark.parametrize("tag,pre_norm", [("decoder_layer_memmask_post", False),
                                          ("decoder_layer_memmask_pre", True)])
def test_decoder_layer_memory_mask(device, tag, pre_norm):
    """TransformerDncoderLayer.forward with a memory_mask (decoder.py:51, 85), reference call
    convention, against the reference's own layer: a boolean band over the encoder frames together
    with both padding masks, and an additive float mask"""
    from aps_amd.asr.transformer.decoder import TransformerDncoderLayer
    g = golden(tag)
    layer = TransformerDncoderLayer(att_dim=64, nhead=2, feedforward_dim=128, pre_norm=pre_norm,
                                    att_dropout=0, ffn_dropout=0).eval()
    layer.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd.")})
    layer = layer.to(device)
    T, S = g["tgt"].shape[0], g["memory"].shape[0]
    tpad = (torch.arange(T)[None] >= g["tgt_len"][:, None]).to(device)
    mpad = (torch.arange(S)[None] >= g["mem_len"][:, None]).to(device)
    tgt, memory = g["tgt"].to(device), g["memory"].to(device)
    out = layer(tgt, memory, memory_mask=g["band"].bool().to(device), tgt_key_padding_mask=tpad,
                memory_key_padding_mask=mpad)
    # a padded target position whose band lies in the memory's padding is softmax over -inf only:
    # NaN in the reference (zeros here); the valid target positions carry the information
    valid = (torch.arange(T)[:, None] < g["tgt_len"][None, :])  # T x N
    assert_close(out.cpu()[valid], g["out_bool"][valid], TOL, tag + " boolean memory_mask")
    out = layer(tgt, memory, memory_mask=g["bias"].to(device))
    assert_close(out, g["out_float"], TOL, tag + " additive memory_mask")
    same = layer(tgt, memory, memory_mask=torch.zeros(T, S, device=device))
    assert_close(same, layer(tgt, memory), 1e-6, "a zero memory_mask changes nothing")
    # round 5: the attention adjoints carry additive mask TENSORS (rounds 1-4 raised here).  Under autograd the
    # layer gives the same output, and the gradients of sum(out * up) w.r.t. both inputs and every parameter are
    # the reference layer's own (recorded by make_golden.py)
    layer.train()
    bias = g["bias"].to(device)
    with torch.enable_grad():   # (this module runs under no_grad)
        tg, mg = tgt.clone().requires_grad_(True), memory.clone().requires_grad_(True)
        out_t = layer(tg, mg, memory_mask=bias)
        assert_close(out_t, g["out_float"], TOL, tag + " additive memory_mask under autograd")
        (out_t * g["up"].to(device)).sum().backward()
    assert_close(tg.grad, g["g_tgt"], TOL, tag + " g_tgt with an additive memory_mask")
    assert_close(mg.grad, g["g_memory"], TOL, tag + " g_memory with an additive memory_mask")
    for name, p_ in layer.named_parameters():
        assert_close(p_.grad, g["g." + name], TOL, f"{tag} g[{name}] with an additive memory_mask")
