"""
Pins oracle/encoder_oracle.py (the torch-CPU restatement of the transformer / conformer encoder)
against activations recorded from the reference's own modules (tests/golden/make_golden.py), so
the GPU parity tests that use it at sizes without a fixture stand on checked ground.
"""
import pytest
import torch

from oracle import encoder_oracle as eo
from tests.conftest import golden, assert_close

CASES = {
    # tag: (arch, pose, layers, heads, kwargs of generic_encoder)
    "encoder_xfmr_abs_post": ("xfmr", "abs", 2, 4, dict(pre_norm=False)),
    "encoder_xfmr_abs_pre": ("xfmr", "abs", 2, 4, dict(pre_norm=True)),
    "encoder_cfmr_rel": ("cfmr", "rel", 2, 4, dict(lradius=6, rradius=9, kernel_size=7,
                                                   pre_norm=True)),
    "encoder_cfmr_abs_plain": ("cfmr", "abs", 1, 2, dict(kernel_size=5, pre_norm=True,
                                                         macaron=False)),
    "encoder_cfmr_rel_post": ("cfmr", "rel", 1, 2, dict(lradius=5, rradius=3, kernel_size=5,
                                                        pre_norm=False)),
    "encoder_xfmr_rel_pre": ("xfmr", "rel", 1, 2, dict(lradius=5, rradius=3, pre_norm=True)),
    "encoder_xfmr_xl_ctx": ("xfmr", "xl", 2, 2, dict(proj="linear", window=(2, 2, 1))),
    "encoder_cfmr_xl_tie": ("cfmr", "xl", 2, 2, dict(proj="conv1d", kernel_size=5, pre_norm=True)),
    "encoder_xfmr_abs_lctx": ("xfmr", "abs", 1, 2, dict(window=(1, 3, 0))),
    # 100 encoder frames, 64-wide heads (make_golden.py:gen_conformer_t100)
    "encoder_cfmr_rel_t100": ("cfmr", "rel", 2, 2, dict(lradius=20, rradius=12, kernel_size=7,
                                                        pre_norm=True)),
    "encoder_xfmr_xl_t100": ("xfmr", "xl", 2, 2, dict()),
}


@pytest.mark.parametrize("tag", sorted(CASES))
def test_encoder_oracle_matches_reference(tag):
    arch, pose, layers, heads, kw = CASES[tag]
    g = golden(tag)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    with torch.no_grad():
        out, n = eo.generic_encoder(sd, g["x"], None, arch, pose, layers, heads, **kw)
        assert n is None
        assert_close(out, g["out_full"], 2e-6, tag + " full")
        out, n = eo.generic_encoder(sd, g["x"], g["lens"], arch, pose, layers, heads, **kw)
        assert torch.equal(n, g["num_frames"])
        ref = g["out_len"]
        if "window" in kw:
            # a padded QUERY whose whole context window is padding is softmax over -inf only: NaN in
            # the reference (and here); only the valid frames carry information
            valid = torch.arange(ref.shape[1])[None] < n[:, None]
            assert not torch.isnan(ref[valid]).any()
            out, ref = out[valid], ref[valid]
        assert_close(out, ref, 2e-6, tag + " ragged")


def test_specialised_entry_points_agree():
    g = golden("encoder_cfmr_rel")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    with torch.no_grad():
        a, _ = eo.cfmr_rel_encoder(sd, g["x"], g["lens"], 2, 4, 6, 9, kernel_size=7)
        b, _ = eo.generic_encoder(sd, g["x"], g["lens"], "cfmr", "rel", 2, 4, lradius=6, rradius=9,
                                  kernel_size=7, pre_norm=True)
    assert torch.equal(a, b)
    g = golden("encoder_xfmr_abs_post")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    with torch.no_grad():
        a, _ = eo.xfmr_abs_encoder(sd, g["x"], g["lens"], 2, 4)
        b, _ = eo.generic_encoder(sd, g["x"], g["lens"], "xfmr", "abs", 2, 4, pre_norm=False)
    assert torch.equal(a, b)


def test_joint_oracle_matches_reference():
    """EnhASRBase data path (STFT -> IPD features -> LSTM masks -> MVDR -> abs-mel-log-cmvn ->
    conformer -> CTC head) restated in oracle/joint_oracle.py vs the reference's activations"""
    from oracle import joint_oracle as jo
    g = golden("joint_mvdr_cfmr")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    for tag, lens in (("full", None), ("ragged", g["lens"])):
        with torch.no_grad():
            o = jo.joint_forward(sd, g["wav"], lens, num_mels=40, rnn_layers=2, enc_layers=2,
                                 nhead=2, lradius=4, rradius=4, kernel_size=5)
        assert_close(o["enh"][0], g[f"{tag}.enh_real"], 2e-5, tag + " enh real")
        assert_close(o["enh"][1], g[f"{tag}.enh_imag"], 2e-5, tag + " enh imag")
        assert_close(o["asr_feats"], g[f"{tag}.asr_feats"], 2e-5, tag + " asr feats")
        assert_close(o["enc_out"], g[f"{tag}.enc_out"], 1e-5, tag + " enc out")
        assert_close(o["enc_ctc"], g[f"{tag}.enc_ctc"], 1e-5, tag + " ctc")
        if lens is not None:
            assert torch.equal(o["num_frames"], g["ragged.num_frames"])
            assert torch.equal(o["enc_len"], g["ragged.enc_len"])


DCCRN_SMALL = dict(K="3,3;3,3;3,3", S="2,1;2,1;2,1", P="1,1,1", O="0,0,0", num_spks=2,
                   rnn_layers=2, frame_len=64, frame_hop=32, window="hann")


DCCRN_VARIANTS = [
    ("dccrn_shared", dict(share_decoder=True, non_linear="tanh")),
    ("dccrn_split", dict(share_decoder=False, non_linear="sigmoid")),
    ("dccrn_cat_causal", dict(share_decoder=True, non_linear="tanh", connection="cat",
                              causal_conv=True)),
    ("dccrn_real", dict(cplx=False, share_decoder=True, non_linear="sigmoid")),
    ("dccrn_real_cat", dict(cplx=False, share_decoder=False, non_linear="relu", connection="cat",
                            causal_conv=True))]


@pytest.mark.parametrize("tag,kw", DCCRN_VARIANTS)
def test_dccrn_oracle_matches_reference(tag, kw):
    from oracle import dccrn_oracle as do
    g = golden(tag)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    with torch.no_grad():
        wav = do.dccrn_forward(sd, g["mix"], mode="time", **DCCRN_SMALL, **kw)
        msk = do.dccrn_forward(sd, g["mix"], mode="freq", **DCCRN_SMALL, **kw)
    for s in range(2):
        assert_close(wav[s], g[f"wav{s}"], 1e-5, f"{tag} wav {s}")
        assert_close(msk[s], g[f"mask{s}"], 1e-5, f"{tag} mask {s}")
        # mask_predict returns the same masks as S x N x T x F (x 2)
        assert_close(msk[s].transpose(1, 2), g["pred"][s], 1e-5, f"{tag} mask_predict {s}")


DCCRN_TRAIN_CASES = [(tag, kw, "time") for tag, kw in DCCRN_VARIANTS] + \
    [(tag, kw, "freq") for tag, kw in DCCRN_VARIANTS if tag in ("dccrn_shared", "dccrn_real")]


def dccrn_train_reference(tag, kw, mode):
    """-> (weights sd, mix, fixture of the reference's train()-mode step)"""
    fwd, step = golden(tag), golden(f"{tag}_train_{mode}")
    return {k[3:]: v for k, v in fwd.items() if k.startswith("sd.")}, step["mix"], step


def assert_grad_close(got, ref, name, tol, what):
    """a gradient against the reference's, at the scale of the layer: a convolution bias in front
    of a batch-statistics BatchNorm has the exact gradient 0 (the mean subtraction removes it), what
    either side holds there is rounding noise, so a bias is judged at the scale of its layer's
    weight gradient"""
    want = ref["grad." + name].double()
    scale = want.abs().max()
    sibling = "grad." + name[:-4] + "weight"
    if name.endswith(".bias") and sibling in ref:
        scale = max(scale, ref[sibling].abs().max().double())
    err = ((got.detach().cpu().double() - want).abs().max() / scale.clamp_min(1e-30)).item()
    assert err <= tol, f"{what} grad {name}: scaled max error {err:.3e} > {tol:.1e}"


@pytest.mark.parametrize("tag,kw,mode", DCCRN_TRAIN_CASES)
def test_dccrn_oracle_train_mode_matches_reference(tag, kw, mode):
    """the oracle with train=True (batch-statistics BatchNorm) under torch autograd against the
    reference module's own train()-mode step: outputs, every parameter's gradient, the running
    statistics -- what the GPU gradient tests (tests/test_gpu_dccrn_train.py) then lean on"""
    from oracle import dccrn_oracle as do
    sd, mix, ref = dccrn_train_reference(tag, kw, mode)
    sd = {k: (v.clone().requires_grad_(True) if "running_" not in k else v.clone())
          for k, v in sd.items()}
    out = do.dccrn_forward(sd, mix, mode=mode, train=True, **DCCRN_SMALL, **kw)
    loss = sum((o * ref[f"probe{s}"]).sum() for s, o in enumerate(out))
    loss.backward()
    for s in range(2):
        assert_close(out[s], ref[f"out{s}"], 1e-5, f"{tag} train out {s}")
    names = [k[5:] for k in ref if k.startswith("grad.")]
    assert names
    assert ref["relu_margin"].item() >= 2e-6  # (no LeakyReLU input at the kink: make_golden.py)
    for k in names:
        assert_grad_close(sd[k].grad, ref, k, 5e-5, f"{tag} {mode}")
    for k in [k[5:] for k in ref if k.startswith("stat.")]:
        assert_close(sd[k], ref["stat." + k], 1e-5, f"{tag} {mode} running statistic {k}")


@pytest.mark.parametrize("tag,pre_norm", [("decoder_xfmr_post", False), ("decoder_xfmr_pre", True)])
def test_decoder_oracle_matches_reference(tag, pre_norm):
    from oracle import encoder_oracle as eo
    g = golden(tag)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    with torch.no_grad():
        out = eo.transformer_decoder(sd, g["enc_out"], g["enc_len"], g["tgt_pad"], g["tgt_len"], 2, 2,
                                     pre_norm=pre_norm)
        full = eo.transformer_decoder(sd, g["enc_out"], None, g["tgt_pad"], None, 2, 2,
                                      pre_norm=pre_norm)
        # step(pre_emb = first 4 tokens' embeddings, 2 more tokens, out_idx = -1) == position 5 of
        # the teacher-forced forward over the first 6 tokens
        six = eo.transformer_decoder(sd, g["enc_out"], None, g["tgt_pad"][:, :6], None, 2, 2,
                                     pre_norm=pre_norm)
    assert_close(out, g["out_len"], 1e-5, tag + " with lengths")
    assert_close(full, g["out_full"], 1e-5, tag + " without lengths")
    assert_close(six[:, -1], g["step_out"], 1e-5, tag + " step")


def test_causal_conformer_layer_oracle():
    g = golden("cfmr_layer_causal")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    pad = torch.arange(21)[None, :] >= g["lens"][:, None]
    with torch.no_grad():
        out = eo.conformer_layer(sd, "", g["src"], pad, 2, None, kernel_size=5, pre_norm=True,
                                 casual_conv1d=True)
    assert_close(out, g["out"], 2e-6, "causal conformer layer")


ATT_CASES = {"att_decoder_ctx": dict(kind="ctx", input_feeding=False),
             "att_decoder_dot": dict(kind="dot", input_feeding=True, scaled=True),
             "att_decoder_loc": dict(kind="loc", input_feeding=False, loc_context=5),
             "att_decoder_mhctx": dict(kind="mhctx", input_feeding=False, heads=3),
             "att_decoder_mhdot": dict(kind="mhdot", input_feeding=True, scaled=True, heads=4),
             "att_decoder_mhloc": dict(kind="mhloc", input_feeding=False, loc_context=4, heads=2),
             # the other cells / wrappers of TorchRNNDecoder (decoder.py:18-110)
             "att_decoder_gru": dict(kind="ctx", input_feeding=False, rnn="gru"),
             "att_decoder_lstm_ln": dict(kind="loc", input_feeding=False, loc_context=5, add_ln=True),
             "att_decoder_lstmp": dict(kind="dot", input_feeding=True, scaled=True),
             "att_decoder_onehot": dict(kind="ctx", input_feeding=False, onehot_embed=True),
             "att_decoder_tanh_ln": dict(kind="dot", input_feeding=True, scaled=False, rnn="rnn_tanh",
                                         add_ln=True),
             "att_decoder_lstmp_ln": dict(kind="ctx", input_feeding=False, add_ln=True)}


@pytest.mark.parametrize("tag", sorted(ATT_CASES))
def test_att_decoder_oracle_matches_reference(tag):
    from oracle import att_oracle as ao
    g = golden(tag)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    with torch.no_grad():
        outs, alis = ao.rnn_att_decoder(sd, g["enc_out"], g["enc_len"], g["tgt_pad"], num_layers=2,
                                        **ATT_CASES[tag])
        outs_f, alis_f = ao.rnn_att_decoder(sd, g["enc_out"], None, g["tgt_pad"], num_layers=2,
                                            **ATT_CASES[tag])
    assert_close(outs, g["outs"], 1e-5, tag + " outs")
    assert_close(alis, g["alis"], 1e-5, tag + " alis")
    assert_close(outs_f, g["outs_full"], 1e-5, tag + " outs (no lengths)")
    assert_close(alis_f, g["alis_full"], 1e-5, tag + " alis (no lengths)")


@pytest.mark.parametrize("tag,pre_norm", [("decoder_layer_memmask_post", False),
                                          ("decoder_layer_memmask_pre", True)])
def test_decoder_layer_memory_mask_oracle_matches_reference(tag, pre_norm):
    """the decoder layer called with a memory_mask (boolean band + padding masks, additive float)
    against the reference's own layer"""
    from oracle import encoder_oracle as eo
    g = golden(tag)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    T, S = g["tgt"].shape[0], g["memory"].shape[0]
    sub = torch.zeros(T, T).masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool), 1), float("-inf"))
    tpad = torch.arange(T)[None] >= g["tgt_len"][:, None]
    mpad = torch.arange(S)[None] >= g["mem_len"][:, None]
    with torch.no_grad():
        out = eo.decoder_layer(sd, "", g["tgt"], g["memory"], sub, tpad, mpad, 2, pre_norm,
                               memory_mask=g["band"].bool())
        # a padded target position whose band lies in the memory's padding is softmax over -inf
        # only: NaN in the reference; the valid target positions carry the information
        valid = (torch.arange(T)[:, None] < g["tgt_len"][None, :])  # T x N
        assert not torch.isnan(g["out_bool"][valid]).any() and torch.isnan(g["out_bool"]).any()
        assert_close(out[valid], g["out_bool"][valid], 2e-6, tag + " boolean memory_mask")
        out = eo.decoder_layer(sd, "", g["tgt"], g["memory"], sub, None, None, 2, pre_norm,
                               memory_mask=g["bias"])
        assert_close(out, g["out_float"], 2e-6, tag + " additive memory_mask")


# ------------------------------------------------------------------------------------------------
# gradients: the oracle under torch autograd against the gradients of the reference's own modules
# (tests/golden/make_golden.py:gen_train_grads) -- what the GPU gradient tests of the transformer
# decoder, the Transformer-XL / windowed encoders and the causal conformer convolution lean on
# ------------------------------------------------------------------------------------------------
def _leaves(g):
    return {k[3:]: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k
                    and "div_term" not in k else v.clone())
            for k, v in g.items() if k.startswith("sd.")}


def _check_grads(sd, ref, what, tol=2e-5):
    names = [k[5:] for k in ref if k.startswith("grad.")]
    assert names
    for k in names:
        assert sd[k].grad is not None, f"{what}: no gradient for {k}"
        assert_grad_close(sd[k].grad, ref, k, tol, what)


@pytest.mark.parametrize("tag,pre_norm", [("decoder_xfmr_post", False), ("decoder_xfmr_pre", True)])
def test_decoder_oracle_gradients_match_reference(tag, pre_norm):
    g, ref = golden(tag), golden(tag + "_grad")
    sd = _leaves(g)
    enc_out = g["enc_out"].clone().requires_grad_(True)
    out = eo.transformer_decoder(sd, enc_out, g["enc_len"], g["tgt_pad"], g["tgt_len"], 2, 2,
                                 pre_norm=pre_norm)
    valid = (torch.arange(out.shape[1])[None] < g["tgt_len"][:, None])[..., None]
    loss = (torch.where(valid, out, torch.zeros_like(out)) * ref["probe"]).sum()
    loss.backward()
    assert_close(loss.detach(), ref["loss"], 1e-5, tag + " loss")
    assert_close(enc_out.grad, ref["g_enc_out"], 2e-5, tag + " g_enc_out")
    _check_grads(sd, ref, tag)


@pytest.mark.parametrize("tag", ["encoder_xfmr_xl_ctx", "encoder_xfmr_abs_lctx"])
def test_windowed_encoder_oracle_gradients_match_reference(tag):
    g, ref = golden(tag), golden(tag + "_grad")
    arch, pose, layers, heads, kw = CASES[tag]
    sd = _leaves(g)
    x = g["x"].clone().requires_grad_(True)
    out, _ = eo.generic_encoder(sd, x, None, arch, pose, layers, heads, **kw)
    loss = (out * ref["probe"]).sum()
    loss.backward()
    assert_close(loss.detach(), ref["loss"], 1e-5, tag + " loss")
    assert_close(x.grad, ref["g_x"], 2e-5, tag + " g_x")
    _check_grads(sd, ref, tag)


def test_causal_conformer_layer_oracle_gradients_match_reference():
    g, ref = golden("cfmr_layer_causal"), golden("cfmr_layer_causal_grad")
    sd = _leaves(g)
    src = g["src"].clone().requires_grad_(True)
    pad = torch.arange(21)[None, :] >= g["lens"][:, None]
    out = eo.conformer_layer(sd, "", src, pad, 2, None, kernel_size=5, pre_norm=True, casual_conv1d=True)
    valid = (~pad).transpose(0, 1)[..., None]
    loss = (torch.where(valid, out, torch.zeros_like(out)) * ref["probe"]).sum()
    loss.backward()
    assert_close(loss.detach(), ref["loss"], 1e-5, "causal conformer layer loss")
    assert_close(src.grad, ref["g_src"], 2e-5, "causal conformer layer g_src")
    _check_grads(sd, ref, "causal conformer layer")


def test_joint_oracle_gradients_match_reference():
    """the north-star model under autograd: oracle/joint_oracle.py against the gradients of the
    reference's own EnhTransform -> RNNMaskMvdr -> AsrTransform -> CtcASR modules (ragged lengths) --
    what tests/test_gpu_train.py::test_joint_backward_vs_oracle leans on"""
    from oracle import joint_oracle as jo
    g, ref = golden("joint_mvdr_cfmr"), golden("joint_mvdr_cfmr_grad")
    names = [k[5:] for k in ref if k.startswith("grad.")]
    sd = {k[3:]: (v.clone().requires_grad_(True) if k[3:] in names else v.clone())
          for k, v in g.items() if k.startswith("sd.")}
    o = jo.joint_forward(sd, g["wav"], g["lens"], num_mels=40, rnn_layers=2, enc_layers=2, nhead=2,
                         lradius=4, rradius=4, kernel_size=5)
    valid = (torch.arange(o["enc_out"].shape[1])[None] < o["enc_len"][:, None])[..., None]
    loss = (torch.where(valid, o["enc_out"], torch.zeros_like(o["enc_out"])) * ref["probe_out"]).sum() + \
        (torch.where(valid, o["enc_ctc"], torch.zeros_like(o["enc_ctc"])) * ref["probe_ctc"]).sum()
    loss.backward()
    assert_close(loss.detach(), ref["loss"], 2e-5, "joint loss")
    assert len(names) >= 60
    worst = 0.0
    for k in names:
        assert sd[k].grad is not None, f"no gradient for {k}"
        if k.endswith("gvec.bias"):  # (a shift of a softmax input: exact gradient zero on both sides)
            continue
        want = ref["grad." + k].double()
        err = ((sd[k].grad.double() - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()
        worst = max(worst, err)
        assert err <= 2e-4, f"joint grad {k}: scaled max error {err:.3e}"
    print(f"[oracle] joint gradients vs the reference's: worst {worst:.2e} over {len(names)} tensors")


def test_xfmr_asr_oracle_matches_reference_forward_and_gradients():
    """asr@xfmr end to end (features -> conv2d projection -> transformer encoder + CTC branch ->
    transformer decoder) restated from the oracle's parts against the reference's own model: outputs and
    the gradient of every parameter (what tests/test_gpu_decoder.py::test_xfmr_asr_forward and
    tests/test_gpu_train.py::test_xfmr_asr_backward_vs_oracle lean on)"""
    import torch.nn.functional as F
    from oracle import aps_oracle as orc
    g = golden("xfmr_asr")
    names = [k[5:] for k in g if k.startswith("grad.")]
    sd = {k[3:]: (v.clone().requires_grad_(True) if k[3:] in names else v.clone())
          for k, v in g.items() if k.startswith("sd.")}
    feats = orc.asr_features(g["wav"], "fbank-log-cmvn", frame_len=400, frame_hop=160,
                             window_name="hamm", num_mels=40)
    n = torch.tensor([orc.num_frames(int(v), 512, 160, False) for v in g["wav_len"]])
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    enc_out, enc_len = eo.generic_encoder(enc, feats, n, "xfmr", "abs", 2, 2)
    dec_out = eo.transformer_decoder(sd, enc_out, enc_len, g["y"], g["y_len"], 2, 2, prefix="decoder.")
    enc_ctc = F.linear(enc_out, sd["ctc.weight"], sd["ctc.bias"])
    assert torch.equal(enc_len, g["enc_len"])
    vd = (torch.arange(dec_out.shape[1])[None] < g["y_len"][:, None])[..., None]
    ve = (torch.arange(enc_ctc.shape[1])[None] < enc_len[:, None])[..., None]
    assert_close(dec_out.detach() * vd, g["dec_out"] * vd, 1e-5, "asr@xfmr decoder output")
    assert_close(enc_ctc.detach() * ve, g["enc_ctc"] * ve, 1e-5, "asr@xfmr CTC branch")
    loss = (torch.where(vd, dec_out, torch.zeros_like(dec_out)) * g["probe_dec"]).sum() + \
        (torch.where(ve, enc_ctc, torch.zeros_like(enc_ctc)) * g["probe_ctc"]).sum()
    loss.backward()
    assert_close(loss.detach(), g["loss"], 2e-5, "asr@xfmr loss")
    assert len(names) >= 70
    worst = 0.0
    for k in names:
        assert sd[k].grad is not None, f"no gradient for {k}"
        want = g["grad." + k].double()
        err = ((sd[k].grad.double() - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()
        worst = max(worst, err)
        assert err <= 1e-4, f"asr@xfmr grad {k}: scaled max error {err:.3e}"
    print(f"[oracle] asr@xfmr gradients vs the reference's: worst {worst:.2e} over {len(names)} tensors")
