"""
VariantRNNEncoder ("variant_rnn", aps/asr/base/encoder.py:225-308 with VariantRNN,
component.py:389-449): projection + BatchNorm + tanh between BLSTM layers, the pyramidal stack with
GroupNorm and summed directions, a plain unidirectional stack.  CPU: built from the recipe, the
reference's state dict loads strictly; GPU: forward with / without lengths against activations
recorded from the reference (tests/golden/variant_rnn_*.npz, make_golden.py gen_variant_rnn).
"""
import pytest
import torch

from tests.conftest import assert_close, golden

TAGS = ["variant_rnn_bn", "variant_rnn_plain", "variant_rnn_pyramid"]


def build(g):
    from aps_amd.asr.ctc import encoder_instance
    enc = encoder_instance("variant_rnn", 40, 56, g.cfg)
    missing, unexpected = enc.load_state_dict(
        {k[3:]: v for k, v in g.items() if k.startswith("sd.")}, strict=False)
    assert not unexpected and all("num_batches" in k for k in missing)
    return enc.eval()


@pytest.mark.parametrize("tag", TAGS)
def test_variant_rnn_builds_like_the_reference(tag):
    g = golden(tag)
    enc = build(g)
    want = {k[3:]: tuple(v.shape) for k, v in g.items() if k.startswith("sd.")}
    mine = {k: tuple(v.shape) for k, v in enc.state_dict().items() if "num_batches" not in k}
    assert mine == want
    assert enc.out_features == 56


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_variant_rnn_forward(device, tag):
    g = golden(tag)
    enc = build(g).to(device)
    with torch.no_grad():
        out, out_len = enc(g["x"].to(device), g["lens"].to(device))
        out_full, _ = enc(g["x"].to(device), None)
    assert out_len.cpu().tolist() == g["out_len"].tolist()
    assert out.shape == g["out"].shape and out_full.shape == g["out_full"].shape
    assert_close(out, g["out"], 1e-4, "with lengths")
    assert_close(out_full, g["out_full"], 1e-4, "without lengths")
