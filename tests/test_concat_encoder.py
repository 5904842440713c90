"""
encoder_instance("concat", ...) (aps/asr/base/encoder.py:21-72): the conv1d + BLSTM / conv2d + LSTM
encoders of the reference's LAS recipes.  CPU: the chain is built stage by stage with the
reference's widths and its state dict loads strictly; GPU: forward with and without lengths against
activations recorded from the reference (tests/golden/concat_*.npz, make_golden.py
gen_concat_encoder).
"""
import pytest
import torch

from tests.conftest import assert_close, golden

TAGS = ["concat_conv1d_blstm", "concat_conv2d_lstm"]


def build(g):
    from aps_amd.asr.ctc import encoder_instance
    enc = encoder_instance("concat", 40, 56, g.cfg)
    enc.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd.")}, strict=False)
    return enc.eval()


@pytest.mark.parametrize("tag", TAGS)
def test_concat_encoder_builds_like_the_reference(tag):
    from aps_amd.asr.ctc import ConcatEncoder, encoder_instance
    g = golden(tag)
    enc = build(g)
    assert isinstance(enc, ConcatEncoder) and len(enc) == 2
    want = {k[3:]: tuple(v.shape) for k, v in g.items() if k.startswith("sd.")}
    mine = {k: tuple(v.shape) for k, v in enc.state_dict().items() if "num_batches" not in k}
    assert mine == want
    assert enc[-1].out_features == 56 and enc[1].inp_features == enc[0].out_features
    with pytest.raises(ValueError):
        encoder_instance("concat", 40, 56, {"pytorch_rnn": {}})
    with pytest.raises(RuntimeError):
        encoder_instance("concat", 40, 56, {"conv1d": {}, "nope": {}})


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
def test_concat_encoder_forward(device, tag):
    g = golden(tag)
    enc = build(g).to(device)
    with torch.no_grad():
        out, out_len = enc(g["x"].to(device), g["lens"].to(device))
        out_full, none_len = enc(g["x"].to(device), None)
    assert none_len is None and out_len.cpu().tolist() == g["out_len"].tolist()
    assert out.shape == g["out"].shape
    assert_close(out, g["out"], 1e-4, "with lengths")
    assert_close(out_full, g["out_full"], 1e-4, "without lengths")
