"""
A checkpoint directory of the reference (train.yaml + best.pt.tar) loads into this build through
aps_amd/eval/wrapper.py (aps/eval/wrapper.py:16-86) for the registered joint nets asr@enh_xfmr and
asr@enh_att: CPU -- the recipe builds, the reference's state dict loads strictly; GPU -- the
teacher-forced forward reproduces the activations the reference recorded for the same checkpoint
(tests/golden/checkpoint_*.npz, make_golden.py gen_checkpoints).
"""
import os

import pytest
import torch
import yaml

from tests.conftest import assert_close, golden

TAGS = {"checkpoint_enh_xfmr": "EnhXfmrASR", "checkpoint_enh_att": "EnhAttASR"}


def write_checkpoint(g, folder):
    with open(os.path.join(folder, "train.yaml"), "w") as f:
        yaml.safe_dump(g.cfg, f)
    state = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    torch.save({"model_state": state, "epoch": int(g["epoch"])}, os.path.join(folder, "best.pt.tar"))
    return state


@pytest.mark.parametrize("tag", sorted(TAGS))
def test_reference_checkpoint_loads(tag, tmp_path):
    from aps_amd.eval.wrapper import NnetEvaluator, load_checkpoint
    g = golden(tag)
    state = write_checkpoint(g, str(tmp_path))
    stats = load_checkpoint(str(tmp_path))
    assert type(stats["nnet"]).__name__ == TAGS[tag]
    assert stats["epoch"] == int(g["epoch"]) and stats["accept_raw"] == bool(int(g["accept_raw"]))
    assert stats["conf"] == g.cfg
    mine = stats["nnet"].state_dict()
    assert {k: tuple(v.shape) for k, v in mine.items()} == {k: tuple(v.shape) for k, v in state.items()}
    for k, v in state.items():
        assert torch.equal(mine[k], v), k
    ev = NnetEvaluator(str(tmp_path))
    assert not ev.nnet.training and ev.device.type == "cpu" and ev.epoch == int(g["epoch"])
    with pytest.raises(NotImplementedError):
        ev.run()
    with pytest.raises(FileNotFoundError):
        load_checkpoint(str(tmp_path), cpt_tag="last")


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(TAGS))
def test_reference_checkpoint_forward(device, tag, tmp_path):
    from aps_amd.eval.wrapper import NnetEvaluator
    g = golden(tag)
    write_checkpoint(g, str(tmp_path))
    ev = NnetEvaluator(str(tmp_path), device_id=0)
    with torch.no_grad():
        dec_out, enc_ctc, enc_len = ev.nnet(g["wav"].to(device), g["lens"].to(device),
                                            g["tgt"].to(device), g["tgt_len"].to(device))
    assert enc_len.cpu().tolist() == g["enc_len"].tolist()
    assert dec_out.shape == g["dec_out"].shape and enc_ctc.shape == g["enc_ctc"].shape
    assert_close(enc_ctc, g["enc_ctc"], 1e-4, "encoder / CTC branch")
    assert_close(dec_out, g["dec_out"], 1e-4, "decoder output")
