"""
CPU: the oracle's restatement of the training-mode augmentation layers (speed perturbation,
SpecAugment; oracle/aps_oracle.py) against seeded runs of the reference layers
(tests/golden/speed_perturb_train.npz, spec_augment_train.npz; make_golden.py gen_augment_train).
The random draws are part of the algorithm: the oracle consumes the generators like the reference.
"""
import random

import pytest
import torch

from oracle import aps_oracle as orc
from tests.conftest import assert_close, golden

AUG = {"zero": dict(max_frame=12, num_time_masks=2, max_bands=8, num_freq_masks=2),
       "mean": dict(max_frame=40, num_time_masks=1, max_bands=30, num_freq_masks=1),
       "adaptive": dict(pm=0.04, ps=0.1, max_frame=40, num_time_masks=4, max_bands=10,
                        num_freq_masks=1),
       "coin": dict(max_frame=12, num_time_masks=1, max_bands=8, num_freq_masks=1)}
PROB = {"zero": 1.0, "mean": 1.0, "adaptive": 1.0, "coin": 0.5}


def test_speed_perturb_oracle():
    from aps_amd.transform.utils import speed_perturb_filter
    g = golden("speed_perturb_train")
    weights = [speed_perturb_filter(16000, 14400), speed_perturb_filter(16000, 17600)]
    torch.manual_seed(int(g["seed"]))
    choice = torch.randint(0, 3, (g["wav"].shape[0],))
    assert choice.tolist() == g["choice"].tolist()
    out = orc.speed_perturb(g["wav"], weights, choice)
    assert out.shape == g["out"].shape
    assert_close(out, g["out"], 2e-6, "resampled batch")
    src, dst = torch.tensor([10, 10, 1]), torch.tensor([9, 11, 1])
    assert (g["lens"] // src[choice] * dst[choice]).tolist() == g["out_len"].tolist()


@pytest.mark.parametrize("tag", sorted(AUG))
def test_spec_augment_oracle(tag):
    g = golden("spec_augment_train")
    x = g[f"{tag}.x"]
    seed = int(g[f"{tag}.seed"])
    torch.manual_seed(seed)
    random.seed(seed)
    for key in ("y", "y2"):
        if torch.rand(1).item() < PROB[tag]:
            bands = orc.tf_bands(x.shape[0], x.shape[-2], x.shape[-1], **AUG[tag])
            y = orc.spec_augment(x, bands, mask_zero=(tag != "mean"))
        else:
            y = x
        assert_close(y, g[f"{tag}.{key}"], 1e-6, f"{tag} {key}")
        if tag != "mean":
            assert torch.equal(y == 0, g[f"{tag}.{key}"] == 0)
