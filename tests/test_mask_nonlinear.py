"""
MaskNonLinear (aps/sse/base.py:112-156): the oracle's restatement (CPU) and the
aps_mask_nonlinear kernel (GPU) against outputs recorded from the reference layer
(tests/golden/mask_nonlinear.npz, make_golden.py gen_mask_nonlinear).
"""
import pytest
import torch

from oracle import aps_oracle as orc
from tests.conftest import assert_close, golden

CASES = {"relu_scaled": ("relu", dict(scale=2.0, vmax=3.0)),
         "sigmoid": ("sigmoid", dict()),
         "softplus": ("softplus", dict(vmax=10.0)),
         "tanh_clamped": ("tanh", dict(scale=1.5, vmax=1.2, vmin=-0.5)),
         "softmax": ("softmax", dict(scale=1.0, vmin=0.05)),
         "none": ("none", dict(vmax=2.0, vmin=-2.0))}


@pytest.mark.parametrize("tag", sorted(CASES))
def test_mask_nonlinear_oracle(tag):
    g = golden("mask_nonlinear")
    name, kw = CASES[tag]
    for key in ("3", "4"):
        assert_close(orc.mask_nonlinear(g["x" + key], name, **kw), g[f"{tag}.y{key}"], 1e-6, tag)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", sorted(CASES))
def test_mask_nonlinear_kernel(device, tag):
    from aps_amd.sse.base import MaskNonLinear
    g = golden("mask_nonlinear")
    name, kw = CASES[tag]
    layer = MaskNonLinear(name, enable="all", **kw)
    with torch.no_grad():
        for key in ("3", "4"):
            y = layer(g["x" + key].to(device))
            assert y.shape == g[f"{tag}.y{key}"].shape
            assert_close(y, g[f"{tag}.y{key}"], 1e-5, f"{tag} {key}-D")
        with pytest.raises(RuntimeError):
            layer(torch.randn(4, 4, device=device))
