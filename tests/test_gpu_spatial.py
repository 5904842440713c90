"""
GPU parity of the geometry-dependent enh layers (aps_amd/transform/spatial.py on
csrc/spatial.hip): FixedBeamformer and DfTransform against activations recorded from the
reference layers, and against the CPU oracle at the sizes of the reference's own tests
(tests/python/test_transform.py:153-186).  Tolerance 1e-4 of the output scale.
"""
import math

import pytest
import torch

from oracle import aps_oracle as orc
from tests.conftest import assert_close, golden

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def test_fixed_beamformer_golden(device):
    from aps_amd.transform import FixedBeamformer
    g = golden("fixed_beamformer")
    bf = FixedBeamformer(5, 4, 33)
    bf.load_state_dict({"real": g["w_real"], "imag": g["w_imag"]}, strict=True)
    bf = bf.to(device)
    xr, xi = g["xr"].to(device), g["xi"].to(device)
    r, i = bf(xr, xi)
    assert_close(r, g["all_r"], TOL, "all beams re")
    assert_close(i, g["all_i"], TOL, "all beams im")
    r, i = bf(xr, xi, beam=2)
    assert_close(r, g["one_r"], TOL, "beam 2 re")
    assert_close(i, g["one_i"], TOL, "beam 2 im")
    r, i = bf(xr, xi, beam=g["beams"].to(device))
    assert_close(r, g["sel_r"], TOL, "per-utterance beams re")
    assert_close(i, g["sel_i"], TOL, "per-utterance beams im")
    r, i = bf(xr, xi, beam=0, trans=True)
    assert_close(r, g["tr_r"], TOL, "trans re")
    assert_close(i, g["tr_i"], TOL, "trans im")
    with pytest.raises(RuntimeError):
        bf(xr[:, :3], xi[:, :3])
    with pytest.raises(IndexError):
        bf(xr, xi, beam=5)


@pytest.mark.parametrize("num_channels", [4, 8, 11])
@pytest.mark.parametrize("num_bins", [257, 513])
@pytest.mark.parametrize("num_directions", [8, 16])
def test_fixed_beamformer_oracle(device, num_channels, num_bins, num_directions):
    from aps_amd.transform import FixedBeamformer
    torch.manual_seed(num_channels * num_bins + num_directions)
    bf = FixedBeamformer(num_directions, num_channels, num_bins)
    frames = int(torch.randint(50, 300, (1,)))
    xr, xi = torch.rand(4, num_channels, num_bins, frames), torch.rand(4, num_channels, num_bins, frames)
    wr, wi = bf.real[..., 0].detach(), bf.imag[..., 0].detach()
    bf = bf.to(device)
    r, i = bf(xr.to(device), xi.to(device))
    assert r.shape == (4, num_directions, num_bins, frames)
    want_r, want_i = orc.fixed_beamform(xr, xi, wr, wi)
    assert_close(r, want_r, TOL, "re")
    assert_close(i, want_i, TOL, "im")
    r, i = bf(xr.to(device), xi.to(device), beam=0)
    assert r.shape == (4, num_bins, frames)
    want_r, want_i = orc.fixed_beamform(xr, xi, wr, wi, beam=0)
    assert_close(r, want_r, TOL, "beam 0 re")
    assert_close(i, want_i, TOL, "beam 0 im")


def test_df_transform_golden(device):
    from aps_amd.transform import DfTransform
    g = golden("df_transform")
    p = g["phase"].to(device)
    doa_a, doa_b = g["doa_a"].to(device), g["doa_b"].to(device)
    known = DfTransform(num_bins=33, num_doas=1)
    known.load_state_dict({"omega": g["omega"]}, strict=True)
    known = known.to(device)
    assert_close(known(p, doa_a), g["af_known"], TOL, "known DoA")
    assert_close(known(p[0], doa_a[:1]), g["af_single"], TOL, "3-D phase")
    pairs = DfTransform(num_bins=33, num_doas=1, af_index="1,4;2,5;3,6;0,2").to(device)
    af = pairs(p, [doa_a, doa_b])
    assert af.shape == (3, 13, 66)
    assert_close(af, g["af_two"], TOL, "two speakers")
    sampled = DfTransform(num_bins=33, num_doas=8, sr=8000, velocity=343).to(device)
    af = sampled(p, doa_a)
    assert af.shape == (3, 8, 13, 33)
    assert_close(af, g["af_sampled"], TOL, "sampled DoAs")
    with pytest.raises(RuntimeError):
        sampled(p, [doa_a, doa_b])
    with pytest.raises(RuntimeError):
        DfTransform(geometric="4@")


@pytest.mark.parametrize("num_bins", [257, 513])
@pytest.mark.parametrize("num_doas", [1, 8])
def test_df_transform_oracle(device, num_bins, num_doas):
    from aps_amd.transform import DfTransform
    torch.manual_seed(num_bins + num_doas)
    layer = DfTransform(num_bins=num_bins, num_doas=num_doas).to(device)
    frames = int(torch.randint(50, 100, (1,)))
    phase = (torch.rand(4, 7, frames, num_bins) * 2 - 1) * math.pi
    doa = torch.rand(4) * 2 * math.pi
    af = layer(phase.to(device), doa.to(device))
    want = orc.directional_feature(phase, doa, layer.index_l, layer.index_r, num_doas=num_doas)
    assert af.shape == want.shape
    assert_close(af, want, TOL, "angle feature")


@pytest.mark.parametrize("select", [None, 1, "per_utt"])
def test_fixed_beamformer_trains(device, select):
    """FixedBeamformer(requires_grad=True) (aps/transform/enh.py:303-384): the module under autograd -- outputs,
    the input's gradient and the coefficients' against autograd through the oracle, for all beams, one beam and
    one beam per utterance"""
    import oracle.aps_oracle as orc
    from aps_amd.transform.spatial import FixedBeamformer
    torch.manual_seed(9)
    N, C, F, T, B = 3, 4, 33, 20, 6
    fb = FixedBeamformer(B, C, F, requires_grad=True)
    g = torch.Generator().manual_seed(4)
    xr, xi = torch.randn(N, C, F, T, generator=g), torch.randn(N, C, F, T, generator=g)
    beam = None if select is None else (1 if select == 1 else torch.tensor([5, 0, 2]))
    wr = fb.real.detach().squeeze(-1).clone().requires_grad_(True)
    wi = fb.imag.detach().squeeze(-1).clone().requires_grad_(True)
    xr_r, xi_r = xr.clone().requires_grad_(True), xi.clone().requires_grad_(True)
    with torch.enable_grad():
        br, bi = orc.fixed_beamform(xr_r, xi_r, wr, wi, beam)
        ur, ui = torch.randn(br.shape, generator=g), torch.randn(bi.shape, generator=g)
        ((br * ur).sum() + (bi * ui).sum()).backward()
    fb = fb.to(device)
    xd_r, xd_i = xr.to(device).requires_grad_(True), xi.to(device).requires_grad_(True)
    dbeam = beam.to(device) if isinstance(beam, torch.Tensor) else beam
    with torch.enable_grad():   # (this module runs under no_grad)
        yr, yi = fb(xd_r, xd_i, beam=dbeam)
        assert_close(yr, br, 1e-5, "beam real")
        assert_close(yi, bi, 1e-5, "beam imag")
        ((yr * ur.to(device)).sum() + (yi * ui.to(device)).sum()).backward()
    assert_close(xd_r.grad, xr_r.grad, 1e-5, "g_x real")
    assert_close(xd_i.grad, xi_r.grad, 1e-5, "g_x imag")
    assert_close(fb.real.grad.squeeze(-1), wr.grad, 1e-5, "g_w real")
    assert_close(fb.imag.grad.squeeze(-1), wi.grad, 1e-5, "g_w imag")
