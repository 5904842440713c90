"""
aps_conformer_stack (csrc/conformer_mega.hip): the conformer encoder stack as one launch per batch, a workgroup per
utterance, against its per-launch twin (the oracle-checked path of aps_amd/asr/transformer/impl.py) and, through the
joint model, against the CPU oracle.  Reference: aps/asr/transformer/impl.py:432-541, 718-756.
"""
import pytest
import torch

from tests.conftest import assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def _encoder(layers, seed=3):
    from aps_amd.asr.transformer.impl import get_xfmr_encoder
    torch.manual_seed(seed)
    enc = get_xfmr_encoder("cfmr", "rel", layers, {"att_dim": 512, "nhead": 8, "feedforward_dim": 1024,
                                                  "att_dropout": 0, "ffn_dropout": 0, "kernel_size": 15})
    # non-trivial LayerNorm / BatchNorm parameters and running statistics: the folds must carry them
    for m in enc.modules():
        if isinstance(m, torch.nn.LayerNorm):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.3, 0.3)
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.2, 0.2)
    return enc.eval()


@pytest.mark.parametrize("N,T,layers", [(3, 63, 1), (5, 40, 2), (32, 63, 3), (2, 64, 1), (1, 7, 1)])
def test_conformer_stack_equals_the_per_launch_path(device, N, T, layers):
    """one launch per batch against one launch per projection, same derived weights: within 2e-6 of the output's scale
    (the FFN's second projection is two K = 512 phases chained through the residual: a different summation order),
    ragged lengths included"""
    from aps_amd import mega
    enc = _encoder(layers).to(device)
    g = torch.Generator().manual_seed(N * 100 + T)
    x = torch.randn(N, T, 512, generator=g).to(device)
    rel = (0.2 * torch.randn(2 * T - 1, 64, generator=g)).to(device)
    lens = torch.randint(max(1, T // 2), T + 1, (N,), generator=g)
    lens[0] = T
    lens_d = lens.to(device)
    saved = mega.ENABLED
    try:
        mega.ENABLED = False
        want = enc.run(x, lens_d, rel=rel)
        want_full = enc.run(x, None, rel=rel)
        mega.ENABLED = True
        assert mega.supported(enc, x, rel, None)
        got = enc.run(x, lens_d, rel=rel)
        got_full = enc.run(x, None, rel=rel)
    finally:
        mega.ENABLED = saved
    assert torch.isfinite(got).all()
    for n in range(N):   # (frames beyond an utterance's length see masked keys only: compared too, the two paths agree there)
        assert_close(got[n], want[n], 5e-6, f"utterance {n} (len {int(lens[n])})")
    assert_close(got_full, want_full, 5e-6, "no lengths")


def test_conformer_stack_fp32_path_on_an_outlier(device):
    """an activation whose row spans more than the planes hold (an outlier against tiny values) sends its blocks
    to the fp32 recomputation inside the launch: same answer as the per-launch path, counted"""
    from aps_amd import mega, nn_ops
    enc = _encoder(1, seed=5).to(device)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 33, 512, generator=g)
    x[1, 3, :] *= 1e-9
    x[1, 3, 17] = 3.0e4
    x = x.to(device)
    rel = (0.2 * torch.randn(65, 64, generator=g)).to(device)
    saved = mega.ENABLED
    try:
        mega.ENABLED = False
        want = enc.run(x, None, rel=rel)
        before = nn_ops.fp16x2_wide_tiles(device)
        mega.ENABLED = True
        got = enc.run(x, None, rel=rel)
        counted = nn_ops.fp16x2_wide_tiles(device) - before
    finally:
        mega.ENABLED = saved
    assert counted > 0
    assert_close(got, want, 5e-6, "outlier row")


def test_conformer_stack_follows_updated_and_replaced_parameters(device):
    """the layer table points at derived weights (LayerNorm folds, two-plane images) cached per encoder: an in-place
    update (optimiser step, load_state_dict), a REPLACED Parameter object and a changed BatchNorm buffer all rebuild it"""
    from aps_amd import mega
    enc = _encoder(1, seed=7).to(device)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(2, 30, 512, generator=g).to(device)
    rel = (0.2 * torch.randn(59, 64, generator=g)).to(device)
    saved = mega.ENABLED

    def both():
        mega.ENABLED = False
        want = enc.run(x, None, rel=rel)
        mega.ENABLED = True
        return enc.run(x, None, rel=rel), want

    try:
        got0, want0 = both()
        assert_close(got0, want0, 5e-6, "before any change")
        layer = enc.layers[0]
        layer.feedforward1[0].weight.mul_(1.5)     # in place, as an optimiser step is (a write through `.data` does
        #                                            not move the version counter the derived weights are keyed on:
        #                                            the library's caches, per-launch and here, share that rule)
        got1, want1 = both()
        assert_close(got1, want1, 5e-6, "after an in-place update")
        assert float((got1 - got0).abs().max()) > 1e-3
        layer.self_attn.out_proj.weight = torch.nn.Parameter(             # a new Parameter object, same shape
            0.05 * torch.randn(512, 512, generator=g).to(device))
        got2, want2 = both()
        assert_close(got2, want2, 5e-6, "after a replaced parameter")
        assert float((got2 - got1).abs().max()) > 1e-3
        layer.convolution[3].running_var.mul_(2.0)                       # a buffer of the folded BatchNorm
        got3, want3 = both()
        assert_close(got3, want3, 5e-6, "after a changed BatchNorm buffer")
        assert float((got3 - got2).abs().max()) > 1e-4
    finally:
        mega.ENABLED = saved


def test_conformer_stack_refuses_what_it_is_not_built_for(device):
    from aps_amd import mega
    from aps_amd.asr.transformer.impl import get_xfmr_encoder
    enc = _encoder(1).to(device)
    rel = torch.randn(2 * 70 - 1, 64, device=device)
    assert not mega.supported(enc, torch.randn(2, 70, 512, device=device), rel, None)           # T > 64
    assert not mega.supported(enc, torch.randn(2, 30, 512, device=device), None, None)          # no table
    assert not mega.supported(enc, torch.randn(2, 30, 512, device=device), rel[:59], (1, 2, 2))  # a context window
    small = get_xfmr_encoder("cfmr", "rel", 1, {"att_dim": 256, "nhead": 4, "feedforward_dim": 512,
                                                "att_dropout": 0, "ffn_dropout": 0, "kernel_size": 15}).eval().to(device)
    assert not mega.supported(small, torch.randn(2, 30, 256, device=device), rel[:59], None)     # D != 512
    enc.train()
    assert not mega.supported(enc, torch.randn(2, 30, 512, device=device), rel[:59], None)
