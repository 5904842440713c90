"""
GPU parity of the transformer encoder path (fp32 MFMA GEMM with fused epilogues, LayerNorm,
sinusoid add, attention core, and the assembled TransformerEncoder) against the golden vectors
recorded from the reference and against the CPU oracle (oracle/encoder_oracle.py).
Tolerance: 1e-4 of the activation scale (north star), measured ~1e-6.
"""
import numpy as np
import pytest
import torch

from tests.conftest import golden, assert_close

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (300, 200, 100), (1, 7, 5), (3200, 512, 512),
                                   (130, 1536, 512), (100, 512, 5120), (257, 96, 82)])
@pytest.mark.parametrize("relu,res,bias", [(False, False, True), (True, False, True),
                                           (False, True, True), (True, True, False)])
def test_linear_kernel(device, M, N, K, relu, res, bias):
    from aps_amd.nn_ops import linear
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K**0.5
    b = torch.randn(N, generator=g) if bias else None
    r = torch.randn(M, N, generator=g) if res else None
    ref = x.double() @ w.double().T
    if bias:
        ref = ref + b.double()
    if relu:
        ref = ref.relu()
    if res:
        ref = ref + r.double()
    out = linear(x.to(device), w.to(device), None if b is None else b.to(device),
                 None if r is None else r.to(device), relu)
    assert out.shape == (M, N)
    # fp32 accumulation: error grows ~sqrt(K) * 6e-8 of the scale
    assert_close(out, ref, 2e-6 if K <= 1024 else 6e-6, f"linear {M}x{N}x{K}")


def test_linear_batched_leading_dims(device):
    from aps_amd.nn_ops import linear
    g = torch.Generator().manual_seed(0)
    x, w = torch.randn(3, 17, 80, generator=g), torch.randn(40, 80, generator=g)
    out = linear(x.to(device), w.to(device))
    assert out.shape == (3, 17, 40)
    assert_close(out, x.double() @ w.double().T, 2e-6)


@pytest.mark.parametrize("D", [128, 512, 96, 1100])
def test_layernorm_kernel(device, D):
    from aps_amd.nn_ops import layernorm
    g = torch.Generator().manual_seed(D)
    x, r = torch.randn(37, D, generator=g) * 3 + 1, torch.randn(37, D, generator=g)
    w, b = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g)
    ref = torch.nn.functional.layer_norm((x + r).double(), (D,), w.double(), b.double(), 1e-5)
    out = layernorm(x.to(device), w.to(device), b.to(device), 1e-5, residual=r.to(device))
    assert_close(out, ref, 1e-5)
    out = layernorm(x.to(device), w.to(device), b.to(device), 1e-5)
    assert_close(out, torch.nn.functional.layer_norm(x.double(), (D,), w.double(), b.double()), 1e-5)


def test_posenc_kernel(device):
    from aps_amd.asr.transformer.pose import InputSinPosEncoding
    from oracle import encoder_oracle as eo
    pe = InputSinPosEncoding(128, scaled=True).to(device)
    x = torch.randn(2, 100, 128)
    ref = x * 128**0.5 + eo.sin_pos_enc(100, 128)
    assert_close(pe.add(x.to(device)), ref, 1e-5)
    assert pe(x.to(device)).shape == (100, 2, 128)  # reference layout at the module boundary


@pytest.mark.parametrize("T,H,dh", [(13, 4, 32), (100, 8, 64), (300, 2, 64), (50, 2, 128)])
def test_attention_core(device, T, H, dh):
    from aps_amd.nn_ops import attention_core
    g = torch.Generator().manual_seed(T)
    N, D = 3, H * dh
    qkv = torch.randn(N, T, 3 * D, generator=g)
    lens = torch.tensor([T, max(1, T - 7), max(1, T // 2)])

    def ref(lens_):
        q, k, v = [m.reshape(N, T, H, dh).permute(0, 2, 1, 3).double() for m in qkv.chunk(3, -1)]
        s = q @ k.transpose(-1, -2) / dh**0.5
        if lens_ is not None:
            pad = torch.arange(T)[None] >= lens_[:, None]
            s = s.masked_fill(pad[:, None, None, :], float("-inf"))
        return (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(N, T, D)

    assert_close(attention_core(qkv.to(device), H), ref(None), 1e-5, "no mask")
    assert_close(attention_core(qkv.to(device), H, lens.to(device)), ref(lens), 1e-5, "lens")


def _load_encoder(tag, device, pre, outp):
    from aps_amd.asr.transformer import TransformerEncoder
    enc = TransformerEncoder("xfmr", 40, output_proj=outp, num_layers=2, proj="conv2d",
                             proj_kwargs={"conv_channels": 16, "num_layers": 2}, pose="abs",
                             pose_kwargs={"dropout": 0},
                             arch_kwargs={"att_dim": 128, "nhead": 4, "feedforward_dim": 256,
                                          "att_dropout": 0, "ffn_dropout": 0, "pre_norm": pre})
    g = golden(tag)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    enc.load_state_dict(sd, strict=False)  # fixtures omit BatchNorm.num_batches_tracked
    return enc.eval().to(device), g


@pytest.mark.parametrize("tag,pre,outp", [("encoder_xfmr_abs_post", False, -1),
                                          ("encoder_xfmr_abs_pre", True, 96)])
def test_encoder_golden(device, tag, pre, outp):
    enc, g = _load_encoder(tag, device, pre, outp)
    out, n = enc(g["x"].to(device), None)
    assert n is None and out.shape == g["out_full"].shape
    assert_close(out, g["out_full"], TOL, tag + " full")
    out, n = enc(g["x"].to(device), g["lens"].to(device))
    assert torch.equal(n.cpu(), g["num_frames"])  # subsampled lengths: exact
    assert_close(out, g["out_len"], TOL, tag + " ragged")


def test_encoder_layer_reference_layout(device):
    """stand-alone layer / stack keep the reference's T x N x D call convention"""
    enc, g = _load_encoder("encoder_xfmr_abs_post", device, False, -1)
    x = torch.randn(3, 13, 128, device=device)
    a = enc.encoder.run(x, None)
    b = enc.encoder(x.transpose(0, 1)).transpose(0, 1)
    assert torch.equal(a, b)


def test_config4_encoder_vs_oracle(device):
    """BASELINE config 4 geometry (12 x 512, FF 2048, conv2d 256 x 2, 400 frames -> 100) on a
    small batch against the CPU oracle, random-initialised weights (seed 5)."""
    from aps_amd.asr.transformer import TransformerEncoder
    from oracle import encoder_oracle as eo
    torch.manual_seed(5)
    enc = TransformerEncoder("xfmr", 80, num_layers=12, proj="conv2d",
                             proj_kwargs={"conv_channels": 256, "num_layers": 2}, pose="abs",
                             pose_kwargs={"dropout": 0},
                             arch_kwargs={"att_dim": 512, "nhead": 8, "feedforward_dim": 2048,
                                          "att_dropout": 0, "ffn_dropout": 0,
                                          "pre_norm": False}).eval()
    assert sum(p.numel() for p in enc.parameters()) == 41_044_480  # 41.04 M
    g = torch.Generator().manual_seed(6)
    x = torch.randn(4, 400, 80, generator=g)
    lens = torch.tensor([400, 400, 333, 250])
    sd = {k: v.detach() for k, v in enc.state_dict().items()}
    ref, rn = eo.xfmr_abs_encoder(sd, x, lens, 12, 8)
    out, n = enc.to(device)(x.to(device), lens.to(device))
    assert out.shape == (4, 100, 512) and n.tolist() == rn.tolist() == [100, 100, 84, 63]
    assert_close(out, ref, TOL, "config 4 encoder")
