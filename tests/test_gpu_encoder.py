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


@pytest.fixture(params=["one-tile", "split-bd", "split-fp16", "split-panel", "panel-32x128", "panel-64x128",
                        "panel-occ4", "kgroup-16-waves", "kgroup-8-waves", "panel-dma", "panel-dma-64"])
def gemm_variant(request):
    """the kernels behind `linear`: the fp32 MFMA GEMM (small launches), the bf16 three-plane GEMM on
    the fragment image (aps_linear_split, layout 1), the fp16 two-plane GEMM with a planes pass over A
    (aps_linear_fp16x2, layout 2) and its panel form (aps_linear_panel, layout 3: the default for
    large launches), the split forms forced on for every launch whose weight is a Parameter and whose
    K is a multiple of 4"""
    from aps_amd import nn_ops
    name = request.param
    saved = nn_ops.SPLIT_MODE, nn_ops.SPLIT_LAYOUT, nn_ops.PANEL_FORM
    nn_ops.SPLIT_MODE = "0" if name == "one-tile" else "1"
    nn_ops.SPLIT_LAYOUT = {"split-bd": 1, "split-fp16": 2}.get(name, 3)
    # (0: the default = "occ4", four workgroups per CU)
    nn_ops.PANEL_FORM = {"panel-32x128": 1, "panel-64x128": 2, "panel-occ4": 3, "kgroup-16-waves": 4,
                         "kgroup-8-waves": 5, "panel-dma": 6, "panel-dma-64": 7}.get(name, 0)
    yield name
    nn_ops.SPLIT_MODE, nn_ops.SPLIT_LAYOUT, nn_ops.PANEL_FORM = saved


# last four shapes: whole tiles, ragged M and N edges, a short K loop (4 K steps),
# many tiles per CU
@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (300, 200, 100), (1, 7, 5), (70, 130, 20), (3200, 512, 512),
                                   (130, 1536, 512), (100, 512, 5120), (257, 96, 82),
                                   (8064, 512, 512), (8000, 520, 192), (4100, 1000, 128),
                                   (8064, 1536, 128),
                                   # the 32-utterance step's own launches (K groups of 128 and of 256), a K
                                   # extent that leaves the last K group short and one that leaves two empty
                                   (2016, 512, 1024), (2016, 1024, 512), (500, 260, 776), (333, 130, 224)])
@pytest.mark.parametrize("relu,res,bias", [(False, False, True), (True, False, True),
                                           (False, True, True), (True, True, False)])
def test_linear_kernel(device, M, N, K, relu, res, bias, gemm_variant):
    from aps_amd.nn_ops import linear
    if gemm_variant != "one-tile" and K % 4:
        pytest.skip("odd K goes to the fp32 kernel (padded operands)")
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
    wd = torch.nn.Parameter(w.to(device), requires_grad=False)  # (the split planes are cached on it)
    out = linear(x.to(device), wd, None if b is None else b.to(device),
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


@pytest.mark.parametrize("T,H,dh", [(13, 4, 32), (100, 8, 64), (300, 2, 64), (50, 2, 128),
                                    (63, 8, 64), (64, 2, 64), (17, 3, 64), (1, 2, 64),
                                    (65, 3, 64), (128, 2, 64), (120, 4, 64)])
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


def test_config4_encoder_full_batch_on_the_fp16_kernel_vs_oracle(device):
    """BASELINE config 4 as the bench runs it: batch 128 x 400 frames (M = 12 800 rows per
    projection), where the default dispatch takes the fp16 two-plane GEMM (asserted); the first 4
    utterances -- the inputs of test_config4_encoder_vs_oracle -- against the CPU oracle"""
    from aps_amd import nn_ops
    from aps_amd.asr.transformer import TransformerEncoder
    from oracle import encoder_oracle as eo
    assert nn_ops.SPLIT_MODE is None and nn_ops.SPLIT_LAYOUT == 3, "the default dispatch is under test"
    torch.manual_seed(5)
    enc = TransformerEncoder("xfmr", 80, num_layers=12, proj="conv2d",
                             proj_kwargs={"conv_channels": 256, "num_layers": 2}, pose="abs",
                             pose_kwargs={"dropout": 0},
                             arch_kwargs={"att_dim": 512, "nhead": 8, "feedforward_dim": 2048,
                                          "att_dropout": 0, "ffn_dropout": 0,
                                          "pre_norm": False}).eval()
    g = torch.Generator().manual_seed(6)
    x4 = torch.randn(4, 400, 80, generator=g)
    x = torch.cat([x4, torch.randn(124, 400, 80, generator=g)], 0)
    lens = torch.tensor([400, 400, 333, 250] + [400] * 124)
    sd = {k: v.detach() for k, v in enc.state_dict().items()}
    ref, rn = eo.xfmr_abs_encoder(sd, x4, lens[:4], 12, 8)
    nn_ops.GEMM_TIMELINE = timeline = []
    try:
        out, n = enc.to(device)(x.to(device), lens.to(device))
    finally:
        nn_ops.GEMM_TIMELINE = None
    kinds = {}
    for _, _, _, kind in timeline:
        kinds[kind] = kinds.get(kind, 0) + 1
    print(f"[config 4, batch 128] GEMM launches by kernel: {kinds}")
    assert kinds.get("split", 0) + kinds.get("panel", 0) + kinds.get("kgroup", 0) >= 48 and kinds.get("f32", 0) <= 2, kinds   # 4 per layer + projection
    assert n.tolist()[:4] == rn.tolist()
    assert_close(out[:4], ref, TOL, "config 4 encoder, batch 128 (fp16 two-plane projections)")


# ------------------------------------------------------------------------------------------------
# conformer / relative positions
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("act,alpha", [("swish", 1.0), (None, 0.5), ("swish", 0.5), ("relu", 2.0)])
def test_linear_activation_alpha(device, act, alpha):
    from aps_amd.nn_ops import linear
    g = torch.Generator().manual_seed(3)
    x, w = torch.randn(190, 96, generator=g), torch.randn(130, 96, generator=g) / 96**0.5
    b, r = torch.randn(130, generator=g), torch.randn(190, 130, generator=g)
    ref = x.double() @ w.double().T + b.double()
    if act == "swish":
        ref = ref * torch.sigmoid(ref)
    if act == "relu":
        ref = ref.relu()
    ref = ref * alpha + r.double()
    out = linear(x.to(device), w.to(device), b.to(device), r.to(device), act=act, alpha=alpha)
    assert_close(out, ref, 2e-6, f"linear act={act} alpha={alpha}")


@pytest.mark.parametrize("T,H,dh,rad", [(13, 4, 32, (4, 6)), (100, 8, 64, (256, 256)),
                                         (300, 2, 64, (100, 50)), (70, 2, 128, (16, 16)),
                                         (63, 8, 64, (256, 256)), (64, 2, 64, (10, 20)),
                                         (17, 3, 64, (5, 3)), (1, 2, 64, (2, 2)),
                                         # 64 < T <= 128, 64-wide heads: the two-tile MFMA form
                                         (65, 2, 64, (70, 70)), (100, 4, 64, (20, 12)),
                                         (127, 2, 64, (5, 200)), (128, 3, 64, (128, 128))])
def test_attention_core_relative(device, T, H, dh, rad):
    """score(i, j) = (q_i k_j + q_i E[clamp(j - i)]) / sqrt(dh) against the explicit float64 form"""
    from aps_amd.nn_ops import attention_core
    g = torch.Generator().manual_seed(T + 1)
    N, D = 3, H * dh
    qkv = torch.randn(N, T, 3 * D, generator=g)
    emb = torch.randn(rad[0] + rad[1] + 1, dh, generator=g)
    rel = emb[torch.arange(-T + 1, T).clamp(-rad[0], rad[1]) + rad[0]]  # 2T-1 x dh
    lens = torch.tensor([T, max(1, T - 7), max(1, T // 2)])
    q, k, v = [m.reshape(N, T, H, dh).permute(0, 2, 1, 3).double() for m in qkv.chunk(3, -1)]
    idx = torch.arange(T)[None, :] - torch.arange(T)[:, None] + T - 1
    s = (q @ k.transpose(-1, -2) + torch.einsum("nhld,lsd->nhls", q, rel.double()[idx])) / dh**0.5

    def ref(lens_):
        sc = s
        if lens_ is not None:
            pad = torch.arange(T)[None] >= lens_[:, None]
            sc = s.masked_fill(pad[:, None, None, :], float("-inf"))
        return (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(N, T, D)

    out = attention_core(qkv.to(device), H, rel=rel.to(device))
    assert_close(out, ref(None), 1e-5, "rel, no mask")
    out = attention_core(qkv.to(device), H, lens.to(device), rel=rel.to(device))
    assert_close(out, ref(lens), 1e-5, "rel, lens")


@pytest.mark.parametrize("N,T,D,K", [(2, 50, 128, 15), (3, 130, 512, 31), (1, 5, 40, 7),
                                     (2, 64, 300, 3), (2, 65, 256, 1)])
def test_glu_dwconv_kernel(device, N, T, D, K):
    from aps_amd.nn_ops import glu_dwconv
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(T * K)
    x = torch.randn(N, T, 2 * D, generator=g)
    w, b = torch.randn(D, 1, K, generator=g) / K**0.5, torch.randn(D, generator=g)
    scale, shift = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g)
    h = F.glu(x.double().transpose(1, 2), dim=-2)
    h = F.conv1d(h, w.double(), b.double(), padding=(K - 1) // 2, groups=D)
    h = h * scale.double()[:, None] + shift.double()[:, None]
    ref = (h * torch.sigmoid(h)).transpose(1, 2)
    out = glu_dwconv(x.to(device), w.to(device), b.to(device), scale.to(device), shift.to(device))
    assert out.shape == (N, T, D)
    assert_close(out, ref, 1e-5, f"glu_dwconv K={K}")
    out = glu_dwconv(x.to(device), w.to(device), None, None, None, swish=False)
    ref = F.conv1d(F.glu(x.double().transpose(1, 2), dim=-2), w.double(), None,
                   padding=(K - 1) // 2, groups=D).transpose(1, 2)
    assert_close(out, ref, 1e-5, f"glu_dwconv plain K={K}")


def _load_conformer(device):
    from aps_amd.asr.transformer import TransformerEncoder
    enc = TransformerEncoder("cfmr", 40, num_layers=2, proj="conv2d",
                             proj_kwargs={"conv_channels": 16, "num_layers": 2}, pose="rel",
                             pose_kwargs={"dropout": 0, "lradius": 6, "rradius": 9},
                             arch_kwargs={"att_dim": 128, "nhead": 4, "feedforward_dim": 256,
                                          "att_dropout": 0, "ffn_dropout": 0, "kernel_size": 7})
    g = golden("encoder_cfmr_rel")
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    missing, unexpected = enc.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    return enc.eval().to(device), g


def test_conformer_rel_golden(device):
    """TransformerEncoder("cfmr", pose "rel") against the activations recorded from the reference
    (clamped relative offsets: T' = 18 > radii 6 / 9)"""
    enc, g = _load_conformer(device)
    assert enc.encoder.norm is None  # reference quirk: final norm needs an explicit pre_norm kwarg
    out, n = enc(g["x"].to(device), None)
    assert n is None and out.shape == g["out_full"].shape
    assert_close(out, g["out_full"], TOL, "cfmr_rel full")
    out, n = enc(g["x"].to(device), g["lens"].to(device))
    assert torch.equal(n.cpu(), g["num_frames"])
    assert_close(out, g["out_len"], TOL, "cfmr_rel ragged")


@pytest.mark.parametrize("tag,arch,pose,pose_kwargs,kw", [
    ("encoder_cfmr_rel_t100", "cfmr", "rel", {"dropout": 0, "lradius": 20, "rradius": 12},
     {"kernel_size": 7}),
    ("encoder_xfmr_xl_t100", "xfmr", "xl", {"dropout": 0}, {})])
def test_relative_encoders_at_100_frames(device, tag, arch, pose, pose_kwargs, kw):
    """400 input frames -> 100 encoder frames, 64-wide heads (the chime4 conformer's geometry):
    learnt relative positions (clamped at radii 20 / 12) and the Transformer-XL form against the
    activations recorded from the reference; the attention runs on the two-tile MFMA form and must
    agree with the streaming kernel"""
    import os
    from aps_amd.asr.transformer.encoder import TransformerEncoder
    enc = TransformerEncoder(arch, 40, num_layers=2, proj="conv2d",
                             proj_kwargs={"conv_channels": 8, "num_layers": 2}, pose=pose,
                             pose_kwargs=pose_kwargs,
                             arch_kwargs={"att_dim": 128, "nhead": 2, "feedforward_dim": 192,
                                          "att_dropout": 0, "ffn_dropout": 0, **kw})
    g = golden(tag)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    missing, unexpected = enc.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    enc = enc.eval().to(device)
    out, n = enc(g["x"].to(device), None)
    assert n is None and out.shape == g["out_full"].shape and out.shape[1] == 100
    assert_close(out, g["out_full"], TOL, tag + " full")
    out_len, n = enc(g["x"].to(device), g["lens"].to(device))
    assert torch.equal(n.cpu(), g["num_frames"])
    assert_close(out_len, g["out_len"], TOL, tag + " ragged")
    os.environ["APS_ATT_GENERIC"] = "1"
    try:
        generic, _ = enc(g["x"].to(device), None)
    finally:
        del os.environ["APS_ATT_GENERIC"]
    assert not torch.equal(generic, out)  # (a different kernel ran)
    assert_close(out, generic, 1e-5, tag + " MFMA form vs streaming kernel")


def test_conformer_layer_reference_layout(device):
    enc, g = _load_conformer(device)
    x = torch.randn(3, 21, 128, device=device)
    rel = enc.pose.table(21)
    assert rel.shape == (41, 32)
    a = enc.encoder.run(x, None, rel=rel)
    b = enc.encoder(x.transpose(0, 1), inj_pose=rel).transpose(0, 1)
    assert torch.equal(a, b)
    layer = enc.encoder.layers[0]
    y = layer.conv(x.transpose(0, 1))  # T x N x D in and out (impl.py:491-505)
    assert y.shape == (21, 3, 128)
    with pytest.raises(RuntimeError):
        enc.encoder.run(x, None)  # relative layers need the table


@pytest.mark.parametrize("arch,pose,kw", [("cfmr", "rel", {}), ("cfmr", "abs", {"macaron": False}),
                                          ("cfmr", "rel", {"pre_norm": False}),
                                          ("xfmr", "rel", {"pre_norm": True})])
def test_chime4_conformer_vs_oracle(device, arch, pose, kw):
    """conf/asr/chime4/1a.yaml geometry (conv2d 128 x 2, rel radius 256, 512 / 8 heads / FF 1024,
    kernel 15; 4 layers here) against the torch-CPU restatement run by the reference modules'
    own weights; variants: no macaron, post-norm, relative transformer"""
    from aps_amd.asr.transformer import TransformerEncoder
    from oracle import encoder_oracle as eo
    torch.manual_seed(11)
    arch_kwargs = {"att_dim": 256, "nhead": 4, "feedforward_dim": 512, "att_dropout": 0,
                   "ffn_dropout": 0, **kw}
    if arch == "cfmr":
        arch_kwargs["kernel_size"] = 15
    pose_kwargs = {"dropout": 0, "lradius": 30, "rradius": 20} if pose == "rel" else {"dropout": 0}
    enc = TransformerEncoder(arch, 80, num_layers=3, proj="conv2d",
                             proj_kwargs={"conv_channels": 32, "num_layers": 2}, pose=pose,
                             pose_kwargs=pose_kwargs, arch_kwargs=arch_kwargs).eval()
    g = torch.Generator().manual_seed(12)
    for m in enc.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(0.1 * torch.randn(m.num_features, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.num_features, generator=g))
    x = torch.randn(3, 300, 80, generator=g)
    lens = torch.tensor([300, 222, 150])
    sd = {k: v.detach() for k, v in enc.state_dict().items()}
    ref, rn = eo.generic_encoder(sd, x, lens, arch=arch, pose=pose, num_layers=3, nhead=4,
                                 lradius=30, rradius=20, kernel_size=15,
                                 pre_norm=kw.get("pre_norm", arch == "cfmr"),
                                 macaron=kw.get("macaron", True))
    out, n = enc.to(device)(x.to(device), lens.to(device))
    assert n.tolist() == rn.tolist()
    assert_close(out, ref, TOL, f"{arch}_{pose} {kw}")


@pytest.mark.parametrize("tag,arch,pose,kw", [
    ("encoder_cfmr_abs_plain", "cfmr", "abs", {"macaron": False, "kernel_size": 5}),
    ("encoder_cfmr_rel_post", "cfmr", "rel", {"pre_norm": False, "kernel_size": 5}),
    ("encoder_xfmr_rel_pre", "xfmr", "rel", {"pre_norm": True})])
def test_encoder_variant_golden(device, tag, arch, pose, kw):
    """conformer without macaron / post-norm conformer / relative transformer against the
    reference's recorded activations"""
    from aps_amd.asr.transformer import TransformerEncoder
    pose_kwargs = {"dropout": 0, "lradius": 5, "rradius": 3} if pose == "rel" else {"dropout": 0}
    enc = TransformerEncoder(arch, 24, num_layers=1, proj="conv2d",
                             proj_kwargs={"conv_channels": 8, "num_layers": 2}, pose=pose,
                             pose_kwargs=pose_kwargs,
                             arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 96,
                                          "att_dropout": 0, "ffn_dropout": 0, **kw})
    g = golden(tag)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    missing, unexpected = enc.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    enc = enc.eval().to(device)
    out, _ = enc(g["x"].to(device), None)
    assert_close(out, g["out_full"], TOL, tag + " full")
    out, n = enc(g["x"].to(device), g["lens"].to(device))
    assert torch.equal(n.cpu(), g["num_frames"])
    assert_close(out, g["out_len"], TOL, tag + " ragged")


# ------------------------------------------------------------------------------------------------
# persistent LSTM (mask estimator recurrence)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,T,D,H,layers,bidir,ragged", [
    (3, 20, 40, 128, 1, False, False), (32, 60, 96, 512, 2, False, True),
    (5, 33, 64, 256, 2, True, True), (40, 25, 80, 640, 1, True, False),
    (70, 12, 32, 128, 1, False, True), (2, 249, 64, 512, 1, False, False),
    # decompositions: 128 rows in one launch, wide unit blocks, a chunk that must be halved
    (128, 14, 48, 512, 1, False, False), (100, 9, 40, 256, 2, False, True),
    (64, 11, 32, 1024, 1, True, True), (96, 10, 24, 320, 1, True, False),
    (17, 21, 16, 64, 3, False, True)])
def test_lstm_persistent_kernel(device, N, T, D, H, layers, bidir, ragged):
    """aps_lstm_layer against torch's CPU nn.LSTM in float64 (packed sequences for ragged
    batches); 1e-5 of the output scale after up to 249 recurrent steps"""
    from aps_amd import nn_ops
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    torch.manual_seed(N * 100 + T)
    rnn = torch.nn.LSTM(D, H, layers, batch_first=True, bidirectional=bidir).eval()
    x = torch.randn(N, T, D)
    lens = None
    if ragged:
        lens = torch.randint(1, T + 1, (N,))
        lens[0] = T
    ref_rnn = torch.nn.LSTM(D, H, layers, batch_first=True, bidirectional=bidir).double()
    ref_rnn.load_state_dict({k: v.double() for k, v in rnn.state_dict().items()})
    if ragged:
        packed = pack_padded_sequence(x.double(), lens.tolist(), batch_first=True,
                                      enforce_sorted=False)
        ref, _ = pad_packed_sequence(ref_rnn(packed)[0], batch_first=True, total_length=T)
    else:
        ref = ref_rnn(x.double())[0]
    assert nn_ops.lstm_supported(rnn.to(device), x.to(device))
    nn_ops.LSTM_CHECK = True
    try:
        out = nn_ops.lstm_forward(rnn, x.to(device), None if lens is None else lens.to(device))
    finally:
        nn_ops.LSTM_CHECK = False
    assert out.shape == ref.shape
    assert_close(out, ref, 1e-5, f"lstm N={N} T={T} H={H} L={layers} bidir={bidir}")


@pytest.mark.parametrize("N,T,H,L,ragged", [(32, 249, 512, 2, False), (7, 40, 128, 3, True),
                                            (20, 33, 64, 4, True), (16, 50, 256, 2, False),
                                            (32, 19, 512, 4, True), (30, 27, 256, 3, False)])
def test_lstm_stack_equals_per_layer(device, N, T, H, L, ragged):
    """layer-pipelined single launch vs one launch per layer (the upper layers' input projection
    moves from a batched GEMM into the recurrence's MFMA loop: same math, different summation
    order), both within 1e-5 of torch's float64 LSTM"""
    from aps_amd import nn_ops
    torch.manual_seed(N + T + H)
    rnn = torch.nn.LSTM(H // 2, H, L, batch_first=True).eval()
    x = torch.randn(N, T, H // 2)
    lens = None
    if ragged:
        lens = torch.randint(1, T + 1, (N,))
        lens[0] = T
    ref_rnn = torch.nn.LSTM(H // 2, H, L, batch_first=True).double()
    ref_rnn.load_state_dict({k: v.double() for k, v in rnn.state_dict().items()})
    if ragged:
        from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
        packed = pack_padded_sequence(x.double(), lens.tolist(), batch_first=True, enforce_sorted=False)
        ref, _ = pad_packed_sequence(ref_rnn(packed)[0], batch_first=True, total_length=T)
    else:
        ref = ref_rnn(x.double())[0]
    rnn = rnn.to(device)
    xd, ld = x.to(device), None if lens is None else lens.to(device)
    nn_ops.LSTM_CHECK = True
    try:
        assert nn_ops.LSTM_STACK
        stacked = nn_ops.lstm_forward(rnn, xd, ld)
        nn_ops.LSTM_STACK = False
        layered = nn_ops.lstm_forward(rnn, xd, ld)
    finally:
        nn_ops.LSTM_STACK, nn_ops.LSTM_CHECK = True, False
    assert_close(stacked, ref, 1e-5, "stacked vs float64")
    assert_close(layered, ref, 1e-5, "per layer vs float64")
    assert_close(stacked, layered, 1e-5, "stacked vs per layer")


def test_rnn_encoder_uses_persistent_lstm(device):
    """PyTorchRNNEncoder (mask net): own LSTM path == torch/MIOpen path on the same weights"""
    from aps_amd.asr.base.encoder import PyTorchRNNEncoder
    from aps_amd import nn_ops
    torch.manual_seed(2)
    enc = PyTorchRNNEncoder(100, 60, input_proj=64, rnn="lstm", num_layers=2, hidden=128,
                            dropout=0.0, bidirectional=True, non_linear="sigmoid").eval().to(device)
    x = torch.randn(4, 30, 100, device=device)
    lens = torch.tensor([30, 22, 17, 9], device=device)
    out, _ = enc(x, lens)
    saved = nn_ops.LSTM_HIDDEN_SIZES
    nn_ops.LSTM_HIDDEN_SIZES = ()
    try:
        ref, _ = enc(x, lens)  # the same LSTM step by step (aps_rnn_step): no persistent kernel
    finally:
        nn_ops.LSTM_HIDDEN_SIZES = saved
    assert out.shape == ref.shape == (4, 30, 60)
    assert_close(out, ref, 1e-5, "rnn encoder")


def _torch_rnn_reference(rnn, x, lens):
    """float64 CPU evaluation of an nn.RNNBase with packed-sequence semantics"""
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    import copy
    ref = copy.deepcopy(rnn).cpu().double()
    xd = x.detach().cpu().double()
    if lens is None:
        return ref(xd)[0]
    packed = pack_padded_sequence(xd, lens.cpu().tolist(), batch_first=True, enforce_sorted=False)
    out, _ = pad_packed_sequence(ref(packed)[0], batch_first=True, total_length=x.shape[1])
    return out


@pytest.mark.parametrize("mode,hidden,proj", [("GRU", 96, 0), ("RNN_TANH", 50, 0), ("RNN_RELU", 64, 0),
                                              ("LSTM", 100, 0), ("LSTM", 128, 48)])
@pytest.mark.parametrize("bidir", [False, True])
@pytest.mark.parametrize("use_lens", [False, True])
def test_rnn_step_forward(device, mode, hidden, proj, bidir, use_lens):
    """the recurrences without a persistent kernel (GRU, tanh / relu RNN, LSTMs of other sizes or
    with a projection) step by step on aps_rnn_step + aps_linear, vs torch in float64 on the CPU
    with packed sequences; two layers, both directions, ragged lengths"""
    from aps_amd.asr.base.encoder import PyTorchRNN, var_len_rnn_forward
    from aps_amd.nn_ops import rnn_step_supported
    torch.manual_seed(hidden + proj)
    rnn = PyTorchRNN(mode, 40, hidden, num_layers=2, proj_size=proj, bidirectional=bidir).eval()
    x = torch.randn(5, 23, 40)
    lens = torch.tensor([23, 23, 14, 9, 1]) if use_lens else None
    ref = _torch_rnn_reference(rnn, x, lens)
    rnn = rnn.to(device)
    with torch.no_grad():
        assert rnn_step_supported(rnn, x.to(device))
        out = var_len_rnn_forward(rnn, x.to(device), None if lens is None else lens.to(device))
    assert out.shape == ref.shape
    assert_close(out, ref, 1e-5, f"{mode} {hidden}/{proj} bidir={bidir}")


def test_variant_rnn_encoder_with_gru(device):
    """PyTorchRNNEncoder(rnn="gru") -- the reference's encoder factory with a non-LSTM cell -- runs on
    the step kernels and equals torch"""
    from aps_amd.asr.base.encoder import PyTorchRNNEncoder
    torch.manual_seed(8)
    enc = PyTorchRNNEncoder(30, 20, input_proj=32, rnn="gru", num_layers=2, hidden=48, dropout=0.0,
                            bidirectional=True, non_linear="tanh").eval()
    x = torch.randn(3, 17, 30)
    lens = torch.tensor([17, 11, 6])
    h = torch.relu(enc.proj(x)).double()
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    import copy
    impl = copy.deepcopy(enc.impl).double()
    packed = pack_padded_sequence(h, lens.tolist(), batch_first=True, enforce_sorted=False)
    y, _ = pad_packed_sequence(impl(packed)[0], batch_first=True)
    ref = torch.tanh(y @ enc.outp.weight.double().T + enc.outp.bias.double())
    out, _ = enc.to(device)(x.to(device), lens.to(device))
    assert_close(out, ref, 1e-5, "gru encoder")


# ------------------------------------------------------------------------------------------------
# channels-last implicit-GEMM convolution
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,H,W,Ci,Co,k,s,p,tr,op", [
    (2, 20, 11, 32, 64, (3, 3), (2, 1), (1, 1), False, (0, 0)),     # DCCRN encoder block
    (3, 33, 9, 64, 48, (5, 2), (2, 1), (2, 1), False, (0, 0)),      # DCUNet-style kernel, ragged Co
    (2, 40, 30, 1, 16, (3, 3), (2, 2), (1, 1), False, (0, 0)),      # conv2d subsampling, layer 1
    (2, 20, 15, 128, 128, (3, 3), (2, 2), (1, 1), False, (0, 0)),   # conv2d subsampling, layer 2
    (2, 9, 7, 2, 32, (3, 3), (2, 1), (1, 1), False, (0, 0)),        # complex first layer (2 ch)
    (2, 5, 8, 64, 32, (3, 3), (2, 1), (1, 1), True, (0, 0)),        # decoder block
    (2, 6, 5, 32, 4, (3, 3), (2, 1), (1, 1), True, (1, 0)),         # last decoder layer, out pad
    (3, 21, 70, 64, 40, (3, 3), (1, 2), (1, 1), True, (0, 1)),      # channels-last DCCRN decoder: F strided
    (2, 13, 17, 32, 64, (5, 4), (3, 2), (2, 1), True, (2, 1)),      # 6 residue classes, ragged taps
    (2, 9, 11, 32, 16, (1, 2), (2, 3), (0, 0), True, (1, 2)),       # kernel < stride: classes with no tap
    (2, 10, 9, 64, 64, (3, 3), (2, 2), (1, 1), True, (1, 1)),       # 2 x 2 upsampling
    (1, 7, 6, 96, 70, (1, 1), (1, 1), (0, 0), False, (0, 0))])      # 1 x 1
@pytest.mark.parametrize("split", [False, True, "fp16"], ids=["fp32-mfma", "bf16-split", "fp16x2"])
def test_conv2d_nhwc(device, N, H, W, Ci, Co, k, s, p, tr, op, split):
    """implicit-GEMM / direct convolution vs torch's float64 NCHW conv2d / conv_transpose2d; the
    bf16-split form (aps_conv2d_nhwc_split) and the fp16 two-plane form (aps_conv2d_nhwc_fp16x2) are
    forced on for every shape they take (Ci % 32 == 0)"""
    from aps_amd import nn_ops
    from aps_amd.nn_ops import conv2d_nhwc
    import torch.nn.functional as F
    if split and Ci % 32:
        pytest.skip("the split forms need whole 32-channel K steps")
    saved = nn_ops.SPLIT_MODE, nn_ops.CONV_SPLIT_MIN_CO, nn_ops.SPLIT_LAYOUT, nn_ops.CONV_FP16X2
    nn_ops.SPLIT_MODE, nn_ops.CONV_SPLIT_MIN_CO, nn_ops.SPLIT_LAYOUT = ("1" if split else "0"), 1, 1
    nn_ops.CONV_FP16X2 = split == "fp16"  # (None = per call site; here forced either way)
    try:
        _conv2d_nhwc_case(device, N, H, W, Ci, Co, k, s, p, tr, op, bool(split), conv2d_nhwc, F)
    finally:
        nn_ops.SPLIT_MODE, nn_ops.CONV_SPLIT_MIN_CO, nn_ops.SPLIT_LAYOUT, nn_ops.CONV_FP16X2 = saved


def _conv2d_nhwc_case(device, N, H, W, Ci, Co, k, s, p, tr, op, split, conv2d_nhwc, F):
    g = torch.Generator().manual_seed(H * W + Ci)
    x = torch.randn(N, Ci, H, W, generator=g)
    fan = Ci * k[0] * k[1]
    scale, shift = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
    if tr:
        w = torch.randn(Ci, Co, *k, generator=g) / fan**0.5     # nn.ConvTranspose2d layout
        ref = F.conv_transpose2d(x.double(), w.double(), None, s, p, op)
        wl = w.permute(1, 2, 3, 0).contiguous()
    else:
        w = torch.randn(Co, Ci, *k, generator=g) / fan**0.5     # nn.Conv2d layout
        ref = F.conv2d(x.double(), w.double(), None, s, p)
        wl = w.permute(0, 2, 3, 1).contiguous()
    ref = ref * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    res = torch.randn(ref.shape, generator=g)
    ref = F.leaky_relu(ref, 0.01) + res.double()
    wd = wl.to(device)
    if split:
        wd._aps_persistent = True  # what the modules' weight caches do: the planes hang on it
    out = conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(device), wd,
                      scale.to(device), shift.to(device), s, p, tr, op, "leaky_relu", 0.01,
                      res.permute(0, 2, 3, 1).contiguous().to(device))
    assert out.shape == ref.permute(0, 2, 3, 1).shape
    assert ("_aps_split" in wd.__dict__) == split
    assert_close(out, ref.permute(0, 2, 3, 1), 1e-5, f"conv {N}x{H}x{W}x{Ci}->{Co} tr={tr}")
    out = conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(device), wd, None, None,
                      s, p, tr, op, "relu")
    plain = (F.conv_transpose2d(x.double(), w.double(), None, s, p, op) if tr else
             F.conv2d(x.double(), w.double(), None, s, p)).relu()
    assert_close(out, plain.permute(0, 2, 3, 1), 1e-5, "plain relu")


# ------------------------------------------------------------------------------------------------
# Transformer-XL attention, context windows, linear / conv1d projections
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag,arch,pose,kw,top", [
    ("encoder_xfmr_xl_ctx", "xfmr", "xl", {},
     dict(proj="linear", proj_kwargs={}, lctx=2, rctx=1, chunk_size=2, num_layers=2)),
    ("encoder_cfmr_xl_tie", "cfmr", "xl", {"kernel_size": 5, "tie": True},
     dict(proj="conv1d", proj_kwargs={"dim": 32, "num_layers": 2}, num_layers=2)),
    ("encoder_xfmr_abs_lctx", "xfmr", "abs", {}, dict(lctx=3, rctx=0, chunk_size=1))])
def test_encoder_xl_ctx_golden(device, tag, arch, pose, kw, top):
    from aps_amd.asr.transformer import TransformerEncoder
    top = dict(dict(num_layers=1, proj="conv2d", proj_kwargs={"conv_channels": 8, "num_layers": 2}),
               **top)
    pose_kwargs = {"dropout": 0, "lradius": 5, "rradius": 3} if pose == "rel" else {"dropout": 0}
    enc = TransformerEncoder(arch, 24, pose=pose, pose_kwargs=pose_kwargs,
                             arch_kwargs={"att_dim": 64, "nhead": 2, "feedforward_dim": 96,
                                          "att_dropout": 0, "ffn_dropout": 0, **kw}, **top)
    g = golden(tag)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    missing, unexpected = enc.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing,
                                                                                       unexpected)
    enc = enc.eval().to(device)
    out, _ = enc(g["x"].to(device), None)
    assert_close(out, g["out_full"], TOL, tag + " full")
    out, n = enc(g["x"].to(device), g["lens"].to(device))
    assert torch.equal(n.cpu(), g["num_frames"])
    # padded queries whose whole window is padding are NaN in the reference, zeros here: valid frames
    valid = torch.arange(out.shape[1])[None] < g["num_frames"][:, None]
    assert_close(out.cpu()[valid], g["out_len"][valid], TOL, tag + " ragged")


@pytest.mark.parametrize("T,H,dh,win", [(50, 2, 64, (1, 3, 0)), (63, 4, 64, (4, 1, 1)),
                                        (150, 2, 64, (8, 2, 0)), (40, 2, 32, (1, -1, 0)),
                                        (100, 2, 64, (8, 2, 0)), (128, 3, 64, (1, -1, -1)),
                                        (65, 2, 64, (16, 1, 1))])
def test_attention_xl_window_kernels(device, T, H, dh, win):
    """XL biases, per-head tables, value-as-query and context windows in both attention kernels
    against the float64 explicit form"""
    from aps_amd.nn_ops import attention_core
    from oracle import encoder_oracle as eo
    g = torch.Generator().manual_seed(T + dh)
    N, D = 2, H * dh
    qkv = torch.randn(N, T, 3 * D, generator=g)
    table = torch.randn(H, 2 * T - 1, dh, generator=g)
    u, v = torch.randn(H, dh, generator=g), torch.randn(H, dh, generator=g)
    lens = torch.tensor([T, max(1, T - 9)])
    val, key = [m.reshape(N, T, H, dh).double() for m in (qkv[..., 2 * D:], qkv[..., D:2 * D])]
    idx = torch.arange(T)[None, :] - torch.arange(T)[:, None] + T - 1
    ac = torch.einsum("nlhd,nshd->nhls", val + u.double(), key)
    bd = torch.einsum("nlhd,hlsd->nhls", val + v.double(), table.double()[:, idx])
    score = (ac + bd) / dh**0.5 + eo.context_mask(T, *win).double()[None, None]
    score = score.masked_fill((torch.arange(T)[None] >= lens[:, None])[:, None, None, :], float("-inf"))
    ref = torch.einsum("nhls,nshd->nlhd", torch.softmax(score, -1), val).reshape(N, T, D)
    out = attention_core(qkv.to(device), H, lens.to(device), rel=table.to(device),
                         rel_u=u.to(device), rel_v=v.to(device), query_from_value=True,
                         chunk_size=win[0], lctx=win[1], rctx=win[2])
    valid = ~torch.isnan(ref).any(-1)
    assert valid[0].all()
    assert_close(out.cpu()[valid], ref[valid], 1e-5, f"xl window T={T}")


@pytest.mark.parametrize("M,N,K,act,res", [(2016, 512, 512, None, True), (300, 1024, 512, "swish", False),
                                           (70, 96, 256, "relu", True), (129, 1536, 516, None, False),
                                           (5, 7, 81, None, False),
                                           # the merged-batch shapes (many tiles per CU)
                                           (8064, 1024, 512, "swish", True), (8001, 520, 128, None, False),
                                           (8064, 512, 192, "relu", True)])
def test_linear_with_folded_layernorm(device, M, N, K, act, res, gemm_variant):
    """LN(x) W^T + b inside one GEMM launch (weights pre-scaled by gamma, row statistics accumulated
    in the kernel) against float64 LayerNorm + matmul; rows with a large mean included"""
    from aps_amd.nn_ops import linear
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g) * 2 + torch.randn(M, 1, generator=g) * 3
    w = torch.randn(N, K, generator=g) / K**0.5
    b = torch.randn(N, generator=g)
    ln = torch.nn.LayerNorm(K)
    ln.weight.data.copy_(0.5 + torch.rand(K, generator=g))
    ln.bias.data.copy_(0.3 * torch.randn(K, generator=g))
    r = torch.randn(M, N, generator=g) if res else None
    ref = torch.nn.functional.layer_norm(x.double(), (K,), ln.weight.double(), ln.bias.double(), ln.eps)
    ref = ref @ w.double().T + b.double()
    if act == "swish":
        ref = ref * torch.sigmoid(ref)
    if act == "relu":
        ref = ref.relu()
    ref = ref * 0.5 + (r.double() if res else 0)
    wd = torch.nn.Parameter(w.to(device), requires_grad=False)
    out = linear(x.to(device), wd, b.to(device), None if r is None else r.to(device),
                 act=act, alpha=0.5, ln=ln.to(device))
    assert_close(out, ref, 1e-5, f"linear+LN {M}x{N}x{K}")


def test_split_planes_follow_the_weight(device):
    """the bf16 planes of a weight are cached on its Parameter and rebuilt when it changes in place
    (an optimiser step, load_state_dict); views of one Parameter share the entry; temporaries are
    never cached and take the fp32 kernel"""
    from aps_amd import nn_ops
    saved = nn_ops.SPLIT_MODE
    nn_ops.SPLIT_MODE = "1"
    try:
        g = torch.Generator().manual_seed(3)
        x = torch.randn(500, 96, generator=g).to(device)
        conv = torch.nn.Conv1d(96, 80, 1).to(device)  # weight 80 x 96 x 1, used as .view(80, 96)
        for p in conv.parameters():
            p.requires_grad_(False)
        w2 = conv.weight.view(80, 96)
        ref = lambda: (x.double() @ conv.weight.double().view(80, 96).T).cpu()
        assert_close(nn_ops.linear(x, w2), ref(), 2e-6, "split on a view")
        assert "_aps_split" in conv.weight.__dict__
        first = conv.weight.__dict__["_aps_split"]["w"][1]
        assert nn_ops.linear(x, conv.weight.view(80, 96)) is not None
        assert conv.weight.__dict__["_aps_split"]["w"][1] is first  # second view: cache hit
        with torch.no_grad():
            conv.weight.mul_(-0.5)  # in place, as an optimiser step / load_state_dict does: version bump
        assert_close(nn_ops.linear(x, conv.weight.view(80, 96)), ref(), 2e-6, "after the update")
        assert conv.weight.__dict__["_aps_split"]["w"][1] is not first
        tmp = conv.weight.view(80, 96) * 2.0  # a temporary: no owner, fp32 kernel, nothing cached
        assert nn_ops._weight_owner(tmp) is None
        assert_close(nn_ops.linear(x, tmp), 2 * ref(), 2e-6, "temporary weight")
    finally:
        nn_ops.SPLIT_MODE = saved


@pytest.mark.parametrize("layout", [2, 3], ids=["planes-pass", "panel"])
def test_fp16x2_non_finite_rows(device, layout):
    """a NaN or an Inf in a row of A stays in that row of C (the row's exponent comes from its
    finite maximum / is clamped; no other row sees it); zero rows and rows of subnormals are exact"""
    from aps_amd import nn_ops
    saved = nn_ops.SPLIT_MODE, nn_ops.SPLIT_LAYOUT
    nn_ops.SPLIT_MODE, nn_ops.SPLIT_LAYOUT = "1", layout
    try:
        g = torch.Generator().manual_seed(11)
        M, K, N = 200, 256, 160
        x = torch.randn(M, K, generator=g)
        x[3, 17] = float("nan")
        x[5, 100] = float("inf")
        x[7] = 0.0
        x[9] = torch.randn(K, generator=g) * 1e-41  # subnormals
        w = torch.randn(N, K, generator=g) / K**0.5
        out = nn_ops.linear(x.to(device), torch.nn.Parameter(w.to(device), requires_grad=False)).cpu()
        ref = x.double() @ w.double().T
        good = [r for r in range(M) if r not in (3, 5)]
        assert torch.isfinite(out[good]).all()
        assert_close(out[good], ref[good], 2e-6, "finite rows")
        assert not torch.isfinite(out[3]).any() and not torch.isfinite(out[5]).any()
        assert (out[7] == 0).all()
        # (rows 3 and 5 send this 64-row tile to the fp32 path -- the finite elements of a row with an
        # Inf sit 2^113 below its "maximum" -- so the subnormal row is summed like plain fp32 sums it:
        # every product rounded to the subnormal grid of 1.4e-45)
        sub = ref[9].abs().max()
        assert ((out[9].double() - ref[9]).abs().max() <= 1e-5 * sub + 64 * 1.5e-45)
    finally:
        nn_ops.SPLIT_MODE, nn_ops.SPLIT_LAYOUT = saved


def _componentwise(out, a, w, extra=None):
    """max |out - a w^T (+ extra)| / (|a| |w|^T), float64 (the bound the fp16 two-plane GEMM is held to)"""
    a64, w64 = a.double(), w.double()
    ref = a64 @ w64.T
    if extra is not None:
        ref = ref + extra.double()
    bound = a64.abs() @ w64.abs().T
    q = (out.double().cpu() - ref).abs() / bound.clamp_min(1e-300)
    return q[bound > 0].max().item()


@pytest.fixture(params=[(2, 0), (3, 0), (3, 1), (3, 2), (3, 3), (3, 4), (3, 5), (3, 6), (3, 7)],
                ids=["planes-pass", "panel", "panel-32x128-occ2", "panel-64x128", "panel-occ4", "kgroup-16-waves",
                     "kgroup-8-waves", "panel-dma", "panel-dma-64"])
def fp16x2_forced(request):
    """both forms of the fp16 two-plane GEMM (aps_linear_fp16x2: planes of A from a pass of their own, a
    power of two per row; aps_linear_panel: planes formed in the kernel, a power of two per row and
    K chunk), forced on for every launch"""
    from aps_amd import nn_ops
    saved = nn_ops.SPLIT_MODE, nn_ops.SPLIT_LAYOUT, nn_ops.PANEL_FORM
    nn_ops.SPLIT_MODE, (nn_ops.SPLIT_LAYOUT, nn_ops.PANEL_FORM) = "1", request.param
    yield nn_ops
    nn_ops.SPLIT_MODE, nn_ops.SPLIT_LAYOUT, nn_ops.PANEL_FORM = saved


@pytest.mark.parametrize("in_row_range", [1e5, 1e6, 1e7, 1e8, 1e9, 1e10])
@pytest.mark.parametrize("M,N,K", [(500, 300, 512), (8064, 512, 512)])
def test_fp16x2_wide_range_outlier_meets_zero_weight(device, fp16x2_forced, in_row_range, M, N, K):
    """the round-2 verdict's counter-example on the KERNEL: A ~ 1e-3 N(0,1) with one column
    `in_row_range` times larger that meets a zero weight column -- the row maximum bounds no output.
    Component-wise bound against float64 at every range; beyond what two planes hold the tiles are
    recomputed on the fp32 MFMA inside the launch (counted), below nothing is"""
    nn_ops = fp16x2_forced
    g = torch.Generator().manual_seed(int(np.log10(in_row_range)) + M)
    a = 1e-3 * torch.randn(M, K, generator=g)
    a[:, 3] = 1e-3 * in_row_range * torch.sign(torch.randn(M, generator=g))
    w = torch.randn(N, K, generator=g)
    w[:, 3] = 0
    b = torch.randn(N, generator=g) * 1e-3
    before = nn_ops.fp16x2_wide_tiles(device)
    out = nn_ops.linear(a.to(device), torch.nn.Parameter(w.to(device), requires_grad=False), b.to(device))
    redone = nn_ops.fp16x2_wide_tiles(device) - before
    tiles = nn_ops.fp16x2_tiles(M, N)
    # (the bias is added in fp32: one more rounding of the output, 2^-24 of |out| <= the bound's scale)
    q = _componentwise(out - b.to(device), a, w)
    print(f"[fp16x2] range {in_row_range:.0e} {M}x{N}x{K}: 2^{np.log2(q):.1f} of sum|a||w|, "
          f"{redone} of {tiles} tiles in fp32")
    assert q <= 2.0 ** -19
    if in_row_range >= 1e9:
        assert redone == tiles          # every row holds elements 2^-31 below its maximum
    if in_row_range <= 1e5:
        # (2^-30 below a maximum 1e5 times the typical element = 1e-4 of it: one Gaussian value in
        # 12 000 falls there, i.e. most 64-row tiles hold one)
        assert redone < tiles


def test_fp16x2_well_scaled_operands_never_take_the_fp32_path(device, fp16x2_forced):
    nn_ops = fp16x2_forced
    g = torch.Generator().manual_seed(2)
    before = nn_ops.fp16x2_wide_tiles(device)
    for M, N, K in ((8064, 1024, 512), (3000, 384, 1024)):
        a = torch.randn(M, K, generator=g) * torch.exp(torch.empty(M, 1).uniform_(-11, 11, generator=g))
        w = torch.randn(N, K, generator=g) / K**0.5
        out = nn_ops.linear(a.to(device), torch.nn.Parameter(w.to(device), requires_grad=False))
        assert _componentwise(out, a, w) <= 2.0 ** -20.5
    assert nn_ops.fp16x2_wide_tiles(device) == before


@pytest.mark.parametrize("case", ["lognormal", "relu-sparse", "rows-1e-6..1e6"])
def test_fp16x2_heavy_tails_componentwise(device, fp16x2_forced, case):
    nn_ops = fp16x2_forced
    g = torch.Generator().manual_seed(17)
    M, N, K = 1000, 640, 768
    z = torch.randn(M, K, generator=g)
    if case == "lognormal":
        a = torch.exp(3 * z) * torch.sign(torch.randn(M, K, generator=g))
    elif case == "relu-sparse":
        a = torch.relu(z - 1.0) * 50          # mostly exact zeros
    else:
        a = z * torch.exp(torch.empty(M, 1).uniform_(-13.8, 13.8, generator=g))
    w = torch.randn(N, K, generator=g) / K**0.5
    wd = torch.nn.Parameter(w.to(device), requires_grad=False)
    before = nn_ops.fp16x2_wide_tiles(device)
    out = nn_ops.linear(a.to(device), wd)
    redone = nn_ops.fp16x2_wide_tiles(device) - before
    q = _componentwise(out, a, w)
    # what a plain fp32 evaluation of the same product loses (the fp32 MFMA kernel: K / 2 sequential
    # fp32 additions per output; the rows the planes cannot hold are summed exactly like that)
    nn_ops.SPLIT_MODE = "0"
    q32 = _componentwise(nn_ops.linear(a.to(device), wd), a, w)
    nn_ops.SPLIT_MODE = "1"
    print(f"[fp16x2] {case}: 2^{np.log2(q):.1f} of sum|a||w| (fp32 MFMA kernel: 2^{np.log2(q32):.1f}), "
          f"{redone} tiles in fp32")
    # (the maxima of two different summation orders over 640 000 outputs: within a factor of 1.5)
    assert q <= max(2.0 ** -19, 1.5 * q32)


def test_fp16x2_wide_weight_rows(device, fp16x2_forced):
    """the same guard on the weight side: a weight row (output column) with an element 2^-31 below
    its maximum is flagged when the image is built and its tiles take the fp32 path"""
    nn_ops = fp16x2_forced
    g = torch.Generator().manual_seed(23)
    M, N, K = 700, 300, 256
    a = torch.randn(M, K, generator=g)
    a[:, 5] = 0                                  # nothing looks at the weights' outlier column
    w = 1e-4 * torch.randn(N, K, generator=g)
    w[130:140, 5] = 1e7                          # rows 130..139: in-row range 1e11
    before = nn_ops.fp16x2_wide_tiles(device)
    out = nn_ops.linear(a.to(device), torch.nn.Parameter(w.to(device), requires_grad=False))
    redone = nn_ops.fp16x2_wide_tiles(device) - before
    assert _componentwise(out, a, w) <= 2.0 ** -19
    # the one column tile that holds rows 128..255, in every row panel
    # the one column tile that holds rows 128..255, in every row panel
    rows = nn_ops.nat.load().aps_linear_panel_rows(M, N, nn_ops.PANEL_FORM) if nn_ops.SPLIT_LAYOUT == 3 else 64
    assert redone == (M + rows - 1) // rows


@pytest.mark.parametrize("in_row_range", [1e4, 1e9])
def test_fp16x2_layernorm_fold_with_outlier_channel(device, fp16x2_forced, in_row_range):
    """the LayerNorm-folded form runs on the RAW pre-norm rows -- where trained transformers carry
    outlier channels.  One channel `in_row_range` times the others, which the folded weight ignores
    (gamma = 0 there): against float64 LayerNorm + Linear, on the planes and on the fp32 path"""
    nn_ops = fp16x2_forced
    g = torch.Generator().manual_seed(31)
    M, N, K = 2000, 384, 512
    x = torch.randn(M, K, generator=g)
    x[:, 9] = in_row_range * (1 + 0.1 * torch.randn(M, generator=g))
    ln = torch.nn.LayerNorm(K)
    ln.weight.data = torch.rand(K, generator=g) + 0.5
    ln.weight.data[9] = 0
    ln.bias.data = 0.1 * torch.randn(K, generator=g)
    w = torch.randn(N, K, generator=g) / K**0.5
    b = 0.1 * torch.randn(N, generator=g)
    ref = torch.nn.functional.layer_norm(x.double(), (K,), ln.weight.double(), ln.bias.double(), ln.eps) \
        @ w.double().T + b.double()
    ln = ln.to(device)
    for p in ln.parameters():
        p.requires_grad_(False)
    before = nn_ops.fp16x2_wide_tiles(device)
    out = nn_ops.linear(x.to(device), torch.nn.Parameter(w.to(device), requires_grad=False), b.to(device), ln=ln)
    redone = nn_ops.fp16x2_wide_tiles(device) - before
    # (at 1e4 only a chance value near zero -- below 2^-30 of the outlier = 1e-5 of a typical element --
    # sends a tile to the fp32 path: a handful of them; at 1e9 every row holds such elements)
    tiles = nn_ops.fp16x2_tiles(M, N)
    assert redone == tiles if in_row_range >= 1e9 else redone <= tiles // 4
    # what the fold can deliver in fp32 is bounded by its own cancellation x W' - mean colsum: the
    # same bound as the fp32 MFMA kernel's fold (nn.hip) on these rows
    nn_ops.SPLIT_MODE = "0"
    f32 = nn_ops.linear(x.to(device), torch.nn.Parameter(w.to(device), requires_grad=False), b.to(device), ln=ln)
    nn_ops.SPLIT_MODE = "1"
    from tests.conftest import rel_err
    e16, e32 = rel_err(out, ref), rel_err(f32, ref)
    print(f"[fp16x2] LN fold, outlier {in_row_range:.0e}: {e16:.2e} of scale (fp32 MFMA fold {e32:.2e}), "
          f"{redone} tiles in fp32")
    assert e16 <= max(2e-6, 2 * e32)


@pytest.mark.parametrize("norm,dilation,stride", [("IN", 1, 2), ("IN", (2, 1), (2, 1)), ("BN", 2, 2),
                                                  ("BN", (1, 2), 1), ("IN", 2, 1)])
def test_conv2d_block_instance_norm_and_dilation(device, norm, dilation, stride):
    """Conv2d -> Norm -> ReLU (component.py:251-307) with InstanceNorm2d and with dilated kernels:
    the dilated convolution runs as its dense equivalent on aps_conv2d_nhwc, InstanceNorm as the
    all-band CMVN of every (utterance, channel) plane; against the torch modules in float64, and
    the two-layer Conv2dEncoder chain with its output projection"""
    from aps_amd.asr.base.component import Conv2d
    from aps_amd.asr.base.encoder import Conv2dEncoder
    torch.manual_seed(5)
    blk = Conv2d(3, 24, kernel_size=3, stride=stride, dilation=dilation, norm=norm).eval()
    if norm == "BN":
        bn = blk.norm.norm
        bn.running_mean.normal_(0, 0.3)
        bn.running_var.uniform_(0.5, 1.5)
        bn.weight.data.uniform_(0.5, 1.5)
        bn.bias.data.normal_(0, 0.2)
    x = torch.randn(4, 3, 37, 20) * 2 + 0.5
    ref = torch.relu(blk.norm.double()(blk.conv.double()(x.double())))
    blk = blk.float().to(device)
    assert blk.fusible()
    out = blk(x.to(device))
    assert out.shape == ref.shape
    assert_close(out, ref, 2e-5, f"Conv2d block {norm} dilation {dilation}")
    enc = Conv2dEncoder(20, 32, channel=[8, 16], num_layers=2, norm=norm).eval()
    x1 = torch.randn(3, 50, 20)
    lens = torch.tensor([50, 41, 33])
    ref1, rl = x1.double()[:, None], lens
    encd = enc.double()
    for c in encd.enc_layers:
        ref1 = torch.relu(c.norm(c.conv(ref1)))
    ref1 = encd.outp(ref1.transpose(1, 2).contiguous().view(3, ref1.shape[2], -1))
    enc = enc.float().to(device)
    got, gl = enc(x1.to(device), lens.to(device))
    assert_close(got, ref1, 2e-5, f"Conv2dEncoder {norm}")
    assert gl.tolist() == [int(c) for c in enc.enc_layers[1].compute_outp_dim(
        enc.enc_layers[0].compute_outp_dim(lens, 0), 0)]


@pytest.mark.parametrize("layout", [2, 3], ids=["planes-pass", "panel"])
@pytest.mark.parametrize("M,D,F", [(300, 96, 200), (8064, 512, 2048), (130, 128, 520), (65, 36, 40)])
def test_fp16x2_rows_of_very_different_scales(device, M, D, F, layout):
    """the two-plane fp16 GEMM scales every A row by a power of two taken from the row's maximum (found
    by the pass that forms the planes): a feed-forward pair on rows whose scales differ by 10 orders of
    magnitude, ragged M / N / K (K not a multiple of the 32-element K step), every row against ITS
    scale -- small rows must be as good as large ones"""
    from aps_amd import nn_ops
    saved = nn_ops.SPLIT_MODE, nn_ops.SPLIT_LAYOUT
    nn_ops.SPLIT_MODE, nn_ops.SPLIT_LAYOUT = "1", layout
    try:
        g = torch.Generator().manual_seed(M + F)
        scale = torch.exp(torch.empty(M, 1).uniform_(-11.5, 11.5, generator=g))  # 1e-5 .. 1e5 per row
        x = torch.randn(M, D, generator=g) * scale
        w1 = torch.randn(F, D, generator=g) / D**0.5
        b1 = torch.randn(F, generator=g) * 0.1
        w2 = torch.randn(D, F, generator=g) / F**0.5
        p1 = torch.nn.Parameter(w1.to(device), requires_grad=False)
        p2 = torch.nn.Parameter(w2.to(device), requires_grad=False)
        xd, b1d = x.to(device), b1.to(device)
        h_ref = torch.relu(x.double() @ w1.double().T + b1.double())
        y_ref = h_ref @ w2.double().T + x.double()

        def rel_rows(out, ref):
            return ((out.double().cpu() - ref).abs().amax(1) / ref.abs().amax(1).clamp_min(1e-30)).max().item()

        h = nn_ops.linear(xd, p1, b1d, relu=True, chain=True)   # (`chain` is accepted and ignored)
        y = nn_ops.linear(h, p2, residual=xd)
        assert rel_rows(h, h_ref) < 2e-6 and rel_rows(y, y_ref) < 4e-6
        # a column slice of a wider buffer (row pitch != K) and a second call on the same input
        wide = torch.cat([xd, xd], 1)
        assert torch.equal(nn_ops.linear(wide[:, :D], p1, b1d, relu=True), h)
        assert torch.equal(nn_ops.linear(xd, p1, b1d, relu=True), h)
    finally:
        nn_ops.SPLIT_MODE, nn_ops.SPLIT_LAYOUT = saved


@pytest.mark.parametrize("T,win", [(100, (4, 2, 1)), (128, (1, 5, 0)), (40, (8, 0, 0))])
def test_attention_window_abs(device, T, win):
    """context window without relative terms (T <= 64 and 64 < T <= 128 MFMA kernels)"""
    from aps_amd.nn_ops import attention_core
    from oracle import encoder_oracle as eo
    g = torch.Generator().manual_seed(T)
    N, H, dh = 2, 2, 64
    D = H * dh
    qkv = torch.randn(N, T, 3 * D, generator=g)
    lens = torch.tensor([T, T - 11])
    q, k, v = [m.reshape(N, T, H, dh).permute(0, 2, 1, 3).double() for m in qkv.chunk(3, -1)]
    s = q @ k.transpose(-1, -2) / dh**0.5 + eo.context_mask(T, *win).double()[None, None]
    s = s.masked_fill((torch.arange(T)[None] >= lens[:, None])[:, None, None, :], float("-inf"))
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(N, T, D)
    out = attention_core(qkv.to(device), H, lens.to(device), chunk_size=win[0], lctx=win[1],
                         rctx=win[2])
    valid = ~torch.isnan(ref).any(-1)
    assert_close(out.cpu()[valid], ref[valid], 1e-5, f"window T={T}")


def test_causal_conformer_layer(device):
    """casual_conv1d (impl.py:446-505): K - 1 frames of left context, padded frames = glu(bias)"""
    from aps_amd.asr.transformer.impl import ApsConformerEncoderLayer, ApsMultiheadAttention
    g = golden("cfmr_layer_causal")
    layer = ApsConformerEncoderLayer(64, ApsMultiheadAttention(64, 2, dropout=0), feedforward_dim=96,
                                     dropout=0, kernel_size=5, casual_conv1d=True)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    missing, unexpected = layer.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing)
    layer = layer.eval().to(device)
    src = g["src"].to(device)
    pad = (torch.arange(21)[None, :] >= g["lens"][:, None]).to(device)
    assert_close(layer.conv(src), g["conv"], TOL, "causal conv module")
    assert_close(layer(src, src_key_padding_mask=pad), g["out"], TOL, "causal conformer layer")


@pytest.mark.parametrize("kind", ["abs", "rel"])
def test_layer_with_arbitrary_additive_mask(device, kind):
    """the layers' `src_mask` argument as a plain T x T additive tensor (not a context window):
    random -inf pattern plus a finite bias, against the oracle layer"""
    from aps_amd.asr.transformer.impl import TransformerEncoderLayers
    from oracle import encoder_oracle as eo
    torch.manual_seed(33)
    T, N, D, H = 37, 2, 64, 2
    layer = TransformerEncoderLayers[f"xfmr_{kind}"](att_dim=D, nhead=H, feedforward_dim=96,
                                                     att_dropout=0, ffn_dropout=0).eval()
    g = torch.Generator().manual_seed(34)
    src = torch.randn(T, N, D, generator=g)
    mask = 0.5 * torch.randn(T, T, generator=g)
    mask[torch.rand(T, T, generator=g) < 0.3] = float("-inf")
    mask.fill_diagonal_(0.0)  # every query keeps one visible key
    lens = torch.tensor([T, 29])
    pad = torch.arange(T)[None, :] >= lens[:, None]
    mask[:, 29:] = float("-inf")
    mask[torch.arange(T), torch.arange(T).clamp(max=28)] = 0.0
    rel = torch.randn(2 * T - 1, D // H, generator=g) if kind == "rel" else None
    sd = {k: v.detach() for k, v in layer.state_dict().items()}
    ref = eo.encoder_layer(sd, "", src, pad, H, False, rel, kind, mask)
    layer = layer.to(device)
    out = layer(src.to(device), inj_pose=None if rel is None else rel.to(device),
                src_mask=mask.to(device), src_key_padding_mask=pad.to(device))
    assert_close(out, ref, TOL, f"xfmr_{kind} layer with an additive mask")


def test_gelu_feedforward(device):
    """activation = "gelu" (transformer/utils.py:113-123): erf-form GELU in the GEMM epilogue"""
    from aps_amd.asr.transformer.impl import TransformerEncoderLayers
    from aps_amd.nn_ops import linear
    torch.manual_seed(35)
    x = torch.randn(50, 96)
    w, b = torch.randn(64, 96) / 96**0.5, torch.randn(64)
    ref = torch.nn.functional.gelu(x.double() @ w.double().t() + b.double())
    out = linear(x.to(device), w.to(device), b.to(device), act="gelu")
    assert_close(out, ref, 1e-5, "gelu epilogue")
    layer = TransformerEncoderLayers["xfmr_abs"](att_dim=64, nhead=2, feedforward_dim=96,
                                                 att_dropout=0, ffn_dropout=0, activation="gelu").eval()
    src = torch.randn(17, 2, 64)
    sd = {k: v.detach() for k, v in layer.state_dict().items()}
    import torch.nn.functional as F
    from oracle import encoder_oracle as eo
    att = src + eo.self_attention(sd, "self_attn.", src, None, 2)
    h = F.layer_norm(att, (64,), sd["norm1.weight"], sd["norm1.bias"])
    ff = F.linear(F.gelu(F.linear(h, sd["feedforward.0.weight"], sd["feedforward.0.bias"])),
                  sd["feedforward.3.weight"], sd["feedforward.3.bias"])
    ref = F.layer_norm(h + ff, (64,), sd["norm2.weight"], sd["norm2.bias"])
    assert_close(layer.to(device)(src.to(device)), ref, TOL, "gelu transformer layer")


@pytest.mark.parametrize("act", ["swish", "relu", "gelu", "none"])
def test_glu_dwconv_activations(device, act):
    """the activation after the conformer convolution module's BatchNorm (impl.py:478-489) runs in
    the GLU + depthwise kernel for every activation the reference accepts"""
    import torch.nn.functional as F
    from aps_amd.nn_ops import glu_dwconv
    torch.manual_seed(41)
    N, T, D, K = 3, 37, 48, 7
    x = torch.randn(N, T, 2 * D)
    w, b = torch.randn(D, 1, K) * 0.3, torch.randn(D) * 0.1
    scale, shift = 0.5 + torch.rand(D), 0.1 * torch.randn(D)
    g = F.glu(x.double(), dim=-1).transpose(1, 2)                      # N x D x T
    y = F.conv1d(g, w.double(), b.double(), padding=(K - 1) // 2, groups=D).transpose(1, 2)
    y = y * scale.double() + shift.double()
    ref = {"swish": lambda v: v * torch.sigmoid(v), "relu": torch.relu, "gelu": F.gelu,
           "none": lambda v: v}[act](y)
    out = glu_dwconv(x.to(device), w.to(device), b.to(device), scale.to(device), shift.to(device),
                     act=act)
    assert_close(out, ref, 1e-5, f"glu + dwconv + {act}")
    with pytest.raises(ValueError):
        glu_dwconv(x.to(device), w.to(device), b.to(device), None, None, act="elu")


@pytest.mark.parametrize("N,H,bidir", [(128, 512, False), (16, 512, False), (40, 256, True), (64, 128, False)])
def test_lstm_team_form_paths_agree(device, N, H, bidir):
    """the team form of the layer kernel (hand-off inside one XCD when the 32 workgroups of a team
    report the same XCC id) against its own placement-independent path (APS_LSTM_DEBUG=16: sc1 stores
    although the placement would allow the short path) and against the spread form
    (APS_LSTM_TEAM is read once per process, so that one is covered by the float64 reference):
    identical words, no expired waits; the placement decisions of this run are printed"""
    import os
    from aps_amd import nn_ops
    torch.manual_seed(N + H)
    T, D = 37, 48
    rnn = torch.nn.LSTM(D, H, 1, batch_first=True, bidirectional=bidir).eval().to(device)
    x = torch.randn(N, T, D, device=device)
    lens = torch.randint(1, T + 1, (N,), device=device)
    lens[0] = T
    st = nn_ops._lstm_status(x.device)
    outs = {}
    for dbg in ("32", "48"):  # 32: count the decisions; 48: + forced placement-independent stores
        os.environ["APS_LSTM_DEBUG"] = dbg
        st.ws[1:4] = 0
        nn_ops.LSTM_CHECK = True
        try:
            outs[dbg] = nn_ops.lstm_forward(rnn, x, lens)
        finally:
            nn_ops.LSTM_CHECK = False
            del os.environ["APS_LSTM_DEBUG"]
        word = int(st.ws[1].item()) & 0xffffffff
        print(f"[lstm team] N={N} H={H} bidir={bidir} debug={dbg}: {word & 0xffff} workgroups saw their "
              f"team on one XCD, {word >> 16} did not")
    assert torch.equal(outs["32"], outs["48"])
    ref = torch.nn.LSTM(D, H, 1, batch_first=True, bidirectional=bidir).double()
    ref.load_state_dict({k: v.double().cpu() for k, v in rnn.state_dict().items()})
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    packed = pack_padded_sequence(x.double().cpu(), lens.cpu().tolist(), batch_first=True,
                                  enforce_sorted=False)
    want, _ = pad_packed_sequence(ref(packed)[0], batch_first=True, total_length=T)
    assert_close(outs["32"], want, 1e-5, "team form vs float64")

