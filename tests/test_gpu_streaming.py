"""
GPU parity of the frame-by-frame (i)STFT layers (aps_amd/transform/streaming.py on the STFT /
iSTFT kernels) against activations recorded from the reference's StreamingSTFT / StreamingiSTFT
(512-point, 400 -> 512 librosa-padded and normalized, 400-point kaldi frames), including the
step / reset / flush protocol, and against the block STFT (the reference's own
test_streaming_stft, tests/python/test_transform.py:40-65).  Framing exact, values 1e-4 of scale.
"""
import pytest
import torch

from tests.conftest import assert_close, golden

pytestmark = pytest.mark.gpu
TOL = 1e-4
CASES = {
    "streaming_512": dict(frame_len=512, frame_hop=256, window="sqrthann", mode="librosa"),
    "streaming_400_librosa": dict(frame_len=400, frame_hop=160, window="hamm", mode="librosa",
                                  normalized=True),
    "streaming_400_kaldi": dict(frame_len=400, frame_hop=160, window="hann", mode="kaldi",
                                round_pow_of_two=False),
}


@pytest.fixture(autouse=True)
def _inference_mode():
    with torch.no_grad():
        yield


def window_energy(w, hop, T):
    """overlap-added w^2 of T frames: where it is tiny (the first / last samples under a window
    that starts at zero) the normalised output num / (energy + eps) amplifies fp32 rounding of the
    inverse FFT by up to 1 / (2 sqrt(eps)) = 1450 -- those samples get their own, looser bound"""
    W = w.shape[0]
    den = torch.zeros((T - 1) * hop + W)
    for t in range(T):
        den[t * hop:t * hop + W] += w.cpu()**2
    return den


def assert_wav_close(got, ref, den, tol, what):
    solid = den >= 1e-3
    scale = ref.abs().max().item()
    err = (got.cpu() - ref).abs()
    assert err[..., solid].max().item() <= tol * scale, f"{what}: {err[..., solid].max().item():.3e}"
    if (~solid).any():
        assert err[..., ~solid].max().item() <= 2e-3 * scale, \
            f"{what} (ill-conditioned edge samples): {err[..., ~solid].max().item():.3e}"


@pytest.mark.parametrize("tag", sorted(CASES))
def test_streaming_golden(device, tag):
    from aps_amd.transform.streaming import StreamingSTFT, StreamingiSTFT
    g, cfg = golden(tag), CASES[tag]
    fwd, inv = StreamingSTFT(**cfg).to(device), StreamingiSTFT(**cfg).to(device)
    W, hop = fwd.win_length, cfg["frame_hop"]
    assert W == int(g["win_length"])
    assert_close(fwd.w, g["w"], 1e-6, "window parameter")
    wav = g["wav"].to(device)
    packed = fwd(wav)
    assert packed.shape == g["packed"].shape
    assert_close(packed, g["packed"], TOL, "forward")
    polar = fwd(wav, return_polar=True)
    assert_close(polar[..., 0], g["polar"][..., 0], TOL, "magnitude")
    assert_close(fwd.step(wav[:, :W]), g["first"], TOL, "step 0")
    third = fwd.step(wav[:, 2 * hop:2 * hop + W], return_polar=True)
    assert_close(third[..., 0], g["third_polar"][..., 0], TOL, "polar step 2")
    with pytest.raises(RuntimeError):
        fwd.step(wav[:, :W - 1])
    ref_packed = g["packed"].to(device)
    rebuilt = inv(ref_packed)
    assert rebuilt.shape == g["rebuilt"].shape
    den = window_energy(g["w"], hop, ref_packed.shape[-2])
    assert_wav_close(rebuilt, g["rebuilt"], den, TOL, "inverse")
    assert_wav_close(inv(g["polar"].to(device), return_polar=True), g["rebuilt_polar"], den, TOL,
                     "inverse from polar")
    inv.reset()
    chunks = []
    for k in range(3):
        chunks.append(inv.step(ref_packed[..., k, :].clone()))
        assert chunks[-1].shape == (2, hop)
    chunks.append(inv.flush())
    want = torch.cat([g["step0"], g["step1"], g["step2"], g["tail"]], -1)
    assert_wav_close(torch.cat(chunks, -1), want, window_energy(g["w"], hop, 3), TOL,
                     "inverse steps + flush")


@pytest.mark.parametrize("frame_len,frame_hop", [(512, 256), (256, 128), (400, 160)])
@pytest.mark.parametrize("window", ["hamm", "sqrthann"])
def test_streaming_equals_block(device, frame_len, frame_hop, window):
    """the reference's own check: the streaming layers give what STFT / iSTFT give"""
    from aps_amd.transform import STFT, iSTFT
    from aps_amd.transform.streaming import StreamingSTFT, StreamingiSTFT
    cfg = dict(frame_len=frame_len, frame_hop=frame_hop, window=window, center=False,
               round_pow_of_two=True, mode="librosa")
    torch.manual_seed(frame_len + frame_hop)
    wav = 0.2 * torch.randn(1, 16000, device=device)
    packed = STFT(**cfg).to(device)(wav)
    streamed = StreamingSTFT(**cfg).to(device)(wav)
    assert torch.equal(packed, streamed)
    rebuilt = iSTFT(**cfg).to(device)(packed)
    streamer = StreamingiSTFT(**cfg).to(device)
    assert torch.equal(rebuilt, streamer(packed))
    # and frame by frame
    streamer.reset()
    chunks = [streamer.step(packed[..., t, :].clone()) for t in range(packed.shape[-2])]
    stepped = torch.cat(chunks + [streamer.flush()], -1)
    assert stepped.shape == rebuilt.shape
    den = window_energy(streamer.w.data, frame_hop, packed.shape[-2])
    assert_wav_close(stepped, rebuilt.cpu(), den, 1e-5, "step by step")
