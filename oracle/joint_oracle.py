"""
TEST INFRASTRUCTURE -- CPU restatement of the joint front-end data path of the reference
(`EnhASRBase.forward`, aps/asr/enh_att.py:83-95, with the encoder-side model of
aps/asr/ctc.py:113-134 as the ASR), composed from the per-stage restatements in aps_oracle.py /
encoder_oracle.py plus the RNN mask estimator.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  Pinned by tests/test_oracle_encoder.py against the fixture
joint_mvdr_cfmr recorded from the reference's own modules.
"""
import torch
import torch.nn.functional as F

from oracle import aps_oracle as ao
from oracle import encoder_oracle as eo


def lstm_stack(sd, prefix, x, num_layers, lens=None):
    """unidirectional nn.LSTM (batch_first), gate order i, f, g, o; x N x T x D -> N x T x H.
    With lengths the reference packs the batch (aps/asr/base/component.py:26-55): outputs past an
    utterance's length are zero."""
    N, T, _ = x.shape
    for layer in range(num_layers):
        w_ih, w_hh = sd[f"{prefix}weight_ih_l{layer}"], sd[f"{prefix}weight_hh_l{layer}"]
        b = sd[f"{prefix}bias_ih_l{layer}"] + sd[f"{prefix}bias_hh_l{layer}"]
        H = w_hh.shape[1]
        h, c = x.new_zeros(N, H), x.new_zeros(N, H)
        pre = F.linear(x, w_ih, b)  # N x T x 4H
        outs = []
        for t in range(T):
            i, f, g, o = (pre[:, t] + F.linear(h, w_hh)).chunk(4, -1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        x = torch.stack(outs, 1)
        if lens is not None:
            x = x * (torch.arange(T)[None, :, None] < lens[:, None, None])
    return x


def mask_net(sd, prefix, feats, lens, num_layers):
    """PyTorchRNNEncoder (aps/asr/base/encoder.py:87-184): Linear + ReLU -> LSTM -> Linear ->
    sigmoid; feats N x T x D -> N x T x 2F"""
    h = torch.relu(F.linear(feats, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"]))
    h = lstm_stack(sd, prefix + "impl.", h, num_layers, lens)
    return torch.sigmoid(F.linear(h, sd[prefix + "outp.weight"], sd[prefix + "outp.bias"]))


def joint_forward(sd, wav, lens, *, frame_len=512, frame_hop=256, window="sqrthann",
                  ipd_index="0,1;0,2;0,3", num_mels=80, rnn_layers=2, enc_layers=12, nhead=8,
                  lradius=256, rradius=256, kernel_size=15, arch="cfmr", pose="rel",
                  pre_norm=True):
    """wav N x C x S, lens N (samples) | None -> dict(enh, asr_feats, enc_out, enc_ctc, enc_len);
    sd = state_dict of ModuleDict(enh_transform, asr_transform, enh_net, asr)"""
    packed = ao.stft(wav, frame_len, frame_hop, window)  # N x C x F x T x 2
    n = None
    if lens is not None:
        n = torch.tensor([ao.num_frames(int(s), frame_len, frame_hop, False) for s in lens])
    feats = ao.enh_features(packed, "spectrogram-log-cmvn-ipd", ipd_index, cos_ipd=True)
    mask = mask_net(sd, "enh_net.mask_net.", feats, n, rnn_layers)
    mask_s, mask_n = mask.chunk(2, -1)
    att = tuple(sd["enh_net.mvdr_net.ref." + k]
                for k in ("proj.weight", "proj.bias", "gvec.weight", "gvec.bias"))
    yr, yi, _ = ao.mvdr_forward(mask_s, packed[..., 0], packed[..., 1], att, mask_n=mask_n, x_len=n)
    mel_w = ao.mel_weights(frame_len, num_mels=num_mels)
    asr_feats = ao.abs_mel_log_cmvn(yr, yi, mel_w)
    enc_sd = {k[len("asr.encoder."):]: v for k, v in sd.items() if k.startswith("asr.encoder.")}
    enc_out, enc_len = eo.generic_encoder(enc_sd, asr_feats, n, arch, pose, enc_layers, nhead,
                                          lradius=lradius, rradius=rradius,
                                          kernel_size=kernel_size, pre_norm=pre_norm)
    enc_ctc = enc_out
    if "asr.ctc.weight" in sd:
        enc_ctc = F.linear(enc_out, sd["asr.ctc.weight"], sd["asr.ctc.bias"])
    return {"enh": (yr, yi), "mask": mask, "asr_feats": asr_feats, "enc_out": enc_out,
            "enc_ctc": enc_ctc, "enc_len": enc_len, "num_frames": n}
