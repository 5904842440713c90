"""
TEST INFRASTRUCTURE -- CPU restatement of the reference's DCCRN forward (aps/sse/bss/dccrn.py:139-349
with the blocks of aps/sse/enh/dcunet.py:24-275) as functional torch-CPU ops on a state_dict, in
the reference's own layout (N x C x 2F x T, four real convolutions per complex layer).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  Pinned by
tests/test_oracle_encoder.py against the fixtures dccrn_shared / dccrn_split / dccrn_cat_causal /
dccrn_real / dccrn_real_cat recorded from the reference's module.
"""
import torch
import torch.nn.functional as F

from oracle import aps_oracle as ao
from oracle.joint_oracle import lstm_stack


def parse_1d(s):
    return [int(v) for v in s.split(",")]


def parse_2d(s):
    return [parse_1d(t) for t in s.split(";")]


def cplx_conv(sd, p, x, stride, padding, transposed=False, output_padding=(0, 0)):
    """ComplexConv2d / ComplexConvTranspose2d (dcunet.py:24-68) on real | imag stacked along F"""
    xr, xi = torch.chunk(x, 2, -2)
    if transposed:
        def op(part, v):
            return F.conv_transpose2d(v, sd[p + part + ".weight"], sd[p + part + ".bias"], stride,
                                      padding, output_padding)
    else:
        def op(part, v):
            return F.conv2d(v, sd[p + part + ".weight"], sd[p + part + ".bias"], stride, padding)
    yr = op("real", xr) - op("imag", xi)
    yi = op("imag", xr) + op("real", xi)
    return torch.cat([yr, yi], -2)


def cplx_bn(sd, p, x, train=False):
    """ComplexBatchNorm2d (dcunet.py:71-87); train: batch statistics, the running statistics in
    `sd` updated in place with nn.BatchNorm2d's default momentum 0.1"""
    xr, xi = torch.chunk(x, 2, -2)

    def bn(part, v):
        q = p + part
        return F.batch_norm(v, sd[q + ".running_mean"], sd[q + ".running_var"], sd[q + ".weight"],
                            sd[q + ".bias"], train, 0.1 if train else 0.0, 1e-5)
    return torch.cat([bn("real_bn", xr), bn("imag_bn", xi)], -2)


def lstmp(sd, p, inp, num_layers):
    """LSTMP (dccrn.py:16-51): N x T x C x F -> N x T x C x F"""
    N, T, C, _ = inp.shape
    out = lstm_stack(sd, p + "lstm.", inp.reshape(N, T, -1), num_layers)
    return F.linear(out, sd[p + "proj.weight"]).view(N, T, C, -1)


def real_conv(sd, p, x, stride, padding, transposed=False, output_padding=(0, 0)):
    """nn.Conv2d / nn.ConvTranspose2d block 0 of the real-valued network"""
    if transposed:
        return F.conv_transpose2d(x, sd[p[:-1] + ".weight"], sd[p[:-1] + ".bias"], stride, padding,
                                  output_padding)
    return F.conv2d(x, sd[p[:-1] + ".weight"], sd[p[:-1] + ".bias"], stride, padding)


def real_bn(sd, p, x, train=False):
    q = p[:-1]
    return F.batch_norm(x, sd[q + ".running_mean"], sd[q + ".running_var"], sd[q + ".weight"],
                        sd[q + ".bias"], train, 0.1 if train else 0.0, 1e-5)


def dccrn_forward(sd, mix, *, K, S, P, O, num_spks=2, rnn_layers=2, share_decoder=True,
                  non_linear="tanh", frame_len=512, frame_hop=256, window="sqrthann", mode="time",
                  eps=ao.EPSILON, cplx=True, connection="sum", causal_conv=False, train=False):
    """mix N x S -> list over speakers of N x S (mode "time") or masks ("freq": N x F x T x 2
    complex, N x F x T real for cplx = False).  train: the module in train() mode (BatchNorm with
    batch statistics; `sd`'s running statistics are updated in place)"""
    K, S, P, O = parse_2d(K), parse_2d(S), parse_1d(P), parse_1d(O)
    conv, bn = (cplx_conv, cplx_bn) if cplx else (real_conv, real_bn)
    packed = ao.stft(mix, frame_len, frame_hop, window)  # N x F x T x 2
    sr, si = packed[..., 0], packed[..., 1]
    if cplx:
        x = torch.cat([sr, si], -2)[:, None]  # N x 1 x 2F x T
    else:
        x = ((sr**2 + si**2 + eps)**0.5)[:, None]  # dccrn.py:259
    enc_h = []
    L = len(K)
    norm_idx = 2 if causal_conv else 1  # CasualTruncated sits at index 1 (dcunet.py:130-133)

    def time_pad(k):
        return k - 1 if causal_conv else (k - 1) // 2

    def truncate(v, k):  # CasualTruncated (dcunet.py:90-100)
        return v[..., :-time_pad(k)] if causal_conv else v

    for i in range(L):
        p = f"encoder.layers.{i}.block."
        x = truncate(conv(sd, p + "0.", x, tuple(S[i]), (P[i], time_pad(K[i][1]))), K[i][1])
        x = F.leaky_relu(bn(sd, p + f"{norm_idx}.", x, train), 0.01)
        if i + 1 != L:
            enc_h.append(x)
    h = torch.einsum("ncft->ntcf", x)
    if cplx:  # complex LSTM (dccrn.py:54-94)
        hr, hi = torch.chunk(h, 2, -1)
        out_r = lstmp(sd, "rnn.lstm.real.", hr, rnn_layers) - lstmp(sd, "rnn.lstm.imag.", hi, rnn_layers)
        out_i = lstmp(sd, "rnn.lstm.real.", hi, rnn_layers) + lstmp(sd, "rnn.lstm.imag.", hr, rnn_layers)
        out_h = torch.cat([out_r, out_i], -1)
    else:
        out_h = lstmp(sd, "rnn.lstm.", h, rnn_layers)
    out_h = torch.einsum("ntcf->ncft", out_h)
    x = x + out_h if connection == "sum" else torch.cat([out_h, x], 1)  # dccrn.py:268-271
    enc_h = enc_h[::-1]
    Kd, Sd, Pd, Od = K[::-1], S[::-1], P[::-1], O[::-1]

    def decode(d, x):
        for i in range(L):
            p = f"decoder.{d}.layers.{i}.block."
            if i == 0:
                inp = x
            elif connection == "sum":
                inp = x + enc_h[i - 1]
            else:
                inp = torch.cat([x, enc_h[i - 1]], 1)
            x = conv(sd, p + "0.", inp, tuple(Sd[i]), (Pd[i], Kd[i][1] - 1 - time_pad(Kd[i][1])),
                     True, (Od[i], 0))
            x = truncate(x, Kd[i][1])
            if i != L - 1:
                x = F.leaky_relu(bn(sd, p + f"{norm_idx}.", x, train), 0.01)
        return x

    if share_decoder:
        masks = decode(0, x)
    else:
        masks = torch.cat([decode(d, x) for d in range(num_spks)], 1)
    nl = {"none": lambda v: v, "relu": torch.relu, "tanh": torch.tanh, "softplus": F.softplus,
          "sigmoid": torch.sigmoid}[non_linear]
    outs = []
    for s in range(num_spks):
        if cplx:
            mr, mi = torch.chunk(masks[:, s], 2, -2)
            m_abs = (mr**2 + mi**2 + eps)**0.5
            m_mag = nl(m_abs)
            mr, mi = m_mag * mr / m_abs, m_mag * mi / m_abs
            if mode == "freq":
                outs.append(torch.stack([mr, mi], -1))
                continue
            spec = torch.stack([sr * mr - si * mi, sr * mi + si * mr], -1)
        else:
            m = nl(masks[:, s])
            if mode == "freq":
                outs.append(m)
                continue
            spec = torch.stack([sr * m, si * m], -1)
        outs.append(ao.istft(spec, frame_len, frame_hop, window))
    return outs
