import sys; sys.path.insert(0, "/root/repo")
import torch
from tests.conftest import golden
from aps_amd.transform import AsrTransform
from oracle import aps_oracle as orc
g = golden("asr_spectrogram_cmvn_allband")
cfg = dict(g.cfg); x = g["in_egs1"]
kw = dict(frame_len=512, frame_hop=256, window_name="hann", pre_emphasis=0, center=True)
with torch.no_grad():
    for feats in ("spectrogram", "spectrogram-log", "spectrogram-log-cmvn"):
        c = dict(cfg, feats=feats)
        t = AsrTransform(**c).cuda()
        out, _ = t(x.cuda(), None)
        truth = orc.asr_features(x, feats=feats, dtype=torch.float64, norm_per_band=False, **kw)
        ref32 = orc.asr_features(x, feats=feats, dtype=torch.float32, norm_per_band=False, **kw)
        sc = truth.abs().max()
        e_got = (out.cpu().double() - truth).abs(); e_ref = (ref32.double() - truth).abs()
        print(feats, "scale", sc.item(), "got", (e_got.max() / sc).item(), "ref32", (e_ref.max() / sc).item())
        if feats == "spectrogram":
            mag = truth
            i = e_got.flatten().argmax(); print(" worst at", i.item(), "mag", mag.flatten()[i].item(), "err", e_got.flatten()[i].item())
            small = mag < 1e-4 * mag.max()
            print(" abs err on small bins: got", e_got[small].max().item(), "ref", e_ref[small].max().item(), "n small", small.sum().item())
