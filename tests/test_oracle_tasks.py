"""
The task-side oracle (oracle/task_oracle.py: spectral-approximation objectives and the
maximum-likelihood objective) against fixtures recorded from the reference's own
task classes (tests/golden/make_golden.py:gen_tasks).
"""
import json

import torch

from oracle import task_oracle as to
from tests.conftest import golden


def test_freq_sa_losses_match_the_reference():
    g = golden("task_freq_sa")
    cfg = g.cfg
    masks, refs = [g["mask0"], g["mask1"]], [g["ref0"], g["ref1"]]
    for tag, kw in cfg.items():
        kw = dict(kw)
        mel = None
        if tag.startswith("mel"):
            mel = g["mel"]
            assert mel.shape == (kw.pop("num_mels"), 257)
            kw.pop("mel_scale")
        loss = to.freq_sa_loss(masks, g["mix"], refs, mel=mel, **kw)
        want = g["loss." + tag]
        assert abs(loss.item() - want.item()) <= 1e-5 * abs(want.item()), (tag, loss, want)


def test_ml_objective_matches_the_reference():
    g = golden("task_enh_ml")
    xr, xi = g["obs_r"].transpose(1, 2), g["obs_i"].transpose(1, 2)
    lp = to.ml_log_pdf(g["ms"].transpose(-1, -2), xr, xi)
    assert lp.shape == g["log_pdf"].shape
    scale = g["log_pdf"].abs().max()
    assert (lp - g["log_pdf"]).abs().max() <= 1e-5 * scale
    loss = to.ml_loss(g["ms"], g["obs_r"], g["obs_i"])
    assert abs(loss.item() - g["loss"].item()) <= 1e-5 * abs(g["loss"].item())
