"""
Loading a trained checkpoint of the reference into this build -- the surface of
aps/eval/wrapper.py:16-86: a checkpoint directory holds `train.yaml` (the recipe: `nnet`,
`nnet_conf`, `asr_transform` / `enh_transform`) and `<tag>.pt.tar` (`model_state`, `epoch`).  The
same recipe builds the same-named network here (aps_amd.libs registries), and because the modules
keep the reference's parameter names the state dict loads strictly.
"""
import pathlib
from typing import Dict

import torch as th
import yaml

from aps_amd.libs import aps_nnet, aps_transform


def load_checkpoint(cpt_dir: str, cpt_tag: str = "best", nnet_cls: object = None) -> Dict:
    """-> {"epoch", "accept_raw", "nnet", "conf"} (wrapper.py:16-56)"""
    cpt_dir = pathlib.Path(cpt_dir)
    cpt = th.load(cpt_dir / f"{cpt_tag}.pt.tar", map_location="cpu")
    with open(cpt_dir / "train.yaml", "r") as f:
        conf = yaml.full_load(f)
    if nnet_cls is None:
        nnet_cls = aps_nnet(conf["nnet"])
    transforms, accept_raw = {}, False
    if "asr_transform" in conf:
        transforms["asr_transform"] = aps_transform("asr")(**conf["asr_transform"])
        # features instead of waveforms if the chain has no STFT layer
        accept_raw = transforms["asr_transform"].spectra_index != -1
    if "enh_transform" in conf:
        transforms["enh_transform"] = aps_transform("enh")(**conf["enh_transform"])
        accept_raw = True
    nnet = nnet_cls(**transforms, **conf["nnet_conf"])
    nnet.load_state_dict(cpt["model_state"])
    return {"epoch": cpt["epoch"], "accept_raw": accept_raw, "nnet": nnet, "conf": conf}


class NnetEvaluator(object):
    """model + recipe of a checkpoint directory, in eval mode on the chosen device
    (wrapper.py:59-86).  There is no CPU execution path in this build: device_id < 0 keeps the
    parameters on the host (inspection, re-saving); running the network needs device_id >= 0."""

    def __init__(self, cpt_dir: str, cpt_tag: str = "best", device_id: int = -1) -> None:
        stats = load_checkpoint(cpt_dir, cpt_tag=cpt_tag)
        self.conf = stats["conf"]
        self.nnet = stats["nnet"]
        self.accept_raw = stats["accept_raw"]
        self.epoch = stats["epoch"]
        if device_id < 0:
            self.device = th.device("cpu")
        else:
            self.device = th.device(f"cuda:{device_id:d}")
            self.nnet.to(self.device)
        self.nnet.eval()

    def run(self, *args, **kwargs):
        raise NotImplementedError
