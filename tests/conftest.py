import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden(dict):
    """npz fixture -> dict of torch tensors (+ .cfg dict when the fixture carries one)"""

    def __init__(self, name):
        data = np.load(os.path.join(GOLDEN, name + ".npz"))
        super().__init__()
        self.cfg = None
        for k in data.files:
            if k == "cfg":
                self.cfg = json.loads(str(data[k]))
            else:
                self[k] = torch.from_numpy(np.asarray(data[k]))


def golden(name):
    return Golden(name)


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def rel_err(got, ref):
    """max |got - ref| relative to the scale (max |ref|) of the reference tensor: the
    north-star tolerance (1e-4 rel) is defined against the activation scale, because FFT vs
    dense-DFT summation order makes per-element relative error meaningless at near-zero bins."""
    got = torch.as_tensor(got).detach().cpu().double()
    ref = torch.as_tensor(ref).detach().cpu().double()
    assert got.shape == ref.shape, f"shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    scale = ref.abs().max().clamp_min(1e-30)
    return ((got - ref).abs().max() / scale).item()


def assert_close(got, ref, tol=1e-4, what=""):
    err = rel_err(got, ref)
    assert err <= tol, f"{what}: scaled max error {err:.3e} > {tol:.1e}"


@pytest.fixture(scope="session")
def device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def assert_as_accurate(got, ref, truth, tol=1e-4, slack=3.0, what=""):
    """For ill-conditioned outputs (log of near-zero bins, phase of near-zero bins) the fp32
    reference itself sits a measurable distance from exact arithmetic.  `truth` is the float64
    evaluation of the same formula on the same fp32 inputs; the HIP result must be within `tol`
    of it, or as close to it as the reference is (x slack: both errors are maxima over a few
    near-zero bins, i.e. the same order but not the same bins)."""
    e_ref = rel_err(ref, truth)
    e_got = rel_err(got, truth)
    bound = max(tol, slack * e_ref)
    assert e_got <= bound, (f"{what}: error vs float64 truth {e_got:.3e} > {bound:.3e} "
                            f"(reference's own error {e_ref:.3e})")
