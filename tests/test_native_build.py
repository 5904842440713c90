"""CPU: the C++ core math (shared by host and device) against a double precision DFT."""
import os
import subprocess

from tests.conftest import ROOT


def test_fft_core_host_emulation(tmp_path):
    exe = str(tmp_path / "test_fft_core")
    src = os.path.join(ROOT, "tests", "csrc", "test_fft_core.cc")
    subprocess.run(["g++", "-O2", "-std=c++17", src, "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "wave fft512" in out
