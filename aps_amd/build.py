"""
Ahead-of-time build of the HIP extension (libaps_amd.so) for gfx950.  In-tree, so the binary
travels with the repo snapshot; no JIT at import time.
"""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libaps_amd.so")
SOURCES = ["aps_core.hip", "stft.hip", "feats.hip", "mvdr.hip", "nn.hip", "lstm.hip", "context.hip", "conv.hip", "decoder.hip", "spatial.hip", "augment.hip"]
HEADERS = ["common.h", "fft_core.h", "twiddles.h", os.path.join("..", "..", "include", "aps_amd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-mcode-object-version=5",
    "-Wno-unused-value"
]


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > built for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source into libaps_amd.so (hipcc cross-compiles without a GPU)."""
    if not force and not stale():
        return LIB
    cmd = [HIPCC] + FLAGS + SOURCES + ["-o", LIB]
    if verbose:
        print("[aps_amd.build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
