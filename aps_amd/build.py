"""
Ahead-of-time build of the HIP extension (libaps_amd.so) for gfx950.  In-tree, so the binary
travels with the repo snapshot; no JIT at import time.  Every .hip source is compiled to its own
object (in parallel, only when it or a header changed) and the objects are linked into one library.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libaps_amd.so")
OBJ_DIR = os.path.join(CSRC, "_obj")
SOURCES = ["aps_core.hip", "stft.hip", "feats.hip", "mvdr.hip", "nn.hip", "lstm.hip", "context.hip",
           "conv.hip", "decoder.hip", "spatial.hip", "augment.hip", "grad.hip",
           "gemm_split.hip", "gemm_fp16x2.hip", "gemm_panel.hip", "gemm_tn.hip", "conformer_mega.hip"]
HEADERS = ["common.h", "fft_core.h", "twiddles.h", "conv_core.h", "grad_core.h", "grad_api.inc",
           os.path.join("..", "..", "include", "aps_amd.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# No packed-fp32 VALU instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in any kernel: round 3
# traced the cross-stream disturbance of round 2 (isolated wrong values in lanes 48-63 of an STFT
# wavefront while a particular MFMA kernel build of ANOTHER stream shared its CU) to them -- the same
# STFT source compiled without them shows 0 differing replays where the packed build shows 26-30 in
# 12 rounds (DESIGN.md "co-residency", profiles/r03_disturbance.txt).  The feature is a DEVICE target
# feature; the host pass of hipcc does not know it and says so once per translation unit (filtered
# below).  tests/test_native_build.py disassembles the library and holds the rule.
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mcode-object-version=5",
         "-Wno-unused-value", "-Wno-pass-failed"] + NO_PACKED_FP32
_HOST_NOISE = "'-packed-fp32-ops' is not a recognized feature for this target"


def _mtime(path: str) -> float:
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _obj(src: str) -> str:
    return os.path.join(OBJ_DIR, src.replace(".hip", ".o"))


def _stale_objects():
    hdr = max(_mtime(os.path.join(CSRC, h)) for h in HEADERS)
    return [s for s in _sources()
            if _mtime(_obj(s)) < max(_mtime(os.path.join(CSRC, s)), hdr)]


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in _sources() + HEADERS]
    return any(_mtime(d) > built for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source into libaps_amd.so (hipcc cross-compiles without a GPU)."""
    if not force and not stale():
        return LIB
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = _sources() if force else _stale_objects()

    def compile_one(src):
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", _obj(src)]
        if verbose:
            print("[aps_amd.build]", " ".join(cmd), file=sys.stderr)
        done = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True)
        noise = [ln for ln in done.stderr.splitlines() if _HOST_NOISE not in ln]
        if noise:
            print("\n".join(noise), file=sys.stderr)
        if done.returncode:
            raise subprocess.CalledProcessError(done.returncode, cmd)

    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4) or 1) as pool:
        list(pool.map(compile_one, todo))
    link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in _sources()] + \
        ["-o", LIB]
    if verbose:
        print("[aps_amd.build]", " ".join(link), file=sys.stderr)
    subprocess.run(link, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
