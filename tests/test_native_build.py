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


def _built_library():
    import pytest
    lib = os.path.join(ROOT, "aps_amd", "csrc", "libaps_amd.so")
    if not os.path.exists(lib) or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("library / llvm-objdump not here")
    return lib


def test_no_kernel_contains_packed_fp32_instructions():
    """Round 3 traced the cross-stream disturbance of round 2 (wrong values in lanes 48-63 of an STFT
    wavefront while a particular MFMA kernel build of another stream shared its CU) to packed-fp32
    VALU instructions in the victim: the library is built with that target feature off
    (aps_amd/build.py: NO_PACKED_FP32) and this test disassembles every shipped code object."""
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import device_isa
    kernels = device_isa.kernels(_built_library())
    assert len(kernels) > 200, len(kernels)          # the disassembly saw the library
    pat = re.compile(r"\bv_pk_(add|mul|fma)_f32\b|\bv_pk_mov_b32\b")
    bad = {name: sum(1 for ins in body if pat.search(ins)) for name, body in kernels.items()}
    bad = {k: v for k, v in bad.items() if v}
    assert not bad, f"packed-fp32 instructions in {len(bad)} kernels, e.g. {sorted(bad.items())[:3]}"


def test_forward_path_kernels_use_no_scratch():
    """the kernels of the benchmarked forward path keep everything in registers (a kernel that needs
    scratch cost the joint step a fifth in round 2): the code-object metadata says so"""
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import device_isa
    meta = device_isa.kernel_metadata(_built_library())
    hot = ["gemm_fp16x2_kernel", "gemm_f32_kernel", "fp16x2_split_kernel", "lstm_layer_kernelILi32ELi2ELi2E",
           "lstm_layer_kernelILi32ELi1ELi4ELb1ELi1E", "lstm_stack_kernelILi32ELi2ELi1E", "attention_small_kernel",
           "glu_dwconv_kernel", "features_rows_kernelILi5E", "covariance_partial_kernelILi4ELi64E",
           "beamform_kernelILi4E", "row_exp_kernel", "stft512_wave_kernelILb0ELb0E"]
    seen = {h: 0 for h in hot}
    for name, m in meta.items():
        for h in hot:
            if h in name:
                seen[h] += 1
                assert m["scratch"] == 0, f"{name}: {m['scratch']} bytes of scratch per lane"
    assert all(seen.values()), seen
