#!/usr/bin/env python
"""Device code of a HIP fat binary (libaps_amd.so or one object): extract every gfx950 code object
from its clang offload bundles and disassemble it.

    python scripts/device_isa.py aps_amd/csrc/libaps_amd.so            # per-kernel instruction census
    python scripts/device_isa.py lib.so --grep 'v_pk_(add|mul|fma)_f32'  # kernels that contain a pattern

Used by tests/test_native_build.py to hold the rule that no shipped kernel contains packed-fp32 VALU
instructions ("co-residency")."""
import os
import re
import struct
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def code_objects(path):
    """[(triple, bytes)] of every device code object bundled in `path`"""
    data = open(path, "rb").read()
    out, pos = [], 0
    while True:
        at = data.find(MAGIC, pos)
        if at < 0:
            return out
        n = struct.unpack_from("<Q", data, at + 24)[0]
        p = at + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "amdgcn" in triple and size:
                out.append((triple, data[at + off:at + off + size]))
        pos = at + 24


def kernels(path):
    """{kernel symbol: [instruction lines]} over all gfx950 code objects of `path`"""
    res = {}
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(blob)
        try:
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True,
                                 text=True, check=True).stdout
        finally:
            os.unlink(f.name)
        name = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                name = m.group(1)
                res.setdefault(name, [])
            elif name and line.startswith("\t"):
                res[name].append(line.strip())
    return res


def kernel_metadata(path):
    """{kernel symbol: {"vgpr": n, "sgpr": n, "scratch": bytes per lane, "lds": bytes}} from the code
    objects' msgpack metadata (llvm-readelf --notes)"""
    res = {}
    readelf = OBJDUMP.replace("llvm-objdump", "llvm-readelf")
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(blob)
        try:
            txt = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True,
                                 check=True).stdout
        finally:
            os.unlink(f.name)
        # keys of a kernel's map come sorted: .group_segment_fixed_size before .name, the rest after
        cur, pending_lds = {}, None
        for line in txt.splitlines():
            line = line.strip()
            if line.startswith("- "):
                line = line[2:]
            if line.startswith(".group_segment_fixed_size:"):
                pending_lds = int(line.split(":", 1)[1])
            elif line.startswith(".name:"):
                if "name" in cur and "vgpr" in cur:
                    res[cur["name"]] = cur
                cur = {"name": line.split(":", 1)[1].strip(), "lds": pending_lds}
            else:
                for key, tag in ((".vgpr_count:", "vgpr"), (".sgpr_count:", "sgpr"),
                                 (".private_segment_fixed_size:", "scratch")):
                    if line.startswith(key) and "name" in cur:
                        cur[tag] = int(line.split(":", 1)[1])
        if "name" in cur and "vgpr" in cur:
            res[cur["name"]] = cur
    return res


def main():
    path = sys.argv[1]
    pat = re.compile(sys.argv[sys.argv.index("--grep") + 1]) if "--grep" in sys.argv else None
    ks = kernels(path)
    hits = 0
    for name, ins in sorted(ks.items()):
        if pat is None:
            print(f"{len(ins):7d}  {name}")
        else:
            n = sum(1 for i in ins if pat.search(i))
            if n:
                hits += 1
                print(f"{n:6d} of {len(ins):6d}  {name}")
    if pat is not None:
        print(f"{hits} of {len(ks)} functions match")


if __name__ == "__main__":
    main()
