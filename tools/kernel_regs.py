#!/usr/bin/env python
"""Register / LDS / spill report of the gfx950 kernels inside a host object or libdecnet_hip.so.

    python tools/kernel_regs.py decnet_amd/lib/obj/spamat_mfma.hip.o [name-substring]

Carves the embedded code objects (ELF images inside the .hip_fatbin clang-offload-bundle) out of the file and
reads their AMDGPU metadata notes with llvm-readelf."""
import re
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(blob):
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    pos = blob.find(magic)
    while pos >= 0:
        n = int.from_bytes(blob[pos + 24:pos + 32], "little")
        p = pos + 32
        for _ in range(n):
            off, size, tlen = (int.from_bytes(blob[p + 8 * i:p + 8 * i + 8], "little") for i in range(3))
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "amdgcn" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(magic, pos + 1)
    return out


def main():
    blob = open(sys.argv[1], "rb").read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for b in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", b).group(1)
            if want not in name:
                continue
            g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            print("%-70s vgpr %3s agpr %3s sgpr %3s spill v%s s%s lds %6s scratch %s" % (
                dem[:70], g("vgpr_count"), b.split()[0], g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"),
                g("group_segment_fixed_size"), g("private_segment_fixed_size")))


if __name__ == "__main__":
    main()
