#!/usr/bin/env python
"""Disassemble one gfx950 kernel out of a host object / shared library.

    python tools/kernel_disasm.py decnet_amd/lib/obj/spamat_mfma.hip.o 'spamat_fwd_mfma<15, 2, 2>' > k.s
"""
import subprocess
import sys
import tempfile

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from kernel_regs import code_objects  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def main():
    blob = open(sys.argv[1], "rb").read()
    want = sys.argv[2]
    for co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            syms = subprocess.run([OBJDUMP, "-t", f.name], capture_output=True, text=True).stdout
            names = [ln.split()[-1] for ln in syms.splitlines() if " F " in ln and ".text" in ln]
            for n in names:
                dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
                if want in dem:
                    out = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", "--disassemble-symbols=" + n, f.name],
                                         capture_output=True, text=True).stdout
                    print("; " + dem)
                    print(out)
                    return


if __name__ == "__main__":
    main()
