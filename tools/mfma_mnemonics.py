#!/usr/bin/env python
"""Count the MFMA mnemonics in the gfx950 code objects of libndmps_hip.so (works on a copy in a temporary
directory: llvm-objdump --offloading drops its extracted bundles next to the file it is given).
usage: python tools/mfma_mnemonics.py"""
import collections
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "img-compression-mps_amd", "libndmps_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def mnemonics():
    counts = collections.Counter()
    with tempfile.TemporaryDirectory() as tmp:
        lib = os.path.join(tmp, "lib.so")
        shutil.copy(LIB, lib)
        subprocess.run([OBJDUMP, "--offloading", lib], check=True, capture_output=True, cwd=tmp)
        for obj in glob.glob(os.path.join(tmp, "*gfx950*")):
            text = subprocess.run([OBJDUMP, "-d", obj], check=True, capture_output=True, text=True).stdout
            counts.update(re.findall(r"\bv_mfma_[a-z0-9_]+", text))
    return counts


if __name__ == "__main__":
    for name, n in sorted(mnemonics().items()):
        print(f"{n:6d}  {name}")
    sys.exit(0)
