#!/usr/bin/env python
"""Scan the gfx950 code of an object file (or of libndmps_hip.so) for global loads that sit right behind an
`s_waitcnt vmcnt(0)`: the signature of loads in per-lane guarded (exec-masked) blocks, which the compiler makes wait
for every load before them (DESIGN.md 5.7 e).  Prints, per kernel, the number of vector-memory loads and how many of
them have such a wait within the six instructions before them.
usage: python tools/serialised_loads.py img-compression-mps_amd/csrc/gemm.o [kernel-name substring]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def main():
    obj = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    tmp = tempfile.mkdtemp()
    try:
        shutil.copy(obj, os.path.join(tmp, "g.o"))
        subprocess.run([OBJDUMP, "--offloading", "g.o"], cwd=tmp, capture_output=True)
        code = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not code:
            raise SystemExit("no gfx950 code object in " + obj)
        text = subprocess.run([OBJDUMP, "-d", code[0]], cwd=tmp, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp)
    for part in re.split(r"\n(?=[0-9a-f]{16} <)", text):
        m = re.match(r"[0-9a-f]{16} <([^>]+)>", part)
        if not m:
            continue
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = name.replace("(anonymous namespace)::", "")
        if want not in name:
            continue
        lines = [ln.split("//")[0].strip() for ln in part.split("\n")[1:]]
        loads = behind = 0
        for i, ln in enumerate(lines):
            if ln.startswith(("global_load", "buffer_load")):
                loads += 1
                behind += any("s_waitcnt vmcnt(0)" in x for x in lines[max(0, i - 6):i])
        if loads:
            print(f"{name[:110]:110s} loads={loads:4d} behind_vmcnt0={behind}")


if __name__ == "__main__":
    main()
