#!/usr/bin/env python3
"""Which kernels does a change touch?  Compiles lepton_amd/csrc/lep_gpu.hip of a git revision and of the working tree to gfx950
assembly and compares the instruction streams kernel by kernel (comments and directives stripped).  Used when the GPU budget is
spent: a kernel whose ISA is IDENTICAL to a revision that passed on hardware needs no new visit.
usage: python scripts/isa_diff.py <git revision>"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"


def asm(src, out):
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-o", out, src],
                          stderr=subprocess.DEVNULL)
    kernels, cur, body = {}, None, []
    for line in open(out):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m and "kernel" in m.group(1):
            cur, body = m.group(1), []
            continue
        if cur is None:
            continue
        t = line.strip()
        if t.startswith("s_endpgm"):
            kernels[cur] = "\n".join(body)
            cur = None
        elif t and not t.startswith(";") and not t.startswith("."):
            body.append(re.sub(r"\s*;.*$", "", t))
    return kernels


def main():
    rev = sys.argv[1]
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call("git archive %s lepton_amd/csrc include | tar -x -C %s" % (rev, tmp), shell=True, cwd=ROOT)
        old = asm(os.path.join(tmp, "lepton_amd", "csrc", "lep_gpu.hip"), os.path.join(tmp, "old.s"))
        new = asm(os.path.join(ROOT, "lepton_amd", "csrc", "lep_gpu.hip"), os.path.join(tmp, "new.s"))
    for name in sorted(set(old) | set(new)):
        m = re.search(r"lep_\w+?kernel(ILi\d+E)?", name)
        short = m.group(0) if m else name[:50]
        if name in old and name in new:
            print("%-46s %6d instr  %s" % (short, new[name].count("\n") + 1, "IDENTICAL" if old[name] == new[name] else "DIFFERENT"))
        else:
            print("%-46s only in %s" % (short, rev if name in old else "the working tree"))


if __name__ == "__main__":
    main()
