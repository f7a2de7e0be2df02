#!/usr/bin/env python3
"""Writes tests/golden/ref_benchmark_{hdr,rep}.bin: the two pieces of the synthetic JPEG `lepton -benchmark` codes when it is
given no file (src/lepton/benchmark.cc:116-119: bigger_hdr + 76 x bigger_rep = 2,589,088 bytes; the arrays are held in
src/lepton/smalljpg.hh:491-3506).  Run once in the container that has /root/reference; bench.py and the tests assemble the
file from the two pieces (lepton_amd.corpus.reference_benchmark_jpeg) so that numbers line up with `lepton -benchmark`."""
import os
import re

SRC = "/root/reference/src/lepton/smalljpg.hh"
HERE = os.path.dirname(os.path.abspath(__file__))


def array(text, name):
    body = text[text.index(name + "[]"):]
    body = body[body.index("{") + 1: body.index("};")]
    return bytes(int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", body))


if __name__ == "__main__":
    t = open(SRC).read()
    hdr, rep = array(t, "bigger_hdr"), array(t, "bigger_rep")
    assert len(hdr) == 2048 and len(rep) == 34040, (len(hdr), len(rep))
    open(os.path.join(HERE, "ref_benchmark_hdr.bin"), "wb").write(hdr)
    open(os.path.join(HERE, "ref_benchmark_rep.bin"), "wb").write(rep)
    print("wrote", len(hdr), len(rep))
