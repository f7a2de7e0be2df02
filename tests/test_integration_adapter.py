"""integration/mi355x_coders.cc -- the BaseEncoder / BaseDecoder adapter a dropbox/lepton maintainer would add -- must
type-check against the reference's own headers (src/lepton/base_coders.hh, uncompressed_components.hh, io/MuxReader.hh ...)
with the reference's default defines.  Only possible where a reference checkout exists (the build container)."""
import os
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lepton")), reason="no reference checkout on this machine")
def test_adapter_type_checks_against_the_reference_headers():
    src = os.path.join(ROOT, "integration", "mi355x_coders.cc")
    inc = ["-I" + os.path.join(ROOT, "include"), "-I.", "-I..", "-I../vp8/util", "-I../vp8/model", "-I../vp8/encoder", "-I../vp8/decoder",
           "-I../../dependencies/md5"]
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-w", "-DNDEBUG", "-DDEFAULT_ALLOW_PROGRESSIVE", "-DHIGH_MEMORY", "-msse4.2"] + inc + [src]
    r = subprocess.run(cmd, cwd=os.path.join(REF, "lepton"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_adapter_uses_only_declared_entry_points():
    # every lep_* symbol the adapter calls is declared in include/lepton_mi355x.h and exported by the library
    import re
    from lepton_amd import abi

    text = open(os.path.join(ROOT, "integration", "mi355x_coders.cc")).read()
    header = open(os.path.join(ROOT, "include", "lepton_mi355x.h")).read()
    used = set(re.findall(r"\b(lep_(?:gpu|jpeg|file)_[a-z_0-9]+)\s*\(", text))
    assert used, "adapter does not call the library?"
    for name in used:
        assert re.search(r"\b%s\s*\(" % name, header), name + " is not declared in the header"
        assert name in abi.EXPORTS and hasattr(abi.lib(), name)
