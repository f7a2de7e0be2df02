"""integration/mi355x_coders.cc -- the BaseEncoder / BaseDecoder adapter a dropbox/lepton maintainer would add -- LINKED INTO
the reference and RUN.  oracle/Makefile.ref compiles the reference's own sources where they lie, swaps the coder factories of
jpgcoder.cc:440-471 for the adapter's (oracle/mi355x_factories.sed, applied to a copy under oracle/_ref/) and links
  * oracle/_ref/lepton-mi355x       against liblepton_mi355x.so: the drop-in dropped in (GPU test below), and
  * oracle/_ref/lepton-adapter-oracle  with the four entry points the adapter calls served by the CPU oracle
    (tests/emu/abi_over_oracle.c): the adapter's own code -- mux slices, size trailer, decode_row from several re-coder
    threads, decode_chunk's progress signalling, truncated geometry -- executes here, without a GPU.
Round 1 only type-checked this file; running it found four defects (full components touched on the baseline path, workers
never registered, decode_row racing between re-coder threads, decode_chunk not signalling component progress)."""
import os
import subprocess

import pytest

from conftest import ROOT, golden, golden_cases

MI355X_BIN = os.path.join(ROOT, "oracle", "_ref", "lepton-mi355x")
ADAPTER_ORACLE_BIN = os.path.join(ROOT, "oracle", "_ref", "lepton-adapter-oracle")


def _round_trip_through(binary, names, tmp_path, verify_flags=("-skipverify",)):
    for n in names:
        jpg, lep = golden(n)
        jp, lp, bp = (str(tmp_path / (n + e)) for e in (".jpg", ".lep", ".back.jpg"))
        open(jp, "wb").write(jpg)
        r = subprocess.run([binary, "-unjailed"] + list(verify_flags) + [jp, lp], capture_output=True, timeout=120)
        assert r.returncode == 0, (n, r.returncode, r.stderr[-400:])
        assert open(lp, "rb").read() == lep, n + ": .lep differs from the reference's"
        r = subprocess.run([binary, "-unjailed", lp, bp], capture_output=True, timeout=120)
        assert r.returncode == 0, (n, r.returncode, r.stderr[-400:])
        assert open(bp, "rb").read() == jpg, n + ": restored JPEG differs"


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/lepton"), reason="no reference checkout on this machine")
def test_adapter_runs_inside_the_reference_over_the_oracle(tmp_path):
    """every golden fixture, both directions, through the reference binary with the adapter in place of its coders"""
    subprocess.check_call(["make", "-s", "-j8", "-f", "Makefile.ref", "adapter_check"], cwd=os.path.join(ROOT, "oracle"))
    _round_trip_through(ADAPTER_ORACLE_BIN, golden_cases(), tmp_path)
    # and with the reference's default round-trip validation in force (it decodes through the adapter too)
    _round_trip_through(ADAPTER_ORACLE_BIN, ["c420_160x120", "prog_c420_320x240", "q30_256x256_4seg"], tmp_path, verify_flags=())


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(MI355X_BIN), reason="oracle/_ref/lepton-mi355x is built where /root/reference exists and travels with the snapshot")
def test_gpu_reference_binary_with_the_adapter_linked_in(tmp_path):
    """`lepton-mi355x in.jpg out.lep` and back: the reference's own main(), file IO, header writer, Huffman re-coder and
    thread pool around the HIP kernels -- byte-identical to the files the unmodified reference wrote"""
    names = ["c420_160x120", "c420_odd_203x149", "q30_256x256_4seg", "lay_440_640x480_2seg", "gray_120x88", "rst_c420_176x112", "truncated",
             "prog_c420_320x240", "prog_gray_120x88", "prog_truncated_mid", "one_block_8x8"]
    names = [n for n in names if n in golden_cases()]
    assert len(names) >= 8
    _round_trip_through(MI355X_BIN, names, tmp_path)
    _round_trip_through(MI355X_BIN, ["c420_160x120", "prog_c420_320x240"], tmp_path, verify_flags=())


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
