"""Container reader against claims its data cannot back and the chained-header rules of `lepton -lepcat`
files (read_ujpg, jpgcoder.cc:4139-4188, 4326-4343).  The verdicts are the reference binary's (probed with oracle/_ref/lepton:
a 46-byte file whose HDR section claims 127 MiB is UNSUPPORTED_JPEG there too -- after zero-filling and walking 127 MiB)."""
import os
import struct
import subprocess
import sys
import zlib

import pytest

from conftest import GOLDEN, ROOT
from lepton_amd.codec import LepFile, LeptonError, lep_stream


def craft(payload, jpeg_size=1000, nthreads=1, version=1, flag=b"Z", zsize=None, tail=b"CMP"):
    z = zlib.compress(payload, 9)
    return (b"\xcf\x84" + bytes([version]) + flag + bytes([nthreads]) + b"\0" * 15
            + struct.pack("<II", jpeg_size, len(z) if zsize is None else zsize) + z + tail)


def refusal(blob):
    with pytest.raises(LeptonError) as e:
        LepFile(blob)
    return e.value.code


def test_unbacked_header_claim_is_refused_like_the_reference():
    assert refusal(craft(b"HDR" + struct.pack("<I", 127 << 20))) == 42            # UNSUPPORTED_JPEG
    assert refusal(craft(b"HDR" + struct.pack("<I", 2 << 20) + b"\xff\xd8")) == 42


def test_unbacked_claims_do_not_allocate():
    """one request must not cost the serving process 127 MB: peak RSS of a fresh process that opens such files 20 times"""
    code = (
        "import sys, struct, zlib, resource\n"
        "sys.path.insert(0, %r)\n"
        "from lepton_amd.codec import LepFile, LeptonError\n"
        "def craft(p):\n"
        "    z = zlib.compress(p, 9)\n"
        "    return b'\\xcf\\x84\\x01Z\\x01' + b'\\0' * 15 + struct.pack('<II', 1000, len(z)) + z + b'CMP'\n"
        "hdr = open(%r, 'rb').read()\n"
        "base = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss\n"
        "for sec in (b'HDR', None):\n"
        "    for _ in range(10):\n"
        "        try:\n"
        "            LepFile(craft(b'HDR' + struct.pack('<I', 127 << 20)) if sec else b'')\n"
        "        except Exception:\n"
        "            pass\n"
        "print(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - base)\n"
    ) % (ROOT, os.path.join(GOLDEN, "c420_160x120.lep"))
    grown_kb = int(subprocess.check_output([sys.executable, "-c", code]).split()[-1])
    assert grown_kb < 16 << 10, grown_kb


def test_unbacked_garbage_sections_are_refused():
    """GRB / PGR sections of megabytes with nothing behind them: STREAM_INCONSISTENT, like CRS / FRS counts the data cannot
    back (documented deviation: the reference allocates and zero-fills)"""
    good = open(os.path.join(GOLDEN, "c420_160x120.lep"), "rb").read()
    zsize = struct.unpack("<I", good[24:28])[0]
    payload = zlib.decompress(good[28:28 + zsize])
    for tag in (b"GRB", b"PGR"):
        p = payload + tag + struct.pack("<I", 100 << 20)
        z = zlib.compress(p, 9)
        blob = good[:24] + struct.pack("<I", len(z)) + z + good[28 + zsize:]
        assert refusal(blob) == 7
    # a small unbacked claim keeps the reference's semantics (zero-filled to the claim)
    p = payload + b"GRB" + struct.pack("<I", 64) + b"\xff\xd9"
    z = zlib.compress(p, 9)
    f = LepFile(good[:24] + struct.pack("<I", len(z)) + z + good[28 + zsize:])
    assert f is not None


def test_chained_header_rules():
    """a file that follows a `CNT` section has no header bytes of its own: a non-zero compressed size is an assertion failure
    ("Special concatenation requires 0 size header"), and a CNT with nothing behind it still owns the next file's header"""
    v2 = os.path.join(GOLDEN, "v2")
    cat = open(os.path.join(v2, "concat_3.lep"), "rb").read()
    files = lep_stream(cat)
    assert len(files) == 3
    second = files[0].consumed
    assert struct.unpack("<I", cat[second + 24:second + 28])[0] == 0
    bad = bytearray(cat)
    bad[second + 24:second + 28] = struct.pack("<I", 1)
    with pytest.raises(LeptonError) as e:
        lep_stream(bytes(bad))
    assert e.value.code == 1


def test_refusals_of_the_header_come_before_a_packet_for_an_unbound_stream():
    """two mutants the structure-aware differential fuzz found (tests/fuzz/diff_lep_structured.py, seed 11, format-2 fixtures):
    a progressive file whose embedded JPEG header is damaged (an SOF length field of 0xffff; an SOS length field of 3) AND whose
    packets are mis-framed so that one is addressed to a stream id no hand-off created.  The reference binary answers
    UNSUPPORTED_JPEG for both ("out of memory error": the general re-coder's empty scan table; "unknown marker found") -- the
    header's refusals come before its decoder ever routes the stray packet ("Cannot send to thread that wasn't bound")."""
    d = os.path.join(GOLDEN, "fuzz")
    for name in ("v2_prog_sof_length_ffff_and_misframed_packets", "v2_prog_sos_length_3_and_misframed_packets"):
        assert refusal(open(os.path.join(d, name + ".lep"), "rb").read()) == 42, name
