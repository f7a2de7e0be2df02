"""Host logic: ABI surface, hand-off wire format (reference golden vector), mux framing, error codes."""
import ctypes as C
import os
import random
import subprocess

import pytest

from conftest import GOLDEN, golden
from lepton_amd import abi
from lepton_amd.codec import JpegImage
from lepton_amd.codec import LeptonError, LepFile

# test_suite/test_invariants.cc:332-343 -- serialised form of the 8 hand-offs listed at :300-331
HANDOFF_GOLDEN = bytes([
    0x48, 0x08, 0x67, 0x45, 0xc6, 0x23, 0x7b, 0x32, 0x69, 0x03, 0xaf, 0xa3, 0x4a, 0x14, 0x29, 0x1f, 0x00, 0x00,
    0xba, 0x58, 0xab, 0xd7, 0x7e, 0x50, 0xf2, 0x03, 0xe3, 0x29, 0x7c, 0x00, 0x54, 0x08, 0x00, 0x00, 0x1b, 0x23,
    0xe8, 0xe9, 0x16, 0x1f, 0xe7, 0x05, 0x8a, 0xf0, 0xd2, 0x86, 0xcd, 0xbd, 0x00, 0x00, 0xc9, 0xc4, 0x9a, 0x07,
    0x68, 0x6b, 0x66, 0x02, 0x0d, 0x50, 0x31, 0x3a, 0xa3, 0x30, 0x00, 0x00, 0x25, 0x61, 0x5d, 0x89, 0x8c, 0x62,
    0x05, 0x07, 0xa8, 0xd7, 0x5e, 0x04, 0xab, 0x3d, 0x00, 0x00, 0xcd, 0xd0, 0xc6, 0xe0, 0x03, 0x0b, 0x9b, 0x04,
    0xac, 0xdb, 0xf2, 0xbb, 0x8c, 0x87, 0x00, 0x00, 0x21, 0xf5, 0x3d, 0xbd, 0x3d, 0x7c, 0xdc, 0x07, 0x70, 0x1a,
    0x3e, 0x48, 0x41, 0x42, 0x00, 0x00, 0xfc, 0xad, 0x67, 0x23, 0x07, 0x05, 0x3e, 0x01, 0x7e, 0x46, 0xea, 0x39,
    0x95, 0xac, 0x00, 0x00])
HANDOFF_VALUES = [
    (17767, 22714, 846930886, 105, 3, (-23633, 5194, 7977)), (22714, 8987, 1350490027, 242, 3, (10723, 124, 2132)),
    (8987, 50377, 521595368, 231, 5, (-3958, -31022, -16947)), (50377, 24869, 1801979802, 102, 2, (20493, 14897, 12451)),
    (24869, 53453, 1653377373, 5, 7, (-10328, 1118, 15787)), (53453, 62753, 184803526, 155, 4, (-9300, -17422, -30836)),
    (62753, 44540, 2084420925, 220, 7, (6768, 18494, 16961)), (44540, 0, 84353895, 62, 1, (18046, 14826, -21355))]


def test_library_exports_every_declared_symbol():
    L = abi.lib()
    hdr = open(os.path.join(os.path.dirname(GOLDEN), "..", "include", "lepton_mi355x.h")).read()
    import re

    declared = set(re.findall(r"\b(lep_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(abi.EXPORTS), declared ^ set(abi.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    out = subprocess.check_output(["nm", "-D", "--defined-only", abi.LIB_PATH]).decode()
    for name in declared:
        assert (" T " + name) in out, name
    assert b"gfx950" in L.lep_version()


def test_handoff_golden_vector_roundtrip():
    L = abi.lib()
    arr = (abi.Handoff * 16)()
    n = L.lep_handoffs_parse(HANDOFF_GOLDEN, len(HANDOFF_GOLDEN), arr, 16)
    assert n == 8
    for h, (ys, ye, size, ob, nb, dc) in zip(arr, HANDOFF_VALUES):
        assert (h.luma_y_start, h.segment_size, h.overhang_byte, h.num_overhang_bits) == (ys, size, ob, nb)
        assert tuple(h.last_dc[:3]) == dc
    for h, (_, ye, *_r) in list(zip(arr, HANDOFF_VALUES))[:7]:
        assert h.luma_y_end == ye   # luma end is implied by the next record
    buf = C.create_string_buffer(2 + 16 * 8)
    assert L.lep_handoffs_serialize(arr, 8, buf, len(buf)) == len(HANDOFF_GOLDEN)
    assert buf.raw == HANDOFF_GOLDEN


def _mux(streams, version=1):
    L = abi.lib()
    n = len(streams)
    arr = (abi.Bytes * n)()
    keep = []
    for i, s in enumerate(streams):
        b = C.create_string_buffer(s, max(1, len(s)))
        keep.append(b)
        arr[i].data, arr[i].len, arr[i].cap = C.cast(b, C.c_void_p).value, len(s), len(s)
    out = abi.Bytes()
    assert L.lep_mux(arr, n, version, C.byref(out)) == 0
    data = out.tobytes()
    L.lep_free(out.data)
    return data


def _demux(data):
    L = abi.lib()
    arr = (abi.Bytes * 16)()
    assert L.lep_demux(data, len(data), arr) == 0
    res = [arr[i].tobytes() for i in range(16)]
    for i in range(16):
        L.lep_free(arr[i].data)
    return res


@pytest.mark.parametrize("sizes", [[1], [255, 256, 257], [4096, 4095, 1], [70000, 3, 200000, 0, 65536], [300000] * 8, [1] * 16])
def test_mux_demux_roundtrip(sizes):
    rnd = random.Random(sum(sizes))
    streams = [bytes(rnd.getrandbits(8) for _ in range(n)) for n in sizes]
    framed = _mux(streams)
    got = _demux(framed + b"\x00\x00\x00\x00")   # v1 files are followed by a 4-byte size word
    for i, s in enumerate(streams):
        assert got[i] == s
    assert all(len(g) == 0 for g in got[len(streams):])


def test_mux_packet_shapes_match_reference_policy():
    # 8 equal 204 kB streams: first packets are 4 KiB fixed-size ones per stream in id order (SURVEY.md A.2 [probe])
    framed = _mux([bytes([i]) * 204000 for i in range(8)])
    pos = 0
    for i in range(8):
        assert framed[pos] == (i | (1 << 4))
        pos += 1 + 4096
    assert framed[pos] == (0 | (3 << 4))


def test_reject_non_jpeg_and_progressive_when_disallowed():
    with pytest.raises(LeptonError) as e:
        JpegImage(b"not a jpeg at all")
    assert e.value.code == 42   # UNSUPPORTED_JPEG (src/vp8/util/memory.hh:37; the reference binary exits 42 on such files)
    from lepton_amd import corpus

    prog = corpus.synth_jpeg(64, 64, 5, progressive=True)
    with pytest.raises(LeptonError) as e:
        JpegImage(prog, allow_progressive=False)
    assert e.value.code == 8   # PROGRESSIVE_UNSUPPORTED


def test_lep_header_fields():
    jpg, lep = golden("c420_160x120")
    f = LepFile(lep)
    assert lep[:2] == b"\xcf\x84" and lep[2] == 1 and lep[3:4] == b"Z"
    assert abi.lib().lep_file_jpeg_size(f.handle) == len(jpg)
    assert int.from_bytes(lep[-4:], "little") == len(lep)
    assert len(f.segments) == lep[4] and f.segments[-1].is_last == 1


def test_bytes_after_a_lep_file_are_ignored_like_the_reference():
    # Concatenated .lep files (jpgcoder.cc:1881-1897, test_suite/test_concat.sh) only chain with the brotli container
    # (v2+); for the v1 files written here the reference binary decodes the first file and ignores whatever follows
    # (checked against it: another .lep, zeros, 0xff, a ramp).  Same here.
    import oracle_binding as ob

    jpg, lep = golden("c420_160x120")
    _, other = golden("gray_120x88")
    for tail in (other, bytes(100), bytes([255]) * 50, bytes(range(256)) * 3, b"\x01"):
        f = LepFile(lep + tail)
        ob.oracle_decode(f.desc, f.segments, f.streams)
        assert f.recode() == jpg


def test_batch_chunk_plan():
    """lep_batch_plan: how lep_compress_batch cuts a batch (the GPU pipeline itself needs a device; its chunking does not)"""
    import ctypes as C

    L = abi.lib()

    def plan(file_bytes, frame_bytes=None, **opt):
        n = len(file_bytes)
        fb = (C.c_size_t * n)(*file_bytes)
        fr = (C.c_size_t * n)(*(frame_bytes or [25_000_000 if b >= 500_000 else 6_000_000 for b in file_bytes]))
        o = abi.BatchOptions()
        for k, v in opt.items():
            setattr(o, k, v)
        first = (C.c_int * (n + 2))()
        k = L.lep_batch_plan(fb, fr, n, C.byref(o), first, n + 2)
        assert k >= 0 and first[k] == n and list(first[: k + 1]) == sorted(first[: k + 1])
        return [first[i + 1] - first[i] for i in range(k)]

    big, small = 2_200_000, 60_000                     # 8 thread segments / 1 thread segment
    assert plan([big] * 1024) == [1024]                # fits one launch: nothing to overlap with, not split
    assert plan([big] * 2688) == [896, 896, 896]       # three launches either way: balanced
    assert plan([big] * 3072) == [1024, 1024, 1024]    # 8192 segments per chunk: the wavefronts the chip holds at once
    assert plan([big] * 2048) == [1024, 1024]
    assert plan([big] * 2100) == [700, 700, 700]       # balanced, not 1024 + 1024 + 52
    sizes = plan([big] * 1025)
    assert len(sizes) == 2 and abs(sizes[0] - sizes[1]) <= 2
    assert plan([small] * 5000) == [1000] * 5           # images, not segments, bound chunks of small files
    assert all(s * 8 <= 8192 + 8 for s in plan([big] * 10000))
    mixed = plan([big, small] * 3000)
    assert sum(mixed) == 6000 and max(mixed) <= 1024
    # unusable files (frame_bytes 0) ride along in whatever chunk they fall into and do not count
    assert sum(plan([big] * 10, frame_bytes=[0, 25_000_000] * 5)) == 10
    # explicit settings override the automatic budget
    assert plan([big] * 2048, chunk_images=1024) == [1024, 1024]
    assert plan([big] * 100, chunk_images=7) == [7] * 14 + [2]
    assert len(plan([big] * 64, chunk_frame_bytes=101_000_000)) == 16   # 4 x 25 MB frames per chunk
    assert plan([big] * 2048, host_huffman=1) == [1024, 1024]
    assert plan([]) == []


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "lepton")),
                    reason="needs the reference binary (built where /root/reference exists)")
def test_encode_thread_options_match_the_reference_flags(tmp_path):
    """-maxencodethreads / -minencodethreads / -evensplit change the segment count and cut points, i.e. the .lep bytes"""
    import subprocess
    import oracle_binding as ob
    from conftest import ROOT

    ref = os.path.join(ROOT, "oracle", "_ref", "lepton")
    jp = os.path.join(ROOT, "tests", "golden", "slice_4seg_q97.jpg")   # a 400 kB baseline file (the fixture's whole input)
    jpg = open(jp, "rb").read()
    for flags, (mx, mn, ev) in ((["-maxencodethreads=2"], (2, 0, 0)), (["-minencodethreads=8"], (0, 8, 0)), (["-evensplit"], (0, 0, 1)),
                                (["-maxencodethreads=4", "-minencodethreads=4", "-evensplit"], (4, 4, 1))):
        out = str(tmp_path / "o.lep")
        assert subprocess.run([ref, "-unjailed", "-skipverify"] + flags + [jp, out], capture_output=True).returncode == 0
        want = open(out, "rb").read()
        img = JpegImage(jpg)
        assert abi.lib().lep_jpeg_set_encode_options(img.handle, mx, mn, ev) == 0
        segs = img.plan(max_threads=0)
        streams, _ = ob.oracle_encode(img.desc, segs)
        assert len(segs) == want[4] and img.write_lep(streams, max_threads=0) == want


def test_the_library_says_which_sources_it_was_built_from(monkeypatch):
    """build() compiles a hash of every source file into lep_version(); the loaded library is held against
    the files on the box, and a library built from other sources -- a stale .so that travelled with a snapshot -- fails build(),
    hence smoke() and bench.py"""
    import __graft_entry__ as ge
    from lepton_amd import abi

    ok, said = ge.library_matches_sources()
    assert ok and ("src " + ge.source_sha16()) in said and abi.lib().lep_version().decode() == said
    monkeypatch.setattr(ge, "source_sha16", lambda: "0123456789abcdef")     # as if a source file had changed under the library
    assert ge.library_matches_sources()[0] is False
    monkeypatch.setattr(ge, "_compile_library", lambda record: None)          # (and the build had not noticed)
    monkeypatch.setattr(ge, "_build_locked", lambda record: None)
    import pytest
    with pytest.raises(RuntimeError, match="not built from the sources"):
        ge.build()
