"""`lepton -startbyte=<s> -trunc=<t>` (src/lepton/jpgcoder.cc:1132-1140, 3801-3843; the reference's test_suite/test_2nd_block.sh,
test_3rd_block.sh, test_last_block.sh, test_trunc.sh): a JPEG stored as fixed-size blocks is compressed block by block; the
.lep (format flag 'Y') restores bytes [start_byte, trunc) only.  Fixtures written by the reference binary
(tests/golden/make_golden.py, SLICES); the arithmetic coder starts in the middle of the image for them."""
import ctypes as C

import pytest

import oracle_binding as ob
from conftest import embedded_cases, golden, slice_cases
from lepton_amd.codec import JpegImage, LepFile, LeptonError

CASES = slice_cases()


def test_there_are_slice_fixtures():
    assert len(CASES) >= 5 and any(golden(n)[1][4] >= 4 for n, _, _ in CASES)   # at least one with four thread segments


@pytest.mark.parametrize("name,start,trunc", CASES)
def test_slice_lep_equals_the_reference(name, start, trunc):
    jpg, lep = golden(name)
    img = JpegImage(jpg, start_byte=start, trunc=trunc)
    segs = img.plan()
    assert segs[0].luma_y_start > 0 or start < 1000      # the hot path begins mid-image
    streams, _ = ob.oracle_encode(img.desc, segs)
    got = img.write_lep(streams)
    assert got[3:4] == b"Y" and got == lep


@pytest.mark.parametrize("name,start,trunc", CASES)
def test_slice_restores_exactly_its_bytes(name, start, trunc):
    jpg, lep = golden(name)
    f = LepFile(lep)
    ob.oracle_decode(f.desc, f.segments, f.streams)
    assert f.recode() == jpg[start:(trunc or len(jpg))]


def test_slice_errors_are_the_references():
    jpg, _ = golden("c420_160x120")
    with pytest.raises(LeptonError) as e:        # nothing but garbage behind start_byte
        JpegImage(jpg, start_byte=len(jpg) + 10)
    assert e.value.code == 14                    # ONLY_GARBAGE_NO_JPEG
    prog, _ = golden("prog_c420_320x240")
    with pytest.raises(LeptonError) as e:        # "Encode of partial progressive images not allowed" (jpgcoder.cc:1205-1208)
        JpegImage(prog, start_byte=3000)
    assert e.value.code == 8


from test_core_emulation import emu  # noqa: E402,F401  (the module-scoped fixture that builds tests/emu/libcore_emu.so)


@pytest.mark.parametrize("name,start,trunc", CASES)
def test_kernel_sources_code_slices_like_the_oracle(emu, name, start, trunc):
    """the v3 encoder / v4 decoder kernel sources (lane-loop emulation) on segments that start mid-image"""
    jpg, _ = golden(name)
    img = JpegImage(jpg, start_byte=start, trunc=trunc)
    d = img.desc
    segs = img.plan()
    want, _ = ob.oracle_encode(d, segs)
    for s, w in zip(segs, want):
        cap = len(w) + 4096
        buf = C.create_string_buffer(cap)
        n = C.c_uint32(0)
        assert emu.emu_encode_segment_v3(C.byref(d), s.luma_y_start, s.luma_y_end, s.is_last, buf, cap, C.byref(n), None) == 0
        assert buf.raw[: n.value] == w
    for c in range(d.ncomp):
        C.memset(d.blocks[c], 0, d.nblocks(c) * 128)
    for s, w in zip(segs, want):
        assert emu.emu_decode_segment_v4(C.byref(d), s.luma_y_start, s.luma_y_end, s.is_last, w, len(w), None) == 0
    # rows in front of the slice are not coded: only the coded rows come back
    f = LepFile(golden(name)[1])
    ob.oracle_decode(f.desc, f.segments, f.streams)
    for c in range(d.ncomp):
        assert C.string_at(d.blocks[c], d.nblocks(c) * 128) == C.string_at(f.desc.blocks[c], f.desc.nblocks(c) * 128)


@pytest.mark.parametrize("name,offset", embedded_cases())
def test_embedded_jpeg_equals_the_reference(name, offset):
    """`lepton -embedding=<n>` (jpgcoder.cc:1135-1137, 2275-2282; test_suite/test_embedded.sh): a JPEG inside a larger blob --
    the bytes in front travel in a 'PGE' section, the bytes behind EOI as ordinary garbage, the .lep restores the blob"""
    blob, lep = golden(name)
    img = JpegImage(blob, embedding=offset)
    segs = img.plan()
    streams, _ = ob.oracle_encode(img.desc, segs)
    assert img.write_lep(streams) == lep
    f = LepFile(lep)
    ob.oracle_decode(f.desc, f.segments, f.streams)
    assert f.recode() == blob
    with pytest.raises(LeptonError):
        JpegImage(blob, embedding=offset + 1)     # no SOI there


# ---- on the GPU -----------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name,start,trunc", CASES)
def test_gpu_slices(gpu_codec, name, start, trunc):
    jpg, lep = golden(name)
    assert gpu_codec.compress_slice(jpg, start, trunc) == lep
    assert gpu_codec.decompress(lep) == jpg[start:(trunc or len(jpg))]


@pytest.mark.gpu
@pytest.mark.parametrize("name,offset", embedded_cases())
def test_gpu_embedded(gpu_codec, name, offset):
    blob, lep = golden(name)
    assert gpu_codec.compress_embedded(blob, offset) == lep
    assert gpu_codec.decompress(lep) == blob


@pytest.mark.gpu
def test_gpu_batch_decompress_takes_slices():
    from lepton_amd.codec import GpuCodec

    codec = GpuCodec(0)
    names = [n for n, _, _ in CASES]
    leps = [golden(n)[1] for n in names] + [golden("c420_160x120")[1]]
    want = [golden(n)[0][s:(t or None)] for n, s, t in CASES] + [golden("c420_160x120")[0]]
    out, status, _ = codec.decompress_batch(leps)
    assert status == [0] * len(leps) and out == want


def test_permissive_files_decode():
    """`lepton -permissive` wraps bytes that are not a JPEG at all: a stock header, every byte in a 'PGE' section, empty coder
    streams (generic_compress.cc:60-215).  Such files exist in stores written by the reference, so they must come back."""
    blob, lep = golden("permissive_5000")
    assert lep[3:4] == b"Y" and blob[:2] != b"\\xff\\xd8"
    f = LepFile(lep)
    ob.oracle_decode(f.desc, f.segments, f.streams)
    assert f.recode() == blob
