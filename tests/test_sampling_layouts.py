"""Frames and Huffman layers that libjpeg's front end (PIL) cannot be asked for, written in the coefficient domain by
tests/jpeg_writer.py:

 * sampling layouts -- 4:4:0, true 4:1:1 (factor 4: refused), every component with its own factors, chroma sampled finer
   than luma, all components 2x2, one and two components, four components -- against the real reference binary (decision,
   exit code, .lep bytes, restored bytes) and through the current kernel sources in the lane-loop emulation;
 * legal-to-decode but non-canonical Huffman layers (ZRL + EOB, fill bytes before restart markers, mixed pad bits, a symbol
   coded twice, restart markers out of step, entropy-coded bytes behind the last MCU): the reference stops at errorlevel 1 / fails its round trip; here the host
   parser must refuse them (UNSUPPORTED_JPEG) and the GPU scan decoder must hand them to the host parser;
 * the Huffman half of the round-trip check (lep_jpeg_check_restores): 0 on every fixture, ROUNDTRIP_FAILURE where the
   reference's own default run fails (images/roundtripfail.jpg, test_suite/test_roundtrip.sh; one component with factors
   2x2 + restart markers), with the decode direction staying bug-compatible."""
import ctypes as C
import hashlib
import os
import subprocess
import zlib

import numpy as np
import pytest

import jpeg_writer as jw
import oracle_binding as ob
from conftest import REF_IMAGES, ROOT, embedded_cases, golden, golden_cases, roundtrip_failure_cases, slice_cases
from lepton_amd import abi
from lepton_amd.codec import JpegImage, LepFile, LeptonError

REF = os.path.join(ROOT, "oracle", "_ref", "lepton")
needs_ref = pytest.mark.skipif(not os.path.exists(REF), reason="needs the reference binary (built where /root/reference exists)")


def Y(h, v, i=1):
    return (i, h, v, 0, 0, 0)


def Cx(h, v, i):
    return (i, h, v, 1, 1, 1)


LAYOUTS = {
    "440": [Y(1, 2), Cx(1, 1, 2), Cx(1, 1, 3)],
    "411": [Y(4, 1), Cx(1, 1, 2), Cx(1, 1, 3)],            # SAMPLING_BEYOND_TWO_UNSUPPORTED
    "v4": [Y(1, 4), Cx(1, 1, 2), Cx(1, 1, 3)],
    "h3": [Y(3, 1), Cx(1, 1, 2), Cx(1, 1, 3)],
    "mixed": [Y(2, 2), Cx(2, 1, 2), Cx(1, 1, 3)],
    "mixed2": [Y(2, 2), Cx(1, 2, 2), Cx(2, 1, 3)],
    "chromafine": [Y(1, 1), Cx(2, 2, 2), Cx(1, 1, 3)],
    "chromafine2": [Y(1, 1), Cx(1, 2, 2), Cx(2, 1, 3)],
    "all22": [Y(2, 2), Cx(2, 2, 2), Cx(2, 2, 3)],
    "gray22": [Y(2, 2)],
    "gray21": [Y(2, 1)],
    "two": [Y(2, 1), Cx(1, 1, 2)],
    "two22": [Y(2, 2), Cx(1, 1, 2)],
    "cmyk": [Y(1, 1, 1), Y(1, 1, 2), Y(1, 1, 3), Y(1, 1, 4)],            # UNSUPPORTED_4_COLORS
    "cmykmixed": [Y(2, 2, 1), Cx(1, 1, 2), Cx(1, 1, 3), Y(2, 2, 4)],
    "ids": [Y(2, 2, 0), Cx(1, 1, 200), Cx(1, 1, 7)],
}
SAID = {b"UNSUPPORTED_4_COLORS": 4, b"SAMPLING_BEYOND_TWO_UNSUPPORTED": 10, b"SAMPLING_BEYOND_FOUR_UNSUPPORTED": 11}


def oracle_compress(jpg):
    img = JpegImage(jpg)
    segs = img.plan()
    streams, _ = ob.oracle_encode(img.desc, segs)
    return img, img.write_lep(streams)


@needs_ref
@pytest.mark.parametrize("name", sorted(LAYOUTS))
def test_layout_against_the_reference_binary(name, tmp_path):
    comps = LAYOUTS[name]
    jp, lp, bp = (str(tmp_path / n) for n in ("s.jpg", "s.lep", "s.back"))
    # restart markers only where the reference can restore them (one component with factors > 1: see the round-trip test)
    grey_sampled = len(comps) == 1 and comps[0][1] * comps[0][2] > 1
    for w, h, ri in [(97, 50, 0), (33, 70, 0 if grey_sampled else 3), (8, 8, 0), (200, 333, 0), (640, 480, 0 if grey_sampled else 7)]:
        jpg, _ = jw.write_baseline(w, h, comps, np.random.default_rng(zlib.crc32(("%s %d" % (name, w)).encode())), restart_interval=ri)
        open(jp, "wb").write(jpg)
        for f in (lp, bp):
            if os.path.exists(f):
                os.unlink(f)
        r = subprocess.run([REF, "-unjailed", "-skipverify", jp, lp], capture_output=True)
        said = [v for k, v in SAID.items() if k + b"\n" in r.stderr]   # the exit code itself is not reliable on error paths
        ref_ok = r.returncode == 0 and os.path.exists(lp) and os.path.getsize(lp) > 0 and not said
        try:
            _, got = oracle_compress(jpg)
            code = 0
        except LeptonError as e:
            got, code = None, e.code
        assert (got is not None) == ref_ok, (name, w, h, code, r.returncode, said)
        if not ref_ok:
            assert said and code == said[0], (name, w, h, code, said)
            continue
        assert got == open(lp, "rb").read(), "%s %dx%d: .lep differs from the reference's" % (name, w, h)
        assert subprocess.run([REF, "-unjailed", lp, bp], capture_output=True).returncode == 0
        f = LepFile(got)
        ob.oracle_decode(f.desc, f.segments, f.streams)
        back = f.recode()
        assert back == open(bp, "rb").read() and back == jpg, "%s %dx%d: restored file" % (name, w, h)


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(ROOT, "tests", "emu", "libcore_emu_layouts.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "emu", "core_emu.cc")])
    return C.CDLL(so)


def gpu_scan_decode(emu, jpg):
    """lep_jpeg_open_gpu + the scan decode kernel source in the emulation: 'host' (not eligible), ('irregular', status) or
    ('ok', planes)"""
    L = abi.lib()
    h, img, ok = C.c_void_p(), abi.HuffDecImage(), C.c_int(0)
    rc = L.lep_jpeg_open_gpu(jpg, len(jpg), C.byref(h), C.byref(img), C.byref(ok))
    if rc:
        return ("refused", rc)
    try:
        if not ok.value:
            return ("host", 0)
        p, n = C.c_void_p(), C.c_size_t(0)
        L.lep_jpeg_scan_bytes(h, C.byref(p), C.byref(n))
        scan = C.create_string_buffer(C.string_at(p, n.value) + b"\0" * 64, n.value + 64)
        img.scan = C.addressof(scan)
        d = abi.ImageDesc()
        L.lep_jpeg_describe(h, C.byref(d))
        planes = []
        for c in range(d.ncomp):
            b = C.create_string_buffer(d.nblocks(c) * 128)
            planes.append(b)
            img.blocks[c] = C.cast(b, C.c_void_p).value
        rows = (abi.HuffDecRow * (img.mcuv + 1))()
        assert emu.emu_huffman_decode_image(C.byref(img), rows) == 0
        status = rows[img.mcuv].aux >> 8
        if status:
            return ("irregular", status)
        if L.lep_jpeg_finish_gpu(h, rows):
            return ("irregular", "finish")                     # e.g. bytes left over behind the last MCU
        return ("ok", [p.raw for p in planes])
    finally:
        L.lep_jpeg_close(h)


@pytest.mark.parametrize("name", [n for n in sorted(LAYOUTS) if n not in ("411", "v4", "h3", "cmyk", "cmykmixed")])
def test_layout_through_the_kernel_sources(emu, name):
    """coder kernels (v3 encoder, v4 decoder) and the Huffman scan decoder, lane-loop emulation, against the oracle / host parser"""
    for w, h, ri in [(97, 50, 0), (33, 70, 0), (160, 120, 5)]:
        if ri and len(LAYOUTS[name]) == 1:
            ri = 0
        jpg, _ = jw.write_baseline(w, h, LAYOUTS[name], np.random.default_rng(77 + w), restart_interval=ri, density=0.4)
        img = JpegImage(jpg)
        d, segs = img.desc, img.plan()
        want, _ = ob.oracle_encode(d, segs)
        orig = [C.string_at(d.blocks[c], d.nblocks(c) * 128) for c in range(d.ncomp)]
        for s, wv in zip(segs, want):
            cap = len(wv) + 4096
            b = C.create_string_buffer(cap)
            n = C.c_uint32(0)
            assert emu.emu_encode_segment_v3(C.byref(d), s.luma_y_start, s.luma_y_end, s.is_last, b, cap, C.byref(n), None) == 0
            assert b.raw[: n.value] == wv, (name, w, h)
        for c in range(d.ncomp):
            C.memset(d.blocks[c], 0, d.nblocks(c) * 128)
        for s, wv in zip(segs, want):
            assert emu.emu_decode_segment_v4(C.byref(d), s.luma_y_start, s.luma_y_end, s.is_last, wv, len(wv), None) == 0
        assert [C.string_at(d.blocks[c], d.nblocks(c) * 128) for c in range(d.ncomp)] == orig, (name, w, h)
        kind, planes = gpu_scan_decode(emu, jpg)
        assert kind == "ok", "a single sequential scan of all components must be eligible for the GPU decoder"
        assert planes == orig, (name, w, h)


@pytest.mark.parametrize("name", ["gray22", "gray21", "gray12", "gray11"])
def test_one_component_scans_on_the_gpu_scan_decoders(emu, name):
    """A file of ONE component is never interleaved, whatever its sampling factors: the scan walks the nch x ncv blocks of the picture and
    steps over the blocks that pad the frame to whole MCUs (next_mcuposn, jpgcoder.cc).  The kernels see it as nch x ncv MCUs of one block
    with block rows bch apart (parse_jpeg_prepare_gpu) and parse_jpeg_finish_gpu picks the hand-off of every MCU row out of the records per
    block row: single-wave and lane-per-subsequence kernels in the emulation must leave the host parser's frame,
    and the container written from the GPU-side parse must be the host's, byte for byte -- widths and heights with and without padding
    blocks, one block row, one block."""
    L = abi.lib()
    comps = LAYOUTS.get(name) or {"gray12": [Y(1, 2)], "gray11": [Y(1, 1)]}[name]
    sizes = [(97, 50), (33, 70), (8, 8), (9, 9), (16, 16), (17, 33), (200, 333), (1, 1), (640, 40), (24, 481)]
    for w, h in sizes:
        jpg, _ = jw.write_baseline(w, h, comps, np.random.default_rng(zlib.crc32(("one %s %d %d" % (name, w, h)).encode())), density=0.5)
        host = JpegImage(jpg)
        d = host.desc
        orig = [C.string_at(d.blocks[c], d.nblocks(c) * 128) for c in range(d.ncomp)]
        want = host.write_lep(ob.oracle_encode(d, host.plan())[0])
        for kernel in ("wave", "lanes"):
            hnd, img, ok = C.c_void_p(), abi.HuffDecImage(), C.c_int(0)
            assert L.lep_jpeg_open_gpu(jpg, len(jpg), C.byref(hnd), C.byref(img), C.byref(ok)) == 0
            assert ok.value, "%s %dx%d: not planned for the GPU scan decoder" % (name, w, h)
            k = JpegImage(jpg)
            assert img.ncomp == 1 and img.hs[0] == 1 and img.vs[0] == 1 and img.mcuc == img.mcuh * img.mcuv
            p, n = C.c_void_p(), C.c_size_t(0)
            L.lep_jpeg_scan_bytes(hnd, C.byref(p), C.byref(n))
            scan = C.create_string_buffer(C.string_at(p, n.value) + bytes(80), n.value + 80)
            img.scan = C.addressof(scan)
            plane = C.create_string_buffer(d.nblocks(0) * 128)
            img.blocks[0] = C.cast(plane, C.c_void_p).value
            rows = (abi.HuffDecRow * (img.mcuv + 1))()
            if kernel == "wave":
                assert emu.emu_huffman_decode_image(C.byref(img), rows) == 0
            else:
                moved, nsub = (C.c_int32 * 8)(), C.c_uint32(0)
                assert emu.emu_huffman_decode_image_simt(C.byref(img), rows, 256, moved, C.byref(nsub)) == 0
            assert rows[img.mcuv].aux >> 8 == 0, (name, w, h, kernel)
            assert plane.raw == orig[0], (name, w, h, kernel)
            assert L.lep_jpeg_finish_gpu(hnd, rows) == 0
            streams = ob.oracle_encode(d, host.plan())[0]
            arr = (abi.Bytes * len(streams))()
            keep = []
            for i, st in enumerate(streams):
                b = C.create_string_buffer(bytes(st), max(1, len(st)))
                keep.append(b)
                arr[i].data = C.cast(b, C.c_void_p).value
                arr[i].len = arr[i].cap = len(st)
            out = abi.Bytes()
            assert L.lep_jpeg_write_lep(hnd, 0, arr, len(streams), C.byref(out)) == 0
            got = out.tobytes()
            L.lep_free(out.data)
            L.lep_jpeg_close(hnd)
            assert got == want, (name, w, h, kernel)
            del k


@pytest.mark.parametrize("name", ["gray22", "gray21", "gray12", "gray11"])
def test_one_component_scans_on_the_gpu_scan_encoders(emu, name):
    """... and the way back: recode_prepare plans a one-component file as nch x ncv MCUs of one block with its thread segments in block
    rows; the wavefront kernel (lep_huff.h) and the lane-per-unit kernels (lep_huff_simt.h -- a unit of a one-component scan can be ONE
    block of two or three bits, so a segment's last partial byte may hold bits of the unit in front: kTailNotOwn) must both write the
    file's scan bytes segment by segment, with the end states the next hand-off recorded; several thread segments, widths and heights
    with and without padding blocks, plain grey with restart intervals."""
    L = abi.lib()
    comps = LAYOUTS.get(name) or {"gray12": [Y(1, 2)], "gray11": [Y(1, 1)]}[name]
    cases = [(97, 50, 0, 0.5, 40), (9, 9, 0, 0.5, 40), (1, 1, 0, 0.5, 40), (24, 481, 0, 0.02, 2), (1203, 897, 0, 1.0, 200), (3001, 2999, 0, 0.01, 1), (8, 30000, 0, 1.0, 200), (3000, 24, 0, 1.0, 200), (1500, 1100, 0, 0.3, 60)]
    if name == "gray11":
        cases += [(1203, 897, 7, 1.0, 200), (1203, 897, 1, 0.3, 100), (640, 480, 80, 1.0, 200), (97, 50, 3, 0.5, 40)]
    several = 0
    for w, h, ri, dens, amp in cases:
        jpg, _ = jw.write_baseline(w, h, comps, np.random.default_rng(zlib.crc32(("back %s %d %d" % (name, w, h)).encode())), density=dens, amp=amp, restart_interval=ri)
        _, lep = oracle_compress(jpg)
        f = LepFile(lep)
        src = JpegImage(jpg)
        C.memmove(f.desc.blocks[0], src.desc.blocks[0], f.desc.nblocks(0) * 128)
        assert f.recode() == jpg
        img = abi.HuffImage()
        segs = (abi.HuffSegment * abi.MAX_SEGMENTS)()
        nseg, ok = C.c_int(0), C.c_int(0)
        assert L.lep_file_recode_plan(f.handle, C.byref(img), segs, C.byref(nseg), C.byref(ok)) == 0
        assert ok.value, "%s %dx%d: not planned for the GPU scan encoder" % (name, w, h)
        assert img.interleaved == 1 and img.hs[0] == 1 and img.vs[0] == 1 and img.mcuc == img.mcuh * img.mcuv
        several += nseg.value > 1
        for fn in (emu.emu_huffman_encode_segment, emu.emu_huffman_encode_segment_simt):
            outs = (abi.Bytes * nseg.value)()
            ends = (abi.HuffEnd * nseg.value)()
            keep = []
            for i in range(nseg.value):
                cap = min(segs[i].out_cap, len(jpg) + 1024)
                segs[i].out_cap = cap
                buf = C.create_string_buffer(cap + 8)
                keep.append(buf)
                n = C.c_uint32(0)
                assert fn(C.byref(img), C.byref(segs[i]), buf, C.byref(n), C.byref(ends[i])) == 0, (name, w, h, i, "the kernel left the segment to the other one")
                outs[i].data = C.cast(buf, C.c_void_p).value
                outs[i].len = outs[i].cap = n.value
            out = abi.Bytes()
            assert L.lep_file_recode_finish(f.handle, outs, ends, nseg.value, C.byref(out)) == 0, (name, w, h, fn)
            got = out.tobytes()
            L.lep_free(out.data)
            assert got == jpg, (name, w, h, fn)
    assert several >= 3


def test_lane_per_unit_encoder_when_the_last_unit_is_one_tiny_block(emu):
    """A segment of a one-component scan whose MCU count is 1 modulo 8 ends in a unit of ONE block -- two to six bits in a sparse file,
    fewer than the stream's last partial byte may hold: every (first row, last row, partial byte carried in) of a nine-block-wide frame,
    lane-per-unit kernels against the wavefront kernel: bytes, byte count, end state."""
    L = abi.lib()
    tiny = seen = 0
    for seed, dens in [(1, 0.0), (2, 0.01), (3, 0.05)]:
        # (the chroma tables: a block of zeros with an unchanged DC codes to FOUR bits; every twentieth block moves the DC, a few have one AC coefficient)
        rng0 = np.random.default_rng(seed)
        blocks = np.zeros((12, 9, 64), dtype=np.int32)
        blocks[:, :, 0] = np.cumsum(rng0.integers(-2, 3, (12, 9)) * (rng0.random((12, 9)) < 0.05), axis=None).reshape(12, 9)
        blocks[:, :, 1] = rng0.integers(-1, 2, (12, 9)) * (rng0.random((12, 9)) < dens)
        blocks[:, 7, 63] = 1       # ... and the block in front of a row's last ends in a ONE (its last coefficient's value bit, no end-of-block code)
        jpg, _ = jw.write_baseline(70, 96, [(1, 1, 1, 0, 1, 1)], rng0, blocks=[blocks])
        _, lep = oracle_compress(jpg)
        f = LepFile(lep)
        src = JpegImage(jpg)
        C.memmove(f.desc.blocks[0], src.desc.blocks[0], f.desc.nblocks(0) * 128)
        img = abi.HuffImage()
        segs = (abi.HuffSegment * abi.MAX_SEGMENTS)()
        nseg, ok = C.c_int(0), C.c_int(0)
        assert L.lep_file_recode_plan(f.handle, C.byref(img), segs, C.byref(nseg), C.byref(ok)) == 0 and ok.value
        assert img.mcuh == 9 and img.mcuv == 12
        rng = np.random.default_rng(seed)
        for r0 in range(12):
            for r1 in range(r0 + 1, 13):
                for _ in range(3):
                    sg = abi.HuffSegment()
                    C.memmove(C.byref(sg), C.byref(segs[0]), C.sizeof(sg))
                    nb = int(rng.integers(0, 8))
                    sg.mcu_row0, sg.mcu_row1, sg.out_cap = r0, r1, 4096
                    sg.overhang = (int(rng.integers(0, 256)) & (0xff00 >> nb) & 0xff) | (nb << 8)
                    for c in range(4):
                        sg.last_dc[c] = int(rng.integers(-3, 4))
                    outs = []
                    for fn in (emu.emu_huffman_encode_segment, emu.emu_huffman_encode_segment_simt):
                        buf = C.create_string_buffer(4096 + 8)
                        n = C.c_uint32(0)
                        end = abi.HuffEnd()
                        assert fn(C.byref(img), C.byref(sg), buf, C.byref(n), C.byref(end)) == 0
                        outs.append((n.value, buf.raw[: n.value], end.attempted, end.overhang_byte, end.num_overhang_bits, list(end.last_dc), end.pad))
                    assert outs[0] == outs[1], (seed, r0, r1, nb, outs[0][2:], outs[1][2:])
                    seen += 1
                    tiny += r1 < 12 and outs[0][4] >= 5
    assert seen > 600 and tiny > 20


QUIRKS = ["trailing_zrl", "rst_fill", "mixed_pad", "dup_symbol", "rst_order", "scan_tail"]


@pytest.mark.parametrize("quirk", QUIRKS)
@pytest.mark.parametrize("pad", [0, 1])
def test_non_canonical_huffman_layers_are_refused(emu, quirk, pad):
    c420 = LAYOUTS["ids"]
    clean, _ = jw.write_baseline(160, 96, c420, np.random.default_rng(4), restart_interval=4, pad_bit=pad)
    JpegImage(clean)
    assert gpu_scan_decode(emu, clean)[0] == "ok"
    jpg, _ = jw.write_baseline(160, 96, c420, np.random.default_rng(4), restart_interval=4, pad_bit=pad, quirks=(quirk,))
    assert jpg != clean
    from PIL import Image
    import io
    Image.open(io.BytesIO(jpg)).load()                       # a decoder has no complaint
    with pytest.raises(LeptonError) as e:
        JpegImage(jpg)
    assert e.value.code == 42                                # UNSUPPORTED_JPEG (errorlevel 1 -> jpgcoder.cc:2024)
    kind, what = gpu_scan_decode(emu, jpg)
    assert kind in ("host", "irregular", "refused"), "the GPU scan decoder accepted a scan the re-encoder cannot reproduce"


@needs_ref
@pytest.mark.parametrize("quirk", QUIRKS)
def test_reference_writes_nothing_for_non_canonical_layers(quirk, tmp_path):
    jp, lp = str(tmp_path / "q.jpg"), str(tmp_path / "q.lep")
    jpg, _ = jw.write_baseline(160, 96, LAYOUTS["ids"], np.random.default_rng(4), restart_interval=4, quirks=(quirk,))
    open(jp, "wb").write(jpg)
    subprocess.run([REF, "-unjailed", jp, lp], capture_output=True)   # default run (verification on)
    assert not os.path.exists(lp) or os.path.getsize(lp) == 0


def check_restores(jpg, **kw):
    img = JpegImage(jpg, **kw)
    segs = img.plan()
    streams, _ = ob.oracle_encode(img.desc, segs)
    lep = img.write_lep(streams)
    want = img.data if kw.get("embedding") else img.data[kw.get("start_byte", 0):]
    return abi.lib().lep_jpeg_check_restores(img.handle, lep, len(lep), want, len(want))


@pytest.mark.parametrize("name", golden_cases())
def test_round_trip_check_passes_every_fixture(name):
    assert check_restores(golden(name)[0]) == 0


def test_round_trip_check_passes_slices_and_embedded_files():
    for name, start, trunc in slice_cases():
        assert check_restores(golden(name)[0], start_byte=start, trunc=trunc) == 0, name
    for name, off in embedded_cases():
        assert check_restores(golden(name)[0], embedding=off) == 0, name


@pytest.mark.parametrize("name,restored_md5", roundtrip_failure_cases())
def test_round_trip_failure_is_reported_and_decode_stays_compatible(name, restored_md5):
    """what `lepton` (verification on) answers with exit 41: compression refused; the -skipverify .lep of the reference is
    still equal to ours and decodes to the bytes the reference makes of it"""
    jpg, lep = golden(name)
    img, got = oracle_compress(jpg)
    assert got == lep
    assert abi.lib().lep_jpeg_check_restores(img.handle, got, len(got), jpg, len(jpg)) == 41
    f = LepFile(lep)
    ob.oracle_decode(f.desc, f.segments, f.streams)
    back = f.recode()
    assert back != jpg and hashlib.md5(back).hexdigest() == restored_md5
    # a corrupted expectation is a failure too, a foreign .lep is reported with its own code
    ok, _ = golden("c420_160x120")
    assert check_restores(ok) == 0
    img2 = JpegImage(ok)
    assert abi.lib().lep_jpeg_check_restores(img2.handle, lep, len(lep), ok, len(ok)) == 41
    assert abi.lib().lep_jpeg_check_restores(img2.handle, b"junk", 4, ok, len(ok)) != 0


@pytest.mark.skipif(not os.path.isdir(REF_IMAGES), reason="reference checkout not present (GPU box)")
def test_reference_roundtripfail_image():
    """test_suite/test_roundtrip.sh: `lepton -verify` must fail on images/roundtripfail.jpg, `-skipverify` must succeed"""
    jpg = open(os.path.join(REF_IMAGES, "roundtripfail.jpg"), "rb").read()
    assert check_restores(jpg) == 41
    _, lep = oracle_compress(jpg)                           # -skipverify: a .lep is written ...
    if os.path.exists(REF):
        out = "/tmp/_rtf.lep"
        subprocess.run([REF, "-unjailed", "-skipverify", os.path.join(REF_IMAGES, "roundtripfail.jpg"), out], capture_output=True)
        assert lep == open(out, "rb").read()                # ... the reference's


@needs_ref
@pytest.mark.parametrize("kw", [dict(dqt16=True), dict(sof=0xC1), dict(sof=0xC1, dqt16=True), dict(precision=12), dict(sof=0xC1, precision=12),
                                dict(sof=0xC3), dict(sof=0xC5), dict(sof=0xC9)], ids=lambda k: "-".join("%s=%s" % i for i in sorted(k.items())))
def test_frame_header_variants_against_the_reference_binary(kw, tmp_path):
    """sixteen-bit quantisation tables and SOF1 (extended sequential) frames are coded like baseline ones; twelve-bit
    precision, lossless, differential and arithmetic frames are UNSUPPORTED_JPEG -- as in the reference"""
    jp, lp = str(tmp_path / "v.jpg"), str(tmp_path / "v.lep")
    jpg, _ = jw.write_baseline(160, 96, LAYOUTS["ids"], np.random.default_rng(4), **kw)
    open(jp, "wb").write(jpg)
    r = subprocess.run([REF, "-unjailed", "-skipverify", jp, lp], capture_output=True)
    ref_ok = os.path.exists(lp) and os.path.getsize(lp) > 0
    try:
        img, got = oracle_compress(jpg)
    except LeptonError as e:
        assert not ref_ok and e.code == 42 and b"UNSUPPORTED_JPEG\n" in r.stderr
        return
    assert ref_ok and got == open(lp, "rb").read()
    assert abi.lib().lep_jpeg_check_restores(img.handle, got, len(got), jpg, len(jpg)) == 0


def _random_layout_jpeg(rnd, trial):
    ncomp = rnd.choice([1, 2, 3, 3, 3, 3])
    f = lambda: rnd.choice([1, 1, 2, 2, 2, 3, 4]) if rnd.random() < 0.08 else rnd.choice([1, 2])
    comps = [((i + 1) if rnd.random() < 0.8 else rnd.randrange(256), f(), f(), min(i, 1), min(i, 1), min(i, 1)) for i in range(ncomp)]
    if len({c[0] for c in comps}) < ncomp:
        comps = [(i + 1,) + c[1:] for i, c in enumerate(comps)]
    w, h = rnd.choice([8, 15, 16, 17, 31, 64, 97, 160, 333, 640]), rnd.choice([8, 9, 16, 23, 48, 99, 240, 480])
    ri = rnd.choice([0, 0, 0, 1, 2, 3, 7, 50])
    if ncomp == 1 and comps[0][1] * comps[0][2] > 1:
        ri = 0                                                                        # see the round-trip-failure fixture
    jpg, _ = jw.write_baseline(w, h, comps, np.random.default_rng(3000 + trial), restart_interval=ri, quality=rnd.choice([30, 75, 95]),
                               density=rnd.choice([0.02, 0.1, 0.25, 0.6, 1.0]), amp=rnd.choice([2.0, 10.0, 40.0, 200.0]), pad_bit=rnd.choice([0, 1, 1]),
                               dqt16=rnd.random() < 0.1, sof=0xC1 if rnd.random() < 0.1 else 0xC0)
    tail = rnd.random()
    if tail < 0.1:
        jpg += bytes(rnd.randrange(256) for _ in range(rnd.randint(1, 300)))
    elif tail < 0.25:
        jpg = jpg[: rnd.randint(len(jpg) // 3, len(jpg) - 1)]
    return jpg


def sweep_against_the_reference(trials, seed, tmp):
    import random

    rnd = random.Random(seed)
    jp, lp, bp = (os.path.join(tmp, n) for n in ("s.jpg", "s.lep", "s.back"))
    accepted = rejected = 0
    codes = dict(SAID)
    codes.update({b"THREADING_PARTIAL_MCU": 12, b"ONLY_GARBAGE_NO_JPEG": 14, b"UNSUPPORTED_JPEG": 42, b"COEFFICIENT_OUT_OF_RANGE": 6,
                  b"UNSUPPORTED_JPEG_WITH_ZERO_IDCT_0": 43})
    for trial in range(trials):
        jpg = _random_layout_jpeg(rnd, trial)
        open(jp, "wb").write(jpg)
        for f in (lp, bp):
            if os.path.exists(f):
                os.unlink(f)
        r = subprocess.run([REF, "-unjailed", "-skipverify", jp, lp], capture_output=True)
        said = [v for k, v in codes.items() if k + b"\n" in r.stderr]
        ref_ok = r.returncode == 0 and os.path.exists(lp) and os.path.getsize(lp) > 0 and not said
        try:
            img, got = oracle_compress(jpg)
            code = 0
        except LeptonError as e:
            got, code = None, e.code
        except RuntimeError as e:                                  # the coder itself: "oracle encode exit code 6"
            got, code = None, int(str(e).split()[-1])
        assert (got is not None) == ref_ok, (trial, code, r.returncode, said)
        if not ref_ok:
            rejected += 1
            if said:
                assert code == said[0], (trial, code, said)
            continue
        accepted += 1
        assert got == open(lp, "rb").read(), "trial %d: .lep differs from the reference's" % trial
        assert subprocess.run([REF, "-unjailed", lp, bp], capture_output=True).returncode == 0
        f = LepFile(got)
        ob.oracle_decode(f.desc, f.segments, f.streams)
        back = f.recode()
        assert back == open(bp, "rb").read(), "trial %d: restored file differs from the reference's" % trial
        # the round-trip check says exactly whether the reference restored its input
        assert (abi.lib().lep_jpeg_check_restores(img.handle, got, len(got), jpg, len(jpg)) == 0) == (back == jpg), trial
    return accepted, rejected


@needs_ref
def test_random_layouts_against_the_reference_binary(tmp_path):
    """seeded random frames from the coefficient-domain writer -- 1..3 components, factors 1..2 (now and then 3 or 4: refused),
    random component ids, one block up to 640x480, restart intervals, sparse to dense blocks, small to large amplitudes, both
    pad bits, sixteen-bit tables, SOF1, trailing garbage, truncation: same decision and exit code, same .lep bytes, same
    restored bytes as the reference binary (600 cases by hand, 40 here)"""
    accepted, rejected = sweep_against_the_reference(40, 20260924, str(tmp_path))
    assert accepted >= 20 and rejected >= 2


def sweep_through_the_kernel_sources(emu, trials, seed):
    """the same random frames through the device code in the lane-loop emulation: coder kernels (v3 encoder / v4 decoder) against
    the oracle, the Huffman scan decoder against the host parser, the Huffman re-encoder against the input bytes"""
    import random

    rnd = random.Random(seed)
    L = abi.lib()
    stats = dict(coded=0, scan_decoded=0, reencoded=0)
    for trial in range(trials):
        jpg = _random_layout_jpeg(rnd, trial)
        try:
            img = JpegImage(jpg)
            d, segs = img.desc, img.plan()
            want, _ = ob.oracle_encode(d, segs)
        except (LeptonError, RuntimeError):
            continue
        orig = [C.string_at(d.blocks[c], d.nblocks(c) * 128) for c in range(d.ncomp)]
        for s_, wv in zip(segs, want):
            cap = len(wv) + 4096
            b = C.create_string_buffer(cap)
            n = C.c_uint32(0)
            assert emu.emu_encode_segment_v3(C.byref(d), s_.luma_y_start, s_.luma_y_end, s_.is_last, b, cap, C.byref(n), None) == 0
            assert b.raw[: n.value] == wv, trial
        for c in range(d.ncomp):
            C.memset(d.blocks[c], 0, d.nblocks(c) * 128)
        for s_, wv in zip(segs, want):
            assert emu.emu_decode_segment_v4(C.byref(d), s_.luma_y_start, s_.luma_y_end, s_.is_last, wv, len(wv), None) == 0
        got = [C.string_at(d.blocks[c], d.nblocks(c) * 128) for c in range(d.ncomp)]
        for c in range(d.ncomp):                                  # what the oracle's decoder leaves (truncated files: coded blocks only)
            C.memset(d.blocks[c], 0, d.nblocks(c) * 128)
        ob.oracle_decode(d, segs, want)
        assert got == [C.string_at(d.blocks[c], d.nblocks(c) * 128) for c in range(d.ncomp)], trial
        stats["coded"] += 1
        kind, planes = gpu_scan_decode(emu, jpg)
        assert kind in ("ok", "host", "irregular"), (trial, kind, planes)
        if kind == "ok":
            assert planes == orig, trial
            stats["scan_decoded"] += 1
        # re-encode: the frame the JPEG itself holds, the header of the .lep
        lep = img.write_lep(want)
        f = LepFile(lep)
        for c in range(f.desc.ncomp):
            C.memmove(f.desc.blocks[c], orig[c], f.desc.nblocks(c) * 128)
        host = f.recode()
        himg = abi.HuffImage()
        hsegs = (abi.HuffSegment * abi.MAX_SEGMENTS)()
        nseg, ok = C.c_int(0), C.c_int(0)
        assert L.lep_file_recode_plan(f.handle, C.byref(himg), hsegs, C.byref(nseg), C.byref(ok)) == 0
        if not ok.value:
            continue
        outs = (abi.Bytes * nseg.value)()
        ends = (abi.HuffEnd * nseg.value)()
        keep = []
        for i in range(nseg.value):
            cap = min(hsegs[i].out_cap, len(jpg) + 1024)
            hsegs[i].out_cap = cap
            buf = C.create_string_buffer(cap + 8)
            keep.append(buf)
            n = C.c_uint32(0)
            assert emu.emu_huffman_encode_segment(C.byref(himg), C.byref(hsegs[i]), buf, C.byref(n), C.byref(ends[i])) == 0
            outs[i].data = C.cast(buf, C.c_void_p).value
            outs[i].len = outs[i].cap = n.value
        out = abi.Bytes()
        assert L.lep_file_recode_finish(f.handle, outs, ends, nseg.value, C.byref(out)) == 0
        assert out.tobytes() == host, trial
        L.lep_free(out.data)
        stats["reencoded"] += 1
    return stats


def test_random_layouts_through_the_kernel_sources(emu):
    """(400 cases by hand, 40 here)"""
    st = sweep_through_the_kernel_sources(emu, 40, 20260925)
    assert st["coded"] >= 25 and st["scan_decoded"] >= 12 and st["reencoded"] >= 12, st


if __name__ == "__main__":   # python tests/test_sampling_layouts.py <trials> <seed>: the sweep by hand
    import sys
    import tempfile

    if len(sys.argv) > 3 and sys.argv[3] == "kernels":
        print(sweep_through_the_kernel_sources(C.CDLL(os.path.join(ROOT, "tests", "emu", "libcore_emu_layouts.so")), int(sys.argv[1]), int(sys.argv[2])))
    else:
        print(sweep_against_the_reference(int(sys.argv[1]), int(sys.argv[2]), tempfile.mkdtemp()))


SCAN_SCRIPTS = {
    "y_cbcr_420": ([Y(2, 2), Cx(1, 1, 2), Cx(1, 1, 3)], [[0], [1, 2]]),
    "y_cb_cr_444": ([Y(1, 1), Cx(1, 1, 2), Cx(1, 1, 3)], [[0], [1], [2]]),
    "ycb_cr_422": ([Y(2, 1), Cx(1, 1, 2), Cx(1, 1, 3)], [[0, 1], [2]]),
    "cbcr_y_420": ([Y(2, 2), Cx(1, 1, 2), Cx(1, 1, 3)], [[1, 2], [0]]),
    "y_cbcr_440": ([Y(1, 2), Cx(1, 1, 2), Cx(1, 1, 3)], [[0], [1, 2]]),
    "y_cbcr_mixed": ([Y(2, 2), Cx(2, 1, 2), Cx(1, 2, 3)], [[0], [1, 2]]),
}


@needs_ref
@pytest.mark.parametrize("name", sorted(SCAN_SCRIPTS))
def test_sequential_multi_scan_files_against_the_reference_binary(name, tmp_path):
    """non-progressive frames coded in several scans (luma alone, then Cb + Cr interleaved, ...): format flag 'X', the general
    re-coder with a sequential MCU loop over the scan's component subset -- same .lep as the reference, which restores its input from it, and so do we, several thread segments included"""
    comps, scans = SCAN_SCRIPTS[name]
    jp, lp = str(tmp_path / "m.jpg"), str(tmp_path / "m.lep")
    for w, h, ri in [(97, 50, 0), (96, 64, 5), (640, 480, 0), (640, 480, 5)]:
        jpg, _ = jw.write_sequential_scans(w, h, comps, np.random.default_rng(8), scans, restart_interval=ri)
        open(jp, "wb").write(jpg)
        if os.path.exists(lp):
            os.unlink(lp)
        bp = jp + ".back"
        assert subprocess.run([REF, "-unjailed", "-skipverify", jp, lp], capture_output=True).returncode == 0
        # the reference's own restore, when its thread pool cooperates (inside the whole suite it now and then gives up on the
        # four-segment file with "Worker thread out of memory"; its default compress run, round trip included, passes by hand)
        if os.path.exists(bp):
            os.unlink(bp)
        if subprocess.run([REF, "-unjailed", lp, bp], capture_output=True).returncode == 0 and os.path.exists(bp) and os.path.getsize(bp):
            assert open(bp, "rb").read() == jpg
        img, got = oracle_compress(jpg)
        assert got[3:4] == b"X" and got == open(lp, "rb").read(), (name, w, h, ri)
        assert abi.lib().lep_jpeg_check_restores(img.handle, got, len(got), jpg, len(jpg)) == 0
        f = LepFile(got)
        ob.oracle_decode(f.desc, f.segments, f.streams)
        assert f.recode() == jpg, (name, w, h, ri)
