"""The reference's OWN corpus -- images/*.jpg and the known-answer .lep files its test-suite is built on (Makefile.am:277-353:
test_iphone, test_SLR, test_misc, test_trailing_header, test_truncate, test_single_row_truncate, test_odd_rst, test_trailing_rst,
test_truncated_zero_run, test_nofsync, test_gray2sf, test_colorswap, test_progressive, test_arithmetic_failfast, test_bad_zero_run;
test_suite/test_16threads.sh, test_legacy.sh, test_future_compat.sh, test_roundtrip.sh) -- from tests/golden/ref/ (written by
tests/golden/make_golden_ref.py with the real reference binary), so that the same files go through

 * the CPU oracle and the kernel sources single-stepped on the CPU (`-m "not gpu"`, here), and
 * the HIP path on the MI355X through the C ABI (`-m gpu`): per file, through the batch pipeline, and the refusals.

Bar: bit-exact -- `compress(jpg) == the reference's .lep`, `decompress(.lep) == what the reference restores` (== the jpg, except for
roundtripfail.jpg, which the reference itself restores wrongly and refuses with 41 when verifying)."""
import ctypes as C
import hashlib
import os

import pytest

import oracle_binding as ob
from conftest import REF_GOLDEN, ROOT, ref_cases, ref_golden, ref_manifest, ref_refused_cases
from lepton_amd.codec import JpegImage, LepFile, LeptonError, lep_stream

MAN = ref_manifest()
BIG = {k for k, v in MAN["jpegs"].items() if v["jpg_size"] > 1000000}   # iphone, iphonecity, slrhills, slrindoor (+ arithmetic)


def md5(b):
    return hashlib.md5(b).hexdigest()


# ------------------------------------------------------------------------------------------------ CPU: oracle + host container code
@pytest.mark.parametrize("name", ref_cases())
def test_oracle_writes_the_reference_s_lep(name):
    jpg, lep = ref_golden(name)
    img = JpegImage(jpg)
    segs = img.plan()
    streams, _ = ob.oracle_encode(img.desc, segs)
    assert img.write_lep(streams) == lep


@pytest.mark.parametrize("name", ref_cases())
def test_oracle_restores_what_the_reference_restores(name):
    _, lep = ref_golden(name)
    f = LepFile(lep)
    ob.oracle_decode(f.desc, f.segments, f.streams)
    assert md5(f.recode()) == MAN["jpegs"][name]["restored_md5"]


@pytest.mark.parametrize("lep", sorted(MAN["known_answers"]))
def test_oracle_known_answer_files(lep):
    """iphone16.lep (16 thread segments), gold-legacy.lep (34 segments, the legacy hand-offs), narrowrst.lep (format 4, brotli header)"""
    out = b""
    for f in lep_stream(open(os.path.join(REF_GOLDEN, "known_" + lep), "rb").read()):
        ob.oracle_decode(f.desc, f.segments, f.streams)
        out += f.recode()
    assert md5(out) == MAN["known_answers"][lep]["restored_md5"]


@pytest.mark.parametrize("name,code", ref_refused_cases())
def test_host_parser_refuses_what_the_reference_refuses(name, code):
    """arithmetic.jpg: UNSUPPORTED_JPEG (42); badzerorun.jpg: the reference's always_assert (ASSERTION_FAILURE, 1; its debug build
    aborts) -- a zero run past the end of a block in a file that is not truncated"""
    jpg = open(os.path.join(REF_GOLDEN, name + ".jpg"), "rb").read()
    with pytest.raises(LeptonError) as e:
        JpegImage(jpg)
    assert e.value.code == (code if code > 0 else 1)


@pytest.mark.parametrize("name", ref_cases(progressive=True))
def test_progressive_files_need_the_flag(name):
    """test_progressive_disallowed: without -allowprogressive a progressive scan ends in PROGRESSIVE_UNSUPPORTED (8),
    jpgcoder.cc:2911-2925.  (The reference BINARY's exit status for this is not reproducible -- 8, 0 or 41 from run to run on the same
    file, its worker threads racing its exit handler -- so the manifest does not record it; the source is the authority here.)"""
    jpg, _ = ref_golden(name)
    with pytest.raises(LeptonError) as e:
        JpegImage(jpg, allow_progressive=False)
    assert e.value.code == 8


# ------------------------------------------------------------------------------------------------ CPU: the kernel sources, single-stepped
EMU_SO = os.path.join(ROOT, "tests", "emu", "libcore_emu_ref.so")


@pytest.fixture(scope="module")
def emu():
    import subprocess

    tmp = "%s.%d" % (EMU_SO, os.getpid())
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", tmp, os.path.join(ROOT, "tests", "emu", "core_emu.cc")])
    os.replace(tmp, EMU_SO)
    return C.CDLL(EMU_SO)


@pytest.mark.parametrize("name", [n for n in ref_cases() if n not in BIG] + ["iphone"])
def test_kernel_sources_on_cpu_code_the_reference_s_images(emu, name):
    """lep_enc3.h, lep_enc5.h (the walks' two halves apart) and lep_dec4.h as 64-lane loop emulations over the reference's photographs:
    every thread segment's stream == the oracle's (whose container == the reference's .lep, above), and the decoder returns the frame"""
    jpg, _ = ref_golden(name)
    img = JpegImage(jpg)
    d = img.desc
    segs = img.plan()
    want, bins = ob.oracle_encode(d, segs)
    for s, w in zip(segs, want):
        cap = len(w) + 4096
        for entry, extra in (("emu_encode_segment_v3", ()), ("emu_encode_segment_v5_halves", (None, 0))):
            buf = C.create_string_buffer(cap)
            n, nb = C.c_uint32(0), C.c_uint32(0)
            assert getattr(emu, entry)(C.byref(d), s.luma_y_start, s.luma_y_end, s.is_last, buf, cap, C.byref(n), C.byref(nb), *extra) == 0
            assert buf.raw[: n.value] == w, entry
    orig = [C.string_at(d.blocks[c], d.nblocks(c) * 128) for c in range(d.ncomp)]
    for c in range(d.ncomp):
        C.memset(d.blocks[c], 0, d.nblocks(c) * 128)
    total = 0
    for s, w in zip(segs, want):
        nb = C.c_uint32(0)
        assert emu.emu_decode_segment_v4(C.byref(d), s.luma_y_start, s.luma_y_end, s.is_last, w, len(w), C.byref(nb)) == 0
        total += nb.value
    assert total == bins
    for c in range(d.ncomp):
        n = d.coded_blocks[c] * 128
        assert C.string_at(d.blocks[c], n) == orig[c][:n]


# ------------------------------------------------------------------------------------------------ MI355X: the HIP path through the C ABI
@pytest.mark.gpu
@pytest.mark.parametrize("name", ref_cases())
def test_gpu_compress_equals_the_reference_s_lep(gpu_codec, name):
    jpg, lep = ref_golden(name)
    if MAN["jpegs"][name]["default_exit"] == 41:   # roundtripfail.jpg: lep_compress verifies like the reference's default run
        with pytest.raises(LeptonError) as e:
            gpu_codec.compress(jpg)
        assert e.value.code == 41
        got, status, _ = gpu_codec.compress_batch([jpg], verify=False)
        assert status == [0] and got[0] == lep
    else:
        assert gpu_codec.compress(jpg) == lep


@pytest.mark.gpu
@pytest.mark.parametrize("name", ref_cases())
def test_gpu_decompress_restores_what_the_reference_restores(gpu_codec, name):
    jpg, lep = ref_golden(name)
    got = gpu_codec.decompress(lep)
    assert md5(got) == MAN["jpegs"][name]["restored_md5"]
    if MAN["jpegs"][name]["restored_equals_input"]:
        assert got == jpg


@pytest.mark.gpu
@pytest.mark.parametrize("lep", sorted(MAN["known_answers"]))
def test_gpu_known_answer_files(gpu_codec, lep):
    """test_16threads.sh / test_legacy.sh / test_future_compat.sh through the GPU decoder: per file and through the batch pipeline"""
    data = open(os.path.join(REF_GOLDEN, "known_" + lep), "rb").read()
    want = MAN["known_answers"][lep]["restored_md5"]
    assert md5(gpu_codec.decompress(data)) == want
    got, status, _ = gpu_codec.decompress_batch([data])
    assert status == [0] and md5(got[0]) == want


@pytest.mark.gpu
def test_gpu_streams_of_the_reference_s_images_equal_the_oracle(gpu_codec):
    """every thread segment of every baseline reference image in ONE launch (mixed geometries: 4:2:0 / 4:2:2 / 4:4:4 / grey / 2x2 grey,
    restart intervals, truncations): streams == the oracle's"""
    names = ref_cases(progressive=False)
    imgs = [JpegImage(ref_golden(n)[0]) for n in names]
    plans = [im.plan() for im in imgs]
    got = gpu_codec.encode(imgs, plans)
    for n, im, p, g in zip(names, imgs, plans, got):
        want, _ = ob.oracle_encode(im.desc, p)
        assert g == want, n


@pytest.mark.gpu
@pytest.mark.parametrize("host_huffman", [False, True])
def test_gpu_batch_pipeline_on_the_reference_s_corpus(gpu_codec, host_huffman):
    """all of images/*.jpg in one lep_compress_batch call (the refused ones among them) and all the .lep files in one
    lep_decompress_batch call, with the GPU scan kernels and with the host Huffman coders: every file byte-equal to the reference's
    answer, every refusal with the reference's exit code, neighbours untouched"""
    ok = ref_cases()
    refused = ref_refused_cases()
    names = ok[:5] + [refused[0][0]] + ok[5:12] + [r[0] for r in refused[1:]] + ok[12:]
    jpgs = [open(os.path.join(REF_GOLDEN, n + ".jpg"), "rb").read() for n in names]
    want_status = {n: 0 for n in ok}
    want_status.update({n: (c if c > 0 else 1) for n, c in refused})
    got, status, stats = gpu_codec.compress_batch(jpgs, verify=False, host_huffman=host_huffman)
    assert status == [want_status[n] for n in names]
    for n, g in zip(names, got):
        if want_status[n] == 0:
            assert g == ref_golden(n)[1], n
        else:
            assert g is None
    # with verification on, the one file the reference cannot restore is refused with 41 and nothing else changes
    got_v, status_v, _ = gpu_codec.compress_batch(jpgs, verify=True, host_huffman=host_huffman)
    for n, g, s, g0 in zip(names, got_v, status_v, got):
        if MAN["jpegs"][n].get("default_exit") == 41 and want_status[n] == 0:
            assert s == 41 and g is None
        else:
            assert s == want_status[n] and g == g0, n
    leps = [ref_golden(n)[1] for n in ok] + [open(os.path.join(REF_GOLDEN, "known_" + k), "rb").read() for k in sorted(MAN["known_answers"])]
    md5s = [MAN["jpegs"][n]["restored_md5"] for n in ok] + [MAN["known_answers"][k]["restored_md5"] for k in sorted(MAN["known_answers"])]
    back, status, _ = gpu_codec.decompress_batch(leps, host_huffman=host_huffman)
    assert status == [0] * len(leps)
    assert [md5(b) for b in back] == md5s


@pytest.mark.gpu
@pytest.mark.parametrize("name,code", ref_refused_cases())
def test_gpu_refuses_what_the_reference_refuses(gpu_codec, name, code):
    jpg = open(os.path.join(REF_GOLDEN, name + ".jpg"), "rb").read()
    with pytest.raises(LeptonError) as e:
        gpu_codec.compress(jpg)
    assert e.value.code == (code if code > 0 else 1)


@pytest.mark.gpu
def test_gpu_scan_kernels_take_files_cut_inside_their_scan(gpu_codec):
    """the file `lepton -benchmark` codes (src/lepton/benchmark.cc:116-119: no EOI, the scan ends in mid-image) and the truncated fixtures
    through the batch pipeline: the GPU scan decoder / encoder take them (lep_batch_stats.gpu_huffman_files), the .lep is the per-file
    path's (== the reference's for the fixtures), the file comes back byte for byte"""
    import bench
    from conftest import golden

    rb = bench.reference_benchmark_jpeg()
    cut = [rb, golden("truncated")[0], golden("truncated_short")[0], ref_golden("iphone")[0][:1500000], ref_golden("androidcrop")[0][:60000]]
    whole = [golden("c420_160x120")[0]]
    jpgs = cut + whole + cut[:1] * 3
    leps, status, cs = gpu_codec.compress_batch(jpgs, verify=False)
    assert status == [0] * len(jpgs)
    assert cs["gpu_huffman_files"] >= len(jpgs) - 2, cs      # (a cut whose last block is irregular may be the host parser's)
    assert leps[1] == golden("truncated")[1] and leps[2] == golden("truncated_short")[1] and leps[5] == golden("c420_160x120")[1]
    assert leps[0] == gpu_codec.compress(rb) and leps[6] == leps[0]
    back, status, ds = gpu_codec.decompress_batch(leps)
    assert status == [0] * len(jpgs) and back == jpgs
    assert ds["gpu_huffman_files"] >= len(jpgs) - 2, ds
    # and with the host Huffman coders: the same bytes
    leps_h, status, _ = gpu_codec.compress_batch(jpgs, verify=False, host_huffman=True)
    assert status == [0] * len(jpgs) and leps_h == leps


@pytest.mark.gpu
def test_gpu_lane_per_restart_interval_scan_decode_in_the_compress_pipeline(gpu_codec, monkeypatch):
    """files with restart intervals through lep_compress_batch: the lane-per-interval form of lep_huffdec_simt.h takes the scans whose
    markers stand where they should (the positions travel behind the scan bytes), the single-wave kernel and the host parser the others
    -- the .lep is the host parser's whichever way a file went, == the reference's for the fixtures; a second codec object with the
    lane-per-piece kernels switched off gives the same bytes"""
    import test_core_emulation as emu_tests
    from conftest import golden
    from lepton_amd.codec import GpuCodec

    cases = emu_tests._restart_interval_jpegs()
    jpgs = [j for _, j in cases] + [ref_golden("trailingrst")[0], ref_golden("trailingrst2")[0], golden("rst_rows_gray_64x96")[0]]
    want, want_status, _ = gpu_codec.compress_batch(jpgs, verify=False, host_huffman=True)   # the host parser's answer for every file (no round-trip check:
    got, status, stats = gpu_codec.compress_batch(jpgs, verify=False)                        #  a scan with a marker out of place does not restore)
    assert status == want_status
    for g, w in zip(got, want):
        assert g == w
    names = [n for n, _ in cases]
    for n in ("rst_c420_176x112", "lay_mixed_rst_104x72", "lay_ids_pad0_64x64", "lay_440_640x480_2seg"):
        assert got[names.index(n)] == golden(n)[1], n
    assert stats["gpu_huffman_files"] >= 9, stats
    monkeypatch.setenv("LEP_HUFFDEC_SIMT", "0")
    other = GpuCodec(0)
    try:
        got2, status2, _ = other.compress_batch(jpgs, verify=False)
        assert status2 == status and got2 == got
    finally:
        other.close()
    clean = [k for k, g in enumerate(got) if g is not None and (k >= len(names) or "marker" not in names[k])]   # (a scan with a marker out of place does not restore: 41 with verification)
    back, st, _ = gpu_codec.decompress_batch([got[k] for k in clean])
    assert st == [0] * len(clean) and back == [jpgs[k] for k in clean]


@pytest.mark.gpu
def test_gpu_lane_per_unit_scan_encoder_writes_restart_intervals(gpu_codec, monkeypatch):
    """files with restart intervals through lep_decompress_batch: since round 5 the lane-per-unit kernels (lep_huff_simt.h) write their scans
    too -- a unit ends where its interval does and carries pad bits and marker, the stuffing pass leaves the markers' FFs alone.  The JPEG
    bytes are the originals, == the wavefront-per-segment kernel's (LEP_HUFFENC_SIMT=0, a second codec object) == the host re-coder's;
    intervals of one MCU, of a few, of whole rows, intervals longer than a thread segment, 1080p files of several segments"""
    import io
    import numpy as np
    from PIL import Image
    import test_core_emulation as emu_tests
    from lepton_amd.codec import GpuCodec

    cases = [(n, j) for n, j in emu_tests._restart_interval_jpegs() if "marker" not in n and n != "narrowrst"]
    rng = np.random.default_rng(78)
    base = np.asarray(Image.fromarray(rng.integers(0, 256, (135, 240, 3), dtype=np.uint8), "RGB").resize((1920, 1080), Image.BICUBIC)).astype(np.int16)
    big = Image.fromarray(np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8), "RGB")
    for tag, kw in (("big_rst_5mcu", dict(restart_marker_blocks=5)), ("big_rst_row", dict(restart_marker_rows=1)), ("big_rst_1000mcu", dict(restart_marker_blocks=1000)),
                    ("big_rst_13mcu_444", dict(restart_marker_blocks=13, subsampling="4:4:4"))):
        buf = io.BytesIO()
        big.save(buf, format="JPEG", quality=90, **{"subsampling": "4:2:0", **kw})
        cases.append((tag, buf.getvalue()))
    jpgs = [j for _, j in cases]
    leps, status, _ = gpu_codec.compress_batch(jpgs)
    assert status == [0] * len(jpgs)
    back, st, stats = gpu_codec.decompress_batch(leps)
    assert st == [0] * len(jpgs)
    for (n, j), b in zip(cases, back):
        assert b == j, n
    assert stats["gpu_huffman_files"] == len(jpgs), stats
    host, st_h, _ = gpu_codec.decompress_batch(leps, host_huffman=True)
    assert st_h == st and host == back
    monkeypatch.setenv("LEP_HUFFENC_SIMT", "0")
    other = GpuCodec(0)
    try:
        back2, st2, stats2 = other.decompress_batch(leps)
        assert st2 == st and back2 == back and stats2["gpu_huffman_files"] == len(jpgs)
    finally:
        other.close()


@pytest.mark.gpu
def test_gpu_lane_per_restart_interval_kernel_equals_the_single_wave_kernel(gpu_codec):
    """lep_gpu_huffman_decode_simt_device on device-resident scans with the restart positions behind the scan bytes, against
    lep_gpu_huffman_decode_device (which walks the scan as the reference does): same frames, same hand-off records, same pad byte"""
    import ctypes as C
    import test_core_emulation as emu_tests
    from lepton_amd import abi

    L = abi.lib()
    g = gpu_codec.handle
    names = ["rst_c420_176x112", "lay_mixed_rst_104x72", "lay_ids_pad0_64x64", "lay_440_640x480_2seg", "rst_1mcu", "rst_7mcu", "rst_row", "rst_3rows_444", "rst_422_optimized"]
    jpgs = [dict(emu_tests._restart_interval_jpegs())[n] for n in names]

    def dmalloc(n):
        p = C.c_void_p()
        assert L.lep_gpu_malloc(g, n, C.byref(p)) == 0
        return p

    results = []
    for simt in (0, 1):
        imgs = (abi.HuffDecImage * len(jpgs))()
        dev, planes_dev, handles = [], [], []
        rows_total = 0
        for k, jpg in enumerate(jpgs):
            h = C.c_void_p()
            img = abi.HuffDecImage()
            ok = C.c_int(0)
            assert L.lep_jpeg_open_gpu(jpg, len(jpg), C.byref(h), C.byref(img), C.byref(ok)) == 0 and ok.value
            handles.append(h)
            assert img.rsti > 0 and (img.flags & 2), names[k]
            p, n = C.c_void_p(), C.c_size_t(0)
            L.lep_jpeg_scan_bytes(h, C.byref(p), C.byref(n))
            rp, rn = C.POINTER(C.c_uint32)(), C.c_size_t(0)
            L.lep_jpeg_scan_restarts(h, C.byref(rp), C.byref(rn))
            room = (n.value + 64 + 15) & ~15
            table = bytes(C.cast(rp, C.POINTER(C.c_uint8 * (4 * rn.value))).contents)
            blob = C.string_at(p, n.value) + b"\0" * (room - n.value) + table + b"\0" * 64
            dscan = dmalloc(len(blob))
            assert L.lep_gpu_memcpy_h2d(g, dscan, blob, len(blob)) == 0
            dev.append(dscan)
            C.memmove(C.byref(imgs[k]), C.byref(img), C.sizeof(abi.HuffDecImage))
            imgs[k].scan = dscan.value
            for c in range(img.ncomp):
                nb = img.bch[c] * img.vs[c] * img.mcuv * 128
                q = dmalloc(nb)
                assert L.lep_gpu_memset(g, q, 0, nb) == 0
                dev.append(q)
                planes_dev.append((q, nb))
                imgs[k].blocks[c] = q.value
            imgs[k].rows_off = rows_total
            rows_total += img.mcuv + 1
        nrow_bytes = rows_total * C.sizeof(abi.HuffDecRow)
        drows = dmalloc(nrow_bytes)
        assert L.lep_gpu_memset(g, drows, 0, nrow_bytes) == 0
        fn = L.lep_gpu_huffman_decode_simt_device if simt else L.lep_gpu_huffman_decode_device
        assert fn(g, imgs, len(jpgs), drows, None) == 0, gpu_codec.last_error()
        assert L.lep_gpu_sync(g) == 0
        rows = (abi.HuffDecRow * rows_total)()
        assert L.lep_gpu_memcpy_d2h(g, rows, drows, nrow_bytes) == 0
        for k in range(len(jpgs)):
            fin = rows[imgs[k].rows_off + imgs[k].mcuv]
            assert (fin.aux >> 8) & 0x3fffff == 0, (names[k], simt, fin.aux)
        out = [bytes(rows)]
        for q, nb in planes_dev:
            buf = C.create_string_buffer(nb)
            assert L.lep_gpu_memcpy_d2h(g, buf, q, nb) == 0
            out.append(buf.raw)
        results.append(out)
        for q in dev + [drows]:
            L.lep_gpu_free(g, q)
        for h in handles:
            L.lep_jpeg_close(h)
    for i, (x, y) in enumerate(zip(results[0], results[1])):
        assert x == y, "buffer %d differs" % i
