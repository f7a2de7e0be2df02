"""Parity tests proper: the HIP path (through the C ABI) against (1) .lep files written by the real
reference (tests/golden), (2) the CPU oracle on seeded inputs, (3) size-independent properties at
BASELINE.json sizes (encode -> decode round trip, bit-exact JPEG restoration).  Bar: bit-exact."""
import ctypes as C
import hashlib

import pytest

import oracle_binding as ob
from conftest import golden, golden_cases
from lepton_amd import abi, corpus
from lepton_amd.codec import GpuCodec, JpegImage, LepFile, LeptonError

pytestmark = pytest.mark.gpu


def test_native_library_is_loaded_and_gpu_present(gpu_codec):
    assert b"gfx950" in abi.lib().lep_version()
    assert gpu_codec.handle


def test_gpu_kernel_arithmetic_selftest(gpu_codec):
    """exhaustive on-device check: float-reciprocal Branch probability == integer division for all 255 x 255 count pairs"""
    assert abi.lib().lep_gpu_selftest(gpu_codec.handle) == 0


@pytest.mark.parametrize("name", golden_cases())
def test_gpu_encode_equals_reference_lep(gpu_codec, name):
    jpg, lep = golden(name)
    assert gpu_codec.compress(jpg) == lep


@pytest.mark.parametrize("name", golden_cases())
def test_gpu_decode_restores_reference_jpeg(gpu_codec, name):
    jpg, lep = golden(name)
    assert gpu_codec.decompress(lep) == jpg


def test_gpu_batch_streams_equal_oracle(gpu_codec):
    """mixed geometry batch in ONE launch: every segment's stream equals the oracle's"""
    jpgs = [corpus.synth_jpeg(w, h, s, quality=q) for (w, h, s, q) in
            [(320, 240, 1, 90), (203, 149, 2, 75), (640, 360, 3, 95), (64, 64, 4, 50), (1280, 720, 5, 92)]]
    imgs = [JpegImage(j) for j in jpgs]
    plans = [im.plan() for im in imgs]
    got = gpu_codec.encode(imgs, plans)
    for im, p, g in zip(imgs, plans, got):
        want, _ = ob.oracle_encode(im.desc, p)
        assert g == want
    # and the decode direction through whole files
    for im, g, j in zip(imgs, got, jpgs):
        assert gpu_codec.decompress(im.write_lep(g)) == j


def test_gpu_1080p_equals_oracle_and_roundtrips(gpu_codec):
    jpg = corpus.synth_jpeg(1920, 1080, 10000)
    img = JpegImage(jpg)
    plan = img.plan()
    assert len(plan) == 8
    streams = gpu_codec.encode([img], [plan])[0]
    want, _ = ob.oracle_encode(img.desc, plan)
    assert streams == want
    assert gpu_codec.decompress(img.write_lep(streams)) == jpg


def test_gpu_4k_roundtrip_property(gpu_codec):
    """BASELINE configs[1] (single 4K 4:2:0), the exact input the headline is quoted on: the oracle's eight streams (2 s of CPU)
    from the single-kernel encoder (a launch of 8 segments) AND from the split-phase encoder (the same image 8 times: a launch of
    64), the oracle's streams decoded back to the frame, then the properties: bit-exact round trip JPEG -> .lep -> JPEG,
    decode(encode(frame)) == frame, the trailer's size field."""
    jpg = corpus.synth_jpeg(3840, 2160, 1234)
    img = JpegImage(jpg)
    plan = img.plan()
    assert len(plan) == 8 and img.desc.total_blocks() == 194400
    want, _ = ob.oracle_encode(img.desc, plan)
    assert gpu_codec.encode([img], [plan])[0] == want           # split-phase encoder, stitched writer at 64 chunks per segment
    assert b"enc5" in abi.lib().lep_gpu_last_kernel_name(gpu_codec.handle)
    got = gpu_codec.encode([img] * 8, [plan] * 8)               # ... at 32 chunks per segment
    assert all(g == want for g in got)
    for env in ({"LEP_ENC5_MIN": "0"}, {"LEP_ENC5_WCHUNKS": "0"}):   # the single-kernel encoder; the lane-per-segment writer beside gather
        import os
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            other = GpuCodec(0)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        try:
            assert other.encode([img], [plan])[0] == want
            assert (b"enc5" in abi.lib().lep_gpu_last_kernel_name(other.handle)) == ("LEP_ENC5_WCHUNKS" in env)
        finally:
            other.close()
    orig = [C.string_at(img.desc.blocks[c], img.desc.nblocks(c) * 128) for c in range(3)]
    for c in range(3):
        C.memset(img.desc.blocks[c], 0, img.desc.nblocks(c) * 128)
    assert not any(_gpu_decode_streams(gpu_codec, img.desc, plan, want))
    assert [C.string_at(img.desc.blocks[c], img.desc.nblocks(c) * 128) for c in range(3)] == orig
    lep = gpu_codec.compress(jpg)
    f = LepFile(lep)
    gpu_codec.decode([f])
    for c in range(3):
        n = img.desc.nblocks(c) * 128
        assert hashlib.md5(C.string_at(f.desc.blocks[c], n)).digest() == hashlib.md5(C.string_at(img.desc.blocks[c], n)).digest()
    assert f.recode() == jpg
    assert int.from_bytes(lep[-4:], "little") == len(lep)


def test_gpu_rejects_out_of_range_coefficient(gpu_codec):
    """COEFFICIENT_OUT_OF_RANGE (exit code 6) like encoder.cc:124,265 when |coef| needs > 11 bits"""
    img = JpegImage(corpus.synth_jpeg(64, 64, 9))
    C.cast(img.desc.blocks[0], C.POINTER(C.c_int16))[5] = 4096
    with pytest.raises(LeptonError) as e:
        gpu_codec.encode([img], [img.plan()])
    assert e.value.code == 6


def test_gpu_decode_of_garbage_stream_is_contained(gpu_codec):
    """a corrupt stream must come back as an error code or a (wrong) frame, never hang or crash"""
    jpg, lep = golden("c420_160x120")
    f = LepFile(lep)
    f.streams[0] = bytes(len(f.streams[0]))
    try:
        gpu_codec.decode([f])
    except LeptonError as e:
        assert e.code in (6, 7, 43)


def test_gpu_v3_encoder_many_bins_per_block(gpu_codec):
    """blocks with more than 512 bins go through several lane ranges of the bin list (lep_enc3.h); streams == oracle"""
    import numpy as np

    img = JpegImage(corpus.synth_jpeg(64, 48, 11, quality=100))
    d = img.desc
    rng = np.random.default_rng(5)
    for c in range(d.ncomp):
        n = d.nblocks(c) * 64
        arr = (C.c_int16 * n).from_address(d.blocks[c])
        vals = rng.integers(-255, 256, n)
        vals[rng.random(n) < 0.1] = 0
        for i in range(n):
            arr[i] = int(vals[i])
        for b in range(d.nblocks(c)):
            arr[b * 64 + 49] = 0
    plan = img.plan()
    want, _ = ob.oracle_encode(d, plan)
    assert gpu_codec.encode([img], [plan])[0] == want


@pytest.mark.parametrize("waves", ["4", "8"])
def test_gpu_decoder_register_budget_builds(waves, monkeypatch):
    """both register-budget builds of the decode kernel (8 waves per SIMD: 64 VGPRs with spills to scratch, chosen for launches
    that fill the chip; 4 waves: no spills, chosen for small launches) restore the same JPEGs"""
    monkeypatch.setenv("LEP_DEC_WAVES", waves)
    codec = GpuCodec(0)
    try:
        for name in ("c420_odd_203x149", "q30_256x256_4seg", "rst_c420_176x112"):
            jpg, lep = golden(name)
            assert codec.decompress(lep) == jpg
        jpg = corpus.synth_jpeg(1280, 720, 77, quality=95)
        assert codec.decompress(codec.compress(jpg)) == jpg
        assert abi.lib().lep_gpu_last_kernel_name(codec.handle).decode() in ("lep_decode_v4_kernel<%s>" % waves, "lep_huffman_encode_kernel")
    finally:
        codec.close()


def _gpu_decode_streams(codec, desc, segs, streams):
    """lep_gpu_decode_host on raw per-segment streams; returns the per-segment exit codes"""
    n = len(segs)
    descs = (abi.ImageDesc * 1)(desc)
    flat = (abi.Segment * n)(*[abi.Segment(0, s.luma_y_start, s.luma_y_end, s.is_last) for s in segs])
    keep = [C.create_string_buffer(bytes(w), max(1, len(w))) for w in streams]
    arr = (abi.Bytes * n)()
    for k, b in enumerate(keep):
        arr[k].data, arr[k].len, arr[k].cap = C.cast(b, C.c_void_p).value, len(streams[k]), len(streams[k])
    status = (C.c_int32 * n)()
    abi.lib().lep_gpu_decode_host(codec.handle, descs, 1, flat, n, arr, status)
    return [status[k] for k in range(n)]


@pytest.mark.gpu
@pytest.mark.parametrize("env,kernel", [({"LEP_DEC_WAVES": "4"}, "lep_decode_v4_kernel<4>"), ({"LEP_DEC_WAVES": "8"}, "lep_decode_v4_kernel<8>")])
def test_gpu_decoder_register_budget_forms(env, kernel, monkeypatch):
    """the two builds of the decode kernel a launch can take -- 4 wavefronts per SIMD (128 VGPRs, no spills: launches that cannot fill the
    chip) and 8 (64 VGPRs) -- forced here for launches of any size (files of 1, 2, 4 and 8 segments): restore the reference-written
    goldens byte for byte, return the oracle's frame from the oracle's streams, and refuse a garbage stream in one segment without
    disturbing its neighbours.  (Round 4's second decoder generation, lep_dec5.h, was measured slower and is gone: LAB_NOTES.md 4.)"""
    import numpy as np
    import oracle_binding as ob

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    codec = GpuCodec(0)
    try:
        for name in golden_cases():
            jpg, lep = golden(name)
            assert codec.decompress(lep) == jpg, name
        img = JpegImage(corpus.synth_jpeg(1920, 1080, 31, skew=2.0))
        d, segs = img.desc, img.plan()
        assert len(segs) >= 4
        want, _ = ob.oracle_encode(d, segs)
        orig = [C.string_at(d.blocks[c], d.nblocks(c) * 128) for c in range(d.ncomp)]
        for c in range(d.ncomp):
            C.memset(d.blocks[c], 0, d.nblocks(c) * 128)
        st = _gpu_decode_streams(codec, d, segs, want)
        assert kernel in abi.lib().lep_gpu_last_kernel_name(codec.handle).decode()
        assert not any(st) and [C.string_at(d.blocks[c], d.nblocks(c) * 128) for c in range(d.ncomp)] == orig
        bad = list(want)
        bad[1] = bytes(np.random.default_rng(3).integers(0, 256, 300, dtype=np.uint8))
        for c in range(d.ncomp):
            C.memset(d.blocks[c], 0, d.nblocks(c) * 128)
        st = _gpu_decode_streams(codec, d, segs, bad)
        assert all(rc == 0 for i, rc in enumerate(st) if i != 1) and st[1] in (0, 6, 7, 43)
        w = d.width_blocks[0]
        got = C.string_at(d.blocks[0], d.nblocks(0) * 128)
        for i, s in enumerate(segs):
            if i != 1:
                a, b = s.luma_y_start * w * 128, (d.height_blocks[0] if s.is_last else s.luma_y_end) * w * 128
                assert got[a:b] == orig[0][a:b], i
    finally:
        codec.close()


def test_gpu_v4_decoder_large_coefficients(gpu_codec):
    """rare paths of lep_dec4.h on the GPU: exponent bins beyond the prefetched groups, residual bits >= 4, threshold bins,
    interior runs over several windows; decode(oracle streams) == frame, and the v3 encoder writes those streams"""
    import numpy as np

    img = JpegImage(corpus.synth_jpeg(64, 48, 11, quality=100))
    d = img.desc
    rng = np.random.default_rng(5)
    for c in range(d.ncomp):
        n = d.nblocks(c) * 64
        arr = (C.c_int16 * n).from_address(d.blocks[c])
        vals = rng.integers(-255, 256, n)
        vals[rng.random(n) < 0.1] = 0
        for i in range(n):
            arr[i] = int(vals[i])
        for b in range(d.nblocks(c)):
            arr[b * 64 + 49] = 0
    orig = [C.string_at(d.blocks[c], d.nblocks(c) * 128) for c in range(d.ncomp)]
    plan = img.plan()
    streams = gpu_codec.encode([img], [plan])[0]
    want, _ = ob.oracle_encode(d, plan)
    assert streams == want
    f = LepFile(img.write_lep(streams))
    gpu_codec.decode([f])
    for c in range(d.ncomp):
        n = d.coded_blocks[c] * 128
        assert C.string_at(f.desc.blocks[c], n) == orig[c][:n]


def test_gpu_batch_pipeline_equals_per_file_and_reference(gpu_codec):
    """lep_compress_batch / lep_decompress_batch (host pool + overlapped copies + kernels) give exactly the per-file
    results: reference-written .lep bytes for the golden fixtures, original JPEGs back; small chunks force several
    pipeline iterations; a broken file and a progressive file only fail themselves"""
    names = golden_cases()
    jpgs = [golden(n)[0] for n in names]
    leps = [golden(n)[1] for n in names]
    extra = [corpus.synth_jpeg(512, 384, 41), b"not a jpeg at all", corpus.synth_jpeg(256, 256, 42, progressive=True)]
    prog_ref = gpu_codec.compress(extra[2])   # progressive files: host Huffman coders (jpeg_progressive.cc), same GPU hot path
    got, status, stats = gpu_codec.compress_batch(jpgs + extra, chunk_bytes=300000)
    assert status[: len(names)] == [0] * len(names)
    assert got[: len(names)] == leps
    assert status[len(names)] == 0 and got[len(names)] == gpu_codec.compress(extra[0])
    assert status[len(names) + 1] == 42 and got[len(names) + 1] is None      # UNSUPPORTED_JPEG, like the reference
    assert status[len(names) + 2] == 0 and got[len(names) + 2] == prog_ref
    assert stats["h2d_bytes"] > 0 and stats["d2h_bytes"] > 0
    back, status2, _ = gpu_codec.decompress_batch(leps + [got[len(names)], prog_ref, b"\xcf\x84garbage"], chunk_bytes=300000)
    assert status2[: len(names) + 2] == [0] * (len(names) + 2)
    assert back[: len(names)] == jpgs and back[len(names)] == extra[0] and back[len(names) + 1] == extra[2]
    assert status2[-1] != 0 and back[-1] is None


def test_gpu_batch_pipeline_with_overlapped_launches(gpu_codec, monkeypatch):
    """LEP_BATCH_OVERLAP=1: consecutive chunks' coder kernels on two streams / two workspace sets of the library (the next
    chunk starts in the wave slots the current one's long segments leave free) -- same bytes, with and without verification"""
    monkeypatch.setenv("LEP_BATCH_OVERLAP", "1")
    names = golden_cases()
    jpgs = [golden(n)[0] for n in names] * 3
    leps = [golden(n)[1] for n in names] * 3
    for verify in (False, True):
        got, status, _ = gpu_codec.compress_batch(jpgs, chunk_bytes=300000, verify=verify)
        assert status == [0] * len(jpgs) and got == leps


def test_gpu_overlapped_chunks_of_unequal_length_with_verification(gpu_codec, monkeypatch):
    """LEP_BATCH_OVERLAP=1 + verify with chunks of very unequal lengths: a short chunk reaches its scan encoder (the Huffman half of the
    round-trip check) on the second stream BEFORE the long chunk in front of it does on the first.  Until the end of round 6 the scan
    encoder's descriptors and scratch were one buffer for both streams: the short chunk's replaced the ones the long chunk's kernels were
    still to read -- a memory fault (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION), seen once in the closing visit and then every time under
    scripts/stress_overlap_verify.py.  They are per workspace set now (lep_gpu.hip d_huff / d_huffenc)."""
    monkeypatch.setenv("LEP_BATCH_OVERLAP", "1")
    names = golden_cases()
    big = [corpus.synth_jpeg(3840, 2160, 900 + i) for i in range(2)]
    want_big = [gpu_codec.compress(j) for j in big]
    jpgs, leps = [], []
    for _ in range(2):
        jpgs += big; leps += want_big
        jpgs += [golden(n)[0] for n in names]; leps += [golden(n)[1] for n in names]
    for chunk_bytes in (300000, 40000, 2500000):
        got, status, _ = gpu_codec.compress_batch(jpgs, chunk_bytes=chunk_bytes, verify=True)
        assert status == [0] * len(jpgs), (chunk_bytes, [s for s in status if s][:8])
        assert got == leps, chunk_bytes


def test_gpu_batch_pipeline_verifies_on_the_gpu(gpu_codec):
    """verify=1: every file is decoded again on the GPU and compared with its input frame before its .lep is released"""
    jpgs = [corpus.synth_jpeg(640, 480, 51), corpus.synth_jpeg(320, 200, 52, quality=75), golden("c420_odd_203x149")[0]]
    got, status, _ = gpu_codec.compress_batch(jpgs, verify=True)
    assert status == [0, 0, 0]
    assert got == [gpu_codec.compress(j) for j in jpgs]


def test_gpu_huffman_reencode_matches_host_and_reference(gpu_codec):
    """decode direction end to end on the GPU: arithmetic decode + JPEG Huffman re-encode (lep_huff.h); eligible fixtures
    come back as scan bytes only and must equal the reference's input JPEGs, like the host re-coder's output"""
    names = golden_cases()
    leps = [golden(n)[1] for n in names]
    jpgs = [golden(n)[0] for n in names]
    extra = corpus.synth_jpeg(1280, 720, 61, quality=85)
    leps.append(gpu_codec.compress(extra)); jpgs.append(extra)
    a, sa, stats_gpu = gpu_codec.decompress_batch(leps)
    b, sb, stats_host = gpu_codec.decompress_batch(leps, host_huffman=True)
    assert sa == [0] * len(leps) and sb == sa
    assert a == jpgs and b == jpgs
    # frames no longer cross PCIe for the eligible files (progressive / truncated / grey fixtures keep the host path)
    _, _, stats_gpu = gpu_codec.decompress_batch([leps[-1]] * 4)
    _, _, stats_host = gpu_codec.decompress_batch([leps[-1]] * 4, host_huffman=True)
    assert stats_gpu["d2h_bytes"] < stats_host["d2h_bytes"] / 3


def test_gpu_huffman_reencode_restart_markers_and_grey(gpu_codec):
    from PIL import Image
    import io
    import numpy as np

    rng = np.random.default_rng(3)
    img = Image.fromarray(rng.integers(0, 256, (120, 200), dtype=np.uint8), "L")
    buf = io.BytesIO(); img.save(buf, format="JPEG", quality=80)
    grey = buf.getvalue()
    cases = [grey, golden("rst_c420_176x112")[0]]
    leps = [gpu_codec.compress(j) for j in cases]
    out, st, _ = gpu_codec.decompress_batch(leps)
    assert st == [0, 0] and out == cases


def test_gpu_huffman_decode_matches_host_and_reference(gpu_codec):
    """encode direction end to end on the GPU: JPEG Huffman scan decode (lep_huffdec.h) + arithmetic encode; the .lep files
    equal the reference's (golden fixtures) and those of the host-parser path, and far fewer bytes cross PCIe"""
    import io
    import numpy as np
    from PIL import Image

    names = golden_cases()
    jpgs = [golden(n)[0] for n in names]
    leps = [golden(n)[1] for n in names]
    rng = np.random.default_rng(9)
    im = Image.fromarray(rng.integers(0, 256, (96, 160, 3), dtype=np.uint8), "RGB")
    buf = io.BytesIO(); im.save(buf, format="JPEG", quality=70, subsampling="4:2:0", restart_marker_blocks=3)
    jpgs += [corpus.synth_jpeg(1280, 720, 62, quality=85), buf.getvalue(), corpus.synth_jpeg(333, 211, 63, subsampling="4:4:4")]
    a, sa, stats_gpu = gpu_codec.compress_batch(jpgs)
    b, sb, stats_host = gpu_codec.compress_batch(jpgs, host_huffman=True)
    assert sa == [0] * len(jpgs) and sb == sa
    assert a == b and a[: len(names)] == leps
    _, _, stats_gpu = gpu_codec.compress_batch([jpgs[len(names)]] * 4)
    _, _, stats_host = gpu_codec.compress_batch([jpgs[len(names)]] * 4, host_huffman=True)
    assert stats_gpu["h2d_bytes"] < stats_host["h2d_bytes"] / 2
    back, st, _ = gpu_codec.decompress_batch(a)
    assert st == [0] * len(jpgs) and back == jpgs


def test_gpu_verify_is_blind_to_stale_staging(gpu_codec):
    """Round-1 driver failure: `verify` compared a frame's 256-byte-rounded ROOM in the slot, and host-parsed frames were
    uploaded with that room -- so whatever an earlier batch had left in the pinned staging behind a frame with an odd
    block count (prog_gray_120x88: 165 blocks) surfaced as a false ROUNDTRIP_FAILURE in a re-used slot.  The staging is
    poisoned on purpose here; files that take the host parser (progressive, truncated, grey) with odd block counts must
    verify, batch after batch, in both slots."""
    from lepton_amd import abi

    L = abi.lib()
    names = ["prog_gray_120x88", "prog_c444_203x149", "prog_truncated_mid", "gray_120x88", "one_block_8x8", "c420_odd_203x149", "prog_truncated_dc"]
    jpgs = [golden(n)[0] for n in names]
    leps = [golden(n)[1] for n in names]
    filler = [corpus.synth_jpeg(640, 480, 60 + i) for i in range(6)]
    # grow the pinned staging (host Huffman path stages whole frames) and leave it dirty
    _, st, _ = gpu_codec.compress_batch(filler * 4, host_huffman=True, chunk_images=8, verify=True)
    assert st == [0] * len(st)
    for it in range(6):
        L.lep_batch_debug_poison(0xA5 if it % 2 == 0 else 0xFF)
        batch = (jpgs * 3)[it % 3:] + filler[:2]
        got, st, _ = gpu_codec.compress_batch(batch, verify=True, chunk_images=5)   # several chunks: both slots are re-used
        assert st == [0] * len(batch), (it, st)
        want = (leps * 3)[it % 3:]
        assert got[: len(want)] == want
    L.lep_batch_debug_poison(0x5A)
    back, st2, _ = gpu_codec.decompress_batch(leps * 2, chunk_images=3)
    assert st2 == [0] * (2 * len(leps)) and back == jpgs * 2


def test_gpu_batch_redoes_files_whose_streams_outgrow_their_reservation(gpu_codec):
    """the batch pipeline reserves stream space from a segment's JPEG bytes; a file that needs more (dense noise at q100,
    where the arithmetic coder gains nothing) must come out exactly like the per-file path, not as BUFFER_TOO_SMALL"""
    import io
    import numpy as np
    from PIL import Image

    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(a, "RGB").save(buf, format="JPEG", quality=100, subsampling=0)
    noisy = buf.getvalue()
    jpgs = [noisy, golden("c420_160x120")[0], noisy]
    got, st, stats = gpu_codec.compress_batch(jpgs, verify=True)
    assert st == [0, 0, 0]
    assert got[0] == gpu_codec.compress(noisy) and got[2] == got[0] and got[1] == golden("c420_160x120")[1]
    assert gpu_codec.decompress(got[0]) == noisy


def test_gpu_hostile_handoff_sizes_do_not_size_the_arena(gpu_codec):
    """a .lep whose hand-offs claim 4 GB segments (or 200 of them) is one request among many: it must fail or fall back
    alone, without a multi-GB reservation and without taking the batch down"""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))
    import mutate as mu

    jpg, lep = golden("q30_256x256_4seg")
    hostile = [mu.with_handoffs(lep, segment_size=0xffffffff), mu.with_handoffs(lep, count=200), mu.with_handoffs(lep, count=17)]
    good = [golden(n)[1] for n in ("c420_160x120", "c444_96x80")]
    back, st, stats = gpu_codec.decompress_batch([good[0]] + hostile + [good[1]])
    assert st[0] == 0 and st[-1] == 0 and back[0] == golden("c420_160x120")[0] and back[-1] == golden("c444_96x80")[0]
    assert st[2] == 1 and st[3] == 1              # more hand-offs than stream ids: the reference dies in an assertion
    assert st[1] != 0 or back[1] == jpg            # absurd sizes (the reference segfaults on them): refused, or restored right


def test_gpu_single_wave_huffman_decode_in_the_compress_pipeline(gpu_codec, monkeypatch):
    """LEP_HUFFDEC_SIMT=0: one wavefront per image decodes the JPEG scan (lep_huffdec.h, the fallback of the lane-per-subsequence
    kernels and the form that takes restart intervals) -- same .lep bytes as the reference's"""
    names = golden_cases()
    jpgs = [golden(n)[0] for n in names] + [corpus.synth_jpeg(1280, 720, 61), corpus.synth_jpeg(640, 480, 62, quality=97)]
    monkeypatch.setenv("LEP_HUFFDEC_SIMT", "0")
    got, st, _ = gpu_codec.compress_batch(jpgs, chunk_images=12)
    assert st == [0] * len(jpgs)
    assert got[: len(names)] == [golden(n)[1] for n in names]
    assert got[len(names):] == [gpu_codec.compress(j) for j in jpgs[len(names):]]


@pytest.mark.parametrize("bits", ["0", "1024", "4096", "65536"])
def test_gpu_lane_per_subsequence_huffman_decode_in_the_compress_pipeline(gpu_codec, monkeypatch, bits):
    """one lane per subsequence decodes the JPEG scan (lep_huffdec_simt.h, the default of the batch compressor; LEP_HUFFDEC_SIMT_BITS
    forces the subsequence length: 1024 bits leave the settle passes work to do and some scans to the fallback, 65536 make most
    fixtures a single lane) -- same .lep bytes as the reference's, whichever path a file ends up on"""
    names = golden_cases()
    import test_core_emulation as emu_tests
    jpgs = [golden(n)[0] for n in names] + [corpus.synth_jpeg(1280, 720, 61), corpus.synth_jpeg(640, 480, 62, quality=97), corpus.synth_jpeg(1920, 1080, 63),
                                             emu_tests._jpeg_for_huffman_tests("optimized_noise_444"), emu_tests._jpeg_for_huffman_tests("optimized_q95")]
    monkeypatch.setenv("LEP_HUFFDEC_SIMT", "1")
    if bits != "0":
        monkeypatch.setenv("LEP_HUFFDEC_SIMT_BITS", bits)
    codec = GpuCodec(0)        # the knob is read when the codec object is made
    try:
        got, st, stats = codec.compress_batch(jpgs, chunk_images=12)
        assert st == [0] * len(jpgs)
        assert got[: len(names)] == [golden(n)[1] for n in names]
        assert got[len(names):] == [gpu_codec.compress(j) for j in jpgs[len(names):]]
        assert stats["gpu_huffman_files"] > len(jpgs) // 2
    finally:
        codec.close()


@pytest.mark.parametrize("simt", ["0", "1"])
def test_gpu_scan_encoders_in_the_decompress_pipeline(monkeypatch, simt):
    """the decompressor's JPEG scan bytes from the lane-per-unit kernels (lep_huff_simt.h, the default) and from the
    wavefront-per-segment kernel (LEP_HUFFENC_SIMT=0): the original files, byte for byte, from both -- fixtures with restart
    intervals, several thread segments, odd sizes, and three synthetic sizes"""
    names = golden_cases()
    jpgs = [golden(n)[0] for n in names] + [corpus.synth_jpeg(1280, 720, 64), corpus.synth_jpeg(1920, 1080, 65, quality=95), corpus.synth_jpeg(203, 149, 66)]
    leps = [golden(n)[1] for n in names]
    monkeypatch.setenv("LEP_HUFFENC_SIMT", simt)
    codec = GpuCodec(0)
    try:
        leps += [codec.compress(j) for j in jpgs[len(names):]]
        back, st, stats = codec.decompress_batch(leps, chunk_images=16)
        ok = [i for i in range(len(leps)) if st[i] == 0]
        assert len(ok) >= len(leps) - 8                 # (a few fixtures are files the reference refuses to restore)
        for i in ok:
            assert back[i] == jpgs[i], i
        assert stats["gpu_huffman_files"] > len(leps) // 2
        name = abi.lib().lep_gpu_last_kernel_name(codec.handle).decode()
        assert ("simt" in name) == (simt == "1") or "progressive" in name or "decode" in name, name
    finally:
        codec.close()


def test_gpu_lane_per_subsequence_huffman_decode_equals_the_single_wave_kernel(gpu_codec):
    """the two scan decoders called directly on device-resident scans of 24 different 4:2:0 images: same frames, same hand-off records"""
    import test_core_emulation as emu_tests

    L = abi.lib()
    g = gpu_codec.handle
    jpgs = [corpus.synth_jpeg(640 + 64 * (i % 5), 360 + 40 * (i % 3), 300 + i, quality=70 + i) for i in range(24)]

    def dmalloc(n):
        p = C.c_void_p()
        assert L.lep_gpu_malloc(g, n, C.byref(p)) == 0
        return p

    results = []
    for simt in (0, 1):
        frames = [emu_tests._huffdec_setup(jpg) for jpg in jpgs]
        imgs = (abi.HuffDecImage * len(jpgs))()
        dev, planes_dev = [], []
        rows_total = 0
        for k, (img, scan, planes, d) in enumerate(frames):
            n = len(scan.raw)
            dscan = dmalloc(n)
            assert L.lep_gpu_memcpy_h2d(g, dscan, scan, n) == 0
            dev.append(dscan)
            C.memmove(C.byref(imgs[k]), C.byref(img), C.sizeof(abi.HuffDecImage))
            imgs[k].scan = dscan.value
            for c in range(d.ncomp):
                nb = len(planes[c].raw)
                p = dmalloc(nb)
                assert L.lep_gpu_memset(g, p, 0, nb) == 0
                dev.append(p)
                planes_dev.append((p, nb))
                imgs[k].blocks[c] = p.value
            imgs[k].rows_off = rows_total
            rows_total += img.mcuv + 1
        nrow_bytes = rows_total * C.sizeof(abi.HuffDecRow)
        drows = dmalloc(nrow_bytes)
        assert L.lep_gpu_memset(g, drows, 0, nrow_bytes) == 0
        fn = L.lep_gpu_huffman_decode_simt_device if simt else L.lep_gpu_huffman_decode_device
        assert fn(g, imgs, len(jpgs), drows, None) == 0
        assert L.lep_gpu_sync(g) == 0
        rows = C.create_string_buffer(nrow_bytes)
        assert L.lep_gpu_memcpy_d2h(g, rows, drows, nrow_bytes) == 0
        out = [rows.raw]
        for p, nb in planes_dev:
            buf = C.create_string_buffer(nb)
            assert L.lep_gpu_memcpy_d2h(g, buf, p, nb) == 0
            out.append(buf.raw)
        results.append(out)
        for p in dev + [drows]:
            L.lep_gpu_free(g, p)
    assert len(results[0]) == len(results[1])
    for i, (x, y) in enumerate(zip(results[0], results[1])):
        assert x == y, "buffer %d differs" % i


@pytest.mark.parametrize("waves", ["2", "4", "8"])
def test_gpu_encoder_builds_agree(waves, monkeypatch):
    """the encoder's three launch forms -- two wavefronts per segment (producer / bool coder, small launches), one wavefront
    with 128 VGPRs, one with 64 -- write the same streams: golden .lep bytes, oracle streams for blocks with more bins
    than one chunk holds, and the same exit code for a coefficient the format cannot hold"""
    import numpy as np

    monkeypatch.setenv("LEP_ENC_WAVES", waves)
    codec = GpuCodec(0)
    try:
        for name in ("c420_odd_203x149", "q30_256x256_4seg", "gray_120x88", "prog_c420_320x240", "truncated", "one_block_8x8"):
            jpg, lep = golden(name)
            assert codec.compress(jpg) == lep, name
        assert abi.lib().lep_gpu_last_kernel_name(codec.handle).decode() == {"2": "lep_encode_v3x2_kernel", "4": "lep_encode_v3_kernel<4>", "8": "lep_encode_v3_kernel<8>"}[waves]
        img = JpegImage(corpus.synth_jpeg(64, 48, 11, quality=100))
        d = img.desc
        rng = np.random.default_rng(5)
        for c in range(d.ncomp):
            n = d.nblocks(c) * 64
            arr = (C.c_int16 * n).from_address(d.blocks[c])
            vals = rng.integers(-255, 256, n)
            vals[rng.random(n) < 0.1] = 0
            for i in range(n):
                arr[i] = int(vals[i])
            for b in range(d.nblocks(c)):
                arr[b * 64 + 49] = 0
        plan = img.plan()
        want, _ = ob.oracle_encode(d, plan)
        assert codec.encode([img], [plan])[0] == want
        arr = (C.c_int16 * 64).from_address(d.blocks[0])
        arr[3] = 3000                                   # 12 bits: COEFFICIENT_OUT_OF_RANGE, and the launch must come back
        with pytest.raises(LeptonError) as e:
            codec.encode([img], [plan])
        assert e.value.code == 6
        jpg = corpus.synth_jpeg(1280, 720, 78, quality=92)
        assert codec.decompress(codec.compress(jpg)) == jpg
    finally:
        codec.close()


def test_gpu_progressive_scans_are_recoded_on_the_gpu(gpu_codec):
    """decode direction of progressive files (BASELINE.json configs[4]): arithmetic decode + lep_huffprog.h (one wavefront per
    scan: DC / AC first-stage and refinement scans, end-of-band runs, held-back correction bits); the restored files equal the
    reference's inputs, and only scan bytes cross PCIe (the frame of an eligible file stays on the device)"""
    names = [n for n in golden_cases() if n.startswith("prog_")]
    leps = [golden(n)[1] for n in names]
    jpgs = [golden(n)[0] for n in names]
    big = [corpus.synth_jpeg(1920, 1080, 91, progressive=True), corpus.synth_jpeg(640, 480, 92, progressive=True, subsampling="4:4:4", quality=97),
           corpus.synth_jpeg(800, 600, 93, progressive=True, quality=35)]
    big_lep = [gpu_codec.compress(j) for j in big]
    back, st, stats = gpu_codec.decompress_batch(leps + big_lep + [golden("c420_160x120")[1]])
    assert st == [0] * (len(names) + 4)
    assert back[: len(names)] == jpgs and back[len(names): len(names) + 3] == big and back[-1] == golden("c420_160x120")[0]
    # the three big files alone: frames 1080p = 6.2 MB + ... would cross PCIe on the host path; on the GPU path only their scans do
    _, st2, stats2 = gpu_codec.decompress_batch(big_lep)
    assert st2 == [0, 0, 0] and stats2["d2h_bytes"] < 1.2 * sum(map(len, big)) + 65536
    # and the host path still agrees
    back3, st3, _ = gpu_codec.decompress_batch(big_lep, host_huffman=True)
    assert st3 == [0, 0, 0] and back3 == big


@pytest.mark.parametrize("simt", ["0", "1"])
def test_gpu_progressive_scan_writers_in_the_decompress_pipeline(monkeypatch, simt):
    """the decompressor's progressive scan bytes from the lane-per-unit kernels (lep_huffprog_simt.h, the default: count / place /
    assign / code / stuff) and from the wavefront-per-scan kernel (LEP_HUFFPROG_SIMT=0): the original files, byte for byte, from
    both -- the progressive fixtures (one with restart intervals, which stays with the wavefront kernel), the reference's own
    progressive images, and synthetic files up to 4K in three sampling layouts"""
    from conftest import ref_cases, ref_golden

    names = [n for n in golden_cases() if n.startswith("prog_") and "truncated" not in n]
    jpgs = [golden(n)[0] for n in names]
    leps = [golden(n)[1] for n in names]
    for n in ref_cases(progressive=True):
        j, l = ref_golden(n)
        jpgs.append(j); leps.append(l)
    big = [corpus.synth_jpeg(3840, 2160, 191, progressive=True), corpus.synth_jpeg(640, 480, 192, progressive=True, subsampling="4:4:4", quality=97),
           corpus.synth_jpeg(800, 600, 193, progressive=True, quality=35), corpus.synth_jpeg(333, 241, 194, progressive=True, subsampling="4:2:2"),
           corpus.synth_jpeg(2048, 2048, 195, progressive=True, quality=10)]
    monkeypatch.setenv("LEP_HUFFPROG_SIMT", simt)
    codec = GpuCodec(0)        # the knob is read when the codec object is made
    try:
        leps += [codec.compress(j) for j in big]
        jpgs += big
        back, st, stats = codec.decompress_batch(leps, chunk_images=64)
        assert st == [0] * len(leps), st
        for i in range(len(leps)):
            assert back[i] == jpgs[i], i
        assert stats["gpu_huffman_files"] >= len(leps) - 3, stats   # (two of the reference's images are not whole progressive frames: the host re-coder's)
        name = abi.lib().lep_gpu_last_kernel_name(codec.handle).decode()
        assert ("huffprog_simt" in name) == (simt == "1"), name
    finally:
        codec.close()


def test_gpu_progressive_scans_are_decoded_on_the_gpu(gpu_codec):
    """encode direction of progressive files: lep_huffprogdec.h (one wavefront per scan, dependency levels) + the arithmetic
    coder; the .lep files equal the reference's byte for byte, truncated progressive files still take the host parser, and
    only scan bytes cross PCIe for the eligible ones -- with and without the round-trip check."""
    names = [n for n in golden_cases() if n.startswith("prog_")]
    jpgs = [golden(n)[0] for n in names]
    leps = [golden(n)[1] for n in names]
    big = [corpus.synth_jpeg(1920, 1080, 94, progressive=True), corpus.synth_jpeg(640, 480, 95, progressive=True, subsampling="4:4:4", quality=97),
           corpus.synth_jpeg(800, 600, 96, progressive=True, quality=35), corpus.synth_jpeg(333, 241, 97, progressive=True, subsampling="4:2:2")]
    want_big = [gpu_codec.compress(j) for j in big]          # per-file path: host parser
    got, st, stats = gpu_codec.compress_batch(jpgs + big + [golden("c420_160x120")[0]], chunk_images=7)
    assert st == [0] * (len(names) + 5)
    assert got[: len(names)] == leps and got[len(names): len(names) + 4] == want_big and got[-1] == golden("c420_160x120")[1]
    _, st2, stats2 = gpu_codec.compress_batch(big)
    assert st2 == [0] * 4 and stats2["h2d_bytes"] < 1.3 * sum(map(len, big)) + 65536     # frames (6.2 MB for the 1080p one alone) never crossed
    # with verify the scans still go to the GPU decoder, and the Huffman half of the round-trip check is made there as well:
    # every scan written again from the device frame (lep_huffprog.h) and compared with the file's own bytes
    got3, st3, stats3 = gpu_codec.compress_batch(big, verify=True)
    assert st3 == [0] * 4 and got3 == want_big
    assert stats3["gpu_huffman_files"] == 4 and stats3["redone_files"] == 0
    assert stats3["h2d_bytes"] < 2.3 * sum(map(len, big)) + 65536     # un-stuffed scans + the file's own scan bytes; still no frame
    got3b, st3b, stats3b = gpu_codec.compress_batch(jpgs + big, verify=True, chunk_images=5)
    assert st3b == [0] * (len(names) + 4) and got3b == leps + want_big
    # a progressive file damaged inside a scan: whatever the GPU decoder makes of it, the answer is the host parser's
    bad = bytearray(big[2]); bad[len(bad) // 2] ^= 0x10
    try:
        want_bad, code = gpu_codec.compress(bytes(bad)), 0
    except LeptonError as e:
        want_bad, code = None, e.code
    got4, st4, _ = gpu_codec.compress_batch([bytes(bad), big[0]])
    assert st4[1] == 0 and got4[1] == want_big[0]
    assert (st4[0], got4[0]) == (code, want_bad) or (code == 41 and st4[0] == 0)   # per-file compress also runs the round-trip check


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c420_odd_203x149", "truncated", "q30_256x256_4seg"])
def test_gpu_decoder_follows_the_reference_on_impossible_edge_counts(gpu_codec, name):
    """VERDICT round 2, weak #1 on the MI355X: streams that claim more edge non-zeros than positions remain (written by the
    oracle's biased encoder, oracle/lepton_oracle.c lor_test_edge_count_bias) decode, like in the reference (decoder.cc:58-141
    indexes its tables with the claimed count), to the frame the oracle restores from them -- the original one"""
    L = ob.oracle()
    knob = C.c_int.in_dll(L, "lor_test_edge_count_bias")
    jpg, _ = golden(name)
    img = JpegImage(jpg)
    d = img.desc
    plan = img.plan()
    orig = [C.string_at(d.blocks[c], d.nblocks(c) * 128) for c in range(d.ncomp)]
    knob.value = 2
    try:
        streams, _ = ob.oracle_encode(d, plan)
    finally:
        knob.value = 0
    assert streams != ob.oracle_encode(d, plan)[0]
    ob.oracle_decode(d, plan, streams)
    assert [C.string_at(d.blocks[c], d.nblocks(c) * 128) for c in range(d.ncomp)] == orig
    f = LepFile(img.write_lep(streams))
    gpu_codec.decode([f])
    for c in range(d.ncomp):
        n = d.coded_blocks[c] * 128
        assert C.string_at(f.desc.blocks[c], n) == orig[c][:n]


@pytest.fixture(scope="module", params=[(2, 64), (1, 64), (2, 0), (2, 4)], ids=["two_wavefronts_per_segment", "one_wavefront_per_segment", "lane_per_segment_writer", "four_chunks_per_segment"])
def gpu_codec_v5(request):
    """a codec object that takes the split-phase encoder (lep_enc5.h) for every launch, however small; its walks with two
    wavefronts per segment (the default) and with one; its writer stitched from up to 64 chunks per segment (the default for
    small launches), from 4, and as one lane per segment"""
    import os
    old = {k: os.environ.get(k) for k in ("LEP_ENC5_MIN", "LEP_ENC5_WAVES", "LEP_ENC5_WCHUNKS")}
    os.environ["LEP_ENC5_MIN"] = "1"
    os.environ["LEP_ENC5_WAVES"] = str(request.param[0])
    os.environ["LEP_ENC5_WCHUNKS"] = str(request.param[1])
    try:
        return GpuCodec(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


@pytest.mark.gpu
def test_gpu_split_phase_encoder_equals_oracle(gpu_codec_v5):
    """lep_enc5.h on the MI355X: count / emit / fold / gather / write kernels; every golden fixture's streams == the oracle's,
    one launch per image and all images in ONE launch (mixed geometries, 60+ segments, 64 segments per fold / write wavefront)"""
    names = golden_cases()
    imgs = [JpegImage(golden(n)[0]) for n in names]
    plans = [im.plan() for im in imgs]
    wants = [ob.oracle_encode(im.desc, p)[0] for im, p in zip(imgs, plans)]
    for n, im, p, w in zip(names, imgs, plans, wants):
        assert gpu_codec_v5.encode([im], [p])[0] == w, n
        assert b"enc5" in gpu_codec_v5._L.lep_gpu_last_kernel_name(gpu_codec_v5.handle)
    got = gpu_codec_v5.encode(imgs * 3, plans * 3)
    assert got == wants * 3


@pytest.mark.gpu
def test_gpu_split_phase_encoder_large_coefficients_and_refusals(gpu_codec_v5):
    """entries of several units, threshold units, saturating Branches; and the serial coder's refusals in its order"""
    import numpy as np

    img = JpegImage(corpus.synth_jpeg(64, 48, 11, quality=100))
    d = img.desc
    rng = np.random.default_rng(5)
    for c in range(d.ncomp):
        n = d.nblocks(c) * 64
        arr = (C.c_int16 * n).from_address(d.blocks[c])
        vals = rng.integers(-255, 256, n)
        vals[rng.random(n) < 0.1] = 0
        big = rng.random(n) < 0.004
        vals[big] = rng.choice([-2047, 2047, 1024, -1500], int(big.sum()))
        for i in range(n):
            arr[i] = int(vals[i])
        for b in range(d.nblocks(c)):
            arr[b * 64 + 49] = 0
    plan = img.plan()
    want, _ = ob.oracle_encode(d, plan)
    assert gpu_codec_v5.encode([img], [plan])[0] == want
    luma = C.cast(d.blocks[0], C.POINTER(C.c_int16))
    luma[64 * 3 + 5] = 4096
    with pytest.raises(LeptonError) as e:
        gpu_codec_v5.encode([img], [plan])
    assert e.value.code == 6


@pytest.mark.gpu
def test_gpu_split_phase_encoder_4k(gpu_codec_v5):
    """a 4K image (8 segments of ~24,000 blocks, 380 tiles each) and a photograph-like one"""
    for kw in ({}, {"skew": 2.0}):
        img = JpegImage(corpus.synth_jpeg(3840, 2160, 4321, **kw))
        plan = img.plan()
        want, _ = ob.oracle_encode(img.desc, plan)
        assert gpu_codec_v5.encode([img], [plan])[0] == want


@pytest.mark.gpu
def test_gpu_verify_executes_the_huffman_half_for_baseline_files(gpu_codec):
    """with `verify`, a baseline file the GPU decoded has the Huffman half of the round-trip check
    (validation.cc:97-218) EXECUTED on the GPU -- every thread segment's scan bytes written again from the device frame by the
    decompressor's kernel and compared with the file's own -- not argued from the decoder's acceptance.  Same .lep bytes as
    without the check; the counter says how many segments / scans went through it."""
    names = [n for n in golden_cases() if not n.startswith(("prog", "truncated", "slice", "embedded", "permissive"))]
    jpgs = [golden(n)[0] for n in names] + [corpus.synth_jpeg(1024, 768, 77), corpus.synth_jpeg(640, 480, 78, skew=2.0)]
    plain, st0, stats0 = gpu_codec.compress_batch(jpgs)
    checked, st1, stats1 = gpu_codec.compress_batch(jpgs, verify=True)
    assert st0 == st1 == [0] * len(jpgs) and plain == checked
    assert stats0["gpu_verified_scans"] == 0
    assert stats1["gpu_verified_scans"] >= stats1["gpu_huffman_files"] > len(jpgs) // 2   # at least one segment per GPU-decoded file
    back, st2, _ = gpu_codec.decompress_batch(checked)
    assert st2 == [0] * len(jpgs) and back == jpgs


@pytest.mark.gpu
def test_gpu_split_phase_encoder_without_room_for_its_scratch(monkeypatch):
    """lep_gpu.hip launch_enc5: when the scratch of the split-phase encoder cannot be had (LEP_ENC5_SCRATCH_MAX stands in for a
    failed hipMalloc), the launch goes to the single-kernel encoder, which needs none -- same streams, no error"""
    monkeypatch.setenv("LEP_ENC5_MIN", "1")
    monkeypatch.setenv("LEP_ENC5_SCRATCH_MAX", "1")
    codec = GpuCodec(0)
    names = golden_cases()[:6]
    imgs = [JpegImage(golden(n)[0]) for n in names]
    plans = [im.plan() for im in imgs]
    wants = [ob.oracle_encode(im.desc, p)[0] for im, p in zip(imgs, plans)]
    assert codec.encode(imgs, plans) == wants
    assert b"lep_encode_v3" in codec._L.lep_gpu_last_kernel_name(codec.handle)


@pytest.mark.gpu
@pytest.mark.parametrize("win", ["1", "0"], ids=["window_of_codes", "uniform_vector_code"])
@pytest.mark.parametrize("pipelined", ["1", "0"], ids=["one_pipelined_launch", "level_by_level"])
def test_gpu_progressive_scans_pipelined_or_level_by_level(monkeypatch, pipelined, win):
    """the progressive scan decoders both ways on the MI355X: all dependency levels of the progressive files of a batch as ONE launch in which
    a scan follows the scans of its file MCU row by MCU row (small launches), or a launch per level (LEP_HUFFPROG_PIPELINE=0 /
    large launches): the same .lep bytes as the reference, damaged files included (a scan that gives up still tells the scans
    waiting for it that it is done).  win: lep_huffprogdec_win.h (the default: every lane decodes the code at its bit, the chain
    hops between them) or lep_huffprogdec.h alone (LEP_HUFFPROGDEC_WIN=0)"""
    monkeypatch.setenv("LEP_HUFFPROG_PIPELINE", pipelined)
    monkeypatch.setenv("LEP_HUFFPROGDEC_WIN", win)
    from conftest import ref_cases, ref_golden

    codec = GpuCodec(0)
    names = [n for n in golden_cases() if n.startswith("prog_")]
    jpgs = [golden(n)[0] for n in names]
    leps = [golden(n)[1] for n in names]
    for n in ref_cases(progressive=True):        # the reference's own progressive images and what its binary writes for them
        j, l = ref_golden(n)
        jpgs.append(j); leps.append(l); names.append(n)
    big = [corpus.synth_jpeg(1920, 1080, 94, progressive=True), corpus.synth_jpeg(640, 480, 95, progressive=True, subsampling="4:4:4", quality=97),
           corpus.synth_jpeg(333, 241, 97, progressive=True, subsampling="4:2:2")]
    want_big = [codec.compress(j) for j in big]
    for _ in range(3):   # (a race between a scan and the one it follows would not show every time)
        got, st, _ = codec.compress_batch(jpgs + big * 4)
        assert st == [0] * (len(names) + 12) and got == leps + want_big * 4
    bad = []
    for k in (3, 5, 7):   # damage inside different scans: early ones are followed by others
        b = bytearray(big[0]); b[len(b) * k // 9] ^= 0x24
        bad.append(bytes(b))
    want = []
    for b in bad:
        try:
            want.append((0, codec.compress(b)))
        except LeptonError as e:
            want.append((e.code, None))
    got2, st2, _ = codec.compress_batch(bad + big)
    assert st2[3:] == [0, 0, 0] and got2[3:] == want_big
    for (code, w), s, g_ in zip(want, st2[:3], got2[:3]):
        assert (s, g_) == (code, w) or (code == 41 and s == 0)   # per-file compress also runs the round-trip check
    codec.close()


def test_gpu_progressive_pipelined_launch_beyond_what_is_resident(gpu_codec):
    """the pipelined progressive scan decoder spin-waits on scans of the same launch, so its scans must be taken up in
    an order in which a scan's predecessors are running or done.  Workgroups now draw a ticket when they start and take that scan.
    Here: 640 progressive files = 6400+ scans in one launch -- more than the 4096 wavefronts of this kernel the chip holds at once --
    every file must go through the GPU scan decoder (none gives up and falls back to the host parser) and equal the reference's
    .lep bytes."""
    jpg, lep = golden("prog_c420_320x240")
    n = 640
    got, st, stats = gpu_codec.compress_batch([jpg] * n)
    assert st == [0] * n and all(g == lep for g in got)
    assert stats["gpu_huffman_files"] == n
    from lepton_amd import abi
    assert abi.lib().lep_jpeg_gpu_scan_wait_timeouts() == 0
