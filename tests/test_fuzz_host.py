"""Memory safety of the host-side parsers on hostile input (they run in-process in the serving daemon, where the reference
forks a seccomp-jailed child per request): tests/fuzz/host_fuzz.cc built from the library's host sources with
-fsanitize=address,undefined, fed seeded mutations of the golden JPEG / .lep files plus a few hand-made killers."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, golden

FUZZ = os.path.join(ROOT, "tests", "fuzz")
CSRC = os.path.join(ROOT, "lepton_amd", "csrc")
HOST_SOURCES = ["lep_api.cc", "jpeg_scan.cc", "jpeg_progressive.cc", "lep_container.cc", "jpeg_recode.cc"]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fuzz") / "host_fuzz")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-o", exe, os.path.join(FUZZ, "host_fuzz.cc")] + [os.path.join(CSRC, s) for s in HOST_SOURCES] + ["-lz", "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer runtime for g++ here: " + r.stderr[-200:])
    return exe


def _run(exe, files, cwd):
    for i in range(0, len(files), 200):
        r = subprocess.run([exe] + files[i:i + 200], cwd=cwd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, "sanitizer finding:\n" + r.stderr[-3000:]


def test_host_parsers_survive_mutated_files(harness, tmp_path):
    subprocess.check_call([sys.executable, os.path.join(FUZZ, "mutate.py"), str(tmp_path), "1500", "20260923"])
    _run(harness, sorted(os.listdir(tmp_path)), str(tmp_path))


def test_host_parsers_survive_hand_made_killers(harness, tmp_path):
    j, l = golden("c420_160x120")
    sof = j.find(b"\xff\xc0")
    dht = j.find(b"\xff\xc4")
    cases = {}
    big = bytearray(j); big[sof + 5:sof + 9] = (65500).to_bytes(2, "big") * 2
    cases["huge_dims.jpg"] = bytes(big)                         # 12 GB frame announced by a 5 KB file
    cat = bytearray(j); cat[dht + 21:dht + 33] = bytes([255]) * 12
    cases["dc_category_255.jpg"] = bytes(cat)                   # DC symbols used as bit counts
    cases["soi_only.jpg"] = b"\xff\xd8"
    cases["soi_sof_cut.jpg"] = j[:sof + 7]
    cases["no_scan.jpg"] = j[:j.find(b"\xff\xda")]
    cases["empty.lep"] = b""
    cases["magic_only.lep"] = l[:4]
    cases["header_cut.lep"] = l[:40]
    cases["huge_sizes.lep"] = l[:20] + b"\xff\xff\xff\x7f" * 4 + l[36:]
    # more hand-off records than stream ids (a stack overflow in the callers' 16-entry arrays before parse_lep capped it),
    # the section several times over, segment sizes of 4 GB, and a zlib bomb for a header
    sys.path.insert(0, FUZZ)
    import mutate as mu
    l4 = golden("q30_256x256_4seg")[1]
    for n in (17, 200, 255):
        cases["handoffs_%d.lep" % n] = mu.with_handoffs(l4, count=n)
    cases["handoffs_twice.lep"] = mu.with_handoffs(l4, repeat=2)
    cases["handoffs_20x255.lep"] = mu.with_handoffs(l4, count=255, repeat=20)
    cases["handoffs_4gb.lep"] = mu.with_handoffs(l4, segment_size=0xffffffff)
    fixed, payload, rest = mu.lep_split(l4)
    cases["header_bomb.lep"] = mu.lep_join(fixed, payload + bytes(64 << 20), rest)
    for name, data in cases.items():
        (tmp_path / name).write_bytes(data)
    _run(harness, sorted(cases), str(tmp_path))


def test_frame_budget_is_the_references():
    # 4,423,680 blocks = (576 MiB - 36 MiB) / 128 B: UncompressedComponents::max_number_of_blocks of the default build
    from lepton_amd.codec import JpegImage, LeptonError

    j, _ = golden("c420_160x120")
    sof = j.find(b"\xff\xc0")
    big = bytearray(j); big[sof + 5:sof + 9] = (65500).to_bytes(2, "big") * 2
    with pytest.raises(LeptonError) as e:
        JpegImage(bytes(big))
    assert e.value.code == 38   # TOO_MUCH_MEMORY_NEEDED
    ok = bytearray(j); ok[sof + 5:sof + 7] = (9000).to_bytes(2, "big"); ok[sof + 7:sof + 9] = (12000).to_bytes(2, "big")
    img = JpegImage(bytes(ok))    # 2.5 M blocks: inside the budget; the short scan simply ends early, like in the reference
    assert img.desc.width_blocks[0] == 1500


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_server_is_thread_and_memory_safe_under_load(sanitizer, tmp_path):
    """lep_serve.cc (IO thread + batcher thread + client-facing sockets) under ThreadSanitizer and AddressSanitizer: 24 client
    threads for a few seconds -- good requests, hang-ups mid-upload, answers never read, oversized and unknown files, failing
    files, a time bound and a connection cap in force (tests/fuzz/serve_stress.cc)"""
    exe = str(tmp_path / "serve_stress")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + sanitizer, "-fno-omit-frame-pointer", "-o", exe,
           os.path.join(FUZZ, "serve_stress.cc"), os.path.join(CSRC, "lep_serve.cc"), "-lz", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no %s sanitizer runtime for g++ here: %s" % (sanitizer, r.stderr[-200:]))
    sock = "/tmp/lep-stress-%d-%s" % (os.getpid(), sanitizer[:4])
    r = subprocess.run([exe, sock, "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert "good answers" in r.stderr



# The kernel headers themselves (coders, Huffman kernels, the parallel Huffman decoder) have been through their parity tests
# compiled with -fsanitize=address,undefined (397 tests at the end of round 2, no finding; an out-of-bounds LDS / model / frame index is silent on
# the GPU, not there).  By hand, because a preloaded sanitizer runtime inside pytest-in-pytest proved fragile:
#   ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
#   LEP_EMU_SO=/tmp/libcore_emu_san.so LEP_EMU_DEFINES="-fsanitize=address,undefined -fno-omit-frame-pointer -g" \
#   LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:log_path=/tmp/asanlog \
#   python -m pytest tests/test_core_emulation.py tests/test_slices.py -x -q -m "not gpu" -p no:cacheprovider


def test_lep_container_rules_found_by_the_structure_aware_fuzz():
    """what the hand-off section may hold, as the reference behaves (probed with the reference binary: 17 and 32 records abort
    in always_assert, 200 die with SIGSEGV; a second HH section replaces the first; the header may inflate to the announced
    JPEG size + 2048 bytes and no further, jpgcoder.cc:4158-4166)"""
    sys.path.insert(0, FUZZ)
    import mutate as mu
    from lepton_amd.codec import LepFile, LeptonError

    l4 = golden("q30_256x256_4seg")[1]
    for n in (17, 32, 200, 255):
        with pytest.raises(LeptonError) as e:
            LepFile(mu.with_handoffs(l4, count=n))
        assert e.value.code == 1
    f = LepFile(mu.with_handoffs(l4, repeat=3))
    assert len(f.segments) == len(LepFile(l4).segments) and [s.luma_y_start for s in f.segments] == [s.luma_y_start for s in LepFile(l4).segments]
    assert len(LepFile(mu.with_handoffs(l4, count=16, thread_byte=2)).segments) == 16
    fixed, payload, rest = mu.lep_split(l4)
    with pytest.raises(LeptonError):
        LepFile(mu.lep_join(fixed, payload + bytes(1 << 20), rest))     # header inflates far beyond jpeg_size + 2048
    with pytest.raises(LeptonError) as e:
        LepFile(mu.lep_join(fixed, payload + bytes(1000), rest))        # "unknown data found" -> errorlevel 2 (jpgcoder.cc:4326-4337)
    assert e.value.code == 42
    LepFile(mu.lep_join(fixed, payload + b"CMP" + bytes(50), rest))     # an explicit end mark stops the section loop


def test_recoder_rules_found_by_the_structure_aware_fuzz():
    """two things the differential fuzz against the reference binary turned up in the baseline re-coder, both reached only through
    a hostile flag byte: (1) a progressive header forced through it codes with Huffman tables no DHT before the first SOS
    defined -- the reference's tables are zeroed globals (zero-length codes), ours were uninitialised memory and the output
    changed from run to run; (2) an SOS whose length field reaches past the stored header -- the reference writes that many
    bytes from its (zero-filled) header arena, recoder.cc:443-456.  Outputs below were checked byte for byte against
    oracle/_ref/lepton (tests/fuzz/diff_lep_structured.py, seeds 4242 / 777 / 9001)."""
    sys.path.insert(0, FUZZ)
    import hashlib
    import mutate as mu
    import oracle_binding as ob
    from lepton_amd.codec import LepFile

    def restore(lep):
        f = LepFile(lep)
        ob.oracle_decode(f.desc, f.segments, f.streams)
        return f.recode()

    # (1) prog_truncated_mid with the flag byte forced odd: deterministic, and the short output of zero-length codes
    lep = bytearray(golden("prog_truncated_mid")[1]); lep[3] = 0x7F
    outs = {hashlib.md5(restore(bytes(lep))).hexdigest() for _ in range(4)}
    assert len(outs) == 1
    assert len(restore(bytes(lep))) < len(golden("prog_truncated_mid")[0])
    # (2) the SOS length of a baseline file set to 0xffff, flag byte odd: header, then zeros up to the file's size
    fixed, payload, rest = mu.lep_split(golden("lay_all22_64x48")[1])
    k = payload.rfind(b"\xff\xda\x00\x0c")
    assert k > 0
    bad = payload[: k + 2] + b"\xff\xff" + payload[k + 4:]
    fixed = bytearray(fixed); fixed[3] = 0x3F
    out = restore(mu.lep_join(bytes(fixed), bad, rest))
    jpg = golden("lay_all22_64x48")[0]
    assert len(out) == len(jpg)
    sos = out.find(b"\xff\xda\xff\xff")
    assert sos > 0 and out[: sos] == jpg[: sos] and set(out[sos + 14: -2]) == {0} and out[-2:] == b"\xff\xd9"


def test_blocks_behind_a_truncation_point_are_the_ring_rows():
    """a truncated baseline file whose second stream packet is dropped: the bool decoder runs into zeros, the scan comes out shorter
    than the file's byte bound, and what the re-coder writes for the blocks BEHIND the point where the JPEG was cut becomes
    visible: the reference's baseline decoder keeps two block rows per component and its re-coder reads row y - 2 there
    (block_based_image.hh:60-66,84-95), not zeros.  Output pinned to the reference binary's (5745 bytes, md5 below; found by the
    byte-level differential fuzz as "both succeed with different bytes")."""
    import hashlib
    import struct
    import oracle_binding as ob
    from lepton_amd.codec import LepFile

    lep = golden("truncated")[1]
    at = 28 + struct.unpack("<I", lep[24:28])[0] + 3
    assert lep[at] == 0x10                                   # a 4096-byte packet of stream 0
    cut = lep[: at + 1 + 4096] + lep[-4:]
    f = LepFile(cut)
    ob.oracle_decode(f.desc, f.segments, f.streams)
    out = f.recode()
    assert len(out) == 5745 and hashlib.md5(out).hexdigest() == "7b0b34cb4ca9001f1b4f321619826b63"


def test_lazy_decode_rules_of_truncated_files():
    """three more places where what the reference's re-coder reads for a truncated file is not what its stream codes -- all invisible in
    an intact file (the byte bound cuts the output first), all found by the differential fuzz with damaged streams or headers:
    a baseline row's FIRST block is decoded even behind the truncation point (decode_row always takes it), so the two-row-ring rule
    excludes it; a one-thread PROGRESSIVE file's blocks behind the point are never decoded at all (the decoder only runs as far as
    the re-coder waits for it, and that wait is clamped to the last block the JPEG held); and a DC category > 32 from a corrupt DHT,
    used as a bit count, consumes what is left of the reference's 8-byte buffer plus one refill."""
    import hashlib
    import struct
    import zlib
    sys.path.insert(0, FUZZ)
    import mutate as mu
    import oracle_binding as ob
    from lepton_amd.codec import JpegImage, LepFile, LeptonError

    def restore(lep):
        f = LepFile(lep)
        ob.oracle_decode(f.desc, f.segments, f.streams)
        return f.recode()

    # header cut inside the EEE section: every component "held" one block; row 1's first luma block is still decoded and written
    fixed, payload, rest = mu.lep_split(golden("truncated")[1])
    k = payload.find(b"EEE")
    out = restore(mu.lep_join(fixed, payload[: k + 14], rest))
    assert len(out) == 724 and hashlib.md5(out).hexdigest() == "6ee884c157102edb5dcd2379d2f3e783"
    # progressive, one thread, one damaged stream byte: the tail block the stream codes is written as zeros
    lep = bytearray(golden("prog_truncated_dc")[1]); lep[844] ^= 4
    out = restore(bytes(lep))
    assert len(out) == 1100 and hashlib.md5(out).hexdigest() == "2c92db3a672a0476b06a1baaa7fe55d1"
    # a DC Huffman table with category 0x58: both sides get as far as the coder and refuse with COEFFICIENT_OUT_OF_RANGE
    jpg = bytearray(golden("trailing_garbage")[0]); assert jpg[205] == 0x07; jpg[205] = 0x58
    img = JpegImage(bytes(jpg))
    with pytest.raises(RuntimeError) as e:
        ob.oracle_encode(img.desc, img.plan())
    assert "exit code 6" in str(e.value)


def test_a_truncated_multi_segment_file_is_refused_like_the_reference():
    """the state a thread segment ends in must be the state the next hand-off recorded (recode_physical_thread, recoder.cc:625-640:
    partial byte and its bit count, last DC per component, a bound filled exactly): a .lep cut inside its first segment's stream
    decodes "successfully" into zeros, and it is these assertions that refuse it (the reference aborts with "Assert Failed:
    outth.num_overhang_bits == ..."); cut further back, both sides restore the same shorter file.  Probed against the reference
    binary at six cut points of two fixtures."""
    import hashlib
    import oracle_binding as ob
    from lepton_amd.codec import LepFile, LeptonError

    lep = golden("q30_256x256_4seg")[1]
    for frac in (0.15, 0.30, 0.45):
        f = LepFile(lep[: int(len(lep) * frac)])
        ob.oracle_decode(f.desc, f.segments, f.streams)
        with pytest.raises(LeptonError) as e:
            f.recode()
        assert e.value.code == 1
    for frac in (0.60, 0.97):          # the second segment's stream is what is missing: nothing follows it to be held against
        f = LepFile(lep[: int(len(lep) * frac)])
        ob.oracle_decode(f.desc, f.segments, f.streams)
        assert len(f.recode()) == 98760
    f = LepFile(lep)
    ob.oracle_decode(f.desc, f.segments, f.streams)
    assert f.recode() == golden("q30_256x256_4seg")[0]


def hand_off_field_cases():
    """[(mutated .lep, what the reference binary answers: (restored length, md5) or its exit code)] -- see
    test_hand_off_fields_a_damaged_file_can_carry"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz"))
    import mutate as mu
    from conftest import GOLDEN

    def v2(n):
        return open(os.path.join(GOLDEN, "v2", n + ".lep"), "rb").read()

    g = lambda n: golden(n)[1]
    two = lambda lep, a, b: mu.with_handoffs(mu.with_handoffs(lep, field=a), field=b)
    cases = [
        (mu.with_handoffs(g("lay_mixed_200x120"), field=(0, 0, 2, 1)), (11705, "8398c8cf9247384a9eb5c42cb69b2351")),
        (mu.with_handoffs(g("truncated"), field=(0, 0, 2, 7)), (3458, "8a1a4b929284954d465b2105fc31cb81")),
        (mu.with_handoffs(v2("narrowrst"), field=(0, 7, 1, 0x43)), (35243, "2c8cf49462c47d283b569fabde42087c")),
        (mu.with_handoffs(v2("rst_c420_176x112"), field=(0, 7, 1, 0xb6)), (5108, "09c67225382fbf90c4cad8547a5c235d")),
        (mu.with_handoffs(g("lay_mixed_200x120"), field=(0, 7, 1, 0xa3)), (13235, "646533b09590f8141c9bc6261f19fbd6")),
        (mu.with_handoffs(g("lay_mixed_200x120"), field=(0, 7, 1, 0x40)), (13235, "54e435f09a320c604a8f8c67e02dc58d")),
        (two(g("c420_160x120"), (0, 7, 1, 9), (0, 6, 1, 0xff)), (5799, "e6c9c839632a4a0d774b326a978dc4ff")),
        (two(g("c420_160x120"), (0, 7, 1, 3), (0, 6, 1, 0xff)), (5801, "497555f575b54e08ced6294cec536a0f")),
        (mu.with_handoffs(v2("gray_120x88"), field=(0, 2, 4, 0xffffffff)), 1),
        (mu.with_handoffs(g("q30_256x256_4seg"), field=(1, 2, 4, 0x7fffffff)), 37),
        (mu.with_handoffs(g("q30_256x256_4seg"), field=(1, 2, 4, 500 << 20)), (178322, "74fcf2c3adf1fb8dd53b30d717b329f5")),
        (mu.with_handoffs(v2("q30_256x256_4seg"), field=(1, 7, 1, 0xff)), (178304, "f69435c2a9144ad58b1b660d09c681ca")),
        (mu.with_handoffs(g("q30_256x256_4seg"), field=(1, 7, 1, 0xff)), (178304, "f69435c2a9144ad58b1b660d09c681ca")),
        (mu.with_handoffs(v2("q30_256x256_4seg"), field=(1, 2, 4, 0)), (89187, "adf95715ee513e460ee019f730634684")),   # format 2: a worker bound of zero bytes
    ]
    return cases


def test_hand_off_fields_a_damaged_file_can_carry():
    """field-level mutants of the hand-off records (tests/fuzz/mutate.py with_handoffs(field=...), format 1 and 2), each pinned
    to what the reference binary answers (tests/fuzz/diff_lep_structured.py found them):
    * luma_y_start inside an MCU row: the baseline re-coder takes MCU rows whose FIRST luma row is not in front of it
      (recode_row_range, recoder.cc:505-510) -- the decoder starts at the next MCU row, as a top row;
    * overhang bit counts of 8..64 are the byte plus zero bits; beyond 64 the reference's 64-bit buffer goes negative: the first
      value written is widened by the excess or, past the buffer's width, dropped with 64 zero bits in its place (bitops.hh:120-163);
      bits of the byte below the count it claims are OR-ed with what is written next;
    * format 2: the first thread's bound is a 32-bit sum that wraps in front of what is written: assertion (bitops.cc:402);
    * a worker's buffer is allocated at its bound from a 576 MiB arena: OOM before anything is decoded (recoder.cc:770-782);
    * a pre-hand-off record (bit count 0xff) at the head of a worker's range starts clean from its own record (recoder.cc:584-592)"""
    import hashlib
    import oracle_binding as ob
    from lepton_amd.codec import LepFile, LeptonError

    def restore(b):
        try:
            f = LepFile(b)
            ob.oracle_decode(f.desc, f.segments, f.streams)
            d = f.recode()
            return len(d), hashlib.md5(d).hexdigest()
        except LeptonError as e:
            return e.code

    for i, (lep, want) in enumerate(hand_off_field_cases()):
        assert restore(lep) == want, i


def test_sos_length_past_the_header_of_a_progressive_file():
    """the general re-coder's merge writes "null bytes beyond buffer" for an SOS length field that reaches past the stored header
    (merge_jpeg_streaming, jpgcoder.cc:2594-2598) -- found by the structured sweep over progressive fixtures, pinned to the
    reference's output; and seventeen hand-offs are CODING_ERROR there, not the baseline re-coder's assertion"""
    import hashlib
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz"))
    import mutate as mu
    import oracle_binding as ob
    from lepton_amd.codec import LepFile, LeptonError

    lep = golden("prog_truncated_dc")[1]
    fixed, payload, rest = mu.lep_split(lep)
    i = payload.index(b"\xff\xda\x00\x0c")
    f = LepFile(mu.lep_join(fixed, payload[: i + 2] + b"\xff" + payload[i + 3:], rest))
    ob.oracle_decode(f.desc, f.segments, f.streams)
    out = f.recode()
    assert len(out) == 1100 and hashlib.md5(out).hexdigest() == "a33a16cf2195a549bb63e213c301a940"
    with pytest.raises(LeptonError) as e:
        LepFile(mu.with_handoffs(golden("prog_c420_320x240")[1], count=17))
    assert e.value.code == 2


def test_general_re_coder_no_scan_and_unbound_stream_ids():
    """two more answers of the general re-coder, found on larger progressive files: a header walk that never meets an SOS (a
    segment length that reaches past the header) leaves the reference's scan table empty -- "out of memory error", errorlevel 2
    (jpgcoder.cc:3708-3714); a packet for a stream id that no hand-off created is `Cannot send to thread that wasn't bound`
    (vp8_decoder.cc:236).  Probed against the reference binary (its answers: 42, abort)."""
    import io
    import os
    import sys
    import numpy as np
    from PIL import Image
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz"))
    import mutate as mu
    import oracle_binding as ob
    from lepton_amd.codec import JpegImage, LepFile, LeptonError

    fixed, payload, rest = mu.lep_split(golden("prog_c420_320x240")[1])
    i = payload.index(b"\xff\xc2\x00\x11")
    with pytest.raises(LeptonError) as e:      # answered when the file is opened: with no scan nothing is ever decoded
        LepFile(mu.lep_join(fixed, payload[: i + 2] + b"\xff\xff" + payload[i + 4:], rest))
    assert e.value.code == 42

    rng = np.random.default_rng(8)
    base = rng.integers(0, 256, (16, 20, 3), dtype=np.uint8)
    a = np.asarray(Image.fromarray(base).resize((500, 400), Image.BICUBIC)).astype(np.int16)
    a = np.clip(a + rng.normal(0, 14, a.shape), 0, 255).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format="JPEG", quality=97, subsampling=0, progressive=True)
    img = JpegImage(buf.getvalue())
    segs = img.plan()
    assert len(segs) > 1                      # a progressive file of this size is coded on several threads
    streams, _ = ob.oracle_encode(img.desc, segs)
    lep = img.write_lep(streams)
    g = LepFile(lep)
    ob.oracle_decode(g.desc, g.segments, g.streams)
    assert g.recode() == buf.getvalue()
    with pytest.raises(LeptonError) as e:
        LepFile(mu.with_handoffs(lep, count=1))
    assert e.value.code == 1
