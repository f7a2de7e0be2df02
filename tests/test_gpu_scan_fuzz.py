"""The JPEG scan kernels of the HIP BUILD under a drawn corpus: round 5's sweeps of the lane-per-piece
scan kernels (tests/fuzz/emu_scan_fuzz.py, emu_cut_fuzz.py) ran the g++ lane-loop emulation -- and it was one file on the GPU box that
caught a compiler mode changing a kernel's output.  Here the same kind of corpus goes through the library the GPU box loads:

  * 640 files drawn (about 560 of them codable) as quantised coefficients (tests/jpeg_writer.py: eight sampling layouts from one block up, densities to "every
    coefficient set", amplitudes to the 8-bit limits, restart intervals of 1 .. 100 MCUs in half of them, files cut inside their scan);
  * 24 PIL-written files large enough for several thread segments, with restart intervals that make segments START INSIDE an interval
    with a partial byte (the case the compiler mode broke), and 8 progressive files;
  * compress: lep_compress_batch (GPU scan decoders: lane per subsequence / lane per restart interval / window of speculative codes)
    must write, for every file, the bytes of the .lep the CPU ORACLE's streams make (tests/oracle_binding.py -- the checker; the
    container around them is the host library's) or refuse with the per-file path's code;
  * decompress: lep_decompress_batch (GPU scan encoders: lane per unit / lane per 32 blocks) must restore every file byte for byte.
"""
import os
import sys

import numpy as np
import pytest

import oracle_binding as ob
from lepton_amd import corpus
from lepton_amd.codec import GpuCodec, JpegImage, LeptonError

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz"))

pytestmark = pytest.mark.gpu


def _drawn(seed0, cases):
    import jpeg_writer as jw
    from emu_coeff_fuzz import LAYOUTS

    out = []
    for k in range(cases):
        rng = np.random.default_rng(seed0 * 100003 + k)
        comps = LAYOUTS[rng.integers(len(LAYOUTS))]
        w, h = int(rng.integers(1, 500)), int(rng.integers(1, 300))
        if rng.random() < 0.25:
            w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        kw = dict(quality=int(rng.choice([1, 10, 40, 75, 90, 98, 100])), density=float(rng.choice([0.01, 0.05, 0.25, 0.6, 1.0, 4.0])),
                  amp=float(rng.choice([0.5, 4, 40, 200, 900])), restart_interval=int(rng.choice([0, 0, 0, 0, 1, 2, 3, 5, 8, 9, 17, 33, 64, 100])))
        try:
            jpg = jw.write_baseline(w, h, comps, rng, **kw)[0]
        except (ValueError, AssertionError):
            continue
        cut = False
        if rng.random() < 0.12:      # a file cut inside its scan (no EOI)
            sos = jpg.find(b"\xff\xda")
            jpg = jpg[: int(rng.integers(sos + 14, len(jpg) - 1))]
            cut = True
        out.append((jpg, dict(k=k, w=w, h=h, comps=[c[1:3] for c in comps], cut=cut, **kw)))
    return out


def _large(seed0):
    import io

    from PIL import Image

    out = []
    rng = np.random.default_rng(seed0)
    for k in range(24):
        w, h = [(1920, 1080), (1280, 960), (2048, 1536), (1600, 1200)][k % 4]
        a = np.asarray(Image.open(io.BytesIO(corpus.synth_jpeg(w, h, 7000 + k, quality=int(rng.choice([75, 90, 96]))))).convert("RGB"))
        buf = io.BytesIO()
        kw = dict(format="JPEG", quality=int(rng.choice([75, 90, 96])), subsampling=int(rng.choice([0, 1, 2])))
        kw["restart_marker_blocks"] = int(rng.choice([1, 2, 3, 7, 13, 50, 100, 1000]))
        Image.fromarray(a).save(buf, **kw)
        out.append((buf.getvalue(), dict(large=k, w=w, h=h, **kw)))
    out += [(corpus.synth_jpeg(640 + 64 * k, 480 + 16 * k, 7100 + k, progressive=True, quality=[30, 75, 92, 97][k % 4], subsampling=["4:2:0", "4:4:4", "4:2:2"][k % 3]), dict(progressive=k)) for k in range(8)]
    return out


def test_gpu_scan_kernels_on_a_drawn_corpus():
    drawn = _drawn(611, 640) + _large(612)
    jpgs = [j for j, _ in drawn]
    meta = [m for _, m in drawn]
    want = []
    for j in jpgs:
        try:
            img = JpegImage(j)
            streams, _ = ob.oracle_encode(img.desc, img.plan())
            want.append((0, img.write_lep(streams)))
        except LeptonError as e:
            want.append((e.code, None))
        except RuntimeError as e:            # the oracle's own refusal ("oracle encode exit code 6": a coefficient the coder does not take)
            want.append((int(str(e).split()[-1]), None))
    assert sum(1 for c, _ in want if c == 0) >= 500
    codec = GpuCodec(0)
    try:
        got, st, cstats = codec.compress_batch(jpgs, chunk_images=128)
        bad = [i for i in range(len(jpgs)) if (st[i], got[i] if st[i] == 0 else None) != want[i]]
        assert not bad, [(i, st[i], want[i][0], meta[i]) for i in bad[:12]]
        ok = [i for i in range(len(jpgs)) if st[i] == 0]
        back, st2, dstats = codec.decompress_batch([got[i] for i in ok], chunk_images=128)
        # (one class of files the REFERENCE cannot restore: a single component with 2x2 sampling factors and restart markers, its
        # images/roundtripfail.jpg -- `lepton` without -skipverify answers ROUNDTRIP_FAILURE, the decode direction is bug-compatible.
        # For those the expectation is what the per-file path writes: the host re-coder, pinned to the reference's bytes by the fixture
        # rtfail_gray22_rst_64x64)
        def expected(i):
            m = meta[i]
            if m.get("comps") == [(2, 2)] and m.get("restart_interval"):
                return codec.decompress(got[i])
            return jpgs[i]
        bad2 = [(ok[k], st2[k], len(back[k] or b""), len(jpgs[ok[k]]), meta[ok[k]]) for k in range(len(ok)) if st2[k] != 0 or back[k] != expected(ok[k])]
        assert not bad2, (len(bad2), bad2[:12])
        # the kernels were what ran: most of the corpus is theirs in both directions
        assert cstats["gpu_huffman_files"] >= 0.5 * len(ok) and dstats["gpu_huffman_files"] >= 0.5 * len(ok), (cstats, dstats)
    finally:
        codec.close()
