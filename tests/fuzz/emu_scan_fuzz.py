"""Sweep of the lane-per-piece JPEG scan kernels (lep_huffdec_simt.h, lep_huff_simt.h) as lane-loop emulations against the
wavefront kernels they replace (lep_huffdec.h, lep_huff.h) -- test infrastructure, run by hand:
`python tests/fuzz/emu_scan_fuzz.py <seed> <cases>`.  Files are drawn as quantised coefficients (tests/jpeg_writer.py): geometry
from one block up, every sampling layout the writer knows, density from almost empty to every coefficient set, amplitudes up to
the 8-bit limits, default and per-file Huffman tables.  For every file the GPU path is eligible for:
  decode: frame, hand-off records and pad bit of the lane-per-subsequence decoder == the single-wave decoder's, at three
          subsequence lengths (a status instead is allowed only where the subsequences are too short to fall into step in);
  encode: every segment's bytes, byte count (under its own bound and under one that cuts it short) and end state from the
          lane-per-unit encoder == the wavefront-per-segment encoder's.
Prints one line per mismatch and a summary."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import jpeg_writer as jw          # noqa: E402
import oracle_binding as ob       # noqa: E402
from emu_coeff_fuzz import LAYOUTS  # noqa: E402

from lepton_amd import abi  # noqa: E402
from lepton_amd.codec import JpegImage, LepFile, LeptonError  # noqa: E402


def decode_setup(L, jpg):
    h = C.c_void_p()
    img = abi.HuffDecImage()
    ok = C.c_int(0)
    if L.lep_jpeg_open_gpu(jpg, len(jpg), C.byref(h), C.byref(img), C.byref(ok)) != 0:
        return None
    if not ok.value:
        L.lep_jpeg_close(h)
        return None
    p, n = C.c_void_p(), C.c_size_t(0)
    L.lep_jpeg_scan_bytes(h, C.byref(p), C.byref(n))
    if img.flags & 2:   # LEP_HUFFDEC_RST_TABLE: the markers' positions behind the scan bytes, at LEP_HUFFDEC_SCAN_ROOM(scan_len)
        rp, rn = C.POINTER(C.c_uint32)(), C.c_size_t(0)
        L.lep_jpeg_scan_restarts(h, C.byref(rp), C.byref(rn))
        room = (n.value + 64 + 15) & ~15
        table = bytes(C.cast(rp, C.POINTER(C.c_uint8 * (4 * rn.value))).contents) if rn.value else b""
        scan = C.create_string_buffer(C.string_at(p, n.value) + bytes(room - n.value) + table + bytes(64), room + len(table) + 64)
    else:
        scan = C.create_string_buffer(C.string_at(p, n.value) + bytes(64), n.value + 64)
    img.scan = C.addressof(scan)
    d = JpegImage(jpg).desc
    planes = [C.create_string_buffer(d.nblocks(c) * 128) for c in range(d.ncomp)]
    for c in range(d.ncomp):
        img.blocks[c] = C.cast(planes[c], C.c_void_p).value
    L.lep_jpeg_close(h)
    return img, scan, planes, d


def main():
    seed0, cases = int(sys.argv[1]), int(sys.argv[2])
    so = os.path.join(ROOT, "tests", "emu", "libcore_emu_fuzz.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "emu", "core_emu.cc")])
    emu = C.CDLL(so)
    L = abi.lib()
    dec_ran = dec_gave_up = enc_ran = enc_segments = skipped = bad = rst_ran = 0
    for k in range(cases):
        rng = np.random.default_rng(seed0 * 100003 + k)
        comps = LAYOUTS[rng.integers(len(LAYOUTS))]
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 420))
        if rng.random() < 0.25:
            w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        kw = dict(quality=int(rng.choice([1, 10, 40, 75, 90, 98, 100])), density=float(rng.choice([0.01, 0.05, 0.25, 0.6, 1.0, 4.0])),
                  amp=float(rng.choice([0.5, 4, 40, 200, 900])), restart_interval=int(rng.choice([0, 0, 0, 1, 2, 3, 5, 8, 9, 17, 100])))
        try:
            jpg = jw.write_baseline(w, h, comps, rng, **kw)[0]
            src = JpegImage(jpg)
        except (LeptonError, ValueError, AssertionError):
            skipped += 1
            continue
        # ---- decode direction
        one = decode_setup(L, jpg)
        if one is not None and not one[0].rsti:
            img1, scan1, planes1, d = one
            rows1 = (abi.HuffDecRow * (img1.mcuv + 1))()
            emu.emu_huffman_decode_image(C.byref(img1), rows1)
            if rows1[img1.mcuv].aux >> 8 == 0:
                blocks = sum(d.nblocks(c) for c in range(d.ncomp))
                for sub_bits in (1024, 8192, 32768):
                    img2, scan2, planes2, _ = decode_setup(L, jpg)
                    rows2 = (abi.HuffDecRow * (img2.mcuv + 1))()
                    moved = (C.c_int32 * 8)()
                    emu.emu_huffman_decode_image_simt(C.byref(img2), rows2, sub_bits, moved, None)
                    dec_ran += 1
                    if rows2[img2.mcuv].aux >> 8:
                        if (sub_bits < 8192 or sub_bits < 64 * img2.scan_len * 8 // blocks) and moved[3]:
                            dec_gave_up += 1
                        else:
                            bad += 1
                            print("DECODE GAVE UP", seed0, k, w, h, comps, kw, sub_bits, list(moved)[:4], flush=True)
                        continue
                    same = all(planes2[c].raw == planes1[c].raw for c in range(d.ncomp)) and all(
                        (rows2[r].bitpos, list(rows2[r].last_dc), rows2[r].aux) == (rows1[r].bitpos, list(rows1[r].last_dc), rows1[r].aux) for r in range(img1.mcuv + 1))
                    if not same:
                        bad += 1
                        print("DECODE MISMATCH", seed0, k, w, h, comps, kw, sub_bits, flush=True)
        # ---- decode direction, restart intervals: lane = interval (the markers' positions behind the scan bytes) against the single-wave
        #      decoder, which walks the scan as the reference does
        if one is not None and one[0].rsti and (one[0].flags & 2):
            img1, scan1, planes1, d = one
            rows1 = (abi.HuffDecRow * (img1.mcuv + 1))()
            img1.flags &= ~2   # (the single-wave kernel reads no table)
            emu.emu_huffman_decode_image(C.byref(img1), rows1)
            if rows1[img1.mcuv].aux >> 8 == 0:
                img2, scan2, planes2, _ = decode_setup(L, jpg)
                rows2 = (abi.HuffDecRow * (img2.mcuv + 1))()
                emu.emu_huffman_decode_image_simt(C.byref(img2), rows2, 8192, None, None)
                rst_ran += 1
                if (rows2[img2.mcuv].aux >> 8) & 0x3fffff:
                    bad += 1
                    print("RST DECODE GAVE UP", seed0, k, w, h, comps, kw, flush=True)
                else:
                    same = all(planes2[c].raw == planes1[c].raw for c in range(d.ncomp)) and all(
                        (rows2[r].bitpos, list(rows2[r].last_dc), rows2[r].aux & 0xff) == (rows1[r].bitpos, list(rows1[r].last_dc), rows1[r].aux & 0xff) for r in range(img1.mcuv + 1))
                    if not same:
                        bad += 1
                        print("RST DECODE MISMATCH", seed0, k, w, h, comps, kw, flush=True)
        # ---- encode direction
        try:
            segs0 = src.plan()
            streams, _ = ob.oracle_encode(src.desc, segs0)
            lep = src.write_lep(streams)
            f = LepFile(lep)
        except (LeptonError, RuntimeError):
            continue
        for c in range(f.desc.ncomp):
            C.memmove(f.desc.blocks[c], src.desc.blocks[c], f.desc.nblocks(c) * 128)
        himg = abi.HuffImage()
        hsegs = (abi.HuffSegment * abi.MAX_SEGMENTS)()
        nseg, ok = C.c_int(0), C.c_int(0)
        if L.lep_file_recode_plan(f.handle, C.byref(himg), hsegs, C.byref(nseg), C.byref(ok)) != 0 or not ok.value:
            continue
        enc_ran += 1
        for i in range(nseg.value):
            for cap in (min(hsegs[i].out_cap, len(jpg) + 1024), 37):
                hsegs[i].out_cap = cap
                outs = []
                for fn in (emu.emu_huffman_encode_segment, emu.emu_huffman_encode_segment_simt):
                    buf = C.create_string_buffer(cap + 8)
                    n = C.c_uint32(0)
                    end = abi.HuffEnd()
                    rc = fn(C.byref(himg), C.byref(hsegs[i]), buf, C.byref(n), C.byref(end))
                    outs.append((rc, n.value, buf.raw[: n.value], end.overhang_byte, end.num_overhang_bits, list(end.last_dc)))
                if outs[1][0] == 1:
                    continue
                enc_segments += 1
                if outs[0] != outs[1]:
                    bad += 1
                    print("ENCODE MISMATCH", seed0, k, w, h, comps, kw, "segment", i, "cap", cap, outs[0][1], outs[1][1], outs[0][3:], outs[1][3:], flush=True)
    print(f"seed {seed0}: decodes {dec_ran} (gave up where allowed {dec_gave_up}), restart-interval decodes {rst_ran}, files encoded {enc_ran} ({enc_segments} segment runs), not a file {skipped}, bad {bad}")


if __name__ == "__main__":
    main()
