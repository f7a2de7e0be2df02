"""Files cut inside their scan (no EOI) through the lane-per-piece scan kernels as lane-loop emulations -- test infrastructure, run by hand:
`python tests/fuzz/emu_cut_fuzz.py <seed> <cases>`.  Files are drawn like emu_scan_fuzz.py's (tests/jpeg_writer.py: every layout, densities,
amplitudes, table kinds) without restart intervals, then cut at a drawn byte of the scan -- in the first MCU row, anywhere, in the last
bytes, behind an FF.  For every cut the host parser accepts:
  decode: lep_huffdec_simt.h + parse_jpeg_finish_gpu leave the host parser's frame (coded blocks), truncation bounds and .lep -- or a status
          (the file is the host parser's); never a different result;
  encode: lep_huff_simt.h's segments glued by lep_file_recode_finish are the file, or LEP_GPU_PATH_DECLINED (the cut met before the byte
          bound: the host re-coder's); never different bytes.
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


def main():
    seed0, cases = int(sys.argv[1]), int(sys.argv[2])
    so = os.path.join(ROOT, "tests", "emu", "libcore_emu_fuzz.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "emu", "core_emu.cc")])
    emu = C.CDLL(so)
    L = abi.lib()
    dec_ok = dec_status = dec_ineligible = enc_ok = enc_declined = enc_ineligible = refused = bad = 0
    for k in range(cases):
        rng = np.random.default_rng(seed0 * 100019 + k)
        comps = LAYOUTS[rng.integers(len(LAYOUTS))]
        w, h = int(rng.integers(9, 500)), int(rng.integers(9, 320))
        kw = dict(quality=int(rng.choice([10, 40, 75, 90, 98])), density=float(rng.choice([0.05, 0.25, 0.6, 1.0])), amp=float(rng.choice([4, 40, 200, 900])))
        try:
            whole = jw.write_baseline(w, h, comps, rng, **kw)[0]
            JpegImage(whole)
        except (LeptonError, ValueError, AssertionError):
            continue
        sos = whole.find(b"\xff\xda")
        first = sos + 2 + ((whole[sos + 2] << 8) | whole[sos + 3])
        kind = int(rng.integers(4))
        if kind == 0:
            cut = int(rng.integers(first + 1, min(len(whole) - 2, first + 200) + 1))
        elif kind == 1:
            cut = int(rng.integers(first + 1, len(whole) - 1))
        elif kind == 2:
            cut = int(rng.integers(max(first + 1, len(whole) - 60), len(whole) - 1))
        else:
            ffs = [i + 1 for i in range(first, len(whole) - 3) if whole[i] == 0xFF]
            cut = int(rng.choice(ffs)) if ffs else len(whole) - 2
        jpg = whole[:cut]
        try:
            host = JpegImage(jpg)
        except LeptonError:
            refused += 1
            continue
        d = host.desc
        tag = (seed0, k, w, h, comps, kw, "cut", cut, "of", len(whole))
        # ---- decode direction
        hdl = C.c_void_p()
        img = abi.HuffDecImage()
        ok = C.c_int(0)
        if L.lep_jpeg_open_gpu(jpg, len(jpg), C.byref(hdl), C.byref(img), C.byref(ok)) != 0:
            bad += 1
            print("OPEN_GPU REFUSES WHAT THE HOST PARSER TAKES", *tag, flush=True)
            continue
        if not ok.value:
            dec_ineligible += 1
            L.lep_jpeg_close(hdl)
        else:
            p, n = C.c_void_p(), C.c_size_t(0)
            L.lep_jpeg_scan_bytes(hdl, C.byref(p), C.byref(n))
            scan = C.create_string_buffer(C.string_at(p, n.value) + b"\0" * 64, n.value + 64)
            img.scan = C.addressof(scan)
            planes = []
            for c in range(d.ncomp):
                b = C.create_string_buffer(d.nblocks(c) * 128)
                planes.append(b)
                img.blocks[c] = C.cast(b, C.c_void_p).value
            rows = (abi.HuffDecRow * (img.mcuv + 1))()
            emu.emu_huffman_decode_image_simt(C.byref(img), rows, int(rng.choice([1024, 8192])), None, None)
            if (rows[img.mcuv].aux >> 8) & 0x3fffff or L.lep_jpeg_finish_gpu(hdl, rows) != 0:
                dec_status += 1
            else:
                gd = abi.ImageDesc()
                L.lep_jpeg_describe(hdl, C.byref(gd))
                same = all(gd.coded_blocks[c] == d.coded_blocks[c] and gd.coded_height[c] == d.coded_height[c] and
                           planes[c].raw[: d.coded_blocks[c] * 128] == C.string_at(d.blocks[c], d.coded_blocks[c] * 128) for c in range(d.ncomp))
                streams = None
                if same:
                    try:
                        streams, _ = ob.oracle_encode(d, host.plan())
                    except RuntimeError:
                        pass    # (coefficients the coder refuses: COEFFICIENT_OUT_OF_RANGE -- the frame comparison above is the test)
                if same and streams is not None:
                    want = host.write_lep(streams)
                    arr = (abi.Bytes * len(streams))()
                    keep = []
                    for i, s in enumerate(streams):
                        b = C.create_string_buffer(bytes(s), max(1, len(s)))
                        keep.append(b)
                        arr[i].data = C.cast(b, C.c_void_p).value
                        arr[i].len = arr[i].cap = len(s)
                    out = abi.Bytes()
                    same = L.lep_jpeg_write_lep(hdl, 0, arr, len(streams), C.byref(out)) == 0 and out.tobytes() == want
                    if out.data:
                        L.lep_free(out.data)
                if same:
                    dec_ok += 1
                else:
                    bad += 1
                    print("DECODE MISMATCH", *tag, flush=True)
            L.lep_jpeg_close(hdl)
        # ---- encode direction
        try:
            segs0 = host.plan()
            streams, _ = ob.oracle_encode(d, segs0)
            f = LepFile(host.write_lep(streams))
            ob.oracle_decode(f.desc, f.segments, f.streams)
            restored = f.recode()
        except (LeptonError, RuntimeError):
            continue
        if restored != jpg:
            continue     # (the host re-coder itself does not restore this cut: the reference's ROUNDTRIP_FAILURE class, not this tool's business)
        himg = abi.HuffImage()
        hsegs = (abi.HuffSegment * abi.MAX_SEGMENTS)()
        nseg, okc = C.c_int(0), C.c_int(0)
        if L.lep_file_recode_plan(f.handle, C.byref(himg), hsegs, C.byref(nseg), C.byref(okc)) != 0 or not okc.value:
            enc_ineligible += 1
            continue
        n = nseg.value
        bufs, arr, ends = [], (abi.Bytes * n)(), (abi.HuffEnd * n)()
        taken = True
        for i in range(n):
            cap = min(hsegs[i].out_cap, len(jpg) + 1024)
            hsegs[i].out_cap = cap
            buf = C.create_string_buffer(cap + 8)
            ln = C.c_uint32(0)
            if emu.emu_huffman_encode_segment_simt(C.byref(himg), C.byref(hsegs[i]), buf, C.byref(ln), C.byref(ends[i])) != 0:
                taken = False
                break
            bufs.append(buf)
            arr[i].data = C.cast(buf, C.c_void_p).value
            arr[i].len = arr[i].cap = ln.value
        if not taken:
            if any(himg.trunc_bc[c] for c in range(4)):
                bad += 1
                print("ENCODER LEAVES A SEGMENT OF A CUT FILE THE PLAN LET THROUGH", *tag, flush=True)
            continue
        out = abi.Bytes()
        rc = L.lep_file_recode_finish(f.handle, arr, ends, n, C.byref(out))
        if rc == 101:
            enc_declined += 1
        elif rc == 0 and out.tobytes() == jpg:
            enc_ok += 1
        else:
            bad += 1
            print("ENCODE MISMATCH", *tag, "rc", rc, flush=True)
        if out.data:
            L.lep_free(out.data)
    print(f"seed {seed0}: cuts the host parser refuses {refused}; decode: same {dec_ok}, left to the host parser {dec_status}, not eligible {dec_ineligible}; "
          f"encode: restored {enc_ok}, declined {enc_declined}, not eligible {enc_ineligible}; bad {bad}")


if __name__ == "__main__":
    main()
