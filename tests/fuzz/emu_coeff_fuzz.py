"""Coefficient-domain sweep of the CURRENT coder kernels (single-kernel encoder, split-phase encoder, decoder) as lane-loop emulations against the oracle (test infrastructure, run by
hand: `python tests/fuzz/emu_coeff_fuzz.py <seed> <cases>`).  Frames are drawn directly as quantised coefficients
(tests/jpeg_writer.py): geometry from one block up, every sampling layout the writer knows, density from almost empty to every
coefficient set, amplitudes up to the 8-bit limits, flat and steep quantisation tables, restart intervals.  For every case:
encoder streams == oracle streams, decoder frames == oracle frames.  Prints one line per mismatch and a summary."""
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

from lepton_amd.codec import JpegImage, LeptonError  # noqa: E402

LAYOUTS = [
    [(1, 1, 1, 0, 0, 0)],
    [(1, 1, 1, 0, 0, 0), (2, 1, 1, 1, 1, 1), (3, 1, 1, 1, 1, 1)],
    [(1, 2, 1, 0, 0, 0), (2, 1, 1, 1, 1, 1), (3, 1, 1, 1, 1, 1)],
    [(1, 2, 2, 0, 0, 0), (2, 1, 1, 1, 1, 1), (3, 1, 1, 1, 1, 1)],
    [(1, 1, 2, 0, 0, 0), (2, 1, 1, 1, 1, 1), (3, 1, 1, 1, 1, 1)],
    [(1, 2, 2, 0, 0, 0), (2, 2, 1, 1, 1, 1), (3, 1, 2, 1, 1, 1)],
    [(1, 2, 2, 0, 0, 0)],
    [(1, 4, 1, 0, 0, 0), (2, 1, 1, 1, 1, 1), (3, 1, 1, 1, 1, 1)],
]


def main():
    seed0, cases = int(sys.argv[1]), int(sys.argv[2])
    so = os.path.join(ROOT, "tests", "emu", "libcore_emu_fuzz.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "emu", "core_emu.cc")])
    emu = C.CDLL(so)
    ran = bad = refused = refused_both = 0
    for k in range(cases):
        rng = np.random.default_rng(seed0 * 100003 + k)
        comps = LAYOUTS[rng.integers(len(LAYOUTS))]
        w, h = int(rng.integers(1, 330)), int(rng.integers(1, 200))
        if rng.random() < 0.3:
            w, h = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        kw = dict(quality=int(rng.choice([1, 10, 40, 75, 90, 98, 100])), density=float(rng.choice([0.01, 0.05, 0.25, 0.6, 1.0, 4.0])),
                  amp=float(rng.choice([0.5, 4, 40, 200, 900])), restart_interval=int(rng.choice([0, 0, 0, 1, 3, 7])))
        try:
            jpg = jw.write_baseline(w, h, comps, rng, **kw)[0]
            img = JpegImage(jpg)
        except (LeptonError, ValueError, AssertionError) as e:
            refused += 1
            continue
        d, segs = img.desc, img.plan()
        try:
            want, _ = ob.oracle_encode(d, segs)
        except RuntimeError as e:      # a frame the coder refuses (coefficient out of range, zero cosine term): same code from the kernel
            code = int(str(e).split()[-1])
            rcs = []
            for s in segs:
                b = C.create_string_buffer(1 << 22)
                n = C.c_uint32(0)
                rcs.append(emu.emu_encode_segment_v3(C.byref(d), s.luma_y_start, s.luma_y_end, s.is_last, b, 1 << 22, C.byref(n), None))
            first = next((r for r in rcs if r), 0)
            if first != code:
                bad += 1
                print("REFUSAL MISMATCH", seed0, k, w, h, comps, kw, "oracle", code, "kernel", rcs, flush=True)
            rcs5 = []    # ... and from the split-phase encoder (two wavefronts per segment / gather and write in parts, and one / seven)
            for s in segs:
                b = C.create_string_buffer(1 << 22)
                n, nb = C.c_uint32(0), C.c_uint32(0)
                rcs5.append(emu.emu_encode_segment_v5_parts(C.byref(d), s.luma_y_start, s.luma_y_end, s.is_last, b, 1 << 22, C.byref(n), C.byref(nb), None, 0))
            first5 = next((r for r in rcs5 if r), 0)
            if first5 != code:
                bad += 1
                print("REFUSAL MISMATCH (split-phase)", seed0, k, w, h, comps, kw, "oracle", code, "kernel", rcs5, flush=True)
            refused_both += 1
            continue
        ok = True
        for s, wv in zip(segs, want):
            cap = len(wv) + 4096
            b = C.create_string_buffer(cap)
            n = C.c_uint32(0)
            rc = emu.emu_encode_segment_v3(C.byref(d), s.luma_y_start, s.luma_y_end, s.is_last, b, cap, C.byref(n), None)
            if rc != 0 or b.raw[: n.value] != wv:
                ok = False
                print("ENCODE MISMATCH", seed0, k, w, h, comps, kw, "rc", rc, flush=True)
                break
            n5, nb5 = C.c_uint32(0), C.c_uint32(0)   # the split-phase encoder (lep_enc5.h), both launch forms, gather / write in parts
            rc = emu.emu_encode_segment_v5_parts(C.byref(d), s.luma_y_start, s.luma_y_end, s.is_last, b, cap, C.byref(n5), C.byref(nb5), None, 0)
            if rc != 0 or b.raw[: n5.value] != wv:
                ok = False
                print("ENCODE MISMATCH (split-phase)", seed0, k, w, h, comps, kw, "rc", rc, flush=True)
                break
        frames = [C.string_at(d.blocks[c], d.nblocks(c) * 128) for c in range(d.ncomp)]
        for c in range(d.ncomp):
            C.memset(d.blocks[c], 0, d.nblocks(c) * 128)
        for s, wv in zip(segs, want):
            rc = emu.emu_decode_segment_v4(C.byref(d), s.luma_y_start, s.luma_y_end, s.is_last, wv, len(wv), None)
            if rc != 0:
                ok = False
                print("DECODE RC", rc, seed0, k, w, h, comps, kw, flush=True)
        got = [C.string_at(d.blocks[c], d.nblocks(c) * 128) for c in range(d.ncomp)]
        if got != frames:
            ok = False
            print("DECODE MISMATCH", seed0, k, w, h, comps, kw, flush=True)
        ran += 1
        bad += not ok
    print(f"seed {seed0}: ran {ran} not a file {refused} refused by both {refused_both} bad {bad}")


if __name__ == "__main__":
    main()
