#!/usr/bin/env python3
"""Generates tests/golden/*.jpg (small seeded PIL JPEGs covering the edge cases the reference's own
test-suite covers: grayscale, 4:4:4 / 4:2:2 / 4:2:0, restart intervals, odd sizes, one-block-wide
images, truncated files, trailing garbage) and, for each, the .lep the REAL reference produces
(oracle/_ref/lepton, built from /root/reference by oracle/Makefile.ref) plus, for the decode
direction, the JPEG the reference restores from that .lep (== input except for truncated inputs).
Run here (where /root/reference exists); the fixtures travel to the GPU box."""
import hashlib
import io
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lepton_amd import corpus  # noqa: E402
import jpeg_writer  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "lepton")


def pil_jpeg(w, h, seed, mode="RGB", **save):
    import numpy as np
    from PIL import Image

    base = corpus.synth_jpeg(w, h, seed, quality=95, subsampling="4:4:4")
    img = Image.open(io.BytesIO(base)).convert(mode)
    buf = io.BytesIO()
    img.save(buf, format="JPEG", **save)
    return buf.getvalue()


def layout(w, h, comps, seed, **kw):
    """frames PIL cannot write (tests/jpeg_writer.py): comps = [(id, h, v, quant table, dc table, ac table)]"""
    import numpy as np

    return jpeg_writer.write_baseline(w, h, comps, np.random.default_rng(seed), **kw)[0]



def refine_overrun_cut():
    """A hand-made grey progressive file cut inside an AC refinement scan, right behind a code that asks for the 7th still-zero position of
    a band that has five.  The reference's block decoder (jpgcoder.cc:5192-5207) notices the overrun only at the band's
    last position -- after reading a correction bit for every non-zero position on the way, which runs it into the end of the file -- so
    the file is ACCEPTED as a truncated one; a decoder that refuses at the code itself never reaches the end of the file and answers
    UNSUPPORTED_JPEG."""
    import struct

    dqt, dht = jpeg_writer.annex_k_tables(85)
    dc, ac = jpeg_writer._codes(*dht[(0, 0)]), jpeg_writer._codes(*dht[(1, 0)])
    nblk = 4
    out = bytearray(b"\xff\xd8")
    out += b"\xff\xdb" + struct.pack(">H", 67) + b"\x00" + dqt[0]
    out += b"\xff\xc2" + struct.pack(">HBHHB", 11, 8, 8, 8 * nblk, 1) + bytes([1, 0x11, 0])
    for (cls, tid) in ((0, 0), (1, 0)):
        bits, vals = dht[(cls, tid)]
        out += b"\xff\xc4" + struct.pack(">H", 19 + len(vals)) + bytes([cls << 4 | tid]) + bytes(bits) + bytes(vals)

    def sos(ss, se, ah, al):
        return b"\xff\xda" + struct.pack(">HB", 8, 1) + bytes([1, 0x00, ss, se, ah << 4 | al])

    out += sos(0, 0, 0, 0)                        # DC
    bw, pred = jpeg_writer._Bits(), 0
    for b in range(nblk):
        s, extra = jpeg_writer._magnitude(10 * b - pred)
        pred = 10 * b
        bw.put(*dc[s])
        if s:
            bw.put(extra, s)
    bw.flush()
    out += bw.out
    out += sos(1, 20, 0, 1)                       # AC band 1..20, point transform 1: positions 1..15 = +-1, 16..20 zero
    bw = jpeg_writer._Bits()
    for b in range(nblk):
        for k in range(1, 16):
            bw.put(*ac[0x01])
            bw.put((k + b) & 1, 1)
        bw.put(*ac[0x00])
    bw.flush()
    out += bw.out
    out += sos(1, 20, 1, 0)                       # its refinement: block 0 regular, block 1 the overrunning code, then nothing
    bw = jpeg_writer._Bits()
    bw.put(*ac[0x00])
    bw.put(0b101100111000101, 15)
    bw.put(*ac[0x61])
    bw.put(1, 1)
    bw.put(0b101, 3)
    bw.flush()
    return bytes(out + bw.out)


Y, CB, CR = (lambda h, v, i=1: (i, h, v, 0, 0, 0)), (lambda h, v, i=2: (i, h, v, 1, 1, 1)), (lambda h, v, i=3: (i, h, v, 1, 1, 1))

CASES = {
    "c420_160x120": lambda: corpus.synth_jpeg(160, 120, 101),
    "c420_odd_203x149": lambda: corpus.synth_jpeg(203, 149, 102, quality=85),
    "c444_96x80": lambda: pil_jpeg(96, 80, 103, quality=92, subsampling="4:4:4"),
    "c422_128x72": lambda: pil_jpeg(128, 72, 104, quality=80, subsampling="4:2:2"),
    "gray_120x88": lambda: pil_jpeg(120, 88, 105, mode="L", quality=90),
    "rst_c420_176x112": lambda: pil_jpeg(176, 112, 106, quality=88, subsampling="4:2:0", restart_marker_blocks=5),
    "rst_rows_gray_64x96": lambda: pil_jpeg(64, 96, 107, mode="L", quality=75, restart_marker_rows=1),
    "one_block_8x8": lambda: pil_jpeg(8, 8, 108, quality=90, subsampling="4:4:4"),
    "one_col_8x64": lambda: pil_jpeg(8, 64, 109, quality=90, subsampling="4:4:4"),
    "one_col_420_16x80": lambda: pil_jpeg(16, 80, 110, quality=90, subsampling="4:2:0"),
    "q100_64x64": lambda: pil_jpeg(64, 64, 111, quality=100, subsampling="4:2:0"),
    "q30_256x256_4seg": lambda: corpus.synth_jpeg(640, 480, 112, quality=97),   # > 125 kB of scan -> 2..4 segments
    "trailing_garbage": lambda: corpus.synth_jpeg(96, 96, 113) + b"tail-bytes\x00\xff\xd9more",
    "truncated": lambda: corpus.synth_jpeg(160, 160, 114)[:-1500],
    "truncated_short": lambda: corpus.synth_jpeg(128, 128, 115)[:2600],
    # progressive files (BASELINE.json configs[4]): libjpeg's 10-scan script with successive approximation
    "prog_c420_320x240": lambda: corpus.synth_jpeg(320, 240, 116, progressive=True),
    "prog_c444_203x149": lambda: corpus.synth_jpeg(203, 149, 117, progressive=True, subsampling="4:4:4", quality=75),
    "prog_gray_120x88": lambda: pil_jpeg(120, 88, 118, mode="L", quality=90, progressive=True),
    "prog_c422_rst_176x112": lambda: pil_jpeg(176, 112, 119, quality=88, subsampling="4:2:2", progressive=True, restart_marker_blocks=4),
    "prog_c420_q97_800x600": lambda: corpus.synth_jpeg(800, 600, 120, quality=97, progressive=True),   # several thread segments
    "prog_trailing_garbage": lambda: corpus.synth_jpeg(96, 96, 121, progressive=True) + b"tail\x00\xff\xd9more",
    # truncated progressive files: cut in the last refinement scan, in the middle of an AC scan, inside the first DC scan
    "prog_truncated_tail": lambda: corpus.synth_jpeg(320, 240, 131, progressive=True)[:-700],
    "prog_truncated_mid": lambda: (lambda b: b[:len(b) // 2])(corpus.synth_jpeg(320, 240, 132, progressive=True)),
    "prog_truncated_dc": lambda: corpus.synth_jpeg(320, 240, 133, progressive=True)[:1100],
    "prog_truncated_refine_overrun": refine_overrun_cut,
    "prog_truncated_q97_800x600": lambda: (lambda b: b[:len(b) * 2 // 3])(corpus.synth_jpeg(800, 600, 134, quality=97, progressive=True)),
    # sampling layouts beyond libjpeg's front end (coefficient-domain writer, tests/jpeg_writer.py)
    "lay_440_97x50": lambda: layout(97, 50, [Y(1, 2), CB(1, 1), CR(1, 1)], 201),                         # 4:4:0
    "lay_mixed_200x120": lambda: layout(200, 120, [Y(2, 2), CB(2, 1), CR(1, 1)], 202),                   # every component its own factors
    "lay_mixed_rst_104x72": lambda: layout(104, 72, [Y(2, 2), CB(1, 2), CR(2, 1)], 203, restart_interval=3),
    "lay_chromafine_96x64": lambda: layout(96, 64, [Y(1, 1), CB(2, 2), CR(1, 1)], 204),                  # chroma sampled finer than luma
    "lay_all22_64x48": lambda: layout(64, 48, [Y(2, 2), CB(2, 2), CR(2, 2)], 205),                       # four blocks of each per MCU
    "lay_two_components_120x40": lambda: layout(120, 40, [Y(2, 1), CB(1, 1)], 206),
    "lay_gray22_80x56": lambda: layout(80, 56, [Y(2, 2)], 207),                                          # one component, factors 2x2 (images/gray2sf.jpg)
    "lay_ids_pad0_64x64": lambda: layout(64, 64, [Y(2, 2, 0), CB(1, 1, 200), CR(1, 1, 7)], 208, pad_bit=0, restart_interval=4),
    "lay_440_640x480_2seg": lambda: layout(640, 480, [Y(1, 2), CB(1, 1), CR(1, 1)], 209, restart_interval=7),   # 170 kB of scan: two thread segments
}
# sequential frames coded in several scans, one of them interleaving a subset of the components ('X' files, general re-coder)
CASES.update({
    "seq_y_cbcr_420_rst_97x50": lambda: jpeg_writer.write_sequential_scans(97, 50, [Y(2, 2), CB(1, 1), CR(1, 1)], __import__("numpy").random.default_rng(211), [[0], [1, 2]], restart_interval=5)[0],
    "seq_ycb_cr_422_640x480_2seg": lambda: jpeg_writer.write_sequential_scans(640, 480, [Y(2, 1), CB(1, 1), CR(1, 1)], __import__("numpy").random.default_rng(212), [[0, 1], [2]])[0],
})
# Files the reference compresses with -skipverify but cannot restore (its default run ends in ROUNDTRIP_FAILURE, exit 41;
# test_suite/test_roundtrip.sh does the same with images/roundtripfail.jpg): the .lep is the -skipverify one, restored_md5 what
# the reference makes of it.  We must refuse the compression with 41 and decode the .lep to the same wrong bytes.
ROUNDTRIP_FAILURES = {
    "rtfail_gray22_rst_64x64": lambda: layout(64, 64, [Y(2, 2)], 210, restart_interval=7),   # one component, factors 2x2, restart markers, > 1 MCU row
}

# `lepton -startbyte=<s> -trunc=<t>` slices (format flag 'Y'): name -> (input case above, start_byte, trunc; 0 = to the end).
# The .jpg of a slice fixture is the WHOLE input file; the .lep restores bytes [start_byte, trunc) of it.
SLICES = {
    "slice_mid_q97": ("q30_256x256_4seg", 60000, 140000),      # several thread segments, starts and ends inside the scan
    "slice_tail_q97": ("q30_256x256_4seg", 150000, 0),          # to the end of the file: EOI included
    "slice_head_cut": ("c420_160x120", 700, 4000),               # start inside the header area... the first rows, cut short
    "slice_rst": ("rst_c420_176x112", 2000, 0),                  # restart markers inside the slice
    "slice_4seg_q97": ("_q97_960x720", 90000, 400000),          # 310 kB of scan: four thread segments, the first starts mid-image
}
# `lepton -embedding=<n>`: a JPEG n bytes into a larger blob (name -> (input case, prefix bytes, trailer bytes)); the .jpg of the
# fixture is the whole blob, which the .lep restores
EMBEDDED = {"embedded_c422": ("c422_128x72", 1001, 2003)}
# `lepton -permissive` on something that is not a JPEG: the bytes travel verbatim in a 'PGE' section behind a stock header
# (generic_compress.cc:60-215); only the decode direction is supported here.  name -> number of bytes
PERMISSIVE = {"permissive_5000": 5000}
SLICE_ONLY_INPUTS = {"_q97_960x720": lambda: corpus.synth_jpeg(960, 720, 140, quality=97)}


def main():
    only = set(sys.argv[1:])   # names to (re)generate; default: all
    mpath = os.path.join(HERE, "manifest.json")
    manifest = json.load(open(mpath)) if only and os.path.exists(mpath) else {}
    for name, make in list(CASES.items()) + list(ROUNDTRIP_FAILURES.items()):
        if only and name not in only:
            continue
        jpg = make()
        jp = os.path.join(HERE, name + ".jpg")
        lp = os.path.join(HERE, name + ".lep")
        open(jp, "wb").write(jpg)
        if os.path.exists(lp):
            os.unlink(lp)
        r = subprocess.run([REF, "-unjailed", "-skipverify", jp, lp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        entry = {"jpg_md5": hashlib.md5(jpg).hexdigest(), "jpg_size": len(jpg), "encode_exit": r.returncode}
        if r.returncode == 0:
            lep = open(lp, "rb").read()
            back = subprocess.run([REF, "-unjailed", lp, "/tmp/_golden_back.jpg"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            restored = open("/tmp/_golden_back.jpg", "rb").read() if back.returncode == 0 else b""
            entry.update(lep_md5=hashlib.md5(lep).hexdigest(), lep_size=len(lep), segments=lep[4],
                         restored_md5=hashlib.md5(restored).hexdigest(), restored_equals_input=restored == jpg)
            if name in ROUNDTRIP_FAILURES:
                assert restored != jpg and back.returncode == 0
                dflt = subprocess.run([REF, "-unjailed", jp, "/tmp/_golden_default.lep"], capture_output=True)
                assert b"ROUNDTRIP_FAILURE" in dflt.stderr, dflt.stderr[-200:]
                entry["roundtrip_failure"] = True
        elif os.path.exists(lp):
            os.unlink(lp)
        manifest[name] = entry
        print(name, entry)
    for name, (src, start, trunc) in SLICES.items():
        if only and name not in only:
            continue
        jpg = (CASES.get(src) or SLICE_ONLY_INPUTS[src])()
        jp = os.path.join(HERE, name + ".jpg")
        lp = os.path.join(HERE, name + ".lep")
        open(jp, "wb").write(jpg)
        if os.path.exists(lp):
            os.unlink(lp)
        flags = ["-startbyte=%d" % start] + (["-trunc=%d" % trunc] if trunc else [])
        r = subprocess.run([REF, "-unjailed", "-skipverify"] + flags + [jp, lp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        entry = {"jpg_md5": hashlib.md5(jpg).hexdigest(), "jpg_size": len(jpg), "encode_exit": r.returncode, "slice": [start, trunc]}
        if r.returncode == 0:
            lep = open(lp, "rb").read()
            back = subprocess.run([REF, "-unjailed", lp, "/tmp/_golden_back.jpg"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            restored = open("/tmp/_golden_back.jpg", "rb").read() if back.returncode == 0 else b""
            entry.update(lep_md5=hashlib.md5(lep).hexdigest(), lep_size=len(lep), segments=lep[4], flag=chr(lep[3]),
                         restored_md5=hashlib.md5(restored).hexdigest(), restored_equals_input=restored == jpg[start:(trunc or len(jpg))])
        manifest[name] = entry
        print(name, entry)
    for name, (src, npre, npost) in EMBEDDED.items():
        if only and name not in only:
            continue
        blob = bytes((i * 7 + 3) & 255 for i in range(npre)) + CASES[src]() + bytes((i * 13 + 5) & 255 for i in range(npost))
        jp = os.path.join(HERE, name + ".jpg")
        lp = os.path.join(HERE, name + ".lep")
        open(jp, "wb").write(blob)
        if os.path.exists(lp):
            os.unlink(lp)
        r = subprocess.run([REF, "-unjailed", "-skipverify", "-embedding=%d" % npre, jp, lp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        entry = {"jpg_md5": hashlib.md5(blob).hexdigest(), "jpg_size": len(blob), "encode_exit": r.returncode, "embedding": npre}
        if r.returncode == 0:
            lep = open(lp, "rb").read()
            back = subprocess.run([REF, "-unjailed", lp, "/tmp/_golden_back.jpg"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            restored = open("/tmp/_golden_back.jpg", "rb").read() if back.returncode == 0 else b""
            entry.update(lep_md5=hashlib.md5(lep).hexdigest(), lep_size=len(lep), segments=lep[4],
                         restored_md5=hashlib.md5(restored).hexdigest(), restored_equals_input=restored == blob)
        manifest[name] = entry
        print(name, entry)
    for name, nbytes in PERMISSIVE.items():
        if only and name not in only:
            continue
        blob = bytes(((i * 2654435761) >> 13) & 255 for i in range(nbytes))
        jp = os.path.join(HERE, name + ".jpg")       # (not a JPEG: the fixture's input)
        lp = os.path.join(HERE, name + ".lep")
        open(jp, "wb").write(blob)
        if os.path.exists(lp):
            os.unlink(lp)
        r = subprocess.run([REF, "-unjailed", "-permissive", jp, lp], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        entry = {"jpg_md5": hashlib.md5(blob).hexdigest(), "jpg_size": len(blob), "encode_exit": r.returncode, "permissive": True}
        if r.returncode == 0:
            lep = open(lp, "rb").read()
            back = subprocess.run([REF, "-unjailed", lp, "/tmp/_golden_back.jpg"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            restored = open("/tmp/_golden_back.jpg", "rb").read() if back.returncode == 0 else b""
            entry.update(lep_md5=hashlib.md5(lep).hexdigest(), lep_size=len(lep), segments=lep[4], restored_equals_input=restored == blob)
        manifest[name] = entry
        print(name, entry)
    json.dump(manifest, open(os.path.join(HERE, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
