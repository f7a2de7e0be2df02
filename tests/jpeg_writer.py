"""Test infrastructure: a coefficient-domain baseline JPEG writer (ITU T.81 sequential Huffman, 8-bit) with ARBITRARY
per-component sampling factors, component counts, restart intervals and component ids -- the layouts PIL / libjpeg's
front end cannot be asked for (4:4:0, true 4:1:1, chroma sampled finer than luma, grey with sampling 2x2, two- and
four-component frames with mixed factors).  Blocks are drawn directly as quantised coefficients (no pixels, no DCT):
the coder under test never sees anything else.  Huffman and quantisation tables are lifted from a PIL-written file
(the Annex K tables), so no table constants are typed in here.

Used by tests/test_sampling_layouts.py; never imported by the product."""
import io
import struct

import numpy as np

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _segments(jpg):
    """[(marker, payload)] of the header segments up to SOS"""
    pos, out = 2, []
    while pos < len(jpg):
        assert jpg[pos] == 0xFF
        m = jpg[pos + 1]
        n = struct.unpack(">H", jpg[pos + 2: pos + 4])[0]
        out.append((m, jpg[pos + 4: pos + 2 + n]))
        if m == 0xDA:
            break
        pos += 2 + n
    return out


_TABLES = {}


def annex_k_tables(quality=85):
    """(dqt payloads by id, dht (class, id) -> (bits[16], vals)) from a PIL-written 4:2:0 file"""
    if quality in _TABLES:
        return _TABLES[quality]
    from PIL import Image

    buf = io.BytesIO()
    Image.new("RGB", (16, 16)).save(buf, format="JPEG", quality=quality, subsampling=2, optimize=False)
    dqt, dht = {}, {}
    for m, p in _segments(buf.getvalue()):
        if m == 0xDB:
            while p:
                assert p[0] >> 4 == 0
                dqt[p[0] & 15] = bytes(p[1:65])
                p = p[65:]
        elif m == 0xC4:
            while p:
                bits = list(p[1:17])
                n = sum(bits)
                dht[(p[0] >> 4, p[0] & 15)] = (bits, list(p[17: 17 + n]))
                p = p[17 + n:]
    assert set(dqt) == {0, 1} and set(dht) == {(0, 0), (0, 1), (1, 0), (1, 1)}
    _TABLES[quality] = (dqt, dht)
    return dqt, dht


def _codes(bits, vals):
    """symbol -> (code, length), T.81 Annex C"""
    out, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(bits[ln - 1]):
            out[vals[k]] = (code, ln)
            code += 1
            k += 1
        code <<= 1
    return out


class _Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, code, ln):
        self.acc = (self.acc << ln) | (code & ((1 << ln) - 1))
        self.n += ln
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self, pad_bit=1):
        if self.n:
            k = 8 - self.n
            self.put(((1 << k) - 1) if pad_bit else 0, k)

    def marker(self, m):
        self.out += bytes([0xFF, m])


def _magnitude(v):
    a = abs(v)
    s = a.bit_length()
    return s, (v if v >= 0 else v - 1) & ((1 << s) - 1)


def random_blocks(rng, n, density=0.25, amp=40.0, dc_step=30.0, max_ac=1023):
    """n blocks x 64 zig-zag-ordered quantised coefficients: DC a bounded random walk, AC sparse, Laplacian, decaying with
    the zig-zag index (so end-of-block, short runs and 16-zero runs all occur)"""
    z = np.arange(64)
    keep = rng.random((n, 64)) < density * np.exp(-z / 14.0) * 3.0
    mag = rng.laplace(0.0, amp * np.exp(-z / 10.0) + 0.6, (n, 64))
    ac = np.clip(np.rint(mag), -max_ac, max_ac).astype(np.int32) * keep
    dc = np.clip(np.cumsum(np.rint(rng.normal(0, dc_step, n))), -1000, 1000).astype(np.int32)
    ac[:, 0] = dc
    return ac


def write_baseline(width, height, comps, rng, restart_interval=0, quality=85, density=0.25, amp=40.0, pad_bit=1, blocks=None,
                   extra_segments=(), quirks=(), sof=0xC0, precision=8, dqt16=False, scans=None):
    """comps: [(component id, h, v, quant table id, dc table id, ac table id)], one interleaved scan (one component: the
    non-interleaved geometry of T.81 A.2.2).  Returns (jpeg bytes, per-component [rows][cols] zig-zag block arrays).

    scans: None = one scan of all components; else a list of component-index lists, e.g. [[0], [1, 2]] -- a sequential
    multi-scan file (a one-component scan is non-interleaved, T.81 A.2.2: only the blocks that cover the image are coded; the
    frame needs more than one component).

    quirks: legal-to-decode but non-canonical Huffman layers, which a decode -> re-encode cannot reproduce unless it notices:
      "trailing_zrl"  every third block that ends before coefficient 47 codes ZRL + EOB instead of EOB
      "rst_fill"      an extra 0xFF fill byte in front of every restart marker (T.81 B.1.1.2)
      "mixed_pad"     pad bits alternate between all-ones and all-zeros from one restart interval to the next
      "dup_symbol"    the AC tables code symbol 0x01 twice (the unused 0xFA's code is reassigned); every other use takes the long one
      "rst_order"     restart markers count 0, 2, 4 ... instead of 0, 1, 2 ...
      "scan_tail"     two more entropy-coded bytes after the last MCU ("unneeded data found after coded image data")"""
    dqt, dht = annex_k_tables(quality)
    if "dup_symbol" in quirks:
        dht = dict(dht)
        for t in (0, 1):
            bits, vals = dht[(1, t)]
            vals = list(vals)
            vals[vals.index(0xFA)] = 0x01
            dht[(1, t)] = (bits, vals)
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    if len(comps) == 1:
        cid, h, v = comps[0][:3]
        bw = -(-(-(-width * h // hmax)) // 8)
        bh = -(-(-(-height * v // vmax)) // 8)
        dims = [(bw, bh)]
        mcux, mcuy = bw, bh
        per = [(1, 1)]
    else:
        mcux, mcuy = -(-width // (8 * hmax)), -(-height // (8 * vmax))
        dims = [(mcux * c[1], mcuy * c[2]) for c in comps]
        per = [(c[1], c[2]) for c in comps]
    if blocks is None:
        blocks = [random_blocks(rng, bw * bh, density, amp).reshape(bh, bw, 64) for bw, bh in dims]
    enc_dc = [_codes(*dht[(0, c[4])]) for c in comps]
    enc_ac = [_codes(*dht[(1, c[5])]) for c in comps]
    alt_01 = None
    if "dup_symbol" in quirks:   # _codes keeps the LAST code of a repeated symbol (the long one); find the first (short) one too
        alt_01 = []
        for c in comps:
            bits, vals = dht[(1, c[5])]
            first = dict(_codes(bits, vals))
            code, k = 0, 0
            for ln in range(1, 17):
                for _ in range(bits[ln - 1]):
                    if vals[k] == 0x01 and 0x01 in first:
                        alt_01.append((code, ln))
                        first.pop(0x01)
                    code += 1
                    k += 1
                code <<= 1
    nblk = 0
    out = bytearray(b"\xff\xd8")
    for marker, payload in extra_segments:
        out += bytes([0xFF, marker]) + struct.pack(">H", len(payload) + 2) + payload
    for tq in sorted({c[3] for c in comps}):
        if dqt16:   # Pq = 1: sixteen-bit entries (some above 255)
            out += b"\xff\xdb" + struct.pack(">H", 131) + bytes([0x10 | tq]) + b"".join(struct.pack(">H", q * (3 if i > 40 else 1)) for i, q in enumerate(dqt[tq]))
        else:
            out += b"\xff\xdb" + struct.pack(">H", 67) + bytes([tq]) + dqt[tq]
    out += bytes([0xFF, sof]) + struct.pack(">HBHHB", 8 + 3 * len(comps), precision, height, width, len(comps))
    for c in comps:
        out += bytes([c[0], (c[1] << 4) | c[2], c[3]])
    for key in sorted({(0, c[4]) for c in comps} | {(1, c[5]) for c in comps}):
        bits, vals = dht[key]
        out += b"\xff\xc4" + struct.pack(">H", 19 + len(vals)) + bytes([(key[0] << 4) | key[1]]) + bytes(bits) + bytes(vals)
    if restart_interval:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart_interval)
    out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(comps), len(comps))
    for c in comps:
        out += bytes([c[0], (c[4] << 4) | c[5]])
    out += b"\x00\x3f\x00"
    bw_ = _Bits()
    pred = [0] * len(comps)
    rst = 0
    nmcu = mcux * mcuy
    for m in range(nmcu):
        my, mx = divmod(m, mcux)
        for ci, (ph, pv) in enumerate(per):
            for by in range(pv):
                for bx in range(ph):
                    blk = blocks[ci][my * pv + by][mx * ph + bx]
                    d = int(blk[0]) - pred[ci]
                    pred[ci] = int(blk[0])
                    s, extra = _magnitude(d)
                    bw_.put(*enc_dc[ci][s])
                    if s:
                        bw_.put(extra, s)
                    run = 0
                    last = max([k for k in range(1, 64) if blk[k]], default=0)
                    for k in range(1, last + 1):
                        v = int(blk[k])
                        if not v:
                            run += 1
                            continue
                        while run > 15:
                            bw_.put(*enc_ac[ci][0xF0])
                            run -= 16
                        s, extra = _magnitude(v)
                        sym = (run << 4) | s
                        if sym == 0x01 and alt_01 and (nblk + k) & 1:
                            bw_.put(*alt_01[ci])
                        else:
                            bw_.put(*enc_ac[ci][sym])
                        bw_.put(extra, s)
                        run = 0
                    nblk += 1
                    if last < 63:
                        if "trailing_zrl" in quirks and last < 47 and nblk % 3 == 0:
                            bw_.put(*enc_ac[ci][0xF0])
                        bw_.put(*enc_ac[ci][0x00])
        if restart_interval and (m + 1) % restart_interval == 0 and m + 1 < nmcu:
            bw_.flush((pad_bit ^ (rst & 1)) if "mixed_pad" in quirks else pad_bit)
            if "rst_fill" in quirks:
                bw_.out.append(0xFF)
            bw_.marker(0xD0 + ((rst * 2 if "rst_order" in quirks else rst) & 7))
            rst += 1
            pred = [0] * len(comps)
    bw_.flush(pad_bit)
    if "scan_tail" in quirks:
        bw_.out += b"\x12\x34"
    out += bw_.out + b"\xff\xd9"
    return bytes(out), blocks


def _put_block(bw, blk, pred, dc, ac):
    """one block, sequential Huffman coding (T.81 F.1.2); returns the new DC predictor"""
    s, extra = _magnitude(int(blk[0]) - pred)
    bw.put(*dc[s])
    if s:
        bw.put(extra, s)
    run = 0
    last = max([k for k in range(1, 64) if blk[k]], default=0)
    for k in range(1, last + 1):
        v = int(blk[k])
        if not v:
            run += 1
            continue
        while run > 15:
            bw.put(*ac[0xF0])
            run -= 16
        s, extra = _magnitude(v)
        bw.put(*ac[(run << 4) | s])
        bw.put(extra, s)
        run = 0
    if last < 63:
        bw.put(*ac[0x00])
    return int(blk[0])


def write_sequential_scans(width, height, comps, rng, scans, restart_interval=0, quality=85, density=0.25, amp=40.0, restart_intervals=None):
    """A sequential (SOF0) frame coded in SEVERAL scans, e.g. scans=[[0], [1, 2]]: luma alone, then the chroma components
    interleaved -- what `jpegtran -scans` writes for non-progressive multi-scan files.  A one-component scan is
    non-interleaved (T.81 A.2.2: only the blocks covering the image are coded, MCU = one block, the restart interval counts
    blocks); the frame geometry is the interleaved one.  restart_intervals: one interval per scan, each with a DRI segment of its own in
    front of its SOS (0 = none for that scan) -- what phone cameras do in progressive files.  Returns (jpeg bytes, per-component block arrays)."""
    dqt, dht = annex_k_tables(quality)
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mcux, mcuy = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    dims = [(mcux * c[1], mcuy * c[2]) for c in comps]
    blocks = [random_blocks(rng, bw * bh, density, amp).reshape(bh, bw, 64) for bw, bh in dims]
    out = bytearray(b"\xff\xd8")
    for tq in sorted({c[3] for c in comps}):
        out += b"\xff\xdb" + struct.pack(">H", 67) + bytes([tq]) + dqt[tq]
    out += b"\xff\xc0" + struct.pack(">HBHHB", 8 + 3 * len(comps), 8, height, width, len(comps))
    for c in comps:
        out += bytes([c[0], (c[1] << 4) | c[2], c[3]])
    for key in sorted({(0, c[4]) for c in comps} | {(1, c[5]) for c in comps}):
        bits, vals = dht[key]
        out += b"\xff\xc4" + struct.pack(">H", 19 + len(vals)) + bytes([(key[0] << 4) | key[1]]) + bytes(bits) + bytes(vals)
    if restart_interval and restart_intervals is None:
        out += b"\xff\xdd" + struct.pack(">HH", 4, restart_interval)
    for si, scan in enumerate(scans):
        if restart_intervals is not None:
            restart_interval = restart_intervals[si]
            out += b"\xff\xdd" + struct.pack(">HH", 4, restart_interval)
        out += b"\xff\xda" + struct.pack(">HB", 6 + 2 * len(scan), len(scan))
        for ci in scan:
            out += bytes([comps[ci][0], (comps[ci][4] << 4) | comps[ci][5]])
        out += b"\x00\x3f\x00"
        bw = _Bits()
        pred = {ci: 0 for ci in scan}
        enc = {ci: (_codes(*dht[(0, comps[ci][4])]), _codes(*dht[(1, comps[ci][5])])) for ci in scan}
        units = []   # one entry per MCU: [(ci, row, col)]
        if len(scan) == 1:
            ci = scan[0]
            bw_c = -(-(-(-width * comps[ci][1] // hmax)) // 8)
            bh_c = -(-(-(-height * comps[ci][2] // vmax)) // 8)
            units = [[(ci, r, c)] for r in range(bh_c) for c in range(bw_c)]
        else:
            for my in range(mcuy):
                for mx in range(mcux):
                    units.append([(ci, my * comps[ci][2] + by, mx * comps[ci][1] + bx)
                                  for ci in scan for by in range(comps[ci][2]) for bx in range(comps[ci][1])])
        rst = 0
        for m, unit in enumerate(units):
            for ci, r, c in unit:
                pred[ci] = _put_block(bw, blocks[ci][r][c], pred[ci], *enc[ci])
            if restart_interval and (m + 1) % restart_interval == 0 and m + 1 < len(units):
                bw.flush(1)
                bw.marker(0xD0 + (rst & 7))
                rst += 1
                pred = {ci: 0 for ci in scan}
        bw.flush(1)
        out += bw.out
    out += b"\xff\xd9"
    return bytes(out), blocks
