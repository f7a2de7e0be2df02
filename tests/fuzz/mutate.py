#!/usr/bin/env python3
"""Seeded mutations of the golden JPEG / .lep files for tests/fuzz/host_fuzz.cc.  Structure-blind: bit flips, byte stomps,
truncations, insertions, block duplications, 16-bit length-field edits near markers.  Structure-aware (.lep only, every
third mutant): the zlib-compressed header is inflated, mutated and deflated again -- a blind mutation of it dies in inflate /
Adler-32 and never reaches the section parser (HH / CRS / FRS / EEE / GRB / PGR, the embedded JPEG header) or the re-coder --
and the flag / thread-count bytes and the hand-off section are edited by field (count > 16, several HH sections, huge
segment sizes).  usage: mutate.py <outdir> <count> <seed>"""
import glob
import os
import random
import struct
import sys
import zlib


def _brotli(encode, data):
    """format >= 2 headers are brotli streams; the system's libbrotli through ctypes (any valid stream will do for a mutant)"""
    import ctypes as C
    if encode:
        lib = C.CDLL("libbrotlienc.so.1")
        cap = C.c_size_t(len(data) + (len(data) >> 2) + 1024)
        out = C.create_string_buffer(cap.value)
        ok = lib.BrotliEncoderCompress(C.c_int(5), C.c_int(22), C.c_int(0), C.c_size_t(len(data)), data, C.byref(cap), out)
        return out.raw[:cap.value] if ok else None
    lib = C.CDLL("libbrotlidec.so.1")
    cap = C.c_size_t(max(1 << 16, len(data) * 64))
    out = C.create_string_buffer(cap.value)
    ok = lib.BrotliDecoderDecompress(C.c_size_t(len(data)), data, C.byref(cap), out)
    return out.raw[:cap.value] if ok == 1 else None


def lep_split(lep):
    """(28-byte fixed prefix, decompressed header payload, everything from "CMP" on) of a .lep (zlib header: format 1, brotli:
    formats 2 and 4), or None"""
    if len(lep) < 32 or lep[:2] != b"\xcf\x84":
        return None
    zs = struct.unpack("<I", lep[24:28])[0]
    if lep[2] == 1:
        try:
            payload = zlib.decompress(lep[28:28 + zs])
        except zlib.error:
            return None
    else:
        payload = _brotli(False, lep[28:28 + zs])
        if payload is None:
            return None
    return lep[:28], payload, lep[28 + zs:]


def lep_join(fixed, payload, rest):
    z = zlib.compress(payload, 9) if fixed[2] == 1 else _brotli(True, payload)
    out = bytearray(fixed[:24]) + struct.pack("<I", len(z)) + z + rest
    if len(out) >= 4:
        out[-4:] = struct.pack("<I", len(out))   # the size trailer (vp8_encoder.cc:602-614)
    return bytes(out)


def find_handoffs(payload):
    """offset of the "HH" section in an inflated header (behind HDR + P0D), or -1"""
    if payload[:3] != b"HDR" or len(payload) < 7:
        return -1
    pos = 7 + struct.unpack("<I", payload[3:7])[0] + 4
    return pos if payload[pos:pos + 2] == b"HH" else -1


def with_handoffs(lep, count=None, repeat=1, segment_size=None, thread_byte=None, field=None):
    """re-packs a .lep with `count` hand-off records (the last one repeated), the HH section `repeat` times, every
    segment_size overwritten, byte 4 (thread hint) replaced; field = (record, byte offset, width, value): one field of one
    record (luma_y_start u16 @0, segment_size u32 @2, overhang_byte @6, num_overhang_bits @7, last_dc[4] s16 @8)"""
    parts = lep_split(lep)
    if not parts:
        return lep
    fixed, p, rest = parts
    pos = find_handoffs(p)
    if pos < 0:
        return lep
    n = p[pos + 2]
    recs = [bytearray(p[pos + 3 + 16 * i:pos + 19 + 16 * i]) for i in range(n)]
    if count is not None and recs:
        recs = (recs + [bytearray(recs[-1]) for _ in range(max(0, count - n))])[:count]
    if segment_size is not None:
        for r in recs:
            r[2:6] = struct.pack("<I", segment_size)
    if field is not None and recs:
        i, off, width, value = field
        recs[i % len(recs)][off:off + width] = (value & ((1 << (8 * width)) - 1)).to_bytes(width, "little")
    sect = b"HH" + bytes([len(recs) & 255]) + b"".join(bytes(r) for r in recs)
    p2 = p[:pos] + sect * repeat + p[pos + 3 + 16 * n:]
    fixed = bytearray(fixed)
    if thread_byte is not None:
        fixed[4] = thread_byte
    return lep_join(bytes(fixed), p2, rest)


def mutate_lep_structured(rng, lep):
    parts = lep_split(lep)
    if not parts:
        return mutate(rng, lep)
    k = rng.randrange(8)
    if k >= 6:    # one field of one hand-off record
        off, width = rng.choice([(0, 2), (2, 4), (6, 1), (7, 1), (8, 2), (10, 2), (12, 2), (14, 2)])
        value = rng.choice([0, 1, 2, 7, 8, 9, 0xff, 0x100, 0x7fff, 0xffff, rng.randrange(1 << (8 * width)), rng.randrange(64)])
        return with_handoffs(lep, field=(rng.randrange(16), off, width, value))
    if k == 0:
        return with_handoffs(lep, count=rng.choice([0, 1, 9, 16, 17, 32, 200, 255]))
    if k == 1:
        return with_handoffs(lep, repeat=rng.choice([2, 3, 20]), count=rng.choice([None, 16, 255]))
    if k == 2:
        return with_handoffs(lep, segment_size=rng.choice([0, 1, 0x7fffffff, 0xffffffff, rng.randrange(1 << 32)]))
    if k == 3:
        return with_handoffs(lep, thread_byte=rng.choice([0, 1, 8, 9, 16, 17, 255]))
    fixed, p, rest = parts
    dht = [i for i in range(len(p) - 20) if p[i] == 0xff and p[i + 1] == 0xc4]
    if dht and rng.random() < 0.3:    # a Huffman table of the embedded JPEG header: other code counts, other / repeated symbols
        p = bytearray(p)
        i = rng.choice(dht)
        ln = (p[i + 2] << 8) | p[i + 3]
        for _ in range(rng.choice([1, 2, 4])):
            j = i + 5 + rng.randrange(16) if rng.random() < 0.5 else min(len(p) - 1, i + 4 + rng.randrange(max(ln - 2, 1)))
            p[j] = rng.choice([p[j] ^ (1 << rng.randrange(8)), (p[j] + 1) & 255, (p[j] - 1) & 255, rng.randrange(256)])
        p = bytes(p)
    else:
        p = mutate(rng, p)
    fixed = bytearray(fixed)
    if rng.random() < 0.3:
        fixed[3] = rng.choice([ord("Z"), ord("X"), ord("Y"), rng.randrange(256)])
    if rng.random() < 0.2:
        fixed[20:24] = struct.pack("<I", rng.choice([0, 1, len(p), 1 << 20, (128 << 20) + 1, rng.randrange(1 << 32)]))
    return lep_join(bytes(fixed), p, rest if rng.random() < 0.7 else mutate(rng, rest))


def mutate(rng, d):
    d = bytearray(d)
    for _ in range(rng.choice([1, 1, 2, 3, 8])):
        k = rng.randrange(9)
        n = len(d)
        if n < 4:
            break
        # headers matter most: bias positions toward the front
        pos = int(n * (rng.random() ** 3)) if rng.random() < 0.7 else rng.randrange(n)
        pos = min(pos, n - 1)
        if k == 0:
            d[pos] ^= 1 << rng.randrange(8)
        elif k == 1:
            d[pos] = rng.choice([0, 1, 0x7f, 0x80, 0xff, rng.randrange(256)])
        elif k == 2:
            del d[pos:]
        elif k == 3:
            d[pos:pos] = bytes(rng.randrange(256) for _ in range(rng.choice([1, 2, 4, 64])))
        elif k == 4:
            m = rng.choice([1, 2, 16, 256, 4096])
            d[pos:pos] = d[pos:pos + m]
        elif k == 5:
            del d[pos:pos + rng.choice([1, 2, 16, 256])]
        elif k == 6 and n > 8:   # 16-bit big-endian field after an FF xx marker
            i = d.find(b"\xff", pos)
            if 0 <= i < n - 4:
                v = rng.choice([0, 1, 2, 3, 0xffff, 0x7fff, rng.randrange(65536)])
                d[i + 2] = v >> 8
                d[i + 3] = v & 255
        elif k == 7 and n > 16:  # 32-bit little-endian field (the .lep sections carry these)
            v = rng.choice([0, 1, 0xffffffff, 0x7fffffff, 0x80000000, rng.randrange(1 << 32), rng.randrange(1 << 16)])
            d[pos:pos + 4] = v.to_bytes(4, "little")
        else:
            a, b = sorted((pos, rng.randrange(n)))
            d[a:b] = bytes(b - a) if rng.random() < 0.5 else bytes([0xff]) * (b - a)
    return bytes(d)


def main():
    out, count, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    here = os.path.dirname(os.path.abspath(__file__))
    seeds = sorted(glob.glob(os.path.join(here, "..", "golden", "*.jpg")) + glob.glob(os.path.join(here, "..", "golden", "*.lep")) +
                   glob.glob(os.path.join(here, "..", "golden", "v2", "*.lep")))
    seeds = [s for s in seeds if os.path.getsize(s) < 60000]
    blobs = [open(s, "rb").read() for s in seeds]
    rng = random.Random(seed)
    os.makedirs(out, exist_ok=True)
    for i in range(count):
        k = rng.randrange(len(blobs))
        ext = os.path.splitext(seeds[k])[1]
        m = mutate_lep_structured(rng, blobs[k]) if ext == ".lep" and i % 3 == 0 else mutate(rng, blobs[k])
        open(os.path.join(out, "m%06d%s" % (i, ext)), "wb").write(m)


if __name__ == "__main__":
    main()
