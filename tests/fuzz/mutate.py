#!/usr/bin/env python3
"""Seeded structure-blind mutations of the golden JPEG / .lep files for tests/fuzz/host_fuzz.cc: bit flips, byte stomps,
truncations, insertions, block duplications, 16-bit length-field edits near markers.  usage: mutate.py <outdir> <count> <seed>"""
import glob
import os
import random
import sys


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
    seeds = sorted(glob.glob(os.path.join(here, "..", "golden", "*.jpg")) + glob.glob(os.path.join(here, "..", "golden", "*.lep")))
    seeds = [s for s in seeds if os.path.getsize(s) < 60000]
    blobs = [open(s, "rb").read() for s in seeds]
    rng = random.Random(seed)
    os.makedirs(out, exist_ok=True)
    for i in range(count):
        k = rng.randrange(len(blobs))
        ext = os.path.splitext(seeds[k])[1]
        open(os.path.join(out, "m%06d%s" % (i, ext)), "wb").write(mutate(rng, blobs[k]))


if __name__ == "__main__":
    main()
