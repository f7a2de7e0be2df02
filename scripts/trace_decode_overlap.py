#!/usr/bin/env python3
"""A mixed 1080p / 4K corpus of two pipeline chunks through lep_decompress_batch (warm call, then the traced one): run under
`rocprofv3 --kernel-trace` to see whether the two chunks' decode kernels overlap (scripts/_ab_visit.sh prints their start / end times)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lepton_amd import corpus
from lepton_amd.codec import GpuCodec
small = [corpus.synth_jpeg(1920, 1080, 20000 + i) for i in range(8)]
big = [corpus.synth_jpeg(3840, 2160, 21000 + i) for i in range(8)]
jpgs = [(big if i & 1 else small)[(i // 2) % 8] for i in range(1536)]
c = GpuCodec(0)
leps, st, _ = c.compress_batch(jpgs)
assert not any(st)
c.decompress_batch(leps)
print("MARK", flush=True)
t0 = time.time(); back, st, ds = c.decompress_batch(leps); dt = time.time() - t0
assert back == jpgs
print("decompress %.0f MB/s wall %.3f s" % (sum(map(len, jpgs)) / 1e6 / dt, dt))
