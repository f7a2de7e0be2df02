#!/usr/bin/env python3
"""One 4K file, host memory to host memory (GPU box): lep_compress / lep_decompress against one-file calls of the batch pipeline
(GPU scan kernels, pinned staging), with and without the round-trip check."""
import sys
import time

sys.path.insert(0, '.')
from lepton_amd import corpus
from lepton_amd.codec import GpuCodec

jpg = corpus.synth_jpeg(3840, 2160, 10000)
c = GpuCodec(0)
lep = c.compress(jpg)
ts = []
for i in range(5):
    t0 = time.perf_counter(); lep = c.compress(jpg); ts.append((time.perf_counter() - t0) * 1e3)
td = []
for i in range(3):
    t0 = time.perf_counter(); back = c.decompress(lep); td.append((time.perf_counter() - t0) * 1e3)
assert back == jpg
print("whole-file lep_compress ms", [round(t, 1) for t in ts], "lep_decompress ms", [round(t, 1) for t in td])
for verify in (True, False):
    tb = []
    for i in range(5):
        out, st, cs = c.compress_batch([jpg], verify=verify)
        assert st == [0] and out[0] == lep
        tb.append(cs["wall_s"] * 1e3)
    print("lep_compress_batch of ONE file, verify=%s: ms %s  (%s)" % (verify, [round(t, 1) for t in tb], {k: round(v, 4) for k, v in cs.items() if k.endswith("_s")}))
tb = []
for i in range(3):
    out, st, ds = c.decompress_batch([lep])
    assert st == [0] and out[0] == jpg
    tb.append(ds["wall_s"] * 1e3)
print("lep_decompress_batch of ONE file: ms %s  (%s)" % ([round(t, 1) for t in tb], {k: round(v, 4) for k, v in ds.items() if k.endswith("_s")}))
