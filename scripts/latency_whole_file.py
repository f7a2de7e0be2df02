import sys, time
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
print("whole-file compress ms", [round(t, 1) for t in ts], "decompress ms", [round(t, 1) for t in td])
