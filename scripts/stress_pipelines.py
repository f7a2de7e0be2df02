"""Stress of both batch pipelines with overlapped launches forced on (LEP_BATCH_OVERLAP=1, LEP_BATCH_DEC_OVERLAP=1): every committed fixture (all
layouts: one interleaved scan, one component, several scans, progressive, cut files, restart intervals), the reference's own images and a few 4K files,
shuffled anew every round, chunk sizes from 40 kB to 8 MB, with and without verification; every answer compared.
python scripts/stress_pipelines.py <rounds> [seed]"""
import os, random, sys
os.environ["LEP_BATCH_OVERLAP"] = "1"
os.environ["LEP_BATCH_DEC_OVERLAP"] = "1"
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import golden, golden_cases, ref_cases, ref_golden
from lepton_amd import corpus
from lepton_amd.codec import GpuCodec
pairs = [golden(n) for n in golden_cases()]
pairs += [ref_golden(n) for n in ref_cases() if n != "roundtripfail"]   # (the image the reference itself cannot restore: ROUNDTRIP_FAILURE under verify, by design)
c = GpuCodec(0)
big = [corpus.synth_jpeg(3840, 2160, 950 + i) for i in range(2)] + [corpus.synth_jpeg(1920, 1080, 960, progressive=True), corpus.synth_jpeg(2048, 1536, 961, subsampling="4:4:4")]
pairs += [(j, c.compress(j)) for j in big]
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    order = list(range(len(pairs))) * 2
    rng.shuffle(order)
    jpgs = [pairs[i][0] for i in order]; leps = [pairs[i][1] for i in order]
    cb = rng.choice([40000, 300000, 2500000, 8000000])
    verify = rng.random() < 0.5
    got, st, _ = c.compress_batch(jpgs, chunk_bytes=cb, verify=verify)
    wrong = [k for k in range(len(jpgs)) if st[k] != 0 or got[k] != leps[k]]
    back, st2, _ = c.decompress_batch(leps, chunk_bytes=cb)
    wrong2 = [k for k in range(len(jpgs)) if st2[k] != 0 or back[k] != jpgs[k]]
    if wrong or wrong2:
        bad += 1
        print("round", it, "chunk_bytes", cb, "verify", verify, "compress wrong", [(order[k], st[k]) for k in wrong[:6]], "decompress wrong", [(order[k], st2[k]) for k in wrong2[:6]])
print("rounds done, bad", bad, "files per round", 2 * len(pairs))
