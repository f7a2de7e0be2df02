"""tests/test_gpu_scan_fuzz.py with other seeds (by hand, on the GPU box): python scripts/gpu_scan_fuzz_seeds.py <seed> [<seed> ...]"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_scan_fuzz as t
for seed in map(int, sys.argv[1:]):
    drawn, large = t._drawn, t._large
    t._drawn = lambda s, n, seed=seed: drawn(seed, n)
    t._large = lambda s, seed=seed: large(seed + 1)
    try:
        t.test_gpu_scan_kernels_on_a_drawn_corpus()
        print("seed", seed, "ok")
    except AssertionError as e:
        print("seed", seed, "FAILED", str(e)[:600])
    t._drawn, t._large = drawn, large
