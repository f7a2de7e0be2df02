#!/usr/bin/env python3
"""What giving the cached workspaces back costs (`lep_gpu_trim` between phases halved the 1080p figure): a big
phase (1024 x 4K through the pipeline: ~140 GB of encoder scratch and 24 GB of models come and go), then the 1080p figure three times
-- as is, after a trim, after another trim.  Run once with LEP_VMM=0 (hipMalloc / hipFree workspaces) and once with LEP_VMM=1."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def fig(codec, jpgs):
    mb = sum(map(len, jpgs)) / 1e6
    leps, st, cs = codec.compress_batch(jpgs)
    back, st2, ds = codec.decompress_batch(leps)
    assert not any(st) and not any(st2) and back == jpgs
    return round(mb / cs["wall_s"], 1), round(mb / ds["wall_s"], 1)


def main():
    from lepton_amd import abi, corpus
    from lepton_amd.codec import GpuCodec

    codec = GpuCodec(0)
    L = abi.lib()
    big = corpus.make_corpus(16, 3840, 2160, 30001)
    small = corpus.make_corpus(32, 1920, 1080, 31001)
    big = [big[i % 16] for i in range(1024)]
    small = [small[i % 32] for i in range(1024)]
    out = {"LEP_VMM": os.environ.get("LEP_VMM", "(default)")}
    fig(codec, small)                                   # warm the staging
    out["1080p_fresh"] = fig(codec, small)
    out["4k_first"] = fig(codec, big)
    out["4k_warm"] = fig(codec, big)
    out["1080p_after_4k"] = fig(codec, small)
    t0 = time.perf_counter(); L.lep_gpu_trim(codec.handle); out["trim_s"] = round(time.perf_counter() - t0, 3)
    out["1080p_after_trim"] = fig(codec, small)
    out["1080p_after_trim_again"] = fig(codec, small)
    L.lep_gpu_trim(codec.handle)
    out["4k_after_trim"] = fig(codec, big)
    L.lep_gpu_trim(codec.handle)
    out["1080p_after_second_trim"] = fig(codec, small)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
