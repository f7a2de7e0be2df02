#!/usr/bin/env python3
"""Decode (and encode) latency of small launches -- 1, 8, 64, 256 4K images, device-resident frames and streams, launch + kernel + sync,
best of 5 -- for the library LEP_LIB_PATH names (scripts/gpu_ab.sh -b latency-decode).  One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from lepton_amd import corpus  # noqa: E402


def main():
    dev = bench.HipDevice(0)
    uniq = corpus.make_corpus(8, 3840, 2160, 1234)
    res = dev.resident(uniq, 256, 1, 1, lambda: None, check_parity=False, with_latency=True, latency_sizes=(1, 8, 64, 256), latency_repeats=5, whole_file=False)
    print(json.dumps({"results": res["latency"], "lib": os.environ.get("LEP_LIB_PATH", "product")}))


if __name__ == "__main__":
    main()
