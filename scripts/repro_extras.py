#!/usr/bin/env python3
"""bench.py's secondary corpora through the host-to-host pipeline, one after the other in ONE process (for a debugger, or to see
what a phase leaves behind for the next): comma-separated list of skewed | c1080p | progressive | refbench | mixed | e2e; an
optional file count applies to every phase"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import faulthandler

faulthandler.enable()
import bench
from lepton_amd import corpus

keys = sys.argv[1].split(",")
n_arg = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = bench.HipDevice(0)
for key in keys:
    n = n_arg
    if key == "skewed":
        u = corpus.make_corpus(16, 3840, 2160, 30001, skew=2.0); n = n or 1024
    elif key == "c1080p":
        u = corpus.make_corpus(32, 1920, 1080, 31001); n = n or 1024
    elif key == "progressive":
        u = corpus.make_corpus(8, 3840, 2160, 32001, progressive=True); n = n or 256
    elif key == "mixed":
        u = bench.mixed_corpus(32); n = n or 1024
    elif key == "e2e":
        u = corpus.make_corpus(64, 3840, 2160, 20001); n = n or 2688
    else:
        u = [bench.reference_benchmark_jpeg()]; n = n or 512
    fig = dev.pipeline([u[i % len(u)] for i in range(n)], key)
    fig.pop("_cs"); fig.pop("_ds")
    print(key, fig, flush=True)
