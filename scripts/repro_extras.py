#!/usr/bin/env python3
"""one of bench.py's secondary corpora through the host-to-host pipeline, alone (for a debugger): skewed | c1080p | progressive | refbench"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import faulthandler

faulthandler.enable()
import bench
from lepton_amd import corpus

key = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = bench.HipDevice(0)
if key == "skewed":
    u = corpus.make_corpus(16, 3840, 2160, 30001, skew=2.0); n = n or 1024
elif key == "c1080p":
    u = corpus.make_corpus(32, 1920, 1080, 31001); n = n or 1024
elif key == "progressive":
    u = corpus.make_corpus(8, 3840, 2160, 32001, progressive=True); n = n or 256
else:
    u = [bench.reference_benchmark_jpeg()]; n = n or 512
fig = dev.pipeline([u[i % len(u)] for i in range(n)], key)
fig.pop("_cs"); fig.pop("_ds")
print(key, fig)
