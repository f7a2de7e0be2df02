#!/bin/bash
# What the first GPU visit of the next round should run (everything here was prepared without GPU minutes left):
#   1. the parity suite;  2. the parallel Huffman scan decoder's first time on hardware (parity under a timeout, then
#   single-chunk batches at 8 / 16 / 32 wavefronts per image);  3. overlapped coder launches on a photograph-like corpus;
#   4. the default bench line (now with cpu_baseline.all_cores).     usage: scripts/gpu_next_visit.sh <tag>     (~6 GPU-minutes)
set -u
TAG=${1:-next}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash scripts/gpu_huffpar.sh $TAG/huffpar
bash scripts/gpu_overlap3.sh $TAG/overlap
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench.json
