#!/bin/bash
# First hardware visit of the next round (prepared at the end of round 2, when the GPU budget was spent):
#  1. the GPU suite on the tree as it stands -- the host-side parity rules added after round 2's last visit (two-row ring behind a
#     truncation point, hand-off consistency checks and field rules, refusals answered at open in the reference's order, output
#     flush timing of the baseline re-coder) have only been through the CPU suite and the lane-loop emulation;
#  2. the instruction-rate micro-benchmark with its new rows: does a vector instruction cost less under a narrowed exec mask
#     (1 / 16 / 32 lanes)?  If it does, the serial rounds of both coders belong in lanes 0..15 and nothing else matters as much;
#  3. the default bench line.
set -u
TAG=${1:-r04a}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -3 $OUT/pytest_gpu.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -o /tmp/inst_rates scripts/proto/inst_rates.hip && timeout 300 /tmp/inst_rates > $OUT/inst_rates.txt 2>&1; echo "inst_rates rc=$?"; tail -6 $OUT/inst_rates.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? ($(( $(date +%s)-t0 )) s)"; cut -c1-600 $OUT/bench.json
echo "total $(( $(date +%s)-t0 )) s"
