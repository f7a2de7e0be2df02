#!/bin/bash
# Round 2, closing measurements of the shipped kernels: memory-side request counters / L2 hit rate / wave-time split / instruction
# counts (two --pmc passes), latency by launch size, the progressive corpus through the pipeline with and without verification.
set -u
TAG=${1:-r02z}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
bash scripts/gpu_counters.sh $TAG lepton_amd/liblepton_mi355x.so; echo "counters ($(( $(date +%s)-t0 )) s)"
timeout 280 python scripts/latency_sweep.py > $OUT/latency_sweep.json 2> $OUT/latency_sweep.err; echo "latency rc=$? ($(( $(date +%s)-t0 )) s)"
timeout 400 python scripts/bench_batch.py --images 1024 --unique 16 --width 3840 --height 2160 --progressive > $OUT/batch_prog_1024.json 2> $OUT/batch_prog_1024.err; echo "prog 1024 rc=$?"; cut -c1-700 $OUT/batch_prog_1024.json
timeout 400 python scripts/bench_batch.py --images 1024 --unique 16 --width 3840 --height 2160 --progressive --verify > $OUT/batch_prog_verify_1024.json 2> $OUT/batch_prog_verify_1024.err; echo "prog verify 1024 rc=$?"; cut -c1-500 $OUT/batch_prog_verify_1024.json
timeout 400 python scripts/bench_batch.py --images 1024 --unique 16 --width 3840 --height 2160 --verify > $OUT/batch_4k_verify_1024.json 2> $OUT/batch_4k_verify_1024.err; echo "4k verify 1024 rc=$?"; cut -c1-500 $OUT/batch_4k_verify_1024.json
echo "total $(( $(date +%s)-t0 )) s"
