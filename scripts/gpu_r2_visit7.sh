#!/bin/bash
# Round 2, a later hardware visit: the round-trip check of progressive files made on the GPU (parity under a timeout first), the
# progressive corpus through the pipeline with and without verification, the per-phase shader-clock profile of the decoder
# (profiling build, scripts/prof_phases.py), the GPU suite.
set -u
TAG=${1:-r02j}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 300 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "progressive" > $OUT/pytest_progressive.log 2>&1; echo "progressive parity rc=$? ($(( $(date +%s)-t0 )) s)"; tail -n 12 $OUT/pytest_progressive.log
timeout 400 python scripts/bench_batch.py --images 1024 --unique 16 --width 3840 --height 2160 --progressive --verify > $OUT/batch_prog_verify_1024.json 2> $OUT/batch_prog_verify_1024.err; echo "prog verify 1024 rc=$?"; cut -c1-900 $OUT/batch_prog_verify_1024.json
timeout 400 python scripts/bench_batch.py --images 1024 --unique 16 --width 3840 --height 2160 --progressive > $OUT/batch_prog_1024.json 2> $OUT/batch_prog_1024.err; echo "prog 1024 rc=$?"; cut -c1-900 $OUT/batch_prog_1024.json
echo "== phases ($(( $(date +%s)-t0 )) s)"
if [ -f lepton_amd/liblepton_mi355x_prof.so ]; then
  timeout 300 python scripts/prof_phases.py --images 16 --replicate 64 > $OUT/dec4_phase_cycles.txt 2> $OUT/dec4_phase_cycles.err; echo "phases rc=$?"; cat $OUT/dec4_phase_cycles.txt
fi
echo "== pytest ($(( $(date +%s)-t0 )) s)"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -n 4 $OUT/pytest_gpu.log
echo "total $(( $(date +%s)-t0 )) s"
