#!/bin/bash
# Round 2, sixth hardware visit: progressive scans decoded on the GPU (parity under a timeout first), the progressive corpus
# through the pipeline again, the GPU suite.
set -u
TAG=${1:-r02f}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 240 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "progressive_scans_are_decoded" > $OUT/pytest_progdec.log 2>&1; echo "progressive decode parity rc=$? ($(( $(date +%s)-t0 )) s)"; tail -n 12 $OUT/pytest_progdec.log
timeout 400 python scripts/bench_batch.py --images 256 --unique 8 --width 3840 --height 2160 --progressive > $OUT/batch_prog_gpu.json 2> $OUT/batch_prog_gpu.err; echo "prog gpu rc=$?"; cut -c1-900 $OUT/batch_prog_gpu.json
timeout 400 python scripts/bench_batch.py --images 1024 --unique 16 --width 3840 --height 2160 --progressive > $OUT/batch_prog_gpu_1024.json 2> $OUT/batch_prog_gpu_1024.err; echo "prog gpu 1024 rc=$?"; cut -c1-900 $OUT/batch_prog_gpu_1024.json
echo "== pytest ($(( $(date +%s)-t0 )) s)"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -n 4 $OUT/pytest_gpu.log
echo "total $(( $(date +%s)-t0 )) s"
