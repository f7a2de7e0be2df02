#!/bin/bash
# Round 2, fourth hardware visit: the two-wavefronts-per-segment encoder (parity first, under a timeout: a barrier
# protocol that has only run in the emulation must not be allowed to hang the box), then encode latency by launch size for
# the three launch forms; the adapter binary after the exit_group fix; the GPU suite.
set -u
TAG=${1:-r02d}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 240 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "encoder_builds_agree" > $OUT/pytest_pair.log 2>&1; echo "pair parity rc=$? ($(( $(date +%s)-t0 )) s)"; tail -n 3 $OUT/pytest_pair.log
if grep -q "3 passed" $OUT/pytest_pair.log; then
  timeout 600 python scripts/latency_sweep.py > $OUT/latency_sweep.json 2> $OUT/latency_sweep.err; echo "sweep rc=$?"; cat $OUT/latency_sweep.json | head -40
fi
echo "== adapter binary ($(( $(date +%s)-t0 )) s)"
timeout 300 python -m pytest tests/test_integration_adapter.py -q -p no:cacheprovider -m gpu > $OUT/pytest_adapter.log 2>&1; echo "adapter rc=$?"; tail -n 4 $OUT/pytest_adapter.log
echo "== pytest ($(( $(date +%s)-t0 )) s)"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -n 4 $OUT/pytest_gpu.log
echo "total $(( $(date +%s)-t0 )) s"
