#!/bin/bash
# round 3, visit 4: the split-phase walks with two wavefronts per segment -- parity tests, then A/B against one wavefront
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3v4; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split_phase or builds_agree or selftest or self_test" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
bash scripts/gpu_r3_ab.sh r3v4/ab LEP_ENC5_WAVES=2 LEP_ENC5_WAVES=1
