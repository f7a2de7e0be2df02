#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3parts; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "split_phase or builds_agree" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
bash scripts/gpu_r3_ab.sh r3parts/ab LEP_ENC5_PARTS=4 LEP_ENC5_PARTS=1 LEP_ENC5_PARTS=8
