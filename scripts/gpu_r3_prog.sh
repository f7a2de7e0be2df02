#!/bin/bash
# progressive scan decode: one pipelined launch against a launch per dependency level -- parity tests, then BASELINE configs[4] (256 x 4K) both ways
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3prog; mkdir -p $OUT
timeout 420 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "progressive" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
for v in 1 0; do
  LEP_HUFFPROG_PIPELINE=$v timeout 300 python scripts/repro_extras.py progressive 256 > $OUT/p256_$v.out 2> $OUT/p256_$v.err; echo "pipeline=$v 256 files rc=$?"; tail -1 $OUT/p256_$v.out | cut -c1-260
done
LEP_HUFFPROG_PIPELINE_MAX=100000 timeout 300 python scripts/repro_extras.py progressive 1024 > $OUT/p1024_1.out 2> $OUT/p1024_1.err; echo "pipeline (forced) 1024 files rc=$?"; tail -1 $OUT/p1024_1.out | cut -c1-260
LEP_HUFFPROG_PIPELINE=0 timeout 300 python scripts/repro_extras.py progressive 1024 > $OUT/p1024_0.out 2> $OUT/p1024_0.err; echo "level by level 1024 files rc=$?"; tail -1 $OUT/p1024_0.out | cut -c1-260
