#!/bin/bash
# First hardware visit of the parallel Huffman scan decoder (lep_huffdec_par.h, LEP_HUFFDEC_PAR=<n>): parity under a tight
# timeout first -- a kernel that has only run in the emulation must not be allowed to hang the box -- then single-chunk
# batches (the serving daemon's case), where the single-wave decoder is fully exposed.   usage: scripts/gpu_huffpar.sh <tag>
set -u
TAG=${1:-huffpar}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 120 python -m pytest tests/test_gpu_experimental.py -m gpu_experimental -x -q 2>&1 | tail -3 | tee $OUT/pytest.log
grep -q passed $OUT/pytest.log || { echo "parity failed or timed out: not benchmarking"; exit 1; }
for par in 0 8 16 32; do
  LEP_HUFFDEC_PAR=$par timeout 200 python scripts/bench_batch.py --images 1024 --unique 16 --width 3840 --height 2160 > $OUT/batch_4k_1024_par$par.json 2> $OUT/batch_par$par.err
  python - $OUT/batch_4k_1024_par$par.json $par <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("LEP_HUFFDEC_PAR=%s compress %s MB/s (warm), pipeline_s %s" % (sys.argv[2], d["compress"]["MBps_pipeline"], d["compress"]["pipeline_s"]))
except Exception as e:
    print("LEP_HUFFDEC_PAR=%s FAILED %s" % (sys.argv[2], e))
PY
done
