#!/bin/bash
# A/B of overlapped coder launches (LEP_BATCH_OVERLAP) on a photograph-like corpus with unequal thread segments.
set -u
TAG=${1:-overlap3}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 200 python -m pytest tests -m gpu -x -q -k "overlapped or batch_pipeline" 2>&1 | tail -2
for ov in 0 1; do
  LEP_BATCH_OVERLAP=$ov timeout 200 python scripts/bench_batch.py --images ${IMAGES:-2016} --unique 16 --width 3840 --height 2160 --skew 2 > $OUT/batch_skew_ov$ov.json 2> $OUT/batch_skew_ov$ov.err
  python - $OUT/batch_skew_ov$ov.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("overlap", d["overlap_launches"], "jpeg_MB", d["jpeg_MB"], "compress", d["compress"]["MBps_pipeline"], "decompress", d["decompress"]["MBps_pipeline"])
except Exception as e:
    print("FAILED", e, open(sys.argv[1].replace(".json", ".err")).read()[-400:])
PY
done
