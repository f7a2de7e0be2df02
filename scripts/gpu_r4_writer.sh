#!/bin/bash
# round 4: the stitched writer on hardware -- encoder parity tests, then latency by launch size A/B
set -u
TAG=${1:-r5k}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "split_phase or enc5 or parity or roundtrip or batch" > $OUT/pytest_enc.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -3 $OUT/pytest_enc.log
timeout 900 python scripts/latency_writer_ab.py > $OUT/latency_writer_ab.json 2> $OUT/latency.err; echo "latency rc=$? ($(( $(date +%s)-t0 )) s)"
python - <<PY
import json
d = json.load(open("$OUT/latency_writer_ab.json"))["results"]
for k, v in d.items(): print(k, v.get("encode_ms") or v)
PY
echo "total $(( $(date +%s)-t0 )) s"
