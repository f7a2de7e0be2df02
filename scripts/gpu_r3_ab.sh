#!/bin/bash
# A/B of environment variants of the bench's resident step, no profiler attached: "NAME=VALUE,..." per variant
set -u
TAG=${1:-ab}; shift; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
ARGS="--steps 2 --warmup 1 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0"
i=0
for v in "$@"; do
  i=$((i+1))
  env $(echo $v | tr ',' ' ') timeout 600 python bench.py $ARGS > $OUT/v$i.json 2> $OUT/v$i.err
  python - <<PY
import json
try:
    o=json.loads([x for x in open("$OUT/v$i.json") if x.startswith("{")][0])
    print("$v", {k:o[k] for k in ("value","encode_MBps","decode_MBps")}, o["roofline"]["encode_kernel_ms"], o["roofline"]["encode_stages_ms"], o["config"]["parity"][:30])
except Exception as e: print("$v", "failed", e)
PY
done
