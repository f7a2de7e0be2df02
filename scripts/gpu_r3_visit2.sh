#!/bin/bash
# round 3, visit 2: the split-phase encoder's first time on hardware -- parity tests, then a short bench with stage times
set -u
TAG=${1:-r04b}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "split_phase or impossible" > $OUT/pytest_v5.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -15 $OUT/pytest_v5.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0 > $OUT/bench_v5.json 2> $OUT/bench_v5.err; echo "bench rc=$? ($(( $(date +%s)-t0 )) s)"; tail -3 $OUT/bench_v5.err; python - <<PY
import json
try:
    o=json.load(open("$OUT/bench_v5.json"))
    print({k:o[k] for k in ("value","encode_MBps","decode_MBps","ms_per_step")}, o["roofline"]["encode_stages_ms"], o["roofline"]["kernels"], o["config"]["parity"])
except Exception as e: print("no json", e)
PY
echo "total $(( $(date +%s)-t0 )) s"
