#!/bin/bash
# quick GPU visit: parity tests + small benches for kernel comparisons.  usage: scripts/gpu_quick.sh <tag> [images]
set -u
TAG=${1:-q}; IMAGES=${2:-64}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for dk in ${DECS:-3 2}; do
  LEP_DECODE_KERNEL=$dk timeout 600 python bench.py --images $IMAGES --unique 4 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_dec$dk.json 2> $OUT/bench_dec$dk.err; echo "dec kernel $dk rc=$?"
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_dec$dk.json"))
    print("dec$dk", "value", d["value"], "enc", d["encode_MBps"], "dec", d["decode_MBps"], "enc_ms", d["roofline"]["encode_kernel_ms"], "dec_ms", d["roofline"]["decode_kernel_ms"])
except Exception as e:
    print("no json", e); print(open("$OUT/bench_dec$dk.err").read()[-2000:])
PY
done
