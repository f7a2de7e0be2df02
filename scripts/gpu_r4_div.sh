#!/bin/bash
# decoder parity subset + resident bench (decode kernel ms is what changed: divisions by multiplication)
set -u
TAG=${1:-r5t}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "decod or roundtrip or selftest or golden or large or garbage" > $OUT/pytest_dec.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -3 $OUT/pytest_dec.log
B="python bench.py --steps 3 --warmup 1 --unique 16 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0"
for i in 1 2; do
  timeout 300 $B > $OUT/b$i.json 2>> $OUT/err.txt; python -c "
import json;d=json.load(open('$OUT/b$i.json'));r=d['roofline'];print(d['value'], r['encode_kernel_ms'], r['decode_kernel_ms'], r.get('encode_stages_ms'))"
done
echo "total $(( $(date +%s)-t0 )) s"
