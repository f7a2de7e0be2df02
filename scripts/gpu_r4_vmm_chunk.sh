#!/bin/bash
set -u
TAG=${1:-r5q}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
B="python bench.py --steps 3 --warmup 1 --unique 16 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0"
for V in "LEP_VMM=0" "LEP_VMM_CHUNK_MB=64" "LEP_VMM_CHUNK_MB=512" "LEP_VMM_CHUNK_MB=2048" "LEP_VMM=0" "LEP_VMM_CHUNK_MB=64"; do
  env $V timeout 300 $B > $OUT/b.json 2>> $OUT/err.txt; python -c "
import json;d=json.load(open('$OUT/b.json'));r=d['roofline'];print('$V', d['value'], r['encode_kernel_ms'], r['decode_kernel_ms'], r.get('encode_stages_ms'))"
done
echo "total $(( $(date +%s)-t0 )) s"
