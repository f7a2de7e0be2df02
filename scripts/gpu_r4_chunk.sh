#!/bin/bash
# pipeline chunks of 7168 thread segments (rounds 2-3) against 8192 (the wavefronts the chip holds at once), 3072 x 4K files
set -u
TAG=${1:-r5v}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
for V in 7168 8192 7168 8192; do
  LEP_BATCH_CHUNK_SEGMENTS=$V timeout 600 python scripts/bench_batch.py --images 3072 --unique 16 --width 3840 --height 2160 > $OUT/batch_3072_$V.json 2>> $OUT/err.txt
  echo "rc=$? $V ($(( $(date +%s)-t0 )) s)"; python -c "
import json;d=json.load(open('$OUT/batch_3072_$V.json'));print({k:d[k] for k in d if 'MBps' in k or 'chunks' in k or k in ('compress_s','decompress_s')})"
done
rocm-smi --showmeminfo vram 2>/dev/null | tail -4
echo "total $(( $(date +%s)-t0 )) s"
