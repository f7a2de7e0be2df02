#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3repro; mkdir -p $OUT
for seq in c1080p mixed,c1080p e2e,c1080p; do
  timeout 400 python scripts/repro_extras.py $seq > $OUT/seq_$seq.out 2> $OUT/seq_$seq.err; echo "$seq rc=$?"
  grep "^c1080p" $OUT/seq_$seq.out | python -c "
import sys, ast
for l in sys.stdin:
    d = ast.literal_eval(l[len('c1080p '):]); print('   c1080p', d['compress_MBps'], d['decompress_MBps'], d['seconds'])"
done
