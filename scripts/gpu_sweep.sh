#!/bin/bash
# occupancy sweep of the decode kernel variants.  usage: scripts/gpu_sweep.sh "<images...>" "<waves...>"
export TMPDIR=/tmp
for n in $1; do for w in $2; do
  LEP_DEC3_WAVES=$w timeout 900 python bench.py --images $n --unique 4 --steps 1 --warmup 1 --no-cpu-baseline 2>gpurun_out/sweep.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('images', $n, 'waves', $w, 'value', d['value'], 'enc', d['encode_MBps'], 'dec', d['decode_MBps'], 'dec_ms', d['roofline']['decode_kernel_ms'], d.get('bins_per_s',{}).get('decode'))
except Exception as e: print('fail', $n, $w, e)"
done; done
