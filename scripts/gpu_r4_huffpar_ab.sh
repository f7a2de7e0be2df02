#!/bin/bash
# round 4: parallel scan decode for every chunk of the batch pipeline (LEP_HUFFDEC_PAR) -- A/B on one box, one with the trace
set -u
TAG=${1:-r5c}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
IM=${IMAGES:-2688}
for P in 16 32 8; do
  LEP_HUFFDEC_PAR=$P timeout 300 python scripts/bench_batch.py --images $IM --unique 64 --width 3840 --height 2160 > $OUT/batch_par$P.json 2>> $OUT/batch.err
  echo "par $P rc=$? ($(( $(date +%s)-t0 )) s)"; python -c "import json;d=json.load(open('$OUT/batch_par$P.json'));print(d['compress']['MBps_wall'], d['decompress']['MBps_wall'])"
done
LEP_HUFFDEC_PAR=16 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/prof -o t --output-format csv -- python scripts/bench_batch.py --images $IM --unique 64 --width 3840 --height 2160 > $OUT/batch_par16_under_trace.json 2>> $OUT/batch.err
python scripts/trace_timeline.py $OUT/prof 50 > $OUT/timeline_par16.txt 2>&1
rm -rf $OUT/prof; echo "total $(( $(date +%s)-t0 )) s"
