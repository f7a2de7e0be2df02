#!/bin/bash
# kernel + copy trace of the batch pipeline (2688 x 4K, three chunks) on the shipped build -> timeline table
set -u
TAG=${1:-r5a}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
IM=${IMAGES:-2688}
timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/prof -o t --output-format csv -- python scripts/bench_batch.py --images $IM --unique 64 --width 3840 --height 2160 > $OUT/batch_4k_${IM}_under_trace.json 2> $OUT/batch.err
echo "trace rc=$? ($(( $(date +%s)-t0 )) s)"; tail -2 $OUT/batch.err
python scripts/trace_timeline.py $OUT/prof 50 > $OUT/timeline_50ms.txt 2>&1
python scripts/trace_timeline.py $OUT/prof 10 > $OUT/timeline_10ms.txt 2>&1
head -c 3000 $OUT/batch_4k_${IM}_under_trace.json
for f in $(find $OUT/prof -name "*kernel_trace.csv"); do cp $f $OUT/kernel_trace.csv; done; rm -rf $OUT/prof; echo "total $(( $(date +%s)-t0 )) s"
