#!/bin/bash
# round 4: batch pipeline after the changes (parallel scan decode for every chunk, one download per chunk, 8 hardware queues) -- A/B of the
# queue count on one box, the last run under the kernel + copy trace
set -u
TAG=${1:-r5g}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
IM=${IMAGES:-2688}
for Q in 8; do
  GPU_MAX_HW_QUEUES=$Q timeout 300 python scripts/bench_batch.py --images $IM --unique 64 --width 3840 --height 2160 > $OUT/batch_q$Q.json 2>> $OUT/batch.err
  echo "queues $Q rc=$? ($(( $(date +%s)-t0 )) s)"; python -c "import json;d=json.load(open('$OUT/batch_q$Q.json'));print(d['compress']['MBps_wall'], d['decompress']['MBps_wall'])"
done
LEP_BATCH_FIRST_CHUNK_DIV=4 timeout 300 python scripts/bench_batch.py --images $IM --unique 64 --width 3840 --height 2160 > $OUT/batch_firstdiv4.json 2>> $OUT/batch.err
echo "first chunk / 4: $(python -c "import json;d=json.load(open('$OUT/batch_firstdiv4.json'));print(d['compress']['MBps_wall'], d['decompress']['MBps_wall'])")"
timeout 300 python scripts/bench_batch.py --images 1024 --unique 32 --width 1920 --height 1080 > $OUT/batch_1080p.json 2>> $OUT/batch.err
echo "1080p x 1024: $(python -c "import json;d=json.load(open('$OUT/batch_1080p.json'));print(d['compress']['MBps_wall'], d['decompress']['MBps_wall'])")"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "batch or pipeline or verify or daemon or generations or 4k_roundtrip" > $OUT/pytest_batch.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_batch.log
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/prof -o t --output-format csv -- python scripts/bench_batch.py --images $IM --unique 64 --width 3840 --height 2160 > $OUT/batch_under_trace.json 2>> $OUT/batch.err
python scripts/trace_timeline.py $OUT/prof 50 > $OUT/timeline.txt 2>&1
rm -rf $OUT/prof; echo "total $(( $(date +%s)-t0 )) s"
