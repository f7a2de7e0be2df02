#!/bin/bash
# round 4: progressive corpus after the refinement-scan prefetch and the ticket launch
set -u
TAG=${1:-r5o}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "progressive" > $OUT/pytest_prog.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -3 $OUT/pytest_prog.log
for N in 256 1024; do
timeout 400 python scripts/bench_batch.py --images $N --unique 8 --width 3840 --height 2160 --progressive > $OUT/batch_prog_$N.json 2>> $OUT/batch.err
echo "progressive 4K x $N: $(python -c "import json;d=json.load(open('$OUT/batch_prog_$N.json'));print(d['compress']['MBps_wall'], d['decompress']['MBps_wall'])") ($(( $(date +%s)-t0 )) s)"
done
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/prof -o t --output-format csv -- python scripts/bench_batch.py --images 256 --unique 8 --width 3840 --height 2160 --progressive > $OUT/batch_prog_256_trace.json 2>> $OUT/batch.err
python scripts/trace_timeline.py $OUT/prof 50 > $OUT/timeline_prog_256.txt 2>&1; rm -rf $OUT/prof
echo "total $(( $(date +%s)-t0 )) s"
