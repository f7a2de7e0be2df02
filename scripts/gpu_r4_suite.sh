#!/bin/bash
# round 4: the whole GPU suite + the batch pipeline figures on the current build
set -u
TAG=${1:-r5j}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -3 $OUT/pytest_gpu.log
timeout 300 python scripts/bench_batch.py --images 2688 --unique 64 --width 3840 --height 2160 > $OUT/batch_4k_2688.json 2>> $OUT/batch.err
echo "4K x 2688: $(python -c "import json;d=json.load(open('$OUT/batch_4k_2688.json'));print(d['compress']['MBps_wall'], d['decompress']['MBps_wall'])") ($(( $(date +%s)-t0 )) s)"
LEP_BATCH_FIRST_CHUNK_DIV=4 timeout 300 python scripts/bench_batch.py --images 2688 --unique 64 --width 3840 --height 2160 > $OUT/batch_4k_2688_firstdiv4.json 2>> $OUT/batch.err
echo "  first chunk / 4: $(python -c "import json;d=json.load(open('$OUT/batch_4k_2688_firstdiv4.json'));print(d['compress']['MBps_wall'], d['decompress']['MBps_wall'])")"
timeout 300 python scripts/bench_batch.py --images 1024 --unique 32 --width 1920 --height 1080 > $OUT/batch_1080p_1024.json 2>> $OUT/batch.err
echo "1080p x 1024: $(python -c "import json;d=json.load(open('$OUT/batch_1080p_1024.json'));print(d['compress']['MBps_wall'], d['decompress']['MBps_wall'])")"
timeout 300 python scripts/bench_batch.py --images 256 --unique 8 --width 3840 --height 2160 --progressive > $OUT/batch_prog_256.json 2>> $OUT/batch.err
echo "progressive 4K x 256: $(python -c "import json;d=json.load(open('$OUT/batch_prog_256.json'));print(d['compress']['MBps_wall'], d['decompress']['MBps_wall'])")"
timeout 300 python scripts/bench_batch.py --images 1024 --unique 16 --width 3840 --height 2160 --verify > $OUT/batch_4k_1024_verify.json 2>> $OUT/batch.err
echo "4K x 1024 verify: $(python -c "import json;d=json.load(open('$OUT/batch_4k_1024_verify.json'));print(d['compress']['MBps_wall'], d['decompress']['MBps_wall'])")"
echo "total $(( $(date +%s)-t0 )) s"
