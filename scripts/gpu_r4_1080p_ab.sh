#!/bin/bash
set -u
TAG=${1:-r5i}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
run() { local name=$1; shift; env "$@" timeout 200 python scripts/bench_batch.py --images 1024 --unique 32 --width 1920 --height 1080 > $OUT/b_$name.json 2>> $OUT/err.txt; echo "$name: $(python -c "import json;d=json.load(open('$OUT/b_$name.json'));print(d['compress']['MBps_wall'], d['decompress']['MBps_wall'], d['compress']['wall_s'])") ($(( $(date +%s)-t0 )) s)"; }
run q4 GPU_MAX_HW_QUEUES=4
run q8 GPU_MAX_HW_QUEUES=8
run q4_par0 GPU_MAX_HW_QUEUES=4 LEP_HUFFDEC_PAR=0
run q8_par0 GPU_MAX_HW_QUEUES=8 LEP_HUFFDEC_PAR=0
B="python bench.py --steps 2 --warmup 1 --unique 16 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0"
for Q in 4 8; do GPU_MAX_HW_QUEUES=$Q timeout 300 $B > $OUT/bench_q$Q.json 2>> $OUT/err.txt; python -c "
import json;d=json.load(open('$OUT/bench_q$Q.json'));r=d['roofline'];print('resident q$Q', d['value'], r['encode_kernel_ms'], r['decode_kernel_ms'], r.get('encode_stages_ms'))"; done
GPU_MAX_HW_QUEUES=8 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/prof -o t --output-format csv -- python scripts/bench_batch.py --images 1024 --unique 32 --width 1920 --height 1080 > $OUT/b_trace.json 2>> $OUT/err.txt
python scripts/trace_timeline.py $OUT/prof 20 > $OUT/timeline_1080p_q8.txt 2>&1; rm -rf $OUT/prof
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "batch or pipeline or verify or daemon or generations or 4k_roundtrip" > $OUT/pytest_batch.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_batch.log
echo "total $(( $(date +%s)-t0 )) s"
