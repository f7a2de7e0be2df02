#!/bin/bash
# One GPU-box visit: parity tests, a bench line, a rocprofv3 kernel trace.  Outputs under gpurun_out/.
# usage: scripts/gpu_check.sh <tag> [images] [extra bench args]
set -u
TAG=${1:-run}; IMAGES=${2:-256}; shift 2 || true
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
timeout 900 python bench.py --images $IMAGES --steps 2 --warmup 1 "$@" > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace --output-format csv -- python bench.py --images 64 --unique 4 --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_prof.json 2> $OUT/bench_prof.err; echo "rocprof rc=$?"
find $OUT/prof -name '*kernel_stats*' | head -3 | xargs -r head -8
