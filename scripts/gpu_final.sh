#!/bin/bash
# Round-end evidence run (hot kernels unchanged since the PMC passes of r01j): full GPU parity suite, smoke(), the default
# bench line, rocprofv3 --kernel-trace --stats of the same bench command (without its CPU / end-to-end legs).
# usage: scripts/gpu_final.sh <tag>
set -u
TAG=${1:-final}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 ))s] $*"; }
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log; stamp pytest
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log; stamp smoke
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-1800 $OUT/bench.json; stamp bench
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace --output-format csv -- python bench.py --no-end-to-end --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
echo "rocprof rc=$?"; find $OUT/prof -name '*kernel_stats*' | head -1 | xargs -r head -6; find $OUT/prof -name '*kernel_trace*' -size +4M -delete; stamp rocprof
