#!/bin/bash
# One full GPU-box visit for a round: parity tests, the default bench line, rocprofv3 --kernel-trace --stats of the bench's
# timed launches, the batch pipeline.
# usage: scripts/gpu_round.sh <tag>      outputs under gpurun_out/<tag>/
set -u
TAG=${1:-round}; PMC_IMAGES=${2:-128}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 ))s] $*"; }

if [ -z "${SKIP_PYTEST:-}" ]; then
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log; stamp pytest
fi

# (the PMC passes that make profiles/pmc_traffic.json live in scripts/gpu_r2_visit1.sh / visit3.sh: memory-side request counters,
# calibrated; this script no longer touches that file)

timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; stamp bench

# the same command without the secondary corpora and the latency table (their launches have other sizes and would blur the
# per-kernel average): the line printed under the profiler is kept beside the stats
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace --output-format csv -- python bench.py --no-extras --no-end-to-end --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
echo "rocprof rc=$?"; find $OUT/prof -name '*kernel_stats*' | head -1 | xargs -r head -6; stamp rocprof
# keep only the summaries (the per-dispatch trace is large)
find $OUT/prof -name '*kernel_trace*' -size +8M -delete


# end-to-end batch pipeline (host memory -> host memory) and the kernel times of its Huffman stages
timeout 600 python scripts/bench_batch.py --images 4096 --unique 64 > $OUT/batch_1080p_4096.json 2> $OUT/batch.err; echo "batch 1080p rc=$?"
timeout 600 python scripts/bench_batch.py --images 2048 --unique 16 --width 3840 --height 2160 > $OUT/batch_4k_2048.json 2>> $OUT/batch.err; echo "batch 4k rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_batch -o trace --output-format csv -- python scripts/bench_batch.py --images 1024 --unique 16 --width 3840 --height 2160 > $OUT/batch_4k_1024_under_rocprof.json 2>> $OUT/batch.err
find $OUT/prof_batch -name '*kernel_stats*' | head -1 | xargs -r head -8; find $OUT/prof_batch -name '*kernel_trace*' -size +8M -delete
tail -n1 $OUT/batch_1080p_4096.json | cut -c1-600; tail -n1 $OUT/batch_4k_2048.json | cut -c1-600; stamp batch
