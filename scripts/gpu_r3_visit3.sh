#!/bin/bash
# split-phase encoder iteration: parity tests, short bench with stage times, per-kernel times from rocprofv3
set -u
TAG=${1:-r04d}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "split_phase" > $OUT/pytest_v5.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -3 $OUT/pytest_v5.log
ARGS="--steps 2 --warmup 1 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o b --output-format csv -- python bench.py $ARGS > $OUT/bench_v5.json 2> $OUT/bench_v5.err; echo "bench rc=$? ($(( $(date +%s)-t0 )) s)"
python - <<PY
import json, glob, csv
try:
    l=[x for x in open("$OUT/bench_v5.json") if x.startswith("{")][0]
    o=json.loads(l)
    print({k:o[k] for k in ("value","encode_MBps","decode_MBps","ms_per_step")}, o["roofline"]["encode_stages_ms"], o["config"]["parity"][:40])
except Exception as e: print("no json", e)
for f in glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:14]: print("%-60s calls %5s avg_ms %10.3f total_ms %10.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e6, float(r["TotalDurationNs"])/1e6))
    import shutil; shutil.copy(f, "$OUT/kernel_stats.csv")
PY
rm -rf $OUT/prof
echo "total $(( $(date +%s)-t0 )) s"
