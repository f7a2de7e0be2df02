#!/bin/bash
# round 4: the v5 decoder on hardware -- parity tests with workgroups of four everywhere, then decode timing A/B in one visit
set -u
TAG=${1:-r5d}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
LEP_DEC5_GROUP_MIN=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu_parity_groups.log 2>&1; echo "pytest(groups of four) rc=$? ($(( $(date +%s)-t0 )) s)"; tail -3 $OUT/pytest_gpu_parity_groups.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_gpu_parity.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -3 $OUT/pytest_gpu_parity.log
B="python bench.py --steps 2 --warmup 1 --unique 16 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 400 $B > $OUT/bench_$name.json 2> $OUT/bench_$name.err; echo "$name rc=$? ($(( $(date +%s)-t0 )) s)"
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_$name.json"))
    r = d["roofline"]
    print("  value", d["value"], "ms/step", d["ms_per_step"], "kernels", {k: (v.get("kernel_ms"), v.get("kernel")) for k, v in r.get("per_kernel", {}).items()})
except Exception as e:
    print("  no line:", e)
PY
}
run dec5_group4 LEP_DEC5=1
run dec4 LEP_DEC5=0
run dec5_group1 LEP_DEC5=1 LEP_DEC5_GROUP=1
echo "total $(( $(date +%s)-t0 )) s"
