#!/bin/bash
# A/B of library builds / run-time variants on one box: scripts/gpu_ab.sh <tag> <lib>[@ENV=VAL[,ENV=VAL]]...
# (each lib = path of a liblepton_mi355x*.so variant; the optional environment selects kernel variants inside it)
set -u
TAG=$1; shift; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
for round in 1 2; do
for spec in "$@"; do
  lib=${spec%%@*}; envs=""; [ "$spec" != "$lib" ] && envs=$(echo "${spec#*@}" | tr ',' ' ')
  name=$(basename $lib .so)$(echo "$envs" | tr -d ' ' | tr '=' '_')
  env $envs LEP_LIB_PATH=$PWD/$lib timeout 300 python bench.py --steps 2 --warmup 1 --no-end-to-end --no-cpu-baseline > $OUT/${name}_$round.json 2> $OUT/${name}_$round.err
  python - $OUT/${name}_$round.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s value %8.2f  enc %8.1f ms  dec %8.1f ms" % (sys.argv[2], d["value"], d["roofline"]["encode_kernel_ms"], d["roofline"]["decode_kernel_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
done
