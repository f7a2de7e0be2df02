#!/bin/bash
# A/B of experiment builds of the library on one box: scripts/gpu_ab.sh <tag> <lib> [<lib> ...]   (paths relative to the repo root;
# LEP_LIB_PATH selects the build, lepton_amd/abi.py).  Each build runs the bench's hot path (1024 x 4K, 8 distinct, oracle parity
# for all of them) twice, interleaved, so that drift of the box shows up as a difference between the two passes.
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
for pass in 1 2; do
  for lib in "$@"; do
    name=$(basename $lib .so)
    LEP_LIB_PATH=$PWD/$lib timeout 400 python bench.py --images 1024 --unique 8 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-extras > $OUT/ab_${name}_$pass.json 2> $OUT/ab_${name}_$pass.err
    python - $OUT/ab_${name}_$pass.json $name $pass <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-28s pass %s: value %7.1f MB/s  encode %7.1f ms  decode %7.1f ms" % (sys.argv[2], sys.argv[3], d["value"], r["encode_kernel_ms"], r["decode_kernel_ms"]))
except Exception as e:
    print("%s pass %s FAILED: %s" % (sys.argv[2], sys.argv[3], e))
PY
  done
done
