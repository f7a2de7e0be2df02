#!/bin/bash
# instruction-mix counters for the bench kernels.  usage: scripts/gpu_pmc.sh <tag> <images> "<counters>"
set -u
TAG=$1; IMAGES=$2; CNT=$3
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 800 rocprofv3 --pmc $CNT --kernel-trace -d $OUT/pmc -o pmc --output-format csv -- python bench.py --images $IMAGES --unique 4 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
echo "rc=$?"
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "lep_" not in k: continue
        k = k.split("lep_")[1].split("(")[0][:24]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
blocks = 194400.0 * $IMAGES
for k, v in acc.items():
    print(k, "per block:", {a.replace("SQ_", ""): "%.0f" % (b / blocks) for a, b in sorted(v.items())})
PY
