#!/bin/bash
# the fold stage taken apart: every kind of chain as a launch of its own, under the kernel trace (durations per launch)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3fold; mkdir -p $OUT
ARGS="--steps 1 --warmup 1 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0"
LEP_ENC5_FOLD_APART=1 timeout 900 rocprofv3 --kernel-trace -d $OUT/trace -o fold --output-format csv -- python bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r3fold/trace/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
out = []
for r in rows:
    n = r["Kernel_Name"]
    if "enc5" not in n: continue
    out.append((n.split("(")[0].replace("(anonymous namespace)::", ""), int(r["Grid_Size"]) // 64 if "Grid_Size" in r else 0, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
for o in out[-16:]: print("%-40s grid %8d  %8.2f ms" % o)
open("gpurun_out/r3fold/summary.txt", "w").write("\n".join("%-40s grid %8d  %8.2f ms" % o for o in out))
PY
