#!/bin/bash
# dynamic instruction counts of the coder kernels per 8x8 block for experiment builds: scripts/gpu_insts_ab.sh <tag> <lib> [<lib> ...]
# (separate rocprofv3 --pmc pass per build, --kernel-trace only)
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
for lib in "$@"; do
  name=$(basename $lib .so)
  LEP_LIB_PATH=$PWD/$lib timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_$name -o pmc --output-format csv -- python bench.py --images 128 --unique 4 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-extras > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  python - $OUT/pmc_$name $name <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(dict)
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k = (int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]))
        acc[k][r["Counter_Name"]] = acc[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
seen = set()
for (d, kn, grid), v in sorted(acc.items()):
    if "lep_decode_v4" not in kn and "lep_encode_v3" not in kn: continue
    short = "lep_" + kn.split("lep_")[1].split("(")[0]
    if (short, grid) in seen: continue
    seen.add((short, grid))
    blocks = 194400.0 * (grid / 64 / 8)      # 8 segments per 4K image
    if "x2" in short: blocks /= 2
    print("%-12s %-26s grid %7d: per block VALU %7.1f SALU %7.1f LDS %6.1f VMEM %5.1f SMEM %5.1f" % (sys.argv[2], short, grid, v.get("SQ_INSTS_VALU", 0) / blocks,
          v.get("SQ_INSTS_SALU", 0) / blocks, v.get("SQ_INSTS_LDS", 0) / blocks, (v.get("SQ_INSTS_VMEM_RD", 0) + v.get("SQ_INSTS_VMEM_WR", 0)) / blocks, v.get("SQ_INSTS_SMEM", 0) / blocks))
PY
done
