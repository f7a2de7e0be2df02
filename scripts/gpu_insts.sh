#!/bin/bash
# instruction mix of the bench kernels (per 8x8 block).  usage: scripts/gpu_insts.sh <tag> <images>
set -u
TAG=$1; IMAGES=$2
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_insts -o pmc --output-format csv -- python bench.py --images $IMAGES --unique 4 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_insts.json 2> $OUT/pmc_insts.err
timeout 600 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_FLAT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_insts2 -o pmc --output-format csv -- python bench.py --images $IMAGES --unique 4 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_insts2.json 2> $OUT/pmc_insts2.err
python - <<PY
import csv, glob, collections, json
res = {}
for tag in ("insts", "insts2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for fn in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if "lep_" not in k: continue
            k = "lep_" + k.split("lep_")[1].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        res.setdefault(k, {}).update(v)
blocks = 194400.0 * $IMAGES
out = {k: {a.replace("SQ_", ""): round(b / blocks, 1) for a, b in sorted(v.items())} for k, v in res.items()}
json.dump({"images_per_launch": $IMAGES, "per_block": out}, open("$OUT/insts_per_block.json", "w"), indent=1)
for k, v in out.items(): print(k, v)
PY
