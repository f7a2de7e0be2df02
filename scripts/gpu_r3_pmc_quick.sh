#!/bin/bash
# two quick PMC passes over the coder kernels (requests; wave time) -- for an A/B of one kernel
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3pq; mkdir -p $OUT
B="python bench.py --steps 1 --warmup 0 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0"
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_mem -o pmc --output-format csv -- $B > $OUT/m.json 2> $OUT/m.err; echo "mem rc=$?"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $OUT/pmc_sq -o pmc --output-format csv -- $B > $OUT/s.json 2> $OUT/s.err; echo "sq rc=$?"
python - <<'PY'
import csv, glob, collections, re
res = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in glob.glob("gpurun_out/r3pq/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        m = re.search(r"lep_\w+(<[\d, ]+>)?", r["Kernel_Name"])
        if m: res[m.group(0)][r["Counter_Name"]] += float(r["Counter_Value"])
NB = 1024 * 194400
for k, v in res.items():
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    print("%-34s rd/blk %6.1f wr/blk %6.1f L2hit %.2f | wait %.2f stall %.2f issue %.2f | valu/blk %7.1f salu/blk %7.1f lds/blk %6.1f" % (
        k[:34], v.get("TCC_EA0_RDREQ_sum", 0) / NB, v.get("TCC_EA0_WRREQ_sum", 0) / NB, v.get("TCC_HIT_sum", 0) / ((v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) or 1),
        v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_ACTIVE_INST_ANY", 0) / wc, v.get("SQ_INSTS_VALU", 0) / NB, v.get("SQ_INSTS_SALU", 0) / NB, v.get("SQ_INSTS_LDS", 0) / NB))
PY
rm -rf $OUT/pmc_mem $OUT/pmc_sq
