#!/bin/bash
# what bounds the v5 decoder -- instruction counts, wave time, instruction cache, for v4 / v5 one wavefront / v5 groups of four
set -u
TAG=${1:-r5e}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|INST_LEVEL|SQ_INSTS_|SQ_WAIT|SQ_ACTIVE|SQ_BUSY|SQ_WAVE|SQC_" | head -80 > $OUT/counters_avail.txt
B="python bench.py --steps 1 --warmup 0 --unique 8 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0 --images ${IMAGES:-1024}"
pass() {  # name, counters... (env in ENVV)
  local name=$1; shift
  env $ENVV timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/pmc_$name -o pmc --output-format csv -- $B > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err; echo "$name rc=$? $(( $(date +%s)-t0 )) s"
}
for V in ${VARIANTS:-v5g4 v5g1 v4}; do
  case $V in v4) ENVV="LEP_DEC_WAVES=8";; *) ENVV="LEP_LIB_PATH=$PWD/lepton_amd/liblepton_$V.so";; esac   # v4 = the product; any other name = an experiment build (scripts/build_variant.sh)
  pass ${V}_in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES
  pass ${V}_sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS
  pass ${V}_ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_IFETCH SQC_DCACHE_REQ SQC_DCACHE_MISSES
  pass ${V}_mem TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum
done
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for d in glob.glob("$OUT/pmc_*/"):
    name = d.rstrip("/").split("pmc_")[-1]
    acc = collections.defaultdict(float); dur = collections.defaultdict(float)
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if "lep_decode" not in k: continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
    var, kind = name.rsplit("_", 1)
    res[var].update(acc)
blocks = 194400.0 * ${IMAGES:-1024}
out = {v: {k.replace("SQ_", "").replace("_sum", ""): round(x / blocks, 2) for k, x in sorted(c.items())} for v, c in res.items()}
json.dump({"per_block": out, "images": ${IMAGES:-1024}}, open("$OUT/dec_pmc_per_block.json", "w"), indent=1)
for v, c in out.items(): print(v, c)
PY
rm -rf $OUT/pmc_*/
echo "total $(( $(date +%s)-t0 )) s"
