#!/bin/bash
# Round 2, third hardware visit: the decoder's group-interleaved model layout against round 1's row-major one (A/B in one
# run + memory-side request counters of both), the GPU suite (adapter binary after the exit-order fix).
set -u
TAG=${1:-r02c}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
B="python bench.py --images 1024 --unique 8 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-extras"
for v in rowmajor mi355x rowmajor mi355x; do
  LEP_LIB_PATH=$PWD/lepton_amd/liblepton_$v.so timeout 300 $B > $OUT/ab_$v.json 2> $OUT/ab_$v.err
  python - $OUT/ab_$v.json $v <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-9s dec %s MB/s (%s ms) enc %s MB/s (%s ms)" % (sys.argv[2], d["decode_MBps"], d["roofline"]["decode_kernel_ms"], d["encode_MBps"], d["roofline"]["encode_kernel_ms"]))
except Exception as e:
    print(sys.argv[2], "FAILED", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-800:])
PY
done
echo "== counters ($(( $(date +%s)-t0 )) s)"
for v in rowmajor mi355x; do
  LEP_LIB_PATH=$PWD/lepton_amd/liblepton_$v.so timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum --kernel-trace -d $OUT/pmc_$v -o pmc --output-format csv -- python bench.py --images 1024 --unique 8 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-extras > $OUT/pmc_$v.json 2> $OUT/pmc_$v.err
  LEP_LIB_PATH=$PWD/lepton_amd/liblepton_$v.so timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $OUT/pmc2_$v -o pmc --output-format csv -- python bench.py --images 1024 --unique 8 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-extras > $OUT/pmc2_$v.json 2> $OUT/pmc2_$v.err
done
python - <<PY | tee $OUT/layout_ab_counters.txt
import csv, glob, collections, json
res = {}
for v in ("rowmajor", "mi355x"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for tag in ("pmc", "pmc2"):
        for fn in glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (tag, v), recursive=True):
            for r in csv.DictReader(open(fn)):
                k = r["Kernel_Name"]
                if "lep_" not in k: continue
                k = "lep_" + k.split("lep_")[1].split("(")[0]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    res[v] = {k: dict(x) for k, x in acc.items()}
    for k, x in acc.items(): print(v, k, {a: "%.4g" % b for a, b in sorted(x.items())})
json.dump(res, open("$OUT/layout_ab_counters.json", "w"), indent=1)
PY
echo "== pytest ($(( $(date +%s)-t0 )) s)"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -n 4 $OUT/pytest_gpu.log
echo "total $(( $(date +%s)-t0 )) s"
