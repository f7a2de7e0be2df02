#!/bin/bash
# Round 2, first hardware visit: (1) the whole GPU suite, no -x; (2) the parallel Huffman scan decoder's first time on hardware;
# (3) lane-per-segment (SIMT) experiment with the single-lane coder at 8192 / 16384 / 24576 segments; (4) what bounds the v4
# decoder: L2 hit rate, memory-side request sizes, wave wait attribution (separate --pmc passes, kernel trace only).
set -u
TAG=${1:-r02a}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -5 $OUT/pytest_gpu.log
timeout 150 python -m pytest tests/test_gpu_experimental.py -m gpu_experimental -x -q -p no:cacheprovider > $OUT/pytest_experimental.log 2>&1; echo "experimental rc=$?"; tail -3 $OUT/pytest_experimental.log
if grep -q passed $OUT/pytest_experimental.log; then
  for par in 0 8 16 32; do
    LEP_HUFFDEC_PAR=$par timeout 240 python scripts/bench_batch.py --images 1024 --unique 16 --width 3840 --height 2160 > $OUT/batch_4k_1024_par$par.json 2> $OUT/batch_par$par.err
    python - $OUT/batch_4k_1024_par$par.json $par <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("LEP_HUFFDEC_PAR=%s compress %s MB/s (warm), pipeline_s %s wall %s" % (sys.argv[2], d["compress"]["MBps_pipeline"], d["compress"]["pipeline_s"], d["compress"].get("wall_s")))
except Exception as e:
    print("LEP_HUFFDEC_PAR=%s FAILED %s" % (sys.argv[2], e))
PY
  done
fi
echo "== SIMT experiment ($(( $(date +%s)-t0 )) s)"
for n in 1024 2048 3072; do
  LEP_ENCODE_KERNEL=5 LEP_DECODE_KERNEL=5 timeout 400 python bench.py --images $n --unique 8 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > $OUT/simt_$n.json 2> $OUT/simt_$n.err
  python - $OUT/simt_$n.json $n <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("SIMT images=%s enc %s MB/s (%s ms) dec %s MB/s (%s ms)" % (sys.argv[2], d["encode_MBps"], d["roofline"]["encode_kernel_ms"], d["decode_MBps"], d["roofline"]["decode_kernel_ms"]))
except Exception as e:
    print("SIMT images=%s FAILED %s" % (sys.argv[2], e)); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
done
echo "== PMC passes ($(( $(date +%s)-t0 )) s)"
B="python bench.py --images 1024 --unique 8 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end"
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace -d $OUT/pmc_l2 -o pmc --output-format csv -- $B > $OUT/pmc_l2.json 2> $OUT/pmc_l2.err
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -d $OUT/pmc_ea -o pmc --output-format csv -- $B > $OUT/pmc_ea.json 2> $OUT/pmc_ea.err
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --kernel-trace -d $OUT/pmc_sq -o pmc --output-format csv -- $B > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for tag in ("l2", "ea", "sq"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for fn in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if "lep_" not in k: continue
            k = "lep_" + k.split("lep_")[1].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items(): res[k].update(v)
json.dump(res, open("$OUT/pmc_bound_summary.json", "w"), indent=1)
for k, v in res.items(): print(k, {a: "%.4g" % b for a, b in sorted(v.items())})
PY
tail -3 $OUT/pmc_l2.err $OUT/pmc_ea.err $OUT/pmc_sq.err | cut -c1-300
echo "total $(( $(date +%s)-t0 )) s"
