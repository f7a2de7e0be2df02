#!/bin/bash
# workspaces as owned virtual-memory ranges (LEP_VMM=1, the default) against hipMalloc / hipFree (LEP_VMM=0): parity subset, the
# trim experiment, the resident bench
set -u
TAG=${1:-r5l}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_parity.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -3 $OUT/pytest_parity.log
for V in 1; do
  LEP_VMM=$V timeout 600 python scripts/trim_cliff.py > $OUT/trim_vmm$V.json 2> $OUT/trim_vmm$V.err; echo "trim vmm=$V rc=$? ($(( $(date +%s)-t0 )) s)"; cat $OUT/trim_vmm$V.json
done
B="python bench.py --steps 2 --warmup 1 --unique 16 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0"
timeout 1500 python bench.py --trim-between-phases --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_trim.json 2> $OUT/bench_trim.err; echo "bench with trims rc=$? ($(( $(date +%s)-t0 )) s)"
timeout 1500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_notrim.json 2> $OUT/bench_notrim.err; echo "bench without rc=$? ($(( $(date +%s)-t0 )) s)"
python - <<PY
import json
for n in ("trim", "notrim"):
    d = json.load(open("$OUT/bench_%s.json" % n))
    print(n, d["value"], {k: (v.get("compress_MBps"), v.get("decompress_MBps")) for k, v in d.get("extra", {}).items()}, d.get("end_to_end", {}).get("compress_MBps"), d.get("end_to_end", {}).get("decompress_MBps"), d.get("mixed", {}).get("value"), d.get("latency"))
PY
for V in; do LEP_VMM=$V timeout 300 $B > $OUT/bench_vmm$V.json 2>> $OUT/err.txt; python -c "
import json;d=json.load(open('$OUT/bench_vmm$V.json'));r=d['roofline'];print('resident vmm=$V', d['value'], r['encode_kernel_ms'], r['decode_kernel_ms'])"; done
echo "total $(( $(date +%s)-t0 )) s"
