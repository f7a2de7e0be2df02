#!/bin/bash
# round 4: encoder change (sign chains read ahead) -- parity subset + resident bench
set -u
TAG=${1:-r5r}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "split_phase or enc5 or parity or roundtrip" > $OUT/pytest_enc.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -2 $OUT/pytest_enc.log
B="python bench.py --steps 3 --warmup 1 --unique 16 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0"
for i in 1 2; do timeout 300 $B > $OUT/b$i.json 2>> $OUT/err.txt; python -c "
import json;d=json.load(open('$OUT/b$i.json'));r=d['roofline'];print(d['value'], r['encode_kernel_ms'], r['decode_kernel_ms'], r.get('encode_stages_ms'))"; done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o b --output-format csv -- $B > $OUT/b_prof.json 2>> $OUT/err.txt
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done; rm -rf $OUT/prof
grep -i "fold\|walk\|write\|decode" $OUT/kernel_stats.csv | cut -c1-160 | head -12
echo "total $(( $(date +%s)-t0 )) s"
