#!/bin/bash
# Round 2, second hardware visit: the GPU suite (adapter binary, format v2, parallel Huffman decode now in it), the counter
# calibration micro-benchmark under the EA / FETCH_SIZE counters, the new default bench line (64 distinct images, oracle parity
# for all of them, latency table, skewed / 1080p / progressive corpora), LEP_DEC4_CANDS variants of the decoder.
set -u
TAG=${1:-r02b}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -n 5 $OUT/pytest_gpu.log
echo "== calibration ($(( $(date +%s)-t0 )) s)"
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -d $OUT/calib_ea -o pmc --output-format csv -- scripts/proto/scatter_calib 24 > $OUT/calib_ea.txt 2> $OUT/calib_ea.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/calib_fetch -o pmc --output-format csv -- scripts/proto/scatter_calib 24 > $OUT/calib_fetch.txt 2> $OUT/calib_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/calib_write -o pmc --output-format csv -- scripts/proto/scatter_calib 24 > $OUT/calib_write.txt 2> $OUT/calib_write.err
python - <<PY | tee $OUT/fetch_calibration.txt
import csv, glob, collections
print(open("$OUT/calib_ea.txt").read())
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for tag in ("ea", "fetch", "write"):
    for fn in glob.glob("$OUT/calib_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].split("(")[0]
            if not k.startswith("k_"): continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(acc.items()): print(k, {a: "%.5g" % b for a, b in sorted(v.items())})
PY
echo "== bench ($(( $(date +%s)-t0 )) s)"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? ($(( $(date +%s)-t0 )) s)"; cut -c1-1500 $OUT/bench.json; tail -n 3 $OUT/bench.err
echo "== decoder candidate-count variants ($(( $(date +%s)-t0 )) s)"
for c in 2 3; do
  if [ -f lepton_amd/liblepton_cands$c.so ]; then
    LEP_LIB_PATH=$PWD/lepton_amd/liblepton_cands$c.so timeout 300 python bench.py --images 1024 --unique 8 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-extras > $OUT/bench_cands$c.json 2> $OUT/bench_cands$c.err
    python - $OUT/bench_cands$c.json $c <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("LEP_DEC4_CANDS=%s dec %s MB/s (%s ms) enc %s MB/s" % (sys.argv[2], d["decode_MBps"], d["roofline"]["decode_kernel_ms"], d["encode_MBps"]))
except Exception as e:
    print("cands %s FAILED %s" % (sys.argv[2], e))
PY
  fi
done
LEP_DUMMY=1 timeout 300 python bench.py --images 1024 --unique 8 --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --no-extras > $OUT/bench_cands4.json 2> $OUT/bench_cands4.err
python -c "
import json;d=json.loads(open('$OUT/bench_cands4.json').read().strip().splitlines()[-1]);print('LEP_DEC4_CANDS=4 (default) dec %s MB/s (%s ms) enc %s MB/s'%(d['decode_MBps'],d['roofline']['decode_kernel_ms'],d['encode_MBps']))"
echo "total $(( $(date +%s)-t0 )) s"
