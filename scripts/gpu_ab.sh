#!/bin/bash
# One GPU visit: an optional parity subset, then the same measurement under several variants, back to back on one box.
#
#   scripts/gpu_ab.sh <tag> [-k "<pytest -k expression>"] [-b resident|batch4k|batch1080p|progressive|latency|latency-decode] [-e "<extra args of the bench>"] -- "<variant>" ...
#
# A variant is a string of environment assignments ("" = the product as built): knobs of the library (LEP_DEC_WAVES=4,
# LEP_ENC5_WCHUNKS=0, LEP_VMM_CHUNK_MB=64, LEP_BATCH_CHUNK_SEGMENTS=7168, GPU_MAX_HW_QUEUES=4, ...) or an experiment build of it
# (LEP_LIB_PATH=$PWD/lepton_amd/liblepton_<name>.so from scripts/build_variant.sh <name> -D...).  Results: gpurun_out/<tag>/.
# Round 4's A/B runs in profiles/r05* were taken this way (the decoder forms, the stitched writer, the workspace chunk size, the
# pipeline chunk size, the issue-port sensitivity builds, the divisions, the sign chains).
set -u
TAG=$1; shift
K=""; BENCH=resident; EXTRA=""
while [ $# -gt 0 ] && [ "$1" != "--" ]; do
  case $1 in -k) K=$2; shift 2;; -b) BENCH=$2; shift 2;; -e) EXTRA=$2; shift 2;; *) echo "unknown option $1"; exit 2;; esac
done
[ $# -gt 0 ] && shift
[ $# -eq 0 ] && set -- ""
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
if [ -n "$K" ]; then
  timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "$K" > $OUT/pytest.log 2>&1; echo "pytest rc=$? ($(( $(date +%s)-t0 )) s)"; tail -2 $OUT/pytest.log
fi
case $BENCH in
  resident)    CMD="python bench.py --steps 3 --warmup 1 --unique 16 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0 $EXTRA";;
  batch4k)     CMD="python scripts/bench_batch.py --images 2688 --unique 16 --width 3840 --height 2160 $EXTRA";;
  batch1080p)  CMD="python scripts/bench_batch.py --images 1024 --unique 32 --width 1920 --height 1080 $EXTRA";;
  progressive) CMD="python scripts/bench_batch.py --images 256 --unique 8 --width 3840 --height 2160 --progressive $EXTRA";;
  latency)     CMD="python scripts/latency_writer_ab.py $EXTRA";;
  latency-decode) CMD="python scripts/latency_decode_ab.py $EXTRA";;
  *) echo "unknown bench $BENCH"; exit 2;;
esac
i=0
for V in "$@"; do
  i=$((i+1)); f=$OUT/${BENCH}_$i.json
  env $V timeout 600 $CMD > $f 2>> $OUT/err.txt; rc=$?
  python - "$f" "$V" "$rc" <<'PY' | tee -a $OUT/ab.txt
import json, sys
f, v, rc = sys.argv[1:4]
try:
    d = json.load(open(f))
except Exception as e:
    print("[%s] rc=%s no line: %s" % (v, rc, e)); raise SystemExit
if "roofline" in d:
    r = d["roofline"]
    print("[%s] value %s MB/s, encode %s ms, decode %s ms, stages %s" % (v, d["value"], r.get("encode_kernel_ms"), r.get("decode_kernel_ms"), r.get("encode_stages_ms")))
elif "compress" in d:
    print("[%s] compress %s MB/s, decompress %s MB/s" % (v, d["compress"]["MBps_wall"], d["decompress"]["MBps_wall"]))
else:
    print("[%s] %s" % (v, json.dumps(d.get("results", d))[:600]))
PY
done
echo "total $(( $(date +%s)-t0 )) s"
