#!/bin/bash
# GPU visit: Huffman decode of chunk k+1 co-resident with the arithmetic coder of chunk k (7 + 1 waves per SIMD) vs the
# back-to-back schedule (explicit 1024-image chunks), plus the serving tests.   usage: scripts/gpu_overlap.sh <tag>
set -u
TAG=${1:-overlap}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 ))s] $*"; }
timeout 600 python -m pytest tests -m gpu -x -q -k "huff or batch or serve or daemon or smoke" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log; stamp pytest
IM=${IMAGES:-2688}
timeout 300 python scripts/bench_batch.py --images $IM --unique 16 --width 3840 --height 2160 > $OUT/batch_4k_auto.json 2> $OUT/batch_auto.err; echo "auto rc=$?"
python - <<PY
import json; d=json.loads(open("$OUT/batch_4k_auto.json").read().strip().splitlines()[-1]); print("auto    ", d["compress"]["MBps_pipeline"], d["compress"]["MBps_wall"], "dec", d["decompress"]["MBps_pipeline"])
PY
stamp auto
timeout 300 python scripts/bench_batch.py --images $IM --unique 16 --width 3840 --height 2160 --chunk-images 1024 > $OUT/batch_4k_c1024.json 2> $OUT/batch_c1024.err; echo "c1024 rc=$?"
python - <<PY
import json; d=json.loads(open("$OUT/batch_4k_c1024.json").read().strip().splitlines()[-1]); print("c1024   ", d["compress"]["MBps_pipeline"], d["compress"]["MBps_wall"], "dec", d["decompress"]["MBps_pipeline"])
PY
stamp c1024
timeout 300 python scripts/bench_batch.py --images 4096 --unique 64 > $OUT/batch_1080p_auto.json 2> $OUT/batch_1080p.err; echo "1080p rc=$?"
python - <<PY
import json; d=json.loads(open("$OUT/batch_1080p_auto.json").read().strip().splitlines()[-1]); print("1080p   ", d["compress"]["MBps_pipeline"], d["compress"]["MBps_wall"], "dec", d["decompress"]["MBps_pipeline"])
PY
stamp 1080p
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_batch -o trace --output-format csv -- python scripts/bench_batch.py --images $IM --unique 16 --width 3840 --height 2160 > $OUT/batch_4k_auto_under_rocprof.json 2>> $OUT/batch_auto.err
find $OUT/prof_batch -name '*kernel_stats*' | head -1 | xargs -r head -8; find $OUT/prof_batch -name '*kernel_trace*' -size +8M -delete
stamp rocprof
