#!/bin/bash
# GPU visit: full parity suite, batch pipeline (cold + warm call) with the co-resident Huffman decode, serving under load,
# the default bench line and its rocprofv3 kernel stats.   usage: scripts/gpu_overlap2.sh <tag>
set -u
TAG=${1:-overlap2}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s); stamp() { echo "[$(( $(date +%s) - t0 ))s] $*"; }
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log; stamp pytest
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1].split("/")[-1], "compress", d["compress"]["MBps_pipeline"], "(alloc", d["compress"]["alloc_s"], ") decompress", d["decompress"]["MBps_pipeline"], "(alloc", d["decompress"]["alloc_s"], ") cold", d.get("cold_first_call"))
PY
}
timeout 300 python scripts/bench_batch.py --images 2688 --unique 16 --width 3840 --height 2160 > $OUT/batch_4k_2688.json 2> $OUT/batch.err; echo "4k rc=$?"; show $OUT/batch_4k_2688.json; stamp batch4k
timeout 300 python scripts/bench_batch.py --images 4096 --unique 64 > $OUT/batch_1080p_4096.json 2>> $OUT/batch.err; echo "1080p rc=$?"; show $OUT/batch_1080p_4096.json; stamp batch1080p
timeout 300 python scripts/bench_serve.py --requests 2048 --clients 1024 --skipverify > $OUT/serve_4k.json 2> $OUT/serve.err; echo "serve rc=$?"; tail -n1 $OUT/serve_4k.json | cut -c1-700; tail -2 $OUT/serve.err; stamp serve
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json | cut -c1-2500; stamp bench
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_batch -o trace --output-format csv -- python scripts/bench_batch.py --images 2688 --unique 16 --width 3840 --height 2160 > $OUT/batch_4k_under_rocprof.json 2>> $OUT/batch.err
find $OUT/prof_batch -name '*kernel_stats*' | head -1 | xargs -r head -8
python - <<PY
import csv, glob
for fn in glob.glob("$OUT/prof_batch/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(fn)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    with open("$OUT/batch_4k_timeline.txt", "w") as f:
        for r in rows:
            n = r["Kernel_Name"]
            if "lep_" in n:
                f.write("%-26s start %9.1f ms  end %9.1f ms  dur %8.1f ms  grid %s\n" % (n.split("lep_")[1].split("(")[0][:26], (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Grid_Size_X")))
PY
find $OUT/prof_batch -name '*kernel_trace*' -size +2M -delete
tail -16 $OUT/batch_4k_timeline.txt; stamp rocprof
