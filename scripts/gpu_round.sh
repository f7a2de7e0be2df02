#!/bin/bash
# One full GPU-box visit for a round: parity tests, the default bench line, rocprofv3 --kernel-trace --stats of the SAME
# command, FETCH_SIZE / WRITE_SIZE PMC passes (separate, kernel-trace only), instruction-mix counters, phase profile.
# usage: scripts/gpu_round.sh <tag> [pmc_images]      outputs under gpurun_out/<tag>/
set -u
TAG=${1:-round}; PMC_IMAGES=${2:-128}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 ))s] $*"; }

if [ -z "${SKIP_PYTEST:-}" ]; then
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log; stamp pytest
fi

for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $cnt --kernel-trace -d $OUT/pmc_$cnt -o pmc --output-format csv -- python bench.py --images $PMC_IMAGES --unique 4 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$cnt.json 2> $OUT/pmc_$cnt.err
  echo "pmc $cnt rc=$?"
done
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_insts -o pmc --output-format csv -- python bench.py --images $PMC_IMAGES --unique 4 --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_insts.json 2> $OUT/pmc_insts.err
echo "pmc insts rc=$?"
python - <<PY
import csv, glob, collections, json
res = {}
for tag in ("FETCH_SIZE", "WRITE_SIZE", "insts"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for fn in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            if "lep_" not in k: continue
            k = "lep_" + k.split("lep_")[1].split("(")[0]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r.get("Dispatch_Id"))
    for k, v in acc.items():
        res.setdefault(k, {}).update({c: x / max(1, len(calls[k])) for c, x in v.items()})
        res[k]["launches_" + tag] = len(calls[k])
json.dump({"images_per_launch": $PMC_IMAGES, "per_launch": res}, open("$OUT/pmc_summary.json", "w"), indent=1)
# bench.py's roofline.traffic table: HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950: FETCH_SIZE counts 128-byte requests as 64 B)
traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python bench.py --images $PMC_IMAGES --unique 4 --steps 1 --warmup 0 --no-cpu-baseline",
           "units": "counters are KiB; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half of the bytes fetched)",
           "images_per_launch": $PMC_IMAGES, "kernels": {}}
for k, v in res.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        b = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
        traffic["kernels"][k.split("<")[0]] = {"FETCH_SIZE_KiB": v["FETCH_SIZE"], "WRITE_SIZE_KiB": v["WRITE_SIZE"], "hbm_bytes_per_launch": b, "hbm_bytes_per_image": b / $PMC_IMAGES}
json.dump(traffic, open("$OUT/pmc_traffic.json", "w"), indent=1)
blocks = 194400.0 * $PMC_IMAGES
for k, v in res.items():
    print(k, {a.replace("SQ_", ""): round(b / blocks, 1) for a, b in sorted(v.items()) if not a.startswith("launches")}, "(per block)")
PY
stamp pmc
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json   # so that the bench line below carries roofline.traffic

timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json; stamp bench

timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace --output-format csv -- python bench.py > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
echo "rocprof rc=$?"; find $OUT/prof -name '*kernel_stats*' | head -1 | xargs -r head -6; stamp rocprof
# keep only the summaries (the per-dispatch trace is large)
find $OUT/prof -name '*kernel_trace*' -size +8M -delete


# end-to-end batch pipeline (host memory -> host memory) and the kernel times of its Huffman stages
timeout 600 python scripts/bench_batch.py --images 4096 --unique 64 > $OUT/batch_1080p_4096.json 2> $OUT/batch.err; echo "batch 1080p rc=$?"
timeout 600 python scripts/bench_batch.py --images 2048 --unique 16 --width 3840 --height 2160 > $OUT/batch_4k_2048.json 2>> $OUT/batch.err; echo "batch 4k rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_batch -o trace --output-format csv -- python scripts/bench_batch.py --images 1024 --unique 16 --width 3840 --height 2160 > $OUT/batch_4k_1024_under_rocprof.json 2>> $OUT/batch.err
find $OUT/prof_batch -name '*kernel_stats*' | head -1 | xargs -r head -8; find $OUT/prof_batch -name '*kernel_trace*' -size +8M -delete
tail -n1 $OUT/batch_1080p_4096.json | cut -c1-600; tail -n1 $OUT/batch_4k_2048.json | cut -c1-600; stamp batch
