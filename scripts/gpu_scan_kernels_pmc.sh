#!/bin/bash
# what the lane-per-piece scan kernels do per launch: instruction and memory-side request counters (two --pmc passes, kernel trace
# only) over the batch pipeline (1024 x 4K, one chunk each way) -> gpurun_out/<tag>/scan_kernels_pmc.json
set -u
TAG=${1:-r7d}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
B="python scripts/bench_batch.py --images 1024 --unique 16 --width 3840 --height 2160"
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/pmc_a -o pmc --output-format csv -- $B > $OUT/a.json 2> $OUT/a.err; echo "pass a rc=$? $(( $(date +%s)-t0 )) s"
timeout 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_b -o pmc --output-format csv -- $B > $OUT/b.json 2> $OUT/b.err; echo "pass b rc=$? $(( $(date +%s)-t0 )) s"
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for tag in ("pmc_a", "pmc_b"):
    for fn in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        seen = set()
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            i = k.find("(")
            k = k[:i] if i > 0 else k
            if "simt" not in k and "lep_zero" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if tag == "pmc_a" and r["Counter_Name"] == "SQ_INSTS_VALU": calls[k] += 1
out = {}
for k, c in acc.items():
    n = max(calls[k], 1)
    out[k] = {"launches": n, **{name: round(v / n) for name, v in sorted(c.items())}}
    rd, wr = c.get("TCC_EA0_RDREQ_sum", 0) / n, c.get("TCC_EA0_WRREQ_sum", 0) / n
    out[k]["hbm_GB_per_launch_64B_requests"] = round(64e-9 * (rd + wr), 2)
json.dump(out, open("$OUT/scan_kernels_pmc.json", "w"), indent=1)
for k, v in out.items(): print(k, v)
PY
rm -rf $OUT/pmc_a $OUT/pmc_b
echo "total $(( $(date +%s)-t0 )) s"
