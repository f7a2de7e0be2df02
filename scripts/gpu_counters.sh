#!/bin/bash
# (four counters per pass is what the hardware takes; IMAGES=<n> sizes the launch, ONLY_EA=1 skips the second pass)
# memory-side request counters + L2 hit rate + wave-time split of the coder kernels for experiment builds (two separate rocprofv3
# --pmc passes per build, --kernel-trace only): scripts/gpu_counters.sh <tag> <lib> [<lib> ...] -> gpurun_out/<tag>/counters.json
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
IMAGES=${IMAGES:-1024}
B="python bench.py --images $IMAGES --unique 8 --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end --no-extras"
for lib in "$@"; do
  v=$(basename $lib .so)
  LEP_LIB_PATH=$PWD/$lib timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum --kernel-trace -d $OUT/pmc_$v -o pmc --output-format csv -- $B > $OUT/pmc_$v.json 2> $OUT/pmc_$v.err
  [ -n "${ONLY_EA:-}" ] || LEP_LIB_PATH=$PWD/$lib timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace -d $OUT/pmc2_$v -o pmc --output-format csv -- $B > $OUT/pmc2_$v.json 2> $OUT/pmc2_$v.err
done
python - "$@" <<PY | tee $OUT/counters.txt
import csv, glob, collections, json, sys, os
res = {}
images = int(os.environ.get('IMAGES', '1024'))
blocks = 194400.0 * images
for lib in sys.argv[1:]:
    v = os.path.basename(lib)[:-3]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for tag in ("pmc", "pmc2"):
        for fn in glob.glob("$OUT/%s_%s/**/*counter_collection.csv" % (tag, v), recursive=True):
            for r in csv.DictReader(open(fn)):
                k = r["Kernel_Name"]
                if "lep_decode_v4" not in k and "lep_encode_v3" not in k: continue
                if int(r["Grid_Size"]) != images * 8 * 64: continue      # the full-size launches only
                k = "lep_" + k.split("lep_")[1].split("(")[0].split("<")[0]
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    res[v] = {}
    for k, x in acc.items():
        per = {c: val / max(1, len(n[(k, c)])) for c, val in x.items()}
        rd = per.get("TCC_EA0_RDREQ_sum", 0) * 64.0
        wr = per.get("TCC_EA0_WRREQ_64B_sum", 0) * 64.0 + (per.get("TCC_EA0_WRREQ_sum", 0) - per.get("TCC_EA0_WRREQ_64B_sum", 0)) * 32.0
        hit, miss = per.get("TCC_HIT_sum", 0), per.get("TCC_MISS_sum", 0)
        wc = per.get("SQ_WAVE_CYCLES", 1)
        res[v][k] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr, "hbm_bytes_per_image": (rd + wr) / images,
                     "read_sectors_per_block": per.get("TCC_EA0_RDREQ_sum", 0) / blocks, "l2_requests_per_block": per.get("TCC_REQ_sum", 0) / blocks,
                     "l2_hit_rate": hit / max(1.0, hit + miss), "valu_per_block": per.get("SQ_INSTS_VALU", 0) / blocks, "salu_per_block": per.get("SQ_INSTS_SALU", 0) / blocks,
                     "wave_time": {"waiting_on_memory_or_lds(SQ_WAIT_ANY)": per.get("SQ_WAIT_ANY", 0) / wc, "issue_stalled(SQ_WAIT_INST_ANY)": per.get("SQ_WAIT_INST_ANY", 0) / wc,
                                   "issuing(SQ_ACTIVE_INST_ANY)": per.get("SQ_ACTIVE_INST_ANY", 0) / wc}}
        print(v, k, json.dumps({a: (round(b, 3) if isinstance(b, float) else b) for a, b in res[v][k].items()}))
json.dump(res, open("$OUT/counters.json", "w"), indent=1)
PY
