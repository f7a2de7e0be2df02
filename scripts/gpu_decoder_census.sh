#!/bin/bash
# instructions per block of the decode kernels on made-up frames (scripts/dec_microbench.py)
set -u
TAG=${1:-r5f}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
t0=$(date +%s)
for V in ${VARIANTS:-v4}; do
  case $V in v4) ENVV="LEP_DEC_WAVES=8";; *) ENVV="LEP_LIB_PATH=$PWD/lepton_amd/liblepton_$V.so";; esac   # v4 = the product; any other name = an experiment build (scripts/build_variant.sh)
  env $ENVV timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $OUT/pmc_$V -o pmc --output-format csv -- python scripts/dec_microbench.py --images ${IMAGES:-32} > $OUT/micro_$V.json 2> $OUT/micro_$V.err
  echo "$V rc=$? $(( $(date +%s)-t0 )) s"
done
python - <<PY
import csv, glob, collections, json
res = {}
for V in "${VARIANTS:-v4}".split():
    try: meta = json.load(open("$OUT/micro_%s.json" % V))
    except Exception as e: print(V, "no output", e); continue
    rows = collections.defaultdict(dict)
    for fn in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % V, recursive=True):
        for r in csv.DictReader(open(fn)):
            if "lep_decode" not in r["Kernel_Name"]: continue
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(rows)
    blocks = meta["images"] * meta["blocks_per_image"]
    tab = []
    for w, i in zip(meta["workloads"], ids):
        c = rows[i]
        tab.append({"kind": w["kind"], "ok": w["ok"], "kernel_ms": w["kernel_ms"], **{k.replace("SQ_", ""): round(v / blocks, 1) for k, v in sorted(c.items())}})
    res[V] = tab
    print(V, meta["workloads"][0]["kernel"])
    for t in tab: print("  ", t)
json.dump(res, open("$OUT/dec_micro_per_block.json", "w"), indent=1)
PY
rm -rf $OUT/pmc_*/; echo "total $(( $(date +%s)-t0 )) s"
