#!/bin/bash
# PMC passes over the coder kernels of the default bench launch (separate --pmc runs, kernel trace only): where the wave time
# goes (parked / issue-stalled / issuing, by unit), instruction mix per kernel, LDS conflicts, memory-side requests
set -u
TAG=${1:-r04e}; export TMPDIR=/tmp
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="python bench.py --steps 1 --warmup 0 --no-extras --no-end-to-end --no-cpu-baseline --mixed-images 0 ${BENCH_EXTRA:-}"
t0=$(date +%s)
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace -d $OUT/pmc_sq -o pmc --output-format csv -- $B > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err; echo "sq rc=$? $(( $(date +%s)-t0 )) s"
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --kernel-trace -d $OUT/pmc_in -o pmc --output-format csv -- $B > $OUT/pmc_in.json 2> $OUT/pmc_in.err; echo "insts rc=$? $(( $(date +%s)-t0 )) s"
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/pmc_mem -o pmc --output-format csv -- $B > $OUT/pmc_mem.json 2> $OUT/pmc_mem.err; echo "mem rc=$? $(( $(date +%s)-t0 )) s"
python - <<PY
import csv, glob, json, collections
res = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for tag in ("sq", "in", "mem"):
    for fn in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(fn)):
            import re
            m = re.search(r"lep_\w+(<\d+>)?", r["Kernel_Name"])
            if not m: continue
            k = m.group(0)
            res[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
out = {k: dict(v, dispatches=len(calls[k])) for k, v in res.items()}
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
for k, v in out.items():
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    print(k[:44], "disp", v["dispatches"], "| parked %.2f stalled %.2f issuing %.2f | VALU %.3g SALU %.3g LDS %.3g VMEMr %.3g VMEMw %.3g SMEM %.3g waves %.3g | ldsconf %.2f | EArd %.3g EAwr %.3g L2hit %.2f" % (
        v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0), v.get("SQ_INSTS_LDS", 0), v.get("SQ_INSTS_VMEM_RD", 0), v.get("SQ_INSTS_VMEM_WR", 0), v.get("SQ_INSTS_SMEM", 0), v.get("SQ_WAVES", 0),
        v.get("SQ_LDS_BANK_CONFLICT", 0) / (v.get("SQ_LDS_IDX_ACTIVE", 0) or 1), v.get("TCC_EA0_RDREQ_sum", 0), v.get("TCC_EA0_WRREQ_sum", 0),
        v.get("TCC_HIT_sum", 0) / ((v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) or 1)))
PY
rm -rf $OUT/pmc_sq $OUT/pmc_in $OUT/pmc_mem
tail -2 $OUT/pmc_sq.err | cut -c1-200
echo "total $(( $(date +%s)-t0 )) s"
