#!/bin/bash
# Instructions per block of the progressive scan decoder, by kind of scan: one rocprofv3 --pmc pass (kernel trace only) over a compress call of
# 64 x 4K progressive files launched level by level and kind by kind (LEP_HUFFPROG_PIPELINE=0 LEP_HUFFPROG_SPLIT=1).
#   scripts/prog_scan_counters.sh <tag>   -> gpurun_out/<tag>/prog_counters.txt
set -u
export TMPDIR=/tmp
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
export LEP_HUFFPROG_PIPELINE=0 LEP_HUFFPROG_SPLIT=1
cat > /tmp/prog_once.py <<'PY'
import sys
sys.path.insert(0, '.')
import __graft_entry__ as ge
ge.build()
from lepton_amd import corpus
from lepton_amd.codec import GpuCodec
c = GpuCodec(0)
u = corpus.make_corpus(8, 3840, 2160, 10000, progressive=True)
leps, st, _ = c.compress_batch([u[i % 8] for i in range(64)])
assert not any(st)
PY
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace -d $OUT/pmc -o pmc --output-format csv -- python /tmp/prog_once.py > $OUT/pmc.log 2>&1
python - <<PY | tee $OUT/prog_counters.txt
import csv, glob, collections
rows = collections.OrderedDict()
for fn in glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "progdec" not in r["Kernel_Name"]: continue
        rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = ["DC first (interleaved)", "luma AC 1-5 first", "luma AC 6-63 first", "Cb AC first", "Cr AC first", "DC refinement", "luma AC refinement Al=1", "Cb AC refinement", "Cr AC refinement", "luma AC refinement Al=0"]
blocks = [194400, 129600, 129600, 32400, 32400, 194400, 129600, 32400, 32400, 129600]
print("64 x 4K 4:2:0 progressive files, one wavefront per scan, per BLOCK of the scan (counter / 64 scans / blocks)")
print("%-28s %8s %8s %8s %8s %8s %10s %8s %8s" % ("kind of scan", "VALU", "SALU", "LDS", "SMEM", "VMEM rd", "wave cyc", "waiting", "issuing"))
for (d, c), nm, nb in zip(sorted(rows.items()), names, blocks):
    per = lambda k: c.get(k, 0.0) / 64.0 / nb
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    print("%-28s %8.1f %8.1f %8.1f %8.1f %8.1f %10.0f %8.2f %8.2f" % (nm, per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_LDS"), per("SQ_INSTS_SMEM"), per("SQ_INSTS_VMEM_RD"), per("SQ_WAVE_CYCLES") * 4, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc))
PY
