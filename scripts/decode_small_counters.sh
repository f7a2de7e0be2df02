#!/bin/bash
# Where a SMALL decode launch spends its wave time: one rocprofv3 --pmc pass (kernel trace only) over device-resident decode launches of
# 1, 8, 64 and 256 4K images (8 .. 2048 thread segments).   scripts/decode_small_counters.sh <tag>  -> gpurun_out/<tag>/decode_small_counters.txt
set -u
export TMPDIR=/tmp
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
cat > /tmp/dec_small.py <<'PY'
import ctypes as C, sys
sys.path.insert(0, '.')
import __graft_entry__ as ge
ge.build()
from lepton_amd import abi, corpus
from lepton_amd.codec import GpuCodec, JpegImage
L = abi.lib(); codec = GpuCodec(0); g = codec.handle
uniq = corpus.make_corpus(8, 3840, 2160, 20001)
imgs = [JpegImage(j) for j in uniq]; plans = [im.plan() for im in imgs]
nmax = 256
def dmalloc(n):
    p = C.c_void_p(); assert L.lep_gpu_malloc(g, n, C.byref(p)) == 0; return p.value
descs = (abi.ImageDesc * nmax)(); dec = (abi.ImageDesc * nmax)(); flat = []; first = {}
for k in range(nmax):
    u = k % 8; d = imgs[u].desc
    C.memmove(C.byref(descs[k]), C.byref(d), C.sizeof(abi.ImageDesc)); C.memmove(C.byref(dec[k]), C.byref(d), C.sizeof(abi.ImageDesc))
    for c in range(d.ncomp):
        n = d.nblocks(c) * 128
        if (u, c) not in first:
            p = dmalloc(n); assert L.lep_gpu_memcpy_h2d(g, p, d.blocks[c], n) == 0; first[(u, c)] = p
        descs[k].blocks[c] = first[(u, c)]
        dec[k].blocks[c] = dmalloc(n)
    for s in plans[u]: flat.append(abi.Segment(k, s.luma_y_start, s.luma_y_end, s.is_last))
nseg = len(flat); segs = (abi.Segment * nseg)(*flat)
offs = (C.c_uint64 * (nseg + 1))()
for i, s in enumerate(flat):
    d = descs[s.image]; offs[i + 1] = offs[i] + ((d.total_blocks() * 40 // 8 + 65536 + 255) & ~255)
d_streams = dmalloc(offs[nseg]); d_len = dmalloc(4 * nseg); d_status = dmalloc(4 * nseg)
assert L.lep_gpu_encode_device(g, descs, nmax, segs, nseg, d_streams, offs, d_len, d_status, None) == 0; L.lep_gpu_sync(g)
for nb in (1, 8, 64, 256):
    ns = nb * 8
    assert L.lep_gpu_decode_device(g, dec, nb, segs, ns, d_streams, offs, d_len, d_status, None) == 0; L.lep_gpu_sync(g)
    print(nb, L.lep_gpu_last_kernel_name(g).decode(), L.lep_gpu_last_kernel_ms(g))
PY
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc -o pmc --output-format csv -- python /tmp/dec_small.py > $OUT/pmc.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_FLAT --kernel-trace -d $OUT/pmc2 -o pmc --output-format csv -- python /tmp/dec_small.py > $OUT/pmc2.log 2>&1
python - <<PY | tee $OUT/decode_small_counters.txt
import csv, glob, collections
rows = collections.OrderedDict()
for fn in sorted(glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(fn)):
        if "lep_decode_v4" not in r["Kernel_Name"]: continue
        rows.setdefault(int(r["Grid_Size"]) // 64, {})[r["Counter_Name"]] = float(r["Counter_Value"])
print("lep_decode_v4_kernel on device-resident 4K images, per BLOCK (24,300 blocks per thread segment); wave cycles = SQ_WAVE_CYCLES x 4 / segments / blocks")
names = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_FLAT"]
print("%9s %10s " % ("segments", "wave cyc") + " ".join("%9s" % n[8:] for n in names) + "   waiting  wait_inst  issuing  | busy VALU  SCA  LDS  VMEM  wait_LDS (fractions of wave cycles)")
for seg, c in sorted(rows.items()):
    nb = seg * 24300.0
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    f = lambda k: c.get(k, 0.0) / wc
    print("%9d %10.0f " % (seg, wc * 4 / nb) + " ".join("%9.1f" % (c.get(n, 0.0) / nb) for n in names) +
          "   %7.2f %9.2f %8.2f  | %9.2f %5.2f %4.2f %5.2f %8.2f" % (f("SQ_WAIT_ANY"), f("SQ_WAIT_INST_ANY"), f("SQ_ACTIVE_INST_ANY"), f("SQ_ACTIVE_INST_VALU"), f("SQ_ACTIVE_INST_SCA"), f("SQ_ACTIVE_INST_LDS"), f("SQ_INST_CYCLES_VMEM"), f("SQ_WAIT_INST_LDS")))
PY
rm -rf $OUT/pmc $OUT/pmc2
