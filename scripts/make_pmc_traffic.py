#!/usr/bin/env python3
"""PMC passes of scripts/gpu_closing_visit.sh -> <out>/pmc_summary.json (raw sums per kernel) and <out>/pmc_traffic.json (what
bench.py's roofline.traffic / bound_by look up; copied to profiles/pmc_traffic.json).  Runs on the GPU box right after the
passes, so `kernel_source_sha16` is the hash of the sources the measured library was built from (bench.kernel_source_sha)."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out_dir = sys.argv[1]
BLOCKS_4K = 480 * 270 + 2 * 240 * 135   # 8x8 blocks of a 3840x2160 4:2:0 image


def sums(tag):
    res = collections.defaultdict(lambda: collections.defaultdict(float))
    for fn in glob.glob(os.path.join(out_dir, "pmc_%s" % tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            m = re.search(r"lep_\w+(<[\d, ]+>)?", r["Kernel_Name"])
            if m:
                res[m.group(0)][r["Counter_Name"]] += float(r["Counter_Value"])
    return res


full = collections.defaultdict(dict)
for tag in ("sq", "in", "mem", "mem2"):
    for k, v in sums(tag).items():
        full[k].update(v)
small = sums("mem256")
json.dump({"images_1024": full, "images_256": small}, open(os.path.join(out_dir, "pmc_summary.json"), "w"), indent=1)



def kernel_source_sha():   # bench.kernel_source_sha: sha256 over lepton_amd/csrc/*.h, *.hip
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "lepton_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    for f in ("lep_gpu.hip", "lep_batch.hip"):   # (and the compile flags build() recorded for them)
        p = os.path.join(ROOT, "lepton_amd", "build", "obj", f + ".o.flags")
        h.update(open(p, "rb").read().split(b" |src:")[0] if os.path.exists(p) else b"")   # (the flags, not the input hash build() keeps beside them)
    return h.hexdigest()[:16]


sha = kernel_source_sha()


def entry(v, images):
    rd, wr = v.get("TCC_EA0_RDREQ_sum", 0.0), v.get("TCC_EA0_WRREQ_sum", 0.0)
    wr64 = v.get("TCC_EA0_WRREQ_64B_sum")
    rd32 = v.get("TCC_EA0_RDREQ_32B_sum", 0.0)
    rb = 64.0 * (rd - rd32) + 32.0 * rd32
    wb = 64.0 * wr64 + 32.0 * (wr - wr64) if wr64 is not None else 64.0 * wr
    wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
    return {
        "kernel_source_sha16": sha, "images_per_launch": images,
        "read_bytes_per_launch": rb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": rb + wb, "hbm_bytes_per_image": (rb + wb) / images,
        "read_sectors_per_block": round(rd / (images * BLOCKS_4K), 1), "write_requests_per_block": round(wr / (images * BLOCKS_4K), 1),
        "instructions_per_block": {"valu": round(v.get("SQ_INSTS_VALU", 0) / (images * BLOCKS_4K), 1), "salu": round(v.get("SQ_INSTS_SALU", 0) / (images * BLOCKS_4K), 1),
                                   "lds": round(v.get("SQ_INSTS_LDS", 0) / (images * BLOCKS_4K), 1), "vmem": round((v.get("SQ_INSTS_VMEM_RD", 0) + v.get("SQ_INSTS_VMEM_WR", 0)) / (images * BLOCKS_4K), 1)},
        "bound": {"l2_hit_rate": round(hit / ((hit + miss) or 1.0), 3),
                  "wave_time": {"waiting_on_memory_or_lds(SQ_WAIT_ANY)": round(v.get("SQ_WAIT_ANY", 0) / wc, 3), "issue_stalled(SQ_WAIT_INST_ANY)": round(v.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
                                "issuing(SQ_ACTIVE_INST_ANY)": round(v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)}},
    }


kernels = {k.split("<")[0] if k.startswith("lep_decode") else k: entry(v, 1024) for k, v in full.items() if "TCC_EA0_RDREQ_sum" in v}
enc5 = [k for k in full if k.startswith("lep_enc5_")]
if enc5:   # the split-phase encoder is a sequence of launches: bench.py names it as one
    tot = collections.defaultdict(float)
    for k in enc5:
        for c, x in full[k].items():
            tot[c] += x
    kernels["lep_enc5 (count | emit | fold | gather | write)"] = dict(entry(tot, 1024), launches=sorted(enc5))
dec = [k for k in small if k.startswith("lep_decode")]
note_256 = None
if dec:
    v = small[dec[0]]
    note_256 = {"images_per_launch": 256, "read_sectors_per_block": round(v.get("TCC_EA0_RDREQ_sum", 0) / (256 * BLOCKS_4K), 1),
                "l2_hit_rate": round(v.get("TCC_HIT_sum", 0) / ((v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) or 1.0), 3)}
# the same table for the 256-image launch (request counters only: bytes counted at 64 per request, no 32-byte split) -- bench.py
# looks a launch size up here and reports no traffic figure for a size that was not measured (it used to scale the 1024-image
# figure linearly; the decoder's bytes per block DOUBLE between 256 and 1024 images)
def small_entry(v):
    rd, wr = v.get("TCC_EA0_RDREQ_sum", 0.0), v.get("TCC_EA0_WRREQ_sum", 0.0)
    hit, miss = v.get("TCC_HIT_sum", 0.0), v.get("TCC_MISS_sum", 0.0)
    return {"kernel_source_sha16": sha, "images_per_launch": 256, "hbm_bytes_per_launch": 64.0 * (rd + wr), "hbm_bytes_per_image": 64.0 * (rd + wr) / 256,
            "read_sectors_per_block": round(rd / (256 * BLOCKS_4K), 1), "write_requests_per_block": round(wr / (256 * BLOCKS_4K), 1),
            "bound": {"l2_hit_rate": round(hit / ((hit + miss) or 1.0), 3)}, "note": "64 B per request assumed (no 32-byte split in this pass)"}
kernels_256 = {k.split("<")[0] if k.startswith("lep_decode") else k: small_entry(v) for k, v in small.items() if "TCC_EA0_RDREQ_sum" in v}
enc5s = [k for k in small if k.startswith("lep_enc5_")]
if enc5s:
    tot = collections.defaultdict(float)
    for k in enc5s:
        for c, x in small[k].items():
            tot[c] += x
    kernels_256["lep_enc5 (count | emit | fold | gather | write)"] = small_entry(tot)
json.dump({
    "by_images_per_launch": {"1024": kernels, "256": kernels_256},
    "source": "rocprofv3 --pmc, five separate passes (SQ wave time; SQ instruction counts; TCC_EA0_RDREQ / WRREQ / HIT / MISS; WRREQ_64B / RDREQ_32B / REQ / READ; "
              "the request counters again at 256 images), --kernel-trace only -- python bench.py --steps 1 --warmup 0 --no-extras --no-end-to-end --no-cpu-baseline "
              "--mixed-images 0 (scripts/gpu_closing_visit.sh, scripts/make_pmc_traffic.py; raw sums: pmc_summary.json beside this file's source in profiles/)",
    "units": "hbm_bytes = 64 B x (TCC_EA0_RDREQ - RDREQ_32B) + 32 B x RDREQ_32B + 64 B x WRREQ_64B + 32 B x the other write requests, per launch (calibration of the "
             "request size for this access pattern: profiles/r02b_fetch_calibration.txt)",
    "kernel_source_sha16": sha, "images_per_launch": 1024, "blocks_per_image": BLOCKS_4K,
    "decoder_at_256_images": note_256,
    "kernels": kernels,
}, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1)
for k, e in kernels.items():
    print("%-52s rd sectors/block %7.1f  wr req/block %6.1f  hbm GB/launch %8.1f  L2 hit %.2f  VALU/block %7.1f SALU/block %7.1f" % (
        k[:52], e["read_sectors_per_block"], e["write_requests_per_block"], e["hbm_bytes_per_launch"] / 1e9, e["bound"]["l2_hit_rate"],
        e["instructions_per_block"]["valu"], e["instructions_per_block"]["salu"]))
print("decoder at 256 images:", note_256)
