#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the v3 kernels (profiling build, -DLEP_PROF).
usage: python scripts/prof_phases.py [--build-only] [--images N] [--replicate R]
Builds lepton_amd/liblepton_mi355x_prof.so (here, cross-compiled); on the GPU box runs a batch through it."""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "lepton_amd", "csrc")
PROF_LIB = os.path.join(ROOT, "lepton_amd", "liblepton_mi355x_prof.so")
SOURCES = ["lep_gpu.hip", "lep_batch.hip", "lep_api.cc", "jpeg_scan.cc", "jpeg_progressive.cc", "lep_container.cc", "jpeg_recode.cc", "lep_serve.cc"]


def build():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLEP_PROF", "-o", PROF_LIB]
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ["-lz", "-ldl", "-lpthread"]
    subprocess.check_call(cmd, cwd=ROOT)


DEC_NAMES = {0: "staging", 1: "prologue", 20: "lakhani", 21: "idct+dcpred", 22: "publish"}
KINDS = ["NZ", "77", "TREEH", "EDGEH", "EDGEV", "DC"]
for k, n in enumerate(KINDS):
    DEC_NAMES[2 + 3 * k] = "%s prefetch+wait" % n
    DEC_NAMES[3 + 3 * k] = "%s serial" % n
    DEC_NAMES[4 + 3 * k] = "%s update" % n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--images", type=int, default=8)
    ap.add_argument("--replicate", type=int, default=1, help="decode the same files R times in one launch (occupancy)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    args = ap.parse_args()
    if args.build_only:
        build()
        return
    from lepton_amd import abi
    abi.LIB_PATH = PROF_LIB
    from lepton_amd import corpus
    from lepton_amd.codec import GpuCodec, JpegImage, LepFile

    L = abi.lib()
    L.lep_gpu_debug_prof.argtypes = [C.c_void_p, C.c_void_p]
    codec = GpuCodec(0)
    jpgs = corpus.make_corpus(args.images, args.width, args.height, 20001)
    imgs = [JpegImage(j) for j in jpgs]
    plans = [im.plan() for im in imgs]
    streams = codec.encode(imgs, plans)
    leps = [im.write_lep(s) for im, s in zip(imgs, streams)]
    files = [LepFile(l) for l in leps for _ in range(args.replicate)]
    codec.decode(files)
    ms = L.lep_gpu_last_kernel_ms(codec.handle)
    prof = (C.c_uint64 * (64 * 32))()
    assert L.lep_gpu_debug_prof(codec.handle, prof) == 0
    nseg_all = sum(len(f.segments) for f in files)
    if os.environ.get("LEP_DECODE_KERNEL", "4") == "4":   # v4: 64 global buckets summed over all waves, slots named in lep_dec4.h
        names = ["staging", "prologue", "nz_prefetch", "nz_serial", "nz_update", "77_prefetch", "77_serial", "77_update", "lakhani",
                 "edge_prefetch", "edge_serial", "edge_update", "idct_dcpred", "dc_prefetch", "dc_serial", "dc_update", "publish", "store"]
        tot = [sum(prof[s * 32 + i] for s in range(64)) / nseg_all for i in range(32)]
        blocks = sum(f.desc.total_blocks() for f in files) / nseg_all
        cyc = sum(tot)
        print("decode kernel %.1f ms, %d segments, %.0f blocks/segment; accounted %.0f Mcycles/segment (%.0f shader-clock cycles/block)" %
              (ms, nseg_all, blocks, cyc / 1e6, cyc / blocks))
        for i, n in enumerate(names + ["other"]):
            v = tot[i] if i < len(names) else sum(tot[len(names):])
            print("  %-14s %8.0f cycles/block  %5.1f%%" % (n, v / blocks, 100 * v / cyc))
        return
    nseg = min(64, nseg_all)
    tot = [sum(prof[s * 32 + i] for s in range(nseg)) / nseg for i in range(32)]
    blocks = sum(f.desc.total_blocks() for f in files) / sum(len(f.segments) for f in files)
    cyc = sum(tot[:24])
    print("decode kernel %.1f ms, %d segments, %.0f blocks/segment; accounted %.0f Mcycles/segment (%.0f cycles/block)" %
          (ms, sum(len(f.segments) for f in files), blocks, cyc / 1e6, cyc / blocks))
    for i in range(24):
        if tot[i]:
            extra = ""
            if i >= 2 and i < 20 and (i - 2) % 3 == 0:
                extra = "  rounds/block %.2f" % (tot[24 + (i - 2) // 3] / blocks)
            print("  %-22s %8.0f cycles/block  %5.1f%%%s" % (DEC_NAMES.get(i, str(i)), tot[i] / blocks, 100 * tot[i] / cyc, extra))


if __name__ == "__main__":
    main()
