#!/usr/bin/env python3
"""Encode / decode latency by launch size (device-resident frames, launch + kernel + sync, best of 3) for the encoder's forms for
small launches (round 4: the stitched writer).  Each form runs in its own process (the library reads its switches when the codec
object is created)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, json, os, sys, time
sys.path.insert(0, %r)
from lepton_amd import abi, corpus
from lepton_amd.codec import GpuCodec, JpegImage
L = abi.lib(); codec = GpuCodec(0); g = codec.handle
uniq = corpus.make_corpus(8, 3840, 2160, 20001)
imgs = [JpegImage(j) for j in uniq]; plans = [im.plan() for im in imgs]
sizes = [1, 2, 8, 32, 64, 128, 256, 512]
nmax = max(sizes)
def dmalloc(n):
    p = C.c_void_p(); assert L.lep_gpu_malloc(g, n, C.byref(p)) == 0; return p.value
descs = (abi.ImageDesc * nmax)(); dec = (abi.ImageDesc * nmax)(); flat = []
first = {}
for k in range(nmax):
    u = k %% 8; d = imgs[u].desc
    C.memmove(C.byref(descs[k]), C.byref(d), C.sizeof(abi.ImageDesc)); C.memmove(C.byref(dec[k]), C.byref(d), C.sizeof(abi.ImageDesc))
    for c in range(d.ncomp):
        n = d.nblocks(c) * 128
        if (u, c) not in first:
            p = dmalloc(n); assert L.lep_gpu_memcpy_h2d(g, p, d.blocks[c], n) == 0; first[(u, c)] = p
        descs[k].blocks[c] = first[(u, c)]          # encode only reads: replicas share the frame
        q = dmalloc(n); dec[k].blocks[c] = q
    for s in plans[u]: flat.append(abi.Segment(k, s.luma_y_start, s.luma_y_end, s.is_last))
nseg = len(flat); segs = (abi.Segment * nseg)(*flat)
offs = (C.c_uint64 * (nseg + 1))()
for i, s in enumerate(flat):
    d = descs[s.image]; offs[i + 1] = offs[i] + ((d.total_blocks() * 40 // 8 + 65536 + 255) & ~255)
d_streams = dmalloc(offs[nseg]); d_len = dmalloc(4 * nseg); d_status = dmalloc(4 * nseg)
out = {"encode_ms": {}, "decode_ms": {}, "kernel": {}}
for nb in sizes:
    ns = sum(len(plans[k %% 8]) for k in range(nb))
    be = bd = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); assert L.lep_gpu_encode_device(g, descs, nb, segs, ns, d_streams, offs, d_len, d_status, None) == 0; L.lep_gpu_sync(g); t1 = time.perf_counter()
        ke = L.lep_gpu_last_kernel_name(g).decode()
        assert L.lep_gpu_decode_device(g, dec, nb, segs, ns, d_streams, offs, d_len, d_status, None) == 0; L.lep_gpu_sync(g); t2 = time.perf_counter()
        be = min(be, t1 - t0); bd = min(bd, t2 - t1)
    st = (C.c_int32 * ns)(); L.lep_gpu_memcpy_d2h(g, st, d_status, 4 * ns); assert not any(st)
    out["encode_ms"][str(nb)] = round(be * 1e3, 1); out["decode_ms"][str(nb)] = round(bd * 1e3, 1); out["kernel"][str(nb)] = ke
print(json.dumps(out))
'''


def main():
    res = {}
    # the encoder's forms for small launches: the default choice (round 4: the split-phase encoder with the stitched writer from 64
    # segments on, the two-wavefront single-kernel encoder below), the lane-per-segment writer instead of the stitched one, and the
    # split-phase encoder for launches of any size
    forms = (("default", {}), ("LEP_ENC5_WCHUNKS=0", {"LEP_ENC5_WCHUNKS": "0"}), ("LEP_ENC5_MIN=1", {"LEP_ENC5_MIN": "1"}),
             ("LEP_ENC5_MIN=1 LEP_ENC5_WCHUNKS=0", {"LEP_ENC5_MIN": "1", "LEP_ENC5_WCHUNKS": "0"}))
    if len(sys.argv) > 1:      # the forms named on the command line instead: "LEP_ENC5_WLANES=65536" "LEP_ENC5_WLANES=262144 LEP_ENC5_WCHUNKS=128" ...
        forms = tuple((a or "default", dict(kv.split("=") for kv in a.split())) for a in sys.argv[1:])
    for name, extra in forms:
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, env=env, timeout=400)
        try:
            res[name] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            res[name] = {"error": r.stderr[-500:]}
    print(json.dumps({"workload": "4K 4:2:0 baseline images (8 thread segments each), ms per launch by images per launch", "results": res}, indent=1))


if __name__ == "__main__":
    main()
