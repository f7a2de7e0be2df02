#!/usr/bin/env python3
"""Where the decode kernel's instructions go: made-up coefficient frames (4K geometry) that isolate one kind of work each, coded by the
GPU encoder, decoded under `rocprofv3 --pmc SQ_INSTS_*` -- one decode launch per workload, in the order printed.  The counters of
launch k divided by the blocks give instructions per block for workload k; differences between workloads give the cost of a block's
fixed part, of a zero / non-zero interior coefficient, of an edge coefficient, of a 7x7 window round ...
usage: rocprofv3 --pmc ... -- python scripts/dec_microbench.py [--images 32]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=32)
    args = ap.parse_args()
    from lepton_amd import abi, corpus
    from lepton_amd.codec import GpuCodec, JpegImage

    codec = GpuCodec(0)
    img = JpegImage(corpus.synth_jpeg(3840, 2160, 1234))
    d = img.desc
    segs = img.plan()
    arrs = []
    for c in range(d.ncomp):
        n = d.nblocks(c)
        arrs.append(np.ctypeslib.as_array(C.cast(d.blocks[c], C.POINTER(C.c_int16)), shape=(n, 64)))
    real = [a.copy() for a in arrs]
    rng = np.random.default_rng(1)

    def fill(kind):
        for a, r in zip(arrs, real):
            n = a.shape[0]
            a[:] = 0
            if kind == "real":
                a[:] = r
            elif kind == "zero":
                pass
            elif kind == "dc":
                a[:, 49] = rng.integers(-40, 41, n)
            elif kind.startswith("i"):      # iK_S: K interior non-zeros (+-1) at zig-zag positions 0, S, 2S, ...
                k, step = (int(x) for x in kind[1:].split("_"))
                for j in range(k):
                    a[:, j * step] = rng.choice([-1, 1], n)
            elif kind.startswith("I"):      # IK: K interior coefficients of magnitude 2..3 (one residual bit) at positions 0..K-1
                k = int(kind[1:])
                for j in range(k):
                    a[:, j] = rng.choice([-3, -2, 2, 3], n)
            elif kind.startswith("e"):      # eK: K non-zeros (+-1) at the first K positions of both edges
                k = int(kind[1:])
                for j in range(k):
                    a[:, 50 + j] = rng.choice([-1, 1], n)
                    a[:, 57 + j] = rng.choice([-1, 1], n)

    kinds = ["zero", "dc", "i1_1", "i4_1", "i8_1", "i8_2", "i8_4", "I8", "e1", "e3", "e6", "real"]
    out = []
    N = args.images
    for kind in kinds:
        fill(kind)
        streams = codec.encode([img], [segs])[0]
        nseg = len(segs) * N
        descs = (abi.ImageDesc * N)(*([d] * N))
        flat = (abi.Segment * nseg)(*[abi.Segment(i, s.luma_y_start, s.luma_y_end, s.is_last) for i in range(N) for s in segs])
        keep = [C.create_string_buffer(bytes(w), max(1, len(w))) for w in streams]
        arr = (abi.Bytes * nseg)()
        for i in range(N):
            for k, b in enumerate(keep):
                arr[i * len(segs) + k].data, arr[i * len(segs) + k].len, arr[i * len(segs) + k].cap = C.cast(b, C.c_void_p).value, len(streams[k]), len(streams[k])
        status = (C.c_int32 * nseg)()
        want = [a.copy() for a in arrs]
        rc = abi.lib().lep_gpu_decode_host(codec.handle, descs, N, flat, nseg, arr, status)
        ok = rc == 0 and all((a == w).all() for a, w in zip(arrs, want))
        out.append({"kind": kind, "ok": bool(ok), "stream_bytes": sum(map(len, streams)), "kernel": abi.lib().lep_gpu_last_kernel_name(codec.handle).decode(),
                    "kernel_ms": round(abi.lib().lep_gpu_last_kernel_ms(codec.handle), 3)})
    print(json.dumps({"images": N, "blocks_per_image": int(sum(a.shape[0] for a in arrs)), "workloads": out}))


if __name__ == "__main__":
    main()
