#!/usr/bin/env python3
"""End-to-end batch pipeline (BASELINE.json configs[2]): JPEG files in host memory -> .lep files in host memory and back,
through lep_compress_batch / lep_decompress_batch (host-pool Huffman, overlapped PCIe copies, GPU coder kernels).
PCIe- and host-inclusive: this is NOT bench.py's `value` (which keeps frames resident in HBM); see DESIGN.md.
usage: python scripts/bench_batch.py [--images 1024] [--unique 64] [--width 1920 --height 1080] [--verify] [--threads 0]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=1024)
    ap.add_argument("--unique", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--chunk-mb", type=int, default=0)
    ap.add_argument("--chunk-images", type=int, default=0)
    ap.add_argument("--verify", action="store_true")
    ap.add_argument("--skew", type=float, default=0.0, help="photograph-like corpus: detail grows from top to bottom (exponent of the ramp), unequal thread segments")
    ap.add_argument("--progressive", action="store_true", help="progressive corpus (BASELINE.json configs[4])")
    ap.add_argument("--host-huffman", action="store_true", help="decompress: JPEG Huffman re-encode on the host pool instead of the GPU")
    args = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    from lepton_amd import corpus
    from lepton_amd.codec import GpuCodec

    codec = GpuCodec(0)
    nu = max(1, min(args.unique, args.images))
    uniq = corpus.make_corpus(nu, args.width, args.height, 10000, skew=args.skew, progressive=args.progressive)
    jpgs = [uniq[i % nu] for i in range(args.images)]
    mb = sum(map(len, jpgs)) / 1e6
    kw = dict(threads=args.threads, chunk_bytes=args.chunk_mb << 20, chunk_images=args.chunk_images, host_huffman=args.host_huffman)
    # first call: cold (kernel images, staging buffers of both pipeline slots are allocated inside it); second call: the
    # steady state of a serving process, which keeps the staging between batches
    tc0 = time.perf_counter()
    leps_c, st, cold_c = codec.compress_batch(jpgs, verify=args.verify, **kw)
    tc1 = time.perf_counter()
    assert not any(st), sorted(set(st))
    _, _, cold_d = codec.decompress_batch(leps_c, **kw)
    del leps_c
    t0 = time.perf_counter()
    leps, st, cs = codec.compress_batch(jpgs, verify=args.verify, **kw)
    t1 = time.perf_counter()
    assert not any(st), sorted(set(st))
    back, st2, ds = codec.decompress_batch(leps, **kw)
    from lepton_amd import abi
    last_kernel = (abi.lib().lep_gpu_last_kernel_name(codec.handle).decode(), round(abi.lib().lep_gpu_last_kernel_ms(codec.handle), 3))
    t2 = time.perf_counter()
    assert not any(st2) and back == jpgs, "round trip is not bit exact"
    out = {
        "workload": "%d x %dx%d 4:2:0 baseline JPEGs (%d distinct), host memory -> host memory, %s" % (
            args.images, args.width, args.height, nu, "with on-GPU round-trip verification" if args.verify else "no verification"),
        "jpeg_MB": round(mb, 1), "lep_MB": round(sum(map(len, leps)) / 1e6, 1), "host_threads": args.threads or "cgroup quota",
        "compress": {"MBps_pipeline": round(mb / cs["pipeline_s"], 1), "MBps_wall": round(mb / cs["wall_s"], 1), "python_wall_s": round(t1 - t0, 3),
                     "h2d_GBps": round(cs["h2d_bytes"] / cs["pipeline_s"] / 1e9, 2), **{k: round(v, 3) for k, v in cs.items()}},
        "decompress": {"MBps_pipeline": round(mb / ds["pipeline_s"], 1), "MBps_wall": round(mb / ds["wall_s"], 1), "python_wall_s": round(t2 - t1, 3),
                       "d2h_GBps": round(ds["d2h_bytes"] / ds["pipeline_s"] / 1e9, 2), **{k: round(v, 3) for k, v in ds.items()}},
        "cold_first_call": {"compress_MBps_wall": round(mb / cold_c["wall_s"], 1), "compress_alloc_s": round(cold_c["alloc_s"], 3),
                            "decompress_MBps_wall": round(mb / cold_d["wall_s"], 1), "decompress_alloc_s": round(cold_d["alloc_s"], 3)},
        "skew": args.skew, "overlap_launches": os.environ.get("LEP_BATCH_OVERLAP", "0"),
        "roundtrip": "bit exact (%d files)" % args.images, "huffman": "host pool" if args.host_huffman else "GPU (lep_huffman_decode_kernel / lep_huffman_encode_kernel)", "last_kernel_of_last_chunk_ms": last_kernel,
    }
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
