#!/usr/bin/env python3
"""Files with restart intervals through the batch pipeline, scan kernels A/B (GPU box):
   python scripts/bench_restart_corpora.py [--images 896] [--unique 8]
For each layout (a marker per MCU row, every 8 MCUs, none) the same pictures are compressed and restored twice: with the lane-per-piece scan
kernels (lep_huffdec_simt.h / lep_huff_simt.h, the default) and with the wavefront forms (LEP_HUFFDEC_SIMT=0 LEP_HUFFENC_SIMT=0)."""
import argparse
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pictures(n, w, h, **kw):
    import numpy as np
    from PIL import Image
    out = []
    for i in range(n):
        rng = np.random.default_rng(5000 + i)
        base = np.asarray(Image.fromarray(rng.integers(0, 256, (h // 16, w // 16, 3), dtype=np.uint8), "RGB").resize((w, h), Image.BICUBIC)).astype(np.int16)
        img = Image.fromarray(np.clip(base + rng.normal(0, 8, base.shape), 0, 255).astype(np.uint8), "RGB")
        buf = io.BytesIO()
        img.save(buf, format="JPEG", quality=90, subsampling="4:2:0", **kw)
        out.append(buf.getvalue())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=896)
    ap.add_argument("--unique", type=int, default=8)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--verbose", action="store_true", help="every decompress call's phase times")
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--only", default="", help="row | 8mcu | none: one layout only (profiling runs)")
    ap.add_argument("--simt", default="1,0", help="which forms of the scan kernels: 1,0 | 1 | 0")
    args = ap.parse_args()
    from lepton_amd.codec import GpuCodec

    for key, label, kw in (("row", "one restart interval per MCU row", dict(restart_marker_rows=1)), ("8mcu", "a restart interval of 8 MCUs", dict(restart_marker_blocks=8)),
                           ("none", "no restart intervals", {})):
        if args.only and args.only != key:
            continue
        uniq = pictures(args.unique, args.width, args.height, **kw)
        jpgs = [uniq[i % len(uniq)] for i in range(args.images)]
        mb = sum(map(len, jpgs)) / 1e6
        for simt in args.simt.split(","):
            os.environ["LEP_HUFFDEC_SIMT"] = simt
            os.environ["LEP_HUFFENC_SIMT"] = simt
            codec = GpuCodec(0)
            try:
                best_c = best_d = 1e9
                for _ in range(args.repeats):            # (the second call has its staging warm)
                    leps, st, cs = codec.compress_batch(jpgs, verify=False)
                    best_c = min(best_c, cs["wall_s"])           # (the library's clock around the call: the Python wrapper's copies of the outputs are not the pipeline's)
                    assert st == [0] * len(jpgs)
                    t0 = time.perf_counter()
                    back, st, ds = codec.decompress_batch(leps)
                    best_d = min(best_d, ds["wall_s"])
                    assert st == [0] * len(jpgs) and back == jpgs
                    if args.verbose:
                        print("    decompress call: %.3f s  %s" % (time.perf_counter() - t0, {k: round(v, 3) for k, v in ds.items() if isinstance(v, float) and k.endswith("_s")}), flush=True)
                print("lane-per-piece scan kernels %s: %d x %dx%d, %s: compress %.0f MB/s (gpu_huffman_files %d), decompress %.0f MB/s (gpu_huffman_files %d, write_s %.3f)"
                      % ("on " if simt == "1" else "off", len(jpgs), args.width, args.height, label, mb / best_c, cs["gpu_huffman_files"], mb / best_d, ds["gpu_huffman_files"], ds["write_s"]), flush=True)
            finally:
                codec.close()


if __name__ == "__main__":
    main()
