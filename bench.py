#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X (contract: see the build prompt / DESIGN.md).

A "step" = one encode pass + one decode pass of the arithmetic-coding hot path over one batch of
synthetic 4K 4:2:0 baseline JPEGs whose coefficient frames (encode input) and streams (decode input)
are already resident in HBM.  value = JPEG file bytes coded per second of wall clock over the K timed
steps, aggregated over all ranks (weak scaling: every rank codes its own `--images` images)."""
import argparse
import ctypes as C
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baseline(jpgs, budget_s=20.0):
    """Reference binary (oracle/_ref/lepton, built from the real reference) timed on this host:
    `lepton -singlethread -unjailed -skipverify` encode then decode per file (one core; also the arithmetic-coding
    interval alone, TS_ARITH_STARTED..FINISHED from its stderr, for a like-for-like hot-path number), then the same
    files with the reference's default thread pool (one thread per segment, 8 for these files)."""
    ref = os.path.join(ROOT, "oracle", "_ref", "lepton")
    if not os.path.exists(ref):
        return None
    tmp = tempfile.mkdtemp(prefix="lepbench")

    def arith(err):
        st = [float(x) for x in re.findall(r"TS_ARITH_STARTED\s+\(\d+\)\s+([0-9.]+)", err)]
        fi = [float(x) for x in re.findall(r"TS_ARITH_FINISHED\s+\(\d+\)\s+([0-9.]+)", err)]
        return sum(f - s for s, f in zip(st, fi) if f > s)

    def run(flags, budget):
        enc_s = dec_s = arith_enc = arith_dec = 0.0
        nbytes = n = 0
        t_start = time.perf_counter()
        for i, j in enumerate(jpgs):
            jp, lp, bp = (os.path.join(tmp, "%d.%s" % (i, e)) for e in ("jpg", "lep", "back.jpg"))
            open(jp, "wb").write(j)
            t0 = time.perf_counter()
            r = subprocess.run([ref] + flags + ["-unjailed", "-skipverify", jp, lp], capture_output=True, text=True)
            t1 = time.perf_counter()
            if r.returncode:
                continue
            r2 = subprocess.run([ref] + flags + ["-unjailed", lp, bp], capture_output=True, text=True)
            t2 = time.perf_counter()
            if r2.returncode or open(bp, "rb").read() != j:
                continue
            enc_s += t1 - t0; dec_s += t2 - t1
            arith_enc += arith(r.stderr); arith_dec += arith(r2.stderr)
            nbytes += len(j); n += 1
            if time.perf_counter() - t_start > budget:
                break
        return n, nbytes / 1e6, enc_s, dec_s, arith_enc, arith_dec

    n, mb, enc_s, dec_s, arith_enc, arith_dec = run(["-singlethread"], budget_s)
    if not n:
        return None
    out = {
        "value": round(mb / (enc_s + dec_s), 3), "unit": "MB/s", "cores": 1, "kind": "reference",
        "sample": "%d of the bench's 4K JPEGs (%.1f MB), reference `lepton -singlethread -unjailed -skipverify`, encode then decode, whole process wall clock" % (n, mb),
        "encode_MBps": round(mb / enc_s, 3), "decode_MBps": round(mb / dec_s, 3),
        "hot_path_only_encode_MBps": round(mb / arith_enc, 3) if arith_enc else None,
        "hot_path_only_decode_MBps": round(mb / arith_dec, 3) if arith_dec else None,
        "host_cpus": os.cpu_count(),
    }
    # throughput mode (SURVEY.md 8d iii, what `lepton -benchmark` does with its "Loaded N" rows): P single-threaded
    # processes side by side, P = the CPUs this container may use (affinity mask capped by the cgroup quota)
    try:
        from concurrent.futures import ThreadPoolExecutor

        P = len(os.sched_getaffinity(0))
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                P = max(1, min(P, int(int(q) / int(per))))
        except Exception:
            pass
        deadline = time.perf_counter() + budget_s / 2

        def worker(w):
            done = 0
            k = w
            while time.perf_counter() < deadline:
                j = jpgs[k % len(jpgs)]
                jp, lp, bp = (os.path.join(tmp, "p%d.%s" % (w, e)) for e in ("jpg", "lep", "back.jpg"))
                open(jp, "wb").write(j)
                if subprocess.run([ref, "-singlethread", "-unjailed", "-skipverify", jp, lp], capture_output=True).returncode:
                    break
                if subprocess.run([ref, "-singlethread", "-unjailed", lp, bp], capture_output=True).returncode:
                    break
                done += len(j)
                k += P
            return done

        t0 = time.perf_counter()
        with ThreadPoolExecutor(P) as ex:
            total = sum(ex.map(worker, range(P)))
        wall = time.perf_counter() - t0
        if total:
            out["all_cores"] = {"processes": P, "value": round(total / 1e6 / wall, 3), "unit": "MB/s (encode + decode of every file, aggregate)",
                                "sample": "%d single-threaded reference processes side by side for %.1f s (%.1f MB round-tripped)" % (P, wall, total / 1e6)}
    except Exception as e:
        out["all_cores"] = {"error": repr(e)[:200]}
    n2, mb2, e2, d2, _, _ = run([], budget_s / 3)
    if n2:
        out["multithread"] = {"threads": 8, "value": round(mb2 / (e2 + d2), 3), "encode_MBps": round(mb2 / e2, 3), "decode_MBps": round(mb2 / d2, 3),
                              "sample": "%d files, reference default thread pool (`lepton -unjailed -skipverify`, one thread per segment)" % n2}
    return out


def pmc_traffic(kernel, images):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 passes (profiles/pmc_traffic.json: memory-side request
    counters TCC_EA0_RDREQ x 64 B + TCC_EA0_WRREQ by size, separate --pmc passes, scripts/gpu_r2_visit1.sh), scaled from that
    run's batch to this one; None when the kernel has not been through a PMC pass.  A table lookup, not a measurement of this
    run: the entry names the kernel build it was taken from."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"]
        per_image = table[kernel.split("<")[0]]["hbm_bytes_per_image"]
        return int(per_image * images)
    except Exception:
        return None


def pmc_bound(kernel):
    """what the counters say bounds `kernel` (profiles/pmc_traffic.json: L2 hit rate, wave time split)"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"][kernel.split("<")[0]].get("bound")
    except Exception:
        return None


def pipeline_figure(codec, jpgs, label, verify=False):
    """JPEG files in host memory -> .lep files in host memory and back through the batch pipeline; one warm-up call (staging
    buffers, kernel images), one timed call each way; every file must come back bit-exact"""
    mb = sum(map(len, jpgs)) / 1e6
    warm, st0, _ = codec.compress_batch(jpgs, verify=verify)
    assert not any(st0), sorted(set(st0))
    codec.decompress_batch(warm)
    del warm
    leps, st1, cs = codec.compress_batch(jpgs, verify=verify)
    back, st2, ds = codec.decompress_batch(leps)
    assert not any(st1) and not any(st2) and back == jpgs, label + ": round trip is not bit exact"
    return {"workload": label, "jpeg_MB": round(mb, 1), "lep_MB": round(sum(map(len, leps)) / 1e6, 1), "files": len(jpgs),
            "compress_MBps": round(mb / cs["wall_s"], 1), "decompress_MBps": round(mb / ds["wall_s"], 1),
            "value": round(2 * mb / (cs["wall_s"] + ds["wall_s"]), 1), "files_per_s": round(2 * len(jpgs) / (cs["wall_s"] + ds["wall_s"]), 1),
            "parity": "every file restored bit-exact", "_cs": cs, "_ds": ds}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=1024,
                    help="4K images per GPU per step (8 thread segments each; 1024 = 8192 segments = 8 wavefronts per SIMD)")
    ap.add_argument("--unique", type=int, default=64, help="distinct synthetic images per GPU (replicated up to --images)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary corpora (skewed, 1080p, progressive) and the latency table")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-memory -> host-memory pipeline measurement (lep_compress_batch / lep_decompress_batch)")
    ap.add_argument("--e2e-images", type=int, default=2688)   # 3 pipeline chunks of 896 images = 7168 thread segments each
    args = ap.parse_args()

    import __graft_entry__ as ge
    from lepton_amd import abi, corpus, shard
    from lepton_amd.codec import GpuCodec, JpegImage

    rank, local_rank, world = shard.dist_env()
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if rank == 0:
        ge.build()
    if dist:
        dist.barrier()
    L = abi.lib()
    codec = GpuCodec(local_rank)
    g = codec.handle

    # ---- corpus: weak scaling, every rank has its own distinct images
    t0 = time.perf_counter()
    nuniq = max(1, min(args.unique, args.images))
    seeds = shard.weak_seeds(nuniq, rank, 20001)
    uniq = corpus.make_corpus(nuniq, args.width, args.height, seeds[0])
    log("[rank %d] corpus: %d unique %dx%d JPEGs in %.1fs" % (rank, nuniq, args.width, args.height, time.perf_counter() - t0))
    imgs = [JpegImage(j) for j in uniq]
    plans = [im.plan() for im in imgs]
    order = [i % nuniq for i in range(args.images)]
    jpeg_bytes = sum(len(uniq[i]) for i in order)
    nimg = len(order)

    # ---- make everything resident in HBM
    def dmalloc(n):
        p = C.c_void_p()
        rc = L.lep_gpu_malloc(g, n, C.byref(p))
        assert rc == 0, codec.last_error()
        return p.value

    descs = (abi.ImageDesc * nimg)()
    dec_descs = (abi.ImageDesc * nimg)()
    flat = []
    nblocks = 0
    first_copy = {}   # unique image -> its device frame: uploaded once over PCIe, replicated device-to-device
    for k, u in enumerate(order):
        d = imgs[u].desc
        C.memmove(C.byref(descs[k]), C.byref(d), C.sizeof(abi.ImageDesc))
        C.memmove(C.byref(dec_descs[k]), C.byref(d), C.sizeof(abi.ImageDesc))
        for c in range(d.ncomp):
            n = d.nblocks(c) * 128
            p = dmalloc(n)
            if (u, c) in first_copy:
                assert L.lep_gpu_memcpy_d2d(g, p, first_copy[(u, c)], n) == 0
            else:
                assert L.lep_gpu_memcpy_h2d(g, p, d.blocks[c], n) == 0
                first_copy[(u, c)] = p
            descs[k].blocks[c] = p
            q = dmalloc(n)
            assert L.lep_gpu_memset(g, q, 0, n) == 0
            dec_descs[k].blocks[c] = q
        nblocks += d.total_blocks()
        for s in plans[u]:
            flat.append(abi.Segment(k, s.luma_y_start, s.luma_y_end, s.is_last))
    nseg = len(flat)
    segs = (abi.Segment * nseg)(*flat)
    offs = (C.c_uint64 * (nseg + 1))()
    for i, s in enumerate(flat):
        d = descs[s.image]
        per = len(plans[order[s.image]])
        offs[i + 1] = offs[i] + ((d.total_blocks() * 40 // per + 65536 + 255) & ~255)
    d_streams = dmalloc(offs[nseg])
    d_len = dmalloc(4 * nseg)
    d_status = dmalloc(4 * nseg)

    names = {}

    def step():
        rc = L.lep_gpu_encode_device(g, descs, nimg, segs, nseg, d_streams, offs, d_len, d_status, None)
        assert rc == 0, (rc, codec.last_error())
        rc = L.lep_gpu_sync(g)
        assert rc == 0, codec.last_error()
        e_ms = L.lep_gpu_last_kernel_ms(g)
        names["encode"] = L.lep_gpu_last_kernel_name(g).decode()
        rc = L.lep_gpu_decode_device(g, dec_descs, nimg, segs, nseg, d_streams, offs, d_len, d_status, None)
        assert rc == 0, (rc, codec.last_error())
        rc = L.lep_gpu_sync(g)
        assert rc == 0, codec.last_error()
        names["decode"] = L.lep_gpu_last_kernel_name(g).decode()
        return e_ms, L.lep_gpu_last_kernel_ms(g)

    def barrier():
        if dist:
            dist.barrier()
        L.lep_gpu_sync(g)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    enc_ms = dec_ms = 0.0
    for _ in range(args.steps):
        e, d = step()
        enc_ms += e; dec_ms += d
    barrier()
    elapsed = time.perf_counter() - t0

    # ---- validity (outside the timed region): statuses, stream parity with the oracle, exact round trip
    status = (C.c_int32 * nseg)()
    lens = (C.c_uint32 * nseg)()
    L.lep_gpu_memcpy_d2h(g, status, d_status, 4 * nseg)
    L.lep_gpu_memcpy_d2h(g, lens, d_len, 4 * nseg)
    assert not any(status), "segment exit codes: %s" % sorted(set(status))
    stream_bytes = sum(lens)
    import hashlib
    for k in range(min(nimg, nuniq)):
        d = imgs[order[k]].desc
        for c in range(d.ncomp):
            n = d.nblocks(c) * 128
            buf = C.create_string_buffer(n)
            L.lep_gpu_memcpy_d2h(g, buf, dec_descs[k].blocks[c], n)
            assert hashlib.md5(buf.raw).digest() == hashlib.md5(C.string_at(d.blocks[c], n)).digest(), "decode(encode(x)) != x"
    parity = "roundtrip-only"
    try:
        import oracle_binding as ob
        from concurrent.futures import ThreadPoolExecutor

        ob.oracle()
        nchk = min(nimg, nuniq)
        with ThreadPoolExecutor(min(16, len(os.sched_getaffinity(0)))) as ex:   # the oracle runs outside the GIL (ctypes)
            wants = list(ex.map(lambda k: ob.oracle_encode(imgs[order[k]].desc, plans[order[k]]), range(nchk)))
        base = 0
        for k in range(nchk):            # every DISTINCT image: byte-equal streams, segment by segment
            for i, w in enumerate(wants[k][0]):
                buf = C.create_string_buffer(max(1, lens[base + i]))
                L.lep_gpu_memcpy_d2h(g, buf, d_streams + offs[base + i], lens[base + i])
                assert buf.raw[: lens[base + i]] == w, "GPU stream %d of image %d differs from the oracle" % (i, k)
            base += len(plans[order[k]])
        parity = "streams==oracle(all %d distinct images) + exact round trip(all distinct images)" % nchk
        bins_per_image = sum(w[1] for w in wants) / nchk
    except ImportError:
        bins_per_image = None

    # ---- latency (configs[1]: ONE 4K image, then small batches): a thread segment is one serial chain whatever the batch, so
    # the chip is as slow for 1 image as for 1024 -- stated next to the reference's own per-image times (cpu_baseline below)
    latency = None
    if world == 1 and not args.no_extras:
        latency = {"unit": "ms, device-resident frames, launch + kernel + sync (best of 3)", "encode": {}, "decode": {}}
        for nb in (1, 8, 64):
            if nb > nimg:
                break
            ns = sum(len(plans[order[k]]) for k in range(nb))
            best_e = best_d = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                assert L.lep_gpu_encode_device(g, descs, nb, segs, ns, d_streams, offs, d_len, d_status, None) == 0
                L.lep_gpu_sync(g)
                t1 = time.perf_counter()
                assert L.lep_gpu_decode_device(g, dec_descs, nb, segs, ns, d_streams, offs, d_len, d_status, None) == 0
                L.lep_gpu_sync(g)
                t2 = time.perf_counter()
                best_e = min(best_e, t1 - t0); best_d = min(best_d, t2 - t1)
            latency["encode"]["%d_image%s" % (nb, "" if nb == 1 else "s")] = round(best_e * 1e3, 1)
            latency["decode"]["%d_image%s" % (nb, "" if nb == 1 else "s")] = round(best_d * 1e3, 1)
        t0 = time.perf_counter(); one = codec.compress(uniq[0]); t1 = time.perf_counter(); assert codec.decompress(one) == uniq[0]; t2 = time.perf_counter()
        latency["whole_file_host_to_host"] = {"compress_ms": round((t1 - t0) * 1e3, 1), "decompress_ms": round((t2 - t1) * 1e3, 1),
                                              "note": "lep_compress / lep_decompress of one 4K JPEG: host Huffman + PCIe + kernels + container"}

    for k in range(nimg):   # the resident frames are no longer needed; the end-to-end measurement wants the memory
        for c in range(imgs[order[k]].desc.ncomp):
            L.lep_gpu_free(g, descs[k].blocks[c]); L.lep_gpu_free(g, dec_descs[k].blocks[c])
    L.lep_gpu_free(g, d_streams)

    # PCIe- and host-inclusive companion figure (never `value`): JPEG files in host memory -> .lep files in host memory and
    # back through the batch pipeline (host split, GPU Huffman decode, GPU arithmetic coding, containers on the host pool; and
    # the mirror image).  Every rank runs it on its own files; rank 0 reports the aggregate.
    e2e = None
    e2e_local = {"e2e_bytes": 0.0, "e2e_c_s_max": 0.0, "e2e_d_s_max": 0.0}
    if not args.no_end_to_end:
        try:
            n_e2e = args.e2e_images if world == 1 else min(args.e2e_images, 1024)
            e2e = pipeline_figure(codec, [uniq[i % nuniq] for i in range(n_e2e)],
                                  "%d of the bench's 4K JPEGs per GPU, host memory -> host memory (lep_compress_batch / lep_decompress_batch), staging warm (second call)" % n_e2e)
            e2e_local = {"e2e_bytes": e2e["jpeg_MB"] * 1e6, "e2e_c_s_max": e2e["_cs"]["wall_s"], "e2e_d_s_max": e2e["_ds"]["wall_s"]}
        except Exception as e:   # the headline figure must not depend on it
            e2e = {"error": repr(e)[:300]}
    local = {"jpeg_bytes": jpeg_bytes, "images": nimg, "segments": nseg, "blocks": nblocks,
             "stream_bytes": stream_bytes, "elapsed_max": elapsed, "enc_ms_max": enc_ms, "dec_ms_max": dec_ms}
    local.update(e2e_local)
    agg = shard.aggregate(local, backend_device=("cuda:%d" % local_rank) if dist else None)
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return
    K = args.steps
    t = agg["elapsed_max"]
    mb = agg["jpeg_bytes"] / 1e6
    value = mb * K / t
    b_alg = 128.0 * agg["blocks"] / world + agg["stream_bytes"] / world      # per launch, one GPU
    enc_kernel_s = agg["enc_ms_max"] / K / 1e3
    dec_kernel_s = agg["dec_ms_max"] / K / 1e3
    dominant = "decode" if dec_kernel_s >= enc_kernel_s else "encode"
    dom_s = max(enc_kernel_s, dec_kernel_s)
    achieved = b_alg / dom_s / 1e9
    traffic = pmc_traffic(names.get(dominant, ""), args.images)
    out = {
        "metric": "encode+decode MB/s (JPEG bytes/sec), bit-exact round trip", "value": round(value, 3), "unit": "MB/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(t / K * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16/u8 (integer coder, no floating point)",
        "data": "synthetic (seeded PIL baseline JPEGs, q90 4:2:0; %d distinct per GPU replicated to %d)" % (nuniq, args.images),
        "config": {"workload": "%d x %dx%d 4:2:0 baseline JPEG corpus per GPU, %d thread segments, coefficient frames and streams resident in HBM"
                   % (args.images, args.width, args.height, int(agg["segments"] / world)),
                   "images_per_gpu": args.images, "segments_per_gpu": int(agg["segments"] / world), "jpeg_MB_per_step": round(mb, 3),
                   "parallelism": "image-sharded x%d, one wavefront per thread segment" % world, "parity": parity},
        "encode_MBps": round(mb / (agg["enc_ms_max"] / K / 1e3), 3), "decode_MBps": round(mb / (agg["dec_ms_max"] / K / 1e3), 3),
        "roofline": {"bound": "hbm", "kernel": names.get(dominant, dominant), "achieved": round(achieved, 4), "peak": 8000.0,
                     "unit": "GB/s", "frac": round(achieved / 8000.0, 7), "traffic": traffic,
                     "traffic_frac": round(traffic / dom_s / 8e12, 4) if traffic else None,   # measured HBM bytes / kernel time / 8 TB/s
                     "algorithmic_bytes_per_launch": int(b_alg), "kernel_ms": round(dom_s * 1e3, 3),
                     "encode_kernel_ms": round(enc_kernel_s * 1e3, 3), "decode_kernel_ms": round(dec_kernel_s * 1e3, 3),
                     "kernels": names, "bound_by": pmc_bound(names.get(dominant, "")),
                     "note": "frac = algorithmic bytes (128 B per block + stream bytes) / kernel time / 8 TB/s; traffic = HBM bytes from the memory-side request counters of the committed PMC passes (profiles/pmc_traffic.json), scaled to this batch; bound_by = what those passes say limits the kernel (instruction issue, not bandwidth: DESIGN.md 4)"},
    }
    if bins_per_image:
        bins_launch = bins_per_image * args.images
        out["bins_per_s"] = {"encode": round(bins_launch / enc_kernel_s / 1e6, 1), "decode": round(bins_launch / dec_kernel_s / 1e6, 1), "unit": "Mbins/s (per GPU, image 0's bin count x images)"}
    if e2e is not None:
        if "error" in e2e:
            out["end_to_end"] = e2e
        else:
            cs, ds = e2e.pop("_cs"), e2e.pop("_ds")
            emb = agg["e2e_bytes"] / 1e6
            out["end_to_end"] = dict(e2e, **{
                "compress_MBps": round(emb / agg["e2e_c_s_max"], 1), "decompress_MBps": round(emb / agg["e2e_d_s_max"], 1),
                "value": round(2 * emb / (agg["e2e_c_s_max"] + agg["e2e_d_s_max"]), 1), "n_gpus": world,
                "unit": "MB/s (JPEG bytes, compress + decompress; wall clock of the two C-ABI calls, max over ranks, bytes summed over ranks)",
                "rank0_h2d_GB": round((cs["h2d_bytes"] + ds["h2d_bytes"]) / 1e9, 2), "rank0_d2h_GB": round((cs["d2h_bytes"] + ds["d2h_bytes"]) / 1e9, 2),
                "rank0_host_pool_seconds": {"compress_parse": round(cs["parse_s"], 3), "compress_write": round(cs["write_s"], 3),
                                            "decompress_parse": round(ds["parse_s"], 3), "decompress_write": round(ds["write_s"], 3)},
                "note": "JPEG Huffman decode / re-encode on the GPU, host only splits files and writes containers; lep bytes == reference for the fixtures (tests)"})
    if latency:
        out["latency"] = latency
    if world == 1 and not args.no_extras:
        # secondary corpora through the same host-to-host pipeline: a photograph-like one (detail growing from top to bottom:
        # thread segments of equal compressed size then differ several-fold in blocks), BASELINE.json configs[2] (1024 x 1080p)
        # and configs[4] (4K progressive: Huffman layer on the host pool today)
        extras = {}
        for key, label, n, nu, kw in (
                ("skewed", "1024 x 4K 4:2:0 baseline, photograph-like (corpus.synth_jpeg skew=2), 16 distinct", 1024, 16, dict(width=3840, height=2160, skew=2.0)),
                ("c1080p", "1024 x 1080p 4:2:0 baseline (BASELINE.json configs[2]), 32 distinct", 1024, 32, dict(width=1920, height=1080)),
                ("progressive", "256 x 4K 4:2:0 progressive (BASELINE.json configs[4]; -allowprogressive), 8 distinct", 256, 8, dict(width=3840, height=2160, progressive=True))):
            try:
                w, h = kw.pop("width"), kw.pop("height")
                u = corpus.make_corpus(nu, w, h, 30001 + 1000 * len(extras), **kw)
                fig = pipeline_figure(codec, [u[i % nu] for i in range(n)], label)
                fig.pop("_cs"); fig.pop("_ds")
                extras[key] = fig
            except Exception as e:
                extras[key] = {"workload": label, "error": repr(e)[:300]}
        out["extra"] = extras
    if world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(uniq)
        if cb:
            out["cpu_baseline"] = cb
    print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
