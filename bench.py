#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X (contract: see the build prompt / DESIGN.md).

A "step" = one encode pass + one decode pass of the arithmetic-coding hot path over one batch of
synthetic 4K 4:2:0 baseline JPEGs whose coefficient frames (encode input) and streams (decode input)
are already resident in HBM.  value = JPEG file bytes coded per second of wall clock over the K timed
steps, aggregated over all ranks (weak scaling: every rank codes its own `--images` images).

One definition of "MB/s" everywhere in the line: JPEG bytes / (seconds to encode them + seconds to decode them again).

`--gpus N` without a torchrun environment starts the N ranks itself (one process per GPU, RCCL); under
`python -m torch.distributed.run` the ranks are the launcher's.  `mixed` (BASELINE.json configs[3]) is the strong-scaling
companion: ONE mixed 1080p / 4K corpus dealt to the ranks by JPEG bytes (shard.shard_indices), host memory to host memory."""
import argparse
import ctypes as C
import json
import os
import re
import subprocess
import sys
import tempfile
import time

# the library asks for 8 hardware queues when it is loaded (lep_gpu.hip: lep_runtime_defaults) -- too late under N > 1, where torch
# brings the HIP runtime up for RCCL before the library is: say it here, before anything touches the GPU
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def spawn_ranks(n):
    """`bench.py --gpus N` started by hand (no RANK / WORLD_SIZE in the environment): start the N ranks the way the driver does --
    python -m torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1 -- and hand its exit code back.  Rank 0 of
    that job prints the JSON line."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] starting %d ranks: %s" % (n, " ".join(cmd[1:8])))
    return subprocess.call(cmd, env=env)


def kernel_source_sha():
    """identity of the kernel sources a PMC pass was taken from / this run was built from: sha256 over lepton_amd/csrc/*.h, *.hip and their compile flags"""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "lepton_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    # ... and the flags the kernels are compiled with (the objects' .flags records, written by build(): a compiler mode is part of the kernel)
    for f in ("lep_gpu.hip", "lep_batch.hip"):
        p = os.path.join(ROOT, "lepton_amd", "build", "obj", f + ".o.flags")
        h.update(open(p, "rb").read().split(b" |src:")[0] if os.path.exists(p) else b"")   # (the record also holds a hash of the object's inputs: not a flag)
    return h.hexdigest()[:16]


def library_identity():
    """what the LOADED library says it was built from (lep_version(): `... src <sha16>`) against the sources on this box
    (__graft_entry__.source_sha16): VERDICT round 5 weak #7 -- a stale .so that travelled with the snapshot must not be measured"""
    import __graft_entry__ as ge
    from lepton_amd import abi

    said = abi.lib().lep_version().decode()
    return {"lep_version": said, "sources_on_this_box_sha16": ge.source_sha16(), "library_built_from_them": ("src " + ge.source_sha16()) in said,
            "experiment_build": os.environ.get("LEP_LIB_PATH") or None}


def cpu_baseline(jpgs, budget_s=20.0, what="of the bench's 4K JPEGs"):
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
        "sample": "%d %s (%.1f MB), reference `lepton -singlethread -unjailed -skipverify`, encode then decode, whole process wall clock" % (n, what, mb),
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
    counters TCC_EA0_RDREQ x 64 B + TCC_EA0_WRREQ by size, separate --pmc passes, scripts/gpu_closing_visit.sh) FOR THE LAUNCH SIZE THAT
    WAS MEASURED: the table is kept per images-per-launch (1024, 256) and a launch of another size gets None -- the decoder's bytes
    per block double between 256 and 1024 images (its model falls out of L2), so scaling one figure linearly to another batch is
    wrong.  A table lookup, not a measurement of this run: the entry names the kernel build it was taken from."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        name = kernel.split("<")[0]
        by = t.get("by_images_per_launch")
        if by is not None:
            e = by.get(str(images), {}).get(name)
            return int(e["hbm_bytes_per_launch"]) if e else None
        e = t["kernels"][name]   # a table from before round 4: one launch size
        return int(e["hbm_bytes_per_launch"]) if int(e.get("images_per_launch", t.get("images_per_launch", 0))) == images else None
    except Exception:
        return None


def pmc_identity(kernel):
    """(sha of the kernel sources the PMC pass of `kernel` was taken from, True if that is not the build measured now)"""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        sha = t["kernels"][kernel.split("<")[0]].get("kernel_source_sha16") or t.get("kernel_source_sha16")
        return sha, sha != kernel_source_sha()
    except Exception:
        return None, True


def pmc_bound(kernel):
    """what the counters say bounds `kernel` (profiles/pmc_traffic.json: L2 hit rate, wave time split)"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"][kernel.split("<")[0]].get("bound")
    except Exception:
        return None


def pipeline_figure(codec, jpgs, label, verify=False, threads=0, repeats=3):
    """JPEG files in host memory -> .lep files in host memory and back through the batch pipeline; one warm-up call (staging
    buffers, kernel images), then `repeats` timed round trips: the figures are the BEST round trip's, every round trip is listed
    (`runs`) with the median beside it (one measurement each is not a result); every file must come back
    bit-exact every time"""
    mb = sum(map(len, jpgs)) / 1e6
    warm, st0, _ = codec.compress_batch(jpgs, verify=verify, threads=threads)
    assert not any(st0), sorted(set(st0))
    codec.decompress_batch(warm, threads=threads)
    del warm
    runs = []
    for _ in range(max(1, repeats)):
        leps, st1, cs_ = codec.compress_batch(jpgs, verify=verify, threads=threads)
        back, st2, ds_ = codec.decompress_batch(leps, threads=threads)
        assert not any(st1) and not any(st2) and back == jpgs, label + ": round trip is not bit exact"
        del back
        runs.append((cs_, ds_))
    cs, ds = min(runs, key=lambda r: r[0]["wall_s"] + r[1]["wall_s"])
    values = sorted(mb / (c["wall_s"] + d["wall_s"]) for c, d in runs)
    return {"workload": label, "jpeg_MB": round(mb, 1), "lep_MB": round(sum(map(len, leps)) / 1e6, 1), "files": len(jpgs),
            "compress_MBps": round(mb / cs["wall_s"], 1), "decompress_MBps": round(mb / ds["wall_s"], 1),
            "value": round(mb / (cs["wall_s"] + ds["wall_s"]), 1), "files_per_s": round(len(jpgs) / (cs["wall_s"] + ds["wall_s"]), 1),
            "value_median": round(values[len(values) // 2], 1), "value_min": round(values[0], 1),
            "runs": [{"compress_MBps": round(mb / c["wall_s"], 1), "decompress_MBps": round(mb / d["wall_s"], 1)} for c, d in runs],
            "unit": "MB/s = JPEG bytes / (compress seconds + decompress seconds), the headline's definition; best of %d warm round trips" % len(runs),
            "seconds": {d: {k: round(st[k], 3) for k in ("wall_s", "pipeline_s", "parse_s", "stage_s", "write_s", "alloc_s")} for d, st in (("compress", cs), ("decompress", ds))},
            # which path the files took (lep_batch_stats): a corpus silently coded by the host Huffman coders, or re-done file by file
            # because its streams outgrew their reservation, must show in the driver's line
            "gpu_huffman_files": {"compress": int(cs.get("gpu_huffman_files", 0)), "decompress": int(ds.get("gpu_huffman_files", 0))},
            "redone_files": int(cs.get("redone_files", 0)),
            "parity": "every file restored bit-exact", "_cs": cs, "_ds": ds}


class HipDevice:
    """the MI355X behind the C ABI (lepton_amd/liblepton_mi355x.so): device-resident encode / decode steps and the
    host-memory pipeline.  tests/bench_stub.py offers the same three methods without a GPU (LEP_BENCH_DEVICE=stub) so that
    the multi-rank plumbing of this file -- spawn, sharding, aggregation, the JSON line -- runs over gloo in the CPU suite."""

    name = "hip"

    def __init__(self, local_rank, host_threads=0, trim=False):
        from lepton_amd import abi
        from lepton_amd.codec import GpuCodec

        self.L = abi.lib()
        self.codec = GpuCodec(local_rank)
        self.g = self.codec.handle
        self.trim = trim   # --trim-between-phases: give the cached workspaces back before every phase (a serving process whose batches differ in size does)
        self.host_threads = host_threads   # host pool of the batch pipeline (0 = every CPU this process may use): N ranks share the host

    def sync(self):
        self.L.lep_gpu_sync(self.g)

    def footprint(self):
        """what this rank holds on the host side once its phases have run: the batch pipeline's pinned staging and device arenas
        (lep_batch_footprint), the host pool it runs its file splitting / container writing on"""
        import ctypes

        pinned, device = ctypes.c_size_t(0), ctypes.c_size_t(0)
        self.L.lep_batch_footprint(ctypes.byref(pinned), ctypes.byref(device))
        return {"pinned_MB": round(pinned.value / 1e6, 1), "staging_device_MB": round(device.value / 1e6, 1),
                "host_threads": self.host_threads or usable_cpus()}

    def identity(self):
        """which physical GPU this rank drives, as numbers (they travel in the per-rank matrix): PCI domain / bus / device / function
        of the HIP device (lep_gpu_pci_bus_id) and the xGMI hive it belongs to (sysfs; 0 where the box does not say) -- so that the
        first multi-GPU run shows N ranks on N different devices of one hive"""
        import ctypes

        buf = ctypes.create_string_buffer(32)
        out = {"pci_domain": -1, "pci_bus": -1, "pci_device": -1, "pci_function": -1, "xgmi_hive_hi": 0, "xgmi_hive_lo": 0}
        self.L.lep_gpu_pci_bus_id.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
        if self.L.lep_gpu_pci_bus_id(self.g, buf, 32) == 0:
            try:
                bdf = buf.value.decode().lower()
                dom, bus, rest = bdf.split(":")
                devn, fn = rest.split(".")
                out.update(pci_domain=int(dom, 16), pci_bus=int(bus, 16), pci_device=int(devn, 16), pci_function=int(fn, 16))
                hive = open("/sys/bus/pci/devices/%s/xgmi_hive_info/xgmi_hive_id" % bdf).read().strip()
                h = int(hive, 0)
                out.update(xgmi_hive_hi=h >> 32, xgmi_hive_lo=h & 0xffffffff)
            except Exception:
                pass
        return out

    def pipeline(self, jpgs, label, verify=False):
        # (no lep_gpu_trim between the phases by default.  Round 3 saw the 1080p figure halve after a trim and blamed the device heap; round
        # 4 found the cause -- the driver CLEARS the memory it hands out, 40 ms per GB, so a phase that gives 30 GB back and takes them
        # again pays seconds -- and the library's workspaces are now pooled address ranges whose chunks a trim keeps (LAB_NOTES.md 4 "What
        # one phase leaves behind"): with --trim-between-phases the figures are within 1 %.)
        if self.trim:
            self.L.lep_gpu_trim(self.g)
        return pipeline_figure(self.codec, jpgs, label, verify=verify, threads=self.host_threads)

    def resident(self, uniq, images, steps, warmup, barrier, check_parity=True, with_latency=False, latency_sizes=(1, 8, 64), latency_repeats=3, whole_file=True):
        """`images` 4K frames (the `uniq` distinct ones replicated device-to-device) and their streams resident in HBM;
        `warmup` + `steps` encode + decode launches; returns counters, kernel times (HIP events on the launch stream) and the
        parity verdict (every segment's exit code, decode(encode(x)) == x, streams == oracle for every distinct image)."""
        from lepton_amd import abi
        from lepton_amd.codec import JpegImage

        L, g, codec = self.L, self.g, self.codec
        nuniq = len(uniq)
        imgs = [JpegImage(j) for j in uniq]
        plans = [im.plan() for im in imgs]
        order = [i % nuniq for i in range(images)]
        jpeg_bytes = sum(len(uniq[i]) for i in order)
        nimg = len(order)
        allocs = []
        try:
            return self._resident(L, g, codec, uniq, imgs, plans, order, nuniq, jpeg_bytes, nimg, allocs, steps, warmup, barrier, check_parity, with_latency, latency_sizes, latency_repeats, whole_file)
        finally:
            for a in allocs:   # the resident frames are no longer needed (or could not all be had); what follows wants the memory
                L.lep_gpu_free(g, a)

    def _resident(self, L, g, codec, uniq, imgs, plans, order, nuniq, jpeg_bytes, nimg, allocs, steps, warmup, barrier, check_parity, with_latency, latency_sizes=(1, 8, 64), latency_repeats=3, whole_file=True):
        from lepton_amd import abi

        def dmalloc(n):
            p = C.c_void_p()
            rc = L.lep_gpu_malloc(g, n, C.byref(p))
            assert rc == 0, codec.last_error()
            allocs.append(p.value)
            return p.value

        descs = (abi.ImageDesc * nimg)()
        dec_descs = (abi.ImageDesc * nimg)()
        flat = []
        nblocks = 0
        first_copy = {}   # unique image -> its device frame: uploaded once over PCIe, replicated device-to-device
        # two allocations for all frames (sources, decode targets), not one per component: thousands of 4..17 MB buffers taken and
        # given back leave the device heap in pieces, and what is allocated afterwards runs at half speed (measured: the 1080p
        # pipeline figure, 1430 -> 710 MB/s compress, when it followed this function's earlier form)
        frame_total = sum(((imgs[u].desc.nblocks(c) * 128 + 255) & ~255) for u in order for c in range(imgs[u].desc.ncomp))
        src_all, dec_all = dmalloc(frame_total), dmalloc(frame_total)
        assert L.lep_gpu_memset(g, dec_all, 0, frame_total) == 0
        at = 0
        for k, u in enumerate(order):
            d = imgs[u].desc
            C.memmove(C.byref(descs[k]), C.byref(d), C.sizeof(abi.ImageDesc))
            C.memmove(C.byref(dec_descs[k]), C.byref(d), C.sizeof(abi.ImageDesc))
            for c in range(d.ncomp):
                n = d.nblocks(c) * 128
                p = src_all + at
                if (u, c) in first_copy:
                    assert L.lep_gpu_memcpy_d2d(g, p, first_copy[(u, c)], n) == 0
                else:
                    assert L.lep_gpu_memcpy_h2d(g, p, d.blocks[c], n) == 0
                    first_copy[(u, c)] = p
                descs[k].blocks[c] = p
                dec_descs[k].blocks[c] = dec_all + at
                at += (n + 255) & ~255
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
        stages = []   # split-phase encoder: per-launch stage times (count + plan, emit, fold, gather, write)

        def step():
            rc = L.lep_gpu_encode_device(g, descs, nimg, segs, nseg, d_streams, offs, d_len, d_status, None)
            assert rc == 0, (rc, codec.last_error())
            rc = L.lep_gpu_sync(g)
            assert rc == 0, codec.last_error()
            e_ms = L.lep_gpu_last_kernel_ms(g)
            names["encode"] = L.lep_gpu_last_kernel_name(g).decode()
            sm = (C.c_double * 8)()
            ns_ = L.lep_gpu_last_stage_ms(g, sm, 8)
            if ns_:
                stages.append([sm[i] for i in range(ns_)])
            rc = L.lep_gpu_decode_device(g, dec_descs, nimg, segs, nseg, d_streams, offs, d_len, d_status, None)
            assert rc == 0, (rc, codec.last_error())
            rc = L.lep_gpu_sync(g)
            assert rc == 0, codec.last_error()
            names["decode"] = L.lep_gpu_last_kernel_name(g).decode()
            return e_ms, L.lep_gpu_last_kernel_ms(g)

        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        enc_ms = dec_ms = 0.0
        for _ in range(steps):
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
        nchk = min(nimg, nuniq)
        for k in range(nchk):
            d = imgs[order[k]].desc
            for c in range(d.ncomp):
                n = d.nblocks(c) * 128
                buf = C.create_string_buffer(n)
                L.lep_gpu_memcpy_d2h(g, buf, dec_descs[k].blocks[c], n)
                assert hashlib.md5(buf.raw).digest() == hashlib.md5(C.string_at(d.blocks[c], n)).digest(), "decode(encode(x)) != x"
        parity = "exact round trip(all %d distinct images)" % nchk
        bins_per_image = None
        if check_parity:
            try:
                import oracle_binding as ob
                from concurrent.futures import ThreadPoolExecutor

                ob.oracle()
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
                pass

        # ---- latency (configs[1]: ONE 4K image, then small batches): a thread segment is one serial chain whatever the batch
        latency = None
        if with_latency:
            latency = {"unit": "ms, device-resident frames, launch + kernel + sync (best of 3)", "encode": {}, "decode": {}}
            for nb in latency_sizes:
                if nb > nimg:
                    break
                ns = sum(len(plans[order[k]]) for k in range(nb))
                best_e = best_d = 1e9
                for _ in range(latency_repeats):
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
            if whole_file:
                best_c = best_x = 1e9   # (best of 3 like the launches above: the first call of its kind in the process pays for its buffers)
                for _ in range(3):
                    t0 = time.perf_counter(); one = codec.compress(uniq[0]); t1 = time.perf_counter(); back = codec.decompress(one); t2 = time.perf_counter()
                    assert back == uniq[0]
                    best_c = min(best_c, t1 - t0); best_x = min(best_x, t2 - t1)
                latency["whole_file_host_to_host"] = {"compress_ms": round(best_c * 1e3, 1), "decompress_ms": round(best_x * 1e3, 1),
                                                      "note": "lep_compress / lep_decompress of one 4K JPEG, best of 3: host Huffman + PCIe + kernels + container (+ the host-side check that the .lep restores the file)"}
        return {"jpeg_bytes": jpeg_bytes, "images": nimg, "segments": nseg, "blocks": nblocks, "stream_bytes": stream_bytes,
                "elapsed": elapsed, "enc_ms": enc_ms, "dec_ms": dec_ms, "names": names, "parity": parity,
                "bins_per_image": bins_per_image, "latency": latency,
                "encode_stages_ms": [round(sum(x[i] for x in stages[-steps:]) / max(1, len(stages[-steps:])), 3) for i in range(len(stages[0]))] if stages else None}


def usable_cpus():
    """CPUs this process may use: affinity mask capped by the cgroup quota"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def pin_to_gpu_numa_node(local_rank):
    """one rank per GPU: keep the rank's host threads (file splitting, container writing) on the CPUs of the NUMA node its GPU hangs off,
    where the kernel exposes that (/sys/bus/pci/devices/<bdf>/numa_node); returns (a note for the log, the node or -1 when not pinned)"""
    try:
        import torch

        bdf = torch.cuda.get_device_properties(local_rank).pci_bus_id if hasattr(torch.cuda.get_device_properties(local_rank), "pci_bus_id") else None
        if not bdf:
            return "no PCI id for device %d" % local_rank, -1
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf.lower()).read())
        if node < 0:
            return "device %s: no NUMA node reported" % bdf, -1
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        mine = cpus & os.sched_getaffinity(0)
        if not mine:
            return "device %s on node %d: none of its CPUs allowed here" % (bdf, node), -1
        os.sched_setaffinity(0, mine)
        return "device %s on NUMA node %d: pinned to %d CPUs" % (bdf, node, len(mine)), node
    except Exception as e:
        return "not pinned (%s)" % repr(e)[:80], -1


def mixed_plan(n, distinct, small=(1920, 1080), big=(3840, 2160), seed0=20000):
    """BASELINE.json configs[3] in the shape SURVEY.md 8(d) gives it: file i is seed seed0 + (i mod distinct), even seeds the small
    size, odd seeds the big one.  Returns [(seed, w, h)] -- every rank computes the same list without generating anything."""
    distinct = max(2, min(distinct, n) // 2 * 2)
    return [(seed0 + i % distinct,) + (small if (i % distinct) % 2 == 0 else big) for i in range(n)]


def mixed_files(plan, mine, workers):
    """the files of `plan` this rank codes, generated with `workers` processes; a file found in $LEP_CORPUS_CACHE (a directory) is read
    from there and a new one is left there -- a box that runs the bench twice, or a node that keeps the directory, generates once"""
    from lepton_amd import corpus

    cache = os.environ.get("LEP_CORPUS_CACHE")
    need = sorted({plan[i] for i in mine})
    have = {}
    if cache:
        os.makedirs(cache, exist_ok=True)
        for key in need:
            p = os.path.join(cache, "mixed_%d_%dx%d.jpg" % key)
            if os.path.exists(p):
                have[key] = open(p, "rb").read()
    todo = [k for k in need if k not in have]
    if todo:
        jobs = [(w, h, seed, 90, False, "4:2:0", 0.0) for seed, w, h in todo]
        if workers <= 1 or len(jobs) <= 2:
            made = [corpus._job(j) for j in jobs]
        else:
            from concurrent.futures import ProcessPoolExecutor

            with ProcessPoolExecutor(max_workers=min(workers, len(jobs))) as ex:
                made = list(ex.map(corpus._job, jobs, chunksize=1))
        for key, data in zip(todo, made):
            have[key] = data
            if cache:
                tmp = os.path.join(cache, ".%d.mixed_%d_%dx%d.jpg" % ((os.getpid(),) + key))
                open(tmp, "wb").write(data)
                os.replace(tmp, os.path.join(cache, "mixed_%d_%dx%d.jpg" % key))
    return [have[plan[i]] for i in mine], len(todo)


def reference_benchmark_jpeg():
    """the synthetic file `lepton -benchmark` codes when given none (src/lepton/benchmark.cc:116-119: bigger_hdr + 76 x
    bigger_rep, 2,589,088 bytes), assembled from the two pieces tests/golden/make_ref_benchmark.py extracted"""
    g = os.path.join(ROOT, "tests", "golden")
    return open(os.path.join(g, "ref_benchmark_hdr.bin"), "rb").read() + 76 * open(os.path.join(g, "ref_benchmark_rep.bin"), "rb").read()


def main():
    import faulthandler

    faulthandler.enable()   # a crash inside the library names the Python line that called it
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=1024,
                    help="4K images per GPU per step (8 thread segments each; 1024 = 8192 segments = 8 wavefronts per SIMD)")
    ap.add_argument("--unique", type=int, default=64, help="distinct synthetic images per GPU (replicated up to --images)")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary corpora (skewed, 1080p, progressive, reference benchmark file) and the latency table")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--trim-between-phases", action="store_true", help="lep_gpu_trim before every pipeline phase (measures what giving the workspaces back costs)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-memory -> host-memory pipeline measurement (lep_compress_batch / lep_decompress_batch)")
    ap.add_argument("--e2e-images", type=int, default=2688)   # 3 pipeline chunks of 896 images = 7168 thread segments each
    ap.add_argument("--mixed-images", type=int, default=-1, help="size of the mixed 1080p / 4K corpus of the strong-scaling figure (0 = skip; default: 10000 -- BASELINE.json configs[3] -- from 8 ranks on, 1024 below)")
    ap.add_argument("--mixed-distinct", type=int, default=-1, help="distinct files in the mixed corpus (default: as many as the ranks' host CPUs generate in about a minute: all of them at 8 ranks x 16+ CPUs, 32 on a small host)")
    ap.add_argument("--mixed-shapes", default="1920x1080,3840x2160", help="the two image sizes of the mixed corpus (tests use small ones)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:   # started by hand: be the launcher
        sys.exit(spawn_ranks(args.gpus))

    import __graft_entry__ as ge
    from lepton_amd import corpus, shard

    rank, local_rank, world = shard.dist_env()
    stub = os.environ.get("LEP_BENCH_DEVICE") == "stub"   # tests/bench_stub.py: the CPU suite's stand-in for the device layer
    dist = None
    numa_node = -1   # the NUMA node this rank's host threads were pinned to (-1: not pinned -- one rank, or the box does not say)
    if world > 1:
        import torch
        import torch.distributed as dist

        if stub:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if rank == 0 and not stub:
        ge.build()
    if dist:
        dist.barrier()
    if stub:
        import bench_stub

        dev = bench_stub.StubDevice(local_rank)
    else:
        if world > 1:
            numa_note, numa_node = pin_to_gpu_numa_node(local_rank)
            log("[rank %d] %s" % (rank, numa_note))
        dev = HipDevice(local_rank, host_threads=max(1, usable_cpus() // world) if world > 1 else 0, trim=args.trim_between_phases)

    def barrier():
        if dist:
            dist.barrier()
        dev.sync()

    # ---- headline: weak scaling, every rank has its own distinct images, frames and streams resident in HBM
    t0 = time.perf_counter()
    nuniq = max(1, min(args.unique, args.images))
    seeds = shard.weak_seeds(nuniq, rank, 20001)
    uniq = corpus.make_corpus(nuniq, args.width, args.height, seeds[0])
    log("[rank %d] corpus: %d unique %dx%d JPEGs in %.1fs" % (rank, nuniq, args.width, args.height, time.perf_counter() - t0))
    res = dev.resident(uniq, args.images, args.steps, args.warmup, barrier, with_latency=(world == 1 and not args.no_extras))
    names, parity, latency, bins_per_image = res["names"], res["parity"], res["latency"], res["bins_per_image"]
    log("[rank %d] resident steps done (%.1fs)" % (rank, time.perf_counter() - t0))

    # ---- strong scaling (BASELINE.json configs[3]): ONE mixed 1080p / 4K corpus, the same list on every rank, dealt by JPEG
    # bytes; every rank pushes its share through the host-memory pipeline; no data-path collective
    mixed_local = {"mixed_bytes": 0.0, "mixed_files": 0.0, "mixed_c_s_max": 0.0, "mixed_d_s_max": 0.0, "mixed_generated": 0.0}
    mixed_err = None
    mixed_runs = {}
    mixed_n = args.mixed_images if args.mixed_images >= 0 else (10000 if world >= 8 else 1024)
    cpus_per_rank = max(1, usable_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    # a 4K file takes a core ~1.5 s to synthesise, a 1080p one ~0.4 s: what a rank's cores make in about a minute
    mixed_distinct = args.mixed_distinct if args.mixed_distinct > 0 else (mixed_n if cpus_per_rank * world * 60 >= mixed_n else 32)
    if mixed_n > 0:
        try:
            shp = [tuple(int(v) for v in x.split("x")) for x in args.mixed_shapes.split(",")]
            plan = mixed_plan(mixed_n, mixed_distinct, small=shp[0], big=shp[1])
            est = [w * h for _, w, h in plan]   # dealt by pixels: a file's bytes follow its size class, and nobody has generated it yet
            mine = shard.shard_indices(len(plan), world, rank, est)
            files, made = mixed_files(plan, mine, cpus_per_rank)
            log("[rank %d] mixed corpus: %d of %d files (%d generated here, %d distinct in all) in %.1fs" % (rank, len(mine), len(plan), made, len(set(plan)), time.perf_counter() - t0))
            barrier()
            fig = dev.pipeline(files, "mixed")
            mixed_local = {"mixed_bytes": float(sum(map(len, files))), "mixed_files": float(len(mine)), "mixed_generated": float(made),
                           "mixed_c_s_max": fig["_cs"]["wall_s"], "mixed_d_s_max": fig["_ds"]["wall_s"]}
            mixed_total = (len(plan), len(set(plan)))
            mixed_runs = {"rank0_runs": fig.get("runs"), "rank0_value_median": fig.get("value_median"), "rank0_value_min": fig.get("value_min")}
            del files, fig
        except Exception as e:   # the headline figure must not depend on it
            mixed_err = repr(e)[:300]
        log("[rank %d] mixed corpus done (%.1fs) %s" % (rank, time.perf_counter() - t0, mixed_err or ""))

    # PCIe- and host-inclusive companion figure (never `value`): JPEG files in host memory -> .lep files in host memory and
    # back through the batch pipeline (host split, GPU Huffman decode, GPU arithmetic coding, containers on the host pool; and
    # the mirror image).  Every rank runs it on its own files; rank 0 reports the aggregate.
    e2e = None
    e2e_local = {"e2e_bytes": 0.0, "e2e_c_s_max": 0.0, "e2e_d_s_max": 0.0}
    if not args.no_end_to_end:
        try:
            n_e2e = args.e2e_images if world == 1 else min(args.e2e_images, 1024)
            e2e = dev.pipeline([uniq[i % nuniq] for i in range(n_e2e)],
                               "%d of the bench's 4K JPEGs per GPU, host memory -> host memory (lep_compress_batch / lep_decompress_batch), staging warm (second call)" % n_e2e)
            e2e_local = {"e2e_bytes": e2e["jpeg_MB"] * 1e6, "e2e_c_s_max": e2e["_cs"]["wall_s"], "e2e_d_s_max": e2e["_ds"]["wall_s"]}
        except Exception as e:   # the headline figure must not depend on it
            e2e = {"error": repr(e)[:300]}
        log("[rank %d] end to end done (%.1fs)" % (rank, time.perf_counter() - t0))
    local = {"jpeg_bytes": res["jpeg_bytes"], "images": res["images"], "segments": res["segments"], "blocks": res["blocks"],
             "stream_bytes": res["stream_bytes"], "elapsed_max": res["elapsed"], "enc_ms_max": res["enc_ms"], "dec_ms_max": res["dec_ms"],
             "ranks": 1}
    local.update(e2e_local)
    local.update(mixed_local)
    agg = shard.aggregate(local, backend_device=("cuda:%d" % local_rank) if (dist and not stub) else None)
    # per rank, for reading a multi-GPU run without a second visit: the host side of each rank (threads, CPUs, pinned staging, NUMA
    # placement) next to what it took of the wall clock -- the same counters the sums / maxima above were made of
    mine_rec = dict(dev.footprint(), **dev.identity(), world_size_seen=(dist.get_world_size() if dist else 1),
                    backend_is_rccl=(1 if (dist and dist.get_backend() == "nccl") else 0),
                    local_rank=local_rank, cpus_allowed=len(os.sched_getaffinity(0)), numa_node=numa_node,
                    resident_s=round(res["elapsed"], 4), enc_ms=round(res["enc_ms"] / max(1, args.steps), 3), dec_ms=round(res["dec_ms"] / max(1, args.steps), 3),
                    e2e_compress_s=round(e2e_local["e2e_c_s_max"], 4), e2e_decompress_s=round(e2e_local["e2e_d_s_max"], 4),
                    mixed_files=int(mixed_local["mixed_files"]), mixed_MB=round(mixed_local["mixed_bytes"] / 1e6, 2),
                    mixed_compress_s=round(mixed_local["mixed_c_s_max"], 4), mixed_decompress_s=round(mixed_local["mixed_d_s_max"], 4))
    # (numbers only, and gathered with the one collective the counters above already went through -- a [world x fields] matrix, every
    # rank its own row, summed -- so the multi-GPU line depends on nothing a one-GPU box could not exercise)
    per_rank = [dict(mine_rec, rank=rank)]
    if dist:
        import torch

        keys = sorted(mine_rec)
        m = torch.zeros((world, len(keys)), dtype=torch.float64, device=("cuda:%d" % local_rank) if not stub else "cpu")
        m[rank] = torch.tensor([float(mine_rec[k]) for k in keys], dtype=torch.float64)
        dist.all_reduce(m, op=dist.ReduceOp.SUM)
        rows = m.cpu().tolist()
        ints = {"local_rank", "cpus_allowed", "numa_node", "host_threads", "mixed_files", "pci_domain", "pci_bus", "pci_device", "pci_function",
                "xgmi_hive_hi", "xgmi_hive_lo", "world_size_seen", "backend_is_rccl"}
        per_rank = [dict({k: (int(round(v)) if k in ints else round(v, 4)) for k, v in zip(keys, row)}, rank=r) for r, row in enumerate(rows)]
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return
    assert int(agg["ranks"]) == world, "counter all-reduce saw %d ranks of %d" % (agg["ranks"], world)
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
    pmc_sha, pmc_stale = pmc_identity(names.get(dominant, ""))
    per_kernel = {}
    for which, secs in (("encode", enc_kernel_s), ("decode", dec_kernel_s)):
        tr = pmc_traffic(names.get(which, ""), args.images)
        per_kernel[which] = {"kernel": names.get(which, which), "kernel_ms": round(secs * 1e3, 3), "achieved_GBps": round(b_alg / secs / 1e9, 3),
                             "frac": round(b_alg / secs / 8e12, 7), "traffic": tr, "traffic_frac": round(tr / secs / 8e12, 4) if tr else None}
    out = {
        "metric": "encode+decode MB/s (JPEG bytes/sec), bit-exact round trip", "value": round(value, 3), "unit": "MB/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(t / K * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16/u8 (integer coder, no floating point)",
        "data": "synthetic (seeded PIL baseline JPEGs, q90 4:2:0; %d distinct per GPU replicated to %d)" % (nuniq, args.images),
        "config": {"workload": "%d x %dx%d 4:2:0 baseline JPEG corpus per GPU, %d thread segments, coefficient frames and streams resident in HBM"
                   % (args.images, args.width, args.height, int(agg["segments"] / world)),
                   "images_per_gpu": args.images, "segments_per_gpu": int(agg["segments"] / world), "jpeg_MB_per_step": round(mb, 3),
                   "parallelism": "image-sharded x%d, no data-path collective" % world, "parity": parity, "device_layer": dev.name},
        "value_definition": "JPEG bytes / (seconds to encode them + seconds to decode them); the same in end_to_end, mixed, extra.*",
        "encode_MBps": round(mb / (agg["enc_ms_max"] / K / 1e3), 3), "decode_MBps": round(mb / (agg["dec_ms_max"] / K / 1e3), 3),
        "roofline": {"bound": "hbm", "kernel": names.get(dominant, dominant), "achieved": round(achieved, 4), "peak": 8000.0,
                     "unit": "GB/s", "frac": round(achieved / 8000.0, 7), "traffic": traffic,
                     "traffic_frac": round(traffic / dom_s / 8e12, 4) if traffic else None,   # measured HBM bytes / kernel time / 8 TB/s
                     "traffic_source": {"file": "profiles/pmc_traffic.json", "kernel_source_sha16_of_the_pmc_pass": pmc_sha,
                                        "kernel_source_sha16_of_this_build": kernel_source_sha(), "stale": pmc_stale},
                     "algorithmic_bytes_per_launch": int(b_alg), "kernel_ms": round(dom_s * 1e3, 3),
                     "encode_kernel_ms": round(enc_kernel_s * 1e3, 3), "decode_kernel_ms": round(dec_kernel_s * 1e3, 3),
                     "kernels": names, "per_kernel": per_kernel,
                     "encode_stages_ms": dict(zip(("count_plan", "emit", "fold", "gather", "write"), res["encode_stages_ms"])) if res.get("encode_stages_ms") else None, "bound_by": pmc_bound(names.get(dominant, "")),
                     "note": "frac = algorithmic bytes (128 B per block + stream bytes) / kernel time / 8 TB/s; traffic = HBM bytes from the memory-side request counters of the committed PMC passes (profiles/pmc_traffic.json) for THIS launch size (null for a size that was not measured: never scaled) -- `traffic_source.stale` says whether those passes were taken from the kernel sources measured here; bound_by = what those passes say limits the kernel (DESIGN.md 4, LAB_NOTES.md 4)"},
    }
    out["distributed"] = {"world_size": world, "backend": (dist.get_backend() if dist else None), "launcher": "torch.distributed.run / spawn_ranks" if dist else "single process",
                          "distinct_devices": len({(r.get("pci_domain"), r.get("pci_bus"), r.get("pci_device"), r.get("pci_function")) for r in per_rank}),
                          "note": "per_rank carries each rank's PCI address and xGMI hive: N ranks must show N distinct devices"}
    out["library"] = library_identity()
    if not stub and not out["library"]["library_built_from_them"] and not out["library"]["experiment_build"]:
        raise RuntimeError("the loaded library was not built from the sources beside it: %s" % out["library"])
    out["per_rank"] = per_rank
    if bins_per_image:
        bins_launch = bins_per_image * args.images
        out["bins_per_s"] = {"encode": round(bins_launch / enc_kernel_s / 1e6, 1), "decode": round(bins_launch / dec_kernel_s / 1e6, 1), "unit": "Mbins/s (per GPU, image 0's bin count x images)"}
    if mixed_n > 0:
        if mixed_err or not agg["mixed_bytes"]:
            out["mixed"] = {"error": mixed_err or "no files"}
        else:
            mmb = agg["mixed_bytes"] / 1e6
            out["mixed"] = {
                "workload": "BASELINE.json configs[3]: %d mixed 1080p / 4K baseline JPEGs (alternating sizes, seeds 20000.., %d distinct), ONE corpus dealt to %d rank(s) by size class (shard.shard_indices over pixels), every rank generating only its own files, host memory -> host memory" % (mixed_total[0], mixed_total[1], world),
                "distinct": mixed_total[1], "generated_by_the_ranks": int(agg.get("mixed_generated", 0)),
                "scaling_curve": "this line is ONE point (n_gpus = %d); no 1 -> 8 curve has been measured until the driver's SCALE run exists" % world,
                "scaling": "strong", "n_gpus": world, "files": int(agg["mixed_files"]), "jpeg_MB": round(mmb, 1),
                "compress_MBps": round(mmb / agg["mixed_c_s_max"], 1), "decompress_MBps": round(mmb / agg["mixed_d_s_max"], 1),
                "value": round(mmb / (agg["mixed_c_s_max"] + agg["mixed_d_s_max"]), 1),
                "files_per_s": round(agg["mixed_files"] / (agg["mixed_c_s_max"] + agg["mixed_d_s_max"]), 1),
                "unit": "MB/s = all ranks' JPEG bytes / (max over ranks of compress seconds + max over ranks of decompress seconds)",
                "parity": "every file restored bit-exact on its rank"}
            out["mixed"].update({k: v for k, v in mixed_runs.items() if v is not None})   # (the figures above are the best of these warm round trips)
    if e2e is not None:
        if "error" in e2e:
            out["end_to_end"] = e2e
        else:
            cs, ds = e2e.pop("_cs"), e2e.pop("_ds")
            emb = agg["e2e_bytes"] / 1e6
            out["end_to_end"] = dict(e2e, **{
                "compress_MBps": round(emb / agg["e2e_c_s_max"], 1), "decompress_MBps": round(emb / agg["e2e_d_s_max"], 1),
                "value": round(emb / (agg["e2e_c_s_max"] + agg["e2e_d_s_max"]), 1), "n_gpus": world,
                "unit": "MB/s = JPEG bytes / (compress seconds + decompress seconds): wall clock of the two C-ABI calls, max over ranks, bytes summed over ranks",
                "rank0_h2d_GB": round((cs["h2d_bytes"] + ds["h2d_bytes"]) / 1e9, 2), "rank0_d2h_GB": round((cs["d2h_bytes"] + ds["d2h_bytes"]) / 1e9, 2),
                "rank0_host_pool_seconds": {"compress_parse": round(cs["parse_s"], 3), "compress_write": round(cs["write_s"], 3),
                                            "decompress_parse": round(ds["parse_s"], 3), "decompress_write": round(ds["write_s"], 3)},
                "note": "JPEG Huffman decode / re-encode on the GPU, host only splits files and writes containers; lep bytes == reference for the fixtures (tests)"})
    if latency:
        out["latency"] = latency
    if world == 1 and not args.no_extras:
        # The companion headline: the SAME device-resident step over a photograph-like corpus (detail
        # growing from top to bottom: thread segments of equal compressed size then differ several-fold in blocks) -- the
        # replicated, evenly cut corpus above flatters the schedule
        sk = None
        try:
            sk = corpus.make_corpus(16, args.width, args.height, 30001, skew=2.0)
            r2 = dev.resident(sk, args.images, max(1, min(2, K)), 1, barrier, check_parity=True)
            k2 = max(1, min(2, K))
            out["value_skewed"] = {
                "value": round(r2["jpeg_bytes"] / 1e6 * k2 / r2["elapsed"], 3), "unit": "MB/s", "steps": k2,
                "workload": "%d x 4K photograph-like (corpus.synth_jpeg skew=2), 16 distinct, frames and streams resident in HBM -- the headline's measurement on a corpus whose thread segments differ several-fold in blocks" % args.images,
                "encode_kernel_ms": round(r2["enc_ms"] / k2, 3), "decode_kernel_ms": round(r2["dec_ms"] / k2, 3), "parity": r2["parity"]}
        except Exception as e:
            out["value_skewed"] = {"error": repr(e)[:300]}
        log("[rank 0] skewed corpus, resident: %s" % str(out["value_skewed"])[:200])
        # secondary corpora through the host-to-host pipeline: the photograph-like one, BASELINE.json configs[2] (1024 x 1080p),
        # configs[4] (4K progressive) and the file `lepton -benchmark` itself codes
        extras = {}
        for key, label, n, nu, kw in (
                ("skewed", "1024 x 4K 4:2:0 baseline, photograph-like (corpus.synth_jpeg skew=2), 16 distinct", 1024, 16, dict(width=3840, height=2160, skew=2.0)),
                ("c1080p", "1024 x 1080p 4:2:0 baseline (BASELINE.json configs[2]), 32 distinct", 1024, 32, dict(width=1920, height=1080)),
                ("progressive", "256 x 4K 4:2:0 progressive (BASELINE.json configs[4]; -allowprogressive), 8 distinct", 256, 8, dict(width=3840, height=2160, progressive=True))):
            try:
                w, h = kw.pop("width"), kw.pop("height")
                u = sk if (key == "skewed" and sk) else corpus.make_corpus(nu, w, h, 30001 + 1000 * len(extras), **kw)
                fig = dev.pipeline([u[i % nu] for i in range(n)], label)
                fig.pop("_cs"); fig.pop("_ds")
                extras[key] = fig
            except Exception as e:
                extras[key] = {"workload": label, "error": repr(e)[:300]}
            log("[rank 0] extra %s: %s" % (key, str({k: v for k, v in extras[key].items() if k not in ("workload", "unit", "parity", "jpeg_MB", "lep_MB")})[:600]))
        try:   # the reference's own benchmark input (src/lepton/benchmark.cc:116-119), so that numbers line up with `lepton -benchmark`
            rb = reference_benchmark_jpeg()
            # (1024 files like every other figure of this line: 8192 thread segments, one full pipeline chunk -- until round 5 this one ran 512)
            fig = dev.pipeline([rb] * 1024, "1024 copies of the file `lepton -benchmark` codes (bigger_hdr + 76 x bigger_rep, 2,589,088 B, 3264x2448 4:2:0, cut inside its scan: no EOI)")
            fig.pop("_cs"); fig.pop("_ds")
            extras["reference_benchmark_file"] = fig
        except Exception as e:
            extras["reference_benchmark_file"] = {"error": repr(e)[:300]}
        out["extra"] = extras
    if world == 1 and not args.no_cpu_baseline and not stub:
        cb = cpu_baseline(uniq)
        if cb:
            try:   # the same file through the reference binary, for the `lepton -benchmark` row
                rbc = cpu_baseline([reference_benchmark_jpeg()], budget_s=4.0, what="x the file `lepton -benchmark` codes (3264x2448 4:2:0, no EOI)")
                if rbc:
                    cb["reference_benchmark_file"] = {k: rbc[k] for k in ("value", "encode_MBps", "decode_MBps", "sample") if k in rbc}
            except Exception:
                pass
            out["cpu_baseline"] = cb
    print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
