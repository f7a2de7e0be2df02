#!/usr/bin/env python3
"""Serving surface under load: lepton_served on a Unix socket, many concurrent clients speaking the
`lepton -socket` protocol (send file, half-close, read answer), every answer checked.  Clients are forked processes x threads
so the GIL is not the limit.   usage: python scripts/bench_serve.py [--requests 2048] [--clients 1024] [--width 3840 --height 2160]"""
import argparse
import json
import multiprocessing as mp
import os
import subprocess
import sys
import threading
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def client_proc(name, payloads, idxs, threads, q):
    from lepton_amd.serve import request
    out = {}

    def work(k):
        for i in idxs[k::threads]:
            out[i] = request(name, payloads[i % len(payloads)], timeout=600)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    # answers are large: return one per distinct payload plus a digest of the rest
    import hashlib
    q.put({i: hashlib.md5(v).hexdigest() for i, v in out.items()})


def fan(name, payloads, nreq, clients, procs):
    q = mp.Queue()
    per = max(1, clients // procs)
    ps = [mp.Process(target=client_proc, args=(name, payloads, list(range(p, nreq, procs)), per, q)) for p in range(procs)]
    t0 = time.perf_counter()
    [p.start() for p in ps]
    res = {}
    for _ in ps:
        res.update(q.get())
    [p.join() for p in ps]
    return time.perf_counter() - t0, res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--requests", type=int, default=2048)
    ap.add_argument("--clients", type=int, default=1024)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--unique", type=int, default=16)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--window-us", type=int, default=20000)
    ap.add_argument("--skipverify", action="store_true")
    args = ap.parse_args()
    import hashlib
    import __graft_entry__ as ge
    ge.build()
    from lepton_amd import corpus
    from lepton_amd.serve import request

    uniq = corpus.make_corpus(args.unique, args.width, args.height, 10000)
    name = "/tmp/lep-bench-%s" % uuid.uuid4().hex[:10]
    cmd = [os.path.join(ROOT, "lepton_amd", "lepton_served"), "-socket=" + name, "-batchwindow=%d" % args.window_us, "-listenbacklog=4096"]
    if args.skipverify:
        cmd.append("-skipverify")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        assert proc.stdout.readline().strip().decode() == name
        leps = [request(name, j, timeout=600) for j in uniq]           # also warms kernels
        assert all(l[:2] == b"\xcf\x84" for l in leps)
        assert [request(name, l, timeout=600) for l in leps[:2]] == uniq[:2]
        # one client, one request at a time: what a lone caller waits (the reference's forked child answers a 4K file in ~280 / ~60 ms)
        lat = {}
        for label, payloads in (("compress", uniq), ("decompress", leps)):
            ts = []
            for i in range(12):
                t0 = time.perf_counter(); request(name, payloads[i % len(payloads)], timeout=600); ts.append((time.perf_counter() - t0) * 1e3)
            ts.sort()
            lat[label] = {"p50_ms": round(ts[len(ts) // 2], 1), "min_ms": round(ts[0], 1), "max_ms": round(ts[-1], 1)}
        fan(name, uniq, min(args.requests, 1024), args.clients, args.procs)   # staging buffers of the batch pipeline
        mb = sum(len(uniq[i % len(uniq)]) for i in range(args.requests)) / 1e6
        tc, rc = fan(name, uniq, args.requests, args.clients, args.procs)
        want = [hashlib.md5(l).hexdigest() for l in leps]
        assert all(rc[i] == want[i % len(uniq)] for i in range(args.requests)), "a compressed answer differs"
        td, rd = fan(name, leps, args.requests, args.clients, args.procs)
        wantj = [hashlib.md5(j).hexdigest() for j in uniq]
        assert all(rd[i] == wantj[i % len(uniq)] for i in range(args.requests)), "a decompressed answer differs"
    finally:
        proc.terminate()
        err = proc.stderr.read().decode()[-400:]
        proc.wait()
    print(json.dumps({
        "workload": "%d requests of %dx%d JPEGs (%d distinct) over a Unix socket, %d concurrent clients, lepton_served %s" % (
            args.requests, args.width, args.height, len(uniq), args.clients, "without verification" if args.skipverify else "with on-GPU round-trip verification"),
        "jpeg_MB": round(mb, 1), "compress_MBps": round(mb / tc, 1), "compress_requests_per_s": round(args.requests / tc, 1),
        "decompress_MBps": round(mb / td, 1), "decompress_requests_per_s": round(args.requests / td, 1),
        "one_client_latency": lat,
        "answers": "every answer checked against the first (md5)", "server_log": err.strip().splitlines()[-1] if err.strip() else ""}))


if __name__ == "__main__":
    main()
