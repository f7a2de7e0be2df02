"""TEST INFRASTRUCTURE: a stand-in for bench.py's device layer (bench.HipDevice) so that the multi-rank plumbing of the
benchmark -- `--gpus N` starting its own ranks, the gloo / RCCL process group, image sharding, the counter all-reduce and the
one JSON line -- runs in the CPU suite (tests/test_bench_ranks.py, LEP_BENCH_DEVICE=stub).  It codes nothing: every "kernel"
is a sleep proportional to the bytes it was handed, and each call is logged so the test can see which rank got which files."""
import json
import os
import time


class StubDevice:
    name = "stub (tests/bench_stub.py: no coding, sleeps)"

    def __init__(self, local_rank):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = local_rank
        self.log = os.environ.get("LEP_BENCH_STUB_LOG")

    def _note(self, what, **kw):
        if self.log:
            with open("%s.%d" % (self.log, self.rank), "a") as f:
                f.write(json.dumps(dict(kw, what=what, rank=self.rank)) + "\n")

    def sync(self):
        pass

    def footprint(self):
        return {"pinned_MB": 0.0, "staging_device_MB": 0.0, "host_threads": 1}

    def identity(self):
        # (a made-up PCI address per rank: the plumbing that carries it is what the CPU suite exercises)
        return {"pci_domain": 0, "pci_bus": 0x10 + self.local_rank, "pci_device": 0, "pci_function": 0, "xgmi_hive_hi": 0, "xgmi_hive_lo": 0x1234}

    def resident(self, uniq, images, steps, warmup, barrier, check_parity=True, with_latency=False):
        nb = sum(len(uniq[i % len(uniq)]) for i in range(images))
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            time.sleep(nb / 50e6)
        barrier()
        self._note("resident", images=images, bytes=nb)
        return {"jpeg_bytes": nb, "images": images, "segments": images, "blocks": nb // 16, "stream_bytes": nb * 3 // 4,
                "elapsed": time.perf_counter() - t0, "enc_ms": steps * nb / 50e6 * 400, "dec_ms": steps * nb / 50e6 * 600,
                "names": {"encode": "stub_encode", "decode": "stub_decode"}, "parity": "stub: nothing was coded",
                "bins_per_image": None, "latency": None}

    def pipeline(self, jpgs, label, verify=False):
        nb = sum(map(len, jpgs))
        t = nb / 100e6 + 0.01
        time.sleep(2 * t)
        import hashlib

        self._note("pipeline", label=label[:20], files=len(jpgs), bytes=nb, digest=hashlib.md5(b"".join(hashlib.md5(j).digest() for j in jpgs)).hexdigest())
        st = {"wall_s": t, "h2d_bytes": nb, "d2h_bytes": nb, "parse_s": 0.0, "write_s": 0.0}
        return {"workload": label, "jpeg_MB": round(nb / 1e6, 3), "lep_MB": round(nb * 0.78 / 1e6, 3), "files": len(jpgs),
                "compress_MBps": round(nb / 1e6 / t, 1), "decompress_MBps": round(nb / 1e6 / t, 1), "value": round(nb / 1e6 / (2 * t), 1),
                "files_per_s": round(len(jpgs) / (2 * t), 1), "parity": "stub", "_cs": dict(st), "_ds": dict(st)}
