"""bench.py's N > 1 path without a GPU (`--gpus N` used to run one rank): started the way the driver
starts a 1-GPU run -- `python bench.py --gpus 2`, no torchrun environment -- it must start two ranks itself, shard the mixed
corpus between them by JPEG bytes, all-reduce the counters and print ONE line that says n_gpus 2.  The device layer is
tests/bench_stub.py (LEP_BENCH_DEVICE=stub): process group gloo, nothing is coded."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(tmp_path, extra, env_extra=None):
    log = str(tmp_path / "stub")
    env = dict(os.environ, LEP_BENCH_DEVICE="stub", LEP_BENCH_STUB_LOG=log, PYTHONPATH=os.path.join(ROOT, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--images", "6", "--unique", "2", "--width", "160",
           "--height", "120", "--mixed-images", "12", "--mixed-shapes", "96x64,320x240", "--e2e-images", "4", "--no-extras", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout            # exactly one JSON line, from rank 0
    notes = []
    for rank in range(8):
        p = "%s.%d" % (log, rank)
        if os.path.exists(p):
            notes += [json.loads(l) for l in open(p)]
    return json.loads(lines[0]), notes


def test_gpus_2_starts_two_ranks_and_shards_the_mixed_corpus(tmp_path):
    out, notes = run_bench(tmp_path, ["--gpus", "2"])
    assert out["n_gpus"] == 2 and out["config"]["device_layer"].startswith("stub")
    assert out["scaling"] == "weak" and out["config"]["images_per_gpu"] == 6
    res = [n for n in notes if n["what"] == "resident"]
    assert sorted(n["rank"] for n in res) == [0, 1]                       # both ranks ran the weak-scaling step
    assert abs(out["config"]["jpeg_MB_per_step"] * 1e6 - sum(n["bytes"] for n in res)) < 1000   # whole-job aggregate, not one rank's
    mixed = [n for n in notes if n["what"] == "pipeline" and n["label"].startswith("mixed")]
    assert sorted(n["rank"] for n in mixed) == [0, 1]
    assert sum(n["files"] for n in mixed) == 12 == out["mixed"]["files"]   # ONE corpus, every file on exactly one rank
    assert mixed[0]["digest"] != mixed[1]["digest"]
    b = sorted(n["bytes"] for n in mixed)
    assert b[1] / (sum(b) / 2) < 1.2                                        # dealt by bytes, not by count
    assert out["mixed"]["scaling"] == "strong" and out["mixed"]["n_gpus"] == 2
    assert abs(out["mixed"]["jpeg_MB"] * 1e6 - sum(b)) < 1e5
    assert out["end_to_end"]["n_gpus"] == 2
    # every rank's host side is in the line: threads, CPUs, pinned staging, its share of the mixed corpus and of the wall clock
    pr = out["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and all({"host_threads", "pinned_MB", "cpus_allowed", "numa_node", "resident_s", "mixed_compress_s"} <= set(r) for r in pr)
    assert sum(r["mixed_files"] for r in pr) == 12 and max(r["mixed_compress_s"] for r in pr) > 0
    # one definition of MB/s in the whole line
    for fig in (out["mixed"], out["end_to_end"]):
        assert abs(1 / fig["value"] - (1 / fig["compress_MBps"] + 1 / fig["decompress_MBps"])) / (1 / fig["value"]) < 0.08   # (figures are rounded to 0.1 MB/s)


def test_one_rank_runs_the_same_corpus_alone(tmp_path):
    out, notes = run_bench(tmp_path, [])
    assert out["n_gpus"] == 1 and out["mixed"]["files"] == 12 and out["mixed"]["n_gpus"] == 1
    assert len(out["per_rank"]) == 1 and out["per_rank"][0]["mixed_files"] == 12
    assert [n["rank"] for n in notes if n["what"] == "resident"] == [0]


def test_under_torchrun_the_ranks_are_the_launcher_s(tmp_path):
    """the driver's N > 1 command line: python -m torch.distributed.run ... bench.py --gpus 2"""
    import socket

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    log = str(tmp_path / "stub")
    env = dict(os.environ, LEP_BENCH_DEVICE="stub", LEP_BENCH_STUB_LOG=log, PYTHONPATH=os.path.join(ROOT, "tests"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--images", "4", "--unique", "2", "--width", "96", "--height", "64",
           "--mixed-images", "0", "--no-end-to-end", "--no-extras", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_gpus_8_codes_the_ten_thousand_file_corpus_once(tmp_path):
    """BASELINE.json configs[3] at its stated size before an 8-GPU node exists: `bench.py --gpus 8` starts
    eight ranks, every rank generates ONLY its own share of the 10,000 distinct files (seeds 20000..29999, dealt by size class -- no
    rank needs the others' bytes to know its share), the line says n_gpus 8, mixed.files 10000, and that one point is not a curve."""
    out, notes = run_bench(tmp_path, ["--gpus", "8", "--mixed-images", "10000", "--mixed-distinct", "10000", "--mixed-shapes", "32x24,64x48", "--no-end-to-end"])
    assert out["n_gpus"] == 8 and [r["rank"] for r in out["per_rank"]] == list(range(8))
    # the first SCALE run must be able to show N ranks on N devices: every rank's PCI address and xGMI hive, the world size each rank
    # saw and the backend are in the line
    assert out["distributed"]["world_size"] == 8 and out["distributed"]["distinct_devices"] == 8 and out["distributed"]["backend"] == "gloo"
    assert all(r["world_size_seen"] == 8 and r["backend_is_rccl"] == 0 and r["pci_bus"] == 0x10 + r["local_rank"] and r["xgmi_hive_lo"] == 0x1234 for r in out["per_rank"])
    mixed = [n for n in notes if n["what"] == "pipeline" and n["label"].startswith("mixed")]
    assert sorted(n["rank"] for n in mixed) == list(range(8))
    assert sum(n["files"] for n in mixed) == 10000 == out["mixed"]["files"] and out["mixed"]["distinct"] == 10000
    assert out["mixed"]["generated_by_the_ranks"] == 10000               # every file made exactly once, on the rank that codes it
    assert len({n["digest"] for n in mixed}) == 8
    b = [n["bytes"] for n in mixed]
    assert max(b) / (sum(b) / 8) < 1.1
    assert "no 1 -> 8 curve" in out["mixed"]["scaling_curve"] and out["mixed"]["n_gpus"] == 8


def test_the_corpus_cache_is_filled_once(tmp_path):
    """LEP_CORPUS_CACHE: a second run on the same box reads the mixed corpus instead of generating it"""
    cache = str(tmp_path / "corpus")
    out1, _ = run_bench(tmp_path, ["--gpus", "2"], {"LEP_CORPUS_CACHE": cache})
    n = len(os.listdir(cache))
    assert out1["mixed"]["generated_by_the_ranks"] == n == out1["mixed"]["distinct"]
    out2, _ = run_bench(tmp_path, ["--gpus", "2"], {"LEP_CORPUS_CACHE": cache})
    assert out2["mixed"]["generated_by_the_ranks"] == 0 and out2["mixed"]["jpeg_MB"] == out1["mixed"]["jpeg_MB"]
