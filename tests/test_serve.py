"""The serving surface: wire behaviour of `lepton -socket` (src/lepton/socket_serve.cc, the reference's
test_suite/sockettester.py) in front of the GPU batch pipeline.  CPU tests drive the server's IO / batching / time-bound
logic with a table-lookup processor (golden pairs written by the reference) and compare its answers with the real reference
server where that binary exists; the GPU tests run the daemon itself."""
import os
import subprocess
import threading
import time
import uuid
import zlib

import pytest

from conftest import ROOT, golden, golden_cases
from lepton_amd.codec import LeptonError
from lepton_amd.serve import Server, request, zlib0_wrap

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "lepton")
SERVED = os.path.join(ROOT, "lepton_amd", "lepton_served")


def _name():
    return "/tmp/lep-%s" % uuid.uuid4().hex[:12]


class Table:
    """stands in for the GPU on CPU-only runs: answers from the golden pairs the reference wrote"""

    def __init__(self, delay=0.0):
        self.fwd, self.back = {}, {}
        for c in golden_cases():
            j, l = golden(c)
            self.fwd[j] = l
            self.back[l] = j
        self.calls = []
        self.delay = delay

    def __call__(self, kind, files):
        self.calls.append((kind, len(files)))
        if self.delay:
            time.sleep(self.delay)
        table = self.back if kind else self.fwd
        return [(0, table[f]) if f in table else (42 if kind == 0 else 7, None) for f in files]


def _fan_out(address, payloads, threads=16):
    out = [None] * len(payloads)

    def work(k):
        for i in range(k, len(payloads), threads):
            out[i] = request(address, payloads[i])

    ts = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    return out


def test_zlib0_framing():
    # 78 01, stored blocks of 65535, last one final, Adler-32 (src/io/Zlib0.cc:36-120)
    for n in (0, 1, 5, 65534, 65535, 65536, 131070, 131071, 300000):
        d = (bytes(range(256)) * (n // 256 + 1))[:n]
        w = zlib0_wrap(d)
        assert w[:2] == b"\x78\x01" and zlib.decompress(w) == d
        nblocks = max(1, -(-n // 65535))
        assert len(w) == 2 + 5 * nblocks + n + 4
        assert w[2] == (1 if nblocks == 1 else 0)
        last = 2 + (nblocks - 1) * (5 + 65535)
        assert w[last] == 1 and int.from_bytes(w[last + 1:last + 3], "little") == n - (nblocks - 1) * 65535


def test_serve_round_trips_and_batches():
    name = _name()
    tab = Table()
    cases = golden_cases()
    with Server(name, process=tab, batch_window_us=300000, max_batch=64) as srv:
        assert os.path.exists(name) and os.path.exists(name + ".z0") and os.path.exists(name + ".lock")
        jpgs = [golden(c)[0] for c in cases]
        leps = [golden(c)[1] for c in cases]
        got = _fan_out(name, jpgs, threads=len(jpgs))
        assert got == leps
        back = _fan_out(name, leps, threads=len(leps))
        assert back == jpgs
        zs = _fan_out(name + ".z0", leps[:6], threads=6)
        assert [zlib.decompress(z) for z in zs] == jpgs[:6] and zs[0] == zlib0_wrap(jpgs[0])
        # the zlib socket only wraps decoded JPEGs; a compression through it is answered plainly (jpgcoder.cc:2204-2221)
        assert request(name + ".z0", jpgs[0]) == leps[0]
        # a .lep under the zeta magic asks for the zlib answer on the plain socket (jpgcoder.cc:552, 2204)
        assert request(name, b"\xce\xb6" + leps[1][2:]) == zlib0_wrap(jpgs[1])
        st = srv.stats()
        n = 2 * len(cases) + 6 + 2
        assert st["accepted"] == n and st["answered"] == n and st["failed"] == 0 and st["timed_out"] == 0
        # concurrency became batches: far fewer processor calls than requests
        assert st["largest_batch"] >= 4 and st["batches"] < n // 2
        assert st["bytes_in"] >= sum(map(len, jpgs)) + sum(map(len, leps))
    assert not os.path.exists(name) and not os.path.exists(name + ".z0") and not os.path.exists(name + ".lock")


def test_serve_failures_close_without_bytes():
    name = _name()
    with Server(name, process=Table(), batch_window_us=1000, max_file_bytes=100000) as srv:
        assert request(name, b"hello, not an image") == b""          # unknown file type
        assert request(name, b"") == b"" and request(name, b"\xff") == b""
        assert request(name, b"\xff\xd8 a jpeg the coder refuses") == b""
        assert srv.stats()["last_failure_code"] == 42
        assert request(name, b"\xcf\x84 a lepton file the coder refuses") == b""
        assert srv.stats()["last_failure_code"] == 7
        assert request(name, b"\xff\xd8" + bytes(200000)) == b""       # over max_file_bytes: dropped while uploading
        st = srv.stats()
        assert st["failed"] == 5 and st["rejected"] == 1 and st["answered"] == 0
        # the server is still healthy
        j, l = golden("c420_160x120")
        assert request(name, j) == l


def test_serve_name_is_exclusive_and_tcp():
    name = _name()
    with Server(name, process=Table()):
        with pytest.raises(LeptonError) as e:   # a second server on the same name must not start (sockettester.py:52-59)
            Server(name, process=Table())
        assert e.value.code == 33
        assert os.path.exists(name)
    port = 20000 + os.getpid() % 20000
    j, l = golden("c444_96x80")
    with Server(None, tcp_port=port, zlib_tcp_port=port + 1, process=Table()):
        assert request(("127.0.0.1", port), j) == l
        assert zlib.decompress(request(("127.0.0.1", port + 1), l)) == j


def test_serve_time_bound():
    # -timebound counts from the first byte received and closes the connection whatever state the request is in
    name = _name()
    j, l = golden("c420_160x120")
    with Server(name, process=Table(delay=2.0), time_bound_ms=150, batch_window_us=1000) as srv:
        t0 = time.time()
        assert request(name, j) == b""
        assert time.time() - t0 < 1.5           # closed at the bound, not when the batch came back (generous: loaded CI hosts)
        time.sleep(2.2)
        assert srv.stats()["timed_out"] == 1 and srv.stats()["answered"] == 0
    with Server(name, process=Table(), time_bound_ms=5000) as srv:
        assert request(name, j) == l
        # a client that never finishes its upload is cut off too
    with Server(name, process=Table(), time_bound_ms=100) as srv:
        import socket as so

        s = so.socket(so.AF_UNIX, so.SOCK_STREAM)
        s.connect(name)
        s.sendall(j[:100])
        s.settimeout(2.0)
        assert s.recv(10) == b""
        s.close()
        assert srv.stats()["timed_out"] == 1


def test_serve_max_connections_queues_in_backlog():
    name = _name()
    tab = Table(delay=0.05)
    j, l = golden("c422_128x72")
    with Server(name, process=tab, max_connections=2, batch_window_us=1000) as srv:
        got = _fan_out(name, [j] * 10, threads=10)
        assert got == [l] * 10
        assert srv.stats()["largest_batch"] <= 2 and srv.stats()["answered"] == 10


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="the reference binary is only built where /root/reference exists")
def test_same_answers_as_the_reference_server():
    ref_name, name = _name(), _name()
    proc = subprocess.Popen([REF_BIN, "-socket=" + ref_name, "-timebound=50000ms", "-preload"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    try:
        assert proc.stdout.readline().strip().decode() == ref_name
        with Server(name, process=Table()) as _:
            for case in ("c420_160x120", "gray_120x88", "truncated", "prog_trailing_garbage"):
                j, l = golden(case)
                if case.startswith("prog"):
                    continue   # the reference server would need -allowprogressive; covered by the golden .lep itself
                for payload in (j, l, b"\xce\xb6" + l[2:], b"garbage"):
                    for suffix in ("", ".z0"):
                        assert request(name + suffix, payload) == request(ref_name + suffix, payload)
    finally:
        proc.terminate()
        proc.wait()
    assert not os.path.exists(ref_name)


# ---- on the GPU: the daemon itself -----------------------------------------------------------------------------------
def _start_daemon(*extra):
    name = _name()
    proc = subprocess.Popen([SERVED, "-socket=" + name] + list(extra), stdout=subprocess.PIPE)
    line = proc.stdout.readline().strip().decode()
    assert line == name, "lepton_served did not come up"
    return proc, name


@pytest.mark.gpu
def test_daemon_serves_golden_files_bit_exact():
    cases = golden_cases()
    proc, name = _start_daemon("-timebound=120000ms", "-batchwindow=20000")
    try:
        jpgs = [golden(c)[0] for c in cases]
        leps = [golden(c)[1] for c in cases]
        assert _fan_out(name, jpgs * 4, threads=32) == leps * 4
        assert _fan_out(name, leps * 4, threads=32) == jpgs * 4
        assert [zlib.decompress(z) for z in _fan_out(name + ".z0", leps, threads=16)] == jpgs
        assert request(name, b"\xff\xd8 not really") == b""
        # a second daemon on the name prints nothing and exits (sockettester.py:52-59)
        dup = subprocess.Popen([SERVED, "-socket=" + name], stdout=subprocess.PIPE)
        assert dup.stdout.readline() == b"" and dup.wait() != 0
        assert request(name, jpgs[0]) == leps[0]
    finally:
        proc.terminate()
        assert proc.wait() == 0
    assert not os.path.exists(name) and not os.path.exists(name + ".z0")


@pytest.mark.gpu
def test_daemon_batches_a_burst():
    from lepton_amd import corpus

    proc, name = _start_daemon("-batchwindow=50000", "-maxbatch=256")
    try:
        jpgs = [corpus.synth_jpeg(640, 480, 4000 + i) for i in range(48)]
        leps = _fan_out(name, jpgs, threads=48)
        assert all(l[:2] == b"\xcf\x84" for l in leps)
        assert _fan_out(name, leps, threads=48) == jpgs
    finally:
        proc.terminate()
        proc.wait()


def test_daemon_command_line():
    """lepton_served's option handling (no GPU needed: every case ends before or at device creation)"""
    def run(*args):
        r = subprocess.run([SERVED] + list(args), capture_output=True, text=True, timeout=60)
        return r.returncode, r.stdout, r.stderr

    assert os.path.exists(SERVED), "lepton_served is built by __graft_entry__.build()"
    rc, out, err = run()
    assert rc == 1 and "usage" in err and out == ""
    rc, out, err = run("-socket=/tmp/lep-x", "-bogus")
    assert rc == 1 and "unknown option -bogus" in err
    rc, out, err = run("-listen=2402", "-timebound=5s")          # jpgcoder.cc:1209-1212
    assert rc == 1 and "Time bound action only supported with UNIX domain sockets" in err
    rc, out, err = run("-devices=0,1", "-socket")                 # children need a name to derive theirs from
    assert rc == 1 and "-devices needs -socket=<name>" in err
    import torch
    if not torch.cuda.is_available():
        name = _name()
        rc, out, err = run("-socket=" + name, "-timebound=10000ms", "-maxchildren=8", "-listenbacklog=64", "-skipverify", "-preload")
        assert rc == 120 and "no usable gfx950 device" in err and out == ""     # no CPU fallback: nothing is served without a GPU
        assert not os.path.exists(name)


def test_daemon_supervises_one_process_per_device():
    """`lepton_served -devices=0,1`: one serving process per GPU under a supervisor (multi-GPU serving): a child
    that is killed while serving is reported and started again on the same sockets; children that end at once (here: no
    device) are reported and stay down; SIGTERM to the supervisor takes everything down and removes the socket files.
    Runs without a GPU: LEP_SERVED_NO_DEVICE makes the children listen and answer every request with a failure."""
    import signal

    import psutil

    name = _name()
    env = dict(os.environ, LEP_SERVED_NO_DEVICE="1")
    sup = subprocess.Popen([SERVED, "-devices=0,1", "-socket=" + name, "-skipverify"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    try:
        socks = [name + ".0", name + ".1"]

        def up(timeout=20.0):
            t0 = time.time()
            while time.time() - t0 < timeout:
                if all(os.path.exists(s) for s in socks):
                    return True
                time.sleep(0.05)
            return False

        assert up()
        assert request(socks[0], golden("c420_160x120")[0]) == b"" and request(socks[1], b"\xcf\x84junk") == b""   # no device: failures only
        kids = psutil.Process(sup.pid).children()
        assert len(kids) == 2
        victim = kids[1]
        time.sleep(0.3)
        victim.send_signal(signal.SIGKILL)
        t0 = time.time()
        while time.time() - t0 < 20 and len([k for k in psutil.Process(sup.pid).children() if k.pid != victim.pid and k.is_running()]) < 2:
            time.sleep(0.05)
        now = psutil.Process(sup.pid).children()
        assert len(now) == 2 and victim.pid not in [k.pid for k in now], "the killed child was not replaced"
        t0 = time.time()
        while time.time() - t0 < 20:   # the replacement owns the sockets again
            try:
                if request(socks[0], b"\xff\xd8x") == b"" and request(socks[1], b"\xff\xd8x") == b"":
                    break
            except OSError:
                time.sleep(0.1)
        else:
            raise AssertionError("sockets did not come back")
        assert sup.poll() is None
    finally:
        sup.terminate()
        out, err = sup.communicate(timeout=30)
    assert b"restarting" in err and b"signal 9" in err
    assert not any(os.path.exists(s) for s in (name + ".0", name + ".1", name + ".0.z0", name + ".1.z0"))
    # without the hook and without a GPU every child ends at once: reported, not restarted, and the supervisor gives up with their code
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([SERVED, "-devices=0,1", "-socket=" + _name()], capture_output=True, text=True, timeout=60)
        assert r.returncode == 120 and r.stderr.count("ended (exit code 120)") == 2 and "restarting" not in r.stderr
