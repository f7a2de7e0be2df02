"""ctypes binding of the serving surface (include/lepton_mi355x.h, lep_serve_*) plus the client side of the wire protocol
of `lepton -socket` (test_suite/sockettester.py of the reference: send the file, half-close, read until EOF).
No compute here: the server hands batches to the GPU pipeline inside the library (or, in CPU tests, to a callback)."""
import ctypes as C
import socket

from . import abi
from .codec import LeptonError

_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]


def request(address, data, timeout=60.0):
    """One request: address is a UDS path (str / bytes) or a (host, port) pair.  Returns the answer (b'' = the server
    refused, failed or ran into its time bound: the reference's forked worker just dies and the socket closes)."""
    if isinstance(address, tuple):
        s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    else:
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    try:
        s.connect(address)     # blocking: with the listen backlog full a Unix-socket connect waits (a timed one fails with EAGAIN)
        s.settimeout(timeout)
        try:
            s.sendall(data)
            s.shutdown(socket.SHUT_WR)
        except OSError:
            pass   # the server may have dropped us already (oversized upload, time bound)
        parts = []
        while True:
            try:
                b = s.recv(1 << 20)
            except ConnectionResetError:
                break
            if not b:
                break
            parts.append(b)
        return b"".join(parts)
    finally:
        s.close()


def zlib0_wrap(data):
    out = abi.Bytes()
    rc = abi.lib().lep_zlib0_wrap(data, len(data), C.byref(out))
    if rc:
        raise LeptonError(rc, "lep_zlib0_wrap")
    try:
        return out.tobytes()
    finally:
        abi.lib().lep_free(out.data)


class Server:
    """lep_serve_start .. lep_serve_stop.  process (tests only): callable(kind, [bytes]) -> [(exit_code, bytes or None)]."""

    def __init__(self, uds_path=None, gpu=None, tcp_port=0, zlib_tcp_port=0, time_bound_ms=0, max_batch=0, batch_window_us=0,
                 max_connections=0, max_file_bytes=0, verify=True, host_huffman=False, process=None, zlib_uds_path=None):
        self._L = abi.lib()
        o = abi.ServeOptions()
        self._keep = [uds_path.encode() if isinstance(uds_path, str) else uds_path,
                      zlib_uds_path.encode() if isinstance(zlib_uds_path, str) else zlib_uds_path]
        o.uds_path, o.zlib_uds_path = self._keep
        o.tcp_port, o.zlib_tcp_port = tcp_port, zlib_tcp_port
        o.time_bound_ms, o.max_batch, o.batch_window_us = time_bound_ms, max_batch, batch_window_us
        o.max_connections, o.max_file_bytes = max_connections, max_file_bytes
        o.batch.verify, o.batch.host_huffman = int(verify), int(host_huffman)
        if gpu is not None:
            o.gpu = gpu.handle
        if process is not None:
            def thunk(_user, kind, ins, n, outs, status):
                try:
                    res = process(kind, [C.string_at(ins[i].data, ins[i].len) for i in range(n)])
                    for i, (code, data) in enumerate(res):
                        status[i] = code
                        if code == 0 and data:
                            p = _libc.malloc(len(data))
                            C.memmove(p, data, len(data))
                            outs[i].data, outs[i].len, outs[i].cap = p, len(data), len(data)
                    return 0
                except Exception:   # a broken callback must not take the server thread down
                    return 1
            self._cb = abi.SERVE_PROCESS_FN(thunk)
            o.process = self._cb
        self.handle = C.c_void_p()
        rc = self._L.lep_serve_start(C.byref(o), C.byref(self.handle))
        if rc:
            self.handle = C.c_void_p()
            raise LeptonError(rc, "lep_serve_start")

    def stats(self):
        st = abi.ServeStats()
        self._L.lep_serve_get_stats(self.handle, C.byref(st))
        return {k: getattr(st, k) for k, _ in abi.ServeStats._fields_}

    def stop(self):
        if self.handle:
            self._L.lep_serve_stop(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.stop()
