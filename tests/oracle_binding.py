"""ctypes binding of oracle/liblepton_oracle.so -- TEST INFRASTRUCTURE (the CPU restatement)."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liblepton_oracle.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "lepton")


class LorImage(C.Structure):
    _fields_ = [
        ("ncomp", C.c_int),
        ("blocks", C.c_void_p * 4),
        ("width_blocks", C.c_int * 4),
        ("height_blocks", C.c_int * 4),
        ("coded_blocks", C.c_int * 4),
        ("coded_height", C.c_int * 4),
        ("mcu_rows", C.c_int),
        ("qtable_zigzag", (C.c_uint16 * 64) * 4),
    ]


_lib = None


def oracle():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(
            os.path.join(ORACLE_DIR, "lepton_oracle.c")
        ):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])
        L = C.CDLL(ORACLE_SO)
        L.lor_encode_segment.argtypes = [C.POINTER(LorImage), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_size_t), C.POINTER(C.c_uint64)]
        L.lor_decode_segment.argtypes = [C.POINTER(LorImage), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_uint64)]
        L.lor_model_bytes.restype = C.c_size_t
        _lib = L
    return _lib


def to_lor(desc):
    """lepton_amd.abi.ImageDesc -> LorImage (same frame, oracle's struct)."""
    im = LorImage()
    im.ncomp = desc.ncomp
    im.mcu_rows = desc.mcu_rows
    for c in range(desc.ncomp):
        im.blocks[c] = desc.blocks[c]
        im.width_blocks[c] = desc.width_blocks[c]
        im.height_blocks[c] = desc.height_blocks[c]
        im.coded_blocks[c] = desc.coded_blocks[c]
        im.coded_height[c] = desc.coded_height[c]
        for i in range(64):
            im.qtable_zigzag[c][i] = desc.qtable_zigzag[c][i]
    return im


def oracle_encode(desc, segs):
    """returns (list of per-segment streams, total bins)"""
    L = oracle()
    im = to_lor(desc)
    cap = max(1 << 20, desc.total_blocks() * 160 + (1 << 16))
    buf = C.create_string_buffer(cap)
    out, bins_total = [], 0
    for s in segs:
        n = C.c_size_t(0)
        bins = C.c_uint64(0)
        rc = L.lor_encode_segment(C.byref(im), s.luma_y_start, s.luma_y_end, s.is_last, buf, cap, C.byref(n), C.byref(bins))
        if rc:
            raise RuntimeError("oracle encode exit code %d" % rc)
        out.append(buf.raw[: n.value])
        bins_total += bins.value
    return out, bins_total


def oracle_decode(desc, segs, streams):
    L = oracle()
    im = to_lor(desc)
    for s, st in zip(segs, streams):
        b = C.create_string_buffer(bytes(st), len(st)) if len(st) else C.create_string_buffer(1)
        rc = L.lor_decode_segment(C.byref(im), s.luma_y_start, s.luma_y_end, s.is_last, b, len(st), None)
        if rc:
            raise RuntimeError("oracle decode exit code %d" % rc)
