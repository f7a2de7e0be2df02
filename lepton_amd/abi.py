"""ctypes binding of include/lepton_mi355x.h (the C ABI of liblepton_mi355x.so).

This is the reference-side binding a maintainer would write for a Python caller; the C++ adapter that
plugs the same ABI behind BaseEncoder/BaseDecoder (src/lepton/base_coders.hh:26-65) is shown in
INTEGRATION.md.  Nothing here computes: every function forwards to the shared library, and loading
fails loudly if the library has not been built (``python -c 'import __graft_entry__ as g; g.build()'``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LEP_LIB_PATH") or os.path.join(_HERE, "liblepton_mi355x.so")   # LEP_LIB_PATH: experiment builds of the same library

MAX_COMPONENTS = 3
MAX_SEGMENTS = 16


class ImageDesc(C.Structure):
    _fields_ = [
        ("ncomp", C.c_int32),
        ("mcu_rows", C.c_int32),
        ("width_blocks", C.c_int32 * MAX_COMPONENTS),
        ("height_blocks", C.c_int32 * MAX_COMPONENTS),
        ("coded_blocks", C.c_int32 * MAX_COMPONENTS),
        ("coded_height", C.c_int32 * MAX_COMPONENTS),
        ("qtable_zigzag", (C.c_uint16 * 64) * MAX_COMPONENTS),
        ("blocks", C.c_void_p * MAX_COMPONENTS),
    ]

    def nblocks(self, c):
        return self.width_blocks[c] * self.height_blocks[c]

    def total_blocks(self):
        return sum(self.nblocks(c) for c in range(self.ncomp))


class Segment(C.Structure):
    _fields_ = [("image", C.c_int32), ("luma_y_start", C.c_int32), ("luma_y_end", C.c_int32), ("is_last", C.c_int32)]


class Handoff(C.Structure):
    _fields_ = [("luma_y_start", C.c_uint16), ("luma_y_end", C.c_uint16), ("segment_size", C.c_uint32),
                ("overhang_byte", C.c_uint8), ("num_overhang_bits", C.c_uint8), ("last_dc", C.c_int16 * 4)]


class Bytes(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t), ("cap", C.c_size_t)]

    def tobytes(self):
        return C.string_at(self.data, self.len) if self.len else b""


class HuffImage(C.Structure):
    _fields_ = [("ncomp", C.c_int32), ("mcuh", C.c_int32), ("mcuv", C.c_int32), ("mcuc", C.c_int32), ("rsti", C.c_int32), ("padbit", C.c_int32),
                ("rst_limit", C.c_uint32), ("interleaved", C.c_int32), ("hs", C.c_int32 * 4), ("vs", C.c_int32 * 4), ("bch", C.c_int32 * 4),
                ("dc_tbl", C.c_int32 * 4), ("ac_tbl", C.c_int32 * 4), ("scan_cmp", C.c_int32 * 4), ("trunc_bc", C.c_int32 * 4), ("blocks", C.c_void_p * 4),
                ("code", (C.c_uint32 * 256) * 4)]


class HuffSegment(C.Structure):
    _fields_ = [("image", C.c_int32), ("mcu_row0", C.c_int32), ("mcu_row1", C.c_int32), ("overhang", C.c_uint32), ("last_dc", C.c_int16 * 4),
                ("out_off", C.c_uint64), ("out_cap", C.c_uint32), ("pad", C.c_uint32)]


class HuffEnd(C.Structure):
    _fields_ = [("attempted", C.c_uint32), ("overhang_byte", C.c_uint8), ("num_overhang_bits", C.c_uint8), ("last_dc", C.c_int16 * 4), ("pad", C.c_uint16)]


class HuffProgImage(C.Structure):
    _fields_ = [("ncomp", C.c_int32), ("mcuh", C.c_int32), ("mcuv", C.c_int32), ("mcuc", C.c_int32), ("rsti", C.c_int32), ("padbit", C.c_int32),
                ("hs", C.c_int32 * 4), ("vs", C.c_int32 * 4), ("bch", C.c_int32 * 4), ("bcv", C.c_int32 * 4), ("nch", C.c_int32 * 4),
                ("ncv", C.c_int32 * 4), ("mbs", C.c_int32 * 4), ("blocks", C.c_void_p * 4)]


class HuffProgScan(C.Structure):
    _fields_ = [("image", C.c_int32), ("cmpc", C.c_int32), ("cmp", C.c_int32 * 4), ("from_", C.c_int32), ("to", C.c_int32), ("sah", C.c_int32),
                ("sal", C.c_int32), ("max_eobrun", C.c_int32), ("tbl", C.c_int32 * 4), ("rsti", C.c_int32), ("out_off", C.c_uint64), ("out_cap", C.c_uint32),
                ("corr_off", C.c_uint32), ("corr_cap", C.c_uint32), ("file_bound", C.c_uint32), ("code", (C.c_uint32 * 256) * 4)]


class HuffDecImage(C.Structure):
    _fields_ = [("scan", C.c_void_p), ("scan_len", C.c_uint32), ("ncomp", C.c_int32), ("mcuh", C.c_int32), ("mcuv", C.c_int32), ("mcuc", C.c_int32),
                ("rsti", C.c_int32), ("flags", C.c_int32), ("reserved0", C.c_int32), ("hs", C.c_int32 * 4), ("vs", C.c_int32 * 4), ("bch", C.c_int32 * 4), ("dc_tbl", C.c_int32 * 4),
                ("ac_tbl", C.c_int32 * 4), ("scan_cmp", C.c_int32 * 4), ("blocks", C.c_void_p * 4), ("rows_off", C.c_uint64),
                ("lut", (C.c_uint16 * 512) * 4), ("maxcode", (C.c_int32 * 8) * 4), ("valoff", (C.c_int32 * 8) * 4), ("longsym", (C.c_uint8 * 256) * 4)]


class HuffProgDecScan(C.Structure):
    _fields_ = [("t", HuffDecImage), ("cmpc", C.c_int32), ("cmp", C.c_int32 * 4), ("from_", C.c_int32), ("to", C.c_int32), ("sah", C.c_int32),
                ("sal", C.c_int32), ("bcv", C.c_int32 * 4), ("nch", C.c_int32 * 4), ("ncv", C.c_int32 * 4), ("mbs", C.c_int32 * 4),
                ("tbl", C.c_int32 * 4), ("max_eobrun", C.c_int32), ("want_rows", C.c_int32), ("level", C.c_int32), ("pad", C.c_int32),
                ("result_off", C.c_uint64)]


class HuffDecRow(C.Structure):
    _fields_ = [("bitpos", C.c_uint32), ("last_dc", C.c_int16 * 4), ("aux", C.c_int32)]


class BatchOptions(C.Structure):
    _fields_ = [("host_threads", C.c_int32), ("verify", C.c_int32), ("chunk_frame_bytes", C.c_size_t), ("host_huffman", C.c_int32), ("chunk_images", C.c_int32), ("overlap_launches", C.c_int32)]


class BatchStats(C.Structure):
    _fields_ = [("wall_s", C.c_double), ("pipeline_s", C.c_double), ("parse_s", C.c_double), ("stage_s", C.c_double),
                ("write_s", C.c_double), ("h2d_bytes", C.c_double), ("d2h_bytes", C.c_double), ("alloc_s", C.c_double), ("redone_files", C.c_double),
                ("gpu_huffman_files", C.c_double), ("gpu_verified_scans", C.c_double)]


SERVE_PROCESS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(Bytes), C.c_int, C.POINTER(Bytes), C.POINTER(C.c_int32))


class ServeOptions(C.Structure):
    _fields_ = [("uds_path", C.c_char_p), ("zlib_uds_path", C.c_char_p), ("tcp_port", C.c_int32), ("zlib_tcp_port", C.c_int32),
                ("listen_backlog", C.c_int32), ("max_connections", C.c_int32), ("max_file_bytes", C.c_uint32),
                ("time_bound_ms", C.c_uint32), ("max_batch", C.c_int32), ("batch_window_us", C.c_int32), ("gpu", C.c_void_p),
                ("batch", BatchOptions), ("process", SERVE_PROCESS_FN), ("process_user", C.c_void_p)]


class ServeStats(C.Structure):
    _fields_ = [("accepted", C.c_uint64), ("answered", C.c_uint64), ("failed", C.c_uint64), ("timed_out", C.c_uint64),
                ("rejected", C.c_uint64), ("batches", C.c_uint64), ("largest_batch", C.c_uint64), ("bytes_in", C.c_uint64),
                ("bytes_out", C.c_uint64), ("last_failure_code", C.c_int32)]


_lib = None


def lib():
    """Load liblepton_mi355x.so (built in-tree by __graft_entry__.build()); no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "liblepton_mi355x.so is not built (%s). Run `python -c \"import __graft_entry__ as g; g.build()\"`."
                % LIB_PATH
            )
        L = C.CDLL(LIB_PATH)
        P = C.POINTER
        vp = C.c_void_p
        L.lep_version.restype = C.c_char_p
        L.lep_serve_start.argtypes = [P(ServeOptions), P(vp)]
        L.lep_serve_get_stats.argtypes = [vp, P(ServeStats)]
        L.lep_serve_get_stats.restype = None
        L.lep_serve_stop.argtypes = [vp]
        L.lep_serve_stop.restype = None
        L.lep_zlib0_wrap.argtypes = [C.c_char_p, C.c_size_t, P(Bytes)]
        L.lep_jpeg_check_restores.argtypes = [vp, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.lep_jpeg_open_slice.argtypes = [vp, C.c_size_t, C.c_size_t, P(vp)]
        L.lep_jpeg_plan_scan_check.argtypes = [vp, C.c_size_t, vp, vp, P(C.c_uint32), P(C.c_uint32), C.c_int, P(C.c_int), P(C.c_int)]
        L.lep_jpeg_scan_file_range.argtypes = [vp, P(C.c_uint32), P(C.c_uint32)]
        L.lep_jpeg_open_embedded.argtypes = [vp, C.c_size_t, C.c_size_t, P(vp)]
        L.lep_batch_plan.argtypes = [P(C.c_size_t), P(C.c_size_t), C.c_int, P(BatchOptions), P(C.c_int), C.c_int]
        L.lep_jpeg_set_encode_options.argtypes = [vp, C.c_int, C.c_int, C.c_int]
        L.lep_compress_embedded.argtypes = [vp, vp, C.c_size_t, C.c_size_t, P(Bytes)]
        L.lep_compress_slice.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, P(Bytes)]
        L.lep_free.argtypes = [vp]
        L.lep_free.restype = None
        L.lep_jpeg_open.argtypes = [vp, C.c_size_t, C.c_int, P(vp)]
        L.lep_jpeg_close.argtypes = [vp]
        L.lep_jpeg_close.restype = None
        L.lep_jpeg_describe.argtypes = [vp, P(ImageDesc)]
        L.lep_jpeg_plan.argtypes = [vp, C.c_int, P(Segment), C.c_int]
        L.lep_jpeg_write_lep.argtypes = [vp, C.c_int, P(Bytes), C.c_int, P(Bytes)]
        L.lep_file_open.argtypes = [vp, C.c_size_t, P(vp)]
        L.lep_file_close.argtypes = [vp]
        L.lep_file_close.restype = None
        L.lep_file_describe.argtypes = [vp, P(ImageDesc)]
        L.lep_file_segments.argtypes = [vp, P(Segment), P(Bytes), C.c_int]
        L.lep_file_jpeg_size.argtypes = [vp]
        L.lep_file_jpeg_size.restype = C.c_uint32
        L.lep_file_recode.argtypes = [vp, P(Bytes)]
        L.lep_gpu_create.argtypes = [C.c_int, P(vp)]
        L.lep_gpu_destroy.argtypes = [vp]
        L.lep_gpu_destroy.restype = None
        L.lep_gpu_last_error.argtypes = [vp]
        L.lep_gpu_last_error.restype = C.c_char_p
        L.lep_gpu_encode_host.argtypes = [vp, P(ImageDesc), C.c_int, P(Segment), C.c_int, P(Bytes), P(C.c_int32)]
        L.lep_gpu_decode_host.argtypes = [vp, P(ImageDesc), C.c_int, P(Segment), C.c_int, P(Bytes), P(C.c_int32)]
        L.lep_gpu_encode_device.argtypes = [vp, P(ImageDesc), C.c_int, P(Segment), C.c_int, vp, P(C.c_uint64), vp, vp, vp]
        L.lep_gpu_decode_device.argtypes = [vp, P(ImageDesc), C.c_int, P(Segment), C.c_int, vp, P(C.c_uint64), vp, vp, vp]
        L.lep_gpu_sync.argtypes = [vp]
        L.lep_gpu_last_kernel_ms.argtypes = [vp]
        L.lep_gpu_last_kernel_ms.restype = C.c_double
        L.lep_gpu_last_stage_ms.argtypes = [vp, P(C.c_double), C.c_int]
        L.lep_gpu_last_kernel_name.argtypes = [vp]
        L.lep_gpu_last_kernel_name.restype = C.c_char_p
        L.lep_gpu_selftest.argtypes = [vp]
        L.lep_gpu_trim.argtypes = [vp]
        L.lep_gpu_release_memory.argtypes = [vp]
        L.lep_gpu_debug_prof.argtypes = [vp, vp]
        L.lep_gpu_malloc.argtypes = [vp, C.c_size_t, P(vp)]
        L.lep_gpu_free.argtypes = [vp, vp]
        L.lep_gpu_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
        L.lep_gpu_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
        L.lep_gpu_memcpy_d2d.argtypes = [vp, vp, vp, C.c_size_t]
        L.lep_gpu_memset.argtypes = [vp, vp, C.c_int, C.c_size_t]
        L.lep_compress.argtypes = [vp, vp, C.c_size_t, P(Bytes)]
        L.lep_decompress.argtypes = [vp, vp, C.c_size_t, P(Bytes)]
        L.lep_jpeg_is_progressive.argtypes = [vp]
        L.lep_compress_batch.argtypes = [vp, P(Bytes), C.c_int, P(Bytes), P(C.c_int32), P(BatchOptions), P(BatchStats)]
        L.lep_decompress_batch.argtypes = [vp, P(Bytes), C.c_int, P(Bytes), P(C.c_int32), P(BatchOptions), P(BatchStats)]
        L.lep_jpeg_open_gpu.argtypes = [vp, C.c_size_t, P(vp), P(HuffDecImage), P(C.c_int)]
        L.lep_jpeg_scan_bytes.argtypes = [vp, P(vp), P(C.c_size_t)]
        L.lep_jpeg_finish_gpu.argtypes = [vp, P(HuffDecRow)]
        L.lep_gpu_huffman_decode_device.argtypes = [vp, P(HuffDecImage), C.c_int, vp, vp]
        L.lep_gpu_huffman_decode_simt_device.argtypes = [vp, P(HuffDecImage), C.c_int, vp, vp]
        L.lep_file_recode_plan.argtypes = [vp, P(HuffImage), P(HuffSegment), P(C.c_int), P(C.c_int)]
        L.lep_file_recode_finish.argtypes = [vp, P(Bytes), vp, C.c_int, P(Bytes)]
        L.lep_gpu_huffman_encode_device.argtypes = [vp, P(HuffImage), C.c_int, P(HuffSegment), C.c_int, vp, vp, vp, vp]
        L.lep_batch_release.argtypes = []
        L.lep_batch_release.restype = None
        L.lep_batch_footprint.argtypes = [P(C.c_size_t), P(C.c_size_t)]
        L.lep_batch_footprint.restype = None
        L.lep_file_recode_plan_progressive.argtypes = [vp, P(HuffProgImage), P(HuffProgScan), C.c_int, P(C.c_int), P(C.c_int)]
        L.lep_file_recode_finish_progressive.argtypes = [vp, P(Bytes), C.c_int, P(Bytes)]
        L.lep_gpu_huffman_progressive_encode_device.argtypes = [vp, P(HuffProgImage), C.c_int, P(HuffProgScan), C.c_int, vp, vp, vp, vp]
        L.lep_jpeg_open_gpu_progressive.argtypes = [vp, P(HuffProgDecScan), C.c_int, P(C.c_int), P(C.c_int), P(C.c_int)]
        L.lep_jpeg_finish_gpu_progressive.argtypes = [vp, P(HuffProgDecScan), C.c_int, P(HuffDecRow)]
        L.lep_gpu_huffman_progressive_decode_device.argtypes = [vp, P(HuffProgDecScan), C.c_int, vp, vp]
        L.lep_jpeg_plan_progressive_check.argtypes = [vp, C.c_size_t, P(HuffProgImage), P(HuffProgScan), P(C.c_uint32), P(C.c_uint32), C.c_int, P(C.c_int), P(C.c_int)]
        L.lep_file_consumed.argtypes = [vp]
        L.lep_file_consumed.restype = C.c_size_t
        L.lep_chained_file_follows.argtypes = [vp, C.c_size_t, C.c_size_t]
        L.lep_file_open_next.argtypes = [vp, C.c_size_t, vp, P(vp)]
        L.lep_jpeg_gpu_scan_wait_timeouts.argtypes = []
        L.lep_jpeg_gpu_scan_wait_timeouts.restype = C.c_uint64
        L.lep_batch_debug_poison.argtypes = [C.c_int]
        L.lep_batch_debug_poison.restype = None
        L.lep_jpeg_plan_handoffs.argtypes = [vp, C.c_int, P(Handoff), C.c_int]
        L.lep_handoffs_serialize.argtypes = [P(Handoff), C.c_int, vp, C.c_size_t]
        L.lep_handoffs_parse.argtypes = [vp, C.c_size_t, P(Handoff), C.c_int]
        L.lep_mux.argtypes = [P(Bytes), C.c_int, C.c_int, P(Bytes)]
        L.lep_demux.argtypes = [vp, C.c_size_t, P(Bytes)]
        _lib = L
    return _lib


EXPORTS = [
    "lep_gpu_create", "lep_gpu_destroy", "lep_gpu_last_error", "lep_gpu_device", "lep_gpu_pci_bus_id", "lep_gpu_debug_huffenc", "lep_gpu_encode_host", "lep_gpu_decode_host",
    "lep_gpu_encode_device", "lep_gpu_decode_device", "lep_gpu_sync", "lep_gpu_last_kernel_ms", "lep_gpu_last_kernel_name", "lep_gpu_selftest", "lep_gpu_trim", "lep_gpu_release_memory",
    "lep_gpu_debug_prof", "lep_gpu_malloc",
    "lep_gpu_free", "lep_gpu_memcpy_h2d", "lep_gpu_memcpy_d2h", "lep_gpu_memcpy_d2d", "lep_gpu_memset", "lep_jpeg_open", "lep_jpeg_close",
    "lep_jpeg_describe", "lep_jpeg_plan", "lep_jpeg_write_lep", "lep_file_open", "lep_file_close", "lep_file_describe",
    "lep_file_segments", "lep_file_jpeg_size", "lep_file_recode", "lep_compress", "lep_decompress", "lep_free",
    "lep_version", "lep_jpeg_open_into", "lep_jpeg_peek_frame_bytes", "lep_file_describe_into", "lep_file_frame_bytes", "lep_jpeg_is_progressive", "lep_compress_batch", "lep_decompress_batch", "lep_batch_release", "lep_batch_footprint", "lep_file_recode_plan", "lep_file_recode_finish", "lep_gpu_huffman_encode_device", "lep_jpeg_open_gpu", "lep_jpeg_scan_bytes", "lep_jpeg_scan_restarts", "lep_jpeg_finish_gpu", "lep_gpu_huffman_decode_device", "lep_handoffs_serialize", "lep_handoffs_parse", "lep_mux", "lep_demux",
    "lep_serve_start", "lep_serve_get_stats", "lep_serve_stop", "lep_zlib0_wrap", "lep_jpeg_open_slice", "lep_compress_slice", "lep_jpeg_open_embedded", "lep_compress_embedded", "lep_gpu_use_arena", "lep_gpu_expect_company", "lep_gpu_settle_uploads", "lep_batch_plan", "lep_jpeg_set_encode_options", "lep_gpu_huffman_decode_simt_device",
    "lep_jpeg_check_restores", "lep_jpeg_gpu_scan_wait_timeouts", "lep_batch_debug_poison", "lep_jpeg_plan_handoffs", "lep_file_consumed", "lep_chained_file_follows", "lep_file_open_next",
    "lep_file_recode_plan_progressive", "lep_file_recode_finish_progressive", "lep_gpu_huffman_progressive_encode_device",
    "lep_jpeg_open_gpu_progressive", "lep_jpeg_finish_gpu_progressive", "lep_gpu_huffman_progressive_decode_device",
    "lep_jpeg_plan_progressive_check", "lep_gpu_last_stage_ms", "lep_jpeg_set_container_version", "lep_container_can_write_version", "lep_jpeg_plan_scan_check", "lep_jpeg_scan_file_range",
]
