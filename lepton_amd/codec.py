"""Python mirror of the reference's encoder/decoder interface for the hot path, on top of the C ABI.

Reference interface mirrored (src/lepton/base_coders.hh:26-65):
    BaseEncoder::encode_chunk(const UncompressedComponents*, IOUtil::FileWriter*, const ThreadHandoff*, unsigned)
    BaseDecoder::initialize(DecoderReader*, const std::vector<ThreadHandoff>&) / decode_chunk(UncompressedComponents*)
Here `GpuCodec.encode` takes whole batches of images because on MI355X the unit of parallelism is
(image x thread segment) = one wavefront each, and a useful launch carries hundreds of them.
All computation happens in liblepton_mi355x.so on the GPU; there is no Python or CPU fallback.
"""
import ctypes as C

from . import abi


class LeptonError(RuntimeError):
    """Carries the reference's process exit code (src/vp8/util/memory.hh:13-40)."""

    def __init__(self, code, what):
        super().__init__("%s: exit code %d" % (what, code))
        self.code = code


def _check(rc, what):
    if rc:
        raise LeptonError(rc, what)


class JpegImage:
    """A parsed JPEG: coefficient frame + hand-offs (host memory owned by the library)."""

    def __init__(self, data, allow_progressive=True, start_byte=0, trunc=0, embedding=0):
        """start_byte / trunc: `lepton -startbyte= -trunc=`: the .lep written from this object restores bytes [start_byte, trunc);
        embedding: `lepton -embedding=`: the JPEG starts that many bytes into `data`, the .lep restores all of `data`"""
        self._L = abi.lib()
        self.data = bytes(data[:trunc] if trunc else data)
        self.handle = C.c_void_p()
        if embedding:
            _check(self._L.lep_jpeg_open_embedded(self.data, len(self.data), embedding, C.byref(self.handle)), "lep_jpeg_open_embedded")
        elif start_byte:
            _check(self._L.lep_jpeg_open_slice(self.data, len(self.data), start_byte, C.byref(self.handle)), "lep_jpeg_open_slice")
        else:
            _check(self._L.lep_jpeg_open(self.data, len(self.data), 1 if allow_progressive else 0, C.byref(self.handle)), "lep_jpeg_open")
        self.desc = abi.ImageDesc()
        _check(self._L.lep_jpeg_describe(self.handle, C.byref(self.desc)), "lep_jpeg_describe")

    def plan(self, max_threads=8, image_index=0):
        segs = (abi.Segment * abi.MAX_SEGMENTS)()
        n = self._L.lep_jpeg_plan(self.handle, max_threads, segs, image_index)
        return [segs[i] for i in range(n)]

    def write_lep(self, streams, max_threads=8):
        n = len(streams)
        arr = (abi.Bytes * n)()
        keep = []
        for i, s in enumerate(streams):
            b = C.create_string_buffer(bytes(s), max(1, len(s)))
            keep.append(b)
            arr[i].data = C.cast(b, C.c_void_p).value
            arr[i].len = arr[i].cap = len(s)
        out = abi.Bytes()
        _check(self._L.lep_jpeg_write_lep(self.handle, max_threads, arr, n, C.byref(out)), "lep_jpeg_write_lep")
        data = out.tobytes()
        self._L.lep_free(out.data)
        return data

    def close(self):
        if self.handle:
            self._L.lep_jpeg_close(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LepFile:
    """A parsed .lep: streams + frame geometry; the frame is filled by GpuCodec.decode."""

    def __init__(self, data, prev=None):
        """prev: the previous file of a chained stream of format-version >= 2 files (see lep_stream); it hands over the header
        bytes a `lepton -lepcat` file keeps for its successors"""
        self._L = abi.lib()
        self.data = bytes(data)
        self.handle = C.c_void_p()
        _check(self._L.lep_file_open_next(self.data, len(self.data), prev.handle if prev else None, C.byref(self.handle)), "lep_file_open")
        self.consumed = self._L.lep_file_consumed(self.handle)
        self.more = bool(self._L.lep_chained_file_follows(self.data, len(self.data), self.consumed))
        self.desc = abi.ImageDesc()
        _check(self._L.lep_file_describe(self.handle, C.byref(self.desc)), "lep_file_describe")
        segs = (abi.Segment * abi.MAX_SEGMENTS)()
        st = (abi.Bytes * abi.MAX_SEGMENTS)()
        n = self._L.lep_file_segments(self.handle, segs, st, 0)
        self.segments = [segs[i] for i in range(n)]
        self.streams = [st[i].tobytes() for i in range(n)]

    def recode(self):
        out = abi.Bytes()
        _check(self._L.lep_file_recode(self.handle, C.byref(out)), "lep_file_recode")
        data = out.tobytes()
        self._L.lep_free(out.data)
        return data

    def close(self):
        if self.handle:
            self._L.lep_file_close(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def lep_stream(data):
    """the files of a stream of .lep files written back to back (`cat a.lep b.lep | lepton -`, jpgcoder.cc:1868-1897)"""
    data = bytes(data)
    off, prev, out = 0, None, []
    while True:
        f = LepFile(data[off:], prev)
        out.append(f)
        if not f.more:
            return out
        off += f.consumed
        prev = f


class GpuCodec:
    """The MI355X stand-in for VP8ComponentEncoder / VP8ComponentDecoder (one HIP stream, one device)."""

    def __init__(self, device=0):
        self._L = abi.lib()
        self.handle = C.c_void_p()
        rc = self._L.lep_gpu_create(device, C.byref(self.handle))
        if rc:
            raise LeptonError(rc, "lep_gpu_create (no gfx950 device / HIP runtime: there is no CPU fallback)")

    def last_error(self):
        return (self._L.lep_gpu_last_error(self.handle) or b"").decode()

    def encode(self, images, plans):
        """images: [JpegImage]; plans: per image list of Segment. Returns per image list of stream bytes."""
        L = self._L
        nimg = len(images)
        descs = (abi.ImageDesc * nimg)(*[im.desc for im in images])
        flat = []
        for i, p in enumerate(plans):
            for s in p:
                flat.append(abi.Segment(i, s.luma_y_start, s.luma_y_end, s.is_last))
        nseg = len(flat)
        segs = (abi.Segment * nseg)(*flat)
        out = (abi.Bytes * nseg)()
        bufs = []
        for k, s in enumerate(flat):
            d = images[s.image].desc
            cap = d.total_blocks() * 160 // max(1, len(plans[s.image])) + d.total_blocks() * 16 + 65536
            b = C.create_string_buffer(cap)
            bufs.append(b)
            out[k].data = C.cast(b, C.c_void_p).value
            out[k].cap = cap
        status = (C.c_int32 * nseg)()
        rc = L.lep_gpu_encode_host(self.handle, descs, nimg, segs, nseg, out, status)
        if rc:
            raise LeptonError(rc, "lep_gpu_encode_host [%s]" % self.last_error())
        res, k = [], 0
        for p in plans:
            res.append([bufs[k + j].raw[: out[k + j].len] for j in range(len(p))])
            k += len(p)
        return res

    def decode(self, files):
        """files: [LepFile]; fills each file's coefficient frame on the GPU."""
        L = self._L
        nimg = len(files)
        descs = (abi.ImageDesc * nimg)(*[f.desc for f in files])
        flat, ins, keep = [], [], []
        for i, f in enumerate(files):
            for s, st in zip(f.segments, f.streams):
                flat.append(abi.Segment(i, s.luma_y_start, s.luma_y_end, s.is_last))
                b = C.create_string_buffer(st, max(1, len(st)))
                keep.append(b)
                ins.append((C.cast(b, C.c_void_p).value, len(st)))
        nseg = len(flat)
        segs = (abi.Segment * nseg)(*flat)
        arr = (abi.Bytes * nseg)()
        for k, (p, n) in enumerate(ins):
            arr[k].data, arr[k].len, arr[k].cap = p, n, n
        status = (C.c_int32 * nseg)()
        rc = L.lep_gpu_decode_host(self.handle, descs, nimg, segs, nseg, arr, status)
        if rc:
            raise LeptonError(rc, "lep_gpu_decode_host [%s]" % self.last_error())

    def compress(self, jpg):
        out = abi.Bytes()
        rc = self._L.lep_compress(self.handle, bytes(jpg), len(jpg), C.byref(out))
        if rc:
            raise LeptonError(rc, "lep_compress [%s]" % self.last_error())
        data = out.tobytes()
        self._L.lep_free(out.data)
        return data

    def compress_slice(self, jpg, start_byte, trunc=0):
        out = abi.Bytes()
        _check(self._L.lep_compress_slice(self.handle, jpg, len(jpg), start_byte, trunc, C.byref(out)), "lep_compress_slice")
        data = out.tobytes()
        self._L.lep_free(out.data)
        return data

    def compress_embedded(self, blob, offset):
        out = abi.Bytes()
        _check(self._L.lep_compress_embedded(self.handle, blob, len(blob), offset, C.byref(out)), "lep_compress_embedded")
        data = out.tobytes()
        self._L.lep_free(out.data)
        return data

    def decompress(self, lep):
        out = abi.Bytes()
        rc = self._L.lep_decompress(self.handle, bytes(lep), len(lep), C.byref(out))
        if rc:
            raise LeptonError(rc, "lep_decompress [%s]" % self.last_error())
        data = out.tobytes()
        self._L.lep_free(out.data)
        return data

    def _batch(self, fn, blobs, verify, threads, chunk_bytes, chunk_images=0, host_huffman=False):
        n = len(blobs)
        ins = (abi.Bytes * n)()
        keep = []
        for i, b in enumerate(blobs):
            buf = C.create_string_buffer(bytes(b), max(1, len(b)))
            keep.append(buf)
            ins[i].data = C.cast(buf, C.c_void_p).value
            ins[i].len = ins[i].cap = len(b)
        outs = (abi.Bytes * n)()
        status = (C.c_int32 * n)()
        opt = abi.BatchOptions(threads, 1 if verify else 0, chunk_bytes, 1 if host_huffman else 0, chunk_images)
        stats = abi.BatchStats()
        rc = fn(self.handle, ins, n, outs, status, C.byref(opt), C.byref(stats))
        if rc:
            raise LeptonError(rc, "batch pipeline [%s]" % self.last_error())
        res = []
        for i in range(n):
            res.append(outs[i].tobytes() if not status[i] else None)
            if outs[i].data:
                self._L.lep_free(outs[i].data)
        return res, list(status), {k: getattr(stats, k) for k, _ in abi.BatchStats._fields_}

    def compress_batch(self, jpgs, verify=False, threads=0, chunk_bytes=0, chunk_images=0, host_huffman=False):
        """[jpeg bytes] -> ([.lep bytes or None], [exit code per file], pipeline statistics); the JPEG Huffman scan decode runs
        on the GPU for eligible files unless host_huffman is set"""
        return self._batch(self._L.lep_compress_batch, jpgs, verify, threads, chunk_bytes, chunk_images, host_huffman)

    def decompress_batch(self, leps, threads=0, chunk_bytes=0, chunk_images=0, host_huffman=False):
        """[.lep bytes] -> ([jpeg bytes or None], [exit code per file], pipeline statistics); the JPEG Huffman re-encode runs
        on the GPU for eligible files unless host_huffman is set"""
        return self._batch(self._L.lep_decompress_batch, leps, False, threads, chunk_bytes, chunk_images, host_huffman)

    def close(self):
        if self.handle:
            self._L.lep_gpu_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
