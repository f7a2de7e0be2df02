#!/usr/bin/env python3
"""Which pass of the lane-per-unit scan encoder (lep_huff_simt.h: count / place / code / stuff) two builds of the library disagree in.

  dump:     python scripts/diag_scan_encode_isolate.py dump <out.pkl> <name under tests/golden/ref> [...]     (LEP_LIB_PATH names the build)
            every thread segment of the files alone through lep_gpu_huffman_encode_device; kept per segment: the encoder's work area
            (lep_gpu_debug_huffenc: descriptors, unit positions, plain prefix sums, bit buffer + marker map), the output bytes, the end state
  compare:  python scripts/diag_scan_encode_isolate.py compare <a.pkl> <b.pkl>
            per segment the FIRST region the two builds differ in -- the regions are written by the passes in this order:
              unit_plain (the prefix sum of the units' bit counts: pass 1's counts, summed by pass 2)   -> count differs
              unit position (plain + the pad bits and markers of the intervals in front: pass 2)       -> place differs
              SimtEncSeg.total_bits / cut                                                             -> place differs
              bit buffer / marker map (pass 3) / SimtEncSeg.tail                                      -> code differs
              output bytes, length, end state (pass 4)                                                -> stuff differs
round 5 found `-mllvm -structurizecfg-skip-uniform-regions=1` changing this encoder's output (profiles/r12d_*) without saying where."""
import ctypes as C
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def up(v):
    return (v + 255) & ~255


def dump(out_path, names):
    from lepton_amd import abi
    from lepton_amd.codec import GpuCodec, JpegImage, LepFile

    L = abi.lib()
    L.lep_gpu_debug_huffenc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.lep_gpu_debug_huffenc.restype = C.c_size_t
    codec = GpuCodec(0)
    g = codec.handle
    rec = {"lib": L.lep_version().decode(), "path": os.environ.get("LEP_LIB_PATH", "product"), "segments": {}}
    for name in names:
        jpg = open(os.path.join(ROOT, "tests", "golden", "ref", name + ".jpg"), "rb").read()
        lep = open(os.path.join(ROOT, "tests", "golden", "ref", name + ".lep"), "rb").read()
        f = LepFile(lep)
        src = JpegImage(jpg)
        for c in range(f.desc.ncomp):
            C.memmove(f.desc.blocks[c], src.desc.blocks[c], f.desc.nblocks(c) * 128)
        img = abi.HuffImage()
        segs = (abi.HuffSegment * abi.MAX_SEGMENTS)()
        nseg, ok = C.c_int(0), C.c_int(0)
        assert L.lep_file_recode_plan(f.handle, C.byref(img), segs, C.byref(nseg), C.byref(ok)) == 0
        if not ok.value:
            continue
        dimg = abi.HuffImage.from_buffer_copy(img)
        planes = []
        for c in range(f.desc.ncomp):
            p = C.c_void_p()
            nb = f.desc.nblocks(c) * 128
            assert L.lep_gpu_malloc(g, nb, C.byref(p)) == 0 and L.lep_gpu_memcpy_h2d(g, p, f.desc.blocks[c], nb) == 0
            planes.append(p)
            dimg.blocks[c] = p.value
        for i in range(nseg.value):
            seg = abi.HuffSegment.from_buffer_copy(segs[i])
            seg.out_cap = min(seg.out_cap, len(jpg) + 1024)
            seg.image, seg.out_off = 0, 0
            cap = (seg.out_cap + 15) & ~15
            d_out, d_len, d_end = C.c_void_p(), C.c_void_p(), C.c_void_p()
            assert L.lep_gpu_malloc(g, cap + 256, C.byref(d_out)) == 0 and L.lep_gpu_malloc(g, 16, C.byref(d_len)) == 0 and L.lep_gpu_malloc(g, 32, C.byref(d_end)) == 0
            L.lep_gpu_memset(g, d_out, 0xEE, cap + 256)
            one = (abi.HuffSegment * 1)(seg)
            assert L.lep_gpu_huffman_encode_device(g, C.byref(dimg), 1, one, 1, d_out, d_len, d_end, None) == 0
            L.lep_gpu_sync(g)
            kernel = L.lep_gpu_last_kernel_name(g).decode()
            ln = (C.c_uint32 * 1)()
            end = abi.HuffEnd()
            L.lep_gpu_memcpy_d2h(g, ln, d_len, 4); L.lep_gpu_memcpy_d2h(g, C.byref(end), d_end, C.sizeof(abi.HuffEnd))
            out = C.create_string_buffer(cap + 256)
            L.lep_gpu_memcpy_d2h(g, out, d_out, cap + 256)
            work = C.create_string_buffer(64 << 20)
            nw = L.lep_gpu_debug_huffenc(g, work, len(work)) if "simt" in kernel else 0
            one_rec = dict(kernel=kernel, rows=(seg.mcu_row0, seg.mcu_row1), overhang=seg.overhang, out_cap=seg.out_cap, rsti=img.rsti, mcuh=img.mcuh,
                           length=ln[0], out=out.raw[: ln[0]], end=(end.attempted, end.overhang_byte, end.num_overhang_bits, tuple(end.last_dc), end.pad),
                           work=work.raw[: nw])
            if nw:      # (the bit buffer and the marker map as digests: the dump travels back from the GPU box)
                import hashlib
                r = regions(one_rec)
                for nm in ("buffer", "marker_map"):
                    o, n = r[nm]
                    one_rec["md5_" + nm] = hashlib.md5(one_rec["work"][o: o + n]).hexdigest()
                one_rec["work"] = one_rec["work"][: r["buffer"][0]]
            rec["segments"][(name, i)] = one_rec
            for p in (d_out, d_len, d_end):
                L.lep_gpu_free(g, p)
        for p in planes:
            L.lep_gpu_free(g, p)
    pickle.dump(rec, open(out_path, "wb"))
    print("dumped %d segments from %s (%s)" % (len(rec["segments"]), rec["path"], rec["lib"]))


def regions(seg):
    """the work area of a ONE-segment launch (lep_gpu_huffman_encode_device): [name, offset, size]"""
    import struct

    w = seg["work"]
    _, first_unit, nunits, total_bits, buf_off, buf_bytes, tail, cut, map_bytes = struct.unpack_from("<IIIIQIIII", w, 0)
    nwaves = (nunits + 63) // 64
    o_wv = up(40)
    o_ub = o_wv + up(nwaves * 8)
    o_sc = o_ub + up(nunits * 8)
    return dict(nunits=nunits, total_bits=total_bits, tail=tail, cut=cut, buf_bytes=buf_bytes, map_bytes=map_bytes,
                position=(o_ub, nunits * 4), plain=(o_ub + nunits * 4, nunits * 4), buffer=(o_sc + buf_off, buf_bytes), marker_map=(o_sc + buf_off + buf_bytes, map_bytes))


def compare(pa, pb):
    import struct

    a, b = pickle.load(open(pa, "rb")), pickle.load(open(pb, "rb"))
    print("A: %s (%s)\nB: %s (%s)" % (a["path"], a["lib"], b["path"], b["lib"]))
    verdicts = {}
    for key in sorted(a["segments"]):
        sa, sb = a["segments"][key], b["segments"].get(key)
        if sb is None or "simt" not in sa["kernel"] or "simt" not in sb["kernel"]:
            continue
        ra, rb = regions(sa), regions(sb)
        where = None
        if ra["nunits"] != rb["nunits"]:
            where = "descriptor (host side)"
        else:
            def diff(name):
                (oa, n), (ob_, _) = ra[name], rb[name]
                xa, xb = sa["work"][oa: oa + n], sb["work"][ob_: ob_ + n]
                if xa == xb:
                    return None
                k = next(i for i in range(min(len(xa), len(xb))) if xa[i] != xb[i])
                return k
            names = (("plain", "pass 1, count (the units' bit counts: their prefix sum differs)"), ("position", "pass 2, place (same counts, other positions)"))
            if sa["rsti"] == 0:
                names = names[1:]     # (without restart intervals the second array is not written)
            for name, verdict in names:
                k = diff(name)
                if k is not None and where is None:
                    u = k // 4
                    n_ = ra["nunits"]
                    xa = struct.unpack_from("<%dI" % n_, sa["work"], ra[name][0])
                    xb = struct.unpack_from("<%dI" % n_, sb["work"], rb[name][0])
                    # the units whose own count (difference of consecutive prefix sums) is not the same in the two builds
                    odd = [(q, (xb[q + 1] - xb[q]) - (xa[q + 1] - xa[q])) for q in range(n_ - 1) if xb[q + 1] - xb[q] != xa[q + 1] - xa[q]]
                    # which MCU each of them starts with (SimtUnitMap: the head in runs of eight, then every interval in runs of eight)
                    mcuh, rsti, m_begin = sa["mcuh"], sa["rsti"], sa["rows"][0] * sa["mcuh"]
                    first_end = min(sa["rows"][1] * mcuh, -(-m_begin // rsti) * rsti) if rsti else sa["rows"][1] * mcuh
                    head_units, per = -(-(first_end - m_begin) // 8), (-(-rsti // 8) if rsti else 1)
                    def starts(q):
                        if q < head_units:
                            return m_begin + 8 * q
                        v = q - head_units
                        return first_end + (v // per) * rsti + (v % per) * 8
                    where = "%s -- first at unit %d of %d: A %d, B %d (difference %d bits); units whose count differs [(unit, B - A bits, its first MCU, first of its restart interval)]: %s" % (
                        verdict, u, n_, xa[u], xb[u], xb[u] - xa[u], [(q, dlt, starts(q), bool(rsti) and starts(q) % rsti == 0) for q, dlt in odd[:12]])
            if where is None and (ra["total_bits"], ra["cut"]) != (rb["total_bits"], rb["cut"]):
                where = "pass 2, place (total_bits / cut: A %s, B %s)" % ((ra["total_bits"], ra["cut"]), (rb["total_bits"], rb["cut"]))
            if where is None:
                for name in ("buffer", "marker_map"):
                    if sa.get("md5_" + name) != sb.get("md5_" + name) and where is None:
                        where = "pass 3, code (same positions, the %s differs)" % name
            if where is None and ra["tail"] != rb["tail"]:
                where = "pass 3, code (SimtEncSeg.tail)"
            if where is None and (sa["length"], sa["out"], sa["end"]) != (sb["length"], sb["out"], sb["end"]):
                where = "pass 4, stuff (same bit buffer, other bytes / end state: A %s B %s)" % ((sa["length"], sa["end"]), (sb["length"], sb["end"]))
        same_out = (sa["length"], sa["out"], sa["end"]) == (sb["length"], sb["out"], sb["end"])
        verdicts[key] = where
        if where or not same_out:
            print("%s seg %d rows %s overhang %#x rsti %d (MCU rows of %d): output %s; first difference: %s" % (
                key[0], key[1], sa["rows"], sa["overhang"], sa["rsti"], sa["mcuh"], "same" if same_out else "DIFFERS", where))
    n = sum(1 for v in verdicts.values() if v)
    print("%d segments compared, %d differ" % (len(verdicts), n))


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        names = sys.argv[3:]
        if names == ["ALL"]:
            from conftest import ref_cases
            names = ref_cases(progressive=False)
        dump(sys.argv[2], names)
    else:
        compare(sys.argv[2], sys.argv[3])
