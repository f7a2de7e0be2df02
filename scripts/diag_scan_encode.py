#!/usr/bin/env python3
"""GPU box: lep_gpu_huffman_encode_device on one file's frame, segment by segment, against the lane-loop emulation of the same kernels
(tests/emu/libcore_emu.so).  usage: python scripts/diag_scan_encode.py <name under tests/golden/ref> [...]   (LEP_LIB_PATH / LEP_HUFFENC_SIMT as usual)"""
import ctypes as C
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from lepton_amd import abi
    from lepton_amd.codec import GpuCodec, JpegImage, LepFile

    so = os.path.join(ROOT, "tests", "emu", "libcore_emu.so")
    if not os.path.exists(so):      # (git- and gpurun-ignored: built where it is needed, like the test fixture does)
        import subprocess
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "emu", "core_emu.cc")])
    emu = C.CDLL(so)
    L = abi.lib()
    codec = GpuCodec(0)
    g = codec.handle
    bad = 0
    names = sys.argv[1:]
    if names == ["ALL"]:
        from conftest import ref_cases
        names = ref_cases(progressive=False)
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
            print(name, "not eligible"); continue
        n = nseg.value
        # emulation, host frame
        want = []
        for i in range(n):
            segs[i].out_cap = min(segs[i].out_cap, len(jpg) + 1024)
            buf = C.create_string_buffer(segs[i].out_cap + 8)
            ln = C.c_uint32(0)
            end = abi.HuffEnd()
            if emu.emu_huffman_encode_segment_simt(C.byref(img), C.byref(segs[i]), buf, C.byref(ln), C.byref(end)) != 0:
                assert emu.emu_huffman_encode_segment(C.byref(img), C.byref(segs[i]), buf, C.byref(ln), C.byref(end)) == 0
            want.append((ln.value, hashlib.md5(buf.raw[: ln.value]).hexdigest()[:8], end.attempted, end.overhang_byte, end.num_overhang_bits, list(end.last_dc)[:3], end.pad, buf.raw[: ln.value]))
        # device: frame planes, output arena, lens, ends
        dimg = abi.HuffImage.from_buffer_copy(img)
        planes = []
        for c in range(f.desc.ncomp):
            p = C.c_void_p()
            nb = f.desc.nblocks(c) * 128
            assert L.lep_gpu_malloc(g, nb, C.byref(p)) == 0
            assert L.lep_gpu_memcpy_h2d(g, p, f.desc.blocks[c], nb) == 0
            planes.append(p)
            dimg.blocks[c] = p.value
        dsegs = (abi.HuffSegment * n)()
        off = 0
        for i in range(n):
            C.memmove(C.byref(dsegs[i]), C.byref(segs[i]), C.sizeof(abi.HuffSegment))
            dsegs[i].image = 0
            dsegs[i].out_off = off
            off += (segs[i].out_cap + 15) & ~15
        d_out, d_len, d_end = C.c_void_p(), C.c_void_p(), C.c_void_p()
        assert L.lep_gpu_malloc(g, off + 256, C.byref(d_out)) == 0 and L.lep_gpu_malloc(g, 4 * n, C.byref(d_len)) == 0 and L.lep_gpu_malloc(g, 16 * n, C.byref(d_end)) == 0
        for rep in range(int(os.environ.get("REPS", "3"))):
            assert L.lep_gpu_memset(g, d_out, 0xEE, off + 256) == 0
            rc = L.lep_gpu_huffman_encode_device(g, C.byref(dimg), 1, dsegs, n, d_out, d_len, d_end, None)
            assert rc == 0, rc
            L.lep_gpu_sync(g)
            lens = (C.c_uint32 * n)()
            ends = (abi.HuffEnd * n)()
            out = C.create_string_buffer(off + 256)
            L.lep_gpu_memcpy_d2h(g, lens, d_len, 4 * n); L.lep_gpu_memcpy_d2h(g, ends, d_end, 16 * n); L.lep_gpu_memcpy_d2h(g, out, d_out, off + 256)
            for i in range(n):
                b = out.raw[dsegs[i].out_off: dsegs[i].out_off + lens[i]]
                got = (lens[i], hashlib.md5(b).hexdigest()[:8], ends[i].attempted, ends[i].overhang_byte, ends[i].num_overhang_bits, list(ends[i].last_dc)[:3], ends[i].pad)
                same = got == want[i][:7]
                if not same:
                    bad += 1
                    first = next((k for k in range(min(len(b), len(want[i][7]))) if b[k] != want[i][7][k]), None)
                    print("%s rep %d seg %d rows %d..%d  DIFFERS\n   gpu  %s\n   emu  %s\n   first differing byte %s of %d: gpu %s emu %s" % (
                        name, rep, i, segs[i].mcu_row0, segs[i].mcu_row1, got, want[i][:7], first, len(b),
                        b[first - 4: first + 12].hex() if first is not None else "-", want[i][7][first - 4: first + 12].hex() if first is not None else "-"))
                elif rep == 0:
                    print("%s seg %d rows %d..%d ok %s" % (name, i, segs[i].mcu_row0, segs[i].mcu_row1, got))
        print(name, "kernel:", L.lep_gpu_last_kernel_name(g).decode())
        for p in planes + [d_out, d_len, d_end]:
            L.lep_gpu_free(g, p)
    print("DIFFERING SEGMENT RESULTS:", bad)


if __name__ == "__main__":
    main()
