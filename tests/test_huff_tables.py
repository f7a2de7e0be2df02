"""The host parser's Huffman tables -- code words as sorted intervals of 16-bit patterns (jpeg_scan.cc build_huff_table, jpeg_bits.h
next_huffcode) -- against the reference's tree as oracle/jpeg_huff_tree.h restates it (jpgcoder.cc:5507-5606, 5407-5425): same
refusals, same code per symbol, and for EVERY 16-bit pattern at four distances from the end of the data the same symbol, the same
number of bits taken and the same end-of-data flag; then runs of codes over random bytes from unaligned starts.  Tables that follow
T.81 Annex C, and tables that do not (codes that extend other codes, symbols listed twice, more inner nodes than the reference's
tree holds, segments cut short): a .lep header is taken as it was written, so those must decode as the reference decodes them."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "emu", "libhuff_table_check.so")


@pytest.fixture(scope="module")
def chk():
    from lepton_amd import abi
    abi.lib()
    src = os.path.join(ROOT, "tests", "emu", "huff_table_check.cc")
    deps = [src, os.path.join(ROOT, "oracle", "jpeg_huff_tree.h"), os.path.join(ROOT, "lepton_amd", "liblepton_mi355x.so")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        tmp = "%s.%d" % (SO, os.getpid())
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", tmp, src, "-L" + os.path.join(ROOT, "lepton_amd"),
                               "-llepton_mi355x", "-Wl,-rpath," + os.path.join(ROOT, "lepton_amd")])
        os.replace(tmp, SO)
    lib = C.CDLL(SO)
    lib.huff_table_check.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    lib.huff_table_words.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    return lib


def check(lib, counts, syms, strict, rng, counts_avail=None, syms_avail=None):
    counts = bytes(counts)
    syms = bytes(syms)
    stream = rng.integers(0, 256, 96, dtype=np.uint8).tobytes()
    detail = C.c_int(0)
    rc = lib.huff_table_check(counts, len(counts) if counts_avail is None else counts_avail, syms, len(syms) if syms_avail is None else syms_avail,
                              strict, stream, len(stream), C.byref(detail))
    assert rc == 0, "kind %d, detail 0x%x, counts %s, symbols %s" % (rc, detail.value & 0xffffffff, list(counts), list(syms))


def annex_c_counts(rng, nsyms, longest=16):
    """code lengths of a prefix-free table with the all-ones code left free, as an encoder would write them: a full code tree grown by
    splitting random leaves until it has one leaf more than there are symbols, the deepest leaf left unused"""
    leaves = [1, 1]
    while len(leaves) < nsyms + 1:
        open_ = [i for i, d in enumerate(leaves) if d < longest]
        if not open_:
            longest += 1
            continue
        i = open_[int(rng.integers(0, len(open_)))]
        leaves[i] += 1
        leaves.append(leaves[i])
    lens = np.sort(np.array(leaves))[:nsyms]
    return [int((lens == b).sum()) for b in range(1, 17)]


STD_DC_LUMA = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
STD_AC_LUMA = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d],
               [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
                0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
                0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
                0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
                0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
                0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
                0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
                0xfa])


def test_the_tables_of_annex_k(chk):
    rng = np.random.default_rng(1)
    for counts, syms in (STD_DC_LUMA, STD_AC_LUMA):
        for strict in (0, 1):
            check(chk, counts, syms, strict, rng)
    assert chk.huff_table_words(bytes(STD_AC_LUMA[0]), 16, bytes(STD_AC_LUMA[1]), 162, 1) == 162


def test_tables_that_follow_annex_c(chk):
    rng = np.random.default_rng(2)
    for trial in range(40):
        n = int(rng.integers(1, 257))
        counts = annex_c_counts(rng, n, longest=int(rng.integers(2, 17)))
        syms = rng.permutation(256)[:n].astype(np.uint8)
        check(chk, counts, syms, trial & 1, rng)


def test_tables_that_need_more_inner_nodes_than_the_tree_has(chk):
    """prefix-free tables whose codes spread out: past 255 inner nodes the reference refuses the table as a JPEG and, from a .lep header,
    reads the overflowing node numbers as symbols and drops the codes that needed them"""
    rng = np.random.default_rng(3)
    overflowing = 0
    for counts in ([0] * 7 + [128] + [0] * 7 + [127], [0] * 7 + [128] + [0] * 7 + [100], [0] * 7 + [2, 4, 8, 16, 32, 64, 64, 50, 15],
                   [0] * 3 + [7] + [0] * 3 + [60] + [0] * 3 + [80] + [0] * 3 + [100], [0] * 9 + [3, 9, 27, 60, 60, 60, 30]):
        n = sum(counts)
        for rep in range(3):
            syms = rng.permutation(256)[:n].astype(np.uint8)
            for strict in (0, 1):
                check(chk, counts, syms, strict, rng)
            if chk.huff_table_words(bytes(counts), 16, syms.tobytes(), n, 1) < 0:
                assert 0 < chk.huff_table_words(bytes(counts), 16, syms.tobytes(), n, 0) < n
                overflowing += 1
    assert overflowing >= 6


def test_tables_that_break_annex_c(chk):
    rng = np.random.default_rng(4)
    for trial in range(60):
        kind = trial % 4
        if kind == 0:      # over-full: the code counter runs past its length
            counts = rng.integers(0, 6, 16)
        elif kind == 1:    # sparse and long
            counts = np.zeros(16, dtype=np.int64)
            counts[rng.integers(8, 16, 5)] = rng.integers(1, 60, 5)
        elif kind == 2:    # more than 256 codes: the symbol index wraps
            counts = rng.integers(0, 40, 16)
        else:              # short codes early, everything after them extends one
            counts = np.concatenate([rng.integers(1, 3, 3), rng.integers(0, 12, 13)])
        counts = np.minimum(counts, 255).astype(np.uint8)
        n = int(counts.sum())
        syms = rng.integers(0, 256 if trial % 3 else 24, max(n, 1), dtype=np.uint8)       # symbols repeat
        check(chk, counts, syms[:256], 0, rng)
        check(chk, counts, syms[:256], 1, rng)


def test_segments_cut_short(chk):
    rng = np.random.default_rng(5)
    for trial in range(12):
        counts = annex_c_counts(rng, 60)
        syms = rng.permutation(256)[:60].astype(np.uint8)
        check(chk, counts, syms, 0, rng, counts_avail=int(rng.integers(0, 17)), syms_avail=int(rng.integers(0, 61)))
    check(chk, [0] * 16, b"", 0, rng)
    check(chk, [0] * 16, b"", 1, rng)
