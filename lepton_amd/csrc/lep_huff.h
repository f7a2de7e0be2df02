// lep_huff.h -- JPEG Huffman re-encode of a decoded coefficient frame ON THE GPU (decode direction):
// the step after the arithmetic decoder, which the reference runs on the CPU in recode_one_mcu_row / encode_block_seq
// (src/lepton/recoder.cc:316-412, 245-314).  With it the decoded frame (24.9 MB per 4K image) never crosses PCIe: only
// the 2.2 MB of scan bytes do, and the host just glues header, segments and trailer together (jpeg_recode.cc).
//
// One wavefront per thread segment (the same unit as the coder kernels: a hand-off record gives the bit-exact start
// state -- partial byte, last DC per component -- src/lepton/thread_handoff.hh:8-39).  Per 8x8 block, lane = coefficient
// in zig-zag order:
//   * one coalesced 128-byte load of the AlignedBlock, permuted to zig-zag order by the lane's own table entry;
//   * ballot of the non-zeros -> every lane knows its zero run (previous non-zero = highest set bit below it), its
//     Huffman code (table in LDS) and magnitude bits, i.e. its piece of the block's bit string and its length;
//   * a wave prefix sum of the lengths gives every lane its bit offset; the pieces are OR-ed into an LDS bit buffer;
//   * whole bytes leave the buffer through a second ballot that inserts the 00 after every FF (stuffing positions =
//     popcount of FFs in lower lanes); the trailing bits stay in the buffer for the next block.
// HBM-bound by design: 128 B read + ~12 B written per block, ~100 wave instructions.
// The kernel logic is written on the SPMD layer of lep_wave.h so that tests/emu runs it on the CPU against the host
// re-encoder (jpeg_recode.cc), which in turn is checked against the reference's JPEG bytes.
#pragma once
#include "lep_core.h"
#include "lep_wave.h"

namespace lephuff {
using lepdev::bitlen;

struct HuffImage {          // one image, device-visible
    int32_t ncomp, mcuh, mcuv, mcuc;
    int32_t rsti, padbit;
    uint32_t rst_limit;     // RST markers allowed in the scan (0xffffffff = no limit)
    int32_t interleaved;    // 1: a scan of MCUs of hs x vs blocks per component (a one-component file: mcuh x mcuv = its nch x ncv blocks, hs = vs = 1,
                            // block rows bch apart -- recode_prepare); 0: one component, block `mcu` of a frame without padding blocks (older callers)
    int32_t hs[4], vs[4], bch[4];
    int32_t dc_tbl[4], ac_tbl[4];
    int32_t scan_cmp[4];    // component order inside the MCU
    int32_t trunc_bc[4];    // a file cut inside its scan: blocks of each component in front of the cut (JpegFile::trunc_bc); 0 = whole
    const int16_t* blocks[4];
    uint32_t code[4][256];  // [0..1] DC tables, [2..3] AC tables: length << 16 | code
};

struct HuffSegment {
    int32_t image, mcu_row0, mcu_row1;   // MCU rows [row0, row1)
    uint32_t overhang;                   // overhang_byte | num_overhang_bits << 8
    int16_t last_dc[4];
    uint64_t out_off;                    // into the output arena
    uint32_t out_cap;                    // bytes this segment may produce (BoundedMemWriter bound); more is dropped
    uint32_t pad;
};

struct HuffEnd {             // state a segment's writer ends in: what the next hand-off recorded (recoder.cc:625-640)
    uint32_t attempted;     // bytes the segment tried to write (not clipped to out_cap)
    uint8_t overhang_byte, num_overhang_bits;
    int16_t last_dc[4];
    uint16_t pad;           // kHuffEndCut: the segment stopped at the cut of a truncated file (its bytes are those in front of the cut; overhang_byte /
                            // num_overhang_bits of such an end state are undefined);
                            // kHuffEndRefused: a truncated file's segment this kernel does not take -- the host re-coder's
};
constexpr uint16_t kHuffEndCut = 1, kHuffEndRefused = 2;
constexpr uint32_t kHuffSegSimt = 1, kHuffSegRefuse = 2;   // HuffSegment::pad: which kernel owns the segment

struct HuffShared {
    uint32_t code[4][256];
    uint32_t bits[72];      // MSB-first bit buffer of the block being coded (+ carried partial byte in word 0)
    uint8_t z2a[64];
};

#ifdef __HIP_DEVICE_COMPILE__
#define LEPH_TABLE __constant__ static const
#else
#define LEPH_TABLE static const
#endif
LEPH_TABLE uint8_t kZ2A[64] = {   // zig-zag index -> aligned index (aligned_block.hh:32-76)
    49, 50, 57, 58, 0, 51, 52, 1, 2, 59, 60, 3, 4, 5, 53, 54, 6, 7, 8, 9, 61, 62, 10, 11, 12, 13, 14, 55, 56, 15, 16, 17,
    18, 19, 20, 63, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48};

// the same table where a compile-time index needs it (lep_huff_simt.h unrolls a block's 63 positions)
constexpr int kZ2A_const(int k) {
    constexpr uint8_t t[64] = {49, 50, 57, 58, 0, 51, 52, 1, 2, 59, 60, 3, 4, 5, 53, 54, 6, 7, 8, 9, 61, 62, 10, 11, 12, 13, 14, 55, 56, 15, 16, 17,
                               18, 19, 20, 63, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48};
    return t[k];
}

WDEV void lds_or(uint32_t* p, uint32_t v) {
#if LEP_ON_GPU
    __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    *p |= v;
#endif
}

struct HuffWave {
    const HuffImage* img;
    HuffShared* sh;
    uint8_t* out;
    uint32_t cap;
    uint32_t written;     // bytes attempted so far (stores are clipped to cap)
    int pend;             // bits waiting in sh->bits[0] (top-aligned), 0..7
    int lastdc[4];

    // OR an n-bit field (n <= 32, right-aligned in v) into the bit buffer at bit position p
    WDEV void put_field(uint32_t v, int n, int p) {
        const int d = p >> 5, shft = p & 31;
        const uint64_t v64 = (uint64_t)v << (64 - n - shft);
        lds_or(&sh->bits[d], (uint32_t)(v64 >> 32));
        const uint32_t lo = (uint32_t)v64;
        if (lo) lds_or(&sh->bits[d + 1], lo);
    }
    // move the whole bytes of the first `total_bits` bits of the buffer to the output, stuffing 00 after FF; keeps the rest
    WDEV void flush_bytes(int total_bits) {
        const int nb = total_bits >> 3;
        for (int i0 = 0; i0 < nb; i0 += 64) {
            LV(int, ff); LV(uint32_t, bytev);
            LANES(l) {
                const int i = i0 + l;
                uint32_t b = 0;
                if (i < nb) b = (sh->bits[i >> 2] >> (24 - 8 * (i & 3))) & 255u;
                L(bytev) = b; L(ff) = i < nb && b == 0xff;
            }
            const uint64_t ffm = lepwave::wave_ballot(ff);
            LANES(l) {
                const int i = i0 + l;
                if (i < nb) {
                    const uint32_t pos = written + (uint32_t)l + (uint32_t)lepwave::popc64(ffm & ((1ull << l) - 1));
                    if (pos < cap) out[pos] = (uint8_t)L(bytev);
                    if (L(ff) && pos + 1 < cap) out[pos + 1] = 0;
                }
            }
            written += (uint32_t)(nb - i0 < 64 ? nb - i0 : 64) + (uint32_t)lepwave::popc64(ffm);
        }
        // carry the partial byte to word 0, clear the rest
        const int rem = total_bits & 7;
        uint32_t carry = 0;
        if (rem) carry = ((sh->bits[nb >> 2] >> (24 - 8 * (nb & 3))) & 255u) << 24;
        LSYNC();
        LANES(l) { sh->bits[l] = l == 0 ? carry : 0u; if (l < 8) sh->bits[64 + l] = 0u; }
        LSYNC();
        pend = rem;
    }
    // abitwriter::pad (src/lepton/bitops.hh): fill the current byte with the pad-bit pattern, LSB of the pattern first
    WDEV void pad_byte(int padbit) {
        if (!pend) return;
        const int n = 8 - pend;
        uint32_t v = 0;
        for (int j = 0; j < n; ++j) v = (v << 1) | (uint32_t)((padbit >> j) & 1);
        LANES(l) if (l == 0) put_field(v, n, pend);
        LSYNC();
        flush_bytes(8);
    }
    WDEV void raw_bytes2(uint8_t a, uint8_t b) {   // marker bytes: no stuffing
        LANES(l) if (l == 0) {
            if (written < cap) out[written] = a;
            if (written + 1 < cap) out[written + 1] = b;
        }
        written += 2;
    }

    // one block (encode_block_seq, recoder.cc:245-314)
    WDEV void code_block(int cmp, int dpos) {
        HuffShared& S = *sh;
        const int16_t* blk = img->blocks[cmp] + (int64_t)dpos * 64;
        const int dct = img->dc_tbl[cmp], act = 2 + img->ac_tbl[cmp];
        LV(int, tv); LV(int, nzf);
        LANES(l) L(tv) = blk[S.z2a[l]];
        const int dc = (int16_t)lepwave::wave_read((const uint32_t*)tv, 0);
        const int diff = (int16_t)(dc - lastdc[cmp]);
        lastdc[cmp] = dc;
        LANES(l) { if (l == 0) L(tv) = diff; L(nzf) = L(tv) != 0; }
        const uint64_t m = lepwave::wave_ballot(nzf);
        const uint64_t acm = m & ~1ull;
        const int end = acm ? 63 - __builtin_clzll(acm) : 0;
        const uint32_t zrl = S.code[act][0xF0];
        const int zrl_len = (int)(zrl >> 16);
        LV(int, total); LV(int, off); LV(int, nn); LV(uint32_t, fb); LV(int, kk);
        LANES(l) {
            const int t = L(tv);
            const int at = (t < 0 ? -t : t) & 0xffff;
            const int s = bitlen((uint32_t)at);
            const uint32_t val = (uint32_t)((t > 0) ? t : (t - 1) + (1 << s)) & ((1u << s) - 1u);
            int n = 0, k = 0;
            uint32_t bits = 0;
            if (l == 0) {
                const uint32_t e = S.code[dct][s & 255];
                n = (int)(e >> 16) + s; bits = ((e & 0xffffu) << s) | val;
            } else if (t != 0) {
                const uint64_t pm = acm & ((1ull << l) - 1);
                const int prev = pm ? 63 - __builtin_clzll(pm) : 0;
                const int run = l - prev - 1;
                k = run >> 4;
                const uint32_t e = S.code[act][(((run & 15) << 4) + s) & 255];
                n = (int)(e >> 16) + s; bits = ((e & 0xffffu) << s) | val;
            } else if (l == end + 1 && end != 63) {
                const uint32_t e = S.code[act][0];
                n = (int)(e >> 16); bits = e & 0xffffu;
            }
            L(nn) = n; L(fb) = bits; L(kk) = k; L(total) = n + k * zrl_len;
        }
        const int B = lepwave::wave_excl_scan(total, off);
        LANES(l) if (L(total)) {
            int p = pend + L(off);
            for (int i = 0; i < L(kk); ++i) { put_field(zrl & 0xffffu, zrl_len, p); p += zrl_len; }
            if (L(nn)) put_field(L(fb), L(nn), p);
        }
        LSYNC();
        flush_bytes(pend + B);
    }

    // recode_one_mcu_row (recoder.cc:316-412) for MCU rows [row0, row1) of one segment; returns bytes produced (clipped to cap)
    WDEV uint32_t run(const HuffImage* image, const HuffSegment& seg, HuffShared* shared, uint8_t* arena) {
        img = image; sh = shared; out = arena + seg.out_off; cap = seg.out_cap; written = 0;
        LANES(l) {
            for (int i = l; i < 1024; i += 64) (&sh->code[0][0])[i] = (&img->code[0][0])[i];
            sh->z2a[l] = kZ2A[l];
            sh->bits[l] = l == 0 ? (uint32_t)(seg.overhang & 255u) << 24 : 0u;
            if (l < 8) sh->bits[64 + l] = 0u;
        }
        LSYNC();
        pend = (int)((seg.overhang >> 8) & 255u);
        for (int c = 0; c < 4; ++c) lastdc[c] = seg.last_dc[c];
        const int ncomp = img->ncomp, mcuh = img->mcuh, rsti = img->rsti;
        for (int row = seg.mcu_row0; row < seg.mcu_row1; ++row) {
            int mcu = row * mcuh;
            int rstw = rsti ? rsti - mcu % rsti : 0;
            uint32_t cum_rst = rstw ? (uint32_t)(mcu / rsti) : 0u;
            for (int mx = 0; mx < mcuh; ++mx, ++mcu) {
                if (img->interleaved) {
                    for (int ci = 0; ci < ncomp; ++ci) {
                        const int cmp = img->scan_cmp[ci];
                        const int hs = img->hs[cmp], vs = img->vs[cmp], bch = img->bch[cmp];
                        for (int v = 0; v < vs; ++v)
                            for (int h = 0; h < hs; ++h) code_block(cmp, (row * vs + v) * bch + mx * hs + h);
                    }
                } else {
                    code_block(img->scan_cmp[0], mcu);
                }
                // next_mcupos (jpgcoder.cc): restart interval / end of scan after this MCU
                int sta = 0;
                if (mcu + 1 >= img->mcuc) sta = 2;
                else if (rsti > 0 && --rstw == 0) sta = 1;
                if (sta) {
                    pad_byte(img->padbit);
                    if (sta == 1) {
                        if (cum_rst < img->rst_limit) { raw_bytes2(0xFF, (uint8_t)(0xD0 + (cum_rst & 7u))); ++cum_rst; }
                        rstw = rsti;
                        lastdc[0] = lastdc[1] = lastdc[2] = lastdc[3] = 0;
                    }
                }
            }
        }
        return written < cap ? written : cap;
    }
    // the partial byte, its bit count and the last DCs this segment ends in (lane 0 stores them)
    WDEV void export_end(HuffEnd* e) const {
        LANES(l) if (l == 0) {
            e->attempted = written;
            e->overhang_byte = (uint8_t)(pend ? sh->bits[0] >> 24 : 0u);
            e->num_overhang_bits = (uint8_t)pend;
            for (int c = 0; c < 4; ++c) e->last_dc[c] = (int16_t)lastdc[c];
            e->pad = 0;
        }
    }
};

}  // namespace lephuff
