// lep_huffdec.h -- JPEG Huffman scan DECODE on the GPU (encode direction): the step in front of the
// arithmetic encoder, which the reference runs on the CPU in decode_jpeg / decode_block_seq
// (src/lepton/jpgcoder.cc:2799-3302, 4893-4966) and which is 95 % of our host parser's time.  With it the host only splits
// the file (markers, FF00 un-stuffing: memcpy speed) and 2.2 MB of scan bytes cross PCIe instead of a 24.9 MB frame.
//
// A sequential JPEG scan has no entry points (every code's position depends on all codes before it), so the unit of
// parallelism is the image: one wavefront per image, thousands of images per launch.  The wave decodes serially as
// uniform vector code (the recurrence is "window -> table entry -> shift", a scalar chain like the bool coder's); the
// lanes are used for what is parallel: the decoded block is assembled in LDS and leaves as one coalesced 128-byte
// store in the coder's AlignedBlock order, tables live in LDS (first-level look-up: 9 bits for AC codes, 8 for DC; longer
// codes are resolved in ONE step by seven lanes testing "are the first L bits <= the largest code of length L" side by
// side, the canonical-code test of Annex F.2.2.3 turned sideways).  The wave runs at raised priority with <= 64 VGPRs and
// 4.5 KB of LDS so that it fits beside seven coder wavefronts per SIMD: the batch pipeline decodes chunk k+1 in the wave
// slots that the arithmetic coder of chunk k leaves free (lep_batch.hip).
// Per MCU row the wave records (bit position, last DC per component): exactly what the ThreadHandoff records of the
// .lep header are made of (src/lepton/jpgcoder.cc:2520-2560); the host turns them into file offsets / overhang bits.
// Anything irregular (decode error, zero run past the block, data running out, grey or non-interleaved scans,
// progressive files) makes the image's status non-zero and the host parser (jpeg_scan.cc) takes that file.
#pragma once
#include "lep_enc3.h"   // lep3::vec / ucond / uni
#include "lep_huff.h"   // kZ2A

namespace lephuff {
using lep3::ucond;
using lep3::uni;
using lep3::vec;

constexpr int32_t kHuffDecEarlyEof = 1;              // HuffDecImage::flags
constexpr int32_t kHuffDecRstTable = 2;              // ... the restart positions follow the scan bytes at scan + huffdec_scan_room(scan_len)
WDEV uint32_t huffdec_scan_room(uint32_t scan_len) { return (scan_len + 64u + 15u) & ~15u; }
constexpr int32_t kHuffDecRowTruncated = 0x40000000; // final HuffDecRow::aux: the data ran out in mid-image; bitpos = blocks decoded
constexpr int32_t kHuffDecRowUnwritten = (int32_t)0x80000000;   // ... set in front of the lane-per-piece write pass: no lane has written the record
struct HuffDecImage {       // one image, device-visible
    const uint8_t* scan;    // un-stuffed entropy-coded bytes (RSTn removed), 16-byte aligned, followed by >= 16 zero bytes
    uint32_t scan_len;
    int32_t ncomp, mcuh, mcuv, mcuc, rsti;
    int32_t flags;          // kHuffDecEarlyEof: the file ends inside its scan
    int32_t reserved0;
    int32_t hs[4], vs[4], bch[4], dc_tbl[4], ac_tbl[4], scan_cmp[4];
    int16_t* blocks[4];     // zero-filled frame (device)
    uint64_t rows_off;      // this image's first record in the row arena (mcuv + 1 records)
    uint16_t lut[4][512];   // [0..1] DC, [2..3] AC: first 9 bits of the window -> code length << 8 | symbol; 0 = longer code
    // codes of 9..16 bits ([k] = length 9 + k), canonical order: a window whose first L bits are <= maxcode is a code of
    // length L (no shorter one matched), its symbol is longsym[valoff + those bits]; maxcode = -1: no code of that length
    int32_t maxcode[4][8];
    int32_t valoff[4][8];
    uint8_t longsym[4][256];
};

struct HuffDecRow {         // one per MCU row + one final
    uint32_t bitpos;        // position of the next unread bit in the un-stuffed scan
    int16_t last_dc[4];
    int32_t aux;            // final record: padbit in bits 0..7 (0xff = never determined), status in bits 8..
};

struct HuffDecShared {      // 4544 bytes
    uint16_t lut_ac[2][512];
    uint16_t lut_dc[2][256];   // DC codes: 8-bit first level (a 9-bit DC code takes the long path)
    int32_t maxcode[4][8];
    int32_t valoff[4][8];
    uint8_t longsym[4][256];
    int16_t blk[64];        // aligned order
    uint8_t z2a[64];
};

struct HuffDecWave {
    const HuffDecImage* img;
    HuffDecShared* sh;
    // bit reader (uniform vector): 64-bit top-aligned window, refilled with aligned big-endian dwords
    uint32_t hi, lo;
    int navail;
    uint32_t wi;            // index of the dword `ahead` holds (scalar)
    uint32_t ahead;         // that dword, requested one refill before it is needed: the load's latency (a serial chain has nothing
                            // else to hide it behind) passes while the window's current 32 bits are decoded
    uint32_t bitpos;        // bits consumed
    int status;

    WDEV uint32_t fetch(uint32_t k) const {
        const uint32_t* words = reinterpret_cast<const uint32_t*>(img->scan);
        // data past the end reads as zero (the arena is zero padded); running past it is detected through bitpos
        return k * 4 < img->scan_len + 16 ? lep3_vload(words + k) : 0u;
    }
    WDEV void start_reader(uint32_t first_word) { wi = first_word; ahead = fetch(first_word); }
    WDEV void refill() {
        const uint32_t w = __builtin_bswap32(ahead);
        ++wi;
        ahead = fetch(wi);
        const uint64_t add = ((uint64_t)w << 32) >> navail;
        hi |= (uint32_t)(add >> 32); lo |= (uint32_t)add;
        navail += 32;
    }
    static WDEV uint32_t lep3_vload(const uint32_t* p) {
#if LEP_ON_GPU
        uintptr_t a = (uintptr_t)p;
        __asm__ volatile("" : "+v"(a));
        return *reinterpret_cast<const uint32_t*>(a);
#else
        return *p;
#endif
    }
    WDEV void consume(uint32_t n) {
        const uint64_t v = (((uint64_t)hi << 32) | lo) << n;
        hi = (uint32_t)(v >> 32); lo = (uint32_t)v;
        navail -= (int)n; bitpos += n;
        if (ucond(navail <= 32)) refill();
    }
    WDEV uint32_t read(uint32_t n) {   // n <= 16
        const uint32_t v = n ? hi >> (32 - n) : 0u;
        consume(n);
        return v;
    }
    // a code of 9..16 bits: lane k tests length 9 + k; the shortest length that fits is the code (canonical Huffman codes:
    // the codes of one length are consecutive and larger than every shorter code extended with zeros).  -1 = not a code.
    WDEV int symbol_long(int t, uint32_t* len) {
        LV(int, ok);
        LV(uint32_t, cand);
        LANES(l) {
            const int k = l & 7;
            const uint32_t code = hi >> (23 - k);
            L(ok) = l < 8 && (int)code <= sh->maxcode[t][k];
            L(cand) = (uint32_t)(sh->valoff[t][k] + (int)code);
        }
        const uint64_t m = lepwave::wave_ballot(ok);
        if (!m) return -1;
        const int k = __builtin_ctzll(m);
        *len = 9u + (uint32_t)k;
        return (int)sh->longsym[t][lepwave::wave_read(cand, k) & 255u];
    }
    static WDEV int devli(uint32_t s, uint32_t n) { return s == 0 ? (int)n : (n >= (1u << (s - 1)) ? (int)n : (int)n + 1 - (1 << s)); }

    // Huffman symbol of table t AND the `s` magnitude bits that follow it in one window step (code <= 16 bits, s <= 15:
    // both inside the top 32 bits of the window).  dc: s = symbol, else s = low nibble.  Returns the symbol (-1 invalid)
    // and the raw magnitude bits through *n.
    WDEV int symbol_and_bits(int t, bool dc, uint32_t* n) {
        const uint32_t e = dc ? sh->lut_dc[t][hi >> 24] : sh->lut_ac[t - 2][hi >> 23];
        uint32_t len = e >> 8, sym = e & 255u;
        if (ucond(len == 0)) {
            const int r = symbol_long(t, &len);
            if (r < 0) return -1;
            sym = (uint32_t)r;
        }
        const uint32_t s = dc ? sym : (sym & 15u);
        if (ucond(s > 15)) return -1;
        *n = s ? (hi << len) >> (32 - s) : 0u;
        consume(len + s);
        return (int)sym;
    }

    // decode_block_seq (jpgcoder.cc:4893-4966) into sh->blk; returns the DC difference through *diff; false = irregular
    WDEV bool decode_block(int dct, int act, int* diff) {
        uint32_t n = 0;
        int hc = symbol_and_bits(dct, true, &n);
        if (ucond(hc < 0)) return false;
        *diff = devli((uint32_t)hc & 255u, n);
        uint32_t bpos = vec(1);
        uint32_t last_s = vec(1);                     // magnitude category of the last coefficient written (0 = a coded zero)
#pragma nounroll
        while (ucond(bpos < 64)) {
            hc = symbol_and_bits(act, false, &n);
            if (ucond(hc < 0)) return false;
            if (ucond(hc == 0)) break;                // EOB
            const uint32_t z = ((uint32_t)hc >> 4) & 15u, s = (uint32_t)hc & 15u;
            if (ucond(z + bpos >= 64)) return false;  // zero run past the block: the host parser knows what the reference does
            bpos += z;
            sh->blk[sh->z2a[bpos]] = (int16_t)devli(s, n);
            ++bpos;
            last_s = s;
        }
        // a block whose last coded coefficient is zero (ZRL + EOB, a run/size symbol with size 0): the re-encoder would write
        // it differently, and the reference refuses the file ("cannot encode image with eob after last 0", jpgcoder.cc:2950)
        return !ucond(last_s == 0);
    }

    // BitReader::unpad (bitops.hh): the pad-bit pattern of the current partial byte, consuming it
    WDEV int unpad(int fillbit) {
        const int rem = (int)(uni(bitpos) & 7u);
        if (!rem) return fillbit;
        int nb = 8 - rem, last = (int)uni(read(1)), off = 1;
        int f = last;
        for (int i = 1; i < nb; ++i) { last = (int)uni(read(1)); f |= last << off; ++off; }
        while (off < 7) { f |= last << off; ++off; }
        return f & 255;
    }

    WDEV void run(const HuffDecImage* image, HuffDecShared* shared, HuffDecRow* rows_arena) {
        img = image; sh = shared; status = 0;
        LANES(l) {
            for (int i = l; i < 2 * 512; i += 64) (&sh->lut_ac[0][0])[i] = (&img->lut[2][0])[i];
            for (int i = l; i < 2 * 256; i += 64) {   // 8-bit DC table from the 9-bit one: entries of codes up to 8 bits come in pairs
                const uint16_t e = img->lut[i >> 8][(i & 255) * 2];
                (&sh->lut_dc[0][0])[i] = (e >> 8) <= 8 ? e : (uint16_t)0;
            }
            if (l < 32) { (&sh->maxcode[0][0])[l] = (&img->maxcode[0][0])[l]; (&sh->valoff[0][0])[l] = (&img->valoff[0][0])[l]; }
            for (int i = l; i < 4 * 256; i += 64) (&sh->longsym[0][0])[i] = (&img->longsym[0][0])[i];
            sh->z2a[l] = kZ2A[l];
            sh->blk[l] = 0;
        }
        LSYNC();
        hi = vec(0); lo = vec(0); navail = 0; bitpos = vec(0);
        start_reader(0);
        refill(); refill();
        HuffDecRow* rows = rows_arena + img->rows_off;
        int lastdc[4] = {0, 0, 0, 0};
        int padbit = -1;
        const int ncomp = img->ncomp, mcuh = img->mcuh, mcuv = img->mcuv, rsti = img->rsti;
        int rstw = rsti;
        int mcu = 0;
        for (int row = 0; row < mcuv && !status; ++row) {
            // hand-off record of this MCU row (make_handoff, jpeg_scan.cc): where the row starts and the DC predictors there
            {
                const uint32_t bp = uni(bitpos);
                LANES(l) if (l == 0) {
                    rows[row].bitpos = bp;
                    for (int c = 0; c < 4; ++c) rows[row].last_dc[c] = (int16_t)lastdc[c];
                    rows[row].aux = 0;
                }
            }
            for (int mx = 0; mx < mcuh && !status; ++mx, ++mcu) {
                for (int ci = 0; ci < ncomp && !status; ++ci) {
                    const int cmp = img->scan_cmp[ci];
                    const int hs = img->hs[cmp], vs = img->vs[cmp], bch = img->bch[cmp];
                    for (int v = 0; v < vs && !status; ++v)
                        for (int h = 0; h < hs && !status; ++h) {
                            int diff = 0;
                            if (!decode_block(img->dc_tbl[cmp], 2 + img->ac_tbl[cmp], &diff)) { status = 1; break; }
                            const int dc = (int16_t)(uni((uint32_t)diff) + lastdc[cmp]);
                            lastdc[cmp] = dc;
                            LSYNC();
                            int16_t* dst = img->blocks[cmp] + (int64_t)((row * vs + v) * bch + mx * hs + h) * 64;
                            LANES(l) { dst[l] = l == 49 ? (int16_t)dc : sh->blk[l]; sh->blk[l] = 0; }
                            LSYNC();
                            if (uni(bitpos) > img->scan_len * 8u) status = 2;   // ran out of data inside a block
                        }
                }
                if (status) break;
                // next_mcupos: end of scan / restart interval
                int sta = 0;
                if (mcu + 1 >= img->mcuc) sta = 2;
                else if (rsti > 0 && --rstw == 0) sta = 1;
                if (sta) {
                    const int got = unpad(padbit == -1 ? 255 : padbit);
                    if (padbit == -1) padbit = (int8_t)got;
                    else if (padbit != got) { padbit = 1; status = 3; }   // "inconsistent use of padbits" (jpgcoder.cc:3255): refused by the host parser
                    if (sta == 1) { rstw = rsti; lastdc[0] = lastdc[1] = lastdc[2] = lastdc[3] = 0; }
                }
            }
        }
        if (!status && uni(bitpos) > img->scan_len * 8u) status = 2;
        const uint32_t bp = uni(bitpos);
        LANES(l) if (l == 0) {
            rows[mcuv].bitpos = bp;
            for (int c = 0; c < 4; ++c) rows[mcuv].last_dc[c] = (int16_t)lastdc[c];
            rows[mcuv].aux = (padbit & 255) | (status << 8);
        }
    }
};

}  // namespace lephuff
