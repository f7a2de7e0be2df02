// lep_huffdec_par.h -- JPEG Huffman scan decode with SEVERAL wavefronts per image (experimental, opt-in: LEP_HUFFDEC_PAR=<n>).
//
// lep_huffdec.h decodes an image with one wavefront: a ~0.9 s dependency chain per 4K image that the batch pipeline hides
// behind the previous chunk's arithmetic coder -- but a batch of one chunk (the serving daemon's usual case) has nothing to
// hide it behind.  A sequential scan has no entry points, yet Huffman-coded JPEG data SELF-SYNCHRONISES: a decoder started
// at an arbitrary bit with an arbitrary guess of its position inside the MCU falls into step with the true decoder -- same
// bit position AND same block-within-MCU -- after a few blocks (measured on the bench corpus: 1..49 blocks, <= 2.5 kbit, in
// 24 of 24 random starts; Klein & Wiseman 2003, Weissenberger & Schmidt 2018 for the GPU formulation).  So the scan is cut
// into n subsequences of equal length and decoded in two full passes and a short one, n wavefronts per image each:
//   A  sync    wave i decodes subsequence i from its first bit, speculatively (nothing stored, errors ignored), counting
//              blocks and summing DC differences per component; it LOGS its first kHuffParLog (192) block boundaries (bit position,
//              block phase, running count / sums) and records where its first boundary at or behind the subsequence's end
//              lies: E_i = (bit position, block phase);
//   S  stitch  wave i (i >= 1) starts at E_(i-1) -- the TRUE state if wave i-1 was in step by its end -- and decodes until
//              it stands on a boundary of wave i's log: from there on wave i's pass-A trajectory, and with it E_i, is the true
//              decode.  By induction from wave 0 (whose start is the start of the scan) a found match is a proof; no match
//              within the log sends the image to the single-wave kernel / host parser (status != 0): speculation can cost
//              time, never correctness.  The true content of region i = [E_(i-1), E_i) is then (stitched blocks) + (wave i's
//              totals - its log entry at the match);
//   C  write   prefix sums of the region counts / DC sums give every wave its first block's position in the frame and its DC
//              predictors; it decodes its region again, now storing blocks and the per-MCU-row hand-off records exactly as
//              lep_huffdec.h does.
// Two passes of 1/n of the chain each plus a few dozen blocks of stitching.  Files with restart intervals keep the
// single-wave kernel (a restart resets state at an MCU count that a speculative wave does not know).
#pragma once
#include "lep_huffdec.h"

namespace lephuff {

constexpr int kHuffParLog = 192;  // boundaries a wave logs at its start (synchronisation is usually there within 50 blocks, rarely beyond 100)

struct HuffParState {       // one per (image, subsequence), device memory, zeroed before pass A
    uint32_t end_bitpos;    // A: first block boundary at or behind the end of the subsequence
    uint32_t end_phase;     //    and the block-within-MCU there
    uint32_t nblocks;       // A: blocks this wave decoded (speculative);  S: blocks of the true region [E_(i-1), E_i)
    int32_t status;         // S / C: non-zero = irregular or not synchronised: the image takes the fallback
    int16_t dcsum[4];       // A / S: sum of the DC differences per component, like nblocks (int16 wrap, like the predictor)
    uint32_t nlog;          // A: entries of the log
    uint32_t log_bitpos[kHuffParLog];      // boundary BEFORE block k of this wave's pass-A decode
    uint8_t log_phase[kHuffParLog];
    int16_t log_dcsum[kHuffParLog][4];     // sums over the blocks before that boundary
};

struct HuffParShared : HuffDecShared {
    uint8_t ph_cmp[16], ph_v[16], ph_h[16];   // block-within-MCU -> component, row and column inside the MCU
};

constexpr int kHuffParMaxSub = 64;

struct HuffParWave : HuffDecWave {
    HuffParShared* psh;
    int nphase;
    uint32_t scan_bits;

    WDEV void setup(const HuffDecImage* image, HuffParShared* shared) {
        img = image; sh = shared; psh = shared; status = 0;
        LANES(l) {
            for (int i = l; i < 2 * 512; i += 64) (&sh->lut_ac[0][0])[i] = (&img->lut[2][0])[i];
            for (int i = l; i < 2 * 256; i += 64) {
                const uint16_t e = img->lut[i >> 8][(i & 255) * 2];
                (&sh->lut_dc[0][0])[i] = (e >> 8) <= 8 ? e : (uint16_t)0;
            }
            if (l < 32) { (&sh->maxcode[0][0])[l] = (&img->maxcode[0][0])[l]; (&sh->valoff[0][0])[l] = (&img->valoff[0][0])[l]; }
            for (int i = l; i < 4 * 256; i += 64) (&sh->longsym[0][0])[i] = (&img->longsym[0][0])[i];
            sh->z2a[l] = kZ2A[l];
            sh->blk[l] = 0;
            if (l == 0) {
                int p = 0;
                for (int ci = 0; ci < img->ncomp; ++ci) {
                    const int cmp = img->scan_cmp[ci];
                    for (int v = 0; v < img->vs[cmp]; ++v)
                        for (int h = 0; h < img->hs[cmp]; ++h)
                            if (p < 16) { psh->ph_cmp[p] = (uint8_t)cmp; psh->ph_v[p] = (uint8_t)v; psh->ph_h[p] = (uint8_t)h; ++p; }
                }
            }
        }
        LSYNC();
        nphase = 0;
        for (int ci = 0; ci < img->ncomp; ++ci) nphase += img->hs[img->scan_cmp[ci]] * img->vs[img->scan_cmp[ci]];
        scan_bits = img->scan_len * 8u;
    }
    static WDEV uint32_t chunk_bits_of(uint32_t bits, int nsub) { return ((bits + (uint32_t)nsub - 1u) / (uint32_t)nsub + 31u) & ~31u; }

    // bit reader positioned at an arbitrary bit
    WDEV void seek(uint32_t bp) {
        hi = vec(0); lo = vec(0); navail = 0;
        start_reader(bp >> 5);
        bitpos = vec(bp & ~31u);
        refill(); refill();
        if (bp & 31u) consume(bp & 31u);
    }

    // one block without storing it; the DC difference through *diff.  speculative: an impossible code costs one bit and ends
    // the block (the wave is not in step yet); otherwise false = irregular scan.
    WDEV bool skip_block(int dct, int act, int* diff, bool speculative) {
        uint32_t n = 0;
        int hc = symbol_and_bits(dct, true, &n);
        if (ucond(hc < 0)) { if (speculative) consume(1); return speculative; }
        *diff = devli((uint32_t)hc & 255u, n);
        uint32_t bpos = vec(1);
#pragma nounroll
        while (ucond(bpos < 64)) {
            hc = symbol_and_bits(act, false, &n);
            if (ucond(hc < 0)) { if (speculative) consume(1); return speculative; }
            if (ucond(hc == 0)) break;
            const uint32_t z = ((uint32_t)hc >> 4) & 15u;
            if (ucond(z + bpos >= 64)) return speculative;
            bpos += z + 1;
        }
        return true;
    }

    // ---- pass A -----------------------------------------------------------------------------------------------------------
    WDEV void run_sync(const HuffDecImage* image, HuffParShared* shared, HuffParState* st, int sub, int nsub) {
        setup(image, shared);
        const uint32_t cb = chunk_bits_of(scan_bits, nsub);
        const uint32_t start = (uint32_t)sub * cb, end = (uint32_t)(sub + 1) * cb;
        uint32_t eb = 0xffffffffu, ep = 0, count = 0, nlog = 0;
        int sum[4] = {0, 0, 0, 0};
        if (start < scan_bits && sub + 1 < nsub) {           // the last region ends where the image ends: pass C counts it down
            seek(start);
            int phase = 0;
            const uint32_t stop = end < scan_bits ? end : scan_bits;
            while (uni(bitpos) < stop) {
                if (nlog < (uint32_t)kHuffParLog) {
                    const uint32_t b = uni(bitpos);
                    LANES(l) if (l == 0) {
                        st[sub].log_bitpos[nlog] = b; st[sub].log_phase[nlog] = (uint8_t)phase;
                        for (int c = 0; c < 4; ++c) st[sub].log_dcsum[nlog][c] = (int16_t)sum[c];
                    }
                    ++nlog;
                }
                int diff = 0;
                const int cmp = psh->ph_cmp[phase];
                skip_block(img->dc_tbl[cmp], 2 + img->ac_tbl[cmp], &diff, true);
                sum[cmp] = (int16_t)(sum[cmp] + diff);
                ++count;
                phase = phase + 1 == nphase ? 0 : phase + 1;
            }
            eb = uni(bitpos); ep = (uint32_t)phase;
        }
        LANES(l) if (l == 0) {
            st[sub].end_bitpos = eb; st[sub].end_phase = ep; st[sub].nblocks = count; st[sub].nlog = nlog;
            for (int c = 0; c < 4; ++c) st[sub].dcsum[c] = (int16_t)sum[c];
        }
    }

    // true start of subsequence `sub`: the start of the scan, or where the wave in front said its last block ended
    WDEV void true_start(const HuffParState* st, int sub, uint32_t* bp, int* phase) {
        if (sub == 0) { *bp = 0; *phase = 0; }
        else { *bp = st[sub - 1].end_bitpos; *phase = (int)st[sub - 1].end_phase; }
    }

    // ---- pass S -----------------------------------------------------------------------------------------------------------
    // Region i = [E_(i-1), E_i).  Wave 0's region is its own pass-A decode (its start was the true start).  Wave i >= 1 walks
    // from E_(i-1) until it stands on a logged boundary of wave i; what wave i decoded BEFORE that boundary was out of step and
    // is replaced by what was walked here.  Runs after pass A of the whole image; rewrites nblocks / dcsum of its own entry only.
    WDEV void run_stitch(const HuffDecImage* image, HuffParShared* shared, HuffParState* st, int sub, int nsub) {
        setup(image, shared);
        if (sub == 0 || sub + 1 >= nsub) return;           // wave 0: nothing to replace; the last region is counted down in pass C
        uint32_t bp; int phase;
        true_start(st, sub, &bp, &phase);
        const uint32_t nlog = st[sub].nlog;
        int bad = 0;
        uint32_t walked = 0, k = 0;
        int sum[4] = {0, 0, 0, 0};
        if (bp > scan_bits || st[sub].end_bitpos == 0xffffffffu || phase >= nphase || nlog == 0) bad = 3;
        if (!bad) {
            seek(bp);
            for (;;) {
                const uint32_t b = uni(bitpos);
                while (k < nlog && st[sub].log_bitpos[k] < b) ++k;            // both sequences only move forward
                if (k < nlog && st[sub].log_bitpos[k] == b && st[sub].log_phase[k] == (uint8_t)phase) break;   // in step
                if (k >= nlog || b >= st[sub].end_bitpos) { bad = 3; break; }   // not synchronised inside the log
                int diff = 0;
                const int cmp = psh->ph_cmp[phase];
                if (!skip_block(img->dc_tbl[cmp], 2 + img->ac_tbl[cmp], &diff, false)) { bad = 1; break; }
                sum[cmp] = (int16_t)(sum[cmp] + diff);
                ++walked;
                phase = phase + 1 == nphase ? 0 : phase + 1;
            }
        }
        LANES(l) if (l == 0) {
            if (!bad) {
                st[sub].nblocks = walked + (st[sub].nblocks - k);
                for (int c = 0; c < 4; ++c) st[sub].dcsum[c] = (int16_t)(sum[c] + st[sub].dcsum[c] - st[sub].log_dcsum[k][c]);
            }
            st[sub].status = bad;
        }
    }

    // ---- pass C -----------------------------------------------------------------------------------------------------------
    // returns the status to be OR-ed into the image's (0 = fine)
    WDEV int run_write(const HuffDecImage* image, HuffParShared* shared, const HuffParState* st, HuffDecRow* rows_arena, int sub, int nsub) {
        setup(image, shared);
        HuffDecRow* rows = rows_arena + img->rows_off;
        uint32_t before = 0;
        int lastdc[4] = {0, 0, 0, 0};
        for (int j = 0; j < sub; ++j) {
            if (st[j].status) return st[j].status;
            before += st[j].nblocks;
            for (int c = 0; c < 4; ++c) lastdc[c] = (int16_t)(lastdc[c] + st[j].dcsum[c]);
        }
        const uint32_t total = (uint32_t)img->mcuc * (uint32_t)nphase;
        const bool last = sub + 1 >= nsub;
        if (!last && st[sub].status) return st[sub].status;
        if (before > total) return 3;
        uint32_t mine = last ? total - before : st[sub].nblocks;
        if (before + mine > total) return 3;
        uint32_t bp; int phase;
        true_start(st, sub, &bp, &phase);
        if (bp > scan_bits || phase >= nphase || (uint32_t)phase != before % (uint32_t)nphase) return 3;
        seek(bp);
        int mcu = (int)(before / (uint32_t)nphase);
        const int mcuh = img->mcuh;
        for (uint32_t k = 0; k < mine; ++k) {
            if (phase == 0 && mcu % mcuh == 0) {            // hand-off record of the MCU row that starts here
                const uint32_t b = uni(bitpos);
                const int row = mcu / mcuh;
                LANES(l) if (l == 0) {
                    rows[row].bitpos = b;
                    for (int c = 0; c < 4; ++c) rows[row].last_dc[c] = (int16_t)lastdc[c];
                    rows[row].aux = 0;
                }
            }
            const int cmp = psh->ph_cmp[phase], v = psh->ph_v[phase], h = psh->ph_h[phase];
            int diff = 0;
            if (!decode_block(img->dc_tbl[cmp], 2 + img->ac_tbl[cmp], &diff)) return 1;
            const int dc = (int16_t)(uni((uint32_t)diff) + lastdc[cmp]);
            lastdc[cmp] = dc;
            LSYNC();
            const int row = mcu / mcuh, mx = mcu - row * mcuh;
            int16_t* dst = img->blocks[cmp] + (int64_t)((row * img->vs[cmp] + v) * img->bch[cmp] + mx * img->hs[cmp] + h) * 64;
            LANES(l) { dst[l] = l == 49 ? (int16_t)dc : sh->blk[l]; sh->blk[l] = 0; }
            LSYNC();
            if (uni(bitpos) > scan_bits) return 2;          // ran out of data inside a block
            if (++phase == nphase) { phase = 0; ++mcu; }
        }
        if (!last && (uni(bitpos) != st[sub].end_bitpos || (uint32_t)phase != st[sub].end_phase)) return 3;   // must stand where the next region starts
        if (last) {
            if (phase != 0 || mcu != img->mcuc) return 3;
            const int padbit = (int8_t)unpad(255);
            if (uni(bitpos) > scan_bits) return 2;
            const uint32_t b = uni(bitpos);
            LANES(l) if (l == 0) {
                rows[img->mcuv].bitpos = b;
                for (int c = 0; c < 4; ++c) rows[img->mcuv].last_dc[c] = (int16_t)lastdc[c];
                rows[img->mcuv].aux = padbit & 255;
            }
        }
        return 0;
    }
};

}  // namespace lephuff
