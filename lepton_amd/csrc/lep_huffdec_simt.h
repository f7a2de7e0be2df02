// lep_huffdec_simt.h -- JPEG Huffman scan decode with one LANE per subsequence (round 4).
//
// lep_huffdec.h decodes a scan as uniform code of one wavefront (rounds 2-4 also cut it into <= 64 subsequences of one
// wavefront each, lep_huffdec_par.h, since removed): either way 63 of 64 lanes idle through the serial part, a code takes ~750 wave cycles, and the scan decode of a
// pipeline chunk (896 4K files, 0.29 s) was what the batch compressor had left besides the coder kernels.  Here the unit of
// parallelism is the lane: a scan is cut into subsequences of `sub_bits` bits (a few thousand per 4K file), lane l of a
// wavefront decodes subsequence first_sub + l of its image with a bit reader of its own, and the tables sit in LDS where 64
// lanes look up 64 different codes with one instruction.  What makes that possible is the property the wavefront-per-subsequence form rested
// on -- Huffman-coded JPEG data self-synchronises (Klein & Wiseman 2003; Weissenberger & Schmidt 2018 for the GPU form):
//
//   A  guess   lane i decodes subsequence i from its first bit as if a block started there (speculative: nothing stored, an
//              impossible code costs a bit), and records E_i = (bit position, block-within-MCU) of its first block boundary at
//              or behind the subsequence's end, with the blocks and DC differences it counted;
//   S  settle  lane i >= 1 decodes again from E_(i-1) -- the TRUE state if lane i-1 was in step by its end -- and records its own
//              end state and counts anew.  If no lane of the image saw its end state change, every lane started from a true
//              state (induction from lane 0, whose start is the start of the scan), so every E_i and every count is true.  If some
//              did, the pass is repeated from the new states (kSimtSettle times at most: then the image takes the fallback);
//   P  place   prefix sums over the image's subsequences: every lane's first block (position in the frame) and DC predictors;
//   C  write   lane i decodes its region [E_(i-1), E_i) once more, assembling every block in LDS and storing it whole into the
//              frame, and the hand-off record of every MCU row that starts in it; every irregularity a single-wave decode
//              would meet is met here, from true states, and sets the image's status exactly as there.
//
// Three passes over the bits, 64 codes per instruction.  The last subsequence is not guessed at: its region runs to the end of
// the scan and pass C counts its blocks down.  Files with restart intervals keep the single-wave kernel.
#pragma once
#include "lep_huffdec.h"

namespace lephuff {

constexpr int kSimtSettle = 3;          // settle passes at most
constexpr uint32_t kSimtMinBits = 8192; // a subsequence shorter than this rarely falls into step before it ends

struct SimtSub {            // per (image, subsequence); two copies, read / written alternately by the settle passes
    uint32_t end_bitpos;    // E_i: first block boundary at or behind the end of the subsequence (0xffffffff: the last one, never guessed)
    uint32_t end_phase;
    uint32_t nblocks;       // blocks between the state the lane started from and E_i
    int16_t dcsum[4];       // sums of their DC differences per component (int16 wrap, like the predictor)
};
struct SimtPlace {          // per (image, subsequence), pass P
    uint32_t before;        // blocks of the image in front of the region
    int16_t dc[4];          // DC predictors at its start
};
struct SimtImage {          // per image
    uint32_t first;         // its first entry in the SimtSub / SimtPlace arrays
    uint32_t nsub;
    uint32_t sub_bits;      // multiple of 32
    int32_t status;         // OR of what the passes found (0 = fine)
    int32_t changed[kSimtSettle + 1];   // [k]: settle pass k saw an end state move ([0] is set by pass A: the first settle pass always runs)
};
struct SimtWave { uint32_t image, first_sub; };   // lane l = subsequence first_sub + l of that image

struct SimtShared : HuffDecShared {
    uint8_t ph_cmp[16], ph_v[16], ph_h[16];   // block-within-MCU -> component, row and column inside the MCU
};
// pass C only: the block every lane is assembling, dword w of lane l at [w * 64 + l] (a lane's 16-bit writes then fall into a bank of
// their own).  A block leaves as eight 16-byte stores: written coefficient by coefficient into the frame, every 2-byte store was a
// read-modify-write of a 32-byte sector in HBM -- 167 GB per 896 4K files, the whole 45 ms of the pass (kernel trace, round 4).
struct SimtTile { uint32_t w[32 * 64]; };
typedef uint16_t __attribute__((may_alias)) TileHalf;   // the tile is written in halves and read in dwords: the compiler must know they meet

// bit reader of one lane: 64-bit window, refilled 32 bits at a time from sixteen bytes the lane holds in registers.  (Fetched dword by
// dword, every 64-byte line of the scan came from HBM again for most of its dwords -- 64 lanes x 32 wavefronts per CU walk 64 x 32
// different lines, far more than the vector cache holds: 30 GB per pass over 2.3 GB of scan bits, PMC passes of round 4,
// profiles/r07d_*; sixteen bytes per request are a quarter of the requests.)
struct LaneBits {
    const uint32_t* words;
    uint32_t nwords;        // dwords that hold scan bytes (the buffer is padded with zeros to 16 bytes behind the scan)
    uint64_t w;
    int navail;
    uint32_t bitpos, wi;    // wi: index of the next dword to enter the window
    uint32_t q0, q1, q2, q3;   // dwords (wi & ~3) .. + 3 as loaded
    WDEV void load_quad(uint32_t k) {      // k: multiple of 4; the buffer is 16-byte aligned and zero-padded behind the scan
        if (k < nwords) {
#if LEP_ON_GPU
            typedef uint32_t Quad __attribute__((vector_size(16)));
            const Quad v = *reinterpret_cast<const Quad*>(words + k);      // one global_load_dwordx4
            q0 = v[0]; q1 = v[1]; q2 = v[2]; q3 = v[3];
#else
            uint32_t v[4];
            memcpy(v, words + k, 16);
            q0 = v[0]; q1 = v[1]; q2 = v[2]; q3 = v[3];
#endif
        } else { q0 = q1 = q2 = q3 = 0u; }
    }
    WDEV uint32_t next_word() {            // dword wi (0 behind the scan's last byte), big-endian, and on to the next
        const uint32_t k = wi & 3u;
        if (k == 0) load_quad(wi);
        const uint32_t raw = k == 0 ? q0 : (k == 1 ? q1 : (k == 2 ? q2 : q3));
        const uint32_t v = wi < nwords ? __builtin_bswap32(raw) : 0u;
        ++wi;
        return v;
    }
    WDEV void seek(uint32_t bp) {
        wi = bp >> 5;
        if (wi & 3u) load_quad(wi & ~3u);
        const uint32_t hi = next_word(), lo = next_word();
        w = ((uint64_t)hi << 32) | lo;
        navail = 64; bitpos = bp & ~31u;
        if (bp & 31u) consume(bp & 31u);
    }
    WDEV uint32_t top() const { return (uint32_t)(w >> 32); }
    WDEV void consume(uint32_t n) {   // n <= 31
        w <<= n; navail -= (int)n; bitpos += n;
        if (navail <= 32) { w |= (uint64_t)next_word() << (32 - navail); navail += 32; }
    }
};

struct SimtLane {
    const HuffDecImage* img;
    const SimtShared* sh;
    LaneBits br;

    // Huffman symbol of table t and the magnitude bits behind it (lep_huffdec.h symbol_and_bits, per lane): -1 = not a code
    WDEV int symbol_and_bits(int t, bool dc, uint32_t* n) {
        const uint32_t top = br.top();
        const uint32_t e = dc ? sh->lut_dc[t][top >> 24] : sh->lut_ac[t - 2][top >> 23];
        uint32_t len = e >> 8, sym = e & 255u;
        if (len == 0) {            // 9..16 bits: the shortest length whose first bits do not exceed that length's largest code
            int k = 0;
            while (k < 8 && (int)(top >> (23 - k)) > sh->maxcode[t][k]) ++k;
            if (k == 8) return -1;
            len = 9u + (uint32_t)k;
            sym = sh->longsym[t][(uint32_t)(sh->valoff[t][k] + (int)(top >> (23 - k))) & 255u];
        }
        const uint32_t s = dc ? sym : (sym & 15u);
        if (s > 15) return -1;
        *n = s ? (top << len) >> (32 - s) : 0u;
        br.consume(len + s);
        return (int)sym;
    }
    static WDEV int extend(uint32_t s, uint32_t n) { return s == 0 ? (int)n : (n >= (1u << (s - 1)) ? (int)n : (int)n + 1 - (1 << s)); }

    // one block, nothing stored (speculative form)
    WDEV void skip_block(int dct, int act, int* diff) {
        uint32_t n = 0;
        int hc = symbol_and_bits(dct, true, &n);
        if (hc < 0) { br.consume(1); return; }
        *diff = extend((uint32_t)hc & 255u, n);
        for (uint32_t bpos = 1; bpos < 64;) {
            hc = symbol_and_bits(act, false, &n);
            if (hc < 0) { br.consume(1); return; }
            if (hc == 0) break;
            const uint32_t z = ((uint32_t)hc >> 4) & 15u;
            if (z + bpos >= 64) return;
            bpos += z + 1;
        }
    }
    // one block into the lane's tile (zero so far): lep_huffdec.h decode_block; false = irregular
    WDEV bool store_block(int dct, int act, TileHalf* tile_lane, int* diff) {
        uint32_t n = 0;
        int hc = symbol_and_bits(dct, true, &n);
        if (hc < 0) return false;
        *diff = extend((uint32_t)hc & 255u, n);
        uint32_t last_s = 1;
        for (uint32_t bpos = 1; bpos < 64;) {
            hc = symbol_and_bits(act, false, &n);
            if (hc < 0) return false;
            if (hc == 0) break;
            const uint32_t z = ((uint32_t)hc >> 4) & 15u, s = (uint32_t)hc & 15u;
            if (z + bpos >= 64) return false;
            bpos += z;
            if (s) { const uint32_t a = sh->z2a[bpos]; tile_lane[(a >> 1) * 128 + (a & 1)] = (uint16_t)extend(s, n); }
            ++bpos;
            last_s = s;
        }
        return last_s != 0;      // a coded zero as the block's last coefficient: the reference refuses the file (lep_huffdec.h)
    }
    // BitReader::unpad: the pad-bit pattern of the current partial byte, consuming it
    WDEV int unpad(int fillbit) {
        const int rem = (int)(br.bitpos & 7u);
        if (!rem) return fillbit;
        const int nb = 8 - rem;
        int last = (int)(br.top() >> 31), off = 1;
        br.consume(1);
        int f = last;
        for (int i = 1; i < nb; ++i) { last = (int)(br.top() >> 31); br.consume(1); f |= last << off; ++off; }
        while (off < 7) { f |= last << off; ++off; }
        return f & 255;
    }
};

// tables of the wavefront's image into LDS (every lane of a wavefront belongs to the same image); returns blocks per MCU
WDEV int simt_setup(const HuffDecImage* img, SimtShared* sh) {
    LANES(l) {
        for (int i = l; i < 2 * 512; i += 64) (&sh->lut_ac[0][0])[i] = (&img->lut[2][0])[i];
        for (int i = l; i < 2 * 256; i += 64) {
            const uint16_t e = img->lut[i >> 8][(i & 255) * 2];
            (&sh->lut_dc[0][0])[i] = (e >> 8) <= 8 ? e : (uint16_t)0;
        }
        if (l < 32) { (&sh->maxcode[0][0])[l] = (&img->maxcode[0][0])[l]; (&sh->valoff[0][0])[l] = (&img->valoff[0][0])[l]; }
        for (int i = l; i < 4 * 256; i += 64) (&sh->longsym[0][0])[i] = (&img->longsym[0][0])[i];
        sh->z2a[l] = kZ2A[l];
        if (l == 0) {
            int p = 0;
            for (int ci = 0; ci < img->ncomp; ++ci) {
                const int cmp = img->scan_cmp[ci];
                for (int v = 0; v < img->vs[cmp]; ++v)
                    for (int h = 0; h < img->hs[cmp]; ++h)
                        if (p < 16) { sh->ph_cmp[p] = (uint8_t)cmp; sh->ph_v[p] = (uint8_t)v; sh->ph_h[p] = (uint8_t)h; ++p; }
            }
        }
    }
    LSYNC();
    int nphase = 0;
    for (int ci = 0; ci < img->ncomp; ++ci) nphase += img->hs[img->scan_cmp[ci]] * img->vs[img->scan_cmp[ci]];
    return nphase;
}

// An image whose blocks ALL use the same DC and the same AC table (Cb + Cr coded together in a scan of their own; files written with one
// pair of tables for every component): the lanes fall into step with the block boundaries like any others, but WHICH block of the MCU a
// lane stands on is not in the bits -- a block decodes the same whatever component it is taken for -- so a guessed position inside the
// MCU never corrects itself, and the true one would travel one lane per settle pass.  For such an image (2 .. 4 blocks per MCU) the
// passes do without it: a lane counts its blocks from SLOT 0 of an MCU it does not know and sums the DC differences per slot
// (SimtSub::dcsum by slot, not by component); pass P learns every lane's true slot from the prefix sum of the block counts and only then
// turns slot sums into component sums; pass C starts from that slot.  Returns the blocks per MCU of such an image, else 0.
WDEV int simt_blind_phases(const HuffDecImage* img) {
    int nphase = 0;
    bool same = true;
    const int c0 = img->scan_cmp[0];
    for (int ci = 0; ci < img->ncomp; ++ci) {
        const int cmp = img->scan_cmp[ci];
        nphase += img->hs[cmp] * img->vs[cmp];
        same = same && img->dc_tbl[cmp] == img->dc_tbl[c0] && img->ac_tbl[cmp] == img->ac_tbl[c0];
    }
    return (same && nphase >= 2 && nphase <= 4 && !(img->flags & kHuffDecRstTable)) ? nphase : 0;
}

// passes A (settle = 0) and S (settle = 1..): `in` is read (S), `out` written
WDEV void simt_guess_or_settle(const HuffDecImage* img, SimtShared* sh, SimtImage* si, const SimtSub* in, SimtSub* out, uint32_t first_sub, int settle) {
    if (img->flags & kHuffDecRstTable) return;             // restart intervals: every piece's start is known (simt_write_intervals)
    if (settle > 0 && !si->changed[settle - 1]) {          // the pass before this one moved nothing: its states are the answer
        LANES(l) { const uint32_t i = first_sub + (uint32_t)l; if (i < si->nsub) out[i] = in[i]; }
        return;
    }
    const int nphase = simt_setup(img, sh);
    const bool blind = simt_blind_phases(img) != 0;
    const uint32_t scan_bits = img->scan_len * 8u, L = si->sub_bits, nsub = si->nsub;
    LV(int, moved);
    LANES(l) {
        L(moved) = 0;
        const uint32_t i = first_sub + (uint32_t)l;
        if (i < nsub) {
            SimtSub r;
            r.end_bitpos = 0xffffffffu; r.end_phase = 0; r.nblocks = 0;
            r.dcsum[0] = r.dcsum[1] = r.dcsum[2] = r.dcsum[3] = 0;
            if (settle > 0 && (i == 0 || i + 1 >= nsub)) r = in[i];             // lane 0's guess was no guess; the last region is pass C's
            else if (i + 1 < nsub) {
                uint32_t bp = i * L;
                int phase = 0;
                bool ok = true;
                if (settle > 0) { bp = in[i - 1].end_bitpos; phase = blind ? 0 : (int)in[i - 1].end_phase; ok = bp <= scan_bits && phase < nphase; }
                const uint32_t stop = (i + 1) * L < scan_bits ? (i + 1) * L : scan_bits;
                if (ok) {
                    SimtLane d;
                    d.img = img; d.sh = sh;
                    d.br.words = reinterpret_cast<const uint32_t*>(img->scan); d.br.nwords = (img->scan_len + 3) >> 2;
                    d.br.seek(bp);
                    int sum[4] = {0, 0, 0, 0};
                    uint32_t count = 0;
                    while (d.br.bitpos < stop) {
                        int diff = 0;
                        const int at = sh->ph_cmp[phase];
                        d.skip_block(img->dc_tbl[at], 2 + img->ac_tbl[at], &diff);
                        const int cmp = blind ? phase : at;            // (blind: the sums go by slot, counted from the lane's first block)
                        if (cmp == 0) sum[0] = (int16_t)(sum[0] + diff); else if (cmp == 1) sum[1] = (int16_t)(sum[1] + diff);
                        else if (cmp == 2) sum[2] = (int16_t)(sum[2] + diff); else sum[3] = (int16_t)(sum[3] + diff);
                        ++count;
                        phase = phase + 1 == nphase ? 0 : phase + 1;
                    }
                    r.end_bitpos = d.br.bitpos; r.end_phase = (uint32_t)phase; r.nblocks = count;
                    for (int c = 0; c < 4; ++c) r.dcsum[c] = (int16_t)sum[c];
                }
                if (settle > 0 && (r.end_bitpos != in[i].end_bitpos || r.end_phase != in[i].end_phase)) L(moved) = 1;
            }
            out[i] = r;
        }
    }
    if (lepwave::wave_ballot(moved) || settle == 0) { LANES(l) if (l == 0) si->changed[settle] = 1; }
}

// pass P: one wavefront per image
// (... and, before the write pass runs, marks the image's final record as not written: the record array is reused from call to call,
// and a write pass in which no lane gets to it must not leave an older image's record standing there with status 0 -- ADVICE round 5)
WDEV void simt_place(const HuffDecImage* img, SimtImage* si, const SimtSub* sub, SimtPlace* place, int passes, HuffDecRow* rows_arena = nullptr) {
    if (rows_arena) { LANES(l) if (l == 0) { HuffDecRow* fin = rows_arena + img->rows_off + img->mcuv; fin->bitpos = 0; fin->aux = kHuffDecRowUnwritten; } }
    if (img->flags & kHuffDecRstTable) return;
    int nphase = 0;
    for (int ci = 0; ci < img->ncomp; ++ci) nphase += img->hs[img->scan_cmp[ci]] * img->vs[img->scan_cmp[ci]];
    const uint32_t nsub = si->nsub, total = (uint32_t)img->mcuc * (uint32_t)nphase;
    int bad = si->changed[passes] ? 3 : 0;                 // the last settle pass still moved an end state: not synchronised
    uint32_t run = 0;
    int dc[4] = {0, 0, 0, 0};
    // (blind images, simt_blind_phases: component of every slot of the MCU)
    const int blind = simt_blind_phases(img);
    int slot_cmp[4] = {0, 0, 0, 0};
    if (blind) {
        int p = 0;
        for (int ci = 0; ci < img->ncomp; ++ci)
            for (int k = img->hs[img->scan_cmp[ci]] * img->vs[img->scan_cmp[ci]]; k > 0; --k) { if (p < 4) slot_cmp[p] = img->scan_cmp[ci] & 3; ++p; }
    }
    for (uint32_t base = 0; base < nsub; base += 64) {
        LV(int, nb); LV(int, ex);
        LV(int, d0); LV(int, d1); LV(int, d2); LV(int, d3);
        LV(int, e0); LV(int, e1); LV(int, e2); LV(int, e3);
        LANES(l) {
            const uint32_t i = base + (uint32_t)l;
            const bool in = i < nsub;
            L(nb) = in ? (int)sub[i].nblocks : 0;
            L(d0) = in ? sub[i].dcsum[0] : 0; L(d1) = in ? sub[i].dcsum[1] : 0; L(d2) = in ? sub[i].dcsum[2] : 0; L(d3) = in ? sub[i].dcsum[3] : 0;
        }
        const int tn = lepwave::wave_excl_scan(nb, ex);
        if (blind) {                                       // slot sums -> component sums, now that the lane's first slot is known
            LANES(l) {
                const uint32_t first = (run + (uint32_t)L(ex)) % (uint32_t)blind;
                int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                for (int r = 0; r < blind; ++r) {
                    const int v = r == 0 ? L(d0) : (r == 1 ? L(d1) : (r == 2 ? L(d2) : L(d3)));
                    const uint32_t slot = (first + (uint32_t)r) % (uint32_t)blind;
                    const int cmp = slot == 0 ? slot_cmp[0] : (slot == 1 ? slot_cmp[1] : (slot == 2 ? slot_cmp[2] : slot_cmp[3]));
                    if (cmp == 0) c0 += v; else if (cmp == 1) c1 += v; else if (cmp == 2) c2 += v; else c3 += v;
                }
                L(d0) = (int16_t)c0; L(d1) = (int16_t)c1; L(d2) = (int16_t)c2; L(d3) = (int16_t)c3;
            }
        }
        const int t0 = lepwave::wave_excl_scan(d0, e0), t1 = lepwave::wave_excl_scan(d1, e1), t2 = lepwave::wave_excl_scan(d2, e2), t3 = lepwave::wave_excl_scan(d3, e3);
        LANES(l) {
            const uint32_t i = base + (uint32_t)l;
            if (i < nsub) {
                SimtPlace p;
                p.before = run + (uint32_t)L(ex);
                p.dc[0] = (int16_t)(dc[0] + L(e0)); p.dc[1] = (int16_t)(dc[1] + L(e1)); p.dc[2] = (int16_t)(dc[2] + L(e2)); p.dc[3] = (int16_t)(dc[3] + L(e3));
                place[i] = p;
            }
        }
        run += (uint32_t)tn;
        dc[0] = (int16_t)(dc[0] + t0); dc[1] = (int16_t)(dc[1] + t1); dc[2] = (int16_t)(dc[2] + t2); dc[3] = (int16_t)(dc[3] + t3);
        if (run > total) bad = 3;
    }
    if (bad) { LANES(l) if (l == 0) si->status |= bad; }
}

// Scans with RESTART INTERVALS whose markers all stand where they should (kHuffDecRstTable): a restart marker is a piece boundary that
// needs no guessing -- the bit stream is byte-aligned behind it, the DC predictors start from zero, and the MCU it starts with follows
// from its number (decode_jpeg resets all three at every marker: jpgcoder.cc:2895-2910, 3230-3260).  The host splitter, which sees
// every marker, sends their positions in the un-stuffed scan along behind the scan bytes; lane = interval: `sub` index i covers MCUs
// [i * rsti, (i + 1) * rsti), starts at byte rst[i - 1] and must end -- last MCU, then the pad bits to the byte boundary -- exactly at
// byte rst[i].  The pad-bit patterns of all intervals must agree (the reference notes the first one and takes offence at any other:
// decode_scans); they are collected in si->changed[0..2] (and / or / count), which a restart-table image does not otherwise use.
// Anything else -- a code that is none, an interval that ends elsewhere, differing pad bits -- sets the image's status: the
// single-wave kernel and then the host parser decode the file the reference's way.
WDEV void simt_write_intervals(const HuffDecImage* img, SimtShared* sh, SimtTile* tile, SimtImage* si, HuffDecRow* rows_arena, uint32_t first_sub) {
    const int nphase = simt_setup(img, sh);
    const uint32_t nsub = si->nsub;                         // intervals
    const uint32_t* rst = reinterpret_cast<const uint32_t*>(img->scan + huffdec_scan_room(img->scan_len));
    HuffDecRow* rows = rows_arena + img->rows_off;
    LANES(l) for (int w = 0; w < 32; ++w) tile->w[w * 64 + l] = 0u;
    LSYNC();
    LV(int, rc); LV(int, padv);
    LANES(l) {
        L(rc) = 0; L(padv) = -1;
        TileHalf* tile_lane = reinterpret_cast<TileHalf*>(tile->w + l);
        const uint32_t i = first_sub + (uint32_t)l;
        if (i < nsub) {
            const uint32_t first_byte = i ? rst[i - 1] : 0u, end_byte = i + 1 < nsub ? rst[i] : img->scan_len;
            int bad = 0;
            if (first_byte > end_byte || end_byte > img->scan_len) bad = 3;
            if (!bad) {
                SimtLane d;
                d.img = img; d.sh = sh;
                d.br.words = reinterpret_cast<const uint32_t*>(img->scan); d.br.nwords = (img->scan_len + 3) >> 2;
                d.br.seek(first_byte * 8u);
                const uint32_t end_bits = end_byte * 8u;
                const int mcuh = img->mcuh;
                int mcu = (int)(i * (uint32_t)img->rsti);
                const int mcu_end = mcu + img->rsti < img->mcuc ? mcu + img->rsti : img->mcuc;
                int row = mcu / mcuh, mx = mcu - row * mcuh;
                int lastdc[4] = {0, 0, 0, 0};
                for (; mcu < mcu_end && !bad; ++mcu) {
                    for (int phase = 0; phase < nphase; ++phase) {
                        if (phase == 0 && mx == 0) {       // hand-off record of the MCU row that starts here
                            rows[row].bitpos = d.br.bitpos;
                            for (int c = 0; c < 4; ++c) rows[row].last_dc[c] = (int16_t)lastdc[c];
                            rows[row].aux = 0;
                        }
                        const int cmp = sh->ph_cmp[phase], v = sh->ph_v[phase], h = sh->ph_h[phase];
                        int16_t* dst = img->blocks[cmp] + (int64_t)((row * img->vs[cmp] + v) * img->bch[cmp] + mx * img->hs[cmp] + h) * 64;
                        int diff = 0;
                        const bool fine = d.store_block(img->dc_tbl[cmp], 2 + img->ac_tbl[cmp], tile_lane, &diff);
                        const int cur = cmp == 0 ? lastdc[0] : (cmp == 1 ? lastdc[1] : (cmp == 2 ? lastdc[2] : lastdc[3]));
                        const int dc = (int16_t)(diff + cur);
                        if (cmp == 0) lastdc[0] = dc; else if (cmp == 1) lastdc[1] = dc; else if (cmp == 2) lastdc[2] = dc; else lastdc[3] = dc;
                        tile_lane[(49 >> 1) * 128 + (49 & 1)] = (uint16_t)dc;
                        {
                            typedef uint32_t Quad __attribute__((vector_size(16)));
                            Quad* out = reinterpret_cast<Quad*>(dst);
                            for (int q = 0; q < 8; ++q) {
                                Quad v4;
                                v4[0] = tile->w[(4 * q + 0) * 64 + l]; v4[1] = tile->w[(4 * q + 1) * 64 + l]; v4[2] = tile->w[(4 * q + 2) * 64 + l]; v4[3] = tile->w[(4 * q + 3) * 64 + l];
                                tile->w[(4 * q + 0) * 64 + l] = 0u; tile->w[(4 * q + 1) * 64 + l] = 0u; tile->w[(4 * q + 2) * 64 + l] = 0u; tile->w[(4 * q + 3) * 64 + l] = 0u;
                                out[q] = v4;
                            }
                        }
                        if (!fine) { bad = 1; break; }
                        if (d.br.bitpos > end_bits) { bad = 2; break; }       // ran into the next interval (or out of data) inside a block
                    }
                    if (++mx == mcuh) { mx = 0; ++row; }
                }
                if (!bad) {
                    if (d.br.bitpos & 7u) L(padv) = d.unpad(255) & 255;      // (an interval that ends on a byte boundary says nothing about the pad bits)
                    if (d.br.bitpos != end_bits) bad = 3;                      // bytes left over in front of the marker, or the pad bits ran past it
                    else if (i + 1 == nsub) {                                  // the scan's end: the final record (its pad byte is filled in by the finish pass)
                        rows[img->mcuv].bitpos = d.br.bitpos;
                        for (int c = 0; c < 4; ++c) rows[img->mcuv].last_dc[c] = (int16_t)lastdc[c];
                        rows[img->mcuv].aux = 255;
                    }
                }
            }
            L(rc) = bad;
        }
    }
    // the intervals' pad patterns: all that were seen must be one and the same
    LANES(l) {
        if (L(padv) >= 0) {
#if LEP_ON_GPU
            atomicAnd(&si->changed[0], L(padv)); atomicOr(&si->changed[1], L(padv)); atomicAdd(&si->changed[2], 1);
#else
            si->changed[0] &= L(padv); si->changed[1] |= L(padv); si->changed[2] += 1;
#endif
        }
    }
    LV(int, any);
    int all = 0;
    for (int b = 1; b <= 2; b <<= 1) {
        LANES(l) L(any) = (L(rc) & b) != 0;
        if (lepwave::wave_ballot(any)) all |= b;
    }
    if (all) {
#if LEP_ON_GPU
        if (lep_lane_now() == 0) atomicOr(&si->status, all);
#else
        si->status |= all;
#endif
    }
}
// what the finish pass makes of a restart-table image's pad patterns: the status bit for patterns that differ, the pad byte of the final record
WDEV int simt_intervals_pad(const SimtImage* si, int* status_bits) {
    if (si->changed[2] == 0) return 255;                                    // never determined
    if (si->changed[0] != si->changed[1]) { *status_bits |= 1; return 255; }
    return si->changed[1] & 255;
}

// pass C; the image's status collects what the lanes find
WDEV void simt_write(const HuffDecImage* img, SimtShared* sh, SimtTile* tile, SimtImage* si, const SimtSub* sub, const SimtPlace* place, HuffDecRow* rows_arena, uint32_t first_sub) {
    if (si->status) return;                                // pass P refused the image: the fallback decodes it
    if (img->flags & kHuffDecRstTable) { simt_write_intervals(img, sh, tile, si, rows_arena, first_sub); return; }
    const int nphase = simt_setup(img, sh);
    const bool blind = simt_blind_phases(img) != 0;
    const uint32_t scan_bits = img->scan_len * 8u, nsub = si->nsub, total = (uint32_t)img->mcuc * (uint32_t)nphase;
    // A file that ends inside its scan (no EOI; kHuffDecEarlyEof): the reference decodes block after block until the read that takes
    // the data's last bit (bits behind it read as zeros: the buffer's padding), keeps the block that read it and stops
    // (jpgcoder.cc:3034-3069 with bitops.hh:262-300; jpeg_scan.cc decode_scans).  The lane whose region holds that block stops there
    // and leaves the final record {blocks decoded, last DC, kHuffDecRowTruncated}; regions behind it have nothing to decode.
    const bool cut = (img->flags & kHuffDecEarlyEof) != 0;
    HuffDecRow* rows = rows_arena + img->rows_off;
    LANES(l) for (int w = 0; w < 32; ++w) tile->w[w * 64 + l] = 0u;
    LSYNC();
    LV(int, rc);
    LANES(l) {
        L(rc) = 0;
        TileHalf* tile_lane = reinterpret_cast<TileHalf*>(tile->w + l);        // halfword h of dword w: tile_lane[w * 128 + h]
        const uint32_t i = first_sub + (uint32_t)l;
        if (i < nsub) {
            const bool last = i + 1 >= nsub;
            const uint32_t before = place[i].before;
            int lastdc[4] = {place[i].dc[0], place[i].dc[1], place[i].dc[2], place[i].dc[3]};
            uint32_t bp = 0;
            int phase = 0;
            if (i > 0) { bp = sub[i - 1].end_bitpos; phase = blind ? (int)(before % (uint32_t)nphase) : (int)sub[i - 1].end_phase; }
            int bad = 0;
            uint32_t mine = 0;
            bool stopped = false;                          // this lane met the end of a cut file's data
            const bool behind_the_end = cut && i > 0 && bp >= scan_bits;
            // (the LAST lane behind the end with every block counted: a lane in front decoded the image's last block out of the data's
            // last bits and nobody has written the final record -- the single-wave kernel's case, it knows what the reference does there)
            if (behind_the_end) { if (last && before >= total) bad = 2; }
            else if (before > total) bad = 3;
            else {
                mine = last ? total - before : sub[i].nblocks;
                if (before + mine > total || bp > scan_bits || phase >= nphase || (uint32_t)phase != before % (uint32_t)nphase) bad = 3;
            }
            if (!bad && !behind_the_end) {
                SimtLane d;
                d.img = img; d.sh = sh;
                d.br.words = reinterpret_cast<const uint32_t*>(img->scan); d.br.nwords = (img->scan_len + 3) >> 2;
                d.br.seek(bp);
                const int mcuh = img->mcuh;
                int mcu = (int)(before / (uint32_t)nphase);
                int row = mcu / mcuh, mx = mcu - row * mcuh;
                for (uint32_t k = 0; k < mine; ++k) {
                    if (phase == 0 && mx == 0) {           // hand-off record of the MCU row that starts here
                        rows[row].bitpos = d.br.bitpos;
                        for (int c = 0; c < 4; ++c) rows[row].last_dc[c] = (int16_t)lastdc[c];
                        rows[row].aux = 0;
                    }
                    const int cmp = sh->ph_cmp[phase], v = sh->ph_v[phase], h = sh->ph_h[phase];
                    int16_t* dst = img->blocks[cmp] + (int64_t)((row * img->vs[cmp] + v) * img->bch[cmp] + mx * img->hs[cmp] + h) * 64;
                    int diff = 0;
                    const bool fine = d.store_block(img->dc_tbl[cmp], 2 + img->ac_tbl[cmp], tile_lane, &diff);
                    const int cur = cmp == 0 ? lastdc[0] : (cmp == 1 ? lastdc[1] : (cmp == 2 ? lastdc[2] : lastdc[3]));
                    const int dc = (int16_t)(diff + cur);
                    if (cmp == 0) lastdc[0] = dc; else if (cmp == 1) lastdc[1] = dc; else if (cmp == 2) lastdc[2] = dc; else lastdc[3] = dc;
                    tile_lane[(49 >> 1) * 128 + (49 & 1)] = (uint16_t)dc;
                    {   // the block to the frame, the tile cleared for the next one (an irregular block goes out too: the fallback decodes the file again)
                        typedef uint32_t Quad __attribute__((vector_size(16)));     // (one 16-byte store: gcc and clang both know this form)
                        Quad* out = reinterpret_cast<Quad*>(dst);
                        for (int q = 0; q < 8; ++q) {
                            Quad v;
                            v[0] = tile->w[(4 * q + 0) * 64 + l]; v[1] = tile->w[(4 * q + 1) * 64 + l]; v[2] = tile->w[(4 * q + 2) * 64 + l]; v[3] = tile->w[(4 * q + 3) * 64 + l];
                            tile->w[(4 * q + 0) * 64 + l] = 0u; tile->w[(4 * q + 1) * 64 + l] = 0u; tile->w[(4 * q + 2) * 64 + l] = 0u; tile->w[(4 * q + 3) * 64 + l] = 0u;
                            out[q] = v;
                        }
                    }
                    if (!fine) { bad = 1; break; }             // (also in the block that meets the end: the reference's rules there are the host's)
                    if (d.br.bitpos > scan_bits && !cut) { bad = 2; break; }   // ran out of data inside a block
                    if (++phase == nphase) { phase = 0; ++mcu; if (++mx == mcuh) { mx = 0; ++row; } }
                    if (cut && d.br.bitpos >= scan_bits && before + k + 1 < total) {   // the data's last bit has been read, blocks are still missing
                        rows[img->mcuv].bitpos = before + k + 1;
                        for (int c = 0; c < 4; ++c) rows[img->mcuv].last_dc[c] = (int16_t)lastdc[c];
                        rows[img->mcuv].aux = 255 | kHuffDecRowTruncated;
                        stopped = true;
                        break;
                    }
                }
                if (!bad && !stopped && !last && (d.br.bitpos != sub[i].end_bitpos || (!blind && (uint32_t)phase != sub[i].end_phase))) bad = 3;   // must stand where the next region starts
                if (!bad && !stopped && last) {
                    if (phase != 0 || mcu != img->mcuc) bad = 3;
                    else {
                        const int padbit = (int8_t)d.unpad(255);
                        if (d.br.bitpos > scan_bits) bad = 2;
                        else {
                            rows[img->mcuv].bitpos = d.br.bitpos;
                            for (int c = 0; c < 4; ++c) rows[img->mcuv].last_dc[c] = (int16_t)lastdc[c];
                            rows[img->mcuv].aux = padbit & 255;
                        }
                    }
                }
            }
            L(rc) = bad;
        }
    }
    LV(int, any);
    int all = 0;
    for (int b = 1; b <= 2; b <<= 1) {
        LANES(l) L(any) = (L(rc) & b) != 0;
        if (lepwave::wave_ballot(any)) all |= b;
    }
    if (all) {
#if LEP_ON_GPU
        if (lep_lane_now() == 0) atomicOr(&si->status, all);
#else
        si->status |= all;
#endif
    }
}

// subsequence length for a launch: long enough to synchronise in, short enough that the launch fills the chip
inline uint32_t simt_sub_bits(uint64_t launch_bits, uint64_t target_lanes) {
    uint64_t L = (launch_bits / (target_lanes ? target_lanes : 1) + 31) & ~(uint64_t)31;
    if (L < kSimtMinBits) L = kSimtMinBits;
    if (L > (1u << 20)) L = 1u << 20;
    return (uint32_t)L;
}

}  // namespace lephuff
