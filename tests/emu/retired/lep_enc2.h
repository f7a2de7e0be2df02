// lep_enc2.h -- wave-cooperative encoder ("v2"): one wavefront codes one thread segment, but a block is
// no longer walked bin by bin on one lane.  Per 8x8 block:
//   P0  coefficient blocks (here / above; left / above-left are kept from the previous block) staged in LDS,
//       coalesced 128-byte loads, next block prefetched into registers;
//   P1  lane = coefficient (49 interior in zig-zag order, 7+7 edge, DC): every lane derives its own context
//       (aavrg / Lakhani / DC prediction from a 2x8-lane integer IDCT), its bit length and its number of bins;
//       ballots + popcounts give "non-zeros left", a wave scan gives each lane its slot in the block's bin list;
//   P2  lanes write their bins (branch index, bit) into the LDS bin list in stream order;
//   P3  probabilities: all bins whose Branch cannot repeat inside a block load / adapt / store their model
//       word in parallel (one HBM round trip per 64 bins instead of one per bin); sign and threshold bins,
//       whose Branch can repeat, are resolved by a short in-wave forwarding loop;
//   P4  lane 0 runs the bool-coder state machine over (bit, probability) pairs read from LDS.
// Syntax / contexts are those of lep_core.h (same reference citations); results are bit-identical.
#pragma once
#include "../../../lepton_amd/csrc/lep_core.h"
#include "../../../lepton_amd/csrc/lep_wave.h"

namespace lepdev {

constexpr int kMaxBins = 1440;   // 6 + 49*22 + 2*(3 + 7*22) + 22
constexpr int kMaxDup = 160;     // 14*10 threshold bins
constexpr uint32_t kResidentFlag = 1u << 30;   // bin whose Branch lives in LDS (sign table); resolved by the serial lane

struct EncShared {
    uint32_t bins[kMaxBins];   // P2: branch index | bit << 31;  after P3: probability | bit << 8
    uint16_t dup[kMaxDup];     // positions (into bins) of bins whose Branch may repeat within the block
    uint32_t inv[512];         // ceil(2^32 / d): exact division for the probability update
    uint32_t sign[96];         // the sign Branches live in LDS for the whole segment (never written back)
    int32_t t[64];             // IDCT intermediate
    int32_t icos_x[64], icos_y[64];
    int16_t here[64], left[64], above[64], aleft[64];   // aligned order
    int16_t pix[64];
    uint16_t q[64];
    uint8_t thr[64];
    uint8_t r2a[64], a2r[64], nzbin[64];
    NSum ns_left, ns_above, ns_here;
};

WDEV uint32_t branch_update_fast(uint32_t w, int obs, const uint32_t* inv) {
    uint32_t f = w & 255, t = (w >> 8) & 255;
    uint32_t mine = obs ? t : f, other = obs ? f : t;
    if (mine == 255) {
        if (other == 1) return (w & 0xffff) | ((obs ? 0u : 255u) << 16);
        f = (1 + f) >> 1; t = (1 + t) >> 1;
        if (obs) t = 129; else f = 129;
    } else {
        if (obs) ++t; else ++f;
    }
    uint32_t p = (uint32_t)(((uint64_t)(f << 8) * inv[f + t]) >> 32);
    return f | (t << 8) | (p << 16);
}

struct EncWave {
    // wave-uniform state
    const ImageDev* img;
    uint32_t* model;
    EncShared* sh;
    int comp, ci;
    // lane-0 coder state (kept in every lane's copy on the GPU; only lane 0's is used)
    BoolCoder<false> bc;
    uint32_t nbins;

    WDEV void init_tables() {
        LANES(l) {
            sh->r2a[l] = kR2A[l]; sh->a2r[l] = kA2R[l]; sh->nzbin[l] = l < 50 ? kNzBin[l] : 9;
            for (int d = l; d < 512; d += 64) sh->inv[d] = d < 2 ? 0u : (uint32_t)((0x100000000ull + d - 1) / d);
            for (int d = l; d < 96; d += 64) sh->sign[d] = kBranchInit;
            if (l < (int)(sizeof(NSum) / 4)) { ((uint32_t*)&sh->ns_left)[l] = 0; ((uint32_t*)&sh->ns_above)[l] = 0; }
        }
        WSYNC();
    }

    WDEV void stage_component(int c) {
        comp = c; ci = c ? 1 : 0;
        LANES(l) {
            sh->q[l] = img->q[c][l]; sh->icos_x[l] = img->icos_x[c][l]; sh->icos_y[l] = img->icos_y[c][l];
            sh->thr[l] = img->min_thresh[c][l];
        }
        WSYNC();
    }

    // one pass of the integer IDCT (idct.cc:35-161), rows then columns, 8 lanes each, DC forced to zero
    WDEV void idct_rows() {
        constexpr int w1 = 2841, w2 = 2676, w3 = 2408, w5 = 1609, w6 = 1108, w7 = 565, r2 = 181;
        constexpr int w1pw7 = w1 + w7, w1mw7 = w1 - w7, w2pw6 = w2 + w6, w2mw6 = w2 - w6, w3pw5 = w3 + w5, w3mw5 = w3 - w5;
        LANES(l) if (l < 8) {
            const int y8 = l * 8;
#define LEP_CQ2(i) ((int32_t)sh->here[sh->r2a[i]] * (int32_t)sh->q[i])
            int32_t x0 = (l == 0 ? 0 : (int32_t)((uint32_t)LEP_CQ2(y8) << 11)) + 128;
            int32_t x1 = (int32_t)((uint32_t)LEP_CQ2(y8 + 4) << 11);
            int32_t x2 = LEP_CQ2(y8 + 6), x3 = LEP_CQ2(y8 + 2), x4 = LEP_CQ2(y8 + 1), x5 = LEP_CQ2(y8 + 7), x6 = LEP_CQ2(y8 + 5),
                    x7 = LEP_CQ2(y8 + 3), x8;
#undef LEP_CQ2
            x8 = w7 * (x4 + x5); x4 = x8 + w1mw7 * x4; x5 = x8 - w1pw7 * x5;
            x8 = w3 * (x6 + x7); x6 = x8 - w3mw5 * x6; x7 = x8 - w3pw5 * x7;
            x8 = x0 + x1; x0 -= x1;
            x1 = w6 * (x3 + x2); x2 = x1 - w2pw6 * x2; x3 = x1 + w2mw6 * x3;
            x1 = x4 + x6; x4 -= x6; x6 = x5 + x7; x5 -= x7;
            x7 = x8 + x3; x8 -= x3; x3 = x0 + x2; x0 -= x2;
            x2 = (r2 * (x4 + x5) + 128) >> 8;
            x4 = (r2 * (x4 - x5) + 128) >> 8;
            int32_t* t = sh->t + y8;
            t[0] = (x7 + x1) >> 8; t[1] = (x3 + x2) >> 8; t[2] = (x0 + x4) >> 8; t[3] = (x8 + x6) >> 8;
            t[4] = (x8 - x6) >> 8; t[5] = (x0 - x4) >> 8; t[6] = (x3 - x2) >> 8; t[7] = (x7 - x1) >> 8;
        }
        WSYNC();
        LANES(l) if (l < 8) {
            const int32_t* t = sh->t + l;
            int32_t y0 = (int32_t)((uint32_t)t[0] << 8) + 8192, y1 = (int32_t)((uint32_t)t[32] << 8);
            int32_t y2 = t[48], y3 = t[16], y4 = t[8], y5 = t[56], y6 = t[40], y7 = t[24], y8;
            y8 = w7 * (y4 + y5) + 4; y4 = (y8 + w1mw7 * y4) >> 3; y5 = (y8 - w1pw7 * y5) >> 3;
            y8 = w3 * (y6 + y7) + 4; y6 = (y8 - w3mw5 * y6) >> 3; y7 = (y8 - w3pw5 * y7) >> 3;
            y8 = y0 + y1; y0 -= y1;
            y1 = w6 * (y3 + y2) + 4; y2 = (y1 - w2pw6 * y2) >> 3; y3 = (y1 + w2mw6 * y3) >> 3;
            y1 = y4 + y6; y4 -= y6; y6 = y5 + y7; y5 -= y7;
            y7 = y8 + y3; y8 -= y3; y3 = y0 + y2; y0 -= y2;
            y2 = (r2 * (y4 + y5) + 128) >> 8;
            y4 = (r2 * (y4 - y5) + 128) >> 8;
            int16_t* o = sh->pix + l;
            o[0] = (int16_t)((y7 + y1) >> 11); o[8] = (int16_t)((y3 + y2) >> 11); o[16] = (int16_t)((y0 + y4) >> 11);
            o[24] = (int16_t)((y8 + y6) >> 11); o[32] = (int16_t)((y8 - y6) >> 11); o[40] = (int16_t)((y0 - y4) >> 11);
            o[48] = (int16_t)((y3 - y2) >> 11); o[56] = (int16_t)((y7 - y1) >> 11);
        }
        WSYNC();
    }

    static WDEV int half16(int d) { return (int16_t)d / 2; }

    // Encodes the block whose coefficients are already staged in sh->here (+ left/above/aleft when present).
    // Returns 0 or an exit code (wave-uniform).
    WDEV int encode_block(bool has_left, bool has_above) {
        EncShared& S = *sh;
        LV(int, nzf); LV(int, tx); LV(int, ty);
        LV(int, cnt); LV(int, ndup); LV(int, off); LV(int, doff); LV(int, bad);
        LV(int, len_); LV(int, val_); LV(int, pos_); LV(int, nexp_); LV(int, coded_); LV(int, thr_); LV(int, isedge_);
        LV(uint32_t, expbase_); LV(uint32_t, signidx_); LV(uint32_t, resbase_); LV(uint32_t, thrbase_);

        LANES(l) {
            const int a = l < 49 ? l : (l == 63 ? 49 : l + 1);
            L(nzf) = S.here[a] != 0;
        }
        const uint64_t m = lepwave::wave_ballot(nzf);
        const uint64_t mask7 = m & ((1ull << 49) - 1);
        const uint32_t maskh = (uint32_t)(m >> 49) & 0x7f, maskv = (uint32_t)(m >> 56) & 0x7f;
        const int nz = lepwave::popc64(mask7), neh = __builtin_popcount(maskh), nev = __builtin_popcount(maskv);
        LANES(l) {
            int ex = 0, ey = 0;
            if (l < 49 && L(nzf)) { int coord = S.a2r[l]; ex = coord & 7; ey = coord >> 3; }
            L(tx) = ex; L(ty) = ey;
        }
        const int eob_x = lepwave::wave_max(tx), eob_y = lepwave::wave_max(ty);

        idct_rows();   // S.pix = IDCT of the block without its DC

        int nzctx = 0;
        if (has_left && has_above) nzctx = (S.ns_above.nz + S.ns_left.nz + 2) / 4;
        else if (has_above) nzctx = (S.ns_above.nz + 1) / 2;
        else if (has_left) nzctx = (S.ns_left.nz + 1) / 2;

        // ---- P1: per-lane analysis -------------------------------------------------------------
        LANES(l) {
            const int a = l < 49 ? l : (l == 63 ? 49 : l + 1);
            int c = S.here[a];
            int v = c < 0 ? -c : c, len = bitlen((uint32_t)v), pos = c >= 0;
            int coded = 0, n = 0, nd = 0, nexp = 0, thr = 0, isedge = 0, err = 0;
            uint32_t expbase = 0, signidx = 0, resbase = 0, thrbase = 0;
            if (l < 49) {
                const int before = lepwave::popc64(mask7 & ((1ull << l) - 1));
                const int left_before = nz - before;
                coded = left_before > 0;
                if (coded) {
                    int prior;
                    if (has_left && has_above) prior = (uint16_t)((iabs(S.left[l]) + iabs(S.above[l])) * 13 + 6 * iabs(S.aleft[l])) >> 5;
                    else if (has_left) prior = (int16_t)iabs(S.left[l]);
                    else if (has_above) prior = (int16_t)iabs(S.above[l]);
                    else prior = 0;
                    const int nb = S.nzbin[left_before];
                    const int bsr = bitlen((uint32_t)imin(iabs(prior), 1023));
                    expbase = kExp7 + ((((uint32_t)ci * 10 + nb) * 49 + l) * 12 + bsr) * 11;
                    signidx = kSign + (uint32_t)ci * 48;
                    resbase = kRes + (((uint32_t)ci * 64 + S.a2r[l]) * 10 + nb) * 10;
                }
            } else if (l < 63) {
                const bool horizontal = l < 56;
                const int j = horizontal ? l - 49 : l - 56;
                const uint32_t mk = horizontal ? maskh : maskv;
                const int ne = horizontal ? neh : nev;
                const int ne_before = ne - __builtin_popcount(mk & ((1u << j) - 1));
                coded = ne_before > 0;
                isedge = 1;
                if (coded) {
                    const int coord = horizontal ? j + 1 : (j + 1) * 8;
                    int32_t prior = 0;
                    const bool nbr_ok = horizontal ? has_above : has_left;
                    if (nbr_ok) {
                        const int16_t* nbr = horizontal ? S.above : S.left;
                        const int32_t* icos = horizontal ? S.icos_x + coord * 8 : S.icos_y + coord;
                        const int step = horizontal ? 8 : 1;
                        if (icos[0] == 0) err = 43;
                        else {
                            uint32_t acc = (uint32_t)(int32_t)nbr[S.r2a[coord]] * (uint32_t)icos[0];
                            for (int i = 1; i < 8; ++i) {
                                int32_t xi = S.here[S.r2a[coord + i * step]], ai = nbr[S.r2a[coord + i * step]];
                                int32_t term = (i & 1) ? xi + ai : xi - ai;
                                acc -= (uint32_t)icos[i] * (uint32_t)term;
                            }
                            prior = (int32_t)acc / icos[0];
                        }
                    }
                    const uint32_t aprior = prior < 0 ? 0u - (uint32_t)prior : (uint32_t)prior;
                    const int bsr = bitlen(aprior > 1023 ? 1023 : aprior);
                    const int16_t p16 = (int16_t)prior;
                    const int sctx = p16 == 0 ? 0 : (p16 > 0 ? 1 : 2);
                    thr = S.thr[coord];
                    expbase = kExpX + ((((uint32_t)ci * 10 + ne_before) * 15 + (horizontal ? j : j + 7)) * 12 + bsr) * 11;
                    signidx = kSign + ((uint32_t)ci * 4 + sctx) * 12 + bsr;
                    resbase = kRes + (((uint32_t)ci * 64 + coord) * 10 + ne_before) * 10;
                    if (len > 1 && len - 2 >= thr) {
                        nd = len - 1 - thr;
                        thrbase = kThresh + ((((uint32_t)ci * 256 + (uint32_t)imin((int)((aprior & 0xffff) >> thr), 255)) * 8) +
                                             (uint32_t)imin(len - thr, 7)) * 128;
                    }
                }
            } else {   // DC
                coded = 1;
                int32_t avgmed = 0, unc = 0, unc2 = 0;
                if (has_left || has_above) {
                    int cntest = 0, sum0 = 0, sum1 = 0, mn = 0, mx = 0;
                    for (int side = 0; side < 2; ++side) {
                        if (side == 0 ? !has_left : !has_above) continue;
                        for (int i = 0; i < 8; ++i, ++cntest) {
                            int e;
                            if (side == 0) e = (int16_t)(S.ns_left.vert[i] - half16(S.pix[i * 8] - S.pix[i * 8 + 1]) - (S.pix[i * 8] + 1024));
                            else e = (int16_t)(S.ns_above.horiz[i] - half16(S.pix[i] - S.pix[i + 8]) - (S.pix[i] + 1024));
                            if (cntest < 8) sum0 += e; else sum1 += e;
                            if (cntest == 0) { mn = mx = e; }
                            if (e < mn) mn = e;
                            if (e > mx) mx = e;
                        }
                    }
                    if (cntest == 8) sum1 = sum0;
                    avgmed = (sum0 + sum1) >> 1;
                    unc = (mx - mn) >> 3;
                    sum0 -= avgmed; sum1 -= avgmed;
                    unc2 = (iabs(sum0) < iabs(sum1) ? sum0 : sum1) >> 3;
                }
                const int pred = (avgmed / (int)S.q[0] + 4) >> 3;
                const int ua = imin(bitlen((uint32_t)iabs(unc) & 0xffff), 11), ub = imin(bitlen((uint32_t)iabs(unc2) & 0xffff), 16);
                int d = c - pred;
                if (d < -1024) d += 2049;
                if (d > 1024) d -= 2049;
                int back = d + pred;
                if (back < -1024) back += 2049;
                if (back > 1024) back -= 2049;
                if (back != c) err = 6;
                v = iabs(d); len = bitlen((uint32_t)v & 0xffff); pos = d >= 0;
                expbase = kExpDc + ((uint32_t)ua * 17 + ub) * 11;
                signidx = kSign + (uint32_t)ci * 48 + (unc2 >= 0 ? (unc2 == 0 ? 3 : 2) : 1);
                resbase = kResDc + (uint32_t)ua * 10;
            }
            if (coded) {
                if (len > 11) err = 6;
                nexp = len < 11 ? len + 1 : 11;
                n = nexp + (len ? 1 : 0) + (len > 1 ? len - 1 : 0);
            } else nd = 0;
            if (l == 0) n += 6;
            if (l == 49 || l == 56) n += 3;
            L(cnt) = n; L(ndup) = nd; L(bad) = err;
            L(len_) = len; L(val_) = v; L(pos_) = pos; L(nexp_) = nexp; L(coded_) = coded; L(thr_) = thr; L(isedge_) = isedge;
            L(expbase_) = expbase; L(signidx_) = signidx; L(resbase_) = resbase; L(thrbase_) = thrbase;
        }
        const uint64_t badmask = lepwave::wave_ballot(bad);
        if (badmask) {   // report what the serial coder would have hit first (lane order = stream order)
            LV(int, bad43);
            LANES(l) L(bad43) = L(bad) == 43;
            const uint64_t m43 = lepwave::wave_ballot(bad43);
            return ((m43 >> __builtin_ctzll(badmask)) & 1) ? 43 : 6;
        }
        const int N = lepwave::wave_excl_scan(cnt, off);
        const int D = lepwave::wave_excl_scan(ndup, doff);

        // ---- P2: bin emission ---------------------------------------------------------------------
        LANES(l) {
            int j = L(off), dj = L(doff);
            if (l == 0) {
                const uint32_t T = kNz7x7 + ((uint32_t)ci * 26 + S.nzbin[nzctx]) * 192;
                int so_far = 0;
                for (int i = 5; i >= 0; --i) { int b = (nz >> i) & 1; S.bins[j++] = (T + i * 32 + so_far) | ((uint32_t)b << 31); so_far = (so_far << 1) | b; }
            }
            if (l == 49 || l == 56) {
                const bool horizontal = l == 49;
                const uint32_t T = (horizontal ? kNz8x1 : kNz1x8) + (((uint32_t)ci * 8 + (horizontal ? eob_x : eob_y)) * 8 + (nz + 3) / 7) * 12;
                const int ne = horizontal ? neh : nev;
                int so_far = 0;
                for (int i = 2; i >= 0; --i) { int b = (ne >> i) & 1; S.bins[j++] = (T + i * 4 + so_far) | ((uint32_t)b << 31); so_far = (so_far << 1) | b; }
            }
            if (L(coded_)) {
                const int len = L(len_), v = L(val_), nexp = L(nexp_);
                for (int i = 0; i < nexp; ++i) S.bins[j++] = (L(expbase_) + i) | ((uint32_t)(len != i) << 31);
                if (len) S.bins[j++] = (L(signidx_) - kSign) | kResidentFlag | ((uint32_t)L(pos_) << 31);
                if (len > 1) {
                    int b = len - 2;
                    if (L(isedge_) && b >= L(thr_)) {
                        int s = 1;
                        for (; b >= L(thr_); --b) {
                            int bit = (v >> b) & 1;
                            S.dup[dj++] = (uint16_t)j;
                            S.bins[j++] = (L(thrbase_) + s) | ((uint32_t)bit << 31);
                            s = imin((s << 1) | bit, 127);
                        }
                    }
                    for (; b >= 0; --b) S.bins[j++] = (L(resbase_) + b) | ((uint32_t)((v >> b) & 1) << 31);
                }
            }
        }
        WSYNC();

        // ---- P3a: bins with a block-unique Branch: parallel load / adapt / store ------------------
        // (two rounds of loads are issued before the first use, so their HBM latencies overlap)
        for (int base = 0; base < N; base += 128) {
            LV(uint32_t, w0); LV(uint32_t, w1);
            LANES(l) {
                const int j0 = base + l, j1 = base + 64 + l;
                uint32_t a = 0, b = 0;
                if (j0 < N) { const uint32_t e = S.bins[j0]; if (!(e & kResidentFlag) && (e & 0x3fffffffu) < kThresh) a = model[e & 0x3fffffffu]; }
                if (j1 < N) { const uint32_t e = S.bins[j1]; if (!(e & kResidentFlag) && (e & 0x3fffffffu) < kThresh) b = model[e & 0x3fffffffu]; }
                L(w0) = a; L(w1) = b;
            }
            LANES(l) {
                for (int h = 0; h < 2; ++h) {
                    const int j = base + h * 64 + l;
                    if (j < N) {
                        const uint32_t e = S.bins[j], idx = e & 0x3fffffffu;
                        if (!(e & kResidentFlag) && idx < kThresh) {
                            const uint32_t w = h ? L(w1) : L(w0);
                            const int bit = (int)(e >> 31);
                            model[idx] = branch_update_fast(w, bit, S.inv);
                            S.bins[j] = (w >> 16) | ((uint32_t)bit << 8);
                        }
                    }
                }
            }
        }
        // ---- P3b: threshold bins (rare; their Branch can repeat inside a block): in-order forwarding ---
        for (int cb = 0; cb < D; cb += 64) {
            LV(uint32_t, didx); LV(uint32_t, dw); LV(uint32_t, dbit); LV(int, djpos); LV(int, dlast);
            const int n = D - cb < 64 ? D - cb : 64;
            LANES(l) {
                uint32_t idx = 0xffffffffu, w = 0, bit = 0;
                int j = -1;
                if (l < n) {
                    j = S.dup[cb + l];
                    const uint32_t e = S.bins[j];
                    idx = e & 0x3fffffffu; bit = e >> 31;
                    w = model[idx];
                }
                L(didx) = idx; L(dw) = w; L(dbit) = bit; L(djpos) = j; L(dlast) = 1;
            }
            for (int r = 0; r < n; ++r) {
                const uint32_t ridx = lepwave::wave_read(didx, r), rw = lepwave::wave_read(dw, r), rbit = lepwave::wave_read(dbit, r);
                const uint32_t nw = branch_update_fast(rw, (int)rbit, S.inv);
                LANES(l) {
                    if (l == r) { S.bins[L(djpos)] = (rw >> 16) | (rbit << 8); L(dw) = nw; }
                    else if (L(didx) == ridx) { if (l > r) L(dw) = nw; else L(dlast) = 0; }
                }
            }
            LANES(l) if (l < n && L(dlast)) model[L(didx)] = L(dw);
            WSYNC();
        }
        WSYNC();

        // ---- P4: bool coder over the resolved (bit, probability) pairs ------------------------------
        LANES(l) if (l == 0) {
            for (int j = 0; j < N; ++j) {
                const uint32_t e = S.bins[j];
                if (e & kResidentFlag) {
                    uint32_t* slot = &S.sign[e & 127];
                    const uint32_t w = *slot;
                    const int bit = (int)(e >> 31);
                    bc.put(bit, w >> 16);
                    *slot = branch_update_fast(w, bit, S.inv);
                } else bc.put((int)(e >> 8) & 1, e & 255);
            }
        }
        nbins += (uint32_t)N;

        // ---- P5: neighbour summary of this block ----------------------------------------------------
        LANES(l) {
            if (l < 16) {
                const int i = l & 7;
                const int dcq = S.here[49] * (int)S.q[0];
                if (l < 8) S.ns_here.horiz[i] = (int16_t)(dcq + S.pix[56 + i] + 1024 + half16(S.pix[56 + i] - S.pix[48 + i]));
                else S.ns_here.vert[i] = (int16_t)(dcq + S.pix[i * 8 + 7] + 1024 + half16(S.pix[i * 8 + 7] - S.pix[i * 8 + 6]));
            }
            if (l == 16) S.ns_here.nz = nz;
        }
        WSYNC();
        return 0;
    }

    // whole segment; ns = this segment's NSum area (zeroed); returns exit code
    WDEV int run(const ImageDev* image, const SegDev& seg, uint32_t* model_words, NSum* ns, EncShared* shared, uint8_t* stream,
                 uint32_t cap) {
        img = image; model = model_words; sh = shared; nbins = 0;
        init_tables();
        bc.init_stream(stream, cap);
        bool top[3] = {true, true, true};
        SegmentCoder<false> sched;   // only its row schedule is used
        sched.img = image;
        for (uint32_t idx = 0;; ++idx) {
            SegmentCoder<false>::RowSpec r = sched.row_spec(idx);
            if (r.done) break;
            if (r.luma_y >= seg.y1 && !seg.is_last) break;
            if (r.skip) continue;
            if (r.luma_y < seg.y0) continue;
            stage_component(r.component);
            const int w = img->width[comp], yb = r.curr_y;
            const int16_t* row = img->blocks[comp] + (int64_t)yb * w * 64;
            const bool has_above = !top[comp];
            const int16_t* arow = has_above ? row - (int64_t)w * 64 : nullptr;
            NSum* nrow = ns + img->ns_offset[comp] + (yb & 1) * w;
            const NSum* narow = ns + img->ns_offset[comp] + ((yb & 1) ^ 1) * w;
            top[comp] = false;
            for (int x = 0; x < w; ++x) {
                // P0: stage blocks. left / above-left come from the previous block's LDS copies.
                LANES(l) {
                    if (x) { sh->left[l] = sh->here[l]; sh->aleft[l] = sh->above[l]; }
                    if (l < (int)(sizeof(NSum) / 4)) {
                        if (x) ((uint32_t*)&sh->ns_left)[l] = ((const uint32_t*)&sh->ns_here)[l];
                        if (has_above) ((uint32_t*)&sh->ns_above)[l] = ((const uint32_t*)&narow[x])[l];
                    }
                }
                WSYNC();
                LANES(l) {
                    sh->here[l] = row[(int64_t)x * 64 + l];
                    if (has_above) sh->above[l] = arow[(int64_t)x * 64 + l];
                }
                WSYNC();
                int rc = encode_block(x > 0, has_above);
                if (rc) return rc;
                LANES(l) if (l < (int)(sizeof(NSum) / 4)) ((uint32_t*)&nrow[x])[l] = ((const uint32_t*)&sh->ns_here)[l];
                if (x + 1 < w && yb * w + x + 1 >= img->coded_blocks[comp]) break;
            }
        }
        return 0;
    }
};

}  // namespace lepdev
