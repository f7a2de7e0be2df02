// lep_dec2.h -- wave-cooperative decoder ("v2").  Decoding is serial by nature (the next context depends on
// the bit just decoded), so the bool-decoder state machine stays on lane 0 -- but the wavefront removes the
// HBM round trip per bin that the single-lane kernel pays:
//   * contexts that do not depend on bits of the current block (aavrg priors of all 49 interior positions,
//     Lakhani priors of all 14 edge positions once the interior is known, DC prediction once the ACs are
//     known) are computed lane-parallel;
//   * the Branch words the serial lane can need next are fetched by all lanes in "prefetch rounds":
//       R1  the whole 6x32 nz-count tree slice (192 words),
//       R2+ for the next 16 interior positions: exponent words 0..7 and residual words 0..7 for the current
//           non-zeros-left bin (one dwordx4 load per lane); a new round starts when that bin changes,
//       R3  both edges at once: the two 3x4 count slices and, for every edge position and every possible
//           "edge non-zeros left" value, exponent words 0..7,
//       R4  the DC exponent slice;
//     every prefetched word is read at most once per block, so the serial lane stores the adapted word
//     straight back to HBM (write-through, no dirty tracking);
//   * the sign and DC-residual tables (216 Branches) live in LDS for the whole segment.
// Syntax / contexts are those of lep_core.h (same reference citations); results are bit-identical.
#pragma once
#include "lep_enc2.h"

namespace lepdev {

struct DecShared {
    uint32_t inv[512];
    uint32_t nzT[192];           // R1
    uint32_t pbuf[16][16];       // R2: [position in window][E0..E7, R0..R7]
    uint32_t eE[2][7][7][8];     // R3: [edge][position][ne-1][E0..E7]
    uint32_t eT[2][12];          // R3: count slices
    uint32_t dcE[12];            // R4
    uint32_t sign[96];           // resident
    uint32_t resdc[120];         // resident
    int32_t t[64];
    int32_t icos_x[64], icos_y[64];
    int32_t eprior[14];
    int32_t ctl[16];             // lane 0 -> wave hand-off: nz, zz, left, eob_x, eob_y, rc, pred, a, b, unc2
    int16_t here[64], left[64], above[64], aleft[64];
    int16_t pix[64];
    uint16_t q[64];
    uint8_t thr[64];
    uint8_t r2a[64], a2r[64], nzbin[64], bsr[64];
    uint8_t ebad[16];            // edge position whose prior needs a division by a zero table entry
    NSum ns_left, ns_above, ns_here;
};

struct DecWave {
    const ImageDev* img;
    uint32_t* model;
    DecShared* sh;
    int comp, ci;
    BoolCoder<true> bc;   // lane 0
    uint32_t nbins;       // lane 0

    WDEV void init_tables() {
        LANES(l) {
            sh->r2a[l] = kR2A[l]; sh->a2r[l] = kA2R[l]; sh->nzbin[l] = l < 50 ? kNzBin[l] : 9;
            for (int d = l; d < 512; d += 64) sh->inv[d] = d < 2 ? 0u : (uint32_t)((0x100000000ull + d - 1) / d);
            for (int d = l; d < 96; d += 64) sh->sign[d] = kBranchInit;
            for (int d = l; d < 120; d += 64) sh->resdc[d] = kBranchInit;
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

    // ---- serial-lane helpers (lane 0 only) ---------------------------------------------------------
    WDEV int dec_global(uint32_t idx) {                 // Branch fetched on demand
        const uint32_t w = model[idx];
        const int bit = bc.get(w >> 16);
        model[idx] = branch_update_fast(w, bit, sh->inv);
        ++nbins;
        return bit;
    }
    WDEV int dec_prefetched(uint32_t w, uint32_t idx) {  // Branch word already in LDS; adapted word goes to HBM
        const int bit = bc.get(w >> 16);
        model[idx] = branch_update_fast(w, bit, sh->inv);
        ++nbins;
        return bit;
    }
    WDEV int dec_resident(uint32_t* slot) {              // Branch lives in LDS
        const uint32_t w = *slot;
        const int bit = bc.get(w >> 16);
        *slot = branch_update_fast(w, bit, sh->inv);
        ++nbins;
        return bit;
    }
    WDEV int dec_exponent(const uint32_t* pre, uint32_t base) {   // unary, words 0..7 prefetched
        int i = 0;
        for (; i < 11; ++i) {
            const int bit = i < 8 ? dec_prefetched(pre[i], base + i) : dec_global(base + i);
            if (!bit) break;
        }
        return i;
    }

    // integer IDCT without DC (same arithmetic as lep_enc2.h), 8 lanes per pass
    WDEV void idct_rows() {
        constexpr int w1 = 2841, w2 = 2676, w3 = 2408, w5 = 1609, w6 = 1108, w7 = 565, r2 = 181;
        constexpr int w1pw7 = w1 + w7, w1mw7 = w1 - w7, w2pw6 = w2 + w6, w2mw6 = w2 - w6, w3pw5 = w3 + w5, w3mw5 = w3 - w5;
        LANES(l) if (l < 8) {
            const int y8 = l * 8;
#define LEP_CQ3(i) ((int32_t)sh->here[sh->r2a[i]] * (int32_t)sh->q[i])
            int32_t x0 = (l == 0 ? 0 : (int32_t)((uint32_t)LEP_CQ3(y8) << 11)) + 128;
            int32_t x1 = (int32_t)((uint32_t)LEP_CQ3(y8 + 4) << 11);
            int32_t x2 = LEP_CQ3(y8 + 6), x3 = LEP_CQ3(y8 + 2), x4 = LEP_CQ3(y8 + 1), x5 = LEP_CQ3(y8 + 7), x6 = LEP_CQ3(y8 + 5),
                    x7 = LEP_CQ3(y8 + 3), x8;
#undef LEP_CQ3
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

    // Decodes one block into sh->here (aligned order). left / above / aleft / ns_* are staged by the caller.
    WDEV int decode_block(bool has_left, bool has_above) {
        DecShared& S = *sh;
        // ---- contexts that do not depend on this block's bits ------------------------------------------
        LANES(l) {
            S.here[l] = 0;
            if (l < 49) {
                int prior;
                if (has_left && has_above) prior = (uint16_t)((iabs(S.left[l]) + iabs(S.above[l])) * 13 + 6 * iabs(S.aleft[l])) >> 5;
                else if (has_left) prior = (int16_t)iabs(S.left[l]);
                else if (has_above) prior = (int16_t)iabs(S.above[l]);
                else prior = 0;
                S.bsr[l] = (uint8_t)bitlen((uint32_t)imin(iabs(prior), 1023));
            }
        }
        int nzctx = 0;
        if (has_left && has_above) nzctx = (S.ns_above.nz + S.ns_left.nz + 2) / 4;
        else if (has_above) nzctx = (S.ns_above.nz + 1) / 2;
        else if (has_left) nzctx = (S.ns_left.nz + 1) / 2;
        // ---- R1: nz-count slice --------------------------------------------------------------------------
        WSYNC();
        const uint32_t nzbase = kNz7x7 + ((uint32_t)ci * 26 + S.nzbin[nzctx]) * 192;
        LANES(l) { S.nzT[l] = model[nzbase + l]; S.nzT[l + 64] = model[nzbase + l + 64]; S.nzT[l + 128] = model[nzbase + l + 128]; }
        WSYNC();
        LANES(l) if (l == 0) {
            int nz = 0, so_far = 0;
            for (int i = 5; i >= 0; --i) {
                const int bit = dec_prefetched(S.nzT[i * 32 + so_far], nzbase + i * 32 + so_far);
                nz |= bit << i;
                so_far = (so_far << 1) | bit;
            }
            S.ctl[0] = nz; S.ctl[1] = 0; S.ctl[2] = nz; S.ctl[3] = 0; S.ctl[4] = 0;
        }
        WSYNC();
        const int nz = S.ctl[0];
        if (nz > 49) return 7;
        // ---- R2+: interior coefficients, 16 positions per prefetch round -------------------------------------
        for (;;) {
            const int zz0 = S.ctl[1], left0 = S.ctl[2];
            if (zz0 >= 49 || left0 <= 0) break;
            const int nb = S.nzbin[left0];
            WSYNC();
            LANES(l) {
                const int p = zz0 + (l >> 2), part = l & 3;
                if (p < 49) {
                    const uint32_t base = part < 2
                        ? kExp7 + ((((uint32_t)ci * 10 + nb) * 49 + p) * 12 + S.bsr[p]) * 11 + (part & 1) * 4
                        : kRes + (((uint32_t)ci * 64 + S.a2r[p]) * 10 + nb) * 10 + (part & 1) * 4;
                    uint32_t* dst = &S.pbuf[l >> 2][part * 4];
                    dst[0] = model[base]; dst[1] = model[base + 1]; dst[2] = model[base + 2]; dst[3] = model[base + 3];
                }
            }
            WSYNC();
            LANES(l) if (l == 0) {
                int zz = zz0, left = left0, eob_x = S.ctl[3], eob_y = S.ctl[4];
                for (; zz < 49 && zz < zz0 + 16 && left > 0 && S.nzbin[left] == nb; ++zz) {
                    const uint32_t* pre = S.pbuf[zz - zz0];
                    const uint32_t ebase = kExp7 + ((((uint32_t)ci * 10 + nb) * 49 + zz) * 12 + S.bsr[zz]) * 11;
                    const int len = dec_exponent(pre, ebase);
                    if (len) {
                        const int pos = dec_resident(&S.sign[ci * 48]);
                        const int coord = S.a2r[zz];
                        --left;
                        if ((coord & 7) > eob_x) eob_x = coord & 7;
                        if ((coord >> 3) > eob_y) eob_y = coord >> 3;
                        int v = 1 << (len - 1);
                        if (len > 1) {
                            const uint32_t rbase = kRes + (((uint32_t)ci * 64 + coord) * 10 + nb) * 10;
                            for (int i = len - 2; i >= 0; --i)
                                v |= (i < 8 ? dec_prefetched(pre[8 + i], rbase + i) : dec_global(rbase + i)) << i;
                        }
                        S.here[zz] = (int16_t)(pos ? v : -v);
                    }
                }
                S.ctl[1] = zz; S.ctl[2] = left; S.ctl[3] = eob_x; S.ctl[4] = eob_y;
            }
            WSYNC();
        }
        const int eob_x = S.ctl[3], eob_y = S.ctl[4];
        // ---- R3: both edges ----------------------------------------------------------------------------------
        LANES(l) if (l < 14) {
            const bool horizontal = l < 7;
            const int j = horizontal ? l : l - 7;
            const int coord = horizontal ? j + 1 : (j + 1) * 8;
            int32_t prior = 0;
            const bool nbr_ok = horizontal ? has_above : has_left;
            if (nbr_ok) {
                const int16_t* nbr = horizontal ? S.above : S.left;
                const int32_t* icos = horizontal ? S.icos_x + coord * 8 : S.icos_y + coord;
                const int step = horizontal ? 8 : 1;
                if (icos[0] != 0) {
                    uint32_t acc = (uint32_t)(int32_t)nbr[S.r2a[coord]] * (uint32_t)icos[0];
                    for (int i = 1; i < 8; ++i) {
                        int32_t xi = S.here[S.r2a[coord + i * step]], ai = nbr[S.r2a[coord + i * step]];
                        int32_t term = (i & 1) ? xi + ai : xi - ai;
                        acc -= (uint32_t)icos[i] * (uint32_t)term;
                    }
                    prior = (int32_t)acc / icos[0];
                    S.ebad[l] = 0;
                } else S.ebad[l] = 1;
            } else S.ebad[l] = 0;
            S.eprior[l] = prior;
        }
        WSYNC();
        const uint32_t Th = kNz8x1 + (((uint32_t)ci * 8 + eob_x) * 8 + (nz + 3) / 7) * 12;
        const uint32_t Tv = kNz1x8 + (((uint32_t)ci * 8 + eob_y) * 8 + (nz + 3) / 7) * 12;
        LANES(l) {
            if (l < 24) S.eT[l / 12][l % 12] = model[(l < 12 ? Th : Tv) + (l % 12)];
            // 2 edges x 7 positions x 7 "non-zeros left" values x 2 halves of 4 words = 196 four-word loads
            for (int k = l; k < 196; k += 64) {
                const int half = k & 1, rest = k >> 1, ne = rest % 7 + 1, pj = rest / 7;   // pj = 0..13
                const int e = pj / 7, j = pj % 7;
                const int32_t prior = S.eprior[pj];
                const uint32_t ap = prior < 0 ? 0u - (uint32_t)prior : (uint32_t)prior;
                const int bsr = bitlen(ap > 1023 ? 1023 : ap);
                const uint32_t base = kExpX + ((((uint32_t)ci * 10 + ne) * 15 + (e ? j + 7 : j)) * 12 + bsr) * 11 + half * 4;
                uint32_t* dst = &S.eE[e][j][ne - 1][half * 4];
                dst[0] = model[base]; dst[1] = model[base + 1]; dst[2] = model[base + 2]; dst[3] = model[base + 3];
            }
        }
        WSYNC();
        LANES(l) if (l == 0) {
            int rc = 0;
            for (int e = 0; e < 2 && !rc; ++e) {
                const bool horizontal = e == 0;
                const uint32_t T = horizontal ? Th : Tv;
                int ne = 0, so_far = 0;
                for (int i = 2; i >= 0; --i) {
                    const int bit = dec_prefetched(S.eT[e][i * 4 + so_far], T + i * 4 + so_far);
                    ne |= bit << i;
                    so_far = (so_far << 1) | bit;
                }
                const int delta = horizontal ? 1 : 8, a_off = horizontal ? 50 : 57;
                int coord = delta;
                for (int j = 0; j < 7 && ne; ++j, coord += delta) {
                    if (S.ebad[e * 7 + j]) { rc = 43; break; }
                    const int32_t prior = S.eprior[e * 7 + j];
                    const uint32_t ap = prior < 0 ? 0u - (uint32_t)prior : (uint32_t)prior;
                    const int bsr = bitlen(ap > 1023 ? 1023 : ap);
                    const uint32_t ebase = kExpX + ((((uint32_t)ci * 10 + ne) * 15 + (e ? j + 7 : j)) * 12 + bsr) * 11;
                    const int len = dec_exponent(S.eE[e][j][ne - 1], ebase);
                    if (len) {
                        const int16_t p16 = (int16_t)prior;
                        const int sctx = p16 == 0 ? 0 : (p16 > 0 ? 1 : 2);
                        const int thr = S.thr[coord];
                        const int pos = dec_resident(&S.sign[(ci * 4 + sctx) * 12 + bsr]);
                        const int ne_before = ne;
                        --ne;
                        int v = 1 << (len - 1);
                        if (len > 1) {
                            int b = len - 2;
                            if (b >= thr) {
                                const uint32_t Tt = kThresh + ((((uint32_t)ci * 256 + (uint32_t)imin((int)((ap & 0xffff) >> thr), 255)) * 8) +
                                                               (uint32_t)imin(len - thr, 7)) * 128;
                                int s = 1;
                                for (; b >= thr; --b) {
                                    const int bit = dec_global(Tt + s);
                                    v |= bit << b;
                                    s = imin((s << 1) | bit, 127);
                                }
                            }
                            const uint32_t rbase = kRes + (((uint32_t)ci * 64 + coord) * 10 + ne_before) * 10;
                            for (; b >= 0; --b) v |= dec_global(rbase + b) << b;
                        }
                        S.here[a_off + j] = (int16_t)(pos ? v : -v);
                    }
                }
            }
            S.ctl[5] = rc;
        }
        WSYNC();
        if (S.ctl[5]) return S.ctl[5];
        // ---- R4: DC --------------------------------------------------------------------------------------------
        idct_rows();
        LANES(l) if (l == 0) {
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
            S.ctl[6] = (avgmed / (int)S.q[0] + 4) >> 3;
            S.ctl[7] = imin(bitlen((uint32_t)iabs(unc) & 0xffff), 11);
            S.ctl[8] = imin(bitlen((uint32_t)iabs(unc2) & 0xffff), 16);
            S.ctl[9] = unc2 >= 0 ? (unc2 == 0 ? 3 : 2) : 1;
        }
        WSYNC();
        const uint32_t dcbase = kExpDc + ((uint32_t)S.ctl[7] * 17 + (uint32_t)S.ctl[8]) * 11;
        LANES(l) if (l < 11) S.dcE[l] = model[dcbase + l];
        WSYNC();
        LANES(l) if (l == 0) {
            const int pred = S.ctl[6], a = S.ctl[7];
            const int len = dec_exponent(S.dcE, dcbase);   // words 8..10 (|d| >= 128) come from HBM again: same values, rare
            int d = 0;
            if (len) {
                const int pos = dec_resident(&S.sign[ci * 48 + S.ctl[9]]);
                int v = 1 << (len - 1);
                for (int i = len - 2; i >= 0; --i) v |= dec_resident(&S.resdc[a * 10 + i]) << i;
                d = (int16_t)(pos ? v : -v);
            }
            int dc = d + pred;
            if (dc < -1024) dc += 2049;
            if (dc > 1024) dc -= 2049;
            S.here[49] = (int16_t)dc;
        }
        WSYNC();
        // ---- neighbour summary ------------------------------------------------------------------------------------
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

    WDEV int run(const ImageDev* image, const SegDev& seg, uint32_t* model_words, NSum* ns, DecShared* shared, const uint8_t* stream,
                 uint32_t len) {
        img = image; model = model_words; sh = shared; nbins = 0;
        init_tables();
        LANES(l) if (l == 0) bc.init_stream(stream, len);
        bool top[3] = {true, true, true};
        SegmentCoder<false> sched;
        sched.img = image;
        for (uint32_t idx = 0;; ++idx) {
            SegmentCoder<false>::RowSpec r = sched.row_spec(idx);
            if (r.done) break;
            if (r.luma_y >= seg.y1 && !seg.is_last) break;
            if (r.skip) continue;
            if (r.luma_y < seg.y0) continue;
            stage_component(r.component);
            const int w = img->width[comp], yb = r.curr_y;
            int16_t* row = img->blocks[comp] + (int64_t)yb * w * 64;
            const bool has_above = !top[comp];
            const int16_t* arow = has_above ? row - (int64_t)w * 64 : nullptr;
            NSum* nrow = ns + img->ns_offset[comp] + (yb & 1) * w;
            const NSum* narow = ns + img->ns_offset[comp] + ((yb & 1) ^ 1) * w;
            top[comp] = false;
            for (int x = 0; x < w; ++x) {
                LANES(l) {
                    if (x) { sh->left[l] = sh->here[l]; sh->aleft[l] = sh->above[l]; }
                    if (l < (int)(sizeof(NSum) / 4)) {
                        if (x) ((uint32_t*)&sh->ns_left)[l] = ((const uint32_t*)&sh->ns_here)[l];
                        if (has_above) ((uint32_t*)&sh->ns_above)[l] = ((const uint32_t*)&narow[x])[l];
                    }
                    if (l == 0) sh->ctl[5] = 0;
                }
                WSYNC();
                LANES(l) if (has_above) sh->above[l] = arow[(int64_t)x * 64 + l];
                WSYNC();
                int rc = decode_block(x > 0, has_above);
                if (rc) return rc;
                LANES(l) {
                    row[(int64_t)x * 64 + l] = sh->here[l];
                    if (l < (int)(sizeof(NSum) / 4)) ((uint32_t*)&nrow[x])[l] = ((const uint32_t*)&sh->ns_here)[l];
                }
                if (x + 1 < w && yb * w + x + 1 >= img->coded_blocks[comp]) break;
            }
        }
        return 0;
    }
};

}  // namespace lepdev
