// lep_dec3.h -- "v3" decoder: one wavefront per thread segment, serial lane reduced to the bool-decoder
// recurrence (see lep_v3.h for the why).  Per 8x8 block the wave runs short ROUNDS; in each round
//   1. every lane prefetches one (or two) 16-byte groups of Branch words the serial lane may need next,
//      publishes them to LDS and keeps them in registers ("owner" of the group);
//   2. lane 0 decodes as far as the prefetched contexts reach, reading probabilities from LDS only;
//   3. the owners adapt their Branches from the decoded values (branch.hh:82-100) and store the group back.
// Rounds: R1 the 6-bit non-zero-count tree; R2 (repeated) the next 16 interior positions under the current
// "non-zeros left" bin (exponent words 0..7, residual words 0..3); R3a horizontal-edge count tree;
// R3b horizontal edge contexts (every reachable "edge non-zeros left" value per position) + vertical count
// tree; R3c vertical edge contexts; R4 the DC exponent row.  Rare bins outside the prefetched set
// (exponent words >= 8, residual words >= 4, the threshold table) are coded by lane 0 straight from HBM.
// Syntax / contexts: src/vp8/decoder/decoder.cc:27-141,167-318; src/vp8/model/model.hh:463-485,852-871,
// 1033-1122,674-832 (the same citations as lep_core.h, whose results this kernel reproduces bit for bit).
#pragma once
#include "../../../lepton_amd/csrc/lep_v3.h"

namespace lep3 {

// Optional per-phase cycle accounting (profiling builds only: -DLEP_PROF): lane 0 adds the shader-clock delta since the
// previous stamp to slot i.  Slots: 0 staging, 1 block prologue, 2+3k prefetch+wait / 3+3k serial / 4+3k update of round
// kind k (0..5), 20 Lakhani, 21 IDCT+DC prediction, 22 publish+store, 24+k number of rounds of kind k.
#if defined(LEP_PROF) && LEP_ON_GPU
#define LEP_STAMP(i) do { if (threadIdx.x == 0) { const uint64_t t_ = __builtin_readcyclecounter(); sh->prof[i] += t_ - prof_last; prof_last = t_; } } while (0)
#define LEP_COUNT(i) do { if (threadIdx.x == 0) sh->prof[i] += 1; } while (0)
#else
#define LEP_STAMP(i) ((void)0)
#define LEP_COUNT(i) ((void)0)
#endif

struct Dec3Shared {
    uint32_t sign[kSignWords];    // resident Branches
    uint32_t resdc[kResDcWords];
    int32_t t[64];                // IDCT intermediate
    int32_t icos_x[64], icos_y[64];
    int32_t eprior[16];
    int32_t ctl[16];              // lane 0 -> wave: 0 nz, 1 zz, 2 left, 5 rc, 6 pred, 7 a, 8 b, 9 dc sign ctx, 10 ne_h, 11 ne_v, 12 dc length
    int16_t here[64], left[64], above[64], aleft[64];   // aligned order
    int16_t pix[64];
    uint16_t q[64];
    uint8_t thr[64], r2a[64], a2r[64], nzbin[64], bsr[64];
    uint8_t ene[16], nzlo[16];
    NSum ns_left, ns_above, ns_here;
#ifdef LEP_PROF
    uint64_t prof[32];
#endif
};

struct Dec3Wave {
    const ImageDev* img;
    uint32_t* model;
    Dec3Shared* sh;
    int comp, ci;
    BoolDec3 bc;      // lane 0
    uint32_t nbins;   // bins decoded, accounted per coefficient: a coefficient of bit length len costs 2*len+1 bins (22 at len 11)
#ifdef LEP_PROF
    uint64_t prof_last;
#endif

    WDEV void init_tables() {
        LANES(l) {
            sh->r2a[l] = kR2A[l]; sh->a2r[l] = kA2R[l]; sh->nzbin[l] = l < 50 ? kNzBin[l] : 9;
            for (int d = l; d < kSignWords; d += 64) sh->sign[d] = kBranchInit;
            for (int d = l; d < kResDcWords; d += 64) sh->resdc[d] = kBranchInit;
            if (l < (int)(sizeof(NSum) / 4)) { ((uint32_t*)&sh->ns_left)[l] = 0; ((uint32_t*)&sh->ns_above)[l] = 0; }
            if (l < 16) {
                sh->ctl[l] = 0;
                // smallest "non-zeros left" that still maps to bin l (kNzBin is monotone): 0,1,2,3,4,6,9,13,21,32
                sh->nzlo[l] = (uint8_t)(l < 5 ? l : l == 5 ? 6 : l == 6 ? 9 : l == 7 ? 13 : l == 8 ? 21 : 32);
            }
        }
        LSYNC();
    }
    WDEV void stage_component(int c) {
        comp = c; ci = c ? 1 : 0;
        LANES(l) {
            sh->q[l] = img->q[c][l]; sh->icos_x[l] = img->icos_x[c][l]; sh->icos_y[l] = img->icos_y[c][l];
            sh->thr[l] = img->min_thresh[c][l];
        }
        LSYNC();
    }

    // ---- lane-0 helpers ---------------------------------------------------------------------------------
    WDEV int dec_global(uint32_t idx) {   // a Branch outside the prefetched set: coded straight from HBM
        const uint32_t w = uload(model + idx);
        const int bit = bc.get(w >> 16);
        model[idx] = bupd_s(w, bit);
        return bit;
    }
    WDEV int dec_p(uint32_t prob) { return bc.get(prob); }
    // unary exponent (encoder.cc:255-264), bins 0..7 from the packed probabilities of the two exponent groups;
    // returns 8 when all eight were ones (the caller continues with dec_unary_tail)
    WDEV int dec_unary8(uint32_t pk03, uint32_t pk47) {
        uint64_t pk = (uint64_t)pk03 | ((uint64_t)pk47 << 32);
        int i = 0;
#pragma nounroll
        for (; i < 8; ++i) {
            if (!dec_p((uint32_t)pk & 255)) break;
            pk >>= 8;
        }
        return i;
    }
    WDEV int dec_unary_tail(uint32_t gbase) {   // bins 8..10 straight from HBM (|v| >= 128: rare)
        int i = 8;
#pragma nounroll
        for (; i < 11; ++i) if (!dec_global(gbase + i)) break;
        return i;
    }
    // residual bits b..0 (b <= 3) of |v| from the packed probabilities of the residual group
    WDEV int dec_residual(uint32_t pk, int b, int v) {
#pragma nounroll
        for (; b >= 0; --b) v |= dec_p((pk >> (b * 8)) & 255) << b;
        return v;
    }
    // a `levels`-level binary tree decoded MSB first; the d-th decoded level has 2^d nodes stored as whole groups owned
    // by lanes base + first(d) .., first = 0,1,2,3,5,9 (1,1,1,2,4,8 groups per level)
    WDEV int dec_tree(int levels, const uint32_t* PK, int base) {
        int n = 0;
#pragma nounroll
        for (int d = 0; d < levels; ++d) {
            const int g = d < 3 ? d : (1 << (d - 2)) + 1;
            const uint32_t pk = lepwave::wave_read(PK, base + g + (n >> 2));
            n = (n << 1) | dec_p((pk >> ((n & 3) * 8)) & 255);
        }
        nbins += (uint32_t)levels;
        return n;
    }

    // integer IDCT without DC (idct.cc:35-161), 8 lanes per pass
    WDEV void idct_rows() {
        constexpr int w1 = 2841, w2 = 2676, w3 = 2408, w5 = 1609, w6 = 1108, w7 = 565, r2 = 181;
        constexpr int w1pw7 = w1 + w7, w1mw7 = w1 - w7, w2pw6 = w2 + w6, w2mw6 = w2 - w6, w3pw5 = w3 + w5, w3mw5 = w3 - w5;
        LANES(l) if (l < 8) {
            const int y8 = l * 8;
#define LEP_CQ4(i) ((int32_t)sh->here[sh->r2a[i]] * (int32_t)sh->q[i])
            int32_t x0 = (l == 0 ? 0 : (int32_t)((uint32_t)LEP_CQ4(y8) << 11)) + 128;
            int32_t x1 = (int32_t)((uint32_t)LEP_CQ4(y8 + 4) << 11);
            int32_t x2 = LEP_CQ4(y8 + 6), x3 = LEP_CQ4(y8 + 2), x4 = LEP_CQ4(y8 + 1), x5 = LEP_CQ4(y8 + 7), x6 = LEP_CQ4(y8 + 5),
                    x7 = LEP_CQ4(y8 + 3), x8;
#undef LEP_CQ4
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
        LSYNC();
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
        LSYNC();
    }
    static WDEV int half16(int d) { return (int16_t)d / 2; }

    // ---- owner-side adaptation: (used, bits) masks over the 4 words of a group -------------------------------
    // unary-exponent group holding words i0..i0+3 of a coefficient of bit length len (bins 0..min(len,10), bit = len != i)
    static WDEV void mask_exp(int i0, int len, int& used, int& bits) {
        int n = imin(len, 10) - i0 + 1;
        n = n < 0 ? 0 : (n > 4 ? 4 : n);
        used = (1 << n) - 1;
        const int z = len - i0;
        bits = (z >= 0 && z < 4) ? (used & ~(1 << z)) : used;
    }
    // residual group (words 0..3 = bits 0..3 of |v|), bits 0..top coded through it
    static WDEV void mask_res(int top, int v, int& used, int& bits) {
        top = top > 3 ? 3 : top;
        used = top < 0 ? 0 : (1 << (top + 1)) - 1;
        bits = v & used;
    }
    // tree group: level i (bit i of value), nodes 4k..4k+3 of that level
    static WDEV void mask_tree(int i, int k, int value, int& used, int& bits) {
        const int prefix = value >> (i + 1);
        used = (prefix >> 2) == k ? 1 << (prefix & 3) : 0;
        bits = ((value >> i) & 1) ? used : 0;
    }
    static WDEV void apply4(U4& W, int used, int bits) {
        if (used & 1) W.x = bupd(W.x, bits & 1);
        if (used & 2) W.y = bupd(W.y, (bits >> 1) & 1);
        if (used & 4) W.z = bupd(W.z, (bits >> 2) & 1);
        if (used & 8) W.w = bupd(W.w, (bits >> 3) & 1);
    }

    enum { K_NZ = 0, K_77 = 1, K_TREEH = 2, K_EDGEH = 3, K_EDGEV = 4, K_DC = 5, K_DONE = 6 };
    // Group ownership per round kind (lane -> group); PK0 / PK1 = packed probabilities of the lane's W0 / W1:
    //   K_NZ     lanes 0..16  the 17 groups of the 6-level count tree
    //   K_77     lane pi: exponent words 0..3 of window position pi; 16+pi: residual words 0..3; 32+pi: exponent words 4..7
    //   K_TREEH  lanes 0..2   the 3 levels of the horizontal count tree        K_DC  lanes 0..2  exponent words 0..11
    //   K_EDGE*  combo c = j*4 + k (position j, k-th reachable "non-zeros left" value): lane c: W0 = exponent words 0..3,
    //            W1 = words 4..7; lane 28+c: residual words 0..3; lanes 56..58 (K_EDGEH only): the vertical count tree

    // Decodes one block into sh->here (aligned order). left / above / aleft / ns_* are staged by the caller.
    WDEV int decode_block(bool has_left, bool has_above) {
        Dec3Shared& S = *sh;
        // ---- contexts that do not depend on this block's bits ---------------------------------------------
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
        const int nzbin_ctx = S.nzbin[nzctx];
        LSYNC();
        LEP_STAMP(1);

        int kind = K_NZ, eob_x = 0, eob_y = 0;
#pragma nounroll
        while (kind != K_DONE) {
            // round parameters (wave-uniform, from ctl)
            const int nz = (int)uni((uint32_t)S.ctl[0]), zz0 = (int)uni((uint32_t)S.ctl[1]), left0 = (int)uni((uint32_t)S.ctl[2]);
            const int nb = (int)uni(S.nzbin[left0 > 0 ? left0 : 0]);
            const int e = kind == K_EDGEV ? 1 : 0;
            const int ne = (int)uni((uint32_t)(kind == K_EDGEV ? S.ctl[11] : S.ctl[10]));
            const bool horizontal = e == 0;
            // ---- (a) owners prefetch their groups ------------------------------------------------------------
            LV(U4, W0); LV(U4, W1); LV(uint32_t, a0); LV(uint32_t, a1); LV(int, ok); LV(uint32_t, PK0); LV(uint32_t, PK1);
            LV(uint32_t, INFO);   // K_EDGE*, lanes j*4 (one per position): sign slot | threshold | bad-prior flag | prior bits
            LANES(l) {
                int valid = 0;
                uint32_t adr0 = 0, adr1 = 0, info = 0;
                if (kind == K_NZ) {
                    if (l < 17) {
                        int i, k;
                        if (l < 3) { i = 5 - l; k = 0; } else if (l < 5) { i = 2; k = l - 3; } else if (l < 9) { i = 1; k = l - 5; } else { i = 0; k = l - 9; }
                        valid = 1;
                        adr0 = ctx_nz7(ci, nzbin_ctx) + (uint32_t)i * 32 + (uint32_t)k * 4;
                    }
                } else if (kind == K_77) {
                    const int pi = l & 15, part = l >> 4, p = zz0 + pi;
                    if (l < 48 && p < 49) {
                        valid = 1;
                        if (part == 1) adr0 = ctx_res(ci, S.a2r[p], nb);
                        else adr0 = ctx_exp7(ci, nb, p, S.bsr[p]) + (part == 2 ? 4u : 0u);
                    }
                } else if (kind == K_TREEH) {
                    if (l < 3) { valid = 1; adr0 = ctx_nzedge(true, ci, eob_x, (nz + 3) / 7) + (uint32_t)(2 - l) * 4; }
                } else if (kind == K_DC) {
                    if (l < 3) { valid = 1; adr0 = ctx_expdc(S.ctl[7], S.ctl[8]) + (uint32_t)l * 4; }
                } else {   // K_EDGEH / K_EDGEV
                    const int c = l < 28 ? l : l - 28, j = (c >> 2) & 7, k = c & 3;
                    const int nep = imin(ne, 7 - j) - k;
                    const bool reachable = ne > 0 && nep >= 1 && nep >= ne - j;
                    const int coord = horizontal ? j + 1 : (j + 1) * 8;
                    if (l < 28) {
                        const int32_t prior = S.eprior[e * 7 + j];
                        const uint32_t ap = prior < 0 ? 0u - (uint32_t)prior : (uint32_t)prior;
                        const int bsr = bitlen(ap > 1023 ? 1023 : ap);
                        if (reachable) {
                            valid = 3;
                            adr0 = ctx_expx(ci, nep, horizontal ? j : j + 7, bsr);
                            adr1 = adr0 + 4;
                        }
                        const int16_t p16 = (int16_t)prior;
                        const int thr = S.thr[coord];
                        const uint32_t tctx = (uint32_t)imin((int)((ap & 0xffff) >> thr), 255);
                        info = (uint32_t)((ci * 4 + (p16 == 0 ? 0 : (p16 > 0 ? 1 : 2))) * 12 + bsr) | ((uint32_t)thr << 8) | (tctx << 16) |
                               ((uint32_t)bsr << 24) | ((S.eprior[14] >> (e * 7 + j)) & 1 ? 0x80000000u : 0u);
                    } else if (l < 56) {
                        if (reachable) { valid = 1; adr0 = ctx_res(ci, coord, nep); }
                    } else if (l < 59 && kind == K_EDGEH) {
                        valid = 1;
                        adr0 = ctx_nzedge(false, ci, eob_y, (nz + 3) / 7) + (uint32_t)(58 - l) * 4;
                    }
                    if (l < 7) S.ene[e * 7 + l] = 0;
                }
                uint32_t pk0 = 0, pk1 = 0;
                if (valid & 1) { L(W0) = ld4(model + adr0); pk0 = pack_probs(L(W0)); }
                if (valid & 2) { L(W1) = ld4(model + adr1); pk1 = pack_probs(L(W1)); }
                L(a0) = adr0; L(a1) = adr1; L(ok) = valid; L(PK0) = pk0; L(PK1) = pk1; L(INFO) = info;
            }
            LSYNC();
            LEP_STAMP(2 + 3 * kind); LEP_COUNT(24 + kind);
            // ---- (b) the serial lane ---------------------------------------------------------------------------------
            {
                if (kind == K_NZ) {
                    const int n = dec_tree(6, PK0, 0);
                    S.ctl[0] = n; S.ctl[1] = 0; S.ctl[2] = n;
                } else if (kind == K_77) {
                    int zz = zz0, left = left0;
                    const int left_lo = (int)uni(S.nzlo[nb]);
                    uint32_t sgw = uni(S.sign[ci * 48]);
                    const int zz_end = zz0 + 16 < 49 ? zz0 + 16 : 49, left_stop = left_lo > 1 ? left_lo : 1;
#pragma nounroll
                    for (; zz < zz_end && left >= left_stop; ++zz) {
                        const int pi = zz - zz0;
                        int len = dec_unary8(lepwave::wave_read(PK0, pi), lepwave::wave_read(PK0, 32 + pi));
                        ++nbins;
                        if (len) {
                            if (len == 8) len = dec_unary_tail(ctx_exp7(ci, nb, zz, (int)uni(S.bsr[zz])));
                            nbins += (uint32_t)(2 * len - (len == 11));
                            const int pos = dec_p(sgw >> 16);
                            sgw = bupd_s(sgw, pos);
                            --left;
                            int v = 1 << (len - 1);
                            if (len > 1) {
                                int b = len - 2;
                                if (b >= 4) {
                                    const uint32_t rbase = ctx_res(ci, (int)uni(S.a2r[zz]), nb);
#pragma nounroll
                                    for (; b >= 4; --b) v |= dec_global(rbase + b) << b;
                                }
                                v = dec_residual(lepwave::wave_read(PK0, 16 + pi), b, v);
                            }
                            S.here[zz] = (int16_t)(pos ? v : -v);
                        }
                    }
                    S.sign[ci * 48] = sgw;
                    S.ctl[1] = zz; S.ctl[2] = left;
                } else if (kind == K_TREEH) {
                    S.ctl[10] = dec_tree(3, PK0, 0);
                } else if (kind == K_DC) {
                    const int pred = (int)uni((uint32_t)S.ctl[6]), a = (int)uni((uint32_t)S.ctl[7]);
                    const int sslot = ci * 48 + (int)uni((uint32_t)S.ctl[9]);
                    const uint32_t sgw = uni(S.sign[sslot]);
                    int len = 0;
#pragma nounroll
                    for (; len < 11; ++len)
                        if (!dec_p((lepwave::wave_read(PK0, len >> 2) >> ((len & 3) * 8)) & 255)) break;
                    int d = 0;
                    nbins += (uint32_t)(len ? 2 * len + 1 - (len == 11) : 1);
                    if (len) {
                        const int pos = dec_p(sgw >> 16);
                        S.sign[sslot] = bupd_s(sgw, pos);
                        int v = 1 << (len - 1);
#pragma nounroll
                        for (int i = len - 2; i >= 0; --i) {
                            const uint32_t w = uni(S.resdc[a * 12 + i]);
                            const int bit = dec_p(w >> 16);
                            S.resdc[a * 12 + i] = bupd_s(w, bit);
                            v |= bit << i;
                        }
                        d = (int16_t)(pos ? v : -v);
                    }
                    int dc = d + pred;
                    if (dc < -1024) dc += 2049;
                    if (dc > 1024) dc -= 2049;
                    S.here[49] = (int16_t)dc;
                    S.ctl[12] = len;
                }
            }
            if (kind == K_EDGEH || kind == K_EDGEV) {
#ifdef LEP_DEC3_VECTOR_EDGES
                // Option (off: measured 25 % slower at 4 waves/SIMD): the two edge rounds on lane 0 of the VECTOR unit instead of the scalar unit: with many waves per
                // CU the single scalar ALU is the busiest pipe (measured: 5.5k scalar vs 3k vector instructions per
                // block), so ~40 % of the serial work is moved across.  Same code, per-lane state, broadcast afterwards.
                LANES(l) if (l == 0) {
                    bc.value = ((uint64_t)vec((uint32_t)(bc.value >> 32)) << 32) | vec((uint32_t)bc.value);
                    bc.count = (int)vec((uint32_t)bc.count); bc.range = vec(bc.range);
                    int rc = 0, left = (int)vec((uint32_t)ne);
#else
                {
                    int rc = 0, left = ne;
#endif
                    const int a_off = horizontal ? 50 : 57;
#pragma nounroll
                    for (int j = 0; j < 7 && left; ++j) {
                        const uint32_t info = lepwave::wave_read(INFO, j * 4);
                        if (info >> 31) { rc = 43; break; }
                        const int combo = j * 4 + (imin(ne, 7 - j) - left);
                        const int sslot = (int)(info & 255);
                        const uint32_t sgw = uni(S.sign[sslot]);
                        S.ene[e * 7 + j] = (uint8_t)left;
                        int len = dec_unary8(lepwave::wave_read(PK0, combo), lepwave::wave_read(PK1, combo));
                        ++nbins;
                        if (len) {
                            const int coord = horizontal ? j + 1 : (j + 1) * 8;
                            if (len == 8) len = dec_unary_tail(ctx_expx(ci, left, horizontal ? j : j + 7, (int)((info >> 24) & 15)));
                            nbins += (uint32_t)(2 * len - (len == 11));
                            const int pos = dec_p(sgw >> 16);
                            S.sign[sslot] = bupd_s(sgw, pos);
                            int v = 1 << (len - 1);
                            if (len > 1) {
                                int b = len - 2;
                                const int thr = (int)((info >> 8) & 15);
                                if (b >= thr) {
                                    const uint32_t Tt = ctx_thresh(ci, (int)((info >> 16) & 255), imin(len - thr, 7));
                                    int s = 1;
#pragma nounroll
                                    for (; b >= thr; --b) {
                                        const int bit = dec_global(Tt + s);
                                        v |= bit << b;
                                        s = imin((s << 1) | bit, 127);
                                    }
                                }
                                if (b >= 4) {
                                    const uint32_t rbase = ctx_res(ci, coord, left);
#pragma nounroll
                                    for (; b >= 4; --b) v |= dec_global(rbase + b) << b;
                                }
                                v = dec_residual(lepwave::wave_read(PK0, 28 + combo), b, v);
                            }
                            --left;
                            S.here[a_off + j] = (int16_t)(pos ? v : -v);
                        }
                    }
                    if (!rc && kind == K_EDGEH) S.ctl[11] = dec_tree(3, PK0, 56);
                    S.ctl[5] = rc;
                }
#ifdef LEP_DEC3_VECTOR_EDGES
                bc.value = ((uint64_t)uni((uint32_t)(bc.value >> 32)) << 32) | uni((uint32_t)bc.value);
                bc.count = (int)uni((uint32_t)bc.count); bc.range = uni(bc.range); bc.wi = uni(bc.wi); bc.raw = uni(bc.raw);
                nbins = uni(nbins);
#endif
            }
            LSYNC();
            LEP_STAMP(3 + 3 * kind);
            // ---- (c) owners adapt and store their groups ---------------------------------------------------------------
            LANES(l) if (L(ok)) {
                int u0 = 0, b0 = 0, u1 = 0, b1 = 0;
                if (kind == K_NZ) {
                    int i, k;
                    if (l < 3) { i = 5 - l; k = 0; } else if (l < 5) { i = 2; k = l - 3; } else if (l < 9) { i = 1; k = l - 5; } else { i = 0; k = l - 9; }
                    mask_tree(i, k, S.ctl[0], u0, b0);
                } else if (kind == K_77) {
                    const int pi = l & 15, part = l >> 4, p = zz0 + pi;
                    if (p < S.ctl[1]) {
                        const int cf = S.here[p];
                        const int v = cf < 0 ? -cf : cf, len = bitlen((uint32_t)v);
                        if (part == 1) mask_res(len - 2, v, u0, b0); else mask_exp(part == 2 ? 4 : 0, len, u0, b0);
                    }
                } else if (kind == K_TREEH) {
                    mask_tree(2 - l, 0, S.ctl[10], u0, b0);
                } else if (kind == K_DC) {
                    mask_exp(l * 4, S.ctl[12], u0, b0);
                } else if (l < 56) {
                    const int c = l < 28 ? l : l - 28, j = c >> 2, k = c & 3;
                    if ((int)S.ene[e * 7 + j] == imin(ne, 7 - j) - k) {
                        const int cf = S.here[(horizontal ? 50 : 57) + j];
                        const int v = cf < 0 ? -cf : cf, len = bitlen((uint32_t)v);
                        if (l < 28) { mask_exp(0, len, u0, b0); mask_exp(4, len, u1, b1); }
                        else mask_res(imin(len - 2, (int)S.thr[horizontal ? j + 1 : (j + 1) * 8] - 1), v, u0, b0);
                    }
                } else {
                    mask_tree(58 - l, 0, S.ctl[11], u0, b0);
                }
                if (u0) { apply4(L(W0), u0, b0); st4(model + L(a0), L(W0)); }
                if (u1) { apply4(L(W1), u1, b1); st4(model + L(a1), L(W1)); }
            }
            LSYNC();
            LEP_STAMP(4 + 3 * kind);
            // ---- (d) next round (wave-uniform) ------------------------------------------------------------------------------
            if (kind == K_NZ) {
                if (S.ctl[0] > 49) return 7;
                kind = S.ctl[0] > 0 ? K_77 : K_TREEH;
            } else if (kind == K_77) {
                if (S.ctl[1] >= 49 || S.ctl[2] <= 0) kind = K_TREEH;
            } else if (kind == K_TREEH) {
                kind = K_EDGEH;
            } else if (kind == K_EDGEH) {
                if (S.ctl[5]) return S.ctl[5];
                kind = S.ctl[11] ? K_EDGEV : K_DC;
            } else if (kind == K_EDGEV) {
                if (S.ctl[5]) return S.ctl[5];
                kind = K_DC;
            } else kind = K_DONE;

            if (kind == K_TREEH) {
                // the interior is complete: eob_x / eob_y (encoder.cc:246-250) and the Lakhani priors (model.hh:928-1071) of
                // all 14 edge positions, lane-parallel
                LV(int, tx); LV(int, ty); LV(int, badf);
                LANES(l) {
                    int ex = 0, ey = 0, bad = 0;
                    if (l < 49 && S.here[l] != 0) { const int coord = S.a2r[l]; ex = coord & 7; ey = coord >> 3; }
                    if (l < 14) {
                        const bool hz = l < 7;
                        const int j = hz ? l : l - 7;
                        const int coord = hz ? j + 1 : (j + 1) * 8;
                        int32_t prior = 0;
                        if (hz ? has_above : has_left) {
                            const int16_t* nbr = hz ? S.above : S.left;
                            const int32_t* icos = hz ? S.icos_x + coord * 8 : S.icos_y + coord;
                            const int step = hz ? 8 : 1;
                            if (icos[0] != 0) {
                                uint32_t acc = (uint32_t)(int32_t)nbr[S.r2a[coord]] * (uint32_t)icos[0];
                                for (int i = 1; i < 8; ++i) {
                                    int32_t xi = S.here[S.r2a[coord + i * step]], ai = nbr[S.r2a[coord + i * step]];
                                    int32_t term = (i & 1) ? xi + ai : xi - ai;
                                    acc -= (uint32_t)icos[i] * (uint32_t)term;
                                }
                                prior = (int32_t)acc / icos[0];
                            } else bad = 1;
                        }
                        S.eprior[l] = prior;
                    }
                    L(tx) = ex; L(ty) = ey; L(badf) = bad;
                }
                eob_x = lepwave::wave_max(tx); eob_y = lepwave::wave_max(ty);
                const uint64_t badmask = lepwave::wave_ballot(badf);
                S.eprior[14] = (int32_t)(uint32_t)badmask;   // bit p: position p's prior needs a division by zero
                LSYNC();
                LEP_STAMP(20);
            } else if (kind == K_DC) {
                // DC prediction (model.hh:674-832): IDCT of the ACs, 16 edge estimates on 16 lanes
                idct_rows();
                LV(int, emin); LV(int, emax); LV(int, s0); LV(int, s1); LV(int, tmp);
                LANES(l) {
                    int ev = 0, have = 0;
                    if (l < 8 && has_left) { have = 1; ev = (int16_t)(S.ns_left.vert[l] - half16(S.pix[l * 8] - S.pix[l * 8 + 1]) - (S.pix[l * 8] + 1024)); }
                    if (l >= 8 && l < 16 && has_above) { const int i = l - 8; have = 1; ev = (int16_t)(S.ns_above.horiz[i] - half16(S.pix[i] - S.pix[i + 8]) - (S.pix[i] + 1024)); }
                    L(emax) = have ? ev : -0x7fffffff;
                    L(emin) = have ? -ev : -0x7fffffff;
                    L(s0) = l < 8 ? ev : 0;
                    L(s1) = (l >= 8 && l < 16) ? ev : 0;
                }
                const int mx = lepwave::wave_max(emax), mn = -lepwave::wave_max(emin);
                const int sumL = lepwave::wave_excl_scan(s0, tmp), sumA = lepwave::wave_excl_scan(s1, tmp);
                {
                    int32_t avgmed = 0, unc = 0, unc2 = 0;
                    if (has_left || has_above) {
                        int sum0 = has_left ? sumL : sumA, sum1 = (has_left && has_above) ? sumA : sum0;
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
                LSYNC();
                LEP_STAMP(21);
            }
        }
        // ---- neighbour summary (block_context.hh:44-78) -------------------------------------------------------------
        const int nzf = S.ctl[0];
        LANES(l) {
            if (l < 16) {
                const int i = l & 7;
                const int dcq = S.here[49] * (int)S.q[0];
                if (l < 8) S.ns_here.horiz[i] = (int16_t)(dcq + S.pix[56 + i] + 1024 + half16(S.pix[56 + i] - S.pix[48 + i]));
                else S.ns_here.vert[i] = (int16_t)(dcq + S.pix[i * 8 + 7] + 1024 + half16(S.pix[i * 8 + 7] - S.pix[i * 8 + 6]));
            }
            if (l == 16) S.ns_here.nz = nzf;
        }
        LSYNC();
        return 0;
    }

    WDEV int run(const ImageDev* image, const SegDev& seg, uint32_t* model_words, NSum* ns, Dec3Shared* shared, const uint8_t* stream,
                 uint32_t len) {
        img = image; model = model_words; sh = shared; nbins = 0;
        init_tables();
        bc.init_stream(stream, len);
        bool top[3] = {true, true, true};
        SegmentCoder<false> sched;   // only its row schedule is used (lepton_codec.hh:41-100)
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
            LV(int16_t, nxt_above); LV(uint32_t, nxt_ns);
            LANES(l) {   // block 0's neighbours; later blocks' are fetched one block ahead
                L(nxt_above) = has_above ? arow[l] : (int16_t)0;
                L(nxt_ns) = (has_above && l < (int)(sizeof(NSum) / 4)) ? ((const uint32_t*)&narow[0])[l] : 0u;
            }
            for (int x = 0; x < w; ++x) {
                LANES(l) {
                    if (x) { sh->left[l] = sh->here[l]; sh->aleft[l] = sh->above[l]; }
                    if (l < (int)(sizeof(NSum) / 4)) {
                        if (x) ((uint32_t*)&sh->ns_left)[l] = ((const uint32_t*)&sh->ns_here)[l];
                        if (has_above) ((uint32_t*)&sh->ns_above)[l] = L(nxt_ns);
                    }
                    if (l == 0) sh->ctl[5] = 0;
                }
                LSYNC();
                LANES(l) {
                    if (has_above) sh->above[l] = L(nxt_above);
                    if (has_above && x + 1 < w) {
                        L(nxt_above) = arow[(int64_t)(x + 1) * 64 + l];
                        if (l < (int)(sizeof(NSum) / 4)) L(nxt_ns) = ((const uint32_t*)&narow[x + 1])[l];
                    }
                }
                LSYNC();
                LEP_STAMP(0);
                int rc = decode_block(x > 0, has_above);
                if (rc) return rc;
                LANES(l) {
                    row[(int64_t)x * 64 + l] = sh->here[l];
                    if (l < (int)(sizeof(NSum) / 4)) ((uint32_t*)&nrow[x])[l] = ((const uint32_t*)&sh->ns_here)[l];
                }
                LEP_STAMP(22);
                if (x + 1 < w && yb * w + x + 1 >= img->coded_blocks[comp]) break;
            }
        }
        return 0;
    }
};

}  // namespace lep3
