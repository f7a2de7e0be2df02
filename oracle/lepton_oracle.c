/*
 * lepton_oracle.c -- plain-C, scalar CPU restatement of the reference's hot path:
 *   block syntax      src/vp8/encoder/encoder.cc:194-402, src/vp8/decoder/decoder.cc:167-318
 *   edge syntax       src/vp8/encoder/encoder.cc:39-164,  src/vp8/decoder/decoder.cc:27-141
 *   context model     src/vp8/model/model.hh:60-156 (tables), :247-290 (quant-derived tables),
 *                     :463-485 (nz ctx), :852-871 (aavrg), :1033-1071 (lakhani), :674-784 (dc pred)
 *   Branch            src/vp8/model/branch.hh:82-125
 *   bool coder        src/vp8/encoder/boolwriter.hh:48-118, boolwriter.cc:17-35,
 *                     src/vp8/decoder/boolreader.hh:184-258,376-416, boolreader.cc:25-34
 *   idct              src/lepton/idct.cc:35-161
 *   neighbour summary src/vp8/util/block_context.hh:44-78
 *   row scheduling    src/lepton/lepton_codec.hh:41-100, src/lepton/vp8_encoder.cc:83-154,239-445
 *
 * TEST INFRASTRUCTURE ONLY (see lepton_oracle.h).  Parity pinned by tests/test_oracle_golden.py.
 * One routine serves both directions so the encoder and decoder cannot drift apart.
 */
#include "lepton_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- tables */
/* aligned index -> raster coordinate (aligned_block.hh:32-44) */
static const uint8_t A2R[64] = {
    9, 10, 17, 25, 18, 11, 12, 19, 26, 33, 41, 34, 27, 20, 13, 14, 21, 28, 35, 42, 49, 57, 50, 43, 36,
    29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    0, 1, 2, 3, 4, 5, 6, 7, 8, 16, 24, 32, 40, 48, 56};
static uint8_t R2A[64];
/* raster coordinate -> jpeg zig-zag index (jpeg_meta.hh:13-23) */
static const uint8_t R2Z[64] = {
    0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43,
    9, 11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
    21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
/* nonzero_to_bin[NUM_NONZEROS_BINS-1] (jpeg_meta.hh:91-92) */
static const uint8_t NZBIN[50] = {0, 1, 2, 3, 4, 4, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 7, 7, 7, 7, 8, 8, 8, 8,
                                  8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9};
/* first column of icos_base_8192_scaled (jpeg_meta.hh:48-58) */
static const int32_t ICOS_COL0[8] = {8192, 11363, 10703, 9633, 8192, 6436, 4433, 2260};
static const uint16_t FREQMAX[64] = {
    1024, 931, 985, 968, 1020, 968, 1020, 1020, 932, 858, 884, 840, 932, 838, 854, 854,
    985, 884, 871, 875, 985, 878, 871, 854, 967, 841, 876, 844, 967, 886, 870, 837,
    1020, 932, 985, 967, 1020, 969, 1020, 1020, 969, 838, 878, 886, 969, 838, 969, 838,
    1020, 854, 871, 870, 1010, 969, 1020, 1020, 1020, 854, 854, 838, 1020, 838, 1020, 838};

static int blen(uint32_t v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }
static int iabs(int v) { return v < 0 ? -v : v; }
static int imin(int a, int b) { return a < b ? a : b; }

/* ---------------------------------------------------------------- model */
typedef struct { uint8_t c0, c1, p; } Br; /* false count, true count, P(false)*256 */

typedef struct {
    Br nz7x7[2][26][6][32];
    Br nz1x8[2][8][8][3][4];
    Br nz8x1[2][8][8][3][4];
    Br resnoise[2][64][10][10];
    Br resnoise_dc[12][10];
    Br thresh[2][256][8][128];
    Br exp7x7[2][10][49][12][11];
    Br expx[2][10][15][12][11];
    Br expdc[12][17][11];
    Br sign[2][4][12];
} Model;

size_t lor_model_bytes(void) { return sizeof(Model); }

static void model_reset(Model *m) {
    Br *b = (Br *)m;
    size_t n = sizeof(Model) / sizeof(Br), i;
    for (i = 0; i < n; ++i) { b[i].c0 = 1; b[i].c1 = 1; b[i].p = 128; }
}

/* branch.hh:82-100 */
static void br_update(Br *b, int obs) {
    unsigned f = b->c0, t = b->c1;
    unsigned mine = obs ? t : f, other = obs ? f : t;
    if (mine == 255) {
        if (other == 1) {
            b->p = obs ? 0 : 255; /* counts stay (255,1) */
        } else {
            unsigned nf = (1 + f) >> 1, nt = (1 + t) >> 1;
            if (obs) nt = 129; else nf = 129;
            b->c0 = (uint8_t)nf; b->c1 = (uint8_t)nt;
            b->p = (uint8_t)((nf << 8) / (nf + nt));
        }
    } else {
        if (obs) b->c1 = (uint8_t)(t + 1); else b->c0 = (uint8_t)(f + 1);
        b->p = (uint8_t)(((unsigned)b->c0 << 8) / (f + t + 1));
    }
}

/* ---------------------------------------------------------------- bool coder */
typedef struct {
    int decode;
    uint64_t nbins;
    /* writer */
    uint32_t low, range;
    int count;
    uint8_t *out;
    size_t pos, cap;
    int overflow;
    /* reader */
    uint64_t value;
    int rcount;
    const uint8_t *in;
    size_t ipos, ilen;
} Coder;

static int norm_shift(uint32_t range) { /* vpx_norm[]: leading zeros of an 8-bit value */
    int s = 0;
    while (range < 128) { range <<= 1; ++s; }
    return s;
}

/* boolwriter.hh:48-118 */
static void wr(Coder *c, int bit, unsigned prob) {
    uint32_t split = 1 + (((c->range - 1) * prob) >> 8);
    uint32_t range, low = c->low;
    int shift, count = c->count;
    if (bit) { low += split; range = c->range - split; } else range = split;
    shift = norm_shift(range);
    range <<= shift;
    count += shift;
    if (count >= 0) {
        int offset = shift - count;
        if (c->pos + 2 > c->cap) c->overflow = 1;
        if (!c->overflow) {
            if ((low << (offset - 1)) & 0x80000000u) { /* carry into bytes already written */
                long x = (long)c->pos - 1;
                while (x >= 0 && c->out[x] == 0xff) { c->out[x] = 0; --x; }
                if (x >= 0) c->out[x] += 1;
            }
            c->out[c->pos++] = (uint8_t)(low >> (24 - offset));
        }
        low <<= offset;
        shift = count;
        low &= 0xffffff;
        count -= 8;
    }
    low <<= shift;
    c->count = count; c->low = low; c->range = range;
}

static void rd_fill(Coder *c) {
    while (c->rcount <= 48) {
        uint64_t b = c->ipos < c->ilen ? c->in[c->ipos] : 0; /* zeros past the end: boolreader.hh:234-251 */
        c->ipos++;
        c->value |= b << (48 - c->rcount);
        c->rcount += 8;
    }
}

/* boolreader.hh:376-416 */
static int rd(Coder *c, unsigned prob) {
    uint32_t split = (c->range * prob + (256 - prob)) >> 8;
    uint64_t bigsplit;
    int bit, shift;
    if (c->rcount < 0) rd_fill(c);
    bigsplit = (uint64_t)split << 56;
    bit = c->value >= bigsplit;
    if (bit) { c->range -= split; c->value -= bigsplit; } else c->range = split;
    shift = norm_shift(c->range);
    c->range <<= shift; c->value <<= shift; c->rcount -= shift;
#ifdef LEP_ORACLE_TRACE   /* the reference's own -DDEBUG_ARICODER format ("R <n> <prob> <bit>", boolreader.hh:397-413): diffable bin traces */
    { static long n_traced = 0; fprintf(stderr, "R %ld %u %d\n", n_traced++, prob, bit); }
#endif
    return bit;
}

static int code(Coder *c, Br *b, int bit) {
    c->nbins++;
    if (c->decode) bit = rd(c, b->p); else wr(c, bit, b->p);
    br_update(b, bit);
    return bit;
}

/* ---------------------------------------------------------------- per-component derived tables */
typedef struct {
    uint16_t q[64];      /* raster */
    int32_t icos_x[64];  /* model.hh:254 */
    int32_t icos_y[64];  /* model.hh:255 */
    uint8_t min_thresh[64];
} QTab;

static int qtab_init(QTab *t, const uint16_t *qzz, int check_zero) {
    int i, r;
    for (i = 0; i < 64; ++i) t->q[i] = qzz[R2Z[i]];
    for (r = 0; r < 8; ++r)
        for (i = 0; i < 8; ++i) {
            t->icos_x[r * 8 + i] = ICOS_COL0[i] * (int32_t)t->q[i * 8 + r];
            t->icos_y[r * 8 + i] = ICOS_COL0[i] * (int32_t)t->q[r * 8 + i];
        }
    for (r = 0; r < 8; ++r)
        if (t->icos_x[r * 8] == 0 || t->icos_y[r * 8] == 0) { if (check_zero) return LOR_UNSUPPORTED_ZERO_IDCT_0; }
    for (i = 0; i < 64; ++i) {
        unsigned fm = FREQMAX[i] + t->q[i] - 1;
        int len;
        if (t->q[i]) fm /= t->q[i];
        len = blen(fm & 0xffff);
        t->min_thresh[i] = (uint8_t)(len > 7 ? len - 7 : 0);
    }
    return 0;
}

/* idct.cc:35-161 with ignore_dc = true; blk is in aligned order */
static void idct_sans_dc(const int16_t *blk, const uint16_t *q, int16_t *outp) {
    enum { w1 = 2841, w2 = 2676, w3 = 2408, w5 = 1609, w6 = 1108, w7 = 565,
           w1pw7 = w1 + w7, w1mw7 = w1 - w7, w2pw6 = w2 + w6, w2mw6 = w2 - w6,
           w3pw5 = w3 + w5, w3mw5 = w3 - w5, r2 = 181 };
    int32_t t[64];
    int y, x;
#define CQ(i) ((int32_t)blk[R2A[i]] * (int32_t)q[i])
    for (y = 0; y < 8; ++y) {
        int y8 = y * 8;
        int32_t x0 = ((y == 0) ? 0 : (int32_t)((uint32_t)CQ(y8) << 11)) + 128;
        int32_t x1 = (int32_t)((uint32_t)CQ(y8 + 4) << 11);
        int32_t x2 = CQ(y8 + 6), x3 = CQ(y8 + 2), x4 = CQ(y8 + 1), x5 = CQ(y8 + 7), x6 = CQ(y8 + 5), x7 = CQ(y8 + 3);
        int32_t x8;
        x8 = w7 * (x4 + x5); x4 = x8 + w1mw7 * x4; x5 = x8 - w1pw7 * x5;
        x8 = w3 * (x6 + x7); x6 = x8 - w3mw5 * x6; x7 = x8 - w3pw5 * x7;
        x8 = x0 + x1; x0 -= x1;
        x1 = w6 * (x3 + x2); x2 = x1 - w2pw6 * x2; x3 = x1 + w2mw6 * x3;
        x1 = x4 + x6; x4 -= x6; x6 = x5 + x7; x5 -= x7;
        x7 = x8 + x3; x8 -= x3; x3 = x0 + x2; x0 -= x2;
        x2 = (r2 * (x4 + x5) + 128) >> 8;
        x4 = (r2 * (x4 - x5) + 128) >> 8;
        t[y8 + 0] = (x7 + x1) >> 8; t[y8 + 1] = (x3 + x2) >> 8; t[y8 + 2] = (x0 + x4) >> 8; t[y8 + 3] = (x8 + x6) >> 8;
        t[y8 + 4] = (x8 - x6) >> 8; t[y8 + 5] = (x0 - x4) >> 8; t[y8 + 6] = (x3 - x2) >> 8; t[y8 + 7] = (x7 - x1) >> 8;
    }
#undef CQ
    for (x = 0; x < 8; ++x) {
        int32_t y0 = (int32_t)((uint32_t)t[x] << 8) + 8192;
        int32_t y1 = (int32_t)((uint32_t)t[32 + x] << 8);
        int32_t y2 = t[48 + x], y3 = t[16 + x], y4 = t[8 + x], y5 = t[56 + x], y6 = t[40 + x], y7 = t[24 + x];
        int32_t y8;
        y8 = w7 * (y4 + y5) + 4; y4 = (y8 + w1mw7 * y4) >> 3; y5 = (y8 - w1pw7 * y5) >> 3;
        y8 = w3 * (y6 + y7) + 4; y6 = (y8 - w3mw5 * y6) >> 3; y7 = (y8 - w3pw5 * y7) >> 3;
        y8 = y0 + y1; y0 -= y1;
        y1 = w6 * (y3 + y2) + 4; y2 = (y1 - w2pw6 * y2) >> 3; y3 = (y1 + w2mw6 * y3) >> 3;
        y1 = y4 + y6; y4 -= y6; y6 = y5 + y7; y5 -= y7;
        y7 = y8 + y3; y8 -= y3; y3 = y0 + y2; y0 -= y2;
        y2 = (r2 * (y4 + y5) + 128) >> 8;
        y4 = (r2 * (y4 - y5) + 128) >> 8;
        outp[x] = (int16_t)((y7 + y1) >> 11); outp[8 + x] = (int16_t)((y3 + y2) >> 11);
        outp[16 + x] = (int16_t)((y0 + y4) >> 11); outp[24 + x] = (int16_t)((y8 + y6) >> 11);
        outp[32 + x] = (int16_t)((y8 - y6) >> 11); outp[40 + x] = (int16_t)((y0 - y4) >> 11);
        outp[48 + x] = (int16_t)((y3 - y2) >> 11); outp[56 + x] = (int16_t)((y7 - y1) >> 11);
    }
}

/* ---------------------------------------------------------------- neighbour summary */
typedef struct { int16_t vert[8]; int16_t horiz[8]; uint8_t nz; } NSum; /* block_context.hh:17-43 */

/* "shift right, round toward zero" on a value already wrapped to int16 (model.hh:673) */
static int half16(int delta) { int16_t d = (int16_t)delta; return d / 2; }

/* ---------------------------------------------------------------- one block */
typedef struct {
    Model *m;
    Coder *c;
    const QTab *qt;
    int ci; /* color index: 0 luma, 1 chroma (model.hh:373-382) */
} Ctx;

/* unary exponent, shared by 7x7 / edge / dc; returns the length, or -1 if out of range */
static int code_exponent(Ctx *k, Br *E, int len) {
    int i;
    if (!k->c->decode && len > 11) return -1;
    for (i = 0; i < 11; ++i)
        if (!code(k->c, &E[i], len != i)) break;
    return i;
}

static int code_residual(Ctx *k, Br *R, int hi, int v) { /* bits hi..0 of v */
    int i;
    for (i = hi; i >= 0; --i) v = (v & ~(1 << i)) | (code(k->c, &R[i], (v >> i) & 1) << i);
    return v;
}

/* lakhani edge prior: model.hh:1033-1071 (== compute_lak_vec :928-958) */
static int32_t lakhani(const int16_t *here, const int16_t *nbr, const int32_t *icos, int band, int step) {
    uint32_t acc = (uint32_t)nbr[R2A[band]] * (uint32_t)icos[0];
    int i;
    for (i = 1; i < 8; ++i) {
        int32_t xi = here[R2A[band + i * step]];
        int32_t ai = nbr[R2A[band + i * step]];
        int32_t term = (i & 1) ? xi + ai : xi - ai;
        acc -= (uint32_t)icos[i] * (uint32_t)term;
    }
    return (int32_t)acc / icos[0];
}

/* Test knob (tests/test_core_emulation.py, tests/test_gpu_parity.py): the ENCODER claims this many more edge non-zeros than
 * the block holds (capped at 7).  An encoder never writes such a stream; a damaged or hostile one can hold it, and the
 * reference's decoder then simply indexes exponent_counts_x_ with a "non-zeros left" that exceeds the positions left
 * (decoder.cc:58-141) -- the decode direction below does the same, so the stream decodes back to the original block. */
int lor_test_edge_count_bias = 0;
static int code_edge(Ctx *k, int16_t *here, const int16_t *nbr, int horizontal, int nz7x7, int est_eob) {
    Coder *c = k->c;
    int ci = k->ci;
    Br(*T)[4] = horizontal ? k->m->nz8x1[ci][est_eob][(nz7x7 + 3) / 7] : k->m->nz1x8[ci][est_eob][(nz7x7 + 3) / 7];
    int delta = horizontal ? 1 : 8, a_off = horizontal ? 50 : 57, zig15 = horizontal ? 0 : 7;
    int ne = 0, so_far = 0, i, lane, coord;
    if (!c->decode) {
        for (i = 0; i < 7; ++i) ne += here[a_off + i] != 0;
        ne = imin(7, ne + lor_test_edge_count_bias); /* test knob, 0 outside tests: see lepton_oracle.h */
    }
    for (i = 2; i >= 0; --i) {
        int bit = code(c, &T[i][so_far], (ne >> i) & 1);
        if (c->decode) ne |= bit << i;
        so_far = (so_far << 1) | bit;
    }
    if (ne > 7) return LOR_STREAM_INCONSISTENT; /* unreachable with 3 bits; kept for symmetry with decoder.cc:60 */
    coord = delta;
    for (lane = 0; lane < 7 && ne; ++lane, coord += delta, ++zig15) {
        int32_t prior = 0;
        int bsr, len, v, coef;
        if (nbr) {
            const int32_t *icos = horizontal ? k->qt->icos_x + coord * 8 : k->qt->icos_y + coord;
            if (icos[0] == 0) return LOR_UNSUPPORTED_ZERO_IDCT_0;
            prior = lakhani(here, nbr, icos, coord, horizontal ? 8 : 1);
        }
        bsr = blen((uint32_t)imin(prior < 0 ? (prior == INT32_MIN ? 1023 : -prior) : prior, 1023));
        coef = here[a_off + lane];
        v = iabs(coef);
        len = code_exponent(k, k->m->expx[ci][ne][zig15][bsr], blen((uint32_t)v));
        if (len < 0) return LOR_COEFFICIENT_OUT_OF_RANGE;
        if (len) {
            int16_t p16 = (int16_t)prior;
            int sctx = p16 == 0 ? 0 : (p16 > 0 ? 1 : 2);
            int thr = k->qt->min_thresh[coord];
            int pos = code(c, &k->m->sign[ci][sctx][bsr], coef >= 0);
            int ne_before = ne;
            --ne;
            if (c->decode) v = 1 << (len - 1);
            if (len > 1) {
                int b = len - 2;
                if (b >= thr) {
                    unsigned ctx_abs = (uint16_t)(prior < 0 ? -(int64_t)prior : prior);
                    Br *Tt = k->m->thresh[ci][imin((int)(ctx_abs >> thr), 255)][imin(len - thr, 7)];
                    int s = 1;
                    for (; b >= thr; --b) {
                        int bit = code(c, &Tt[s], (v >> b) & 1);
                        v = (v & ~(1 << b)) | (bit << b);
                        s = imin((s << 1) | bit, 127);
                    }
                }
                v = code_residual(k, k->m->resnoise[ci][coord][ne_before], b, v);
            }
            if (c->decode) here[a_off + lane] = (int16_t)(pos ? v : -v);
        } else if (c->decode) {
            here[a_off + lane] = 0;
        }
    }
    return 0;
}

/* here/left/above/aboveleft: 64 int16 in aligned order (NULL when absent);
 * ns_here is written, ns_left / ns_above read. */
static int code_block(Ctx *k, int16_t *here, const int16_t *left, const int16_t *above, const int16_t *aleft,
                      NSum *ns_here, const NSum *ns_left, const NSum *ns_above) {
    Coder *c = k->c;
    Model *m = k->m;
    int ci = k->ci;
    int nz = 0, nzctx = 0, so_far = 0, left_nz, eob_x = 0, eob_y = 0, zz, i, rc;
    int16_t pix[64];
    const uint16_t *q = k->qt->q;

    if (c->decode) memset(here, 0, 128);
    else
        for (i = 0; i < 49; ++i) nz += here[i] != 0; /* aligned_block.hh:132-148 */

    /* 1. number of nonzeros in the 7x7 interior (model.hh:463-485, encoder.cc:200-213) */
    if (left && above) nzctx = (ns_above->nz + ns_left->nz + 2) / 4;
    else if (above) nzctx = (ns_above->nz + 1) / 2;
    else if (left) nzctx = (ns_left->nz + 1) / 2;
    {
        Br(*T)[32] = m->nz7x7[ci][NZBIN[nzctx]];
        for (i = 5; i >= 0; --i) {
            int bit = code(c, &T[i][so_far], (nz >> i) & 1);
            if (c->decode) nz |= bit << i;
            so_far = (so_far << 1) | bit;
        }
    }
    if (nz > 49) return LOR_STREAM_INCONSISTENT;

    /* 2. the 7x7 interior in zig-zag order (encoder.cc:219-285) */
    left_nz = nz;
    for (zz = 0; zz < 49 && left_nz; ++zz) {
        int coord = A2R[zz], prior, nb, bsr, len, v, coef;
        if (left && above) {
            int tot = (iabs(left[zz]) + iabs(above[zz])) * 13 + 6 * iabs(aleft[zz]);
            prior = (uint16_t)tot >> 5;
        } else if (left) prior = (int16_t)iabs(left[zz]);
        else if (above) prior = (int16_t)iabs(above[zz]);
        else prior = 0;
        nb = NZBIN[left_nz];
        bsr = blen((uint32_t)imin(iabs(prior), 1023));
        coef = here[zz];
        v = iabs(coef);
        len = code_exponent(k, m->exp7x7[ci][nb][zz][bsr], blen((uint32_t)v));
        if (len < 0) return LOR_COEFFICIENT_OUT_OF_RANGE;
        if (len) {
            int pos = code(c, &m->sign[ci][0][0], coef >= 0);
            --left_nz;
            if ((coord & 7) > eob_x) eob_x = coord & 7;
            if ((coord >> 3) > eob_y) eob_y = coord >> 3;
            if (c->decode) v = 1 << (len - 1);
            if (len > 1) v = code_residual(k, m->resnoise[ci][coord][nb], len - 2, v);
            if (c->decode) here[zz] = (int16_t)(pos ? v : -v);
        }
    }

    /* 3. the two edges (encoder.cc:166-184) */
    if ((rc = code_edge(k, here, above, 1, nz, eob_x))) return rc;
    if ((rc = code_edge(k, here, left, 0, nz, eob_y))) return rc;

    /* 4. DC: pixel-domain prediction from the neighbours' cached edges (model.hh:674-784) */
    idct_sans_dc(here, q, pix);
    {
        int32_t avgmed = 0, unc = 0, unc2 = 0, pred;
        if (left || above) {
            int16_t est[16];
            int n = 0, sum0 = 0, sum1 = 0, mn, mx;
            if (left)
                for (i = 0; i < 8; ++i, ++n)
                    est[n] = (int16_t)(ns_left->vert[i] - half16(pix[i * 8] - pix[i * 8 + 1]) - (pix[i * 8] + 1024));
            if (above)
                for (i = 0; i < 8; ++i, ++n)
                    est[n] = (int16_t)(ns_above->horiz[i] - half16(pix[i] - pix[i + 8]) - (pix[i] + 1024));
            mn = mx = est[0];
            for (i = 0; i < n; ++i) {
                if (i < 8) sum0 += est[i]; else sum1 += est[i];
                if (est[i] < mn) mn = est[i];
                if (est[i] > mx) mx = est[i];
            }
            if (n == 8) sum1 = sum0;
            avgmed = (sum0 + sum1) >> 1;
            unc = (mx - mn) >> 3;
            sum0 -= avgmed; sum1 -= avgmed;
            unc2 = (iabs(sum0) < iabs(sum1) ? sum0 : sum1) >> 3;
        }
        pred = (avgmed / (int)q[0] + 4) >> 3;
        {
            int a = imin(blen((uint16_t)iabs(unc)), 11), b = imin(blen((uint16_t)iabs(unc2)), 16);
            int dc = here[49], d = 0, len, v, pos = 1;
            if (!c->decode) { /* model.hh:823-832, encoder.cc:305-313 */
                int back;
                d = dc - pred;
                if (d < -1024) d += 2049;
                if (d > 1024) d -= 2049;
                back = d + pred;
                if (back < -1024) back += 2049;
                if (back > 1024) back -= 2049;
                if (back != dc) return LOR_COEFFICIENT_OUT_OF_RANGE;
                d = (int16_t)d;
            }
            v = iabs(d);
            len = code_exponent(k, m->expdc[a][b], blen((uint16_t)v));
            if (len < 0) return LOR_COEFFICIENT_OUT_OF_RANGE;
            if (len) {
                pos = code(c, &m->sign[ci][0][unc2 >= 0 ? (unc2 == 0 ? 3 : 2) : 1], d >= 0);
                if (c->decode) v = 1 << (len - 1);
                if (len > 1) v = code_residual(k, m->resnoise_dc[a], len - 2, v);
            }
            if (c->decode) {
                d = (int16_t)(len ? (pos ? v : -v) : 0);
                dc = d + pred;
                if (dc < -1024) dc += 2049;
                if (dc > 1024) dc -= 2049;
                here[49] = (int16_t)dc;
            }
        }
    }

    /* 5. publish this block's summary (block_context.hh:44-78, SSE semantics: int16 wrap) */
    ns_here->nz = (uint8_t)nz;
    {
        int dcq = here[49] * (int)q[0];
        for (i = 0; i < 8; ++i) {
            ns_here->horiz[i] = (int16_t)(dcq + pix[56 + i] + 1024 + half16(pix[56 + i] - pix[48 + i]));
            ns_here->vert[i] = (int16_t)(dcq + pix[i * 8 + 7] + 1024 + half16(pix[i * 8 + 7] - pix[i * 8 + 6]));
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- row scheduling */
typedef struct { int component, curr_y, luma_y, skip, done; } RowSpec;

/* lepton_codec.hh:41-100 (3 colour channels; absent components have multiple 0) */
static RowSpec row_spec(const lor_image *img, unsigned idx) {
    unsigned mult[3] = {0, 0, 0}, total = 0, mcu_row, place;
    RowSpec r = {3, 0, 0, 0, 0};
    int i;
    for (i = 0; i < 3 && i < img->ncomp; ++i) { mult[i] = (unsigned)img->height_blocks[i] / (unsigned)img->mcu_rows; total += mult[i]; }
    mcu_row = idx / total;
    place = idx - mcu_row * total;
    r.luma_y = (int)(mcu_row * mult[0]);
    for (i = 2; i >= 0; --i) {
        if (place < mult[i]) {
            r.component = i;
            r.curr_y = (int)(mcu_row * mult[i] + place);
            if (r.curr_y >= img->coded_height[i]) {
                int j;
                r.skip = 1; r.done = 1;
                for (j = 0; j < 2; ++j)
                    if ((int)(mcu_row * mult[j]) < (j < img->ncomp ? img->coded_height[j] : 0)) r.done = 0;
            }
            if (i == 0) r.luma_y = r.curr_y;
            return r;
        }
        place -= mult[i];
    }
    r.skip = 1; r.done = 1;
    return r;
}

static int run_segment(lor_image *img, int y0, int y1, int is_last, Coder *c) {
    Model *m = (Model *)malloc(sizeof(Model));
    QTab qt[3];
    NSum *ns[3] = {0, 0, 0};
    int top[3] = {1, 1, 1};
    int rc = 0, i;
    unsigned idx = 0;
    if (!m) return LOR_CODING_ERROR;
    model_reset(m);
    for (i = 0; i < img->ncomp && i < 3; ++i) {
        if ((rc = qtab_init(&qt[i], img->qtable_zigzag[i], !c->decode))) goto out;
        ns[i] = (NSum *)calloc((size_t)img->width_blocks[i] * 2, sizeof(NSum));
    }
    for (;;) {
        RowSpec r = row_spec(img, idx++);
        int w, x, yb, comp;
        int16_t *row, *arow;
        NSum *nrow, *narow;
        Ctx k;
        if (r.done) break;
        if (r.luma_y >= y1 && !is_last) break;
        if (r.skip) continue;
        if (r.luma_y < y0) continue;
        comp = r.component;
        w = img->width_blocks[comp];
        yb = r.curr_y;
        row = img->blocks[comp] + (size_t)yb * w * 64;
        arow = top[comp] ? NULL : row - (size_t)w * 64;
        nrow = ns[comp] + (size_t)(yb & 1) * w;
        narow = ns[comp] + (size_t)((yb & 1) ^ 1) * w;
        top[comp] = 0;
        k.m = m; k.c = c; k.qt = &qt[comp]; k.ci = comp ? 1 : 0;
        for (x = 0; x < w; ++x) {
            int16_t *here = row + (size_t)x * 64;
            const int16_t *l = x ? here - 64 : NULL;
            const int16_t *a = arow ? arow + (size_t)x * 64 : NULL;
            const int16_t *al = (x && arow) ? a - 64 : NULL;
            if ((rc = code_block(&k, here, l, a, al, &nrow[x], x ? &nrow[x - 1] : NULL, arow ? &narow[x] : NULL))) goto out;
            /* truncated files: stop at the last coded block (vp8_encoder.cc:107-111) */
            if (x + 1 < w && yb * w + x + 1 >= img->coded_blocks[comp]) break;
        }
    }
out:
    for (i = 0; i < 3; ++i) free(ns[i]);
    free(m);
    return rc;
}

int lor_encode_segment(const lor_image *img, int y0, int y1, int is_last, uint8_t *out, size_t cap, size_t *out_len,
                       uint64_t *bins) {
    Coder c;
    int rc, i;
    memset(&c, 0, sizeof c);
    if (!R2A[63]) for (i = 0; i < 64; ++i) R2A[A2R[i]] = (uint8_t)i;
    c.out = out; c.cap = cap;
    c.range = 255; c.count = -24; /* boolwriter.cc:17-24 */
    wr(&c, 0, 128);
    rc = run_segment((lor_image *)img, y0, y1, is_last, &c);
    if (rc) return rc;
    for (i = 0; i < 32; ++i) wr(&c, 0, 128); /* boolwriter.cc:26-35 */
    if (c.pos && (c.out[c.pos - 1] & 0xe0) == 0xc0) c.out[c.pos++] = 0;
    if (c.overflow) return LOR_BUFFER_TOO_SMALL;
    *out_len = c.pos;
    if (bins) *bins = c.nbins;
    return 0;
}

int lor_decode_segment(lor_image *img, int y0, int y1, int is_last, const uint8_t *in, size_t len, uint64_t *bins) {
    Coder c;
    int rc, i;
    memset(&c, 0, sizeof c);
    if (!R2A[63]) for (i = 0; i < 64; ++i) R2A[A2R[i]] = (uint8_t)i;
    c.decode = 1; c.in = in; c.ilen = len;
    c.value = 0; c.rcount = -8; c.range = 255; /* boolreader.cc:25-34 */
    rd_fill(&c);
    rd(&c, 128);
    rc = run_segment(img, y0, y1, is_last, &c);
    if (bins) *bins = c.nbins;
    return rc;
}
