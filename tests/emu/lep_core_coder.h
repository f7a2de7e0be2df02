// lep_core_coder.h -- TEST INFRASTRUCTURE: the single-lane coder of round 1 (one thread segment coded by one lane, every bin a model
// round trip) -- the plainest statement of the block syntax in this repository, kept as a CPU cross-check of the wave-cooperative
// kernels (tests/emu/core_emu.cc compiles it with g++).  Not part of the product: nothing under lepton_amd/ includes it.
// Reference citations: lepton_amd/csrc/lep_core.h.
#pragma once
#include "../../lepton_amd/csrc/lep_core.h"

namespace lepdev {

// ---- adaptive binary arithmetic coder ------------------------------------------------------------
template <bool DEC>
struct BoolCoder;

template <>
struct BoolCoder<false> {   // writer
    uint32_t low, range;
    int count;
    uint8_t* out;
    uint32_t pos, cap;
    bool overflow;
    LEP_DEV void init_stream(uint8_t* o, uint32_t c) {
        out = o; cap = c; pos = 0; overflow = false;
        low = 0; range = 255; count = -24;
        put(0, 128);
    }
    LEP_DEV void put(int bit, uint32_t prob) {
        uint32_t split = 1 + (((range - 1) * prob) >> 8);
        uint32_t r = split, l = low;
        if (bit) { l += split; r = range - split; }
        int shift = __builtin_clz(r) - 24;
        r <<= shift;
        int c = count + shift;
        if (c >= 0) {
            int offset = shift - c;
            if (pos + 2 > cap) overflow = true;
            if (!overflow) {
                if ((l << (offset - 1)) & 0x80000000u) {
                    int x = (int)pos - 1;
                    while (x >= 0 && out[x] == 0xff) { out[x] = 0; --x; }
                    if (x >= 0) out[x] = (uint8_t)(out[x] + 1);
                }
                out[pos++] = (uint8_t)(l >> (24 - offset));
            }
            l <<= offset;
            shift = c;
            l &= 0xffffff;
            c -= 8;
        }
        l <<= shift;
        count = c; low = l; range = r;
    }
    LEP_DEV uint32_t finish() {
        for (int i = 0; i < 32; ++i) put(0, 128);
        if (!overflow && pos && (out[pos - 1] & 0xe0) == 0xc0) out[pos++] = 0;
        return pos;
    }
};

template <>
struct BoolCoder<true> {   // reader: 64-bit window, zero bits past the end of the stream
    uint64_t value;
    int count;
    uint32_t range;
    const uint8_t* in;
    uint32_t ipos, ilen;
    LEP_DEV void fill() {
        while (count <= 48) {
            uint64_t b = ipos < ilen ? in[ipos] : 0;
            ++ipos;
            value |= b << (48 - count);
            count += 8;
        }
    }
    LEP_DEV void init_stream(const uint8_t* i, uint32_t n) {
        in = i; ilen = n; ipos = 0;
        value = 0; count = -8; range = 255;
        fill();
        get(128);
    }
    LEP_DEV int get(uint32_t prob) {
        uint32_t split = (range * prob + (256 - prob)) >> 8;
        if (count < 0) fill();
        uint64_t big = (uint64_t)split << 56;
        int bit = value >= big;
        if (bit) { range -= split; value -= big; } else range = split;
#ifdef LEP_TRACE_GET
        LEP_TRACE_GET(prob, bit);
#endif
        int shift = __builtin_clz(range) - 24;
        range <<= shift; value <<= shift; count -= shift;
        return bit;
    }
};

// ---- one segment -------------------------------------------------------------------------------
template <bool DEC>
struct SegmentCoder {
    BoolCoder<DEC> bc;
    uint32_t* model;
    const ImageDev* img;
    int comp, ci;            // component, colour index (0 luma / 1 chroma)
    const uint16_t* q;
    uint32_t nbins;

    LEP_DEV int code(uint32_t idx, int bit) {
        uint32_t w = model[idx];
        if (DEC) bit = bc_get(w >> 16); else bc_put(bit, w >> 16);
        model[idx] = branch_update(w, bit);
        ++nbins;
        return bit;
    }
    LEP_DEV int bc_get(uint32_t p) { return get_impl(bc, p); }
    LEP_DEV void bc_put(int bit, uint32_t p) { put_impl(bc, bit, p); }
    LEP_DEV static int get_impl(BoolCoder<true>& b, uint32_t p) { return b.get(p); }
    LEP_DEV static int get_impl(BoolCoder<false>&, uint32_t) { return 0; }
    LEP_DEV static void put_impl(BoolCoder<false>& b, int bit, uint32_t p) { b.put(bit, p); }
    LEP_DEV static void put_impl(BoolCoder<true>&, int, uint32_t) {}

    // unary exponent over 11 consecutive branches; returns the bit length, -1 if not representable
    LEP_DEV int code_exponent(uint32_t base, int len) {
        if (!DEC && len > 11) return -1;
        int i = 0;
        for (; i < 11; ++i)
            if (!code(base + i, len != i)) break;
        return i;
    }
    LEP_DEV int code_bits(uint32_t base, int hi, int v) {   // bits hi..0 of v through branches base+i
        for (int i = hi; i >= 0; --i) {
            int b = code(base + i, (v >> i) & 1);
            v = (v & ~(1 << i)) | (b << i);
        }
        return v;
    }

    LEP_DEV static int32_t lakhani(const int16_t* here, const int16_t* nbr, const int32_t* icos, int band, int step) {
        uint32_t acc = (uint32_t)(int32_t)nbr[kR2A[band]] * (uint32_t)icos[0];
        for (int i = 1; i < 8; ++i) {
            int32_t xi = here[kR2A[band + i * step]], ai = nbr[kR2A[band + i * step]];
            int32_t term = (i & 1) ? xi + ai : xi - ai;
            acc -= (uint32_t)icos[i] * (uint32_t)term;
        }
        return (int32_t)acc / icos[0];
    }

    LEP_DEV int code_edge(int16_t* here, const int16_t* nbr, bool horizontal, int nz7x7, int est_eob) {
        uint32_t T = (horizontal ? kNz8x1 : kNz1x8) + (((uint32_t)ci * 8 + est_eob) * 8 + (nz7x7 + 3) / 7) * 12;
        const int delta = horizontal ? 1 : 8, a_off = horizontal ? 50 : 57;
        int zig15 = horizontal ? 0 : 7, ne = 0, so_far = 0;
        if (!DEC)
            for (int i = 0; i < 7; ++i) ne += here[a_off + i] != 0;
        for (int i = 2; i >= 0; --i) {
            int bit = code(T + i * 4 + so_far, (ne >> i) & 1);
            if (DEC) ne |= bit << i;
            so_far = (so_far << 1) | bit;
        }
        if (ne > 7) return 7;
        int coord = delta;
        for (int lane = 0; lane < 7 && ne; ++lane, coord += delta, ++zig15) {
            int32_t prior = 0;
            if (nbr) {
                const int32_t* icos = horizontal ? img->icos_x[comp] + coord * 8 : img->icos_y[comp] + coord;
                if (icos[0] == 0) return 43;
                prior = lakhani(here, nbr, icos, coord, horizontal ? 8 : 1);
            }
            uint32_t aprior = prior < 0 ? 0u - (uint32_t)prior : (uint32_t)prior;
            int bsr = bitlen(aprior > 1023 ? 1023 : aprior);
            int coef = here[a_off + lane], v = iabs(coef);
            int len = code_exponent(kExpX + ((((uint32_t)ci * 10 + ne) * 15 + zig15) * 12 + bsr) * 11, bitlen((uint32_t)v));
            if (len < 0) return 6;
            if (len) {
                int16_t p16 = (int16_t)prior;
                int sctx = p16 == 0 ? 0 : (p16 > 0 ? 1 : 2);
                int thr = img->min_thresh[comp][coord];
                int pos = code(kSign + ((uint32_t)ci * 4 + sctx) * 12 + bsr, coef >= 0);
                int ne_before = ne;
                --ne;
                if (DEC) v = 1 << (len - 1);
                if (len > 1) {
                    int b = len - 2;
                    if (b >= thr) {
                        uint32_t ctx_abs = aprior & 0xffff;
                        uint32_t Tt = kThresh + ((((uint32_t)ci * 256 + (uint32_t)imin((int)(ctx_abs >> thr), 255)) * 8) +
                                                 (uint32_t)imin(len - thr, 7)) * 128;
                        int s = 1;
                        for (; b >= thr; --b) {
                            int bit = code(Tt + s, (v >> b) & 1);
                            v = (v & ~(1 << b)) | (bit << b);
                            s = imin((s << 1) | bit, 127);
                        }
                    }
                    v = code_bits(kRes + (((uint32_t)ci * 64 + coord) * 10 + ne_before) * 10, b, v);
                }
                if (DEC) here[a_off + lane] = (int16_t)(pos ? v : -v);
            }
        }
        return 0;
    }

    // integer IDCT of the block with its DC forced to zero (idct.cc:35-161)
    LEP_DEV void idct_sans_dc(const int16_t* blk, int16_t* outp) {
        constexpr int w1 = 2841, w2 = 2676, w3 = 2408, w5 = 1609, w6 = 1108, w7 = 565, r2 = 181;
        constexpr int w1pw7 = w1 + w7, w1mw7 = w1 - w7, w2pw6 = w2 + w6, w2mw6 = w2 - w6, w3pw5 = w3 + w5, w3mw5 = w3 - w5;
        int32_t t[64];
        for (int y = 0; y < 8; ++y) {
            int y8 = y * 8;
#define LEP_CQ(i) ((int32_t)blk[kR2A[i]] * (int32_t)q[i])
            int32_t x0 = (y == 0 ? 0 : (int32_t)((uint32_t)LEP_CQ(y8) << 11)) + 128;
            int32_t x1 = (int32_t)((uint32_t)LEP_CQ(y8 + 4) << 11);
            int32_t x2 = LEP_CQ(y8 + 6), x3 = LEP_CQ(y8 + 2), x4 = LEP_CQ(y8 + 1), x5 = LEP_CQ(y8 + 7), x6 = LEP_CQ(y8 + 5),
                    x7 = LEP_CQ(y8 + 3), x8;
#undef LEP_CQ
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
        for (int x = 0; x < 8; ++x) {
            int32_t y0 = (int32_t)((uint32_t)t[x] << 8) + 8192, y1 = (int32_t)((uint32_t)t[32 + x] << 8);
            int32_t y2 = t[48 + x], y3 = t[16 + x], y4 = t[8 + x], y5 = t[56 + x], y6 = t[40 + x], y7 = t[24 + x], y8;
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

    LEP_DEV static int half16(int d) { return (int16_t)d / 2; }   // int16 wrap, then round toward zero

    LEP_DEV int code_block(int16_t* here, const int16_t* left, const int16_t* above, const int16_t* aleft, NSum* ns_here,
                           const NSum* ns_left, const NSum* ns_above) {
        int nz = 0, nzctx = 0, so_far = 0, eob_x = 0, eob_y = 0;
        if (DEC) { for (int i = 0; i < 64; ++i) here[i] = 0; }
        else { for (int i = 0; i < 49; ++i) nz += here[i] != 0; }
        // 1. number of non-zeros in the 7x7 interior
        if (left && above) nzctx = (ns_above->nz + ns_left->nz + 2) / 4;
        else if (above) nzctx = (ns_above->nz + 1) / 2;
        else if (left) nzctx = (ns_left->nz + 1) / 2;
        {
            uint32_t T = kNz7x7 + ((uint32_t)ci * 26 + kNzBin[nzctx]) * 192;
            for (int i = 5; i >= 0; --i) {
                int bit = code(T + i * 32 + so_far, (nz >> i) & 1);
                if (DEC) nz |= bit << i;
                so_far = (so_far << 1) | bit;
            }
        }
        if (nz > 49) return 7;
        // 2. interior coefficients in zig-zag order
        int left_nz = nz;
        for (int zz = 0; zz < 49 && left_nz; ++zz) {
            int coord = kA2R[zz], prior;
            if (left && above) prior = (uint16_t)((iabs(left[zz]) + iabs(above[zz])) * 13 + 6 * iabs(aleft[zz])) >> 5;
            else if (left) prior = (int16_t)iabs(left[zz]);
            else if (above) prior = (int16_t)iabs(above[zz]);
            else prior = 0;
            int nb = kNzBin[left_nz];
            int bsr = bitlen((uint32_t)imin(iabs(prior), 1023));
            int coef = here[zz], v = iabs(coef);
            int len = code_exponent(kExp7 + ((((uint32_t)ci * 10 + nb) * 49 + zz) * 12 + bsr) * 11, bitlen((uint32_t)v));
            if (len < 0) return 6;
            if (len) {
                int pos = code(kSign + (uint32_t)ci * 48, coef >= 0);
                --left_nz;
                if ((coord & 7) > eob_x) eob_x = coord & 7;
                if ((coord >> 3) > eob_y) eob_y = coord >> 3;
                if (DEC) v = 1 << (len - 1);
                if (len > 1) v = code_bits(kRes + (((uint32_t)ci * 64 + coord) * 10 + nb) * 10, len - 2, v);
                if (DEC) here[zz] = (int16_t)(pos ? v : -v);
            }
        }
        // 3. edges
        int rc = code_edge(here, above, true, nz, eob_x);
        if (rc) return rc;
        rc = code_edge(here, left, false, nz, eob_y);
        if (rc) return rc;
        // 4. DC
        int16_t pix[64];
        idct_sans_dc(here, pix);
        int32_t avgmed = 0, unc = 0, unc2 = 0;
        if (left || above) {
            int16_t est[16];
            int n = 0;
            if (left)
                for (int i = 0; i < 8; ++i, ++n)
                    est[n] = (int16_t)(ns_left->vert[i] - half16(pix[i * 8] - pix[i * 8 + 1]) - (pix[i * 8] + 1024));
            if (above)
                for (int i = 0; i < 8; ++i, ++n)
                    est[n] = (int16_t)(ns_above->horiz[i] - half16(pix[i] - pix[i + 8]) - (pix[i] + 1024));
            int sum0 = 0, sum1 = 0, mn = est[0], mx = est[0];
            for (int i = 0; i < n; ++i) {
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
        int pred = (avgmed / (int)q[0] + 4) >> 3;
        {
            int a = imin(bitlen((uint32_t)iabs(unc) & 0xffff), 11), b = imin(bitlen((uint32_t)iabs(unc2) & 0xffff), 16);
            int dc = here[49], d = 0, pos = 1;
            if (!DEC) {
                d = dc - pred;
                if (d < -1024) d += 2049;
                if (d > 1024) d -= 2049;
                int back = d + pred;
                if (back < -1024) back += 2049;
                if (back > 1024) back -= 2049;
                if (back != dc) return 6;
            }
            int v = iabs(d);
            int len = code_exponent(kExpDc + ((uint32_t)a * 17 + b) * 11, bitlen((uint32_t)v & 0xffff));
            if (len < 0) return 6;
            if (len) {
                pos = code(kSign + (uint32_t)ci * 48 + (unc2 >= 0 ? (unc2 == 0 ? 3 : 2) : 1), d >= 0);
                if (DEC) v = 1 << (len - 1);
                if (len > 1) v = code_bits(kResDc + (uint32_t)a * 10, len - 2, v);
            }
            if (DEC) {
                d = (int16_t)(len ? (pos ? v : -v) : 0);
                dc = d + pred;
                if (dc < -1024) dc += 2049;
                if (dc > 1024) dc -= 2049;
                here[49] = (int16_t)dc;
            }
        }
        // 5. publish the neighbour summary
        ns_here->nz = nz;
        int dcq = here[49] * (int)q[0];
        for (int i = 0; i < 8; ++i) {
            ns_here->horiz[i] = (int16_t)(dcq + pix[56 + i] + 1024 + half16(pix[56 + i] - pix[48 + i]));
            ns_here->vert[i] = (int16_t)(dcq + pix[i * 8 + 7] + 1024 + half16(pix[i * 8 + 7] - pix[i * 8 + 6]));
        }
        return 0;
    }

    // Runs the whole segment on the calling lane. ns: this segment's NSum area (zeroed).
    LEP_DEV int run(const ImageDev* image, const SegDev& seg, uint32_t* model_words, NSum* ns) {
        img = image; model = model_words; nbins = 0;
        bool top[3] = {true, true, true};
        for (uint32_t idx = 0;; ++idx) {
            RowSpec r = row_spec(img, idx);
            if (r.done) break;
            if (r.luma_y >= seg.y1 && !seg.is_last) break;
            if (r.skip) continue;
            if (r.luma_y < seg.y0) continue;
            comp = r.component; ci = comp ? 1 : 0; q = img->q[comp];
            const int w = img->width[comp], yb = r.curr_y;
            int16_t* row = img->blocks[comp] + (int64_t)yb * w * 64;
            const int16_t* arow = top[comp] ? nullptr : row - (int64_t)w * 64;
            NSum* nrow = ns + img->ns_offset[comp] + (yb & 1) * w;
            const NSum* narow = ns + img->ns_offset[comp] + ((yb & 1) ^ 1) * w;
            top[comp] = false;
            for (int x = 0; x < w; ++x) {
                int16_t* here = row + (int64_t)x * 64;
                const int16_t* l = x ? here - 64 : nullptr;
                const int16_t* a = arow ? arow + (int64_t)x * 64 : nullptr;
                const int16_t* al = (x && arow) ? a - 64 : nullptr;
                int rc = code_block(here, l, a, al, &nrow[x], x ? &nrow[x - 1] : nullptr, arow ? &narow[x] : nullptr);
                if (rc) return rc;
                if (x + 1 < w && yb * w + x + 1 >= img->coded_blocks[comp]) break;
            }
        }
        return 0;
    }
};

}  // namespace lepdev
