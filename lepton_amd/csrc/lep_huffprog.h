// lep_huffprog.h -- PROGRESSIVE JPEG scans re-encoded from a decoded coefficient frame ON THE GPU (BASELINE.json configs[4],
// decode direction): what the reference's recode_jpeg does on the CPU for files that are not one sequential scan
// (src/lepton/jpgcoder.cc:3309-3716; encode_dc_prg_fs / _sa, encode_ac_prg_fs / _sa, encode_eobrun, encode_crbits :4991-5400).
//
// Every scan of a progressive file is a function of the FINISHED frame alone -- a first-stage scan codes coefficient >> Al, a
// refinement scan codes bit Al, and which coefficients a refinement scan must treat as "already non-zero" is |coefficient|
// >> (Al + 1) != 0 -- so, unlike decoding, the scans of an image are independent of each other: one wavefront per (image,
// scan), all of them in one launch (a 4K 4:2:0 file written by libjpeg has 10 scans; 256 such files are 2560 wavefronts).
// Inside a scan the order is the file's: restart intervals, end-of-band runs and the correction bits they hold back tie
// consecutive blocks together.  Lane mapping:
//   DC scans      lane = one of 64 consecutive blocks in scan order: difference to the previous block of the same component
//                 (a lane shuffle; across batches a per-component carry), code + magnitude bits, wave prefix sum -> bit offsets
//   AC scans      lane = coefficient in zig-zag order (like lep_huff.h): ballots give zero runs / last non-zero; a first-stage
//                 block is one prefix sum of per-lane fields; a refinement block orders its fields by "next code at or after
//                 me" (codes first, then the correction bits of the already-non-zero coefficients passed since the code before)
//   correction bits behind an end-of-band run wait in a per-scan scratch area in HBM until the run is written
// Bytes leave through lep_huff.h's LDS bit buffer (FF00 stuffing, RSTn markers) into the scan's slot of the output arena; the
// host glues header pieces and scans together (jpeg_progressive.cc: recode_progressive_finish).
// Written on the SPMD layer of lep_wave.h: tests/emu runs it on the CPU against the host re-coder, byte for byte.
#pragma once
#include "lep_huff.h"

namespace lephuff {

struct ProgImage {          // one image, device-visible
    int32_t ncomp, mcuh, mcuv, mcuc;
    int32_t rsti, padbit;
    int32_t hs[4], vs[4], bch[4], bcv[4], nch[4], ncv[4], mbs[4];
    const int16_t* blocks[4];
};

struct ProgScan {
    int32_t image;
    int32_t cmpc, cmp[4];       // components of the scan, in scan order
    int32_t from, to, sah, sal; // spectral band, successive approximation
    int32_t max_eobrun;         // longest end-of-band run the scan's AC table can code
    int32_t tbl[4];             // per scan component: which of code[0..1] it uses (DC scans); AC scans use code[0]
    int32_t rsti;               // this scan's restart interval (DRI may change from scan to scan); < 0: the image's
    uint64_t out_off;           // into the output arena
    uint32_t out_cap;
    uint32_t corr_off;          // this scan's scratch area for held-back correction bits (dwords into the scratch arena)
    uint32_t corr_cap;          // dwords
    uint32_t pad;               // caller: bytes all scans of the image produce together at most (lep_huffprog_scan.file_bound); on the device:
                                // which kernel owns the scan (lep_huffprog_simt.h kProgScanSimt)
    uint32_t code[4][256];      // length << 16 | code; [2..3]: scans of sequential frames only (their AC tables 0 / 1)
};

// SEQUENTIAL frames coded in several scans (the reference's recode_jpeg codes them with encode_block_seq under the general scan walk,
// jpgcoder.cc:3461-3486, 3560-3580; format flag 'X' like progressive files): a scan with from 0 / to 63 -- which no progressive scan has --
// with code[0..1] = DC tables 0 / 1, code[2..3] = AC tables 0 / 1 and tbl[i] = DC table | AC table << 8 of scan component i.  They are written by the sequential scan encoders
// (lep_huff_simt.h, lep_huff.h) as an image of their own with one segment: the scan's components only, and for a scan of one component
// (never interleaved: MCU = one block, the frame's padding blocks stepped over) that component's nch x ncv blocks as MCUs, as
// recode_prepare plans a one-component file.
#if LEP_ON_GPU
__host__ __device__ __forceinline__
#else
inline
#endif
int prog_scan_rsti(const ProgImage& im, const ProgScan& sc) { return sc.rsti >= 0 ? sc.rsti : im.rsti; }   // (DRI may stand in front of any scan)

constexpr uint32_t kProgScanSeq = 2;        // ProgScan::pad on the device: the sequential scan encoders own the scan
inline bool prog_is_sequential(const ProgScan& s) { return s.from == 0 && s.to == 63; }
inline void sequential_scan_segment(const ProgImage& im, const ProgScan& sc, int32_t image_index, HuffImage* hi, HuffSegment* hs) {
    memset(hi, 0, sizeof *hi);
    memset(hs, 0, sizeof *hs);
    hi->ncomp = sc.cmpc; hi->mcuh = im.mcuh; hi->mcuv = im.mcuv; hi->mcuc = im.mcuc; hi->rsti = prog_scan_rsti(im, sc); hi->padbit = im.padbit;
    hi->rst_limit = 0xffffffffu;
    hi->interleaved = 1;
    for (int c = 0; c < 4; ++c) { hi->hs[c] = im.hs[c]; hi->vs[c] = im.vs[c]; hi->bch[c] = im.bch[c]; hi->blocks[c] = im.blocks[c]; }
    for (int i = 0; i < 4; ++i) hi->scan_cmp[i] = i < sc.cmpc ? (sc.cmp[i] & 3) : 0;
    if (sc.cmpc == 1) {
        const int c = sc.cmp[0] & 3;
        hi->mcuh = im.nch[c]; hi->mcuv = im.ncv[c]; hi->mcuc = im.nch[c] * im.ncv[c];
        hi->hs[c] = 1; hi->vs[c] = 1;
    }
    memcpy(hi->code, sc.code, sizeof hi->code);
    for (int i = 0; i < sc.cmpc && i < 4; ++i) { const int c = sc.cmp[i] & 3; hi->dc_tbl[c] = sc.tbl[i] & 1; hi->ac_tbl[c] = (sc.tbl[i] >> 8) & 1; }
    hs->image = image_index; hs->mcu_row0 = 0; hs->mcu_row1 = hi->mcuv;
    hs->out_off = sc.out_off; hs->out_cap = sc.out_cap;
}

struct ProgShared {
    HuffShared h;
    int32_t vals[64];
    uint32_t cb[4];             // staging of one block's held-back correction bits (MSB first)
};

static WDEV int fdiv2(int v, int p) { return v < 0 ? -((-v) >> p) : (v >> p); }
static WDEV uint32_t envli(int s, int v) { return (uint32_t)((v > 0) ? v : (v - 1) + (1 << s)) & ((1u << s) - 1u); }

struct ProgWave : HuffWave {
    const ProgImage* pim;
    const ProgScan* sc;
    ProgShared* ps;
    uint32_t* corr;        // held-back correction bits of this scan (HBM), MSB first
    uint32_t corr_bits;    // how many are waiting
    uint32_t corr_cap_bits;
    bool corr_overflow;
    uint32_t eobrun;

    // ---- the scan's block order ---------------------------------------------------------------------------------------
    // non-interleaved scans walk the component's nch x ncv blocks (next_mcuposn, jpgcoder.cc:5432-5456); interleaved (DC) scans
    // walk MCUs, inside an MCU the scan's components in order, each hs x vs blocks (next_mcupos :5402-5430)
    WDEV void locate(uint32_t idx, int& cmp, int& dpos, int& slot) const {
        if (sc->cmpc == 1) {
            cmp = sc->cmp[0]; slot = 0;
            const uint32_t nch = (uint32_t)pim->nch[cmp];
            dpos = (int)((idx / nch) * (uint32_t)pim->bch[cmp] + idx % nch);
            return;
        }
        uint32_t P = 0;
        for (int i = 0; i < sc->cmpc; ++i) P += (uint32_t)pim->mbs[sc->cmp[i]];
        const uint32_t m = idx / P;
        uint32_t q = idx % P;
        int i = 0;
        while (i + 1 < sc->cmpc && q >= (uint32_t)pim->mbs[sc->cmp[i]]) { q -= (uint32_t)pim->mbs[sc->cmp[i]]; ++i; }
        cmp = sc->cmp[i]; slot = i;
        const uint32_t hs = (uint32_t)pim->hs[cmp], vs = (uint32_t)pim->vs[cmp], mh = (uint32_t)pim->mcuh;
        if (vs > 1) dpos = (int)(((m / mh) * vs + q / hs) * (uint32_t)pim->bch[cmp] + (m % mh) * hs + q % hs);
        else if (hs > 1) dpos = (int)(m * (uint32_t)pim->mbs[cmp] + q);
        else dpos = (int)m;
    }
    WDEV uint32_t blocks_per_unit() const {   // blocks per restart-interval unit (an MCU)
        if (sc->cmpc == 1) return 1;
        uint32_t P = 0;
        for (int i = 0; i < sc->cmpc; ++i) P += (uint32_t)pim->mbs[sc->cmp[i]];
        return P;
    }
    WDEV uint32_t total_units() const {
        return sc->cmpc == 1 ? (uint32_t)pim->nch[sc->cmp[0]] * (uint32_t)pim->ncv[sc->cmp[0]] : (uint32_t)pim->mcuc;
    }

    // ---- held-back correction bits --------------------------------------------------------------------------------------
    WDEV void corr_append(uint32_t hi, uint32_t lo, int n) {   // n <= 64 bits, left-aligned in hi:lo
        if (!n) return;
        if (corr_bits + (uint32_t)n > corr_cap_bits) { corr_overflow = true; return; }
        LANES(l) if (l == 0) {
            const uint32_t w = corr_bits >> 5, sh = corr_bits & 31;
            const uint32_t keep = sh ? corr[w] & ~(0xffffffffu >> sh) : 0u;
            // 96-bit window: [keep | hi:lo >> sh]
            const uint64_t v = ((uint64_t)hi << 32) | lo;
            corr[w] = keep | (uint32_t)(sh ? (v >> 32) >> sh : (v >> 32));
            const uint64_t rest = sh ? v << (32 - sh) : v << 32;   // bits that did not fit the first word, left-aligned
            if ((int)(32 - sh) < n) corr[w + 1] = (uint32_t)(rest >> 32);
            if ((int)(64 - sh) < n) corr[w + 2] = (uint32_t)rest;
        }
        LSYNC();
        corr_bits += (uint32_t)n;
    }
    WDEV void corr_flush() {   // all of them into the stream, in order
        uint32_t done = 0;
        while (done < corr_bits) {
            const uint32_t chunk = corr_bits - done < 1024u ? corr_bits - done : 1024u;   // 32 words at most per round
            LANES(l) {
                const uint32_t b0 = (uint32_t)l * 32u;
                if (b0 < chunk) {
                    const uint32_t nb = chunk - b0 < 32u ? chunk - b0 : 32u;
                    const uint32_t w = corr[(done >> 5) + (uint32_t)l];   // done is a multiple of 1024
                    put_field(w >> (32u - nb), (int)nb, pend + (int)b0);
                }
            }
            LSYNC();
            flush_bytes(pend + (int)chunk);
            done += chunk;
        }
        corr_bits = 0;
    }

    // ---- end-of-band runs (encode_eobrun, jpgcoder.cc:5337-5368) ---------------------------------------------------------------
    WDEV void emit_eobrun() {
        if (!eobrun) return;
        const uint32_t* ac = ps->h.code[0];
        int p = pend;
        while (eobrun > (uint32_t)sc->max_eobrun) {
            const uint32_t e = ac[0xE0];
            LANES(l) if (l == 0) { put_field(e & 0xffffu, (int)(e >> 16), p); put_field(16383u, 14, p + (int)(e >> 16)); }
            LSYNC();
            p += (int)(e >> 16) + 14;
            eobrun -= (uint32_t)sc->max_eobrun;
            flush_bytes(p);
            p = pend;
        }
        int s = bitlen(eobrun & 0xffffu);
        if (s) --s;
        const uint32_t e = ac[(s << 4) & 255];
        LANES(l) if (l == 0) { put_field(e & 0xffffu, (int)(e >> 16), p); if (s) put_field(eobrun - (1u << s), s, p + (int)(e >> 16)); }
        LSYNC();
        flush_bytes(p + (int)(e >> 16) + s);
        eobrun = 0;
    }

    // ---- DC scans: 64 blocks per round -------------------------------------------------------------------------------------------
    WDEV void dc_first_batch(uint32_t base, uint32_t n, int* lastdc_c) {
        const uint32_t P = blocks_per_unit();
        LV(int, cmpv); LV(int, subv); LV(int, slotv); LV(int, val);
        LANES(l) {
            int cmp = 0, dpos = 0, slot = 0, v = 0, sub = 0;
            if ((uint32_t)l < n) {
                locate(base + (uint32_t)l, cmp, dpos, slot);
                v = (int)pim->blocks[cmp][(int64_t)dpos * 64 + 49] >> sc->sal;
                if (sc->cmpc > 1) {
                    uint32_t q = (base + (uint32_t)l) % P;
                    for (int i = 0; i < slot; ++i) q -= (uint32_t)pim->mbs[sc->cmp[i]];
                    sub = (int)q;
                } else sub = (uint32_t)l == 0 ? 0 : 1;
            }
            L(cmpv) = cmp; L(subv) = sub; L(slotv) = slot; L(val) = v;
            ps->vals[l] = v;
        }
        LSYNC();
        LV(int, total); LV(int, off); LV(uint32_t, fb);
        LANES(l) {
            int n_ = 0;
            uint32_t bits = 0;
            if ((uint32_t)l < n) {
                const int cmp = L(cmpv);
                // previous block of the same component: the lane before inside an MCU's group, else that component's last block of
                // the MCU before (interleaved), else the carry from the round before
                int src = sc->cmpc > 1 ? (L(subv) > 0 ? l - 1 : l - (int)P + pim->mbs[cmp] - 1) : l - 1;
                const int prev = src >= 0 ? ps->vals[src] : lastdc_c[L(slotv)];
                const int d = (int16_t)(L(val) - prev);
                const int s = bitlen((uint32_t)(d > 0 ? d : -d) & 0xffffu);
                const uint32_t e = ps->h.code[sc->tbl[L(slotv)] & 1][s & 255];
                n_ = (int)(e >> 16) + s;
                bits = ((e & 0xffffu) << s) | envli(s, d);
            }
            L(total) = n_; L(fb) = bits;
        }
        const int B = lepwave::wave_excl_scan(total, off);
        LANES(l) if (L(total)) put_field(L(fb), L(total), pend + L(off));
        LSYNC();
        flush_bytes(pend + B);
        // carries: each scan component's last value of this round
        for (int i = 0; i < sc->cmpc; ++i) {
            LV(int, mine);
            LANES(l) L(mine) = (uint32_t)l < n && L(slotv) == i;
            const uint64_t m = lepwave::wave_ballot(mine);
            if (m) lastdc_c[i] = (int)lepwave::wave_read((const uint32_t*)val, 63 - __builtin_clzll(m));
        }
    }
    WDEV void dc_refine_batch(uint32_t base, uint32_t n) {
        LV(int, bitv);
        LANES(l) {
            int b = 0;
            if ((uint32_t)l < n) {
                int cmp, dpos, slot;
                locate(base + (uint32_t)l, cmp, dpos, slot);
                b = ((int)pim->blocks[cmp][(int64_t)dpos * 64 + 49] >> sc->sal) & 1;
            }
            L(bitv) = b;
        }
        const uint64_t m = lepwave::wave_ballot(bitv);
        // lane l's bit is stream bit l: reverse so that lane 0 is the most significant of the field
        uint32_t hi = 0, lo = 0;
        for (int i = 0; i < 32; ++i) { hi |= (uint32_t)((m >> i) & 1) << (31 - i); lo |= (uint32_t)((m >> (32 + i)) & 1) << (31 - i); }
        LANES(l) if (l == 0) {
            const int n0 = n < 32u ? (int)n : 32;
            put_field(hi >> (32 - n0), n0, pend);
            if (n > 32u) put_field(lo >> (64 - n), (int)n - 32, pend + 32);
        }
        LSYNC();
        flush_bytes(pend + (int)n);
    }

    // ---- AC first stage: one block (encode_ac_prg_fs, jpgcoder.cc:5230-5290) --------------------------------------------------------
    WDEV void ac_first_block(int cmp, int dpos) {
        const int16_t* blk = pim->blocks[cmp] + (int64_t)dpos * 64;
        const int from = sc->from, to = sc->to, sal = sc->sal;
        const uint32_t* ac = ps->h.code[0];
        LV(int, tv); LV(int, nzf);
        LANES(l) {
            int t = 0;
            if (l >= from && l <= to) t = fdiv2((int)blk[ps->h.z2a[l]], sal);
            L(tv) = t; L(nzf) = t != 0;
        }
        const uint64_t m = lepwave::wave_ballot(nzf);
        if (!m) {
            ++eobrun;
            if (eobrun == (uint32_t)sc->max_eobrun) emit_eobrun();
            return;
        }
        emit_eobrun();
        const int end = 63 - __builtin_clzll(m);
        const uint32_t zrl = ac[0xF0];
        const int zrl_len = (int)(zrl >> 16);
        LV(int, total); LV(int, off); LV(int, nn); LV(uint32_t, fb); LV(int, kk);
        LANES(l) {
            int n = 0, k = 0;
            uint32_t bits = 0;
            const int t = L(tv);
            if (t != 0) {
                const uint64_t pm = m & ((1ull << l) - 1);
                const int prev = pm ? 63 - __builtin_clzll(pm) : from - 1;
                const int run = l - prev - 1;
                k = run >> 4;
                const int s = bitlen((uint32_t)(t > 0 ? t : -t) & 0xffffu);
                const uint32_t e = ac[(((run & 15) << 4) + s) & 255];
                n = (int)(e >> 16) + s; bits = ((e & 0xffffu) << s) | envli(s, t);
            }
            L(nn) = n; L(fb) = bits; L(kk) = k; L(total) = n + k * zrl_len;
        }
        const int B = lepwave::wave_excl_scan(total, off);
        LANES(l) if (L(total)) {
            int p = pend + L(off);
            for (int i = 0; i < L(kk); ++i) { put_field(zrl & 0xffffu, zrl_len, p); p += zrl_len; }
            put_field(L(fb), L(nn), p);
        }
        LSYNC();
        flush_bytes(pend + B);
        if (end < to) {
            ++eobrun;
            if (eobrun == (uint32_t)sc->max_eobrun) emit_eobrun();
        }
    }

    // ---- AC refinement: one block (encode_ac_prg_sa + encode_crbits, jpgcoder.cc:5292-5400) -------------------------------------
    // classes per band position: Z zero, N newly non-zero (+-1), O already non-zero (one correction bit).  Up to the last N
    // ("eob") the stream is: for every code point (an N, or the 16th / 32nd / 48th Z since the N before) its code [+ sign],
    // then the correction bits of the O positions passed since the code point before.  O positions from eob on are held back.
    WDEV void ac_refine_block(int cmp, int dpos) {
        const int16_t* blk = pim->blocks[cmp] + (int64_t)dpos * 64;
        const int from = sc->from, to = sc->to, sal = sc->sal;
        const uint32_t* ac = ps->h.code[0];
        LV(int, tv); LV(int, isn); LV(int, iso); LV(int, isz);
        LANES(l) {
            int t = 0;
            const bool in = l >= from && l <= to;
            if (in) t = fdiv2((int)blk[ps->h.z2a[l]], sal);
            L(tv) = t; L(isn) = in && (t == 1 || t == -1); L(iso) = in && (t > 1 || t < -1); L(isz) = in && t == 0;
        }
        const uint64_t Nm = lepwave::wave_ballot(isn), Om = lepwave::wave_ballot(iso), Zm = lepwave::wave_ballot(isz);
        const int eob = Nm ? 64 - __builtin_clzll(Nm) : from;   // 1 + last N
        if (eob > from && eobrun > 0) { emit_eobrun(); corr_flush(); }
        const uint64_t below_eob = eob >= 64 ? ~0ull : ((1ull << eob) - 1);
        // code points among the zeros: every 16th zero since the N before (only in front of eob)
        LV(int, zrlf); LV(int, zc_);
        LANES(l) {
            int zc = 0, f = 0;
            if (l < eob && (L(isz) || L(isn))) {
                const uint64_t nb = Nm & ((1ull << l) - 1);
                const int lastn = nb ? 63 - __builtin_clzll(nb) : -1;
                const uint64_t range = ((l >= 63 ? ~0ull : ((1ull << (l + 1)) - 1))) & ~((lastn >= 63) ? ~0ull : ((1ull << (lastn + 1)) - 1));
                zc = lepwave::popc64(Zm & range);   // zeros since the N before, this position included
                f = L(isz) && zc > 0 && (zc & 15) == 0;
            }
            L(zrlf) = f; L(zc_) = zc;
        }
        const uint64_t Em = (lepwave::wave_ballot(zrlf) | Nm) & below_eob;   // code points
        const uint32_t zrl = ac[0xF0];
        LV(int, clen); LV(uint32_t, cbits); LV(int, coff);
        LANES(l) {
            int n = 0;
            uint32_t bits = 0;
            if (l < eob) {
                if (L(zrlf)) { n = (int)(zrl >> 16); bits = zrl & 0xffffu; }
                else if (L(isn)) {
                    const uint32_t e = ac[(((L(zc_) & 15) << 4) + 1) & 255];
                    n = (int)(e >> 16) + 1; bits = ((e & 0xffffu) << 1) | envli(1, L(tv));
                }
            }
            L(clen) = n; L(cbits) = bits;
        }
        const int CB = lepwave::wave_excl_scan(clen, coff);   // code bits in front of each lane
        const int nO_front = lepwave::popc64(Om & below_eob);
        LANES(l) {
            if (l < eob) {
                const uint64_t lowmask = (1ull << l) - 1;
                if (L(clen)) {
                    // in front of this code: all earlier codes, and the correction bits flushed by them = the O positions in front
                    // of the code point before this one
                    const uint64_t eb = Em & lowmask;
                    const int preve = eb ? 63 - __builtin_clzll(eb) : -1;
                    const int nO = preve >= 0 ? lepwave::popc64(Om & ((1ull << preve) - 1)) : 0;
                    put_field(L(cbits), L(clen), pend + L(coff) + nO);
                }
            }
        }
        // the O lanes need "code bits up to and including their next code point": coff[nexte] + clen[nexte]; lanes cannot index
        // other lanes' registers, so publish the inclusive prefix through LDS
        LANES(l) ps->vals[l] = L(coff) + L(clen);
        LSYNC();
        LANES(l) {
            if (l < eob && L(iso)) {
                const uint64_t lowmask = (1ull << l) - 1;
                const int nexte = __builtin_ctzll(Em & ~lowmask);
                const int nO = lepwave::popc64(Om & lowmask);
                put_field((uint32_t)(L(tv) & 1), 1, pend + ps->vals[nexte] + nO);
            }
        }
        LSYNC();
        flush_bytes(pend + CB + nO_front);
        // held back: the O positions from eob on, in order
        const uint64_t tail = Om & ~below_eob;
        if (tail) {
            LANES(l) if (l < 4) ps->cb[l] = 0;
            LSYNC();
            LANES(l) if ((tail >> l) & 1) {
                const int k = lepwave::popc64(tail & ((1ull << l) - 1));
                if (L(tv) & 1) lds_or(&ps->cb[k >> 5], 0x80000000u >> (k & 31));
            }
            LSYNC();
            corr_append(ps->cb[0], ps->cb[1], lepwave::popc64(tail));
        }
        if (eob <= to) {
            ++eobrun;
            if (eobrun == (uint32_t)sc->max_eobrun) { emit_eobrun(); corr_flush(); }
        }
    }

    // ---- one scan ---------------------------------------------------------------------------------------------------------------
    // returns the bytes produced (clipped to the slot); bit 31 set: the held-back correction bits outgrew their scratch area
    WDEV uint32_t run_scan(const ProgImage* image, const ProgScan* scan, ProgShared* shared, uint8_t* arena, uint32_t* corr_arena) {
        pim = image; sc = scan; ps = shared; sh = &shared->h; img = nullptr;
        out = arena + scan->out_off; cap = scan->out_cap; written = 0; pend = 0;
        corr = corr_arena + scan->corr_off; corr_bits = 0; corr_cap_bits = scan->corr_cap * 32u; corr_overflow = false;
        eobrun = 0;
        LANES(l) {
            for (int i = l; i < 512; i += 64) (&sh->code[0][0])[i] = (&scan->code[0][0])[i];
            sh->z2a[l] = kZ2A[l];
            sh->bits[l] = 0u;
            if (l < 8) sh->bits[64 + l] = 0u;
        }
        LSYNC();
        const uint32_t units = total_units(), P = blocks_per_unit();
        const uint32_t rsti = (uint32_t)prog_scan_rsti(*pim, *scan);
        const bool dc = scan->to == 0;
        uint32_t unit = 0, cum_rst = 0;
        while (unit < units) {
            const uint32_t unit_end = rsti ? (unit + rsti < units ? unit + rsti : units) : units;   // one restart interval
            if (dc) {
                int lastdc_c[4] = {0, 0, 0, 0};
                for (uint32_t b = unit * P; b < unit_end * P; b += 64) {
                    const uint32_t n = unit_end * P - b < 64u ? unit_end * P - b : 64u;
                    if (scan->sah == 0) dc_first_batch(b, n, lastdc_c); else dc_refine_batch(b, n);
                }
            } else {
                const int cmp = scan->cmp[0];
                const uint32_t nch = (uint32_t)pim->nch[cmp], bch = (uint32_t)pim->bch[cmp];
                for (uint32_t u = unit; u < unit_end; ++u) {
                    const int dpos = (int)((u / nch) * bch + u % nch);
                    if (scan->sah == 0) ac_first_block(cmp, dpos); else ac_refine_block(cmp, dpos);
                }
                emit_eobrun();
                if (scan->sah != 0) corr_flush();
            }
            pad_byte(pim->padbit);
            unit = unit_end;
            if (unit < units) { raw_bytes2(0xFF, (uint8_t)(0xD0 + (cum_rst & 7u))); ++cum_rst; }
        }
        const uint32_t n = written < cap ? written : cap;
        return n | (corr_overflow || written > cap ? 0x80000000u : 0u);
    }
};

}  // namespace lephuff
