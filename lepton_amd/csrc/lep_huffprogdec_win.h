// lep_huffprogdec_win.h -- PROGRESSIVE JPEG scans decoded into the coefficient frame with a WINDOW OF SPECULATIVE CODES
// (round 6; BASELINE.json configs[4], encode direction).  Same frames, records and refusals as lep_huffprogdec.h, which restates
// the progressive branches of decode_jpeg's scan loop (src/lepton/jpgcoder.cc:2975-3260) with decode_dc_prg_fs / _sa,
// decode_ac_prg_fs / _sa, decode_eobrun_sa, skip_eobrun (:4968-5335, :5462-5500).
//
// A progressive scan is one dependent chain -- where a code starts depends on every code in front of it, and what a refinement
// code MEANS depends on which coefficients of its block are already non-zero, so a wavefront started in the middle of a scan
// cannot even parse (no self-synchronisation to lean on as in lep_huffdec_simt.h).  lep_huffprogdec.h runs that chain as uniform
// vector code, ~200 dependent instructions per code: window -> table in LDS -> shift -> refill ... and the last luma bit plane
// of a 4K file (1.4 M codes) takes 0.6 s on a chip that is otherwise empty.  Here the 64 lanes do the table look-ups AHEAD of
// the chain instead of on it:
//
//   * STAGE: lane l decodes the Huffman code that would start at bit base + l of the scan -- its 32-bit window from a ring of
//     scan bytes in LDS, one first-level look-up, the sixteen bits behind the code kept with it (sign / magnitude / run-length
//     bits).  63 of the 64 answers are for codes that do not exist; nobody waits for them.
//   * CHAIN: the wavefront hops from code to code on the SCALAR unit: v_readlane of lane `off`'s answer, a handful of scalar
//     instructions for what the code means, off += bits.  No memory, no LDS, no vector arithmetic on the chain; a new stage
//     every 64 bits.  In a refinement block the position a code's coefficient takes -- the (run + 1)-th zero of the band -- is
//     one compare of a per-lane zero rank against a scalar and a find-first-set; the correction bits of the non-zero positions
//     it passes are not read by the chain at all, only skipped: their count is a difference of two per-lane ranks, and when the
//     block is done every non-zero lane fetches its own bit from the ring (its address: the mark the next code point left, by
//     a suffix minimum over the lanes, plus its rank).
//   * What a code places goes into lane registers under the compare's own mask; one masked lane-parallel store per block.
//   * Everything the chain reads about the scan lives in registers (WinScan): the wavefront-scope fences between the lane-parallel
//     regions make the compiler reload whatever it reads through a pointer, and a scalar load on the chain is 300 cycles.  Block
//     positions are walked with counters, not divisions.
//
// One wavefront per (image, scan) as before, same pipelining between the scans of a file (ProgDeps / progress), same refusals
// (status -> the host parser).  Restart intervals keep lep_huffprogdec.h.
// SPMD layer of lep_wave.h: tests/emu steps it on the CPU against the host parser and against lep_huffprogdec.h.
#pragma once
#include "lep_huffprogdec.h"

namespace lephuff {

constexpr uint32_t kWinRing = 1024;     // dwords of scan data in LDS
constexpr uint32_t kWinChunk = 256;     // dwords per refill (64 lanes x 16 bytes)
constexpr uint32_t kWinAhead = 300;     // dwords the ring stays in front of the window
constexpr uint32_t kWinNoMark = 0x7fffffffu;

struct ProgWinShared {
    uint32_t ring[kWinRing];            // dword d of the scan (big-endian value) at ring[d & 1023]
    uint16_t lut[3][512];               // [0..1] DC tables, [2] the scan's AC table: first 9 bits -> length << 8 | symbol; 0 = longer code
    int32_t maxcode[3][8];
    int32_t valoff[3][8];
    uint8_t longsym[3][256];
    uint8_t z2a[64];
};

struct WinScan {                        // the scan's constants, in registers
    int from, to, sal, sah, max_eobrun, want_rows, tbl0;
    uint32_t plus, minus;               // +-1 << sal as the 16 bits a coefficient is stored as
    int cmp, bch, nch, ncv, vs;         // one-component scans: the component's geometry
    int16_t* blocks;                    // ... and its frame
    const uint8_t* scan;
    uint32_t scan_len, limit;
};

struct ProgWinWave : ProgDecWave {
    ProgWinShared* ws;
    WinScan k;
    uint32_t base = 0;                  // bit position of lane 0's window
    uint32_t off = 0;                   // bits consumed behind it: the next code is lane `off`'s
    uint32_t ring_hi = 0;               // dwords [ring_hi - kWinRing, ring_hi) are in the ring
    uint32_t pf_dword = 0;              // first dword of the chunk waiting in wpf
    bool two_tables = false;            // DC scans: both DC tables are looked up
    LV(uint32_t, win);                  // the 32 bits from base + l on
    LV(uint32_t, pre);                  // length << 8 | symbol | the 16 bits behind the code << 16 (table 0 / the AC table)
    LV(uint32_t, pre1);                 // ... DC table 1
    LV(uint32_t, wpf0); LV(uint32_t, wpf1); LV(uint32_t, wpf2); LV(uint32_t, wpf3);
    LV(uint32_t, zz);                   // the lane's zig-zag position in aligned order

    WDEV uint32_t pos() const { return base + off; }

    // the next chunk of the scan, sixteen bytes per lane, as it lies in memory (swapped to big-endian values when it is committed).
    // The scan is 16-byte aligned and followed by sixteen zero bytes at least: a load that starts at or in front of scan_len is
    // inside that; loads behind it read the last such place instead and are zeroed at the commit.
    WDEV void request(uint32_t first) {
        pf_dword = first;
        const uint32_t last = k.scan_len & ~15u;
        LANES(l) {
            uint32_t b = (first + 4u * (uint32_t)l) * 4u;
            b = b < last ? b : last;
            const uint32_t* p = reinterpret_cast<const uint32_t*>(k.scan + b);
            L(wpf0) = lepwave::gld(p); L(wpf1) = lepwave::gld(p + 1); L(wpf2) = lepwave::gld(p + 2); L(wpf3) = lepwave::gld(p + 3);
        }
    }
    WDEV void commit() {   // the chunk that was requested a refill ago goes into the ring, the next one is requested
        const uint32_t last = k.scan_len & ~15u;
        LANES(l) {
            const uint32_t d = pf_dword + 4u * (uint32_t)l, s = d & (kWinRing - 1);
            const bool in = d * 4u <= last;
            ws->ring[s] = in ? __builtin_bswap32(L(wpf0)) : 0u; ws->ring[s + 1] = in ? __builtin_bswap32(L(wpf1)) : 0u;
            ws->ring[s + 2] = in ? __builtin_bswap32(L(wpf2)) : 0u; ws->ring[s + 3] = in ? __builtin_bswap32(L(wpf3)) : 0u;
        }
        ring_hi = pf_dword + kWinChunk;
        request(ring_hi);
        LSYNC();
    }
    // a lane's answer: symbol | length << 8 | the 16 bits behind the code << 16
    int fast_kind = 0;                  // 0 DC, 1 AC first stage, 2 AC refinement (what stage() puts into the short path's words)
    WDEV uint32_t pack(uint32_t e, uint32_t w) const {
        const uint32_t len = e >> 8;
        return (e & 0x1fffu) | (len ? ((w << len) >> 16) << 16 : 0u);
    }
    // lane l looks at the code that would start at bit base + off + l; off becomes 0
    WDEV void stage() {
        base += off; off = 0;
        while (ring_hi < (base >> 5) + kWinAhead) commit();
        LANES(l) {
            const uint32_t p = base + (uint32_t)l, d = p >> 5, sft = p & 31u;
            const uint32_t w0 = ws->ring[d & (kWinRing - 1)], w1 = ws->ring[(d + 1) & (kWinRing - 1)];
            const uint32_t w = (uint32_t)((((uint64_t)w0 << 32) | w1) >> (32u - sft));
            L(win) = w;
            if (two_tables) { L(pre) = pack(ws->lut[0][w >> 23], w); L(pre1) = pack(ws->lut[1][w >> 23], w); }
            else {
                const uint32_t e = ws->lut[2][w >> 23];
                L(pre) = pack(e, w);
                // what the chain's short path reads (`pre1` is free in AC scans): run | bits consumed << 8 | value placed << 16 for the codes it
                // takes -- refinement: a new +-1 (1 << sal) or a ZRL (value 0), first stage: a coefficient -- and a run of 255, which no
                // position can answer, for everything else
                const uint32_t len = e >> 8, sym = e & 255u, s = sym & 15u;
                uint32_t f = 0xffu;
                if (fast_kind == 2) {
                    if (len != 0u && (s == 1u || sym == 0xF0u)) f = (sym >> 4) | (len + s) << 8 | (s ? ((w << len) >> 31 ? k.plus : k.minus) : 0u) << 16;
                } else if (len != 0u && s != 0u) {
                    const uint32_t n = (w << len) >> (32u - s);
                    f = (sym >> 4) | (len + s) << 8 | ((uint32_t)(uint16_t)((uint16_t)devli(s, n) << k.sal)) << 16;
                }
                L(pre1) = f;
            }
        }
    }
    // a code of 9..16 bits at lane `off`: lane k tests length 9 + k (lep_huffdec.h symbol_long); -1 = not a code
    WDEV int long_code(int t, uint32_t w, uint32_t* len) {
        LV(int, ok);
        LV(uint32_t, cand);
        LANES(l) {
            const int kk = l & 7;
            const uint32_t code = w >> (23 - kk);
            L(ok) = l < 8 && (int)code <= ws->maxcode[t][kk];
            L(cand) = (uint32_t)(ws->valoff[t][kk] + (int)code);
        }
        const uint64_t m = lepwave::wave_ballot(ok);
        if (!m) return -1;
        const int kk = __builtin_ctzll(m);
        *len = 9u + (uint32_t)kk;
        return (int)ws->longsym[t][lepwave::wave_read(cand, kk) & 255u];
    }
    // the code at the chain's position, table t (0 / 1 DC, 2 AC): its length, symbol and the 16 bits behind it; false = no such code
    WDEV bool code_at(int t, uint32_t* len, uint32_t* sym, uint32_t* f16) {
        if (off >= 64u) stage();
        const uint32_t e = t == 1 ? lepwave::wave_read(pre1, (int)off) : lepwave::wave_read(pre, (int)off);
        uint32_t ln = (e >> 8) & 31u;
        if (ln == 0) {
            const uint32_t w = lepwave::wave_read(win, (int)off);
            const int r = long_code(t, w, &ln);
            if (r < 0) return false;
            *len = ln; *sym = (uint32_t)r; *f16 = (w << ln) >> 16;
            return true;
        }
        *len = ln; *sym = e & 255u; *f16 = e >> 16;
        return true;
    }
    // n <= 16 raw bits at the chain's position
    WDEV uint32_t bits_at(uint32_t n) {
        if (off >= 64u) stage();
        const uint32_t w = lepwave::wave_read(win, (int)off);
        off += n;
        return n ? w >> (32u - n) : 0u;
    }
    // the bit at absolute position p of the scan, from the ring (inside LANES; p within the ring)
    WDEV uint32_t ring_bit(uint32_t p) const { return (ws->ring[(p >> 5) & (kWinRing - 1)] >> (31u - (p & 31u))) & 1u; }

    // ---- AC first stage, one block (decode_ac_prg_fs): 0 ok, -1 irregular ------------------------------------------------------------
    WDEV int ac_first_win(int dpos) {
        const uint32_t from = (uint32_t)k.from, to = (uint32_t)k.to;
        const int sal = k.sal;
        if (eobrun > 0) { --eobrun; return 0; }   // inside a run: the band of this block is zero (the frame starts zeroed)
        uint32_t bpos = from, last_s = 1;
        int rc = 0;
        bool run_read = false;
        LV(uint32_t, nv);
        LANES(l) L(nv) = 0;
#pragma nounroll
        while (bpos <= to) {
            // the common codes without leaving the scalar unit: a coefficient behind a run of up to fifteen zeros (the lane's word
            // holds run, bits and value; a run of 255 -- everything else -- lands behind the band)
            if (off >= 64u) stage();
            const uint32_t f = lepwave::wave_read(pre1, (int)off);
            const uint32_t at = bpos + (f & 255u);
            if (at <= to) {
                LANES(l) if ((uint32_t)l == at) L(nv) = f >> 16;
                bpos = at + 1u;
                off += (f >> 8) & 63u;
                last_s = 1;
                continue;
            }
            uint32_t len, sym, f16;
            if (!code_at(2, &len, &sym, &f16)) { rc = -1; break; }
            const uint32_t r = sym >> 4, s = sym & 15u;
            if (r == 15u || s > 0u) {
                if (r + bpos > to) { rc = -1; break; }
                bpos += r;
                if (s > 0u) {
                    const uint32_t n = f16 >> (16u - s);
                    lepwave::wave_write(nv, (int)bpos, (uint32_t)(uint16_t)((uint16_t)devli(s, n) << sal));
                }
                ++bpos;
                off += len + s;
                last_s = s;
            } else {
                // end of band, and of 2^r + extra - 1 further blocks (a run behind a run the encoder had not filled up: host)
                const uint32_t extra = r ? f16 >> (16u - r) : 0u;
                off += len + r;
                if (last_s == 0u) { rc = -1; break; }
                eobrun = extra + (1u << r) - 1u;
                if (bpos == from && peobrun > 0 && peobrun < k.max_eobrun) { rc = -1; break; }
                peobrun = (int)eobrun + 1;
                run_read = true;
                break;
            }
        }
        if (!run_read) {
            if (!rc && last_s == 0u) rc = -1;         // the band ends in a coded zero
            peobrun = 0;
        }
        // (a value whose 16 bits come out zero is not stored: the frame starts zeroed)
        LV(int, chg);
        LANES(l) L(chg) = L(nv) != 0;
        if (lepwave::wave_ballot(chg)) {
            int16_t* dst = k.blocks + (int64_t)dpos * 64;
            LANES(l) if (L(chg)) lepwave::gst(dst + L(zz), (int16_t)L(nv));
        }
        return rc;
    }

    // ---- AC refinement, one block (decode_ac_prg_sa / decode_eobrun_sa) ---------------------------------------------------------------
    // The block's coefficients as the scans in front left them are ALWAYS the ones requested a block ago (request_block: by the
    // block in front, or by the walk where a block row begins), and the next block's are requested right behind -- never a select
    // between a register and a load of this iteration: the compiler then waits for every load in flight, the one just issued
    // included, and each block stands a whole trip to HBM (measured: 1.4 us of a block's 2.2).  For the same reason a block's
    // STORE is issued a block late (flush_store), in front of the next request: the wait for a request then only ever meets memory
    // operations that are a whole block old (the compiler cannot count a store under a condition and waits for all: 0.4 us).
    WDEV void request_block(int dpos) {
        const int16_t* src = k.blocks + (int64_t)dpos * 64;
        LANES(l) L(pf) = lepwave::gld(src + L(zz));
    }
    LV(uint32_t, held);                 // what the block before changed (lane = zig-zag position), not stored yet
    uint64_t held_mask = 0;
    int held_dpos = 0;
    WDEV void flush_store() {
#ifndef LEP_WIN_NOSTORE
        if (held_mask) {
            int16_t* dst = k.blocks + (int64_t)held_dpos * 64;
            LANES(l) if ((held_mask >> l) & 1ull) lepwave::gst(dst + L(zz), (int16_t)L(held));
        }
#endif
        held_mask = 0;
    }
    // next_in_row: the block behind this one is the next of the scan
    WDEV int ac_refine_win(int dpos, bool next_in_row) {
        const int from = k.from, to = k.to, sal = k.sal;
        const uint64_t band = (to >= 63 ? ~0ull : ((1ull << (to + 1)) - 1)) & ~((1ull << from) - 1);
        LV(int, cur); LV(int, nzf);
        LANES(l) L(cur) = (int)L(pf);
        lepwave::wave_select((uint32_t*)cur, ~band, 0u);
        LANES(l) L(nzf) = L(cur) != 0;
        flush_store();
#ifndef LEP_WIN_NOLOAD
        request_block(next_in_row ? dpos + 1 : dpos);   // (no next block in this row: the same one again, for nobody)
#endif
        const uint64_t nzm = lepwave::wave_ballot(nzf);
        const uint64_t zm = band & ~nzm;
        const uint32_t nztotal = (uint32_t)lepwave::popc64(nzm);
        // cmark: for a non-zero position, the bit its correction bit stands at, minus its rank among the non-zero positions
        LV(uint32_t, zrank); LV(uint32_t, nzrank); LV(uint32_t, cmark); LV(uint32_t, nv);
        LANES(l) {
            L(zrank) = (uint32_t)lepwave::mbcnt(zm, l);
            L(nzrank) = (uint32_t)lepwave::mbcnt(nzm, l);
            L(cmark) = kWinNoMark;
            L(nv) = 0;
        }
        lepwave::wave_select(zrank, ~zm, 0xffffu);
        // The chain's state.  acc = (bit position) - (non-zero positions in front of the next position): a code's correction bits start at
        // acc + its own bits, which is also the next acc -- one addition per code for both; the window offset follows from it.  p: the
        // position the last code took (from - 1: none yet); ahead: the lanes behind it.  zr: zero positions in front of the next one.
        uint32_t acc = base + off, zr = 0, last = 0x10000u;
        int p = from - 1;
        uint64_t ahead = ~0ull;
        int rc = 0;
        if (eobrun == 0) {
#pragma nounroll
            for (;;) {
                // the common codes without leaving the scalar unit: a new +-1 behind r zeros, or sixteen zeros (ZRL).  The (r + 1)-th zero
                // position behind p takes it: one compare of the lanes' zero ranks; the non-zero positions in between take correction
                // bits, which the chain only steps over (their count: the difference of two lanes' ranks) -- every lane still AHEAD takes
                // this code's mark for them, the lanes behind the NEXT code's start will be overwritten by that code's.
                if (off >= 64u) stage();
                const uint32_t f = lepwave::wave_read(pre1, (int)off);
                const uint32_t T = zr + (f & 255u);
                LV(int, hit);
                LANES(l) L(hit) = L(zrank) == T;
                const uint64_t m = lepwave::wave_ballot(hit);
                if (m) {
                    p = __builtin_ctzll(m);
                    const uint32_t nz2 = lepwave::wave_read(nzrank, p);
                    const uint32_t mark = acc + ((f >> 8) & 63u);
                    lepwave::wave_select(cmark, ahead, mark);
                    lepwave::wave_select(nv, m, f >> 16);
                    ahead = ~(m | (m - 1));
                    off = mark + nz2 - base;
                    acc = mark; zr = T + 1u; last = f;
                    if (p >= to) break;
                    continue;
                }
                // anything else: a code of more than nine bits, an end of band, or something a canonical encoder does not write
                uint32_t len, sym, f16;
                if (!code_at(2, &len, &sym, &f16)) { rc = -1; break; }
                const uint32_t r = sym >> 4, s = sym & 15u;
                if (r == 15u || s > 0u) {
                    if (s > 1u) { rc = -1; break; }
                    const uint32_t T2 = zr + r;
                    LANES(l) L(hit) = L(zrank) == T2;
                    const uint64_t m2 = lepwave::wave_ballot(hit);
                    if (!m2) { rc = -1; break; }                                 // the walk would leave the band
                    p = __builtin_ctzll(m2);
                    const uint32_t nz2 = lepwave::wave_read(nzrank, p);
                    const uint32_t mark = acc + len + s;
                    const uint32_t val = s ? ((f16 >> 15) ? k.plus : k.minus) : 0u;
                    lepwave::wave_select(cmark, ahead, mark);
                    lepwave::wave_select(nv, m2, val);
                    ahead = ~(m2 | (m2 - 1));
                    off = mark + nz2 - base;
                    acc = mark; zr = T2 + 1u; last = val << 16;
                    if (p >= to) break;
                } else {
                    const uint32_t extra = r ? f16 >> (16u - r) : 0u;
                    off += len + r; acc += len + r;
                    if ((last >> 16) == 0u) { rc = -1; break; }                  // ZRL in front of the end of band: not canonical
                    eobrun = extra + (1u << r);
                    if (p == from - 1 && peobrun > 0 && peobrun < k.max_eobrun - 1) { rc = -1; break; }   // jpgcoder.cc:3229-3236
                    break;
                }
            }
            if (!rc && eobrun == 0 && (last >> 16) == 0u) rc = -1;               // the band ends in a ZRL
        }
        if (!rc && eobrun > 0) {
            if (p < to) {                                                        // the rest of the band: correction bits only
                lepwave::wave_select(cmark, ahead, acc);
                off += nztotal - (base + off - acc);
            }
            --eobrun;
        }
        if (!rc && nztotal) {
            // every non-zero position fetches its own correction bit
            while (ring_hi < ((base + off) >> 5) + 2u) commit();                 // (the bits just stepped over must be in the ring)
            LANES(l) {
                if (((nzm >> l) & 1ull) && L(cmark) != kWinNoMark && ring_bit(L(cmark) + L(nzrank))) {
                    const int old = L(cur);
                    L(nv) = (uint32_t)(uint16_t)(int16_t)(old + (int16_t)((uint16_t)(int16_t)(old > 0 ? 1 : -1) << sal));
                }
            }
        }
        // what changed (new coefficients are not zero, corrected ones neither) is stored a block later
        LV(int, chg);
        LANES(l) { L(chg) = L(nv) != 0; L(held) = L(nv); }
        held_mask = lepwave::wave_ballot(chg);
        held_dpos = dpos;
        peobrun = (int)eobrun;
        return rc;
    }

    // BitReader::unpad (bitops.hh) at the chain's position
    WDEV int unpad_win(int fillbit) {
        const int rem = (int)(pos() & 7u);
        if (!rem) return fillbit;
        const int nb = 8 - rem;
        const uint32_t v = bits_at((uint32_t)nb);                                // first bit read = most significant
        int f = 0, last = 0, o = 0;
        for (int i = 0; i < nb; ++i) { last = (int)((v >> (nb - 1 - i)) & 1u); f |= last << o; ++o; }
        while (o < 7) { f |= last << o; ++o; }
        return f & 255;
    }

    // scan order -> place in the frame (DC refinement scans, once per 64 blocks; next_mcupos / next_mcuposn, jpgcoder.cc:5402-5456)
    WDEV void locate_dc(uint32_t idx, uint32_t P, int* cmp, int* dpos) const {
        if (sc->cmpc == 1) {
            *cmp = sc->cmp[0];
            const uint32_t nch = (uint32_t)sc->nch[*cmp];
            *dpos = (int)((idx / nch) * (uint32_t)sc->t.bch[*cmp] + idx % nch);
            return;
        }
        const uint32_t m = idx / P;
        uint32_t q = idx % P;
        int i = 0;
        while (i + 1 < sc->cmpc && q >= (uint32_t)sc->mbs[sc->cmp[i]]) { q -= (uint32_t)sc->mbs[sc->cmp[i]]; ++i; }
        const int c = sc->cmp[i];
        *cmp = c;
        const uint32_t hs = (uint32_t)sc->t.hs[c], mh = (uint32_t)sc->t.mcuh, vs = (uint32_t)sc->t.vs[c];
        *dpos = (int)(((m / mh) * vs + q / hs) * (uint32_t)sc->t.bch[c] + (m % mh) * hs + q % hs);
    }

    template <bool PIPE = false>
    WDEV void run_scan_win(const ProgDecScan* scan, ProgWinShared* shared, HuffDecRow* rows_arena, const ProgDeps* follow = nullptr, uint32_t* rows_done = nullptr,
                           int index = 0) {
        sc = scan; img = &scan->t; ws = shared; sh = nullptr; status = 0;
        deps = follow; progress = PIPE ? rows_done : nullptr; self = index; ready = 0;
        if (PIPE && progress) {
            bool any = false;
            for (int i = 0; i < 4; ++i) any = any || deps->dep[i] >= 0;
            if (!any) ready = 0x7fffffffu;
        }
        const bool dc = scan->to == 0;
        two_tables = dc;
        fast_kind = dc ? 0 : (scan->sah == 0 ? 1 : 2);
        held_mask = 0;
        {
            const int c = scan->cmp[0];
            k.from = scan->from; k.to = scan->to; k.sal = scan->sal; k.sah = scan->sah; k.max_eobrun = scan->max_eobrun;
            k.want_rows = scan->want_rows; k.tbl0 = scan->tbl[0] & 1;
            k.plus = (uint32_t)(uint16_t)((uint16_t)1 << scan->sal); k.minus = (uint32_t)(uint16_t)((uint16_t)(int16_t)-1 << scan->sal);
            k.cmp = c; k.bch = img->bch[c]; k.nch = scan->nch[c]; k.ncv = scan->ncv[c]; k.vs = img->vs[c];
            k.blocks = img->blocks[c];
            k.scan = img->scan; k.scan_len = img->scan_len; k.limit = img->scan_len * 8u;
        }
        LANES(l) {
            for (int i = l; i < 512; i += 64) { ws->lut[0][i] = img->lut[0][i]; ws->lut[1][i] = img->lut[1][i]; ws->lut[2][i] = img->lut[2][i]; }
            if (l < 24) { (&ws->maxcode[0][0])[l] = (&img->maxcode[0][0])[l]; (&ws->valoff[0][0])[l] = (&img->valoff[0][0])[l]; }
            for (int i = l; i < 3 * 256; i += 64) (&ws->longsym[0][0])[i] = (&img->longsym[0][0])[i];
            ws->z2a[l] = kZ2A[l];
            L(zz) = kZ2A[l];
        }
        LSYNC();
        base = 0; off = 0; ring_hi = 0;
        request(0);
        stage();
        HuffDecRow* rows = rows_arena + img->rows_off;
        int lastdc[4] = {0, 0, 0, 0};
        int padbit = -1;
        const int mcuh = img->mcuh, sal = scan->sal;
        const uint32_t limit = k.limit;
        eobrun = 0; peobrun = 0;
        int sta = 0;
        if (dc && scan->sah != 0) {
            // DC refinement: one bit per block and nothing else -- block i of the scan's order is bit i: 64 blocks per step, lane = block
            uint32_t P = 0;
            for (int i = 0; i < scan->cmpc; ++i) P += (uint32_t)scan->mbs[scan->cmp[i]];
            const uint32_t total = scan->cmpc == 1 ? (uint32_t)k.nch * (uint32_t)k.ncv : (uint32_t)img->mcuc * P;
            const uint32_t per_row = scan->cmpc == 1 ? (uint32_t)k.nch * (uint32_t)k.vs : (uint32_t)mcuh * P;   // blocks of the scan per MCU row
            for (uint32_t i0 = 0; i0 < total && !status; i0 += 64) {
                const uint32_t n = total - i0 < 64u ? total - i0 : 64u;
                if (PIPE) { await((i0 + n - 1) / per_row); if (status) break; }
                if (off) stage();
                LANES(l) {
                    if ((uint32_t)l < n && (L(win) >> 31)) {
                        int cmp, dpos;
                        locate_dc(i0 + (uint32_t)l, P, &cmp, &dpos);
                        int16_t* dst = img->blocks[cmp] + (int64_t)dpos * 64 + 49;
                        lepwave::gst(dst, (int16_t)(lepwave::gld(dst) + (int16_t)(1u << sal)));
                    }
                }
                off = n;
                if (PIPE) { LSYNC(); publish((i0 + n) / per_row); }
            }
            sta = status ? -1 : 2;
            if (pos() > limit) sta = -1;
        } else if (dc && scan->cmpc > 1) {
            // DC first stage over MCUs: one code per block.  Lane q of `where` / `what` describes block q of an MCU: its place relative
            // to the MCU's first block of that component, and component | table << 2 | scan slot << 3 | hs << 8
            LV(uint32_t, where); LV(uint32_t, what);
            uint32_t P = 0;
            LANES(l) { L(where) = 0; L(what) = 0; }
            for (int i = 0; i < scan->cmpc; ++i) {
                const int c = scan->cmp[i];
                const uint32_t hs = (uint32_t)img->hs[c], n = (uint32_t)scan->mbs[c], bch = (uint32_t)img->bch[c];
                LANES(l) {
                    const uint32_t q = (uint32_t)l - P;
                    if ((uint32_t)l >= P && q < n) { L(where) = (q / hs) * bch + q % hs; L(what) = (uint32_t)c | (uint32_t)(scan->tbl[i] & 1) << 2 | (uint32_t)i << 3 | hs << 8; }
                }
                P += n;
            }
            int rowbase[4] = {0, 0, 0, 0};                // component's first block of the current MCU row
            int stride[4];
            int16_t* frame[4];
            for (int c = 0; c < 4; ++c) { stride[c] = img->vs[c] * img->bch[c]; frame[c] = img->blocks[c]; }
            const int mcuc = img->mcuc;
            const int mcuv_rows = (mcuc + mcuh - 1) / mcuh;
            bool bad = false;
            for (int my = 0; my < mcuv_rows && !bad; ++my) {
                if (k.want_rows) {
                    const uint32_t bp = pos();
                    LANES(l) if (l == 0) { rows[my].bitpos = bp; for (int c = 0; c < 4; ++c) rows[my].last_dc[c] = (int16_t)lastdc[c]; rows[my].aux = 0; }
                }
                if (PIPE) { await((uint32_t)my); if (status) { bad = true; break; } }
                const int mx_end = (my + 1) * mcuh <= mcuc ? mcuh : mcuc - my * mcuh;
                for (int mx = 0; mx < mx_end && !bad; ++mx) {
#pragma nounroll
                    for (uint32_t q = 0; q < P; ++q) {
                        const uint32_t wt = lepwave::wave_read(what, (int)q), wh = lepwave::wave_read(where, (int)q);
                        const int c = (int)(wt & 3u);
                        uint32_t len, sym, f16;
                        if (!code_at((int)((wt >> 2) & 1u), &len, &sym, &f16) || sym > 15u) { bad = true; break; }
                        off += len + sym;
                        const int last = c == 0 ? lastdc[0] : (c == 1 ? lastdc[1] : (c == 2 ? lastdc[2] : lastdc[3]));
                        const int v = (int16_t)(devli(sym, sym ? f16 >> (16u - sym) : 0u) + last);
                        if (c == 0) lastdc[0] = v; else if (c == 1) lastdc[1] = v; else if (c == 2) lastdc[2] = v; else lastdc[3] = v;
                        const int rb = c == 0 ? rowbase[0] : (c == 1 ? rowbase[1] : (c == 2 ? rowbase[2] : rowbase[3]));
                        int16_t* fr = c == 0 ? frame[0] : (c == 1 ? frame[1] : (c == 2 ? frame[2] : frame[3]));
                        int16_t* dst = fr + (int64_t)(rb + mx * (int)(wt >> 8) + (int)wh) * 64 + 49;
                        LANES(l) if (l == 0) lepwave::gst(dst, (int16_t)((uint16_t)v << sal));
                        if (pos() > limit) { bad = true; break; }
                    }
                }
                for (int c = 0; c < 4; ++c) rowbase[c] += stride[c];
                if (PIPE && !bad) { LSYNC(); publish((uint32_t)(my + 1)); }
            }
            sta = bad ? -1 : 2;
        } else {
            // one component: DC first stage or an AC scan, the component's nch x ncv blocks row by row (next_mcuposn, jpgcoder.cc:5432-5456)
            int col = 0, row = 0, dpos = 0, cur_row = -1, rstw = 0;
            bool stray = false;        // a run has carried the walk into the frame's padding rows: from there on the reference's own steps
            while (sta == 0) {
                if (row != cur_row) {
                    if (dc && k.want_rows && (k.cmp == 0 || cur_row < 0)) {
                        const uint32_t bp = pos();
                        LANES(l) if (l == 0) { rows[row].bitpos = bp; for (int c = 0; c < 4; ++c) rows[row].last_dc[c] = (int16_t)lastdc[c]; rows[row].aux = 0; }
                    }
                    flush_store();
                    if (PIPE) {
                        if (cur_row >= 0) { LSYNC(); publish((uint32_t)(row / k.vs)); }
                        await((uint32_t)(row / k.vs));
                        if (status) { sta = -1; break; }
                    }
                    cur_row = row;
                    if (!dc && k.sah != 0) request_block(dpos);   // (behind the wait: the scans in front have passed this block row)
                }
                if (dc) {
                    uint32_t len, sym, f16;
                    if (!code_at(k.tbl0, &len, &sym, &f16) || sym > 15u) { sta = -1; break; }
                    off += len + sym;
                    const int v = (int16_t)(devli(sym, sym ? f16 >> (16u - sym) : 0u) + lastdc[0]);
                    lastdc[0] = v;
                    int16_t* dst = k.blocks + (int64_t)dpos * 64 + 49;
                    LANES(l) if (l == 0) lepwave::gst(dst, (int16_t)((uint16_t)v << sal));
                } else {
                    const int rc = k.sah == 0 ? ac_first_win(dpos) : ac_refine_win(dpos, col + 1 < k.nch);
                    if (rc < 0) { sta = -1; break; }
                    if (k.sah == 0 && eobrun) {          // a run: its blocks are passed as a whole
                        if (!stray && col + (int)eobrun < k.nch) { col += (int)eobrun; dpos += (int)eobrun; eobrun = 0; }   // (inside the block row: skip_run comes to the same)
                        else {                           // skip_eobrun's own arithmetic (jpgcoder.cc:5462-5500), divisions and all
                            sta = skip_run(k.cmp, &dpos, &rstw);
                            row = dpos / k.bch; col = dpos - row * k.bch;
                            if (row >= k.ncv || col >= k.nch) stray = true;
                        }
                    }
                }
                if (sta == 0) {
                    if (stray) { sta = next_noninterleaved(k.cmp, &dpos, &rstw); row = dpos / k.bch; col = dpos - row * k.bch; }
                    else {
                        ++col; ++dpos;
                        if (col >= k.nch) { col = 0; ++row; dpos = row * k.bch; }
                        if (row >= k.ncv) sta = 2;
                    }
                }
                if (pos() > limit) { sta = -1; break; }
            }
            flush_store();
            if (sta > 0 && eobrun > 0) sta = -1;
            if (dc) { const int v = lastdc[0]; lastdc[0] = 0; lastdc[k.cmp & 3] = v; }
        }
        if (sta == -1) { if (!PIPE || !status) status = 1; }
        else {
            const int got = unpad_win(255);
            padbit = (int8_t)got;
        }
        if (!status && pos() != limit) status = 2;   // bytes left over, or missing
        if (PIPE) publish(0x7fffffffu);
        const uint32_t bp = pos();
        HuffDecRow* fin = rows_arena + scan->result_off;
        LANES(l) if (l == 0) {
            fin->bitpos = bp;
            for (int c = 0; c < 4; ++c) fin->last_dc[c] = (int16_t)lastdc[c];
            fin->aux = (padbit & 255) | (status << 8);
        }
    }
};

// which scans this form takes (the others keep lep_huffprogdec.h's): no restart intervals
inline bool prog_win_takes(const ProgDecScan& s) { return s.t.rsti == 0 && s.t.scan_len < (1u << 27) && (reinterpret_cast<uintptr_t>(s.t.scan) & 15u) == 0; }
constexpr int32_t kProgDecWin = 1;      // ProgDecScan::pad: this form decodes the scan

}  // namespace lephuff
