// lep_huffprog_simt.h -- PROGRESSIVE JPEG scans written from the finished coefficient frame with one LANE per run of blocks
// (round 6; BASELINE.json configs[4], decode direction).  Same bytes as lep_huffprog.h's wavefront-per-scan writer, which restates
// encode_dc_prg_fs / _sa, encode_ac_prg_fs / _sa, encode_eobrun, encode_crbits (src/lepton/jpgcoder.cc:4991-5400) inside the scan
// loop of recode_jpeg (:3309-3716).
//
// lep_huffprog.h gives a wavefront to one scan and walks its blocks one after the other: the last luma refinement scan of a 4K
// file is 129,600 blocks, ~270 ms, on a chip that 2,560 such wavefronts leave nine tenths empty.  What ties the blocks of a scan
// together is only WHERE a block's bits go and how long the END-OF-BAND RUN is that a block opens -- and both are sums:
//
//   * every block's contribution to the stream is contiguous and in block order once the code of a run is charged to the block
//     the run STARTS with: [the block's codes, with the correction bits in front of its last new coefficient] [EOBn code, if a
//     run (or the next 32767-block stretch of one) starts here] [the correction bits the block holds back].  The reference emits
//     the EOBn code when the run ENDS, but it emits it in front of all the bits its blocks held back (encode_eobrun, then
//     encode_crbits), so in the stream it stands exactly there;
//   * how long that run is: the blocks with an empty band that follow, up to the next block that codes something (or 32767, or
//     the scan's end) -- a suffix minimum over "first coding block at or after unit u";
//   * how far into a run a unit starts: distance to the last block that coded something -- a prefix maximum.
//
// Passes (kernels in lep_gpu.hip; a unit = kProgUnit consecutive blocks of the scan's block order, DC scans of several components
// kProgDcMcus MCUs):
//   1  count   lane = unit: bits of everything but the EOBn codes; per unit which blocks code something (nonE) and which of
//              those leave their band open (P: the last coefficient is not at `to`)
//   2  place   one wavefront per scan: prefix maximum / suffix minimum over the units -> every unit's run state (cin, la); the
//              EOBn codes' bits from the two masks; exclusive prefix sum -> the unit's bit position
//   3  code    lane = unit: the same walk, bits OR-ed into the scan's zero-filled bit buffer (lep_huff_simt.h's LaneSink)
//   4  stuff   one wavefront per scan: pad bits, FF -> FF 00, clipped to the scan's slot
// Restart intervals keep the wavefront kernel (ProgScan.pad says which kernel owns a scan): libjpeg's progressive files have none.
// SPMD layer of lep_wave.h: tests/emu steps every pass on the CPU against lep_huffprog.h and the host re-coder, byte for byte.
#pragma once
#include "lep_huff_simt.h"
#include "lep_huffprog.h"

namespace lephuff {

constexpr int kProgUnit = 32;           // blocks per unit (AC scans, one-component DC scans): the two type masks are one dword each
constexpr int kProgDcMcus = 8;          // MCUs per unit (interleaved DC scans)
constexpr uint32_t kProgScanSimt = 1;   // ProgScan::pad: the lane-per-unit kernels own this scan
constexpr uint32_t kProgNone = 0x7fffffffu;

struct ProgSimtScan {       // per scan taken by this form
    uint32_t scan;          // index into the launch's ProgScan array
    uint32_t first_unit;    // its first entry in the unit arrays
    uint32_t nunits;
    uint32_t nblocks;       // blocks of the scan (AC / one-component DC scans), MCUs (interleaved DC scans)
    uint64_t buf_off;       // its bit buffer (bytes, 16-byte aligned) in the scratch arena
    uint32_t buf_bytes;     // multiple of 16
    uint32_t total_bits;    // pass 2
};
struct ProgSimtWave { uint32_t pscan, first_unit; };   // lane l = unit first_unit + l of ProgSimtScan pscan
// The bit buffers of one image's scans share a region sized by the FILE (its scans are parts of it: together they are shorter than
// the file), handed out once pass 2 knows what every scan needs -- a buffer per scan sized by what the scan MAY need would be ten
// times that.  A region that does not suffice leaves its later scans without a buffer: they answer "outgrew" and the host re-coder
// takes the file.
struct ProgSimtRegion { uint32_t first_ps, nps; uint64_t off, bytes; };

// the unit arrays, each `units` long (one allocation, SoA)
struct ProgSimtUnits {
    uint32_t* bits;         // pass 1: bits without EOBn codes;  pass 2: the unit's bit position
    uint32_t* nonE;         // pass 1: bit i = block i of the unit codes something
    uint32_t* pmask;        // pass 1: bit i = ... and leaves its band open (an end-of-band run starts with it)
    uint32_t* cin;          // pass 2: blocks of the current run in front of the unit's first block (0: none open)
    uint32_t* la;           // pass 2: blocks with an empty band that follow the unit's last block
    WDEV void set(uint32_t* base, size_t units) { bits = base; nonE = base + units; pmask = base + 2 * units; cin = base + 3 * units; la = base + 4 * units; }
};
constexpr int kProgSimtUnitWords = 5;

struct ProgSimtShared { uint32_t code[2][256]; };

// bits of the EOBn code of a run of `run` blocks (encode_eobrun, jpgcoder.cc:5337-5368)
WDEV void prog_eob_code(const uint32_t* ac, uint32_t run, uint32_t* bits, uint32_t* n) {
    int s = bitlen(run & 0xffffu);
    if (s) --s;
    const uint32_t e = ac[(s << 4) & 255];
    *n = (e >> 16) + (uint32_t)s;
    *bits = ((e & 0xffffu) << s) | (s ? run - (1u << s) : 0u);
}

template <bool WRITE>
struct ProgSimtLane {
    const ProgImage* pim;
    const ProgScan* sc;
    const ProgSimtShared* sh;
    LaneSink<WRITE> sink;
    int from, to, sal;

    WDEV void put_coef(int table, uint32_t runsize_hi, int t) {
        const int at = (t < 0 ? -t : t) & 0xffff;
        const uint32_t s = (uint32_t)bitlen((uint32_t)at);
        const uint32_t val = (uint32_t)((t > 0) ? t : (t - 1) + (1 << s)) & ((1u << s) - 1u);
        const uint32_t e = sh->code[table][(runsize_hi + s) & 255u];
        sink.put(((e & 0xffffu) << s) | val, (e >> 16) + s);
    }
    WDEV void load(const int16_t* blk, uint32_t* w) const {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(blk);
        for (int i = 0; i < 32; ++i) w[i] = src[i];
    }

    // ---- AC first stage (encode_ac_prg_fs, jpgcoder.cc:5230-5290): 0 empty band, 1 band left open, 2 band closed --------------
    template <int K>
    WDEV void fs_step(const uint32_t* w, int& prev) {
        if (K < from || K > to) return;
        const int t = fdiv2(coef_at<K>(w), sal);
        if (t != 0) {
            const int run = K - prev - 1;
            if (run >= 16) {
                const uint32_t zrl = sh->code[0][0xF0];
                for (int i = run >> 4; i > 0; --i) sink.put(zrl & 0xffffu, zrl >> 16);
            }
            put_coef(0, (uint32_t)(run & 15) << 4, t);
            prev = K;
        }
    }
    template <int K0, int K1>
    WDEV void fs_range(const uint32_t* w, int& prev) {
        if constexpr (K0 < K1) { fs_step<K0>(w, prev); fs_range<K0 + 1, K1>(w, prev); }
    }
    WDEV int ac_first_block(const uint32_t* w) {
        int prev = from - 1;
        fs_range<1, 64>(w, prev);
        return prev == from - 1 ? 0 : (prev < to ? 1 : 2);
    }

    // ---- AC refinement (encode_ac_prg_sa + encode_crbits, jpgcoder.cc:5292-5400) ----------------------------------------------
    // per band position: Z zero, N newly non-zero (+-1), O already non-zero (one correction bit); masks over zig-zag positions
    template <int K>
    WDEV void sa_step(const uint32_t* w, uint64_t& Nm, uint64_t& Om, uint64_t& Pm) {
        if (K < from || K > to) return;
        const int t = fdiv2(coef_at<K>(w), sal);
        const int a = t < 0 ? -t : t;
        if (a == 1) { Nm |= 1ull << K; if (t > 0) Pm |= 1ull << K; }
        else if (a > 1) { Om |= 1ull << K; if (a & 1) Pm |= 1ull << K; }
    }
    template <int K0, int K1>
    WDEV void sa_range(const uint32_t* w, uint64_t& Nm, uint64_t& Om, uint64_t& Pm) {
        if constexpr (K0 < K1) { sa_step<K0>(w, Nm, Om, Pm); sa_range<K0 + 1, K1>(w, Nm, Om, Pm); }
    }
    // the block's masks: N, O and P (for an N position: the coefficient is positive; for an O position: its correction bit)
    WDEV void refine_masks(const uint32_t* w, uint64_t* Nm, uint64_t* Om, uint64_t* Pm) {
        uint64_t n = 0, o = 0, p = 0;
        sa_range<1, 64>(w, n, o, p);
        *Nm = n; *Om = o; *Pm = p;
    }
    WDEV void put_pending(uint64_t pend, uint32_t cnt) {   // cnt <= 63 correction bits, oldest = most significant
        if (cnt > 32) { sink.put((uint32_t)(pend >> 32) & ((1u << (cnt - 32)) - 1u), cnt - 32); sink.put((uint32_t)pend, 32); }
        else if (cnt) sink.put((uint32_t)pend & (cnt == 32 ? 0xffffffffu : (1u << cnt) - 1u), cnt);
    }
    // the part in front of the last new coefficient; returns 0 / 1 / 2 like ac_first_block
    WDEV int ac_refine_front(uint64_t Nm, uint64_t Om, uint64_t Pm) {
        if (!Nm) return 0;
        const int eob = 64 - __builtin_clzll(Nm);   // 1 + last N
        uint32_t z = 0, cnt = 0;
        uint64_t pend = 0;
        const uint32_t zrl = sh->code[0][0xF0];
        for (int k = from; k < eob; ++k) {
            const uint64_t b = 1ull << k;
            if (Om & b) { pend = (pend << 1) | (uint64_t)((Pm >> k) & 1ull); ++cnt; }
            else if (Nm & b) {
                const uint32_t e = sh->code[0][((z << 4) + 1u) & 255u];
                sink.put(((e & 0xffffu) << 1) | (uint32_t)((Pm >> k) & 1ull), (e >> 16) + 1u);
                put_pending(pend, cnt);
                z = 0; cnt = 0; pend = 0;
            } else if (++z == 16) {
                sink.put(zrl & 0xffffu, zrl >> 16);
                put_pending(pend, cnt);
                z = 0; cnt = 0; pend = 0;
            }
        }
        return eob <= to ? 1 : 2;
    }
    // the correction bits of the O positions from the last new coefficient on: held back behind the EOBn code
    WDEV void ac_refine_tail(uint64_t Nm, uint64_t Om, uint64_t Pm) {
        const int eob = Nm ? 64 - __builtin_clzll(Nm) : from;
        uint64_t tail = eob >= 64 ? 0ull : Om & ~((1ull << eob) - 1);
        uint64_t pend = 0;
        uint32_t cnt = 0;
        while (tail) {
            const int k = __builtin_ctzll(tail);
            tail &= tail - 1;
            pend = (pend << 1) | (uint64_t)((Pm >> k) & 1ull);
            ++cnt;
        }
        put_pending(pend, cnt);
    }

    WDEV void put_eob(uint32_t run) {
        uint32_t bits, n;
        prog_eob_code(sh->code[0], run, &bits, &n);
        sink.put(bits, n);
    }

    // ---- DC (encode_dc_prg_fs / _sa, jpgcoder.cc:4991-5040) ------------------------------------------------------------------------
    WDEV int dc_of(int cmp, int dpos) const { return (int)pim->blocks[cmp][(int64_t)dpos * 64 + kZ2A_const(0)] >> sal; }
    WDEV void dc_block(int slot, int cmp, int dpos, int* last) {
        if (sc->sah == 0) {
            const int v = dc_of(cmp, dpos);
            const int d = (int16_t)(v - *last);
            *last = v;
            put_coef(sc->tbl[slot] & 1, 0, d);
        } else sink.put((uint32_t)(dc_of(cmp, dpos) & 1), 1);
    }
};

WDEV void prog_simt_tables(const ProgScan* sc, ProgSimtShared* sh) {
    LANES(l) for (int i = l; i < 512; i += 64) (&sh->code[0][0])[i] = (&sc->code[0][0])[i];
    LSYNC();
}

// block idx of a one-component scan -> its place in the component's frame (next_mcuposn, jpgcoder.cc:5432-5456)
WDEV int prog_dpos(const ProgImage* pim, int cmp, uint32_t idx) {
    const uint32_t nch = (uint32_t)pim->nch[cmp];
    return (int)((idx / nch) * (uint32_t)pim->bch[cmp] + idx % nch);
}

// passes 1 and 3: lanes = units first_unit .. of scan `ps`
template <bool WRITE>
WDEV void prog_simt_units(const ProgImage* images, const ProgScan* scans, const ProgSimtScan* psp, ProgSimtShared* sh, ProgSimtUnits U, uint8_t* scratch, uint32_t first_unit) {
    const ProgSimtScan ps = *psp;
    const ProgScan* sc = scans + ps.scan;
    const ProgImage* pim = images + sc->image;
    prog_simt_tables(sc, sh);
    const bool dc = sc->to == 0, interleaved = sc->cmpc > 1;
    const uint32_t max = (uint32_t)sc->max_eobrun;
    LANES(l) {
        const uint32_t u = first_unit + (uint32_t)l;
        if (u < ps.nunits) {
            ProgSimtLane<WRITE> d;
            d.pim = pim; d.sc = sc; d.sh = sh; d.from = sc->from; d.to = sc->to; d.sal = sc->sal;
            const size_t gu = (size_t)ps.first_unit + u;
            d.sink.start(WRITE ? U.bits[gu] : 0u, reinterpret_cast<uint32_t*>(scratch + ps.buf_off), ps.buf_bytes >> 2);
            uint32_t nonE = 0, pmask = 0;
            if (dc && interleaved) {
                // MCUs [m0, m1): inside an MCU the scan's components in order, each hs x vs blocks (next_mcupos, jpgcoder.cc:5402-5430)
                const int m0 = (int)u * kProgDcMcus, m1 = m0 + kProgDcMcus < (int)ps.nblocks ? m0 + kProgDcMcus : (int)ps.nblocks;
                const int mcuh = pim->mcuh;
                int last[4] = {0, 0, 0, 0};
                if (m0 > 0 && sc->sah == 0) {       // the DC of each component's last block in the MCU in front
                    const int m = m0 - 1, row = m / mcuh, mx = m - row * mcuh;
                    for (int i = 0; i < sc->cmpc; ++i) {
                        const int cmp = sc->cmp[i], hs = pim->hs[cmp], vs = pim->vs[cmp];
                        const int v = d.dc_of(cmp, (row * vs + vs - 1) * pim->bch[cmp] + mx * hs + hs - 1);
                        if (i == 0) last[0] = v; else if (i == 1) last[1] = v; else if (i == 2) last[2] = v; else last[3] = v;
                    }
                }
                int row = m0 / mcuh, mx = m0 - row * mcuh;
                for (int m = m0; m < m1; ++m) {
                    for (int i = 0; i < sc->cmpc; ++i) {
                        const int cmp = sc->cmp[i], hs = pim->hs[cmp], vs = pim->vs[cmp], bch = pim->bch[cmp];
                        int cur = i == 0 ? last[0] : (i == 1 ? last[1] : (i == 2 ? last[2] : last[3]));
                        for (int v = 0; v < vs; ++v)
                            for (int h = 0; h < hs; ++h) d.dc_block(i, cmp, (row * vs + v) * bch + mx * hs + h, &cur);
                        if (i == 0) last[0] = cur; else if (i == 1) last[1] = cur; else if (i == 2) last[2] = cur; else last[3] = cur;
                    }
                    if (++mx == mcuh) { mx = 0; ++row; }
                }
            } else {
                const int cmp = sc->cmp[0];
                const uint32_t b0 = u * (uint32_t)kProgUnit, n = ps.nblocks - b0 < (uint32_t)kProgUnit ? ps.nblocks - b0 : (uint32_t)kProgUnit;
                if (dc) {
                    int last = (b0 > 0 && sc->sah == 0) ? d.dc_of(cmp, prog_dpos(pim, cmp, b0 - 1)) : 0;
                    for (uint32_t i = 0; i < n; ++i) d.dc_block(0, cmp, prog_dpos(pim, cmp, b0 + i), &last);
                } else {
                    // pass 3 knows from pass 1 which blocks code something and from pass 2 how the unit stands in its runs
                    uint32_t c = 0;                    // blocks of the open run in front of block i (the reference's eobrun)
                    if (WRITE) { nonE = U.nonE[gu]; pmask = U.pmask[gu]; c = U.cin[gu]; }
                    const uint32_t la = WRITE ? U.la[gu] : 0u;
                    for (uint32_t i = 0; i < n; ++i) {
                        uint32_t w[32];
                        d.load(pim->blocks[cmp] + (int64_t)prog_dpos(pim, cmp, b0 + i) * 64, w);
                        int type;
                        uint64_t Nm = 0, Om = 0, Pm = 0;
                        if (sc->sah == 0) type = d.ac_first_block(w);
                        else { d.refine_masks(w, &Nm, &Om, &Pm); type = d.ac_refine_front(Nm, Om, Pm); }
                        if (!WRITE) { if (type) nonE |= 1u << i; if (type == 1) pmask |= 1u << i; }
                        else if (type == 1 || (type == 0 && c == 0)) {
                            const uint32_t rest = i + 1 < 32u ? nonE >> (i + 1) : 0u;
                            const uint32_t follow = rest ? (uint32_t)__builtin_ctz(rest) : n - 1 - i + la;
                            d.put_eob(follow + 1 < max ? follow + 1 : max);
                        }
                        if (WRITE) { if (type) c = type == 1 ? 1u : 0u; else ++c; if (c == max) c = 0; }
                        if (sc->sah != 0) d.ac_refine_tail(Nm, Om, Pm);
                    }
                }
            }
            d.sink.finish();
            if (!WRITE) { U.bits[gu] = d.sink.total; U.nonE[gu] = nonE; U.pmask[gu] = pmask; }
        }
    }
}

// pass 2: one wavefront per scan
WDEV void prog_simt_place(const ProgScan* scans, ProgSimtScan* psp, ProgSimtUnits U) {
    const ProgSimtScan ps = *psp;
    const ProgScan* sc = scans + ps.scan;
    const bool ac = sc->to != 0;
    const uint32_t nunits = ps.nunits, fu = ps.first_unit, max = (uint32_t)sc->max_eobrun;
    if (ac) {
        // (a) la: empty-band blocks behind every unit = (first coding block at or after the next unit, or the scan's end) - the unit's end.
        //     Suffix minimum, batches of 64 units from the back.
        uint32_t carry = ps.nblocks;   // first coding block at or after the batch behind this one
        for (uint32_t top = nunits; top > 0;) {
            const uint32_t base = top > 64 ? top - 64 : 0, cnt = top - base;
            LV(int, v); LV(int, sm);
            LANES(l) {
                // lane l looks at unit base + l + 1 (its own successor)
                const uint32_t nx = base + (uint32_t)l + 1;
                int first = (int)kProgNone;
                if ((uint32_t)l < cnt && nx < nunits) { const uint32_t m = U.nonE[fu + nx]; if (m) first = (int)(nx * (uint32_t)kProgUnit + (uint32_t)__builtin_ctz(m)); }
                L(v) = first;
            }
            lepwave::wave_suffix_min(v, sm);
            LANES(l) {
                const uint32_t u = base + (uint32_t)l;
                if ((uint32_t)l < cnt) {
                    uint32_t nn = (uint32_t)L(sm) < carry ? (uint32_t)L(sm) : carry;
                    const uint32_t end = (u + 1) * (uint32_t)kProgUnit < ps.nblocks ? (u + 1) * (uint32_t)kProgUnit : ps.nblocks;
                    U.la[fu + u] = nn - end;
                }
            }
            {   // the batch in front needs the first coding block at or after unit `base`
                const uint32_t sm0 = lepwave::wave_read((const uint32_t*)sm, 0);
                uint32_t own = kProgNone;
                const uint32_t m = U.nonE[fu + base];
                if (m) own = base * (uint32_t)kProgUnit + (uint32_t)__builtin_ctz(m);
                uint32_t c2 = sm0 < carry ? sm0 : carry;
                carry = own < c2 ? own : c2;
            }
            top = base;
        }
        LSYNC();
        // (b) cin: blocks of the open run in front of every unit.  Block x stands (x - B) mod max blocks into a run, B = the last
        //     coding block in front of it if that one left its band open, the block behind it if it closed it (no coding block
        //     in front: the scan's first block).  Prefix maximum of B over the units.
        uint32_t bcarry = 0;
        for (uint32_t base = 0; base < nunits; base += 64) {
            LV(int, v); LV(int, pm);
            LANES(l) {
                const uint32_t u = base + (uint32_t)l;   // lane l looks at unit u - 1 (its predecessor)
                int b = 0;
                if (u < nunits && u > 0) {
                    const uint32_t m = U.nonE[fu + u - 1];
                    if (m) { const uint32_t i = 31u - (uint32_t)__builtin_clz(m); b = (int)((u - 1) * (uint32_t)kProgUnit + i + (((U.pmask[fu + u - 1] >> i) & 1u) ? 0u : 1u)); }
                }
                L(v) = b;
            }
            lepwave::wave_prefix_max(v, pm);
            LANES(l) {
                const uint32_t u = base + (uint32_t)l;
                if (u < nunits) {
                    const uint32_t B = (uint32_t)L(pm) > bcarry ? (uint32_t)L(pm) : bcarry;
                    U.cin[fu + u] = (u * (uint32_t)kProgUnit - B) % max;
                }
            }
            const uint32_t last = lepwave::wave_read((const uint32_t*)pm, 63);
            bcarry = last > bcarry ? last : bcarry;
        }
        LSYNC();
    }
    // (c) the EOBn codes' bits, then the exclusive prefix sum of the units' bits
    uint32_t run = 0;
    for (uint32_t base = 0; base < nunits; base += 64) {
        LV(int, nb); LV(int, ex);
        LANES(l) {
            const uint32_t u = base + (uint32_t)l;
            uint32_t b = 0;
            if (u < nunits) {
                b = U.bits[fu + u];
                if (ac) {
                    const uint32_t b0 = u * (uint32_t)kProgUnit, n = ps.nblocks - b0 < (uint32_t)kProgUnit ? ps.nblocks - b0 : (uint32_t)kProgUnit;
                    const uint32_t nonE = U.nonE[fu + u], pmask = U.pmask[fu + u], la = U.la[fu + u];
                    uint32_t c = U.cin[fu + u];
                    for (uint32_t i = 0; i < n; ++i) {
                        const int type = (nonE >> i) & 1u ? (((pmask >> i) & 1u) ? 1 : 2) : 0;
                        if (type == 1 || (type == 0 && c == 0)) {
                            const uint32_t rest = i + 1 < 32u ? nonE >> (i + 1) : 0u;
                            const uint32_t follow = rest ? (uint32_t)__builtin_ctz(rest) : n - 1 - i + la;
                            uint32_t bits, nn;
                            prog_eob_code(sc->code[0], follow + 1 < max ? follow + 1 : max, &bits, &nn);
                            b += nn;
                        }
                        if (type) c = type == 1 ? 1u : 0u; else ++c;
                        if (c == max) c = 0;
                    }
                }
            }
            L(nb) = (int)b;
        }
        const int t = lepwave::wave_excl_scan(nb, ex);
        LANES(l) { const uint32_t u = base + (uint32_t)l; if (u < nunits) U.bits[fu + u] = run + (uint32_t)L(ex); }
        run += (uint32_t)t;
    }
    LANES(l) if (l == 0) psp->total_bits = run;
}

// between pass 2 and pass 3: one thread per image hands the image's region out to its scans
WDEV void prog_simt_assign(const ProgSimtRegion& r, ProgSimtScan* ps) {
    uint64_t off = r.off;
    for (uint32_t k = 0; k < r.nps; ++k) {
        ProgSimtScan* s = ps + r.first_ps + k;
        const uint64_t need = ((((uint64_t)s->total_bits + 7) >> 3) + 64 + 15) & ~(uint64_t)15;
        if (off + need <= r.off + r.bytes && need < 0xfffffff0ull) { s->buf_off = off; s->buf_bytes = (uint32_t)need; off += need; }
        else { s->buf_off = r.off; s->buf_bytes = 0; }
    }
}

// pass 4: one wavefront per scan (abitwriter::pad, then the FF00 rule of the JPEG byte stream)
WDEV void prog_simt_stuff(const ProgImage* images, const ProgScan* scans, const ProgSimtScan& ps, uint8_t* scratch, uint8_t* arena, uint32_t* out_len) {
    const ProgScan* sc = scans + ps.scan;
    const ProgImage* pim = images + sc->image;
    uint32_t* buf = reinterpret_cast<uint32_t*>(scratch + ps.buf_off);
    uint32_t total = ps.total_bits;
    if (ps.buf_bytes < 64u) {   // no buffer (prog_simt_assign): the host re-coder's
        LANES(l) if (l == 0) out_len[ps.scan] = 0x80000000u;
        return;
    }
    const uint32_t room = ps.buf_bytes * 8u - 64u;
    const bool over = total > room;
    if (over) total = room;
    if (total & 7u) {
        const uint32_t pend = total & 7u, n = 8u - pend;
        uint32_t v = 0;
        for (uint32_t j = 0; j < n; ++j) v = (v << 1) | (uint32_t)((pim->padbit >> j) & 1);
        LANES(l) if (l == 0) buf[total >> 5] |= v << (32u - (total & 31u) - n);
        LSYNC();
        total += n;
    }
    const uint32_t nb = total >> 3, cap = sc->out_cap;
    uint8_t* out = arena + sc->out_off;
    uint32_t written = 0;
    for (uint32_t base = 0; base < nb; base += 1024) {
        LV(int, nff); LV(int, before);
        LV(uint32_t, w0); LV(uint32_t, w1); LV(uint32_t, w2); LV(uint32_t, w3);
        LANES(l) {
            const uint32_t i = base + 16u * (uint32_t)l;
            uint32_t a = 0, b = 0, c = 0, d = 0;
            int n = 0;
            if (i < nb) {
                const uint32_t* p = buf + (i >> 2);
                a = p[0]; b = p[1]; c = p[2]; d = p[3];
                const uint32_t have = nb - i < 16u ? nb - i : 16u;
                for (uint32_t k = 0; k < have; ++k) {
                    const uint32_t word = k < 4 ? a : (k < 8 ? b : (k < 12 ? c : d));
                    n += ((word >> (24 - 8 * (k & 3))) & 255u) == 0xffu;
                }
            }
            L(w0) = a; L(w1) = b; L(w2) = c; L(w3) = d; L(nff) = n;
        }
        const int ffs = lepwave::wave_excl_scan(nff, before);
        LANES(l) {
            const uint32_t i = base + 16u * (uint32_t)l;
            if (i < nb) {
                const uint32_t have = nb - i < 16u ? nb - i : 16u;
                uint32_t pos = written + 16u * (uint32_t)l + (uint32_t)L(before);
                for (uint32_t k = 0; k < have; ++k) {
                    const uint32_t word = k < 4 ? L(w0) : (k < 8 ? L(w1) : (k < 12 ? L(w2) : L(w3)));
                    const uint32_t byte = (word >> (24 - 8 * (k & 3))) & 255u;
                    if (pos < cap) out[pos] = (uint8_t)byte;
                    ++pos;
                    if (byte == 0xffu) { if (pos < cap) out[pos] = 0; ++pos; }
                }
            }
        }
        written += (nb - base < 1024u ? nb - base : 1024u) + (uint32_t)ffs;
    }
    LANES(l) if (l == 0) out_len[ps.scan] = (written < cap ? written : cap) | ((over || written > cap) ? 0x80000000u : 0u);
}

// which scans this form takes, and how many units it cuts one into
inline bool prog_simt_takes(const ProgImage& im, const ProgScan& sc, uint32_t* nblocks, uint32_t* nunits) {
    if (prog_scan_rsti(im, sc) != 0 || prog_is_sequential(sc)) return false;
    const bool dc = sc.to == 0;
    if (sc.cmpc < 1 || sc.cmpc > 4 || (!dc && (sc.cmpc != 1 || sc.max_eobrun < 1))) return false;
    uint64_t n, per;
    if (sc.cmpc == 1) { const int c = sc.cmp[0]; if (c < 0 || c > 3 || im.nch[c] <= 0 || im.ncv[c] <= 0) return false; n = (uint64_t)im.nch[c] * (uint64_t)im.ncv[c]; per = kProgUnit; }
    else { if (im.mcuc <= 0 || im.mcuh <= 0) return false; n = (uint64_t)im.mcuc; per = kProgDcMcus; }
    if (n == 0 || n > 0x3fffffffu) return false;
    *nblocks = (uint32_t)n; *nunits = (uint32_t)((n + per - 1) / per);
    return true;
}

}  // namespace lephuff
