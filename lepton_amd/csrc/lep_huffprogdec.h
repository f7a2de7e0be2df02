// lep_huffprogdec.h -- PROGRESSIVE JPEG scans decoded into the coefficient frame ON THE GPU (BASELINE.json configs[4], encode
// direction): the scan loop of the reference's decode_jpeg for progressive files (src/lepton/jpgcoder.cc:2975-3260) with
// decode_dc_prg_fs / _sa, decode_ac_prg_fs / _sa, decode_eobrun_sa, skip_eobrun (:4968-5335, :5462-5500).
//
// One wavefront per (image, scan).  Unlike the re-encoder (lep_huffprog.h) the scans of an image are NOT independent here: a
// refinement scan needs to know which coefficients of its band are already non-zero, and both DC scans write the same
// coefficient.  The host sorts the scans of a batch into dependency levels (a scan's level = 1 + the highest level among the
// earlier scans of the same component whose band overlaps its own); one launch per level, at most a handful.
// Inside a scan decoding is serial, written as uniform vector code on lep_huffdec.h's window reader / one-step code
// lookup, one step per CODE: what a code places goes into a block image in LDS, the correction bits of the positions it passes
// are one read and a lane-parallel update, and the coefficients that changed leave with one masked lane-parallel store per
// block (scans of the same level write different positions of the same blocks -- libjpeg's script: luma 1..5 and luma 6..63 in
// parallel -- so whole-block stores would race).
// Anything irregular -- a code that does not exist, a zero run or an end-of-band run past its bounds, data left over or
// missing, pad bits that change, end-of-band runs a canonical encoder would have merged, a block ending in a coded zero --
// ends the scan with a status: the host parser then takes the whole file and answers as the reference does.
// SPMD layer of lep_wave.h: tests/emu runs it on the CPU against the host parser (frame, hand-off rows, pad bit).
#pragma once
#include "lep_huffdec.h"
#include <algorithm>
#include <vector>

namespace lephuff {

struct ProgDecScan {
    HuffDecImage t;         // scan = THIS scan's un-stuffed bytes; lut[0..1] DC tables 0 / 1, lut[2] = the scan's AC table; geometry; blocks;
                            // rows_off = where this scan's per-MCU-row records go (want_rows)
    int32_t cmpc, cmp[4];   // components of the scan, in scan order
    int32_t from, to, sah, sal;
    int32_t bcv[4], nch[4], ncv[4], mbs[4];
    int32_t tbl[4];         // DC scans: table slot (0 / 1) of each scan component
    int32_t max_eobrun;
    int32_t want_rows;      // 1: the first (DC) scan of the file: one record per MCU row, the hand-offs of the .lep header
    int32_t level;          // dependency level (host side; the kernel ignores it)
    int32_t pad;
    uint64_t result_off;    // this scan's final record {bit position, last DC, pad bits | status << 8} in the rows arena
};

// SEQUENTIAL frames coded in several scans (luma alone, then Cb + Cr together, ...; the reference decodes them with its sequential block
// loop under the general scan walk, jpgcoder.cc:3034-3175, and keeps the file's format flag 'X'): their scans come through the same
// descriptors -- from 0, to 63, which no progressive scan has -- with t holding the FRAME (all components, lut[0..1] DC tables 0 / 1,
// lut[2..3] AC tables 0 / 1, dc_tbl / ac_tbl per component) and are decoded by the sequential kernels (lep_huffdec_simt.h, lep_huffdec.h)
// from the image sequential_scan_image makes of them: the scan's components only, and for a scan of ONE component -- never interleaved:
// MCU = one block, the frame's padding blocks stepped over -- that component's nch x ncv blocks as MCUs (as parse_jpeg_prepare_gpu plans a
// one-component file).  rows_off: a record per MCU row of THAT geometry and the final one, which is result_off.
inline bool progdec_is_sequential(const ProgDecScan& s) { return s.from == 0 && s.to == 63; }
inline HuffDecImage sequential_scan_image(const ProgDecScan& s) {
    HuffDecImage im = s.t;
    im.ncomp = s.cmpc;
    for (int i = 0; i < 4; ++i) im.scan_cmp[i] = i < s.cmpc ? (s.cmp[i] & 3) : 0;
    if (s.cmpc == 1) {
        const int c = s.cmp[0] & 3;
        im.mcuh = s.nch[c]; im.mcuv = s.ncv[c]; im.mcuc = s.nch[c] * s.ncv[c];
        im.hs[c] = 1; im.vs[c] = 1;
    }
    im.flags = 0;
    return im;
}

// Whether the lane-per-subsequence decoder (lep_huffdec_simt.h) is the one to send such a scan to: restart intervals come without the
// table of marker positions here, and those go to the single-wave kernel.
inline bool sequential_scan_for_lanes(const HuffDecImage& im) { return im.rsti == 0; }

// Pipelining between the scans of one image (one launch for all dependency levels).  A 4K file of libjpeg's default script is
// ten scans in three levels, and the longest scan of every level is a luma scan (bytes: 276 k first stage, 348 k and 654 k
// refinement): level by level a file waits for their SUM.  Refinement scans cannot be cut into subsequences the way sequential
// scans are (lep_huffdec_simt.h) -- what a code means depends on which coefficients of ITS block are already non-zero, and a
// wavefront started in the middle does not know its block -- but a scan only needs the scans it follows to be AHEAD of it, not
// finished.  So every scan publishes the number of MCU rows it has completed (a release store after the rows' coefficients),
// and a scan that follows others waits, MCU row by MCU row, until all of them have passed the row it is about to enter: the
// file then takes about as long as its longest scan.  Waiting cannot deadlock: a scan only ever waits for scans of a lower
// index in the launch, and a workgroup takes the scan whose index is the TICKET it draws when it starts running (lep_gpu.hip): every
// scan of a lower index belongs to a workgroup that is running or done, whatever order the hardware starts workgroups in, and the
// lowest unfinished one never waits.
struct ProgDeps { int32_t dep[4]; };   // indices (into the launch) of the scans this one follows, -1 = none

struct ProgDecWave : HuffDecWave {
    const ProgDecScan* sc;
    uint32_t eobrun;
    int peobrun;
    const ProgDeps* deps = nullptr;   // pipelined launches only
    LV(int16_t, pf);                  // the next block of a refinement scan, requested one block ahead (ac_refine_block)
    int pf_dpos = -1, pf_cmp = 0;
    uint32_t* progress = nullptr;     // [scan of the launch]: MCU rows completed
    int self = 0;
    uint32_t ready = 0;               // MCU rows every scan this one follows is known to have completed
    static constexpr uint32_t kMaxPolls = 4u << 20;   // x ~2 us of sleep: about eight seconds

    WDEV void publish(uint32_t rows_done) {
        if (!progress) return;
#if LEP_ON_GPU
        __threadfence();   // (every lane: the rows' coefficients were stored lane-parallel)
        if (threadIdx.x == 0) __hip_atomic_store(progress + self, rows_done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
        progress[self] = rows_done;
#endif
    }
    // before the first block of MCU row `row` is touched
    WDEV void await(uint32_t row) {
        if (!progress || row < ready) return;
        for (uint32_t polls = 0;;) {
            uint32_t m = 0x7fffffffu;
            for (int i = 0; i < 4; ++i) {
                const int d = deps->dep[i];
                if (d < 0) continue;
#if LEP_ON_GPU
                const uint32_t v = __hip_atomic_load(progress + d, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
#else
                const uint32_t v = progress[d];
#endif
                m = v < m ? v : m;
            }
            ready = m;
            if (row < ready) return;
#if LEP_ON_GPU
            // (the safety net under the argument above: a scan that has polled for seconds gives up -- status 4, the file goes
            // to the host parser like any irregular one -- instead of holding its wave slot for ever)
            if (++polls > kMaxPolls) { status = 4; ready = 0x7fffffffu; return; }
            __builtin_amdgcn_s_sleep(64);
#else
            return;   // (the emulation runs the scans one after the other, in launch order: never here)
#endif
        }
    }

    // next_mcuposn (jpgcoder.cc:5432-5456): 0 go on, 1 restart interval over, 2 scan over
    WDEV int next_noninterleaved(int cmp, int* dpos, int* rstw) const {
        const int bch = sc->t.bch[cmp], nch = sc->nch[cmp], bcv = sc->bcv[cmp], ncv = sc->ncv[cmp];
        ++*dpos;
        if (bch != nch && *dpos % bch == nch) *dpos += bch - nch;
        if (bcv != ncv && *dpos / bch == ncv) *dpos = bch * bcv;
        if (*dpos >= bch * bcv) return 2;
        if (sc->t.rsti > 0 && --*rstw == 0) return 1;
        return 0;
    }
    // skip_eobrun (jpgcoder.cc:5462-5500): the blocks an end-of-band run covers are passed as a whole; -1 = irregular
    WDEV int skip_run(int cmp, int* dpos, int* rstw) {
        if (!eobrun) return 0;
        const int bch = sc->t.bch[cmp], nch = sc->nch[cmp], bcv = sc->bcv[cmp], ncv = sc->ncv[cmp];
        if (sc->t.rsti > 0) {
            if ((int)eobrun > *rstw) return -1;
            *rstw -= (int)eobrun;
        }
        if (bch != nch) *dpos += (int)((((uint32_t)(*dpos % bch) + eobrun) / (uint32_t)nch) * (uint32_t)(bch - nch));
        if (bcv != ncv && *dpos / bch >= ncv) *dpos += (bcv - ncv) * bch;
        *dpos += (int)eobrun;
        eobrun = 0;
        if (*dpos == bch * bcv) return 2;
        if (*dpos > bch * bcv) return -1;
        if (sc->t.rsti > 0 && *rstw == 0) return 1;
        return 0;
    }

    // one-component scans, pipelined launches: a block row is entered (wait for the scans in front) / left (publish).  MCU row =
    // block row / vertical sampling factor of the component.
    WDEV void enter_row(int cmp, int dpos, int* row_end) {
        const int bch = sc->t.bch[cmp], br = dpos / bch;
        *row_end = (br + 1) * bch;
        await((uint32_t)(br / sc->t.vs[cmp]));
    }
    WDEV void leave_rows(int cmp, int dpos) {
        if (!progress) return;
        publish((uint32_t)((dpos / sc->t.bch[cmp]) / sc->t.vs[cmp]));   // block rows below dpos's are complete: so are the MCU rows below its
    }

    // Both AC block decoders keep the serial part to one step per CODE: what a code places goes into the block image in LDS
    // (zig-zag order) and a 64-bit "changed" mask; which already-non-zero positions it passes is a mask operation, their
    // correction bits one read; the coefficients that changed leave with ONE lane-parallel store at the end of the block
    // (lane = position; scans of the same level write different positions, so whole-block stores would race).
    WDEV void flush_block(int cmp, int dpos, uint64_t changed) {
        if (!changed) return;
        int16_t* dst = sc->t.blocks[cmp] + (int64_t)dpos * 64;
        LSYNC();
        LANES(l) if ((changed >> l) & 1ull) dst[sh->z2a[l]] = sh->blk[l];
        LSYNC();
    }

    // ---- AC first stage, one block (decode_ac_prg_fs): 0 ok, -1 irregular -------------------------------------------------------
    WDEV int ac_first_block(int cmp, int dpos) {
        const int from = sc->from, to = sc->to, sal = sc->sal;
        if (eobrun > 0) { --eobrun; return 0; }   // inside a run: the band of this block is zero (the frame starts zeroed)
        uint32_t bpos = vec((uint32_t)from);
        uint32_t last_s = vec(1);
        uint64_t changed = 0;
        int rc = 0;
#pragma nounroll
        while (ucond(bpos <= (uint32_t)to)) {
            uint32_t n = 0;
            const int hc = symbol_and_bits(2, false, &n);
            if (ucond(hc < 0)) { rc = -1; break; }
            const uint32_t l = ((uint32_t)hc >> 4) & 15u, r = (uint32_t)hc & 15u;
            if (ucond(l == 15u || r > 0u)) {
                if (ucond(l + bpos > (uint32_t)to)) { rc = -1; break; }
                bpos += l;
                if (ucond(r > 0u)) {
                    const uint32_t bp = uni(bpos);
                    const int16_t v = (int16_t)((uint16_t)devli(r, uni(n)) << sal);
                    LANES(ln) if (ln == 0) sh->blk[bp] = v;
                    changed |= 1ull << bp;
                }
                ++bpos;
                last_s = r;
            } else {
                // end of band, and of 2^l + n - 1 further blocks.  A run that follows a run the encoder had not filled up is
                // not what a canonical encoder writes (it would have written ONE longer run): host
                const uint32_t extra = l ? read(l) : 0u;
                if (ucond(last_s == 0u)) { rc = -1; break; }          // coded zeros in front of the end of band
                eobrun = uni(extra) + (1u << uni(l)) - 1u;
                if (ucond(bpos == (uint32_t)from) && peobrun > 0 && peobrun < sc->max_eobrun) { rc = -1; break; }
                peobrun = (int)eobrun + 1;
                flush_block(cmp, dpos, changed);
                return 0;
            }
        }
        if (!rc && ucond(last_s == 0u)) rc = -1;                      // the band ends in a coded zero
        flush_block(cmp, dpos, changed);
        peobrun = 0;
        return rc;
    }

    // the correction bits of the already-non-zero positions in `cm` (a mask of zig-zag positions, consumed in ascending order): one bit
    // each from the stream.  They are only COLLECTED here, in stream order (a block has at most 63 of them); which positions they
    // belong to is the mask of everything passed, and a position whose bit is 1 moves one step away from zero when the block is done
    // (apply_corrections) -- one lane-parallel pass per block instead of one per code: the last bit plane of a 4K luma scan is
    // 130 k blocks of ~6 codes, and the scan is one wavefront's dependent chain.
    uint64_t cbits = 0, cmask = 0;
    int ncb = 0;
    WDEV void correct(uint64_t cm) {
        cmask |= cm;
        int k = lepwave::popc64(cm);
#pragma nounroll
        while (k > 0) {
            const int take = k > 16 ? 16 : k;
            cbits = (cbits << take) | (uint64_t)uni(read((uint32_t)take));
            ncb += take; k -= take;
        }
    }
    WDEV void apply_corrections(uint64_t* changed) {
        if (!ncb) return;
        LV(int, hit);
        LANES(l) {
            int h = 0;
            if ((cmask >> l) & 1ull) {
                const int j = lepwave::popc64(cmask & ((1ull << l) - 1));
                h = (int)((cbits >> (ncb - 1 - j)) & 1ull);
                if (h) { const int old = sh->blk[l]; sh->blk[l] = (int16_t)(old + (int16_t)((uint16_t)(int16_t)(old > 0 ? 1 : -1) << sc->sal)); }
            }
            L(hit) = h;
        }
        *changed |= lepwave::wave_ballot(hit);
    }

    // ---- AC refinement, one block (decode_ac_prg_sa / decode_eobrun_sa) ---------------------------------------------------------------
    // the band's coefficients as they are (sh->blk, zig-zag order); every already non-zero position passed costs one correction
    // bit; a code places one new +-1 behind `z` zero positions
    WDEV int ac_refine_block(int cmp, int dpos) {
        const int from = sc->from, to = sc->to, sal = sc->sal;
        const int16_t* src = sc->t.blocks[cmp] + (int64_t)dpos * 64;
        LV(int, nzf);
        // The block's coefficients as the scans before left them.  A refinement scan is one wavefront and one dependent chain: the
        // trip to HBM for every block (130 k blocks in a 4K luma scan) stood on it -- the NEXT block of the row is requested while
        // this one is decoded (the scans this one follows have passed its whole block row: enter_row), and taken from the register
        // when the walk gets there.
        const bool have = pf_dpos == dpos && pf_cmp == cmp;
        LANES(l) {
            const int v = (l >= from && l <= to) ? (have ? (int)L(pf) : (int)src[sh->z2a[l]]) : 0;
            sh->blk[l] = (int16_t)v; L(nzf) = v != 0;
        }
        pf_dpos = -1;
        if ((dpos + 1) % sc->t.bch[cmp] != 0) {   // (not across a block row: the next row may not be final yet, or not the next block at all)
            LANES(l) L(pf) = (l >= from && l <= to) ? src[64 + sh->z2a[l]] : (int16_t)0;
            pf_dpos = dpos + 1; pf_cmp = cmp;
        }
        LSYNC();
        const uint64_t band = (to >= 63 ? ~0ull : ((1ull << (to + 1)) - 1)) & ~((1ull << from) - 1);
        const uint64_t nzm = lepwave::wave_ballot(nzf);   // already non-zero positions (zig-zag index = bit index)
        const uint64_t zm = band & ~nzm;                  // zero positions of the band
        cbits = 0; cmask = 0; ncb = 0;
        uint32_t bpos = (uint32_t)from;                   // scalar: everything it depends on is read through uni()
        uint32_t last_kind = 1;                           // 0: the last code was a ZRL (sixteen zeros)
        uint64_t changed = 0;
        int rc = 0;
        if (eobrun == 0) {
#pragma nounroll
            while (bpos <= (uint32_t)to) {
                uint32_t n = 0;
                const int hc = symbol_and_bits(2, false, &n);
                if (ucond(hc < 0)) { rc = -1; break; }
                const uint32_t l = ((uint32_t)hc >> 4) & 15u, r = (uint32_t)hc & 15u;
                if (l == 15u || r > 0u) {
                    if (r > 1u) { rc = -1; break; }
                    // the (l + 1)-th zero position at or after bpos takes the new coefficient (or, for a ZRL, nothing); the
                    // non-zero positions in front of it take correction bits
                    const uint64_t zrest = zm & ~((1ull << bpos) - 1);
                    if ((uint32_t)lepwave::popc64(zrest) < l + 1u) { rc = -1; break; }   // the walk would leave the band
                    uint64_t m = zrest;
                    for (uint32_t i = 0; i < l; ++i) m &= m - 1;
                    const uint32_t pos = (uint32_t)__builtin_ctzll(m);
                    const uint64_t passed = nzm & ~((1ull << bpos) - 1) & ((1ull << pos) - 1);
                    const int v = r ? (uni(n) ? 1 : -1) : 0;
                    correct(passed);
                    if (v) {
                        const int16_t nv = (int16_t)((uint16_t)(int16_t)v << sal);
                        LANES(ln) if (ln == 0) sh->blk[pos] = nv;
                        changed |= 1ull << pos;
                    }
                    bpos = pos + 1;
                    last_kind = r;
                } else {
                    const uint32_t extra = l ? read(l) : 0u;
                    if (last_kind == 0u) { rc = -1; break; }         // ZRL in front of the end of band: not canonical
                    eobrun = uni(extra) + (1u << l);
                    if (bpos == (uint32_t)from && peobrun > 0 && peobrun < sc->max_eobrun - 1) { rc = -1; break; }   // jpgcoder.cc:3229-3236
                    break;
                }
            }
            if (!rc && eobrun == 0 && last_kind == 0u) rc = -1;      // the band ends in a ZRL
        }
        if (!rc && eobrun > 0) {
            if (bpos <= (uint32_t)to) correct(nzm & ~((1ull << bpos) - 1) & band);   // the rest of the band: correction bits only
            --eobrun;
        }
        apply_corrections(&changed);
        flush_block(cmp, dpos, changed);
        peobrun = (int)eobrun;
        return rc;
    }

    // PIPE: compiled with the waiting / publishing code (the level-by-level kernel is compiled without: it is held to 64 VGPRs)
    template <bool PIPE = false>
    WDEV void run_scan(const ProgDecScan* scan, HuffDecShared* shared, HuffDecRow* rows_arena, const ProgDeps* follow = nullptr, uint32_t* rows_done = nullptr,
                       int index = 0) {
        sc = scan; img = &scan->t; sh = shared; status = 0;
        deps = follow; progress = PIPE ? rows_done : nullptr; self = index; ready = 0;
        if (PIPE && progress) {   // a scan that follows none never waits
            bool any = false;
            for (int i = 0; i < 4; ++i) any = any || deps->dep[i] >= 0;
            if (!any) ready = 0x7fffffffu;
        }
        LANES(l) {
            for (int i = l; i < 512; i += 64) sh->lut_ac[0][i] = img->lut[2][i];
            for (int i = l; i < 2 * 256; i += 64) {
                const uint16_t e = img->lut[i >> 8][(i & 255) * 2];
                (&sh->lut_dc[0][0])[i] = (e >> 8) <= 8 ? e : (uint16_t)0;
            }
            if (l < 32) { (&sh->maxcode[0][0])[l] = (&img->maxcode[0][0])[l]; (&sh->valoff[0][0])[l] = (&img->valoff[0][0])[l]; }
            for (int i = l; i < 4 * 256; i += 64) (&sh->longsym[0][0])[i] = (&img->longsym[0][0])[i];
            sh->z2a[l] = kZ2A[l];
            sh->blk[l] = 0;
        }
        LSYNC();
        hi = vec(0); lo = vec(0); navail = 0; bitpos = vec(0);
        start_reader(0);
        refill(); refill();
        HuffDecRow* rows = rows_arena + img->rows_off;
        int lastdc[4] = {0, 0, 0, 0};
        int padbit = -1;
        const int rsti = img->rsti, mcuh = img->mcuh;
        const bool dc = scan->to == 0;
        const int sal = scan->sal;
        int cmp = scan->cmp[0], csc = 0, sub = 0, dpos = 0, mcu = 0;
        int row_end = 0;   // one-component scans: the block position at which the current block row ends (0: not entered yet)
        bool do_row = scan->want_rows != 0;
        for (;;) {   // one restart interval per iteration
            lastdc[0] = lastdc[1] = lastdc[2] = lastdc[3] = 0;
            int sta = 0, rstw = rsti;
            eobrun = 0; peobrun = 0;
            if (dc && scan->cmpc > 1) {            // interleaved DC scan (the usual first scan)
                while (sta == 0) {
                    if (do_row) {
                        const uint32_t bp = uni(bitpos);
                        const int r = mcu / mcuh;
                        LANES(l) if (l == 0) { rows[r].bitpos = bp; for (int c = 0; c < 4; ++c) rows[r].last_dc[c] = (int16_t)lastdc[c]; rows[r].aux = 0; }
                        do_row = false;
                    }
                    if (PIPE) { await((uint32_t)(mcu / mcuh)); if (status) { sta = -1; break; } }
                    int16_t* dst = img->blocks[cmp] + (int64_t)dpos * 64 + 49;
                    if (scan->sah == 0) {
                        uint32_t n = 0;
                        const int hc = symbol_and_bits(scan->tbl[csc] & 1, true, &n);
                        if (ucond(hc < 0)) { sta = -1; break; }
                        const int v = (int16_t)(devli((uint32_t)hc & 255u, uni(n)) + lastdc[cmp]);
                        lastdc[cmp] = v;
                        LANES(l) if (l == 0) *dst = (int16_t)((uint16_t)v << sal);
                    } else {
                        const uint32_t bit = uni(read(1));
                        LANES(l) if (l == 0) *dst = (int16_t)(*dst + (int16_t)(bit << sal));
                    }
                    // next_mcupos (jpgcoder.cc:5402-5430)
                    const int old_mcu = mcu;
                    if (++sub >= scan->mbs[cmp]) {
                        sub = 0;
                        if (++csc >= scan->cmpc) {
                            csc = 0; cmp = scan->cmp[0]; ++mcu;
                            if (mcu >= img->mcuc) sta = 2;
                            else if (rsti > 0 && --rstw == 0) sta = 1;
                        } else cmp = scan->cmp[csc];
                    }
                    {
                        const uint32_t hs = (uint32_t)img->hs[cmp], vs = (uint32_t)img->vs[cmp], m = (uint32_t)mcu, sb = (uint32_t)sub, mh = (uint32_t)mcuh;
                        if (vs > 1) dpos = (int)(((m / mh) * vs + sb / hs) * (uint32_t)img->bch[cmp] + (m % mh) * hs + sb % hs);
                        else if (hs > 1) dpos = (int)(m * (uint32_t)scan->mbs[cmp] + sb);
                        else dpos = mcu;
                    }
                    if (old_mcu != mcu && mcu % mcuh == 0) {
                        if (scan->want_rows) do_row = true;
                        if (PIPE) { LSYNC(); publish((uint32_t)(mcu / mcuh)); }
                    }
                    if (ucond(bitpos > img->scan_len * 8u)) { sta = -1; break; }
                }
            } else if (dc) {                       // DC scan of one component
                while (sta == 0) {
                    if (do_row) {
                        const uint32_t bp = uni(bitpos);
                        const int r = dpos / img->bch[cmp];
                        LANES(l) if (l == 0) { rows[r].bitpos = bp; for (int c = 0; c < 4; ++c) rows[r].last_dc[c] = (int16_t)lastdc[c]; rows[r].aux = 0; }
                        do_row = false;
                    }
                    if (PIPE && dpos >= row_end) { enter_row(cmp, dpos, &row_end); if (status) { sta = -1; break; } }
                    int16_t* dst = img->blocks[cmp] + (int64_t)dpos * 64 + 49;
                    if (scan->sah == 0) {
                        uint32_t n = 0;
                        const int hc = symbol_and_bits(scan->tbl[0] & 1, true, &n);
                        if (ucond(hc < 0)) { sta = -1; break; }
                        const int v = (int16_t)(devli((uint32_t)hc & 255u, uni(n)) + lastdc[cmp]);
                        lastdc[cmp] = v;
                        LANES(l) if (l == 0) *dst = (int16_t)((uint16_t)v << sal);
                    } else {
                        const uint32_t bit = uni(read(1));
                        LANES(l) if (l == 0) *dst = (int16_t)(*dst + (int16_t)(bit << sal));
                    }
                    sta = next_noninterleaved(cmp, &dpos, &rstw);
                    if (scan->want_rows && cmp == 0 && dpos % img->bch[cmp] == 0) do_row = true;
                    if (PIPE && dpos >= row_end) { LSYNC(); leave_rows(cmp, dpos); }
                    if (ucond(bitpos > img->scan_len * 8u)) { sta = -1; break; }
                }
            } else {                               // AC scan of one component
                while (sta == 0) {
                    if (PIPE && dpos >= row_end) { enter_row(cmp, dpos, &row_end); if (status) { sta = -1; break; } }
                    const int rc = scan->sah == 0 ? ac_first_block(cmp, dpos) : ac_refine_block(cmp, dpos);
                    if (rc < 0) { sta = -1; break; }
                    if (scan->sah == 0) sta = skip_run(cmp, &dpos, &rstw);
                    if (sta == 0) sta = next_noninterleaved(cmp, &dpos, &rstw);
                    if (PIPE && dpos >= row_end) leave_rows(cmp, dpos);   // (flush_block has ordered the lanes' stores)
                    if (ucond(bitpos > img->scan_len * 8u)) { sta = -1; break; }
                }
                // a run that reaches past the end of its restart interval or scan (the reference tolerates it in the refinement
                // stage and complains in the first): not canonical either way
                if (sta > 0 && eobrun > 0) sta = -1;
            }
            if (sta == -1) { if (!PIPE || !status) status = 1; break; }
            const int got = unpad(padbit == -1 ? 255 : padbit);
            if (padbit == -1) padbit = (int8_t)got;
            else if (padbit != got) { status = 3; break; }
            if (sta == 2) break;
        }
        if (!status && uni(bitpos) != img->scan_len * 8u) status = 2;   // bytes left over, or missing
        if (PIPE) publish(0x7fffffffu);   // whatever happened: nobody waits for this scan any more
        const uint32_t bp = uni(bitpos);
        HuffDecRow* fin = rows_arena + scan->result_off;
        LANES(l) if (l == 0) {
            fin->bitpos = bp;
            for (int c = 0; c < 4; ++c) fin->last_dc[c] = (int16_t)lastdc[c];
            fin->aux = (padbit & 255) | (status << 8);
        }
    }
};

// host side: which scans of the launch (already in launch order: dependency level by dependency level) each scan follows.
// Scan j follows an earlier scan i of the same file (same frame) when they share a component and their bands meet; of those,
// the ones another of them follows in turn are implied.  false: some scan follows more than four others (no pipelining then).
// `order[k]` = the scan's place in its file's scan sequence (the caller's order), so that "earlier" means what it means in the file.
inline bool prog_scan_deps(const ProgDecScan* scans, const int* order, int n, ProgDeps* out) {
    auto meets = [&](int i, int j) {
        const ProgDecScan &a = scans[i], &b = scans[j];
        if (a.t.blocks[0] != b.t.blocks[0]) return false;
        if (a.from > b.to || b.from > a.to) return false;
        for (int x = 0; x < a.cmpc; ++x)
            for (int y = 0; y < b.cmpc; ++y)
                if (a.cmp[x] == b.cmp[y]) return true;
        return false;
    };
    // the scans of one file are few (ten for libjpeg's script): bucket the launch by frame, compare inside the buckets
    struct Ref { const void* key; int idx; };
    std::vector<Ref> refs((size_t)n);
    for (int i = 0; i < n; ++i) refs[(size_t)i] = Ref{(const void*)scans[i].t.blocks[0], i};
    std::stable_sort(refs.begin(), refs.end(), [](const Ref& a, const Ref& b) { return a.key < b.key; });
    bool ok = true;
    for (int b0 = 0; b0 < n && ok;) {
        int b1 = b0;
        while (b1 < n && refs[b1].key == refs[b0].key) ++b1;
        for (int x = b0; x < b1 && ok; ++x) {
            const int j = refs[x].idx;
            int cand[64], nc = 0;
            for (int y = b0; y < b1; ++y) {
                const int i = refs[y].idx;
                if (i != j && order[i] < order[j] && meets(i, j)) { if (nc < 64) cand[nc++] = i; else ok = false; }
            }
            int nd = 0;
            for (int d = 0; d < 4; ++d) out[j].dep[d] = -1;
            for (int u = 0; u < nc && ok; ++u) {
                bool implied = false;
                for (int v = 0; v < nc; ++v)
                    if (v != u && order[cand[u]] < order[cand[v]] && meets(cand[u], cand[v])) { implied = true; break; }
                if (implied) continue;
                if (cand[u] >= j) ok = false;          // a scan may only wait for scans in front of it in the launch
                else if (nd < 4) out[j].dep[nd++] = cand[u];
                else ok = false;
            }
        }
        b0 = b1;
    }
    return ok;
}

}  // namespace lephuff
