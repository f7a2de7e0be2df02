// jpeg_progressive.cc -- progressive JPEG scans on the host (BASELINE.json configs[4]): Huffman decode of DC / AC first
// and refinement scans into the coefficient frame, and the re-encode that restores the original scan bytes.  The GPU hot
// path (the arithmetic coder over the finished coefficient frame) is identical for progressive files; only this
// Huffman side differs.  Behaviour follows the reference's packJPG-derived scan coders:
//   decode: decode_jpeg's scan loop            src/lepton/jpgcoder.cc:2975-3260
//           decode_dc_prg_fs / _sa, decode_ac_prg_fs / _sa, decode_eobrun_sa, skip_eobrun   :4968-5335, :5462-5500
//   encode: recode_jpeg                         src/lepton/jpgcoder.cc:3309-3716
//           encode_dc_prg_*, encode_ac_prg_fs / _sa, encode_eobrun, encode_crbits            :4991-5400
//   output: merge_jpeg_streaming (scan bytes + FF00 stuffing + RSTn at the recorded positions)  :2562-2730
// Successive approximation keeps every coefficient at full scale in the frame: a first-stage scan stores value << Al,
// a refinement scan adds bit << Al (jpgcoder.cc:3001-3006, 3178-3182, 3243-3247).
#include <algorithm>
#include <cstring>

#include "jpeg_bits.h"
#include "lep_container.h"

namespace lep {

int next_mcupos(const JpegFile& jf, int* mcu, int* cmp, int* csc, int* sub, int* dpos, int* rstw, int cs_cmpc);
int next_mcuposn(const JpegFile& jf, int cmp, int* dpos, int* rstw);
Handoff make_handoff_public(BitReader& br, const JpegFile& jf, int mcu_y, const int lastdc[4], int luma_mul);

namespace {

inline unsigned envli(int s, int v) { return (unsigned)((v > 0) ? v : (v - 1) + (1 << s)) & ((1u << s) - 1); }
inline int blen16(unsigned v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }
inline int fdiv2(int v, int p) { return v < 0 ? -((-v) >> p) : (v >> p); }

// ---- decoding one block of a scan ----------------------------------------------------------------------------------
// AC first stage, band from..to of one block: the band's coefficients, or the block's turn inside an end-of-band run.  Returns the
// position behind the band's last coded coefficient (from = nothing coded), -1 for bits that are no code or a run that leaves the band.
int decode_ac_first(BitReader& br, const HuffTable& ac, int16_t* blk, unsigned* eobrun, int from, int to) {
    memset(blk + from, 0, (size_t)(to - from + 1) * sizeof blk[0]);
    if (*eobrun > 0) {
        --*eobrun;
        return from;
    }
    for (int at = from; at <= to;) {
        const int sym = next_huffcode(br, ac);
        if (sym < 0) return -1;
        const int run = sym >> 4 & 15, category = sym & 15;
        if (category == 0 && run < 15) {                   // end of band for this block and 2^run - 1 + (run extra bits) blocks after it
            *eobrun = (unsigned)((1 << run) + (int)br.read(run)) - 1;
            return at;
        }
        const int bits = (int)br.read(category);
        at += run;
        if (at > to) return -1;
        blk[at++] = (int16_t)extend(category, bits);
    }
    return to + 1;
}

// ---- AC refinement (T.81 G.1.2.3) the way the GPU scan coders do it (lep_huffprogdec.h): the band as two position masks ----------
// A refinement scan says two things about a band: where the NEW +-1 coefficients go (a code = "skip z positions that are still zero,
// then place one"), and one correction bit for every position that is ALREADY non-zero, in the order the walk passes them.  Which
// positions are zero does not change while a block is decoded -- a position is never visited twice -- so the block is turned into a
// mask of its zero positions and a mask of its non-zero ones once, a code's target is the (z + 1)-th set bit of the zero mask from the
// cursor on, and the correction bits of the stretch it passes are read in one go.  On return the band holds the DELTAS (new
// coefficient, or +-correction bit; the caller scales and adds them), as the reference's block decoders leave it
// (jpgcoder.cc:5159-5230): same bits consumed, same refusals -- a code that would walk out of the band, a magnitude category other
// than 1, a missing code.
struct RefineBand {
    uint64_t zero = 0, nonzero = 0, negative = 0;   // bit p = position p of the band's block
    int from, to;
    RefineBand(const int16_t* blk, int from_, int to_) : from(from_), to(to_) {
        for (int p = from; p <= to; ++p) {
            if (blk[p] == 0) zero |= 1ull << p;
            else { nonzero |= 1ull << p; if (blk[p] < 0) negative |= 1ull << p; }
        }
    }
    static uint64_t at_or_above(int p) { return p >= 64 ? 0ull : ~0ull << p; }
    static uint64_t below(int p) { return p >= 64 ? ~0ull : (1ull << p) - 1; }
    // correction bits of the non-zero positions in `m`, lowest position first; a position's delta is its bit with the coefficient's sign
    void correct(BitReader& br, int16_t* blk, uint64_t m) const {
        while (m) {
            uint64_t part = m;
            int k = __builtin_popcountll(m);
            if (k > 24) { part = 0; uint64_t r = m; for (int i = 0; i < 24; ++i) { part |= r & (0 - r); r &= r - 1; } k = 24; }
            const unsigned bits = br.read(k);   // first position = most significant of the k bits
            int j = 0;
            for (uint64_t r = part; r; r &= r - 1, ++j) {
                const int p = __builtin_ctzll(r);
                const int n = (int)((bits >> (k - 1 - j)) & 1u);
                blk[p] = (int16_t)(((negative >> p) & 1) ? -n : n);
            }
            m &= ~part;
        }
    }
};

int decode_ac_refine(BitReader& br, const HuffTable& ac, int16_t* blk, unsigned* eobrun, int from, int to) {
    const RefineBand band(blk, from, to);
    int cursor = from, eob = to;
    if (*eobrun == 0) {
        while (cursor <= to) {
            const int hc = next_huffcode(br, ac);
            if (hc < 0) return -1;
            const int run = (hc >> 4) & 15, cat = hc & 15;
            if (run != 15 && cat == 0) {   // end of band, and of 2^run + extra - 1 further blocks
                eob = cursor;
                *eobrun = br.read(run) + (1u << run);
                break;
            }
            if (cat > 1) return -1;
            const int v = cat == 0 ? 0 : (br.read(1) == 0 ? -1 : 1);   // the sign bit stands in front of the correction bits of this stretch
            // target: the (run + 1)-th zero position at or after the cursor
            uint64_t z = band.zero & RefineBand::at_or_above(cursor);
            if (__builtin_popcountll(z) <= run) {
                // the walk would leave the band.  The reference's walk (jpgcoder.cc:5192-5207) only notices at the band's last position:
                // by then it has read a correction bit for every non-zero position from the cursor on and left the +-bit there.  A
                // file cut inside a refinement scan is ACCEPTED with exactly this state (eof turns the -1 into "scan done" and the
                // deltas are added to the frame), so the bits are consumed and the deltas left the same way before refusing.
                band.correct(br, blk, band.nonzero & RefineBand::at_or_above(cursor));
                return -1;
            }
            for (int i = 0; i < run; ++i) z &= z - 1;
            const int target = __builtin_ctzll(z);
            band.correct(br, blk, band.nonzero & RefineBand::at_or_above(cursor) & RefineBand::below(target));
            blk[target] = (int16_t)v;
            cursor = target + 1;
        }
    }
    if (*eobrun > 0) {   // inside a run (this block's own end-of-band code included): the rest of the band takes correction bits only
        band.correct(br, blk, band.nonzero & RefineBand::at_or_above(cursor));
        --*eobrun;
    }
    return eob;
}

int decode_eobrun_refine(BitReader& br, int16_t* blk, unsigned* eobrun, int from, int to) {
    const RefineBand band(blk, from, to);
    band.correct(br, blk, band.nonzero);
    --*eobrun;
    return 0;
}

// blocks covered by an end-of-band run are skipped as a whole (jpgcoder.cc:5462-5500).  The arithmetic is the reference's to the
// letter because damaged streams depend on it: a run that reaches into the padding rows of a non-interleaved component is tested
// against the row count BEFORE the run is added (so it lands on padding blocks and the scan goes on), and the fuzz harnesses hold
// exactly that against the reference binary -- a tidier (row, column) form differs on 8 % of the overshooting runs.
int skip_eobrun(const JpegFile& jf, int cmp, int* dpos, int* rstw, unsigned* eobrun) {
    if (*eobrun == 0) return 0;
    const Component& k = jf.comp[cmp];
    if (jf.rsti > 0) {
        if ((int)*eobrun > *rstw) return -1;
        *rstw -= (int)*eobrun;
    }
    if (k.bch != k.nch) *dpos += (int)((((unsigned)(*dpos % k.bch) + *eobrun) / (unsigned)k.nch) * (unsigned)(k.bch - k.nch));
    if (k.bcv != k.ncv && *dpos / k.bch >= k.ncv) *dpos += (k.bcv - k.ncv) * k.bch;
    *dpos += (int)*eobrun;
    *eobrun = 0;
    if (*dpos == k.bc) return 2;
    if (*dpos > k.bc) return -1;
    if (jf.rsti > 0 && *rstw == 0) return 1;
    return 0;
}

}  // namespace

// One restart interval of one progressive scan (the caller, decode_scans, owns the interval loop, the pad-bit check and
// the scan bookkeeping).  Returns an ExitCode; *sta_io = 0 / 1 restart / 2 scan done / -1 decode error.
int decode_progressive_scan(JpegFile* jf, BitReader& br, int* lastdc, int* sta_io, int* cmp_io, int* dpos_io, int* mcu_io,
                            int* csc_io, int* sub_io, int* rstw_io, unsigned* eobrun_io, int* peobrun_io, bool* do_handoff) {
    int sta = *sta_io, cmp = *cmp_io, dpos = *dpos_io, mcu = *mcu_io, csc = *csc_io, sub = *sub_io, rstw = *rstw_io;
    unsigned eobrun = *eobrun_io;
    int peobrun = *peobrun_io;
    const int luma_mul = jf->comp[0].bcv / jf->mcuv;
    const int from = jf->cs_from, to = jf->cs_to, sal = jf->cs_sal;
    int16_t blk[64];
    auto dc_of = [&](int c, int d) -> int16_t& { return jf->plane[c][(size_t)d * 64 + kZigzagToAligned[0]]; };
    auto note_max = [&]() { if (!br.eof) jf->max_dpos[cmp] = std::max(dpos, jf->max_dpos[cmp]); };

    if (jf->cs_cmpc > 1) {   // interleaved: DC only
        if (jf->cs_sah == 0) {
            while (sta == 0) {
                if (*do_handoff) { jf->rows.push_back(make_handoff_public(br, *jf, mcu / jf->mcuh, lastdc, luma_mul)); *do_handoff = false; }
                note_max();
                const HuffTable& t = jf->htab[0][jf->comp[cmp].dc_tbl];
                const int hc = next_huffcode(br, t);
                int diff = 0;
                if (hc < 0) sta = -1;
                else diff = extend(hc & 255, (int)br.read(hc & 255));
                const int16_t v = (int16_t)((int16_t)diff + lastdc[cmp]);
                lastdc[cmp] = v;
                dc_of(cmp, dpos) = (int16_t)((uint16_t)v << sal);
                const int old_mcu = mcu;
                if (sta != -1) sta = next_mcupos(*jf, &mcu, &cmp, &csc, &sub, &dpos, &rstw, jf->cs_cmpc);
                if (mcu % jf->mcuh == 0 && old_mcu != mcu) *do_handoff = true;
                if (br.eof) { sta = 2; break; }
            }
        } else {
            while (sta == 0) {
                note_max();
                const int bit = (int)br.read(1);
                dc_of(cmp, dpos) = (int16_t)(dc_of(cmp, dpos) + (int16_t)(bit << sal));
                sta = next_mcupos(*jf, &mcu, &cmp, &csc, &sub, &dpos, &rstw, jf->cs_cmpc);
                if (br.eof) { sta = 2; break; }
            }
        }
    } else if (to == 0) {    // non-interleaved DC
        if (jf->cs_sah == 0) {
            while (sta == 0) {
                if (*do_handoff) { jf->rows.push_back(make_handoff_public(br, *jf, dpos / jf->comp[cmp].bch, lastdc, luma_mul)); *do_handoff = false; }
                note_max();
                const HuffTable& t = jf->htab[0][jf->comp[cmp].dc_tbl];
                const int hc = next_huffcode(br, t);
                int diff = 0;
                if (hc < 0) sta = -1;
                else diff = extend(hc & 255, (int)br.read(hc & 255));
                const int16_t v = (int16_t)((int16_t)diff + lastdc[cmp]);
                lastdc[cmp] = v;
                dc_of(cmp, dpos) = (int16_t)((uint16_t)v << sal);
                if (sta != -1) sta = next_mcuposn(*jf, cmp, &dpos, &rstw);
                if (cmp == 0 && dpos % jf->comp[cmp].bch == 0) *do_handoff = true;
                if (br.eof) { sta = 2; break; }
            }
        } else {
            while (sta == 0) {
                note_max();
                const int bit = (int)br.read(1);
                dc_of(cmp, dpos) = (int16_t)(dc_of(cmp, dpos) + (int16_t)(bit << sal));
                sta = next_mcuposn(*jf, cmp, &dpos, &rstw);
                if (br.eof) { sta = 2; break; }
            }
        }
    } else {                 // non-interleaved AC
        const HuffTable& t = jf->htab[1][jf->comp[cmp].ac_tbl];
        if (jf->cs_sah == 0) {
            while (sta == 0) {
                note_max();
                const int eob = decode_ac_first(br, t, blk, &eobrun, from, to);
                if (eob == from && eobrun > 0 && peobrun > 0 && peobrun < t.max_eobrun - 1) jf->warn = std::max(jf->warn, 1);
                int16_t* dst = jf->plane[cmp] + (size_t)dpos * 64;
                for (int b = from; b < eob; ++b) dst[kZigzagToAligned[b]] = (int16_t)((uint16_t)blk[b] << sal);
                if (eob < 0) sta = -1;
                else sta = skip_eobrun(*jf, cmp, &dpos, &rstw, &eobrun);
                if (sta == 0) sta = next_mcuposn(*jf, cmp, &dpos, &rstw);
                if (br.eof) { sta = 2; break; }
            }
        } else {
            while (sta == 0) {
                int16_t* dst = jf->plane[cmp] + (size_t)dpos * 64;
                for (int b = from; b <= to; ++b) blk[b] = dst[kZigzagToAligned[b]];
                int eob;
                if (eobrun == 0) {
                    note_max();
                    eob = decode_ac_refine(br, t, blk, &eobrun, from, to);
                    if (eob == from && eobrun > 0 && peobrun > 0 && peobrun < t.max_eobrun - 1) jf->warn = std::max(jf->warn, 1);
                } else {
                    note_max();
                    eob = decode_eobrun_refine(br, blk, &eobrun, from, to);
                }
                peobrun = (int)eobrun;
                for (int b = from; b <= to; ++b) dst[kZigzagToAligned[b]] = (int16_t)(dst[kZigzagToAligned[b]] + (int16_t)((uint16_t)blk[b] << sal));
                if (eob < 0) sta = -1;
                else sta = next_mcuposn(*jf, cmp, &dpos, &rstw);
                if (br.eof) { sta = 2; break; }
            }
        }
    }
    *sta_io = sta; *cmp_io = cmp; *dpos_io = dpos; *mcu_io = mcu; *csc_io = csc; *sub_io = sub; *rstw_io = rstw;
    *eobrun_io = eobrun; *peobrun_io = peobrun;
    return 0;
}

// ---- re-encoding ----------------------------------------------------------------------------------------------------
namespace {

struct ScanWriter {
    BitWriter w;
    std::vector<uint8_t> corr;   // correction bits waiting for the next code (abytewriter storw)
    void put(unsigned v, int n) { w.put(v, n); }
    void code(const HuffTable& t, int sym) { w.put(t.cval[sym & 255], t.clen[sym & 255]); }
    void flush_corr() { for (uint8_t b : corr) w.put(b, 1); corr.clear(); }
    void eobrun(const HuffTable& ac, unsigned* run) {
        if (*run == 0) return;
        while (*run > (unsigned)ac.max_eobrun) {
            code(ac, 0xE0);
            put((unsigned)(32767 - (1 << 14)), 14);
            *run -= (unsigned)ac.max_eobrun;
        }
        int s = blen16(*run & 0xffff);
        if (s) --s;
        code(ac, s << 4);
        put(*run - (1u << s), s);
        *run = 0;
    }
};

int encode_ac_first(ScanWriter& sw, const HuffTable& ac, const int16_t* blk, unsigned* eobrun, int from, int to) {
    int z = 0;
    for (int bpos = from; bpos <= to; ++bpos) {
        const int t = blk[bpos];
        if (t != 0) {
            sw.eobrun(ac, eobrun);
            while (z >= 16) { sw.code(ac, 0xF0); z -= 16; }
            const int s = blen16((unsigned)(t > 0 ? t : -t) & 0xffff);
            sw.code(ac, (z << 4) + s);
            sw.put(envli(s, t), s);
            z = 0;
        } else ++z;
    }
    if (z > 0) {
        ++*eobrun;
        if (*eobrun == (unsigned)ac.max_eobrun) sw.eobrun(ac, eobrun);
        return 1 + to - z;
    }
    return 1 + to;
}

int encode_ac_refine(ScanWriter& sw, const HuffTable& ac, const int16_t* blk, unsigned* eobrun, int from, int to) {
    int eob = from;
    for (int bpos = to; bpos >= from; --bpos)
        if (blk[bpos] == 1 || blk[bpos] == -1) { eob = bpos + 1; break; }
    if (eob > from && *eobrun > 0) { sw.eobrun(ac, eobrun); sw.flush_corr(); }
    int z = 0, bpos = from;
    for (; bpos < eob; ++bpos) {
        const int t = blk[bpos];
        if (t == 0) {
            if (++z == 16) { sw.code(ac, 0xF0); sw.flush_corr(); z = 0; }
        } else if (t == 1 || t == -1) {
            sw.code(ac, (z << 4) + 1);
            sw.put(envli(1, t), 1);
            sw.flush_corr();
            z = 0;
        } else sw.corr.push_back((uint8_t)(t & 1));
    }
    for (; bpos <= to; ++bpos)
        if (blk[bpos] != 0) sw.corr.push_back((uint8_t)(blk[bpos] & 1));
    if (eob <= to) {
        ++*eobrun;
        if (*eobrun == (unsigned)ac.max_eobrun) { sw.eobrun(ac, eobrun); sw.flush_corr(); }
    }
    return eob;
}

}  // namespace

// recode_jpeg + merge_jpeg for files that are not a single interleaved sequential scan ('X' files)
int recode_progressive(LepFile* lf, std::vector<uint8_t>* result) {
    JpegFile& jf = lf->jpeg;
    const size_t max_file_size = lf->jpeg_size;
    // always_assert(max_file_size > grbs), both ints: where the reference first hands bytes to its output (merge_jpeg_streaming,
    // jpgcoder.cc:2570-2572, called after each restart interval of recode_jpeg) -- what is wrong with the first scan's tables
    // is reported before it
    const bool all_garbage = (int32_t)lf->jpeg_size <= (int32_t)jf.garbage.size();
    bool output_started = false;
    const uint8_t* h = jf.hdr.data();
    const size_t hdrs = jf.hdr.size();

    // 1. all scans into one un-stuffed byte string, remembering where scans start and restart markers go
    ScanWriter sw;
    sw.w.fillbit = (uint8_t)jf.padbit;
    std::vector<size_t> scnp, rstp;
    std::vector<size_t> scan_hdr_end;   // header position after each SOS
    size_t hpos = 0;
    int16_t blk[64];
    // A truncated file decoded on one thread: the reference's decoder runs only as far as its re-coder WAITS for it, and that
    // wait is clamped to the last block the JPEG held (wait_for_worker_on_dpos, uncompressed_components.hh:206-216) -- the first
    // block of each later row, which the stream does code (decode_row always takes a row's first block), is never decoded and
    // its re-coder reads zeros.  Invisible in an intact file (the byte bound cuts the output first); found by the byte-level
    // differential fuzz with a damaged stream.  (Several segments: worker threads decode their rows eagerly; nothing to emulate.)
    static const int16_t kNeverDecoded[64] = {0};
    const bool lazy_tail = jf.early_eof && lf->segs.size() == 1;
    auto block_at = [&](int c, int d) -> const int16_t* {
        return (lazy_tail && d >= jf.trunc_bc[c]) ? kNeverDecoded : jf.plane[c] + (size_t)d * 64;
    };
    for (;;) {
        uint8_t type = 0;
        while (type != 0xDA) {
            if (hpos >= hdrs) break;
            type = hpos + 1 < hdrs ? h[hpos + 1] : 0;
            const unsigned len = 2 + (((unsigned)(hpos + 2 < hdrs ? h[hpos + 2] : 0)) << 8) + (hpos + 3 < hdrs ? h[hpos + 3] : 0);
            if (type == 0xC4 || type == 0xDA || type == 0xDD)
                if (!parse_segment(&jf, type, len, (unsigned)std::min<size_t>(len, hdrs - hpos), h + hpos, false)) return jf.warn < 0 ? -jf.warn : EX_UNSUPPORTED_JPEG;   // parse_jfif_jpg: errorlevel 2
            hpos += len;
        }
        if (type != 0xDA) break;
        scan_hdr_end.push_back(hpos);   // (may lie behind the header: an SOS length field that reaches past it; the merge writes null bytes for the difference, jpgcoder.cc:2594-2598)
        scnp.push_back(sw.w.bytes.size());
        int cmp = jf.cs_cmp[0], csc = 0, mcu = 0, sub = 0, dpos = 0;
        const int from = jf.cs_from, to = jf.cs_to, sal = jf.cs_sal;
        for (;;) {   // one restart interval per iteration
            int lastdc[4] = {0, 0, 0, 0};
            int sta = 0, rstw = jf.rsti;
            unsigned eobrun = 0;
            auto dc_of = [&](int c, int d) -> int { return block_at(c, d)[kZigzagToAligned[0]]; };
            // one whole block of component cmp at dpos, sequential coding (encode_block_seq, jpgcoder.cc:5009-5066)
            auto sequential_block = [&]() {
                const int16_t* src = block_at(cmp, dpos);
                for (int b = 0; b < 64; ++b) blk[b] = src[kZigzagToAligned[b]];
                const int16_t dc = blk[0];
                blk[0] = (int16_t)(blk[0] - lastdc[cmp]);
                lastdc[cmp] = dc;
                const HuffTable& dct = jf.htab[0][jf.comp[cmp].dc_tbl];
                const HuffTable& act = jf.htab[1][jf.comp[cmp].ac_tbl];
                int t = blk[0], s = blen16((unsigned)(t > 0 ? t : -t) & 0xffff);
                sw.code(dct, s); sw.put(envli(s, t), s);
                int end = 63, z = 0;
                while (end && !blk[end]) --end;
                for (int b = 1; b <= end; ++b) {
                    t = blk[b];
                    if (!t) { ++z; continue; }
                    s = blen16((unsigned)(t > 0 ? t : -t) & 0xffff);
                    while (z & 0xf0) { sw.code(act, 0xF0); z -= 16; }
                    sw.code(act, ((z & 0xf) << 4) + s);
                    sw.put(envli(s, t), s);
                    z = 0;
                }
                if (end != 63) sw.code(act, 0);
            };
            if (jf.cs_cmpc > 1 && jf.jpegtype == 1) {
                // sequential multi-scan file, this scan interleaves several components (e.g. luma alone, then Cb + Cr together;
                // recode_jpeg's sequential MCU loop, jpgcoder.cc:3461-3486)
                while (sta == 0) {
                    sequential_block();
                    sta = next_mcupos(jf, &mcu, &cmp, &csc, &sub, &dpos, &rstw, jf.cs_cmpc);
                }
            } else if (jf.cs_cmpc > 1) {
                if (jf.cs_sah == 0) {
                    while (sta == 0) {
                        const int tmp = dc_of(cmp, dpos) >> sal;
                        const int d = (int16_t)(tmp - lastdc[cmp]);
                        lastdc[cmp] = tmp;
                        const int s = blen16((unsigned)(d > 0 ? d : -d) & 0xffff);
                        sw.code(jf.htab[0][jf.comp[cmp].dc_tbl], s);
                        sw.put(envli(s, d), s);
                        sta = next_mcupos(jf, &mcu, &cmp, &csc, &sub, &dpos, &rstw, jf.cs_cmpc);
                    }
                } else {
                    while (sta == 0) {
                        sw.put((unsigned)((dc_of(cmp, dpos) >> sal) & 1), 1);
                        sta = next_mcupos(jf, &mcu, &cmp, &csc, &sub, &dpos, &rstw, jf.cs_cmpc);
                    }
                }
            } else if (jf.jpegtype == 1) {
                while (sta == 0) {   // sequential, one component per scan
                    sequential_block();
                    sta = next_mcuposn(jf, cmp, &dpos, &rstw);
                }
            } else if (to == 0) {
                if (jf.cs_sah == 0) {
                    while (sta == 0) {
                        const int tmp = dc_of(cmp, dpos) >> sal;
                        const int d = (int16_t)(tmp - lastdc[cmp]);
                        lastdc[cmp] = tmp;
                        const int s = blen16((unsigned)(d > 0 ? d : -d) & 0xffff);
                        sw.code(jf.htab[0][jf.comp[cmp].dc_tbl], s);
                        sw.put(envli(s, d), s);
                        sta = next_mcuposn(jf, cmp, &dpos, &rstw);
                    }
                } else {
                    while (sta == 0) {
                        sw.put((unsigned)((dc_of(cmp, dpos) >> sal) & 1), 1);
                        sta = next_mcuposn(jf, cmp, &dpos, &rstw);
                    }
                }
            } else {
                const HuffTable& act = jf.htab[1][jf.comp[cmp].ac_tbl];
                while (sta == 0) {
                    const int16_t* src = block_at(cmp, dpos);
                    for (int b = from; b <= to; ++b) blk[b] = (int16_t)fdiv2(src[kZigzagToAligned[b]], sal);
                    if (jf.cs_sah == 0) encode_ac_first(sw, act, blk, &eobrun, from, to);
                    else encode_ac_refine(sw, act, blk, &eobrun, from, to);
                    sta = next_mcuposn(jf, cmp, &dpos, &rstw);
                }
                sw.eobrun(act, &eobrun);
                if (jf.cs_sah != 0) sw.flush_corr();
            }
            sw.w.pad((uint8_t)jf.padbit);
            if (sta < 0) return EX_CODING_ERROR;
            if (!output_started) { output_started = true; if (all_garbage) return EX_ASSERTION_FAILURE; }
            if (sta == 2) break;
            if (sta == 1 && jf.rsti > 0) rstp.push_back(sw.w.bytes.size() - 1);
        }
    }
    if (all_garbage) return EX_ASSERTION_FAILURE;
    // no scan at all (the walk left the header without meeting an SOS, e.g. behind a segment length that reaches past it): the
    // reference's scan table is still empty when it stores the last position -- "out of memory error", errorlevel 2
    // (jpgcoder.cc:3708-3714)
    if (scan_hdr_end.empty()) return EX_UNSUPPORTED_JPEG;
    scnp.push_back(sw.w.bytes.size());
    const std::vector<uint8_t>& huff = sw.w.bytes;

    // 2. merge: SOI, then per scan the header part up to its SOS, the scan bytes (FF00 stuffing, RSTn after the recorded
    //    bytes while the scan's marker budget lasts), misplaced RSTn at the scan end; then the rest of the header, garbage
    std::vector<uint8_t> out;
    out.reserve(std::min<size_t>(max_file_size, (size_t)128 << 20) + 16);
    const size_t bound = max_file_size - jf.garbage.size();
    auto put = [&](uint8_t b) { if (out.size() < bound) out.push_back(b); };
    if (lf->has_prefix) for (uint8_t b : lf->prefix_garbage) put(b);
    if (lf->embedded || !lf->has_prefix) { put(0xFF); put(0xD8); }
    size_t hp = 0, rpos = 0;
    for (size_t scan = 0; scan < scan_hdr_end.size(); ++scan) {
        for (size_t i = hp; i < scan_hdr_end[scan]; ++i) { if (out.size() >= bound) break; put(i < hdrs ? h[i] : (uint8_t)0); }
        hp = scan_hdr_end[scan];
        unsigned cpos = 0, nrst = 0;
        for (size_t i = scnp[scan]; i < scnp[scan + 1]; ++i) {
            put(huff[i]);
            if (huff[i] == 0xFF) put(0x00);
            if (rpos < rstp.size() && i == rstp[rpos]) {
                const bool ok = !lf->rst_cnt_set || (jf.rst_cnt.size() > scan && nrst < jf.rst_cnt[scan]);
                if (ok) { put(0xFF); put((uint8_t)(0xD0 + (cpos & 7))); ++rpos; ++cpos; ++nrst; }
            }
        }
        if (scan < jf.rst_err.size())
            for (unsigned k = 0; k < jf.rst_err[scan]; ++k) { put(0xFF); put((uint8_t)(0xD0 + (cpos & 7))); ++cpos; }
    }
    for (size_t i = hp; i < hdrs; ++i) put(h[i]);
    for (size_t i = 0; i < jf.garbage.size() && out.size() < max_file_size; ++i) out.push_back(jf.garbage[i]);
    result->swap(out);
    return 0;
}

// ---- the same with the scans coded on the GPU (lep_huffprog.h) ---------------------------------------------------------------------
int recode_progressive_prepare(LepFile* lf, ProgPlan* plan) {
    plan->gpu_ok = false;
    plan->scans.clear(); plan->scan_hdr_end.clear(); plan->markers.clear();
    if (lf->flag == 'Z' || (lf->flag & 1) == ('Y' & 1)) return 0;   // the baseline re-coder's files
    if (int rc = progressive_plan(&lf->jpeg, lf->jpeg_size, lf->rst_cnt_set, plan)) return rc;   // (what is wrong with the tables comes first, as in recode_progressive)
    if ((int32_t)lf->jpeg_size <= (int32_t)lf->jpeg.garbage.size()) return EX_ASSERTION_FAILURE;
    return 0;
}

// The plan itself, from a parsed JPEG: the decompressor's (above, the JpegFile rebuilt from a .lep header) and the
// compressor's round-trip check (lep_batch.hip: the scans of the frame it has just coded are written again on the GPU and
// compared with the file's own bytes).  Walks the header's DHT / DRI / SOS segments once more, leaving the tables in their
// end-of-file state (what the parser left them in).
int progressive_plan(JpegFile* jfp, size_t jpeg_size, bool rst_cnt_set, ProgPlan* plan) {
    JpegFile& jf = *jfp;
    plan->gpu_ok = false;
    plan->scans.clear(); plan->scan_hdr_end.clear(); plan->markers.clear();
    // eligibility: whole progressive frames of up to three components; everything else keeps the host coder
    // (... and SEQUENTIAL frames coded in several scans: the same walk, every scan a descriptor with from 0 / to 63 that the sequential scan
    // encoders write -- lep_huffprog.h sequential_scan_segment)
    const bool sequential = jf.jpegtype == 1;
    bool ok = (jf.jpegtype == 2 || sequential) && !jf.early_eof && jf.ncomp >= 1 && jf.ncomp <= 3 && jf.mcuh > 0 && jf.mcuv > 0;
    for (int c = 0; c < jf.ncomp; ++c) ok = ok && jf.trunc_bcv[c] >= jf.comp[c].bcv && jf.comp[c].nch > 0 && jf.comp[c].ncv > 0;
    if (!ok) return 0;
    ProgImage& im = plan->image;
    memset(&im, 0, sizeof im);
    im.ncomp = jf.ncomp; im.mcuh = jf.mcuh; im.mcuv = jf.mcuv; im.mcuc = jf.mcuc; im.padbit = jf.padbit;
    for (int c = 0; c < jf.ncomp; ++c) {
        const Component& k = jf.comp[c];
        im.hs[c] = k.hs; im.vs[c] = k.vs; im.bch[c] = k.bch; im.bcv[c] = k.bcv; im.nch[c] = k.nch; im.ncv[c] = k.ncv; im.mbs[c] = k.mbs;
    }
    const uint8_t* h = jf.hdr.data();
    const size_t hdrs = jf.hdr.size();
    size_t hpos = 0;
    int rsti_seen = -1;
    for (;;) {
        uint8_t type = 0;
        while (type != 0xDA) {
            if (hpos >= hdrs) break;
            type = hpos + 1 < hdrs ? h[hpos + 1] : 0;
            const unsigned len = 2 + (((unsigned)(hpos + 2 < hdrs ? h[hpos + 2] : 0)) << 8) + (hpos + 3 < hdrs ? h[hpos + 3] : 0);
            if (type == 0xC4 || type == 0xDA || type == 0xDD)
                if (!parse_segment(&jf, type, len, (unsigned)std::min<size_t>(len, hdrs - hpos), h + hpos, false)) return jf.warn < 0 ? -jf.warn : EX_UNSUPPORTED_JPEG;   // parse_jfif_jpg: errorlevel 2
            hpos += len;
        }
        if (type != 0xDA) break;
        plan->scan_hdr_end.push_back(hpos);
        rsti_seen = jf.rsti;
        ProgScan sc;
        memset(&sc, 0, sizeof sc);
        sc.rsti = jf.rsti;                                     // (a DRI in front of any scan: phone cameras set one per scan -- the reference's androidprogressive.jpg, iphoneprogressive2.jpg)
        sc.cmpc = jf.cs_cmpc; sc.from = jf.cs_from; sc.to = jf.cs_to; sc.sah = jf.cs_sah; sc.sal = jf.cs_sal;
        if (sequential) { sc.from = 0; sc.to = 63; sc.sah = 0; sc.sal = 0; }   // (a sequential scan codes whole blocks whatever its SOS says: encode_block_seq)
        if (sc.cmpc < 1 || sc.cmpc > jf.ncomp || sc.sal < 0 || sc.sal > 13 || sc.from < 0 || sc.to > 63 || sc.from > sc.to) return 0;
        const bool dc = sc.to == 0;
        if (!sequential && sc.from == 0 && sc.to == 63) return 0;             // (a progressive scan of the whole band: nothing writes that)
        if (!sequential && !dc && (sc.cmpc != 1 || sc.from < 1)) return 0;
        if (dc && sc.from != 0) return 0;
        size_t blocks = 0, units;
        for (int i = 0; i < sc.cmpc; ++i) {
            const int c = jf.cs_cmp[i];
            if (c < 0 || c >= jf.ncomp) return 0;
            for (int j = 0; j < i; ++j) if (sequential && sc.cmp[j] == c) return 0;
            sc.cmp[i] = c;
        }
        if (sc.cmpc == 1) { const Component& k = jf.comp[sc.cmp[0]]; blocks = units = (size_t)k.nch * k.ncv; }
        else { units = (size_t)jf.mcuc; for (int i = 0; i < sc.cmpc; ++i) blocks += (size_t)jf.comp[sc.cmp[i]].mbs * jf.mcuc; }
        if (sequential) {
            // code[0..1] = DC tables 0 / 1, code[2..3] = AC tables 0 / 1 as they stand at this scan; scans of several components have the frame's MCUs
            for (int i = 0; i < sc.cmpc; ++i) {
                const Component& k = jf.comp[sc.cmp[i]];
                if (k.dc_tbl < 0 || k.dc_tbl > 1 || k.ac_tbl < 0 || k.ac_tbl > 1 || !jf.htab[0][k.dc_tbl].set || !jf.htab[1][k.ac_tbl].set) return 0;
                if (k.bch != jf.mcuh * k.hs || k.bcv != jf.mcuv * k.vs || k.nch > k.bch || k.ncv > k.bcv) return 0;
                sc.tbl[i] = k.dc_tbl | (k.ac_tbl << 8);
            }
            for (int t = 0; t < 2; ++t)
                for (int i = 0; i < 256; ++i) {
                    sc.code[t][i] = jf.htab[0][t].set ? ((uint32_t)jf.htab[0][t].clen[i] << 16) | jf.htab[0][t].cval[i] : 0u;
                    sc.code[2 + t][i] = jf.htab[1][t].set ? ((uint32_t)jf.htab[1][t].clen[i] << 16) | jf.htab[1][t].cval[i] : 0u;
                }
        } else if (dc) {
            for (int i = 0; i < sc.cmpc; ++i) {
                const int t = jf.comp[sc.cmp[i]].dc_tbl;
                if (t < 0 || t > 1 || (sc.sah == 0 && !jf.htab[0][t].set)) return 0;
                sc.tbl[i] = t;
            }
            for (int t = 0; t < 2; ++t)
                for (int i = 0; i < 256; ++i) sc.code[t][i] = jf.htab[0][t].set ? ((uint32_t)jf.htab[0][t].clen[i] << 16) | jf.htab[0][t].cval[i] : 0u;
        } else {
            const int t = jf.comp[sc.cmp[0]].ac_tbl;
            if (t < 0 || t > 3 || !jf.htab[1][t].set) return 0;
            for (int i = 0; i < 256; ++i) sc.code[0][i] = ((uint32_t)jf.htab[1][t].clen[i] << 16) | jf.htab[1][t].cval[i];
            sc.max_eobrun = jf.htab[1][t].max_eobrun;
            if (sc.max_eobrun < 1) return 0;   // a table without any end-of-band code: the host coder's corner
        }
        const size_t nmark = jf.rsti > 0 ? (units + (size_t)jf.rsti - 1) / (size_t)jf.rsti - 1 : 0;
        const size_t scan_index = plan->scans.size();
        if (rst_cnt_set && nmark > 0 && !(jf.rst_cnt.size() > scan_index && nmark <= jf.rst_cnt[scan_index])) return 0;   // markers withheld: host
        plan->markers.push_back((uint32_t)nmark);
        const size_t geo = blocks * ((dc && !sequential) ? 8 : 432) + nmark * 2 + units / 4 + 64;
        sc.out_cap = (uint32_t)std::min<size_t>(std::min<size_t>(jpeg_size + 16, geo), 0xfffffff0u);
        sc.corr_cap = (!dc && sc.sah != 0) ? (uint32_t)std::min<size_t>(blocks * 2 + 8, 0x7fffffffu) : 0u;
        sc.file_bound = (uint32_t)std::min<size_t>(jpeg_size + 16, 0xfffffff0u);   // what ALL scans of the file come to at most: they are parts of it
        plan->scans.push_back(sc);
        if (plan->scans.size() > 256) return 0;
    }
    if (plan->scans.empty()) return 0;
    im.rsti = jf.rsti;
    plan->gpu_ok = true;
    return 0;
}

int recode_progressive_finish(LepFile* lf, const ProgPlan& plan, const std::vector<std::pair<const uint8_t*, size_t>>& scan_bytes,
                              std::vector<uint8_t>* result) {
    JpegFile& jf = lf->jpeg;
    const size_t max_file_size = lf->jpeg_size;
    if (scan_bytes.size() != plan.scans.size()) return EX_ASSERTION_FAILURE;
    const uint8_t* h = jf.hdr.data();
    const size_t hdrs = jf.hdr.size();
    std::vector<uint8_t> out;
    out.reserve(std::min<size_t>(max_file_size, (size_t)128 << 20) + 16);
    const size_t bound = max_file_size - jf.garbage.size();
    auto put = [&](uint8_t b) { if (out.size() < bound) out.push_back(b); };
    if (lf->has_prefix) for (uint8_t b : lf->prefix_garbage) put(b);
    if (lf->embedded || !lf->has_prefix) { put(0xFF); put(0xD8); }
    size_t hp = 0;
    for (size_t scan = 0; scan < plan.scans.size(); ++scan) {
        for (size_t i = hp; i < plan.scan_hdr_end[scan]; ++i) { if (out.size() >= bound) break; put(i < hdrs ? h[i] : (uint8_t)0); }
        hp = plan.scan_hdr_end[scan];
        const size_t room = out.size() < bound ? bound - out.size() : 0, n = std::min(room, scan_bytes[scan].second);
        out.insert(out.end(), scan_bytes[scan].first, scan_bytes[scan].first + n);
        unsigned cpos = plan.markers[scan];
        if (scan < jf.rst_err.size())
            for (unsigned k = 0; k < jf.rst_err[scan]; ++k) { put(0xFF); put((uint8_t)(0xD0 + (cpos & 7))); ++cpos; }
    }
    for (size_t i = hp; i < hdrs; ++i) put(h[i]);
    for (size_t i = 0; i < jf.garbage.size() && out.size() < max_file_size; ++i) out.push_back(jf.garbage[i]);
    result->swap(out);
    return 0;
}

}  // namespace lep
