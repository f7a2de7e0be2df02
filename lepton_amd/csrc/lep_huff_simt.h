// lep_huff_simt.h -- JPEG Huffman re-encode with one LANE per run of MCUs (round 4).
//
// lep_huff.h writes a thread segment with one wavefront: lane = coefficient, the block's bit string assembled with LDS atomics,
// one block after the other -- 0.2 s for the 7168 segments of a pipeline chunk, in which the 64 lanes of a wavefront spend
// most of their time on each other's atomics.  Writing a scan has no serial dependency besides WHERE a block's bits go: the DC
// predictor of a block is the DC of the block before it, which the frame holds, and a block's bit string depends on nothing
// else.  So here a segment is cut into units of kSimtMcus MCUs, a lane codes one unit with a bit accumulator in its registers,
// and the positions come from prefix sums:
//
//   1  count   lane = unit: the bits its blocks code to (nothing written);
//   2  place   one wavefront per segment: exclusive prefix sum of the units' bits behind the segment's overhang bits;
//   3  code    lane = unit: the same walk, its bits OR-ed into the segment's (zero-filled) bit buffer at the unit's position --
//              MSB first, 32 bits at a time (a unit's first and last dword are shared with its neighbours: those two atomically);
//   4  stuff   one wavefront per segment: the pad bits where the scan ends, then the bit buffer's whole bytes to the output
//              with the 00 behind every FF (16 bytes per lane and step, positions from a prefix sum of the FFs), clipped to
//              the segment's byte bound; the partial byte, its bit count and the last DCs are the segment's end state.
//
// Restart intervals (round 5): a unit never straddles an interval's end (SimtUnitMap); the unit an interval ends with appends the pad
// bits and the two marker bytes to its own bits, a unit an interval starts with begins from zero predictors; pass 2 adds those bits
// to the prefix sum (an interval starts on a byte, so its pad is its own bit count's), and a bit per buffer byte tells pass 4 which
// FFs are markers and take no 00 (recoder.cc:364-400).
// One-component files (never interleaved, whatever their sampling factors) come as frames of nch x ncv MCUs of one block with block rows
// bch apart (recode_prepare, jpeg_recode.cc) and are coded like any other; images whose scan ends inside its last MCU row keep
// lep_huff.h's kernel (HuffSegment.pad bit 0 says which kernel owns a segment).  Same bytes, same end states as that kernel
// (tests/emu, GPU parity tests); recoder.cc:245-412 is what both restate.
#pragma once
#include "lep_huff.h"

namespace lephuff {

constexpr int kSimtMcus = 8;            // MCUs per unit

struct SimtEncSeg {         // per segment handled here
    uint32_t seg;           // index into the caller's segment array
    uint32_t first_unit;    // its first entry in the unit array
    uint32_t nunits;
    uint32_t total_bits;    // pass 2: bits of the segment's stream, overhang bits included
    uint64_t buf_off;       // its bit buffer (bytes, 16-byte aligned) in the scratch arena
    uint32_t buf_bytes;     // multiple of 16
    uint32_t tail;          // pass 3: the stream's last partial byte (its total_bits & 7 bits, top-aligned) -- kept beside the buffer because a
                            // segment that overruns its byte bound is cut off in the buffer and still owes its true end state
    uint32_t cut;           // pass 2: a unit met the cut of a truncated file: the stream ends in front of it
    uint32_t map_bytes;     // restart intervals: bytes of the marker map behind the bit buffer (bit q = byte q of the buffer is a marker's FF); else 0
};
constexpr uint32_t kTailNotOwn = 0x100u;        // pass 3, in SimtEncSeg::tail: the last unit's accumulator does not hold all of the partial byte
constexpr uint32_t kUnitMetCut = 0x80000000u;   // pass 1, in a unit's bit count
constexpr uint32_t kUnitDead = 0xffffffffu;     // pass 2, in place of a unit's position: it lies behind the cut
struct SimtEncWave { uint32_t eseg, first_unit; };   // lane l = unit first_unit + l of SimtEncSeg eseg

// Which MCUs a unit codes.  Without restart intervals: runs of kSimtMcus from the segment's first MCU.  With them a unit ends where its
// interval does: the head (the segment's first MCU up to the first interval end in it) in runs of kSimtMcus, then every interval in
// runs of kSimtMcus -- the last run of each as short as it comes out.
#if LEP_ON_GPU
#define LEPH_BOTH __host__ __device__ __forceinline__    // the launch code sizes its arrays with the same functions
#else
#define LEPH_BOTH inline
#endif
struct SimtUnitMap {
    int m_begin, m_end, rsti, first_end, head_units, per_interval;
    LEPH_BOTH void set(int mb, int me, int r) {
        m_begin = mb; m_end = me; rsti = r > 0 ? r : 0;
        first_end = me;
        if (rsti) {
            const int b = ((mb + rsti - 1) / rsti) * rsti;      // (mb itself when the segment starts where an interval does)
            if (b < me) first_end = b;
        }
        head_units = (first_end - mb + kSimtMcus - 1) / kSimtMcus;
        per_interval = rsti ? (rsti + kSimtMcus - 1) / kSimtMcus : 1;
    }
    LEPH_BOTH uint32_t count() const {
        const int rest = m_end - first_end;
        if (rest <= 0) return (uint32_t)head_units;
        return (uint32_t)head_units + (uint32_t)(rest / rsti) * (uint32_t)per_interval + (uint32_t)((rest % rsti + kSimtMcus - 1) / kSimtMcus);
    }
    // unit u: its MCUs [m0, m1), and the unit its interval (or the head) starts with
    LEPH_BOTH void span(uint32_t u, int* m0, int* m1, uint32_t* interval_first) const {
        int from, to;
        if (u < (uint32_t)head_units) { from = m_begin + (int)u * kSimtMcus; to = first_end; *interval_first = 0; }
        else {
            const uint32_t v = u - (uint32_t)head_units, iv = v / (uint32_t)per_interval, k = v - iv * (uint32_t)per_interval;
            const int base = first_end + (int)iv * rsti;
            from = base + (int)k * kSimtMcus;
            to = base + rsti < m_end ? base + rsti : m_end;
            *interval_first = (uint32_t)head_units + iv * (uint32_t)per_interval;
        }
        *m0 = from;
        *m1 = from + kSimtMcus < to ? from + kSimtMcus : to;
    }
};
// MCU m1 is where a restart interval ends inside the scan (next_mcupos, jpgcoder.cc: the end of the scan comes first)
WDEV bool simt_interval_ends_at(const HuffImage* img, int m1) { return img->rsti > 0 && m1 % img->rsti == 0 && m1 < img->mcuc; }
// ... and whether its marker is written (a truncated file counts the markers it had: rst_limit)
WDEV bool simt_marker_written(const HuffImage* img, int m1) { return (uint32_t)(m1 / img->rsti) - 1u < img->rst_limit; }

struct SimtEncShared {
    uint32_t code[4][256];
};

WDEV void simt_or_word(uint32_t* p, uint32_t v) {
#if LEP_ON_GPU
    if (v) __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    *p |= v;
#endif
}

// bit sink of one lane: count only, or OR into the segment's bit buffer
template <bool WRITE>
struct LaneSink {
    uint64_t acc;           // MSB first
    uint32_t fill;          // bits of acc in use (< 32 between calls)
    uint32_t word;          // WRITE: index of the dword acc's top half goes to;  count: unused
    uint32_t nwords;        // WRITE: dwords of the buffer (bits beyond are dropped: the segment overran its bound anyway)
    uint32_t* buf;
    uint32_t total;
    uint32_t first;         // WRITE: the dword the lane's first bit falls into -- shared with the unit in front, like its last one
    WDEV void start(uint32_t bitpos, uint32_t* b, uint32_t nw) { acc = 0; fill = WRITE ? (bitpos & 31u) : 0u; word = first = bitpos >> 5; buf = b; nwords = nw; total = 0; }
    WDEV void put(uint32_t bits, uint32_t n) {   // n <= 32, bits right-aligned
        if (!WRITE) { total += n; return; }
        if (!n) return;
        acc |= (uint64_t)bits << (64 - fill - n);
        fill += n;
        if (fill >= 32) {
            if (word < nwords) {
                if (word == first) simt_or_word(buf + word, (uint32_t)(acc >> 32));
                else buf[word] = (uint32_t)(acc >> 32);           // a full dword behind the first is this unit's alone
            }
            ++word; acc <<= 32; fill -= 32;
        }
    }
    WDEV void finish() { if (WRITE && fill && word < nwords) simt_or_word(buf + word, (uint32_t)(acc >> 32)); }
    WDEV uint32_t bitpos() const { return word * 32u + fill; }   // WRITE: where the next bit goes
    // the bits behind the last whole byte of everything put so far, top-aligned in a byte (WRITE only; they are this lane's own as long as
    // it put at least seven bits)
    WDEV uint32_t tail_byte() const {
        const uint32_t rem = fill & 7u;
        return rem ? (((uint32_t)(acc >> (64 - fill)) & ((1u << rem) - 1u)) << (8 - rem)) : 0u;
    }
};

// the coefficient at zig-zag position K of a block held as 32 dwords in aligned order
template <int K>
WDEV int coef_at(const uint32_t* w) {
    constexpr int a = kZ2A_const(K);
    return (int16_t)(w[a >> 1] >> (16 * (a & 1)));
}

template <bool WRITE>
struct SimtEncLane {
    const HuffImage* img;
    const SimtEncShared* sh;
    LaneSink<WRITE> sink;
    int lastdc[4];

    WDEV void put_coef(int table, uint32_t runsize_hi, int t) {   // code of (run << 4 | size) + magnitude bits of t
        const int at = (t < 0 ? -t : t) & 0xffff;
        const uint32_t s = (uint32_t)bitlen((uint32_t)at);
        const uint32_t val = (uint32_t)((t > 0) ? t : (t - 1) + (1 << s)) & ((1u << s) - 1u);
        const uint32_t e = sh->code[table][(runsize_hi + s) & 255u];
        sink.put(((e & 0xffffu) << s) | val, (e >> 16) + s);
    }
    template <int K>
    WDEV void ac_step(const uint32_t* w, int act, int& prev) {
        const int t = coef_at<K>(w);
        if (t != 0) {
            const int run = K - prev - 1;
            if (run >= 16) {
                const uint32_t zrl = sh->code[act][0xF0];
                for (int i = run >> 4; i > 0; --i) sink.put(zrl & 0xffffu, zrl >> 16);
            }
            put_coef(act, (uint32_t)(run & 15) << 4, t);
            prev = K;
        }
    }
    template <int K0, int K1>
    WDEV void ac_range(const uint32_t* w, int act, int& prev) {
        if constexpr (K0 < K1) { ac_step<K0>(w, act, prev); ac_range<K0 + 1, K1>(w, act, prev); }
    }
    // encode_block_seq (recoder.cc:245-314)
    WDEV void code_block(int cmp, int dpos) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(img->blocks[cmp] + (int64_t)dpos * 64);
        uint32_t w[32];
        for (int i = 0; i < 32; ++i) w[i] = src[i];
        const int dct = img->dc_tbl[cmp], act = 2 + img->ac_tbl[cmp];
        const int dc = coef_at<0>(w);
        const int cur = cmp == 0 ? lastdc[0] : (cmp == 1 ? lastdc[1] : (cmp == 2 ? lastdc[2] : lastdc[3]));
        const int diff = (int16_t)(dc - cur);
        if (cmp == 0) lastdc[0] = dc; else if (cmp == 1) lastdc[1] = dc; else if (cmp == 2) lastdc[2] = dc; else lastdc[3] = dc;
        put_coef(dct, 0, diff);
        int prev = 0;
        ac_range<1, 64>(w, act, prev);
        if (prev != 63) { const uint32_t e = sh->code[act][0]; sink.put(e & 0xffffu, e >> 16); }
    }
    // the DC of the block in front of MCU `mcu` in scan order, per component (mcu > 0): the predictors a unit starts with
    WDEV void predictors_before(int mcu) {
        const int m = mcu - 1, row = m / img->mcuh, mx = m - row * img->mcuh;
        for (int ci = 0; ci < img->ncomp; ++ci) {
            const int cmp = img->scan_cmp[ci];
            const int hs = img->hs[cmp], vs = img->vs[cmp];
            const int16_t* blk = img->blocks[cmp] + (int64_t)((row * vs + vs - 1) * img->bch[cmp] + mx * hs + hs - 1) * 64;
            const int dc = blk[kZ2A_const(0)];
            if (cmp == 0) lastdc[0] = dc; else if (cmp == 1) lastdc[1] = dc; else if (cmp == 2) lastdc[2] = dc; else lastdc[3] = dc;
        }
    }
    // MCUs [m0, m1) of an interleaved scan; true = stopped in front of the first block behind the cut of a truncated file.
    // Such a block -- at or behind its component's trunc_bc and not the first of its block row (RowCoder::mcu_row, jpeg_recode.cc;
    // decode_row, lepton_codec.cc:7-47) -- is one the reference's decoder never wrote: its re-coder reads what an earlier row left in
    // its two-row ring there.  In a file the reference itself compressed the byte bound cuts the output in front of it; the stream
    // ends here, and the host decides whether that is so (lep_file_recode_finish).
    WDEV bool code_mcus(int m0, int m1) {
        const int mcuh = img->mcuh, ncomp = img->ncomp;
        int row = m0 / mcuh, mx = m0 - row * mcuh;
        for (int m = m0; m < m1; ++m) {
            for (int ci = 0; ci < ncomp; ++ci) {
                const int cmp = img->scan_cmp[ci];
                const int hs = img->hs[cmp], vs = img->vs[cmp], bch = img->bch[cmp], cutat = img->trunc_bc[cmp];
                for (int v = 0; v < vs; ++v)
                    for (int h = 0; h < hs; ++h) {
                        const int dpos = (row * vs + v) * bch + mx * hs + h;
                        if (cutat > 0 && dpos >= cutat && dpos % bch != 0) return true;
                        code_block(cmp, dpos);
                    }
            }
            if (++mx == mcuh) { mx = 0; ++row; }
        }
        return false;
    }
};

WDEV void simt_enc_tables(const HuffImage* img, SimtEncShared* sh) {
    LANES(l) for (int i = l; i < 1024; i += 64) (&sh->code[0][0])[i] = (&img->code[0][0])[i];
    LSYNC();
}

// passes 1 and 3: lanes = units first_unit .. of segment `es`
template <bool WRITE>
WDEV void simt_enc_units(const HuffImage* images, const HuffSegment* segs, SimtEncSeg* esp, SimtEncShared* sh, uint32_t* unit_bits, uint8_t* scratch, uint32_t first_unit) {
    const SimtEncSeg es = *esp;
    const HuffSegment seg = segs[es.seg];
    const HuffImage* img = images + seg.image;
    simt_enc_tables(img, sh);
    SimtUnitMap map;
    map.set(seg.mcu_row0 * img->mcuh, seg.mcu_row1 * img->mcuh, img->rsti);
    LANES(l) {
        const uint32_t u = first_unit + (uint32_t)l;
        if (u < es.nunits) {
            int m0, m1;
            uint32_t interval_first;
            map.span(u, &m0, &m1, &interval_first);
            SimtEncLane<WRITE> d;
            d.img = img; d.sh = sh;
            for (int c = 0; c < 4; ++c) d.lastdc[c] = seg.last_dc[c];
            if (u > 0) {
                // behind a restart marker: zero predictors.  ONE lane-dependent condition on purpose -- as `img->rsti > 0 && m0 % img->rsti
                // == 0` (a wave-uniform test in front of the lane's own, merged into one branch by the optimiser) this was the branch that
                // `-mllvm -structurizecfg-skip-uniform-regions` took for the whole wavefront in the COUNT instantiation of this function:
                // the first unit of every restart interval counted its DC differences against the frame's predictors, the code pass
                // wrote them against zero, and everything behind was displaced (round 5's "miscompile"; isolated in round 6 with
                // scripts/diag_scan_encode_isolate.py, profiles/r6k_*).  The default compiler mode never had the problem; the source no
                // longer offers the pattern.
                const int every = img->rsti > 0 ? img->rsti : 0x7fffffff;
                if (m0 % every == 0) d.lastdc[0] = d.lastdc[1] = d.lastdc[2] = d.lastdc[3] = 0;
                else d.predictors_before(m0);
            }
            uint32_t* buf = reinterpret_cast<uint32_t*>(scratch + es.buf_off);
            const uint32_t at = WRITE ? unit_bits[es.first_unit + u] : 0u;
            if (WRITE && u == 0) {               // the partial byte the segment starts with (ThreadHandoff)
                const uint32_t pend = (seg.overhang >> 8) & 255u;
                if (pend) simt_or_word(buf, (seg.overhang & 255u) << 24);   // (as it is: lep_huff.h starts from the byte unmasked too)
            }
            if (!(WRITE && at == kUnitDead)) {   // (a unit behind the cut of a truncated file writes nothing)
                d.sink.start(at, buf, es.buf_bytes >> 2);
                const bool met_cut = d.code_mcus(m0, m1);
                if (WRITE && simt_interval_ends_at(img, m1)) {   // the interval ends with this unit: abitwriter::pad, then the marker
                    const uint32_t n = (0u - d.sink.bitpos()) & 7u;
                    uint32_t v = 0;
                    for (uint32_t j = 0; j < n; ++j) v = (v << 1) | (uint32_t)((img->padbit >> j) & 1);
                    d.sink.put(v, n);
                    if (simt_marker_written(img, m1)) {
                        const uint32_t q = d.sink.bitpos() >> 3;                     // the buffer byte the marker's FF becomes
                        if (q < es.buf_bytes && es.map_bytes) simt_or_word(reinterpret_cast<uint32_t*>(scratch + es.buf_off + es.buf_bytes) + (q >> 5), 1u << (q & 31u));
                        d.sink.put(0xff00u | 0xd0u | (((uint32_t)(m1 / img->rsti) - 1u) & 7u), 16);
                    }
                }
                d.sink.finish();
                if (!WRITE) unit_bits[es.first_unit + u] = d.sink.total | (met_cut ? kUnitMetCut : 0u);
                // (a unit that meets the cut at its first block and puts no bit leaves a tail made of its own accumulator only -- the bits of
                // the partial byte that belong to the unit in front are not in it.  A kHuffEndCut end state's overhang is therefore NOT
                // defined; nobody compares it: a cut end state is only taken for the last thread with its byte bound reached, lep_huff.h
                // kHuffEndCut.  ADVICE round 5.)
                else if (u + 1 == es.nunits || met_cut) {
                    // ... and of a stream's last partial byte the bits in FRONT of this unit's first are not in its accumulator either:
                    // a last unit that put fewer bits than the byte holds (one block of a one-component scan can code to two) says so,
                    // and the stuffing pass takes the byte from the bit buffer instead
                    const uint32_t put = d.sink.bitpos() - at;
                    esp->tail = d.sink.tail_byte() | (put < (d.sink.bitpos() & 7u) ? kTailNotOwn : 0u);
                }
            }
        }
    }
}

// pass 2 with restart intervals: the units' positions are the prefix sum of their bits (`unit_plain`, kept) plus, behind every interval
// that ends in front of them, its pad bits -- an interval starts on a byte, so they are minus its own bit count modulo eight -- and
// the sixteen of its marker
WDEV void simt_enc_place_intervals(const HuffImage* img, const HuffSegment& seg, SimtEncSeg* es, uint32_t* unit_bits, uint32_t* unit_plain) {
    SimtUnitMap map;
    map.set(seg.mcu_row0 * img->mcuh, seg.mcu_row1 * img->mcuh, img->rsti);
    const uint32_t nunits = es->nunits, fu = es->first_unit;
    uint32_t run = (seg.overhang >> 8) & 255u;
    for (uint32_t base = 0; base < nunits; base += 64) {
        LV(int, nb); LV(int, ex);
        LANES(l) { const uint32_t u = base + (uint32_t)l; L(nb) = u < nunits ? (int)(unit_bits[fu + u] & ~kUnitMetCut) : 0; }
        const int t = lepwave::wave_excl_scan(nb, ex);
        LANES(l) { const uint32_t u = base + (uint32_t)l; if (u < nunits) unit_plain[fu + u] = run + (uint32_t)L(ex); }
        run += (uint32_t)t;
    }
    const uint32_t plain_total = run;
    LSYNC();
    uint32_t extra = 0;
    for (uint32_t base = 0; base < nunits; base += 64) {
        LV(int, xb); LV(int, ex); LV(uint32_t, plain);
        LANES(l) {
            const uint32_t u = base + (uint32_t)l;
            int x = 0;
            uint32_t p = 0;
            if (u < nunits) {
                p = unit_plain[fu + u];
                int m0, m1;
                uint32_t f;
                map.span(u, &m0, &m1, &f);
                if (simt_interval_ends_at(img, m1)) {
                    const uint32_t next = u + 1 < nunits ? unit_plain[fu + u + 1] : plain_total;
                    const uint32_t start = f ? unit_plain[fu + f] : 0u;      // (the segment's first interval holds the overhang bits too)
                    x = (int)((0u - (next - start)) & 7u) + (simt_marker_written(img, m1) ? 16 : 0);
                }
            }
            L(xb) = x; L(plain) = p;
        }
        const int t = lepwave::wave_excl_scan(xb, ex);
        LANES(l) { const uint32_t u = base + (uint32_t)l; if (u < nunits) unit_bits[fu + u] = L(plain) + extra + (uint32_t)L(ex); }
        extra += (uint32_t)t;
    }
    LANES(l) if (l == 0) { es->total_bits = plain_total + extra; es->cut = 0u; }
}

// pass 2: one wavefront per segment (`unit_plain`: a second array of the units' size, used for scans with restart intervals)
WDEV void simt_enc_place(const HuffImage* images, const HuffSegment* segs, SimtEncSeg* es, uint32_t* unit_bits, uint32_t* unit_plain) {
    if (images[segs[es->seg].image].rsti > 0) { simt_enc_place_intervals(images + segs[es->seg].image, segs[es->seg], es, unit_bits, unit_plain); return; }
    const uint32_t pend = (segs[es->seg].overhang >> 8) & 255u;
    uint32_t run = pend;
    bool cut = false;                                      // a unit in front has met the cut of a truncated file: the rest is dead
    for (uint32_t base = 0; base < es->nunits; base += 64) {
        LV(int, nb); LV(int, ex); LV(int, met);
        LANES(l) {
            const uint32_t u = base + (uint32_t)l;
            const uint32_t v = u < es->nunits ? unit_bits[es->first_unit + u] : 0u;
            L(met) = (v & kUnitMetCut) != 0;
            L(nb) = (int)(v & ~kUnitMetCut);
        }
        const uint64_t metmask = lepwave::wave_ballot(met);
        const int first_met = cut ? -1 : (metmask ? (int)__builtin_ctzll(metmask) : 64);   // lanes behind it are dead
        LANES(l) if (l > first_met) L(nb) = 0;
        const int t = lepwave::wave_excl_scan(nb, ex);
        LANES(l) { const uint32_t u = base + (uint32_t)l; if (u < es->nunits) unit_bits[es->first_unit + u] = l > first_met ? kUnitDead : run + (uint32_t)L(ex); }
        run += (uint32_t)t;
        if (first_met < 64) cut = true;
    }
    LANES(l) if (l == 0) { es->total_bits = run; es->cut = cut ? 1u : 0u; }
}

// pass 4: one wavefront per segment
WDEV void simt_enc_stuff(const HuffImage* images, const HuffSegment* segs, const SimtEncSeg& es, uint8_t* scratch, uint8_t* arena, uint32_t* out_len, HuffEnd* ends) {
    const HuffSegment seg = segs[es.seg];
    const HuffImage* img = images + seg.image;
    uint32_t* buf = reinterpret_cast<uint32_t*>(scratch + es.buf_off);
    uint32_t total = es.total_bits;
    const uint32_t room = es.buf_bytes * 8u;
    if (total > room) total = room;                       // the segment overran its bound: what is kept is what the bound keeps
    const bool scan_ends = seg.mcu_row1 * img->mcuh >= img->mcuc && !es.cut;   // (a stream that stops at a cut has no end to pad)
    if (scan_ends && (total & 7u)) {                      // abitwriter::pad: the pad-bit pattern, LSB of the pattern first
        const uint32_t pend = total & 7u, n = 8u - pend;
        uint32_t v = 0;
        for (uint32_t j = 0; j < n; ++j) v = (v << 1) | (uint32_t)((img->padbit >> j) & 1);
        LANES(l) if (l == 0) buf[total >> 5] |= v << (32u - (total & 31u) - n);
        LSYNC();
        total += n;
    }
    const uint32_t nb = total >> 3, cap = seg.out_cap;
    uint8_t* out = arena + seg.out_off;
    const uint32_t* marker_map = es.map_bytes ? reinterpret_cast<const uint32_t*>(scratch + es.buf_off + es.buf_bytes) : nullptr;
    uint32_t written = 0;
    for (uint32_t base = 0; base < nb; base += 1024) {
        LV(int, nff); LV(int, before);
        LV(uint32_t, w0); LV(uint32_t, w1); LV(uint32_t, w2); LV(uint32_t, w3); LV(uint32_t, mk);
        LANES(l) {
            const uint32_t i = base + 16u * (uint32_t)l;
            uint32_t a = 0, b = 0, c = 0, d = 0, markers = 0;
            int n = 0;
            if (i < nb) {
                const uint32_t* p = buf + (i >> 2);
                a = p[0]; b = p[1]; c = p[2]; d = p[3];
                if (marker_map) markers = (marker_map[i >> 5] >> (i & 16u)) & 0xffffu;   // bit k: byte i + k is a restart marker's FF
                const uint32_t have = nb - i < 16u ? nb - i : 16u;
                for (uint32_t k = 0; k < have; ++k) {
                    const uint32_t word = k < 4 ? a : (k < 8 ? b : (k < 12 ? c : d));
                    n += (((word >> (24 - 8 * (k & 3))) & 255u) == 0xffu) & (~markers >> k & 1u);
                }
            }
            L(w0) = a; L(w1) = b; L(w2) = c; L(w3) = d; L(nff) = n; L(mk) = markers;
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
                    if (byte == 0xffu && !(L(mk) >> k & 1u)) { if (pos < cap) out[pos] = 0; ++pos; }
                }
            }
        }
        written += (nb - base < 1024u ? nb - base : 1024u) + (uint32_t)ffs;
    }
    const uint32_t rem = scan_ends ? 0u : (es.total_bits & 7u);     // (of the whole stream, whatever the buffer kept of it)
    LANES(l) if (l == 0) {
        out_len[es.seg] = written < cap ? written : cap;
        if (ends) {
            HuffEnd e;
            e.attempted = written;
            e.num_overhang_bits = (uint8_t)rem;
            e.overhang_byte = (uint8_t)(rem ? es.tail : 0u);
            bool tail_lost = false;
            if (rem && (es.tail & kTailNotOwn)) {        // the partial byte as the buffer holds it -- if the buffer kept the stream's end
                if (es.total_bits <= room) e.overhang_byte = (uint8_t)((buf[es.total_bits >> 5] >> (24u - (es.total_bits & 24u))) & (0xff00u >> rem) & 0xffu);
                else tail_lost = true;
            }
            // (lep_huff.h's writer zeroes all four predictors at every restart marker position, the unused fourth too)
            const int m_begin = seg.mcu_row0 * img->mcuh, m_end = seg.mcu_row1 * img->mcuh;
            const int last_end = img->rsti > 0 ? ((m_end < img->mcuc ? m_end : img->mcuc - 1) / img->rsti) * img->rsti : 0;   // the last interval end in (m_begin, m_end]
            const bool reset_inside = img->rsti > 0 && last_end > m_begin;
            for (int c = 0; c < 4; ++c) e.last_dc[c] = reset_inside ? (int16_t)0 : seg.last_dc[c];
            const int m = m_end - 1;
            if (m >= m_begin && !(reset_inside && last_end == m_end)) {
                const int row = m / img->mcuh, mx = m - row * img->mcuh;
                for (int ci = 0; ci < img->ncomp; ++ci) {
                    const int cmp = img->scan_cmp[ci];
                    const int hs = img->hs[cmp], vs = img->vs[cmp];
                    e.last_dc[cmp & 3] = img->blocks[cmp][(int64_t)((row * vs + vs - 1) * img->bch[cmp] + mx * hs + hs - 1) * 64 + kZ2A_const(0)];
                }
            }
            e.pad = es.cut ? kHuffEndCut : (tail_lost ? kHuffEndRefused : (uint16_t)0);
            ends[es.seg] = e;
        }
    }
}

// which segments this form takes
inline bool simt_enc_takes(const HuffImage& img, const HuffSegment& seg) {
    return (img.rsti == 0 || (img.rsti > 0 && !(img.trunc_bc[0] | img.trunc_bc[1] | img.trunc_bc[2] | img.trunc_bc[3]))) && img.interleaved == 1 && img.mcuc == img.mcuh * img.mcuv && seg.mcu_row0 >= 0 && seg.mcu_row1 > seg.mcu_row0 && seg.mcu_row1 <= img.mcuv &&
           ((seg.overhang >> 8) & 255u) < 8u;
}

}  // namespace lephuff
