// lep_enc5.h -- the SPLIT-PHASE encoder ("v5"): the arithmetic coder of one thread segment cut into passes that each have
// their own unit of parallelism, instead of one wavefront walking a segment's blocks with one useful lane in its serial part.
//
// Why it can be cut (encode direction only): every context of the block syntax (encoder.cc:194-402, model.hh:463-1139) is a
// function of the quantised coefficients, which the encoder already holds -- nothing upstream of the bool coder depends on it.
// And a Branch (branch.hh:82-100) adapts only on the bins coded with it, so the model is not ONE serial state but ~1,500
// independent CHAINS per segment (all Branches that share a context family, a position and a "non-zeros left" class never
// touch the others').  The only truly serial thing is the bool writer's range / low recurrence (boolwriter.hh:48-118), and
// that needs nothing but a list of (probability, bit) pairs.
//
//   count   one wavefront per segment, lane = block (64 consecutive blocks of a row = a TILE): how many entries every chain
//           of the segment will get (a function of the block's own coefficients) -> exact arena layout, nothing overflows
//   emit    same walk with the contexts (aavrg / Lakhani priors, IDCT + DC prediction, neighbour summaries): every coded
//           coefficient becomes 4-byte ENTRIES (<= 4 bins each) appended, in stream order, to its chain's private stream
//   fold    lane = chain: walks its stream with its Branches in LDS, replaces every entry by the probabilities its bins
//           are coded with (in place) -- the model never touches HBM (the 2 MB threshold table excepted)
//   gather  the emit walk again: reads the probabilities back, writes the segment's bins in stream order
//   write   lane = segment: the serial bool writer over the bin list, 64 segments per wavefront in true SIMT
//
// Results are bit-identical to lep_enc3.h / the oracle (tests/test_core_emulation.py steps these kernels on the CPU).
// Reference citations are those of lep_core.h; the syntax walk below follows SegmentCoder<false>::code_block line by line.
#pragma once
#include "lep_v3.h"
#ifdef LEP5_DEBUG
#include <cstdio>
#endif

namespace lep5 {
using namespace lep3;

// ---- chains and their streams ------------------------------------------------------------------------------------------
// rows: 0..48 interior positions (zig-zag order), 49..55 horizontal edge (x = 1..7), 56..62 vertical edge (y = 1..7),
// 63 the threshold bins of all edge positions (their Branches are not keyed by the position: residual_threshold_counts_
// [colour][prior class][min(len - thr, 7)][node], model.hh:1072-1099).  A stream = (colour index, row, class k): k =
// kNzBin[non-zeros left] for the interior, edge non-zeros left (1..7) for an edge, min(len - thr, 7) for the threshold row.
constexpr int kRows = 64, kClasses = 10;
constexpr int kPRows = 63 + 14;                         // payload rows of a tile: the 63 coefficient rows + a threshold entry per edge position
constexpr int kStreams = 2 * kRows * kClasses;          // 1280 4-byte-unit streams per segment
WDEV int stream_id(int ci, int row, int k) { return (ci * kRows + row) * kClasses + k; }
// slice of a coefficient chain: exponent Branches [12 bsr][11] then residual Branches [10]
constexpr int kCoefSlice = 12 * 11 + 10 + 1;   // (+ 1: the Branch the bins a unit does not have are pointed at, see fold_coef_unit)
constexpr int kSignSlice = 48, kNzSlice = 6 * 32, kEdgeNzSlice = 8 * 12, kDcSlice = 17 * 11 + 10 + 1;
constexpr uint32_t kThreshWords = 2u * 256 * 8 * 128;   // per segment, HBM: [ci][prior class][lt][128]

// A coefficient's chain-local bin sequence: its exponent bins (nexp = min(len + 1, 11)), then its residual bins that use the
// chain's own Branches (nres: all len - 1 of them in the interior, the ones below the noise threshold on an edge).  It is cut
// into UNITS of four bins; an entry is one unit.  (The sign bin and an edge's threshold bins sit between the two in the
// stream but belong to other chains; `gather` puts them back in order.)
//   bits 0..9 residual bits (low nres bits of |v|), 10..13 nres, 14..17 len, 18..22 bsr (b17 for a DC entry), 23..26 class k
//   (a for a DC entry), 27..29 unit
WDEV uint32_t coef_entry(uint32_t vlow, int nres, int len, int bsr, int k, int u) {
    return vlow | ((uint32_t)nres << 10) | ((uint32_t)len << 14) | ((uint32_t)bsr << 18) | ((uint32_t)k << 23) | ((uint32_t)u << 27);
}
WDEV int coef_units(int len, int nres) { return ((len < 11 ? len + 1 : 11) + nres + 3) >> 2; }
// threshold entry: bits 0..9 the threshold bits (|v| >> thr, n of them), 10..13 n, 14..21 prior class, 23..26 lt, 27..29 unit
WDEV uint32_t thresh_entry(uint32_t tbits, int n, int pcls, int lt, int u) {
    return tbits | ((uint32_t)n << 10) | ((uint32_t)pcls << 14) | ((uint32_t)lt << 23) | ((uint32_t)u << 27);
}
// sign entry (one byte): bits 0..5 Branch slot inside the colour's [4][12] table, bit 6 the bit, bit 7 valid
// sparse records, one per block ordinal of the segment (chains whose key needs the neighbours).  emit writes them in block
// order; `bucket` sorts them into one stream per chain (the fold lanes had filtered the block-ordered records by key before:
// 64 chains x every block's key and record -- two thirds of the fold stage's memory requests) and leaves the places for gather:
//   KEY (4 bytes): ci | 7x7 context bin << 1 | eob_x << 5 | eob_y << 8 | a << 11 -- the record's four chains
//   NZ  (8 bytes): word 0 = number of non-zeros                           -> six probabilities in bytes 0..5
//   EN  (2 x 4 bytes, horizontal then vertical): nzq | ne << 3            -> three probabilities
//   DC  (6 x 4 bytes): word 0 = unit 0 of an entry like a coefficient's with k = a, bsr = b17 -> the units' probabilities
constexpr int kKeyRec = 4, kNzRec = 8, kEnRec = 8, kDcRec = 24;
// the sparse chains: 7x7 counts (ci, context bin), horizontal / vertical edge counts (ci, eob_x / eob_y), DC (a)
// The array of places: rows of a tile, lane by lane -- 63 coefficient rows and the DC's entry (its sign in bit 31).  A word per
// CODED coefficient in the block's own order would be 2.3 x smaller, and was measured: every lane then stores and loads at an
// address of its own (64 sectors per instruction instead of 4) -- emit 129 -> 171 ms, gather 165 -> 242 ms (MI355X, 1024 x 4K).
constexpr int kAtRows = 64;
constexpr int kClsNz = 0, kClsEh = 20, kClsEv = 36, kClsDc = 52, kNumCls = 64;
WDEV void key_classes(uint32_t key, int c[4]) {
    const int ci = (int)(key & 1u);
    c[0] = kClsNz + ci * 10 + (int)((key >> 1) & 15u); c[1] = kClsEh + ci * 8 + (int)((key >> 5) & 7u);
    c[2] = kClsEv + ci * 8 + (int)((key >> 8) & 7u); c[3] = kClsDc + (int)((key >> 11) & 15u);
}

// gather and write run in PARTS (tile ranges of every segment): the writer is 128 wavefronts that take the same time whatever the
// batch, so part k is written while part k + 1 is gathered.  emit leaves what both need to take a segment up at a part's first
// tile: the walk's counters there.
constexpr int kMaxParts = 8;
struct Ckpt5 { uint32_t tile_no, ord0, sign_pos[2], nbins, tcur[16], pad[3]; };   // 96 bytes
WDEV uint32_t part_first_tile(uint32_t ntiles, int part, int nparts) { return (uint32_t)(((uint64_t)ntiles * (uint32_t)part) / (uint32_t)nparts); }

struct SegPlan5 {            // one per segment, device memory; written by plan5 from the counts
    uint64_t arena_off;      // byte offset of the segment's entry arena (16-byte aligned)
    uint64_t bins_off;       // offset of the segment's bin list, in bins (uint16)
    uint32_t sign_base[2], sign_cnt[2];       // byte streams, relative to arena_off
    uint32_t key_base, nz_base, en_base, dc_base;   // byte offsets of the sparse regions (block order, written by emit)
    uint32_t cls_base;       // bucket's table: [kNumCls] first record of the chain's stream, [kNumCls] records
    uint32_t place_base;     // [4][nblocks]: the block's place in its 7x7 count / horizontal / vertical / DC stream
    uint32_t nzs_base, ens_base[2], dcs_base;       // the chains' streams (every chain padded to four records)
    uint32_t ckpt_base;      // [kMaxParts] Ckpt5: where gather / write may take the segment up again (written by emit)
    uint32_t wstate_base;    // the bool writer's state between its parts
    uint32_t at_base, ntiles;                       // the places emit gave out, for gather: [tile][64 rows][64 lanes] dwords
    uint32_t nblocks;        // block ordinals (coded blocks of the segment)
    uint32_t bins_cap;       // room in the bin list
    uint32_t nbins;          // bins written by gather (without the start marker and the stop bins)
    uint32_t arena_bytes;
    int32_t status;          // exit code of the emit walk (COEFFICIENT_OUT_OF_RANGE, ...): the later passes skip the segment
    uint32_t base[kStreams + 1];   // unit streams: index of the first unit (4-byte units, relative to arena_off; a multiple of 4: every
                                   // stream starts on 16 bytes and is padded to 16, so that the fold lanes move whole dwordx4); [kStreams] = end
    uint32_t cnt[kStreams];        // units in the stream
};

// ---- Branch state in LDS: the packed word of lep_core.h (false | true << 8 | probability << 16) ----------------------------
WDEV uint32_t mul24_5(uint32_t a, uint32_t b) {   // low 32 bits of the 24 x 24-bit product
#if LEP_ON_GPU
    return __umul24(a, b);
#else
    return (uint32_t)(((uint64_t)(a & 0xffffff) * (b & 0xffffff)) & 0xffffffffu);
#endif
}
// Branch::record_obs_and_update (branch.hh:82-100) on the packed word.  The probability comes from lep3::prob_of (on the GPU a
// float reciprocal with an exact integer fix-up, no table: a table in LDS would put a second memory round trip behind every bin)
WDEV uint32_t bupd5(uint32_t w, uint32_t obs, const uint32_t* inv24) {
    (void)inv24;
    uint32_t f = (w & 255) + (obs ^ 1), t = ((w >> 8) & 255) + obs;
    if ((f | t) > 255) {   // the incremented count was 255
        const uint32_t f0 = w & 255, t0 = (w >> 8) & 255;
        if ((obs ? f0 : t0) == 1) return (w & 0xffff) | ((obs ? 0u : 255u) << 16);
        f = obs ? (1 + f0) >> 1 : 129u;
        t = obs ? 129u : (1 + t0) >> 1;
    }
    return f | (t << 8) | (prob_of(f, t) << 16);
}
WDEV uint32_t inv24_5(uint32_t d) {   // lep_dec4.h inv24_of: exact for every reachable count pair
    if (d < 2) return 0;
    return (0x1000000u + d - 1) / d - ((d == 337 || d == 469) ? 1u : 0u);
}

// ---- fold: lane = chain --------------------------------------------------------------------------------------------------
// A Branch in LDS is 16 bits: false count | true count << 8.  Its probability is a function of the counts -- (f << 8) / (f + t),
// branch.hh:108-125 -- with ONE exception: a Branch that saturates on "true" (counts (1, 255) met by another true) codes with
// probability 0 from then on, where the same pair reached by counting gives 1 (branch.hh:86-91).  That state is stored as
// t = 0.  (Saturating on "false" leaves (255, 1) with probability 255, which is what the formula gives.)  Half the LDS of the
// packed word of lep_core.h: twice the resident fold wavefronts.
// Both are straight-line code: a lane's bins take every path there is, and 64 lanes take them all at once -- with branches the
// compiler's exec-mask bookkeeping was three quarters of the fold kernels' instructions.
WDEV uint32_t prob16(uint32_t w) {
    const uint32_t f = w & 255u, t = w >> 8;
#if LEP_ON_GPU
    // floor(256 f / (f + t)) from the hardware reciprocal: the product is off by less than 2^-14 (1 ulp of rcp, half of the
    // multiply, quotient < 256) and a quotient that is not whole is at least 1 / 510 away from the next whole number, so adding
    // 2^-10 before truncating lands on the right side in every case (lep_gpu_selftest checks all 255 x 255 pairs on the device)
    const uint32_t p = (uint32_t)__builtin_fmaf((float)(f << 8), __builtin_amdgcn_rcpf((float)(f + t)), 0x1p-10f);
#else
    const uint32_t p = (f << 8) / (f + t);
#endif
    return t ? p : 0u;
}
WDEV uint32_t upd16(uint32_t w, uint32_t obs) {
    const uint32_t sh = obs << 3, t = w >> 8;
    const uint32_t hot = (w >> sh) & 255u, cold = (w >> (8u - sh)) & 255u;   // the count that grows, the other one
    const uint32_t half = (1u + cold) >> 1;
    const uint32_t renorm = (129u << sh) | (half << (8u - sh));            // the grown count was 255: both halve
    const uint32_t stuck = obs ? 1u : w;                                     // ... and the other one 1: saturated (true: the t = 0 state)
    uint32_t r = w + (1u << sh);
    r = hot == 255u ? (cold == 1u ? stuck : renorm) : r;
    return t == 0u ? (obs ? w : (2u | (255u << 8))) : r;                     // saturated true: counts (1, 255); a false makes them (2, 255)
}
constexpr uint32_t kBranchInit16 = 1u | (1u << 8);
// Every lane owns `SLICE` Branches in LDS, laid out [branch][lane] (lanes next to each other: no bank conflicts whatever the
// lanes index).  code(): one bin -- returns the probability it is coded with, adapts the Branch.
struct FoldShared {
    uint16_t slice[kDcSlice * 64];   // the largest slice (197 Branches per lane)
};
struct FoldLane {
    uint16_t* s;   // &slice[lane]
    WDEV uint32_t code(int branch, uint32_t bit) {
        const uint32_t w = s[branch * 64];
        s[branch * 64] = (uint16_t)upd16(w, bit);
        return prob16(w);
    }
};
// where lane l keeps its Branches inside a row of 64: lanes l and l + 32 share a dword (the LDS serves 32 lanes a cycle: two
// lanes of one half-wave in one bank, at different rows, would be a conflict)
WDEV int fold_col(int l) { return ((l & 31) << 1) | (l >> 5); }
WDEV void fold_init(FoldShared* sh, int words_per_lane) {
    LANES(l) {
        for (int i = 0; i < words_per_lane; ++i) sh->slice[i * 64 + fold_col(l)] = (uint16_t)kBranchInit16;
    }
    LSYNC();
}

// the bins of unit u of a coefficient entry through the lane's slice; exponent Branch = bsr * 11 + i, residual = rbase + b.
// Straight-line: a bin the unit does not have goes through the spare Branch at rbase + 10, whose state nobody reads (its byte
// of the result is not one gather looks at).
WDEV uint32_t fold_coef_unit(FoldLane& fl, uint32_t e, int rbase) {
    const int nres = (int)(e >> 10) & 15, len = (int)(e >> 14) & 15, bsr = (int)(e >> 18) & 31, u = (int)(e >> 27) & 7;
    const int nexp = len < 11 ? len + 1 : 11, m = nexp + nres;
    // the (up to) four bins of a unit use four different Branches: all are read before any is adapted and written back, so
    // the lane waits for LDS once per unit, not once per bin
    int br[4]; uint32_t bit[4], w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = 4 * u + q;
        const int b = imax(nres - 1 - (j - nexp), 0);
        br[q] = j < nexp ? bsr * 11 + j : rbase + b;
        bit[q] = j < nexp ? (uint32_t)(len != j) : (e >> b) & 1u;
        br[q] = j >= m ? rbase + 10 : br[q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = fl.s[br[q] * 64];
    uint32_t probs = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { fl.s[br[q] * 64] = (uint16_t)upd16(w[q], bit[q]); probs |= prob16(w[q]) << (8 * q); }
    return probs;
}

// coefficient chains: lane = (segment seg0 + lane, stream sid): the same chain of 64 consecutive segments in one wavefront
// (chains of one kind are about equally long: the lanes finish together)
WDEV void fold_coef_wave(const SegPlan5* plans, uint8_t* arena, int seg0, int nseg, int sid, FoldShared* sh) {
    fold_init(sh, kCoefSlice);
    LANES(l) {
        const int seg = seg0 + l;
        if (seg < nseg) {
            const SegPlan5& P = plans[seg];
            uint32_t* units = reinterpret_cast<uint32_t*>(arena + P.arena_off);
            FoldLane fl{sh->slice + fold_col(l)};
            const uint32_t n = P.status ? 0u : P.cnt[sid];
            uint32_t* p = units + P.base[sid];   // 16-byte aligned, padded to whole groups of four
            U4 nxt = n ? ld4(p) : U4{0, 0, 0, 0};
            for (uint32_t i = 0; i < n; i += 4) {   // four entries per dwordx4, the next group requested before this one is folded
                U4 g = nxt;
                if (i + 4 < n) nxt = ld4(p + i + 4);
                g.x = fold_coef_unit(fl, g.x, 132);   // (behind the stream's last unit: zeros, written by emit -- harmless bins)
                g.y = fold_coef_unit(fl, g.y, 132);
                g.z = fold_coef_unit(fl, g.z, 132);
                g.w = fold_coef_unit(fl, g.w, 132);
                st4(p + i, g);
            }
        }
    }
}

// threshold chains (edge residual bits at or above the noise threshold, encoder.cc:132-152): Branches in HBM,
// thresh[ci][prior class][lt][node], node = 1, then min(2 * node + bit, 127)
WDEV void fold_thresh_wave(const SegPlan5* plans, uint8_t* arena, uint32_t* thresh_models, int seg0, int nseg, int sid, int ci, FoldShared* sh) {
    LANES(l) {
        const int seg = seg0 + l;
        if (seg < nseg) {
            const SegPlan5& P = plans[seg];
            uint32_t* units = reinterpret_cast<uint32_t*>(arena + P.arena_off);
            uint32_t* model = thresh_models + (size_t)seg * kThreshWords;
            for (uint32_t i = P.base[sid]; i < (P.status ? 0u : P.base[sid] + P.cnt[sid]); ++i) {
                const uint32_t e = units[i];
                const int n = (int)(e >> 10) & 15, pcls = (int)(e >> 14) & 255, lt = (int)(e >> 23) & 15, u = (int)(e >> 27) & 7;
                uint32_t* T = model + (((uint32_t)ci * 256 + pcls) * 8 + lt) * 128;

                int node = 1;
                uint32_t probs = 0;
                for (int t = 0; t < n && t < 4 * u + 4; ++t) {
                    const uint32_t bit = (e >> (n - 1 - t)) & 1u;
                    if (t >= 4 * u) {
                        const uint32_t w = T[node];
                        T[node] = bupd5(w, bit, nullptr);
                        probs |= (w >> 16) << (8 * (t - 4 * u));
                    }
                    node = imin((node << 1) | (int)bit, 127);
                }
                units[i] = probs;
            }
        }
    }
}

// sign chains: one byte per entry; lane = segment, the colour's 48 sign Branches in the lane's slice.
// A chain step is LDS read -> probability / update -> LDS write, and half of a segment's signs go through ONE Branch (the interior's),
// so the round trip to LDS stood on the chain 350,000 times per segment.  The Branch of the NEXT entry is read while this entry is
// computed, and if it is the Branch just updated the new state is forwarded instead of what the early read saw (the LDS executes a
// wavefront's accesses in order: an early read can only be stale with respect to the one write behind it).  Straight-line per entry:
// an entry that is no sign (the padding of a stream) goes through the spare Branch 63 and is written back unchanged.
WDEV void fold_sign_wave(const SegPlan5* plans, uint8_t* arena, int seg0, int nseg, int ci, FoldShared* sh) {
    fold_init(sh, kSignSlice > 64 ? kSignSlice : 64);
    LANES(l) {
        const int seg = seg0 + l;
        if (seg < nseg) {
            const SegPlan5& P = plans[seg];
            uint8_t* s = arena + P.arena_off + P.sign_base[ci];
            uint16_t* br = sh->slice + fold_col(l);
            const uint32_t n = P.status ? 0u : P.sign_cnt[ci];
            uint32_t* p = reinterpret_cast<uint32_t*>(s);   // the stream starts on 16 bytes and is padded to 16
            U4 nxt = n ? ld4(p) : U4{0, 0, 0, 0};
            auto slot_of = [](uint32_t e) { return (e & 0x80u) ? (int)(e & 63u) : 63; };
            int cur_b = n ? slot_of(nxt.x & 255u) : 63;
            uint32_t cur_w = br[cur_b * 64];
            for (uint32_t i = 0; i < n; i += 16) {
                U4 g = nxt;
                if (i + 16 < n) nxt = ld4(p + (i >> 2) + 4);
                uint32_t w[4] = {g.x, g.y, g.z, g.w};
                const uint32_t follow = i + 16 < n ? (nxt.x & 255u) : 0u;   // the entry behind this group's last one
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const uint32_t e = (w[q >> 2] >> (8 * (q & 3))) & 255u;
                    const uint32_t en = q < 15 ? (w[(q + 1) >> 2] >> (8 * ((q + 1) & 3))) & 255u : follow;
                    const int nb = slot_of(en);
                    const uint32_t early = br[nb * 64];                     // requested before this entry's update is written
                    const uint32_t isgn = (i + (uint32_t)q < n) ? (e >> 7) & 1u : 0u;   // (the group's tail behind the stream's end is not the chain's)
                    const uint32_t pr = prob16(cur_w);
                    const uint32_t upd = isgn ? upd16(cur_w, (e >> 6) & 1u) : cur_w;
                    br[cur_b * 64] = (uint16_t)upd;
                    const uint32_t outb = isgn ? pr : e;
                    w[q >> 2] = (w[q >> 2] & ~(255u << (8 * (q & 3)))) | (outb << (8 * (q & 3)));
                    cur_w = nb == cur_b ? upd : early;
                    cur_b = nb;
                }
                st4(p + (i >> 2), U4{w[0], w[1], w[2], w[3]});
            }
        }
    }
}

// sparse chains: lane = segment; every chain has a stream of its own (bucket_wave), padded to four records
// number of non-zeros of the 7x7 interior: six bins MSB first through T[level][prefix] (encoder.cc:200-213)
WDEV void fold_nz_wave(const SegPlan5* plans, uint8_t* arena, int seg0, int nseg, int ci, int ctxbin, FoldShared* sh) {
    fold_init(sh, kNzSlice);
    LANES(l) {
        const int seg = seg0 + l;
        if (seg < nseg) {
            const SegPlan5& P = plans[seg];
            const uint32_t* ctab = reinterpret_cast<const uint32_t*>(arena + P.arena_off + P.cls_base);
            const int c = kClsNz + ci * 10 + ctxbin;
            const uint32_t n = P.status ? 0u : ctab[kNumCls + c];
            uint32_t* rec = reinterpret_cast<uint32_t*>(arena + P.arena_off + P.nzs_base) + 2 * (size_t)(n ? ctab[c] : 0u);
            FoldLane fl{sh->slice + fold_col(l)};
            U4 nxt = n ? ld4(rec) : U4{0, 0, 0, 0};
            for (uint32_t i = 0; i < n; i += 2) {   // two records per dwordx4 (behind the last one: zeros)
                U4 g = nxt;
                if (i + 2 < n) nxt = ld4(rec + 2 * (i + 2));
                uint32_t nzv[2] = {g.x, g.z}, lo[2], hi[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int nz = (int)nzv[q] & 63;
                    lo[q] = 0; hi[q] = 0;
                    int so_far = 0;
#pragma unroll
                    for (int b = 5; b >= 0; --b) {
                        const uint32_t bit = (uint32_t)(nz >> b) & 1u;
                        const uint32_t p = fl.code(b * 32 + so_far, bit);
                        const int t = 5 - b;
                        if (t < 4) lo[q] |= p << (8 * t); else hi[q] |= p << (8 * (t - 4));
                        so_far = (so_far << 1) | (int)bit;
                    }
                }
                st4(rec + 2 * i, U4{lo[0], hi[0], lo[1], hi[1]});
            }
        }
    }
}
// edge non-zero counts: three bins MSB first through T[nzq][level][prefix] of (ci, horizontal / vertical, eob)
WDEV void fold_edgenz_wave(const SegPlan5* plans, uint8_t* arena, int seg0, int nseg, int ci, int vertical, int eob, FoldShared* sh) {
    fold_init(sh, kEdgeNzSlice);
    LANES(l) {
        const int seg = seg0 + l;
        if (seg < nseg) {
            const SegPlan5& P = plans[seg];
            const uint32_t* ctab = reinterpret_cast<const uint32_t*>(arena + P.arena_off + P.cls_base);
            const int c = (vertical ? kClsEv : kClsEh) + ci * 8 + eob;
            const uint32_t n = P.status ? 0u : ctab[kNumCls + c];
            uint32_t* rec = reinterpret_cast<uint32_t*>(arena + P.arena_off + P.ens_base[vertical]) + (size_t)(n ? ctab[c] : 0u);
            FoldLane fl{sh->slice + fold_col(l)};
            U4 nxt = n ? ld4(rec) : U4{0, 0, 0, 0};
            for (uint32_t i = 0; i < n; i += 4) {
                U4 g = nxt;
                if (i + 4 < n) nxt = ld4(rec + i + 4);
                uint32_t w[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nzq = (int)w[q] & 7, ne = (int)(w[q] >> 3) & 7;
                    uint32_t probs = 0;
                    int so_far = 0;
#pragma unroll
                    for (int b = 2; b >= 0; --b) {
                        const uint32_t bit = (uint32_t)(ne >> b) & 1u;
                        probs |= fl.code(nzq * 12 + b * 4 + so_far, bit) << (8 * (2 - b));
                        so_far = (so_far << 1) | (int)bit;
                    }
                    w[q] = probs;
                }
                st4(rec + i, U4{w[0], w[1], w[2], w[3]});
            }
        }
    }
}
// DC chains, key a = min(bit length of the uncertainty, 11): exponent Branches [17 b][11], residual Branches [10]
WDEV void fold_dc_wave(const SegPlan5* plans, uint8_t* arena, int seg0, int nseg, int a, FoldShared* sh) {
    fold_init(sh, kDcSlice);
    LANES(l) {
        const int seg = seg0 + l;
        if (seg < nseg) {
            const SegPlan5& P = plans[seg];
            const uint32_t* ctab = reinterpret_cast<const uint32_t*>(arena + P.arena_off + P.cls_base);
            const uint32_t n = P.status ? 0u : ctab[kNumCls + kClsDc + a];
            uint32_t* rec = reinterpret_cast<uint32_t*>(arena + P.arena_off + P.dcs_base) + 6 * (size_t)(n ? ctab[kClsDc + a] : 0u);
            FoldLane fl{sh->slice + fold_col(l)};
            uint32_t nxt = n ? rec[0] : 0u;
            for (uint32_t i = 0; i < n; ++i) {
                const uint32_t e0 = nxt;
                if (i + 1 < n) nxt = rec[6 * (i + 1)];
                const int nu = coef_units((int)(e0 >> 14) & 15, (int)(e0 >> 10) & 15);
                for (int u = 0; u < nu; ++u) rec[6 * i + u] = fold_coef_unit(fl, e0 | ((uint32_t)u << 27), 17 * 11);
            }
        }
    }
}

// ---- write: lane = segment bool writer ------------------------------------------------------------------------------------
// The bool writer (boolwriter.hh:48-118, boolwriter.cc:17-35) as every lane runs it on its own segment.  Two things differ from
// the serial form, neither in the bytes: (1) byte output is DEFERRED -- low is 64 bits wide and the whole bytes above the 24 + 7
// bits the recurrence works on are taken off every four bins (at most 7 bits per bin: 4 bytes), so the per-bin path is the
// recurrence alone and the output path is straight-line code in wave-uniform control flow; (2) bytes go to memory four at a time
// from a staging register, and a carry is the top bit of the number added to it -- only one that leaves the staged bytes (all of
// them 0xFF) ripples back through memory like the serial writer's.  Checked against lepdev::BoolCoder<false> on random and
// adversarial bin sequences (tests/test_core_emulation.py).
struct BoolEnc5 {
    uint64_t low;
    uint32_t q24;       // (range - 1) << 24: the form the recurrence is shortest in (see bin())
    int count;
    uint64_t stage;     // bytes of the code value that are not in memory yet, newest in the low byte
    int nst;            // how many (0..3 between flushes)
    uint8_t* out;
    uint32_t pos, cap;  // bytes handed to memory (written when below cap)
    uint32_t floor_;    // the first byte that is this writer's own (a chunk of a stream, see the chunked writer below): a carry that
    uint32_t carry_out; // would ripple below it is counted here instead and added when the chunks are stitched
    WDEV void init(uint8_t* o, uint32_t c) { out = o; cap = c; pos = 0; low = 0; q24 = 254u << 24; count = -24; stage = 0; nst = 0; floor_ = 0; carry_out = 0; }
    // boolwriter.hh:48-118 without the byte output.  split = 1 + (((range - 1) * prob) >> 8); with the state kept as
    // (range - 1) << 24 that product's high word IS split - 1, and the normalised next range is n << clz(n): the dependent chain of
    // a bin is mul_hi -> add / sub -> select -> clz -> shift-add, five instructions (every lane waits out this chain 2.4 million
    // times per segment: it, not the instruction count, is what the write pass takes)
    WDEV void bin(uint32_t bit, uint32_t prob) {
#if LEP_ON_GPU
        const uint32_t s = __umulhi(q24, prob);
#else
        const uint32_t s = (uint32_t)(((uint64_t)q24 * prob) >> 32);
#endif
        const uint32_t q = q24 >> 24;
        const uint32_t n = bit ? q - s : s + 1;          // the new range before it is normalised (1 .. 255)
        const int f = __builtin_clz(n);                  // 24 .. 31
        q24 = (n << f) - (1u << 24);
        low += bit ? s + 1 : 0u;
        low <<= f - 24;
        count += f - 24;
    }
    // a carry out of the staged bytes: back through the 0xFF bytes in memory (the serial writer's ripple; rare)
    WDEV void ripple() {
        uint32_t x = pos;
        while (x > floor_ && (x > cap || out[x - 1] == 0xffu)) { if (x <= cap) out[x - 1] = 0; --x; }
        if (x > floor_) out[x - 1] = (uint8_t)(out[x - 1] + 1);
        else if (floor_) ++carry_out;   // (x == 0 in a whole stream: the carry of a code value that cannot be, as in the serial writer)
    }
    // every whole byte above the 24 + (count & 7) bits the recurrence still works on joins the staged ones -- a carry is just
    // the top bit of what is added -- and four staged bytes at a time go to memory
    WDEV void flush() {
        const int nb = count >= 0 ? (count >> 3) + 1 : 0;
        count -= 8 * nb;
        const int keep = 32 + count;
        const uint64_t o = low >> keep;
        low &= (1ull << keep) - 1;
        stage = (stage << (8 * nb)) + o;
        nst += nb;
        if ((stage >> (8 * nst)) & 1u) { stage &= (1ull << (8 * nst)) - 1; ripple(); }
        if (nst >= 4) {
            const uint32_t v = (uint32_t)(stage >> (8 * (nst - 4)));
            if (pos + 4 <= cap) { const uint32_t be = __builtin_bswap32(v); __builtin_memcpy(out + pos, &be, 4); }
            else for (int i = 0; i < 4; ++i) if (pos + (uint32_t)i < cap) out[pos + i] = (uint8_t)(v >> (24 - 8 * i));
            pos += 4; nst -= 4;
            stage &= (1ull << (8 * nst)) - 1;
        }
    }
    WDEV uint32_t finish(bool* overflow) {
        flush();
        for (int i = 0; i < 32; ++i) { bin(0, 128); if ((i & 3) == 3) flush(); }
        flush();
        for (; nst; --nst) { if (pos < cap) out[pos] = (uint8_t)(stage >> (8 * (nst - 1))); ++pos; }
        *overflow = pos >= cap;
        if (!*overflow && pos && (out[pos - 1] & 0xe0u) == 0xc0u) { out[pos] = 0; ++pos; }
        return pos;
    }
};

// bins: probability | bit << 8.
// part / nparts: the bins between two of emit's checkpoints (Ckpt5::nbins); the coder's state waits in the segment's arena between
// parts.  The start marker belongs to the first part, the stop bins and the stream's length to the last.
struct WriterState5 { uint64_t low, stage; uint32_t q24; int32_t count, nst; uint32_t pos; };
WDEV void write_wave(const SegPlan5* plans, const uint16_t* bins, const SegDev* segs, int seg0, int nseg, uint8_t* streams, uint32_t* stream_len,
                     int32_t* status, uint8_t* arena_base = nullptr, int part = 0, int nparts = 1) {
    LANES(l) {
        const int seg = seg0 + l;
        if (seg < nseg) {
            const SegPlan5& P = plans[seg];
            const SegDev& sd = segs[seg];
            const bool last = part + 1 >= nparts;
            if (P.status) { if (last) status[sd.slot] = P.status; }
            else {
                const uint16_t* list = bins + P.bins_off;                        // 256-byte aligned; room to the next multiple of 128 bins
                const uint32_t* b = reinterpret_cast<const uint32_t*>(list);
                const Ckpt5* ck = arena_base ? reinterpret_cast<const Ckpt5*>(arena_base + P.arena_off + P.ckpt_base) : nullptr;
                WriterState5* ws = arena_base ? reinterpret_cast<WriterState5*>(arena_base + P.arena_off + P.wstate_base) : nullptr;
                uint32_t i = part > 0 ? ck[part].nbins : 0u;
                const uint32_t n = last ? P.nbins : ck[part + 1].nbins;
                BoolEnc5 bc;
                bc.init(streams + sd.stream_off, sd.stream_cap);
                if (part == 0) bc.bin(0, 128);   // the start marker (vpx_start_encode)
                else { bc.low = ws->low; bc.stage = ws->stage; bc.q24 = ws->q24; bc.count = ws->count; bc.nst = ws->nst; bc.pos = ws->pos; }
                int since = 1;   // bins since the last flush (at most four: seven bits each above the 31 the recurrence keeps)
                for (; i < n && (i & 31u); ++i) {   // up to the next whole sector of the list
                    const uint32_t e = list[i];
                    bc.bin((e >> 8) & 1u, e & 255u);
                    if (++since >= 4) { bc.flush(); since = 0; }
                }
                bc.flush(); since = 0;
                // 32 bins (a 64-byte sector of this lane's list) per round: the NEXT round's four dwordx4 are requested before this
                // round's bins are coded -- a round is ~3,500 cycles of recurrence, enough to cover a trip to HBM.  The body has no
                // per-bin conditions; the last 0..31 bins take the loop behind it.  (The list has room to the next multiple of 128
                // bins, and the arena behind it: the look-ahead never leaves it.)
                if (i < n) {   // (i is a multiple of 32 here)
                U4 n0 = ld4(b + (i >> 1)), n1 = ld4(b + (i >> 1) + 4), n2 = ld4(b + (i >> 1) + 8), n3 = ld4(b + (i >> 1) + 12);
#define LEP5_CODE8(g)                                                                                          \
    bc.bin((g.x >> 8) & 1u, g.x & 255u); bc.bin((g.x >> 24) & 1u, (g.x >> 16) & 255u);                        \
    bc.bin((g.y >> 8) & 1u, g.y & 255u); bc.bin((g.y >> 24) & 1u, (g.y >> 16) & 255u);                        \
    bc.flush();                                                                                                \
    bc.bin((g.z >> 8) & 1u, g.z & 255u); bc.bin((g.z >> 24) & 1u, (g.z >> 16) & 255u);                        \
    bc.bin((g.w >> 8) & 1u, g.w & 255u); bc.bin((g.w >> 24) & 1u, (g.w >> 16) & 255u);                        \
    bc.flush();
                for (; i + 32 <= n; i += 32) {
                    const U4 g0 = n0, g1 = n1, g2 = n2, g3 = n3;
                    const uint32_t* nx = b + (i >> 1) + 16;
                    n0 = ld4(nx); n1 = ld4(nx + 4); n2 = ld4(nx + 8); n3 = ld4(nx + 12);
                    LEP5_CODE8(g0) LEP5_CODE8(g1) LEP5_CODE8(g2) LEP5_CODE8(g3)
                }
#undef LEP5_CODE8
                {
                    const uint32_t w[16] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, n3.x, n3.y, n3.z, n3.w};
                    for (int q = 0; i < n; ++i, ++q) {
                        const uint32_t e = (w[q >> 1] >> (16 * (q & 1))) & 0xffffu;
                        bc.bin((e >> 8) & 1u, e & 255u);
                        if ((q & 3) == 3) bc.flush();
                    }
                }
                }
                bc.flush();
                if (last) {
                    bool overflow = false;
                    stream_len[sd.slot] = bc.finish(&overflow);
                    if (overflow) status[sd.slot] = 100;   // LEP_BUFFER_TOO_SMALL
                } else { ws->low = bc.low; ws->stage = bc.stage; ws->q24 = bc.q24; ws->count = bc.count; ws->nst = bc.nst; ws->pos = bc.pos; }
            }
        }
    }
}


// ---- write, stitched: lane = a CHUNK of a segment's bins --------------------------------------------------------------------
// A launch of few segments leaves the lane-per-segment writer above with a handful of busy lanes on the whole chip and one serial
// chain of 2.3 million bins each (144 ms whatever the launch).  The stream is the big-endian number sum_i add_i * 2^(-T_i) (add_i =
// split, what a 1-bit adds to `low`; T_i = the shifts before bin i), so a stretch of bins can be coded by itself and ADDED in -- if
// the stretch knows (a) the range it starts with and (b) T at its start.  Both come from the range recurrence alone, which forgets
// its past: two runs over the same bins from different start ranges fall into step after some hundreds of bins (measured on real
// bin lists: median 450, 99th percentile 5,000).  So, with K chunks per segment:
//   range   lane = chunk: the recurrence alone (no `low`, no bytes) from kWarm5 bins before the chunk, started from an arbitrary
//           range; leaves the range it reaches the chunk with (a guess), the range it ends with and the chunk's sum of shifts;
//   link    lane = segment: walks its chunks -- a guess that is not what the chunk before ended with (rare) is redone from the
//           right range -- and turns the sums of shifts into every chunk's T, i.e. its first output byte and bit phase;
//   code    lane = chunk: the whole writer (BoolEnc5) from (range, T): bytes into the chunk's own stretch of the stream; what is
//           still in `low` when the bins run out (the next 24..31 bits) is kept aside, so is a carry that would leave the stretch;
//   stitch  lane = segment: adds every chunk's kept bits into the bytes behind it and the carries in front, ripples, applies the
//           stream's end rule (boolwriter.cc:17-35).
// The wavefront scan north_star names is here a scan over chunks: range states and bit offsets first, carries last.
constexpr uint32_t kWarm5 = 16384;
struct WChunk5 { uint32_t q_guess, q_end, shifts, q_start, t_start, keep, keep_bits, carry, pos_end, pad[7]; };   // 64 bytes
static_assert(sizeof(WChunk5) == 64, "one record per (segment, chunk)");
WDEV uint32_t chunk_first_bin(uint32_t nbins, int k, int K) { return (uint32_t)(((uint64_t)nbins * (uint32_t)k) / (uint32_t)K); }
// the range recurrence of BoolEnc5::bin alone; returns the shift
WDEV int range_step(uint32_t& q24, uint32_t bit, uint32_t prob) {
#if LEP_ON_GPU
    const uint32_t s = __umulhi(q24, prob);
#else
    const uint32_t s = (uint32_t)(((uint64_t)q24 * prob) >> 32);
#endif
    const uint32_t q = q24 >> 24;
    const uint32_t n = bit ? q - s : s + 1;
    const int f = __builtin_clz(n);
    q24 = (n << f) - (1u << 24);
    return f - 24;
}
// F(bit, prob) over bins [b0, b1) of a lane's list: whole 64-byte sectors (32 bins) as four 16-byte loads, the NEXT sector requested
// before this one is coded (a lane's bins are its own: nothing coalesces, and a round of 32 bins is about the time of a trip to HBM);
// the ragged head and tail bin by bin.  (A list has room to the next multiple of 128 bins: the look-ahead stays inside it.)
// g(): called at least after every fourth bin (the writer's flush: four bins are at most 28 bits).
template <class F, class G>
WDEV void for_bins(const uint16_t* list, uint32_t b0, uint32_t b1, F&& f, G&& g) {
    uint32_t i = b0;
    for (; i < b1 && (i & 31u); ++i) { const uint32_t e = list[i]; f((e >> 8) & 1u, e & 255u); g(); }
    if (i + 32 <= b1) {
        const uint32_t* b = reinterpret_cast<const uint32_t*>(list);
        U4 n0 = ld4(b + (i >> 1)), n1 = ld4(b + (i >> 1) + 4), n2 = ld4(b + (i >> 1) + 8), n3 = ld4(b + (i >> 1) + 12);
#define LEP5_BINS8(v)                                                                                      \
    f((v.x >> 8) & 1u, v.x & 255u); f((v.x >> 24) & 1u, (v.x >> 16) & 255u);                               \
    f((v.y >> 8) & 1u, v.y & 255u); f((v.y >> 24) & 1u, (v.y >> 16) & 255u); g();                          \
    f((v.z >> 8) & 1u, v.z & 255u); f((v.z >> 24) & 1u, (v.z >> 16) & 255u);                               \
    f((v.w >> 8) & 1u, v.w & 255u); f((v.w >> 24) & 1u, (v.w >> 16) & 255u); g();
        for (; i + 32 <= b1; i += 32) {
            const U4 g0 = n0, g1 = n1, g2 = n2, g3 = n3;
            const uint32_t* nx = b + (i >> 1) + 16;
            n0 = ld4(nx); n1 = ld4(nx + 4); n2 = ld4(nx + 8); n3 = ld4(nx + 12);
            LEP5_BINS8(g0) LEP5_BINS8(g1) LEP5_BINS8(g2) LEP5_BINS8(g3)
        }
#undef LEP5_BINS8
    }
    for (; i < b1; ++i) { const uint32_t e = list[i]; f((e >> 8) & 1u, e & 255u); g(); }
}
// bins [b0, b1) of `list` from range q24 (the start marker in front of bin 0 and the 32 stop bins behind the last one belong to the
// first and the last chunk): the range afterwards, the sum of the shifts
WDEV void range_run(const uint16_t* list, uint32_t b0, uint32_t b1, bool first, bool last, uint32_t& q24, uint32_t& shifts) {
    uint32_t t = 0, q = q24;
    if (first) t += (uint32_t)range_step(q, 0, 128);
    for_bins(list, b0, b1, [&](uint32_t bit, uint32_t prob) { t += (uint32_t)range_step(q, bit, prob); }, []() {});
    if (last) for (int i = 0; i < 32; ++i) t += (uint32_t)range_step(q, 0, 128);
    shifts = t; q24 = q;
}
WDEV void wchunk_range_lane(const SegPlan5& P, const uint16_t* bins, WChunk5* rec, int k, int K, uint32_t warm = kWarm5) {
    if (P.status) return;
    const uint16_t* list = bins + P.bins_off;
    const uint32_t b0 = chunk_first_bin(P.nbins, k, K), b1 = chunk_first_bin(P.nbins, k + 1, K);
    uint32_t q24 = 254u << 24, sh = 0;
    if (k > 0) {   // warm up: from kWarm5 bins before the chunk (or from the stream's true start, marker included, if that is nearer)
        const bool from_start = b0 <= warm;
        range_run(list, from_start ? 0u : b0 - warm, b0, from_start, false, q24, sh);
    }
    rec->q_guess = q24;
    range_run(list, b0, b1, k == 0, k + 1 == K, q24, sh);
    rec->q_end = q24; rec->shifts = sh;
}
// bytes the serial writer has handed out after a total shift of T, and its `count` then (BoolEnc5::flush: count starts at -24)
WDEV uint32_t bytes_at(uint32_t T) { return T >= 24 ? ((T - 24) >> 3) + 1 : 0u; }
WDEV int count_at(uint32_t T) { return (int)T - 24 - 8 * (int)bytes_at(T); }
WDEV void wchunk_link_lane(const SegPlan5& P, const uint16_t* bins, WChunk5* recs, int K) {
    if (P.status) return;
    const uint16_t* list = bins + P.bins_off;
    uint32_t T = 0, q = 254u << 24;
    for (int k = 0; k < K; ++k) {
        WChunk5& r = recs[k];
        if (k > 0 && r.q_guess != q) {   // the warm-up had not fallen into step: this chunk's range run again, from the range it really starts with
            uint32_t q24 = q, sh = 0;
            range_run(list, chunk_first_bin(P.nbins, k, K), chunk_first_bin(P.nbins, k + 1, K), false, k + 1 == K, q24, sh);
            r.q_end = q24; r.shifts = sh;
        }
        r.q_start = q; r.t_start = T;
        q = r.q_end; T += r.shifts;
    }
}
WDEV void wchunk_code_lane(const SegPlan5& P, const uint16_t* bins, const SegDev& sd, uint8_t* streams, WChunk5* rec, int k, int K) {
    if (P.status) return;
    const uint16_t* list = bins + P.bins_off;
    const uint32_t b0 = chunk_first_bin(P.nbins, k, K), b1 = chunk_first_bin(P.nbins, k + 1, K);
    BoolEnc5 bc;
    bc.init(streams + sd.stream_off, sd.stream_cap);
    bc.q24 = rec->q_start;
    bc.pos = bytes_at(rec->t_start); bc.count = count_at(rec->t_start); bc.floor_ = bc.pos;
    if (k == 0) { bc.bin(0, 128); bc.flush(); }
    for_bins(list, b0, b1, [&](uint32_t bit, uint32_t prob) { bc.bin(bit, prob); }, [&]() { bc.flush(); });
    bc.flush();
    if (k + 1 == K) for (int i = 0; i < 32; ++i) { bc.bin(0, 128); if ((i & 3) == 3) bc.flush(); }
    bc.flush();
    for (; bc.nst; --bc.nst) { if (bc.pos < bc.cap) bc.out[bc.pos] = (uint8_t)(bc.stage >> (8 * (bc.nst - 1))); ++bc.pos; }
    // what the recurrence still holds: the 32 + count bits behind the last byte handed out (count is -8 .. -1 after a flush)
    rec->keep = (uint32_t)bc.low; rec->keep_bits = (uint32_t)(32 + bc.count); rec->carry = bc.carry_out; rec->pos_end = bc.pos;
}
// adds `v` (a carry) at byte x - 1 and below
WDEV void add_back(uint8_t* out, uint32_t cap, uint32_t x, uint32_t v) {
    while (v && x > 0) {
        --x;
        if (x < cap) { const uint32_t t = out[x] + v; out[x] = (uint8_t)t; v = t >> 8; }
        else v = 0;   // (beyond the buffer nothing is stored: the stream is reported as too long anyway)
    }
}
WDEV void wchunk_stitch_lane(const SegPlan5& P, const SegDev& sd, uint8_t* streams, uint32_t* stream_len, int32_t* status, const WChunk5* recs, int K) {
    if (P.status) { status[sd.slot] = P.status; return; }
    uint8_t* out = streams + sd.stream_off;
    const uint32_t cap = sd.stream_cap;
    for (int k = 0; k + 1 < K; ++k) {
        const WChunk5& r = recs[k];
        // chunk k's kept bits are the next `keep_bits` bits of the stream from byte pos_end on: a big-endian number of four bytes there
        const uint32_t X = r.keep_bits ? r.keep << (32 - r.keep_bits) : 0u;
        uint32_t carry = 0;
        for (int j = 3; j >= 0; --j) {
            const uint32_t x = r.pos_end + (uint32_t)j, add = (X >> (24 - 8 * j)) & 255u;
            if (x < cap) { const uint32_t t = out[x] + add + carry; out[x] = (uint8_t)t; carry = t >> 8; }
            else carry = 0;
        }
        add_back(out, cap, r.pos_end, carry);
        add_back(out, cap, recs[k + 1].t_start >= 24 ? bytes_at(recs[k + 1].t_start) : 0u, recs[k + 1].carry);   // chunk k+1's carries that left its stretch
    }
    uint32_t pos = recs[K - 1].pos_end;
    const bool overflow = pos >= cap;
    if (!overflow && pos && (out[pos - 1] & 0xe0u) == 0xc0u) { out[pos] = 0; ++pos; }
    stream_len[sd.slot] = pos;
    if (overflow) status[sd.slot] = 100;   // LEP_BUFFER_TOO_SMALL
}

// ---- the walk: count / emit / gather -------------------------------------------------------------------------------------
// One wavefront per segment; a TILE = up to 64 consecutive coded blocks of one block row; lane = block.  Per tile:
//   phase 1  lane = block: the block's own numbers (non-zero counts, where its sign bytes / threshold units / bins start:
//            wave scans), IDCT + neighbour summary, then the contexts that need the neighbours' summaries (DC prediction)
//   phase R  row by row in stream order: every lane works out what its coefficient of that row contributes (priors, classes);
//            the lanes of one (row, class) are ranked in block order with ballots -- that is the entry's place in its stream;
//            emit writes the units there (and sign bytes, threshold units, the sparse records), gather reads the probabilities
//            from the same places and appends the block's bins.  count only adds the units up (LDS atomics: no order needed).
enum { kCount = 0, kEmit = 1, kGather = 2 };

// (members in the order the passes need them: a launch takes the front of the block its pass uses -- gather 18 KB (its staging
// block lies over the cursors the other two passes keep there), count 15 KB, emit all 26 KB -- and the resident workgroups per CU
// follow from that)
struct Walk5Shared {
    uint32_t cur[32 * 65];                 // transposed tile: dword i (coefficients 2i, 2i + 1 in aligned order) of lane c at [i * 65 + c];
                                           // column 64 = the block left of lane 0 (the previous tile's last one)
    uint16_t TB[8 * 65];                   // where the lane's next threshold unit of class lt goes (relative to the tile's first)
    uint16_t q[64];                        // the component's quantisation-derived tables (a load from the image descriptor on the
    uint8_t thr[64];                       // critical path costs a trip to HBM: they are staged when the component changes)
    uint16_t errx[2 * 64];                 // emit: the lanes' first refusal per half of the block (interior; edges and DC)
    uint32_t tcur[2 * 8];                  // emit / gather: threshold units already given out, per colour index / class lt
    union {
        // ---- gather: per wavefront, the 32 bins (one 64-byte sector of the bin list) every lane is filling: dword d of lane l at [d * 64 + l]
        uint32_t stg[2][16 * 64];
        struct {
            // ---- count and emit
            uint32_t cursor[2 * kRows * kClasses]; // units already given out, per colour index / row / class (row 63 = threshold: count only)
            // ---- emit
            NSum ns[65];                           // neighbour summaries of the tile's blocks; [64] = the block left of lane 0
            int32_t icos_x[64], icos_y[64];
            uint32_t abv[32 * 65];                 // the tile of the row above
        };
    };
};
constexpr size_t kWalkLdsGather = offsetof(Walk5Shared, stg) + sizeof(uint32_t) * 2 * 16 * 64, kWalkLdsCount = offsetof(Walk5Shared, ns);
constexpr size_t walk_lds_bytes(int mode) { return mode == 2 ? kWalkLdsGather : (mode == 0 ? kWalkLdsCount : sizeof(Walk5Shared)); }

// The walk's LDS block.  On the GPU it is the kernel's dynamic LDS, named directly at every use: a pointer to it kept in the
// walker object loses its address space as soon as that object's address is taken anywhere, and every access then becomes a
// FLAT instruction (measured: 441 flat against 10 ds instructions in the gather kernel, twice the wave time).
#if LEP_ON_GPU
extern __shared__ __attribute__((aligned(16))) unsigned char lep5_lds[];
#define LEP5_WSH(self) (*reinterpret_cast<lep5::Walk5Shared*>(lep5::lep5_lds))
#else
#define LEP5_WSH(self) (*(self)->sh)
#endif

// Global memory through pointers whose address space the compiler cannot see (they come out of descriptors in memory): a plain
// dereference is a FLAT instruction, which counts against the LDS wait counter as well -- every wait for an LDS read would then
// also wait for the tile prefetch in flight.  gld / gst name the address space.
#if LEP_ON_GPU
template <class T> WDEV T gld(const T* p) { return *(const __attribute__((address_space(1))) T*)(uintptr_t)p; }
template <class T> WDEV void gst(T* p, const T& v) { *(__attribute__((address_space(1))) T*)(uintptr_t)p = v; }
#else
template <class T> WDEV T gld(const T* p) { return *p; }
template <class T> WDEV void gst(T* p, const T& v) { *p = v; }
#endif

WDEV void gst4(uint32_t* p, uint32_t x, uint32_t y, uint32_t z, uint32_t w) {   // one 16-byte store (p: 16-byte aligned)
#if LEP_ON_GPU
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    *(__attribute__((address_space(1))) u32x4*)(uintptr_t)p = u32x4{x, y, z, w};
#else
    p[0] = x; p[1] = y; p[2] = z; p[3] = w;
#endif
}
WDEV NSum gld_ns(const NSum* p) {   // (a struct cannot be assigned through an address-space pointer: nine dwords)
    uint32_t w[sizeof(NSum) / 4];
    for (int i = 0; i < (int)(sizeof(NSum) / 4); ++i) w[i] = gld(reinterpret_cast<const uint32_t*>(p) + i);
    NSum r;
    __builtin_memcpy(&r, w, sizeof r);
    return r;
}
WDEV void gst_ns(NSum* p, const NSum& v) {
    uint32_t w[sizeof(NSum) / 4];
    __builtin_memcpy(w, &v, sizeof v);
    for (int i = 0; i < (int)(sizeof(NSum) / 4); ++i) gst(reinterpret_cast<uint32_t*>(p) + i, w[i]);
}

WDEV int lane_prefix(uint64_t m, int l) {   // set bits of m below lane l
#if LEP_ON_GPU
    (void)l;
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#else
    return __builtin_popcountll(m & ((1ull << l) - 1));
#endif
}
WDEV void lds_add(uint32_t* p, uint32_t v) {
#if LEP_ON_GPU
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    *p += v;
#endif
}

// kNzBin for 0 <= left <= 49 without a memory access
WDEV int nzbin5(int left) { return left < 16 ? (int)((0x7776666555443210ull >> (4 * left)) & 15) : (left < 21 ? 7 : (left < 32 ? 8 : 9)); }
WDEV int tile_get(const uint32_t* T, int a, int col) { return (int16_t)(T[(a >> 1) * 65 + col] >> ((a & 1) * 16)); }

// ---- bucket: the block-ordered sparse records -> one stream per chain --------------------------------------------------------
// One wavefront per segment, lane = block (64 consecutive ordinals at a time).  First pass: records per chain (LDS atomics).
// Second pass: the lanes of one chain are ranked in block order with ballots, as the walk ranks a row's entries; the record goes
// to its chain's stream, the place into place[kind][ordinal] for gather.
struct BucketShared { uint32_t cnt[kNumCls], cur[kNumCls]; };
WDEV void bucket_wave(const SegPlan5* P, uint8_t* arena_base, BucketShared* sh) {
    if (P->status) return;
    uint8_t* arena = arena_base + P->arena_off;
    const uint32_t nb = P->nblocks;
    const uint32_t* keys = reinterpret_cast<const uint32_t*>(arena + P->key_base);
    const uint32_t* nzr = reinterpret_cast<const uint32_t*>(arena + P->nz_base);
    const uint32_t* enr = reinterpret_cast<const uint32_t*>(arena + P->en_base);
    const uint32_t* dcr = reinterpret_cast<const uint32_t*>(arena + P->dc_base);
    uint32_t* ctab = reinterpret_cast<uint32_t*>(arena + P->cls_base);
    uint32_t* place = reinterpret_cast<uint32_t*>(arena + P->place_base);
    uint32_t* nzs = reinterpret_cast<uint32_t*>(arena + P->nzs_base);
    uint32_t* ens[2] = {reinterpret_cast<uint32_t*>(arena + P->ens_base[0]), reinterpret_cast<uint32_t*>(arena + P->ens_base[1])};
    uint32_t* dcs = reinterpret_cast<uint32_t*>(arena + P->dcs_base);
    LANES(l) sh->cnt[l] = 0;
    LSYNC();
    for (uint32_t b0 = 0; b0 < nb; b0 += 64) {
        LANES(l) if (b0 + l < nb) {
            int c[4];
            key_classes(gld(keys + b0 + l), c);
            for (int q = 0; q < 4; ++q) lds_add(&sh->cnt[c[q]], 1u);
        }
    }
    LSYNC();
    LANES(l) if (l < 4) {   // a chain's stream starts where the ones of its kind before it end (each padded to four records)
        const int lo = l == 0 ? kClsNz : (l == 1 ? kClsEh : (l == 2 ? kClsEv : kClsDc)), hi = l == 0 ? kClsEh : (l == 1 ? kClsEv : (l == 2 ? kClsDc : kNumCls));
        uint32_t at = 0;
        for (int c = lo; c < hi; ++c) { sh->cur[c] = at; at += (sh->cnt[c] + 3u) & ~3u; }
    }
    LSYNC();
    LANES(l) {
        const uint32_t first = sh->cur[l], n = sh->cnt[l];
        gst(ctab + l, first); gst(ctab + kNumCls + l, n);
        for (uint32_t j = first + n; j < first + ((n + 3u) & ~3u); ++j) {   // the padding: records that fold to nothing out of bounds
            if (l < kClsEh) { gst(nzs + 2 * j, 0u); gst(nzs + 2 * j + 1, 0u); }
            else if (l < kClsEv) gst(ens[0] + j, 0u);
            else if (l < kClsDc) gst(ens[1] + j, 0u);
            else gst(dcs + 6 * j, 0u);
        }
    }
    LSYNC();
    for (uint32_t b0 = 0; b0 < nb; b0 += 64) {
        LV(int, act); LV(int, cls); LV(uint32_t, at);
        LV(uint32_t, key);
        LANES(l) { L(act) = b0 + l < nb; L(key) = L(act) ? gld(keys + b0 + l) : 0u; }
        for (int q = 0; q < 4; ++q) {
            LANES(l) { int c[4]; key_classes(L(key), c); L(cls) = c[q]; }
            uint64_t rem = lepwave::wave_ballot(act);
            while (rem) {
                const int c0 = (int)lepwave::wave_read((const uint32_t*)cls, __builtin_ctzll(rem));
                const uint32_t first = sh->cur[c0];
                LV(int, g);
                LANES(l) L(g) = L(act) && L(cls) == c0;
                const uint64_t m = lepwave::wave_ballot(g);
                LANES(l) if (L(g)) L(at) = first + (uint32_t)lane_prefix(m, l);
                LSYNC();
                LANES(l) if (l == 0) sh->cur[c0] = first + (uint32_t)lepwave::popc64(m);
                LSYNC();
                rem &= ~m;
            }
            LANES(l) if (L(act)) {
                const uint32_t b = b0 + l, j = L(at);
                gst(place + (size_t)q * nb + b, j);
                if (q == 0) gst(nzs + 2 * j, gld(nzr + 2 * b));
                else if (q < 3) gst(ens[q - 1] + j, gld(enr + 2 * b + (q - 1)));
                else gst(dcs + 6 * j, gld(dcr + 6 * b));
            }
        }
    }
}

// NW = 2: two wavefronts walk a segment together -- both see the same tile in LDS; wavefront 0 codes the 7x7 interiors, wavefront
// 1 everything that needs the neighbours' pixels and the block's sparse records (edges, DC, the three counts).  Each half has its
// own streams, its own run of sign bytes and its own run of bins inside every block, so the halves only meet at the tile's
// boundaries.  (Compiled without a GPU, NW = 2 runs the two halves one after the other: the same split, checked bit for bit.)
#define LEP5_XSYNC() do { if (NW > 1) WSYNC(); else LSYNC(); } while (0)
template <int MODE, int NW = 1>
struct Walk5 {
    static constexpr int kLoadWaves = LEP_ON_GPU ? NW : 1;   // wavefronts that share a tile's loads
    int wave;                  // this wavefront's number inside the workgroup (0 when there is one)
    const ImageDev* img;
    Walk5Shared* sh;
    const SegPlan5* plan;      // emit / gather
    uint8_t* arena;            // segment's entry arena (emit / gather)
    uint16_t* bins;            // segment's bin list (gather)
    int comp, ci;
    uint32_t ord0;             // block ordinal of lane 0 of the current tile
    uint32_t sign_pos[2];      // sign bytes given out per colour index
    uint32_t nbins;            // gather: bins written; count: bins an encoder will need (upper bound through the DC term)
    int status;
    uint32_t tile_no;          // tiles walked so far
    uint32_t* AT;              // emit / gather: the segment's [tile][64][64] array of places
    uint32_t sign_base[2], key_base, nz_base, en_base, dc_base;   // the plan's offsets (read once: a load per tile from the plan would sit on the critical path)
    uint32_t place_base, nzs_base, ens_base[2], dcs_base, plan_nblocks;

    WDEV uint32_t* units() const { return reinterpret_cast<uint32_t*>(arena); }

    // A tile's coefficients travel global memory -> registers -> LDS: the loads of the NEXT tile are issued before the current one
    // is worked on (fetch_tile), and written to the transposed arrays when its turn comes (store_tile).
    struct TileDesc {
        int comp, x0, nb, yb;
        bool has_above;
        const int16_t *row, *arow;
        NSum* nrow;
        const NSum* narow;
    };
    struct TileRegs { uint32_t c[32], a[32]; };
    struct TileIter {   // (an object of its own, not a lambda over the walker: see LEP5_WSH)
        const ImageDev* img;
        SegDev seg;
        NSum* ns;
        bool top[3];
        uint32_t idx;
        int row_x0, row_blocks;
        TileDesc rowd;
        WDEV void init(const ImageDev* image, const SegDev& s, NSum* n) {
            img = image; seg = s; ns = n; top[0] = top[1] = top[2] = true; idx = 0; row_x0 = 0; row_blocks = 0; rowd = TileDesc{};
        }
        WDEV bool next(TileDesc* t) {
            for (;;) {
                if (row_x0 < row_blocks) {
                    *t = rowd;
                    t->x0 = row_x0;
                    t->nb = row_blocks - row_x0 < 64 ? row_blocks - row_x0 : 64;
                    row_x0 += 64;
                    return true;
                }
                RowSpec r = row_spec(img, idx++);
                if (r.done) return false;
                if (r.luma_y >= seg.y1 && !seg.is_last) return false;
                if (r.skip) continue;
                if (r.luma_y < seg.y0) continue;
                const int cmp = r.component, w = img->width[cmp], yb = r.curr_y;
                rowd.comp = cmp; rowd.yb = yb;
                rowd.row = img->blocks[cmp] + (int64_t)yb * w * 64;
                rowd.has_above = !top[cmp];
                rowd.arow = rowd.has_above ? rowd.row - (int64_t)w * 64 : nullptr;
                rowd.nrow = ns + img->ns_offset[cmp] + (yb & 1) * w;
                rowd.narow = ns + img->ns_offset[cmp] + ((yb & 1) ^ 1) * w;
                top[cmp] = false;
                int nbk = img->coded_blocks[cmp] - yb * w;
                row_blocks = nbk < 1 ? 1 : (nbk > w ? w : nbk);
                row_x0 = 0;
            }
        }
    };
    WDEV void fetch_tile(const TileDesc& t, TileRegs* regs) const {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(t.row + (int64_t)t.x0 * 64);
        const uint32_t* asrc = t.arow ? reinterpret_cast<const uint32_t*>(t.arow + (int64_t)t.x0 * 64) : nullptr;
        LANES(l) {
#pragma unroll
            for (int kk = 0; kk < 32 / kLoadWaves; ++kk) {
                const int d = (kk + wave * (32 / kLoadWaves)) * 64 + l, b = d >> 5;
                regs[LEP_LI(l)].c[kk] = b < t.nb ? gld(src + d) : 0u;
                regs[LEP_LI(l)].a[kk] = (asrc && b < t.nb && MODE == kEmit) ? gld(asrc + d) : 0u;
            }
        }
    }
    WDEV void store_tile(const TileDesc& t, const TileRegs* regs) {
        Walk5Shared& S = LEP5_WSH(this);
        const bool first_of_row = t.x0 == 0;
        const bool new_comp = t.comp != comp;
        if (new_comp) { comp = t.comp; ci = comp ? 1 : 0; }
        LEP5_XSYNC();   // (every wavefront is done with the previous tile)
        if (wave == 0) {
            if (new_comp) {   // the component's tables
                LANES(l) {
                    S.q[l] = img->q[comp][l]; S.thr[l] = img->min_thresh[comp][l];
                    if (MODE == kEmit) { S.icos_x[l] = img->icos_x[comp][l]; S.icos_y[l] = img->icos_y[comp][l]; }
                }
            }
            LANES(l) {   // keep the last column of the previous tile as "left of lane 0"
                if (l < 32) {
                    S.cur[l * 65 + 64] = first_of_row ? 0u : S.cur[l * 65 + 63];
                    if (MODE == kEmit) S.abv[l * 65 + 64] = first_of_row ? 0u : S.abv[l * 65 + 63];
                }
                if (MODE == kEmit && l == 0) { if (first_of_row) S.ns[64] = NSum{}; else S.ns[64] = S.ns[63]; }
            }
        }
        LEP5_XSYNC();
        LANES(l) {
#pragma unroll
            for (int kk = 0; kk < 32 / kLoadWaves; ++kk) {
                const int d = (kk + wave * (32 / kLoadWaves)) * 64 + l, b = d >> 5, i = d & 31;
                S.cur[i * 65 + b] = regs[LEP_LI(l)].c[kk];
                if (MODE == kEmit) S.abv[i * 65 + b] = regs[LEP_LI(l)].a[kk];
            }
        }
        LEP5_XSYNC();
    }

    // one tile; has_above: the row above belongs to this segment; returns 0 or an exit code
    struct TileTotals { int nsig, bins; };
    WDEV int tile(int x0, int nb, bool has_above, NSum* nrow, const NSum* narow) {
        Walk5Shared& S = LEP5_WSH(this);
        TileTotals tt;
        if (NW == 1) tt = half<3>(x0, nb, has_above, nrow, narow);
        else if (!LEP_ON_GPU) { half<1>(x0, nb, has_above, nrow, narow); tt = half<2>(x0, nb, has_above, nrow, narow); }
        else if (wave == 0) tt = half<1>(x0, nb, has_above, nrow, narow);
        else tt = half<2>(x0, nb, has_above, nrow, narow);
        if (MODE == kEmit) {   // the first refusal in stream order ends the segment (lane order = block order; inside a block: interior, edges, DC)
            LEP5_XSYNC();
            LV(int, err); LV(int, bad);
            LANES(l) { L(err) = S.errx[l] ? S.errx[l] : (NW > 1 ? S.errx[64 + l] : 0); L(bad) = L(err) != 0; }
            const uint64_t bm = lepwave::wave_ballot(bad);
            if (bm) return (int)(lepwave::wave_read((const uint32_t*)err, __builtin_ctzll(bm)) & 0xffff);
        }
        sign_pos[ci] += (uint32_t)tt.nsig;
        nbins += (uint32_t)tt.bins;
        ord0 += (uint32_t)nb;
        ++tile_no;
        return 0;
    }
    // HALF: 1 = the 7x7 interiors, 2 = the sparse records, the edges and the DC, 3 = the whole block
    template <int HALF>
    WDEV TileTotals half(int x0, int nb, bool has_above, NSum* nrow, const NSum* narow) {
        Walk5Shared& S = LEP5_WSH(this);
        constexpr bool kInt = (HALF & 1) != 0, kEdge = (HALF & 2) != 0;
        const int c = comp;
        LV(int, act); LV(int, nz); LV(int, neh); LV(int, nev); LV(int, nsig); LV(int, err); LV(int, errdc);
        LV(int, eobx); LV(int, eoby);
        LV(int32_t, dc_e0);       // DC entry (unit 0)
        LV(int, dc_sign);         // sign byte of the DC (0 = no sign bin)
        LV(int, nzctxbin);
        LV(int, lbins);           // bins of this block
        LV(int, ibins);           // ... of its 7x7 interior
        LV(NSum, nsa);            // the summary of the block above

        // gather: the block's sparse records (7x7 count, edge counts, DC units) are requested now and used after phase 1
        LV(uint32_t, rdce); LV(uint32_t, rdcp);
        LV(uint32_t, rnz0); LV(uint32_t, rnz1); LV(uint32_t, ren0); LV(uint32_t, ren1); LV(uint32_t, rdc0); LV(uint32_t, rdc1); LV(uint32_t, rdc2);
        if (MODE == kGather) {
            LANES(l) if (l < nb) {
                const uint32_t* pl = reinterpret_cast<const uint32_t*>(arena + place_base) + ord0 + l;   // the block's records, where bucket put them
                const uint32_t nbl = plan_nblocks;
                if (kInt) {    // (the six bins of the 7x7 count open the interior half's run)
                    const uint32_t* r1 = reinterpret_cast<const uint32_t*>(arena + nzs_base) + 2 * (size_t)gld(pl);
                    L(rnz0) = gld(r1); L(rnz1) = gld(r1 + 1);
                }
                if (kEdge) {
                    const uint32_t p1 = gld(pl + nbl), p2 = gld(pl + 2 * (size_t)nbl), p3 = gld(pl + 3 * (size_t)nbl);
                    const uint32_t* r3 = reinterpret_cast<const uint32_t*>(arena + dcs_base) + 6 * (size_t)p3;
                    L(ren0) = gld(reinterpret_cast<const uint32_t*>(arena + ens_base[0]) + p1); L(ren1) = gld(reinterpret_cast<const uint32_t*>(arena + ens_base[1]) + p2);
                    L(rdc0) = gld(r3); L(rdc1) = gld(r3 + 1); L(rdc2) = gld(r3 + 2);
                    L(rdcp) = p3;
                }
                L(rdce) = gld(AT + ((size_t)tile_no * kAtRows + 63) * 64 + l);   // the DC's entry as emit worked it out, its sign in bit 31 (both halves: the DC's bins are part of the block's)
            }
        }
        // ---- phase 1a: own numbers ---------------------------------------------------------------------------------
        LANES(l) {
            const int a = l < nb;
            L(act) = a; L(err) = 0; L(errdc) = 0;
            int n7 = 0, nh = 0, nv = 0, ex = 0, ey = 0, lb = 0, ib = 0;
            if (kEdge) for (int t = 0; t < 8; ++t) S.TB[t * 65 + l] = 0;
            if (a) {
                // bins of the block that do not depend on its neighbours: 6 + 3 + 3 count bins, and per coded coefficient the
                // exponent bins, the sign and len - 1 residual bins; zeros in front of a region's last non-zero cost one bin
                int last = -1;
                for (int z = 0; z < 49; ++z) {
                    const int v = iabs(tile_get(S.cur, z, l));
                    if (v) {
                        ++n7; last = z;
                        const int coord = kA2R[z]; ex = ex > (coord & 7) ? ex : (coord & 7); ey = ey > (coord >> 3) ? ey : (coord >> 3);
                        const int len = bitlen((uint32_t)v), lc = len > 11 ? 11 : len;
                        lb += (lc < 11 ? lc + 1 : 11) + lc;
                    }
                }
                lb += last + 1 - n7;
                ib = lb;
                lb += 12;
                for (int eg = 0; eg < 2; ++eg) {
                    int lastj = -1, cnt = 0;
                    for (int j = 0; j < 7; ++j) {
                        const int v = iabs(tile_get(S.cur, (eg ? 57 : 50) + j, l));
                        if (!v) continue;
                        ++cnt; lastj = j;
                        const int len = bitlen((uint32_t)v), lc = len > 11 ? 11 : len;
                        lb += (lc < 11 ? lc + 1 : 11) + lc;
                        const int thr = S.thr[eg ? (j + 1) * 8 : j + 1];
                        if (kEdge && lc > 1 && lc - 2 >= thr) {   // threshold units of this coefficient (class lt = min(len - thr, 7))
                            const int lt = imin(lc - thr, 7), un = (lc - 1 - thr + 3) >> 2;
                            if (MODE == kCount) lds_add(&S.cursor[stream_id(ci, 63, lt)], (uint32_t)un);
                            else S.TB[lt * 65 + l] = (uint16_t)(S.TB[lt * 65 + l] + un);
                        }
                    }
                    lb += lastj + 1 - cnt;
                    if (eg) nv = cnt; else nh = cnt;
                }
            }
            L(nz) = n7; L(neh) = nh; L(nev) = nv; L(eobx) = ex; L(eoby) = ey; L(lbins) = lb; L(ibins) = ib;
            L(nsig) = a ? n7 + nh + nv + 1 : 0;
            if (MODE == kEmit && kEdge && has_above && a) L(nsa) = gld_ns(&narow[x0 + l]);   // (written when that row was walked)
            else L(nsa) = NSum{};
        }
        // ---- phase 1b: IDCT without DC, neighbour summary (block_context.hh:44-78) -------------------------------------
        struct Px { int16_t r0[8], r1[8], c0[8], c1[8]; };   // pixel rows 0, 1 and columns 0, 1 of the block without its DC
        LV(Px, px);
        if (MODE == kEmit && kEdge) {
            LANES(l) if (L(act)) {
                const uint16_t* q = S.q;
                constexpr int w1 = 2841, w2 = 2676, w3 = 2408, w5 = 1609, w6 = 1108, w7 = 565, r2 = 181;
                constexpr int w1pw7 = w1 + w7, w1mw7 = w1 - w7, w2pw6 = w2 + w6, w2mw6 = w2 - w6, w3pw5 = w3 + w5, w3mw5 = w3 - w5;
                int32_t t[64];
                int16_t pix[64];
#pragma unroll
                for (int y = 0; y < 8; ++y) {
                    const int y8 = y * 8;
#define LEP5_CQ(i) ((int32_t)tile_get(S.cur, kR2A[i], l) * (int32_t)q[i])
                    int32_t x0_ = (y == 0 ? 0 : (int32_t)((uint32_t)LEP5_CQ(y8) << 11)) + 128;
                    int32_t x1 = (int32_t)((uint32_t)LEP5_CQ(y8 + 4) << 11);
                    int32_t x2 = LEP5_CQ(y8 + 6), x3 = LEP5_CQ(y8 + 2), x4 = LEP5_CQ(y8 + 1), x5 = LEP5_CQ(y8 + 7), x6 = LEP5_CQ(y8 + 5), x7 = LEP5_CQ(y8 + 3), x8;
#undef LEP5_CQ
                    x8 = w7 * (x4 + x5); x4 = x8 + w1mw7 * x4; x5 = x8 - w1pw7 * x5;
                    x8 = w3 * (x6 + x7); x6 = x8 - w3mw5 * x6; x7 = x8 - w3pw5 * x7;
                    x8 = x0_ + x1; x0_ -= x1;
                    x1 = w6 * (x3 + x2); x2 = x1 - w2pw6 * x2; x3 = x1 + w2mw6 * x3;
                    x1 = x4 + x6; x4 -= x6; x6 = x5 + x7; x5 -= x7;
                    x7 = x8 + x3; x8 -= x3; x3 = x0_ + x2; x0_ -= x2;
                    x2 = (r2 * (x4 + x5) + 128) >> 8;
                    x4 = (r2 * (x4 - x5) + 128) >> 8;
                    t[y8 + 0] = (x7 + x1) >> 8; t[y8 + 1] = (x3 + x2) >> 8; t[y8 + 2] = (x0_ + x4) >> 8; t[y8 + 3] = (x8 + x6) >> 8;
                    t[y8 + 4] = (x8 - x6) >> 8; t[y8 + 5] = (x0_ - x4) >> 8; t[y8 + 6] = (x3 - x2) >> 8; t[y8 + 7] = (x7 - x1) >> 8;
                }
#pragma unroll
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
                    pix[x] = (int16_t)((y7 + y1) >> 11); pix[8 + x] = (int16_t)((y3 + y2) >> 11);
                    pix[16 + x] = (int16_t)((y0 + y4) >> 11); pix[24 + x] = (int16_t)((y8 + y6) >> 11);
                    pix[32 + x] = (int16_t)((y8 - y6) >> 11); pix[40 + x] = (int16_t)((y0 - y4) >> 11);
                    pix[48 + x] = (int16_t)((y3 - y2) >> 11); pix[56 + x] = (int16_t)((y7 - y1) >> 11);
                }
                const int dcq = tile_get(S.cur, 49, l) * (int)q[0];
                NSum& me = S.ns[l];
                for (int i = 0; i < 8; ++i) {
                    me.horiz[i] = (int16_t)(dcq + pix[56 + i] + 1024 + (int16_t)(pix[56 + i] - pix[48 + i]) / 2);
                    me.vert[i] = (int16_t)(dcq + pix[i * 8 + 7] + 1024 + (int16_t)(pix[i * 8 + 7] - pix[i * 8 + 6]) / 2);
                    L(px).r0[i] = pix[i]; L(px).r1[i] = pix[8 + i]; L(px).c0[i] = pix[i * 8]; L(px).c1[i] = pix[i * 8 + 1];
                }
                me.nz = L(nz);
                gst_ns(&nrow[x0 + l], me);   // for the row below
            }
            LSYNC();
        }
        // ---- phase 1c: contexts that need the neighbours' summaries: 7x7 non-zero context, DC prediction (model.hh:463-485, 674-832)
        LANES(l) {
            int32_t e0 = 0; int sgn = 0, bins_here = 0, ctxbin = 0;   // bins_here: the DC's bins
            if (L(act)) {
                const bool has_left = x0 + l > 0;
                const int dc = tile_get(S.cur, 49, l);
                if (MODE == kGather) {   // what emit worked out: the DC entry and its sign byte
                    e0 = (int32_t)(L(rdce) & 0x7fffffffu); sgn = (int)((L(rdce) >> 31) << 6);
                    const int len = (e0 >> 14) & 15, nres = (e0 >> 10) & 15;
                    bins_here = (len < 11 ? len + 1 : 11) + (len ? 1 : 0) + nres;
                } else if (MODE == kEmit) {
                  if (kEdge) {
                    const NSum& nl = S.ns[(l + 64) % 65];
                    const NSum& na = L(nsa);
                    int nzctx = 0;
                    if (has_left && has_above) nzctx = (na.nz + nl.nz + 2) / 4;
                    else if (has_above) nzctx = (na.nz + 1) / 2;
                    else if (has_left) nzctx = (nl.nz + 1) / 2;
                    ctxbin = nzbin5(nzctx);
                    int32_t avgmed = 0, unc = 0, unc2 = 0;
                    if (has_left || has_above) {
                        int sum0 = 0, sum1 = 0, mn = 0x7fffffff, mx = -0x7fffffff, n = 0;
                        if (has_left)
                            for (int i = 0; i < 8; ++i, ++n) {
                                const int ev = (int16_t)(nl.vert[i] - (int16_t)(L(px).c0[i] - L(px).c1[i]) / 2 - (L(px).c0[i] + 1024));
                                sum0 += ev; mn = ev < mn ? ev : mn; mx = ev > mx ? ev : mx;
                            }
                        if (has_above)
                            for (int i = 0; i < 8; ++i, ++n) {
                                const int ev = (int16_t)(na.horiz[i] - (int16_t)(L(px).r0[i] - L(px).r1[i]) / 2 - (L(px).r0[i] + 1024));
                                if (has_left) sum1 += ev; else sum0 += ev;
                                mn = ev < mn ? ev : mn; mx = ev > mx ? ev : mx;
                            }
                        if (n == 8) sum1 = sum0;
                        avgmed = (sum0 + sum1) >> 1;
                        unc = (mx - mn) >> 3;
                        sum0 -= avgmed; sum1 -= avgmed;
                        unc2 = (iabs(sum0) < iabs(sum1) ? sum0 : sum1) >> 3;
                    }
                    const int pred = (avgmed / (int)S.q[0] + 4) >> 3;
                    const int a = imin(bitlen((uint32_t)iabs(unc) & 0xffff), 11), b17 = imin(bitlen((uint32_t)iabs(unc2) & 0xffff), 16);
                    int d = dc - pred;
                    if (d < -1024) d += 2049;
                    if (d > 1024) d -= 2049;
                    int back = d + pred;
                    if (back < -1024) back += 2049;
                    if (back > 1024) back -= 2049;
                    const int v = iabs(d), len = bitlen((uint32_t)v & 0xffff);
                    if (back != dc || len > 11) L(errdc) = 6;   // the LAST check of the block in stream order: kept apart from the earlier ones
                    const int nres = len > 1 ? len - 1 : 0;
                    e0 = (int32_t)coef_entry((uint32_t)v & ((1u << nres) - 1u), nres, len > 11 ? 11 : len, b17, a, 0);
                    if (len) sgn = 0x80 | (unc2 >= 0 ? (unc2 == 0 ? 3 : 2) : 1) | ((d >= 0) << 6);
                    bins_here = (len < 11 ? len + 1 : 11) + (len ? 1 : 0) + nres;
                  }
                } else bins_here = 22;   // count: an upper bound (the DC needs the neighbours)
            }
            L(dc_e0) = e0; L(dc_sign) = sgn; L(nzctxbin) = ctxbin; L(lbins) += bins_here;
        }

        // ---- where this block's sign bytes, threshold units and bins start: wave scans over the lanes -------------------------
        LV(int, sbase); LV(int, bbase);
        const int nsig_tile = lepwave::wave_excl_scan(nsig, sbase);
        int ttot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (MODE != kCount && kEdge) {
            for (int lt = 2; lt < 8; ++lt) {
                LV(int, tc); LV(int, to);
                LANES(l) L(tc) = S.TB[lt * 65 + l];
                ttot[lt] = lepwave::wave_excl_scan(tc, to);
                LANES(l) S.TB[lt * 65 + l] = (uint16_t)L(to);
            }
        }
        int bins_tile = 0;
        if (MODE == kGather) bins_tile = lepwave::wave_excl_scan(lbins, bbase);
        else { LANES(l) L(bbase) = 0; }
        LSYNC();

        // ---- phase R: rows in stream order ---------------------------------------------------------------------------------
        uint8_t* signs = MODE != kCount ? arena + sign_base[ci] + sign_pos[ci] : nullptr;
        uint32_t* U = MODE != kCount ? units() : nullptr;
        LV(uint32_t, bp);   // gather: next bin of this lane
        LV(uint32_t, bacc); // gather: where this lane stages its bins (see put_bin)
        LV(int, sp);        // next sign byte of this lane
        LV(int, left);
        LANES(l) {   // a block's bins: [7x7 count: 6][interior] -- the interior half's run -- [edges, DC] -- the other half's
            L(bp) = nbins + (uint32_t)L(bbase) + (kInt || !L(act) ? 0u : 6u + (uint32_t)L(ibins));
            L(bacc) = (uint32_t)l | ((L(bp) & 31u) << 8) | ((HALF == 2 ? 1u : 0u) << 16);
            L(sp) = L(sbase) + (kInt ? 0 : L(nz)); L(left) = L(nz);
        }
        // the number of non-zeros of the 7x7 interior (and the key word the sparse chains filter on)
        if (MODE == kEmit && kEdge) {
            LANES(l) if (L(act)) {
                gst(reinterpret_cast<uint32_t*>(arena + nz_base) + 2 * (ord0 + l), (uint32_t)L(nz));
                gst(reinterpret_cast<uint32_t*>(arena + key_base) + ord0 + l,
                    (uint32_t)ci | ((uint32_t)L(nzctxbin) << 1) | ((uint32_t)L(eobx) << 5) | ((uint32_t)L(eoby) << 8) | ((((uint32_t)L(dc_e0) >> 23) & 15u) << 11));
            }
        }
        if (MODE == kGather && kInt) {
            LANES(l) if (L(act)) {
                const uint32_t lo = L(rnz0), hi = L(rnz1);
                for (int i = 5; i >= 0; --i) {
                    const int q = 5 - i;
                    const uint32_t p = q < 4 ? (lo >> (8 * q)) & 255u : (hi >> (8 * (q - 4))) & 255u;
                    put_bin(L(bp), L(bacc), p | ((((uint32_t)L(nz) >> i) & 1u) << 8));
                }
            }
        }
        // gather works one row behind: the probabilities of row r are requested when its places are known and turned into bins
        // while row r + 1 is worked out (a load on the critical path costs more than everything else in the row)
        LV(uint32_t, p_e); LV(uint32_t, p_te); LV(uint32_t, p_at); LV(uint32_t, p_tat); LV(int, p_cf); LV(uint32_t, p_w0); LV(uint32_t, p_w1); LV(uint32_t, p_tw); LV(uint32_t, p_sg);
        LV(uint32_t, enw); LV(uint32_t, at_next);
        constexpr int kRow0 = kInt ? 0 : 49, kRowEnd = kEdge ? 63 : 49;
        LANES(l) { L(p_e) = 0; L(enw) = 0; L(at_next) = MODE == kGather ? gld(AT + ((size_t)tile_no * kAtRows + kRow0) * 64 + l) : 0u; }
        for (int row = kRow0; row < kRowEnd; ++row) {
            const bool edge = row >= 49;
            const int eg = row >= 56 ? 1 : 0, j = edge ? row - 49 - eg * 7 : 0;
            const bool horizontal = eg == 0;
            const bool edge_start = row == 49 || row == 56;   // an edge starts with its non-zero count (three bins)
            if (edge_start) {
                LANES(l) {
                    L(left) = horizontal ? L(neh) : L(nev);
                    if (MODE != kCount && L(act)) {
                        uint32_t* rec = reinterpret_cast<uint32_t*>(arena + en_base) + 2 * (ord0 + l) + eg;
                        if (MODE == kEmit) gst(rec, (uint32_t)((L(nz) + 3) / 7) | ((uint32_t)L(left) << 3));
                        else L(enw) = eg ? L(ren1) : L(ren0);
                    }
                }
            }
            // what the lane's coefficient of this row contributes
            LV(uint32_t, ee); LV(uint32_t, te); LV(int, kk); LV(int, nn); LV(int, cfv); LV(int, slot); LV(int, coded);
            const int a_here = row < 49 ? row : (horizontal ? 50 : 57) + j;
            const int coord = !edge ? 0 : (horizontal ? j + 1 : (j + 1) * 8);
            LANES(l) {
                uint32_t e = 0, t_e = 0; int k = 0, n = 0, sl = 0, cf = 0;
                if (L(act) && L(left) > 0) {
                    cf = tile_get(S.cur, a_here, l);
                    const int v = iabs(cf), len = bitlen((uint32_t)v), lc = len > 11 ? 11 : len;
                    int bsr = 0, nres = lc > 1 ? lc - 1 : 0;
                    if (!edge) {
                        if (MODE == kEmit) {
                            const bool has_left = x0 + l > 0;
                            int prior = 0;
                            if (has_left && has_above) prior = (uint16_t)((iabs(tile_get(S.cur, row, (l + 64) % 65)) + iabs(tile_get(S.abv, row, l))) * 13 + 6 * iabs(tile_get(S.abv, row, (l + 64) % 65))) >> 5;
                            else if (has_left) prior = (int16_t)iabs(tile_get(S.cur, row, (l + 64) % 65));
                            else if (has_above) prior = (int16_t)iabs(tile_get(S.abv, row, l));
                            bsr = bitlen((uint32_t)imin(iabs(prior), 1023));
                            if (len > 11 && !L(err)) L(err) = 6;
                        }
                        k = nzbin5(L(left));
                    } else {
                        const int thr = S.thr[coord];
                        int pcls = 0;
                        if (MODE == kEmit) {
                            const bool nbr_ok = horizontal ? has_above : (x0 + l > 0);
                            int32_t prior = 0;
                            if (nbr_ok) {
                                const uint32_t* NB = horizontal ? S.abv : S.cur;
                                const int ncol = horizontal ? l : (l + 64) % 65;
                                const int32_t* icos = horizontal ? S.icos_x + coord * 8 : S.icos_y + coord;
                                const int step = horizontal ? 8 : 1;
                                if (icos[0] == 0) { if (!L(err)) L(err) = 43; }
                                else {
                                    uint32_t acc = (uint32_t)(int32_t)tile_get(NB, kR2A[coord], ncol) * (uint32_t)icos[0];
#pragma unroll
                                    for (int i = 1; i < 8; ++i) {
                                        const int32_t xi = tile_get(S.cur, kR2A[coord + i * step], l), ai = tile_get(NB, kR2A[coord + i * step], ncol);
                                        const int32_t term = (i & 1) ? xi + ai : xi - ai;
                                        acc -= (uint32_t)icos[i] * (uint32_t)term;
                                    }
                                    prior = (int32_t)acc / icos[0];
                                }
                            }
                            const uint32_t ap = prior < 0 ? 0u - (uint32_t)prior : (uint32_t)prior;
                            bsr = bitlen(ap > 1023 ? 1023 : ap);
                            const int16_t p16 = (int16_t)prior;
                            sl = (p16 == 0 ? 0 : (p16 > 0 ? 1 : 2)) * 12 + bsr;
                            pcls = imin((int)((ap & 0xffff) >> thr), 255);
                            if (len > 11 && !L(err)) L(err) = 6;
                        }
                        if (lc > 1 && lc - 2 >= thr) {
                            const int tn = lc - 1 - thr;
                            t_e = thresh_entry(((uint32_t)v >> thr) & ((1u << tn) - 1u), tn, pcls, imin(lc - thr, 7), 0);
                            nres = thr;
                        }
                        k = L(left);
                    }
                    e = coef_entry((uint32_t)v & ((1u << nres) - 1u), nres, lc, bsr, k, 0);
                    n = coef_units(lc, nres);
                    if (v) --L(left);
                }
                L(ee) = e; L(te) = t_e; L(kk) = k; L(nn) = n; L(cfv) = cf; L(slot) = sl; L(coded) = e != 0;
            }
            if (MODE == kCount) {   // only how many: no order needed
                LANES(l) if (L(coded)) lds_add(&S.cursor[stream_id(ci, row, L(kk))], (uint32_t)L(nn));
                continue;
            }
            LV(uint32_t, at);
            if (MODE == kGather) {   // the places emit gave out: the row's were requested a row ago, the next row's are requested now
                const uint64_t rem = lepwave::wave_ballot(coded);
                const bool skip = !rem && !edge;
                const int next_row = skip ? 49 : row + 1;
                LANES(l) { L(at) = L(at_next); if (next_row < kRowEnd) L(at_next) = gld(AT + ((size_t)tile_no * kAtRows + next_row) * 64 + l); }
                if (skip) { row = 48; continue; }
            } else {
            // rank the lanes of every (row, class) in block order: the entry's place in its stream
            uint64_t rem = lepwave::wave_ballot(coded);
            if (!rem && !edge) { row = 48; continue; }   // no block of the tile has a non-zero left: the interior is done
            while (rem) {
                const int k0 = (int)lepwave::wave_read((const uint32_t*)kk, __builtin_ctzll(rem));
                const int sid = stream_id(ci, row, k0);
                const uint32_t b0 = S.cursor[sid];
                LV(int, g); LV(int, g2);
                LANES(l) { L(g) = L(coded) && L(kk) == k0; L(g2) = L(g) && L(nn) >= 2; }
                const uint64_t m1 = lepwave::wave_ballot(g), m2 = lepwave::wave_ballot(g2);
                int total = lepwave::popc64(m1) + lepwave::popc64(m2);
                LANES(l) if (L(g)) L(at) = b0 + (uint32_t)(lane_prefix(m1, l) + lane_prefix(m2, l));
                for (int q = 3; m2 && q <= 6; ++q) {   // entries of three and more units: large coefficients
                    LANES(l) L(g2) = L(g) && L(nn) >= q;
                    const uint64_t mq = lepwave::wave_ballot(g2);
                    if (!mq) break;
                    total += lepwave::popc64(mq);
                    LANES(l) if (L(g)) L(at) += (uint32_t)lane_prefix(mq, l);
                }
                LSYNC();
                LANES(l) if (l == 0) S.cursor[sid] = b0 + (uint32_t)total;
                LSYNC();
                rem &= ~m1;
            }
            LANES(l) if (L(coded)) gst(AT + ((size_t)tile_no * kAtRows + row) * 64 + l, L(at));   // for gather
            }
            // emit: write the units; gather: request this row's probabilities, turn the PREVIOUS row's into bins
            LANES(l) {
                uint32_t c_e = 0, c_te = 0, c_at = 0, c_tat = 0, c_w0 = 0, c_w1 = 0, c_tw = 0, c_sg = 0;
                if (L(coded)) {
                    const uint32_t e = L(ee), t_e = L(te);
                    const int len = (int)(e >> 14) & 15, n = L(nn);
                    uint32_t tat = 0; int tn = 0;
                    if (t_e) {
                        const int lt = (int)(t_e >> 23) & 15, tsid = stream_id(ci, 63, lt);
                        tn = (int)(t_e >> 10) & 15;
                        tat = S.tcur[ci * 8 + lt] + S.TB[lt * 65 + l];
                        S.TB[lt * 65 + l] = (uint16_t)(S.TB[lt * 65 + l] + ((tn + 3) >> 2));
                    }
                    if (MODE == kEmit) {
                        for (int u = 0; u < n; ++u) gst(U + L(at) + u, e | ((uint32_t)u << 27));
                        for (int u = 0; u < (tn + 3) >> 2; ++u) gst(U + tat + u, t_e | ((uint32_t)u << 27));
                        if (len) gst(signs + L(sp)++, (uint8_t)(0x80u | (uint32_t)L(slot) | ((uint32_t)(L(cfv) >= 0) << 6)));
                    } else {
                        c_e = e; c_te = t_e; c_at = L(at); c_tat = tat;
                        c_w0 = gld(U + c_at);
                        if (n > 1) c_w1 = gld(U + c_at + 1);
                        if (tn) c_tw = gld(U + tat);
                        if (len) c_sg = gld(signs + L(sp)++);
                    }
                }
                if (MODE == kGather) {
                    if (L(p_e)) gather_coef(L(bp), L(bacc), U, L(p_e), L(p_te), L(p_at), L(p_tat), L(p_cf), L(p_w0), L(p_w1), L(p_tw), L(p_sg));
                    if (edge_start && L(act)) {
                        const uint32_t pr = L(enw), ne = (uint32_t)(horizontal ? L(neh) : L(nev));
                        for (int i = 2; i >= 0; --i) put_bin(L(bp), L(bacc), ((pr >> (8 * (2 - i))) & 255u) | (((ne >> i) & 1u) << 8));
                    }
                    L(p_e) = c_e; L(p_te) = c_te; L(p_at) = c_at; L(p_tat) = c_tat; L(p_cf) = L(cfv); L(p_w0) = c_w0; L(p_w1) = c_w1; L(p_tw) = c_tw; L(p_sg) = c_sg;
                }
            }
        }
        if (MODE == kGather) {
            LANES(l) {
                if (L(p_e)) gather_coef(L(bp), L(bacc), U, L(p_e), L(p_te), L(p_at), L(p_tat), L(p_cf), L(p_w0), L(p_w1), L(p_tw), L(p_sg));
                if (!kEdge) flush_bin(L(bp), L(bacc));
            }
        }
        LSYNC();
        if (MODE == kEmit) {   // (inside a block the DC check comes last)
            LANES(l) { if (!L(err)) L(err) = L(errdc); S.errx[(HALF == 2 ? 64 : 0) + l] = (uint16_t)L(err); }
        }
        // DC
        if (MODE != kCount && kEdge) {
            LANES(l) if (L(act)) {
                uint32_t* rec = MODE == kEmit ? reinterpret_cast<uint32_t*>(arena + dc_base) + 6 * (ord0 + l)
                                              : reinterpret_cast<uint32_t*>(arena + dcs_base) + 6 * (size_t)L(rdcp);
                const uint32_t e = (uint32_t)L(dc_e0);
                const int len = (int)(e >> 14) & 15, nres = (int)(e >> 10) & 15, nexp = len < 11 ? len + 1 : 11, m = nexp + nres;
                if (MODE == kEmit) {
                    gst(rec, e); gst(signs + L(sp)++, (uint8_t)L(dc_sign));
                    gst(AT + ((size_t)tile_no * kAtRows + 63) * 64 + l, e | ((((uint32_t)L(dc_sign) >> 6) & 1u) << 31));
                }
                else {
                    uint32_t w = L(rdc0);
                    for (int q = 0; q < nexp; ++q) {
                        if (q == 4) w = L(rdc1); else if (q == 8) w = L(rdc2);
                        put_bin(L(bp), L(bacc), ((w >> (8 * (q & 3))) & 255u) | ((uint32_t)(len != q) << 8));
                    }
                    const uint32_t sb = gld(signs + L(sp)++);
                    if (len) put_bin(L(bp), L(bacc), sb | ((((uint32_t)L(dc_sign) >> 6) & 1u) << 8));
                    for (int q = nexp; q < m; ++q) {
                        if (!(q & 3) || q == nexp) w = (q >> 2) == 0 ? L(rdc0) : ((q >> 2) == 1 ? L(rdc1) : ((q >> 2) == 2 ? L(rdc2) : gld(rec + (q >> 2))));
                        put_bin(L(bp), L(bacc), ((w >> (8 * (q & 3))) & 255u) | (((e >> (nres - 1 - (q - nexp))) & 1u) << 8));
                    }
                    flush_bin(L(bp), L(bacc));
                }
            }
            LSYNC();
            LANES(l) if (l == 0) for (int lt = 2; lt < 8; ++lt) S.tcur[ci * 8 + lt] += (uint32_t)ttot[lt];
            LSYNC();
        }
        if (MODE == kCount || (MODE == kEmit && kEdge)) bins_tile = lepwave::wave_sum(lbins);   // (emit: exact -- the DC's bins are known; for the checkpoints)
        return TileTotals{nsig_tile, bins_tile};
    }

    // gather: append one bin (probability | bit << 8) of this lane.  A lane's bins are consecutive in the segment's list; stored as
    // they come -- four bytes at a time -- a 64-byte sector took sixteen stores spread over microseconds, and L2 wrote it back half
    // filled again and again (35 write requests per block where 3 sectors are filled: the pass was bound by them).  So a lane
    // stages the sector it is filling in LDS (32 bins, Walk5Shared::stg) and writes it when it is full -- four 16-byte stores back
    // to back -- or, where its run starts or ends inside a sector that a neighbour shares, the part that is its own.
    // `acc`: lane | first own bin of the current sector << 8 | which wavefront's staging block << 16.
    WDEV void put_bin(uint32_t& pos, uint32_t& acc, uint32_t v) {
        Walk5Shared& S = LEP5_WSH(this);
        const uint32_t k = pos & 31u, l = acc & 63u;
        reinterpret_cast<uint16_t*>(S.stg[(acc >> 16) & 1u])[(((k >> 1) * 64u + l) << 1) + (k & 1u)] = (uint16_t)v;
        ++pos;
        if ((pos & 31u) == 0u) { write_staged(pos - 32u, (acc >> 8) & 63u, 32u, acc); acc &= ~0x3f00u; }
    }
    // bins [a, b) of the sector that starts at bin `first` leave the staging block
    WDEV void write_staged(uint32_t first, uint32_t a, uint32_t b, uint32_t acc) {
        Walk5Shared& S = LEP5_WSH(this);
        const uint32_t* src = S.stg[(acc >> 16) & 1u] + (acc & 63u);
        uint32_t* dst = reinterpret_cast<uint32_t*>(bins + first);
        if (a == 0u && b == 32u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) gst4(dst + 4 * q, src[(4 * q) * 64], src[(4 * q + 1) * 64], src[(4 * q + 2) * 64], src[(4 * q + 3) * 64]);
            return;
        }
        if (a & 1u) gst(bins + first + a, (uint16_t)(src[(a >> 1) * 64] >> 16));
        for (uint32_t d = (a + 1u) >> 1; d < (b >> 1); ++d) gst(dst + d, src[d * 64]);
        if ((b & 1u) && b > a) gst(bins + first + b - 1u, (uint16_t)src[(b >> 1) * 64]);
    }
    // the bins of one coefficient in stream order: exponent, sign, threshold bits, the residual bits below the threshold
    WDEV void gather_coef(uint32_t& pos, uint32_t& acc, const uint32_t* U, uint32_t e, uint32_t t_e, uint32_t at, uint32_t tat, int cf, uint32_t w0,
                          uint32_t w1, uint32_t tw, uint32_t sg) {
        const int len = (int)(e >> 14) & 15, nres = (int)(e >> 10) & 15, nexp = len < 11 ? len + 1 : 11, m = nexp + nres;
        const int tn = t_e ? (int)(t_e >> 10) & 15 : 0;
        uint32_t w = w0;
        for (int q = 0; q < nexp; ++q) {
            if (q == 4) w = w1; else if (q == 8) w = gld(U + at + 2);
            put_bin(pos, acc, ((w >> (8 * (q & 3))) & 255u) | ((uint32_t)(len != q) << 8));
        }
        if (len) put_bin(pos, acc, sg | ((uint32_t)(cf >= 0) << 8));
        for (int t = 0; t < tn; ++t) {
            if (t && !(t & 3)) tw = gld(U + tat + (t >> 2));
            put_bin(pos, acc, ((tw >> (8 * (t & 3))) & 255u) | (((t_e >> (tn - 1 - t)) & 1u) << 8));
        }
        for (int q = nexp; q < m; ++q) {
            if (q == nexp || !(q & 3)) w = (q >> 2) == 0 ? w0 : ((q >> 2) == 1 ? w1 : gld(U + at + (q >> 2)));
            put_bin(pos, acc, ((w >> (8 * (q & 3))) & 255u) | (((e >> (nres - 1 - (q - nexp))) & 1u) << 8));
        }
    }
    // the lane's run ends here: what it has staged of the sector it stands in
    WDEV void flush_bin(uint32_t pos, uint32_t acc) {
        const uint32_t k = pos & 31u, a = (acc >> 8) & 63u;
        if (k > a) write_staged(pos - k, a, k, acc);
    }

    // whole segment (lepton_codec.hh:41-100 row schedule, vp8_encoder.cc:239-445); ns: the segment's two-row NSum rings (zeroed)
    // emit: the state at the first tile of every part (the wavefront that holds the exact bin count writes it)
    WDEV void checkpoints(uint32_t ntiles, int nparts) {
        if (MODE != kEmit || (LEP_ON_GPU && wave != NW - 1)) return;
        Ckpt5* ck = reinterpret_cast<Ckpt5*>(arena + plan->ckpt_base);
        for (int q = 1; q < nparts; ++q) {
            if (part_first_tile(ntiles, q, nparts) != tile_no) continue;
            LANES(l) {
                if (l < 16) gst(&ck[q].tcur[l], LEP5_WSH(this).tcur[l]);
                if (l == 16) { gst(&ck[q].tile_no, tile_no); gst(&ck[q].ord0, ord0); gst(&ck[q].sign_pos[0], sign_pos[0]); gst(&ck[q].sign_pos[1], sign_pos[1]); gst(&ck[q].nbins, nbins); }
            }
        }
    }
    // part / nparts: gather only (the other passes walk the whole segment; emit leaves the checkpoints for nparts parts)
    WDEV int run(const ImageDev* image, const SegDev& seg, NSum* ns, Walk5Shared* shared, const SegPlan5* pl, uint8_t* arena_base, uint16_t* bins_base,
                 int wave_no = 0, int part = 0, int nparts = 1) {
        img = image; sh = shared; plan = pl; status = 0; wave = wave_no;
        arena = (MODE != kCount) ? arena_base + pl->arena_off : nullptr;
        bins = (MODE == kGather) ? bins_base + pl->bins_off : nullptr;
        ord0 = 0; sign_pos[0] = sign_pos[1] = 0; nbins = 0; tile_no = 0;
        AT = MODE != kCount ? reinterpret_cast<uint32_t*>(arena + pl->at_base) : nullptr;
        if (MODE != kCount) { sign_base[0] = pl->sign_base[0]; sign_base[1] = pl->sign_base[1]; key_base = pl->key_base; nz_base = pl->nz_base; en_base = pl->en_base; dc_base = pl->dc_base;
            place_base = pl->place_base; nzs_base = pl->nzs_base; ens_base[0] = pl->ens_base[0]; ens_base[1] = pl->ens_base[1]; dcs_base = pl->dcs_base; plan_nblocks = pl->nblocks; }
        LANES(l) {   // emit / gather: a cursor is the absolute place of the stream's next unit
            if (MODE != kGather) for (int i = l; i < 2 * kRows * kClasses; i += 64) LEP5_WSH(this).cursor[i] = MODE == kCount ? 0u : pl->base[i];   // (every wavefront: the same values)
            if (MODE != kCount && l < 16) LEP5_WSH(this).tcur[l] = pl->base[stream_id(l >> 3, 63, l & 7)];
        }
        const uint32_t ntiles = MODE != kCount ? pl->ntiles : 0u;
        const uint32_t first = MODE == kGather ? part_first_tile(ntiles, part, nparts) : 0u;
        const uint32_t end = MODE == kGather && part + 1 < nparts ? part_first_tile(ntiles, part + 1, nparts) : 0xffffffffu;
        if (MODE == kGather && first > 0) {   // take the segment up where emit left the checkpoint
            const Ckpt5* ck = reinterpret_cast<const Ckpt5*>(arena + pl->ckpt_base) + part;
            tile_no = gld(&ck->tile_no); ord0 = gld(&ck->ord0); sign_pos[0] = gld(&ck->sign_pos[0]); sign_pos[1] = gld(&ck->sign_pos[1]); nbins = gld(&ck->nbins);
            LANES(l) if (l < 16) LEP5_WSH(this).tcur[l] = gld(&ck->tcur[l]);
        }
        LEP5_XSYNC();
        // the tiles of the segment in stream order (lepton_codec.hh:41-100; vp8_encoder.cc:83-154: a row ends where the file was cut)
        TileIter it;
        it.init(image, seg, ns);
        TileDesc cur_t{}, nxt_t{};
        LV(TileRegs, regs);
        bool have = it.next(&cur_t);
        for (uint32_t skip = 0; have && skip < first; ++skip) have = it.next(&cur_t);   // (the schedule is cheap to step through)
        if (have && first < end) fetch_tile(cur_t, regs); else have = false;
        comp = -1;
        if (MODE == kEmit) checkpoints(ntiles, nparts);   // (parts that start at tile 0)
        while (have) {
            store_tile(cur_t, regs);
            const bool more = it.next(&nxt_t) && tile_no + 1 < end;
            if (more) fetch_tile(nxt_t, regs);   // in flight while this tile is worked on
            const int rc = tile(cur_t.x0, cur_t.nb, cur_t.has_above, cur_t.nrow, cur_t.narow);
            if (rc) return rc;
            if (MODE == kEmit) { LEP5_XSYNC(); checkpoints(ntiles, nparts); }
            cur_t = nxt_t;
            have = more;
        }
        LEP5_XSYNC();   // (the cursors are complete)
        if (MODE == kEmit && wave == 0) {   // every stream is padded to whole groups of four units: the fold takes them group by group
            LANES(l) if (l < 16) LEP5_WSH(this).cursor[stream_id(l >> 3, 63, l & 7)] = LEP5_WSH(this).tcur[l];
            LSYNC();
            LANES(l) {
                for (int i = l; i < kStreams; i += 64)
                    for (uint32_t j = LEP5_WSH(this).cursor[i]; j < pl->base[i + 1]; ++j) gst(units() + j, 0u);
            }
        }
        return 0;
    }
};


// ---- plan: counts -> arena layout --------------------------------------------------------------------------------------------
// counts of one segment as the count walk leaves them: [kStreams] units per stream, then sign bytes of the two colour indices,
// block ordinals, bins (an upper bound through the DC term)
constexpr int kCountWords = kStreams + 5;   // ... and the number of tiles
// one segment's layout (arena_off / bins_off are filled in by the prefix pass over the segments)
WDEV void plan_segment(const uint32_t* counts, SegPlan5* P) {
    uint32_t at = 0;
    for (int i = 0; i < kStreams; ++i) { P->base[i] = at; P->cnt[i] = counts[i]; at += (counts[i] + 3u) & ~3u; }
    P->base[kStreams] = at;
    uint32_t bytes = at * 4;
    P->sign_cnt[0] = counts[kStreams]; P->sign_cnt[1] = counts[kStreams + 1];
    P->sign_base[0] = bytes; bytes += (P->sign_cnt[0] + 15u) & ~15u;
    P->sign_base[1] = bytes; bytes += (P->sign_cnt[1] + 15u) & ~15u;
    P->nblocks = counts[kStreams + 2];
    P->key_base = bytes; bytes += ((P->nblocks * (uint32_t)kKeyRec) + 15u) & ~15u;
    P->nz_base = bytes; bytes += P->nblocks * (uint32_t)kNzRec;
    P->en_base = bytes; bytes += P->nblocks * (uint32_t)kEnRec;
    P->dc_base = bytes; bytes += P->nblocks * (uint32_t)kDcRec;
    P->cls_base = bytes; bytes += 2u * kNumCls * 4u;
    P->place_base = bytes; bytes += 4u * P->nblocks * 4u;
    bytes = (bytes + 15u) & ~15u;
    P->nzs_base = bytes; bytes += (P->nblocks + 4u * 20u) * (uint32_t)kNzRec;
    P->ens_base[0] = bytes; bytes += (P->nblocks + 4u * 16u) * 4u;
    P->ens_base[1] = bytes; bytes += (P->nblocks + 4u * 16u) * 4u;
    P->dcs_base = bytes; bytes += (P->nblocks + 4u * 12u) * (uint32_t)kDcRec;
    bytes = (bytes + 15u) & ~15u;
    P->ckpt_base = bytes; bytes += (uint32_t)(kMaxParts * sizeof(Ckpt5));
    P->wstate_base = bytes; bytes += 64u;
    bytes = (bytes + 255u) & ~255u;
    P->ntiles = counts[kStreams + 4];
    P->at_base = bytes; bytes += P->ntiles * (uint32_t)(kAtRows * 64 * 4);
    P->arena_bytes = (bytes + 255u) & ~255u;
    P->bins_cap = (counts[kStreams + 3] + 127u) & ~127u;
    P->nbins = 0; P->status = 0; P->arena_off = 0; P->bins_off = 0;
}
// the count walk's results -> counts[kCountWords] (lane-strided copy of the cursors)
template <class W>
WDEV void export_counts(const W& w, const Walk5Shared* sh, uint32_t* counts) {
    LANES(l) {
        for (int i = l; i < kStreams; i += 64) counts[i] = sh->cursor[i];
        if (l == 0) { counts[kStreams] = w.sign_pos[0]; counts[kStreams + 1] = w.sign_pos[1]; counts[kStreams + 2] = w.ord0; counts[kStreams + 3] = w.nbins; counts[kStreams + 4] = w.tile_no; }
    }
}

}  // namespace lep5
