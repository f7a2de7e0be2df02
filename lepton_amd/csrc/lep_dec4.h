// lep_dec4.h -- "v4" decoder.  Same boundary, model layout (lep_v3.h) and owner-lane scheme as v3; what changes, and
// why (measured on MI355X, profiles/r01c_*, profiles/r01_issue_microbench.txt):
//   * v3 ran the serial part wave-uniform on the scalar unit: 5.6k SALU instructions per block, and the CU's one scalar
//     unit saturates at ~0.85 instructions/cycle.  The bool-decoder recurrence written as UNIFORM VECTOR code (every lane
//     computes the same value in VGPRs, branches taken on ballots) runs 1.85x faster at 8 waves/SIMD.  v4's serial code is
//     uniform vector code; the scalar unit only keeps loop counters and lane indices.
//   * v3 needed 10.2 prefetch/serial/update rounds per block (5.15 of them for the 7x7 interior, because a round ended
//     whenever "non-zeros left" moved to another bin).  v4 prefetches, per window position, the contexts of FOUR
//     consecutive non-zero bins (lane = position + 16 * candidate), so an interior round normally runs to the end of its
//     16-position window; and it codes both edge count trees and both edges in ONE round (lane = every reachable
//     (position, edge-non-zeros-left) pair: 28 per edge).  Typical block: NZ + 1..2 interior + 1 edge + 1 DC round.
//   * the DC residual Branches (LDS resident) are handed to lanes before the DC round and adapted by them afterwards.
// Syntax / contexts: src/vp8/decoder/decoder.cc:27-141,167-318; src/vp8/model/model.hh:463-485,852-871,1033-1122,674-832
// (the same citations as lep_core.h, whose results this kernel reproduces bit for bit).
#pragma once
#include "lep_v3.h"
#include "lep_enc3.h"   // ucond()

namespace lep4 {
using namespace lep3;

// LEP_MARK(name): phase boundary.  -DLEP_MARKS: a comment in the ISA (static instruction counts per phase, scripts/isa.sh);
// -DLEP_PROF: lane 0 charges the shader-clock time since the previous boundary to the PREVIOUS phase's slot (wall-clock
// share of each phase at the occupancy the kernel really runs at; read back with lep_gpu_debug_prof).
#if LEP_ON_GPU && defined(LEP_MARKS)
#define LEP_MARK(name) __asm__ volatile("; MARK " name)
#elif LEP_ON_GPU && defined(LEP_PROF)
#define LEP_MARK(name) prof_stamp(lep4::prof_slot(name))
#else
#define LEP_MARK(name) ((void)0)
#endif
// -DLEP_DEC4_PAD_VALU=<n> / _SALU / _LDS: n extra independent instructions of that kind in every block (experiment builds,
// scripts/build_variant.sh): the slope of the launch time over n says which issue port the kernel is short of
// (profiles/r05w_decoder_issue_port_sensitivity.txt)
#if LEP_ON_GPU && (defined(LEP_DEC4_PAD_VALU) || defined(LEP_DEC4_PAD_SALU) || defined(LEP_DEC4_PAD_LDS))
#define LEP_PAD8(x) x x x x x x x x
WDEV void pad_block(uint32_t seed, const void* lds) {
    (void)seed; (void)lds;
#ifdef LEP_DEC4_PAD_VALU
    { uint32_t o; for (int i = 0; i < LEP_DEC4_PAD_VALU / 8; ++i) __asm__ volatile(LEP_PAD8("v_add_u32 %0, %1, %1\n") : "=v"(o) : "v"(seed)); }
#endif
#ifdef LEP_DEC4_PAD_SALU
    { uint32_t o; const uint32_t u = uni(seed); for (int i = 0; i < LEP_DEC4_PAD_SALU / 8; ++i) __asm__ volatile(LEP_PAD8("s_add_u32 %0, %1, %1\n") : "=s"(o) : "s"(u) : "scc"); }
#endif
#ifdef LEP_DEC4_PAD_LDS
    { uint32_t o; const uint32_t a = (uint32_t)(uintptr_t)lds; for (int i = 0; i < LEP_DEC4_PAD_LDS / 8; ++i) __asm__ volatile(LEP_PAD8("ds_read_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(o) : "v"(a)); }
#endif
}
#else
WDEV void pad_block(uint32_t, const void*) {}
#endif
constexpr int prof_slot(const char* n) {
    // staging prologue nz_prefetch nz_serial nz_update 77_prefetch 77_serial 77_update lakhani edge_prefetch edge_serial
    // edge_update idct_dcpred dc_prefetch dc_serial dc_update publish store
    const char* names[18] = {"staging", "prologue", "nz_prefetch", "nz_serial", "nz_update", "77_prefetch", "77_serial", "77_update", "lakhani",
                             "edge_prefetch", "edge_serial", "edge_update", "idct_dcpred", "dc_prefetch", "dc_serial", "dc_update", "publish", "store"};
    for (int i = 0; i < 18; ++i) {
        int k = 0;
        while (names[i][k] && names[i][k] == n[k]) ++k;
        if (!names[i][k] && !n[k]) return i;
    }
    return 31;
}

// ---- model layout of this kernel ---------------------------------------------------------------------------------------
// Same tables and sizes as lep_v3.h, but the three big families (interior exponents, edge exponents, residuals) are stored
// [..][group 0..2]["non-zeros left" 0..9][4 words] instead of ["non-zeros left"][..][12 words]: the lanes of a round prefetch
// the SAME position under several consecutive "non-zeros left" values (4 candidates in the interior, 1..7 at an edge
// position), and in this order those 16-byte groups are neighbours -- one or two 64-byte sectors instead of one sector each.
// Measured before the change (MI355X, profiles/r02a_pmc_bound_summary.json, r02b_fetch_calibration.txt): 257 sector reads per
// block, 9 % L2 hit rate, 38 G requests/s -- the rate a pure random-gather kernel reaches on this chip.
// Word i of a row: base + (i / 4) * kGS + (i % 4).
#ifndef LEP_DEC4_ROW_MAJOR
constexpr uint32_t kGS = 10 * 4;   // words from one group of a row to the next
WDEV uint32_t ctx4_exp7(int ci, int nb, int zz, int bsr) { return lep3::kExp7 + (((((uint32_t)ci * 49 + zz) * 12 + bsr) * 3) * 10 + nb) * 4; }
WDEV uint32_t ctx4_expx(int ci, int ne, int zig15, int bsr) { return lep3::kExpX + (((((uint32_t)ci * 15 + zig15) * 12 + bsr) * 3) * 10 + ne) * 4; }
WDEV uint32_t ctx4_res(int ci, int coord, int nb) { return lep3::kRes + ((((uint32_t)ci * 64 + coord) * 3) * 10 + nb) * 4; }
#else   // -DLEP_DEC4_ROW_MAJOR: round 1's layout (contiguous 12-word rows), kept for A/B measurements
constexpr uint32_t kGS = 4;
WDEV uint32_t ctx4_exp7(int ci, int nb, int zz, int bsr) { return lep3::ctx_exp7(ci, nb, zz, bsr); }
WDEV uint32_t ctx4_expx(int ci, int ne, int zig15, int bsr) { return lep3::ctx_expx(ci, ne, zig15, bsr); }
WDEV uint32_t ctx4_res(int ci, int coord, int nb) { return lep3::ctx_res(ci, coord, nb); }
#endif
WDEV uint32_t row_word(uint32_t base, int i) { return base + (uint32_t)(i >> 2) * kGS + (uint32_t)(i & 3); }

struct Dec4Shared {
    uint32_t sign[kSignWords];    // resident Branches
    uint32_t resdc[kResDcWords];
    alignas(16) int32_t t[64];                // IDCT intermediate
    int32_t icos_x[64], icos_y[64];   // [8..63] as in ImageDev; [0..7] (the DC row / column, which no prior reads) hold DivBy multipliers:
                                      // icos_x[p] / icos_y[p - 7] for the divisor of edge position p, icos_x[7] and icos_y[7] = the DC quantiser's pair
    int32_t eprior[16];           // Lakhani priors of the 14 edge positions; [14] = bit mask of positions whose prior divides by zero
    int16_t here[64], left[64], above[64], aleft[64];   // aligned order
    alignas(16) int16_t pix[64];
    uint16_t q[64];
    uint8_t thr[64], r2a[64], a2r[64], nzbin[64], bsr[64];
    uint8_t cj[32], cn[32];       // edge combo -> position / non-zeros-left
    NSum ns_left, ns_above, ns_here;
    uint32_t inv24[512];          // exact 24-bit reciprocals for the Branch probability (inv24_of)
};
// eight wavefronts per SIMD = 32 one-wavefront workgroups per CU share its 160 KB: 5104 bytes fit, 128 more and only 28 workgroups do
// (measured when the division constants first had fields of their own: 1121 -> 1405 ms per launch)
static_assert(sizeof(Dec4Shared) <= 160 * 1024 / 32, "Dec4Shared no longer fits 32 times into a CU's LDS");

// kNzBin for 0 <= left <= 49 without a memory access (scalar arithmetic on the GPU)
WDEV int nzbin_of(int left) {
    return left < 16 ? (int)((0x7776666555443210ull >> (4 * left)) & 15) : (left < 21 ? 7 : (left < 32 ? 8 : 9));
}
// largest "non-zeros left" that still maps to bin b (inverse of kNzBin): 0,1,2,3,5,8,12,20,31,49
WDEV int nzhi_of(int b) { return (int)((0xc5f50c2050c2040ull >> (6 * b)) & 63); }
#ifndef LEP_DEC4_CANDS
#define LEP_DEC4_CANDS 4
#endif
// first combo of edge position j (combos of a position = its reachable "non-zeros left" values 1 .. 7-j): 0,7,13,18,22,25,27
WDEV int combo_base(int j) { return (int)((0x1b65648d1c0ull >> (6 * j)) & 63); }

// Reciprocal m[d] with ((f << 8) * m[d]) >> 24 == (f << 8) / d EXACTLY for every reachable count pair (1 <= f, t <= 255,
// d = f + t), and the product's bits 24..31 (all a probability needs) inside the low 32 bits a v_mul_u32_u24 returns:
// m = ceil(2^24 / d), except d = 337 and d = 469 where only ceil - 1 is exact (exhaustive check: tests/emu, lep_gpu_selftest).
WDEV uint32_t inv24_of(uint32_t d) {
    if (d < 2) return 0;
    return (0x1000000u + d - 1) / d - ((d == 337 || d == 469) ? 1u : 0u);
}
WDEV uint32_t mul24(uint32_t a, uint32_t b) {
#if LEP_ON_GPU
    return __umul24(a, b);
#else
    return (uint32_t)(((uint64_t)(a & 0xffffff) * (b & 0xffffff)) & 0xffffffffu);
#endif
}
// the same where only bits below 24 of the product are wanted: the optimiser then drops the 24-bit masks and emits a full
// v_mul_lo_u32 (quarter rate on CDNA) -- keep the instruction opaque
WDEV uint32_t mul24_low(uint32_t a, uint32_t b) {
#if LEP_ON_GPU
    uint32_t r;
    __asm__("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return mul24(a, b);
#endif
}
// Truncating division of any int32 by a divisor 2 <= d < 2^31 that stays the same for a row of blocks, as one multiplication:
// with s = ceil(log2 d) and m = ceil(2^(31+s) / d) (m < 2^32; m * d = 2^(31+s) + e, 0 <= e < d <= 2^s),
// (u * m) >> (31 + s) = floor(u / d) for every u <= 2^31 -- the error term u * e / (d * 2^(31+s)) stays below 1 / d.
// The compiler's own expansion of `/` is ~26 vector instructions, four of them quarter-rate multiplies; the block has two.
// (exhaustive over the divisors the tables can hold, against `/`: tests/emu emu_check_div_by, lep_gpu_selftest)
struct DivBy {
    uint32_t mul, shift;          // shift = s - 1 (the other 32 come with the high half of the product)
    WDEV static uint32_t shift_of(uint32_t d) { return 31u - (uint32_t)__builtin_clz(d - 1); }   // d >= 2
    WDEV static DivBy of(uint32_t d) {
        DivBy r;
        uint32_t s = 1;
        while (s < 31 && (1u << s) < d) ++s;
        uint32_t q = 0, rem = 1u << (s - 1);                 // 2^(31+s) = rem * 2^32, rem < d: 32 steps of long division
        for (int i = 0; i < 32; ++i) {
            rem <<= 1; q <<= 1;
            if (rem >= d) { rem -= d; q |= 1u; }
        }
        r.mul = q + (rem ? 1u : 0u);
        r.shift = s - 1;
        return r;
    }
};
WDEV uint32_t mulhi_u32(uint32_t a, uint32_t b) {
#if LEP_ON_GPU
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
WDEV int32_t div_by(int32_t n, uint32_t mul, uint32_t shift) {
    const int32_t sign = n >> 31;
    const uint32_t u = ((uint32_t)n ^ (uint32_t)sign) - (uint32_t)sign;      // |n|, 2^31 for the most negative one
    const uint32_t q = mulhi_u32(u, mul) >> shift;
    return (int32_t)((q ^ (uint32_t)sign) - (uint32_t)sign);
}
// Branch::record_obs_and_update (branch.hh:82-100) on the packed word, per lane: straight-line common case (table
// reciprocal), one divergent branch for the count-overflow case (once per ~250 observations of a Branch)
WDEV uint32_t bupd_t(uint32_t w, uint32_t obs, const uint32_t* inv24) {
    const uint32_t f = (w & 255) + (obs ^ 1), t = ((w >> 8) & 255) + obs;
    uint32_t nw = f | (t << 8) | ((mul24(f << 8, inv24[f + t]) >> 24) << 16);
    if ((f | t) > 255) {   // the incremented count was 255
        const uint32_t f0 = w & 255, t0 = (w >> 8) & 255;
        if ((obs ? f0 : t0) == 1) nw = (w & 0xffff) | ((obs ? 0u : 255u) << 16);
        else {
            const uint32_t f2 = obs ? (1 + f0) >> 1 : 129u, t2 = obs ? 129u : (1 + t0) >> 1;
            nw = f2 | (t2 << 8) | ((mul24(f2 << 8, inv24[f2 + t2]) >> 24) << 16);
        }
    }
    return nw;
}
// the same for a uniform-vector word and observation (0 / 1): the count is bumped by adding 1 or 256 to the packed word,
// a wrapped count shows up as a zero count byte and takes the exact rule on a ballot (once per ~250 observations)
WDEV uint32_t bupd_u(uint32_t w, uint32_t obs, const uint32_t* inv24) {
    const uint32_t w2 = w + 1u + obs * 255u;
    const uint32_t f = w2 & 255u, t = (w2 >> 8) & 255u;
    if (ucond((f < t ? f : t) == 0)) {   // the incremented count was 255
        const uint32_t f0 = w & 255, t0 = (w >> 8) & 255;
        if (ucond((obs ? f0 : t0) == 1)) return (w & 0xffff) | ((obs ? 0u : 255u) << 16);
        const uint32_t f2 = obs ? (1 + f0) >> 1 : 129u, t2 = obs ? 129u : (1 + t0) >> 1;
        return f2 | (t2 << 8) | ((mul24(f2 << 8, inv24[f2 + t2]) >> 24) << 16);
    }
    return (w2 & 0xffffu) | (mul24_low(f, inv24[f + t]) & 0xff0000u);
}

// ---- bool decoder (boolreader.hh:184-258, 376-416; boolreader.cc:25-34) as uniform vector code ----------------------
// 64-bit window (vhi:vlo) refilled with one ALIGNED dword at a time, the next dword requested one refill ahead.  Bytes
// outside [0, len) of the stream are never loaded and read as zero bits (the reference's behaviour past the end).
// two dwords as one 64-bit value through a vector bit cast: the compiler then sees a register pair, where `(hi << 32) | lo` on
// scalar registers becomes two moves and an s_or_b64 per bin (LEP_DEC4_PAIR_OR keeps the arithmetic form for A/B builds)
WDEV uint64_t pair64(uint32_t hi, uint32_t lo) {
#if LEP_ON_GPU && !defined(LEP_DEC4_PAIR_OR)
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 p = {lo, hi};
    return __builtin_bit_cast(uint64_t, p);
#else
    return ((uint64_t)hi << 32) | lo;
#endif
}
struct BoolDec4 {
    uint32_t vhi, vlo;      // top-aligned window
    int count;              // valid bits - 8
    uint32_t range;
    const uint32_t* words;  // aligned dword that holds stream byte 0
    uint32_t first, end;    // stream bytes live at byte offsets [first, end) from `words`
    uint32_t wi;            // index of the dword `raw` holds
    uint32_t raw;           // words[wi] as loaded (0 if it holds no stream byte)

    WDEV uint32_t fetch(uint32_t k) const { return k * 4 < end ? words[k] : 0u; }
    WDEV void refill() {
        const uint32_t lo = wi * 4;
        uint32_t w = __builtin_bswap32(raw);
        int nbits = 32;
        if (lo + 4 > end) w = lo < end ? (w & ~(0xffffffffu >> ((end - lo) * 8))) : 0u;   // bytes past the end -> 0
        if (lo < first) { w <<= (first - lo) * 8; nbits -= (int)(first - lo) * 8; }              // bytes before the start
        const uint64_t add = ((uint64_t)w << 32) >> (count + 8);
        vhi |= (uint32_t)(add >> 32); vlo |= (uint32_t)add;
        count += nbits;
        ++wi;
        raw = fetch(wi);
    }
    WDEV void init_stream(const uint8_t* p, uint32_t n) {
        const uint32_t mis = (uint32_t)((uintptr_t)p & 3);
        words = reinterpret_cast<const uint32_t*>(p - mis);
        first = mis; end = mis + n;
        vhi = vec(0); vlo = vec(0); count = (int)vec((uint32_t)-8); range = vec(255); wi = 0;
        raw = fetch(0);
        refill();
        get(128);
    }
    // returns the decoded bit (0 / 1) as a uniform vector value
    WDEV uint32_t get(uint32_t prob) {
#if LEP_ON_GPU
        const uint32_t split = 1 + (__umul24(range - 1, prob) >> 8);
#else
        const uint32_t split = 1 + (((range - 1) * prob) >> 8);
#endif
        if (ucond(count < 0)) refill();
        const uint32_t big = split << 24;
        const uint32_t bit = vhi >= big ? 1u : 0u;
        const uint32_t d = vhi - big;
        vhi = d < vhi ? d : vhi;                 // subtract only when it does not wrap, i.e. when bit = 1 (d == vhi iff big == 0: never)
        range = bit ? range - split : split;     // (one v_sub_co + two selects on its borrow measured 0.4 % slower: profiles/r02w_*)
#ifdef LEP_TRACE_GET
        LEP_TRACE_GET(prob, (int)bit);
#endif
        const int shift = __builtin_clz(range) - 24;
        range <<= shift;
        const uint64_t v = pair64(vhi, vlo) << shift;
        vhi = (uint32_t)(v >> 32); vlo = (uint32_t)v;
        count -= shift;
        return bit;
    }
};

// The same reader with its state on the SCALAR unit (wave-uniform values the compiler keeps in SGPRs).  Why both exist:
// at 8 wavefronts per SIMD the coder kernels are bound by VALU issue (one wave-instruction per ~1.2 cycles per CU,
// scripts/proto/inst_rates.hip, profiles/r02l_inst_rates.txt) while the scalar unit, which issues beside it from other
// wavefronts at about the same rate, sat half idle (3400 VALU against 1500 SALU instructions per block).  A serial round
// coded in this form takes its ~16 instructions per bin off the vector ALUs; the rounds are split between the two forms
// (LEP_DEC4_SCALAR) so that both units are busy.  The state crosses over with four v_readfirstlane / v_mov per switch;
// the stream words keep arriving through the vector cache (BoolDec4::raw, one request ahead).
struct BoolDec4S {
    uint32_t vhi, vlo;
    int count;
    uint32_t range;         // the range in the TOP byte (range << 24): split << 24 is what the window's high word is compared with, so the
                            // product is taken where it is needed -- ((range - 1) << 24) x (prob << 24) >> 32 = (range - 1) * prob << 16, whose top byte
                            // is ((range - 1) * prob) >> 8 -- and the normalisation shift is the count of leading zeros as it comes (round 6:
                            // two scalar instructions fewer per bin than range / split / split << 24 / clz - 24)
#ifdef LEP_DEC4_RANGE_LOW   // (the form until round 6, kept for the A/B: profiles/r6z2_decoder_scalar_bin_ab.txt)
    WDEV void load(const BoolDec4& b) { vhi = uni(b.vhi); vlo = uni(b.vlo); count = (int)uni((uint32_t)b.count); range = uni(b.range); }
    WDEV void store(BoolDec4& b) const { b.vhi = vec(vhi); b.vlo = vec(vlo); b.count = (int)vec((uint32_t)count); b.range = vec(range); }
#else
    WDEV void load(const BoolDec4& b) { vhi = uni(b.vhi); vlo = uni(b.vlo); count = (int)uni((uint32_t)b.count); range = uni(b.range) << 24; }
    WDEV void store(BoolDec4& b) const { b.vhi = vec(vhi); b.vlo = vec(vlo); b.count = (int)vec((uint32_t)count); b.range = vec(range >> 24); }
#endif
    WDEV void refill(BoolDec4& b) {
        const uint32_t lo = b.wi * 4;
        uint32_t w = __builtin_bswap32(uni(b.raw));
        int nbits = 32;
        if (lo + 4 > b.end) w = lo < b.end ? (w & ~(0xffffffffu >> ((b.end - lo) * 8))) : 0u;
        if (lo < b.first) { w <<= (b.first - lo) * 8; nbits -= (int)(b.first - lo) * 8; }
        const uint64_t add = ((uint64_t)w << 32) >> (count + 8);
        vhi |= (uint32_t)(add >> 32); vlo |= (uint32_t)add;
        count += nbits;
        ++b.wi;
        b.raw = b.fetch(b.wi);
    }
    // prob in the low byte of `pk` (the bytes above it are the caller's next probabilities and fall off the shift)
    WDEV uint32_t get(BoolDec4& b, uint32_t pk) {
        if (count < 0) refill(b);
#ifdef LEP_DEC4_RANGE_LOW
        const uint32_t split = 1 + (((range - 1) * (pk & 255u)) >> 8);
        const uint32_t big = split << 24;
        uint32_t was = vhi;
        const uint32_t bit = was >= big ? 1u : 0u;
        vhi = bit ? was - big : was;
        range = bit ? range - split : split;
        const int shift = __builtin_clz(range) - 24;
#else
        const uint32_t big = ((uint32_t)(((uint64_t)(range - 0x01000000u) * (uint64_t)(pk << 24)) >> 32) & 0xff000000u) + 0x01000000u;   // split << 24
        uint32_t was = vhi;
        const uint32_t bit = was >= big ? 1u : 0u;
        vhi = bit ? was - big : was;
        range = bit ? range - big : big;
        const int shift = __builtin_clz(range);
#endif
#ifdef LEP_TRACE_GET
        LEP_TRACE_GET(pk & 255u, (int)bit);
#endif
        range <<= shift;
        const uint64_t v = pair64(vhi, vlo) << shift;
        vhi = (uint32_t)(v >> 32); vlo = (uint32_t)v;
        count -= shift;
#if LEP_ON_GPU && !defined(LEP_DEC4_PAIR_OR)
        // the decision once more for the caller's branch, from the two registers that still hold it: one s_cmp in front of the
        // branch instead of a 64-bit mask made of SCC and carried across the normalisation (its shifts and subtractions write SCC)
        __asm__ volatile("" : "+s"(was));
        return was >= big ? 1u : 0u;
#else
        return bit;
#endif
    }
};

// Issue priority by kind of phase (s_setprio): a wavefront in a SERIAL round lowers its priority -- to 0 where the round runs on the scalar
// unit, to 1 where it is uniform vector code -- and goes back to 2 for everything lane-parallel (prefetch, owners' update, priors, IDCT,
// staging, store).  Why (MI355X, 1024 x 4K, profiles/r09b..e_decoder_priority_*): the eight wavefronts of a SIMD compete for one vector
// and one scalar issue slot per turn.  A wavefront in a serial round issues one DEPENDENT instruction every second to fourth turn
// whatever its priority; one in a lane-parallel phase has independent instructions for every turn and ends in the memory requests its
// next round waits for.  With equal priorities the arbiter served them oldest first; letting the lane-parallel work go first took the
// launch from 1117 to 980 ms (-12 %).  Swept as a table of levels per kind (prefetch / scalar serial / vector serial / update / other
// / staging + store), 26 combinations over three visits: scalar serial lowest is worth -8 %, vector serial one level above it another
// -4 %, the order among the lane-parallel kinds nothing (978..984 ms); raising the serial rounds instead loses (1116 -> 1116..1026 the
// other way round).  -DLEP_DEC4_PRIO=0 builds the kernel without it.
#ifndef LEP_DEC4_PRIO
#define LEP_DEC4_PRIO 1
#endif
#if LEP_ON_GPU && LEP_DEC4_PRIO
#define LEP_PRIO_SERIAL(on_scalar_unit) __builtin_amdgcn_s_setprio((on_scalar_unit) ? 0 : 1)
#define LEP_PRIO_PARALLEL() __builtin_amdgcn_s_setprio(2)
#else
#define LEP_PRIO_SERIAL(on_scalar_unit) ((void)0)
#define LEP_PRIO_PARALLEL() ((void)0)
#endif
// which serial rounds run on the scalar unit: 1 = non-zero count tree, 2 = 7x7 interior, 4 = edges, 8 = DC
#ifndef LEP_DEC4_SCALAR
#define LEP_DEC4_SCALAR 2   // measured (1024 x 4K, MI355X, profiles/r02m_*): 0: 1232 ms, 2: 1204, 3: 1212, 11: 1226, 7: 1400, 15: 1440
#endif

// SCMASK: which serial rounds run on the scalar unit (LEP_DEC4_SCALAR above for the throughput kernel; the launches of a few segments,
// where a wavefront has its SIMD to itself and the dependent-instruction latency is all that counts, take their own: lep_gpu.hip)
template <int SCMASK>
struct Dec4WaveT {
    const ImageDev* img;
    uint32_t* model;
    Dec4Shared* sh;
    int comp, ci;
    BoolDec4 bc;
    uint32_t nbins;   // bins decoded, accounted per coefficient: a coefficient of bit length len costs 2*len+1 bins (22 at len 11)
    // the count is a test aid (the emulation compares it with the oracle's); on the GPU nobody reads it, and kept in the serial
    // loops it costs ~150 scalar instructions per block on a kernel that is bound by instruction issue
#if LEP_ON_GPU && !defined(LEP_COUNT_BINS)
#define LEP_BINS(expr) ((void)0)
#else
#define LEP_BINS(expr) (expr)
#endif
#if LEP_ON_GPU && defined(LEP_PROF)
    uint64_t prof_last;
    unsigned long long* prof_out;   // 32 accumulators of this wave in global memory, written once at the end
    int prof_cur;
    uint32_t prof_acc[20];          // registers: every index below is a compile-time constant
    template <int I> __device__ __forceinline__ void prof_add(uint32_t dt) { if (prof_cur == I) prof_acc[I] += dt; }
    __device__ __forceinline__ void prof_stamp(int slot) {
        const uint64_t t = __builtin_readcyclecounter();
        const uint32_t dt = (uint32_t)(t - prof_last);
        prof_last = t;
        prof_add<0>(dt); prof_add<1>(dt); prof_add<2>(dt); prof_add<3>(dt); prof_add<4>(dt); prof_add<5>(dt); prof_add<6>(dt);
        prof_add<7>(dt); prof_add<8>(dt); prof_add<9>(dt); prof_add<10>(dt); prof_add<11>(dt); prof_add<12>(dt); prof_add<13>(dt);
        prof_add<14>(dt); prof_add<15>(dt); prof_add<16>(dt); prof_add<17>(dt); prof_add<18>(dt);
        prof_cur = slot;
    }
    __device__ __forceinline__ void prof_begin(unsigned long long* out) {
        for (int i = 0; i < 20; ++i) prof_acc[i] = 0;
        prof_out = out; prof_cur = 18; prof_last = __builtin_readcyclecounter();
    }
    __device__ __forceinline__ void prof_end() {
        prof_stamp(18);
        if (threadIdx.x == 0) for (int i = 0; i < 19; ++i) prof_out[i] = prof_acc[i];
    }
#endif

    WDEV void init_tables() {
        LANES(l) {
            sh->r2a[l] = kR2A[l]; sh->a2r[l] = kA2R[l]; sh->nzbin[l] = l < 50 ? kNzBin[l] : 9;
            for (int d = l; d < kSignWords; d += 64) sh->sign[d] = kBranchInit;
            for (int d = l; d < kResDcWords; d += 64) sh->resdc[d] = kBranchInit;
            if (l < (int)(sizeof(NSum) / 4)) { ((uint32_t*)&sh->ns_left)[l] = 0; ((uint32_t*)&sh->ns_above)[l] = 0; }
            for (int d = l; d < 512; d += 64) sh->inv24[d] = inv24_of((uint32_t)d);
            if (l < 32) {
                int j = 0, c = l;
                while (j < 6 && c >= 7 - j) { c -= 7 - j; ++j; }
                sh->cj[l] = (uint8_t)(l < 28 ? j : 7); sh->cn[l] = (uint8_t)(l < 28 ? c + 1 : 0);
            }
        }
        LSYNC();
    }
    WDEV void stage_component(int c) {
        comp = c; ci = c ? 1 : 0;
        LANES(l) {
            sh->q[l] = img->q[c][l];
            sh->thr[l] = img->min_thresh[c][l];
            if (l >= 8) { sh->icos_x[l] = img->icos_x[c][l]; sh->icos_y[l] = img->icos_y[c][l]; }
            if (l < 15) {         // a divisor of 1 (a DC quantiser can be) is marked by shift 32; a zero never divides (eprior[14], the host's check)
                const uint32_t d = l < 7 ? (uint32_t)img->icos_x[c][(l + 1) * 8] : (l < 14 ? (uint32_t)img->icos_y[c][(l - 6) * 8] : (uint32_t)img->q[c][0]);
                const DivBy m = DivBy::of(d < 2 || d > 0x7fffffffu ? 2u : d);
                if (l < 7) sh->icos_x[l] = (int32_t)m.mul;
                else if (l < 14) sh->icos_y[l - 7] = (int32_t)m.mul;
                else { sh->icos_x[7] = (int32_t)m.mul; sh->icos_y[7] = d == 1 ? 32 : (int32_t)m.shift; }
            }
        }
        LSYNC();
    }

    // ---- the serial code of a round, in either form -----------------------------------------------------------------------
    // SC = false: uniform vector code on bc; SC = true: the state is taken over into SGPRs for the round and handed back by
    // done().  Values a round passes in (packed probabilities read from a lane, Branch words) go through U().
    template <bool SC>
    struct Serial {
        Dec4WaveT& w;
        BoolDec4S s;
        WDEV explicit Serial(Dec4WaveT& w_) : w(w_) { if (SC) s.load(w.bc); }
        WDEV void done() { if (SC) s.store(w.bc); }
        static WDEV uint32_t U(uint32_t x) { return SC ? uni(x) : vec(x); }
        static WDEV bool is(bool c) { return SC ? c : ucond(c); }
        WDEV uint32_t get(uint32_t prob) { return SC ? s.get(w.bc, prob) : w.bc.get(prob); }
        WDEV uint32_t bupd(uint32_t word, uint32_t obs) {
            if (!SC) return bupd_u(word, obs, w.sh->inv24);
            const uint32_t w2 = word + 1u + obs * 255u;
            const uint32_t f = w2 & 255u, t = (w2 >> 8) & 255u;
            if ((f < t ? f : t) == 0) {   // the incremented count was 255
                const uint32_t f0 = word & 255, t0 = (word >> 8) & 255;
                if ((obs ? f0 : t0) == 1) return (word & 0xffff) | ((obs ? 0u : 255u) << 16);
                const uint32_t f2 = obs ? (1 + f0) >> 1 : 129u, t2 = obs ? 129u : (1 + t0) >> 1;
                return f2 | (t2 << 8) | ((((f2 << 8) * uni(w.sh->inv24[f2 + t2])) >> 24) << 16);
            }
            return (w2 & 0xffffu) | ((f * uni(w.sh->inv24[f + t])) & 0xff0000u);
        }
        WDEV uint32_t global_bin(uint32_t idx) {   // a Branch outside the prefetched set: coded straight from HBM
            const uint32_t word = U(vload(w.model + idx));
            const uint32_t bit = get(word >> 16);
            const uint32_t nw = bupd(word, bit);
#if LEP_ON_GPU
            if (threadIdx.x == 0) w.model[idx] = nw;
#else
            w.model[idx] = nw;
#endif
            return bit;
        }
        // unary bins k.. of one packed exponent group (k = first bin to decode); returns the number of ones from bin 0 (<= 4)
        template <int K>
        WDEV int unary_from(uint32_t pk) {
            if (K <= 0) { if (!is(get(pk & 255) != 0)) return 0; }
            if (K <= 1) { if (!is(get((pk >> 8) & 255) != 0)) return 1; }
            if (K <= 2) { if (!is(get((pk >> 16) & 255) != 0)) return 2; }
            if (!is(get(pk >> 24) != 0)) return 3;
            return 4;
        }
        WDEV int unary_tail(uint32_t gbase) {   // exponent bins 8..10 straight from HBM (|v| >= 128: rare)
            int i = 8;
#pragma nounroll
            for (; i < 11; ++i) if (!is(global_bin(row_word(gbase, i)) != 0)) break;
            return i;
        }
        // residual bits b..0 (b <= 3) of |v| from the packed probabilities of the residual group (word i = bit i)
        WDEV uint32_t residual(uint32_t pk, int b, uint32_t v) {
            if (b >= 3) v |= get(pk >> 24) << 3;
            if (b >= 2) v |= get((pk >> 16) & 255) << 2;
            if (b >= 1) v |= get((pk >> 8) & 255) << 1;
            if (b >= 0) v |= get(pk & 255);
            return v;
        }
        // a LEVELS-level binary tree decoded MSB first; the d-th decoded level has 2^d nodes stored as whole groups owned
        // by lanes base + first(d) .., first = 0,1,2,3,5,9 (1,1,1,2,4,8 groups per level); levels 0..2 read fixed lanes
        template <int LEVELS>
        WDEV int tree(const uint32_t* PK, int base) {
            uint32_t pk = U(lepwave::wave_read(PK, base));
            uint32_t n = get(pk & 255);
            pk = U(lepwave::wave_read(PK, base + 1));
            n = (n << 1) | get((pk >> (n * 8)) & 255);
            pk = U(lepwave::wave_read(PK, base + 2));
            n = (n << 1) | get((pk >> (n * 8)) & 255);
#pragma unroll
            for (int d = 3; d < LEVELS; ++d) {
                pk = U(lepwave::wave_read(PK, base + (1 << (d - 2)) + 1 + (int)uni(n >> 2)));
                n = (n << 1) | get((pk >> ((n & 3) * 8)) & 255);
            }
            return (int)uni(n);
        }
    };

    // ---- loads of wave-uniform addresses through the vector cache ----------------------------------------------------------
    static WDEV U4 vload4(const uint32_t* p) {   // every lane loads the same group through the vector cache
#if LEP_ON_GPU
        uintptr_t a = (uintptr_t)p;
        __asm__ volatile("" : "+v"(a));
        return ld4(reinterpret_cast<const uint32_t*>(a));
#else
        return ld4(p);
#endif
    }
    static WDEV uint32_t vload(const uint32_t* p) {   // every lane loads the same word through the vector cache
#if LEP_ON_GPU
        uintptr_t a = (uintptr_t)p;
        __asm__ volatile("" : "+v"(a));
        return *reinterpret_cast<const uint32_t*>(a);
#else
        return *p;
#endif
    }
    // integer IDCT without DC (idct.cc:35-161): lep_v3.h idct_no_dc; S.pix is column-major, LEP_PIX(S, y, x)
    WDEV void idct_rows() { idct_no_dc(sh); }
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
    // owners adapt the used words of one group register: (used, bits) masks per lane (0 = lane not involved).
    // Per word slot the common case is STRAIGHT-LINE code for all 64 lanes, no exec masking: the count is bumped by adding
    // 0 / 1 / 256 to the packed word, the probability recomputed through the reciprocal table and blended in only where the
    // slot is used (15 VALU + 1 LDS read, no scalar instruction -- the branchy form cost 20 VALU + 8 SALU per used slot, and a
    // SALU instruction is worth two VALU ones here).  A count that wraps (once per ~250 observations of a Branch) shows up
    // as a zero count byte; those lanes redo the word with the exact rule (bupd_t) under one ballot branch.
    // ALWAYS: bit k set = slot k is used by some lane in nearly every call, do not spend a ballot on skipping it.
    template <int ALWAYS>
    WDEV void adapt_group(U4* W, const int* used, const int* bits) {
        const uint32_t* inv = sh->inv24;
#ifdef LEP_DEC4_ADAPT_BRANCHY   // the previous form (A/B builds): per-slot ballot, the update under the lanes' exec mask
        {
            LV(int, any0);
#define LEP_SLOT0(k, fld)                                                                                             \
            LANES(l) L(any0) = (L(used) >> k) & 1;                                                                    \
            if (lepwave::wave_ballot(any0)) {                                                                         \
                LANES(l) if ((L(used) >> k) & 1) L(W).fld = bupd_t(L(W).fld, (uint32_t)(L(bits) >> k) & 1u, inv);    \
            }
            LEP_SLOT0(0, x) LEP_SLOT0(1, y) LEP_SLOT0(2, z) LEP_SLOT0(3, w)
#undef LEP_SLOT0
            return;
        }
#endif
        LV(int, any); LV(uint32_t, oldw); LV(uint32_t, low);
#ifdef LEP_DEC4_ALWAYS_OVERRIDE
        constexpr int kAlways = LEP_DEC4_ALWAYS_OVERRIDE;
#else
        constexpr int kAlways = ALWAYS;
#endif
#define LEP_SLOT(k, fld)                                                                                              \
        if (!((kAlways >> k) & 1)) { LANES(l) L(any) = (L(used) >> k) & 1; }                                           \
        if (((kAlways >> k) & 1) || lepwave::wave_ballot(any)) {                                                       \
            LANES(l) {                                                                                                \
                const uint32_t w = L(W).fld;                                                                          \
                const uint32_t u1 = ((uint32_t)(L(used) & L(bits)) >> k) & 1u, u0 = ((uint32_t)(L(used) & ~L(bits)) >> k) & 1u;   \
                const uint32_t w2 = w + (u0 | (u1 << 8));                                                             \
                const uint32_t f = w2 & 255u, t = (w2 >> 8) & 255u;                                                   \
                const uint32_t p = mul24_low(f, inv[f + t]);   /* f * 2^24 / (f + t): the probability is bits 16..23 */ \
                const uint32_t m = mul24(u0 | u1, 0xff0000u);                                                         \
                L(oldw) = w; L(low) = (f < t ? f : t) + ((u0 | u1) ^ 1u);   /* 0 iff a bumped count wrapped */ \
                L(W).fld = (p & m) | (w2 & ~m);                                                                       \
            }                                                                                                         \
            LANES(l) L(any) = L(low) == 0;                                                                            \
            if (lepwave::wave_ballot(any)) {                                                                          \
                LANES(l) if (L(low) == 0) L(W).fld = bupd_t(L(oldw), (uint32_t)(L(bits) >> k) & 1u, inv);             \
            }                                                                                                         \
        }
        LEP_SLOT(0, x) LEP_SLOT(1, y) LEP_SLOT(2, z) LEP_SLOT(3, w)
#undef LEP_SLOT
    }

    // ---- round 1: the 6-bit count of interior non-zeros (model.hh:463-485) --------------------------------------------
    WDEV int round_nz(int nzbin_ctx) {
        LEP_MARK("nz_prefetch");
        LV(U4, W0); LV(uint32_t, a0); LV(uint32_t, PK0);
        LANES(l) {
            uint32_t adr = 0, pk = 0;
            if (l < 17) {
                int i, k;
                if (l < 3) { i = 5 - l; k = 0; } else if (l < 5) { i = 2; k = l - 3; } else if (l < 9) { i = 1; k = l - 5; } else { i = 0; k = l - 9; }
                adr = ctx_nz7(ci, nzbin_ctx) + (uint32_t)i * 32 + (uint32_t)k * 4;
                L(W0) = ld4(model + adr); pk = pack_probs(L(W0));
            }
            L(a0) = adr; L(PK0) = pk;
        }
        LEP_MARK("nz_serial"); LEP_PRIO_SERIAL((SCMASK & 1) != 0);
        int nz;
        {
            Serial<(SCMASK & 1) != 0> sr(*this);
            nz = sr.template tree<6>(PK0, 0);
            sr.done();
            LEP_BINS(nbins += 6);
        }
        LEP_MARK("nz_update"); LEP_PRIO_PARALLEL();
        LV(int, u0); LV(int, b0);
        LANES(l) {
            int u = 0, b = 0;
            if (l < 17) {
                int i, k;
                if (l < 3) { i = 5 - l; k = 0; } else if (l < 5) { i = 2; k = l - 3; } else if (l < 9) { i = 1; k = l - 5; } else { i = 0; k = l - 9; }
                mask_tree(i, k, nz, u, b);
            }
            L(u0) = u; L(b0) = b;
        }
        adapt_group<3>(W0, u0, b0);
        LANES(l) if (L(u0)) st4(model + L(a0), L(W0));
        return nz;
    }

    // ---- round 2 (repeated): interior positions zz0 .. zz0+15 under up to four consecutive "non-zeros left" bins ----------
    // lane = pi + 16 * cand: window position pi, candidate bin nb0 - cand.  W0 = exponent words 0..3, W1 = residual words
    // 0..3 (both kept by the owner); exponent words 4..7 are read on demand and re-read by the owner when used.
    WDEV void round_77(int& zz_io, int& left_io) {
        Dec4Shared& S = *sh;
        LEP_MARK("77_prefetch");
        const int zz0 = zz_io, left0 = left_io, nb0 = nzbin_of(left0);
        LV(U4, W0); LV(U4, W1); LV(uint32_t, a0); LV(uint32_t, a1); LV(uint32_t, PK0); LV(uint32_t, PK1); LV(int, ok);
        LANES(l) {
            const int pi = l & 15, cand = l >> 4, p = zz0 + pi, nb = nb0 - cand;
            uint32_t adr0 = 0, adr1 = 0, pk0 = 0, pk1 = 0;
            // candidate `cand` can only be in force at window position pi if enough non-zeros can have come before it
            const int valid = p < 49 && nb >= 1 && cand < LEP_DEC4_CANDS && pi >= left0 - nzhi_of(nb < 0 ? 0 : nb);
            if (valid) {
                adr0 = ctx4_exp7(ci, nb, p, S.bsr[p]);
                adr1 = ctx4_res(ci, S.a2r[p], nb);
                L(W0) = ld4(model + adr0);
                pk0 = pack_probs(L(W0));
                L(W1) = ld4(model + adr1); pk1 = pack_probs(L(W1));

            }
            L(a0) = adr0; L(a1) = adr1; L(PK0) = pk0; L(PK1) = pk1; L(ok) = valid;
        }
        LSYNC();
        // ---- serial (uniform vector) -----------------------------------------------------------------------------
        LEP_MARK("77_serial"); LEP_PRIO_SERIAL((SCMASK & 2) != 0);
        int zz, left = left0, cand = 0;
        const int pi_end = zz0 + 16 < 49 ? 16 : 49 - zz0;   // window positions 0 .. pi_end-1
        {
            // The loop is written around its commonest iteration, a zero coefficient (one bin, bit 0): `lane` is the only
            // induction variable of the inner loop (position and candidate are folded into it), so that iteration is the bin
            // itself + one lane read + add / compare / branch.  Runs while position < pi_end && left > 0 && cand < CANDS
            // (true on entry: the caller checks zz < 49 && left > 0).
            typedef Serial<(SCMASK & 2) != 0> SR;
            SR sr(*this);
            uint32_t sgw = SR::U(S.sign[ci * 48]);
            int pi = 0;
#pragma nounroll
            for (;;) {
                int lane = pi + 16 * cand;
                const int lane_end = pi_end + 16 * cand;
                uint32_t pkv;
                bool found = false;
#pragma nounroll
                for (;;) {
                    pkv = SR::U(lepwave::wave_read(PK0, lane));
                    LEP_BINS(++nbins);
                    if (SR::is(sr.get(pkv & 255) != 0)) { found = true; break; }
                    if (++lane >= lane_end) break;
                }
                pi = lane - 16 * cand;
                if (!found) break;   // the window is exhausted
                zz = zz0 + pi;
                int len = sr.template unary_from<1>(pkv);   // bin 0 was a one: 1..4
                if (len == 4) {
                    // exponent words 4..7 (|v| >= 8: 5 % of the interior non-zeros) are not prefetched -- that third of the
                    // round's traffic was almost all waste; the serial code reads the group when it gets there, the owner
                    // re-reads it to adapt
                    len = 4 + sr.template unary_from<0>(SR::U(pack_probs(vload4(model + ctx4_exp7(ci, nb0 - cand, zz, (int)uni(S.bsr[zz])) + kGS))));
                    if (len == 8) len = sr.unary_tail(ctx4_exp7(ci, nb0 - cand, zz, (int)uni(S.bsr[zz])));
                }
                LEP_BINS(nbins += (uint32_t)(2 * len - (len == 11)));
                const uint32_t pos = sr.get(sgw >> 16);
                sgw = sr.bupd(sgw, pos);
                --left;
                uint32_t v = 1u << (len - 1);
                if (len > 1) {
                    int b = len - 2;
                    if (b >= 4) {
                        const uint32_t rbase = ctx4_res(ci, (int)uni(S.a2r[zz]), nb0 - cand);
#pragma nounroll
                        for (; b >= 4; --b) v |= sr.global_bin(row_word(rbase, b)) << b;
                    }
                    v = sr.residual(SR::U(lepwave::wave_read(PK1, lane)), b, v);
                }
                S.here[zz] = (int16_t)(pos ? (int)v : -(int)v);
                ++pi;
                if (left == 0) break;
                cand = nb0 - nzbin_of(left);
                if (cand >= LEP_DEC4_CANDS || pi >= pi_end) break;
            }
            zz = zz0 + pi;
            sr.done();
            S.sign[ci * 48] = sgw;
        }
        LSYNC();
        // ---- owners adapt ---------------------------------------------------------------------------------------------------
        LEP_MARK("77_update"); LEP_PRIO_PARALLEL();
        LV(int, nzw);
        LANES(l) L(nzw) = l < 16 && zz0 + l < zz && S.here[zz0 + l] != 0;
        const uint32_t nzmask = (uint32_t)lepwave::wave_ballot(nzw);
        LV(int, u0); LV(int, b0); LV(int, u1); LV(int, b1); LV(int, u2); LV(int, b2);
        LANES(l) {
            int ua = 0, ba = 0, ub = 0, bb = 0, uc = 0, bcc = 0;
            const int pi = l & 15, cand_l = l >> 4, p = zz0 + pi;
            if (L(ok) && p < zz) {
                const int left_at = left0 - __builtin_popcount(nzmask & ((1u << pi) - 1));
                if (left_at > 0 && nb0 - (int)S.nzbin[left_at] == cand_l) {
                    const int cf = S.here[p];
                    const int v = cf < 0 ? -cf : cf, len = bitlen((uint32_t)v);
                    mask_exp(0, len, ua, ba);
                    mask_res(len - 2, v, ub, bb);
                    mask_exp(4, len, uc, bcc);
                }
            }
            L(u0) = ua; L(b0) = ba; L(u1) = ub; L(b1) = bb; L(u2) = uc; L(b2) = bcc;
        }
        adapt_group<3>(W0, u0, b0);
        adapt_group<0>(W1, u1, b1);
        LANES(l) {
            if (L(u0)) st4(model + L(a0), L(W0));
            if (L(u1)) st4(model + L(a1), L(W1));
        }
        if (lepwave::wave_ballot(u2)) {   // exponent words 4..7: re-read by the owner (rare in the interior)
            LV(U4, W2);
            LANES(l) if (L(u2)) L(W2) = ld4(model + L(a0) + kGS);
            adapt_group<0>(W2, u2, b2);
            LANES(l) if (L(u2)) st4(model + L(a0) + kGS, L(W2));
        }
        LSYNC();
        zz_io = zz; left_io = left;
    }

    // ---- round 3: both edge count trees and both edges (decoder.cc:27-141) ------------------------------------------------
    // lanes e*28 + combo: edge e (0 horizontal, 1 vertical), combo = (position j, non-zeros-left n) with 1 <= n <= 7-j;
    // W0 / W1 = exponent words 0..3 / 4..7, W2 = residual words 0..3.  Lanes 56..58 / 59..61: the two 3-level count trees.
    WDEV int round_edges(int nz, int eob_x, int eob_y, bool has_left, bool has_above) {
        Dec4Shared& S = *sh;
        LEP_MARK("edge_prefetch");
        LV(U4, W0); LV(U4, W1); LV(U4, W2); LV(uint32_t, a0); LV(uint32_t, a2); LV(uint32_t, PK0); LV(uint32_t, PK1); LV(uint32_t, PK2);
        LV(uint32_t, INFO);   // lanes e*28 + combo_base(j): sign slot | threshold << 8 | threshold ctx << 16 | bsr << 24 | bad prior << 31
        LANES(l) {
            uint32_t adr0 = 0, adr2 = 0, pk0 = 0, pk1 = 0, pk2 = 0, info = 0;
            if (l < 56) {
                const int e = l >= 28 ? 1 : 0, c = l - e * 28, j = S.cj[c], n = S.cn[c];
                const bool horizontal = e == 0;
                const int coord = horizontal ? j + 1 : (j + 1) * 8;
                const int32_t prior = S.eprior[e * 7 + j];
                const uint32_t ap = prior < 0 ? 0u - (uint32_t)prior : (uint32_t)prior;
                const int bsr = bitlen(ap > 1023 ? 1023 : ap);
                adr0 = ctx4_expx(ci, n, horizontal ? j : j + 7, bsr);
                adr2 = ctx4_res(ci, coord, n);
                L(W0) = ld4(model + adr0); L(W2) = ld4(model + adr2);
                pk0 = pack_probs(L(W0)); pk2 = pack_probs(L(W2));
                L(W1) = U4{0, 0, 0, 0};   // exponent words 4..7 (|v| >= 8) are read on demand, like the interior's
                const int16_t p16 = (int16_t)prior;
                const int thr = S.thr[coord];
                const uint32_t tctx = (uint32_t)imin((int)((ap & 0xffff) >> thr), 255);
                info = (uint32_t)((ci * 4 + (p16 == 0 ? 0 : (p16 > 0 ? 1 : 2))) * 12 + bsr) | ((uint32_t)thr << 8) | (tctx << 16) |
                       ((uint32_t)bsr << 24) | ((S.eprior[14] >> (e * 7 + j)) & 1 ? 0x80000000u : 0u);
            } else if (l < 62) {
                const int e = l >= 59 ? 1 : 0, lv = l - 56 - 3 * e;
                adr0 = ctx_nzedge(e == 0, ci, e == 0 ? eob_x : eob_y, (nz + 3) / 7) + (uint32_t)(2 - lv) * 4;
                L(W0) = ld4(model + adr0); pk0 = pack_probs(L(W0));
            }
            L(a0) = adr0; L(a2) = adr2; L(PK0) = pk0; L(PK1) = pk1; L(PK2) = pk2; L(INFO) = info;
        }
        LSYNC();
        // ---- serial (uniform vector) -----------------------------------------------------------------------------
        LEP_MARK("edge_serial"); LEP_PRIO_SERIAL((SCMASK & 4) != 0);
        int ne[2] = {0, 0}, rc = 0;
        typedef Serial<(SCMASK & 4) != 0> SR;
        SR sr(*this);
#pragma nounroll
        for (int e = 0; e < 2 && !rc; ++e) {
            const bool horizontal = e == 0;
            ne[e] = sr.template tree<3>(PK0, 56 + 3 * e);
            LEP_BINS(nbins += 3);
            int left = ne[e];
            if (!left) continue;
            const int a_off = horizontal ? 50 : 57;
            // lane of (position j, `left` non-zeros left) = e * 28 + combo_base(j) + left - 1, and combo_base(j + 1) =
            // combo_base(j) + 7 - j: the lane is the induction variable (+ step per position, - 1 per non-zero), INFO / PK0 / PK2
            // are all read from it (every combo lane of a position carries the position's INFO)
            int lane = e * 28 + left - 1, step = 7;
#pragma nounroll
            for (;;) {
                if (SR::is(left > step)) {
                    // More non-zeros claimed than positions left: no encoder writes that, a damaged stream can hold it.  The
                    // reference indexes exponent_counts_x_ / residual_noise_counts_ with the claimed count all the same
                    // (decoder.cc:58-141); no lane holds such a pair, so the rest of this edge is coded straight from HBM.
                    // Once true it stays true to the end of the edge.
#pragma nounroll
                    for (int j = 7 - step; j < 7 && left; ++j) {
                        const uint32_t info = lepwave::wave_read(INFO, e * 28 + combo_base(j));
                        if (info >> 31) { rc = 43; break; }
                        const int coord = horizontal ? j + 1 : (j + 1) * 8;
                        const uint32_t gb = ctx4_expx(ci, left, horizontal ? j : j + 7, (int)((info >> 24) & 15));
                        int len = 0;
#pragma nounroll
                        for (; len < 11; ++len) if (!SR::is(sr.global_bin(row_word(gb, len)) != 0)) break;
                        LEP_BINS(nbins += (uint32_t)(len ? 2 * len + 1 - (len == 11) : 1));
                        if (!len) continue;
                        const int sslot = (int)(info & 255);
                        const uint32_t sgw = SR::U(S.sign[sslot]);
                        const uint32_t pos = sr.get(sgw >> 16);
                        S.sign[sslot] = sr.bupd(sgw, pos);
                        uint32_t v = 1u << (len - 1);
                        int b = len - 2;
                        const int thr = (int)((info >> 8) & 15);
                        if (b >= thr) {
                            const uint32_t Tt = ctx_thresh(ci, (int)((info >> 16) & 255), imin(len - thr, 7));
                            int sx = 1;
#pragma nounroll
                            for (; b >= thr; --b) {
                                const uint32_t bit = sr.global_bin(Tt + (uint32_t)sx);
                                v |= bit << b;
                                sx = imin((sx << 1) | (int)uni(bit), 127);
                            }
                        }
                        const uint32_t rbase = ctx4_res(ci, coord, left);
#pragma nounroll
                        for (; b >= 0; --b) v |= sr.global_bin(row_word(rbase, b)) << b;
                        S.here[a_off + j] = (int16_t)(pos ? (int)v : -(int)v);
                        --left;
                    }
                    break;
                }
                const uint32_t info = lepwave::wave_read(INFO, lane);
                if (info >> 31) { rc = 43; break; }
                const uint32_t pkv = SR::U(lepwave::wave_read(PK0, lane));
                LEP_BINS(++nbins);
                if (SR::is(sr.get(pkv & 255) != 0)) {
                    const int j = 7 - step;
                    const int coord = horizontal ? j + 1 : (j + 1) * 8;
                    int len = sr.template unary_from<1>(pkv);
                    if (len == 4) {
                        len = 4 + sr.template unary_from<0>(SR::U(pack_probs(vload4(model + ctx4_expx(ci, left, horizontal ? j : j + 7, (int)((info >> 24) & 15)) + kGS))));
                        if (len == 8) len = sr.unary_tail(ctx4_expx(ci, left, horizontal ? j : j + 7, (int)((info >> 24) & 15)));
                    }
                    LEP_BINS(nbins += (uint32_t)(2 * len - (len == 11)));
                    const int sslot = (int)(info & 255);
                    const uint32_t sgw = SR::U(S.sign[sslot]);
                    const uint32_t pos = sr.get(sgw >> 16);
                    S.sign[sslot] = sr.bupd(sgw, pos);
                    uint32_t v = 1u << (len - 1);
                    if (len > 1) {
                        int b = len - 2;
                        const int thr = (int)((info >> 8) & 15);
                        if (b >= thr) {
                            const uint32_t Tt = ctx_thresh(ci, (int)((info >> 16) & 255), imin(len - thr, 7));
                            int sx = 1;
#pragma nounroll
                            for (; b >= thr; --b) {
                                const uint32_t bit = sr.global_bin(Tt + (uint32_t)sx);
                                v |= bit << b;
                                sx = imin((sx << 1) | (int)uni(bit), 127);
                            }
                        }
                        if (b >= 4) {
                            const uint32_t rbase = ctx4_res(ci, coord, left);
#pragma nounroll
                            for (; b >= 4; --b) v |= sr.global_bin(row_word(rbase, b)) << b;
                        }
                        v = sr.residual(SR::U(lepwave::wave_read(PK2, lane)), b, v);
                    }
                    S.here[a_off + j] = (int16_t)(pos ? (int)v : -(int)v);
                    --lane;
                    if (--left == 0) break;
                }
                lane += step;
                if (--step == 0) break;
            }
        }
        sr.done();
        LSYNC();
        if (rc) return rc;
        // ---- owners adapt ---------------------------------------------------------------------------------------------------
        LEP_MARK("edge_update"); LEP_PRIO_PARALLEL();
        LV(int, enz);
        LANES(l) L(enz) = l >= 50 && S.here[l] != 0;   // aligned 50..56 horizontal, 57..63 vertical
        const uint64_t em = lepwave::wave_ballot(enz);
        const uint32_t mh = (uint32_t)(em >> 50) & 0x7f, mv = (uint32_t)(em >> 57) & 0x7f;
        const int neh = ne[0], nev = ne[1];
        LV(int, u0); LV(int, b0); LV(int, u1); LV(int, b1); LV(int, u2); LV(int, b2);
        LANES(l) {
            int ua = 0, ba = 0, ub = 0, bb = 0, uc = 0, bcc = 0;
            if (l < 56) {
                const int e = l >= 28 ? 1 : 0, c = l - e * 28, j = S.cj[c], n = S.cn[c];
                const uint32_t mk = e ? mv : mh;
                const int left_at = (e ? nev : neh) - __builtin_popcount(mk & ((1u << j) - 1));
                if (left_at == n) {   // n >= 1: the position was visited with n non-zeros left
                    const int cf = S.here[(e ? 57 : 50) + j];
                    const int v = cf < 0 ? -cf : cf, len = bitlen((uint32_t)v);
                    mask_exp(0, len, ua, ba);
                    mask_exp(4, len, ub, bb);
                    mask_res(imin(len - 2, (int)S.thr[e ? (j + 1) * 8 : j + 1] - 1), v, uc, bcc);
                }
            } else if (l < 62) {
                const int e = l >= 59 ? 1 : 0, lv = l - 56 - 3 * e;
                mask_tree(2 - lv, 0, e ? nev : neh, ua, ba);
            }
            L(u0) = ua; L(b0) = ba; L(u1) = ub; L(b1) = bb; L(u2) = uc; L(b2) = bcc;
        }
        adapt_group<3>(W0, u0, b0);
        LANES(l) if (L(u0)) st4(model + L(a0), L(W0));
        if (lepwave::wave_ballot(u1)) {
            LANES(l) if (L(u1)) L(W1) = ld4(model + L(a0) + kGS);   // read on demand by the serial code: the owner re-reads it
            adapt_group<0>(W1, u1, b1);
            LANES(l) if (L(u1)) st4(model + L(a0) + kGS, L(W1));
        }
        if (lepwave::wave_ballot(u2)) {
            adapt_group<0>(W2, u2, b2);
            LANES(l) if (L(u2)) st4(model + L(a2), L(W2));
        }
        LSYNC();
        return 0;
    }

    // ---- round 4: DC (decoder.cc:240-318, model.hh:674-832) -----------------------------------------------------------------
    WDEV void round_dc(int pred, int a, int b17, int sctx) {
        Dec4Shared& S = *sh;
        LEP_MARK("dc_prefetch");
        LV(U4, W0); LV(uint32_t, a0); LV(uint32_t, PK0); LV(uint32_t, RW);
        LANES(l) {
            uint32_t adr = 0, pk = 0, rw = 0;
            if (l < 3) { adr = ctx_expdc(a, b17) + (uint32_t)l * 4; L(W0) = ld4(model + adr); pk = pack_probs(L(W0)); }
            else if (l < 13) rw = S.resdc[a * 12 + (l - 3)];   // residual Branch of bit l-3
            L(a0) = adr; L(PK0) = pk; L(RW) = rw;
        }
        LEP_MARK("dc_serial"); LEP_PRIO_SERIAL((SCMASK & 8) != 0);
        const int sslot = ci * 48 + sctx;
        typedef Serial<(SCMASK & 8) != 0> SR;
        SR sr(*this);
        int len = sr.template unary_from<0>(SR::U(lepwave::wave_read(PK0, 0)));
        if (len == 4) {
            len += sr.template unary_from<0>(SR::U(lepwave::wave_read(PK0, 1)));
            if (len == 8) {
                uint32_t pk = SR::U(lepwave::wave_read(PK0, 2));
#pragma nounroll
                for (; len < 11; ++len) { if (!SR::is(sr.get(pk & 255) != 0)) break; pk >>= 8; }
            }
        }
        LEP_BINS(nbins += (uint32_t)(len ? 2 * len + 1 - (len == 11) : 1));
        uint32_t v = 0, pos = 1;
        if (len) {
            const uint32_t sgw = SR::U(S.sign[sslot]);
            pos = sr.get(sgw >> 16);
            S.sign[sslot] = sr.bupd(sgw, pos);
            v = 1u << (len - 1);
#pragma nounroll
            for (int i = len - 2; i >= 0; --i) v |= sr.get(SR::U(lepwave::wave_read(RW, 3 + i)) >> 16) << i;
        }
        sr.done();
        v = vec(v);
        int d = (int16_t)(pos ? (int)v : -(int)v);
        int dc = d + pred;
        dc = dc < -1024 ? dc + 2049 : dc;
        dc = dc > 1024 ? dc - 2049 : dc;
        S.here[49] = (int16_t)dc;
        LSYNC();
        // owners: exponent groups (lanes 0..2), residual Branches (lanes 3..12)
        LEP_MARK("dc_update"); LEP_PRIO_PARALLEL();
        LV(uint32_t, VV); LV(int, u0); LV(int, b0);
        LANES(l) L(VV) = v;   // uniform -> per lane (identity on the GPU)
        LANES(l) {
            int u = 0, b = 0;
            if (l < 3) mask_exp(l * 4, len, u, b);
            L(u0) = u; L(b0) = b;
        }
        adapt_group<7>(W0, u0, b0);
        LANES(l) {
            if (L(u0)) st4(model + L(a0), L(W0));
            if (l >= 3 && l < 13 && l - 3 <= len - 2) S.resdc[a * 12 + (l - 3)] = bupd_t(L(RW), (L(VV) >> (l - 3)) & 1u, S.inv24);
        }
        LSYNC();
    }

    // Decodes one block into sh->here (aligned order). left / above / aleft / ns_* are staged by the caller.
    WDEV int decode_block(bool has_left, bool has_above) {
        Dec4Shared& S = *sh;
        // ---- contexts that do not depend on this block's bits ---------------------------------------------
        LEP_MARK("prologue");
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
        const int nzbin_ctx = (int)uni((uint32_t)nzbin_of(nzctx));
        LSYNC();

        const int nz = round_nz(nzbin_ctx);
        if (nz > 49) return 7;
        {
            int zz = 0, left = nz;
#pragma nounroll
            while (zz < 49 && left > 0) round_77(zz, left);
        }
        // the interior is complete: eob_x / eob_y (encoder.cc:246-250) and the Lakhani priors (model.hh:928-1071) of all
        // 14 edge positions, lane-parallel
        LEP_MARK("lakhani");
        int eob_x, eob_y;
        {
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
                            prior = div_by((int32_t)acc, (uint32_t)(hz ? S.icos_x[j] : S.icos_y[j]), DivBy::shift_of((uint32_t)icos[0]));
                        } else bad = 1;
                    }
                    S.eprior[l] = prior;
                }
                L(tx) = ex; L(ty) = ey; L(badf) = bad;
            }
            eob_x = lepwave::wave_max(tx); eob_y = lepwave::wave_max(ty);
            const uint64_t badmask = lepwave::wave_ballot(badf);
            LANES(l) if (l == 0) S.eprior[14] = (int32_t)(uint32_t)badmask;   // bit p: position p's prior needs a division by zero
            LSYNC();
        }
        {
            const int rc = round_edges(nz, eob_x, eob_y, has_left, has_above);
            if (rc) return rc;
        }
        // DC prediction (model.hh:674-832): IDCT of the ACs, 16 edge estimates on 16 lanes
        LEP_MARK("idct_dcpred");
        idct_rows();
        int pred, a, b17, sctx;
        {
            LV(int, emin); LV(int, emax); LV(int, s0); LV(int, s1); LV(int, tmp);
            LANES(l) {
                int ev = 0, have = 0;
                if (l < 8 && has_left) { have = 1; ev = (int16_t)(S.ns_left.vert[l] - half16(LEP_PIX(S, l, 0) - LEP_PIX(S, l, 1)) - (LEP_PIX(S, l, 0) + 1024)); }
                if (l >= 8 && l < 16 && has_above) { const int i = l - 8; have = 1; ev = (int16_t)(S.ns_above.horiz[i] - half16(LEP_PIX(S, 0, i) - LEP_PIX(S, 1, i)) - (LEP_PIX(S, 0, i) + 1024)); }
                L(emax) = have ? ev : -0x7fffffff;
                L(emin) = have ? -ev : -0x7fffffff;
                L(s0) = l < 8 ? ev : 0;
                L(s1) = (l >= 8 && l < 16) ? ev : 0;
            }
            const int mx = lepwave::wave_max(emax), mn = -lepwave::wave_max(emin);
            const int sumL = lepwave::wave_sum(s0), sumA = lepwave::wave_sum(s1);
            int32_t avgmed = 0, unc = 0, unc2 = 0;
            if (has_left || has_above) {
                int sum0 = has_left ? sumL : sumA, sum1 = (has_left && has_above) ? sumA : sum0;
                avgmed = (sum0 + sum1) >> 1;
                unc = (mx - mn) >> 3;
                sum0 -= avgmed; sum1 -= avgmed;
                unc2 = (iabs(sum0) < iabs(sum1) ? sum0 : sum1) >> 3;
            }
            const uint32_t dc_shift = (uint32_t)S.icos_y[7];
            pred = ((dc_shift == 32 ? avgmed : div_by(avgmed, (uint32_t)S.icos_x[7], dc_shift)) + 4) >> 3;
            a = imin(bitlen((uint32_t)iabs(unc) & 0xffff), 11);
            b17 = imin(bitlen((uint32_t)iabs(unc2) & 0xffff), 16);
            sctx = unc2 >= 0 ? (unc2 == 0 ? 3 : 2) : 1;
        }
        round_dc(pred, a, b17, sctx);
        // ---- neighbour summary (block_context.hh:44-78) -------------------------------------------------------------
        LEP_MARK("publish");
        pad_block((uint32_t)pred, sh);
        LANES(l) {
            if (l < 16) {
                const int i = l & 7;
                const int dcq = S.here[49] * (int)S.q[0];
                if (l < 8) S.ns_here.horiz[i] = (int16_t)(dcq + LEP_PIX(S, 7, i) + 1024 + half16(LEP_PIX(S, 7, i) - LEP_PIX(S, 6, i)));
                else S.ns_here.vert[i] = (int16_t)(dcq + LEP_PIX(S, i, 7) + 1024 + half16(LEP_PIX(S, i, 7) - LEP_PIX(S, i, 6)));
            }
            if (l == 16) S.ns_here.nz = nz;
        }
        LSYNC();
        return 0;
    }

    WDEV int run(const ImageDev* image, const SegDev& seg, uint32_t* model_words, NSum* ns, Dec4Shared* shared, const uint8_t* stream,
                 uint32_t len) {
        img = image; model = model_words; sh = shared; nbins = 0;
        LEP_PRIO_PARALLEL();
        init_tables();
        bc.init_stream(stream, len);
        bool top[3] = {true, true, true};
        for (uint32_t idx = 0;; ++idx) {
            RowSpec r = row_spec(image, idx);
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
            // a truncated image's last row ends with the last coded block (at least one block of a row is always coded)
            const int coded_here = (int)img->coded_blocks[comp] - yb * w;
            const int x_end = imin(w, coded_here < 1 ? 1 : coded_here);
            for (int x = 0; x < x_end; ++x) {
                LEP_MARK("staging");
        LANES(l) {
                    if (x) { sh->left[l] = sh->here[l]; sh->aleft[l] = sh->above[l]; }
                    if (l < (int)(sizeof(NSum) / 4)) {
                        if (x) ((uint32_t*)&sh->ns_left)[l] = ((const uint32_t*)&sh->ns_here)[l];
                        if (has_above) ((uint32_t*)&sh->ns_above)[l] = L(nxt_ns);
                    }
                }
                LSYNC();
                LANES(l) {
                    if (has_above) sh->above[l] = L(nxt_above);
                    if (has_above && x + 1 < x_end) {
                        L(nxt_above) = arow[(int64_t)(x + 1) * 64 + l];
                        if (l < (int)(sizeof(NSum) / 4)) L(nxt_ns) = ((const uint32_t*)&narow[x + 1])[l];
                    }
                }
                LSYNC();
                int rc = decode_block(x > 0, has_above);
                if (rc) return rc;
                LEP_MARK("store");
        LANES(l) {
                    row[(int64_t)x * 64 + l] = sh->here[l];
                    if (l < (int)(sizeof(NSum) / 4)) ((uint32_t*)&nrow[x])[l] = ((const uint32_t*)&sh->ns_here)[l];
                }
            }
        }
        return 0;
    }
};
typedef Dec4WaveT<LEP_DEC4_SCALAR> Dec4Wave;

}  // namespace lep4
