// lep_v3.h -- shared pieces of the "v3" kernels (lep_dec3.h, lep_enc3.h).
//
// What changes against v2 (lep_enc2.h / lep_dec2.h), and why (measured on MI355X, profiles/r01_v2_*):
// v2 spent 570 (encode) / 945 (decode) shader cycles per bin per wave, almost all of it on the one serial
// lane: every bin was an LDS round trip + a 64-bit bool-coder step + a Branch division + an HBM store.
// v3 keeps one wavefront per thread segment but strips the serial lane to the bool-coder recurrence:
//   * the model is laid out so that every context's Branch words start on a 16-byte boundary (exponent /
//     residual rows padded 11,10 -> 12 words); a "group" = 4 consecutive words = one dwordx4 load;
//   * the lane that prefetched a group OWNS it for the round: it publishes the words to LDS, keeps them in
//     registers, and after the serial lane has decoded the round it adapts its own four Branches
//     (branch.hh:82-100) from the decoded coefficient and writes the group back with one dwordx4 store --
//     no per-bin model traffic and no divisions on the serial lane;
//   * the bool decoder uses a 32-bit window refilled 3 bytes at a time from a dword loaded one refill ahead
//     (same bit decisions as boolreader.hh:376-416, zero bits past the end of the stream).
// The model layout is not part of the .lep format, so it may differ between kernels.
#pragma once
#include "lep_core.h"
#include "lep_wave.h"

namespace lep3 {
using namespace lepdev;

// ---- model layout (32-bit words: false_count | true_count << 8 | probability << 16) ------------------
enum : uint32_t {
    kNz7x7 = 0,                                // [2][26][6][32]      model.hh:463-485
    kNz1x8 = kNz7x7 + 2 * 26 * 6 * 32,         // [2][8][8][3][4]     vertical edge count tree
    kNz8x1 = kNz1x8 + 2 * 8 * 8 * 3 * 4,       // [2][8][8][3][4]     horizontal edge count tree
    kExpDc = kNz8x1 + 2 * 8 * 8 * 3 * 4,       // [12][17][12]        (11 used)
    kRes = kExpDc + 12 * 17 * 12,              // [2][64][10][12]     (10 used)
    kExpX = kRes + 2 * 64 * 10 * 12,           // [2][10][15][12][12] (11 used)
    kExp7 = kExpX + 2 * 10 * 15 * 12 * 12,     // [2][10][49][12][12] (11 used)
    kThresh = kExp7 + 2 * 10 * 49 * 12 * 12,   // [2][256][8][128]
    kModelWords = kThresh + 2 * 256 * 8 * 128
};
static_assert(kModelWords == 739472 && kModelWords % 4 == 0, "v3 model layout");
// the sign [2][4][12] and DC-residual [12][12] Branches never leave LDS (they are re-used inside a block)
constexpr int kSignWords = 96, kResDcWords = 144;

struct U4 { uint32_t x, y, z, w; };

// (f << 8) / (f + t) for 1 <= f,t <= 255 (Branch::optimize, branch.hh:108-125).  On the GPU: reciprocal estimate in
// float + exact integer fix-up (|error| of the estimate < 1, so one step either way suffices; checked exhaustively
// against integer division by lep_gpu_selftest / tests/test_gpu_parity.py).  No table, no memory access.
WDEV uint32_t prob_of(uint32_t f, uint32_t t) {
#if LEP_ON_GPU
    const uint32_t n = f << 8, d = f + t;
    uint32_t q = (uint32_t)((float)n * __builtin_amdgcn_rcpf((float)d));
    const int r = (int)n - (int)(q * d);
    q += r >= (int)d ? 1u : 0u;
    q -= r < 0 ? 1u : 0u;
    return q;
#else
    return (f << 8) / (f + t);
#endif
}

// ---- wave-uniform ("scalar") plumbing ---------------------------------------------------------------------
// The serial part of the coder is written as wave-UNIFORM code (every lane computes the same values), which hipcc
// places in SGPRs and executes on the scalar unit: no exec masking, scalar branches, and the vector ALUs stay free
// for the other waves of the SIMD.  uni() pins a value that is uniform by construction (an LDS word read at a uniform
// address) into an SGPR; uload() reads mutable HBM data at a uniform address through the VECTOR cache (the scalar
// cache is not coherent with the vector stores that maintain the model).
WDEV uint32_t uni(uint32_t v) {
#if LEP_ON_GPU
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
    return v;
#endif
}
// the opposite of uni(): hides from the compiler that a value is wave-uniform, so that what is computed from it is
// placed on the vector ALUs (used to move part of the serial work off the scalar unit)
WDEV uint32_t vec(uint32_t v) {
#if LEP_ON_GPU
    __asm__ volatile("" : "+v"(v));
#endif
    return v;
}
WDEV uint32_t uload(const uint32_t* p) {
#if LEP_ON_GPU
    uintptr_t a = (uintptr_t)p;
    __asm__ volatile("" : "+v"(a));
    return uni(*reinterpret_cast<const uint32_t*>(a));
#else
    return *p;
#endif
}

// ceil(2^32 / d): (n * inv) >> 32 == n / d exactly for n < 2^16, d < 2^9 (tests/emu: exhaustive check).  Read-only,
// so the scalar path fetches it with s_load through the scalar cache.
struct InvTable {
    uint32_t v[512];
    constexpr InvTable() : v() {
        for (int d = 2; d < 512; ++d) v[d] = (uint32_t)((0x100000000ull + (uint64_t)d - 1) / (uint64_t)d);
    }
};
#ifdef __HIP_DEVICE_COMPILE__
__constant__ static const InvTable kInv = InvTable();
#else
static const InvTable kInv = InvTable();
#endif
// Branch::record_obs_and_update for a wave-uniform word (scalar unit: s_load of the reciprocal + s_mul_hi).
// Straight-line common case; the count-overflow case (once per ~250 observations of a Branch) is the only branch.
WDEV uint32_t bupd_s(uint32_t w, int obs) {
    uint32_t f = (w & 255) + (uint32_t)(obs ^ 1), t = ((w >> 8) & 255) + (uint32_t)obs;
    if (__builtin_expect((f | t) > 255, 0)) {   // the incremented count was 255
        const uint32_t f0 = w & 255, t0 = (w >> 8) & 255;
        if ((obs ? f0 : t0) == 1) return (w & 0xffff) | ((obs ? 0u : 255u) << 16);
        f = obs ? (1 + f0) >> 1 : 129u;
        t = obs ? 129u : (1 + t0) >> 1;
    }
    const uint32_t p = (uint32_t)(((uint64_t)(f << 8) * kInv.v[f + t]) >> 32);
    return f | (t << 8) | (p << 16);
}

// Branch::record_obs_and_update (branch.hh:82-100) on the packed word
WDEV uint32_t bupd(uint32_t w, int obs) {
    uint32_t f = w & 255, t = (w >> 8) & 255;
    const uint32_t mine = obs ? t : f, other = obs ? f : t;
    if (mine == 255) {
        if (other == 1) return (w & 0xffff) | ((obs ? 0u : 255u) << 16);
        f = (1 + f) >> 1; t = (1 + t) >> 1;
        if (obs) t = 129; else f = 129;
    } else {
        if (obs) ++t; else ++f;
    }
    return f | (t << 8) | (prob_of(f, t) << 16);
}

// 16-byte group load / store (memcpy form: no type punning; with the alignment hint hipcc emits one dwordx4 access)
WDEV U4 ld4(const uint32_t* p) { U4 v; __builtin_memcpy(&v, __builtin_assume_aligned(p, 16), 16); return v; }
WDEV void st4(uint32_t* p, const U4& v) { __builtin_memcpy(__builtin_assume_aligned(p, 16), &v, 16); }
// the four probabilities of a group, one per byte (word j -> bits 8j..8j+7): what the serial lane reads via readlane
WDEV uint32_t pack_probs(const U4& w) {
    return ((w.x >> 16) & 255) | ((w.y >> 8) & 0xff00) | (w.z & 0xff0000) | ((w.w << 8) & 0xff000000u);
}

// ---- integer IDCT without DC (idct.cc:35-161) shared by both coder kernels --------------------------------------------
// SH has: int16_t here[64] (aligned order), uint8_t r2a[64], uint16_t q[64], int32_t t[64] (16-byte aligned),
// int16_t pix[64] (16-byte aligned).  LDS instructions cost 2-4 VALU ones here (profiles/r02n_inst_rates.txt), so the
// passes are arranged around wide accesses: all 64 lanes dequantise into raster order (t), 8 lanes read a row with two
// 16-byte loads and write their results TRANSPOSED, 8 lanes read a column the same way and store its 8 pixels with one
// 16-byte write -- pix is stored column-major: LEP_PIX(S, y, x).  24 LDS instructions instead of 48.
#define LEP_PIX(S, y, x) (S).pix[(x) * 8 + (y)]
template <class SH>
WDEV void idct_no_dc(SH* sh) {
    constexpr int w1 = 2841, w2 = 2676, w3 = 2408, w5 = 1609, w6 = 1108, w7 = 565, r2 = 181;
    constexpr int w1pw7 = w1 + w7, w1mw7 = w1 - w7, w2pw6 = w2 + w6, w2mw6 = w2 - w6, w3pw5 = w3 + w5, w3mw5 = w3 - w5;
    LANES(l) sh->t[l] = l ? (int32_t)sh->here[sh->r2a[l]] * (int32_t)sh->q[l] : 0;   // raster order, DC left out
    LSYNC();
    LV(int32_t, o0); LV(int32_t, o1); LV(int32_t, o2); LV(int32_t, o3); LV(int32_t, o4); LV(int32_t, o5); LV(int32_t, o6); LV(int32_t, o7);
    LANES(l) if (l < 8) {
        const int32_t* in = sh->t + l * 8;
        const U4 a = ld4(reinterpret_cast<const uint32_t*>(in)), b = ld4(reinterpret_cast<const uint32_t*>(in + 4));
        int32_t x0 = (int32_t)(a.x << 11) + 128;
        int32_t x1 = (int32_t)(b.x << 11);
        int32_t x2 = (int32_t)b.z, x3 = (int32_t)a.z, x4 = (int32_t)a.y, x5 = (int32_t)b.w, x6 = (int32_t)b.y, x7 = (int32_t)a.w, x8;
        x8 = w7 * (x4 + x5); x4 = x8 + w1mw7 * x4; x5 = x8 - w1pw7 * x5;
        x8 = w3 * (x6 + x7); x6 = x8 - w3mw5 * x6; x7 = x8 - w3pw5 * x7;
        x8 = x0 + x1; x0 -= x1;
        x1 = w6 * (x3 + x2); x2 = x1 - w2pw6 * x2; x3 = x1 + w2mw6 * x3;
        x1 = x4 + x6; x4 -= x6; x6 = x5 + x7; x5 -= x7;
        x7 = x8 + x3; x8 -= x3; x3 = x0 + x2; x0 -= x2;
        x2 = (r2 * (x4 + x5) + 128) >> 8;
        x4 = (r2 * (x4 - x5) + 128) >> 8;
        L(o0) = (x7 + x1) >> 8; L(o1) = (x3 + x2) >> 8; L(o2) = (x0 + x4) >> 8; L(o3) = (x8 + x6) >> 8;
        L(o4) = (x8 - x6) >> 8; L(o5) = (x0 - x4) >> 8; L(o6) = (x3 - x2) >> 8; L(o7) = (x7 - x1) >> 8;
    }
    LSYNC();   // every row has been read before any is overwritten (the emulation steps lane by lane)
    LANES(l) if (l < 8) {   // transposed: column k of row l goes to t[k * 8 + l]
        int32_t* t = sh->t + l;
        t[0] = L(o0); t[8] = L(o1); t[16] = L(o2); t[24] = L(o3); t[32] = L(o4); t[40] = L(o5); t[48] = L(o6); t[56] = L(o7);
    }
    LSYNC();
    LANES(l) if (l < 8) {
        const int32_t* in = sh->t + l * 8;   // column l, rows 0..7
        const U4 a = ld4(reinterpret_cast<const uint32_t*>(in)), b = ld4(reinterpret_cast<const uint32_t*>(in + 4));
        int32_t y0 = (int32_t)(a.x << 8) + 8192, y1 = (int32_t)(b.x << 8);
        int32_t y2 = (int32_t)b.z, y3 = (int32_t)a.z, y4 = (int32_t)a.y, y5 = (int32_t)b.w, y6 = (int32_t)b.y, y7 = (int32_t)a.w, y8;
        y8 = w7 * (y4 + y5) + 4; y4 = (y8 + w1mw7 * y4) >> 3; y5 = (y8 - w1pw7 * y5) >> 3;
        y8 = w3 * (y6 + y7) + 4; y6 = (y8 - w3mw5 * y6) >> 3; y7 = (y8 - w3pw5 * y7) >> 3;
        y8 = y0 + y1; y0 -= y1;
        y1 = w6 * (y3 + y2) + 4; y2 = (y1 - w2pw6 * y2) >> 3; y3 = (y1 + w2mw6 * y3) >> 3;
        y1 = y4 + y6; y4 -= y6; y6 = y5 + y7; y5 -= y7;
        y7 = y8 + y3; y8 -= y3; y3 = y0 + y2; y0 -= y2;
        y2 = (r2 * (y4 + y5) + 128) >> 8;
        y4 = (r2 * (y4 - y5) + 128) >> 8;
        const uint32_t p0 = (uint32_t)(uint16_t)((y7 + y1) >> 11) | ((uint32_t)(uint16_t)((y3 + y2) >> 11) << 16);
        const uint32_t p1 = (uint32_t)(uint16_t)((y0 + y4) >> 11) | ((uint32_t)(uint16_t)((y8 + y6) >> 11) << 16);
        const uint32_t p2 = (uint32_t)(uint16_t)((y8 - y6) >> 11) | ((uint32_t)(uint16_t)((y0 - y4) >> 11) << 16);
        const uint32_t p3 = (uint32_t)(uint16_t)((y3 - y2) >> 11) | ((uint32_t)(uint16_t)((y7 - y1) >> 11) << 16);
        st4(reinterpret_cast<uint32_t*>(sh->pix + l * 8), U4{p0, p1, p2, p3});   // column l, rows 0..7
    }
    LSYNC();
}

// ---- bool decoder (boolreader.hh:184-258, 376-416; boolreader.cc:25-34) -----------------------------------
// 64-bit window refilled with one ALIGNED dword at a time; the next dword is requested one refill ahead and only
// touched when it is consumed, so the load latency is off the serial chain.  Bytes outside [0, len) of the stream are
// never loaded and read as zero bits (LOTS_OF_BITS behaviour of the reference past the end of the data).
struct BoolDec3 {
    uint64_t value;    // top-aligned window
    int count;         // valid bits - 8
    uint32_t range;
    const uint32_t* words;   // aligned dword that holds stream byte 0
    uint32_t first, end;     // stream bytes live at byte offsets [first, end) from `words`
    uint32_t wi;             // index of the dword `raw` holds
    uint32_t raw;            // words[wi] as loaded (0 if it holds no stream byte)

    WDEV uint32_t fetch(uint32_t k) const { return k * 4 < end ? words[k] : 0u; }
    WDEV void refill() {
        const uint32_t lo = wi * 4;
        uint32_t w = __builtin_bswap32(raw);
        int nbits = 32;
        if (lo + 4 > end) w = lo < end ? (w & ~(0xffffffffu >> ((end - lo) * 8))) : 0u;   // bytes past the end -> 0
        if (lo < first) { w <<= (first - lo) * 8; nbits -= (int)(first - lo) * 8; }              // bytes before the start
        value |= ((uint64_t)w << 32) >> (count + 8);
        count += nbits;
        ++wi;
        raw = fetch(wi);
    }
    WDEV void init_stream(const uint8_t* p, uint32_t n) {
        const uint32_t mis = (uint32_t)((uintptr_t)p & 3);
        words = reinterpret_cast<const uint32_t*>(p - mis);
        first = mis; end = mis + n;
        value = 0; count = -8; range = 255; wi = 0;
        raw = fetch(0);
        refill();
        get(128);
    }
    WDEV int get(uint32_t prob) {
        const uint32_t split = 1 + (((range - 1) * prob) >> 8);
        if (count < 0) refill();
        const uint32_t big = split << 24;
        const int bit = (uint32_t)(value >> 32) >= big;
        if (bit) { range -= split; value -= (uint64_t)big << 32; } else range = split;
        const int shift = __builtin_clz(range) - 24;
#ifdef LEP_TRACE_GET
        LEP_TRACE_GET(prob, bit);
#endif
        range <<= shift; value <<= shift; count -= shift;
        return bit;
    }
};

// context bases ------------------------------------------------------------------------------------------
WDEV uint32_t ctx_exp7(int ci, int nb, int zz, int bsr) { return kExp7 + ((((uint32_t)ci * 10 + nb) * 49 + zz) * 12 + bsr) * 12; }
WDEV uint32_t ctx_expx(int ci, int ne, int zig15, int bsr) { return kExpX + ((((uint32_t)ci * 10 + ne) * 15 + zig15) * 12 + bsr) * 12; }
WDEV uint32_t ctx_res(int ci, int coord, int nb) { return kRes + (((uint32_t)ci * 64 + coord) * 10 + nb) * 12; }
WDEV uint32_t ctx_expdc(int a, int b) { return kExpDc + ((uint32_t)a * 17 + b) * 12; }
WDEV uint32_t ctx_nz7(int ci, int bin) { return kNz7x7 + ((uint32_t)ci * 26 + bin) * 192; }
WDEV uint32_t ctx_nzedge(bool horizontal, int ci, int eob, int nzq) { return (horizontal ? kNz8x1 : kNz1x8) + (((uint32_t)ci * 8 + eob) * 8 + nzq) * 12; }
WDEV uint32_t ctx_thresh(int ci, int ctx, int lenq) { return kThresh + (((uint32_t)ci * 256 + ctx) * 8 + lenq) * 128; }

}  // namespace lep3
