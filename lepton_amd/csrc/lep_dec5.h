// lep_dec5.h -- "v5" decoder: a WORKGROUP of NW wavefronts decodes NW thread segments, and everything that is not the serial
// bool-decoder chain of a segment is done once for all NW of them.
//
// Why (MI355X, profiles/pmc_traffic.json, r02j_dec4_phase_wave_time.txt): lep_dec4.h runs one wavefront per segment at 8
// wavefronts per SIMD and sits on the vector issue port -- 2946 VALU + 1955 SALU instructions per 8x8 block, of which only ~2150
// are the serial rounds (97 bins x ~20).  The rest is lane-parallel work that occupies a 64-lane instruction stream with 14-17
// useful lanes (neighbour priors, Lakhani edge priors, IDCT + DC prediction, neighbour summary, block store) or recomputes four
// Branch words per owner lane four slots at a time (the owner updates).  A narrowed exec mask does not make a vector instruction
// cheaper, a second segment in the other lanes does.  So:
//   * S phases (every wavefront, its own segment): prefetch of the context groups of a round, the serial round itself
//     (lep_dec4.h's code: uniform vector or scalar-unit form), and the owners' (used, bits) masks -- pushed, with the counts the
//     owner holds, as 16-byte entries into the segment's update queue in LDS instead of being adapted on the spot;
//   * P phases (ONE wavefront of the workgroup, the role rotating with the block step so that the work spreads over the SIMDs):
//     a 16-lane row of the wavefront per segment -- priors (prologue), Lakhani + eob, IDCT + DC prediction --
//     and the update pass, lane = queue entry of any of the NW segments: counts bumped four Branches at a time as packed bytes,
//     the four probabilities recomputed with v_rcp_f32, the 12-byte record written back.  A Branch cannot repeat inside a block
//     (every context family is indexed by the coefficient position; sign and DC-residual Branches live in LDS, the threshold
//     family and the exponent / residual tails are read-modify-written directly by the serial code and never queued), so
//     deferring the updates to the end of the block changes nothing.
//   * workgroup barriers between the phases; NW = 1 is the same code with wave-level ordering only (small launches, emulation).
//
// Model layout of this kernel (the in-memory model is not part of the format): a GROUP of four Branches is a
// 12-byte RECORD { f0 f1 f2 f3 | t0 t1 t2 t3 | p0 p1 p2 p3 } -- false counts, true counts, probabilities, one byte each -- so the
// prefetch takes the probabilities as they lie (no unpacking), the update bumps counts with two packed adds, and a context group
// is 12 bytes of HBM instead of 16 (four candidates of a position: 48 bytes).  A saturated "true" Branch (counts (1, 255) met by
// another true: probability 0 from then on, branch.hh:86-91) is stored as f = 0, which the probability formula maps to 0 by
// itself.  Trees are stored in heap order (node h = 1 .. 2^levels - 1, group h >> 2): the 6-bit count of interior non-zeros is
// 16 records, one load per lane for the whole tree.  The threshold family keeps lep_v3.h's 32-bit words.
//
// Syntax / contexts: src/vp8/decoder/decoder.cc:27-141,167-318; src/vp8/model/model.hh:463-485,852-871,1033-1122,674-832;
// src/vp8/model/branch.hh:82-100; src/vp8/decoder/boolreader.hh:184-258,376-416 (as lep_dec4.h, whose results this reproduces).
#pragma once
#include "lep_dec4.h"

namespace lep5d {
using namespace lep4;

// ---- model layout ------------------------------------------------------------------------------------------------------------
enum : uint32_t {
    kGNz7 = 0,                                     // [2][10][16]            heap order, nodes 1..63
    kGNzE = kGNz7 + 2 * 10 * 16,                   // [2 hv][2][8][8][2]     heap order, nodes 1..7
    kGExpDc = kGNzE + 2 * 2 * 8 * 8 * 2,           // [12][17][3]
    kGRes = kGExpDc + 12 * 17 * 3,                 // [2][64][3][10]         group k of (coord, nb): + k * 10
    kGExpX = kGRes + 2 * 64 * 3 * 10,              // [2][15][12][3][10]
    kGExp7 = kGExpX + 2 * 15 * 12 * 3 * 10,        // [2][49][12][3][10]
    kGroups = kGExp7 + 2 * 49 * 12 * 3 * 10
};
static_assert(kGroups == 51364 && kGroups < 65536, "a group index fits 16 bits (queue entries)");
constexpr uint32_t kThreshOff5 = (kGroups * 3 + 63) & ~63u;   // first word of the threshold family (lep3 words)
constexpr uint32_t kThreshWords5 = 2 * 256 * 8 * 128;
constexpr uint32_t kModelWords5 = kThreshOff5 + kThreshWords5;
static_assert(kModelWords5 % 64 == 0, "a segment's model starts on a 256-byte boundary");
constexpr uint32_t kRecInitF = 0x01010101u, kRecInitT = 0x01010101u, kRecInitP = 0x80808080u;

WDEV uint32_t g_nz7(int ci, int bin) { return kGNz7 + ((uint32_t)ci * 10 + bin) * 16; }
WDEV uint32_t g_nze(bool horizontal, int ci, int eob, int nzq) { return kGNzE + ((((uint32_t)(horizontal ? 1 : 0) * 2 + ci) * 8 + eob) * 8 + nzq) * 2; }
WDEV uint32_t g_expdc(int a, int b) { return kGExpDc + ((uint32_t)a * 17 + b) * 3; }
WDEV uint32_t g_res(int ci, int coord, int nb) { return kGRes + (((uint32_t)ci * 64 + coord) * 3) * 10 + nb; }
WDEV uint32_t g_expx(int ci, int ne, int zig15, int bsr) { return kGExpX + ((((uint32_t)ci * 15 + zig15) * 12 + bsr) * 3) * 10 + ne; }
WDEV uint32_t g_exp7(int ci, int nb, int zz, int bsr) { return kGExp7 + ((((uint32_t)ci * 49 + zz) * 12 + bsr) * 3) * 10 + nb; }
constexpr uint32_t kGStep = 10;   // from group k of a row to group k + 1
WDEV uint32_t thresh5(int ci, int ctx, int lenq) { return kThreshOff5 + (((uint32_t)ci * 256 + ctx) * 8 + lenq) * 128; }

struct R3 { uint32_t cF, cT, P; };
WDEV R3 ld3(const uint32_t* p) { R3 v; __builtin_memcpy(&v, __builtin_assume_aligned(p, 4), 12); return v; }
WDEV void st3(uint32_t* p, const R3& v) { __builtin_memcpy(__builtin_assume_aligned(p, 4), &v, 12); }

// floor(256 f / (f + t)), 0 <= f <= 255, 1 <= t <= 255 (f = 0: the saturated state).  One v_rcp_f32: the scaled reciprocal is
// exact (a power of two), the fused multiply-add rounds once, so this is lep_enc5.h's prob16 (all count pairs checked on the
// device by lep_gpu_selftest).
WDEV uint32_t prob8(uint32_t f, uint32_t t) {
#if LEP_ON_GPU
    return (uint32_t)__builtin_fmaf((float)f, __builtin_amdgcn_rcpf((float)(f + t)) * 256.0f, 0x1p-10f);
#else
    return (f << 8) / (f + t);
#endif
}
WDEV uint32_t prob4(uint32_t cF, uint32_t cT) {
    return prob8(cF & 255u, cT & 255u) | (prob8((cF >> 8) & 255u, (cT >> 8) & 255u) << 8) | (prob8((cF >> 16) & 255u, (cT >> 16) & 255u) << 16) |
           (prob8(cF >> 24, cT >> 24) << 24);
}
// Branch::record_obs_and_update (branch.hh:82-100) on one Branch's counts in this kernel's encoding
WDEV void upd_ft(uint32_t& f, uint32_t& t, uint32_t obs) {
    if (f == 0) { if (!obs) f = 2; }   // saturated true ((1, 255), probability 0): a true leaves it, a false makes it (2, 255)
    else if (obs) {
        if (t == 255) { if (f == 1) f = 0; else { f = (1 + f) >> 1; t = 129; } }
        else ++t;
    } else {
        if (f == 255) { if (t != 1) { t = (1 + t) >> 1; f = 129; } }   // (255, 1) met by a false stays: probability 255 = the formula
        else ++f;
    }
}
WDEV uint32_t spread4(uint32_t m) { return (m * 0x00204081u) & 0x01010101u; }   // bit k of a nibble -> byte k
WDEV uint32_t haszero(uint32_t v) { return (v - 0x01010101u) & ~v & 0x80808080u; }   // != 0 iff some byte of v is zero

// ---- LDS ---------------------------------------------------------------------------------------------------------------------
constexpr int kQCap = 128;   // update-queue entries per segment (a typical block queues ~45; a full queue is flushed by its own wavefront)
struct Ctl5 {
    int32_t active;        // the segment has a block in flight
    int32_t cur;           // ring slot of the block in flight (blk[cur] = here, blk[cur ^ 1] = left; abv[cur] = above, abv[cur ^ 1] = above-left)
    int32_t has_left, has_above, ci;
    int32_t nzbin_ctx;     // P prologue -> S
    int32_t nz, eob_x, eob_y, badmask;   // S / P lakhani
    int32_t pred, a, b17, sctx;          // P dcpred -> S
    int32_t qn;            // entries in the queue
    const int32_t* icos_x; // ImageDev tables of the component in flight (global memory)
    const int32_t* icos_y;
};
struct Seg5 {
    alignas(16) int16_t blk[2][64];   // aligned order
    alignas(16) int16_t abv[2][64];
    alignas(16) int32_t t[64];        // IDCT intermediate
    alignas(16) int16_t pix[64];      // column-major: LEP_PIX
    alignas(4) uint8_t bsr[64];
    int32_t eprior[16];
    NSum nsb[2];                      // ns_here / ns_left ring
    NSum nsa;                         // ns_above
    uint32_t sign[kSignWords];        // resident Branches (lep_v3.h words)
    uint32_t resdc[kResDcWords];
    alignas(8) uint16_t q[64];
    alignas(4) uint8_t thr[64];
    Ctl5 c;
    alignas(16) U4 qe[kQCap];         // update queue: x = group | used << 16 | bits << 20 | reload << 24, y = false counts, z = true counts
};
template <int NW>
struct Dec5Shared {
    alignas(4) uint8_t r2a[64], a2r[64];
    uint8_t cj[32], cn[32];
    alignas(8) uint8_t tap[16][8];    // aligned indices of the 8 taps of edge position j (0..6 horizontal, 7..13 vertical)
    Seg5 seg[NW];
};

#if LEP_ON_GPU
static __device__ __forceinline__ int lep5_wave_now() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
#define LEP5_WAVES(w) for (int w = lep5_wave_now(), lep_w1_ = 1; lep_w1_; lep_w1_ = 0)
#define LEP5_WV(w) wv[0]
#define LEP5_NWV(NW) 1
#define LEP5_GSYNC(NW) do { if ((NW) > 1) __syncthreads(); else LSYNC(); } while (0)
#define LEP5_ROLE(NW, step, role) ((NW) == 1 || lep5_wave_now() == (int)(((step) + (role)) & ((NW) - 1)))
#else
#define LEP5_WAVES(w) for (int w = 0; w < NW; ++w)
#define LEP5_WV(w) wv[w]
#define LEP5_NWV(NW) (NW)
#define LEP5_GSYNC(NW) ((void)0)
#define LEP5_ROLE(NW, step, role) true
#endif

#ifndef LEP_DEC5_SCALAR
#define LEP_DEC5_SCALAR 2   // serial rounds on the scalar unit: 1 = count tree, 2 = 7x7 interior, 4 = edges, 8 = DC (as LEP_DEC4_SCALAR)
#endif

// private state of one wavefront = one thread segment
struct Wave5 {
    const ImageDev* img;
    SegDev seg;
    uint32_t* model;
    NSum* ns;
    const uint8_t* stream;
    uint32_t stream_len;
    BoolDec4 bc;
    uint32_t nbins;
    int rc;
    // the walk over the segment's rows (lepton_codec.hh:41-100)
    uint32_t idx;
    int comp, ci, w, yb, x, x_end;
    bool has_above;
    uint32_t not_top;   // bit c: a row of component c has been decoded in this segment
    int16_t* row;
    const int16_t* arow;
    NSum* nrow;
    const NSum* narow;
    LV(int16_t, nxt_above);
    LV(uint32_t, nxt_ns);
};

template <int NW>
struct Dec5Group {
    Dec5Shared<NW>* sh;
    Wave5 wv[LEP5_NWV(NW)];

    // ---- the serial code of a round, in either form (lep_dec4.h Dec4Wave::Serial, on this kernel's model) -----------------------
    template <bool SC>
    struct Serial {
        Wave5& w;
        Seg5& S;
        BoolDec4S s;
        WDEV Serial(Wave5& w_, Seg5& S_) : w(w_), S(S_) { if (SC) s.load(w.bc); }
        WDEV void done() { if (SC) s.store(w.bc); }
        static WDEV uint32_t U(uint32_t x) { return SC ? uni(x) : vec(x); }
        static WDEV bool is(bool c) { return SC ? c : ucond(c); }
        WDEV uint32_t get(uint32_t prob) { return SC ? s.get(w.bc, prob) : w.bc.get(prob); }
        static WDEV uint32_t uprob(uint32_t f, uint32_t t) {   // uniform (f << 8) / (f + t)
            if (SC) return (uint32_t)(((uint64_t)(f << 8) * kInv.v[f + t]) >> 32);
            return prob_of(f, t);
        }
        // Branch update of a resident (LDS) word: lep_v3.h packed form
        WDEV uint32_t bupd(uint32_t word, uint32_t obs) {
            uint32_t f = (word & 255) + (obs ^ 1u), t = ((word >> 8) & 255) + obs;
            if (is((f | t) > 255)) {   // the incremented count was 255
                const uint32_t f0 = word & 255, t0 = (word >> 8) & 255;
                if (is((obs ? f0 : t0) == 1)) return (word & 0xffff) | ((obs ? 0u : 255u) << 16);
                f = obs ? (1 + f0) >> 1 : 129u;
                t = obs ? 129u : (1 + t0) >> 1;
            }
            return f | (t << 8) | (uprob(f, t) << 16);
        }
        WDEV uint32_t global_bin(uint32_t idx) {   // a Branch of the threshold family (32-bit words): coded straight from HBM
            const uint32_t word = U(Dec4Wave::vload(w.model + idx));
            const uint32_t bit = get(word >> 16);
            const uint32_t nw = bupd(word, bit);
#if LEP_ON_GPU
            if ((threadIdx.x & 63) == 0) w.model[idx] = nw;
#else
            w.model[idx] = nw;
#endif
            return bit;
        }
        // Branch `slot` of record `g` (a group no lane holds): coded straight from HBM
        WDEV uint32_t global_bin_rec(uint32_t g, int slot) {
            uint32_t* p = w.model + g * 3;
            const int sh8 = slot * 8;
            uint32_t cF = U(Dec4Wave::vload(p)), cT = U(Dec4Wave::vload(p + 1)), P = U(Dec4Wave::vload(p + 2));
            const uint32_t bit = get((P >> sh8) & 255);
            uint32_t f = (cF >> sh8) & 255, t = (cT >> sh8) & 255;
            if (is(f == 0)) { if (is(bit == 0)) f = 2; }
            else if (is(bit != 0)) {
                if (is(t == 255)) { if (is(f == 1)) f = 0; else { f = (1 + f) >> 1; t = 129; } }
                else ++t;
            } else {
                if (is(f == 255)) { if (is(t != 1)) { t = (1 + t) >> 1; f = 129; } }
                else ++f;
            }
            const uint32_t m = ~(255u << sh8);
            cF = (cF & m) | (f << sh8); cT = (cT & m) | (t << sh8);
            P = (P & m) | ((f ? uprob(f, t) : 0u) << sh8);
#if LEP_ON_GPU
            if ((threadIdx.x & 63) == 0) { p[0] = cF; p[1] = cT; p[2] = P; }
#else
            p[0] = cF; p[1] = cT; p[2] = P;
#endif
            return bit;
        }
        template <int K>
        WDEV int unary_from(uint32_t pk) {
            if (K <= 0) { if (!is(get(pk & 255) != 0)) return 0; }
            if (K <= 1) { if (!is(get((pk >> 8) & 255) != 0)) return 1; }
            if (K <= 2) { if (!is(get((pk >> 16) & 255) != 0)) return 2; }
            if (!is(get(pk >> 24) != 0)) return 3;
            return 4;
        }
        WDEV int unary_tail(uint32_t g0) {   // exponent bins 8..10 (record g0 + 2 groups), straight from HBM
            int i = 8;
#pragma nounroll
            for (; i < 11; ++i) if (!is(global_bin_rec(g0 + 2 * kGStep, i - 8) != 0)) break;
            return i;
        }
        WDEV uint32_t residual(uint32_t pk, int b, uint32_t v) {
            if (b >= 3) v |= get(pk >> 24) << 3;
            if (b >= 2) v |= get((pk >> 16) & 255) << 2;
            if (b >= 1) v |= get((pk >> 8) & 255) << 1;
            if (b >= 0) v |= get(pk & 255);
            return v;
        }
        // a LEVELS-level binary tree in heap order, MSB first: node h lives in byte h & 3 of the record lane base + (h >> 2) holds
        template <int LEVELS>
        WDEV int tree(const uint32_t* PK, int base) {
            uint32_t h = 1;
#pragma unroll
            for (int d = 0; d < LEVELS; ++d) {
                const uint32_t pk = U(lepwave::wave_read(PK, base + (int)uni(h >> 2)));
                h = (h << 1) | get((pk >> ((h & 3) * 8)) & 255);
            }
            return (int)uni(h) - (1 << LEVELS);
        }
    };

    // ---- owner masks -----------------------------------------------------------------------------------------------------------
    // record `g` of a heap-ordered tree of LEVELS levels that coded `value`: the nodes on the path and their bits
    template <int LEVELS>
    static WDEV void mask_heap(int g, int value, int& used, int& bits) {
        const int V = (1 << LEVELS) + value;
        if (g == 0) {   // nodes 1 (depth 0) and 2, 3 (depth 1)
            const int p1 = V >> (LEVELS - 1);   // 2 or 3
            used = 2 | (1 << p1);
            bits = ((p1 & 1) ? 2 : 0) | (((V >> (LEVELS - 2)) & 1) ? 1 << p1 : 0);
        } else {
            const int d = 33 - __builtin_clz((unsigned)g);   // depth of nodes 4g .. 4g+3 (g = 1: 2, g = 2..3: 3, ...)
            const int pn = V >> (LEVELS - d);
            used = (pn >> 2) == g ? 1 << (pn & 3) : 0;
            bits = d < LEVELS && ((V >> (LEVELS - 1 - d)) & 1) ? used : 0;
        }
    }

    // ---- update queue ------------------------------------------------------------------------------------------------------------
    // every lane with used != 0 appends one entry; `reload`: the lane does not hold the record's counts (the pass loads them)
    WDEV void push(int w, const uint32_t* g, const int* used, const int* bits, const uint32_t* cF, const uint32_t* cT, bool reload) {
        Seg5& S = sh->seg[w];
        const uint64_t m = lepwave::wave_ballot(used);
        if (!m) return;
        const int n = lepwave::popc64(m);
        int qn = (int)uni((uint32_t)S.c.qn);
        if (qn + n > kQCap) { LSYNC(); flush_own(w); qn = 0; }
        LANES(l) {
            if (L(used)) {
                const int at = qn + lepwave::rank_below(m, l);
                S.qe[at] = U4{L(g) | ((uint32_t)L(used) << 16) | ((uint32_t)L(bits) << 20) | (reload ? 1u << 24 : 0u), L(cF), L(cT), 0u};
            }
            if (l == 0) S.c.qn = qn + n;
        }
        LSYNC();
    }

    // a segment's own wavefront empties its queue (a block with more entries than the queue holds: dense images).  Not inlined: the
    // seven push sites of a block would each carry a copy of the pass, and the kernel's hot code should fit the instruction cache.
#if LEP_ON_GPU
    static __attribute__((noinline)) __device__ void flush_cold(Dec5Shared<NW>* sh_, uint32_t* m0, size_t stride, int w) { update_pass_on(sh_, m0, stride, 1u << w); }
#else
    static void flush_cold(Dec5Shared<NW>* sh_, uint32_t* m0, size_t stride, int w) { update_pass_on(sh_, m0, stride, 1u << w); }
#endif
    WDEV void flush_own(int w) { flush_cold(sh, model0, model_stride, w); }
    WDEV void update_pass(uint32_t segmask) { update_pass_on(sh, model0, model_stride, segmask); }
    // lane = entry of any of the segments in `segmask`: counts bumped, probabilities recomputed, record stored
    static WDEV void update_pass_on(Dec5Shared<NW>* sh, uint32_t* model0, size_t model_stride, uint32_t segmask) {
        LEP_MARK("update");
        int cnt[NW], total = 0;
        for (int q = 0; q < NW; ++q) { cnt[q] = ((segmask >> q) & 1) ? (int)uni((uint32_t)sh->seg[q].c.qn) : 0; total += cnt[q]; }
        if (!total) return;
#pragma nounroll
        for (int base = 0; base < total; base += 64) {
            LV(uint32_t, e0); LV(uint32_t, cF); LV(uint32_t, cT); LV(int, slow); LV(int, on); LV(int, rel);
            LV(uint32_t*, rec);
            LANES(l) {
                int e = base + l, q = 0;
                for (int k = 0; k + 1 < NW; ++k) if (q == k && e >= cnt[k]) { e -= cnt[k]; ++q; }
                const int live = base + l < total;
                uint32_t x = 0, f = 0, t = 0;
                uint32_t* r = nullptr;
                if (live) {
                    const U4 en = sh->seg[q].qe[e];
                    x = en.x; f = en.y; t = en.z;
                    r = model0 + (size_t)q * model_stride + (x & 0xffffu) * 3;
                }
                L(e0) = x; L(cF) = f; L(cT) = t; L(on) = live; L(rec) = r; L(rel) = live && ((x >> 24) & 1);
            }
            if (lepwave::wave_ballot(rel)) {
                LANES(l) if (L(rel)) { L(cF) = L(rec)[0]; L(cT) = L(rec)[1]; }
            }
            LANES(l) {
                const uint32_t used = (L(e0) >> 16) & 15u, bits = (L(e0) >> 20) & 15u;
                const uint32_t ub = spread4(used) * 255u;
                const uint32_t nF = L(cF) + spread4(used & ~bits), nT = L(cT) + spread4(used & bits);
                // a count that was 255 wraps to a zero byte; a saturated Branch (f = 0) among the used ones takes the exact rule too
                L(slow) = L(on) && (haszero(nF) | haszero(nT) | haszero(L(cF) | ~ub)) != 0;
                if (!L(slow)) { L(cF) = nF; L(cT) = nT; }
            }
            if (lepwave::wave_ballot(slow)) {
                LANES(l) if (L(slow)) {
                    const uint32_t used = (L(e0) >> 16) & 15u, bits = (L(e0) >> 20) & 15u;
                    uint32_t cf = L(cF), ct = L(cT);
                    for (int k = 0; k < 4; ++k) if ((used >> k) & 1) {
                        uint32_t f = (cf >> (8 * k)) & 255u, t = (ct >> (8 * k)) & 255u;
                        upd_ft(f, t, (bits >> k) & 1u);
                        cf = (cf & ~(255u << (8 * k))) | (f << (8 * k));
                        ct = (ct & ~(255u << (8 * k))) | (t << (8 * k));
                    }
                    L(cF) = cf; L(cT) = ct;
                }
            }
            LANES(l) if (L(on)) st3(L(rec), R3{L(cF), L(cT), prob4(L(cF), L(cT))});
        }
        LANES(l) if (l < NW && ((segmask >> l) & 1)) sh->seg[l].c.qn = 0;
        LSYNC();
    }
    // model of segment q of the group (the update pass reaches every segment's): the segments of a workgroup are neighbours in the
    // launch's model arena
    uint32_t* model0;
    size_t model_stride;

    // ---- S phases ---------------------------------------------------------------------------------------------------------------
    // round 1: the 6-bit count of interior non-zeros (model.hh:463-485): the whole tree, one record per lane
    WDEV int round_nz(int w) {
        Wave5& W = LEP5_WV(w);
        Seg5& S = sh->seg[w];
        const int ci = W.ci;
        LEP_MARK("nz_prefetch");
        LV(uint32_t, g0); LV(uint32_t, cF); LV(uint32_t, cT); LV(uint32_t, PK0);
        const int nzbin_ctx = (int)uni((uint32_t)S.c.nzbin_ctx);
        LANES(l) {
            uint32_t g = 0, f = 0, t = 0, pk = 0;
            if (l < 16) { g = g_nz7(ci, nzbin_ctx) + (uint32_t)l; const R3 r = ld3(W.model + g * 3); f = r.cF; t = r.cT; pk = r.P; }
            L(g0) = g; L(cF) = f; L(cT) = t; L(PK0) = pk;
        }
        LEP_MARK("nz_serial");
        int nz;
        {
            Serial<(LEP_DEC5_SCALAR & 1) != 0> sr(W, S);
            nz = sr.template tree<6>(PK0, 0);
            sr.done();
            LEP_BINS(W.nbins += 6);
        }
        LEP_MARK("nz_update");
        LV(int, u0); LV(int, b0);
        LANES(l) {
            int u = 0, b = 0;
            if (l < 16) mask_heap<6>(l, nz, u, b);
            L(u0) = u; L(b0) = b;
        }
        push(w, g0, u0, b0, cF, cT, false);
        return nz;
    }

    // round 2 (repeated): interior positions zz0 .. zz0+15 under up to four consecutive "non-zeros left" bins (lep_dec4.h round_77)
    WDEV void round_77(int w, int& zz_io, int& left_io) {
        Wave5& W = LEP5_WV(w);
        Seg5& S = sh->seg[w];
        const int ci = W.ci;
        const int cur = (int)uni((uint32_t)S.c.cur);
        int16_t* here = S.blk[cur];
        LEP_MARK("77_prefetch");
        const int zz0 = zz_io, left0 = left_io, nb0 = nzbin_of(left0);
        LV(uint32_t, ga); LV(uint32_t, gb); LV(uint32_t, aF); LV(uint32_t, aT); LV(uint32_t, bF); LV(uint32_t, bT); LV(uint32_t, PK0); LV(uint32_t, PK1); LV(int, ok);
        LANES(l) {
            const int pi = l & 15, cand = l >> 4, p = zz0 + pi, nb = nb0 - cand;
            uint32_t g0 = 0, g1 = 0, f0 = 0, t0 = 0, f1 = 0, t1 = 0, pk0 = 0, pk1 = 0;
            const int valid = p < 49 && nb >= 1 && cand < LEP_DEC4_CANDS && pi >= left0 - nzhi_of(nb < 0 ? 0 : nb);
            if (valid) {
                g0 = g_exp7(ci, nb, p, sh->seg[w].bsr[p]);
                g1 = g_res(ci, sh->a2r[p], nb);
                const R3 r0 = ld3(W.model + g0 * 3), r1 = ld3(W.model + g1 * 3);
                f0 = r0.cF; t0 = r0.cT; pk0 = r0.P; f1 = r1.cF; t1 = r1.cT; pk1 = r1.P;
            }
            L(ga) = g0; L(gb) = g1; L(aF) = f0; L(aT) = t0; L(bF) = f1; L(bT) = t1; L(PK0) = pk0; L(PK1) = pk1; L(ok) = valid;
        }
        LSYNC();
        LEP_MARK("77_serial");
        int zz, left = left0, cand = 0;
        const int pi_end = zz0 + 16 < 49 ? 16 : 49 - zz0;
        {
            typedef Serial<(LEP_DEC5_SCALAR & 2) != 0> SR;
            SR sr(W, S);
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
                    LEP_BINS(++W.nbins);
                    if (SR::is(sr.get(pkv & 255) != 0)) { found = true; break; }
                    if (++lane >= lane_end) break;
                }
                pi = lane - 16 * cand;
                if (!found) break;
                zz = zz0 + pi;
                int len = sr.template unary_from<1>(pkv);
                if (len == 4) {
                    // exponent words 4..7 (|v| >= 8) are not prefetched: the serial code reads that record's probabilities when it
                    // gets there, the update pass loads its counts
                    const uint32_t g = g_exp7(ci, nb0 - cand, zz, (int)uni(S.bsr[zz]));
                    len = 4 + sr.template unary_from<0>(SR::U(Dec4Wave::vload(W.model + (g + kGStep) * 3 + 2)));
                    if (len == 8) len = sr.unary_tail(g);
                }
                LEP_BINS(W.nbins += (uint32_t)(2 * len - (len == 11)));
                const uint32_t pos = sr.get(sgw >> 16);
                sgw = sr.bupd(sgw, pos);
                --left;
                uint32_t v = 1u << (len - 1);
                if (len > 1) {
                    int b = len - 2;
                    if (b >= 4) {
                        const uint32_t rg = g_res(ci, (int)uni(sh->a2r[zz]), nb0 - cand);
#pragma nounroll
                        for (; b >= 4; --b) v |= sr.global_bin_rec(rg + (uint32_t)(b >> 2) * kGStep, b & 3) << b;
                    }
                    v = sr.residual(SR::U(lepwave::wave_read(PK1, lane)), b, v);
                }
                here[zz] = (int16_t)(pos ? (int)v : -(int)v);
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
        LEP_MARK("77_update");
        LV(int, nzw);
        LANES(l) L(nzw) = l < 16 && zz0 + l < zz && here[zz0 + l] != 0;
        const uint32_t nzmask = (uint32_t)lepwave::wave_ballot(nzw);
        LV(int, u0); LV(int, b0); LV(int, u1); LV(int, b1); LV(int, u2); LV(int, b2); LV(uint32_t, gc);
        LANES(l) {
            int ua = 0, ba = 0, ub = 0, bb = 0, uc = 0, bcc = 0;
            const int pi = l & 15, cand_l = l >> 4, p = zz0 + pi;
            if (L(ok) && p < zz) {
                const int left_at = left0 - __builtin_popcount(nzmask & ((1u << pi) - 1));
                if (left_at > 0 && nb0 - nzbin_of(left_at) == cand_l) {
                    const int cf = here[p];
                    const int v = cf < 0 ? -cf : cf, len = bitlen((uint32_t)v);
                    Dec4Wave::mask_exp(0, len, ua, ba);
                    Dec4Wave::mask_res(len - 2, v, ub, bb);
                    Dec4Wave::mask_exp(4, len, uc, bcc);
                }
            }
            L(u0) = ua; L(b0) = ba; L(u1) = ub; L(b1) = bb; L(u2) = uc; L(b2) = bcc; L(gc) = L(ga) + kGStep;
        }
        push(w, ga, u0, b0, aF, aT, false);
        push(w, gb, u1, b1, bF, bT, false);
        push(w, gc, u2, b2, aF, aT, true);
        zz_io = zz; left_io = left;
    }

    // round 3: both edge count trees and both edges (decoder.cc:27-141; lep_dec4.h round_edges)
    // lanes e*28 + combo: edge e, combo = (position j, non-zeros left n) with 1 <= n <= 7-j; lanes 56 + 2e + k: record k of edge e's tree
    WDEV int round_edges(int w) {
        Wave5& W = LEP5_WV(w);
        Seg5& S = sh->seg[w];
        const int ci = W.ci;
        const int cur = (int)uni((uint32_t)S.c.cur);
        int16_t* here = S.blk[cur];
        const int nz = (int)uni((uint32_t)S.c.nz), eob_x = (int)uni((uint32_t)S.c.eob_x), eob_y = (int)uni((uint32_t)S.c.eob_y);
        const uint32_t badmask = uni((uint32_t)S.c.badmask);
        LEP_MARK("edge_prefetch");
        LV(uint32_t, ga); LV(uint32_t, gb); LV(uint32_t, aF); LV(uint32_t, aT); LV(uint32_t, bF); LV(uint32_t, bT); LV(uint32_t, PK0); LV(uint32_t, PK2);
        LV(uint32_t, INFO);   // sign slot | threshold << 8 | threshold ctx << 16 | bsr << 24 | bad prior << 31
        LANES(l) {
            uint32_t g0 = 0, g2 = 0, f0 = 0, t0 = 0, f2 = 0, t2 = 0, pk0 = 0, pk2 = 0, info = 0;
            if (l < 56) {
                const int e = l >= 28 ? 1 : 0, c = l - e * 28, j = sh->cj[c], n = sh->cn[c];
                const bool horizontal = e == 0;
                const int coord = horizontal ? j + 1 : (j + 1) * 8;
                const int32_t prior = S.eprior[e * 7 + j];
                const uint32_t ap = prior < 0 ? 0u - (uint32_t)prior : (uint32_t)prior;
                const int bsr = bitlen(ap > 1023 ? 1023 : ap);
                g0 = g_expx(ci, n, horizontal ? j : j + 7, bsr);
                g2 = g_res(ci, coord, n);
                const R3 r0 = ld3(W.model + g0 * 3), r2 = ld3(W.model + g2 * 3);
                f0 = r0.cF; t0 = r0.cT; pk0 = r0.P; f2 = r2.cF; t2 = r2.cT; pk2 = r2.P;
                const int16_t p16 = (int16_t)prior;
                const int thr = S.thr[coord];
                const uint32_t tctx = (uint32_t)imin((int)((ap & 0xffff) >> thr), 255);
                info = (uint32_t)((ci * 4 + (p16 == 0 ? 0 : (p16 > 0 ? 1 : 2))) * 12 + bsr) | ((uint32_t)thr << 8) | (tctx << 16) |
                       ((uint32_t)bsr << 24) | ((badmask >> (e * 7 + j)) & 1 ? 0x80000000u : 0u);
            } else if (l < 60) {
                const int e = (l - 56) >> 1, k = (l - 56) & 1;
                g0 = g_nze(e == 0, ci, e == 0 ? eob_x : eob_y, (nz + 3) / 7) + (uint32_t)k;
                const R3 r0 = ld3(W.model + g0 * 3);
                f0 = r0.cF; t0 = r0.cT; pk0 = r0.P;
            }
            L(ga) = g0; L(gb) = g2; L(aF) = f0; L(aT) = t0; L(bF) = f2; L(bT) = t2; L(PK0) = pk0; L(PK2) = pk2; L(INFO) = info;
        }
        LSYNC();
        LEP_MARK("edge_serial");
        int ne[2] = {0, 0}, rc = 0;
        typedef Serial<(LEP_DEC5_SCALAR & 4) != 0> SR;
        SR sr(W, S);
#pragma nounroll
        for (int e = 0; e < 2 && !rc; ++e) {
            const bool horizontal = e == 0;
            ne[e] = sr.template tree<3>(PK0, 56 + 2 * e);
            LEP_BINS(W.nbins += 3);
            int left = ne[e];
            if (!left) continue;
            const int a_off = horizontal ? 50 : 57;
            int lane = e * 28 + left - 1, step = 7;
#pragma nounroll
            for (;;) {
                if (SR::is(left > step)) {
                    // more non-zeros claimed than positions left (a damaged stream): the reference indexes its tables with the claimed
                    // count all the same (decoder.cc:58-141); no lane holds such a pair, the rest of this edge is coded from HBM
#pragma nounroll
                    for (int j = 7 - step; j < 7 && left; ++j) {
                        const uint32_t info = lepwave::wave_read(INFO, e * 28 + combo_base(j));
                        if (info >> 31) { rc = 43; break; }
                        const int coord = horizontal ? j + 1 : (j + 1) * 8;
                        const uint32_t gx = g_expx(ci, left, horizontal ? j : j + 7, (int)((info >> 24) & 15));
                        int len = 0;
#pragma nounroll
                        for (; len < 11; ++len) if (!SR::is(sr.global_bin_rec(gx + (uint32_t)(len >> 2) * kGStep, len & 3) != 0)) break;
                        LEP_BINS(W.nbins += (uint32_t)(len ? 2 * len + 1 - (len == 11) : 1));
                        if (!len) continue;
                        const int sslot = (int)(info & 255);
                        const uint32_t sgw = SR::U(S.sign[sslot]);
                        const uint32_t pos = sr.get(sgw >> 16);
                        S.sign[sslot] = sr.bupd(sgw, pos);
                        uint32_t v = 1u << (len - 1);
                        int b = len - 2;
                        const int thr = (int)((info >> 8) & 15);
                        if (b >= thr) {
                            const uint32_t Tt = thresh5(ci, (int)((info >> 16) & 255), imin(len - thr, 7));
                            int sx = 1;
#pragma nounroll
                            for (; b >= thr; --b) {
                                const uint32_t bit = sr.global_bin(Tt + (uint32_t)sx);
                                v |= bit << b;
                                sx = imin((sx << 1) | (int)uni(bit), 127);
                            }
                        }
                        const uint32_t rg = g_res(ci, coord, left);
#pragma nounroll
                        for (; b >= 0; --b) v |= sr.global_bin_rec(rg + (uint32_t)(b >> 2) * kGStep, b & 3) << b;
                        here[a_off + j] = (int16_t)(pos ? (int)v : -(int)v);
                        --left;
                    }
                    break;
                }
                const uint32_t info = lepwave::wave_read(INFO, lane);
                if (info >> 31) { rc = 43; break; }
                const uint32_t pkv = SR::U(lepwave::wave_read(PK0, lane));
                LEP_BINS(++W.nbins);
                if (SR::is(sr.get(pkv & 255) != 0)) {
                    const int j = 7 - step;
                    const int coord = horizontal ? j + 1 : (j + 1) * 8;
                    int len = sr.template unary_from<1>(pkv);
                    if (len == 4) {
                        const uint32_t gx = g_expx(ci, left, horizontal ? j : j + 7, (int)((info >> 24) & 15));
                        len = 4 + sr.template unary_from<0>(SR::U(Dec4Wave::vload(W.model + (gx + kGStep) * 3 + 2)));
                        if (len == 8) len = sr.unary_tail(gx);
                    }
                    LEP_BINS(W.nbins += (uint32_t)(2 * len - (len == 11)));
                    const int sslot = (int)(info & 255);
                    const uint32_t sgw = SR::U(S.sign[sslot]);
                    const uint32_t pos = sr.get(sgw >> 16);
                    S.sign[sslot] = sr.bupd(sgw, pos);
                    uint32_t v = 1u << (len - 1);
                    if (len > 1) {
                        int b = len - 2;
                        const int thr = (int)((info >> 8) & 15);
                        if (b >= thr) {
                            const uint32_t Tt = thresh5(ci, (int)((info >> 16) & 255), imin(len - thr, 7));
                            int sx = 1;
#pragma nounroll
                            for (; b >= thr; --b) {
                                const uint32_t bit = sr.global_bin(Tt + (uint32_t)sx);
                                v |= bit << b;
                                sx = imin((sx << 1) | (int)uni(bit), 127);
                            }
                        }
                        if (b >= 4) {
                            const uint32_t rg = g_res(ci, coord, left);
#pragma nounroll
                            for (; b >= 4; --b) v |= sr.global_bin_rec(rg + (uint32_t)(b >> 2) * kGStep, b & 3) << b;
                        }
                        v = sr.residual(SR::U(lepwave::wave_read(PK2, lane)), b, v);
                    }
                    here[a_off + j] = (int16_t)(pos ? (int)v : -(int)v);
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
        LEP_MARK("edge_update");
        LV(int, enz);
        LANES(l) L(enz) = l >= 50 && here[l] != 0;   // aligned 50..56 horizontal, 57..63 vertical
        const uint64_t em = lepwave::wave_ballot(enz);
        const uint32_t mh = (uint32_t)(em >> 50) & 0x7f, mv = (uint32_t)(em >> 57) & 0x7f;
        const int neh = ne[0], nev = ne[1];
        LV(int, u0); LV(int, b0); LV(int, u1); LV(int, b1); LV(int, u2); LV(int, b2); LV(uint32_t, gc);
        LANES(l) {
            int ua = 0, ba = 0, ub = 0, bb = 0, uc = 0, bcc = 0;
            if (l < 56) {
                const int e = l >= 28 ? 1 : 0, c = l - e * 28, j = sh->cj[c], n = sh->cn[c];
                const uint32_t mk = e ? mv : mh;
                const int left_at = (e ? nev : neh) - __builtin_popcount(mk & ((1u << j) - 1));
                if (left_at == n) {
                    const int cf = here[(e ? 57 : 50) + j];
                    const int v = cf < 0 ? -cf : cf, len = bitlen((uint32_t)v);
                    Dec4Wave::mask_exp(0, len, ua, ba);
                    Dec4Wave::mask_exp(4, len, ub, bb);
                    Dec4Wave::mask_res(imin(len - 2, (int)S.thr[e ? (j + 1) * 8 : j + 1] - 1), v, uc, bcc);
                }
            } else if (l < 60) {
                const int e = (l - 56) >> 1, k = (l - 56) & 1;
                mask_heap<3>(k, e ? nev : neh, ua, ba);
            }
            L(u0) = ua; L(b0) = ba; L(u1) = ub; L(b1) = bb; L(u2) = uc; L(b2) = bcc; L(gc) = L(ga) + kGStep;
        }
        push(w, ga, u0, b0, aF, aT, false);
        push(w, gc, u1, b1, aF, aT, true);
        push(w, gb, u2, b2, bF, bT, false);
        return 0;
    }

    // round 4: DC (decoder.cc:240-318, model.hh:674-832)
    WDEV void round_dc(int w) {
        Wave5& W = LEP5_WV(w);
        Seg5& S = sh->seg[w];
        const int ci = W.ci;
        const int cur = (int)uni((uint32_t)S.c.cur);
        const int pred = (int)uni((uint32_t)S.c.pred), a = (int)uni((uint32_t)S.c.a), b17 = (int)uni((uint32_t)S.c.b17), sctx = (int)uni((uint32_t)S.c.sctx);
        LEP_MARK("dc_prefetch");
        LV(uint32_t, g0); LV(uint32_t, cF); LV(uint32_t, cT); LV(uint32_t, PK0); LV(uint32_t, RW);
        LANES(l) {
            uint32_t g = 0, f = 0, t = 0, pk = 0, rw = 0;
            if (l < 3) { g = g_expdc(a, b17) + (uint32_t)l; const R3 r = ld3(W.model + g * 3); f = r.cF; t = r.cT; pk = r.P; }
            else if (l < 13) rw = S.resdc[a * 12 + (l - 3)];
            L(g0) = g; L(cF) = f; L(cT) = t; L(PK0) = pk; L(RW) = rw;
        }
        LEP_MARK("dc_serial");
        const int sslot = ci * 48 + sctx;
        typedef Serial<(LEP_DEC5_SCALAR & 8) != 0> SR;
        SR sr(W, S);
        int len = sr.template unary_from<0>(SR::U(lepwave::wave_read(PK0, 0)));
        if (len == 4) {
            len += sr.template unary_from<0>(SR::U(lepwave::wave_read(PK0, 1)));
            if (len == 8) {
                uint32_t pk = SR::U(lepwave::wave_read(PK0, 2));
#pragma nounroll
                for (; len < 11; ++len) { if (!SR::is(sr.get(pk & 255) != 0)) break; pk >>= 8; }
            }
        }
        LEP_BINS(W.nbins += (uint32_t)(len ? 2 * len + 1 - (len == 11) : 1));
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
        S.blk[cur][49] = (int16_t)dc;
        LSYNC();
        LEP_MARK("dc_update");
        LV(uint32_t, VV); LV(int, u0); LV(int, b0);
        LANES(l) L(VV) = v;
        LANES(l) {
            int u = 0, b = 0;
            if (l < 3) Dec4Wave::mask_exp(l * 4, len, u, b);
            L(u0) = u; L(b0) = b;
            if (l >= 3 && l < 13 && l - 3 <= len - 2) S.resdc[a * 12 + (l - 3)] = lep3::bupd(L(RW), (int)((L(VV) >> (l - 3)) & 1u));
        }
        push(w, g0, u0, b0, cF, cT, false);
    }

    // ---- the walk (lepton_codec.hh:41-100): next row of the segment, next block of the row -----------------------------------------------
    WDEV bool next_row(int w) {
        Wave5& W = LEP5_WV(w);
        Seg5& S = sh->seg[w];
        for (;; ++W.idx) {
            const RowSpec r = row_spec(W.img, W.idx);
            if (r.done) return false;
            if (r.luma_y >= W.seg.y1 && !W.seg.is_last) return false;
            if (r.skip) continue;
            if (r.luma_y < W.seg.y0) continue;
            const int c = r.component;
            W.comp = c; W.ci = c ? 1 : 0;
            LANES(l) { S.q[l] = W.img->q[c][l]; S.thr[l] = W.img->min_thresh[c][l]; }
            W.w = W.img->width[c]; W.yb = r.curr_y;
            W.row = W.img->blocks[c] + (int64_t)W.yb * W.w * 64;
            W.has_above = ((W.not_top >> c) & 1u) != 0;
            W.arow = W.has_above ? W.row - (int64_t)W.w * 64 : nullptr;
            W.nrow = W.ns + W.img->ns_offset[c] + (W.yb & 1) * W.w;
            W.narow = W.ns + W.img->ns_offset[c] + ((W.yb & 1) ^ 1) * W.w;
            W.not_top |= 1u << c;
            const int coded_here = (int)W.img->coded_blocks[c] - W.yb * W.w;
            W.x_end = imin(W.w, coded_here < 1 ? 1 : coded_here);
            W.x = 0;
            LANES(l) {   // block 0's neighbours straight into the ring; block 1's on their way
                if (W.has_above) {
                    S.abv[0][l] = W.arow[l];
                    if (l < (int)(sizeof(NSum) / 4)) ((uint32_t*)&S.nsa)[l] = ((const uint32_t*)&W.narow[0])[l];
                    if (1 < W.x_end) {
                        L(W.nxt_above) = W.arow[64 + l];
                        if (l < (int)(sizeof(NSum) / 4)) L(W.nxt_ns) = ((const uint32_t*)&W.narow[1])[l];
                    }
                }
                if (l == 0) { S.c.ci = W.ci; S.c.icos_x = W.img->icos_x[c]; S.c.icos_y = W.img->icos_y[c]; }
            }
            ++W.idx;
            return true;
        }
    }
    WDEV void begin_block(int w) {   // the block at W.x goes in flight
        Wave5& W = LEP5_WV(w);
        Seg5& S = sh->seg[w];
        LANES(l) if (l == 0) { S.c.cur = W.x & 1; S.c.has_left = W.x > 0; S.c.has_above = W.has_above; S.c.active = 1; }
    }
    WDEV void finish_block(int w) {   // after the DC round: the finished block's destination, then on to the next one
        Wave5& W = LEP5_WV(w);
        Seg5& S = sh->seg[w];
        ++W.x;
        if (W.x < W.x_end) {
            LANES(l) {
                if (W.has_above) {
                    S.abv[W.x & 1][l] = L(W.nxt_above);
                    if (l < (int)(sizeof(NSum) / 4)) ((uint32_t*)&S.nsa)[l] = L(W.nxt_ns);
                    if (W.x + 1 < W.x_end) {
                        L(W.nxt_above) = W.arow[(int64_t)(W.x + 1) * 64 + l];
                        if (l < (int)(sizeof(NSum) / 4)) L(W.nxt_ns) = ((const uint32_t*)&W.narow[W.x + 1])[l];
                    }
                }
            }
            begin_block(w);
        } else if (next_row(w)) begin_block(w);
        else { LANES(l) if (l == 0) S.c.active = 0; }
    }
    WDEV void fail(int w, int rc) {
        Wave5& W = LEP5_WV(w);
        W.rc = rc;
        LANES(l) if (l == 0) sh->seg[w].c.active = 0;
    }

    // ---- P phases: a row of 16 lanes per segment ---------------------------------------------------------------------------------------
    // neighbour priors of the 49 interior positions (model.hh:852-871) -> bsr; context of the non-zero count; `here` cleared
    WDEV void p_prologue() {
        LEP_MARK("prologue");
        LANES(l) {
            const int sg = l >> 4, i = l & 15;
            if (sg < NW && sh->seg[sg].c.active) {
                Seg5& S = sh->seg[sg];
                const int cur = S.c.cur, hl = S.c.has_left, ha = S.c.has_above;
                const int16_t* L4 = S.blk[cur ^ 1] + 4 * i;
                const int16_t* A4 = S.abv[cur] + 4 * i;
                const int16_t* D4 = S.abv[cur ^ 1] + 4 * i;
                uint32_t packed = 0;
                for (int k = 0; k < 4; ++k) {
                    int prior;
                    if (hl && ha) prior = (uint16_t)((iabs(L4[k]) + iabs(A4[k])) * 13 + 6 * iabs(D4[k])) >> 5;
                    else if (hl) prior = (int16_t)iabs(L4[k]);
                    else if (ha) prior = (int16_t)iabs(A4[k]);
                    else prior = 0;
                    packed |= (uint32_t)bitlen((uint32_t)imin(iabs(prior), 1023)) << (8 * k);
                }
                *reinterpret_cast<uint32_t*>(S.bsr + 4 * i) = packed;
                int16_t* H4 = S.blk[cur] + 4 * i;
                H4[0] = 0; H4[1] = 0; H4[2] = 0; H4[3] = 0;
                if (i == 0) {
                    int nzctx = 0;
                    const int nl = S.nsb[cur ^ 1].nz, na = S.nsa.nz;
                    if (hl && ha) nzctx = (na + nl + 2) / 4;
                    else if (ha) nzctx = (na + 1) / 2;
                    else if (hl) nzctx = (nl + 1) / 2;
                    S.c.nzbin_ctx = nzbin_of(nzctx);
                }
            }
        }
        LSYNC();
    }
    // eob_x / eob_y (encoder.cc:246-250) and the Lakhani priors (model.hh:928-1071) of the 14 edge positions
    WDEV void p_lakhani() {
        LEP_MARK("lakhani");
        LV(int, tx); LV(int, ty); LV(int, badf);
        LANES(l) {
            const int sg = l >> 4, i = l & 15;
            int ex = 0, ey = 0, bad = 0;
            if (sg < NW && sh->seg[sg].c.active) {
                Seg5& S = sh->seg[sg];
                const int cur = S.c.cur;
                const int16_t* here = S.blk[cur];
                for (int k = 0; k < 4; ++k) {
                    const int p = 4 * i + k;
                    if (p < 49 && here[p] != 0) { const int coord = sh->a2r[p]; ex = imax(ex, coord & 7); ey = imax(ey, coord >> 3); }
                }
                if (i < 14) {
                    const bool hz = i < 7;
                    const int j = hz ? i : i - 7;
                    int32_t prior = 0;
                    if (hz ? S.c.has_above : S.c.has_left) {
                        const int16_t* nbr = hz ? S.abv[cur] : S.blk[cur ^ 1];
                        const int32_t* icos = (hz ? S.c.icos_x : S.c.icos_y) + (j + 1) * 8;
                        const uint8_t* tap = sh->tap[i];
                        if (icos[0] != 0) {
                            uint32_t acc = (uint32_t)(int32_t)nbr[tap[0]] * (uint32_t)icos[0];
                            for (int k = 1; k < 8; ++k) {
                                const int32_t xi = here[tap[k]], ai = nbr[tap[k]];
                                const int32_t term = (k & 1) ? xi + ai : xi - ai;
                                acc -= (uint32_t)icos[k] * (uint32_t)term;
                            }
                            prior = (int32_t)acc / icos[0];
                        } else bad = 1;
                    }
                    S.eprior[i] = prior;
                }
            }
            L(tx) = ex; L(ty) = ey; L(badf) = bad;
        }
        LV(int, mx); LV(int, my);
        lepwave::row_incl_max(tx, mx); lepwave::row_incl_max(ty, my);
        const uint64_t bm = lepwave::wave_ballot(badf);
        LANES(l) {
            const int sg = l >> 4, i = l & 15;
            if (sg < NW && i == 15 && sh->seg[sg].c.active) {
                Ctl5& c = sh->seg[sg].c;
                c.eob_x = L(mx); c.eob_y = L(my); c.badmask = (int32_t)((bm >> (16 * sg)) & 0x3fffu);
            }
        }
        LSYNC();
    }
    // integer IDCT of the ACs (idct.cc:35-161; lep_v3.h idct_no_dc, 16 lanes per segment) and the DC prediction (model.hh:674-832)
    WDEV void p_idct_dcpred() {
        LEP_MARK("idct_dcpred");
        constexpr int w1 = 2841, w2 = 2676, w3 = 2408, w5 = 1609, w6 = 1108, w7 = 565, r2 = 181;
        constexpr int w1pw7 = w1 + w7, w1mw7 = w1 - w7, w2pw6 = w2 + w6, w2mw6 = w2 - w6, w3pw5 = w3 + w5, w3mw5 = w3 - w5;
        LANES(l) {
            const int sg = l >> 4, i = l & 15;
            if (sg < NW && sh->seg[sg].c.active) {
                Seg5& S = sh->seg[sg];
                const int16_t* here = S.blk[S.c.cur];
                for (int k = 0; k < 4; ++k) { const int r = 4 * i + k; S.t[r] = r ? (int32_t)here[sh->r2a[r]] * (int32_t)S.q[r] : 0; }
            }
        }
        LSYNC();
        LV(int32_t, o0); LV(int32_t, o1); LV(int32_t, o2); LV(int32_t, o3); LV(int32_t, o4); LV(int32_t, o5); LV(int32_t, o6); LV(int32_t, o7);
        LANES(l) {
            const int sg = l >> 4, i = l & 15;
            if (sg < NW && i < 8 && sh->seg[sg].c.active) {
                const int32_t* in = sh->seg[sg].t + i * 8;
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
        }
        LSYNC();
        LANES(l) {
            const int sg = l >> 4, i = l & 15;
            if (sg < NW && i < 8 && sh->seg[sg].c.active) {   // transposed: column k of row i goes to t[k * 8 + i]
                int32_t* t = sh->seg[sg].t + i;
                t[0] = L(o0); t[8] = L(o1); t[16] = L(o2); t[24] = L(o3); t[32] = L(o4); t[40] = L(o5); t[48] = L(o6); t[56] = L(o7);
            }
        }
        LSYNC();
        LANES(l) {
            const int sg = l >> 4, i = l & 15;
            if (sg < NW && i < 8 && sh->seg[sg].c.active) {
                const int32_t* in = sh->seg[sg].t + i * 8;   // column i, rows 0..7
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
                st4(reinterpret_cast<uint32_t*>(sh->seg[sg].pix + i * 8), U4{p0, p1, p2, p3});
            }
        }
        LSYNC();
        // 16 edge estimates on the 16 lanes of the row: lanes 0..7 against the left neighbour, 8..15 against the one above
        LV(int, emax); LV(int, emin); LV(int, ev_);
        LANES(l) {
            const int sg = l >> 4, i = l & 15;
            int ev = 0, have = 0;
            if (sg < NW && sh->seg[sg].c.active) {
                Seg5& S = sh->seg[sg];
                const int cur = S.c.cur;
                if (i < 8 && S.c.has_left) { have = 1; ev = (int16_t)(S.nsb[cur ^ 1].vert[i] - Dec4Wave::half16(LEP_PIX(S, i, 0) - LEP_PIX(S, i, 1)) - (LEP_PIX(S, i, 0) + 1024)); }
                if (i >= 8 && S.c.has_above) { const int k = i - 8; have = 1; ev = (int16_t)(S.nsa.horiz[k] - Dec4Wave::half16(LEP_PIX(S, 0, k) - LEP_PIX(S, 1, k)) - (LEP_PIX(S, 0, k) + 1024)); }
            }
            L(emax) = have ? ev : -0x7fffffff;
            L(emin) = have ? -ev : -0x7fffffff;
            L(ev_) = ev;
        }
        LV(int, smax); LV(int, smin); LV(int, ssum); LV(int, slo);
        lepwave::row_incl_max(emax, smax); lepwave::row_incl_max(emin, smin); lepwave::row_incl_sum(ev_, ssum);
        lepwave::row_shr8(ssum, slo);   // lane 15: the sum of lanes 0..7
        LANES(l) {
            const int sg = l >> 4, i = l & 15;
            if (sg < NW && i == 15 && sh->seg[sg].c.active) {
                Seg5& S = sh->seg[sg];
                const bool has_left = S.c.has_left != 0, has_above = S.c.has_above != 0;
                const int mx = L(smax), mn = -L(smin), sumL = L(slo), sumA = L(ssum) - L(slo);
                int32_t avgmed = 0, unc = 0, unc2 = 0;
                if (has_left || has_above) {
                    int sum0 = has_left ? sumL : sumA, sum1 = (has_left && has_above) ? sumA : sum0;
                    avgmed = (sum0 + sum1) >> 1;
                    unc = (mx - mn) >> 3;
                    sum0 -= avgmed; sum1 -= avgmed;
                    unc2 = (iabs(sum0) < iabs(sum1) ? sum0 : sum1) >> 3;
                }
                S.c.pred = (avgmed / (int)S.q[0] + 4) >> 3;
                S.c.a = imin(bitlen((uint32_t)iabs(unc) & 0xffff), 11);
                S.c.b17 = imin(bitlen((uint32_t)iabs(unc2) & 0xffff), 16);
                S.c.sctx = unc2 >= 0 ? (unc2 == 0 ? 3 : 2) : 1;
            }
        }
        LSYNC();
    }
    // neighbour summary of the finished block (block_context.hh:44-78); the block and its summary to HBM.  Done by the segment's own
    // wavefront right after its DC round: the walk may start a row next whose first blocks have the block just finished above them
    // (rows of one or two blocks), and what it loads must be there.
    WDEV void publish_store(int w) {
        LEP_MARK("publish");
        Wave5& W = LEP5_WV(w);
        Seg5& S = sh->seg[w];
        const int cur = W.x & 1;
        const int16_t* here = S.blk[cur];
        NSum& nh = S.nsb[cur];
        const int nz = (int)uni((uint32_t)S.c.nz);
        LANES(l) {
            if (l < 16) {
                const int k = l & 7;
                const int dcq = here[49] * (int)S.q[0];
                if (l < 8) nh.horiz[k] = (int16_t)(dcq + LEP_PIX(S, 7, k) + 1024 + Dec4Wave::half16(LEP_PIX(S, 7, k) - LEP_PIX(S, 6, k)));
                else nh.vert[k] = (int16_t)(dcq + LEP_PIX(S, k, 7) + 1024 + Dec4Wave::half16(LEP_PIX(S, k, 7) - LEP_PIX(S, k, 6)));
            }
            if (l == 16) nh.nz = nz;
        }
        LSYNC();
        LANES(l) {
            W.row[(int64_t)W.x * 64 + l] = here[l];
            if (l < (int)(sizeof(NSum) / 4)) ((uint32_t*)&W.nrow[W.x])[l] = ((const uint32_t*)&nh)[l];
        }
    }

    WDEV void init_tables() {
        LANES(l) {
            sh->r2a[l] = kR2A[l]; sh->a2r[l] = kA2R[l];
            if (l < 32) {
                int j = 0, c = l;
                while (j < 6 && c >= 7 - j) { c -= 7 - j; ++j; }
                sh->cj[l] = (uint8_t)(l < 28 ? j : 7); sh->cn[l] = (uint8_t)(l < 28 ? c + 1 : 0);
            }
            if (l < 14) {
                const bool hz = l < 7;
                const int j = hz ? l : l - 7, coord = hz ? j + 1 : (j + 1) * 8, step = hz ? 8 : 1;
                for (int k = 0; k < 8; ++k) sh->tap[l][k] = kR2A[coord + k * step];
            }
        }
    }
    WDEV void init_segment(int w) {
        Seg5& S = sh->seg[w];
        LANES(l) {
            for (int d = l; d < kSignWords; d += 64) S.sign[d] = kBranchInit;
            for (int d = l; d < kResDcWords; d += 64) S.resdc[d] = kBranchInit;
            if (l < (int)(sizeof(NSum) / 4)) { ((uint32_t*)&S.nsb[0])[l] = 0; ((uint32_t*)&S.nsb[1])[l] = 0; ((uint32_t*)&S.nsa)[l] = 0; }
            if (l == 0) { S.c.active = 0; S.c.qn = 0; S.c.cur = 0; S.c.nz = 0; S.c.badmask = 0; }
        }
    }

    // the whole group.  present[w]: the wavefront has a segment (the last workgroup of a launch may be short)
    WDEV void run(Dec5Shared<NW>* shared) {
        sh = shared;
        init_tables();
        LEP5_WAVES(w) {
            Wave5& W = LEP5_WV(w);
            init_segment(w);
            LSYNC();
            W.nbins = 0; W.rc = 0; W.idx = 0; W.not_top = 0;
            if (W.img) {
                W.bc.init_stream(W.stream, W.stream_len);
                if (next_row(w)) begin_block(w);
            }
        }
        LEP5_GSYNC(NW);
        if (LEP5_ROLE(NW, 0, 0)) p_prologue();
        LEP5_GSYNC(NW);
        for (uint32_t step = 0;; ++step) {
            int any = 0;
            for (int q = 0; q < NW; ++q) any |= (int)uni((uint32_t)sh->seg[q].c.active);
            if (!any) break;
            LEP5_WAVES(w) {   // S1: the count of interior non-zeros, the interior
                Seg5& S = sh->seg[w];
                if (uni((uint32_t)S.c.active)) {
                    const int nz = round_nz(w);
                    if (nz > 49) fail(w, 7);
                    else {
                        int zz = 0, left = nz;
#pragma nounroll
                        while (zz < 49 && left > 0) round_77(w, zz, left);
                        LANES(l) if (l == 0) S.c.nz = nz;
                        LSYNC();
                    }
                }
            }
            LEP5_GSYNC(NW);
            if (LEP5_ROLE(NW, step, 1)) p_lakhani();
            LEP5_GSYNC(NW);
            LEP5_WAVES(w) {   // S2: the edges
                if (uni((uint32_t)sh->seg[w].c.active)) { const int rc = round_edges(w); if (rc) fail(w, rc); }
            }
            LEP5_GSYNC(NW);
            if (LEP5_ROLE(NW, step, 2)) p_idct_dcpred();
            LEP5_GSYNC(NW);
            LEP5_WAVES(w) {   // S3: DC, then on to the next block
                Seg5& S = sh->seg[w];
                if (uni((uint32_t)S.c.active)) {
                    round_dc(w);
                    publish_store(w);
                    finish_block(w);
                    LSYNC();
                }
            }
            LEP5_GSYNC(NW);
            if (NW == 1) { update_pass(1u); p_prologue(); }
            else {
                if (LEP5_ROLE(NW, step, 3)) p_prologue();
                if (LEP5_ROLE(NW, step, 0)) update_pass((1u << NW) - 1u);
            }
            LEP5_GSYNC(NW);
        }
    }
};

}  // namespace lep5d
