// lep_enc3.h -- "v3" encoder: the wave-cooperative block coder of lep_enc2.h re-cut for 8 wavefronts per SIMD.
//
// Measured on MI355X (profiles/r01c_*): the v2 encoder ran 15 workgroups per CU (10.4 KB of LDS each), spent its
// serial phase as exec-masked lane-0 code (about 25 VALU + 12 SALU instructions per bin, the SALU ones only
// juggling exec masks) and waited on HBM for every block's coefficients.  A micro-benchmark of the bool-coder
// recurrence (profiles/r01_issue_microbench.txt) showed where serial work belongs on this chip: written as
// "uniform vector" code -- every lane computes the same value, branches are taken on ballots -- it runs 1.85x
// faster at 8 waves/SIMD than the same recurrence on the scalar unit, and much faster than exec-masked code.
// So v3 keeps v2's phases (same reference citations) and changes how they are executed:
//   * 4.6 KB of LDS per wavefront (bin list of 512 entries filled lane-range by lane-range, no reciprocal table:
//     Branch probabilities come from lep3::prob_of) and __launch_bounds__(64, 8) -> 32 wavefronts per CU;
//   * P4, the bool coder (boolwriter.hh:48-118), is uniform vector code: bins are read 64 at a time into a
//     register, handed out by v_readlane, the coder state lives in VGPRs, branches are uniform;
//   * the next block's coefficients / neighbour summary are fetched one block ahead;
//   * LDS hand-offs inside the wavefront use compiler-only ordering (LSYNC), not s_barrier.
// Model layout in HBM = lep_core.h's (one word per Branch, 11-word exponent rows); results are bit-identical.
#pragma once
#include "lep_v3.h"

namespace lep3 {

#if LEP_ON_GPU && defined(LEP_MARKS)
#define LEP_EMARK(name) __asm__ volatile("; MARK " name)
#else
#define LEP_EMARK(name) ((void)0)
#endif

constexpr int kBinChunk = 512;
constexpr int kMaxDup3 = 160;                  // 14 * 10 threshold bins
constexpr uint32_t kResident3 = 1u << 30;      // bin whose Branch lives in LDS (sign table); resolved by the coder loop

struct Enc3Shared {
    alignas(16) uint32_t bins[kBinChunk];   // P2: branch index | bit << 31;  after P3: probability | bit << 31
    uint16_t dup[kMaxDup3];     // positions (into bins) of bins whose Branch may repeat within the block
    uint32_t sign[96];          // the sign Branches live in LDS for the whole segment (never written back)
    alignas(16) int32_t t[64];              // IDCT intermediate
    int32_t icos_x[64], icos_y[64];
    int16_t here[64], left[64], above[64], aleft[64];   // aligned order
    alignas(16) int16_t pix[64];
    uint16_t q[64];
    uint8_t thr[64];
    uint8_t r2a[64], a2r[64], nzbin[64];
    NSum ns_left, ns_above, ns_here;
};

// Two wavefronts per thread segment (lep_encode_v3x2_kernel: launches too small to fill the chip with one wavefront per
// segment -- a single image, a serving daemon's trickle).  In the encode direction nothing upstream of the bool coder
// depends on it, so the block coder splits into a PRODUCER wavefront (P0-P3: staging, IDCT, contexts, bin list, model
// round trip -> resolved (bit, probability) pairs in LDS) and a CONSUMER wavefront (P4: the bool coder and the sign
// Branches), one bin-list chunk apart: while the consumer codes chunk i from one buffer the producer fills the other
// with chunk i+1; one workgroup barrier per chunk hands a buffer over in each direction.
struct Enc3Pipe {
    alignas(16) uint32_t bins_b[kBinChunk];   // second bin-list buffer (the first is Enc3Shared::bins)
    int count[2];                             // bins in buffer k; -1 = the producer is done (or gave up)
    uint32_t out_len;                         // consumer -> producer at the very end
    int out_overflow;
};

// on the GPU: true when the (wave-uniform) condition holds; written as a ballot so that a value the compiler cannot
// prove uniform still yields a scalar branch instead of exec-mask control flow
WDEV bool ucond(bool c) {
#if LEP_ON_GPU
    return __builtin_amdgcn_ballot_w64(c) != 0;
#else
    return c;
#endif
}

// bool writer (boolwriter.hh:48-118, boolwriter.cc:17-35) as uniform vector code: low / range / count are identical in
// every lane (VGPRs), the byte position is scalar, stores are done by lane 0
struct BoolEnc3 {
    uint32_t low, range;
    int count;
    uint8_t* out;
    uint32_t pos, cap;
    bool overflow;
    WDEV void emit(uint8_t b) {
#if LEP_ON_GPU
        if ((threadIdx.x & 63) == 0) out[pos] = b;
#else
        out[pos] = b;
#endif
    }
    WDEV void init_stream(uint8_t* o, uint32_t c) {
        out = o; cap = c; pos = 0; overflow = false;
        low = vec(0); range = vec(255); count = (int)vec((uint32_t)-24);
        put(0, 128);
    }
    WDEV void carry() {
        int x = (int)pos - 1;
        while (x >= 0 && uload8(out + x) == 0xff) {
#if LEP_ON_GPU
            if ((threadIdx.x & 63) == 0) out[x] = 0;
#else
            out[x] = 0;
#endif
            --x;
        }
        if (x >= 0) {
            const uint8_t v = (uint8_t)(uload8(out + x) + 1);
#if LEP_ON_GPU
            if ((threadIdx.x & 63) == 0) out[x] = v;
#else
            out[x] = v;
#endif
        }
    }
    static WDEV uint32_t uload8(const uint8_t* p) {
#if LEP_ON_GPU
        uintptr_t a = (uintptr_t)p;
        __asm__ volatile("" : "+v"(a));
        return uni(*reinterpret_cast<const uint8_t*>(a));
#else
        return *p;
#endif
    }
    WDEV void put(uint32_t bit, uint32_t prob) { put_m(0u - bit, prob); }
    WDEV void put_m(uint32_t m, uint32_t prob) {   // m = 0 / 0xffffffff: the bit as a mask (the bin list keeps the bit in bit 31: one arithmetic shift)
#if LEP_ON_GPU
        const uint32_t split = 1 + (__umul24(range - 1, prob) >> 8);
#else
        const uint32_t split = 1 + (((range - 1) * prob) >> 8);
#endif
        uint32_t l = low + (split & m);
        uint32_t r = split + ((range - 2 * split) & m);
        int shift = __builtin_clz(r) - 24;
        r <<= shift;
        int c = count + shift;
        if (ucond(c >= 0)) {
            const int offset = shift - c;
            if (pos + 2 > cap) overflow = true;
            if (!overflow) {
                if (ucond(((l << (offset - 1)) & 0x80000000u) != 0)) carry();
                emit((uint8_t)(l >> (24 - offset)));
                ++pos;
            }
            l <<= offset;
            shift = c;
            l &= 0xffffff;
            c -= 8;
        }
        l <<= shift;
        count = c; low = l; range = r;
    }
    WDEV uint32_t finish() {
        for (int i = 0; i < 32; ++i) put(0, 128);
        if (!overflow && pos && (uload8(out + pos - 1) & 0xe0) == 0xc0) { emit(0); ++pos; }
        return pos;
    }
};

// The same writer with low / range / count on the SCALAR unit (see BoolDec4S in lep_dec4.h for the why): code_chunk hands
// part of the bin groups to this form so that the scalar ALU, idle most of the time beside a VALU-bound coder loop, carries
// some of the recurrence.  Byte emission (position, carry ripple, stores by lane 0) is BoolEnc3's: load() / store() move the
// three state words across.
struct BoolEnc3S {
    uint32_t low, range;
    int count;
    WDEV void load(const BoolEnc3& b) { low = uni(b.low); range = uni(b.range); count = (int)uni((uint32_t)b.count); }
    WDEV void store(BoolEnc3& b) const { b.low = vec(low); b.range = vec(range); b.count = (int)vec((uint32_t)count); }
    WDEV void put(BoolEnc3& b, uint32_t bit, uint32_t prob) {
        const uint32_t split = 1 + (((range - 1) * prob) >> 8);
        uint32_t l = bit ? low + split : low;
        uint32_t r = bit ? range - split : split;
        int shift = __builtin_clz(r) - 24;
        r <<= shift;
        int c = count + shift;
        if (c >= 0) {
            const int offset = shift - c;
            if (b.pos + 2 > b.cap) b.overflow = true;
            if (!b.overflow) {
                if ((l << (offset - 1)) & 0x80000000u) b.carry();
                b.emit((uint8_t)(l >> (24 - offset)));
                ++b.pos;
            }
            l <<= offset;
            shift = c;
            l &= 0xffffff;
            c -= 8;
        }
        l <<= shift;
        count = c; low = l; range = r;
    }
};

// how the bins of a chunk alternate between the two forms: LEP_ENC3_SG groups of four bins on the scalar unit, then LEP_ENC3_VG
// groups on the vector ALU, and so on (the tail of a chunk stays on the vector ALU).  Measured, 1024 x 4K, MI355X
// (profiles/r02o_encoder_ab.json, r02t_knobs_ab.json, r02u_*): all vector 1186 ms; 1+3: 1107; 2+2: 1092 (1077 with the IDCT change);
// 3+1: 1159; all scalar: 1278; 4+4: 1060
#ifndef LEP_ENC3_SG
#define LEP_ENC3_SG 4
#endif
#ifndef LEP_ENC3_VG
#define LEP_ENC3_VG 4
#endif

struct Enc3Wave {
    const ImageDev* img;
    uint32_t* model;
    Enc3Shared* sh;
    int comp, ci;
    BoolEnc3 bc;
    uint32_t nbins;
    Enc3Pipe* pipe = nullptr;   // non-null: this wavefront is the producer half of a two-wave segment
    int pcur = 0;               // buffer the producer fills next
    WDEV uint32_t* bin_buffer(int k) const { return k ? pipe->bins_b : sh->bins; }
    static WDEV void pair_barrier() {
#if LEP_ON_GPU
        __syncthreads();
#endif
    }
    // P4 over one resolved chunk (consumer side; also the single-wave kernel's P4)
    WDEV void code_chunk(const uint32_t* B, int n) {
        // (entries are read back from LDS at a uniform address, four at a time, and everything derived from them stays
        // on the vector ALU: a SALU instruction costs about two VALU ones here, profiles/r01_issue_microbench.txt)
        int j = 0;
#if LEP_ENC3_SG > 0
#pragma nounroll
        for (; j + 4 * (LEP_ENC3_SG + LEP_ENC3_VG) <= n; j += 4 * (LEP_ENC3_SG + LEP_ENC3_VG)) {   // SG groups on the scalar unit, then VG on the vector ALU
            BoolEnc3S sc;
            sc.load(bc);
#pragma unroll   // (rolled: a third of the code, 1.5 % slower -- profiles/r02x_dpp_ab.json)
            for (int g = 0; g < LEP_ENC3_SG; ++g) {
                const U4 q = ld4(B + j + 4 * g);
                code_bin_s(sc, uni(q.x)); code_bin_s(sc, uni(q.y)); code_bin_s(sc, uni(q.z)); code_bin_s(sc, uni(q.w));
            }
            sc.store(bc);
#pragma unroll   // (rolled: a third of the code, 1.5 % slower -- profiles/r02x_dpp_ab.json)
            for (int g = LEP_ENC3_SG; g < LEP_ENC3_SG + LEP_ENC3_VG; ++g) {
                const U4 q = ld4(B + j + 4 * g);
                code_bin(vec(q.x)); code_bin(vec(q.y)); code_bin(vec(q.z)); code_bin(vec(q.w));
            }
        }
#endif
#pragma nounroll
        for (; j + 4 <= n; j += 4) {
            const U4 q = ld4(B + j);   // one 16-byte LDS read
            code_bin(vec(q.x)); code_bin(vec(q.y)); code_bin(vec(q.z)); code_bin(vec(q.w));
        }
#pragma nounroll
        for (; j < n; ++j) code_bin(vec(B[j]));
    }
    // producer: hand the chunk in buffer pcur over, continue in the other buffer
    WDEV void publish_chunk(int n) {
        LANES(l) if (l == 0) pipe->count[pcur] = n;
        LSYNC();
#if LEP_ON_GPU
        pair_barrier();
#else
        code_chunk(bin_buffer(pcur), n);   // lane-loop emulation: the consumer's step, run in place
#endif
        pcur ^= 1;
    }
    // consumer wavefront: codes chunk after chunk until the producer says -1; returns the stream length
    WDEV uint32_t consume(Enc3Shared* shared, Enc3Pipe* p, uint8_t* stream, uint32_t cap) {
        sh = shared; pipe = p;
        bc.init_stream(stream, cap);
        pair_barrier();   // tables (sign Branches) initialised by the producer
        for (int k = 0;; k ^= 1) {
            pair_barrier();
            const int n = (int)uni((uint32_t)p->count[k]);
            if (n < 0) break;
            code_chunk(bin_buffer(k), n);
        }
        const uint32_t len = bc.finish();
        LANES(l) if (l == 0) { p->out_len = len; p->out_overflow = bc.overflow ? 1 : 0; }
        LSYNC();
        pair_barrier();
        return len;
    }

    WDEV void init_tables() {
        LANES(l) {
            sh->r2a[l] = kR2A[l]; sh->a2r[l] = kA2R[l]; sh->nzbin[l] = l < 50 ? kNzBin[l] : 9;
            for (int d = l; d < 96; d += 64) sh->sign[d] = kBranchInit;
            if (l < (int)(sizeof(NSum) / 4)) { ((uint32_t*)&sh->ns_left)[l] = 0; ((uint32_t*)&sh->ns_above)[l] = 0; }
        }
        LSYNC();
    }
    WDEV void stage_component(int c) {
        comp = c; ci = c ? 1 : 0;
        LANES(l) {
            sh->q[l] = img->q[c][l]; sh->icos_x[l] = img->icos_x[c][l]; sh->icos_y[l] = img->icos_y[c][l];
            sh->thr[l] = img->min_thresh[c][l];
        }
        LSYNC();
    }

    // integer IDCT without DC (idct.cc:35-161): lep_v3.h idct_no_dc; S.pix is column-major, LEP_PIX(S, y, x)
    WDEV void idct_rows() { idct_no_dc(sh); }
    static WDEV int half16(int d) { return (int16_t)d / 2; }

    // Branch::record_obs_and_update (branch.hh:82-100) for a uniform-vector word and observation; rare paths on ballots
    static WDEV uint32_t bupd_uv(uint32_t w, uint32_t obs) {
        uint32_t f = (w & 255) + (obs ^ 1), t = ((w >> 8) & 255) + obs;
        if (ucond((f | t) > 255)) {   // the incremented count was 255
            const uint32_t f0 = w & 255, t0 = (w >> 8) & 255;
            if (ucond((obs ? f0 : t0) == 1)) return (w & 0xffff) | ((obs ? 0u : 255u) << 16);
            f = obs ? (1 + f0) >> 1 : 129u;
            t = obs ? 129u : (1 + t0) >> 1;
        }
        return f | (t << 8) | (prob_of(f, t) << 16);
    }
    // the same through the scalar-unit writer (e in an SGPR); the sign Branch's probability is still computed on the vector ALU
    WDEV void code_bin_s(BoolEnc3S& sc, uint32_t e) {
        if (e & kResident3) {
            const uint32_t slot = e & 127, bit = e >> 31;
            const uint32_t w = uni(sh->sign[slot]);
            sc.put(bc, bit, w >> 16);
            sh->sign[slot] = bupd_uv(vec(w), vec(bit));
        } else sc.put(bc, e >> 31, e & 255);
    }
    // one entry of the resolved bin list through the bool coder (uniform vector value e)
    WDEV void code_bin(uint32_t e) {
        if (ucond((e & kResident3) != 0)) {
            const uint32_t slot = e & 127, bit = e >> 31;
            const uint32_t w = sh->sign[slot];
            bc.put(bit, w >> 16);
            sh->sign[slot] = bupd_uv(w, bit);
        } else bc.put_m((uint32_t)((int32_t)e >> 31), e & 255);
    }

    // Encodes the block staged in sh->here (+ left / above / aleft when present).  Returns 0 or an exit code (uniform).
    WDEV int encode_block(bool has_left, bool has_above) {
        Enc3Shared& S = *sh;
        LEP_EMARK("e_ballots");
        LV(int, nzf); LV(int, tx); LV(int, ty);
        LV(int, cnt); LV(int, ndup); LV(int, off); LV(int, doff); LV(int, bad);
        LV(int, len_); LV(int, val_); LV(int, pos_); LV(int, nexp_); LV(int, coded_); LV(int, thr_); LV(int, isedge_);
        LV(uint32_t, expbase_); LV(uint32_t, signidx_); LV(uint32_t, resbase_); LV(uint32_t, thrbase_);

        LANES(l) {
            const int a = l < 49 ? l : (l == 63 ? 49 : l + 1);
            L(nzf) = S.here[a] != 0;
        }
        const uint64_t m = lepwave::wave_ballot(nzf);
        const uint64_t mask7 = m & ((1ull << 49) - 1);
        const uint32_t maskh = (uint32_t)(m >> 49) & 0x7f, maskv = (uint32_t)(m >> 56) & 0x7f;
        const int nz = lepwave::popc64(mask7), neh = __builtin_popcount(maskh), nev = __builtin_popcount(maskv);
        LANES(l) {
            int ex = 0, ey = 0;
            if (l < 49 && L(nzf)) { int coord = S.a2r[l]; ex = coord & 7; ey = coord >> 3; }
            L(tx) = ex; L(ty) = ey;
        }
        const int eob_x = lepwave::wave_max(tx), eob_y = lepwave::wave_max(ty);

        LEP_EMARK("e_idct");
        idct_rows();   // S.pix = IDCT of the block without its DC

        LEP_EMARK("e_p1");
        int nzctx = 0;
        if (has_left && has_above) nzctx = (S.ns_above.nz + S.ns_left.nz + 2) / 4;
        else if (has_above) nzctx = (S.ns_above.nz + 1) / 2;
        else if (has_left) nzctx = (S.ns_left.nz + 1) / 2;

        // DC prediction inputs (model.hh:674-784): the 16 edge estimates on 16 lanes, reduced with wave max / sums
        int32_t dc_avgmed = 0, dc_unc = 0, dc_unc2 = 0;
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
            if (has_left || has_above) {
                int sum0 = has_left ? sumL : sumA, sum1 = (has_left && has_above) ? sumA : sum0;
                dc_avgmed = (sum0 + sum1) >> 1;
                dc_unc = (mx - mn) >> 3;
                sum0 -= dc_avgmed; sum1 -= dc_avgmed;
                dc_unc2 = (iabs(sum0) < iabs(sum1) ? sum0 : sum1) >> 3;
            }
        }

        // ---- P1: per-lane analysis (lane = coefficient: 0..48 interior in zig-zag order, 49..55 / 56..62 edges, 63 DC)
        LANES(l) {
            const int a = l < 49 ? l : (l == 63 ? 49 : l + 1);
            int c = S.here[a];
            int v = c < 0 ? -c : c, len = bitlen((uint32_t)v), pos = c >= 0;
            int coded = 0, n = 0, nd = 0, nexp = 0, thr = 0, isedge = 0, err = 0;
            uint32_t expbase = 0, signidx = 0, resbase = 0, thrbase = 0;
            if (l < 49) {
                const int before = lepwave::popc64(mask7 & ((1ull << l) - 1));
                const int left_before = nz - before;
                coded = left_before > 0;
                if (coded) {
                    int prior;
                    if (has_left && has_above) prior = (uint16_t)((iabs(S.left[l]) + iabs(S.above[l])) * 13 + 6 * iabs(S.aleft[l])) >> 5;
                    else if (has_left) prior = (int16_t)iabs(S.left[l]);
                    else if (has_above) prior = (int16_t)iabs(S.above[l]);
                    else prior = 0;
                    const int nb = S.nzbin[left_before];
                    const int bsr = bitlen((uint32_t)imin(iabs(prior), 1023));
                    expbase = lepdev::kExp7 + ((((uint32_t)ci * 10 + nb) * 49 + l) * 12 + bsr) * 11;
                    signidx = (uint32_t)ci * 48;
                    resbase = lepdev::kRes + (((uint32_t)ci * 64 + S.a2r[l]) * 10 + nb) * 10;
                }
            } else if (l < 63) {
                const bool horizontal = l < 56;
                const int j = horizontal ? l - 49 : l - 56;
                const uint32_t mk = horizontal ? maskh : maskv;
                const int ne = horizontal ? neh : nev;
                const int ne_before = ne - __builtin_popcount(mk & ((1u << j) - 1));
                coded = ne_before > 0;
                isedge = 1;
                if (coded) {
                    const int coord = horizontal ? j + 1 : (j + 1) * 8;
                    int32_t prior = 0;
                    const bool nbr_ok = horizontal ? has_above : has_left;
                    if (nbr_ok) {
                        const int16_t* nbr = horizontal ? S.above : S.left;
                        const int32_t* icos = horizontal ? S.icos_x + coord * 8 : S.icos_y + coord;
                        const int step = horizontal ? 8 : 1;
                        if (icos[0] == 0) err = 43;
                        else {
                            uint32_t acc = (uint32_t)(int32_t)nbr[S.r2a[coord]] * (uint32_t)icos[0];
                            for (int i = 1; i < 8; ++i) {
                                int32_t xi = S.here[S.r2a[coord + i * step]], ai = nbr[S.r2a[coord + i * step]];
                                int32_t term = (i & 1) ? xi + ai : xi - ai;
                                acc -= (uint32_t)icos[i] * (uint32_t)term;
                            }
                            prior = (int32_t)acc / icos[0];
                        }
                    }
                    const uint32_t aprior = prior < 0 ? 0u - (uint32_t)prior : (uint32_t)prior;
                    const int bsr = bitlen(aprior > 1023 ? 1023 : aprior);
                    const int16_t p16 = (int16_t)prior;
                    const int sctx = p16 == 0 ? 0 : (p16 > 0 ? 1 : 2);
                    thr = S.thr[coord];
                    expbase = lepdev::kExpX + ((((uint32_t)ci * 10 + ne_before) * 15 + (horizontal ? j : j + 7)) * 12 + bsr) * 11;
                    signidx = ((uint32_t)ci * 4 + sctx) * 12 + bsr;
                    resbase = lepdev::kRes + (((uint32_t)ci * 64 + coord) * 10 + ne_before) * 10;
                    if (len > 1 && len - 2 >= thr) {
                        nd = len - 1 - thr;
                        thrbase = lepdev::kThresh + ((((uint32_t)ci * 256 + (uint32_t)imin((int)((aprior & 0xffff) >> thr), 255)) * 8) +
                                                     (uint32_t)imin(len - thr, 7)) * 128;
                    }
                }
            } else {   // DC
                coded = 1;
                const int32_t avgmed = dc_avgmed, unc = dc_unc, unc2 = dc_unc2;   // from the 16-lane pass before P1
                const int pred = (avgmed / (int)S.q[0] + 4) >> 3;
                const int ua = imin(bitlen((uint32_t)iabs(unc) & 0xffff), 11), ub = imin(bitlen((uint32_t)iabs(unc2) & 0xffff), 16);
                int d = c - pred;
                if (d < -1024) d += 2049;
                if (d > 1024) d -= 2049;
                int back = d + pred;
                if (back < -1024) back += 2049;
                if (back > 1024) back -= 2049;
                if (back != c) err = 6;
                v = iabs(d); len = bitlen((uint32_t)v & 0xffff); pos = d >= 0;
                expbase = lepdev::kExpDc + ((uint32_t)ua * 17 + ub) * 11;
                signidx = (uint32_t)ci * 48 + (unc2 >= 0 ? (unc2 == 0 ? 3 : 2) : 1);
                resbase = lepdev::kResDc + (uint32_t)ua * 10;
            }
            if (coded) {
                if (len > 11) err = 6;
                nexp = len < 11 ? len + 1 : 11;
                n = nexp + (len ? 1 : 0) + (len > 1 ? len - 1 : 0);
            } else nd = 0;
            if (l == 0) n += 6;
            if (l == 49 || l == 56) n += 3;
            L(cnt) = n; L(ndup) = nd; L(bad) = err;
            L(len_) = len; L(val_) = v; L(pos_) = pos; L(nexp_) = nexp; L(coded_) = coded; L(thr_) = thr; L(isedge_) = isedge;
            L(expbase_) = expbase; L(signidx_) = signidx; L(resbase_) = resbase; L(thrbase_) = thrbase;
        }
        LEP_EMARK("e_scan");
        const uint64_t badmask = lepwave::wave_ballot(bad);
        if (badmask) {   // report what the serial coder would have hit first (lane order = stream order)
            LV(int, bad43);
            LANES(l) L(bad43) = L(bad) == 43;
            const uint64_t m43 = lepwave::wave_ballot(bad43);
            return ((m43 >> __builtin_ctzll(badmask)) & 1) ? 43 : 6;
        }
        const int N = lepwave::wave_excl_scan(cnt, off);
        lepwave::wave_excl_scan(ndup, doff);

        // The bin list is produced, resolved and coded lane-range by lane-range so that it never holds more than kBinChunk
        // entries (a lane emits at most 28 bins; ordinary blocks are one range of all 64 lanes).
        LEP_EMARK("e_chunk");
        int lane0 = 0;
        while (lane0 < 64) {
            // largest lane range [lane0, lane1) whose bins fit
            LV(int, fits);
            const int base = (int)lepwave::wave_read((const uint32_t*)off, lane0);
            LANES(l) L(fits) = l >= lane0 && L(off) + L(cnt) - base <= kBinChunk;
            const uint64_t fm = lepwave::wave_ballot(fits) >> lane0;
            const int lane1 = lane0 + (fm == ~0ull >> lane0 ? 64 - lane0 : __builtin_ctzll(~fm));
            const int n = (lane1 < 64 ? (int)lepwave::wave_read((const uint32_t*)off, lane1) : N) - base;
            const int dbase = (int)lepwave::wave_read((const uint32_t*)doff, lane0);
            const int D = (lane1 < 64 ? (int)lepwave::wave_read((const uint32_t*)doff, lane1) : dbase + 0x7fffffff) ;
            uint32_t* const B = pipe ? bin_buffer(pcur) : S.bins;   // the bin list of this chunk
            // ---- P2: bin emission ---------------------------------------------------------------------
            LEP_EMARK("e_p2");
        LV(int, dcount);
            LANES(l) {
                int dj = L(doff) - dbase;
                if (l >= lane0 && l < lane1) {
                    int j = L(off) - base;
                    if (l == 0) {
                        const uint32_t T = lepdev::kNz7x7 + ((uint32_t)ci * 26 + S.nzbin[nzctx]) * 192;
                        int so_far = 0;
                        for (int i = 5; i >= 0; --i) { int b = (nz >> i) & 1; B[j++] = (T + i * 32 + so_far) | ((uint32_t)b << 31); so_far = (so_far << 1) | b; }
                    }
                    if (l == 49 || l == 56) {
                        const bool horizontal = l == 49;
                        const uint32_t T = (horizontal ? lepdev::kNz8x1 : lepdev::kNz1x8) + (((uint32_t)ci * 8 + (horizontal ? eob_x : eob_y)) * 8 + (nz + 3) / 7) * 12;
                        const int ne = horizontal ? neh : nev;
                        int so_far = 0;
                        for (int i = 2; i >= 0; --i) { int b = (ne >> i) & 1; B[j++] = (T + i * 4 + so_far) | ((uint32_t)b << 31); so_far = (so_far << 1) | b; }
                    }
                    if (L(coded_)) {
                        const int len = L(len_), v = L(val_), nexp = L(nexp_);
                        for (int i = 0; i < nexp; ++i) B[j++] = (L(expbase_) + i) | ((uint32_t)(len != i) << 31);
                        if (len) B[j++] = L(signidx_) | kResident3 | ((uint32_t)L(pos_) << 31);
                        if (len > 1) {
                            int b = len - 2;
                            if (L(isedge_) && b >= L(thr_)) {
                                int s = 1;
                                for (; b >= L(thr_); --b) {
                                    int bit = (v >> b) & 1;
                                    S.dup[dj++] = (uint16_t)j;
                                    B[j++] = (L(thrbase_) + s) | ((uint32_t)bit << 31);
                                    s = imin((s << 1) | bit, 127);
                                }
                            }
                            for (; b >= 0; --b) B[j++] = (L(resbase_) + b) | ((uint32_t)((v >> b) & 1) << 31);
                        }
                    }
                }
                L(dcount) = (l >= lane0 && l < lane1) ? L(ndup) : 0;
            }
            (void)D;
            LEP_EMARK("e_p3a");
        LV(int, dtmp);
            const int Dn = lepwave::wave_sum(dcount);   // threshold bins in this range
            LSYNC();

            // ---- P3: model words of all bins of the range in ONE HBM round trip ------------------------------
            // threshold bins (their Branch can repeat inside a block: in-order forwarding below) are requested first, then the
            // bins with a block-unique Branch, two rounds of loads in flight before the first use
            LV(uint32_t, didx); LV(uint32_t, dw); LV(uint32_t, dbit); LV(int, djpos); LV(int, dlast);
            const int nn0 = Dn < 64 ? Dn : 64;
            LANES(l) {
                uint32_t idx = 0xffffffffu, w = 0, bit = 0;
                int j = -1;
                if (l < nn0) {
                    j = S.dup[l];
                    const uint32_t e = B[j];
                    idx = e & 0x3fffffffu; bit = e >> 31;
                    w = model[idx];
                }
                L(didx) = idx; L(dw) = w; L(dbit) = bit; L(djpos) = j; L(dlast) = 1;
            }
            // ---- P3a: bins with a block-unique Branch: parallel load / adapt / store ------------------
            for (int b0 = 0; b0 < n; b0 += 128) {
                LV(uint32_t, w0); LV(uint32_t, w1);
                LANES(l) {
                    const int j0 = b0 + l, j1 = b0 + 64 + l;
                    uint32_t a = 0, b = 0;
                    if (j0 < n) { const uint32_t e = B[j0]; if (!(e & kResident3) && (e & 0x3fffffffu) < lepdev::kThresh) a = model[e & 0x3fffffffu]; }
                    if (j1 < n) { const uint32_t e = B[j1]; if (!(e & kResident3) && (e & 0x3fffffffu) < lepdev::kThresh) b = model[e & 0x3fffffffu]; }
                    L(w0) = a; L(w1) = b;
                }
                LANES(l) {
                    for (int h = 0; h < 2; ++h) {
                        const int j = b0 + h * 64 + l;
                        if (j < n) {
                            const uint32_t e = B[j], idx = e & 0x3fffffffu;
                            if (!(e & kResident3) && idx < lepdev::kThresh) {
                                const uint32_t w = h ? L(w1) : L(w0);
                                const int bit = (int)(e >> 31);
                                model[idx] = bupd(w, bit);
                                B[j] = (w >> 16) | ((uint32_t)bit << 31);
                            }
                        }
                    }
                }
            }
            // ---- P3b: threshold bins ---------------------------------------------------------------------------
            LEP_EMARK("e_p3b");
            for (int cb = 0; cb < Dn; cb += 64) {
                const int nn = Dn - cb < 64 ? Dn - cb : 64;
                if (cb) {   // further groups of 64 (blocks with very large edge coefficients)
                    LANES(l) {
                        uint32_t idx = 0xffffffffu, w = 0, bit = 0;
                        int j = -1;
                        if (l < nn) {
                            j = S.dup[cb + l];
                            const uint32_t e = B[j];
                            idx = e & 0x3fffffffu; bit = e >> 31;
                            w = model[idx];
                        }
                        L(didx) = idx; L(dw) = w; L(dbit) = bit; L(djpos) = j; L(dlast) = 1;
                    }
                }
                // do two of them share a Branch?  (almost never: then every lane adapts its own word, no forwarding)
                LV(int, conf);
                LANES(l) L(conf) = 0;
                for (int r = 0; r + 1 < nn; ++r) {
                    const uint32_t ridx = lepwave::wave_read(didx, r);
                    LANES(l) if (l > r && l < nn && L(didx) == ridx) L(conf) = 1;
                }
                if (!lepwave::wave_ballot(conf)) {
                    LANES(l) if (l < nn) {
                        B[L(djpos)] = (L(dw) >> 16) | (L(dbit) << 31);
                        model[L(didx)] = bupd(L(dw), (int)L(dbit));
                    }
                } else {
                    for (int r = 0; r < nn; ++r) {
                        const uint32_t ridx = lepwave::wave_read(didx, r), rw = lepwave::wave_read(dw, r), rbit = lepwave::wave_read(dbit, r);
                        const uint32_t nw = bupd_s(rw, (int)rbit);
                        LANES(l) {
                            if (l == r) { B[L(djpos)] = (rw >> 16) | (rbit << 31); L(dw) = nw; }
                            else if (L(didx) == ridx) { if (l > r) L(dw) = nw; else L(dlast) = 0; }
                        }
                    }
                    LANES(l) if (l < nn && L(dlast)) model[L(didx)] = L(dw);
                }
                LSYNC();
            }
            LSYNC();

            // ---- P4: bool coder over the resolved (bit, probability) pairs: uniform vector code ------------
            LEP_EMARK("e_p4");
            if (pipe) publish_chunk(n);   // two-wave segment: the consumer wavefront codes it while this one goes on
            else code_chunk(B, n);
            LSYNC();
            lane0 = lane1;
        }
        LEP_EMARK("e_p5");
        nbins += (uint32_t)N;

        // ---- P5: neighbour summary of this block ----------------------------------------------------
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

    // whole segment; ns = this segment's NSum area (zeroed); returns exit code
    // p != nullptr: this wavefront is the producer of a two-wave segment (the consumer runs consume() with the same p)
    WDEV int run(const ImageDev* image, const SegDev& seg, uint32_t* model_words, NSum* ns, Enc3Shared* shared, uint8_t* stream,
                 uint32_t cap, Enc3Pipe* p = nullptr) {
        img = image; model = model_words; sh = shared; nbins = 0; pipe = p; pcur = 0;
        init_tables();
#if LEP_ON_GPU
        if (pipe) pair_barrier(); else bc.init_stream(stream, cap);
#else
        bc.init_stream(stream, cap);   // lane-loop emulation: this instance also plays the consumer (publish_chunk)
#endif
        const int rc = run_rows(seg, ns);
        if (pipe) {   // tell the consumer to finish, then wait for its verdict
            LANES(l) if (l == 0) pipe->count[pcur] = -1;
            LSYNC();
            pair_barrier();
            pair_barrier();
        }
        return rc;
    }
    WDEV int run_rows(const SegDev& seg, NSum* ns) {
        const ImageDev* image = img;
        bool top[3] = {true, true, true};
        for (uint32_t idx = 0;; ++idx) {
            RowSpec r = row_spec(image, idx);
            if (r.done) break;
            if (r.luma_y >= seg.y1 && !seg.is_last) break;
            if (r.skip) continue;
            if (r.luma_y < seg.y0) continue;
            stage_component(r.component);
            const int w = img->width[comp], yb = r.curr_y;
            const int16_t* row = img->blocks[comp] + (int64_t)yb * w * 64;
            const bool has_above = !top[comp];
            const int16_t* arow = has_above ? row - (int64_t)w * 64 : nullptr;
            NSum* nrow = ns + img->ns_offset[comp] + (yb & 1) * w;
            const NSum* narow = ns + img->ns_offset[comp] + ((yb & 1) ^ 1) * w;
            top[comp] = false;
            LV(int16_t, nxt_here); LV(int16_t, nxt_above); LV(uint32_t, nxt_ns);
            LANES(l) {   // block 0; later blocks are fetched one block ahead
                L(nxt_here) = row[l];
                L(nxt_above) = has_above ? arow[l] : (int16_t)0;
                L(nxt_ns) = (has_above && l < (int)(sizeof(NSum) / 4)) ? ((const uint32_t*)&narow[0])[l] : 0u;
            }
            for (int x = 0; x < w; ++x) {
                // P0: stage blocks.  left / above-left come from the previous block's LDS copies.
                LEP_EMARK("e_staging");
        LANES(l) {
                    if (x) { sh->left[l] = sh->here[l]; sh->aleft[l] = sh->above[l]; }
                    if (l < (int)(sizeof(NSum) / 4)) {
                        if (x) ((uint32_t*)&sh->ns_left)[l] = ((const uint32_t*)&sh->ns_here)[l];
                        if (has_above) ((uint32_t*)&sh->ns_above)[l] = L(nxt_ns);
                    }
                }
                LSYNC();
                LANES(l) {
                    sh->here[l] = L(nxt_here);
                    if (has_above) sh->above[l] = L(nxt_above);
                    if (x + 1 < w) {
                        L(nxt_here) = row[(int64_t)(x + 1) * 64 + l];
                        if (has_above) {
                            L(nxt_above) = arow[(int64_t)(x + 1) * 64 + l];
                            if (l < (int)(sizeof(NSum) / 4)) L(nxt_ns) = ((const uint32_t*)&narow[x + 1])[l];
                        }
                    }
                }
                LSYNC();
                int rc = encode_block(x > 0, has_above);
                if (rc) return rc;
                LEP_EMARK("e_store");
        LANES(l) if (l < (int)(sizeof(NSum) / 4)) ((uint32_t*)&nrow[x])[l] = ((const uint32_t*)&sh->ns_here)[l];
                if (x + 1 < w && yb * w + x + 1 >= img->coded_blocks[comp]) break;
            }
        }
        return 0;
    }
};

}  // namespace lep3
