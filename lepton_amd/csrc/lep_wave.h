// lep_wave.h -- tiny SPMD layer so that wave-cooperative kernel code is written once and
//   * compiles for gfx950 with hipcc (one wavefront = 64 lanes, cross-lane ops are ballot / DPP shuffles),
//   * compiles with g++ as a lane-loop emulation (tests/emu) so kernel logic is checked without a GPU.
// Rules the kernel code follows: per-lane state lives in LV() variables; code that touches it sits inside
// LANES(l){...}; cross-lane primitives (wave_*) are called only BETWEEN LANES regions, in wave-uniform
// control flow; data handed between lanes through shared memory is separated by WSYNC().
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)   // both hipcc passes: these are __device__ functions, never called from host code
#define LEP_ON_GPU 1
#define LEP_NL 1
#define LEP_LI(l) 0
// The lane index is made opaque at every use (an empty asm the optimiser cannot see through): everything derived from it alone
// -- LDS addresses, tap offsets, masks -- is otherwise a loop invariant of the block loop, gets hoisted out of it and, in kernels
// pinned at 64 VGPRs, spilled: every block then reloaded ~20 such values from scratch (a memory instruction each) instead of
// redoing a vector add.  With it the coder kernels' scratch drops from 164 / 196 to 8 / 16 bytes per lane and they run 4 % / 3 %
// faster (profiles/r03b_lane_index_ab.json).
static __device__ __forceinline__ int lep_lane_now() { int v = (int)(threadIdx.x & 63); __asm__ volatile("" : "+v"(v)); return v; }
#define LANES(l) for (int l = lep_lane_now(), lep_once_ = 1; lep_once_; lep_once_ = 0)
#define WSYNC() __syncthreads()
// One wavefront per workgroup: lanes run in lockstep and a wave's LDS / global accesses are performed in program
// order, so handing data between lanes needs no s_barrier and no counter drain -- only that the COMPILER keeps the
// accesses in order.  A wavefront-scope fence + wave_barrier does exactly that (no instructions are emitted).
#define LSYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define WDEV __device__ __forceinline__
#else
#define LEP_ON_GPU 0
#define LEP_NL 64
#define LEP_LI(l) (l)
// every lane is an independent program instance: the lane index is laundered through an empty asm so that the host
// compiler cannot carry value-range facts from one lane's path into another's
static inline int lep_lane_id(int i) { __asm__ volatile("" : "+r"(i)); return i; }
#define LANES(l) for (int lep_i_ = 0; lep_i_ < 64; ++lep_i_) for (int l = lep_lane_id(lep_i_), lep_once_ = 1; lep_once_; lep_once_ = 0)
#define WSYNC() ((void)0)
#define LSYNC() ((void)0)
#define WDEV inline
#endif

#define LV(T, name) T name[LEP_NL]
#define L(name) name[LEP_LI(l)]

namespace lepwave {

WDEV uint64_t wave_ballot(const int* pred) {
#if LEP_ON_GPU
    return __ballot(pred[0] != 0);
#else
    uint64_t m = 0;
    for (int i = 0; i < 64; ++i) m |= (uint64_t)(pred[i] != 0) << i;
    return m;
#endif
}

// Cross-lane reductions and scans on the DPP path of the vector ALU: 4 row_shr steps inside the 16-lane rows, row_bcast15 /
// row_bcast31 across them -- six v_add / v_max instructions.  (The __shfl forms these replace are ds_bpermute_b32 underneath:
// an LDS instruction per step, ~6 cycles per CU each against ~1 for a DPP op, profiles/r02n_inst_rates.txt; the coder kernels
// run 6 (decoder) to 9 (encoder) of these per 8x8 block.)
#if LEP_ON_GPU
template <int CTRL, int ROW_MASK>
WDEV int dpp_or(int keep, int v) { return __builtin_amdgcn_update_dpp(keep, v, CTRL, ROW_MASK, 0xf, false); }   // lanes without a source get `keep`
WDEV int wave_incl_sum(int v) {
    v += dpp_or<0x111, 0xf>(0, v);   // row_shr:1
    v += dpp_or<0x112, 0xf>(0, v);   // row_shr:2
    v += dpp_or<0x114, 0xf>(0, v);   // row_shr:4
    v += dpp_or<0x118, 0xf>(0, v);   // row_shr:8
    v += dpp_or<0x142, 0xa>(0, v);   // row_bcast15 into rows 1 and 3
    v += dpp_or<0x143, 0xc>(0, v);   // row_bcast31 into rows 2 and 3
    return v;
}
WDEV int wave_incl_max(int v) {
    const int lo = (int)0x80000000;
    int t;
    t = dpp_or<0x111, 0xf>(lo, v); v = t > v ? t : v;
    t = dpp_or<0x112, 0xf>(lo, v); v = t > v ? t : v;
    t = dpp_or<0x114, 0xf>(lo, v); v = t > v ? t : v;
    t = dpp_or<0x118, 0xf>(lo, v); v = t > v ? t : v;
    t = dpp_or<0x142, 0xa>(lo, v); v = t > v ? t : v;
    t = dpp_or<0x143, 0xc>(lo, v); v = t > v ? t : v;
    return v;
}
#endif

// exclusive prefix sum over lanes; returns the total
WDEV int wave_excl_scan(const int* in, int* out) {
#if LEP_ON_GPU
    const int v = in[0], s = wave_incl_sum(v);
    out[0] = s - v;
    return __builtin_amdgcn_readlane(s, 63);
#else
    int s = 0;
    for (int i = 0; i < 64; ++i) { int v = in[i]; out[i] = s; s += v; }
    return s;
#endif
}
// sum over all lanes
WDEV int wave_sum(const int* in) {
#if LEP_ON_GPU
    return __builtin_amdgcn_readlane(wave_incl_sum(in[0]), 63);
#else
    int s = 0;
    for (int i = 0; i < 64; ++i) s += in[i];
    return s;
#endif
}

WDEV int wave_max(const int* in) {
#if LEP_ON_GPU
    return __builtin_amdgcn_readlane(wave_incl_max(in[0]), 63);
#else
    int m = in[0];
    for (int i = 1; i < 64; ++i) m = in[i] > m ? in[i] : m;
    return m;
#endif
}

// inclusive prefix maximum (lane l: the largest of lanes 0 .. l)
WDEV void wave_prefix_max(const int* in, int* out) {
#if LEP_ON_GPU
    out[0] = wave_incl_max(in[0]);
#else
    int m = in[0];
    for (int i = 0; i < 64; ++i) { m = in[i] > m ? in[i] : m; out[i] = m; }
#endif
}
// inclusive suffix minimum (lane l: the smallest of lanes l .. 63): the lanes reversed (one ds_bpermute each way), the prefix
// maximum of the negated values in between
WDEV void wave_suffix_min(const int* in, int* out) {
#if LEP_ON_GPU
    const int rev = (63 - (int)(threadIdx.x & 63)) * 4;
    const int r = __builtin_amdgcn_ds_bpermute(rev, -in[0]);
    out[0] = -__builtin_amdgcn_ds_bpermute(rev, wave_incl_max(r));
#else
    int m = in[63];
    for (int i = 63; i >= 0; --i) { m = in[i] < m ? in[i] : m; out[i] = m; }
#endif
}

// value of lane `src` (wave-uniform src)
WDEV uint32_t wave_read(const uint32_t* v, int src) {
#if LEP_ON_GPU
    return (uint32_t)__builtin_amdgcn_readlane((int)v[0], src);
#else
    return v[src];
#endif
}

// lane `dst` (wave-uniform) of v takes `value` (wave-uniform).  (A compare and a select: this compiler has no builtin for
// v_writelane_b32, and on gfx9 the instruction may name one scalar register only -- lane and value would have to share it or go through m0.)
WDEV void wave_write(uint32_t* v, int dst, uint32_t value) {
#if LEP_ON_GPU
    v[0] = (int)(threadIdx.x & 63) == dst ? value : v[0];
#else
    v[dst] = value;
#endif
}

// lanes whose bit is set in the wave-uniform mask m take `value` (wave-uniform); the others keep what they have.  One v_cndmask_b32
// with the mask as its scalar-pair operand: written in C (`(m >> lane) & 1 ? value : v`) the compiler shifts the mask by the lane
// number in every lane -- six vector instructions where the hardware takes the mask as it is.
WDEV void wave_select(uint32_t* v, uint64_t m, uint32_t value) {
#if LEP_ON_GPU
    __asm__("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[0]) : "v"(value), "s"(m));
#else
    for (int i = 0; i < 64; ++i) if ((m >> i) & 1ull) v[i] = value;
#endif
}
WDEV int popc64(uint64_t m) { return __builtin_popcountll(m); }

// Global memory through pointers whose address space the compiler cannot see (they come out of descriptors in memory): a plain
// dereference is a FLAT instruction, which counts against the LDS wait counter as well -- every wait for an LDS read then also
// waits for whatever was requested from HBM and is still in flight.  gld / gst name the address space (lep_enc5.h has the same pair).
#if LEP_ON_GPU
template <class T> WDEV T gld(const T* p) { return *(const __attribute__((address_space(1))) T*)(uintptr_t)p; }
template <class T> WDEV void gst(T* p, const T& v) { *(__attribute__((address_space(1))) T*)(uintptr_t)p = v; }
#else
template <class T> WDEV T gld(const T* p) { return *p; }
template <class T> WDEV void gst(T* p, const T& v) { *p = v; }
#endif

// set bits of a wave-uniform mask below lane l (inside LANES(l)): v_mbcnt_lo / _hi
WDEV int mbcnt(uint64_t m, int l) {
#if LEP_ON_GPU
    (void)l;
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#else
    return __builtin_popcountll(m & ((1ull << l) - 1));
#endif
}

}  // namespace lepwave
