// micro-benchmark: what one wave-instruction of each kind costs a CU when 8 wavefronts per SIMD keep issuing it (the regime
// the coder kernels run in).  Each kernel variant executes N x 16 copies of one instruction on independent registers;
// printed: cycles per instruction per CU (2.4 GHz), relative to v_add_u32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int MODE>
__global__ __launch_bounds__(64, 8) void k(int n, uint32_t* out, const uint32_t* in) {
    __shared__ uint32_t lds[256];
    uint32_t a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 7, d = 16, e = threadIdx.x * 4, a1 = 1, a2 = 2, a3 = 3;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w4 = {1, 2, 3, 4};
    uint64_t q = ((uint64_t)a << 32) | b;
    uint32_t s0 = blockIdx.x & 31, s1 = 3;
    uint64_t sq = blockIdx.x;
    lds[threadIdx.x] = a; lds[threadIdx.x + 64] = b; lds[threadIdx.x + 128] = c; lds[threadIdx.x + 192] = d;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) __asm__ volatile(REP16("v_add_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 1) __asm__ volatile(REP16("v_lshlrev_b32 %0, %1, %0\n") : "+v"(a) : "v"(c));
        if (MODE == 2) __asm__ volatile(REP16("v_lshlrev_b64 %0, %1, %0\n") : "+v"(q) : "v"(c));
        if (MODE == 3) __asm__ volatile(REP16("v_mul_u32_u24 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 4) __asm__ volatile(REP16("v_mul_lo_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 5) __asm__ volatile(REP16("v_mad_u64_u32 %0, vcc, %1, %2, %0\n") : "+v"(q) : "v"(a), "v"(b) : "vcc");
        if (MODE == 6) __asm__ volatile(REP16("v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(a) : "v"(b) : "vcc");
        if (MODE == 7) __asm__ volatile(REP16("v_ffbh_u32 %0, %0\n") : "+v"(a));
        if (MODE == 8) __asm__ volatile(REP16("v_readlane_b32 %0, %1, 5\n") : "+s"(s0) : "v"(a));
        if (MODE == 9) __asm__ volatile(REP16("v_readfirstlane_b32 %0, %1\n") : "+s"(s0) : "v"(a));
        if (MODE == 10) __asm__ volatile(REP16("v_min_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 11) __asm__ volatile(REP16("v_perm_b32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 12) __asm__ volatile(REP16("v_bfe_u32 %0, %0, %1, 8\n") : "+v"(a) : "v"(c));
        if (MODE == 13) __asm__ volatile(REP16("v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n") : "+v"(a) : "v"(b));
        if (MODE == 14) __asm__ volatile(REP16("v_cmp_ge_u32 vcc, %0, %1\n") : : "v"(a), "v"(b) : "vcc");
        if (MODE == 15) __asm__ volatile(REP16("v_cmp_ge_u32_e64 %0, %1, %2\n") : "+s"(sq) : "v"(a), "v"(b));
        if (MODE == 16) __asm__ volatile(REP16("s_add_i32 %0, %0, %1\n") : "+s"(s0) : "s"(s1) : "scc");
        if (MODE == 17) __asm__ volatile(REP16("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "+v"(a) : "v"(d));
        if (MODE == 18) __asm__ volatile(REP16("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(a) : "v"(d));
        if (MODE == 19) __asm__ volatile(REP16("v_alignbit_b32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 20) __asm__ volatile(REP16("v_lshl_add_u32 %0, %0, 2, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 21) __asm__ volatile(REP16("v_sub_co_u32 %0, vcc, %0, %1\n") : "+v"(a) : "v"(b) : "vcc");
        if (MODE == 22) __asm__ volatile(REP16("v_mov_b32 %0, %1\n") : "+v"(a) : "s"(s0));
        if (MODE == 23) __asm__ volatile(REP16("v_add_u32 %0, %0, %2\n s_add_i32 %1, %1, %3\n") : "+v"(a), "+s"(s0) : "v"(b), "s"(s1) : "scc");   // pairs: 1 VALU + 1 SALU
        if (MODE == 24) __asm__ volatile(REP16("v_add_u32 %0, %0, %2\n s_add_i32 %1, %1, %3\n s_add_i32 %1, %1, %3\n") : "+v"(a), "+s"(s0) : "v"(b), "s"(s1) : "scc");   // 1 VALU + 2 SALU
        if (MODE == 25) __asm__ volatile(REP16("v_add_u32 %0, %0, %2\n v_add_u32 %0, %0, %2\n s_add_i32 %1, %1, %3\n") : "+v"(a), "+s"(s0) : "v"(b), "s"(s1) : "scc");   // 2 VALU + 1 SALU
        if (MODE == 26) __asm__ volatile(REP16("v_cmp_ge_u32 vcc, %0, %1\n s_cbranch_vccz 0\n") : : "v"(a), "v"(b) : "vcc");   // compare + untaken/taken-to-next branch
        if (MODE == 27) __asm__ volatile(REP16("ds_read_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "+v"(a) : "v"(d));
        if (MODE == 28) __asm__ volatile(REP16("v_bfi_b32 %0, %1, %0, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if (MODE == 29) __asm__ volatile(REP16("s_lshl_b64 %0, %0, 1\n") : "+s"(sq) : : "scc");
        if (MODE == 30) __asm__ volatile(REP16("s_flbit_i32_b32 %0, %0\n") : "+s"(s0));
        if (MODE == 31) __asm__ volatile(REP16("s_mul_i32 %0, %0, %1\n") : "+s"(s0) : "s"(s1));
        if (MODE == 32) __asm__ volatile(REP4("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n") : "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");
        if (MODE == 33) __asm__ volatile(REP16("v_cmp_ge_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n") : "+v"(a) : "v"(b) : "vcc");
        if (MODE == 34) __asm__ volatile(REP16("v_cndmask_b32_e64 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "s"(sq));
        if (MODE == 35) __asm__ volatile(REP16("v_cmp_ge_u32 vcc, %0, %1\n v_sub_u32 %0, %0, %1\n v_min_u32 %0, %0, %1\n") : "+v"(a) : "v"(b) : "vcc");
        if (MODE == 36) __asm__ volatile(REP16("ds_read_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "+v"(a) : "v"(d));          // aligned, broadcast
        if (MODE == 37) __asm__ volatile(REP16("ds_read_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "+v"(a) : "v"(e));          // aligned, one dword per lane
        if (MODE == 38) __asm__ volatile(REP16("ds_read_b128 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(w4) : "v"(d));        // broadcast 16 bytes
        if (MODE == 39) __asm__ volatile(REP16("ds_write_b32 %1, %0\n") "s_waitcnt lgkmcnt(0)\n" : : "v"(a), "v"(e));
        if (MODE == 40) __asm__ volatile(REP16("ds_read_u8 %0, %1\n") "s_waitcnt lgkmcnt(0)\n" : "+v"(a) : "v"(d));
        if (MODE == 41) __asm__ volatile(REP16("ds_bpermute_b32 %0, %1, %0\n") "s_waitcnt lgkmcnt(0)\n" : "+v"(a) : "v"(e));
        if (MODE == 42) __asm__ volatile(REP16("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n") : "+v"(a) : "v"(d));          // aligned broadcast, waited one by one (latency)
        if (MODE == 43) __asm__ volatile(REP16("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a));
        if (MODE == 44) __asm__ volatile(REP16("v_add_u32 %0, %0, %1\n s_cbranch_scc1 0\n") : "+v"(a) : "v"(b));   // VALU + a never-taken scalar branch
        if (MODE == 45) __asm__ volatile(REP16("s_cmp_lt_u32 %0, %1\n s_cselect_b32 %0, %0, %1\n") : "+s"(s0) : "s"(s1) : "scc");
        if (MODE == 46) __asm__ volatile(REP16("v_and_b32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        if (MODE == 47) __asm__ volatile(REP16("v_sub_u32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        // does a vector instruction cost less when few lanes are enabled?  (16 adds under a narrowed exec mask; the two scalar
        // moves around them issue beside other waves' vector work)
        if (MODE == 48) __asm__ volatile("s_mov_b64 %2, exec\n s_mov_b64 exec, 1\n" REP16("v_add_u32 %0, %0, %1\n") "s_mov_b64 exec, %2\n" : "+v"(a), "+v"(b), "+s"(sq) : );
        if (MODE == 49) __asm__ volatile("s_mov_b64 %2, exec\n s_mov_b64 exec, 0xffff\n" REP16("v_add_u32 %0, %0, %1\n") "s_mov_b64 exec, %2\n" : "+v"(a), "+v"(b), "+s"(sq) : );
        if (MODE == 50) __asm__ volatile("s_mov_b64 %2, exec\n s_mov_b32 exec_lo, -1\n s_mov_b32 exec_hi, 0\n" REP16("v_add_u32 %0, %0, %1\n") "s_mov_b64 exec, %2\n" : "+v"(a), "+v"(b), "+s"(sq) : );
        if (MODE == 51) __asm__ volatile("s_mov_b64 %2, exec\n s_mov_b64 exec, 1\n" REP16("v_lshlrev_b32 %0, %1, %0\n") "s_mov_b64 exec, %2\n" : "+v"(a), "+v"(c), "+s"(sq) : );
        if (MODE == 52) __asm__ volatile("s_mov_b64 %2, exec\n s_mov_b64 exec, 1\n" REP16("ds_read_b32 %0, %1\n") "s_waitcnt lgkmcnt(0)\n s_mov_b64 exec, %2\n" : "+v"(a), "+v"(d), "+s"(sq) : );
    }
    out[blockIdx.x * 64 + threadIdx.x] = a1 + a2 + a3 + e + w4.x + w4.y + w4.z + w4.w + a + b + c + d + (uint32_t)q + (uint32_t)(q >> 32) + s0 + s1 + (uint32_t)sq + lds[(a + threadIdx.x) & 255];
}

template <int MODE> static double run(int wps, uint32_t* d, int n, int per_rep) {
    const int blocks = 256 * 4 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, 100, d, d); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, n, d, d); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double groups = (double)blocks * n * 16;          // instruction groups executed
    return ms * 1e-3 * 2.4e9 * 256 / groups;                // cycles per group per CU
}
#define ROW(M, name, per) { double c1 = run<M>(1, d, n, per), c8 = run<M>(8, d, n, per); \
    printf("%-44s 1 wave/SIMD: %6.2f   8 waves/SIMD: %6.2f cycles per group per CU  (x %.2f of v_add_u32)\n", name, c1, c8, c8 / base); }
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 32 * 64 * 4 + 4096);
    const int n = 4000;
    const double base = run<0>(8, d, n, 1);
    ROW(0, "v_add_u32", 1) ROW(1, "v_lshlrev_b32", 1) ROW(2, "v_lshlrev_b64", 1) ROW(3, "v_mul_u32_u24", 1) ROW(4, "v_mul_lo_u32", 1)
    ROW(5, "v_mad_u64_u32", 1) ROW(6, "v_cndmask_b32 (vcc)", 1) ROW(7, "v_ffbh_u32", 1) ROW(8, "v_readlane_b32", 1) ROW(9, "v_readfirstlane_b32", 1)
    ROW(10, "v_min_u32", 1) ROW(11, "v_perm_b32", 1) ROW(12, "v_bfe_u32", 1) ROW(13, "v_mul_u32_u24_sdwa", 1) ROW(14, "v_cmp_ge_u32 -> vcc", 1)
    ROW(15, "v_cmp_ge_u32_e64 -> sgpr pair", 1) ROW(16, "s_add_i32", 1) ROW(17, "ds_read_b32 + wait (latency chain)", 1) ROW(18, "ds_bpermute_b32 + wait", 1)
    ROW(19, "v_alignbit_b32", 1) ROW(20, "v_lshl_add_u32", 1) ROW(21, "v_sub_co_u32", 1) ROW(22, "v_mov_b32 v, s", 1)
    ROW(23, "pair: v_add + s_add", 2) ROW(24, "triple: v_add + 2 s_add", 3) ROW(25, "triple: 2 v_add + s_add", 3) ROW(26, "v_cmp + s_cbranch_vccz (fallthrough)", 2)
    ROW(27, "ds_read_b32 x16 then one wait", 1) ROW(28, "v_bfi_b32", 1) ROW(29, "s_lshl_b64", 1) ROW(30, "s_flbit_i32_b32", 1) ROW(31, "s_mul_i32", 1)
    ROW(32, "v_cndmask_b32 (vcc), 4 independent", 1) ROW(33, "pair: v_cmp -> vcc + v_cndmask (vcc)", 2) ROW(34, "v_cndmask_b32_e64 (sgpr pair)", 1)
    ROW(35, "triple: v_cmp + v_sub + v_min", 3) ROW(36, "ds_read_b32 aligned broadcast x16, one wait", 1) ROW(37, "ds_read_b32 per-lane dword x16, one wait", 1)
    ROW(38, "ds_read_b128 broadcast x16, one wait", 1) ROW(39, "ds_write_b32 x16, one wait", 1) ROW(40, "ds_read_u8 x16, one wait", 1) ROW(41, "ds_bpermute_b32 x16, one wait", 1)
    ROW(42, "ds_read_b32 aligned + wait each", 1) ROW(43, "v_mov_b32 dpp row_shr", 1) ROW(44, "pair: v_add + s_cbranch_scc1 (not taken)", 2)
    ROW(45, "pair: s_cmp + s_cselect", 2) ROW(46, "v_and_b32", 1) ROW(47, "v_sub_u32", 1)
    ROW(48, "v_add_u32, exec = lane 0 only", 1) ROW(49, "v_add_u32, exec = lanes 0..15", 1) ROW(50, "v_add_u32, exec = lanes 0..31", 1)
    ROW(51, "v_lshlrev_b32, exec = lane 0 only", 1) ROW(52, "ds_read_b32 x16, exec = lane 0 only, one wait", 1)
    return 0;
}
