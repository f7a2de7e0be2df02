// micro-benchmark: peak issue rates per CU.  (a) independent VALU adds (8 accumulators), (b) one dependent VALU chain,
// (c) independent SALU adds, (d) VALU + SALU interleaved in one wave, at 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void k(int n, int mode, uint32_t* out) {
    uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    uint32_t s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    if (mode == 0) {
        for (int i = 0; i < n; ++i) {
            __asm__ volatile("v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %4\n"
                             "v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    } else if (mode == 1) {
        for (int i = 0; i < n; ++i) {
            __asm__ volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                             "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                             : "+v"(a0) : "v"(a1));
        }
    } else if (mode == 2) {
        for (int i = 0; i < n; ++i) {
            __asm__ volatile("s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n"
                             "s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        }
    } else if (mode == 3) {
        for (int i = 0; i < n; ++i) {
            __asm__ volatile("v_add_u32 %0, %0, %1\n s_add_u32 %4, %4, %5\n v_add_u32 %1, %1, %2\n s_add_u32 %5, %5, %6\n"
                             "v_add_u32 %2, %2, %3\n s_add_u32 %6, %6, %7\n v_add_u32 %3, %3, %0\n s_add_u32 %7, %7, %4\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        }
    } else {   // dependent scalar chain
        for (int i = 0; i < n; ++i) {
            __asm__ volatile("s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n"
                             "s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n s_add_u32 %0, %0, %1\n"
                             : "+s"(s0) : "s"(s1) : "scc");
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s0 + s1 + s2 + s3;
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 32 * 64 * 4);
    const int n = 50000;
    const char* names[] = {"valu x8 independent", "valu dependent chain", "salu x4 independent", "valu+salu interleaved", "salu dependent chain"};
    for (int mode = 0; mode < 5; ++mode)
        for (int wps : {1, 2, 4, 8}) {
            int blocks = 256 * 4 * wps;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<<<blocks, 64>>>(1000, mode, d); hipDeviceSynchronize();
            hipEventRecord(e0); k<<<blocks, 64>>>(n, mode, d); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double inst = (double)blocks * n * 8;
            printf("%-24s waves/SIMD %d: %.2f ms, %.3f wave-instructions/cycle/CU at 2.4 GHz (per-wave %.2f cycles/instr)\n", names[mode], wps, ms,
                   inst / (ms * 1e-3) / 2.4e9 / 256, ms * 1e-3 * 2.4e9 / (n * 8.0));
        }
    return 0;
}
