// micro-benchmark: where should the serial bool-coder recurrence run?  The same bin loop (a) wave-uniform on the scalar
// unit, (b) on the vector ALU (values hidden from the uniformity analysis, lane 0 meaningful), (c) a launch that mixes
// both kinds of wave (odd workgroups vector, even scalar), at 1..8 waves per SIMD.  Reports aggregate Gbins/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
struct BD { uint64_t value; int count; uint32_t range; };
template <bool VEC> __device__ __forceinline__ uint32_t hide(uint32_t v) { if (VEC) __asm__ volatile("" : "+v"(v)); return v; }
template <bool VEC>
__device__ __forceinline__ int get(BD& b, uint32_t prob, const uint32_t* words, uint32_t& wi) {
    const uint32_t split = 1 + (((b.range - 1) * prob) >> 8);
    if (b.count < 0) {
        uint32_t w = __builtin_bswap32(words[wi & 1023]);
        if (!VEC) w = __builtin_amdgcn_readfirstlane(w);
        ++wi; b.value |= ((uint64_t)w << 32) >> (b.count + 8); b.count += 32;
    }
    const uint32_t big = split << 24;
    const int bit = (uint32_t)(b.value >> 32) >= big;
    if (bit) { b.range -= split; b.value -= (uint64_t)big << 32; } else b.range = split;
    const int shift = __builtin_clz(b.range) - 24;
    b.range <<= shift; b.value <<= shift; b.count -= shift;
    return bit;
}
template <bool VEC>
__device__ __forceinline__ void body(const uint32_t* words, const uint32_t* pk, int n, int* out, unsigned long long* cyc) {
    BD b; b.value = 0; b.count = (int)hide<VEC>((uint32_t)-8); b.range = hide<VEC>(255);
    uint32_t wi = hide<VEC>(blockIdx.x);
    int acc = 0, bins = 0;
    uint32_t mypk = pk[threadIdx.x];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        uint32_t p = VEC ? hide<VEC>(__builtin_amdgcn_readlane(mypk, i & 63)) : __builtin_amdgcn_readlane(mypk, i & 63);
        int len = 0;
#pragma nounroll
        for (; len < 4; ++len) { ++bins; if (!get<VEC>(b, (p >> (len * 8)) & 255, words, wi)) break; }
        acc += len;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x] = acc + bins; cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = bins; }
}
// mode 0 scalar, 1 vector, 2 mixed by workgroup parity, 3 mixed 1 vector in 4
__global__ __launch_bounds__(64) void k(const uint32_t* words, const uint32_t* pk, int n, int* out, unsigned long long* cyc, int mode) {
    bool vec = mode == 1 || (mode == 2 && (blockIdx.x & 1)) || (mode == 3 && (blockIdx.x & 3) == 3) || (mode == 4 && (blockIdx.x & 3) != 0);
    if (vec) body<true>(words, pk, n, out, cyc); else body<false>(words, pk, n, out, cyc);
}
int main() {
    std::vector<uint32_t> hw(1024), hp(64);
    uint32_t x = 12345;
    for (auto& w : hw) { x = x * 1664525u + 1013904223u; w = x; }
    for (auto& p : hp) { x = x * 1664525u + 1013904223u; p = (x | 0x20202020u) & 0xdfdfdfdfu; }
    uint32_t *dw, *dp; int* dout; unsigned long long* dc;
    const int maxb = 256 * 32;
    hipMalloc(&dw, 4096); hipMalloc(&dp, 256); hipMalloc(&dout, maxb * 4); hipMalloc(&dc, maxb * 16);
    hipMemcpy(dw, hw.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(dp, hp.data(), 256, hipMemcpyHostToDevice);
    const int n = 100000;
    const char* names[] = {"scalar", "vector", "mixed 1:1", "mixed 3s:1v", "mixed 1s:3v"};
    for (int mode = 0; mode < 5; ++mode)
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * 4 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<blocks, 64>>>(dw, dp, 1000, dout, dc, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<blocks, 64>>>(dw, dp, n, dout, dc, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> hc(blocks * 2);
        hipMemcpy(hc.data(), dc, blocks * 16, hipMemcpyDeviceToHost);
        double cyc = 0, bins = 0;
        for (int i = 0; i < blocks; ++i) { cyc += hc[2 * i]; bins += hc[2 * i + 1]; }
        printf("%-12s waves/SIMD %d: %.2f ms, %.1f cycles/bin/wave, %.2f Gbins/s aggregate\n", names[mode], wps, ms, cyc / bins, bins / ms / 1e6);
    }
    return 0;
}
