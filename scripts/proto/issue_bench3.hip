// micro-benchmark 2: bool-decoder bin loop written for the VECTOR ALU with uniform (ballot) branches and selects instead of
// exec-mask control flow, vs the scalar-unit version, vs launches mixing both kinds of wave.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ uint32_t hidev(uint32_t v) { __asm__ volatile("" : "+v"(v)); return v; }
struct BDV { uint32_t vhi, vlo; int count; uint32_t range; };
__device__ __forceinline__ bool getv(BDV& b, uint32_t prob, const uint32_t* words, uint32_t& wi) {
    if (__builtin_amdgcn_ballot_w64(b.count < 0)) {
        uint32_t w = __builtin_bswap32(words[wi & 1023]);
        ++wi;
        uint64_t add = ((uint64_t)w << 32) >> (b.count + 8);
        b.vhi |= (uint32_t)(add >> 32); b.vlo |= (uint32_t)add; b.count += 32;
    }
    const uint32_t split = 1 + (__umul24(b.range - 1, prob) >> 8);
    const uint32_t big = split << 24;
    const bool bit = b.vhi >= big;
    const uint32_t r1 = b.range - split;
    b.range = bit ? r1 : split;
    b.vhi -= bit ? big : 0u;
    const int shift = __builtin_clz(b.range) - 24;
    b.range <<= shift;
    uint64_t v = (((uint64_t)b.vhi << 32) | b.vlo) << shift;
    b.vhi = (uint32_t)(v >> 32); b.vlo = (uint32_t)v;
    b.count -= shift;
    return __builtin_amdgcn_ballot_w64(bit) != 0;
}
struct BD { uint64_t value; int count; uint32_t range; };
__device__ __forceinline__ int gets(BD& b, uint32_t prob, const uint32_t* words, uint32_t& wi) {
    const uint32_t split = 1 + (((b.range - 1) * prob) >> 8);
    if (b.count < 0) {
        uint32_t w = __builtin_amdgcn_readfirstlane(__builtin_bswap32(words[wi & 1023]));
        ++wi; b.value |= ((uint64_t)w << 32) >> (b.count + 8); b.count += 32;
    }
    const uint32_t big = split << 24;
    const int bit = (uint32_t)(b.value >> 32) >= big;
    if (bit) { b.range -= split; b.value -= (uint64_t)big << 32; } else b.range = split;
    const int shift = __builtin_clz(b.range) - 24;
    b.range <<= shift; b.value <<= shift; b.count -= shift;
    return bit;
}
__device__ __forceinline__ void body_v(const uint32_t* words, const uint32_t* pk, int n, int* out, unsigned long long* cyc) {
    BDV b; b.vhi = hidev(0); b.vlo = hidev(0); b.count = (int)hidev((uint32_t)-8); b.range = hidev(255);
    uint32_t wi = hidev(blockIdx.x);
    int acc = 0, bins = 0;
    uint32_t mypk = pk[threadIdx.x];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        uint32_t p = __builtin_amdgcn_readlane(mypk, i & 63);
        int len = 0;
#pragma nounroll
        for (; len < 4; ++len) { ++bins; if (!getv(b, (p >> (len * 8)) & 255, words, wi)) break; }
        acc += len;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x] = acc + bins + b.range; cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = bins; }
}
__device__ __forceinline__ void body_s(const uint32_t* words, const uint32_t* pk, int n, int* out, unsigned long long* cyc) {
    BD b; b.value = 0; b.count = -8; b.range = 255;
    uint32_t wi = blockIdx.x;
    int acc = 0, bins = 0;
    uint32_t mypk = pk[threadIdx.x];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        uint32_t p = __builtin_amdgcn_readlane(mypk, i & 63);
        int len = 0;
#pragma nounroll
        for (; len < 4; ++len) { ++bins; if (!gets(b, (p >> (len * 8)) & 255, words, wi)) break; }
        acc += len;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x] = acc + bins; cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = bins; }
}

struct BDH { uint32_t vhi, vlo; int count; uint32_t range; };
__device__ __forceinline__ bool geth(BDH& b, uint32_t prob, const uint32_t* words, uint32_t& wi) {
    if (b.count < 0) {
        uint32_t w = __builtin_bswap32(words[wi & 1023]);
        ++wi;
        uint64_t add = ((uint64_t)w << 32) >> (b.count + 8);
        b.vhi |= (uint32_t)(add >> 32); b.vlo |= (uint32_t)add; b.count += 32;
    }
    const uint32_t split = 1 + (((b.range - 1) * prob) >> 8);
    const uint32_t big = split << 24;
    const bool bit = __builtin_amdgcn_ballot_w64(b.vhi >= big) != 0;
    const uint32_t d = b.vhi - big;
    b.vhi = d < b.vhi ? d : b.vhi;
    b.range = bit ? b.range - split : split;
    const int shift = __builtin_clz(b.range) - 24;
    b.range <<= shift;
    uint64_t v = (((uint64_t)b.vhi << 32) | b.vlo) << shift;
    b.vhi = (uint32_t)(v >> 32); b.vlo = (uint32_t)v;
    b.count -= shift;
    return bit;
}
__device__ __forceinline__ void body_h(const uint32_t* words, const uint32_t* pk, int n, int* out, unsigned long long* cyc) {
    BDH b; b.vhi = hidev(0); b.vlo = hidev(0); b.count = -8; b.range = 255;
    uint32_t wi = blockIdx.x;
    int acc = 0, bins = 0;
    uint32_t mypk = pk[threadIdx.x];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        uint32_t p = __builtin_amdgcn_readlane(mypk, i & 63);
        int len = 0;
#pragma nounroll
        for (; len < 4; ++len) { ++bins; if (!geth(b, (p >> (len * 8)) & 255, words, wi)) break; }
        acc += len;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x] = acc + bins + b.range; cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = bins; }
}
__global__ __launch_bounds__(64) void k(const uint32_t* words, const uint32_t* pk, int n, int* out, unsigned long long* cyc, int mode) {
    const uint32_t h = (blockIdx.x * 0x9E3779B1u) >> 28;   // placement-independent hash, 0..15
    bool vec = mode == 1 || (mode == 2 && h < 8) || (mode == 3 && h < 4) || (mode == 4 && h < 12) || (mode == 5 && h < 10) || (mode == 6 && h < 14);
    if (mode == 7) body_h(words, pk, n, out, cyc); else if (vec) body_v(words, pk, n, out, cyc); else body_s(words, pk, n, out, cyc);
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
    const char* names[] = {"scalar", "vector", "mixed 8s:8v", "mixed 12s:4v", "mixed 4s:12v", "mixed 6s:10v", "mixed 2s:14v", "hybrid"};
    for (int mode : {0, 1, 7})
    for (int wps : {4, 8}) {
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
