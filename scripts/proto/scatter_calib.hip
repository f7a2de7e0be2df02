// scatter_calib.hip -- what do the memory-side counters (TCC_EA0_RDREQ / _WRREQ, FETCH_SIZE / WRITE_SIZE) report for the coder
// kernels' access pattern?  MI355X_MICROARCH.md calibrates FETCH_SIZE for wide streaming reads only (it reports half of the
// bytes there).  Three kernels over one large buffer, each with a known number of accesses:
//   stream   every lane reads consecutive 16-byte words (coalesced 1 KiB per wave instruction)
//   gather   every lane reads ONE 16-byte word at a random 16-byte-aligned address (the decoder's context-group prefetch)
//   rmw      the same, then writes the word back changed (the owner lanes' adapt + store)
// Run under rocprofv3 --pmc (a --pmc pass of its own); prints accesses and wall time per kernel so that requests per access
// and requests per second can be read off next to the counters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>

__global__ void k_stream(const uint4* __restrict__ p, size_t n16, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__device__ inline size_t mix(size_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__global__ void k_gather(const uint4* __restrict__ p, size_t n16, int per_lane, unsigned* sink) {
    unsigned acc = 0;
    size_t h = mix((size_t)blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (int k = 0; k < per_lane; ++k) { h = mix(h + k); const uint4 v = p[h % n16]; acc ^= v.x ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_rmw(uint4* p, size_t n16, int per_lane) {
    size_t h = mix((size_t)blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (int k = 0; k < per_lane; ++k) { h = mix(h + k); uint4 v = p[h % n16]; v.x += 1; v.w ^= 5; p[h % n16] = v; }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t gb = argc > 1 ? (size_t)atoi(argv[1]) : 24;
    const size_t bytes = gb << 30, n16 = bytes / 16;
    uint4* d = nullptr; unsigned* sink = nullptr;
    if (hipMalloc((void**)&d, bytes) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, bytes); hipDeviceSynchronize();
    const int blocks = 8192, threads = 64, per_lane = 4096;   // the coder kernels' shape: 8192 single-wave workgroups
    double t = now(); hipLaunchKernelGGL(k_stream, dim3(blocks * 4), dim3(256), 0, 0, d, n16, sink); hipDeviceSynchronize();
    printf("stream: %zu accesses of 16 B (%.3f GB) in %.4f s = %.1f GB/s\n", n16, bytes / 1e9, now() - t, bytes / 1e9 / (now() - t));
    const size_t acc = (size_t)blocks * threads * per_lane;
    t = now(); hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(threads), 0, 0, d, n16, per_lane, sink); hipDeviceSynchronize();
    double dt = now() - t;
    printf("gather: %zu accesses of 16 B in %.4f s = %.2f G accesses/s (%.1f GB/s of 64-byte sectors)\n", acc, dt, acc / dt / 1e9, acc * 64 / dt / 1e9);
    t = now(); hipLaunchKernelGGL(k_rmw, dim3(blocks), dim3(threads), 0, 0, d, n16, per_lane); hipDeviceSynchronize();
    dt = now() - t;
    printf("rmw: %zu read+write pairs of 16 B in %.4f s = %.2f G pairs/s\n", acc, dt, acc / dt / 1e9);
    hipFree(d); hipFree(sink);
    return 0;
}
