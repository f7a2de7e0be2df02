// what the pieces of device-memory acquisition cost on this box: hipMalloc / hipFree of a big buffer against hipMemCreate / hipMemMap /
// hipMemSetAccess / hipMemUnmap / hipMemRelease in chunks of 64 MB and 1 GB (and touching the memory afterwards)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void touch(char* p, size_t n) { for (size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * 4096; i < n; i += (size_t)gridDim.x * blockDim.x * 4096) p[i] = 1; }
int main() {
    const size_t total = (size_t)96 << 30;
    CK(hipSetDevice(0));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now(); void* p = nullptr; CK(hipMalloc(&p, total)); double t1 = now();
        touch<<<4096, 256>>>((char*)p, total); CK(hipDeviceSynchronize()); double t2 = now();
        touch<<<4096, 256>>>((char*)p, total); CK(hipDeviceSynchronize()); double t3 = now();
        CK(hipFree(p)); double t4 = now();
        printf("hipMalloc 96 GB: %.3f s, first touch %.3f s, second touch %.3f s, hipFree %.3f s\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    }
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu\n", gran);
    for (size_t chunk : {(size_t)64 << 20, (size_t)1 << 30}) {
        const size_t n = total / chunk;
        void* va = nullptr; double t0 = now(); CK(hipMemAddressReserve(&va, total, 0, nullptr, 0)); double t1 = now();
        std::vector<hipMemGenericAllocationHandle_t> h(n);
        for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&h[i], chunk, &prop, 0));
        double t2 = now();
        for (int rep = 0; rep < 2; ++rep) {
            double a = now();
            for (size_t i = 0; i < n; ++i) CK(hipMemMap((char*)va + i * chunk, chunk, 0, h[i], 0));
            double b = now();
            CK(hipMemSetAccess(va, total, &acc, 1)); double c = now();
            touch<<<4096, 256>>>((char*)va, total); CK(hipDeviceSynchronize()); double d = now();
            touch<<<4096, 256>>>((char*)va, total); CK(hipDeviceSynchronize()); double e = now();
            CK(hipMemUnmap(va, total)); double f = now();
            printf("chunk %4zu MB x %zu: map %.3f s, setaccess %.3f s, first touch %.3f s, second touch %.3f s, unmap %.3f s\n", chunk >> 20, n, b - a, c - b, d - c, e - d, f - e);
        }
        double t3 = now();
        for (size_t i = 0; i < n; ++i) CK(hipMemRelease(h[i]));
        double t4 = now();
        CK(hipMemAddressFree(va, total));
        printf("chunk %4zu MB: reserve %.3f s, create all %.3f s, release all %.3f s\n", chunk >> 20, t1 - t0, t2 - t1, t4 - t3);
    }
    return 0;
}
