// lep_gpu.hip -- layer 1 of the C ABI: HIP kernels for gfx950 and the runtime object that launches them.
// One wavefront per (image, thread segment) work item; per-segment adaptive model resident in HBM
// (2.9 MB each, hot tables first), coefficient frames read/written in place in the reference's
// AlignedBlock layout, streams written to / read from one device arena.  No CPU fallback: every entry
// point returns LEP_GPU_ERROR if HIP reports a failure.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lepton_mi355x.h"

#define LEP_DEV __device__ __forceinline__
#include "lep_core.h"
#include "lep_enc2.h"
#include "lep_dec2.h"

using namespace lepdev;

// helpers so the kernel template compiles for both coder kinds
namespace lepdev {
__device__ inline uint32_t finish_stream(BoolCoder<false>& b) { return b.finish(); }
__device__ inline uint32_t finish_stream(BoolCoder<true>&) { return 0; }
__device__ inline bool stream_overflow(BoolCoder<false>& b) { return b.overflow; }
__device__ inline bool stream_overflow(BoolCoder<true>&) { return false; }
}  // namespace lepdev

namespace {

__device__ void reset_segment_state(uint32_t* model, NSum* ns, int ns_count, int lane) {
    // model reset = Branch::identity() everywhere (model.hh:114-125); 16-byte coalesced stores
    uint4* m4 = reinterpret_cast<uint4*>(model);
    const uint4 init = make_uint4(kBranchInit, kBranchInit, kBranchInit, kBranchInit);
    for (uint32_t i = lane; i < kModelBranches / 4; i += 64) m4[i] = init;
    uint32_t* n32 = reinterpret_cast<uint32_t*>(ns);
    const uint32_t words = (uint32_t)ns_count * (sizeof(NSum) / 4);
    for (uint32_t i = lane; i < words; i += 64) n32[i] = 0;
}

template <bool DEC>
__global__ __launch_bounds__(64) void lep_segment_kernel(const ImageDev* __restrict__ images, const SegDev* __restrict__ segs,
                                                         uint32_t* models, NSum* ns_area, const uint64_t* ns_offsets,
                                                         uint8_t* streams, uint32_t* stream_len, int32_t* status,
                                                         uint32_t* bins) {
    const int s = blockIdx.x, lane = threadIdx.x;
    const SegDev seg = segs[s];
    const ImageDev* img = images + seg.image;
    uint32_t* model = models + (size_t)s * kModelBranches;
    NSum* ns = ns_area + ns_offsets[s];
    reset_segment_state(model, ns, img->ns_total, lane);
    __syncthreads();
    if (lane != 0) return;
    SegmentCoder<DEC> sc;
    if (DEC) sc.bc.init_stream(streams + seg.stream_off, stream_len[s]);
    else sc.bc.init_stream(streams + seg.stream_off, seg.stream_cap);
    int rc = sc.run(img, seg, model, ns);
    if (!DEC) {
        uint32_t n = finish_stream(sc.bc);
        if (stream_overflow(sc.bc)) rc = LEP_BUFFER_TOO_SMALL;
        stream_len[s] = n;
    }
    status[s] = rc;
    bins[s] = sc.nbins;
}

// v2 encoder: wave-cooperative (lep_enc2.h); same arguments as lep_segment_kernel<false>
__global__ __launch_bounds__(64) void lep_encode_v2_kernel(const ImageDev* __restrict__ images, const SegDev* __restrict__ segs,
                                                           uint32_t* models, NSum* ns_area, const uint64_t* ns_offsets,
                                                           uint8_t* streams, uint32_t* stream_len, int32_t* status, uint32_t* bins) {
    __shared__ EncShared sh;
    const int s = blockIdx.x, lane = threadIdx.x;
    const SegDev seg = segs[s];
    const ImageDev* img = images + seg.image;
    uint32_t* model = models + (size_t)s * kModelBranches;
    NSum* ns = ns_area + ns_offsets[s];
    reset_segment_state(model, ns, img->ns_total, lane);
    __syncthreads();
    EncWave w;
    int rc = w.run(img, seg, model, ns, &sh, streams + seg.stream_off, seg.stream_cap);
    if (lane != 0) return;
    uint32_t n = rc ? 0 : w.bc.finish();
    if (!rc && w.bc.overflow) rc = LEP_BUFFER_TOO_SMALL;
    stream_len[s] = n;
    status[s] = rc;
    bins[s] = w.nbins;
}

// v2 decoder: wave-cooperative with prefetch rounds (lep_dec2.h)
__global__ __launch_bounds__(64) void lep_decode_v2_kernel(const ImageDev* __restrict__ images, const SegDev* __restrict__ segs,
                                                           uint32_t* models, NSum* ns_area, const uint64_t* ns_offsets,
                                                           uint8_t* streams, uint32_t* stream_len, int32_t* status, uint32_t* bins) {
    __shared__ DecShared sh;
    const int s = blockIdx.x, lane = threadIdx.x;
    const SegDev seg = segs[s];
    const ImageDev* img = images + seg.image;
    uint32_t* model = models + (size_t)s * kModelBranches;
    NSum* ns = ns_area + ns_offsets[s];
    reset_segment_state(model, ns, img->ns_total, lane);
    __syncthreads();
    DecWave w;
    int rc = w.run(img, seg, model, ns, &sh, streams + seg.stream_off, stream_len[s]);
    if (lane != 0) return;
    status[s] = rc;
    bins[s] = w.nbins;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
struct lep_gpu {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int decode_kernel = 2;   // same switch for the decoder (LEP_DECODE_KERNEL=1)
    int encode_kernel = 2;   // 2 = wave-cooperative (default), 1 = single-lane reference kernel (LEP_ENCODE_KERNEL=1)
    std::string err;
    // grow-only device workspace
    void* d_models = nullptr; size_t models_bytes = 0;
    void* d_ns = nullptr; size_t ns_bytes = 0;
    void* d_meta = nullptr; size_t meta_bytes = 0;      // ImageDev[] | SegDev[] | ns_offsets[] | bins[]
    uint32_t* d_bins = nullptr;
    std::vector<uint32_t> h_bins;
    // host-variant staging
    void* d_blocks = nullptr; size_t blocks_bytes = 0;
    void* d_streams = nullptr; size_t streams_bytes = 0;
    void* d_lens = nullptr; size_t lens_bytes = 0;
};

#define HIPCHK(g, call)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (g)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return LEP_GPU_ERROR;                                                              \
        }                                                                                      \
    } while (0)

static int ensure(lep_gpu* g, void** p, size_t* have, size_t need) {
    if (*have >= need) return 0;
    if (*p) HIPCHK(g, hipFree(*p));
    *p = nullptr; *have = 0;
    size_t want = need + need / 8;
    HIPCHK(g, hipMalloc(p, want));
    *have = want;
    return 0;
}

#include "lep_derive.h"

template <bool DEC>
static int launch(lep_gpu* g, const lep_image_desc* images, int nimg, const lep_segment* segs, int nseg, uint8_t* d_streams,
                  const uint64_t* stream_offsets, uint32_t* d_stream_len, int32_t* d_status, hipStream_t st) {
    if (nseg <= 0) return 0;
    std::vector<ImageDev> himg(nimg);
    for (int i = 0; i < nimg; ++i) {
        int rc = derive_image(images[i], &himg[i], !DEC);
        if (rc) return rc;
    }
    std::vector<SegDev> hseg(nseg);
    std::vector<uint64_t> hns(nseg);
    uint64_t ns_total = 0;
    for (int s = 0; s < nseg; ++s) {
        if (segs[s].image < 0 || segs[s].image >= nimg) return LEP_ASSERTION_FAILURE;
        hseg[s].image = segs[s].image; hseg[s].y0 = segs[s].luma_y_start; hseg[s].y1 = segs[s].luma_y_end;
        hseg[s].is_last = segs[s].is_last;
        hseg[s].stream_off = stream_offsets[s];
        uint64_t cap = stream_offsets[s + 1] - stream_offsets[s];
        hseg[s].stream_cap = (uint32_t)(cap > 0xffffffffu ? 0xffffffffu : cap);
        hns[s] = ns_total;
        ns_total += (uint64_t)himg[segs[s].image].ns_total;
    }
    HIPCHK(g, hipSetDevice(g->device));
    if (int rc = ensure(g, &g->d_models, &g->models_bytes, (size_t)nseg * kModelBranches * 4)) return rc;
    if (int rc = ensure(g, &g->d_ns, &g->ns_bytes, (size_t)ns_total * sizeof(NSum) + 16)) return rc;
    const size_t o_img = 0, o_seg = o_img + ((nimg * sizeof(ImageDev) + 255) & ~(size_t)255),
                 o_ns = o_seg + ((nseg * sizeof(SegDev) + 255) & ~(size_t)255),
                 o_bins = o_ns + ((nseg * sizeof(uint64_t) + 255) & ~(size_t)255), total = o_bins + nseg * sizeof(uint32_t);
    if (int rc = ensure(g, &g->d_meta, &g->meta_bytes, total)) return rc;
    char* meta = (char*)g->d_meta;
    HIPCHK(g, hipMemcpyAsync(meta + o_img, himg.data(), nimg * sizeof(ImageDev), hipMemcpyHostToDevice, st));
    HIPCHK(g, hipMemcpyAsync(meta + o_seg, hseg.data(), nseg * sizeof(SegDev), hipMemcpyHostToDevice, st));
    HIPCHK(g, hipMemcpyAsync(meta + o_ns, hns.data(), nseg * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    HIPCHK(g, hipStreamSynchronize(st));   // the host vectors above go out of scope
    g->d_bins = (uint32_t*)(meta + o_bins);
    g->h_bins.assign(nseg, 0);
    HIPCHK(g, hipEventRecord(g->ev0, st));
    if (DEC && g->decode_kernel == 2)
        hipLaunchKernelGGL(lep_decode_v2_kernel, dim3(nseg), dim3(64), 0, st, (const ImageDev*)(meta + o_img),
                           (const SegDev*)(meta + o_seg), (uint32_t*)g->d_models, (NSum*)g->d_ns, (const uint64_t*)(meta + o_ns),
                           d_streams, d_stream_len, d_status, g->d_bins);
    else if (!DEC && g->encode_kernel == 2)
        hipLaunchKernelGGL(lep_encode_v2_kernel, dim3(nseg), dim3(64), 0, st, (const ImageDev*)(meta + o_img),
                           (const SegDev*)(meta + o_seg), (uint32_t*)g->d_models, (NSum*)g->d_ns, (const uint64_t*)(meta + o_ns),
                           d_streams, d_stream_len, d_status, g->d_bins);
    else
        hipLaunchKernelGGL(lep_segment_kernel<DEC>, dim3(nseg), dim3(64), 0, st, (const ImageDev*)(meta + o_img),
                           (const SegDev*)(meta + o_seg), (uint32_t*)g->d_models, (NSum*)g->d_ns, (const uint64_t*)(meta + o_ns),
                           d_streams, d_stream_len, d_status, g->d_bins);
    HIPCHK(g, hipGetLastError());
    HIPCHK(g, hipEventRecord(g->ev1, st));
    g->timed = true;
    return 0;
}

extern "C" {

int lep_gpu_create(int device, lep_gpu** out) {
    lep_gpu* g = new lep_gpu;
    g->device = device;
    if (const char* e = getenv("LEP_ENCODE_KERNEL")) g->encode_kernel = atoi(e) == 1 ? 1 : 2;
    if (const char* e = getenv("LEP_DECODE_KERNEL")) g->decode_kernel = atoi(e) == 1 ? 1 : 2;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= device) { delete g; return LEP_GPU_ERROR; }
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&g->stream) != hipSuccess ||
        hipEventCreate(&g->ev0) != hipSuccess || hipEventCreate(&g->ev1) != hipSuccess) { delete g; return LEP_GPU_ERROR; }
    *out = g;
    return 0;
}

void lep_gpu_destroy(lep_gpu* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    (void)hipStreamSynchronize(g->stream);
    for (void* p : {g->d_models, g->d_ns, g->d_meta, g->d_blocks, g->d_streams, g->d_lens})
        if (p) (void)hipFree(p);
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

const char* lep_gpu_last_error(lep_gpu* g) { return g ? g->err.c_str() : "no gpu object"; }

int lep_gpu_encode_device(lep_gpu* g, const lep_image_desc* images, int nimg, const lep_segment* segs, int nseg,
                          uint8_t* d_streams, const uint64_t* stream_offsets, uint32_t* d_stream_len, int32_t* d_status,
                          void* hip_stream) {
    return launch<false>(g, images, nimg, segs, nseg, d_streams, stream_offsets, d_stream_len, d_status,
                         hip_stream ? (hipStream_t)hip_stream : g->stream);
}

int lep_gpu_decode_device(lep_gpu* g, const lep_image_desc* images, int nimg, const lep_segment* segs, int nseg,
                          const uint8_t* d_streams, const uint64_t* stream_offsets, const uint32_t* d_stream_len,
                          int32_t* d_status, void* hip_stream) {
    return launch<true>(g, images, nimg, segs, nseg, const_cast<uint8_t*>(d_streams), stream_offsets,
                        const_cast<uint32_t*>(d_stream_len), d_status, hip_stream ? (hipStream_t)hip_stream : g->stream);
}

int lep_gpu_sync(lep_gpu* g) {
    HIPCHK(g, hipSetDevice(g->device));
    HIPCHK(g, hipDeviceSynchronize());
    return 0;
}

double lep_gpu_last_kernel_ms(lep_gpu* g) {
    if (!g->timed) return -1.0;
    float ms = 0;
    if (hipEventSynchronize(g->ev1) != hipSuccess) return -1.0;
    if (hipEventElapsedTime(&ms, g->ev0, g->ev1) != hipSuccess) return -1.0;
    return ms;
}

int lep_gpu_malloc(lep_gpu* g, size_t bytes, void** dptr) { HIPCHK(g, hipSetDevice(g->device)); HIPCHK(g, hipMalloc(dptr, bytes ? bytes : 16)); return 0; }
int lep_gpu_free(lep_gpu* g, void* dptr) { HIPCHK(g, hipFree(dptr)); return 0; }
int lep_gpu_memcpy_h2d(lep_gpu* g, void* dst, const void* src, size_t bytes) { HIPCHK(g, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return 0; }
int lep_gpu_memcpy_d2h(lep_gpu* g, void* dst, const void* src, size_t bytes) { HIPCHK(g, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return 0; }
int lep_gpu_memset(lep_gpu* g, void* dst, int value, size_t bytes) { HIPCHK(g, hipMemset(dst, value, bytes)); return 0; }

// ---- host-buffer variants: stage frames through HBM, run the device path, fetch results --------------
static int stage_images(lep_gpu* g, const lep_image_desc* images, int nimg, std::vector<lep_image_desc>* dev, bool upload) {
    size_t total = 0;
    for (int i = 0; i < nimg; ++i)
        for (int c = 0; c < images[i].ncomp && c < 3; ++c) total += (size_t)images[i].width_blocks[c] * images[i].height_blocks[c] * 128;
    if (int rc = ensure(g, &g->d_blocks, &g->blocks_bytes, total + 256)) return rc;
    dev->assign(images, images + nimg);
    size_t off = 0;
    for (int i = 0; i < nimg; ++i)
        for (int c = 0; c < images[i].ncomp && c < 3; ++c) {
            size_t bytes = (size_t)images[i].width_blocks[c] * images[i].height_blocks[c] * 128;
            (*dev)[i].blocks[c] = (int16_t*)((char*)g->d_blocks + off);
            if (upload) HIPCHK(g, hipMemcpyAsync((char*)g->d_blocks + off, images[i].blocks[c], bytes, hipMemcpyHostToDevice, g->stream));
            else HIPCHK(g, hipMemsetAsync((char*)g->d_blocks + off, 0, bytes, g->stream));
            off += bytes;
        }
    return 0;
}

int lep_gpu_encode_host(lep_gpu* g, const lep_image_desc* images, int nimg, const lep_segment* segs, int nseg, lep_bytes* out,
                        int32_t* status) {
    if (!g) return LEP_GPU_ERROR;
    HIPCHK(g, hipSetDevice(g->device));
    std::vector<lep_image_desc> dev;
    if (int rc = stage_images(g, images, nimg, &dev, true)) return rc;
    std::vector<uint64_t> offs(nseg + 1, 0);
    for (int s = 0; s < nseg; ++s) offs[s + 1] = offs[s] + out[s].cap;
    if (int rc = ensure(g, &g->d_streams, &g->streams_bytes, offs[nseg] + 16)) return rc;
    if (int rc = ensure(g, &g->d_lens, &g->lens_bytes, (size_t)nseg * 8 + 16)) return rc;
    uint32_t* d_len = (uint32_t*)g->d_lens;
    int32_t* d_status = (int32_t*)(d_len + nseg);
    int rc = launch<false>(g, dev.data(), nimg, segs, nseg, (uint8_t*)g->d_streams, offs.data(), d_len, d_status, g->stream);
    if (rc) return rc;
    std::vector<uint32_t> lens(nseg);
    std::vector<int32_t> st(nseg);
    HIPCHK(g, hipMemcpyAsync(lens.data(), d_len, nseg * 4, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(g, hipMemcpyAsync(st.data(), d_status, nseg * 4, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(g, hipStreamSynchronize(g->stream));
    int worst = 0;
    for (int s = 0; s < nseg; ++s) {
        if (status) status[s] = st[s];
        if (st[s] && !worst) worst = st[s];
        out[s].len = st[s] ? 0 : lens[s];
        if (!st[s] && lens[s]) HIPCHK(g, hipMemcpyAsync(out[s].data, (char*)g->d_streams + offs[s], lens[s], hipMemcpyDeviceToHost, g->stream));
    }
    HIPCHK(g, hipStreamSynchronize(g->stream));
    return worst;
}

int lep_gpu_decode_host(lep_gpu* g, const lep_image_desc* images, int nimg, const lep_segment* segs, int nseg,
                        const lep_bytes* in, int32_t* status) {
    if (!g) return LEP_GPU_ERROR;
    HIPCHK(g, hipSetDevice(g->device));
    std::vector<lep_image_desc> dev;
    if (int rc = stage_images(g, images, nimg, &dev, false)) return rc;
    std::vector<uint64_t> offs(nseg + 1, 0);
    std::vector<uint32_t> lens(nseg);
    for (int s = 0; s < nseg; ++s) { offs[s + 1] = offs[s] + in[s].len; lens[s] = (uint32_t)in[s].len; }
    if (int rc = ensure(g, &g->d_streams, &g->streams_bytes, offs[nseg] + 16)) return rc;
    if (int rc = ensure(g, &g->d_lens, &g->lens_bytes, (size_t)nseg * 8 + 16)) return rc;
    uint32_t* d_len = (uint32_t*)g->d_lens;
    int32_t* d_status = (int32_t*)(d_len + nseg);
    for (int s = 0; s < nseg; ++s)
        if (in[s].len) HIPCHK(g, hipMemcpyAsync((char*)g->d_streams + offs[s], in[s].data, in[s].len, hipMemcpyHostToDevice, g->stream));
    HIPCHK(g, hipMemcpyAsync(d_len, lens.data(), nseg * 4, hipMemcpyHostToDevice, g->stream));
    HIPCHK(g, hipStreamSynchronize(g->stream));
    int rc = launch<true>(g, dev.data(), nimg, segs, nseg, (uint8_t*)g->d_streams, offs.data(), d_len, d_status, g->stream);
    if (rc) return rc;
    std::vector<int32_t> st(nseg);
    HIPCHK(g, hipMemcpyAsync(st.data(), d_status, nseg * 4, hipMemcpyDeviceToHost, g->stream));
    for (int i = 0; i < nimg; ++i)
        for (int c = 0; c < images[i].ncomp && c < 3; ++c) {
            size_t bytes = (size_t)images[i].width_blocks[c] * images[i].height_blocks[c] * 128;
            HIPCHK(g, hipMemcpyAsync(images[i].blocks[c], dev[i].blocks[c], bytes, hipMemcpyDeviceToHost, g->stream));
        }
    HIPCHK(g, hipStreamSynchronize(g->stream));
    int worst = 0;
    for (int s = 0; s < nseg; ++s) {
        if (status) status[s] = st[s];
        if (st[s] && !worst) worst = st[s];
    }
    return worst;
}

}  // extern "C"
