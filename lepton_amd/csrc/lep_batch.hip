// lep_batch.hip -- layer 3 of the C ABI for whole batches: `lepton in.jpg out.lep` / `lepton in.lep out.jpg` for many files
// at once, as a pipeline that keeps the GPU hot path fed (BASELINE.json configs[2]: hipStream-overlapped H2D / encode / D2H).
//
//   host pool      parse JPEG (Huffman scan decode, jpeg_scan.cc)  -> coefficient frames in pinned memory
//   copy stream    frames of chunk k+1 go over PCIe while ...
//   compute stream ... the coder kernels of chunk k run (one wavefront per thread segment, lep_gpu.hip)
//                  [optional] on-GPU round-trip verification: decode what was just encoded into a scratch frame and
//                  compare it with the input frame (the reference's default `-verify`, src/lepton/validation.cc:97-218,
//                  costs it a second process and a full decode on the CPU)
//   copy stream    streams of chunk k come back, host pool writes the .lep containers (lep_container.cc)
// and the mirror image for decompression (streams up, frames down, Huffman re-encode on the host pool).
// Only public entry points of the library are used for the GPU work (lep_gpu_encode_device / lep_gpu_decode_device).
#include <hip/hip_runtime.h>
#include <malloc.h>
#include <sched.h>
#include <cstdio>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/lepton_mi355x.h"
#include "jpeg_model.h"
#include "lep_container.h"

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// CPUs this process may really use: affinity mask, capped by the cgroup CPU quota (a container can see 256 CPUs and be
// allowed 16: a pool sized by hardware_concurrency() then runs 3x slower than one sized by the quota)
int effective_cpus() {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, std::min(n, CPU_COUNT(&set)));
    double quota = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota|max> <period>"
        char q[32]; long period = 0;
        if (fscanf(f, "%31s %ld", q, &period) == 2 && q[0] != 'm' && period > 0) quota = atof(q) / (double)period;
        fclose(f);
    } else {
        long q = -1, p = 0;
        if (FILE* a = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(a, "%ld", &q) != 1) q = -1; fclose(a); }
        if (FILE* b = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(b, "%ld", &p) != 1) p = 0; fclose(b); }
        if (q > 0 && p > 0) quota = (double)q / (double)p;
    }
    if (quota >= 1.0) n = std::min(n, (int)(quota + 0.5));
    return std::max(1, n);
}

// run fn(i) for i in [0, n) on `threads` host threads
void parallel_for(int n, int threads, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    threads = std::max(1, std::min(threads, n));
    std::atomic<int> next(0);
    auto worker = [&]() { for (int i; (i = next.fetch_add(1)) < n;) fn(i); };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto& t : pool) t.join();
}

// 16 bytes per lane, grid-stride: frames are compared at HBM speed (2 x bytes read, nothing written unless they differ)
__global__ void lep_compare_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, size_t n16, uint32_t* flag, uint32_t value) {
    bool diff = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 x = a[i], y = b[i];
        diff |= (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
    }
    if (diff) atomicOr(flag, value);
}

// Round-trip check of progressive files: one workgroup per scan compares what the GPU scan encoder has just written
// with the file's own bytes of that scan (uploaded beside the un-stuffed copy the decoder read); any difference -- the
// length, a byte, an encoder that gave up (bit 31 of its length) -- sets bit 1 of the image's flag word.
struct ScanCheck { uint64_t out_off, ref_off; uint32_t ref_len, image; };
__global__ void lep_scan_check_kernel(const uint8_t* __restrict__ out, const uint32_t* __restrict__ out_len, const uint8_t* __restrict__ ref,
                                      const ScanCheck* __restrict__ items, uint32_t* flags) {
    const ScanCheck it = items[blockIdx.x];
    bool diff = out_len[blockIdx.x] != it.ref_len;
    if (!diff) {
        const uint4* a = reinterpret_cast<const uint4*>(out + it.out_off);   // both 16-byte aligned
        const uint4* b = reinterpret_cast<const uint4*>(ref + it.ref_off);
        const uint32_t n16 = it.ref_len / 16;
        for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) {
            const uint4 x = a[i], y = b[i];
            diff |= (x.x != y.x) | (x.y != y.y) | (x.z != y.z) | (x.w != y.w);
        }
        for (uint32_t i = n16 * 16 + threadIdx.x; i < it.ref_len; i += blockDim.x) diff |= out[it.out_off + i] != ref[it.ref_off + i];
    }
    if (diff) atomicOr(flags + it.image, 2u);
}

// the same against reference bytes at any alignment (a baseline file's thread segments, as they stand in the file)
__global__ void lep_scan_check_bytes_kernel(const uint8_t* __restrict__ out, const uint32_t* __restrict__ out_len, const uint8_t* __restrict__ ref,
                                            const ScanCheck* __restrict__ items, uint32_t* flags) {
    const ScanCheck it = items[blockIdx.x];
    bool diff = out_len[blockIdx.x] != it.ref_len;
    if (!diff)
        for (uint32_t i = threadIdx.x; i < it.ref_len; i += blockDim.x) diff |= out[it.out_off + i] != ref[it.ref_off + i];
    if (diff) atomicOr(flags + it.image, 2u);
}

// Hundreds of host threads each allocating and freeing MB-sized vectors (un-stuffed scan data, containers) serialise on
// the process-wide mmap lock when glibc serves them with mmap/munmap; keep such blocks inside the per-thread arenas.
void tune_malloc_for_pool() {
    static std::atomic<bool> done(false);
    if (done.exchange(true)) return;
    mallopt(M_MMAP_THRESHOLD, 256 << 20);
    mallopt(M_TRIM_THRESHOLD, 512 << 20);
    mallopt(M_ARENA_MAX, 512);
}

struct Slot {   // one chunk's buffers (double-buffered)
    char* h_frames = nullptr; char* d_frames = nullptr; char* d_scratch = nullptr; size_t frames_cap = 0, hframes_cap = 0;
    uint8_t* d_streams = nullptr; uint8_t* h_streams = nullptr; size_t streams_cap = 0, hstreams_cap = 0;   // device arena (worst-case sized for encode) / pinned mirror (actual bytes)
    uint32_t* d_len = nullptr; int32_t* d_status = nullptr; uint32_t* d_flags = nullptr; size_t seg_cap = 0, img_cap = 0;
    uint8_t* d_scan = nullptr; uint8_t* h_scan = nullptr; size_t scan_cap = 0;   // JPEG scan bytes of the GPU Huffman encoder
    uint32_t* d_scanlen = nullptr; size_t scanlen_cap = 0;   // nseg byte counts, then (16-byte aligned) nseg lep_huff_end records
    char* d_rows = nullptr; size_t rows_cap = 0;   // row records of the GPU Huffman decoder
    uint8_t* d_pscan = nullptr; size_t pscan_cap = 0;      // progressive files: the scans' device arena (sized for the worst case) ...
    uint8_t* h_pscan = nullptr; size_t hpscan_cap = 0;     // ... and the pinned mirror, packed by the bytes actually written
    uint32_t* d_corr = nullptr; size_t corr_cap = 0;       // held-back correction bits of the refinement scans (dwords)
    uint32_t* d_pscanlen = nullptr; size_t pscanlen_cap = 0;
    ScanCheck* d_pcheck = nullptr; size_t pcheck_cap = 0;   // compression with verify: what lep_scan_check_kernel compares
    uint8_t* d_vscan = nullptr; size_t vscan_cap = 0;       // ... and for baseline files: their scans written again from the device frame,
    uint32_t* d_vscanlen = nullptr; ScanCheck* d_vcheck = nullptr; size_t vseg_cap = 0;   // per thread segment
    hipEvent_t up = nullptr, done = nullptr, decoded = nullptr;
    bool done_recorded = false;   // `done` has been recorded in the current batch call
    void release() {
        if (h_frames) (void)hipHostFree(h_frames);
        if (h_streams) (void)hipHostFree(h_streams);
        if (h_scan) (void)hipHostFree(h_scan);
        if (h_pscan) (void)hipHostFree(h_pscan);
        for (void* p : {(void*)d_vscan, (void*)d_vscanlen, (void*)d_vcheck}) if (p) (void)hipFree(p);
        for (void* p : {(void*)d_pscan, (void*)d_corr, (void*)d_pscanlen, (void*)d_pcheck, (void*)d_frames, (void*)d_scratch, (void*)d_streams, (void*)d_len, (void*)d_status, (void*)d_flags, (void*)d_scan, (void*)d_scanlen, (void*)d_rows})
            if (p) (void)hipFree(p);
        if (up) (void)hipEventDestroy(up);
        if (done) (void)hipEventDestroy(done);
        if (decoded) (void)hipEventDestroy(decoded);
        *this = Slot();
    }
};

#define HIPOK(call) do { if ((call) != hipSuccess) return LEP_GPU_ERROR; } while (0)
// device memory for the pipeline's staging: when the device has none left, what the codec object caches between launches
// (models, the split-phase encoder's scratch -- re-acquired on demand) is given back first
static lep_gpu* g_batch_gpu = nullptr;   // the codec object of the batch call in progress
static hipError_t dev_alloc(void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && g_batch_gpu) {
        (void)hipGetLastError();
        (void)lep_gpu_release_memory(g_batch_gpu);
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) *p = nullptr;
    return e;
}

struct Joiner {   // a background thread that is joined on every way out of the function, error returns included
    std::thread t;
    ~Joiner() { if (t.joinable()) t.join(); }
};

struct StreamSet {   // HIP streams of one batch call, destroyed on every way out
    std::vector<hipStream_t> all;
    lep_gpu* g = nullptr;   // the codec whose launch functions were handed these streams: its upload ring forgets them first
    int make(hipStream_t* s, int priority = 0, bool with_priority = false) {
        const hipError_t e = with_priority ? hipStreamCreateWithPriority(s, hipStreamNonBlocking, priority) : hipStreamCreateWithFlags(s, hipStreamNonBlocking);
        if (e != hipSuccess) return LEP_GPU_ERROR;
        all.push_back(*s);
        return 0;
    }
    ~StreamSet() {
        for (hipStream_t s : all) (void)hipStreamSynchronize(s);
        if (g) (void)lep_gpu_settle_uploads(g);
        for (hipStream_t s : all) (void)hipStreamDestroy(s);
    }
};
template <class T, void (*CLOSE)(T*)>
struct HandleVector {   // parsed files of a batch: whatever is still open when the call returns is closed
    std::vector<T*> v;
    explicit HandleVector(size_t n) : v(n, nullptr) {}
    ~HandleVector() { for (T* p : v) if (p) CLOSE(p); }
    T*& operator[](size_t i) { return v[i]; }
};

double g_alloc_s = 0;   // time spent in (re)allocating staging buffers during the current call (single orchestrator thread)

int slot_reserve_impl(Slot* s, size_t frames, size_t streams, size_t nseg, size_t nimg, bool scratch, size_t host_streams);
// host_streams: bytes of the pinned mirror of the stream arena (decompress: all of it; compress: reserved later, once the
// encoder has reported how many bytes it really wrote -- the arena itself is sized for the worst case, 5x more)
int slot_reserve(Slot* s, size_t frames, size_t streams, size_t nseg, size_t nimg, bool scratch, size_t host_streams) {
    const double t0 = now_s();
    const int rc = slot_reserve_impl(s, frames, streams, nseg, nimg, scratch, host_streams);
    g_alloc_s += now_s() - t0;
    return rc;
}
int slot_reserve_impl(Slot* s, size_t frames, size_t streams, size_t nseg, size_t nimg, bool scratch, size_t host_streams) {
    if (frames > s->frames_cap) {
        if (s->d_frames) (void)hipFree(s->d_frames);
        if (s->d_scratch) (void)hipFree(s->d_scratch);
        s->d_frames = s->d_scratch = nullptr; s->frames_cap = 0;   // (a capacity never outlives its allocation: the next hipMalloc may fail)
        HIPOK(dev_alloc((void**)&s->d_frames, frames));
        s->frames_cap = frames;
    }
    if (scratch && !s->d_scratch) HIPOK(dev_alloc((void**)&s->d_scratch, s->frames_cap));
    if (streams > s->streams_cap) {
        if (s->d_streams) (void)hipFree(s->d_streams);
        s->d_streams = nullptr; s->streams_cap = 0;
        HIPOK(dev_alloc((void**)&s->d_streams, streams));
        s->streams_cap = streams;
    }
    if (host_streams > s->hstreams_cap) {
        if (s->h_streams) (void)hipHostFree(s->h_streams);
        s->h_streams = nullptr; s->hstreams_cap = 0;
        const size_t want = host_streams + host_streams / 4;   // head room: the next chunk of similar files should not reallocate
        HIPOK(hipHostMalloc((void**)&s->h_streams, want, hipHostMallocDefault));
        s->hstreams_cap = want;
    }
    if (nseg > s->seg_cap) {
        if (s->d_len) (void)hipFree(s->d_len);
        if (s->d_status) (void)hipFree(s->d_status);
        s->d_len = nullptr; s->d_status = nullptr; s->seg_cap = 0;
        HIPOK(dev_alloc((void**)&s->d_len, nseg * 4));
        HIPOK(dev_alloc((void**)&s->d_status, nseg * 8));   // [nseg] coder statuses + [nseg] verification-decode statuses
        s->seg_cap = nseg;
    }
    if (nimg > s->img_cap) {
        if (s->d_flags) (void)hipFree(s->d_flags);
        s->d_flags = nullptr; s->img_cap = 0;
        HIPOK(dev_alloc((void**)&s->d_flags, nimg * 4));
        s->img_cap = nimg;
    }
    if (!s->up) HIPOK(hipEventCreateWithFlags(&s->up, hipEventDisableTiming));
    if (!s->done) HIPOK(hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
    if (!s->decoded) HIPOK(hipEventCreateWithFlags(&s->decoded, hipEventDisableTiming));
    return 0;
}

// the end-state records live behind the byte counts of the same allocation (16-byte aligned)
lep_huff_end* huff_ends(Slot* s) { return (lep_huff_end*)((char*)s->d_scanlen + ((s->scanlen_cap * 4 + 15) & ~(size_t)15)); }

int scan_reserve(Slot* s, size_t bytes, size_t nseg) {
    const double t0 = now_s();
    if (bytes > s->scan_cap) {
        if (s->h_scan) (void)hipHostFree(s->h_scan);
        if (s->d_scan) (void)hipFree(s->d_scan);
        s->h_scan = s->d_scan = nullptr; s->scan_cap = 0;
        HIPOK(hipHostMalloc((void**)&s->h_scan, bytes, hipHostMallocDefault));
        HIPOK(dev_alloc((void**)&s->d_scan, bytes));
        s->scan_cap = bytes;
    }
    if (nseg > s->scanlen_cap) {
        if (s->d_scanlen) (void)hipFree(s->d_scanlen);
        s->d_scanlen = nullptr; s->scanlen_cap = 0;
        HIPOK(dev_alloc((void**)&s->d_scanlen, nseg * 4 + 16 + nseg * sizeof(lep_huff_end)));
        s->scanlen_cap = nseg;
    }
    g_alloc_s += now_s() - t0;
    return 0;
}

int prog_reserve(Slot* s, size_t scan_bytes, size_t corr_words, size_t nscan) {
    const double t0 = now_s();
    if (scan_bytes > s->pscan_cap) {
        if (s->d_pscan) (void)hipFree(s->d_pscan);
        s->d_pscan = nullptr; s->pscan_cap = 0;
        HIPOK(dev_alloc((void**)&s->d_pscan, scan_bytes));
        s->pscan_cap = scan_bytes;
    }
    if (corr_words > s->corr_cap) {
        if (s->d_corr) (void)hipFree(s->d_corr);
        s->d_corr = nullptr; s->corr_cap = 0;
        HIPOK(dev_alloc((void**)&s->d_corr, corr_words * 4));
        s->corr_cap = corr_words;
    }
    if (nscan > s->pscanlen_cap) {
        if (s->d_pscanlen) (void)hipFree(s->d_pscanlen);
        s->d_pscanlen = nullptr; s->pscanlen_cap = 0;
        HIPOK(dev_alloc((void**)&s->d_pscanlen, nscan * 4));
        s->pscanlen_cap = nscan;
    }
    if (nscan > s->pcheck_cap) {
        if (s->d_pcheck) (void)hipFree(s->d_pcheck);
        s->d_pcheck = nullptr; s->pcheck_cap = 0;
        HIPOK(dev_alloc((void**)&s->d_pcheck, nscan * sizeof(ScanCheck)));
        s->pcheck_cap = nscan;
    }
    g_alloc_s += now_s() - t0;
    return 0;
}
int vscan_reserve(Slot* s, size_t bytes, size_t nseg) {
    const double t0 = now_s();
    if (bytes > s->vscan_cap) {
        if (s->d_vscan) (void)hipFree(s->d_vscan);
        s->d_vscan = nullptr; s->vscan_cap = 0;
        HIPOK(dev_alloc((void**)&s->d_vscan, bytes));
        s->vscan_cap = bytes;
    }
    if (nseg > s->vseg_cap) {
        if (s->d_vscanlen) (void)hipFree(s->d_vscanlen);
        if (s->d_vcheck) (void)hipFree(s->d_vcheck);
        s->d_vscanlen = nullptr; s->d_vcheck = nullptr; s->vseg_cap = 0;
        HIPOK(dev_alloc((void**)&s->d_vscanlen, nseg * 4));
        HIPOK(dev_alloc((void**)&s->d_vcheck, nseg * sizeof(ScanCheck)));
        s->vseg_cap = nseg;
    }
    g_alloc_s += now_s() - t0;
    return 0;
}
int prog_host_reserve(Slot* s, size_t bytes) {
    if (bytes <= s->hpscan_cap) return 0;
    const double t0 = now_s();
    if (s->h_pscan) (void)hipHostFree(s->h_pscan);
    s->h_pscan = nullptr; s->hpscan_cap = 0;
    const size_t want = bytes + bytes / 4;
    HIPOK(hipHostMalloc((void**)&s->h_pscan, want, hipHostMallocDefault));
    s->hpscan_cap = want;
    g_alloc_s += now_s() - t0;
    return 0;
}

// pinned host staging for whole frames: only needed for files the host Huffman coder handles
int host_frames_reserve(Slot* s, size_t frames) {
    if (frames <= s->hframes_cap) return 0;
    const double t0 = now_s();
    if (s->h_frames) (void)hipHostFree(s->h_frames);
    s->h_frames = nullptr; s->hframes_cap = 0;
    HIPOK(hipHostMalloc((void**)&s->h_frames, frames, hipHostMallocDefault));
    s->hframes_cap = frames;
    g_alloc_s += now_s() - t0;
    return 0;
}

struct Chunk {
    int first = 0, count = 0;
    std::vector<int> live;                 // indices (into the batch) of the images that go to the GPU
    std::vector<lep_image_desc> host_desc, dev_desc, scratch_desc;
    std::vector<size_t> frame_off;         // per live image: offset of its frame in the slot
    std::vector<lep_segment> segs;
    std::vector<uint64_t> offs;            // stream arena offsets (nseg + 1)
    std::vector<int> seg_first;            // per live image: first segment index
    size_t frame_bytes = 0;
    // decompression: GPU Huffman re-encode plan
    std::vector<lep_huff_image> himg;      // eligible images only
    std::vector<lep_huff_segment> hseg;
    std::vector<int> hfirst;               // per live image: first entry of hseg, -1 = host re-coder
    std::vector<uint32_t> hslot;           // per hseg: bytes reserved in the scan arena
    std::vector<uint32_t> hbound;          // per hseg: the segment's real byte bound (larger than the slot for segment 0)
    size_t scan_bytes = 0;
    // decompression of progressive files: one GPU wavefront per (image, scan) (lep_huffprog.h)
    std::vector<lep_huffprog_image> pimg;
    std::vector<lep_huffprog_scan> pscan;
    std::vector<int> pfirst, pcount;       // per live image: first entry of pscan / number of scans, -1 = not on this path
    size_t pscan_bytes = 0, corr_words = 0;
    std::vector<ScanCheck> pcheck;         // compression with verify: per pscan entry, the file's own bytes of that scan
    // compression with verify, baseline files the GPU decoded: the Huffman half of the round trip (their scans written again)
    std::vector<lep_huff_image> vimg;
    std::vector<lep_huff_segment> vseg;
    std::vector<ScanCheck> vcheck;
    std::vector<char> vchecked;            // per live image: its scan goes through that check
    size_t vscan_bytes = 0;
    std::vector<char> pchecked;            // per live image: its scans go through the progressive check (one flag per image: the writer
                                           // used to scan `pcheck` for every image, images x scans comparisons per chunk -- ADVICE round 3)
};

Slot g_slots[2];         // one batch call at a time (the calls are not re-entrant)

// exact bytes of a coefficient frame (a multiple of 128); a frame's ROOM in a slot is this rounded up to 256
size_t frame_exact_bytes(const lep_image_desc& d) {
    size_t b = 0;
    for (int c = 0; c < d.ncomp; ++c) b += (size_t)d.width_blocks[c] * d.height_blocks[c] * 128;
    return b;
}
size_t frame_bytes_of(const lep_image_desc& d) { return (frame_exact_bytes(d) + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

// frees the pinned / device staging buffers the batch calls keep between invocations
void lep_batch_release(void) { for (Slot& s : g_slots) s.release(); }

// what the batch calls keep between invocations: bytes of pinned host memory, and of device memory in the large arenas
// (frames, scratch frames, stream / scan arenas, row records; the per-segment tables are KBs and not counted)
void lep_batch_footprint(size_t* pinned_bytes, size_t* device_bytes) {
    size_t pinned = 0, device = 0;
    for (const Slot& s : g_slots) {
        pinned += (s.h_frames ? s.hframes_cap : 0) + (s.h_streams ? s.hstreams_cap : 0) + (s.h_scan ? s.scan_cap : 0) + (s.h_pscan ? s.hpscan_cap : 0);
        device += (s.d_frames ? s.frames_cap : 0) + (s.d_scratch ? s.frames_cap : 0) + (s.d_streams ? s.streams_cap : 0) + (s.d_scan ? s.scan_cap : 0)
                  + (s.d_rows ? s.rows_cap : 0) + (s.d_pscan ? s.pscan_cap : 0) + (s.d_corr ? s.corr_cap * 4 : 0) + (s.d_vscan ? s.vscan_cap : 0);
    }
    if (pinned_bytes) *pinned_bytes = pinned;
    if (device_bytes) *device_bytes = device;
}

// test hook: fills every pinned staging buffer the batch calls keep between invocations with `value`, so that a test can
// show that nothing stale from an earlier batch (the padding between frames, the tails of streams) reaches a result
void lep_batch_debug_poison(int value) {
    for (Slot& s : g_slots) {
        if (s.h_frames) memset(s.h_frames, value, s.hframes_cap);
        if (s.h_streams) memset(s.h_streams, value, s.hstreams_cap);
        if (s.h_scan) memset(s.h_scan, value, s.scan_cap);
    }
}

// ---- JPEG -> .lep, batch ---------------------------------------------------------------------------------------------
int lep_compress_batch(lep_gpu* g, const lep_bytes* jpgs, int n, lep_bytes* outs, int32_t* status, const lep_batch_options* o,
                       lep_batch_stats* stats) {
    if (!g || n < 0) return LEP_GPU_ERROR;
    g_batch_gpu = g;
    const int threads = o && o->host_threads > 0 ? o->host_threads : effective_cpus();
    // With the Huffman decode on the GPU, chunk k+1 is decoded (lep_huffdec_simt.h, one lane per piece of the scan, on its own stream)
    // WHILE the split-phase encoder's kernels of chunk k run.  The chunking itself is lep_batch_plan (lep_api.cc).
    const bool verify = o && o->verify;
    HIPOK(hipSetDevice(lep_gpu_device(g)));
    tune_malloc_for_pool();
    const double t_begin = now_s();
    lep_batch_stats st;
    memset(&st, 0, sizeof st);
    for (int i = 0; i < n; ++i) { outs[i].data = nullptr; outs[i].len = outs[i].cap = 0; status[i] = 0; }
    // 1. frame sizes from the SOF markers -> the whole batch is cut into chunks before anything is decoded (lep_batch_plan),
    //    so that every image can be Huffman-decoded straight into its place in a staging buffer (no page faults, no second copy)
    std::vector<size_t> fbytes(n, 0), jbytes(n, 0);
    parallel_for(n, threads, [&](int i) { jbytes[i] = jpgs[i].len; if (int rc = lep_jpeg_peek_frame_bytes(jpgs[i].data, jpgs[i].len, &fbytes[i])) { status[i] = rc; fbytes[i] = 0; } });
    std::vector<int> first((size_t)n + 2, 0);
    const int nchunks = lep_batch_plan(jbytes.data(), fbytes.data(), n, o, first.data(), n + 2);
    if (nchunks < 0) return LEP_ASSERTION_FAILURE;
    std::vector<std::unique_ptr<Chunk>> chunks;
    for (int k = 0; k < nchunks; ++k) {
        std::unique_ptr<Chunk> c(new Chunk);
        c->first = first[k];
        size_t bytes = 0;
        for (int i = first[k]; i < first[k + 1]; ++i) {
            if (status[i]) continue;
            c->live.push_back(i); c->frame_off.push_back(bytes);
            bytes += (fbytes[i] + 255) & ~(size_t)255;
        }
        c->count = first[k + 1] - first[k];
        c->frame_bytes = bytes;
        chunks.push_back(std::move(c));
    }
    hipStream_t s_copy = nullptr, s_compute = nullptr, s_compute2 = nullptr, s_down = nullptr, s_huff = nullptr;
    // overlap_launches: chunk k+1's coder kernel goes to a second stream (and the library's second workspace set), so its
    // wavefronts start in the slots that chunk k's long thread segments leave free instead of waiting for the last of them
    // (real photographs: segments of equal compressed size differ several-fold in blocks).  Off by default until measured.
    const bool overlap = (o && o->overlap_launches) || (getenv("LEP_BATCH_OVERLAP") && atoi(getenv("LEP_BATCH_OVERLAP")) != 0);
    StreamSet stream_set;
    stream_set.g = g;
    if (stream_set.make(&s_copy) || stream_set.make(&s_compute) || (overlap && stream_set.make(&s_compute2)) || stream_set.make(&s_down)) return LEP_GPU_ERROR;
    {   // Huffman decode of chunk k+1 beside the coder kernels of chunk k: its workgroups go first whenever a slot is free
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (stream_set.make(&s_huff, hi, true)) return LEP_GPU_ERROR;
    }
    Slot* slots = g_slots;   // grow-only staging cache shared by the batch calls (lep_batch_release frees it)
    g_alloc_s = 0;
    int rc_all = 0, st_redone = 0;
    HandleVector<lep_jpeg, lep_jpeg_close> parsed(n);
    std::vector<char> host_parsed(n, 0);   // files the host parser took (irregular scans): verify also re-codes them on the host

    // 2. per chunk: split the files on the host pool; eligible scans are Huffman-decoded ON THE GPU straight into the device
    //    frame (only the 2 MB of scan bytes cross PCIe), the others by the host parser into the slot's pinned frames; then the
    //    hand-offs are finished, segments planned and the upload event recorded
    const bool gpu_huffman = !(o && o->host_huffman);
    auto host_parse_one = [&](Chunk* c, Slot* s, int k) -> int {   // host Huffman decode of live image k into its pinned frame
        const int i = c->live[k];
        const size_t room = (k + 1 < (int)c->live.size() ? c->frame_off[k + 1] : c->frame_bytes) - c->frame_off[k];
        if (parsed[i]) { lep_jpeg_close(parsed[i]); parsed[i] = nullptr; }
        int rc = lep_jpeg_open_into(jpgs[i].data, jpgs[i].len, 1, s->h_frames + c->frame_off[k], room, &parsed[i]);
        host_parsed[i] = 1;
        if (!rc) {
            lep_jpeg_describe(parsed[i], &c->host_desc[k]);
            const lep_image_desc& d = c->host_desc[k];
            if ((char*)d.blocks[0] != s->h_frames + c->frame_off[k]) {   // SOF peek disagreed with the parser: copy into place
                if (frame_bytes_of(d) > room) rc = LEP_ASSERTION_FAILURE;
                else {
                    size_t off = c->frame_off[k];
                    for (int cc = 0; cc < d.ncomp; ++cc) {
                        const size_t b = (size_t)d.width_blocks[cc] * d.height_blocks[cc] * 128;
                        memcpy(s->h_frames + off, d.blocks[cc], b);
                        off += b;
                    }
                }
            }
        }
        if (rc) { status[i] = rc; if (parsed[i]) { lep_jpeg_close(parsed[i]); parsed[i] = nullptr; } }
        return rc;
    };
    auto parse_and_upload = [&](Chunk* c, Slot* s) -> int {
        if (c->live.empty()) return 0;
        if (int rc = slot_reserve(s, c->frame_bytes, 0, 0, c->live.size(), verify, 0)) return rc;
        const int nl = (int)c->live.size();
        double t0 = now_s();
        c->host_desc.assign(nl, lep_image_desc());
        std::vector<lep_huffdec_image> himg(nl);
        std::vector<char> on_gpu(nl, 0);
        std::vector<char> need_host(nl, gpu_huffman ? 0 : 1);
        // progressive files: their scans go to the GPU scan decoder too (lep_huffprogdec.h).  When the caller wants the round
        // trip verified, the Huffman half of that check is made on the GPU as well: every scan of the frame is written again
        // (lep_huffprog.h, the decompressor's kernel) and compared with the file's own bytes (no "canonical by construction"
        // argument is made for these scans, unlike the sequential kernel's)
        std::vector<std::vector<lep_huffprogdec_scan>> pscans(nl);
        std::vector<int> prow_need(nl, 0);
        std::vector<char> on_prog(nl, 0);
        struct ProgCheck { lep_huffprog_image img; std::vector<lep_huffprog_scan> scans; std::vector<uint32_t> first, len; };
        std::vector<ProgCheck> pchk(verify ? nl : 0);
        if (gpu_huffman)
            parallel_for(nl, threads, [&](int k) {
                const int i = c->live[k];
                int ok = 0;
                int rc = lep_jpeg_open_gpu(jpgs[i].data, jpgs[i].len, &parsed[i], &himg[k], &ok);
                if (rc) { status[i] = rc; return; }      // not a JPEG the reference would take either
                if (ok) { on_gpu[k] = 1; return; }
                {
                    pscans[k].resize(64);
                    int ns = 0, okp = 0;
                    if (!lep_jpeg_open_gpu_progressive(parsed[i], pscans[k].data(), 64, &ns, &prow_need[k], &okp) && okp) {
                        pscans[k].resize((size_t)ns);
                        if (!verify) { on_prog[k] = 1; return; }
                        ProgCheck& pc = pchk[k];
                        pc.scans.resize((size_t)ns); pc.first.resize((size_t)ns); pc.len.resize((size_t)ns);
                        int nc = 0, okc = 0;
                        if (!lep_jpeg_plan_progressive_check(parsed[i], jpgs[i].len, &pc.img, pc.scans.data(), pc.first.data(), pc.len.data(), ns, &nc, &okc) &&
                            okc && nc == ns) { on_prog[k] = 1; return; }
                    }
                    pscans[k].clear();
                }
                need_host[k] = 1;
            });
        bool any_host = false;
        for (int k = 0; k < nl; ++k) any_host |= need_host[k] != 0;
        if (any_host) {
            if (int rc = host_frames_reserve(s, c->frame_bytes)) return rc;
            parallel_for(nl, threads, [&](int k) { if (need_host[k]) host_parse_one(c, s, k); });
        }
        // scan arena (pinned -> device), row records
        std::vector<size_t> scan_off(nl, 0), row_off(nl, 0);
        size_t scan_total = 0, rows_total = 0;
        int ngpu = 0;
        for (int k = 0; k < nl; ++k) if (on_gpu[k]) {
            scan_off[k] = scan_total; scan_total += LEP_HUFFDEC_SCAN_ROOM(himg[k].scan_len);
            if (himg[k].flags & LEP_HUFFDEC_RST_TABLE) scan_total += ((size_t)((himg[k].mcuc - 1) / himg[k].rsti) * 4 + 15) & ~(size_t)15;   // the restart positions behind the scan bytes
            row_off[k] = rows_total; rows_total += (size_t)himg[k].mcuv + 1;
            ++ngpu;
        }
        std::vector<std::vector<size_t>> pscan_off(nl);   // progressive: every scan in its own aligned, zero-padded slot
        int nprog = 0;
        for (int k = 0; k < nl; ++k) if (on_prog[k]) {
            for (const lep_huffprogdec_scan& sc : pscans[k]) { pscan_off[k].push_back(scan_total); scan_total += ((size_t)sc.t.scan_len + 80 + 15) & ~(size_t)15; }
            row_off[k] = rows_total; rows_total += (size_t)prow_need[k];
            ++nprog;
        }
        std::vector<std::vector<size_t>> praw_off(nl);   // ... and, for the round-trip check, as it stands in the file
        if (verify)
            for (int k = 0; k < nl; ++k) if (on_prog[k])
                for (uint32_t n : pchk[k].len) { praw_off[k].push_back(scan_total); scan_total += ((size_t)n + 15) & ~(size_t)15; }
        std::vector<size_t> vraw_off(nl, 0);             // baseline files: the whole scan as it stands in the file, for the same check
        std::vector<uint32_t> vraw_first(nl, 0), vraw_len(nl, 0);
        if (verify)
            for (int k = 0; k < nl; ++k) if (on_gpu[k] && !lep_jpeg_scan_file_range(parsed[c->live[k]], &vraw_first[k], &vraw_len[k])) {
                vraw_off[k] = scan_total; scan_total += ((size_t)vraw_len[k] + 15) & ~(size_t)15;
            }
        ngpu += nprog;
        std::vector<lep_huffdec_row> rows(rows_total);
        if (ngpu) {
            if (int rc = scan_reserve(s, scan_total + 256, 0)) return rc;
            const double ta = now_s();
            if (rows_total * sizeof(lep_huffdec_row) > s->rows_cap) {
                if (s->d_rows) (void)hipFree(s->d_rows);
                s->d_rows = nullptr; s->rows_cap = 0;
                HIPOK(dev_alloc((void**)&s->d_rows, rows_total * sizeof(lep_huffdec_row)));
                s->rows_cap = rows_total * sizeof(lep_huffdec_row);
            }
            g_alloc_s += now_s() - ta;
            parallel_for(nl, threads, [&](int k) {
                if (!on_gpu[k] && !on_prog[k]) return;
                const uint8_t* p = nullptr; size_t len = 0;
                lep_jpeg_scan_bytes(parsed[c->live[k]], &p, &len);
                if (on_prog[k]) {
                    for (size_t q = 0; q < pscans[k].size(); ++q) {
                        const size_t off = (size_t)(uintptr_t)pscans[k][q].t.scan, n = pscans[k][q].t.scan_len, room = ((n + 80 + 15) & ~(size_t)15);
                        memcpy(s->h_scan + pscan_off[k][q], p + off, n);
                        memset(s->h_scan + pscan_off[k][q] + n, 0, room - n);
                    }
                    for (size_t q = 0; q < praw_off[k].size(); ++q)
                        memcpy(s->h_scan + praw_off[k][q], jpgs[c->live[k]].data + pchk[k].first[q], pchk[k].len[q]);
                    return;
                }
                memcpy(s->h_scan + scan_off[k], p, len);
                memset(s->h_scan + scan_off[k] + len, 0, LEP_HUFFDEC_SCAN_ROOM(himg[k].scan_len) - len);
                if (himg[k].flags & LEP_HUFFDEC_RST_TABLE) {
                    const uint32_t* rp = nullptr; size_t rn = 0;
                    lep_jpeg_scan_restarts(parsed[c->live[k]], &rp, &rn);
                    memcpy(s->h_scan + scan_off[k] + LEP_HUFFDEC_SCAN_ROOM(himg[k].scan_len), rp, rn * 4);
                }
                if (vraw_len[k]) memcpy(s->h_scan + vraw_off[k], jpgs[c->live[k]].data + vraw_first[k], vraw_len[k]);
            });
        }
        st.parse_s += now_s() - t0;
        // uploads: zero frames for the GPU-decoded images, host-decoded frames as they are
        HIPOK(hipMemsetAsync(s->d_frames, 0, c->frame_bytes, s_copy));
        // (exactly the frame: the parser zeroes and fills frame_exact_bytes only, the rest of the 256-byte-rounded room in the
        // pinned staging is whatever an earlier batch left there and must not reach the device, where the padding stays zero)
        for (int k = 0; k < nl; ++k) if (!on_gpu[k] && !on_prog[k] && parsed[c->live[k]]) {
            const size_t fb = frame_exact_bytes(c->host_desc[k]);
            HIPOK(hipMemcpyAsync(s->d_frames + c->frame_off[k], s->h_frames + c->frame_off[k], fb, hipMemcpyHostToDevice, s_copy));
            st.h2d_bytes += (double)fb;
        }
        if (ngpu) {
            HIPOK(hipMemcpyAsync(s->d_scan, s->h_scan, scan_total, hipMemcpyHostToDevice, s_copy));
            st.h2d_bytes += (double)scan_total;
            HIPOK(hipEventRecord(s->up, s_copy));
            // Huffman scan decode on the GPU: one wavefront per image
            std::vector<lep_huffdec_image> launch;
            std::vector<int> which;
            for (int k = 0; k < nl; ++k) if (on_gpu[k]) {
                lep_huffdec_image hi = himg[k];
                hi.scan = s->d_scan + scan_off[k];
                hi.rows_off = row_off[k];
                size_t off = c->frame_off[k];
                for (int cc = 0; cc < hi.ncomp; ++cc) {
                    hi.blocks[cc] = (int16_t*)(s->d_frames + off);
                    off += (size_t)hi.bch[cc] * hi.vs[cc] * hi.mcuv * 128;
                }
                launch.push_back(hi); which.push_back(k);
            }
            HIPOK(hipStreamWaitEvent(s_huff, s->up, 0));
            // One LANE per subsequence (lep_huffdec_simt.h: thousands of subsequences per scan, 64 codes per instruction) for the scans
            // without restart intervals, the single-wave kernel (lep_huffdec.h) for the others; LEP_HUFFDEC_SIMT=0 keeps the single-wave
            // kernel for every file.  (Round 2-4's form in between -- several WAVEFRONTS per image, 290 ms per 896-file chunk against 43 --
            // is gone from the product: LAB_NOTES.md 4 "JPEG Huffman kernels" keeps its numbers.)
            const bool simt = !(getenv("LEP_HUFFDEC_SIMT") && atoi(getenv("LEP_HUFFDEC_SIMT")) == 0);
            if (simt) {
                std::vector<lep_huffdec_image> many, one;
                for (const lep_huffdec_image& hi : launch) ((hi.rsti && !(hi.flags & LEP_HUFFDEC_RST_TABLE)) ? one : many).push_back(hi);
                if (!many.empty()) { if (int rc = lep_gpu_huffman_decode_simt_device(g, many.data(), (int)many.size(), (lep_huffdec_row*)s->d_rows, s_huff)) return rc; }
                if (!one.empty()) { if (int rc = lep_gpu_huffman_decode_device(g, one.data(), (int)one.size(), (lep_huffdec_row*)s->d_rows, s_huff)) return rc; }
            } else
            if (int rc = lep_gpu_huffman_decode_device(g, launch.data(), (int)launch.size(), (lep_huffdec_row*)s->d_rows, s_huff)) return rc;
            // progressive files: one wavefront per (image, scan), launched dependency level by dependency level
            std::vector<lep_huffprogdec_scan> plaunch;
            std::vector<size_t> pfirst_desc(nl, 0);
            for (int k = 0; k < nl; ++k) if (on_prog[k]) {
                pfirst_desc[k] = plaunch.size();
                for (size_t q = 0; q < pscans[k].size(); ++q) {
                    lep_huffprogdec_scan sc = pscans[k][q];
                    sc.t.scan = s->d_scan + pscan_off[k][q];
                    size_t off = c->frame_off[k];
                    for (int cc = 0; cc < sc.t.ncomp; ++cc) {
                        sc.t.blocks[cc] = (int16_t*)(s->d_frames + off);
                        off += (size_t)sc.t.bch[cc] * sc.bcv[cc] * 128;
                    }
                    sc.t.rows_off += row_off[k];
                    sc.result_off += row_off[k];
                    plaunch.push_back(sc);
                }
            }
            if (!plaunch.empty()) { if (int rc = lep_gpu_huffman_progressive_decode_device(g, plaunch.data(), (int)plaunch.size(), (lep_huffdec_row*)s->d_rows, s_huff)) return rc; }
            HIPOK(hipMemcpyAsync(rows.data(), s->d_rows, rows_total * sizeof(lep_huffdec_row), hipMemcpyDeviceToHost, s_huff));
            HIPOK(hipStreamSynchronize(s_huff));
            st.d2h_bytes += (double)(rows_total * sizeof(lep_huffdec_row));
            if (simt) {
                // a scan whose subsequences did not synchronise (or that is irregular) gets a second chance with the single-wave
                // kernel before the host parser is bothered: its frame is wiped first (pass C may have written part of it)
                std::vector<lep_huffdec_image> again;
                for (const lep_huffdec_image& hi : launch) {
                    if ((hi.rsti && !(hi.flags & LEP_HUFFDEC_RST_TABLE)) || ((rows[hi.rows_off + (size_t)hi.mcuv].aux >> 8) & 0x3fffff) == 0) continue;   // (bit 30: LEP_HUFFDEC_ROW_TRUNCATED, not a status)
                    for (int cc = 0; cc < hi.ncomp; ++cc)
                        HIPOK(hipMemsetAsync(hi.blocks[cc], 0, (size_t)hi.bch[cc] * hi.vs[cc] * hi.mcuv * 128, s_huff));
                    again.push_back(hi);
                }
                if (!again.empty()) {
                    if (int rc = lep_gpu_huffman_decode_device(g, again.data(), (int)again.size(), (lep_huffdec_row*)s->d_rows, s_huff)) return rc;
                    HIPOK(hipMemcpyAsync(rows.data(), s->d_rows, rows_total * sizeof(lep_huffdec_row), hipMemcpyDeviceToHost, s_huff));
                    HIPOK(hipStreamSynchronize(s_huff));
                }
            }
            // hand-offs from the row records; irregular scans go back to the host parser (and their frames up again)
            t0 = now_s();
            std::vector<char> redo(nl, 0);
            parallel_for(nl, threads, [&](int k) {
                const int i = c->live[k];
                if (on_prog[k]) {
                    if (lep_jpeg_finish_gpu_progressive(parsed[i], plaunch.data() + pfirst_desc[k], (int)pscans[k].size(), rows.data())) { redo[k] = 1; on_prog[k] = 0; return; }
                    lep_jpeg_describe(parsed[i], &c->host_desc[k]);
                    return;
                }
                if (!on_gpu[k]) return;
                if (lep_jpeg_finish_gpu(parsed[i], rows.data() + row_off[k])) { redo[k] = 1; on_gpu[k] = 0; return; }
                lep_jpeg_describe(parsed[i], &c->host_desc[k]);   // geometry; the frame itself only exists on the device
            });
            bool any_redo = false;
            for (int k = 0; k < nl; ++k) any_redo |= redo[k] != 0;
            if (any_redo) {
                if (int rc = host_frames_reserve(s, c->frame_bytes)) return rc;
                parallel_for(nl, threads, [&](int k) { if (redo[k]) host_parse_one(c, s, k); });
            }
            st.parse_s += now_s() - t0;
            for (int k = 0; k < nl; ++k) if (redo[k] && parsed[c->live[k]]) {
                const size_t fb = frame_exact_bytes(c->host_desc[k]);   // the whole frame: the GPU decoder may have written part of it
                HIPOK(hipMemcpyAsync(s->d_frames + c->frame_off[k], s->h_frames + c->frame_off[k], fb, hipMemcpyHostToDevice, s_copy));
                st.h2d_bytes += (double)fb;
            }
        }
        for (int k = 0; k < nl; ++k) if ((on_gpu[k] || on_prog[k]) && parsed[c->live[k]]) st.gpu_huffman_files += 1;
        // drop failed images from the chunk
        Chunk keep;
        std::vector<int> old_k;
        for (int k = 0; k < nl; ++k) {
            const int i = c->live[k];
            if (!parsed[i]) continue;
            keep.live.push_back(i); keep.frame_off.push_back(c->frame_off[k]); keep.host_desc.push_back(c->host_desc[k]);
            old_k.push_back(k);
        }
        c->live.swap(keep.live); c->frame_off.swap(keep.frame_off); c->host_desc.swap(keep.host_desc);
        c->segs.clear(); c->offs.assign(1, 0); c->seg_first.clear();
        for (size_t k = 0; k < c->live.size(); ++k) {
            lep_segment sg[LEP_MAX_SEGMENTS];
            lep_handoff ho[LEP_MAX_SEGMENTS];
            const int ns = lep_jpeg_plan(parsed[c->live[k]], 0, sg, (int)k);
            if (lep_jpeg_plan_handoffs(parsed[c->live[k]], 0, ho, LEP_MAX_SEGMENTS) != ns || ns < 0) return LEP_ASSERTION_FAILURE;   // (the two plans are one choice; ho[] is read below)
            c->seg_first.push_back((int)c->segs.size());
            const lep_image_desc& d = c->host_desc[k];
            size_t blocks = 0;
            for (int cc = 0; cc < d.ncomp; ++cc) blocks += (size_t)d.width_blocks[cc] * d.height_blocks[cc];
            for (int q = 0; q < ns; ++q) {
                // stream space: segments are cut by equal JPEG bytes, not blocks, so the segment's own scan bytes (+ 25 %)
                // are the measure; the per-block term covers progressive files, whose hand-offs only count the first scan.
                // What still overflows is redone per file (end of this function).
                // (baseline files take the byte measure alone: the per-block term made a 4K segment's slot 1 MB for 0.2 MB of stream,
                // and the whole arena comes down in one copy)
                const bool prog = lep_jpeg_is_progressive(parsed[c->live[k]]) != 0;
                const size_t by_bytes = (size_t)ho[q].segment_size + ho[q].segment_size / 4, by_blocks = prog ? blocks * 40 / ns : 0;
                c->segs.push_back(sg[q]);
                c->offs.push_back(c->offs.back() + ((std::max(by_bytes, by_blocks) + 65536 + 255) & ~(size_t)255));
            }
        }
        c->seg_first.push_back((int)c->segs.size());
        if (c->live.empty()) return 0;
        if (int rc = slot_reserve(s, c->frame_bytes, c->offs.back() + 256, c->segs.size(), c->live.size(), verify, 0)) return rc;
        c->dev_desc = c->host_desc; c->scratch_desc = c->host_desc;
        for (size_t k = 0; k < c->live.size(); ++k) {
            size_t off = c->frame_off[k];
            for (int cc = 0; cc < c->host_desc[k].ncomp; ++cc) {
                c->dev_desc[k].blocks[cc] = (int16_t*)(s->d_frames + off);
                if (verify) c->scratch_desc[k].blocks[cc] = (int16_t*)(s->d_scratch + off);
                off += (size_t)c->host_desc[k].width_blocks[cc] * c->host_desc[k].height_blocks[cc] * 128;
            }
        }
        // round-trip check of the progressive files the GPU decoded: their scans written again from the device frame
        c->pimg.clear(); c->pscan.clear(); c->pcheck.clear(); c->pscan_bytes = 0; c->corr_words = 0;
        c->pchecked.assign(c->live.size(), 0);
        if (verify)
            for (size_t nk = 0; nk < c->live.size(); ++nk) {
                const int k = old_k[nk];
                if (!on_prog[k]) continue;
                lep_huffprog_image pi = pchk[k].img;
                for (int cc = 0; cc < 4; ++cc) pi.blocks[cc] = cc < c->dev_desc[nk].ncomp ? c->dev_desc[nk].blocks[cc] : nullptr;
                for (size_t q = 0; q < pchk[k].scans.size(); ++q) {
                    lep_huffprog_scan sc = pchk[k].scans[q];
                    sc.image = (int32_t)c->pimg.size();
                    sc.out_cap = (uint32_t)std::min<size_t>(sc.out_cap, (size_t)pchk[k].len[q] + 64);   // anything longer is a mismatch anyway
                    sc.out_off = c->pscan_bytes;
                    c->pscan_bytes += ((size_t)sc.out_cap + 15) & ~(size_t)15;
                    sc.corr_off = (uint32_t)c->corr_words;
                    c->corr_words += sc.corr_cap;
                    c->pscan.push_back(sc);
                    c->pcheck.push_back(ScanCheck{sc.out_off, (uint64_t)praw_off[k][q], pchk[k].len[q], (uint32_t)nk});
                    c->pchecked[nk] = 1;
                }
                c->pimg.push_back(pi);
            }
        // ... and of the baseline files it decoded (their frame exists only on the device): every thread segment's scan bytes
        // written again by the decompressor's kernel and held against the file's.  A file that cannot take this path -- or whose
        // bytes do not come back -- goes through the per-file path, which restores it on the host and compares.
        c->vimg.clear(); c->vseg.clear(); c->vcheck.clear(); c->vscan_bytes = 0;
        c->vchecked.assign(c->live.size(), 0);
        if (verify)
            for (size_t nk = 0; nk < c->live.size(); ++nk) {
                const int k = old_k[nk];
                if (!on_gpu[k] || !vraw_len[k]) continue;
                lep_huff_image vi;
                lep_huff_segment vs[LEP_MAX_SEGMENTS];
                uint32_t ff[LEP_MAX_SEGMENTS], fl[LEP_MAX_SEGMENTS];
                int ns = 0, ok = 0;
                if (lep_jpeg_plan_scan_check(parsed[c->live[nk]], jpgs[c->live[nk]].len, &vi, vs, ff, fl, LEP_MAX_SEGMENTS, &ns, &ok) || !ok) continue;
                for (int cc = 0; cc < 4; ++cc) vi.blocks[cc] = cc < c->dev_desc[nk].ncomp ? c->dev_desc[nk].blocks[cc] : nullptr;
                for (int q = 0; q < ns; ++q) {
                    vs[q].image = (int32_t)c->vimg.size();
                    vs[q].out_cap = (uint32_t)std::min<size_t>(vs[q].out_cap, (size_t)fl[q] + 64);   // anything longer is a mismatch anyway
                    vs[q].out_off = c->vscan_bytes;
                    c->vscan_bytes += ((size_t)vs[q].out_cap + 15) & ~(size_t)15;
                    c->vseg.push_back(vs[q]);
                    c->vcheck.push_back(ScanCheck{vs[q].out_off, (uint64_t)vraw_off[k] + (ff[q] - vraw_first[k]), fl[q], (uint32_t)nk});
                }
                c->vimg.push_back(vi);
                c->vchecked[nk] = 1;
            }
        if (!c->vseg.empty()) {
            if (int rc = vscan_reserve(s, c->vscan_bytes + 256, c->vseg.size())) return rc;
            HIPOK(hipMemcpyAsync(s->d_vcheck, c->vcheck.data(), c->vcheck.size() * sizeof(ScanCheck), hipMemcpyHostToDevice, s_copy));
        }
        if (!c->pscan.empty()) {
            if (c->corr_words > 0xfffffff0u) return LEP_GPU_ERROR;
            if (int rc = prog_reserve(s, c->pscan_bytes + 256, c->corr_words + 16, c->pscan.size())) return rc;
            HIPOK(hipMemcpyAsync(s->d_pcheck, c->pcheck.data(), c->pcheck.size() * sizeof(ScanCheck), hipMemcpyHostToDevice, s_copy));
        }
        HIPOK(hipEventRecord(s->up, s_copy));
        return 0;
    };

    // coder kernels of one chunk, stream-ordered behind the previous chunk's (the call returns once they are enqueued, i.e.
    // after the previous chunk's kernels have finished: the descriptor upload inside lep_gpu_encode_device is synchronous)
    hipStream_t s_compute_first = s_compute;
    auto launch_chunk = [&](Chunk* c, Slot* s) -> int {
        const int nseg = (int)c->segs.size(), nimg = (int)c->live.size();
        if (!nimg) return 0;
        const int set = overlap ? (int)(s - slots) : 0;            // slot 0 / 1 <-> stream and workspace set 0 / 1
        hipStream_t s_compute = set ? s_compute2 : s_compute_first;
        if (int rc0 = lep_gpu_use_arena(g, set)) return rc0;
        HIPOK(hipStreamWaitEvent(s_compute, s->up, 0));
        int rc = lep_gpu_encode_device(g, c->dev_desc.data(), nimg, c->segs.data(), nseg, s->d_streams, c->offs.data(), s->d_len, s->d_status, s_compute);
        if (rc) return rc;
        if (verify) {
            HIPOK(hipMemsetAsync(s->d_scratch, 0, c->frame_bytes, s_compute));
            HIPOK(hipMemsetAsync(s->d_flags, 0, (size_t)nimg * 4, s_compute));
            rc = lep_gpu_decode_device(g, c->scratch_desc.data(), nimg, c->segs.data(), nseg, s->d_streams, c->offs.data(), s->d_len, s->d_status + nseg, s_compute);
            if (rc) return rc;
            for (int k = 0; k < nimg; ++k) {
                const size_t fb = frame_exact_bytes(c->host_desc[k]);   // the frame itself, not its rounded room in the slot
                hipLaunchKernelGGL(lep_compare_kernel, dim3(256), dim3(256), 0, s_compute, (const uint4*)(s->d_frames + c->frame_off[k]),
                                   (const uint4*)(s->d_scratch + c->frame_off[k]), fb / 16, s->d_flags + k, 1u);
            }
            st.gpu_verified_scans += (double)(c->vseg.size() + c->pscan.size());
            if (!c->vseg.empty()) {    // baseline files: the Huffman half, thread segment by thread segment against the file's bytes
                rc = lep_gpu_huffman_encode_device(g, c->vimg.data(), (int)c->vimg.size(), c->vseg.data(), (int)c->vseg.size(), s->d_vscan, s->d_vscanlen, nullptr, s_compute);
                if (rc) return rc;
                hipLaunchKernelGGL(lep_scan_check_bytes_kernel, dim3((unsigned)c->vseg.size()), dim3(256), 0, s_compute, s->d_vscan, s->d_vscanlen, s->d_scan,
                                   s->d_vcheck, s->d_flags);
            }
            if (!c->pscan.empty()) {   // progressive files: the Huffman half, scan by scan against the file's bytes
                rc = lep_gpu_huffman_progressive_encode_device(g, c->pimg.data(), (int)c->pimg.size(), c->pscan.data(), (int)c->pscan.size(), s->d_pscan,
                                                               s->d_corr, s->d_pscanlen, s_compute);
                if (rc) return rc;
                hipLaunchKernelGGL(lep_scan_check_kernel, dim3((unsigned)c->pscan.size()), dim3(256), 0, s_compute, s->d_pscan, s->d_pscanlen, s->d_scan,
                                   s->d_pcheck, s->d_flags);
            }
        }
        HIPOK(hipEventRecord(s->done, s_compute));
        (void)lep_gpu_use_arena(g, 0);
        return 0;
    };

    Joiner writer_guard;   // writes chunk k-1's containers while chunk k is on the GPU
    std::thread& writer = writer_guard.t;
    size_t ci = 0;
    int slot_i = 0;
    const double t_pipe = now_s();
    if (!chunks.empty()) {
        if (int rc = parse_and_upload(chunks[0].get(), &slots[0])) rc_all = rc;
        else if (int rc2 = launch_chunk(chunks[0].get(), &slots[0])) rc_all = rc2;
    }
    for (; !rc_all && ci < chunks.size(); ++ci, slot_i ^= 1) {
        Slot* s = &slots[slot_i];
        Chunk* c = chunks[ci].get();
        const int nseg = (int)c->segs.size(), nimg = (int)c->live.size();
        // while this chunk's coder kernels run: Huffman-decode the next chunk into the other slot (its previous user has been
        // written out), beside them on the GPU, and queue ITS coder kernels right behind -- only then fetch this chunk's results,
        // so that the download and the container writing hide under the next chunk's kernels
        if (writer.joinable()) writer.join();
        if (ci + 1 < chunks.size()) {
            if (int rc = parse_and_upload(chunks[ci + 1].get(), &slots[slot_i ^ 1])) { rc_all = rc; break; }
            if (int rc = launch_chunk(chunks[ci + 1].get(), &slots[slot_i ^ 1])) { rc_all = rc; break; }
        }
        // fetch this chunk's results
        std::vector<uint32_t> lens(nseg);
        std::vector<size_t> hoff(nseg, 0);
        std::vector<int32_t> sts((size_t)nseg * 2, 0);
        std::vector<uint32_t> flags(nimg, 0);
        if (nimg) {
            HIPOK(hipStreamWaitEvent(s_down, s->done, 0));
            HIPOK(hipMemcpyAsync(lens.data(), s->d_len, (size_t)nseg * 4, hipMemcpyDeviceToHost, s_down));
            HIPOK(hipMemcpyAsync(sts.data(), s->d_status, (size_t)nseg * (verify ? 8 : 4), hipMemcpyDeviceToHost, s_down));
            if (verify) HIPOK(hipMemcpyAsync(flags.data(), s->d_flags, (size_t)nimg * 4, hipMemcpyDeviceToHost, s_down));
            HIPOK(hipStreamSynchronize(s_down));
            // The streams come down in ONE copy of the arena as it lies on the device (every segment's slot is its JPEG bytes + 25 %, so
            // the slack costs a quarter more PCIe bytes) instead of one copy per segment: 7168 copies of ~200 KB took 250 ms per chunk
            // where their bytes need 30 (kernel + copy trace, profiles/r05b_*) -- and the last chunk's are the call's tail.  An arena
            // that is mostly slack (refused segments, tiny files in generous slots) keeps the packed form.
            size_t total = 0, live_bytes = 0;
            for (int k = 0; k < nseg; ++k) if (!sts[k]) live_bytes += lens[k];
            const size_t extent = nseg ? (size_t)c->offs[nseg] : 0;
            const bool whole = extent <= 3 * live_bytes + ((size_t)1 << 20);
            if (whole) {
                for (int k = 0; k < nseg; ++k) hoff[k] = (size_t)c->offs[k];
                if (int rc = slot_reserve(s, 0, 0, 0, 0, false, extent + 256)) { rc_all = rc; break; }
                if (extent) HIPOK(hipMemcpyAsync(s->h_streams, s->d_streams, extent, hipMemcpyDeviceToHost, s_down));
                st.d2h_bytes += (double)extent;
            } else {
                for (int k = 0; k < nseg; ++k) { hoff[k] = total; if (!sts[k]) total += ((size_t)lens[k] + 15) & ~(size_t)15; }
                if (int rc = slot_reserve(s, 0, 0, 0, 0, false, total + 256)) { rc_all = rc; break; }
                for (int k = 0; k < nseg; ++k)
                    if (!sts[k] && lens[k]) {
                        HIPOK(hipMemcpyAsync(s->h_streams + hoff[k], s->d_streams + c->offs[k], lens[k], hipMemcpyDeviceToHost, s_down));
                        st.d2h_bytes += lens[k];
                    }
            }
            HIPOK(hipStreamSynchronize(s_down));
        }
        // containers on the host pool, in the background
        writer = std::thread([&, c, s, lens, sts, flags, hoff]() {
            const double t0 = now_s();
            parallel_for((int)c->live.size(), threads, [&](int k) {
                const int i = c->live[k];
                const int s0 = c->seg_first[k], s1 = c->seg_first[k + 1];
                int rc = 0;
                lep_bytes strs[LEP_MAX_SEGMENTS];
                for (int q = s0; q < s1; ++q) {
                    if (sts[q] && !rc) rc = sts[q];
                    strs[q - s0].data = s->h_streams + hoff[q];
                    strs[q - s0].len = strs[q - s0].cap = lens[q];
                }
                if (!rc && verify) {
                    for (int q = s0; q < s1; ++q) if (sts[c->segs.size() + (size_t)q]) rc = LEP_ROUNDTRIP_FAILURE;
                    if (flags[k] & 1) rc = LEP_ROUNDTRIP_FAILURE;
                    // a progressive scan the GPU encoder did not reproduce (trailing restart markers, which the host appends; a
                    // non-canonical code choice): the per-file path decides (last loop of this function)
                    else if (!rc && (flags[k] & 2)) rc = LEP_BUFFER_TOO_SMALL;
                    // a baseline file the GPU decoded whose scan could not go through the check above: the per-file path restores
                    // it on the host and compares (nothing is released on an argument)
                    else if (!rc && !host_parsed[i] && !c->vchecked[k] && !c->pchecked[k]) rc = LEP_BUFFER_TOO_SMALL;
                }
                if (!rc) rc = lep_jpeg_write_lep(parsed[i], 0, strs, s1 - s0, &outs[i]);
                // the Huffman half of the reference's round-trip check for the files whose scans the host parser took (their
                // frame is still in this slot's pinned staging); the files the GPU decoded had theirs on the GPU (flags, above)
                if (!rc && verify && host_parsed[i]) {
                    rc = lep_jpeg_check_restores(parsed[i], outs[i].data, outs[i].len, jpgs[i].data, jpgs[i].len);
                    if (rc) { lep_free(outs[i].data); outs[i].data = nullptr; outs[i].len = outs[i].cap = 0; }
                }
                status[i] = rc;
                lep_jpeg_close(parsed[i]);
                parsed[i] = nullptr;
            });
            st.write_s += now_s() - t0;
        });
    }
    if (writer.joinable()) writer.join();
    st.pipeline_s = now_s() - t_pipe;
    for (int i = 0; i < n; ++i) if (parsed[i]) { lep_jpeg_close(parsed[i]); parsed[i] = nullptr; }
    // A thread segment that did not fit the stream space reserved for it (sized from its JPEG bytes: the arithmetic coder
    // almost never writes more than the Huffman coder did) is not a refusal the reference would make: such files go
    // through the per-file path, which reserves the worst case, once the pipeline has drained.
    if (!rc_all)
        for (int i = 0; i < n; ++i) {
            if (status[i] != LEP_BUFFER_TOO_SMALL) continue;
            lep_bytes o; o.data = nullptr; o.len = o.cap = 0;
            int rc = lep_compress(g, jpgs[i].data, jpgs[i].len, &o);
            if (!rc && verify) {   // the arithmetic-coder half of the round-trip check, through the per-file decoder
                lep_bytes back; back.data = nullptr; back.len = back.cap = 0;
                rc = lep_decompress(g, o.data, o.len, &back);
                if (!rc && (back.len != jpgs[i].len || memcmp(back.data, jpgs[i].data, back.len))) rc = LEP_ROUNDTRIP_FAILURE;
                lep_free(back.data);
                if (rc) { lep_free(o.data); o.data = nullptr; o.len = o.cap = 0; }
            }
            outs[i] = o;
            status[i] = rc;
            ++st_redone;
        }
    st.alloc_s = g_alloc_s;
    st.redone_files = st_redone;
    st.wall_s = now_s() - t_begin;
    if (stats) *stats = st;
    return rc_all;
}

// ---- .lep -> JPEG, batch ---------------------------------------------------------------------------------------------
int lep_decompress_batch(lep_gpu* g, const lep_bytes* leps, int n, lep_bytes* outs, int32_t* status, const lep_batch_options* o,
                         lep_batch_stats* stats) {
    if (!g || n < 0) return LEP_GPU_ERROR;
    g_batch_gpu = g;
    const int threads = o && o->host_threads > 0 ? o->host_threads : effective_cpus();
    const size_t chunk_budget = o && o->chunk_frame_bytes ? o->chunk_frame_bytes : ((size_t)32 << 30);
    const size_t chunk_images = o && o->chunk_images > 0 ? (size_t)o->chunk_images : 1024;
    const bool gpu_huffman = !(o && o->host_huffman);
    HIPOK(hipSetDevice(lep_gpu_device(g)));
    tune_malloc_for_pool();
    const double t_begin = now_s();
    lep_batch_stats st;
    memset(&st, 0, sizeof st);
    for (int i = 0; i < n; ++i) { outs[i].data = nullptr; outs[i].len = outs[i].cap = 0; status[i] = 0; }
    HandleVector<lep_file, lep_file_close> files(n);
    std::vector<size_t> fbytes(n, 0);
    std::vector<char> chained(n, 0);
    {
        const double t0 = now_s();
        parallel_for(n, threads, [&](int i) {
            int rc = lep_file_open(leps[i].data, leps[i].len, &files[i]);
            if (!rc && lep_chained_file_follows(leps[i].data, leps[i].len, lep_file_consumed(files[i]))) {
                chained[i] = 1;   // a stream of several v2+ files: the per-file path walks it once the pipeline has drained
                lep_file_close(files[i]); files[i] = nullptr;
                return;
            }
            if (!rc) fbytes[i] = (lep_file_frame_bytes(files[i]) + 255) & ~(size_t)255;
            if (rc) { status[i] = rc; if (files[i]) { lep_file_close(files[i]); files[i] = nullptr; } }
        });
        st.parse_s += now_s() - t0;
    }
    hipStream_t s_copy = nullptr, s_compute = nullptr, s_compute_b = nullptr, s_down = nullptr, s_scan = nullptr;
    StreamSet stream_set;
    stream_set.g = g;
    if (stream_set.make(&s_copy) || stream_set.make(&s_compute) || stream_set.make(&s_down) || stream_set.make(&s_scan)) return LEP_GPU_ERROR;
    // (No fifth stream: the runtime deals its 8 hardware queues out to streams in turn, the codec holds three streams of its own, and
    // the stream that then SHARES a queue with the compute stream has its work -- the next chunk's upload, say -- run behind the decode
    // kernel: measured, 0.77 s of a 1.9 s call (LEP_BATCH_TRACE).  The scan stream is free unless LEP_BATCH_SCAN_STREAM=1 asks for it.)
    s_compute_b = s_scan;
    // Consecutive chunks' decode kernels on TWO streams (and the codec's two workspace sets) where that pays: a launch is over when its
    // longest thread segment is, and while the long segments of a ragged chunk (1080p files beside 4K ones; a chunk that does not fill the
    // chip) run on, the wave slots its short ones have left stand empty -- the next chunk's launch, longest segments first, moves into them
    // instead of waiting behind the whole kernel.  A chunk of equal segments that fills the chip keeps the one-stream order: there is no
    // tail to fill, and its scan encoders would only have to fight the next decode kernel for slots (the LEP_BATCH_SCAN_STREAM finding
    // below).  LEP_BATCH_DEC_OVERLAP=0 | 1 forces never / always.
    int dec_overlap = -1;
    if (const char* e = getenv("LEP_BATCH_DEC_OVERLAP")) dec_overlap = atoi(e) ? 1 : 0;
    // The scan encoders run behind the decoder on ITS stream.  On a stream of their own (LEP_BATCH_SCAN_STREAM=1) they compete with the
    // next chunk's decode kernel for wave slots it fills completely, finish when it does, and hold the chunk's download back:
    // 1530 against 1774 MB/s (profiles/r06j_*).
    if (!(getenv("LEP_BATCH_SCAN_STREAM") && atoi(getenv("LEP_BATCH_SCAN_STREAM")) == 1)) s_scan = s_compute;
    else { s_compute_b = nullptr; dec_overlap = 0; }   // (the second compute stream IS the scan stream)
    Slot* slots = g_slots;
    g_alloc_s = 0;
    int rc_all = 0;

    auto cut_chunk = [&](int first) -> std::unique_ptr<Chunk> {
        std::unique_ptr<Chunk> c(new Chunk);
        c->first = first;
        size_t bytes = 0;
        int i = first;
        for (; i < n; ++i) {
            if (!files[i]) continue;
            const size_t fb = fbytes[i];
            if (!c->live.empty() && (bytes + fb > chunk_budget || c->live.size() >= chunk_images)) break;
            c->live.push_back(i); c->frame_off.push_back(bytes);
            bytes += fb;
        }
        c->count = i - first;
        c->frame_bytes = bytes;
        c->offs.push_back(0);
        return c;
    };
    // streams into the slot's pinned arena (packed back to back), upload, frames zeroed on the device
    std::vector<std::vector<uint32_t>> chunk_lens(2);
    const double t_pipe0 = now_s();
    auto stage_and_upload = [&](Chunk* c, Slot* s, std::vector<uint32_t>* lens) -> int {
        if (c->live.empty()) return 0;
        if (getenv("LEP_BATCH_TRACE")) fprintf(stderr, "[batch] stage_and_upload first=%d begins (t=%.3f)\n", c->first, now_s() - t_pipe0);
        c->segs.clear(); c->offs.assign(1, 0); c->seg_first.clear(); lens->clear();
        std::vector<const uint8_t*> src;
        for (size_t k = 0; k < c->live.size(); ++k) {
            lep_segment sg[LEP_MAX_SEGMENTS];
            lep_bytes sb[LEP_MAX_SEGMENTS];
            const int ns = lep_file_segments(files[c->live[k]], sg, sb, (int)k);
            c->seg_first.push_back((int)c->segs.size());
            for (int q = 0; q < ns; ++q) {
                c->segs.push_back(sg[q]);
                lens->push_back((uint32_t)sb[q].len);
                src.push_back(sb[q].data);
                c->offs.push_back(c->offs.back() + sb[q].len);
            }
        }
        c->seg_first.push_back((int)c->segs.size());
        if (int rc = slot_reserve(s, c->frame_bytes, c->offs.back() + 256, c->segs.size(), c->live.size(), false, c->offs.back() + 256)) return rc;
        if (getenv("LEP_BATCH_TRACE")) fprintf(stderr, "[batch]   %s (t=%.3f)\n", "slot reserved", now_s() - t_pipe0);
        const double t0 = now_s();
        parallel_for((int)c->segs.size(), threads, [&](int q) { if ((*lens)[q]) memcpy(s->h_streams + c->offs[q], src[q], (*lens)[q]); });
        if (getenv("LEP_BATCH_TRACE")) fprintf(stderr, "[batch]   %s (t=%.3f)\n", "streams staged", now_s() - t_pipe0);
        st.stage_s += now_s() - t0;
        // geometry of every frame (the frames themselves are decoded into device memory)
        c->host_desc.resize(c->live.size());
        for (size_t k = 0; k < c->live.size(); ++k) lep_file_describe_into(files[c->live[k]], nullptr, (size_t)-1, &c->host_desc[k]);
        c->dev_desc = c->host_desc;
        for (size_t k = 0; k < c->live.size(); ++k) {
            size_t off = c->frame_off[k];
            for (int cc = 0; cc < c->host_desc[k].ncomp; ++cc) {
                c->dev_desc[k].blocks[cc] = (int16_t*)(s->d_frames + off);
                off += (size_t)c->host_desc[k].width_blocks[cc] * c->host_desc[k].height_blocks[cc] * 128;
            }
        }
        // GPU Huffman re-encode plan: eligible files get their scan bytes back instead of their frames
        c->himg.clear(); c->hseg.clear(); c->hfirst.assign(c->live.size(), -1); c->hslot.clear(); c->hbound.clear(); c->scan_bytes = 0;
        if (gpu_huffman) {
            for (size_t k = 0; k < c->live.size(); ++k) {
                lep_huff_image hi;
                lep_huff_segment hs[LEP_MAX_SEGMENTS];
                int ns = 0, ok = 0;
                if (lep_file_recode_plan(files[c->live[k]], &hi, hs, &ns, &ok) || !ok) continue;
                for (int cc = 0; cc < 4; ++cc) hi.blocks[cc] = cc < c->dev_desc[k].ncomp ? c->dev_desc[k].blocks[cc] : nullptr;
                c->hfirst[k] = (int)c->hseg.size();
                size_t later = 0;
                for (int q = 1; q < ns; ++q) later += hs[q].out_cap;
                for (int q = 0; q < ns; ++q) {
                    size_t slot = hs[q].out_cap;
                    if (q == 0) {   // segment 0 is only bounded by the file: give it what the others leave, plus slack
                        const size_t room = hs[0].out_cap;
                        slot = std::min(room, (room > later ? room - later : 0) + 8192);
                    }
                    c->hbound.push_back(hs[q].out_cap);
                    hs[q].image = (int32_t)c->himg.size();
                    hs[q].out_off = c->scan_bytes;
                    hs[q].out_cap = (uint32_t)slot;
                    c->hslot.push_back((uint32_t)slot);
                    c->scan_bytes += (slot + 15) & ~(size_t)15;
                    c->hseg.push_back(hs[q]);
                }
                c->himg.push_back(hi);
            }
            // no room for the scan arena (hostile hand-off sizes are clamped by recode_prepare, so this is a real shortage):
            // the chunk's files take the host re-coder instead of failing everybody's request
            if (!c->hseg.empty() && scan_reserve(s, c->scan_bytes + 256, c->hseg.size())) {
                c->himg.clear(); c->hseg.clear(); c->hfirst.assign(c->live.size(), -1); c->hslot.clear(); c->hbound.clear(); c->scan_bytes = 0;
            }
        }
        // progressive files: every scan is a function of the finished frame -> one GPU wavefront per (image, scan)
        c->pimg.clear(); c->pscan.clear(); c->pfirst.assign(c->live.size(), -1); c->pcount.assign(c->live.size(), 0); c->pscan_bytes = 0; c->corr_words = 0;
        if (gpu_huffman) {
            std::vector<lep_huffprog_scan> tmp(256);
            for (size_t k = 0; k < c->live.size(); ++k) {
                if (c->hfirst[k] >= 0) continue;
                lep_huffprog_image pi;
                int ns = 0, ok = 0;
                if (lep_file_recode_plan_progressive(files[c->live[k]], &pi, tmp.data(), (int)tmp.size(), &ns, &ok) || !ok) continue;
                for (int cc = 0; cc < 4; ++cc) pi.blocks[cc] = cc < c->dev_desc[k].ncomp ? c->dev_desc[k].blocks[cc] : nullptr;
                c->pfirst[k] = (int)c->pscan.size(); c->pcount[k] = ns;
                for (int q = 0; q < ns; ++q) {
                    tmp[q].image = (int32_t)c->pimg.size();
                    tmp[q].out_off = c->pscan_bytes;
                    c->pscan_bytes += ((size_t)tmp[q].out_cap + 15) & ~(size_t)15;
                    tmp[q].corr_off = (uint32_t)c->corr_words;
                    c->corr_words += tmp[q].corr_cap;
                    c->pscan.push_back(tmp[q]);
                }
                c->pimg.push_back(pi);
            }
            if (!c->pscan.empty() && (c->corr_words > 0xfffffff0u || prog_reserve(s, c->pscan_bytes + 256, c->corr_words + 16, c->pscan.size()))) {
                c->pimg.clear(); c->pscan.clear(); c->pfirst.assign(c->live.size(), -1); c->pcount.assign(c->live.size(), 0);   // no room: host re-coder
            }
        }
        // files the host re-coder handles read their frame where the D2H copy puts it: the slot's pinned buffer
        bool any_host = false;
        for (size_t k = 0; k < c->live.size(); ++k) { any_host |= c->hfirst[k] < 0 && c->pfirst[k] < 0; if (c->hfirst[k] >= 0 || c->pfirst[k] >= 0) st.gpu_huffman_files += 1; }
        if (any_host) {
            if (int rc = host_frames_reserve(s, c->frame_bytes)) return rc;
            for (size_t k = 0; k < c->live.size(); ++k)
                if (c->hfirst[k] < 0 && c->pfirst[k] < 0) lep_file_describe_into(files[c->live[k]], s->h_frames + c->frame_off[k], fbytes[c->live[k]], &c->host_desc[k]);
        }
        if (getenv("LEP_BATCH_TRACE")) fprintf(stderr, "[batch]   %s (t=%.3f)\n", "plans made", now_s() - t_pipe0);
        HIPOK(hipMemcpyAsync(s->d_streams, s->h_streams, c->offs.back(), hipMemcpyHostToDevice, s_copy));
        HIPOK(hipMemcpyAsync(s->d_len, lens->data(), lens->size() * 4, hipMemcpyHostToDevice, s_copy));
        // The decode kernel stores every coefficient of every block it decodes: only a file whose frame holds blocks that are NOT coded
        // (a truncated one) needs its frame cleared.  (Clearing the whole chunk was a 25 GB fill kernel that could not start while the
        // previous chunk's decoder held every wave slot, and this function waited for it.)
        for (size_t k = 0; k < c->live.size(); ++k) {
            const lep_image_desc& d = c->host_desc[k];
            bool whole = true;
            for (int cc = 0; cc < d.ncomp; ++cc) whole = whole && d.coded_blocks[cc] == d.width_blocks[cc] * d.height_blocks[cc];
            if (!whole) HIPOK(hipMemsetAsync(s->d_frames + c->frame_off[k], 0, fbytes[c->live[k]], s_copy));
        }
        if (getenv("LEP_BATCH_TRACE")) fprintf(stderr, "[batch]   %s (t=%.3f)\n", "copies queued", now_s() - t_pipe0);
        HIPOK(hipEventRecord(s->up, s_copy));
        HIPOK(hipStreamSynchronize(s_copy));   // `lens` / pinned arena are reused by the caller
        st.h2d_bytes += (double)c->offs.back();
        if (getenv("LEP_BATCH_TRACE")) fprintf(stderr, "[batch] stage_and_upload first=%d done (t=%.3f)\n", c->first, now_s() - t_pipe0);
        return 0;
    };

    // decoder + Huffman re-encode kernels of one chunk: stream-ordered behind the previous chunk's, or beside them on the other stream
    // when the previous chunk leaves wave slots to fill (see dec_overlap above)
    hipStream_t s_prev = nullptr;          // the stream the previous chunk's kernels went to
    Slot* slot_prev = nullptr;
    bool prev_ragged = false;
    int set_prev = 0;
    const hipStream_t s_compute_a = s_compute;
    auto launch_chunk = [&](Chunk* c, Slot* s) -> int {
        const int nseg = (int)c->segs.size(), nimg = (int)c->live.size();
        if (!nimg) return 0;
        static const bool tr = getenv("LEP_BATCH_TRACE") != nullptr;
        const double tt0 = now_s();
        // does THIS chunk leave a tail?  segments whose block counts differ by more than 1.5x, or fewer segments than fill the chip
        int64_t lo = INT64_MAX, hi = 0;
        for (int q = 0; q < nseg; ++q) {
            const lep_image_desc& d = c->dev_desc[(size_t)c->segs[q].image];
            const int y1 = c->segs[q].is_last ? d.height_blocks[0] : std::min<int>(c->segs[q].luma_y_end, d.height_blocks[0]);
            const int64_t w = std::max<int64_t>(1, (int64_t)std::max(0, y1 - c->segs[q].luma_y_start) * d.width_blocks[0]);
            lo = std::min(lo, w); hi = std::max(hi, w);
        }
        const bool ragged = hi * 2 > lo * 3 || nseg < 6144;
        const bool beside = s_prev && (dec_overlap == 1 || (dec_overlap < 0 && prev_ragged));
        // the second workspace set (its own 3 MB of model per segment) only where two launches are in flight; launches in stream order
        // share one
        const int set = !s_prev ? 0 : (beside ? set_prev ^ 1 : set_prev);
        set_prev = set;
        hipStream_t s_compute = !s_prev ? s_compute_a : (beside ? (s_prev == s_compute_a ? s_compute_b : s_compute_a) : s_prev);
        hipStream_t s_scan_here = s_scan == s_compute_a ? s_compute : s_scan;
        // this slot's buffers were last used by the chunk before the previous one, possibly on the other stream -- and a workspace set by
        // either of the two chunks in front
        // (At most ONE earlier chunk is in flight when this runs: the caller has fetched -- waited for the download of -- every chunk but
        // the one in front before it launches this one.  A launch that is not `beside` goes onto the previous launch's stream, so stream
        // order is its dependency; ADVICE round 5: the extra wait that stood here could never fire.)
        if (s->done_recorded) HIPOK(hipStreamWaitEvent(s_compute, s->done, 0));
        HIPOK(hipStreamWaitEvent(s_compute, s->up, 0));
        if (int rc0 = lep_gpu_use_arena(g, set)) return rc0;
        (void)lep_gpu_expect_company(g, dec_overlap != 0 && (beside || ragged) && c->count < n);   // (a call of one chunk has no neighbour)
        int rc = lep_gpu_decode_device(g, c->dev_desc.data(), nimg, c->segs.data(), nseg, s->d_streams, c->offs.data(), s->d_len, s->d_status, s_compute);
        (void)lep_gpu_expect_company(g, 0);
        (void)lep_gpu_use_arena(g, 0);
        if (rc) return rc;
        // the scan encoders' descriptors live in ONE device buffer of the codec: the previous chunk's encoders (other stream) read them
        if (beside && slot_prev && slot_prev->done_recorded) HIPOK(hipStreamWaitEvent(s_scan_here, slot_prev->done, 0));
        s_prev = s_compute; slot_prev = s; prev_ragged = ragged;
        hipStream_t s_scan = s_scan_here;
        const double tt1 = now_s();
        if (s_scan != s_compute) { HIPOK(hipEventRecord(s->decoded, s_compute)); HIPOK(hipStreamWaitEvent(s_scan, s->decoded, 0)); }
        if (!c->hseg.empty()) {
            rc = lep_gpu_huffman_encode_device(g, c->himg.data(), (int)c->himg.size(), c->hseg.data(), (int)c->hseg.size(), s->d_scan, s->d_scanlen, huff_ends(s), s_scan);
            if (rc) return rc;
        }
        if (tr) fprintf(stderr, "[batch] launch_chunk first=%d: decode launch %.3f s, scan-encode launch %.3f s (t=%.3f)\n", c->first, tt1 - tt0, now_s() - tt1, now_s() - t_pipe0);
        if (!c->pscan.empty()) {
            rc = lep_gpu_huffman_progressive_encode_device(g, c->pimg.data(), (int)c->pimg.size(), c->pscan.data(), (int)c->pscan.size(), s->d_pscan, s->d_corr,
                                                           s->d_pscanlen, s_scan);
            if (rc) return rc;
        }
        HIPOK(hipEventRecord(s->done, s_scan));
        s->done_recorded = true;
        return 0;
    };
    Joiner writer_guard;
    std::thread& writer = writer_guard.t;
    for (int k = 0; k < 2; ++k) slots[k].done_recorded = false;   // (events of an earlier call: nothing of it is in flight)
    std::unique_ptr<Chunk> cur = cut_chunk(0), nxt;
    int slot_i = 0;
    if (int rc = stage_and_upload(cur.get(), &slots[0], &chunk_lens[0])) rc_all = rc;
    const double t_pipe = now_s();
    // A host-to-device copy does not run beside a decode kernel on this platform (kernel + copy traces of rounds 4 and 5: the copies sit in
    // the gaps between the kernels; LEP_BATCH_TRACE: the upload of chunk k + 1 returns when chunk k's kernel ends).  For chunks of equal
    // segments that costs nothing -- the next kernel could not start earlier anyway.  But the SECOND chunk of a ragged call is what should
    // move into the wave slots the first one's short segments leave: so when the first chunk is ragged, the second is staged and uploaded
    // BEFORE the first is launched, and both launches go out back to back (the second beside the first: launch_chunk).  Behind a chunk
    // of equal segments the second launch waits in stream order as before, and only saves its upload's place in the gap (~30 ms).
    bool second_is_staged = false;
    if (!rc_all && cur && cur->count > 0 && dec_overlap != 0 && cur->first + cur->count < n) {
        // (whether the second launch then goes BESIDE the first is launch_chunk's decision: only behind a ragged chunk)
        nxt = cut_chunk(cur->first + cur->count);
        if (int rc = stage_and_upload(nxt.get(), &slots[1], &chunk_lens[1])) rc_all = rc;
        second_is_staged = true;
    }
    if (!rc_all && cur && cur->count > 0) rc_all = launch_chunk(cur.get(), &slots[0]);
    if (!rc_all && second_is_staged && nxt && nxt->count > 0) rc_all = launch_chunk(nxt.get(), &slots[1]);
    while (!rc_all && cur && cur->count > 0) {
        Slot* s = &slots[slot_i];
        Chunk* c = cur.get();
        const int nseg = (int)c->segs.size(), nimg = (int)c->live.size();
        // stage the next chunk and queue its kernels behind this chunk's before fetching this chunk's results
        if (writer.joinable()) writer.join();
        if (second_is_staged) second_is_staged = false;   // (the call's second chunk: staged and launched above)
        else {
            nxt = c->first + c->count < n ? cut_chunk(c->first + c->count) : nullptr;
            if (nxt) {
                if (int rc = stage_and_upload(nxt.get(), &slots[slot_i ^ 1], &chunk_lens[slot_i ^ 1])) { rc_all = rc; break; }
                if (nxt->count > 0) { if (int rc = launch_chunk(nxt.get(), &slots[slot_i ^ 1])) { rc_all = rc; break; } }
            }
        }
        std::vector<int32_t> sts(nseg);
        std::vector<uint32_t> slens(c->hseg.size()), plens(c->pscan.size());
        std::vector<lep_huff_end> sends(c->hseg.size());
        std::vector<size_t> poff(c->pscan.size(), 0);   // packed offsets of the progressive scans in the pinned mirror
        if (nimg) {
            HIPOK(hipStreamWaitEvent(s_down, s->done, 0));
            HIPOK(hipMemcpyAsync(sts.data(), s->d_status, (size_t)nseg * 4, hipMemcpyDeviceToHost, s_down));
            if (!c->hseg.empty()) HIPOK(hipMemcpyAsync(slens.data(), s->d_scanlen, c->hseg.size() * 4, hipMemcpyDeviceToHost, s_down));
            if (!c->hseg.empty()) HIPOK(hipMemcpyAsync(sends.data(), huff_ends(s), c->hseg.size() * sizeof(lep_huff_end), hipMemcpyDeviceToHost, s_down));
            if (!c->pscan.empty()) HIPOK(hipMemcpyAsync(plens.data(), s->d_pscanlen, c->pscan.size() * 4, hipMemcpyDeviceToHost, s_down));
            HIPOK(hipStreamSynchronize(s_down));
            if (!c->pscan.empty()) {
                size_t total = 0;
                for (int k = 0; k < nimg; ++k) {
                    if (c->pfirst[k] < 0) continue;
                    bool fits = true;
                    for (int q = c->pfirst[k]; q < c->pfirst[k] + c->pcount[k]; ++q) fits = fits && !(plens[q] & 0x80000000u);
                    if (!fits) {   // a scan outgrew its slot or its scratch: that file takes the host re-coder
                        c->pfirst[k] = -1;
                        if (int rc = host_frames_reserve(s, c->frame_bytes)) { rc_all = rc; break; }
                        lep_file_describe_into(files[c->live[k]], s->h_frames + c->frame_off[k], fbytes[c->live[k]], &c->host_desc[k]);
                        continue;
                    }
                    for (int q = c->pfirst[k]; q < c->pfirst[k] + c->pcount[k]; ++q) { poff[q] = total; total += ((size_t)plens[q] + 15) & ~(size_t)15; }
                }
                if (!rc_all) { if (int rc = prog_host_reserve(s, total + 256)) rc_all = rc; }
                if (rc_all) break;
                for (int k = 0; k < nimg; ++k) {
                    if (c->pfirst[k] < 0) continue;
                    for (int q = c->pfirst[k]; q < c->pfirst[k] + c->pcount[k]; ++q)
                        if (plens[q]) { HIPOK(hipMemcpyAsync(s->h_pscan + poff[q], s->d_pscan + c->pscan[q].out_off, plens[q], hipMemcpyDeviceToHost, s_down)); st.d2h_bytes += plens[q]; }
                }
            }
            // the scan bytes of the GPU-coded files come down in one copy of the arena range they lie in (same reasoning as the
            // compressor's streams: 8195 copies per chunk took 370 ms), unless that range is mostly slack
            // does the GPU's answer for file k stand?  A segment that filled its reserved slot may have been cut short; a truncated file's
            // segments are the file's bytes only if the encoder stopped at the cut in the LAST thread with that thread's byte bound reached
            // (lep_huff_simt.h code_mcus, recode_finish) -- otherwise the host re-coder takes the file
            auto gpu_answer_stands = [&](int k) {
                const int h0 = c->hfirst[k], h1 = h0 + (c->seg_first[k + 1] - c->seg_first[k]);
                for (int q = h0; q < h1; ++q) {
                    if (slens[q] >= c->hslot[q] && c->hslot[q] < c->hbound[q]) return false;
                    if (sends[q].pad & 2) return false;
                    if ((sends[q].pad & 1) && (q + 1 != h1 || sends[q].attempted < c->hbound[q])) return false;
                }
                return true;
            };
            size_t scan_lo = ~(size_t)0, scan_hi = 0, scan_live = 0;
            for (int k = 0; k < nimg; ++k) {
                if (c->pfirst[k] >= 0 || c->hfirst[k] < 0) continue;
                const int h0 = c->hfirst[k], h1 = h0 + (c->seg_first[k + 1] - c->seg_first[k]);
                if (!gpu_answer_stands(k)) continue;
                for (int q = h0; q < h1; ++q) if (slens[q]) {
                    scan_lo = std::min<size_t>(scan_lo, c->hseg[q].out_off); scan_hi = std::max<size_t>(scan_hi, (size_t)c->hseg[q].out_off + slens[q]);
                    scan_live += slens[q];
                }
            }
            const bool scan_whole = scan_hi > scan_lo && scan_hi - scan_lo <= 3 * scan_live + ((size_t)1 << 20);
            if (scan_whole) {
                HIPOK(hipMemcpyAsync(s->h_scan + scan_lo, s->d_scan + scan_lo, scan_hi - scan_lo, hipMemcpyDeviceToHost, s_down));
                st.d2h_bytes += (double)(scan_hi - scan_lo);
            }
            for (int k = 0; k < nimg; ++k) {
                if (c->pfirst[k] >= 0) continue;
                bool on_gpu = c->hfirst[k] >= 0;
                if (on_gpu) {   // a segment that filled its reserved slot may have been cut short: let the host redo that file
                    const int h0 = c->hfirst[k], h1 = h0 + (c->seg_first[k + 1] - c->seg_first[k]);
                    on_gpu = gpu_answer_stands(k);
                    if (!on_gpu) {
                        st.gpu_huffman_files -= 1;
                        c->hfirst[k] = -1;
                        if (int rc = host_frames_reserve(s, c->frame_bytes)) { rc_all = rc; break; }
                        lep_file_describe_into(files[c->live[k]], s->h_frames + c->frame_off[k], fbytes[c->live[k]], &c->host_desc[k]);
                    }
                    else if (!scan_whole) for (int q = h0; q < h1; ++q)
                        if (slens[q]) { HIPOK(hipMemcpyAsync(s->h_scan + c->hseg[q].out_off, s->d_scan + c->hseg[q].out_off, slens[q], hipMemcpyDeviceToHost, s_down)); st.d2h_bytes += slens[q]; }
                }
                if (!on_gpu) {
                    const size_t fb = (k + 1 < nimg ? c->frame_off[k + 1] : c->frame_bytes) - c->frame_off[k];
                    HIPOK(hipMemcpyAsync(s->h_frames + c->frame_off[k], s->d_frames + c->frame_off[k], fb, hipMemcpyDeviceToHost, s_down));
                    st.d2h_bytes += (double)fb;
                }
            }
            HIPOK(hipStreamSynchronize(s_down));
            if (rc_all) break;
        }
        std::shared_ptr<Chunk> keep(cur.release());
        writer = std::thread([&, keep, s, sts, slens, sends, plens, poff]() {
            const double t0 = now_s();
            parallel_for((int)keep->live.size(), threads, [&](int k) {
                const int i = keep->live[k];
                int rc = 0;
                for (int q = keep->seg_first[k]; q < keep->seg_first[k + 1]; ++q) if (sts[q] && !rc) rc = sts[q];
                if (!rc && keep->hfirst[k] >= 0) {   // GPU-coded scan bytes: glue header, segments, trailer
                    const int h0 = keep->hfirst[k], ns = keep->seg_first[k + 1] - keep->seg_first[k];
                    lep_bytes sb[LEP_MAX_SEGMENTS];
                    for (int q = 0; q < ns; ++q) { sb[q].data = s->h_scan + keep->hseg[h0 + q].out_off; sb[q].len = sb[q].cap = slens[h0 + q]; }
                    rc = lep_file_recode_finish(files[i], sb, sends.data() + h0, ns, &outs[i]);
                } else if (!rc && keep->pfirst[k] >= 0) {   // progressive file, scans coded on the GPU: glue header pieces and scans
                    const int p0 = keep->pfirst[k], ns = keep->pcount[k];
                    std::vector<lep_bytes> sb((size_t)ns);
                    for (int q = 0; q < ns; ++q) { sb[q].data = s->h_pscan + poff[p0 + q]; sb[q].len = sb[q].cap = plens[p0 + q]; }
                    rc = lep_file_recode_finish_progressive(files[i], sb.data(), ns, &outs[i]);
                } else if (!rc) rc = lep_file_recode(files[i], &outs[i]);   // host re-coder reads the frame in place (pinned D2H buffer)
                status[i] = rc;
                lep_file_close(files[i]);
                files[i] = nullptr;
            });
            st.write_s += now_s() - t0;
        });
        cur = std::move(nxt);
        slot_i ^= 1;
    }
    if (writer.joinable()) writer.join();
    st.pipeline_s = now_s() - t_pipe;
    for (int i = 0; i < n; ++i) if (files[i]) { lep_file_close(files[i]); files[i] = nullptr; }
    if (!rc_all)
        for (int i = 0; i < n; ++i)
            if (chained[i]) status[i] = lep_decompress(g, leps[i].data, leps[i].len, &outs[i]);
    st.alloc_s = g_alloc_s;
    st.wall_s = now_s() - t_begin;
    if (stats) *stats = st;
    return rc_all;
}

}  // extern "C"
