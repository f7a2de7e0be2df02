// lep_gpu.hip -- layer 1 of the C ABI: HIP kernels for gfx950 and the runtime object that launches them.
// One wavefront per (image, thread segment) work item; per-segment adaptive model resident in HBM
// (2.9 MB each, hot tables first), coefficient frames read/written in place in the reference's
// AlignedBlock layout, streams written to / read from one device arena.  No CPU fallback: every entry
// point returns LEP_GPU_ERROR if HIP reports a failure.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/lepton_mi355x.h"

#define LEP_DEV __device__ __forceinline__
#include "lep_core.h"
#include "lep_enc3.h"
#include "lep_dec4.h"
#include "lep_enc5.h"
#include "lep_huff.h"
#include "lep_huffdec.h"
#include "lep_huffdec_simt.h"
#include "lep_huff_simt.h"
#include "lep_huffprog.h"
#include "lep_huffprog_simt.h"
#include "lep_huffprogdec.h"
#include "lep_huffprogdec_win.h"

using namespace lepdev;

namespace {

// the encoder uses the dense model layout (kModelBranches words), the decoder the group-aligned one (lep3::kModelWords);
// segments are spaced by the larger so that one arena serves both
constexpr size_t kModelStride = lep3::kModelWords > kModelBranches ? lep3::kModelWords : kModelBranches;

template <uint32_t WORDS = kModelBranches>
__device__ void reset_segment_state(uint32_t* model, NSum* ns, int ns_count, int lane) {
    // model reset = Branch::identity() everywhere (model.hh:114-125); 16-byte coalesced stores
    uint4* m4 = reinterpret_cast<uint4*>(model);
    const uint4 init = make_uint4(kBranchInit, kBranchInit, kBranchInit, kBranchInit);
    for (uint32_t i = lane; i < WORDS / 4; i += 64) m4[i] = init;
    uint32_t* n32 = reinterpret_cast<uint32_t*>(ns);
    const uint32_t words = (uint32_t)ns_count * (sizeof(NSum) / 4);
    for (uint32_t i = lane; i < words; i += 64) n32[i] = 0;
}

// The encoder (lep_enc3.h): lane-parallel context / bin-list phases, one model round trip per block, bool coder as uniform
// vector code; 64 VGPRs -> 8 wavefronts per SIMD.  (Earlier generations -- single-lane, exec-masked lane 0, scalar-unit
// serial part -- live on as CPU cross-checks under tests/emu/retired/.)
template <int WAVES>
__global__ __launch_bounds__(64, WAVES) void lep_encode_v3_kernel(const ImageDev* __restrict__ images, const SegDev* __restrict__ segs,
                                                              uint32_t* models, NSum* ns_area, const uint64_t* ns_offsets,
                                                              uint8_t* streams, uint32_t* stream_len, int32_t* status, uint32_t* bins) {
    __shared__ lep3::Enc3Shared sh;
    const int s = blockIdx.x, lane = threadIdx.x;
    const SegDev seg = segs[s];
    const ImageDev* img = images + seg.image;
    uint32_t* model = models + (size_t)s * kModelStride;
    NSum* ns = ns_area + ns_offsets[s];
    reset_segment_state(model, ns, img->ns_total, lane);
    __syncthreads();
    lep3::Enc3Wave w;
    int rc = w.run(img, seg, model, ns, &sh, streams + seg.stream_off, seg.stream_cap);
    uint32_t n = rc ? 0 : w.bc.finish();
    if (lane != 0) return;
    if (!rc && w.bc.overflow) rc = LEP_BUFFER_TOO_SMALL;
    stream_len[seg.slot] = n;
    status[seg.slot] = rc;
    bins[seg.slot] = w.nbins;
}

// Two wavefronts per thread segment (lep_enc3.h, Enc3Pipe): wavefront 0 produces resolved bin-list chunks, wavefront 1
// runs the bool coder one chunk behind.  For launches that cannot fill the chip with one wavefront per segment (a single
// image is 8 segments): the segment's serial chain is about half as long.
__global__ __launch_bounds__(128, 2) void lep_encode_v3x2_kernel(const ImageDev* __restrict__ images, const SegDev* __restrict__ segs,
                                                                 uint32_t* models, NSum* ns_area, const uint64_t* ns_offsets,
                                                                 uint8_t* streams, uint32_t* stream_len, int32_t* status, uint32_t* bins) {
    __shared__ lep3::Enc3Shared sh;
    __shared__ lep3::Enc3Pipe pipe;
    const int s = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const SegDev seg = segs[s];
    const ImageDev* img = images + seg.image;
    uint32_t* model = models + (size_t)s * kModelStride;
    NSum* ns = ns_area + ns_offsets[s];
    if (wave == 0) reset_segment_state(model, ns, img->ns_total, lane);
    if (threadIdx.x == 0) { pipe.count[0] = 0; pipe.count[1] = 0; }
    __syncthreads();
    lep3::Enc3Wave w;
    if (wave == 0) {
        int rc = w.run(img, seg, model, ns, &sh, streams + seg.stream_off, seg.stream_cap, &pipe);
        if (lane != 0) return;
        if (!rc && pipe.out_overflow) rc = LEP_BUFFER_TOO_SMALL;
        stream_len[seg.slot] = rc ? 0 : pipe.out_len;
        status[seg.slot] = rc;
        bins[seg.slot] = w.nbins;
    } else {
        w.consume(&sh, &pipe, streams + seg.stream_off, seg.stream_cap);
    }
}

#ifdef LEP_PROF
__device__ unsigned long long g_prof4[8192][32];   // private accumulators per wave (no atomic contention)
#endif

// exhaustive check of the float-reciprocal / table-reciprocal Branch probabilities against integer division (all 255 x 255 count pairs)
__global__ void lep_selftest_kernel(uint32_t* mismatches) {
    __shared__ uint32_t inv24[512];
    for (uint32_t d = threadIdx.x; d < 512; d += blockDim.x) inv24[d] = lep4::inv24_of(d);
    __syncthreads();
    const uint32_t f = blockIdx.x + 1, t = threadIdx.x + 1;
    if (t > 255) return;
    for (uint32_t obs = 0; obs < 2; ++obs) {   // v4 decoder's table-reciprocal update (lep_dec4.h)
        const uint32_t w = f | (t << 8) | (((f << 8) / (f + t)) << 16);
        if (lep4::bupd_t(w, obs, inv24) != branch_update(w, (int)obs)) atomicAdd(mismatches, 1u);
    }
    if (lep3::prob_of(f, t) != (f << 8) / (f + t)) atomicAdd(mismatches, 1u);
    if (lep5::prob16(f | (t << 8)) != (f << 8) / (f + t)) atomicAdd(mismatches, 1u);   // the fold kernels' form (lep_enc5.h)
    for (int obs = 0; obs < 2; ++obs) {
        const uint32_t w = f | (t << 8) | (((f << 8) / (f + t)) << 16);
        if (lep3::bupd(w, obs) != branch_update(w, obs)) atomicAdd(mismatches, 1u);
    }
    {   // lep4::div_by against the hardware's expansion of `/`: 65025 divisors (quantiser x ICOS column), numerators at their multiples
        const int32_t d = (int32_t)((f * 257u + t) * (f & 1 ? 8192u : (t & 1 ? 2260u : 1u)));
        if (d >= 2) {
            const lep4::DivBy m = lep4::DivBy::of((uint32_t)d);
            const int32_t k = (int32_t)(0x7fffffff / d);
            for (int32_t n : {0, 1, -1, d, d - 1, -d, -d + 1, k * d, k * d - 1, -(k * d), 0x7fffffff, (int32_t)0x80000000, (int32_t)(f * 0x01000193u ^ t * 0x9e3779b9u)})
                if (lep4::div_by(n, m.mul, m.shift) != n / d) atomicAdd(mismatches, 1u);
        }
    }
}

// v4 decoder: serial part as uniform vector code, multi-bin interior windows, one merged edge round (lep_dec4.h).
// WAVES = waves per SIMD the register allocation is held to.
template <int WAVES, int SCMASK = LEP_DEC4_SCALAR>
__global__ __launch_bounds__(64, WAVES) void lep_decode_v4_kernel(const ImageDev* __restrict__ images, const SegDev* __restrict__ segs,
                                                           uint32_t* models, NSum* ns_area, const uint64_t* ns_offsets,
                                                           uint8_t* streams, uint32_t* stream_len, int32_t* status, uint32_t* bins) {
    __shared__ lep4::Dec4Shared sh;
    const int s = blockIdx.x, lane = threadIdx.x;
    const SegDev seg = segs[s];
    const ImageDev* img = images + seg.image;
    uint32_t* model = models + (size_t)s * kModelStride;
    NSum* ns = ns_area + ns_offsets[s];
    reset_segment_state<lep3::kModelWords>(model, ns, img->ns_total, lane);
    __syncthreads();
    lep4::Dec4WaveT<SCMASK> w;
#ifdef LEP_PROF
    w.prof_begin(&g_prof4[s & 8191][0]);
#endif
    int rc = w.run(img, seg, model, ns, &sh, streams + seg.stream_off, stream_len[seg.slot]);
#ifdef LEP_PROF
    w.prof_end();
#endif
    if (lane != 0) return;
    status[seg.slot] = rc;
    bins[seg.slot] = w.nbins;
}


// ---- the split-phase encoder (lep_enc5.h) -------------------------------------------------------------------------------------
// walk: one or two wavefronts per segment (count / emit / gather share the code; NW = 2: lep_enc5.h Walk5); LDS: two transposed
// coefficient tiles, the tile's entry payloads and ranks, the stream cursors
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void lep_enc5_walk_kernel(const ImageDev* __restrict__ images, const SegDev* __restrict__ segs, NSum* ns_all,
                                                         const uint64_t* __restrict__ ns_off, lep5::SegPlan5* plans, uint8_t* arena, uint16_t* bins,
                                                         uint32_t* counts, int part, int nparts) {
    lep5::Walk5Shared* sh = reinterpret_cast<lep5::Walk5Shared*>(lep5::lep5_lds);   // dynamic LDS (walk_lds_bytes(MODE) at launch)
    const int s = blockIdx.x;
    const SegDev seg = segs[s];
    lep5::SegPlan5* P = plans + s;
    if (MODE == lep5::kGather && P->status) return;
    lep5::Walk5<MODE, NW> w;
    const int rc = w.run(images + seg.image, seg, ns_all + ns_off[s], sh, P, arena, bins, (int)(threadIdx.x >> 6), part, nparts);
    if (threadIdx.x >= 64) return;
    if (MODE == lep5::kCount) lep5::export_counts(w, sh, counts + (size_t)s * lep5::kCountWords);
    if (threadIdx.x == 0) {
        if (MODE == lep5::kEmit) P->status = rc;
        if (MODE == lep5::kGather && part + 1 == nparts) P->nbins = w.nbins;
    }
}
// the stitched writer (lep_enc5.h "write, stitched"): PHASE 0 range (lane = chunk), 1 link (lane = segment), 2 code (lane = chunk),
// 3 stitch (lane = segment); K chunks per segment
template <int PHASE>
__global__ __launch_bounds__(64) void lep_enc5_wchunk_kernel(const lep5::SegPlan5* __restrict__ plans, const uint16_t* __restrict__ bins, const SegDev* __restrict__ segs,
                                                         int nseg, int K, lep5::WChunk5* recs, uint8_t* streams, uint32_t* stream_len, int32_t* status, uint32_t* nbins_out) {
    const int id = (int)(blockIdx.x * 64 + threadIdx.x);
    if (PHASE == 0 || PHASE == 2) {
        const int seg = id / K, k = id - seg * K;
        if (seg >= nseg) return;
        if (PHASE == 2) __builtin_amdgcn_s_setprio(3);
        if (PHASE == 0) lep5::wchunk_range_lane(plans[seg], bins, recs + (size_t)seg * K + k, k, K);
        else lep5::wchunk_code_lane(plans[seg], bins, segs[seg], streams, recs + (size_t)seg * K + k, k, K);
    } else {
        if (id >= nseg) return;
        if (PHASE == 1) { status[segs[id].slot] = 0; nbins_out[segs[id].slot] = plans[id].nbins; lep5::wchunk_link_lane(plans[id], bins, recs + (size_t)id * K, K); }
        else lep5::wchunk_stitch_lane(plans[id], segs[id], streams, stream_len, status, recs + (size_t)id * K, K);
    }
}
// counts -> per-segment layout (one thread per segment), then the prefix over the segments (one wavefront)
__global__ void lep_enc5_plan_kernel(const uint32_t* __restrict__ counts, lep5::SegPlan5* plans, int nseg) {
    const int s = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (s < nseg) lep5::plan_segment(counts + (size_t)s * lep5::kCountWords, plans + s);
}
__global__ __launch_bounds__(64) void lep_enc5_offsets_kernel(lep5::SegPlan5* plans, int nseg, uint64_t* totals) {
    uint64_t a = 0, b = 0;
    for (int s0 = 0; s0 < nseg; s0 += 64) {
        const int s = s0 + (int)threadIdx.x;
        const uint32_t ab = s < nseg ? plans[s].arena_bytes : 0u, bc = s < nseg ? plans[s].bins_cap : 0u;
        int oa, ob;
        const int va = (int)(ab >> 8), vb = (int)(bc >> 7);   // (both are multiples of 256 / 128: the scan runs on 32-bit values)
        const int ta = lepwave::wave_excl_scan(&va, &oa), tb = lepwave::wave_excl_scan(&vb, &ob);
        if (s < nseg) { plans[s].arena_off = a + ((uint64_t)(uint32_t)oa << 8); plans[s].bins_off = b + ((uint64_t)(uint32_t)ob << 7); }
        a += (uint64_t)(uint32_t)ta << 8; b += (uint64_t)(uint32_t)tb << 7;
    }
    if (threadIdx.x == 0) { totals[0] = a; totals[1] = b; }
}
__global__ void lep_enc5_fill_kernel(uint4* p, size_t n16, uint32_t v) {
    const uint4 x = make_uint4(v, v, v, v);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = x;
}
// bucket: the sparse records of a segment -> one stream per chain (one wavefront per segment)
__global__ __launch_bounds__(64) void lep_enc5_bucket_kernel(const lep5::SegPlan5* __restrict__ plans, uint8_t* arena) {
    __shared__ lep5::BucketShared sh;
    lep5::bucket_wave(plans + blockIdx.x, arena, &sh);
}
// fold, coefficient chains: block = (64 consecutive segments, stream); 18 KB of LDS
struct Fold5CoefShared { uint16_t slice[lep5::kCoefSlice * 64]; };   // 18 KB: eight wavefronts per CU
__global__ __launch_bounds__(64) void lep_enc5_fold_coef_kernel(const lep5::SegPlan5* __restrict__ plans, uint8_t* arena, int nseg, int groups) {
    __shared__ Fold5CoefShared shc;
    const int grp = (int)blockIdx.x % groups, job = (int)blockIdx.x / groups;   // job = ci * 630 + row * 10 + k, rows 0..62
    const int ci = job / 630, row = (job % 630) / 10, k = job % 10;
    const int sid = lep5::stream_id(ci, row, k), seg = grp * 64 + (int)threadIdx.x;
    const bool work = seg < nseg && !plans[seg].status && plans[seg].cnt[sid] != 0;
    if (!__ballot(work)) return;
    lep5::fold_coef_wave(plans, arena, grp * 64, nseg, sid, reinterpret_cast<lep5::FoldShared*>(&shc));
}
// fold, the other chains, by the LDS their Branches take (a kernel's resident wavefronts are what its largest job leaves room for):
//   small  sign chains (long dependent chains: first), threshold chains (Branches in HBM), edge non-zero counts     12 KB
//   big    DC chains, 7x7 non-zero counts                                                                          25 KB
struct Fold5SmallShared { uint16_t slice[lep5::kEdgeNzSlice * 64]; };
__global__ __launch_bounds__(64) void lep_enc5_fold_small_kernel(const lep5::SegPlan5* __restrict__ plans, uint8_t* arena, uint32_t* thresh_models, int nseg, int groups, int job0) {
    __shared__ Fold5SmallShared shs;
    lep5::FoldShared* sh = reinterpret_cast<lep5::FoldShared*>(&shs);
    const int grp = (int)blockIdx.x % groups;
    int job = (int)blockIdx.x / groups + job0;
    const int seg0 = grp * 64;
    if (job < 2) { lep5::fold_sign_wave(plans, arena, seg0, nseg, job, sh); return; }
    job -= 2;
    if (job < 12) { const int ci = job / 6, lt = 2 + job % 6; lep5::fold_thresh_wave(plans, arena, thresh_models, seg0, nseg, lep5::stream_id(ci, 63, lt), ci, sh); return; }
    job -= 12;
    lep5::fold_edgenz_wave(plans, arena, seg0, nseg, job / 16, (job / 8) & 1, job & 7, sh);   // 32 jobs
}
constexpr int kFold5SmallJobs = 2 + 12 + 32;
__global__ __launch_bounds__(64) void lep_enc5_fold_big_kernel(const lep5::SegPlan5* __restrict__ plans, uint8_t* arena, int nseg, int groups, int job0) {
    __shared__ lep5::FoldShared sh;
    const int grp = (int)blockIdx.x % groups;
    int job = (int)blockIdx.x / groups + job0;
    const int seg0 = grp * 64;
    if (job < 12) { lep5::fold_dc_wave(plans, arena, seg0, nseg, job, &sh); return; }
    job -= 12;
    lep5::fold_nz_wave(plans, arena, seg0, nseg, job / 10, job % 10, &sh);   // 20 jobs
}
constexpr int kFold5BigJobs = 12 + 20;
// write: lane = segment
__global__ __launch_bounds__(64) void lep_enc5_write_kernel(const lep5::SegPlan5* __restrict__ plans, const uint16_t* __restrict__ bins, const SegDev* __restrict__ segs,
                                                          int nseg, uint8_t* streams, uint32_t* stream_len, int32_t* status, uint32_t* nbins_out, uint8_t* arena,
                                                          int part, int nparts) {
    __builtin_amdgcn_s_setprio(3);   // (128 wavefronts that run beside the next part's gather: one long dependency chain each)
    const int seg0 = (int)blockIdx.x * 64, seg = seg0 + (int)threadIdx.x;
    if (seg < nseg) {
        if (part == 0) status[segs[seg].slot] = 0;
        if (part + 1 == nparts) nbins_out[segs[seg].slot] = plans[seg].nbins;
    }
    lep5::write_wave(plans, bins, segs, seg0, nseg, streams, stream_len, status, arena, part, nparts);
}

// JPEG Huffman re-encode of decoded frames: one wavefront per thread segment (lep_huff.h)
__global__ __launch_bounds__(64, 8) void lep_huffman_encode_kernel(const lephuff::HuffImage* __restrict__ images,
                                                                   const lephuff::HuffSegment* __restrict__ segs, uint8_t* out,
                                                                   uint32_t* out_len, lephuff::HuffEnd* ends) {
    __shared__ lephuff::HuffShared sh;
    const int s = blockIdx.x;
    const lephuff::HuffSegment seg = segs[s];
    if (seg.pad & lephuff::kHuffSegSimt) return;   // the lane-per-unit kernels below own this segment
    if (seg.pad & lephuff::kHuffSegRefuse) {       // a truncated file's segment nobody on the GPU takes: said so, the host re-coder's
        if (threadIdx.x == 0) {
            out_len[s] = 0;
            if (ends) { lephuff::HuffEnd e; memset(&e, 0, sizeof e); e.pad = lephuff::kHuffEndRefused; ends[s] = e; }
        }
        return;
    }
    lephuff::HuffWave w;
    const uint32_t n = w.run(images + seg.image, seg, &sh, out);
    if (threadIdx.x == 0) out_len[s] = n;
    if (ends) w.export_end(ends + s);
}
// (the bit buffers are cleared by a kernel of our own: hipMemsetAsync of 2.5 GB held the calling thread until everything queued
// on the stream in front of it had run -- the decode kernel of the chunk, 1.1 s -- measured with LEP_BATCH_TRACE)
__global__ __launch_bounds__(256) void lep_zero_kernel(uint4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = uint4{0u, 0u, 0u, 0u};
}
// ... with one lane per run of MCUs (lep_huff_simt.h): count / place / code / stuff
template <bool WRITE>
__global__ __launch_bounds__(64) void lep_huffman_simt_encode_units_kernel(const lephuff::HuffImage* __restrict__ images, const lephuff::HuffSegment* __restrict__ segs,
                                                                           lephuff::SimtEncSeg* es, const lephuff::SimtEncWave* __restrict__ waves,
                                                                           uint32_t* unit_bits, uint8_t* scratch) {
    __shared__ lephuff::SimtEncShared sh;
    const lephuff::SimtEncWave w = waves[blockIdx.x];
    lephuff::simt_enc_units<WRITE>(images, segs, es + w.eseg, &sh, unit_bits, scratch, w.first_unit);
}
__global__ __launch_bounds__(64) void lep_huffman_simt_encode_place_kernel(const lephuff::HuffImage* __restrict__ images, const lephuff::HuffSegment* __restrict__ segs,
                                                                           lephuff::SimtEncSeg* es, uint32_t* unit_bits, uint32_t* unit_plain) {
    lephuff::simt_enc_place(images, segs, es + blockIdx.x, unit_bits, unit_plain);
}
__global__ __launch_bounds__(64) void lep_huffman_simt_encode_stuff_kernel(const lephuff::HuffImage* __restrict__ images, const lephuff::HuffSegment* __restrict__ segs,
                                                                           const lephuff::SimtEncSeg* __restrict__ es, uint8_t* scratch, uint8_t* out, uint32_t* out_len,
                                                                           lephuff::HuffEnd* ends) {
    lephuff::simt_enc_stuff(images, segs, es[blockIdx.x], scratch, out, out_len, ends);
}

// progressive files: one wavefront per (image, scan) (lep_huffprog.h)
__global__ __launch_bounds__(64, 8) void lep_huffman_progressive_encode_kernel(const lephuff::ProgImage* __restrict__ images,
                                                                               const lephuff::ProgScan* __restrict__ scans, uint8_t* out,
                                                                               uint32_t* corr, uint32_t* out_len) {
    __shared__ lephuff::ProgShared sh;
    const lephuff::ProgScan* sc = scans + blockIdx.x;
    if (sc->pad & (lephuff::kProgScanSimt | lephuff::kProgScanSeq)) return;   // the lane-per-unit kernels below / the sequential scan encoders own this scan
    lephuff::ProgWave w;
    const uint32_t n = w.run_scan(images + sc->image, sc, &sh, out, corr);
    if (threadIdx.x == 0) out_len[blockIdx.x] = n;
}
// scans of sequential frames, written by the sequential scan encoders as segments of their own: their byte counts to the scans' places, with
// the progressive writers' mark (bit 31) on a scan that outgrew its slot
__global__ void lep_huffprog_seq_lens_kernel(const uint32_t* __restrict__ which, const uint32_t* __restrict__ cap, const uint32_t* __restrict__ lens,
                                             const lephuff::HuffEnd* __restrict__ ends, int n, uint32_t* out_len) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const bool over = ends[i].attempted > cap[i] || (ends[i].pad & (lephuff::kHuffEndCut | lephuff::kHuffEndRefused)) != 0;
    out_len[which[i]] = lens[i] | (over ? 0x80000000u : 0u);
}
// ... with one lane per run of blocks (lep_huffprog_simt.h): count / place / code / stuff
template <bool WRITE>
__global__ __launch_bounds__(64) void lep_huffprog_simt_units_kernel(const lephuff::ProgImage* __restrict__ images, const lephuff::ProgScan* __restrict__ scans,
                                                                     const lephuff::ProgSimtScan* __restrict__ ps, const lephuff::ProgSimtWave* __restrict__ waves,
                                                                     uint32_t* unit_words, size_t units, uint8_t* scratch) {
    __shared__ lephuff::ProgSimtShared sh;
    const lephuff::ProgSimtWave w = waves[blockIdx.x];
    lephuff::ProgSimtUnits U;
    U.set(unit_words, units);
    lephuff::prog_simt_units<WRITE>(images, scans, ps + w.pscan, &sh, U, scratch, w.first_unit);
}
__global__ __launch_bounds__(64) void lep_huffprog_simt_place_kernel(const lephuff::ProgScan* __restrict__ scans, lephuff::ProgSimtScan* ps, uint32_t* unit_words, size_t units) {
    lephuff::ProgSimtUnits U;
    U.set(unit_words, units);
    lephuff::prog_simt_place(scans, ps + blockIdx.x, U);
}
// the bit buffers are cleared as far as the scans reach (pass 2 knows; the buffers are sized by what a scan MAY need, ten times that)
__global__ __launch_bounds__(256) void lep_huffprog_simt_zero_kernel(const lephuff::ProgSimtScan* __restrict__ ps, uint8_t* scratch, uint32_t chunk16) {
    const lephuff::ProgSimtScan s = ps[blockIdx.x];
    const uint32_t need16 = (uint32_t)std::min<uint64_t>(((uint64_t)s.total_bits + 7) / 8 / 16 + 2, s.buf_bytes / 16);
    uint4* p = reinterpret_cast<uint4*>(scratch + s.buf_off);
    const uint32_t i0 = blockIdx.y * chunk16, i1 = i0 + chunk16 < need16 ? i0 + chunk16 : need16;
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += blockDim.x) p[i] = uint4{0u, 0u, 0u, 0u};
}
__global__ void lep_huffprog_simt_assign_kernel(const lephuff::ProgSimtRegion* __restrict__ regions, int nregion, lephuff::ProgSimtScan* ps) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < nregion) lephuff::prog_simt_assign(regions[i], ps);
}
__global__ __launch_bounds__(64) void lep_huffprog_simt_stuff_kernel(const lephuff::ProgImage* __restrict__ images, const lephuff::ProgScan* __restrict__ scans,
                                                                     const lephuff::ProgSimtScan* __restrict__ ps, uint8_t* scratch, uint8_t* out, uint32_t* out_len) {
    lephuff::prog_simt_stuff(images, scans, ps[blockIdx.x], scratch, out, out_len);
}

// progressive files, encode direction: one wavefront per (image, scan) of one dependency level (lep_huffprogdec.h)
__global__ __launch_bounds__(64, 8) void lep_huffman_progressive_decode_kernel(const lephuff::ProgDecScan* __restrict__ scans, lephuff::HuffDecRow* rows) {
    __shared__ lephuff::HuffDecShared sh;
    lephuff::ProgDecWave w;
    w.run_scan<false>(scans + blockIdx.x, &sh, rows);
}
// the same two launches with the window of speculative codes (lep_huffprogdec_win.h) for the scans it takes (ProgDecScan.pad)
__global__ __launch_bounds__(64, 2) void lep_huffprogdec_win_kernel(const lephuff::ProgDecScan* __restrict__ scans, lephuff::HuffDecRow* rows) {
    __shared__ lephuff::ProgWinShared ws;
    __shared__ lephuff::HuffDecShared sh;
    const lephuff::ProgDecScan* sc = scans + blockIdx.x;
    if (sc->pad & lephuff::kProgDecWin) { lephuff::ProgWinWave w; w.run_scan_win<false>(sc, &ws, rows); }
    else { lephuff::ProgDecWave w; w.run_scan<false>(sc, &sh, rows); }
}
__global__ __launch_bounds__(64, 2) void lep_huffprogdec_win_pipelined_kernel(const lephuff::ProgDecScan* __restrict__ scans, lephuff::HuffDecRow* rows,
                                                                            const lephuff::ProgDeps* __restrict__ deps, uint32_t* progress, uint32_t* ticket) {
    __shared__ lephuff::ProgWinShared ws;
    __shared__ lephuff::HuffDecShared sh;
    uint32_t t = 0;
    if (threadIdx.x == 0) t = atomicAdd(ticket, 1u);
    const int k = __builtin_amdgcn_readfirstlane((int)t);
    const lephuff::ProgDecScan* sc = scans + k;
    // (the scans of a file run side by side, two or three wavefronts to a SIMD, and each is one chain bound by its own instruction
    // issue: the long ones -- luma AC scans, the file's critical path -- are served first)
    if (sc->to != 0) { if (sc->cmp[0] == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1); }
    if (sc->pad & lephuff::kProgDecWin) { lephuff::ProgWinWave w; w.run_scan_win<true>(sc, &ws, rows, deps + k, progress, k); }
    else { lephuff::ProgDecWave w; w.run_scan<true>(sc, &sh, rows, deps + k, progress, k); }
}
// ... all levels in ONE launch: a scan waits, MCU row by MCU row, for the scans of its file it follows (lep_huffprogdec.h ProgDeps)
// (128 VGPRs: at 64 the waiting code's spills trip a register-pair alignment check in this compiler's backend; a launch of this
// kind is small, what it needs is a short chain)
// A workgroup does not take the scan of its own index: it draws a ticket when it STARTS and takes that scan.  A scan only waits for
// scans of a lower index (prog_scan_deps), i.e. for tickets drawn before its own -- workgroups that are running or done, whatever
// order the hardware starts workgroups in.  (Round 3 relied on index-order dispatch, which holds on this chip but is nobody's
// promise: ADVICE round 3.)
__global__ __launch_bounds__(64, 4) void lep_huffman_progressive_pipelined_kernel(const lephuff::ProgDecScan* __restrict__ scans, lephuff::HuffDecRow* rows,
                                                                                 const lephuff::ProgDeps* __restrict__ deps, uint32_t* progress, uint32_t* ticket) {
    __shared__ lephuff::HuffDecShared sh;
    uint32_t t = 0;
    if (threadIdx.x == 0) t = atomicAdd(ticket, 1u);
    const int k = __builtin_amdgcn_readfirstlane((int)t);
    lephuff::ProgDecWave w;
    w.run_scan<true>(scans + k, &sh, rows, deps + k, progress, k);
}

// JPEG Huffman scan decode: one wavefront per image (lep_huffdec.h)
// <= 64 VGPRs and 4.5 KB of LDS: one of these waves fits on a SIMD beside seven coder waves, and it runs at raised priority
// there (it is one long dependency chain per image; the coder waves around it are the throughput work)
__global__ __launch_bounds__(64, 8) void lep_huffman_decode_kernel(const lephuff::HuffDecImage* __restrict__ images, lephuff::HuffDecRow* rows) {
    __shared__ lephuff::HuffDecShared sh;
    __builtin_amdgcn_s_setprio(3);
    lephuff::HuffDecWave w;
    w.run(images + blockIdx.x, &sh, rows);
}

// One lane per subsequence (lep_huffdec_simt.h): guess / settle / place / write; a wavefront's 64 lanes are 64 consecutive
// subsequences of one image, whose tables the wavefront keeps in LDS.
__global__ __launch_bounds__(64) void lep_huffman_simt_settle_kernel(const lephuff::HuffDecImage* __restrict__ images, lephuff::SimtImage* si, const lephuff::SimtWave* waves,
                                                                     const lephuff::SimtSub* in, lephuff::SimtSub* out, int settle) {
    __shared__ lephuff::SimtShared sh;
    const lephuff::SimtWave w = waves[blockIdx.x];
    lephuff::simt_guess_or_settle(images + w.image, &sh, si + w.image, in + si[w.image].first, out + si[w.image].first, w.first_sub, settle);
}
__global__ __launch_bounds__(64) void lep_huffman_simt_place_kernel(const lephuff::HuffDecImage* __restrict__ images, lephuff::SimtImage* si, const lephuff::SimtSub* sub,
                                                                    lephuff::SimtPlace* place, int passes, lephuff::HuffDecRow* rows) {
    const int i = (int)blockIdx.x;
    lephuff::simt_place(images + i, si + i, sub + si[i].first, place + si[i].first, passes, rows);
}
__global__ __launch_bounds__(64) void lep_huffman_simt_write_kernel(const lephuff::HuffDecImage* __restrict__ images, lephuff::SimtImage* si, const lephuff::SimtWave* waves,
                                                                    const lephuff::SimtSub* sub, const lephuff::SimtPlace* place, lephuff::HuffDecRow* rows) {
    __shared__ lephuff::SimtShared sh;
    __shared__ lephuff::SimtTile tile;
    const lephuff::SimtWave w = waves[blockIdx.x];
    lephuff::simt_write(images + w.image, &sh, &tile, si + w.image, sub + si[w.image].first, place + si[w.image].first, rows, w.first_sub);
}
__global__ void lep_huffman_simt_finish_kernel(const lephuff::HuffDecImage* __restrict__ images, int nimg, lephuff::HuffDecRow* rows, const lephuff::SimtImage* si) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= nimg) return;
    lephuff::HuffDecRow* last = rows + images[i].rows_off + images[i].mcuv;
    int status = si[i].status & 0x3fffff;
    if (last->aux == lephuff::kHuffDecRowUnwritten) { status |= 2; last->aux = 255; }   // no lane of the write pass got to the final record: irregular
    if (images[i].flags & lephuff::kHuffDecRstTable) {      // restart intervals: the pad byte is what all intervals agreed on
        const int pad = lephuff::simt_intervals_pad(si + i, &status);
        last->aux = pad | (status << 8);
        return;
    }
    last->aux = (last->aux & (255 | lephuff::kHuffDecRowTruncated)) | (status << 8);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
struct lep_gpu {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    bool released = false;   // device side already given back (by lep_gpu_destroy or by the exit handler)
    int enc5_min = 8;        // launches of at least this many segments take the split-phase encoder (lep_enc5.h); LEP_ENC5_MIN (0 = never).  64 until
                             // round 4: with the stitched writer a single 4K image (8 segments) takes 109 ms through it against 301 ms through the
                             // two-wavefront single-kernel encoder (profiles/r05k_latency_stitched_writer_ab.json)
    hipStream_t stream2 = nullptr, stream3 = nullptr;   // the split-phase encoder folds its long chains beside its many short ones
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join3 = nullptr;
    int huffprog_pipeline = 1;          // LEP_HUFFPROG_PIPELINE=0: progressive scan decode level by level, whatever the launch size
    int huffprog_split = 0;             // LEP_HUFFPROG_SPLIT=1: level-by-level launches split by kind of scan (measurement aid)
    int huffprogdec_win = 1;            // LEP_HUFFPROGDEC_WIN=0: progressive scans decoded by lep_huffprogdec.h's uniform vector code only
    int huffprog_simt = 1;              // LEP_HUFFPROG_SIMT=0: every progressive scan's bytes from the wavefront-per-scan kernel (lep_huffprog.h)
    void* d_huffseq[2] = {nullptr, nullptr}; size_t huffseq_bytes[2] = {0, 0};             // scans of sequential frames in a progressive launch: which / caps / byte counts / end states
    void* d_huffprogsimt[2] = {nullptr, nullptr}; size_t huffprogsimt_bytes[2] = {0, 0};   // lep_huffprog_simt.h: descriptors, unit arrays, bit buffers (one per arena set)
    int huffenc_simt = 1;               // LEP_HUFFENC_SIMT=0: every segment's scan bytes from the wavefront-per-segment kernel (lep_huff.h)
    void* d_huffenc[2] = {nullptr, nullptr}; size_t huffenc_bytes[2] = {0, 0};   // lep_huff_simt.h: segment / wave descriptors, unit bit counts, bit buffers (one per arena set, like d_huff)
    int simt_sub_bits = 0;              // LEP_HUFFDEC_SIMT_BITS: bits per subsequence of the lane-per-subsequence scan decoder (0 = from the launch's size)
    int huffprog_pipeline_max = 16384;  // LEP_HUFFPROG_PIPELINE_MAX: scans per launch up to which the levels go out as one pipelined launch (measured to 10240:
                                        // 1024 4K files, 798 -> 938 MB/s; a workgroup takes the scan of the ticket it draws when it starts, so a scan's predecessors are always running or done)
    int enc5_waves = 2;      // LEP_ENC5_WAVES: wavefronts per segment in the split-phase walks (1 | 2)
    int enc5_parts = 8;      // LEP_ENC5_PARTS: gather / write in this many parts (1..8), a part written while the next is gathered
                             // (MI355X, 1024 x 4K: 1 part 569 ms per launch, 4 parts 481, 8 parts 475)
    hipEvent_t ev_part[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    long enc5_wlanes = 131072;  // LEP_ENC5_WLANES: chunk lanes a launch of the stitched writer may have (segments x chunks per segment)
    int enc5_wmaxseg = 4096;    // LEP_ENC5_WMAXSEG: launches of more segments keep the lane-per-segment writer beside gather
    int enc5_wchunks = 64;   // LEP_ENC5_WCHUNKS: the stitched writer for launches that leave lanes free: up to this many chunks per segment (power of
                             // two; 0 = always the lane-per-segment writer beside gather)
    int enc5_gather_wgs = 0; // LEP_ENC5_GATHER_WGS: resident gather workgroups per CU held to this (through the LDS a launch asks for)
    int enc5_fold_apart = 0; // LEP_ENC5_FOLD_APART: the fold launches one after the other, a launch per kind of chain (for the profiler)
    size_t enc5_scratch_max = ~(size_t)0;   // LEP_ENC5_SCRATCH_MAX (bytes): a launch that needs more takes the single-kernel encoder (tests: the out-of-memory path)
    hipEvent_t ev_stage[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // stage boundaries of the last split-phase launch
    int nstage = 0;
    int enc_waves = 0;       // the same choice for the encoder (LEP_ENC_WAVES = 4 | 8; 2 = the two-wavefronts-per-segment kernel)
    int enc_pair_max = 1280; // launches of up to this many segments take the two-wave encoder: its 128-thread workgroups are resident 6 per
                             // CU (1536 on the chip), one pass; measured (profiles/r02d_latency_sweep.json, 4K images): 1 .. 128 images
                             // 300-343 ms against 450-496 ms for one wavefront per segment, 256 images 650 against 510.  LEP_ENC_PAIR_MAX
    int dec_company = 0;     // lep_gpu_expect_company: decode launches will overlap with their neighbours on another stream
    int dec_waves = 0;       // register-budget build of the decoder: 0 = by batch size (8 waves per SIMD / 64 VGPRs once a launch can
                             // fill them, else 4 / 128 VGPRs, no spills); LEP_DEC_WAVES = 4 | 8 forces one
    std::string err;
    const char* last_kernel = "";   // name of the kernel the most recent launch used
    // Device memory the object owns (round 4): every grow-only workspace below is a virtual address range of its own (hipMemAddressReserve)
    // into which physical chunks of 512 MB (hipMemCreate) are mapped as it grows -- growing maps more chunks behind what is there, nothing
    // is freed and taken again.  A workspace that is given back (lep_gpu_trim) is unmapped and its chunks go to the object's POOL, from
    // which the next workspace that grows takes them: memory moves between the encoder's scratch, the decoder's models and the rings
    // without passing through the driver.  Why that matters (MI355X, scripts/proto/vmm_costs.hip, profiles/r05m_*): the driver CLEARS
    // what it hands out -- hipMalloc of 96 GB takes 2.9 - 4.1 s, hipMemCreate the same per byte -- while unmapping 96 GB of chunks takes
    // 19 ms and mapping them again 14 ms.  Round 3's "2x cliff" after a trim was the next launch paying ~40 ms per GB for ~140 GB again.
    // Only lep_gpu_release_memory (and an allocation that fails elsewhere) hands the pool to the driver.  LEP_VMM=0: plain hipMalloc /
    // hipFree as before (A/B).
    struct VBuf { char* va = nullptr; size_t reserved = 0, mapped = 0; std::vector<hipMemGenericAllocationHandle_t> chunks; };
    struct Vmm {
        bool on = false;
        size_t chunk = (size_t)512 << 20;   // (64 MB chunks cost the encoder 2 %: 470 -> 480 ms, more page-table fragments; 512 MB: as one hipMalloc -- profiles/r05q_*)
        hipMemAllocationProp prop;
        hipMemAccessDesc access;
        std::map<void**, VBuf> bufs;   // keyed by the member that holds the workspace's pointer
        std::vector<hipMemGenericAllocationHandle_t> pool;   // chunks of workspaces that were given back (lep_gpu_trim): unmapped, still ours --
                                                             // the next workspace that grows, of whatever kind, maps them again
        size_t mapped_total = 0, creates = 0, releases = 0;
    } vmm;
    // grow-only device workspace of a coder launch (models, neighbour summaries, descriptors).  There are two sets so that two
    // launches may be in flight at once on different streams (lep_gpu_use_arena: the next chunk's coder kernel starts in the
    // wave slots that the long segments of the current one leave free); everything else uses set 0.
    struct Arena {
        void* d_models = nullptr; size_t models_bytes = 0;
        void* d_ns = nullptr; size_t ns_bytes = 0;
        void* d_meta = nullptr; size_t meta_bytes = 0;      // ImageDev[] | SegDev[] | ns_offsets[] | bins[]
    } arena[2];
    int cur = 0;
    // split-phase encoder scratch: ONE set (a 4K image takes ~140 MB of it), shared by the two arena sets -- a split-phase launch
    // waits for the one before it (ev_enc5_done), which fills the chip on its own anyway
    struct Enc5Scratch {
        void* d_plans = nullptr; size_t plans_bytes = 0;       // SegPlan5[] | counts | totals
        void* d_entries = nullptr; size_t entries_bytes = 0;   // the chains' entry streams, records, places
        void* d_binlist = nullptr; size_t binlist_bytes = 0;   // the segments' bin lists
        void* d_wchunks = nullptr; size_t wchunks_bytes = 0;   // the stitched writer's records, one per (segment, chunk)
    } enc5;
    hipEvent_t ev_enc5_done = nullptr;
    uint32_t* d_bins = nullptr;
    std::vector<uint32_t> h_bins;
    // host-variant staging
    void* d_blocks = nullptr; size_t blocks_bytes = 0;
    void* d_streams = nullptr; size_t streams_bytes = 0;
    void* d_lens = nullptr; size_t lens_bytes = 0;
    // HuffImage[] | HuffSegment[], one per arena set: the batch compressor's round-trip check writes the scans of chunk k and of chunk k + 1 on two
    // streams (lep_gpu_use_arena picks the set), and a short chunk k + 1 reaches its scan encoder before a long chunk k does -- with one buffer its
    // descriptors replaced the ones chunk k's kernels were still to read (a memory fault in tests/test_gpu_parity.py::*overlapped_launches, once,
    // round 6: the scans of sequential frames in several scans come through here a second time per chunk)
    void* d_huff[2] = {nullptr, nullptr}; size_t huff_bytes[2] = {0, 0};
    void* d_huffprog[2] = {nullptr, nullptr}; size_t huffprog_bytes[2] = {0, 0};   // ProgImage[] | ProgScan[], one per arena set (two launches on two streams)
    struct Staging { void* host = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool used = false; };
    Staging staging[16];                // pinned ring for descriptor uploads (upload()): nothing on a launch path waits for its stream
    int staging_next = 0;
    std::vector<void*> staging_retired;
    void* h_huffprog[2] = {nullptr, nullptr}; size_t h_huffprog_bytes[2] = {0, 0};   // pinned staging of the same: the upload does not wait for the stream
    int huffprog_turn = 0;   // the two sets are used in turn: at most two launches are ever in flight (lep_batch.hip queues chunk k+1 before it fetches chunk k)
    void* d_huffprogdec = nullptr; size_t huffprogdec_bytes = 0;   // ProgDecScan[]
    void* d_huffdec = nullptr; size_t huffdec_bytes = 0;   // HuffDecImage[]
    void* d_huffpar = nullptr; size_t huffpar_bytes = 0;   // the lane-per-subsequence scan decoder's records (SimtImage | SimtWave | SimtSub x 2 | SimtPlace)
    void* d_scan = nullptr; size_t scan_bytes = 0;      // scan bytes of the Huffman encoder (host variant)
    void* d_scanlen = nullptr; size_t scanlen_bytes = 0;
};

#define HIPCHK(g, call)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (g)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return LEP_GPU_ERROR;                                                              \
        }                                                                                      \
    } while (0)

// ---- workspaces ------------------------------------------------------------------------------------------------------------------
static void vmm_init(lep_gpu* g) {
    g->vmm.on = false;
    if (const char* e = getenv("LEP_VMM")) if (atoi(e) == 0) return;
    int ok = 0;
    if (hipDeviceGetAttribute(&ok, hipDeviceAttributeVirtualMemoryManagementSupported, g->device) != hipSuccess || !ok) { (void)hipGetLastError(); return; }
    memset(&g->vmm.prop, 0, sizeof g->vmm.prop);
    g->vmm.prop.type = hipMemAllocationTypePinned;
    g->vmm.prop.location.type = hipMemLocationTypeDevice;
    g->vmm.prop.location.id = g->device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &g->vmm.prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) { (void)hipGetLastError(); return; }
    if (const char* e = getenv("LEP_VMM_CHUNK_MB")) if (atoi(e) > 0) g->vmm.chunk = (size_t)atoi(e) << 20;
    g->vmm.chunk = ((g->vmm.chunk + gran - 1) / gran) * gran;
    g->vmm.access.location = g->vmm.prop.location;
    g->vmm.access.flags = hipMemAccessFlagsProtReadWrite;
    g->vmm.on = true;
}
// gives a workspace back: its chunks unmapped and pooled (or released to the driver), the address range kept for the next time it grows
static void vmm_unmap(lep_gpu* g, lep_gpu::VBuf& b, bool to_driver = false) {
    if (b.mapped) (void)hipMemUnmap(b.va, b.mapped);
    for (hipMemGenericAllocationHandle_t h : b.chunks) {
        if (to_driver) { (void)hipMemRelease(h); ++g->vmm.releases; }
        else g->vmm.pool.push_back(h);
    }
    g->vmm.mapped_total -= b.mapped;
    b.chunks.clear(); b.mapped = 0;
}
static void vmm_drain_pool(lep_gpu* g) {
    for (hipMemGenericAllocationHandle_t h : g->vmm.pool) { (void)hipMemRelease(h); ++g->vmm.releases; }
    g->vmm.pool.clear();
}
static void dev_release(lep_gpu* g, void** p, size_t* have) {
    if (g->vmm.on) {
        auto it = g->vmm.bufs.find(p);
        if (it != g->vmm.bufs.end()) { vmm_unmap(g, it->second); *p = nullptr; if (have) *have = 0; return; }
    }
    if (*p) (void)hipFree(*p);
    *p = nullptr; if (have) *have = 0;
}
static void vmm_destroy(lep_gpu* g) {
    for (auto& kv : g->vmm.bufs) {
        vmm_unmap(g, kv.second, true);
        if (kv.second.va) (void)hipMemAddressFree(kv.second.va, kv.second.reserved);
        *kv.first = nullptr;
    }
    g->vmm.bufs.clear();
    vmm_drain_pool(g);
}
// before an allocation failure is reported: the cached memory no launch that is being set up depends on -- the split-phase
// encoder's scratch (its launches take the single-kernel encoder when they cannot have it) and the other arena set's models
static void release_idle_caches(lep_gpu* g) {
    (void)hipDeviceSynchronize();
    lep_gpu::Arena& O = g->arena[g->cur ^ 1];
    void** ps[] = {&g->enc5.d_entries, &g->enc5.d_binlist, &O.d_models, &O.d_ns};
    size_t* ns[] = {&g->enc5.entries_bytes, &g->enc5.binlist_bytes, &O.models_bytes, &O.ns_bytes};
    for (int i = 0; i < 4; ++i) dev_release(g, ps[i], ns[i]);
}
static int vmm_ensure(lep_gpu* g, void** p, size_t* have, size_t need, bool may_release) {
    lep_gpu::Vmm& V = g->vmm;
    lep_gpu::VBuf& b = V.bufs[p];
    const size_t want = ((need + V.chunk - 1) / V.chunk) * V.chunk;
    if (want > b.reserved) {   // a longer address range (what the workspace held is not kept: every launch fills it anew)
        // The old range is unmapped and freed here, and launches do not wait for their streams any more (upload() ring; lep_batch.hip
        // queues chunk k + 1 while chunk k's kernels run): a kernel still in flight would lose its mapping under its feet.  hipFree
        // used to synchronise implicitly; this path has to ask.  Rare: growth inside the reservation maps behind.
        if (b.mapped) (void)hipDeviceSynchronize();
        vmm_unmap(g, b);
        if (b.va) (void)hipMemAddressFree(b.va, b.reserved);
        b.va = nullptr; b.reserved = 0; *p = nullptr; *have = 0;
        size_t res = want + want / 2;
        res = ((std::max(res, (size_t)256 << 20) + V.chunk - 1) / V.chunk) * V.chunk;
        void* va = nullptr;
        HIPCHK(g, hipMemAddressReserve(&va, res, 0, nullptr, 0));
        b.va = (char*)va; b.reserved = res;
    }
    const size_t old = b.mapped;
    while (b.mapped < want) {
        hipMemGenericAllocationHandle_t h;
        if (!V.pool.empty()) { h = V.pool.back(); V.pool.pop_back(); --V.creates; }
        else if (hipMemCreate(&h, V.chunk, &V.prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (may_release) { may_release = false; release_idle_caches(g); if (!V.pool.empty()) continue; }   // the idle workspaces' chunks are in the pool now
            {   // no room: what this call mapped is given back, the workspace stays as it was
                while (b.mapped > old) { b.mapped -= V.chunk; (void)hipMemUnmap(b.va + b.mapped, V.chunk); V.pool.push_back(b.chunks.back()); b.chunks.pop_back(); V.mapped_total -= V.chunk; }
                g->err = "hipMemCreate: out of device memory";
                return LEP_GPU_ERROR;
            }
        }
        ++V.creates;
        if (hipMemMap(b.va + b.mapped, V.chunk, 0, h, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipMemRelease(h); g->err = "hipMemMap failed"; return LEP_GPU_ERROR; }
        b.chunks.push_back(h); b.mapped += V.chunk; V.mapped_total += V.chunk;
    }
    if (b.mapped > old) HIPCHK(g, hipMemSetAccess(b.va + old, b.mapped - old, &V.access, 1));
    *p = b.va; *have = b.mapped;
    return 0;
}
static int ensure(lep_gpu* g, void** p, size_t* have, size_t need, bool may_release = true) {
    if (*have >= need) return 0;
    // the large workspaces (models, rings, scratch: GBs) are owned address ranges with pooled chunks; descriptors and other small ones
    // (a chunk would be mostly slack) stay plain allocations
    if (g->vmm.on && (need >= ((size_t)32 << 20) || g->vmm.bufs.count(p))) {
        if (*p && !g->vmm.bufs.count(p)) { HIPCHK(g, hipFree(*p)); *p = nullptr; *have = 0; }
        return vmm_ensure(g, p, have, need, may_release);
    }
    if (*p) HIPCHK(g, hipFree(*p));
    *p = nullptr; *have = 0;
    size_t want = need + need / 8;
    if (hipMalloc(p, want) != hipSuccess) {   // (the head room is a convenience: without it before giving up)
        (void)hipGetLastError();
        *p = nullptr; want = need;
        if (hipMalloc(p, want) != hipSuccess) {
            (void)hipGetLastError();
            *p = nullptr;
            if (may_release) release_idle_caches(g);
            HIPCHK(g, hipMalloc(p, want));
        }
    }
    *have = want;
    return 0;
}

#include "lep_derive.h"


// The split-phase encoder (lep_enc5.h): count -> plan -> emit -> fold -> gather -> write.  One host synchronisation in the
// middle: the arena sizes come out of the count pass.  Stage boundaries are recorded as events (lep_gpu_last_stage_ms).
constexpr int kEnc5NoMemory = -1000;   // launch_enc5: the scratch could not be allocated (nothing has been written for the caller yet)
static int launch_enc5(lep_gpu* g, const ImageDev* d_img, const SegDev* d_seg, const uint64_t* d_nsoff, int nseg, uint8_t* d_streams,
                       uint32_t* d_stream_len, int32_t* d_status, hipStream_t st) {
    lep_gpu::Arena& A = g->arena[g->cur];
    lep_gpu::Enc5Scratch& E = g->enc5;
    if (!g->stream2) {
        HIPCHK(g, hipEventCreateWithFlags(&g->ev_enc5_done, hipEventDisableTiming));
        for (auto& e : g->ev_part) HIPCHK(g, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIPCHK(g, hipStreamCreateWithFlags(&g->stream2, hipStreamNonBlocking));
        HIPCHK(g, hipStreamCreateWithFlags(&g->stream3, hipStreamNonBlocking));
        HIPCHK(g, hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming));
        HIPCHK(g, hipEventCreateWithFlags(&g->ev_join, hipEventDisableTiming));
        HIPCHK(g, hipEventCreateWithFlags(&g->ev_join3, hipEventDisableTiming));
        for (auto& e : g->ev_stage) HIPCHK(g, hipEventCreate(&e));
    }
    const size_t o_counts = ((size_t)nseg * sizeof(lep5::SegPlan5) + 255) & ~(size_t)255,
                 o_tot = o_counts + (((size_t)nseg * lep5::kCountWords * 4 + 255) & ~(size_t)255);
    HIPCHK(g, hipStreamWaitEvent(st, g->ev_enc5_done, 0));   // (the scratch is the previous split-phase launch's until its writer is done)
    if (int rc = ensure(g, &E.d_plans, &E.plans_bytes, o_tot + 256)) return rc;
    lep5::SegPlan5* plans = (lep5::SegPlan5*)E.d_plans;
    uint32_t* counts = (uint32_t*)((char*)E.d_plans + o_counts);
    uint64_t* d_tot = (uint64_t*)((char*)E.d_plans + o_tot);
    const int groups = (nseg + 63) / 64;
    g->nstage = 0;
    HIPCHK(g, hipEventRecord(g->ev_stage[0], st));
    const bool two = g->enc5_waves == 2;
    const int nparts = std::min(std::max(g->enc5_parts, 1), (int)lep5::kMaxParts);
    auto walk = [&](int mode, uint8_t* entries, uint16_t* binlist, int part = 0) {
        size_t lds = lep5::walk_lds_bytes(mode);
        if (mode == lep5::kGather && g->enc5_gather_wgs > 0) lds = std::max(lds, (size_t)(160 * 1024 / g->enc5_gather_wgs) & ~(size_t)255);   // (measurement aid: fewer resident workgroups)
#define LEP_WALK(MODE, NW) hipLaunchKernelGGL((lep_enc5_walk_kernel<MODE, NW>), dim3(nseg), dim3(64 * NW), lds, st, d_img, d_seg, (NSum*)A.d_ns, d_nsoff, plans, entries, binlist, counts, part, nparts)
        if (mode == lep5::kCount) { if (two) LEP_WALK(lep5::kCount, 2); else LEP_WALK(lep5::kCount, 1); }
        else if (mode == lep5::kEmit) { if (two) LEP_WALK(lep5::kEmit, 2); else LEP_WALK(lep5::kEmit, 1); }
        else { if (two) LEP_WALK(lep5::kGather, 2); else LEP_WALK(lep5::kGather, 1); }
#undef LEP_WALK
    };
    walk(lep5::kCount, nullptr, nullptr);
    hipLaunchKernelGGL(lep_enc5_plan_kernel, dim3(groups), dim3(64), 0, st, (const uint32_t*)counts, plans, nseg);
    hipLaunchKernelGGL(lep_enc5_offsets_kernel, dim3(1), dim3(64), 0, st, plans, nseg, d_tot);
    // the threshold Branches are the only model state in HBM: 2 MB per segment, reset while the count pass is looked at
    hipLaunchKernelGGL(lep_enc5_fill_kernel, dim3(4096), dim3(256), 0, st, (uint4*)A.d_models, (size_t)nseg * lep5::kThreshWords / 4, kBranchInit);
    uint64_t tot[2] = {0, 0};
    HIPCHK(g, hipMemcpyAsync(tot, d_tot, sizeof tot, hipMemcpyDeviceToHost, st));
    HIPCHK(g, hipStreamSynchronize(st));
    // no room for the scratch (~140 MB per 4K image): the single-kernel encoder takes the launch -- it needs none
    if ((size_t)tot[0] + (size_t)tot[1] * 2 > g->enc5_scratch_max || ensure(g, &E.d_entries, &E.entries_bytes, (size_t)tot[0] + 256, false) || ensure(g, &E.d_binlist, &E.binlist_bytes, (size_t)tot[1] * 2 + 256, false)) {
        (void)hipGetLastError();
        g->err.clear();
        static bool told = false;
        if (!told) { told = true; fprintf(stderr, "lepton-mi355x: no room for the split-phase encoder's scratch (%.1f GB for %d segments): single-kernel encoder\n", ((double)tot[0] + 2.0 * (double)tot[1]) / 1e9, nseg); }
        return kEnc5NoMemory;
    }
    // (Segment groups on streams of their own -- one group's writer and folds beside the next group's walks -- were measured
    // and dropped: MI355X, 1024 x 4K, profiles/r04l_*: 1 group 784 ms, 2: 875, 4: 1232; launches this large do not share the chip.)
    {
        HIPCHK(g, hipEventRecord(g->ev_stage[1], st));
        walk(lep5::kEmit, (uint8_t*)E.d_entries, nullptr);
        HIPCHK(g, hipEventRecord(g->ev_stage[2], st));
        hipLaunchKernelGGL(lep_enc5_bucket_kernel, dim3(nseg), dim3(64), 0, st, (const lep5::SegPlan5*)plans, (uint8_t*)E.d_entries);
        HIPCHK(g, hipEventRecord(g->ev_fork, st));
        HIPCHK(g, hipStreamWaitEvent(g->stream2, g->ev_fork, 0));
        HIPCHK(g, hipStreamWaitEvent(g->stream3, g->ev_fork, 0));
        if (g->enc5_fold_apart) {   // measurement aid (LEP_ENC5_FOLD_APART=1): every kind of chain as a launch of its own, one after the other
            const int small0[4] = {0, 2, 14, 46}, big0[3] = {0, 12, 32};   // sign | threshold | edge counts;  DC | 7x7 counts
            for (int i = 0; i < 3; ++i)
                hipLaunchKernelGGL(lep_enc5_fold_small_kernel, dim3((unsigned)groups * (small0[i + 1] - small0[i])), dim3(64), 0, st, (const lep5::SegPlan5*)plans, (uint8_t*)E.d_entries,
                                   (uint32_t*)A.d_models, nseg, groups, small0[i]);
            for (int i = 0; i < 2; ++i)
                hipLaunchKernelGGL(lep_enc5_fold_big_kernel, dim3((unsigned)groups * (big0[i + 1] - big0[i])), dim3(64), 0, st, (const lep5::SegPlan5*)plans, (uint8_t*)E.d_entries, nseg, groups, big0[i]);
        } else {
        hipLaunchKernelGGL(lep_enc5_fold_small_kernel, dim3((unsigned)groups * kFold5SmallJobs), dim3(64), 0, g->stream2, (const lep5::SegPlan5*)plans, (uint8_t*)E.d_entries,
                           (uint32_t*)A.d_models, nseg, groups, 0);
        hipLaunchKernelGGL(lep_enc5_fold_big_kernel, dim3((unsigned)groups * kFold5BigJobs), dim3(64), 0, g->stream3, (const lep5::SegPlan5*)plans, (uint8_t*)E.d_entries, nseg, groups, 0);
        }
        hipLaunchKernelGGL(lep_enc5_fold_coef_kernel, dim3((unsigned)groups * 1260u), dim3(64), 0, st, (const lep5::SegPlan5*)plans, (uint8_t*)E.d_entries, nseg, groups);
        HIPCHK(g, hipEventRecord(g->ev_join, g->stream2));
        HIPCHK(g, hipEventRecord(g->ev_join3, g->stream3));
        HIPCHK(g, hipStreamWaitEvent(st, g->ev_join, 0));
        HIPCHK(g, hipStreamWaitEvent(st, g->ev_join3, 0));
        HIPCHK(g, hipEventRecord(g->ev_stage[3], st));
        // A launch that leaves the chip's lanes free takes the stitched writer: K chunks per segment (a power of two up to 64), as many as
        // keep the launch under enc5_wlanes chunk lanes.  Round 4 capped the lanes at 4096 (K = 2 at 256 images: two chains of 1.15 M bins
        // per segment, 118 ms of writer on 64 wavefronts); round 6 measured the cap away (profiles/r6o_*: 256 images 253 -> 148 ms per
        // launch with K = 32, 128 images 159 -> 107, 512 images -- which had kept the lane-per-segment writer -- 308 -> 260).  From 4097
        // segments on the lane-per-segment writer beside gather stays: there the walks fill the chip and hide it.
        int K = 0;
        if (g->enc5_wchunks >= 2 && nseg <= g->enc5_wmaxseg) { K = 1; while (K * 2 <= g->enc5_wchunks && (long)nseg * K * 2 <= g->enc5_wlanes) K *= 2; }
        if (K >= 2 && ensure(g, &E.d_wchunks, &E.wchunks_bytes, (size_t)nseg * K * sizeof(lep5::WChunk5) + 256, false)) { (void)hipGetLastError(); g->err.clear(); K = 0; }
        if (K >= 2) {
            for (int part = 0; part < nparts; ++part) walk(lep5::kGather, (uint8_t*)E.d_entries, (uint16_t*)E.d_binlist, part);
            HIPCHK(g, hipEventRecord(g->ev_stage[4], st));
            lep5::WChunk5* recs = (lep5::WChunk5*)E.d_wchunks;
            const int lane_groups = (int)(((long)nseg * K + 63) / 64);
#define LEP_WCHUNK(PHASE, GRID) hipLaunchKernelGGL((lep_enc5_wchunk_kernel<PHASE>), dim3(GRID), dim3(64), 0, st, (const lep5::SegPlan5*)plans, (const uint16_t*)E.d_binlist, d_seg, nseg, K, recs, d_streams, d_stream_len, d_status, g->d_bins)
            LEP_WCHUNK(0, lane_groups); LEP_WCHUNK(1, groups); LEP_WCHUNK(2, lane_groups); LEP_WCHUNK(3, groups);
#undef LEP_WCHUNK
            HIPCHK(g, hipEventRecord(g->ev_stage[5], st));
            g->nstage = 5;
        } else {
        // gather and write in parts (tile ranges of every segment): the writer is 128 wavefronts that take the same time whatever the
        // batch -- part k is written (second stream) while part k + 1 is gathered
        for (int part = 0; part < nparts; ++part) {
            walk(lep5::kGather, (uint8_t*)E.d_entries, (uint16_t*)E.d_binlist, part);
            HIPCHK(g, hipEventRecord(g->ev_part[part], st));
            HIPCHK(g, hipStreamWaitEvent(g->stream2, g->ev_part[part], 0));
            hipLaunchKernelGGL(lep_enc5_write_kernel, dim3(groups), dim3(64), 0, g->stream2, (const lep5::SegPlan5*)plans, (const uint16_t*)E.d_binlist, d_seg, nseg, d_streams,
                               d_stream_len, d_status, g->d_bins, (uint8_t*)E.d_entries, part, nparts);
        }
        HIPCHK(g, hipEventRecord(g->ev_stage[4], st));
        HIPCHK(g, hipEventRecord(g->ev_join, g->stream2));
        HIPCHK(g, hipStreamWaitEvent(st, g->ev_join, 0));
        HIPCHK(g, hipEventRecord(g->ev_stage[5], st));
        g->nstage = 5;
        }
    }
    HIPCHK(g, hipEventRecord(g->ev_enc5_done, st));
    HIPCHK(g, hipGetLastError());
    g->last_kernel = "lep_enc5 (count | emit | fold | gather | write)";
    return 0;
}

// Descriptor bytes to the device, ordered on `st`, WITHOUT waiting for what is queued on it: the bytes are copied into a pinned buffer
// of the codec's own (the caller's arrays may go away) and go up from there.  A launch function that waited for its stream here --
// hipMemcpyAsync from pageable memory, then hipStreamSynchronize -- waited for every kernel queued in front of it: in the batch
// decompressor that was the decode kernel of the chunk being launched (1 s), during which the host could have staged the next chunk
// (kernel trace of round 4: the decoder idle ~0.1 s per chunk).  A ring slot is reused sixteen uploads later; its event says when its
// copy has run.
static int upload(lep_gpu* g, void* dst, const void* src, size_t n, hipStream_t st) {
    if (!n) return 0;
    lep_gpu::Staging& sl = g->staging[g->staging_next];
    g->staging_next = (g->staging_next + 1) % 16;
    if (sl.used) { HIPCHK(g, hipEventSynchronize(sl.ev)); sl.used = false; }
    if (!sl.ev) HIPCHK(g, hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    if (sl.cap < n) {
        // (a buffer that has become too small is kept until the codec is destroyed: hipHostFree waits for the device, i.e. for the
        // very kernels this function exists not to wait for -- measured: 1.1 s per chunk.  8 MB take the descriptors of a 1024-image launch.)
        if (sl.host) g->staging_retired.push_back(sl.host);
        sl.host = nullptr; sl.cap = 0;
        const size_t want = std::max<size_t>(n + n / 4 + 4096, (size_t)8 << 20);
        HIPCHK(g, hipHostMalloc(&sl.host, want, hipHostMallocDefault));
        sl.cap = want;
    }
    memcpy(sl.host, src, n);
    HIPCHK(g, hipMemcpyAsync(dst, sl.host, n, hipMemcpyHostToDevice, st));
    HIPCHK(g, hipEventRecord(sl.ev, st));
    sl.used = true;
    return 0;
}

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
    for (int s = 0; s < nseg; ++s)
        if (segs[s].image < 0 || segs[s].image >= nimg) return LEP_ASSERTION_FAILURE;
    // Launch order.  Workgroups are observed to go to the 8 XCDs round-robin (block b -> XCD b % 8, MI355X_MICROARCH.md
    // "Workgroup dispatch, XCD placement": a speed matter only, nothing here depends on it), and the caller's order is
    // image-major: with 8 thread segments per image, segment k of EVERY image would land on XCD k -- and photographs are not
    // uniform top to bottom (sky above, detail below: up to 2x the bins per row), so one XCD would be the straggler of every
    // launch.  Segment r of image m is therefore queued for XCD (r + m) % 8 and the queues are interleaved; SegDev.slot
    // carries the caller's index for the results.
    const bool permute = nseg > 8;
    std::vector<int> order(nseg);
    if (permute) {
        std::vector<int> q[8];
        int rank = 0;
        for (int s = 0; s < nseg; ++s) {
            rank = (s > 0 && segs[s].image == segs[s - 1].image) ? rank + 1 : 0;
            q[(rank + segs[s].image) & 7].push_back(s);
        }
        size_t at[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < nseg;)
            for (int x = 0; x < 8 && i < nseg; ++x)
                if (at[x] < q[x].size()) order[i++] = q[x][at[x]++];
    } else {
        for (int s = 0; s < nseg; ++s) order[s] = s;
    }
    // Longest first.  A launch is over when its longest thread segment is, and the hardware hands workgroups out in launch order as
    // wave slots come free: with the long segments in front, the short ones of a mixed batch (1080p files beside 4K ones: a quarter of
    // the blocks) back-fill behind them -- in this launch when it has more segments than the chip holds wavefronts, and in the NEXT
    // chunk's launch, which the batch decompressor queues on a second stream (lep_batch.hip) -- instead of the other way round.  A
    // stable sort on the block count's size class (quarter octaves) keeps the XCD interleave above among near-equals, and a corpus whose
    // segments are within 1.5x of each other (the eight segments of equal-sized files differ by a block row) is launched exactly as before.
    if (nseg > 8) {
        std::vector<int> cls(nseg);
        int64_t lo = INT64_MAX, hi = 0;
        for (int s = 0; s < nseg; ++s) {
            const ImageDev& im = himg[segs[s].image];
            const int y1 = segs[s].is_last ? im.height[0] : std::min(segs[s].luma_y_end, im.height[0]);
            const int64_t w = std::max<int64_t>(1, (int64_t)std::max(0, y1 - segs[s].luma_y_start) * im.width[0]);
            lo = std::min(lo, w); hi = std::max(hi, w);
            int c = 0;
            for (int64_t v = w; v > 1; v >>= 1) c += 4;                       // 4 * floor(log2 w) ...
            const int64_t base = (int64_t)1 << (c / 4);
            c += w * 4 >= base * 7 ? 3 : (w * 2 >= base * 3 ? 2 : (w * 4 >= base * 5 ? 1 : 0));   // ... + the quarter inside the octave
            cls[s] = c;
        }
        if (hi * 2 > lo * 3) {
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cls[a] > cls[b]; });
        }
    }
    for (int i = 0; i < nseg; ++i) {
        const int s = order[i];
        hseg[i].image = segs[s].image; hseg[i].y0 = segs[s].luma_y_start; hseg[i].y1 = segs[s].luma_y_end;
        hseg[i].is_last = segs[s].is_last;
        hseg[i].stream_off = stream_offsets[s];
        uint64_t cap = stream_offsets[s + 1] - stream_offsets[s];
        hseg[i].stream_cap = (uint32_t)(cap > 0xffffffffu ? 0xffffffffu : cap);
        hseg[i].slot = (uint32_t)s;
        hns[i] = ns_total;
        ns_total += (uint64_t)himg[segs[s].image].ns_total;
    }
    HIPCHK(g, hipSetDevice(g->device));
    if (int rc = ensure(g, &g->arena[g->cur].d_models, &g->arena[g->cur].models_bytes, (size_t)nseg * kModelStride * 4)) return rc;
    if (int rc = ensure(g, &g->arena[g->cur].d_ns, &g->arena[g->cur].ns_bytes, (size_t)ns_total * sizeof(NSum) + 16)) return rc;
    const size_t o_img = 0, o_seg = o_img + ((nimg * sizeof(ImageDev) + 255) & ~(size_t)255),
                 o_ns = o_seg + ((nseg * sizeof(SegDev) + 255) & ~(size_t)255),
                 o_bins = o_ns + ((nseg * sizeof(uint64_t) + 255) & ~(size_t)255), total = o_bins + nseg * sizeof(uint32_t);
    if (int rc = ensure(g, &g->arena[g->cur].d_meta, &g->arena[g->cur].meta_bytes, total)) return rc;
    char* meta = (char*)g->arena[g->cur].d_meta;
    if (int rc = upload(g, meta + o_img, himg.data(), nimg * sizeof(ImageDev), st)) return rc;      // (the host vectors above go out of scope:
    if (int rc = upload(g, meta + o_seg, hseg.data(), nseg * sizeof(SegDev), st)) return rc;        //  upload() keeps a copy)
    if (int rc = upload(g, meta + o_ns, hns.data(), nseg * sizeof(uint64_t), st)) return rc;
    g->d_bins = (uint32_t*)(meta + o_bins);
    g->h_bins.assign(nseg, 0);
    g->nstage = 0;
    HIPCHK(g, hipEventRecord(g->ev0, st));
    if (DEC) {
#ifdef LEP_PROF
        { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_prof4)) == hipSuccess) (void)hipMemsetAsync(p, 0, sizeof(g_prof4), st); }
#endif
#define LEP_LAUNCH_DEC4(W)                                                                                                     \
    hipLaunchKernelGGL((lep_decode_v4_kernel<W>), dim3(nseg), dim3(64), 0, st, (const ImageDev*)(meta + o_img),                  \
                       (const SegDev*)(meta + o_seg), (uint32_t*)g->arena[g->cur].d_models, (NSum*)g->arena[g->cur].d_ns, (const uint64_t*)(meta + o_ns), \
                       d_streams, d_stream_len, d_status, g->d_bins)
        // more resident waves only pay once the batch can fill them (MI355X, 4K corpus: below ~4600 segments the 4-wave
        // build, which does not spill, is faster)
        int waves = g->dec_waves;
        // (a launch that is to share the chip with the launch before or behind it -- lep_gpu_expect_company, the batch decompressor's
        // chunks on two streams -- takes the 64-VGPR build whatever its size: four 128-VGPR wavefronts hold a SIMD's whole register file,
        // and the other launch's wavefronts could not move into the wave slots they leave empty)
        if (!waves) waves = (nseg > 4608 || (g->dec_company && nseg >= 64)) ? 8 : 4;
        if (waves >= 8) { g->last_kernel = "lep_decode_v4_kernel<8>"; LEP_LAUNCH_DEC4(8); }
        else { g->last_kernel = "lep_decode_v4_kernel<4>"; LEP_LAUNCH_DEC4(4); }
#undef LEP_LAUNCH_DEC4
    } else {
#define LEP_LAUNCH_ENC3(W)                                                                                                     \
    hipLaunchKernelGGL((lep_encode_v3_kernel<W>), dim3(nseg), dim3(64), 0, st, (const ImageDev*)(meta + o_img),                  \
                       (const SegDev*)(meta + o_seg), (uint32_t*)g->arena[g->cur].d_models, (NSum*)g->arena[g->cur].d_ns, (const uint64_t*)(meta + o_ns), \
                       d_streams, d_stream_len, d_status, g->d_bins)
        // like the decoder: a launch that cannot fill 8 wavefronts per SIMD takes the 4-wave build (128 VGPRs, no spills)
        int waves = g->enc_waves;
        if (!waves) waves = nseg > 4608 ? 8 : (nseg <= g->enc_pair_max ? 2 : 4);
        bool done = false;
        if (g->enc5_min > 0 && !g->enc_waves && nseg >= g->enc5_min) {   // (LEP_ENC_WAVES asks for one of the single-kernel forms)
            const int rc = launch_enc5(g, (const ImageDev*)(meta + o_img), (const SegDev*)(meta + o_seg), (const uint64_t*)(meta + o_ns), nseg, d_streams, d_stream_len, d_status, st);
            if (rc && rc != kEnc5NoMemory) return rc;
            done = rc == 0;
        }
        if (done) {}
        else if (waves == 2) {   // few segments: two wavefronts per segment (producer / bool coder), half the serial chain
            g->last_kernel = "lep_encode_v3x2_kernel";
            hipLaunchKernelGGL(lep_encode_v3x2_kernel, dim3(nseg), dim3(128), 0, st, (const ImageDev*)(meta + o_img),
                               (const SegDev*)(meta + o_seg), (uint32_t*)g->arena[g->cur].d_models, (NSum*)g->arena[g->cur].d_ns, (const uint64_t*)(meta + o_ns),
                               d_streams, d_stream_len, d_status, g->d_bins);
        }
        else if (waves >= 8) { g->last_kernel = "lep_encode_v3_kernel<8>"; LEP_LAUNCH_ENC3(8); }
        else { g->last_kernel = "lep_encode_v3_kernel<4>"; LEP_LAUNCH_ENC3(4); }
#undef LEP_LAUNCH_ENC3
    }
    HIPCHK(g, hipGetLastError());
    HIPCHK(g, hipEventRecord(g->ev1, st));
    g->timed = true;
    return 0;
}

// The HIP runtime multiplexes a process's streams onto a few hardware queues -- four per device unless GPU_MAX_HW_QUEUES says
// otherwise -- and two streams that share one run their work one after the other.  The batch pipeline keeps seven streams busy (copies
// up, copies down, two coder launches, the scan decoder at raised priority, the split-phase encoder's two side streams): in the kernel
// and copy trace of round 4 (profiles/r05b_*, r05c_*) the encoder's gather and write parts, which overlap in a process with three
// streams, alternated strictly, and a chunk's upload waited for the previous chunk's decode kernel.  Eight queues unless the
// environment already chose; read by the runtime when it initialises, i.e. before this library's first HIP call.
__attribute__((constructor)) static void lep_runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

extern "C" {

// Process exit.  The HIP runtime registers its own teardown with atexit() lazily, at the first HIP call -- i.e. AFTER the
// static objects of a host program were constructed, so it runs BEFORE their destructors.  A host that keeps its coder in a
// global (the reference does: g_encoder / g_decoder, jpgcoder.cc:428, 478) would call lep_gpu_destroy on a runtime that is
// already gone and hang in hipStreamSynchronize (seen on MI355X with oracle/_ref/lepton-mi355x: work done in 0.1 s, process
// never ended).  So the library registers its own atexit handler after the first successful create -- later than HIP's,
// hence run earlier -- which releases every live object while the runtime still works; lep_gpu_destroy on such an object
// afterwards only frees the host struct.
static std::mutex g_live_mu;
static std::vector<lep_gpu*> g_live;
static bool g_exit_hook = false;
static void release_device_side(lep_gpu* g);
static void release_all_at_exit() {
    std::lock_guard<std::mutex> lk(g_live_mu);
    for (lep_gpu* g : g_live) release_device_side(g);
    g_live.clear();
}

int lep_gpu_create(int device, lep_gpu** out) {
    lep_gpu* g = new lep_gpu;
    g->device = device;
    if (const char* e = getenv("LEP_DEC_WAVES")) g->dec_waves = atoi(e) == 4 ? 4 : (atoi(e) == 8 ? 8 : 0);
    if (const char* e = getenv("LEP_ENC_WAVES")) g->enc_waves = atoi(e) == 4 ? 4 : (atoi(e) == 8 ? 8 : (atoi(e) == 2 ? 2 : 0));
    if (const char* e = getenv("LEP_ENC_PAIR_MAX")) g->enc_pair_max = atoi(e);
    if (const char* e = getenv("LEP_ENC5_MIN")) g->enc5_min = atoi(e);
    if (const char* e = getenv("LEP_HUFFPROG_PIPELINE")) g->huffprog_pipeline = atoi(e);
    if (const char* e = getenv("LEP_HUFFPROG_PIPELINE_MAX")) g->huffprog_pipeline_max = atoi(e);
    if (const char* e = getenv("LEP_HUFFDEC_SIMT_BITS")) g->simt_sub_bits = std::max(0, atoi(e));
    if (const char* e = getenv("LEP_HUFFENC_SIMT")) g->huffenc_simt = atoi(e) != 0;
    if (const char* e = getenv("LEP_HUFFPROG_SIMT")) g->huffprog_simt = atoi(e) != 0;
    if (const char* e = getenv("LEP_HUFFPROGDEC_WIN")) g->huffprogdec_win = atoi(e) != 0;
    if (const char* e = getenv("LEP_HUFFPROG_SPLIT")) g->huffprog_split = atoi(e) != 0;
    if (const char* e = getenv("LEP_ENC5_WAVES")) g->enc5_waves = atoi(e) == 1 ? 1 : 2;
    if (const char* e = getenv("LEP_ENC5_FOLD_APART")) g->enc5_fold_apart = atoi(e);
    if (const char* e = getenv("LEP_ENC5_GATHER_WGS")) g->enc5_gather_wgs = atoi(e);
    if (const char* e = getenv("LEP_ENC5_PARTS")) g->enc5_parts = atoi(e);
    if (const char* e = getenv("LEP_ENC5_WCHUNKS")) g->enc5_wchunks = atoi(e);
    if (const char* e = getenv("LEP_ENC5_WLANES")) g->enc5_wlanes = atol(e);
    if (const char* e = getenv("LEP_ENC5_WMAXSEG")) g->enc5_wmaxseg = atoi(e);
    if (const char* e = getenv("LEP_ENC5_SCRATCH_MAX")) g->enc5_scratch_max = (size_t)strtoull(e, nullptr, 10);
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= device) { delete g; return LEP_GPU_ERROR; }
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&g->stream) != hipSuccess ||
        hipEventCreate(&g->ev0) != hipSuccess || hipEventCreate(&g->ev1) != hipSuccess) { delete g; return LEP_GPU_ERROR; }
    vmm_init(g);
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live.push_back(g);
        if (!g_exit_hook) { g_exit_hook = true; atexit(release_all_at_exit); }
    }
    *out = g;
    return 0;
}

static void release_device_side(lep_gpu* g) {
    if (g->released) return;
    g->released = true;
    (void)hipSetDevice(g->device);
    (void)hipStreamSynchronize(g->stream);
    for (void** p : {&g->enc5.d_plans, &g->enc5.d_entries, &g->enc5.d_binlist, &g->enc5.d_wchunks}) dev_release(g, p, nullptr);
    if (g->ev_enc5_done) (void)hipEventDestroy(g->ev_enc5_done);
    for (auto& e : g->ev_part) if (e) (void)hipEventDestroy(e);
    for (auto& e : g->ev_stage) if (e) (void)hipEventDestroy(e);
    if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
    if (g->ev_join) (void)hipEventDestroy(g->ev_join);
    if (g->ev_join3) (void)hipEventDestroy(g->ev_join3);
    if (g->stream2) (void)hipStreamDestroy(g->stream2);
    if (g->stream3) (void)hipStreamDestroy(g->stream3);
    for (void** p : {&g->arena[0].d_models, &g->arena[0].d_ns, &g->arena[0].d_meta, &g->arena[1].d_models, &g->arena[1].d_ns, &g->arena[1].d_meta, &g->d_blocks, &g->d_streams, &g->d_lens, &g->d_huff[0], &g->d_huff[1],
                     &g->d_huffprog[0], &g->d_huffprog[1], &g->d_huffprogsimt[0], &g->d_huffprogsimt[1], &g->d_huffseq[0], &g->d_huffseq[1], &g->d_huffprogdec, &g->d_huffdec, &g->d_huffpar, &g->d_huffenc[0], &g->d_huffenc[1], &g->d_scan, &g->d_scanlen})
        dev_release(g, p, nullptr);
    vmm_destroy(g);
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    for (void* p : {g->h_huffprog[0], g->h_huffprog[1]}) if (p) (void)hipHostFree(p);
    for (lep_gpu::Staging& sl : g->staging) { if (sl.host) (void)hipHostFree(sl.host); if (sl.ev) (void)hipEventDestroy(sl.ev); sl = lep_gpu::Staging(); }
    for (void* p : g->staging_retired) (void)hipHostFree(p);
    g->staging_retired.clear();
}

void lep_gpu_destroy(lep_gpu* g) {
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        for (size_t i = 0; i < g_live.size(); ++i) if (g_live[i] == g) { g_live.erase(g_live.begin() + i); break; }
        release_device_side(g);   // a no-op when the exit handler got there first
    }
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

static_assert(sizeof(lep_huff_image) == sizeof(lephuff::HuffImage) && sizeof(lep_huff_segment) == sizeof(lephuff::HuffSegment) &&
              sizeof(lep_huff_end) == sizeof(lephuff::HuffEnd) && sizeof(lep_huff_end) == 16, "C ABI mirrors");

int lep_gpu_huffman_encode_device(lep_gpu* g, const lep_huff_image* images, int nimg, const lep_huff_segment* segs, int nseg,
                                  uint8_t* d_out, uint32_t* d_out_len, lep_huff_end* d_ends, void* hip_stream) {
    if (!g) return LEP_GPU_ERROR;
    if (nseg <= 0) return 0;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : g->stream;
    HIPCHK(g, hipSetDevice(g->device));
    const size_t o_seg = (nimg * sizeof(lep_huff_image) + 255) & ~(size_t)255, total = o_seg + nseg * sizeof(lep_huff_segment);
    const int set = g->cur & 1;                            // (the arena set the caller chose: two launches on two streams never share descriptors or scratch)
    if (int rc = ensure(g, &g->d_huff[set], &g->huff_bytes[set], total)) return rc;
    // which segments the lane-per-unit kernels take (lep_huff_simt.h); the wavefront-per-segment kernel keeps the others
    static_assert(sizeof(lep_huff_image) == sizeof(lephuff::HuffImage) && sizeof(lep_huff_segment) == sizeof(lephuff::HuffSegment), "C ABI mirrors");
    std::vector<lep_huff_segment> sv(segs, segs + nseg);
    std::vector<lephuff::SimtEncSeg> es;
    std::vector<lephuff::SimtEncWave> waves;
    size_t nunits = 0, scratch_bytes = 0;
    for (int i = 0; i < nseg; ++i) {   // (a truncated file's segments are the lane-per-unit kernels' or nobody's: the wavefront kernel knows no cut)
        sv[(size_t)i].pad = 0;
        if (sv[(size_t)i].image < 0 || sv[(size_t)i].image >= nimg) continue;
        const lep_huff_image& im = images[sv[(size_t)i].image];
        if (im.trunc_bc[0] | im.trunc_bc[1] | im.trunc_bc[2] | im.trunc_bc[3]) sv[(size_t)i].pad = lephuff::kHuffSegRefuse;
    }
    if (g->huffenc_simt)
        for (int i = 0; i < nseg; ++i) {
            if (sv[(size_t)i].image < 0 || sv[(size_t)i].image >= nimg) continue;
            const lephuff::HuffImage& im = reinterpret_cast<const lephuff::HuffImage&>(images[sv[(size_t)i].image]);
            if (!lephuff::simt_enc_takes(im, reinterpret_cast<const lephuff::HuffSegment&>(sv[(size_t)i]))) continue;
            lephuff::SimtEncSeg e;
            memset(&e, 0, sizeof e);
            if ((int64_t)sv[(size_t)i].mcu_row1 * im.mcuh > 0x7fffffff) continue;
            lephuff::SimtUnitMap map;
            map.set(sv[(size_t)i].mcu_row0 * im.mcuh, sv[(size_t)i].mcu_row1 * im.mcuh, im.rsti);
            e.seg = (uint32_t)i; e.first_unit = (uint32_t)nunits; e.nunits = map.count();
            e.buf_off = scratch_bytes; e.buf_bytes = (uint32_t)std::min<size_t>(((size_t)sv[(size_t)i].out_cap + 64 + 15) & ~(size_t)15, 0xfffffff0u);
            e.map_bytes = im.rsti > 0 ? ((e.buf_bytes >> 3) + 15u) & ~15u : 0u;   // restart intervals: which bytes of the bit buffer are markers
            if (nunits + e.nunits > 0x7fffffffu) continue;
            for (uint32_t f = 0; f < e.nunits; f += 64) waves.push_back(lephuff::SimtEncWave{(uint32_t)es.size(), f});
            nunits += e.nunits; scratch_bytes += (size_t)e.buf_bytes + e.map_bytes;
            sv[(size_t)i].pad = lephuff::kHuffSegSimt;
            es.push_back(e);
        }
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_es = 0, o_wv = up(es.size() * sizeof(lephuff::SimtEncSeg)), o_ub = o_wv + up(waves.size() * sizeof(lephuff::SimtEncWave)),
                 o_sc = o_ub + up(nunits * 8), simt_total = o_sc + up(scratch_bytes);   // (units: bit counts / positions, and the plain prefix sum of scans with restart intervals)
    if (!es.empty()) { if (int rc = ensure(g, &g->d_huffenc[set], &g->huffenc_bytes[set], simt_total)) return rc; }
    if (int rc = upload(g, g->d_huff[set], images, nimg * sizeof(lep_huff_image), st)) return rc;           // (the caller's arrays and ours may go away)
    if (int rc = upload(g, (char*)g->d_huff[set] + o_seg, sv.data(), nseg * sizeof(lep_huff_segment), st)) return rc;
    char* eb = (char*)g->d_huffenc[set];
    if (!es.empty()) {
        if (int rc = upload(g, eb + o_es, es.data(), es.size() * sizeof(lephuff::SimtEncSeg), st)) return rc;
        if (int rc = upload(g, eb + o_wv, waves.data(), waves.size() * sizeof(lephuff::SimtEncWave), st)) return rc;
        hipLaunchKernelGGL(lep_zero_kernel, dim3(8192), dim3(256), 0, st, (uint4*)(eb + o_sc), scratch_bytes / 16);   // (buf_bytes are multiples of 16)
    }
    const lephuff::HuffImage* di = (const lephuff::HuffImage*)g->d_huff[set];
    const lephuff::HuffSegment* ds = (const lephuff::HuffSegment*)((char*)g->d_huff[set] + o_seg);
    HIPCHK(g, hipEventRecord(g->ev0, st));
    if (!es.empty()) {
        lephuff::SimtEncSeg* des = (lephuff::SimtEncSeg*)(eb + o_es);
        const lephuff::SimtEncWave* dwv = (const lephuff::SimtEncWave*)(eb + o_wv);
        uint32_t* dub = (uint32_t*)(eb + o_ub);
        uint8_t* dsc = (uint8_t*)(eb + o_sc);
        hipLaunchKernelGGL((lep_huffman_simt_encode_units_kernel<false>), dim3((unsigned)waves.size()), dim3(64), 0, st, di, ds, des, dwv, dub, dsc);
        hipLaunchKernelGGL(lep_huffman_simt_encode_place_kernel, dim3((unsigned)es.size()), dim3(64), 0, st, di, ds, des, dub, dub + nunits);
        hipLaunchKernelGGL((lep_huffman_simt_encode_units_kernel<true>), dim3((unsigned)waves.size()), dim3(64), 0, st, di, ds, des, dwv, dub, dsc);
        hipLaunchKernelGGL(lep_huffman_simt_encode_stuff_kernel, dim3((unsigned)es.size()), dim3(64), 0, st, di, ds, (const lephuff::SimtEncSeg*)des, dsc, d_out, d_out_len,
                           (lephuff::HuffEnd*)d_ends);
    }
    if ((int)es.size() < nseg)
        hipLaunchKernelGGL(lep_huffman_encode_kernel, dim3(nseg), dim3(64), 0, st, di, ds, d_out, d_out_len, (lephuff::HuffEnd*)d_ends);
    HIPCHK(g, hipGetLastError());
    HIPCHK(g, hipEventRecord(g->ev1, st));
    g->timed = true;
    g->last_kernel = es.empty() ? "lep_huffman_encode_kernel" : "lep_huffman_simt_encode_{units,place,stuff}_kernel";
    return 0;
}

static_assert(sizeof(lep_huffprogdec_scan) == sizeof(lephuff::ProgDecScan), "C ABI mirrors");

int lep_gpu_huffman_progressive_decode_device(lep_gpu* g, const lep_huffprogdec_scan* scans, int nscan, lep_huffdec_row* d_rows, void* hip_stream) {
    if (!g) return LEP_GPU_ERROR;
    if (nscan <= 0) return 0;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : g->stream;
    HIPCHK(g, hipSetDevice(g->device));
    // scans of SEQUENTIAL frames coded in several scans (lep_huffprogdec.h sequential_scan_image): no scan depends on another, each is an image
    // of its own to the sequential kernels -- one lane per subsequence, or the single-wave kernel where there are restart intervals
    std::vector<lep_huffprogdec_scan> progressive_only;
    {
        std::vector<lep_huffdec_image> many, one;
        const bool simt = !(getenv("LEP_HUFFDEC_SIMT") && atoi(getenv("LEP_HUFFDEC_SIMT")) == 0);
        for (int i = 0; i < nscan; ++i) {
            const lephuff::ProgDecScan& sc = reinterpret_cast<const lephuff::ProgDecScan&>(scans[i]);
            if (!lephuff::progdec_is_sequential(sc)) continue;
            if (sc.cmpc < 1 || sc.cmpc > 4) return LEP_ASSERTION_FAILURE;
            const lephuff::HuffDecImage im = lephuff::sequential_scan_image(sc);
            lep_huffdec_image out;
            static_assert(sizeof out == sizeof im, "C ABI mirrors");
            memcpy(&out, &im, sizeof out);
            ((simt && lephuff::sequential_scan_for_lanes(im)) ? many : one).push_back(out);
        }
        if (!many.empty() || !one.empty()) {
            if (!many.empty()) { if (int rc = lep_gpu_huffman_decode_simt_device(g, many.data(), (int)many.size(), d_rows, st)) return rc; }
            if (!one.empty()) { if (int rc = lep_gpu_huffman_decode_device(g, one.data(), (int)one.size(), d_rows, st)) return rc; }
            for (int i = 0; i < nscan; ++i) if (!lephuff::progdec_is_sequential(reinterpret_cast<const lephuff::ProgDecScan&>(scans[i]))) progressive_only.push_back(scans[i]);
            if (progressive_only.empty()) return 0;
            scans = progressive_only.data(); nscan = (int)progressive_only.size();
        }
    }
    // scans ordered by dependency level (stable): one launch per level, stream order is the dependency
    int maxlevel = 0;
    for (int i = 0; i < nscan; ++i) { if (scans[i].level < 0 || scans[i].level > 63) return LEP_ASSERTION_FAILURE; maxlevel = std::max(maxlevel, (int)scans[i].level); }
    std::vector<lep_huffprogdec_scan> sorted;
    sorted.reserve((size_t)nscan);
    std::vector<int> order;
    order.reserve((size_t)nscan);
    // (LEP_HUFFPROG_SPLIT=1, a measurement aid: inside a level the scans of one KIND -- DC / AC, first stage / refinement, component, band --
    // stand together and get a launch of their own, so that a kernel trace shows what each kind of scan takes)
    auto kind = [](const lep_huffprogdec_scan& s) { return (s.to == 0 ? 0 : 1) * 100000 + (s.sah ? 1 : 0) * 10000 + (s.cmpc > 1 ? 9 : s.cmp[0]) * 1000 + s.from * 10 + (s.to > 9 ? 9 : s.to); };
    std::vector<int> cut;   // launch boundaries inside `sorted`
    for (int lv = 0; lv <= maxlevel; ++lv) {
        std::vector<int> idx;
        for (int i = 0; i < nscan; ++i) if (scans[i].level == lv) idx.push_back(i);
        if (g->huffprog_split) std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return kind(scans[a]) < kind(scans[b]); });
        for (size_t q = 0; q < idx.size(); ++q) {
            if (q == 0 || (g->huffprog_split && kind(scans[idx[q]]) != kind(scans[idx[q - 1]]))) cut.push_back((int)sorted.size());
            sorted.push_back(scans[idx[q]]); order.push_back(idx[q]);
        }
    }
    cut.push_back((int)sorted.size());
    // Small launches wait for the chain of a file's dependent scans, not for throughput: all levels go out as ONE launch in which
    // a scan follows the scans in front of it MCU row by MCU row.  (Beyond what is resident at once the levels are launched one
    // after the other as before: the chip is full either way.)
    std::vector<lephuff::ProgDeps> deps;
    bool pipelined = g->huffprog_pipeline && maxlevel > 0 && nscan <= g->huffprog_pipeline_max;
    if (pipelined) {
        deps.resize((size_t)nscan);
        pipelined = lephuff::prog_scan_deps(reinterpret_cast<const lephuff::ProgDecScan*>(sorted.data()), order.data(), nscan, deps.data());
    }
    const size_t o_deps = ((size_t)nscan * sizeof(lep_huffprogdec_scan) + 255) & ~(size_t)255, o_prog = o_deps + (((size_t)nscan * sizeof(lephuff::ProgDeps) + 255) & ~(size_t)255),
                 total = o_prog + (size_t)nscan * 4 + 4;   // (+ the ticket counter behind the progress words)
    if (int rc = ensure(g, &g->d_huffprogdec, &g->huffprogdec_bytes, total)) return rc;
    // which scans the window of speculative codes decodes (lep_huffprogdec_win.h); lep_huffprogdec.h's uniform vector code keeps the others
    bool any_win = false;
    for (lep_huffprogdec_scan& sc : sorted) {
        const bool w = g->huffprogdec_win && lephuff::prog_win_takes(reinterpret_cast<const lephuff::ProgDecScan&>(sc));
        sc.pad = w ? lephuff::kProgDecWin : 0;
        any_win = any_win || w;
    }
    HIPCHK(g, hipMemcpyAsync(g->d_huffprogdec, sorted.data(), (size_t)nscan * sizeof(lep_huffprogdec_scan), hipMemcpyHostToDevice, st));
    if (pipelined) {
        HIPCHK(g, hipMemcpyAsync((char*)g->d_huffprogdec + o_deps, deps.data(), (size_t)nscan * sizeof(lephuff::ProgDeps), hipMemcpyHostToDevice, st));
        HIPCHK(g, hipMemsetAsync((char*)g->d_huffprogdec + o_prog, 0, (size_t)nscan * 4 + 4, st));
    }
    HIPCHK(g, hipStreamSynchronize(st));
    HIPCHK(g, hipEventRecord(g->ev0, st));
    if (pipelined) {
        hipLaunchKernelGGL(any_win ? lep_huffprogdec_win_pipelined_kernel : lep_huffman_progressive_pipelined_kernel, dim3(nscan), dim3(64), 0, st,
                           (const lephuff::ProgDecScan*)g->d_huffprogdec, (lephuff::HuffDecRow*)d_rows,
                           (const lephuff::ProgDeps*)((char*)g->d_huffprogdec + o_deps), (uint32_t*)((char*)g->d_huffprogdec + o_prog), (uint32_t*)((char*)g->d_huffprogdec + o_prog) + nscan);
        HIPCHK(g, hipGetLastError());
        HIPCHK(g, hipEventRecord(g->ev1, st));
        g->timed = true;
        g->last_kernel = any_win ? "lep_huffprogdec_win_pipelined_kernel" : "lep_huffman_progressive_pipelined_kernel";
        return 0;
    }
    for (size_t c = 0; c + 1 < cut.size(); ++c) {   // one launch per level (LEP_HUFFPROG_SPLIT: per level and kind of scan)
        const int n = cut[c + 1] - cut[c];
        if (n <= 0) continue;
        hipLaunchKernelGGL(any_win ? lep_huffprogdec_win_kernel : lep_huffman_progressive_decode_kernel, dim3(n), dim3(64), 0, st,
                           (const lephuff::ProgDecScan*)g->d_huffprogdec + cut[c], (lephuff::HuffDecRow*)d_rows);
    }
    HIPCHK(g, hipGetLastError());
    HIPCHK(g, hipEventRecord(g->ev1, st));
    g->timed = true;
    g->last_kernel = any_win ? "lep_huffprogdec_win_kernel" : "lep_huffman_progressive_decode_kernel";
    return 0;
}

static_assert(sizeof(lep_huffprog_image) == sizeof(lephuff::ProgImage) && sizeof(lep_huffprog_scan) == sizeof(lephuff::ProgScan), "C ABI mirrors");

int lep_gpu_huffman_progressive_encode_device(lep_gpu* g, const lep_huffprog_image* images, int nimg, const lep_huffprog_scan* scans, int nscan,
                                              uint8_t* d_out, uint32_t* d_corr, uint32_t* d_out_len, void* hip_stream) {
    if (!g) return LEP_GPU_ERROR;
    if (nscan <= 0) return 0;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : g->stream;
    HIPCHK(g, hipSetDevice(g->device));
    const size_t o_scan = (nimg * sizeof(lep_huffprog_image) + 255) & ~(size_t)255, total = o_scan + nscan * sizeof(lep_huffprog_scan);
    const int turn = (g->huffprog_turn ^= 1);
    if (int rc = ensure(g, &g->d_huffprog[turn], &g->huffprog_bytes[turn], total)) return rc;
    char* const d_desc = (char*)g->d_huffprog[turn];
    // descriptors through a pinned copy of this set (its previous user, two launches ago, has been waited for by then): the
    // caller's arrays may go away, and nothing here waits for the kernels already queued on the stream
    if (g->h_huffprog_bytes[turn] < total) {
        if (g->h_huffprog[turn]) HIPCHK(g, hipHostFree(g->h_huffprog[turn]));
        g->h_huffprog[turn] = nullptr; g->h_huffprog_bytes[turn] = 0;
        HIPCHK(g, hipHostMalloc(&g->h_huffprog[turn], total + total / 8, hipHostMallocDefault));
        g->h_huffprog_bytes[turn] = total + total / 8;
    }
    char* const h_desc = (char*)g->h_huffprog[turn];
    memcpy(h_desc, images, nimg * sizeof(lep_huffprog_image));
    memcpy(h_desc + o_scan, scans, nscan * sizeof(lep_huffprog_scan));
    // which scans the lane-per-unit kernels take (lep_huffprog_simt.h); the wavefront-per-scan kernel keeps the others
    lephuff::ProgScan* hs = reinterpret_cast<lephuff::ProgScan*>(h_desc + o_scan);
    std::vector<lephuff::ProgSimtScan> ps;
    std::vector<lephuff::ProgSimtWave> waves;
    std::vector<lephuff::ProgSimtRegion> regions;
    size_t nunits = 0, scratch_bytes = 0;
    std::vector<uint32_t> file_bound((size_t)nscan);
    for (int i = 0; i < nscan; ++i) { file_bound[(size_t)i] = hs[i].pad; hs[i].pad = 0; }   // (the caller's field; on the device it says which kernel owns the scan)
    // scans of SEQUENTIAL frames coded in several scans (lep_huffprog.h sequential_scan_segment): the sequential scan encoders write them,
    // each scan an image with one segment; their byte counts are moved to the scans' places behind that launch
    std::vector<lep_huff_image> seq_img;
    std::vector<lep_huff_segment> seq_seg;
    std::vector<uint32_t> seq_which, seq_cap;
    for (int i = 0; i < nscan; ++i) {
        if (!lephuff::prog_is_sequential(hs[i])) continue;
        if (hs[i].image < 0 || hs[i].image >= nimg || hs[i].cmpc < 1 || hs[i].cmpc > 4) return LEP_ASSERTION_FAILURE;
        lephuff::HuffImage hi;
        lephuff::HuffSegment sg;
        lephuff::sequential_scan_segment(reinterpret_cast<const lephuff::ProgImage&>(images[hs[i].image]), hs[i], (int32_t)seq_img.size(), &hi, &sg);
        lep_huff_image ci; lep_huff_segment cs;
        memcpy(&ci, &hi, sizeof ci); memcpy(&cs, &sg, sizeof cs);
        seq_img.push_back(ci); seq_seg.push_back(cs);
        seq_which.push_back((uint32_t)i); seq_cap.push_back(hs[i].out_cap);
        hs[i].pad = lephuff::kProgScanSeq;
    }
    if (!seq_seg.empty()) {
        const size_t n = seq_seg.size(), o_cap = n * 4, o_len = o_cap + n * 4, o_end = (o_len + n * 4 + 15) & ~(size_t)15, bytes = o_end + n * sizeof(lep_huff_end);
        if (int rc = ensure(g, &g->d_huffseq[turn], &g->huffseq_bytes[turn], bytes)) return rc;
        char* b = (char*)g->d_huffseq[turn];
        if (int rc = upload(g, b, seq_which.data(), n * 4, st)) return rc;
        if (int rc = upload(g, b + o_cap, seq_cap.data(), n * 4, st)) return rc;
        if (int rc = lep_gpu_huffman_encode_device(g, seq_img.data(), (int)n, seq_seg.data(), (int)n, d_out, (uint32_t*)(b + o_len), (lep_huff_end*)(b + o_end), st)) return rc;
        hipLaunchKernelGGL(lep_huffprog_seq_lens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint32_t*)b, (const uint32_t*)(b + o_cap), (const uint32_t*)(b + o_len),
                           (const lephuff::HuffEnd*)(b + o_end), (int)n, d_out_len);
        if ((int)n == nscan) { HIPCHK(g, hipGetLastError()); return 0; }
    }
    if (g->huffprog_simt) {
        // scans grouped by image (a region of bit buffers per image): the caller lists them file by file, but nothing here relies on it
        std::vector<int> order((size_t)nscan);
        for (int i = 0; i < nscan; ++i) order[(size_t)i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return hs[a].image < hs[b].image; });
        for (size_t a = 0; a < order.size();) {
            size_t b = a;
            while (b < order.size() && hs[order[b]].image == hs[order[a]].image) ++b;
            const int im = hs[order[a]].image;
            lephuff::ProgSimtRegion r{(uint32_t)ps.size(), 0u, scratch_bytes, 0};
            uint64_t sum_cap = 0, bound = 0;
            for (size_t k = a; k < b && im >= 0 && im < nimg; ++k) {
                const int i = order[k];
                uint32_t nb = 0, nu = 0;
                if (!lephuff::prog_simt_takes(reinterpret_cast<const lephuff::ProgImage&>(images[im]), hs[i], &nb, &nu)) continue;
                if (nunits + nu > 0x7fffffffu) continue;
                lephuff::ProgSimtScan e;
                memset(&e, 0, sizeof e);
                e.scan = (uint32_t)i; e.first_unit = (uint32_t)nunits; e.nunits = nu; e.nblocks = nb;
                for (uint32_t f = 0; f < nu; f += 64) waves.push_back(lephuff::ProgSimtWave{(uint32_t)ps.size(), f});
                nunits += nu;
                sum_cap += (uint64_t)hs[i].out_cap + 96; bound = std::max<uint64_t>(bound, file_bound[(size_t)i]);
                hs[i].pad = lephuff::kProgScanSimt;
                ps.push_back(e);
                ++r.nps;
            }
            if (r.nps) {
                // the file's scans together are shorter than the file (lep_huffprog_scan.file_bound, where the caller said); a region
                // that turns out too small leaves scans without a buffer, and the host re-coder takes the file
                r.bytes = ((bound ? std::min<uint64_t>(sum_cap, bound + 96ull * r.nps + 4096) : sum_cap) + 15) & ~(uint64_t)15;
                scratch_bytes += r.bytes;
                regions.push_back(r);
            }
            a = b;
        }
    }
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_ps = 0, o_wv = up(ps.size() * sizeof(lephuff::ProgSimtScan)), o_rg = o_wv + up(waves.size() * sizeof(lephuff::ProgSimtWave)),
                 o_un = o_rg + up(regions.size() * sizeof(lephuff::ProgSimtRegion)), o_sc = o_un + up(nunits * 4 * lephuff::kProgSimtUnitWords),
                 simt_total = o_sc + up(scratch_bytes);
    if (!ps.empty()) { if (int rc = ensure(g, &g->d_huffprogsimt[turn], &g->huffprogsimt_bytes[turn], simt_total)) return rc; }
    HIPCHK(g, hipMemcpyAsync(d_desc, h_desc, total, hipMemcpyHostToDevice, st));
    char* eb = (char*)g->d_huffprogsimt[turn];
    if (!ps.empty()) {
        if (int rc = upload(g, eb + o_ps, ps.data(), ps.size() * sizeof(lephuff::ProgSimtScan), st)) return rc;
        if (int rc = upload(g, eb + o_wv, waves.data(), waves.size() * sizeof(lephuff::ProgSimtWave), st)) return rc;
        if (int rc = upload(g, eb + o_rg, regions.data(), regions.size() * sizeof(lephuff::ProgSimtRegion), st)) return rc;
    }
    const lephuff::ProgImage* di = (const lephuff::ProgImage*)d_desc;
    const lephuff::ProgScan* ds = (const lephuff::ProgScan*)(d_desc + o_scan);
    HIPCHK(g, hipEventRecord(g->ev0, st));
    if (!ps.empty()) {
        lephuff::ProgSimtScan* dps = (lephuff::ProgSimtScan*)(eb + o_ps);
        const lephuff::ProgSimtWave* dwv = (const lephuff::ProgSimtWave*)(eb + o_wv);
        uint32_t* dun = (uint32_t*)(eb + o_un);
        uint8_t* dsc = (uint8_t*)(eb + o_sc);
        uint32_t longest = 0;
        for (const lephuff::ProgSimtRegion& r : regions) longest = (uint32_t)std::max<uint64_t>(longest, r.bytes);
        const uint32_t chunk16 = 4096;   // 64 KB of a bit buffer per workgroup of the clearing kernel
        hipLaunchKernelGGL((lep_huffprog_simt_units_kernel<false>), dim3((unsigned)waves.size()), dim3(64), 0, st, di, ds, (const lephuff::ProgSimtScan*)dps, dwv, dun, nunits, dsc);
        hipLaunchKernelGGL(lep_huffprog_simt_place_kernel, dim3((unsigned)ps.size()), dim3(64), 0, st, ds, dps, dun, nunits);
        hipLaunchKernelGGL(lep_huffprog_simt_assign_kernel, dim3((unsigned)(regions.size() + 63) / 64), dim3(64), 0, st, (const lephuff::ProgSimtRegion*)(eb + o_rg), (int)regions.size(), dps);
        hipLaunchKernelGGL(lep_huffprog_simt_zero_kernel, dim3((unsigned)ps.size(), (longest / 16 + chunk16 - 1) / chunk16 + 1), dim3(256), 0, st, (const lephuff::ProgSimtScan*)dps, dsc, chunk16);
        hipLaunchKernelGGL((lep_huffprog_simt_units_kernel<true>), dim3((unsigned)waves.size()), dim3(64), 0, st, di, ds, (const lephuff::ProgSimtScan*)dps, dwv, dun, nunits, dsc);
        hipLaunchKernelGGL(lep_huffprog_simt_stuff_kernel, dim3((unsigned)ps.size()), dim3(64), 0, st, di, ds, (const lephuff::ProgSimtScan*)dps, dsc, d_out, d_out_len);
    }
    if (ps.size() + seq_seg.size() < (size_t)nscan)
        hipLaunchKernelGGL(lep_huffman_progressive_encode_kernel, dim3(nscan), dim3(64), 0, st, di, ds, d_out, d_corr, d_out_len);
    HIPCHK(g, hipGetLastError());
    HIPCHK(g, hipEventRecord(g->ev1, st));
    g->timed = true;
    g->last_kernel = ps.empty() ? "lep_huffman_progressive_encode_kernel" : "lep_huffprog_simt_{units,place,stuff}_kernel";
    return 0;
}

static_assert(sizeof(lep_huffdec_image) == sizeof(lephuff::HuffDecImage) && sizeof(lep_huffdec_row) == sizeof(lephuff::HuffDecRow), "C ABI mirrors");

int lep_gpu_huffman_decode_device(lep_gpu* g, const lep_huffdec_image* images, int nimg, lep_huffdec_row* d_rows, void* hip_stream) {
    if (!g) return LEP_GPU_ERROR;
    if (nimg <= 0) return 0;
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : g->stream;
    HIPCHK(g, hipSetDevice(g->device));
    if (int rc = ensure(g, &g->d_huffdec, &g->huffdec_bytes, (size_t)nimg * sizeof(lep_huffdec_image))) return rc;
    HIPCHK(g, hipMemcpyAsync(g->d_huffdec, images, (size_t)nimg * sizeof(lep_huffdec_image), hipMemcpyHostToDevice, st));
    HIPCHK(g, hipStreamSynchronize(st));   // the caller's array may go away
    HIPCHK(g, hipEventRecord(g->ev0, st));
    hipLaunchKernelGGL(lep_huffman_decode_kernel, dim3(nimg), dim3(64), 0, st, (const lephuff::HuffDecImage*)g->d_huffdec,
                       (lephuff::HuffDecRow*)d_rows);
    HIPCHK(g, hipGetLastError());
    HIPCHK(g, hipEventRecord(g->ev1, st));
    g->timed = true;
    g->last_kernel = "lep_huffman_decode_kernel";
    return 0;
}

int lep_gpu_huffman_decode_simt_device(lep_gpu* g, const lep_huffdec_image* images, int nimg, lep_huffdec_row* d_rows, void* hip_stream) {
    if (!g) return LEP_GPU_ERROR;
    if (nimg <= 0) return 0;
    uint64_t bits = 0;
    // (restart intervals: only with the markers' positions behind the scan bytes -- LEP_HUFFDEC_RST_TABLE; the others are the single-wave kernel's)
    for (int i = 0; i < nimg; ++i) { if (images[i].rsti && !(images[i].flags & LEP_HUFFDEC_RST_TABLE)) return LEP_ASSERTION_FAILURE; bits += (uint64_t)images[i].scan_len * 8u; }
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : g->stream;
    HIPCHK(g, hipSetDevice(g->device));
    // subsequences: as many as fill the chip a few times over (lanes = 64 x the wavefronts it holds, twice), but none shorter than a scan
    // needs to fall into step
    const uint32_t L = g->simt_sub_bits ? (uint32_t)((g->simt_sub_bits + 31) & ~31) : lephuff::simt_sub_bits(bits, (uint64_t)64 * 8192 * 2);
    std::vector<lephuff::SimtImage> si((size_t)nimg);
    std::vector<lephuff::SimtWave> waves;
    size_t nsub_all = 0;
    for (int i = 0; i < nimg; ++i) {
        memset(&si[(size_t)i], 0, sizeof(lephuff::SimtImage));
        const uint64_t b = (uint64_t)images[i].scan_len * 8u;
        // a subsequence has to hold enough blocks to fall into step in: 64 of this image's average block at least (a 4:4:4 file of
        // noise at quality 98 codes 1.5 kbit per block and does not settle in 16 kbit)
        uint64_t nblocks = 0;
        for (int ci = 0; ci < images[i].ncomp && ci < 4; ++ci) { const int cmp = images[i].scan_cmp[ci] & 3; nblocks += (uint64_t)images[i].hs[cmp] * images[i].vs[cmp]; }
        nblocks *= (uint64_t)std::max(images[i].mcuc, 1);
        const uint32_t Li = g->simt_sub_bits ? L : (uint32_t)std::min<uint64_t>(std::max<uint64_t>(L, (64 * b / std::max<uint64_t>(nblocks, 1) + 31) & ~(uint64_t)31), 1u << 24);
        uint32_t n = (uint32_t)std::max<uint64_t>(1, (b + Li - 1) / Li);
        if (images[i].flags & LEP_HUFFDEC_RST_TABLE) {        // lane = restart interval; the pad patterns' and / or / count start from 0xff / 0 / 0
            if (images[i].rsti <= 0 || images[i].mcuc <= 0) return LEP_ASSERTION_FAILURE;
            n = (uint32_t)((images[i].mcuc - 1) / images[i].rsti) + 1u;
            si[(size_t)i].changed[0] = 0xff;
        }
        si[(size_t)i].first = (uint32_t)nsub_all; si[(size_t)i].nsub = n; si[(size_t)i].sub_bits = Li;
        for (uint32_t f = 0; f < n; f += 64) waves.push_back(lephuff::SimtWave{(uint32_t)i, f});
        nsub_all += n;
    }
    if (nsub_all > 0x7fffffffu) return LEP_ASSERTION_FAILURE;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_si = 0, o_wv = up(si.size() * sizeof(lephuff::SimtImage)), o_s0 = o_wv + up(waves.size() * sizeof(lephuff::SimtWave)),
                 o_s1 = o_s0 + up(nsub_all * sizeof(lephuff::SimtSub)), o_pl = o_s1 + up(nsub_all * sizeof(lephuff::SimtSub)), total = o_pl + up(nsub_all * sizeof(lephuff::SimtPlace));
    if (int rc = ensure(g, &g->d_huffdec, &g->huffdec_bytes, (size_t)nimg * sizeof(lep_huffdec_image))) return rc;
    if (int rc = ensure(g, &g->d_huffpar, &g->huffpar_bytes, total)) return rc;
    char* base = (char*)g->d_huffpar;
    HIPCHK(g, hipMemcpyAsync(g->d_huffdec, images, (size_t)nimg * sizeof(lep_huffdec_image), hipMemcpyHostToDevice, st));
    HIPCHK(g, hipMemcpyAsync(base + o_si, si.data(), si.size() * sizeof(lephuff::SimtImage), hipMemcpyHostToDevice, st));
    HIPCHK(g, hipMemcpyAsync(base + o_wv, waves.data(), waves.size() * sizeof(lephuff::SimtWave), hipMemcpyHostToDevice, st));
    HIPCHK(g, hipStreamSynchronize(st));   // the caller's array (and ours) may go away
    const lephuff::HuffDecImage* di = (const lephuff::HuffDecImage*)g->d_huffdec;
    lephuff::SimtImage* dsi = (lephuff::SimtImage*)(base + o_si);
    const lephuff::SimtWave* dwv = (const lephuff::SimtWave*)(base + o_wv);
    lephuff::SimtSub* buf[2] = {(lephuff::SimtSub*)(base + o_s0), (lephuff::SimtSub*)(base + o_s1)};
    lephuff::SimtPlace* dpl = (lephuff::SimtPlace*)(base + o_pl);
    const int nw = (int)waves.size();
    HIPCHK(g, hipEventRecord(g->ev0, st));
    hipLaunchKernelGGL(lep_huffman_simt_settle_kernel, dim3(nw), dim3(64), 0, st, di, dsi, dwv, (const lephuff::SimtSub*)buf[1], buf[0], 0);
    for (int k = 1; k <= lephuff::kSimtSettle; ++k)
        hipLaunchKernelGGL(lep_huffman_simt_settle_kernel, dim3(nw), dim3(64), 0, st, di, dsi, dwv, (const lephuff::SimtSub*)buf[(k - 1) & 1], buf[k & 1], k);
    const lephuff::SimtSub* fin = buf[lephuff::kSimtSettle & 1];
    hipLaunchKernelGGL(lep_huffman_simt_place_kernel, dim3(nimg), dim3(64), 0, st, di, dsi, fin, dpl, lephuff::kSimtSettle, (lephuff::HuffDecRow*)d_rows);
    hipLaunchKernelGGL(lep_huffman_simt_write_kernel, dim3(nw), dim3(64), 0, st, di, dsi, dwv, fin, (const lephuff::SimtPlace*)dpl, (lephuff::HuffDecRow*)d_rows);
    hipLaunchKernelGGL(lep_huffman_simt_finish_kernel, dim3((nimg + 255) / 256), dim3(256), 0, st, di, nimg, (lephuff::HuffDecRow*)d_rows, (const lephuff::SimtImage*)dsi);
    HIPCHK(g, hipGetLastError());
    HIPCHK(g, hipEventRecord(g->ev1, st));
    g->timed = true;
    g->last_kernel = "lep_huffman_simt_{settle,place,write}_kernel";
    return 0;
}

int lep_gpu_expect_company(lep_gpu* g, int on) {
    if (!g) return LEP_GPU_ERROR;
    g->dec_company = on ? 1 : 0;
    return 0;
}

int lep_gpu_use_arena(lep_gpu* g, int k) {
    if (!g || k < 0 || k > 1) return LEP_ASSERTION_FAILURE;
    g->cur = k;
    return 0;
}

// Called by a caller that is about to destroy HIP streams it passed to launch functions of this object (the batch pipelines' per-call
// streams): the upload ring's events were recorded on those streams, and an event whose stream is gone must not be waited on again --
// hipEventSynchronize then answers from freed memory (seen on the MI355X as "operation not permitted on an event last recorded in a
// capturing stream" from a call that never captures anything, one run in a few).  Waits for every slot's copy and forgets the events.
int lep_gpu_settle_uploads(lep_gpu* g) {
    if (!g) return LEP_GPU_ERROR;
    for (lep_gpu::Staging& sl : g->staging) {
        if (sl.used && sl.ev) (void)hipEventSynchronize(sl.ev);
        sl.used = false;
        if (sl.ev) { (void)hipEventDestroy(sl.ev); sl.ev = nullptr; }
    }
    (void)hipGetLastError();
    return 0;
}

int lep_gpu_sync(lep_gpu* g) {
    HIPCHK(g, hipSetDevice(g->device));
    HIPCHK(g, hipDeviceSynchronize());
    return 0;
}

const char* lep_gpu_last_kernel_name(lep_gpu* g) { return g ? g->last_kernel : ""; }
int lep_gpu_device(lep_gpu* g) { return g ? g->device : -1; }
// "0000:c1:00.0" of the device the object was created on (hipDeviceGetPCIBusId): which physical GPU a rank of a multi-GPU run drives
int lep_gpu_pci_bus_id(lep_gpu* g, char* out, int cap) {
    if (!g || !out || cap < 16) return LEP_ASSERTION_FAILURE;
    HIPCHK(g, hipDeviceGetPCIBusId(out, cap, g->device));
    return 0;
}

double lep_gpu_last_kernel_ms(lep_gpu* g) {
    if (!g->timed) return -1.0;
    float ms = 0;
    if (hipEventSynchronize(g->ev1) != hipSuccess) return -1.0;
    if (hipEventElapsedTime(&ms, g->ev0, g->ev1) != hipSuccess) return -1.0;
    return ms;
}

// stage times of the most recent split-phase encode launch: count + plan, emit, fold, gather, write (ms; -1 when the launch was
// not one); returns the number of stages
int lep_gpu_last_stage_ms(lep_gpu* g, double* ms, int cap) {
    if (!g || g->nstage <= 0) return 0;
    if (hipEventSynchronize(g->ev_stage[g->nstage]) != hipSuccess) return 0;
    int n = 0;
    for (; n < g->nstage && n < cap; ++n) {
        float t = 0;
        ms[n] = hipEventElapsedTime(&t, g->ev_stage[n], g->ev_stage[n + 1]) == hipSuccess ? (double)t : -1.0;
    }
    return n;
}

// device self-test of the arithmetic that has no CPU twin (float-reciprocal division in lep3::prob_of); 0 = all exact
int lep_gpu_selftest(lep_gpu* g) {
    HIPCHK(g, hipSetDevice(g->device));
    uint32_t* d = nullptr;
    HIPCHK(g, hipMalloc((void**)&d, 4));
    HIPCHK(g, hipMemset(d, 0, 4));
    hipLaunchKernelGGL(lep_selftest_kernel, dim3(255), dim3(256), 0, g->stream, d);
    uint32_t h = 1;
    HIPCHK(g, hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, g->stream));
    HIPCHK(g, hipStreamSynchronize(g->stream));
    HIPCHK(g, hipFree(d));
    return h == 0 ? 0 : LEP_ASSERTION_FAILURE;
}

// diagnosis: the lane-per-unit scan encoder's work area as the last lep_gpu_huffman_encode_device left it (segment / wave descriptors,
// unit positions, plain prefix sums, bit buffers + marker maps; the layout is that function's), copied to `out`; returns the bytes copied.
// scripts/diag_scan_encode_isolate.py compares two builds of the library pass by pass with it.
size_t lep_gpu_debug_huffenc(lep_gpu* g, void* out, size_t cap) {
    if (!g || !g->d_huffenc[g->cur & 1]) return 0;
    if (hipSetDevice(g->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 0;
    const size_t n = std::min(cap, g->huffenc_bytes[g->cur & 1]);
    return hipMemcpy(out, g->d_huffenc[g->cur & 1], n, hipMemcpyDeviceToHost) == hipSuccess ? n : 0;
}

// profiling builds (-DLEP_PROF) only: per-phase shader-clock totals folded over the segments of the last decoder launch
int lep_gpu_debug_prof(lep_gpu* g, uint64_t* out /* [64][32] */) {
#ifdef LEP_PROF
    std::vector<unsigned long long> h(8192 * 32);   // fold the per-wave accumulators into the 64 x 32 report
    HIPCHK(g, hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_prof4), sizeof(unsigned long long) * 8192 * 32));
    memset(out, 0, sizeof(uint64_t) * 64 * 32);
    for (int s = 0; s < 8192; ++s)
        for (int i = 0; i < 32; ++i) out[(s & 63) * 32 + i] += h[(size_t)s * 32 + i];
    return 0;
#else
    (void)g; (void)out;
    return LEP_GPU_ERROR;
#endif
}

// give back what the object caches between launches (models, neighbour rings, the split-phase encoder's scratch): all of it
// is re-acquired by the next launch that needs it.  Waits for the device first.
int lep_gpu_trim(lep_gpu* g) {
    HIPCHK(g, hipSetDevice(g->device));
    HIPCHK(g, hipDeviceSynchronize());
    void** ps[] = {&g->enc5.d_entries, &g->enc5.d_binlist, &g->arena[0].d_models, &g->arena[1].d_models, &g->arena[0].d_ns, &g->arena[1].d_ns};
    size_t* ns[] = {&g->enc5.entries_bytes, &g->enc5.binlist_bytes, &g->arena[0].models_bytes, &g->arena[1].models_bytes, &g->arena[0].ns_bytes, &g->arena[1].ns_bytes};
    for (int i = 0; i < 6; ++i) dev_release(g, ps[i], ns[i]);
    return 0;
}
// ... and hand them to the driver (the pool too): for a process that wants the device's memory for something else
int lep_gpu_release_memory(lep_gpu* g) {
    if (int rc = lep_gpu_trim(g)) return rc;
    vmm_drain_pool(g);
    return 0;
}
int lep_gpu_malloc(lep_gpu* g, size_t bytes, void** dptr) {
    HIPCHK(g, hipSetDevice(g->device));
    if (hipMalloc(dptr, bytes ? bytes : 16) != hipSuccess) {   // (no room: the caches first, pool included, then the batch calls' staging)
        (void)hipGetLastError();
        if (int rc = lep_gpu_release_memory(g)) return rc;
        if (hipMalloc(dptr, bytes ? bytes : 16) != hipSuccess) {
            (void)hipGetLastError();
            lep_batch_release();
            HIPCHK(g, hipMalloc(dptr, bytes ? bytes : 16));
        }
    }
    return 0;
}
int lep_gpu_free(lep_gpu* g, void* dptr) { HIPCHK(g, hipFree(dptr)); return 0; }
int lep_gpu_memcpy_h2d(lep_gpu* g, void* dst, const void* src, size_t bytes) { HIPCHK(g, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return 0; }
int lep_gpu_memcpy_d2h(lep_gpu* g, void* dst, const void* src, size_t bytes) { HIPCHK(g, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return 0; }
int lep_gpu_memcpy_d2d(lep_gpu* g, void* dst, const void* src, size_t bytes) { HIPCHK(g, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice)); return 0; }
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
