// jpeg_recode.cc -- coefficient frame -> the original JPEG bytes (sequential JPEGs): the host-side
// consumer of the decode hot path.  Behaviour follows
//   recode_baseline_jpeg   src/lepton/recoder.cc:694-889
//   recode_physical_thread src/lepton/recoder.cc:560-652  (per-segment byte bounds)
//   recode_row_range       src/lepton/recoder.cc:471-545
//   recode_one_mcu_row     src/lepton/recoder.cc:316-412
//   encode_block_seq       src/lepton/recoder.cc:245-314
//   handle_initial_segments src/lepton/recoder.cc:414-460
#include <algorithm>
#include <thread>
#include <cstring>

#include "jpeg_bits.h"
#include "lep_container.h"
#include "../../include/lepton_mi355x.h"   // lep_huff_end

namespace lep {

int next_mcupos(const JpegFile& jf, int* mcu, int* cmp, int* csc, int* sub, int* dpos, int* rstw, int cs_cmpc);
int next_mcuposn(const JpegFile& jf, int cmp, int* dpos, int* rstw);

namespace {

struct BoundedOut {   // bounded_iostream / BoundedMemWriter: bytes past the bound are dropped but counted
    std::vector<uint8_t> buf;
    size_t bound = 0, attempted = 0;
    bool shut = false;   // a bound of zero bytes (a worker's buffer resized to nothing); bound == 0 alone means "none"
    void put(uint8_t b) { ++attempted; if (!shut && (!bound || buf.size() < bound)) buf.push_back(b); }
    void write(const uint8_t* d, size_t n) { for (size_t i = 0; i < n; ++i) put(d[i]); }
    bool exceeded() const { return shut ? attempted > 0 : (bound && attempted > bound); }
    bool reached() const { return shut || (bound && buf.size() >= bound); }
};

static inline int blen16(unsigned v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }
static inline unsigned envli(int s, int v) { return (unsigned)((v > 0) ? v : (v - 1) + (1 << s)) & ((1u << s) - 1); }

void encode_block(BitWriter& w, const HuffTable& dc, const HuffTable& ac, const int16_t* blk) {
    int t = blk[0];
    int s = blen16((unsigned)(t > 0 ? t : -t) & 0xffff);
    w.put(dc.cval[s], dc.clen[s]);
    w.put(envli(s, t), s);
    int end = 63;
    while (end && !blk[end]) --end;
    int z = 0;
    for (int b = 1; b <= end; ++b) {
        t = blk[b];
        if (!t) { ++z; continue; }
        s = blen16((unsigned)(t > 0 ? t : -t) & 0xffff);
        while (z & 0xf0) { w.put(ac.cval[0xF0], ac.clen[0xF0]); z -= 16; }
        int hc = ((z & 0xf) << 4) + s;
        w.put(ac.cval[hc & 255], ac.clen[hc & 255]);
        w.put(envli(s, t), s);
        z = 0;
    }
    if (end != 63) w.put(ac.cval[0], ac.clen[0]);
}

// move whole bytes from the bit writer to the output, stuffing 00 after FF
void drain(BitWriter& w, BoundedOut& out) {
    for (uint8_t b : w.bytes) { out.put(b); if (b == 0xFF) out.put(0); }
    w.bytes.clear();
}

struct RowCoder {
    LepFile& lf;
    JpegFile& jf;
    int ncomp;
    int seg_first_mcu_row = 0;   // first MCU row of the thread segment being written
    RowCoder(LepFile& l) : lf(l), jf(l.jpeg), ncomp(l.jpeg.ncomp) {}

    // Huffman-code one MCU row starting at MCU index `mcu`; false on error
    void mcu_row(BitWriter& w, int mcu, BoundedOut& out, int16_t lastdc[4]) {
        int cmp = jf.cs_cmp[0], csc = 0, sub = 0;
        const int mcumul = jf.comp[cmp].hs * jf.comp[cmp].vs;
        int dpos = mcu * mcumul;
        int rstw = jf.rsti ? jf.rsti - mcu % jf.rsti : 0;
        unsigned cum_rst = rstw ? (unsigned)(mcu / jf.rsti) : 0;
        bool end_of_row = false;
        int16_t blk[64];
        while (!end_of_row) {
            int sta = 0;
            while (sta == 0) {
                const int16_t* src = jf.plane[cmp] + (size_t)dpos * 64;
                if (jf.early_eof && dpos >= jf.trunc_bc[cmp] && dpos % jf.comp[cmp].bch != 0) {   // (a row's first block is always decoded: decode_row, lepton_codec.cc:7-47)
                    // A block behind the point where the file was cut: the reference's baseline decoder keeps only two block rows
                    // per component (block_based_image.hh:60-66,84-95) and never touches these blocks, so its re-coder reads what
                    // row y - 2 left in the ring (zeros if this thread never decoded that row).  Unobservable in an intact file --
                    // the byte bound cuts the output first -- but not when damaged streams make the scan shorter
                    // (tests/test_fuzz_host.py::test_recoder_rules_...).
                    static const int16_t kZeroBlock[64] = {0};
                    const Component& kc = jf.comp[cmp];
                    const int up = dpos - 2 * kc.bch, mult = std::max(kc.bcv / std::max(jf.mcuv, 1), 1);
                    src = (up >= 0 && (up / kc.bch) / mult >= seg_first_mcu_row) ? jf.plane[cmp] + (size_t)up * 64 : kZeroBlock;
                }
                for (int b = 0; b < 64; ++b) blk[b] = src[kZigzagToAligned[b]];
                int16_t dc = blk[0];
                blk[0] = (int16_t)(blk[0] - lastdc[cmp]);
                lastdc[cmp] = dc;
                const Component& k = jf.comp[cmp];
                encode_block(w, jf.htab[0][k.dc_tbl], jf.htab[1][k.ac_tbl], blk);
                int old_mcu = mcu;
                if (ncomp == 1) { sta = next_mcuposn(jf, cmp, &dpos, &rstw); mcu = dpos / mcumul; }
                else sta = next_mcupos(jf, &mcu, &cmp, &csc, &sub, &dpos, &rstw, ncomp);
                if (sta == 0 && w.buffer_empty()) drain(w, out);   // (the bytes of a row reach the output at its end, see BitWriter::phase)
                if (out.exceeded()) sta = 2;
                if (old_mcu != mcu && mcu % jf.mcuh == 0) {
                    end_of_row = true;
                    if (sta == 0) return;
                }
            }
            w.pad((uint8_t)jf.padbit);
            drain(w, out);
            if (sta == 2) break;
            if (sta == 1 && jf.rsti > 0) {
                if (jf.rst_cnt.empty() || !lf.rst_cnt_set || cum_rst < jf.rst_cnt[0]) {
                    out.put(0xFF);
                    out.put((uint8_t)(0xD0 + (cum_rst & 7)));
                    ++cum_rst;
                }
                rstw = jf.rsti;
                lastdc[0] = lastdc[1] = lastdc[2] = lastdc[3] = 0;
            }
        }
    }
};

}  // namespace

// ---- split form of recode_jpeg: everything except the Huffman coding of the segments ---------------------------------------
// recode_prepare: header walk (DHT / DRI / SOS), the bytes in front of the scan, and -- when the file is eligible -- the
// parameters the GPU Huffman encoder (lep_huff.h) needs per image and per thread segment.
int recode_prepare(LepFile* lf, RecodePlan* plan) {
    JpegFile& jf = lf->jpeg;
    plan->gpu_ok = false;
    plan->segs.clear();
    if (lf->flag != 'Z') return 0;   // progressive / multi-scan files: recode_progressive (jpeg_progressive.cc), host only
    const size_t max_file_size = lf->jpeg_size;
    if ((int32_t)lf->jpeg_size <= (int32_t)jf.garbage.size()) return EX_ASSERTION_FAILURE;   // always_assert(max_file_size > grbs), both ints
    plan->scan_bound = max_file_size - jf.garbage.size();
    size_t pos = 0;
    const uint8_t* h = jf.hdr.data();
    const size_t hdrs = jf.hdr.size();
    for (;;) {
        if (pos + 3 >= hdrs) return EX_UNSUPPORTED_JPEG;   // "overran headers" / "not start of segment": recode_baseline_jpeg fails, errorlevel 2 (jpgcoder.cc:1334-1338)
        if (h[pos] != 0xff) return EX_UNSUPPORTED_JPEG;
        uint8_t type = h[pos + 1];
        unsigned len = 2 + ((unsigned)h[pos + 2] << 8) + h[pos + 3];
        if (type == 0xC4 || type == 0xDD || type == 0xDA)
            if (!parse_segment(&jf, type, len, (unsigned)std::min<size_t>(len, hdrs - pos), h + pos, false)) return jf.warn < 0 ? -jf.warn : EX_UNSUPPORTED_JPEG;   // parse_jfif_jpg: errorlevel 2
        pos += len;
        if (type == 0xDA) break;
    }
    plan->hdr_pos = pos;
    plan->head.clear();
    if (lf->has_prefix) plan->head.insert(plan->head.end(), lf->prefix_garbage.begin(), lf->prefix_garbage.end());
    if (lf->embedded || !lf->has_prefix) {
        plan->head.push_back(0xFF); plan->head.push_back(0xD8);
        plan->head.insert(plan->head.end(), h, h + std::min(pos, hdrs));
        // an SOS whose length field reaches past the stored header: the reference writes `pos` bytes from its header buffer
        // (handle_initial_segments, recoder.cc:443-456), i.e. reads on into zero-filled arena memory -- zeros up to the bound
        if (pos > hdrs) plan->head.resize(plan->head.size() + std::min(pos - hdrs, plan->scan_bound), 0);
    }
    if (plan->head.size() > plan->scan_bound) plan->head.resize(plan->scan_bound);

    // eligibility for the GPU encoder: whole, untruncated frames; modern hand-offs; an MCU-interleaved scan of all
    // components or a plain single-component scan.  Everything else takes the host path (recode_jpeg).
    plan->gpu_ok = false;
    plan->segs.clear();
    // A file that was cut inside its scan (early_eof): its blocks behind the cut are ones the reference's decoder never wrote, and what its
    // re-coder reads there (RowCoder::mcu_row above) only shows in damaged files -- in a file the reference compressed, the byte bound of
    // the last thread cuts the output in front of them.  The lane-per-unit scan encoder (lep_huff_simt.h) stops at the first such block
    // and says so; the caller takes its bytes when the bound was reached by then and the host re-coder otherwise.  Only that kernel knows
    // the cut: interleaved scans of two or three components without restart intervals.
    const bool cut = jf.early_eof;
    bool ok = jf.ncomp >= 1 && jf.ncomp <= 3 && jf.cs_cmpc == jf.ncomp && jf.mcuh > 0 && jf.mcuv > 0 &&
              (cut ? (jf.ncomp >= 2 && jf.rsti == 0) : jf.trunc_bcv[0] >= jf.comp[0].bcv) && !lf->segs.empty();
    for (const Handoff& th : lf->segs) if (th.num_overhang_bits == 0xff || th.num_overhang_bits > 7) ok = false;
    for (size_t q = 1; q < lf->segs.size(); ++q) if (lf->version > 1 && !lf->segs[q].segment_size) ok = false;   // a worker bound of nothing: host path
    if ((size_t)std::min(lf->nthreads, 8) != lf->segs.size()) ok = false;   // several logical threads folded onto one worker (a damaged thread hint): cumulative bounds, host path
    if (ok && lf->version > 1 && (uint64_t)plan->head.size() + lf->segs[0].segment_size > 0xffffffffull) ok = false;   // the first thread's bound wraps (recode_jpeg): host path
    // One component: never interleaved -- the scan codes the nch x ncv blocks the picture covers and steps over the blocks that pad the
    // frame to whole MCUs (recode_one_mcu_row with next_mcuposn, recoder.cc:316-412); an MCU row of the hand-offs is bcv / mcuv block
    // rows.  The kernels get that as a frame of nch x ncv MCUs of one block, block rows bch apart, and segments in block rows.  With
    // sampling factors or padding blocks the re-coder counts restart intervals in MCUs of hs x vs blocks where the scan counted blocks
    // (the reference cannot restore such files: tests/test_sampling_layouts.py): those stay with the host re-coder, which is held to that.
    const bool planar = jf.ncomp == 1;
    bool plain = false;
    if (ok && planar) {
        const Component& k = jf.comp[jf.cs_cmp[0]];
        plain = k.hs == 1 && k.vs == 1 && k.bch == k.nch && k.bcv == k.ncv && k.bc == jf.mcuc;
        if (!plain) ok = jf.rsti == 0 && jf.cs_cmp[0] == 0 && k.nch >= 1 && k.ncv >= 1 && k.nch <= k.bch && k.ncv <= k.bcv && k.bcv % jf.mcuv == 0 && (jf.mcuv - 1) * (k.bcv / jf.mcuv) < k.ncv;
    }
    for (int c = 0; ok && c < jf.ncomp; ++c) {
        const Component& k = jf.comp[c];
        if (!jf.htab[0][k.dc_tbl].set || !jf.htab[1][k.ac_tbl].set || k.dc_tbl > 1 || k.ac_tbl > 1) ok = false;
        if (jf.ncomp > 1 && (k.bch != jf.mcuh * k.hs || k.bcv != jf.mcuv * k.vs)) ok = false;
    }
    if (!ok) return 0;
    RecodeImage& im = plan->image;
    memset(&im, 0, sizeof im);
    im.ncomp = jf.ncomp; im.mcuh = jf.mcuh; im.mcuv = jf.mcuv; im.mcuc = jf.mcuc; im.rsti = jf.rsti; im.padbit = jf.padbit;
    im.rst_limit = (jf.rst_cnt.empty() || !lf->rst_cnt_set) ? 0xffffffffu : jf.rst_cnt[0];
    im.interleaved = 1;
    if (planar) { im.mcuh = jf.comp[0].nch; im.mcuv = jf.comp[0].ncv; im.mcuc = im.mcuh * im.mcuv; }
    for (int c = 0; c < 4; ++c) im.trunc_bc[c] = (cut && c < jf.ncomp) ? std::max(1, jf.trunc_bc[c]) : 0;
    for (int c = 0; c < jf.ncomp; ++c) {
        im.hs[c] = planar ? 1 : jf.comp[c].hs; im.vs[c] = planar ? 1 : jf.comp[c].vs; im.bch[c] = jf.comp[c].bch;
        im.dc_tbl[c] = jf.comp[c].dc_tbl; im.ac_tbl[c] = jf.comp[c].ac_tbl;
        im.scan_cmp[c] = jf.cs_cmp[c];
    }
    for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 256; ++i) {
            im.code[t][i] = jf.htab[0][t].set ? ((uint32_t)jf.htab[0][t].clen[i] << 16) | jf.htab[0][t].cval[i] : 0u;
            im.code[2 + t][i] = jf.htab[1][t].set ? ((uint32_t)jf.htab[1][t].clen[i] << 16) | jf.htab[1][t].cval[i] : 0u;
        }
    const int luma_mul = jf.comp[0].bcv / jf.mcuv;
    size_t blocks_per_mcu = 0, total_cap = 0;
    for (int c = 0; c < jf.ncomp; ++c) blocks_per_mcu += jf.ncomp > 1 ? (size_t)jf.comp[c].hs * jf.comp[c].vs : 1;
    for (size_t s = 0; s < lf->segs.size(); ++s) {
        const Handoff& th = lf->segs[s];
        RecodeSegment g;
        memset(&g, 0, sizeof g);
        int r0 = jf.mcuv, r1 = 0;
        for (int row = 0; row < jf.mcuv; ++row) {
            const int y0 = row * luma_mul, y1 = y0 + luma_mul;
            if (y0 >= jf.trunc_bcv[0]) break;
            if (y0 < th.luma_y_start) continue;
            if (y1 > th.luma_y_end) break;
            r0 = std::min(r0, row); r1 = row + 1;
        }
        if (r1 <= r0) { r0 = r1 = 0; }
        g.mcu_row0 = r0; g.mcu_row1 = r1;
        if (planar) { g.mcu_row0 = std::min(r0 * luma_mul, im.mcuv); g.mcu_row1 = std::min(r1 * luma_mul, im.mcuv); }   // block rows
        g.overhang = (uint32_t)th.overhang_byte | ((uint32_t)th.num_overhang_bits << 8);
        memcpy(g.last_dc, th.last_dc, sizeof g.last_dc);
        const size_t room = plan->scan_bound - plan->head.size();
        // format version 1 leaves the first thread's output unbounded; from version 2 on every thread is bound by its
        // segment size (recode_physical_thread, recoder.cc:598-613)
        size_t cap = (s == 0 && lf->version == 1) ? room : (th.segment_size ? (size_t)th.segment_size : max_file_size);
        // segment_size comes from an untrusted header (up to 4 GB per hand-off) and sizes a pinned + device arena slot in
        // the batch pipeline: never reserve more than the file may hold, nor more than the segment's blocks can possibly
        // code to -- 64 coefficients x (16-bit code + 11 magnitude bits) = 216 bytes, every one of them 0xFF and stuffed,
        // plus a restart marker per MCU at worst
        const size_t seg_mcus = planar ? (size_t)(g.mcu_row1 - g.mcu_row0) * (size_t)im.mcuh : (size_t)(r1 - r0) * (size_t)jf.mcuh;
        const size_t seg_blocks = seg_mcus * std::max<size_t>(1, blocks_per_mcu);
        const size_t geo = seg_blocks * 432 + seg_mcus * 2 + 64;
        cap = std::min(std::min(cap, room), geo);
        if (s > 0) total_cap += cap;   // segment 0 is only bounded by the file; the caller gives it what the others leave
        g.out_cap = (uint32_t)std::min<size_t>(cap, 0xffffffffu);
        plan->segs.push_back(g);
    }
    // later segments that together claim more than the file can hold are not what an encoder writes: the host re-coder
    // (whose buffers grow with what is really written) takes such files
    if (total_cap > (plan->scan_bound - plan->head.size()) + lf->segs.size() * 8192) { plan->segs.clear(); return 0; }
    plan->gpu_ok = true;
    return 0;
}

// recode_finish: glue head + per-segment scan bytes + misplaced RST markers + the rest of the header + garbage together
int recode_finish(LepFile* lf, const RecodePlan& plan, const std::vector<std::pair<const uint8_t*, size_t>>& seg_bytes, const lep_huff_end* ends,
                  std::vector<uint8_t>* result) {
    JpegFile& jf = lf->jpeg;
    const size_t max_file_size = lf->jpeg_size;
    BoundedOut out;
    out.bound = plan.scan_bound;
    out.buf.reserve(std::min<size_t>(max_file_size, (size_t)128 << 20) + 16);   // (a SIZ section can claim up to 2^31 - 1)
    out.write(plan.head.data(), plan.head.size());
    // The state a logical thread ends in must be the state the next hand-off recorded: the reference asserts partial byte, bit
    // count and last DCs at every segment end (recode_physical_thread, recoder.cc:625-640; recode_jpeg above does the same for
    // the host path).  `ends` is what lep_huffman_encode_kernel hands back (one logical thread per physical thread here:
    // recode_prepare sends everything else to the host re-coder).  The byte count -- a segment that restores its part of the
    // file exactly writes exactly segment_size bytes -- stays as the check for callers without end states.
    // a truncated file's segments (recode_prepare): what the scan encoder said about the cut.  Bytes that stop at the cut are the file's
    // only if the thread's byte bound was reached by then (then nothing behind the cut can show) and only in the last thread; anything
    // else is the host re-coder's, which knows what the reference reads behind a cut: EX_GPU_PATH_DECLINED tells the caller so
    if (ends)
        for (size_t q = 0; q < seg_bytes.size() && q < plan.segs.size(); ++q) {
            if (ends[q].pad & 2) return EX_GPU_PATH_DECLINED;
            if ((ends[q].pad & 1) && (q + 1 != seg_bytes.size() || ends[q].attempted < plan.segs[q].out_cap)) return EX_GPU_PATH_DECLINED;
        }
    if (seg_bytes.size() == lf->segs.size())
        for (size_t q = 0; q + 1 < seg_bytes.size(); ++q) {
            const Handoff& nx = lf->segs[q + 1];
            if (nx.num_overhang_bits == 0xff) continue;
            if (nx.luma_y_start != nx.luma_y_end || lf->version == 1) {
                if (ends) {
                    if (ends[q].num_overhang_bits != nx.num_overhang_bits || ends[q].overhang_byte != nx.overhang_byte) return EX_ASSERTION_FAILURE;
                    if (memcmp(ends[q].last_dc, nx.last_dc, 3 * sizeof(int16_t))) return EX_ASSERTION_FAILURE;
                }
                if (seg_bytes[q].second != lf->segs[q].segment_size && !(q == 0 && out.buf.size() + seg_bytes[q].second >= out.bound)) return EX_ASSERTION_FAILURE;
            }
        }
    for (const auto& sb : seg_bytes) out.write(sb.first, sb.second);
    if (!jf.rst_err.empty()) {
        unsigned cum = jf.rsti ? (unsigned)((jf.mcuh * jf.mcuv - 1) / jf.rsti) : 0;
        for (unsigned i = 0; i < jf.rst_err[0]; ++i) { out.put(0xFF); out.put((uint8_t)(0xD0 + ((cum + i) & 7))); }
    }
    const uint8_t* h = jf.hdr.data();
    const size_t hdrs = jf.hdr.size();
    if (!out.reached() && plan.hdr_pos < hdrs) out.write(h + plan.hdr_pos, hdrs - plan.hdr_pos);
    out.bound = max_file_size;
    out.write(jf.garbage.data(), jf.garbage.size());
    result->swap(out.buf);
    return 0;
}

// The thread segments of a planned file (recode_prepare said gpu_ok) written on host threads, one per segment: what the GPU scan
// encoders do, with the row coder above.  Every segment starts from its hand-off (partial byte, last DCs) and is bound by its
// out_cap; bytes and end states go to recode_finish like the kernels'.  Used by lep_jpeg_check_restores, where the one-thread walk of
// recode_jpeg was a quarter of a 4K file's compression time.
int recode_segments_on_threads(LepFile* lf, const RecodePlan& plan, std::vector<std::vector<uint8_t>>* seg_bytes, std::vector<lep_huff_end>* ends) {
    if (!plan.gpu_ok) return EX_ASSERTION_FAILURE;
    const JpegFile& jf = lf->jpeg;
    const size_t n = plan.segs.size();
    seg_bytes->assign(n, std::vector<uint8_t>());
    ends->assign(n, lep_huff_end());
    auto one = [&](size_t s) {
        const RecodeSegment& g = plan.segs[s];
        RowCoder rc(*lf);
        rc.seg_first_mcu_row = g.mcu_row0;
        BitWriter w;
        w.fillbit = (uint8_t)jf.padbit;
        w.seed((uint8_t)(g.overhang & 255u), (int)((g.overhang >> 8) & 255u));
        BoundedOut o;
        o.bound = g.out_cap;
        o.shut = g.out_cap == 0;
        int16_t lastdc[4];
        memcpy(lastdc, g.last_dc, sizeof lastdc);
        for (int row = g.mcu_row0; row < g.mcu_row1; ++row) {
            rc.mcu_row(w, row * jf.mcuh, o, lastdc);
            drain(w, o);
            w.row_flush();
        }
        lep_huff_end& e = (*ends)[s];
        e.attempted = (uint32_t)std::min<size_t>(o.attempted, 0xffffffffu);
        e.overhang_byte = w.overhang_bits() ? w.overhang_byte() : (uint8_t)0;
        e.num_overhang_bits = (uint8_t)w.overhang_bits();
        memcpy(e.last_dc, lastdc, sizeof lastdc);
        e.pad = 0;
        (*seg_bytes)[s].swap(o.buf);
    };
    // Nothing may leave this function as an exception: it is called from extern "C" entry points, and a vector of joinable threads
    // unwound by one would end the process (a thread limit under lepton_served, an allocation failure in a worker).  A segment that
    // could not be coded is reported and the caller falls back to the one-thread re-coder.
    std::vector<char> failed(n, 0);
    auto guarded = [&](size_t s) {
        try { one(s); } catch (...) { failed[s] = 1; }
    };
    std::vector<std::thread> pool;
    pool.reserve(n);
    size_t started = 1;
    try {
        for (; started < n; ++started) pool.emplace_back(guarded, started);
    } catch (...) {
        // std::thread's constructor threw (EAGAIN): the segments that have no thread are coded on this one
    }
    if (n) guarded(0);
    for (size_t s = started; s < n; ++s) guarded(s);
    for (std::thread& t : pool) t.join();
    for (size_t s = 0; s < n; ++s) if (failed[s]) return EX_ASSERTION_FAILURE;
    return 0;
}

int recode_progressive(LepFile* lf, std::vector<uint8_t>* result);   // jpeg_progressive.cc

// The part of recode_baseline_jpeg that runs before the first row is decoded (recoder.cc:694-705): the all-garbage assertion and
// handle_initial_segments -- the header walked up to the first SOS, its DHT / DRI / SOS segments interpreted.  A file refused
// here is refused with this verdict whatever its streams hold; lep_file_open_next asks before anything is decoded.
int baseline_header_pass(LepFile* lf) {
    JpegFile& jf = lf->jpeg;
    size_t pos = 0;
    const uint8_t* h = jf.hdr.data();
    const size_t hdrs = jf.hdr.size();
    if (!(lf->flag == 'Z' || (lf->flag & 1) == ('Y' & 1))) {
        // the general re-coder (recode_jpeg, jpgcoder.cc:3345-3372) waits for decoded blocks position by position; the tables in
        // front of its first scan are interpreted before it waits for anything
        uint8_t type = 0;
        while (type != 0xDA) {
            if (pos >= hdrs) break;
            type = pos + 1 < hdrs ? h[pos + 1] : 0;
            const unsigned len = 2 + (((unsigned)(pos + 2 < hdrs ? h[pos + 2] : 0)) << 8) + (pos + 3 < hdrs ? h[pos + 3] : 0);
            if (type == 0xC4 || type == 0xDA || type == 0xDD)
                if (!parse_segment(&jf, type, len, (unsigned)std::min<size_t>(len, hdrs - pos), h + pos, false)) return jf.warn < 0 ? -jf.warn : EX_UNSUPPORTED_JPEG;
            pos += len;
        }
        // no scan at all: nothing is ever waited for, and the re-coder ends in its all-garbage assertion or its empty scan
        // table ("out of memory error", errorlevel 2) whatever the streams hold -- see recode_progressive
        if (type != 0xDA) return (int32_t)lf->jpeg_size <= (int32_t)jf.garbage.size() ? EX_ASSERTION_FAILURE : EX_UNSUPPORTED_JPEG;
        return 0;
    }
    if ((int32_t)lf->jpeg_size <= (int32_t)jf.garbage.size()) return EX_ASSERTION_FAILURE;
    for (;;) {
        if (pos + 3 >= hdrs) return EX_UNSUPPORTED_JPEG;   // "overran headers"
        if (h[pos] != 0xff) return EX_UNSUPPORTED_JPEG;    // "not start of segment"
        const uint8_t type = h[pos + 1];
        const unsigned len = 2 + ((unsigned)h[pos + 2] << 8) + h[pos + 3];
        if (type == 0xC4 || type == 0xDD || type == 0xDA)
            if (!parse_segment(&jf, type, len, (unsigned)std::min<size_t>(len, hdrs - pos), h + pos, false)) return jf.warn < 0 ? -jf.warn : EX_UNSUPPORTED_JPEG;
        pos += len;
        if (type == 0xDA) return 0;
    }
}

int recode_jpeg(LepFile* lf, std::vector<uint8_t>* result) {
    JpegFile& jf = lf->jpeg;
    // 'Z' and 'Y' (a -startbyte slice) take the baseline re-coder, 'X' the general one: read_fixed_ujpg_header tests
    // header[1] == 'Z' || (header[1] & 1) == ('Y' & 1), jpgcoder.cc:2162-2166
    if (!(lf->flag == 'Z' || (lf->flag & 1) == ('Y' & 1))) return recode_progressive(lf, result);
    const size_t max_file_size = lf->jpeg_size;
    if ((int32_t)lf->jpeg_size <= (int32_t)jf.garbage.size()) return EX_ASSERTION_FAILURE;   // always_assert(max_file_size > grbs), both ints
    BoundedOut out;
    out.bound = max_file_size - jf.garbage.size();

    // 1. header segments up to and including the first SOS (parsing DHT / DRI / SOS on the way)
    size_t pos = 0;
    const uint8_t* h = jf.hdr.data();
    const size_t hdrs = jf.hdr.size();
    for (;;) {
        if (pos + 3 >= hdrs) return EX_UNSUPPORTED_JPEG;   // "overran headers" / "not start of segment": recode_baseline_jpeg fails, errorlevel 2 (jpgcoder.cc:1334-1338)
        if (h[pos] != 0xff) return EX_UNSUPPORTED_JPEG;
        uint8_t type = h[pos + 1];
        unsigned len = 2 + ((unsigned)h[pos + 2] << 8) + h[pos + 3];
        if (type == 0xC4 || type == 0xDD || type == 0xDA)
            if (!parse_segment(&jf, type, len, (unsigned)std::min<size_t>(len, hdrs - pos), h + pos, false)) return jf.warn < 0 ? -jf.warn : EX_UNSUPPORTED_JPEG;   // parse_jfif_jpg: errorlevel 2
        pos += len;
        if (type == 0xDA) break;
    }
    if (lf->has_prefix) out.write(lf->prefix_garbage.data(), lf->prefix_garbage.size());
    if (lf->embedded || !lf->has_prefix) {
        out.put(0xFF); out.put(0xD8);
        out.write(h, std::min(pos, hdrs));
        for (size_t k = hdrs; k < pos; ++k) out.put(0);   // (see recode_prepare: a length field past the stored header)
    }

    // 2. the scan, logical thread by logical thread, grouped the way the reference groups them onto its physical threads
    // (recode_baseline_jpeg / recode_physical_thread, recoder.cc:560-651, 757-800): physical thread 0 writes straight into the
    // bounded output, every other one into a buffer of its own that is as large as its logical threads' segment sizes together
    // (the whole file if that is zero) and is appended when all are done.  A file is normally one logical thread per physical
    // thread; a damaged thread hint or hand-off count folds several onto one, and then the bounds are cumulative.
    RowCoder rc(*lf);
    const int luma_mul = jf.comp[0].bcv / jf.mcuv;
    const int L = (int)lf->segs.size();
    const bool one_thread = lf->segs[0].num_overhang_bits == 0xff;   // a pre-hand-off first record: g_threaded = false (recoder.cc:730-732)
    const int P = one_thread ? 1 : std::max(1, std::min(lf->nthreads, 8));
    auto range_of = [&](int p, int* a, int* b) {   // logical_thread_range_from_physical_thread_id, recoder.cc:547-559
        *a = p * L / P; *b = std::min((p + 1) * L / P, L);
        if (L < P) { *a = std::min(p, L); *b = std::min(p + 1, L); }
    };
    const size_t file_bound = out.bound;
    for (int p = 0; p < P; ++p) {
        int first, last;
        range_of(p, &first, &last);
        if (first >= last) continue;
        BoundedOut worker;                 // physical threads >= 1
        BoundedOut* o = p ? &worker : &out;
        size_t original_bound = file_bound;
        if (p) {
            int32_t work = 0;
            for (int l = first; l < last; ++l) work = (int32_t)((uint32_t)work + lf->segs[l].segment_size);
            if (!work) work = (int32_t)lf->jpeg_size;
            original_bound = work > 0 ? (size_t)work : (size_t)0x7fffffff;   // (a negative size is OOM when the file is opened)
            worker.bound = original_bound;
        }
        bool changed_bounds = false;
        Handoff carry = lf->segs[first];
        for (int s = first; s < last; ++s) {
            Handoff th = lf->segs[s];
            const bool legacy = th.num_overhang_bits == 0xff;
            if (legacy) {
                // a pre-hand-off record takes the state the previous logical thread of the SAME physical thread ended in; the first
                // one a physical thread runs starts clean: no pending bits, the byte and the last DCs of its own record
                // (recoder.cc:584-592)
                if (s == first) carry.num_overhang_bits = 0;
                th.overhang_byte = carry.overhang_byte;
                th.num_overhang_bits = carry.num_overhang_bits;
                memcpy(th.last_dc, carry.last_dc, sizeof th.last_dc);
            } else {
                // format 1 leaves the first thread's output unbounded and bounds a worker by its buffer; several logical threads
                // on one physical thread -- and, from format 2 on, every thread -- are bound to bytes_written + segment_size where
                // that is tighter (recoder.cc:598-613).  For physical thread 0 that is a sum of two 32-bit values
                // (bounded_iostream::bytes_written() is an unsigned int): a segment size near 2^32 wraps it to a bound in front of
                // what is already written, and the next write trips always_assert(byte_position <= byte_bound) (bitops.cc:402);
                // a bound of exactly zero means "none" there.  A worker's buffer is RESIZED to its bound: zero leaves room for nothing.
                const bool many_to_one = last - first != 1 && s != 0;
                if (many_to_one || lf->version > 1) {
                    if (p == 0) {
                        const uint32_t nb = (uint32_t)o->buf.size() + th.segment_size;
                        if ((size_t)nb < original_bound) {
                            if (nb && (size_t)nb < o->buf.size()) return EX_ASSERTION_FAILURE;
                            o->bound = nb; o->attempted = std::min(o->attempted, o->buf.size()); changed_bounds = true;
                        } else if (o->bound != original_bound) { o->bound = original_bound; o->attempted = std::min(o->attempted, o->buf.size()); }
                    } else {
                        const size_t nb = o->buf.size() + (size_t)th.segment_size;
                        if (nb < original_bound) { o->bound = nb; o->shut = nb == 0; changed_bounds = true; }
                        else if (o->bound != original_bound) { o->bound = original_bound; o->shut = false; }
                    }
                }
            }
            BitWriter w;
            w.fillbit = (uint8_t)jf.padbit;
            w.seed(th.overhang_byte, th.num_overhang_bits);
            int16_t lastdc[4];
            memcpy(lastdc, th.last_dc, sizeof lastdc);
            rc.seg_first_mcu_row = th.luma_y_start / std::max(luma_mul, 1);
            for (int mcu_row = 0; mcu_row < jf.mcuv; ++mcu_row) {
                int y0 = mcu_row * luma_mul, y1 = y0 + luma_mul;
                if (y0 >= jf.trunc_bcv[0]) break;                 // rows past the coded height are skipped
                if (y0 < th.luma_y_start) continue;
                if (y1 > th.luma_y_end) break;
                rc.mcu_row(w, mcu_row * jf.mcuh, *o, lastdc);
                drain(w, *o);
                w.row_flush();
            }
            carry.overhang_byte = w.overhang_byte();
            carry.num_overhang_bits = (uint8_t)w.overhang_bits();
            memcpy(carry.last_dc, lastdc, sizeof lastdc);
            // The state a logical thread ends in must be the state the next hand-off recorded -- the reference asserts it
            // (recoder.cc:625-645) and that is what stops a truncated or damaged multi-segment .lep from being "restored" as
            // garbage: partial byte, its bit count, the last DC of every component (skipped for an empty next segment of a
            // format >= 2 file), and a worker that has written anything must have filled its bound exactly.
            if (s + 1 < L && lf->segs[s + 1].num_overhang_bits != 0xff) {
                const Handoff& nx = lf->segs[s + 1];
                if (nx.luma_y_start != nx.luma_y_end || lf->version == 1) {
                    if (!one_thread && (carry.num_overhang_bits != nx.num_overhang_bits || carry.overhang_byte != nx.overhang_byte)) return EX_ASSERTION_FAILURE;
                    if ((!one_thread || nx.segment_size > 1) && memcmp(carry.last_dc, nx.last_dc, 3 * sizeof(int16_t))) return EX_ASSERTION_FAILURE;
                }
                if (p > 0 && !o->buf.empty() && o->bound != o->buf.size()) return EX_ASSERTION_FAILURE;
            }
        }
        if (p == 0) { if (changed_bounds || out.bound != file_bound) { out.bound = file_bound; out.attempted = std::min(out.attempted, out.buf.size()); } }
        else out.write(worker.buf.data(), worker.buf.size());
    }

    // 3. wrongly placed RST markers at the end of the scan, then the rest of the header, then garbage
    if (!jf.rst_err.empty()) {
        unsigned cum = jf.rsti ? (unsigned)((jf.mcuh * jf.mcuv - 1) / jf.rsti) : 0;
        for (unsigned i = 0; i < jf.rst_err[0]; ++i) { out.put(0xFF); out.put((uint8_t)(0xD0 + ((cum + i) & 7))); }
    }
    if (!out.reached() && pos < hdrs) out.write(h + pos, hdrs - pos);
    out.bound = max_file_size;
    out.write(jf.garbage.data(), jf.garbage.size());
    result->swap(out.buf);
    return 0;
}

}  // namespace lep
