// lep_api.cc -- C ABI, layers 2 and 3 (host-side callers of the GPU hot path).  See include/lepton_mi355x.h.
#include "../../include/lepton_mi355x.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "jpeg_model.h"
#include "lep_container.h"

struct lep_jpeg {
    lep::JpegFile jf;
    lep::EncodeOptions opt;
};
struct lep_file {
    lep::LepFile lf;
    bool frame_ready = false;
    lep::RecodePlan plan;
    bool planned = false;
    lep::ProgPlan prog;
    bool prog_planned = false;
};
static_assert(sizeof(lep_huffprog_image) == sizeof(lep::ProgImage) && sizeof(lep_huffprog_scan) == sizeof(lep::ProgScan), "C ABI mirrors");
static_assert(sizeof(lep_huffdec_image) == sizeof(lep::ScanDecodePlan) && sizeof(lep_huffdec_row) == sizeof(lep::ScanDecodeRow), "C ABI mirrors");
static_assert(sizeof(lep_huffprogdec_scan) == sizeof(lep::ProgScanDecodePlan), "C ABI mirrors");
static_assert(sizeof(lep_huff_image) == sizeof(lep::RecodeImage) && sizeof(lep_huff_segment) == sizeof(lep::RecodeSegment), "C ABI mirrors");

static void fill_desc(const lep::JpegFile& jf, lep_image_desc* d, int16_t* const* planes) {
    memset(d, 0, sizeof *d);
    d->ncomp = jf.ncomp;
    d->mcu_rows = jf.mcuv;
    for (int c = 0; c < jf.ncomp && c < LEP_MAX_COMPONENTS; ++c) {
        d->width_blocks[c] = jf.comp[c].bch;
        d->height_blocks[c] = jf.comp[c].bcv;
        d->coded_blocks[c] = jf.trunc_bc[c];
        d->coded_height[c] = jf.trunc_bcv[c];
        memcpy(d->qtable_zigzag[c], jf.qtables[jf.comp[c].qidx], 128);
        d->blocks[c] = planes[c];
    }
}

static int to_bytes(const std::vector<uint8_t>& v, lep_bytes* out) {
    out->data = (uint8_t*)malloc(v.size() ? v.size() : 1);
    if (!out->data) return LEP_OS_ERROR;
    memcpy(out->data, v.data(), v.size());
    out->len = out->cap = v.size();
    return 0;
}

extern "C" {

#ifndef LEP_SRC_SHA16
#define LEP_SRC_SHA16 "unknown"
#endif
// (build() compiles the hash of every source file into this string: a loaded library can be held against the files beside it)
const char* lep_version(void) { return "lepton-mi355x 0.1 (format v1, gfx950, src " LEP_SRC_SHA16 ")"; }
void lep_free(void* p) { free(p); }

int lep_jpeg_open(const uint8_t* jpg, size_t len, int allow_progressive, lep_jpeg** out) {
    return lep_jpeg_open_into(jpg, len, allow_progressive, nullptr, 0, out);
}

// SOF geometry only (no scan decode): bytes of the coefficient frame parse_jpeg will produce, MCU padding included
int lep_jpeg_peek_frame_bytes(const uint8_t* d, size_t n, size_t* bytes) {
    *bytes = 0;
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return LEP_UNSUPPORTED_JPEG;
    size_t p = 2;
    while (p + 4 <= n) {
        if (d[p] != 0xFF) return LEP_UNSUPPORTED_JPEG;
        const uint8_t m = d[p + 1];
        if (m == 0xFF) { ++p; continue; }
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) { p += 2; continue; }
        const size_t len = ((size_t)d[p + 2] << 8) | d[p + 3];
        if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            if (p + 2 + len > n || len < 8) return LEP_UNSUPPORTED_JPEG;
            const int h = (d[p + 5] << 8) | d[p + 6], w = (d[p + 7] << 8) | d[p + 8], nc = d[p + 9];
            if (nc < 1 || nc > 4 || len < 8 + 3 * (size_t)nc || !w || !h) return LEP_UNSUPPORTED_JPEG;
            int hs[4], vs[4], hmax = 1, vmax = 1;
            for (int c = 0; c < nc; ++c) {
                hs[c] = d[p + 11 + 3 * c] >> 4; vs[c] = d[p + 11 + 3 * c] & 15;
                if (!hs[c] || !vs[c]) return LEP_UNSUPPORTED_JPEG;
                hmax = std::max(hmax, hs[c]); vmax = std::max(vmax, vs[c]);
            }
            const size_t mcuh = ((size_t)w + 8 * hmax - 1) / (8 * hmax), mcuv = ((size_t)h + 8 * vmax - 1) / (8 * vmax);
            size_t b = 0;
            for (int c = 0; c < nc; ++c) b += mcuh * hs[c] * mcuv * vs[c] * 128;
            if (b > lep::kMaxFrameBlocks * 128) return lep::EX_TOO_MUCH_MEMORY_NEEDED;   // the frame budget of setup_frame
            *bytes = b;
            return 0;
        }
        if (m == 0xDA || m == 0xD9) break;
        p += 2 + len;
    }
    return LEP_UNSUPPORTED_JPEG;
}

int lep_jpeg_open_into(const uint8_t* jpg, size_t len, int allow_progressive, void* frame_mem, size_t frame_cap, lep_jpeg** out) {
    std::unique_ptr<lep_jpeg> j(new lep_jpeg);
    j->jf.ext_mem = (int16_t*)frame_mem; j->jf.ext_cap = frame_mem ? frame_cap : 0;
    j->opt.allow_progressive = allow_progressive != 0;
    int rc = lep::parse_jpeg(jpg, len, allow_progressive != 0, &j->jf);
    if (rc) return rc;
    *out = j.release();
    return 0;
}
int lep_jpeg_open_slice(const uint8_t* jpg, size_t len, size_t start_byte, lep_jpeg** out) {
    if (start_byte > 0xffffffffu) return LEP_ASSERTION_FAILURE;
    std::unique_ptr<lep_jpeg> j(new lep_jpeg);
    j->jf.start_byte = (uint32_t)start_byte;
    j->opt.allow_progressive = start_byte == 0;
    int rc = lep::parse_jpeg(jpg, len, start_byte == 0, &j->jf);
    if (rc) return rc;
    *out = j.release();
    return 0;
}
int lep_jpeg_open_embedded(const uint8_t* blob, size_t len, size_t offset, lep_jpeg** out) {
    if (offset + 4 > len || len > 0xffffffffu) return LEP_UNSUPPORTED_JPEG;
    std::unique_ptr<lep_jpeg> j(new lep_jpeg);
    j->opt.allow_progressive = true;
    int rc = lep::parse_jpeg(blob + offset, len - offset, true, &j->jf);
    if (rc) return rc;
    // hand-off byte positions are only ever used as differences (plan_segments), so they may stay relative to the JPEG;
    // the file size and the prefix are the blob's
    j->jf.embedded = true;
    j->jf.prefix_garbage.assign(blob, blob + offset);
    j->jf.file_size = (uint32_t)len;
    *out = j.release();
    return 0;
}
int lep_jpeg_open_gpu(const uint8_t* jpg, size_t len, lep_jpeg** out, lep_huffdec_image* image, int* eligible) {
    std::unique_ptr<lep_jpeg> j(new lep_jpeg);
    j->opt.allow_progressive = true;
    bool ok = false;
    int rc = lep::parse_jpeg_prepare_gpu(jpg, len, &j->jf, reinterpret_cast<lep::ScanDecodePlan*>(image), &ok);
    if (rc) return rc;
    *eligible = ok ? 1 : 0;
    *out = j.release();
    return 0;
}
int lep_jpeg_open_gpu_progressive(lep_jpeg* j, lep_huffprogdec_scan* scans, int cap, int* nscan, int* rows_needed, int* eligible) {
    *eligible = 0; *nscan = 0; *rows_needed = 0;
    std::vector<lep::ProgScanDecodePlan> v;
    bool ok = false;
    int need = 0;
    int rc = lep::parse_jpeg_prepare_gpu_progressive(&j->jf, &v, &need, &ok);
    if (rc) return rc;
    if (!ok || (int)v.size() > cap) return 0;
    memcpy(scans, v.data(), v.size() * sizeof(lep_huffprogdec_scan));
    *nscan = (int)v.size(); *rows_needed = need; *eligible = 1;
    return 0;
}
int lep_jpeg_finish_gpu_progressive(lep_jpeg* j, const lep_huffprogdec_scan* scans, int nscan, const lep_huffdec_row* rows) {
    if (nscan <= 0 || !scans || !rows) return LEP_ASSERTION_FAILURE;   // (public entry: v[0] is read below)
    std::vector<lep::ProgScanDecodePlan> v((size_t)nscan);
    memcpy(v.data(), scans, (size_t)nscan * sizeof(lep_huffprogdec_scan));
    // the descriptors as the caller launched them carry device addresses and arena offsets: rebase on the file's own
    const uint64_t rows0 = v[0].t.rows_off;
    for (auto& sc : v) { sc.result_off -= rows0; sc.t.rows_off -= rows0; }
    return lep::parse_jpeg_finish_gpu_progressive(&j->jf, v, reinterpret_cast<const lep::ScanDecodeRow*>(rows) + rows0) ? LEP_UNSUPPORTED_JPEG : 0;
}
int lep_jpeg_plan_progressive_check(lep_jpeg* j, size_t jpeg_len, lep_huffprog_image* image, lep_huffprog_scan* scans, uint32_t* file_first,
                                    uint32_t* file_len, int cap, int* nscan, int* eligible) {
    *eligible = 0; *nscan = 0;
    lep::ProgPlan plan;
    // (rst_cnt as the parser counted it: a scan that holds fewer restart markers than its length asks for is not planned)
    if (lep::progressive_plan(&j->jf, jpeg_len, true, &plan) || !plan.gpu_ok) return 0;
    const size_t n = plan.scans.size();
    if ((int)n > cap || n != j->jf.scan_file_range.size()) return 0;
    for (size_t q = 0; q < n; ++q) {
        const auto& r = j->jf.scan_file_range[q];
        if (r.second <= r.first || r.second > jpeg_len) return 0;
        file_first[q] = r.first; file_len[q] = r.second - r.first;
    }
    memcpy(image, &plan.image, sizeof *image);
    memcpy(scans, plan.scans.data(), sizeof(lep_huffprog_scan) * n);
    *nscan = (int)n;
    *eligible = 1;
    return 0;
}
int lep_jpeg_scan_bytes(const lep_jpeg* j, const uint8_t** data, size_t* len) {
    *data = j->jf.scan.data(); *len = j->jf.scan.size();
    return 0;
}
int lep_jpeg_scan_restarts(const lep_jpeg* j, const uint32_t** pos, size_t* count) {
    *pos = j->jf.rst_pos.data(); *count = j->jf.rst_pos.size();
    return 0;
}
int lep_jpeg_finish_gpu(lep_jpeg* j, const lep_huffdec_row* rows) {
    return lep::parse_jpeg_finish_gpu(&j->jf, reinterpret_cast<const lep::ScanDecodeRow*>(rows)) ? LEP_UNSUPPORTED_JPEG : 0;
}
void lep_jpeg_close(lep_jpeg* j) { delete j; }

int lep_jpeg_describe(const lep_jpeg* j, lep_image_desc* d) {
    int16_t* planes[4];
    for (int c = 0; c < 4; ++c) planes[c] = j->jf.plane[c];
    fill_desc(j->jf, d, planes);
    return 0;
}

int lep_jpeg_set_encode_options(lep_jpeg* j, int max_threads, int min_threads, int even_split) {
    if (!j) return LEP_ASSERTION_FAILURE;
    if (max_threads > 0) j->opt.max_threads = (unsigned)std::min(max_threads, LEP_MAX_SEGMENTS);
    if (min_threads > 0) j->opt.min_threads = (unsigned)std::min(min_threads, LEP_MAX_SEGMENTS);
    j->opt.even_split = even_split != 0;
    return 0;
}

// `lepton -brotliheader` (jpgcoder.cc:1116-1119): container format 2 -- brotli header, packet end marker.  Needs the reference's own
// brotli encoder in the library (build() compiles it from where it lies); LEP_VERSION_UNSUPPORTED when it is not there.
int lep_jpeg_set_container_version(lep_jpeg* j, int version) {
    if (!j) return LEP_ASSERTION_FAILURE;
    if (version != 1 && version != 2) return LEP_VERSION_UNSUPPORTED;
    if (version == 2 && !lep::brotli_encoder_available()) return LEP_VERSION_UNSUPPORTED;
    j->opt.format_version = version;
    return 0;
}
int lep_container_can_write_version(int version) { return version == 1 || (version == 2 && lep::brotli_encoder_available()) ? 1 : 0; }

int lep_jpeg_is_progressive(const lep_jpeg* j) { return j->jf.progressive_needed ? 1 : 0; }

int lep_jpeg_plan(const lep_jpeg* j, int max_threads, lep_segment* segs, int image_index) {
    lep::EncodeOptions o = j->opt;
    if (max_threads > 0) o.max_threads = (unsigned)max_threads;
    std::vector<lep::Handoff> s = lep::plan_segments(j->jf, o);
    for (size_t i = 0; i < s.size(); ++i) {
        segs[i].image = image_index;
        segs[i].luma_y_start = s[i].luma_y_start;
        segs[i].luma_y_end = s[i].luma_y_end;
        segs[i].is_last = i + 1 == s.size();
    }
    return (int)s.size();
}

// the same choice as lep_jpeg_plan, as hand-off records: segment_size = the JPEG scan bytes the segment covers
int lep_jpeg_plan_handoffs(const lep_jpeg* j, int max_threads, lep_handoff* out, int cap) {
    lep::EncodeOptions o = j->opt;
    if (max_threads > 0) o.max_threads = (unsigned)max_threads;
    std::vector<lep::Handoff> s = lep::plan_segments(j->jf, o);
    if ((int)s.size() > cap) return -LEP_BUFFER_TOO_SMALL;
    for (size_t i = 0; i < s.size(); ++i) {
        out[i].luma_y_start = s[i].luma_y_start; out[i].luma_y_end = s[i].luma_y_end; out[i].segment_size = s[i].segment_size;
        out[i].overhang_byte = s[i].overhang_byte; out[i].num_overhang_bits = s[i].num_overhang_bits;
        memcpy(out[i].last_dc, s[i].last_dc, sizeof out[i].last_dc);
    }
    return (int)s.size();
}

// The Huffman half of the round-trip check for a baseline file, planned for the GPU (validation.cc:97-218 restores the file and
// compares): the parameters lep_gpu_huffman_encode_device needs to write the file's scan again from its coefficient frame --
// exactly what lep_file_recode_plan would hand out for the .lep this file is about to become (same header, same hand-offs: the
// plan is made through the same recode_prepare) -- and, per thread segment, where the bytes it must reproduce stand in the file.
int lep_jpeg_plan_scan_check(lep_jpeg* j, size_t jpeg_len, lep_huff_image* image, lep_huff_segment* segs, uint32_t* file_first, uint32_t* file_len, int cap,
                             int* nseg, int* eligible) {
    *eligible = 0; *nseg = 0;
    if (!j || j->jf.scan_file_range.size() != 1 || j->jf.progressive_needed || j->jf.start_byte || j->jf.embedded || j->jf.early_eof) return 0;
    std::vector<lep::Handoff> hs = lep::plan_segments(j->jf, j->opt);
    if (hs.empty() || (int)hs.size() > cap) return 0;
    lep::LepFile lf;
    lf.version = j->opt.format_version; lf.flag = 'Z'; lf.nthreads = (int)hs.size(); lf.jpeg_size = (uint32_t)jpeg_len;
    lf.segs = hs;
    lf.rst_cnt_set = !j->jf.rst_cnt.empty();
    // the parsed JPEG is lent to the plan (recode_prepare re-reads the tables in front of the scan from the same header bytes)
    // and handed back whatever happens; an empty garbage section stands for the default EOI, as in a .lep that is read back
    struct Lend {
        lep::JpegFile& home; lep::JpegFile& away; bool default_eoi;
        Lend(lep::JpegFile& h, lep::JpegFile& a) : home(h), away(a), default_eoi(h.garbage.empty()) { away = std::move(home); if (default_eoi) away.garbage = {0xFF, 0xD9}; }
        ~Lend() { if (default_eoi) away.garbage.clear(); home = std::move(away); }
    };
    lep::RecodePlan plan;
    int rc;
    {
        Lend lend(j->jf, lf.jpeg);
        rc = lep::recode_prepare(&lf, &plan);
    }
    if (rc || !plan.gpu_ok || plan.segs.size() != hs.size()) return 0;
    const auto& r = j->jf.scan_file_range[0];
    uint64_t at = r.first;
    for (size_t q = 0; q < hs.size(); ++q) {
        file_first[q] = (uint32_t)at; file_len[q] = hs[q].segment_size;
        at += hs[q].segment_size;
    }
    // the hand-offs' byte counts must tile the scan as it stands in the file: up to the marker that ends it, less the restart
    // markers that stand wrongly at its end (the re-coder appends those from the FRS section; they are not a segment's bytes)
    const uint64_t tail_rst = j->jf.rst_err.empty() ? 0u : 2u * (uint64_t)j->jf.rst_err[0];
    if (at + tail_rst != r.second || at > jpeg_len) return 0;
    memcpy(image, &plan.image, sizeof *image);
    memcpy(segs, plan.segs.data(), sizeof(lep_huff_segment) * plan.segs.size());
    *nseg = (int)hs.size();
    *eligible = 1;
    return 0;
}

// where the (single) scan of a parsed baseline file stands in the file: first entropy-coded byte, length up to the marker that ends it
int lep_jpeg_scan_file_range(const lep_jpeg* j, uint32_t* first, uint32_t* len) {
    if (!j || j->jf.scan_file_range.size() != 1) return LEP_ASSERTION_FAILURE;
    *first = j->jf.scan_file_range[0].first; *len = j->jf.scan_file_range[0].second - j->jf.scan_file_range[0].first;
    return 0;
}

int lep_jpeg_write_lep(const lep_jpeg* j, int max_threads, const lep_bytes* streams, int nstreams, lep_bytes* out) {
    lep::EncodeOptions o = j->opt;
    if (max_threads > 0) o.max_threads = (unsigned)max_threads;
    std::vector<lep::Handoff> s = lep::plan_segments(j->jf, o);
    if ((int)s.size() != nstreams) return LEP_ASSERTION_FAILURE;
    std::vector<std::vector<uint8_t>> st(nstreams);
    for (int i = 0; i < nstreams; ++i) st[i].assign(streams[i].data, streams[i].data + streams[i].len);
    std::vector<uint8_t> file;
    int rc = lep::write_lep(j->jf, s, st, &file, j->opt.format_version);
    if (rc) return rc;
    return to_bytes(file, out);
}

int lep_file_open(const uint8_t* d, size_t len, lep_file** out) { return lep_file_open_next(d, len, nullptr, out); }

size_t lep_file_consumed(const lep_file* f) { return f->lf.consumed; }

int lep_file_open_next(const uint8_t* d, size_t len, const lep_file* prev, lep_file** out) {
    std::unique_ptr<lep_file> f(new lep_file);
    int rc = lep::parse_lep(d, len, &f->lf, prev && prev->lf.header_pending ? &prev->lf.pending_header : nullptr);
    if (rc) return rc;
    lep::JpegFile& jf = f->lf.jpeg;
    // (the embedded header has been interpreted inside parse_lep, where the reference does it)
    if (jf.warn > 0) return LEP_UNSUPPORTED_JPEG;   // errorlevel 1 in setup_imginfo_jpg (an unknown marker in the embedded header, jpgcoder.cc:4836-4840) stops the reference too
    if (jf.ncomp > 3) return LEP_UNSUPPORTED_4_COLORS;
    if (jf.early_eof) {
        for (int c = 0; c < jf.ncomp; ++c) {
            const lep::Component& k = jf.comp[c];
            int tbc = jf.max_dpos[c] + 1;
            int lines = std::min(tbc / k.bch + (tbc % k.bch ? 1 : 0), k.bcv);
            int ratio = std::max(k.bcv / jf.mcuv, 1);
            while (lines % ratio != 0 && lines + 1 <= k.bcv) ++lines;
            jf.trunc_bcv[c] = lines;
            jf.trunc_bc[c] = tbc;
        }
    }
    // pre-hand-off split tables must cut on MCU rows: luma_y_end % (luma block rows per MCU row) -> THREADING_PARTIAL_MCU
    // (vp8_decoder.cc:353-360; the last entry is not read from the file and not checked)
    if (!f->lf.segs.empty() && f->lf.segs[0].num_overhang_bits == 0xff && jf.mcuv > 0) {
        const int lcm = std::max(jf.comp[0].bcv / jf.mcuv, 1);
        for (size_t i = 0; i + 1 < f->lf.segs.size(); ++i)
            if (f->lf.segs[i].luma_y_end % lcm) return LEP_THREADING_PARTIAL_MCU;
    }
    // the general re-coder decodes through decode_chunk, which gives up when the file has more logical threads than the
    // decoder was started with: `num_threads_needed > NUM_THREADS` -> CODING_ERROR (vp8_decoder.cc:415-417), NUM_THREADS =
    // min(8, thread hint) (read_fixed_ujpg_header, jpgcoder.cc:2167-2171) -- after the split table has been read and checked
    if (!(f->lf.flag == 'Z' || (f->lf.flag & 1) == ('Y' & 1)) && f->lf.segs.size() > (size_t)std::min(f->lf.nthreads, 8)) return LEP_CODING_ERROR;
    if (!f->lf.segs.empty()) f->lf.segs.back().luma_y_end = (uint16_t)jf.trunc_bcv[0];   // vp8_decoder.cc:366-368
    // the baseline re-coder looks at the header's tables and allocates its workers' buffers before it decodes a row
    if (int hrc = lep::baseline_header_pass(&f->lf)) return hrc;
    if (lep::worker_bounds_exceed_arena(f->lf, f->lf.consumed)) return LEP_OOM;   // this file's own extent, not what is concatenated behind it
    if (f->lf.unbound_stream_packet) return LEP_ASSERTION_FAILURE;   // "Cannot send to thread that wasn't bound": behind the header's own refusals
    *out = f.release();
    return 0;
}
void lep_file_close(lep_file* f) { delete f; }
uint32_t lep_file_jpeg_size(const lep_file* f) { return f->lf.jpeg_size; }
size_t lep_file_frame_bytes(const lep_file* f) {
    size_t b = 0;
    for (int c = 0; c < f->lf.jpeg.ncomp; ++c) b += (size_t)f->lf.jpeg.comp[c].bc * 128;
    return b;
}

int lep_file_describe(lep_file* f, lep_image_desc* d) { return lep_file_describe_into(f, nullptr, 0, d); }

// frame_mem (optional): caller-provided storage for the coefficient frame (planes back to back), not zeroed by this call --
// the batch pipeline points it at the pinned buffer the decoded frame is copied into
int lep_file_describe_into(lep_file* f, void* frame_mem, size_t frame_cap, lep_image_desc* d) {
    lep::JpegFile& jf = f->lf.jpeg;
    if (!frame_mem && frame_cap == (size_t)-1) {   // geometry only: the frame lives elsewhere (device memory)
        int16_t* none[4] = {nullptr, nullptr, nullptr, nullptr};
        fill_desc(jf, d, none);
        return 0;
    }
    if (!f->frame_ready || frame_mem) {
        jf.ext_mem = (int16_t*)frame_mem; jf.ext_cap = frame_mem ? frame_cap : 0;
        jf.place_frame(false);
        f->frame_ready = true;
    }
    fill_desc(jf, d, jf.plane);
    return 0;
}

int lep_file_segments(const lep_file* f, lep_segment* segs, lep_bytes* streams, int image_index) {
    const auto& s = f->lf.segs;   // <= LEP_MAX_SEGMENTS: parse_lep refuses files with more
    // The baseline re-coder drives its decoder MCU row by MCU row: a row is taken when the MCU row's FIRST luma row is not in
    // front of the hand-off's luma_y_start and its end not behind luma_y_end (recode_row_range, recoder.cc:505-510), so a
    // (damaged) hand-off that starts or ends inside an MCU row skips that MCU row in every component.  The general re-coder's
    // decoder (vp8_decode_thread, lepton_codec.cc:284-300) compares luma rows one by one, which is what the kernels do.
    const lep::JpegFile& jf = f->lf.jpeg;
    const bool by_mcu_row = (f->lf.flag == 'Z' || (f->lf.flag & 1) == ('Y' & 1)) && jf.mcuv > 0;
    const int mul = by_mcu_row ? std::max(jf.comp[0].bcv / jf.mcuv, 1) : 1;
    for (size_t i = 0; i < s.size(); ++i) {
        segs[i].image = image_index;
        segs[i].luma_y_start = (s[i].luma_y_start + mul - 1) / mul * mul;
        segs[i].luma_y_end = i + 1 == s.size() ? s[i].luma_y_end : s[i].luma_y_end / mul * mul;
        segs[i].is_last = i + 1 == s.size();
        streams[i].data = const_cast<uint8_t*>(f->lf.streams[i].data());
        streams[i].len = streams[i].cap = f->lf.streams[i].size();
    }
    return (int)s.size();
}

}  // extern "C"
namespace lep {
// jpeg_recode.cc: the planned segments' scan bytes and end states from host threads (one per segment); recode_finish takes them like the GPU's
int recode_segments_on_threads(LepFile* lf, const RecodePlan& plan, std::vector<std::vector<uint8_t>>* seg_bytes, std::vector<lep_huff_end>* ends);
}
// A file of several thread segments that the split re-coder takes (recode_prepare: the files the GPU scan encoders take), its segments
// written on host threads and glued by recode_finish.  false = not such a file, or something in it the split form refuses: the
// one-thread walk (recode_jpeg) then decides, as it always did.  (8 segments of a 4K file: 5 ms against 40 on the MI355X host.)
static bool recode_on_threads(lep_file* f, std::vector<uint8_t>* jpg) {
    // one file at a time: a caller that re-codes many files side by side (the batch calls' host pool) already has its cores busy
    static std::atomic<int> busy{0};
    struct Turn { bool mine; Turn() : mine(busy.fetch_add(1) == 0) {} ~Turn() { busy.fetch_sub(1); } } turn;
    if (!turn.mine) return false;
    lep::RecodePlan plan;
    if (lep::recode_prepare(&f->lf, &plan) != 0 || !plan.gpu_ok || plan.segs.size() < 2) return false;
    std::vector<std::vector<uint8_t>> bytes;
    std::vector<lep_huff_end> ends;
    if (lep::recode_segments_on_threads(&f->lf, plan, &bytes, &ends) != 0) return false;
    std::vector<std::pair<const uint8_t*, size_t>> sb;
    for (const std::vector<uint8_t>& b : bytes) sb.emplace_back(b.data(), b.size());
    jpg->clear();
    if (lep::recode_finish(&f->lf, plan, sb, ends.data(), jpg) != 0) { jpg->clear(); return false; }
    return true;
}
extern "C" {

int lep_file_recode(lep_file* f, lep_bytes* out) {
    std::vector<uint8_t> jpg;
    if (recode_on_threads(f, &jpg)) return to_bytes(jpg, out);
    int rc = lep::recode_jpeg(&f->lf, &jpg);
    if (rc) return rc;
    return to_bytes(jpg, out);
}

int lep_jpeg_check_restores(const lep_jpeg* j, const uint8_t* lepdata, size_t lep_len, const uint8_t* want, size_t want_len) {
    lep_file* f = nullptr;
    if (int rc = lep_file_open(lepdata, lep_len, &f)) return rc;
    std::unique_ptr<lep_file> hold(f);
    const lep::JpegFile& src = j->jf;
    lep::JpegFile& dst = f->lf.jpeg;
    if (dst.ncomp != src.ncomp) return LEP_ROUNDTRIP_FAILURE;
    for (int c = 0; c < src.ncomp; ++c)
        if (dst.comp[c].bch != src.comp[c].bch || dst.comp[c].bcv != src.comp[c].bcv || !src.plane[c]) return LEP_ROUNDTRIP_FAILURE;
    // the re-coder only reads coefficients: lend it the parsed frame (what the arithmetic decoder would have produced)
    for (int c = 0; c < src.ncomp; ++c) dst.plane[c] = src.plane[c];
    f->frame_ready = true;
    std::vector<uint8_t> jpg;
    // several thread segments: on host threads first.  Only a match is believed -- anything else is decided by the one-thread walk below.
    if (recode_on_threads(f, &jpg) && jpg.size() == want_len && (want_len == 0 || memcmp(jpg.data(), want, want_len) == 0)) return 0;
    jpg.clear();
    if (lep::recode_jpeg(&f->lf, &jpg)) return LEP_ROUNDTRIP_FAILURE;
    return jpg.size() == want_len && (want_len == 0 || memcmp(jpg.data(), want, want_len) == 0) ? 0 : LEP_ROUNDTRIP_FAILURE;
}

int lep_file_recode_plan(lep_file* f, lep_huff_image* image, lep_huff_segment* segs, int* nseg, int* gpu_ok) {
    int rc = lep::recode_prepare(&f->lf, &f->plan);
    if (rc) return rc;
    f->planned = true;
    *gpu_ok = f->plan.gpu_ok ? 1 : 0;
    *nseg = 0;
    if (f->plan.gpu_ok) {
        memcpy(image, &f->plan.image, sizeof *image);
        for (int c = 0; c < f->lf.jpeg.ncomp; ++c) image->blocks[c] = f->lf.jpeg.plane[c];
        *nseg = (int)f->plan.segs.size();
        memcpy(segs, f->plan.segs.data(), sizeof(lep_huff_segment) * f->plan.segs.size());
    }
    return 0;
}

int lep_file_recode_plan_progressive(lep_file* f, lep_huffprog_image* image, lep_huffprog_scan* scans, int cap, int* nscan, int* gpu_ok) {
    *gpu_ok = 0; *nscan = 0;
    int rc = lep::recode_progressive_prepare(&f->lf, &f->prog);
    if (rc) return rc;
    f->prog_planned = true;
    if (!f->prog.gpu_ok || (int)f->prog.scans.size() > cap) { f->prog.gpu_ok = false; return 0; }
    memcpy(image, &f->prog.image, sizeof *image);
    for (int c = 0; c < f->lf.jpeg.ncomp; ++c) image->blocks[c] = f->lf.jpeg.plane[c];
    memcpy(scans, f->prog.scans.data(), sizeof(lep_huffprog_scan) * f->prog.scans.size());
    *nscan = (int)f->prog.scans.size();
    *gpu_ok = 1;
    return 0;
}

int lep_file_recode_finish_progressive(lep_file* f, const lep_bytes* scan_bytes, int nscan, lep_bytes* out) {
    if (!f->prog_planned || !f->prog.gpu_ok) return LEP_ASSERTION_FAILURE;
    std::vector<std::pair<const uint8_t*, size_t>> sb;
    for (int i = 0; i < nscan; ++i) sb.emplace_back(scan_bytes[i].data, scan_bytes[i].len);
    std::vector<uint8_t> jpg;
    int rc = lep::recode_progressive_finish(&f->lf, f->prog, sb, &jpg);
    if (rc) return rc;
    return to_bytes(jpg, out);
}

int lep_file_recode_finish(lep_file* f, const lep_bytes* seg_bytes, const lep_huff_end* ends, int nseg, lep_bytes* out) {
    if (!f->planned) return LEP_ASSERTION_FAILURE;
    std::vector<std::pair<const uint8_t*, size_t>> sb;
    for (int i = 0; i < nseg; ++i) sb.emplace_back(seg_bytes[i].data, seg_bytes[i].len);
    std::vector<uint8_t> jpg;
    int rc = lep::recode_finish(&f->lf, f->plan, sb, ends, &jpg);
    if (rc) return rc;
    return to_bytes(jpg, out);
}

}  // extern "C"

// ---- layer 3: whole files (JPEG -> .lep -> JPEG) through the GPU hot path ---------------------------
extern "C" {

int lep_compress(lep_gpu* g, const uint8_t* jpg, size_t len, lep_bytes* out) { return lep_compress_slice(g, jpg, len, 0, 0, out); }

// want: the bytes the .lep has to restore (the reference's default validation; a file that fails it is refused, not written)
static int compress_parsed(lep_gpu* g, lep_jpeg* j, const uint8_t* want, size_t want_len, lep_bytes* out) {
    std::unique_ptr<lep_jpeg> hold(j);
    lep_image_desc d;
    lep_jpeg_describe(j, &d);
    lep_segment segs[LEP_MAX_SEGMENTS];
    int n = lep_jpeg_plan(j, 0, segs, 0);
    size_t blocks = 0;
    for (int c = 0; c < d.ncomp; ++c) blocks += (size_t)d.width_blocks[c] * d.height_blocks[c];
    std::vector<std::vector<uint8_t>> bufs(n);
    lep_bytes streams[LEP_MAX_SEGMENTS];
    int32_t status[LEP_MAX_SEGMENTS];
    for (int i = 0; i < n; ++i) {
        bufs[i].resize(blocks * 160 / n + blocks * 16 + 65536);   // worst case is far below 2 bytes per coefficient
        streams[i].data = bufs[i].data(); streams[i].cap = bufs[i].size(); streams[i].len = 0;
    }
    int rc = lep_gpu_encode_host(g, &d, 1, segs, n, streams, status);
    if (rc) return rc;
    rc = lep_jpeg_write_lep(j, 0, streams, n, out);
    if (rc) return rc;
    rc = lep_jpeg_check_restores(j, out->data, out->len, want, want_len);
    if (rc) { lep_free(out->data); out->data = nullptr; out->len = out->cap = 0; }
    return rc;
}

int lep_compress_slice(lep_gpu* g, const uint8_t* jpg, size_t len, size_t start_byte, size_t trunc, lep_bytes* out) {
    if (!g) return LEP_GPU_ERROR;
    if (trunc && trunc < len) len = trunc;   // -trunc bounds the reader (check_file -> BindFdToReader, jpgcoder.cc:2181)
    lep_jpeg* j = nullptr;
    int rc = lep_jpeg_open_slice(jpg, len, start_byte, &j);
    if (rc) return rc;
    return compress_parsed(g, j, jpg + start_byte, len - start_byte, out);
}

int lep_compress_embedded(lep_gpu* g, const uint8_t* blob, size_t len, size_t offset, lep_bytes* out) {
    if (!g) return LEP_GPU_ERROR;
    lep_jpeg* j = nullptr;
    int rc = lep_jpeg_open_embedded(blob, len, offset, &j);
    if (rc) return rc;
    return compress_parsed(g, j, blob, len, out);
}

uint64_t lep_jpeg_gpu_scan_wait_timeouts(void) { return lep::prog_wait_timeouts(); }

// How the batch calls cut a batch into pipeline chunks (pure host logic, unit-tested on the CPU).  A chunk's thread segments
// are one decoder wavefront each and the chip holds 8192 of them at once (rounds 2-3 kept the eighth wave slot of every SIMD for
// a single-wavefront Huffman kernel beside a one-kernel encoder; neither is what runs beside the coders any more: the decode kernel's
// time is proportional to its segments either way, profiles/r05v_*), so an automatic chunk holds at most 8192 segments and 1024 images
// (LEP_BATCH_CHUNK_SEGMENTS: measurement knob); a kernel takes as long for a small chunk as for a full one, so the chunks are
// balanced (k equal chunks, not k - 1 full ones and a remainder), and a batch that fits one launch is not split at all.
int lep_batch_plan(const size_t* file_bytes, const size_t* frame_bytes, int n, const lep_batch_options* o, int* chunk_first, int cap) {
    if (n < 0 || cap < 2) return -1;
    const size_t chunk_budget = o && o->chunk_frame_bytes ? o->chunk_frame_bytes : ((size_t)32 << 30);
    size_t chunk_images = o && o->chunk_images > 0 ? (size_t)o->chunk_images : 1024;
    const bool auto_chunks = !(o && o->chunk_images > 0) && !(o && o->host_huffman);
    size_t cap_segments = 8192;
    if (const char* e = getenv("LEP_BATCH_CHUNK_SEGMENTS")) cap_segments = (size_t)std::min(8192, std::max(64, atoi(e)));
    size_t chunk_segments = auto_chunks ? cap_segments : (size_t)1 << 30;
    // thread segments a file will get, from its size (write_ujpg's rule on the scan size, jpgcoder.cc:3856-3871; the file
    // size over-estimates the scan a little, which only makes a chunk slightly smaller)
    auto segments_guess = [&](int i) -> size_t { const size_t b = file_bytes[i]; return b < 125000 ? 1 : b < 250000 ? 2 : b < 500000 ? 4 : 8; };
    if (auto_chunks) {
        size_t live = 0, segs = 0;
        for (int i = 0; i < n; ++i) if (frame_bytes[i]) { ++live; segs += segments_guess(i); }
        if (live <= 1024 && segs <= 8192) chunk_segments = 8192;
        else {
            const size_t k = std::max({(segs + cap_segments - 1) / cap_segments, (live + 1023) / 1024, (size_t)1});
            chunk_segments = std::min<size_t>(cap_segments, ((segs + k - 1) / k + 7) & ~(size_t)7);
            chunk_images = std::min<size_t>(1024, (live + k - 1) / k);
        }
    }
    // The first chunk's upload and scan decode have nothing to hide behind: LEP_BATCH_FIRST_CHUNK_DIV=<d> makes the first chunk of a
    // call of several chunks 1/d of the others (measurement knob; 1 = all chunks alike, the default).
    int first_div = 1;
    if (auto_chunks && chunk_segments < 8192) {
        if (const char* e = getenv("LEP_BATCH_FIRST_CHUNK_DIV")) first_div = std::max(1, atoi(e));
    }
    int nchunks = 0;
    for (int i = 0; i < n;) {
        if (nchunks + 1 >= cap) return -1;
        const size_t div = nchunks == 0 ? (size_t)first_div : 1;
        chunk_first[nchunks++] = i;
        size_t bytes = 0, nsegs = 0, count = 0;
        for (; i < n; ++i) {
            if (!frame_bytes[i]) continue;
            const size_t fb = (frame_bytes[i] + 255) & ~(size_t)255, sg = segments_guess(i);
            if (count && (bytes + fb > chunk_budget || count >= std::max<size_t>(chunk_images / div, 1) || nsegs + sg > std::max<size_t>(chunk_segments / div, 8))) break;
            bytes += fb; nsegs += sg; ++count;
        }
    }
    chunk_first[nchunks] = n;
    return nchunks;
}

// A stream of format-version >= 2 files back to back restores the concatenation of their JPEGs (`cat a.lep b.lep | lepton -`,
// jpgcoder.cc:1868-1897, test_suite/test_concat.sh): after each file the reader looks for another header behind the size
// trailer.  A version-1 file runs to the end of the input, so nothing can follow it.
int lep_decompress(lep_gpu* g, const uint8_t* lepdata, size_t len, lep_bytes* out) {
    if (!g) return LEP_GPU_ERROR;
    std::vector<uint8_t> all;
    std::unique_ptr<lep_file> prev;
    size_t off = 0;
    int files = 0;
    for (;;) {
        lep_file* f = nullptr;
        int rc = lep_file_open_next(lepdata + off, len - off, prev.get(), &f);
        if (rc) return rc;
        std::unique_ptr<lep_file> hold(f);
        lep_image_desc d;
        lep_file_describe(f, &d);
        lep_segment segs[LEP_MAX_SEGMENTS];
        lep_bytes streams[LEP_MAX_SEGMENTS];
        int32_t status[LEP_MAX_SEGMENTS];
        int n = lep_file_segments(f, segs, streams, 0);
        rc = lep_gpu_decode_host(g, &d, 1, segs, n, streams, status);
        if (rc) return rc;
        const size_t used = lep_file_consumed(f);
        const bool more = lep_chained_file_follows(lepdata + off, len - off, used);
        if (!files && !more) return lep_file_recode(f, out);   // the usual case: one file, no copy
        lep_bytes one = {nullptr, 0, 0};
        rc = lep_file_recode(f, &one);
        if (rc) return rc;
        all.insert(all.end(), one.data, one.data + one.len);
        lep_free(one.data);
        ++files;
        if (!more) break;
        off += used;
        prev = std::move(hold);
    }
    return to_bytes(all, out);
}

int lep_chained_file_follows(const uint8_t* lepdata, size_t len, size_t consumed) {
    return consumed + 2 <= len && lepdata[consumed] == 0xCF && lepdata[consumed + 1] == 0x84 ? 1 : 0;
}

}  // extern "C"

// ---- framing pieces -----------------------------------------------------------------------------
extern "C" {

int lep_handoffs_serialize(const lep_handoff* h, int n, uint8_t* out, size_t out_cap) {
    if (n < 0 || n > 255 || out_cap < (size_t)n * 16 + 2) return LEP_BUFFER_TOO_SMALL;
    std::vector<lep::Handoff> v(n);
    for (int i = 0; i < n; ++i) {
        v[i].luma_y_start = h[i].luma_y_start; v[i].luma_y_end = h[i].luma_y_end; v[i].segment_size = h[i].segment_size;
        v[i].overhang_byte = h[i].overhang_byte; v[i].num_overhang_bits = h[i].num_overhang_bits;
        memcpy(v[i].last_dc, h[i].last_dc, sizeof v[i].last_dc);
    }
    std::vector<uint8_t> b = lep::serialize_handoffs(v);
    memcpy(out, b.data(), b.size());
    return (int)b.size();
}

int lep_handoffs_parse(const uint8_t* data, size_t len, lep_handoff* out, int out_cap) {
    std::vector<lep::Handoff> v;
    if (!lep::deserialize_handoffs(data, len, &v)) return -LEP_VERSION_UNSUPPORTED;
    if ((int)v.size() > out_cap) return -LEP_BUFFER_TOO_SMALL;
    for (size_t i = 0; i < v.size(); ++i) {
        out[i].luma_y_start = v[i].luma_y_start; out[i].luma_y_end = v[i].luma_y_end; out[i].segment_size = v[i].segment_size;
        out[i].overhang_byte = v[i].overhang_byte; out[i].num_overhang_bits = v[i].num_overhang_bits;
        memcpy(out[i].last_dc, v[i].last_dc, sizeof v[i].last_dc);
    }
    return (int)v.size();
}

int lep_mux(const lep_bytes* streams, int nstreams, int version, lep_bytes* out) {
    std::vector<std::vector<uint8_t>> st(nstreams);
    for (int i = 0; i < nstreams; ++i) st[i].assign(streams[i].data, streams[i].data + streams[i].len);
    std::vector<uint8_t> o;
    lep::mux_streams(st, version, &o);
    return to_bytes(o, out);
}

int lep_demux(const uint8_t* data, size_t len, lep_bytes* streams16) {
    std::vector<std::vector<uint8_t>> st;
    lep::demux_packets(data, len, 0, &st);
    for (int i = 0; i < 16; ++i)
        if (int rc = to_bytes(st[i], &streams16[i])) return rc;
    return 0;
}

}  // extern "C"
