// lep_container.h -- .lep container read/write (host side); see lep_container.cc for citations.
#pragma once
#include <cstdint>
#include <vector>
#include "jpeg_model.h"

struct lep_huff_end;   // include/lepton_mi355x.h: end state of a GPU-coded scan segment

namespace lep {

struct EncodeOptions {
    unsigned max_threads = 8;    // -maxencodethreads (jpgcoder.cc:1081)
    unsigned min_threads = 1;    // -minencodethreads (jpgcoder.cc:1088)
    bool even_split = false;     // -evensplit        (jpgcoder.cc:1064)
    bool allow_progressive = true;   // build default -DDEFAULT_ALLOW_PROGRESSIVE (CMakeLists.txt:343)
    int format_version = 1;      // 2 = `lepton -brotliheader` (jpgcoder.cc:1116-1119): brotli header, packet end marker
};
bool brotli_encoder_available();   // the vendored brotli 1.0.0 encoder was compiled into this library (build())

struct LepFile {
    int version = 0;
    uint8_t flag = 0;            // 'Z' baseline, 'X' progressive, 'Y' partial
    int nthreads = 0;
    uint32_t jpeg_size = 0;
    JpegFile jpeg;               // hdr / padbit / rst info / garbage / truncation filled from the header
    std::vector<Handoff> segs;
    bool rst_cnt_set = false;
    bool garbage_default_eoi = true;
    bool has_prefix = false, embedded = false;
    std::vector<uint8_t> prefix_garbage;
    std::vector<std::vector<uint8_t>> streams;   // de-multiplexed, index = stream id = segment index
    size_t consumed = 0;                         // bytes of the input this file occupies (less than the input for a chained stream of v2+ files)
    bool unbound_stream_packet = false;          // general re-coder: a packet for a stream id no hand-off created (vp8_decoder.cc:236)
    bool header_pending = false;                 // a "CNT" section was met: the next file of the stream reads pending_header, empty or not
    std::vector<uint8_t> pending_header;         // header bytes behind a "CNT" section: they belong to the next file of the stream
};

std::vector<Handoff> plan_segments(const JpegFile& jf, const EncodeOptions& opt);
std::vector<uint8_t> serialize_handoffs(const std::vector<Handoff>& segs);
bool deserialize_handoffs(const uint8_t* d, size_t n, std::vector<Handoff>* out);
void mux_streams(const std::vector<std::vector<uint8_t>>& streams, int version, std::vector<uint8_t>* out);
// version 1: zlib header (vendored zlib 1.2.8 == system zlib at level 9: cmp-equal files); version 2: brotli header --
// byte-equal to the reference's only with ITS encoder (dependencies/brotli 1.0.0, compiled in from where it lies when
// build() finds it; the system's 1.0.9 writes other bytes): VERSION_UNSUPPORTED without it
int write_lep(const JpegFile& jf, const std::vector<Handoff>& segs, const std::vector<std::vector<uint8_t>>& streams,
              std::vector<uint8_t>* out, int format_version = 1);
int parse_lep(const uint8_t* d, size_t n, LepFile* lf, const std::vector<uint8_t>* carried = nullptr);
bool worker_bounds_exceed_arena(const LepFile& lf, size_t file_bytes);   // lep_container.cc
int baseline_header_pass(LepFile* lf);                                   // jpeg_recode.cc: what recode_baseline_jpeg checks before it decodes a row
size_t demux_packets(const uint8_t* d, size_t n, size_t at, std::vector<std::vector<uint8_t>>* streams, bool* saw_eof = nullptr);
bool brotli_available();

// decode side (jpeg_recode.cc): coefficients -> JPEG bytes
int recode_jpeg(LepFile* lf, std::vector<uint8_t>* out);

// The same in three steps, with the Huffman coding of the segments done elsewhere (the GPU encoder, lep_huff.h).
// RecodeImage / RecodeSegment are laid out exactly like lephuff::HuffImage / HuffSegment and lep_huff_image / lep_huff_segment.
struct RecodeImage {
    int32_t ncomp, mcuh, mcuv, mcuc;
    int32_t rsti, padbit;
    uint32_t rst_limit;
    int32_t interleaved;
    int32_t hs[4], vs[4], bch[4];
    int32_t dc_tbl[4], ac_tbl[4];
    int32_t scan_cmp[4];
    int32_t trunc_bc[4];
    const int16_t* blocks[4];
    uint32_t code[4][256];
};
struct RecodeSegment {
    int32_t image, mcu_row0, mcu_row1;
    uint32_t overhang;
    int16_t last_dc[4];
    uint64_t out_off;
    uint32_t out_cap;
    uint32_t pad;
};
struct RecodePlan {
    std::vector<uint8_t> head;      // everything in front of the scan (prefix garbage, SOI, header up to the first SOS)
    size_t hdr_pos = 0;             // header bytes consumed by `head`
    size_t scan_bound = 0;          // bytes the file may hold before the trailing garbage
    bool gpu_ok = false;            // eligible for the GPU Huffman encoder; image / segs are filled
    RecodeImage image;
    std::vector<RecodeSegment> segs;
};
int recode_prepare(LepFile* lf, RecodePlan* plan);

// Progressive files with the Huffman coding of the scans done elsewhere (the GPU, lep_huffprog.h): _prepare walks the header
// (one entry per SOS: band, successive approximation, tables) and decides eligibility; _finish glues header pieces, the scans'
// bytes (already stuffed, restart markers in place), misplaced restart markers and garbage together.  ProgImage / ProgScan are
// laid out exactly like lephuff::ProgImage / ProgScan and the C ABI's lep_huffprog_image / lep_huffprog_scan.
struct ProgImage {
    int32_t ncomp, mcuh, mcuv, mcuc;
    int32_t rsti, padbit;
    int32_t hs[4], vs[4], bch[4], bcv[4], nch[4], ncv[4], mbs[4];
    const int16_t* blocks[4];
};
struct ProgScan {
    int32_t image;
    int32_t cmpc, cmp[4];
    int32_t from, to, sah, sal;
    int32_t max_eobrun;
    int32_t tbl[4];
    int32_t rsti;                         // this scan's restart interval (-1: the image's)
    uint64_t out_off;
    uint32_t out_cap;
    uint32_t corr_off;
    uint32_t corr_cap;
    uint32_t file_bound;                  // bytes all scans of the file produce together at most (0 = unknown)
    uint32_t code[4][256];                // (scans of sequential frames use all four: DC tables 0 / 1, AC tables 0 / 1)
};
struct ProgPlan {
    bool gpu_ok = false;
    ProgImage image;
    std::vector<ProgScan> scans;          // out_cap / corr_cap = what the scan may need; offsets are the caller's
    std::vector<size_t> scan_hdr_end;     // header position behind each SOS
    std::vector<uint32_t> markers;        // restart markers each scan writes itself
};
int recode_progressive_prepare(LepFile* lf, ProgPlan* plan);
int progressive_plan(JpegFile* jf, size_t jpeg_size, bool rst_cnt_set, ProgPlan* plan);
int recode_progressive_finish(LepFile* lf, const ProgPlan& plan, const std::vector<std::pair<const uint8_t*, size_t>>& scan_bytes,
                              std::vector<uint8_t>* out);
int recode_finish(LepFile* lf, const RecodePlan& plan, const std::vector<std::pair<const uint8_t*, size_t>>& seg_bytes, const lep_huff_end* ends,
                  std::vector<uint8_t>* out);

}  // namespace lep
