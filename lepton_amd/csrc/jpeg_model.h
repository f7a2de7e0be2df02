// jpeg_model.h -- host-side data model for one JPEG file split the way the reference splits it
// (src/lepton/jpgcoder.cc:2269-2466 read_jpeg, :2799-3302 decode_jpeg): marker segments, the
// un-stuffed entropy-coded bytes, trailing garbage, and the quantised DCT coefficients in the
// reference's "aligned" block order (src/vp8/util/aligned_block.hh:32-44).
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace lep {

// exit codes = src/vp8/util/memory.hh:13-40 (the process exit status is the reference's error API)
enum ExitCode : int {
    EX_SUCCESS = 0, EX_ASSERTION_FAILURE = 1, EX_CODING_ERROR = 2, EX_SHORT_READ = 3,
    EX_UNSUPPORTED_4_COLORS = 4, EX_THREAD_PROTOCOL_ERROR = 5, EX_COEFFICIENT_OUT_OF_RANGE = 6,
    EX_STREAM_INCONSISTENT = 7, EX_PROGRESSIVE_UNSUPPORTED = 8, EX_FILE_NOT_FOUND = 9,
    EX_SAMPLING_BEYOND_TWO_UNSUPPORTED = 10, EX_SAMPLING_BEYOND_FOUR_UNSUPPORTED = 11,
    EX_THREADING_PARTIAL_MCU = 12, EX_VERSION_UNSUPPORTED = 13, EX_ONLY_GARBAGE_NO_JPEG = 14,
    EX_OS_ERROR = 33, EX_HEADER_TOO_LARGE = 34, EX_BLOCK_OFFSET_OOM = 37,
    EX_TOO_MUCH_MEMORY_NEEDED = 38, EX_ROUNDTRIP_FAILURE = 41, EX_UNSUPPORTED_JPEG = 42, EX_UNSUPPORTED_JPEG_WITH_ZERO_IDCT_0 = 43,
    EX_INVALID_RESET_MARKER_FOUND = 40, EX_UNSUPPORTED_4_COLORS_B = 4,
    EX_GPU_PATH_DECLINED = 101,   // ours: a GPU scan coder left a file to the host coder (recode_finish on a truncated file): not an error of the file
    EX_GPU_ERROR = 120,   // ours: HIP runtime failure (no reference equivalent)
};

// aligned index -> raster / zig-zag maps (aligned_block.hh:32-76)
extern const uint8_t kAlignedToRaster[64];
extern const uint8_t kZigzagToAligned[64];
extern const uint8_t kZigzagToRaster[64];

struct Component {
    int jid = 0;      // component id from SOF
    int hs = 0;       // horizontal sampling factor (reference name: sfv)
    int vs = 0;       // vertical sampling factor   (reference name: sfh)
    int qidx = 0;     // quantisation table index
    int dc_tbl = 0, ac_tbl = 0;
    int bch = 0, bcv = 0, bc = 0;   // blocks per row / column / total (MCU padded)
    int nch = 0, ncv = 0;           // non-interleaved dimensions
    int mbs = 0;                    // blocks per MCU
};

// src/lepton/thread_handoff.hh:8-39
struct Handoff {
    uint16_t luma_y_start = 0, luma_y_end = 0;
    uint32_t segment_size = 0;
    uint8_t overhang_byte = 0, num_overhang_bits = 0;
    int16_t last_dc[4] = {0, 0, 0, 0};
};

constexpr uint64_t kMaxFrameBlocks = 4423680;   // (576 MiB - 36 MiB) / 128 B, the reference's default frame budget

struct HuffTable {
    bool set = false;
    // zero-filled until a DHT segment defines the table: the reference's tables are zeroed globals, and a file that codes with
    // a table it never defined (a progressive header forced through the baseline re-coder by a hostile flag byte) is written
    // with zero-length codes there -- found by the structured differential fuzz as a run-to-run difference
    uint16_t clen[256] = {};
    uint16_t cval[256] = {};
    int max_eobrun = 0;
    // decode side: the code words a bit stream can actually end on, as disjoint intervals of 16-bit patterns sorted by their
    // first pattern -- word k covers wfirst[k] .. wfirst[k] + (1 << (16 - wlen[k])) - 1.  For a DHT that follows Annex C these
    // are its codes; for one that does not (codes that extend other codes, more inner nodes than the reference's 256-entry
    // tree holds) they are what the reference's tree would let a walk reach (build_huff_table says how)
    uint16_t wfirst[256] = {};
    uint8_t wlen[256] = {}, wsym[256] = {};
    int nwords = 0;
    // first-level lookup over the same words: index = the next 10 bits, entry = code length << 8 | symbol, 0 = longer than 10
    // bits or no such code
    uint16_t lut[1024] = {};
    // the word that begins with these 16 bits, or -1
    int word_at(unsigned pattern16) const {
        int lo = 0, hi = nwords;                                  // last word whose first pattern is <= pattern16
        while (lo < hi) { int mid = (lo + hi) >> 1; if (wfirst[mid] <= pattern16) lo = mid + 1; else hi = mid; }
        return lo > 0 && pattern16 - wfirst[lo - 1] < (1u << (16 - wlen[lo - 1])) ? lo - 1 : -1;
    }
};

struct JpegFile {
    // --- container split
    std::vector<uint8_t> hdr;       // every marker segment after SOI (hdrdata)
    std::vector<uint8_t> scan;      // entropy-coded bytes, FF00 un-stuffed, RSTn removed (huffdata)
    std::vector<uint8_t> garbage;   // bytes from EOI on, empty if exactly FF D9 (grbgdata)
    std::vector<std::pair<uint32_t, uint32_t>> scan_to_file;   // huff_input_offsets
    std::vector<uint32_t> scan_start;   // offset in `scan` where the entropy-coded bytes behind each SOS begin
    std::vector<std::pair<uint32_t, uint32_t>> scan_file_range;   // the same scans in the FILE: [first entropy-coded byte, the marker that ends them)
    std::vector<uint32_t> rst_cnt;  // RST markers seen per scan
    std::vector<uint32_t> rst_pos;  // first scan: offset (in `scan`, the un-stuffed bytes) at which each of its restart markers stood
    std::vector<uint8_t> rst_err;   // wrongly placed RST markers at scan end, per scan
    bool early_eof = false;
    int padbit = -1;
    // -startbyte=<n> (jpgcoder.cc:364, 1132-1133, 3801-3843): the .lep restores only the bytes from start_byte on; set before
    // parse_jpeg.  prefix_garbage = the raw bytes from start_byte up to the first MCU row that starts at or after it.
    uint32_t start_byte = 0;
    std::vector<uint8_t> prefix_garbage;
    // -embedding=<n> (jpgcoder.cc:365, 1135-1137, 2275-2282): the JPEG sits n bytes into a larger blob; those n bytes are kept as
    // prefix garbage ('PGE' section), what follows EOI as ordinary garbage, and the .lep restores the whole blob
    bool embedded = false;
    uint32_t file_size = 0;
    // --- frame
    int width = 0, height = 0, ncomp = 0;
    int jpegtype = 0;               // 1 sequential, 2 progressive
    Component comp[4];
    uint16_t qtables[4][64];        // zig-zag order, as stored
    int hmax = 0, vmax = 0;
    int mcuh = 0, mcuv = 0, mcuc = 0;   // MCUs per row / per column / total
    int rsti = 0;
    HuffTable htab[2][4];
    bool progressive_needed = false;    // not a single interleaved sequential scan
    // --- scan state (current SOS)
    int cs_cmpc = 0, cs_cmp[4] = {0, 0, 0, 0}, cs_from = 0, cs_to = 0, cs_sah = 0, cs_sal = 0;
    int scan_count = 0;
    // --- coefficients
    std::vector<int16_t> coef[4];       // bc * 64, aligned order (owned storage; unused when the frame lives in ext_mem)
    int16_t* plane[4] = {nullptr, nullptr, nullptr, nullptr};   // where the frame is: coef[c].data() or inside ext_mem
    int16_t* ext_mem = nullptr;         // optional caller-provided frame storage (e.g. pinned staging memory), planes back to back
    size_t ext_cap = 0;                 // bytes
    // points plane[] at zeroed storage for the frame geometry in comp[]; ext_mem is used when it is large enough
    void place_frame(bool zero) {
        size_t need = 0;
        for (int c = 0; c < ncomp; ++c) need += (size_t)comp[c].bc * 128;
        if (ext_mem && need <= ext_cap) {
            size_t off = 0;
            for (int c = 0; c < ncomp; ++c) { plane[c] = ext_mem + off / 2; off += (size_t)comp[c].bc * 128; }
            if (zero) __builtin_memset(ext_mem, 0, need);
        } else {
            for (int c = 0; c < ncomp; ++c) { coef[c].assign((size_t)comp[c].bc * 64, 0); plane[c] = coef[c].data(); }
        }
    }
    std::vector<Handoff> rows;          // one per MCU row + final
    // truncation
    int max_cmp = 0, max_bpos = 0, max_sah = 0, max_dpos[4] = {0, 0, 0, 0};
    int trunc_bcv[4] = {0, 0, 0, 0}, trunc_bc[4] = {0, 0, 0, 0};
    int warn = 0;                       // reference errorlevel 1
    std::string error;
};

// Parses `data` (a whole JPEG file). Returns an ExitCode (0 = ok). allow_progressive mirrors
// -allowprogressive (jpgcoder.cc:1090) -- without it, non-baseline files return
// EX_PROGRESSIVE_UNSUPPORTED like the reference does.
int parse_jpeg(const uint8_t* data, size_t size, bool allow_progressive, JpegFile* out);

// GPU Huffman scan decode (lep_huffdec.h): host-side halves.  ScanDecodePlan / ScanDecodeRow are laid out exactly like
// lephuff::HuffDecImage / HuffDecRow and the C ABI's lep_huffdec_image / lep_huffdec_row.
constexpr int32_t kScanEarlyEof = 1;               // ScanDecodePlan::flags (== LEP_HUFFDEC_EARLY_EOF)
constexpr int32_t kScanRstTable = 2;               // ... (== LEP_HUFFDEC_RST_TABLE): the restart positions follow the scan bytes (lep_jpeg_scan_restarts)
constexpr int32_t kScanRowTruncated = 0x40000000;  // final ScanDecodeRow::aux (== LEP_HUFFDEC_ROW_TRUNCATED)
struct ScanDecodePlan {
    const uint8_t* scan;
    uint32_t scan_len;
    int32_t ncomp, mcuh, mcuv, mcuc, rsti;
    int32_t flags, reserved0;              /* flags: kScanEarlyEof */
    int32_t hs[4], vs[4], bch[4], dc_tbl[4], ac_tbl[4], scan_cmp[4];
    int16_t* blocks[4];
    uint64_t rows_off;
    uint16_t lut[4][512];
    int32_t maxcode[4][8], valoff[4][8];   /* codes of 9..16 bits: largest code per length (-1 none), symbol index - code */
    uint8_t longsym[4][256];               /* their symbols in canonical order */
};
struct ScanDecodeRow {
    uint32_t bitpos;
    int16_t last_dc[4];
    int32_t aux;
};
int parse_jpeg_prepare_gpu(const uint8_t* data, size_t size, JpegFile* jf, ScanDecodePlan* plan, bool* eligible);
int parse_jpeg_finish_gpu(JpegFile* jf, const ScanDecodeRow* rows);

// Progressive files on the GPU scan decoder (lep_huffprogdec.h).  ProgScanDecodePlan is laid out exactly like
// lephuff::ProgDecScan and the C ABI's lep_huffprogdec_scan.
struct ProgScanDecodePlan {
    ScanDecodePlan t;
    int32_t cmpc, cmp[4];
    int32_t from, to, sah, sal;
    int32_t bcv[4], nch[4], ncv[4], mbs[4];
    int32_t tbl[4];
    int32_t max_eobrun;
    int32_t want_rows;
    int32_t level;
    int32_t pad;
    uint64_t result_off;
};
// after parse_jpeg_prepare_gpu said "not eligible" for a progressive file: one plan per scan (t.scan = offset into jf->scan as a
// pointer-sized integer, t.scan_len, t.rows_off / result_off relative to the file's first record; rows_needed records in all)
int parse_jpeg_prepare_gpu_progressive(JpegFile* jf, std::vector<ProgScanDecodePlan>* scans, int* rows_needed, bool* eligible);
int parse_jpeg_finish_gpu_progressive(JpegFile* jf, const std::vector<ProgScanDecodePlan>& scans, const ScanDecodeRow* rows);
uint64_t prog_wait_timeouts();   // scans that ended with status 4 (gave up waiting) since the process started

bool build_huff_table(const uint8_t* clen, size_t clen_avail, const uint8_t* cval, size_t cval_avail,
                      HuffTable* t, bool strict);
bool parse_segment(JpegFile* jf, uint8_t type, unsigned len, unsigned avail, const uint8_t* seg, bool strict);
bool setup_frame(JpegFile* jf);

}  // namespace lep
