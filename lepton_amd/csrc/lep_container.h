// lep_container.h -- .lep container read/write (host side); see lep_container.cc for citations.
#pragma once
#include <cstdint>
#include <vector>
#include "jpeg_model.h"

namespace lep {

struct EncodeOptions {
    unsigned max_threads = 8;    // -maxencodethreads (jpgcoder.cc:1081)
    unsigned min_threads = 1;    // -minencodethreads (jpgcoder.cc:1088)
    bool even_split = false;     // -evensplit        (jpgcoder.cc:1064)
    bool allow_progressive = true;   // build default -DDEFAULT_ALLOW_PROGRESSIVE (CMakeLists.txt:343)
};

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
};

std::vector<Handoff> plan_segments(const JpegFile& jf, const EncodeOptions& opt);
std::vector<uint8_t> serialize_handoffs(const std::vector<Handoff>& segs);
bool deserialize_handoffs(const uint8_t* d, size_t n, std::vector<Handoff>* out);
void mux_streams(const std::vector<std::vector<uint8_t>>& streams, int version, std::vector<uint8_t>* out);
int write_lep(const JpegFile& jf, const std::vector<Handoff>& segs, const std::vector<std::vector<uint8_t>>& streams,
              std::vector<uint8_t>* out);
int parse_lep(const uint8_t* d, size_t n, LepFile* lf);
void demux_packets(const uint8_t* d, size_t n, size_t at, std::vector<std::vector<uint8_t>>* streams);

// decode side (jpeg_recode.cc): coefficients -> JPEG bytes
int recode_jpeg(LepFile* lf, std::vector<uint8_t>* out);

}  // namespace lep
