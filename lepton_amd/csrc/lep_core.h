// lep_core.h -- what every kernel of the hot path shares: the device-side image / segment descriptors, the constant tables, the packed
// Branch word and its update rule, the row schedule of a thread segment.  Plain C++ without the standard library, so that hipcc
// compiles it for gfx950 and tests/emu can compile the kernel headers with g++ (lane-loop emulation).  (The single-lane coder that
// round 1 shipped from this file -- the plainest statement of the block syntax -- is test infrastructure now:
// tests/emu/lep_core_coder.h.)
//
// What it implements (reference file:line):
//   block syntax      src/vp8/encoder/encoder.cc:194-402 / src/vp8/decoder/decoder.cc:167-318
//   edge syntax       src/vp8/encoder/encoder.cc:39-164  / src/vp8/decoder/decoder.cc:27-141
//   contexts          src/vp8/model/model.hh:463-485, 852-871, 1033-1071, 674-784, 823-832, 1072-1122
//   Branch adaptation src/vp8/model/branch.hh:82-100
//   bool coder        src/vp8/encoder/boolwriter.hh:48-118, boolwriter.cc:17-35,
//                     src/vp8/decoder/boolreader.hh:184-258, 376-416, boolreader.cc:25-34
//   integer IDCT      src/lepton/idct.cc:35-161
//   neighbour summary src/vp8/util/block_context.hh:44-78
//   row schedule      src/lepton/lepton_codec.hh:41-100, src/lepton/vp8_encoder.cc:83-154, 239-445
#pragma once
#include <stdint.h>

#ifndef LEP_DEV
#define LEP_DEV inline
#endif

namespace lepdev {

// ---- model layout in HBM -----------------------------------------------------------------------
// One 32-bit word per Branch: false_count | true_count << 8 | probability << 16, so that a bin is one
// dword load and one dword store.  The in-memory layout of the model is not part of the .lep format
// (SURVEY.md App. B), so tables are ordered hot-first and the 1.5 MB threshold table last.
enum : uint32_t {
    kNz7x7 = 0,                             // [2][26][6][32]
    kNz1x8 = kNz7x7 + 2 * 26 * 6 * 32,      // [2][8][8][3][4]
    kNz8x1 = kNz1x8 + 2 * 8 * 8 * 3 * 4,    // [2][8][8][3][4]
    kSign = kNz8x1 + 2 * 8 * 8 * 3 * 4,     // [2][4][12]
    kExpDc = kSign + 2 * 4 * 12,            // [12][17][11]
    kResDc = kExpDc + 12 * 17 * 11,         // [12][10]
    kRes = kResDc + 12 * 10,                // [2][64][10][10]
    kExpX = kRes + 2 * 64 * 10 * 10,        // [2][10][15][12][11]
    kExp7 = kExpX + 2 * 10 * 15 * 12 * 11,  // [2][10][49][12][11]
    kThresh = kExp7 + 2 * 10 * 49 * 12 * 11,  // [2][256][8][128]
    kModelBranches = kThresh + 2 * 256 * 8 * 128
};
static_assert(kModelBranches == 721564, "Model has 721,564 branches (SURVEY.md 8a11)");
constexpr uint32_t kBranchInit = 1u | (1u << 8) | (128u << 16);

struct NSum {            // NeighborSummary: right column / bottom row predicted pixels + 7x7 nonzero count
    int16_t vert[8];
    int16_t horiz[8];
    int32_t nz;
};

struct ImageDev {        // one image, device-visible
    int32_t ncomp, mcu_rows;
    int32_t width[3], height[3], coded_blocks[3], coded_height[3];
    int16_t* blocks[3];
    uint16_t q[3][64];       // raster order
    int32_t icos_x[3][64];   // model.hh:254
    int32_t icos_y[3][64];   // model.hh:255
    uint8_t min_thresh[3][64];
    int32_t ns_offset[3];    // offset (in NSum) of each component's 2-row ring inside a segment's NSum area
    int32_t ns_total;
};

struct SegDev {
    int32_t image, y0, y1, is_last;
    uint64_t stream_off;     // into the stream arena
    uint32_t stream_cap;
    uint32_t slot;           // index of this segment in the caller's arrays (status / stream_len / bins): the launch order may differ
};

// ---- constant tables -----------------------------------------------------------------------------
#ifdef __HIP_DEVICE_COMPILE__
#define LEP_TABLE __constant__ static const
#else
#define LEP_TABLE static const
#endif
LEP_TABLE uint8_t kA2R[64] = {
    9, 10, 17, 25, 18, 11, 12, 19, 26, 33, 41, 34, 27, 20, 13, 14, 21, 28, 35, 42, 49, 57, 50, 43, 36,
    29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    0, 1, 2, 3, 4, 5, 6, 7, 8, 16, 24, 32, 40, 48, 56};
LEP_TABLE uint8_t kR2A[64] = {
    49, 50, 51, 52, 53, 54, 55, 56, 57, 0, 1, 5, 6, 14, 15, 27, 58, 2, 4, 7, 13, 16, 26, 28,
    59, 3, 8, 12, 17, 25, 29, 38, 60, 9, 11, 18, 24, 30, 37, 39, 61, 10, 19, 23, 31, 36, 40, 45,
    62, 20, 22, 32, 35, 41, 44, 46, 63, 21, 33, 34, 42, 43, 47, 48};
LEP_TABLE uint8_t kNzBin[50] = {0, 1, 2, 3, 4, 4, 5, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 7, 7, 7, 7, 8, 8, 8, 8,
                                8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9};

LEP_DEV int bitlen(uint32_t v) { return v ? 32 - __builtin_clz(v) : 0; }
LEP_DEV int iabs(int v) { return v < 0 ? -v : v; }
LEP_DEV int imin(int a, int b) { return a < b ? a : b; }
LEP_DEV int imax(int a, int b) { return a > b ? a : b; }

// branch.hh:82-100 on the packed word
LEP_DEV uint32_t branch_update(uint32_t w, int obs) {
    uint32_t f = w & 255, t = (w >> 8) & 255, p;
    uint32_t mine = obs ? t : f, other = obs ? f : t;
    if (mine == 255) {
        if (other == 1) return (w & 0xffff) | ((obs ? 0u : 255u) << 16);
        f = (1 + f) >> 1; t = (1 + t) >> 1;
        if (obs) t = 129; else f = 129;
        p = (f << 8) / (f + t);
    } else {
        if (obs) ++t; else ++f;
        p = (f << 8) / (f + t);
    }
    return f | (t << 8) | (p << 16);
}

// ---- the row schedule of a thread segment (lepton_codec.hh:41-100, vp8_encoder.cc:83-154): row `idx` of the interleaved walk over the
// components' block rows -- which component, which block row, the luma row it belongs to, whether it is coded at all
struct RowSpec { int component, curr_y, luma_y; bool skip, done; };
LEP_DEV RowSpec row_spec(const ImageDev* img, uint32_t idx) {
    uint32_t mult[3] = {0, 0, 0}, total = 0;
    for (int i = 0; i < 3 && i < img->ncomp; ++i) { mult[i] = (uint32_t)img->height[i] / (uint32_t)img->mcu_rows; total += mult[i]; }
    uint32_t mcu_row = idx / total, place = idx - mcu_row * total;
    RowSpec r = {3, 0, (int)(mcu_row * mult[0]), false, false};
    for (int i = 2; i >= 0; --i) {
        if (place < mult[i]) {
            r.component = i;
            r.curr_y = (int)(mcu_row * mult[i] + place);
            if (r.curr_y >= img->coded_height[i]) {
                r.skip = true; r.done = true;
                for (int j = 0; j < 2; ++j)
                    if ((int)(mcu_row * mult[j]) < (j < img->ncomp ? img->coded_height[j] : 0)) r.done = false;
            }
            if (i == 0) r.luma_y = r.curr_y;
            return r;
        }
        place -= mult[i];
    }
    r.skip = true; r.done = true;
    return r;
}

}  // namespace lepdev
