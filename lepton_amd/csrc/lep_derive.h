// lep_derive.h -- host-side preparation of the per-image constants the kernels read
// (quantisation-derived tables).  Shared by lep_gpu.hip and the CPU single-step test of lep_core.h.
#pragma once
#include <cstring>
#include "../../include/lepton_mi355x.h"
#include "lep_core.h"

namespace lepdev {

static const uint8_t kR2Z_host[64] = {0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43,
                                      9, 11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60,
                                      21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
static const int32_t kIcosCol0[8] = {8192, 11363, 10703, 9633, 8192, 6436, 4433, 2260};
static const uint16_t kFreqMax[64] = {1024, 931, 985, 968, 1020, 968, 1020, 1020, 932, 858, 884, 840, 932, 838, 854, 854,
                                      985, 884, 871, 875, 985, 878, 871, 854, 967, 841, 876, 844, 967, 886, 870, 837,
                                      1020, 932, 985, 967, 1020, 969, 1020, 1020, 969, 838, 878, 886, 969, 838, 969, 838,
                                      1020, 854, 871, 870, 1010, 969, 1020, 1020, 1020, 854, 854, 838, 1020, 838, 1020, 838};

// ProbabilityTablesBase::set_quantization_table (src/vp8/model/model.hh:247-290), per image instead of process-global
inline int derive_image(const lep_image_desc& d, ImageDev* o, bool encoding) {
    memset(o, 0, sizeof *o);
    if (d.ncomp < 1 || d.ncomp > 3 || d.mcu_rows < 1) return LEP_UNSUPPORTED_JPEG;
    o->ncomp = d.ncomp; o->mcu_rows = d.mcu_rows;
    int ns = 0;
    for (int c = 0; c < d.ncomp; ++c) {
        o->width[c] = d.width_blocks[c]; o->height[c] = d.height_blocks[c];
        o->coded_blocks[c] = d.coded_blocks[c]; o->coded_height[c] = d.coded_height[c];
        o->blocks[c] = d.blocks[c];
        if (o->width[c] < 1 || o->height[c] < d.mcu_rows) return LEP_UNSUPPORTED_JPEG;
        for (int i = 0; i < 64; ++i) o->q[c][i] = d.qtable_zigzag[c][kR2Z_host[i]];
        for (int r = 0; r < 8; ++r)
            for (int i = 0; i < 8; ++i) {
                o->icos_x[c][r * 8 + i] = kIcosCol0[i] * (int32_t)o->q[c][i * 8 + r];
                o->icos_y[c][r * 8 + i] = kIcosCol0[i] * (int32_t)o->q[c][r * 8 + i];
            }
        for (int r = 0; r < 8; ++r)
            if (encoding && (o->icos_x[c][r * 8] == 0 || o->icos_y[c][r * 8] == 0)) return LEP_UNSUPPORTED_JPEG_WITH_ZERO_IDCT_0;
        if (o->q[c][0] == 0) return LEP_UNSUPPORTED_JPEG_WITH_ZERO_IDCT_0;
        for (int i = 0; i < 64; ++i) {
            unsigned fm = kFreqMax[i] + o->q[c][i] - 1;
            if (o->q[c][i]) fm /= o->q[c][i];
            fm &= 0xffff;
            int len = fm ? 32 - __builtin_clz(fm) : 0;
            o->min_thresh[c][i] = (uint8_t)(len > 7 ? len - 7 : 0);
        }
        o->ns_offset[c] = ns;
        ns += 2 * o->width[c];
    }
    o->ns_total = ns;
    return 0;
}


}  // namespace lepdev
