// core_emu.cc -- TEST ONLY: compiles the kernel source lepton_amd/csrc/lep_core.h with g++ and runs one
// segment on the CPU exactly as lane 0 of the wavefront would, so kernel logic can be single-stepped
// and diffed against the oracle without a GPU.  Never linked into the product.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#define LEP_DEV inline
#include "../../lepton_amd/csrc/lep_derive.h"

using namespace lepdev;

template <bool DEC>
static int run(const lep_image_desc* d, int y0, int y1, int is_last, uint8_t* stream, uint32_t* len, uint32_t cap, uint32_t* bins) {
    ImageDev img;
    int rc = derive_image(*d, &img, !DEC);
    if (rc) return rc;
    std::vector<uint32_t> model(kModelBranches, kBranchInit);
    std::vector<NSum> ns(img.ns_total);
    memset(ns.data(), 0, ns.size() * sizeof(NSum));
    SegDev seg;
    seg.image = 0; seg.y0 = y0; seg.y1 = y1; seg.is_last = is_last; seg.stream_off = 0; seg.stream_cap = cap;
    SegmentCoder<DEC> sc;
    if constexpr (DEC) sc.bc.init_stream(stream, *len); else sc.bc.init_stream(stream, cap);
    rc = sc.run(&img, seg, model.data(), ns.data());
    if constexpr (!DEC) { *len = sc.bc.finish(); if (sc.bc.overflow) rc = LEP_BUFFER_TOO_SMALL; }
    if (bins) *bins = sc.nbins;
    return rc;
}

extern "C" int emu_encode_segment(const lep_image_desc* d, int y0, int y1, int is_last, uint8_t* out, uint32_t cap, uint32_t* len, uint32_t* bins) {
    return run<false>(d, y0, y1, is_last, out, len, cap, bins);
}
extern "C" int emu_decode_segment(const lep_image_desc* d, int y0, int y1, int is_last, const uint8_t* in, uint32_t len, uint32_t* bins) {
    return run<true>(d, y0, y1, is_last, const_cast<uint8_t*>(in), &len, 0, bins);
}
