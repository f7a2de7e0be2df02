// core_emu.cc -- TEST ONLY: compiles the kernel headers under lepton_amd/csrc with g++ as lane-loop emulations (lep_wave.h) -- and the
// single-lane coder of round 1 (lep_core_coder.h) -- so that kernel logic can be single-stepped and diffed against the oracle
// without a GPU.  Never linked into the product.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#define LEP_DEV inline
#include "../../lepton_amd/csrc/lep_derive.h"
#include "lep_core_coder.h"   // the single-lane coder of round 1: test infrastructure

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


// The v3 / v4 decoders read the stream as ALIGNED dwords (the dword that holds the first and the one that holds the last byte
// are loaded whole, the bytes outside the stream masked off) -- on the device every stream sits in an arena that is padded to
// 256 bytes.  Give the emulation the same guarantee (and keep the caller's alignment phase), so that an address sanitizer
// run of these tests only reports real out-of-bounds accesses.
struct PaddedStream {
    std::vector<uint8_t> arena;
    const uint8_t* p;
    PaddedStream(const uint8_t* in, uint32_t len) : arena((size_t)len + 24, 0) {
        uint8_t* base = arena.data() + 8;
        base -= (uintptr_t)base & 3;                 // 4-byte aligned, at least 4 bytes into the arena
        uint8_t* q = base + ((uintptr_t)in & 3);     // same phase as the caller's pointer
        if (len) memcpy(q, in, len);
        p = q;
    }
};

// v3 encoder (lep_enc3.h) as a 64-lane loop emulation
#include "../../lepton_amd/csrc/lep_enc3.h"
extern "C" int emu_encode_segment_v3(const lep_image_desc* d, int y0, int y1, int is_last, uint8_t* out, uint32_t cap, uint32_t* len, uint32_t* bins) {
    ImageDev img;
    int rc = derive_image(*d, &img, true);
    if (rc) return rc;
    std::vector<uint32_t> model(kModelBranches, kBranchInit);
    std::vector<NSum> ns(img.ns_total);
    memset(ns.data(), 0, ns.size() * sizeof(NSum));
    SegDev seg;
    seg.image = 0; seg.y0 = y0; seg.y1 = y1; seg.is_last = is_last; seg.stream_off = 0; seg.stream_cap = cap;
    static lep3::Enc3Shared sh;
    static lep3::Enc3Pipe pipe;   // LEP_EMU_ENC_PIPE=1: the producer half of the two-wave kernel; its hand-over runs the consumer's step in place
    lep3::Enc3Wave w;
    rc = w.run(&img, seg, model.data(), ns.data(), &sh, out, cap, getenv("LEP_EMU_ENC_PIPE") ? &pipe : nullptr);
    if (rc) return rc;
    *len = w.bc.finish();
    if (w.bc.overflow) return LEP_BUFFER_TOO_SMALL;
    if (bins) *bins = w.nbins;
    return 0;
}

// v4 decoder (lep_dec4.h) as a 64-lane loop emulation
#include "../../lepton_amd/csrc/lep_dec4.h"
extern "C" int emu_decode_segment_v4(const lep_image_desc* d, int y0, int y1, int is_last, const uint8_t* in, uint32_t len, uint32_t* bins) {
    ImageDev img;
    int rc = derive_image(*d, &img, false);
    if (rc) return rc;
    std::vector<lep3::U4> model(lep3::kModelWords / 4, lep3::U4{kBranchInit, kBranchInit, kBranchInit, kBranchInit});
    std::vector<NSum> ns(img.ns_total);
    memset(ns.data(), 0, ns.size() * sizeof(NSum));
    SegDev seg;
    seg.image = 0; seg.y0 = y0; seg.y1 = y1; seg.is_last = is_last; seg.stream_off = 0; seg.stream_cap = 0;
    static lep4::Dec4Shared sh;
    lep4::Dec4Wave w;
    PaddedStream ps(in, len);
    rc = w.run(&img, seg, reinterpret_cast<uint32_t*>(model.data()), ns.data(), &sh, ps.p, len);
    if (bins) *bins = w.nbins;
    return rc;
}


// exhaustive check of the 24-bit table reciprocal used by lep4::bupd_t / bupd_u against Branch::record_obs_and_update
extern "C" int emu_check_inv24_update() {
    static uint32_t inv[512];
    for (uint32_t d = 0; d < 512; ++d) inv[d] = lep4::inv24_of(d);
    for (uint32_t f = 1; f < 256; ++f)
        for (uint32_t t = 1; t < 256; ++t) {
            if (inv[f + t] >= (1u << 24)) return 2;
            if ((lep4::mul24(f << 8, inv[f + t]) >> 24) != (f << 8) / (f + t)) return 3;
            for (uint32_t p = 0; p < 256; p += 51)
                for (uint32_t obs = 0; obs < 2; ++obs) {
                    uint32_t w = f | (t << 8) | (p << 16);
                    if (branch_update(w, (int)obs) != lep4::bupd_t(w, obs, inv)) return 4;
                    if (branch_update(w, (int)obs) != lep4::bupd_u(w, obs, inv)) return 5;
                }
        }
    return 0;
}

// lep4::div_by against `/`: every divisor the tables can hold (quantiser values 1..65535 as they are and times the eight
// ICOS column constants), numerators around the multiples of the divisor, the extremes and random ones
extern "C" int emu_check_div_by() {
    static const int32_t col0[9] = {1, 8192, 11363, 10703, 9633, 8192, 6436, 4433, 2260};
    uint64_t rnd = 0x9e3779b97f4a7c15ull;
    for (int c = 0; c < 9; ++c)
        for (uint32_t q = 1; q < 65536; ++q) {
            const int64_t d64 = (int64_t)col0[c] * q;
            if (d64 < 2 || d64 > 0x7fffffff) continue;
            const int32_t d = (int32_t)d64;
            const lep4::DivBy m = lep4::DivBy::of((uint32_t)d);
            int64_t ns[40];
            int k = 0;
            for (int64_t v : {(int64_t)0, (int64_t)1, (int64_t)-1, (int64_t)0x7fffffff, -(int64_t)0x80000000ll, (int64_t)d, (int64_t)d - 1, (int64_t)d + 1}) ns[k++] = v;
            for (int i = 0; i < 8; ++i) {
                rnd = rnd * 6364136223846793005ull + 1442695040888963407ull;
                const int64_t mult = (int64_t)((rnd >> 33) % (uint64_t)(0x7fffffff / d + 1));
                ns[k++] = mult * d; ns[k++] = mult * d - 1; ns[k++] = -(mult * d); ns[k++] = -(mult * d) + 1;
            }
            for (int i = 0; i < k; ++i) {
                if (ns[i] > 0x7fffffff || ns[i] < -(int64_t)0x80000000ll) continue;
                const int32_t n = (int32_t)ns[i];
                const int32_t want = (n == INT32_MIN && d == -1) ? 0 : (int32_t)((int64_t)n / d);
                if (lep4::div_by(n, m.mul, m.shift) != want) return 1 + c;
            }
        }
    return 0;
}

// GPU Huffman re-encoder (lep_huff.h) as a 64-lane loop emulation: scan bytes of one thread segment
#include "../../lepton_amd/csrc/lep_huff.h"
extern "C" int emu_huffman_encode_segment(const lep_huff_image* img, const lep_huff_segment* seg, uint8_t* out, uint32_t* len, lep_huff_end* end) {
    static lephuff::HuffShared sh;
    lephuff::HuffWave w;
    lephuff::HuffSegment s;
    memcpy(&s, seg, sizeof s);
    s.out_off = 0; s.image = 0;
    *len = w.run(reinterpret_cast<const lephuff::HuffImage*>(img), s, &sh, out);
    if (end) w.export_end(reinterpret_cast<lephuff::HuffEnd*>(end));
    return 0;
}

// ... with one lane per run of MCUs (lep_huff_simt.h): count, place, code, stuff for ONE segment; returns 1 when that form does not
// take the segment (restart intervals, ...), and then writes nothing
#include "../../lepton_amd/csrc/lep_huff_simt.h"
extern "C" int emu_huffman_encode_segment_simt(const lep_huff_image* img, const lep_huff_segment* seg, uint8_t* out, uint32_t* len, lep_huff_end* end) {
    static lephuff::SimtEncShared sh;
    const lephuff::HuffImage* im = reinterpret_cast<const lephuff::HuffImage*>(img);
    lephuff::HuffSegment s;
    memcpy(&s, seg, sizeof s);
    s.out_off = 0; s.image = 0;
    if (!lephuff::simt_enc_takes(*im, s)) return 1;
    lephuff::SimtEncSeg es;
    memset(&es, 0, sizeof es);
    lephuff::SimtUnitMap map;
    map.set(s.mcu_row0 * im->mcuh, s.mcu_row1 * im->mcuh, im->rsti);
    es.seg = 0; es.first_unit = 0; es.nunits = map.count();
    es.buf_off = 0; es.buf_bytes = (uint32_t)(((size_t)s.out_cap + 64 + 15) & ~(size_t)15);
    es.map_bytes = im->rsti > 0 ? ((es.buf_bytes >> 3) + 15u) & ~15u : 0u;
    std::vector<uint32_t> unit_bits(es.nunits), unit_plain(es.nunits);
    std::vector<uint32_t> scratch(((size_t)es.buf_bytes + es.map_bytes) / 4 + 4, 0u);
    uint8_t* sc = reinterpret_cast<uint8_t*>(scratch.data());
    for (uint32_t f = 0; f < es.nunits; f += 64) lephuff::simt_enc_units<false>(im, &s, &es, &sh, unit_bits.data(), sc, f);
    lephuff::simt_enc_place(im, &s, &es, unit_bits.data(), unit_plain.data());
    for (uint32_t f = 0; f < es.nunits; f += 64) lephuff::simt_enc_units<true>(im, &s, &es, &sh, unit_bits.data(), sc, f);
    uint32_t n = 0;
    lephuff::simt_enc_stuff(im, &s, es, sc, out, &n, reinterpret_cast<lephuff::HuffEnd*>(end));
    *len = n;
    return 0;
}

// progressive scans (lep_huffprog.h): every scan of one image, one emulated wavefront after the other
#include "../../lepton_amd/csrc/lep_huffprog.h"
// (a scan of a SEQUENTIAL frame coded in several scans goes where the launch code sends it: to the sequential scan encoders, as an image with
// one segment -- lep_huffprog.h sequential_scan_segment, lep_gpu.hip lep_gpu_huffman_progressive_encode_device)
static uint32_t emu_sequential_scan(const lephuff::ProgImage* im, const lephuff::ProgScan& sc, uint8_t* out, bool lanes) {
    lephuff::HuffImage hi;
    lephuff::HuffSegment sg;
    lephuff::sequential_scan_segment(*im, sc, 0, &hi, &sg);
    uint32_t n = 0;
    lep_huff_end end;
    memset(&end, 0, sizeof end);
    int rc = 1;
    if (lanes) rc = emu_huffman_encode_segment_simt(reinterpret_cast<const lep_huff_image*>(&hi), reinterpret_cast<const lep_huff_segment*>(&sg), out + sc.out_off, &n, &end);
    if (rc == 1) emu_huffman_encode_segment(reinterpret_cast<const lep_huff_image*>(&hi), reinterpret_cast<const lep_huff_segment*>(&sg), out + sc.out_off, &n, &end);
    const bool over = end.attempted > sc.out_cap || (end.pad & (lephuff::kHuffEndCut | lephuff::kHuffEndRefused)) != 0;
    return n | (over ? 0x80000000u : 0u);
}
extern "C" int emu_huffman_progressive_encode(const lep_huffprog_image* img, const lep_huffprog_scan* scans, int nscan, uint8_t* out, uint32_t* corr, uint32_t* out_len) {
    static lephuff::ProgShared sh;
    for (int i = 0; i < nscan; ++i) {
        if (lephuff::prog_is_sequential(*reinterpret_cast<const lephuff::ProgScan*>(scans + i))) {
            out_len[i] = emu_sequential_scan(reinterpret_cast<const lephuff::ProgImage*>(img), *reinterpret_cast<const lephuff::ProgScan*>(scans + i), out, false);
            continue;
        }
        lephuff::ProgWave w;
        out_len[i] = w.run_scan(reinterpret_cast<const lephuff::ProgImage*>(img), reinterpret_cast<const lephuff::ProgScan*>(scans + i), &sh, out, corr);
    }
    return 0;
}

// ... with one lane per run of blocks (lep_huffprog_simt.h): count, place, assign, code, stuff, every pass one emulated wavefront
// after the other.  taken[i]: 1 when that form took scan i (the others are written by the wavefront form above, as on the GPU).
// region_bytes: > 0 stands in for the launch code's region size (tests: a region that does not suffice)
#include "../../lepton_amd/csrc/lep_huffprog_simt.h"
extern "C" int emu_huffman_progressive_encode_simt(const lep_huffprog_image* img, const lep_huffprog_scan* scans, int nscan, uint8_t* out, uint32_t* corr, uint32_t* out_len,
                                                   int32_t* taken, uint64_t region_bytes) {
    static lephuff::ProgSimtShared sh;
    static lephuff::ProgShared shw;
    const lephuff::ProgImage* im = reinterpret_cast<const lephuff::ProgImage*>(img);
    std::vector<lephuff::ProgScan> sv((size_t)nscan);
    memcpy(sv.data(), scans, sizeof(lephuff::ProgScan) * (size_t)nscan);
    std::vector<lephuff::ProgSimtScan> ps;
    size_t nunits = 0;
    uint64_t sum_cap = 0, bound = 0;
    for (int i = 0; i < nscan; ++i) {
        bound = std::max<uint64_t>(bound, sv[(size_t)i].pad);   // (lep_huffprog_scan.file_bound)
        sv[(size_t)i].pad = 0; sv[(size_t)i].image = 0; taken[i] = 0;
        if (lephuff::prog_is_sequential(sv[(size_t)i])) { out_len[i] = emu_sequential_scan(im, sv[(size_t)i], out, true); taken[i] = 1; continue; }
        uint32_t nb = 0, nu = 0;
        if (!lephuff::prog_simt_takes(*im, sv[(size_t)i], &nb, &nu)) continue;
        lephuff::ProgSimtScan e;
        memset(&e, 0, sizeof e);
        e.scan = (uint32_t)i; e.first_unit = (uint32_t)nunits; e.nunits = nu; e.nblocks = nb;
        nunits += nu;
        sum_cap += (uint64_t)sv[(size_t)i].out_cap + 96;
        sv[(size_t)i].pad = lephuff::kProgScanSimt; taken[i] = 1;
        ps.push_back(e);
    }
    lephuff::ProgSimtRegion r{0u, (uint32_t)ps.size(), 0, 0};
    r.bytes = region_bytes ? region_bytes : ((bound ? std::min<uint64_t>(sum_cap, bound + 96ull * r.nps + 4096) : sum_cap) + 15) & ~(uint64_t)15;
    std::vector<uint32_t> words(nunits * lephuff::kProgSimtUnitWords + 1, 0xdeadbeefu);
    std::vector<uint32_t> scratch((size_t)r.bytes / 4 + 8, 0xa5a5a5a5u);   // (garbage: the clearing pass has to do its work)
    uint8_t* scb = reinterpret_cast<uint8_t*>(scratch.data());
    lephuff::ProgSimtUnits U;
    U.set(words.data(), nunits);
    for (auto& e : ps) for (uint32_t f = 0; f < e.nunits; f += 64) lephuff::prog_simt_units<false>(im, sv.data(), &e, &sh, U, scb, f);
    for (auto& e : ps) lephuff::prog_simt_place(sv.data(), &e, U);
    if (!ps.empty()) lephuff::prog_simt_assign(r, ps.data());
    for (auto& e : ps) {   // (lep_huffprog_simt_zero_kernel)
        const uint64_t need16 = std::min<uint64_t>(((uint64_t)e.total_bits + 7) / 8 / 16 + 2, e.buf_bytes / 16);
        memset(scb + e.buf_off, 0, (size_t)need16 * 16);
    }
    for (auto& e : ps) for (uint32_t f = 0; f < e.nunits; f += 64) lephuff::prog_simt_units<true>(im, sv.data(), &e, &sh, U, scb, f);
    for (auto& e : ps) lephuff::prog_simt_stuff(im, sv.data(), e, scb, out, out_len);
    for (int i = 0; i < nscan; ++i)
        if (!taken[i]) { lephuff::ProgWave w; out_len[i] = w.run_scan(im, &sv[(size_t)i], &shw, out, corr); }
    return 0;
}

// GPU Huffman scan decoder (lep_huffdec.h) as a 64-lane loop emulation: one image
#include "../../lepton_amd/csrc/lep_huffdec.h"
extern "C" int emu_huffman_decode_image(const lep_huffdec_image* img, lep_huffdec_row* rows) {
    static lephuff::HuffDecShared sh;
    lephuff::HuffDecWave w;
    lephuff::HuffDecImage im;
    memcpy(&im, img, sizeof im);
    im.rows_off = 0;
    w.run(&im, &sh, reinterpret_cast<lephuff::HuffDecRow*>(rows));
    return 0;
}

// progressive scan decoder (lep_huffprogdec.h): the scans of one image, level by level, one emulated wavefront after the other
#include "../../lepton_amd/csrc/lep_huffprogdec.h"
// (a scan of a SEQUENTIAL frame coded in several scans goes where the launch code sends it: to the sequential scan decoders, as an image of
// its own -- lep_huffprogdec.h sequential_scan_image, lep_gpu.hip lep_gpu_huffman_progressive_decode_device; defined at the end of the file)
static void emu_sequential_scan_decode(const lephuff::ProgDecScan& sc, lephuff::HuffDecRow* rows, bool lanes);
extern "C" int emu_huffman_progressive_decode(const lep_huffprogdec_scan* scans, int nscan, lep_huffdec_row* rows) {
    static lephuff::HuffDecShared sh;
    for (int lv = 0; lv < 64; ++lv)
        for (int i = 0; i < nscan; ++i)
            if (scans[i].level == lv && lephuff::progdec_is_sequential(*reinterpret_cast<const lephuff::ProgDecScan*>(scans + i)))
                emu_sequential_scan_decode(*reinterpret_cast<const lephuff::ProgDecScan*>(scans + i), reinterpret_cast<lephuff::HuffDecRow*>(rows), false);
            else
            if (scans[i].level == lv) { lephuff::ProgDecWave w; w.run_scan(reinterpret_cast<const lephuff::ProgDecScan*>(scans + i), &sh, reinterpret_cast<lephuff::HuffDecRow*>(rows)); }
    return 0;
}

// ... as ONE pipelined launch (all levels; a scan follows the scans of its file in front of it MCU row by MCU row): the launch
// order, the dependencies the host works out (returned through deps_out[nscan][4], indices into the caller's array) and the
// waiting / publishing code, run one scan after the other in launch order.  waits_out: how often a scan found the scans in front
// of it not far enough (0 here by construction; the code path that polls is the GPU's).
extern "C" int emu_huffman_progressive_decode_pipelined(const lep_huffprogdec_scan* scans, int nscan, lep_huffdec_row* rows, int32_t* deps_out) {
    static lephuff::HuffDecShared sh;
    std::vector<lephuff::ProgDecScan> sorted;
    std::vector<int> order;
    for (int lv = 0; lv < 64; ++lv)
        for (int i = 0; i < nscan; ++i)
            if (scans[i].level == lv) { sorted.push_back(*reinterpret_cast<const lephuff::ProgDecScan*>(scans + i)); order.push_back(i); }
    if ((int)sorted.size() != nscan) return -1;
    std::vector<lephuff::ProgDeps> deps((size_t)nscan);
    if (!lephuff::prog_scan_deps(sorted.data(), order.data(), nscan, deps.data())) return -2;
    std::vector<uint32_t> progress((size_t)nscan, 0u);
    for (int k = 0; k < nscan; ++k) {
        for (int d = 0; d < 4; ++d) {
            const int j = deps[(size_t)k].dep[d];
            if (j >= k) return -3;                                  // a scan may only follow scans in front of it in the launch
            deps_out[order[(size_t)k] * 4 + d] = j < 0 ? -1 : order[(size_t)j];
        }
        if (lephuff::progdec_is_sequential(sorted[(size_t)k])) { emu_sequential_scan_decode(sorted[(size_t)k], reinterpret_cast<lephuff::HuffDecRow*>(rows), false); progress[(size_t)k] = 0x7fffffffu; continue; }
        lephuff::ProgDecWave w;
        w.run_scan<true>(&sorted[(size_t)k], &sh, reinterpret_cast<lephuff::HuffDecRow*>(rows), &deps[(size_t)k], progress.data(), k);
        if (progress[(size_t)k] != 0x7fffffffu) return -4;          // every scan says when it is done, whatever happened to it
    }
    return 0;
}



// ... with the window of speculative codes (lep_huffprogdec_win.h) for the scans that form takes: level by level (pipelined = 0) or
// in the one launch's order with its waiting / publishing code (pipelined = 1).  taken: how many scans the window form decoded.
#include "../../lepton_amd/csrc/lep_huffprogdec_win.h"
extern "C" int emu_huffman_progressive_decode_win(const lep_huffprogdec_scan* scans, int nscan, lep_huffdec_row* rows, int pipelined, int32_t* taken) {
    static lephuff::HuffDecShared sh;
    static lephuff::ProgWinShared ws;
    std::vector<lephuff::ProgDecScan> sorted;
    std::vector<int> order;
    for (int lv = 0; lv < 64; ++lv)
        for (int i = 0; i < nscan; ++i)
            if (scans[i].level == lv) { sorted.push_back(*reinterpret_cast<const lephuff::ProgDecScan*>(scans + i)); order.push_back(i); }
    if ((int)sorted.size() != nscan) return -1;
    std::vector<lephuff::ProgDeps> deps((size_t)nscan);
    if (pipelined && !lephuff::prog_scan_deps(sorted.data(), order.data(), nscan, deps.data())) return -2;
    std::vector<uint32_t> progress((size_t)nscan, 0u);
    *taken = 0;
    for (int k = 0; k < nscan; ++k) {
        lephuff::HuffDecRow* r = reinterpret_cast<lephuff::HuffDecRow*>(rows);
        if (lephuff::progdec_is_sequential(sorted[(size_t)k])) { emu_sequential_scan_decode(sorted[(size_t)k], r, true); progress[(size_t)k] = 0x7fffffffu; continue; }
        const bool win = lephuff::prog_win_takes(sorted[(size_t)k]);
        *taken += win;
        if (win) {
            lephuff::ProgWinWave w;
            if (pipelined) w.run_scan_win<true>(&sorted[(size_t)k], &ws, r, &deps[(size_t)k], progress.data(), k);
            else w.run_scan_win<false>(&sorted[(size_t)k], &ws, r);
        } else {
            lephuff::ProgDecWave w;
            if (pipelined) w.run_scan<true>(&sorted[(size_t)k], &sh, r, &deps[(size_t)k], progress.data(), k);
            else w.run_scan<false>(&sorted[(size_t)k], &sh, r);
        }
        if (pipelined && progress[(size_t)k] != 0x7fffffffu) return -4;
    }
    return 0;
}

// one lane per subsequence (lep_huffdec_simt.h): guess, settle passes, place, write -- every pass one wavefront after the other.
// settle_moved[k] (k = 0 .. kSimtSettle): whether pass k saw an end state move; nsub_out: subsequences the scan was cut into.
#include "../../lepton_amd/csrc/lep_huffdec_simt.h"
extern "C" int emu_huffman_decode_image_simt(const lep_huffdec_image* img, lep_huffdec_row* rows, uint32_t sub_bits, int32_t* settle_moved, uint32_t* nsub_out) {
    static lephuff::SimtShared sh;
    lephuff::HuffDecImage im;
    memcpy(&im, img, sizeof im);
    im.rows_off = 0;
    const uint32_t L = (sub_bits + 31u) & ~31u;
    if (!L) return -1;
    lephuff::SimtImage si;
    memset(&si, 0, sizeof si);
    si.first = 0; si.sub_bits = L;
    si.nsub = (uint32_t)std::max<uint64_t>(1, ((uint64_t)im.scan_len * 8u + L - 1) / L);
    if (im.flags & lephuff::kHuffDecRstTable) {   // lane = restart interval (lep_gpu_huffman_decode_simt_device)
        if (im.rsti <= 0 || im.mcuc <= 0) return -1;
        si.nsub = (uint32_t)((im.mcuc - 1) / im.rsti) + 1u;
        si.changed[0] = 0xff;
    }
    std::vector<lephuff::SimtSub> buf[2] = {std::vector<lephuff::SimtSub>(si.nsub), std::vector<lephuff::SimtSub>(si.nsub)};
    std::vector<lephuff::SimtPlace> place(si.nsub);
    for (int k = 0; k <= lephuff::kSimtSettle; ++k)
        for (uint32_t f = 0; f < si.nsub; f += 64) lephuff::simt_guess_or_settle(&im, &sh, &si, buf[(k + 1) & 1].data(), buf[k & 1].data(), f, k);
    const lephuff::SimtSub* fin = buf[lephuff::kSimtSettle & 1].data();
    lephuff::simt_place(&im, &si, fin, place.data(), lephuff::kSimtSettle, reinterpret_cast<lephuff::HuffDecRow*>(rows));
    static lephuff::SimtTile tile;
    for (uint32_t f = 0; f < si.nsub; f += 64) lephuff::simt_write(&im, &sh, &tile, &si, fin, place.data(), reinterpret_cast<lephuff::HuffDecRow*>(rows), f);
    if (rows[im.mcuv].aux == lephuff::kHuffDecRowUnwritten) { si.status |= 2; rows[im.mcuv].aux = 255; }   // (lep_huffman_simt_finish_kernel)
    if (im.flags & lephuff::kHuffDecRstTable) {
        int status = si.status & 0x3fffff;
        const int pad = lephuff::simt_intervals_pad(&si, &status);
        rows[im.mcuv].aux = pad | (status << 8);
    } else
    rows[im.mcuv].aux = (rows[im.mcuv].aux & (255 | lephuff::kHuffDecRowTruncated)) | ((si.status & 0x3fffff) << 8);
    if (settle_moved) for (int k = 0; k <= lephuff::kSimtSettle; ++k) settle_moved[k] = si.changed[k];
    if (nsub_out) *nsub_out = si.nsub;
    return 0;
}

// split-phase encoder (lep_enc5.h): count -> plan -> emit -> fold (every chain) -> gather -> write, one segment, every pass
// stepped as a 64-lane loop emulation; bins_out (optional): the (probability | bit << 8) list the writer consumed
#include "../../lepton_amd/csrc/lep_enc5.h"
template <int NW>
static int encode_segment_v5(const lep_image_desc* d, int y0, int y1, int is_last, uint8_t* out, uint32_t cap, uint32_t* len, uint32_t* bins,
                             uint16_t* bins_out, uint32_t bins_out_cap, int nparts = 1) {
    using namespace lep5;
    ImageDev img;
    int rc = derive_image(*d, &img, true);
    if (rc) return rc;
    std::vector<NSum> ns(img.ns_total);
    SegDev seg;
    seg.image = 0; seg.y0 = y0; seg.y1 = y1; seg.is_last = is_last; seg.stream_off = 0; seg.stream_cap = cap; seg.slot = 0;
    static Walk5Shared wsh;
    static FoldShared fsh;
    std::vector<uint32_t> counts(kCountWords, 0);
    static SegPlan5 plan;
    {
        Walk5<kCount, NW> w;
        memset(ns.data(), 0, ns.size() * sizeof(NSum));
        w.run(&img, seg, ns.data(), &wsh, &plan, nullptr, nullptr);
        export_counts(w, &wsh, counts.data());
    }
    plan_segment(counts.data(), &plan);
    std::vector<uint8_t> arena(plan.arena_bytes + 64, 0xA5);   // poisoned: every byte read must have been written
    std::vector<uint16_t> binlist(plan.bins_cap + 64, 0xA5A5);
    {
        Walk5<kEmit, NW> w;
        memset(ns.data(), 0, ns.size() * sizeof(NSum));
        rc = w.run(&img, seg, ns.data(), &wsh, &plan, arena.data(), nullptr, 0, 0, nparts);
        if (rc) return rc;
        if (w.sign_pos[0] != plan.sign_cnt[0] || w.sign_pos[1] != plan.sign_cnt[1] || w.ord0 != plan.nblocks) return 1001;
        for (int i = 0; i < kStreams; ++i) if (wsh.cursor[i] != plan.base[i] + plan.cnt[i]) return 1002;
    }
    {
        static BucketShared bsh;
        bucket_wave(&plan, arena.data(), &bsh);
    }
    std::vector<uint32_t> thresh(kThreshWords, kBranchInit);
    for (int ci = 0; ci < 2; ++ci) {
        for (int row = 0; row < kRows; ++row)
            for (int k = 0; k < kClasses; ++k) {
                const int sid = stream_id(ci, row, k);
                if (!plan.cnt[sid]) continue;
                if (row < 63) fold_coef_wave(&plan, arena.data(), 0, 1, sid, &fsh);   // (row 63: the threshold chains of this colour index, class = lt)
                else fold_thresh_wave(&plan, arena.data(), thresh.data(), 0, 1, sid, ci, &fsh);
            }
        fold_sign_wave(&plan, arena.data(), 0, 1, ci, &fsh);
        for (int b = 0; b < 10; ++b) fold_nz_wave(&plan, arena.data(), 0, 1, ci, b, &fsh);
        for (int v = 0; v < 2; ++v) for (int e = 0; e < 8; ++e) fold_edgenz_wave(&plan, arena.data(), 0, 1, ci, v, e, &fsh);
    }
    for (int a = 0; a < 12; ++a) fold_dc_wave(&plan, arena.data(), 0, 1, a, &fsh);
    for (int part = 0; part < nparts; ++part) {   // (every part: a walker of its own, as on the GPU)
        Walk5<kGather, NW> w;
        memset(ns.data(), 0, ns.size() * sizeof(NSum));
        rc = w.run(&img, seg, ns.data(), &wsh, &plan, arena.data(), binlist.data(), 0, part, nparts);
        if (rc) return rc;
        if (w.nbins > plan.bins_cap) return 1003;
        if (part + 1 < nparts && w.nbins != reinterpret_cast<const Ckpt5*>(arena.data() + plan.ckpt_base)[part + 1].nbins) return 1004;   // emit's bin count at the checkpoint
        plan.nbins = w.nbins;
    }
    if (bins) *bins = plan.nbins;
    if (bins_out) memcpy(bins_out, binlist.data(), 2 * (size_t)(plan.nbins < bins_out_cap ? plan.nbins : bins_out_cap));
    uint32_t slen = 0;
    int32_t status = 0;
    for (int part = 0; part < nparts; ++part) write_wave(&plan, binlist.data(), &seg, 0, 1, out, &slen, &status, arena.data(), part, nparts);
    *len = slen;
    {   // the stitched writer over the same bin list (K chunks, K and the warm-up varied with the list): the same bytes, the same verdict
        const int K = 2 << (plan.nbins % 6);
        const uint32_t warms[4] = {kWarm5, 100, 3000, 1u << 22};
        std::vector<WChunk5> recs((size_t)K);
        memset(recs.data(), 0, recs.size() * sizeof(WChunk5));
        std::vector<uint8_t> again((size_t)cap + 64, 0x5A);
        uint32_t slen2 = 0;
        int32_t status2 = 0;
        for (int k = 0; k < K; ++k) wchunk_range_lane(plan, binlist.data(), &recs[(size_t)k], k, K, warms[(plan.nbins >> 3) & 3]);
        wchunk_link_lane(plan, binlist.data(), recs.data(), K);
        for (int k = 0; k < K; ++k) wchunk_code_lane(plan, binlist.data(), seg, again.data(), &recs[(size_t)k], k, K);
        wchunk_stitch_lane(plan, seg, again.data(), &slen2, &status2, recs.data(), K);
        if ((status2 == 100) != (status == 100)) return 1006;
        if (status != 100 && (slen2 != slen || memcmp(again.data(), out, slen))) return 1007;
    }
    return status == 100 ? LEP_BUFFER_TOO_SMALL : status;
}
extern "C" int emu_encode_segment_v5(const lep_image_desc* d, int y0, int y1, int is_last, uint8_t* out, uint32_t cap, uint32_t* len, uint32_t* bins,
                                     uint16_t* bins_out, uint32_t bins_out_cap) {
    return encode_segment_v5<1>(d, y0, y1, is_last, out, cap, len, bins, bins_out, bins_out_cap);
}
// ... with the walks split into their two halves (what two wavefronts per segment run side by side on the GPU)
extern "C" int emu_encode_segment_v5_halves(const lep_image_desc* d, int y0, int y1, int is_last, uint8_t* out, uint32_t cap, uint32_t* len, uint32_t* bins,
                                            uint16_t* bins_out, uint32_t bins_out_cap) {
    return encode_segment_v5<2>(d, y0, y1, is_last, out, cap, len, bins, bins_out, bins_out_cap);
}
// ... with gather and write in parts (tile ranges of the segment, taken up from emit's checkpoints; the writer's state carried over)
extern "C" int emu_encode_segment_v5_parts(const lep_image_desc* d, int y0, int y1, int is_last, uint8_t* out, uint32_t cap, uint32_t* len, uint32_t* bins,
                                           uint16_t* bins_out, uint32_t bins_out_cap) {
    const int rc = encode_segment_v5<2>(d, y0, y1, is_last, out, cap, len, bins, bins_out, bins_out_cap, 4);
    if (rc) return rc;
    std::vector<uint8_t> again(cap);
    uint32_t len2 = 0, bins2 = 0;
    const int rc2 = encode_segment_v5<1>(d, y0, y1, is_last, again.data(), cap, &len2, &bins2, nullptr, 0, 7);   // one wavefront, seven parts: the same bytes
    if (rc2) return rc2;
    return (len2 == *len && !memcmp(again.data(), out, len2)) ? 0 : 1005;
}


// lep5::BoolEnc5 (deferred byte output, carry cache) against the serial writer of lep_core.h on random and adversarial bin
// sequences: uniform, nearly-certain bins coded wrong (long shifts), runs that produce 0xFF bytes and carries into them, tiny
// output buffers (overflow verdicts).  Returns 0, or 1 + the trial that differed.
extern "C" int emu_check_bool_writer5(int trials) {
    uint64_t rs = 88172645463325252ull;
    auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); };
    std::vector<uint8_t> a(1 << 16), b(1 << 16);
    for (int trial = 0; trial < trials; ++trial) {
        const int n = (int)(rnd() % 3000), mode = trial % 6;
        std::vector<uint16_t> bins((size_t)n);
        for (int i = 0; i < n; ++i) {
            uint32_t p, bit;
            if (mode == 0) { p = rnd() % 256; bit = rnd() & 1; }
            else if (mode == 1) { p = 1 + rnd() % 3; bit = (rnd() % 8) != 0; }
            else if (mode == 2) { p = 250 + rnd() % 6; bit = (rnd() % 8) == 0; }
            else if (mode == 3) { p = 128; bit = 1; }
            else if (mode == 4) { p = rnd() % 256; bit = (rnd() % 256) < p ? 0 : 1; }
            else { p = (rnd() & 1) ? 255 : 1; bit = rnd() & 1; }
            bins[(size_t)i] = (uint16_t)(p | (bit << 8));
        }
        const uint32_t cap = (trial % 50 == 0) ? rnd() % 200 : 1u << 16;
        memset(a.data(), 0xAA, a.size()); memset(b.data(), 0xBB, b.size());
        lepdev::BoolCoder<false> ref;
        ref.init_stream(a.data(), cap);
        for (int i = 0; i < n; ++i) ref.put(bins[(size_t)i] >> 8, bins[(size_t)i] & 255);
        const uint32_t la = ref.finish();
        lep5::BoolEnc5 e;
        e.init(b.data(), cap);
        e.bin(0, 128);
        int g = 1;
        for (int i = 0; i < n; ++i) { e.bin(bins[(size_t)i] >> 8, bins[(size_t)i] & 255); if (++g == 4) { e.flush(); g = 0; } }
        bool ov = false;
        const uint32_t lb = e.finish(&ov);
        if (ref.overflow != ov) return 1 + trial;
        if (!ov && (la != lb || memcmp(a.data(), b.data(), la))) return 1 + trial;
    }
    return 0;
}


// the stitched writer (lep_enc5.h: range / link / code / stitch over K chunks of a bin list) against lepdev::BoolCoder<false>: bin
// lists of up to 40,000 bins in the six flavours above, K = 2 .. 64, warm-ups from 8 bins (every guess wrong: the link pass redoes
// the chunks) to longer than the list (every chunk warmed up from the stream's start), tiny buffers.  Returns 0 or 1 + the trial;
// *redone counts the chunks whose guessed start range was wrong.
extern "C" int emu_check_stitched_writer5(int trials, int* redone) {
    uint64_t rs = 0x2545F4914F6CDD1Dull;
    auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); };
    std::vector<uint8_t> a(1 << 17), b(1 << 17);
    int wrong = 0;
    for (int trial = 0; trial < trials; ++trial) {
        const int n = (int)(rnd() % (trial % 7 == 0 ? 40000 : 3000)), mode = trial % 6;
        std::vector<uint16_t> bins((size_t)n + 256, 0);
        for (int i = 0; i < n; ++i) {
            uint32_t p, bit;
            if (mode == 0) { p = rnd() % 256; bit = rnd() & 1; }
            else if (mode == 1) { p = 1 + rnd() % 3; bit = (rnd() % 8) != 0; }
            else if (mode == 2) { p = 250 + rnd() % 6; bit = (rnd() % 8) == 0; }
            else if (mode == 3) { p = 128; bit = 1; }
            else if (mode == 4) { p = rnd() % 256; bit = (rnd() % 256) < p ? 0 : 1; }
            else { p = (rnd() & 1) ? 255 : 1; bit = rnd() & 1; }
            bins[(size_t)i] = (uint16_t)(p | (bit << 8));
        }
        const uint32_t cap = (trial % 50 == 0) ? rnd() % 200 : 1u << 17;
        memset(a.data(), 0xAA, a.size()); memset(b.data(), 0xBB, b.size());
        lepdev::BoolCoder<false> ref;
        ref.init_stream(a.data(), cap);
        for (int i = 0; i < n; ++i) ref.put(bins[(size_t)i] >> 8, bins[(size_t)i] & 255);
        const uint32_t la = ref.finish();
        const int K = 2 << (rnd() % 6);
        const uint32_t warms[5] = {8, 64, 700, 16384, 1u << 20};
        const uint32_t warm = warms[rnd() % 5];
        lep5::SegPlan5 P;
        memset(&P, 0, sizeof P);
        P.nbins = (uint32_t)n; P.bins_off = 0; P.status = 0;
        SegDev sd;
        memset(&sd, 0, sizeof sd);
        sd.stream_off = 0; sd.stream_cap = cap; sd.slot = 0;
        std::vector<lep5::WChunk5> recs((size_t)K);
        memset(recs.data(), 0, recs.size() * sizeof(lep5::WChunk5));
        for (int k = 0; k < K; ++k) lep5::wchunk_range_lane(P, bins.data(), &recs[(size_t)k], k, K, warm);
        for (int k = 1; k < K; ++k) wrong += recs[(size_t)k].q_guess != recs[(size_t)k - 1].q_end;   // (before link: against the unlinked ends -- an upper bound)
        lep5::wchunk_link_lane(P, bins.data(), recs.data(), K);
        for (int k = K - 1; k >= 0; --k) lep5::wchunk_code_lane(P, bins.data(), sd, b.data(), &recs[(size_t)k], k, K);   // any order: the chunks do not meet
        uint32_t lb = 0;
        int32_t stt = 0;
        lep5::wchunk_stitch_lane(P, sd, b.data(), &lb, &stt, recs.data(), K);
        const bool ov = stt == 100;
        if (ref.overflow != ov) return 1 + trial;
        if (!ov && (la != lb || memcmp(a.data(), b.data(), la))) return 1 + trial;
    }
    if (redone) *redone = wrong;
    return 0;
}

// the 16-bit Branch of the fold lanes (lep5::upd16 / prob16) against the packed word of lep_core.h (branch_update): random walks
// with every bias, long enough to saturate either way and to renormalise; returns 0 or 1 + the walk that differed
extern "C" int emu_check_branch16(int walks) {
    uint64_t rs = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 11); };
    for (int w = 0; w < walks; ++w) {
        uint32_t a = lepdev::kBranchInit, b = lep5::kBranchInit16;
        const uint32_t bias = (w % 9) * 32;   // 0: always false ... 256: always true
        for (int i = 0; i < 3000; ++i) {
            if (lep5::prob16(b) != (a >> 16)) return 1 + w;
            uint32_t obs = (rnd() % 256) < bias ? 1u : 0u;
            if (w % 7 == 3 && i > 300 && i < 310) obs ^= 1u;   // a few surprises after saturation
            a = lepdev::branch_update(a, (int)obs);
            b = lep5::upd16(b, obs);
        }
    }
    return 0;
}

// the count walk's result for one segment: counts[lep5::kCountWords] (units per stream, sign bytes, blocks, bins bound, tiles)
extern "C" int emu_v5_counts(const lep_image_desc* d, int y0, int y1, int is_last, uint32_t* counts) {
    using namespace lep5;
    ImageDev img;
    int rc = derive_image(*d, &img, true);
    if (rc) return rc;
    std::vector<NSum> ns(img.ns_total);
    SegDev seg;
    seg.image = 0; seg.y0 = y0; seg.y1 = y1; seg.is_last = is_last; seg.stream_off = 0; seg.stream_cap = 0; seg.slot = 0;
    static Walk5Shared wsh;
    static SegPlan5 plan;
    Walk5<kCount> w;
    memset(ns.data(), 0, ns.size() * sizeof(NSum));
    w.run(&img, seg, ns.data(), &wsh, &plan, nullptr, nullptr);
    export_counts(w, &wsh, counts);
    return 0;
}

// lep_huffprogdec.h prog_scan_deps on descriptors a test made up (only the frame pointer, the components and the band are looked at)
extern "C" int emu_prog_scan_deps(const lep_huffprogdec_scan* scans, const int* order, int n, int32_t* deps_out) {
    std::vector<lephuff::ProgDeps> deps((size_t)n);
    const bool ok = lephuff::prog_scan_deps(reinterpret_cast<const lephuff::ProgDecScan*>(scans), order, n, deps.data());
    for (int i = 0; i < n; ++i) for (int d = 0; d < 4; ++d) deps_out[4 * i + d] = deps[(size_t)i].dep[d];
    return ok ? 1 : 0;
}

static void emu_sequential_scan_decode(const lephuff::ProgDecScan& sc, lephuff::HuffDecRow* rows, bool lanes) {
    const lephuff::HuffDecImage im = lephuff::sequential_scan_image(sc);
    lep_huffdec_row* at = reinterpret_cast<lep_huffdec_row*>(rows + im.rows_off);
    if (lanes && lephuff::sequential_scan_for_lanes(im)) {
        int32_t moved[lephuff::kSimtSettle + 1];
        uint32_t nsub = 0;
        // (subsequences as the launch code cuts them: 8192 bits, or 64 of the scan's average block if that is more)
        uint64_t nblocks = 0;
        for (int ci = 0; ci < im.ncomp && ci < 4; ++ci) { const int cmp = im.scan_cmp[ci] & 3; nblocks += (uint64_t)im.hs[cmp] * im.vs[cmp]; }
        nblocks *= (uint64_t)std::max(im.mcuc, 1);
        const uint64_t b = (uint64_t)im.scan_len * 8u;
        const uint32_t sub_bits = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(8192, (64 * b / std::max<uint64_t>(nblocks, 1) + 31) & ~(uint64_t)31), 1u << 24);
        emu_huffman_decode_image_simt(reinterpret_cast<const lep_huffdec_image*>(&im), at, sub_bits, moved, &nsub);
    } else emu_huffman_decode_image(reinterpret_cast<const lep_huffdec_image*>(&im), at);
}
