// jpeg_scan.cc -- JPEG file -> (header bytes, un-stuffed scan bytes, garbage, DCT coefficients,
// one Huffman hand-off record per MCU row).  Host-side caller of the hot path; behaviour follows
//   read_jpeg          src/lepton/jpgcoder.cc:2269-2466
//   setup_imginfo_jpg  src/lepton/jpgcoder.cc:4450-4539
//   parse_jfif_jpg     src/lepton/jpgcoder.cc:4545-4843
//   decode_jpeg        src/lepton/jpgcoder.cc:2799-3302   (sequential scans; progressive: jpeg_progressive.cc)
//   decode_block_seq   src/lepton/jpgcoder.cc:4893-4966
//   build_huffcodes    src/lepton/jpgcoder.cc:5507-5606
//   abitreader         src/lepton/bitops.hh:229-362
// so that the .lep container written from it is byte-identical to the reference's.
#include "jpeg_model.h"
#include "jpeg_bits.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>

namespace lep {

const uint8_t kAlignedToRaster[64] = {
    9, 10, 17, 25, 18, 11, 12, 19, 26, 33, 41, 34, 27, 20, 13, 14, 21, 28, 35, 42, 49, 57, 50, 43, 36,
    29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
    0, 1, 2, 3, 4, 5, 6, 7, 8, 16, 24, 32, 40, 48, 56};
const uint8_t kZigzagToRaster[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34,
    27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const uint8_t kZigzagToAligned[64] = {
    49, 50, 57, 58, 0, 51, 52, 1, 2, 59, 60, 3, 4, 5, 53, 54, 6, 7, 8, 9, 61, 62, 10, 11,
    12, 13, 14, 55, 56, 15, 16, 17, 18, 19, 20, 63, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32,
    33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48};

static inline unsigned be16(unsigned a, unsigned b) { return (a << 8) + b; }

// ------------------------------------------------------------------------------------------------
// Huffman tables.  Encode side: code and length per symbol in DHT order (T.81 Annex C; a symbol listed twice keeps its last
// code).  Decode side: the table's code words as sorted, disjoint intervals of 16-bit patterns (HuffTable::wfirst).
//
// A DHT that follows Annex C is prefix-free and its words are its codes.  One that does not is still decoded the way the
// reference would (jpgcoder.cc:5507-5606 enters the symbols 0..255, in that order, into a binary tree of at most 256 inner
// nodes, and its walk trusts whatever link it finds).  Stated on code words, symbol by symbol:
//   * a code that extends a word already entered is dropped (the walk would stop at that word);
//   * a code that other words extend replaces them (they can no longer be reached);
//   * every bit of a code but its last that no earlier code shares costs one inner node, numbered 1, 2, ... as they are made;
//     the node that would be number 256 or more is not an inner node but reads as the symbol (number - 256), the code that
//     needed it is dropped, and the bits up to that node have become a word of their own.
// strict (a file being compressed): a dropped code refuses the table; not strict (.lep headers are taken as they were written).
bool build_huff_table(const uint8_t* clen, size_t clen_avail, const uint8_t* cval, size_t cval_avail,
                      HuffTable* t, bool strict) {
    *t = HuffTable();
    struct Word { uint16_t first; uint8_t len, sym; };
    Word words[256];                                      // kept sorted by `first`
    int nw = 0;
    unsigned listed = 0, code = 0, inner = 0;
    bool annex_c = true;                                  // every symbol once, no code past its length, at most 255 inner nodes
    for (unsigned len = 1; len <= 16; ++len, code <<= 1) {
        const unsigned n = len - 1 < clen_avail ? clen[len - 1] : 0;
        for (unsigned j = 0; j < n; ++j, ++listed, ++code) {
            const unsigned at = listed & 0xff;
            const uint8_t sym = at < cval_avail ? cval[at] : 0;
            annex_c = annex_c && listed < 256 && !t->clen[sym] && code < (1u << len);
            t->clen[sym] = (uint16_t)len;
            t->cval[sym] = (uint16_t)code;
            if (!annex_c) continue;
            // codes come in rising order: the bits of this one (bar the last) that are not inner nodes yet are those it does not
            // share with the code before it
            const unsigned first = code << (16 - len);
            unsigned shared = 0;
            if (nw) {
                const unsigned diff = first ^ words[nw - 1].first;
                shared = std::min<unsigned>({diff ? (unsigned)__builtin_clz(diff) - 16 : 16u, words[nw - 1].len - 1u, len - 1});
            }
            inner += len - 1 - shared;
            words[nw++] = Word{(uint16_t)first, (uint8_t)len, sym};
        }
    }
    for (int i = 14; i >= 0; --i)                         // longest end-of-band run an AC table of a progressive scan can code
        if (t->clen[(i << 4) & 255] > 0) { t->max_eobrun = (2 << i) - 1; break; }
    if (!annex_c || inner > 255) nw = 0;                  // the general case below; otherwise the words are the codes as listed

    uint16_t seen_first[256];                             // codes entered so far: every bit but the last is an inner node
    uint8_t seen_len[256];
    int nseen = 0;
    unsigned next_node = 1;
    auto put = [&](Word w) {                              // w replaces the words that begin with it
        const unsigned end = w.first + (1u << (16 - w.len));
        int a = 0;
        while (a < nw && words[a].first < w.first) ++a;
        int b = a;
        while (b < nw && words[b].first < end) ++b;
        memmove(words + a + 1, words + b, (size_t)(nw - b) * sizeof(Word));
        words[a] = w;
        nw += a + 1 - b;
    };
    for (unsigned sym = 0; sym < 256 && !(annex_c && inner <= 255); ++sym) {
        const int len = t->clen[sym];
        if (!len) continue;
        const unsigned first = ((unsigned)t->cval[sym] << (16 - len)) & 0xffff;   // (an over-full table counts past its length: the low bits are the code)
        bool dropped = false;
        for (int k = 0; k < nw && words[k].first <= first; ++k)
            if (words[k].len < len && first - words[k].first < (1u << (16 - words[k].len))) dropped = true;
        int path = len;
        if (!dropped) {
            int shared = 0;                               // leading bits that are inner nodes already
            for (int k = 0; k < nseen; ++k) {
                const unsigned diff = first ^ seen_first[k];
                const int same = diff ? __builtin_clz(diff) - 16 : 16;
                shared = std::max(shared, std::min(same, seen_len[k] - 1));
            }
            for (int depth = std::min(shared, len - 1) + 1; depth < len; ++depth) {
                const unsigned id = next_node++;
                if (id < 256) continue;
                put(Word{(uint16_t)(first & (0xffff0000u >> depth)), (uint8_t)depth, (uint8_t)(id - 256)});
                path = depth;
                dropped = true;
                break;
            }
            seen_first[nseen] = (uint16_t)first;
            seen_len[nseen++] = (uint8_t)path;
            if (!dropped) put(Word{(uint16_t)first, (uint8_t)len, (uint8_t)sym});
        }
        if (dropped && strict) return false;
    }
    t->nwords = nw;
    for (int k = 0; k < nw; ++k) {
        t->wfirst[k] = words[k].first;
        t->wlen[k] = words[k].len;
        t->wsym[k] = words[k].sym;
        if (words[k].len <= 10)
            for (unsigned x = 0; x < (1u << (10 - words[k].len)); ++x)
                t->lut[(words[k].first >> 6) + x] = (uint16_t)((words[k].len << 8) | words[k].sym);
    }
    t->set = true;
    return true;
}

// ------------------------------------------------------------------------------------------------
bool parse_segment(JpegFile* jf, uint8_t type, unsigned len, unsigned avail, const uint8_t* seg, bool strict) {
    auto at = [&](unsigned i) -> unsigned { return i < avail ? seg[i] : 0u; };
    unsigned hpos = 4;
    switch (type) {
    case 0xC4: {  // DHT
        while (hpos < len) {
            unsigned cls = at(hpos) >> 4, id = at(hpos) & 15;
            if (cls >= 2 || id >= 4) break;
            ++hpos;
            if (!build_huff_table(seg + hpos, avail > hpos ? avail - hpos : 0, seg + hpos + 16,
                                  avail > hpos + 16 ? avail - hpos - 16 : 0, &jf->htab[cls][id], strict)) {
                jf->error = "huffman table out of space";
                return false;
            }
            unsigned skip = 16;
            for (unsigned i = 0; i < 16; ++i) skip += at(hpos + i);
            hpos += skip;
        }
        if (hpos != len) { jf->error = "size mismatch in dht marker"; return false; }
        return true;
    }
    case 0xDB: {  // DQT
        while (hpos < len) {
            unsigned prec = at(hpos) >> 4, id = at(hpos) & 15;
            if (prec >= 2 || id >= 4) break;
            ++hpos;
            if (prec == 0) {
                for (unsigned i = 0; i < 64; ++i) {
                    jf->qtables[id][i] = (uint16_t)at(hpos + i);
                    if (jf->qtables[id][i] == 0) break;
                }
                hpos += 64;
            } else {
                for (unsigned i = 0; i < 64; ++i) {
                    jf->qtables[id][i] = (uint16_t)be16(at(hpos + 2 * i), at(hpos + 2 * i + 1));
                    if (jf->qtables[id][i] == 0) break;
                }
                hpos += 128;
            }
        }
        if (hpos != len) { jf->error = "size mismatch in dqt marker"; return false; }
        return true;
    }
    case 0xDD: jf->rsti = (int)be16(at(hpos), at(hpos + 1)); return true;
    case 0xDA: {  // SOS
        jf->cs_cmpc = (int)at(hpos);
        if (jf->cs_cmpc > jf->ncomp) { jf->error = "too many components in scan"; return false; }
        ++hpos;
        for (int i = 0; i < jf->cs_cmpc; ++i) {
            int c = 0;
            while (c < jf->ncomp && (int)at(hpos) != jf->comp[c].jid) ++c;
            if (c == jf->ncomp) { jf->error = "component id mismatch in start-of-scan"; return false; }
            jf->cs_cmp[i] = c;
            jf->comp[c].dc_tbl = (int)(at(hpos + 1) >> 4);
            jf->comp[c].ac_tbl = (int)(at(hpos + 1) & 15);
            if (jf->comp[c].dc_tbl >= 4 || jf->comp[c].ac_tbl >= 4) { jf->error = "huffman table number mismatch"; return false; }
            hpos += 2;
        }
        jf->cs_from = (int)at(hpos);
        jf->cs_to = (int)at(hpos + 1);
        jf->cs_sah = (int)(at(hpos + 2) >> 4);
        jf->cs_sal = (int)(at(hpos + 2) & 15);
        if (jf->cs_from > jf->cs_to || jf->cs_from > 63 || jf->cs_to > 63) { jf->error = "spectral selection out of range"; return false; }
        if (jf->cs_sah >= 12 || jf->cs_sal >= 12) { jf->error = "successive approximation out of range"; return false; }
        return true;
    }
    case 0xC0: case 0xC1: case 0xC2: {
        jf->jpegtype = type == 0xC2 ? 2 : 1;
        if (at(hpos) != 8) { jf->error = "only 8 bit precision supported"; return false; }
        jf->height = (int)be16(at(hpos + 1), at(hpos + 2));
        jf->width = (int)be16(at(hpos + 3), at(hpos + 4));
        jf->ncomp = (int)at(hpos + 5);
        if (jf->ncomp > 4) { jf->ncomp = 4; jf->error = "max 4 components"; return false; }
        hpos += 6;
        for (int c = 0; c < jf->ncomp; ++c) {
            jf->comp[c].jid = (int)at(hpos);
            jf->comp[c].hs = (int)(at(hpos + 1) >> 4);
            jf->comp[c].vs = (int)(at(hpos + 1) & 15);
            if (jf->comp[c].hs > 4 || jf->comp[c].vs > 4) { jf->error = "sampling"; jf->warn = -EX_SAMPLING_BEYOND_FOUR_UNSUPPORTED; return false; }
            if (jf->comp[c].hs > 2 || jf->comp[c].vs > 2) { jf->error = "sampling"; jf->warn = -EX_SAMPLING_BEYOND_TWO_UNSUPPORTED; return false; }
            unsigned q = at(hpos + 2);
            if (q >= 4) { jf->error = "quantisation table index"; return false; }
            jf->comp[c].qidx = (int)q;
            hpos += 3;
        }
        return true;
    }
    case 0xC3: case 0xC5: case 0xC6: case 0xC7: case 0xC9: case 0xCA: case 0xCB: case 0xCD: case 0xCE: case 0xCF:
        jf->error = "unsupported SOF type (lossless / differential / arithmetic)";
        return false;
    case 0xD0: case 0xD1: case 0xD2: case 0xD3: case 0xD4: case 0xD5: case 0xD6: case 0xD7:
        jf->error = "rst marker found out of place"; return false;
    case 0xD8: jf->error = "soi marker found out of place"; return false;
    case 0xD9: jf->error = "eoi marker found out of place"; return false;
    default:
        if ((type >= 0xE0 && type <= 0xEF) || type == 0xFE) return true;
        jf->warn = std::max(jf->warn, 1);  // unknown marker: warning only
        return true;
    }
}

// ------------------------------------------------------------------------------------------------
bool setup_frame(JpegFile* jf) {
    size_t hpos = 0, hdrs = jf->hdr.size();
    const uint8_t* h = jf->hdr.data();
    while (hpos < hdrs) {
        uint8_t type = hpos + 1 < hdrs ? h[hpos + 1] : 0;
        unsigned len = 2 + be16(hpos + 2 < hdrs ? h[hpos + 2] : 0, hpos + 3 < hdrs ? h[hpos + 3] : 0);
        if (type != 0xDA && type != 0xC4 && type != 0xDD)
            if (!parse_segment(jf, type, len, (unsigned)(hdrs - hpos), h + hpos, true)) return false;
        hpos += len;
    }
    if (jf->ncomp == 0) { jf->error = "header contains incomplete information"; return false; }
    for (int c = 0; c < jf->ncomp; ++c)
        if (jf->comp[c].hs == 0 || jf->comp[c].vs == 0 || jf->qtables[jf->comp[c].qidx][0] == 0 || jf->jpegtype == 0) {
            jf->error = "header information is incomplete";
            return false;
        }
    jf->hmax = jf->vmax = 0;
    for (int c = 0; c < jf->ncomp; ++c) {
        jf->hmax = std::max(jf->hmax, jf->comp[c].hs);
        jf->vmax = std::max(jf->vmax, jf->comp[c].vs);
    }
    jf->mcuv = (int)ceilf((float)jf->height / (float)(8 * jf->vmax));
    jf->mcuh = (int)ceilf((float)jf->width / (float)(8 * jf->hmax));
    jf->mcuc = jf->mcuv * jf->mcuh;
    for (int c = 0; c < jf->ncomp; ++c) {
        Component& k = jf->comp[c];
        k.mbs = k.hs * k.vs;
        k.bcv = jf->mcuv * k.vs;
        k.bch = jf->mcuh * k.hs;
        k.bc = k.bcv * k.bch;
        k.ncv = (int)ceilf((float)jf->height * ((float)k.vs / (8.0f * jf->vmax)));
        k.nch = (int)ceilf((float)jf->width * ((float)k.hs / (8.0f * jf->hmax)));
        if (k.bch == 0 || k.bcv == 0) { jf->error = "zero sized component"; return false; }
        jf->trunc_bcv[c] = k.bcv;
        jf->trunc_bc[c] = k.bc;
    }
    // The reference's default build budgets (576 - 36) MiB for the coefficient frame = 4,423,680 blocks
    // (UncompressedComponents::max_number_of_blocks, jpgcoder.cc:826-885; ~190 Mpixel 4:2:0) and squeezes a larger image into
    // that budget as if it were a truncated file (uncompressed_components.hh:109-139).  We refuse such a frame instead
    // (TOO_MUCH_MEMORY_NEEDED): a serving process must not allocate 12 GB because a 5 KB header says 65500 x 65500.
    uint64_t total_blocks = 0;
    for (int c = 0; c < jf->ncomp; ++c) total_blocks += (uint64_t)jf->comp[c].bc;
    if (total_blocks > kMaxFrameBlocks) { jf->error = "coefficient frame beyond the memory budget"; jf->warn = -EX_TOO_MUCH_MEMORY_NEEDED; return false; }
    return true;
}

// ------------------------------------------------------------------------------------------------
// Step 1: split the file into header segments / scan bytes / garbage.
static int split_file(const uint8_t* d, size_t n, JpegFile* jf) {
    size_t pos = 2;  // past SOI
    uint8_t last2[2] = {0, 0};
    auto rd1 = [&](uint8_t* o) -> bool {
        if (pos >= n) return false;
        *o = d[pos++]; last2[0] = last2[1]; last2[1] = *o;
        return true;
    };
    auto rdn = [&](uint8_t* o, unsigned cnt) -> unsigned {
        if (cnt == 1) return rd1(o) ? 1 : 0;
        unsigned got = (unsigned)std::min<size_t>(cnt, n - pos);
        memcpy(o, d + pos, got);
        pos += got;
        if (got >= 2) { last2[0] = o[got - 2]; last2[1] = o[got - 1]; }   // (reference indexes by the requested size;
        else if (got) { last2[0] = last2[1]; last2[1] = o[0]; }          //  identical whenever the read is complete)
        return got;
    };
    std::vector<uint8_t> seg(1024);
    uint8_t type = 0, tmp = 0;
    int scnc = 0;
    bool have_hdr = false;
    for (;;) {
        if (type == 0xDA) {
            unsigned cpos = 0, crst = 0;
            jf->scan_start.push_back((uint32_t)jf->scan.size());
            jf->scan_file_range.emplace_back((uint32_t)pos, (uint32_t)pos);
            for (;;) {
                jf->scan_to_file.emplace_back((uint32_t)jf->scan.size(), (uint32_t)pos);
                if (!rd1(&tmp)) { jf->early_eof = true; have_hdr = true; break; }
                bool dead = false;
                if (tmp != 0xFF) {
                    crst = 0;
                    while (tmp != 0xFF) {
                        jf->scan.push_back(tmp);
                        if (!rd1(&tmp)) { jf->early_eof = true; have_hdr = true; dead = true; break; }
                    }
                }
                if (dead || tmp != 0xFF) break;
                if (!rd1(&tmp)) { jf->early_eof = true; have_hdr = true; break; }
                if (tmp == 0x00) { crst = 0; jf->scan.push_back(0xFF); }
                else if (tmp == 0xD0 + (cpos & 7)) {
                    if (scnc == 0) jf->rst_pos.push_back((uint32_t)jf->scan.size());
                    ++cpos; ++crst;
                    if (jf->rst_cnt.size() <= (size_t)scnc) jf->rst_cnt.resize(scnc + 1, 0);
                    ++jf->rst_cnt[scnc];
                } else {
                    if ((int)jf->rst_err.size() < scnc) jf->rst_err.resize(scnc, 0);
                    jf->rst_err.push_back((uint8_t)crst);
                    ++scnc;
                    seg[0] = 0xFF; seg[1] = tmp;
                    jf->scan_file_range.back().second = (uint32_t)(pos - 2);
                    break;
                }
            }
            if (jf->early_eof) { /* fall through to the stale-segment path below, as the reference does */ }
        } else {
            if (rdn(seg.data(), 2) != 2) break;
            if (seg[0] != 0xFF) {
                bool recovered = false;
                if (type == 0xFE) {
                    if (rdn(seg.data(), 1) != 1) break;
                    if (seg[0] == 0xFF) { recovered = true; jf->warn = std::max(jf->warn, 1); }
                }
                if (!recovered) { jf->error = "size mismatch in marker segment"; return EX_UNSUPPORTED_JPEG; }
            }
        }
        type = seg[1];
        if (type == 0xD9) { have_hdr = true; break; }
        if (rdn(seg.data() + 2, 2) != 2) break;
        unsigned len = 2 + be16(seg[2], seg[3]);
        if (len < 4) break;
        if (seg.size() < len) seg.resize(len);
        if (rdn(seg.data() + 4, len - 4) != (uint16_t)(len - 4)) break;
        // a slice keeps only the segments needed to decode the scan (is_needed_for_second_block, jpgcoder.cc:2242-2264, :2414)
        const bool needed = type == 0xC4 || type == 0xDB || type == 0xDD || type == 0xDA || type == 0xC0 || type == 0xC1 || type == 0xC2;
        if (jf->start_byte == 0 || needed) jf->hdr.insert(jf->hdr.end(), seg.begin(), seg.begin() + len);
    }
    if (!have_hdr || jf->hdr.empty()) { jf->error = "unexpected end of data encountered in header"; return EX_UNSUPPORTED_JPEG; }
    if (jf->scan.empty()) { jf->error = "unexpected end of data encountered in huffman"; return EX_UNSUPPORTED_JPEG; }
    jf->garbage.push_back(last2[0]);
    jf->garbage.push_back(last2[1]);
    jf->garbage.insert(jf->garbage.end(), d + pos, d + n);
    pos = n;
    if (jf->garbage.size() == 2 && jf->garbage[0] == 0xFF && jf->garbage[1] == 0xD9) jf->garbage.clear();
    jf->file_size = (uint32_t)n;
    return 0;
}

// ------------------------------------------------------------------------------------------------
static Handoff make_handoff(BitReader& br, const JpegFile& jf, int mcu_y, const int lastdc[4], int luma_mul) {
    const auto& offs = jf.scan_to_file;
    uint32_t p = (uint32_t)br.getpos();
    auto it = std::lower_bound(offs.begin(), offs.end(), std::pair<uint32_t, uint32_t>(p, p));
    if (it != offs.begin()) --it;
    uint32_t mapped = 0;
    if (it != offs.end()) mapped = it->second + (p - it->first);
    Handoff h;
    h.segment_size = mapped;
    for (int i = 0; i < 4; ++i) h.last_dc[i] = (int16_t)lastdc[i];
    h.luma_y_start = (uint16_t)(luma_mul * mcu_y);
    h.luma_y_end = (uint16_t)(luma_mul * (mcu_y + 1));
    br.overhang(&h.num_overhang_bits, &h.overhang_byte);
    return h;
}

// One block of a sequential scan into blk[0..63] (zigzag order, DC as a difference).  Returns the position behind the last
// coefficient the scan coded (1..64), -1 for bits that are no code, -2 for a zero run that leaves the block anywhere but at the end
// of the data (there the reference truncates the block and marks its last coefficient; elsewhere it asserts).
static int decode_block_seq(BitReader& br, const HuffTable& dc, const HuffTable& ac, int16_t* blk) {
    const int dc_category = next_huffcode(br, dc);
    if (dc_category < 0) return -1;
    memset(blk, 0, 64 * sizeof blk[0]);
    blk[0] = (int16_t)extend(dc_category & 0xff, (int)br.read(dc_category & 0xff));
    for (int at = 1; at < 64;) {
        const int sym = next_huffcode(br, ac);             // run of zeros << 4 | magnitude category
        if (sym < 0) return -1;
        if (sym == 0) return at;                           // end of block
        const int bits = (int)br.read(sym & 15);
        at += sym >> 4;
        if (at >= 64) {
            if (!br.eof) return -2;
            blk[63] = 1;
            return 64;
        }
        blk[at++] = (int16_t)extend(sym & 15, bits);
    }
    return 64;
}

// next block position, interleaved scan (recoder.cc:186-241)
int next_mcupos(const JpegFile& jf, int* mcu, int* cmp, int* csc, int* sub, int* dpos, int* rstw, int cs_cmpc) {
    int sta = 0;
    if (++(*sub) >= jf.comp[*cmp].mbs) {
        *sub = 0;
        if (++(*csc) >= cs_cmpc) {
            *csc = 0;
            *cmp = jf.cs_cmp[0];
            ++(*mcu);
            if (*mcu >= jf.mcuc) sta = 2;
            else if (jf.rsti > 0 && --(*rstw) == 0) sta = 1;
        } else *cmp = jf.cs_cmp[*csc];
    }
    const Component& k = jf.comp[*cmp];
    unsigned m = (unsigned)*mcu, sb = (unsigned)*sub, mh = (unsigned)jf.mcuh;
    if (k.vs > 1) *dpos = (int)(((m / mh) * k.vs + sb / k.hs) * k.bch + (m % mh) * k.hs + sb % k.hs);
    else if (k.hs > 1) *dpos = (int)(m * k.mbs + sb);
    else *dpos = *mcu;
    return sta;
}

// next block position, non-interleaved scan (jpgcoder.cc:5432-5456)
int next_mcuposn(const JpegFile& jf, int cmp, int* dpos, int* rstw) {
    const Component& k = jf.comp[cmp];
    ++(*dpos);
    if (k.bch != k.nch && (*dpos) % k.bch == k.nch) *dpos += k.bch - k.nch;
    if (k.bcv != k.ncv && (*dpos) / k.bch == k.ncv) *dpos = k.bc;
    if (*dpos >= k.bc) return 2;
    if (jf.rsti > 0 && --(*rstw) == 0) return 1;
    return 0;
}

static int min_vertical_multiple(const JpegFile& jf, int c) {   // uncompressed_components.cc:26-35
    return jf.comp[c].bcv / jf.mcuv;
}
// a file that ended inside its scan: how much of every component was decoded (uncompressed_components.hh:166-185), from max_dpos
static void note_truncation(JpegFile* jf) {
    for (int c = 0; c < jf->ncomp; ++c) {
        const Component& k = jf->comp[c];
        int tbc = jf->max_dpos[c] + 1;
        int lines = std::min(tbc / k.bch + (tbc % k.bch ? 1 : 0), k.bcv);
        int ratio = std::max(min_vertical_multiple(*jf, c), 1);
        while (lines % ratio != 0 && lines + 1 <= k.bcv) ++lines;
        jf->trunc_bcv[c] = lines;
        jf->trunc_bc[c] = tbc;
    }
}

int decode_progressive_scan(JpegFile* jf, BitReader& br, int* lastdc, int* sta_io, int* cmp_io, int* dpos_io,
                            int* mcu_io, int* csc_io, int* sub_io, int* rstw_io, unsigned* eobrun_io, int* peobrun_io,
                            bool* do_handoff);

// Step 2: Huffman-decode every scan into the coefficient planes.
static int decode_scans(JpegFile* jf, bool allow_progressive) {
    BitReader br(jf->scan.data(), (int)jf->scan.size());
    const uint8_t* h = jf->hdr.data();
    size_t hdrs = jf->hdr.size(), hpos = 0;
    int lastdc[4] = {0, 0, 0, 0};
    int mcu = 0;
    jf->place_frame(true);
    jf->scan_count = 0;
    const int luma_mul = jf->comp[0].bcv / jf->mcuv;
    int16_t blk[64];
    for (;;) {
        uint8_t type = 0;
        while (type != 0xDA) {
            if (3 + (uint64_t)hpos >= hdrs) break;
            type = h[hpos + 1];
            unsigned len = 2 + be16(h[hpos + 2], h[hpos + 3]);
            if (type == 0xC4 || type == 0xDA || type == 0xDD) {
                std::vector<uint8_t> over;
                const uint8_t* sp = h + hpos;
                if ((uint64_t)hpos + len > hdrs) { over.assign(h + hpos, h + hdrs); over.resize(len, 0); sp = over.data(); }
                if (!parse_segment(jf, type, len, len, sp, true)) return EX_UNSUPPORTED_JPEG;
            }
            hpos += len;
        }
        if (type != 0xDA) break;
        for (int i = 0; i < jf->cs_cmpc; ++i) {
            const Component& k = jf->comp[jf->cs_cmp[i]];
            bool need_dc = jf->jpegtype == 1 || ((jf->cs_cmpc > 1 || jf->cs_to == 0) && jf->cs_sah == 0);
            if ((need_dc && !jf->htab[0][k.dc_tbl].set) || (jf->jpegtype == 1 && !jf->htab[1][k.dc_tbl].set) ||
                (jf->cs_cmpc == 1 && jf->cs_to > 0 && jf->cs_sah == 0 && !jf->htab[1][k.ac_tbl].set)) {
                jf->error = "huffman table missing in scan";
                return EX_UNSUPPORTED_JPEG;
            }
        }
        int cmp = jf->cs_cmp[0], csc = 0, sub = 0, dpos = 0;
        mcu = 0;
        if (!br.eof) {
            jf->max_bpos = std::max(jf->max_bpos, jf->cs_to);
            jf->max_sah = std::max(jf->max_sah, std::max(jf->cs_sal, jf->cs_sah));
            for (int i = 0; i < jf->cs_cmpc; ++i) jf->max_cmp = std::max(jf->max_cmp, jf->cs_cmp[i]);
        }
        bool do_handoff = true;
        for (;;) {  // one restart interval per iteration
            lastdc[0] = lastdc[1] = lastdc[2] = lastdc[3] = 0;
            int sta = 0, rstw = jf->rsti;
            unsigned eobrun = 0;
            int peobrun = 0;
            if (jf->cs_cmpc != jf->ncomp || jf->jpegtype != 1) {
                if (!allow_progressive) return EX_PROGRESSIVE_UNSUPPORTED;
                jf->progressive_needed = true;
            }
            if (jf->jpegtype == 1 && jf->cs_cmpc > 1) {
                while (sta == 0) {
                    if (do_handoff) { jf->rows.push_back(make_handoff(br, *jf, mcu / jf->mcuh, lastdc, luma_mul)); do_handoff = false; }
                    if (!br.eof) jf->max_dpos[cmp] = std::max(dpos, jf->max_dpos[cmp]);
                    const Component& k = jf->comp[cmp];
                    int eob = decode_block_seq(br, jf->htab[0][k.dc_tbl], jf->htab[1][k.ac_tbl], blk);
                    if (eob == -2) return EX_ASSERTION_FAILURE;
                    if (eob > 1 && !blk[eob - 1]) jf->warn = std::max(jf->warn, 1);
                    blk[0] = (int16_t)(blk[0] + lastdc[cmp]);
                    lastdc[cmp] = blk[0];
                    int16_t* dst = jf->plane[cmp] + (size_t)dpos * 64;
                    for (int b = 0; b < eob; ++b) dst[kZigzagToAligned[b]] = blk[b];
                    int old_mcu = mcu;
                    if (eob < 0) sta = -1;
                    else sta = next_mcupos(*jf, &mcu, &cmp, &csc, &sub, &dpos, &rstw, jf->cs_cmpc);
                    if (mcu % jf->mcuh == 0 && old_mcu != mcu) do_handoff = true;
                    if (br.eof) { sta = 2; break; }
                }
            } else if (jf->jpegtype == 1) {
                const int vmul = jf->comp[0].bcv / jf->mcuv, hmul = jf->comp[0].bch / jf->mcuh;
                while (sta == 0) {
                    if (do_handoff) { jf->rows.push_back(make_handoff(br, *jf, (dpos / (hmul * vmul)) / jf->mcuh, lastdc, luma_mul)); do_handoff = false; }
                    if (!br.eof) jf->max_dpos[cmp] = std::max(dpos, jf->max_dpos[cmp]);
                    const Component& k = jf->comp[cmp];
                    int eob = decode_block_seq(br, jf->htab[0][k.dc_tbl], jf->htab[1][k.ac_tbl], blk);
                    if (eob == -2) return EX_ASSERTION_FAILURE;
                    if (eob > 1 && !blk[eob - 1]) jf->warn = std::max(jf->warn, 1);
                    blk[0] = (int16_t)(blk[0] + lastdc[cmp]);
                    lastdc[cmp] = blk[0];
                    int16_t* dst = jf->plane[cmp] + (size_t)dpos * 64;
                    for (int b = 0; b < eob; ++b) dst[kZigzagToAligned[b]] = blk[b];
                    if (eob < 0) sta = -1;
                    else sta = next_mcuposn(*jf, cmp, &dpos, &rstw);
                    mcu = dpos / (hmul * vmul);
                    if (cmp == 0 && (mcu % jf->mcuh == 0) && (dpos % (hmul * vmul) == 0)) do_handoff = true;
                    if (br.eof) { sta = 2; break; }
                }
            } else {
                int rc = decode_progressive_scan(jf, br, lastdc, &sta, &cmp, &dpos, &mcu, &csc, &sub, &rstw, &eobrun,
                                                 &peobrun, &do_handoff);
                if (rc) return rc;
            }
            if (jf->padbit != -1) {
                if (jf->padbit != (int)br.unpad((uint8_t)jf->padbit)) { jf->padbit = 1; jf->warn = std::max(jf->warn, 1); }
            } else {
                jf->padbit = (int8_t)br.unpad((uint8_t)jf->padbit);
            }
            if (sta == -1) { jf->error = "decode error in scan"; return EX_UNSUPPORTED_JPEG; }
            if (sta == 2) { ++jf->scan_count; break; }
        }
    }
    if (jf->early_eof) note_truncation(jf);
    jf->rows.push_back(make_handoff(br, *jf, (uint16_t)(mcu / jf->mcuh), lastdc, luma_mul));
    for (size_t i = 1; i < jf->rows.size(); ++i)
        if (jf->rows[i].luma_y_start < jf->rows[i - 1].luma_y_end) jf->rows[i].luma_y_start = jf->rows[i - 1].luma_y_end;
    if (!br.eof) jf->warn = std::max(jf->warn, 1);  // "unneeded data found after coded image data"
    return 0;
}

Handoff make_handoff_public(BitReader& br, const JpegFile& jf, int mcu_y, const int lastdc[4], int luma_mul) {
    return make_handoff(br, jf, mcu_y, lastdc, luma_mul);
}

// the GPU decoders' view of one Huffman table: 9-bit first-level LUT + the one-step canonical test for codes of 9..16 bits.
// That test equals the code tree only if the table IS canonical (Annex C: codes of one length consecutive, the next length
// continues at (last + 1) << 1, no overflow); false = a DHT that breaks this: the file stays with the host parser.
static bool fill_decode_tables(const HuffTable& t, uint16_t* lut, int32_t* maxcode, int32_t* valoff, uint8_t* longsym) {
    for (int sym = 0; sym < 256; ++sym) {
        const int len = t.clen[sym];
        if (len < 1 || len > 9) continue;
        const unsigned first = (unsigned)t.cval[sym] << (9 - len);
        for (unsigned x = 0; x < (1u << (9 - len)); ++x) lut[first + x] = (uint16_t)((len << 8) | sym);
    }
    int order[256], no = 0;
    for (int len = 1; len <= 16; ++len)
        for (int sym = 0; sym < 256; ++sym)
            if (t.clen[sym] == len) order[no++] = sym;
    std::stable_sort(order, order + no, [&](int a, int b) { return t.clen[a] != t.clen[b] ? t.clen[a] < t.clen[b] : t.cval[a] < t.cval[b]; });
    unsigned next = 0;
    int prev_len = no ? t.clen[order[0]] : 0, nlong = 0;
    for (int k = 0; k < 8; ++k) { maxcode[k] = -1; valoff[k] = 0; }
    for (int q = 0; q < no; ++q) {
        const int sym = order[q], len = t.clen[sym];
        next <<= (len - prev_len);
        prev_len = len;
        if (t.cval[sym] != next || next >= (1u << len)) return false;
        if (len >= 9) {
            if (maxcode[len - 9] < 0) valoff[len - 9] = nlong - (int)next;
            maxcode[len - 9] = (int32_t)next;
            longsym[nlong++] = (uint8_t)sym;
        }
        ++next;
    }
    return true;
}

// ---- GPU Huffman decode (lep_huffdec.h): the host does everything except decoding the scan ---------------------------------------
// parse_jpeg_prepare_gpu: split the file, set up the frame, read the tables of the first scan; *eligible when the scan is a
// single MCU-interleaved sequential scan of all (2..3) components with at most two tables per class and the file is whole.
int parse_jpeg_prepare_gpu(const uint8_t* data, size_t size, JpegFile* jf, ScanDecodePlan* plan, bool* eligible) {
    *eligible = false;
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) { jf->error = "not a jpeg"; return EX_UNSUPPORTED_JPEG; }
    memset(jf->qtables, 0, sizeof jf->qtables);
    int rc = split_file(data, size, jf);
    if (rc) return rc;
    if (!setup_frame(jf)) return jf->warn < 0 ? -jf->warn : EX_UNSUPPORTED_JPEG;
    if (jf->ncomp > 3) return EX_UNSUPPORTED_4_COLORS;
    if (jf->jpegtype != 1) return 0;
    // (a file that ends inside its scan -- no EOI -- is eligible: the kernels decode up to the block that reads the data's last bit and
    // say how far they came; anything odd in that last block is left to the host parser, which knows the reference's rules there)
    const uint8_t* h = jf->hdr.data();
    const size_t hdrs = jf->hdr.size();
    size_t hpos = 0;
    int nsos = 0;
    while (3 + (uint64_t)hpos < hdrs) {
        const uint8_t type = h[hpos + 1];
        const unsigned len = 2 + be16(h[hpos + 2], h[hpos + 3]);
        if ((uint64_t)hpos + len > hdrs) return 0;             // truncated segment: host parser
        if (type == 0xDA) ++nsos;
        if (nsos <= 1 && (type == 0xC4 || type == 0xDA || type == 0xDD))
            if (!parse_segment(jf, type, len, len, h + hpos, true)) return 0;
        if (nsos > 1) return 0;
        hpos += len;
    }
    if (nsos != 1 || jf->cs_cmpc != jf->ncomp || jf->scan.empty()) return 0;
    memset(plan, 0, sizeof *plan);
    plan->scan_len = (uint32_t)jf->scan.size();
    plan->ncomp = jf->ncomp; plan->mcuh = jf->mcuh; plan->mcuv = jf->mcuv; plan->mcuc = jf->mcuc; plan->rsti = jf->rsti;
    plan->flags = jf->early_eof ? kScanEarlyEof : 0;
    if (jf->early_eof && jf->rsti) return 0;   // (restart intervals in a cut file: the single-wave kernel has no notion of the cut)
    // One component: the scan is not interleaved whatever the sampling factors say -- its MCU is one block and it walks the nch x ncv
    // blocks the picture covers, stepping over the blocks that pad the frame to whole MCUs (next_mcuposn; jpgcoder.cc:3135-3175 with
    // :3855-3890).  To the kernels that is a frame of nch x ncv MCUs of one block whose block rows lie bch blocks apart: they leave one
    // row record per BLOCK row, of which parse_jpeg_finish_gpu takes every (bcv / mcuv)-th.  The plain case (factors 1x1) is the same plan
    // as ever.  Cut files and restart intervals of the other cases stay with the host parser (a restart interval counts blocks in the
    // scan and MCUs of hs x vs blocks in the re-coder: the reference cannot restore such a file, tests/test_sampling_layouts.py).
    const bool planar = jf->ncomp == 1;
    if (planar) {
        const Component& k = jf->comp[jf->cs_cmp[0]];
        const bool plain = k.hs == 1 && k.vs == 1 && k.bch == k.nch && k.bcv == k.ncv && k.bc == jf->mcuc;
        if (!plain) {
            if (jf->early_eof || jf->rsti || jf->cs_cmp[0] != 0 || k.nch < 1 || k.ncv < 1 || k.nch > k.bch || k.ncv > k.bcv || jf->mcuv < 1 || k.bcv % jf->mcuv) return 0;
            plan->mcuh = k.nch; plan->mcuv = k.ncv; plan->mcuc = k.nch * k.ncv;
        }
    }
    // restart intervals: when the scan holds exactly the markers its length asks for -- one behind every rsti MCUs but the last run --
    // their positions travel with the scan bytes and every interval is decoded by a lane of its own (lep_huffdec_simt.h); otherwise the
    // single-wave kernel walks the scan as the reference does
    if (jf->rsti > 0 && jf->mcuc > 0) {
        const size_t want = (size_t)((jf->mcuc - 1) / jf->rsti);
        bool ok = jf->rst_pos.size() == want && (jf->rst_cnt.empty() ? want == 0 : jf->rst_cnt[0] == want) && (jf->rst_err.empty() || jf->rst_err[0] == 0);
        for (size_t q = 0; ok && q < want; ++q) ok = jf->rst_pos[q] <= jf->scan.size() && (q == 0 || jf->rst_pos[q] >= jf->rst_pos[q - 1]);
        if (ok && want > 0) plan->flags |= kScanRstTable;
    }
    for (int i = 0; i < jf->ncomp; ++i) {
        const Component& k = jf->comp[i];
        if (k.dc_tbl > 1 || k.ac_tbl > 1 || !jf->htab[0][k.dc_tbl].set || !jf->htab[1][k.ac_tbl].set) return 0;
        if (k.bch != jf->mcuh * k.hs || k.bcv != jf->mcuv * k.vs || k.hs < 1 || k.vs < 1) return 0;
        plan->hs[i] = planar ? 1 : k.hs; plan->vs[i] = planar ? 1 : k.vs; plan->bch[i] = k.bch; plan->dc_tbl[i] = k.dc_tbl; plan->ac_tbl[i] = k.ac_tbl;
        plan->scan_cmp[i] = jf->cs_cmp[i];
    }
    for (int cls = 0; cls < 2; ++cls)
        for (int id = 0; id < 2; ++id) {
            const HuffTable& t = jf->htab[cls][id];
            if (!t.set) continue;
            if (!fill_decode_tables(t, plan->lut[cls * 2 + id], plan->maxcode[cls * 2 + id], plan->valoff[cls * 2 + id], plan->longsym[cls * 2 + id])) return 0;
        }
    *eligible = true;
    return 0;
}

// parse_jpeg_finish_gpu: what decode_scans leaves behind besides the coefficients -- hand-off records from the bit
// positions the kernel recorded per MCU row, the pad-bit pattern, the scan bookkeeping.  A non-zero kernel status (or a
// record that makes no sense) is returned as -1: the caller re-parses the file on the host.
int parse_jpeg_finish_gpu(JpegFile* jf, const ScanDecodeRow* all_rows) {
    // (one component: the kernels' frame is nch x ncv MCUs of one block, parse_jpeg_prepare_gpu -- a record per block row, of which
    // every luma_mul-th starts an MCU row of the file's; the final record stands behind the ncv-th)
    const int mcuv = jf->mcuv;
    const int luma_mul = jf->comp[0].bcv / jf->mcuv;
    const bool planar = jf->ncomp == 1;
    const int step = planar ? std::max(luma_mul, 1) : 1;
    const int nrec = planar ? jf->comp[0].ncv : mcuv;
    std::vector<ScanDecodeRow> picked;
    const ScanDecodeRow* rows = all_rows;
    if (planar && (step != 1 || nrec != mcuv)) {
        if ((mcuv - 1) * step >= nrec) return -1;
        picked.resize((size_t)mcuv + 1);
        for (int r = 0; r < mcuv; ++r) picked[(size_t)r] = all_rows[r * step];
        picked[(size_t)mcuv] = all_rows[nrec];
        rows = picked.data();
        if (all_rows[nrec].aux & kScanRowTruncated) return -1;   // (cut files of this kind never get here)
    }
    const int status = (rows[mcuv].aux >> 8) & 0x3fffff;
    if (status) return -1;
    const bool truncated = (rows[mcuv].aux & kScanRowTruncated) != 0;
    if (truncated && !jf->early_eof) return -1;
    const uint32_t total_bits = (uint32_t)jf->scan.size() * 8u;
    // blocks of an MCU in scan order; a truncated scan: how many MCU rows were entered, and where the walk stood when the data ended
    int nphase = 0;
    for (int ci = 0; ci < jf->cs_cmpc; ++ci) nphase += jf->comp[jf->cs_cmp[ci]].hs * jf->comp[jf->cs_cmp[ci]].vs;
    if (nphase < 1 || jf->mcuh < 1) return -1;
    const uint32_t done = truncated ? rows[mcuv].bitpos : (uint32_t)jf->mcuc * (uint32_t)nphase;   // blocks decoded
    if (truncated && (done == 0 || done >= (uint32_t)jf->mcuc * (uint32_t)nphase)) return -1;
    const uint32_t mcu_last = (done - 1) / (uint32_t)nphase;                    // MCU of the last decoded block
    const uint32_t mcu_after = done / (uint32_t)nphase;                         // where next_mcupos left `mcu`
    const int rows_entered = truncated ? (int)(mcu_last / (uint32_t)jf->mcuh) + 1 : mcuv;
    jf->rows.clear();
    const auto& offs = jf->scan_to_file;
    auto handoff_at = [&](uint32_t bp, const int16_t* last_dc, int mcu_y) {
        const uint32_t p = (bp >> 3) + 1;   // BitReader::getpos: 1 + index of the byte holding the next unread bit
        auto it = std::lower_bound(offs.begin(), offs.end(), std::pair<uint32_t, uint32_t>(p, p));
        if (it != offs.begin()) --it;
        uint32_t mapped = 0;
        if (it != offs.end()) mapped = it->second + (p - it->first);
        Handoff hnd;
        hnd.segment_size = mapped;
        for (int i = 0; i < 4; ++i) hnd.last_dc[i] = last_dc[i];
        hnd.luma_y_start = (uint16_t)(luma_mul * mcu_y);
        hnd.luma_y_end = (uint16_t)(luma_mul * (mcu_y + 1));
        const int rem = (int)(bp & 7u);
        hnd.num_overhang_bits = (uint8_t)rem;
        hnd.overhang_byte = rem ? (uint8_t)(jf->scan[bp >> 3] & (uint8_t)(((1 << rem) - 1) << (8 - rem))) : 0;
        jf->rows.push_back(hnd);
    };
    for (int r = 0; r < rows_entered; ++r) {
        if (rows[r].bitpos > total_bits) return -1;
        handoff_at(rows[r].bitpos, rows[r].last_dc, r);
    }
    if (truncated) {
        // the reader stands at the end of the data (every bit read: getpos = size + 1, nothing overhanging); the record is the one
        // decode_scans appends behind its loop, for the MCU row the walk had reached
        handoff_at(total_bits, rows[mcuv].last_dc, (int)(mcu_after / (uint32_t)jf->mcuh));
    } else {
        if (rows[mcuv].bitpos > total_bits) return -1;
        handoff_at(rows[mcuv].bitpos, rows[mcuv].last_dc, mcuv);
    }
    for (size_t i = 1; i < jf->rows.size(); ++i)
        if (jf->rows[i].luma_y_start < jf->rows[i - 1].luma_y_end) jf->rows[i].luma_y_start = jf->rows[i - 1].luma_y_end;
    // entropy-coded bytes left over behind the last MCU and its padding ("unneeded data found after coded image data",
    // jpgcoder.cc:3290): the host parser refuses the file as the reference does
    if (!truncated && rows[mcuv].bitpos != total_bits) return -1;
    jf->padbit = (int8_t)(rows[mcuv].aux & 255);
    jf->scan_count = 1;
    jf->max_bpos = std::max(jf->max_bpos, jf->cs_to);
    jf->max_sah = std::max(jf->max_sah, std::max(jf->cs_sal, jf->cs_sah));
    for (int i = 0; i < jf->cs_cmpc; ++i) jf->max_cmp = std::max(jf->max_cmp, jf->cs_cmp[i]);
    for (int c = 0; c < jf->ncomp; ++c) jf->max_dpos[c] = jf->comp[c].bc - 1;
    if (planar) jf->max_dpos[0] = (jf->comp[0].ncv - 1) * jf->comp[0].bch + jf->comp[0].nch - 1;   // the last block the scan codes: padding blocks are stepped over
    if (truncated) {
        // max_dpos: the largest block position of every component among the decoded blocks (decode_scans notes it at the start of every
        // block): all of the MCU rows in front of the last one entered, and of that row what its MCUs up to the last block hold
        const int row = (int)(mcu_last / (uint32_t)jf->mcuh);
        for (int c = 0; c < jf->ncomp; ++c) jf->max_dpos[c] = row > 0 ? row * jf->comp[c].vs * jf->comp[c].bch - 1 : 0;
        const uint32_t first = (uint32_t)row * (uint32_t)jf->mcuh * (uint32_t)nphase;
        for (uint32_t b = first; b < done; ++b) {
            const int mx = (int)((b / (uint32_t)nphase) % (uint32_t)jf->mcuh);
            int ph = (int)(b % (uint32_t)nphase);
            for (int ci = 0; ci < jf->cs_cmpc; ++ci) {
                const int c = jf->cs_cmp[ci];
                const Component& k = jf->comp[c];
                if (ph < k.hs * k.vs) {
                    const int dpos = (row * k.vs + ph / k.hs) * k.bch + mx * k.hs + ph % k.hs;
                    jf->max_dpos[c] = std::max(jf->max_dpos[c], dpos);
                    break;
                }
                ph -= k.hs * k.vs;
            }
        }
    }
    if (jf->early_eof) note_truncation(jf);
    jf->progressive_needed = false;
    return 0;
}

// ---- progressive files on the GPU scan decoder (lep_huffprogdec.h) ------------------------------------------------------------------
// Called after parse_jpeg_prepare_gpu (which split the file and set up the frame) found the file not eligible for the
// sequential kernel.  Eligible here: whole progressive frames whose FIRST scan is the one DC first-stage scan of all
// components (what libjpeg writes), every other scan a DC refinement or a single-component AC scan, canonical tables, one
// restart interval for the whole file.  Everything else -- and anything the kernels then find irregular -- is the host's.
// SEQUENTIAL frames coded in several scans -- luma alone, then Cb + Cr together; a scan per component (`jpegtran -scans`, some scanners).
// To the reference they are format 'X' like progressive files (cs_cmpc != cmpc: jpgcoder.cc:3009-3014) but their scans are decoded by the
// sequential block loop (:3034-3175), each scan leaving hand-off records of its own.  Here: every scan one descriptor with from 0 / to 63
// (lep_huffprogdec.h sequential_scan_image), t = the frame with all four tables, a record per MCU row of the SCAN's geometry -- the frame's
// MCU rows for a scan of several components, the component's ncv block rows for a scan of one.  Eligible: whole files, two or three
// components each coded by exactly one scan, tables 0 / 1, canonical codes.
static int prepare_gpu_sequential_scans(JpegFile* jf, std::vector<ProgScanDecodePlan>* scans, int* rows_needed, bool* eligible) {
    if (jf->early_eof || jf->ncomp < 2 || jf->ncomp > 3 || jf->scan.empty() || jf->start_byte || jf->mcuh < 1 || jf->mcuv < 1) return 0;
    const uint8_t* h = jf->hdr.data();
    const size_t hdrs = jf->hdr.size();
    size_t hpos = 0;
    unsigned coded = 0;
    uint64_t next_row = 0;
    while (3 + (uint64_t)hpos < hdrs) {
        const uint8_t type = h[hpos + 1];
        const unsigned len = 2 + be16(h[hpos + 2], h[hpos + 3]);
        if ((uint64_t)hpos + len > hdrs) return 0;
        if (type == 0xC4 || type == 0xDA || type == 0xDD)
            if (!parse_segment(jf, type, len, len, h + hpos, true)) return 0;
        hpos += len;
        if (type != 0xDA) continue;
        const size_t k = scans->size();
        if (k >= jf->scan_start.size() || k >= 4) return 0;
        ProgScanDecodePlan sc;
        memset(&sc, 0, sizeof sc);
        sc.cmpc = jf->cs_cmpc; sc.from = 0; sc.to = 63;
        if (jf->cs_from != 0 || jf->cs_to != 63 || jf->cs_sah != 0 || jf->cs_sal != 0 || sc.cmpc < 1 || sc.cmpc > jf->ncomp) return 0;
        ScanDecodePlan& t = sc.t;
        t.scan = (const uint8_t*)(uintptr_t)jf->scan_start[k];
        const size_t end = k + 1 < jf->scan_start.size() ? jf->scan_start[k + 1] : jf->scan.size();
        if (end <= jf->scan_start[k]) return 0;
        t.scan_len = (uint32_t)(end - jf->scan_start[k]);
        t.ncomp = jf->ncomp; t.mcuh = jf->mcuh; t.mcuv = jf->mcuv; t.mcuc = jf->mcuc; t.rsti = jf->rsti;
        for (int c = 0; c < jf->ncomp; ++c) {
            const Component& q = jf->comp[c];
            if (q.hs < 1 || q.vs < 1 || q.nch < 1 || q.ncv < 1 || q.nch > q.bch || q.ncv > q.bcv || q.bch != jf->mcuh * q.hs || q.bcv != jf->mcuv * q.vs) return 0;
            t.hs[c] = q.hs; t.vs[c] = q.vs; t.bch[c] = q.bch; t.dc_tbl[c] = q.dc_tbl; t.ac_tbl[c] = q.ac_tbl;
            sc.bcv[c] = q.bcv; sc.nch[c] = q.nch; sc.ncv[c] = q.ncv; sc.mbs[c] = q.mbs;
        }
        for (int i = 0; i < sc.cmpc; ++i) {
            const int c = jf->cs_cmp[i];
            if (c < 0 || c >= jf->ncomp || (coded >> c & 1u)) return 0;            // (a component coded twice: the host parser's)
            coded |= 1u << c;
            sc.cmp[i] = c; t.scan_cmp[i] = c;
            const Component& q = jf->comp[c];
            if (q.dc_tbl < 0 || q.dc_tbl > 1 || q.ac_tbl < 0 || q.ac_tbl > 1 || !jf->htab[0][q.dc_tbl].set || !jf->htab[1][q.ac_tbl].set) return 0;
            if (!fill_decode_tables(jf->htab[0][q.dc_tbl], t.lut[q.dc_tbl], t.maxcode[q.dc_tbl], t.valoff[q.dc_tbl], t.longsym[q.dc_tbl])) return 0;
            if (!fill_decode_tables(jf->htab[1][q.ac_tbl], t.lut[2 + q.ac_tbl], t.maxcode[2 + q.ac_tbl], t.valoff[2 + q.ac_tbl], t.longsym[2 + q.ac_tbl])) return 0;
        }
        const int scan_rows = sc.cmpc > 1 ? jf->mcuv : jf->comp[sc.cmp[0]].ncv;
        sc.level = 0; sc.want_rows = 1;
        t.rows_off = next_row;
        sc.result_off = next_row + (uint64_t)scan_rows;
        next_row += (uint64_t)scan_rows + 1;
        scans->push_back(sc);
    }
    if (scans->size() < 2 || scans->size() != jf->scan_start.size() || coded != (1u << jf->ncomp) - 1u || next_row > 0x7fffffffu) { scans->clear(); return 0; }
    *rows_needed = (int)next_row;
    *eligible = true;
    return 0;
}

// what decode_scans leaves behind for such a file (jpeg_scan.cc decode_scans, the two jpegtype == 1 loops): a hand-off record in front of
// the first block of every scan, and from then on in front of the first block of every MCU row -- in a scan of several components; in a
// scan of one component only if it is component 0 (`cmp == 0 && mcu % mcuh == 0 && dpos % (hmul * vmul) == 0`, mcu = dpos / (hmul *
// vmul) with LUMA's factors) -- and the record behind the last scan, whose MCU row is the one the last scan's walk left `mcu` at.
static int finish_gpu_sequential_scans(JpegFile* jf, const std::vector<ProgScanDecodePlan>& scans, const ScanDecodeRow* rows) {
    if (jf->mcuv < 1 || jf->mcuh < 1 || scans.size() > jf->scan_start.size()) return -1;
    const int luma_mul = jf->comp[0].bcv / jf->mcuv, hmul = jf->comp[0].bch / jf->mcuh;
    if (luma_mul < 1 || hmul < 1) return -1;
    for (const ProgScanDecodePlan& sc : scans) {           // (descriptors come back through the public entry: hold them to what prepare made)
        if (sc.cmpc < 1 || sc.cmpc > jf->ncomp) return -1;
        for (int i = 0; i < sc.cmpc; ++i) if (sc.cmp[i] < 0 || sc.cmp[i] >= jf->ncomp) return -1;
    }
    int padbit = -1;
    jf->max_bpos = 0; jf->max_sah = 0; jf->max_cmp = 0;
    for (const ProgScanDecodePlan& sc : scans) {
        const ScanDecodeRow& fin = rows[sc.result_off];
        if (fin.aux == (int32_t)0x80000000 || ((fin.aux >> 8) & 0x3fffff) || (fin.aux & kScanRowTruncated)) return -1;   // not written / irregular / ran out of data
        if (fin.bitpos != sc.t.scan_len * 8u) return -1;                       // bytes left over behind the scan's last MCU
        const int pb = (int8_t)(fin.aux & 255);
        if (pb != -1) { if (padbit == -1) padbit = pb; else if (padbit != pb) return -1; }   // "inconsistent use of padbits"
        jf->max_bpos = 63;
        for (int i = 0; i < sc.cmpc; ++i) jf->max_cmp = std::max(jf->max_cmp, sc.cmp[i]);
    }
    jf->rows.clear();
    const auto& offs = jf->scan_to_file;
    auto record = [&](uint32_t bit_in_all, const int16_t* last_dc, int mcu_y) {
        const uint32_t p = (bit_in_all >> 3) + 1;   // BitReader::getpos
        auto it = std::lower_bound(offs.begin(), offs.end(), std::pair<uint32_t, uint32_t>(p, p));
        if (it != offs.begin()) --it;
        uint32_t mapped = 0;
        if (it != offs.end()) mapped = it->second + (p - it->first);
        Handoff hnd;
        hnd.segment_size = mapped;
        for (int i = 0; i < 4; ++i) hnd.last_dc[i] = last_dc[i];
        hnd.luma_y_start = (uint16_t)(luma_mul * mcu_y);
        hnd.luma_y_end = (uint16_t)(luma_mul * (mcu_y + 1));
        const int rem = (int)(bit_in_all & 7u);
        hnd.num_overhang_bits = (uint8_t)rem;
        hnd.overhang_byte = rem ? (uint8_t)(jf->scan[bit_in_all >> 3] & (uint8_t)(((1 << rem) - 1) << (8 - rem))) : 0;
        jf->rows.push_back(hnd);
    };
    int last_mcu = 0;
    for (size_t k = 0; k < scans.size(); ++k) {
        const ProgScanDecodePlan& sc = scans[k];
        const uint32_t base = jf->scan_start[k] * 8u;
        const ScanDecodeRow* r = rows + sc.t.rows_off;
        if (sc.cmpc > 1) {
            for (int y = 0; y < jf->mcuv; ++y) { if (r[y].bitpos > sc.t.scan_len * 8u) return -1; record(base + r[y].bitpos, r[y].last_dc, y); }
            last_mcu = jf->mcuc;
        } else {
            const Component& q = jf->comp[sc.cmp[0]];
            if (sc.cmp[0] == 0) {
                for (int y = 0; y < jf->mcuv; ++y) {
                    if (y * luma_mul >= q.ncv) return -1;
                    const ScanDecodeRow& e = r[y * luma_mul];
                    if (e.bitpos > sc.t.scan_len * 8u) return -1;
                    record(base + e.bitpos, e.last_dc, y);
                }
            } else {
                record(base + r[0].bitpos, r[0].last_dc, 0);
            }
            last_mcu = q.bc / (hmul * luma_mul);
        }
    }
    const ScanDecodeRow& fin = rows[scans.back().result_off];
    record((uint32_t)jf->scan.size() * 8u, fin.last_dc, last_mcu / jf->mcuh);
    for (size_t i = 1; i < jf->rows.size(); ++i)
        if (jf->rows[i].luma_y_start < jf->rows[i - 1].luma_y_end) jf->rows[i].luma_y_start = jf->rows[i - 1].luma_y_end;
    jf->padbit = (int8_t)padbit;
    jf->scan_count = (int)scans.size();
    for (int c = 0; c < jf->ncomp; ++c) jf->max_dpos[c] = jf->comp[c].bc - 1;
    for (const ProgScanDecodePlan& sc : scans)
        if (sc.cmpc == 1) { const Component& q = jf->comp[sc.cmp[0]]; jf->max_dpos[sc.cmp[0]] = (q.ncv - 1) * q.bch + q.nch - 1; }
    jf->progressive_needed = true;
    return 0;
}

int parse_jpeg_prepare_gpu_progressive(JpegFile* jf, std::vector<ProgScanDecodePlan>* scans, int* rows_needed, bool* eligible) {
    *eligible = false;
    scans->clear();
    if (jf->jpegtype == 1) return prepare_gpu_sequential_scans(jf, scans, rows_needed, eligible);
    if (jf->early_eof || jf->jpegtype != 2 || jf->ncomp < 1 || jf->ncomp > 3 || jf->scan.empty() || jf->start_byte) return 0;
    const uint8_t* h = jf->hdr.data();
    const size_t hdrs = jf->hdr.size();
    size_t hpos = 0;
    int rsti_seen = -1;
    const int nrows = std::max(jf->mcuv, jf->comp[0].bcv);
    while (3 + (uint64_t)hpos < hdrs) {
        const uint8_t type = h[hpos + 1];
        const unsigned len = 2 + be16(h[hpos + 2], h[hpos + 3]);
        if ((uint64_t)hpos + len > hdrs) return 0;
        if (type == 0xC4 || type == 0xDA || type == 0xDD)
            if (!parse_segment(jf, type, len, len, h + hpos, true)) return 0;
        hpos += len;
        if (type != 0xDA) continue;
        const size_t k = scans->size();
        if (k >= jf->scan_start.size() || k >= 256) return 0;
        rsti_seen = jf->rsti;                                  // (it may change from scan to scan: every descriptor carries its own, t.rsti)
        ProgScanDecodePlan sc;
        memset(&sc, 0, sizeof sc);
        sc.cmpc = jf->cs_cmpc; sc.from = jf->cs_from; sc.to = jf->cs_to; sc.sah = jf->cs_sah; sc.sal = jf->cs_sal;
        if (sc.cmpc < 1 || sc.cmpc > jf->ncomp || sc.sal < 0 || sc.sal > 13 || sc.sah < 0 || sc.sah > 13 || sc.from < 0 || sc.to > 63 || sc.from > sc.to) return 0;
        const bool dc = sc.to == 0;
        if (dc && sc.from != 0) return 0;
        if (!dc && (sc.cmpc != 1 || sc.from < 1)) return 0;
        if (k == 0 && !(dc && sc.sah == 0 && sc.cmpc == jf->ncomp)) return 0;   // the hand-off rows come from this scan
        if (k > 0 && dc && sc.sah == 0) return 0;                               // a second DC first-stage scan: host
        if (sc.sah != 0 && sc.sah != sc.sal + 1) return 0;
        for (int i = 0; i < sc.cmpc; ++i) {
            const int c = jf->cs_cmp[i];
            if (c < 0 || c >= jf->ncomp) return 0;
            for (int j = 0; j < i; ++j) if (sc.cmp[j] == c) return 0;
            sc.cmp[i] = c;
        }
        if (sc.cmpc > 1) for (int i = 0; i < sc.cmpc; ++i) if (sc.cmp[i] != i) return 0;   // interleaved scans in frame order
        ScanDecodePlan& t = sc.t;
        t.scan = (const uint8_t*)(uintptr_t)jf->scan_start[k];
        const size_t end = k + 1 < jf->scan_start.size() ? jf->scan_start[k + 1] : jf->scan.size();
        if (end <= jf->scan_start[k]) return 0;
        t.scan_len = (uint32_t)(end - jf->scan_start[k]);
        t.ncomp = jf->ncomp; t.mcuh = jf->mcuh; t.mcuv = jf->mcuv; t.mcuc = jf->mcuc; t.rsti = jf->rsti;
        for (int c = 0; c < jf->ncomp; ++c) {
            const Component& q = jf->comp[c];
            if (q.hs < 1 || q.vs < 1 || q.nch < 1 || q.ncv < 1 || q.bch != jf->mcuh * q.hs || q.bcv != jf->mcuv * q.vs) return 0;
            if (jf->ncomp == 1 && (q.bch != q.nch || q.bcv != q.ncv)) return 0;
            t.hs[c] = q.hs; t.vs[c] = q.vs; t.bch[c] = q.bch;
            sc.bcv[c] = q.bcv; sc.nch[c] = q.nch; sc.ncv[c] = q.ncv; sc.mbs[c] = q.mbs;
        }
        if (dc) {
            for (int i = 0; i < sc.cmpc; ++i) {
                const int id = jf->comp[sc.cmp[i]].dc_tbl;
                if (id < 0 || id > 1) return 0;
                sc.tbl[i] = id;
                if (sc.sah == 0) {
                    if (!jf->htab[0][id].set) return 0;
                    if (!fill_decode_tables(jf->htab[0][id], t.lut[id], t.maxcode[id], t.valoff[id], t.longsym[id])) return 0;
                }
            }
        } else {
            const int id = jf->comp[sc.cmp[0]].ac_tbl;
            if (id < 0 || id > 3 || !jf->htab[1][id].set) return 0;
            if (!fill_decode_tables(jf->htab[1][id], t.lut[2], t.maxcode[2], t.valoff[2], t.longsym[2])) return 0;
            sc.max_eobrun = jf->htab[1][id].max_eobrun;
        }
        // dependency level: behind every earlier scan of the same component whose band overlaps
        int level = 0;
        for (const ProgScanDecodePlan& e : *scans) {
            bool shares = false;
            for (int i = 0; i < sc.cmpc; ++i) for (int j = 0; j < e.cmpc; ++j) shares |= sc.cmp[i] == e.cmp[j];
            if (shares && e.from <= sc.to && sc.from <= e.to) level = std::max(level, e.level + 1);
        }
        sc.level = level;
        sc.want_rows = k == 0 ? 1 : 0;
        t.rows_off = 0;
        sc.result_off = (uint64_t)(nrows + 1) + k;
        scans->push_back(sc);
    }
    if (scans->empty() || scans->size() != jf->scan_start.size()) return 0;
    *rows_needed = nrows + 1 + (int)scans->size();
    *eligible = true;
    return 0;
}

static std::atomic<uint64_t> g_prog_wait_timeouts{0};   // scans of the pipelined launch that gave up waiting for a scan in front of them (status 4)
uint64_t prog_wait_timeouts() { return g_prog_wait_timeouts.load(std::memory_order_relaxed); }

int parse_jpeg_finish_gpu_progressive(JpegFile* jf, const std::vector<ProgScanDecodePlan>& scans, const ScanDecodeRow* rows) {
    if (!scans.empty() && scans[0].from == 0 && scans[0].to == 63) return finish_gpu_sequential_scans(jf, scans, rows);
    const int nrows = std::max(jf->mcuv, jf->comp[0].bcv);
    const int luma_mul = jf->comp[0].bcv / jf->mcuv;
    int padbit = -1;
    jf->max_bpos = 0; jf->max_sah = 0; jf->max_cmp = 0;
    for (size_t k = 0; k < scans.size(); ++k) {
        const ScanDecodeRow& fin = rows[scans[k].result_off];
        if ((fin.aux >> 8) == 4) g_prog_wait_timeouts.fetch_add(1, std::memory_order_relaxed);
        if (fin.aux >> 8) return -1;                                       // irregular somewhere in this scan
        if (fin.bitpos != scans[k].t.scan_len * 8u) return -1;
        const int pb = (int8_t)(fin.aux & 255);
        if (pb != -1) { if (padbit == -1) padbit = pb; else if (padbit != pb) return -1; }   // "inconsistent use of padbits"
        jf->max_bpos = std::max(jf->max_bpos, scans[k].to);
        jf->max_sah = std::max(jf->max_sah, std::max(scans[k].sal, scans[k].sah));
        for (int i = 0; i < scans[k].cmpc; ++i) jf->max_cmp = std::max(jf->max_cmp, scans[k].cmp[i]);
    }
    const ProgScanDecodePlan& first = scans[0];
    const int row_count = first.cmpc > 1 ? jf->mcuv : jf->comp[0].ncv;   // rows the first scan started
    jf->rows.clear();
    const auto& offs = jf->scan_to_file;
    auto record = [&](uint32_t bit_in_all, const int16_t* last_dc, int mcu_y) {
        const uint32_t p = (bit_in_all >> 3) + 1;   // BitReader::getpos
        auto it = std::lower_bound(offs.begin(), offs.end(), std::pair<uint32_t, uint32_t>(p, p));
        if (it != offs.begin()) --it;
        uint32_t mapped = 0;
        if (it != offs.end()) mapped = it->second + (p - it->first);
        Handoff hnd;
        hnd.segment_size = mapped;
        for (int i = 0; i < 4; ++i) hnd.last_dc[i] = last_dc[i];
        hnd.luma_y_start = (uint16_t)(luma_mul * mcu_y);
        hnd.luma_y_end = (uint16_t)(luma_mul * (mcu_y + 1));
        const int rem = (int)(bit_in_all & 7u);
        hnd.num_overhang_bits = (uint8_t)rem;
        hnd.overhang_byte = rem ? (uint8_t)(jf->scan[bit_in_all >> 3] & (uint8_t)(((1 << rem) - 1) << (8 - rem))) : 0;
        jf->rows.push_back(hnd);
    };
    const uint32_t base0 = jf->scan_start.empty() ? 0u : jf->scan_start[0] * 8u;   // (the descriptors' scan pointers are the caller's device addresses by now)
    for (int r = 0; r < row_count && r < nrows; ++r) {
        if (rows[r].bitpos > first.t.scan_len * 8u) return -1;
        record(base0 + rows[r].bitpos, rows[r].last_dc, r);
    }
    // the record after the last scan (decode_scans' final make_handoff): end of all data, the last scan's DC predictors, and
    // the MCU row its MCU counter stood at -- which only interleaved scans advance
    const ProgScanDecodePlan& last = scans.back();
    const ScanDecodeRow& fin = rows[last.result_off];
    record((uint32_t)jf->scan.size() * 8u, fin.last_dc, last.cmpc > 1 ? jf->mcuc / jf->mcuh : 0);
    for (size_t i = 1; i < jf->rows.size(); ++i)
        if (jf->rows[i].luma_y_start < jf->rows[i - 1].luma_y_end) jf->rows[i].luma_y_start = jf->rows[i - 1].luma_y_end;
    jf->padbit = (int8_t)padbit;
    jf->scan_count = (int)scans.size();
    for (int c = 0; c < jf->ncomp; ++c) jf->max_dpos[c] = jf->comp[c].bc - 1;
    jf->progressive_needed = true;
    return 0;
}

int parse_jpeg(const uint8_t* data, size_t size, bool allow_progressive, JpegFile* jf) {
    if (size < 4 || data[0] != 0xFF || data[1] != 0xD8) { jf->error = "not a jpeg"; return EX_UNSUPPORTED_JPEG; }
    memset(jf->qtables, 0, sizeof jf->qtables);
    int rc = split_file(data, size, jf);
    if (rc) return rc;
    if (!setup_frame(jf)) return jf->warn < 0 ? -jf->warn : EX_UNSUPPORTED_JPEG;
    if (jf->ncomp > 3) return EX_UNSUPPORTED_4_COLORS;
    if (jf->start_byte) allow_progressive = false;   // "Encode of partial progressive images not allowed" (jpgcoder.cc:1205-1208)
    rc = decode_scans(jf, allow_progressive);
    // the reference's errorlevel 1 ("warnings": an unknown marker, a block ending in a coded zero, inconsistent pad bits,
    // non-optimal end-of-band runs, data left over in the scan) stops it before write_ujpg (err_tresh = 1,
    // jpgcoder.cc:528, 2047) and exits with UNSUPPORTED_JPEG (jpgcoder.cc:2024): none of those files can be restored
    // byte for byte by a canonical re-encoder
    if (!rc && jf->warn > 0) { if (jf->error.empty()) jf->error = "non-canonical JPEG (reference errorlevel 1)"; return EX_UNSUPPORTED_JPEG; }
    if (rc || !jf->start_byte) return rc;
    // -startbyte: drop the hand-off rows that lie in front of start_byte (the last record always stays) and keep the raw
    // bytes between start_byte and the first remaining row as prefix garbage (write_ujpg, jpgcoder.cc:3801-3843)
    std::vector<Handoff> kept;
    for (size_t i = 0; i < jf->rows.size(); ++i)
        if (i + 1 == jf->rows.size() || jf->rows[i].segment_size >= jf->start_byte) kept.push_back(jf->rows[i]);
    jf->rows.swap(kept);
    if (jf->rows.empty() || jf->rows[0].segment_size < jf->start_byte) return EX_ONLY_GARBAGE_NO_JPEG;
    uint32_t prefix = jf->rows[0].segment_size - jf->start_byte;
    if (jf->rows.size() > 1 && prefix) --prefix;   // the reference's own "FIXME why is this ?!" (jpgcoder.cc:3823-3827)
    jf->prefix_garbage.clear();
    if (prefix && jf->start_byte < size)
        jf->prefix_garbage.assign(data + jf->start_byte, data + jf->start_byte + std::min<size_t>(prefix, size - jf->start_byte));
    jf->prefix_garbage.resize(prefix, 0);
    return 0;
}

}  // namespace lep
