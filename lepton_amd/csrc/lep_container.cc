// lep_container.cc -- the .lep container around the arithmetic-coded streams (host side).
//   segment choice        src/lepton/jpgcoder.cc:3856-3934 (write_ujpg)
//   header sections       src/lepton/jpgcoder.cc:3953-4027 (write), :4117-4362 (read_ujpg)
//   fixed 28-byte prefix  src/lepton/jpgcoder.cc:4045-4069, :2140-2176 (read_fixed_ujpg_header)
//   hand-off records      src/lepton/thread_handoff.cc:4-76
//   stream multiplexing   src/io/MuxReader.hh:336-522 (MuxWriter), :38-334 (MuxReader);
//                         slice order src/lepton/vp8_encoder.cc:575-594; size trailer :602-614
#include "lep_container.h"

#include <dlfcn.h>
#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <mutex>

namespace lep {

// ------------------------------------------------------------------------------------------------
std::vector<Handoff> plan_segments(const JpegFile& jf, const EncodeOptions& opt) {
    const std::vector<Handoff>& rows = jf.rows;
    unsigned n = std::min(8u, opt.max_threads);
    const uint32_t scan_bytes = rows.back().segment_size - rows.front().segment_size;
    const uint32_t nrows = (uint32_t)rows.size();
    if (nrows / 2 < n) {
        unsigned want = std::max(nrows / 2, opt.min_threads);
        n = std::min(std::max(want, 1u), n);
    }
    if (scan_bytes < 125000) n = std::min(std::max(opt.min_threads, 1u), n);
    else if (scan_bytes < 250000) n = std::min(std::max(opt.min_threads, 2u), n);
    else if (scan_bytes < 500000) n = std::min(std::max(opt.min_threads, 4u), n);

    std::vector<int> cut(n, 0);
    if (!opt.even_split) {
        for (uint32_t i = 0; i + 1 < n; ++i) {
            uint32_t target = rows.back().segment_size - rows.front().segment_size;
            target = (uint32_t)((uint64_t)target * (i + 1)) ;   // 32-bit wrap matches the reference's uint32 math
            target /= n;
            target += rows.front().segment_size;
            auto it = std::lower_bound(rows.begin() + 1, rows.end(), target,
                                       [](const Handoff& a, uint32_t v) { return a.segment_size < v; });
            if (it != rows.begin() + 1) --it;
            cut[i] = (int)(it - rows.begin());
        }
    } else {
        for (uint32_t i = 0; i + 1 < n; ++i) cut[i] = (int)(rows.size() * (i + 1) / n);
    }
    for (uint32_t i = 0; i + 1 < n; ++i)
        if (cut[i] == cut[i + 1]) {
            for (uint32_t j = 0; j + 1 < n; ++j) cut[j] = (int)((j + 1) * rows.size() / n);
            break;
        }
    cut[n - 1] = (int)rows.size() - 1;
    std::vector<Handoff> segs(n);
    size_t begin = 0;
    for (size_t i = 0; i < n; ++i) {
        size_t end = (size_t)cut[i];
        Handoff s = rows[begin];                 // start-of-range state ...
        s.luma_y_end = rows[end].luma_y_start;   // ... up to where the end record starts
        s.segment_size = rows[end].segment_size - rows[begin].segment_size;
        if (i + 1 == n && rows[end].num_overhang_bits) ++s.segment_size;
        segs[i] = s;
        begin = end;
    }
    return segs;
}

// ------------------------------------------------------------------------------------------------
static void put_le32(std::vector<uint8_t>& v, uint32_t x) {
    for (int i = 0; i < 4; ++i) v.push_back((uint8_t)(x >> (8 * i)));
}
static uint32_t get_le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

std::vector<uint8_t> serialize_handoffs(const std::vector<Handoff>& segs) {
    std::vector<uint8_t> out;
    out.push_back('H');
    out.push_back((uint8_t)segs.size());
    for (const Handoff& h : segs) {
        out.push_back(h.luma_y_start & 255);
        out.push_back(h.luma_y_start >> 8);
        put_le32(out, h.segment_size);
        out.push_back(h.overhang_byte);
        out.push_back(h.num_overhang_bits);
        for (int i = 0; i < 3; ++i) { uint16_t dc = (uint16_t)h.last_dc[i]; out.push_back(dc & 255); out.push_back(dc >> 8); }
        out.push_back(0); out.push_back(0);   // 4th channel slot (3 colour channels in the default build)
    }
    return out;
}

bool deserialize_handoffs(const uint8_t* d, size_t n, std::vector<Handoff>* out) {
    if (n < 2 || d[0] != 'H') return false;
    int cnt = d[1];
    if ((size_t)cnt * 16 + 2 > n) return false;
    d += 2;
    for (int i = 0; i < cnt; ++i, d += 16) {
        Handoff h;
        h.luma_y_start = (uint16_t)(d[0] | (d[1] << 8));
        h.segment_size = get_le32(d + 2);
        h.overhang_byte = d[6];
        h.num_overhang_bits = d[7];
        for (int k = 0; k < 4; ++k) h.last_dc[k] = (int16_t)(d[8 + 2 * k] | (d[9 + 2 * k] << 8));
        h.last_dc[3] = 0;
        out->push_back(h);
    }
    for (size_t i = 1; i < out->size(); ++i) (*out)[i - 1].luma_y_end = (*out)[i].luma_y_start;
    return true;
}

static std::vector<uint8_t> build_header_payload(const JpegFile& jf, const std::vector<Handoff>& segs) {
    std::vector<uint8_t> p;
    p.insert(p.end(), {'H', 'D', 'R'});
    put_le32(p, (uint32_t)jf.hdr.size());
    p.insert(p.end(), jf.hdr.begin(), jf.hdr.end());
    p.insert(p.end(), {'P', '0', 'D'});
    p.push_back((uint8_t)(int8_t)jf.padbit);
    p.push_back('H');
    std::vector<uint8_t> hs = serialize_handoffs(segs);
    p.insert(p.end(), hs.begin(), hs.end());
    if (!jf.rst_cnt.empty()) {
        p.insert(p.end(), {'C', 'R', 'S'});
        put_le32(p, (uint32_t)jf.rst_cnt.size());
        for (uint32_t c : jf.rst_cnt) put_le32(p, c);
    }
    if (!jf.rst_err.empty()) {
        p.insert(p.end(), {'F', 'R', 'S'});
        put_le32(p, (uint32_t)jf.rst_err.size());
        p.insert(p.end(), jf.rst_err.begin(), jf.rst_err.end());
    }
    if (jf.early_eof) {
        p.insert(p.end(), {'E', 'E', 'E'});
        put_le32(p, (uint32_t)jf.max_cmp);
        put_le32(p, (uint32_t)jf.max_bpos);
        put_le32(p, (uint32_t)jf.max_sah);
        for (int i = 0; i < 4; ++i) put_le32(p, (uint32_t)jf.max_dpos[i]);
    }
    if (jf.start_byte || jf.embedded) {   // written even when empty (prefix_grbgdata != NULL, jpgcoder.cc:4009-4017)
        p.insert(p.end(), {(uint8_t)'P', (uint8_t)'G', (uint8_t)(jf.embedded ? 'E' : 'R')});
        put_le32(p, (uint32_t)jf.prefix_garbage.size());
        p.insert(p.end(), jf.prefix_garbage.begin(), jf.prefix_garbage.end());
    }
    if (!jf.garbage.empty()) {
        p.insert(p.end(), {'G', 'R', 'B'});
        put_le32(p, (uint32_t)jf.garbage.size());
        p.insert(p.end(), jf.garbage.begin(), jf.garbage.end());
    }
    return p;
}

// zlib level 9, one deflate(Z_NO_FLUSH) then Z_FINISH (src/io/ZlibCompression.cc:44-75)
static bool zlib9(const std::vector<uint8_t>& in, std::vector<uint8_t>* out) {
    z_stream s;
    memset(&s, 0, sizeof s);
    if (deflateInit(&s, 9) != Z_OK) return false;
    out->resize(compressBound((uLong)in.size()));
    s.next_in = (Bytef*)in.data(); s.avail_in = (uInt)in.size();
    s.next_out = out->data(); s.avail_out = (uInt)out->size();
    int r = deflate(&s, Z_NO_FLUSH);
    while (r != Z_STREAM_END) {
        r = deflate(&s, Z_FINISH);
        if (r != Z_OK && r != Z_STREAM_END && r != Z_BUF_ERROR) { deflateEnd(&s); return false; }
    }
    out->resize(out->size() - s.avail_out);
    deflateEnd(&s);
    return true;
}

// limit: the reference inflates at most max_file_size + 2048 bytes of header and treats a longer one like a corrupt one
// (ZlibDecoderDecompressionReader::Decompress(..., max_file_size + 2048), jpgcoder.cc:4158-4166) -- which also keeps a
// zlib bomb in a request from blowing up a shared serving process
static bool unzlib(const uint8_t* d, size_t n, size_t limit, std::vector<uint8_t>* out) {
    z_stream s;
    memset(&s, 0, sizeof s);
    if (inflateInit(&s) != Z_OK) return false;
    s.next_in = (Bytef*)d; s.avail_in = (uInt)n;
    out->clear();
    uint8_t buf[65536];
    int r;
    do {
        s.next_out = buf; s.avail_out = sizeof buf;
        r = inflate(&s, Z_NO_FLUSH);
        if (r != Z_OK && r != Z_STREAM_END) { inflateEnd(&s); return false; }
        out->insert(out->end(), buf, buf + (sizeof buf - s.avail_out));
        if (out->size() > limit) { inflateEnd(&s); return false; }
    } while (r != Z_STREAM_END);
    inflateEnd(&s);
    return true;
}

// Headers of format versions >= 2 are brotli streams (`lepton -brotliheader`, jpgcoder.cc:1116-1119, 4038, 4172;
// src/io/BrotliCompression.cc:100-150).  The reference vendors brotli 1.0.0; decoding is version-independent (RFC 7932), so
// the system's libbrotlidec is bound at first use.  Without it such files are refused with VERSION_UNSUPPORTED, loudly.
namespace {
struct BrotliDec {
    void* lib = nullptr;
    void* (*create)(void*, void*, void*) = nullptr;
    int (*stream)(void*, size_t*, const uint8_t**, size_t*, uint8_t**, size_t*) = nullptr;
    void (*destroy)(void*) = nullptr;
    bool ok = false;
};
const BrotliDec& brotli_dec() {
    static BrotliDec b;
    static std::once_flag once;
    std::call_once(once, []() {
        for (const char* name : {"libbrotlidec.so.1", "libbrotlidec.so"}) {
            b.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (b.lib) break;
        }
        if (!b.lib) return;
        b.create = (void* (*)(void*, void*, void*))dlsym(b.lib, "BrotliDecoderCreateInstance");
        b.stream = (int (*)(void*, size_t*, const uint8_t**, size_t*, uint8_t**, size_t*))dlsym(b.lib, "BrotliDecoderDecompressStream");
        b.destroy = (void (*)(void*))dlsym(b.lib, "BrotliDecoderDestroyInstance");
        b.ok = b.create && b.stream && b.destroy;
    });
    return b;
}
}  // namespace

bool brotli_available() { return brotli_dec().ok; }

// BrotliCodec::Decompress: the whole input must be one complete stream (left-over input is an error), output bounded
static bool unbrotli(const uint8_t* d, size_t n, size_t limit, std::vector<uint8_t>* out) {
    const BrotliDec& b = brotli_dec();
    if (!b.ok) return false;
    void* st = b.create(nullptr, nullptr, nullptr);
    if (!st) return false;
    out->clear();
    uint8_t buf[65536];
    size_t avail_in = n, total = 0;
    const uint8_t* next_in = d;
    bool good = false;
    for (;;) {
        size_t avail_out = sizeof buf;
        uint8_t* next_out = buf;
        const int r = b.stream(st, &avail_in, &next_in, &avail_out, &next_out, &total);   // 0 error, 1 success, 2 more input, 3 more output
        out->insert(out->end(), buf, buf + (sizeof buf - avail_out));
        if (out->size() > limit) break;
        if (r == 1) { good = avail_in == 0; break; }
        if (r == 3) continue;
        break;   // error, or the input ended inside the stream
    }
    b.destroy(st);
    return good;
}

// ------------------------------------------------------------------------------------------------
// Multiplexer: same packetisation policy as MuxWriter, expressed over per-stream pending queues.
namespace {
struct Mux {
    std::vector<uint8_t>& out;
    std::vector<uint8_t> pend[16];     // bytes accepted but not yet emitted
    uint32_t flushed[16] = {0};
    uint32_t low_water[16] = {0};
    uint32_t total = 0;
    explicit Mux(std::vector<uint8_t>& o) : out(o) {}

    static uint32_t high_water(uint32_t flushed) { return (flushed & 0xffffc000u) ? 65536 : (flushed & 0xfffff000u) ? 16384 : 4096; }

    void emit_all(int id) {   // variable-length packets, <= 65536 each ("flushFull")
        std::vector<uint8_t>& q = pend[id];
        size_t off = 0;
        while (off < q.size()) {
            uint32_t len = (uint32_t)std::min<size_t>(q.size() - off, 65536);
            out.push_back((uint8_t)id);
            out.push_back((uint8_t)((len - 1) & 0xff));
            out.push_back((uint8_t)((len - 1) >> 8));
            out.insert(out.end(), q.begin() + off, q.begin() + off + len);
            off += len; total += len; flushed[id] += len;
        }
        if (!q.empty()) { q.clear(); low_water[id] = total; }
    }
    void emit_fixed(int id) {   // power-of-four sized packets with a 1-byte header ("flushPartial")
        std::vector<uint8_t>& q = pend[id];
        uint32_t have = (uint32_t)q.size(), len, code;
        if (have < 4096) { emit_all(id); return; }
        if (have < 16384) { if (have > 8192) { emit_all(id); return; } len = 4096; code = 1; }
        else if (have < 65536) { if (have > 32768) { emit_all(id); return; } len = 16384; code = 2; }
        else { if (have > 131072) { emit_all(id); return; } len = 65536; code = 3; }
        uint32_t off = 0;
        for (; off + len <= have; off += len) {
            out.push_back((uint8_t)(id | (code << 4)));
            out.insert(out.end(), q.begin() + off, q.begin() + off + len);
            total += len; flushed[id] += len;
        }
        q.erase(q.begin(), q.begin() + off);
        uint32_t behind = (uint32_t)q.size();
        low_water[id] = behind > total ? 0 : total - behind;
    }
    void flush_for(int id) {
        for (int i = 0; i < 16; ++i) {
            uint32_t have = (uint32_t)pend[i].size();
            if (i == id || !have) continue;
            bool urgent = total - low_water[i] > 65537;
            if (have < 4096) { if (urgent) emit_all(i); }
            else if (urgent && have < 16384) emit_all(i);
            else emit_fixed(i);
        }
        emit_fixed(id);
    }
    void write(int id, const uint8_t* d, size_t n) {
        pend[id].insert(pend[id].end(), d, d + n);
        if (pend[id].size() >= high_water(flushed[id])) flush_for(id);
    }
    void close(int version) {
        for (int i = 0; i < 16; ++i) emit_all(i);
        if (version > 1) out.insert(out.end(), {0xFF, 0xFE, 0xFF});
    }
};
}  // namespace

void mux_streams(const std::vector<std::vector<uint8_t>>& streams, int version, std::vector<uint8_t>* out) {
    Mux m(*out);
    std::vector<size_t> off(streams.size(), 0);
    bool any = true;
    while (any) {
        any = false;
        for (size_t i = 0; i < streams.size() && i < 16; ++i) {
            if (streams[i].size() <= off[i]) continue;
            any = true;
            size_t slice = off[i] == 0 ? 256 : off[i] == 256 ? 4096 : 65536;
            size_t n = std::min(slice, streams[i].size() - off[i]);
            m.write((int)i, streams[i].data() + off[i], n);
            off[i] += n;
        }
    }
    m.close(version);
}

// ------------------------------------------------------------------------------------------------
// BrotliCodec::Compress (src/io/BrotliCompression.cc:45-98) with the reference's own encoder: quality 10 (BrotliCompression.hh:47),
// size hint, lgwin = lgblock = bit length of the size + 1, clamped to the encoder's window range
#ifdef LEP_HAVE_BROTLI_ENC
}  // namespace lep
// The encoder's C ABI (brotli/encode.h of brotli 1.0.0), declared here so that nothing but the object files is needed to build:
// they are compiled from the reference's dependency tree in this container and travel with the snapshot.
extern "C" {
struct BrotliEncoderStateStruct;
typedef struct BrotliEncoderStateStruct BrotliEncoderState;
BrotliEncoderState* BrotliEncoderCreateInstance(void* (*alloc)(void*, size_t), void (*dealloc)(void*, void*), void* opaque);
int BrotliEncoderSetParameter(BrotliEncoderState* state, int param, uint32_t value);
size_t BrotliEncoderMaxCompressedSize(size_t input_size);
int BrotliEncoderCompressStream(BrotliEncoderState* state, int op, size_t* available_in, const uint8_t** next_in, size_t* available_out, uint8_t** next_out, size_t* total_out);
int BrotliEncoderIsFinished(BrotliEncoderState* state);
void BrotliEncoderDestroyInstance(BrotliEncoderState* state);
}
enum { BROTLI_OPERATION_PROCESS = 0, BROTLI_OPERATION_FINISH = 2, BROTLI_PARAM_QUALITY = 1, BROTLI_PARAM_LGWIN = 2, BROTLI_PARAM_LGBLOCK = 3, BROTLI_PARAM_SIZE_HINT = 5,
       BROTLI_MIN_WINDOW_BITS = 10, BROTLI_MAX_WINDOW_BITS = 24 };
namespace lep {
bool brotli_encoder_available() { return true; }
static bool brotli10(const std::vector<uint8_t>& in, std::vector<uint8_t>* out) {
    BrotliEncoderState* st = BrotliEncoderCreateInstance(nullptr, nullptr, nullptr);
    if (!st) return false;
    size_t size = in.size();
    BrotliEncoderSetParameter(st, BROTLI_PARAM_SIZE_HINT, (uint32_t)size);
    BrotliEncoderSetParameter(st, BROTLI_PARAM_QUALITY, 10);
    uint32_t lgwin = 1;
    for (size_t t = size; t; t >>= 1) ++lgwin;
    lgwin = std::min<uint32_t>(std::max<uint32_t>(lgwin, BROTLI_MIN_WINDOW_BITS), BROTLI_MAX_WINDOW_BITS);
    BrotliEncoderSetParameter(st, BROTLI_PARAM_LGWIN, lgwin);
    BrotliEncoderSetParameter(st, BROTLI_PARAM_LGBLOCK, lgwin);
    out->resize(BrotliEncoderMaxCompressedSize(size) + 16);
    const uint8_t* next_in = in.data();
    uint8_t* next_out = out->data();
    size_t avail_out = out->size(), total = 0;
    bool ok = true;
    for (;;) {
        if (!BrotliEncoderCompressStream(st, size == 0 ? BROTLI_OPERATION_FINISH : BROTLI_OPERATION_PROCESS, &size, &next_in, &avail_out, &next_out, &total)) { ok = false; break; }
        if (size == 0 && BrotliEncoderIsFinished(st)) break;
    }
    BrotliEncoderDestroyInstance(st);
    if (ok) out->resize((size_t)(next_out - out->data()));
    return ok;
}
#else
bool brotli_encoder_available() { return false; }
static bool brotli10(const std::vector<uint8_t>&, std::vector<uint8_t>*) { return false; }
#endif

int write_lep(const JpegFile& jf, const std::vector<Handoff>& segs, const std::vector<std::vector<uint8_t>>& streams,
              std::vector<uint8_t>* out, int format_version) {
    if (format_version != 1 && format_version != 2) return EX_VERSION_UNSUPPORTED;
    if (format_version == 2 && !brotli_encoder_available()) return EX_VERSION_UNSUPPORTED;   // said loudly: the system's brotli writes other bytes
    std::vector<uint8_t> payload = build_header_payload(jf, segs), z;
    if (format_version == 1 ? !zlib9(payload, &z) : !brotli10(payload, &z)) return EX_OS_ERROR;
    out->clear();
    out->reserve(z.size() + 64);
    out->push_back(0xCF); out->push_back(0x84);
    out->push_back((uint8_t)format_version);
    out->push_back(jf.start_byte ? 'Y' : (jf.progressive_needed ? 'X' : 'Z'));   // jpgcoder.cc:4046-4052
    out->push_back((uint8_t)segs.size());
    out->insert(out->end(), 3, 0);
    out->insert(out->end(), 12, 0);                        // git revision: zeros, as an out-of-git reference build writes
    put_le32(*out, jf.file_size - jf.start_byte);
    put_le32(*out, (uint32_t)z.size());
    out->insert(out->end(), z.begin(), z.end());
    out->insert(out->end(), {'C', 'M', 'P'});
    mux_streams(streams, format_version, out);
    put_le32(*out, (uint32_t)out->size() + 4);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// returns the offset behind the last packet taken; *saw_eof: stopped on the end marker FF FE FF that format versions >= 2
// write behind the packets (MuxWriter::Close, MuxReader.hh:507-517) -- the marker itself is not consumed
size_t demux_packets(const uint8_t* d, size_t n, size_t at, std::vector<std::vector<uint8_t>>* streams, bool* saw_eof) {
    streams->assign(16, {});
    if (saw_eof) *saw_eof = false;
    while (at + 3 <= n) {
        uint8_t h = d[at];
        if (d[at] == 0xFF && d[at + 1] == 0xFE && d[at + 2] == 0xFF) { if (saw_eof) *saw_eof = true; break; }
        int id = h & 15, fl = (h >> 4) & 3;
        size_t len, hl;
        if (fl == 0) { len = (size_t)d[at + 1] + ((size_t)d[at + 2] << 8) + 1; hl = 3; }
        else { len = (size_t)1024 << (2 * fl); hl = 1; }
        if (at + hl + len + 3 > n) break;   // the reader needs the next 3 header bytes too
        (*streams)[id].insert((*streams)[id].end(), d + at + hl, d + at + hl + len);
        at += hl + len;
    }
    return at;
}

// carried: what a previous file of the same stream left in the header reader behind a "CNT" section (`lepton -lepcat`
// merges the headers of all its inputs into the first file's, concat.cc:67-108); this file then has no header of its own
// Every worker thread's output buffer is allocated up front at its byte bound -- the sum of its logical threads' segment sizes
// as an int, the whole file's size if that is zero (recode_baseline_jpeg, recoder.cc:770-782; BoundedMemWriter::set_bound
// resizes) -- from the main arena of the default build (1024 MiB - 7 x 64 MiB, jpgcoder.cc:827-838): hand-offs that claim
// more than it holds end in OOM before a single bin is decoded, whatever the streams are.  Probed against the binary: the
// worker bounds may sum to 575.5 MiB beside a 256x256 image and 571.7 MiB beside a 1600x1200 one (what else lives in the
// arena grows with the file); the first thread writes to the output and is not counted.
bool worker_bounds_exceed_arena(const LepFile& lf, size_t file_bytes) {
    const bool baseline_recoder = lf.flag == 'Z' || (lf.flag & 1) == ('Y' & 1);
    if (!baseline_recoder || lf.segs.empty() || lf.segs[0].num_overhang_bits == 0xff) return false;
    const int P = std::min(lf.nthreads, 8), L = (int)lf.segs.size();
    uint64_t total = 0;
    for (int p = 1; p < P; ++p) {
        int a = p * L / P, b = std::min((p + 1) * L / P, L);
        if (L < P) { a = std::min(p, L); b = std::min(p + 1, L); }   // logical_thread_range_from_physical_thread_id, recoder.cc:547-559
        int32_t work = 0;
        for (int l = a; l < b; ++l) work = (int32_t)((uint32_t)work + lf.segs[l].segment_size);
        if (!work) work = (int32_t)lf.jpeg_size;
        if (work < 0) return true;   // (size_t)(int) of a negative bound: no arena holds it
        total += (uint64_t)work;
    }
    return P > 1 && total + (512u << 10) + 4 * (uint64_t)file_bytes > ((uint64_t)576 << 20);
}

int parse_lep(const uint8_t* d, size_t n, LepFile* lf, const std::vector<uint8_t>* carried) {
    if (n >= 2 && n < 28 && d[0] == 0xCF && d[1] == 0x84) return EX_SHORT_READ;   // read_fixed_ujpg_header: ReadFull(22) != 22
    if (n < 28 || d[0] != 0xCF || d[1] != 0x84) return EX_VERSION_UNSUPPORTED;
    lf->version = d[2];
    lf->flag = d[3];
    lf->nthreads = d[4];
    lf->consumed = n;
    if (lf->version < 1 || lf->version > 4) return EX_VERSION_UNSUPPORTED;   // read_fixed_ujpg_header, jpgcoder.cc:2148-2153
    if (lf->nthreads == 0) return EX_ASSERTION_FAILURE;    // always_assert(num_threads_hint != 0), jpgcoder.cc:2168
    if (lf->version == 3) return EX_ASSERTION_FAILURE;     // ANS-coded streams: "ANS compile flag not selected" (jpgcoder.cc:461-468)
    lf->jpeg_size = get_le32(d + 20);
    uint32_t zsize = get_le32(d + 24);
    // "Only support images < 128 megs" (jpgcoder.cc:4133-4136).  max_file_size is an int there: a size of 2^31 and more passes
    // this test and trips always_assert(max_file_size > grbs) once the header has been read (refusals of the header come first)
    if (zsize > (128u << 20) || (int32_t)lf->jpeg_size > (128 << 20)) return EX_ASSERTION_FAILURE;
    const size_t sane_size = (int32_t)lf->jpeg_size < 0 ? ((size_t)128 << 20) : (size_t)lf->jpeg_size;   // what a header may inflate to is never sized from such a claim
    // header_reader != NULL (a "CNT" section of the previous file left it open, even with nothing behind it): the compressed
    // header bytes are not read, and always_assert(compressed_header_size == 0 && "Special concatenation requires 0 size
    // header") (jpgcoder.cc:4139, 4186-4188)
    if (carried && zsize != 0) return EX_ASSERTION_FAILURE;
    if (28 + (uint64_t)zsize + 3 > n) return EX_SHORT_READ;
    std::vector<uint8_t> p;
    if (carried) {
        p = *carried;   // (may be empty: "HDR marker not found" below, as the reference's drained reader answers)
    } else if (lf->version == 1) {
        if (!unzlib(d + 28, zsize, sane_size + 2048, &p)) return EX_STREAM_INCONSISTENT;
    } else {
        if (!brotli_available()) return EX_VERSION_UNSUPPORTED;   // no libbrotlidec on this host: said loudly, never guessed
        if (!unbrotli(d + 28, zsize, sane_size * 2 + ((size_t)128 << 20), &p)) return EX_STREAM_INCONSISTENT;
    }
    // The header sections, read the way the reference reads them (read_ujpg, jpgcoder.cc:4193-4339): every field goes through
    // ReadFull into one 64-byte scratch buffer, and a header that ends early simply leaves that buffer -- or the zeroed
    // destination -- as it was.  Reproduced literally (scratch buffer included), because which damaged headers are still
    // accepted, and as what, is part of the decode direction's parity (tests/fuzz/diff_lep_structured.py).
    size_t pos = 0;
    uint8_t mrk[64] = {0};
    auto read_full = [&](uint8_t* dst, size_t k) { const size_t got = std::min(k, p.size() - pos); if (got) memcpy(dst, p.data() + pos, got); pos += got; return got; };
    auto left = [&]() { return p.size() - pos; };
    JpegFile& jf = lf->jpeg;
    read_full(mrk, 3);
    if (memcmp(mrk, "HDR", 3)) return EX_UNSUPPORTED_JPEG;   // "HDR marker not found": errorlevel 2
    read_full(mrk, 4);
    const uint32_t hdrs = get_le32(mrk);
    // Sizes the reference would try to allocate from its main arena (576 MiB in the default build, jpgcoder.cc:827-838) and
    // fail: OOM, whatever else is wrong with the file.  Between 128 MiB and the arena the reference allocates, zero-fills and
    // reads what is there; an in-process daemon does not follow it there (garbage sections of that size are refused as
    // STREAM_INCONSISTENT: documented deviation; the header's case is decided without the allocation, below).
    const uint64_t kArena = ((uint64_t)576 << 20) - (1u << 20);
    if (hdrs > kArena) return EX_BLOCK_OFFSET_OOM;
    if (hdrs > (128u << 20)) {
        // the reference allocates and zero-fills that much, reads what the file holds into it and interprets it; nothing is left
        // for the sections behind it: "PAD marker not found" unless the header itself is refused first.  The same verdicts from
        // what the file holds plus a short zero tail, without the allocation.
        const size_t backed = std::min<size_t>(left(), hdrs);   // (more inflated data may follow the claim: it is not the header's)
        jf.hdr.assign(backed + std::min<size_t>((size_t)hdrs - backed, 16), 0);   // never longer than the claim
        read_full(jf.hdr.data(), backed);
        memset(jf.qtables, 0, sizeof jf.qtables);
        if (!setup_frame(&jf)) return jf.warn < 0 ? -jf.warn : EX_UNSUPPORTED_JPEG;
        return EX_UNSUPPORTED_JPEG;
    }
    if (hdrs > (1u << 20) && (uint64_t)hdrs > (uint64_t)left() + 16) {
        // A claim of megabytes that the inflated header cannot back (54 bytes drove the shared daemon to +127 MB
        // per request).  The reference zero-fills the claim and interprets it; what follows is decided by the bytes that exist
        // plus at most one marker segment reaching into the zeros, and nothing is left for the "P0D" section behind it: the
        // same verdicts from the backed bytes and a zero tail of two maximal segments, without the allocation.
        // (the tail is the rest of the claim where that is shorter -- the reference's buffer is exactly the claim, and a last marker
        // segment reaching past it is refused there: ADVICE round 3)
        const size_t have = left();
        jf.hdr.assign(have + std::min<size_t>((size_t)hdrs - have, 2 * 65540), 0);
        read_full(jf.hdr.data(), have);
        memset(jf.qtables, 0, sizeof jf.qtables);
        if (!setup_frame(&jf)) return jf.warn < 0 ? -jf.warn : EX_UNSUPPORTED_JPEG;
        return EX_UNSUPPORTED_JPEG;   // "PAD marker not found"
    }
    jf.hdr.assign(hdrs, 0);
    read_full(jf.hdr.data(), hdrs);
    // the embedded JPEG header is interpreted here, before the sections behind it are looked at (setup_imginfo_jpg, called at
    // jpgcoder.cc:4215): its refusals (sampling factors, frame budget, ...) come before "PAD marker not found"
    memset(jf.qtables, 0, sizeof jf.qtables);
    if (!setup_frame(&jf)) return jf.warn < 0 ? -jf.warn : EX_UNSUPPORTED_JPEG;
    read_full(mrk, 3);
    if (!memcmp(mrk, "P0D", 3)) { uint8_t b = 0xff; read_full(&b, 1); jf.padbit = (int8_t)b; }
    else if (!memcmp(mrk, "PAD", 3)) {
        uint8_t b = 0xff; read_full(&b, 1);
        const int8_t pb = (int8_t)b;
        if (!(pb == 0 || pb == 1 || pb == -1)) return EX_STREAM_INCONSISTENT;
        jf.padbit = pb == 1 ? 0x7f : pb;
    } else return EX_UNSUPPORTED_JPEG;                         // "PAD marker not found": errorlevel 2
    lf->garbage_default_eoi = true;
    while (read_full(mrk, 3) == 3) {
        if (!memcmp(mrk, "CRS", 3)) {
            read_full(mrk, 4);
            const uint32_t c = get_le32(mrk);
            if ((uint64_t)c * 4 > kArena) return EX_BLOCK_OFFSET_OOM;
            if (c > (1u << 22) && (uint64_t)c > left() / 4 + 16) return EX_STREAM_INCONSISTENT;   // a count the data cannot back: a vector of hundreds of MB in the reference
            lf->rst_cnt_set = true;
            jf.rst_cnt.resize(c);
            for (uint32_t i = 0; i < c; ++i) { read_full(mrk, 4); jf.rst_cnt[i] = get_le32(mrk); }
        } else if (mrk[0] == 'H' && mrk[1] == 'H') {   // only the first two bytes are looked at; the third is the count
            const size_t bytes = (size_t)mrk[2] * 16 + 2;
            std::vector<uint8_t> z(bytes, 0);
            z[0] = mrk[1]; z[1] = mrk[2];
            read_full(z.data() + 2, bytes - 2);
            lf->segs.clear();   // a later section replaces an earlier one (thread_handoff = ThreadHandoff::deserialize(...), jpgcoder.cc:4266)
            if (!deserialize_handoffs(z.data(), bytes, &lf->segs)) return EX_VERSION_UNSUPPORTED;
        } else if (!memcmp(mrk, "FRS", 3)) {
            read_full(mrk, 4);
            const uint32_t c = get_le32(mrk);
            if (c >= jf.rst_err.size() && (uint64_t)c - jf.rst_err.size() > kArena) return EX_BLOCK_OFFSET_OOM;
            if ((c > (1u << 24) && (uint64_t)c > left() + 16) || c < jf.rst_err.size()) return EX_STREAM_INCONSISTENT;
            jf.rst_err.resize(c, 0);
            read_full(jf.rst_err.data(), c);
        } else if (!memcmp(mrk, "GRB", 3)) {
            read_full(mrk, 4);
            const uint32_t c = get_le32(mrk);
            if (c > kArena) return EX_BLOCK_OFFSET_OOM;
            if (c > (128u << 20)) return EX_STREAM_INCONSISTENT;
            if (c > (1u << 20) && (uint64_t)c > (uint64_t)left() + 16) return EX_STREAM_INCONSISTENT;   // megabytes of zero fill from a claim the data cannot back: refused like CRS / FRS (documented deviation; the reference allocates)
            jf.garbage.assign(c, 0);
            read_full(jf.garbage.data(), c);
            lf->garbage_default_eoi = false;
        } else if (!memcmp(mrk, "PGR", 3) || !memcmp(mrk, "PGE", 3)) {
            lf->embedded = lf->embedded || mrk[2] == 'E';
            read_full(mrk, 4);
            const uint32_t c = get_le32(mrk);
            if (c > kArena) return EX_BLOCK_OFFSET_OOM;
            if (c > (128u << 20)) return EX_STREAM_INCONSISTENT;
            if (c > (1u << 20) && (uint64_t)c > (uint64_t)left() + 16) return EX_STREAM_INCONSISTENT;   // as for GRB
            lf->has_prefix = true;
            lf->prefix_garbage.assign(c, 0);
            read_full(lf->prefix_garbage.data(), c);
        } else if (!memcmp(mrk, "SIZ", 3)) {
            read_full(mrk, 4);
            lf->jpeg_size = get_le32(mrk);
        } else if (!memcmp(mrk, "EEE", 3)) {
            read_full(mrk, 28);
            jf.max_cmp = (int)get_le32(mrk); jf.max_bpos = (int)get_le32(mrk + 4); jf.max_sah = (int)get_le32(mrk + 8);
            for (int i = 0; i < 4; ++i) jf.max_dpos[i] = (int)get_le32(mrk + 12 + 4 * i);
            jf.early_eof = true;
        } else if (!memcmp(mrk, "CNT", 3)) {   // the rest of this header belongs to the next file of the stream
            lf->pending_header.assign(p.begin() + pos, p.end());
            lf->header_pending = true;
            break;
        } else if (!memcmp(mrk, "CMP", 3)) {
            break;
        } else {
            return EX_UNSUPPORTED_JPEG;   // "unknown data found": errorlevel 2 (jpgcoder.cc:4326-4337)
        }
    }
    if (lf->garbage_default_eoi) jf.garbage = {0xFF, 0xD9};
    const uint8_t* q = d + 28 + zsize;
    if (memcmp(q, "CMP", 3)) return EX_STREAM_INCONSISTENT;
    // de-multiplex until the bytes run out (v1 has no end marker; the 4-byte size trailer never parses
    // as a complete packet -- MuxReader::nextDataPacket, MuxReader.hh:230-283)
    size_t at = 28 + (size_t)zsize + 3;
    if (lf->segs.empty()) {
        // pre-hand-off files: count byte + (count-1) LE16 luma split rows follow "CMP"
        // (VP8ComponentDecoder::initialize_decoder_state, src/lepton/vp8_decoder.cc:337-364)
        if (at >= n) return EX_SHORT_READ;
        unsigned mark = d[at++];
        if (mark == 0) return EX_THREADING_PARTIAL_MCU;
        if (at + 2 * (mark - 1) > n) return EX_SHORT_READ;
        Handoff th;
        th.num_overhang_bits = 0xff;   // LEGACY_OVERHANG_BITS: state is carried from the previous segment
        lf->segs.assign(mark, th);
        for (unsigned i = 0; i + 1 < mark; ++i, at += 2) lf->segs[i].luma_y_end = (uint16_t)(d[at] | (d[at + 1] << 8));
        for (unsigned i = 1; i < mark; ++i) lf->segs[i].luma_y_start = lf->segs[i - 1].luma_y_end;
    }
    // the decoder runs min(thread hint, MAX_NUM_THREADS = 8) physical threads (one, for a pre-hand-off file: recoder.cc:731-733);
    // one without a hand-off of its own trips always_assert(logical_thread_start < thread_handoffs.size()) (recoder.cc:547-578)
    // more logical threads than stream ids (MuxReader::MAX_STREAM_ID = 16, MuxReader.hh:201): the reference dies in an
    // always_assert / out-of-bounds access (probed: abort with 17 and 32 hand-offs, SIGSEGV with 200); every caller of
    // lep_file_segments sizes its arrays LEP_MAX_SEGMENTS
    const bool baseline_recoder = lf->flag == 'Z' || (lf->flag & 1) == ('Y' & 1);   // jpgcoder.cc:2162; the general re-coder is single-threaded
    if (lf->segs.size() > 16) return baseline_recoder ? EX_ASSERTION_FAILURE : EX_CODING_ERROR;   // (the general re-coder's decoder: more threads needed than it was started with, vp8_decoder.cc:415-417)
    if (baseline_recoder && lf->segs[0].num_overhang_bits != 0xff && (size_t)std::min(lf->nthreads, 8) > lf->segs.size()) return EX_ASSERTION_FAILURE;
    // (worker bounds beyond the arena: worker_bounds_exceed_arena, called by lep_file_open_next behind the re-coder's header pass)
    // (more logical threads than the general re-coder's decoder was started with: lep_file_open_next, behind the split-table check)
    bool saw_eof = false;
    const size_t end = demux_packets(d, n, at, &lf->streams, &saw_eof);
    // Format versions >= 2 end their packets with a marker, so the reader knows where the file stops: 3 marker bytes, the
    // 4-byte size trailer, and whatever follows is the next file of a chained stream (`cat a.lep b.lep | lepton -`,
    // jpgcoder.cc:1881-1897, test_suite/test_concat.sh).  Version 1 has no marker: its reader runs to the end of the input.
    // the general re-coder's decoder hands every packet to the thread bound to its stream id; one for a thread that no hand-off
    // created is always_assert(false && "Cannot send to thread that wasn't bound") (vp8_decoder.cc:236)
    // -- met when the decoder routes that packet, i.e. behind every refusal the header earns (errorlevel from the embedded JPEG
    // header, the re-coder's table pass, its empty scan table): recorded here, answered by lep_file_open_next after those
    if (!baseline_recoder)
        for (size_t i = lf->segs.size(); i < lf->streams.size(); ++i)
            if (!lf->streams[i].empty()) lf->unbound_stream_packet = true;
    if (lf->version > 1 && saw_eof && end + 7 <= n) lf->consumed = end + 7;
    return 0;
}

}  // namespace lep
