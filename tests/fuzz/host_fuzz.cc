// host_fuzz.cc -- sanitizer harness for the host-side parsers that face untrusted bytes in a serving process: the JPEG
// splitter / Huffman decoder / progressive decoder (lep_jpeg_open*), the .lep container parser and the JPEG re-coder
// (lep_file_*).  The reference runs this code under seccomp in a forked child (src/lepton/socket_serve.cc:86-116); the
// batching daemon parses in-process, so these paths must be memory-safe on any input.  Built by tests/test_fuzz_host.py with
// -fsanitize=address,undefined from the library's host sources; the two GPU entry points they reference are stubbed.
//   usage: host_fuzz <file>...      (prints nothing on success; a sanitizer report + abort on a finding)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/lepton_mi355x.h"

extern "C" int lep_gpu_encode_host(lep_gpu*, const lep_image_desc*, int, const lep_segment*, int, lep_bytes*, int32_t*) { return LEP_GPU_ERROR; }
extern "C" int lep_gpu_decode_host(lep_gpu*, const lep_image_desc*, int, const lep_segment*, int, const lep_bytes*, int32_t*) { return LEP_GPU_ERROR; }

static void run_jpeg(const uint8_t* d, size_t n) {
    for (int allow = 0; allow < 2; ++allow) {
        lep_jpeg* j = nullptr;
        if (lep_jpeg_open(d, n, allow, &j) != 0 || !j) continue;
        lep_image_desc desc;
        lep_jpeg_describe(j, &desc);
        lep_segment segs[LEP_MAX_SEGMENTS];
        const int ns = lep_jpeg_plan(j, 0, segs, 0);
        if (ns > 0 && ns <= LEP_MAX_SEGMENTS) {
            uint8_t fake[64];
            memset(fake, 0x5a, sizeof fake);
            lep_bytes st[LEP_MAX_SEGMENTS];
            for (int i = 0; i < ns; ++i) { st[i].data = fake; st[i].len = st[i].cap = sizeof fake; }
            lep_bytes out = {nullptr, 0, 0};
            if (lep_jpeg_write_lep(j, 0, st, ns, &out) == 0) lep_free(out.data);
        }
        lep_jpeg_close(j);
    }
    // -startbyte slices and -embedding blobs: the same parser behind two more front doors
    const size_t starts[] = {1, 17, n / 3, n / 2, n - 1, n, n + 5};
    for (size_t sb : starts) {
        lep_jpeg* j = nullptr;
        if (lep_jpeg_open_slice(d, n, sb, &j) != 0 || !j) continue;
        lep_segment segs[LEP_MAX_SEGMENTS];
        const int ns = lep_jpeg_plan(j, 0, segs, 0);
        if (ns > 0 && ns <= LEP_MAX_SEGMENTS) {
            uint8_t fake[16] = {0};
            lep_bytes st[LEP_MAX_SEGMENTS];
            for (int i = 0; i < ns; ++i) { st[i].data = fake; st[i].len = st[i].cap = sizeof fake; }
            lep_bytes out = {nullptr, 0, 0};
            if (lep_jpeg_write_lep(j, 0, st, ns, &out) == 0) lep_free(out.data);
        }
        lep_jpeg_close(j);
    }
    for (size_t off : {(size_t)0, (size_t)2, n / 4}) {
        lep_jpeg* j = nullptr;
        if (lep_jpeg_open_embedded(d, n, off, &j) == 0 && j) lep_jpeg_close(j);
    }
    // the GPU-assisted front end: split + table set-up only (the scan decode itself runs on the device)
    lep_jpeg* j = nullptr;
    lep_huffdec_image img;
    int ok = 0;
    if (lep_jpeg_open_gpu(d, n, &j, &img, &ok) == 0 && j) {
        if (ok) {
            const uint8_t* p = nullptr; size_t len = 0;
            lep_jpeg_scan_bytes(j, &p, &len);
            std::vector<lep_huffdec_row> rows((size_t)img.mcuv + 1);
            memset(rows.data(), 0, rows.size() * sizeof(lep_huffdec_row));
            for (size_t r = 0; r < rows.size(); ++r) rows[r].bitpos = (uint32_t)(r * 977u);   // nonsense records must be refused, not trusted
            lep_jpeg_finish_gpu(j, rows.data());
        }
        lep_jpeg_close(j);
    }
}

static void run_lep(const uint8_t* d, size_t n) {
    lep_file* f = nullptr;
    if (lep_file_open(d, n, &f) != 0 || !f) return;
    if (lep_file_frame_bytes(f) <= ((size_t)1 << 30)) {   // a header may announce a huge frame: the pipeline caps it the same way
        lep_image_desc desc;
        if (lep_file_describe(f, &desc) == 0) {
            lep_segment segs[LEP_MAX_SEGMENTS];
            lep_bytes streams[LEP_MAX_SEGMENTS];
            lep_file_segments(f, segs, streams, 0);
            lep_bytes out = {nullptr, 0, 0};
            if (lep_file_recode(f, &out) == 0) lep_free(out.data);
        }
    }
    lep_file_close(f);
}

int main(int argc, char** argv) {
    for (int a = 1; a < argc; ++a) {
        FILE* fp = fopen(argv[a], "rb");
        if (!fp) continue;
        std::vector<uint8_t> buf;
        uint8_t tmp[65536];
        size_t n;
        while ((n = fread(tmp, 1, sizeof tmp, fp)) > 0) buf.insert(buf.end(), tmp, tmp + n);
        fclose(fp);
        fprintf(stderr, "%s\n", argv[a]);
        // exact-size heap copy so that any over-read lands in a red zone
        uint8_t* d = static_cast<uint8_t*>(malloc(buf.size() ? buf.size() : 1));
        if (!buf.empty()) memcpy(d, buf.data(), buf.size());
        if (buf.size() >= 2 && d[0] == 0xff && d[1] == 0xd8) run_jpeg(d, buf.size());
        else run_lep(d, buf.size());
        free(d);
    }
    return 0;
}
