// host-pool scaling of the JPEG parser (Huffman scan decode): aggregate MB/s vs thread count
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "../../include/lepton_mi355x.h"
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> jpg(n); if (fread(jpg.data(), 1, n, f) != (size_t)n) return 1; fclose(f);
    size_t fb = 0; lep_jpeg_peek_frame_bytes(jpg.data(), n, &fb);
    for (int into = 0; into < 2; ++into)
    for (int nt : {1, 4, 16, 64, 128, 256}) {
        const int per = 8, total = nt * per;
        std::vector<char> arena(into ? (size_t)nt * fb : 1);
        std::atomic<int> next(0);
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back([&, t]() {
            for (int i; (i = next.fetch_add(1)) < total;) {
                lep_jpeg* j = nullptr;
                int rc = into ? lep_jpeg_open_into(jpg.data(), n, 1, arena.data() + (size_t)t * fb, fb, &j) : lep_jpeg_open(jpg.data(), n, 1, &j);
                if (rc) abort();
                lep_jpeg_close(j);
            }
        });
        for (auto& t : pool) t.join();
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%s %3d threads: %.1f ms per file per thread, aggregate %.0f MB/s\n", into ? "into-arena" : "own-vectors", nt, s / per * 1e3, total * (double)n / 1e6 / s);
    }
    return 0;
}
