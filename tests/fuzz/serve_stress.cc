// serve_stress.cc -- thread- and memory-safety harness for lep_serve.cc (built by tests/test_fuzz_host.py with
// -fsanitize=thread and again with -fsanitize=address,undefined): a server with a trivial in-process processor, hammered by
// well-behaved clients, clients that hang up mid-upload, clients that never read their answer, oversized uploads and unknown
// file types, with a time bound running.  Exit code 0 and a silent sanitizer = pass.
//   usage: serve_stress <socket path> <seconds>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../../include/lepton_mi355x.h"

// the batch entry points lep_serve.cc references; never reached (the harness installs its own processor)
extern "C" int lep_compress_batch(lep_gpu*, const lep_bytes*, int, lep_bytes*, int32_t*, const lep_batch_options*, lep_batch_stats*) { return LEP_GPU_ERROR; }
extern "C" int lep_decompress_batch(lep_gpu*, const lep_bytes*, int, lep_bytes*, int32_t*, const lep_batch_options*, lep_batch_stats*) { return LEP_GPU_ERROR; }

static std::atomic<long> g_batches{0};

// "compress": answer = cf 84 + payload reversed; "decompress": undo it.  Every 7th file fails.
static int process(void*, int kind, const lep_bytes* in, int n, lep_bytes* outs, int32_t* status) {
    ++g_batches;
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
    for (int i = 0; i < n; ++i) {
        if (in[i].len % 7 == 3) { status[i] = kind ? 7 : 42; continue; }
        const size_t m = in[i].len;
        uint8_t* o = static_cast<uint8_t*>(malloc(m + 2));
        if (kind == 0) { o[0] = 0xcf; o[1] = 0x84; for (size_t k = 0; k < m; ++k) o[2 + k] = in[i].data[m - 1 - k]; outs[i].len = m + 2; }
        else { for (size_t k = 2; k < m; ++k) o[k - 2] = in[i].data[m - 1 - (k - 2)]; outs[i].len = m >= 2 ? m - 2 : 0; }
        outs[i].data = o; outs[i].cap = m + 2;
        status[i] = 0;
    }
    return 0;
}

static int dial(const char* path) {
    const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un a;
    memset(&a, 0, sizeof a);
    a.sun_family = AF_UNIX;
    strncpy(a.sun_path, path, sizeof a.sun_path - 1);
    if (connect(fd, reinterpret_cast<sockaddr*>(&a), sizeof a) != 0) { close(fd); return -1; }
    return fd;
}

static std::atomic<long> g_ok{0}, g_bad{0}, g_empty{0};

static void client(const char* path, unsigned seed, double seconds) {
    std::mt19937 rng(seed);
    const auto until = std::chrono::steady_clock::now() + std::chrono::duration<double>(seconds);
    while (std::chrono::steady_clock::now() < until) {
        const int mode = (int)(rng() % 10);
        const size_t n = mode == 7 ? 300000 : 2 + rng() % 20000;
        std::vector<uint8_t> msg(n);
        for (auto& b : msg) b = (uint8_t)rng();
        msg[0] = 0xff; msg[1] = 0xd8;
        if (mode == 8) { msg[0] = 'x'; msg[1] = 'y'; }           // unknown file type
        const int fd = dial(path);
        if (fd < 0) continue;
        const size_t send_n = mode == 5 ? n / 2 : n;              // 5: hang up mid-upload
        size_t off = 0;
        while (off < send_n) { const ssize_t w = send(fd, msg.data() + off, send_n - off, MSG_NOSIGNAL); if (w <= 0) break; off += (size_t)w; }
        if (mode == 5) { close(fd); continue; }
        shutdown(fd, SHUT_WR);
        if (mode == 6) { std::this_thread::sleep_for(std::chrono::milliseconds(30)); close(fd); continue; }   // never reads
        std::vector<uint8_t> ans;
        uint8_t buf[65536];
        for (;;) { const ssize_t r = recv(fd, buf, sizeof buf, 0); if (r <= 0) break; ans.insert(ans.end(), buf, buf + r); }
        close(fd);
        if (ans.empty()) { ++g_empty; continue; }
        bool good = ans.size() == n + 2 && ans[0] == 0xcf && ans[1] == 0x84;
        for (size_t k = 0; good && k < n; ++k) good = ans[2 + k] == msg[n - 1 - k];
        good ? ++g_ok : ++g_bad;
    }
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const char* path = argv[1];
    const double seconds = atof(argv[2]);
    lep_serve_options o;
    memset(&o, 0, sizeof o);
    o.uds_path = path;
    o.max_file_bytes = 200000;
    o.time_bound_ms = 400;
    o.max_batch = 64;
    o.batch_window_us = 1500;
    o.max_connections = 48;
    o.process = process;
    lep_server* srv = nullptr;
    if (lep_serve_start(&o, &srv) != 0) { fprintf(stderr, "cannot start\n"); return 3; }
    std::vector<std::thread> ts;
    for (unsigned i = 0; i < 24; ++i) ts.emplace_back(client, path, 1000 + i, seconds);
    for (auto& t : ts) t.join();
    lep_serve_stats st;
    lep_serve_get_stats(srv, &st);
    lep_serve_stop(srv);
    fprintf(stderr, "clients: %ld good answers, %ld wrong, %ld empty; server: %llu accepted, %llu answered, %llu failed, %llu rejected, %llu timed out, %ld batches\n",
            g_ok.load(), g_bad.load(), g_empty.load(), (unsigned long long)st.accepted, (unsigned long long)st.answered, (unsigned long long)st.failed,
            (unsigned long long)st.rejected, (unsigned long long)st.timed_out, g_batches.load());
    return g_bad.load() == 0 && g_ok.load() > 0 ? 0 : 1;
}
