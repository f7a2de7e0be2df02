// lepton_served -- the daemon around lep_serve_start: the reference's serving flags
// (src/lepton/jpgcoder.cc:1026-1186, help text :2089-2110) in front of the GPU batch pipeline.  One process per GPU.
//   lepton_served -socket[=name] [-listen[=port]] [-zliblisten=port] [-listenbacklog=n] [-maxchildren=n]
//                 [-timebound=<n>ms|s|us] [-skipverify|-verify] [-device=k | -devices=a,b,...] [-maxbatch=n]
//                 [-batchwindow=<us>] [-hosthuffman]
// -devices=0,1,...: one serving process per GPU (forked before any HIP call), sockets <name>.<device> (TCP: port + device,
// zlib port + device); the parent only waits and passes SIGTERM / SIGINT / SIGQUIT on.
// Prints the socket name on stdout once it is listening (socket_serve.cc:383-384) and serves until SIGINT / SIGTERM /
// SIGQUIT, removing its socket files on the way out (cleanup_socket, socket_serve.cc:71-84).
#include <signal.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lepton_mi355x.h"

static volatile sig_atomic_t g_quit = 0;
static void on_signal(int) { g_quit = 1; }

static bool starts(const char* a, const char* p) { return strncmp(a, p, strlen(p)) == 0; }

int main(int argc, char** argv) {
    lep_serve_options o;
    memset(&o, 0, sizeof o);
    o.batch.verify = 1;   // the reference round-trips every compression unless told -skipverify (jpgcoder.cc:1598-1690)
    std::string uds, zuds;
    bool want_uds = false;
    int device = 0;
    std::vector<int> devices;
    for (int i = 1; i < argc; ++i) {
        const char* a = argv[i];
        if (starts(a, "-socket")) { want_uds = true; if (a[7] == '=') uds = a + 8; }
        else if (starts(a, "-listenbacklog=")) o.listen_backlog = atoi(a + 15);
        else if (starts(a, "-listen")) { o.tcp_port = a[7] == '=' ? atoi(a + 8) : 2402; if (!o.zlib_tcp_port) o.zlib_tcp_port = 2403; }
        else if (starts(a, "-zliblisten=")) o.zlib_tcp_port = atoi(a + 12);
        else if (starts(a, "-maxchildren=")) o.max_connections = atoi(a + 13);
        else if (starts(a, "-timebound=")) {
            char* end = nullptr;
            unsigned long long v = strtoull(a + 11, &end, 10);
            if (end && !strcmp(end, "s")) v *= 1000;
            else if (end && !strcmp(end, "us")) v /= 1000;
            o.time_bound_ms = (uint32_t)v;
        }
        else if (!strcmp(a, "-skipverify") || !strcmp(a, "-skipvalidate")) o.batch.verify = 0;
        else if (!strcmp(a, "-verify") || !strcmp(a, "-validate")) o.batch.verify = 1;
        else if (starts(a, "-devices=")) { for (const char* q = a + 9; *q;) { devices.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q) ++q; } }
        else if (starts(a, "-device=")) device = atoi(a + 8);
        else if (starts(a, "-maxbatch=")) o.max_batch = atoi(a + 10);
        else if (starts(a, "-batchwindow=")) o.batch_window_us = atoi(a + 13);
        else if (!strcmp(a, "-hosthuffman")) o.batch.host_huffman = 1;
        else if (!strcmp(a, "-preload") || !strcmp(a, "-unjailed") || !strcmp(a, "-allowprogressive") || !strcmp(a, "-singlethread")) {}   // no-ops here
        else { fprintf(stderr, "lepton_served: unknown option %s\n", a); return 1; }
    }
    if (!want_uds && !o.tcp_port) { fprintf(stderr, "usage: lepton_served -socket[=name] | -listen[=port] [...]\n"); return 1; }
    if (o.time_bound_ms && !want_uds) { fprintf(stderr, "Time bound action only supported with UNIX domain sockets\n"); return 1; }   // jpgcoder.cc:1209-1212
    if (devices.size() > 1) {   // one process per GPU, each with its own sockets; this process only supervises
        if (want_uds && uds.empty()) { fprintf(stderr, "lepton_served: -devices needs -socket=<name> (children listen on <name>.<device>)\n"); return 1; }
        // Supervision: a child that ends while the others serve is reported; one that had been serving (alive for 2 s or
        // more, or killed by a signal) is started again -- a GPU that falls over must not take its socket away for good --
        // at most 5 times per device; one that ends at once (no such device, socket name taken) stays down.
        struct Kid { pid_t pid; int dev; double born; int restarts; };
        std::vector<Kid> kids;
        auto now_s = []() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + t.tv_nsec * 1e-9; };
        bool is_child = false;
        auto spawn = [&](int dev, int restarts) {
            const pid_t pid = fork();
            if (pid == 0) {
                device = dev;
                if (want_uds) uds += "." + std::to_string(dev);
                if (o.tcp_port) o.tcp_port += dev;
                if (o.zlib_tcp_port) o.zlib_tcp_port += dev;
                is_child = true;
            } else if (pid > 0) kids.push_back(Kid{pid, dev, now_s(), restarts});
            else fprintf(stderr, "lepton_served: fork for device %d failed\n", dev);
        };
        for (int dev : devices) { spawn(dev, 0); if (is_child) break; }
        if (!is_child) {
            struct sigaction sa;
            memset(&sa, 0, sizeof sa);
            sa.sa_handler = on_signal;
            sigaction(SIGINT, &sa, nullptr); sigaction(SIGTERM, &sa, nullptr); sigaction(SIGQUIT, &sa, nullptr);
            int worst = 0;
            while (!kids.empty() && !g_quit && !is_child) {
                int st = 0;
                const pid_t r = waitpid(-1, &st, 0);
                if (r <= 0) continue;   // EINTR: a signal for us, g_quit says which
                for (size_t k = 0; k < kids.size(); ++k) {
                    if (kids[k].pid != r) continue;
                    const Kid kid = kids[k];
                    kids.erase(kids.begin() + k);
                    const bool was_serving = now_s() - kid.born >= 2.0 || WIFSIGNALED(st);
                    if (WIFEXITED(st) && WEXITSTATUS(st)) worst = WEXITSTATUS(st);
                    fprintf(stderr, "lepton_served: child %d (device %d) ended (%s %d)%s\n", (int)r, kid.dev, WIFSIGNALED(st) ? "signal" : "exit code",
                            WIFSIGNALED(st) ? WTERMSIG(st) : WEXITSTATUS(st), was_serving && kid.restarts < 5 && !g_quit ? ", restarting" : "");
                    if (was_serving && kid.restarts < 5 && !g_quit) { usleep(200000); spawn(kid.dev, kid.restarts + 1); }
                    break;
                }
            }
            if (!is_child) {
                for (const Kid& k : kids) kill(k.pid, SIGTERM);
                for (;;) { int st = 0; const pid_t r = waitpid(-1, &st, 0); if (r <= 0) break; if (WIFEXITED(st) && WEXITSTATUS(st)) worst = WEXITSTATUS(st); }
                return worst;
            }
        }
        devices.clear();
    } else if (devices.size() == 1) device = devices[0];
    if (want_uds && uds.empty()) {   // /tmp/<random id>.uport and .z0 (name_socket, socket_serve.cc:40-69)
        unsigned char r[16] = {0};
        if (FILE* f = fopen("/dev/urandom", "rb")) { size_t n = fread(r, 1, sizeof r, f); (void)n; fclose(f); }
        std::string base = "/tmp/";
        char hex[3];
        for (int i = 0; i < 16; ++i) {
            snprintf(hex, sizeof hex, "%02x", r[i]);
            base += hex;
            if (i == 4 || i == 6 || i == 8 || i == 14) base += '-';
        }
        uds = base + ".uport";
        zuds = base + ".z0";
    }
    if (want_uds) { o.uds_path = uds.c_str(); if (!zuds.empty()) o.zlib_uds_path = zuds.c_str(); }

    lep_gpu* gpu = nullptr;
    // LEP_SERVED_NO_DEVICE=1 (supervision tests on machines without a GPU): listen and answer every request with a failure
    // instead of exiting -- nothing is ever coded without a device
    const bool no_device = getenv("LEP_SERVED_NO_DEVICE") && atoi(getenv("LEP_SERVED_NO_DEVICE")) != 0;
    int rc = no_device ? 0 : lep_gpu_create(device, &gpu);
    if (rc) { fprintf(stderr, "lepton_served: no usable gfx950 device %d (code %d)\n", device, rc); return rc; }
    o.gpu = gpu;
    signal(SIGPIPE, SIG_IGN);
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_signal;
    sigaction(SIGINT, &sa, nullptr); sigaction(SIGTERM, &sa, nullptr); sigaction(SIGQUIT, &sa, nullptr);
    lep_server* srv = nullptr;
    rc = lep_serve_start(&o, &srv);
    if (rc) { lep_gpu_destroy(gpu); return rc; }   // e.g. another server holds <name>.lock: nothing is printed, like the reference
    if (want_uds) fprintf(stdout, "%s\n", uds.c_str());
    else fprintf(stdout, "%d\n", o.tcp_port);
    fflush(stdout);
    while (!g_quit) pause();
    lep_serve_stats st;
    lep_serve_get_stats(srv, &st);
    lep_serve_stop(srv);
    lep_batch_release();
    if (gpu) lep_gpu_destroy(gpu);
    fprintf(stderr, "lepton_served: %llu accepted, %llu answered, %llu failed, %llu timed out, %llu batches (largest %llu)\n",
            (unsigned long long)st.accepted, (unsigned long long)st.answered, (unsigned long long)st.failed,
            (unsigned long long)st.timed_out, (unsigned long long)st.batches, (unsigned long long)st.largest_batch);
    return 0;
}
