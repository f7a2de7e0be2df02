// lep_serve.cc -- the serving surface: `lepton -socket / -listen` semantics as a batching daemon.
// The reference forks one process per accepted connection and lets the kernel multiplex them over the CPU cores
// (src/lepton/socket_serve.cc:86-116, 166-290).  A GPU wants the opposite: thousands of files per launch.  So one IO
// thread owns every socket (poll, non-blocking reads until the client's half-close, non-blocking writes), and one batcher
// thread turns whatever has arrived into ONE lep_compress_batch + ONE lep_decompress_batch call; while that batch is on
// the GPU the IO thread keeps receiving the next one.  Wire behaviour is the reference's: whole file in, half-close,
// whole file out, close; nothing but the close on failure; <name>.z0 / -zliblisten answers with stored-block zlib.
#include <arpa/inet.h>
#include <errno.h>
#include <fcntl.h>
#include <netinet/in.h>
#include <poll.h>
#include <signal.h>
#include <sys/file.h>
#include <sys/resource.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lepton_mi355x.h"

namespace {
using Clock = std::chrono::steady_clock;

struct Conn {
    int fd = -1;
    bool zlib = false;                     // accepted on the zlib listener
    enum State { READING, QUEUED, WRITING } state = READING;
    int kind = -1;                         // 0 JPEG -> .lep, 1 .lep -> JPEG
    std::vector<uint8_t> in, out;
    size_t out_pos = 0;
    bool has_deadline = false;
    Clock::time_point deadline;
    std::atomic<bool> dead{false};         // closed by the IO thread while the batcher held it
};
using ConnPtr = std::shared_ptr<Conn>;

void set_nonblock(int fd) {
    const int fl = fcntl(fd, F_GETFL, 0);
    if (fl >= 0) fcntl(fd, F_SETFL, fl | O_NONBLOCK);
}
void close_retry(int fd) { while (close(fd) < 0 && errno == EINTR) {} }

int default_process(void* user, int kind, const lep_bytes* in, int n, lep_bytes* outs, int32_t* status);
}  // namespace

struct lep_server {
    lep_serve_options opt{};
    std::string uds, zuds, lock_path;
    int lock_fd = -1;
    bool own_files = false;
    struct Listener { int fd; bool zlib; };
    std::vector<Listener> listeners;
    int wake_r = -1, wake_w = -1;
    std::thread io, batcher;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<ConnPtr> ready, done;       // IO -> batcher, batcher -> IO
    std::atomic<bool> stop{false};
    lep_serve_stats stats{};
    std::vector<ConnPtr> conns;            // IO thread only

    void wake() { const char c = 1; ssize_t r; do { r = write(wake_w, &c, 1); } while (r < 0 && errno == EINTR); }
    void io_loop();
    void batch_loop();
    void drop(const ConnPtr& c, uint64_t lep_serve_stats::*counter);
    void finish_upload(const ConnPtr& c);
};

namespace {

int default_process(void* user, int kind, const lep_bytes* in, int n, lep_bytes* outs, int32_t* status) {
    lep_server* s = static_cast<lep_server*>(user);
    if (!s->opt.gpu) return LEP_GPU_ERROR;   // no CPU fallback: a server without a device answers nothing
    return kind == 0 ? lep_compress_batch(s->opt.gpu, in, n, outs, status, &s->opt.batch, nullptr)
                     : lep_decompress_batch(s->opt.gpu, in, n, outs, status, &s->opt.batch, nullptr);
}

int listen_uds(const std::string& path, int backlog) {
    sockaddr_un a;
    memset(&a, 0, sizeof a);
    a.sun_family = AF_UNIX;
    if (path.size() + 1 > sizeof a.sun_path) return -1;
    memcpy(a.sun_path, path.c_str(), path.size());
    const int fd = socket(PF_UNIX, SOCK_STREAM, 0);
    if (fd < 0) return -1;
    if (bind(fd, reinterpret_cast<sockaddr*>(&a), sizeof a) != 0 || listen(fd, backlog) != 0) { close_retry(fd); return -1; }
    chmod(path.c_str(), 0666);   // socket_serve.cc:305
    set_nonblock(fd);
    return fd;
}

int listen_tcp(int port, int backlog) {
    const int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return -1;
    sockaddr_in a;
    memset(&a, 0, sizeof a);
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(INADDR_ANY);
    a.sin_port = htons((uint16_t)port);
    int one = 1;
    setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    if (bind(fd, reinterpret_cast<sockaddr*>(&a), sizeof a) != 0 || listen(fd, backlog) != 0) { close_retry(fd); return -1; }
    set_nonblock(fd);
    return fd;
}

}  // namespace

void lep_server::drop(const ConnPtr& c, uint64_t lep_serve_stats::*counter) {
    if (counter) { std::lock_guard<std::mutex> g(mu); ++(stats.*counter); }   // counted before the peer can see the close
    c->dead.store(true);
    if (c->fd >= 0) { close_retry(c->fd); c->fd = -1; }
}

// the client half-closed: classify by the first two bytes (jpgcoder.cc:2178-2235) and queue for the batcher
void lep_server::finish_upload(const ConnPtr& c) {
    const std::vector<uint8_t>& d = c->in;
    int kind = -1;
    if (d.size() >= 2) {
        if (d[0] == 0xff && d[1] == 0xd8) kind = 0;
        else if (d[0] == 0xcf && d[1] == 0x84) kind = 1;
        else if (d[0] == 0xce && d[1] == 0xb6) { kind = 1; c->zlib = true; c->in[0] = 0xcf; c->in[1] = 0x84; }   // zeta-lepton: zlib answer
    }
    if (kind < 0) {   // "filetype of file is unknown" (jpgcoder.cc:2230-2235); UJG files are a build option we do not have
        { std::lock_guard<std::mutex> g(mu); stats.last_failure_code = d.size() < 2 ? LEP_SHORT_READ : LEP_CODING_ERROR; }
        drop(c, &lep_serve_stats::failed);
        return;
    }
    c->kind = kind;
    c->state = Conn::QUEUED;
    {
        std::lock_guard<std::mutex> g(mu);
        stats.bytes_in += d.size();
        ready.push_back(c);
    }
    cv.notify_one();
}

void lep_server::io_loop() {
    const size_t max_bytes = opt.max_file_bytes ? opt.max_file_bytes : (256u << 20);
    std::vector<pollfd> pf;
    std::vector<int> who;   // index into conns, or -1 - listener index, or INT32_MIN for the wake pipe
    std::vector<uint8_t> buf(1 << 20);
    while (!stop.load()) {
        pf.clear(); who.clear();
        pf.push_back({wake_r, POLLIN, 0}); who.push_back(INT32_MIN);
        const bool room = opt.max_connections <= 0 || (int)conns.size() < opt.max_connections;
        if (room)
            for (size_t i = 0; i < listeners.size(); ++i) { pf.push_back({listeners[i].fd, POLLIN, 0}); who.push_back(-1 - (int)i); }
        int timeout = 1000;
        const Clock::time_point now0 = Clock::now();
        for (size_t i = 0; i < conns.size(); ++i) {
            Conn& c = *conns[i];
            if (c.state == Conn::READING) { pf.push_back({c.fd, POLLIN, 0}); who.push_back((int)i); }
            else if (c.state == Conn::WRITING) { pf.push_back({c.fd, POLLOUT, 0}); who.push_back((int)i); }
            if (c.has_deadline) {
                const long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(c.deadline - now0).count();
                timeout = (int)std::max(0l, std::min<long>(timeout, ms + 1));
            }
        }
        const int pr = poll(pf.data(), (nfds_t)pf.size(), timeout);
        if (pr < 0 && errno != EINTR) break;
        if (stop.load()) break;

        for (size_t k = 0; pr > 0 && k < pf.size(); ++k) {
            if (!pf[k].revents) continue;
            const int w = who[k];
            if (w == INT32_MIN) {
                char tmp[256];
                while (read(wake_r, tmp, sizeof tmp) > 0) {}
            } else if (w < 0) {   // a listener: take everything that is waiting (bounded by max_connections)
                const Listener& l = listeners[(size_t)(-1 - w)];
                for (;;) {
                    if (opt.max_connections > 0 && (int)conns.size() >= opt.max_connections) break;
                    const int fd = accept(l.fd, nullptr, nullptr);
                    if (fd < 0) {
                        if (errno == EMFILE || errno == ENFILE) usleep(1000);   // out of descriptors: do not spin on a readable listener
                        break;
                    }
                    set_nonblock(fd);
                    auto c = std::make_shared<Conn>();
                    c->fd = fd; c->zlib = l.zlib;
                    conns.push_back(c);
                    std::lock_guard<std::mutex> g(mu);
                    ++stats.accepted;
                }
            } else {
                const ConnPtr c = conns[(size_t)w];
                if (c->fd < 0) continue;
                if (c->state == Conn::READING) {
                    for (;;) {
                        const ssize_t n = read(c->fd, buf.data(), buf.size());
                        if (n > 0) {
                            if (!c->has_deadline && opt.time_bound_ms) {   // "enforce a timeout since first byte received"
                                c->has_deadline = true;
                                c->deadline = Clock::now() + std::chrono::milliseconds(opt.time_bound_ms);
                            }
                            if (c->in.size() + (size_t)n > max_bytes) { drop(c, &lep_serve_stats::rejected); break; }
                            c->in.insert(c->in.end(), buf.data(), buf.data() + n);
                            continue;
                        }
                        if (n == 0) { finish_upload(c); break; }
                        if (errno == EINTR) continue;
                        if (errno != EAGAIN && errno != EWOULDBLOCK) drop(c, &lep_serve_stats::failed);
                        break;
                    }
                } else if (c->state == Conn::WRITING) {
                    for (;;) {
                        if (c->out_pos == c->out.size()) {
                            { std::lock_guard<std::mutex> g(mu); ++stats.answered; stats.bytes_out += c->out.size(); }
                            drop(c, nullptr);
                            break;
                        }
                        const ssize_t n = send(c->fd, c->out.data() + c->out_pos, c->out.size() - c->out_pos, MSG_NOSIGNAL);
                        if (n > 0) { c->out_pos += (size_t)n; continue; }
                        if (n < 0 && errno == EINTR) continue;
                        if (n < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) break;
                        drop(c, &lep_serve_stats::failed);   // the client went away
                        break;
                    }
                }
            }
        }

        // answers from the batcher
        std::deque<ConnPtr> fin;
        { std::lock_guard<std::mutex> g(mu); fin.swap(done); }
        for (const ConnPtr& c : fin) {
            if (c->dead.load() || c->fd < 0) continue;
            if (c->out.empty()) { drop(c, &lep_serve_stats::failed); continue; }   // the reference's child died with a code: just the close
            c->state = Conn::WRITING;
            c->out_pos = 0;
        }
        // -timebound
        const Clock::time_point now = Clock::now();
        for (const ConnPtr& c : conns)
            if (c->fd >= 0 && c->has_deadline && now >= c->deadline) drop(c, &lep_serve_stats::timed_out);
        size_t keep = 0;
        for (size_t i = 0; i < conns.size(); ++i)
            if (conns[i]->fd >= 0) conns[keep++] = conns[i];
        conns.resize(keep);
    }
    for (const ConnPtr& c : conns) drop(c, nullptr);
    conns.clear();
}

void lep_server::batch_loop() {
    const size_t max_batch = opt.max_batch > 0 ? (size_t)opt.max_batch : 1024;
    const auto window = std::chrono::microseconds(opt.batch_window_us > 0 ? opt.batch_window_us : 2000);
    lep_serve_process_fn fn = opt.process ? opt.process : default_process;
    void* user = opt.process ? opt.process_user : this;
    for (;;) {
        std::vector<ConnPtr> batch;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return stop.load() || !ready.empty(); });
            if (stop.load()) return;
            // a request is complete: give the others that are in flight a moment to join its launch
            // (system_clock: pthread_cond_timedwait, which ThreadSanitizer models -- the steady-clock wait is pthread_cond_clockwait,
            //  which the gcc 11 runtime does not, and every lock after it would be reported)
            cv.wait_until(lk, std::chrono::system_clock::now() + window, [&] { return stop.load() || ready.size() >= max_batch; });
            if (stop.load()) return;
            while (!ready.empty() && batch.size() < max_batch) { batch.push_back(ready.front()); ready.pop_front(); }
        }
        size_t live = 0;
        for (int kind = 0; kind < 2; ++kind) {
            std::vector<Conn*> group;
            for (const ConnPtr& c : batch)
                if (c->kind == kind && !c->dead.load()) group.push_back(c.get());
            if (group.empty()) continue;
            live += group.size();
            const int n = (int)group.size();
            std::vector<lep_bytes> in((size_t)n), outs((size_t)n);
            std::vector<int32_t> status((size_t)n, LEP_CODING_ERROR);
            for (int i = 0; i < n; ++i) { in[(size_t)i] = {group[(size_t)i]->in.data(), group[(size_t)i]->in.size(), group[(size_t)i]->in.size()}; outs[(size_t)i] = {nullptr, 0, 0}; }
            const int rc = fn(user, kind, in.data(), n, outs.data(), status.data());
            for (int i = 0; i < n; ++i) {
                Conn& c = *group[(size_t)i];
                const int code = rc ? rc : status[(size_t)i];
                lep_bytes& o = outs[(size_t)i];
                if (code == 0 && o.data && o.len) {
                    if (kind == 1 && c.zlib) {
                        lep_bytes z{nullptr, 0, 0};
                        if (lep_zlib0_wrap(o.data, o.len, &z) == 0) { c.out.assign(z.data, z.data + z.len); free(z.data); }
                    } else {
                        c.out.assign(o.data, o.data + o.len);
                    }
                } else {
                    fprintf(stderr, "request (%zu bytes, %s) failed with code %d\n", c.in.size(), kind ? "lepton" : "jpeg", code ? code : LEP_CODING_ERROR);
                    std::lock_guard<std::mutex> g(mu);
                    stats.last_failure_code = code ? code : LEP_CODING_ERROR;
                }
                if (o.data) free(o.data);
                std::vector<uint8_t>().swap(c.in);
            }
        }
        {
            std::lock_guard<std::mutex> g(mu);
            if (live) { ++stats.batches; stats.largest_batch = std::max<uint64_t>(stats.largest_batch, live); }
            for (const ConnPtr& c : batch) done.push_back(c);
        }
        wake();
    }
}

extern "C" {

int lep_zlib0_wrap(const uint8_t* data, size_t len, lep_bytes* out) {
    // Zlib0Writer (src/io/Zlib0.cc:36-120): 78 01, stored blocks of 65535 bytes, the last one (1..65535 bytes, or 0 for an
    // empty input) flagged final, then the Adler-32 big-endian
    const size_t kChunk = 65535;
    const size_t nchunks = len ? (len + kChunk - 1) / kChunk : 1;
    const size_t total = 2 + nchunks * 5 + len + 4;
    uint8_t* o = static_cast<uint8_t*>(malloc(total));
    if (!o) return LEP_OS_ERROR;
    size_t p = 0;
    o[p++] = 0x78; o[p++] = 0x01;
    size_t pos = 0;
    for (size_t k = 0; k < nchunks; ++k) {
        const size_t n = std::min(kChunk, len - pos);
        o[p++] = k + 1 == nchunks ? 1 : 0;
        o[p++] = (uint8_t)(n & 0xff); o[p++] = (uint8_t)(n >> 8);
        o[p++] = (uint8_t)(~n & 0xff); o[p++] = (uint8_t)((~n >> 8) & 0xff);
        if (n) memcpy(o + p, data + pos, n);
        p += n; pos += n;
    }
    uLong ad = adler32(0L, Z_NULL, 0);
    for (size_t q = 0; q < len;) { const size_t n = std::min<size_t>(len - q, 1u << 30); ad = adler32(ad, data + q, (uInt)n); q += n; }
    o[p++] = (uint8_t)(ad >> 24); o[p++] = (uint8_t)(ad >> 16); o[p++] = (uint8_t)(ad >> 8); o[p++] = (uint8_t)ad;
    out->data = o; out->len = p; out->cap = total;
    return 0;
}

int lep_serve_start(const lep_serve_options* opt, lep_server** out) {
    if (!opt || !out) return LEP_ASSERTION_FAILURE;
    *out = nullptr;
    if (!opt->uds_path && !opt->tcp_port && !opt->zlib_tcp_port) return LEP_ASSERTION_FAILURE;
    std::unique_ptr<lep_server> s(new lep_server);
    s->opt = *opt;
    {   // one descriptor per connection in flight (the reference spends a process on each): lift the soft limit
        rlimit rl;
        if (getrlimit(RLIMIT_NOFILE, &rl) == 0 && rl.rlim_cur < rl.rlim_max) { rl.rlim_cur = rl.rlim_max; (void)setrlimit(RLIMIT_NOFILE, &rl); }
    }
    const int backlog = opt->listen_backlog > 0 ? opt->listen_backlog : 16;
    auto fail = [&](int code) {
        for (auto& l : s->listeners) close_retry(l.fd);
        if (s->own_files) { unlink(s->uds.c_str()); unlink(s->zuds.c_str()); }
        if (s->lock_fd >= 0) close_retry(s->lock_fd);
        return code;
    };
    if (opt->uds_path) {
        s->uds = opt->uds_path;
        s->zuds = opt->zlib_uds_path ? std::string(opt->zlib_uds_path) : s->uds + ".z0";
        s->lock_path = s->uds + ".lock";
        s->opt.uds_path = s->uds.c_str();
        s->opt.zlib_uds_path = s->zuds.c_str();
        // whoever holds the lock owns the name and may remove stale socket files (socket_serve.cc:331-356)
        do { s->lock_fd = open(s->lock_path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644); } while (s->lock_fd < 0 && errno == EINTR);
        if (s->lock_fd < 0) return LEP_OS_ERROR;
        int err;
        do { err = flock(s->lock_fd, LOCK_EX | LOCK_NB); } while (err < 0 && errno == EINTR);
        if (err != 0) return fail(LEP_OS_ERROR);   // a live server owns this name
        unlink(s->uds.c_str());
        unlink(s->zuds.c_str());
        s->own_files = true;
        const int a = listen_uds(s->uds, backlog), b = listen_uds(s->zuds, backlog);
        if (a >= 0) s->listeners.push_back({a, false});
        if (b >= 0) s->listeners.push_back({b, true});
        if (a < 0 || b < 0) return fail(LEP_OS_ERROR);
    }
    if (opt->tcp_port) {
        const int a = listen_tcp(opt->tcp_port, backlog);
        if (a < 0) return fail(LEP_OS_ERROR);   // COULD_NOT_BIND_PORT in the reference
        s->listeners.push_back({a, false});
    }
    if (opt->zlib_tcp_port) {
        const int a = listen_tcp(opt->zlib_tcp_port, backlog);
        if (a < 0) return fail(LEP_OS_ERROR);
        s->listeners.push_back({a, true});
    }
    int p[2];
    if (pipe(p) != 0) return fail(LEP_OS_ERROR);
    s->wake_r = p[0]; s->wake_w = p[1];
    set_nonblock(s->wake_r); set_nonblock(s->wake_w);
    lep_server* raw = s.release();
    raw->io = std::thread([raw] { raw->io_loop(); });
    raw->batcher = std::thread([raw] { raw->batch_loop(); });
    *out = raw;
    return 0;
}

void lep_serve_get_stats(lep_server* s, lep_serve_stats* out) {
    if (!s || !out) return;
    std::lock_guard<std::mutex> g(s->mu);
    *out = s->stats;
}

void lep_serve_stop(lep_server* s) {
    if (!s) return;
    { std::lock_guard<std::mutex> g(s->mu); s->stop.store(true); }
    s->cv.notify_all();
    s->wake();
    if (s->io.joinable()) s->io.join();
    if (s->batcher.joinable()) s->batcher.join();
    for (auto& l : s->listeners) close_retry(l.fd);
    if (s->own_files) { unlink(s->uds.c_str()); unlink(s->zuds.c_str()); unlink(s->lock_path.c_str()); }
    if (s->lock_fd >= 0) close_retry(s->lock_fd);
    close_retry(s->wake_r); close_retry(s->wake_w);
    delete s;
}

}  // extern "C"
