// ss_comm.cpp — the one exchange step of the path: SUM all-reduce of the corpus histograms.
//
// One process per GPU; streams are sharded over the ranks with no data-path exchange, and the corpus-level
// integrated-LUFS gate needs exactly one collective: the element-wise sum of every rank's two 1000-bin u64
// histograms (ebur128 loudness_global_multiple semantics).  The reference has no collective (it analyses one
// file on one thread); this is north_star's extension.
//
// Transport SS_COMM_RCCL: the library opens librccl itself and issues
//     ncclAllReduce(buf, buf, 2000, ncclUint64, ncclSum, comm, batch_stream)
// in place on the batch's device histograms — 16 000 bytes, latency-bound on xGMI.  No PyTorch, no MPI.
// Transport SS_COMM_HOST_TCP: the same entry points staged through host memory over loopback TCP (star through
// rank 0).  It exists so the rank logic can be run on CPU-only machines and by ranks that share one GPU.
//
// Rendezvous (one node): rank 0 listens on an ephemeral loopback port and writes
//     "<magic> <tcp port> <rccl unique id in hex> <64-bit nonce>"
// to the rendezvous file (created with O_EXCL | O_NOFOLLOW, mode 0600, then renamed into place, so readers never see a
// partial file and a planted symlink is not followed).  Every other rank polls for the file, connects to the port and
// sends (its rank, the nonce it read); rank 0 answers with the nonce.  A file left behind by a crashed or earlier job
// therefore cannot be consumed: its port is dead (connection refused) or answers with another nonce, and the rank
// simply keeps polling until the live rank 0 has replaced the file.  For SS_COMM_HOST_TCP those connections are the
// data path; for SS_COMM_RCCL they carry only the join handshake and the ranks' verdicts on ncclCommInitRank (all ranks
// succeed or all fail), which itself runs under a watchdog.  Rank 0 removes the file once every rank has joined.
#include "../../include/soundscope_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and enum values only: the functions are resolved with dlsym at ss_comm_init

#include <arpa/inet.h>
#include <dlfcn.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <fcntl.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "ss_internal.h"

namespace {

// RCCL shares device buffers between the ranks' processes through HSA IPC.  Hosts whose driver supports only dmabuf IPC
// need HSA_ENABLE_IPC_MODE_LEGACY=0 (without it ncclCommInitRank fails with "hipIpcGetMemHandle: invalid argument"), and the
// HSA runtime reads the variable once, when the first HIP call starts it.  The library does NOT touch the process
// environment when it is loaded: ss_comm_init (RCCL transport, more than one rank) puts the default in place if the
// variable is unset — in time when ss_comm_init comes before the process' first HIP call (the order INTEGRATION.md
// prescribes for launchers; bench.py and the C client export it themselves), harmless otherwise.
void default_hsa_ipc_mode() { ::setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0); }

constexpr const char *kMagic = "ssc2";
// SS_COMM_TIMEOUT_S (seconds, default 180): how long a rank waits for the others at the rendezvous and inside ncclCommInitRank
int join_timeout_s()
{
    if (const char *e = std::getenv("SS_COMM_TIMEOUT_S")) { const int v = std::atoi(e); if (v >= 1 && v <= 3600) return v; }
    return 180;
}

uint64_t fresh_nonce()
{
    uint64_t v = 0;
    if (FILE *f = std::fopen("/dev/urandom", "rb")) { if (std::fread(&v, sizeof v, 1, f) != 1) v = 0; std::fclose(f); }
    if (!v) v = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9E3779B97F4A7C15ull ^ (uint64_t)::getpid();
    return v ? v : 1;
}

struct Rccl {
    void *dl = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;           // optional
};

bool rccl_open(Rccl &r, std::string &err)
{
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        r.dl = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.dl) break;
    }
    if (!r.dl) { err = std::string("dlopen(librccl): ") + (dlerror() ? dlerror() : "not found"); return false; }
    auto sym = [&](const char *n) { return dlsym(r.dl, n); };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.CommCount || !r.AllReduce || !r.GetErrorString) {
        err = "librccl lacks an expected nccl* symbol";
        return false;
    }
    return true;
}

// ---- blocking socket helpers --------------------------------------------------------------------------------
bool send_all(int fd, const void *buf, size_t n)
{
    const char *p = static_cast<const char *>(buf);
    while (n) {
        const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
        if (k < 0) { if (errno == EINTR) continue; return false; }
        p += k; n -= (size_t)k;
    }
    return true;
}

bool recv_all(int fd, void *buf, size_t n)
{
    char *p = static_cast<char *>(buf);
    while (n) {
        const ssize_t k = ::recv(fd, p, n, 0);
        if (k < 0) { if (errno == EINTR) continue; return false; }
        if (k == 0) return false;
        p += k; n -= (size_t)k;
    }
    return true;
}

std::string hex_of(const unsigned char *p, size_t n)
{
    static const char *d = "0123456789abcdef";
    std::string s(2 * n, '0');
    for (size_t i = 0; i < n; i++) { s[2 * i] = d[p[i] >> 4]; s[2 * i + 1] = d[p[i] & 15]; }
    return s;
}

bool unhex(const std::string &s, unsigned char *p, size_t n)
{
    if (s.size() != 2 * n) return false;
    auto v = [](char c) { return c >= '0' && c <= '9' ? c - '0' : (c >= 'a' && c <= 'f' ? c - 'a' + 10 : -1); };
    for (size_t i = 0; i < n; i++) {
        const int a = v(s[2 * i]), b = v(s[2 * i + 1]);
        if (a < 0 || b < 0) return false;
        p[i] = (unsigned char)(a * 16 + b);
    }
    return true;
}

bool write_rendezvous(const std::string &path, int port, const ncclUniqueId *id, uint64_t nonce)
{
    const std::string tmp = path + ".tmp." + std::to_string((long)::getpid());
    ::unlink(tmp.c_str());
    const int fd = ::open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return false;
    const std::string hex = id ? hex_of(reinterpret_cast<const unsigned char *>(id->internal), sizeof id->internal) : std::string("-");
    char head[64];
    std::snprintf(head, sizeof head, "%s %d ", kMagic, port);
    const std::string text = std::string(head) + hex + " " + std::to_string((unsigned long long)nonce) + "\n";
    const bool ok = ::write(fd, text.data(), text.size()) == (ssize_t)text.size();
    ::close(fd);
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) { ::unlink(tmp.c_str()); return false; }
    return true;
}

bool read_rendezvous(const std::string &path, int *port, std::string *hex, uint64_t *nonce)
{
    const int fd = ::open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return false;
    FILE *f = ::fdopen(fd, "r");
    if (!f) { ::close(fd); return false; }
    char magic[16] = {0}, idbuf[600] = {0};
    int p = 0;
    unsigned long long nn = 0;
    const int n = std::fscanf(f, "%15s %d %599s %llu", magic, &p, idbuf, &nn);
    std::fclose(f);
    if (n != 4 || std::strcmp(magic, kMagic) != 0 || p <= 0 || nn == 0) return false;
    *port = p; *hex = idbuf; *nonce = (uint64_t)nn;
    return true;
}

}  // namespace

struct ss_comm {
    int transport = SS_COMM_RCCL, rank = 0, world = 1, device = 0;
    std::string file;
    // RCCL
    Rccl rccl;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    void *dev_scratch = nullptr;           // kScratchBytes, for the small host-buffer collectives
    // host TCP (star through rank 0)
    int listen_fd = -1;
    std::vector<int> peers;                // rank 0: socket of rank r at [r]; others: [0] = socket to rank 0
    static constexpr size_t kScratchBytes = 1 << 16;
};

namespace {

int fail(const std::string &text)
{
    ssi::set_last_error(text);
    return SS_ERR_DEVICE;
}

#define COMM_HIP(expr)                                                            \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) return fail(std::string(#expr ": ") + hipGetErrorString(e_)); \
    } while (0)
#define COMM_NCCL(c, expr)                                                        \
    do {                                                                          \
        ncclResult_t r_ = (expr);                                                 \
        if (r_ != ncclSuccess) return fail(std::string(#expr ": ") + (c)->rccl.GetErrorString(r_)); \
    } while (0)

// ---- host TCP: every rank connects to rank 0 -----------------------------------------------------------------
int tcp_listen(ss_comm *c, int *port_out)
{
    c->listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (c->listen_fd < 0) return fail("socket() failed");
    sockaddr_in a{};
    a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = 0;     // ephemeral port on 127.0.0.1
    if (::bind(c->listen_fd, reinterpret_cast<sockaddr *>(&a), sizeof a) != 0 || ::listen(c->listen_fd, 128) != 0)
        return fail("bind/listen on 127.0.0.1 failed");
    socklen_t len = sizeof a;
    ::getsockname(c->listen_fd, reinterpret_cast<sockaddr *>(&a), &len);
    *port_out = ntohs(a.sin_port);
    return SS_OK;
}

// rank 0: accept world - 1 joins.  A connection that does not present (rank, this init's nonce) is not one of ours
// (a rank that read a stale file, a port scanner): it is dropped and the wait goes on.
int tcp_accept_all(ss_comm *c, uint64_t nonce)
{
    c->peers.assign((size_t)c->world, -1);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(join_timeout_s());
    for (int joined = 1; joined < c->world;) {
        const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
        if (left <= 0) return fail("rank 0: timed out waiting for the other ranks");
        timeval tv{(time_t)(left / 1000), (suseconds_t)((left % 1000) * 1000)};
        fd_set rd; FD_ZERO(&rd); FD_SET(c->listen_fd, &rd);
        if (::select(c->listen_fd + 1, &rd, nullptr, nullptr, &tv) <= 0) continue;
        const int fd = ::accept(c->listen_fd, nullptr, nullptr);
        if (fd < 0) continue;
        int one = 1;
        ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        timeval rto{5, 0};
        ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &rto, sizeof rto);
        uint64_t hello[2] = {0, 0};                 // (rank, nonce)
        if (!recv_all(fd, hello, sizeof hello) || hello[1] != nonce || hello[0] == 0 || hello[0] >= (uint64_t)c->world ||
            c->peers[(size_t)hello[0]] != -1) { ::close(fd); continue; }
        if (!send_all(fd, &nonce, sizeof nonce)) { ::close(fd); continue; }
        timeval none{0, 0};
        ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &none, sizeof none);
        c->peers[(size_t)hello[0]] = fd;
        joined++;
    }
    return SS_OK;
}

// other ranks: 0 joined; 1 the file is stale (nobody listens there, or whoever does is not this job's rank 0): poll again
int tcp_try_join(ss_comm *c, int port, uint64_t nonce)
{
    const int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return fail("socket() failed");
    sockaddr_in a{};
    a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = htons((uint16_t)port);
    if (::connect(fd, reinterpret_cast<sockaddr *>(&a), sizeof a) != 0) { ::close(fd); return 1; }
    int one = 1;
    ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    timeval rto{5, 0};
    ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &rto, sizeof rto);
    const uint64_t hello[2] = {(uint64_t)c->rank, nonce};
    uint64_t echo = 0;
    if (!send_all(fd, hello, sizeof hello) || !recv_all(fd, &echo, sizeof echo) || echo != nonce) { ::close(fd); return 1; }
    timeval none{0, 0};
    ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &none, sizeof none);
    c->peers.assign(1, fd);
    return 0;
}

// element-wise reduction of `n` 8-byte words across the ranks, in place (op 0: u64 sum, 1: f64 max)
int tcp_allreduce(ss_comm *c, void *buf, size_t n, int op)
{
    if (c->world == 1) return SS_OK;
    const size_t bytes = n * 8;
    if (c->rank == 0) {
        std::vector<unsigned char> tmp(bytes);
        for (int r = 1; r < c->world; r++) {
            if (!recv_all(c->peers[(size_t)r], tmp.data(), bytes)) return fail("host all-reduce: receive failed");
            if (op == 0) {
                uint64_t *a = static_cast<uint64_t *>(buf);
                const uint64_t *b = reinterpret_cast<const uint64_t *>(tmp.data());
                for (size_t i = 0; i < n; i++) a[i] += b[i];
            } else {
                double *a = static_cast<double *>(buf);
                const double *b = reinterpret_cast<const double *>(tmp.data());
                for (size_t i = 0; i < n; i++) a[i] = b[i] > a[i] ? b[i] : a[i];
            }
        }
        for (int r = 1; r < c->world; r++)
            if (!send_all(c->peers[(size_t)r], buf, bytes)) return fail("host all-reduce: send failed");
    } else {
        if (!send_all(c->peers[0], buf, bytes) || !recv_all(c->peers[0], buf, bytes)) return fail("host all-reduce: exchange with rank 0 failed");
    }
    return SS_OK;
}

// RCCL all-reduce of a small host buffer through the communicator's device scratch
int rccl_allreduce_host(ss_comm *c, void *buf, size_t n, ncclDataType_t dt, ncclRedOp_t op)
{
    const size_t bytes = n * 8;
    if (bytes > ss_comm::kScratchBytes) return SS_ERR_CAPACITY;
    COMM_HIP(hipMemcpyAsync(c->dev_scratch, buf, bytes, hipMemcpyHostToDevice, c->stream));
    COMM_NCCL(c, c->rccl.AllReduce(c->dev_scratch, c->dev_scratch, n, dt, op, c->comm, c->stream));
    COMM_HIP(hipMemcpyAsync(buf, c->dev_scratch, bytes, hipMemcpyDeviceToHost, c->stream));
    COMM_HIP(hipStreamSynchronize(c->stream));
    return SS_OK;
}

struct DeviceScope {
    int prev = -1; bool switched = false;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = (hipSetDevice(device) == hipSuccess);
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};

}  // namespace

extern "C" {

void ss_comm_destroy(ss_comm *c)
{
    if (!c) return;
    if (c->transport == SS_COMM_RCCL) {
        DeviceScope ds(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (c->comm && c->rccl.CommDestroy) (void)c->rccl.CommDestroy(c->comm);
        if (c->dev_scratch) (void)hipFree(c->dev_scratch);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        // librccl stays loaded: unloading a library that owns device state is not worth the risk
    }
    for (int fd : c->peers) if (fd >= 0) ::close(fd);
    if (c->listen_fd >= 0) ::close(c->listen_fd);
    if (c->rank == 0 && !c->file.empty()) ::unlink(c->file.c_str());
    delete c;
}

// device_wraps: `device` came from LOCAL_RANK — a value beyond the visible devices maps onto them (see ss_comm_init_from_env);
// env_error: the environment named no usable device at all (this rank's verdict, exchanged like any other local failure)
static int comm_init_impl(int transport, int rank, int world, int device, const char *rendezvous_file, ss_comm **out,
                          bool device_wraps = false, const char *env_error = nullptr)
{
    if (!out) return SS_ERR_INVALID_ARG;
    *out = nullptr;
    if ((transport != SS_COMM_RCCL && transport != SS_COMM_HOST_TCP) || world < 1 || rank < 0 || rank >= world) return SS_ERR_INVALID_ARG;
    if (world > 1 && (!rendezvous_file || !*rendezvous_file)) return SS_ERR_INVALID_ARG;
    struct Guard { ss_comm *c; ~Guard() { if (c) ss_comm_destroy(c); } } g{new ss_comm()};
    ss_comm *c = g.c;
    c->transport = transport; c->rank = rank; c->world = world;
    c->file = rendezvous_file ? rendezvous_file : "";
    std::string err;
    ncclUniqueId id;
    std::memset(&id, 0, sizeof id);
    // What can go wrong on THIS rank before the ranks have met — no such device, no librccl, no memory — is kept as a verdict
    // (world > 1) and exchanged right behind the join: every rank then fails within seconds with the reason, instead of the
    // healthy ones waiting out SS_COMM_TIMEOUT_S at the rendezvous or inside ncclCommInitRank for a rank that has already left
    // (tools/launch_path_two_ranks.sh: a rank whose LOCAL_RANK names no device cost its peer 2 x 180 s).
    std::string local_why;
    auto local_setup = [&]() -> bool {
        if (transport != SS_COMM_RCCL) return true;
        if (world > 1) default_hsa_ipc_mode();            // before this call's (possibly the process' first) HIP call
        if (env_error) { local_why = env_error; return false; }
        const int n_dev = ss_device_count();
        if (n_dev <= 0) { local_why = "no HIP device"; return false; }
        if (device_wraps && device >= n_dev) device = device % n_dev;      // one masked GPU per rank: every rank's device 0
        // device >= 0 (ss_comm_init_on_device, ss_comm_init_from_env): this call makes the rank's GPU current ITSELF, behind the
        // setenv above — the order a rank off device 0 cannot get by calling ss_set_device first (that call starts the HSA runtime)
        hipError_t e = device >= 0 ? hipSetDevice(device) : hipSuccess;
        if (e != hipSuccess) { local_why = "hipSetDevice(" + std::to_string(device) + "): " + hipGetErrorString(e); return false; }
        e = hipGetDevice(&c->device);
        if (e != hipSuccess) { local_why = std::string("hipGetDevice: ") + hipGetErrorString(e); return false; }
        if (!rccl_open(c->rccl, local_why)) return false;
        if (rank == 0) {
            const ncclResult_t r = c->rccl.GetUniqueId(&id);
            if (r != ncclSuccess) { local_why = std::string("ncclGetUniqueId: ") + c->rccl.GetErrorString(r); return false; }
        }
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(&c->dev_scratch, ss_comm::kScratchBytes);
        if (e != hipSuccess) { local_why = std::string("stream / scratch: ") + hipGetErrorString(e); return false; }
        return true;
    };
    const bool local_ok = local_setup();
    if (!local_ok && world == 1) return fail(local_why);
    int port = 0;
    if (world > 1) {
        if (rank == 0) {
            const uint64_t nonce = fresh_nonce();
            int rc = tcp_listen(c, &port);                      // both transports: the join handshake runs over loopback TCP
            if (rc) return rc;
            ::unlink(c->file.c_str());                          // whatever an earlier job left there
            if (!write_rendezvous(c->file, port, transport == SS_COMM_RCCL ? &id : nullptr, nonce)) return fail("cannot write the rendezvous file " + c->file);
            rc = tcp_accept_all(c, nonce);
            if (rc) return rc;
        } else {
            std::string hex;
            uint64_t nonce = 0;
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                if (read_rendezvous(c->file, &port, &hex, &nonce)) {
                    const int j = tcp_try_join(c, port, nonce);
                    if (j == 0) break;
                    if (j != 1) return j;
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(join_timeout_s()))
                    return fail("timed out waiting for a live rank 0 at the rendezvous file " + c->file);
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
            if (transport == SS_COMM_RCCL && !unhex(hex, reinterpret_cast<unsigned char *>(id.internal), sizeof id.internal))
                return fail("rendezvous file holds no RCCL id (mixed transports?)");
        }
    }
    if (world > 1) {                                        // the ranks have met: does every one of them stand?
        uint64_t bad = local_ok ? 0 : 1;
        const int rc = tcp_allreduce(c, &bad, 1, 0);
        if (rc != SS_OK || bad) return fail(!local_ok ? local_why : rc != SS_OK ? std::string("the join handshake broke") : std::string("another rank could not set up its device or librccl"));
    }
    if (transport == SS_COMM_RCCL) {
        // ncclCommInitRank blocks until every rank has called it; a rank that failed earlier would leave the others there
        // for good, so it runs under a watchdog (the helper thread is abandoned on a timeout: the process is about to fail)
        struct InitJob { Rccl *r; ncclComm_t comm = nullptr; int world, rank, device; ncclUniqueId id; };
        auto job = std::make_shared<InitJob>();
        job->r = &c->rccl; job->world = world; job->rank = rank; job->device = c->device; job->id = id;
        auto done = std::make_shared<std::promise<ncclResult_t>>();
        std::future<ncclResult_t> fut = done->get_future();
        std::thread([job, done]() {
            (void)hipSetDevice(job->device);
            done->set_value(job->r->CommInitRank(&job->comm, job->world, job->id, job->rank));
        }).detach();
        int verdict = SS_OK;
        std::string why;
        if (fut.wait_for(std::chrono::seconds(join_timeout_s())) != std::future_status::ready) { verdict = SS_ERR_DEVICE; why = "ncclCommInitRank: timed out"; }
        else {
            const ncclResult_t r = fut.get();
            if (r != ncclSuccess) {
                verdict = SS_ERR_DEVICE;
                why = std::string("ncclCommInitRank: ") + c->rccl.GetErrorString(r);
                const char *ipc = std::getenv("HSA_ENABLE_IPC_MODE_LEGACY");
                why += std::string(" (HSA_ENABLE_IPC_MODE_LEGACY=") + (ipc ? ipc : "unset") +
                       " in this process; dmabuf-only hosts need 0, in place before the process' first HIP call: export it in the launcher, "
                       "or create the communicator with ss_comm_init_on_device / ss_comm_init_from_env BEFORE any other GPU call)";
            }
            else c->comm = job->comm;
        }
        // all ranks succeed or all fail: the verdicts are summed over the join sockets
        if (world > 1) {
            uint64_t bad = verdict != SS_OK;
            if (tcp_allreduce(c, &bad, 1, 0) != SS_OK || bad) { if (why.empty()) why = "ncclCommInitRank failed on another rank"; verdict = SS_ERR_DEVICE; }
            for (int fd : c->peers) if (fd >= 0) ::close(fd);
            c->peers.clear();
            if (c->listen_fd >= 0) { ::close(c->listen_fd); c->listen_fd = -1; }
        }
        if (verdict != SS_OK) return fail(why);
    }
    if (rank == 0 && world > 1) { ::unlink(c->file.c_str()); }
    g.c = nullptr;
    *out = c;
    return SS_OK;
}

// Launchers that export RANK / WORLD_SIZE (torchrun, used purely as a process launcher): the rendezvous file is
// SS_COMM_FILE if set, otherwise /tmp/ss_comm_<launcher pid>_<launcher start time>_<MASTER_PORT> — all ranks are
// children of one launcher process, and the start time keeps a recycled pid from matching a stale file.
int ss_comm_init(int transport, int rank, int world, const char *rendezvous_file, ss_comm **out)
{
    return comm_init_impl(transport, rank, world, -1, rendezvous_file, out);
}

int ss_comm_init_on_device(int transport, int rank, int world, int device, const char *rendezvous_file, ss_comm **out)
{
    if (device < 0) return SS_ERR_INVALID_ARG;
    return comm_init_impl(transport, rank, world, device, rendezvous_file, out);
}

int ss_comm_init_from_env(int transport, ss_comm **out)
{
    const char *r = std::getenv("RANK"), *w = std::getenv("WORLD_SIZE");
    const int rank = r ? std::atoi(r) : 0, world = w ? std::atoi(w) : 1;
    std::string file;
    if (const char *f = std::getenv("SS_COMM_FILE")) {
        file = f;
    } else {
        const long ppid = (long)::getppid();
        unsigned long long start = 0;
        char path[64];
        std::snprintf(path, sizeof path, "/proc/%ld/stat", ppid);
        if (FILE *f = std::fopen(path, "r")) {
            char buf[1024];
            if (std::fgets(buf, sizeof buf, f)) {
                if (const char *p = std::strrchr(buf, ')')) {           // field 22 (starttime), counted after the command name
                    int field = 2;
                    for (p++; *p && field < 22; p++) if (*p == ' ') field++;
                    start = std::strtoull(p, nullptr, 10);
                }
            }
            std::fclose(f);
        }
        const char *mp = std::getenv("MASTER_PORT");
        file = "/tmp/ss_comm_" + std::to_string(ppid) + "_" + std::to_string(start) + "_" + (mp ? mp : "0") + ".rdzv";
    }
    // the rank's GPU: SS_COMM_DEVICE (explicit: must name a visible device, anything else is an error), else LOCAL_RANK (one process
    // per GPU), else whatever device is current.  Launchers that MASK one GPU per task (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES
    // per rank, usual under SLURM) leave every rank with a single visible device 0 while LOCAL_RANK still counts 0 .. n-1: a
    // LOCAL_RANK beyond the visible devices maps onto them (device 0 when there is one, LOCAL_RANK modulo the count otherwise).
    int device = -1;
    bool wrap = false;
    std::string bad;                       // a value that names no device: becomes this rank's verdict, so that its peers hear of it too
    auto parse_int = [](const char *t, long *v) {
        char *end = nullptr;
        errno = 0;
        *v = std::strtol(t, &end, 10);
        return end != t && *end == '\0' && errno == 0 && *v >= 0 && *v <= 1 << 20;
    };
    if (transport == SS_COMM_RCCL) {
        long v = 0;
        if (const char *d = std::getenv("SS_COMM_DEVICE")) {
            if (parse_int(d, &v)) device = (int)v;
            else bad = std::string("SS_COMM_DEVICE is not a device ordinal: '") + d + "'";
        } else if (const char *l = std::getenv("LOCAL_RANK")) {
            if (parse_int(l, &v)) { device = (int)v; wrap = true; }
            else bad = std::string("LOCAL_RANK is not a number: '") + l + "'";
        }
    }
    return comm_init_impl(transport, rank, world, transport == SS_COMM_RCCL ? device : -1, file.c_str(), out, wrap, bad.empty() ? nullptr : bad.c_str());
}

int ss_comm_rank(const ss_comm *c) { return c ? c->rank : -1; }

int ss_comm_size(const ss_comm *c)
{
    if (!c) return 0;
    if (c->transport == SS_COMM_RCCL && c->comm) {
        int n = 0;
        if (c->rccl.CommCount(c->comm, &n) == ncclSuccess) return n;
        return 0;
    }
    if (c->world == 1) return 1;
    if (c->rank == 0) { int n = 1; for (int fd : c->peers) if (fd >= 0) n++; return n; }
    return c->world;
}

int ss_comm_library_version(const ss_comm *c)
{
    if (!c || c->transport != SS_COMM_RCCL || !c->rccl.GetVersion) return 0;
    int v = 0;
    return c->rccl.GetVersion(&v) == ncclSuccess ? v : 0;
}

const char *ss_comm_transport_name(const ss_comm *c)
{
    if (!c) return "none";
    return c->transport == SS_COMM_RCCL ? "rccl" : "host-tcp";
}

int ss_comm_allreduce_u64_sum(ss_comm *c, uint64_t *inout, size_t n)
{
    if (!c || (!inout && n)) return SS_ERR_INVALID_ARG;
    if (!n) return SS_OK;
    if (c->transport == SS_COMM_RCCL) { DeviceScope ds(c->device); return rccl_allreduce_host(c, inout, n, ncclUint64, ncclSum); }
    return tcp_allreduce(c, inout, n, 0);
}

int ss_comm_allreduce_f64_max(ss_comm *c, double *inout, size_t n)
{
    if (!c || (!inout && n)) return SS_ERR_INVALID_ARG;
    if (!n) return SS_OK;
    if (c->transport == SS_COMM_RCCL) { DeviceScope ds(c->device); return rccl_allreduce_host(c, inout, n, ncclFloat64, ncclMax); }
    return tcp_allreduce(c, inout, n, 1);
}

int ss_comm_barrier(ss_comm *c)
{
    uint64_t one = 1;
    int rc = ss_comm_allreduce_u64_sum(c, &one, 1);
    if (rc) return rc;
    return one == (uint64_t)c->world ? SS_OK : fail("barrier: rank count mismatch");
}

int ss_batch_allreduce_histograms(ss_batch *b, ss_comm *c, uint64_t *out2000)
{
    if (!b || !c) return SS_ERR_INVALID_ARG;
    void *hist = ssi::batch_corpus_device(b);
    if (!hist) return SS_ERR_INVALID_MODE;                     // the batch runs no meter pass
    DeviceScope ds(ssi::batch_device(b));
    hipStream_t s = ssi::batch_stream(b);
    constexpr size_t kWords = 2000;
    // The reduction is in place, so it must happen once per pass: a second call for the same pass (or
    // ss_batch_corpus_gate_enqueue after ss_batch_allreduce_histograms) returns the sums already there.
    bool &reduced = ssi::batch_corpus_reduced(b);
    if (reduced) {
        if (out2000) {
            COMM_HIP(hipMemcpyAsync(out2000, hist, kWords * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
            COMM_HIP(hipStreamSynchronize(s));
        }
        return SS_OK;
    }
    if (c->transport == SS_COMM_RCCL) {
        if (c->device != ssi::batch_device(b)) return SS_ERR_INVALID_ARG;
        // in place on the batch's stream, behind the kernels of ss_batch_run
        COMM_NCCL(c, c->rccl.AllReduce(hist, hist, kWords, ncclUint64, ncclSum, c->comm, s));
        reduced = true;
        if (out2000) {
            COMM_HIP(hipMemcpyAsync(out2000, hist, kWords * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
            COMM_HIP(hipStreamSynchronize(s));
        }
        return SS_OK;
    }
    std::vector<uint64_t> h(kWords);
    COMM_HIP(hipMemcpyAsync(h.data(), hist, kWords * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    COMM_HIP(hipStreamSynchronize(s));
    int rc = tcp_allreduce(c, h.data(), kWords, 0);
    if (rc) return rc;
    COMM_HIP(hipMemcpyAsync(hist, h.data(), kWords * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    COMM_HIP(hipStreamSynchronize(s));
    reduced = true;
    if (out2000) std::memcpy(out2000, h.data(), kWords * sizeof(uint64_t));
    return SS_OK;
}

}  // extern "C"
