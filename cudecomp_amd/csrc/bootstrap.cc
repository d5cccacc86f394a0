// bootstrap.cc -- see bootstrap.h
#include "bootstrap.h"

#include <arpa/inet.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>

#include "errors.h"

namespace cudecomp {

// ---- collectives derived from allgather -------------------------------------------------------
void Bootstrap::barrier() {
  char c = 0;
  std::vector<char> all(size());
  allgather(&c, all.data(), 1);
}

void Bootstrap::bcast(void* buf, size_t bytes, int root) {
  if (size() == 1 || bytes == 0) return;
  std::vector<char> all(bytes * size());
  allgather(buf, all.data(), bytes);
  std::memcpy(buf, all.data() + bytes * root, bytes);
}

namespace {
template <typename T, typename F>
T reduceAll(Bootstrap& b, T v, F f) {
  std::vector<T> all(b.size());
  b.allgather(&v, all.data(), sizeof(T));
  T r = all[0];
  for (int i = 1; i < b.size(); ++i) r = f(r, all[i]);
  return r;
}
}  // namespace

double Bootstrap::allreduceMin(double v) { return reduceAll(*this, v, [](double a, double b) { return std::min(a, b); }); }
double Bootstrap::allreduceMax(double v) { return reduceAll(*this, v, [](double a, double b) { return std::max(a, b); }); }
double Bootstrap::allreduceSum(double v) { return reduceAll(*this, v, [](double a, double b) { return a + b; }); }
int64_t Bootstrap::allreduceMaxI64(int64_t v) {
  return reduceAll(*this, v, [](int64_t a, int64_t b) { return std::max(a, b); });
}
bool Bootstrap::allreduceOr(bool v) {
  return reduceAll(*this, (int)v, [](int a, int b) { return a | b; }) != 0;
}

// ---- environment --------------------------------------------------------------------------------
namespace {
bool envInt(const char* name, int* out) {
  const char* v = std::getenv(name);
  if (!v || !*v) return false;
  char* end = nullptr;
  long x = std::strtol(v, &end, 10);
  if (end == v) return false;
  *out = (int)x;
  return true;
}
}  // namespace

LaunchEnv detectLaunchEnv() {
  LaunchEnv e;
  static const char* kPairs[][2] = {{"RANK", "WORLD_SIZE"},
                                    {"PMI_RANK", "PMI_SIZE"},
                                    {"OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE"},
                                    {"SLURM_PROCID", "SLURM_NTASKS"}};
  for (auto& p : kPairs) {
    int r, s;
    if (envInt(p[0], &r) && envInt(p[1], &s) && s >= 1 && r >= 0 && r < s) {
      e.rank = r;
      e.size = s;
      break;
    }
  }
  if (const char* a = std::getenv("CUDECOMP_BOOTSTRAP_ADDR")) e.addr = a;
  else if (const char* m = std::getenv("MASTER_ADDR")) e.addr = m;
  int port;
  if (envInt("CUDECOMP_BOOTSTRAP_PORT", &port)) e.port = port;
  else if (envInt("MASTER_PORT", &port)) e.port = port + 17;  // MASTER_PORT itself belongs to the launcher's store
  return e;
}

// ---- local --------------------------------------------------------------------------------------
namespace {

class LocalBootstrap : public Bootstrap {
 public:
  int rank() const override { return 0; }
  int size() const override { return 1; }
  void allgather(const void* send, void* recv, size_t bytes) override {
    if (send != recv) std::memcpy(recv, send, bytes);
  }
  std::unique_ptr<Bootstrap> split(int, int) override { return std::make_unique<LocalBootstrap>(); }
};

// ---- tcp ----------------------------------------------------------------------------------------
struct OpHeader {
  uint64_t comm_id;
  int32_t nmembers;
  int32_t index;
  uint64_t bytes;
};
// first message of every connection: who is calling, and proof that it belongs to this job
struct Hello {
  uint32_t magic;
  int32_t rank;
  uint64_t secret;
};
constexpr uint32_t kHelloMagic = 0x43444250;            // "CDBP"
constexpr uint64_t kMaxContribution = 64ull << 20;       // no control-plane message comes near this

uint64_t fnv1a(uint64_t h, const char* s) {
  for (; s && *s; ++s) h = (h ^ (unsigned char)*s) * 0x100000001b3ULL;
  return h;
}
// Shared by the ranks of one job and nobody else who merely guesses the port: CUDECOMP_BOOTSTRAP_SECRET if the launcher
// exports one, else derived from what identifies the job in the launcher environment.  It keeps stray or stale
// processes (another job on the same port, a rank of a previous run) out of the hub; it is not cryptography.
uint64_t jobSecret(const LaunchEnv& env) {
  uint64_t h = 0xcbf29ce484222325ULL;
  if (const char* s = std::getenv("CUDECOMP_BOOTSTRAP_SECRET")) return fnv1a(h, s);
  static const char* kJobIds[] = {"TORCHELASTIC_RUN_ID", "SLURM_JOB_ID", "SLURM_STEP_ID", "PMI_JOBID", "PMIX_NAMESPACE",
                                  "OMPI_MCA_orte_hnp_uri", "CUDECOMP_TEST_JOB"};
  for (const char* k : kJobIds) h = fnv1a(h, std::getenv(k));
  h = fnv1a(h, env.addr.c_str());
  h = fnv1a(h, std::to_string(env.port).c_str());
  h = fnv1a(h, std::to_string(env.size).c_str());
  return h;
}

int timeoutSeconds() {
  int t = 120;
  envInt("CUDECOMP_BOOTSTRAP_TIMEOUT", &t);
  return t;
}

void sendAll(int fd, const void* buf, size_t n) {
  const char* p = static_cast<const char*>(buf);
  while (n > 0) {
    ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
    if (k < 0) {
      if (errno == EINTR) continue;
      CD_BOOTSTRAP_ERROR(std::string("bootstrap send failed: ") + std::strerror(errno));
    }
    p += k;
    n -= (size_t)k;
  }
}

void recvAll(int fd, void* buf, size_t n, int timeout_s) {
  char* p = static_cast<char*>(buf);
  while (n > 0) {
    pollfd pf{fd, POLLIN, 0};
    int pr = ::poll(&pf, 1, timeout_s * 1000);
    if (pr == 0) CD_BOOTSTRAP_ERROR("bootstrap timed out waiting for the other ranks");
    if (pr < 0) {
      if (errno == EINTR) continue;
      CD_BOOTSTRAP_ERROR(std::string("bootstrap poll failed: ") + std::strerror(errno));
    }
    ssize_t k = ::recv(fd, p, n, 0);
    if (k == 0) CD_BOOTSTRAP_ERROR("bootstrap peer closed the connection");
    if (k < 0) {
      if (errno == EINTR) continue;
      CD_BOOTSTRAP_ERROR(std::string("bootstrap recv failed: ") + std::strerror(errno));
    }
    p += k;
    n -= (size_t)k;
  }
}

// The hub: runs in rank 0's process, gathers one contribution per member of a communicator and
// answers every member with the concatenation.
class Hub {
 public:
  Hub(int listen_fd, int nranks, uint64_t secret) : listen_fd_(listen_fd), nranks_(nranks), secret_(secret) {
    thread_ = std::thread([this] { run(); });
  }
  ~Hub() {
    // Rank 0 may be done (and finalizing) while other ranks are still inside a collective of a sub-communicator it does
    // not belong to: keep serving until every rank has hung up (the loop ends by itself then), within reason.
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(std::min(timeoutSeconds(), 20));
    while (!done_ && std::chrono::steady_clock::now() < deadline) std::this_thread::sleep_for(std::chrono::milliseconds(2));
    stop_ = true;
    if (thread_.joinable()) thread_.join();
    for (int fd : fds_)
      if (fd >= 0) ::close(fd);
    for (auto& g : greeting_) ::close(g.first);
    ::close(listen_fd_);
  }

 private:
  struct Pending {
    int count = 0;
    int nmembers = 0;
    uint64_t bytes = 0;
    std::vector<char> data;
    std::vector<int> fds;
  };

  void drop(int fd, int* open) {
    for (int& f : fds_)
      if (f == fd) {
        ::close(f);
        f = -1;
        --*open;
      }
  }

  void run() {
    fds_.assign(nranks_, -1);
    int connected = 0, open = 0;
    try {
      while (!stop_) {
        std::vector<pollfd> pfs;
        if (connected < nranks_) pfs.push_back({listen_fd_, POLLIN, 0});
        for (auto& g : greeting_) pfs.push_back({g.first, POLLIN, 0});
        for (int fd : fds_)
          if (fd >= 0) pfs.push_back({fd, POLLIN, 0});
        if (pfs.empty()) break;  // everybody came and left
        int pr = ::poll(pfs.data(), pfs.size(), 200);
        const auto now = std::chrono::steady_clock::now();
        for (auto it = greeting_.begin(); it != greeting_.end();) {  // strangers that never introduce themselves
          if (now > it->second) {
            ::close(it->first);
            it = greeting_.erase(it);
          } else {
            ++it;
          }
        }
        if (pr <= 0) continue;
        for (auto& pf : pfs) {
          if (!(pf.revents & (POLLIN | POLLHUP | POLLERR))) continue;
          if (pf.fd == listen_fd_) {
            int fd = ::accept(listen_fd_, nullptr, nullptr);
            if (fd < 0) continue;
            int one = 1;
            ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            // the greeting is read when it has arrived (never block the hub on one client)
            greeting_[fd] = now + std::chrono::seconds(30);
          } else if (greeting_.count(pf.fd)) {
            Hello hello;
            ssize_t k = ::recv(pf.fd, &hello, sizeof(hello), MSG_PEEK | MSG_DONTWAIT);
            if (k == 0 || (k < 0 && errno != EAGAIN && errno != EWOULDBLOCK)) {
              ::close(pf.fd);
              greeting_.erase(pf.fd);
              continue;
            }
            if (k < (ssize_t)sizeof(hello)) continue;
            recvAll(pf.fd, &hello, sizeof(hello), 5);
            greeting_.erase(pf.fd);
            if (hello.magic != kHelloMagic || hello.secret != secret_ || hello.rank < 0 || hello.rank >= nranks_ ||
                fds_[hello.rank] >= 0) {
              ::close(pf.fd);  // not one of ours (or a rank that is already connected)
              continue;
            }
            fds_[hello.rank] = pf.fd;
            ++connected;
            ++open;
          } else {
            OpHeader h;
            ssize_t k = ::recv(pf.fd, &h, sizeof(h), MSG_PEEK | MSG_DONTWAIT);
            if (k == 0 || (k < 0 && errno != EAGAIN && errno != EWOULDBLOCK)) {
              drop(pf.fd, &open);
              continue;
            }
            if (k < (ssize_t)sizeof(h)) continue;
            recvAll(pf.fd, &h, sizeof(h), 30);
            // nothing from the wire is used as an index or a size before it has been checked
            const bool sane = h.nmembers >= 1 && h.nmembers <= nranks_ && h.index >= 0 && h.index < h.nmembers &&
                              h.bytes <= kMaxContribution;
            auto pit = pending_.find(h.comm_id);
            const bool consistent = sane && (pit == pending_.end() || (pit->second.bytes == h.bytes &&
                                                                      pit->second.nmembers == h.nmembers &&
                                                                      pit->second.fds[h.index] < 0));
            if (!consistent) {
              drop(pf.fd, &open);  // a confused or hostile client: its collective will time out on the others
              continue;
            }
            Pending& p = pending_[h.comm_id];
            if (p.count == 0) {
              p.bytes = h.bytes;
              p.nmembers = h.nmembers;
              p.data.assign((size_t)h.bytes * h.nmembers, 0);
              p.fds.assign(h.nmembers, -1);
            }
            if (h.bytes) recvAll(pf.fd, p.data.data() + (size_t)h.index * h.bytes, h.bytes, 30);
            p.fds[h.index] = pf.fd;
            if (++p.count == p.nmembers) {
              for (int fd : p.fds) sendAll(fd, p.data.data(), p.data.size());
              pending_.erase(h.comm_id);
            }
          }
        }
        if (connected == nranks_ && open == 0) break;
      }
    } catch (const std::exception&) {
      // a broken client connection surfaces on the clients as a timeout / closed socket
    }
    done_ = true;
  }

  int listen_fd_, nranks_;
  uint64_t secret_;
  std::vector<int> fds_;
  std::map<int, std::chrono::steady_clock::time_point> greeting_;  // accepted, not yet introduced: fd -> deadline
  std::map<uint64_t, Pending> pending_;
  std::atomic<bool> stop_{false}, done_{false};
  std::thread thread_;
};

struct TcpShared {
  int fd = -1;
  int instances = 0;         // bootstraps created on THIS hub connection so far
  std::unique_ptr<Hub> hub;  // rank 0 only
  ~TcpShared() {
    if (fd >= 0) ::close(fd);
    hub.reset();
  }
};

uint64_t mix(uint64_t h, uint64_t v) {
  h ^= v + 0x9e3779b97f4a7c15ULL + (h << 6) + (h >> 2);
  return h * 0xff51afd7ed558ccdULL;
}

class TcpBootstrap : public Bootstrap {
 public:
  TcpBootstrap(std::shared_ptr<TcpShared> sh, uint64_t id, int rank, int size)
      : sh_(std::move(sh)), id_(id), rank_(rank), size_(size) {}
  int rank() const override { return rank_; }
  int size() const override { return size_; }

  void allgather(const void* send, void* recv, size_t bytes) override {
    if (size_ == 1) {
      if (send != recv) std::memcpy(recv, send, bytes);
      return;
    }
    OpHeader h{mix(id_, seq_++), size_, rank_, bytes};
    std::vector<char> msg(sizeof(h) + bytes);
    std::memcpy(msg.data(), &h, sizeof(h));
    if (bytes) std::memcpy(msg.data() + sizeof(h), send, bytes);
    sendAll(sh_->fd, msg.data(), msg.size());
    recvAll(sh_->fd, recv, bytes * size_, timeoutSeconds());
  }

  std::unique_ptr<Bootstrap> split(int color, int key) override {
    struct Entry {
      int color, key, rank;
    };
    Entry mine{color, key, rank_};
    std::vector<Entry> all(size_);
    allgather(&mine, all.data(), sizeof(Entry));
    std::vector<Entry> members;
    for (auto& e : all)
      if (e.color == color) members.push_back(e);
    std::stable_sort(members.begin(), members.end(),
                     [](const Entry& a, const Entry& b) { return a.key != b.key ? a.key < b.key : a.rank < b.rank; });
    int new_rank = 0;
    for (size_t i = 0; i < members.size(); ++i)
      if (members[i].rank == rank_) new_rank = (int)i;
    uint64_t nid = mix(mix(id_, 0x5117 + splits_++), (uint64_t)(uint32_t)color);
    return std::make_unique<TcpBootstrap>(sh_, nid, new_rank, (int)members.size());
  }

 private:
  std::shared_ptr<TcpShared> sh_;
  uint64_t id_;
  int rank_, size_;
  uint64_t seq_ = 0, splits_ = 0;
};

}  // namespace

std::unique_ptr<Bootstrap> makeLocalBootstrap() { return std::make_unique<LocalBootstrap>(); }

std::unique_ptr<Bootstrap> makeTcpBootstrap(const LaunchEnv& env, int instance) {
  // One hub connection per process, shared by every handle (and every split of it); handles are told
  // apart by their communicator ids.  Not thread-safe, like the rest of the library.
  static std::weak_ptr<TcpShared> g_shared;
  auto sh = g_shared.lock();
  if (!sh) {
    sh = std::make_shared<TcpShared>();

    addrinfo hints{}, *res = nullptr;
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    if (::getaddrinfo(env.addr.c_str(), std::to_string(env.port).c_str(), &hints, &res) != 0 || !res)
      CD_BOOTSTRAP_ERROR("cannot resolve bootstrap address " + env.addr);
    sockaddr_in sa = *reinterpret_cast<sockaddr_in*>(res->ai_addr);
    ::freeaddrinfo(res);

    if (env.rank == 0) {
      int lfd = ::socket(AF_INET, SOCK_STREAM, 0);
      int one = 1;
      ::setsockopt(lfd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
      // listen on the interface the ranks were told to call (MASTER_ADDR), not on every interface of the host; only if
      // that address is not configured on this machine (NAT, a virtual IP) fall back to all interfaces
      sockaddr_in bind_addr = sa;
      // A HOST NAME that resolves to a loopback alias on rank 0's own machine (Debian's 127.0.1.1 entry) says nothing
      // about where the other nodes will call: listen everywhere then (the job-secret greeting still guards the hub).
      // A literal loopback address means a single-node job and stays on loopback.
      const bool loopback = (ntohl(sa.sin_addr.s_addr) >> 24) == 127;
      in_addr literal{};
      if (loopback && ::inet_pton(AF_INET, env.addr.c_str(), &literal) != 1 && env.addr != "localhost")
        bind_addr.sin_addr.s_addr = htonl(INADDR_ANY);
      int brc = ::bind(lfd, reinterpret_cast<sockaddr*>(&bind_addr), sizeof(bind_addr));
      if (brc != 0 && errno == EADDRNOTAVAIL) {
        bind_addr.sin_addr.s_addr = htonl(INADDR_ANY);
        brc = ::bind(lfd, reinterpret_cast<sockaddr*>(&bind_addr), sizeof(bind_addr));
      }
      if (brc != 0 || ::listen(lfd, 128) != 0) {
        const std::string why = std::strerror(errno);
        ::close(lfd);
        CD_BOOTSTRAP_ERROR("cannot listen on bootstrap port " + std::to_string(env.port) + ": " + why +
                           " (set CUDECOMP_BOOTSTRAP_PORT)");
      }
      sh->hub = std::make_unique<Hub>(lfd, env.size, jobSecret(env));
    }

    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(timeoutSeconds());
    for (;;) {
      int fd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (::connect(fd, reinterpret_cast<sockaddr*>(&sa), sizeof(sa)) == 0) {
        int one = 1;
        ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
        sh->fd = fd;
        break;
      }
      ::close(fd);
      if (std::chrono::steady_clock::now() > deadline)
        CD_BOOTSTRAP_ERROR("cannot connect to bootstrap hub at " + env.addr + ":" + std::to_string(env.port));
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    const Hello hello{kHelloMagic, (int32_t)env.rank, jobSecret(env)};
    sendAll(sh->fd, &hello, sizeof(hello));
    g_shared = sh;
  }
  // The communicator id counts the bootstraps of this hub CONNECTION, not of the process: when the last handle is finalized
  // the connection goes away, and a process that later joins another job (a long-lived worker that serves several launches,
  // tests/mp.py) starts from zero like the fresh processes it meets there.
  (void)instance;
  auto b = std::make_unique<TcpBootstrap>(sh, mix(0xC0DEC0DEULL, (uint64_t)sh->instances++), env.rank, env.size);
  b->barrier();  // everyone is connected and agrees on the instance number
  return b;
}

}  // namespace cudecomp
