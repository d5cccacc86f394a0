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
  Hub(int listen_fd, int nranks) : listen_fd_(listen_fd), nranks_(nranks) {
    thread_ = std::thread([this] { run(); });
  }
  ~Hub() {
    stop_ = true;
    if (thread_.joinable()) thread_.join();
    for (int fd : fds_)
      if (fd >= 0) ::close(fd);
    ::close(listen_fd_);
  }

 private:
  struct Pending {
    int count = 0;
    uint64_t bytes = 0;
    std::vector<char> data;
    std::vector<int> fds;
  };

  void run() {
    fds_.assign(nranks_, -1);
    int connected = 0, open = 0;
    try {
      while (!stop_) {
        std::vector<pollfd> pfs;
        if (connected < nranks_) pfs.push_back({listen_fd_, POLLIN, 0});
        for (int fd : fds_)
          if (fd >= 0) pfs.push_back({fd, POLLIN, 0});
        if (pfs.empty()) break;  // everybody came and left
        int pr = ::poll(pfs.data(), pfs.size(), 200);
        if (pr <= 0) continue;
        for (auto& pf : pfs) {
          if (!(pf.revents & (POLLIN | POLLHUP | POLLERR))) continue;
          if (pf.fd == listen_fd_) {
            int fd = ::accept(listen_fd_, nullptr, nullptr);
            if (fd < 0) continue;
            int one = 1;
            ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            int32_t r = -1;
            recvAll(fd, &r, sizeof(r), 30);
            if (r < 0 || r >= nranks_ || fds_[r] >= 0) {
              ::close(fd);
              continue;
            }
            fds_[r] = fd;
            ++connected;
            ++open;
          } else {
            OpHeader h;
            ssize_t k = ::recv(pf.fd, &h, sizeof(h), MSG_PEEK | MSG_DONTWAIT);
            if (k == 0 || (k < 0 && errno != EAGAIN && errno != EWOULDBLOCK)) {
              for (int& fd : fds_)
                if (fd == pf.fd) {
                  ::close(fd);
                  fd = -1;
                  --open;
                }
              continue;
            }
            if (k < (ssize_t)sizeof(h)) continue;
            recvAll(pf.fd, &h, sizeof(h), 30);
            Pending& p = pending_[h.comm_id];
            if (p.count == 0) {
              p.bytes = h.bytes;
              p.data.assign((size_t)h.bytes * h.nmembers, 0);
              p.fds.assign(h.nmembers, -1);
            }
            if (h.bytes) recvAll(pf.fd, p.data.data() + (size_t)h.index * h.bytes, h.bytes, 30);
            p.fds[h.index] = pf.fd;
            if (++p.count == h.nmembers) {
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
  }

  int listen_fd_, nranks_;
  std::vector<int> fds_;
  std::map<uint64_t, Pending> pending_;
  std::atomic<bool> stop_{false};
  std::thread thread_;
};

struct TcpShared {
  int fd = -1;
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
      sockaddr_in bind_addr = sa;
      bind_addr.sin_addr.s_addr = htonl(INADDR_ANY);
      if (::bind(lfd, reinterpret_cast<sockaddr*>(&bind_addr), sizeof(bind_addr)) != 0 || ::listen(lfd, 128) != 0) {
        const std::string why = std::strerror(errno);
        ::close(lfd);
        CD_BOOTSTRAP_ERROR("cannot listen on bootstrap port " + std::to_string(env.port) + ": " + why +
                           " (set CUDECOMP_BOOTSTRAP_PORT)");
      }
      sh->hub = std::make_unique<Hub>(lfd, env.size);
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
    int32_t r = env.rank;
    sendAll(sh->fd, &r, sizeof(r));
    g_shared = sh;
  }
  auto b = std::make_unique<TcpBootstrap>(sh, mix(0xC0DEC0DEULL, (uint64_t)instance), env.rank, env.size);
  b->barrier();  // everyone is connected and agrees on the instance number
  return b;
}

}  // namespace cudecomp
