// transport.cc -- see transport.h
#include "transport.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <functional>
#include <thread>

#include "errors.h"

namespace cudecomp {

// ================================================================================================
// world bootstrap
// ================================================================================================
#ifndef CUDECOMP_WITH_MPI
std::unique_ptr<Bootstrap> makeWorldBootstrap(MPI_Comm comm, int instance) {
  // Built without MPI: the communicator is only a token for "all ranks the launcher started".
  if (comm == MPI_COMM_NULL) CD_INVALID_USAGE("null communicator");
  const LaunchEnv env = detectLaunchEnv();
  if (env.size == 1) return makeLocalBootstrap();
  return makeTcpBootstrap(env, instance);
}
MPI_Comm commFromFortran(MPI_Fint f) { return (MPI_Comm)f; }
#endif

// ================================================================================================
// RCCL
// ================================================================================================
class RcclContext {
 public:
  explicit RcclContext(cudecompHandle_t h) {
    ncclUniqueId id;
    std::memset(&id, 0, sizeof(id));
    if (h->rank == 0) CD_CHECK_RCCL(ncclGetUniqueId(&id));
    h->boot->bcast(&id, sizeof(id), 0);
    CD_CHECK_RCCL(ncclCommInitRank(&comm_, h->nranks, id, h->rank));
  }
  ~RcclContext() {
    if (comm_) ncclCommDestroy(comm_);
  }
  ncclComm_t comm() const { return comm_; }

 private:
  ncclComm_t comm_ = nullptr;
};

// ================================================================================================
// PEER: IPC-mapped buffers + one-sided xGMI copies
// ================================================================================================
// Platform quirk (ROCm 7.x, dmabuf IPC): hipIpcOpenMemHandle never returns for an allocation whose byte size has
// bit 31 set (2-4 GiB, 6-8 GiB, ...); every other size maps and transfers correctly (verified block by block with
// cudecompExtPeerProbe up to 17 GiB).  Library allocations are rounded up past such sizes; foreign buffers of
// such a size are refused instead of hanging.
inline bool ipcSizeHangs(size_t bytes) { return (bytes & 0x80000000ull) != 0; }
inline size_t ipcSafeSize(size_t bytes) {
  return ipcSizeHangs(bytes) ? ((bytes >> 32) + 1) << 32 : bytes;
}

class PeerContext {
 public:
  struct Region {
    char* base = nullptr;
    size_t bytes = 0;
    std::vector<char*> peer_base;  // by global rank; [my rank] = base
    bool from_library = false;     // allocated by cudecompMalloc
  };

  // collective over the handle's communicator
  explicit PeerContext(cudecompHandle_t h) : h_(h) { openBoard(); }

  ~PeerContext() {
    if (board_) ::munmap(board_, board_bytes_);
    for (auto& kv : regions_) closePeers(kv.second);
    for (hipStream_t s : copy_streams_) (void)hipStreamDestroy(s);
    for (hipEvent_t e : copy_events_) (void)hipEventDestroy(e);
  }

  Region* find(const void* ptr) {
    const char* p = static_cast<const char*>(ptr);
    auto it = regions_.upper_bound(const_cast<char*>(p));
    if (it == regions_.begin()) return nullptr;
    --it;
    Region& r = it->second;
    return (p >= r.base && p < r.base + r.bytes) ? &r : nullptr;
  }

  // collective over the handle's communicator
  Region* registerRegion(void* base, size_t bytes, bool from_library) {
    struct Wire {
      hipIpcMemHandle_t handle;
      unsigned long long bytes;
      int pid;
    };
    enablePeerAccessOnce();
    // Failures are agreed on collectively (a rank that threw on its own would leave the others waiting in the
    // next collective): everybody tries, then everybody learns whether anybody failed.
    Wire mine{};
    std::string error;
    hipError_t e = hipIpcGetMemHandle(&mine.handle, base);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      error = std::string("hipIpcGetMemHandle failed: ") + hipGetErrorString(e);
    }
    mine.bytes = error.empty() ? bytes : 0;  // 0 = "I have nothing to offer"
    mine.pid = (int)::getpid();
    std::vector<Wire> all(h_->nranks);
    h_->boot->allgather(&mine, all.data(), sizeof(Wire));
    Region r;
    r.base = static_cast<char*>(base);
    r.bytes = bytes;
    r.from_library = from_library;
    r.peer_base.assign(h_->nranks, nullptr);
    for (int p = 0; p < h_->nranks && error.empty(); ++p) {
      if (p == h_->rank) {
        r.peer_base[p] = r.base;
        continue;
      }
      if (h_->hostnames[p] != h_->hostnames[h_->rank]) continue;  // no xGMI path: not mappable
      if (all[p].bytes == 0) {
        error = "a peer rank could not export its buffer over IPC";
        break;
      }
      void* mapped = nullptr;
      e = hipIpcOpenMemHandle(&mapped, all[p].handle, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        error = std::string("hipIpcOpenMemHandle failed: ") + hipGetErrorString(e) +
                " (is HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)";
        break;
      }
      r.peer_base[p] = static_cast<char*>(mapped);
    }
    const bool anyone_failed = h_->boot->allreduceOr(!error.empty());  // also: everybody has finished mapping
    if (anyone_failed) {
      closePeers(r);
      CD_PEER_ERROR(error.empty() ? std::string("IPC mapping failed on another rank") : error);
    }
    auto ins = regions_.emplace(r.base, std::move(r));
    return &ins.first->second;
  }

  // Best effort: a process that sees several GPUs (torchrun exposes all of them to every rank) turns on peer access
  // from its device to the others, so kernels and copy engines may address IPC-mapped memory of any of them.
  void enablePeerAccessOnce() {
    if (peer_access_done_) return;
    peer_access_done_ = true;
    int dev = -1, count = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceCount(&count) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    for (int d = 0; d < count; ++d) {
      int can = 0;
      if (d == dev || hipDeviceCanAccessPeer(&can, dev, d) != hipSuccess || !can) continue;
      (void)hipDeviceEnablePeerAccess(d, 0);  // "already enabled" is fine
    }
    (void)hipGetLastError();
  }

  // collective
  void unregisterRegion(void* base) {
    auto it = regions_.find(static_cast<char*>(base));
    if (it == regions_.end()) return;
    h_->boot->barrier();  // nobody is still writing into a mapping that is about to disappear
    closePeers(it->second);
    regions_.erase(it);
    h_->boot->barrier();
  }

  // pointer through which `global_rank`'s copy of my buffer location `local` can be written; registers
  // the enclosing allocation on first sight (collective: every rank reaches this in the same call)
  char* translate(const void* local, int global_rank) {
    Region* r = find(local);
    if (!r) {
      void* base = nullptr;
      size_t bytes = 0;
      CD_CHECK_HIP(hipMemGetAddressRange(&base, &bytes, const_cast<void*>(local)));
      if (ipcSizeHangs(bytes))
        CD_PEER_ERROR("this buffer's allocation (" + std::to_string(bytes) + " bytes) cannot be shared over IPC on this "
                      "platform; obtain the workspace from cudecompMalloc");
      r = registerRegion(base, bytes, false);
    }
    char* pb = r->peer_base[global_rank];
    if (!pb) CD_PEER_ERROR("peer buffer is not reachable over xGMI/IPC (rank on another host?)");
    return pb + (static_cast<const char*>(local) - r->base);
  }

  bool anyUnregistered(const void* local) { return find(local) == nullptr; }

  // ---- host barrier among the members of a row / column communicator -------------------------------
  // Hot path of the host-ordered exchanges: a shared-memory epoch board (one cache line per rank and
  // communicator slot) when all members share this host, the bootstrap's all-gather otherwise.
  static constexpr int kSlots = 256;

  void barrier(cudecompCommInfo& ci) {
    if (!board_ || ci.ngroups != 1 || ci.barrier_slot < 0) {
      ci.boot->barrier();
      return;
    }
    const uint64_t epoch = ++ci.barrier_epoch;
    cell(ci.barrier_slot, h_->rank).store(epoch, std::memory_order_release);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
    for (int m = 0; m < ci.nranks; ++m) {
      auto& c = cell(ci.barrier_slot, ci.global_ranks[m]);
      int spins = 0;
      while (c.load(std::memory_order_acquire) < epoch) {
        if (++spins > 2000) {
          std::this_thread::yield();
          if ((spins & 0xfff) == 0 && std::chrono::steady_clock::now() > deadline)
            CD_PEER_ERROR("timed out in the shared-memory barrier (a peer rank died?)");
        }
      }
    }
  }

  // a communicator slot is (re)initialised by its members BEFORE the collective that creates the communicator
  void resetSlot(int slot) {
    if (!board_ || slot < 0) return;
    cell(slot, h_->rank).store(0, std::memory_order_release);
    if (flags_) {
      ready(slot, h_->rank).store(0, std::memory_order_release);
      for (int s = 0; s < h_->nranks; ++s) landed(slot, h_->rank, s).store(0, std::memory_order_release);
    }
  }
  bool hasBoard() const { return board_ != nullptr; }

  // ---- pairwise flags of the pipelined exchange (same shared segment, behind the barrier cells) -----------
  // ready(slot, r)      = last epoch for which rank r's receive area was free
  // landed(slot, d, s)  = last epoch whose chunk from rank s has completely arrived in rank d's receive area
  bool pipelineAvailable(const cudecompCommInfo& ci) const { return flags_ && ci.ngroups == 1 && ci.barrier_slot >= 0; }
  std::atomic<uint64_t>& ready(int slot, int rank) { return flags_[((size_t)slot * h_->nranks + rank) * (h_->nranks + 1)]; }
  std::atomic<uint64_t>& landed(int slot, int dst, int src) {
    return flags_[((size_t)slot * h_->nranks + dst) * (h_->nranks + 1) + 1 + src];
  }
  void waitFlag(std::atomic<uint64_t>& f, uint64_t epoch, const char* what) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(120);
    int spins = 0;
    while (f.load(std::memory_order_acquire) < epoch) {
      if (++spins > 2000) {
        std::this_thread::yield();
        if ((spins & 0xfff) == 0 && std::chrono::steady_clock::now() > deadline)
          CD_PEER_ERROR(std::string("timed out waiting for a peer in the pipelined exchange (") + what + ")");
      }
    }
  }
  hipEvent_t copyEvent(int i) {
    while ((int)copy_events_.size() <= i) {
      hipEvent_t e;
      CD_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      copy_events_.push_back(e);
    }
    return copy_events_[i];
  }

  hipStream_t copyStream(int i) {
    while ((int)copy_streams_.size() <= i) {
      hipStream_t s;
      CD_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      copy_streams_.push_back(s);
    }
    return copy_streams_[i];
  }

 private:
  void closePeers(Region& r) {
    for (int p = 0; p < (int)r.peer_base.size(); ++p)
      if (p != h_->rank && r.peer_base[p]) (void)hipIpcCloseMemHandle(r.peer_base[p]);
  }

  std::atomic<uint64_t>& cell(int slot, int rank) {
    return *reinterpret_cast<std::atomic<uint64_t>*>(board_ + ((size_t)slot * h_->nranks + rank) * 64);
  }

  void openBoard() {
    // every host gets its own segment; its name is agreed through the bootstrap, the creator unlinks it as
    // soon as all local ranks have mapped it, so nothing is left behind even if a rank crashes later
    const size_t cells_bytes = (size_t)kSlots * h_->nranks * 64;
    // pairwise flags: (1 + nranks) counters per slot and rank; left out for very large worlds (the pipelined
    // exchange then falls back to the barrier-ordered one)
    const size_t flags_bytes = h_->nranks <= 32 ? (size_t)kSlots * h_->nranks * (h_->nranks + 1) * sizeof(uint64_t) : 0;
    board_bytes_ = cells_bytes + flags_bytes;
    char name[128] = {0};
    if (h_->local_rank == 0)
      snprintf(name, sizeof(name), "/cudecomp_%d_%llx", (int)::getpid(),
               (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    std::vector<char> all((size_t)128 * h_->nranks);
    h_->boot->allgather(name, all.data(), 128);
    int creator = -1;
    for (int r = 0; r < h_->nranks; ++r)
      if (h_->hostnames[r] == h_->hostnames[h_->rank] && h_->rank_to_local_rank[r] == 0) creator = r;
    const char* shm_name = all.data() + (size_t)128 * creator;
    int fd = -1;
    if (h_->local_rank == 0) {
      fd = ::shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd >= 0 && ::ftruncate(fd, (off_t)board_bytes_) != 0) {
        ::close(fd);
        fd = -1;
      }
    }
    h_->boot->barrier();
    if (h_->local_rank != 0) fd = ::shm_open(shm_name, O_RDWR, 0600);
    void* p = (fd >= 0) ? ::mmap(nullptr, board_bytes_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0) : MAP_FAILED;
    if (fd >= 0) ::close(fd);
    const bool ok = (p != MAP_FAILED);
    const bool all_ok = !h_->boot->allreduceOr(!ok);  // also: everybody has mapped it
    if (h_->local_rank == 0) ::shm_unlink(shm_name);
    if (ok && all_ok) {
      board_ = static_cast<char*>(p);
      if (flags_bytes) flags_ = reinterpret_cast<std::atomic<uint64_t>*>(board_ + cells_bytes);
    } else if (ok) {
      ::munmap(p, board_bytes_);  // fall back to bootstrap barriers everywhere
    }
  }

  cudecompHandle_t h_;
  std::map<char*, Region> regions_;
  std::vector<hipStream_t> copy_streams_;
  std::vector<hipEvent_t> copy_events_;
  std::atomic<uint64_t>* flags_ = nullptr;
  bool peer_access_done_ = false;
  char* board_ = nullptr;
  size_t board_bytes_ = 0;
};

void peerResetBarrierSlot(cudecompHandle_t h, int slot) {
  if (h->peer) h->peer->resetSlot(slot);
}

int peerProbe(cudecompHandle_t h, void* buffer, size_t bytes) {
  if (h->nranks == 1) return 0;
  if (!h->peer) CD_INVALID_USAGE("peer transport not active for this handle");
  PeerContext& pc = *h->peer;
  const size_t blk = 4096;
  if (bytes < 4 * blk) CD_INVALID_USAGE("buffer too small to probe");
  const int next = (h->rank + 1) % h->nranks, prev = (h->rank + h->nranks - 1) % h->nranks;
  std::vector<size_t> offs = {0, (bytes / 2) & ~(blk - 1), (bytes - blk) & ~(blk - 1)};
  for (size_t g = (size_t)1 << 30; g + blk <= bytes; g += (size_t)1 << 30) offs.push_back(g);  // every GiB boundary
  std::vector<unsigned int> pat(blk / 4);
  char* remote = pc.translate(buffer, next);
  CD_CHECK_HIP(hipMemset(buffer, 0, bytes));
  CD_CHECK_HIP(hipDeviceSynchronize());
  h->boot->barrier();
  for (size_t o : offs) {
    for (size_t i = 0; i < pat.size(); ++i) pat[i] = (unsigned int)(0x9e3779b9u * (h->rank + 1) + o / blk * 2654435761u + i);
    CD_CHECK_HIP(hipMemcpy(remote + o, pat.data(), blk, hipMemcpyHostToDevice));
  }
  CD_CHECK_HIP(hipDeviceSynchronize());
  h->boot->barrier();
  int bad = 0;
  std::vector<unsigned int> got(blk / 4);
  for (size_t o : offs) {
    CD_CHECK_HIP(hipMemcpy(got.data(), static_cast<char*>(buffer) + o, blk, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < got.size(); ++i)
      if (got[i] != (unsigned int)(0x9e3779b9u * (prev + 1) + o / blk * 2654435761u + i)) {
        ++bad;
        break;
      }
  }
  h->boot->barrier();
  return bad;
}

void prepareTransports(cudecompHandle_t h, bool need_rccl, bool need_peer) {
  if (h->nranks == 1) return;  // every communicator has one member: nothing ever travels
  if (need_rccl && !h->rccl) {
    ensureDevice(h);
    h->rccl = std::make_shared<RcclContext>(h);
  }
  if (need_peer && !h->peer) h->peer = std::make_shared<PeerContext>(h);  // touches the device on first use only
}

void* workspaceAllocRaw(cudecompHandle_t h, size_t bytes, bool peer_capable) {
  void* ptr = nullptr;
  if (h->nranks > 1 && peer_capable) {
    // one-sided writes address the peer's workspace by offset: make it the same size everywhere
    bytes = (size_t)h->boot->allreduceMaxI64((int64_t)bytes);
    bytes = ipcSafeSize(bytes);
    prepareTransports(h, false, true);
    CD_CHECK_HIP(hipMalloc(&ptr, bytes));
    try {
      h->peer->registerRegion(ptr, bytes, true);
    } catch (const Error& e) {
      // Agreed on by all ranks (registerRegion fails collectively).  The buffer is still a valid workspace for the
      // RCCL / MPI transports; an operation that needs the one-sided transport will report the IPC problem itself.
      if (h->rank == 0 && !h->ipc_warned) {
        fprintf(stderr, "CUDECOMP:WARN: workspace could not be shared over IPC (%s); one-sided (NVSHMEM*/default MPI*) "
                        "backends are unavailable with it\n", e.what());
      }
      h->ipc_warned = true;
    }
    return ptr;
  }
  CD_CHECK_HIP(hipMalloc(&ptr, bytes));
  return ptr;
}

void workspaceFreeRaw(cudecompHandle_t h, void* ptr) {
  if (h->peer) {
    auto* r = h->peer->find(ptr);
    if (r && r->base == ptr) h->peer->unregisterRegion(ptr);
  }
  CD_CHECK_HIP(hipFree(ptr));
}

void* workspaceAlloc(cudecompHandle_t h, cudecompGridDesc_t gd, size_t bytes) {
  const bool peer_backend = !transposeBackendIsRccl(gd->config.transpose_comm_backend) ||
                            !haloBackendIsRccl(gd->config.halo_comm_backend);
  return workspaceAllocRaw(h, bytes, peer_backend);
}

void workspaceFree(cudecompHandle_t h, cudecompGridDesc_t, void* ptr) { workspaceFreeRaw(h, ptr); }

// ================================================================================================
// all-to-all
// ================================================================================================
namespace {

bool usesRccl(cudecompTransposeCommBackend_t b) { return transposeBackendIsRccl(b); }

// uniform chunks laid out back to back, exchanged by the whole RCCL communicator in rank order
bool nativeAlltoallEligible(cudecompHandle_t h, const cudecompCommInfo& ci, const TransposePlan& p) {
  if (ci.nranks != h->nranks) return false;
  for (int i = 0; i < ci.nranks; ++i) {
    if (ci.global_ranks[i] != i) return false;
    if (p.send_cnt[i] != p.send_cnt[0] || p.recv_cnt[i] != p.send_cnt[0]) return false;
    if (p.send_off[i] != i * p.send_cnt[0] || p.recv_off[i] != i * p.send_cnt[0]) return false;
  }
  return true;
}

void rcclAlltoall(cudecompHandle_t h, cudecompCommInfo& ci, const TransposePlan& p, const ExchangeBuffers& b, int es,
                  hipStream_t stream) {
  if (!h->rccl) CD_INTERNAL_ERROR("RCCL communicator was not created for this grid descriptor");
  ncclComm_t comm = h->rccl->comm();
  if (nativeAlltoallEligible(h, ci, p)) {
    CD_CHECK_RCCL(ncclAllToAll(b.send, b.recv, (size_t)p.send_cnt[0] * es, ncclInt8, comm, stream));
    return;
  }
  CD_CHECK_RCCL(ncclGroupStart());
  for (int i = 0; i < ci.nranks; ++i) {
    const int peer = ci.global_ranks[i];
    if (p.send_cnt[i]) CD_CHECK_RCCL(ncclSend(b.send + p.send_off[i] * es, (size_t)p.send_cnt[i] * es, ncclInt8, peer, comm, stream));
    if (p.recv_cnt[i]) CD_CHECK_RCCL(ncclRecv(b.recv + p.recv_off[i] * es, (size_t)p.recv_cnt[i] * es, ncclInt8, peer, comm, stream));
  }
  CD_CHECK_RCCL(ncclGroupEnd());
}

// (Copies into a peer's mapping use hipMemcpyDefault: source and destination may live on different devices and the
// runtime picks the engine from the pointers.)
// One-sided exchange.  Host-ordered: (1) my chunks are packed (stream sync), (2) everybody's are and
// everybody's receive area is free (barrier), (3) P-1 concurrent xGMI copies, one stream per peer so that
// every link / SDMA queue is busy, (4) all copies landed everywhere (sync + barrier).
void peerAlltoall(cudecompHandle_t h, cudecompCommInfo& ci, const TransposePlan& p, const ExchangeBuffers& b, int es,
                  hipStream_t stream) {
  if (!h->peer) CD_INTERNAL_ERROR("peer transport was not created for this grid descriptor");
  PeerContext& pc = *h->peer;
  CD_CHECK_HIP(hipStreamSynchronize(stream));
  // translate first: registering a foreign receive buffer is itself collective
  std::vector<char*> remote(ci.nranks, nullptr);
  for (int j = 0; j < ci.nranks; ++j) {
    const int d = p.schedule_dst[j];
    remote[d] = pc.translate(b.recv, ci.global_ranks[d]) + p.remote_recv_off[d] * es;
  }
  pc.barrier(ci);
  for (int j = 0; j < ci.nranks; ++j) {
    const int d = p.schedule_dst[j];
    if (p.send_cnt[d] == 0) continue;
    CD_CHECK_HIP(hipMemcpyAsync(remote[d], b.send + p.send_off[d] * es, (size_t)p.send_cnt[d] * es,
                                hipMemcpyDefault, pc.copyStream(j)));
  }
  for (int j = 0; j < ci.nranks; ++j) CD_CHECK_HIP(hipStreamSynchronize(pc.copyStream(j)));
  pc.barrier(ci);
}

}  // namespace

// "SM"-style exchange (NVSHMEM_SM enum): the pack kernels write each chunk straight into the receiver's
// IPC-mapped receive area over xGMI -- one launch feeds all links at once, and the send area, the copy
// engines and one full HBM pass disappear.  The reference's counterpart is the NVSHMEM block-put kernel
// (include/internal/cudecomp_kernels.cuh:86-122); here it is the ordinary move kernel with a remote
// destination.  Host-ordered like peerAlltoall.
void peerPutExchange(cudecompHandle_t h, cudecompCommInfo& ci, const TransposePlan& p, void* const bufs[3], int es,
                     hipStream_t stream) {
  if (!h->peer) CD_INTERNAL_ERROR("peer transport was not created for this grid descriptor");
  PeerContext& pc = *h->peer;
  char* recv_local = static_cast<char*>(bufs[p.recv_buf]) + p.recv_base * es;
  CD_CHECK_HIP(hipStreamSynchronize(stream));
  std::vector<char*> remote(ci.nranks);
  for (int d = 0; d < ci.nranks; ++d) remote[d] = pc.translate(recv_local, ci.global_ranks[d]);
  pc.barrier(ci);  // every member's receive area is free again

  std::vector<Move3D> moves;
  std::vector<void*> dst_base;
  if (!p.pack.empty()) {
    for (const Move3D& m : p.pack) {
      Move3D r = m;
      r.dst_off = p.remote_recv_off[m.peer];
      moves.push_back(r);
      dst_base.push_back(remote[m.peer]);
    }
  } else {  // chunks already sit packed in the send buffer (skip-pack plans): plain copies to the peers
    for (int j = 0; j < ci.nranks; ++j) {
      const int d = p.schedule_dst[j];
      Move3D r;
      r.src_buf = p.send_buf;
      r.src_off = p.send_base + p.send_off[d];
      r.dst_off = p.remote_recv_off[d];
      r.extent[0] = p.send_cnt[d];
      r.ss[0] = r.ds[0] = 1;
      r.peer = d;
      moves.push_back(r);
      dst_base.push_back(remote[d]);
    }
  }
  launchMoves(moves.data(), (int)moves.size(), bufs, es, stream, &h->tuning, nullptr, dst_base.data());
  CD_CHECK_HIP(hipStreamSynchronize(stream));
  pc.barrier(ci);  // every chunk has landed everywhere
}

bool peerPipelineAvailable(cudecompHandle_t h, const cudecompCommInfo& ci) {
  return h->peer && h->peer->pipelineAvailable(ci);
}

// Per-peer pipeline of the one-sided transport (NVSHMEM_PL / MPI_P2P_PL enums): chunk by chunk
//   pack(d) [caller, event per destination] -> copy to d over xGMI [one stream per peer] -> d unpacks it
// with pairwise flags in the shared board instead of communicator-wide barriers: a copy to d starts as soon as
// d's receive area is free and d's chunk is packed, and the unpack of the chunk from s is launched as soon as s
// reports it landed -- packs, the P-1 link transfers and unpacks overlap.  Host-driven like the other
// host-ordered exchanges (returns when every incoming chunk has been handed to an unpack launch).
void peerPipelinedExchange(cudecompHandle_t h, cudecompGridDesc_t gd, cudecompCommInfo& ci, const TransposePlan& plan,
                           void* const bufs[3], const ExchangeBuffers& b, int es, hipEvent_t entry, hipStream_t stream) {
  PeerContext& pc = *h->peer;
  const int P = plan.nranks, me = plan.comm_rank, slot = ci.barrier_slot;
  const uint64_t epoch = ++ci.pipeline_epoch;
  std::vector<char*> remote(P, nullptr);  // registering a foreign receive buffer is collective: do it first
  for (int d = 0; d < P; ++d) remote[d] = pc.translate(b.recv, ci.global_ranks[d]) + plan.remote_recv_off[d] * es;

  // my receive area is free once everything that was on the stream before this call has completed
  CD_CHECK_HIP(hipEventSynchronize(entry));
  pc.ready(slot, h->rank).store(epoch, std::memory_order_release);

  for (int j = 0; j < P; ++j) {
    const int d = (j == 0) ? me : plan.schedule_dst[j];
    hipStream_t cs = pc.copyStream(j);
    CD_CHECK_HIP(hipStreamWaitEvent(cs, gd->events[d], 0));  // chunk for d is packed
    if (d != me) pc.waitFlag(pc.ready(slot, ci.global_ranks[d]), epoch, "receive area of the destination");
    if (plan.send_cnt[d])
      CD_CHECK_HIP(hipMemcpyAsync(remote[d], b.send + plan.send_off[d] * es, (size_t)plan.send_cnt[d] * es,
                                  hipMemcpyDefault, cs));
    CD_CHECK_HIP(hipEventRecord(pc.copyEvent(j), cs));
  }
  for (int j = 0; j < P; ++j) {
    const int d = (j == 0) ? me : plan.schedule_dst[j];
    const int s = (j == 0) ? me : plan.schedule_src[j];
    CD_CHECK_HIP(hipEventSynchronize(pc.copyEvent(j)));  // my chunk for d has landed
    pc.landed(slot, ci.global_ranks[d], h->rank).store(epoch, std::memory_order_release);
    if (s != me) pc.waitFlag(pc.landed(slot, h->rank, ci.global_ranks[s]), epoch, "chunk from the source");
    for (const Move3D& m : plan.unpack)
      if (m.peer == s) launchMoves(&m, 1, bufs, es, stream, &h->tuning);
  }
}

void alltoallExchange(cudecompHandle_t h, cudecompGridDesc_t, cudecompCommInfo& ci, const TransposePlan& plan,
                      const ExchangeBuffers& b, int es, cudecompTransposeCommBackend_t backend, hipStream_t stream) {
  if (usesRccl(backend)) return rcclAlltoall(h, ci, plan, b, es, stream);
#ifdef CUDECOMP_WITH_MPI
  if (transposeBackendIsMpi(backend) && mpiTransportAvailable(ci)) return mpiAlltoall(h, ci, plan, b, es, stream);
#endif
  peerAlltoall(h, ci, plan, b, es, stream);
}

void alltoallExchangePeers(cudecompHandle_t h, cudecompGridDesc_t gd, cudecompCommInfo& ci, const TransposePlan& plan,
                           const ExchangeBuffers& b, int es, cudecompTransposeCommBackend_t backend,
                           const std::vector<int>& src_members, const std::vector<int>& dst_members,
                           hipStream_t stream) {
  if (src_members.empty()) return;
  const int me = ci.rank;
  if (!usesRccl(backend)) {
    // The one-sided transport synchronises on the host, so there is nothing to gain from splitting the
    // exchange by peer: run all of it when the schedule reaches the self step, which comes first.
    if (src_members[0] != me) return;
#ifdef CUDECOMP_WITH_MPI
    if (transposeBackendIsMpi(backend) && mpiTransportAvailable(ci)) return mpiAlltoall(h, ci, plan, b, es, stream);
#endif
    peerAlltoall(h, ci, plan, b, es, stream);
    return;
  }
  if (!h->rccl) CD_INTERNAL_ERROR("RCCL communicator was not created for this grid descriptor");
  if (h->streams.empty()) {
    int lo = 0, hi = 0;
    CD_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t s;
    CD_CHECK_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
    h->streams.push_back(s);
  }
  hipStream_t side = h->streams[0];
  ncclComm_t comm = h->rccl->comm();
  bool grouped = false;
  for (size_t i = 0; i < src_members.size(); ++i) {
    const int s = src_members[i], d = dst_members[i];
    if (s == me) {
      CD_CHECK_HIP(hipMemcpyAsync(b.recv + plan.recv_off[me] * es, b.send + plan.send_off[me] * es,
                                  (size_t)plan.send_cnt[me] * es, hipMemcpyDeviceToDevice, stream));
      continue;
    }
    CD_CHECK_HIP(hipStreamWaitEvent(side, gd->events[d], 0));  // chunk for d is packed
    if (!grouped) {
      CD_CHECK_RCCL(ncclGroupStart());
      grouped = true;
    }
    if (plan.send_cnt[d])
      CD_CHECK_RCCL(ncclSend(b.send + plan.send_off[d] * es, (size_t)plan.send_cnt[d] * es, ncclInt8,
                             ci.global_ranks[d], comm, side));
    if (plan.recv_cnt[s])
      CD_CHECK_RCCL(ncclRecv(b.recv + plan.recv_off[s] * es, (size_t)plan.recv_cnt[s] * es, ncclInt8,
                             ci.global_ranks[s], comm, side));
  }
  if (grouped) CD_CHECK_RCCL(ncclGroupEnd());
  for (size_t i = 0; i < src_members.size(); ++i) {
    if (src_members[i] == me) continue;
    const int d = dst_members[i];
    CD_CHECK_HIP(hipEventRecord(gd->events[d], side));
    CD_CHECK_HIP(hipStreamWaitEvent(stream, gd->events[d], 0));  // chunk has arrived: unpack may start
  }
}

// ================================================================================================
// halo exchange
// ================================================================================================
void haloExchange(cudecompHandle_t h, cudecompGridDesc_t gd, const HaloExchange& x, cudecompHaloCommBackend_t backend,
                  hipStream_t stream) {
  if (haloBackendIsRccl(backend)) {
    if (!h->rccl) CD_INTERNAL_ERROR("RCCL communicator was not created for this grid descriptor");
    ncclComm_t comm = h->rccl->comm();
    // Between one pair of ranks RCCL matches sends and receives in issue order.  With two ranks along a
    // periodic dimension both neighbours are the same peer, so the HIGH face must be sent first: it pairs
    // with the peer's first receive, its LOW halo slot.
    CD_CHECK_RCCL(ncclGroupStart());
    for (int i = 0; i < 2; ++i) {
      const int s = 1 - i;  // face sent in this step
      if (x.neighbor[s] != -1)
        CD_CHECK_RCCL(ncclSend(x.send + x.send_off[s], (size_t)x.bytes, ncclInt8, x.neighbor[s], comm, stream));
      if (x.neighbor[i] != -1)
        CD_CHECK_RCCL(ncclRecv(x.recv + x.recv_off[i], (size_t)x.bytes, ncclInt8, x.neighbor[i], comm, stream));
    }
    CD_CHECK_RCCL(ncclGroupEnd());
    return;
  }
#ifdef CUDECOMP_WITH_MPI
  if (haloBackendIsMpi(backend) && h->boot->nativeComm()) return mpiHaloExchange(h, x, stream);
#endif
  if (!h->peer) CD_INTERNAL_ERROR("peer transport was not created for this grid descriptor");
  PeerContext& pc = *h->peer;
  cudecompCommInfo& ci = gd->comm(x.comm_axis);
  CD_CHECK_HIP(hipStreamSynchronize(stream));
  char* remote[2] = {nullptr, nullptr};
  // registration of a foreign buffer is collective over the world: do it unconditionally and first
  if (pc.anyUnregistered(x.recv)) (void)pc.translate(x.recv, h->rank);
  for (int i = 0; i < 2; ++i)
    if (x.neighbor[i] != -1) remote[i] = pc.translate(x.recv, x.neighbor[i]) + x.remote_off[i];
  pc.barrier(ci);
  for (int i = 0; i < 2; ++i)
    if (x.neighbor[i] != -1)
      CD_CHECK_HIP(hipMemcpyAsync(remote[i], x.send + x.send_off[i], (size_t)x.bytes, hipMemcpyDefault,
                                  pc.copyStream(i)));
  for (int i = 0; i < 2; ++i) CD_CHECK_HIP(hipStreamSynchronize(pc.copyStream(i)));
  pc.barrier(ci);
}

// ------------------------------------------------------------------------------------------------
// packed halo update, pack / exchange / unpack overlapped
// ------------------------------------------------------------------------------------------------
constexpr i64 kHaloOverlapMinBytes = 1 << 20;

bool haloExchangePackedOverlapped(cudecompHandle_t h, cudecompGridDesc_t gd, const HaloExchange& x, const HaloPlan& plan,
                                  void* const bufs[3], int es, cudecompHaloCommBackend_t backend, hipStream_t stream) {
  const bool rccl = haloBackendIsRccl(backend);
#ifdef CUDECOMP_WITH_MPI
  if (!rccl && haloBackendIsMpi(backend) && h->boot->nativeComm()) return false;  // MPI flavour: plain path
#endif
  auto moveOf = [](const std::vector<Move3D>& v, int tag) -> const Move3D* {
    for (const Move3D& m : v)
      if (m.peer == tag) return &m;
    return nullptr;
  };
  if ((int)gd->events.size() < 4) {
    const size_t old = gd->events.size();
    gd->events.resize(4);
    for (size_t i = old; i < gd->events.size(); ++i)
      CD_CHECK_HIP(hipEventCreateWithFlags(&gd->events[i], hipEventDisableTiming));
  }
  hipEvent_t* packed = &gd->events[0];   // [face]
  hipEvent_t* arrived = &gd->events[2];  // [direction]

  // two RCCL groups cost two collective launches: worth it only when a face takes longer to move than to launch
  if (rccl && x.bytes < kHaloOverlapMinBytes && !h->halo_overlap_force) return false;
  if (rccl) {
    if (!h->rccl) CD_INTERNAL_ERROR("RCCL communicator was not created for this grid descriptor");
    ncclComm_t comm = h->rccl->comm();
    if (h->streams.empty()) {
      int lo = 0, hi = 0;
      CD_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      hipStream_t s;
      CD_CHECK_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
      h->streams.push_back(s);
    }
    hipStream_t side = h->streams[0];
    // Direction d moves data towards neighbour d: my face d goes there, and from the OTHER side arrives that
    // neighbour's face d, which fills my halo slot 1-d.  Everybody's direction-d group holds exactly the matching
    // send / receive pairs, so the two groups cannot wait on each other around a periodic ring.
    for (int d = 0; d < 2; ++d) {
      if (const Move3D* m = moveOf(plan.pre, d)) launchMoves(m, 1, bufs, es, stream, &h->tuning);
      CD_CHECK_HIP(hipEventRecord(packed[d], stream));
    }
    for (int d = 0; d < 2; ++d) {
      CD_CHECK_HIP(hipStreamWaitEvent(side, packed[d], 0));
      if (x.neighbor[d] != -1 || x.neighbor[1 - d] != -1) {
        CD_CHECK_RCCL(ncclGroupStart());
        if (x.neighbor[d] != -1)
          CD_CHECK_RCCL(ncclSend(x.send + x.send_off[d], (size_t)x.bytes, ncclInt8, x.neighbor[d], comm, side));
        if (x.neighbor[1 - d] != -1)
          CD_CHECK_RCCL(ncclRecv(x.recv + x.recv_off[1 - d], (size_t)x.bytes, ncclInt8, x.neighbor[1 - d], comm, side));
        CD_CHECK_RCCL(ncclGroupEnd());
      }
      CD_CHECK_HIP(hipEventRecord(arrived[d], side));
      CD_CHECK_HIP(hipStreamWaitEvent(stream, arrived[d], 0));
      if (const Move3D* m = moveOf(plan.post, 1 - d)) launchMoves(m, 1, bufs, es, stream, &h->tuning);
    }
    return true;
  }

  // one-sided transport, host-ordered: the barrier ("every receive slot is free") overlaps the packs, each face's
  // copy waits for its own pack only
  if (!h->peer) CD_INTERNAL_ERROR("peer transport was not created for this grid descriptor");
  PeerContext& pc = *h->peer;
  cudecompCommInfo& ci = gd->comm(x.comm_axis);
  if (!gd->entry_event) CD_CHECK_HIP(hipEventCreateWithFlags(&gd->entry_event, hipEventDisableTiming));
  CD_CHECK_HIP(hipEventRecord(gd->entry_event, stream));
  for (int i = 0; i < 2; ++i) {
    if (const Move3D* m = moveOf(plan.pre, i)) launchMoves(m, 1, bufs, es, stream, &h->tuning);
    CD_CHECK_HIP(hipEventRecord(packed[i], stream));
  }
  char* remote[2] = {nullptr, nullptr};
  if (pc.anyUnregistered(x.recv)) (void)pc.translate(x.recv, h->rank);  // registration is collective: do it first
  for (int i = 0; i < 2; ++i)
    if (x.neighbor[i] != -1) remote[i] = pc.translate(x.recv, x.neighbor[i]) + x.remote_off[i];
  CD_CHECK_HIP(hipEventSynchronize(gd->entry_event));  // my previous use of the receive slots is over
  pc.barrier(ci);
  for (int i = 0; i < 2; ++i)
    if (x.neighbor[i] != -1) {
      CD_CHECK_HIP(hipStreamWaitEvent(pc.copyStream(i), packed[i], 0));
      CD_CHECK_HIP(hipMemcpyAsync(remote[i], x.send + x.send_off[i], (size_t)x.bytes, hipMemcpyDefault,
                                  pc.copyStream(i)));
    }
  for (int i = 0; i < 2; ++i) CD_CHECK_HIP(hipStreamSynchronize(pc.copyStream(i)));
  pc.barrier(ci);
  launchMoves(plan.post.data(), (int)plan.post.size(), bufs, es, stream, &h->tuning);
  return true;
}

}  // namespace cudecomp
