// transport.cc -- see transport.h
#include "transport.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

#include <dirent.h>
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
  if (comm == MPI_COMM_NULL) CD_INVALID_USAGE("null communicator");
  // A program that brought an (MPICH-ABI) MPI with it and initialised it gets exactly the communicator it passed,
  // sub-communicators included; the control plane then runs over that MPI.
  if (auto b = makeDynMpiBootstrap((int)comm)) return b;
  // Otherwise the communicator is only a token for "all ranks the launcher started".
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
// PEER: IPC-mapped buffers + one-sided xGMI copies, ordered ON THE STREAM by flags in a shared board
// ================================================================================================
// Platform quirk (ROCm 7.x, dmabuf IPC): hipIpcOpenMemHandle never returns for an allocation whose byte size has
// bit 31 set (2-4 GiB, 6-8 GiB, ...); every other size maps and transfers correctly (verified block by block with
// cudecompExtPeerProbe up to 17 GiB).  Library allocations are rounded up past such sizes; foreign buffers of
// such a size are reported as not mappable instead of hanging.
inline bool ipcSizeHangs(size_t bytes) { return (bytes & 0x80000000ull) != 0; }
inline size_t ipcSafeSize(size_t bytes) {
  return ipcSizeHangs(bytes) ? ((bytes >> 32) + 1) << 32 : bytes;
}

namespace {
using u64 = unsigned long long;

// what a rank tells the other members of a communicator about one buffer of the current call
struct BufDesc {
  uint64_t offset;       // of the buffer inside its region / allocation
  uint64_t alloc_base;   // foreign buffers: identity of the allocation in the owner's address space
  uint64_t alloc_bytes;
  int64_t region_id;     // >= 0: library region (mapped everywhere since cudecompMalloc); -1: foreign allocation
  uint32_t mappable;     // 0: the owner could not export it
  uint32_t flags;        // call-specific bits
  hipIpcMemHandle_t handle;
};
constexpr int kMailBufs = 2;
struct Mail {
  std::atomic<uint64_t> seq;
  uint64_t nbuf;
  BufDesc buf[kMailBufs];
};
constexpr size_t kMailBytes = (sizeof(Mail) + 63) / 64 * 64;
}  // namespace

class PeerContext {
 public:
  struct Region {
    int64_t id = -1;
    char* base = nullptr;
    size_t bytes = 0;
    std::vector<char*> peer_base;  // by global rank; [my rank] = base
  };
  static constexpr int kSlots = 256;

  // collective over the handle's communicator
  explicit PeerContext(cudecompHandle_t h) : h_(h) {
    debug_ = std::getenv("CUDECOMP_DEBUG_PEER") != nullptr;
    // CUDECOMP_WORKSPACE_POOL_MIB: how much released workspace memory stays parked (0 = none: every cudecompFree really
    // frees); default 1/8 of the device's memory, at most 32 GiB.  CUDECOMP_VERIFY_IPC_MAPPINGS=0 skips the page tags.
    pool_limit_ = (size_t)32 << 30;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) pool_limit_ = std::min(pool_limit_, total_b / 8);
    else (void)hipGetLastError();
    if (const char* v = std::getenv("CUDECOMP_WORKSPACE_POOL_MIB")) pool_limit_ = (size_t)std::strtoull(v, nullptr, 10) << 20;
    if (const char* v = std::getenv("CUDECOMP_VERIFY_IPC_MAPPINGS")) verify_mappings_ = std::strtol(v, nullptr, 10) != 0;
    openBoard();
  }

  ~PeerContext() {
    if (board_registered_) (void)hipHostUnregister(board_);
    if (board_) unmapBoard(board_, board_bytes_);
    for (auto& kv : regions_) closePeers(kv.second);
    for (auto& pk : pool_) (void)hipFree(pk.base);  // parked workspaces: released with the library
    for (auto& kv : imports_)
      if (kv.second.mapped) (void)hipIpcCloseMemHandle(kv.second.mapped);
    for (char* q : retired_imports_) (void)hipIpcCloseMemHandle(q);
    for (int p = 0; p < (int)flag_base_.size(); ++p)
      if (p != h_->rank && flag_base_[p]) (void)hipIpcCloseMemHandle(flag_base_[p]);
    if (flag_mem_) (void)hipFree(flag_mem_);
    if (epoch_slab_) (void)hipFree(epoch_slab_);
    if (verify_count_) (void)hipFree(verify_count_);
    for (hipStream_t s : copy_streams_) (void)hipStreamDestroy(s);
    for (hipEvent_t e : copy_events_) (void)hipEventDestroy(e);
    (void)hipGetLastError();
  }

  bool hasBoard() const { return board_ != nullptr; }

  // ---- library regions (cudecompMalloc): mapped into every rank of the node at allocation -------------------
  Region* find(const void* ptr) {
    const char* p = static_cast<const char*>(ptr);
    auto it = regions_.upper_bound(const_cast<char*>(p));
    if (it == regions_.begin()) return nullptr;
    --it;
    Region& r = it->second;
    return (p >= r.base && p < r.base + r.bytes) ? &r : nullptr;
  }
  Region* findById(int64_t id) {
    auto it = region_by_id_.find(id);
    return it == region_by_id_.end() ? nullptr : find(it->second);
  }

  // collective over the handle's communicator
  // `stale` (optional): set when some member's NEW mapping of a peer's buffer does not lead to that buffer (page tags do
  // not read back); the region is then not registered and nullptr is returned -- agreed by all ranks.
  Region* registerRegion(void* base, size_t bytes, bool* stale = nullptr) {
    struct Wire {
      hipIpcMemHandle_t handle;
      unsigned long long bytes;
      unsigned long long seed;  // of the page tags the owner stamped into the buffer
      int pid;
    };
    enablePeerAccessOnce();
    if (stale) *stale = false;
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
    // Stamp every page before anybody maps it: an importer that ends up with a mapping of something else (the platform
    // can hand back the mapping of a PREDECESSOR of this allocation) must not go unnoticed.
    mine.seed = ((unsigned long long)mine.pid << 32) ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() ^
                ((unsigned long long)next_region_id_ << 20);
    if (verify_mappings_ && error.empty()) {
      bool launched = true;
      try {
        launchTagPages(base, bytes, mine.seed, nullptr);
      } catch (const Error&) {  // a local launch failure is agreed on below like every other failure
        launched = false;
      }
      if (!launched || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        error = "stamping the page tags of a new workspace failed";
        mine.bytes = 0;
      }
    }
    std::vector<Wire> all(h_->nranks);
    h_->boot->allgather(&mine, all.data(), sizeof(Wire));
    Region r;
    r.id = next_region_id_++;  // the same number on every rank: registrations are collective and ordered
    r.base = static_cast<char*>(base);
    r.bytes = bytes;
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
    // read every peer's page tags back through the mapping just made
    bool my_stale = false;
    if (verify_mappings_ && error.empty()) {
      if (!verify_count_) (void)hipMalloc(reinterpret_cast<void**>(&verify_count_), sizeof(unsigned long long));
      for (int p = 0; p < h_->nranks && verify_count_; ++p) {
        if (p == h_->rank || !r.peer_base[p]) continue;
        unsigned long long bad = 0;
        if (hipMemcpy(verify_count_, &bad, sizeof(bad), hipMemcpyHostToDevice) != hipSuccess) break;
        bool launched = true;
        try {
          launchCheckPages(r.peer_base[p], (size_t)all[p].bytes, all[p].seed, verify_count_, nullptr);
        } catch (const Error&) {
          launched = false;
        }
        if (!launched || hipMemcpy(&bad, verify_count_, sizeof(bad), hipMemcpyDeviceToHost) != hipSuccess) {
          (void)hipGetLastError();
          bad = 1;
        }
        if (bad) {
          my_stale = true;
          stale_mappings_seen_++;
          if (debug_ || std::getenv("CUDECOMP_VERBOSE"))
            fprintf(stderr, "CUDECOMP:WARN rank %d: the new IPC mapping of rank %d's workspace (%llu bytes) does not show its "
                            "page tags on %llu of %llu pages: stale mapping, the workspace will be re-created\n", h_->rank, p,
                    all[p].bytes, bad, all[p].bytes / 4096);
        }
      }
    }
    const bool anyone_failed = h_->boot->allreduceOr(!error.empty());  // also: everybody has finished mapping
    if (anyone_failed) {
      closePeers(r);
      CD_PEER_ERROR(error.empty() ? std::string("IPC mapping failed on another rank") : error);
    }
    if (verify_mappings_ && h_->boot->allreduceOr(my_stale)) {
      closePeers(r);
      h_->boot->barrier();  // everybody has let go of everybody's buffer: the owners may dispose of them
      if (stale) *stale = true;
      if (!stale) CD_PEER_ERROR("a new IPC mapping of a peer's workspace does not lead to that workspace (stale mapping)");
      return nullptr;
    }
    region_by_id_[r.id] = r.base;
    h_->region_generation++;
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
    (void)hipDeviceSynchronize();  // my copies into the peers' mappings are done ...
    h_->boot->barrier();           // ... and so are everybody else's into mine
    closePeers(it->second);
    region_by_id_.erase(it->second.id);
    regions_.erase(it);
    h_->region_generation++;
    h_->boot->barrier();
  }

  // ---- pool of released regions ---------------------------------------------------------------------------------
  // cudecompFree parks a library region -- allocation AND the peers' mappings stay -- and cudecompMalloc hands it out
  // again for requests it fits.  Two reasons: mapping a buffer into every rank of the node is the expensive part of
  // cudecompMalloc, and re-creating allocations is what exposes the platform's stale-IPC-mapping behaviour (DESIGN.md
  // section 9).  Every rank sees the same sequence of (collective) malloc / free calls with the same (max-reduced)
  // sizes, so the pools are identical everywhere and so is every decision taken from them.
  void* takeFromPool(size_t bytes) {
    int best = -1;
    for (int i = 0; i < (int)pool_.size(); ++i) {
      if (pool_[i].bytes < bytes || pool_[i].bytes > std::max(2 * bytes, bytes + ((size_t)64 << 20))) continue;
      if (best < 0 || pool_[i].bytes < pool_[best].bytes) best = i;
    }
    if (best < 0) return nullptr;
    void* p = pool_[best].base;
    pool_bytes_ -= pool_[best].bytes;
    pool_.erase(pool_.begin() + best);
    pool_hits_++;
    return p;
  }
  // returns the regions that must really be released now (oldest first) to stay under the pool's limit
  std::vector<void*> park(void* base) {
    std::vector<void*> evict;
    Region* r = find(base);
    if (!r || r->base != base) return evict;
    pool_.push_back(Parked{r->base, r->bytes});
    pool_bytes_ += r->bytes;
    while (pool_.size() > 1 && (pool_bytes_ > pool_limit_ || (int)pool_.size() > kMaxParked)) {
      evict.push_back(pool_.front().base);
      pool_bytes_ -= pool_.front().bytes;
      pool_.erase(pool_.begin());
    }
    if (pool_.size() == 1 && pool_bytes_ > pool_limit_) {  // larger than the whole pool: not kept at all
      evict.push_back(pool_.front().base);
      pool_bytes_ = 0;
      pool_.clear();
    }
    return evict;
  }
  bool poolEnabled() const { return pool_limit_ > 0; }
  // The pools and their eviction lists must be identical on every rank (park / take decisions involve collectives): the
  // limit each rank derived from ITS device (or environment) is replaced by the smallest one.  Collective; called once,
  // when the transport comes up, before any park or take.
  void agreePoolLimit() {
    const int64_t mine = (int64_t)std::min<size_t>(pool_limit_, (size_t)1 << 62);
    pool_limit_ = (size_t)(-h_->boot->allreduceMaxI64(-mine));
  }
  size_t poolBytes() const { return pool_bytes_; }
  // Retired mappings of re-created user buffers (see map()) each keep the exporter's FREED memory referenced: keep the
  // newest `keep`, close the older ones.  Called where every rank has drained its device and nothing is being imported
  // (grid descriptor destruction), never right before an import -- that is the sequence section 9 A warns about.
  void trimRetiredImports(size_t keep) {
    if (retired_imports_.size() <= keep) return;
    (void)hipDeviceSynchronize();  // nothing this rank enqueued may still be copying through them
    const size_t n = retired_imports_.size() - keep;
    for (size_t i = 0; i < n; ++i) (void)hipIpcCloseMemHandle(retired_imports_[i]);
    (void)hipGetLastError();
    retired_imports_.erase(retired_imports_.begin(), retired_imports_.begin() + n);
  }
  size_t retiredImports() const { return retired_imports_.size(); }
  std::vector<void*> drainPool() {
    std::vector<void*> all;
    for (auto& p : pool_) all.push_back(p.base);
    pool_.clear();
    pool_bytes_ = 0;
    return all;
  }
  int64_t poolHits() const { return pool_hits_; }
  int64_t staleMappingsSeen() const { return stale_mappings_seen_; }

  // ---- host barrier among the members of a row / column communicator (set-up paths, probes) ---------------
  void barrier(cudecompCommInfo& ci) {
    if (!board_ || ci.ngroups != 1 || ci.barrier_slot < 0) {
      ci.boot->barrier();
      return;
    }
    const uint64_t epoch = ++ci.barrier_epoch;
    cell(ci.barrier_slot, h_->rank).store(epoch, std::memory_order_release);
    for (int m = 0; m < ci.nranks; ++m)
      spinUntil(cell(ci.barrier_slot, ci.global_ranks[m]), epoch, "the shared-memory barrier");
  }

  // ---- device view of the board ----------------------------------------------------------------------
  // The board is registered with the HIP runtime on first use (geometry-only runs never touch a device).
  void ensureDeviceView() {
    if (dboard_) return;
    if (!board_) CD_PEER_ERROR("the shared-memory board of the peer transport could not be set up on this node");
    CD_CHECK_HIP(hipHostRegister(board_, board_bytes_, hipHostRegisterMapped | hipHostRegisterPortable));
    board_registered_ = true;
    void* d = nullptr;
    CD_CHECK_HIP(hipHostGetDevicePointer(&d, board_, 0));
    dboard_ = static_cast<char*>(d);
  }
  bool usable(const cudecompCommInfo& ci) const { return board_ && ci.ngroups == 1 && ci.barrier_slot >= 0; }

  // Where the flags live.  Board mode (fallback): row `rank` of the host-pinned board holds ready[rank] and
  // landed[rank][*]; everybody polls and writes host memory (about 8 us per flag round trip between processes).
  // Device mode: every rank owns a small UNCACHED device buffer, IPC-mapped into all ranks of the node once, when the
  // transport comes up; a flag is always POLLED in the poller's own HBM and WRITTEN into the poller's buffer by whoever
  // raises it (one posted store over xGMI) -- the counterpart of the reference's NVSHMEM signals in the symmetric heap
  // (include/internal/cudecomp_kernels.cuh:51-84).  Row layout per slot: ready_from[nranks], landed[landed_n].
  //   dReady(slot, r)        address I poll to see rank r's "begun" flag
  //   dReadyAt(slot, m)      address rank m polls to see MINE (written by my epoch kernel)
  //   dLanded(slot, dst, i)  landed flag i of rank dst: polled by dst (dst == me: local), written by the sender
  u64* dReady(int slot, int rank) {
    if (flag_base_.empty()) return reinterpret_cast<u64*>(dboard_ + flagRowOff(slot, rank));
    return flag_base_[h_->rank] + (size_t)slot * flagRowWords() + rank;
  }
  u64* dReadyAt(int slot, int at_rank) {
    if (flag_base_.empty()) return reinterpret_cast<u64*>(dboard_ + flagRowOff(slot, h_->rank));
    return flag_base_[at_rank] + (size_t)slot * flagRowWords() + h_->rank;
  }
  u64* dLanded(int slot, int dst, int idx) {
    if (flag_base_.empty()) return reinterpret_cast<u64*>(dboard_ + flagRowOff(slot, dst)) + 1 + idx;
    return flag_base_[dst] + (size_t)slot * flagRowWords() + h_->nranks + idx;
  }
  bool deviceFlags() const { return !flag_base_.empty(); }
  size_t flagRowWords() const { return (size_t)h_->nranks + landed_n_; }

  // collective over the handle's communicator (called when the transport comes up on a job that has devices)
  void setupDeviceFlags() {
    // OPT-IN (CUDECOMP_FLAGS_IN_DEVICE_MEMORY=1).  Measured with ranks sharing one GPU (profiles/r03_tuning.md): per
    // exchange the device flags are as fast or faster than the board, but eight ranks on one device ran the case sweeps
    // 2.7x slower with them and, rarely, a wait kernel never saw a flag that had been raised (30-s device-side
    // timeouts in two suite runs); the board has no such history, so it stays the default until the device flags can
    // be qualified on a node with a GPU per rank.
    const char* want = std::getenv("CUDECOMP_FLAGS_IN_DEVICE_MEMORY");
    if (!board_ || h_->nranks > 64 || !want || std::strtol(want, nullptr, 10) != 1) return;
    for (int r = 0; r < h_->nranks; ++r)
      if (h_->hostnames[r] != h_->hostnames[h_->rank]) return;  // (multi-node jobs keep the per-node board)
    if (h_->rank == 0)
      fprintf(stderr, "CUDECOMP:WARN: CUDECOMP_FLAGS_IN_DEVICE_MEMORY=1 is EXPERIMENTAL: not yet qualified with one GPU per rank "
                      "(profiles/r05_tuning.md); the default keeps the flags in host-pinned memory.\n");
    const auto t_setup = std::chrono::steady_clock::now();
    struct Wire {
      hipIpcMemHandle_t handle;
      int ok;
    };
    Wire mine{};
    const size_t bytes = ((size_t)kSlots * flagRowWords() * sizeof(u64) + 4095) / 4096 * 4096;
    u64* buf = nullptr;
    enablePeerAccessOnce();
    bool ok = hipExtMallocWithFlags(reinterpret_cast<void**>(&buf), bytes, hipDeviceMallocUncached) == hipSuccess && buf;
    if (ok) ok = hipMemset(buf, 0, bytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    if (ok) ok = hipIpcGetMemHandle(&mine.handle, buf) == hipSuccess;
    (void)hipGetLastError();
    mine.ok = ok ? 1 : 0;
    std::vector<Wire> all(h_->nranks);
    h_->boot->allgather(&mine, all.data(), sizeof(Wire));
    for (auto& w : all) ok = ok && w.ok;
    std::vector<u64*> base(h_->nranks, nullptr);
    if (ok) {
      for (int p = 0; p < h_->nranks && ok; ++p) {
        if (p == h_->rank) {
          base[p] = buf;
          continue;
        }
        void* m = nullptr;
        ok = hipIpcOpenMemHandle(&m, all[p].handle, hipIpcMemLazyEnablePeerAccess) == hipSuccess;
        base[p] = static_cast<u64*>(m);
      }
      (void)hipGetLastError();
    }
    const bool all_ok = !h_->boot->allreduceOr(!ok);  // also: everybody has finished mapping
    if (all_ok) {
      flag_base_ = base;
      flag_mem_ = buf;
    } else {
      for (int p = 0; p < h_->nranks; ++p)
        if (p != h_->rank && base[p]) (void)hipIpcCloseMemHandle(base[p]);
      if (buf) (void)hipFree(buf);
      (void)hipGetLastError();
    }
    h_->boot->barrier();
    if (h_->rank == 0 && std::getenv("CUDECOMP_VERBOSE"))
      fprintf(stderr, "CUDECOMP: one-sided exchange flags live in %s (set-up %.1f ms)\n",
              all_ok ? "device memory (polled locally, written by the peer)" : "the host-pinned board",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_setup).count());
  }
  u64* dStatus() { return reinterpret_cast<u64*>(dboard_ + status_off_ + (size_t)h_->rank * 64); }
  // a wait kernel of an earlier call gave up: report it now (the data of that call is incomplete)
  void checkStatus() {
    if (!board_) return;
    auto& st = *reinterpret_cast<std::atomic<uint64_t>*>(board_ + status_off_ + (size_t)h_->rank * 64);
    const uint64_t v = st.load(std::memory_order_relaxed);
    if (v == 0) return;
    st.store(0, std::memory_order_relaxed);
    CD_PEER_ERROR("a one-sided exchange timed out on the device waiting for a peer's signal (call " +
                  std::to_string(v >> 8) + ", flag " + std::to_string((v & 0xff) - 1) + "): a peer rank died or never "
                  "entered the matching call; the results of that exchange are incomplete");
  }

  // Device call counter of a communicator: one cell per board row (slot) in a slab that lives as long as this context.
  // The cells used to be a hipMalloc / hipFree per communicator, so a communicator created right after another one was
  // destroyed got the SAME address back -- and with several processes on one GPU a compute die can keep serving a re-allocated
  // address's previous life (found on the test programs' data buffers).  A call counter read stale would
  // put a rank's call numbers out of step with its peers' and let flags of earlier calls satisfy its waits.  Never observed
  // (round 6 suspected it for a failure that turned out to be a test's own race, profiles/r06_pooled_suite_failure.md); kept
  // as the sturdier design: never re-allocated while the handle lives, uncached (no compute-die cache holds them), written
  // and read with system-scope atomics, and a communicator that takes over a row writes its agreed base into the row's cell.
  u64* devEpoch(cudecompCommInfo& ci) {
    if (!ci.dev_epoch) {
      if (ci.barrier_slot < 0) CD_INTERNAL_ERROR("one-sided exchange on a communicator without a board row");
      if (!epoch_slab_) {
        const size_t bytes = (size_t)kSlots * kEpochStride * sizeof(u64);
        if (hipExtMallocWithFlags(reinterpret_cast<void**>(&epoch_slab_), bytes, hipDeviceMallocUncached) != hipSuccess || !epoch_slab_) {
          (void)hipGetLastError();
          CD_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&epoch_slab_), bytes));
        }
        CD_CHECK_HIP(hipMemset(epoch_slab_, 0, bytes));
      }
      ci.dev_epoch = epoch_slab_ + (size_t)ci.barrier_slot * kEpochStride;
      const u64 base = ci.epoch_base;
      CD_CHECK_HIP(hipMemcpy(ci.dev_epoch, &base, sizeof(base), hipMemcpyHostToDevice));
      // the counter is first touched by a kernel on the CALLER's stream, which need not be ordered behind a synchronous
      // copy from pageable memory (it may return once the bytes are staged): make sure they are in place -- once per
      // communicator
      CD_CHECK_HIP(hipDeviceSynchronize());
    }
    return ci.dev_epoch;
  }

  // ---- per-call rendezvous: every member publishes where its buffers are, then reads the others' ------------
  // Returns remote[b][m]: the address through which THIS rank writes member m's buffer b (nullptr: not reachable),
  // and flags[b][m] as posted by m.  Blocks the host until every member has ENTERED the same call (not until any
  // GPU work is done).  Handles buffers that are not from cudecompMalloc, at any offset of any allocation, and
  // notices when an allocation was freed and re-created at the same address.
  struct Resolved {
    std::vector<char*> remote[kMailBufs];
    std::vector<uint32_t> flags[kMailBufs], mappable[kMailBufs];
  };
  Resolved rendezvous(cudecompCommInfo& ci, const void* const* ptrs, const uint32_t* flags, int nbuf) {
    if (!usable(ci)) CD_PEER_ERROR("the one-sided transport needs all members of the communicator on one node");
    const uint64_t seq = ++ci.mail_seq;
    Mail& mine = mail(ci.barrier_slot, h_->rank, (int)(seq & 1));
    // every descriptor of the mailbox is (re)written on every call: a member that posts fewer buffers than a peer asks
    // for must never show that peer what it posted two calls ago
    mine.nbuf = (uint64_t)nbuf;
    for (int b = 0; b < kMailBufs; ++b) mine.buf[b] = (b < nbuf) ? describe(ptrs[b], flags ? flags[b] : 0) : describe(nullptr, 0);
    mine.seq.store(seq, std::memory_order_release);
    Resolved out;
    for (int b = 0; b < nbuf; ++b) {
      out.remote[b].assign(ci.nranks, nullptr);
      out.flags[b].assign(ci.nranks, 0);
      out.mappable[b].assign(ci.nranks, 0);
    }
    for (int m = 0; m < ci.nranks; ++m) {
      const int g = ci.global_ranks[m];
      Mail& theirs = mail(ci.barrier_slot, g, (int)(seq & 1));
      spinUntil(theirs.seq, seq, "the per-call rendezvous of a one-sided exchange");
      const int their_nbuf = (int)std::min<uint64_t>(theirs.nbuf, (uint64_t)kMailBufs);
      for (int b = 0; b < nbuf; ++b) {
        // a buffer the member did not post counts as "not offered" (flags 0, not mappable): the members of a call may
        // disagree about optional buffers (the direct put's output pencil), never about what that means
        const BufDesc d = (b < their_nbuf) ? theirs.buf[b] : BufDesc{};
        out.flags[b][m] = d.flags;
        out.mappable[b][m] = d.mappable;
        if (g == h_->rank) out.remote[b][m] = static_cast<char*>(const_cast<void*>(ptrs[b]));
        else out.remote[b][m] = map(g, d);
        if (debug_)
          fprintf(stderr, "CUDECOMP:DEBUG rank %d slot %d seq %llu buf %d member %d (rank %d): region %lld base %llx bytes %llu "
                          "off %llu mappable %u flags %u -> %p\n", h_->rank, ci.barrier_slot, (unsigned long long)seq, b, m, g,
                  (long long)d.region_id, (unsigned long long)d.alloc_base, (unsigned long long)d.alloc_bytes,
                  (unsigned long long)d.offset, d.mappable, d.flags, (void*)out.remote[b][m]);
      }
    }
    return out;
  }

  // symmetric resolution without any host communication: the buffer must come from cudecompMalloc and sit at the same
  // offset of its region on every rank (the contract of the reference's NVSHMEM backends)
  std::vector<char*> symmetric(const cudecompCommInfo& ci, const void* ptr, const char* what) {
    Region* r = find(ptr);
    if (!r)
      CD_INVALID_USAGE(std::string(what) + " must be allocated with cudecompMalloc for the NVSHMEM backends (symmetric "
                       "one-sided access); use an MPI_* backend for arbitrary device buffers");
    std::vector<char*> out(ci.nranks);
    for (int m = 0; m < ci.nranks; ++m) {
      char* pb = r->peer_base[ci.global_ranks[m]];
      if (!pb) CD_PEER_ERROR("peer buffer is not reachable over xGMI/IPC (rank on another host?)");
      out[m] = pb + (static_cast<const char*>(ptr) - r->base);
    }
    return out;
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

  bool debug() const { return debug_; }
  // debugging aid (CUDECOMP_DEBUG_PEER=1): device call counter and the flags of my row, read on the host after a device sync
  void dumpRow(cudecompCommInfo& ci, const char* when) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(nullptr, &cap);  // (never inside a capture: the dump synchronizes)
    if (hipGetLastError() != hipSuccess || cap != hipStreamCaptureStatusNone) return;
    if (hipDeviceSynchronize() != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    unsigned long long e = 0;
    if (ci.dev_epoch) (void)hipMemcpy(&e, ci.dev_epoch, sizeof(e), hipMemcpyDeviceToHost);
    char line[512];
    int n = snprintf(line, sizeof(line), "CUDECOMP:DEBUG rank %d %s slot %d P %d base %llu device epoch %llu ready", h_->rank, when,
                     ci.barrier_slot, ci.nranks, (unsigned long long)ci.epoch_base, e);
    if (board_ && ci.barrier_slot >= 0) {
      const u64* row = reinterpret_cast<const u64*>(board_ + flagRowOff(ci.barrier_slot, h_->rank));
      n += snprintf(line + n, sizeof(line) - n, " %llu landed", (unsigned long long)row[0]);
      for (int i = 0; i < landed_n_ && n < 480; ++i) n += snprintf(line + n, sizeof(line) - n, " %llu", (unsigned long long)row[1 + i]);
    }
    fprintf(stderr, "%s\n", line);
  }

  // highest counter value any rank may have left in row `slot` that involves me
  uint64_t slotHigh(int slot) {
    if (!board_ || slot < 0) return 0;
    uint64_t v = cell(slot, h_->rank).load(std::memory_order_relaxed);
    const u64* row = reinterpret_cast<const u64*>(board_ + flagRowOff(slot, h_->rank));
    for (int i = 0; i < 1 + landed_n_; ++i)  // (flags hold call * kFlagScale + step: round up to whole calls)
      v = std::max<uint64_t>(v, (reinterpret_cast<const std::atomic<uint64_t>*>(row + i)->load() + kFlagScale - 1) / kFlagScale);
    for (int par = 0; par < 2; ++par) v = std::max<uint64_t>(v, mail(slot, h_->rank, par).seq.load());
    // (flags in device memory are not read back: whatever anybody wrote into my buffer carries that rank's call number
    // on the communicator that used the slot, which that rank reports itself -- buildCommInfo reduces over ALL ranks)
    return v;
  }

 private:
  void spinUntil(std::atomic<uint64_t>& a, uint64_t v, const char* what) {
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(h_->peer_timeout_s);
    int spins = 0;
    while (a.load(std::memory_order_acquire) < v) {
      if (++spins > 2000) {
        std::this_thread::yield();
        if ((spins & 0xfff) == 0 && std::chrono::steady_clock::now() > deadline)
          CD_PEER_ERROR(std::string("timed out waiting for a peer rank in ") + what + " (did it die, or skip this call?)");
      }
    }
  }

  // ---- foreign buffers: export on demand, cached per allocation; validated on every call ---------------------
  struct Export {
    size_t bytes = 0;
    unsigned long long buffer_id = 0;
    bool has_id = false;
    hipIpcMemHandle_t handle;
  };
  struct Import {
    hipIpcMemHandle_t handle;
    size_t bytes = 0;
    char* mapped = nullptr;
  };

  BufDesc describe(const void* ptr, uint32_t flags) {
    BufDesc d{};
    d.flags = flags;
    d.region_id = -1;
    if (!ptr) return d;
    if (Region* r = find(ptr)) {
      d.region_id = r->id;
      d.offset = (uint64_t)(static_cast<const char*>(ptr) - r->base);
      d.mappable = 1;
      return d;
    }
    void* base = nullptr;
    size_t bytes = 0;
    if (hipMemGetAddressRange(&base, &bytes, const_cast<void*>(ptr)) != hipSuccess || ipcSizeHangs(bytes)) {
      (void)hipGetLastError();
      return d;  // not a device allocation, or of a size this platform cannot share: mappable = 0
    }
    d.alloc_base = (uint64_t)(uintptr_t)base;
    d.alloc_bytes = bytes;
    d.offset = (uint64_t)(static_cast<const char*>(ptr) - static_cast<const char*>(base));
    // Same address and size as last time does not mean the same allocation (free + malloc can return it again): the
    // runtime's buffer id tells them apart; without one the handle is re-exported on every call.
    unsigned long long id = 0;
    const bool has_id = hipPointerGetAttribute(&id, HIP_POINTER_ATTRIBUTE_BUFFER_ID, const_cast<void*>(ptr)) == hipSuccess;
    if (!has_id) (void)hipGetLastError();
    auto it = exports_.find((char*)base);
    if (it == exports_.end() || it->second.bytes != bytes || !has_id || !it->second.has_id || it->second.buffer_id != id) {
      Export e;
      e.bytes = bytes;
      e.buffer_id = id;
      e.has_id = has_id;
      enablePeerAccessOnce();
      if (hipIpcGetMemHandle(&e.handle, base) != hipSuccess) {
        (void)hipGetLastError();
        exports_.erase((char*)base);
        return d;
      }
      it = exports_.insert_or_assign((char*)base, e).first;
    }
    d.handle = it->second.handle;
    d.mappable = 1;
    return d;
  }

  char* map(int g, const BufDesc& d) {
    if (!d.mappable) return nullptr;
    if (d.region_id >= 0) {
      Region* r = findById(d.region_id);
      if (!r || !r->peer_base[g] || d.offset >= r->bytes) return nullptr;
      return r->peer_base[g] + d.offset;
    }
    if (h_->hostnames[g] != h_->hostnames[h_->rank]) return nullptr;
    const auto key = std::make_pair(g, d.alloc_base);
    auto it = imports_.find(key);
    if (it != imports_.end() &&
        (it->second.bytes != d.alloc_bytes || std::memcmp(&it->second.handle, &d.handle, sizeof(d.handle)) != 0)) {
      // the owner freed that allocation and made a new one at the same address.  The old mapping is RETIRED, not closed:
      // closing a mapping right before importing its successor is the sequence that can hand back the predecessor's
      // memory on this platform, while importers that never close are safe (scripts/probe/ipc_remap_probe.cpp, modes
      // 1 vs 257).  Retired mappings are closed with the transport.
      (void)hipDeviceSynchronize();
      if (it->second.mapped) retired_imports_.push_back(it->second.mapped);
      imports_.erase(it);
      it = imports_.end();
    }
    if (it == imports_.end()) {
      Import im;
      im.handle = d.handle;
      im.bytes = d.alloc_bytes;
      void* mapped = nullptr;
      enablePeerAccessOnce();
      if (hipIpcOpenMemHandle(&mapped, d.handle, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
      }
      im.mapped = static_cast<char*>(mapped);
      it = imports_.emplace(key, im).first;
    }
    return it->second.mapped + d.offset;
  }

  void closePeers(Region& r) {
    for (int p = 0; p < (int)r.peer_base.size(); ++p)
      if (p != h_->rank && r.peer_base[p]) (void)hipIpcCloseMemHandle(r.peer_base[p]);
  }

  // ---- board layout ----------------------------------------------------------------------------------------
  //   cells   [kSlots][nranks] x 64 B    host barrier epochs
  //   flags   [kSlots][nranks] rows      u64 ready; u64 landed[max(nranks, 2)]   (row padded to 64 B)
  //   mail    [kSlots][nranks][2]        per-call buffer descriptors, double buffered by call parity
  //   status  [nranks] x 64 B            written by a wait kernel that gave up
  size_t flagRowOff(int slot, int rank) const { return flags_off_ + ((size_t)slot * h_->nranks + rank) * flag_row_bytes_; }
  std::atomic<uint64_t>& cell(int slot, int rank) {
    return *reinterpret_cast<std::atomic<uint64_t>*>(board_ + ((size_t)slot * h_->nranks + rank) * 64);
  }
  Mail& mail(int slot, int rank, int parity) {
    return *reinterpret_cast<Mail*>(board_ + mail_off_ + (((size_t)slot * h_->nranks + rank) * 2 + parity) * kMailBytes);
  }

  // Where a board is mapped.  The board is registered with HIP and polled by kernels; a process that finalizes a handle and
  // creates another (a job server, a long-lived test worker) would normally get the NEW board at the address the old one
  // had.  Given what this platform does with recycled addresses when several processes share a GPU
  // boards are placed in a reserved address arena, each at a fresh address, and a retired board's range stays reserved: no
  // board address is ever used twice by a process.  A precaution, not the fix of an observed failure (round 6 suspected a
  // recycled board for a failure that was a test's own race: arms with and without the arena failed alike,
  // profiles/r06_pooled_suite_failure.md).  CUDECOMP_BOARD_FRESH_ADDRESS=0 restores plain mmap.
  static constexpr size_t kArenaBytes = (size_t)4 << 30, kArenaAlign = (size_t)2 << 20;
  static char*& arenaBase() {
    static char* base = nullptr;
    return base;
  }
  static size_t& arenaUsed() {
    static size_t used = 0;
    return used;
  }
  static bool freshAddresses() {
    const char* v = std::getenv("CUDECOMP_BOARD_FRESH_ADDRESS");
    return !v || std::strtol(v, nullptr, 10) != 0;
  }
  void* mapBoard(int fd, size_t bytes) {
    board_in_arena_ = false;
    if (freshAddresses()) {
      if (!arenaBase()) {
        void* a = ::mmap(nullptr, kArenaBytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (a != MAP_FAILED) arenaBase() = static_cast<char*>(a);
      }
      const size_t need = (bytes + kArenaAlign - 1) / kArenaAlign * kArenaAlign;
      if (arenaBase() && arenaUsed() + need <= kArenaBytes) {
        void* p = ::mmap(arenaBase() + arenaUsed(), bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0);
        if (p != MAP_FAILED) {
          arenaUsed() += need;
          board_in_arena_ = true;
          return p;
        }
      }
    }
    return ::mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  }
  void unmapBoard(void* p, size_t bytes) {
    // inside the arena the range goes back to "reserved, inaccessible": the shared pages are dropped, the address is not reused
    if (board_in_arena_) (void)::mmap(p, bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED, -1, 0);
    else ::munmap(p, bytes);
  }

  void openBoard() {
    // every host gets its own segment; its name is agreed through the bootstrap, the creator unlinks it as
    // soon as all local ranks have mapped it, so nothing is left behind even if a rank crashes later
    landed_n_ = std::max(h_->nranks, 2);
    flag_row_bytes_ = ((size_t)(1 + landed_n_) * sizeof(uint64_t) + 63) / 64 * 64;
    const size_t cells_bytes = (size_t)kSlots * h_->nranks * 64;
    flags_off_ = cells_bytes;
    mail_off_ = flags_off_ + (size_t)kSlots * h_->nranks * flag_row_bytes_;
    status_off_ = mail_off_ + (size_t)kSlots * h_->nranks * 2 * kMailBytes;
    board_bytes_ = (status_off_ + (size_t)h_->nranks * 64 + 4095) / 4096 * 4096;
    if (h_->nranks > 64) return;  // the flag lists of the wait / signal kernels hold 64 entries: no peer transport
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
    void* p = (fd >= 0) ? mapBoard(fd, board_bytes_) : MAP_FAILED;
    if (fd >= 0) ::close(fd);
    const bool ok = (p != MAP_FAILED);
    const bool all_ok = !h_->boot->allreduceOr(!ok);  // also: everybody has mapped it
    if (h_->local_rank == 0) ::shm_unlink(shm_name);
    if (ok && all_ok) board_ = static_cast<char*>(p);
    else if (ok) unmapBoard(p, board_bytes_);  // bootstrap barriers, no one-sided transport
  }

  cudecompHandle_t h_;
  std::map<char*, Region> regions_;
  std::map<int64_t, char*> region_by_id_;
  int64_t next_region_id_ = 0;
  std::map<char*, Export> exports_;
  std::map<std::pair<int, uint64_t>, Import> imports_;
  std::vector<char*> retired_imports_;
  std::vector<hipStream_t> copy_streams_;
  std::vector<hipEvent_t> copy_events_;
  std::vector<u64*> flag_base_;  // device-memory flags: base of every rank's flag buffer as mapped here (empty: board mode)
  u64* flag_mem_ = nullptr;      // my own
  static constexpr int kEpochStride = 8;  // u64 words between the epoch cells of two rows (one 64-byte unit each)
  u64* epoch_slab_ = nullptr;    // the communicators' device call counters, one cell per board row (see devEpoch)
  struct Parked {
    char* base;
    size_t bytes;
  };
  static constexpr int kMaxParked = 32;
  std::vector<Parked> pool_;
  size_t pool_bytes_ = 0, pool_limit_ = 0;
  int64_t pool_hits_ = 0, stale_mappings_seen_ = 0;
  bool verify_mappings_ = true;
  unsigned long long* verify_count_ = nullptr;
  bool peer_access_done_ = false;
  bool debug_ = false;
  bool board_registered_ = false;
  bool board_in_arena_ = false;
  char* board_ = nullptr;
  char* dboard_ = nullptr;
  size_t board_bytes_ = 0, flags_off_ = 0, mail_off_ = 0, status_off_ = 0, flag_row_bytes_ = 0;
  int landed_n_ = 2;
};

uint64_t peerSlotHigh(cudecompHandle_t h, int slot) { return h->peer ? h->peer->slotHigh(slot) : 0; }

void peerPoolCounters(cudecompHandle_t h, int64_t* pool_hits, int64_t* stale_mappings, int64_t* pool_bytes, int64_t* retired) {
  *pool_hits = h->peer ? h->peer->poolHits() : 0;
  *stale_mappings = h->peer ? h->peer->staleMappingsSeen() : 0;
  if (pool_bytes) *pool_bytes = h->peer ? (int64_t)h->peer->poolBytes() : 0;
  if (retired) *retired = h->peer ? (int64_t)h->peer->retiredImports() : 0;
}

void peerTrimRetiredImports(cudecompHandle_t h, size_t keep) {
  if (h->peer) h->peer->trimRetiredImports(keep);
}

namespace {
bool readSmallFile(const std::string& path, std::string* out) {
  FILE* f = std::fopen(path.c_str(), "r");
  if (!f) return false;
  char buf[256];
  const size_t n = std::fread(buf, 1, sizeof(buf) - 1, f);
  std::fclose(f);
  buf[n] = 0;
  *out = buf;
  while (!out->empty() && (out->back() == '\n' || out->back() == ' ')) out->pop_back();
  return true;
}
std::vector<std::string> listDir(const std::string& path) {
  std::vector<std::string> out;
  if (DIR* d = ::opendir(path.c_str())) {
    while (dirent* e = ::readdir(d))
      if (e->d_name[0] != '.') out.push_back(e->d_name);
    ::closedir(d);
  }
  return out;
}
}  // namespace

bool peerQueueCensusRequested() {
  const char* v = std::getenv("CUDECOMP_QUEUE_CENSUS");
  return v && std::strtol(v, nullptr, 10) != 0;
}

int peerQueueCensus(cudecompHandle_t h, bool warn, int* slots_out) {
  // Which KFD node is my GPU?  (Not through my pid: inside a container's pid namespace getpid() is not the pid the driver
  // files my queues under.)  The topology node with my PCI location.
  if (slots_out) *slots_out = 0;
  if (h->device < 0) return -1;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), h->device) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  unsigned int dom = 0, b = 0, d = 0, f = 0;
  if (std::sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) != 4) return -1;
  const long long want_loc = ((long long)b << 8) | ((long long)d << 3) | f;
  std::string my_gpu;
  int slots = 24;  // hardware queue slots for user compute queues (num_cp_queues; 24 on MI355X) if the node does not say
  const std::string nodes = "/sys/class/kfd/kfd/topology/nodes";
  for (const std::string& n : listDir(nodes)) {
    FILE* fp = std::fopen((nodes + "/" + n + "/properties").c_str(), "r");
    if (!fp) continue;
    char key[64];
    long long val = 0, loc = -1, domain = 0, simd = 0, cpq = 0;
    while (std::fscanf(fp, "%63s %lld", key, &val) == 2) {
      if (!std::strcmp(key, "location_id")) loc = val;
      else if (!std::strcmp(key, "domain")) domain = val;
      else if (!std::strcmp(key, "simd_count")) simd = val;
      else if (!std::strcmp(key, "num_cp_queues")) cpq = val;
    }
    std::fclose(fp);
    if (simd <= 0 || loc != want_loc || domain != (long long)dom) continue;
    if (!readSmallFile(nodes + "/" + n + "/gpu_id", &my_gpu)) my_gpu.clear();
    if (cpq > 0) slots = (int)cpq;
    break;
  }
  if (my_gpu.empty()) return -1;
  const std::string root = "/sys/class/kfd/kfd/proc";
  int compute = 0;
  for (const std::string& pid : listDir(root)) {
    const std::string qd = root + "/" + pid + "/queues";
    for (const std::string& q : listDir(qd)) {
      std::string id, type;
      if (!readSmallFile(qd + "/" + q + "/gpuid", &id) || id != my_gpu) continue;
      if (readSmallFile(qd + "/" + q + "/type", &type) && (type == "0" || type == "compute")) ++compute;
    }
  }
  if (slots_out) *slots_out = slots;
  h->census_compute_queues = compute;
  h->census_queue_slots = slots;
  if (warn && compute > slots && !h->queue_warned) {
    h->queue_warned = true;
    fprintf(stderr, "CUDECOMP:WARN: rank %d: the processes sharing this GPU hold %d compute queues, more than its %d hardware queue "
                    "slots: the kernel driver time-slices ALL of them (expect every GPU operation to take several times longer; rare "
                    "wrong results of kernels of any kind were observed in this regime).  Use fewer processes or streams per GPU.\n",
            h->rank, compute, slots);
  }
  return compute;
}

void peerCheckStatus(cudecompHandle_t h) {
  if (h->peer) h->peer->checkStatus();
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
  PeerContext::Region* r = pc.find(buffer);
  if (!r || !r->peer_base[next]) CD_INVALID_USAGE("buffer to probe must come from cudecompMalloc and be mapped on the next rank");
  char* remote = r->peer_base[next] + (static_cast<char*>(buffer) - r->base);
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

// ------------------------------------------------------------------------------------------------
// one copy into a peer's mapping: copy engines (hipMemcpyAsync; the runtime picks SDMA or a blit kernel from the
// pointers) or the library's own row-copy kernel with write-through stores, one engine per handle
// ------------------------------------------------------------------------------------------------
namespace {
void peerCopy(cudecompHandle_t h, char* dst, const char* src, size_t bytes, hipStream_t stream, int engine = -1) {
  if (bytes == 0) return;
  if (engine < 0) engine = h->peer_copy_engine;
  if (engine == 1 && bytes % 4 == 0) {
    const int es = (bytes % 16 == 0) ? 16 : (bytes % 8 == 0 ? 8 : 4);
    Move3D m;
    m.src_buf = BUF_IN;
    m.dst_buf = BUF_OUT;
    m.extent[0] = (i64)(bytes / es);
    m.ss[0] = m.ds[0] = 1;
    void* bufs[3] = {const_cast<char*>(src), nullptr, nullptr};
    void* dst_base[1] = {dst};
    launchMoves(&m, 1, bufs, es, stream, &h->tuning, nullptr, dst_base);
    return;
  }
  CD_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, stream));
}
}  // namespace

// One-direction copy rate to the next rank of the node through both engines, measured once when the peer transport
// comes up (collective).  Feeds the autotuner's cost model, the choice of the copy engine and bench.py's bisection
// figure ("measure the p2p rate at start-up").
void peerMeasureLink(cudecompHandle_t h) {
  if (h->nranks < 2 || !h->peer || !h->peer->hasBoard() || std::getenv("CUDECOMP_SKIP_LINK_PROBE")) return;
  bool have_dev = true;
  try {
    ensureDevice(h);
  } catch (const Error&) {
    have_dev = false;
  }
  if (h->boot->allreduceOr(!have_dev)) return;  // geometry-only job (no GPU): nothing to measure
  // which physical GPU is each rank on?
  char bus[64] = {0};
  (void)hipDeviceGetPCIBusId(bus, sizeof(bus), h->device);
  std::vector<char> all((size_t)64 * h->nranks);
  h->boot->allgather(bus, all.data(), 64);
  const int next = (h->rank + 1) % h->nranks;
  h->link_crosses_devices = std::strncmp(all.data() + (size_t)64 * next, bus, 64) != 0;
  // Ranks that SHARE a GPU are a test configuration, and one with a known platform hazard (DESIGN.md section 9,
  // profiles/r05_stale_xcd_view.md): a buffer that a process frees and re-allocates while its peers hold IPC mappings of
  // the pooled workspaces is sometimes served stale by ONE XCD, before any library call.  Production -- one process per GPU,
  // memory never recycled between processes -- is not exposed; a program that runs this way hears about it once.
  static bool warned_shared = false;
  if (!h->link_crosses_devices && h->rank == 0 && !warned_shared && !std::getenv("CUDECOMP_WORKSPACE_POOL_MIB")) {
    const char* quiet = std::getenv("CUDECOMP_SHARED_GPU_WARNING");
    if (!quiet || std::strtol(quiet, nullptr, 10) != 0)
      fprintf(stderr, "CUDECOMP:WARN: the ranks of this job share a GPU (a test configuration).  Known platform behaviour: data buffers "
                      "that are freed and re-allocated between calls while peers hold IPC mappings of pooled workspaces can be read stale "
                      "by one XCD; keep data buffers alive across calls or set CUDECOMP_WORKSPACE_POOL_MIB=0 "
                      "(CUDECOMP_SHARED_GPU_WARNING=0 silences this line).\n");
    warned_shared = true;
  }
  const size_t bytes = (size_t)64 << 20;
  char* buf = nullptr;
  bool ok = true;
  std::string err;
  try {
    buf = static_cast<char*>(workspaceAllocRaw(h, 2 * bytes, true));
  } catch (const Error& e) {
    ok = false;
    err = e.what();
  }
  PeerContext::Region* r = (ok && buf) ? h->peer->find(buf) : nullptr;
  if (!r || !r->peer_base[next]) ok = false;
  if (h->boot->allreduceOr(!ok)) {
    if (buf) workspaceFreeRaw(h, buf);
    return;
  }
  double ms[2] = {0, 0};
  try {
    char* remote = r->peer_base[next] + bytes;  // second half of the next rank's buffer
    hipStream_t st = h->peer->copyStream(0);
    hipEvent_t e0, e1;
    CD_CHECK_HIP(hipEventCreate(&e0));
    CD_CHECK_HIP(hipEventCreate(&e1));
    for (int engine = 0; engine < 2; ++engine) {
#ifdef CUDECOMP_TUNING_VARIANTS  // (code-size bisect, scripts/probe/code_size_bisect.sh: probe with one engine only)
      if (const char* only = std::getenv("CUDECOMP_LINK_PROBE_ENGINES"))
        if ((engine == 0) != (std::strcmp(only, "sdma") == 0)) continue;
#endif
      for (int rep = 0; rep < 3; ++rep) {  // rep 0 warms up (page mapping, code load)
        h->boot->barrier();
        CD_CHECK_HIP(hipEventRecord(e0, st));
        peerCopy(h, remote, buf, bytes, st, engine);
        CD_CHECK_HIP(hipEventRecord(e1, st));
        CD_CHECK_HIP(hipStreamSynchronize(st));
        float t = 0;
        CD_CHECK_HIP(hipEventElapsedTime(&t, e0, e1));
        if (rep == 1 || t < ms[engine]) ms[engine] = t;
      }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  } catch (const Error& e) {
    ok = false;
    err = e.what();
    (void)hipGetLastError();
  }
  const bool failed = h->boot->allreduceOr(!ok);
  if (!failed) {
    // every rank pushes at the same time: the slowest direction is what an exchange will see
    const double t_sdma = h->boot->allreduceMax(ms[0]), t_cu = h->boot->allreduceMax(ms[1]);
    h->link_gbps_sdma = t_sdma > 0 ? bytes / (t_sdma * 1e-3) / 1e9 : 0;
    h->link_gbps_cu = t_cu > 0 ? bytes / (t_cu * 1e-3) / 1e9 : 0;
    // Ranks that SHARE a device always copy with kernels: same-device "copies engines" are blit kernels anyway, and the
    // kernel path needs one extra stream where the copy-engine path needs one per peer -- every stream is a hardware
    // queue, and processes that together exceed the device's queue slots get time-sliced.
    if (!h->peer_copy_engine_pinned)
      h->peer_copy_engine = (!h->link_crosses_devices || h->link_gbps_cu > 1.05 * h->link_gbps_sdma) ? 1 : 0;
    if (h->rank == 0 && std::getenv("CUDECOMP_VERBOSE"))
      fprintf(stderr, "CUDECOMP: peer link probe (%s): copy engines %.1f GB/s, compute-unit copy %.1f GB/s per direction; using %s\n",
              h->link_crosses_devices ? "across GPUs" : "ranks share a GPU", h->link_gbps_sdma, h->link_gbps_cu,
              h->peer_copy_engine ? "compute-unit copies" : "copy engines");
  } else if (h->rank == 0) {
    fprintf(stderr, "CUDECOMP:WARN: peer link probe failed (%s)\n", err.c_str());
  }
  workspaceFreeRaw(h, buf);
}

void prepareTransports(cudecompHandle_t h, bool need_rccl, bool need_peer) {
  if (h->nranks == 1 && !h->self_exchange) return;  // every communicator has one member: nothing ever travels
  if (need_rccl && !h->rccl) {
    ensureDevice(h);
    h->rccl = std::make_shared<RcclContext>(h);
  }
  if (need_peer && !h->peer) {
    h->peer = std::make_shared<PeerContext>(h);  // touches the device on first use only
    bool have_dev = true;
    try {
      ensureDevice(h);
    } catch (const Error&) {
      have_dev = false;
    }
    h->peer->agreePoolLimit();
    if (!h->boot->allreduceOr(!have_dev)) h->peer->setupDeviceFlags();  // (geometry-only jobs have no device)
    peerMeasureLink(h);
    // Opt-in (CUDECOMP_QUEUE_CENSUS=1): count the compute queues of ALL processes on this GPU from the driver's tables and
    // warn when they exceed the device's hardware queue slots (ranks sharing a GPU; profiles/r04_tuning.md).  Production --
    // one process per GPU -- never comes near that, and reading the tables of every process is not free.
    if (h->nranks > 1 && have_dev && peerQueueCensusRequested()) (void)peerQueueCensus(h, true);
  }
}

namespace {
void releaseRegionForReal(cudecompHandle_t h, void* ptr) {
  h->peer->unregisterRegion(ptr);
  CD_CHECK_HIP(hipFree(ptr));
}
}  // namespace

void* workspaceAllocRaw(cudecompHandle_t h, size_t bytes, bool peer_capable) {
  void* ptr = nullptr;
  if ((h->nranks > 1 || h->self_exchange) && peer_capable) {
    // one-sided writes address the peer's workspace by offset: make it the same size everywhere
    bytes = (size_t)h->boot->allreduceMaxI64((int64_t)bytes);
    bytes = ipcSafeSize(bytes);
    prepareTransports(h, false, true);
    if (void* pooled = h->peer->takeFromPool(bytes)) return pooled;  // (the same decision on every rank)
    // A new allocation is mapped into every rank and the mappings are VERIFIED (page tags).  The platform can answer
    // hipIpcOpenMemHandle with the mapping of a predecessor of the allocation (freed, re-created at the same address,
    // DESIGN.md section 9); such a buffer is set aside -- so that the next attempt gets another address -- and
    // released once a good one is registered.
    // Buffers whose mappings turned out stale are SET ASIDE -- so that the next attempt gets another address -- and
    // released as soon as a later candidate exists (at most one is alive beside the candidate: peak 2x the request) or
    // on any exit, also an exceptional one.
    struct SetAside {
      std::vector<void*> bufs;
      ~SetAside() {
        for (void* q : bufs) (void)hipFree(q);
      }
      void keepOnlyLatest() {
        while (bufs.size() > 1) {
          (void)hipFree(bufs.front());
          bufs.erase(bufs.begin());
        }
      }
    } set_aside;
    std::string failure;
    bool drained = false;
    for (int attempt = 0; attempt < 4 && !ptr; ++attempt) {
      void* cand = nullptr;
      // A failing hipMalloc is AGREED ON before anybody acts on it (a rank that threw on its own would leave the others
      // in registerRegion's collectives).  First remedy: everything the pool has parked is released -- collectively, the
      // pools are identical on every rank -- and the allocation is tried again.
      hipError_t me = hipMalloc(&cand, bytes);
      if (me != hipSuccess) {
        (void)hipGetLastError();
        cand = nullptr;
      }
      if (h->boot->allreduceOr(cand == nullptr)) {
        if (cand) (void)hipFree(cand);
        if (!drained) {
          drained = true;
          for (void* q : h->peer->drainPool()) releaseRegionForReal(h, q);
          --attempt;  // the retry after draining does not count as a mapping attempt
          continue;
        }
        CD_HIP_ERROR(std::string("cudecompMalloc: hipMalloc of ") + std::to_string(bytes) + " bytes failed on " +
                     (me != hipSuccess ? "this rank" : "another rank") + " (out of memory), also after releasing the workspace pool");
      }
      set_aside.keepOnlyLatest();
      bool stale = false;
      try {
        if (h->peer->registerRegion(cand, bytes, &stale)) ptr = cand;
      } catch (const Error& e) {
        // Agreed on by all ranks (registerRegion fails collectively): export or import refused.  The same platform
        // behaviour shows this way too ("invalid argument" / "invalid device pointer" for a re-created allocation), so
        // another address gets its chance before the library gives up on sharing the workspace.
        failure = e.what();
      }
      if (!ptr) set_aside.bufs.push_back(cand);
    }
    if (!ptr) {
      // not shared: still a valid workspace for the RCCL / MPI transports; an operation that needs the one-sided
      // transport will report the IPC problem itself
      if (failure.empty()) failure = "every attempt to map the workspace into the other ranks ended with a stale mapping";
      ptr = set_aside.bufs.back();
      set_aside.bufs.pop_back();
    } else {
      failure.clear();
    }
    if (!failure.empty()) {
      if (h->rank == 0 && !h->ipc_warned) {
        fprintf(stderr, "CUDECOMP:WARN: workspace could not be shared over IPC (%s); one-sided (NVSHMEM*/default MPI*) "
                        "backends are unavailable with it\n", failure.c_str());
      }
      h->ipc_warned = true;
    }
    return ptr;
  }
  CD_CHECK_HIP(hipMalloc(&ptr, bytes));
  return ptr;
}

void workspaceTrimPool(cudecompHandle_t h) {
  if (!h->peer) return;
  CD_CHECK_HIP(hipDeviceSynchronize());
  for (void* q : h->peer->drainPool()) releaseRegionForReal(h, q);
}

void workspaceFreeRaw(cudecompHandle_t h, void* ptr) {
  if (h->peer) {
    auto* r = h->peer->find(ptr);
    if (r && r->base == ptr) {
      // like hipFree, return only when everything this rank enqueued that may touch the buffer is done
      CD_CHECK_HIP(hipDeviceSynchronize());
      if (h->peer->poolEnabled()) {
        for (void* q : h->peer->park(ptr)) releaseRegionForReal(h, q);  // (the same list on every rank)
        return;
      }
      releaseRegionForReal(h, ptr);
      return;
    }
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
  if (!h->rccl_native_alltoall || ci.nranks != h->nranks) return false;
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

PeerContext& peerOf(cudecompHandle_t h, const cudecompCommInfo& ci) {
  if (!h->peer) CD_INTERNAL_ERROR("peer transport was not created for this grid descriptor");
  if (!h->peer->usable(ci)) CD_PEER_ERROR("the one-sided transport needs all members of the communicator on one node");
  h->peer->ensureDeviceView();
  return *h->peer;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// stream-ordered one-sided exchanges
// ------------------------------------------------------------------------------------------------
// Every call on a communicator has a number (epoch), kept in device memory and advanced by the first kernel of the
// call.  Flags in the shared board carry epochs:
//   ready[r]        last call for which r's stream has reached the exchange: everything r enqueued earlier is done,
//                   so r's receive area (or output pencil) may be overwritten
//   landed[d][s]    last call whose data from s has completely arrived at d
// A sender waits (on its copy stream / in front of its put kernel) for ready[d], moves the data, then raises
// landed[d][me]; a receiver waits for landed[me][s] in front of its unpack.  Nothing blocks the host, and because the
// epoch lives on the device the whole sequence can be captured into a hipGraph and replayed.
// Reference counterpart: the NVSHMEM signal / wait choreography, include/internal/comm_routines.h:122-258 and
// include/internal/cudecomp_kernels.cuh:51-122.

PeerCall peerBegin(cudecompHandle_t h, cudecompCommInfo& ci, bool rendezvous, const void* recv_area, const void* output,
                   bool want_direct, hipStream_t stream) {
  PeerContext& pc = peerOf(h, ci);
  pc.checkStatus();
  PeerCall call;
  call.nranks = ci.nranks;
  if (rendezvous) {
    // (host code only: under stream capture it runs once, at capture time, like every pointer in the graph)
    // Direct put: only into output pencils that come from cudecompMalloc.  Those were mapped into every rank when
    // they were allocated, collectively and once; mapping arbitrary user allocations on the fly proved fragile on
    // this platform (an allocation that is freed and re-created at the same address cannot always be re-imported by
    // a peer that still remembers its predecessor), and a transpose must not depend on that.
    const bool out_ok = want_direct && pc.find(output) != nullptr;
    const void* ptrs[2] = {recv_area, out_ok ? output : nullptr};
    const uint32_t flags[2] = {0, out_ok ? 1u : 0u};
    auto res = pc.rendezvous(ci, ptrs, flags, want_direct ? 2 : 1);
    call.remote_recv = res.remote[0];
    // every member wants it and every member's output is a library region -- all ranks read the same mailboxes, so
    // all of them reach the same verdict
    call.direct = want_direct;
    if (want_direct) {
      call.remote_out = res.remote[1];
      for (int m = 0; m < ci.nranks; ++m)
        if (!res.flags[1][m] || !res.mappable[1][m] || !call.remote_out[m]) call.direct = false;
    }
    if (!call.direct)
      for (int m = 0; m < ci.nranks; ++m)
        if (!call.remote_recv[m])
          CD_PEER_ERROR("the workspace of a peer rank cannot be mapped over IPC (an allocation of 2-4 GiB, 6-8 GiB, ... cannot "
                        "be shared on this platform, nor can memory that is not a device allocation, nor -- reliably -- a "
                        "buffer that was freed and re-created at the same address): obtain it from cudecompMalloc");
  } else {
    call.remote_recv = pc.symmetric(ci, recv_area, "the workspace");
    call.direct = false;
  }
  call.epoch = pc.devEpoch(ci);
  if (pc.debug()) pc.dumpRow(ci, "begin");  // CUDECOMP_DEBUG_PEER=1 (host-synchronous): the row's counters as this call finds them
  FlagList begun;  // "my call has begun": where each member polls it (device flags) / my board cell (board mode)
  if (pc.deviceFlags()) {
    // (my own buffer too: a one-member communicator in the self-exchange test mode waits for itself)
    for (int m = 0; m < ci.nranks; ++m) begun.add(pc.dReadyAt(ci.barrier_slot, ci.global_ranks[m]));
  } else {
    begun.add(pc.dReadyAt(ci.barrier_slot, h->rank));
  }
  launchEpochBegin(call.epoch, begun, stream);
  return call;
}

// One wait for "every receiver's receive area is free" on the CALLER's stream, then an event the copy streams wait
// for.  No copy stream ever parks a spinning wait kernel: the runtime multiplexes a process's streams onto a handful
// of hardware queues (GPU_MAX_HW_QUEUES, default 4), and a wait kernel parked on one of them would hold up the copies
// of every stream that shares its queue.  The pipelined exchange calls this right after the FIRST chunk is packed, so
// a receiver that is late delays nothing that could have run.
void peerReadyGate(cudecompHandle_t h, cudecompCommInfo& ci, const TransposePlan& p, const PeerCall& call, hipStream_t stream) {
  PeerContext& pc = peerOf(h, ci);
  const int P = ci.nranks;
  FlagList ready;
  for (int j = 1; j < P; ++j) ready.add(pc.dReady(ci.barrier_slot, ci.global_ranks[p.schedule_dst[j]]));
  launchWait(call.epoch, ready, pc.dStatus(), h->peer_timeout_s, stream, kFlagBegun);
  CD_CHECK_HIP(hipEventRecord(pc.copyEvent(2 * P), stream));  // "go": receivers ready
}

// All chunks at once: they are packed (caller), the receivers' ready flags are awaited once on `stream`, then the P-1
// chunks travel concurrently -- copy engines: one hipMemcpyAsync per peer on a stream of its own (every link / SDMA
// queue busy); kernel copies: ONE launch on one extra stream whose workgroups serve the destinations round robin.
// `stream` continues when every incoming chunk has landed and every outgoing copy is done (the send area may be reused).
void peerAlltoall(cudecompHandle_t h, cudecompCommInfo& ci, const TransposePlan& p, const ExchangeBuffers& b, int es,
                  const PeerCall& call, hipStream_t stream) {
  PeerContext& pc = peerOf(h, ci);
  const int P = ci.nranks, me = ci.rank;
  // ONE wait for every receiver's ready flag on the caller's stream, then the copies fan out (see peerReadyGate)
  peerReadyGate(h, ci, p, call, stream);
  hipEvent_t go = pc.copyEvent(2 * P);  // chunks packed, receivers ready
  const bool cu = h->peer_copy_engine == 1 && P > 1;
  const int ncopy = cu ? 1 : P - 1;
  if (cu) {
    // compute-unit copies: ONE launch whose workgroups serve the destinations round robin feeds every link for the
    // whole launch, on ONE extra stream (a stream per peer brings nothing here and costs a hardware queue each)
    hipStream_t cs = pc.copyStream(0);
    CD_CHECK_HIP(hipStreamWaitEvent(cs, go, 0));
    std::vector<Move3D> moves;
    std::vector<void*> dst_base;
    FlagList landed;
    void* bufs[3] = {b.send, nullptr, nullptr};
    for (int j = 1; j < P; ++j) {
      const int d = p.schedule_dst[j];
      landed.add(pc.dLanded(ci.barrier_slot, ci.global_ranks[d], h->rank));
      if (p.send_cnt[d] == 0) continue;
      Move3D r;
      r.src_buf = BUF_IN;
      r.src_off = p.send_off[d];
      r.dst_off = p.remote_recv_off[d];
      r.extent[0] = p.send_cnt[d];
      r.ss[0] = r.ds[0] = 1;
      r.peer = d;
      moves.push_back(r);
      dst_base.push_back(call.remote_recv[d]);
    }
    if (!moves.empty()) launchMoves(moves.data(), (int)moves.size(), bufs, es, cs, &h->tuning, nullptr, dst_base.data());
    launchSignal(call.epoch, landed, cs);
    CD_CHECK_HIP(hipEventRecord(pc.copyEvent(0), cs));
  } else {
    for (int j = 1; j < P; ++j) {
      const int d = p.schedule_dst[j];
      hipStream_t cs = pc.copyStream(j);
      CD_CHECK_HIP(hipStreamWaitEvent(cs, go, 0));
      peerCopy(h, call.remote_recv[d] + p.remote_recv_off[d] * es, b.send + p.send_off[d] * es, (size_t)p.send_cnt[d] * es, cs);
      FlagList landed;
      landed.add(pc.dLanded(ci.barrier_slot, ci.global_ranks[d], h->rank));
      launchSignal(call.epoch, landed, cs);
      CD_CHECK_HIP(hipEventRecord(pc.copyEvent(j), cs));
    }
  }
  // my own chunk: a local copy (reference: comm_routines.h:405-410), or through the engine under test
  if (p.send_cnt[me])
    peerCopy(h, b.recv + p.recv_off[me] * es, b.send + p.send_off[me] * es, (size_t)p.send_cnt[me] * es, stream,
             h->self_exchange ? -1 : 0);
  FlagList incoming;
  if (cu) CD_CHECK_HIP(hipStreamWaitEvent(stream, pc.copyEvent(0), 0));
  else
    for (int j = 1; j <= ncopy; ++j) CD_CHECK_HIP(hipStreamWaitEvent(stream, pc.copyEvent(j), 0));
  for (int j = 1; j < P; ++j) incoming.add(pc.dLanded(ci.barrier_slot, h->rank, ci.global_ranks[p.schedule_src[j]]));
  launchWait(call.epoch, incoming, pc.dStatus(), h->peer_timeout_s, stream);
}

// "SM"-style exchange (NVSHMEM_SM enum): the pack kernels write each chunk straight into the receiver's memory over
// xGMI -- one launch feeds all links at once, and the send area, the copy engines and one full HBM pass disappear.
// With call.direct the destination is the receiver's OUTPUT pencil in its final layout and the unpack pass
// disappears as well: one read and one write per element for the whole transpose.  The reference's counterpart is the
// NVSHMEM block-put kernel (include/internal/cudecomp_kernels.cuh:86-122); here it is the ordinary move kernel with
// remote destinations and write-through stores.
void peerPutExchange(cudecompHandle_t h, cudecompCommInfo& ci, const TransposePlan& p, void* const bufs[3], int es,
                     const PeerCall& call, hipStream_t stream) {
  PeerContext& pc = peerOf(h, ci);
  const int P = ci.nranks;
  FlagList ready, landed, incoming;
  for (int m = 0; m < P; ++m) {
    if (m == ci.rank) continue;
    ready.add(pc.dReady(ci.barrier_slot, ci.global_ranks[m]));
    landed.add(pc.dLanded(ci.barrier_slot, ci.global_ranks[m], h->rank));
    incoming.add(pc.dLanded(ci.barrier_slot, h->rank, ci.global_ranks[m]));
  }
  launchWait(call.epoch, ready, pc.dStatus(), h->peer_timeout_s, stream, kFlagBegun);  // every destination may be written

  std::vector<Move3D> moves;
  std::vector<void*> dst_base;
  if (call.direct) {
    for (const Move3D& m : p.direct) {
      moves.push_back(m);
      dst_base.push_back(call.remote_out[m.peer]);
    }
  } else if (!p.pack.empty()) {
    for (const Move3D& m : p.pack) {
      Move3D r = m;
      r.dst_off = p.remote_recv_off[m.peer];
      moves.push_back(r);
      dst_base.push_back(call.remote_recv[m.peer]);
    }
  } else {  // chunks already sit packed in the send buffer (skip-pack plans): plain copies to the peers
    for (int j = 0; j < P; ++j) {
      const int d = p.schedule_dst[j];
      Move3D r;
      r.src_buf = p.send_buf;
      r.src_off = p.send_base + p.send_off[d];
      r.dst_off = p.remote_recv_off[d];
      r.extent[0] = p.send_cnt[d];
      r.ss[0] = r.ds[0] = 1;
      r.peer = d;
      moves.push_back(r);
      dst_base.push_back(call.remote_recv[d]);
    }
  }
  launchMoves(moves.data(), (int)moves.size(), bufs, es, stream, &h->tuning, nullptr, dst_base.data());
  launchSignal(call.epoch, landed, stream);
  launchWait(call.epoch, incoming, pc.dStatus(), h->peer_timeout_s, stream);
}

// ---- two-hop relay (plan.h RelayPlan) -----------------------------------------------------------
bool peerRelayApplies(cudecompHandle_t h, cudecompGridDesc_t gd, const TransposePlan& plan, cudecompTransposeCommBackend_t backend,
                      bool inplace) {
  (void)inplace;
  if (!h->two_hop_relay || backend != CUDECOMP_TRANSPOSE_COMM_NVSHMEM || !plan.exchange || h->self_exchange) return false;
  if (!relayWorthwhile(plan.nranks, h->nranks)) return false;
  // flags of a communicator of all ranks, all of them on this node and on the shared board
  return h->peer && gd->world.nranks == h->nranks && h->nranks <= kMaxFlags && h->peer->usable(gd->world);
}

bool peerRelayEnsureRegion(cudecompHandle_t h, const RelayPlan& rp, int es) {
  // the same number on every rank (it is derived from the whole decomposition), so this is collective by construction
  const size_t need = (size_t)rp.relayElements() * es;
  if (h->relay_buf && h->relay_bytes >= need) return true;
  if (h->relay_buf) {
    // Regrowing: my previous relayed call (possibly on another stream) and my peers' previous scatters into the region must be
    // over before it goes away.  Mine: the event every relayed call ends with (never recorded if that call threw before
    // its end -- a deliberate no-op then); the peers': a barrier after everybody has waited for its own.
    if (h->relay_last_call) CD_CHECK_HIP(hipEventSynchronize(h->relay_last_call));
    h->boot->barrier();
    workspaceFreeRaw(h, h->relay_buf);
  }
  h->relay_buf = nullptr;
  h->relay_bytes = 0;
  void* p = workspaceAllocRaw(h, need, true);
  PeerContext::Region* r = h->peer->find(p);
  bool ok = r != nullptr;
  for (int g = 0; ok && g < h->nranks; ++g) ok = r->peer_base[g] != nullptr;
  if (h->boot->allreduceOr(!ok)) {
    // agreed by all ranks: the relay is off for this handle from here on (no retry per call), exchanges go direct
    workspaceFreeRaw(h, p);
    h->two_hop_relay = false;
    if (h->rank == 0)
      fprintf(stderr, "CUDECOMP:WARN: the relay region of the two-hop exchange could not be mapped into every rank of the node; "
                      "CUDECOMP_TWO_HOP_RELAY is off for this handle.\n");
    return false;
  }
  h->relay_buf = p;
  h->relay_bytes = need;
  return true;
}

void peerRelayAlltoall(cudecompHandle_t h, cudecompCommInfo& world, const TransposePlan& p, const RelayPlan& rp,
                       const ExchangeBuffers& b, int es, const PeerCall& call, hipStream_t stream) {
  PeerContext& pc = peerOf(h, world);
  const int n = world.nranks, me = h->rank, slot = world.barrier_slot;
  PeerContext::Region* rr = pc.find(h->relay_buf);
  if (!rr) CD_INTERNAL_ERROR("two-hop relay without its relay region");
  FlagList ready, landed, incoming;
  for (int q = 0; q < n; ++q) {
    if (q == me) continue;
    ready.add(pc.dReady(slot, q));
    landed.add(pc.dLanded(slot, q, me));
    incoming.add(pc.dLanded(slot, me, q));
  }
  // every rank has begun this call: its relay slots (forwarded in its previous call, earlier on its stream) and its
  // receive area are free
  launchWait(call.epoch, ready, pc.dStatus(), h->peer_timeout_s, stream, kFlagBegun);
  auto run = [&](const std::vector<RelayMove>& list, char* src_base) {
    std::vector<Move3D> moves;
    std::vector<void*> dst_base;
    for (const RelayMove& m : list) {
      Move3D r;
      r.src_buf = BUF_IN;
      r.src_off = m.src_off;
      r.dst_off = m.dst_off;
      r.extent[0] = m.count;
      r.ss[0] = r.ds[0] = 1;
      r.peer = m.dst_rank;
      moves.push_back(r);
      dst_base.push_back(m.to_relay ? rr->peer_base[m.dst_rank] : call.remote_recv[m.dst_rank]);
    }
    void* bufs[3] = {src_base, nullptr, nullptr};
    if (!moves.empty()) launchMoves(moves.data(), (int)moves.size(), bufs, es, stream, &h->tuning, nullptr, dst_base.data());
  };
  // step 1: my slices to the relays (and the two direct slices of every chunk to its destination)
  run(rp.scatter, b.send);
  launchSignal(call.epoch, landed, stream, 1);
  // my own chunk never leaves the device (reference: comm_routines.h:405-410)
  if (p.send_cnt[p.comm_rank])
    peerCopy(h, b.recv + p.recv_off[p.comm_rank] * es, b.send + p.send_off[p.comm_rank] * es, (size_t)p.send_cnt[p.comm_rank] * es, stream, 0);
  // step 2: what the others parked in my relay region goes on to its destinations
  launchWait(call.epoch, incoming, pc.dStatus(), h->peer_timeout_s, stream, 1);
  run(rp.forward, static_cast<char*>(h->relay_buf));
  launchSignal(call.epoch, landed, stream, kFlagDone);
  // everybody's direct slices (done before their step-1 signal) and forwards have reached my receive area
  launchWait(call.epoch, incoming, pc.dStatus(), h->peer_timeout_s, stream, kFlagDone);
}

bool peerPipelineAvailable(cudecompHandle_t h, const cudecompCommInfo& ci) { return h->peer && h->peer->usable(ci); }

// CUDECOMP_DEBUG_VERIFY_EXCHANGE=1 (host-synchronous debugging aid): after a one-sided exchange,
// tell LATE data from data that NEVER ARRIVED HERE.  Every member checksums each chunk it received as soon as its own
// stream has drained (the landed flags said "complete"), again after a barrier of the communicator (now nothing is in
// flight anywhere), and compares with the checksum the SENDER takes of the chunk it sent.
//   first != second            the flag overtook the data (late arrival)
//   second != sender's         the data did not arrive in this buffer at all (misdirected / lost), or was damaged
void peerVerifyExchange(cudecompHandle_t h, cudecompCommInfo& ci, const TransposePlan& p, const ExchangeBuffers& b, int es,
                        bool sender_side_valid, const char* what, hipStream_t stream) {
  PeerContext& pc = peerOf(h, ci);
  const int P = ci.nranks, me = ci.rank;
  CD_CHECK_HIP(hipStreamSynchronize(stream));
  unsigned long long* dev = nullptr;
  CD_CHECK_HIP(hipMalloc(reinterpret_cast<void**>(&dev), sizeof(unsigned long long) * 2 * 3 * P));
  CD_CHECK_HIP(hipMemset(dev, 0, sizeof(unsigned long long) * 2 * 3 * P));
  auto sums = [&](int pass) {  // pass 0 / 1: incoming chunks; pass 2: outgoing chunks
    for (int m = 0; m < P; ++m) {
      const char* ptr = pass < 2 ? b.recv + p.recv_off[m] * es : b.send + p.send_off[m] * es;
      const size_t bytes = (size_t)(pass < 2 ? p.recv_cnt[m] : p.send_cnt[m]) * es;
      launchChecksum(ptr, bytes, dev + 2 * (pass * P + m), stream);
    }
  };
  sums(0);
  if (sender_side_valid) sums(2);
  CD_CHECK_HIP(hipStreamSynchronize(stream));
  pc.barrier(ci);
  sums(1);
  CD_CHECK_HIP(hipStreamSynchronize(stream));
  std::vector<unsigned long long> host((size_t)2 * 3 * P);
  CD_CHECK_HIP(hipMemcpy(host.data(), dev, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  (void)hipFree(dev);
  // everybody's "what I sent to member d": [member][d][2]
  std::vector<unsigned long long> sent((size_t)2 * P * P);
  ci.boot->allgather(host.data() + (size_t)2 * 2 * P, sent.data(), sizeof(unsigned long long) * 2 * P);
  for (int s = 0; s < P; ++s) {
    if (p.recv_cnt[s] == 0) continue;
    const unsigned long long* r1 = &host[2 * (0 * P + s)];
    const unsigned long long* r2 = &host[2 * (1 * P + s)];
    const unsigned long long* ex = &sent[2 * ((size_t)s * P + me)];
    const bool late = r1[0] != r2[0] || r1[1] != r2[1];
    const bool wrong = sender_side_valid && (r2[0] != ex[0] || r2[1] != ex[1]);
    if (late || wrong)
      fprintf(stderr, "CUDECOMP:VERIFY rank %d %s: chunk from member %d (rank %d, %lld bytes at recv offset %lld): %s%s  "
                      "[at flag %016llx/%016llx, after barrier %016llx/%016llx, sender %016llx/%016llx]\n",
              h->rank, what, s, ci.global_ranks[s], (long long)p.recv_cnt[s] * es, (long long)p.recv_off[s] * es,
              late ? "LATE (changed after the landed flag) " : "", wrong ? "NOT WHAT WAS SENT even after everyone drained" : "",
              r1[0], r1[1], r2[0], r2[1], ex[0], ex[1]);
  }
  (void)me;
}

// Staged pipeline of the one-sided transport (NVSHMEM_PL / MPI_P2P_PL enums).
//
// The reference pipelines PER PEER (comm_routines.h:427-631, transpose.h:470-513, 683-744): one peer's chunk is packed,
// sent and unpacked after the other.  On a full xGMI mesh that is the wrong unit: every peer has a link of its own, so a
// per-peer pipeline keeps one link busy at a time while the others wait for their turn, and each per-peer unpack launch
// writes one slice of every destination row (a 2-KiB piece of every 8-KiB row leaves most DRAM channels idle:
// 0.54-0.57 of the HBM peak, profiles/r02_local_phases.json).  Here the pipeline runs over STAGES instead: all chunks
// are cut into K ranges along their slowest wire dim (contiguous sub-chunks in the send and receive areas), and stage k
//   pack(k): one launch, all destinations            [caller's stream, event per stage]
//   send(k): sub-chunk k to EVERY peer at once        [copy stream(s): all links busy in every stage] + landed = k
//   unpack(k): one launch, all sources, whole rows   [caller's stream, behind a wait for stage k of every source]
// overlaps pack(k+1) with send(k) and send(k+1) with unpack(k).  The receivers' "ready" is awaited once, on the
// caller's stream behind the first pack; no stream ever parks a spinning kernel in front of work that could run.
void peerStagedExchange(cudecompHandle_t h, cudecompGridDesc_t gd, cudecompCommInfo& ci, const TransposePlan& plan,
                        void* const bufs[3], const ExchangeBuffers& b, int es, const PeerCall& call, hipStream_t stream) {
  (void)gd;
  PeerContext& pc = peerOf(h, ci);
  const int P = plan.nranks, me = plan.comm_rank, ax = plan.stage_axis;
  const int K = stageCount(plan, h->pipeline_stages, es, h->pipeline_min_stage_bytes);
  auto stepOf = [&](int k) { return k == K - 1 ? kFlagDone : k + 1; };
  // events: [0, P) last copy of each copy stream, 2P "go", then one "packed" event per stage
  auto packedEvent = [&](int k) { return pc.copyEvent(2 * P + 2 + k); };
  const bool cu = h->peer_copy_engine == 1;
  const int ncopy = (P > 1) ? (cu ? 1 : P - 1) : 0;  // compute-unit copies: one launch feeds all links; SDMA: a stream per peer

  FlagList ready, landed_all, incoming;
  for (int j = 1; j < P; ++j) {
    ready.add(pc.dReady(ci.barrier_slot, ci.global_ranks[plan.schedule_dst[j]]));
    landed_all.add(pc.dLanded(ci.barrier_slot, ci.global_ranks[plan.schedule_dst[j]], h->rank));
    incoming.add(pc.dLanded(ci.barrier_slot, h->rank, ci.global_ranks[plan.schedule_src[j]]));
  }
  hipEvent_t go = pc.copyEvent(2 * P);

  std::vector<Move3D> moves;
  for (int k = 0; k < K; ++k) {
    // ---- pack stage k (all destinations, self included) on the caller's stream
    if (!plan.pack.empty()) {
      moves.clear();
      for (const Move3D& m : plan.pack) moves.push_back(stageOfMove(m, ax, k, K));
      launchMoves(moves.data(), (int)moves.size(), bufs, es, stream, &h->tuning);
    }
    CD_CHECK_HIP(hipEventRecord(packedEvent(k), stream));
    if (k == 0) {  // receivers ready? (once, behind the first pack)
      launchWait(call.epoch, ready, pc.dStatus(), h->peer_timeout_s, stream, kFlagBegun);
      CD_CHECK_HIP(hipEventRecord(go, stream));
    }
    // ---- send stage k to every peer
    if (cu && P > 1) {
      hipStream_t cs = pc.copyStream(0);
      CD_CHECK_HIP(hipStreamWaitEvent(cs, packedEvent(k), 0));
      if (k == 0) CD_CHECK_HIP(hipStreamWaitEvent(cs, go, 0));
      moves.clear();
      std::vector<void*> dst_base;
      for (int j = 1; j < P; ++j) {
        const int d = plan.schedule_dst[j];
        const i64 n = plan.send_n[d], per = plan.send_cnt[d] / n, lo = n * k / K, hi = n * (k + 1) / K;
        if (hi == lo) continue;
        Move3D r;
        r.src_buf = plan.send_buf;
        r.src_off = plan.send_base + plan.send_off[d] + lo * per;
        r.dst_off = plan.remote_recv_off[d] + lo * per;
        r.extent[0] = (hi - lo) * per;
        r.ss[0] = r.ds[0] = 1;
        r.peer = d;
        moves.push_back(r);
        dst_base.push_back(call.remote_recv[d]);
      }
      if (!moves.empty()) launchMoves(moves.data(), (int)moves.size(), bufs, es, cs, &h->tuning, nullptr, dst_base.data());
      launchSignal(call.epoch, landed_all, cs, stepOf(k));
      if (k == K - 1) CD_CHECK_HIP(hipEventRecord(pc.copyEvent(0), cs));
    } else {
      for (int j = 1; j < P; ++j) {
        const int d = plan.schedule_dst[j];
        hipStream_t cs = pc.copyStream(j);
        CD_CHECK_HIP(hipStreamWaitEvent(cs, packedEvent(k), 0));
        if (k == 0) CD_CHECK_HIP(hipStreamWaitEvent(cs, go, 0));
        const i64 n = plan.send_n[d], per = plan.send_cnt[d] / n, lo = n * k / K, hi = n * (k + 1) / K;
        peerCopy(h, call.remote_recv[d] + (plan.remote_recv_off[d] + lo * per) * es, b.send + (plan.send_off[d] + lo * per) * es,
                 (size_t)((hi - lo) * per) * es, cs);
        FlagList landed;
        landed.add(pc.dLanded(ci.barrier_slot, ci.global_ranks[d], h->rank));
        launchSignal(call.epoch, landed, cs, stepOf(k));
        if (k == K - 1) CD_CHECK_HIP(hipEventRecord(pc.copyEvent(j), cs));
      }
    }
  }
  // my own chunk: a local copy (reference: comm_routines.h:405-410), or through the engine under test
  if (plan.send_cnt[me])
    peerCopy(h, b.recv + plan.recv_off[me] * es, b.send + plan.send_off[me] * es, (size_t)plan.send_cnt[me] * es, stream,
             h->self_exchange ? -1 : 0);
  // ---- unpack stage by stage: every source's sub-chunk k has landed -> one launch that writes whole rows
  for (int k = 0; k < K; ++k) {
    launchWait(call.epoch, incoming, pc.dStatus(), h->peer_timeout_s, stream, stepOf(k));
    moves.clear();
    for (const Move3D& m : plan.unpack) moves.push_back(stageOfMove(m, ax, k, K));
    if (!moves.empty()) launchMoves(moves.data(), (int)moves.size(), bufs, es, stream, &h->tuning);
  }
  if (cu && P > 1) CD_CHECK_HIP(hipStreamWaitEvent(stream, pc.copyEvent(0), 0));  // send area reusable
  else
    for (int j = 1; j <= ncopy; ++j) CD_CHECK_HIP(hipStreamWaitEvent(stream, pc.copyEvent(j), 0));
}

void alltoallExchange(cudecompHandle_t h, cudecompGridDesc_t, cudecompCommInfo& ci, const TransposePlan& plan,
                      const ExchangeBuffers& b, int es, cudecompTransposeCommBackend_t backend, const PeerCall* call,
                      hipStream_t stream) {
  if (usesRccl(backend)) return rcclAlltoall(h, ci, plan, b, es, stream);
#ifdef CUDECOMP_WITH_MPI
  if (transposeBackendIsMpi(backend) && mpiTransportAvailable(ci)) return mpiAlltoall(h, ci, plan, b, es, stream);
#endif
  if (!call) CD_INTERNAL_ERROR("one-sided exchange without its call state");
  peerAlltoall(h, ci, plan, b, es, *call, stream);
}

void alltoallExchangePeers(cudecompHandle_t h, cudecompGridDesc_t gd, cudecompCommInfo& ci, const TransposePlan& plan,
                           const ExchangeBuffers& b, int es, cudecompTransposeCommBackend_t backend,
                           const std::vector<int>& src_members, const std::vector<int>& dst_members,
                           hipStream_t stream) {
  if (src_members.empty()) return;
  const int me = ci.rank;
  if (!usesRccl(backend)) {
#ifdef CUDECOMP_WITH_MPI
    // MPI synchronises on the host, so there is nothing to gain from splitting the exchange by peer: run all of
    // it when the schedule reaches the self step, which comes first.
    if (src_members[0] != me) return;
    if (transposeBackendIsMpi(backend) && mpiTransportAvailable(ci)) return mpiAlltoall(h, ci, plan, b, es, stream);
#endif
    CD_INTERNAL_ERROR("per-peer exchange requested for a transport that has its own pipeline");
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
    if (s == me && !h->self_exchange) {
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
    if (src_members[i] == me && !h->self_exchange) continue;
    const int d = dst_members[i];
    CD_CHECK_HIP(hipEventRecord(gd->events[d], side));
    CD_CHECK_HIP(hipStreamWaitEvent(stream, gd->events[d], 0));  // chunk has arrived: unpack may start
  }
}

// ================================================================================================
// halo exchange
// ================================================================================================
namespace {

void ensureSideStream(cudecompHandle_t h) {
  if (!h->streams.empty()) return;
  int lo = 0, hi = 0;
  CD_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t s;
  CD_CHECK_HIP(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
  h->streams.push_back(s);
}

// resolves where my two faces land: remote[i] = address of neighbour i's halo slot (1 - i) as written by me
PeerCall haloBegin(cudecompHandle_t h, cudecompGridDesc_t gd, const HaloExchange& x, cudecompHaloCommBackend_t backend,
                   hipStream_t stream) {
  cudecompCommInfo& ci = gd->comm(x.comm_axis);
  // the NVSHMEM enums require a symmetric workspace from cudecompMalloc (as in the reference); the MPI enums take any
  // device buffer and pay a host rendezvous per call (the reference's MPI backends block the host as well)
  return peerBegin(h, ci, !haloBackendIsPeer(backend), x.recv, nullptr, false, stream);
}

int memberOf(const cudecompCommInfo& ci, int global_rank) {
  for (int m = 0; m < ci.nranks; ++m)
    if (ci.global_ranks[m] == global_rank) return m;
  CD_INTERNAL_ERROR("halo neighbour is not a member of the exchanging communicator");
}

// faces -> neighbours' halo slots.  packed[i] (may be null: face data ready on `stream` already) gates face i.
// Slot i of mine is filled by neighbour i, who raises landed[me][i]; I fill slot 1-i of neighbour i.
// one wait for both neighbours' "my halo slots are free" on the caller's stream; returns the event the copy streams
// wait for (no copy stream parks a wait kernel, see peerReadyGate)
hipEvent_t haloReadyGate(cudecompHandle_t h, cudecompGridDesc_t gd, const HaloExchange& x, const PeerCall& call, hipStream_t stream) {
  cudecompCommInfo& ci = gd->comm(x.comm_axis);
  PeerContext& pc = peerOf(h, ci);
  FlagList both;
  for (int i = 0; i < 2; ++i)
    if (x.neighbor[i] != -1 && (i == 0 || x.neighbor[1] != x.neighbor[0])) both.add(pc.dReady(ci.barrier_slot, x.neighbor[i]));
  launchWait(call.epoch, both, pc.dStatus(), h->peer_timeout_s, stream, kFlagBegun);
  hipEvent_t ready_ev = pc.copyEvent(2 * ci.nranks + 1);
  CD_CHECK_HIP(hipEventRecord(ready_ev, stream));
  return ready_ev;
}

void peerHaloExchange(cudecompHandle_t h, cudecompGridDesc_t gd, const HaloExchange& x, const PeerCall& call,
                      hipEvent_t* packed, hipEvent_t ready_ev, hipStream_t stream) {
  cudecompCommInfo& ci = gd->comm(x.comm_axis);
  PeerContext& pc = peerOf(h, ci);
  // plain sequence (packed == nullptr): the gate comes after both packs, then the two copies fan out;
  // overlapped sequence: the caller placed it behind the first pack, each face also waits for its own pack
  if (!ready_ev) ready_ev = haloReadyGate(h, gd, x, call, stream);
  FlagList incoming;
  const bool one_stream = h->peer_copy_engine == 1;  // kernel copies: both faces share one extra stream (see peerAlltoall)
  for (int i = 0; i < 2; ++i) {
    if (x.neighbor[i] == -1) continue;
    const int m = memberOf(ci, x.neighbor[i]);
    hipStream_t cs = pc.copyStream(one_stream ? 0 : i);
    if (packed) CD_CHECK_HIP(hipStreamWaitEvent(cs, packed[i], 0));
    CD_CHECK_HIP(hipStreamWaitEvent(cs, ready_ev, 0));
    FlagList landed;
    landed.add(pc.dLanded(ci.barrier_slot, x.neighbor[i], 1 - i));
    peerCopy(h, call.remote_recv[m] + x.remote_off[i], x.send + x.send_off[i], (size_t)x.bytes, cs);
    launchSignal(call.epoch, landed, cs);
    CD_CHECK_HIP(hipEventRecord(pc.copyEvent(i), cs));
    incoming.add(pc.dLanded(ci.barrier_slot, h->rank, i));
  }
  for (int i = 0; i < 2; ++i)
    if (x.neighbor[i] != -1) CD_CHECK_HIP(hipStreamWaitEvent(stream, pc.copyEvent(i), 0));
  launchWait(call.epoch, incoming, pc.dStatus(), h->peer_timeout_s, stream);
}

}  // namespace

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
  // The one-sided transport only runs packed plans (runHalo: force_packed), whose faces were packed on `stream`
  // before this call; the epoch starts here, i.e. "my halo slots are free" is published after the packs -- harmless,
  // the overlapped variant below publishes it before them.
  const PeerCall call = haloBegin(h, gd, x, backend, stream);
  peerHaloExchange(h, gd, x, call, nullptr, nullptr, stream);
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
    ensureSideStream(h);
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

  // one-sided transport: "my halo slots are free" goes out before the packs, each face travels as soon as ITS pack is
  // done (on its own copy stream) while the other is still being packed; the unpacks follow the landed flags.  Small
  // faces take the plain sequence (both packs in one launch, both unpacks in one launch): there is nothing to overlap
  // and every launch saved counts at that size.
  if (x.bytes < kHaloOverlapMinBytes && !h->halo_overlap_force) return false;
  const PeerCall call = haloBegin(h, gd, x, backend, stream);
  hipEvent_t ready_ev = nullptr;
  for (int i = 0; i < 2; ++i) {
    if (const Move3D* m = moveOf(plan.pre, i)) launchMoves(m, 1, bufs, es, stream, &h->tuning);
    CD_CHECK_HIP(hipEventRecord(packed[i], stream));
    if (i == 0) ready_ev = haloReadyGate(h, gd, x, call, stream);  // behind the first pack: the second overlaps the wait's tail
  }
  peerHaloExchange(h, gd, x, call, packed, ready_ev, stream);
  launchMoves(plan.post.data(), (int)plan.post.size(), bufs, es, stream, &h->tuning);
  return true;
}

}  // namespace cudecomp
