// internal.h -- the two opaque objects of the C API and the pieces they own.
#pragma once
#include <array>
#include <map>
#include <set>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "bootstrap.h"
#include "cudecomp.h"
#include "decomp.h"
#include "kernels.h"
#include "plan.h"

namespace cudecomp {
enum ExecPath { PATH_LOCAL = 0, PATH_RCCL, PATH_MPI, PATH_PEER_BARRIER, PATH_PEER_FUSED, PATH_PEER_PIPELINED };
class RcclContext;  // transport.cc
class PeerContext;  // transport.cc
}  // namespace cudecomp

// One row or column of the process grid as seen by this rank.
struct cudecompCommInfo {
  int rank = 0, nranks = 0;
  int ngroups = 1, npergroup = 1;           // fast-interconnect groups (hosts) inside the communicator
  std::vector<int> global_ranks;            // member -> rank in the handle's communicator
  std::unique_ptr<cudecomp::Bootstrap> boot;  // control-plane communicator of the members
  // Row of this communicator in the node's shared-memory board (peer transport): host barrier cells, device-visible
  // ready / landed flags and the per-call mailboxes.  Counters only ever grow: a communicator that takes over a row
  // starts from the highest value any of its members has seen there (agreed when it is built), so nothing is reset
  // and a late write that belongs to a destroyed communicator can never be mistaken for a new one.
  int barrier_slot = -1;
  uint64_t barrier_epoch = 0;                 // host barrier
  uint64_t mail_seq = 0;                      // per-call host rendezvous (buffer descriptors)
  uint64_t epoch_base = 0;                    // first device epoch of this communicator is epoch_base + 1
  unsigned long long* dev_epoch = nullptr;    // the call counter of the stream-ordered exchanges, in DEVICE memory
  cudecompHandle_t owner = nullptr;           // for releasing the row
  void release();  // gives the row back; the object can be filled again
  ~cudecompCommInfo();
  cudecompCommInfo() = default;
  cudecompCommInfo(const cudecompCommInfo&) = delete;
  cudecompCommInfo& operator=(const cudecompCommInfo&) = delete;
};

struct cudecompHandle {
  bool initialized = false;
  int rank = 0, nranks = 1;
  int local_rank = 0, local_nranks = 1;
  std::unique_ptr<cudecomp::Bootstrap> boot;
  std::vector<std::string> hostnames;  // by rank
  std::vector<int> rank_to_local_rank;

  // device (probed lazily so that geometry queries also work on a machine without a GPU)
  bool device_probed = false;
  int device = -1;
  int num_cus = 0;
  std::vector<hipStream_t> streams;  // side streams for pipelined transports

  std::shared_ptr<cudecomp::RcclContext> rccl;  // RCCL communicator over all ranks (created on demand)
  std::shared_ptr<cudecomp::PeerContext> peer;  // xGMI / IPC peer-mapping registry (created on demand)

  // environment switches (same names as the reference, docs/env_vars.rst)
  bool graphs_enable = false;
  bool performance_report_enable = false;
  int performance_report_detail = 0;          // 0 summary, 1 + samples of rank 0, 2 + samples of all ranks
  int performance_report_samples = 20;        // ring size per configuration
  int performance_report_warmup_samples = 3;  // first calls of a configuration that are not sampled
  std::string performance_report_write_dir;   // CSV output directory ("" = none)
  bool col_major_env_warned = false;
  bool ipc_warned = false;
  long long fuse_small_bytes = 1ll << 20;  // CUDECOMP_FUSE_SMALL_EXCHANGES_KIB: NVSHMEM-enum exchanges of pencils up to this size
                                           // run as a fused pack + put on one stream (0 = never)
  bool two_hop_relay = false;         // CUDECOMP_TWO_HOP_RELAY=1: low-fan-out exchanges of the NVSHMEM enum travel through all ranks of the node
  void* relay_buf = nullptr;          // my relay region (a library region mapped into every rank), grown on demand
  size_t relay_bytes = 0;
  hipEvent_t relay_last_call = nullptr;  // end of the handle's last relayed transpose (relayed calls run one after the other)
  int census_compute_queues = -1, census_queue_slots = 0;  // last census of the device's compute queues (transport.cc)
  bool queue_warned = false;          // the "hardware queues oversubscribed" note (ranks sharing a device) was printed
  bool halo_overlap_disable = false;  // CUDECOMP_DISABLE_HALO_OVERLAP=1
  bool halo_overlap_force = false;    // CUDECOMP_FORCE_HALO_OVERLAP=1: also for faces below the size threshold (tests)
  bool self_exchange = false;         // CUDECOMP_TEST_SELF_EXCHANGE=1: one-member communicators exchange with themselves
                                      // through the selected transport (drives real RCCL / the peer transport on one GPU)
  bool rccl_native_alltoall = true;   // CUDECOMP_RCCL_NATIVE_ALLTOALL=0: grouped send/recv even where ncclAllToAll applies
  bool debug_verify_exchange = false;  // CUDECOMP_DEBUG_VERIFY_EXCHANGE=1: checksum what one-sided exchanges delivered (host-synchronous)
  bool direct_put = true;             // CUDECOMP_DISABLE_DIRECT_PUT=1: NVSHMEM_SM always lands in the receive area + unpack
  bool inplace_rotation = true;       // CUDECOMP_DISABLE_INPLACE_ROTATION=1: single-rank in-place transposes always stage through the workspace
  long long pipeline_min_stage_bytes = 8ll << 20;  // CUDECOMP_PIPELINE_MIN_STAGE_MIB: no stage smaller than this
  int pipeline_stages = 4;            // CUDECOMP_PIPELINE_STAGES: stages of the one-sided pipelined exchange (1..15)
  double peer_timeout_s = 120.0;      // CUDECOMP_PEER_TIMEOUT: how long a rank waits for a peer (host rendezvous, device flags)
  int peer_copy_engine = 0;           // 0 = copy engines (hipMemcpyAsync), 1 = compute-unit copy kernel; CUDECOMP_PEER_COPY_ENGINE
  bool peer_copy_engine_pinned = false;
  // one-direction copy rate to the next rank measured when the peer transport came up (GB/s; 0 = not measured)
  double link_gbps_sdma = 0.0, link_gbps_cu = 0.0;
  bool link_crosses_devices = false;  // the ranks of this node sit on different GPUs

  cudecomp::KernelTuning tuning;
  // rows of the shared-memory board: handed out lowest-free-first, identically on every rank (communicators are
  // created and destroyed collectively, in the same order everywhere), returned when the communicator goes away
  std::vector<bool> slot_used;
  std::vector<uint64_t> slot_high;  // highest counter value this rank has used in each row
  int acquireSlot();
  void releaseSlot(int slot, uint64_t high);
  // bumped whenever a library region is mapped or unmapped (cudecompMalloc / cudecompFree): captured whole-operation
  // graphs hold the peers' IPC addresses of such regions and must not outlive them
  uint64_t region_generation = 0;

  ~cudecompHandle();
};

struct cudecompGridDesc {
  bool initialized = false;
  cudecompHandle_t handle = nullptr;
  cudecompGridDescConfig_t config{};
  bool gdims_dist_set = false;
  bool mem_order_set = false;

  cudecomp::GridShape shape;
  std::array<int32_t, 2> pidx{};
  cudecompCommInfo row, col;
  cudecompCommInfo world;  // all ranks of the handle (only built for the two-hop relay, CUDECOMP_TWO_HOP_RELAY=1)
  std::map<std::tuple<int, std::array<int32_t, 12>, bool, bool, bool>, cudecomp::RelayPlan> relay_plans;
  int64_t relayed = 0;     // transposes whose exchange went through the two-hop relay

  std::vector<hipEvent_t> events;  // one per communicator member, for per-peer pipelining


  // plan caches (key: op, halos, padding, in-place flag, transport traits)
  using TransposeKey = std::tuple<int, std::array<int32_t, 12>, bool, bool, bool>;
  std::map<TransposeKey, cudecomp::TransposePlan> transpose_plans;
  using HaloKey = std::tuple<int, int, std::array<int32_t, 6>, std::array<bool, 3>, bool>;
  std::map<HaloKey, cudecomp::HaloPlan> halo_plans;

  // CUDECOMP_ENABLE_CUDA_GRAPHS=1: the per-peer pack loop of the pipelined backends (one kernel + one event
  // record per destination) is captured once per (plan, buffers, element size) and replayed as one graph launch
  // (reference: graphCache, src/graph.cc, include/internal/transpose.h:458-519)
  using PackGraphKey = std::tuple<TransposeKey, const void*, const void*, const void*, int>;
  std::map<PackGraphKey, hipGraphExec_t> pack_graphs;
  // ... and, for the one-sided backends that need no host communication per call, the WHOLE operation (transpose.cc:
  // runAsGraph); op_graph_seen: (plan, buffers) that ran eagerly once and are captured on their next call
  std::map<PackGraphKey, hipGraphExec_t> op_graphs;
  std::set<PackGraphKey> op_graph_seen;
  uint64_t op_graph_generation = 0;  // handle->region_generation the graphs above were captured under
  hipStream_t graph_stream = nullptr;
  int64_t graph_launches = 0;
  int64_t direct_puts = 0;  // NVSHMEM_SM transposes that wrote straight into the peers' output pencils
  int64_t rotations = 0;    // single-rank in-place transposes executed as ONE in-place rotation kernel (kernels_rotate.hip)
  std::array<int64_t, 6> path_count{};  // transposes executed per path (cudecomp::ExecPath), for cudecompExtGetCounters
  bool graphs_failed = false;  // the runtime refused a capture: stay on plain launches

  // performance samples (CUDECOMP_ENABLE_PERFORMANCE_REPORT=1), one ring of event quadruples
  // [start, local phase 1 done, exchange done, end] per distinct call configuration, recorded on the caller's stream
  struct PerfSample {
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool used = false;
  };
  struct PerfCollection {
    std::vector<PerfSample> ring;
    int64_t calls = 0;       // calls seen, warm-up included
    int64_t wire_bytes = 0;  // bytes the exchange moved in the last call (reference: transpose.h:316, halo.h:236)
  };
  // (op, dtype, [in halo, out halo, in pad, out pad], in place)
  using TransposePerfKey = std::tuple<int, int, std::array<int32_t, 12>, bool>;
  // (axis, dim, dtype, halo, periods, padding)
  using HaloPerfKey = std::tuple<int, int, int, std::array<int32_t, 3>, std::array<bool, 3>, std::array<int32_t, 3>>;
  std::map<TransposePerfKey, PerfCollection> perf_transpose;
  std::map<HaloPerfKey, PerfCollection> perf_halo;

  cudecompCommInfo& comm(cudecomp::CommAxis a) { return a == cudecomp::COMM_ROW ? row : col; }
  ~cudecompGridDesc();
};

namespace cudecomp {

// one direction of one xGMI link between two MI355X (a link is quoted at 153.6 GB/s counting both directions); the cost
// model and bench.py use the rate MEASURED at start-up when the ranks sit on different GPUs, this figure otherwise
constexpr double kNominalLinkGBpsPerDirection = 76.8;

void ensureDevice(cudecompHandle_t handle);  // throws if no HIP device is usable
void buildCommInfo(cudecompHandle_t handle, cudecompGridDesc_t gd);
void resetCommInfo(cudecompGridDesc_t gd);

inline int elementSize(cudecompDataType_t dtype) {
  switch (dtype) {
    case CUDECOMP_FLOAT: return 4;
    case CUDECOMP_DOUBLE:
    case CUDECOMP_FLOAT_COMPLEX: return 8;
    default: return 16;
  }
}

inline bool transposeBackendIsMpi(cudecompTransposeCommBackend_t b) {
  return b == CUDECOMP_TRANSPOSE_COMM_MPI_P2P || b == CUDECOMP_TRANSPOSE_COMM_MPI_P2P_PL ||
         b == CUDECOMP_TRANSPOSE_COMM_MPI_A2A;
}
inline bool transposeBackendIsRccl(cudecompTransposeCommBackend_t b) {
  return b == CUDECOMP_TRANSPOSE_COMM_NCCL || b == CUDECOMP_TRANSPOSE_COMM_NCCL_PL;
}
inline bool transposeBackendIsPeer(cudecompTransposeCommBackend_t b) {
  return b == CUDECOMP_TRANSPOSE_COMM_NVSHMEM || b == CUDECOMP_TRANSPOSE_COMM_NVSHMEM_PL ||
         b == CUDECOMP_TRANSPOSE_COMM_NVSHMEM_SM;
}
inline bool transposeBackendIsPipelined(cudecompTransposeCommBackend_t b) {
  return b == CUDECOMP_TRANSPOSE_COMM_MPI_P2P_PL || b == CUDECOMP_TRANSPOSE_COMM_NCCL_PL ||
         b == CUDECOMP_TRANSPOSE_COMM_NVSHMEM_PL;
}
inline bool haloBackendIsMpi(cudecompHaloCommBackend_t b) {
  return b == CUDECOMP_HALO_COMM_MPI || b == CUDECOMP_HALO_COMM_MPI_BLOCKING;
}
inline bool haloBackendIsRccl(cudecompHaloCommBackend_t b) { return b == CUDECOMP_HALO_COMM_NCCL; }
inline bool haloBackendIsPeer(cudecompHaloCommBackend_t b) {
  return b == CUDECOMP_HALO_COMM_NVSHMEM || b == CUDECOMP_HALO_COMM_NVSHMEM_BLOCKING;
}

// Which backends travel over the one-sided xGMI peer transport: the NVSHMEM enums always, the MPI enums unless
// a real MPI communicator is behind the handle (MPI flavour of the library, launched under MPI).
inline bool usesPeerTransport(cudecompHandle_t h, cudecompTransposeCommBackend_t b) {
  if (transposeBackendIsRccl(b)) return false;
  return transposeBackendIsPeer(b) || h->boot->nativeComm() == nullptr;
}
inline bool usesPeerTransport(cudecompHandle_t h, cudecompHaloCommBackend_t b) {
  if (haloBackendIsRccl(b)) return false;
  return haloBackendIsPeer(b) || h->boot->nativeComm() == nullptr;
}

// executors
void runTranspose(cudecompHandle_t handle, cudecompGridDesc_t gd, TransposeOp op, void* input, void* output, void* work,
                  cudecompDataType_t dtype, const int32_t* in_halo, const int32_t* out_halo, const int32_t* in_pad,
                  const int32_t* out_pad, hipStream_t stream);
void runHalo(cudecompHandle_t handle, cudecompGridDesc_t gd, int axis, void* input, void* work,
             cudecompDataType_t dtype, const int32_t* halo, const bool* periods, int dim, const int32_t* pad,
             hipStream_t stream);

// perf.cc
struct TransposeTimings {
  int64_t calls = 0, samples = 0;
  double total_ms = 0, pack_ms = 0, exchange_ms = 0, unpack_ms = 0;  // averages over the retained samples
  int64_t pencil_bytes = 0;
};
hipEvent_t* perfBeginTranspose(cudecompHandle_t h, cudecompGridDesc_t gd, int op, cudecompDataType_t dtype,
                               const std::array<int32_t, 12>& halos_pads, bool inplace, int64_t wire_bytes,
                               hipStream_t stream);
hipEvent_t* perfBeginHalo(cudecompHandle_t h, cudecompGridDesc_t gd, int axis, int dim, cudecompDataType_t dtype,
                          const std::array<int32_t, 3>& halo, const std::array<bool, 3>& periods,
                          const std::array<int32_t, 3>& padding, int64_t wire_bytes, hipStream_t stream);
inline void perfMark(hipEvent_t* ev, int which, hipStream_t stream) {
  if (ev) (void)hipEventRecord(ev[which], stream);
}
TransposeTimings perfCollectHalo(cudecompGridDesc_t gd, int axis, int dim);  // same fields, wire bytes in pencil_bytes
TransposeTimings perfCollect(cudecompGridDesc_t gd, int op);  // all configurations of one op; synchronises the device
void perfReport(cudecompHandle_t h, cudecompGridDesc_t gd);   // collective
void perfReset(cudecompGridDesc_t gd);
void perfDestroy(cudecompGridDesc_t gd);

// autotune.cc
void autotuneTranspose(cudecompHandle_t handle, cudecompGridDesc_t gd, const cudecompGridDescAutotuneOptions_t* opt,
                       bool autotune_backend, bool autotune_pdims);
// analytic cost (ms) of one X->Y->Z->Y->X cycle for a (grid, backend) candidate on an xGMI full mesh
double estimateTransposeCycleMs(cudecompHandle_t h, const GridShape& g, int es, cudecompTransposeCommBackend_t backend,
                                bool library_buffers, const bool inplace[4]);
void autotuneHalo(cudecompHandle_t handle, cudecompGridDesc_t gd, const cudecompGridDescAutotuneOptions_t* opt,
                  bool autotune_backend, bool autotune_pdims);
std::vector<cudecompTransposeCommBackend_t> transposeBackendCandidates(const cudecompGridDescAutotuneOptions_t* opt);
std::vector<cudecompHaloCommBackend_t> haloBackendCandidates(const cudecompGridDescAutotuneOptions_t* opt);
std::vector<std::array<int32_t, 2>> pdimCandidates(int nranks, bool col_major);

}  // namespace cudecomp
