// api.cc -- the extern "C" boundary: the 24 entry points of cudecomp.h.
//
// Validation order, result codes and in/out semantics follow NVIDIA/cuDecomp v0.7.0 (reference
// src/cudecomp.cc:903-2045; the behaviours asserted by reference tests/ctest/api_tests.cc are the
// specification).  Nothing throws across this file's functions.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <set>

#include <unistd.h>

#include "errors.h"
#include "internal.h"
#include "transport.h"

using namespace cudecomp;

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
namespace {

int g_bootstrap_instances = 0;

cudecompResult_t report(const Error& e) {
  std::cerr << e.what();
  return e.code();
}
cudecompResult_t reportUnexpected(const char* what) {
  std::cerr << "CUDECOMP:ERROR: Internal error. (" << what << ")\n";
  return CUDECOMP_RESULT_INTERNAL_ERROR;
}

#define CD_API_CATCH(...)                            \
  catch (const ::cudecomp::Error& e) {               \
    __VA_ARGS__;                                     \
    return report(e);                                \
  }                                                  \
  catch (const std::exception& e) {                  \
    __VA_ARGS__;                                     \
    return reportUnexpected(e.what());               \
  }                                                  \
  catch (...) {                                      \
    __VA_ARGS__;                                     \
    return reportUnexpected("unknown exception");    \
  }

bool envIsOne(const char* name) {
  const char* v = std::getenv(name);
  return v && std::strtol(v, nullptr, 10) == 1;
}

void checkHandle(cudecompHandle_t h) {
  if (!h || !h->initialized) CD_INVALID_USAGE("invalid handle");
}
void checkGridDesc(cudecompHandle_t h, cudecompGridDesc_t gd) {
  if (!gd || !gd->initialized) CD_INVALID_USAGE("invalid grid descriptor");
  if (gd->handle != h) CD_INVALID_USAGE("grid descriptor belongs to a different handle");
}
void checkDataType(cudecompDataType_t t) {
  switch (t) {
    case CUDECOMP_FLOAT:
    case CUDECOMP_DOUBLE:
    case CUDECOMP_FLOAT_COMPLEX:
    case CUDECOMP_DOUBLE_COMPLEX: return;
    default: CD_INVALID_USAGE("unknown data type");
  }
}
// Enum fields of caller-filled structs are read as the 32-bit integers they are in memory: a value outside the enum's range
// (exactly what these checks are for) must not be loaded through the enum type first (undefined behaviour; UBSan flags it).
template <typename E>
int32_t rawEnum(const E& field) {
  static_assert(sizeof(E) == sizeof(int32_t), "enum fields of the public structs are 32-bit");
  int32_t v;
  std::memcpy(&v, &field, sizeof(v));
  return v;
}
void checkTransposeBackend(int32_t b) {
  if (b < CUDECOMP_TRANSPOSE_COMM_MPI_P2P || b > CUDECOMP_TRANSPOSE_COMM_NVSHMEM_SM)
    CD_INVALID_USAGE("unknown transpose communication type");
}
void checkHaloBackend(int32_t b) {
  if (b < CUDECOMP_HALO_COMM_MPI || b > CUDECOMP_HALO_COMM_NVSHMEM_BLOCKING)
    CD_INVALID_USAGE("unknown halo communication type");
}
void checkRankOrder(int32_t r) {
  if (r != CUDECOMP_RANK_ORDER_DEFAULT && r != CUDECOMP_RANK_ORDER_ROW_MAJOR && r != CUDECOMP_RANK_ORDER_COL_MAJOR)
    CD_INVALID_USAGE("unknown rank order");
}

// ---- versioned struct plumbing (layout version 1 is the only one that exists) ---------------------
constexpr int64_t kConfigSizeV1 = 104, kOptionsSizeV1 = 320, kPencilInfoSizeV1 = 96;
static_assert(sizeof(cudecompGridDescConfig_t) == kConfigSizeV1, "config ABI size");
static_assert(sizeof(cudecompGridDescAutotuneOptions_t) == kOptionsSizeV1, "autotune options ABI size");
static_assert(sizeof(cudecompPencilInfo_t) == kPencilInfoSizeV1, "pencil info ABI size");

int64_t expectedSize(int32_t version, int32_t current, int64_t size_v1, const char* what) {
  if (version > current)
    CD_INVALID_USAGE(std::string(what) + " was initialized with a newer cuDecomp header than this runtime library supports");
  if (version != 1) CD_INVALID_USAGE(std::string(what) + " layout version is unsupported");
  return size_v1;
}
int64_t configSize(int32_t v) { return expectedSize(v, CUDECOMP_GRID_DESC_CONFIG_VERSION, kConfigSizeV1, "config"); }
int64_t optionsSize(int32_t v) {
  return expectedSize(v, CUDECOMP_GRID_DESC_AUTOTUNE_OPTIONS_VERSION, kOptionsSizeV1, "options");
}
int64_t pencilInfoSize(int32_t v) {
  return expectedSize(v, CUDECOMP_PENCIL_INFO_VERSION, kPencilInfoSizeV1, "pencil_info");
}
// bytes of meaningful payload (header + fields, without trailing alignment padding)
constexpr size_t kConfigPayload = offsetof(cudecompGridDescConfig_t, halo_comm_backend) + sizeof(cudecompHaloCommBackend_t);
constexpr size_t kOptionsPayload = offsetof(cudecompGridDescAutotuneOptions_t, halo_padding) + sizeof(int32_t[3]);
constexpr size_t kPencilPayload = offsetof(cudecompPencilInfo_t, size) + sizeof(int64_t);

void fillConfigDefaults(cudecompGridDescConfig_t* c, int64_t struct_size, int32_t version) {
  cudecompGridDescConfig_t d{};
  d.transpose_comm_backend = CUDECOMP_TRANSPOSE_COMM_MPI_P2P;
  d.halo_comm_backend = CUDECOMP_HALO_COMM_MPI;
  d.rank_order = CUDECOMP_RANK_ORDER_DEFAULT;
  for (auto& row : d.transpose_mem_order)
    for (auto& v : row) v = -1;
  std::memcpy(c, &d, kConfigPayload);
  c->struct_size = struct_size;
  c->magic = CUDECOMP_GRID_DESC_CONFIG_MAGIC;
  c->version = version;
}

void fillOptionsDefaults(cudecompGridDescAutotuneOptions_t* o, int64_t struct_size, int32_t version) {
  cudecompGridDescAutotuneOptions_t d{};
  d.n_warmup_trials = 3;
  d.n_trials = 5;
  d.grid_mode = CUDECOMP_AUTOTUNE_GRID_TRANSPOSE;
  d.dtype = CUDECOMP_DOUBLE;
  d.allow_uneven_decompositions = true;
  d.skip_threshold = 0.0;
  for (double& w : d.transpose_op_weights) w = 1.0;
  std::memcpy(o, &d, kOptionsPayload);
  o->struct_size = struct_size;
  o->magic = CUDECOMP_GRID_DESC_AUTOTUNE_OPTIONS_MAGIC;
  o->version = version;
}

void copyConfigOut(cudecompGridDescConfig_t* dst, int64_t struct_size, int32_t version, const cudecompGridDesc_t gd) {
  if (struct_size != configSize(version)) CD_INVALID_USAGE("config struct_size does not match its cuDecomp layout version");
  std::memcpy(dst, &gd->config, kConfigPayload);
  dst->struct_size = struct_size;
  dst->magic = CUDECOMP_GRID_DESC_CONFIG_MAGIC;
  dst->version = version;
  // fields the user left at "default" are reported back as such
  if (!gd->gdims_dist_set)
    for (int i = 0; i < 3; ++i) dst->gdims_dist[i] = 0;
  if (!gd->mem_order_set)
    for (auto& row : dst->transpose_mem_order)
      for (auto& v : row) v = -1;
}

void validateConfig(cudecompHandle_t h, const cudecompGridDescConfig_t& c, bool autotune_transpose, bool autotune_halo) {
  if (!autotune_transpose) checkTransposeBackend(rawEnum(c.transpose_comm_backend));
  if (!autotune_halo) checkHaloBackend(rawEnum(c.halo_comm_backend));
  checkRankOrder(rawEnum(c.rank_order));
  if (c.pdims[0] < 0 || c.pdims[1] < 0) CD_INVALID_USAGE("pdims values are invalid");
  const int64_t prod = (int64_t)c.pdims[0] * c.pdims[1];
  if (prod == 0) {
    if (c.pdims[0] != 0 || c.pdims[1] != 0) CD_INVALID_USAGE("pdims values are invalid");
  } else if (prod != h->nranks) {
    CD_INVALID_USAGE("product of pdims values must equal number of ranks");
  }
  const bool set = c.transpose_mem_order[0][0] >= 0;
  for (auto& row : c.transpose_mem_order)
    for (int v : row)
      if (set != (v >= 0)) CD_INVALID_USAGE("transpose_mem_order only partially set");
  if (set) {
    for (auto& row : c.transpose_mem_order) {
      std::set<int32_t> vals(row, row + 3);
      if (vals.size() != 3 || *vals.begin() != 0 || *vals.rbegin() != 2)
        CD_INVALID_USAGE("transpose_mem_order setting is invalid");
    }
  }
}

void resolveRankOrder(cudecompHandle_t h, cudecompGridDesc_t gd) {
  const char* env = std::getenv("CUDECOMP_USE_COL_MAJOR_RANK_ORDER");
  if (env && !h->col_major_env_warned) {
    if (h->rank == 0)
      printf("CUDECOMP:WARN: CUDECOMP_USE_COL_MAJOR_RANK_ORDER is deprecated and will be removed in a future "
             "release. Set cudecompGridDescConfig_t::rank_order instead.\n");
    h->col_major_env_warned = true;
  }
  if (gd->config.rank_order == CUDECOMP_RANK_ORDER_DEFAULT)
    gd->config.rank_order = (env && envIsOne("CUDECOMP_USE_COL_MAJOR_RANK_ORDER")) ? CUDECOMP_RANK_ORDER_COL_MAJOR
                                                                                  : CUDECOMP_RANK_ORDER_ROW_MAJOR;
}

void syncShapeFromConfig(cudecompGridDesc_t gd) {
  GridShape& s = gd->shape;
  for (int i = 0; i < 3; ++i) {
    s.gdims[i] = gd->config.gdims[i];
    s.gdims_dist[i] = gd->config.gdims_dist[i];
    for (int j = 0; j < 3; ++j) s.mem_order[i][j] = gd->config.transpose_mem_order[i][j];
  }
  s.pdims = {gd->config.pdims[0], gd->config.pdims[1]};
  s.col_major = gd->config.rank_order == CUDECOMP_RANK_ORDER_COL_MAJOR;
}

}  // namespace

namespace cudecomp {

void ensureDevice(cudecompHandle_t h) {
  if (!h->device_probed) {
    h->device_probed = true;
    int dev = -1;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0) {
      h->device = dev;
      (void)hipDeviceGetAttribute(&h->num_cus, hipDeviceAttributeMultiprocessorCount, dev);
    } else {
      (void)hipGetLastError();
    }
  }
  if (h->device < 0) CD_THROW(CUDECOMP_RESULT_CUDA_ERROR, "CUDA error.", "no usable HIP device (the library has no CPU fallback)");
}

void resetCommInfo(cudecompGridDesc_t gd) {
  gd->row.release();
  gd->col.release();
  gd->world.release();
}

// Row / column communicators of the process grid.  Members of my row share pidx[0]; they are ordered by
// their column index, which is what the reference's MPI_Comm_split(color = pidx[0], key = rank) yields
// for both rank orders (reference include/internal/common.h:496-531).
void buildCommInfo(cudecompHandle_t h, cudecompGridDesc_t gd) {
  syncShapeFromConfig(gd);
  resetCommInfo(gd);
  gd->pidx = gridIndexOfRank(gd->shape, h->rank);
  struct {
    cudecompCommInfo* info;
    CommAxis axis;
  } both[2] = {{&gd->row, COMM_ROW}, {&gd->col, COMM_COL}};
  for (auto& e : both) {
    cudecompCommInfo& ci = *e.info;
    // row of this communicator in the node's shared board (the same index on every rank: descriptors are created and
    // destroyed collectively, in the same order everywhere); -1 when all rows are taken: host barriers then go through
    // the bootstrap and the one-sided transport is unavailable for this descriptor
    ci.owner = h;
    ci.barrier_slot = h->acquireSlot();
    ci.nranks = gd->shape.pdims[e.axis == COMM_ROW ? 1 : 0];
    ci.rank = gd->pidx[e.axis == COMM_ROW ? 1 : 0];
    ci.boot = h->boot->split(gd->pidx[e.axis == COMM_ROW ? 0 : 1], h->rank);
    if (ci.boot->size() != ci.nranks || ci.boot->rank() != ci.rank)
      CD_INTERNAL_ERROR("communicator split disagrees with the process grid");
    ci.global_ranks.resize(ci.nranks);
    std::map<std::string, int> per_host;
    for (int i = 0; i < ci.nranks; ++i) {
      ci.global_ranks[i] = globalRankOf(gd->shape, gd->pidx, e.axis, i);
      per_host[h->hostnames[ci.global_ranks[i]]]++;
    }
    // largest homogeneous group of ranks that share a host (= an xGMI mesh)
    int count = 0;
    for (auto& kv : per_host) {
      int a = count, b = kv.second;
      while (b) {
        int t = a % b;
        a = b;
        b = t;
      }
      count = (count == 0) ? kv.second : a;
    }
    ci.npergroup = count;
    ci.ngroups = ci.nranks / ci.npergroup;
    // counters of the row continue above anything a member has ever seen there (nothing is reset, see internal.h)
    uint64_t high = 0;
    if (ci.barrier_slot >= 0) high = std::max<uint64_t>(h->slot_high[ci.barrier_slot], peerSlotHigh(h, ci.barrier_slot));
    // over ALL ranks of the handle, not only the new members: the row may hold values written by ranks that shared the
    // row's previous communicator with a member but are not members now
    if (h->nranks > 1) high = (uint64_t)h->boot->allreduceMaxI64((int64_t)high);
    ci.barrier_epoch = ci.mail_seq = ci.epoch_base = high;
  }
  // the two-hop relay orders its two steps with flags of a communicator of ALL ranks
  if (h->two_hop_relay && h->nranks >= 4) {
    cudecompCommInfo& ci = gd->world;
    ci.owner = h;
    ci.barrier_slot = h->acquireSlot();
    ci.nranks = h->nranks;
    ci.rank = h->rank;
    ci.boot = h->boot->split(0, h->rank);
    ci.global_ranks.resize(ci.nranks);
    std::map<std::string, int> per_host;
    for (int i = 0; i < ci.nranks; ++i) {
      ci.global_ranks[i] = i;
      per_host[h->hostnames[i]]++;
    }
    ci.ngroups = (int)per_host.size();  // (the relay needs one node: ngroups == 1)
    ci.npergroup = ci.nranks / std::max(ci.ngroups, 1);
    uint64_t high = 0;
    if (ci.barrier_slot >= 0) high = std::max<uint64_t>(h->slot_high[ci.barrier_slot], peerSlotHigh(h, ci.barrier_slot));
    high = (uint64_t)h->boot->allreduceMaxI64((int64_t)high);
    ci.barrier_epoch = ci.mail_seq = ci.epoch_base = high;
  }
}

}  // namespace cudecomp

int cudecompHandle::acquireSlot() {
  for (size_t i = 0; i < slot_used.size(); ++i)
    if (!slot_used[i]) {
      slot_used[i] = true;
      return (int)i;
    }
  return -1;
}

void cudecompHandle::releaseSlot(int slot, uint64_t high) {
  if (slot < 0 || slot >= (int)slot_used.size()) return;
  slot_used[slot] = false;
  slot_high[slot] = std::max(slot_high[slot], high);
}

void cudecompCommInfo::release() {
  uint64_t high = std::max({barrier_epoch, mail_seq, epoch_base});
  if (dev_epoch) {
    // everything this rank enqueued on the communicator has run: its signals are out, the ones it waited for are in
    (void)hipDeviceSynchronize();
    unsigned long long v = 0;
    if (hipMemcpy(&v, dev_epoch, sizeof(v), hipMemcpyDeviceToHost) == hipSuccess) high = std::max<uint64_t>(high, v);
    // (the cell belongs to the handle's slab of epoch cells, transport.cc devEpoch: nothing to free here)
    (void)hipGetLastError();
    dev_epoch = nullptr;
  }
  if (owner && barrier_slot >= 0) owner->releaseSlot(barrier_slot, high);
  owner = nullptr;
  barrier_slot = -1;
  rank = nranks = 0;
  ngroups = npergroup = 1;
  global_ranks.clear();
  boot.reset();
  barrier_epoch = mail_seq = epoch_base = 0;
}

cudecompCommInfo::~cudecompCommInfo() { release(); }

cudecompHandle::~cudecompHandle() {
  if (relay_buf && peer) {
    try {
      cudecomp::workspaceFreeRaw(this, relay_buf);  // (collective, like the cudecompFinalize it runs in)
    } catch (...) {
    }
    relay_buf = nullptr;
  }
  if (relay_last_call) (void)hipEventDestroy(relay_last_call);
  for (hipStream_t s : streams) (void)hipStreamDestroy(s);
  rccl.reset();
  peer.reset();
}

cudecompGridDesc::~cudecompGridDesc() {
  cudecomp::perfDestroy(this);
  for (hipEvent_t e : events) (void)hipEventDestroy(e);
  for (auto& kv : pack_graphs) (void)hipGraphExecDestroy(kv.second);
  for (auto& kv : op_graphs) (void)hipGraphExecDestroy(kv.second);
  if (graph_stream) (void)hipStreamDestroy(graph_stream);
}

// ------------------------------------------------------------------------------------------------
// library lifetime
// ------------------------------------------------------------------------------------------------
extern "C" {

cudecompResult_t cudecompInit(cudecompHandle_t* handle_out, MPI_Comm mpi_comm) {
  cudecompHandle_t h = nullptr;
  try {
    if (!handle_out) CD_INVALID_USAGE("handle argument cannot be null");
    h = new cudecompHandle;
    h->boot = makeWorldBootstrap(mpi_comm, g_bootstrap_instances++);
    h->rank = h->boot->rank();
    h->nranks = h->boot->size();

    // hostnames -> which ranks share an xGMI node
    char name[256] = {0};
    ::gethostname(name, sizeof(name) - 1);
    // test hook (the reference's tests overwrite handle->hostnames for the same purpose,
    // tests/ctest/transpose_tests.cc:430-456): make one node look like several
    if (const char* fake = std::getenv("CUDECOMP_HOSTNAME_OVERRIDE")) std::snprintf(name, sizeof(name), "%s", fake);
    std::vector<char> all((size_t)256 * h->nranks);
    h->boot->allgather(name, all.data(), 256);
    h->hostnames.resize(h->nranks);
    h->rank_to_local_rank.assign(h->nranks, 0);
    std::map<std::string, int> seen;
    for (int r = 0; r < h->nranks; ++r) {
      h->hostnames[r] = std::string(all.data() + (size_t)256 * r);
      h->rank_to_local_rank[r] = seen[h->hostnames[r]]++;
    }
    h->local_rank = h->rank_to_local_rank[h->rank];
    h->local_nranks = seen[h->hostnames[h->rank]];

    h->graphs_enable = envIsOne("CUDECOMP_ENABLE_CUDA_GRAPHS") || envIsOne("CUDECOMP_ENABLE_HIP_GRAPHS");
    h->performance_report_enable = envIsOne("CUDECOMP_ENABLE_PERFORMANCE_REPORT");
    // report options, docs/env_vars.rst of the reference (defaults 0 / 20 / 3 / unset; bad values warn and keep them)
    auto envInt = [&](const char* var, int lo, int hi, int dflt) {
      const char* v = std::getenv(var);
      if (!v || !*v) return dflt;
      char* end = nullptr;
      const long x = std::strtol(v, &end, 10);
      if (*end != '\0' || x < lo || x > hi) {
        if (h->rank == 0) printf("CUDECOMP:WARN: Invalid %s value (%s). Using default (%d).\n", var, v, dflt);
        return dflt;
      }
      return (int)x;
    };
    h->performance_report_detail = envInt("CUDECOMP_PERFORMANCE_REPORT_DETAIL", 0, 2, 0);
    h->performance_report_samples = envInt("CUDECOMP_PERFORMANCE_REPORT_SAMPLES", 1, 1 << 20, 20);
    h->performance_report_warmup_samples = envInt("CUDECOMP_PERFORMANCE_REPORT_WARMUP_SAMPLES", 0, 1 << 30, 3);
    if (const char* v = std::getenv("CUDECOMP_PERFORMANCE_REPORT_WRITE_DIR")) h->performance_report_write_dir = v;
    h->halo_overlap_disable = envIsOne("CUDECOMP_DISABLE_HALO_OVERLAP");
    h->halo_overlap_force = envIsOne("CUDECOMP_FORCE_HALO_OVERLAP");
    h->self_exchange = envIsOne("CUDECOMP_TEST_SELF_EXCHANGE");
    if (const char* v = std::getenv("CUDECOMP_RCCL_NATIVE_ALLTOALL")) h->rccl_native_alltoall = std::strtol(v, nullptr, 10) != 0;
    h->direct_put = !envIsOne("CUDECOMP_DISABLE_DIRECT_PUT");
    h->inplace_rotation = !envIsOne("CUDECOMP_DISABLE_INPLACE_ROTATION");
    h->two_hop_relay = envIsOne("CUDECOMP_TWO_HOP_RELAY");
    if (const char* v = std::getenv("CUDECOMP_FUSE_SMALL_EXCHANGES_KIB")) h->fuse_small_bytes = std::strtoll(v, nullptr, 10) << 10;
    h->debug_verify_exchange = envIsOne("CUDECOMP_DEBUG_VERIFY_EXCHANGE");
    if (h->debug_verify_exchange) h->fuse_small_bytes = 0;  // (the verification checksums the send area, which a fused put never fills)
    if (const char* v = std::getenv("CUDECOMP_PIPELINE_MIN_STAGE_MIB")) h->pipeline_min_stage_bytes = std::strtoll(v, nullptr, 10) << 20;
    if (const char* v = std::getenv("CUDECOMP_PIPELINE_STAGES")) {
      const long k = std::strtol(v, nullptr, 10);
      if (k >= 1 && k <= 14) h->pipeline_stages = (int)k;  // (flags carry call * 16 + step: steps 1..14 are stages, 15 = done)
      else if (h->rank == 0) printf("CUDECOMP:WARN: Invalid CUDECOMP_PIPELINE_STAGES value (%s); expected 1..14.\n", v);
    }
    if (const char* v = std::getenv("CUDECOMP_PEER_TIMEOUT")) {
      const double t = std::strtod(v, nullptr);
      if (t > 0) h->peer_timeout_s = t;
    }
    if (const char* v = std::getenv("CUDECOMP_PEER_COPY_ENGINE")) {
      // sdma: hipMemcpyAsync (copy engines / runtime blit); cu: the library's copy kernel; default: whichever the link
      // probe at start-up finds faster
      const std::string e(v);
      if (e == "sdma" || e == "cu") {
        h->peer_copy_engine = (e == "cu") ? 1 : 0;
        h->peer_copy_engine_pinned = true;
      } else if (e != "auto" && h->rank == 0) {
        printf("CUDECOMP:WARN: Invalid CUDECOMP_PEER_COPY_ENGINE value (%s); expected sdma, cu or auto.\n", v);
      }
    }
    h->slot_used.assign(256, false);
    h->slot_high.assign(256, 0);
    h->tuning.no_streaming = envIsOne("CUDECOMP_DISABLE_STREAMING_ACCESS");
    // A transpose normally never touches the halo / padding cells of its OUTPUT pencil (reference transpose.h:830-895).  Two
    // kernels here read the few cells between consecutive output rows and write them back unchanged to write whole cache
    // lines (rows_dense_kernel, transpose_lines_kernel, transpose_rowlines_kernel): a caller who writes those cells on another stream WHILE the
    // transpose runs opts out with this switch.
    if (envIsOne("CUDECOMP_PRESERVE_OUTPUT_HALOS")) h->tuning.dense_rows = 0;
    // Tuning switches (kernel variants, walk orders, diagnostic store policies): read by `make TUNING_VARIANTS=1` builds only
    // (cudecomp_amd/lib_tuning); the default build has neither the variants nor the switches and says so once.
    auto tuningSwitch = [&](const char* var, int* out) {
      const char* v = std::getenv(var);
      if (!v || !*v) return;
#ifdef CUDECOMP_TUNING_VARIANTS
      *out = (int)std::strtol(v, nullptr, 10);
#else
      (void)out;
      if (h->rank == 0)
        fprintf(stderr, "CUDECOMP:WARN: %s is a tuning switch of `make TUNING_VARIANTS=1` builds of this library; ignored.\n", var);
#endif
    };
    tuningSwitch("CUDECOMP_INTERLEAVE_ROWS", &h->tuning.interleave_rows);
    tuningSwitch("CUDECOMP_WINDOW_STORES", &h->tuning.window_mode);
    tuningSwitch("CUDECOMP_WINDOW_WIDE", &h->tuning.window_wide);
    tuningSwitch("CUDECOMP_TILE_WALK", &h->tuning.walk_order);
    tuningSwitch("CUDECOMP_TILE_SHAPE", &h->tuning.tile_shape);
    tuningSwitch("CUDECOMP_LINES_MODE", &h->tuning.lines_mode);
    tuningSwitch("CUDECOMP_LINES_UNIT", &h->tuning.lines_unit);
    tuningSwitch("CUDECOMP_LINES_RUN_KIB", &h->tuning.lines_run_kib);
    tuningSwitch("CUDECOMP_LINES_WALK", &h->tuning.lines_walk);
    tuningSwitch("CUDECOMP_LINES_GROUP", &h->tuning.lines_group);
    tuningSwitch("CUDECOMP_ROTATE_WALK", &h->tuning.rotate_walk);
    if (const char* v = std::getenv("CUDECOMP_FORCE_GENERIC_KERNELS"))
      if (std::strtol(v, nullptr, 10) == 1) h->tuning.force_class = MOVE_GENERIC;

    h->initialized = true;
    *handle_out = h;
  }
  CD_API_CATCH(delete h)
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompInit_F(cudecompHandle_t* handle_out, MPI_Fint mpi_comm_f) {
  return cudecompInit(handle_out, commFromFortran(mpi_comm_f));
}

cudecompResult_t cudecompFinalize(cudecompHandle_t handle) {
  try {
    checkHandle(handle);
    handle->initialized = false;
    std::unique_ptr<Error> pending;
    if (handle->peer) {
      // same for exchanges whose descriptor is still alive (or leaked) when the library goes away
      (void)hipDeviceSynchronize();
      (void)hipGetLastError();
      try {
        peerCheckStatus(handle);
      } catch (const Error& e) {
        pending = std::make_unique<Error>(e);
      }
      // opt-in (CUDECOMP_QUEUE_CENSUS=1): one more look at the device's hardware queues (the first was when the transport
      // came up; never per call)
      if (handle->nranks > 1 && peerQueueCensusRequested()) (void)peerQueueCensus(handle, true);
    }
    delete handle;
    if (pending) throw *pending;
  }
  CD_API_CATCH()
  return CUDECOMP_RESULT_SUCCESS;
}

// ------------------------------------------------------------------------------------------------
// grid descriptor
// ------------------------------------------------------------------------------------------------
cudecompResult_t cudecompGridDescConfigSetDefaultsVersioned(cudecompGridDescConfig_t* config, int64_t struct_size,
                                                            int32_t version) {
  try {
    if (!config) CD_INVALID_USAGE("config argument cannot be null");
    if (struct_size != configSize(version)) CD_INVALID_USAGE("config struct_size does not match its cuDecomp layout version");
    fillConfigDefaults(config, struct_size, version);
  }
  CD_API_CATCH()
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompGridDescAutotuneOptionsSetDefaultsVersioned(cudecompGridDescAutotuneOptions_t* options,
                                                                     int64_t struct_size, int32_t version) {
  try {
    if (!options) CD_INVALID_USAGE("options argument cannot be null");
    if (struct_size != optionsSize(version)) CD_INVALID_USAGE("options struct_size does not match its cuDecomp layout version");
    fillOptionsDefaults(options, struct_size, version);
  }
  CD_API_CATCH()
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompGridDescCreateVersioned(cudecompHandle_t handle, cudecompGridDesc_t* grid_desc_out,
                                                 cudecompGridDescConfig_t* config, int64_t config_struct_size,
                                                 int32_t config_version,
                                                 const cudecompGridDescAutotuneOptions_t* options,
                                                 int64_t options_struct_size, int32_t options_version) {
  cudecompGridDesc_t gd = nullptr;
  try {
    checkHandle(handle);
    if (!grid_desc_out) CD_INVALID_USAGE("grid_desc argument cannot be null");
    if (!config) CD_INVALID_USAGE("config argument cannot be null");
    if (config_struct_size != configSize(config_version))
      CD_INVALID_USAGE("config struct_size does not match its cuDecomp layout version");
    if (config->magic != CUDECOMP_GRID_DESC_CONFIG_MAGIC)
      CD_INVALID_USAGE("config is not initialized; call cudecompGridDescConfigSetDefaults() before cudecompGridDescCreate()");
    if (config->struct_size != configSize(config->version))
      CD_INVALID_USAGE("config struct_size does not match its cuDecomp layout version");
    if (config->struct_size != config_struct_size || config->version != config_version)
      CD_INVALID_USAGE("config metadata does not match the requested cuDecomp layout version");

    cudecompGridDescConfig_t cfg{};
    fillConfigDefaults(&cfg, (int64_t)sizeof(cfg), CUDECOMP_GRID_DESC_CONFIG_VERSION);
    std::memcpy(&cfg, config, kConfigPayload);
    cfg.struct_size = (int64_t)sizeof(cfg);
    cfg.version = CUDECOMP_GRID_DESC_CONFIG_VERSION;

    cudecompGridDescAutotuneOptions_t opt{};
    bool have_opt = false;
    if (options) {
      if (options_struct_size != optionsSize(options_version))
        CD_INVALID_USAGE("options struct_size does not match its cuDecomp layout version");
      if (options->magic != CUDECOMP_GRID_DESC_AUTOTUNE_OPTIONS_MAGIC)
        CD_INVALID_USAGE("options are not initialized; call cudecompGridDescAutotuneOptionsSetDefaults() before "
                         "cudecompGridDescCreate()");
      if (options->struct_size != optionsSize(options->version))
        CD_INVALID_USAGE("options struct_size does not match its cuDecomp layout version");
      if (options->struct_size != options_struct_size || options->version != options_version)
        CD_INVALID_USAGE("options metadata does not match the requested cuDecomp layout version");
      fillOptionsDefaults(&opt, (int64_t)sizeof(opt), CUDECOMP_GRID_DESC_AUTOTUNE_OPTIONS_VERSION);
      std::memcpy(&opt, options, kOptionsPayload);
      opt.struct_size = (int64_t)sizeof(opt);
      opt.version = CUDECOMP_GRID_DESC_AUTOTUNE_OPTIONS_VERSION;
      have_opt = true;
    }
    const bool tune_tb = have_opt && opt.autotune_transpose_backend;
    const bool tune_hb = have_opt && opt.autotune_halo_backend;
    validateConfig(handle, cfg, tune_tb, tune_hb);
    const bool tune_pdims = (cfg.pdims[0] == 0 && cfg.pdims[1] == 0);
    if (tune_pdims && !have_opt) CD_INVALID_USAGE("options argument cannot be null if autotuning pdims");

    gd = new cudecompGridDesc;
    gd->initialized = true;
    gd->handle = handle;
    gd->config = cfg;
    resolveRankOrder(handle, gd);

    std::vector<cudecompTransposeCommBackend_t> t_cand;
    std::vector<cudecompHaloCommBackend_t> h_cand;
    if (tune_tb) t_cand = transposeBackendCandidates(&opt);
    if (tune_hb) h_cand = haloBackendCandidates(&opt);
    if (tune_pdims) (void)pdimCandidates(handle->nranks, gd->config.rank_order == CUDECOMP_RANK_ORDER_COL_MAJOR);

    gd->mem_order_set = cfg.transpose_mem_order[0][0] >= 0;
    if (!gd->mem_order_set)
      for (int axis = 0; axis < 3; ++axis)
        for (int i = 0; i < 3; ++i)
          gd->config.transpose_mem_order[axis][i] = cfg.transpose_axis_contiguous[axis] ? (axis + i) % 3 : i;

    for (int i = 0; i < 3; ++i)
      if (cfg.gdims_dist[i] > cfg.gdims[i]) CD_INVALID_USAGE("gdims_dist entries must be less than or equal to gdims entries");
    gd->gdims_dist_set = cfg.gdims_dist[0] != 0 && cfg.gdims_dist[1] != 0 && cfg.gdims_dist[2] != 0;
    if (!gd->gdims_dist_set)
      for (int i = 0; i < 3; ++i) gd->config.gdims_dist[i] = cfg.gdims[i];

    // transports any candidate may need (collective set-up happens here, not inside the first transpose)
    bool need_rccl = (!tune_tb && transposeBackendIsRccl(cfg.transpose_comm_backend)) ||
                     (!tune_hb && haloBackendIsRccl(cfg.halo_comm_backend));
    bool need_peer = (!tune_tb && !transposeBackendIsRccl(cfg.transpose_comm_backend)) ||
                     (!tune_hb && !haloBackendIsRccl(cfg.halo_comm_backend));
    // (RCCL for autotune CANDIDATES is set up inside the sweep, which tolerates its absence)
    for (auto b : t_cand)
      if (!transposeBackendIsRccl(b)) need_peer = true;
    for (auto b : h_cand)
      if (!haloBackendIsRccl(b)) need_peer = true;
    prepareTransports(handle, need_rccl, need_peer);

    if (have_opt) {
      const int32_t grid_mode = rawEnum(opt.grid_mode);  // (validated here: never loaded through the enum type first)
      if (grid_mode == CUDECOMP_AUTOTUNE_GRID_TRANSPOSE) {
        if (tune_tb || tune_pdims) autotuneTranspose(handle, gd, &opt, tune_tb, tune_pdims);
        if (tune_hb) autotuneHalo(handle, gd, &opt, tune_hb, false);
      } else if (grid_mode == CUDECOMP_AUTOTUNE_GRID_HALO) {
        if (tune_hb || tune_pdims) autotuneHalo(handle, gd, &opt, tune_hb, tune_pdims);
        if (tune_tb) autotuneTranspose(handle, gd, &opt, tune_tb, false);
      } else {
        CD_INVALID_USAGE("unknown value of autotune_grid_mode encountered.");
      }
    }

    buildCommInfo(handle, gd);
    gd->transpose_plans.clear();
    gd->relay_plans.clear();
    gd->halo_plans.clear();
    perfReset(gd);  // autotuning trials are not part of the user's performance report

    *grid_desc_out = gd;
    copyConfigOut(config, config_struct_size, config_version, gd);
  }
  CD_API_CATCH(delete gd)
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompGridDescDestroy(cudecompHandle_t handle, cudecompGridDesc_t grid_desc) {
  try {
    checkHandle(handle);
    checkGridDesc(handle, grid_desc);
    perfReport(handle, grid_desc);
    grid_desc->initialized = false;
    const bool one_sided = grid_desc->row.dev_epoch || grid_desc->col.dev_epoch;
    delete grid_desc;  // (drains the device if the descriptor ran one-sided exchanges)
    // the LAST exchange of a descriptor has no later call that would report a wait kernel that gave up: do it here
    if (one_sided) peerCheckStatus(handle);
    // mappings of user buffers their owners have re-created since are kept open on purpose (transport.cc map()); bound them
    peerTrimRetiredImports(handle, 32);

  }
  CD_API_CATCH()
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompGetGridDescConfigVersioned(cudecompHandle_t handle, cudecompGridDesc_t grid_desc,
                                                    cudecompGridDescConfig_t* config, int64_t struct_size,
                                                    int32_t version) {
  try {
    checkHandle(handle);
    checkGridDesc(handle, grid_desc);
    if (!config) CD_INVALID_USAGE("config argument cannot be null.");
    copyConfigOut(config, struct_size, version, grid_desc);
  }
  CD_API_CATCH()
  return CUDECOMP_RESULT_SUCCESS;
}

// ------------------------------------------------------------------------------------------------
// queries
// ------------------------------------------------------------------------------------------------
cudecompResult_t cudecompGetPencilInfoVersioned(cudecompHandle_t handle, cudecompGridDesc_t grid_desc,
                                                cudecompPencilInfo_t* pencil_info, int64_t struct_size, int32_t version,
                                                int32_t axis, const int32_t halo_extents[], const int32_t padding[]) {
  try {
    checkHandle(handle);
    checkGridDesc(handle, grid_desc);
    if (!pencil_info) CD_INVALID_USAGE("pencil_info argument cannot be null.");
    if (struct_size != pencilInfoSize(version))
      CD_INVALID_USAGE("pencil_info struct_size does not match its cuDecomp layout version");
    if (axis < 0 || axis > 2) CD_INVALID_USAGE("axis argument out of range");
    const Pencil p = makePencil(grid_desc->shape, grid_desc->pidx, axis, halo_extents, padding);
    cudecompPencilInfo_t out{};
    for (int i = 0; i < 3; ++i) {
      out.shape[i] = p.shape[i];
      out.lo[i] = p.lo[i];
      out.hi[i] = p.hi[i];
      out.order[i] = p.order[i];
      out.halo_extents[i] = p.halo[i];
      out.padding[i] = p.pad[i];
    }
    out.size = p.size;
    std::memcpy(pencil_info, &out, kPencilPayload);
    pencil_info->struct_size = struct_size;
    pencil_info->magic = CUDECOMP_PENCIL_INFO_MAGIC;
    pencil_info->version = version;
  }
  CD_API_CATCH()
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompGetTransposeWorkspaceSize(cudecompHandle_t handle, cudecompGridDesc_t grid_desc,
                                                   int64_t* workspace_size) {
  try {
    checkHandle(handle);
    checkGridDesc(handle, grid_desc);
    if (!workspace_size) CD_INVALID_USAGE("workspace_size argument cannot be null.");
    *workspace_size = transposeWorkspaceElements(grid_desc->shape);
  }
  CD_API_CATCH()
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompGetHaloWorkspaceSize(cudecompHandle_t handle, cudecompGridDesc_t grid_desc, int32_t axis,
                                              const int32_t halo_extents[], int64_t* workspace_size) {
  try {
    checkHandle(handle);
    checkGridDesc(handle, grid_desc);
    if (axis < 0 || axis > 2) CD_INVALID_USAGE("axis argument out of range");
    if (!halo_extents) CD_INVALID_USAGE("halo_extents argument cannot be null.");
    if (!workspace_size) CD_INVALID_USAGE("workspace_size argument cannot be null.");
    *workspace_size = haloWorkspaceElements(grid_desc->shape, grid_desc->pidx, axis, halo_extents);
  }
  CD_API_CATCH()
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompGetDataTypeSize(cudecompDataType_t dtype, int64_t* dtype_size) {
  try {
    checkDataType(dtype);
    if (!dtype_size) CD_INVALID_USAGE("dtype_size cannot be null.");
    *dtype_size = elementSize(dtype);
  }
  CD_API_CATCH()
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompGetShiftedRank(cudecompHandle_t handle, cudecompGridDesc_t grid_desc, int32_t axis,
                                        int32_t dim, int32_t displacement, bool periodic, int32_t* shifted_rank) {
  try {
    checkHandle(handle);
    checkGridDesc(handle, grid_desc);
    if (axis < 0 || axis > 2) CD_INVALID_USAGE("axis argument out of range");
    if (dim < 0 || dim > 2) CD_INVALID_USAGE("dim argument out of range");
    if (!shifted_rank) CD_INVALID_USAGE("shifted_rank argument cannot be null.");
    *shifted_rank = shiftedRank(grid_desc->shape, handle->rank, axis, dim, displacement, periodic);
  }
  CD_API_CATCH()
  return CUDECOMP_RESULT_SUCCESS;
}

const char* cudecompTransposeCommBackendToString(cudecompTransposeCommBackend_t b) {
  switch (b) {
    case CUDECOMP_TRANSPOSE_COMM_NCCL: return "NCCL";
    case CUDECOMP_TRANSPOSE_COMM_NCCL_PL: return "NCCL (pipelined)";
    case CUDECOMP_TRANSPOSE_COMM_MPI_P2P: return "MPI_P2P";
    case CUDECOMP_TRANSPOSE_COMM_MPI_P2P_PL: return "MPI_P2P (pipelined)";
    case CUDECOMP_TRANSPOSE_COMM_MPI_A2A: return "MPI_A2A";
    case CUDECOMP_TRANSPOSE_COMM_NVSHMEM: return "NVSHMEM";
    case CUDECOMP_TRANSPOSE_COMM_NVSHMEM_PL: return "NVSHMEM (pipelined)";
    case CUDECOMP_TRANSPOSE_COMM_NVSHMEM_SM: return "NVSHMEM_SM";
    default: return "ERROR";
  }
}

const char* cudecompHaloCommBackendToString(cudecompHaloCommBackend_t b) {
  switch (b) {
    case CUDECOMP_HALO_COMM_NCCL: return "NCCL";
    case CUDECOMP_HALO_COMM_MPI: return "MPI";
    case CUDECOMP_HALO_COMM_MPI_BLOCKING: return "MPI (blocking)";
    case CUDECOMP_HALO_COMM_NVSHMEM: return "NVSHMEM";
    case CUDECOMP_HALO_COMM_NVSHMEM_BLOCKING: return "NVSHMEM (blocking)";
    default: return "ERROR";
  }
}

// ------------------------------------------------------------------------------------------------
// workspace allocation
// ------------------------------------------------------------------------------------------------
cudecompResult_t cudecompMalloc(cudecompHandle_t handle, cudecompGridDesc_t grid_desc, void** buffer,
                                size_t buffer_size_bytes) {
  try {
    checkHandle(handle);
    checkGridDesc(handle, grid_desc);
    if (!buffer) CD_INVALID_USAGE("buffer argument cannot be null");
    if (buffer_size_bytes == 0) CD_INVALID_USAGE("buffer size cannot be zero");
    ensureDevice(handle);
    *buffer = workspaceAlloc(handle, grid_desc, buffer_size_bytes);
  }
  CD_API_CATCH()
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompFree(cudecompHandle_t handle, cudecompGridDesc_t grid_desc, void* buffer) {
  try {
    checkHandle(handle);
    checkGridDesc(handle, grid_desc);
    if (buffer) workspaceFree(handle, grid_desc, buffer);
  }
  CD_API_CATCH()
  return CUDECOMP_RESULT_SUCCESS;
}

// ------------------------------------------------------------------------------------------------
// transposes and halo updates
// ------------------------------------------------------------------------------------------------
static cudecompResult_t transposeEntry(TransposeOp op, cudecompHandle_t handle, cudecompGridDesc_t grid_desc,
                                       void* input, void* output, void* work, cudecompDataType_t dtype,
                                       const int32_t in_halo[], const int32_t out_halo[], const int32_t in_pad[],
                                       const int32_t out_pad[], hipStream_t stream) {
  try {
    checkHandle(handle);
    checkGridDesc(handle, grid_desc);
    checkDataType(dtype);
    if (!input) CD_INVALID_USAGE("input argument cannot be null");
    if (!output) CD_INVALID_USAGE("output argument cannot be null");
    if (!work) CD_INVALID_USAGE("work argument cannot be null");
    runTranspose(handle, grid_desc, op, input, output, work, dtype, in_halo, out_halo, in_pad, out_pad, stream);
  }
  CD_API_CATCH()
  return CUDECOMP_RESULT_SUCCESS;
}

#define CD_DEFINE_TRANSPOSE(NAME, OP)                                                                              \
  cudecompResult_t NAME(cudecompHandle_t handle, cudecompGridDesc_t grid_desc, void* input, void* output, void* work, \
                        cudecompDataType_t dtype, const int32_t input_halo_extents[],                              \
                        const int32_t output_halo_extents[], const int32_t input_padding[],                        \
                        const int32_t output_padding[], hipStream_t stream) {                                      \
    return transposeEntry(OP, handle, grid_desc, input, output, work, dtype, input_halo_extents,                   \
                          output_halo_extents, input_padding, output_padding, stream);                             \
  }
CD_DEFINE_TRANSPOSE(cudecompTransposeXToY, OP_X_TO_Y)
CD_DEFINE_TRANSPOSE(cudecompTransposeYToZ, OP_Y_TO_Z)
CD_DEFINE_TRANSPOSE(cudecompTransposeZToY, OP_Z_TO_Y)
CD_DEFINE_TRANSPOSE(cudecompTransposeYToX, OP_Y_TO_X)

static cudecompResult_t haloEntry(int axis, cudecompHandle_t handle, cudecompGridDesc_t grid_desc, void* input,
                                  void* work, cudecompDataType_t dtype, const int32_t halo_extents[],
                                  const bool halo_periods[], int32_t dim, const int32_t padding[], hipStream_t stream) {
  try {
    checkHandle(handle);
    checkGridDesc(handle, grid_desc);
    checkDataType(dtype);
    if (!halo_extents) CD_INVALID_USAGE("halo_extents argument cannot be null");
    if (halo_extents[0] == 0 && halo_extents[1] == 0 && halo_extents[2] == 0) return CUDECOMP_RESULT_SUCCESS;
    if (!input) CD_INVALID_USAGE("input argument cannot be null");
    if (!work) CD_INVALID_USAGE("work argument cannot be null");
    if (dim < 0 || dim > 2) CD_INVALID_USAGE("dim argument out of range");
    runHalo(handle, grid_desc, axis, input, work, dtype, halo_extents, halo_periods, dim, padding, stream);
  }
  CD_API_CATCH()
  return CUDECOMP_RESULT_SUCCESS;
}

#define CD_DEFINE_HALO(NAME, AXIS)                                                                                \
  cudecompResult_t NAME(cudecompHandle_t handle, cudecompGridDesc_t grid_desc, void* input, void* work,           \
                        cudecompDataType_t dtype, const int32_t halo_extents[], const bool halo_periods[],        \
                        int32_t dim, const int32_t padding[], hipStream_t stream) {                               \
    return haloEntry(AXIS, handle, grid_desc, input, work, dtype, halo_extents, halo_periods, dim, padding, stream); \
  }
CD_DEFINE_HALO(cudecompUpdateHalosX, 0)
CD_DEFINE_HALO(cudecompUpdateHalosY, 1)
CD_DEFINE_HALO(cudecompUpdateHalosZ, 2)

}  // extern "C"
