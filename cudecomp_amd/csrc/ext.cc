// ext.cc -- cudecomp_ext.h: plan introspection and a single-move kernel entry for test harnesses.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "cudecomp_ext.h"
#include "errors.h"
#include "internal.h"
#include "rotate_walk.h"
#include "transport.h"

using namespace cudecomp;

namespace {

cudecompResult_t fail(const Error& e) {
  std::cerr << e.what();
  return e.code();
}

// tests: "the cells between consecutive destination rows are the move's to rewrite" -- the pitch of consecutive rows is the
// smallest destination stride among the dims that are not the unit-stride one
i64 wholeRowsPitchOf(const Move3D& m) {
  i64 pitch = 0;
  for (int i = 0; i < 3; ++i)
    if (m.extent[i] > 1 && m.ds[i] > 1 && (pitch == 0 || m.ds[i] < pitch)) pitch = m.ds[i];
  return pitch;
}

void exportMove(const Move3D& m, cudecompExtMove_t* o) {
  o->src_buf = m.src_buf;
  o->dst_buf = m.dst_buf;
  o->src_off = m.src_off;
  o->dst_off = m.dst_off;
  for (int i = 0; i < 3; ++i) {
    o->extent[i] = m.extent[i];
    o->ss[i] = m.ss[i];
    o->ds[i] = m.ds[i];
  }
  o->peer = m.peer;
  o->row_pitch = (int32_t)std::min<i64>(m.dst_row_pitch, 0x7fffffff);
}

void exportTransposePlan(const TransposePlan& p, const std::vector<int>& global_ranks, cudecompExtTransposePlan_t* out) {
    if (p.nranks > CUDECOMP_EXT_MAX_MEMBERS) CD_NOT_SUPPORTED("communicator too large for cudecompExtTransposePlan_t");
    std::memset(out, 0, sizeof(*out));
    out->noop = p.noop;
    out->exchange = p.exchange;
    out->comm_axis = p.comm_axis;
    out->nranks = p.nranks;
    out->comm_rank = p.comm_rank;
    out->send_buf = p.send_buf;
    out->recv_buf = p.recv_buf;
    out->send_base = p.send_base;
    out->recv_base = p.recv_base;
    out->n_pack = (int32_t)p.pack.size();
    out->n_unpack = (int32_t)p.unpack.size();
    out->rotate = p.rotate;
    for (int i = 0; i < p.nranks; ++i) {
      out->member_global_rank[i] = global_ranks[i];
      if (p.exchange) {
        out->send_cnt[i] = p.send_cnt[i];
        out->send_off[i] = p.send_off[i];
        out->recv_cnt[i] = p.recv_cnt[i];
        out->recv_off[i] = p.recv_off[i];
        out->remote_recv_off[i] = p.remote_recv_off[i];
      }
      out->schedule_dst[i] = p.schedule_dst[i];
    }
    for (size_t i = 0; i < p.pack.size(); ++i) exportMove(p.pack[i], &out->pack[i]);
    for (size_t i = 0; i < p.unpack.size(); ++i) exportMove(p.unpack[i], &out->unpack[i]);
    out->n_direct = (int32_t)p.direct.size();
    for (size_t i = 0; i < p.direct.size(); ++i) exportMove(p.direct[i], &out->direct[i]);
    out->stage_axis = p.stage_axis;
    out->stage_limit = p.stage_limit;
    for (int i = 0; i < p.nranks && p.exchange; ++i) {
      out->send_n[i] = p.send_n[i];
      out->recv_n[i] = p.recv_n[i];
    }
}

void exportHaloPlan(const HaloPlan& p, cudecompExtHaloPlan_t* out) {
    std::memset(out, 0, sizeof(*out));
    out->kind = (int32_t)p.kind;
    out->comm_axis = p.comm_axis;
    out->xbuf = p.xbuf;
    out->face_elements = p.face_elements;
    for (int i = 0; i < 2; ++i) {
      out->neighbor[i] = p.neighbor[i];
      out->send_off[i] = p.send_off[i];
      out->recv_off[i] = p.recv_off[i];
    }
    out->n_pre = (int32_t)p.pre.size();
    out->n_post = (int32_t)p.post.size();
    for (size_t i = 0; i < p.pre.size(); ++i) exportMove(p.pre[i], &out->pre[i]);
    for (size_t i = 0; i < p.post.size(); ++i) exportMove(p.post[i], &out->post[i]);
}

GridShape shapeFromSpec(const cudecompExtGridSpec_t* spec) {
  if (!spec) CD_INVALID_USAGE("grid spec cannot be null");
  GridShape g;
  for (int i = 0; i < 3; ++i) {
    g.gdims[i] = spec->gdims[i];
    g.gdims_dist[i] = spec->gdims_dist[i] > 0 ? spec->gdims_dist[i] : spec->gdims[i];
    if (g.gdims[i] < 1 || g.gdims_dist[i] > g.gdims[i]) CD_INVALID_USAGE("bad gdims / gdims_dist in grid spec");
    bool seen[3] = {false, false, false};
    for (int j = 0; j < 3; ++j) {
      const int v = spec->mem_order[i][j];
      if (v < 0 || v > 2 || seen[v]) CD_INVALID_USAGE("mem_order rows of a grid spec must be permutations of 0,1,2");
      seen[v] = true;
      g.mem_order[i][j] = v;
    }
  }
  g.pdims = {spec->pdims[0], spec->pdims[1]};
  if (g.pdims[0] < 1 || g.pdims[1] < 1) CD_INVALID_USAGE("bad pdims in grid spec");
  g.col_major = spec->col_major != 0;
  return g;
}

}  // namespace

extern "C" {

cudecompResult_t cudecompExtGetTransposePlan(cudecompHandle_t handle, cudecompGridDesc_t gd, int32_t op,
                                             const int32_t in_halo[], const int32_t out_halo[], const int32_t in_pad[],
                                             const int32_t out_pad[], bool inplace, int32_t backend_override,
                                             cudecompExtTransposePlan_t* out) {
  try {
    if (!handle || !handle->initialized) CD_INVALID_USAGE("invalid handle");
    if (!gd || !gd->initialized || gd->handle != handle) CD_INVALID_USAGE("invalid grid descriptor");
    if (!out) CD_INVALID_USAGE("plan argument cannot be null");
    if (op < 0 || op > 3) CD_INVALID_USAGE("op out of range");
    const auto backend =
        backend_override ? (cudecompTransposeCommBackend_t)backend_override : gd->config.transpose_comm_backend;
    TransportTraits traits;
    traits.pipelined = transposeBackendIsPipelined(backend);
    traits.symmetric_recv = usesPeerTransport(handle, backend);
    traits.self_exchange = handle->self_exchange;
    const CommAxis ca = (op == OP_X_TO_Y || op == OP_Y_TO_X) ? COMM_COL : COMM_ROW;
    const cudecompCommInfo& ci = gd->comm(ca);
    const TransposePlan p = buildTransposePlan(gd->shape, handle->rank, (TransposeOp)op, in_halo, out_halo, in_pad,
                                               out_pad, inplace, traits, ci.npergroup);
    exportTransposePlan(p, ci.global_ranks, out);
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtGetHaloPlan(cudecompHandle_t handle, cudecompGridDesc_t gd, int32_t axis,
                                        const int32_t halo[], const bool periods[], int32_t dim, const int32_t pad[],
                                        int32_t backend_override, cudecompExtHaloPlan_t* out) {
  try {
    if (!handle || !handle->initialized) CD_INVALID_USAGE("invalid handle");
    if (!gd || !gd->initialized || gd->handle != handle) CD_INVALID_USAGE("invalid grid descriptor");
    if (!out || !halo) CD_INVALID_USAGE("null argument");
    if (axis < 0 || axis > 2 || dim < 0 || dim > 2) CD_INVALID_USAGE("axis/dim out of range");
    const auto backend = backend_override ? (cudecompHaloCommBackend_t)backend_override : gd->config.halo_comm_backend;
    const int32_t zero[3] = {0, 0, 0};
    const HaloPlan p = buildHaloPlan(gd->shape, handle->rank, axis, dim, halo, periods, pad ? pad : zero,
                                     usesPeerTransport(handle, backend), handle->self_exchange);
    exportHaloPlan(p, out);
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtGetTransposeTimings(cudecompHandle_t handle, cudecompGridDesc_t gd, int32_t op,
                                                cudecompExtTransposeTimings_t* out) {
  try {
    if (!handle || !handle->initialized) CD_INVALID_USAGE("invalid handle");
    if (!gd || !gd->initialized || gd->handle != handle) CD_INVALID_USAGE("invalid grid descriptor");
    if (!out || op < 0 || op > 3) CD_INVALID_USAGE("bad argument");
    const TransposeTimings t = perfCollect(gd, op);
    out->calls = t.calls;
    out->samples = t.samples;
    out->total_ms = t.total_ms;
    out->pack_ms = t.pack_ms;
    out->exchange_ms = t.exchange_ms;
    out->unpack_ms = t.unpack_ms;
    out->pencil_bytes = t.pencil_bytes;
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtGetHaloTimings(cudecompHandle_t handle, cudecompGridDesc_t gd, int32_t axis, int32_t dim,
                                           cudecompExtTransposeTimings_t* out) {
  try {
    if (!handle || !handle->initialized) CD_INVALID_USAGE("invalid handle");
    if (!gd || !gd->initialized || gd->handle != handle) CD_INVALID_USAGE("invalid grid descriptor");
    if (!out || axis < 0 || axis > 2 || dim < 0 || dim > 2) CD_INVALID_USAGE("bad argument");
    const TransposeTimings t = perfCollectHalo(gd, axis, dim);
    out->calls = t.calls;
    out->samples = t.samples;
    out->total_ms = t.total_ms;
    out->pack_ms = t.pack_ms;
    out->exchange_ms = t.exchange_ms;
    out->unpack_ms = t.unpack_ms;
    out->pencil_bytes = t.pencil_bytes;
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtPlanTranspose(const cudecompExtGridSpec_t* grid, int32_t rank, int32_t op,
                                          const int32_t in_halo[], const int32_t out_halo[], const int32_t in_pad[],
                                          const int32_t out_pad[], bool inplace, int32_t pipelined,
                                          int32_t symmetric_recv, int32_t npergroup, cudecompExtTransposePlan_t* out) {
  try {
    const GridShape g = shapeFromSpec(grid);
    if (!out) CD_INVALID_USAGE("plan argument cannot be null");
    if (op < 0 || op > 3) CD_INVALID_USAGE("op out of range");
    if (rank < 0 || rank >= g.pdims[0] * g.pdims[1]) CD_INVALID_USAGE("rank out of range");
    TransportTraits traits;
    traits.pipelined = pipelined != 0;
    traits.symmetric_recv = symmetric_recv != 0;
    const CommAxis ca = (op == OP_X_TO_Y || op == OP_Y_TO_X) ? COMM_COL : COMM_ROW;
    const int P = g.pdims[ca == COMM_COL ? 0 : 1];
    const TransposePlan p = buildTransposePlan(g, rank, (TransposeOp)op, in_halo, out_halo, in_pad, out_pad, inplace,
                                               traits, npergroup > 0 ? npergroup : P);
    std::vector<int> members(P);
    const auto pidx = gridIndexOfRank(g, rank);
    for (int i = 0; i < P; ++i) members[i] = globalRankOf(g, pidx, ca, i);
    exportTransposePlan(p, members, out);
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtPlanRelay(const cudecompExtGridSpec_t* grid, int32_t rank, int32_t op, const int32_t in_halo[],
                                      const int32_t out_halo[], const int32_t in_pad[], const int32_t out_pad[], bool inplace,
                                      cudecompExtRelayPlan_t* out) {
  try {
    const GridShape g = shapeFromSpec(grid);
    if (!out) CD_INVALID_USAGE("plan argument cannot be null");
    if (op < 0 || op > 3) CD_INVALID_USAGE("op out of range");
    const int n = g.pdims[0] * g.pdims[1];
    if (rank < 0 || rank >= n) CD_INVALID_USAGE("rank out of range");
    TransportTraits traits;
    traits.symmetric_recv = true;
    const CommAxis ca = (op == OP_X_TO_Y || op == OP_Y_TO_X) ? COMM_COL : COMM_ROW;
    const RelayPlan rp = buildRelayPlan(g, n, rank, (TransposeOp)op, in_halo, out_halo, in_pad, out_pad, inplace, traits,
                                        g.pdims[ca == COMM_COL ? 0 : 1]);
    std::memset(out, 0, sizeof(*out));
    out->applies = rp.applies ? 1 : 0;
    out->nranks = rp.nranks;
    out->slots_per_source = rp.slots_per_source;
    out->slot_elements = rp.slot_elements;
    out->relay_elements = rp.relayElements();
    if ((int)rp.scatter.size() > CUDECOMP_EXT_MAX_RELAY_MOVES || (int)rp.forward.size() > CUDECOMP_EXT_MAX_RELAY_MOVES)
      CD_NOT_SUPPORTED("relay plan too large for the export structure");
    auto put = [](const std::vector<RelayMove>& v, cudecompExtRelayMove_t* o) {
      for (size_t k = 0; k < v.size(); ++k) o[k] = {v[k].dst_rank, v[k].to_relay ? 1 : 0, v[k].src_off, v[k].dst_off, v[k].count};
    };
    out->n_scatter = (int32_t)rp.scatter.size();
    out->n_forward = (int32_t)rp.forward.size();
    put(rp.scatter, out->scatter);
    put(rp.forward, out->forward);
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtPlanHalo(const cudecompExtGridSpec_t* grid, int32_t rank, int32_t axis, const int32_t halo[],
                                     const bool periods[], int32_t dim, const int32_t pad[], int32_t force_packed,
                                     cudecompExtHaloPlan_t* out) {
  try {
    const GridShape g = shapeFromSpec(grid);
    if (!out || !halo) CD_INVALID_USAGE("null argument");
    if (axis < 0 || axis > 2 || dim < 0 || dim > 2) CD_INVALID_USAGE("axis/dim out of range");
    if (rank < 0 || rank >= g.pdims[0] * g.pdims[1]) CD_INVALID_USAGE("rank out of range");
    const int32_t zero[3] = {0, 0, 0};
    const bool none[3] = {false, false, false};
    const HaloPlan p = buildHaloPlan(g, rank, axis, dim, halo, periods ? periods : none, pad ? pad : zero, force_packed != 0);
    exportHaloPlan(p, out);
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtPencilInfo(const cudecompExtGridSpec_t* grid, int32_t rank, int32_t axis, const int32_t halo[],
                                       const int32_t pad[], cudecompPencilInfo_t* out) {
  try {
    const GridShape g = shapeFromSpec(grid);
    if (!out) CD_INVALID_USAGE("pencil_info argument cannot be null");
    if (axis < 0 || axis > 2) CD_INVALID_USAGE("axis argument out of range");
    if (rank < 0 || rank >= g.pdims[0] * g.pdims[1]) CD_INVALID_USAGE("rank out of range");
    const Pencil p = makePencil(g, gridIndexOfRank(g, rank), axis, halo, pad);
    std::memset(out, 0, sizeof(*out));
    for (int i = 0; i < 3; ++i) {
      out->shape[i] = p.shape[i];
      out->lo[i] = p.lo[i];
      out->hi[i] = p.hi[i];
      out->order[i] = p.order[i];
      out->halo_extents[i] = p.halo[i];
      out->padding[i] = p.pad[i];
    }
    out->size = p.size;
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtShiftedRank(const cudecompExtGridSpec_t* grid, int32_t rank, int32_t axis, int32_t dim,
                                        int32_t displacement, bool periodic, int32_t* shifted_rank) {
  try {
    const GridShape g = shapeFromSpec(grid);
    if (!shifted_rank) CD_INVALID_USAGE("shifted_rank argument cannot be null");
    if (axis < 0 || axis > 2 || dim < 0 || dim > 2) CD_INVALID_USAGE("axis/dim out of range");
    if (rank < 0 || rank >= g.pdims[0] * g.pdims[1]) CD_INVALID_USAGE("rank out of range");
    *shifted_rank = shiftedRank(g, rank, axis, dim, displacement, periodic);
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtWorkspaceSizes(const cudecompExtGridSpec_t* grid, int32_t rank, int32_t axis,
                                           const int32_t halo[], int64_t* transpose_ws, int64_t* halo_ws) {
  try {
    const GridShape g = shapeFromSpec(grid);
    if (axis < 0 || axis > 2) CD_INVALID_USAGE("axis argument out of range");
    if (rank < 0 || rank >= g.pdims[0] * g.pdims[1]) CD_INVALID_USAGE("rank out of range");
    if (transpose_ws) *transpose_ws = transposeWorkspaceElements(g);
    if (halo_ws) {
      if (!halo) CD_INVALID_USAGE("halo_extents argument cannot be null");
      *halo_ws = haloWorkspaceElements(g, gridIndexOfRank(g, rank), axis, halo);
    }
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtGetCounters(cudecompHandle_t handle, cudecompGridDesc_t gd, cudecompExtCounters_t* out) {
  try {
    if (!handle || !handle->initialized) CD_INVALID_USAGE("invalid handle");
    if (!gd || !gd->initialized || gd->handle != handle) CD_INVALID_USAGE("invalid grid descriptor");
    if (!out) CD_INVALID_USAGE("null argument");
    out->graphs_captured = (int64_t)(gd->pack_graphs.size() + gd->op_graphs.size());
    out->graph_launches = gd->graph_launches;
    out->local = gd->path_count[PATH_LOCAL];
    out->rccl = gd->path_count[PATH_RCCL];
    out->mpi = gd->path_count[PATH_MPI];
    out->peer_barrier = gd->path_count[PATH_PEER_BARRIER];
    out->peer_fused = gd->path_count[PATH_PEER_FUSED];
    out->peer_pipelined = gd->path_count[PATH_PEER_PIPELINED];
    out->direct_puts = gd->direct_puts;
    peerPoolCounters(handle, &out->workspace_pool_hits, &out->stale_ipc_mappings, &out->workspace_pool_bytes, &out->retired_imports);
    // (the LAST census -- taken when the transport came up or by cudecompExtQueueCensus -- not a fresh one: reading the
    // driver's tables per call slows ranks that share a device)
    out->compute_queues_on_device = handle->census_compute_queues;
    out->hardware_queue_slots = handle->census_queue_slots;
    out->relayed = gd->relayed;
    out->rotations = gd->rotations;
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtQueueCensus(cudecompHandle_t handle, int32_t* compute_queues, int32_t* slots) {
  try {
    if (!handle || !handle->initialized) CD_INVALID_USAGE("invalid handle");
    int s = 0;
    const int c = peerQueueCensus(handle, false, &s);
    if (compute_queues) *compute_queues = c;
    if (slots) *slots = s;
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtTrimWorkspacePool(cudecompHandle_t handle) {
  try {
    if (!handle || !handle->initialized) CD_INVALID_USAGE("invalid handle");
    workspaceTrimPool(handle);
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

const char* cudecompExtLastKernelName(void) { return lastKernelName(); }

cudecompResult_t cudecompExtEstimateCycleMs(cudecompHandle_t handle, const cudecompExtGridSpec_t* grid, int32_t es,
                                            int32_t backend, int32_t library_buffers, int32_t inplace, double* ms) {
  try {
    if (!handle || !handle->initialized) CD_INVALID_USAGE("invalid handle");
    if (!grid || !ms) CD_INVALID_USAGE("null argument");
    if (es != 4 && es != 8 && es != 16) CD_INVALID_USAGE("element size must be 4, 8 or 16");
    if (backend < 1 || backend > 8) CD_INVALID_USAGE("backend out of range");
    const GridShape g = shapeFromSpec(grid);
    const bool ip[4] = {inplace != 0, inplace != 0, inplace != 0, inplace != 0};
    *ms = estimateTransposeCycleMs(handle, g, es, (cudecompTransposeCommBackend_t)backend, library_buffers != 0, ip);
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtGetLinkInfo(cudecompHandle_t handle, cudecompExtLinkInfo_t* out) {
  try {
    if (!handle || !handle->initialized) CD_INVALID_USAGE("invalid handle");
    if (!out) CD_INVALID_USAGE("null argument");
    out->gbps_sdma = handle->link_gbps_sdma;
    out->gbps_cu = handle->link_gbps_cu;
    out->measured = (handle->link_gbps_sdma > 0 || handle->link_gbps_cu > 0) ? 1 : 0;
    out->crosses_devices = handle->link_crosses_devices ? 1 : 0;
    out->copy_engine = handle->peer_copy_engine;
    out->reserved = 0;
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtPeerProbe(cudecompHandle_t handle, void* buffer, size_t bytes, int32_t* mismatches) {
  try {
    if (!handle || !handle->initialized) CD_INVALID_USAGE("invalid handle");
    if (!buffer || !mismatches) CD_INVALID_USAGE("null argument");
    *mismatches = peerProbe(handle, buffer, bytes);
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtRunLocalPhases(const cudecompExtGridSpec_t* grid, int32_t rank, int32_t op, int32_t pipelined,
                                           int32_t symmetric_recv, int32_t phases, void* input, void* output, void* work,
                                           int32_t es, hipStream_t stream) {
  try {
    const GridShape g = shapeFromSpec(grid);
    if (op < 0 || op > 3) CD_INVALID_USAGE("op out of range");
    if (rank < 0 || rank >= g.pdims[0] * g.pdims[1]) CD_INVALID_USAGE("rank out of range");
    if (es != 4 && es != 8 && es != 16) CD_INVALID_USAGE("element size must be 4, 8 or 16");
    if (!input || !output || !work) CD_INVALID_USAGE("null buffer");
    TransportTraits traits;
    traits.pipelined = pipelined != 0;
    traits.symmetric_recv = symmetric_recv != 0;
    const CommAxis ca = (op == OP_X_TO_Y || op == OP_Y_TO_X) ? COMM_COL : COMM_ROW;
    const int P = g.pdims[ca == COMM_COL ? 0 : 1];
    const int32_t zero[3] = {0, 0, 0};
    const TransposePlan p = buildTransposePlan(g, rank, (TransposeOp)op, zero, zero, zero, zero, input == output, traits, P);
    void* bufs[3] = {input, output, work};
    KernelTuning t;
#ifdef CUDECOMP_TUNING_VARIANTS
    if (const char* v = std::getenv("CUDECOMP_INTERLEAVE_ROWS")) t.interleave_rows = (int)std::strtol(v, nullptr, 10);
#endif
    // the launches of the executor: one batched launch per phase; pipelined: one launch per STAGE with all peers in it
    // for the one-sided transport (transport.cc: peerStagedExchange), one launch per PEER for RCCL / MPI (the reference's
    // pipeline, transpose.h:470-513, 683-744)
    int stages = 4;
    if (const char* v = std::getenv("CUDECOMP_PIPELINE_STAGES")) stages = (int)std::strtol(v, nullptr, 10);
    i64 min_stage = (i64)8 << 20;
    if (const char* v = std::getenv("CUDECOMP_PIPELINE_MIN_STAGE_MIB")) min_stage = std::strtoll(v, nullptr, 10) << 20;
    const int K = stageCount(p, stages, es, min_stage);
    auto run = [&](const std::vector<Move3D>& moves) {
      if (moves.empty()) return;
      if (traits.pipelined && traits.symmetric_recv && p.exchange) {
        std::vector<Move3D> part;
        for (int k = 0; k < K; ++k) {
          part.clear();
          for (const Move3D& m : moves) part.push_back(stageOfMove(m, p.stage_axis, k, K));
          launchMoves(part.data(), (int)part.size(), bufs, es, stream, &t);
        }
      } else if (traits.pipelined) {
        for (const Move3D& m : moves) launchMoves(&m, 1, bufs, es, stream, &t);
      } else {
        launchMoves(moves.data(), (int)moves.size(), bufs, es, stream, &t);
      }
    };
    if (phases & 1) run(p.pack);
    if (phases & 2) run(p.unpack);
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtRotateWalk(int32_t nb, int32_t walk, int64_t first, int64_t count, int32_t* blocks, int64_t* grid) {
  try {
    if (nb < 1 || nb > 1024 || !grid || first < 0 || count < 0 || (count > 0 && !blocks)) CD_INVALID_USAGE("bad argument");
    const int w = rotateWalkFor(walk, nb);
    *grid = rotateWalkGrid(nb, w);
    if (first + count > *grid) CD_INVALID_USAGE("workgroups beyond the grid");
    for (int64_t i = 0; i < count; ++i) {
      int b0 = -1, b1 = -1, b2 = -1;
      if (!rotateWalkBlock((unsigned int)(first + i), (unsigned int)nb, w, &b0, &b1, &b2)) b0 = b1 = b2 = -1;
      blocks[3 * i] = b0, blocks[3 * i + 1] = b1, blocks[3 * i + 2] = b2;
    }
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtDescribeMove(uint64_t src_address, uint64_t dst_address, int32_t es, const int64_t extent[3],
                                        const int64_t ss[3], const int64_t ds[3], int32_t flags, int64_t out[10]) {
  try {
    if (!extent || !ss || !ds || !out) CD_INVALID_USAGE("null argument");
    if (es != 4 && es != 8 && es != 16) CD_INVALID_USAGE("element size must be 4, 8 or 16");
    Move3D m;
    for (int i = 0; i < 3; ++i) {
      m.extent[i] = extent[i];
      m.ss[i] = ss[i];
      m.ds[i] = ds[i];
    }
    if (flags & 256) m.dst_row_pitch = wholeRowsPitchOf(m);
    if (flags >> 12) m.dst_row_pitch = flags >> 12;  // the planner's own word (cudecompExtMove_t::row_pitch)
    KernelTuning t;
    if (flags & 2) t.force_streaming = true;
    if (flags & 4) t.window_mode = 1;
    if (flags & 8) t.dense_rows = 0;  // as CUDECOMP_PRESERVE_OUTPUT_HALOS=1
    if (flags & 64) t.walk_order = 0;
    if (flags & 128) t.walk_order = 1;
    long long o[10];
    describeMove(m, reinterpret_cast<const void*>(src_address), reinterpret_cast<void*>(dst_address), es, &t, o);
    for (int i = 0; i < 10; ++i) out[i] = o[i];
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

cudecompResult_t cudecompExtMove3D(const void* src, void* dst, int32_t es, const int64_t extent[3],
                                   const int64_t ss[3], const int64_t ds[3], int32_t force_generic,
                                   int32_t* kernel_class, hipStream_t stream) {
  try {
    if (!src || !dst || !extent || !ss || !ds) CD_INVALID_USAGE("null argument");
    if (es != 4 && es != 8 && es != 16) CD_INVALID_USAGE("element size must be 4, 8 or 16");
    Move3D m;
    m.src_buf = BUF_IN;
    m.dst_buf = BUF_OUT;
    for (int i = 0; i < 3; ++i) {
      m.extent[i] = extent[i];
      m.ss[i] = ss[i];
      m.ds[i] = ds[i];
    }
    if (force_generic & 256) m.dst_row_pitch = wholeRowsPitchOf(m);  // the cells between consecutive destination rows are the move's to rewrite
    void* bufs[3] = {const_cast<void*>(src), dst, nullptr};
    KernelTuning t;
    if (force_generic & 1) t.force_class = MOVE_GENERIC;
    if (force_generic & 2) t.force_streaming = true;
    if (force_generic & 4) t.window_mode = 1;  // window kernel whenever the destination rows are off the 64-byte grid
    if (force_generic & 8) t.window_mode = 0;  // never
    if (force_generic & 16) t.tile_shape = 1;  // 4-byte transposes: 128 x 64 tiles
    if (force_generic & 32) t.tile_shape = 0;  // 4-byte transposes: 64 x 64 tiles (the default is 64 x 128)
    if (force_generic & 64) t.walk_order = 0;  // transposes: i first
    if (force_generic & 128) t.walk_order = 1;  // transposes: j first (no runs)
    KernelStats st;
    launchMoves(&m, 1, bufs, es, stream, &t, &st);
    if (kernel_class) {
      *kernel_class = -1;
      for (int c = 0; c < MOVE_CLASS_COUNT; ++c)
        if (st.launches[c]) *kernel_class = c;
    }
  } catch (const Error& e) {
    return fail(e);
  } catch (...) {
    return CUDECOMP_RESULT_INTERNAL_ERROR;
  }
  return CUDECOMP_RESULT_SUCCESS;
}

}  // extern "C"
